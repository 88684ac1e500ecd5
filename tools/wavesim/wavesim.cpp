// wavesim runtime: lane fibers, the wave / workgroup rendezvous scheduler, the grid launcher and the host-API stand-ins.
// See wavesim.h.  Test infrastructure only.
#include "wavesim.h"

#include <stdarg.h>
#include <sys/mman.h>

#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

// ---- context switch (x86-64 SysV): saves the callee-saved registers on the current stack, swaps stack pointers ----------
extern "C" void wavesim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl wavesim_switch
.type wavesim_switch,@function
wavesim_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size wavesim_switch,.-wavesim_switch
)");

// AddressSanitizer build (build_sim.py --asan): stack switches are announced so that ASan follows the fibers' stacks
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define WAVESIM_ASAN 1
#endif
#endif
#ifdef WAVESIM_ASAN
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#define WAVESIM_ASAN_START(save, bottom, size) __sanitizer_start_switch_fiber(save, bottom, size)
#define WAVESIM_ASAN_FINISH(save, bottom_old, size_old) __sanitizer_finish_switch_fiber(save, bottom_old, size_old)
#else
#define WAVESIM_ASAN_START(save, bottom, size) ((void)0)
#define WAVESIM_ASAN_FINISH(save, bottom_old, size_old) ((void)0)
#endif

// ThreadSanitizer build (build_sim.py --tsan): the KERNEL translation units are instrumented, this file is not.  Every lane
// is a TSan fiber; switches carry no synchronisation, so the only happens-before edges inside a launch are the ones the
// hardware gives: a workgroup barrier orders the workgroup's lanes, a wave-level operation (or CACO_WAVE_LDS_SYNC) orders the
// lanes of that wave, the end of a workgroup orders it before the next workgroup on the same worker.  Two lanes of different
// waves that touch the same LDS bytes with no barrier in between, or two workgroups that touch the same global bytes, are
// reported as a data race with both source lines.
#ifdef WAVESIM_TSAN
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
void __tsan_acquire(void* addr);
void __tsan_release(void* addr);
}
#define TSAN_SWITCH(f) __tsan_switch_to_fiber(f, 1u /* no_sync */)
#define TSAN_ACQUIRE(a) __tsan_acquire(a)
#define TSAN_RELEASE(a) __tsan_release(a)
#else
#define TSAN_SWITCH(f) ((void)0)
#define TSAN_ACQUIRE(a) ((void)0)
#define TSAN_RELEASE(a) ((void)0)
#endif

namespace wavesim {

WAVESIM_TLS Lane* cur = nullptr;
WAVESIM_TLS idx3 block_idx = {0, 0, 0}, block_dim = {1, 1, 1}, grid_dim = {1, 1, 1};
int dma_late = [] { const char* e = getenv("WAVESIM_DMA"); return (e && !strcmp(e, "eager")) ? 0 : 1; }();
// WAVESIM_ORDER=reverse: waves (and the lanes inside a wave) are run last to first between rendezvous points.  Results must
// not depend on it: a kernel that only works because wave 0 happens to run first (a missing barrier, a read of another
// wave's LDS data before its covering wait) gives different bytes under the other order.
// WAVESIM_ORDER=random:<seed>: a fresh permutation of the waves for every pass of the scheduler (every interval between two
// workgroup barriers), lanes forward or backward at random.
static int order_mode(unsigned* seed) {         // 0 forward, 1 reverse, 2 random
  const char* e = getenv("WAVESIM_ORDER");      // read per launch: tests flip it at run time
  if (!e) return 0;
  if (!strcmp(e, "reverse")) return 1;
  if (!strncmp(e, "random", 6)) { *seed = e[6] == ':' ? (unsigned)atoi(e + 7) : 1u; return 2; }
  return 0;
}
static inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

namespace {

constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;
constexpr size_t DYN_LDS = 160 * 1024;

struct Worker {                 // per OS thread: fiber stacks, lane / wave records, the dynamic LDS block
  char* stacks = nullptr;
  Lane* lanes = nullptr;
  Wave* waves = nullptr;
  char* lds = nullptr;                          // this launch's dynamic LDS (inside lds_region)
  char* lds_region = nullptr;
  size_t lds_bytes = 0;
  void* main_sp = nullptr;
  const void* main_stack_bottom = nullptr;      // ASan: the scheduler's own stack, learnt at the first switch into a lane
  size_t main_stack_size = 0;
  const std::function<void()>* body = nullptr;
  int bar_gen = 0;                              // completed workgroup barriers of the running block
  int bar_or[2] = {0, 0};                       // __syncthreads_or accumulators, by barrier parity
  void* main_fiber = nullptr;                   // TSan: the scheduler's own context and one fiber per lane of the running block
  void** fibers = nullptr;
  char tag_block = 0, tag_epoch[2] = {0, 0};    // TSan: addresses the barrier / workgroup-boundary edges hang on
  unsigned serial = 0;                          // workgroups this worker has run (parity picks the epoch tag)
  char tag_wave[MAX_THREADS / 64] = {};
  ~Worker() {
    if (stacks) munmap(stacks, STACK_BYTES * MAX_THREADS);
    free(lanes);
    free(waves);
    if (lds_region) munmap(lds_region, DYN_LDS + 4096);
  }
  void ensure() {
    if (stacks) return;
    stacks = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == MAP_FAILED) fail("mmap of fiber stacks failed");
    lanes = (Lane*)calloc(MAX_THREADS, sizeof(Lane));
    waves = (Wave*)aligned_alloc(64, sizeof(Wave) * (MAX_THREADS / 64));
    // dynamic LDS: DYN_LDS bytes followed by an inaccessible page.  A launch that declares n bytes gets the LAST n bytes
    // before that page, so a kernel that touches more dynamic LDS than it asked for faults here as it would on the device.
    lds_region = (char*)mmap(nullptr, DYN_LDS + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (lds_region == MAP_FAILED) fail("mmap of the LDS region failed");
    mprotect(lds_region + DYN_LDS, 4096, PROT_NONE);
    lds = lds_region;
  }
};
Worker g_workers[65];                 // slot 0: the launching thread itself; never torn down (reused by every launch)
WAVESIM_TLS Worker* tl_w = &g_workers[0];

void lane_entry() {
  Worker& W = *tl_w;
  WAVESIM_ASAN_FINISH(nullptr, &W.main_stack_bottom, &W.main_stack_size);
  TSAN_ACQUIRE(&W.tag_epoch[(W.serial ^ 1) & 1]);      // after the PREVIOUS workgroup on this worker (its LDS, stacks and statics are
                                                       // reused), never after a lane of this one that happened to finish first
  (*W.body)();
  Lane* L = cur;
  L->state = 3;
  while (L->vm_count) vm_retire_one(L);     // outstanding DMA still lands (nobody can observe it any more)
  TSAN_RELEASE(&W.tag_epoch[W.serial & 1]);
  WAVESIM_ASAN_START(nullptr, W.main_stack_bottom, W.main_stack_size);      // null: this fiber never comes back
  TSAN_SWITCH(W.main_fiber);
  wavesim_switch(&L->sp, W.main_sp);
  fail("resumed a finished lane");
}

void yield_to_scheduler() {
  Lane* L = cur;
  Worker& W = *tl_w;
  void* fake = nullptr;
  (void)fake;
  WAVESIM_ASAN_START(&fake, W.main_stack_bottom, W.main_stack_size);
  TSAN_SWITCH(W.main_fiber);
  wavesim_switch(&L->sp, W.main_sp);
  WAVESIM_ASAN_FINISH(fake, nullptr, nullptr);
}

void run_block(idx3 bidx, idx3 bdim, idx3 gdim, const std::function<void()>& body, size_t lds_bytes) {
  Worker& W = *tl_w;
  W.ensure();
  W.body = &body;
  W.lds_bytes = (lds_bytes + 15) & ~(size_t)15;
  W.lds = W.lds_region + DYN_LDS - W.lds_bytes;
  memset(W.lds, 0xff, W.lds_bytes);             // LDS is not initialised on the device: unwritten bytes read as NaN patterns
  block_idx = bidx;
  block_dim = bdim;
  grid_dim = gdim;
  const int nthreads = (int)(bdim.x * bdim.y * bdim.z);
  if (nthreads > MAX_THREADS || nthreads <= 0) fail("workgroup of %d threads", nthreads);
  const int nwaves = (nthreads + 63) / 64;
  for (int w = 0; w < nwaves; ++w) {
    W.waves[w].nlanes = 0;
    W.waves[w].gen = 0;
  }
  for (int t = 0; t < nthreads; ++t) {
    Lane& L = W.lanes[t];
    L.stack = W.stacks + (size_t)t * STACK_BYTES;
    L.tid.x = (unsigned)t % bdim.x;
    L.tid.y = ((unsigned)t / bdim.x) % bdim.y;
    L.tid.z = (unsigned)t / (bdim.x * bdim.y);
    L.lane = t & 63;
    L.state = 0;
    L.site = nullptr;
    L.wave = &W.waves[t >> 6];
    L.vm_head = L.vm_count = 0;
    L.wave->lanes[L.lane] = &L;
    L.wave->nlanes = L.lane + 1;
    // initial frame: six callee-saved registers, then the entry address, then a pad so that the entry sees rsp % 16 == 8
    uintptr_t top = ((uintptr_t)L.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** f = reinterpret_cast<void**>(top);
    f[-1] = nullptr;
    f[-2] = reinterpret_cast<void*>(&lane_entry);
    for (int i = 3; i <= 8; ++i) f[-i] = nullptr;
    L.sp = &f[-8];
  }
  W.bar_gen = 0;
  W.bar_or[0] = W.bar_or[1] = 0;
#ifdef WAVESIM_TSAN
  W.main_fiber = __tsan_get_current_fiber();
  if (!W.fibers) W.fibers = (void**)calloc(MAX_THREADS, sizeof(void*));
  for (int t = 0; t < nthreads; ++t) W.fibers[t] = __tsan_create_fiber(0);
  ++W.serial;
  TSAN_RELEASE(&W.tag_epoch[(W.serial ^ 1) & 1]);      // the launching side's writes (and the LDS fill above) come before every lane
#endif
  unsigned seed = 1;
  const int mode = order_mode(&seed);
  seed = seed * 2654435761u + bidx.x * 97u + bidx.y * 7919u + bidx.z * 104729u;
  int perm[MAX_THREADS / 64];
  int live = nthreads;
  while (live > 0) {
    bool progress = false;
    int at_barrier = 0;
    live = 0;
    for (int i = 0; i < nwaves; ++i) perm[i] = mode == 1 ? nwaves - 1 - i : i;
    if (mode == 2)
      for (int i = nwaves - 1; i > 0; --i) { const int j = (int)(lcg(seed) % (unsigned)(i + 1)); const int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    for (int wi = 0; wi < nwaves; ++wi) {
      const int w = perm[wi];
      Wave& wv = W.waves[w];
      const int rev = mode == 1 || (mode == 2 && (lcg(seed) & 1));
      for (;;) {
        for (int li = 0; li < wv.nlanes; ++li) {
          const int l = rev ? wv.nlanes - 1 - li : li;
          Lane* L = wv.lanes[l];
          while (L->state == 0) {
            cur = L;
            void* fake = nullptr;
            (void)fake;
            WAVESIM_ASAN_START(&fake, L->stack, STACK_BYTES);
#ifdef WAVESIM_TSAN
            TSAN_SWITCH(W.fibers[L - W.lanes]);
#endif
            wavesim_switch(&W.main_sp, L->sp);
            WAVESIM_ASAN_FINISH(fake, nullptr, nullptr);
            progress = true;
          }
        }
        int n_op = 0, n_bar = 0;
        const void* site = nullptr;
        bool same = true;
        for (int l = 0; l < wv.nlanes; ++l) {
          Lane* L = wv.lanes[l];
          if (L->state == 1) {
            if (n_op++ == 0) site = L->site;
            else same = same && site == L->site;
          } else if (L->state == 2) ++n_bar;
        }
        if (n_op == 0) { at_barrier += n_bar; live += n_bar; break; }
        if (n_bar) fail("block (%u,%u,%u) wave %d: %d lanes wait at a wave operation while %d wait at the workgroup barrier", bidx.x, bidx.y, bidx.z, w, n_op, n_bar);
        if (!same) fail("block (%u,%u,%u) wave %d: lanes arrived at different wave-operation call sites (divergent control flow around a cross-lane operation)", bidx.x, bidx.y, bidx.z, w);
        unsigned long long act = 0;
        for (int l = 0; l < wv.nlanes; ++l)
          if (wv.lanes[l]->state == 1) { wv.lanes[l]->state = 0; act |= 1ull << l; }
        wv.active[wv.gen & 1] = act;
        ++wv.gen;
      }
    }
    if (at_barrier > 0) {
      for (int t = 0; t < nthreads; ++t)
        if (W.lanes[t].state == 2) W.lanes[t].state = 0;
      ++W.bar_gen;
      W.bar_or[W.bar_gen & 1] = 0;              // the accumulator of the NEXT barrier; the released lanes read the other one
      progress = true;
    }
    if (!progress && live > 0) fail("deadlock in block (%u,%u,%u)", bidx.x, bidx.y, bidx.z);
  }
  cur = nullptr;
#ifdef WAVESIM_TSAN
  TSAN_ACQUIRE(&W.tag_epoch[W.serial & 1]);     // the worker thread (and, through its join, the host) sees what the lanes wrote
  for (int t = 0; t < nthreads; ++t) __tsan_destroy_fiber(W.fibers[t]);
#endif
}

}  // namespace

void wave_rendezvous(const void* site) {
  Lane* L = cur;
  L->site = site;
  L->state = 1;
  [[maybe_unused]] char* tag = &tl_w->tag_wave[L->wave - tl_w->waves];
  TSAN_RELEASE(tag);
  yield_to_scheduler();
  TSAN_ACQUIRE(tag);
}

void block_barrier() {
  cur->state = 2;
  TSAN_RELEASE(&tl_w->tag_block);
  yield_to_scheduler();
  TSAN_ACQUIRE(&tl_w->tag_block);
}

int block_barrier_or(int pred) {
  Worker& W = *tl_w;
  const int slot = W.bar_gen & 1;
  if (pred) W.bar_or[slot] = 1;
  block_barrier();
  return W.bar_or[slot];
}

char* dyn_lds() { return tl_w->lds; }

void s_waitcnt(const char* text) {
  const char* p = strstr(text, "vmcnt(");
  if (p) vm_wait(atoi(p + 6));
}

void fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "[wavesim] ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  abort();
}

static std::mutex g_attr_mu;
static std::map<const void*, int> g_attr_lds;
void declare_dyn_lds(const void* fn, int bytes) {
  std::lock_guard<std::mutex> lk(g_attr_mu);
  g_attr_lds[fn] = bytes;
}

void launch(idx3 grid, idx3 block, size_t lds_bytes, const std::function<void()>& body, const char* what, const void* fn) {
  if (lds_bytes > DYN_LDS) fail("launch with %zu bytes of dynamic LDS", lds_bytes);
  if (lds_bytes > 64 * 1024 && fn) {      // above the default limit the kernel must have been given the attribute, at least this large
    std::lock_guard<std::mutex> lk(g_attr_mu);
    auto it = g_attr_lds.find(fn);
    if (it == g_attr_lds.end() || (size_t)it->second < lds_bytes)
      fail("launch of %s with %zu bytes of dynamic LDS but hipFuncAttributeMaxDynamicSharedMemorySize = %d", what, lds_bytes,
           it == g_attr_lds.end() ? 0 : it->second);
  }
  if (const char* tr = getenv("WAVESIM_TRACE"))            // WAVESIM_TRACE=1: one line per launch on stderr
    if (atoi(tr)) fprintf(stderr, "[wavesim] launch %s grid (%u,%u,%u) block %u lds %zu\n", what, grid.x, grid.y, grid.z, block.x, lds_bytes);
  const long nblocks = (long)grid.x * grid.y * grid.z;
  static const int max_threads = [] {
    const char* e = getenv("WAVESIM_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : (n > 64 ? 64 : n);
  }();
  const int nthr = (int)std::min<long>(max_threads, nblocks);
  std::atomic<long> next{0};
  auto work = [&](int slot) {
    tl_w = &g_workers[slot];
#ifdef WAVESIM_TSAN
    long mine = slot > 0 ? slot - 1 : 0;        // static round-robin: neighbouring workgroups always run on different workers, and
#endif                                           // workgroups of different workers are never ordered (those of one worker are)
    for (;;) {
#ifdef WAVESIM_TSAN
      const long b = mine;
      mine += nthr > 1 ? nthr : 1;
#else
      const long b = next.fetch_add(1);
#endif
      if (b >= nblocks) break;
      idx3 bi;
      bi.x = (unsigned)(b % grid.x);
      bi.y = (unsigned)((b / grid.x) % grid.y);
      bi.z = (unsigned)(b / ((long)grid.x * grid.y));
      run_block(bi, block, grid, body, lds_bytes);
    }
  };
  static std::mutex launch_mu;           // one launch at a time: the worker slots are shared
  std::lock_guard<std::mutex> lk(launch_mu);
  if (nthr <= 1) { work(0); return; }
  std::vector<std::thread> pool;
  for (int i = 0; i < nthr; ++i) pool.emplace_back(work, i + 1);
  for (auto& t : pool) t.join();
}

}  // namespace wavesim

// ---- host API stand-ins ------------------------------------------------------------------------------------------------
hipError_t hipMalloc(void** p, size_t bytes) {
  void* q = nullptr;
  if (posix_memalign(&q, 256, bytes ? bytes : 1)) return hipErrorOutOfMemory;
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  free(p);
  return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  const char* e = getenv("WAVESIM_CUS");
  p->multiProcessorCount = e ? atoi(e) : 16;
  snprintf(p->name, sizeof(p->name), "wavesim (gfx950 functional model)");
  return hipSuccess;
}
