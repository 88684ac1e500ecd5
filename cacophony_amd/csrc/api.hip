// C-ABI layer of libcaco_hip.so (include/caco_hip.h): model lifetime, weight ingestion by reference
// state-dict key, workspace arena, and the forward orchestration that strings the kernels of
// gemm.hip / attention.hip / norm.hip / pool.hip / mel.hip together on the caller's stream.
// No exception leaves this file; every entry point returns a status and records a message.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <climits>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace caco {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return CACO_OK;
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return CACO_ERR_HIP;
}

// run-time switches (kernels.h): environment read once per switch, then caco_set_switch only
struct SwitchSlot {
  const char* name;
  int def;
  std::atomic<int> value;
};
static SwitchSlot g_switches[SW_COUNT] = {
    {"CACO_PINGPONG", 0, {INT_MIN}},     {"CACO_POS_FUSE", 0, {INT_MIN}},  {"CACO_POOL_FUSE", 0, {INT_MIN}},
    {"CACO_ATTN_SMALL", 0, {INT_MIN}},   {"CACO_ATTN_ROWS", 64, {INT_MIN}}, {"CACO_W_NGROUP", -1, {INT_MIN}},
    {"CACO_W8_MIN_TILES", 128, {INT_MIN}}, {"CACO_W4H_MAX_TILES", 0, {INT_MIN}},
};
// per-switch ranges, one rule for both ways in (environment, caco_set_switch): INT_MIN is the "environment not read yet" sentinel and
// must never be stored (it would re-arm the getenv), and an out-of-range value would go straight to a launch path
static bool switch_value_ok(Switch s, int value) {
  if (value == INT_MIN) return false;
  switch (s) {
    case SW_PINGPONG: case SW_POS_FUSE: case SW_POOL_FUSE: case SW_ATTN_SMALL: return value == 0 || value == 1;
    case SW_ATTN_ROWS: return value == 32 || value == 64;
    case SW_W_NGROUP: return value >= -1 && value <= 4096;
    case SW_W8_MIN_TILES: case SW_W4H_MAX_TILES: return value >= 0;
    default: return true;
  }
}
int sw(Switch s) {
  SwitchSlot& slot = g_switches[s];
  int v = slot.value.load(std::memory_order_relaxed);
  if (v == INT_MIN) {                      // first use in this process: the environment's value, or the default
    const char* e = getenv(slot.name);
    v = e && *e ? atoi(e) : slot.def;
    if (!switch_value_ok(s, v)) {          // same ranges as caco_set_switch: a bad environment value never reaches a launch path
      fprintf(stderr, "[cacophony_amd] %s=%s is out of range: using the default %d\n", slot.name, e ? e : "", slot.def);
      v = slot.def;
    }
    int expect = INT_MIN;
    if (!slot.value.compare_exchange_strong(expect, v, std::memory_order_relaxed)) v = expect;
  }
  return v;
}

// per (kernel, device): dynamic-LDS attribute set once; per device: CU count (kernels.h)
int prepare_launch(const void* kern, int dyn_lds_bytes, int* num_cu) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> attr_bytes;     // largest size declared so far (a kernel's LDS may depend on the shape)
  static int cus[CACO_MAX_DEVICES] = {};
  int dev = 0;
  CACO_HIP(hipGetDevice(&dev));
  CACO_REQUIRE(dev >= 0 && dev < CACO_MAX_DEVICES, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lk(mu);
  if (dyn_lds_bytes > 0) {
    int& have = attr_bytes[std::make_pair(kern, dev)];
    if (dyn_lds_bytes > have) {
      CACO_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_lds_bytes));
      have = dyn_lds_bytes;
    }
  }
  if (num_cu) {
    if (!cus[dev]) {
      hipDeviceProp_t prop;
      CACO_HIP(hipGetDeviceProperties(&prop, dev));
      cus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    *num_cu = cus[dev];
  }
  return CACO_OK;
}

// ------------------------------------------------------------------------------------------------
// optional per-stage timing: HIP event pairs recorded on the caller's stream around each launch group.
// Process-global by design (one report for everything that ran while it was on); the record list is guarded by a
// mutex so that forwards on several host threads may run while it is enabled.
// ------------------------------------------------------------------------------------------------
struct ProfRec { const char* name; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;

static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
struct ProfScope {
  hipStream_t st; bool on; hipEvent_t b = nullptr;
  ProfScope(const char* name, hipStream_t s) : st(s), on(g_prof_on) {
    if (on) {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      ProfRec r{name, prof_event(), prof_event()};
      b = r.b;
      (void)hipEventRecord(r.a, st);
      g_prof_recs.push_back(r);
    }
  }
  ~ProfScope() { if (on) (void)hipEventRecord(b, st); }
};
#define CACO_STAGE(name, expr)            \
  do {                                    \
    int _rc;                              \
    { caco::ProfScope _ps(name, st); _rc = (expr); } \
    if (_rc) return _rc;                  \
  } while (0)

namespace {

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct Lin {          // bf16 weight [out, in] + fp32 bias
  bf16_t* w = nullptr;
  float* b = nullptr;
  int out = 0, in = 0;
};
struct LNp {
  float* g = nullptr;
  float* b = nullptr;
};
struct LinFold {      // Linear with the preceding LayerNorm folded in: w = bf16(gamma * W), b = W.beta + bias,
  Lin lin;            // c1[n] = sum_k w[n,k]:  LN(x) W^T + bias = rstd * (x w^T - mean * c1) + b
  float* c1 = nullptr;
};
struct AudioLayer {   // AudioEncoderLayer, audio_models/mae.py:64-99
  LNp ln1, ln2;
  Lin qkv, o, fc1, fc2;
  LinFold qkv_f, fc1_f;
};
struct AudioStack {   // AudioEncoder / AudioDecoder trunk
  Lin input_proj;
  float* freq_table = nullptr;   // [num_freq, H]
  float* restore_patch = nullptr;
  std::vector<AudioLayer> layers;
  LNp norm;
  Lin output_proj;               // decoder only
};
struct TextLayer {    // RobertaLayer, text_models/roberta.py:181-215
  Lin qkv, attn_out, inter, out;
  LNp ln_attn, ln_out;
};
struct DecLayer {     // RobertaLayer(has_cross_attention=True), text_models/roberta.py:181-215
  TextLayer t;
  Lin cq, ckv, cattn_out;   // crossattention.self.query | key;value (fused) | crossattention.output.dense
  LNp ln_cross;
};

}  // namespace
}  // namespace caco

using namespace caco;

struct caco_model {
  caco_config cfg;
  int device = 0;                    // the device the model was created on; every entry point checks it is current
  int ln_fold = 0;                   // LayerNorm folding mode of the audio stack (caco_model_set_ln_fold)
  bool finalized = false;
  std::map<std::string, HostTensor> pending;
  std::vector<void*> owned;          // every device allocation holding weights
  // audio tower + pooler (caco.py:100-107)
  AudioStack enc, dec;
  // pooler with both projections folded out of the token dimension (pool.hip): wq = s * Wk_h^T q_h per head [heads, H];
  // value projection as fp32 [H, H] + bias, applied to the pooled rows
  float* pool_wq = nullptr;
  float *pool_v_w = nullptr, *pool_v_b = nullptr;
  float *pool_out_w = nullptr, *pool_out_b = nullptr;   // fp32 [proj, H]
  // text tower (caco.py:110-113)
  float *word = nullptr, *pos = nullptr, *type0 = nullptr;
  LNp emb_ln;
  std::vector<TextLayer> tlayers;
  float* tpool_wq = nullptr;                            // [1, H]
  float *tpool_v_w = nullptr, *tpool_v_b = nullptr;
  float *text_proj_w = nullptr, *text_proj_b = nullptr;
  // caption decoder (RobertaDecoder, roberta.py:329-373): cross-attention layers + vocabulary projection whose rows are
  // zero-padded to a multiple of 256 so that the GEMM needs no column tail
  std::vector<DecLayer> dlayers;
  Lin dec_proj;
  float logit_scale = 0.f;
  // workspace arenas, one per (tower, stream): forwards enqueued on DIFFERENT streams (audio next to text, or two
  // half batches) never share scratch memory and may overlap on the GPU; calls on one stream reuse one arena
  struct WsBuf { char* p = nullptr; size_t bytes = 0; };
  std::map<std::pair<int, void*>, WsBuf> arenas;
};

namespace caco {
namespace {

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
struct Builder {
  caco_model* m;
  std::string err;

  const HostTensor* get(const std::string& key, std::initializer_list<int64_t> shape) {
    auto it = m->pending.find(key);
    if (it == m->pending.end()) {
      if (err.empty()) err = "missing tensor '" + key + "'";
      return nullptr;
    }
    std::vector<int64_t> want(shape);
    if (it->second.shape != want) {
      if (err.empty()) {
        err = "tensor '" + key + "' has shape [";
        for (auto d : it->second.shape) err += std::to_string(d) + ",";
        err += "], expected [";
        for (auto d : want) err += std::to_string(d) + ",";
        err += "]";
      }
      return nullptr;
    }
    return &it->second;
  }

  float* upload_f32(const float* src, size_t n) {
    void* d = nullptr;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) { if (err.empty()) err = "hipMalloc failed"; return nullptr; }
    m->owned.push_back(d);
    if (hipMemcpy(d, src, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess && err.empty()) err = "hipMemcpy failed";
    return reinterpret_cast<float*>(d);
  }
  bf16_t* upload_bf16(const float* src, size_t n) {
    std::vector<uint16_t> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_bf16_host(src[i]);
    void* d = nullptr;
    if (hipMalloc(&d, n * 2) != hipSuccess) { if (err.empty()) err = "hipMalloc failed"; return nullptr; }
    m->owned.push_back(d);
    if (hipMemcpy(d, tmp.data(), n * 2, hipMemcpyHostToDevice) != hipSuccess && err.empty()) err = "hipMemcpy failed";
    return reinterpret_cast<bf16_t*>(d);
  }

  float* vec(const std::string& key, int64_t n) {
    const HostTensor* t = get(key, {n});
    return t ? upload_f32(t->data.data(), (size_t)n) : nullptr;
  }
  float* mat_f32(const std::string& key, int64_t r, int64_t c) {
    const HostTensor* t = get(key, {r, c});
    return t ? upload_f32(t->data.data(), (size_t)(r * c)) : nullptr;
  }
  LNp ln(const std::string& p, int h) {
    LNp o;
    o.g = vec(p + ".weight", h);
    o.b = vec(p + ".bias", h);
    return o;
  }
  // Linear from rows [r0, r0+rows) of a [total, in] weight and its bias
  Lin lin_rows(const std::string& wkey, const std::string& bkey, int total, int in, int r0, int rows) {
    Lin o;
    const HostTensor* w = get(wkey, {total, in});
    const HostTensor* b = get(bkey, {total});
    if (!w || !b) return o;
    o.w = upload_bf16(w->data.data() + (size_t)r0 * in, (size_t)rows * in);
    o.b = upload_f32(b->data.data() + r0, (size_t)rows);
    o.out = rows;
    o.in = in;
    return o;
  }
  Lin lin(const std::string& p, int out, int in) { return lin_rows(p + ".weight", p + ".bias", out, in, 0, out); }
  // Linear [out, in] stored with `out_pad` >= out rows (zero rows / zero bias past `out`)
  Lin lin_padded(const std::string& p, int out, int in, int out_pad) {
    Lin o;
    const HostTensor* w = get(p + ".weight", {out, in});
    const HostTensor* b = get(p + ".bias", {out});
    if (!w || !b) return o;
    std::vector<float> wp((size_t)out_pad * in, 0.f), bp((size_t)out_pad, 0.f);
    std::copy(w->data.begin(), w->data.end(), wp.begin());
    std::copy(b->data.begin(), b->data.end(), bp.begin());
    o.w = upload_bf16(wp.data(), wp.size());
    o.b = upload_f32(bp.data(), bp.size());
    o.out = out_pad;
    o.in = in;
    return o;
  }
  TextLayer text_layer(const std::string& p, int H, int I) {
    TextLayer L;
    L.qkv = lin_cat({p + ".attention.self.query", p + ".attention.self.key", p + ".attention.self.value"}, H, H);
    L.attn_out = lin(p + ".attention.output.dense", H, H);
    L.ln_attn = ln(p + ".attention.output.LayerNorm", H);
    L.inter = lin(p + ".intermediate.dense", I, H);
    L.out = lin(p + ".output.dense", H, I);
    L.ln_out = ln(p + ".output.LayerNorm", H);
    return L;
  }
  // Linear whose rows are the concatenation of several [rows_i, in] Linears (fused projections)
  Lin lin_cat(const std::vector<std::string>& ps, int rows_each, int in) {
    Lin o;
    std::vector<float> w, b;
    for (auto& p : ps) {
      const HostTensor* tw = get(p + ".weight", {rows_each, in});
      const HostTensor* tb = get(p + ".bias", {rows_each});
      if (!tw || !tb) return o;
      w.insert(w.end(), tw->data.begin(), tw->data.end());
      b.insert(b.end(), tb->data.begin(), tb->data.end());
    }
    o.w = upload_bf16(w.data(), w.size());
    o.b = upload_f32(b.data(), b.size());
    o.out = rows_each * (int)ps.size();
    o.in = in;
    return o;
  }

  // rows [r0, r0+rows) of a [total, in] weight with LayerNorm(gamma, beta) folded in
  LinFold lin_fold(const std::string& wkey, const std::string& bkey, const std::string& lnp, int total, int in, int r0, int rows) {
    LinFold o;
    const HostTensor* w = get(wkey, {total, in});
    const HostTensor* b = get(bkey, {total});
    const HostTensor* g = get(lnp + ".weight", {in});
    const HostTensor* be = get(lnp + ".bias", {in});
    if (!w || !b || !g || !be) return o;
    std::vector<uint16_t> wq((size_t)rows * in);
    std::vector<float> c1(rows), c2(rows);
    for (int n = 0; n < rows; ++n) {
      const float* wr = w->data.data() + (size_t)(r0 + n) * in;
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < in; ++k) {
        const uint16_t q = f32_to_bf16_host(g->data[k] * wr[k]);
        wq[(size_t)n * in + k] = q;
        uint32_t u = (uint32_t)q << 16;
        float f;
        __builtin_memcpy(&f, &u, 4);
        s1 += f;                               // row sum of the ROUNDED weights: what the MFMA will see
        s2 += (double)be->data[k] * wr[k];
      }
      c1[n] = (float)s1;
      c2[n] = (float)s2 + b->data[r0 + n];
    }
    void* d = nullptr;
    if (hipMalloc(&d, wq.size() * 2) != hipSuccess) { if (err.empty()) err = "hipMalloc failed"; return o; }
    m->owned.push_back(d);
    if (hipMemcpy(d, wq.data(), wq.size() * 2, hipMemcpyHostToDevice) != hipSuccess && err.empty()) err = "hipMemcpy failed";
    o.lin.w = reinterpret_cast<bf16_t*>(d);
    o.lin.b = upload_f32(c2.data(), c2.size());
    o.c1 = upload_f32(c1.data(), c1.size());
    o.lin.out = rows;
    o.lin.in = in;
    return o;
  }

  // wq[h, k] = scale * sum_d q[h*hd + d] * Wk[h*hd + d, k]  (Wk = rows [k_row0, k_row0 + H) of `wkey`, [wrows, H])
  float* pool_wq(const float* q, const std::string& wkey, int wrows, int k_row0, int H, int heads, float scale) {
    const HostTensor* w = get(wkey, {wrows, H});
    if (!w || !q) return nullptr;
    const int hd = H / heads;
    std::vector<float> out((size_t)heads * H);
    for (int h = 0; h < heads; ++h)
      for (int k = 0; k < H; ++k) {
        double acc = 0.0;
        for (int d = 0; d < hd; ++d) acc += (double)q[h * hd + d] * w->data[(size_t)(k_row0 + h * hd + d) * H + k];
        out[(size_t)h * H + k] = (float)(acc * scale);
      }
    return upload_f32(out.data(), out.size());
  }
  float* rows_f32(const std::string& key, int total, int cols, int r0, int rows) {
    const HostTensor* t = cols > 0 ? get(key, {total, cols}) : get(key, {total});
    return t ? upload_f32(t->data.data() + (size_t)r0 * (cols > 0 ? cols : 1), (size_t)rows * (cols > 0 ? cols : 1)) : nullptr;
  }

  void audio_layers(AudioStack& s, const std::string& prefix, int nlayers, int H, int I) {
    for (int n = 0; n < nlayers; ++n) {
      const std::string p = prefix + ".layers." + std::to_string(n);
      AudioLayer L;
      L.ln1 = ln(p + ".norm1", H);
      // packed in_proj rows [Wq; Wk; Wv] (torch.nn.MultiheadAttention): one GEMM, output row = Q | K | V
      L.qkv = lin_rows(p + ".attn.in_proj_weight", p + ".attn.in_proj_bias", 3 * H, H, 0, 3 * H);
      L.o = lin(p + ".attn.out_proj", H, H);
      L.ln2 = ln(p + ".norm2", H);
      L.fc1 = lin(p + ".mlp.fc1", I, H);
      L.fc2 = lin(p + ".mlp.fc2", H, I);
      if (H % 256 == 0 && I % 256 == 0) {     // LayerNorm-folded twins for the big-batch path (run_audio_layers)
        L.qkv_f = lin_fold(p + ".attn.in_proj_weight", p + ".attn.in_proj_bias", p + ".norm1", 3 * H, H, 0, 3 * H);
        L.fc1_f = lin_fold(p + ".mlp.fc1.weight", p + ".mlp.fc1.bias", p + ".norm2", I, H, 0, I);
      }
      s.layers.push_back(L);
    }
  }
};

int build_weights(caco_model* m) {
  const caco_config& c = m->cfg;
  Builder B{m, ""};
  const bool mae_names = m->pending.count("encoder.input_proj.weight") > 0;
  if (c.has_audio) {
    const std::string ap = mae_names ? "encoder" : "audio_module";
    const int H = c.audio_hidden;
    m->enc.input_proj = B.lin(ap + ".input_proj", H, c.patch_size);
    m->enc.freq_table = B.mat_f32(ap + ".freq_positional_embedding", c.num_freq_patches, H);
    B.audio_layers(m->enc, ap, c.audio_layers, H, c.audio_intermediate);
    m->enc.norm = B.ln(ap + ".norm", H);
    if (!mae_names) {
      // kv_proj rows [0, H) = keys, [H, 2H) = values (caco.py:50-51 chunk(2)); query scaled by 1/sqrt(head_dim) (:55-63)
      const HostTensor* pq = B.get("audio_attention_pool.query", {H});
      const int phd = H / c.pool_heads;
      m->pool_wq = B.pool_wq(pq ? pq->data.data() : nullptr, "audio_attention_pool.kv_proj.weight", 2 * H, 0, H, c.pool_heads,
                             1.0f / sqrtf((float)phd));
      m->pool_v_w = B.rows_f32("audio_attention_pool.kv_proj.weight", 2 * H, H, H, H);
      m->pool_v_b = B.rows_f32("audio_attention_pool.kv_proj.bias", 2 * H, 0, H, H);
      m->pool_out_w = B.mat_f32("audio_attention_pool.out_proj.weight", c.projection_size, H);
      m->pool_out_b = B.vec("audio_attention_pool.out_proj.bias", c.projection_size);
    }
    if (c.mae_decoder_layers > 0) {
      m->dec.input_proj = B.lin("decoder.input_proj", H, H);
      m->dec.freq_table = B.mat_f32("decoder.freq_positional_embedding", c.num_freq_patches, H);
      m->dec.restore_patch = B.vec("decoder.restore_patch", H);
      B.audio_layers(m->dec, "decoder", c.mae_decoder_layers, H, c.audio_intermediate);
      m->dec.norm = B.ln("decoder.norm", H);
      m->dec.output_proj = B.lin("decoder.output_proj", c.patch_size, H);
    }
  }
  if (c.has_text) {
    const int H = c.text_hidden;
    const std::string e = "text_module.embeddings";
    m->word = B.mat_f32(e + ".word_embeddings.weight", c.text_vocab, H);
    m->pos = B.mat_f32(e + ".position_embeddings.weight", c.text_max_pos, H);
    m->type0 = B.mat_f32(e + ".token_type_embeddings.weight", c.text_type_vocab, H);
    m->emb_ln = B.ln(e + ".LayerNorm", H);
    for (int n = 0; n < c.text_layers; ++n) {
      const std::string p = "text_module.encoder.layers." + std::to_string(n);
      m->tlayers.push_back(B.text_layer(p, H, c.text_intermediate));
    }
    // key = key_proj(h) / sqrt(H) (roberta.py:259); one query, one head
    const HostTensor* q = B.get("text_module.pooler.attention_pool_query", {1, H});
    m->tpool_wq = B.pool_wq(q ? q->data.data() : nullptr, "text_module.pooler.key_proj.weight", H, 0, H, 1, 1.0f / sqrtf((float)H));
    (void)B.get("text_module.pooler.key_proj.bias", {H});      // required by the contract; a constant score shift: no effect
    m->tpool_v_w = B.mat_f32("text_module.pooler.value_proj.weight", H, H);
    m->tpool_v_b = B.vec("text_module.pooler.value_proj.bias", H);
    m->text_proj_w = B.mat_f32("text_proj.weight", c.projection_size, H);
    m->text_proj_b = B.vec("text_proj.bias", c.projection_size);
    for (int n = 0; n < c.caption_decoder_layers; ++n) {
      const std::string p = "decoder_module.encoder.layers." + std::to_string(n);
      DecLayer L;
      L.t = B.text_layer(p, H, c.text_intermediate);
      L.cq = B.lin(p + ".crossattention.self.query", H, H);
      L.ckv = B.lin_cat({p + ".crossattention.self.key", p + ".crossattention.self.value"}, H, H);
      L.cattn_out = B.lin(p + ".crossattention.output.dense", H, H);
      L.ln_cross = B.ln(p + ".crossattention.output.LayerNorm", H);
      m->dlayers.push_back(L);
    }
    if (c.caption_decoder_layers > 0)
      m->dec_proj = B.lin_padded("decoder_module.decoder_proj", c.text_vocab, H, (c.text_vocab + 255) / 256 * 256);
  }
  auto ls = m->pending.find("logit_scale");
  if (ls != m->pending.end() && ls->second.data.size() == 1) m->logit_scale = ls->second.data[0];
  if (!B.err.empty()) {
    set_error("caco_finalize_weights: %s", B.err.c_str());
    return CACO_ERR_INVALID;
  }
  return CACO_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
enum { WS_AUDIO = 0, WS_TEXT = 1, WS_FRONTEND = 2 };
struct Arena {
  caco_model::WsBuf* w;
  size_t off = 0;
  Arena(caco_model* mm, int tower, hipStream_t st) : w(&mm->arenas[std::make_pair(tower, (void*)st)]) {}
  size_t reserve(size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
  int commit(hipStream_t st) {
    if (off <= w->bytes) return CACO_OK;
    CACO_HIP(hipStreamSynchronize(st));       // earlier work on this stream may still be using the old arena
    if (w->p) CACO_HIP(hipFree(w->p));
    w->p = nullptr;
    w->bytes = 0;
    CACO_HIP(hipMalloc(reinterpret_cast<void**>(&w->p), off));
    w->bytes = off;
    return CACO_OK;
  }
  template <typename T>
  T* at(size_t o) const { return reinterpret_cast<T*>(w->p + o); }
};

// order (ping-pong traversal, run_audio_layers): 0 = the kernel's default tile order; 1 / 2 = one n-tile group, so that every
// XCD owns one contiguous range of M, walked first to last / last to first
inline void set_order(GemmArgs& g, int order) {
  if (order) {
    g.ngroup = g.N / 256 > 0 ? g.N / 256 : 1;
    g.reverse = order == 2;
  }
}
int linear_bf16(const Lin& L, const bf16_t* a, int64_t M, int act, bf16_t* out, hipStream_t st, int ldc = 0, int order = 0) {
  GemmArgs g{a, L.w, L.b, nullptr, out, M, L.out, L.in, ldc ? ldc : L.out};
  set_order(g, order);
  return gemm_bf16(g, EPI_BF16, act, st);
}
// Row stride (elements) of the fused Q | K | V buffer: rows padded to a multiple of 512 elements (3H = 2304 -> 2560).
// In ISOLATION (the same GEMM launched back to back, tools/gemm_ldc_bench.py) the 4608-byte pitch of 3H = 2304 is
// pathological for the 128 000-row QKV GEMM's stores: 496 us at 2304, 425 at 2432, 401 at 2560, 398 at 3072, while every
// other pitch on the path (1536, 3072, 6144 bytes) is insensitive.  Inside the pipeline the GEMM itself does not care
// (5.04 ms per step either way, A/B on one box); the attention kernel reading the buffer gains 1.5 %.
inline int qkv_ld(int H) { return (3 * H + 511) / 512 * 512; }
int linear_f32(const Lin& L, const bf16_t* a, int64_t M, const float* resid, float* out, hipStream_t st, int order = 0) {
  GemmArgs g{a, L.w, L.b, resid, out, M, L.out, L.in, L.out};
  set_order(g, order);
  return gemm_bf16(g, EPI_F32, ACT_NONE, st);
}
#define CACO_TRY(expr)       \
  do {                       \
    int _rc = (expr);        \
    if (_rc) return _rc;     \
  } while (0)

struct AudioWs {
  size_t x, h, qkv, o, a, part, mr;
  void plan(Arena& A, int64_t M, int batch, int seq, int H, int I) {
    x = A.reserve((size_t)M * H * 4);
    h = A.reserve((size_t)M * H * 2);
    part = A.reserve((size_t)M * (H / 64) * 8);     // LayerNorm folding: per-row partial (sum, sumsq) per 64 columns
    mr = A.reserve((size_t)M * 8);                  // per-row (mean, rstd)
    qkv = A.reserve((size_t)M * qkv_ld(H) * 2);
    o = A.reserve((size_t)M * H * 2);
    a = A.reserve((size_t)M * I * 2);
  }
};

// 0 = separate LayerNorm passes (default), 1 = folded, -1 = folded when the GEMMs fill the chip.  Measured at batch 256
// (same box, interleaved): folding removes 2.2 ms of LayerNorm passes per step but the richer GEMM epilogues (bf16
// copy + row statistics on out-proj / fc2, per-row correction on QKV / fc1) cost 2.7 ms, because a 256x256 tile's
// epilogue runs with the matrix pipe idle.  Kept as an option (and tested) until the epilogue overlaps the K-loop.
static int g_ln_fold = getenv("CACO_LN_FOLD") ? atoi(getenv("CACO_LN_FOLD")) : 0;      // default of NEW models

int linear_fold(const LinFold& L, const bf16_t* xb, const float* mr, int64_t M, int act, bf16_t* out, hipStream_t st, int ldc = 0) {
  GemmArgs g{xb, L.lin.w, L.lin.b, nullptr, out, M, L.lin.out, L.lin.in, ldc ? ldc : L.lin.out};
  g.fold_mr = mr;
  g.fold_c1 = L.c1;
  return gemm_bf16(g, EPI_BF16, act, st);
}
// out-proj / fc2 with residual: x (fp32, in place) + bf16 copy of the new rows + their partial row statistics
int linear_resid_stats(const Lin& L, const bf16_t* a, int64_t M, float* x, bf16_t* xb, float* part, hipStream_t st) {
  GemmArgs g{a, L.w, L.b, x, x, M, L.out, L.in, L.out};
  g.xb_out = xb;
  g.stats_part = part;
  return gemm_bf16(g, EPI_F32, ACT_NONE, st);
}

// 12 x AudioEncoderLayer.forward (mae.py:80-99) on the fp32 residual stream x[M, H].
// Optional LayerNorm-FOLDED form (caco_set_ln_fold): no LayerNorm pass at all inside the stack.  The GEMM that follows a
// LayerNorm consumes the raw rows (bf16 copy xb) with gamma-scaled weights and applies mean / rstd per row in its
// epilogue; the GEMM that produces new residual rows (out-proj, fc2) also writes xb and per-row partial sums, which a
// tiny kernel turns into (mean, rstd).  Saves 590 MB of HBM traffic and a launch per LayerNorm.
// CACO_PINGPONG=1 (switch SW_PINGPONG; round 3, untimed): consecutive kernels of a layer walk the rows in OPPOSITE directions
// inside the 8 row ranges the XCDs own (LayerNorm: layernorm_ranges_kernel; attention: contiguous clips per XCD; GEMMs: one
// n-tile group, reversed tile list).  Every kernel then starts on the rows its producer wrote last - the part of the
// producer's output that is still in the 256 MiB Infinity Cache - instead of on the rows that were written first and have
// been evicted by the rest of the same output (fc1's output alone is 780 MB).  `kdir` counts kernels; its parity is the direction.
bool pingpong_enabled() { return sw(SW_PINGPONG) != 0; }

int run_audio_layers(caco_model* m, const std::vector<AudioLayer>& layers, const Arena& A, const AudioWs& w,
                     const float* mask, int batch, int seq, int heads, float eps, hipStream_t st, int* kdir = nullptr) {
  const int H = m->cfg.audio_hidden;
  const int64_t M = (int64_t)batch * seq;
  float* x = A.at<float>(w.x);
  bf16_t* h = A.at<bf16_t>(w.h);
  bf16_t* qkv = A.at<bf16_t>(w.qkv);
  bf16_t* o = A.at<bf16_t>(w.o);
  bf16_t* a = A.at<bf16_t>(w.a);
  float* part = A.at<float>(w.part);
  float* mr = A.at<float>(w.mr);
  const bool can_fold = !layers.empty() && layers[0].qkv_f.c1 != nullptr && H <= 1024;
  const bool fold = can_fold && (m->ln_fold == 1 || (m->ln_fold < 0 && ((M + 255) / 256) * (H / 128) >= 4 * 256));
  if (fold) {
    bf16_t* xb = h;                 // the bf16 operand buffer holds the RAW rows in this form
    CACO_STAGE("audio.ln", row_stats_bf16(x, M, H, eps, xb, mr, st));       // once per stack, at its entry
    for (const AudioLayer& L : layers) {
      CACO_STAGE("audio.gemm_qkv", linear_fold(L.qkv_f, xb, mr, M, ACT_NONE, qkv, st, qkv_ld(H)));
      CACO_STAGE("audio.attention", attention(qkv, qkv_ld(H), H, 2 * H, mask, batch, seq, heads, H / heads, 0, o, st));
      CACO_STAGE("audio.gemm_out", linear_resid_stats(L.o, o, M, x, xb, part, st));
      CACO_STAGE("audio.ln_stats", ln_stats_finalize(part, H / 64, M, H, eps, mr, st));
      CACO_STAGE("audio.gemm_fc1", linear_fold(L.fc1_f, xb, mr, M, ACT_SILU, a, st));
      CACO_STAGE("audio.gemm_fc2", linear_resid_stats(L.fc2, a, M, x, xb, part, st));
      CACO_STAGE("audio.ln_stats", ln_stats_finalize(part, H / 64, M, H, eps, mr, st));
    }
    return CACO_OK;
  }
  int kd_local = 0;
  int& kd = kdir ? *kdir : kd_local;
  const bool pp = kdir != nullptr && pingpong_enabled();
  auto ord = [&]() { const int o = pp ? 1 + (kd & 1) : 0; ++kd; return o; };
  for (const AudioLayer& L : layers) {
    CACO_STAGE("audio.ln", layernorm(x, L.ln1.g, L.ln1.b, M, H, eps, nullptr, h, st, ord()));
    CACO_STAGE("audio.gemm_qkv", linear_bf16(L.qkv, h, M, ACT_NONE, qkv, st, qkv_ld(H), ord()));
    CACO_STAGE("audio.attention", attention(qkv, qkv_ld(H), H, 2 * H, mask, batch, seq, heads, H / heads, 0, o, st, ord()));
    CACO_STAGE("audio.gemm_out", linear_f32(L.o, o, M, x, x, st, ord()));
    CACO_STAGE("audio.ln", layernorm(x, L.ln2.g, L.ln2.b, M, H, eps, nullptr, h, st, ord()));
    CACO_STAGE("audio.gemm_fc1", linear_bf16(L.fc1, h, M, ACT_SILU, a, st, 0, ord()));
    CACO_STAGE("audio.gemm_fc2", linear_f32(L.fc2, a, M, x, x, st, ord()));
  }
  return CACO_OK;
}

// CACO_POS_FUSE=1 (switch SW_POS_FUSE; opt-in until timed on hardware): positional embedding inside the patch-embed GEMM
bool pos_fuse_enabled() { return sw(SW_POS_FUSE) != 0; }

// CACO_POOL_FUSE=1 (switch SW_POOL_FUSE; opt-in until timed): the audio tower's final LayerNorm inside the pooling kernel
bool pool_fuse_enabled() { return sw(SW_POOL_FUSE) != 0; }

// The model's weights, arenas and per-device kernel state live on ONE device: a call with another device current would
// dereference foreign memory.  Reject it instead.
int check_device(const caco_model* m) {
  int dev = -1;
  CACO_HIP(hipGetDevice(&dev));
  if (dev != m->device) {
    set_error("model lives on device %d but device %d is current (hipSetDevice / torch.cuda.device first)", m->device, dev);
    return CACO_ERR_STATE;
  }
  return CACO_OK;
}

int check_audio_shapes(const caco_model* m, int batch, int seq) {
  CACO_REQUIRE(m && m->finalized, "model is null or weights not finalized");
  CACO_TRY(check_device(m));
  CACO_REQUIRE(m->cfg.has_audio, "model was created without the audio tower");
  CACO_REQUIRE(batch > 0 && seq > 0, "bad audio shape B=%d S=%d", batch, seq);
  const int hd = m->cfg.audio_hidden / m->cfg.audio_heads;
  CACO_REQUIRE(hd == 96 || hd == 64, "audio head_dim %d unsupported", hd);
  return CACO_OK;
}

// patches (f32 | bf16) -> bf16 [M, P] operand, casting into scratch if needed
int patches_as_bf16(const void* patches, int dtype, int64_t n, bf16_t* scratch, const bf16_t** out, hipStream_t st) {
  if (dtype == CACO_DTYPE_BF16) {
    *out = reinterpret_cast<const bf16_t*>(patches);
    return CACO_OK;
  }
  CACO_REQUIRE(dtype == CACO_DTYPE_F32, "patch dtype %d unknown", dtype);
  CACO_TRY(cast_f32_to_bf16(reinterpret_cast<const float*>(patches), scratch, n, st));
  *out = scratch;
  return CACO_OK;
}

}  // namespace
}  // namespace caco

// ================================================================================================
// extern "C" surface
// ================================================================================================
extern "C" {

const char* caco_version(void) { return "cacophony_amd 0.2 (gfx950)"; }
int32_t caco_config_size(void) { return (int32_t)sizeof(caco_config); }
const char* caco_last_error(void) { return caco::g_err; }

void caco_default_config(caco_config* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->audio_hidden = 768; c->audio_layers = 12; c->audio_heads = 8; c->audio_intermediate = 3072;
  c->patch_size = 256; c->num_freq_patches = 8; c->audio_ln_eps = 1e-5f;
  c->text_vocab = 50265; c->text_hidden = 768; c->text_layers = 12; c->text_heads = 12; c->text_intermediate = 3072;
  c->text_max_pos = 514; c->text_type_vocab = 1; c->text_ln_eps = 1e-5f;
  c->projection_size = 768; c->pool_heads = 2; c->logit_scale = 2.6592f;
  c->has_audio = 1; c->has_text = 1; c->mae_decoder_layers = 0; c->caption_decoder_layers = 0;
}

int caco_create(const caco_config* cfg, caco_model** out) {
  CACO_REQUIRE(cfg && out, "caco_create: null argument");
  CACO_REQUIRE(cfg->has_audio || cfg->has_text, "caco_create: model needs at least one tower");
  if (cfg->has_audio) {
    CACO_REQUIRE(cfg->audio_hidden % 128 == 0 && cfg->audio_intermediate % 128 == 0 && cfg->patch_size % 128 == 0,
                 "caco_create: audio hidden/intermediate/patch sizes must be multiples of 128");
    CACO_REQUIRE(cfg->audio_heads > 0 && cfg->audio_hidden % cfg->audio_heads == 0, "caco_create: bad audio head count");
    CACO_REQUIRE(cfg->pool_heads > 0 && cfg->audio_hidden % cfg->pool_heads == 0 && (cfg->pool_heads == 1 || cfg->pool_heads % 2 == 0),
                 "caco_create: the audio pooler takes 1 or an even number of heads dividing the hidden size (got %d)", cfg->pool_heads);
    CACO_REQUIRE(cfg->projection_size % 8 == 0, "caco_create: projection_size must be a multiple of 8");
  }
  if (cfg->has_text) {
    CACO_REQUIRE(cfg->text_hidden % 128 == 0 && cfg->text_intermediate % 128 == 0, "caco_create: text sizes must be multiples of 128");
    CACO_REQUIRE(cfg->text_heads > 0 && cfg->text_hidden / cfg->text_heads == 64 && cfg->text_hidden % cfg->text_heads == 0,
                 "caco_create: text head_dim must be 64");
  }
  CACO_REQUIRE(cfg->caption_decoder_layers >= 0 && (cfg->caption_decoder_layers == 0 || (cfg->has_text && cfg->has_audio &&
               cfg->audio_hidden == cfg->text_hidden)),
               "caco_create: the caption decoder needs both towers with equal hidden sizes");
  int dev = 0;
  CACO_HIP(hipGetDevice(&dev));
  caco_model* m = new (std::nothrow) caco_model();
  CACO_REQUIRE(m, "caco_create: out of host memory");
  m->cfg = *cfg;
  m->device = dev;
  m->ln_fold = g_ln_fold;
  m->logit_scale = cfg->logit_scale;
  *out = m;
  return CACO_OK;
}

void caco_destroy(caco_model* m) {
  if (!m) return;
  for (void* p : m->owned) (void)hipFree(p);
  for (auto& kv : m->arenas) if (kv.second.p) (void)hipFree(kv.second.p);
  delete m;
}

int caco_load_tensor(caco_model* m, const char* name, const float* host, const int64_t* shape, int32_t ndim) {
  CACO_REQUIRE(m && name && host && (shape || ndim == 0) && ndim >= 0 && ndim <= 4, "caco_load_tensor: bad arguments");
  if (m->finalized) {
    set_error("caco_load_tensor: weights already finalized");
    return CACO_ERR_STATE;
  }
  const std::string key(name);
  if (key.rfind("decoder_module.", 0) == 0 && m->cfg.caption_decoder_layers == 0) return CACO_OK;   // model built without it
  static const char* known[] = {"audio_module.", "audio_attention_pool.", "text_module.", "text_proj.", "logit_scale",
                                "encoder.", "decoder.", "decoder_module."};
  bool ok = false;
  for (const char* k : known) ok = ok || key.rfind(k, 0) == 0;
  CACO_REQUIRE(ok, "caco_load_tensor: unknown state-dict key '%s'", name);
  HostTensor t;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    CACO_REQUIRE(shape[i] > 0, "caco_load_tensor: non-positive dimension in '%s'", name);
    t.shape.push_back(shape[i]);
    n *= shape[i];
  }
  t.data.assign(host, host + n);
  m->pending[key] = std::move(t);
  return CACO_OK;
}

int caco_finalize_weights(caco_model* m) {
  CACO_REQUIRE(m, "caco_finalize_weights: null model");
  if (m->finalized) return CACO_OK;
  int rc = build_weights(m);
  if (rc) return rc;
  CACO_HIP(hipDeviceSynchronize());
  m->pending.clear();
  m->finalized = true;
  return CACO_OK;
}

int caco_set_logit_scale(caco_model* m, float v) {
  CACO_REQUIRE(m, "null model");
  m->logit_scale = v;
  return CACO_OK;
}
float caco_get_logit_scale(const caco_model* m) { return m ? m->logit_scale : 0.f; }
int64_t caco_workspace_bytes(const caco_model* m) {
  int64_t n = 0;
  if (m) for (auto& kv : m->arenas) n += (int64_t)kv.second.bytes;
  return n;
}
int32_t caco_set_gemm_tile(int32_t tile) { return set_gemm_tile_config(tile); }
int caco_set_switch(const char* name, int32_t value) {
  CACO_REQUIRE(name, "caco_set_switch: null name");
  for (int i = 0; i < SW_COUNT; ++i)
    if (!strcmp(name, g_switches[i].name)) {
      CACO_REQUIRE(switch_value_ok((Switch)i, value), "caco_set_switch: value %d out of range for %s", (int)value, name);
      sw((Switch)i);                               // settle the environment's initial value first, so that it cannot overwrite this one
      g_switches[i].value.store(value, std::memory_order_relaxed);
      return CACO_OK;
    }
  set_error("caco_set_switch: unknown switch '%s'", name);
  return CACO_ERR_INVALID;
}
int32_t caco_get_switch(const char* name) {
  if (name)
    for (int i = 0; i < SW_COUNT; ++i)
      if (!strcmp(name, g_switches[i].name)) return sw((Switch)i);
  return INT32_MIN;
}
int32_t caco_set_ln_fold(int32_t mode) {
  if (mode >= -1 && mode <= 1) g_ln_fold = mode;
  return g_ln_fold;
}
int32_t caco_model_set_ln_fold(caco_model* m, int32_t mode) {
  if (!m) return 0;
  if (mode >= -1 && mode <= 1) m->ln_fold = mode;
  return m->ln_fold;
}

int caco_profile_enable(int32_t on) {
  g_prof_on = on != 0;
  return CACO_OK;
}

int64_t caco_profile_report(char* buf, int64_t buflen) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  std::map<std::string, std::pair<double, long>> acc;
  for (ProfRec& r : g_prof_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& e = acc[r.name];
      e.first += ms;
      e.second += 1;
    }
    g_prof_pool.push_back(r.a);
    g_prof_pool.push_back(r.b);
  }
  g_prof_recs.clear();
  std::string js = "{";
  bool first = true;
  for (auto& kv : acc) {
    char tmp[256];
    snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"ms\": %.6f, \"n\": %ld}", first ? "" : ", ", kv.first.c_str(), kv.second.first,
             kv.second.second);
    js += tmp;
    first = false;
  }
  js += "}";
  if (buf && buflen > 0) {
    const size_t n = js.size() < (size_t)buflen - 1 ? js.size() : (size_t)buflen - 1;
    memcpy(buf, js.data(), n);
    buf[n] = 0;
  }
  return (int64_t)js.size() + 1;
}

int64_t caco_mel_num_frames(int64_t n_samples) { return (n_samples + 159) / 160; }

int caco_mel_spectrogram(const float* wav, int32_t batch, int64_t n_samples, float scale, float bias, float* mel, void* stream) {
  return mel_frontend(wav, batch, n_samples, 0, scale, bias, mel, MEL_NATURAL_F32, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int caco_mel_patches_lens(const float* wav, const int64_t* lengths, int32_t batch, int64_t n_samples, int32_t max_patches,
                          float scale, float bias, void* patches, int32_t dtype, float* tinds, float* finds, float* mask,
                          void* stream) {
  CACO_REQUIRE(dtype == CACO_DTYPE_F32 || dtype == CACO_DTYPE_BF16, "caco_mel_patches: patch dtype %d unknown", dtype);
  return mel_frontend(wav, batch, n_samples, max_patches, scale, bias, patches,
                      dtype == CACO_DTYPE_BF16 ? MEL_PATCH_BF16 : MEL_PATCH_F32, tinds, finds, mask, (hipStream_t)stream, lengths);
}
int caco_mel_patches(const float* wav, int32_t batch, int64_t n_samples, int32_t max_patches, float scale, float bias,
                     void* patches, int32_t dtype, float* tinds, float* finds, float* mask, void* stream) {
  return caco_mel_patches_lens(wav, nullptr, batch, n_samples, max_patches, scale, bias, patches, dtype, tinds, finds, mask, stream);
}

}  // extern "C"

// emb rows are written with stride ld_emb (0 = projection_size): both towers can fill one packed [B, 2, P] send buffer
static int audio_forward_impl(caco_model* m, const void* patches, int32_t dtype, const float* tinds, const float* finds,
                              const float* mask, int32_t batch, int32_t seq, int32_t normalize, float* emb, int32_t ld_emb,
                              float* hidden, void* stream) {
  CACO_TRY(check_audio_shapes(m, batch, seq));
  CACO_REQUIRE(ld_emb == 0 || (normalize && ld_emb >= m->cfg.projection_size), "audio forward: an embedding row stride needs normalize = 1 and ld >= projection_size");
  CACO_REQUIRE(patches && tinds && finds && mask && emb, "caco_audio_forward: null argument");
  CACO_REQUIRE(m->pool_wq, "caco_audio_forward: model has no audio pooler (AudioMAE-only weights)");
  hipStream_t st = (hipStream_t)stream;
  const caco_config& c = m->cfg;
  const int H = c.audio_hidden, P = c.patch_size;
  const int64_t M = (int64_t)batch * seq;
  Arena A(m, WS_AUDIO, st);
  AudioWs w;
  w.plan(A, M, batch, seq, H, c.audio_intermediate);
  const size_t o_pb = A.reserve((size_t)M * P * 2);
  const size_t o_pool = A.reserve((size_t)batch * c.pool_heads * H * 4);
  const size_t o_pv = A.reserve((size_t)batch * H * 4);
  const size_t o_emb = A.reserve((size_t)batch * c.projection_size * 4);
  const int nf = c.num_freq_patches, pos_tmax = (seq + nf - 1) / nf + 1;        // time patches a front end can produce for `seq`
  const size_t o_ptab = A.reserve((size_t)pos_tmax * nf * H * 4);
  const size_t o_pidx = A.reserve((size_t)M * 4);
  CACO_TRY(A.commit(st));
  float* x = A.at<float>(w.x);
  bf16_t* h = A.at<bf16_t>(w.h);
  const bf16_t* pb = nullptr;
  CACO_STAGE("audio.patch_cast", patches_as_bf16(patches, dtype, M * P, A.at<bf16_t>(o_pb), &pb, st));
  // AudioEncoder.forward, mae.py:125-148: x = input_proj(patches) + sincos(time) + freq_table[freq]
  GemmArgs gpe{pb, m->enc.input_proj.w, m->enc.input_proj.b, nullptr, x, M, m->enc.input_proj.out, m->enc.input_proj.in, m->enc.input_proj.out};
  if (pos_fuse_enabled() && gemm_bf16_picks_w8(gpe, EPI_F32)) {
    // the positional embedding rides in the patch-embed GEMM's epilogue as a gathered residual (norm.hip pos_prepare): no
    // separate 780 MB read-modify-write pass over x.  Rows whose time index is not a small integer are finished exactly by
    // add_pos_embed_rest (none with the indices any front end produces: that launch reads M ints and exits).
    float* ptab = A.at<float>(o_ptab);
    int* pidx = A.at<int>(o_pidx);
    CACO_STAGE("audio.pos_embed", pos_prepare(m->enc.freq_table, pos_tmax, nf, H, ptab, tinds, finds, M, pidx, st));
    gpe.resid = ptab;
    gpe.resid_idx = pidx;
    CACO_STAGE("audio.patch_embed", gemm_bf16(gpe, EPI_F32, ACT_NONE, st));
    CACO_STAGE("audio.pos_embed", add_pos_embed_rest(x, tinds, finds, m->enc.freq_table, pidx, M, H, nf, st));
  } else {
    CACO_STAGE("audio.patch_embed", gemm_bf16(gpe, EPI_F32, ACT_NONE, st));
    CACO_STAGE("audio.pos_embed", add_pos_embed(x, nullptr, tinds, finds, m->enc.freq_table, M, H, nf, st));
  }
  int kdir = 1;                                   // the patch-embed GEMM was kernel 0 (first to last)
  CACO_TRY(run_audio_layers(m, m->enc.layers, A, w, mask, batch, seq, c.audio_heads, c.audio_ln_eps, st, &kdir));
  // AudioAttentionPooler.forward, caco.py:41-79 (projections folded out of the token loop, pool.hip)
  float* pooled = A.at<float>(o_pool);                  // [B, heads, H]: softmax-weighted token means per head
  float* pv = A.at<float>(o_pv);                        // [B, H]: value projection of the pooled rows, heads concatenated
  const int phd = H / c.pool_heads;
  if (!hidden && pool_fuse_enabled()) {
    // nobody reads the normalised rows but the pooler: it normalises them itself, on the way in (opt-in, round 3)
    CACO_STAGE("audio.pool", attn_pool_rows_ln(x, m->enc.norm.g, m->enc.norm.b, c.audio_ln_eps, m->pool_wq, mask, batch, seq, H,
                                               c.pool_heads, pooled, st));
  } else {
    // the fp32 hidden states are only written when the caller asks for them (encode_audio does not: 390 MB per batch of 256)
    CACO_STAGE("audio.ln", layernorm(x, m->enc.norm.g, m->enc.norm.b, M, H, c.audio_ln_eps, hidden, h, st));
    CACO_STAGE("audio.pool", attn_pool_rows(h, m->pool_wq, mask, batch, seq, H, c.pool_heads, pooled, st));
  }
  for (int hh = 0; hh < c.pool_heads; ++hh)
    CACO_STAGE("audio.pool", gemm_f32(pooled + (size_t)hh * H, m->pool_v_w + (size_t)hh * phd * H, m->pool_v_b + hh * phd,
                                      pv + hh * phd, batch, phd, H, H, 1.0f, st, c.pool_heads * H));
  float* e = normalize ? A.at<float>(o_emb) : emb;
  CACO_STAGE("audio.proj_norm", gemm_f32(pv, m->pool_out_w, m->pool_out_b, e, batch, c.projection_size, H, c.projection_size, 1.0f, st));
  if (normalize) CACO_STAGE("audio.proj_norm", l2_normalize(e, batch, c.projection_size, emb, st, ld_emb));
  return CACO_OK;
}

static int text_forward_impl(caco_model* m, const int64_t* ids, const int64_t* mask, const int64_t* pos_ids, int32_t batch,
                             int32_t seq, int32_t normalize, float* emb, int32_t ld_emb, float* hidden, void* stream) {
  CACO_REQUIRE(m && m->finalized, "model is null or weights not finalized");
  CACO_TRY(check_device(m));
  CACO_REQUIRE(ld_emb == 0 || (normalize && ld_emb >= m->cfg.projection_size), "text forward: an embedding row stride needs normalize = 1 and ld >= projection_size");
  CACO_REQUIRE(m->cfg.has_text, "model was created without the text tower");
  CACO_REQUIRE(ids && mask && emb && batch > 0 && seq > 0, "caco_text_forward: bad arguments");
  const caco_config& c = m->cfg;
  CACO_REQUIRE(pos_ids || seq <= c.text_max_pos, "caco_text_forward: T=%d exceeds max_position_embeddings=%d", seq, c.text_max_pos);
  hipStream_t st = (hipStream_t)stream;
  const int H = c.text_hidden, I = c.text_intermediate;
  const int64_t M = (int64_t)batch * seq;
  Arena A(m, WS_TEXT, st);
  const size_t o_x = A.reserve((size_t)M * H * 4), o_y = A.reserve((size_t)M * H * 4), o_xb = A.reserve((size_t)M * H * 2);
  const size_t o_qkv = A.reserve((size_t)M * qkv_ld(H) * 2);
  const size_t o_o = A.reserve((size_t)M * H * 2), o_a = A.reserve((size_t)M * I * 2);
  const size_t o_mask = A.reserve((size_t)M * 4), o_pv = A.reserve((size_t)batch * H * 4);
  const size_t o_pool = A.reserve((size_t)batch * H * 4), o_emb = A.reserve((size_t)batch * c.projection_size * 4);
  CACO_TRY(A.commit(st));
  float* x = A.at<float>(o_x);
  float* y = A.at<float>(o_y);
  bf16_t* xb = A.at<bf16_t>(o_xb);
  bf16_t* qkv = A.at<bf16_t>(o_qkv);
  bf16_t* o = A.at<bf16_t>(o_o);
  bf16_t* a = A.at<bf16_t>(o_a);
  float* fmask = A.at<float>(o_mask);
  CACO_TRY(mask_i64_to_f32(mask, fmask, M, st));
  // RobertaEmbeddings.forward, roberta.py:35-53
  CACO_STAGE("text.embed_ln", text_embed_ln(ids, pos_ids, m->word, m->pos, m->type0, m->emb_ln.g, m->emb_ln.b, M, seq, H,
                                            c.text_vocab, c.text_max_pos, c.text_ln_eps, x, xb, st));
  // RobertaEncoder: 12 x RobertaLayer.forward (post-LN), roberta.py:191-215
  const int nl = (int)m->tlayers.size();
  for (int n = 0; n < nl; ++n) {
    const TextLayer& L = m->tlayers[n];
    CACO_STAGE("text.gemm_qkv", linear_bf16(L.qkv, xb, M, ACT_NONE, qkv, st, qkv_ld(H)));
    CACO_STAGE("text.attention", attention(qkv, qkv_ld(H), H, 2 * H, fmask, batch, seq, c.text_heads, H / c.text_heads, 1, o, st));
    CACO_STAGE("text.gemm_out", linear_f32(L.attn_out, o, M, x, y, st));
    CACO_STAGE("text.ln", layernorm(y, L.ln_attn.g, L.ln_attn.b, M, H, c.text_ln_eps, x, xb, st));
    CACO_STAGE("text.gemm_fc1", linear_bf16(L.inter, xb, M, ACT_GELU, a, st));
    CACO_STAGE("text.gemm_fc2", linear_f32(L.out, a, M, x, y, st));
    float* xo = (n == nl - 1 && hidden) ? hidden : x;
    CACO_STAGE("text.ln", layernorm(y, L.ln_out.g, L.ln_out.b, M, H, c.text_ln_eps, xo, xb, st));
  }
  if (nl == 0 && hidden) CACO_HIP(hipMemcpyAsync(hidden, x, (size_t)M * H * 4, hipMemcpyDeviceToDevice, st));
  // AttentionPooler.forward (roberta.py:253-271) + text_proj (caco.py:169)
  float* pooled = A.at<float>(o_pool);
  float* pv = A.at<float>(o_pv);
  CACO_STAGE("text.pool", attn_pool_rows(xb, m->tpool_wq, fmask, batch, seq, H, 1, pooled, st));
  CACO_STAGE("text.pool", gemm_f32(pooled, m->tpool_v_w, m->tpool_v_b, pv, batch, H, H, H, 1.0f, st));
  float* e = normalize ? A.at<float>(o_emb) : emb;
  CACO_STAGE("text.proj_norm", gemm_f32(pv, m->text_proj_w, m->text_proj_b, e, batch, c.projection_size, H, c.projection_size, 1.0f, st));
  if (normalize) CACO_STAGE("text.proj_norm", l2_normalize(e, batch, c.projection_size, emb, st, ld_emb));
  return CACO_OK;
}

extern "C" {

int caco_audio_forward(caco_model* m, const void* patches, int32_t dtype, const float* tinds, const float* finds,
                       const float* mask, int32_t batch, int32_t seq, int32_t normalize, float* emb, float* hidden,
                       void* stream) {
  return audio_forward_impl(m, patches, dtype, tinds, finds, mask, batch, seq, normalize, emb, 0, hidden, stream);
}
int caco_text_forward(caco_model* m, const int64_t* ids, const int64_t* mask, const int64_t* pos_ids, int32_t batch,
                      int32_t seq, int32_t normalize, float* emb, float* hidden, void* stream) {
  return text_forward_impl(m, ids, mask, pos_ids, batch, seq, normalize, emb, 0, hidden, stream);
}
int caco_encode_text(caco_model* m, const int64_t* ids, const int64_t* mask, int32_t batch, int32_t seq, float* emb,
                     int32_t ld_emb, void* stream) {
  return text_forward_impl(m, ids, mask, nullptr, batch, seq, 1, emb, ld_emb, nullptr, stream);
}

// RobertaDecoder.forward (roberta.py:337-373) as called by CACO.get_decoder_logits (caco.py:212-240): teacher-forced
// logits over the vocabulary for every caption position, cross-attending to the audio encoder's hidden states.
int caco_decoder_forward(caco_model* m, const float* text_hidden, const int64_t* text_mask, const float* audio_hidden,
                         const float* audio_mask, int32_t batch, int32_t seq_t, int32_t seq_a, float* logits, void* stream) {
  CACO_REQUIRE(m && m->finalized, "model is null or weights not finalized");
  CACO_TRY(check_device(m));
  CACO_REQUIRE(!m->dlayers.empty() && m->dec_proj.w, "Decoder module not initialized");
  CACO_REQUIRE(text_hidden && text_mask && audio_hidden && audio_mask && logits && batch > 0 && seq_t > 0 && seq_a > 0,
               "caco_decoder_forward: bad arguments");
  const caco_config& c = m->cfg;
  hipStream_t st = (hipStream_t)stream;
  const int H = c.text_hidden, I = c.text_intermediate, V = c.text_vocab, Vp = m->dec_proj.out;
  const int heads = c.text_heads, hd = H / heads;
  const int64_t M = (int64_t)batch * seq_t, Ma = (int64_t)batch * seq_a;
  Arena A(m, WS_TEXT, st);
  const size_t o_x = A.reserve((size_t)M * H * 4), o_y = A.reserve((size_t)M * H * 4), o_xb = A.reserve((size_t)M * H * 2);
  const size_t o_qkv = A.reserve((size_t)M * qkv_ld(H) * 2), o_o = A.reserve((size_t)M * H * 2), o_a = A.reserve((size_t)M * I * 2);
  const size_t o_mask = A.reserve((size_t)M * 4), o_ab = A.reserve((size_t)Ma * H * 2), o_kv = A.reserve((size_t)Ma * 2 * H * 2);
  const size_t o_lg = A.reserve((size_t)M * Vp * 4);
  CACO_TRY(A.commit(st));
  float* x = A.at<float>(o_x);
  float* y = A.at<float>(o_y);
  bf16_t* xb = A.at<bf16_t>(o_xb);
  bf16_t* qkv = A.at<bf16_t>(o_qkv);
  bf16_t* o = A.at<bf16_t>(o_o);
  bf16_t* a = A.at<bf16_t>(o_a);
  float* fmask = A.at<float>(o_mask);
  bf16_t* ab = A.at<bf16_t>(o_ab);
  bf16_t* kv = A.at<bf16_t>(o_kv);
  float* lg = A.at<float>(o_lg);
  CACO_TRY(mask_i64_to_f32(text_mask, fmask, M, st));
  CACO_HIP(hipMemcpyAsync(x, text_hidden, (size_t)M * H * 4, hipMemcpyDeviceToDevice, st));
  CACO_STAGE("decoder.cast", cast_f32_to_bf16(x, xb, M * H, st));
  CACO_STAGE("decoder.cast", cast_f32_to_bf16(audio_hidden, ab, Ma * H, st));
  for (const DecLayer& L : m->dlayers) {
    // self-attention: causal AND caption-padding mask (roberta.py:347-356)
    CACO_STAGE("decoder.gemm_qkv", linear_bf16(L.t.qkv, xb, M, ACT_NONE, qkv, st, qkv_ld(H)));
    CACO_STAGE("decoder.attention", attention(qkv, qkv_ld(H), H, 2 * H, fmask, batch, seq_t, heads, hd, 1, o, st));
    CACO_STAGE("decoder.gemm_out", linear_f32(L.t.attn_out, o, M, x, y, st));
    CACO_STAGE("decoder.ln", layernorm(y, L.t.ln_attn.g, L.t.ln_attn.b, M, H, c.text_ln_eps, x, xb, st));
    // cross-attention over the audio tokens: queries from the caption, keys / values from the audio hidden states,
    // padded audio tokens masked (roberta.py:204-210, :358-362)
    CACO_STAGE("decoder.gemm_cross_q", linear_bf16(L.cq, xb, M, ACT_NONE, qkv, st));
    CACO_STAGE("decoder.gemm_cross_kv", linear_bf16(L.ckv, ab, Ma, ACT_NONE, kv, st));
    CACO_STAGE("decoder.cross_attention", attention_qkv(qkv, H, seq_t, kv, 2 * H, 0, H, audio_mask, batch, seq_a, heads, hd, 0, o, st));
    CACO_STAGE("decoder.gemm_out", linear_f32(L.cattn_out, o, M, x, y, st));
    CACO_STAGE("decoder.ln", layernorm(y, L.ln_cross.g, L.ln_cross.b, M, H, c.text_ln_eps, x, xb, st));
    CACO_STAGE("decoder.gemm_fc1", linear_bf16(L.t.inter, xb, M, ACT_GELU, a, st));
    CACO_STAGE("decoder.gemm_fc2", linear_f32(L.t.out, a, M, x, y, st));
    CACO_STAGE("decoder.ln", layernorm(y, L.t.ln_out.g, L.t.ln_out.b, M, H, c.text_ln_eps, x, xb, st));
  }
  // decoder_proj (roberta.py:371): [M, H] x [Vp, H]^T into the padded scratch rows, then the V real columns out
  CACO_STAGE("decoder.gemm_vocab", linear_f32(m->dec_proj, xb, M, nullptr, lg, st));
  CACO_HIP(hipMemcpy2DAsync(logits, (size_t)V * 4, lg, (size_t)Vp * 4, (size_t)V * 4, (size_t)M, hipMemcpyDeviceToDevice, st));
  return CACO_OK;
}

// ------------------------------------------------------------------------------------------------
// Incremental caption decoding with key / value caches: the JAX path's get_next_decoder_logits loop
// (src/caco/caco.py:154-230).  Both stacks are causal, so the keys and values of earlier caption positions never
// change: a step embeds ONE new token per clip, runs it through the 12 text layers and the decoder layers against the
// cached rows, appends its own K | V rows, and projects the last hidden state to the vocabulary.  The cross-attention
// keys / values of the audio tokens are computed once per clip batch.  Same kernels as the full-prefix form
// (caco_text_forward + caco_decoder_forward), which it reproduces position by position.
// ------------------------------------------------------------------------------------------------
struct caco_decode_state {
  caco_model* m = nullptr;
  int batch = 0, seq_a = 0, max_len = 0, pos = 0;
  std::vector<bf16_t*> text_kv, dec_kv, cross_kv;     // [layers] x [B, max_len, 2H] / [B*S, 2H]
  float* audio_mask = nullptr;                        // [B, S] copy
  char* ws = nullptr;                                 // per-step scratch
  size_t ws_bytes = 0;
  std::vector<void*> owned;
};

void caco_decode_end(caco_decode_state* s) {
  if (!s) return;
  for (void* p : s->owned) (void)hipFree(p);
  delete s;
}

int caco_decode_begin(caco_model* m, const float* audio_hidden, const float* audio_mask, int32_t batch, int32_t seq_audio,
                      int32_t max_len, caco_decode_state** out, void* stream) {
  CACO_REQUIRE(m && m->finalized && out, "caco_decode_begin: bad arguments");
  CACO_TRY(check_device(m));
  CACO_REQUIRE(!m->dlayers.empty() && m->dec_proj.w, "Decoder module not initialized");
  CACO_REQUIRE(audio_hidden && audio_mask && batch > 0 && seq_audio > 0 && max_len > 0, "caco_decode_begin: bad arguments");
  CACO_REQUIRE(max_len <= m->cfg.text_max_pos, "caco_decode_begin: max_len %d exceeds max_position_embeddings %d", max_len, m->cfg.text_max_pos);
  hipStream_t st = (hipStream_t)stream;
  const caco_config& c = m->cfg;
  const int H = c.text_hidden, I = c.text_intermediate, Vp = m->dec_proj.out;
  const int64_t Ma = (int64_t)batch * seq_audio;
  caco_decode_state* s = new (std::nothrow) caco_decode_state();
  CACO_REQUIRE(s, "caco_decode_begin: out of host memory");
  s->m = m; s->batch = batch; s->seq_a = seq_audio; s->max_len = max_len;
  auto dev = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    s->owned.push_back(p);
    return p;
  };
  bool ok = true;
  const size_t cache_bytes = (size_t)batch * max_len * 2 * H * 2;
  for (size_t n = 0; n < m->tlayers.size() && ok; ++n) { s->text_kv.push_back((bf16_t*)dev(cache_bytes)); ok = s->text_kv.back() != nullptr; }
  for (size_t n = 0; n < m->dlayers.size() && ok; ++n) { s->dec_kv.push_back((bf16_t*)dev(cache_bytes)); ok = s->dec_kv.back() != nullptr; }
  for (size_t n = 0; n < m->dlayers.size() && ok; ++n) { s->cross_kv.push_back((bf16_t*)dev((size_t)Ma * 2 * H * 2)); ok = s->cross_kv.back() != nullptr; }
  s->audio_mask = ok ? (float*)dev((size_t)Ma * 4) : nullptr;
  // per-step scratch: x, y fp32 [B,H]; xb, o bf16 [B,H]; qkv bf16 [B, qkv_ld]; a bf16 [B,I]; logits fp32 [B,Vp]
  s->ws_bytes = (size_t)batch * ((size_t)H * 4 * 2 + (size_t)H * 2 * 2 + (size_t)qkv_ld(H) * 2 + (size_t)I * 2 + (size_t)Vp * 4) + 4096;
  s->ws = ok && s->audio_mask ? (char*)dev(s->ws_bytes) : nullptr;
  bf16_t* ab = ok && s->ws ? (bf16_t*)dev((size_t)Ma * H * 2) : nullptr;       // bf16 audio rows (only needed here)
  if (!ok || !s->audio_mask || !s->ws || !ab) {
    caco_decode_end(s);
    set_error("caco_decode_begin: out of device memory");
    return CACO_ERR_HIP;
  }
  int rc = CACO_OK;
  if (hipMemcpyAsync(s->audio_mask, audio_mask, (size_t)Ma * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = CACO_ERR_HIP;
  if (!rc) rc = cast_f32_to_bf16(audio_hidden, ab, Ma * H, st);
  for (size_t n = 0; n < m->dlayers.size() && !rc; ++n) rc = linear_bf16(m->dlayers[n].ckv, ab, Ma, ACT_NONE, s->cross_kv[n], st);
  if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = CACO_ERR_HIP;         // `ab` is released below
  if (rc) {
    caco_decode_end(s);
    return rc;
  }
  (void)hipFree(ab);
  s->owned.pop_back();
  *out = s;
  return CACO_OK;
}

// token_ids int64 [B] = the token at position s->pos of every clip (BOS at the first call); logits fp32 [B, vocab] = the
// decoder's distribution over the NEXT token.
int caco_decode_step(caco_decode_state* s, const int64_t* token_ids, float* logits, void* stream) {
  CACO_REQUIRE(s && s->m && token_ids && logits, "caco_decode_step: bad arguments");
  CACO_TRY(check_device(s->m));
  CACO_REQUIRE(s->pos < s->max_len, "caco_decode_step: position %d reached max_len %d", s->pos, s->max_len);
  caco_model* m = s->m;
  hipStream_t st = (hipStream_t)stream;
  const caco_config& c = m->cfg;
  const int H = c.text_hidden, I = c.text_intermediate, V = c.text_vocab, Vp = m->dec_proj.out;
  const int heads = c.text_heads, hd = H / heads, B = s->batch, ldq = qkv_ld(H), pos = s->pos, T = pos + 1;
  char* w = s->ws;
  auto take = [&](size_t bytes) { char* p = w; w += (bytes + 255) & ~(size_t)255; return p; };
  float* x = (float*)take((size_t)B * H * 4);
  float* y = (float*)take((size_t)B * H * 4);
  bf16_t* xb = (bf16_t*)take((size_t)B * H * 2);
  bf16_t* o = (bf16_t*)take((size_t)B * H * 2);
  bf16_t* qkv = (bf16_t*)take((size_t)B * ldq * 2);
  bf16_t* a = (bf16_t*)take((size_t)B * I * 2);
  float* lg = (float*)take((size_t)B * Vp * 4);
  CACO_REQUIRE((size_t)(w - s->ws) <= s->ws_bytes, "caco_decode_step: scratch overflow");
  // one layer's self-attention on the new rows: fused QKV, K | V appended to the cache, one query per clip against T keys
  auto self_attn = [&](const TextLayer& L, bf16_t* cache) -> int {
    CACO_TRY(linear_bf16(L.qkv, xb, B, ACT_NONE, qkv, st, ldq));
    CACO_HIP(hipMemcpy2DAsync(cache + (size_t)pos * 2 * H, (size_t)s->max_len * 2 * H * 2, qkv + H, (size_t)ldq * 2, (size_t)2 * H * 2, (size_t)B,
                              hipMemcpyDeviceToDevice, st));
    CACO_TRY(attention_qkv(qkv, ldq, 1, cache, 2 * H, 0, H, nullptr, B, T, heads, hd, 0, o, st, s->max_len));
    CACO_TRY(linear_f32(L.attn_out, o, B, x, y, st));
    return layernorm(y, L.ln_attn.g, L.ln_attn.b, B, H, c.text_ln_eps, x, xb, st);
  };
  auto mlp = [&](const TextLayer& L) -> int {
    CACO_TRY(linear_bf16(L.inter, xb, B, ACT_GELU, a, st));
    CACO_TRY(linear_f32(L.out, a, B, x, y, st));
    return layernorm(y, L.ln_out.g, L.ln_out.b, B, H, c.text_ln_eps, x, xb, st);
  };
  // text tower on the new token (RobertaEmbeddings + 12 causal layers; every prefix token is kept, as in the reference's loop)
  CACO_STAGE("decode.embed_ln", text_embed_ln(token_ids, nullptr, m->word, m->pos, m->type0, m->emb_ln.g, m->emb_ln.b, B, 1, H, c.text_vocab,
                                              c.text_max_pos, c.text_ln_eps, x, xb, st, pos));
  for (size_t n = 0; n < m->tlayers.size(); ++n) {
    CACO_STAGE("decode.text_layer", self_attn(m->tlayers[n], s->text_kv[n]));
    CACO_STAGE("decode.text_layer", mlp(m->tlayers[n]));
  }
  // decoder layers: self-attention (cache), cross-attention over the audio tokens (keys / values from caco_decode_begin), MLP
  for (size_t n = 0; n < m->dlayers.size(); ++n) {
    const DecLayer& L = m->dlayers[n];
    CACO_STAGE("decode.dec_layer", self_attn(L.t, s->dec_kv[n]));
    CACO_STAGE("decode.dec_layer", linear_bf16(L.cq, xb, B, ACT_NONE, qkv, st));
    CACO_STAGE("decode.dec_layer", attention_qkv(qkv, H, 1, s->cross_kv[n], 2 * H, 0, H, s->audio_mask, B, s->seq_a, heads, hd, 0, o, st));
    CACO_STAGE("decode.dec_layer", linear_f32(L.cattn_out, o, B, x, y, st));
    CACO_STAGE("decode.dec_layer", layernorm(y, L.ln_cross.g, L.ln_cross.b, B, H, c.text_ln_eps, x, xb, st));
    CACO_STAGE("decode.dec_layer", mlp(L.t));
  }
  CACO_STAGE("decode.vocab", linear_f32(m->dec_proj, xb, B, nullptr, lg, st));
  CACO_HIP(hipMemcpy2DAsync(logits, (size_t)V * 4, lg, (size_t)Vp * 4, (size_t)V * 4, (size_t)B, hipMemcpyDeviceToDevice, st));
  s->pos = T;
  return CACO_OK;
}

int caco_encode_audio_ex(caco_model* m, const float* wav, const int64_t* lengths, int32_t batch, int64_t n_samples,
                         int32_t max_patches, float* emb, int32_t ld_emb, void* stream) {
  CACO_TRY(check_audio_shapes(m, batch, max_patches));
  CACO_REQUIRE(wav && emb && n_samples > 0, "caco_encode_audio: bad arguments");
  // the fused front end emits 16 x 16 patches of a 128-bin mel spectrogram: 256 values per patch, 8 frequency patches
  CACO_REQUIRE(m->cfg.patch_size == 256 && m->cfg.num_freq_patches == 8,
               "caco_encode_audio: the fused front end needs patch_size 256 / num_freq_patches 8 (model has %d / %d)",
               m->cfg.patch_size, m->cfg.num_freq_patches);
  hipStream_t st = (hipStream_t)stream;
  // Only the embedding leaves this entry point, and a padded patch (mask 0) can neither be attended to nor pooled: the tower
  // runs on the patches the longest clip really has - 496 of the reference's patches_seq_len = 500 for 10 s clips
  // (eval_caco_torch.py:573: 0.8 % of every GEMM / LayerNorm / attention row) - instead of on the padded window.
  // (caco_audio_forward keeps the caller's S: it returns hidden states for every position, padded ones included.)
  {
    const int64_t full = ((n_samples + 159) / 160 / 16) * 8;
    if (full >= 1 && full < max_patches) max_patches = (int32_t)full;
  }
  // front-end outputs live in their own arena so that the forward's arena growth cannot move them
  const size_t n_tok = (size_t)batch * max_patches;
  Arena F(m, WS_FRONTEND, st);
  const size_t o_p = F.reserve(n_tok * 256 * 2), o_i = F.reserve(3 * n_tok * 4);
  CACO_TRY(F.commit(st));
  bf16_t* patches = F.at<bf16_t>(o_p);
  float* tinds = F.at<float>(o_i);
  float* finds = tinds + n_tok;
  float* mask = finds + n_tok;
  CACO_STAGE("mel.patches", mel_frontend(wav, batch, n_samples, max_patches, 0.2f, 0.9f, patches, MEL_PATCH_BF16, tinds, finds, mask, st, lengths));
  return audio_forward_impl(m, patches, CACO_DTYPE_BF16, tinds, finds, mask, batch, max_patches, 1, emb, ld_emb, nullptr, stream);
}
int caco_encode_audio(caco_model* m, const float* wav, int32_t batch, int64_t n_samples, int32_t max_patches, float* emb,
                      void* stream) {
  return caco_encode_audio_ex(m, wav, nullptr, batch, n_samples, max_patches, emb, 0, stream);
}

int caco_similarity_ld(const float* a, int32_t na, int32_t lda, const float* t, int32_t nt, int32_t ldt, int32_t dim, float scale,
                       float* out, int32_t ld_out, void* stream) {
  CACO_REQUIRE(a && t && out, "caco_similarity: null argument");
  hipStream_t st = (hipStream_t)stream;
  CACO_STAGE("similarity", gemm_f32(a, t, nullptr, out, na, nt, dim, ld_out, scale, st, lda, ldt));
  return CACO_OK;
}
int caco_similarity(const float* a, int32_t na, const float* t, int32_t nt, int32_t dim, float scale, float* out,
                    int32_t ld_out, void* stream) {
  CACO_REQUIRE(a && t && out, "caco_similarity: null argument");
  hipStream_t st = (hipStream_t)stream;
  CACO_STAGE("similarity", gemm_f32(a, t, nullptr, out, na, nt, dim, ld_out, scale, st));
  return CACO_OK;
}

int caco_topk(const float* sim, int32_t rows, int32_t cols, int64_t row_stride, int64_t col_stride, int32_t k, int32_t* idx,
              float* val, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  CACO_STAGE("retrieval.topk", topk_rows(sim, rows, cols, row_stride, col_stride, k, idx, val, st));
  return CACO_OK;
}

int caco_token_group_mean(const float* hidden, int32_t batch, int32_t seq, int32_t dim, int32_t group, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  CACO_STAGE("hear.token_group_mean", token_group_mean(hidden, batch, seq, dim, group, out, st));
  return CACO_OK;
}

int caco_l2_normalize(const float* x, int32_t rows, int32_t dim, float* out, void* stream) {
  return l2_normalize(x, rows, dim, out, (hipStream_t)stream);
}

int caco_mae_forward(caco_model* m, const void* patches, int32_t dtype, const float* mask, const float* tinds,
                     const float* finds, const float* rtinds, const float* rfinds, const float* rmask, int32_t batch,
                     int32_t nv, int32_t nr, float* out, void* stream) {
  CACO_TRY(check_audio_shapes(m, batch, nv));
  CACO_REQUIRE(m->cfg.mae_decoder_layers > 0 && m->dec.restore_patch, "caco_mae_forward: model has no AudioMAE decoder");
  CACO_REQUIRE(patches && mask && tinds && finds && rtinds && rfinds && rmask && out && nr >= 0, "caco_mae_forward: null argument");
  hipStream_t st = (hipStream_t)stream;
  const caco_config& c = m->cfg;
  const int H = c.audio_hidden, P = c.patch_size, S = nv + nr;
  const int64_t Mv = (int64_t)batch * nv, Mr = (int64_t)batch * nr, M = (int64_t)batch * S;
  Arena A(m, WS_AUDIO, st);
  AudioWs w;
  w.plan(A, M, batch, S, H, c.audio_intermediate);      // sized for the decoder (S >= V); the encoder reuses it
  const size_t o_pb = A.reserve((size_t)Mv * P * 2), o_tmp = A.reserve((size_t)M * H * 4);
  const size_t o_mask = A.reserve((size_t)M * 4);
  CACO_TRY(A.commit(st));
  float* x = A.at<float>(w.x);
  bf16_t* h = A.at<bf16_t>(w.h);
  float* tmp = A.at<float>(o_tmp);
  float* cmask = A.at<float>(o_mask);
  const bf16_t* pb = nullptr;
  CACO_TRY(patches_as_bf16(patches, dtype, Mv * P, A.at<bf16_t>(o_pb), &pb, st));
  // encoder on the visible patches (mae.py:228-234)
  CACO_TRY(linear_f32(m->enc.input_proj, pb, Mv, nullptr, x, st));
  CACO_TRY(add_pos_embed(x, nullptr, tinds, finds, m->enc.freq_table, Mv, H, c.num_freq_patches, st));
  CACO_TRY(run_audio_layers(m, m->enc.layers, A, w, mask, batch, nv, c.audio_heads, c.audio_ln_eps, st));
  CACO_TRY(layernorm(x, m->enc.norm.g, m->enc.norm.b, Mv, H, c.audio_ln_eps, nullptr, h, st));
  // AudioDecoder.forward (mae.py:166-207)
  CACO_TRY(linear_f32(m->dec.input_proj, h, Mv, nullptr, tmp, st));                                            // :177
  CACO_TRY(add_pos_embed(tmp, nullptr, tinds, finds, m->dec.freq_table, Mv, H, c.num_freq_patches, st));       // :179-186
  CACO_TRY(copy_rows(tmp, x, batch, nv, S, 0, H, st));
  if (nr > 0) {
    CACO_TRY(add_pos_embed(tmp, m->dec.restore_patch, rtinds, rfinds, m->dec.freq_table, Mr, H, c.num_freq_patches, st));  // :188-196
    CACO_TRY(copy_rows(tmp, x, batch, nr, S, nv, H, st));                                                      // :198
    CACO_HIP(hipMemcpy2DAsync(cmask + nv, (size_t)S * 4, rmask, (size_t)nr * 4, (size_t)nr * 4, batch, hipMemcpyDeviceToDevice, st));
  }
  CACO_HIP(hipMemcpy2DAsync(cmask, (size_t)S * 4, mask, (size_t)nv * 4, (size_t)nv * 4, batch, hipMemcpyDeviceToDevice, st));  // :199
  CACO_TRY(run_audio_layers(m, m->dec.layers, A, w, cmask, batch, S, c.audio_heads, c.audio_ln_eps, st));
  CACO_TRY(layernorm(x, m->dec.norm.g, m->dec.norm.b, M, H, c.audio_ln_eps, nullptr, h, st));                  // :204
  return linear_f32(m->dec.output_proj, h, M, nullptr, out, st);                                               // :205
}

// ---- op-level entry points (bench roofline leg + unit tests) -------------------------------------
int caco_op_gemm_bf16(const void* a, const void* w, const float* bias, int64_t M, int32_t N, int32_t K, int32_t act,
                      void* out, void* stream) {
  GemmArgs g{(const bf16_t*)a, (const bf16_t*)w, bias, nullptr, out, M, N, K, N};
  return gemm_bf16(g, EPI_BF16, act, (hipStream_t)stream);
}
int caco_op_gemm_bf16_strided(const void* a, int32_t lda, const void* w, int32_t ldw, const float* bias, int64_t M, int32_t N,
                              int32_t K, int32_t act, void* out, int32_t ldc, void* stream) {
  GemmArgs g{(const bf16_t*)a, (const bf16_t*)w, bias, nullptr, out, M, N, K, ldc, lda, ldw};
  return gemm_bf16(g, EPI_BF16, act, (hipStream_t)stream);
}
int caco_op_gemm_bf16_f32out(const void* a, const void* w, const float* bias, const float* resid, int64_t M, int32_t N,
                             int32_t K, float* out, void* stream) {
  GemmArgs g{(const bf16_t*)a, (const bf16_t*)w, bias, resid, out, M, N, K, N};
  return gemm_bf16(g, EPI_F32, ACT_NONE, (hipStream_t)stream);
}
int caco_op_layernorm(const float* x, const float* g, const float* b, int64_t rows, int32_t dim, float eps, float* of,
                      void* ob, void* stream) {
  return layernorm(x, g, b, rows, dim, eps, of, (bf16_t*)ob, (hipStream_t)stream);
}

int caco_op_attention_qkv(const void* q, int32_t q_ld, int32_t seq_q, const void* kv, int32_t ld, int32_t k_off, int32_t v_off,
                          const float* mask, int32_t batch, int32_t seq, int32_t heads, int32_t head_dim, int32_t causal,
                          void* out, void* stream) {
  CACO_REQUIRE(q && kv && out, "caco_op_attention_qkv: null argument");
  return attention_qkv((const bf16_t*)q, q_ld, seq_q, (const bf16_t*)kv, ld, k_off, v_off, mask, batch, seq, heads, head_dim,
                       causal, (bf16_t*)out, (hipStream_t)stream);
}

int caco_op_attention(const void* qkv, int32_t ld, int32_t k_off, int32_t v_off, const float* mask, int32_t batch,
                      int32_t seq, int32_t heads, int32_t head_dim, int32_t causal, void* out, void* stream) {
  CACO_REQUIRE(qkv && out, "caco_op_attention: null argument");
  return attention((const bf16_t*)qkv, ld, k_off, v_off, mask, batch, seq, heads, head_dim, causal, (bf16_t*)out,
                   (hipStream_t)stream);
}

}  // extern "C"
