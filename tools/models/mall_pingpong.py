#!/usr/bin/env python3
"""What could ping-pong traversal (CACO_PINGPONG=1) save?  A paper model, not a measurement.

The audio layer at batch 256 as a sequence of kernels over M = 126 976 rows (496 panels of 256 rows), each kernel a stream of
panel-sized reads and writes in the order its workgroups reach them, through ONE 256 MiB cache with LRU replacement that
allocates on reads and on writes (what the Infinity Cache looked like in round 2's copy probe: working sets that fit run
faster and cooler; whether `nt` stores bypass it is unknown).  Weights are ignored (small, always hot).  Output: bytes read
from HBM per layer = read misses, for
  default   GEMMs walk the 8 XCD row ranges first to last, LayerNorm and attention walk the rows globally first to last
  pingpong  every kernel walks the 8 ranges, direction alternating from kernel to kernel
Orders are idealised (all 8 ranges advance in lockstep, a panel's accesses are atomic)."""
from collections import OrderedDict

PANELS, RANGES = 496, 8
PER = PANELS // RANGES
ROWS = 256
MB = 1 << 20
SIZE = {"x": ROWS * 768 * 4, "h": ROWS * 768 * 2, "qkv": ROWS * 2304 * 2, "o": ROWS * 768 * 2, "a": ROWS * 3072 * 2}
# (kernel, reads, writes)
LAYER = [("ln1", ["x"], ["h"]), ("qkv", ["h"], ["qkv"]), ("attn", ["qkv"], ["o"]), ("out", ["o", "x"], ["x"]),
         ("ln2", ["x"], ["h"]), ("fc1", ["h"], ["a"]), ("fc2", ["a", "x"], ["x"])]
RANGE_KERNELS = {"qkv", "out", "fc1", "fc2"}            # persistent GEMMs: XCD-owned ranges also in the default build


class LRU:
    def __init__(self, cap):
        self.cap, self.used, self.d = cap, 0, OrderedDict()

    def touch(self, key, size):
        hit = key in self.d
        if hit:
            self.d.move_to_end(key)
        else:
            self.d[key] = size
            self.used += size
            while self.used > self.cap:
                _, s = self.d.popitem(last=False)
                self.used -= s
        return hit


def panel_order(ranges, desc):
    if not ranges:
        return list(range(PANELS - 1, -1, -1)) if desc else list(range(PANELS))
    out = []
    for s in range(PER):
        for j in range(RANGES):
            out.append(j * PER + (PER - 1 - s if desc else s))
    return out


def run(pingpong, layers=3, cap=256 * MB, read_alloc=True):
    """read_alloc=False: reads do not allocate (what streaming loads would do if the `nt` hint reaches this cache): every
    buffer of the layer is consumed once, so only written lines are worth keeping."""
    cache, k, per_layer = LRU(cap), 0, []
    for layer in range(layers):
        miss = total = 0
        for name, reads, writes in LAYER:
            ranges = pingpong or name in RANGE_KERNELS
            desc = pingpong and (k & 1)
            for p in panel_order(ranges, desc):
                for b in reads:
                    total += SIZE[b]
                    if (b, p) in cache.d:
                        cache.d.move_to_end((b, p))
                    else:
                        miss += SIZE[b]
                        if read_alloc:
                            cache.touch((b, p), SIZE[b])
                for b in writes:
                    cache.touch((b, p), SIZE[b])
            k += 1
        per_layer.append((miss, total))
    return per_layer[-1]


if __name__ == "__main__":
    d, p = run(False, read_alloc=False), run(True, read_alloc=False)
    print(f"reads that do not allocate, 256 MiB: from HBM  default {d[0] / 1e9:.2f} GB   ping-pong {p[0] / 1e9:.2f} GB")
    for cap in (256, 192, 128):
        d, p = run(False, cap=cap * MB), run(True, cap=cap * MB)
        print(f"cache {cap:3d} MiB: reads per layer {d[1] / 1e9:.2f} GB; from HBM  default {d[0] / 1e9:.2f} GB   ping-pong {p[0] / 1e9:.2f} GB"
              f"   ({(d[0] - p[0]) / 1e9:.2f} GB = {100 * (d[0] - p[0]) / (d[1] + 2.73e9):.0f} % of the layer's 6.24 GB of traffic)")
