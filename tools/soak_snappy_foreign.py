#!/usr/bin/env python3
"""soak: composite data x filters as FOREIGN Snappy frames (the oracle's 64 KiB-block encoder, no unit index) through the element discovery and the
parallel decoders (k_sn_dec_units, the symbolic decoder behind it; both workspaces through the device API: tests/test_gpu_f3.py);
device bytes == input, and how many decoded in parallel"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hipblosc as hb, oracle as O
t0 = time.time()
bad = cases = par = 0
def piece(rng, n):
    k = int(rng.integers(0, 9))
    if k == 0: return rng.integers(0, 256, n, dtype=np.uint8)
    if k == 1: return np.zeros(n, np.uint8)
    if k == 2: return (np.arange(n // 4 + 1, dtype=np.float32) * 0.1).view(np.uint8)[:n]
    if k == 3: return np.frombuffer((b"the quick brown fox jumps over the lazy dog. " * (n // 45 + 1))[:n], np.uint8)
    if k == 4: return (rng.integers(0, 4, n, dtype=np.uint8) * 64)
    if k == 5: return np.tile(rng.integers(0, 256, int(rng.integers(3, 9000)), dtype=np.uint8), n // 3 + 1)[:n]
    if k == 6: y = rng.integers(0, 256, n, dtype=np.uint8); y[rng.integers(0, n, n // 3 + 1)] = 0xF4; return y          # noise full of literal-header tags
    if k == 7: y = rng.integers(0, 256, n, dtype=np.uint8); y[rng.integers(0, n, n // 5 + 1)] = 255; return y
    return O.synth(O.D_F32, n // 4 + 1).view(np.uint8)[:n]
for seed in range(5000, 5000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 30)):
    rng = np.random.default_rng(seed)
    total = int(rng.choice([1 << 20, 3 << 20, 9 << 20, 33 << 20])) + int(rng.choice([0, 0, 4096 * 4, 12345]))
    parts = []; left = total
    while left > 0:
        m = min(left, int(rng.integers(1, max(2, total // 3))))
        parts.append(piece(rng, m)); left -= m
    x = np.concatenate(parts)[:total]
    shuffle, ts = [(0, 1), (1, 4), (2, 4), (1, 8), (1, 2)][seed % 5]
    x = x[: total - total % ts]
    f = O.compress_frame(x, codec=O.SNAPPY, shuffle=shuffle, typesize=ts).tobytes()
    if hb.GetInfo(f).IsMemcpy():
        continue
    ok = hb.Decompress(f) == x.tobytes()
    p = hb.lib().hb_last_result_flags() & 1
    cases += 1; par += p
    if not ok:
        bad += 1; print("MISMATCH seed", seed, "n", x.size, "shuffle", shuffle, "ts", ts, "parallel", p, flush=True)
    if cases % 10 == 0:
        print(f"{cases} cases, {bad} bad, parallel {par}, {time.time() - t0:.0f} s", flush=True)
print("done:", cases, "cases", bad, "bad", "parallel", par, f"{time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
