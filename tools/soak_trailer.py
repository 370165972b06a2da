#!/usr/bin/env python3
"""soak: composite data (noise, zeros, ramps, text, few-valued, periodic, runs of every length class) x filters x frame shapes as THIS library writes it
(round 4: run step, INNER step body, gated extension, lean window parser with pointer doubling) through the host API, against the input and against the
oracle's restatement of the reference decoder; a batch of the same frames through hb_decompress_frames_batch (rebuilt indexes) on top"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hipblosc as hb, oracle as O
t0 = time.time()
bad = cases = 0
def runs(rng, n, lo, hi, alphabet):
    out = np.empty(n + hi, np.uint8); i = 0
    while i < n:
        k = int(rng.integers(lo, hi + 1)); out[i:i + k] = rng.integers(0, alphabet); i += k
    return out[:n]
def piece(rng, n):
    k = int(rng.integers(0, 10))
    if k == 0: return rng.integers(0, 256, n, dtype=np.uint8)
    if k == 1: return np.zeros(n, np.uint8)
    if k == 2: return (np.arange(n // 4 + 1, dtype=np.float32) * 0.1).view(np.uint8)[:n]
    if k == 3: return np.frombuffer((b"the quick brown fox jumps over the lazy dog. " * (n // 45 + 1))[:n], np.uint8)
    if k == 4: return (rng.integers(0, 4, n, dtype=np.uint8) * 64)
    if k == 5: return np.tile(rng.integers(0, 256, int(rng.integers(3, 9000)), dtype=np.uint8), n // 3 + 1)[:n]
    if k == 6: lo = int(rng.integers(1, 40)); return runs(rng, n, lo, lo + int(rng.integers(0, 40)), int(rng.choice([2, 4, 256])))
    if k == 7: u = runs(rng, int(rng.integers(200, 3000)), 4, 12, 256); return np.tile(u, n // u.size + 1)[:n]
    if k == 8: y = rng.integers(0, 256, n, dtype=np.uint8); y[rng.integers(0, n, n // 3 + 1)] = 255; return y
    return np.full(n, int(rng.integers(0, 256)), np.uint8)
frames_for_batch = []
for seed in range(3000, 3000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 30)):
    rng = np.random.default_rng(seed)
    total = int(rng.choice([1 << 20, 3 << 20, 9 << 20, 33 << 20])) + int(rng.choice([0, 0, 4096 * 4, 12345]))
    parts = []; left = total
    while left > 0:
        m = min(left, int(rng.integers(1, max(2, total // 3))))
        parts.append(piece(rng, m)); left -= m
    x = np.concatenate(parts)[:total]
    shuffle, ts = [(0, 1), (1, 4), (2, 4), (1, 8), (1, 2), (2, 8)][seed % 6]
    raw = x.tobytes()
    for opts in (hb.OPT_INDEX_TRAILER, 0):
        f = hb.Compress(raw, hb.LZ4, 5, shuffle, ts, opts=opts)
        ok1 = hb.Decompress(f) == raw
        par = hb.lib().hb_last_result_flags() & 1
        ok2 = O.decompress_frame(np.frombuffer(f, np.uint8)).tobytes() == raw
        cases += 1
        if not (ok1 and ok2):
            bad += 1
            print("MISMATCH seed", seed, "n", total, "shuffle", shuffle, "ts", ts, "opts", opts, "device ok", ok1, "oracle ok", ok2, "parallel", par, flush=True)
        if opts == 0 and total <= (4 << 20):
            frames_for_batch.append((f, raw))
    if cases % 10 == 0:
        print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
if frames_for_batch:
    got = hb.DecompressBatch([f for f, _ in frames_for_batch])
    for i, (f, raw) in enumerate(frames_for_batch):
        cases += 1
        if got[i] != raw:
            bad += 1; print("BATCH MISMATCH frame", i, flush=True)
print("done:", cases, "cases", bad, "bad", f"{time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
