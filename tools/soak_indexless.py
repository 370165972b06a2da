#!/usr/bin/env python3
"""soak: composite data (noise, zeros, ramps, text, few-valued, periodic) x filters, as this library writes it without the trailer (index from stored
tokens) and as the oracle (reference-shaped encoder) writes it (symbolic decoder, token-fed), through the host API, against the input"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hipblosc as hb, oracle as O
t0 = time.time()
bad = cases = 0
def piece(rng, n):
    k = int(rng.integers(0, 7))
    if k == 0: return rng.integers(0, 256, n, dtype=np.uint8)
    if k == 1: return np.zeros(n, np.uint8)
    if k == 2: return (np.arange(n // 4 + 1, dtype=np.float32) * 0.1).view(np.uint8)[:n]
    if k == 3: return np.frombuffer((b"the quick brown fox jumps over the lazy dog. " * (n // 45 + 1))[:n], np.uint8)
    if k == 4: return (rng.integers(0, 4, n, dtype=np.uint8) * 64)
    if k == 5: return np.tile(rng.integers(0, 256, int(rng.integers(3, 9000)), dtype=np.uint8), n // 3 + 1)[:n]
    return np.full(n, int(rng.integers(0, 256)), np.uint8)
for seed in range(2000, 2000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 30)):
    rng = np.random.default_rng(seed)
    total = int(rng.choice([1 << 20, 5 << 20, 17 << 20, 40 << 20])) + int(rng.integers(0, 5000))
    parts = []
    left = total
    while left > 0:
        m = min(left, int(rng.integers(1, max(2, total // 3))))
        parts.append(piece(rng, m)); left -= m
    x = np.concatenate(parts)[:total]
    shuffle, ts = [(0, 1), (1, 4), (2, 4), (1, 8), (1, 2), (2, 8)][seed % 6]
    raw = x.tobytes()
    f_own = hb.Compress(raw, hb.LZ4, 5, shuffle, ts, opts=0)
    ok1 = hb.Decompress(f_own) == raw
    p1 = hb.lib().hb_last_result_flags() & 1
    f_ref = O.compress_frame(x, shuffle=shuffle, typesize=ts).tobytes()
    ok2 = hb.Decompress(f_ref) == raw
    p2 = hb.lib().hb_last_result_flags() & 1
    cases += 1
    if not (ok1 and ok2):
        bad += 1
        print("MISMATCH seed", seed, "n", total, "shuffle", shuffle, "ts", ts, "own ok", ok1, "ref ok", ok2, flush=True)
    if cases % 5 == 0:
        print(f"{cases} cases, {bad} bad, parallel own/ref {p1}/{p2}, {time.time() - t0:.0f} s", flush=True)
print("done:", cases, "cases", bad, "bad", f"{time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
