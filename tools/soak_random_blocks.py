#!/usr/bin/env python3
"""soak: random valid LZ4 blocks (tests/tools/lz4_stream_gen.py) at many seeds / sizes / regimes through the host API (larger workspace: token store,
symbolic decoder) against the oracle decoder"""
import os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import hipblosc as hb, oracle as O, lz4_stream_gen as G
t0 = time.time()
bad = 0
cases = 0
for seed in range(1000, 1000 + int(sys.argv[1]) if len(sys.argv) > 1 else 1040):
    rng = np.random.default_rng(seed)
    size = int(rng.choice([1 << 20, 3 << 20, 9 << 20, 24 << 20, 48 << 20])) + seed * 4099
    regime = (8 << 10) << int(rng.integers(0, 8))
    flags, ts = [(0, 1), (1, 4), (4, 4), (1, 8), (0, 1)][seed % 5]
    block, n = G.random_block(rng, size, regime_len=regime, align=32 if flags == 4 else ts)
    frame = struct.pack("<BBBBIII", 2, hb.LZ4, flags, ts, n, n, 16 + len(block)) + block
    want = O.decompress_frame(np.frombuffer(frame, np.uint8)).tobytes()
    got = hb.Decompress(frame)
    par = hb.lib().hb_last_result_flags() & 1
    cases += 1
    if got != want:
        bad += 1
        print("MISMATCH seed", seed, "size", n, "regime", regime, "flags", flags, "parallel", par, flush=True)
    if cases % 10 == 0:
        print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done:", cases, "cases", bad, "bad", f"{time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
