#!/usr/bin/env python3
"""decode rate of C-Blosc-1 frames (written by c-blosc 1.21, /opt/conda/lib/libblosc.so.1) on the device.

  python tools/cblosc_rates.py [--mib 256]
Prints per-stage ms (HIP events, hb_profile_*) of hb_cblosc_decompress for a few writer settings, next to libblosc's own
single-thread decode rate on the host.
"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np

import hipblosc as hb
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=256)
    a = ap.parse_args()
    L = hb.lib()
    assert L.hb_init() == 0
    B = ctypes.CDLL("/opt/conda/lib/libblosc.so.1")
    B.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    B.blosc_decompress_ctx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    n = a.mib << 20
    rows = []
    for kind, ts, shuffle, clevel, bs in (("f32", 4, 1, 5, 0), ("f32", 4, 1, 9, 0), ("f32", 4, 1, 1, 0), ("f64", 8, 1, 5, 0), ("i32", 4, 2, 5, 0), ("f32", 4, 1, 5, 65536)):
        x = bench.synth_host(kind, n, 0)
        dst = np.empty(n + (1 << 20), np.uint8)
        c = B.blosc_compress_ctx(clevel, shuffle, ts, n, x.ctypes.data, dst.ctypes.data, dst.size, b"lz4", bs, 1)
        f = dst[:c].tobytes()
        h = hb.CBloscParseHeader(f)
        back = np.empty(n, np.uint8)
        t0 = time.perf_counter(); r = B.blosc_decompress_ctx(dst.ctypes.data, back.ctypes.data, n, 1); t_cpu = time.perf_counter() - t0
        assert r == n
        fr = np.frombuffer(f, np.uint8)
        assert L.hb_cblosc_decompress(fr.ctypes.data, fr.size, back.ctypes.data, n, 0) == n     # first call allocates
        assert np.array_equal(back, x)
        L.hb_profile_enable(1)
        t0 = time.perf_counter(); r = L.hb_cblosc_decompress(fr.ctypes.data, fr.size, back.ctypes.data, n, 0); dt = time.perf_counter() - t0
        st = bench.stage_times()
        L.hb_profile_enable(0)
        assert r == n
        # the other direction: this library writes (block size 4096 x typesize), c-blosc reads
        cap = L.hb_cblosc_bound(n, ts)
        ours = np.empty(cap, np.uint8)
        cc = L.hb_cblosc_compress(x.ctypes.data, n, ours.ctypes.data, cap, shuffle, ts, 0)
        assert cc > 0
        L.hb_profile_enable(1)
        cc = L.hb_cblosc_compress(x.ctypes.data, n, ours.ctypes.data, cap, shuffle, ts, 0)
        stc = bench.stage_times()
        L.hb_profile_enable(0)
        t0 = time.perf_counter(); r2 = B.blosc_decompress_ctx(ours.ctypes.data, back.ctypes.data, n, 1); t_cpu2 = time.perf_counter() - t0
        assert r2 == n and np.array_equal(back, x)
        enc_ms = sum(sum(v) for v in stc.values())
        print(f"    written here: ratio {cc / n:.3f}, stages {dict((k, round(sum(v), 3)) for k, v in stc.items())} = {n / enc_ms / 1e6:.1f} GB/s device-resident; "
              f"c-blosc reads it at {n / t_cpu2 / 1e9:.2f} GB/s (1 thread)")
        rows.append({"data": kind, "written_here": {"ratio": round(cc / n, 4), "stage_ms": {k: round(sum(v), 3) for k, v in stc.items()},
                                                    "device_resident_GBps": round(n / enc_ms / 1e6, 1), "libblosc_reads_it_1_thread_GBps": round(n / t_cpu2 / 1e9, 2)}, "typesize": ts, "shuffle": shuffle, "clevel": clevel, "blocksize": h.blocksize, "ratio": round(c / n, 4),
                     "stage_ms": {k: round(sum(v), 3) for k, v in st.items()}, "device_resident_GBps": round(n / sum(sum(v) for v in st.values()) / 1e6, 1),
                     "host_to_host_GBps_pageable": round(n / dt / 1e9, 2), "libblosc_1_thread_GBps": round(n / t_cpu / 1e9, 2)})
        dev_ms = sum(sum(v) for v in st.values())
        print(f"{kind} ts{ts} shuffle{shuffle} clevel{clevel}: blocksize {h.blocksize}, ratio {c / n:.3f}, stages {dict((k, round(sum(v), 3)) for k, v in st.items())} "
              f"= {n / dev_ms / 1e6:.1f} GB/s device-resident; host->host {n / dt / 1e9:.2f} GB/s; libblosc 1 thread {n / t_cpu / 1e9:.2f} GB/s")
    import json
    print(json.dumps({"workload": f"{a.mib} MiB frames written by c-blosc 1.21 (lz4), decoded by hb_cblosc_decompress on one MI355X", "rows": rows}))


if __name__ == "__main__":
    main()

