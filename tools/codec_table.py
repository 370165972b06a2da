#!/usr/bin/env python3
"""Ratio / speed table of the device codecs: LZ4 at levels 1 / 5 / 9 (the level is a skip-acceleration knob),
LZ4HC at the four level classes of codec.go:96-106, Snappy -- device-resident compress + decompress of one D-f32 frame,
Shuffle1 typesize 4 -- next to what the CPU libraries reach on a 64 MiB sample of the same shuffled data (liblz4 fast / HC
on independent 4 KiB chunks and on one block, libsnappy).  The numbers quoted in DESIGN.md come from this script.

  python tools/codec_table.py [--mib 1024] [--out gpurun_out/codec_table.json]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np
import torch

import hipblosc as hb
import bench
import oracle as O          # data + the CPU library comparison only


def cpu_refs(sample):
    out = {}
    lz = None
    for p in ("/usr/lib/x86_64-linux-gnu/liblz4.so.1", "/opt/conda/lib/liblz4.so.1"):
        if os.path.exists(p):
            lz = ctypes.CDLL(p)
            break
    n = sample.size
    buf = sample.tobytes()
    if lz is not None:
        cap = lz.LZ4_compressBound(n)
        dst = ctypes.create_string_buffer(cap)
        out["liblz4 fast, one block"] = lz.LZ4_compress_default(buf, dst, n, cap) / n
        out["liblz4 HC level 9, one block"] = lz.LZ4_compress_HC(buf, dst, n, cap, 9) / n
        tot_f = tot_h = 0
        small = ctypes.create_string_buffer(lz.LZ4_compressBound(4096))
        m = 0
        for off in range(0, n, 4 * 4096):                      # every 4th chunk: all four byte planes of the sample
            c = buf[off:off + 4096]
            tot_f += lz.LZ4_compress_default(c, small, len(c), len(small))
            tot_h += lz.LZ4_compress_HC(c, small, len(c), len(small), 9)
            m += len(c)
        out["liblz4 fast, 4 KiB chunks"] = tot_f / m
        out["liblz4 HC level 9, 4 KiB chunks"] = tot_h / m
    p = "/opt/conda/lib/libsnappy.so.1"
    if os.path.exists(p):
        sn = ctypes.CDLL(p)
        sn.snappy_max_compressed_length.restype = ctypes.c_size_t
        sn.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]
        sn.snappy_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
        cap = sn.snappy_max_compressed_length(n)
        dst = ctypes.create_string_buffer(cap)
        ol = ctypes.c_size_t(cap)
        sn.snappy_compress(buf, n, dst, ctypes.byref(ol))
        out["libsnappy, one block (64 KiB windows)"] = ol.value / n
    return {k: round(v, 4) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    L = hb.lib()
    assert L.hb_init() == 0
    n = a.mib << 20
    d = bench.Dev(n, torch.device("cuda", 0))
    host = bench.synth_host("f32", n, 0)
    d.src.copy_(torch.from_numpy(host))
    rows = []
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for name, codec, level in (("LZ4 level 1", hb.LZ4, 1), ("LZ4 level 5 (default)", hb.LZ4, 5), ("LZ4 level 9", hb.LZ4, 9),
                               ("LZ4HC level 1-3", hb.LZ4HC, 1), ("LZ4HC level 4-5", hb.LZ4HC, 5), ("LZ4HC level 6-9", hb.LZ4HC, 9),
                               ("Snappy", hb.Snappy, 5)):
        d.compress(1, 4, hb.OPT_INDEX_TRAILER, codec, level)
        d.back.zero_(); d.decompress(); torch.cuda.synchronize()
        rc, rd = d.results()
        assert rc["status"] == 0 and rd["status"] == 0 and (rd["flags"] & 1) and torch.equal(d.back, d.src), name
        tc, td = [], []
        for _ in range(5):
            ev[0].record(); d.compress(1, 4, hb.OPT_INDEX_TRAILER, codec, level); ev[1].record(); d.decompress(); ev[2].record()
            torch.cuda.synchronize()
            tc.append(ev[0].elapsed_time(ev[1])); td.append(ev[1].elapsed_time(ev[2]))
        c, dd = float(np.median(tc)), float(np.median(td))
        rows.append({"codec": name, "ratio": round(rc["bytes"] / n, 4), "compress_ms": round(c, 3), "decompress_ms": round(dd, 3),
                     "compress_GBps": round(n / c / 1e6, 1), "decompress_GBps": round(n / dd / 1e6, 1)})
        print(rows[-1], flush=True)
    sample = O.filter(O.OP_SHUFFLE, host[: min(n, 64 << 20)], 4)
    out = {"workload": f"{a.mib} MiB D-f32, Shuffle1 typesize 4, device-resident, index trailer", "device": rows, "cpu_libraries_ratio": cpu_refs(sample)}
    print(json.dumps(out))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
