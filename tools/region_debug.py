#!/usr/bin/env python3
"""Diagnostics for the region decoder (csrc/hb_lz4_region.hip): decode an ORACLE-written (index-less) frame on the device and
print what the guess-and-verify stages did -- chain verification, output total, whether the rebuilt index was used, per-stage ms.

  python tools/region_debug.py [--mib 64] [--dataset f32] [--shuffle 1] [--typesize 4]
"""
import argparse
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np
import torch

import hipblosc as hb
import bench
import oracle as O

RG_MAXREG, REG_BYTES = 16384, 64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=64)
    ap.add_argument("--dataset", default="f32")
    ap.add_argument("--shuffle", type=int, default=1)
    ap.add_argument("--typesize", type=int, default=4)
    ap.add_argument("--writer", default="oracle", choices=["oracle", "device"],
                    help="oracle: the restated reference encoder (64 KiB window); device: this library WITHOUT the index trailer")
    ap.add_argument("--small-work", action="store_true", help="workspace without the symbolic decoder's scratch (foreign frames then decode on one wavefront)")
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--dump", default="", help="lo:hi -- print the final state of these regions")
    a = ap.parse_args()
    L = hb.lib()
    assert L.hb_init() == 0
    n = a.mib << 20
    x = bench.synth_host(a.dataset, n, 0)
    if a.writer == "oracle":
        f = O.compress_frame(x, shuffle=a.shuffle, typesize=a.typesize)
    else:
        f = np.frombuffer(hb.Compress(x.tobytes(), hb.LZ4, 5, a.shuffle, a.typesize, opts=0), np.uint8)
    dev = torch.device("cuda", 0)
    d_frame = torch.from_numpy(f.copy()).to(dev)
    pad = torch.zeros(64, dtype=torch.uint8, device=dev)
    d_frame = torch.cat([d_frame, pad])
    d_out = torch.zeros(n, dtype=torch.uint8, device=dev)
    wb = L.hb_decompress_frame_workspace(n) if a.small_work else L.hb_decompress_frame_workspace_foreign(n)
    work = torch.zeros(wb, dtype=torch.uint8, device=dev)
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    import time
    for rep in range(a.reps):
        L.hb_profile_enable(1)
        t0 = time.perf_counter()
        rc = L.hb_decompress_frame_dev(d_frame.data_ptr(), f.size, d_out.data_ptr(), n, 0, work.data_ptr(), wb, res.data_ptr(), stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = bench.stage_times()
        L.hb_profile_enable(0)
        print(f"rep {rep}: {dt * 1e3:.2f} ms wall with profiling events = {n / dt / 1e9:.1f} GB/s")
    r = res.cpu().numpy().view(np.uint8)
    print("rc", rc, "status", int(r[:4].view(np.int32)[0]), "flags", int(r[4:8].view(np.uint32)[0]), "bytes", int(r[8:16].view(np.uint64)[0]))
    print("ratio", f.size / n, "stage ms", {k: round(sum(v), 3) for k, v in st.items()})
    print("per launch:", {k: [round(x, 3) for x in v] for k, v in st.items() if len(v) > 1})
    w = work.cpu().numpy()
    base = ((n + 255) & ~255) + 256 + 256            # staged buffer, DecPlan, then the region workspace
    plan = struct.unpack("<4IQ", w[base:base + 24].tobytes())
    print("RgPlan ok/fail/nreg/rs/total", plan)
    nreg = plan[2]
    regs = w[base + 256: base + 256 + nreg * REG_BYTES].view(np.uint32).reshape(nreg, 16)
    b, entry, exit_, outlen, entry0, exit0, outlen0, ntrace, needfull = [regs[:, i] for i in range(9)]
    print("needfull", int(needfull.sum()), "invalid exits", int((exit_ == 0xFFFFFFFF).sum()), "entry!=prev exit", int((entry[1:] != exit_[:-1]).sum()),
          "entry moved from guess", int((entry != b).sum()), "empty regions", int((outlen == 0).sum()))
    for i in np.nonzero((needfull != 0) | (exit_ == 0xFFFFFFFF))[0][:10]:
        print(f"  pending r{i}: b {b[i]} entry {entry[i]} exit {exit_[i]} outlen {outlen[i]} | entry0 {entry0[i]} exit0 {exit0[i]} outlen0 {outlen0[i]} ntrace {ntrace[i]} needfull {needfull[i]} pad0 {regs[i, 9]}")
    if a.dump:
        lo, hi = (int(v) for v in a.dump.split(":"))
        for i in range(lo, min(hi, nreg)):
            print(f"  r{i}: b {b[i]} entry {entry[i]} exit {exit_[i]} outlen {outlen[i]} | entry0 {entry0[i]} exit0 {exit0[i]} outlen0 {outlen0[i]} ntrace {ntrace[i]} needfull {needfull[i]} pad0 {regs[i, 9]}")
    bad = np.nonzero(entry[1:] != exit_[:-1])[0][:8]
    for i in bad:
        print("  region", i + 1, "b", b[i + 1], "entry", entry[i + 1], "prev exit", exit_[i], "exit", exit_[i + 1], "needfull", needfull[i + 1])
    print("equal to input:", bool(np.array_equal(d_out.cpu().numpy(), x if a.shuffle == 0 else x)))


if __name__ == "__main__":
    main()
