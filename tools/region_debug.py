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
    ap.add_argument("--codec", default="lz4", choices=["lz4", "snappy"], help="snappy: the oracle's Snappy encoder (64 KiB blocks) -> element discovery + k_sn_dec_units")
    ap.add_argument("--small-work", action="store_true", help="workspace without the symbolic decoder's scratch (foreign frames then decode on one wavefront)")
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--dump", default="", help="lo:hi -- print the final state of these regions")
    ap.add_argument("--save-index", default="", help="write the rebuilt restart index (and the regions) to this .npz")
    a = ap.parse_args()
    L = hb.lib()
    assert L.hb_init() == 0
    n = a.mib << 20
    x = bench.synth_host(a.dataset, n, 0)
    codec_o, codec_d = (O.SNAPPY, hb.Snappy) if a.codec == "snappy" else (O.LZ4, hb.LZ4)
    if a.writer == "oracle":
        f = O.compress_frame(x, codec=codec_o, shuffle=a.shuffle, typesize=a.typesize)
    else:
        f = np.frombuffer(hb.Compress(x.tobytes(), codec_d, 5, a.shuffle, a.typesize, opts=0), np.uint8)
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
    print("rc", rc, "status", int(r[:4].view(np.int32)[0]), "flags", int(r[4:8].view(np.uint32)[0]), "bytes", int(r[8:16].view(np.uint64)[0]),
          "output == input:", bool(np.array_equal(d_out.cpu().numpy(), x.view(np.uint8).reshape(-1))))
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
    rep_us = regs[:, 15]
    if rep_us.any():                                 # a -DRG_DEBUG_TIMES build: microseconds every region spent in re-parses (k_rg_parse behind the first), kind 2 = full
        order = np.argsort(-rep_us.astype(np.int64))[:8]
        print("re-parsed regions", int((rep_us != 0).sum()), "total us", int(rep_us.sum()), "full re-parses", int((regs[:, 14] == 2).sum()))
        for i in order:
            print(f"  r{i}: {rep_us[i]} us kind {regs[i, 14]} | b {b[i]} entry {entry[i]} exit {exit_[i]} outlen {outlen[i]} entry0 {entry0[i]} exit0 {exit0[i]} ntrace {ntrace[i]} pad0 {regs[i, 9]}")
    # k_rg_index_fast's verdict per region (round 3): the done[] words live where k_rg_pmax's scratch was (RgLayout.pmax)
    al = lambda v: (v + 255) & ~255
    bound = n + n // 255 + 16
    nr_cap = min(max(bound // 8192 + 2, min(bound, 8 << 20) // 4096 + 2), RG_MAXREG)      # rg_max_regions (csrc/hb_lz4_region.h)
    off_pmax = base + al(64) + al(nr_cap * REG_BYTES)
    done = w[off_pmax: off_pmax + 4 * nreg].view(np.uint32)
    pad0 = regs[:, 9]
    live = outlen != 0
    print("fast index: regions with output", int(live.sum()), "all done", int(((done == 0xFFFFFFFF) & live).sum()), "head left", int(((done >= 16) & (done != 0xFFFFFFFF) & live).sum()),
          "left to the wave walk", int(((done < 16) & live).sum()), "| of those: record unusable", int(((done < 16) & live & ((pad0 == 0xFFFFFFFF) | (exit0 != exit_))).sum()),
          "out > 128 KiB", int(((done < 16) & live & (outlen > 131072)).sum()), "| reason codes (1 nb>32, 2 steps, 3 far, 4 exit, 5 ext, 6 bad)", np.bincount(done[(done < 16) & live], minlength=8).tolist())
    timed = np.nonzero((done >= 0x80000000) & (done != 0xFFFFFFFF))[0]
    if timed.size:                                   # a -DRG_DEBUG_TIMES build: microseconds the wave walk spent per region
        us = (done[timed] & 0x7FFFFFFF).astype(np.int64)
        order = np.argsort(-us)[:12]
        print("wave-walk regions", timed.size, "total us", int(us.sum()), "max", int(us.max()))
        for j in order:
            i = timed[j]
            print(f"  r{i}: {us[j]} us | stream {exit_[i] - entry[i]} B outlen {outlen[i]} ntrace {ntrace[i]}")
    slow = np.nonzero((done < 16) & live & (pad0 != 0xFFFFFFFF) & (exit0 == exit_) & (outlen <= 131072))[0][:6]
    for i in slow:
        print(f"  slow r{i}: b {b[i]} entry {entry[i]} exit {exit_[i]} outlen {outlen[i]} | entry0 {entry0[i]} exit0 {exit0[i]} outlen0 {outlen0[i]} pad0 {pad0[i]}")
    if a.save_index:
        off_idx = off_pmax + al(nr_cap * 4) + al(nr_cap * 256 * 8)
        nun = (n + 4095) // 4096
        np.savez(a.save_index, index=w[off_idx: off_idx + 32 + 16 * (nun + 1)].copy(), regs=regs.copy(), done=done.copy(), opos=(regs[:, 10].astype(np.uint64) | (regs[:, 11].astype(np.uint64) << 32)))
    # host model of k_rg_index_fast for the first slow regions: which boundary gives up, and why
    off_trace = off_pmax + al(nr_cap * 4)
    traces = w[off_trace: off_trace + nreg * 256 * 8].view(np.uint32).reshape(nreg, 256, 2)
    payload = f[16:]
    opos_all = regs[:, 10].astype(np.uint64) | (regs[:, 11].astype(np.uint64) << 32)
    for i in slow[:3]:
        start, exitp, ol, p0 = int(entry[i]), int(exit_[i]), int(outlen[i]), int(pad0[i])
        adj = (ol - int(outlen0[i])) & 0xFFFFFFFF
        opos = int(opos_all[i])
        U0 = (opos + 4095) & ~4095
        end = min(opos + ol, n)
        usable = [(int(t[0]), (int(t[1]) + adj) & 0xFFFFFFFF) for t in traces[i, 128:] if t[0] != 0xFFFFFFFF and t[0] >= p0 and t[0] > start and t[0] < exitp]
        print(f"   r{i}: opos {opos} boundaries {(end - U0 + 4095) // 4096} usable records {len(usable)} first {usable[:2]} last {usable[-1:]}")
        for U in range(U0, end, 4096):
            R = U - opos
            pp, cum, has = start, 0, R == 0
            for (ux, uc) in usable:
                if uc <= R:
                    pp, cum, has = ux, uc, True
            if not has:
                print(f"     U {U} R {R}: head"); continue
            q, steps, why = pp, 0, None
            while True:
                steps += 1
                if q >= exitp: why = "reached exit"; break
                tok = int(payload[q]); q += 1
                ll = tok >> 4
                if ll == 15:
                    while True:
                        xb = int(payload[q]); q += 1; ll += xb
                        if xb != 255: break
                if R == cum: why = f"AT_TOKEN after {steps} tokens"; break
                if R < cum + ll: why = f"in literals after {steps} tokens"; break
                q += ll
                ml = 0
                if q != len(payload):
                    q += 2; ml = (tok & 15) + 4
                    if (tok & 15) == 15:
                        while True:
                            xb = int(payload[q]); q += 1; ml += xb
                            if xb != 255: break
                if R < cum + ll + ml: why = f"INSIDE MATCH after {steps} tokens (cum {cum} ll {ll} ml {ml})"; break
                cum += ll + ml
            print(f"     U {U} R {R}: from {pp} (+{R - cum if why else 0} out) -> {why}")
    bad = np.nonzero(entry[1:] != exit_[:-1])[0][:8]
    for i in bad:
        print("  region", i + 1, "b", b[i + 1], "entry", entry[i + 1], "prev exit", exit_[i], "exit", exit_[i + 1], "needfull", needfull[i + 1])
    got = d_out.cpu().numpy()
    want = np.ascontiguousarray(x).view(np.uint8).reshape(-1)
    eq = bool(np.array_equal(got, want))
    print("equal to input:", eq, got.dtype, got.shape, want.dtype, want.shape)
    if not eq and got.shape == want.shape:
        d = np.nonzero(got != want)[0]
        print("  mismatching bytes", d.size, "first", d[:8], "last", d[-4:], "| got", got[d[:8]], "want", want[d[:8]])


if __name__ == "__main__":
    main()
