#!/usr/bin/env python3
"""tools/lab/ab.py — kernel variants of libhipblosc.so side by side on one GPU, in ONE gpurun call.

  python tools/lab/ab.py [--mib 1024] [--cases f32:1:4,f64:1:8,i32:2:4] [--reps 5] name1 name2 ...

`nameK` = a build made by tools/lab/build_variant.sh (go-blosc_amd/lib/libhipblosc_<name>.so; `-` = the product library).  Every
variant gets a process of its own (the library is loaded once per process: HIPBLOSC_LIB), runs each case (data set : shuffle : typesize)
as a device-resident round trip with the index trailer, checks that the decode reproduces the input bit for bit, and prints the stage
times (HIP events on the launch stream, median of `reps`), the ratio and the round-trip rate.  Lab tooling: no product code path.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(a):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd"))
    import numpy as np
    import torch
    import hipblosc as hb
    import bench
    L = hb.lib()
    assert L.hb_init() == 0
    n = a.mib << 20
    d = bench.Dev(n, torch.device("cuda", 0))
    rows = {}
    for case in a.cases.split(","):
        kind, shuffle, ts = case.split(":")
        shuffle, ts = int(shuffle), int(ts)
        d.src.copy_(torch.from_numpy(bench.synth_host(kind, n, 0)).view(torch.uint8))
        opts = 0 if a.no_trailer else hb.OPT_INDEX_TRAILER
        if a.nofusion:
            opts |= hb.OPT_NO_FUSION
        d.back.zero_()
        d.compress(shuffle, ts, opts)
        torch.cuda.synchronize()
        fb = d.results()[0]["bytes"] if a.no_trailer else None      # a frame without the trailer ends at NBytesComp (else the bytes behind it are taken for one)
        d.decompress(fb, foreign=a.no_trailer)
        torch.cuda.synchronize()
        rc, rd = d.results()
        ok = bool(torch.equal(d.back, d.src)) and rc["status"] == 0 and rd["status"] == 0
        for _ in range(2):
            d.compress(shuffle, ts, opts); d.decompress(fb, foreign=a.no_trailer)
        torch.cuda.synchronize()
        L.hb_profile_enable(1)
        for _ in range(a.reps):
            d.compress(shuffle, ts, opts); d.decompress(fb, foreign=a.no_trailer)
            torch.cuda.synchronize()
        st = bench.stage_times()
        L.hb_profile_enable(0)
        med = {k: round(float(np.median(v)), 4) for k, v in st.items()}
        # un-profiled wall clock of the round trip (events around the whole step)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            d.compress(shuffle, ts, opts); d.decompress(fb, foreign=a.no_trailer)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        rows[case] = dict(ok=ok, ratio=round(rc["bytes"] / n, 5), parallel=rd["flags"] & 1, ms_step=round(ms, 4), GBps=round(n / ms / 1e6, 1), stages=med)
    print(json.dumps(rows))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--cases", default="f32:1:4")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-trailer", action="store_true")
    ap.add_argument("--nofusion", action="store_true", help="HB_OPT_NO_FUSION: separate filter passes (A/B of the fused kernels)")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.child:
        return child(a)
    res = {}
    for name in a.names or ["-"]:
        env = dict(os.environ)
        if name != "-":
            env["HIPBLOSC_LIB"] = os.path.join(ROOT, "go-blosc_amd", "lib", f"libhipblosc_{name}.so")
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--mib", str(a.mib), "--cases", a.cases, "--reps", str(a.reps)]
        if a.no_trailer:
            cmd.append("--no-trailer")
        if a.nofusion:
            cmd.append("--nofusion")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        if r.returncode != 0:
            res[name] = {"error": (r.stderr or r.stdout)[-600:]}
        else:
            res[name] = json.loads(r.stdout.strip().splitlines()[-1])
        row = res[name]
        for case, v in row.items():
            if case == "error":
                print(f"{name:24s} ERROR {v}", flush=True)
                continue
            st = v["stages"]
            keys = ["filter_shuffle", "filter_bitshuffle", "k_match_fused", "k_match", "k_tiles", "k_scan", "k_stitch", "k_dec_plan", "k_dec_indexed", "k_dec_serial"]
            if a.no_trailer:
                keys = list(st.keys())                                   # the discovery's kernels too
            s = " ".join(f"{k[2:] if k.startswith('k_') else k}={st[k]:.3f}" for k in keys if k in st)
            print(f"{name:24s} {case:10s} ok={int(v['ok'])} par={v['parallel']} ratio={v['ratio']:.4f} step={v['ms_step']:.3f}ms {v['GBps']:.0f}GB/s | {s}", flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
    if any("error" in v for v in res.values()):
        raise SystemExit(1)                       # a variant crashed (e.g. a GPU fault): the call must read as failed


if __name__ == "__main__":
    main()
