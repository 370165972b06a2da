#!/usr/bin/env python3
"""Per-byte-plane timing of the two fused LZ4 kernels on the headline workload.

hb_debug_plane_mask(1 << j) makes k_match_fused / k_dec_indexed work on byte plane j of every element block only
(timing only: the other planes' work items are skipped, the frame is garbage while the mask is set), so the time of
one launch is the cost of that plane at the kernel's normal occupancy.  The planes' times add up to more than the
full launch because a single plane leaves the persistent grid's tail less balanced.

  python tools/plane_times.py [--mib 1024] [--dataset f32] [--out gpurun_out/planes.json]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd"))

import numpy as np
import torch

import hipblosc as hb
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--dataset", default="f32")
    ap.add_argument("--typesize", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    L = hb.lib()
    assert L.hb_init() == 0
    dev = torch.device("cuda", 0)
    n = a.mib << 20
    ts = a.typesize
    d = bench.Dev(n, dev)
    host = bench.synth_host(a.dataset, n, 0)
    d.src.copy_(torch.from_numpy(host))
    opts = hb.OPT_INDEX_TRAILER

    def run(mask):
        L.hb_debug_plane_mask(mask)
        # the decoder needs a GOOD frame: compress with all planes first, time the masked kernels after
        L.hb_debug_plane_mask(0xFFFFFFFF)
        d.compress(1, ts, opts)
        torch.cuda.synchronize()
        L.hb_debug_plane_mask(mask)
        for _ in range(2):
            d.decompress()
        L.hb_profile_enable(1)
        for _ in range(a.reps):
            d.decompress()
            torch.cuda.synchronize()
        dec = bench.stage_times().get("k_dec_indexed", [])
        L.hb_profile_enable(0)
        for _ in range(2):
            d.compress(1, ts, opts)
        L.hb_profile_enable(1)
        for _ in range(a.reps):
            d.compress(1, ts, opts)
            torch.cuda.synchronize()
        enc = bench.stage_times().get("k_match_fused", [])
        L.hb_profile_enable(0)
        L.hb_debug_plane_mask(0xFFFFFFFF)
        return float(np.median(enc)), float(np.median(dec))

    rows = {}
    full = run(0xFFFFFFFF)
    rows["all"] = {"k_match_fused_ms": round(full[0], 4), "k_dec_indexed_ms": round(full[1], 4)}
    for j in range(ts):
        e, c = run(1 << j)
        rows[f"plane{j}"] = {"k_match_fused_ms": round(e, 4), "k_dec_indexed_ms": round(c, 4)}
    # plane statistics of the frame (sequences are what both kernels pay for): tokens per plane from the index
    out = {"workload": f"{a.mib} MiB {a.dataset}, Shuffle1 ts={ts} + LZ4, index trailer, 1 GPU", "reps": a.reps, "planes": rows}
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
