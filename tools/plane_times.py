#!/usr/bin/env python3
"""Per-byte-plane timing of the two fused LZ4 kernels on the headline workload.

HIPBLOSC_DEBUG_PLANE_MASK=<1 << j> (read once when the library loads; not part of the ABI) makes k_match_fused / k_dec_indexed
work on byte plane j of every element block only
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
    ap.add_argument("--child", default="", help="internal: path of the good frame; this process runs under HIPBLOSC_DEBUG_PLANE_MASK")
    a = ap.parse_args()
    n = a.mib << 20
    ts = a.typesize
    opts = hb.OPT_INDEX_TRAILER
    if not a.child:
        # The plane mask is read once from the environment when the library loads (it is not part of the ABI), so every mask gets
        # a process of its own.  The decoder needs a GOOD frame: this parent (all planes) writes it to a file for the children.
        import subprocess
        import tempfile
        assert hb.lib().hb_init() == 0
        d = bench.Dev(n, torch.device("cuda", 0))
        d.src.copy_(torch.from_numpy(bench.synth_host(a.dataset, n, 0)))
        d.compress(1, ts, opts)
        torch.cuda.synchronize()
        tmp = tempfile.NamedTemporaryFile(suffix=".frame", delete=False)
        d.frame.cpu().numpy().tofile(tmp.name)
        del d
        torch.cuda.empty_cache()
        rows = {}
        for name, mask in [("all", 0xFFFFFFFF)] + [(f"plane{j}", 1 << j) for j in range(ts)]:
            env = dict(os.environ, HIPBLOSC_DEBUG_PLANE_MASK=f"{mask:x}")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--mib", str(a.mib), "--dataset", a.dataset, "--typesize", str(ts),
                                "--reps", str(a.reps), "--child", tmp.name], env=env, capture_output=True, text=True, check=True)
            rows[name] = json.loads(r.stdout.strip().splitlines()[-1])
        os.unlink(tmp.name)
        out = {"workload": f"{a.mib} MiB {a.dataset}, Shuffle1 ts={ts} + LZ4, index trailer, 1 GPU", "reps": a.reps, "planes": rows}
        print(json.dumps(out))
        if a.out:
            os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
            json.dump(out, open(a.out, "w"), indent=1)
        return
    # ---- child: one mask (from the environment), the good frame from the file ----
    L = hb.lib()
    assert L.hb_init() == 0
    d = bench.Dev(n, torch.device("cuda", 0))
    d.src.copy_(torch.from_numpy(bench.synth_host(a.dataset, n, 0)))
    good = torch.from_numpy(np.fromfile(a.child, dtype=np.uint8))
    d.frame[: good.numel()].copy_(good)
    for _ in range(2):
        d.decompress()
    L.hb_profile_enable(1)
    for _ in range(a.reps):
        d.decompress()
        torch.cuda.synchronize()
    dec = bench.stage_times().get("k_dec_indexed", [])
    L.hb_profile_enable(0)
    for _ in range(2):
        d.compress(1, ts, opts)          # (the frame is garbage under a partial mask: timing only)
    L.hb_profile_enable(1)
    for _ in range(a.reps):
        d.compress(1, ts, opts)
        torch.cuda.synchronize()
    enc = bench.stage_times().get("k_match_fused", [])
    L.hb_profile_enable(0)
    print(json.dumps({"k_match_fused_ms": round(float(np.median(enc)), 4), "k_dec_indexed_ms": round(float(np.median(dec)), 4)}))


if __name__ == "__main__":
    main()
