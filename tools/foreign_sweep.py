#!/usr/bin/env python3
"""Decode time of frames WITHOUT the restart index over data sets and filters (profiles/r2z_foreign_sweep_1GiB.txt): for each of
f64 / int32 / ramp / random x byte shuffle / bit shuffle, a frame as the restated reference encoder writes it (one LZ4 block, 64 KiB
window: token discovery + symbolic decode) and one as this library writes it without the trailer (token discovery + rebuilt index),
through tools/region_debug.py; prints whether the parallel path took the frame (flags 1) and the stage times that show a derailed
discovery (k_rg_settle in tens of ms, k_dec_serial in hundreds).

  python tools/foreign_sweep.py [--mib 1024] [--datasets f64,i32,ramp,rand] [--shuffles 1,2]
"""
import argparse
import ast
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--datasets", default="f64,i32,ramp,rand")
    ap.add_argument("--shuffles", default="1,2")
    ap.add_argument("--codec", default="lz4", choices=["lz4", "snappy"], help="snappy: the oracle's Snappy encoder / this library's Snappy frames without the unit index")
    a = ap.parse_args()
    for ds in a.datasets.split(","):
        for sh in a.shuffles.split(","):
            for w in ("oracle", "device"):
                cmd = [sys.executable, os.path.join(ROOT, "tools", "region_debug.py"), "--mib", str(a.mib), "--reps", "4", "--dataset", ds,
                       "--shuffle", sh, "--typesize", "8" if ds == "f64" else "4", "--writer", w, "--codec", a.codec]
                t = subprocess.run(cmd, capture_output=True, text=True).stdout      # (a child per frame: a fresh workspace each time)
                # the best of the reps behind the first (a frame's decode right after another process has returned gigabytes of device memory has been seen to
                # take 100+ ms once, wall clock, with every stage at its usual time: the driver's business, not the decoder's)
                reps = [(float(a_), float(b_)) for a_, b_ in re.findall(r"rep [1-9]: ([0-9.]+) ms.*= ([0-9.]+) GB/s", t)]
                m = min(reps) if reps else None
                st = re.search(r"stage ms (\{.*\})", t)
                fl = re.search(r"flags (\d+)", t)
                s = ast.literal_eval(st.group(1)) if st else {}
                print(f"{ds} sh{sh} {w}: {m[0] if m else '?'} ms {m[1] if m else '?'} GB/s flags {fl.group(1) if fl else '?'} "
                      f"settle {s.get('k_rg_settle', s.get('k_snr_settle'))} serial {s.get('k_dec_serial', s.get('k_sn_dec_serial'))} sy_decode {s.get('k_sy_decode')} "
                      f"big {s.get('k_sy_big')} units {s.get('k_sn_dec_units')}", flush=True)


if __name__ == "__main__":
    main()
