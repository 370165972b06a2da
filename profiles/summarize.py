#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the summaries kept under profiles/.

  summarize.py traffic <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json> "<workload text>"
  summarize.py counters <dir> <out.csv>          # per-kernel sums of every counter in a --pmc pass

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected in separate passes, the raw
unit is KiB... (32 B beats scaled by the tool), and gfx950 reports half the bytes of wide coalesced reads, so FETCH_SIZE
is doubled.  Values are averaged per dispatch of each kernel.
"""
import collections
import csv
import glob
import json
import sys


def load(d):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    return tot, cnt


def main():
    if sys.argv[1] == "traffic":
        fd, wd, out, text = sys.argv[2:6]
        ft, fc = load(fd)
        wt, wc = load(wd)
        kernels = {}
        for k in ft:
            if "FETCH_SIZE" not in ft[k] or "WRITE_SIZE" not in wt.get(k, {}):
                continue
            f = ft[k]["FETCH_SIZE"] / fc[(k, "FETCH_SIZE")]
            w = wt[k]["WRITE_SIZE"] / wc[(k, "WRITE_SIZE")]
            kernels[k] = {"fetch_bytes": int(f * 1024 * 2), "write_bytes": int(w * 1024),
                          "hbm_bytes": int(f * 1024 * 2 + w * 1024), "raw_FETCH_SIZE_KiB": f, "raw_WRITE_SIZE_KiB": w,
                          "launches": fc[(k, "FETCH_SIZE")]}
        json.dump({"workload": text,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (TCC slots), per-dispatch "
                             "average; raw unit = KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced "
                             "reads, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as is",
                   "kernels": kernels}, open(out, "w"), indent=1)
    else:
        d, out = sys.argv[2:4]
        tot, cnt = load(d)
        names = sorted({c for k in tot for c in tot[k]})
        with open(out, "w") as f:
            f.write("kernel,launches," + ",".join(names) + "\n")
            for k in tot:
                f.write(k + "," + str(max(cnt[(k, c)] for c in tot[k])) + "," + ",".join("%d" % tot[k].get(c, 0) for c in names) + "\n")


if __name__ == "__main__":
    main()
