#!/usr/bin/env python3
"""Generates tests/golden/filters_kat.json and frames_kat.json.

The reference (Go) cannot be run in this image (no toolchain, un-vendored modules) and holds no golden
vectors of its own, so these are NOT outputs of the reference.  They are:
  * `hand`  : vectors derived by hand from the scalar semantics of shuffle.go:59-72 / :176-216
              (SURVEY.md Appendix A), written out literally below;
  * `numpy` : outputs of the numpy twin in oracle/oracle.py (an independent restatement of the formulas),
              on small deterministic inputs covering the lengths the reference's tests use
              (shuffle_test.go:146-168, :284-316, :382-435; shuffle_amd64_test.go:133);
  * frames  : header bytes the reference's tests pin (blosc_test.go:165-192) and LZ4 streams written by hand
              from the block format, with the decoded bytes / the expected rejection.
Run from the repo root:  python tests/golden/make_kats.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as O  # noqa: E402

HAND = [
    dict(op="shuffle", ts=4, src=list(range(16)), dst=[0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]),
    dict(op="shuffle", ts=8, src=list(range(16)), dst=[0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15]),
    dict(op="shuffle", ts=4, src=list(range(10)), dst=[0, 4, 1, 5, 2, 6, 3, 7, 8, 9]),
    dict(op="bitshuffle", ts=4, src=list(range(32)),
         dst=[0, 0, 0, 15, 51, 85, 0, 0, 0, 0, 0, 15, 51, 85, 0, 255, 0, 0, 0, 15, 51, 85, 255, 0, 0, 0, 0, 15, 51, 85, 255, 255]),
    dict(op="bitshuffle", ts=4, src=list(range(35)),
         dst=[0, 0, 0, 15, 51, 85, 0, 0, 0, 0, 0, 15, 51, 85, 0, 255, 0, 0, 0, 15, 51, 85, 255, 0, 0, 0, 0, 15, 51, 85, 255, 255, 32, 33, 34]),
    dict(op="bitshuffle", ts=4, src=list(range(28)), dst=list(range(28))),
    dict(op="bitshuffle", ts=2, src=list(range(16)), dst=[0, 0, 0, 0, 15, 51, 85, 0, 0, 0, 0, 0, 15, 51, 85, 255]),
]
OPS = {"shuffle": O.OP_SHUFFLE, "unshuffle": O.OP_UNSHUFFLE, "bitshuffle": O.OP_BITSHUFFLE, "bitunshuffle": O.OP_BITUNSHUFFLE}
CASES = [(1003, 4), (28, 4), (35, 4), (12, 4), (127, 8), (37, 4), (97, 4), (13, 4), (103, 8), (10, 4), (64, 2), (96, 16),
         (50, 3), (7, 8), (33, 1), (256, 4), (255, 8)]

LZ4_STREAMS = [   # (hex stream, capacity, expected hex output or None for "must be rejected")
    ("", 10, ""), ("00", 10, ""), ("30616263", 100, "616263"), ("104101 00".replace(" ", ""), 100, "4141414141"),
    ("1f4101 00 05".replace(" ", "") + "50" + "4243444546", 100, "41" * 25 + "4243444546"),
    ("ffffffff", 100, None), ("10410000", 100, None), ("10410500", 100, None), ("404142", 100, None),
    ("10410100", 3, None), ("f0", 100, None),
]


def main():
    rng = np.random.default_rng(20261003)
    kats = [dict(kind="hand", **h) for h in HAND]
    for n, ts in CASES:
        src = rng.integers(0, 256, n, dtype=np.uint8)
        for name, op in OPS.items():
            kats.append(dict(kind="numpy", op=name, ts=ts, src=src.tolist(), dst=O.NP_FILTERS[op](src, ts).tolist()))
    with open(os.path.join(HERE, "filters_kat.json"), "w") as f:
        json.dump(kats, f, separators=(",", ":"))
    frames = dict(
        header_1000_lz4_shuffle4="02010104e8030000e8030000",     # blosc_test.go:165-192: first 12 bytes of Compress(1000 B, LZ4, 5, Shuffle1, 4)
        lz4_streams=[dict(stream=s, cap=c, out=o) for s, c, o in LZ4_STREAMS],
    )
    with open(os.path.join(HERE, "frames_kat.json"), "w") as f:
        json.dump(frames, f, indent=1)
    print(f"{len(kats)} filter vectors, {len(LZ4_STREAMS)} lz4 streams")


if __name__ == "__main__":
    main()
