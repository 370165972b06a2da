"""GPU parity for the parse and decode paths round 4 added: the matcher's run step (windows made of runs take the runs
from the wave-wide mask; at most seven runs per window, else the table), its INNER step body, the gated 20-byte extension, and the indexed
decoder's lean window parser (token positions only, fields parsed by the drain, chain by pointer doubling or scalar walk, rewind at
multi-byte length extensions).

The reference pins no compressed bytes: every frame must decode, bit for bit, through the oracle's restatement of
lz4.UncompressBlock (codec.go:77-84) and through liblz4, and through the device decoder itself -- with and without the restart index, with
and without the fused filters.  The inputs are built to sit on the new code's edges.
"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _liblz4():
    for p in ("/usr/lib/x86_64-linux-gnu/liblz4.so.1", "/opt/conda/lib/liblz4.so.1"):
        if os.path.exists(p):
            return ctypes.CDLL(p)
    return None


def _runs(rng, n, lo, hi, alphabet=256):
    """bytes made of runs with lengths in [lo, hi]"""
    out = np.empty(n + hi, np.uint8)
    i = 0
    while i < n:
        k = int(rng.integers(lo, hi + 1))
        out[i:i + k] = rng.integers(0, alphabet)
        i += k
    return out[:n]


def _inputs():
    rng = np.random.default_rng(404)
    n = 1 << 20
    cases = {}
    # run lengths around the run step's thresholds: < 4 (no run lane), 4..9 (eight or more runs per 64-byte window: table), 8..16 and 12..40
    # (run steps), runs longer than a window and longer than a chunk, runs that end exactly at window / chunk edges
    for lo, hi in ((1, 3), (2, 6), (4, 9), (7, 9), (8, 16), (12, 40), (60, 70), (64, 64), (100, 700), (4000, 4200), (4096, 4096)):
        cases[f"runs_{lo}_{hi}"] = _runs(rng, n, lo, hi)
    cases["runs_8_16_two_symbols"] = _runs(rng, n, 8, 16, alphabet=2)
    # a periodic pattern of short runs (what the low mantissa plane of a float ramp looks like): the table must keep it
    unit = _runs(rng, 1280, 7, 9)
    cases["periodic_short_runs"] = np.tile(unit, n // unit.size + 1)[:n]
    # runs with single noisy bytes at their edges (the headline's plane 2)
    x = _runs(rng, n, 8, 16)
    flips = rng.integers(0, n, n // 12)
    x[flips] ^= 1
    cases["noisy_runs"] = x
    # few-symbol noise (the headline's plane 0): matches of 4-7 bytes, none reaches 12 -> the gated extension never opens
    cases["four_symbols"] = (rng.integers(0, 4, n) << 6).astype(np.uint8)
    cases["two_symbols"] = (rng.integers(0, 2, n) << 7).astype(np.uint8)
    # long literal runs between long matches: literal-length extension bytes (15 .. 270+) and match-length extensions (19 .. 274+) in the tokens
    blocks = []
    motif = rng.integers(0, 256, 2000, dtype=np.uint8)
    while sum(b.size for b in blocks) < n:
        blocks.append(rng.integers(0, 256, int(rng.integers(10, 700)), dtype=np.uint8))
        k = int(rng.integers(4, 1200))
        s = int(rng.integers(0, motif.size - k)) if k < motif.size else 0
        blocks.append(motif[s:s + k])
    cases["long_literals_long_matches"] = np.concatenate(blocks)[:n]
    # bytes 0xFF everywhere a length extension could be mistaken for one
    y = rng.integers(0, 256, n, dtype=np.uint8)
    y[rng.integers(0, n, n // 3)] = 255
    y[n // 2: n // 2 + 70000] = 255
    cases["ff_heavy"] = y
    # a chunk's last steps: content that keeps matching right up to the end of every 4 KiB chunk (INNER / tail step boundary, end-of-block rules)
    z = np.tile(rng.integers(0, 256, 37, dtype=np.uint8), n // 37 + 1)[:n]
    cases["period37"] = z
    return cases


@pytest.mark.parametrize("shuffle,ts", [(0, 1), (1, 4), (1, 2), (2, 4)])
def test_new_parse_paths_round_trip_through_oracle_liblz4_and_device(hb, O, shuffle, ts):
    lz = _liblz4()
    for name, x in _inputs().items():
        raw = x.tobytes()
        for opts in (0, hb.OPT_INDEX_TRAILER):
            frame = hb.Compress(raw, hb.LZ4, 5, shuffle, ts, opts=opts)
            h = hb.GetInfo(frame)
            assert h.NBytesOrig == len(raw), name
            back = O.decompress_frame(np.frombuffer(frame, np.uint8)).tobytes()
            assert back == raw, (name, shuffle, ts, opts, "oracle decoder")
            assert hb.Decompress(frame) == raw, (name, shuffle, ts, opts, "device decoder")
            if lz is not None and not h.IsMemcpy():
                filt = O.filter({1: O.OP_SHUFFLE, 2: O.OP_BITSHUFFLE}[shuffle], x, ts).tobytes() if shuffle and ts > 1 else raw
                d = ctypes.create_string_buffer(len(raw))
                payload = frame[16:h.NBytesComp]
                r = lz.LZ4_decompress_safe(payload, d, len(payload), len(raw))
                assert r == len(raw) and d.raw == filt, (name, shuffle, ts, opts, "liblz4")


def test_run_step_keeps_periodic_short_runs_with_the_table(hb, O):
    """ENC_GATE_MAXRUNS: a window of eight or more short runs goes through the table, so a pattern of short runs that repeats
    inside the chunk compresses as a repeat (a few long matches), not as one sequence per run."""
    rng = np.random.default_rng(7)
    unit = _runs(rng, 1280, 7, 9)
    x = np.tile(unit, 1 << 10)[: 1 << 20]
    frame = hb.Compress(x.tobytes(), hb.LZ4, 5, hb.NoShuffle, 1)
    ratio = hb.GetInfo(frame).NBytesComp / x.size
    # one sequence per ~8-byte run would be ~0.4; the repeat at distance 1280 inside every 4 KiB chunk brings it far below that
    assert ratio < 0.2, ratio
    assert O.decompress_frame(np.frombuffer(frame, np.uint8)).tobytes() == x.tobytes()


def test_headline_ratio_stays_below_the_bound(hb, O):
    # ratio <= 0.522 on the headline data (16 MiB of it here; the 1 GiB frame is in test_gpu_fullsize.py)
    x = O.synth(O.D_F32, 1 << 22)
    frame = hb.Compress(x.tobytes(), hb.LZ4, 5, hb.Shuffle1, 4)
    assert hb.GetInfo(frame).NBytesComp / x.nbytes <= 0.522
