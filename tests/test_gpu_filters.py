"""GPU parity: the four filters through the C ABI vs the CPU oracle, bit-exact.

Mirrors the reference's filter tests (shuffle_test.go:13-452, shuffle_amd64_test.go:16-115): typesizes
1/2/4/8/16 and odd ones, the remainder lengths it exercises (1003, 13, 103, 10, 28, 35, 12, 127, 37, 97,
100003), no-op cases, in-place wrappers, plus tile-boundary lengths of the vector kernels.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TYPESIZES = [1, 2, 3, 4, 5, 7, 8, 12, 16, 32, 255]
LENGTHS = [0, 1, 3, 10, 12, 13, 28, 35, 37, 64, 97, 103, 127, 1003, 4096, 4097, 8191, 16384, 32768 + 5,
           100003, 262144, 1 << 20, (1 << 20) + 4099]


def _data(n, seed=7):
    return np.random.default_rng(seed + n).integers(0, 256, n, dtype=np.uint8)


@pytest.mark.parametrize("ts", TYPESIZES)
def test_filters_match_oracle(hb, O, ts):
    fns = [hb.shuffleBytes, hb.unshuffleBytes, hb.bitShuffle, hb.bitUnshuffle]
    for n in LENGTHS:
        x = _data(n)
        for op, fn in enumerate(fns):
            got = np.frombuffer(fn(x.tobytes(), ts), np.uint8)
            want = O.filter(op, x, ts)
            assert got.size == n
            assert np.array_equal(got, want), f"op {op} ts {ts} n {n}: first diff at {np.flatnonzero(got != want)[:4]}"


@pytest.mark.parametrize("ts", [2, 4, 8, 16, 3])
def test_filter_round_trips(hb, ts):
    # shuffle_test.go:13-130, :146-168, :284-316
    for n in [1000, 1003, 4096 * ts, 4096 * ts + ts - 1, 300007]:
        x = _data(n, 11).tobytes()
        assert hb.unshuffleBytes(hb.shuffleBytes(x, ts), ts) == x
        assert hb.bitUnshuffle(hb.bitShuffle(x, ts), ts) == x


def test_kat_vectors(hb):
    # tests/golden/filters_kat.json restated inline for the device path
    a16 = bytes(range(16))
    assert list(hb.shuffleBytes(a16, 4)) == [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]
    assert list(hb.shuffleBytes(a16, 8)) == [0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15]
    assert list(hb.shuffleBytes(bytes(range(10)), 4)) == [0, 4, 1, 5, 2, 6, 3, 7, 8, 9]
    bs = [0, 0, 0, 15, 51, 85, 0, 0, 0, 0, 0, 15, 51, 85, 0, 255, 0, 0, 0, 15, 51, 85, 255, 0, 0, 0, 0, 15, 51, 85, 255, 255]
    assert list(hb.bitShuffle(bytes(range(32)), 4)) == bs
    assert list(hb.bitShuffle(bytes(range(35)), 4)) == bs + [32, 33, 34]
    assert list(hb.bitShuffle(bytes(range(28)), 4)) == list(range(28))
    assert list(hb.bitShuffle(a16, 2)) == [0, 0, 0, 0, 15, 51, 85, 0, 0, 0, 0, 0, 15, 51, 85, 255]


def test_simd_table_kats_on_device(hb):
    # the vectors derived from the reference's own SIMD tables (tests/golden/make_simd_kat.py) against the device kernels
    from test_oracle import check_simd_table_kats
    check_simd_table_kats(lambda b: hb.shuffleBytes(b, 4), lambda b: hb.unshuffleBytes(b, 4))


def test_bitshuffle_asm_kats_on_device(hb):
    # the vectors computed from the reference's own bitShuffleAVX2 / bitUnshuffleAVX2 instruction streams (tests/golden/make_bitshuffle_asm_kat.py)
    from test_oracle import check_bitshuffle_asm_kats
    check_bitshuffle_asm_kats(lambda b, ts: hb.bitShuffle(b, ts), lambda b, ts: hb.bitUnshuffle(b, ts))


def test_noop_cases(hb):
    # typeSize <= 1 or len < typeSize returns the input (shuffle.go:17-19, shuffle_test.go:318-380)
    x = bytes(range(7))
    for fn in (hb.shuffleBytes, hb.unshuffleBytes, hb.bitShuffle, hb.bitUnshuffle):
        assert fn(x, 1) == x and fn(x, 0) == x and fn(x, -3) == x and fn(x, 8) == x
        assert fn(b"", 4) == b""


def test_inplace_wrappers(hb, O):
    # ShuffleBuffer / UnshuffleBuffer, shuffle.go:298-323, shuffle_test.go:212-282 (unknown mode is a no-op)
    x = _data(1003, 3)
    for mode, op in [(hb.Shuffle1, 0), (hb.BitShuffle, 2)]:
        buf = bytearray(x.tobytes())
        hb.ShuffleBuffer(buf, 4, mode)
        assert bytes(buf) == O.filter(op, x, 4).tobytes()
        hb.UnshuffleBuffer(buf, 4, mode)
        assert bytes(buf) == x.tobytes()
    buf = bytearray(x.tobytes())
    hb.ShuffleBuffer(buf, 4, 99)
    hb.ShuffleBuffer(buf, 4, hb.NoShuffle)
    assert bytes(buf) == x.tobytes()


def test_filter_large_float32(hb, O):
    # 64 MiB of the headline data set (config 2 shape, scaled so the oracle finishes in seconds)
    x = O.synth(O.D_F32, 1 << 24)
    got = np.frombuffer(hb.shuffleBytes(x.tobytes(), 4), np.uint8)
    want = O.filter(0, x, 4)
    assert np.array_equal(got, want)
    assert hb.unshuffleBytes(got.tobytes(), 4) == x.tobytes()
