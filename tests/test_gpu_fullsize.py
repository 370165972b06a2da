"""GPU parity at BASELINE.json's full sizes (configs 2, 3, 4: 1 GiB per frame), through the C ABI.

The oracle still finishes in seconds for the filters and the LZ4 DECODER at 1 GiB, so the checks are exact:
filter output == oracle bit for bit; every device frame decodes through the oracle decoder (the restated
reference `Decompress`) to the input; device decode == input.  Plus size-independent properties: round trip
idempotence and the header fields the reference pins (blosc_test.go:165-192).
"""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GIB = 1 << 30


def test_config2_shuffle_1gib_float32(hb, O):
    x = O.synth(O.D_F32, GIB // 4)
    want = O.filter(O.OP_SHUFFLE, x, 4)
    got = np.frombuffer(hb.shuffleBytes(x, 4), np.uint8)
    assert np.array_equal(got, want), "1 GiB shuffle differs from the oracle"
    back = np.frombuffer(hb.unshuffleBytes(got, 4), np.uint8)
    assert np.array_equal(back, x), "1 GiB unshuffle(shuffle(x)) != x"
    del want, got, back
    y = np.frombuffer(hb.bitShuffle(x, 4), np.uint8)
    assert np.array_equal(y, O.filter(O.OP_BITSHUFFLE, x, 4)), "1 GiB bitshuffle differs from the oracle"
    assert np.array_equal(np.frombuffer(hb.bitUnshuffle(y, 4), np.uint8), x)


def _frame_case(hb, O, x, shuffle, ts, header_prefix):
    f = hb.Compress(x, hb.LZ4, 5, shuffle, ts, opts=hb.OPT_INDEX_TRAILER)
    h = hb.ParseHeader(f[:16])
    assert f[:12] == header_prefix
    assert (h.NBytesOrig, h.BlockSize) == (x.size, x.size) and 16 < h.NBytesComp < x.size and not h.IsMemcpy()
    fa = np.frombuffer(f, np.uint8)
    assert np.array_equal(O.decompress_frame(fa), x), "oracle (reference decoder) cannot reproduce the input"
    y = hb.Decompress(f)
    assert hb.lib().hb_last_result_flags() & 1, "indexed decoder rejected its own index"
    assert np.array_equal(np.frombuffer(y, np.uint8), x), "device round trip differs"
    return h.NBytesComp / x.size


def test_config3_float64_shuffle8_lz4_1gib(hb, O):
    x = O.synth(O.D_F64, GIB // 8)
    ratio = _frame_case(hb, O, x, hb.Shuffle1, 8, bytes([2, 1, 1, 8]) + struct.pack("<II", GIB, GIB))
    assert ratio < 0.35            # ~0.26 for one 1 GiB block, ~0.265 for 4 KiB chunks


def test_config4_int32_bitshuffle_lz4_1gib(hb, O):
    x = O.synth(O.D_I32, GIB // 4, frame=3)          # frame 3 of the 8-frame job; one frame per GPU
    ratio = _frame_case(hb, O, x, hb.BitShuffle, 4, bytes([2, 1, 4, 4]) + struct.pack("<II", GIB, GIB))
    assert ratio < 0.70            # ~0.62


def test_headline_float32_shuffle4_lz4_1gib(hb, O):
    x = O.synth(O.D_F32, GIB // 4)
    ratio = _frame_case(hb, O, x, hb.Shuffle1, 4, bytes([2, 1, 1, 4]) + struct.pack("<II", GIB, GIB))
    assert ratio < 0.56            # ~0.52
