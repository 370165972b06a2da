"""GPU parity: LZ4 block codec and frame layer through the C ABI vs the CPU oracle.

The reference pins no compressed bytes: what it pins are round trips, decoder
rejections, header fields, error identities and a few inequalities (blosc_test.go, codec_test.go,
example_test.go, fuzz_test.go).  Those are restated here, with the oracle decoder standing in for the
reference's `Decompress` and liblz4 (when present) as an independent LZ4 block decoder.
"""
import ctypes
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _liblz4():
    for p in ("/usr/lib/x86_64-linux-gnu/liblz4.so.1", "/opt/conda/lib/liblz4.so.1"):
        if os.path.exists(p):
            return ctypes.CDLL(p)
    return None


def _lz4_safe_decode(lz, c, n):
    d = ctypes.create_string_buffer(max(n, 1))
    r = lz.LZ4_decompress_safe(bytes(c), d, len(c), n)
    return r, d.raw[:max(r, 0)]


def _patterns(O):
    rng = np.random.default_rng(5)
    out = {
        "mod256_10000": (np.arange(10000) % 256).astype(np.uint8),           # blosc_test.go:13-29
        "mod64_1000": (np.arange(1000) % 64).astype(np.uint8),               # example_test.go:13-33
        "zeros_10000": np.zeros(10000, np.uint8),                            # example_test.go:124-149
        "f32_ramp": O.synth(O.D_RAMP, 25000),                                # blosc_test.go:107-134 / config 1
        "f32_head": O.synth(O.D_F32, 1 << 16),
        "f64_head": O.synth(O.D_F64, 1 << 15),
        "i32_head": O.synth(O.D_I32, 1 << 16),
        "rand_f32": O.synth(O.D_RAND, 1 << 14),
        "random": rng.integers(0, 256, 50000, dtype=np.uint8),               # blosc_test.go:243-266
        "bytes256_100000": O.synth(O.D_BYTES256, 100000),                    # blosc_test.go:363-371
        "tiny_1": np.array([42], np.uint8),
        "tiny_5": np.arange(5, dtype=np.uint8),
        "tiny_13": np.arange(13, dtype=np.uint8),
        "run_4095": np.full(4095, 7, np.uint8),
        "run_4096": np.full(4096, 7, np.uint8),
        "run_4097": np.full(4097, 7, np.uint8),
        "run_70000": np.full(70000, 9, np.uint8),
        "period3": np.tile(np.array([1, 2, 3], np.uint8), 9000),
        "period100": np.tile(rng.integers(0, 256, 100, dtype=np.uint8), 700),
        "mixed": np.concatenate([rng.integers(0, 256, 9000, dtype=np.uint8), np.zeros(9000, np.uint8),
                                 rng.integers(0, 4, 9000, dtype=np.uint8), np.arange(9000, dtype=np.uint32).view(np.uint8)]),
    }
    return out


def test_lz4_block_decodes_with_oracle_and_liblz4(hb, O):
    codec = hb.HipLZ4Codec()
    lz = _liblz4()
    for name, x in _patterns(O).items():
        c = codec.Compress(x.tobytes(), 5)
        assert 0 < len(c) <= hb.lib().hb_lz4_bound(x.size), name
        back = O.lz4_decompress(np.frombuffer(c, np.uint8), x.size)
        assert np.array_equal(back, x), f"{name}: oracle decode of the device block differs"
        if lz is not None:
            r, d = _lz4_safe_decode(lz, c, x.size)
            assert r == x.size and d == x.tobytes(), f"{name}: liblz4 rejects/differs ({r})"
        assert codec.Decompress(c, x.size) == x.tobytes(), f"{name}: device decode of the device block differs"


def test_lz4_device_decodes_oracle_blocks(hb, O):
    # streams shaped like the reference's (no index): serial single-wavefront decoder
    codec = hb.HipLZ4Codec()
    for name, x in _patterns(O).items():
        c = O.lz4_compress(x)
        assert codec.Decompress(c.tobytes(), x.size) == x.tobytes(), name


def test_lz4_encoder_is_deterministic(hb, O):
    codec = hb.HipLZ4Codec()
    x = O.filter(0, O.synth(O.D_F32, 1 << 18), 4).tobytes()
    a = codec.Compress(x, 5)
    for _ in range(3):
        assert codec.Compress(x, 5) == a


def test_lz4_decoder_rejections(hb):
    # codec_test.go:203-233, :276-295: FF FF FF FF must fail; plus offset 0, offset beyond start,
    # truncated stream, output overflow
    codec = hb.HipLZ4Codec()
    bad = [
        (b"\xff\xff\xff\xff", 100),
        (b"\x10A\x00\x00", 100),                 # offset 0
        (b"\x10A\x05\x00", 100),                 # offset 5 with 1 byte produced
        (b"\x40AB", 100),                        # 4 literals announced, 2 present
        (b"\x10A\x01\x00", 3),                   # match of 4 overflows a 3-byte output
        (b"\xf0", 100),                          # truncated length extension
    ]
    for stream, cap in bad:
        with pytest.raises(hb.ErrDecompressionFailed):
            codec.Decompress(stream, cap)
    assert codec.Decompress(b"", 10) == b""       # UncompressBlock: empty src -> 0, nil
    assert codec.Decompress(b"\x00", 10) == b""
    assert codec.Decompress(b"\x10A\x01\x00", 100) == b"AAAAA"     # stream may end right after a match
    assert codec.Decompress(b"\x30abc", 100) == b"abc"            # short output is not a codec error (codec.go:83)


@pytest.mark.parametrize("shuffle,ts", [(0, 1), (1, 4), (2, 4), (1, 8), (2, 8), (1, 2), (1, 16), (2, 2), (1, 3)])
def test_frame_round_trips(hb, O, shuffle, ts):
    # blosc_test.go:13-163, :290-312; every device frame must decode through the reference decoder (oracle)
    for name, x in _patterns(O).items():
        for opts in (0, hb.OPT_INDEX_TRAILER):
            f = hb.Compress(x.tobytes(), hb.LZ4, 5, shuffle, ts, opts=opts)
            h = hb.ParseHeader(f)
            assert (h.Version, h.VersionLZ, h.TypeSize, h.NBytesOrig, h.BlockSize) == (2, hb.LZ4, ts, x.size, x.size), name
            assert h.ShuffleMode() == shuffle
            if opts == 0:
                assert h.NBytesComp == len(f)
            else:
                assert h.NBytesComp <= len(f)
            assert np.array_equal(O.decompress_frame(np.frombuffer(f, np.uint8)), x), f"{name}: oracle decode differs"
            assert hb.Decompress(f) == x.tobytes(), f"{name}: device decode differs"
            assert hb.GetDecompressedSize(f) == x.size


def test_device_decodes_oracle_frames(hb, O):
    for name, x in _patterns(O).items():
        for shuffle, ts in [(0, 1), (1, 4), (2, 4), (1, 8)]:
            for policy in (0, O.POLICY_REFERENCE_MEMCPY):
                f = O.compress_frame(x, shuffle=shuffle, typesize=ts, policy=policy)
                want = O.decompress_frame(f)                  # bug-compatible for memcpy+filter frames (blosc.go:398-426)
                assert hb.Decompress(f.tobytes()) == want.tobytes(), (name, shuffle, ts, policy)


def test_header_and_inequalities(hb, O):
    x = (np.arange(1000) % 64).astype(np.uint8).tobytes()
    f = hb.Compress(x, hb.LZ4, 5, hb.Shuffle1, 4)
    assert f[:12] == bytes([2, 1, 1, 4]) + struct.pack("<II", 1000, 1000)       # blosc_test.go:165-192, Appendix A
    assert len(f) < len(x)                                                       # example_test.go:29-32
    z = hb.Compress(bytes(10000), hb.LZ4, 5, hb.NoShuffle, 1)
    assert hb.GetInfo(z).NBytesComp < 10000                                      # example_test.go:141-148
    pat = np.tile(np.array([0x12, 0x34, 0x56, 0x78], np.uint8), 1000).tobytes()  # example_test.go:209-230 shape
    pat = (np.arange(1000, dtype=np.uint32) * 1).view(np.uint8).tobytes()
    assert len(hb.Compress(pat, hb.LZ4, 5, hb.Shuffle1, 4)) < len(hb.Compress(pat, hb.LZ4, 5, hb.NoShuffle, 4))
    # memcpy frames: random bytes do not compress (blosc_test.go:243-266)
    r = np.random.default_rng(1).integers(0, 256, 20000, dtype=np.uint8).tobytes()
    fm = hb.Compress(r, hb.LZ4, 5, hb.NoShuffle, 1)
    hm = hb.GetInfo(fm)
    assert hm.IsMemcpy() and hm.NBytesComp == 16 + len(r) and fm[16:] == r
    assert hb.Decompress(fm) == r


def test_memcpy_policy(hb, O):
    # incompressible input + filter.  Default: filtered bytes are stored, so the
    # reference decoder reproduces the input; OPT_REFERENCE_MEMCPY restates blosc.go:342-345 bit for bit.
    r = np.random.default_rng(2).integers(0, 256, 40000, dtype=np.uint8)
    f = hb.Compress(r.tobytes(), hb.LZ4, 5, hb.Shuffle1, 4)
    assert hb.GetInfo(f).IsMemcpy()
    assert np.array_equal(O.decompress_frame(np.frombuffer(f, np.uint8)), r)
    assert f == O.compress_frame(r, shuffle=1, typesize=4).tobytes()
    g = hb.Compress(r.tobytes(), hb.LZ4, 5, hb.Shuffle1, 4, opts=hb.OPT_REFERENCE_MEMCPY)
    assert g == O.compress_frame(r, shuffle=1, typesize=4, policy=O.POLICY_REFERENCE_MEMCPY).tobytes()
    assert g[16:] == r.tobytes()
    assert hb.Decompress(g) == O.decompress_frame(np.frombuffer(g, np.uint8)).tobytes() != r.tobytes()


def test_errors(hb, O):
    with pytest.raises(hb.ErrInvalidData):                      # blosc_test.go:211-215
        hb.Compress(b"", hb.LZ4, 5, hb.NoShuffle, 1)
    with pytest.raises(hb.ErrInvalidHeader):                    # blosc_test.go:217-225
        hb.Decompress(b"\x02\x01\x00")
    x = bytes(range(200)) * 5
    f = bytearray(hb.Compress(x, hb.LZ4, 5, hb.NoShuffle, 1))
    bad = bytearray(f); bad[0] = 99
    with pytest.raises(hb.ErrInvalidVersion):                   # blosc_test.go:518-542
        hb.Decompress(bytes(bad))
    bad = bytearray(f); bad[1] = 77
    with pytest.raises(hb.ErrInvalidCodec):                     # codec_test.go:37-58
        hb.Decompress(bytes(bad))
    with pytest.raises(hb.ErrInvalidCodec):
        hb.Compress(x, 77, 5, hb.NoShuffle, 1)
    bad = bytearray(f); bad[12:16] = struct.pack("<I", len(f) + 10)
    with pytest.raises(hb.ErrInvalidData):                      # blosc_test.go:544-558
        hb.Decompress(bytes(bad))
    bad = bytearray(f); bad[12:16] = struct.pack("<I", 8)
    with pytest.raises(hb.ErrInvalidData):
        hb.Decompress(bytes(bad))
    bad = bytearray(f); bad[4:8] = struct.pack("<I", 2000)      # NBytesOrig 1000 -> 2000, codec_test.go:60-79
    with pytest.raises(hb.ErrSizeMismatch):
        hb.Decompress(bytes(bad))
    bad = bytearray(f); bad[4:8] = struct.pack("<I", 500)       # output overflow -> decode error
    with pytest.raises(hb.ErrDecompressionFailed):
        hb.Decompress(bytes(bad))
    bad = bytearray(f)
    for i in range(16, len(bad)):
        bad[i] ^= 0xFF                                          # blosc_test.go:593-611
    with pytest.raises(hb.BloscError):
        hb.Decompress(bytes(bad))
    assert hb.Decompress(bytes(f) + b"trailing garbage") == x   # bytes after cbytes are ignored, blosc.go:385-393


def test_option_clamping(hb):
    # blosc_test.go:613-655: level -5 / 100 and typeSize -1 / 0 are accepted
    x = bytes(range(256)) * 8
    for level in (-5, 0, 100):
        assert hb.Decompress(hb.Compress(x, hb.LZ4, level, hb.Shuffle1, 4)) == x
    for ts in (-1, 0):
        f = hb.Compress(x, hb.LZ4, 5, hb.Shuffle1, ts)
        assert hb.GetInfo(f).TypeSize == 1
        assert hb.Decompress(f) == x
    o = hb.DefaultOptions()
    assert (o.Codec, o.Level, o.Shuffle, o.TypeSize, o.BlockSize) == (hb.LZ4, 5, hb.Shuffle1, 4, 0)
    o.BlockSize, o.NumThreads = 12345, 8                        # accepted, ignored (blosc.go:232-233)
    assert hb.Decompress(hb.CompressWithOptions(x, o)) == x


def test_typesize_override_and_wide_typesize(hb, O):
    x = O.synth(O.D_F32, 5000)
    f = hb.Compress(x.tobytes(), hb.LZ4, 5, hb.Shuffle1, 4)
    assert hb.DecompressWithSize(f, 4) == x.tobytes()
    assert hb.DecompressWithSize(f, 2) == O.decompress_frame(np.frombuffer(f, np.uint8), 2).tobytes()
    # typeSize 256 is truncated to 0 in the header (blosc.go:362) -> no unshuffle on decode (Appendix D)
    y = np.random.default_rng(9).integers(0, 3, 256 * 40, dtype=np.uint8)
    g = hb.Compress(y.tobytes(), hb.LZ4, 5, hb.Shuffle1, 256)
    assert hb.GetInfo(g).TypeSize == 0
    assert hb.Decompress(g) == O.decompress_frame(np.frombuffer(g, np.uint8)).tobytes()


def test_tampered_index_falls_back(hb, O):
    # the restart index is never trusted: corrupting it must not change the decoded bytes
    x = O.synth(O.D_F32, 1 << 16).tobytes()
    f = bytearray(hb.Compress(x, hb.LZ4, 5, hb.Shuffle1, 4, opts=hb.OPT_INDEX_TRAILER))
    cb = hb.GetInfo(bytes(f)).NBytesComp
    ioff = (cb + 7) & ~7
    assert len(f) > ioff + 32
    for pos, val in [(ioff + 32 + 16 * 3, 0x55), (ioff + 32 + 16 * 5 + 8, 0x01), (ioff + 32 + 16 * 7 + 4, 0x10), (ioff + 8, 0x02)]:
        g = bytearray(f)
        g[pos] ^= val
        assert hb.Decompress(bytes(g)) == x


def test_multi_tile_frames_use_the_index(hb, O):
    # several scan tiles (256 chunks each), literal runs of MiB (255-extension of tens of KiB), ragged ends;
    # the indexed decoder must accept its own index (flag bit0), not silently fall back to the serial one
    rng = np.random.default_rng(11)
    cases = {
        "f32_5MiB": (O.synth(O.D_F32, (5 << 20) // 4 + 3), 1, 4),
        "f64_3MiB": (O.synth(O.D_F64, (3 << 20) // 8 + 1), 1, 8),
        "i32_bitshuffle_3MiB": (O.synth(O.D_I32, (3 << 20) // 4), 2, 4),
        "rand_then_zeros": (np.concatenate([rng.integers(0, 256, (2 << 20) + 77, dtype=np.uint8), np.zeros(1 << 20, np.uint8),
                                            rng.integers(0, 256, 123457, dtype=np.uint8), np.full(70001, 3, np.uint8)]), 0, 1),
        "ramp_2MiB": (O.synth(O.D_RAMP, (2 << 20) // 4 + 5), 1, 4),
    }
    for name, (x, shuffle, ts) in cases.items():
        f = hb.Compress(x.tobytes(), hb.LZ4, 5, shuffle, ts, opts=hb.OPT_INDEX_TRAILER)
        assert np.array_equal(O.decompress_frame(np.frombuffer(f, np.uint8)), x), f"{name}: oracle decode differs"
        assert hb.Decompress(f) == x.tobytes(), f"{name}: device decode differs"
        if not hb.GetInfo(f).IsMemcpy():
            assert hb.lib().hb_last_result_flags() & 1, f"{name}: indexed decoder rejected its own index"
        cb = hb.GetInfo(f).NBytesComp
        assert hb.Decompress(f[:cb]) == x.tobytes(), f"{name}: decode with the index cut off differs"
        # without the trailer: payloads from 256 KiB up (from 16 KiB when they decode to 2 MiB and more) get their index rebuilt on the
        # device (csrc/hb_lz4_region.hip), smaller ones are decoded by the single wavefront
        rebuilt = (not hb.GetInfo(f).IsMemcpy()) and hb.indexless_parallel(cb - 16, x.nbytes)
        assert bool(hb.lib().hb_last_result_flags() & 1) == rebuilt, name


@pytest.mark.parametrize("ts", [2, 4, 8])
def test_fused_shuffle_equals_separate_filter_pass(hb, O, ts):
    # byte shuffle on whole blocks of 4096 elements is fused into the matcher (no filtered buffer in HBM);
    # the frame must be byte-identical to the one built with the filter as its own pass
    rng = np.random.default_rng(ts)
    blk = 4096 * ts
    cases = {
        "f32": O.synth(O.D_F32, 3 * blk // 4 * 4)[: 12 * blk],
        "f64": O.synth(O.D_F64, 5 * blk // 8 * 8)[: 5 * blk],
        "noise": rng.integers(0, 256, 7 * blk, dtype=np.uint8),                       # -> memcpy frame
        "mixed": np.concatenate([rng.integers(0, 256, 2 * blk, dtype=np.uint8), np.zeros(3 * blk, np.uint8),
                                 O.synth(O.D_I32, blk // 4 * 4)[: 4 * blk]]),
        "one_block": O.synth(O.D_RAMP, blk // 4),
    }
    for name, x in cases.items():
        assert x.size % blk == 0
        for base in (0, hb.OPT_INDEX_TRAILER, hb.OPT_REFERENCE_MEMCPY):
            fused = hb.Compress(x.tobytes(), hb.LZ4, 5, hb.Shuffle1, ts, opts=base)
            plain = hb.Compress(x.tobytes(), hb.LZ4, 5, hb.Shuffle1, ts, opts=base | hb.OPT_NO_FUSION)
            if not (base & hb.OPT_REFERENCE_MEMCPY):
                for which, fr in (("fused", fused), ("two-pass", plain)):               # each against the oracle, not against each other
                    assert np.array_equal(O.decompress_frame(np.frombuffer(fr, np.uint8)), x), (name, which)
                assert hb.Decompress(fused) == x.tobytes(), name
            else:                                                                       # blosc.go:342-345 bit for bit: header + raw input
                for fr in (fused, plain):
                    h = hb.GetInfo(fr)
                    assert (not h.IsMemcpy()) or fr[16:16 + x.size] == x.tobytes(), name
            assert fused == plain, f"{name} opts={base}: frames must not depend on where the filter ran"


def test_fused_bitshuffle_equals_separate_filter_pass(hb, O):
    # bitshuffle (typesize 4) is an in-place transform of 32-byte windows: fused into the matcher and into the
    # indexed decoder when the frame holds only whole windows; the serial fallback goes through the staging buffer
    rng = np.random.default_rng(44)
    cases = {
        "i32": O.synth(O.D_I32, 40000),                                               # 160000 B, whole windows
        "f32": O.synth(O.D_F32, 3 * 4096 + 8),
        "noise": rng.integers(0, 256, 64 * 1024, dtype=np.uint8),                     # -> memcpy frame
        "ragged": O.synth(O.D_I32, 10001),                                            # 40004 B: not whole windows -> two-pass path
        "tiny": O.synth(O.D_I32, 8),
    }
    for name, x in cases.items():
        for base in (0, hb.OPT_INDEX_TRAILER, hb.OPT_REFERENCE_MEMCPY):
            fused = hb.Compress(x.tobytes(), hb.LZ4, 5, hb.BitShuffle, 4, opts=base)
            plain = hb.Compress(x.tobytes(), hb.LZ4, 5, hb.BitShuffle, 4, opts=base | hb.OPT_NO_FUSION)
            if not (base & hb.OPT_REFERENCE_MEMCPY):
                for which, fr in (("fused", fused), ("two-pass", plain)):               # each against the oracle, not against each other
                    assert np.array_equal(O.decompress_frame(np.frombuffer(fr, np.uint8)), x), (name, which)
            want = O.decompress_frame(np.frombuffer(fused, np.uint8)).tobytes()
            assert hb.Decompress(fused) == want, name
            assert fused == plain, f"{name} opts={base}: frames must not depend on where the filter ran"
    # tampered index on a bitshuffled frame: indexed decoder refuses, serial decoder + gated un-filter take over
    x = O.synth(O.D_I32, 1 << 16).tobytes()
    f = bytearray(hb.Compress(x, hb.LZ4, 5, hb.BitShuffle, 4, opts=hb.OPT_INDEX_TRAILER))
    assert hb.Decompress(bytes(f)) == x and hb.lib().hb_last_result_flags() & 1
    ioff = (hb.GetInfo(bytes(f)).NBytesComp + 7) & ~7
    f[ioff + 32 + 16 * 9 + 4] ^= 0x20
    assert hb.Decompress(bytes(f)) == x and not (hb.lib().hb_last_result_flags() & 1)


def test_concurrent_calls_from_many_threads(hb, O):
    # "All functions ... safe for concurrent use" (blosc.go:37-39): the C ABI is called from 6 OS threads at once
    import threading
    xs = [O.synth(O.D_F32, 60000 + 1000 * k, frame=k).tobytes() for k in range(6)]
    out, errs = [None] * 6, []

    def work(k):
        try:
            for _ in range(4):
                f = hb.Compress(xs[k], hb.LZ4, 5, hb.Shuffle1 if k % 2 else hb.BitShuffle, 4, opts=hb.OPT_INDEX_TRAILER)
                assert hb.Decompress(f) == xs[k]
                assert hb.unshuffleBytes(hb.shuffleBytes(xs[k], 4), 4) == xs[k]
            out[k] = f
        except Exception as e:          # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for k in range(6):
        assert np.array_equal(O.decompress_frame(np.frombuffer(out[k], np.uint8)), np.frombuffer(xs[k], np.uint8))


def test_config5_device_shuffle_host_zstd(hb, O):
    # BASELINE.json config 5 (scaled): Shuffle1 + ZSTD level 3; filter on the device, zstd on host threads.
    # Oracle for the codec: libzstd itself (ZSTD_decompress decodes concatenated frames like DecodeAll), then the
    # restated unshuffle.
    zs = None
    for p in ("/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/opt/conda/lib/libzstd.so.1"):
        if os.path.exists(p):
            zs = ctypes.CDLL(p)
            break
    if zs is None:
        pytest.skip("libzstd not in this image")
    zs.ZSTD_decompress.restype = ctypes.c_size_t
    zs.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    for x, shuffle, ts in [(O.synth(O.D_F32, (40 << 20) // 4 + 3), hb.Shuffle1, 4),          # 3 slices, ragged
                           (O.synth(O.D_I32, 100000), hb.BitShuffle, 4),
                           (np.random.default_rng(3).integers(0, 256, 300000, dtype=np.uint8), hb.Shuffle1, 4),   # memcpy
                           (O.synth(O.D_F64, 50000), hb.NoShuffle, 8)]:
        f = hb.Compress(x.tobytes(), hb.ZSTD, 3, shuffle, ts)
        h = hb.ParseHeader(f)
        assert (h.Version, h.VersionLZ, h.TypeSize, h.NBytesOrig, h.NBytesComp) == (2, hb.ZSTD, ts, x.size, len(f))
        payload = f[16:]
        if h.IsMemcpy():
            filt = np.frombuffer(payload, np.uint8)
        else:
            out = ctypes.create_string_buffer(x.size)
            r = zs.ZSTD_decompress(out, x.size, payload, len(payload))
            assert r == x.size
            filt = np.frombuffer(out.raw, np.uint8)
        op = {hb.Shuffle1: O.OP_UNSHUFFLE, hb.BitShuffle: O.OP_BITUNSHUFFLE}.get(shuffle)
        back = filt if op is None or ts <= 1 else O.filter(op, filt, ts)
        assert np.array_equal(back, x), "libzstd + restated unshuffle cannot reproduce the input"
        assert hb.Decompress(f) == x.tobytes()
    with pytest.raises(hb.ErrDecompressionFailed):
        g = bytearray(hb.Compress(O.synth(O.D_F32, 20000).tobytes(), hb.ZSTD, 3, hb.Shuffle1, 4))
        g[30] ^= 0xFF; g[40] ^= 0xFF
        hb.Decompress(bytes(g))


def test_frame_queue_matches_one_call_api(hb, O):
    # hb_queue_* (SURVEY.md §8 f1): frames in flight on their own streams must give exactly what the one-call API
    # gives (same frames byte for byte, same decoded bytes), for mixed shapes, in and out of order, with slot reuse.
    import ctypes
    sizes = [4096 * 4 * 5, 100000, 1 << 20, 12, 333333, 4096 * 4 * 64, 7, 65536 * 4]
    xs = [O.synth(O.D_F32, (s + 3) // 4, frame=k).tobytes()[:s] for k, s in enumerate(sizes)]
    mx = max(sizes)
    cap = hb.lib().hb_frame_bound(mx)
    q = hb.FrameQueue(mx, depth=3)
    pin_in = [hb.PinnedBuffer(mx) for _ in sizes]
    pin_out = [hb.PinnedBuffer(cap) for _ in sizes]
    modes = [(hb.Shuffle1, 4), (hb.BitShuffle, 4), (hb.NoShuffle, 1), (hb.Shuffle1, 8)]
    tickets = []
    for k, x in enumerate(xs):
        ctypes.memmove(pin_in[k].ptr, x, len(x))
        sh, ts = modes[k % 4]
        tickets.append(q.compress(pin_in[k].ptr, len(x), pin_out[k].ptr, cap, hb.LZ4, 5, sh, ts, hb.OPT_INDEX_TRAILER))
        if k >= 2:                                  # keep 3 in flight; wait for the oldest
            j = k - 2
            nb = q.wait(tickets[j])
            sh, ts = modes[j % 4]
            assert bytes(pin_out[j].view[:nb]) == hb.Compress(xs[j], hb.LZ4, 5, sh, ts, opts=hb.OPT_INDEX_TRAILER)
            # ... and, not only equal to the one-call API: the ORACLE decoder turns the queued frame back into the input
            assert O.decompress_frame(np.frombuffer(bytes(pin_out[j].view[:nb]), np.uint8)).tobytes() == xs[j]
    frames = {}
    for j in (len(xs) - 1, len(xs) - 2):            # out of order
        nb = q.wait(tickets[j])
        sh, ts = modes[j % 4]
        assert bytes(pin_out[j].view[:nb]) == hb.Compress(xs[j], hb.LZ4, 5, sh, ts, opts=hb.OPT_INDEX_TRAILER)
        assert O.decompress_frame(np.frombuffer(bytes(pin_out[j].view[:nb]), np.uint8)).tobytes() == xs[j]
    with pytest.raises(hb.HipBloscError):
        q.wait(tickets[0])                          # a ticket is waited for once
    # decode through the queue: frames from pin_out[k] back into pin_in[k]
    for k, x in enumerate(xs):
        sh, ts = modes[k % 4]
        frames[k] = hb.Compress(x, hb.LZ4, 5, sh, ts, opts=hb.OPT_INDEX_TRAILER)
        ctypes.memmove(pin_out[k].ptr, frames[k], len(frames[k]))
    tk = [q.decompress(pin_out[k].ptr, len(frames[k]), pin_in[k].ptr, mx) for k in range(3)]
    for k in range(3, len(xs)):
        assert q.wait(tk[k - 3]) == len(xs[k - 3]) and bytes(pin_in[k - 3].view[:len(xs[k - 3])]) == xs[k - 3]
        tk.append(q.decompress(pin_out[k].ptr, len(frames[k]), pin_in[k].ptr, mx))
    for k in range(len(xs) - 3, len(xs)):
        assert q.wait(tk[k]) == len(xs[k]) and bytes(pin_in[k].view[:len(xs[k])]) == xs[k]
    # frames written by the ORACLE (no index trailer: the reference's own output shape) decode through the queue too
    for k in (0, 1, 3):
        sh, ts = modes[k % 4]
        of = O.compress_frame(np.frombuffer(xs[k], np.uint8), shuffle=sh, typesize=ts).tobytes()
        ctypes.memmove(pin_out[k].ptr, of, len(of))
        t = q.decompress(pin_out[k].ptr, len(of), pin_in[k].ptr, mx)
        assert q.wait(t) == len(xs[k]) and bytes(pin_in[k].view[:len(xs[k])]) == xs[k]
    # a ticket whose slot was re-used before it was waited for keeps its result (hipblosc.h: newest 4 * depth kept)
    late = []
    for k in range(5):                              # depth 3: tickets 0 and 1 of this batch lose their slots
        ctypes.memmove(pin_in[k].ptr, xs[k], len(xs[k]))
        sh, ts = modes[k % 4]
        late.append(q.compress(pin_in[k].ptr, len(xs[k]), pin_out[k].ptr, cap, hb.LZ4, 5, sh, ts, 0))
    for k in (0, 1, 4, 3, 2):
        sh, ts = modes[k % 4]
        nb = q.wait(late[k])
        assert bytes(pin_out[k].view[:nb]) == hb.Compress(xs[k], hb.LZ4, 5, sh, ts, opts=0), k
    with pytest.raises(hb.HipBloscError):
        q.wait(late[0])                             # ... once
    # errors keep the reference's identities
    with pytest.raises(hb.ErrInvalidData):
        q.compress(pin_in[0].ptr, 0, pin_out[0].ptr, cap)                       # blosc.go:269-271
    with pytest.raises(hb.ErrInvalidHeader):
        q.decompress(pin_out[0].ptr, 8, pin_in[0].ptr, mx)                      # blosc.go:297-299
    bad = bytearray(frames[1]); bad[20:40] = bytes(20)                          # corrupt payload + stale index
    ctypes.memmove(pin_out[1].ptr, bytes(bad), len(bad))
    t = q.decompress(pin_out[1].ptr, len(bad), pin_in[1].ptr, mx)
    try:
        q.wait(t)                                   # may decode to other bytes; must not fault (fuzz_test.go:135-159)
    except (hb.ErrDecompressionFailed, hb.ErrSizeMismatch, hb.ErrInvalidData):
        pass
    q.close()


@pytest.mark.parametrize("ts", [2, 4, 8, 16])
def test_fused_unshuffle_in_decoder_matches_the_oracle(hb, O, ts):
    # byte-shuffled frames made of whole planes of whole 4 KiB chunks are un-shuffled by the indexed decoder itself
    # (byte-strided stores); the result must equal the oracle's decode of the same frame (= the input)
    n = ts * 4096 * 9
    x = O.synth(O.D_F64 if ts == 8 else O.D_F32, n // (8 if ts == 8 else 4), frame=ts).tobytes()
    if ts == 2:                                   # float32 cut into 2-byte elements barely compresses: use 16-bit steps
        x = ((np.arange(n // 2) // 5) & 0xFFFF).astype(np.uint16).tobytes()
    f = hb.Compress(x, hb.LZ4, 5, hb.Shuffle1, ts, opts=hb.OPT_INDEX_TRAILER)
    a = hb.Decompress(f)
    # the index was used (not the serial fallback) -- unless the data did not compress and the frame is a memcpy frame
    assert hb.ParseHeader(f).IsMemcpy() or (hb.lib().hb_last_result_flags() & 1)
    assert a == x
    assert np.array_equal(O.decompress_frame(np.frombuffer(f, np.uint8)), np.frombuffer(x, np.uint8))
    # a stale index (payload byte changed) must still fall back safely: serial decoder -> staged -> gated un-shuffle
    bad = bytearray(f); bad[16 + 7] ^= 0x40
    try:
        y = hb.Decompress(bytes(bad))
        ref = None
        try:
            ref = O.decompress_frame(np.frombuffer(bytes(bad), np.uint8)).tobytes()
        except Exception:       # noqa: BLE001
            pass
        if ref is not None:
            assert y == ref
    except (hb.ErrDecompressionFailed, hb.ErrSizeMismatch, hb.ErrInvalidData):
        pass


def test_window_that_is_one_run_is_one_match(hb, O):
    # Skip acceleration can land a 64-position window in the middle of a long run; the run then fills the whole window
    # ("equals the byte before" mask all ones, lane 0 included).  That case once took count-trailing-zeros of 0 and cut
    # the run into 31-byte matches, with a result that depended on what LDS held past the end of the chunk.
    rng = np.random.default_rng(5)
    for head in (2300, 1000, 3100, 517):
        x = np.concatenate([rng.integers(0, 256, head, dtype=np.uint8), np.zeros(4096 - head, np.uint8)])
        x = np.tile(x, 64)                                      # 64 chunks, so workgroups also see used LDS
        fr = [hb.Compress(x.tobytes(), hb.LZ4, 5, hb.NoShuffle, 1, opts=hb.OPT_INDEX_TRAILER) for _ in range(4)]
        assert all(f == fr[0] for f in fr), "encoder output differs between runs"
        h = hb.ParseHeader(fr[0])
        # per chunk: a literal run (the random head, plus whatever of the run the growing stride jumped over) and one
        # or two matches -- not a string of 31-byte ones
        c = fr[0][16:h.NBytesComp]
        i = nseq = 0
        while i < len(c):
            t = c[i]; i += 1
            lit = t >> 4
            if lit == 15:
                while True:
                    b = c[i]; i += 1; lit += b
                    if b != 255:
                        break
            i += lit
            if i >= len(c):
                break
            i += 2
            if (t & 15) == 15:
                while c[i] == 255:
                    i += 1
                i += 1
            nseq += 1
        assert nseq <= 64 * 3, (head, nseq)
        assert hb.Decompress(fr[0]) == x.tobytes()
        assert np.array_equal(O.decompress_frame(np.frombuffer(fr[0], np.uint8)), x)


def _seeds():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_seeds.json")))


def test_reference_fuzz_seeds_on_device(hb, O):
    # The deterministic seed inputs of the reference's own fuzz targets (fuzz_test.go:26-133, :293-363), restated in
    # tests/golden/reference_seeds.json with outcomes derived by hand from blosc.go.  The device must give, for
    # Decompress and for DecompressWithSize(ts) with ts in {0,1,2,4,8} (fuzz_test.go:155-158): the stated outcome,
    # and -- independently of the hand derivation -- exactly what the oracle gives.
    S = _seeds()
    by_name = {"ErrInvalidData": hb.ErrInvalidData, "ErrInvalidHeader": hb.ErrInvalidHeader, "ErrInvalidVersion": hb.ErrInvalidVersion,
               "ErrInvalidCodec": hb.ErrInvalidCodec, "ErrSizeMismatch": hb.ErrSizeMismatch, "ErrDecompressionFailed": hb.ErrDecompressionFailed}
    by_code = {-1: hb.ErrInvalidData, -2: hb.ErrInvalidHeader, -3: hb.ErrInvalidVersion, -4: hb.ErrInvalidCodec,
               -5: hb.ErrSizeMismatch, -8: hb.ErrDecompressionFailed}

    def device(data, ts):
        try:
            return None, hb.DecompressWithSize(data, ts)
        except hb.BloscError as e:
            return type(e), None

    def oracle(data, ts):
        try:
            return None, O.decompress_frame(np.frombuffer(data, np.uint8), typesize_override=ts).tobytes()
        except O.OracleError as e:
            return by_code[e.code], None

    for s in S["decompress_seeds"]:
        data = bytes.fromhex(s["data"])
        for ts in S["decompress_with_size_sweep"]:
            err, out = device(data, ts)
            if s["expect"] == "ok":
                assert err is None and out == bytes.fromhex(s["out"]), (s["name"], ts, err)
            else:
                assert err is by_name[s["expect"]], (s["name"], ts, err)
            assert (err, out) == oracle(data, ts), (s["name"], ts)
    for s in S["header_seeds"]:
        data = bytes.fromhex(s["data"])
        if s["parse"] == "ok":
            h = hb.ParseHeader(data)
            f = s["fields"]
            assert (h.Version, h.VersionLZ, h.Flags, h.TypeSize, h.NBytesOrig, h.BlockSize, h.NBytesComp) == \
                   (f["Version"], f["VersionLZ"], f["Flags"], f["TypeSize"], f["NBytesOrig"], f["BlockSize"], f["NBytesComp"])
            assert h.Bytes() == data[:16]                                       # fuzz_test.go:400-421
            assert hb.GetDecompressedSize(data) == f["NBytesOrig"]              # fuzz_test.go:436-447
        else:
            with pytest.raises(by_name[s["parse"]]):
                hb.ParseHeader(data)
        err, out = device(data, 0)
        assert err is (None if s["decompress"] == "ok" else by_name[s["decompress"]]), (s["name"], err)
        assert (err, out) == oracle(data, 0), s["name"]


def test_reference_compress_seeds_on_device(hb, O):
    # FuzzCompress's seed inputs (fuzz_test.go:167-203): NoShuffle round trips must hold at every level the reference tries
    # (:256-266), every shuffle x typesize combination must come back too (the default memcpy policy makes even those exact),
    # odd type sizes are accepted (:269-274); every frame also decodes through the ORACLE decoder.
    S = _seeds()
    for s in S["compress_seeds"]:
        x = bytes.fromhex(s["data"])
        xa = np.frombuffer(x, np.uint8)
        for level in S["compress_levels"]:
            f = hb.Compress(x, hb.LZ4, level, hb.NoShuffle, 1)
            assert hb.Decompress(f) == x and O.decompress_frame(np.frombuffer(f, np.uint8)).tobytes() == x, (s["name"], level)
        for shuffle in (hb.NoShuffle, hb.Shuffle1, hb.BitShuffle):
            for ts in (1, 2, 4, 8):
                f = hb.Compress(x, hb.LZ4, 5, shuffle, ts)
                assert hb.Decompress(f) == x, (s["name"], shuffle, ts)
                assert O.decompress_frame(np.frombuffer(f, np.uint8)).tobytes() == x, (s["name"], shuffle, ts)
                assert hb.Decompress(O.compress_frame(xa, shuffle=shuffle, typesize=ts).tobytes()) == x, (s["name"], shuffle, ts)
        for ts in S["compress_odd_typesizes"]:
            f = hb.Compress(x, hb.LZ4, 5, hb.NoShuffle, ts)
            assert hb.Decompress(f) == x and f[3] == (max(ts, 1) & 0xFF), (s["name"], ts)
    with pytest.raises(hb.ErrInvalidData):                                      # fuzz_test.go:205-214
        hb.Compress(b"", hb.LZ4, 5, hb.NoShuffle, 1)


def _forge_index(nunits_entries, payload_bytes, nbytes, chunk_field):
    """A checksum-correct HBIX header + entries (hb_lz4.h layout) for a literal-only payload."""
    magic, ver = 0x58494248, 1 | (16 << 16)
    h = [magic, ver, len(nunits_entries) - 1, chunk_field, payload_bytes, nbytes, 0]
    h.append(h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4] ^ h[5])
    out = struct.pack("<8I", *h)
    for e in nunits_entries:
        out += struct.pack("<4I", *e)
    return out


def test_forged_self_consistent_index_is_not_trusted(hb, O):
    # an index whose units are NOT one HB_CHUNK of output each -- but which is checksum-correct and
    # self-consistent -- must not steer the fused un-shuffle into decoding one plane twice and another never.
    # Frame: 64 KiB of data, Shuffle1 ts=4, payload = ONE literal-only LZ4 sequence holding the shuffled bytes.
    rng = np.random.default_rng(77)
    n = 65536
    x = rng.integers(0, 256, n, dtype=np.uint8)
    sh = O.filter(O.OP_SHUFFLE, x, 4).tobytes()
    ext = n - 15
    hdr = bytes([0xF0]) + b"\xff" * (ext // 255) + bytes([ext % 255])
    payload = hdr + sh
    cbytes = 16 + len(payload)
    head = struct.pack("<BBBBIII", 2, hb.LZ4, 0x1, 4, n, n, cbytes)
    pad = b"\0" * (((cbytes + 7) & ~7) - cbytes)
    want = O.decompress_frame(np.frombuffer(head + payload, np.uint8)).tobytes()
    assert want == x.tobytes()
    for unit in (2048, 4096, 8192):
        ents = []
        for d in range(0, n, unit):                       # entry: inside the literal run, rem = n - d, token at payload offset 0
            ents.append((len(hdr) + d, d, n - d, 0))
        ents[0] = (0, 0, 0xFFFFFFFF, 0)
        ents.append((len(payload), n, 0, 0))
        for chunk_field in (unit, 4096):
            idx = _forge_index(ents, len(payload), n, chunk_field)
            got = hb.Decompress(head + payload + pad + idx)
            assert got == want, (unit, chunk_field)
            used = hb.lib().hb_last_result_flags() & 1
            assert used == (1 if (unit == 4096 and chunk_field == 4096) else 0), (unit, chunk_field, used)


def test_final_token_with_match_nibble_is_rejected_like_the_reference(hb, O):
    # a block that ends after a literal run whose token announces a match (low nibble != 0) is an error in
    # UncompressBlock (si == len(src) needs matchNibble == 0; oracle ob_lz4_decompress) -- with or without an index.
    n = 8192
    x = np.random.default_rng(5).integers(0, 256, n, dtype=np.uint8).tobytes()
    ext = n - 15
    for nib, ok in ((0, True), (3, False)):
        hdr = bytes([0xF0 | nib]) + b"\xff" * (ext // 255) + bytes([ext % 255])
        payload = hdr + x
        cbytes = 16 + len(payload)
        head = struct.pack("<BBBBIII", 2, hb.LZ4, 0, 1, n, n, cbytes)
        pad = b"\0" * (((cbytes + 7) & ~7) - cbytes)
        ents = [(0, 0, 0xFFFFFFFF, 0), (len(hdr) + 4096, 4096, n - 4096, 0), (len(payload), n, 0, 0)]
        idx = _forge_index(ents, len(payload), n, 4096)
        for frame in (head + payload, head + payload + pad + idx):
            try:
                oracle = O.decompress_frame(np.frombuffer(frame, np.uint8)).tobytes()
            except O.OracleError as e:
                oracle = e.code
            assert (oracle == x) if ok else (oracle == -8)
            if ok:
                assert hb.Decompress(frame) == x
            else:
                with pytest.raises(hb.ErrDecompressionFailed):
                    hb.Decompress(frame)
