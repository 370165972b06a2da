"""GPU parity for SURVEY §8 row f3: LZ4HC (codec.go:90-128) and Snappy (codec.go:228-244) on the device, and Options.Level as
a speed knob for LZ4 (codec.go:63-66 ignores it; any level must still give a valid block).

Checkers: the oracle's decoders (restated reference `Decompress`), liblz4 for the LZ4 block behind codec id 2, libsnappy (the
format's own library) for codec id 3.  The reference pins no compressed bytes for these codecs either (third-party encoders
that are not in its tree): what is pinned is decodability by the reference's decoders + its error identities.
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


def _libsnappy():
    p = "/opt/conda/lib/libsnappy.so.1"
    if not os.path.exists(p):
        return None
    lib = ctypes.CDLL(p)
    lib.snappy_uncompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    lib.snappy_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    lib.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]; lib.snappy_max_compressed_length.restype = ctypes.c_size_t
    return lib


def _cases(O):
    rng = np.random.default_rng(12)
    return {
        "f32_1MiB": O.synth(O.D_F32, 1 << 18), "f64": O.synth(O.D_F64, 70000), "i32": O.synth(O.D_I32, 1 << 16),
        "ramp": O.synth(O.D_RAMP, 50000), "mod256": (np.arange(100000) % 256).astype(np.uint8),     # blosc_test.go:363-371
        "zeros": np.zeros(10000, np.uint8), "random": rng.integers(0, 256, 50001, dtype=np.uint8),
        "run_70000": np.full(70000, 9, np.uint8), "tiny_1": np.array([7], np.uint8), "tiny_13": np.arange(13, dtype=np.uint8),
        "period100": np.tile(rng.integers(0, 256, 100, dtype=np.uint8), 700),
        "few_valued": (rng.integers(0, 4, 1 << 18, dtype=np.uint8) * 64),
        "ragged_multi_tile": O.synth(O.D_F32, (3 << 20) // 4 + 5).view(np.uint8)[: (3 << 20) + 17].copy(),
    }


MODES = [(0, 1), (1, 4), (2, 4), (1, 8), (1, 2), (1, 3)]


def test_lz4hc_frames_every_level(hb, O):
    # blosc_test.go:91-105 (LZ4HC round trip) + codec.go:96-106 (level map): every level gives codec id 2 and an LZ4 block that
    # the reference's decoder (oracle), liblz4 and the device decode to the input; deeper levels never compress worse on the
    # structured sets
    lz = _liblz4()
    for name, x in _cases(O).items():
        xb = x.tobytes()
        sizes = {}
        for level in (1, 5, 7, 9):
            for shuffle, ts in (MODES if level == 5 else MODES[:2]):
                f = hb.Compress(xb, hb.LZ4HC, level, shuffle, ts, opts=hb.OPT_INDEX_TRAILER)
                h = hb.ParseHeader(f)
                assert h.VersionLZ == hb.LZ4HC and h.NBytesOrig == len(xb)
                assert O.decompress_frame(np.frombuffer(f, np.uint8)).tobytes() == xb, (name, level, shuffle, ts)
                assert hb.Decompress(f) == xb, (name, level, shuffle, ts)
                if not h.IsMemcpy():
                    assert hb.lib().hb_last_result_flags() & 1, (name, level)
                    if lz is not None:
                        out = ctypes.create_string_buffer(len(xb))
                        filt = xb if (shuffle == 0 or ts <= 1) else O.filter({1: O.OP_SHUFFLE, 2: O.OP_BITSHUFFLE}[shuffle], x, ts).tobytes()
                        assert lz.LZ4_decompress_safe(f[16:h.NBytesComp], out, h.NBytesComp - 16, len(xb)) == len(xb) and out.raw == filt
                if (shuffle, ts) == (1, 4):
                    sizes[level] = h.NBytesComp
        if name in ("f32_1MiB", "few_valued", "i32"):
            assert sizes[9] <= sizes[7] <= sizes[5] <= sizes[1] * 1.001, (name, sizes)
            lz4 = hb.ParseHeader(hb.Compress(xb, hb.LZ4, 5, 1, 4)).NBytesComp
            assert sizes[9] <= lz4, (name, sizes, lz4)
            if name != "i32":                                               # (two random + two zero byte planes: nothing to find)
                assert sizes[9] < lz4, (name, sizes, lz4)                   # the deeper search must buy something


def test_lz4_level_is_a_speed_knob_only(hb, O):
    # codec.go:63-66: lz4Codec ignores the level; here it only changes how hard the matcher skips -- validity never
    for name, x in _cases(O).items():
        xb = x.tobytes()
        got = {}
        for level in (-5, 1, 3, 5, 7, 9, 100):                              # blosc_test.go:613-655: out-of-range levels are clamped
            f = hb.Compress(xb, hb.LZ4, level, hb.Shuffle1, 4, opts=hb.OPT_INDEX_TRAILER)
            assert hb.Decompress(f) == xb and O.decompress_frame(np.frombuffer(f, np.uint8)).tobytes() == xb, (name, level)
            got[level] = hb.ParseHeader(f).NBytesComp
        assert got[-5] == got[1] and got[100] == got[9]
        assert got[9] <= got[5] * 1.0005 and got[5] <= got[1] * 1.0005, (name, got)


def test_snappy_frames(hb, O):
    # blosc_test.go:69-89 (Snappy round trip): codec id 3; the payload is ONE Snappy block that the reference's decoder (oracle),
    # libsnappy and the device decode to the filtered input
    sn = _libsnappy()
    for name, x in _cases(O).items():
        xb = x.tobytes()
        for shuffle, ts in MODES:
            for opts in (hb.OPT_INDEX_TRAILER, 0):
                f = hb.Compress(xb, hb.Snappy, 5, shuffle, ts, opts=opts)
                h = hb.ParseHeader(f)
                assert (h.Version, h.VersionLZ, h.TypeSize, h.NBytesOrig) == (2, hb.Snappy, ts, len(xb))
                assert O.decompress_frame(np.frombuffer(f, np.uint8)).tobytes() == xb, (name, shuffle, ts)
                assert hb.Decompress(f) == xb, (name, shuffle, ts, opts)
                if not h.IsMemcpy():
                    # without the stored unit index: the element discovery + 64 KiB units when the block is worth it (hb_indexless_parallel)
                    payload = h.NBytesComp - 16
                    worth = payload >= (256 << 10) or (payload >= (16 << 10) and len(xb) >= (2 << 20))
                    assert (hb.lib().hb_last_result_flags() & 1) == (1 if (opts or worth) else 0), (name, shuffle, ts, opts)
                    if sn is not None and opts:
                        out = ctypes.create_string_buffer(len(xb))
                        ol = ctypes.c_size_t(len(xb))
                        assert sn.snappy_uncompress(f[16:h.NBytesComp], h.NBytesComp - 16, out, ctypes.byref(ol)) == 0 and ol.value == len(xb)
                        filt = xb if (shuffle == 0 or ts <= 1) else O.filter({1: O.OP_SHUFFLE, 2: O.OP_BITSHUFFLE}[shuffle], x, ts).tobytes()
                        assert out.raw == filt, (name, shuffle, ts)
    with pytest.raises(hb.ErrInvalidData):
        hb.Compress(b"", hb.Snappy, 5, hb.NoShuffle, 1)                     # blosc.go:269-271


def test_device_decodes_foreign_snappy_frames(hb, O):
    # frames whose payload was written by the oracle's encoder (64 KiB blocks, offsets up to 65535) and by libsnappy: no
    # index -> the single-wavefront decoder, or (payloads from 256 KiB: the next test) the discovery + 64 KiB units; plus a hand-made block with
    # a 4-byte-offset copy reaching 100 000 bytes back
    sn = _libsnappy()
    for name, x in _cases(O).items():
        xb = x.tobytes()
        for shuffle, ts in MODES[:3]:
            f = O.compress_frame(x, codec=O.SNAPPY, shuffle=shuffle, typesize=ts).tobytes()
            assert hb.Decompress(f) == xb, (name, shuffle, ts)
            h = hb.GetInfo(f)
            payload = h.NBytesComp - 16
            if not h.IsMemcpy() and (payload >= (256 << 10) or (payload >= (16 << 10) and len(xb) >= (2 << 20))):
                assert hb.lib().hb_last_result_flags() & 1, (name, shuffle, ts, "a foreign Snappy block of this size decodes in parallel")
        if sn is not None and len(xb) > 20:
            cap = sn.snappy_max_compressed_length(len(xb))
            b = ctypes.create_string_buffer(cap)
            bl = ctypes.c_size_t(cap)
            assert sn.snappy_compress(xb, len(xb), b, ctypes.byref(bl)) == 0
            if bl.value < len(xb):
                frame = struct.pack("<BBBBIII", 2, hb.Snappy, 0, 1, len(xb), len(xb), 16 + bl.value) + b.raw[:bl.value]
                assert hb.Decompress(frame) == xb, name
    rng = np.random.default_rng(3)
    head = rng.integers(0, 256, 120000, dtype=np.uint8).tobytes()
    n = len(head) + 40

    def lit(b):
        x = len(b) - 1
        return (bytes([x << 2]) if x < 60 else bytes([61 << 2, x & 255, x >> 8]) if x < 65536 else bytes([62 << 2, x & 255, (x >> 8) & 255, x >> 16])) + b
    block = bytes([n & 127 | 128, (n >> 7) & 127 | 128, n >> 14]) + lit(head) + bytes([(40 - 1) << 2 | 3]) + struct.pack("<I", 100000)
    want = head + head[len(head) - 100000:][:40]
    assert O.snappy_decompress(np.frombuffer(block, np.uint8), n).tobytes() == want
    frame = struct.pack("<BBBBIII", 2, hb.Snappy, 0, 1, n, n, 16 + len(block)) + block
    assert hb.Decompress(frame) == want


def test_foreign_snappy_frames_decode_block_parallel(hb, O):
    """VERDICT r2/r3 item 7 (codec.go:236-244): a Snappy block that comes without this library's unit index -- written by another encoder; here the
    oracle's, which like golang/snappy and libsnappy compresses 64 KiB blocks that share nothing -- goes through the element discovery
    (hb_lz4_region.hip with the element parser) and is decoded one 64 KiB unit per wavefront (k_sn_dec_units): flags & 1.  A stream whose
    copies cross those units (what klauspost's s2.EncodeSnappy writes for one large block) must come out right too: the single wavefront."""
    rng = np.random.default_rng(77)
    n = 12 << 20
    sets = {
        "f32": (O.synth(O.D_F32, n // 4).view(np.uint8), 1, 4),
        "f64": (O.synth(O.D_F64, n // 8).view(np.uint8), 1, 8),
        "i32_bits": (O.synth(O.D_I32, n // 4).view(np.uint8), 2, 4),
        "ramp": (O.synth(O.D_RAMP, n // 4).view(np.uint8), 1, 4),
        "text": (np.frombuffer((b"It was the best of times, it was the worst of times, " * (n // 50))[:n], np.uint8), 0, 1),
        "ragged": (O.synth(O.D_F32, n // 4 + 3).view(np.uint8)[: n + 12345 - (n + 12345) % 4], 1, 4),
        # half noise (literal elements of 64 KiB: regions with no element start at all), half structure
        "mixed": (np.concatenate([rng.integers(0, 256, n // 2, dtype=np.uint8), O.synth(O.D_F32, n // 8).view(np.uint8)]), 0, 1),
    }
    for name, (x, shuffle, ts) in sets.items():
        xb = x.tobytes()
        f = O.compress_frame(x, codec=O.SNAPPY, shuffle=shuffle, typesize=ts).tobytes()
        h = hb.GetInfo(f)
        if h.IsMemcpy():
            continue
        assert hb.Decompress(f) == xb, name
        par = hb.lib().hb_last_result_flags() & 1
        if name != "mixed":                                      # (long literal runs may leave the chain to the single wavefront: correct either way)
            assert par, (name, "expected the block-parallel path")
    # copies that reach across the 64 KiB units: valid Snappy, not block-structured
    base = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()

    def lit(b):
        x = len(b) - 1
        return (bytes([x << 2]) if x < 60 else bytes([61 << 2, x & 255, x >> 8]) if x < 65536 else bytes([62 << 2, x & 255, (x >> 8) & 255, x >> 16])) + b
    body = lit(base)
    want = bytearray(base)
    while len(want) < (1 << 20):
        off = int(rng.integers(66000, 70000))
        ln = int(rng.integers(8, 65))
        body += bytes([(ln - 1) << 2 | 3]) + struct.pack("<I", off)
        want += want[len(want) - off: len(want) - off + ln]
        extra = rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8).tobytes()
        body += lit(extra); want += extra
    n2 = len(want)
    var = b""
    v = n2
    while v >= 128:
        var += bytes([v & 127 | 128]); v >>= 7
    var += bytes([v])
    block = var + body
    assert O.snappy_decompress(np.frombuffer(block, np.uint8), n2).tobytes() == bytes(want)
    frame = struct.pack("<BBBBIII", 2, hb.Snappy, 0, 1, n2, n2, 16 + len(block)) + block
    assert hb.Decompress(frame) == bytes(want)
    assert not (hb.lib().hb_last_result_flags() & 1)


def test_foreign_snappy_frames_through_both_workspaces(hb, O):
    """Device-pointer API: with either workspace an index-less Snappy block is decoded one 64 KiB unit per wavefront (k_sn_dec_units; needs an encoder
    that compresses 64 KiB blocks that share nothing); hb_decompress_frame_workspace_foreign() adds the symbolic decoder behind it (hb_lz4_sym.hip fed
    with elements: any stream whose offsets fit 16 bits).  A stream with copies ACROSS the 64 KiB units tells them apart:
    the single wavefront with the small workspace, in parallel with the large one; same bytes every time."""
    L = hb.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    H2D, D2H = 1, 2

    def dmalloc(nb):
        ptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(ptr), nb) == 0
        return ptr

    rng = np.random.default_rng(5)
    x = O.synth(O.D_F32, (8 << 20) // 4)
    block_structured = O.compress_frame(x, codec=O.SNAPPY, shuffle=1, typesize=4)
    # copies that reach 40 000 - 60 000 bytes back, every few dozen bytes: across every 64 KiB boundary, offsets in 16 bits
    base = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()

    def lit(b):
        v = len(b) - 1
        return (bytes([v << 2]) if v < 60 else bytes([61 << 2, v & 255, v >> 8]) if v < 65536 else bytes([62 << 2, v & 255, (v >> 8) & 255, v >> 16])) + b
    body = lit(base[:65536]) + lit(base[65536:])
    want = bytearray(base)
    while len(want) < (2 << 20):
        off = int(rng.integers(40000, 60000))
        ln = int(rng.integers(8, 65))
        body += bytes([(ln - 1) << 2 | 2]) + struct.pack("<H", off)
        want += want[len(want) - off: len(want) - off + ln]
        extra = rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8).tobytes()
        body += lit(extra); want += extra
    n2 = len(want)
    var = b""
    v = n2
    while v >= 128:
        var += bytes([v & 127 | 128]); v >>= 7
    var += bytes([v])
    block = var + body
    assert O.snappy_decompress(np.frombuffer(block, np.uint8), n2).tobytes() == bytes(want)
    crossing = np.frombuffer(struct.pack("<BBBBIII", 2, hb.Snappy, 0, 1, n2, n2, 16 + len(block)) + block, np.uint8)
    for f, raw, flags_small, flags_large in ((block_structured, x.view(np.uint8).reshape(-1), 1, 1), (crossing, np.frombuffer(bytes(want), np.uint8), 0, 1)):
        n = raw.size
        small, large = L.hb_decompress_frame_workspace(n), L.hb_decompress_frame_workspace_foreign(n)
        d_frame, d_out, d_res, d_work = dmalloc(f.size + 64), dmalloc(n), dmalloc(64), dmalloc(large)
        try:
            assert hip.hipMemcpy(d_frame, f.ctypes.data, f.size, H2D) == 0
            for wb, want_flag in ((small, flags_small), (large, flags_large)):
                assert hip.hipMemset(d_out, 0, n) == 0
                rc = L.hb_decompress_frame_dev(d_frame, f.size, d_out, n, 0, d_work, wb, d_res, None)
                assert rc == 0 and hip.hipDeviceSynchronize() == 0
                res = np.zeros(32, np.uint8)
                back = np.empty(n, np.uint8)
                assert hip.hipMemcpy(res.ctypes.data, d_res, 32, D2H) == 0 and hip.hipMemcpy(back.ctypes.data, d_out, n, D2H) == 0
                assert int(res[:4].view(np.int32)[0]) == 0
                assert np.array_equal(back, raw)
                assert int(res[4:8].view(np.uint32)[0]) & 1 == want_flag, (f.size, wb == large)
        finally:
            for ptr in (d_frame, d_out, d_res, d_work):
                hip.hipFree(ptr)


def test_snappy_errors_match_the_oracle(hb, O):
    # mutated Snappy frames: the device must report what the restated reference decoder reports (error class, or the same bytes)
    rng = np.random.default_rng(99)
    by_code = {-1: hb.ErrInvalidData, -2: hb.ErrInvalidHeader, -3: hb.ErrInvalidVersion, -4: hb.ErrInvalidCodec,
               -5: hb.ErrSizeMismatch, -8: hb.ErrDecompressionFailed}
    x = O.synth(O.D_F32, 6000).tobytes()
    seeds = [hb.Compress(x, hb.Snappy, 5, 1, 4, opts=hb.OPT_INDEX_TRAILER), hb.Compress(x, hb.Snappy, 5, 0, 1),
             O.compress_frame(np.frombuffer(x, np.uint8), codec=O.SNAPPY, shuffle=2, typesize=4).tobytes()]
    checked = 0
    for f in seeds:
        cb = hb.ParseHeader(f).NBytesComp
        for trial in range(120):
            g = bytearray(f)
            kind = trial % 4
            if kind == 0:
                g[int(rng.integers(16, cb))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                pos = int(rng.integers(16, cb))
                g[pos:pos + 4] = rng.integers(0, 256, min(4, len(g) - pos), dtype=np.uint8).tobytes()
            elif kind == 2:
                cut = int(rng.integers(17, cb))
                g = g[:cut]; g[12:16] = struct.pack("<I", cut)
            else:
                g[4:8] = struct.pack("<I", max(1, int.from_bytes(g[4:8], "little") + int(rng.integers(-50, 50))))
            g = bytes(g)
            try:
                want = (None, O.decompress_frame(np.frombuffer(g, np.uint8)).tobytes())
            except O.OracleError as e:
                want = (by_code[e.code], None)
            try:
                got = (None, hb.Decompress(g))
            except hb.BloscError as e:
                got = (type(e), None)
            assert got == want, (trial, kind, got[0], want[0])
            checked += 1
    assert checked == 360


def test_mutated_foreign_snappy_frames_on_the_parallel_paths(hb, O):
    """The same contract as test_snappy_errors_match_the_oracle at a size where a block without an index goes through the element discovery and the
    parallel decoders (payload >= 256 KiB): whatever a mutation does to the chain -- copies that reach too far, elements that run off the stream, a
    declared length that no longer fits, literals of megabytes -- the device reports what the restated reference decoder reports, or the same bytes."""
    rng = np.random.default_rng(1234)
    by_code = {-1: hb.ErrInvalidData, -2: hb.ErrInvalidHeader, -3: hb.ErrInvalidVersion, -4: hb.ErrInvalidCodec,
               -5: hb.ErrSizeMismatch, -8: hb.ErrDecompressionFailed}
    x = np.concatenate([O.synth(O.D_F32, (1 << 20) // 4).view(np.uint8), rng.integers(0, 256, 1 << 19, dtype=np.uint8),
                        np.frombuffer((b"a foreign snappy frame, mutated. " * 20000)[: 1 << 19], np.uint8)])
    seeds = [O.compress_frame(x, codec=O.SNAPPY, shuffle=0, typesize=1).tobytes(), O.compress_frame(x, codec=O.SNAPPY, shuffle=1, typesize=4).tobytes()]
    checked = parallel = 0
    for f in seeds:
        cb = hb.ParseHeader(f).NBytesComp
        assert cb - 16 >= (256 << 10)
        for trial in range(48):
            g = bytearray(f)
            kind = trial % 6
            if kind == 0:
                g[int(rng.integers(16, cb))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                pos = int(rng.integers(16, cb))
                g[pos:pos + 4] = rng.integers(0, 256, min(4, len(g) - pos), dtype=np.uint8).tobytes()
            elif kind == 2:
                cut = int(rng.integers(cb // 2, cb))
                g = g[:cut]; g[12:16] = struct.pack("<I", cut)
            elif kind == 3:
                g[4:8] = struct.pack("<I", max(1, int.from_bytes(g[4:8], "little") + int(rng.integers(-50, 50))))
            elif kind == 4:                                   # a tag that announces a literal of up to 4 GiB / a copy with a 4-byte offset
                g[int(rng.integers(16 + 8, cb - 8))] = int(rng.choice([0xF8, 0xFC, 0xFF, 0x03, 0xF4]))
            else:                                             # a stretch of the stream replaced by noise
                pos = int(rng.integers(16 + 8, cb - 5000))
                g[pos:pos + 4096] = rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()
            g = bytes(g)
            try:
                want = (None, O.decompress_frame(np.frombuffer(g, np.uint8)).tobytes())
            except O.OracleError as e:
                want = (by_code[e.code], None)
            try:
                got = (None, hb.Decompress(g))
                parallel += hb.lib().hb_last_result_flags() & 1
            except hb.BloscError as e:
                got = (type(e), None)
            assert got == want, (trial, kind, got[0], want[0])
            checked += 1
    assert checked == 96


def test_snappy_forged_index_is_not_trusted(hb, O):
    # a checksum-correct HBSX index with the wrong unit geometry, or with offsets that do not sit on element boundaries, must
    # not change the decoded bytes
    x = O.synth(O.D_F32, 8192).tobytes()
    f = hb.Compress(x, hb.Snappy, 5, hb.Shuffle1, 4, opts=hb.OPT_INDEX_TRAILER)
    h = hb.ParseHeader(f)
    ioff = (h.NBytesComp + 7) & ~7
    idx = bytearray(f[ioff:])
    w = list(struct.unpack("<8I", idx[:32]))
    assert w[0] == 0x58534248 and w[2] == 8
    assert hb.Decompress(f) == x and hb.lib().hb_last_result_flags() & 1
    for edit in ("half_units", "shift_entry", "swap"):
        g = bytearray(idx)
        ww = list(w)
        ents = list(struct.unpack("<9I", g[32:32 + 36]))
        if edit == "half_units":
            ww[2] = 4; ww[3] = 8192; ents = ents[0:9:2]
        elif edit == "shift_entry":
            ents[3] += 1
        else:
            ents[2], ents[5] = ents[5], ents[2]
        ww[7] = ww[0] ^ ww[1] ^ ww[2] ^ ww[3] ^ ww[4] ^ ww[5] ^ ww[6]
        forged = f[:ioff] + struct.pack("<8I", *ww) + struct.pack(f"<{len(ents)}I", *ents)
        assert hb.Decompress(forged) == x, edit
        assert not (hb.lib().hb_last_result_flags() & 1), edit


def test_f3_codecs_through_the_queue_and_at_full_size(hb, O):
    L = hb.lib()
    n = 64 << 20
    x = O.synth(O.D_F32, n // 4, frame=2)
    cap = L.hb_frame_bound(n)
    pin_in, pin_out, back = hb.PinnedBuffer(n), hb.PinnedBuffer(cap), hb.PinnedBuffer(n)
    ctypes.memmove(pin_in.ptr, x.ctypes.data, n)
    q = hb.FrameQueue(n, depth=2)
    ratios = {}
    for codec, level in ((hb.Snappy, 5), (hb.LZ4HC, 9), (hb.LZ4HC, 5), (hb.LZ4, 5)):
        c = q.wait(q.compress(pin_in.ptr, n, pin_out.ptr, cap, codec, level, hb.Shuffle1, 4, hb.OPT_INDEX_TRAILER))
        frame = bytes(pin_out.view[:c])
        h = hb.ParseHeader(frame)
        assert h.VersionLZ == codec and not h.IsMemcpy()
        assert np.array_equal(O.decompress_frame(np.frombuffer(frame[:h.NBytesComp], np.uint8)), x), (codec, level)
        assert q.wait(q.decompress(pin_out.ptr, c, back.ptr, n)) == n
        assert bytes(back.view[:n]) == x.tobytes(), (codec, level)
        ratios[(codec, level)] = h.NBytesComp / n
    q.close()
    assert ratios[(hb.LZ4HC, 9)] < ratios[(hb.LZ4HC, 5)] < ratios[(hb.LZ4, 5)] < ratios[(hb.Snappy, 5)] + 0.2, ratios


def test_codec_seam_bare_blocks(hb, O):
    # CodecInterface (codec.go:15-24) for the two f3 codecs: Compress gives the codec's own block -- never the input back, the
    # memcpy rule is the frame layer's (blosc.go:342) -- and Decompress takes any block of that format
    lz, sn = _liblz4(), _libsnappy()
    rng = np.random.default_rng(3)
    sets = dict(_cases(O))
    sets["f32_5MiB"] = O.synth(O.D_F32, (5 << 20) // 4 + 1)
    hc, sy = hb.codecs[hb.LZ4HC], hb.codecs[hb.Snappy]
    assert (hc.Name(), sy.Name()) == ("lz4hc", "snappy")                # codec.go:92, :230
    for name, x in sets.items():
        xb = x.tobytes()
        for level in (1, 9):
            b = hc.Compress(xb, level)
            assert O.lz4_decompress(np.frombuffer(b, np.uint8), len(xb)).tobytes() == xb, (name, level)
            assert hc.Decompress(b, len(xb)) == xb, (name, level)
            if lz is not None:
                back = ctypes.create_string_buffer(len(xb) + 1)
                assert lz.LZ4_decompress_safe(b, back, len(b), len(xb)) == len(xb) and back.raw[:len(xb)] == xb, (name, level)
        b = sy.Compress(xb, 5)
        assert O.snappy_decompress(np.frombuffer(b, np.uint8), len(xb)).tobytes() == xb, name
        assert sy.Decompress(b, len(xb)) == xb, name
        if sn is not None:
            ln = ctypes.c_size_t(len(xb) + 1)
            back = ctypes.create_string_buffer(len(xb) + 1)
            assert sn.snappy_uncompress(b, len(b), back, ctypes.byref(ln)) == 0 and back.raw[:ln.value] == xb, name
            # the other way: the format library's block through the device decoder
            cap = sn.snappy_max_compressed_length(len(xb)); cl = ctypes.c_size_t(cap)
            fb = ctypes.create_string_buffer(cap)
            assert sn.snappy_compress(xb, len(xb), fb, ctypes.byref(cl)) == 0
            assert sy.Decompress(fb.raw[:cl.value], len(xb)) == xb, name
    # random bytes: the frame layer would store them, the codec still returns a block (longer than the input)
    r = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()
    assert len(hc.Compress(r, 9)) > len(r) and len(sy.Compress(r, 5)) > len(r)
    # empty input: snappy.Encode gives the one length byte, an LZ4 block of nothing is nothing
    assert sy.Compress(b"", 5) == b"\x00" and sy.Decompress(b"\x00", 0) == b""
    assert hc.Compress(b"", 9) == b""
    # errors: a declared length above the caller's size, a cut block, a codec without a device implementation
    b = sy.Compress(sets["f32_1MiB"].tobytes(), 5)
    with pytest.raises(hb.BloscError):
        sy.Decompress(b, 1000)
    with pytest.raises(hb.ErrDecompressionFailed):
        sy.Decompress(b[:len(b) // 2], sets["f32_1MiB"].nbytes)
    with pytest.raises(hb.ErrDecompressionFailed):
        hc.Decompress(hc.Compress(sets["f64"].tobytes(), 9)[:-3], sets["f64"].nbytes)
    L = hb.lib()
    buf = ctypes.create_string_buffer(64)
    assert L.hb_codec_bound(hb.ZSTD, 100) == 0
    assert L.hb_codec_compress(hb.ZSTD, 5, buf, 8, buf, 64, 0) == -4       # HB_ERR_INVALID_CODEC (codec.go:46-55 GetCodec)
