"""GPU parity: frames WITHOUT a restart index.

Two kinds exist.  (1) Frames this library writes without HB_OPT_INDEX_TRAILER (the drop-in default: a reference frame ends at
NBytesComp): their matches never leave a 4 KiB chunk, so the device rebuilds the index from the stream (guess-and-verify token
discovery, csrc/hb_lz4_region.hip) and decodes chunk-parallel.  (2) Frames the REFERENCE writes (one LZ4 block, 64 KiB window,
codec.go:63-75 -- here: the oracle's restatement of lz4.CompressBlock, and liblz4 as a second foreign parse): no index can hold
for them (matches cross every chunk boundary); they are decoded in parallel from the verified token chain, symbolically
(csrc/hb_lz4_sym.hip), when the caller's workspace has room for it, and by one wavefront otherwise.  Whatever path runs, bytes and
errors must be the oracle decoder's (the restated reference `Decompress`).
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


def _cases(O):
    rng = np.random.default_rng(2026)
    f32 = O.synth(O.D_F32, (24 << 20) // 4 + 3)
    out = {
        # name: (data, shuffle, typesize, expect the parallel path)
        "f32_24MiB_shuffle": (f32, 1, 4, True),
        "f32_24MiB_noshuffle": (f32, 0, 1, None),                     # (barely compressible: may be a memcpy frame)
        "f64_16MiB_shuffle8": (O.synth(O.D_F64, (16 << 20) // 8), 1, 8, True),
        "i32_16MiB_bitshuffle": (O.synth(O.D_I32, (16 << 20) // 4), 2, 4, True),
        # byte-shuffled integers: three planes of noise (literal runs of tens of MiB) in front of a plane of long runs, whose stream is
        # PERIODIC -- parses that start off the chain there stay off it (the discovery's slowest case: region after region)
        "i32_96MiB_shuffle": (O.synth(O.D_I32, (96 << 20) // 4), 1, 4, True),
        "ramp_8MiB": (O.synth(O.D_RAMP, (8 << 20) // 4 + 1), 1, 4, None),            # tiny payload: long matches, few tokens
        "few_valued_12MiB": (rng.integers(0, 4, 12 << 20, dtype=np.uint8) * 64, 0, 1, True),      # token-dense, short offsets
        "period_8192": (np.tile(rng.integers(0, 256, 8192, dtype=np.uint8), 1500), 0, 1, None),   # matches chain period by period
        "noisy_period": (None, 0, 1, None),
        "rand_then_zeros_then_text": (np.concatenate([rng.integers(0, 256, (3 << 20) + 5, dtype=np.uint8), np.zeros((9 << 20) + 1, np.uint8),
                                                      np.frombuffer((b"the quick brown fox jumps over the lazy dog. " * 90000), np.uint8),
                                                      rng.integers(0, 256, 70001, dtype=np.uint8), np.full(3 << 20, 7, np.uint8)]), 0, 1, None),
        "far_offsets": (None, 0, 1, None),
    }
    base = np.tile(rng.integers(0, 256, 8192, dtype=np.uint8), 1200)
    noise = rng.integers(0, base.size, base.size // 300)
    base[noise] ^= 0x5A
    out["noisy_period"] = (base, 0, 1, None)
    # blocks of 40 KiB random bytes, each repeated once 40 KiB later: every match reaches 40960 bytes back (beyond the LDS history)
    blk = [rng.integers(0, 256, 40960, dtype=np.uint8) for _ in range(60)]
    out["far_offsets"] = (np.concatenate([np.concatenate([b, b]) for b in blk]), 0, 1, None)
    return out


def _many_big_runs(rng):
    # more runs of SY_BIG bytes and above in ONE region of the stream than pass A has launches to park them in (the last launch
    # copies inline), separated by a few literals, then enough random bytes for the payload to qualify for the region path
    parts = []
    for k in range(12):
        parts.append(np.full((300 << 10) + 17 * k, k + 1, np.uint8))
        parts.append(rng.integers(0, 256, 5 + k, dtype=np.uint8))
    parts.append(np.tile(np.arange(7, dtype=np.uint8), 100000))            # one long match with period 7
    parts.append(rng.integers(0, 256, (1 << 20) + 3, dtype=np.uint8))
    parts.append(np.zeros((5 << 20) + 1, np.uint8))
    return np.concatenate(parts)


def test_reference_shaped_frames_decode_exactly(hb, O):
    cases = _cases(O)
    cases["many_big_runs"] = (_many_big_runs(np.random.default_rng(4)), 0, 1, None)
    parallel = {}
    for name, (x, shuffle, ts, _) in cases.items():
        f = O.compress_frame(x, shuffle=shuffle, typesize=ts)
        want = O.decompress_frame(f)
        assert np.array_equal(want, x.view(np.uint8).reshape(-1))
        assert hb.Decompress(f.tobytes()) == x.tobytes(), name
        parallel[name] = bool(hb.lib().hb_last_result_flags() & 1)
    # the symbolic decoder took every frame whose payload is large enough for the region path (256 KiB)
    for name in ("f32_24MiB_shuffle", "f64_16MiB_shuffle8", "i32_16MiB_bitshuffle", "few_valued_12MiB", "rand_then_zeros_then_text",
                 "far_offsets", "many_big_runs"):
        assert parallel[name], (name, parallel)


def test_foreign_frames_need_the_larger_workspace(hb, O):
    # device-pointer API: with hb_decompress_frame_workspace() a reference-shaped frame goes to the single wavefront, with
    # hb_decompress_frame_workspace_foreign() to the symbolic decoder; same bytes.  (Device buffers through the HIP runtime the
    # library itself is linked to -- a second runtime in the process, e.g. torch's, would not see the GPU.)
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

    x = O.synth(O.D_F32, (8 << 20) // 4)
    f = O.compress_frame(x, shuffle=1, typesize=4)
    n = x.nbytes
    small, large = L.hb_decompress_frame_workspace(n), L.hb_decompress_frame_workspace_foreign(n)
    assert large > small + 2 * n
    d_frame, d_out, d_res, d_work = dmalloc(f.size + 64), dmalloc(n), dmalloc(64), dmalloc(large)
    try:
        assert hip.hipMemcpy(d_frame, f.ctypes.data, f.size, H2D) == 0
        for wb, want_flag in ((small, 0), (large, 1)):
            assert hip.hipMemset(d_out, 0, n) == 0
            rc = L.hb_decompress_frame_dev(d_frame, f.size, d_out, n, 0, d_work, wb, d_res, None)
            assert rc == 0 and hip.hipDeviceSynchronize() == 0
            res = np.zeros(32, np.uint8)
            back = np.empty(n, np.uint8)
            assert hip.hipMemcpy(res.ctypes.data, d_res, 32, D2H) == 0 and hip.hipMemcpy(back.ctypes.data, d_out, n, D2H) == 0
            assert int(res[:4].view(np.int32)[0]) == 0
            assert int(res[4:8].view(np.uint32)[0]) & 1 == want_flag
            assert np.array_equal(back, x)
        # the workspace must be 256-byte aligned (include/hipblosc.h: its records are read and written with 16-byte vectors): refused, nothing launched
        HB_ERR_BAD_ARG = -11
        odd = ctypes.c_void_p(d_work.value + 8)
        assert L.hb_decompress_frame_dev(d_frame, f.size, d_out, n, 0, odd, large - 8, d_res, None) == HB_ERR_BAD_ARG
    finally:
        for ptr in (d_frame, d_out, d_res, d_work):
            hip.hipFree(ptr)


def test_rebuilt_index_is_the_same_from_stored_tokens_and_from_bucket_records(hb, O):
    # An own frame without the trailer, decoded through the device-pointer API twice: with the small workspace the index is rebuilt from the
    # discovery's bucket records (k_rg_index_fast) + the wave walk, with the larger one from the tokens the first parse stored (k_rg_index_tok,
    # csrc/hb_lz4_region.hip).  Same bytes out, and the SAME index in the workspace (its place: tools/region_debug.py --save-index).
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

    al = lambda v: (v + 255) & ~255
    for x, shuffle, ts in ((O.synth(O.D_F32, (48 << 20) // 4), 1, 4), (O.synth(O.D_I32, (24 << 20) // 4), 2, 4), (O.synth(O.D_F64, (32 << 20) // 8), 1, 8)):
        f = np.frombuffer(hb.Compress(x.tobytes(), hb.LZ4, 5, shuffle, ts, opts=0), np.uint8)
        n = x.nbytes
        small, large = L.hb_decompress_frame_workspace(n), L.hb_decompress_frame_workspace_foreign(n)
        bound = n + n // 255 + 16
        nr_cap = min(max(bound // 8192 + 2, min(bound, 8 << 20) // 4096 + 2), 16384)                 # rg_max_regions (csrc/hb_lz4_region.h)
        off_idx = al(n) + 256 + 256 + al(64) + al(nr_cap * 64) + al(nr_cap * 4) + al(nr_cap * 256 * 8)
        nun = (n + 4095) // 4096
        idx_bytes = 32 + 16 * (nun + 1)
        d_frame, d_out, d_res, d_work = dmalloc(f.size + 64), dmalloc(n), dmalloc(64), dmalloc(large)
        try:
            assert hip.hipMemcpy(d_frame, f.ctypes.data, f.size, H2D) == 0
            got = []
            for wb in (small, large):
                assert hip.hipMemset(d_out, 0, n) == 0 and hip.hipMemset(d_work, 0, off_idx + idx_bytes) == 0
                rc = L.hb_decompress_frame_dev(d_frame, f.size, d_out, n, 0, d_work, wb, d_res, None)
                assert rc == 0 and hip.hipDeviceSynchronize() == 0
                res = np.zeros(32, np.uint8)
                back = np.empty(n, np.uint8)
                index = np.empty(idx_bytes, np.uint8)
                assert hip.hipMemcpy(res.ctypes.data, d_res, 32, D2H) == 0 and hip.hipMemcpy(back.ctypes.data, d_out, n, D2H) == 0
                assert hip.hipMemcpy(index.ctypes.data, ctypes.c_void_p(d_work.value + off_idx), idx_bytes, D2H) == 0
                assert int(res[:4].view(np.int32)[0]) == 0 and int(res[4:8].view(np.uint32)[0]) & 1 == 1
                assert np.array_equal(back, x.view(np.uint8).reshape(-1))
                assert index[:4].tobytes() == b"HBIX" or int(index[:4].view(np.uint32)[0]) != 0
                got.append(index)
            assert np.array_equal(got[0], got[1]), (shuffle, ts)
        finally:
            for ptr in (d_frame, d_out, d_res, d_work):
                hip.hipFree(ptr)


def test_own_frames_without_the_trailer_decode_in_parallel(hb, O):
    # what Compress() returns by default (opts = 0): the index is rebuilt on the device and checked like a stored one
    for name, (x, shuffle, ts, _) in _cases(O).items():
        f = hb.Compress(x.tobytes(), hb.LZ4, 5, shuffle, ts, opts=0)
        h = hb.ParseHeader(f)
        assert len(f) == h.NBytesComp                                   # nothing behind the frame
        assert O.decompress_frame(np.frombuffer(f, np.uint8)).tobytes() == x.tobytes(), name
        assert hb.Decompress(f) == x.tobytes(), name
        if not h.IsMemcpy() and hb.indexless_parallel(h.NBytesComp - 16, h.NBytesOrig):
            assert hb.lib().hb_last_result_flags() & 1, f"{name}: the rebuilt index was not used"
        with_index = hb.Compress(x.tobytes(), hb.LZ4, 5, shuffle, ts, opts=hb.OPT_INDEX_TRAILER)
        assert with_index[:h.NBytesComp] == f                            # same frame, the trailer is only appended
    # LZ4HC frames are LZ4 blocks too
    x = _cases(O)["f32_24MiB_shuffle"][0]
    f = hb.Compress(x.tobytes(), hb.LZ4HC, 9, hb.Shuffle1, 4, opts=0)
    assert hb.Decompress(f) == x.tobytes() and hb.lib().hb_last_result_flags() & 1


def test_liblz4_blocks_decode_exactly(hb, O):
    # a second foreign encoder (different parse: LZ4_compress_default), wrapped in a go-blosc header by hand
    lz = _liblz4()
    if lz is None:
        pytest.skip("liblz4 not in this image")
    rng = np.random.default_rng(5)
    for name, x in {"f32_shuffled": O.filter(O.OP_SHUFFLE, O.synth(O.D_F32, (16 << 20) // 4), 4),
                    "text": np.frombuffer(b"".join(bytes(str(i * 7919 % 100003), "ascii") + b", " for i in range(1500000)), np.uint8),
                    "few_valued": rng.integers(0, 3, 10 << 20, dtype=np.uint8)}.items():
        n = x.size
        cap = lz.LZ4_compressBound(n)
        b = ctypes.create_string_buffer(cap)
        c = lz.LZ4_compress_default(x.tobytes(), b, n, cap)
        assert 0 < c < n
        frame = struct.pack("<BBBBIII", 2, hb.LZ4, 0, 1, n, n, 16 + c) + b.raw[:c]
        assert O.decompress_frame(np.frombuffer(frame, np.uint8)).tobytes() == x.tobytes()
        assert hb.Decompress(frame) == x.tobytes(), name
        # the bare-block entry point (the codec plugin seam, codec.go:77-84)
        assert hb.codecs[hb.LZ4].Decompress(b.raw[:c], n) == x.tobytes(), name


def test_malformed_foreign_frames_report_what_the_reference_reports(hb, O):
    # large index-less frames with damage: the region decoder must step aside and the result (error class or bytes) must be
    # the restated reference decoder's
    by_code = {-1: hb.ErrInvalidData, -2: hb.ErrInvalidHeader, -3: hb.ErrInvalidVersion, -4: hb.ErrInvalidCodec,
               -5: hb.ErrSizeMismatch, -8: hb.ErrDecompressionFailed}
    rng = np.random.default_rng(77)
    x = O.synth(O.D_F32, (2 << 20) // 4)
    for f in (O.compress_frame(x, shuffle=1, typesize=4).tobytes(), hb.Compress(x.tobytes(), hb.LZ4, 5, hb.Shuffle1, 4, opts=0)):
        _mutations_agree(hb, O, f, by_code, rng)


def _mutations_agree(hb, O, f, by_code, rng):
    cb = hb.ParseHeader(f).NBytesComp
    assert cb > (512 << 10)
    for trial in range(40):
        g = bytearray(f)
        kind = trial % 5
        if kind == 0:
            g[int(rng.integers(16, cb))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            pos = int(rng.integers(16, cb - 8))
            g[pos:pos + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
        elif kind == 2:
            cut = int(rng.integers(cb // 2, cb))
            g = g[:cut]; g[12:16] = struct.pack("<I", cut)
        elif kind == 3:
            g[4:8] = struct.pack("<I", int.from_bytes(g[4:8], "little") + int(rng.integers(-3000, 3000)))
        else:
            pos = int(rng.integers(16, cb - 2))
            g[pos:pos + 2] = b"\x00\x00"                                  # very likely an offset 0 somewhere on the chain
        g = bytes(g)
        try:
            want = (None, O.decompress_frame(np.frombuffer(g, np.uint8)).tobytes())
        except O.OracleError as e:
            want = (by_code[e.code], None)
        try:
            got = (None, hb.Decompress(g))
        except hb.BloscError as e:
            got = (type(e), None)
        assert got[0] is want[0], (trial, kind, got[0], want[0])
        if want[1] is not None:
            assert got[1] == want[1], (trial, kind)


def test_index_less_frames_at_full_size(hb, O):
    # BASELINE.json's headline frame without the trailer: written by this library (index rebuilt, parallel decode) and as the
    # reference would have written it (single wavefront; 256 MiB of it, to keep the test in seconds)
    import time
    L = hb.lib()
    n = 1 << 30
    x = O.synth(O.D_F32, n // 4)
    cap = L.hb_frame_bound(n)
    out = np.empty(cap, np.uint8)
    c = L.hb_compress_frame(x.ctypes.data, n, out.ctypes.data, cap, hb.LZ4, 5, hb.Shuffle1, 4, 0, 0)
    assert c == hb.ParseHeader(out[:16].tobytes()).NBytesComp
    back = np.empty(n, np.uint8)
    t0 = time.perf_counter()
    assert L.hb_decompress_frame(out.ctypes.data, c, back.ctypes.data, n, 0, 0) == n
    dt = time.perf_counter() - t0
    assert L.hb_last_result_flags() & 1, "the rebuilt index was not used on the 1 GiB frame"
    assert np.array_equal(back, x), "decode through the rebuilt index differs from the input at full size"
    print(f"index-less 1 GiB frame host->host: {n / dt / 1e9:.2f} GB/s")
    m = 256 << 20
    f = O.compress_frame(x[:m], shuffle=1, typesize=4)
    back[:m] = 0
    t0 = time.perf_counter()
    assert L.hb_decompress_frame(f.ctypes.data, f.size, back.ctypes.data, m, 0, 0) == m
    dt = time.perf_counter() - t0
    assert L.hb_last_result_flags() & 1, "the reference-shaped frame was not decoded in parallel"
    assert np.array_equal(back[:m], x[:m]), "reference-shaped frame: device decode differs"
    print(f"reference-shaped 256 MiB frame host->host: {m / dt / 1e9:.2f} GB/s")


def test_queue_with_room_for_foreign_frames(hb, O):
    # hb_queue_create_ex(HB_QUEUE_FOREIGN_FRAMES): reference-shaped frames through the pipelined queue decode in parallel
    x = O.synth(O.D_F32, (8 << 20) // 4)
    f = O.compress_frame(x, shuffle=1, typesize=4)
    n = x.nbytes
    for foreign, want in ((False, 0), (True, 1)):
        q = hb.FrameQueue(n, depth=2, foreign_frames=foreign)
        src, dst = hb.PinnedBuffer(f.size), hb.PinnedBuffer(n)
        ctypes.memmove(src.ptr, f.ctypes.data, f.size)
        tickets = [q.decompress(src.ptr, f.size, dst.ptr, n) for _ in range(3)]
        for t in tickets:
            assert q.wait(t) == n
        assert np.array_equal(np.frombuffer(dst.view, np.uint8), x)
        q.close(); src.close(); dst.close()
    # the batch entry point looks at its frames and picks the larger workspace itself
    L = hb.lib()
    outs = [np.zeros(n, np.uint8) for _ in range(2)]
    frames = (ctypes.c_void_p * 2)(f.ctypes.data, f.ctypes.data)
    sizes = (ctypes.c_size_t * 2)(f.size, f.size)
    dsts = (ctypes.c_void_p * 2)(outs[0].ctypes.data, outs[1].ctypes.data)
    caps = (ctypes.c_size_t * 2)(n, n)
    rcs = (ctypes.c_int64 * 2)()
    assert L.hb_decompress_frames_multi(2, frames, sizes, dsts, caps, rcs, 0) == 0
    assert list(rcs) == [n, n] and all(np.array_equal(o, x) for o in outs)


def test_random_valid_blocks_decode_exactly(hb, O):
    # blocks no encoder wrote: sequences drawn at random by tests/tools/lz4_stream_gen.py (offsets to 65535, every small period, runs and
    # literal runs of hundreds of KiB, sequences without literals ...); expected bytes: the oracle decoder's
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import lz4_stream_gen as G
    for seed, flags, ts in ((11, 0, 1), (12, 1, 4), (13, 4, 4), (14, 0, 1), (15, 1, 8)):
        rng = np.random.default_rng(seed)
        block, n = G.random_block(rng, (10 << 20) + seed * 4099, align=32 if flags == 4 else ts)
        frame = struct.pack("<BBBBIII", 2, hb.LZ4, flags, ts, n, n, 16 + len(block)) + block
        want = O.decompress_frame(np.frombuffer(frame, np.uint8)).tobytes()
        assert len(want) == n
        assert hb.Decompress(frame) == want, seed
        assert hb.lib().hb_last_result_flags() & 1, seed
        if not flags:
            assert hb.codecs[hb.LZ4].Decompress(block, n) == want, seed


def test_many_small_random_blocks(hb, O):
    # the same generator at sizes where every special path of the symbolic decoder sits close to the next one (regimes of 16-256 KiB:
    # image slides, sequences that bypass the image, sources just in front of / behind a unit's first byte, parked copies)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import lz4_stream_gen as G
    for seed in range(100, 124):
        rng = np.random.default_rng(seed)
        block, n = G.random_block(rng, (1 << 20) + seed * 4099, regime_len=(16 << 10) << (seed % 5))
        want = O.lz4_decompress(np.frombuffer(block, np.uint8), n).tobytes()
        assert len(want) == n
        if hb.indexless_parallel(len(block), n):
            frame = struct.pack("<BBBBIII", 2, hb.LZ4, 0, 1, n, n, 16 + len(block)) + block
            assert hb.Decompress(frame) == want, seed
            assert hb.lib().hb_last_result_flags() & 1, seed
        assert hb.codecs[hb.LZ4].Decompress(block, n) == want, seed


def test_unit_order_of_the_unfused_decoder_covers_ragged_multi_pass_frames(hb, O):
    # k_dec_indexed without a fused un-filter visits unit 8 * (k P mod m) + (x + k) mod 8 (hb_lz4_dec.hip): every unit exactly once also
    # with several passes per workgroup (more than 65536 units) and a unit count that is no multiple of 8 -- 300 MiB + 4 KiB + 13
    # bytes of already shuffled float32, no filter (76802 units), with the stored index and with the rebuilt one
    n = (300 << 20) + 4096 + 13
    x = np.ascontiguousarray(O.filter(O.OP_SHUFFLE, O.synth(O.D_F32, n // 4 + 1), 4)[:n])   # byte planes: compressible without a filter
    L = hb.lib()
    for opts in (hb.OPT_INDEX_TRAILER, 0):
        cap = L.hb_frame_bound(n) + (n // 4096 + 4) * 16 + 64
        out = np.empty(cap, np.uint8)
        c = L.hb_compress_frame(x.ctypes.data, n, out.ctypes.data, cap, hb.LZ4, 5, hb.NoShuffle, 1, opts, 0)
        assert c > 0
        back = np.zeros(n, np.uint8)
        assert L.hb_decompress_frame(out.ctypes.data, c, back.ctypes.data, n, 0, 0) == n
        assert L.hb_last_result_flags() & 1
        assert np.array_equal(back, x), opts


def test_long_length_extensions_do_not_derail_the_discovery(hb, O):
    # a match of 100 MiB has 400 KB of FF bytes behind its offset: dozens of regions of the discovery BEGIN inside that run.  Parsed "as if
    # a token started here" they used to report exits megabytes away, the first belief round took those for real, and the repair ran
    # out of rounds (the frame then went to the single wavefront: 0.8 GB/s at 1 GiB).  D-f64 byte-shuffled (six of its eight planes
    # compress 200:1) and the ramp (220:1 overall) as the reference writes them, 256 MiB each
    L = hb.lib()
    for name, x, ts in (("f64", O.synth(O.D_F64, (256 << 20) // 8), 8), ("ramp", O.synth(O.D_RAMP, (256 << 20) // 4), 4)):
        f = O.compress_frame(x, shuffle=1, typesize=ts)
        n = x.nbytes
        back = np.zeros(n, np.uint8)
        assert L.hb_decompress_frame(f.ctypes.data, f.size, back.ctypes.data, n, 0, 0) == n, name
        assert L.hb_last_result_flags() & 1, f"{name}: the discovery gave up, the single wavefront decoded"
        assert np.array_equal(back, x.view(np.uint8).reshape(-1)), name
