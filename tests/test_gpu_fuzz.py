"""GPU robustness: hostile frames must never fault the device, and the device decoder must agree with the
restated reference decoder (oracle) on every one of them — same bytes, or the same error class.

Mirrors fuzz_test.go:11-160 (FuzzDecompress: never panic; if it succeeds, len == NBytesOrig) with a fixed seed.
The restart index is attacker-controlled too: mutations hit header, payload and index alike.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ERRS = {-1: "ErrInvalidData", -2: "ErrInvalidHeader", -3: "ErrInvalidVersion", -4: "ErrInvalidCodec",
        -5: "ErrSizeMismatch", -8: "ErrDecompressionFailed"}


def _oracle(O, frame):
    try:
        return ("ok", O.decompress_frame(np.frombuffer(frame, np.uint8)).tobytes())
    except O.OracleError as e:
        return (ERRS.get(e.code, str(e.code)), None)


def _device(hb, frame):
    try:
        return ("ok", hb.Decompress(frame))
    except hb.BloscError as e:
        return (type(e).__name__, None)


def test_mutated_frames_agree_with_the_reference_decoder(hb, O):
    rng = np.random.default_rng(20261003)
    bases = []
    for x, shuffle, ts in [(O.synth(O.D_F32, 9000), 1, 4), (O.synth(O.D_I32, 6000), 2, 4),
                           (np.tile(np.arange(97, dtype=np.uint8), 300), 0, 1),
                           (np.concatenate([rng.integers(0, 256, 9000, dtype=np.uint8), np.zeros(20000, np.uint8)]), 1, 8),
                           # whole planes of whole 4 KiB chunks: the decoder un-shuffles these itself (byte-strided stores)
                           (O.synth(O.D_F32, 4096 * 3), 1, 4), (O.synth(O.D_F64, 4096 * 2), 1, 8)]:
        for opts in (0, hb.OPT_INDEX_TRAILER):
            bases.append(hb.Compress(x.tobytes(), hb.LZ4, 5, shuffle, ts, opts=opts))
    checked = 0
    for f in bases:
        assert _device(hb, f) == _oracle(O, f)
        for _ in range(40):
            g = bytearray(f)
            kind = rng.integers(0, 5)
            if kind == 0:                                   # flip a few random bytes anywhere
                for pos in rng.integers(0, len(g), rng.integers(1, 4)):
                    g[pos] ^= int(rng.integers(1, 256))
            elif kind == 1:                                 # hit the payload
                pos = int(rng.integers(16, min(len(g), 16 + 4000)))
                g[pos] = int(rng.integers(0, 256))
            elif kind == 2:                                 # truncate
                g = g[: int(rng.integers(0, len(g)))]
            elif kind == 3:                                 # hit the tail (the index, when there is one)
                pos = int(rng.integers(max(16, len(g) - 600), len(g)))
                g[pos] ^= int(rng.integers(1, 256))
            else:                                           # tamper with header flags / typesize / sizes
                pos = int(rng.integers(2, 16))
                g[pos] ^= int(rng.integers(1, 256))
            g = bytes(g)
            want = _oracle(O, g)
            # a huge declared size is legal input for the reference (it just allocates); keep the test bounded
            if len(g) >= 16 and int.from_bytes(g[4:8], "little") > (64 << 20):
                continue
            got = _device(hb, g)
            assert got == want, f"device {got[0]} vs reference {want[0]} on a mutated frame (kind {kind})"
            checked += 1
    assert checked > 400


def test_frames_multi_entry_point(hb, O):
    # hb_compress_frames_multi: independent frames, frame k -> device k mod G
    import ctypes
    L = hb.lib()
    xs = [O.synth(O.D_F32, 50000, frame=k) for k in range(3)]
    n = len(xs)
    caps = [L.hb_frame_bound(x.size) for x in xs]
    outs = [ctypes.create_string_buffer(c) for c in caps]
    src = (ctypes.c_void_p * n)(*[x.ctypes.data for x in xs])
    dst = (ctypes.c_void_p * n)(*[ctypes.addressof(o) for o in outs])
    lens = (ctypes.c_size_t * n)(*[x.size for x in xs])
    cps = (ctypes.c_size_t * n)(*caps)
    rcs = (ctypes.c_int64 * n)()
    assert L.hb_compress_frames_multi(n, src, lens, dst, cps, rcs, hb.LZ4, 5, hb.Shuffle1, 4, hb.OPT_INDEX_TRAILER) == 0
    for k in range(n):
        assert rcs[k] > 16
        f = outs[k].raw[: rcs[k]]
        assert hb.Decompress(f) == xs[k].tobytes()
        assert np.array_equal(O.decompress_frame(np.frombuffer(f, np.uint8)), xs[k])
        assert f == hb.Compress(xs[k].tobytes(), hb.LZ4, 5, hb.Shuffle1, 4, opts=hb.OPT_INDEX_TRAILER)


def test_frames_multi_decompress_and_mixed_batches(hb, O):
    # hb_decompress_frames_multi (config 4 is compress AND decompress across GPUs): more frames than queue slots, device
    # frames with the index, ORACLE-written frames without it, a memcpy frame, and malformed frames whose per-frame rc must be
    # the reference's error (blosc.go:297-299, :180-182, :385-390, :403-407) while the other frames of the batch still decode.
    import ctypes
    L = hb.lib()
    rng = np.random.default_rng(31)
    xs = [O.synth(O.D_F32, 30000 + 4096 * k, frame=k).tobytes() for k in range(5)]
    xs.append(rng.integers(0, 256, 70000, dtype=np.uint8).tobytes())                      # incompressible -> memcpy frame
    xs.append(O.synth(O.D_I32, 65536, frame=9).tobytes())
    frames = [hb.Compress(x, hb.LZ4, 5, hb.Shuffle1, 4, opts=hb.OPT_INDEX_TRAILER) for x in xs[:3]]
    frames += [O.compress_frame(np.frombuffer(x, np.uint8), shuffle=1, typesize=4).tobytes() for x in xs[3:5]]   # reference-shaped
    frames.append(hb.Compress(xs[5], hb.LZ4, 5, hb.NoShuffle, 1))
    frames.append(hb.Compress(xs[6], hb.LZ4, 5, hb.BitShuffle, 4, opts=hb.OPT_INDEX_TRAILER))
    want = [(len(x), x) for x in xs]
    bad = bytearray(frames[0]); bad[0] = 9
    frames.append(bytes(bad)); want.append((-3, None))                                    # ErrInvalidVersion
    frames.append(frames[1][:10]); want.append((-2, None))                                # ErrInvalidHeader
    bad = bytearray(frames[2]); bad[12:16] = (len(frames[2]) + 100).to_bytes(4, "little")
    frames.append(bytes(bad)); want.append((-1, None))                                    # ErrInvalidData
    bad = bytearray(frames[3]); bad[1] = 200
    frames.append(bytes(bad)); want.append((-4, None))                                    # ErrInvalidCodec
    n = len(frames)
    keep = [np.frombuffer(f, np.uint8).copy() for f in frames]
    outs = [np.zeros(max(len(x), 16), np.uint8) for x in xs] + [np.zeros(1 << 17, np.uint8) for _ in range(n - len(xs))]
    fp = (ctypes.c_void_p * n)(*[k.ctypes.data for k in keep])
    fl = (ctypes.c_size_t * n)(*[k.size for k in keep])
    dp = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    dc = (ctypes.c_size_t * n)(*[o.size for o in outs])
    rcs = (ctypes.c_int64 * n)(*([-99] * n))
    assert L.hb_decompress_frames_multi(n, fp, fl, dp, dc, rcs, 0) == 0
    for k, (rc, x) in enumerate(want):
        assert rcs[k] == rc, (k, rcs[k], rc)
        if x is not None:
            assert outs[k][:rc].tobytes() == x, k
    # compress side: a batch with an empty frame in the middle (ErrInvalidData, blosc.go:269-271) and mixed sizes
    srcs = [np.frombuffer(x, np.uint8).copy() for x in xs[:4]]
    lens = [s.size for s in srcs]; lens[2] = 0
    caps = [L.hb_frame_bound(s.size) for s in srcs]
    couts = [np.zeros(c, np.uint8) for c in caps]
    m = len(srcs)
    rc2 = (ctypes.c_int64 * m)()
    assert L.hb_compress_frames_multi(m, (ctypes.c_void_p * m)(*[s.ctypes.data for s in srcs]), (ctypes.c_size_t * m)(*lens),
                                      (ctypes.c_void_p * m)(*[o.ctypes.data for o in couts]), (ctypes.c_size_t * m)(*caps), rc2,
                                      hb.LZ4, 5, hb.Shuffle1, 4, 0) == 0
    for k in range(m):
        if k == 2:
            assert rc2[k] == -1
        else:
            assert rc2[k] > 16 and O.decompress_frame(couts[k][:rc2[k]]).tobytes() == xs[k], k


def test_random_shapes_round_trip_and_decode_with_the_reference_decoder(hb, O):
    # seeded sweep over sizes (ragged, window-sized, multi-window, multi-tile), filters, typesizes and data textures:
    # every frame must decode on the device to the input AND through the restated reference decoder
    rng = np.random.default_rng(424242)
    sizes = [1, 2, 3, 15, 16, 17, 63, 64, 65, 255, 256, 4095, 4096, 4097, 8191, 12288, 16384 * 3, 65536 + 13, 4096 * 4 * 33,
             (1 << 20) + 4096 * 8, 4096 * 8 * 17]
    sizes += [int(s) for s in rng.integers(1, 3 << 20, 40)]
    checked = 0
    for n in sizes:
        kind = int(rng.integers(0, 5))
        if kind == 0:
            x = rng.integers(0, 256, n, dtype=np.uint8)                               # incompressible
        elif kind == 1:
            x = np.zeros(n, np.uint8); x[rng.integers(0, n, max(1, n // 500))] = 7      # sparse
        elif kind == 2:
            x = np.repeat(rng.integers(0, 256, n // 37 + 1, dtype=np.uint8), 37)[:n].copy()   # runs
        elif kind == 3:
            x = O.synth(O.D_F32, n // 4 + 1).view(np.uint8)[:n].copy()
        else:
            x = np.tile(rng.integers(0, 4, 1000, dtype=np.uint8) * 64, n // 1000 + 1)[:n].copy()   # few-valued noise, period 1000
        shuffle = int(rng.integers(0, 3))
        ts = int(rng.choice([1, 2, 3, 4, 8, 16]))
        opts = hb.OPT_INDEX_TRAILER if rng.integers(0, 4) else 0
        f = hb.Compress(x.tobytes(), hb.LZ4, 5, shuffle, ts, opts=opts)
        assert hb.Decompress(f) == x.tobytes(), (n, kind, shuffle, ts, opts)
        assert np.array_equal(O.decompress_frame(np.frombuffer(f, np.uint8)), x), (n, kind, shuffle, ts, opts)
        checked += 1
    assert checked == len(sizes)
