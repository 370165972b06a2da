"""GPU parity for SURVEY §8 row f1 "frame batches": many small frames through ONE set of launches.

The reference's own benchmark is a 100 000-byte frame (blosc_test.go:363-413); hb_compress_frames_batch / hb_decompress_frames_batch put
K independent Compress / Decompress calls (blosc.go:257-303) through the kernels of one frame, flattened.  Contract checked here: every
frame of a batch is BYTE-IDENTICAL to what the one-frame entry point writes for the same input and options, every frame decodes through
the oracle (the restated reference decoder) to its input, the batch decoder returns what the one-frame decoder returns -- bytes and error
identities -- for frames with and without the restart index, frames the oracle (= reference-shaped encoder) wrote, memcpy frames,
damaged frames; one bad frame never disturbs its neighbours.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(O, rng):
    xs = []
    xs.append(bytes(i % 256 for i in range(100000)))                              # the reference's benchmark frame, blosc_test.go:363-371
    xs.append(O.synth(O.D_RAMP, 25000).tobytes())                                 # f32 i*0.1, 100 000 B (blosc_test.go:109-111 shape)
    for k in range(6):
        xs.append(O.synth(O.D_F32, 16384 * (k + 1), frame=k).tobytes())           # whole element blocks: the fused shuffle path
    xs.append(O.synth(O.D_F32, 30001, frame=7).tobytes())                         # ragged: two-pass path
    xs.append(O.synth(O.D_I32, 8 * 4096 + 24, frame=8).tobytes())
    xs.append(rng.integers(0, 256, 70000, dtype=np.uint8).tobytes())              # incompressible: memcpy frame (blosc.go:342-345)
    xs.append(bytes(5000))                                                        # zeros
    xs.append(b"abc")                                                             # shorter than an element
    xs.append(bytes(range(13)))
    xs.append(O.synth(O.D_F64, 4096 * 3, frame=9).tobytes())
    return xs


@pytest.mark.parametrize("shuffle,ts", [(0, 1), (1, 4), (2, 4), (1, 8), (1, 2), (2, 8), (1, 3)])
@pytest.mark.parametrize("opts_name", ["default", "index"])
def test_batch_frames_are_the_one_call_frames(hb, O, shuffle, ts, opts_name):
    rng = np.random.default_rng(100 * shuffle + ts)
    xs = _inputs(O, rng)
    opts = hb.OPT_INDEX_TRAILER if opts_name == "index" else 0
    frames = hb.CompressBatch(xs, hb.LZ4, 5, shuffle, ts, opts=opts)
    assert len(frames) == len(xs)
    for i, (x, f) in enumerate(zip(xs, frames)):
        assert isinstance(f, bytes), (i, f)
        one = hb.Compress(x, hb.LZ4, 5, shuffle, ts, opts=opts)
        assert f == one, f"frame {i} ({len(x)} B): the batch wrote a different frame than the one-frame entry point"
        h = hb.GetInfo(f)
        assert h.NBytesOrig == len(x) and h.TypeSize == ts
        if not (h.IsMemcpy() and shuffle and len(x) >= ts):                       # (memcpy + filter: the reference's own defect)
            assert O.decompress_frame(np.frombuffer(f[:h.NBytesComp], np.uint8)).tobytes() == x, i
    back = hb.DecompressBatch(frames)
    for i, (x, b) in enumerate(zip(xs, back)):
        assert b == x, f"frame {i}: batch decode differs"


def test_all_fusable_batch_and_hc(hb, O):
    # every frame whole element blocks: the filter is fused into the matcher for the whole batch; LZ4HC levels ride the same launches
    xs = [O.synth(O.D_F32, 16384 * (1 + k % 5), frame=k).tobytes() for k in range(40)]
    for codec, level in ((hb.LZ4, 5), (hb.LZ4, 1), (hb.LZ4HC, 5), (hb.LZ4HC, 9)):
        for opts in (0, hb.OPT_INDEX_TRAILER):
            frames = hb.CompressBatch(xs, codec, level, hb.Shuffle1, 4, opts=opts)
            for k, (x, f) in enumerate(zip(xs, frames)):
                assert f == hb.Compress(x, codec, level, hb.Shuffle1, 4, opts=opts), (codec, level, opts, k)
            assert O.decompress_frame(np.frombuffer(frames[3], np.uint8)).tobytes() == xs[3]
            assert hb.DecompressBatch(frames) == xs


def test_batch_decodes_what_others_wrote_and_reports_like_the_one_call_decoder(hb, O):
    rng = np.random.default_rng(5)
    xs = _inputs(O, rng)
    frames = []
    for i, x in enumerate(xs):
        a = np.frombuffer(x, np.uint8)
        # frames of the oracle's reference-shaped encoder (one block, 64 KiB window, no index): Shuffle1 ts 4, NoShuffle, BitShuffle ts 8
        frames.append(O.compress_frame(a, shuffle=(1, 0, 2)[i % 3], typesize=(4, 1, 8)[i % 3]).tobytes())
    # damaged frames in between: what the reference's tests damage (blosc_test.go:593-611 payload xor, codec_test.go:60-79 nbytes, cbytes)
    good = hb.Compress(xs[2], hb.LZ4, 5, hb.Shuffle1, 4, opts=hb.OPT_INDEX_TRAILER)
    bad1 = bytearray(good); bad1[16:200] = bytes(b ^ 0xFF for b in bad1[16:200])
    bad2 = bytearray(good); bad2[4:8] = (len(xs[2]) * 2).to_bytes(4, "little"); bad2[8:12] = bad2[4:8]
    bad3 = bytearray(good); bad3[12:16] = (len(good) + 100).to_bytes(4, "little")
    bad4 = bytearray(good); bad4[0] = 7
    bad5 = bytes([2, 1, 0, 4]) + (1000).to_bytes(4, "little") * 2 + (20).to_bytes(4, "little") + b"\xff\xff\xff\xff"   # codec_test.go:276-284
    frames[3:3] = [bytes(bad1), bytes(bad2)]
    frames += [bytes(bad3), bytes(bad4), bad5, good[:10], good]
    got = hb.DecompressBatch(frames)
    assert len(got) == len(frames)
    for i, f in enumerate(frames):
        try:
            want = hb.Decompress(f)
        except hb.BloscError as e:
            want = type(e)
        if isinstance(want, type):
            assert isinstance(got[i], want), (i, got[i], want)
            try:                                                                  # ... and the oracle (the restated reference) agrees on the class
                O.decompress_frame(np.frombuffer(f, np.uint8))
                assert False, i
            except Exception:                                                     # noqa: BLE001
                pass
        else:
            assert got[i] == want, i


def test_default_shape_frames_of_a_batch_get_their_index_rebuilt(hb, O):
    # round 4: frames of a batch that end at NBytesComp (what Compress returns by default, blosc.go:369-371) and are large enough for the token discovery
    # (>= 256 KiB of payload) get their restart index rebuilt for the whole batch in one set of launches (hb_lz4_region.hip `_b` kernels) and decode
    # chunk-parallel; frames somebody else wrote (one block, 64 KiB window) and damaged frames fall through to the stream decoder / the authority and
    # must report exactly what the one-call decoder reports
    rng = np.random.default_rng(11)
    n = 3 << 19                                                            # 1.5 MiB
    xs = [O.synth(O.D_F32, n // 4, frame=1).tobytes(), O.synth(O.D_I32, n // 4).tobytes(), O.synth(O.D_F64, n // 8).tobytes(),
          O.synth(O.D_RAND, n // 4).tobytes(), rng.integers(0, 7, n, dtype=np.uint8).tobytes(), O.synth(O.D_RAMP, n // 4).tobytes()]
    cfg = [(hb.Shuffle1, 4), (hb.BitShuffle, 4), (hb.Shuffle1, 8), (hb.Shuffle1, 4), (hb.NoShuffle, 1), (hb.Shuffle1, 2)]
    frames = [hb.Compress(x, hb.LZ4, 5, sh, ts, opts=0) for x, (sh, ts) in zip(xs, cfg)]
    for f, x in zip(frames, xs):
        assert len(f) == hb.GetInfo(f).NBytesComp                           # no trailer
    foreign = O.compress_frame(np.frombuffer(xs[0], np.uint8), shuffle=1, typesize=4).tobytes()
    bad = bytearray(frames[0]); bad[5000:5040] = bytes(b ^ 0x5A for b in bad[5000:5040])
    trunc = bytearray(frames[1]); trunc[12:16] = (len(trunc) - 4000).to_bytes(4, "little")      # a shorter NBytesComp: the stream ends early
    batch = frames + [foreign, bytes(bad), bytes(trunc[: len(trunc) - 4000])] + frames[::-1]
    got = hb.DecompressBatch(batch)
    for i, f in enumerate(batch):
        try:
            want = hb.Decompress(f)
        except hb.BloscError as e:
            want = type(e)
        if isinstance(want, type):
            assert isinstance(got[i], want), (i, got[i], want)
        else:
            assert got[i] == want, i
    assert got[: len(xs)] == xs and got[len(xs)] == xs[0]


def test_large_batch_of_the_reference_benchmark_frame(hb, O):
    # 512 x the 100 000-byte byte(i % 256) frame of blosc_test.go:363-371 + 512 x a float frame: every frame equal to the one-call frame
    a = bytes(i % 256 for i in range(100000))
    b = O.synth(O.D_F32, 25000, frame=3).tobytes()
    xs = [a, b] * 512
    fa, fb = hb.Compress(a, hb.LZ4, 5, hb.Shuffle1, 4), hb.Compress(b, hb.LZ4, 5, hb.Shuffle1, 4)
    frames = hb.CompressBatch(xs, hb.LZ4, 5, hb.Shuffle1, 4)
    assert all(f == (fa, fb)[i & 1] for i, f in enumerate(frames))
    assert O.decompress_frame(np.frombuffer(fa, np.uint8)).tobytes() == a and O.decompress_frame(np.frombuffer(fb, np.uint8)).tobytes() == b
    back = hb.DecompressBatch(frames)
    assert all(x == (a, b)[i & 1] for i, x in enumerate(back))


def test_batch_argument_errors(hb, O):
    # empty input is ErrInvalidData for that frame only (blosc.go:269-271), like the one-call API
    xs = [b"hello world, hello world, hello world", b"", O.synth(O.D_F32, 5000).tobytes()]
    fr = hb.CompressBatch(xs, hb.LZ4, 5, hb.Shuffle1, 4)
    assert isinstance(fr[1], hb.ErrInvalidData) and fr[0] == hb.Compress(xs[0], hb.LZ4, 5, hb.Shuffle1, 4) and fr[2] == hb.Compress(xs[2], hb.LZ4, 5, hb.Shuffle1, 4)
    # Snappy frames are not carried by the batch kernels: answered by the one-frame entry point, same bytes
    sn = hb.CompressBatch([xs[0], xs[2]], hb.Snappy, 5, hb.Shuffle1, 4)
    assert sn == [hb.Compress(xs[0], hb.Snappy, 5, hb.Shuffle1, 4), hb.Compress(xs[2], hb.Snappy, 5, hb.Shuffle1, 4)]
    assert hb.DecompressBatch(sn + [fr[0]]) == [xs[0], xs[2], xs[0]]
    assert hb.CompressBatch([]) == [] and hb.DecompressBatch([]) == []


def test_random_blocks_and_mutations_in_one_batch(hb, O):
    # blocks no encoder wrote (tests/tools/lz4_stream_gen.py: offsets to 65535, every small period, long runs and literal runs, sequences without
    # literals) as frames with every filter flag, next to mutated copies: the batch decoder must return, frame by frame, what the one-frame
    # decoder returns -- and that one is held to the oracle (test_gpu_foreign.py, test_gpu_fuzz.py)
    import os
    import struct
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import lz4_stream_gen as G
    rng = np.random.default_rng(77)
    frames = []
    for k in range(40):
        flags, ts = ((0, 1), (1, 4), (4, 4), (1, 8), (1, 2), (4, 8))[k % 6]
        block, n = G.random_block(rng, int(rng.integers(2000, 300000)), regime_len=(4 << 10) << (k % 5), align=32 if flags == 4 else ts)
        f = struct.pack("<BBBBIII", 2, hb.LZ4 if k % 3 else hb.LZ4HC, flags, ts, n, n, 16 + len(block)) + block
        frames.append(f)
        if k % 4 == 0:                                                            # damaged copies: payload bit flips, wrong sizes, truncation
            g = bytearray(f); p = int(rng.integers(16, len(f))); g[p] ^= 1 << int(rng.integers(0, 8)); frames.append(bytes(g))
            g = bytearray(f); g[4:8] = (n + int(rng.integers(1, 5000))).to_bytes(4, "little"); frames.append(bytes(g))
            frames.append(f[: 16 + (len(f) - 16) // 2][:12] + (16 + (len(f) - 16) // 2).to_bytes(4, "little") + f[16: 16 + (len(f) - 16) // 2])
    got = hb.DecompressBatch(frames)
    agree_err = 0
    for i, f in enumerate(frames):
        try:
            want = hb.Decompress(f)
        except hb.BloscError as e:
            want = type(e)
        if isinstance(want, type):
            assert isinstance(got[i], want), (i, got[i], want)
            agree_err += 1
        else:
            assert got[i] == want, i
    assert agree_err >= 5
    # one frame's oracle check per filter kind: the one-frame decoder above is not the only witness
    for i in (0, 1, 2):
        f = [x for x in frames if x[2] == (0, 1, 4)[i]][0]
        assert got[frames.index(f)] == O.decompress_frame(np.frombuffer(f, np.uint8)).tobytes()


def test_frames_multi_batches_its_small_frames(hb, O):
    # hb_compress_frames_multi / hb_decompress_frames_multi send the small frames of every device's share through the batch kernels and
    # the large ones through the device's queue: 200 frames of 20 KB - 1.2 MiB around two frames of 6 MiB -- same bytes as one call each
    import ctypes
    L = hb.lib()
    rng = np.random.default_rng(9)
    xs = [O.synth(O.D_F32, int(rng.integers(5000, 300000)), frame=k) for k in range(200)]
    xs.insert(17, O.synth(O.D_F32, 6 << 18, frame=300)); xs.insert(120, O.synth(O.D_I32, 6 << 18, frame=301))
    xs.append(rng.integers(0, 256, 50000, dtype=np.uint8))                                 # -> memcpy frame
    m = len(xs)
    caps = [L.hb_frame_bound(x.size) for x in xs]
    outs = [np.empty(c, np.uint8) for c in caps]
    vp, sz, i64 = ctypes.c_void_p * m, ctypes.c_size_t * m, ctypes.c_int64 * m
    rcs = i64()
    assert L.hb_compress_frames_multi(m, vp(*[x.ctypes.data for x in xs]), sz(*[x.size for x in xs]), vp(*[o.ctypes.data for o in outs]), sz(*caps), rcs,
                                      hb.LZ4, 5, hb.Shuffle1, 4, hb.OPT_INDEX_TRAILER) == 0
    frames = []
    for k in range(m):
        assert rcs[k] > 16, (k, rcs[k])
        f = outs[k][: rcs[k]].tobytes()
        if k % 9 == 0 or xs[k].size > (4 << 20):
            assert f == hb.Compress(xs[k].tobytes(), hb.LZ4, 5, hb.Shuffle1, 4, opts=hb.OPT_INDEX_TRAILER), k
            h = hb.GetInfo(f)
            if not h.IsMemcpy():
                assert np.array_equal(O.decompress_frame(np.frombuffer(f, np.uint8)), xs[k]), k
        frames.append(np.frombuffer(f, np.uint8).copy())
    # a foreign (oracle-written) frame and a damaged one among them on the way back
    frames[5] = O.compress_frame(xs[5], shuffle=1, typesize=4)
    bad = frames[6].copy(); bad[0] = 9
    backs = [np.zeros(max(x.size, 16), np.uint8) for x in xs]
    rcd = i64()
    fr2 = list(frames); fr2[6] = bad
    assert L.hb_decompress_frames_multi(m, vp(*[f.ctypes.data for f in fr2]), sz(*[f.size for f in fr2]), vp(*[b.ctypes.data for b in backs]),
                                        sz(*[b.size for b in backs]), rcd, 0) == 0
    for k in range(m):
        if k == 6:
            assert rcd[k] == -3, rcd[k]                                                    # ErrInvalidVersion for that frame only
        else:
            assert rcd[k] == xs[k].size and np.array_equal(backs[k][: xs[k].size], xs[k]), (k, rcd[k])


def test_adjacent_host_buffers_take_the_one_copy_paths(hb, O):
    # inputs that follow each other exactly in host memory go up in ONE copy, many small frames come down packed in ONE copy (compress),
    # results that follow each other inside their capacities come down in ONE copy (decompress): same frames, same bytes, and a failed
    # frame in the batch switches the download back to one copy per frame without touching its neighbours' results
    import ctypes
    L = hb.lib()
    K, n = 64, 50000
    xs = [O.synth(O.D_F32, n // 4, frame=k) for k in range(K)]
    slab = np.concatenate(xs)                                              # exactly adjacent inputs
    cap = L.hb_frame_bound(n)
    out = np.zeros(K * cap, np.uint8)
    vp, sz, i64 = ctypes.c_void_p * K, ctypes.c_size_t * K, ctypes.c_int64 * K
    rcs = i64()
    assert L.hb_compress_frames_batch(K, vp(*[slab.ctypes.data + k * n for k in range(K)]), sz(*[n] * K), vp(*[out.ctypes.data + k * cap for k in range(K)]),
                                      sz(*[cap] * K), rcs, hb.LZ4, 5, hb.Shuffle1, 4, 0, 0) == 0
    frames = []
    for k in range(K):
        f = out[k * cap: k * cap + rcs[k]].tobytes()
        assert f == hb.Compress(xs[k].tobytes(), hb.LZ4, 5, hb.Shuffle1, 4, opts=0), k
        frames.append(f)
    fslab = np.frombuffer(b"".join(frames), np.uint8).copy()               # exactly adjacent frames
    offs = np.cumsum([0] + [len(f) for f in frames])
    room = n + 24                                                           # results 24 bytes apart inside their capacities
    back = np.full(K * room, 0xAB, np.uint8)
    rcd = i64()
    assert L.hb_decompress_frames_batch(K, vp(*[fslab.ctypes.data + int(offs[k]) for k in range(K)]), sz(*[len(f) for f in frames]),
                                        vp(*[back.ctypes.data + k * room for k in range(K)]), sz(*[room] * K), rcd, 0, 0) == 0
    for k in range(K):
        assert rcd[k] == n and back[k * room: k * room + n].tobytes() == xs[k].tobytes(), k
    # one damaged frame among them
    bad = fslab.copy(); bad[int(offs[7]) + 30: int(offs[7]) + 60] ^= 0xFF
    back[:] = 0xCD
    assert L.hb_decompress_frames_batch(K, vp(*[bad.ctypes.data + int(offs[k]) for k in range(K)]), sz(*[len(f) for f in frames]),
                                        vp(*[back.ctypes.data + k * room for k in range(K)]), sz(*[room] * K), rcd, 0, 0) == 0
    try:
        want7 = hb.Decompress(bad[int(offs[7]): int(offs[8])].tobytes())
    except hb.BloscError as e:
        want7 = e.code
    for k in range(K):
        if k == 7:
            assert (rcd[k] == want7) if isinstance(want7, int) else (rcd[k] == n and back[k * room: k * room + n].tobytes() == want7)
        else:
            assert rcd[k] == n and back[k * room: k * room + n].tobytes() == xs[k].tobytes(), k
