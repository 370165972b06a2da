#!/usr/bin/env python3
"""PCIe-inclusive (host to host) rates quoted in DESIGN.md / INTEGRATION.md, reproducible from tracked files.

  python tools/host_rates.py one-call     1 GiB D-f32 frames through hb_compress_frame / hb_decompress_frame, pageable and pinned
  python tools/host_rates.py queue        the same frames through hb_queue_* (3 in flight) next to the one-call API
  python tools/host_rates.py small        small frames (1-16 MiB) through hb_queue_* at several depths (launch-latency regime)
  python tools/host_rates.py config5      BASELINE.json config 5: 1 GiB D-f32, device Shuffle1 overlapped with host ZSTD level 3
  python tools/host_rates.py batch        1024 x 1 MiB and 4096 x 100 000 B frames through hb_*_frames_batch and hb_*_frames_multi (host to host)
  python tools/host_rates.py multi        hb_compress_frames_multi / hb_decompress_frames_multi on 8 x 256 MiB frames (all visible GPUs)

These are never the `value` of bench.py (device-resident by contract); `bench.py --host` embeds the first two.
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np

import hipblosc as hb
import oracle as O          # data generation + the final equality check only


def arr(p, k):
    return np.ctypeslib.as_array((ctypes.c_uint8 * k).from_address(p))


def one_call(L):
    n = 1 << 30
    x = O.synth(O.D_F32, n // 4)
    cap = L.hb_frame_bound(n)

    def run(src, dst, bk, label):
        for _ in range(3):
            t0 = time.perf_counter()
            c = L.hb_compress_frame(src.ctypes.data, n, dst.ctypes.data, cap, hb.LZ4, 5, hb.Shuffle1, 4, hb.OPT_INDEX_TRAILER, 0)
            t1 = time.perf_counter()
            d = L.hb_decompress_frame(dst.ctypes.data, c, bk.ctypes.data, n, 0, 0)
            t2 = time.perf_counter()
            assert d == n
        assert np.array_equal(bk[:n], src[:n])
        print(f"{label} compress {n / (t1 - t0) / 1e9:.2f} GB/s, decompress {n / (t2 - t1) / 1e9:.2f} GB/s (host->host, includes PCIe)", flush=True)
    run(x, np.empty(cap, np.uint8), np.empty(n, np.uint8), "pageable")
    ps, pd, pb = hb.PinnedBuffer(n), hb.PinnedBuffer(cap), hb.PinnedBuffer(n)
    xs = arr(ps.ptr, n); xs[:] = x
    run(xs, arr(pd.ptr, cap), arr(pb.ptr, n), "pinned  ")


def queue(L, sizes=((1024, 3, 8),)):
    for mib, depth, nf in sizes:
        n = mib << 20
        x = O.synth(O.D_F32, n // 4)
        cap = L.hb_frame_bound(n)
        pin_in = [hb.PinnedBuffer(n) for _ in range(depth)]
        pin_out = [hb.PinnedBuffer(cap) for _ in range(depth)]
        back = [hb.PinnedBuffer(n) for _ in range(depth)]
        for b in pin_in:
            ctypes.memmove(b.ptr, x.ctypes.data, n)
        q = hb.FrameQueue(n, depth=depth)
        c = 0
        for what in ("compress", "decompress"):
            for rep in range(2):
                t0 = time.perf_counter(); tk = []
                for k in range(nf):
                    s = k % depth
                    if what == "compress":
                        tk.append(q.compress(pin_in[s].ptr, n, pin_out[s].ptr, cap, hb.LZ4, 5, hb.Shuffle1, 4, hb.OPT_INDEX_TRAILER))
                    else:
                        tk.append(q.decompress(pin_out[s].ptr, c, back[s].ptr, n))
                    if k >= depth - 1:
                        r = q.wait(tk[k - depth + 1])
                for k in range(max(0, nf - depth + 1), nf):
                    r = q.wait(tk[k])
                dt = time.perf_counter() - t0
            if what == "compress":
                c = r
            print(f"frames of {mib} MiB, {depth} in flight: queue {what} {nf * n / dt / 1e9:.2f} GB/s ({dt / nf * 1e6:.0f} us per frame)", flush=True)
        assert np.array_equal(arr(back[0].ptr, n), x)
        q.close()


def config5(L):
    n = 1 << 30
    x = O.synth(O.D_F32, n // 4)
    cap = L.hb_frame_bound(n)
    out, back = np.empty(cap, np.uint8), np.empty(n, np.uint8)
    for _ in range(3):
        t0 = time.perf_counter(); c = L.hb_compress_frame(x.ctypes.data, n, out.ctypes.data, cap, hb.ZSTD, 3, hb.Shuffle1, 4, 0, 0)
        t1 = time.perf_counter(); d = L.hb_decompress_frame(out.ctypes.data, c, back.ctypes.data, n, 0, 0); t2 = time.perf_counter()
        assert d == n and np.array_equal(back, x)
        print(f"cfg5 1 GiB f32 Shuffle1+ZSTD L3: ratio {c / n:.4f} compress {n / (t1 - t0) / 1e9:.2f} GB/s decompress {n / (t2 - t1) / 1e9:.2f} GB/s (host->host)", flush=True)


def multi(L):
    nf, n = 8, 256 << 20
    xs = [O.synth(O.D_I32, n // 4, frame=k) for k in range(nf)]
    cap = L.hb_frame_bound(n)
    pin_in = [hb.PinnedBuffer(n) for _ in range(nf)]
    pin_out = [hb.PinnedBuffer(cap) for _ in range(nf)]
    for b, x in zip(pin_in, xs):
        ctypes.memmove(b.ptr, x.ctypes.data, n)
    vp, sz = ctypes.c_void_p * nf, ctypes.c_size_t * nf
    rcs = (ctypes.c_int64 * nf)()
    for _ in range(2):
        t0 = time.perf_counter()
        L.hb_compress_frames_multi(nf, vp(*[b.ptr for b in pin_in]), sz(*[n] * nf), vp(*[b.ptr for b in pin_out]), sz(*[cap] * nf), rcs,
                                   hb.LZ4, 5, hb.BitShuffle, 4, hb.OPT_INDEX_TRAILER)
        t1 = time.perf_counter()
    cs = list(rcs)
    assert min(cs) > 16
    for _ in range(2):
        t2 = time.perf_counter()
        L.hb_decompress_frames_multi(nf, vp(*[b.ptr for b in pin_out]), sz(*cs), vp(*[b.ptr for b in pin_in]), sz(*[n] * nf), rcs, 0)
        t3 = time.perf_counter()
    assert list(rcs) == [n] * nf and all(np.array_equal(arr(b.ptr, n), x) for b, x in zip(pin_in, xs))
    print(f"{nf} frames of {n >> 20} MiB i32 BitShuffle+LZ4 over {L.hb_device_count()} device(s): frames_multi compress "
          f"{nf * n / (t1 - t0) / 1e9:.2f} GB/s, decompress {nf * n / (t3 - t2) / 1e9:.2f} GB/s (host->host)", flush=True)


def batch(L):
    """Small frames host to host through the batch entry points (one set of launches per call) and through hb_*_frames_multi, which
    batches the small frames of every device's share itself; pinned buffers."""
    for nf, n, label in ((1024, 1 << 20, "1 MiB D-f32"), (4096, 100000, "100 000 B byte(i % 256) (blosc_test.go:363-371)")):
        if n == 100000:
            x = np.frombuffer(bytes(i % 256 for i in range(n)), np.uint8)
            xs = [x] * nf
        else:
            big = O.synth(O.D_F32, nf * n // 4)
            xs = [big[k * n:(k + 1) * n] for k in range(nf)]
        cap = L.hb_frame_bound(n)
        slab_in, slab_out = hb.PinnedBuffer(nf * n), hb.PinnedBuffer(nf * ((cap + 63) & ~63))
        for k, x in enumerate(xs):
            ctypes.memmove(slab_in.ptr + k * n, x.ctypes.data, n)
        vp, sz = ctypes.c_void_p * nf, ctypes.c_size_t * nf
        src = vp(*[slab_in.ptr + k * n for k in range(nf)])
        dst = vp(*[slab_out.ptr + k * ((cap + 63) & ~63) for k in range(nf)])
        rcs = (ctypes.c_int64 * nf)()
        for name, comp, dec in (("frames_batch", lambda: L.hb_compress_frames_batch(nf, src, sz(*[n] * nf), dst, sz(*[cap] * nf), rcs, hb.LZ4, 5, hb.Shuffle1, 4, 0, 0),
                                 lambda cs: L.hb_decompress_frames_batch(nf, dst, sz(*cs), src, sz(*[n] * nf), rcs, 0, 0)),
                                ("frames_multi", lambda: L.hb_compress_frames_multi(nf, src, sz(*[n] * nf), dst, sz(*[cap] * nf), rcs, hb.LZ4, 5, hb.Shuffle1, 4, 0),
                                 lambda cs: L.hb_decompress_frames_multi(nf, dst, sz(*cs), src, sz(*[n] * nf), rcs, 0))):
            for _ in range(2):
                t0 = time.perf_counter(); assert comp() == 0; t1 = time.perf_counter()
            cs = list(rcs)
            assert min(cs) > 16
            for _ in range(2):
                t2 = time.perf_counter(); assert dec(cs) == 0; t3 = time.perf_counter()
            assert list(rcs) == [n] * nf and all(np.array_equal(arr(slab_in.ptr + k * n, n), xs[k]) for k in (0, nf // 2, nf - 1))
            print(f"{nf} frames of {label}, default frame shape, {name}: compress {nf * n / (t1 - t0) / 1e9:.2f} GB/s, decompress {nf * n / (t3 - t2) / 1e9:.2f} GB/s "
                  f"(host->host, ratio {sum(cs) / (nf * n):.3f})", flush=True)
        slab_in.close(); slab_out.close()


if __name__ == "__main__":
    L = hb.lib()
    assert L.hb_init() == 0, "needs a HIP device (no CPU fallback)"
    what = sys.argv[1] if len(sys.argv) > 1 else "queue"
    {"one-call": one_call, "queue": queue, "config5": config5, "multi": multi, "batch": batch,
     "small": lambda L: queue(L, ((1, 1, 256), (1, 8, 256), (1, 16, 256), (4, 8, 128), (16, 4, 64)))}[what](L)
