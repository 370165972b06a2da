"""CPU test of the multi-GPU path of bench.py: frames shard one per rank, no data-path collective; the only
communication is the barrier and the MAX-over-ranks of the elapsed time.  Runs with gloo, world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import oracle as O
    frames = bench.frames_of_rank(8, rank, world)
    host = bench.synth_host("f32", 1 << 16, frame=frames[0])
    f = O.compress_frame(host, shuffle=1, typesize=4)                     # the oracle stands in for the device here
    assert np.array_equal(O.decompress_frame(f), host)
    elapsed = bench.max_over_ranks(0.5 + rank, torch.device("cpu"))
    digest = int(np.frombuffer(host.tobytes(), np.uint32).sum() % (1 << 31))
    all_d = [None] * world
    dist.all_gather_object(all_d, (frames, digest, int(f.size)))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, elapsed, all_d, bench.aggregate_gbps(1 << 16, world, 3, elapsed)))


def test_frames_shard_across_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, elapsed, all_d, gbps in res:
        assert elapsed == pytest.approx(1.5)                               # MAX over ranks of (0.5, 1.5)
        assert gbps == pytest.approx(world * (1 << 16) * 3 / 1.5 / 1e9)    # whole-job aggregate
    frames0, frames1 = res[0][2][0][0], res[0][2][1][0]
    assert frames0 == [0, 2, 4, 6] and frames1 == [1, 3, 5, 7]             # frame k -> rank k mod G
    assert res[0][2][0][1] != res[0][2][1][1], "ranks must work on different frames"
