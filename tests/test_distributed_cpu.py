"""N>1 path on CPU: two gloo ranks shard independent streams (stream s -> rank s % world), each
processes its own streams (here with the oracle standing in for the GPU -- this test covers the
distribution plumbing of dist_util.py, not the kernels), and the control collectives used by
bench.py (MAX of elapsed, SUM of sample counts / checksums) agree with a single-process run."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

import dist_util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream_job(s, frames=3000):
    import oracle as orc
    x = orc.lcg_pcm(frames * 2, 12345 + s).reshape(frames, 2)
    out, used = orc.Oracle(2, 44100, 48000, 3).process(x, 1 << 20)
    return out.shape[0], int(out.astype(np.int64).sum())


def _worker(rank, world, port, total_streams, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    w, r, _ = dist_util.init("gloo")
    assert (w, r) == (world, rank)
    dev = torch.device("cpu")
    mine = dist_util.shard_streams(total_streams, w, r)
    frames = checksum = 0
    for s in mine:
        n, c = _stream_job(s)
        frames += n
        checksum += c
    dist_util.barrier()
    slowest = dist_util.reduce_scalar(1.0 + rank, "max", dev)
    total_frames = dist_util.reduce_int(frames, dev)
    total_sum = dist_util.reduce_int(checksum, dev)
    q.put((rank, mine, slowest, total_frames, total_sum))
    dist_util.finish()


def test_two_rank_stream_sharding_over_gloo():
    total_streams, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = [_stream_job(s) for s in range(total_streams)]
    want_frames, want_sum = sum(a for a, _ in single), sum(b for _, b in single)
    owned = sorted(s for _, mine, _, _, _ in results for s in mine)
    assert owned == list(range(total_streams))           # every stream exactly once
    for rank, mine, slowest, frames, csum in results:
        assert mine == [s for s in range(total_streams) if s % world == rank]
        assert slowest == 2.0 and frames == want_frames and csum == want_sum


def test_shard_streams_is_a_partition():
    for total in (0, 1, 7, 256):
        for world in (1, 2, 4, 8):
            parts = [dist_util.shard_streams(total, world, r) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(total))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
