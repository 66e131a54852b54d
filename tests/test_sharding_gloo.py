"""CPU, world_size 2, gloo: the N>1 path (shard -> local packed buffer -> one all-gather -> unpack).
The local pipeline is replaced by a deterministic stand-in producing the packed layout, so the
partitioning, padding of ragged shards, the collective and the global re-ordering are what is tested."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_local(kmax):
    """frames (b,H,W) u8 -> packed int32 tensor exactly in infer_batch_device's layout.
    Corner k of a frame derives from the frame's pixel content only (so results identify frames)."""
    def run(frames):
        b = frames.shape[0]
        packed = np.zeros(b + b * kmax * 4 + b * kmax * 2, np.int32)
        rows = packed[b:b + b * kmax * 4].reshape(b, kmax, 4)
        xy = packed[b + b * kmax * 4:].view(np.float32).reshape(b, kmax, 2)
        for i in range(b):
            tag = int(frames[i, 0, 0])
            k = tag % (kmax + 1)
            packed[i] = k
            for j in range(k):
                rows[i, j] = [tag + j, 2 * tag + j, (tag * 7 + j * 3) % 16, j]
                xy[i, j] = [tag + j + 0.125, 2 * tag + j + 0.5]
        return torch.from_numpy(packed)
    return run


def _expected(frames, kmax):
    from deepcharuco_amd.inference import unpack_results
    packed = _fake_local(kmax)(frames).numpy()
    return unpack_results(packed, frames.shape[0], kmax, True)[0]


def _worker(rank, world, port, n_frames, kmax, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deepcharuco_amd.sharding import infer_frames_sharded
        frames = np.zeros((n_frames, 8, 8), np.uint8)
        frames[:, 0, 0] = (np.arange(n_frames) * 5 + 3) % 251
        res = infer_frames_sharded(frames, 16, None, refinenet=object(), kmax=kmax, run_local=_fake_local(kmax))
        exp = _expected(frames, kmax)
        ok = len(res) == n_frames and all(
            a.shape == e.shape and a.dtype == e.dtype and np.array_equal(a, e) for a, e in zip(res, exp))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n_frames, kmax, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, kmax, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(world))
    assert got == [(r, True) for r in range(world)]


def test_even_shards_world2():
    _run(n_frames=8, kmax=5)


def test_ragged_shards_world2():
    _run(n_frames=7, kmax=4)   # rank 0 gets 4 frames, rank 1 gets 3 (+1 padded)


def test_fewer_frames_than_ranks():
    _run(n_frames=1, kmax=3)   # rank 1 owns no frame at all
