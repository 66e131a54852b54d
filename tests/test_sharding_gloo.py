"""CPU, world_size 2, gloo: the N>1 path (shard -> local packed buffer -> one all-gather -> unpack).
The local pipeline is replaced by a deterministic stand-in producing the packed layout, so the
partitioning, padding of ragged shards, the collective and the global re-ordering are what is tested."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_local(kcap):
    """(frames (b,H,W) u8, pool) -> packed int32 tensor exactly in infer_batch_device's layout (counts | starts | rows | xy).
    Corner k of a frame derives from the frame's pixel content only (so results identify frames); a frame has tag % (kcap + 1)
    corners; frames are placed in the pool in REVERSE frame order (the real pipeline places them in completion order) and rows
    beyond the pool are dropped, as the kernels do."""
    def run(frames, pool):
        b = frames.shape[0]
        packed = np.zeros(2 * b + 6 * pool, np.int32)
        rows = packed[2 * b:2 * b + 4 * pool].reshape(pool, 4)
        xy = packed[2 * b + 4 * pool:].view(np.float32).reshape(pool, 2)
        cursor = 0
        for i in reversed(range(b)):
            tag = int(frames[i, 0, 0])
            k = tag % (kcap + 1)
            packed[i] = k
            packed[b + i] = cursor
            for j in range(k):
                if cursor + j < pool:
                    rows[cursor + j] = [tag + j, 2 * tag + j, (tag * 7 + j * 3) % 16, j]
                    xy[cursor + j] = [tag + j + 0.125, 2 * tag + j + 0.5]
            cursor += k
        return torch.from_numpy(packed)
    return run


def _expected(frames, kcap):
    from deepcharuco_amd.inference import unpack_results
    pool = max(1, frames.shape[0] * kcap)
    packed = _fake_local(kcap)(frames, pool).numpy()
    return unpack_results(packed, frames.shape[0], pool, True)[0]


def _worker(rank, world, port, n_frames, kmax, kcap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deepcharuco_amd.sharding import infer_frames_sharded
        frames = np.zeros((n_frames, 8, 8), np.uint8)
        frames[:, 0, 0] = (np.arange(n_frames) * 5 + 3) % 251
        calls = []
        fake = _fake_local(kcap)

        def run_local(fr, pool):
            calls.append(pool)
            return fake(fr, pool)
        res = infer_frames_sharded(frames, 16, None, refinenet=object(), kmax=kmax, run_local=run_local)
        exp = _expected(frames, kcap)
        ok = len(res) == n_frames and all(
            a is not None and a.shape == e.shape and a.dtype == e.dtype and np.array_equal(a, e) for a, e in zip(res, exp))
        q.put((rank, ok, len(calls)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n_frames, kmax, world=2, kcap=None, passes=1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    kcap = kmax if kcap is None else kcap
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, kmax, kcap, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(world))
    assert [(r, ok) for r, ok, _ in got] == [(r, True) for r in range(world)]
    # ranks that own frames ran the local pipeline `passes` times (a rank without frames never calls it)
    assert all(n in (0, passes) for _, _, n in got) and any(n == passes for _, _, n in got)


def test_even_shards_world2():
    _run(n_frames=8, kmax=5)


def test_ragged_shards_world2():
    _run(n_frames=7, kmax=4)   # rank 0 gets 4 frames, rank 1 gets 3 (+1 padded)


def test_fewer_frames_than_ranks():
    _run(n_frames=1, kmax=3)   # rank 1 owns no frame at all


def test_one_frame_may_use_most_of_the_pool():
    """No per-frame cap: corner counts 3 8 0 5 | 10 2 7 12 in pools sized for 8 per frame on average (32 slots per rank) -- frames
    with 10 and 12 corners, one pass, nothing dropped."""
    _run(n_frames=8, kmax=8, kcap=12)


def test_pool_overflow_is_repeated_collectively():
    """The same frames with pools of 4 x 5 = 20 slots: rank 1's shard fires 31 cells.  Every rank sees that in the gathered counts
    and both run the batch a second time with a pool of 31 -- complete results, no exception, no truncation."""
    _run(n_frames=8, kmax=5, kcap=12, passes=2)
