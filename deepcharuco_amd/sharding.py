"""Frame sharding across the GPUs of one node + the single exchange step of the path.

The reference has no multi-device code; frames are independent
(inference.py:32-70 keeps no cross-frame state, BN is in eval mode), so the path shards
embarrassingly: rank r of R owns frames [lo, hi) and runs the whole detect+refine pipeline on
them.  The only collective is one ``all_gather_into_tensor`` of the fixed-shape packed corner
buffer per batch (RCCL over xGMI with backend "nccl"; "gloo" for the CPU tests).  The payload
is KB-scale (B/R * (2 + 6*kmax) int32), i.e. latency-bound, so it is sent as ONE fused buffer.

Every rank's buffer is a corner POOL of ``pool = ceil(B/R) * kmax`` slots (inference.infer_batch_device): a frame may fire any
number of cells; only a rank whose whole shard fires more than its pool overflows, and then ALL ranks (they all see all counts
after the gather) run that batch again with a pool of the size the first pass reported -- nothing is truncated, nothing raises.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .inference import packed_len, unpack_results  # noqa: F401  (packed_len is re-exported: callers size buffers with it)


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first ``n_frames % world`` ranks get one extra frame."""
    if not (0 <= rank < world):
        raise ValueError("bad rank")
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pad_packed(packed: torch.Tensor, batch: int, pool: int, batch_max: int) -> torch.Tensor:
    """Re-lay a rank's packed buffer [counts|starts|rows|xy] for ``batch`` frames into the layout for ``batch_max`` frames with
    the same pool (missing frames get count 0) so every rank contributes the same shape."""
    if batch == batch_max:
        return packed
    out = torch.zeros(packed_len(batch_max, pool), dtype=packed.dtype, device=packed.device)
    out[:batch] = packed[:batch]
    out[batch_max:batch_max + batch] = packed[batch:2 * batch]
    out[2 * batch_max:] = packed[2 * batch:]
    return out


def gather_packed(packed_local: torch.Tensor, group=None, async_op: bool = False):
    """One all-gather of the packed corner buffers. Returns (gathered (R, len) tensor, work|None)."""
    world = dist.get_world_size(group)
    out = torch.empty((world, packed_local.numel()), dtype=packed_local.dtype, device=packed_local.device)
    work = dist.all_gather_into_tensor(out.view(-1), packed_local.contiguous(), group=group, async_op=async_op)
    return out, work


class OverlappedGather:
    """The path's exchange step, off the compute stream ("one HIP stream for compute + one for
    comms", the collective overlapped with the next batch's conv work).

    ``depth`` packed output buffers rotate: step i writes its corner lists into ``acquire(i)`` on the compute stream;
    ``launch(i)`` makes the side stream wait for that step's completion event, issues ONE ``all_gather_into_tensor`` of
    the fused buffer (RCCL over xGMI) with ``async_op=True`` and copies the gathered lists to pinned host memory -- all
    on the side stream, so step i+1's convolutions run underneath.  A slot is handed out again only after the side
    stream has finished reading it (an event the compute stream waits on; the host never blocks).
    ``backend="gloo"`` (several ranks on ONE GPU, smoke tests) has no device collective: the side stream does the D2H
    and the host-side gloo all-gather is completed lazily in ``retire`` / ``result`` -- after the next step has been
    enqueued, so the GPU still overlaps it."""

    def __init__(self, n_i32: int, device: torch.device, group=None, backend: str = "nccl", depth: int = 2, timing: bool = False):
        """``timing=True``: a timing-enabled event pair on the side stream brackets every exchange (the all-gather + the D2H of the
        gathered lists; gloo: the D2H of the local list); ``gather_ms`` collects the durations as slots are reused / read, so that a
        multi-GPU run can say what the exchange step costs and whether it stayed under the next step's convolutions."""
        self.n, self.dev, self.group, self.backend, self.depth = n_i32, device, group, backend, depth
        self.world = dist.get_world_size(group)
        self.overlapped = True
        self.timing = bool(timing)
        self.gather_ms: List[float] = []
        with torch.cuda.device(device):
            self.side = torch.cuda.Stream()
            self.out = [torch.empty((n_i32,), dtype=torch.int32, device=device) for _ in range(depth)]
            self.ev_compute = [torch.cuda.Event() for _ in range(depth)]
            self.ev_side = [torch.cuda.Event(enable_timing=self.timing) for _ in range(depth)]
            self.ev_g0 = [torch.cuda.Event(enable_timing=True) for _ in range(depth)] if self.timing else None
            self.host = [torch.empty((self.world, n_i32), dtype=torch.int32).pin_memory() for _ in range(depth)]
            if backend == "nccl":
                self.gathered = [torch.empty((self.world, n_i32), dtype=torch.int32, device=device) for _ in range(depth)]
            else:
                self.host_local = [torch.empty((n_i32,), dtype=torch.int32).pin_memory() for _ in range(depth)]
        self._pending = [False] * depth          # gloo: D2H issued, host all-gather not yet done
        self._used = [False] * depth
        self._timed = [False] * depth            # timing: an exchange of this slot has been bracketed and not yet harvested

    def warm_up(self, rounds: int = 64) -> None:
        """Run the exchange ``rounds`` times on the (still unused) buffers and wait for it.  RCCL / c10d finish initialising
        lazily during their first few dozen collectives (measured on MI355X: ~50 ms of host-side stalls spread over the first
        ~35 calls, none afterwards); a caller that times steady-state steps pays that here instead."""
        if self.backend != "nccl":
            return
        with torch.cuda.stream(self.side):
            for r in range(rounds):
                s = r % self.depth
                work = dist.all_gather_into_tensor(self.gathered[s].view(-1), self.out[s], group=self.group, async_op=True)
                work.wait()
                self.host[s].copy_(self.gathered[s], non_blocking=True)
        self.side.synchronize()

    def acquire(self, i: int) -> torch.Tensor:
        """Output buffer of step i; the compute stream waits until the side stream has released it."""
        s = i % self.depth
        if self._used[s]:
            torch.cuda.current_stream(self.dev).wait_event(self.ev_side[s])
            self._harvest(s)
        return self.out[s]

    def _harvest(self, s: int) -> None:
        """Duration of the slot's last exchange, if it has completed (never blocks: a slot still in flight is harvested later)."""
        if self.timing and self._timed[s] and self.ev_side[s].query():
            self.gather_ms.append(self.ev_g0[s].elapsed_time(self.ev_side[s]))
            self._timed[s] = False

    def launch(self, i: int) -> None:
        """Call right after step i's kernels have been enqueued on the current (compute) stream."""
        s = i % self.depth
        compute = torch.cuda.current_stream(self.dev)
        self.ev_compute[s].record(compute)
        self._harvest(s)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_compute[s])
            if self.timing:
                self.ev_g0[s].record(self.side)
                self._timed[s] = True
            if self.backend == "nccl":
                work = dist.all_gather_into_tensor(self.gathered[s].view(-1), self.out[s], group=self.group, async_op=True)
                work.wait()                                   # the SIDE stream waits for RCCL's stream; the host does not
                self.host[s].copy_(self.gathered[s], non_blocking=True)
            else:
                self.host_local[s].copy_(self.out[s], non_blocking=True)
                self._pending[s] = True
            self.ev_side[s].record(self.side)
        self._used[s] = True

    def retire(self, i: int) -> None:
        """gloo only: complete step i's host-side all-gather (no-op for nccl, for i < 0 and when already done)."""
        if i < 0:
            return
        s = i % self.depth
        if self._pending[s]:
            self.ev_side[s].synchronize()
            dist.all_gather_into_tensor(self.host[s].view(-1), self.host_local[s], group=self.group)
            self._pending[s] = False

    def drain(self) -> None:
        for s in range(self.depth):
            if self._pending[s]:
                self.ev_side[s].synchronize()
                dist.all_gather_into_tensor(self.host[s].view(-1), self.host_local[s], group=self.group)
                self._pending[s] = False
        self.side.synchronize()
        for s in range(self.depth):
            self._harvest(s)

    def result(self, i: int) -> np.ndarray:
        """(world, n) int32 host array of step i's gathered corner lists (blocks until they have arrived)."""
        s = i % self.depth
        self.retire(i)
        self.ev_side[s].synchronize()
        return self.host[s].numpy().copy()


def unpack_gathered(gathered: np.ndarray, n_frames: int, world: int, pool: int, refined: bool):
    """Host side: (R, len) gathered buffers (each laid out for ceil(n/R) frames and ``pool`` slots) -> (list of n_frames keypoint
    arrays in global frame order -- ``None`` for frames of a rank whose pool overflowed --, counts (n_frames,), the largest number
    of corners any ONE rank produced: > pool means "run again with at least that pool")."""
    batch_max = shard_range(n_frames, 0, world)[1]
    res: List[np.ndarray] = []
    counts_all = []
    need = 0
    for r in range(world):
        lo, hi = shard_range(n_frames, r, world)
        rr, cc = unpack_results(gathered[r], batch_max, pool, refined)
        res.extend(rr[:hi - lo])
        counts_all.extend(cc[:hi - lo].tolist())
        # ALL batch_max counts of the rank: blank padding frames of a ragged shard (infer_batches_sharded runs them) take pool
        # slots too, in whatever order the frames finished -- a pool that holds the real frames alone is not enough then
        need = max(need, int(cc.astype(np.int64).sum()))
    return res, np.asarray(counts_all, np.int32), need


def infer_frames_sharded(frames_gray: np.ndarray, dust_bin_ids: int, deepc, refinenet=None, kmax: int = 64,
                         group=None, run_local=None, pool: Optional[int] = None):
    """Collective version of ``inference.infer_batch``: every rank passes the same (B,H,W) gray or (B,H,W,3) BGR host
    array (or at least its own slice), processes frames [lo,hi) and receives ALL results.

    ``run_local(frames_slice, pool) -> packed int32 tensor`` defaults to the HIP pipeline; the gloo CPU
    tests inject a stand-in so that the sharding / packing / exchange logic is covered without a GPU.
    ``pool`` = corner slots per rank (default ceil(B/R) * kmax).  If some rank's shard fires more cells than that, every rank
    sees it in the gathered counts and all of them repeat the call with the pool the first pass asked for.
    """
    from .inference import infer_batch_device  # local import: needs the GPU library
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = frames_gray.shape[0]
    lo, hi = shard_range(n, rank, world)
    batch_max = shard_range(n, 0, world)[1]
    if pool is None:
        pool = max(1, batch_max * kmax)
    if run_local is None:
        det = deepc.model if hasattr(deepc, "model") else deepc

        def run_local(fr, pool_):
            d = torch.from_numpy(np.ascontiguousarray(fr)).to(det.device)
            return infer_batch_device(d, dust_bin_ids, deepc, refinenet, pool=pool_)
    host_exchange = dist.get_backend(group) == "gloo"        # gloo (CPU tests, several ranks on one GPU): through host memory
    while True:
        packed = run_local(frames_gray[lo:hi], pool) if hi > lo else None
        if packed is None:
            dev = "cpu" if host_exchange else torch.device("cuda", torch.cuda.current_device())
            packed = torch.zeros(packed_len(0, pool), dtype=torch.int32, device=dev)
        elif host_exchange and packed.device.type != "cpu":
            packed = packed.cpu()
        packed = pad_packed(packed, hi - lo, pool, batch_max)
        gathered, _ = gather_packed(packed, group)
        res, counts, need = unpack_gathered(gathered.cpu().numpy(), n, world, pool, refinenet is not None)
        if need <= pool:
            return res
        pool = need        # the same number on every rank: the repeat is collective


def infer_batches_sharded(batches, dust_bin_ids: int, deepc, refinenet=None, kmax: int = 64, group=None,
                          backend: Optional[str] = None, depth: int = 2):
    """Pipelined collective caller: ``batches`` is an iterable of (B,H,W) gray or (B,H,W,3) BGR uint8 host arrays of ONE shape
    (every rank passes the same sequence); yields, in order, the list of B keypoint arrays of each batch on every rank.

    Batch i+1's upload and kernels are enqueued before batch i's gathered corner lists are waited for, and the gather
    itself runs on :class:`OverlappedGather`'s side stream, so the exchange step costs no GPU time on the compute
    stream.  Ranks with fewer frames than rank 0 (ragged split) pad with blank frames whose results are dropped.
    Capacity: every rank's corner pool holds ceil(B/R) * kmax corners for its whole shard (no per-frame cap); a batch on which
    some rank needs more is repeated by all ranks with :func:`infer_frames_sharded` at the pool the first pass reported."""
    from .inference import infer_batch_device  # local import: needs the GPU library
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    backend = backend or dist.get_backend(group)
    det = deepc.model if hasattr(deepc, "model") else deepc
    dev = det.device
    og = None
    pending = []          # (step index, frames)
    i = 0

    def finish(step, frames_j):
        res, _, need = unpack_gathered(og.result(step), frames_j.shape[0], world, pool, refinenet is not None)
        # every rank sees the same gathered buffers, so every rank takes this branch (the second condition cannot hold without the
        # first -- a frame only comes back as None when its rank's pool overflowed -- and is kept as a guard)
        if need > pool or any(r is None for r in res):
            res = infer_frames_sharded(frames_j, dust_bin_ids, deepc, refinenet, group=group, pool=need)
        return res

    for frames in batches:
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if not (frames.ndim == 3 or (frames.ndim == 4 and frames.shape[3] == 3)):
            raise ValueError("expected (B,H,W) gray or (B,H,W,3) BGR uint8 frames")
        n = frames.shape[0]
        lo, hi = shard_range(n, rank, world)
        bmax = shard_range(n, 0, world)[1]
        if og is None:
            shape0 = frames.shape
            pool = max(1, bmax * kmax)
            og = OverlappedGather(packed_len(bmax, pool), dev, group, backend, depth)
        elif frames.shape != shape0:
            raise ValueError("infer_batches_sharded needs batches of one shape; flush and start a new call")
        local = np.zeros((bmax,) + frames.shape[1:], np.uint8)
        local[:hi - lo] = frames[lo:hi]
        if len(pending) == depth:                       # the slot about to be reused: hand its results out first
            j, fj = pending.pop(0)
            yield finish(j, fj)
        d = torch.from_numpy(local).to(dev, non_blocking=True)
        infer_batch_device(d, dust_bin_ids, deepc, refinenet, out=og.acquire(i), pool=pool)
        og.launch(i)
        pending.append((i, frames))
        i += 1
    for j, fj in pending:
        yield finish(j, fj)
