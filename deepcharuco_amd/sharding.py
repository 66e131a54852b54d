"""Frame sharding across the GPUs of one node + the single exchange step of the path.

The reference has no multi-device code; frames are independent
(inference.py:32-70 keeps no cross-frame state, BN is in eval mode), so the path shards
embarrassingly: rank r of R owns frames [lo, hi) and runs the whole detect+refine pipeline on
them.  The only collective is one ``all_gather_into_tensor`` of the fixed-shape packed corner
buffer per batch (RCCL over xGMI with backend "nccl"; "gloo" for the CPU tests).  The payload
is KB-scale (B/R * (1 + 6*kmax) int32), i.e. latency-bound, so it is sent as ONE fused buffer.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .inference import unpack_results


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first ``n_frames % world`` ranks get one extra frame."""
    if not (0 <= rank < world):
        raise ValueError("bad rank")
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def packed_len(batch: int, kmax: int) -> int:
    return batch + batch * kmax * 4 + batch * kmax * 2


def pad_packed(packed: torch.Tensor, batch: int, kmax: int, batch_max: int) -> torch.Tensor:
    """Re-lay a rank's packed buffer [counts|rows|xy] for ``batch`` frames into the layout for
    ``batch_max`` frames (missing frames get count 0) so every rank contributes the same shape."""
    if batch == batch_max:
        return packed
    out = torch.zeros(packed_len(batch_max, kmax), dtype=packed.dtype, device=packed.device)
    out[:batch] = packed[:batch]
    r0, r1 = batch, batch + batch * kmax * 4
    out[batch_max:batch_max + batch * kmax * 4] = packed[r0:r1]
    out[batch_max + batch_max * kmax * 4:batch_max + batch_max * kmax * 4 + batch * kmax * 2] = packed[r1:]
    return out


def gather_packed(packed_local: torch.Tensor, group=None, async_op: bool = False):
    """One all-gather of the packed corner buffers. Returns (gathered (R, len) tensor, work|None)."""
    world = dist.get_world_size(group)
    out = torch.empty((world, packed_local.numel()), dtype=packed_local.dtype, device=packed_local.device)
    work = dist.all_gather_into_tensor(out.view(-1), packed_local.contiguous(), group=group, async_op=async_op)
    return out, work


def unpack_gathered(gathered: np.ndarray, n_frames: int, world: int, kmax: int, refined: bool):
    """Host side: (R, len) gathered buffers -> list of n_frames keypoint arrays in global frame order."""
    batch_max = shard_range(n_frames, 0, world)[1]
    res: List[np.ndarray] = []
    counts_all = []
    for r in range(world):
        lo, hi = shard_range(n_frames, r, world)
        rr, cc = unpack_results(gathered[r], batch_max, kmax, refined)
        res.extend(rr[:hi - lo])
        counts_all.extend(cc[:hi - lo].tolist())
    return res, np.asarray(counts_all, np.int32)


def infer_frames_sharded(frames_gray: np.ndarray, dust_bin_ids: int, deepc, refinenet=None, kmax: int = 64,
                         group=None, run_local=None):
    """Collective version of ``inference.infer_batch``: every rank passes the same (B,H,W) host
    array (or at least its own slice), processes frames [lo,hi) and receives ALL results.

    ``run_local(frames_slice) -> packed int32 tensor`` defaults to the HIP pipeline; the gloo CPU
    tests inject a stand-in so that the sharding / packing / exchange logic is covered without a GPU.
    """
    from .inference import infer_batch_device  # local import: needs the GPU library
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = frames_gray.shape[0]
    lo, hi = shard_range(n, rank, world)
    batch_max = shard_range(n, 0, world)[1]
    if run_local is None:
        det = deepc.model if hasattr(deepc, "model") else deepc

        def run_local(fr):
            d = torch.from_numpy(np.ascontiguousarray(fr)).to(det.device)
            return infer_batch_device(d, dust_bin_ids, deepc, refinenet, kmax)
    if hi > lo:
        packed = run_local(frames_gray[lo:hi])
    else:
        packed = None
    if packed is None:
        dev = "cpu" if dist.get_backend(group) == "gloo" else torch.device("cuda", torch.cuda.current_device())
        packed = torch.zeros(packed_len(0, kmax), dtype=torch.int32, device=dev)
    packed = pad_packed(packed, hi - lo, kmax, batch_max)
    gathered, _ = gather_packed(packed, group)
    res, counts = unpack_gathered(gathered.cpu().numpy(), n, world, kmax, refinenet is not None)
    if int(counts.max(initial=0)) > kmax:
        raise RuntimeError(f"a frame fired {int(counts.max())} cells > kmax={kmax}; raise kmax")
    return res
