"""Worker of tests/test_gpu_sharded.py: one rank of the N>1 PRODUCT path (real HIP kernels) on the GPU box.

Launched with `python -m torch.distributed.run --nproc-per-node R ... sharded_gpu_worker.py <backend> <out.json>`.
backend "gloo": R ranks share the one visible GPU (exchange through host memory); "nccl": one GPU per rank (RCCL).
Every rank runs `infer_frames_sharded` (ragged split) and the pipelined `infer_batches_sharded` (side-stream gather,
double-buffered) through the HIP pipeline; rank 0 compares every frame with the oracle and writes the verdict."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from deepcharuco_amd import weights as W  # noqa: E402
from deepcharuco_amd import workload as WL  # noqa: E402
from deepcharuco_amd.models.net import dcModel, lModel  # noqa: E402
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet  # noqa: E402
from deepcharuco_amd.sharding import infer_batches_sharded, infer_frames_sharded, shard_range  # noqa: E402


def main():
    backend, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % ndev if backend == "gloo" else local)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    n_frames, h, w = 11, 120, 160                      # ragged: 11 frames over 2 ranks -> 6 + 5
    frames = np.concatenate([W.synthetic_frames("board", 7000, 7, h, w), W.synthetic_frames("noise", 7100, 4, h, w)])
    sd_dc = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 61), torch.from_numpy(frames).to(dev), dev, per_frame=12)
    # every rank calibrated on the same frames with the same kernels -> identical weights; assert it
    bias = torch.tensor([float(sd_dc["convDb.bias"][16])], dtype=torch.float64)
    gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    if backend == "gloo":
        dist.all_gather(gathered, bias)
    else:
        gathered = [g.to(dev) for g in gathered]
        dist.all_gather(gathered, bias.to(dev))
    assert all(float(g) == float(bias) for g in gathered), "ranks calibrated different weights"
    sd_rn = W.synthetic_state_dict("refinenet", 62)
    dc, rn = lModel(dcModel(16, sd_dc, dev)), lRefineNet(RefineNet(sd_rn, dev))

    res_single = infer_frames_sharded(frames, 16, dc, rn, kmax=64)
    # pipelined: 5 batches of 11 frames (rotated so that batches differ), two in flight
    batches = [np.roll(frames, s, axis=0) for s in range(5)]
    res_pipe = list(infer_batches_sharded(batches, 16, dc, rn, kmax=64))
    # empty shard: 1 frame over 2 ranks
    res_one = infer_frames_sharded(frames[:1], 16, dc, rn, kmax=64)
    # BGR frames (what the reference's callers hold): gray replicated x3 converts back to itself, so the results must not change;
    # and pools far too small (2 slots per frame): both ranks repeat the batch collectively with the pool the first pass reported
    bgr = np.ascontiguousarray(np.repeat(frames[..., None], 3, axis=3))
    res_bgr = infer_frames_sharded(bgr, 16, dc, rn, kmax=64)
    (res_bgr_pipe,) = list(infer_batches_sharded([bgr], 16, dc, rn, kmax=2))

    verdict = None
    if rank == 0:
        from oracle import deepcharuco_oracle as O
        t_dc, t_rn = O.to_torch_state_dict(sd_dc), O.to_torch_state_dict(sd_rn)
        exp = [O.infer_image(None, 16, t_dc, t_rn, gray=f) for f in frames]
        same = lambda a, b: a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)
        bad_single = sum(not same(a, b) for a, b in zip(res_single, exp))
        bad_pipe = 0
        for s, res in enumerate(res_pipe):
            order = np.roll(np.arange(n_frames), s)
            bad_pipe += sum(not same(res[i], exp[order[i]]) for i in range(n_frames))
        verdict = dict(backend=backend, world=world, devices=ndev, frames=n_frames,
                       split=[shard_range(n_frames, r, world) for r in range(world)],
                       corners=int(sum(e.shape[0] for e in exp if e.ndim == 2)),
                       mismatched_single=int(bad_single), mismatched_pipelined=int(bad_pipe),
                       pipelined_batches=len(res_pipe), one_frame_ok=bool(len(res_one) == 1 and same(res_one[0], exp[0])),
                       mismatched_bgr=int(sum(not same(a, b) for a, b in zip(res_bgr, exp)) + sum(not same(a, b) for a, b in zip(res_bgr_pipe, exp))))
        with open(out_path, "w") as f:
            json.dump(verdict, f)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        ok = verdict["mismatched_single"] == 0 and verdict["mismatched_pipelined"] == 0 and verdict["one_frame_ok"] and verdict["mismatched_bgr"] == 0 \
            and verdict["corners"] > 50
        sys.exit(0 if ok else 4)


if __name__ == "__main__":
    main()
