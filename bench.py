#!/usr/bin/env python3
"""bench.py -- end-to-end frames/s of the detect+refine path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the whole hot path (detector -> decode -> patch gather -> RefineNet ->
sub-pixel xy, + for N>1 one RCCL all-gather of the packed corner lists) over one batch of
synthetic 320x240 gray frames ALREADY RESIDENT in HBM, ending with the async D2H of the packed
result into pinned host memory.  Weak scaling: every rank processes `--batch` frames per step.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
hipEvent-timed inside the timed region) and, at N=1, `cpu_baseline` (the oracle on host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from deepcharuco_amd import _lib, weights as W  # noqa: E402
from deepcharuco_amd.inference import infer_batch_device, unpack_results  # noqa: E402
from deepcharuco_amd.models.net import dcModel, lModel  # noqa: E402
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet  # noqa: E402

METRIC = "frames/sec end-to-end (detect+refine) at 320x240; corner-id match vs ref"
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
DET_GFLOP_240x320 = 12.879052800  # 2 * 6,439,526,400 MAC   
REF_GFLOP_PER_PATCH = 0.871072256  # 2 * 435,536,128 MAC
REFERENCE_README_FPS = 200.0      # "> 200 fps" GTX1080Ti, bs=1 (README.md:42-44)


def calibrate_dustbin(sd_dc, frames_dev, dev, n_ids=16, per_frame=16):
    """Set convDb.bias[n_ids] so that on average per_frame cells fire per frame.
    Setup only (outside every timed region): HIP detector logits -> host numpy."""
    det = dcModel(n_ids, sd_dc, dev)
    out = det.forward_u8(frames_dev)
    loc, ids = out["loc"].cpu().numpy(), out["ids"].cpu().numpy()
    la = loc.argmax(1)
    m = ids[:, :n_ids].max(1) - ids[:, n_ids]
    m = np.sort(np.where(la == 64, -1e30, m).ravel())[::-1]
    k = per_frame * frames_dev.shape[0]
    delta = np.float32((m[k - 1] + m[k]) / 2)
    sd = {k_: v.copy() for k_, v in sd_dc.items()}
    sd["convDb.bias"][n_ids] = np.float32(sd["convDb.bias"][n_ids] + delta)
    del det
    return sd


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_pmc_traffic.json:
    FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE); None when no profile matches."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
        return d.get(kernel_name)
    except Exception:
        return None


def cpu_baseline(sd_dc, sd_rn, frames_u8, budget_s=24.0):
    """The oracle (CPU restatement of the reference, verified identical to it) timed on this host's cores, on a
    bounded sample of the same workload, two ways:
      * the reference's own protocol (src/benchmark.py:37-53): bs=1 infer_image loop after warm-up;
      * one batched pass per thread count (detector on all B frames at once, RefineNet on all patches at once),
        which is what a CPU user after throughput would run.
    oneDNN does not scale to every core of a big host at these sizes, so a few thread counts are tried inside the time
    budget; `value` is the best rate found, `cores` the threads that produced it."""
    from oracle import deepcharuco_oracle as O
    t_dc, t_rn = O.to_torch_state_dict(sd_dc), O.to_torch_state_dict(sd_rn)
    ncpu = os.cpu_count() or 1
    cands = sorted({min(ncpu, c) for c in (8, 16, 32, 64)})   # more threads only get slower at these sizes (tried 256)
    B = len(frames_u8)
    x_all = torch.from_numpy(np.stack([O.pre_bgr_image(f) for f in frames_u8]))          # (B,1,H,W)

    def batched():
        loc, ids = O.detector_forward(t_dc, x_all)
        patches, kp = [], []
        for b in range(B):
            k, _ = O.pred_to_keypoints(loc[b:b + 1], ids[b:b + 1], 16)
            if k.shape[0]:
                patches.append(O.extract_patches(x_all[b], k)); kp.append(k)
        if patches:
            O.refinenet_infer_patches(t_rn, torch.cat(patches), torch.cat(kp))

    single, batch = {}, {}
    for c in cands:
        torch.set_num_threads(c)
        O.infer_image(None, 16, t_dc, t_rn, gray=frames_u8[0])   # warm-up
        n, t0 = 0, time.time()
        while (time.time() - t0) < 0.5 * budget_s / len(cands) and n < 64:
            O.infer_image(None, 16, t_dc, t_rn, gray=frames_u8[n % B])
            n += 1
        single[c] = (n / (time.time() - t0), n)
        if c in (16, 32):
            t0 = time.time()
            batched()
            batch[c] = B / (time.time() - t0)
    best_s = max(single, key=lambda c: single[c][0])
    best_b = max(batch, key=lambda c: batch[c]) if batch else None
    use_batch = best_b is not None and batch[best_b] > single[best_s][0]
    return {"value": round(batch[best_b] if use_batch else single[best_s][0], 3), "unit": "frames/s",
            "cores": int(best_b if use_batch else best_s), "kind": "port",
            "sample": ("best of two CPU protocols on the same weights/frames as the GPU run (torch-CPU fp32 restatement of the "
                       "reference = oracle): bs=1 infer_image loop [" + ", ".join(f"{c} thr: {v[0]:.1f} fps" for c, v in single.items())
                       + f"] ({single[best_s][1]} frames at the best setting); one batched pass of {B} frames 320x240 ["
                       + ", ".join(f"{c} thr: {v:.1f} fps" for c, v in batch.items()) + f"]; host has {ncpu} logical CPUs")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--kmax", type=int, default=64, help="corner capacity per frame")
    ap.add_argument("--frames", default="board", choices=["board", "noise"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo only to smoke-test the multi-process flow "
                         "with several ranks on one GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but only {ndev} GPUs visible")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    L = _lib.lib()

    B, H, Wd, kmax = args.batch, args.height, args.width, args.kmax
    frames = W.synthetic_frames(args.frames, 1000 + rank * B, B, H, Wd)
    d_frames = torch.from_numpy(frames).to(dev)
    # every rank calibrates on the SAME frames (rank 0's), so all ranks run identical weights
    calib = d_frames if rank == 0 else torch.from_numpy(W.synthetic_frames(args.frames, 1000, B, H, Wd)).to(dev)
    sd_dc = calibrate_dustbin(W.synthetic_state_dict("detector", 1234), calib, dev)
    del calib
    sd_rn = W.synthetic_state_dict("refinenet", 1235)
    dc, rn = lModel(dcModel(16, sd_dc, dev)), lRefineNet(RefineNet(sd_rn, dev))

    n_i32 = B + B * kmax * 6
    out_dev = torch.empty((n_i32,), dtype=torch.int32, device=dev)
    host_local = torch.empty((n_i32,), dtype=torch.int32).pin_memory()
    host_all = torch.empty((world, n_i32), dtype=torch.int32).pin_memory() if world > 1 else None   # every rank (gloo mode fills it everywhere)
    gathered = torch.empty((world, n_i32), dtype=torch.int32, device=dev) if world > 1 else None

    def step():
        packed = infer_batch_device(d_frames, 16, dc, rn, kmax, out=out_dev)
        if world > 1 and args.backend == "nccl":   # the path's only exchange step: one fused all-gather of the corner lists
            dist.all_gather_into_tensor(gathered.view(-1), packed)
            if rank == 0:
                host_all.copy_(gathered, non_blocking=True)
        elif world > 1:                             # gloo smoke mode: exchange through host memory
            dist.all_gather_into_tensor(host_all.view(-1), packed.cpu())
        else:
            host_local.copy_(packed, non_blocking=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import ctypes as C

    def fetch_profile(total_patches):
        """-> {kernel name: [flop, ms, launches, clock-weighted ms]} of the launches recorded since profile_enable(1)."""
        n = L.dcx_profile_count()
        ids_ = (C.c_int * max(n, 1))(); nimg = (C.c_int * max(n, 1))(); lim = (C.c_int * max(n, 1))()
        fl = (C.c_double * max(n, 1))(); ms = (C.c_float * max(n, 1))(); ghz = (C.c_float * max(n, 1))()
        n = L.dcx_profile_fetch(ids_, nimg, lim, fl, ms, n)
        L.dcx_profile_clocks(ghz, n)
        agg = {}
        for i in range(n):
            imgs = total_patches if lim[i] else nimg[i]     # RefineNet launches cover only the live patches
            a = agg.setdefault(int(ids_[i]), [0.0, 0.0, 0, 0.0])
            a[0] += fl[i] * imgs
            a[1] += ms[i]
            a[2] += 1
            a[3] += ghz[i] * ms[i]
        return agg

    # warm-up (every conv launch hipEvent-bracketed: finds the dominant kernel for the timed region)
    if not args.no_profile:
        L.dcx_profile_filter(-1)
        L.dcx_profile_enable(1)
    for _ in range(args.warmup):
        step()
    fence()
    dom_id = -1
    if not args.no_profile:
        warm = fetch_profile(16.0 * B)
        L.dcx_profile_enable(0)
        if warm:
            dom_id = max(warm.items(), key=lambda kv: kv[1][1])[0]
        L.dcx_profile_filter(dom_id)          # timed region: only the dominant kernel is bracketed (2 launches/step)
        L.dcx_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- what the timed steps produced (outside the timed region)
    local_counts = unpack_results(out_dev.cpu().numpy(), B, kmax, True)[1]
    if world > 1 and rank == 0:
        counts = np.concatenate([unpack_results(host_all[r].numpy(), B, kmax, True)[1] for r in range(world)])
    else:
        counts = local_counts
    mean_k = float(np.minimum(counts, kmax).mean())
    overflow = int((counts > kmax).sum())
    total_patches = float(np.minimum(local_counts, kmax).sum())

    # ---- roofline of the dominant kernel: its launches inside the timed region, hipEvent-bracketed on the
    # stream they run on; the per-kernel table comes from 3 extra, fully bracketed steps after the timed region
    roofline = None
    if not args.no_profile:
        timed = fetch_profile(total_patches)
        L.dcx_profile_enable(0)
        L.dcx_profile_filter(-1)
        L.dcx_profile_enable(1)
        extra_steps = 3
        for _ in range(extra_steps):
            step()
        fence()
        full = fetch_profile(total_patches)
        L.dcx_profile_enable(0)
        kname = lambda k: L.dcx_profile_kernel_name(k).decode()
        if dom_id not in timed:                 # e.g. --warmup 0: everything was bracketed, pick the dominant kernel now
            dom_id = max(timed.items(), key=lambda kv: kv[1][1])[0]
        flop, msum, launches, clk = timed[dom_id]
        achieved = flop / (msum * 1e-3) / 1e12
        conv_ms = sum(a[1] for a in full.values())
        conv_flop = sum(a[0] for a in full.values())
        # The Winograd kernels execute only part of a layer's ALGORITHMIC multiply-adds on the matrix cores (1-D F(2,3): 4
        # products per 2 outputs and kernel row instead of 6 = 2/3; 2-D F(2x2,3x3): 16 per 2x2 tile instead of 36 = 4/9), so
        # `achieved` (algorithmic FLOP / time, the figure this contract asks for) can exceed the fp32-MFMA peak;
        # `executed_*` is what the matrix pipe really ran.
        scale_of = lambda nm: 4.0 / 9.0 if "wino2" in nm else 2.0 / 3.0 if "wino" in nm else 1.0
        exec_scale = scale_of(kname(dom_id))
        conv_exec = sum(a[0] * scale_of(kname(k)) for k, a in full.items())
        algo = ("winograd F(2x2,3x3): 4/9 of the algorithmic MACs are executed" if "wino2" in kname(dom_id)
                else "winograd F(2,3) along x: 2/3 of the algorithmic MACs are executed" if "wino" in kname(dom_id)
                else "direct implicit GEMM")
        roofline = {"bound": "mfma", "kernel": kname(dom_id), "launches": launches,
                    "algorithm": algo,
                    "executed_achieved": round(achieved * exec_scale, 2),
                    "executed_frac": round(achieved * exec_scale / PEAK_F32_MFMA_TFLOPS, 4),
                    "avg_launch_ms": round(msum / launches, 4),
                    "algorithmic_gflop_per_launch": round(flop / launches / 1e9, 3),
                    "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": pmc_traffic(kname(dom_id)),
                    "shader_clock_ghz": round(clk / msum, 3),
                    "all_conv_kernels": {"achieved": round(conv_flop / (conv_ms * 1e-3) / 1e12, 2),
                                         "ms_per_step": round(conv_ms / extra_steps, 3),
                                         "frac": round(conv_flop / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                         "executed_frac": round(conv_exec / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                         "source": f"{extra_steps} fully bracketed steps after the timed region"},
                    "per_kernel": {kname(k): {"ms_per_step": round(v[1] / extra_steps, 4),
                                              "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2) if v[1] > 0 else None,
                                              "launches_per_step": v[2] / extra_steps,
                                              "clock_ghz": round(v[3] / v[1], 3) if v[1] > 0 else None}
                                   for k, v in full.items()}}

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    fps = world * B * args.steps / elapsed
    gflop_frame = DET_GFLOP_240x320 * (H * Wd) / (240 * 320) + REF_GFLOP_PER_PATCH * mean_k
    line = {
        "metric": METRIC, "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": round(fps / REFERENCE_README_FPS, 3), "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"bs={B} {Wd}x{H} frames per GPU, full detect+refine pipeline (BASELINE configs[1])",
                   "batch_per_gpu": B, "global_batch": B * world, "height": H, "width": Wd, "kmax": kmax,
                   "frames": args.frames, "mean_corners_per_frame": round(mean_k, 2), "frames_over_kmax": overflow,
                   "weights": "numpy-seeded synthetic (seed 1234/1235), dust-bin bias calibrated to ~16 corners/frame",
                   "parallelism": f"frames sharded, 1 process/GPU x{world}" + (", RCCL all-gather of corner lists" if world > 1 else ""),
                   "algorithmic_gflop_per_frame": round(gflop_frame, 3),
                   "e2e_frac_of_f32_mfma_peak": round(fps * gflop_frame / 1e3 / (PEAK_F32_MFMA_TFLOPS * world), 4),
                   "vs_baseline_note": "reference README '>200 fps' (GTX1080Ti, bs=1, src/benchmark.py)"},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sd_dc, sd_rn, frames)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
