#!/usr/bin/env python3
"""bench.py -- end-to-end frames/s of the detect+refine path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5]
      N = 1: runs in this process.  N > 1 from a bare shell (no RANK/WORLD_SIZE): starts its own N ranks under
      torch.distributed.run on 127.0.0.1 at a free port (self_launch) and relays rank 0's JSON line + the exit code.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     (ranks started by the caller)

A "step" = one pass of the whole hot path (detector -> decode -> patch gather -> RefineNet -> sub-pixel xy, + for N>1
one RCCL all-gather of the packed corner lists, issued on a side stream so that it overlaps the next step's
convolutions) over one batch of synthetic gray frames ALREADY RESIDENT in HBM, ending with the async D2H of the packed
result into pinned host memory.  Weak scaling: every rank processes the same number of frames per step for every N
(32 by default = BASELINE configs[1] per GPU), so the driver's 1/2/4/8-GPU values form a true weak-scaling curve.
The K timed steps go through the product's non-blocking caller for HBM-resident frames (stream.ResidentStream; at N>1
the same launches under sharding.OverlappedGather's side-stream gather) on ONE HIP stream.  `--streams S` alternates
consecutive steps between S streams (S batches in flight); the roofline block is then measured in a second pass of the
same K steps on one stream (`single_stream`).  Rounds 2-5 reported "+6 ... +15 % with two batches in flight": that was a
measurement artifact -- the second batch of that comparison came from another seed and fired 38 % fewer corners; with
equal work in every batch a second stream LOSES 1-4 % (profiles/experiments/r05_batches_in_flight_equal_work.txt).

Rank 0 prints ONE JSON line (contract in the task statement) with
  roofline        dominant kernel, hipEvent-timed inside the timed region;
  parity          frames of the TIMED batch compared with the oracle (ids, cells, sub-pixel xy identical); any
                  mismatch makes the process exit non-zero;
  cpu_baseline    (N=1) the oracle timed on the host cores;
  other_configs   the other BASELINE configs at their per-GPU size (cfg3, cfg4 load, cfg5 load with exactly 16 corners
                  per frame; at N>1 cfg4 / cfg5 run on all ranks with the gather -- at N=8 those ARE configs[3] and [4]),
                  each with its own parity check, plus (N=1) the reference's own bs=1 protocol (src/benchmark.py:37-53).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from deepcharuco_amd import _lib, weights as W  # noqa: E402
from deepcharuco_amd import workload as WL  # noqa: E402
from deepcharuco_amd.inference import infer_batch_device, infer_image, packed_len, unpack_results  # noqa: E402
from deepcharuco_amd.models.net import dcModel, lModel  # noqa: E402
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet  # noqa: E402
from deepcharuco_amd.sharding import OverlappedGather  # noqa: E402
from deepcharuco_amd.stream import ResidentStream  # noqa: E402

METRIC = "frames/sec end-to-end (detect+refine) at 320x240; corner-id match vs ref"
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
DET_GFLOP_240x320 = 12.879052800  # 2 * 6,439,526,400 MAC   
REF_GFLOP_PER_PATCH = 0.871072256  # 2 * 435,536,128 MAC
REFERENCE_README_FPS = 200.0      # "> 200 fps" GTX1080Ti, bs=1 (README.md:42-44)
FRAME_SEED = 1000
SETTLE_S = 0.15                   # untimed settle phase ahead of the warm-up steps (GPU power state, RCCL lazy init): see run_config


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/pmc_traffic.json: FETCH_SIZE
    with the calibrated gfx950 correction + WRITE_SIZE).  The file is stamped with a hash of the kernel sources it was
    measured on; a stale stamp (kernels changed since) yields None rather than a silently outdated number."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    stamp = d.get("_csrc_sha256")
    if stamp is not None and stamp != _lib.csrc_sha256():
        return None
    v = d.get(kernel_name)
    return v if isinstance(v, (int, float)) else None


class Oracle:
    """The checker: oracle/deepcharuco_oracle.py (CPU restatement of the reference, pinned to it by make_golden.py)."""

    def __init__(self, sd_dc, sd_rn):
        from oracle import deepcharuco_oracle as O
        self.O = O
        self.t_dc, self.t_rn = O.to_torch_state_dict(sd_dc), O.to_torch_state_dict(sd_rn)
        self.cache = {}

    def frame(self, gray, key=None, use_cache=True):
        """use_cache=False always computes (the timed CPU baseline) but still records the result for the parity block."""
        if use_cache and key is not None and key in self.cache:
            return self.cache[key]
        r = self.O.infer_image(None, 16, self.t_dc, self.t_rn, gray=gray)
        if key is not None:
            self.cache[key] = r
        return r


def parity_block(oracle, checks):
    """checks: list of (label, gray frame, HIP result).  -> the JSON block; identical arrays required."""
    corners = mism = 0
    bad = []
    for label, gray, got in checks:
        exp = oracle.frame(gray, label)
        corners += 0 if exp.ndim == 1 else exp.shape[0]
        if got is None or got.shape != exp.shape or got.dtype != exp.dtype or not np.array_equal(got, exp):
            mism += 1
            bad.append(label)
    blk = {"frames_checked": len(checks), "corners": int(corners), "mismatched_frames": int(mism),
           "against": "oracle (CPU restatement pinned to the reference): ids, cells and sub-pixel xy identical"}
    if bad:
        blk["mismatched"] = bad[:8]
    return blk


def cpu_baseline(oracle, name, frames_u8, budget_s=24.0):
    """The oracle timed on this host's cores, on a bounded sample of the same workload, two ways:
      * the reference's own protocol (src/benchmark.py:37-53): bs=1 infer_image loop after warm-up;
      * one batched pass per thread count (detector on all B frames at once, RefineNet on all patches at once).
    oneDNN does not scale to every core of a big host at these sizes, so a few thread counts are tried inside the time
    budget; `value` is the best rate found, `cores` the threads that produced it."""
    O = oracle.O
    ncpu = os.cpu_count() or 1
    cands = sorted({min(ncpu, c) for c in (8, 16, 32, 64)})
    B = len(frames_u8)
    x_all = torch.from_numpy(np.stack([O.pre_bgr_image(f) for f in frames_u8]))

    def batched():
        loc, ids = O.detector_forward(oracle.t_dc, x_all)
        patches, kp = [], []
        for b in range(B):
            k, _ = O.pred_to_keypoints(loc[b:b + 1], ids[b:b + 1], 16)
            if k.shape[0]:
                patches.append(O.extract_patches(x_all[b], k)); kp.append(k)
        if patches:
            O.refinenet_infer_patches(oracle.t_rn, torch.cat(patches), torch.cat(kp))

    single, batch = {}, {}
    for c in cands:
        torch.set_num_threads(c)
        oracle.frame(frames_u8[0])   # warm-up
        n, t0 = 0, time.time()
        while (time.time() - t0) < 0.5 * budget_s / len(cands) and n < 64:
            oracle.frame(frames_u8[n % B], key=(name, 0, n % B), use_cache=False)
            n += 1
        single[c] = (n / (time.time() - t0), n)
        if c in (16, 32):
            t0 = time.time()
            batched()
            batch[c] = B / (time.time() - t0)
    torch.set_num_threads(min(ncpu, 16))
    best_s = max(single, key=lambda c: single[c][0])
    best_b = max(batch, key=lambda c: batch[c]) if batch else None
    use_batch = best_b is not None and batch[best_b] > single[best_s][0]
    return {"value": round(batch[best_b] if use_batch else single[best_s][0], 3), "unit": "frames/s",
            "cores": int(best_b if use_batch else best_s), "kind": "port",
            "sample": ("best of two CPU protocols on the same weights/frames as the GPU run (torch-CPU fp32 restatement of the "
                       "reference = oracle): bs=1 infer_image loop [" + ", ".join(f"{c} thr: {v[0]:.1f} fps" for c, v in single.items())
                       + f"] ({single[best_s][1]} frames at the best setting; every timed call computes, its result is kept "
                       f"for the parity block); one batched pass of {B} frames [" +
                       ", ".join(f"{c} thr: {v:.1f} fps" for c, v in batch.items()) + f"]; host has {ncpu} logical CPUs")}


class Ctx:
    pass


def fetch_profile(L, total_patches):
    """-> {kernel id: [flop, ms, launches, clock-weighted ms]} of the launches recorded since profile_enable(1)."""
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


def run_config(cx, name, batch, height, width, kmax, frames_kind, fixed_k, steps, warmup, profile, n_check,
               want_cpu_baseline=False):
    """One measured configuration on every rank.  Returns (rank 0) the dict for the JSON line."""
    L, dev, rank, world, dist = cx.L, cx.dev, cx.rank, cx.world, cx.dist
    B, H, Wd = batch, height, width
    seed0 = FRAME_SEED + rank * 100000
    # ---- weights: every rank calibrates on the SAME frames (rank 0's first batch), so all ranks run identical weights
    calib_kind = frames_kind
    calib = torch.from_numpy(W.synthetic_frames(calib_kind, FRAME_SEED, min(B, 128), H, Wd)).to(dev)
    sd_dc = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), calib, dev, diverse_ids=True)
    del calib
    sd_rn = W.synthetic_state_dict("refinenet", 1235)
    dc, rn = lModel(dcModel(16, sd_dc, dev)), lRefineNet(RefineNet(sd_rn, dev))
    if fixed_k:
        frames, kept = WL.select_fixed_k_frames(frames_kind, seed0, B, H, Wd, fixed_k, dc, dev, max_candidates=20000)
    else:
        # Weak scaling needs the SAME work on every rank.  The dust-bin calibration fits the frames it was run on: rank 0's batch
        # fires 16.0 cells per frame, another seed's frames 9.9 under the same weights (tools/resident_stream_probe.py) -- i.e. 23
        # instead of 26.8 GFLOP per frame.  So every rank processes rank 0's frames, rotated by its rank: frame b of rank r is
        # frame (b + r) % B of rank 0 (rounds 1-5 gave every rank its own seed, which made ranks > 0 ~15 % lighter).
        frames = np.roll(W.synthetic_frames(frames_kind, FRAME_SEED, B, H, Wd), -rank, axis=0)
    d_frames = torch.from_numpy(frames).to(dev)
    kept_all = None
    if fixed_k and (world > 1 or cx.force_dist):
        # which candidate frames every rank selected (frame j of a rank = synthetic_frames(kind, its seed + kept[j])): rank 0 renders
        # only the few frames of the other ranks it checks instead of repeating every rank's whole selection (minutes at 1280x960)
        kept_all = [None] * world
        dist.all_gather_object(kept_all, [int(k) for k in kept])

    # the batch's corner pool: B * kmax slots shared by all frames of the batch -- a frame may fire any number of cells (the
    # reference refines every firing cell, inference.py:51-57); kmax is only the AVERAGE the buffers are sized for
    pool = B * kmax
    n_i32 = packed_len(B, pool)
    dist_on = world > 1 or cx.force_dist      # --force-dist: the N>1 code path (process group, side-stream gather, barrier) with ONE rank
    og = OverlappedGather(n_i32, dev, backend=cx.backend, timing=True) if dist_on else None
    if og is not None:
        og.warm_up()      # RCCL's lazy initialisation (tens of ms over its first dozens of calls) is not steady-state throughput
    # Batches in flight: consecutive steps run on alternating HIP streams (the product's pipelined callers: stream.ResidentStream
    # at N = 1, the same alternation under the side-stream gather at N > 1), so that batch i+1's convolutions fill the matrix
    # cores that batch i's HBM-bound / small launches, ramps and partial last rounds leave idle.  Every step is still one whole
    # pass of the path over one batch; bit-identical results (frames are independent, the default kernels are batch invariant).
    S = max(1, int(cx.streams))
    if og is not None:
        S = min(S, og.depth)
    state = {"i": 0, "last": None, "S": S}
    rs_of = {}                                  # ResidentStream per stream count (N = 1)
    cs = [torch.cuda.Stream() for _ in range(S)] if og is not None else None
    torch.cuda.synchronize()                    # d_frames / weights were produced on the default stream

    def resident(n_streams):
        if n_streams not in rs_of:
            rs_of[n_streams] = ResidentStream(16, dc, rn, batch=B, height=H, width=Wd, kmax=kmax, compute_streams=n_streams, raw=True,
                                              timing=True)
        return rs_of[n_streams]

    def step():
        i = state["i"]; state["i"] += 1
        if og is None:
            r = resident(state["S"]).submit(d_frames)      # the product call: pipeline + async D2H of the packed result, in flight
            if r is not None:
                state["last"] = r[1]
            return
        # N>1: the path's only exchange step -- ONE fused all-gather of the packed corner lists, on the side stream:
        # step i's gather (+ rank 0's D2H of all lists) overlaps step i+1's convolutions; slots are double-buffered
        og.retire(i - og.depth)                         # host-side completion of the step that used this slot (gloo only)
        with torch.cuda.stream(cs[i % state["S"]]):
            out = og.acquire(i)
            infer_batch_device(d_frames, 16, dc, rn, out=out, pool=pool)
            og.launch(i)

    def fence():
        if og is not None:
            og.drain()
        else:
            for rs in rs_of.values():
                for r in rs.flush():
                    state["last"] = r[1]
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- settle phase (set-up, not one of the W warm-up steps): untimed steps for SETTLE_S seconds before anything is counted.
    # Two things need it.  (1) The GPU's power management: after the idle stretch in which the host renders the frames the GPU sits
    # in a low power state; under load it raises the shader clock in steps, and ~30 ms into a run one transition stalls the stream for
    # ~0.9 ms (per-batch GPU times of a cold 20-step run: 3.33, 3.24, 3.24, 4.08, 3.20 ... 3.12 ms; with the settle phase: 3.16, 3.13,
    # 3.11 ... -- profiles/experiments/r06_power_state_ramp.txt).  W = 5 warm-up steps are 16 ms: a 20-step timed region would
    # measure the ramp, not the path (that was the driver-run 9,617 fps of round 5 against the builder's 50-step 10,253).
    # (2) N > 1: RCCL / c10d keep initialising lazily while the first collectives overlap real work (~50 ms of host-side stalls
    # during the first ~35 steps, none afterwards -- tools/gather_overlap_probe.py).  The timed region is still EXACTLY K steps and
    # every one of them is listed in step_breakdown.gpu_ms_per_batch.
    settle_steps = 0
    if SETTLE_S > 0:
        t_s = time.perf_counter()
        # N > 1: a FIXED count (every step is a collective: all ranks must run the same number); N = 1: by the clock
        while (settle_steps < 48) if dist_on else (settle_steps < 4 or (time.perf_counter() - t_s < SETTLE_S and settle_steps < 400)):
            step()
            settle_steps += 1
        fence()
    state["settle_steps"] = settle_steps
    if profile:
        L.dcx_profile_filter(-1)
        L.dcx_profile_enable(1)          # the dominant-kernel choice uses the W warm-up steps
    for _ in range(max(0, S - warmup)):      # set-up, not warm-up: every stream's scratch buffers exist before anything is timed
        step()
    for _ in range(warmup):
        step()
    fence()
    dom_id = -1
    if profile:
        warm = fetch_profile(L, 16.0 * B)
        L.dcx_profile_enable(0)
        warm = {k: a for k, a in warm.items() if k < 100}       # the convolution families (ids >= 100: conv1a, tail, finalize brackets)
        if warm:
            dom_id = max(warm.items(), key=lambda kv: kv[1][1])[0]

    def timed_pass(n_steps):
        """EXACTLY n_steps steps between two fences; with profiling only the dominant kernel is bracketed, and only every 5th of
        its launches (a hipEvent bracket idles its stream for ~11 us: 45 us per step if all four of a step were; 5 is coprime with
        the 4 launches per step, so every layer is sampled)."""
        if profile:
            L.dcx_profile_filter(dom_id)
            L.dcx_profile_sample(5)
            L.dcx_profile_enable(1)
        for rs in rs_of.values():
            rs.reset_stats()
        state["host_step_s"] = 0.0
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        state["host_step_s"] = time.perf_counter() - t0          # host time until the last step is enqueued (before the drain)
        fence()
        el = time.perf_counter() - t0
        state["el_local"] = el
        if dist_on:
            t = torch.tensor([el], dtype=torch.float64, device=dev if cx.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed_pass(steps)              # THE timed region: `value`
    per_rank_info = None
    if dist_on:
        # every rank's own clock and its exchange step, so that a multi-GPU line is diagnosable in one shot: ms_per_step of the
        # rank (before the MAX over ranks), duration of the side-stream exchange (event pair around all_gather_into_tensor + the
        # D2H of the gathered lists), and whether it stayed under a step's convolutions
        gm = np.asarray(og.gather_ms[-steps:] if og.gather_ms else [], dtype=np.float64)
        mine = {"rank": rank, "device": int(dev.index), "ms_per_step": round(1e3 * state["el_local"] / steps, 4),
                "gather_ms": ({"median": round(float(np.median(gm)), 4), "max": round(float(gm.max()), 4), "n": int(gm.size)} if gm.size else None)}
        mine["gather_overlapped"] = bool(og.overlapped and (gm.size == 0 or float(np.median(gm)) < mine["ms_per_step"]))
        per_rank_info = [None] * world
        dist.all_gather_object(per_rank_info, mine)
    # host / per-batch GPU timers of the timed region (N = 1: stream.ResidentStream keeps them)
    rs_stats = None
    if og is None:
        rs = rs_of[state["S"]]
        g = np.asarray(rs.gpu_ms, dtype=np.float64)
        rs_stats = {"gpu_ms": g, "host_enqueue_s": rs.host_enqueue_s, "host_wait_s": rs.host_wait_s,
                    "host_loop_s": state["host_step_s"]}

    # ---- what the timed steps produced (outside the timed region)
    if og is None:
        local = state["last"]                         # packed corner lists of the LAST timed step (pinned host copy)
        per_rank = [local]
    else:
        last = og.result(state["i"] - 1)              # (world, n_i32) as gathered by the LAST timed step
        per_rank = [last[r] for r in range(world)]
        local = per_rank[rank]
    if og is None and len(local) != n_i32:            # the last batch overflowed its pool and was re-run with the pool it asked for
        pool = (len(local) - 2 * B) // 6
    res_local, counts_local = unpack_results(local, B, pool, True)
    total_patches = float(min(int(counts_local.astype(np.int64).sum()), pool))

    timed_hdr = None
    if profile:
        timed_hdr = fetch_profile(L, total_patches)      # the dominant kernel as sampled INSIDE the timed region
        L.dcx_profile_enable(0)
    elapsed_1s = None
    if S > 1 and profile:
        # Per-kernel durations only mean something when nothing else shares the GPU: the roofline block is taken from a second
        # pass of the SAME K steps on ONE stream (its throughput is reported beside `value` as single_stream)
        state["S"] = 1
        for _ in range(2):
            step()
        fence()
        elapsed_1s = timed_pass(steps)
    el_roof = elapsed_1s if elapsed_1s is not None else elapsed

    roofline = None
    if profile:
        timed = fetch_profile(L, total_patches)
        L.dcx_profile_enable(0)
        L.dcx_profile_filter(-1)
        L.dcx_profile_sample(1)
        L.dcx_profile_enable(1)
        extra_steps = 3
        for _ in range(extra_steps):
            step()
        fence()
        full = fetch_profile(L, total_patches)
        L.dcx_profile_enable(0)
        kname = lambda k: L.dcx_profile_kernel_name(k).decode()
        if dom_id not in timed:
            dom_id = max(timed.items(), key=lambda kv: kv[1][1])[0]
        flop, msum, launches, clk = timed[dom_id]
        achieved = flop / (msum * 1e-3) / 1e12
        misc = {k: a for k, a in full.items() if k >= 100}      # conv1a of both nets, tail, finalize (dcx_prof_begin ids)
        full = {k: a for k, a in full.items() if k < 100}       # the convolution families
        conv_ms = sum(a[1] for a in full.values())
        conv_flop = sum(a[0] for a in full.values())
        # The Winograd families execute only part of a layer's ALGORITHMIC multiply-adds on the matrix cores (2-D F(2x2,3x3):
        # 4/9; phases x F(2x2,2x2) behind an up-sampling: 1/4), so the algorithmic rate (FLOP of the layer as written / time)
        # can exceed the fp32-MFMA peak; the roofline fraction is quoted on what the matrix pipe really ran.
        scale_of = lambda nm: 0.25 if "wino2p" in nm else 4.0 / 9.0 if "wino2h" in nm else 1.0   # executed / algorithmic MACs
        exec_scale = scale_of(kname(dom_id))
        conv_exec = sum(a[0] * scale_of(kname(k)) for k, a in full.items())
        algo = ("winograd F(2x2,3x3): 4/9 of the algorithmic MACs are executed" if "wino2h" in kname(dom_id)
                else "x2 up-sampled input as four phases x winograd F(2x2,2x2): 1/4 of the algorithmic MACs are executed" if "wino2p" in kname(dom_id)
                else "direct implicit GEMM")
        # `achieved` / `frac` are what the matrix pipe EXECUTED (<= peak by construction); the algorithmic rate of the layer as
        # written (what a direct convolution would have to sustain to be as fast) is reported beside it and may exceed the peak
        roofline = {"bound": "mfma", "kernel": kname(dom_id), "launches": launches, "algorithm": algo,
                    "achieved": round(achieved * exec_scale, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved * exec_scale / PEAK_F32_MFMA_TFLOPS, 4),
                    "executed_over_algorithmic_flop": round(exec_scale, 4),
                    "algorithmic_achieved": round(achieved, 2),
                    "algorithmic_over_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    "avg_launch_ms": round(msum / launches, 4),
                    "algorithmic_gflop_per_launch": round(flop / launches / 1e9, 3),
                    "traffic": pmc_traffic(kname(dom_id)),
                    "shader_clock_ghz": round(clk / msum, 3),
                    # executed matrix FLOP of one whole step / the step's wall time (timed region) / peak
                    "e2e_executed_frac": round(conv_exec / extra_steps / (elapsed / steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                    "all_conv_kernels": {"achieved": round(conv_exec / (conv_ms * 1e-3) / 1e12, 2),
                                         "ms_per_step": round(conv_ms / extra_steps, 3),
                                         "frac": round(conv_exec / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                         "algorithmic_achieved": round(conv_flop / (conv_ms * 1e-3) / 1e12, 2),
                                         "algorithmic_over_peak": round(conv_flop / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                         "source": f"{extra_steps} fully bracketed steps after the timed region"},
                    "per_kernel": {kname(k): {"ms_per_step": round(v[1] / extra_steps, 4),
                                              "algorithmic_tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2) if v[1] > 0 else None,
                                              "frac": round(v[0] * scale_of(kname(k)) / (v[1] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if v[1] > 0 else None,
                                              "launches_per_step": v[2] / extra_steps,
                                              "clock_ghz": round(v[3] / v[1], 3) if v[1] > 0 else None}
                                   for k, v in full.items()}}
        # ---- where a step's time goes.  Kernel durations: the 3 fully bracketed steps; wall / host: the timed region
        ms_step = 1e3 * el_roof / steps
        nonconv_ms = sum(a[1] for a in misc.values()) / extra_steps
        brk = {"ms_per_step": round(ms_step, 4),
               "conv_kernels_ms": round(conv_ms / extra_steps, 4),
               "non_conv_kernels_ms": round(nonconv_ms, 4),
               "non_conv_kernels": {kname(k): round(a[1] / extra_steps, 4) for k, a in sorted(misc.items())},
               # what is left of a step's wall time: launch gaps between the 22 kernels, the D2H copy of the packed result, event waits
               "gaps_and_d2h_ms": round(ms_step - conv_ms / extra_steps - nonconv_ms, 4),
               "kernel_launches_per_step": int(round((sum(a[2] for a in full.values()) + sum(a[2] for a in misc.values())) / extra_steps)),
               "shader_clock_ghz": round(clk / msum, 3),
               "source": f"kernel durations: hipEvent brackets around every launch of {extra_steps} extra steps after the timed region; "
                         "ms_per_step / host timers / per-batch GPU times: the timed region itself"}
        if rs_stats is not None and elapsed_1s is None and len(rs_stats["gpu_ms"]):
            g = rs_stats["gpu_ms"]
            brk.update({
                # per batch: from the moment its stream reached it to the last byte of its result in pinned memory
                "gpu_ms_per_batch": {"min": round(float(g.min()), 4), "median": round(float(np.median(g)), 4),
                                     "max": round(float(g.max()), 4), "first": round(float(g[0]), 4), "n": int(g.size),
                                     "all": [round(float(v), 3) for v in g[:64]]},
                # wall time of a step during which the stream had NO batch to work on (the host did not enqueue fast enough)
                "stream_starved_ms": round(ms_step - float(g.mean()), 4),
                "host_enqueue_ms": round(1e3 * rs_stats["host_enqueue_s"] / steps, 4),
                "host_wait_ms": round(1e3 * rs_stats["host_wait_s"] / steps, 4),
                "host_loop_ms": round(1e3 * rs_stats["host_loop_s"] / steps, 4)})
        roofline["step_conv_ms"] = brk["conv_kernels_ms"]
        roofline["step_non_conv_ms"] = brk["non_conv_kernels_ms"]
        roofline["step_gaps_and_d2h_ms"] = brk["gaps_and_d2h_ms"]
        if "host_enqueue_ms" in brk:
            roofline["step_host_enqueue_ms"] = brk["host_enqueue_ms"]
            roofline["step_stream_starved_ms"] = brk["stream_starved_ms"]
        state["breakdown"] = brk
        if elapsed_1s is not None:
            # S > 1: the block above is the single-stream pass; what the same kernel looked like INSIDE the timed region (its
            # launches share the CUs with the other stream's, so they last longer although the step is faster) is kept beside it
            roofline["measured_in"] = (f"a second pass of the same {steps} steps on ONE HIP stream right after the timed region "
                                       f"(per-kernel durations are only meaningful when nothing else shares the GPU); "
                                       f"the timed region itself keeps {S} batches in flight")
            roofline["e2e_executed_frac_single_stream"] = round(conv_exec / extra_steps / (elapsed_1s / steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
            if timed_hdr and dom_id in timed_hdr and timed_hdr[dom_id][1] > 0:
                f2, m2, l2, c2 = timed_hdr[dom_id]
                roofline["in_timed_region"] = {"launches": l2, "avg_launch_ms": round(m2 / l2, 4),
                                               "frac": round(f2 * exec_scale / (m2 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                               "shader_clock_ghz": round(c2 / m2, 3),
                                               "note": f"{S} batches in flight: launches of different batches overlap"}
        else:
            roofline["measured_in"] = "the timed region (one HIP stream)"
    if rank != 0:
        return None

    counts = np.concatenate([unpack_results(p, B, pool, True)[1] for p in per_rank])
    ids_seen = sorted({int(i) for r in res_local if r is not None and r.ndim == 2 for i in r[:, 2]})
    mean_k = float(counts.mean())
    # frames whose corners were NOT all refined in the timed step: only possible if a rank's whole batch overflowed its pool
    truncated = sum(int(r is None) for p in per_rank for r in unpack_results(p, B, pool, True)[0])
    fps = world * B * steps / elapsed
    gflop_frame = DET_GFLOP_240x320 * (H * Wd) / (240 * 320) + REF_GFLOP_PER_PATCH * mean_k

    # ---- parity of the timed batch: rank 0's own frames, and (N>1) frames of the LAST rank out of the gathered buffer
    oracle = Oracle(sd_dc, sd_rn)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    cpu = cpu_baseline(oracle, name, frames) if want_cpu_baseline else None     # also fills the oracle's result cache
    pick = list(range(min(n_check, B)))
    busiest = int(np.argmax(counts_local))           # the frame of the timed batch with the most corners is always checked
    if busiest not in pick:
        pick.append(busiest)
    if os.environ.get("DCX_BENCH_CORRUPT_PARITY"):   # test hook: prove that the gate gates (first checked frame that has corners)
        for b in pick:
            if res_local[b] is not None and res_local[b].ndim == 2:
                res_local[b] = res_local[b].copy(); res_local[b][0, 0] += 1.0
                break
    checks = [((name, 0, b), frames[b], res_local[b]) for b in pick]
    if dist_on:
        # frames of EVERY other rank out of the gathered buffer (2 per rank; the last rank gets as many as rank 0 / 2)
        for r in range(1, world):
            res_r = unpack_results(per_rank[r], B, pool, True)[0]
            n_r = max(2, n_check // 2) if r == world - 1 else 2
            for b in ([0, B - 1] if n_r == 2 else pick[:n_r]):
                if fixed_k:
                    fr = W.synthetic_frames(frames_kind, FRAME_SEED + r * 100000 + kept_all[r][b], 1, H, Wd)[0]   # frame b of rank r's batch
                else:
                    fr = frames[(b + r) % B]                                      # rank r runs rank 0's frames rotated by r
                checks.append(((name, r, b), fr, res_r[b]))
    parity = parity_block(oracle, checks)

    out = {
        "value": round(fps, 2), "unit": "frames/s", "ms_per_step": round(1e3 * elapsed / steps, 4), "steps": steps,
        "warmup": warmup,
        "settle": {"steps": state["settle_steps"], "seconds": SETTLE_S,
                   "why": "untimed set-up ahead of the W warm-up steps: GPU power state (clock ramp + one ~0.9 ms stall ~30 ms into a run)"
                          + (" and RCCL / c10d lazy initialisation" if dist_on else "")},
        "batches_in_flight": S,
        "config": {"workload": WL.workload_label(B, H, Wd, world, fixed_k), "preset": name,
                   "batch_per_gpu": B, "global_batch": B * world, "height": H, "width": Wd,
                   "corner_pool_per_gpu": pool, "corner_pool_note": "no per-frame cap: every firing cell of every frame is refined, as in the reference; the pool is shared by the batch",
                   "frames": frames_kind, "mean_corners_per_frame": round(mean_k, 2), "truncated_frames": truncated,
                   "corners_per_frame_min_max": [int(counts.min()), int(counts.max())],
                   "busiest_frame_checked": {"frame": busiest, "corners": int(counts_local[busiest])},
                   "distinct_ids_in_batch": len(ids_seen),
                   "weights": "numpy-seeded synthetic (seed 1234/1235), ids-head biases equalised per class (all ids fire), dust-bin bias calibrated to ~16 corners/frame"
                              + (f"; frames selected by the workload generator so that exactly {fixed_k} cells fire in each" if fixed_k else ""),
                   "parallelism": f"frames sharded, 1 process/GPU x{world}" + (", RCCL all-gather of corner lists on a side stream" if world > 1 else "")
                                  + (f"; consecutive batches alternate between {S} HIP streams ({S} batches in flight per GPU)" if S > 1 else ""),
                   "algorithmic_gflop_per_frame": round(gflop_frame, 3),
                   # the layers AS WRITTEN / peak: > 1 is possible because the Winograd families execute 4/9 (1/4) of those MACs;
                   # the executed fraction of the whole step is roofline.e2e_executed_frac
                   "e2e_algorithmic_over_peak": round(fps * gflop_frame / 1e3 / (PEAK_F32_MFMA_TFLOPS * world), 4)},
        "parity": parity,
    }
    if elapsed_1s is not None:
        out["single_stream"] = {"value": round(world * B * steps / elapsed_1s, 2), "unit": "frames/s",
                                "ms_per_step": round(1e3 * elapsed_1s / steps, 4), "steps": steps,
                                "note": "the same steps on ONE HIP stream (one batch in flight): the pass the roofline block is measured in"}
    if dist_on:
        out["gather_overlapped"] = bool(og.overlapped) and all(p["gather_overlapped"] for p in per_rank_info)
        out["per_rank"] = per_rank_info
    if roofline is not None:
        out["roofline"] = roofline
        out["step_breakdown"] = state.get("breakdown")
    if cpu is not None:
        out["cpu_baseline"] = cpu
    del dc, rn
    return out


def two_stream_pipelined(cx, steps=40, warmup=6):
    """cfg2's workload (the SAME 32 frames in every batch) with consecutive batches alternating between two HIP streams, i.e. two
    batches in flight (stream.ResidentStream(compute_streams=2) does the same).  Reported beside the one-stream headline; with
    equal work per batch it is within +-2 % of it -- the launches of one batch already keep every CU's two workgroup slots busy."""
    dev = cx.dev
    p = WL.PRESETS["cfg2"]
    B, H, Wd, kmax = p["batch"], p["height"], p["width"], p["kmax"]
    # EQUAL work in both streams: the same frames (rounds 2-5 took a second seed for stream 1, whose frames fire 9.9 instead of 16.0
    # cells each under weights calibrated on the first -- the "+6 ... +15 %" those rounds quoted was the lighter batch, not overlap)
    f0 = W.synthetic_frames("board", FRAME_SEED, B, H, Wd)
    frames = [f0, f0.copy()]
    sd_dc = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), torch.from_numpy(frames[0]).to(dev), dev, diverse_ids=True)
    sd_rn = W.synthetic_state_dict("refinenet", 1235)
    dc, rn = lModel(dcModel(16, sd_dc, dev)), lRefineNet(RefineNet(sd_rn, dev))
    d = [torch.from_numpy(f).to(dev) for f in frames]
    pool = B * kmax
    n = packed_len(B, pool)
    out = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(2)]
    host = [torch.empty((n,), dtype=torch.int32).pin_memory() for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def step(i):
        k = i & 1
        with torch.cuda.stream(streams[k]):
            infer_batch_device(d[k], 16, dc, rn, out=out[k], pool=pool)
            host[k].copy_(out[k], non_blocking=True)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    oracle = Oracle(sd_dc, sd_rn)
    checks = []
    for k in range(2):
        res, cnt = unpack_results(host[k].numpy(), B, pool, True)
        checks += [(("pipelined", k, b), frames[k][b], res[b]) for b in sorted({0, B - 1, int(np.argmax(cnt))})]
    return {"value": round(B * steps / el, 2), "unit": "frames/s", "ms_per_step": round(1e3 * el / steps, 4),
            "mode": "bs=32 320x240 batches (the same frames, i.e. equal work) alternating between two HIP streams (two batches in flight)",
            "parity": parity_block(oracle, checks)}


def bs1_reference_protocol(cx, n_iter=500):
    """The reference's own measurement (src/benchmark.py:37-53): ONE 320x240 BGR host image, 5 warm-up + n timed
    infer_image calls (BGR->gray, H2D, both nets, D2H, sort inside every call), fps = n / elapsed.  The image IS the
    reference's (src/reference/samples_test/IMG_7412.png, benchmark.py:34-35), carried as data by the golden fixture
    tests/golden/img7412_240x320.npz (written by oracle/make_golden.py from the reference's own infer_image); the weights are
    that fixture's (synthetic seed 1234, dust-bin bias calibrated so that 16 cells fire on this photo), so the result of every
    call is compared with what the REFERENCE returned, not only with the oracle."""
    dev = cx.dev
    fx = np.load(os.path.join(REPO, "tests", "golden", "img7412_240x320.npz"))
    meta = json.loads(str(fx["meta"]))
    sd_dc = W.synthetic_state_dict("detector", meta["wseed"], meta["n_ids"])
    sd_dc["convDb.bias"] = fx["convDb_bias"].astype(np.float32).copy()
    sd_rn = W.synthetic_state_dict("refinenet", meta["wseed"] + 1)
    if W.state_dict_sha256(sd_dc, "detector", meta["n_ids"]) != str(fx["sha_dc"]):
        raise SystemExit("bs1_reference_protocol: regenerated weights differ from the fixture's")
    dc, rn = lModel(dcModel(16, sd_dc, dev)), lRefineNet(RefineNet(sd_rn, dev))
    bgr = np.ascontiguousarray(fx["bgr_image"])
    for _ in range(5):
        kp, _ = infer_image(bgr, 16, dc, rn, draw_pred=False, device="cuda")
    t0 = time.time()
    for _ in range(n_iter):
        kp, _ = infer_image(bgr, 16, dc, rn, draw_pred=False, device="cuda")
    el = time.time() - t0
    exp = fx["final_rn"]
    same = kp.dtype == exp.dtype and kp.shape == exp.shape and np.array_equal(kp, exp)
    par = {"frames_checked": 1, "corners": int(exp.shape[0]), "mismatched_frames": 0 if same else 1,
           "against": "what the REFERENCE's infer_image returned for this image and these weights (tests/golden/img7412_240x320.npz): "
                      "ids, cells and sub-pixel xy identical"}
    return {"value": round(n_iter / el, 1), "unit": "frames/s", "ms_per_call": round(1e3 * el / n_iter, 4), "iters": n_iter,
            "protocol": "src/benchmark.py:37-53: bs=1 infer_image loop from one BGR host image, 5 warm-up calls",
            "image": "the reference's benchmark image IMG_7412.png (320x240 colour photo, benchmark.py:34-35)",
            "corners": int(kp.shape[0]) if kp.ndim == 2 else 0, "parity": par,
            "vs_reference_readme_200fps": round(n_iter / el / REFERENCE_README_FPS, 2)}


def self_launch(n, backend, argv):
    """`python bench.py --gpus N` from a bare shell (no RANK/WORLD_SIZE in the environment) for N > 1: start the N ranks
    ourselves under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a free port), forward every flag,
    let the ranks' stdout through (rank 0 prints the single JSON line) and return the launcher's exit code.  RCCL needs one
    GPU per rank: with fewer GPUs visible this is an error, never a silent gloo run."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < n:
        print(f"bench.py: --gpus {n} with backend nccl (RCCL) but only {ndev} GPU(s) visible; "
              "pass --backend gloo explicitly to smoke-test the multi-process flow on fewer GPUs", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(WL.PRESETS), help="BASELINE config (per-GPU load); default cfg2 = configs[1]")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (overrides the preset)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--kmax", type=int, default=None, help="AVERAGE corners per frame the corner pool of a batch is sized for (pool = batch * kmax; no per-frame cap)")
    ap.add_argument("--frames", default=None, choices=["board", "board4", "noise"])
    ap.add_argument("--fixed-k", type=int, default=None, help="select frames with exactly this many corners (cfg5: 16)")
    ap.add_argument("--streams", type=int, default=1,
                    help="batches in flight per GPU: consecutive steps alternate between this many HIP streams (stream.ResidentStream). "
                         "Default 1: with EQUAL work in every batch a second stream gains nothing (-1 ... -4 %, "
                         "profiles/experiments/r05_batches_in_flight_equal_work.txt); > 1 adds a single-stream pass for the roofline block")
    ap.add_argument("--parity-frames", type=int, default=8)
    ap.add_argument("--settle", type=float, default=None,
                    help=f"seconds of untimed steps ahead of the warm-up steps (default {SETTLE_S}; 0 = none: the cold A/B that shows the "
                         "GPU's clock ramp inside the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip other_configs (the other BASELINE configs, bs=1 protocol)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (process group, side-stream all-gather, barrier, MAX all-reduce) even with one rank: "
                         "RCCL self-check on a 1-GPU box")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo only to smoke-test the multi-process flow "
                         "with several ranks on one GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus, args.backend, sys.argv[1:]))

    cx = Ctx()
    cx.rank = rank = int(os.environ.get("RANK", "0"))
    cx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cx.backend = args.backend
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but only {ndev} GPUs visible: RCCL needs one GPU per rank (no silent gloo fallback; "
                         "--backend gloo is an explicit smoke-test mode)")
    torch.cuda.set_device(local_rank % ndev)
    cx.dev = dev = torch.device("cuda", local_rank % ndev)
    cx.dist = None
    cx.force_dist = bool(args.force_dist)
    cx.streams = args.streams
    if args.settle is not None:
        globals()["SETTLE_S"] = max(0.0, float(args.settle))
    dist_on = world > 1 or cx.force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        cx.dist = dist
        # what the process group really is: rank count seen by RCCL / gloo and the device every rank computes on
        objs = [None] * world
        dist.all_gather_object(objs, {"rank": rank, "device": int(dev.index), "pid": os.getpid()})
        cx.ranks_seen = {"world_size": int(dist.get_world_size()), "backend": dist.get_backend(),
                         "device_of_rank": [o["device"] for o in objs], "visible_gpus": ndev,
                         "distinct_processes": len({o["pid"] for o in objs})}
    cx.L = _lib.lib()

    p = dict(WL.PRESETS[args.config])
    for k_, v in (("batch", args.batch), ("height", args.height), ("width", args.width), ("kmax", args.kmax),
                  ("frames", args.frames), ("fixed_k", args.fixed_k)):
        if v is not None:
            p[k_] = v
    main_res = run_config(cx, args.config, p["batch"], p["height"], p["width"], p["kmax"], p["frames"], p["fixed_k"],
                          args.steps, args.warmup, not args.no_profile, args.parity_frames,
                          want_cpu_baseline=(world == 1 and not args.no_cpu_baseline))

    others = {}
    if not args.no_extras:
        # the other BASELINE configs at their per-GPU size, short runs (5 warm-up protocol of src/benchmark.py kept)
        todo = [n for n in ("cfg3", "cfg4", "cfg5") if n != args.config] if world == 1 else \
               [n for n in ("cfg4", "cfg5") if n != args.config]
        for n in todo:
            q = WL.PRESETS[n]
            st = 10 if q["height"] <= 240 else 5
            # single GPU: every config carries its own roofline block (dominant kernel, per-kernel table, e2e executed fraction)
            r = run_config(cx, n, q["batch"], q["height"], q["width"], q["kmax"], q["frames"], q["fixed_k"], st, 3,
                           world == 1 and not args.no_profile, 4)
            if rank == 0:
                r.pop("steps", None)
                others[n] = r
        if world == 1:
            others["cfg2_two_batches_in_flight"] = two_stream_pipelined(cx)
            others["bs1_reference_protocol"] = bs1_reference_protocol(cx)

    if rank != 0:
        if dist_on:
            cx.dist.barrier()
            cx.dist.destroy_process_group()
        return

    fps = main_res["value"]
    line = {
        "metric": METRIC, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True,
        "scaling": "weak",
        # no number is published for THIS metric (batched, HBM-resident frames): the reference's only figure is "> 200 fps" for
        # its bs=1 infer_image loop on a GTX1080Ti -- the like-for-like ratio is other_configs.bs1_reference_protocol
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": dict(main_res["config"], vs_baseline_note="null: BASELINE.json publishes nothing for bs=32; the reference README's "
                       "'>200 fps' (GTX1080Ti, bs=1, src/benchmark.py) is compared like for like in other_configs.bs1_reference_protocol"),
        "parity": main_res["parity"],
    }
    if dist_on:
        line["ranks"] = cx.ranks_seen
    for k_ in ("settle", "batches_in_flight", "single_stream", "gather_overlapped", "per_rank", "step_breakdown", "roofline", "cpu_baseline"):
        if k_ in main_res:
            line[k_] = main_res[k_]
    if others:
        line["other_configs"] = others
    try:
        C.CDLL(None).fflush(None)      # RCCL's version banner sits in the C stdout buffer until exit: flush it now, so that the JSON
    except Exception:                  # line is the LAST line of stdout (a driver that reads the last line must find it)
        pass
    print(json.dumps(line), flush=True)
    bad = main_res["parity"]["mismatched_frames"] + sum(v.get("parity", {}).get("mismatched_frames", 0) for v in others.values())
    if dist_on:
        cx.dist.barrier()
        cx.dist.destroy_process_group()
    if bad:
        print(f"PARITY FAILURE: {bad} frame(s) differ from the oracle", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
