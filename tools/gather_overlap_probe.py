#!/usr/bin/env python3
"""What does the side-stream gather cost a step?  One rank, RCCL process group of size 1, bs=32 320x240; variants of
OverlappedGather.launch().  usage: python tools/gather_overlap_probe.py"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import weights as W, workload as WL
from deepcharuco_amd.inference import infer_batch_device
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
from deepcharuco_amd.inference import packed_len
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B, kmax = 32, 64
frames = torch.from_numpy(W.synthetic_frames("board", 1000, B, 240, 320)).to(dev)
sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), frames, dev)
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
n = packed_len(B, B * kmax)
side = torch.cuda.Stream()
out = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(2)]
gath = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(2)]
host = [torch.empty((n,), dtype=torch.int32).pin_memory() for _ in range(2)]
evc = [torch.cuda.Event() for _ in range(2)]; evs = [torch.cuda.Event() for _ in range(2)]

def run(variant, steps=40):
    used = [False, False]
    def step(i):
        s = i % 2
        cur = torch.cuda.current_stream(dev)
        if used[s] and variant not in ("none", "same_stream"):
            cur.wait_event(evs[s])
        infer_batch_device(frames, 16, dc, rn, kmax, out=out[s])
        if variant == "none":
            host[s].copy_(out[s], non_blocking=True); return
        if variant == "same_stream":
            dist.all_gather_into_tensor(gath[s], out[s]); host[s].copy_(gath[s], non_blocking=True); return
        evc[s].record(cur)
        with torch.cuda.stream(side):
            side.wait_event(evc[s])
            if variant == "full":
                w = dist.all_gather_into_tensor(gath[s], out[s], async_op=True); w.wait(); host[s].copy_(gath[s], non_blocking=True)
            elif variant == "no_d2h":
                w = dist.all_gather_into_tensor(gath[s], out[s], async_op=True); w.wait()
            elif variant == "sync_op":
                dist.all_gather_into_tensor(gath[s], out[s]); host[s].copy_(gath[s], non_blocking=True)
            elif variant == "d2h_only":
                host[s].copy_(out[s], non_blocking=True)
            elif variant == "copy_kernel":      # device copy instead of the collective (what a one-rank all-gather amounts to)
                gath[s].copy_(out[s]); host[s].copy_(gath[s], non_blocking=True)
            evs[s].record(side)
        used[s] = True
    for i in range(5): step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps): step(i)
    t_host = (time.perf_counter() - t) / steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print(f"{variant:12s} {1e3 * dt:.3f} ms/step  {B / dt:8.1f} fps   (host enqueue {1e3 * t_host:.3f} ms/step)")

def barrier_cost():
    for k in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); dist.barrier(); torch.cuda.synchronize()
        print(f"dist.barrier() #{k}: {1e3 * (time.perf_counter() - t):.3f} ms")
    x = torch.zeros(1, dtype=torch.float64, device=dev)
    for k in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); dist.all_reduce(x, op=dist.ReduceOp.MAX); v = float(x.item())
        print(f"all_reduce + item #{k}: {1e3 * (time.perf_counter() - t):.3f} ms")

barrier_cost()
for rep in range(6):
    run("full", steps=30)
barrier_cost()
dist.destroy_process_group()
