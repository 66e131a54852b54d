#!/usr/bin/env python3
"""One bs=32 batch as two half-batches on two HIP streams, with a fork/join per step (what a single infer_batch call could do
inside): plain, with stream priorities, and free-running (no join = "two batches in flight" at half size) for comparison."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import weights as W, workload as WL
from deepcharuco_amd.inference import infer_batch_device
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = torch.from_numpy(W.synthetic_frames("board", 1000, B, 240, 320)).to(dev)
sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), frames, dev, diverse_ids=True)
sd_rn = W.synthetic_state_dict("refinenet", 1235)
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(sd_rn, dev))

def bench(name, step, steps=40):
    for _ in range(4): step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / steps)
    print(f"{name:58s} {B / best:9.1f} fps  {1e3 * best:.3f} ms/step", flush=True)

out1 = None
def single():
    global out1
    out1 = infer_batch_device(frames, 16, dc, rn, 64, out=out1)
bench("1 stream, whole batch", single)

def make_split(parts, prio=None, join=True, stagger=0):
    streams = [torch.cuda.Stream(priority=(prio[i] if prio else 0)) for i in range(parts)]
    outs = [None] * parts
    lo = [B * i // parts for i in range(parts + 1)]
    e0 = torch.cuda.Event(); ends = [torch.cuda.Event() for _ in range(parts)]
    def step():
        main = torch.cuda.current_stream()
        if join:
            e0.record(main)
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                if join:
                    st.wait_event(e0)
                outs[i] = infer_batch_device(frames[lo[i]:lo[i + 1]], 16, dc, rn, 64, out=outs[i])
                if join:
                    ends[i].record(st)
        if join:
            for e in ends:
                main.wait_event(e)
    return step
bench("2 halves, fork/join per step", make_split(2))
bench("2 halves, fork/join, priorities (high, low)", make_split(2, prio=(-1, 0)))
bench("2 halves (3/4 + 1/4)...skipped", lambda: None) if False else None
bench("4 quarters, fork/join per step", make_split(4))
bench("2 halves, free running (no join)", make_split(2, join=False))
bench("1 stream, whole batch (again)", single)
