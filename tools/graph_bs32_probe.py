#!/usr/bin/env python3
"""Does replaying the bs=32 step as ONE hipGraph beat 22 eager launches (inter-kernel gaps)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import weights as W, workload as WL
from deepcharuco_amd.inference import infer_batch_device, packed_len
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = torch.from_numpy(W.synthetic_frames("board", 1000, B, 240, 320)).to(dev)
sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), frames, dev, diverse_ids=True)
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
pool = B * 64
out = torch.empty(packed_len(B, pool), dtype=torch.int32, device=dev)
host = torch.empty(packed_len(B, pool), dtype=torch.int32).pin_memory()
st = torch.cuda.Stream()
def eager():
    infer_batch_device(frames, 16, dc, rn, out=out, pool=pool)
    host.copy_(out, non_blocking=True)
with torch.cuda.stream(st):
    for _ in range(3): eager()
st.synchronize()
ref = host.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    eager()
def bench(name, fn, steps=60):
    with torch.cuda.stream(st):
        for _ in range(5): fn()
        st.synchronize()
        best = 1e9
        for _ in range(4):
            t = time.perf_counter()
            for _ in range(steps): fn()
            st.synchronize()
            best = min(best, (time.perf_counter() - t) / steps)
    print(f"{name:40s} {B / best:9.1f} fps  {1e3 * best:.4f} ms/step", flush=True)
bench("eager launches", eager)
bench("one hipGraph replay per step", g.replay)
bench("eager launches (again)", eager)
bench("one hipGraph replay per step (again)", g.replay)
print("same counts:", bool((host[:B] == ref[:B]).all()))
