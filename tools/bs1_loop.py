#!/usr/bin/env python3
"""bs=1 HBM-resident pipeline loop for rocprofv3 (per-kernel times of the reference's one-frame-per-call protocol).
usage: rocprofv3 --kernel-trace --stats -d out -o bs1 -- python tools/bs1_loop.py [iters] [batch]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import weights as W, workload as WL
from deepcharuco_amd.inference import infer_batch_device
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
frames = W.synthetic_frames("board", 1000, 32, 240, 320)
sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), torch.from_numpy(frames).to(dev), dev)
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
d = torch.from_numpy(frames[:B]).to(dev)
out = None
for _ in range(iters):
    out = infer_batch_device(d, 16, dc, rn, 64, out=out)
torch.cuda.synchronize()
print("corners", out[:B].cpu().tolist())
