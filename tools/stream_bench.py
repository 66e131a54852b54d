#!/usr/bin/env python3
"""PCIe-inclusive throughput: host numpy frames -> FrameStream (pinned, double-buffered) -> host keypoint arrays,
next to the synchronous infer_batch from host memory.  bs=32, 320x240."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import weights as W, workload as WL
from deepcharuco_amd.inference import infer_batch
from deepcharuco_amd.stream import FrameStream
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
dev = torch.device("cuda", 0)
B, NB = 32, 40
frames = W.synthetic_frames("board", 1000, B, 240, 320)
sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), torch.from_numpy(frames).to(dev), dev)   # mean 16 corners per frame
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
for _ in range(3): r = infer_batch(frames, 16, dc, rn)
print("corners/frame", np.mean([0 if a.ndim == 1 else a.shape[0] for a in r]))
t = time.time()
for _ in range(NB): infer_batch(frames, 16, dc, rn)
dt = time.time() - t
print(f"infer_batch from host memory (synchronous, pageable H2D + D2H + unpack): {B * NB / dt:.0f} fps")
for depth, cs in ((1, 1), (2, 1), (3, 1), (3, 2), (4, 2)):
    fs = FrameStream(16, dc, rn, batch=B, height=240, width=320, depth=depth, compute_streams=cs)
    list(fs.run([frames] * 3))
    t = time.time()
    n = sum(len(item[1]) for item in fs.run([frames] * NB))
    dt = time.time() - t
    print(f"FrameStream depth={depth} compute_streams={cs} (pinned, async H2D/compute/D2H, host unpack incl.): {n / dt:.0f} fps")
