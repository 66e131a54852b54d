#!/usr/bin/env python3
"""Does overlapping kernel tails/ramps across two streams help?  1 stream x B=32 vs 2 streams x B=16 (and x B=32)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import _lib, weights as W
from deepcharuco_amd.inference import infer_batch_device
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
dev = torch.device("cuda", 0)
frames = torch.from_numpy(W.synthetic_frames("board", 1000, 128, 240, 320)).to(dev)
sd = W.synthetic_state_dict("detector", 1234); sd["convDb.bias"][16] += np.float32(3.75)
sd_rn = W.synthetic_state_dict("refinenet", 1235)
# one model pair per stream (the pipeline workspace is owned by the detector object and keyed by the current stream)
pairs = [(lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(sd_rn, dev))) for _ in range(4)]

def run(nstreams, B, steps=20):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = [None] * nstreams
    def one():
        for s in range(nstreams):
            with torch.cuda.stream(streams[s]):
                outs[s] = infer_batch_device(frames[s * B:(s + 1) * B], 16, pairs[s][0], pairs[s][1], 64, out=outs[s])
    for _ in range(3): one()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps): one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    k = float(np.mean([int(x) for o in outs for x in o[:B].cpu().numpy()]))
    print(f"{nstreams} stream(s) x B={B}: {nstreams * B * steps / dt:8.1f} fps   ({1e3 * dt / steps:.3f} ms per round, {k:.1f} corners/frame)")

run(1, 32); run(2, 16); run(4, 8); run(1, 32); run(2, 16); run(3, 11); run(2, 32); run(4, 16); run(1, 64)
