#!/usr/bin/env python3
"""How many batches in flight pay?  stream.ResidentStream (HBM-resident bs=B batches on alternating HIP streams) at
compute_streams = 1 .. 4, raw and unpacked hand-out, best of 3 x N batches.   python tools/resident_stream_probe.py [B] [N]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import weights as W, workload as WL
from deepcharuco_amd.stream import ResidentStream
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
fr = [W.synthetic_frames("board", 1000 + 500 * i, B, 240, 320) for i in range(2)]
sd_dc = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), torch.from_numpy(fr[0]).to(dev), dev, diverse_ids=True)
dc, rn = lModel(dcModel(16, sd_dc, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
d = [torch.from_numpy(f).to(dev) for f in fr]
print("corners per frame: set 0", float(WL.frame_counts(d[0], dc).mean()), " set 1", float(WL.frame_counts(d[1], dc).mean()), flush=True)
if os.environ.get("PROBE_SAME_FRAMES", "1") == "1":
    d[1] = d[0].clone()          # the SAME work in every batch (a second frame set fires a different number of corners)
torch.cuda.synchronize()


def run(streams, depth, raw):
    rs = ResidentStream(16, dc, rn, batch=B, height=240, width=320, kmax=64, compute_streams=streams, depth=depth, raw=raw)
    best = 0.0
    for rep in range(4):
        n = 8 if rep == 0 else N
        t0 = time.perf_counter()
        for i in range(n):
            rs.submit(d[i & 1])
        for _ in rs.flush():
            pass
        torch.cuda.synchronize()
        if rep:
            best = max(best, B * n / (time.perf_counter() - t0))
    print(f"streams {streams} depth {rs.depth} {'raw     ' if raw else 'unpacked'}: {best:9.1f} fps   {1e3 * B / best:.3f} ms/batch", flush=True)


for streams, depth, raw in [(1, None, True), (2, None, True), (3, None, True), (4, None, True), (2, 2, True), (2, 3, True), (2, 6, True),
                            (1, None, False), (2, None, False), (3, None, False), (1, None, True), (2, None, True)]:
    run(streams, depth, raw)
