#!/usr/bin/env python3
"""bs=1 reference protocol (src/benchmark.py:37-53) under the graph's zero-copy modes (DCX_GRAPH_ZEROCOPY bit 0: read the frame
from pinned host memory, bit 1: write the corner list to pinned host memory)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import weights as W, workload as WL
from deepcharuco_amd.graph import clear_graph_cache
from deepcharuco_amd.inference import infer_image
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
frames = W.synthetic_frames("board", 1000, 32, 240, 320)
sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), torch.from_numpy(frames).to(dev), dev, diverse_ids=True)
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
counts = WL.frame_counts(torch.from_numpy(frames).to(dev), dc)
pick = int(np.argmin(np.abs(counts.astype(np.int64) - 16)))
bgr = np.ascontiguousarray(np.repeat(frames[pick][..., None], 3, axis=2))
ref = None
for zc in (0, 1, 2, 3, 0, 3):
    os.environ["DCX_GRAPH_ZEROCOPY"] = str(zc)
    clear_graph_cache()
    for _ in range(20):
        kp, _ = infer_image(bgr, 16, dc, rn, device="cuda")
    best = 1e9
    for _ in range(3):
        t = time.time()
        for _ in range(n):
            kp, _ = infer_image(bgr, 16, dc, rn, device="cuda")
        best = min(best, (time.time() - t) / n)
    if ref is None:
        ref = kp
    print(f"zero_copy={zc}: {1 / best:8.1f} calls/s  {1e6 * best:.1f} us/call  corners {kp.shape[0]}  same {np.array_equal(kp, ref)}", flush=True)
