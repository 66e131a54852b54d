"""Long-running parity stress (not part of the test suite): many seeded batches through the default (Winograd) path vs the
oracle, frame by frame.  usage: python tools/stress_parity.py [rounds]   (run on an MI355X; ~10 s per round)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from deepcharuco_amd import weights as W
from deepcharuco_amd.inference import infer_batch
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
from oracle import deepcharuco_oracle as O

dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tot_frames = tot_corners = bad = 0
for r in range(rounds):
    h, w = [(240, 320), (120, 160), (480, 640), (64, 96)][r % 4]
    n = {240: 48, 120: 64, 480: 12, 64: 64}[h]
    frames = np.concatenate([W.synthetic_frames("noise", 9000 + 100 * r, n // 2, h, w),
                             W.synthetic_frames("board", 9500 + 100 * r, n // 2, h, w)])
    sd_dc = T._calibrated(3000 + r, frames[:: max(1, n // 8)], target_per_frame=12)
    sd_rn = W.synthetic_state_dict("refinenet", 4000 + r)
    dc, rn = lModel(dcModel(16, sd_dc, dev)), lRefineNet(RefineNet(sd_rn, dev))
    got = infer_batch(frames, 16, dc, rn, kmax=64)
    t_dc, t_rn = O.to_torch_state_dict(sd_dc), O.to_torch_state_dict(sd_rn)
    for b in range(len(frames)):
        exp = O.infer_image(None, 16, t_dc, t_rn, gray=frames[b])
        tot_frames += 1
        tot_corners += 0 if exp.ndim == 1 else exp.shape[0]
        if got[b].shape != exp.shape or not np.array_equal(got[b], exp):
            bad += 1
    print(f"round {r}: {h}x{w} x{n}  cumulative frames {tot_frames} corners {tot_corners} mismatched frames {bad}", flush=True)
print("RESULT", tot_frames, tot_corners, bad)
