#!/usr/bin/env python3
"""A/B of the per-XCD item shares (dcx_calibrate_xcd) on the cfg2 step (bs=32 320x240, the product's resident-stream caller):
equal shares vs calibrated ones, alternating, per-batch GPU times from timing-enabled events.   usage: python tools/xcd_calib_probe.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcharuco_amd import weights as W, workload as WL
from deepcharuco_amd.inference import calibrate_xcd, get_xcd_weights, set_xcd_weights
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
from deepcharuco_amd.stream import ResidentStream
dev = torch.device("cuda", 0)
B, H, Wd = 32, 240, 320
frames = W.synthetic_frames("board", 1000, B, H, Wd)
d = torch.from_numpy(frames).to(dev)
sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), d, dev, diverse_ids=True)
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
rs = ResidentStream(16, dc, rn, batch=B, height=H, width=Wd, raw=True, timing=True)


def run(n):
    rs.reset_stats()
    t0 = time.perf_counter()
    last = None
    for _ in range(n):
        r = rs.submit(d)
        last = r[1] if r is not None else last
    for r in rs.flush():
        last = r[1]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return 1e3 * el / n, float(np.median(rs.gpu_ms)), last


run(80)                                             # settle
base = run(40)[2]
cal = []
for i in range(3):
    cal.append(calibrate_xcd(dev, 8))
    print("calibration", i, " ".join(f"{v:.4f}" for v in cal[-1]), flush=True)
w = np.mean(cal, axis=0)
if os.environ.get("BT_WEIGHTS"):
    w = np.array([float(v) for v in os.environ["BT_WEIGHTS"].split(",")])
print("weights used:", " ".join(f"{v:.4f}" for v in w))
for rnd in range(4):
    set_xcd_weights(None, dev)
    a = run(60)
    set_xcd_weights(w, dev)
    b = run(60)
    same = np.array_equal(np.sort(a[2][:B]), np.sort(b[2][:B]))
    print(f"round {rnd}: equal shares {a[0]:.4f} ms/step (median batch {a[1]:.4f})   calibrated {b[0]:.4f} ms/step (median batch {b[1]:.4f})"
          f"   -> {100 * (a[1] / b[1] - 1):+.2f} %   counts identical: {same}", flush=True)
cal2 = calibrate_xcd(dev, 8)
print("calibration after the A/B:", " ".join(f"{v:.4f}" for v in cal2))
