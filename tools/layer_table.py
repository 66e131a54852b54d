#!/usr/bin/env python3
"""Per-launch table of one detect+refine step (launch order): kernel, images, algorithmic GFLOP, ms, TFLOP/s, clock.
    python tools/layer_table.py [batch] [height] [width]
Every launch is hipEvent-bracketed by the library's profile hooks (dcx_profile_*), so the step itself runs slower than
the un-instrumented one; the per-launch durations are what this prints (median over the repeats)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import _lib, weights as W, workload as WL
from deepcharuco_amd.inference import infer_batch_device
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 240
Wd = int(sys.argv[3]) if len(sys.argv) > 3 else 320
REPS = 7
dev = torch.device("cuda", 0)
L = _lib.lib()
calib = torch.from_numpy(W.synthetic_frames("board", 1000, min(B, 128), H, Wd)).to(dev)
sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), calib, dev)
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
frames = torch.from_numpy(W.synthetic_frames("board", 1000, B, H, Wd)).to(dev)
out = None
for _ in range(3):
    out = infer_batch_device(frames, 16, dc, rn, 64, out=out)
torch.cuda.synchronize()
patches = int(out[:B].sum().item())
rows = None
for rep in range(REPS):
    L.dcx_profile_filter(-1)
    L.dcx_profile_enable(1)
    infer_batch_device(frames, 16, dc, rn, 64, out=out)
    torch.cuda.synchronize()
    L.dcx_profile_enable(0)
    n = L.dcx_profile_count()
    ids = (C.c_int * n)(); nimg = (C.c_int * n)(); lim = (C.c_int * n)()
    fl = (C.c_double * n)(); ms = (C.c_float * n)(); ghz = (C.c_float * n)()
    n = L.dcx_profile_fetch(ids, nimg, lim, fl, ms, n)
    L.dcx_profile_clocks(ghz, n)
    if rows is None:
        rows = [{"k": L.dcx_profile_kernel_name(ids[i]).decode(), "imgs": patches if lim[i] else nimg[i],
                 "gf": fl[i] * (patches if lim[i] else nimg[i]) / 1e9, "ms": [], "ghz": []} for i in range(n)]
    for i in range(n):
        rows[i]["ms"].append(ms[i]); rows[i]["ghz"].append(ghz[i])
tot = 0.0
print(f"bs={B} {H}x{Wd}, {patches} live patches; conv launches in order (median of {REPS})")
for r in rows:
    m = float(np.median(r["ms"])); tot += m
    name = r["k"].replace("dcx_conv_", "").replace("_kernel", "").replace("DCX_EPI_", "")
    print(f"  {name:58s} imgs {r['imgs']:5d}  {r['gf']:8.2f} GFLOP  {m * 1e3:8.1f} us  {r['gf'] / m:7.1f} TFLOP/s  clk {float(np.median(r['ghz'])):.2f}")
print(f"  sum {tot:.3f} ms")
