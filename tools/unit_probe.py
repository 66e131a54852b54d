"""Kernel-tuning aid: per-unit cycle counts of workgroup 0 for the conv launches of one pipeline step."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import _lib, weights as W
from deepcharuco_amd.inference import infer_batch_device
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
dev = torch.device("cuda", 0)
L = _lib.lib()
B = int(os.environ.get("PROBE_B", "32"))
frames = torch.from_numpy(W.synthetic_frames("board", 1000, B, 240, 320)).to(dev)
sd = W.synthetic_state_dict("detector", 1234); sd["convDb.bias"][16] += np.float32(3.0)
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
for _ in range(3): infer_batch_device(frames, 16, dc, rn, 64)
torch.cuda.synchronize()
L.dcx_profile_enable(1)
infer_batch_device(frames, 16, dc, rn, 64)
torch.cuda.synchronize()
n = L.dcx_profile_count()
L.dcx_profile_enable(0)
ids = (C.c_int * n)(); nimg = (C.c_int * n)(); lim = (C.c_int * n)(); fl = (C.c_double * n)(); ms = (C.c_float * n)()
L.dcx_profile_fetch(ids, nimg, lim, fl, ms, n)
for i in range(n):
    w = (C.c_ulonglong * 64)()
    L.dcx_profile_probe_words(i, w)
    name = L.dcx_profile_kernel_name(ids[i]).decode()[31:-2]
    t = np.array(w[4:64], dtype=np.int64).reshape(20, 3)
    valid = (t[:, 0] > 0).sum()
    if valid < 3: 
        print(f"{i:2d} {name:40s} ms {ms[i]:.3f} units<3"); continue
    t = t[:valid]
    barrier = t[:, 1] - t[:, 0]
    loop = t[:, 2] - t[:, 1]
    tail = np.r_[t[1:, 0] - t[:-1, 2], 0]      # after-loop (epilogue or nothing) until next unit's barrier
    clk = (w[2] - w[0]) / max(w[3] - w[1], 1) * 0.1
    print(f"{i:2d} {name:40s} ms {ms[i]:.3f} wg0 alive {(w[3] - w[1]) * 1e-5:.3f} ms = {w[2] - w[0]} cyc  clk {clk:.2f} units {valid}: loop {np.median(loop):.0f} (min {loop.min()} max {loop.max()}) "
          f"barrier med {np.median(barrier):.0f} max {barrier.max()}  tail {tail[:8].tolist()}")
