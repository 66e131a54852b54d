#!/usr/bin/env python3
"""Time ONE layer shape under every Winograd tile that can run it (DCX_FORCE_CFG, per-launch hipEvent brackets of the profile hooks):
is the cost model's pick the fastest?   usage (MI355X): python tools/conv_shape_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from deepcharuco_amd import _lib
import test_gpu_parity as T
L = _lib.lib()
dev = torch.device("cuda", 0)
W2H = ["dcx_conv_wino2h_kernel<DcxWino2hCfg<8,16,%d,1>>", "dcx_conv_wino2h_kernel<DcxWino2hCfg<6,20,%d,1>>", "dcx_conv_wino2h_kernel<DcxWino2hCfg<8,8,%d,1,1>>"]
SHAPES = [("det conv4a/4b bs32", 32, 128, 128, 30, 40, 0), ("det heads bs32", 32, 128, 512, 30, 40, 0), ("det conv3a bs32", 32, 64, 128, 60, 80, 0),
          ("det conv3b bs32 (pool)", 32, 128, 128, 60, 80, 1), ("det conv2a bs32", 32, 64, 64, 120, 160, 0),
          ("ref conv1b 512 (valid 22->20)", 512, 64, 64, 22, 22, 0), ("ref conv2a 512 (valid 20->18)", 512, 64, 128, 20, 20, 0),
          ("ref conv4b 512 16x16", 512, 128, 128, 16, 16, 0), ("ref conv5b 512 32x32", 512, 64, 64, 32, 32, 0)]
for name, n, cin, cout, h, w, pool in SHAPES:
    g = torch.Generator().manual_seed(1)
    pad = 0 if "valid" in name else 1
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    os.environ.pop("DCX_FORCE_CFG", None)
    pick = L.dcx_conv_pick_name(n, cin, ho, wo, cout, 3, pool, 0).decode()
    res = {}
    for cfg in [c % pool for c in W2H]:
        os.environ["DCX_FORCE_CFG"] = cfg
        for _ in range(2):
            T._conv_layer(x, wt, b, bn, pad, 0, bool(pool), 3)
        L.dcx_profile_filter(-1); L.dcx_profile_sample(1); L.dcx_profile_enable(1)
        for _ in range(7):
            T._conv_layer(x, wt, b, bn, pad, 0, bool(pool), 3)
        torch.cuda.synchronize()
        k = L.dcx_profile_count()
        ids = (C.c_int * k)(); ni = (C.c_int * k)(); lim = (C.c_int * k)(); fl = (C.c_double * k)(); ms = (C.c_float * k)()
        k = L.dcx_profile_fetch(ids, ni, lim, fl, ms, k)
        L.dcx_profile_enable(0)
        names = {L.dcx_profile_kernel_name(ids[i]).decode() for i in range(k)}
        res[cfg] = (float(np.median([ms[i] for i in range(k)])) * 1e3, names == {cfg})
    os.environ.pop("DCX_FORCE_CFG", None)
    flop = 2.0 * cout * cin * 9 * ho * wo * n * 4 / 9
    print(f"{name:34s} pick {pick.split('Cfg<')[1][:-2]:12s} | " + "  ".join(f"{c.split('Cfg<')[1][:-2]}: {t:7.1f} us ({flop / t / 1e6 / 157.3:.2f}){'' if ok else ' [not run]'}" for c, (t, ok) in res.items()), flush=True)
