"""Kernel bring-up aid: run ONE conv instantiation (DCX_FORCE_CFG) on a few shapes and locate elements that differ from its
exact-order C restatement.  usage (MI355X): python tools/conv_debug.py "dcx_conv_wino2h_kernel<DcxWino2hCfg<8,16,0,1>>" """
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
os.environ["DCX_FORCE_CFG"] = sys.argv[1] if len(sys.argv) > 1 else "dcx_conv_wino2h_kernel<DcxWino2hCfg<8,16,0,1>>"
import test_gpu_parity as T
from oracle.conv_exact import conv_exact
dev = torch.device("cuda", 0)
for (n, cin, cout, h, w) in [(1, 32, 64, 16, 16), (1, 64, 64, 16, 16), (1, 64, 64, 16, 64), (2, 64, 64, 16, 64), (1, 64, 64, 32, 32)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    got = T._conv_layer(x.to(dev), wt, b, bn, 1, 0, False, 3).cpu().numpy()
    cfg = os.environ["DCX_FORCE_CFG"]
    ref = conv_exact(x.numpy(), wt.numpy(), b.numpy(), [t.numpy() for t in bn], pad=1,
                     family=T._family(cfg))
    bad = got.view(np.uint32) != ref.view(np.uint32)
    print((n, cin, cout, h, w), "bad", int(bad.sum()), "of", bad.size, "maxabs", float(np.abs(got - ref).max()))
    if bad.any():
        idx = np.argwhere(bad)
        print("  n:", np.unique(idx[:, 0]), " cout:", np.unique(idx[:, 1])[:20], " y:", np.unique(idx[:, 2]), " x:", np.unique(idx[:, 3]))
