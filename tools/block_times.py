#!/usr/bin/env python3
"""Tuning aid: start / end time of EVERY workgroup of one half-tile Winograd launch (needs a library built with
    make -C deepcharuco_amd/csrc -B EXTRA=-DDCX_W2H_BLOCKTIMES
-- the stamps overrun the launch's probe slot, so only one profiled launch is made).  Prints the distribution per XCD.
    python tools/block_times.py [n cin cout h w pool]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_parity as T
from deepcharuco_amd import _lib
n, cin, cout, h, w, pool = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else "32 64 64 240 320 1".split())]
dev = torch.device("cuda", 0)
L = _lib.lib()
g = torch.Generator().manual_seed(1)
x = torch.randn(n, cin, h, w, generator=g).to(dev)
wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
b = torch.randn(cout, generator=g) * 0.1
bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
if os.environ.get("BT_WEIGHTS"):      # per-XCD item shares (dcx_set_xcd_weights): e.g. BT_WEIGHTS=1.0125,0.9875,1.0125,0.9875,1.0125,0.9875,1.0125,0.9875
    from deepcharuco_amd.inference import set_xcd_weights, get_xcd_weights
    set_xcd_weights([float(v) for v in os.environ["BT_WEIGHTS"].split(",")], dev)
    print("XCD weights:", " ".join(f"{v:.4f}" for v in get_xcd_weights(dev)))
for _ in range(2):
    T._conv_layer(x, wt, b, bn, 1, 0, bool(pool), 3)
L.dcx_profile_filter(-1); L.dcx_profile_enable(1)
T._conv_layer(x, wt, b, bn, 1, 0, bool(pool), 3)
L.dcx_profile_enable(0)
cnt = L.dcx_profile_count()
ids = (C.c_int * cnt)(); nimg = (C.c_int * cnt)(); lim = (C.c_int * cnt)(); fl = (C.c_double * cnt)(); ms = (C.c_float * cnt)()
L.dcx_profile_fetch(ids, nimg, lim, fl, ms, cnt)
print(L.dcx_profile_kernel_name(ids[0]).decode(), f"{ms[0] * 1e3:.1f} us")
words = []
for s in range(0, 18):
    wbuf = (C.c_ulonglong * 64)()
    L.dcx_profile_probe_words(s, wbuf)
    words += list(wbuf)
t = np.array(words[64:64 + 2 * 512], dtype=np.int64).reshape(512, 2)
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01     # us (100 MHz counter)
print(f"start: min {st.min():.1f} max {st.max():.1f} us;  end: min {en.min():.1f} median {np.median(en):.1f} max {en.max():.1f} us")
for xcd in range(8):
    e = en[xcd::8]
    print(f"  XCD {xcd}: start max {st[xcd::8].max():6.1f}  end min {e.min():7.1f} median {np.median(e):7.1f} max {e.max():7.1f}")
late = np.argsort(en)[-12:]
print("latest blocks:", [(int(b_), round(float(st[b_]), 1), round(float(en[b_]), 1)) for b_ in late])
if os.environ.get("BT_DETAIL"):
    for xcd in (0, 5):
        print(f"XCD {xcd}: end time by j = block >> 3")
        e = en[xcd::8]
        for j0 in range(0, 64, 8):
            print("   ", " ".join(f"{v:6.1f}" for v in e[j0:j0 + 8]))
