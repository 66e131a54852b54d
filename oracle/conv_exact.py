"""ORACLE -- ctypes wrapper of oracle/conv_exact.c (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libconv_exact.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
    return _lib


FAMILIES = ("direct", "w2h", "w2p")


def conv_exact(x, wt, bias, bn=None, pad=1, ups=False, pool=False, family="direct"):
    """x NCHW float32; bn = (gamma, beta, mean, var) or None (raw).  Returns NCHW float32 in the exact fp32 summation order of
    a kernel family of deepcharuco_amd/csrc (dcx_conv_mfma.hip: family_of):
      "direct"  dcx_conv_mfma.h   every multiply-add of the layer as written (any layer; 1x1 heads; deterministic mode);
      "w2h"     dcx_conv_wino2h.h 2-D Winograd F(2x2,3x3) (3x3 + BN + ReLU layers, cin % 16 == 0);
      "w2p"     dcx_conv_wino2p.h 3x3 pad 1 + BN + ReLU over a x2 up-sampled input: four phases x F(2x2,2x2)."""
    assert family in FAMILIES, family
    x = np.ascontiguousarray(x, np.float32)
    wt = np.ascontiguousarray(wt, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    cout, _, ks, _ = wt.shape
    alpha = beta = None
    if bn is not None:
        g, be, mu, var = [np.ascontiguousarray(t, np.float32) for t in bn]
        alpha, beta = np.empty(cout, np.float32), np.empty(cout, np.float32)
        lib().dcx_oracle_fold_bn(p(g), p(be), p(mu), p(var), cout, p(alpha), p(beta))
    if family == "w2p":
        assert ups and bn is not None and pad == 1 and not pool and ks == 3 and x.shape[1] % 16 == 0
        n, cin, h, w = x.shape
        y = np.empty((n, cout, 2 * h, 2 * w), np.float32)
        lib().dcx_oracle_conv_ups2w_exact(p(x), n, cin, h, w, p(wt), p(bias), p(alpha), p(beta), cout, p(y))
        return y
    if ups:
        x = np.ascontiguousarray(x.repeat(2, axis=2).repeat(2, axis=3))
    n, cin, h, w = x.shape
    ho, wo = h + 2 * pad - (ks - 1), w + 2 * pad - (ks - 1)
    y = np.empty((n, cout, ho, wo), np.float32)
    if family == "w2h":
        assert bn is not None and ks == 3 and cin % 16 == 0
        lib().dcx_oracle_conv_wino2h_exact(p(x), n, cin, h, w, p(wt), p(bias), p(alpha), p(beta), cout, pad, p(y))
    else:
        lib().dcx_oracle_conv_exact(p(x), n, cin, h, w, p(wt), p(bias), p(alpha) if bn is not None else None,
                                    p(beta) if bn is not None else None, cout, ks, pad, p(y))
    if pool:
        y = y[:, :, :ho // 2 * 2, :wo // 2 * 2].reshape(n, cout, ho // 2, 2, wo // 2, 2).max(axis=(3, 5))
    return y
