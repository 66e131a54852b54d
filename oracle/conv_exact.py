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


def conv_exact(x, wt, bias, bn=None, pad=1, ups=False, pool=False, wino=False):
    """x NCHW float32; bn = (gamma, beta, mean, var) or None (raw).  Returns NCHW float32.
    wino=True: the summation order of the 1-D Winograd F(2,3) kernel, wino=2: of the 2-D F(2x2,3x3) kernel
    (3x3 + BN layers only), wino=4: of its half-tile variant (dcx_conv_wino2h.h); wino=3: the phase variant of the direct kernel (3x3 pad 1 + BN over an up-sampled input), wino=5: the phase variant with
    F(2x2,2x2) per phase (dcx_conv_wino2p.h)."""
    x = np.ascontiguousarray(x, np.float32)
    if wino in (3, 5):
        assert ups and bn is not None and pad == 1 and not pool
        wt = np.ascontiguousarray(wt, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        n, cin, h, w = x.shape
        cout = wt.shape[0]
        assert wt.shape[2] == 3 and cin % 16 == 0
        y = np.empty((n, cout, 2 * h, 2 * w), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        g, be, mu, var = [np.ascontiguousarray(t, np.float32) for t in bn]
        alpha, beta = np.empty(cout, np.float32), np.empty(cout, np.float32)
        lib().dcx_oracle_fold_bn(p(g), p(be), p(mu), p(var), cout, p(alpha), p(beta))
        fn = lib().dcx_oracle_conv_ups2_exact if wino == 3 else lib().dcx_oracle_conv_ups2w_exact
        fn(p(x), n, cin, h, w, p(wt), p(bias), p(alpha), p(beta), cout, p(y))
        return y
    if ups:
        x = np.ascontiguousarray(x.repeat(2, axis=2).repeat(2, axis=3))
    wt = np.ascontiguousarray(wt, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    n, cin, h, w = x.shape
    cout, _, ks, _ = wt.shape
    ho, wo = h + 2 * pad - (ks - 1), w + 2 * pad - (ks - 1)
    y = np.empty((n, cout, ho, wo), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    alpha = beta = None
    if bn is not None:
        g, be, mu, var = [np.ascontiguousarray(t, np.float32) for t in bn]
        alpha, beta = np.empty(cout, np.float32), np.empty(cout, np.float32)
        lib().dcx_oracle_fold_bn(p(g), p(be), p(mu), p(var), cout, p(alpha), p(beta))
    if wino:     # True / 1: 1-D F(2,3) along x;  2: 2-D F(2x2,3x3)
        assert bn is not None and ks == 3 and cin % 16 == 0
        fn = (lib().dcx_oracle_conv_wino2_exact if wino == 2 else lib().dcx_oracle_conv_wino2h_exact if wino == 4
              else lib().dcx_oracle_conv_wino_exact)
        fn(p(x), n, cin, h, w, p(wt), p(bias), p(alpha), p(beta), cout, pad, p(y))
    else:
        lib().dcx_oracle_conv_exact(p(x), n, cin, h, w, p(wt), p(bias), p(alpha) if bn is not None else None,
                                    p(beta) if bn is not None else None, cout, ks, pad, p(y))
    if pool:
        y = y[:, :, :ho // 2 * 2, :wo // 2 * 2].reshape(n, cout, ho // 2, 2, wo // 2, 2).max(axis=(3, 5))
    return y
