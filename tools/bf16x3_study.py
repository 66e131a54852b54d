"""Exploratory numerics study (CPU only, never on the product path): would a conv layer computed on the
bf16 matrix pipe with every fp32 operand split into THREE bf16 terms (a = a1 + a2 + a3 exactly: 3 x 8 mantissa bits; six
cross products a1b1, a1b2, a2b1, a2b2, a1b3, a3b1 accumulated in fp32) be as accurate as the fp32 kernels?

One conv1b-shaped layer (64 -> 64, 3x3, pad 1, K = 576) on post-ReLU-like activations and He-normal weights; error of every
variant against an fp64 evaluation of the same fp32 inputs:
  fp32 direct      the direct family's fmaf chain                      (oracle/conv_exact.c, bit-exact model of the kernel)
  fp32 winograd    the 2-D Winograd F(2x2,3x3) family                  (oracle/conv_exact.c)
  bf16x3 (6 terms) products exact in fp32 (8 x 8 mantissa bits), fp32 accumulation; two accumulation models because the
                   internal order of v_mfma_f32_32x32x16_bf16 is not documented: (a) one rounding per product (sequential
                   chain), (b) one rounding per 16-product block (what a fused block adder would do)
  bf16x3 (8 terms) + a2b3, a3b2 (relative 2^-24 each)
usage: python tools/bf16x3_study.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.conv_exact import conv_exact  # noqa: E402


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x1 = bf16(x)
    r = (x - x1).astype(np.float32)
    x2 = bf16(r)
    x3 = bf16((r - x2).astype(np.float32))
    assert np.array_equal((x1.astype(np.float64) + x2 + x3).astype(np.float32), x)
    return x1, x2, x3


def main():
    rng = np.random.default_rng(7)
    n, cin, cout, h, w = 1, 64, 64, 24, 32
    x = np.maximum(rng.standard_normal((n, cin, h, w)).astype(np.float32) * 0.8 + 0.2, 0).astype(np.float32)   # post-ReLU-like
    wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    b = np.zeros(cout, np.float32)
    bn = (np.ones(cout, np.float32), np.zeros(cout, np.float32), np.zeros(cout, np.float32), np.full(cout, 1.0 - 1e-5, np.float32))
    # im2col (pad 1): A [pixels][K], B [K][cout], K ordered chunk(16 ch) / tap / channel like the direct kernel
    xp = np.pad(x[0], ((0, 0), (1, 1), (1, 1)))
    cols = np.stack([xp[:, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], 0)       # [tap][cin][h][w]
    A = cols.transpose(2, 3, 1, 0).reshape(h * w, cin, 9)                                          # [px][cin][tap]
    A = A.reshape(h * w, cin // 16, 16, 9).transpose(0, 1, 3, 2).reshape(h * w, cin * 9)          # chunk / tap / channel
    B = wt.reshape(cout, cin // 16, 16, 9).transpose(1, 3, 2, 0).reshape(cin * 9, cout)
    ref = A.astype(np.float64) @ B.astype(np.float64)                                              # pre-activation, fp64
    pre = lambda y: y[0].transpose(1, 2, 0).reshape(h * w, cout).astype(np.float64)
    # ReLU hides negative outputs: compare on the positive ones only for the C restatements
    pos = ref > 0.05
    out = {}
    for fam in ("direct", "w2h"):
        y = conv_exact(x, wt, b, bn, pad=1, family=fam)
        out["fp32 " + ("direct (kernel order)" if fam == "direct" else "winograd F(2x2,3x3) (kernel order)")] = np.abs(pre(y) - ref)[pos]
    a1, a2, a3 = split3(A)
    b1, b2, b3 = split3(B)
    for label, terms in (("bf16x3, 6 terms", [(a2, b2), (a1, b3), (a3, b1), (a1, b2), (a2, b1), (a1, b1)]),
                         ("bf16x3, 8 terms", [(a2, b3), (a3, b2), (a2, b2), (a1, b3), (a3, b1), (a1, b2), (a2, b1), (a1, b1)])):
        K = A.shape[1]
        acc_seq = np.zeros((h * w, cout), np.float32)
        acc_blk = np.zeros((h * w, cout), np.float32)
        for k0 in range(0, K, 16):
            for ta, tb in terms:
                prod = ta[:, k0:k0 + 16, None].astype(np.float64) * tb[None, k0:k0 + 16, :].astype(np.float64)   # exact in fp32
                for k in range(16):
                    acc_seq = (acc_seq.astype(np.float64) + prod[:, k]).astype(np.float32)            # one rounding per product
                acc_blk = (acc_blk.astype(np.float64) + prod.sum(1)).astype(np.float32)               # one rounding per block
        out[label + ", one rounding per product"] = np.abs(acc_seq - ref)[pos]
        out[label + ", one rounding per 16-product block"] = np.abs(acc_blk - ref)[pos]
    print(f"conv1b-shaped layer, K = {cin * 9}, {int(pos.sum())} outputs > 0.05 (max |out| {np.abs(ref).max():.2f}); error vs fp64")
    for k, e in out.items():
        print(f"  {k:62s} max {e.max():.3e}   mean {e.mean():.3e}")
    print("matrix-pipe cost per 16 x (32x32) MACs: fp32 direct 8 x 64 = 512 cycles; fp32 Winograd 4/9 of that = 228;\n"
          "  bf16x3 direct 6 x 32 = 192 (8 terms: 256); bf16x3 inside the Winograd GEMMs would be 4/9 of that = 85 (114)")


if __name__ == "__main__":
    main()
