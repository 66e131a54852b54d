/* ORACLE -- test infrastructure, NOT product code (see oracle/deepcharuco_oracle.py header).
 *
 * Plain-C restatement of one convolution layer of the reference
 *   (Conv2d + BatchNorm2d(eval) + ReLU [+ MaxPool2d(2,2)], /root/reference/src/models/net.py:60-77,
 *    /root/reference/src/models/refinenet.py:56-81)
 * evaluated in the EXACT fp32 summation order the gfx950 kernels document
 * (deepcharuco_amd/csrc/dcx_conv_mfma.h header, DESIGN.md "Numerics"), so that kernel outputs can
 * be compared bit for bit:
 *   acc = 0
 *   for each 16-channel chunk c0, each tap (dy-major), s = 0..1, j = 0..3:
 *       acc = fmaf(w[c0+8s+j],   x[c0+8s+j],   acc)
 *       acc = fmaf(w[c0+8s+4+j], x[c0+8s+4+j], acc)
 *   y = max(fmaf(acc, alpha, beta2), 0)   with alpha = gamma * (1/sqrt(var+eps)), beta = bn_beta - mean*alpha,
 *                                          beta2 = fmaf(bias, alpha, beta)  (conv bias folded into the BN shift; fp32)
 *   (raw 1x1 heads: y = acc + bias;  Cin = 1 first layers: y = max(fmaf(acc + bias, alpha, beta), 0))
 * For cin == 1 (first layers) the order is simply tap 0..8.
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC   (see oracle/Makefile)
 */
#include <math.h>
#include <stddef.h>

void dcx_oracle_fold_bn(const float* gamma, const float* bbeta, const float* mean, const float* var, int c,
                        float* alpha, float* beta) {
    for (int i = 0; i < c; ++i) {
        const float inv = 1.0f / sqrtf(var[i] + 1e-5f);
        const float a = gamma[i] * inv;
        alpha[i] = a;
        beta[i] = bbeta[i] - mean[i] * a;
    }
}

/* x: NCHW [n][cin][h][w] (the LOGICAL input: caller applies the nearest x2 up-sampling first)
 * wt: OIHW, y: NCHW [n][cout][ho][wo] with ho = h + 2*pad - (ks-1).  alpha == NULL -> raw conv + bias. */
void dcx_oracle_conv_exact(const float* x, int n, int cin, int h, int w, const float* wt, const float* bias,
                           const float* alpha, const float* beta, int cout, int ks, int pad, float* y) {
    const int ho = h + 2 * pad - (ks - 1), wo = w + 2 * pad - (ks - 1);
    const int taps = ks * ks;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int co = 0; co < cout; ++co)
            for (int oy = 0; oy < ho; ++oy)
                for (int ox = 0; ox < wo; ++ox) {
                    float acc = 0.0f;
                    if (cin == 1) {
                        for (int t = 0; t < taps; ++t) {
                            const int iy = oy - pad + t / ks, ix = ox - pad + t % ks;
                            const int inb = iy >= 0 && iy < h && ix >= 0 && ix < w;
                            const float xv = inb ? x[((size_t)b * h + iy) * w + ix] : 0.0f;
                            acc = fmaf(wt[(size_t)co * taps + t], xv, acc);
                        }
                    } else {
                        for (int c0 = 0; c0 < cin; c0 += 16)
                            for (int t = 0; t < taps; ++t) {
                                const int iy = oy - pad + t / ks, ix = ox - pad + t % ks;
                                const int inb = iy >= 0 && iy < h && ix >= 0 && ix < w;
                                for (int s = 0; s < 2; ++s)
                                    for (int j = 0; j < 4; ++j)
                                        for (int k = 0; k < 2; ++k) {
                                            const int ci = c0 + 8 * s + 4 * k + j;
                                            const float xv = inb ? x[(((size_t)b * cin + ci) * h + iy) * w + ix] : 0.0f;
                                            acc = fmaf(wt[((size_t)co * cin + ci) * taps + t], xv, acc);
                                        }
                            }
                    }
                    float v;
                    if (alpha == NULL) v = acc + bias[co];                       /* raw 1x1 heads */
                    else if (cin == 1) v = fmaxf(fmaf(acc + bias[co], alpha[co], beta[co]), 0.0f);   /* direct first layers */
                    else v = fmaxf(fmaf(acc, alpha[co], fmaf(bias[co], alpha[co], beta[co])), 0.0f);   /* MFMA layers: bias folded */
                    y[(((size_t)b * cout + co) * ho + oy) * wo + ox] = v;
                }
}
