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

/* Same layer through the 1-D Winograd F(2,3) kernel (deepcharuco_amd/csrc/dcx_conv_wino.h), in ITS exact fp32 order:
 *   per output pair (x0 = 2i, x0+1), kernel row ky, channel:  d_e = in[oy-pad+ky][x0-pad+e], e = 0..3 (0 outside)
 *     v0 = d0-d2  v1 = d1+d2  v2 = d2-d1  v3 = d1-d3;   u0 = g0  u1 = ((g0+g1)+g2)*0.5f  u2 = ((g0-g1)+g2)*0.5f  u3 = g2
 *   m_p = 0;  for chunk c0 / ky / s / j / k:  m_p = fmaf(u_p[ci], v_p[ci], m_p),  ci = c0 + 8s + 4k + j
 *   out[x0] = (m0+m1)+m2,  out[x0+1] = (m1-m2)-m3,  y = max(fmaf(out, alpha, fmaf(bias, alpha, beta)), 0)
 * 3x3 + BN + ReLU layers only (cin % 16 == 0). */
void dcx_oracle_conv_wino_exact(const float* x, int n, int cin, int h, int w, const float* wt, const float* bias,
                                const float* alpha, const float* beta, int cout, int pad, float* y) {
    const int ho = h + 2 * pad - 2, wo = w + 2 * pad - 2;
    const int npair = (wo + 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int co = 0; co < cout; ++co)
            for (int oy = 0; oy < ho; ++oy)
                for (int pr = 0; pr < npair; ++pr) {
                    const int x0 = 2 * pr;
                    float m[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    for (int c0 = 0; c0 < cin; c0 += 16)
                        for (int ky = 0; ky < 3; ++ky) {
                            const int iy = oy - pad + ky;
                            for (int s = 0; s < 2; ++s)
                                for (int j = 0; j < 4; ++j)
                                    for (int k = 0; k < 2; ++k) {
                                        const int ci = c0 + 8 * s + 4 * k + j;
                                        float d[4];
                                        for (int e = 0; e < 4; ++e) {
                                            const int ix = x0 - pad + e;
                                            const int inb = iy >= 0 && iy < h && ix >= 0 && ix < w;
                                            d[e] = inb ? x[(((size_t)b * cin + ci) * h + iy) * w + ix] : 0.0f;
                                        }
                                        const float* g = wt + ((size_t)co * cin + ci) * 9 + ky * 3;
                                        const float u[4] = {g[0], ((g[0] + g[1]) + g[2]) * 0.5f, ((g[0] - g[1]) + g[2]) * 0.5f, g[2]};
                                        const float v[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
                                        for (int p = 0; p < 4; ++p) m[p] = fmaf(u[p], v[p], m[p]);
                                    }
                        }
                    const float o0 = (m[0] + m[1]) + m[2], o1 = (m[1] - m[2]) - m[3];
                    const float b2 = fmaf(bias[co], alpha[co], beta[co]);
                    float* row = y + (((size_t)b * cout + co) * ho + oy) * wo;
                    row[x0] = fmaxf(fmaf(o0, alpha[co], b2), 0.0f);
                    if (x0 + 1 < wo) row[x0 + 1] = fmaxf(fmaf(o1, alpha[co], b2), 0.0f);
                }
}
