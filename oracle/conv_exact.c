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
#include <stdlib.h>

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

/* Same layer through the 2-D Winograd F(2x2,3x3) kernel family (deepcharuco_amd/csrc/dcx_conv_wino2h.h), in ITS exact fp32 order:
 *   per 2x2 output tile (rows 2ty, 2ty+1; columns 2tx, 2tx+1), channel: d[r][c] = in[2ty-pad+r][2tx-pad+c], r, c = 0..3
 *     rows     t0 = d[0]-d[2]  t1 = d[1]+d[2]  t2 = d[2]-d[1]  t3 = d[1]-d[3]          (per column c)
 *     columns  v[xi][0] = t[xi][0]-t[xi][2]  v1 = t1+t2  v2 = t2-t1  v3 = t1-t3
 *     weights  h[xi][kx] over ky: h0 = g0, h1 = ((g0+g1)+g2)*0.5f, h2 = ((g0-g1)+g2)*0.5f, h3 = g2; then the same over kx
 *   m[xi][nu] = 0;  for chunk c0 (16 cin) / j in 0..3 / g in 0..3:  m = fmaf(u[ci], v[ci], m),  ci = c0 + 4g + j
 *             (v_mfma_f32_16x16x4_f32: MFMA j of a position consumes component j of the lanes' channel quads g = 0..3)
 *   y[i][j] = 0; for p = 4 xi + nu ascending: y[i][j] = fmaf(AT[i][xi]*AT[j][nu], m[xi][nu], y[i][j]), AT = [[1,1,1,0],[0,1,-1,-1]]
 *             (the kernel runs this chain on the matrix cores, v_mfma_f32_4x4x1: all 16 terms, zero coefficients included)
 *   out = max(fmaf(y, alpha, fmaf(bias, alpha, beta)), 0) */
void dcx_oracle_conv_wino2h_exact(const float* x, int n, int cin, int h, int w, const float* wt, const float* bias,
                                  const float* alpha, const float* beta, int cout, int pad, float* y) {
    const int ho = h + 2 * pad - 2, wo = w + 2 * pad - 2;
    const int nty = (ho + 1) / 2, ntx = (wo + 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int co = 0; co < cout; ++co)
            for (int ty = 0; ty < nty; ++ty)
                for (int tx = 0; tx < ntx; ++tx) {
                    float m[4][4] = {{0.0f}};
                    for (int c0 = 0; c0 < cin; c0 += 16)
                        for (int step = 0; step < 16; ++step) {
                                    const int ci = c0 + 4 * (step & 3) + (step >> 2);      /* step = 4j + g */
                                    float d[4][4], t[4][4], v[4][4], hh[4][3], u[4][4];
                                    for (int r = 0; r < 4; ++r)
                                        for (int c = 0; c < 4; ++c) {
                                            const int iy = 2 * ty - pad + r, ix = 2 * tx - pad + c;
                                            const int inb = iy >= 0 && iy < h && ix >= 0 && ix < w;
                                            d[r][c] = inb ? x[(((size_t)b * cin + ci) * h + iy) * w + ix] : 0.0f;
                                        }
                                    for (int c = 0; c < 4; ++c) {
                                        t[0][c] = d[0][c] - d[2][c]; t[1][c] = d[1][c] + d[2][c];
                                        t[2][c] = d[2][c] - d[1][c]; t[3][c] = d[1][c] - d[3][c];
                                    }
                                    for (int xi = 0; xi < 4; ++xi) {
                                        v[xi][0] = t[xi][0] - t[xi][2]; v[xi][1] = t[xi][1] + t[xi][2];
                                        v[xi][2] = t[xi][2] - t[xi][1]; v[xi][3] = t[xi][1] - t[xi][3];
                                    }
                                    const float* g = wt + ((size_t)co * cin + ci) * 9;
                                    for (int kx = 0; kx < 3; ++kx) {
                                        const float g0 = g[kx], g1 = g[3 + kx], g2 = g[6 + kx];
                                        hh[0][kx] = g0; hh[1][kx] = ((g0 + g1) + g2) * 0.5f;
                                        hh[2][kx] = ((g0 - g1) + g2) * 0.5f; hh[3][kx] = g2;
                                    }
                                    for (int xi = 0; xi < 4; ++xi) {
                                        u[xi][0] = hh[xi][0]; u[xi][1] = ((hh[xi][0] + hh[xi][1]) + hh[xi][2]) * 0.5f;
                                        u[xi][2] = ((hh[xi][0] - hh[xi][1]) + hh[xi][2]) * 0.5f; u[xi][3] = hh[xi][2];
                                    }
                                    for (int xi = 0; xi < 4; ++xi)
                                        for (int nu = 0; nu < 4; ++nu) m[xi][nu] = fmaf(u[xi][nu], v[xi][nu], m[xi][nu]);
                                }
                    /* output transform: a sequential fmaf chain over all 16 positions (xi-major, ascending) with the coefficient
                     * AT[i][xi]*AT[j][nu] in {0, +1, -1}, AT = [[1,1,1,0],[0,1,-1,-1]], starting from +0 */
                    static const int AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
                    float o[2][2];
                    for (int i = 0; i < 2; ++i)
                        for (int jj = 0; jj < 2; ++jj) {
                            float acc2 = 0.0f;
                            for (int xi = 0; xi < 4; ++xi)
                                for (int nu = 0; nu < 4; ++nu)
                                    acc2 = fmaf((float)(AT[i][xi] * AT[jj][nu]), m[xi][nu], acc2);
                            o[i][jj] = acc2;
                        }
                    const float b2 = fmaf(bias[co], alpha[co], beta[co]);
                    for (int i = 0; i < 2; ++i)
                        for (int jj = 0; jj < 2; ++jj) {
                            const int oy = 2 * ty + i, ox = 2 * tx + jj;
                            if (oy < ho && ox < wo)
                                y[(((size_t)b * cout + co) * ho + oy) * wo + ox] = fmaxf(fmaf(o[i][jj], alpha[co], b2), 0.0f);
                        }
                }
}

/* 3x3 convolution (pad 1) + BN + ReLU over a nearest-x2 UP-SAMPLED input as four phase convolutions, each a 2-D Winograd
 * F(2x2,2x2) (deepcharuco_amd/csrc/dcx_conv_wino2p.h), computed on the low-resolution tensor x [n][cin][h][w]; y is
 * [n][cout][2h][2w].  Up-sampling repeats every pixel 2x2, so the 3x3 window of output (2Y+a, 2X+b) sees a 2x2 block of
 * distinct low-resolution pixels and each phase (a, b) is a 2x2-tap convolution with weights
 *   Wp[a][b][dy][dx] = the 3x3 kernel summed over the rows / columns that fall on the same low-resolution pixel: rows
 *   a = 0: dy 0 <- {ky 0}, dy 1 <- {ky 1, 2};  a = 1: dy 0 <- {ky 0, 1}, dy 1 <- {ky 2};  columns alike with b;  fp32, rows
 *   first, then columns, left to right (dcx_api.hip: pack_conv_ups2);
 *   Wc[dy] = (Wp[dy][0], Wp[dy][0] + Wp[dy][1], Wp[dy][1]);  U[.][nu] = (Wc[0][nu], Wc[0][nu] + Wc[1][nu], Wc[1][nu]);
 *   per phase (a, b) and 2x2 tile of low-resolution positions (y0, x0), y0, x0 even:
 *     d[r][s] = x[y0 - (1-a) + r][x0 - (1-b) + s] (0 outside);  t[0] = d[0] - d[1], t[1] = d[1], t[2] = d[2] - d[1] (per column s);
 *     v[xi][0] = t[xi][0] - t[xi][1], v[xi][1] = t[xi][1], v[xi][2] = t[xi][2] - t[xi][1];
 *     m[p = 3 xi + nu] = 0; for chunk c0 (16 cin) / j in 0..3 / g in 0..3: m = fmaf(U[ci], v[ci], m), ci = c0 + 4 g + j;
 *     y_k (k = 2 i + j) = 0; for p in 0..8: y_k = fmaf(AT[i][xi] * AT[j][nu], m[p], y_k), AT = [[1,1,0],[0,1,1]];
 *     out[2 (y0 + i) + a][2 (x0 + j) + b] = max(fmaf(y_k, alpha, fmaf(bias, alpha, beta)), 0). */
void dcx_oracle_conv_ups2w_exact(const float* x, int n, int cin, int h, int w, const float* wt, const float* bias,
                                 const float* alpha, const float* beta, int cout, float* y) {
    static const int lo[2][2] = {{0, 1}, {0, 2}}, hi[2][2] = {{0, 2}, {1, 2}};
    static const float AT[2][3] = {{1.f, 1.f, 0.f}, {0.f, 1.f, 1.f}};
    const int ho = 2 * h, wo = 2 * w;
    const int ty_n = (h + 1) / 2, tx_n = (w + 1) / 2;
    /* transformed weights U[phase][p][co][ci] */
    float* U = (float*)malloc(sizeof(float) * 36 * (size_t)cout * cin);
    for (int ph = 0; ph < 4; ++ph)
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci) {
                const int pa = ph >> 1, pb = ph & 1;
                const float* g = wt + ((size_t)co * cin + ci) * 9;
                float wp[2][2], wc[2][3];
                for (int dy = 0; dy < 2; ++dy)
                    for (int dx = 0; dx < 2; ++dx) {
                        float r[3];
                        for (int kx = 0; kx < 3; ++kx) {
                            r[kx] = g[lo[pa][dy] * 3 + kx];
                            for (int ky = lo[pa][dy] + 1; ky <= hi[pa][dy]; ++ky) r[kx] = r[kx] + g[ky * 3 + kx];
                        }
                        float wv = r[lo[pb][dx]];
                        for (int kx = lo[pb][dx] + 1; kx <= hi[pb][dx]; ++kx) wv = wv + r[kx];
                        wp[dy][dx] = wv;
                    }
                for (int dy = 0; dy < 2; ++dy) { wc[dy][0] = wp[dy][0]; wc[dy][1] = wp[dy][0] + wp[dy][1]; wc[dy][2] = wp[dy][1]; }
                for (int nu = 0; nu < 3; ++nu) {
                    U[((size_t)(ph * 9 + 0 + nu) * cout + co) * cin + ci] = wc[0][nu];
                    U[((size_t)(ph * 9 + 3 + nu) * cout + co) * cin + ci] = wc[0][nu] + wc[1][nu];
                    U[((size_t)(ph * 9 + 6 + nu) * cout + co) * cin + ci] = wc[1][nu];
                }
            }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int ty = 0; ty < ty_n; ++ty) {
            float* V = (float*)malloc(sizeof(float) * 9 * (size_t)cin);
            for (int tx = 0; tx < tx_n; ++tx)
                for (int ph = 0; ph < 4; ++ph) {
                    const int pa = ph >> 1, pb = ph & 1;
                    const int y0 = 2 * ty, x0 = 2 * tx;
                    for (int ci = 0; ci < cin; ++ci) {
                        float d[3][3], t[3][3];
                        for (int r = 0; r < 3; ++r)
                            for (int s = 0; s < 3; ++s) {
                                const int iy = y0 - (1 - pa) + r, ix = x0 - (1 - pb) + s;
                                d[r][s] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? x[(((size_t)b * cin + ci) * h + iy) * w + ix] : 0.0f;
                            }
                        for (int s = 0; s < 3; ++s) { t[0][s] = d[0][s] - d[1][s]; t[1][s] = d[1][s]; t[2][s] = d[2][s] - d[1][s]; }
                        for (int xi = 0; xi < 3; ++xi) {
                            V[(size_t)(xi * 3 + 0) * cin + ci] = t[xi][0] - t[xi][1];
                            V[(size_t)(xi * 3 + 1) * cin + ci] = t[xi][1];
                            V[(size_t)(xi * 3 + 2) * cin + ci] = t[xi][2] - t[xi][1];
                        }
                    }
                    for (int co = 0; co < cout; ++co) {
                        float m[9];
                        for (int p = 0; p < 9; ++p) {
                            const float* u = U + ((size_t)(ph * 9 + p) * cout + co) * cin;
                            const float* v = V + (size_t)p * cin;
                            float acc = 0.0f;
                            for (int c0 = 0; c0 < cin; c0 += 16)
                                for (int j = 0; j < 4; ++j)
                                    for (int g = 0; g < 4; ++g) acc = fmaf(u[c0 + 4 * g + j], v[c0 + 4 * g + j], acc);
                            m[p] = acc;
                        }
                        for (int i = 0; i < 2; ++i)
                            for (int j = 0; j < 2; ++j) {
                                float yk = 0.0f;
                                for (int p = 0; p < 9; ++p) yk = fmaf(AT[i][p / 3] * AT[j][p % 3], m[p], yk);
                                const int ly = y0 + i, lx = x0 + j;
                                if (ly < h && lx < w)
                                    y[(((size_t)b * cout + co) * ho + 2 * ly + pa) * wo + 2 * lx + pb] =
                                        fmaxf(fmaf(yk, alpha[co], fmaf(bias[co], alpha[co], beta[co])), 0.0f);
                            }
                    }
                }
            free(V);
        }
    free(U);
}
