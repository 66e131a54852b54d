// dcx_api.hip -- C ABI of libdeepcharuco_amd.so: model handles, weight packing, the layer
// chains of both networks and the sync-free batched pipeline.  See include/deepcharuco_amd.h.
#include "dcx_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

namespace {

constexpr float kBnEps = 1e-5f;   // torch.nn.BatchNorm2d default (never overridden by the reference)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- host-side packing ---------------------------------------------------------------------

// OIHW -> [tap][cin/4][cout_pad][4 = cin % 4], zero padded in cout.
std::vector<float> pack_conv(const float* w, int cout, int cin, int ks, int cout_pad) {
    const int taps = ks * ks, cq = cin / 4;
    std::vector<float> out((size_t)taps * cq * cout_pad * 4, 0.0f);
    for (int o = 0; o < cout; ++o)
        for (int i = 0; i < cin; ++i)
            for (int t = 0; t < taps; ++t)
                out[(((size_t)t * cq + (i >> 2)) * cout_pad + o) * 4 + (i & 3)] = w[((size_t)o * cin + i) * taps + t];
    return out;
}

// 3x3 OIHW -> 2-D Winograd F(2x2,3x3): [xi*4 + nu][cin/4][cout_pad][4], U = G g G^T in fp32, rows (ky) first, then columns
// (dcx_conv_wino2h.h; restated by oracle/conv_exact.c).
std::vector<float> pack_conv_wino2(const float* w, int cout, int cin, int cout_pad) {
    const int cq = cin / 4;
    std::vector<float> out((size_t)16 * cq * cout_pad * 4, 0.0f);
    for (int o = 0; o < cout; ++o)
        for (int i = 0; i < cin; ++i) {
            const float* g = w + ((size_t)o * cin + i) * 9;
            float h[4][3], u[4][4];
            for (int kx = 0; kx < 3; ++kx) {
                const float g0 = g[kx], g1 = g[3 + kx], g2 = g[6 + kx];
                h[0][kx] = g0; h[1][kx] = ((g0 + g1) + g2) * 0.5f; h[2][kx] = ((g0 - g1) + g2) * 0.5f; h[3][kx] = g2;
            }
            for (int xi = 0; xi < 4; ++xi) {
                u[xi][0] = h[xi][0]; u[xi][1] = ((h[xi][0] + h[xi][1]) + h[xi][2]) * 0.5f;
                u[xi][2] = ((h[xi][0] - h[xi][1]) + h[xi][2]) * 0.5f; u[xi][3] = h[xi][2];
            }
            for (int pos = 0; pos < 16; ++pos)
                out[(((size_t)pos * cq + (i >> 2)) * cout_pad + o) * 4 + (i & 3)] = u[pos >> 2][pos & 3];
        }
    return out;
}

// 3x3 OIHW -> the four phase kernels of a 3x3 convolution over a nearest-x2 up-sampled input (host-side intermediate of
// pack_conv_ups2w; dcx_conv_wino2p.h explains the phases):
// [phase = 2a + b][tap = 2 dy + dx][cin/4][cout_pad][4].  Row sets: a = 0: dy 0 <- {ky 0}, dy 1 <- {ky 1, 2}; a = 1: dy 0 <-
// {ky 0, 1}, dy 1 <- {ky 2}; columns alike with b.  fp32, rows first then columns, left to right (restated by
// oracle/conv_exact.c: dcx_oracle_conv_ups2_exact).
std::vector<float> pack_conv_ups2(const float* w, int cout, int cin, int cout_pad) {
    const int cq = cin / 4;
    std::vector<float> out((size_t)16 * cq * cout_pad * 4, 0.0f);
    static const int lo[2][2] = {{0, 1}, {0, 2}}, hi[2][2] = {{0, 2}, {1, 2}};   // [a or b][dy or dx]: inclusive range of ky / kx
    for (int o = 0; o < cout; ++o)
        for (int i = 0; i < cin; ++i) {
            const float* g = w + ((size_t)o * cin + i) * 9;
            for (int ph = 0; ph < 4; ++ph)
                for (int tap = 0; tap < 4; ++tap) {
                    const int a = ph >> 1, b = ph & 1, dy = tap >> 1, dx = tap & 1;
                    float r[3];
                    for (int kx = 0; kx < 3; ++kx) {
                        r[kx] = g[lo[a][dy] * 3 + kx];
                        for (int ky = lo[a][dy] + 1; ky <= hi[a][dy]; ++ky) r[kx] = r[kx] + g[ky * 3 + kx];
                    }
                    float v = r[lo[b][dx]];
                    for (int kx = lo[b][dx] + 1; kx <= hi[b][dx]; ++kx) v = v + r[kx];
                    out[(((size_t)(ph * 4 + tap) * cq + (i >> 2)) * cout_pad + o) * 4 + (i & 3)] = v;
                }
        }
    return out;
}

// The same four phase kernels, each transformed for the 2-D Winograd F(2x2,2x2) of dcx_conv_wino2p.h:
// [phase][xi*3 + nu][cin/4][cout_pad][4];  Wc[dy][.] = (Wp[dy][0], Wp[dy][0] + Wp[dy][1], Wp[dy][1]),
// U[.][nu] = (Wc[0][nu], Wc[0][nu] + Wc[1][nu], Wc[1][nu]) -- fp32, columns first, one addition each.
std::vector<float> pack_conv_ups2w(const float* w, int cout, int cin, int cout_pad) {
    const int cq = cin / 4;
    const std::vector<float> ph = pack_conv_ups2(w, cout, cin, cout_pad);      // [phase*4 + dy*2 + dx][cq][cout_pad][4]
    std::vector<float> out((size_t)36 * cq * cout_pad * 4, 0.0f);
    const size_t plane = (size_t)cq * cout_pad * 4;
    for (int p = 0; p < 4; ++p)
        for (size_t e = 0; e < plane; ++e) {
            float wc[2][3];
            for (int dy = 0; dy < 2; ++dy) {
                const float g0 = ph[(size_t)(p * 4 + dy * 2 + 0) * plane + e], g1 = ph[(size_t)(p * 4 + dy * 2 + 1) * plane + e];
                wc[dy][0] = g0; wc[dy][1] = g0 + g1; wc[dy][2] = g1;
            }
            for (int nu = 0; nu < 3; ++nu) {
                out[(size_t)(p * 9 + 0 * 3 + nu) * plane + e] = wc[0][nu];
                out[(size_t)(p * 9 + 1 * 3 + nu) * plane + e] = wc[0][nu] + wc[1][nu];
                out[(size_t)(p * 9 + 2 * 3 + nu) * plane + e] = wc[1][nu];
            }
        }
    return out;
}

// eval-mode BatchNorm2d as ATen's CPU inference path evaluates it: y = x * alpha + beta with
// alpha = gamma * (1 / sqrt(var + eps)), beta = bn_bias - mean * alpha   (fp32 throughout).
void fold_bn(const float* gamma, const float* bbeta, const float* mean, const float* var, int c, int c_pad,
             std::vector<float>& alpha, std::vector<float>& beta) {
    alpha.assign(c_pad, 0.0f);
    beta.assign(c_pad, 0.0f);
    for (int i = 0; i < c; ++i) {
        const float inv = 1.0f / sqrtf(var[i] + kBnEps);
        const float a = gamma[i] * inv;
        alpha[i] = a;
        beta[i] = bbeta[i] - mean[i] * a;
    }
}

std::vector<float> pad_vec(const float* v, int c, int c_pad) {
    std::vector<float> out(c_pad, 0.0f);
    memcpy(out.data(), v, sizeof(float) * c);
    return out;
}

struct DevLayer {       // one MFMA convolution's parameters on the device
    float* w = nullptr;
    float* w_wino2 = nullptr;  // 3x3 + BN layers only
    float* w_ups2w = nullptr;  // 3x3 + BN layers that read a x2 up-sampled input only: Winograd-transformed phase kernels
    float* bias = nullptr;
    float* alpha = nullptr;
    float* beta = nullptr;
    int cin = 0, cout = 0, cout_pad = 0, ks = 3;
};

int upload(const std::vector<float>& h, float** d) {
    DCX_CHECK_HIP(hipMalloc((void**)d, h.size() * sizeof(float)));
    DCX_CHECK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

void free_layer(DevLayer& l) {
    if (l.w) (void)hipFree(l.w);
    if (l.w_wino2) (void)hipFree(l.w_wino2);
    if (l.w_ups2w) (void)hipFree(l.w_ups2w);
    if (l.bias) (void)hipFree(l.bias);
    if (l.alpha) (void)hipFree(l.alpha);
    if (l.beta) (void)hipFree(l.beta);
    l = DevLayer();
}

// weight/bias (+ optional BN) host pointers of one conv, in state_dict_keys() order
struct HostConv {
    const float* w; const float* b;
    const float* g; const float* be; const float* mu; const float* var;   // null when no BN follows
};

int make_layer(const HostConv& h, int cin, int cout, int ks, DevLayer* out, bool ups_input = false) {
    DevLayer l;
    l.cin = cin; l.cout = cout; l.ks = ks;
    l.cout_pad = dcx_conv_cout_pad(cout);
    int rc = upload(pack_conv(h.w, cout, cin, ks, l.cout_pad), &l.w);
    if (rc == 0) rc = upload(pad_vec(h.b, cout, l.cout_pad), &l.bias);
    if (rc == 0 && h.g != nullptr) {
        std::vector<float> al, be;
        fold_bn(h.g, h.be, h.mu, h.var, cout, l.cout_pad, al, be);
        // MFMA epilogue evaluates y = fma(acc, alpha, beta2) with the conv bias folded in: beta2 = fma(bias, alpha, beta)
        for (int i = 0; i < cout; ++i) be[i] = fmaf(h.b[i], al[i], be[i]);
        rc = upload(al, &l.alpha);
        if (rc == 0) rc = upload(be, &l.beta);
        if (rc == 0 && ks == 3) rc = upload(pack_conv_wino2(h.w, cout, cin, l.cout_pad), &l.w_wino2);
        if (rc == 0 && ks == 3 && ups_input) rc = upload(pack_conv_ups2w(h.w, cout, cin, l.cout_pad), &l.w_ups2w);
    }
    if (rc != 0) { free_layer(l); return rc; }
    *out = l;
    return 0;
}

// Cin = 1 first layer: weights as [tap][64]
int make_first_layer(const HostConv& h, DevLayer* out) {
    DevLayer l;
    l.cin = 1; l.cout = 64; l.cout_pad = 64; l.ks = 3;
    std::vector<float> w(9 * 64);
    for (int o = 0; o < 64; ++o)
        for (int t = 0; t < 9; ++t) w[t * 64 + o] = h.w[o * 9 + t];
    std::vector<float> al, be;
    fold_bn(h.g, h.be, h.mu, h.var, 64, 64, al, be);
    int rc = upload(w, &l.w);
    if (rc == 0) rc = upload(pad_vec(h.b, 64, 64), &l.bias);
    if (rc == 0) rc = upload(al, &l.alpha);
    if (rc == 0) rc = upload(be, &l.beta);
    if (rc != 0) { free_layer(l); return rc; }
    *out = l;
    return 0;
}

// ---- timing ---------------------------------------------------------------------------------
bool g_timing = false;
hipEvent_t g_ev[5];
bool g_ev_ok = false;
int timing_mark(int i, hipStream_t s) {
    if (!g_timing) return 0;
    if (!g_ev_ok) {
        for (auto& e : g_ev) DCX_CHECK_HIP(hipEventCreate(&e));
        g_ev_ok = true;
    }
    DCX_CHECK_HIP(hipEventRecord(g_ev[i], s));
    return 0;
}

}  // namespace

// =============================================================================================
struct dcx_detector {
    int n_ids = 16;
    DevLayer first;                 // conv1a
    DevLayer enc[7];                // conv1b conv2a conv2b conv3a conv3b conv4a conv4b
    DevLayer heads_a;               // convPa | convDa fused along cout (128 -> 512)
    DevLayer head_loc, head_ids;    // convPb (256 -> 65), convDb (256 -> n_ids+1), 1x1 raw
};

struct dcx_refiner {
    DevLayer first;                 // conv1a
    DevLayer mid[9];                // conv1b conv2a conv2b conv3a conv3b conv4a conv4b conv5a conv5b
    DevLayer head_a;                // convPa (64 -> 64)
    float* head_w = nullptr;        // convPb weights [64]
    float head_b = 0.f;
};

namespace {

struct DetWs {
    size_t buf0, buf1, loc, ids, codes, conf, total;
    int ids_quads;
};
DetWs det_layout(int n_ids, int b, int h, int w) {
    DetWs L;
    const size_t hw = (size_t)h * w, cells = (size_t)(h / 8) * (w / 8);
    L.ids_quads = (n_ids + 1 + 3) / 4;
    size_t off = 0;
    L.buf0 = off; off = align_up(off + (size_t)b * 64 * hw * 4, 256);
    L.buf1 = off; off = align_up(off + (size_t)b * 16 * hw * 4, 256);
    L.loc = off;  off = align_up(off + (size_t)b * 68 * cells * 4, 256);
    L.ids = off;  off = align_up(off + (size_t)b * L.ids_quads * 4 * cells * 4, 256);
    L.codes = off; off = align_up(off + (size_t)b * cells * 4, 256);    // decode scratch: packed arg-max per cell
    L.conf = off; off = align_up(off + (size_t)b * cells * 8, 256);     // soft-max probability of the winning (loc, ids) class per cell
    L.total = off;
    return L;
}

struct RefWs {
    size_t buf0, buf1, pval, pidx, total;
};
constexpr int kRefTiles = 32;     // capacity: 64x64 heat-map in 4x32 tiles (Winograd head; 16 tiles of 8x32 for the direct head)
RefWs ref_layout(int p) {
    RefWs L;
    size_t off = 0;
    L.buf0 = off; off = align_up(off + (size_t)p * 65536 * 4, 256);
    L.buf1 = off; off = align_up(off + (size_t)p * 65536 * 4, 256);
    L.pval = off; off = align_up(off + (size_t)p * kRefTiles * 4, 256);
    L.pidx = off; off = align_up(off + (size_t)p * kRefTiles * 4, 256);
    L.total = off;
    return L;
}

DcxConvArgs conv_args(const DevLayer& l, const float* in, int n, int in_cq_total, int in_cq_off, int hin, int win,
                      int ups, int pad, float* out, int out_cq_total, const int* n_limit) {
    DcxConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.w = l.w; a.w_wino2 = l.w_wino2; a.w_ups2w = l.w_ups2w; a.bias = l.bias; a.alpha = l.alpha; a.beta = l.beta; a.out = out;
    a.n_limit = n_limit;
    a.n = n; a.in_cq_total = in_cq_total; a.in_cq_off = in_cq_off; a.cin = l.cin;
    a.hin = hin; a.win = win; a.ups = ups; a.pad = pad;
    a.ho = (hin << ups) + 2 * pad - (l.ks - 1);
    a.wo = (win << ups) + 2 * pad - (l.ks - 1);
    a.out_cq_total = out_cq_total; a.out_cq_off = 0;
    a.cout_pad = l.cout_pad; a.cout_quads = (l.cout + 3) / 4; a.cout_real = l.cout;
    return a;
}

}  // namespace

extern "C" const char* dcx_version(void) { return "deepcharuco_amd 0.1 (gfx950, fp32 MFMA)"; }

extern "C" const char* dcx_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case DCX_E_ARG: return "DCX_E_ARG: null pointer or bad scalar argument";
        case DCX_E_SHAPE: return "DCX_E_SHAPE: unsupported shape (H/W below 8, patches not 24x24, batch x kmax too large, ...)";
        case DCX_E_WS: return "DCX_E_WS: workspace too small";
        case DCX_E_NIDS: return "DCX_E_NIDS: n_ids outside [1, 62] or dust_bin outside [0, 255]";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown dcx error";
    }
}

// ---- detector ---------------------------------------------------------------------------------
extern "C" int dcx_detector_create(dcx_detector** out, const float* const* t, int n_tensors, int n_ids) {
    if (!out || !t) return DCX_E_ARG;
    if (n_ids < 1 || n_ids > 62) return DCX_E_NIDS;
    if (n_tensors != 64) return DCX_E_ARG;   // 10 conv+BN (6 tensors) + 2 raw convs (2 tensors)
    for (int i = 0; i < n_tensors; ++i)
        if (!t[i]) return DCX_E_ARG;
    dcx_detector* d = new (std::nothrow) dcx_detector();
    if (!d) return (int)hipErrorOutOfMemory;
    d->n_ids = n_ids;
    auto hc = [&](int base, bool bn) {
        return HostConv{t[base], t[base + 1], bn ? t[base + 2] : nullptr, bn ? t[base + 3] : nullptr,
                        bn ? t[base + 4] : nullptr, bn ? t[base + 5] : nullptr};
    };
    // tensor index of each conv in state_dict_keys("detector") order
    // conv1a 0, conv1b 6, conv2a 12, conv2b 18, conv3a 24, conv3b 30, conv4a 36, conv4b 42,
    // convPa 48, convPb 54, convDa 56, convDb 62
    static const int enc_cin[7] = {64, 64, 64, 64, 128, 128, 128};
    static const int enc_cout[7] = {64, 64, 64, 128, 128, 128, 128};
    int rc = make_first_layer(hc(0, true), &d->first);
    for (int i = 0; rc == 0 && i < 7; ++i) rc = make_layer(hc(6 + 6 * i, true), enc_cin[i], enc_cout[i], 3, &d->enc[i]);
    if (rc == 0) {   // fuse convPa | convDa: same input, 128 -> 256 + 256
        const HostConv pa = hc(48, true), da = hc(56, true);
        const size_t wsz = (size_t)256 * 128 * 9;
        std::vector<float> w(2 * wsz), b(512), g(512), be(512), mu(512), var(512);
        memcpy(w.data(), pa.w, wsz * 4); memcpy(w.data() + wsz, da.w, wsz * 4);
        auto cat = [](std::vector<float>& dst, const float* x, const float* y) {
            memcpy(dst.data(), x, 256 * 4); memcpy(dst.data() + 256, y, 256 * 4);
        };
        cat(b, pa.b, da.b); cat(g, pa.g, da.g); cat(be, pa.be, da.be); cat(mu, pa.mu, da.mu); cat(var, pa.var, da.var);
        rc = make_layer(HostConv{w.data(), b.data(), g.data(), be.data(), mu.data(), var.data()}, 128, 512, 3,
                        &d->heads_a);
    }
    if (rc == 0) rc = make_layer(hc(54, false), 256, 65, 1, &d->head_loc);
    if (rc == 0) rc = make_layer(hc(62, false), 256, n_ids + 1, 1, &d->head_ids);
    if (rc != 0) { dcx_detector_destroy(d); return rc; }
    *out = d;
    return 0;
}

extern "C" int dcx_detector_destroy(dcx_detector* d) {
    if (!d) return 0;
    free_layer(d->first);
    for (auto& l : d->enc) free_layer(l);
    free_layer(d->heads_a); free_layer(d->head_loc); free_layer(d->head_ids);
    delete d;
    return 0;
}

extern "C" size_t dcx_detector_workspace_bytes(const dcx_detector* det, int batch, int height, int width) {
    if (!det || batch <= 0 || height <= 0 || width <= 0) return 0;
    return det_layout(det->n_ids, batch, height, width).total;
}

namespace {
// conv1a .. convPa|convDa; with_heads: also the two raw 1x1 heads into the workspace's C4 logit buffers (dcModel.forward).
// Without them the 512-channel activation is left in buf0 for the fused tail kernel (dcx_tail.hip).
int detector_run(const dcx_detector* det, const uint8_t* d_frames_u8, long frame_stride, int pitch, int pix,
                 const float* d_images_f32, int batch, int height, int width, void* d_ws, size_t ws_bytes,
                 bool with_heads, float* d_loc_nchw, float* d_ids_nchw, int32_t* zero_words, int n_zero, void* stream);
}  // namespace

extern "C" int dcx_detector_forward(const dcx_detector* det, const uint8_t* d_frames_u8, long frame_stride, int pitch,
                                    const float* d_images_f32, int batch, int height, int width, void* d_ws,
                                    size_t ws_bytes, float* d_loc_nchw, float* d_ids_nchw, void* stream) {
    return detector_run(det, d_frames_u8, frame_stride, pitch, DCX_PIX_GRAY8, d_images_f32, batch, height, width, d_ws, ws_bytes, true,
                        d_loc_nchw, d_ids_nchw, nullptr, 0, stream);
}

namespace {
int detector_run(const dcx_detector* det, const uint8_t* d_frames_u8, long frame_stride, int pitch, int pix,
                 const float* d_images_f32, int batch, int height, int width, void* d_ws, size_t ws_bytes,
                 bool with_heads, float* d_loc_nchw, float* d_ids_nchw, int32_t* zero_words, int n_zero, void* stream) {
    if (!det || !d_ws) return DCX_E_ARG;
    if ((d_frames_u8 == nullptr) == (d_images_f32 == nullptr)) return DCX_E_ARG;
    if (batch <= 0 || height < 8 || width < 8) return DCX_E_SHAPE;      // any size >= 8: the three poolings floor (net.py:16)
    const DetWs L = det_layout(det->n_ids, batch, height, width);
    if (ws_bytes < L.total) return DCX_E_WS;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d_ws;
    float* buf0 = (float*)(ws + L.buf0);
    float* buf1 = (float*)(ws + L.buf1);
    float* loc = (float*)(ws + L.loc);
    float* ids = (float*)(ws + L.ids);
    const int h = height, w = width;
    int rc;
    // conv1a + bn1a + relu (net.py:60)
    const int pt0 = dcx_prof_begin(DCX_PROF_CONV1A, batch, s);
    if (d_frames_u8)
        rc = dcx_launch_conv1_u8(d_frames_u8, frame_stride, pitch, pix, batch, h, w, 1, det->first.w, det->first.bias,
                                 det->first.alpha, det->first.beta, buf0, nullptr, zero_words, n_zero, s);
    else
        rc = dcx_launch_conv1_f32(d_images_f32, (long)h * w, w, batch, h, w, 1, det->first.w, det->first.bias,
                                  det->first.alpha, det->first.beta, buf0, nullptr, s);
    if (rc) return rc;
    if ((rc = dcx_prof_end(pt0, s))) return rc;
    // encoder (net.py:61-70): {layer, input divisor, pool}
    struct Step { int layer, div, pool; };
    static const Step steps[7] = {{0, 1, 1}, {1, 2, 0}, {2, 2, 1}, {3, 4, 0}, {4, 4, 1}, {5, 8, 0}, {6, 8, 0}};
    float* src = buf0;
    float* dst = buf1;
    for (const Step& st : steps) {
        const DevLayer& l = det->enc[st.layer];
        DcxConvArgs a = conv_args(l, src, batch, l.cin / 4, 0, h / st.div, w / st.div, 0, 1, dst, l.cout / 4, nullptr);
        rc = dcx_launch_conv_mfma(a, 3, st.pool, DCX_EPI_BNRELU, s);
        if (rc) return rc;
        float* t = src; src = dst; dst = t;
    }
    // after 7 steps: src = buf1 holds conv4b output (128 ch @ H/8 x W/8), dst = buf0
    const int hc = h / 8, wc = w / 8;
    {   // convPa|convDa + BN + ReLU (net.py:73,76) -> 512 channels
        DcxConvArgs a = conv_args(det->heads_a, src, batch, 32, 0, hc, wc, 0, 1, dst, 128, nullptr);
        rc = dcx_launch_conv_mfma(a, 3, 0, DCX_EPI_BNRELU, s);
        if (rc) return rc;
    }
    if (!with_heads) return 0;
    {   // convPb 1x1 (net.py:74): channels 0..255 -> 65 logits; image flattened to 1 x cells
        DcxConvArgs a = conv_args(det->head_loc, dst, batch, 128, 0, 1, hc * wc, 0, 0, loc, 17, nullptr);
        rc = dcx_launch_conv_mfma(a, 1, 0, DCX_EPI_RAW, s);
        if (rc) return rc;
    }
    {   // convDb 1x1 (net.py:77): channels 256..511 -> n_ids+1 logits
        DcxConvArgs a = conv_args(det->head_ids, dst, batch, 128, 64, 1, hc * wc, 0, 0, ids, L.ids_quads, nullptr);
        rc = dcx_launch_conv_mfma(a, 1, 0, DCX_EPI_RAW, s);
        if (rc) return rc;
    }
    if (d_loc_nchw) { rc = dcx_c4_to_nchw(loc, batch, 65, hc, wc, d_loc_nchw, stream); if (rc) return rc; }
    if (d_ids_nchw) { rc = dcx_c4_to_nchw(ids, batch, det->n_ids + 1, hc, wc, d_ids_nchw, stream); if (rc) return rc; }
    return 0;
}
}  // namespace

extern "C" int dcx_detector_decode(const dcx_detector* det, int batch, int height, int width, void* d_ws,
                                   int dust_bin, int kmax, int32_t* d_counts, int32_t* d_rows, int32_t* d_loc_argmax,
                                   int32_t* d_ids_argmax, void* stream) {
    if (!det || !d_ws) return DCX_E_ARG;
    if (batch <= 0 || height < 8 || width < 8) return DCX_E_SHAPE;      // any size >= 8: the three poolings floor (net.py:16)
    const DetWs L = det_layout(det->n_ids, batch, height, width);
    const int hc = height / 8, wc = width / 8;
    const long cells = (long)hc * wc;
    char* ws = (char*)d_ws;
    DcxLogitView lv{(const float*)(ws + L.loc), 17 * 4 * cells, 4 * cells, 4, 1};
    DcxLogitView iv{(const float*)(ws + L.ids), (long)L.ids_quads * 4 * cells, 4 * cells, 4, 1};
    return dcx_launch_decode(lv, iv, batch, 65, det->n_ids + 1, hc, wc, dust_bin, kmax, d_counts, d_rows,
                             d_loc_argmax, d_ids_argmax, (int32_t*)(ws + L.codes), (hipStream_t)stream);
}

// ---- refiner ----------------------------------------------------------------------------------
extern "C" int dcx_refiner_create(dcx_refiner** out, const float* const* t, int n_tensors) {
    if (!out || !t) return DCX_E_ARG;
    if (n_tensors != 68) return DCX_E_ARG;   // 11 conv+BN (6 tensors) + convPb (2 tensors)
    for (int i = 0; i < n_tensors; ++i)
        if (!t[i]) return DCX_E_ARG;
    dcx_refiner* r = new (std::nothrow) dcx_refiner();
    if (!r) return (int)hipErrorOutOfMemory;
    auto hc = [&](int base) { return HostConv{t[base], t[base + 1], t[base + 2], t[base + 3], t[base + 4], t[base + 5]}; };
    // conv1a 0, conv1b 6, conv2a 12, conv2b 18, conv3a 24, conv3b 30, conv4a 36, conv4b 42, conv5a 48,
    // conv5b 54, convPa 60, convPb 66
    static const int cin[9] = {64, 64, 128, 128, 128, 128, 128, 128, 64};
    static const int cout[9] = {64, 128, 128, 128, 128, 128, 128, 64, 64};
    int rc = make_first_layer(hc(0), &r->first);
    // conv4a (i = 5), conv5a (i = 7) and convPa read the x2 up-sampled output of the layer before them (refinenet.py:66,72,78)
    for (int i = 0; rc == 0 && i < 9; ++i) rc = make_layer(hc(6 + 6 * i), cin[i], cout[i], 3, &r->mid[i], i == 5 || i == 7);
    if (rc == 0) rc = make_layer(hc(60), 64, 64, 3, &r->head_a, true);
    if (rc == 0) {
        std::vector<float> hw(t[66], t[66] + 64);   // convPb.weight (1,64,1,1)
        rc = upload(hw, &r->head_w);
        r->head_b = t[67][0];
    }
    if (rc != 0) { dcx_refiner_destroy(r); return rc; }
    *out = r;
    return 0;
}

extern "C" int dcx_refiner_destroy(dcx_refiner* r) {
    if (!r) return 0;
    free_layer(r->first);
    for (auto& l : r->mid) free_layer(l);
    free_layer(r->head_a);
    if (r->head_w) (void)hipFree(r->head_w);
    delete r;
    return 0;
}

extern "C" size_t dcx_refiner_workspace_bytes(const dcx_refiner* rf, int max_patches) {
    if (!rf || max_patches <= 0) return 0;
    return ref_layout(max_patches).total;
}

namespace {
// n_hint: how many of the max_patches slots are expected to be live (0: unknown).  The launches are sized for the capacity
// and skip dead patches through d_total on the device; the hint only steers the tile cost model (at bs=1 the capacity is 64
// slots but ~16 corners fire: tiles chosen for 64 patches run a 4x longer serial chain per workgroup than needed).
struct FrameSrc { const uint8_t* frames; long frame_stride; int pitch, pix, height, width; };   // pipeline: patches come out of the frames
int refiner_run(const dcx_refiner* rf, const float* d_patches, const FrameSrc* fsrc, int max_patches, int n_hint,
                const int32_t* d_total, const int32_t* d_table, void* d_ws, size_t ws_bytes, int32_t* d_corners, float* d_xy,
                float* d_heat, void* stream);
}  // namespace

extern "C" int dcx_refiner_forward(const dcx_refiner* rf, const float* d_patches, int max_patches,
                                   const int32_t* d_total, const int32_t* d_table, void* d_ws, size_t ws_bytes,
                                   int32_t* d_corners, float* d_xy, float* d_heat, void* stream) {
    return refiner_run(rf, d_patches, nullptr, max_patches, 0, d_total, d_table, d_ws, ws_bytes, d_corners, d_xy, d_heat, stream);
}

namespace {
int refiner_run(const dcx_refiner* rf, const float* d_patches, const FrameSrc* fsrc, int max_patches, int n_hint,
                const int32_t* d_total, const int32_t* d_table, void* d_ws, size_t ws_bytes, int32_t* d_corners, float* d_xy,
                float* d_heat, void* stream) {
    if (!rf || !d_ws || ((d_patches == nullptr) == (fsrc == nullptr))) return DCX_E_ARG;
    if (fsrc != nullptr && (d_total == nullptr || d_table == nullptr)) return DCX_E_ARG;
    if (d_xy != nullptr && d_table == nullptr) return DCX_E_ARG;
    if (max_patches <= 0 || max_patches > (1 << 22)) return DCX_E_SHAPE;
    const RefWs L = ref_layout(max_patches);
    if (ws_bytes < L.total) return DCX_E_WS;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d_ws;
    float* buf0 = (float*)(ws + L.buf0);
    float* buf1 = (float*)(ws + L.buf1);
    const int p = max_patches;
    const int* lim = d_total;
    // conv1a (pad 0) 24 -> 22 (refinenet.py:56); in the pipeline fused with extract_patches (model_utils.py:19-36)
    const int pt0 = dcx_prof_begin(DCX_PROF_PATCHES, p, s);
    int rc = fsrc != nullptr
        ? dcx_launch_conv1_patches_u8(fsrc->frames, fsrc->frame_stride, fsrc->pitch, fsrc->pix, fsrc->height, fsrc->width, d_table, d_total, p, n_hint,
                                      rf->first.w, rf->first.bias, rf->first.alpha, rf->first.beta, buf0, s)
        : dcx_launch_conv1_f32(d_patches, 576, 24, p, 24, 24, 0, rf->first.w, rf->first.bias, rf->first.alpha,
                               rf->first.beta, buf0, lim, s);
    if (rc) return rc;
    if ((rc = dcx_prof_end(pt0, s))) return rc;
    // refinenet.py:57-78: {layer, input size, ups-on-read, pad, pool}
    struct Step { int layer, hin, ups, pad, pool; };
    static const Step steps[9] = {
        {0, 22, 0, 0, 0},   // conv1b 22 -> 20
        {1, 20, 0, 0, 0},   // conv2a 20 -> 18
        {2, 18, 0, 0, 1},   // conv2b 18 -> 16 -> pool 8
        {3, 8, 0, 1, 0},    // conv3a
        {4, 8, 0, 1, 0},    // conv3b (its x2 up-sampling is applied by the next layer's read)
        {5, 8, 1, 1, 0},    // conv4a on 16x16
        {6, 16, 0, 1, 0},   // conv4b
        {7, 16, 1, 1, 0},   // conv5a on 32x32
        {8, 32, 0, 1, 0},   // conv5b
    };
    float* src = buf0;
    float* dst = buf1;
    for (const Step& st : steps) {
        const DevLayer& l = rf->mid[st.layer];
        DcxConvArgs a = conv_args(l, src, p, l.cin / 4, 0, st.hin, st.hin, st.ups, st.pad, dst, l.cout / 4, lim);
        a.n_hint = lim != nullptr ? n_hint : 0;
        rc = dcx_launch_conv_mfma(a, 3, st.pool, DCX_EPI_BNRELU, s);
        if (rc) return rc;
        float* t = src; src = dst; dst = t;
    }
    int heat_tiles = 0;
    {   // convPa on the up-sampled 64x64 + BN + ReLU + convPb 1x1 + per-tile arg-max (refinenet.py:80-81,108-111)
        DcxConvArgs a = conv_args(rf->head_a, src, p, 16, 0, 32, 32, 1, 1, nullptr, 16, lim);
        a.head_w = rf->head_w; a.head_b = rf->head_b; a.heat = d_heat;
        a.part_val = (float*)(ws + L.pval); a.part_idx = (int*)(ws + L.pidx);
        heat_tiles = dcx_conv_heat_tiles(a.ho, a.wo, a.w_ups2w != nullptr ? 1 : 0);
        if (heat_tiles <= 0 || heat_tiles > kRefTiles) return DCX_E_SHAPE;
        rc = dcx_launch_conv_mfma(a, 3, 0, DCX_EPI_HEAT, s);
        if (rc) return rc;
    }
    // (finalising inside the head kernel -- per-patch tickets, the last work item reduces -- was built and measured in round 5: the
    //  device-scope fence ahead of the ticket drains the item's software pipeline, 0.359 -> 0.693 ms at bs=32;
    //  profiles/experiments/r05_fused_finalize.md)
    const int pt1 = dcx_prof_begin(DCX_PROF_FINALIZE, p, s);
    rc = dcx_launch_refine_finalize((const float*)(ws + L.pval), (const int*)(ws + L.pidx), heat_tiles, 64, p, lim,
                                    d_table, d_corners, d_xy, s);
    if (rc) return rc;
    return dcx_prof_end(pt1, s);
}
}  // namespace

// ---- whole pipeline -----------------------------------------------------------------------------
namespace {
// ctrl: int32 words cleared by the detector's first kernel: [0] = pool cursor (ends as the batch's firing-cell count = RefineNet's
// n_limit), [64 .. 64 + B) = the frames' tail tickets
struct PipeWs { size_t det, table, ctrl, ref, total; int ctrl_words; };
PipeWs pipe_layout(const dcx_detector* det, const dcx_refiner* rf, int b, int h, int w, int pool) {
    PipeWs L;
    size_t off = 0;
    L.det = off; off = align_up(off + det_layout(det->n_ids, b, h, w).total, 256);
    L.table = off; off = align_up(off + (size_t)pool * 16, 256);
    L.ctrl_words = 64 + b;
    L.ctrl = off; off = align_up(off + (size_t)L.ctrl_words * 4, 256);
    L.ref = off; off = align_up(off + (rf ? ref_layout(pool).total : 0), 256);
    L.total = off;
    return L;
}
}  // namespace

extern "C" size_t dcx_pipeline_workspace_bytes(const dcx_detector* det, const dcx_refiner* rf, int batch, int height,
                                               int width, int pool) {
    if (!det || batch <= 0 || height <= 0 || width <= 0 || pool <= 0) return 0;
    return pipe_layout(det, rf, batch, height, width, pool).total;
}

namespace {
// the whole path for frames [0, batch) on ONE stream
int infer_range(const dcx_detector* det, const dcx_refiner* rf, const uint8_t* d_frames_u8, long frame_stride, int pitch, int pix,
                int batch, int height, int width, int dust_bin, int pool, char* ws, size_t ws_bytes, int32_t* d_counts,
                int32_t* d_starts, int32_t* d_rows, float* d_xy, float* d_conf, hipStream_t s, bool timing) {
    void* stream = (void*)s;
    const PipeWs L = pipe_layout(det, rf, batch, height, width, pool);
    if (ws_bytes < L.total) return DCX_E_WS;
    int rc = timing ? timing_mark(0, s) : 0;
    if (rc) return rc;
    int32_t* ctrl = (int32_t*)(ws + L.ctrl);
    // detector up to convPa|convDa (its first kernel also clears the pool cursor and the frame tickets), then ONE kernel for the
    // 1x1 heads + per-cell arg-max + dust-bin rule + ordered compaction into the batch's corner pool (the logits never reach HBM;
    // dcModel.forward keeps the separate heads because it has to return them)
    rc = detector_run(det, d_frames_u8, frame_stride, pitch, pix, nullptr, batch, height, width, ws + L.det, L.table - L.det,
                      false, nullptr, nullptr, ctrl, L.ctrl_words, stream);
    if (rc) return rc;
    if (timing && (rc = timing_mark(1, s))) return rc;
    int32_t* table = (int32_t*)(ws + L.table);
    {
        const DetWs D = det_layout(det->n_ids, batch, height, width);
        const int hc = height / 8, wc = width / 8;
        const float* act = (const float*)(ws + L.det + D.buf0);
        int32_t* codes = (int32_t*)(ws + L.det + D.codes);
        DcxPoolOut po;
        po.tickets = ctrl + 64; po.cursor = ctrl; po.counts = d_counts; po.starts = d_starts; po.rows = d_rows;
        po.table = rf ? table : nullptr; po.conf = d_conf; po.conf_cells = (float*)(ws + L.det + D.conf);
        po.wc = wc; po.pool = pool;
        const int pt = dcx_prof_begin(DCX_PROF_TAIL, batch, s);
        rc = dcx_launch_tail(act, batch, hc * wc, det->head_loc.w, det->head_loc.bias, det->head_ids.w, det->head_ids.bias,
                             det->head_ids.cout_pad, det->n_ids + 1, dust_bin, codes, nullptr, nullptr, &po, s);
        if (rc) return rc;
        if ((rc = dcx_prof_end(pt, s))) return rc;
    }
    if (timing && (rc = timing_mark(2, s))) return rc;
    if (rf == nullptr) return timing ? timing_mark(3, s) : 0;
    // expected live patches: a board has n_ids corners, so ~n_ids fire per frame (the pool is usually larger)
    const long expect = (long)batch * det->n_ids;
    const FrameSrc src{d_frames_u8, frame_stride, pitch, pix, height, width};       // RefineNet's conv1a gathers its 24x24 patches itself
    rc = refiner_run(rf, nullptr, &src, pool, (int)(expect < pool ? expect : pool), ctrl, table,
                     ws + L.ref, L.total - L.ref, nullptr, d_xy, nullptr, stream);
    if (rc) return rc;
    return timing ? timing_mark(3, s) : 0;
}
}  // namespace

extern "C" int dcx_infer_batch(const dcx_detector* det, const dcx_refiner* rf, const uint8_t* d_frames_u8,
                               long frame_stride, int pitch, int pixel_format, int batch, int height, int width, int dust_bin,
                               int pool, void* d_ws, size_t ws_bytes, int32_t* d_counts, int32_t* d_starts, int32_t* d_rows,
                               float* d_xy, float* d_conf, void* stream) {
    if (!det || !d_frames_u8 || !d_ws || !d_counts || !d_starts || !d_rows) return DCX_E_ARG;
    if (rf != nullptr && d_xy == nullptr) return DCX_E_ARG;
    if (pixel_format != DCX_PIX_GRAY8 && pixel_format != DCX_PIX_BGR8 && pixel_format != DCX_PIX_BGR8_LEGACY14) return DCX_E_ARG;
    if (pool <= 0 || pool > (1 << 22)) return DCX_E_SHAPE;
    if (batch <= 0 || batch > (1 << 20) || height < 8 || width < 8) return DCX_E_SHAPE;      // any size >= 8: the three poolings floor (net.py:16)
    if (pitch < width * (pixel_format == DCX_PIX_GRAY8 ? 1 : 3)) return DCX_E_SHAPE;
    if (dust_bin < 0 || dust_bin > 255) return DCX_E_NIDS;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d_ws;
    // (A fork/join two-stream variant -- two half-batches of one call on two streams -- was measured and dropped: -1 % at
    //  bs=32.  What does gain ~7 % is overlapping CONSECUTIVE batches on two streams, which is the caller's business:
    //  tools/two_stream_probe.py.)
    return infer_range(det, rf, d_frames_u8, frame_stride, pitch, pixel_format, batch, height, width, dust_bin, pool, ws, ws_bytes,
                       d_counts, d_starts, d_rows, d_xy, d_conf, s, true);
}

extern "C" int dcx_set_timing(int enabled) { g_timing = enabled != 0; return 0; }
extern "C" int dcx_get_timing(void) { return g_timing ? 1 : 0; }

extern "C" int dcx_last_timings(float* h_ms4) {
    if (!h_ms4) return DCX_E_ARG;
    if (!g_ev_ok) return DCX_E_ARG;
    DCX_CHECK_HIP(hipEventElapsedTime(&h_ms4[0], g_ev[0], g_ev[1]));
    DCX_CHECK_HIP(hipEventElapsedTime(&h_ms4[1], g_ev[1], g_ev[2]));
    DCX_CHECK_HIP(hipEventElapsedTime(&h_ms4[2], g_ev[2], g_ev[3]));
    DCX_CHECK_HIP(hipEventElapsedTime(&h_ms4[3], g_ev[0], g_ev[3]));
    return 0;
}

// ---- stage-level test entry -----------------------------------------------------------------------
extern "C" int dcx_conv_layer(const float* d_in, int n, int cin, int hin, int win, const float* h_w, const float* h_b,
                              const float* h_g, const float* h_be, const float* h_mu, const float* h_var, int cout,
                              int ksize, int pad, int ups, int pool, int relu, float* d_out, void* stream) {
    if (!d_in || !h_w || !h_b || !d_out) return DCX_E_ARG;
    if (cin % 32 != 0 || (ksize != 3 && ksize != 1) || n <= 0) return DCX_E_SHAPE;
    const bool bn = h_g != nullptr;
    if (bn != (relu != 0)) return DCX_E_ARG;   // kernels implement conv+BN+ReLU or raw conv+bias
    if (bn && (!h_be || !h_mu || !h_var)) return DCX_E_ARG;
    DevLayer l;
    int rc = make_layer(HostConv{h_w, h_b, h_g, h_be, h_mu, h_var}, cin, cout, ksize, &l, ups != 0);
    if (rc) return rc;
    DcxConvArgs a;
    if (ksize == 1) {
        if (pad != 0 || ups != 0 || pool != 0) { free_layer(l); return DCX_E_SHAPE; }
        a = conv_args(l, d_in, n, cin / 4, 0, 1, hin * win, 0, 0, d_out, (cout + 3) / 4, nullptr);
    } else {
        a = conv_args(l, d_in, n, cin / 4, 0, hin, win, ups, pad, d_out, (cout + 3) / 4, nullptr);
    }
    rc = dcx_launch_conv_mfma(a, ksize, pool, bn ? DCX_EPI_BNRELU : DCX_EPI_RAW, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    free_layer(l);
    if (rc) return rc;
    return (int)e;
}
