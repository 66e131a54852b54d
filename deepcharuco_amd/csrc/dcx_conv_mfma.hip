// dcx_conv_mfma.hip -- instantiations and tile selection for the MFMA convolution kernel.
#include "dcx_conv_mfma.h"
#include "dcx_conv_wino.h"
#include "dcx_conv_wino2.h"
#include "dcx_conv_wino2h.h"
#include "dcx_conv_wino2p.h"

#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

struct CfgEntry {
    int cout_tile, cap, th, tw, ks, pool, epi, acc_tiles;
    int inlane;   // pooled layout with the 2x2 window inside one lane (MT=1, NT=4): no cross-lane max
    int group;    // images per work item (2-D Winograd grouped tiles for small maps); 1 otherwise
    int wino;     // 1: 1-D Winograd F(2,3) kernel (dcx_conv_wino.h), 2/3 of the MFMAs; 2: 2-D F(2x2,3x3) (dcx_conv_wino2.h), 4/9
    int ups2;     // 1: phase variant of the direct kernel for 3x3 layers that read a x2 up-sampled input (4/9 of the MFMAs, no
                  //    transform): th x tw is a LOW-RESOLUTION tile, every tile is four work items (dcx_conv_mfma.h)
    int (*launch)(DcxConvArgs, hipStream_t);
    const char* name;
};

#define DCX_CFG(WM, WN, MT, NT, TH, TW, KS, POOL, EPI)                                              \
    { WM * MT * 32, WN * NT * 32, TH, TW, KS, POOL, EPI, MT * NT, ((POOL) != 0 && MT == 1 && NT == 4) ? 1 : 0, 1, 0, 0, \
      &dcx_conv_launch_cfg<DcxConvCfg<WM, WN, MT, NT, TH, TW, KS, (POOL) != 0, EPI>>,                 \
      "dcx_conv_mfma_kernel<DcxConvCfg<" #WM "," #WN "," #MT "," #NT "," #TH "," #TW "," #KS "," #POOL "," #EPI ">>" }

// phase variant (x2 up-sampled input): the entry's ks stays 3 (the LAYER is 3x3), the kernel runs 2x2 taps
#define DCX_PCFG(WM, WN, MT, NT, TH, TW, EPI)                                                          \
    { WM * MT * 32, WN * NT * 32, TH, TW, 3, 0, EPI, MT * NT, 0, 1, 0, 1,                                   \
      &dcx_conv_launch_cfg<DcxConvCfg<WM, WN, MT, NT, TH, TW, 2, false, EPI, true>>,                       \
      "dcx_conv_mfma_kernel<DcxConvCfg<" #WM "," #WN "," #MT "," #NT "," #TH "," #TW ",2,0," #EPI ",PH>>" }

#define DCX_WCFG(WM, WN, TH, TW, POOL)                                                                \
    { WM * 32, WN * 64, TH, TW, 3, POOL, DCX_EPI_BNRELU, 4, 0, 1, 1, 0,                                     \
      &dcx_conv_wino_launch_cfg<DcxWinoCfg<WM, WN, TH, TW, (POOL) != 0>>,                                  \
      "dcx_conv_wino_kernel<DcxWinoCfg<" #WM "," #WN "," #TH "," #TW "," #POOL ">>" }
#define DCX_WCFG_HEAT(WM, WN, TH, TW)                                                                  \
    { WM * 32, WN * 64, TH, TW, 3, 0, DCX_EPI_HEAT, 4, 0, 1, 1, 0,                                          \
      &dcx_conv_wino_launch_cfg<DcxWinoCfg<WM, WN, TH, TW, false, DCX_EPI_HEAT>>,                          \
      "dcx_conv_wino_kernel<DcxWinoCfg<" #WM "," #WN "," #TH "," #TW ",0,DCX_EPI_HEAT>>" }

#define DCX_W2CFG(TH, TW, POOL)                                                                       \
    { 64, 256, TH, TW, 3, POOL, DCX_EPI_BNRELU, 16, 0, 1, 2, 0,                                              \
      &dcx_conv_wino2_launch_cfg<DcxWino2Cfg<TH, TW, (POOL) != 0>>,                                        \
      "dcx_conv_wino2_kernel<DcxWino2Cfg<" #TH "," #TW "," #POOL ">>" }

// 2-D Winograd on half-size tiles (dcx_conv_wino2h.h): 64 couts x 32 2x2-tiles, 128 accumulators, two workgroups per CU
#define DCX_W2HCFG(TH, TW, POOL)                                                                      \
    { 64, 128, TH, TW, 3, POOL, DCX_EPI_BNRELU, 8, 0, 1, 3, 0,                                               \
      &dcx_conv_wino2h_launch_cfg<DcxWino2hCfg<TH, TW, (POOL) != 0>>,                                      \
      "dcx_conv_wino2h_kernel<DcxWino2hCfg<" #TH "," #TW "," #POOL ">>" }

// phase variant as a 2-D Winograd F(2x2,2x2) per phase (dcx_conv_wino2p.h): th x tw is a LOW-RESOLUTION tile of one phase
#define DCX_W2PCFG(TH, TW, EPI)                                                                       \
    { 64, 128, TH, TW, 3, 0, EPI, 5, 0, 1, 4, 1,                                                             \
      &dcx_conv_wino2p_launch_cfg<DcxWino2pCfg<TH, TW, EPI>>,                                              \
      "dcx_conv_wino2p_kernel<DcxWino2pCfg<" #TH "," #TW "," #EPI ">>" }

#define DCX_W2PCFG_G(TH, TW, G)                                                                       \
    { 64, 128, TH, TW, 3, 0, DCX_EPI_BNRELU, 5, 0, G, 4, 1,                                                  \
      &dcx_conv_wino2p_launch_cfg<DcxWino2pCfg<TH, TW, DCX_EPI_BNRELU, G>>,                                \
      "dcx_conv_wino2p_kernel<DcxWino2pCfg<" #TH "," #TW ",DCX_EPI_BNRELU," #G ">>" }

#define DCX_W2CFG_G(TH, TW, G)                                                                        \
    { 64, 256, TH, TW, 3, 0, DCX_EPI_BNRELU, 16, 0, G, 2, 0,                                                \
      &dcx_conv_wino2_launch_cfg<DcxWino2Cfg<TH, TW, false, DCX_EPI_BNRELU, G>>,                           \
      "dcx_conv_wino2_kernel<DcxWino2Cfg<" #TH "," #TW ",0,DCX_EPI_BNRELU," #G ">>" }
#define DCX_W2CFG_HEAT(TH, TW)                                                                        \
    { 64, 256, TH, TW, 3, 0, DCX_EPI_HEAT, 16, 0, 1, 2, 0,                                                   \
      &dcx_conv_wino2_launch_cfg<DcxWino2Cfg<TH, TW, false, DCX_EPI_HEAT>>,                                \
      "dcx_conv_wino2_kernel<DcxWino2Cfg<" #TH "," #TW ",0,DCX_EPI_HEAT>>" }

// Wave layouts:  A = 1x4 waves, 64 couts x 256 px   B = 2x2 waves, 128 couts x 128 px
//                C = 4x1 waves, 128 couts x 64 px
const CfgEntry kCfgs[] = {
    // 3x3 + BN + ReLU
    DCX_CFG(1, 4, 2, 4, 16, 32, 3, 0, DCX_EPI_BNRELU),   // big wave tile: 64 cout x 512 px, 1 workgroup/CU
    DCX_CFG(1, 4, 2, 2, 8, 32, 3, 0, DCX_EPI_BNRELU),
    DCX_CFG(1, 4, 2, 2, 12, 20, 3, 0, DCX_EPI_BNRELU),
    DCX_CFG(1, 4, 2, 2, 6, 40, 3, 0, DCX_EPI_BNRELU),
    DCX_CFG(1, 4, 2, 2, 16, 16, 3, 0, DCX_EPI_BNRELU),
    DCX_CFG(1, 4, 2, 2, 10, 20, 3, 0, DCX_EPI_BNRELU),
    DCX_CFG(2, 2, 2, 2, 6, 18, 3, 0, DCX_EPI_BNRELU),
    DCX_CFG(2, 2, 2, 2, 8, 16, 3, 0, DCX_EPI_BNRELU),
    DCX_CFG(4, 1, 1, 2, 8, 8, 3, 0, DCX_EPI_BNRELU),
    DCX_CFG(2, 2, 1, 1, 8, 8, 3, 0, DCX_EPI_BNRELU),     // S: 64 cout x 64 px, 32x32 per wave (small batches / small maps)
    // 3x3 + BN + ReLU + 2x2 max-pool
    DCX_CFG(1, 4, 2, 4, 16, 32, 3, 1, DCX_EPI_BNRELU),
    DCX_CFG(1, 4, 2, 2, 8, 32, 3, 1, DCX_EPI_BNRELU),
    DCX_CFG(1, 4, 2, 2, 12, 20, 3, 1, DCX_EPI_BNRELU),
    DCX_CFG(1, 4, 2, 2, 6, 40, 3, 1, DCX_EPI_BNRELU),
    DCX_CFG(1, 4, 2, 2, 16, 16, 3, 1, DCX_EPI_BNRELU),
    DCX_CFG(2, 2, 2, 2, 8, 16, 3, 1, DCX_EPI_BNRELU),
    DCX_CFG(2, 2, 1, 1, 8, 8, 3, 1, DCX_EPI_BNRELU),
    // 1x1, raw (image flattened to 1 x P by the caller)
    DCX_CFG(1, 4, 2, 2, 1, 256, 1, 0, DCX_EPI_RAW),
    // RefineNet head: 3x3 + BN + ReLU + 1x1 -> 1 channel + tile arg-max
    DCX_CFG(1, 4, 2, 4, 16, 32, 3, 0, DCX_EPI_HEAT),
    DCX_CFG(1, 4, 2, 2, 8, 32, 3, 0, DCX_EPI_HEAT),
    // pooled, in-lane window (2x2 waves, 32 couts x 128 px per wave): measured +1% on conv1b/conv2b at 8x32,
    // no gain at 12x20 / 16x16 (A/B in profiles/README.md), so only the 8x32 tile has this variant
    DCX_CFG(2, 2, 1, 4, 8, 32, 3, 1, DCX_EPI_BNRELU),
    // 1-D Winograd F(2,3): 2x2 waves, 64 couts x 64 output pairs (128 px)
    DCX_WCFG(2, 2, 4, 32, 0),
    DCX_WCFG(2, 2, 8, 16, 0),
    DCX_WCFG(2, 2, 6, 20, 0),
    DCX_WCFG(2, 2, 4, 32, 1),
    DCX_WCFG(2, 2, 8, 16, 1),
    DCX_WCFG(2, 2, 6, 20, 1),
    DCX_WCFG_HEAT(2, 2, 4, 32),
    // 2-D Winograd F(2x2,3x3): 64 couts x 64 2x2-tiles, 256 accumulator registers, one workgroup per CU
    DCX_W2CFG(16, 16, 0),
    DCX_W2CFG(8, 32, 0),
    DCX_W2CFG(6, 40, 0),      // 30x40 maps: 3 x 20 tiles
    DCX_W2CFG(24, 10, 0),     // RefineNet's 24 / 22 / 20-pixel maps: 12 x 5 tiles
    DCX_W2CFG(16, 16, 1),
    DCX_W2CFG(8, 32, 1),
    DCX_W2CFG_HEAT(16, 16),
    DCX_W2CFG_G(8, 8, 4),     // four whole 8x8 maps (RefineNet after its pool) per work item
    // phase variant for the layers behind RefineNet's three x2 up-samplings (conv4a 8->16, conv5a 16->32, convPa 32->64):
    // low-resolution tiles 8x32 / 16x16 (A layout, 64 couts x 256 px) and 8x8 (S layout)
    DCX_PCFG(1, 4, 2, 2, 8, 32, DCX_EPI_BNRELU),
    DCX_PCFG(1, 4, 2, 2, 16, 16, DCX_EPI_BNRELU),
    DCX_PCFG(2, 2, 1, 1, 8, 8, DCX_EPI_BNRELU),
    DCX_PCFG(1, 4, 2, 2, 8, 32, DCX_EPI_HEAT),
    // 2-D Winograd, half-size tiles, two workgroups per CU (cout_pad <= 128: the per-channel constants must leave room for two)
    DCX_W2HCFG(8, 16, 0),
    DCX_W2HCFG(8, 16, 1),
    DCX_W2HCFG(6, 20, 0),     // 3 x 10 tiles: 30x40 and 60x80 maps without padding, RefineNet's 18/20-pixel maps at 83-90 %
    DCX_W2HCFG(6, 20, 1),
    // phase variant + F(2x2,2x2) per phase: 2.25 multiply-adds per output pixel (the layer as written: 9)
    DCX_W2PCFG(8, 16, DCX_EPI_BNRELU),
    DCX_W2PCFG(8, 16, DCX_EPI_HEAT),
    DCX_W2PCFG_G(8, 8, 2),    // two whole 8x8 low-resolution maps (RefineNet conv4a) per work item
};

int dcx_wino2h_mode() {   // DCX_WINO2H: 0 = never, 1 = cost model (default), 2 = whenever it can run the layer (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_WINO2H"); v = e ? atoi(e) : 1; }
    return v;
}

int dcx_ups2w_mode() {   // DCX_UPS2W: 0 = never, 1 = cost model (default), 2 = whenever it can run the layer (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_UPS2W"); v = e ? atoi(e) : 1; }
    return v;
}

int dcx_ups2_enabled() {   // on by default; DCX_UPS2=0 keeps up-sampled layers on the Winograd / direct kernels (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_UPS2"); v = (e && !atoi(e)) ? 0 : 1; }
    return v;
}

int dcx_wino2_enabled() {   // on by default; DCX_WINO2=0 keeps the 1-D Winograd / direct kernels (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_WINO2"); v = (e && !atoi(e)) ? 0 : 1; }
    return v;
}

int dcx_wino_enabled() {   // on by default; DCX_WINO=0 keeps every layer on the direct kernels (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_WINO"); v = (e && !atoi(e)) ? 0 : 1; }
    return v;
}

int dcx_inlane_pool_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_INLANE_POOL"); v = (e && !atoi(e)) ? 0 : 1; }
    return v;
}

// Deterministic mode (dcx_set_deterministic / DCX_DETERMINISTIC=1): every layer runs on the direct kernels, whose fp32
// summation order per output element (chunk / tap / s / j, dcx_conv_mfma.h) does not depend on the tile, the batch size
// or the CU count -- logits are then bit-identical for a frame alone and inside any batch, on any device.  Default off:
// the cost model may pick a Winograd family for large launches, whose logits differ from the direct ones in the last
// bits (each family is bit-exact against its own restatement; arg-max outputs agree except on exact near-ties).
int g_deterministic = -1;
int dcx_deterministic_enabled() {
    if (g_deterministic < 0) { const char* e = getenv("DCX_DETERMINISTIC"); g_deterministic = (e && atoi(e)) ? 1 : 0; }
    return g_deterministic;
}

int dcx_big_tiles_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_BIG_TILES"); v = (e && atoi(e)) ? 0 : 1; }   // opt-in: measured slower
    return v;
}

// Tile choice = argmin of a small cost model (cycles on the busiest CU), calibrated with the in-kernel probes
// (tools/unit_probe.py):  rounds = ceil(items / #CU) work items run one after the other on a CU (co-resident
// workgroups share its FMA pipes), each costing its MFMA cycles -- padded pixels included, so tile utilisation is
// accounted for -- plus ~520 cycles per 16-channel unit (barrier, first LDS wait, scalar bookkeeping) and an epilogue
// of ~40 (60 pooled) cycles per accumulator register.  Big tiles win when there is plenty of work (less halo, fewer
// units), the 64x64 S tile when a launch has few items (bs=1, 30x40 maps) or would leave CUs idle in the last round.
const CfgEntry* pick(int n, int cin, int ho, int wo, int cout_pad, int ks, int pool, int epi, int allow_group = 1, int ups = 0) {
    const CfgEntry* best = nullptr;
    double best_cost = 0.0;
    const int n_cu = dcx_device_cu_count();
    // test hook: DCX_FORCE_CFG=<kernel name> restricts the choice to that instantiation when it can run the layer
    // (read on every call so that a test can walk through all instantiations); otherwise the cost model decides
    const char* force = getenv("DCX_FORCE_CFG");
    if (force != nullptr && force[0] != 0) {
        for (const CfgEntry& c : kCfgs)
            if (strcmp(c.name, force) == 0 && c.ks == ks && c.pool == pool && c.epi == epi && cout_pad % c.cout_tile == 0 &&
                (c.group == 1 || (c.wino == 4 ? (ho <= 2 * c.th && wo <= 2 * c.tw) : (allow_group && ho <= c.th && wo <= c.tw))) &&
                (!c.ups2 || ups == 1) &&
                (c.wino != 3 || (cout_pad <= 128 && cin >= 2 * DCX_CCH)) &&
                (c.wino != 4 || (cin >= 2 * DCX_CCH && (epi != DCX_EPI_HEAT || cout_pad == 64))))
                return &c;
    }
    for (const CfgEntry& c : kCfgs) {
        if (c.ks != ks || c.pool != pool || c.epi != epi) continue;
        if (cout_pad % c.cout_tile != 0) continue;
        if (c.inlane && !dcx_inlane_pool_enabled()) continue;
        if (c.wino == 1 && (!dcx_wino_enabled() || dcx_deterministic_enabled())) continue;
        if (c.wino == 2 && (!dcx_wino2_enabled() || dcx_deterministic_enabled())) continue;
        if (c.wino == 3) {   // half-tile 2-D Winograd: two co-resident workgroups share a CU's matrix pipes
            if (!dcx_wino2_enabled() || dcx_deterministic_enabled() || dcx_wino2h_mode() == 0 || cout_pad > 128 || cin < 2 * DCX_CCH) continue;
            const long ht = (long)((ho + c.th - 1) / c.th) * ((wo + c.tw - 1) / c.tw);
            const double items_h = (double)n * (cout_pad / c.cout_tile) * ht;
            const int units_h = cin / DCX_CCH;
            // per item: 64 MFMA-equivalents per unit at 64 cycles + the transform's serial VALU + half an epilogue; stalls are
            // hidden by the co-resident workgroup (measured factor, tools/unit_probe.py)
            const double item_cost_h = (double)units_h * (64 * 64.0 + 560.0) + 2600.0;
            double cost_h = (double)(((long)items_h + n_cu - 1) / n_cu) * item_cost_h;
            if (dcx_wino2h_mode() == 2) cost_h = 1.0;
            if (best == nullptr || cost_h < best_cost) { best_cost = cost_h; best = &c; }
            continue;
        }
        if (c.group > 1 && c.wino != 4 && (!allow_group || ho > c.th || wo > c.tw)) continue;   // grouped tiles: whole small maps only
        if (c.ups2) {      // phase variant: only for layers reading a x2 up-sampled input; tiles are low-resolution, x4 items,
                           // 8 k-steps (4 taps x 2) per 16-channel unit
            if (ups != 1 || !dcx_ups2_enabled() || dcx_deterministic_enabled()) continue;
            if (c.wino == 4) {   // Winograd per phase: 9 x 8 MFMAs of 32 cycles per unit, two co-resident workgroups per CU
                if (dcx_ups2w_mode() == 0 || !dcx_wino2_enabled() || cin < 2 * DCX_CCH || (epi == DCX_EPI_HEAT && cout_pad != 64)) continue;
                if (c.group > 1 && (ho > 2 * c.th || wo > 2 * c.tw)) continue;        // grouped tiles: whole low-resolution maps only
                const long wt = (long)((ho / 2 + c.th - 1) / c.th) * ((wo / 2 + c.tw - 1) / c.tw);
                const double items_w = (double)((n + c.group - 1) / c.group) * (cout_pad / c.cout_tile) * wt * 4;
                const double item_cost_w = (double)(cin / DCX_CCH) * (72 * 32.0 + 500.0) + 2400.0;
                double cost_w = (double)(((long)items_w + n_cu - 1) / n_cu) * item_cost_w;
                if (dcx_ups2w_mode() == 2) cost_w = 1.0;
                if (best == nullptr || cost_w < best_cost) { best_cost = cost_w; best = &c; }
                continue;
            }
            const long lt = (long)((ho / 2 + c.th - 1) / c.th) * ((wo / 2 + c.tw - 1) / c.tw);
            const double items_p = (double)n * (cout_pad / c.cout_tile) * lt * 4;
            const int units_p = cin / DCX_CCH;
            const double item_cost_p = (double)units_p * 8 * (4 * c.acc_tiles) * 64.0 + units_p * 520.0 + c.acc_tiles * 16 * 40.0;
            const double cost_p = (double)(((long)items_p + n_cu - 1) / n_cu) * item_cost_p;
            if (best == nullptr || cost_p < best_cost) { best_cost = cost_p; best = &c; }
            continue;
        }
        const long tiles = (long)((ho + c.th - 1) / c.th) * ((wo + c.tw - 1) / c.tw);
        if (c.cap > 256 && (dcx_big_tiles_disabled() || (double)ho * wo / ((double)tiles * c.cap) < 0.999)) continue;
        const double items = (double)((n + c.group - 1) / c.group) * (cout_pad / c.cout_tile) * tiles;
        const int units = cin / DCX_CCH;
        const int steps = (c.wino ? 3 : ks * ks) * (DCX_CCH / 8);   // Winograd: (ky, 8 channels), 4 positions inside the step
        // per-unit overhead: barrier + first LDS wait + bookkeeping (520); the Winograd units also pay their input
        // transform and the scattered staging loads inside the k-loop (measured ~1100 per unit, tools/unit_probe.py)
        double item_cost = (double)units * steps * (4 * c.acc_tiles) * 64.0 + units * (c.wino ? 1100.0 : 520.0)
                         + c.acc_tiles * 16 * (c.wino ? 25.0 : c.inlane ? 35.0 : (pool ? 60.0 : 40.0));
        if (c.wino == 2)   // 128 MFMAs per unit, one workgroup per CU: nothing hides the transform / barrier / epilogue
            item_cost = (double)units * (128 * 64.0 + 1950.0) + 6200.0;   // measured: k-loop 9,700, barrier + bookkeeping 420, epilogue 6,200
        const double rounds = (double)(((long)items + n_cu - 1) / n_cu);
        double cost = rounds * item_cost;
        if (c.cout_tile == 64 && c.cap == 256) cost *= 0.999;   // deterministic tie-break towards the A layout
        if (best == nullptr || cost < best_cost) { best_cost = cost; best = &c; }
    }
    return best;
}

// ---- per-launch profiling ---------------------------------------------------------------------
struct ProfRec { int kernel_id, n, limited; double flops_per_image; hipEvent_t e0, e1; int slot; };
unsigned long long* g_clk_dev = nullptr;   // [kClkSlots][kClkWords] shader-clock probes of workgroup 0
constexpr int kClkSlots = 1024;
constexpr int kClkWords = 64;   // per launch: 4 clock words + 3 timestamps per unit for the first 20 units
bool g_prof = false;
int g_prof_filter = -1;   // -1: record every launch; k: only launches of kernel id k
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_used = 0;

hipEvent_t prof_event() {
    if (g_pool_used == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_used++];
}

}  // namespace

// which instantiation the cost model picks for a launch shape (no GPU needed; used by tests and tools)
extern "C" const char* dcx_conv_pick_name(int n, int cin, int ho, int wo, int cout, int ks, int pool, int epi) {
    const CfgEntry* c = pick(n, cin, ho, wo, dcx_conv_cout_pad(cout), ks, pool, epi);
    return c ? c->name : "";
}
// same for a layer that reads its input through a nearest x2 up-sampling (ho x wo = the up-sampled output size)
extern "C" const char* dcx_conv_pick_name_ups(int n, int cin, int ho, int wo, int cout, int ks, int pool, int epi, int ups) {
    const CfgEntry* c = pick(n, cin, ho, wo, dcx_conv_cout_pad(cout), ks, pool, epi, ups == 0, ups);
    return c ? c->name : "";
}

extern "C" int dcx_set_deterministic(int enabled) { g_deterministic = enabled ? 1 : 0; return 0; }
extern "C" int dcx_get_deterministic(void) { return dcx_deterministic_enabled(); }

extern "C" int dcx_profile_enable(int enabled) {
    g_prof = enabled != 0;
    if (g_prof) { g_recs.clear(); g_pool_used = 0; }
    return 0;
}
extern "C" int dcx_profile_filter(int kernel_id) { g_prof_filter = kernel_id; return 0; }
extern "C" int dcx_profile_count(void) { return (int)g_recs.size(); }
extern "C" const char* dcx_profile_kernel_name(int id) {
    const int n = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));
    return id >= 0 && id < n ? kCfgs[id].name : "?";
}
// raw probe words of record i (debug aid for kernel tuning; layout documented in dcx_conv_mfma.h)
extern "C" int dcx_profile_probe_words(int record, unsigned long long* out64) {
    if (!out64 || record < 0 || record >= kClkSlots || g_clk_dev == nullptr) return DCX_E_ARG;
    DCX_CHECK_HIP(hipMemcpy(out64, g_clk_dev + (size_t)kClkWords * record, kClkWords * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dcx_profile_clocks(float* ghz, int max_records) {
    if (!ghz) return DCX_E_ARG;
    const int n = (int)g_recs.size() < max_records ? (int)g_recs.size() : max_records;
    std::vector<unsigned long long> h((size_t)kClkWords * kClkSlots, 0ull);
    if (g_clk_dev != nullptr) DCX_CHECK_HIP(hipMemcpy(h.data(), g_clk_dev, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        ghz[i] = 0.f;
        if (i < kClkSlots) {
            const unsigned long long* p = &h[(size_t)kClkWords * i];
            if (p[3] > p[1]) ghz[i] = (float)((double)(p[2] - p[0]) / (double)(p[3] - p[1]) * 0.1);   // realtime = 100 MHz
        }
    }
    return n;
}

extern "C" int dcx_profile_fetch(int* kernel_ids, int* n_images, int* limited, double* flops_per_image, float* ms,
                                 int max_records) {
    if (!kernel_ids || !n_images || !limited || !flops_per_image || !ms) return DCX_E_ARG;
    int k = 0;
    for (const ProfRec& r : g_recs) {
        if (k >= max_records) break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) t = -1.f;
        kernel_ids[k] = r.kernel_id; n_images[k] = r.n; limited[k] = r.limited;
        flops_per_image[k] = r.flops_per_image; ms[k] = t;
        ++k;
    }
    return k;
}

// cout <= 64 uses the 64-cout wave layout; anything larger is padded to the 128-cout layouts' multiple.
int dcx_conv_cout_pad(int cout) { return cout <= 64 ? 64 : (cout + 127) / 128 * 128; }

int dcx_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < DCX_MAX_DEVICES ? dev : DCX_MAX_DEVICES - 1;
}

int dcx_device_cu_count() {
    static int cus[DCX_MAX_DEVICES] = {};      // per device: a process may drive several GPUs (dcModel.to('cuda:1'))
    const int dev = dcx_current_device();
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus[dev] = v;
        else
            cus[dev] = 256;   // MI355X
    }
    return cus[dev];
}

int dcx_xcd_walk_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_XCD_WALK"); v = (e && !atoi(e)) ? 0 : 1; }
    return v;
}

int dcx_occupancy_override() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_OCC"); v = e ? atoi(e) : 0; }
    return v;
}

int dcx_conv_heat_tiles(int ho, int wo, int ups) {
    const CfgEntry* c = pick(1 << 20, 64, ho, wo, 64, 3, 0, DCX_EPI_HEAT, 1, ups);
    if (c == nullptr) return 0;
    if (c->ups2) return 4 * ((ho / 2 + c->th - 1) / c->th) * ((wo / 2 + c->tw - 1) / c->tw);
    return ((ho + c->th - 1) / c->th) * ((wo + c->tw - 1) / c->tw);
}

int dcx_launch_conv_mfma(DcxConvArgs a, int ks, int pool, int epi, hipStream_t stream) {
    if (a.in == nullptr || a.w == nullptr || a.bias == nullptr) return DCX_E_ARG;
    if (epi != DCX_EPI_HEAT && a.out == nullptr) return DCX_E_ARG;
    if (epi != DCX_EPI_RAW && (a.alpha == nullptr || a.beta == nullptr)) return DCX_E_ARG;
    if (pool && ((a.ho | a.wo) & 1)) return DCX_E_SHAPE;
    // the fused-head launch must use the tiling dcx_conv_heat_tiles() sized part_val / part_idx for
    const CfgEntry* c = pick(epi == DCX_EPI_HEAT ? (1 << 20) : (a.n_hint > 0 && a.n_hint < a.n ? a.n_hint : a.n), a.cin, a.ho, a.wo, a.cout_pad, ks, pool, epi,
                             a.ups == 0 && a.pad == 1,    // grouped tiles: same-size convolutions read without up-sampling
                             (a.ups == 1 && a.pad == 1 && ks == 3 && a.w_ups2 != nullptr) ? 1 : 0);
    if (c == nullptr) return DCX_E_SHAPE;
    if (!g_prof || (g_prof_filter >= 0 && g_prof_filter != (int)(c - kCfgs))) return c->launch(a, stream);
    ProfRec r;
    r.kernel_id = (int)(c - kCfgs);
    r.n = a.n;
    r.limited = a.n_limit != nullptr;
    r.flops_per_image = 2.0 * a.cout_real * a.cin * ks * ks
                      * (double)a.ho * a.wo;
    r.e0 = prof_event();
    r.e1 = prof_event();
    r.slot = (int)g_recs.size();
    if (g_clk_dev == nullptr) {
        if (hipMalloc((void**)&g_clk_dev, sizeof(unsigned long long) * kClkWords * kClkSlots) != hipSuccess) g_clk_dev = nullptr;
        else (void)hipMemset(g_clk_dev, 0, sizeof(unsigned long long) * kClkWords * kClkSlots);
    }
    if (g_clk_dev != nullptr && r.slot < kClkSlots) a.clk_probe = g_clk_dev + (size_t)kClkWords * r.slot;
    { static int u0 = -1; if (u0 < 0) { const char* e = getenv("DCX_PROBE_U0"); u0 = e ? atoi(e) : 0; } a.probe_u0 = u0; }
    if (!r.e0 || !r.e1) return (int)hipErrorOutOfMemory;
    DCX_CHECK_HIP(hipEventRecord(r.e0, stream));
    const int rc = c->launch(a, stream);
    DCX_CHECK_HIP(hipEventRecord(r.e1, stream));
    if (rc == 0) g_recs.push_back(r);
    return rc;
}
