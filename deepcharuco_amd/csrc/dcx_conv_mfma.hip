// dcx_conv_mfma.hip -- instantiations of the convolution kernels, kernel-FAMILY choice (from the layer shape only) and tile
// choice (cost model) per launch.
#include "dcx_conv_mfma.h"
#include "dcx_conv_wino2h.h"
#include "dcx_conv_wino2hs.h"
#include "dcx_conv_wino2p.h"
#include "dcx_conv_wino2ps.h"

#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

// Kernel families = summation orders.  Every instantiation of a family produces bit-identical outputs for a given layer (the
// order of each output element's fmaf chain does not depend on tile, grouping, batch size or CU count; tests:
// test_every_conv_instantiation_bit_exact), so WHICH family runs a layer decides the bits, and which TILE only the speed.
enum { FAM_DIRECT = 0,   // dcx_conv_mfma.h: implicit GEMM, every multiply-add of the layer as written
       FAM_W2H = 3,      // dcx_conv_wino2h.h: 2-D Winograd F(2x2,3x3), 4/9 of the MACs
       FAM_W2P = 4 };    // dcx_conv_wino2p.h: x2 up-sampled input as four phases x F(2x2,2x2), 1/4 of the MACs

struct CfgEntry {
    int cout_tile, cap, th, tw, ks, pool, epi, acc_tiles;
    int inlane;   // direct pooled layout with the 2x2 window inside one lane (MT=1, NT=4): no cross-lane max
    int group;    // images per work item (grouped tiles for whole small maps); 1 otherwise
    int fam;      // FAM_*
    int (*launch)(DcxConvArgs, hipStream_t);
    const char* name;
};

#define DCX_CFG(WM, WN, MT, NT, TH, TW, KS, POOL, EPI)                                              \
    { WM * MT * 32, WN * NT * 32, TH, TW, KS, POOL, EPI, MT * NT, ((POOL) != 0 && MT == 1 && NT == 4) ? 1 : 0, 1, FAM_DIRECT, \
      &dcx_conv_launch_cfg<DcxConvCfg<WM, WN, MT, NT, TH, TW, KS, (POOL) != 0, EPI>>,                 \
      "dcx_conv_mfma_kernel<DcxConvCfg<" #WM "," #WN "," #MT "," #NT "," #TH "," #TW "," #KS "," #POOL "," #EPI ">>" }

// 2-D Winograd (dcx_conv_wino2h.h): 64 couts x 32 2x2-tiles, 128 accumulators, two workgroups per CU
#define DCX_W2HCFG(TH, TW, POOL, G)                                                                   \
    { 64, 128, TH, TW, 3, POOL, DCX_EPI_BNRELU, 8, 0, G, FAM_W2H,                                            \
      &dcx_conv_wino2h_launch_cfg<DcxWino2hCfg<TH, TW, (POOL) != 0, G>>,                                   \
      "dcx_conv_wino2h_kernel<DcxWino2hCfg<" #TH "," #TW "," #POOL "," #G ">>" }
// the same kernel on 16-tile work items (one MFMA tile block per wave, 64 accumulators): half the serial chain per item, for
// launches too small to fill the chip
#define DCX_W2HCFG_S(TH, TW, POOL)                                                                    \
    { 64, 64, TH, TW, 3, POOL, DCX_EPI_BNRELU, 4, 0, 1, FAM_W2H,                                             \
      &dcx_conv_wino2h_launch_cfg<DcxWino2hCfg<TH, TW, (POOL) != 0, 1, 1>>,                                \
      "dcx_conv_wino2h_kernel<DcxWino2hCfg<" #TH "," #TW "," #POOL ",1,1>>" }

// the same family with an item's 16 positions split over the waves of the workgroup (dcx_conv_wino2hs.h): launches that cannot fill
// the chip (one frame, ~16 patches); 16 CG couts x 16 tiles per item, acc_tiles = 1 marks them for the cost model
#define DCX_W2HSCFG(POOL, CG)                                                                         \
    { 16 * CG, 64, 8, 8, 3, POOL, DCX_EPI_BNRELU, 1, 0, 1, FAM_W2H,                                          \
      &dcx_conv_wino2hs_launch_cfg<DcxWino2hsCfg<(POOL) != 0, CG>>,                                        \
      "dcx_conv_wino2hs_kernel<DcxWino2hsCfg<" #POOL "," #CG ">>" }

// x2 up-sampled input: 2-D Winograd F(2x2,2x2) per phase (dcx_conv_wino2p.h): th x tw is a LOW-RESOLUTION tile of one phase
#define DCX_W2PCFG(TH, TW, EPI, G)                                                                    \
    { 64, 128, TH, TW, 3, 0, EPI, 5, 0, G, FAM_W2P,                                                          \
      &dcx_conv_wino2p_launch_cfg<DcxWino2pCfg<TH, TW, EPI, G>>,                                           \
      "dcx_conv_wino2p_kernel<DcxWino2pCfg<" #TH "," #TW "," #EPI "," #G ">>" }

// Direct-kernel wave layouts:  A = 1x4 waves, 64 couts x 256 px   S / P = 2x2 waves, 64 couts x 64 / 256 px
// (Round 4 trimmed the direct family from 18 to 6 instantiations: on the default path it only runs the raw 1x1 heads; the 3x3
//  tiles below are what deterministic mode -- the A/B reference of the Winograd families -- needs to run every layer shape, not a
//  tuned set: the 12x20 / 6x40 / 16x16 / 10x20 / 6x18 / 8x16 / 4x1-wave variants were speed-ups of a mode no BASELINE config uses.)
// ... and its small-launch shape (dcx_conv_wino2ps.h): 9 positions split over 3 CG waves, 16 CG couts x 16 low-resolution tiles per item
#define DCX_W2PSCFG(CG)                                                                               \
    { 16 * CG, 64, 8, 8, 3, 0, DCX_EPI_BNRELU, 1, 0, 1, FAM_W2P,                                             \
      &dcx_conv_wino2ps_launch_cfg<DcxWino2psCfg<CG>>,                                                     \
      "dcx_conv_wino2ps_kernel<DcxWino2psCfg<" #CG ">>" }

const CfgEntry kCfgs[] = {
    // ---- direct family: the 1x1 heads, deterministic mode (every layer), cin < 32
    // 3x3 + BN + ReLU
    DCX_CFG(1, 4, 2, 2, 8, 32, 3, 0, DCX_EPI_BNRELU),
    DCX_CFG(2, 2, 1, 1, 8, 8, 3, 0, DCX_EPI_BNRELU),     // S: 64 cout x 64 px, 32x32 per wave (small batches / small maps)
    // 3x3 + BN + ReLU + 2x2 max-pool
    DCX_CFG(2, 2, 1, 1, 8, 8, 3, 1, DCX_EPI_BNRELU),
    DCX_CFG(2, 2, 1, 4, 8, 32, 3, 1, DCX_EPI_BNRELU),    // P: pooling window inside one lane (2x2 waves, 32 couts x 128 px per wave)
    // 1x1, raw (image flattened to 1 x P by the caller)
    DCX_CFG(1, 4, 2, 2, 1, 256, 1, 0, DCX_EPI_RAW),
    // RefineNet head: 3x3 + BN + ReLU + 1x1 -> 1 channel + tile arg-max
    DCX_CFG(1, 4, 2, 2, 8, 32, 3, 0, DCX_EPI_HEAT),
    // ---- 2-D Winograd family: every 3x3 + BN + ReLU (+ pool) layer with cin >= 32 that does not read through an up-sampling
    DCX_W2HCFG(8, 16, 0, 1),
    DCX_W2HCFG(8, 16, 1, 1),
    DCX_W2HCFG(6, 20, 0, 1),  // 3 x 10 tiles: 30x40 / 60x80 maps without padding, RefineNet's 18/20-pixel maps at 83-90 %
    DCX_W2HCFG(6, 20, 1, 1),
    DCX_W2HCFG(8, 8, 0, 2),   // two whole 8x8 maps (RefineNet conv3a / conv3b) per work item
    DCX_W2HCFG_S(8, 8, 0),    // small launches (bs = 1: one frame, ~16 patches)
    DCX_W2HCFG_S(8, 8, 1),
    DCX_W2HSCFG(0, 4), DCX_W2HSCFG(0, 2), DCX_W2HSCFG(0, 1),      // positions split over waves: single-round launches only
    DCX_W2HSCFG(1, 4), DCX_W2HSCFG(1, 2), DCX_W2HSCFG(1, 1),
    // ---- phase x Winograd family: every 3x3 + BN + ReLU layer (cin >= 32) that reads a x2 up-sampled input
    DCX_W2PCFG(8, 16, DCX_EPI_BNRELU, 1),
    DCX_W2PCFG(8, 16, DCX_EPI_HEAT, 1),
    DCX_W2PCFG(8, 8, DCX_EPI_BNRELU, 2),   // two whole 8x8 low-resolution maps (RefineNet conv4a) per work item
    DCX_W2PSCFG(4), DCX_W2PSCFG(2), DCX_W2PSCFG(1),      // positions split over waves: single-round launches only
};

// Deterministic mode (dcx_set_deterministic / DCX_DETERMINISTIC=1): every layer runs on the direct family -- each multiply-add
// of the layers as written is executed (the A/B reference for the Winograd families, ~0.5x the throughput).
int g_deterministic = -1;
int dcx_deterministic_enabled() {
    if (g_deterministic < 0) { const char* e = getenv("DCX_DETERMINISTIC"); g_deterministic = (e && atoi(e)) ? 1 : 0; }
    return g_deterministic;
}

// The kernel family of a layer: a function of the LAYER (cin, cout, kernel size, pooling, epilogue, up-sampled input) and of the
// process-wide mode only -- never of the batch size, the number of live patches or the CU count.  infer_batch(frames)[b] is
// therefore bit-identical to infer_image(frames[b]) (reference semantics: inference.py:32-70 is per frame).
int family_of(int cin, int cout_pad, int ks, int pool, int epi, int ups_phase) {
    if (dcx_deterministic_enabled() || ks != 3 || epi == DCX_EPI_RAW || cin < 2 * DCX_CCH) return FAM_DIRECT;
    if (ups_phase && !pool && (epi == DCX_EPI_BNRELU || (epi == DCX_EPI_HEAT && cout_pad == 64))) return FAM_W2P;
    if (epi == DCX_EPI_BNRELU) return FAM_W2H;
    return FAM_DIRECT;
}

// Tile choice inside the family = argmin of a small cost model (cycles on the busiest CU), calibrated with the in-kernel probes
// (tools/unit_probe.py): rounds = ceil(items / #CU) work items run one after the other on a CU (co-resident workgroups share its
// matrix pipes), each costing its MFMA cycles -- padded pixels included, so tile utilisation is accounted for -- plus per-unit
// and per-item overheads.  This may depend on the launch size: it changes the speed, not the bits.
const CfgEntry* pick(int n, int cin, int ho, int wo, int cout_pad, int ks, int pool, int epi, int allow_group = 1, int ups = 0) {
    const CfgEntry* best = nullptr;
    double best_cost = 0.0;
    const int n_cu = dcx_device_cu_count();
    auto can_run = [&](const CfgEntry& c) {
        if (c.ks != ks || c.pool != pool || c.epi != epi || cout_pad % c.cout_tile != 0) return false;
        if (c.fam == FAM_W2P) return ups == 1 && cin >= 2 * DCX_CCH && (epi != DCX_EPI_HEAT || cout_pad == 64) &&
                                     (c.group == 1 || (ho <= 2 * c.th && wo <= 2 * c.tw));
        if (c.fam == FAM_W2H) return cin >= 2 * DCX_CCH && (c.group == 1 || (allow_group && ups == 0 && ho <= c.th && wo <= c.tw));
        return true;
    };
    // test hook: DCX_FORCE_CFG=<kernel name> restricts the choice to that instantiation when it can run the layer (read on
    // every call so that a test can walk through all instantiations); otherwise family_of + the cost model decide
    const char* force = getenv("DCX_FORCE_CFG");
    if (force != nullptr && force[0] != 0) {
        for (const CfgEntry& c : kCfgs)
            if (strcmp(c.name, force) == 0 && can_run(c)) return &c;
    }
    const int fam = family_of(cin, cout_pad, ks, pool, epi, ups);
    const int units = cin / DCX_CCH;
    for (const CfgEntry& c : kCfgs) {
        if (c.fam != fam || !can_run(c)) continue;
        double cost;
        if (c.fam == FAM_W2H) {
            const long ht = (long)((ho + c.th - 1) / c.th) * ((wo + c.tw - 1) / c.tw);
            const long items = (long)((n + c.group - 1) / c.group) * (cout_pad / c.cout_tile) * ht;
            // per item: 128 MFMAs of 32 cycles per unit + the transform's serial VALU + half an epilogue; stalls are hidden by the
            // co-resident workgroup (measured, tools/unit_probe.py); the 6x20 tile's transform reads are 2-way bank-conflicted
            double item_cost = (double)units * (64 * 64.0 + (c.tw == 20 ? 700.0 : 560.0)) + 2600.0;
            if (c.acc_tiles == 4)        // 16-tile items: half the serial chain per item, but twice the weight loads and 1.56x the halo per
                                         // MFMA -- they win where a launch cannot fill the chip (bs = 1: the launch takes ONE item's chain)
                                         // and where 32-tile items quantise badly (conv4a / conv4b at bs = 32: 768 items on 512 slots;
                                         // measured 60.0 vs 63.5 us), nowhere else (conv3a 121 vs 107 us, heads 229 vs 196: tools/layer_table.py)
                item_cost = (double)units * (32 * 64.0 + 600.0) + 2000.0;
            if (c.acc_tiles == 1) {      // dcx_conv_wino2hs.h: a wave's chain is 4 positions; the cout_tile / 16 waves of a SIMD interleave theirs.
                                         // Only where every item gets a CU of its own (a launch = one item's chain): the smallest cout tile
                                         // that still fits wins.  DCX_W2HS=<mask> restricts the choice (0: none; A/B runs)
                static int w2hs = -1;     // bit mask of the cout-group counts in the choice (1 | 2 | 4)
                if (w2hs < 0) { const char* e = getenv("DCX_W2HS"); w2hs = e ? atoi(e) : 7; }
                static int w2hs_rounds = -1;      // DCX_W2HS_ROUNDS (experiments): launches of up to this many items per CU may use these kernels
                if (w2hs_rounds < 0) { const char* e = getenv("DCX_W2HS_ROUNDS"); w2hs_rounds = e ? atoi(e) : 1; }
                if (!(w2hs & (c.cout_tile / 16)) || items > (long)w2hs_rounds * n_cu) continue;
                item_cost = (double)units * (8 * 64.0 * (c.cout_tile / 16) + 400.0) + 1200.0;
                if (w2hs_rounds > 1 && items > n_cu) item_cost *= 0.8;      // (experiment: prefer them wherever they are allowed)
            }
            cost = (double)((items + n_cu - 1) / n_cu) * item_cost * (1.0 + 1e-6 * (double)items);   // ties: fewer work items
        } else if (c.fam == FAM_W2P) {   // 9 x 8 MFMAs of 32 cycles per unit; tiles are low-resolution, x4 phases
            const long wt = (long)((ho / 2 + c.th - 1) / c.th) * ((wo / 2 + c.tw - 1) / c.tw);
            const long items = (long)((n + c.group - 1) / c.group) * (cout_pad / c.cout_tile) * wt * 4;
            double item_cost = (double)units * (72 * 32.0 + 500.0) + 2400.0;
            if (c.acc_tiles == 1) {      // dcx_conv_wino2ps.h: a wave's chain is 3 positions; 3 cout_tile / 16 waves share four SIMDs.  As for
                                         // dcx_conv_wino2hs.h: only where every item gets a CU of its own (DCX_W2PS=<mask of cout groups>, 0 = off)
                static int w2ps = -1;
                if (w2ps < 0) { const char* e = getenv("DCX_W2PS"); w2ps = e ? atoi(e) : 7; }
                if (!(w2ps & (c.cout_tile / 16)) || items > n_cu) continue;
                item_cost = (double)units * (12 * 32.0 * ((3 * (c.cout_tile / 16) + 3) / 4) + 400.0) + 1200.0;
            }
            cost = (double)((items + n_cu - 1) / n_cu) * item_cost * (1.0 + 1e-6 * (double)items);
        } else {
            const long tiles = (long)((ho + c.th - 1) / c.th) * ((wo + c.tw - 1) / c.tw);
            const long items = (long)n * (cout_pad / c.cout_tile) * tiles;
            const int steps = ks * ks * (DCX_CCH / 8);
            // per-unit overhead: barrier + first LDS wait + bookkeeping (520); epilogue ~40 (60 pooled, 35 in-lane) cycles per register
            const double item_cost = (double)units * steps * (4 * c.acc_tiles) * 64.0 + units * 520.0
                                   + c.acc_tiles * 16 * (c.inlane ? 35.0 : (pool ? 60.0 : 40.0));
            cost = (double)((items + n_cu - 1) / n_cu) * item_cost;
            if (c.cout_tile == 64 && c.cap == 256) cost *= 0.999;   // deterministic tie-break towards the A layout
        }
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
int g_prof_every = 1;     // record every g_prof_every-th matching launch (each bracket costs ~11 us of idle GPU)
long g_prof_seen = 0;
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
extern "C" int dcx_profile_enabled(void) { return g_prof ? 1 : 0; }
extern "C" int dcx_profile_filter(int kernel_id) { g_prof_filter = kernel_id; return 0; }
extern "C" int dcx_profile_sample(int every) { g_prof_every = every > 1 ? every : 1; g_prof_seen = 0; return 0; }
extern "C" int dcx_profile_count(void) { return (int)g_recs.size(); }
extern "C" const char* dcx_profile_kernel_name(int id) {
    const int n = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));
    switch (id) {      // the pipeline's launches outside the convolution families (bracketed by dcx_api.hip: dcx_prof_begin / _end)
        case DCX_PROF_CONV1A: return "dcx_conv1_kernel (detector conv1a)";
        case DCX_PROF_PATCHES: return "dcx_conv1_patches_kernel (RefineNet conv1a + patch gather)";
        case DCX_PROF_TAIL: return "dcx_tail_kernel (1x1 heads + arg-max + compaction)";
        case DCX_PROF_FINALIZE: return "dcx_refine_finalize_kernel";
        default: break;
    }
    return id >= 0 && id < n ? kCfgs[id].name : "?";
}

// hipEvent bracket around one non-conv launch of the pipeline (same record list as the convolutions; flops = 0).  Only while
// profiling is on and unfiltered (a filter selects one convolution instantiation).  -> token for dcx_prof_end, -1 = not recording
int dcx_prof_begin(int kernel_id, int n, hipStream_t stream) {
    if (!g_prof || g_prof_filter >= 0) return -1;
    ProfRec r;
    r.kernel_id = kernel_id; r.n = n; r.limited = 0; r.flops_per_image = 0.0;
    r.e0 = prof_event();
    r.e1 = prof_event();
    r.slot = (int)g_recs.size();
    if (!r.e0 || !r.e1 || hipEventRecord(r.e0, stream) != hipSuccess) return -1;
    g_recs.push_back(r);
    return r.slot;
}
int dcx_prof_end(int token, hipStream_t stream) {
    if (token < 0 || token >= (int)g_recs.size()) return 0;
    DCX_CHECK_HIP(hipEventRecord(g_recs[token].e1, stream));
    return 0;
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

// ---- per-XCD item-range weights -------------------------------------------------------------------------------------------------
// The eight XCDs of one MI355X do not run this load at the same speed (3-6 % between the fastest and the slowest, a property of the
// chip: profiles/experiments/r06_xcd_weights.txt), and a launch ends with its slowest XCD.  The XCD-aware walk therefore gives XCD x
// the items [total * cum[x], total * cum[x + 1]) with cum from these weights (equal until dcx_calibrate_xcd / dcx_set_xcd_weights).
namespace {
// Device tables of cumulative shares, [kXcdTables][16] ints per device, used round robin: a table is written ONCE (before any launch
// can read it) and never modified afterwards, so every workgroup of a launch -- and every replay of a hipGraph captured with it -- sees
// one partition even while the host installs new weights; a slot is only reused after kXcdTables further weight changes.
constexpr int kXcdTables = 256;
int* g_xcd_tab[DCX_MAX_DEVICES] = {};
int g_xcd_next[DCX_MAX_DEVICES] = {};
const int* g_xcd_cur[DCX_MAX_DEVICES] = {};
float g_xcd_w[DCX_MAX_DEVICES][8];
int xcd_set(int dev, const float* w8) {
    if (g_xcd_tab[dev] == nullptr)
        DCX_CHECK_HIP(hipMalloc((void**)&g_xcd_tab[dev], sizeof(int) * 16 * kXcdTables));
    double sum = 0.0;
    for (int x = 0; x < 8; ++x) { g_xcd_w[dev][x] = w8 ? w8[x] : 1.0f; sum += g_xcd_w[dev][x]; }
    double acc = 0.0;
    int cum[16] = {};
    for (int x = 0; x < 8; ++x) {
        g_xcd_w[dev][x] = (float)(g_xcd_w[dev][x] / sum);
        acc += g_xcd_w[dev][x];
        cum[x + 1] = x == 7 ? 65536 : (int)(acc * 65536.0 + 0.5);
    }
    int* slot = g_xcd_tab[dev] + 16 * (g_xcd_next[dev]++ % kXcdTables);
    DCX_CHECK_HIP(hipMemcpy(slot, cum, sizeof(cum), hipMemcpyHostToDevice));      // synchronous: complete before any launch that uses it
    g_xcd_cur[dev] = slot;
    return 0;
}
}  // namespace

void dcx_fill_xcd_cum(DcxConvArgs& a) {
    const int dev = dcx_current_device();
    if (g_xcd_cur[dev] == nullptr) (void)xcd_set(dev, nullptr);      // first launch on this device (never under graph capture: the
                                                                      // callers warm up eagerly first)
    a.xcd_cum = g_xcd_cur[dev];
}

// w8: relative speeds of XCD 0..7 of the CURRENT device (any positive scale; NULL = equal).  Weights further than 25 % from equal
// are refused (a measurement gone wrong must not starve an XCD).  Speed only: the set of work items and their bits do not change.
extern "C" int dcx_set_xcd_weights(const float* w8) {
    if (w8 != nullptr) {
        double sum = 0.0;
        for (int x = 0; x < 8; ++x) { if (!(w8[x] > 0.f)) return DCX_E_ARG; sum += w8[x]; }
        for (int x = 0; x < 8; ++x) { const double r = 8.0 * w8[x] / sum; if (r < 0.75 || r > 1.25) return DCX_E_ARG; }
    }
    return xcd_set(dcx_current_device(), w8);
}
extern "C" int dcx_get_xcd_weights(float* w8) {
    if (!w8) return DCX_E_ARG;
    const int dev = dcx_current_device();
    if (g_xcd_cur[dev] == nullptr) { const int rc = xcd_set(dev, nullptr); if (rc) return rc; }
    for (int x = 0; x < 8; ++x) w8[x] = 8.0f * g_xcd_w[dev][x];      // 1.0 = an equal share
    return 0;
}

namespace {
// pseudo-random operand values for the calibration launches: the matrix pipe's power draw -- and with it the per-XCD clock behaviour
// under test -- depends on the operands (constant operands showed a quarter of the spread random ones do)
__global__ void dcx_fill_hash_kernel(float* __restrict__ p, size_t n, float lo, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = lo + scale * (float)(h >> 8) * (1.0f / 16777216.0f);
    }
}
}  // namespace

// Measures the XCDs of the current device and sets their weights: `rounds` launches of the dominant kernel's own instantiation on a
// synthetic conv1b-sized layer (64 -> 64, 3x3, pooled, 32 frames of 320x240: 19,200 work items, 37.5 per workgroup, ~0.8 ms each),
// every workgroup adding its end time to its XCD's sum; speed of XCD x = its item share / its mean workgroup duration, averaged
// over the rounds after two warm-up launches.  Synchronous (set-up code: call it once the GPU is warm, outside any timed region and
// outside hipGraph capture); ~0.8 GB of scratch is allocated and freed.  w8_out (nullable): the weights set (1.0 = equal share).
extern "C" int dcx_calibrate_xcd(int rounds, float* w8_out, void* stream) {
    if (rounds < 1 || rounds > 64) return DCX_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int n = 32, c = 64, h = 240, w = 320;
    const size_t in_elems = (size_t)n * c * h * w, out_elems = in_elems / 4, w_elems = (size_t)16 * c * c;
    float *d_in = nullptr, *d_out = nullptr, *d_w = nullptr, *d_ab = nullptr;
    unsigned long long* d_stat = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_w); (void)hipFree(d_ab); (void)hipFree(d_stat); };
    hipError_t e = hipMalloc((void**)&d_in, in_elems * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, out_elems * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_w, w_elems * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_ab, 3 * c * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_stat, 17 * 8);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)d_ab, 0x3f800000, 3 * c, s);         // alpha = beta = bias = 1
    if (e != hipSuccess) { cleanup(); return (int)e; }
    hipLaunchKernelGGL(dcx_fill_hash_kernel, dim3(4096), dim3(256), 0, s, d_in, in_elems, -1.0f, 2.0f);
    hipLaunchKernelGGL(dcx_fill_hash_kernel, dim3(64), dim3(256), 0, s, d_w, w_elems, -0.05f, 0.1f);
    if ((e = hipGetLastError()) != hipSuccess) { cleanup(); return (int)e; }
    DcxConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = d_in; a.w = d_w; a.w_wino2 = d_w; a.bias = d_ab; a.alpha = d_ab + c; a.beta = d_ab + 2 * c; a.out = d_out;
    a.n = n; a.in_cq_total = c / 4; a.cin = c; a.hin = h; a.win = w; a.pad = 1; a.ho = h; a.wo = w;
    a.out_cq_total = c / 4; a.cout_pad = c; a.cout_quads = c / 4; a.cout_real = c;
    const int dev = dcx_current_device();
    float keep[8];
    for (int x = 0; x < 8; ++x) keep[x] = g_xcd_cur[dev] != nullptr ? g_xcd_w[dev][x] : 0.125f;
    (void)xcd_set(dev, nullptr);                             // measure with equal shares
    double speed[8] = {};
    int rc = 0, used = 0;
    for (int r = 0; r < rounds + 2 && rc == 0; ++r) {
        if ((e = hipMemsetAsync(d_stat, 0, 17 * 8, s)) != hipSuccess) { rc = (int)e; break; }
        a.xcd_stat = r >= 2 ? d_stat : nullptr;
        rc = dcx_launch_conv_mfma(a, 3, 1, DCX_EPI_BNRELU, s);
        if (rc != 0 || r < 2) continue;
        unsigned long long hst[17];
        if ((e = hipMemcpyAsync(hst, d_stat, sizeof(hst), hipMemcpyDeviceToHost, s)) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess) { rc = (int)e; break; }
        bool ok = hst[8] != 0;
        double dur[8];
        for (int x = 0; x < 8 && ok; ++x) {
            ok = hst[9 + x] > 0;
            if (ok) { dur[x] = (double)hst[x] / (double)hst[9 + x] - (double)hst[8]; ok = dur[x] > 0.0; }
        }
        if (!ok) continue;                                   // e.g. the flat walk (DCX_XCD_WALK=0): nothing to weight
        for (int x = 0; x < 8; ++x) speed[x] += 1.0 / dur[x];
        ++used;
    }
    (void)hipStreamSynchronize(s);
    cleanup();
    if (rc == 0 && used > 0) {
        float w8[8];
        for (int x = 0; x < 8; ++x) w8[x] = (float)(speed[x] / used);
        rc = dcx_set_xcd_weights(w8);
    } else if (rc == 0) {
        rc = DCX_E_SHAPE;
    }
    if (rc != 0) (void)xcd_set(dev, keep);
    if (rc == 0 && w8_out != nullptr) (void)dcx_get_xcd_weights(w8_out);
    return rc;
}

int dcx_occupancy_override() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DCX_OCC"); v = e ? atoi(e) : 0; }
    return v;
}

int dcx_conv_heat_tiles(int ho, int wo, int ups) {
    const CfgEntry* c = pick(1 << 20, 64, ho, wo, 64, 3, 0, DCX_EPI_HEAT, 1, ups);
    if (c == nullptr) return 0;
    if (c->fam == FAM_W2P) return 4 * ((ho / 2 + c->th - 1) / c->th) * ((wo / 2 + c->tw - 1) / c->tw);
    return ((ho + c->th - 1) / c->th) * ((wo + c->tw - 1) / c->tw);
}

int dcx_launch_conv_mfma(DcxConvArgs a, int ks, int pool, int epi, hipStream_t stream) {
    if (a.in == nullptr || a.w == nullptr || a.bias == nullptr) return DCX_E_ARG;
    if (epi != DCX_EPI_HEAT && a.out == nullptr) return DCX_E_ARG;
    if (epi != DCX_EPI_RAW && (a.alpha == nullptr || a.beta == nullptr)) return DCX_E_ARG;
    if (pool && (a.ho < 2 || a.wo < 2)) return DCX_E_SHAPE;      // odd sizes are fine: the pooling floors like MaxPool2d(2,2)
    // the fused-head launch must use the tiling dcx_conv_heat_tiles() sized part_val / part_idx for
    const CfgEntry* c = pick(epi == DCX_EPI_HEAT ? (1 << 20) : (a.n_hint > 0 && a.n_hint < a.n ? a.n_hint : a.n), a.cin, a.ho, a.wo, a.cout_pad, ks, pool, epi,
                             a.ups == 0 && a.pad == 1,    // grouped tiles: same-size convolutions read without up-sampling
                             (a.ups == 1 && a.pad == 1 && ks == 3 && a.w_ups2w != nullptr) ? 1 : 0);
    if (c == nullptr) return DCX_E_SHAPE;
    if (!g_prof || (g_prof_filter >= 0 && g_prof_filter != (int)(c - kCfgs))) return c->launch(a, stream);
    if (g_prof_every > 1 && (g_prof_seen++ % g_prof_every) != 0) return c->launch(a, stream);
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
