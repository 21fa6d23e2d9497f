// Host side of the split-bf16 pointwise contractions with RESIDENT weight images (kernel: pws_kernel.h; the weight-streaming launcher for
// the deep contractions of layer 4 is pwstream.hip).
#include "pws_kernel.h"

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static int g_pws_terms = -1;      // 0 = off (fp32 MFMA kernels), 3 / 6 = MFMAs per k-block
int pws_terms_now() {
    if (g_pws_terms < 0) {
        const char* e = getenv("CFN_PW_SPLIT");
        g_pws_terms = e ? atoi(e) : 6;
        if (g_pws_terms != 0 && g_pws_terms != 3 && g_pws_terms != 6) g_pws_terms = 6;
    }
    return g_pws_terms;
}
extern "C" int cfn_pw_split_terms(int terms) {
    const int prev = pws_terms_now();
    if (terms == 0 || terms == 3 || terms == 6) g_pws_terms = terms;
    else if (terms != -1) return cfn_fail(CFN_ERR_ARG, "cfn_pw_split_terms: terms must be 0 (fp32 MFMA), 3 or 6 (or -1 to query)"), -2;
    return prev;
}

static size_t pws_lds(int BM, int Kp, int rowb, int NS) {
    const size_t red = (size_t)PWS_WAVES * 4 * (BM * 2 > 32 * 20 ? BM * 2 : 32 * 20);
    return (size_t)NS * BM * rowb + (size_t)Kp * 16 + (size_t)BM * 8 + red;
}

template <int MODE, bool STATS, int ACT, bool TWO, int NS>
static int pws_go_mt(const PwArgs& a, int MT, int NP, unsigned blocks, size_t lds, hipStream_t st) {
#define PWS_GO(MTV, NPV)                                                                                                   \
    do {                                                                                                                   \
        auto k = pws_kernel<MTV, NPV, MODE, STATS, ACT, TWO, NS>;                                                          \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * PWS_WAVES), lds, st, a);                                             \
    } while (0)
    if (NP == 1) {      // 32-position tiles: the many-row slabs
        if constexpr (MODE == PW_DGRAD && TWO) {
            switch (MT) { case 3: PWS_GO(3, 1); break; case 4: PWS_GO(4, 1); break; case 5: PWS_GO(5, 1); break; default: PWS_GO(6, 1); break; }
        } else {
            switch (MT) { case 3: PWS_GO(3, 1); break; case 4: PWS_GO(4, 1); break; case 5: PWS_GO(5, 1); break; case 6: PWS_GO(6, 1); break; default: PWS_GO(7, 1); break; }
        }
    } else if constexpr (MODE == PW_DGRAD && TWO) {
        if (MT == 1) PWS_GO(1, 2); else PWS_GO(2, 2);
    } else if constexpr (MODE == PW_DGRAD) {
        switch (MT) { case 1: PWS_GO(1, 2); break; case 2: PWS_GO(2, 2); break; default: PWS_GO(3, 2); break; }
    } else {
        switch (MT) { case 1: PWS_GO(1, 2); break; case 2: PWS_GO(2, 2); break; default: PWS_GO(3, 2); break; }
    }
#undef PWS_GO
    return cfn_check_launch("pwconv(split bf16)");
}

template <int MODE, bool STATS, bool TWO, int NS>
static int pws_go_act(const PwArgs& a, int MT, int NP, unsigned blocks, size_t lds, hipStream_t st) {
    if constexpr (MODE == PW_DGRAD && !STATS) {
        return pws_go_mt<MODE, STATS, CFN_ACT_NONE, TWO, NS>(a, MT, NP, blocks, lds, st);
    } else {
        switch (a.act) {
            case CFN_ACT_RELU: return pws_go_mt<MODE, STATS, CFN_ACT_RELU, TWO, NS>(a, MT, NP, blocks, lds, st);
            case CFN_ACT_SWISH: return pws_go_mt<MODE, STATS, CFN_ACT_SWISH, TWO, NS>(a, MT, NP, blocks, lds, st);
            default: return pws_go_mt<MODE, STATS, CFN_ACT_NONE, TWO, NS>(a, MT, NP, blocks, lds, st);
        }
    }
}

template <int NS>
static int pws_go(const PwArgs& a, int mode, bool stats, int MT, int NP, unsigned blocks, size_t lds, hipStream_t st) {
    if (mode == PW_FWD) return stats ? pws_go_act<PW_FWD, true, false, NS>(a, MT, NP, blocks, lds, st) : pws_go_act<PW_FWD, false, false, NS>(a, MT, NP, blocks, lds, st);
    if (a.src2) return stats ? pws_go_act<PW_DGRAD, true, true, NS>(a, MT, NP, blocks, lds, st) : pws_go_act<PW_DGRAD, false, true, NS>(a, MT, NP, blocks, lds, st);
    return stats ? pws_go_act<PW_DGRAD, true, false, NS>(a, MT, NP, blocks, lds, st) : pws_go_act<PW_DGRAD, false, false, NS>(a, MT, NP, blocks, lds, st);
}

// returns -1 when the shape is not handled (caller falls through to the fp32-MFMA kernels), otherwise the launch status.
// FWD prologue: pa == nullptr means A = 1, B = 0 (act still applies, as in the fp32-MFMA kernels); DGRAD: stats == (ea != nullptr)
int pws_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st) {
    const int terms = pws_terms_now();
    if (terms == 0) return -1;
    // the compact shortcut gradient `acc` (first block of a stage: 3 calls per step) stays on the fp32-MFMA kernel: its lattice
    // loads cost the many-row variants 90+ registers
    if (a.stem || a.stride != 1 || a.K < 48 || a.M <= 32 || (a.Q & 3) || a.acc) return -1;      // Q % 4: whole 16-byte groups
    if (a.act != CFN_ACT_NONE && a.act != CFN_ACT_RELU && a.act != CFN_ACT_SWISH) return -1;
    // The split costs VALU work per CONTRACTION-side element (prologue + 9 instructions per pair for three terms) and saves matrix
    // time per product: measured (8 clips, T = 256, profiles/r03_microbench_b8.txt) it wins up to K = 108 (layer 2 both convs,
    // layer 3 conv1 forward 0.18 vs 0.21 ms, conv3 data gradient 0.24 vs 0.33 ms) and loses from K = 216 on (layer 3 conv3
    // forward 0.22-0.26 vs 0.20 ms): those stay on the fp32-MFMA kernel
    static const int maxk_env = getenv("CFN_PWS_MAXK") ? atoi(getenv("CFN_PWS_MAXK")) : 128;
    if (a.K > maxk_env) return -1;
    if ((long)a.K * a.Q * 4 >= 0x3ffffff0L || (long)a.M * a.Q * 4 >= 0x3ffffff0L) return -1;
    if (((uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)(a.src2 ? a.src2 : a.src) | (uintptr_t)(a.ex ? a.ex : a.src)) & 15) return -1;
    const int NS = terms == 3 ? 2 : 3;
    PwArgs b = a;
    b.Kpad = (a.K + 47) / 48 * 48;
    b.kres = b.Kpad * 2 + 16;
    if (((b.kres / 16) & 1) == 0) b.kres += 16;                          // odd number of 16-byte slots per row
    if (mode == PW_DGRAD && !stats) b.act = CFN_ACT_NONE;
    // rows per slab, limited by registers (accumulators + the 3-set operand ring; measured with -Rpass-analysis: the next size
    // spills 50-230 bytes per lane).  64-position tiles (NP = 2): 96 rows forward, 96 / 64 backward (one / two staged tensors).
    // 32-position tiles (NP = 1): 224 rows, 192 backward with two staged tensors.  The NS weight images of a slab must fit
    // LDS.  Fewer slabs win (every extra slab re-reads the activations), then the wider tile.
    auto fit = [&](int mt) { while (mt > 0 && pws_lds(32 * mt, b.Kpad, b.kres, NS) > 160 * 1024) --mt; return mt; };
    const int mt2 = fit(mode == PW_FWD ? 3 : (a.src2 ? 2 : 3)), mt1 = fit(mode == PW_DGRAD && a.src2 ? 6 : 7);
    if (mt2 < 1) return -1;
    static const int np_env = getenv("CFN_PWS_NP") ? atoi(getenv("CFN_PWS_NP")) : 0;
    int NP = 2, mt_max = mt2;
    if (mt1 >= 3 && (cfn_cdiv(a.M, 32 * mt1) < cfn_cdiv(a.M, 32 * mt2) || np_env == 1) && np_env != 2) { NP = 1; mt_max = mt1; }
    int slabs = cfn_cdiv(a.M, 32 * mt_max);
    const int per = cfn_cdiv(a.M, slabs);
    int MT = cfn_cdiv(per, 32);
    if (NP == 1 && MT < 3) MT = 3;
    slabs = cfn_cdiv(a.M, 32 * MT);
    // three or more slabs (X3D layer 4: the weight images of 432 x 192 do not fit LDS in fewer) re-read every activation that
    // often: measured slower than the fp32-MFMA kernel with its 4-byte weight image (0.19-0.31 vs 0.20-0.24 ms) -- declined
    static const int slab_env = getenv("CFN_PWS_MAXSLABS") ? atoi(getenv("CFN_PWS_MAXSLABS")) : 2;
    if (slabs > slab_env) return -1;
    b.mtiles = slabs;
    const size_t lds = pws_lds(32 * MT, b.Kpad, b.kres, NS);
    const int ntiles = cfn_cdiv(a.Q, 32 * NP);
    const long groups = (long)a.N * slabs;
    static const int wg_env = getenv("CFN_PWS_WGS") ? atoi(getenv("CFN_PWS_WGS")) : 0;
    long wgs = cfn_cdiv(wg_env > 0 ? wg_env : 256, groups);              // one 8-wave workgroup per CU (up to 256 VGPRs)
    const long maxw = cfn_cdiv(ntiles, PWS_WAVES);
    if (wgs > maxw) wgs = maxw;
    if (wgs < 1) wgs = 1;
    b.nstrips = (int)wgs;
    const unsigned blocks = (unsigned)(groups * wgs);
    return NS == 2 ? pws_go<2>(b, mode, stats, MT, NP, blocks, lds, st) : pws_go<3>(b, mode, stats, MT, NP, blocks, lds, st);
}
