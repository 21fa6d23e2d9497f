// Weight-STREAMING launcher of pws_kernel (pws_kernel.h, KCH = 3) for the deep pointwise contractions of layer 4 (x3d_fine.py:100-105: conv3
// 432 -> 192 forward, the data gradient of conv1 192 -> 432 = a contraction over its 432 output channels, conv5's data gradient).
//
// Why: the split-bf16 product wants every activation converted ONCE (prologue + 3-term split = ~12 VALU instructions per element, and these
// kernels are instruction-issue bound), i.e. a wave that owns ALL output rows of its positions -- pws_kernel's structure.  Its weight images
// are resident in LDS, and three images of 192 x 432 are 498 KB: at K = 432 a slab is 32 rows, six slabs convert every activation six times
// and the shape stayed on the fp32-MFMA pw_deep_kernel (64-row weight image, 52 TFLOP/s, 0.21-0.23 ms).  pwq_kernel (round 5, measured, not
// in the tree: DESIGN 4.4) kept the weights in registers instead -- three slabs of 64 rows, four k slices -- and was issue bound at 0.20 ms.
// Here the weights are split ONCE per launch into a workspace by pws_presplit_kernel (the byte image of the LDS chunk buffers, 580 KB,
// L2 resident) and stream through two LDS chunk buffers of 3 k-blocks while the waves multiply: 192 rows per slab, one conversion per element.
#include "pws_kernel.h"
#include <map>
#include <mutex>
#include <utility>

#define PWT_KCH 3                 // k-blocks per chunk = one trip of pws_kernel's operand ring
#define PWT_ROWB (PWT_KCH * 32 + 16)

// out[slab][chunk][term][BM][PWT_ROWB bytes]: term s of W[m0 + m][48 c + kk] (FWD: w is (M, K)) or W[48 c + kk][m0 + m] (DGRAD: w is (K, M)),
// zero beyond M / K.  grid (chunks, slabs, PWT_PRE_Z pieces of a chunk: one pair per thread and trip -- 9 workgroups of 18 dependent load -> store
// trips each took 14-17 us per launch); the 16 pad bytes of a row are never read
#define PWT_PRE_Z 8
template <int MODE>
__global__ __launch_bounds__(256) void pws_presplit_kernel(const float* __restrict__ w, int pitch, int M, int K, int BM, unsigned char* __restrict__ out) {
    const int c = blockIdx.x, slab = blockIdx.y, nchunks = gridDim.x, m0 = slab * BM;
    unsigned char* blob = out + ((size_t)slab * nchunks + c) * 3 * BM * PWT_ROWB;
    const int total = BM * (PWT_KCH * 8);                                  // pairs of one chunk
    for (int e = blockIdx.z * 256 + threadIdx.x; e < total; e += 256 * PWT_PRE_Z) {
        int m, kk;
        if (MODE == PW_FWD) { m = e / (PWT_KCH * 8); kk = (e - m * (PWT_KCH * 8)) * 2; }      // consecutive threads along k (w rows)
        else { kk = (e / BM) * 2; m = e - (e / BM) * BM; }                                     // consecutive threads along m (w rows)
        const int k = c * (PWT_KCH * 16) + kk;
        float v0 = 0.0f, v1 = 0.0f;
        if (m0 + m < M) {
            if (MODE == PW_FWD) {
                if (k < K) v0 = w[(long)(m0 + m) * pitch + k];
                if (k + 1 < K) v1 = w[(long)(m0 + m) * pitch + k + 1];
            } else {
                if (k < K) v0 = w[(long)k * pitch + m0 + m];
                if (k + 1 < K) v1 = w[(long)(k + 1) * pitch + m0 + m];
            }
        }
        unsigned p[3];
        pws_split<3>(v0, v1, p);
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<unsigned*>(blob + ((size_t)s * BM + m) * PWT_ROWB + kk * 2) = p[s];
    }
}

// One workspace per (device, stream), grown geometrically, never freed (a captured graph may have the address baked in); no allocation
// while the stream is being captured: the call is then declined and the fp32-MFMA kernel runs.  Launches on one stream run in order, so
// the next conv's pre-split cannot overtake this conv's contraction.  A hipGraph bakes the address in at capture time: it has to be REPLAYED ON THE
// STREAM IT WAS CAPTURED ON (cfn_hip/graph.py GraphedStep._replay does), otherwise eager work on the capture stream -- or a second graph captured
// on it and replayed elsewhere -- overwrites the split weights mid-contraction; the same holds for pwss_workspace (pwsplitw.hip).  Retired
// buffers stay allocated on purpose (a captured graph may still hold them); geometric growth bounds them by the size of the live one.
static unsigned char* pwt_workspace(size_t bytes, hipStream_t st) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<unsigned char*, size_t>> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    auto& b = bufs[std::make_pair(dev, st)];
    if (b.second >= bytes) return b.first;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    const size_t want = bytes > 2 * b.second ? bytes : 2 * b.second;
    unsigned char* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    b = {p, want};
    return p;
}

static size_t pwt_lds(int BM, int Kp) {
    const size_t red = (size_t)PWS_WAVES * 4 * (BM * 2 > 32 * 20 ? BM * 2 : 32 * 20);
    return (size_t)2 * 3 * BM * PWT_ROWB + (size_t)Kp * 16 + (size_t)BM * 8 + red;
}

template <int MODE, bool STATS, int ACT, bool TWO>
static int pwt_go_mt(const PwArgs& a, int MT, unsigned blocks, size_t lds, hipStream_t st) {
#define PWT_GO(MTV)                                                                                                        \
    do {                                                                                                                   \
        auto k = pws_kernel<MTV, 1, MODE, STATS, ACT, TWO, 3, PWT_KCH>;                                                    \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * PWS_WAVES), lds, st, a);                                             \
    } while (0)
    switch (MT) { case 3: PWT_GO(3); break; case 4: PWT_GO(4); break; case 5: PWT_GO(5); break; default: PWT_GO(6); break; }
#undef PWT_GO
    return cfn_check_launch("pwconv(split bf16, streamed weights)");
}

template <int MODE, bool STATS, bool TWO>
static int pwt_go_act(const PwArgs& a, int MT, unsigned blocks, size_t lds, hipStream_t st) {
    if constexpr (MODE == PW_DGRAD && !STATS) {
        return pwt_go_mt<MODE, STATS, CFN_ACT_NONE, TWO>(a, MT, blocks, lds, st);
    } else {
        switch (a.act) {
            case CFN_ACT_RELU: return pwt_go_mt<MODE, STATS, CFN_ACT_RELU, TWO>(a, MT, blocks, lds, st);
            case CFN_ACT_SWISH: return pwt_go_mt<MODE, STATS, CFN_ACT_SWISH, TWO>(a, MT, blocks, lds, st);
            default: return pwt_go_mt<MODE, STATS, CFN_ACT_NONE, TWO>(a, MT, blocks, lds, st);
        }
    }
}

// returns -1 when the shape is not handled.  DGRAD: stats == (ea != nullptr), i.e. with the act' epilogue
int pwt_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st) {
    static const int on = getenv("CFN_PWT") ? atoi(getenv("CFN_PWT")) : 7;               // bit 0: forward, bit 1: data gradient without act' epilogue, bit 2: with
    static const int mink = getenv("CFN_PWT_MINK") ? atoi(getenv("CFN_PWT_MINK")) : 400;
    static const int mink_epi = getenv("CFN_PWT_MINK_EPI") ? atoi(getenv("CFN_PWT_MINK_EPI")) : 160;
    if (pws_terms_now() != 6 || a.stem || a.stride != 1 || a.acc) return -1;
    if (!(on & (mode == PW_FWD ? 1 : (stats ? 4 : 2)))) return -1;
    // K >= 400: every mode (measured, 8 clips x 256 frames @7x7, against pw_deep_kernel: forward 432 -> 192 0.223 -> 0.142 ms, data gradient with
    // two staged operands 0.229 -> 0.161, 432 rows 0.50 -> 0.31, 96 rows @14x14 0.54 -> 0.37); the data gradient WITH the act' epilogue into more
    // than 256 rows also from K = 160 (layer-4 conv3: contraction over 192, 432 rows in three slabs: 0.273 -> 0.223 ms; pwk_kernel takes the
    // shapes of up to 256 rows first)
    const bool deep = a.K >= mink, epi = mode == PW_DGRAD && stats && a.K >= mink_epi && a.M > 256;
    // whole 16-byte groups per row (Q % 4 == 0) are only needed by the forward's transposed epilogue (16-byte stores); the data gradient moves 4-byte
    // elements both ways -- the coarse stream's layer 4 (65 x 7 x 7 = 3,185 positions per row) runs its data gradients here
    if (!(deep || epi) || a.M <= 32 || (mode == PW_FWD && (a.Q & 3))) return -1;
    if (a.act != CFN_ACT_NONE && a.act != CFN_ACT_RELU && a.act != CFN_ACT_SWISH) return -1;
    if ((long)a.K * a.Q * 4 >= 0x3ffffff0L || (long)a.M * a.Q * 4 >= 0x3ffffff0L) return -1;
    if (((uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)(a.src2 ? a.src2 : a.src) | (uintptr_t)(a.ex ? a.ex : a.src)) & (mode == PW_FWD ? 15 : 3)) return -1;
    PwArgs b = a;
    b.Kpad = (a.K + 47) / 48 * 48;
    if (mode == PW_DGRAD && !stats) b.act = CFN_ACT_NONE;
    // rows per slab: 6 row tiles (two chunk buffers of 7 do not fit LDS); as few slabs as possible, then as even as possible
    int slabs = cfn_cdiv(a.M, 32 * 6);
    int MT = cfn_cdiv(cfn_cdiv(a.M, slabs), 32);
    if (MT < 3) MT = 3;
    slabs = cfn_cdiv(a.M, 32 * MT);
    const int BM = 32 * MT, nchunks = b.Kpad / (16 * PWT_KCH);
    const size_t lds = pwt_lds(BM, b.Kpad);
    if (lds > 160 * 1024) return -1;
    unsigned char* ws = pwt_workspace((size_t)slabs * nchunks * 3 * BM * PWT_ROWB, st);
    if (!ws) return -1;
    if (mode == PW_FWD) hipLaunchKernelGGL(pws_presplit_kernel<PW_FWD>, dim3(nchunks, slabs, PWT_PRE_Z), dim3(256), 0, st, a.w, a.Cin, a.M, a.K, BM, ws);
    else hipLaunchKernelGGL(pws_presplit_kernel<PW_DGRAD>, dim3(nchunks, slabs, PWT_PRE_Z), dim3(256), 0, st, a.w, a.Cin, a.M, a.K, BM, ws);
    { const int rc = cfn_check_launch("pwconv(split bf16, streamed weights) pre-split"); if (rc) return rc; }
    b.w = reinterpret_cast<const float*>(ws);
    b.mtiles = slabs;
    b.kres = PWT_ROWB;
    const int ntiles = cfn_cdiv(a.Q, 32);
    const long groups = (long)a.N * slabs;
    static const int wg_env = getenv("CFN_PWT_WGS") ? atoi(getenv("CFN_PWT_WGS")) : 0;
    long wgs = (wg_env > 0 ? wg_env : 256) / groups;                        // one 8-wave workgroup per CU, never more than one round of the chip
    const long maxw = cfn_cdiv(ntiles, PWS_WAVES);
    if (wgs > maxw) wgs = maxw;
    if (wgs < 1) wgs = 1;
    b.nstrips = (int)wgs;
    const unsigned blocks = (unsigned)(groups * wgs);
    if (mode == PW_FWD) return stats ? pwt_go_act<PW_FWD, true, false>(b, MT, blocks, lds, st) : pwt_go_act<PW_FWD, false, false>(b, MT, blocks, lds, st);
    if (a.src2) return stats ? pwt_go_act<PW_DGRAD, true, true>(b, MT, blocks, lds, st) : pwt_go_act<PW_DGRAD, false, true>(b, MT, blocks, lds, st);
    return stats ? pwt_go_act<PW_DGRAD, true, false>(b, MT, blocks, lds, st) : pwt_go_act<PW_DGRAD, false, false>(b, MT, blocks, lds, st);
}
