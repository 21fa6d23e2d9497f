/* cfn_hip.h -- C ABI of libcfn_hip.so: the MI355X (gfx950) kernels behind the Coarse-Fine hot path.
 *
 * The reference (kkahatapitiya/Coarse-Fine-Networks) is pure Python/PyTorch: it has no FFI of its
 * own, every heavy op is an ATen call.  Each entry point below therefore names the reference CALL
 * SITE (file:line under the reference tree) whose ATen op(s) it replaces; INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers to DEVICE memory + sizes; no torch types.  Tensors are dense NCDHW fp32 (the *_bf16 entry points
 *     at the end of this file take bf16 activations as `unsigned short*`).
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); kernels are enqueued on it and
 *     the call returns immediately; no allocation, no synchronisation, no host<->device copy inside
 *     (graph-capture safe).
 *   - return 0 on success, otherwise an error code (1 bad argument, 2 launch failure, 3 unsupported
 *     shape); cfn_last_error() returns a thread-local message.  Nothing ever aborts.
 *   - "prologue": many ops read their input x through  a = act(A[n,c]*x + B[n,c])  where A/B (float,
 *     N*C, may be NULL = identity) carry the folded SubBatchNorm3d scale/shift (+ BN affine + SE gate)
 *     and act is 0 none / 1 ReLU / 2 Swish / 3 sigmoid (sigmoid: cfn_affine_act_* only).  This is how SubBatchNorm3d.forward (x3d_fine.py:51-62),
 *     relu_ (:151), Swish (:74-86) and the SE multiply (:163) disappear into the next conv.
 *   - "stats": fp64 accumulators per (n,c) that the caller zero-fills; kernels atomically add
 *     per-workgroup partial sums.  sum/sumsq of a conv output are what batch_norm (train) and the SE
 *     average pool (x3d_fine.py:158) need; gsum/gsumsq are the gradients flowing back into them.
 */
#ifndef CFN_HIP_H
#define CFN_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* cfn_last_error(void);
const char* cfn_version(void);
int cfn_device_info(int* cus, int* lds_per_cu, char* name, int name_len);

/* opt-in HIP-event timing per kernel family (bench.py roofline leg). family: 0 dwconv fwd, 1 dwconv
 * bwd-data, 2 pwconv fwd, 3 pwconv bwd-data, 4 gridpool, 5 elementwise, 6 stem, 7 fusion, 8 pwconv bwd-weight,
 * 9 dwconv bwd-weight, 10 dense (Grid Pool saliency) conv fwd, 11 gridpool bwd (family 4 = Grid Pool resampler fwd).  collect() sums and
 * clears: total device ms between the bracketing events, launches, algorithmic bytes. */
int cfn_prof_enable(int family, int on);
int cfn_prof_collect(int family, double* total_ms, long* launches, double* total_bytes);

/* ---- depthwise 3x3x3, pad 1, stride (1,s,s): conv3x3x3 x3d_fine.py:89-97 used at :117 / x3d_coarse.py:87-95 ----
 * fwd   y[N,C,T,Ho,Wo] = dwconv(act(A x + B)); sum/sumsq[N*C] += sum y, sum y^2 (NULL to skip)
 * bwd_data   g' = gy + gsum[n,c] + 2 y gsumsq[n,c];  da = dwconv^T(g');  gx = da*act'(A x+B)*A;
 *            gA[N*C] += sum da*act'*x, gB += sum da*act'   (autograd of F.conv3d + batch_norm + relu_)
 * bwd_weight gw[C*27] (fp64, zero-filled by caller) += sum g' * act(A x + B)[tap] */
int cfn_dwconv3d_fwd(const float* x, const double* A, const double* B, int act, const float* w, float* y, double* sum,
                     double* sumsq, int N, int C, int T, int Hi, int Wi, int stride, void* stream);
int cfn_dwconv3d_bwd_data(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                          const float* x, const double* A, const double* B, int act, float* gx, double* gA, double* gB,
                          int N, int C, int T, int Hi, int Wi, int stride, void* stream);
int cfn_dwconv3d_bwd_weight(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* x,
                            const double* A, const double* B, int act, double* gw, int N, int C, int T, int Hi, int Wi,
                            int stride, void* stream);
/* stride-1 data AND weight gradient in one pass over gy, y, x (4 tensor passes instead of 7): same outputs as the two
 * calls above.  Returns -1 without launching when the geometry is not handled (small planes): call the two kernels. */
int cfn_dwconv3d_bwd_fused(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                           const float* x, const double* A, const double* B, int act, float* gx, double* gA, double* gB,
                           double* gw, int N, int C, int T, int H, int W, void* stream);
/* the same for the STRIDE-2 conv (H, W: input size; x is read once instead of twice).  Returns -1 without launching when the
 * geometry is not handled (handled: 112 -> 56, 56 -> 28, 28 -> 14 with a none / ReLU prologue): call the two kernels. */
int cfn_dwconv3d_bwd_fused_s2(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                              const float* x, const double* A, const double* B, int act, float* gx, double* gA, double* gB,
                              double* gw, int N, int C, int T, int H, int W, void* stream);

/* ---- depthwise 5x1x1, pad (2,0,0): conv1_t x3d_fine.py:216-222 / x3d_coarse.py:502-508 ; plane = H*W ---- */
int cfn_dwconv_t5_fwd(const float* x, const float* w, float* y, double* sum, double* sumsq, int N, int C, int T,
                      long plane, void* stream);
int cfn_dwconv_t5_bwd_data(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                           float* gx, int N, int C, int T, long plane, void* stream);
int cfn_dwconv_t5_bwd_weight(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* x,
                             double* gw, int N, int C, int T, long plane, void* stream);
/* data AND weight gradient of conv1_t in one pass over gy, y, x (4 tensor passes instead of 6).  Returns -1 without launching
 * when the plane is not a whole number of float4s: call the two entry points above. */
int cfn_dwconv_t5_bwd_fused(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                            const float* x, float* gx, double* gw, int N, int C, int T, long plane, void* stream);

/* ---- pointwise 1x1x1, spatial stride s in {1,2}: conv1x1x1 x3d_fine.py:100-105 (conv1 :115, conv3 :119,
 * shortcut :284-287), conv5 :245-250, fc1 :256; fp32 MFMA (v_mfma_f32_32x32x2_f32).  w is (Cout,Cin).
 * bwd_data with stride 2 writes only the strided positions: caller zero-fills gx.
 * bwd_weight accumulates into gw (Cout,Cin) fp64, zero-filled by the caller (one atomic per element per workgroup). ---- */
int cfn_pwconv_fwd(const float* x, const double* A, const double* B, int act, const float* w, float* y, double* sum,
                   double* sumsq, int N, int Cin, int Cout, int T, int Hi, int Wi, int stride, void* stream);
int cfn_pwconv_bwd_data(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                        const float* x, const double* A, const double* B, int act, float* gx, double* gA, double* gB, int N,
                        int Cin, int Cout, int T, int Hi, int Wi, int stride, void* stream);
/* same, plus `acc`: compact gradient (N,Cin,T,ceil(Hi/acc_stride),ceil(Wi/acc_stride)) of a spatially strided shortcut conv
 * over the same input (x3d_fine.py:284-287), added on its lattice before the act' epilogue (requires stride == 1);
 * and `gscale` (N,Cout) fp64, may be NULL: gy is the UNSCALED gradient of a block tail (cfn_bn_add_relu_bwd_g) and stands
 * for gscale[n,c] * gy, i.e. g' = gscale*gy + gsum + 2*y*gsumsq */
int cfn_pwconv_bwd_data_acc(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                        const float* x, const double* A, const double* B, int act, float* gx, double* gA, double* gB, int N,
                        int Cin, int Cout, int T, int Hi, int Wi, int stride, const float* acc, int acc_stride,
                        const double* gscale, void* stream);
int cfn_pwconv_bwd_weight(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* x,
                          const double* A, const double* B, int act, double* gw, int N, int Cin, int Cout, int T, int Hi,
                          int Wi, int stride, const double* gscale, void* stream);

/* Arithmetic of the fp32 pointwise contractions with >= 48 channels on both sides (layers 2-4; conv1x1x1 x3d_fine.py:100-105):
 * 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32), 3 / 6 = split-bf16: fp32 tensors, each operand split into 2 / 3 bf16 terms on load,
 * 3 / 6 v_mfma_f32_32x32x16_bf16 per k-block with fp32 accumulation (relative error of a product <= 3*2^-18 / 3*2^-27).
 * Default 6 (env CFN_PW_SPLIT).  -1 only queries.  Returns the previous setting.  Host-side, no stream. */
int cfn_pw_split_terms(int terms);

/* Deterministic mode ("Threading / streams": "deterministic reductions preferred for parity tests (atomics order) -- offer a
 * deterministic mode"; the reference's own reductions are PyTorch's, train_fine.py:199-226 runs them in whatever mode torch is in).
 * on = 1: every cross-workgroup fp64 accumulation of the library (BN statistics, coefficient / weight gradients: ~100 sites) is recorded as an
 * (address, addend) pair instead of being added atomically; at the end of each entry point's launches the records are sorted by (address, addend
 * bits) and committed in that order, so results do not depend on the order in which workgroups finish: two identical calls give identical bits by
 * construction.  Costs a device synchronisation and a sort per entry point (parity runs, not benchmarks); not usable during hipGraph capture.
 * on = 0: fp64 atomics (default; env CFN_DETERMINISTIC=1 switches the mode on when the library is loaded by cfn_hip).  -1 only queries.
 * Returns the previous setting, < 0 on error.  Per process, bound to the device that is current at the call.  Host-side, no stream. */
int cfn_deterministic(int on);

/* Data AND weight gradient of a stride-1 pointwise conv with few channels in ONE pass (Cin, Cout <= 64 and not both > 32:
 * X3D layer 1, where the backward is HBM bound): gy, y, x leave HBM once for both products.  Same arguments and results
 * as cfn_pwconv_bwd_data_acc (stride 1) + cfn_pwconv_bwd_weight; x is always required.  Returns -1 WITHOUT launching
 * when the shape / alignment is not handled (callers then use the two separate entry points). */
int cfn_pwconv_bwd_fused(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                         const float* x, const double* A, const double* B, int act, float* gx, double* gA, double* gB,
                         double* gw, int N, int Cin, int Cout, int T, int Hi, int Wi, const float* acc, int acc_stride,
                         const double* gscale, void* stream);

/* ---- stem 1x3x3 stride (1,2,2) pad (0,1,1) dense conv: conv1_s x3d_fine.py:210-215 (im2col view on MFMA);
 * gw is fp64 (Cout, Cimg*9), zero-filled by caller.  The clip needs no gradient. ---- */
int cfn_stem_conv_fwd(const float* x, const float* w, float* y, int N, int Cimg, int Cout, int T, int Hi, int Wi,
                      void* stream);
int cfn_stem_conv_bwd_weight(const float* gy, const float* x, double* gw, int N, int Cimg, int Cout, int T, int Hi, int Wi,
                             void* stream);

/* ---- SubBatchNorm3d statistics -> (A,B) prologue, fused with the SE gate: SubBatchNorm3d.forward x3d_fine.py:51-62
 * (split groups, shared affine), nn.BatchNorm3d running-stat update, SE branch x3d_fine.py:157-163.
 * training: s,q (N,C) fp64 sums over `count` positions; run_mean/run_var = split_bn buffers (S*C) updated in place,
 * nbt = num_batches_tracked.  eval: run_mean/run_var = bn buffers (C).  Wd>0 enables SE (fc1 (Wd,C), fc2 (C,Wd),
 * pool_count = positions of the SE average).  Saved tensors (mean,rstd (S,C) fp64; A0,B0,gate,pooled (N,C); hbuf (N,Wd))
 * feed cfn_bn_fold_bwd, which returns gs,gq (N,C) fp64 (eval mode with SE: gs alone -- the gate still depends on sum(y) --,
 * gq NULL) and the parameter gradients (all overwritten). ---- */
int cfn_bn_fold_fwd(const double* s, const double* q, const float* gamma, const float* beta, float* run_mean, float* run_var,
                    long* nbt, int training, int N, int C, int S, double count, double eps, double momentum, const float* w1,
                    const float* b1, const float* w2, const float* b2, int Wd, double pool_count, double* A, double* B,
                    double* mean, double* rstd, float* A0, float* B0, float* gate, float* hbuf, float* pooled, void* stream);
int cfn_bn_fold_bwd(const double* gA, const double* gB, const double* s, const float* gamma, const double* mean,
                    const double* rstd, const float* A0, const float* B0, const float* gate, const float* hbuf,
                    const float* pooled, const float* w1, const float* w2, int training, int N, int C, int S, int Wd,
                    double count, double pool_count, double* gs, double* gq, float* ggamma, float* gbeta, float* gw1,
                    float* gb1, float* gw2, float* gb2, double* tA, double* tB, void* stream);

/* ---- dense 3-D convolution as implicit GEMM (no im2col buffer): Grid Pool saliency convs x3d_coarse.py:362-366,
 * :379-381 (and the stem, which is the geom {1,3,3, 1,2,2, 0,1,1} case).  geom = int[9] {kT,kH,kW, sT,sH,sW, pT,pH,pW}
 * in HOST memory.  w (Cout, Cin*kT*kH*kW); no bias (a bias is algebraically folded into the next prologue / the
 * statistics by the caller).  Prologue act: none or relu; zero padding is applied after the prologue. ---- */
int cfn_conv3d_dense_fwd(const float* x, const double* A, const double* B, int act, const float* w, float* y, double* sum,
                         double* sumsq, int N, int Cin, int Cout, int T, int Hi, int Wi, const int* geom, void* stream);
int cfn_conv3d_dense_bwd_data(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                              const float* x, const double* A, const double* B, int act, float* gx, double* gA, double* gB,
                              int N, int Cin, int Cout, int T, int Hi, int Wi, const int* geom, void* stream);
int cfn_conv3d_dense_bwd_weight(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* x,
                                const double* A, const double* B, int act, double* gw, int N, int Cin, int Cout, int T, int Hi,
                                int Wi, const int* geom, void* stream);

/* ---- Gaussian temporal alignment: Gaussian.forward x3d_coarse.py:256-286 (called at :651 / :657).
 * meta (B,4) int64 [start, frames, nf, step] (only columns 0 and 3 are read), mask (B,Tf), gx (B*crops,K) CDF knots or NULL
 * (tl = arange(K), the non-grid modes); tx = coarse clip length.  crops > 1 = validation-time multi-crop: row r = b*crops + j
 * starts at meta[b,0] + meta[b,3]*j (:264-266).  GX (B*crops,Tf,K).  bwd: ggx (B*crops,K) = d loss / d gx. ---- */
int cfn_gauss_align_fwd(const long* meta, const float* mask, const float* gx, double tx, double ratio, float* GX, int B, int crops,
                        int Tf, int K, void* stream);
int cfn_gauss_align_bwd(const float* gGX, const long* meta, const float* mask, const float* gx, double tx, double ratio,
                        float* ggx, int B, int crops, int Tf, int K, void* stream);

/* ---- Multi-stage Fusion temporal-alignment gather: RewightLayer.forward x3d_coarse.py:209-223 at the fine
 * features' native resolution, with the attention sigmoid (:219), the mask multiply (:213-216) and the multi-crop
 * repeat (:209-211) folded in.  x (B,C,Tf,P) fine features, at_raw (B,Tf,P) attention logits (at = sigmoid(at_raw +
 * at_bias[0]); at_bias may be NULL), GX (B*crops,Tf,K), mask (B,Tf);  row r = b*crops + j:
 *   z[r,c,k,p] = sum_t x[b] at[b] GX[r] mask[b] / (sum_t at GX mask + 1e-6), den (B*crops,K,P) saved for the backward.
 * bwd: gx (B,C,Tf,P), gat (B,Tf,P) = d loss / d at_raw, gGX (B*crops,Tf,K) (each may be NULL);
 * dw (B*crops,Tf,K,P) is scratch. ---- */
int cfn_fusion_gather_fwd(const float* x, const float* at_raw, const float* at_bias, const float* GX, const float* mask, float* z,
                          float* den, int B, int crops, int C, int Tf, int K, int P, void* stream);
int cfn_fusion_gather_bwd(const float* gz, const float* z, const float* den, const float* x, const float* at_raw,
                          const float* at_bias, const float* GX, const float* mask, float* gx, float* gat, float* gGX, float* dw,
                          int B, int crops, int C, int Tf, int K, int P, void* stream);

/* ---- block tail  out = relu(A y + B + (Ar res + Br)) : bn3 + (downsample bn) + `out += residual` + relu,
 * x3d_fine.py:167-173.  Ar/Br NULL = identity shortcut.  vol = T*H*W, NC = N*C.  bwd: gout2 (may be NULL) is a second
 * upstream gradient of the same output (the block output feeds the next conv1 AND the next residual); it is added to
 * gout on the fly. ---- */
int cfn_bn_add_relu_fwd(const float* y, const double* A, const double* B, const float* res, const double* Ar, const double* Br,
                        float* out, int* mask, long NC, long vol, void* stream);
int cfn_bn_add_relu_bwd(const float* gout, const float* gout2, const float* out, const float* y, const double* A, const float* res,
                        const double* Ar, float* gy, float* gres, double* gA, double* gB, double* gAr, long NC, long vol,
                        void* stream);
/* The tail's backward as ONE output tensor: g = (gout [+ gout2]) * (out > 0) serves as d/dy (the conv3 backward applies A
 * through `gscale`) and as d/dres (identity shortcut: exact; conv shortcut: `gscale` = Ar) -- 4 tensor passes instead of
 * 6-7.  The ReLU mask is either `out` or the bit mask cfn_bn_add_relu_fwd wrote (`mask`, cfn_bn_add_relu_mask_words(NC,
 * vol) 32-bit words, 0 = shape has no mask; bit layout private to the two kernels): exactly one of the two is given.
 * gA += sum g*y, gB += sum g, gAr += sum g*res (gAr may be NULL). */
long cfn_bn_add_relu_mask_words(long NC, long vol);
int cfn_bn_add_relu_bwd_g(const float* gout, const float* gout2, const float* out, const int* mask, const float* y,
                          const float* res, float* g, double* gA, double* gB, double* gAr, long NC, long vol, void* stream);

/* ---- materialised prologue out = act(A x + B) (SubBatchNorm3d.forward on its own, x3d_fine.py:51-62) and
 * per-(n,c) sum / sumsq of a tensor (batch statistics of an arbitrary input) ---- */
int cfn_affine_act_fwd(const float* x, const double* A, const double* B, int act, float* out, long NC, long vol, void* stream);
int cfn_affine_act_bwd(const float* gout, const float* x, const double* A, const double* B, int act, float* gx, double* gA,
                       double* gB, long NC, long vol, void* stream);
int cfn_channel_stats(const float* x, double* sum, double* sumsq, long NC, long vol, void* stream);

/* ---- adaptive spatial mean of act(A x + B) to (OH,OW): adaptive_avg_pool3d((None,1,1)) x3d_fine.py:255/366 and
 * ((None,7,7)) :345-363 (ATen window rule: [floor(o*S/O), ceil((o+1)*S/O)) ) ---- */
int cfn_pool_hw_fwd(const float* x, const double* A, const double* B, int act, float* out, long NC, int T, int H, int W, int OH,
                    int OW, void* stream);
int cfn_pool_hw_bwd(const float* gout, const float* x, const double* A, const double* B, int act, float* gx, double* gA,
                    double* gB, long NC, int T, int H, int W, int OH, int OW, void* stream);

/* ---- FiLM x*m + c with (H/f x W/f)-block-constant m, c: `x = x * m2 + c2` x3d_coarse.py:663-679 after the
 * fusion branch was evaluated at its native 7x7 ---- */
int cfn_film_fwd(const float* x, const float* m, const float* c, float* out, long NC, int T, int H, int W, int f,
                 void* stream);
int cfn_film_bwd(const float* g, const float* x, const float* m, float* gx, float* gm, float* gc, long NC, int T, int H,
                 int W, int f, void* stream);

/* ---- Grid Pool / Unpool resampler: 5-D F.grid_sample(align_corners=True) of x3d_coarse.py:403 / :447 as a
 * 2-tap lerp along t at i_t = ((2(cdf-0.5)+1)/2)(Tin-1).  i0 = floor(i_t) is bit-exact w.r.t. ATen.
 * x (B,C,Tin,P), cdf (B,K) -> out (B,C,K,P).  bwd: gx (B,C,Tin,P) fully written, gcdf (B,K) fp64 zero-filled
 * by the caller; either may be NULL. ---- */
int cfn_grid_time_index(const float* cdf, int n, int Tin, int* i0, float* w1, void* stream);
/* saliency logits g (B,Kin) (+ bias[0], may be NULL) -> CDF knots (B,Kin+1): x3d_coarse.py:384-392
 * (1 - sigmoid(g/2), normalise, cumsum in fp64, leading 0).  bwd: gg (B,Kin) from gcdf (B,Kin+1). */
int cfn_grid_cdf_fwd(const float* g, const float* bias, float* cdf, int B, int Kin, void* stream);
int cfn_grid_cdf_bwd(const float* gcdf, const float* g, const float* bias, float* gg, int B, int Kin, void* stream);
int cfn_time_sample_fwd(const float* x, const float* cdf, float* out, int B, int C, int Tin, int K, long P, void* stream);
int cfn_time_sample_bwd(const float* g, const float* x, const float* cdf, float* gx, double* gcdf, int B, int C, int Tin,
                        int K, long P, void* stream);

/* ---- fixed temporal pooling t_pool = 'avg' (mode 0) | 'max' (mode 1): nn.AvgPool3d / nn.MaxPool3d((R,1,1), stride (R,1,1))
 * x3d_coarse.py:489-492 / :640-643.  x (BC,Tin,P) -> out (BC,floor(Tin/R),P); bwd writes all of gx (max: first maximal frame) ---- */
int cfn_time_pool_fwd(const float* x, float* out, int mode, long BC, int Tin, int R, long P, void* stream);
int cfn_time_pool_bwd(const float* g, const float* x, float* gx, int mode, long BC, int Tin, int R, long P, void* stream);

/* ---- Interp1d.forward interp1d.py:8-147: x,y (B or 1 rows, N), xnew (B or 1 rows, Pq) -> ynew (B,Pq), ind int64
 * (searchsorted-left - 1, clamped to [0,N-2]).  *row flags: 1 = one row per batch entry, 0 = shared row.
 * bwd overwrites gx/gy/gq (any may be NULL) in gather form: every sum has a fixed order (reproducible). ---- */
int cfn_interp1d_fwd(const float* x, const float* y, const float* xnew, float* ynew, long* ind, int B, int N, int Pq,
                     int xrow, int yrow, int qrow, void* stream);
int cfn_interp1d_bwd(const float* g, const float* x, const float* y, const float* xnew, const long* ind, float* gx, float* gy,
                     float* gq, int B, int N, int Pq, int xrow, int yrow, int qrow, void* stream);

/* ---- linear resize along t: F.interpolate(mode='linear') x3d_coarse.py:725, the t axis of :449 (align_corners=1) and
 * the loss upsampling train_coarse_fineFEAT.py:226 / train_fine.py:204 (align_corners=0, half-pixel centres).
 * x (BC,Kin,P) -> out (BC,Lout,P) ---- */
int cfn_time_resize_fwd(const float* x, float* out, long BC, int Kin, int Lout, long P, int align_corners, void* stream);
int cfn_time_resize_bwd(const float* g, float* gx, long BC, int Kin, int Lout, long P, int align_corners, void* stream);

/* =====================================================================================================================
 * bf16 activation path (BASELINE.json configs[1] "X3D-M fwd+bwd bf16"; the reference itself is fp32 only: these entry
 * points are the same call sites with activations and activation gradients stored as bf16 in HBM).
 *   - `unsigned short*` = bf16 tensor (round to nearest even on store); weights, prologue coefficients, statistics and
 *     every reduction stay fp32 / fp64 exactly as in the fp32 entry points; all arithmetic is fp32.
 *   - pointwise contractions run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; BN statistics are taken over the
 *     ROUNDED outputs (what the consumer reads back).
 *   - the stem conv (fp32 clip in) keeps an fp32 output; conv1_t (cfn_dwconv_t5_*_bf16) reads fp32 and writes bf16, and
 *     returns an fp32 input gradient.  Pooled head / feature-tower tensors are fp32 (cfn_pool_hw_*_bf16 leaves the
 *     bf16 domain).
 *   - shape limits: pointwise needs T*H*W even (fwd / bwd_data) and a multiple of 8 (bwd_weight); a spatially strided
 *     pointwise conv = cfn_subsample_hw_bf16 + the stride-1 contraction on the compact tensor.
 * ===================================================================================================================== */
/* conv1x1x1 x3d_fine.py:100-105 (stride 1; Q = T*H*W positions per (n, channel) row) */
int cfn_pwconv_fwd_bf16(const unsigned short* x, const double* A, const double* B, int act, const float* w, unsigned short* y,
                        double* sum, double* sumsq, int N, int Cin, int Cout, long Q, void* stream);
/* as cfn_pwconv_bwd_data_acc: acc = compact (N,Cin,T,ceil(H/s),ceil(W/s)) gradient of a strided second consumer, gscale (N,Cout) */
int cfn_pwconv_bwd_data_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                             const float* w, const unsigned short* x, const double* A, const double* B, int act,
                             unsigned short* gx, double* gA, double* gB, int N, int Cin, int Cout, int T, int H, int W,
                             const unsigned short* acc, int acc_stride, const double* gscale, void* stream);
int cfn_pwconv_bwd_weight_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                               const unsigned short* x, const double* A, const double* B, int act, double* gw, int N, int Cin,
                               int Cout, long Q, const double* gscale, void* stream);
/* x[..., ::s, ::s] of (planes = N*C*T, H, W): the gather in front of a strided shortcut conv (x3d_fine.py:284-287) */
int cfn_subsample_hw_bf16(const unsigned short* x, unsigned short* out, long planes, int H, int W, int s, void* stream);

/* conv3x3x3 depthwise x3d_fine.py:89-97: cfn_dwconv3d_* with bf16 tensors (same kernels compiled for 2-byte elements) */
int cfn_dwconv3d_fwd_bf16(const unsigned short* x, const double* A, const double* B, int act, const float* w, unsigned short* y,
                          double* sum, double* sumsq, int N, int C, int T, int Hi, int Wi, int stride, void* stream);
int cfn_dwconv3d_bwd_data_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                               const float* w, const unsigned short* x, const double* A, const double* B, int act,
                               unsigned short* gx, double* gA, double* gB, int N, int C, int T, int Hi, int Wi, int stride,
                               void* stream);
int cfn_dwconv3d_bwd_weight_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                 const unsigned short* x, const double* A, const double* B, int act, double* gw, int N, int C,
                                 int T, int Hi, int Wi, int stride, void* stream);
int cfn_dwconv3d_bwd_fused_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                const float* w, const unsigned short* x, const double* A, const double* B, int act,
                                unsigned short* gx, double* gA, double* gB, double* gw, int N, int C, int T, int H, int W,
                                void* stream);
int cfn_dwconv3d_bwd_fused_s2_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                   const float* w, const unsigned short* x, const double* A, const double* B, int act,
                                   unsigned short* gx, double* gA, double* gB, double* gw, int N, int C, int T, int H, int W,
                                   void* stream);

/* conv1_t depthwise 5x1x1 x3d_fine.py:216-222: x / gx fp32, y / gy bf16 */
int cfn_dwconv_t5_fwd_bf16(const float* x, const float* w, unsigned short* y, double* sum, double* sumsq, int N, int C, int T,
                           long plane, void* stream);
int cfn_dwconv_t5_bwd_data_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                const float* w, float* gx, int N, int C, int T, long plane, void* stream);
int cfn_dwconv_t5_bwd_weight_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                  const float* x, double* gw, int N, int C, int T, long plane, void* stream);
int cfn_dwconv_t5_bwd_fused_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                 const float* w, const float* x, float* gx, double* gw, int N, int C, int T, long plane,
                                 void* stream);

/* block tail x3d_fine.py:167-173 (cfn_bn_add_relu_fwd / _bwd_g) and spatial pooling :255,:345-366 (bf16 in, fp32 pooled) */
long cfn_bn_add_relu_mask_words_bf16(long NC, long vol);
int cfn_bn_add_relu_fwd_bf16(const unsigned short* y, const double* A, const double* B, const unsigned short* res,
                             const double* Ar, const double* Br, unsigned short* out, int* mask, long NC, long vol, void* stream);
int cfn_bn_add_relu_bwd_g_bf16(const unsigned short* gout, const unsigned short* gout2, const unsigned short* out, const int* mask,
                               const unsigned short* y, const unsigned short* res, unsigned short* g, double* gA, double* gB,
                               double* gAr, long NC, long vol, void* stream);
int cfn_pool_hw_fwd_bf16(const unsigned short* x, const double* A, const double* B, int act, float* out, long NC, int T, int H,
                         int W, int OH, int OW, void* stream);
int cfn_pool_hw_bwd_bf16(const float* gout, const unsigned short* x, const double* A, const double* B, int act,
                         unsigned short* gx, double* gA, double* gB, long NC, int T, int H, int W, int OH, int OW, void* stream);

/* =====================================================================================================================
 * fp16 activation path (BASELINE configs[4]: "fp16 MFMA pointwise"; the reference itself is fp32 only, x3d_fine.py:100-105).
 * The entry points of the bf16 path above with IEEE-half tensors: `unsigned short*` = fp16 tensor (round to nearest even on
 * store), pointwise contractions on v_mfma_f32_32x32x16_f16 with fp32 accumulation, everything else exactly as for bf16
 * (csrc/h16.h: the same sources compiled for the other 2-byte element kind).  fp16 has 3 more mantissa bits than bf16 and 5 exponent
 * bits: activation GRADIENTS need a loss scale (train_joint.py LOSS_SCALE; weight gradients accumulate in fp64 and are unscaled there).
 * The weight gradient keeps the direct-operand kernel (the LDS-staged one of pwsplitw.hip serves fp32 and bf16 tensors).
 * ===================================================================================================================== */
/* conv1x1x1 x3d_fine.py:100-105 (stride 1; Q = T*H*W positions per (n, channel) row) */
int cfn_pwconv_fwd_f16(const unsigned short* x, const double* A, const double* B, int act, const float* w, unsigned short* y,
                        double* sum, double* sumsq, int N, int Cin, int Cout, long Q, void* stream);
/* as cfn_pwconv_bwd_data_acc: acc = compact (N,Cin,T,ceil(H/s),ceil(W/s)) gradient of a strided second consumer, gscale (N,Cout) */
int cfn_pwconv_bwd_data_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                             const float* w, const unsigned short* x, const double* A, const double* B, int act,
                             unsigned short* gx, double* gA, double* gB, int N, int Cin, int Cout, int T, int H, int W,
                             const unsigned short* acc, int acc_stride, const double* gscale, void* stream);
int cfn_pwconv_bwd_weight_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                               const unsigned short* x, const double* A, const double* B, int act, double* gw, int N, int Cin,
                               int Cout, long Q, const double* gscale, void* stream);
/* x[..., ::s, ::s] of (planes = N*C*T, H, W): the gather in front of a strided shortcut conv (x3d_fine.py:284-287) */
int cfn_subsample_hw_f16(const unsigned short* x, unsigned short* out, long planes, int H, int W, int s, void* stream);

/* conv3x3x3 depthwise x3d_fine.py:89-97: cfn_dwconv3d_* with fp16 tensors (same kernels compiled for 2-byte elements) */
int cfn_dwconv3d_fwd_f16(const unsigned short* x, const double* A, const double* B, int act, const float* w, unsigned short* y,
                          double* sum, double* sumsq, int N, int C, int T, int Hi, int Wi, int stride, void* stream);
int cfn_dwconv3d_bwd_data_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                               const float* w, const unsigned short* x, const double* A, const double* B, int act,
                               unsigned short* gx, double* gA, double* gB, int N, int C, int T, int Hi, int Wi, int stride,
                               void* stream);
int cfn_dwconv3d_bwd_weight_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                 const unsigned short* x, const double* A, const double* B, int act, double* gw, int N, int C,
                                 int T, int Hi, int Wi, int stride, void* stream);
int cfn_dwconv3d_bwd_fused_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                const float* w, const unsigned short* x, const double* A, const double* B, int act,
                                unsigned short* gx, double* gA, double* gB, double* gw, int N, int C, int T, int H, int W,
                                void* stream);
int cfn_dwconv3d_bwd_fused_s2_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                   const float* w, const unsigned short* x, const double* A, const double* B, int act,
                                   unsigned short* gx, double* gA, double* gB, double* gw, int N, int C, int T, int H, int W,
                                   void* stream);

/* conv1_t depthwise 5x1x1 x3d_fine.py:216-222: x / gx fp32, y / gy fp16 */
int cfn_dwconv_t5_fwd_f16(const float* x, const float* w, unsigned short* y, double* sum, double* sumsq, int N, int C, int T,
                           long plane, void* stream);
int cfn_dwconv_t5_bwd_data_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                const float* w, float* gx, int N, int C, int T, long plane, void* stream);
int cfn_dwconv_t5_bwd_weight_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                  const float* x, double* gw, int N, int C, int T, long plane, void* stream);
int cfn_dwconv_t5_bwd_fused_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                 const float* w, const float* x, float* gx, double* gw, int N, int C, int T, long plane,
                                 void* stream);

/* block tail x3d_fine.py:167-173 (cfn_bn_add_relu_fwd / _bwd_g) and spatial pooling :255,:345-366 (fp16 in, fp32 pooled) */
long cfn_bn_add_relu_mask_words_f16(long NC, long vol);
int cfn_bn_add_relu_fwd_f16(const unsigned short* y, const double* A, const double* B, const unsigned short* res,
                             const double* Ar, const double* Br, unsigned short* out, int* mask, long NC, long vol, void* stream);
int cfn_bn_add_relu_bwd_g_f16(const unsigned short* gout, const unsigned short* gout2, const unsigned short* out, const int* mask,
                               const unsigned short* y, const unsigned short* res, unsigned short* g, double* gA, double* gB,
                               double* gAr, long NC, long vol, void* stream);
int cfn_pool_hw_fwd_f16(const unsigned short* x, const double* A, const double* B, int act, float* out, long NC, int T, int H,
                         int W, int OH, int OW, void* stream);
int cfn_pool_hw_bwd_f16(const float* gout, const unsigned short* x, const double* A, const double* B, int act,
                         unsigned short* gx, double* gA, double* gB, long NC, int T, int H, int W, int OH, int OW, void* stream);

#ifdef __cplusplus
}
#endif
#endif
