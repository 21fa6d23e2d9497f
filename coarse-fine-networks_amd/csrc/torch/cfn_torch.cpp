// TORCH_LIBRARY registration of the hot path's operators INSIDE a shared library ("a torch extension .so (C++/HIP) registering ops in
// a private namespace via TORCH_LIBRARY, each with fwd + bwd"; north_star: "hand-written HIP C++ exposed as custom torch ops").
//
// libcfn_torch.so = this file (plain C++, compiled with g++ against the torch headers) linked against libcfn_hip.so: every operator below is a thin
// at::Tensor front of the C ABI of include/cfn_hip.h -- allocate outputs with the caching allocator, take the CURRENT HIP stream of the tensors'
// device, call the entry point, surface its error string as a c10::Error (-> RuntimeError).  Schemas are the ones cfn_hip/torchlib.py declares for
// the same names; when this library is present torchlib.py does not define those operators in Python, it only attaches the fake (meta)
// implementations and the autograd formulas to the native ones, so torch.compile / torch.export / opcheck see native dispatcher operators.
//   cfn::dwconv3d / dwconv3d_backward     depthwise 3x3x3          x3d_fine.py:89-97     (fp32, bf16, fp16 tensors)
//   cfn::pwconv / pwconv_backward         pointwise 1x1x1          x3d_fine.py:100-105   (fp32 tensors)
//   cfn::time_sample / _backward          Grid Pool resampler      x3d_coarse.py:393-403
// round 6: the remaining 13 pairs of the section-8(b) set -- dwconv_t5, stem_conv, conv3d_dense, bn_fold, bn_add_relu, affine_act, pool_hw, interp1d,
// grid_cdf, gauss_align, fusion_gather, film, time_resize (each with its _backward) -- see the second half of this file.
#include <ATen/ATen.h>
// (a ROCm build of torch presents its HIP devices as "cuda": the masquerading guard / stream classes are the ones that accept them)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <algorithm>
#include <tuple>
#include <vector>

#include "../../../include/cfn_hip.h"

namespace {

using at::Tensor;
using OptT = const c10::optional<Tensor>&;

inline void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream(); }
inline void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", cfn_last_error()); }
inline Tensor c64(OptT t) { return t.has_value() && t->defined() ? t->to(at::kDouble).contiguous() : Tensor(); }
inline const double* dptr(const Tensor& t) { return t.defined() ? t.data_ptr<double>() : nullptr; }
inline void check_act(const Tensor& t, const char* op) {
    TORCH_CHECK(t.is_cuda(), op, ": device tensors only (there is no CPU path)");
    TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kHalf, op, ": fp32 / bf16 / fp16 tensors, got ", t.scalar_type());
}
inline Tensor f64(at::IntArrayRef shape, const Tensor& like) { return at::zeros(shape, like.options().dtype(at::kDouble)); }
// The C ABI sees pointers and sizes only: everything a kernel will index is checked against the shapes HERE, so that a tensor of another size /
// type raises instead of being read out of bounds on the device (cfn_hip/ops.py has the same checks on the ctypes route).
inline void check_coef(OptT A, OptT B, int64_t N, int64_t C, const Tensor& x, const char* op) {
    const bool a = A.has_value() && A->defined(), b = B.has_value() && B->defined();
    TORCH_CHECK(a == b, op, ": the prologue coefficients A and B come together");
    if (!a) return;
    TORCH_CHECK(A->numel() == N * C && B->numel() == N * C, op, ": per-sample coefficients of ", N, " x ", C, " channels expected, got A ", A->sizes(), ", B ", B->sizes());
    TORCH_CHECK(A->device() == x.device() && B->device() == x.device(), op, ": A / B live on another device than x");
}
inline void check_like(const Tensor& t, const Tensor& ref, const char* op, const char* name, const char* refname) {
    TORCH_CHECK(t.sizes() == ref.sizes(), op, ": ", name, " ", t.sizes(), " does not have the shape of ", refname, " ", ref.sizes());
    TORCH_CHECK(t.scalar_type() == ref.scalar_type(), op, ": ", name, " is ", t.scalar_type(), ", ", refname, " is ", ref.scalar_type());
    TORCH_CHECK(t.device() == ref.device(), op, ": ", name, " lives on another device than ", refname);
}
inline void check_stat(const Tensor& g, int64_t N, int64_t C, const Tensor& x, const char* op, const char* name) {
    TORCH_CHECK(g.numel() == N * C && g.device() == x.device(), op, ": ", name, " must hold ", N, " x ", C, " values on x's device, got ", g.sizes());
}
inline void check_out_shape(const Tensor& y, const Tensor& x, int64_t Cout, int64_t stride, const char* op) {
    TORCH_CHECK(stride >= 1, op, ": stride ", stride);
    TORCH_CHECK(y.dim() == 5 && y.size(0) == x.size(0) && y.size(1) == Cout && y.size(2) == x.size(2) && y.size(3) == (x.size(3) - 1) / stride + 1 &&
                    y.size(4) == (x.size(4) - 1) / stride + 1,
                op, ": y ", y.sizes(), " is not the output of x ", x.sizes(), " at stride ", stride, " with ", Cout, " channels");
}

// ---- depthwise 3x3x3 --------------------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> dwconv3d(const Tensor& x_, const Tensor& w, OptT A, OptT B, int64_t act, int64_t stride) {
    check_act(x_, "cfn::dwconv3d");
    TORCH_CHECK(x_.dim() == 5, "cfn::dwconv3d: x must be (N, C, T, H, W)");
    TORCH_CHECK(stride >= 1 && w.numel() == x_.size(1) * 27 && w.device() == x_.device(), "cfn::dwconv3d: w must hold C x 27 taps on x's device, got ", w.sizes());
    check_coef(A, B, x_.size(0), x_.size(1), x_, "cfn::dwconv3d");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous();
    const int64_t N = x.size(0), C = x.size(1), T = x.size(2), H = x.size(3), W = x.size(4);
    const int64_t Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    Tensor y = at::empty({N, C, T, Ho, Wo}, x.options());
    Tensor s = f64({N, C}, x), q = f64({N, C}, x);
    const Tensor A64 = c64(A), B64 = c64(B), w2 = w.reshape({C, 27}).to(at::kFloat).contiguous();
    void* st = stream_of(x);
    int rc;
    if (x.scalar_type() == at::kFloat)
        rc = cfn_dwconv3d_fwd(x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act, w2.data_ptr<float>(), y.data_ptr<float>(), s.data_ptr<double>(),
                              q.data_ptr<double>(), (int)N, (int)C, (int)T, (int)H, (int)W, (int)stride, st);
    else if (x.scalar_type() == at::kBFloat16)
        rc = cfn_dwconv3d_fwd_bf16((const unsigned short*)x.data_ptr(), dptr(A64), dptr(B64), (int)act, w2.data_ptr<float>(), (unsigned short*)y.data_ptr(),
                                   s.data_ptr<double>(), q.data_ptr<double>(), (int)N, (int)C, (int)T, (int)H, (int)W, (int)stride, st);
    else
        rc = cfn_dwconv3d_fwd_f16((const unsigned short*)x.data_ptr(), dptr(A64), dptr(B64), (int)act, w2.data_ptr<float>(), (unsigned short*)y.data_ptr(),
                                  s.data_ptr<double>(), q.data_ptr<double>(), (int)N, (int)C, (int)T, (int)H, (int)W, (int)stride, st);
    ok(rc, "cfn_dwconv3d_fwd");
    return {y, s, q};
}

// -> (gx, gw, gA, gB); gA / gB are zeros (1 element) when there is no prologue
std::tuple<Tensor, Tensor, Tensor, Tensor> dwconv3d_backward(const Tensor& gy_, const Tensor& gs, const Tensor& gq, const Tensor& x_, const Tensor& w,
                                                             const Tensor& y_, OptT A, OptT B, int64_t act, int64_t stride) {
    check_act(x_, "cfn::dwconv3d_backward");
    TORCH_CHECK(x_.dim() == 5, "cfn::dwconv3d_backward: x must be (N, C, T, H, W)");
    TORCH_CHECK(w.numel() == x_.size(1) * 27 && w.device() == x_.device(), "cfn::dwconv3d_backward: w must hold C x 27 taps on x's device, got ", w.sizes());
    check_out_shape(y_, x_, x_.size(1), stride, "cfn::dwconv3d_backward");
    TORCH_CHECK(y_.scalar_type() == x_.scalar_type() && y_.device() == x_.device(), "cfn::dwconv3d_backward: y must have x's element type and device");
    check_like(gy_, y_, "cfn::dwconv3d_backward", "gy", "y");          // gy is reinterpreted with x's element type below
    check_stat(gs, x_.size(0), x_.size(1), x_, "cfn::dwconv3d_backward", "gs");
    check_stat(gq, x_.size(0), x_.size(1), x_, "cfn::dwconv3d_backward", "gq");
    check_coef(A, B, x_.size(0), x_.size(1), x_, "cfn::dwconv3d_backward");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), y = y_.contiguous(), gy = gy_.contiguous();
    const int64_t N = x.size(0), C = x.size(1), T = x.size(2), H = x.size(3), W = x.size(4);
    const Tensor w2 = w.reshape({C, 27}).to(at::kFloat).contiguous(), A64 = c64(A), B64 = c64(B);
    const Tensor gs64 = gs.to(at::kDouble).contiguous(), gq64 = gq.to(at::kDouble).contiguous();
    Tensor gx = at::empty_like(x), gw = f64({C, 27}, x);
    const bool pro = A64.defined();
    Tensor ab = pro ? f64({2, N, C}, x) : Tensor();
    double* a64 = pro ? ab.data_ptr<double>() : nullptr;
    double* b64 = pro ? a64 + N * C : nullptr;
    void* st = stream_of(x);
    const int n = (int)N, c = (int)C, t = (int)T, h = (int)H, wd = (int)W, a = (int)act, sd = (int)stride;
#define CFN_DW_BWD(SFX, ET)                                                                                                                      \
    do {                                                                                                                                         \
        const ET* gyp = (const ET*)gy.data_ptr(); const ET* yp = (const ET*)y.data_ptr(); const ET* xp = (const ET*)x.data_ptr(); ET* gxp = (ET*)gx.data_ptr(); \
        int rc = stride == 1 ? cfn_dwconv3d_bwd_fused##SFX(gyp, yp, dptr(gs64), dptr(gq64), w2.data_ptr<float>(), xp, dptr(A64), dptr(B64), a, gxp, a64, b64, \
                                                          gw.data_ptr<double>(), n, c, t, h, wd, st)                                             \
                             : cfn_dwconv3d_bwd_fused_s2##SFX(gyp, yp, dptr(gs64), dptr(gq64), w2.data_ptr<float>(), xp, dptr(A64), dptr(B64), a, gxp, a64, \
                                                             b64, gw.data_ptr<double>(), n, c, t, h, wd, st);                                    \
        if (rc == -1) {      /* geometry not served by the fused kernels: data and weight gradient apart */                                      \
            ok(cfn_dwconv3d_bwd_data##SFX(gyp, yp, dptr(gs64), dptr(gq64), w2.data_ptr<float>(), xp, dptr(A64), dptr(B64), a, gxp, a64, b64, n, c, t, h, wd, sd, st), \
               "cfn_dwconv3d_bwd_data");                                                                                                         \
            ok(cfn_dwconv3d_bwd_weight##SFX(gyp, yp, dptr(gs64), dptr(gq64), xp, dptr(A64), dptr(B64), a, gw.data_ptr<double>(), n, c, t, h, wd, sd, st), \
               "cfn_dwconv3d_bwd_weight");                                                                                                       \
        } else ok(rc, "cfn_dwconv3d_bwd_fused");                                                                                                 \
    } while (0)
    if (x.scalar_type() == at::kFloat) CFN_DW_BWD(, float);
    else if (x.scalar_type() == at::kBFloat16) CFN_DW_BWD(_bf16, unsigned short);
    else CFN_DW_BWD(_f16, unsigned short);
#undef CFN_DW_BWD
    Tensor gwf = gw.to(at::kFloat).view(w.sizes());
    if (!pro) return {gx, gwf, at::zeros({1}, x.options().dtype(at::kFloat)), at::zeros({1}, x.options().dtype(at::kFloat))};
    return {gx, gwf, ab[0].to(at::kFloat), ab[1].to(at::kFloat)};
}

// ---- pointwise 1x1x1 (fp32 tensors) -----------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> pwconv(const Tensor& x_, const Tensor& w, OptT A, OptT B, int64_t act, int64_t stride) {
    TORCH_CHECK(x_.is_cuda() && x_.scalar_type() == at::kFloat, "cfn::pwconv: fp32 device tensors (the bf16 / fp16 pointwise path is reached through cfn_hip.ops)");
    TORCH_CHECK(x_.dim() == 5, "cfn::pwconv: x must be (N, Cin, T, H, W)");
    TORCH_CHECK(stride >= 1 && w.dim() >= 2 && w.numel() == w.size(0) * x_.size(1) && w.device() == x_.device(), "cfn::pwconv: w must be (Cout, Cin[, 1, 1, 1]) on x's device, got ", w.sizes());
    check_coef(A, B, x_.size(0), x_.size(1), x_, "cfn::pwconv");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous();
    const int64_t N = x.size(0), Cin = x.size(1), T = x.size(2), H = x.size(3), W = x.size(4), Cout = w.size(0);
    const int64_t Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    Tensor y = at::empty({N, Cout, T, Ho, Wo}, x.options());
    Tensor s = f64({N, Cout}, x), q = f64({N, Cout}, x);
    const Tensor A64 = c64(A), B64 = c64(B), w2 = w.reshape({Cout, Cin}).to(at::kFloat).contiguous();
    ok(cfn_pwconv_fwd(x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act, w2.data_ptr<float>(), y.data_ptr<float>(), s.data_ptr<double>(), q.data_ptr<double>(),
                      (int)N, (int)Cin, (int)Cout, (int)T, (int)H, (int)W, (int)stride, stream_of(x)),
       "cfn_pwconv_fwd");
    return {y, s, q};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> pwconv_backward(const Tensor& gy_, const Tensor& gs, const Tensor& gq, const Tensor& x_, const Tensor& w,
                                                           const Tensor& y_, OptT A, OptT B, int64_t act, int64_t stride) {
    TORCH_CHECK(x_.is_cuda() && x_.scalar_type() == at::kFloat, "cfn::pwconv_backward: fp32 device tensors");
    TORCH_CHECK(x_.dim() == 5, "cfn::pwconv_backward: x must be (N, Cin, T, H, W)");
    TORCH_CHECK(w.dim() >= 2 && w.numel() == w.size(0) * x_.size(1) && w.device() == x_.device(), "cfn::pwconv_backward: w must be (Cout, Cin[, 1, 1, 1]) on x's device, got ", w.sizes());
    check_out_shape(y_, x_, w.size(0), stride, "cfn::pwconv_backward");
    TORCH_CHECK(y_.scalar_type() == x_.scalar_type() && y_.device() == x_.device(), "cfn::pwconv_backward: y must have x's element type and device");
    check_like(gy_, y_, "cfn::pwconv_backward", "gy", "y");
    check_stat(gs, x_.size(0), w.size(0), x_, "cfn::pwconv_backward", "gs");
    check_stat(gq, x_.size(0), w.size(0), x_, "cfn::pwconv_backward", "gq");
    check_coef(A, B, x_.size(0), x_.size(1), x_, "cfn::pwconv_backward");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), y = y_.contiguous(), gy = gy_.contiguous();
    const int64_t N = x.size(0), Cin = x.size(1), T = x.size(2), H = x.size(3), W = x.size(4), Cout = w.size(0);
    const Tensor w2 = w.reshape({Cout, Cin}).to(at::kFloat).contiguous(), A64 = c64(A), B64 = c64(B);
    const Tensor gs64 = gs.to(at::kDouble).contiguous(), gq64 = gq.to(at::kDouble).contiguous();
    Tensor gx = stride != 1 ? at::zeros_like(x) : at::empty_like(x), gw = f64({Cout, Cin}, x);
    const bool pro = A64.defined();
    Tensor ab = pro ? f64({2, N, Cin}, x) : Tensor();
    double* a64 = pro ? ab.data_ptr<double>() : nullptr;
    double* b64 = pro ? a64 + N * Cin : nullptr;
    void* st = stream_of(x);
    // one pass over gy, y, x where the library has a fused kernel for the shape (layers 1 and 2: pwfused.hip, pwfuseds.hip; -1 = it declines), as cfn_hip.ops does
    int fused = -1;
    if (stride == 1)
        fused = cfn_pwconv_bwd_fused(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), w2.data_ptr<float>(), x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act,
                                     gx.data_ptr<float>(), a64, b64, gw.data_ptr<double>(), (int)N, (int)Cin, (int)Cout, (int)T, (int)H, (int)W, nullptr, 1, nullptr, st);
    if (fused != -1) ok(fused, "cfn_pwconv_bwd_fused");
    else {
        ok(cfn_pwconv_bwd_data(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), w2.data_ptr<float>(), x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act,
                               gx.data_ptr<float>(), a64, b64, (int)N, (int)Cin, (int)Cout, (int)T, (int)H, (int)W, (int)stride, st),
           "cfn_pwconv_bwd_data");
        ok(cfn_pwconv_bwd_weight(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act, gw.data_ptr<double>(),
                                 (int)N, (int)Cin, (int)Cout, (int)T, (int)H, (int)W, (int)stride, nullptr, st),
           "cfn_pwconv_bwd_weight");
    }
    Tensor gwf = gw.to(at::kFloat).view(w.sizes());
    if (!pro) return {gx, gwf, at::zeros({1}, x.options()), at::zeros({1}, x.options())};
    return {gx, gwf, ab[0].to(at::kFloat), ab[1].to(at::kFloat)};
}

// ---- Grid Pool / Grid Unpool resampler --------------------------------------------------------------------------------------------------
Tensor time_sample(const Tensor& x_, const Tensor& cdf_) {
    TORCH_CHECK(x_.is_cuda() && x_.scalar_type() == at::kFloat && cdf_.scalar_type() == at::kFloat, "cfn::time_sample: fp32 device tensors");
    TORCH_CHECK(x_.dim() >= 3 && cdf_.dim() == 2 && cdf_.size(0) == x_.size(0) && cdf_.device() == x_.device(), "cfn::time_sample: x (B, C, T, ...), cdf (B, K) on one device");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), cdf = cdf_.contiguous();
    const int64_t B = x.size(0), C = x.size(1), Tin = x.size(2), K = cdf.size(1);
    std::vector<int64_t> shape = x.sizes().vec();
    shape[2] = K;
    Tensor out = at::empty(shape, x.options());
    const int64_t inner = Tin > 0 && B * C > 0 ? x.numel() / (B * C * Tin) : 1;
    ok(cfn_time_sample_fwd(x.data_ptr<float>(), cdf.data_ptr<float>(), out.data_ptr<float>(), (int)B, (int)C, (int)Tin, (int)K, (long)inner, stream_of(x)), "cfn_time_sample_fwd");
    return out;
}

std::tuple<Tensor, Tensor> time_sample_backward(const Tensor& g_, const Tensor& x_, const Tensor& cdf_) {
    TORCH_CHECK(x_.is_cuda() && x_.scalar_type() == at::kFloat && cdf_.scalar_type() == at::kFloat, "cfn::time_sample_backward: fp32 device tensors");
    TORCH_CHECK(x_.dim() >= 3 && cdf_.dim() == 2 && cdf_.size(0) == x_.size(0) && cdf_.device() == x_.device(), "cfn::time_sample_backward: x (B, C, T, ...), cdf (B, K) on one device");
    {
        std::vector<int64_t> gshape = x_.sizes().vec();
        gshape[2] = cdf_.size(1);
        TORCH_CHECK(g_.sizes() == at::IntArrayRef(gshape) && g_.scalar_type() == at::kFloat && g_.device() == x_.device(), "cfn::time_sample_backward: g ", g_.sizes(),
                    " is not the fp32 gradient of the (B, C, K, ...) output");
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor g = g_.contiguous(), x = x_.contiguous(), cdf = cdf_.contiguous();
    const int64_t B = x.size(0), C = x.size(1), Tin = x.size(2), K = cdf.size(1);
    Tensor gx = at::empty_like(x), g64 = f64({B, K}, x);
    const int64_t inner = x.numel() / (B * C * Tin);
    ok(cfn_time_sample_bwd(g.data_ptr<float>(), x.data_ptr<float>(), cdf.data_ptr<float>(), gx.data_ptr<float>(), g64.data_ptr<double>(), (int)B, (int)C, (int)Tin, (int)K,
                           (long)inner, stream_of(x)),
       "cfn_time_sample_bwd");
    return {gx, g64.to(at::kFloat)};
}


// =====================================================================================================================================
// The rest of the section-8(b) operator set (round 6): the same schemas cfn_hip/torchlib.py used to define in Python over ctypes, now defined
// and implemented here.  fp32 tensors (the 16-bit activation paths are reached through cfn_hip.ops); prologue coefficients may be fp32 or fp64
// (the C ABI takes fp64), their gradients come back in the coefficient's own type; "no tensor" results are 1-element fp32 placeholders
// (results of a dispatcher operator may not be None).
// =====================================================================================================================================
inline Tensor opt(OptT t) { return t.has_value() && t->defined() ? *t : Tensor(); }
inline Tensor f32c(const Tensor& t) { return t.defined() ? t.to(at::kFloat).contiguous() : t; }
inline Tensor f64c(const Tensor& t) { return t.defined() ? t.to(at::kDouble).contiguous() : t; }
inline const float* fptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline float* fmut(Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline double* dmut(Tensor& t) { return t.defined() ? t.data_ptr<double>() : nullptr; }
inline Tensor z1(const Tensor& ref) { return at::zeros({1}, ref.options().dtype(at::kFloat)); }
// a result that aliases nothing, in the type of `like`
inline Tensor own(const Tensor& t, const Tensor& like) { return t.scalar_type() == like.scalar_type() ? t.clone() : t.to(like.scalar_type()); }
inline void check_f32(const Tensor& t, const char* op, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat, op, ": ", name, " must be an fp32 device tensor (there is no CPU path), got ", t.scalar_type(), " on ", t.device());
}
inline void check_same(const Tensor& t, const Tensor& ref, const char* op, const char* name, const char* refname) {
    check_f32(t, op, name);
    TORCH_CHECK(t.sizes() == ref.sizes() && t.device() == ref.device(), op, ": ", name, " ", t.sizes(), " does not match ", refname, " ", ref.sizes());
}
inline void check_nc(const Tensor& t, int64_t N, int64_t C, const Tensor& x, const char* op, const char* name) {
    TORCH_CHECK(t.defined() && t.numel() == N * C && t.device() == x.device(), op, ": ", name, " must hold ", N, " x ", C, " per-sample coefficients on x's device");
}

// ---- conv1_t: depthwise 5x1x1 (x3d_fine.py:216-222) ----------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> dwconv_t5(const Tensor& x_, const Tensor& w) {
    check_f32(x_, "cfn::dwconv_t5", "x");
    TORCH_CHECK(x_.dim() == 5 && w.numel() == x_.size(1) * 5 && w.device() == x_.device(), "cfn::dwconv_t5: x (N, C, T, H, W), w C x 5 taps on x's device");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), w2 = f32c(w.reshape({x_.size(1), 5}));
    const int64_t N = x.size(0), C = x.size(1), T = x.size(2), plane = x.size(3) * x.size(4);
    Tensor y = at::empty_like(x), s = f64({N, C}, x), q = f64({N, C}, x);
    ok(cfn_dwconv_t5_fwd(x.data_ptr<float>(), w2.data_ptr<float>(), y.data_ptr<float>(), s.data_ptr<double>(), q.data_ptr<double>(), (int)N, (int)C, (int)T, (long)plane, stream_of(x)),
       "cfn_dwconv_t5_fwd");
    return {y, s, q};
}

std::tuple<Tensor, Tensor> dwconv_t5_backward(const Tensor& gy_, const Tensor& gs, const Tensor& gq, const Tensor& x_, const Tensor& w, const Tensor& y_) {
    check_f32(x_, "cfn::dwconv_t5_backward", "x");
    TORCH_CHECK(x_.dim() == 5 && w.numel() == x_.size(1) * 5, "cfn::dwconv_t5_backward: x (N, C, T, H, W), w C x 5 taps");
    check_same(y_, x_, "cfn::dwconv_t5_backward", "y", "x");
    check_same(gy_, x_, "cfn::dwconv_t5_backward", "gy", "x");
    check_stat(gs, x_.size(0), x_.size(1), x_, "cfn::dwconv_t5_backward", "gs");
    check_stat(gq, x_.size(0), x_.size(1), x_, "cfn::dwconv_t5_backward", "gq");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), y = y_.contiguous(), gy = gy_.contiguous(), w2 = f32c(w.reshape({x_.size(1), 5}));
    const Tensor gs64 = f64c(gs), gq64 = f64c(gq);
    const int n = (int)x.size(0), c = (int)x.size(1), t = (int)x.size(2);
    const long plane = (long)(x.size(3) * x.size(4));
    Tensor gx = at::empty_like(x), gw = f64({x.size(1), 5}, x);
    void* st = stream_of(x);
    int rc = cfn_dwconv_t5_bwd_fused(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), w2.data_ptr<float>(), x.data_ptr<float>(), gx.data_ptr<float>(),
                                     gw.data_ptr<double>(), n, c, t, plane, st);
    if (rc == -1) {          // planes that are not whole float4s: data and weight gradient apart
        ok(cfn_dwconv_t5_bwd_data(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), w2.data_ptr<float>(), gx.data_ptr<float>(), n, c, t, plane, st), "cfn_dwconv_t5_bwd_data");
        ok(cfn_dwconv_t5_bwd_weight(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), x.data_ptr<float>(), gw.data_ptr<double>(), n, c, t, plane, st), "cfn_dwconv_t5_bwd_weight");
    } else ok(rc, "cfn_dwconv_t5_bwd_fused");
    return {gx, gw.to(at::kFloat).view(w.sizes())};
}

// ---- conv1_s: dense 1x3x3 stem conv, stride (1,2,2) (x3d_fine.py:210-215); the clip gets no gradient -------------------------------------
Tensor stem_conv(const Tensor& x_, const Tensor& w) {
    check_f32(x_, "cfn::stem_conv", "x");
    TORCH_CHECK(x_.dim() == 5 && w.dim() >= 2 && w.numel() == w.size(0) * x_.size(1) * 9 && w.device() == x_.device(), "cfn::stem_conv: x (N, Ci, T, H, W), w (Co, Ci, 1, 3, 3) on x's device");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous();
    const int64_t N = x.size(0), Ci = x.size(1), T = x.size(2), H = x.size(3), W = x.size(4), Co = w.size(0);
    const Tensor w2 = f32c(w.reshape({Co, Ci * 9}));
    Tensor y = at::empty({N, Co, T, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1}, x.options());
    ok(cfn_stem_conv_fwd(x.data_ptr<float>(), w2.data_ptr<float>(), y.data_ptr<float>(), (int)N, (int)Ci, (int)Co, (int)T, (int)H, (int)W, stream_of(x)), "cfn_stem_conv_fwd");
    return y;
}

Tensor stem_conv_backward(const Tensor& gy_, const Tensor& x_, const Tensor& w) {
    check_f32(x_, "cfn::stem_conv_backward", "x");
    check_f32(gy_, "cfn::stem_conv_backward", "gy");
    TORCH_CHECK(x_.dim() == 5 && w.dim() >= 2 && w.numel() == w.size(0) * x_.size(1) * 9, "cfn::stem_conv_backward: x (N, Ci, T, H, W), w (Co, Ci, 1, 3, 3)");
    const int64_t N = x_.size(0), Ci = x_.size(1), T = x_.size(2), H = x_.size(3), W = x_.size(4), Co = w.size(0);
    TORCH_CHECK(gy_.dim() == 5 && gy_.size(0) == N && gy_.size(1) == Co && gy_.size(2) == T && gy_.size(3) == (H + 2 - 3) / 2 + 1 && gy_.size(4) == (W + 2 - 3) / 2 + 1 &&
                    gy_.device() == x_.device(), "cfn::stem_conv_backward: gy ", gy_.sizes(), " is not the gradient of the stem conv of x ", x_.sizes());
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), gy = gy_.contiguous();
    Tensor gw = f64({Co, Ci * 9}, x);
    ok(cfn_stem_conv_bwd_weight(gy.data_ptr<float>(), x.data_ptr<float>(), gw.data_ptr<double>(), (int)N, (int)Ci, (int)Co, (int)T, (int)H, (int)W, stream_of(x)), "cfn_stem_conv_bwd_weight");
    return gw.to(at::kFloat).view(w.sizes());
}

// ---- dense conv3d (Grid Pool saliency convs, x3d_coarse.py:362-366) -------------------------------------------------------------------
struct Geom { int g[9]; int64_t o[3]; };
inline Geom geom_of(const Tensor& x, at::IntArrayRef kernel, at::IntArrayRef stride, at::IntArrayRef padding, const char* op) {
    TORCH_CHECK(kernel.size() == 3 && stride.size() == 3 && padding.size() == 3, op, ": kernel / stride / padding take three values each");
    Geom r;
    for (int i = 0; i < 3; ++i) {
        TORCH_CHECK(kernel[i] >= 1 && stride[i] >= 1 && padding[i] >= 0, op, ": bad geometry");
        r.g[i] = (int)kernel[i]; r.g[3 + i] = (int)stride[i]; r.g[6 + i] = (int)padding[i];
        r.o[i] = (x.size(2 + i) + 2 * padding[i] - kernel[i]) / stride[i] + 1;
    }
    return r;
}

std::tuple<Tensor, Tensor, Tensor> conv3d_dense(const Tensor& x_, const Tensor& w, at::IntArrayRef kernel, at::IntArrayRef stride, at::IntArrayRef padding, OptT A, OptT B, int64_t act) {
    check_f32(x_, "cfn::conv3d_dense", "x");
    TORCH_CHECK(x_.dim() == 5, "cfn::conv3d_dense: x must be (N, Ci, T, H, W)");
    const Geom gm = geom_of(x_, kernel, stride, padding, "cfn::conv3d_dense");
    const int64_t N = x_.size(0), Ci = x_.size(1), Co = w.size(0);
    TORCH_CHECK(w.numel() == Co * Ci * kernel[0] * kernel[1] * kernel[2] && w.device() == x_.device(), "cfn::conv3d_dense: w ", w.sizes(), " is not (Co, Ci, kt, kh, kw) on x's device");
    check_coef(A, B, N, Ci, x_, "cfn::conv3d_dense");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), w2 = f32c(w.reshape({Co, -1})), A64 = c64(A), B64 = c64(B);
    Tensor y = at::empty({N, Co, gm.o[0], gm.o[1], gm.o[2]}, x.options()), s = f64({N, Co}, x), q = f64({N, Co}, x);
    ok(cfn_conv3d_dense_fwd(x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act, w2.data_ptr<float>(), y.data_ptr<float>(), s.data_ptr<double>(), q.data_ptr<double>(), (int)N, (int)Ci, (int)Co,
                            (int)x.size(2), (int)x.size(3), (int)x.size(4), gm.g, stream_of(x)),
       "cfn_conv3d_dense_fwd");
    return {y, s, q};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> conv3d_dense_backward(const Tensor& gy_, const Tensor& gs, const Tensor& gq, const Tensor& x_, const Tensor& w, const Tensor& y_,
                                                                 at::IntArrayRef kernel, at::IntArrayRef stride, at::IntArrayRef padding, OptT A, OptT B, int64_t act) {
    check_f32(x_, "cfn::conv3d_dense_backward", "x");
    TORCH_CHECK(x_.dim() == 5, "cfn::conv3d_dense_backward: x must be (N, Ci, T, H, W)");
    const Geom gm = geom_of(x_, kernel, stride, padding, "cfn::conv3d_dense_backward");
    const int64_t N = x_.size(0), Ci = x_.size(1), Co = w.size(0);
    TORCH_CHECK(w.numel() == Co * Ci * kernel[0] * kernel[1] * kernel[2] && w.device() == x_.device(), "cfn::conv3d_dense_backward: w ", w.sizes(), " is not (Co, Ci, kt, kh, kw) on x's device");
    check_f32(y_, "cfn::conv3d_dense_backward", "y");
    TORCH_CHECK(y_.dim() == 5 && y_.size(0) == N && y_.size(1) == Co && y_.size(2) == gm.o[0] && y_.size(3) == gm.o[1] && y_.size(4) == gm.o[2] && y_.device() == x_.device(),
                "cfn::conv3d_dense_backward: y ", y_.sizes(), " is not the output of this conv on x ", x_.sizes());
    check_same(gy_, y_, "cfn::conv3d_dense_backward", "gy", "y");
    check_stat(gs, N, Co, x_, "cfn::conv3d_dense_backward", "gs");
    check_stat(gq, N, Co, x_, "cfn::conv3d_dense_backward", "gq");
    check_coef(A, B, N, Ci, x_, "cfn::conv3d_dense_backward");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), y = y_.contiguous(), gy = gy_.contiguous(), w2 = f32c(w.reshape({Co, -1})), A64 = c64(A), B64 = c64(B), gs64 = f64c(gs), gq64 = f64c(gq);
    Tensor gx = at::empty_like(x), gw = f64({Co, w2.size(1)}, x);
    const bool pro = A64.defined();
    Tensor ab = pro ? f64({2, N, Ci}, x) : Tensor();
    double* a64 = pro ? ab.data_ptr<double>() : nullptr;
    double* b64 = pro ? a64 + N * Ci : nullptr;
    void* st = stream_of(x);
    const int n = (int)N, ci = (int)Ci, co = (int)Co, t = (int)x.size(2), h = (int)x.size(3), wd = (int)x.size(4);
    ok(cfn_conv3d_dense_bwd_data(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), w2.data_ptr<float>(), x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act, gx.data_ptr<float>(),
                                 a64, b64, n, ci, co, t, h, wd, gm.g, st),
       "cfn_conv3d_dense_bwd_data");
    ok(cfn_conv3d_dense_bwd_weight(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act, gw.data_ptr<double>(), n, ci, co, t, h, wd,
                                   gm.g, st),
       "cfn_conv3d_dense_bwd_weight");
    Tensor gwf = gw.to(at::kFloat).view(w.sizes());
    if (!pro) return {gx, gwf, z1(x), z1(x)};
    return {gx, gwf, own(ab[0], *A), own(ab[1], *B)};
}

// ---- SubBatchNorm3d statistics -> prologue coefficients (+ squeeze-excite gate) (x3d_fine.py:13-62, :157-163) -----------------------------
// -> [A, B, mean, rstd, A0, B0, gate, hbuf, pooled, new_run_mean, new_run_var, new_nbt]: FUNCTIONAL (the running statistics are returned, the caller
// copies them into its buffers); A0 .. pooled are 1-element placeholders without an SE branch
std::vector<Tensor> bn_fold(OptT s_, OptT q_, OptT gamma_, OptT beta_, const Tensor& run_mean, const Tensor& run_var, const Tensor& nbt, bool training, int64_t N, int64_t C, int64_t S,
                            double count, double eps, double momentum, OptT w1_, OptT b1_, OptT w2_, OptT b2_, double pool_count) {
    const char* op = "cfn::bn_fold";
    TORCH_CHECK(run_mean.is_cuda() && run_mean.scalar_type() == at::kFloat && run_var.scalar_type() == at::kFloat && nbt.scalar_type() == at::kLong, op, ": running statistics fp32 / int64 on the device");
    TORCH_CHECK(N >= 1 && C >= 1 && S >= 1 && N % S == 0, op, ": N = ", N, " samples do not split into S = ", S, " groups");
    const int64_t Se = training ? S : 1;
    TORCH_CHECK(run_mean.numel() == Se * C && run_var.numel() == Se * C, op, ": running statistics must hold ", Se, " x ", C, " values, got ", run_mean.sizes());
    c10::hip::HIPGuardMasqueradingAsCUDA guard(run_mean.device());
    const Tensor s = f64c(opt(s_)), q = f64c(opt(q_)), gamma = f32c(opt(gamma_)), beta = f32c(opt(beta_)), b1 = f32c(opt(b1_)), b2 = f32c(opt(b2_));
    TORCH_CHECK(!training || (s.defined() && q.defined()), op, ": training mode needs sum and sumsq");
    if (s.defined()) check_nc(s, N, C, run_mean, op, "s");
    if (q.defined()) check_nc(q, N, C, run_mean, op, "q");
    TORCH_CHECK(!gamma.defined() || gamma.numel() == C, op, ": gamma must hold C values");
    TORCH_CHECK(!beta.defined() || beta.numel() == C, op, ": beta must hold C values");
    const Tensor w1 = opt(w1_), w2 = opt(w2_);
    TORCH_CHECK(w1.defined() == w2.defined(), op, ": the SE matrices come together");
    const int64_t Wd = w1.defined() ? w1.size(0) : 0;
    Tensor w1c, w2c;
    if (Wd) {
        TORCH_CHECK(w1.numel() == Wd * C && w2.numel() == C * Wd && s.defined(), op, ": SE gate: fc1 (Wd, C), fc2 (C, Wd) and the per-sample sums are needed");
        TORCH_CHECK(!b1.defined() || b1.numel() == Wd, op, ": fc1 bias must hold Wd values");
        TORCH_CHECK(!b2.defined() || b2.numel() == C, op, ": fc2 bias must hold C values");
        w1c = f32c(w1.reshape({Wd, C}));
        w2c = f32c(w2.reshape({C, Wd}));
    }
    Tensor rm = run_mean.clone(), rv = run_var.clone(), nb = nbt.clone();
    const auto o64 = run_mean.options().dtype(at::kDouble), o32 = run_mean.options().dtype(at::kFloat);
    Tensor A = at::empty({N, C}, o64), B = at::empty({N, C}, o64), mean = at::empty({Se, C}, o64), rstd = at::empty({Se, C}, o64);
    Tensor A0, B0, gate, hbuf, pooled;
    if (Wd) { A0 = at::empty({N, C}, o32); B0 = at::empty({N, C}, o32); gate = at::empty({N, C}, o32); pooled = at::empty({N, C}, o32); hbuf = at::empty({N, Wd}, o32); }
    ok(cfn_bn_fold_fwd(dptr(s), dptr(q), fptr(gamma), fptr(beta), rm.data_ptr<float>(), rv.data_ptr<float>(), training ? (long*)nb.data_ptr<int64_t>() : nullptr, (int)training, (int)N, (int)C, (int)S, count,
                       eps, momentum, fptr(w1c), fptr(b1), fptr(w2c), fptr(b2), (int)Wd, pool_count, A.data_ptr<double>(), B.data_ptr<double>(), mean.data_ptr<double>(), rstd.data_ptr<double>(),
                       fmut(A0), fmut(B0), fmut(gate), fmut(hbuf), fmut(pooled), stream_of(run_mean)),
       "cfn_bn_fold_fwd");
    auto orz = [&](const Tensor& t) { return t.defined() ? t : z1(A); };
    return {A, B, mean, rstd, orz(A0), orz(B0), orz(gate), orz(hbuf), orz(pooled), rm, rv, nb};
}

// -> [gs, gq, ggamma, gbeta, gw1, gb1, gw2, gb2] (1-element placeholders where there is nothing); saved = [mean, rstd, A0, B0, gate, hbuf, pooled]
std::vector<Tensor> bn_fold_backward(const Tensor& gA_, const Tensor& gB_, OptT s_, OptT gamma_, at::TensorList saved, bool training, int64_t N, int64_t C, int64_t S, double count,
                                     double pool_count, OptT w1_, OptT w2_) {
    const char* op = "cfn::bn_fold_backward";
    TORCH_CHECK(saved.size() == 7, op, ": saved = [mean, rstd, A0, B0, gate, hbuf, pooled]");
    const Tensor& mean = saved[0];
    const Tensor& rstd = saved[1];
    TORCH_CHECK(mean.is_cuda() && mean.scalar_type() == at::kDouble && rstd.scalar_type() == at::kDouble, op, ": mean / rstd are the fp64 device tensors of the forward");
    const int64_t Se = training ? S : 1;
    TORCH_CHECK(mean.numel() == Se * C && rstd.numel() == Se * C, op, ": mean / rstd must hold ", Se, " x ", C, " values");
    check_nc(gA_, N, C, mean, op, "gA");
    check_nc(gB_, N, C, mean, op, "gB");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(mean.device());
    const Tensor gA = f64c(gA_), gB = f64c(gB_), s = f64c(opt(s_)), gamma = f32c(opt(gamma_)), w1 = opt(w1_), w2 = opt(w2_);
    if (s.defined()) check_nc(s, N, C, mean, op, "s");
    TORCH_CHECK(!gamma.defined() || gamma.numel() == C, op, ": gamma must hold C values");
    TORCH_CHECK(w1.defined() == w2.defined(), op, ": the SE matrices come together");
    const int64_t Wd = w1.defined() ? w1.size(0) : 0;
    Tensor A0, B0, gate, hbuf, pooled, w1c, w2c;
    if (Wd) {
        TORCH_CHECK(w1.numel() == Wd * C && w2.numel() == C * Wd && s.defined(), op, ": SE gate: fc1 (Wd, C), fc2 (C, Wd) and the per-sample sums are needed");
        A0 = f32c(saved[2]); B0 = f32c(saved[3]); gate = f32c(saved[4]); hbuf = f32c(saved[5]); pooled = f32c(saved[6]);
        TORCH_CHECK(A0.numel() == N * C && B0.numel() == N * C && gate.numel() == N * C && pooled.numel() == N * C && hbuf.numel() == N * Wd, op, ": SE intermediates of another shape");
        w1c = f32c(w1.reshape({Wd, C}));
        w2c = f32c(w2.reshape({C, Wd}));
    }
    const auto o64 = mean.options().dtype(at::kDouble), o32 = mean.options().dtype(at::kFloat);
    Tensor gs, gq, gg, gbt, gw1, gb1, gw2, gb2, tA, tB;
    if (training) { gs = at::empty({N, C}, o64); gq = at::empty({N, C}, o64); }
    else if (Wd) gs = at::empty({N, C}, o64);          // eval mode: the statistics are constants, the SE gate still depends on sum(y)
    if (gamma.defined()) { gg = at::empty({C}, o32); gbt = at::empty({C}, o32); }
    if (Wd) { gw1 = at::empty({Wd, C}, o32); gb1 = at::empty({Wd}, o32); gw2 = at::empty({C, Wd}, o32); gb2 = at::empty({C}, o32); tA = at::empty({N, C}, o64); tB = at::empty({N, C}, o64); }
    ok(cfn_bn_fold_bwd(gA.data_ptr<double>(), gB.data_ptr<double>(), dptr(s), fptr(gamma), mean.data_ptr<double>(), rstd.data_ptr<double>(), fptr(A0), fptr(B0), fptr(gate), fptr(hbuf), fptr(pooled),
                       fptr(w1c), fptr(w2c), (int)training, (int)N, (int)C, (int)S, (int)Wd, count, pool_count, dmut(gs), dmut(gq), fmut(gg), fmut(gbt), fmut(gw1), fmut(gb1), fmut(gw2), fmut(gb2),
                       dmut(tA), dmut(tB), stream_of(mean)),
       "cfn_bn_fold_bwd");
    if (Wd) { gw1 = gw1.view(w1.sizes()); gw2 = gw2.view(w2.sizes()); }
    auto orz = [&](const Tensor& t) { return t.defined() ? t : z1(mean); };
    return {orz(gs), orz(gq), orz(gg), orz(gbt), orz(gw1), orz(gb1), orz(gw2), orz(gb2)};
}

// ---- block tail: relu(A y + B + (Ar res + Br)) (x3d_fine.py:165-175) ---------------------------------------------------------------------
Tensor bn_add_relu(const Tensor& y_, const Tensor& A, const Tensor& B, const Tensor& res_, OptT Ar, OptT Br) {
    const char* op = "cfn::bn_add_relu";
    check_f32(y_, op, "y");
    TORCH_CHECK(y_.dim() >= 2, op, ": y must be (N, C, ...)");
    check_same(res_, y_, op, "res", "y");
    const int64_t N = y_.size(0), C = y_.size(1);
    check_nc(A, N, C, y_, op, "A");
    check_nc(B, N, C, y_, op, "B");
    check_coef(Ar, Br, N, C, y_, op);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(y_.device());
    const Tensor y = y_.contiguous(), res = res_.contiguous(), A64 = f64c(A), B64 = f64c(B), Ar64 = c64(Ar), Br64 = c64(Br);
    Tensor out = at::empty_like(y);
    ok(cfn_bn_add_relu_fwd(y.data_ptr<float>(), dptr(A64), dptr(B64), res.data_ptr<float>(), dptr(Ar64), dptr(Br64), out.data_ptr<float>(), nullptr, (long)(N * C), (long)(y.numel() / (N * C)), stream_of(y)),
       "cfn_bn_add_relu_fwd");
    return out;
}

// -> [gy, gA, gB, gres, gAr, gBr]
std::vector<Tensor> bn_add_relu_backward(const Tensor& gout_, const Tensor& y_, const Tensor& A, const Tensor& res_, const Tensor& out_, OptT Ar) {
    const char* op = "cfn::bn_add_relu_backward";
    check_f32(y_, op, "y");
    TORCH_CHECK(y_.dim() >= 2, op, ": y must be (N, C, ...)");
    check_same(res_, y_, op, "res", "y");
    check_same(out_, y_, op, "out", "y");
    check_same(gout_, y_, op, "gout", "y");
    const int64_t N = y_.size(0), C = y_.size(1);
    check_nc(A, N, C, y_, op, "A");
    const Tensor Art = opt(Ar);
    if (Art.defined()) check_nc(Art, N, C, y_, op, "Ar");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(y_.device());
    const Tensor y = y_.contiguous(), res = res_.contiguous(), out = out_.contiguous(), gout = gout_.contiguous(), A64 = f64c(A), Ar64 = f64c(Art);
    Tensor gy = at::empty_like(y), gres = at::empty_like(res), t3 = f64({3, N, C}, y);
    double* t = t3.data_ptr<double>();
    ok(cfn_bn_add_relu_bwd(gout.data_ptr<float>(), nullptr, out.data_ptr<float>(), y.data_ptr<float>(), A64.data_ptr<double>(), res.data_ptr<float>(), dptr(Ar64), gy.data_ptr<float>(),
                           gres.data_ptr<float>(), t, t + N * C, Art.defined() ? t + 2 * N * C : nullptr, (long)(N * C), (long)(y.numel() / (N * C)), stream_of(y)),
       "cfn_bn_add_relu_bwd");
    if (!Art.defined()) return {gy, own(t3[0], A), own(t3[1], A), gres, z1(y), z1(y)};
    return {gy, own(t3[0], A), own(t3[1], A), gres, own(t3[2], Art), own(t3[1], Art)};      // (d/dBr = d/dB)
}

// ---- act(A x + B) materialised ---------------------------------------------------------------------------------------------------------
Tensor affine_act(const Tensor& x_, const Tensor& A, const Tensor& B, int64_t act) {
    const char* op = "cfn::affine_act";
    check_f32(x_, op, "x");
    TORCH_CHECK(x_.dim() >= 2, op, ": x must be (N, C, ...)");
    const int64_t N = x_.size(0), C = x_.size(1);
    check_nc(A, N, C, x_, op, "A");
    check_nc(B, N, C, x_, op, "B");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), A64 = f64c(A), B64 = f64c(B);
    Tensor out = at::empty_like(x);
    ok(cfn_affine_act_fwd(x.data_ptr<float>(), A64.data_ptr<double>(), B64.data_ptr<double>(), (int)act, out.data_ptr<float>(), (long)(N * C), (long)(x.numel() / (N * C)), stream_of(x)), "cfn_affine_act_fwd");
    return out;
}

std::tuple<Tensor, Tensor, Tensor> affine_act_backward(const Tensor& gout_, const Tensor& x_, const Tensor& A, const Tensor& B, int64_t act) {
    const char* op = "cfn::affine_act_backward";
    check_f32(x_, op, "x");
    TORCH_CHECK(x_.dim() >= 2, op, ": x must be (N, C, ...)");
    check_same(gout_, x_, op, "gout", "x");
    const int64_t N = x_.size(0), C = x_.size(1);
    check_nc(A, N, C, x_, op, "A");
    check_nc(B, N, C, x_, op, "B");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), gout = gout_.contiguous(), A64 = f64c(A), B64 = f64c(B);
    Tensor gx = at::empty_like(x), ab = f64({2, N, C}, x);
    ok(cfn_affine_act_bwd(gout.data_ptr<float>(), x.data_ptr<float>(), A64.data_ptr<double>(), B64.data_ptr<double>(), (int)act, gx.data_ptr<float>(), ab.data_ptr<double>(), ab.data_ptr<double>() + N * C,
                          (long)(N * C), (long)(x.numel() / (N * C)), stream_of(x)),
       "cfn_affine_act_bwd");
    return {gx, own(ab[0], A), own(ab[1], B)};
}

// ---- adaptive (OH, OW) spatial mean of act(A x + B) (x3d_fine.py:356-363) ------------------------------------------------------------------
Tensor pool_hw(const Tensor& x_, int64_t OH, int64_t OW, OptT A, OptT B, int64_t act) {
    const char* op = "cfn::pool_hw";
    check_f32(x_, op, "x");
    TORCH_CHECK(x_.dim() == 5 && OH >= 1 && OW >= 1 && OH <= x_.size(3) && OW <= x_.size(4), op, ": x (N, C, T, H, W) pooled to 1 <= OH <= H, 1 <= OW <= W");
    const int64_t N = x_.size(0), C = x_.size(1);
    check_coef(A, B, N, C, x_, op);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), A64 = c64(A), B64 = c64(B);
    Tensor out = at::empty({N, C, x.size(2), OH, OW}, x.options());
    ok(cfn_pool_hw_fwd(x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act, out.data_ptr<float>(), (long)(N * C), (int)x.size(2), (int)x.size(3), (int)x.size(4), (int)OH, (int)OW, stream_of(x)), "cfn_pool_hw_fwd");
    return out;
}

std::tuple<Tensor, Tensor, Tensor> pool_hw_backward(const Tensor& gout_, const Tensor& x_, int64_t OH, int64_t OW, OptT A, OptT B, int64_t act) {
    const char* op = "cfn::pool_hw_backward";
    check_f32(x_, op, "x");
    check_f32(gout_, op, "gout");
    TORCH_CHECK(x_.dim() == 5 && OH >= 1 && OW >= 1 && OH <= x_.size(3) && OW <= x_.size(4), op, ": x (N, C, T, H, W) pooled to 1 <= OH <= H, 1 <= OW <= W");
    const int64_t N = x_.size(0), C = x_.size(1);
    TORCH_CHECK(gout_.dim() == 5 && gout_.size(0) == N && gout_.size(1) == C && gout_.size(2) == x_.size(2) && gout_.size(3) == OH && gout_.size(4) == OW && gout_.device() == x_.device(), op,
                ": gout ", gout_.sizes(), " is not the gradient of the (N, C, T, OH, OW) output");
    check_coef(A, B, N, C, x_, op);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), gout = gout_.contiguous(), A64 = c64(A), B64 = c64(B);
    const bool pro = A64.defined();
    Tensor gx = at::empty_like(x), ab = pro ? f64({2, N, C}, x) : Tensor();
    double* a64 = pro ? ab.data_ptr<double>() : nullptr;
    ok(cfn_pool_hw_bwd(gout.data_ptr<float>(), x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act, gx.data_ptr<float>(), a64, pro ? a64 + N * C : nullptr, (long)(N * C), (int)x.size(2), (int)x.size(3),
                       (int)x.size(4), (int)OH, (int)OW, stream_of(x)),
       "cfn_pool_hw_bwd");
    if (!pro) return {gx, z1(x), z1(x)};
    return {gx, own(ab[0], *A), own(ab[1], *B)};
}

// ---- Interp1d (interp1d.py:8-147): 2-D inputs, rows broadcast when a tensor has a single row -> (ynew, ind) ---------------------------------
inline void interp_shapes(const Tensor& x, const Tensor& y, const Tensor& xnew, const char* op, int64_t& B, int (&rows)[3]) {
    check_f32(x, op, "x"); check_f32(y, op, "y"); check_f32(xnew, op, "xnew");
    TORCH_CHECK(x.dim() == 2 && y.dim() == 2 && xnew.dim() == 2 && y.size(1) == x.size(1) && x.size(1) >= 2, op, ": x (Bx, N), y (By, N), xnew (Bq, P), N >= 2");
    B = std::max(x.size(0), std::max(y.size(0), xnew.size(0)));
    const Tensor* ts[3] = {&x, &y, &xnew};
    for (int i = 0; i < 3; ++i) {
        TORCH_CHECK(ts[i]->size(0) == 1 || ts[i]->size(0) == B, op, ": row counts must be 1 or ", B);
        TORCH_CHECK(ts[i]->device() == x.device(), op, ": tensors on different devices");
        rows[i] = ts[i]->size(0) > 1 ? 1 : 0;
    }
}

std::tuple<Tensor, Tensor> interp1d(const Tensor& x_, const Tensor& y_, const Tensor& xnew_) {
    int64_t B;
    int rows[3];
    interp_shapes(x_, y_, xnew_, "cfn::interp1d", B, rows);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), y = y_.contiguous(), xnew = xnew_.contiguous();
    const int64_t Pq = xnew.size(1);
    Tensor ynew = at::empty({B, Pq}, x.options()), ind = at::empty({B, Pq}, x.options().dtype(at::kLong));
    ok(cfn_interp1d_fwd(x.data_ptr<float>(), y.data_ptr<float>(), xnew.data_ptr<float>(), ynew.data_ptr<float>(), (long*)ind.data_ptr<int64_t>(), (int)B, (int)x.size(1), (int)Pq, rows[0], rows[1], rows[2],
                        stream_of(x)),
       "cfn_interp1d_fwd");
    return {ynew, ind};
}

std::tuple<Tensor, Tensor, Tensor> interp1d_backward(const Tensor& g_, const Tensor& x_, const Tensor& y_, const Tensor& xnew_, const Tensor& ind_) {
    int64_t B;
    int rows[3];
    interp_shapes(x_, y_, xnew_, "cfn::interp1d_backward", B, rows);
    const int64_t Pq = xnew_.size(1);
    check_f32(g_, "cfn::interp1d_backward", "g");
    TORCH_CHECK(g_.dim() == 2 && g_.size(0) == B && g_.size(1) == Pq && ind_.dim() == 2 && ind_.size(0) == B && ind_.size(1) == Pq && ind_.scalar_type() == at::kLong && ind_.device() == x_.device() &&
                    g_.device() == x_.device(),
                "cfn::interp1d_backward: g / ind must be the (B, P) results of the forward");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), y = y_.contiguous(), xnew = xnew_.contiguous(), g = g_.contiguous(), ind = ind_.contiguous();
    Tensor gx = at::zeros_like(x), gy = at::zeros_like(y), gq = at::zeros_like(xnew);
    ok(cfn_interp1d_bwd(g.data_ptr<float>(), x.data_ptr<float>(), y.data_ptr<float>(), xnew.data_ptr<float>(), (const long*)ind.data_ptr<int64_t>(), gx.data_ptr<float>(), gy.data_ptr<float>(),
                        gq.data_ptr<float>(), (int)B, (int)x.size(1), (int)Pq, rows[0], rows[1], rows[2], stream_of(x)),
       "cfn_interp1d_bwd");
    return {gx, gy, gq};
}

// ---- Grid Pool CDF (x3d_coarse.py:384-392): saliency logits (B, Kin) (+ scalar bias) -> CDF knots (B, Kin + 1) ------------------------------
Tensor grid_cdf(const Tensor& g_, OptT bias_) {
    check_f32(g_, "cfn::grid_cdf", "g");
    TORCH_CHECK(g_.dim() == 2, "cfn::grid_cdf: g must be (B, Kin)");
    const Tensor bias = f32c(opt(bias_));
    TORCH_CHECK(!bias.defined() || (bias.numel() == 1 && bias.device() == g_.device()), "cfn::grid_cdf: the bias is one value on g's device");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(g_.device());
    const Tensor g = g_.contiguous();
    Tensor cdf = at::empty({g.size(0), g.size(1) + 1}, g.options());
    ok(cfn_grid_cdf_fwd(g.data_ptr<float>(), fptr(bias), cdf.data_ptr<float>(), (int)g.size(0), (int)g.size(1), stream_of(g)), "cfn_grid_cdf_fwd");
    return cdf;
}

std::tuple<Tensor, Tensor> grid_cdf_backward(const Tensor& gcdf_, const Tensor& g_, OptT bias_) {
    check_f32(g_, "cfn::grid_cdf_backward", "g");
    check_f32(gcdf_, "cfn::grid_cdf_backward", "gcdf");
    TORCH_CHECK(g_.dim() == 2 && gcdf_.dim() == 2 && gcdf_.size(0) == g_.size(0) && gcdf_.size(1) == g_.size(1) + 1 && gcdf_.device() == g_.device(), "cfn::grid_cdf_backward: g (B, Kin), gcdf (B, Kin + 1)");
    const Tensor bias = f32c(opt(bias_));
    TORCH_CHECK(!bias.defined() || (bias.numel() == 1 && bias.device() == g_.device()), "cfn::grid_cdf_backward: the bias is one value on g's device");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(g_.device());
    const Tensor g = g_.contiguous(), gcdf = gcdf_.contiguous();
    Tensor gg = at::empty_like(g);
    ok(cfn_grid_cdf_bwd(gcdf.data_ptr<float>(), g.data_ptr<float>(), fptr(bias), gg.data_ptr<float>(), (int)g.size(0), (int)g.size(1), stream_of(g)), "cfn_grid_cdf_bwd");
    return {gg, bias.defined() ? gg.sum().view(bias_->sizes()) : z1(g)};
}

// ---- Gaussian temporal alignment (x3d_coarse.py:251-286): meta (B, 4) int64, mask (B, Tf), gx (B*crops, K) or None -> GX (B*crops, Tf, K) -------
Tensor gauss_align(const Tensor& meta_, const Tensor& mask_, OptT gx_, double tx, double ratio, int64_t crops, int64_t K) {
    const char* op = "cfn::gauss_align";
    check_f32(mask_, op, "mask");
    TORCH_CHECK(mask_.dim() == 2 && meta_.dim() == 2 && meta_.size(0) == mask_.size(0) && meta_.size(1) == 4 && meta_.device() == mask_.device() && crops >= 1 && K >= 1, op,
                ": meta (B, 4), mask (B, Tf) on one device, crops >= 1, K >= 1");
    const int64_t B = mask_.size(0), Tf = mask_.size(1);
    const Tensor gxo = opt(gx_);
    if (gxo.defined()) {
        check_f32(gxo, op, "gx");
        TORCH_CHECK(gxo.dim() == 2 && gxo.size(0) == B * crops && gxo.size(1) == K && gxo.device() == mask_.device(), op, ": gx must be the (B * crops, K) CDF");
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(mask_.device());
    const Tensor meta = meta_.to(at::kLong).contiguous(), mask = mask_.contiguous(), gx = gxo.defined() ? gxo.contiguous() : gxo;
    Tensor GX = at::empty({B * crops, Tf, K}, mask.options());
    ok(cfn_gauss_align_fwd((const long*)meta.data_ptr<int64_t>(), mask.data_ptr<float>(), fptr(gx), tx, ratio, GX.data_ptr<float>(), (int)B, (int)crops, (int)Tf, (int)K, stream_of(mask)), "cfn_gauss_align_fwd");
    return GX;
}

Tensor gauss_align_backward(const Tensor& gGX_, const Tensor& meta_, const Tensor& mask_, const Tensor& gx_, double tx, double ratio, int64_t crops, int64_t K) {
    const char* op = "cfn::gauss_align_backward";
    check_f32(mask_, op, "mask");
    check_f32(gx_, op, "gx");
    check_f32(gGX_, op, "gGX");
    TORCH_CHECK(mask_.dim() == 2 && meta_.dim() == 2 && meta_.size(0) == mask_.size(0) && meta_.size(1) == 4 && meta_.device() == mask_.device() && crops >= 1 && K >= 1, op,
                ": meta (B, 4), mask (B, Tf) on one device, crops >= 1, K >= 1");
    const int64_t B = mask_.size(0), Tf = mask_.size(1);
    TORCH_CHECK(gx_.dim() == 2 && gx_.size(0) == B * crops && gx_.size(1) == K && gx_.device() == mask_.device(), op, ": gx must be the (B * crops, K) CDF");
    TORCH_CHECK(gGX_.dim() == 3 && gGX_.size(0) == B * crops && gGX_.size(1) == Tf && gGX_.size(2) == K && gGX_.device() == mask_.device(), op, ": gGX must be (B * crops, Tf, K)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(mask_.device());
    const Tensor meta = meta_.to(at::kLong).contiguous(), mask = mask_.contiguous(), gx = gx_.contiguous(), gGX = gGX_.contiguous();
    Tensor ggx = at::empty_like(gx);
    ok(cfn_gauss_align_bwd(gGX.data_ptr<float>(), (const long*)meta.data_ptr<int64_t>(), mask.data_ptr<float>(), gx.data_ptr<float>(), tx, ratio, ggx.data_ptr<float>(), (int)B, (int)crops, (int)Tf, (int)K,
                           stream_of(mask)),
       "cfn_gauss_align_bwd");
    return ggx;
}

// ---- Multi-stage-Fusion gather (x3d_coarse.py:199-247): x (B, C, Tf, P), at_raw (B, Tf, P), at_bias (1,) or None, GX (B*crops, Tf, K), mask (B, Tf) ---
inline void fusion_shapes(const Tensor& x, const Tensor& at_raw, const Tensor& GX, const Tensor& mask, int64_t crops, const char* op) {
    check_f32(x, op, "x"); check_f32(at_raw, op, "at_raw"); check_f32(GX, op, "GX"); check_f32(mask, op, "mask");
    TORCH_CHECK(x.dim() == 4 && GX.dim() == 3 && crops >= 1, op, ": x (B, C, Tf, P), GX (B * crops, Tf, K)");
    const int64_t B = x.size(0), Tf = x.size(2), P = x.size(3);
    TORCH_CHECK(GX.size(0) == B * crops && GX.size(1) == Tf && mask.dim() == 2 && mask.size(0) == B && mask.size(1) == Tf && at_raw.dim() == 3 && at_raw.size(0) == B && at_raw.size(1) == Tf &&
                    at_raw.size(2) == P && GX.device() == x.device() && mask.device() == x.device() && at_raw.device() == x.device(),
                op, ": inconsistent shapes x ", x.sizes(), " at ", at_raw.sizes(), " GX ", GX.sizes(), " mask ", mask.sizes(), " crops ", crops);
}

std::tuple<Tensor, Tensor> fusion_gather(const Tensor& x_, const Tensor& at_raw_, OptT at_bias_, const Tensor& GX_, const Tensor& mask_, int64_t crops) {
    const char* op = "cfn::fusion_gather";
    fusion_shapes(x_, at_raw_, GX_, mask_, crops, op);
    const Tensor bias = f32c(opt(at_bias_));
    TORCH_CHECK(!bias.defined() || (bias.numel() == 1 && bias.device() == x_.device()), op, ": the attention bias is one value on x's device");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), at_raw = at_raw_.contiguous(), GX = GX_.contiguous(), mask = mask_.contiguous();
    const int64_t B = x.size(0), C = x.size(1), Tf = x.size(2), P = x.size(3), K = GX.size(2);
    Tensor z = at::empty({B * crops, C, K, P}, x.options()), den = at::empty({B * crops, K, P}, x.options());
    ok(cfn_fusion_gather_fwd(x.data_ptr<float>(), at_raw.data_ptr<float>(), fptr(bias), GX.data_ptr<float>(), mask.data_ptr<float>(), z.data_ptr<float>(), den.data_ptr<float>(), (int)B, (int)crops, (int)C,
                             (int)Tf, (int)K, (int)P, stream_of(x)),
       "cfn_fusion_gather_fwd");
    return {z, den};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> fusion_gather_backward(const Tensor& gz_, const Tensor& z_, const Tensor& den_, const Tensor& x_, const Tensor& at_raw_, OptT at_bias_, const Tensor& GX_,
                                                                  const Tensor& mask_, int64_t crops) {
    const char* op = "cfn::fusion_gather_backward";
    fusion_shapes(x_, at_raw_, GX_, mask_, crops, op);
    const Tensor bias = f32c(opt(at_bias_));
    TORCH_CHECK(!bias.defined() || (bias.numel() == 1 && bias.device() == x_.device()), op, ": the attention bias is one value on x's device");
    const int64_t B = x_.size(0), C = x_.size(1), Tf = x_.size(2), P = x_.size(3), K = GX_.size(2);
    check_f32(z_, op, "z"); check_f32(den_, op, "den");
    TORCH_CHECK(z_.dim() == 4 && z_.size(0) == B * crops && z_.size(1) == C && z_.size(2) == K && z_.size(3) == P && z_.device() == x_.device(), op, ": z must be the (B * crops, C, K, P) result of the forward");
    TORCH_CHECK(den_.dim() == 3 && den_.size(0) == B * crops && den_.size(1) == K && den_.size(2) == P && den_.device() == x_.device(), op, ": den must be the (B * crops, K, P) result of the forward");
    check_same(gz_, z_, op, "gz", "z");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), at_raw = at_raw_.contiguous(), GX = GX_.contiguous(), mask = mask_.contiguous(), z = z_.contiguous(), den = den_.contiguous(), gz = gz_.contiguous();
    Tensor gx = at::empty_like(x), gat = at::empty_like(at_raw), gGX = at::empty_like(GX), dw = at::empty({B * crops, Tf, K, P}, x.options());
    ok(cfn_fusion_gather_bwd(gz.data_ptr<float>(), z.data_ptr<float>(), den.data_ptr<float>(), x.data_ptr<float>(), at_raw.data_ptr<float>(), fptr(bias), GX.data_ptr<float>(), mask.data_ptr<float>(),
                             gx.data_ptr<float>(), gat.data_ptr<float>(), gGX.data_ptr<float>(), dw.data_ptr<float>(), (int)B, (int)crops, (int)C, (int)Tf, (int)K, (int)P, stream_of(x)),
       "cfn_fusion_gather_bwd");
    return {gx, gat, bias.defined() ? gat.sum().view(at_bias_->sizes()) : z1(x), gGX};
}

// ---- block-broadcast FiLM: x * m + c with m, c constant over f x f spatial blocks ((N, C, T, H / f, W / f)) ---------------------------------------
inline void film_shapes(const Tensor& x, const Tensor& m, int64_t f, const char* op) {
    check_f32(x, op, "x"); check_f32(m, op, "m");
    TORCH_CHECK(x.dim() == 5 && f >= 1 && x.size(3) % f == 0 && x.size(4) % f == 0 && m.dim() == 5 && m.size(0) == x.size(0) && m.size(1) == x.size(1) && m.size(2) == x.size(2) &&
                    m.size(3) == x.size(3) / f && m.size(4) == x.size(4) / f && m.device() == x.device(),
                op, ": coefficients ", m.sizes(), " do not tile x ", x.sizes(), " with ", f, " x ", f, " blocks");
}

Tensor film(const Tensor& x_, const Tensor& m_, const Tensor& c_, int64_t f) {
    film_shapes(x_, m_, f, "cfn::film");
    check_same(c_, m_, "cfn::film", "c", "m");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), m = m_.contiguous(), c = c_.contiguous();
    Tensor out = at::empty_like(x);
    ok(cfn_film_fwd(x.data_ptr<float>(), m.data_ptr<float>(), c.data_ptr<float>(), out.data_ptr<float>(), (long)(x.size(0) * x.size(1)), (int)x.size(2), (int)x.size(3), (int)x.size(4), (int)f, stream_of(x)),
       "cfn_film_fwd");
    return out;
}

std::tuple<Tensor, Tensor, Tensor> film_backward(const Tensor& g_, const Tensor& x_, const Tensor& m_, int64_t f) {
    film_shapes(x_, m_, f, "cfn::film_backward");
    check_same(g_, x_, "cfn::film_backward", "g", "x");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous(), m = m_.contiguous(), g = g_.contiguous();
    Tensor gx = at::empty_like(x), gm = at::empty_like(m), gc = at::empty_like(m);
    ok(cfn_film_bwd(g.data_ptr<float>(), x.data_ptr<float>(), m.data_ptr<float>(), gx.data_ptr<float>(), gm.data_ptr<float>(), gc.data_ptr<float>(), (long)(x.size(0) * x.size(1)), (int)x.size(2), (int)x.size(3),
                    (int)x.size(4), (int)f, stream_of(x)),
       "cfn_film_bwd");
    return {gx, gm, gc};
}

// ---- linear resize along t (F.interpolate mode='linear', both align_corners conventions; train_fine.py:203, x3d_coarse.py:446) ------------------
Tensor time_resize(const Tensor& x_, int64_t L, bool align_corners) {
    check_f32(x_, "cfn::time_resize", "x");
    TORCH_CHECK(x_.dim() >= 3 && L >= 1 && x_.size(2) >= 1, "cfn::time_resize: x (B, C, K, ...), L >= 1");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x_.device());
    const Tensor x = x_.contiguous();
    std::vector<int64_t> shape = x.sizes().vec();
    const int64_t BC = x.size(0) * x.size(1), Kin = x.size(2), P = BC * Kin > 0 ? x.numel() / (BC * Kin) : 1;
    shape[2] = L;
    Tensor out = at::empty(shape, x.options());
    ok(cfn_time_resize_fwd(x.data_ptr<float>(), out.data_ptr<float>(), (long)BC, (int)Kin, (int)L, (long)P, (int)align_corners, stream_of(x)), "cfn_time_resize_fwd");
    return out;
}

Tensor time_resize_backward(const Tensor& g_, at::IntArrayRef shape, int64_t L, bool align_corners) {
    check_f32(g_, "cfn::time_resize_backward", "g");
    TORCH_CHECK(shape.size() >= 3 && g_.dim() == (int64_t)shape.size() && L >= 1, "cfn::time_resize_backward: shape (B, C, K, ...) of the forward input");
    int64_t P = 1;
    for (size_t i = 0; i < shape.size(); ++i) {
        TORCH_CHECK(g_.size(i) == (i == 2 ? L : shape[i]), "cfn::time_resize_backward: g ", g_.sizes(), " is not the gradient of the resized ", shape);
        if (i >= 3) P *= shape[i];
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(g_.device());
    const Tensor g = g_.contiguous();
    Tensor gx = at::empty(shape, g.options());
    ok(cfn_time_resize_bwd(g.data_ptr<float>(), gx.data_ptr<float>(), (long)(shape[0] * shape[1]), (int)shape[2], (int)L, (long)P, (int)align_corners, stream_of(g)), "cfn_time_resize_bwd");
    return gx;
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(cfn, m) {
    m.def("dwconv3d(Tensor x, Tensor w, Tensor? A=None, Tensor? B=None, SymInt act=0, SymInt stride=1) -> (Tensor, Tensor, Tensor)");
    m.def("dwconv3d_backward(Tensor gy, Tensor gs, Tensor gq, Tensor x, Tensor w, Tensor y, Tensor? A=None, Tensor? B=None, SymInt act=0, SymInt stride=1) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("pwconv(Tensor x, Tensor w, Tensor? A=None, Tensor? B=None, SymInt act=0, SymInt stride=1) -> (Tensor, Tensor, Tensor)");
    m.def("pwconv_backward(Tensor gy, Tensor gs, Tensor gq, Tensor x, Tensor w, Tensor y, Tensor? A=None, Tensor? B=None, SymInt act=0, SymInt stride=1) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("time_sample(Tensor x, Tensor cdf) -> Tensor");
    m.def("time_sample_backward(Tensor g, Tensor x, Tensor cdf) -> (Tensor, Tensor)");
    m.def("dwconv_t5(Tensor x, Tensor w) -> (Tensor, Tensor, Tensor)");
    m.def("dwconv_t5_backward(Tensor gy, Tensor gs, Tensor gq, Tensor x, Tensor w, Tensor y) -> (Tensor, Tensor)");
    m.def("stem_conv(Tensor x, Tensor w) -> Tensor");
    m.def("stem_conv_backward(Tensor gy, Tensor x, Tensor w) -> Tensor");
    m.def("conv3d_dense(Tensor x, Tensor w, SymInt[] kernel, SymInt[] stride, SymInt[] padding, Tensor? A=None, Tensor? B=None, SymInt act=0) -> (Tensor, Tensor, Tensor)");
    m.def("conv3d_dense_backward(Tensor gy, Tensor gs, Tensor gq, Tensor x, Tensor w, Tensor y, SymInt[] kernel, SymInt[] stride, SymInt[] padding, Tensor? A=None, Tensor? B=None, SymInt act=0) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("bn_fold(Tensor? s, Tensor? q, Tensor? gamma, Tensor? beta, Tensor run_mean, Tensor run_var, Tensor nbt, bool training, SymInt N, SymInt C, SymInt S, float count, float eps, float momentum, Tensor? w1=None, Tensor? b1=None, Tensor? w2=None, Tensor? b2=None, float pool_count=1.) -> Tensor[]");
    m.def("bn_fold_backward(Tensor gA, Tensor gB, Tensor? s, Tensor? gamma, Tensor[] saved, bool training, SymInt N, SymInt C, SymInt S, float count, float pool_count, Tensor? w1=None, Tensor? w2=None) -> Tensor[]");
    m.def("bn_add_relu(Tensor y, Tensor A, Tensor B, Tensor res, Tensor? Ar=None, Tensor? Br=None) -> Tensor");
    m.def("bn_add_relu_backward(Tensor gout, Tensor y, Tensor A, Tensor res, Tensor out, Tensor? Ar=None) -> Tensor[]");
    m.def("affine_act(Tensor x, Tensor A, Tensor B, SymInt act=0) -> Tensor");
    m.def("affine_act_backward(Tensor gout, Tensor x, Tensor A, Tensor B, SymInt act) -> (Tensor, Tensor, Tensor)");
    m.def("pool_hw(Tensor x, SymInt OH, SymInt OW, Tensor? A=None, Tensor? B=None, SymInt act=0) -> Tensor");
    m.def("pool_hw_backward(Tensor gout, Tensor x, SymInt OH, SymInt OW, Tensor? A=None, Tensor? B=None, SymInt act=0) -> (Tensor, Tensor, Tensor)");
    m.def("interp1d(Tensor x, Tensor y, Tensor xnew) -> (Tensor, Tensor)");
    m.def("interp1d_backward(Tensor g, Tensor x, Tensor y, Tensor xnew, Tensor ind) -> (Tensor, Tensor, Tensor)");
    m.def("grid_cdf(Tensor g, Tensor? bias=None) -> Tensor");
    m.def("grid_cdf_backward(Tensor gcdf, Tensor g, Tensor? bias=None) -> (Tensor, Tensor)");
    m.def("gauss_align(Tensor meta, Tensor mask, Tensor? gx, float tx, float ratio, SymInt crops, SymInt K) -> Tensor");
    m.def("gauss_align_backward(Tensor gGX, Tensor meta, Tensor mask, Tensor gx, float tx, float ratio, SymInt crops, SymInt K) -> Tensor");
    m.def("fusion_gather(Tensor x, Tensor at_raw, Tensor? at_bias, Tensor GX, Tensor mask, SymInt crops=1) -> (Tensor, Tensor)");
    m.def("fusion_gather_backward(Tensor gz, Tensor z, Tensor den, Tensor x, Tensor at_raw, Tensor? at_bias, Tensor GX, Tensor mask, SymInt crops) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("film(Tensor x, Tensor m, Tensor c, SymInt f) -> Tensor");
    m.def("film_backward(Tensor g, Tensor x, Tensor m, SymInt f) -> (Tensor, Tensor, Tensor)");
    m.def("time_resize(Tensor x, SymInt L, bool align_corners=True) -> Tensor");
    m.def("time_resize_backward(Tensor g, SymInt[] shape, SymInt L, bool align_corners) -> Tensor");
}

TORCH_LIBRARY_IMPL(cfn, CUDA, m) {      // (the HIP backend of a ROCm build of torch dispatches on the CUDA key)
    m.impl("dwconv3d", dwconv3d);
    m.impl("dwconv3d_backward", dwconv3d_backward);
    m.impl("pwconv", pwconv);
    m.impl("pwconv_backward", pwconv_backward);
    m.impl("time_sample", time_sample);
    m.impl("time_sample_backward", time_sample_backward);
    m.impl("dwconv_t5", dwconv_t5);
    m.impl("dwconv_t5_backward", dwconv_t5_backward);
    m.impl("stem_conv", stem_conv);
    m.impl("stem_conv_backward", stem_conv_backward);
    m.impl("conv3d_dense", conv3d_dense);
    m.impl("conv3d_dense_backward", conv3d_dense_backward);
    m.impl("bn_fold", bn_fold);
    m.impl("bn_fold_backward", bn_fold_backward);
    m.impl("bn_add_relu", bn_add_relu);
    m.impl("bn_add_relu_backward", bn_add_relu_backward);
    m.impl("affine_act", affine_act);
    m.impl("affine_act_backward", affine_act_backward);
    m.impl("pool_hw", pool_hw);
    m.impl("pool_hw_backward", pool_hw_backward);
    m.impl("interp1d", interp1d);
    m.impl("interp1d_backward", interp1d_backward);
    m.impl("grid_cdf", grid_cdf);
    m.impl("grid_cdf_backward", grid_cdf_backward);
    m.impl("gauss_align", gauss_align);
    m.impl("gauss_align_backward", gauss_align_backward);
    m.impl("fusion_gather", fusion_gather);
    m.impl("fusion_gather_backward", fusion_gather_backward);
    m.impl("film", film);
    m.impl("film_backward", film_backward);
    m.impl("time_resize", time_resize);
    m.impl("time_resize_backward", time_resize_backward);
}
