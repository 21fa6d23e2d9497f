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
#include <ATen/ATen.h>
// (a ROCm build of torch presents its HIP devices as "cuda": the masquerading guard / stream classes are the ones that accept them)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <tuple>

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
    ok(cfn_pwconv_bwd_data(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), w2.data_ptr<float>(), x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act,
                           gx.data_ptr<float>(), a64, b64, (int)N, (int)Cin, (int)Cout, (int)T, (int)H, (int)W, (int)stride, st),
       "cfn_pwconv_bwd_data");
    ok(cfn_pwconv_bwd_weight(gy.data_ptr<float>(), y.data_ptr<float>(), dptr(gs64), dptr(gq64), x.data_ptr<float>(), dptr(A64), dptr(B64), (int)act, gw.data_ptr<double>(),
                             (int)N, (int)Cin, (int)Cout, (int)T, (int)H, (int)W, (int)stride, nullptr, st),
       "cfn_pwconv_bwd_weight");
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

}  // namespace

TORCH_LIBRARY_FRAGMENT(cfn, m) {
    m.def("dwconv3d(Tensor x, Tensor w, Tensor? A=None, Tensor? B=None, SymInt act=0, SymInt stride=1) -> (Tensor, Tensor, Tensor)");
    m.def("dwconv3d_backward(Tensor gy, Tensor gs, Tensor gq, Tensor x, Tensor w, Tensor y, Tensor? A=None, Tensor? B=None, SymInt act=0, SymInt stride=1) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("pwconv(Tensor x, Tensor w, Tensor? A=None, Tensor? B=None, SymInt act=0, SymInt stride=1) -> (Tensor, Tensor, Tensor)");
    m.def("pwconv_backward(Tensor gy, Tensor gs, Tensor gq, Tensor x, Tensor w, Tensor y, Tensor? A=None, Tensor? B=None, SymInt act=0, SymInt stride=1) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("time_sample(Tensor x, Tensor cdf) -> Tensor");
    m.def("time_sample_backward(Tensor g, Tensor x, Tensor cdf) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(cfn, CUDA, m) {      // (the HIP backend of a ROCm build of torch dispatches on the CUDA key)
    m.impl("dwconv3d", dwconv3d);
    m.impl("dwconv3d_backward", dwconv3d_backward);
    m.impl("pwconv", pwconv);
    m.impl("pwconv_backward", pwconv_backward);
    m.impl("time_sample", time_sample);
    m.impl("time_sample_backward", time_sample_backward);
}
