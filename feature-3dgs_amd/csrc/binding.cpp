// binding.cpp — pybind11 / libtorch front-end `diff_gaussian_rasterization._C`.
//
// Reproduces the three Python-visible entry points of the reference binding
//   rasterize_gaussians            (reference: rasterize_points.cu:35-124,  ext.cpp:16)
//   rasterize_gaussians_backward   (reference: rasterize_points.cu:126-215, ext.cpp:17)
//   mark_visible                   (reference: rasterize_points.cu:217-236, ext.cpp:18)
// with identical positional signatures and return tuples, on top of the C ABI of
// libf3dgs_hip.so (include/f3dgs.h).  PyTorch is plumbing only: tensor allocation through the
// caching allocator, the current HIP stream, and the three growable byte buffers.
//
// Differences to the reference binding, all deliberate:
//   * the feature dimension C is read from semantic_feature.size(-1) instead of a compile-time macro;
//   * work is enqueued on PyTorch's CURRENT stream (the reference uses the legacy default stream);
//   * outputs / gradients are torch::empty (the kernels overwrite every element) — no zero-fill passes;
//   * inputs that are not on a HIP device raise instead of silently reading host memory.

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <exception>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/f3dgs.h"

namespace py = pybind11;

namespace {

const float* fptr(const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }

char* resize_hook(void* ctx, size_t n) {
    auto* t = static_cast<torch::Tensor*>(ctx);
    t->resize_({(long long)n});
    return reinterpret_cast<char*>(t->data_ptr());
}

void check_status(int rc, const char* where) {
    if (rc != F3DGS_OK) throw std::runtime_error(std::string(where) + ": " + f3dgs_last_error());
}

torch::Tensor dev_f32(const torch::Tensor& t, const char* name) {
    if (t.numel() == 0) return t;
    TORCH_CHECK(t.is_cuda(), name, " must live on a HIP device (got ", t.device(), "): the MI355X rasterizer has no CPU path");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
    torch::Tensor c = t.contiguous();
    // the library reads rows with 16-byte accesses (f3dgs.h, Alignment): a view at an odd storage offset is copied into a fresh
    // (allocator-aligned) tensor instead of being refused - the reference accepts such views
    if (reinterpret_cast<uintptr_t>(c.data_ptr()) & 15) c = c.clone();
    return c;
}

// The kernels index the feature tensor as (P, C): the reference's layout is (P, 1, C) (scene/gaussian_model.py:
// `_semantic_feature`, rasterize_points.cu:160).  Anything else with more than C floats per Gaussian would be
// read with the wrong stride, so it is rejected instead.
int feature_channels(const torch::Tensor& semantic_feature, int64_t P) {
    if (semantic_feature.numel() == 0) return semantic_feature.dim() >= 1 ? (int)semantic_feature.size(-1) : 0;
    TORCH_CHECK(semantic_feature.dim() == 3 && semantic_feature.size(0) == P && semantic_feature.size(1) == 1,
                "semantic_feature must have dimensions (num_points, 1, C); got ", semantic_feature.sizes());
    return (int)semantic_feature.size(2);
}

// Python callable invoked with dL_dsemantic_feature as soon as the blend backward has been enqueued
// (include/f3dgs.h: f3dgs_set_feature_grad_ready_callback).  The GIL is held throughout the binding call.
py::object& feature_grad_hook() {
    static py::object* hook = new py::object(py::none());   // leaked on purpose: no destructor after interpreter exit
    return *hook;
}
// A Python exception must not unwind through the extern "C" frame of f3dgs_backward: it is parked here and rethrown by the
// binding once the C call has returned and the callback registration has been cleared.
std::exception_ptr& pending_hook_error() {
    thread_local std::exception_ptr e;
    return e;
}
void feature_ready_trampoline(void* ctx, void* /*stream*/) {
    py::object& hook = feature_grad_hook();
    if (hook.is_none() || pending_hook_error()) return;
    try {
        hook(*static_cast<torch::Tensor*>(ctx));
    } catch (...) {
        pending_hook_error() = std::current_exception();
    }
}
// clears the thread-local callback registration on every way out of the backward binding
struct FeatureCallbackGuard {
    bool armed;
    explicit FeatureCallbackGuard(bool on, void* ctx) : armed(on) {
        if (armed) f3dgs_set_feature_grad_ready_callback(feature_ready_trampoline, ctx);
    }
    ~FeatureCallbackGuard() {
        if (armed) f3dgs_set_feature_grad_ready_callback(nullptr, nullptr);
    }
};

// Accumulation buffer for the feature gradient across the views of one optimiser step (include/f3dgs.h:
// f3dgs_set_feature_grad_accumulate): while it is set, the backward binding hands IT to the library as
// dL_dsemantic_feature (accumulate mode) and returns an empty tensor in its place - the Python side then reports "no
// gradient" for the input to autograd, the sum lives in the buffer (normally the leaf's .grad).
torch::Tensor& feature_grad_accumulator() {
    static torch::Tensor* t = new torch::Tensor();
    return *t;
}
// set_feature_grad_lowres: (gx (Hg,Wg,C), scale (0-dim) or undefined), consumed by the next backward call
std::pair<torch::Tensor, torch::Tensor>& feature_grad_lowres() {
    static std::pair<torch::Tensor, torch::Tensor> t;
    return t;
}

struct AccumulateGuard {
    bool armed;
    explicit AccumulateGuard(bool on) : armed(on) { if (armed) f3dgs_set_feature_grad_accumulate(1); }
    ~AccumulateGuard() { if (armed) f3dgs_set_feature_grad_accumulate(0); }
};

// Python callable invoked as fn(row_begin, row_end, {"sh": dL_dsh, "means3D": ..., ...}) after every row chunk of the
// per-Gaussian stage has been enqueued (include/f3dgs.h: f3dgs_set_grad_rows_ready_callback), and its chunk count.
py::object& grad_rows_hook() {
    static py::object* hook = new py::object(py::none());
    return *hook;
}
int& grad_rows_chunks() {
    static int chunks = 1;
    return chunks;
}
struct RowsCallbackCtx {
    const torch::Tensor *sh, *means3D, *scales, *rotations, *opacities, *colors, *means2D, *cov3D;
};
void rows_ready_trampoline(void* ctx, void* /*stream*/, int row_begin, int row_end) {
    py::object& hook = grad_rows_hook();
    if (hook.is_none() || pending_hook_error()) return;
    try {
        const auto* c = static_cast<const RowsCallbackCtx*>(ctx);
        py::dict grads;
        grads["sh"] = *c->sh; grads["means3D"] = *c->means3D; grads["scales"] = *c->scales; grads["rotations"] = *c->rotations;
        grads["opacities"] = *c->opacities; grads["colors_precomp"] = *c->colors; grads["means2D"] = *c->means2D;
        grads["cov3Ds_precomp"] = *c->cov3D;
        hook(row_begin, row_end, grads);
    } catch (...) {
        pending_hook_error() = std::current_exception();
    }
}
struct RowsCallbackGuard {
    bool armed;
    RowsCallbackGuard(bool on, void* ctx, int chunks) : armed(on) {
        if (armed) f3dgs_set_grad_rows_ready_callback(rows_ready_trampoline, ctx, chunks);
    }
    ~RowsCallbackGuard() {
        if (armed) f3dgs_set_grad_rows_ready_callback(nullptr, nullptr, 1);
    }
};

void* current_stream(const torch::Tensor& ref) {
    return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(ref.device().index()).stream();
}

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussians(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                   const torch::Tensor& semantic_feature, const torch::Tensor& opacity, const torch::Tensor& scales,
                   const torch::Tensor& rotations, const float scale_modifier, const torch::Tensor& cov3D_precomp,
                   const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                   const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh,
                   const int degree, const torch::Tensor& campos, const bool prefiltered, const bool debug) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) {
        AT_ERROR("means3D must have dimensions (num_points, 3)");  // rasterize_points.cu:58-60
    }
    TORCH_CHECK(means3D.is_cuda(), "means3D must live on a HIP device: the MI355X rasterizer has no CPU path");
    const int P = means3D.size(0);
    const int H = image_height, W = image_width;
    const int C = feature_channels(semantic_feature, P);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());

    auto f32 = means3D.options().dtype(torch::kFloat32);
    torch::Tensor out_color = torch::empty({3, H, W}, f32);
    torch::Tensor out_depth = torch::empty({1, H, W}, f32);
    torch::Tensor out_feature_map = torch::empty({C, H, W}, f32);
    torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
    auto u8 = means3D.options().dtype(torch::kByte);
    torch::Tensor geomBuffer = torch::empty({0}, u8);
    torch::Tensor binningBuffer = torch::empty({0}, u8);
    torch::Tensor imgBuffer = torch::empty({0}, u8);

    auto bg = dev_f32(background, "bg"), m3 = dev_f32(means3D, "means3D"), col = dev_f32(colors, "colors_precomp"),
         sf = dev_f32(semantic_feature, "semantic_feature"), op = dev_f32(opacity, "opacities"),
         sc = dev_f32(scales, "scales"), rot = dev_f32(rotations, "rotations"), cov = dev_f32(cov3D_precomp, "cov3D_precomp"),
         vm = dev_f32(viewmatrix, "viewmatrix"), pm = dev_f32(projmatrix, "projmatrix"), shs = dev_f32(sh, "sh"),
         cp = dev_f32(campos, "campos");
    int M = 0;
    if (shs.numel() != 0) M = shs.size(1);

    int rendered = 0;
    const int rc = f3dgs_forward(resize_hook, &geomBuffer, resize_hook, &binningBuffer, resize_hook, &imgBuffer, P, degree,
                                 M, C, fptr(bg), W, H, fptr(m3), fptr(shs), fptr(col), fptr(sf), fptr(op), fptr(sc),
                                 scale_modifier, fptr(rot), fptr(cov), fptr(vm), fptr(pm), fptr(cp), tan_fovx, tan_fovy,
                                 prefiltered ? 1 : 0, out_color.data_ptr<float>(),
                                 C ? out_feature_map.data_ptr<float>() : nullptr, out_depth.data_ptr<float>(),
                                 P ? radii.data_ptr<int>() : nullptr, debug ? 1 : 0, current_stream(means3D), &rendered);
    check_status(rc, "rasterize_gaussians");
    return std::make_tuple(rendered, out_color, out_feature_map, out_depth, radii, geomBuffer, binningBuffer, imgBuffer);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor>
RasterizeGaussiansBackward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                           const torch::Tensor& colors, const torch::Tensor& semantic_feature,
                           const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier,
                           const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                           const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                           const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_feature,
                           const torch::Tensor& dL_dout_depth, const torch::Tensor& sh, const int degree,
                           const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                           const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug) {
    TORCH_CHECK(means3D.is_cuda(), "means3D must live on a HIP device: the MI355X rasterizer has no CPU path");
    const int P = means3D.size(0);
    const int H = dL_dout_color.size(1), W = dL_dout_color.size(2);  // rasterize_points.cu:154-155
    const int C = feature_channels(semantic_feature, P);
    const int F1 = 1;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());

    auto shs = dev_f32(sh, "sh");
    int M = 0;
    if (shs.numel() != 0) M = shs.size(1);

    auto o = means3D.options().dtype(torch::kFloat32);
    torch::Tensor dL_dmeans3D = torch::empty({P, 3}, o);
    torch::Tensor dL_dmeans2D = torch::empty({P, 3}, o);
    torch::Tensor dL_dcolors = torch::empty({P, 3}, o);
    torch::Tensor& acc = feature_grad_accumulator();
    const bool accumulate = acc.defined() && P > 0 && C > 0;
    if (accumulate) {
        TORCH_CHECK(acc.is_cuda() && acc.device() == means3D.device() && acc.scalar_type() == torch::kFloat32 && acc.is_contiguous() &&
                    acc.numel() == (int64_t)P * C,
                    "feature gradient accumulator must be a contiguous float32 tensor of ", P, " x ", C, " elements on the op's device");
    }
    torch::Tensor dL_dsemantic_feature = accumulate ? acc.view({P, F1, C}) : torch::empty({P, F1, C}, o);
    torch::Tensor dL_dopacity = torch::empty({P, 1}, o);
    torch::Tensor dL_dcov3D = torch::empty({P, 6}, o);
    torch::Tensor dL_dsh = torch::empty({P, M, 3}, o);
    torch::Tensor dL_dscales = torch::empty({P, 3}, o);
    torch::Tensor dL_drotations = torch::empty({P, 4}, o);
    auto sc = dev_f32(scales, "scales"), rot = dev_f32(rotations, "rotations");
    if (sc.numel() == 0) {  // cov3D_precomp path: these grads are defined as zero (rasterize_points.cu:171-172)
        dL_dscales.zero_();
        dL_drotations.zero_();
    }
    torch::Tensor scratch = torch::empty({(long long)f3dgs_backward_scratch_bytes(P, C)}, means3D.options().dtype(torch::kByte));

    auto bg = dev_f32(background, "bg"), m3 = dev_f32(means3D, "means3D"), col = dev_f32(colors, "colors_precomp"),
         sf = dev_f32(semantic_feature, "semantic_feature"), cov = dev_f32(cov3D_precomp, "cov3D_precomp"),
         vm = dev_f32(viewmatrix, "viewmatrix"), pm = dev_f32(projmatrix, "projmatrix"), cp = dev_f32(campos, "campos"),
         gc = dev_f32(dL_dout_color, "dL_dout_color"), gd = dev_f32(dL_dout_depth, "dL_dout_depth");
    // The feature-map gradient at the loss's resolution (feature_loss.py, lowres_grad=True): one call only.  What autograd hands
    // over as dL_dout_feature is then the loss's placeholder - a zero-stride expansion of one 0, never materialised - unless
    // another consumer of the feature map contributed a dense gradient, which the kernel adds.
    std::pair<torch::Tensor, torch::Tensor> low;
    std::swap(low, feature_grad_lowres());
    const bool lowres = low.first.defined() && P > 0 && C > 0;
    bool dense_feature_grad = true;
    if (lowres) {
        const auto& gx = low.first;
        TORCH_CHECK(gx.is_cuda() && gx.device() == means3D.device() && gx.scalar_type() == torch::kFloat32 && gx.is_contiguous() &&
                    gx.dim() == 3 && gx.size(2) == C, "low-resolution feature-map gradient must be a contiguous float32 (Hg, Wg, ", C,
                    ") tensor on the op's device");
        TORCH_CHECK(gx.numel() < (1ll << 31), "low-resolution feature-map gradient too large");
        if (low.second.defined())
            TORCH_CHECK(low.second.is_cuda() && low.second.scalar_type() == torch::kFloat32 && low.second.numel() == 1,
                        "the scale of the low-resolution gradient must be a float32 scalar on the device");
        bool all_zero_stride = dL_dout_feature.dim() > 0;
        for (int64_t d = 0; d < dL_dout_feature.dim(); d++) all_zero_stride = all_zero_stride && dL_dout_feature.stride(d) == 0;
        dense_feature_grad = !(dL_dout_feature.numel() == 0 || all_zero_stride);
    }
    auto gf = dense_feature_grad ? dev_f32(dL_dout_feature, "dL_dout_feature") : torch::Tensor();
    auto gfptr = [&]() -> const float* { return dense_feature_grad ? fptr(gf) : nullptr; };
    TORCH_CHECK(radii.is_cuda() || P == 0, "radii must live on a HIP device");
    auto rad = radii.contiguous();

    const bool notify = !feature_grad_hook().is_none() && P > 0 && C > 0;
    pending_hook_error() = nullptr;
    int rc;
    {
    AccumulateGuard guard_acc(accumulate);
    FeatureCallbackGuard guard_cb(notify, &dL_dsemantic_feature);
    RowsCallbackCtx rows_ctx = {&dL_dsh, &dL_dmeans3D, &dL_dscales, &dL_drotations, &dL_dopacity, &dL_dcolors, &dL_dmeans2D, &dL_dcov3D};
    RowsCallbackGuard guard_rows(!grad_rows_hook().is_none() && P > 0, &rows_ctx, grad_rows_chunks());
    if (lowres)
        check_status(f3dgs_set_feature_grad_lowres(low.first.data_ptr<float>(), (int)low.first.size(0), (int)low.first.size(1),
                                                   low.second.defined() ? low.second.data_ptr<float>() : nullptr),
                     "set_feature_grad_lowres");
    rc = f3dgs_backward(
        P, degree, M, C, R, fptr(bg), W, H, fptr(m3), fptr(shs), fptr(col), fptr(sf), fptr(sc), scale_modifier, fptr(rot),
        fptr(cov), fptr(vm), fptr(pm), fptr(cp), tan_fovx, tan_fovy, P ? rad.data_ptr<int>() : nullptr,
        reinterpret_cast<const char*>(geomBuffer.data_ptr()), reinterpret_cast<const char*>(binningBuffer.data_ptr()),
        reinterpret_cast<const char*>(imageBuffer.data_ptr()), fptr(gc), gfptr(), fptr(gd),
        P ? dL_dmeans2D.data_ptr<float>() : nullptr, nullptr, P ? dL_dopacity.data_ptr<float>() : nullptr,
        P ? dL_dcolors.data_ptr<float>() : nullptr, (P && C) ? dL_dsemantic_feature.data_ptr<float>() : nullptr,
        P ? dL_dmeans3D.data_ptr<float>() : nullptr, P ? dL_dcov3D.data_ptr<float>() : nullptr,
        (P && M) ? dL_dsh.data_ptr<float>() : nullptr, P ? dL_dscales.data_ptr<float>() : nullptr,
        P ? dL_drotations.data_ptr<float>() : nullptr, nullptr, P ? scratch.data_ptr() : nullptr, debug ? 1 : 0,
        current_stream(means3D));
    }
    if (pending_hook_error()) {
        std::exception_ptr e = pending_hook_error();
        pending_hook_error() = nullptr;
        std::rethrow_exception(e);
    }
    check_status(rc, "rasterize_gaussians_backward");
    // accumulate mode: the sum lives in the caller's buffer; an empty tensor tells the Python side "no gradient to hand on"
    return std::make_tuple(dL_dmeans2D, dL_dcolors, accumulate ? torch::empty({0}, o) : dL_dsemantic_feature, dL_dopacity, dL_dmeans3D,
                           dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix) {
    TORCH_CHECK(means3D.is_cuda(), "means3D must live on a HIP device: the MI355X rasterizer has no CPU path");
    const int P = means3D.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
    torch::Tensor present = torch::empty({P}, means3D.options().dtype(at::kBool));
    if (P != 0) {
        auto m3 = dev_f32(means3D, "means3D"), vm = dev_f32(viewmatrix, "viewmatrix"), pm = dev_f32(projmatrix, "projmatrix");
        const int rc = f3dgs_mark_visible(P, fptr(m3), fptr(vm), fptr(pm), reinterpret_cast<uint8_t*>(present.data_ptr<bool>()),
                                          current_stream(means3D));
        check_status(rc, "mark_visible");
    }
    return present;
}

// fused resize -> 1x1 decoder -> L1 (include/f3dgs.h: f3dgs_feature_l1); returns (loss, d_feature_map, d_weight, d_bias, gx).
// dense = false: d_feature_map is not produced (empty); gx = dL/d(resized map), (Hg, Wg, C), a view of the call's scratch,
// is what set_feature_grad_lowres hands to the rasterizer's backward.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
FeatureL1(const torch::Tensor& feature_map, const torch::Tensor& gt, const torch::Tensor& weight, const torch::Tensor& bias, bool dense) {
    TORCH_CHECK(feature_map.is_cuda() && gt.is_cuda(), "feature_l1: tensors must live on a HIP device (no CPU path)");
    TORCH_CHECK(feature_map.dim() == 3 && gt.dim() == 3, "feature_l1: feature_map (C,H,W) and gt (Cout,Hg,Wg) expected");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(feature_map.device());
    auto fm = dev_f32(feature_map, "feature_map"), g = dev_f32(gt, "gt_feature_map");
    const bool dec = weight.numel() > 0;
    torch::Tensor w = weight, b = bias;
    if (dec) {
        w = dev_f32(weight, "weight");
        b = dev_f32(bias, "bias");
        TORCH_CHECK(w.dim() == 2 && w.size(1) == fm.size(0) && w.size(0) == g.size(0) && b.numel() == g.size(0),
                    "feature_l1: weight (Cout,C) / bias (Cout) do not match feature_map / gt");
    }
    const int C = fm.size(0), H = fm.size(1), W = fm.size(2), Cout = g.size(0), Hg = g.size(1), Wg = g.size(2);
    auto o = fm.options();
    torch::Tensor loss = torch::empty({}, o), d_fm = dense ? torch::empty_like(fm) : torch::empty({0}, o);
    torch::Tensor d_w = dec ? torch::empty_like(w) : torch::empty({0}, o), d_b = dec ? torch::empty_like(b) : torch::empty({0}, o);
    torch::Tensor scratch = torch::empty({(long long)f3dgs_feature_l1_scratch_bytes(C, Cout, Hg, Wg, dec ? 1 : 0)},
                                         o.dtype(torch::kByte));
    const int rc = f3dgs_feature_l1(C, H, W, Cout, Hg, Wg, fm.data_ptr<float>(), dec ? w.data_ptr<float>() : nullptr,
                                    dec ? b.data_ptr<float>() : nullptr, g.data_ptr<float>(), loss.data_ptr<float>(),
                                    dense ? d_fm.data_ptr<float>() : nullptr, dec ? d_w.data_ptr<float>() : nullptr,
                                    dec ? d_b.data_ptr<float>() : nullptr, scratch.data_ptr(), current_stream(fm));
    check_status(rc, "feature_l1");
    const float* gxp = f3dgs_feature_l1_lowres_grad(C, Cout, Hg, Wg, dec ? 1 : 0, scratch.data_ptr());
    TORCH_CHECK(gxp != nullptr, "feature_l1: no low-resolution gradient");
    const int64_t off = reinterpret_cast<const char*>(gxp) - reinterpret_cast<const char*>(scratch.data_ptr());
    torch::Tensor gx = scratch.slice(0, off, off + (int64_t)Hg * Wg * C * 4).view(torch::kFloat32).view({Hg, Wg, C});
    return std::make_tuple(loss, d_fm, d_w, d_b, gx);
}

// forward-only resize -> 1x1 decoder (include/f3dgs.h: f3dgs_feature_decode); returns the (Cout, Hg, Wg) map, fp32 or fp16
torch::Tensor FeatureDecode(const torch::Tensor& feature_map, int64_t Hg, int64_t Wg, const torch::Tensor& weight,
                            const torch::Tensor& bias, bool half) {
    TORCH_CHECK(feature_map.is_cuda(), "feature_decode: tensors must live on a HIP device (no CPU path)");
    TORCH_CHECK(feature_map.dim() == 3, "feature_decode: feature_map (C,H,W) expected");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(feature_map.device());
    auto fm = dev_f32(feature_map, "feature_map");
    const bool dec = weight.numel() > 0;
    torch::Tensor w = weight, b = bias;
    const int C = fm.size(0), H = fm.size(1), W = fm.size(2);
    int Cout = C;
    if (dec) {
        w = dev_f32(weight, "weight");
        b = dev_f32(bias, "bias");
        TORCH_CHECK(w.dim() == 2 && w.size(1) == C && b.numel() == w.size(0), "feature_decode: weight (Cout,C) / bias (Cout) do not match feature_map");
        Cout = w.size(0);
    }
    torch::Tensor out = torch::empty({Cout, Hg, Wg}, fm.options().dtype(half ? torch::kFloat16 : torch::kFloat32));
    torch::Tensor scratch = torch::empty({(long long)f3dgs_feature_decode_scratch_bytes(C, (int)Hg, (int)Wg, dec ? 1 : 0)}, fm.options().dtype(torch::kByte));
    const int rc = f3dgs_feature_decode(C, H, W, Cout, (int)Hg, (int)Wg, fm.data_ptr<float>(), dec ? w.data_ptr<float>() : nullptr,
                                        dec ? b.data_ptr<float>() : nullptr, out.data_ptr(), half ? 1 : 0,
                                        dec ? scratch.data_ptr() : nullptr, current_stream(fm));
    check_status(rc, "feature_decode");
    return out;
}

void AdamStep(torch::Tensor& param, const torch::Tensor& grad, torch::Tensor& exp_avg, torch::Tensor& exp_avg_sq, double lr,
              double beta1, double beta2, double eps, int64_t step, const c10::optional<torch::Tensor>& row_mask) {
    TORCH_CHECK(param.is_cuda() && grad.is_cuda() && exp_avg.is_cuda() && exp_avg_sq.is_cuda(), "adam_step: HIP tensors only");
    TORCH_CHECK(param.scalar_type() == torch::kFloat32 && grad.scalar_type() == torch::kFloat32, "adam_step: float32 only");
    TORCH_CHECK(param.is_contiguous() && exp_avg.is_contiguous() && exp_avg_sq.is_contiguous(), "adam_step: contiguous state");
    TORCH_CHECK(grad.numel() == param.numel() && exp_avg.numel() == param.numel() && exp_avg_sq.numel() == param.numel(),
                "adam_step: size mismatch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(param.device());
    auto g = grad.contiguous();
    const uint8_t* mask = nullptr;
    size_t width = 1;
    torch::Tensor mk;
    if (row_mask.has_value() && row_mask->defined()) {
        TORCH_CHECK(param.dim() >= 1 && row_mask->numel() == param.size(0), "adam_step: one mask entry per row");
        TORCH_CHECK(row_mask->is_cuda(), "adam_step: the mask must be a HIP tensor");
        mk = row_mask->to(torch::kUInt8).contiguous();      // bool / uint8 / int masks alike
        mask = mk.data_ptr<uint8_t>();
        width = param.size(0) ? (size_t)(param.numel() / param.size(0)) : 1;
    }
    check_status(f3dgs_adam_step_rows((size_t)param.numel(), width, mask, param.data_ptr<float>(), g.data_ptr<float>(),
                                      exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(), lr, beta1, beta2, eps, (int)step,
                                      current_stream(param)),
                 "adam_step");
}

// torch.optim.Adam's step over a list of tensors in one launch (include/f3dgs.h: f3dgs_adam_step_multi).
void AdamStepMulti(std::vector<torch::Tensor> params, std::vector<torch::Tensor> grads, std::vector<torch::Tensor> exp_avgs,
                   std::vector<torch::Tensor> exp_avg_sqs, std::vector<double> lrs, double beta1, double beta2, double eps,
                   std::vector<int64_t> steps, const c10::optional<torch::Tensor>& row_mask) {
    const size_t n = params.size();
    TORCH_CHECK(n <= F3DGS_ADAM_MAX_TENSORS, "adam_step_multi: at most ", F3DGS_ADAM_MAX_TENSORS, " tensors per call");
    TORCH_CHECK(grads.size() == n && exp_avgs.size() == n && exp_avg_sqs.size() == n && lrs.size() == n && steps.size() == n,
                "adam_step_multi: list lengths differ");
    if (n == 0) return;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(params[0].device());
    std::vector<torch::Tensor> keep;       // contiguous copies of gradients stay alive until the launch is enqueued
    f3dgs_adam_tensor tab[F3DGS_ADAM_MAX_TENSORS];
    for (size_t i = 0; i < n; i++) {
        TORCH_CHECK(params[i].is_cuda() && grads[i].is_cuda() && exp_avgs[i].is_cuda() && exp_avg_sqs[i].is_cuda(), "adam_step_multi: HIP tensors only");
        TORCH_CHECK(params[i].scalar_type() == torch::kFloat32 && grads[i].scalar_type() == torch::kFloat32, "adam_step_multi: float32 only");
        TORCH_CHECK(params[i].is_contiguous() && exp_avgs[i].is_contiguous() && exp_avg_sqs[i].is_contiguous(), "adam_step_multi: contiguous state");
        TORCH_CHECK(grads[i].numel() == params[i].numel() && exp_avgs[i].numel() == params[i].numel() &&
                    exp_avg_sqs[i].numel() == params[i].numel(), "adam_step_multi: size mismatch");
        keep.push_back(grads[i].contiguous());
        tab[i] = f3dgs_adam_tensor{params[i].data_ptr<float>(), keep.back().data_ptr<float>(), exp_avgs[i].data_ptr<float>(),
                                   exp_avg_sqs[i].data_ptr<float>(), (size_t)params[i].numel(), lrs[i], (int)steps[i]};
    }
    const uint8_t* mask = nullptr;
    size_t rows = 0;
    torch::Tensor mk;
    if (row_mask.has_value() && row_mask->defined()) {
        TORCH_CHECK(row_mask->is_cuda(), "adam_step_multi: the mask must be a HIP tensor");
        mk = row_mask->to(torch::kUInt8).contiguous();
        mask = mk.data_ptr<uint8_t>();
        rows = (size_t)mk.numel();
    }
    check_status(f3dgs_adam_step_multi((int)n, tab, beta1, beta2, eps, mask, rows, current_stream(params[0])), "adam_step_multi");
}

// One gather over all per-Gaussian tensors (densify.py builds the plan).  `modes[i]`: F3DGS_DENSIFY_*;
// `overrides[i]` is the child-row array for mode OVERRIDE_CHILD (an empty tensor otherwise).
void DensifyGather(const torch::Tensor& src_row, const torch::Tensor& kind, const torch::Tensor& override_row,
                   const std::vector<torch::Tensor>& srcs, std::vector<torch::Tensor>& dsts,
                   const std::vector<torch::Tensor>& overrides, const std::vector<int64_t>& modes) {
    const size_t n_out = (size_t)src_row.numel();
    TORCH_CHECK(src_row.is_cuda() && src_row.scalar_type() == torch::kInt32 && src_row.is_contiguous(), "densify_gather: src_row must be a contiguous int32 HIP tensor");
    TORCH_CHECK(kind.is_cuda() && kind.scalar_type() == torch::kUInt8 && kind.is_contiguous() && (size_t)kind.numel() == n_out, "densify_gather: kind must be uint8 of the same length");
    TORCH_CHECK(override_row.is_cuda() && override_row.scalar_type() == torch::kInt32 && override_row.is_contiguous() && (size_t)override_row.numel() == n_out,
                "densify_gather: override_row must be int32 of the same length");
    TORCH_CHECK(srcs.size() == dsts.size() && srcs.size() == modes.size() && srcs.size() == overrides.size(), "densify_gather: list lengths differ");
    std::vector<f3dgs_densify_tensor> table(srcs.size());
    for (size_t i = 0; i < srcs.size(); i++) {
        const auto& a = srcs[i];
        auto& d = dsts[i];
        TORCH_CHECK(a.is_cuda() && d.is_cuda() && a.scalar_type() == torch::kFloat32 && d.scalar_type() == torch::kFloat32, "densify_gather: float32 HIP tensors only");
        TORCH_CHECK(a.is_contiguous() && d.is_contiguous(), "densify_gather: contiguous tensors only");
        int64_t width = 1;
        for (int64_t k = 1; k < a.dim(); k++) width *= a.size(k);
        TORCH_CHECK(a.dim() >= 1 && width >= 1, "densify_gather: tensors are (rows, ...) with non-empty rows");
        TORCH_CHECK((size_t)d.numel() >= n_out * (size_t)width, "densify_gather: destination ", i, " too small");
        table[i].src = a.data_ptr<float>();
        table[i].dst = d.data_ptr<float>();
        table[i].width = (int)width;
        table[i].mode = (int)modes[i];
        table[i].override_src = nullptr;
        if (modes[i] == F3DGS_DENSIFY_OVERRIDE_CHILD) {
            const auto& o = overrides[i];
            TORCH_CHECK(o.is_cuda() && o.scalar_type() == torch::kFloat32 && o.is_contiguous(), "densify_gather: override must be a contiguous float32 HIP tensor");
            table[i].override_src = o.numel() ? o.data_ptr<float>() : a.data_ptr<float>();   // no children: never read
        }
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(src_row.device());
    check_status(f3dgs_densify_gather(n_out, src_row.data_ptr<int32_t>(), kind.data_ptr<uint8_t>(), override_row.data_ptr<int32_t>(),
                                      (int)table.size(), table.data(), current_stream(src_row)),
                 "densify_gather");
}

}  // namespace

PYBIND11_MODULE(_C, m) {
    m.def("rasterize_gaussians", &RasterizeGaussians);
    m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackward);
    m.def("mark_visible", &markVisible);
    m.def("feature_l1", &FeatureL1, py::arg("feature_map"), py::arg("gt"), py::arg("weight"), py::arg("bias"), py::arg("dense") = true);
    m.def("set_feature_grad_lowres", [](py::object gx, py::object scale) {
              auto& h = feature_grad_lowres();
              h.first = gx.is_none() ? torch::Tensor() : gx.cast<torch::Tensor>();
              h.second = (gx.is_none() || scale.is_none()) ? torch::Tensor() : scale.cast<torch::Tensor>();
          },
          py::arg("gx"), py::arg("scale") = py::none(),
          "gx (Hg, Wg, C) float32 = dL/d(resized feature map) as feature_l1 returns it, scale = 0-dim device tensor or None: the NEXT "
          "rasterize_gaussians_backward call takes its feature-map gradient from there (transposed resize applied per tile); None clears");
    m.def("feature_decode", &FeatureDecode);
    m.def("adam_step", &AdamStep, py::arg("param"), py::arg("grad"), py::arg("exp_avg"), py::arg("exp_avg_sq"), py::arg("lr"),
          py::arg("beta1"), py::arg("beta2"), py::arg("eps"), py::arg("step"), py::arg("row_mask") = py::none());
    m.def("adam_step_multi", &AdamStepMulti, py::arg("params"), py::arg("grads"), py::arg("exp_avgs"), py::arg("exp_avg_sqs"), py::arg("lrs"),
          py::arg("beta1"), py::arg("beta2"), py::arg("eps"), py::arg("steps"), py::arg("row_mask") = py::none());
    m.def("densify_gather", &DensifyGather);
    m.def("version", []() { return f3dgs_version(); });
    m.def("set_feature_grad_hook", [](py::object fn) { feature_grad_hook() = std::move(fn); },
          "callable(dL_dsemantic_feature) run inside rasterize_gaussians_backward once that tensor is final on the stream; None removes it");
    m.def("set_feature_grad_accumulator", [](py::object t) {
              if (t.is_none()) feature_grad_accumulator() = torch::Tensor();
              else feature_grad_accumulator() = t.cast<torch::Tensor>();
          },
          "tensor (P*C float32, contiguous, on the op's device) that the following backward calls ADD dL/dsemantic_feature into "
          "(they then return an empty tensor for that gradient); None restores the default");
    m.def("set_grad_rows_hook", [](py::object fn, int chunks) { grad_rows_hook() = std::move(fn); grad_rows_chunks() = chunks > 0 ? chunks : 1; },
          py::arg("fn"), py::arg("chunks") = 4,
          "callable(row_begin, row_end, grads: dict) run inside rasterize_gaussians_backward after each of `chunks` row ranges of the "
          "per-Gaussian gradients is final on the stream; None removes it");
    m.def("set_option", [](const std::string& name, int value) { check_status(f3dgs_set_option(name.c_str(), value), "set_option"); });
    m.def("set_tile_band", [](int row_begin, int row_end) { f3dgs_set_tile_band(row_begin, row_end); }, py::arg("tile_row_begin"), py::arg("tile_row_end"),
          "forward calls of this thread list and blend only tile rows [begin, end) of the view (0, 0: the whole view); include/f3dgs.h");
    m.def("forward_counts", []() -> py::object {
        const uint32_t* w = f3dgs_forward_counts();
        if (!w) return py::none();
        // (volatile: kernel-written pinned words)
        const volatile uint32_t* v = w;
        return py::make_tuple((long long)v[0], (long long)v[1], (long long)v[2], (long long)v[3], (long long)v[4]);
    }, "(entries of the instance lists, the reference's num_rendered, long-axis flag, entries provided for, sticky no-room word) of the "
       "calling thread's most recent forward call - final once that frame's work has completed; None before the first call");
    m.def("forward_counts_address", []() { return (uintptr_t)f3dgs_forward_counts(); },
          "host address of the five uint32 words of forward_counts() (0 before the first call): a captured step keeps it and reads the words after a replay");
    m.def("clear_forward_overflow", [](uintptr_t address) { if (address) reinterpret_cast<volatile uint32_t*>(address)[4] = 0u; },
          "clears the sticky no-room word behind an address returned by forward_counts_address()");
    m.def("last_backward_contraction", []() { return f3dgs_last_backward_contraction(); },
          "1: the last blend backward of this process contracted on bf16 matrix instructions (two-term operands); 0: exact fp32; -1: none yet");
    m.def("get_option", [](const std::string& name) {
        int v = 0;
        check_status(f3dgs_get_option(name.c_str(), &v), "get_option");
        return v;
    });
    m.def("profile_reset", []() { f3dgs_profile_reset(); });
    m.def("profile_read", []() {
        const char* names[64];
        double ms[64];
        long calls[64];
        const int n = f3dgs_profile_read(names, ms, calls, 64);
        std::vector<std::tuple<std::string, double, long>> out;
        for (int i = 0; i < n; i++) out.emplace_back(names[i], ms[i], calls[i]);
        return out;
    });
}
