// render_common.h — pieces shared by the forward and backward blend kernels.
//
// Work decomposition (both kernels): one 16x16 tile per workgroup; the tile is cut into four 8x8
// quadrants.  A wave (64 lanes) owns PPL quadrants (PPL = pixels per lane in {1,2,4}) and walks the
// tile's depth-sorted instance list autonomously in wavefront-sized chunks of 64 instances that it
// stages into its private LDS slice — no workgroup barrier anywhere, waves retire independently as
// soon as their own pixels are finished (ballot-based early-out).  Lane l of a wave handles pixel
// (l & 7, l >> 3) of each of its quadrants, so a wave-uniform ballot per quadrant skips the whole
// blend body for splats that miss the 8x8 block.
#pragma once

#include "common.h"

namespace f3dgs {

// Gaussian evaluation shared by forward and backward so both take identical skip decisions
// (R/forward.cu:340-353, R/backward.cu:525-535).  The blend kernels scale the conic once per staged instance
// (a' = -a log2(e) / 2, b' = -b log2(e), c' = -c log2(e) / 2) so that the exponent of 2 costs five instructions per
// (pixel, instance) pair instead of eight:  power log2(e) = (a' dx + b' dy) dx + (c' dy) dy,  G = exp2 of it
// (v_exp_f32 is a base-2 exponential).  Explicit fmaf keeps the instruction sequence - and with it every
// alpha >= 1/255 and power > 0 decision - identical in the forward and the backward kernel.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float CONIC_SCALE_AC = -0.5f * LOG2E, CONIC_SCALE_B = -LOG2E;
constexpr float CONIC_UNSCALE_AC = 1.0f / CONIC_SCALE_AC, CONIC_UNSCALE_B = 1.0f / CONIC_SCALE_B;
__device__ __forceinline__ float splat_power2(float dx, float dy, float ca2, float cb2, float cc2) {
    return fmaf(fmaf(ca2, dx, cb2 * dy), dx, (cc2 * dy) * dy);
}

constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_MIN = 0.0001f;

// ---- DPP wave reduction (gfx9 row_bcast forms; result valid in lane 63) ---------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_get<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(v);       // row_half_mirror
    v += dpp_get<0x140>(v);       // row_mirror      -> every lane holds its row's sum
    v += dpp_get<0x142, 0xA>(v);  // row_bcast:15    -> rows 1,3 += previous row
    v += dpp_get<0x143, 0xC>(v);  // row_bcast:31    -> rows 2,3 += lane 31
    return v;
}
__device__ __forceinline__ float wave_sum_uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_lane63(v)), 63));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// Wave-uniform floats without a vector register: float arithmetic is vector-pipe work on this chip, so a uniform value computed in
// float (a tile's pixel rectangle, the pixel scales) occupies a vector register for as long as it lives - across the whole list
// walk - and is the first thing spilled.  The INTEGER it is made from lives in a scalar register; it is converted where it is
// used, and this empty asm makes the integer opaque there so that the conversion is not hoisted back out of the loop.  (A
// v_readfirstlane in inline asm was tried first: the compiler may place or duplicate such an asm where it likes and knows
// nothing of the hazards of the instruction inside - correct in two kernels, wrong in a third.)
__device__ __forceinline__ int sgpr_opaque(int x) {
    asm volatile("" : "+s"(x));
    return x;
}

// Exact-safe footprint test of one splat against a pixel-centre rectangle [x0, x1] x [y0, y1]: the splat can
// reach alpha >= 1/255 at some pixel of the rectangle only if the minimum of its (convex) quadratic form over
// the rectangle - 0 if the mean lies inside, otherwise attained on one of the four edges at the clamped 1-D
// minimiser - is below 2 ln(255 opacity); a safety margin far above fp32 round-off keeps every borderline splat.
__device__ __forceinline__ bool rect_hit(float mx, float my, float ca, float cb, float cc, float opacity, float x0,
                                         float x1, float y0, float y1) {
    const float s = 255.0f * opacity;
    if (!(s > 1.0f)) return false;
    // + 32 eps * kappa: fp32 round-off of the quadratic form of needle-shaped splats (see make_cull, preprocess.hip)
    const float kappa = fminf(fmaxf((ca * cc) * __builtin_amdgcn_rcpf(ca * cc - cb * cb), 1.0f), 1e6f);
    const float thresh = 2.0f * (__logf(s) * 1.0001f + 2e-3f) * (1.0f + 2e-6f * kappa);
    const float xlo = mx - x1, xhi = mx - x0, ylo = my - y1, yhi = my - y0;
    if (xlo <= 0.0f && xhi >= 0.0f && ylo <= 0.0f && yhi >= 0.0f) return true;
    const float ia = __builtin_amdgcn_rcpf(ca), ic = __builtin_amdgcn_rcpf(cc);
    auto qf = [&](float dx, float dy) { return (ca * dx) * dx + 2.0f * ((cb * dx) * dy) + (cc * dy) * dy; };
    float best = qf(xlo, fminf(yhi, fmaxf(ylo, -(cb * xlo) * ic)));
    best = fminf(best, qf(xhi, fminf(yhi, fmaxf(ylo, -(cb * xhi) * ic))));
    best = fminf(best, qf(fminf(xhi, fmaxf(xlo, -(cb * ylo) * ia)), ylo));
    best = fminf(best, qf(fminf(xhi, fmaxf(xlo, -(cb * yhi) * ia)), yhi));
    return !(best > thresh * 1.001f);   // NaN keeps the splat; extra 0.1 % for the approximate log / rcp
}

// Arguments of the blend backward kernels (render_bwd.hip: instance-lane formulation; render_bwd_pl.hip: pixel-lane
// formulation).
struct BwdArgs {
    const uint2* ranges;
    const uint32_t* point_list;
    const SplatRec* rec;
    const float* bg;
    const float* final_T;
    const uint32_t* n_contrib;
    const float* dL_dpix;
    const float* dL_dfeat;
    const float* dL_ddepth;
    float* grec;         // P x GREC
    float* dL_dfeature;  // P x C
    int W, H, gx, gy;
    int C, c0, nc;
    int write_base;  // 1: also accumulate the 10 geometric sums (first channel window only)
    int half;         // instance-lane kernel: chunks of 32 instances, the two halves of the wave take different pixels
    const uint32_t* order;   // workgroup -> tile, longest walk first (null: XCD-contiguous tile order)
    int m44;                 // pixel-lane kernel, first window: the colour wave on 4 x 4 matrix blocks
    int split16;             // pixel-lane kernel, up to 16 channels: feature and moment blocks split over the waves by quadrants
    int bf16;                // pixel-lane kernel: contractions on bf16 matrix instructions, operands split into two bf16 terms
    // first window of the pixel-lane kernel inside a graph (contraction 3, api.hip): BOTH the bf16 and the hybrid kernel are
    // launched and the frame's long-axis word (GeomState::counters[2]) lets exactly one of them run: a workgroup leaves at once
    // unless (*gate != 0) == (gate_want != 0).  Null: no gate.
    const uint32_t* gate;
    int gate_want;
    uint32_t band_b0, band_tb;   // band_perm (common.h): the forward call's band, (0, 0) = whole view
    float neg_half_w, neg_half_h;   // pixel-lane kernel: -W / 2, -H / 2 (the NDC scale of dL/dmean2D, Q8)
    // pixel-lane kernel: feature-map gradient at the loss's resolution, (gHg gWg, C) pixel-major (null: none); dL_dfeat may
    // then be null.  gsy / gsx: the resize scales (H - 1) / (gHg - 1), (W - 1) / (gWg - 1); gscale: device scalar or null
    const float* glow;
    const float* gscale;
    int gHg, gWg;
    float gsy, gsx;
#ifdef F3DGS_DEV
    int dev;          // development builds only (make DEV=1): bit0 skip flush atomics, bit1 skip pixel trips, bit2 skip MFMAs, bit3 phase timing
    unsigned long long* dev_cycles;   // [0] staging, [1] window walk, [2] pixel trips, [3] flush, [4] waves
#endif
};

struct TileGeom {
    int tx, ty;        // tile coordinates
};

}  // namespace f3dgs
