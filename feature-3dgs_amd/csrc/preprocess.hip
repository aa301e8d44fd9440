// preprocess.hip — per-Gaussian stages: frustum test, projection (forward) and the fused
// cov2D / mean / SH / cov3D backward.
//
// BUILT WITH -ffp-contract=off: these kernels produce the integer artefacts (radii, tile rects,
// depth sort keys) and must agree bit-for-bit with the oracle, which is built the same way.  They
// are streaming, HBM-bound kernels; the lost FMA contraction is irrelevant for their speed.
//
// Semantics follow (R = submodules/diff-gaussian-rasterization-feature/cuda_rasterizer):
//   R/auxiliary.h:145-170 (near cull), R/forward.cu:156-256 (preprocess), :119-153 (cov3D),
//   :75-114 (EWA cov2D), :20-72 (SH colour); R/backward.cu:144-274 + :346-404 (fused here), :20-139, :278-341.
// The code is written from the mathematics (standard row/column matrix algebra), not from the
// reference's GLM expressions; operation order inside each 3x3 product is ((p0+p1)+p2).

#include "common.h"
#include "lookback.h"

namespace f3dgs {

namespace {

struct M3 {
    float v[3][3];
};
__device__ __forceinline__ M3 mul3(const M3& A, const M3& B) {
    M3 R;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) R.v[r][c] = A.v[r][0] * B.v[0][c] + A.v[r][1] * B.v[1][c] + A.v[r][2] * B.v[2][c];
    return R;
}
__device__ __forceinline__ M3 tr3(const M3& A) {
    M3 R;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) R.v[r][c] = A.v[c][r];
    return R;
}

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ V3 xform3(const float* m, V3 p) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}

__device__ __forceinline__ float ndc_to_pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// `band` = tile rows [band.x, band.y) this call lists (the whole grid: 0, gy - the reference's rectangle; a tile band of a view
// that is split over several GPUs, f3dgs_set_tile_band: the rectangle is clipped to the band's rows, so the count, the emission
// and the reference-style total all speak of the band's tiles only)
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int2 band, int& x0, int& y0, int& x1,
                                          int& y1) {
    x0 = min(gx, max(0, (int)((px - radius) / TILE)));
    y0 = min(band.y, max(band.x, (int)((py - radius) / TILE)));
    x1 = min(gx, max(0, (int)((px + radius + TILE - 1) / TILE)));
    y1 = min(band.y, max(band.x, (int)((py + radius + TILE - 1) / TILE)));
}


// ---- exact-safe tile culling ---------------------------------------------------------------------------
// A splat blends at a pixel only if alpha = opacity * exp(power) >= 1/255, i.e. iff
//   q(d) = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 opacity)      (d = mean - pixel centre).
// The reference emits one instance for EVERY tile of the 3-sigma bounding rectangle
// (rasterizer_impl.cu:98-109); about half of those can never pass the test above for any pixel of the
// tile.  A tile is kept unless the ellipse {q <= threshold} misses its pixel-centre rectangle by a safety margin that is orders of
// magnitude above the round-off of the blend kernels' own power / expf evaluation (row_span() below).  Dropped instances
// would have been skipped for every pixel, so images, gradients and radii are unchanged; only the private
// instance lists get shorter.  The count and the emission run the same code in this translation unit
// (no contraction), so they always agree.
// conditioning of the conic, clamped to [1, 1e6] (a NaN from a degenerate conic falls back to 1)
// (The culling test is this library's own - the reference has none - and only has to be conservative and identical between the count
// and the emission: its logarithm, square roots and reciprocals are the one-instruction approximations, ~1 ulp, far inside the
// margins below; the IEEE sequences cost ~130 instructions per visible Gaussian.)
__device__ __forceinline__ float cull_kappa(float ca, float cb, float cc) {
    return fminf(fmaxf((ca * cc) * __builtin_amdgcn_rcpf(ca * cc - cb * cb), 1.0f), 1e6f);
}
struct CullParams {
    float mx, my, a, b, c, thresh;   // thresh = 2 * (ln(255 o) + margin); negative => never visible
};
__device__ __forceinline__ CullParams make_cull(float mx, float my, float ca, float cb, float cc, float opacity) {
    CullParams k;
    k.mx = mx; k.my = my; k.a = ca; k.b = cb; k.c = cc;
    const float s = 255.0f * opacity;
    // The blend kernels evaluate the quadratic form in fp32; at a pixel ON the threshold ellipse its three terms are as
    // large as kappa * thresh with kappa = a c / det >= 1 (thin splats at an angle: the terms cancel), so their round-off
    // is ~ eps * kappa * thresh.  The threshold is inflated by 32 eps kappa on top of the fixed margin: a needle-shaped
    // splat keeps every tile it could blend in.
    k.thresh = s > 1.0f ? 2.0f * (__logf(s) * 1.0001f + 1e-3f) * (1.0f + 2e-6f * cull_kappa(ca, cb, cc)) : -1.0f;
    return k;
}
__device__ __forceinline__ float quad_form(const CullParams& k, float dx, float dy) {
    return (k.a * dx) * dx + 2.0f * ((k.b * dx) * dy) + (k.c * dy) * dy;
}
// The test is evaluated per tile ROW: the tiles of tile-row `ty` that the ellipse {q <= thresh} can touch form ONE
// contiguous span, because (ellipse ∩ horizontal band) is convex: a tile intersects the ellipse iff its x-range
// intersects the x-projection of (ellipse ∩ band).  The projection's right end is the concave function
// dx_r(dy) = (-b dy + sqrt(thresh a - det dy^2)) / a maximised over the band (at the ellipse's extreme point if
// that lies in the band, else at the nearer band edge); the left end symmetrically.  One pair of square roots
// per tile ROW instead of four quadratic-form evaluations per TILE.  The interval is widened by a small margin,
// so the span is a superset of the exact tile set (never drops a blending tile); count and emission share it.
__device__ __forceinline__ bool row_span(const CullParams& k, float det_inv, int ty, int x0, int x1, int& xa, int& xb) {
    if (k.thresh < 0.0f) return false;
    const float ylo = k.my - (float)(ty * TILE + TILE - 1), yhi = k.my - (float)(ty * TILE);   // dy = mean - pixel
    const float ymax = __builtin_amdgcn_sqrtf(fmaxf(0.0f, k.thresh * k.a * det_inv));
    const float lo = fmaxf(ylo, -ymax), hi = fminf(yhi, ymax);
    if (lo > hi + 1e-3f) return false;
    const float X = __builtin_amdgcn_sqrtf(fmaxf(0.0f, k.thresh * k.c * det_inv));      // half-width of the whole ellipse
    const float ic = __builtin_amdgcn_rcpf(k.c), ia = __builtin_amdgcn_rcpf(k.a);
    const float det = __builtin_amdgcn_rcpf(det_inv);
    auto root = [&](float dy) { return __builtin_amdgcn_sqrtf(fmaxf(0.0f, k.thresh * k.a - det * dy * dy)); };
    // right end: maximiser dy_r = -(b/c) X ; left end: minimiser dy_l = +(b/c) X
    const float dyr = fminf(hi, fmaxf(lo, -(k.b * ic) * X));
    const float dyl = fminf(hi, fmaxf(lo, (k.b * ic) * X));
    const float dxr = (-(k.b * dyr) + root(dyr)) * ia;
    const float dxl = (-(k.b * dyl) - root(dyl)) * ia;
    // pixel x = mean_x - dx  in  [mx - dxr, mx - dxl]; widen by a margin (relative + absolute)
    const float m = 2e-3f + 1e-4f * (fabsf(dxr) + fabsf(dxl));
    const float pxl = k.mx - dxr - m, pxr = k.mx - dxl + m;
    xa = max(x0, (int)ceilf((pxl - (float)(TILE - 1)) * (1.0f / TILE)));
    xb = min(x1 - 1, (int)floorf(pxr * (1.0f / TILE)));
    return xb >= xa;
}

__device__ __forceinline__ M3 quat_to_rot(float4 q) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;  // no normalisation (forward.cu:128)
    M3 R;
    R.v[0][0] = 1.f - 2.f * (y * y + z * z); R.v[0][1] = 2.f * (x * y - r * z); R.v[0][2] = 2.f * (x * z + r * y);
    R.v[1][0] = 2.f * (x * y + r * z); R.v[1][1] = 1.f - 2.f * (x * x + z * z); R.v[1][2] = 2.f * (y * z - r * x);
    R.v[2][0] = 2.f * (x * z - r * y); R.v[2][1] = 2.f * (y * z + r * x); R.v[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

// Sigma = R S^2 R^T as (S R^T)^T (S R^T); six unique entries.
__device__ __forceinline__ void cov3d_from_scale_rot(V3 scale, float mod, float4 q, float* cov6) {
    const float s[3] = {mod * scale.x, mod * scale.y, mod * scale.z};
    const M3 R = quat_to_rot(q);
    M3 Mm;
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 3; i++) Mm.v[k][i] = s[k] * R.v[i][k];
    const M3 Sg = mul3(tr3(Mm), Mm);
    cov6[0] = Sg.v[0][0]; cov6[1] = Sg.v[1][0]; cov6[2] = Sg.v[2][0];
    cov6[3] = Sg.v[1][1]; cov6[4] = Sg.v[2][1]; cov6[5] = Sg.v[2][2];
}

struct Ewa {
    M3 T, Vrk, Wm;
    float t[3];
    bool clamp_x, clamp_y;
};
__device__ __forceinline__ Ewa ewa_setup(V3 mean, const ViewParams& vp, const float* cov6) {
    Ewa e;
    const V3 tv = xform3(vp.view, mean);
    e.t[0] = tv.x; e.t[1] = tv.y; e.t[2] = tv.z;
    const float limx = 1.3f * vp.tanx, limy = 1.3f * vp.tany;
    const float txtz = e.t[0] / e.t[2], tytz = e.t[1] / e.t[2];
    e.t[0] = fminf(limx, fmaxf(-limx, txtz)) * e.t[2];
    e.t[1] = fminf(limy, fmaxf(-limy, tytz)) * e.t[2];
    e.clamp_x = (txtz < -limx || txtz > limx);
    e.clamp_y = (tytz < -limy || tytz > limy);
    M3 J;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) J.v[r][c] = 0.f;
    J.v[0][0] = vp.fx / e.t[2];
    J.v[2][0] = -(vp.fx * e.t[0]) / (e.t[2] * e.t[2]);
    J.v[1][1] = vp.fy / e.t[2];
    J.v[2][1] = -(vp.fy * e.t[1]) / (e.t[2] * e.t[2]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) e.Wm.v[r][c] = vp.view[4 * r + c];
    e.T = mul3(e.Wm, J);
    e.Vrk.v[0][0] = cov6[0]; e.Vrk.v[0][1] = cov6[1]; e.Vrk.v[0][2] = cov6[2];
    e.Vrk.v[1][0] = cov6[1]; e.Vrk.v[1][1] = cov6[3]; e.Vrk.v[1][2] = cov6[4];
    e.Vrk.v[2][0] = cov6[2]; e.Vrk.v[2][1] = cov6[4]; e.Vrk.v[2][2] = cov6[5];
    return e;
}
__device__ __forceinline__ void ewa_cov(const Ewa& e, float& a, float& b, float& c) {
    const M3 cov = mul3(mul3(tr3(e.T), tr3(e.Vrk)), e.T);
    a = cov.v[0][0] + 0.3f;
    b = cov.v[1][0];
    c = cov.v[1][1] + 0.3f;
}

__constant__ const float K0 = 0.28209479177387814f;
__constant__ const float K1 = 0.4886025119029199f;
__constant__ const float K2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                  -1.0925484305920792f, 0.5462742152960396f};
__constant__ const float K3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                  0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                  -0.5900435899266435f};

// The SH block of one Gaussian is M*3 contiguous floats; read as V3 (12-byte) elements.
__device__ __forceinline__ V3 ldv3(const float* p, int k) { return {p[3 * k], p[3 * k + 1], p[3 * k + 2]}; }

// The colour sum in the reference's order of operations (forward.cu:30-61; no contraction: bit-equal colours), cut where the
// two halves of the staged SH row meet: coefficients 0..7 (`sh_first`) and 8..15 (`sh_second`, given at indices 0..7).  The
// band-2 sum `res + t4 + t5 + t6 + t7 + t8` is left-associative, so the cut after t7 changes nothing.
struct ShDir {
    float x, y, z;
};
__device__ __forceinline__ ShDir sh_dir(V3 mean, const float* campos) {
    V3 dir = {mean.x - campos[0], mean.y - campos[1], mean.z - campos[2]};
    const float len = sqrtf(dot3(dir, dir));
    return {dir.x / len, dir.y / len, dir.z / len};
}
__device__ __forceinline__ V3 sh_first(int deg, ShDir d, const float* sh) {
    V3 res = K0 * ldv3(sh, 0);
    if (deg > 0) {
        const float x = d.x, y = d.y, z = d.z;
        res = res - (K1 * y) * ldv3(sh, 1) + (K1 * z) * ldv3(sh, 2) - (K1 * x) * ldv3(sh, 3);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + (K2[0] * xy) * ldv3(sh, 4) + (K2[1] * yz) * ldv3(sh, 5) +
                  (K2[2] * (2.0f * zz - xx - yy)) * ldv3(sh, 6) + (K2[3] * xz) * ldv3(sh, 7);
        }
    }
    return res;
}
__device__ __forceinline__ V3 sh_second(int deg, ShDir d, V3 res, const float* sh8, uint8_t& clamp_bits) {
    if (deg > 1) {
        const float x = d.x, y = d.y, z = d.z;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y;
        res = res + (K2[4] * (xx - yy)) * ldv3(sh8, 0);
        if (deg > 2) {
            res = res + ((K3[0] * y) * (3.0f * xx - yy)) * ldv3(sh8, 1) + ((K3[1] * xy) * z) * ldv3(sh8, 2) +
                  ((K3[2] * y) * (4.0f * zz - xx - yy)) * ldv3(sh8, 3) +
                  ((K3[3] * z) * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * ldv3(sh8, 4) +
                  ((K3[4] * x) * (4.0f * zz - xx - yy)) * ldv3(sh8, 5) + ((K3[5] * z) * (xx - yy)) * ldv3(sh8, 6) +
                  ((K3[6] * x) * (xx - 3.0f * yy)) * ldv3(sh8, 7);
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamp_bits = (uint8_t)((res.x < 0 ? 1 : 0) | (res.y < 0 ? 2 : 0) | (res.z < 0 ? 4 : 0));
    return {fmaxf(res.x, 0.0f), fmaxf(res.y, 0.0f), fmaxf(res.z, 0.0f)};
}
// all sixteen coefficients at `sh` (any M >= the degree's count: coefficients beyond it are not read)
__device__ V3 sh_to_rgb(int deg, V3 mean, const float* campos, const float* sh, uint8_t& clamp_bits) {
    const ShDir d = sh_dir(mean, campos);
    return sh_second(deg, d, sh_first(deg, d, sh), sh + 24, clamp_bits);
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D, ViewParams vp,
                                                           uint8_t* __restrict__ present) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const V3 p = {means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]};
    present[i] = xform3(vp.view, p).z > 0.2f ? 1 : 0;
}

// M3C = 3 M when known at compile time (48: sixteen SH coefficients), 0 = any M / precomputed colours.  With M3C the SH
// block of a wave's 64 Gaussians goes through a wave-private LDS tile: twelve coalesced 16-byte requests per lane, all in
// flight together, rows of culled Gaussians skipped (their requests re-read the wave's first line), then per-lane rows with
// an odd stride.  Round 3 read the coefficients per lane straight from global memory: 48 dword requests per lane with a
// 192-byte lane stride (64 cache lines per request, the 48 KB a workgroup touches do not fit its 32 KB vector cache) in
// four dependent rounds (one per SH band) - 3.8 TB/s, 47 % of the kernel's cycles vector-pipe busy.
constexpr int PF_ROW = 25;        // LDS row stride of the SH tile (floats; half a row of 48 + 1): odd, so that the per-lane rows are conflict-free
template <int M3C>
__global__ void __launch_bounds__(256)
preprocess_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                  const float* __restrict__ rotations, const float* __restrict__ opacities,
                  const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
                  const float* __restrict__ colors_precomp, ViewParams vp, int* __restrict__ radii,
                  SplatRec* __restrict__ rec, uint8_t* __restrict__ clamped, uint32_t* __restrict__ tiles_touched,
                  uint32_t* __restrict__ depth_key, int cull, uint32_t* __restrict__ ref_partial,
                  uint32_t* __restrict__ depth_hist, uint32_t* __restrict__ big_ctl) {
    extern __shared__ __attribute__((aligned(16))) float sh_tile[];      // M3C: four wave-private tiles of 64 x PF_ROW floats
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // the digit histograms of the depth sort are accumulated by the NEXT kernel: zero them here (1024 words), and with them the
    // slot states of the emit kernel's big-splat queue - four workgroups of 256 threads clear exactly 4 x 256 words of each
    // (launch_preprocess starts at least four): a stale state 2 would silently drop a splat's list entries
    static_assert(BIGQ_CAP == 4 * 256, "slot states zeroed by workgroups 0..3 of this kernel, 256 words each");
    if (blockIdx.x < 4) {
        depth_hist[blockIdx.x * 256 + threadIdx.x] = 0;
        big_ctl[4 + blockIdx.x * 256 + threadIdx.x] = 0;         // the emit kernel's big-splat queue: slot states ...
        if (blockIdx.x == 0 && threadIdx.x < 4) big_ctl[threadIdx.x] = 0;      // ... and counters
    }
    int out_radius = 0;
    uint32_t out_tiles = 0, out_key = 0xFFFFFFFFu, bbox_tiles = 0;
    // ---- part 1: projection and culling.  Every input of the Gaussian is requested first (index clamped: no branch in
    // front of a load), `alive` says whether it survived
    const size_t si = (size_t)(i < P ? i : 0);
    const V3 p = {means3D[3 * si], means3D[3 * si + 1], means3D[3 * si + 2]};
    const float opacity = opacities[si];
    const bool have_sr = !cov3D_precomp;
    const float* sp = have_sr ? scales + 3 * si : means3D + 3 * si;             // valid either way, see preprocess_backward_kernel
    const float4* rp = have_sr ? reinterpret_cast<const float4*>(rotations) + si : reinterpret_cast<const float4*>(rec + si);
    const V3 sc_in = {sp[0], sp[1], sp[2]};
    const float4 q_in = *rp;
    bool alive = false;
    float px = 0.f, py = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, rad = 0.f, depth = 0.f;
    bool long_axis = false;       // a visible Gaussian longer than vp.max_axis_ratio times its width (3D scales or screen-space footprint)
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (i < P) do {
        const V3 pv = xform3(vp.view, p);
        if (pv.z <= 0.2f) break;
        const float* pm = vp.proj;
        const float hx = pm[0] * p.x + pm[4] * p.y + pm[8] * p.z + pm[12];
        const float hy = pm[1] * p.x + pm[5] * p.y + pm[9] * p.z + pm[13];
        const float hw = pm[3] * p.x + pm[7] * p.y + pm[11] * p.z + pm[15];
        const float pw = 1.0f / (hw + 0.0000001f);
        const float projx = hx * pw, projy = hy * pw;
        float cov6[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cov6[k] = cov3D_precomp[6 * (size_t)i + k];
        } else {
            cov3d_from_scale_rot(sc_in, vp.scale_modifier, q_in, cov6);
        }
        const Ewa e = ewa_setup(p, vp, cov6);
        float a, b, c;
        ewa_cov(e, a, b, c);
        const float det = a * c - b * b;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        ca = c * det_inv; cb = -b * det_inv; cc = a * det_inv;
        const float mid = 0.5f * (a + c);
        const float root = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float l1 = mid + root, l2 = mid - root;
        rad = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        px = ndc_to_pix(projx, vp.W); py = ndc_to_pix(projy, vp.H);
        tile_rect(px, py, (int)rad, vp.gx, make_int2(vp.band0, vp.band1), x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) break;
        depth = pv.z;
        alive = true;
        // What the blend backward's contraction precision is chosen by (api.hip: option bwd_bf16 = -1): the covariance chain
        // (cov2D -> cov3D -> scale / rotation, backward.cu:144-341) amplifies an error of the blend-level sums by the square of
        // the Gaussian's axis ratio.  Screen space: sqrt(lambda_1 / lambda_2) with lambda_2 = det / lambda_1 (the product of the
        // eigenvalues), i.e. lambda_1^2 > ratio^2 det; 3D: largest scale > ratio x smallest.  Written so that a NaN (det <= 0,
        // non-finite scales) reads as "too long"; a zero scale does as well.
        const float mr = vp.max_axis_ratio, lmax = fmaxf(l1, l2);
        long_axis = !(lmax * lmax <= (mr * mr) * det);
        if (have_sr) {
            const float s0 = fabsf(sc_in.x), s1 = fabsf(sc_in.y), s2 = fabsf(sc_in.z);
            long_axis = long_axis || !(fmaxf(s0, fmaxf(s1, s2)) <= mr * fminf(s0, fminf(s1, s2))) || fminf(s0, fminf(s1, s2)) == 0.f;
        }
    } while (false);

    // ---- part 2 (M3C): the SH rows of the wave's surviving Gaussians -> the wave's LDS tile, in two HALVES (coefficients 0..7,
    // then 8..15, through the same 64 x 25 floats: 25 KB per workgroup instead of 49 - the tile, not the 96 registers, capped
    // the kernel at three waves per SIMD).  All twelve requests of a lane go out up front; the second half waits in registers
    // while the first is summed (sh_first), then takes its place (sh_second).
    float* const tile = sh_tile + w * (64 * PF_ROW);
    V3 col_part = {0.f, 0.f, 0.f};
    ShDir sdir = {0.f, 0.f, 0.f};
    if constexpr (M3C != 0) {
        static_assert(M3C == 48, "two halves of eight coefficients");
        const unsigned long long amask = __ballot(alive);
        const int wave_first = blockIdx.x * 256 + 64 * w;        // the wave's first Gaussian
        if (amask != 0ull) {                                      // (wave-uniform; a wave with a survivor starts inside [0, P))
            constexpr int NH = 6;                                 // 16-byte requests per lane and half: 64 rows x 24 floats
            const float4* src = reinterpret_cast<const float4*>(shs + (size_t)wave_first * M3C);
            float4 q[2][NH];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int k = 0; k < NH; k++) {
                    const int f = lane + 64 * k;                  // request f of a half: floats [4 (f % 6), + 4) of that half of row f / 6
                    const int r = f / NH, c4 = f % NH;
                    const bool on = (amask >> r) & 1ull;
                    q[h][k] = src[on ? r * (M3C / 4) + h * NH + c4 : 0];
                }
            if (alive) sdir = sh_dir(p, vp.campos);
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int k = 0; k < NH; k++) {
                    const int f = lane + 64 * k;
                    const int r = f / NH, c4 = f % NH;
                    if ((amask >> r) & 1ull) {
                        float* d = tile + r * PF_ROW + 4 * c4;    // row stride 25: the per-lane rows are conflict-free
                        d[0] = q[h][k].x; d[1] = q[h][k].y; d[2] = q[h][k].z; d[3] = q[h][k].w;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes have landed
                __builtin_amdgcn_wave_barrier();
                if (h == 0) {
                    if (alive && !colors_precomp) col_part = sh_first(D, sdir, tile + lane * PF_ROW);
                    __builtin_amdgcn_s_waitcnt(0xc07f);           // every lane has read its row: the tile takes the second half
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }

    // ---- part 3: colour, record, tile count
    if (alive) {
        V3 col;
        uint8_t cl = 0;
        if (colors_precomp) {
            col = {colors_precomp[3 * (size_t)i], colors_precomp[3 * (size_t)i + 1], colors_precomp[3 * (size_t)i + 2]};
        } else if constexpr (M3C != 0) {
            col = sh_second(D, sdir, col_part, tile + lane * PF_ROW, cl);
        } else {
            col = sh_to_rgb(D, p, vp.campos, shs + 3 * (size_t)M * i, cl);
        }
        SplatRec r;
        r.q0 = make_float4(px, py, ca, cb);
        r.q1 = make_float4(cc, opacity, col.x, col.y);
        r.q2 = make_float4(col.z, depth, __int_as_float((int)rad), 0.f);  // q2.z = radius bits (for emit)
        rec[i] = r;
        clamped[i] = cl;
        out_radius = (int)rad;
        bbox_tiles = (uint32_t)((y1 - y0) * (x1 - x0));
        if (cull) {
            const CullParams ck = make_cull(px, py, ca, cb, cc, opacity);
            const float cdet_inv = __builtin_amdgcn_rcpf(ca * cc - cb * cb);
            for (int ty = y0; ty < y1; ty++) {
                int xa, xb;
                if (row_span(ck, cdet_inv, ty, x0, x1, xa, xb)) out_tiles += (uint32_t)(xb - xa + 1);
            }
        } else {
            out_tiles = bbox_tiles;
        }
        out_key = __float_as_uint(depth);
    }
    if (i < P) {
        if (radii) radii[i] = out_radius;
        tiles_touched[i] = out_tiles;
        depth_key[i] = out_key;
        if (out_radius == 0) clamped[i] = 0;
    }
    // the reference's num_rendered = sum of bounding-rectangle tile counts (rasterizer_impl.cu:279-283)
    // Same-address atomics serialise at ~12 ns each on MI355X, so every workgroup writes ONE partial sum; the first
    // kernel of the depth sort adds the partials up (binning.hip: TotalsJob) and stores the totals into the host's
    // pinned words.  The same goes for the length of our own (culled) instance lists: both totals are known right
    // after this kernel, long before the host needs them (it waits for them while the depth sort runs).
    // Third partial: does the workgroup hold a visible Gaussian with a long axis (one ballot per wave).
    __shared__ uint32_t wsum[3][4];
    uint32_t v = bbox_tiles, u = out_tiles;
    const uint32_t ar = __ballot(alive && long_axis) != 0ull ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += (uint32_t)__shfl_xor((int)v, d, 64);
        u += (uint32_t)__shfl_xor((int)u, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { wsum[0][threadIdx.x >> 6] = v; wsum[1][threadIdx.x >> 6] = u; wsum[2][threadIdx.x >> 6] = ar; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nbp = (P + 255) / 256;      // padding workgroups (tiny P) have nothing to report
        if ((int)blockIdx.x < nbp) {
            ref_partial[blockIdx.x] = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
            ref_partial[nbp + blockIdx.x] = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
            ref_partial[2 * nbp + blockIdx.x] = max(max(wsum[2][0], wsum[2][1]), max(wsum[2][2], wsum[2][3]));
        }
    }
}

// Emit (tile, id) instances in depth order (same tile test as the count above).
// A lane owns one Gaussian and produces its run of the list; the 64 runs of a wave are adjacent in the list (offsets are
// an exclusive scan in this very order), so the wave first lays them out in LDS and then copies the whole range out with
// contiguous stores (a lane storing straight to its own run writes one dword to 64 different places per instruction:
// 0.061 ms for 42 MB at c3).  Ranges longer than the LDS slice are staged window by window (round 5).
// One splat of many tiles emitted by a whole wave: lane = tile row for the closed-form spans (a wave scan over the row counts),
// then row by row lane = tile column, consecutive lanes storing consecutive entries - the order of the per-lane walk (rows
// ascending, columns ascending).  `o`: the splat's first position in the list.
__device__ __forceinline__ void emit_big_splat(uint32_t gb, uint32_t o, const SplatRec* __restrict__ rec, int gx, int gy, int2 band, int cull,
                                               uint32_t* __restrict__ inst_tile, uint32_t* __restrict__ inst_id, int lane) {
    const float4 b0 = rec[gb].q0, b1 = rec[gb].q1;            // (one request: every lane asks for the same record)
    const int bradius = __float_as_int(rec[gb].q2.z);
    int bx0, by0, bx1, by1;
    tile_rect(b0.x, b0.y, bradius, gx, band, bx0, by0, bx1, by1);
    const CullParams bk = make_cull(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y);
    const float bdet_inv = __builtin_amdgcn_rcpf(b0.z * b1.x - b0.w * b0.w);
    for (int yb = by0; yb < by1; yb += 64) {
        const int yl = yb + lane;
        int xa = bx0, n = 0;
        if (yl < by1) {
            int xb_ = bx1 - 1;
            if (!cull || row_span(bk, bdet_inv, yl, bx0, bx1, xa, xb_)) n = xb_ - xa + 1;
        }
        const uint32_t inc_r = wave_incl_scan((uint32_t)n, lane);
        const uint32_t roff = inc_r - (uint32_t)n;
        const uint32_t btot = (uint32_t)__shfl((int)inc_r, 63, 64);
        const int rows = min(64, by1 - yb);
        for (int r = 0; r < rows; r++) {
            const int nr = __builtin_amdgcn_readlane(n, r);
            if (nr == 0) continue;
            const int xar = __builtin_amdgcn_readlane(xa, r);
            const uint32_t ro = o + (uint32_t)__builtin_amdgcn_readlane((int)roff, r);
            const uint32_t t0 = (uint32_t)((yb + r) * gx + xar);
            for (int k = lane; k < nr; k += 64) {
                inst_tile[ro + k] = t0 + (uint32_t)k;
                inst_id[ro + k] = gb;
            }
        }
        o += btot;
    }
}

// The big splats of a view are its NEAREST Gaussians - neighbours in depth order, i.e. lanes of the same one or two waves: emitted
// by their own wave they run one after the other (33 splats of ~5000 tiles at c3 rotated by 25 degrees: 0.2 ms on one wave
// while the chip idles).  So the owner PUBLISHES them - {Gaussian, first list position} in a slot of a small queue - before it
// does anything else, and every wave of the launch, when its own range is written, takes what it finds published (one claim
// per slot: 1 -> 2 by compare-and-swap); the owner takes whatever nobody claimed when it comes back.  No wave ever waits.
__device__ __forceinline__ uint32_t big_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool emit_big_claim(uint32_t* big_ctl, uint32_t slot, int lane) {
    uint32_t was = 0;
    if (lane == 0) was = atomicCAS(&big_ctl[4 + slot], 1u, 2u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)was) == 1u;
}
__device__ __forceinline__ void emit_big_steal(uint32_t* big_ctl, const uint2* big_items, const SplatRec* __restrict__ rec, int gx, int gy, int2 band,
                                               int cull, uint32_t* __restrict__ inst_tile, uint32_t* __restrict__ inst_id, int lane) {
    for (;;) {
        const uint32_t handed = min(big_load(&big_ctl[0]), (uint32_t)BIGQ_CAP);
        if (big_load(&big_ctl[1]) >= handed) return;             // (plain loads first: an empty or drained queue costs no atomic)
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(&big_ctl[1], 1u);
        i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        if (i >= handed) return;
        if (emit_big_claim(big_ctl, i, lane)) {                  // (a slot handed out but not yet published is left to its owner)
            __threadfence();
            const uint32_t gb = big_load(&big_items[i].x), o = big_load(&big_items[i].y);
            emit_big_splat(gb, o, rec, gx, gy, band, cull, inst_tile, inst_id, lane);
        }
    }
}

constexpr int EMIT_CAP = 1024;      // list entries per wave in LDS (8 KB)
constexpr int EMIT_BIG = 512;       // tiles from which on a splat is emitted by the whole wave
__global__ void __launch_bounds__(256)
emit_instances_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ chunk_sums,
                      const uint32_t* __restrict__ sub, const uint32_t* __restrict__ tiles_touched,
                      const SplatRec* __restrict__ rec, int gx, int gy, int2 band, int cull, uint32_t* __restrict__ inst_tile,
                      uint32_t* __restrict__ inst_id, uint2* __restrict__ ranges_enc, uint32_t* __restrict__ tile_len,
                      uint32_t* __restrict__ big_ctl, uint2* __restrict__ big_items, uint32_t cap, uint32_t* __restrict__ cap_word,
                      const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t s_tile[4][EMIT_CAP];
    __shared__ uint32_t s_id[4][EMIT_CAP];
    if (blockIdx.x == 0 && threadIdx.x == 0) *cap_word = cap;
    // The arrays hold `cap` entries: the exact list length where the host read it before this launch (n_dev null), a provision
    // where it did not (sync-free forward: n_dev = the device's count).  A frame that does not fit is VOID - no wave stores
    // anything, the tile sort takes its count as zero, the tile ranges stay empty - and the host learns it from the count (for
    // frames replayed from a graph: from the sticky `overflow` word) and runs the frame again with room.
    const bool no_room = n_dev && *n_dev > cap;
    if (no_room && overflow && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // all-ones = "no entry yet" for both halves of the encoded tile ranges (BinState::ranges_enc; the final sort pass,
    // two launches further on, lowers them with atomicMin); zero for the tiles' walk lengths (raised by the blend forward)
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < (uint32_t)(gx * gy); t += gridDim.x * 256) {
        ranges_enc[t] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        tile_len[t] = 0u;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool valid = i < P;
    const uint32_t g = valid ? order[i] : 0u;
    const uint32_t cnt = valid ? tiles_touched[g] : 0u;
    // the wave's range of the list starts behind the runs of 64 in front of its own inside its chunk of SCAN_CHUNK Gaussians
    // (binning.hip: scan_reduce_kernel); a lane's run starts behind the lanes in front of it
    const uint32_t first = (uint32_t)(blockIdx.x * 256 + 64 * w);              // the wave's first Gaussian (depth order)
    const uint32_t chunk = first / SCAN_CHUNK, run = (first % SCAN_CHUNK) / 64;
    uint32_t before = (uint32_t)lane < run ? sub[(size_t)chunk * 64 + lane] : 0u;
    // ... and the chunks in front of its own: their totals are summed here (a few coalesced reads) rather than scanned by a
    // launch of their own
    for (uint32_t c = lane; c < chunk; c += 64) before += chunk_sums[c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += (uint32_t)__shfl_xor((int)before, d, 64);
    const uint32_t base = before;
    const uint32_t inc = wave_incl_scan(cnt, lane);
    const uint32_t off0 = base + inc - cnt;
    const uint32_t total = (uint32_t)__shfl((int)inc, 63, 64);
    if (no_room || base + total > cap || base + total < base) return;      // (the second test never fires on consistent counts)
    // A splat of more than EMIT_BIG tiles (a Gaussian close to the camera plane of a rotated view: thousands of tiles, up to the
    // whole grid) would keep ONE lane looping for all of them while 63 idle: it is emitted by a whole wave (emit_big_splat) - by
    // whichever wave of the launch gets to it first (emit_big_steal).  A wave that holds such splats (wave-uniform ballot; rare)
    // publishes them, lets its other lanes store straight from their loops, then helps with the queue and finally takes what is
    // left of its own.  (c3 rotated by 25 degrees: emit 0.27 ms with the per-lane walk, 0.20 with the owner's wave emitting them
    // one after the other, 0.08 with the queue.)
    const unsigned long long bigm = __ballot(cnt > (uint32_t)EMIT_BIG);
    if (bigm != 0ull) {
        // (one slot reservation for all of the wave's big splats, the items written by their own lanes, one fence: publishing
        // them one by one - an atomic round trip, a fence and an exchange each - cost the owner 3 us per splat)
        const bool isbig = cnt > (uint32_t)EMIT_BIG;
        uint32_t slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(&big_ctl[0], (uint32_t)__popcll(bigm));
        slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0);
        const uint32_t myslot = isbig ? slot0 + (uint32_t)__popcll(bigm & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
        const bool queued = isbig && myslot < (uint32_t)BIGQ_CAP;
        if (queued) big_items[myslot] = make_uint2(g, off0);
        __threadfence();
        if (queued) __hip_atomic_store(&big_ctl[4 + myslot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // published
        if (cnt != 0 && cnt <= (uint32_t)EMIT_BIG) {
            uint32_t off_d = off0;
            const float4 d0 = rec[g].q0, d1 = rec[g].q1;
            const int dradius = __float_as_int(rec[g].q2.z);
            int dx0, dy0, dx1, dy1;
            tile_rect(d0.x, d0.y, dradius, gx, band, dx0, dy0, dx1, dy1);
            const CullParams dk = make_cull(d0.x, d0.y, d0.z, d0.w, d1.x, d1.y);
            const float ddet_inv = __builtin_amdgcn_rcpf(d0.z * d1.x - d0.w * d0.w);
            for (int y = dy0; y < dy1; y++) {
                int xa = dx0, xb = dx1 - 1;
                if (cull && !row_span(dk, ddet_inv, y, dx0, dx1, xa, xb)) continue;
                for (int x = xa; x <= xb; x++) {
                    inst_tile[off_d] = (uint32_t)(y * gx + x);
                    inst_id[off_d] = g;
                    off_d++;
                }
            }
        }
        emit_big_steal(big_ctl, big_items, rec, gx, gy, band, cull, inst_tile, inst_id, lane);
        for (unsigned long long m = bigm; m != 0ull; m &= m - 1ull) {          // what nobody took (or what found the queue full)
            const int b = __builtin_ctzll(m);
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)myslot, b);
            if (slot >= (uint32_t)BIGQ_CAP || emit_big_claim(big_ctl, slot, lane))
                emit_big_splat((uint32_t)__builtin_amdgcn_readlane((int)g, b), (uint32_t)__builtin_amdgcn_readlane((int)off0, b), rec, gx, gy, band,
                               cull, inst_tile, inst_id, lane);
        }
        return;
    }
    // Every wave stages its range of the list in LDS in windows of EMIT_CAP entries and stores each window contiguously
    // (coalesced 256-byte requests).  A lane walks its splat's rows and columns as a resumable state machine - (y, x, xb) survive
    // from window to window - so ranges of any length are staged: until round 5 a range above one window (c5: 33 tiles per
    // splat, 2100 entries per wave; a splat larger than the image: 8160 entries from ONE lane) was stored straight from the
    // per-lane loops, 64 scattered dwords per store instruction - the memory pipeline's address rate, not the loop, bound it.
    uint32_t off = off0;
    const uint32_t my_end = off0 + cnt;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    CullParams ck{};
    float cdet_inv = 0.f;
    if (cnt != 0) {
        q0 = rec[g].q0; q1 = rec[g].q1;
        const int radius = __float_as_int(rec[g].q2.z);  // integer bits stored by preprocess_kernel
        tile_rect(q0.x, q0.y, radius, gx, band, x0, y0, x1, y1);
        ck = make_cull(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
        cdet_inv = __builtin_amdgcn_rcpf(q0.z * q1.x - q0.w * q0.w);
    }
    if (total <= (uint32_t)EMIT_CAP) {          // (wave-uniform) one window: the plain loops (c3: 0.044 ms; the state machine 0.046)
        if (cnt != 0) {
            for (int y = y0; y < y1; y++) {
                int xa = x0, xb = x1 - 1;
                if (cull && !row_span(ck, cdet_inv, y, x0, x1, xa, xb)) continue;
                for (int x = xa; x <= xb; x++) {
                    s_tile[w][off - base] = (uint32_t)(y * gx + x);
                    s_id[w][off - base] = g;
                    off++;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS writes have landed
        for (uint32_t k = lane; k < total; k += 64) {
            inst_tile[base + k] = s_tile[w][k];
            inst_id[base + k] = s_id[w][k];
        }
        emit_big_steal(big_ctl, big_items, rec, gx, gy, band, cull, inst_tile, inst_id, lane);       // (two loads where no view holds a big splat)
        return;
    }
    int y = y0 - 1, x = 0, xb = -1;          // "row exhausted": the first step advances to row y0
    for (uint32_t wbase = base; wbase < base + total; wbase += (uint32_t)EMIT_CAP) {
        const uint32_t wend = min(wbase + (uint32_t)EMIT_CAP, base + total);
        while (off < my_end && off < wend) {
            if (x > xb) {                        // next row of the rectangle that has tiles under the ellipse
                y++;
                if (y >= y1) { off = my_end; break; }       // (cannot happen: the count came from the same spans; never spin)
                int xa = x0;
                xb = x1 - 1;
                if (cull && !row_span(ck, cdet_inv, y, x0, x1, xa, xb)) { xb = -1; x = 0; continue; }
                x = xa;
                continue;
            }
            s_tile[w][off - wbase] = (uint32_t)(y * gx + x);
            s_id[w][off - wbase] = g;
            off++;
            x++;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS writes have landed
        for (uint32_t k = wbase + lane; k < wend; k += 64) {
            inst_tile[k] = s_tile[w][k - wbase];
            inst_id[k] = s_id[w][k - wbase];
        }
        __builtin_amdgcn_wave_barrier();          // the window is re-used
    }
    emit_big_steal(big_ctl, big_items, rec, gx, gy, band, cull, inst_tile, inst_id, lane);
}


// Single-pass flavour of scan + emit (option sort_onesweep).  Workgroup vb (ticket order) owns Gaussians
// [256 vb, 256 vb + 256) of the depth order: it sums their tile counts, obtains its list offset by decoupled
// look-back (lookback.h) and emits.  On the way it builds the two digit histograms of the tile sort in LDS and
// adds them to one of HIST_COPIES private global copies (same-line global atomics serialise at 25-100 ns apiece
// on MI355X, so thousands of workgroups must not meet on 32 cache lines), clears the look-back words of the tile
// sort and presets the encoded tile ranges (BinState::ranges_enc) for the final sort pass; the workgroup
// that finishes LAST folds the histogram copies into the totals the sort passes read.
constexpr int HIST_COPIES = 64;

__global__ void __launch_bounds__(256)
emit_scan_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tiles_touched,
                 const SplatRec* __restrict__ rec, int gx, int gy, int2 band, int cull, uint32_t* __restrict__ inst_tile,
                 uint32_t* __restrict__ inst_id, uint32_t* __restrict__ tickets, uint32_t* __restrict__ status,
                 uint32_t* __restrict__ hist_copies, uint32_t* __restrict__ zero_words, size_t n_zero,
                 uint32_t* __restrict__ tile_hist, uint2* __restrict__ ranges) {
    __shared__ uint32_t sh[8];
    __shared__ uint32_t s_vb, s_base, s_last;
    __shared__ uint32_t h[2][256];
    if (threadIdx.x == 0) s_vb = atomicAdd(&tickets[4], 1u);
    h[0][threadIdx.x] = 0;
    h[1][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t vb = s_vb;
    for (size_t k = (size_t)vb * 256 + threadIdx.x; k < n_zero; k += (size_t)gridDim.x * 256) zero_words[k] = 0;
    const uint32_t tiles = (uint32_t)(gx * gy);
    for (uint32_t t = vb * 256 + threadIdx.x; t < tiles; t += gridDim.x * 256) ranges[t] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    const int i = (int)(vb * 256 + threadIdx.x);
    uint32_t g = 0, cnt = 0;
    if (i < P) {
        g = order[i];
        cnt = tiles_touched[g];
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan_256(cnt, sh, &tot);
    if (threadIdx.x < 64) {
        const uint32_t base = lookback_wave(status, vb, tot, (int)threadIdx.x);
        if (threadIdx.x == 0) s_base = base;
    }
    __syncthreads();
    if (cnt) {
        uint32_t off = s_base + ex;
        const float4 q0 = rec[g].q0, q1 = rec[g].q1;
        const int radius = __float_as_int(rec[g].q2.z);  // integer bits stored by preprocess_kernel
        int x0, y0, x1, y1;
        tile_rect(q0.x, q0.y, radius, gx, band, x0, y0, x1, y1);
        const CullParams ck = make_cull(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
        const float cdet_inv = __builtin_amdgcn_rcpf(q0.z * q1.x - q0.w * q0.w);
        for (int y = y0; y < y1; y++) {
            int xa = x0, xb = x1 - 1;
            if (cull && !row_span(ck, cdet_inv, y, x0, x1, xa, xb)) continue;
            for (int x = xa; x <= xb; x++) {
                const uint32_t t = (uint32_t)(y * gx + x);
                inst_tile[off] = t;
                inst_id[off] = g;
                atomicAdd(&h[0][t & 255], 1u);
                atomicAdd(&h[1][(t >> 8) & 255], 1u);
                off++;
            }
        }
    }
    __syncthreads();
    {
        uint32_t* mine = hist_copies + (size_t)(vb % HIST_COPIES) * 512;
        const uint32_t c0 = h[0][threadIdx.x], c1 = h[1][threadIdx.x];
        if (c0) atomicAdd(&mine[threadIdx.x], c0);
        if (c1) atomicAdd(&mine[256 + threadIdx.x], c1);
    }
    // ---- last workgroup out: fold the copies ----------------------------------------------------------------
    // (workgroup-scope release = "my atomics above have been acknowledged"; they are performed at the memory side,
    // so no L2 write-back is needed - an agent-scope fence here would flush the 40 MB this kernel just wrote)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&tickets[7], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    uint32_t a0 = 0, a1 = 0;
    for (int c = 0; c < HIST_COPIES; c++) {
        a0 += lb_load(&hist_copies[(size_t)c * 512 + threadIdx.x]);
        a1 += lb_load(&hist_copies[(size_t)c * 512 + 256 + threadIdx.x]);
    }
    tile_hist[threadIdx.x] = a0;
    tile_hist[256 + threadIdx.x] = a1;
}

// ---- backward: K8 (cov2D) + K9 (mean / SH / cov3D) fused, one thread per Gaussian ---------------------
__device__ void sh_grad(int deg, V3 mean, const float* campos, const float* sh, uint8_t clamp_bits, V3 dcol,
                        V3& dmean, float* dsh) {
    const V3 dir_o = {mean.x - campos[0], mean.y - campos[1], mean.z - campos[2]};
    const float len = sqrtf(dot3(dir_o, dir_o));
    const V3 dir = {dir_o.x / len, dir_o.y / len, dir_o.z / len};
    const V3 g = {dcol.x * ((clamp_bits & 1) ? 0.f : 1.f), dcol.y * ((clamp_bits & 2) ? 0.f : 1.f),
                  dcol.z * ((clamp_bits & 4) ? 0.f : 1.f)};
    V3 ddx = {0, 0, 0}, ddy = {0, 0, 0}, ddz = {0, 0, 0};
    const float x = dir.x, y = dir.y, z = dir.z;
    auto put = [&](int k, V3 v) { dsh[3 * k] = v.x; dsh[3 * k + 1] = v.y; dsh[3 * k + 2] = v.z; };
    put(0, K0 * g);
    if (deg > 0) {
        put(1, (-K1 * y) * g); put(2, (K1 * z) * g); put(3, (-K1 * x) * g);
        ddx = (-K1) * ldv3(sh, 3); ddy = (-K1) * ldv3(sh, 1); ddz = K1 * ldv3(sh, 2);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            put(4, (K2[0] * xy) * g); put(5, (K2[1] * yz) * g); put(6, (K2[2] * (2.f * zz - xx - yy)) * g);
            put(7, (K2[3] * xz) * g); put(8, (K2[4] * (xx - yy)) * g);
            ddx = ddx + ((K2[0] * y) * ldv3(sh, 4) + (K2[2] * 2.f * -x) * ldv3(sh, 6) + (K2[3] * z) * ldv3(sh, 7) +
                         (K2[4] * 2.f * x) * ldv3(sh, 8));
            ddy = ddy + ((K2[0] * x) * ldv3(sh, 4) + (K2[1] * z) * ldv3(sh, 5) + (K2[2] * 2.f * -y) * ldv3(sh, 6) +
                         (K2[4] * 2.f * -y) * ldv3(sh, 8));
            ddz = ddz + ((K2[1] * y) * ldv3(sh, 5) + (K2[2] * 2.f * 2.f * z) * ldv3(sh, 6) + (K2[3] * x) * ldv3(sh, 7));
            if (deg > 2) {
                put(9, ((K3[0] * y) * (3.f * xx - yy)) * g);
                put(10, ((K3[1] * xy) * z) * g);
                put(11, ((K3[2] * y) * (4.f * zz - xx - yy)) * g);
                put(12, ((K3[3] * z) * (2.f * zz - 3.f * xx - 3.f * yy)) * g);
                put(13, ((K3[4] * x) * (4.f * zz - xx - yy)) * g);
                put(14, ((K3[5] * z) * (xx - yy)) * g);
                put(15, ((K3[6] * x) * (xx - 3.f * yy)) * g);
                ddx = ddx + ((((K3[0] * ldv3(sh, 9)) * 3.f) * 2.f) * xy + (K3[1] * ldv3(sh, 10)) * yz +
                             ((K3[2] * ldv3(sh, 11)) * -2.f) * xy + (((K3[3] * ldv3(sh, 12)) * -3.f) * 2.f) * xz +
                             (K3[4] * ldv3(sh, 13)) * (-3.f * xx + 4.f * zz - yy) + ((K3[5] * ldv3(sh, 14)) * 2.f) * xz +
                             ((K3[6] * ldv3(sh, 15)) * 3.f) * (xx - yy));
                ddy = ddy + (((K3[0] * ldv3(sh, 9)) * 3.f) * (xx - yy) + (K3[1] * ldv3(sh, 10)) * xz +
                             (K3[2] * ldv3(sh, 11)) * (-3.f * yy + 4.f * zz - xx) +
                             (((K3[3] * ldv3(sh, 12)) * -3.f) * 2.f) * yz + ((K3[4] * ldv3(sh, 13)) * -2.f) * xy +
                             ((K3[5] * ldv3(sh, 14)) * -2.f) * yz + (((K3[6] * ldv3(sh, 15)) * -3.f) * 2.f) * xy);
                ddz = ddz + ((K3[1] * ldv3(sh, 10)) * xy + (((K3[2] * ldv3(sh, 11)) * 4.f) * 2.f) * yz +
                             ((K3[3] * ldv3(sh, 12)) * 3.f) * (2.f * zz - xx - yy) +
                             (((K3[4] * ldv3(sh, 13)) * 4.f) * 2.f) * xz + (K3[5] * ldv3(sh, 14)) * (xx - yy));
            }
        }
    }
    const V3 ddir = {dot3(ddx, g), dot3(ddy, g), dot3(ddz, g)};
    const V3 v = dir_o;
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean.x += ((+sum2 - v.x * v.x) * ddir.x - v.y * v.x * ddir.y - v.z * v.x * ddir.z) * inv32;
    dmean.y += (-v.x * v.y * ddir.x + (sum2 - v.y * v.y) * ddir.y - v.z * v.y * ddir.z) * inv32;
    dmean.z += (-v.x * v.z * ddir.x - v.y * v.z * ddir.y + (sum2 - v.z * v.z) * ddir.z) * inv32;
}

// PB_BLOCK Gaussians per workgroup: one wave and 12.3 KB of LDS at SH degree 3, so twelve workgroups share a CU (three
// with 256 threads, 49 KB each) and their load / compute / store phases interleave more finely: 0.160 -> 0.151 ms at
// c3, 0.74 -> 0.68 at c5 (128 threads: 0.156 / 0.69).
constexpr int PB_BLOCK = 64;
// M3C = 3 M when known at compile time (48: the model's sixteen SH coefficients), 0 = any M.  With M3C the kernel issues
// EVERY global load of a Gaussian block up front and branch-free - twelve 16-byte SH requests per lane, radius, gradient
// record, mean, scale, rotation: one memory latency per wave.  (Round 3 staged the SH block in a rolled loop - one request,
// one wait, twelve times - and fetched the gradient record behind the radius test: ~17 dependent round trips per wave,
// 3.8 TB/s with 12 waves per CU.)
template <int M3C>
__global__ void __launch_bounds__(PB_BLOCK)
preprocess_backward_kernel(int P, int D, int M, const float* __restrict__ means3D, const int* __restrict__ radii,
                           const float* __restrict__ shs, const float* __restrict__ scales,
                           const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, ViewParams vp,
                           const uint8_t* __restrict__ clamped, const float* __restrict__ grec,
                           float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
                           float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor,
                           float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
                           float* __restrict__ dL_dscale, float* __restrict__ dL_drot, float* __restrict__ dL_dz,
                           int block0) {
    // block0: first workgroup of this launch's row chunk (the rows of a call may be covered by several launches, see
    // f3dgs_set_grad_rows_ready_callback)
    const int blk = block0 + (int)blockIdx.x;
    const int i = blk * PB_BLOCK + threadIdx.x;
    // SH coefficients in, SH gradients out: both are contiguous per workgroup and go through one LDS tile
    // (coalesced 16-byte global accesses; per-thread rows with an odd stride).
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];
    const int row = 3 * M + 1;
    const bool use_sh = dL_dsh && M > 0;
    const size_t sh_first = (size_t)blk * PB_BLOCK * 3 * M;
    const int sh_count = use_sh ? (int)min((size_t)PB_BLOCK * 3 * M, (size_t)P * 3 * M - sh_first) : 0;
    const int m3 = M3C ? M3C : 3 * M;               // LDS index of element e: e + e / m3 (row stride m3 + 1)
    const bool sh_vec = (m3 & 3) == 0;              // rows are whole float4s (sh_first = 768 M is always 16-B aligned)
    const bool in_range = i < P;
    const size_t si = (size_t)(in_range ? i : 0);
    constexpr int NQ = M3C ? M3C / 4 : 1;           // 16-byte SH requests per lane (PB_BLOCK rows of M3C floats)
    float4 shq[NQ];
    if constexpr (M3C != 0) {
        if (use_sh) {
            const float4* src = reinterpret_cast<const float4*>(shs + sh_first);
            const int last = sh_count / 4 - 1;      // the block's tail (P not a multiple of 64) re-reads its last request
#pragma unroll
            for (int k = 0; k < NQ; k++) shq[k] = src[min((int)threadIdx.x + PB_BLOCK * k, last)];
        }
    }
    // per-Gaussian inputs, requested unconditionally (a culled Gaussian's record is read and dropped: 14 % of the rows, one
    // dependent latency less for everyone)
    const int radius = radii[si];
    float gr[GREC];
    {
        const float4* g4 = reinterpret_cast<const float4*>(grec + si * GREC);
        const float4 a = g4[0], b = g4[1], c = g4[2];
        gr[0] = a.x; gr[1] = a.y; gr[2] = a.z; gr[3] = a.w; gr[4] = b.x; gr[5] = b.y; gr[6] = b.z; gr[7] = b.w;
        gr[8] = c.x; gr[9] = c.y; gr[10] = c.z; gr[11] = c.w;
    }
    const V3 mean = {means3D[3 * si], means3D[3 * si + 1], means3D[3 * si + 2]};
    // scale / rotation: a valid address either way (precomputed covariances: the mean / the gradient record are re-read and
    // dropped) - a branch here would make the compiler wait for every request above before it merges the two paths
    const bool have_sr = scales && !cov3D_precomp;
    const float* sp = have_sr ? scales + 3 * si : means3D + 3 * si;
    const float4* rp = have_sr ? reinterpret_cast<const float4*>(rotations) + si : reinterpret_cast<const float4*>(grec + si * GREC);
    const V3 sc_in = {sp[0], sp[1], sp[2]};
    const float4 q_in = *rp;
    const uint8_t clamp_bits = clamped[si];
    if (use_sh) {
        if constexpr (M3C != 0) {
#pragma unroll
            for (int k = 0; k < NQ; k++) {
                const int e = 4 * ((int)threadIdx.x + PB_BLOCK * k);
                if (e < sh_count) {
                    float* d = sh_lds + e + e / M3C;
                    d[0] = shq[k].x; d[1] = shq[k].y; d[2] = shq[k].z; d[3] = shq[k].w;
                }
            }
        } else if (sh_vec) {
            for (int e = 4 * threadIdx.x; e < sh_count; e += 4 * PB_BLOCK) {
                const float4 q = *reinterpret_cast<const float4*>(shs + sh_first + e);
                float* d = sh_lds + e + e / m3;
                d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
            }
        } else {
            for (int e = threadIdx.x; e < sh_count; e += PB_BLOCK) sh_lds[e + e / m3] = shs[sh_first + e];
        }
        __syncthreads();
    }
    const bool vis = in_range && radius > 0;
    if (!vis) {
#pragma unroll
        for (int k = 0; k < GREC; k++) gr[k] = 0.f;
    }
    // pass-through outputs
    if (in_range) {
        dL_dmean2D[3 * si] = gr[0]; dL_dmean2D[3 * si + 1] = gr[1]; dL_dmean2D[3 * si + 2] = 0.f;
        dL_dopacity[si] = gr[5];
        dL_dcolor[3 * si] = gr[6]; dL_dcolor[3 * si + 1] = gr[7]; dL_dcolor[3 * si + 2] = gr[8];
        if (dL_dconic) reinterpret_cast<float4*>(dL_dconic)[si] = make_float4(gr[2], gr[3], 0.f, gr[4]);
        if (dL_dz) dL_dz[si] = gr[9];
    }

    float dcov6[6] = {0, 0, 0, 0, 0, 0};
    V3 dmean = {0, 0, 0};
    float dscale[3] = {0, 0, 0};
    float drot[4] = {0, 0, 0, 0};
    if (vis) {
        float cov6[6];
        const V3 sc = sc_in;
        const float4 q = q_in;
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cov6[k] = cov3D_precomp[6 * si + k];
        } else {
            cov3d_from_scale_rot(sc, vp.scale_modifier, q, cov6);
        }
        // ---- conic -> cov2D -> (cov3D, t), in matrix form ------------------------------------------------
        // Notation: A = J W (2x3, rows = screen axes), S = cov3D (symmetric 3x3), C = A S A^T + 0.3 I, conic = C^-1.
        // ewa_setup keeps A transposed: A[i][k] = e.T.v[k][i]; W[m][l] = e.Wm.v[l][m].
        //   dL/dC   = -C^-1 Gk C^-1          Gk = [[dca, dcb], [dcb, dcc]]  (the blend backward accumulates the
        //                                    off-diagonal conic parameter once, so it enters both entries whole)
        //   dL/dS   = A^T (dL/dC) A          (off-diagonal pairs of the six-parameter form add up)
        //   dL/dA   = 2 (dL/dC) A S
        //   dL/dJ   = (dL/dA) W^T            of which only J00, J02, J11, J12 depend on t
        const Ewa e = ewa_setup(mean, vp, cov6);
        float ca, cb, cc;
        ewa_cov(e, ca, cb, cc);
        const float det = ca * cc - cb * cb;
        const float det2inv = 1.0f / ((det * det) + 0.0000001f);   // the reference's regularised 1 / det^2
        // adj(C) = det * conic; dL/dC = -adj Gk adj / det^2 (symmetric: three entries)
        const float k00 = cc, k01 = -cb, k11 = ca;
        const float m00 = k00 * gr[2] + k01 * gr[3], m01 = k00 * gr[3] + k01 * gr[4];
        const float m10 = k01 * gr[2] + k11 * gr[3], m11 = k01 * gr[3] + k11 * gr[4];
        const float gC[2][2] = {{-(m00 * k00 + m01 * k01) * det2inv, -(m00 * k01 + m01 * k11) * det2inv},
                                {-(m10 * k00 + m11 * k01) * det2inv, -(m10 * k01 + m11 * k11) * det2inv}};
        float A[2][3], B[2][3];        // B = (dL/dC) A
#pragma unroll
        for (int k = 0; k < 3; k++) { A[0][k] = e.T.v[k][0]; A[1][k] = e.T.v[k][1]; }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) B[i][k] = gC[i][0] * A[0][k] + gC[i][1] * A[1][k];
        float dS[3][3];                // A^T B
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int l = 0; l < 3; l++) dS[k][l] = A[0][k] * B[0][l] + A[1][k] * B[1][l];
        dcov6[0] = dS[0][0]; dcov6[1] = dS[0][1] + dS[1][0]; dcov6[2] = dS[0][2] + dS[2][0];
        dcov6[3] = dS[1][1]; dcov6[4] = dS[1][2] + dS[2][1]; dcov6[5] = dS[2][2];
        float dJ[2][3];                // 2 B S W^T
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float dA[3];
#pragma unroll
            for (int l = 0; l < 3; l++)
                dA[l] = 2.0f * (B[i][0] * e.Vrk.v[0][l] + B[i][1] * e.Vrk.v[1][l] + B[i][2] * e.Vrk.v[2][l]);
#pragma unroll
            for (int m = 0; m < 3; m++) dJ[i][m] = dA[0] * e.Wm.v[0][m] + dA[1] * e.Wm.v[1][m] + dA[2] * e.Wm.v[2][m];
        }
        // J = [[fx/tz, 0, -fx tx/tz^2], [0, fy/tz, -fy ty/tz^2]]; the frustum clamp of tx, ty is a constant (Q7)
        const float itz = 1.f / e.t[2], itz2 = itz * itz, itz3 = itz2 * itz;
        V3 dt;
        dt.x = e.clamp_x ? 0.f : -vp.fx * itz2 * dJ[0][2];
        dt.y = e.clamp_y ? 0.f : -vp.fy * itz2 * dJ[1][2];
        dt.z = -itz2 * (vp.fx * dJ[0][0] + vp.fy * dJ[1][1]) + 2.0f * itz3 * (vp.fx * e.t[0] * dJ[0][2] + vp.fy * e.t[1] * dJ[1][2]);
        // t = W p + const  ->  dL/dp = W^T dL/dt
        dmean.x = e.Wm.v[0][0] * dt.x + e.Wm.v[0][1] * dt.y + e.Wm.v[0][2] * dt.z;
        dmean.y = e.Wm.v[1][0] * dt.x + e.Wm.v[1][1] * dt.y + e.Wm.v[1][2] * dt.z;
        dmean.z = e.Wm.v[2][0] * dt.x + e.Wm.v[2][1] * dt.y + e.Wm.v[2][2] * dt.z;
        // ---- screen-space mean + depth -> 3D mean ------------------------------------------------------------
        // ndc = h.xy / (h.w + eps) with h = P p: quotient rule, dL/dp = (P_x^T g_x + P_y^T g_y) / w - (g . h.xy) P_w^T / w^2
        const float* pr = vp.proj;      // element (row r, col c) at [4c + r]
        const float hx = pr[0] * mean.x + pr[4] * mean.y + pr[8] * mean.z + pr[12];
        const float hy = pr[1] * mean.x + pr[5] * mean.y + pr[9] * mean.z + pr[13];
        const float hw = pr[3] * mean.x + pr[7] * mean.y + pr[11] * mean.z + pr[15];
        const float iw = 1.0f / (hw + 0.0000001f);
        const float gh = (gr[0] * hx + gr[1] * hy) * iw * iw;
        const float gxw = gr[0] * iw, gyw = gr[1] * iw, dz = gr[9];
        dmean.x += pr[0] * gxw + pr[1] * gyw - pr[3] * gh + dz * vp.view[2];
        dmean.y += pr[4] * gxw + pr[5] * gyw - pr[7] * gh + dz * vp.view[6];
        dmean.z += pr[8] * gxw + pr[9] * gyw - pr[11] * gh + dz * vp.view[10];
        // ---- cov3D -> scale / rotation -------------------------------------------------------------------------
        // S = L L^T with L = R diag(s), s = modifier * scale:  dL/dL = 2 G L  (G = symmetric per-entry gradient),
        // dL/ds_k = column k of dL/dL . column k of R  (returned w.r.t. s, as the reference does),  dL/dR = dL/dL diag(s).
        if (scales) {
            const M3 R = quat_to_rot(q);
            const float sv[3] = {vp.scale_modifier * sc.x, vp.scale_modifier * sc.y, vp.scale_modifier * sc.z};
            const float G[3][3] = {{dcov6[0], 0.5f * dcov6[1], 0.5f * dcov6[2]},
                                   {0.5f * dcov6[1], dcov6[3], 0.5f * dcov6[4]},
                                   {0.5f * dcov6[2], 0.5f * dcov6[4], dcov6[5]}};
            float dR[3][3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float dLk[3];          // column k of 2 G L
                dscale[k] = 0.f;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    dLk[i] = 2.0f * sv[k] * (G[i][0] * R.v[0][k] + G[i][1] * R.v[1][k] + G[i][2] * R.v[2][k]);
                    dscale[k] += dLk[i] * R.v[i][k];
                    dR[i][k] = dLk[i] * sv[k];
                }
            }
            // R(q) for the un-normalised q = (r, v): with Y = dR + dR^T (off-diagonal) and the axial vector
            // w = (dR21 - dR12, dR02 - dR20, dR10 - dR01):
            //   dL/dr = 2 v . w,   dL/dv = 2 (Y v + r w) - 4 v * (trace(dR) - diag(dR))
            const float r = q.x, vx = q.y, vy = q.z, vz = q.w;
            const float wx = dR[2][1] - dR[1][2], wy = dR[0][2] - dR[2][0], wz = dR[1][0] - dR[0][1];
            const float y01 = dR[0][1] + dR[1][0], y02 = dR[0][2] + dR[2][0], y12 = dR[1][2] + dR[2][1];
            drot[0] = 2.0f * (vx * wx + vy * wy + vz * wz);
            drot[1] = 2.0f * (y01 * vy + y02 * vz + r * wx) - 4.0f * vx * (dR[1][1] + dR[2][2]);
            drot[2] = 2.0f * (y01 * vx + y12 * vz + r * wy) - 4.0f * vy * (dR[0][0] + dR[2][2]);
            drot[3] = 2.0f * (y02 * vx + y12 * vy + r * wz) - 4.0f * vz * (dR[0][0] + dR[1][1]);
        }
    }
    // ---- SH gradients (writes all M coefficients; zero where unused / culled) ---------------------
    if (use_sh) {
        // the thread's LDS row holds its SH coefficients; the gradients overwrite it in place: sh_grad reads
        // coefficient k only for direction terms, which are accumulated into ddx/ddy/ddz BEFORE put(k) of a
        // higher band can alias... (they do alias: keep a private copy of the row first)
        float* rowp = sh_lds + threadIdx.x * row;
        float shc[48];                                 // private copy: the gradients overwrite the row in place
#pragma unroll
        for (int k = 0; k < 48; k++) shc[k] = k < m3 ? rowp[k] : 0.f;
        const int used = vis ? (D + 1) * (D + 1) : 0;
        const int usedc = used < M ? used : M;
        for (int k = 3 * usedc; k < m3; k++) rowp[k] = 0.f;
        if (vis) {
            const V3 dcol = {gr[6], gr[7], gr[8]};
            sh_grad(D, mean, vp.campos, shc, clamp_bits, dcol, dmean, rowp);
        }
        __syncthreads();
        if constexpr (M3C != 0) {
#pragma unroll
            for (int k = 0; k < NQ; k++) {
                const int e = 4 * ((int)threadIdx.x + PB_BLOCK * k);
                if (e < sh_count) {
                    const float* d = sh_lds + e + e / M3C;
                    *reinterpret_cast<float4*>(dL_dsh + sh_first + e) = make_float4(d[0], d[1], d[2], d[3]);
                }
            }
        } else if (sh_vec) {
            for (int e = 4 * threadIdx.x; e < sh_count; e += 4 * PB_BLOCK) {
                const float* d = sh_lds + e + e / m3;
                *reinterpret_cast<float4*>(dL_dsh + sh_first + e) = make_float4(d[0], d[1], d[2], d[3]);
            }
        } else {
            for (int e = threadIdx.x; e < sh_count; e += PB_BLOCK) dL_dsh[sh_first + e] = sh_lds[e + e / m3];
        }
    }
    if (!in_range) return;
    dL_dmean3D[3 * si] = dmean.x; dL_dmean3D[3 * si + 1] = dmean.y; dL_dmean3D[3 * si + 2] = dmean.z;
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov3D[6 * si + k] = dcov6[k];
    if (dL_dscale) { dL_dscale[3 * si] = dscale[0]; dL_dscale[3 * si + 1] = dscale[1]; dL_dscale[3 * si + 2] = dscale[2]; }
    if (dL_drot) reinterpret_cast<float4*>(dL_drot)[si] = make_float4(drot[0], drot[1], drot[2], drot[3]);
}

}  // namespace

void launch_mark_visible(int P, const float* means3D, const float* view_dev, uint8_t* present, hipStream_t s) {
    ViewParams vp = {};
    vp.view = view_dev;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, vp, present);
}

void launch_preprocess(int P, int D, int M, const float* means3D, const float* scales, const float* rotations,
                       const float* opacities, const float* shs, const float* cov3D_precomp,
                       const float* colors_precomp, const ViewParams& vp, int* radii, GeomState g, int cull, hipStream_t s) {
    // at least 4 workgroups so that the zero-fill of depth_hist AND of the big-splat queue's slot states (big_ctl) at the top of the
    // kernel is complete even for tiny P
    const int grid = max(4, (P + 255) / 256);
    if (M == 16 && shs && !colors_precomp)
        hipLaunchKernelGGL(preprocess_kernel<48>, dim3(grid), dim3(256), 4 * 64 * PF_ROW * sizeof(float), s, P, D, M, means3D, scales,
                           rotations, opacities, shs, cov3D_precomp, colors_precomp, vp, radii, g.rec, g.clamped,
                           g.tiles_touched, g.depth_key, cull, g.ref_partial, g.depth_hist, g.big_ctl);
    else
        hipLaunchKernelGGL(preprocess_kernel<0>, dim3(grid), dim3(256), 0, s, P, D, M, means3D, scales, rotations,
                           opacities, shs, cov3D_precomp, colors_precomp, vp, radii, g.rec, g.clamped, g.tiles_touched,
                           g.depth_key, cull, g.ref_partial, g.depth_hist, g.big_ctl);
}

void launch_emit_instances(int P, const GeomState& g, const uint32_t* order, int gx, int gy, int2 band, int cull,
                           uint32_t* inst_tile, uint32_t* inst_id, uint2* ranges_enc, uint32_t* tile_len, uint32_t cap,
                           const uint32_t* n_dev, uint32_t* overflow, hipStream_t s) {
    hipLaunchKernelGGL(emit_instances_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, order, g.scan_tmp, g.scan_sub,
                       g.tiles_touched, g.rec, gx, gy, band, cull, inst_tile, inst_id, ranges_enc, tile_len, g.big_ctl, g.big_items,
                       cap, g.counters + 3, n_dev, overflow);
}

void launch_emit_scan(int P, const GeomState& g, const BinState& b, const uint32_t* order, int gx, int gy, int2 band, int cull,
                      uint32_t* inst_tile, uint32_t* inst_id, uint32_t N, uint2* ranges, hipStream_t s) {
    const size_t n_zero = 2 * sort_blocks(N) * 256;
    hipLaunchKernelGGL(emit_scan_kernel, dim3(emit_blocks(P)), dim3(256), 0, s, P, order, g.tiles_touched, g.rec, gx, gy, band,
                       cull, inst_tile, inst_id, g.tickets, g.emit_status, g.hist_copies, b.tile_status, n_zero,
                       g.tile_hist, ranges);
}

void launch_preprocess_backward(int P, int D, int M, int C, const float* means3D, const int* radii, const float* shs,
                                const float* scales, const float* rotations, const float* cov3D_precomp,
                                const ViewParams& vp, const GeomState& g, const float* grec, float* dL_dmean2D,
                                float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                                float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dz,
                                int row_begin, int row_end, hipStream_t s) {
    (void)C;
    // rows [row_begin, row_end): row_begin is a multiple of PB_BLOCK (preprocess_backward_row_align())
    const int block0 = row_begin / PB_BLOCK, nblocks = (row_end - row_begin + PB_BLOCK - 1) / PB_BLOCK;
    if (nblocks <= 0) return;
    const size_t lds = (dL_dsh && M > 0) ? (size_t)PB_BLOCK * (3 * M + 1) * sizeof(float) : 0;
    if (M == 16)
        hipLaunchKernelGGL(preprocess_backward_kernel<48>, dim3(nblocks), dim3(PB_BLOCK), lds, s, P, D, M, means3D, radii, shs,
                           scales, rotations, cov3D_precomp, vp, g.clamped, grec, dL_dmean2D, dL_dconic, dL_dopacity,
                           dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dL_dz, block0);
    else
        hipLaunchKernelGGL(preprocess_backward_kernel<0>, dim3(nblocks), dim3(PB_BLOCK), lds, s, P, D, M, means3D, radii, shs,
                           scales, rotations, cov3D_precomp, vp, g.clamped, grec, dL_dmean2D, dL_dconic, dL_dopacity,
                           dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dL_dz, block0);
}

int preprocess_backward_row_align() { return PB_BLOCK; }

}  // namespace f3dgs
