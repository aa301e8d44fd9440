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

__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0, int& x1,
                                          int& y1) {
    x0 = min(gx, max(0, (int)((px - radius) / TILE)));
    y0 = min(gy, max(0, (int)((py - radius) / TILE)));
    x1 = min(gx, max(0, (int)((px + radius + TILE - 1) / TILE)));
    y1 = min(gy, max(0, (int)((py + radius + TILE - 1) / TILE)));
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
struct CullParams {
    float mx, my, a, b, c, thresh;   // thresh = 2 * (ln(255 o) + margin); negative => never visible
};
__device__ __forceinline__ CullParams make_cull(float mx, float my, float ca, float cb, float cc, float opacity) {
    CullParams k;
    k.mx = mx; k.my = my; k.a = ca; k.b = cb; k.c = cc;
    const float s = 255.0f * opacity;
    k.thresh = s > 1.0f ? 2.0f * (logf(s) * 1.0001f + 1e-3f) : -1.0f;
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
    const float ymax = sqrtf(fmaxf(0.0f, k.thresh * k.a * det_inv));
    const float lo = fmaxf(ylo, -ymax), hi = fminf(yhi, ymax);
    if (lo > hi + 1e-3f) return false;
    const float X = sqrtf(fmaxf(0.0f, k.thresh * k.c * det_inv));      // half-width of the whole ellipse
    const float ic = 1.0f / k.c, ia = 1.0f / k.a;
    const float det = 1.0f / det_inv;
    auto root = [&](float dy) { return sqrtf(fmaxf(0.0f, k.thresh * k.a - det * dy * dy)); };
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

__device__ V3 sh_to_rgb(int deg, V3 mean, const float* campos, const float* sh, uint8_t& clamp_bits) {
    V3 dir = {mean.x - campos[0], mean.y - campos[1], mean.z - campos[2]};
    const float len = sqrtf(dot3(dir, dir));
    dir = {dir.x / len, dir.y / len, dir.z / len};
    V3 res = K0 * ldv3(sh, 0);
    if (deg > 0) {
        const float x = dir.x, y = dir.y, z = dir.z;
        res = res - (K1 * y) * ldv3(sh, 1) + (K1 * z) * ldv3(sh, 2) - (K1 * x) * ldv3(sh, 3);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + (K2[0] * xy) * ldv3(sh, 4) + (K2[1] * yz) * ldv3(sh, 5) +
                  (K2[2] * (2.0f * zz - xx - yy)) * ldv3(sh, 6) + (K2[3] * xz) * ldv3(sh, 7) +
                  (K2[4] * (xx - yy)) * ldv3(sh, 8);
            if (deg > 2) {
                res = res + ((K3[0] * y) * (3.0f * xx - yy)) * ldv3(sh, 9) + ((K3[1] * xy) * z) * ldv3(sh, 10) +
                      ((K3[2] * y) * (4.0f * zz - xx - yy)) * ldv3(sh, 11) +
                      ((K3[3] * z) * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * ldv3(sh, 12) +
                      ((K3[4] * x) * (4.0f * zz - xx - yy)) * ldv3(sh, 13) + ((K3[5] * z) * (xx - yy)) * ldv3(sh, 14) +
                      ((K3[6] * x) * (xx - 3.0f * yy)) * ldv3(sh, 15);
            }
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamp_bits = (uint8_t)((res.x < 0 ? 1 : 0) | (res.y < 0 ? 2 : 0) | (res.z < 0 ? 4 : 0));
    return {fmaxf(res.x, 0.0f), fmaxf(res.y, 0.0f), fmaxf(res.z, 0.0f)};
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D, ViewParams vp,
                                                           uint8_t* __restrict__ present) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const V3 p = {means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]};
    present[i] = xform3(vp.view, p).z > 0.2f ? 1 : 0;
}

__global__ void __launch_bounds__(256)
preprocess_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                  const float* __restrict__ rotations, const float* __restrict__ opacities,
                  const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
                  const float* __restrict__ colors_precomp, ViewParams vp, int* __restrict__ radii,
                  SplatRec* __restrict__ rec, uint8_t* __restrict__ clamped, uint32_t* __restrict__ tiles_touched,
                  uint32_t* __restrict__ depth_key, int cull, uint32_t* __restrict__ ref_partial,
                  uint32_t* __restrict__ depth_hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    // the digit histograms of the depth sort are accumulated by the NEXT kernel: zero them here (1024 words)
    if (blockIdx.x < 4) depth_hist[blockIdx.x * 256 + threadIdx.x] = 0;
    int out_radius = 0;
    uint32_t out_tiles = 0, out_key = 0xFFFFFFFFu, bbox_tiles = 0;
    if (i < P) do {
        const V3 p = {means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]};
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
            const V3 sc = {scales[3 * (size_t)i], scales[3 * (size_t)i + 1], scales[3 * (size_t)i + 2]};
            const float4 q = reinterpret_cast<const float4*>(rotations)[i];
            cov3d_from_scale_rot(sc, vp.scale_modifier, q, cov6);
        }
        const Ewa e = ewa_setup(p, vp, cov6);
        float a, b, c;
        ewa_cov(e, a, b, c);
        const float det = a * c - b * b;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float ca = c * det_inv, cb = -b * det_inv, cc = a * det_inv;
        const float mid = 0.5f * (a + c);
        const float root = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float l1 = mid + root, l2 = mid - root;
        const float rad = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        const float px = ndc_to_pix(projx, vp.W), py = ndc_to_pix(projy, vp.H);
        int x0, y0, x1, y1;
        tile_rect(px, py, (int)rad, vp.gx, vp.gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) break;
        V3 col;
        uint8_t cl = 0;
        if (colors_precomp) {
            col = {colors_precomp[3 * (size_t)i], colors_precomp[3 * (size_t)i + 1], colors_precomp[3 * (size_t)i + 2]};
        } else {
            col = sh_to_rgb(D, p, vp.campos, shs + 3 * (size_t)M * i, cl);   // (LDS staging measured slower here)
        }
        SplatRec r;
        r.q0 = make_float4(px, py, ca, cb);
        r.q1 = make_float4(cc, opacities[i], col.x, col.y);
        r.q2 = make_float4(col.z, pv.z, __int_as_float((int)rad), 0.f);  // q2.z = radius bits (for emit)
        rec[i] = r;
        clamped[i] = cl;
        out_radius = (int)rad;
        bbox_tiles = (uint32_t)((y1 - y0) * (x1 - x0));
        if (cull) {
            const CullParams ck = make_cull(px, py, ca, cb, cc, opacities[i]);
            const float cdet_inv = 1.0f / (ca * cc - cb * cb);
            for (int ty = y0; ty < y1; ty++) {
                int xa, xb;
                if (row_span(ck, cdet_inv, ty, x0, x1, xa, xb)) out_tiles += (uint32_t)(xb - xa + 1);
            }
        } else {
            out_tiles = bbox_tiles;
        }
        out_key = __float_as_uint(pv.z);
    } while (false);
    if (i < P) {
        if (radii) radii[i] = out_radius;
        tiles_touched[i] = out_tiles;
        depth_key[i] = out_key;
        if (out_radius == 0) clamped[i] = 0;
    }
    // the reference's num_rendered = sum of bounding-rectangle tile counts (rasterizer_impl.cu:279-283)
    // Same-address atomics serialise at ~12 ns each on MI355X, so every workgroup writes ONE partial sum and
    // the scan's spine kernel adds the partials up.
    // The same goes for the length of our own (culled) instance lists: both totals are known right after this
    // kernel, long before the host needs them (api.hip reads them back while the depth sort runs).
    __shared__ uint32_t wsum[2][4];
    uint32_t v = bbox_tiles, u = out_tiles;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += (uint32_t)__shfl_xor((int)v, d, 64);
        u += (uint32_t)__shfl_xor((int)u, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { wsum[0][threadIdx.x >> 6] = v; wsum[1][threadIdx.x >> 6] = u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nbp = (P + 255) / 256;      // padding workgroups (tiny P) have nothing to report
        if ((int)blockIdx.x < nbp) {
            ref_partial[blockIdx.x] = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
            ref_partial[nbp + blockIdx.x] = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
        }
    }
}

// counters[0] = instances in our lists, counters[1] = the reference's bounding-rectangle count
__global__ void __launch_bounds__(256) count_totals_kernel(const uint32_t* __restrict__ partial, int nb,
                                                           uint32_t* __restrict__ counters) {
    __shared__ uint32_t sh[2][4];
    uint32_t v = 0, u = 0;
    for (int i = threadIdx.x; i < nb; i += 256) { v += partial[i]; u += partial[nb + i]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += (uint32_t)__shfl_xor((int)v, d, 64);
        u += (uint32_t)__shfl_xor((int)u, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = v; sh[1][threadIdx.x >> 6] = u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[1] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        counters[0] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

// Emit (tile, id) instances in depth order (same tile test as the count above).
__global__ void __launch_bounds__(256)
emit_instances_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                      const uint32_t* __restrict__ tiles_touched, const SplatRec* __restrict__ rec, int gx, int gy,
                      int cull, uint32_t* __restrict__ inst_tile, uint32_t* __restrict__ inst_id) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t g = order[i];
    if (tiles_touched[g] == 0) return;
    uint32_t off = offsets[i];
    const float4 q0 = rec[g].q0, q1 = rec[g].q1;
    const int radius = __float_as_int(rec[g].q2.z);  // integer bits stored by preprocess_kernel
    int x0, y0, x1, y1;
    tile_rect(q0.x, q0.y, radius, gx, gy, x0, y0, x1, y1);
    const CullParams ck = make_cull(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
    const float cdet_inv = 1.0f / (q0.z * q1.x - q0.w * q0.w);
    for (int y = y0; y < y1; y++) {
        int xa = x0, xb = x1 - 1;
        if (cull && !row_span(ck, cdet_inv, y, x0, x1, xa, xb)) continue;
        for (int x = xa; x <= xb; x++) {
            inst_tile[off] = (uint32_t)(y * gx + x);
            inst_id[off] = g;
            off++;
        }
    }
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
                 const SplatRec* __restrict__ rec, int gx, int gy, int cull, uint32_t* __restrict__ inst_tile,
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
        tile_rect(q0.x, q0.y, radius, gx, gy, x0, y0, x1, y1);
        const CullParams ck = make_cull(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
        const float cdet_inv = 1.0f / (q0.z * q1.x - q0.w * q0.w);
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

__global__ void __launch_bounds__(256)
preprocess_backward_kernel(int P, int D, int M, const float* __restrict__ means3D, const int* __restrict__ radii,
                           const float* __restrict__ shs, const float* __restrict__ scales,
                           const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, ViewParams vp,
                           const uint8_t* __restrict__ clamped, const float* __restrict__ grec,
                           float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
                           float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor,
                           float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
                           float* __restrict__ dL_dscale, float* __restrict__ dL_drot, float* __restrict__ dL_dz) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    // SH coefficients in, SH gradients out: both are contiguous per workgroup and go through one LDS tile
    // (coalesced 16-byte global accesses; per-thread rows with an odd stride).
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];
    const int row = 3 * M + 1;
    const bool use_sh = dL_dsh && M > 0;
    const size_t sh_first = (size_t)blockIdx.x * 256 * 3 * M;
    const int sh_count = use_sh ? (int)min((size_t)256 * 3 * M, (size_t)P * 3 * M - sh_first) : 0;
    const int m3 = 3 * M;                           // LDS index of element e: e + e / m3 (row stride m3 + 1)
    const bool sh_vec = (m3 & 3) == 0;              // rows are whole float4s (sh_first = 768 M is always 16-B aligned)
    if (use_sh) {
        if (sh_vec) {
            for (int e = 4 * threadIdx.x; e < sh_count; e += 4 * 256) {
                const float4 q = *reinterpret_cast<const float4*>(shs + sh_first + e);
                float* d = sh_lds + e + e / m3;
                d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
            }
        } else {
            for (int e = threadIdx.x; e < sh_count; e += 256) sh_lds[e + e / m3] = shs[sh_first + e];
        }
        __syncthreads();
    }
    const bool in_range = i < P;
    const size_t si = (size_t)(in_range ? i : 0);
    const bool vis = in_range && radii[si] > 0;
    float gr[GREC];
    if (vis) {
        const float4* g4 = reinterpret_cast<const float4*>(grec + si * GREC);
        const float4 a = g4[0], b = g4[1], c = g4[2];
        gr[0] = a.x; gr[1] = a.y; gr[2] = a.z; gr[3] = a.w; gr[4] = b.x; gr[5] = b.y; gr[6] = b.z; gr[7] = b.w;
        gr[8] = c.x; gr[9] = c.y; gr[10] = c.z; gr[11] = c.w;
    } else {
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
    const V3 mean = {means3D[3 * si], means3D[3 * si + 1], means3D[3 * si + 2]};
    if (vis) {
        float cov6[6];
        V3 sc = {0, 0, 0};
        float4 q = make_float4(0, 0, 0, 0);
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cov6[k] = cov3D_precomp[6 * si + k];
        } else {
            sc = {scales[3 * si], scales[3 * si + 1], scales[3 * si + 2]};
            q = reinterpret_cast<const float4*>(rotations)[si];
            cov3d_from_scale_rot(sc, vp.scale_modifier, q, cov6);
        }
        // ---- conic -> cov2D -> (cov3D, t) --------------------------------------------------------
        const Ewa e = ewa_setup(mean, vp, cov6);
        float a, b, c;
        ewa_cov(e, a, b, c);
        const float dca = gr[2], dcb = gr[3], dcc = gr[4];
        const float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        // column i / row j of the reference's GLM matrices = (row j, col i) here
#define TT(i, j) e.T.v[j][i]
#define VV(i, j) e.Vrk.v[j][i]
#define WW(i, j) e.Wm.v[j][i]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dca + 2 * b * c * dcb + (denom - a * c) * dcc);
            dL_dc = denom2inv * (-a * a * dcc + 2 * a * b * dcb + (denom - a * c) * dca);
            dL_db = denom2inv * 2 * (b * c * dca - (denom + 2 * b * b) * dcb + a * b * dcc);
            dcov6[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
            dcov6[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
            dcov6[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
            dcov6[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
            dcov6[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
            dcov6[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
        }
        const float dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                           (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
        const float dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                           (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
        const float dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                           (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
        const float dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                           (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
        const float dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                           (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
        const float dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                           (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
        const float dJ00 = WW(0, 0) * dT00 + WW(0, 1) * dT01 + WW(0, 2) * dT02;
        const float dJ02 = WW(2, 0) * dT00 + WW(2, 1) * dT01 + WW(2, 2) * dT02;
        const float dJ11 = WW(1, 0) * dT10 + WW(1, 1) * dT11 + WW(1, 2) * dT12;
        const float dJ12 = WW(2, 0) * dT10 + WW(2, 1) * dT11 + WW(2, 2) * dT12;
#undef TT
#undef VV
#undef WW
        const float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float xm = e.clamp_x ? 0.f : 1.f, ym = e.clamp_y ? 0.f : 1.f;
        const float dtx = xm * -vp.fx * tz2 * dJ02;
        const float dty = ym * -vp.fy * tz2 * dJ12;
        const float dtz = -vp.fx * tz2 * dJ00 - vp.fy * tz2 * dJ11 + (2 * vp.fx * e.t[0]) * tz3 * dJ02 +
                          (2 * vp.fy * e.t[1]) * tz3 * dJ12;
        const float* m = vp.view;
        dmean.x = m[0] * dtx + m[1] * dty + m[2] * dtz;
        dmean.y = m[4] * dtx + m[5] * dty + m[6] * dtz;
        dmean.z = m[8] * dtx + m[9] * dty + m[10] * dtz;
        // ---- screen-space mean + depth -> 3D mean -----------------------------------------------
        const float* pr = vp.proj;
        const float mhw = pr[3] * mean.x + pr[7] * mean.y + pr[11] * mean.z + pr[15];
        const float mw = 1.0f / (mhw + 0.0000001f);
        const float mul1 = (pr[0] * mean.x + pr[4] * mean.y + pr[8] * mean.z + pr[12]) * mw * mw;
        const float mul2 = (pr[1] * mean.x + pr[5] * mean.y + pr[9] * mean.z + pr[13]) * mw * mw;
        const float gx2 = gr[0], gy2 = gr[1];
        V3 dm;
        dm.x = (pr[0] * mw - pr[3] * mul1) * gx2 + (pr[1] * mw - pr[3] * mul2) * gy2;
        dm.y = (pr[4] * mw - pr[7] * mul1) * gx2 + (pr[5] * mw - pr[7] * mul2) * gy2;
        dm.z = (pr[8] * mw - pr[11] * mul1) * gx2 + (pr[9] * mw - pr[11] * mul2) * gy2;
        const float dz = gr[9];
        dm.x += dz * vp.view[2]; dm.y += dz * vp.view[6]; dm.z += dz * vp.view[10];
        dmean.x += dm.x; dmean.y += dm.y; dmean.z += dm.z;
        // ---- cov3D -> scale / rotation ------------------------------------------------------------
        if (scales) {
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            const M3 R = quat_to_rot(q);
            const float s[3] = {vp.scale_modifier * sc.x, vp.scale_modifier * sc.y, vp.scale_modifier * sc.z};
            M3 M2, dS;
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int j = 0; j < 3; j++) M2.v[k][j] = 2.0f * (s[k] * R.v[j][k]);
            dS.v[0][0] = dcov6[0]; dS.v[0][1] = 0.5f * dcov6[1]; dS.v[0][2] = 0.5f * dcov6[2];
            dS.v[1][0] = 0.5f * dcov6[1]; dS.v[1][1] = dcov6[3]; dS.v[1][2] = 0.5f * dcov6[4];
            dS.v[2][0] = 0.5f * dcov6[2]; dS.v[2][1] = 0.5f * dcov6[4]; dS.v[2][2] = dcov6[5];
            M3 dM = mul3(M2, dS);
#pragma unroll
            for (int k = 0; k < 3; k++)
                dscale[k] = R.v[0][k] * dM.v[k][0] + R.v[1][k] * dM.v[k][1] + R.v[2][k] * dM.v[k][2];
#pragma unroll
            for (int j = 0; j < 3; j++) { dM.v[0][j] *= s[0]; dM.v[1][j] *= s[1]; dM.v[2][j] *= s[2]; }
#define DM(a, b) dM.v[a][b]
            drot[0] = 2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1));
            drot[1] = 2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1));
            drot[2] = 2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0));
            drot[3] = 2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0));
#undef DM
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
            sh_grad(D, mean, vp.campos, shc, clamped[si], dcol, dmean, rowp);
        }
        __syncthreads();
        if (sh_vec) {
            for (int e = 4 * threadIdx.x; e < sh_count; e += 4 * 256) {
                const float* d = sh_lds + e + e / m3;
                *reinterpret_cast<float4*>(dL_dsh + sh_first + e) = make_float4(d[0], d[1], d[2], d[3]);
            }
        } else {
            for (int e = threadIdx.x; e < sh_count; e += 256) dL_dsh[sh_first + e] = sh_lds[e + e / m3];
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
                       const float* colors_precomp, const ViewParams& vp, int* radii, GeomState g, int cull,
                       bool totals_kernel, hipStream_t s) {
    // at least 4 workgroups so that the depth_hist zero-fill above is complete even for tiny P
    const int grid = max(4, (P + 255) / 256);
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid), dim3(256), 0, s, P, D, M, means3D, scales, rotations,
                       opacities, shs, cov3D_precomp, colors_precomp, vp, radii, g.rec, g.clamped, g.tiles_touched,
                       g.depth_key, cull, g.ref_partial, g.depth_hist);
    // (the single-pass sort computes the totals in its prologue kernel instead)
    if (totals_kernel)
        hipLaunchKernelGGL(count_totals_kernel, dim3(1), dim3(256), 0, s, g.ref_partial, (P + 255) / 256, g.counters);
}

void launch_emit_instances(int P, const GeomState& g, const uint32_t* order, int gx, int gy, int cull,
                           uint32_t* inst_tile, uint32_t* inst_id, hipStream_t s) {
    hipLaunchKernelGGL(emit_instances_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, order, g.offsets,
                       g.tiles_touched, g.rec, gx, gy, cull, inst_tile, inst_id);
}

void launch_emit_scan(int P, const GeomState& g, const BinState& b, const uint32_t* order, int gx, int gy, int cull,
                      uint32_t* inst_tile, uint32_t* inst_id, uint32_t N, uint2* ranges, hipStream_t s) {
    const size_t n_zero = 2 * sort_blocks(N) * 256;
    hipLaunchKernelGGL(emit_scan_kernel, dim3(emit_blocks(P)), dim3(256), 0, s, P, order, g.tiles_touched, g.rec, gx, gy,
                       cull, inst_tile, inst_id, g.tickets, g.emit_status, g.hist_copies, b.tile_status, n_zero,
                       g.tile_hist, ranges);
}

void launch_preprocess_backward(int P, int D, int M, int C, const float* means3D, const int* radii, const float* shs,
                                const float* scales, const float* rotations, const float* cov3D_precomp,
                                const ViewParams& vp, const GeomState& g, const float* grec, float* dL_dmean2D,
                                float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                                float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dz,
                                hipStream_t s) {
    (void)C;
    const size_t lds = (dL_dsh && M > 0) ? (size_t)256 * (3 * M + 1) * sizeof(float) : 0;
    hipLaunchKernelGGL(preprocess_backward_kernel, dim3((P + 255) / 256), dim3(256), lds, s, P, D, M, means3D, radii, shs,
                       scales, rotations, cov3D_precomp, vp, g.clamped, grec, dL_dmean2D, dL_dconic, dL_dopacity,
                       dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dL_dz);
}

}  // namespace f3dgs
