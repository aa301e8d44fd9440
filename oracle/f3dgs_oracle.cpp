// f3dgs_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A scalar, single-threaded fp32 restatement of the algorithm of the
// Feature-3DGS differentiable rasterizer, written from the mathematics and the
// quirk list Q1-Q12 of SURVEY.md section 8(a).  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may load this library, and only
// as the checker.  The product (libf3dgs_hip.so) never links or calls it.
//
// PARITY STATUS: "parity unpinned by reference-owned vectors".  The reference
// repository ships no tests, golden vectors or CPU implementation of this path
// (SURVEY.md section 4); the CUDA sources cannot be built in this image (no
// nvcc).  What pins this file: (i) the reference's two Python fall-backs for
// sub-steps (SH->RGB, scale/rot->cov3D), frozen in tests/golden/ by
// tests/golden/make_reference_fallback_vectors.py; (ii) an independent
// PyTorch-autograd restatement (oracle/torch_oracle.py) whose autograd
// gradients must equal the analytic backward below.  There is no oracle/_ref:
// compiling the reference in place with hipcc was tried and fails on constructs
// that would need source rewriting ("<< <grid, block >> >" launch tokens,
// __trap(), a NUM_CHANNELS macro colliding with hipCUB) - see DESIGN.md.
//
// Reference files restated (R = submodules/diff-gaussian-rasterization-feature):
//   R/cuda_rasterizer/forward.cu:20-72    SH -> RGB              -> sh_to_rgb()
//   R/cuda_rasterizer/forward.cu:75-114   EWA cov2D              -> project_cov()
//   R/cuda_rasterizer/forward.cu:119-153  scale/rot -> cov3D     -> cov3d_from_scale_rot()
//   R/cuda_rasterizer/forward.cu:156-256  per-Gaussian preprocess-> stage_project()
//   R/cuda_rasterizer/rasterizer_impl.cu:35-50,70-138  keys, sort, ranges -> stage_bin()
//   R/cuda_rasterizer/forward.cu:261-396  tile blend             -> stage_blend()
//   R/cuda_rasterizer/backward.cu:407-620 blend backward         -> stage_blend_grad()
//   R/cuda_rasterizer/backward.cu:144-274 cov2D backward         -> cov2d_grad()
//   R/cuda_rasterizer/backward.cu:278-341 cov3D backward         -> cov3d_grad()
//   R/cuda_rasterizer/backward.cu:20-139  SH backward            -> sh_grad()
//   R/cuda_rasterizer/backward.cu:346-404 preprocess backward    -> stage_project_grad()
//   R/cuda_rasterizer/auxiliary.h:41-170  helpers
//
// Build: g++ -O2 -ffp-contract=off -std=c++17 -shared -fPIC (oracle/Makefile).
// fp contraction must stay off: the HIP preprocess kernels are built the same
// way so that integer artefacts (radii, tile rects, sort order) agree exactly.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int TILE = 16;  // R/cuda_rasterizer/config.h:18-19

// ---- small linear algebra with MATHEMATICAL (row, col) indexing -------------
struct M3 {
    float v[3][3];
};
// Product with the summation order GLM uses: ((p0 + p1) + p2).
inline M3 mul(const M3& A, const M3& B) {
    M3 R;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            R.v[r][c] = A.v[r][0] * B.v[0][c] + A.v[r][1] * B.v[1][c] + A.v[r][2] * B.v[2][c];
    return R;
}
inline M3 transpose(const M3& A) {
    M3 R;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R.v[r][c] = A.v[c][r];
    return R;
}

// Matrices arrive as 16 floats with element (r,c) at m[4c+r] (auxiliary.h:58-77).
inline void xform_point(const float* m, const float p[3], float out[3]) {
    out[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    out[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    out[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
inline void xform_point4(const float* m, const float p[3], float out[4]) {
    out[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    out[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    out[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    out[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

// auxiliary.h:41-44 — evaluated in double because of the literals 1.0 / 0.5.
inline float ndc_to_pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:46-56
inline void tile_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
    x0 = std::min(gx, std::max(0, (int)((px - radius) / TILE)));
    y0 = std::min(gy, std::max(0, (int)((py - radius) / TILE)));
    x1 = std::min(gx, std::max(0, (int)((px + radius + TILE - 1) / TILE)));
    y1 = std::min(gy, std::max(0, (int)((py + radius + TILE - 1) / TILE)));
}

// rasterizer_impl.cu:35-50
inline uint32_t higher_msb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// SH basis constants (auxiliary.h:22-39)
const float K0 = 0.28209479177387814f;
const float K1 = 0.4886025119029199f;
const float K2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                     0.5462742152960396f};
const float K3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                     -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct V3 {
    float x, y, z;
};
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// forward.cu:20-72
V3 sh_to_rgb(int deg, int M, const float* mean, const float* campos, const float* sh_all, uint8_t clamped[3]) {
    V3 dir = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
    float len = std::sqrt(dot3(dir, dir));
    dir = {dir.x / len, dir.y / len, dir.z / len};
    const V3* sh = reinterpret_cast<const V3*>(sh_all);
    (void)M;
    V3 res = K0 * sh[0];
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        res = res - (K1 * y) * sh[1] + (K1 * z) * sh[2] - (K1 * x) * sh[3];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + (K2[0] * xy) * sh[4] + (K2[1] * yz) * sh[5] + (K2[2] * (2.0f * zz - xx - yy)) * sh[6] +
                  (K2[3] * xz) * sh[7] + (K2[4] * (xx - yy)) * sh[8];
            if (deg > 2) {
                res = res + ((K3[0] * y) * (3.0f * xx - yy)) * sh[9] + ((K3[1] * xy) * z) * sh[10] +
                      ((K3[2] * y) * (4.0f * zz - xx - yy)) * sh[11] +
                      ((K3[3] * z) * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * sh[12] +
                      ((K3[4] * x) * (4.0f * zz - xx - yy)) * sh[13] + ((K3[5] * z) * (xx - yy)) * sh[14] +
                      ((K3[6] * x) * (xx - 3.0f * yy)) * sh[15];
            }
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamped[0] = res.x < 0; clamped[1] = res.y < 0; clamped[2] = res.z < 0;
    return {std::max(res.x, 0.0f), std::max(res.y, 0.0f), std::max(res.z, 0.0f)};
}

// Rotation matrix of the (unnormalised, Q9) quaternion q = (r,x,y,z), standard
// row-major sense.  forward.cu:128-139 builds its transpose in GLM storage.
inline M3 quat_to_rot(const float* q) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    M3 R;
    R.v[0][0] = 1.f - 2.f * (y * y + z * z); R.v[0][1] = 2.f * (x * y - r * z); R.v[0][2] = 2.f * (x * z + r * y);
    R.v[1][0] = 2.f * (x * y + r * z); R.v[1][1] = 1.f - 2.f * (x * x + z * z); R.v[1][2] = 2.f * (y * z - r * x);
    R.v[2][0] = 2.f * (x * z - r * y); R.v[2][1] = 2.f * (y * z + r * x); R.v[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

// forward.cu:119-153.  Sigma = R S^2 R^T, evaluated as (S R^T)^T (S R^T).
void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* cov6) {
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    M3 R = quat_to_rot(q);
    // Mm(k, i) = s_k * R(i, k)   (the reference's M = S * R_glm in math indexing)
    M3 Mm;
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 3; i++) Mm.v[k][i] = s[k] * R.v[i][k];
    M3 Sg = mul(transpose(Mm), Mm);
    cov6[0] = Sg.v[0][0]; cov6[1] = Sg.v[1][0]; cov6[2] = Sg.v[2][0];
    cov6[3] = Sg.v[1][1]; cov6[4] = Sg.v[2][1]; cov6[5] = Sg.v[2][2];
}

struct Cam {
    const float* view;
    const float* proj;
    float tanx, tany, fx, fy;
    int W, H, gx, gy;
};

// Shared by forward (forward.cu:75-114) and backward (backward.cu:164-199):
// builds T = (J * Rw2c)^T and Vrk, returns clamp flags.
struct Ewa {
    M3 T, Vrk, Wm;
    float t[3];
    bool clamp_x, clamp_y;
};
Ewa ewa_setup(const float* mean, const Cam& c, const float* cov6) {
    Ewa e;
    xform_point(c.view, mean, e.t);
    float limx = 1.3f * c.tanx, limy = 1.3f * c.tany;
    float txtz = e.t[0] / e.t[2], tytz = e.t[1] / e.t[2];
    e.t[0] = std::min(limx, std::max(-limx, txtz)) * e.t[2];
    e.t[1] = std::min(limy, std::max(-limy, tytz)) * e.t[2];
    e.clamp_x = (txtz < -limx || txtz > limx);
    e.clamp_y = (tytz < -limy || tytz > limy);
    // J_math = (2x3 Jacobian)^T padded with a zero column
    M3 J = {};
    J.v[0][0] = c.fx / e.t[2];
    J.v[2][0] = -(c.fx * e.t[0]) / (e.t[2] * e.t[2]);
    J.v[1][1] = c.fy / e.t[2];
    J.v[2][1] = -(c.fy * e.t[1]) / (e.t[2] * e.t[2]);
    // W_math(r,c) = view[4r + c]  (= Rw2c^T)
    for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) e.Wm.v[r][cc] = c.view[4 * r + cc];
    e.T = mul(e.Wm, J);
    // GLM Vrk(c0..c2) is symmetric; math indexing identical.
    e.Vrk.v[0][0] = cov6[0]; e.Vrk.v[0][1] = cov6[1]; e.Vrk.v[0][2] = cov6[2];
    e.Vrk.v[1][0] = cov6[1]; e.Vrk.v[1][1] = cov6[3]; e.Vrk.v[1][2] = cov6[4];
    e.Vrk.v[2][0] = cov6[2]; e.Vrk.v[2][1] = cov6[4]; e.Vrk.v[2][2] = cov6[5];
    return e;
}
// cov = T^T * Vrk^T * T, +0.3 low-pass; returns (a, b, c) = (cov00, cov(1,0), cov11)
inline void ewa_cov(const Ewa& e, float& a, float& b, float& cc) {
    M3 cov = mul(mul(transpose(e.T), transpose(e.Vrk)), e.T);
    a = cov.v[0][0] + 0.3f;
    b = cov.v[1][0];
    cc = cov.v[1][1] + 0.3f;
}

struct State {
    int P = 0, D = 0, M = 0, C = 0, W = 0, H = 0, gx = 0, gy = 0, N = 0;
    std::vector<float> means2D, depths, cov3D, conic_opacity, rgb;
    std::vector<uint8_t> clamped;
    std::vector<int> radii;
    std::vector<uint32_t> tiles_touched, offsets, point_list;
    std::vector<uint64_t> keys;
    std::vector<uint32_t> ranges;  // 2 per tile
    std::vector<float> final_T;
    std::vector<uint32_t> n_contrib;
    // backward intermediates (kept for per-stage comparisons)
    std::vector<float> dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dz;
    std::string err;
};

// ---------------- forward stage 1: per-Gaussian projection -------------------
void stage_project(State& s, const Cam& cam, const float* means3D, const float* shs, const float* colors_precomp,
                   const float* opacities, const float* scales, float mod, const float* rotations,
                   const float* cov3D_precomp, const float* campos) {
    const int P = s.P;
    s.means2D.assign(2 * (size_t)P, 0.f);
    s.depths.assign(P, 0.f);
    s.cov3D.assign(6 * (size_t)P, 0.f);
    s.conic_opacity.assign(4 * (size_t)P, 0.f);
    s.rgb.assign(3 * (size_t)P, 0.f);
    s.clamped.assign(3 * (size_t)P, 0);
    s.radii.assign(P, 0);
    s.tiles_touched.assign(P, 0);
    for (int i = 0; i < P; i++) {
        const float* p = means3D + 3 * (size_t)i;
        float pv[3], ph[4];
        xform_point4(cam.proj, p, ph);
        xform_point(cam.view, p, pv);
        if (pv[2] <= 0.2f) continue;  // auxiliary.h:160 (near cull only)
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float projx = ph[0] * pw, projy = ph[1] * pw;
        const float* cov6;
        if (cov3D_precomp) {
            cov6 = cov3D_precomp + 6 * (size_t)i;
        } else {
            cov3d_from_scale_rot(scales + 3 * (size_t)i, mod, rotations + 4 * (size_t)i, &s.cov3D[6 * (size_t)i]);
            cov6 = &s.cov3D[6 * (size_t)i];
        }
        Ewa e = ewa_setup(p, cam, cov6);
        float a, b, c;
        ewa_cov(e, a, b, c);
        float det = a * c - b * b;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {c * det_inv, -b * det_inv, a * det_inv};
        float mid = 0.5f * (a + c);
        float root = std::sqrt(std::max(0.1f, mid * mid - det));
        float l1 = mid + root, l2 = mid - root;
        float rad = std::ceil(3.f * std::sqrt(std::max(l1, l2)));
        float px = ndc_to_pix(projx, cam.W), py = ndc_to_pix(projy, cam.H);
        int x0, y0, x1, y1;
        tile_rect(px, py, (int)rad, cam.gx, cam.gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        if (!colors_precomp) {
            V3 col = sh_to_rgb(s.D, s.M, p, campos, shs + 3 * (size_t)s.M * i, &s.clamped[3 * (size_t)i]);
            s.rgb[3 * (size_t)i + 0] = col.x; s.rgb[3 * (size_t)i + 1] = col.y; s.rgb[3 * (size_t)i + 2] = col.z;
        }
        s.depths[i] = pv[2];
        s.radii[i] = (int)rad;
        s.means2D[2 * (size_t)i] = px; s.means2D[2 * (size_t)i + 1] = py;
        s.conic_opacity[4 * (size_t)i + 0] = conic[0];
        s.conic_opacity[4 * (size_t)i + 1] = conic[1];
        s.conic_opacity[4 * (size_t)i + 2] = conic[2];
        s.conic_opacity[4 * (size_t)i + 3] = opacities[i];
        s.tiles_touched[i] = (uint32_t)((y1 - y0) * (x1 - x0));
    }
}

// ---------------- forward stage 2: instance keys, stable sort, ranges --------
void stage_bin(State& s) {
    const int P = s.P;
    s.offsets.resize(P);
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += s.tiles_touched[i]; s.offsets[i] = run; }
    s.N = P ? (int)run : 0;
    std::vector<std::pair<uint64_t, uint32_t>> inst((size_t)s.N);
    for (int i = 0; i < P; i++) {
        if (s.radii[i] <= 0) continue;
        uint32_t off = i == 0 ? 0 : s.offsets[i - 1];
        int x0, y0, x1, y1;
        tile_rect(s.means2D[2 * (size_t)i], s.means2D[2 * (size_t)i + 1], s.radii[i], s.gx, s.gy, x0, y0, x1, y1);
        uint32_t dbits;
        std::memcpy(&dbits, &s.depths[i], 4);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                uint64_t key = (uint64_t)(y * s.gx + x);
                key = (key << 32) | dbits;
                inst[off++] = {key, (uint32_t)i};
            }
    }
    // cub::DeviceRadixSort::SortPairs over bits [0, 32+bit): stable, ascending.
    const uint32_t bit = higher_msb((uint32_t)(s.gx * s.gy));
    const uint64_t mask = (bit + 32 >= 64) ? ~0ull : ((1ull << (32 + bit)) - 1);
    std::stable_sort(inst.begin(), inst.end(),
                     [mask](const auto& a, const auto& b) { return (a.first & mask) < (b.first & mask); });
    s.keys.resize(s.N);
    s.point_list.resize(s.N);
    for (int k = 0; k < s.N; k++) { s.keys[k] = inst[k].first; s.point_list[k] = inst[k].second; }
    s.ranges.assign(2 * (size_t)s.gx * s.gy, 0);
    for (int k = 0; k < s.N; k++) {
        uint32_t cur = (uint32_t)(s.keys[k] >> 32);
        if (k == 0) s.ranges[2 * (size_t)cur] = 0;
        else {
            uint32_t prev = (uint32_t)(s.keys[k - 1] >> 32);
            if (cur != prev) { s.ranges[2 * (size_t)prev + 1] = k; s.ranges[2 * (size_t)cur] = k; }
        }
        if (k == s.N - 1) s.ranges[2 * (size_t)cur + 1] = s.N;
    }
}

// ---------------- forward stage 3: front-to-back blend per pixel -------------
void stage_blend(State& s, const float* bg, const float* colors, const float* feat, float* out_color,
                 float* out_feat, float* out_depth) {
    const int W = s.W, H = s.H, C = s.C;
    const size_t HW = (size_t)W * H;
    s.final_T.assign(HW, 0.f);
    s.n_contrib.assign(HW, 0);
    std::vector<float> acc(C > 0 ? C : 1);
    for (int ty = 0; ty < s.gy; ty++)
        for (int tx = 0; tx < s.gx; tx++) {
            const uint32_t lo = s.ranges[2 * ((size_t)ty * s.gx + tx)], hi = s.ranges[2 * ((size_t)ty * s.gx + tx) + 1];
            for (int py = ty * TILE; py < std::min(H, (ty + 1) * TILE); py++)
                for (int px = tx * TILE; px < std::min(W, (tx + 1) * TILE); px++) {
                    float T = 1.0f, col[3] = {0, 0, 0}, dep = 0.f;
                    std::fill(acc.begin(), acc.end(), 0.f);
                    uint32_t last = 0;
                    const float fx = (float)px, fy = (float)py;
                    for (uint32_t k = lo; k < hi; k++) {
                        const uint32_t g = s.point_list[k];
                        const float dx = s.means2D[2 * (size_t)g] - fx, dy = s.means2D[2 * (size_t)g + 1] - fy;
                        const float* co = &s.conic_opacity[4 * (size_t)g];
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float alpha = std::min(0.99f, co[3] * std::exp(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        const float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) break;  // Q5: this Gaussian is NOT blended
                        for (int ch = 0; ch < 3; ch++) col[ch] += colors[3 * (size_t)g + ch] * alpha * T;
                        const float w = alpha * T;
                        dep += s.depths[g] * w;
                        for (int ch = 0; ch < C; ch++) acc[ch] += feat[(size_t)g * C + ch] * alpha * T;
                        T = test_T;
                        last = k - lo + 1;
                    }
                    const size_t pid = (size_t)py * W + px;
                    s.final_T[pid] = T;
                    s.n_contrib[pid] = last;
                    for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pid] = col[ch] + T * bg[ch];
                    out_depth[pid] = dep;  // Q4: no background on depth / features
                    for (int ch = 0; ch < C; ch++) out_feat[ch * HW + pid] = acc[ch];
                }
        }
}

// ---------------- backward stage 1: blend gradients (back to front) ----------
void stage_blend_grad(State& s, const float* bg, const float* colors, const float* dL_dpix, const float* dL_dfeat,
                      const float* dL_ddepth, float* dL_dfeature /* (P,C) */) {
    const int W = s.W, H = s.H, C = s.C, P = s.P;
    const size_t HW = (size_t)W * H;
    s.dL_dmean2D.assign(3 * (size_t)P, 0.f);
    s.dL_dconic.assign(4 * (size_t)P, 0.f);
    s.dL_dopacity.assign(P, 0.f);
    s.dL_dcolor.assign(3 * (size_t)P, 0.f);
    s.dL_dz.assign(P, 0.f);
    std::fill(dL_dfeature, dL_dfeature + (size_t)P * C, 0.f);
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;  // Q8
    for (int ty = 0; ty < s.gy; ty++)
        for (int tx = 0; tx < s.gx; tx++) {
            const uint32_t lo = s.ranges[2 * ((size_t)ty * s.gx + tx)];
            for (int py = ty * TILE; py < std::min(H, (ty + 1) * TILE); py++)
                for (int px = tx * TILE; px < std::min(W, (tx + 1) * TILE); px++) {
                    const size_t pid = (size_t)py * W + px;
                    const float T_final = s.final_T[pid];
                    float T = T_final;
                    const uint32_t last = s.n_contrib[pid];
                    float gpix[3] = {dL_dpix[pid], dL_dpix[HW + pid], dL_dpix[2 * HW + pid]};
                    const float gdep = dL_ddepth[pid];
                    float behind[3] = {0, 0, 0}, behind_d = 0.f;  // colour / depth composited behind
                    float prev_alpha = 0.f, prev_col[3] = {0, 0, 0}, prev_d = 0.f;
                    const float fx = (float)px, fy = (float)py;
                    for (uint32_t k = lo + last; k-- > lo;) {
                        const uint32_t g = s.point_list[k];
                        const float dx = s.means2D[2 * (size_t)g] - fx, dy = s.means2D[2 * (size_t)g + 1] - fy;
                        const float* co = &s.conic_opacity[4 * (size_t)g];
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float G = std::exp(power);
                        const float alpha = std::min(0.99f, co[3] * G);
                        if (alpha < 1.0f / 255.0f) continue;
                        T = T / (1.f - alpha);
                        const float w = alpha * T;
                        float dL_dalpha = 0.f;
                        for (int ch = 0; ch < 3; ch++) {
                            const float c = colors[3 * (size_t)g + ch];
                            behind[ch] = prev_alpha * prev_col[ch] + (1.f - prev_alpha) * behind[ch];
                            prev_col[ch] = c;
                            dL_dalpha += (c - behind[ch]) * gpix[ch];
                            s.dL_dcolor[3 * (size_t)g + ch] += w * gpix[ch];
                        }
                        const float cd = s.depths[g];
                        behind_d = prev_alpha * prev_d + (1.f - prev_alpha) * behind_d;
                        prev_d = cd;
                        dL_dalpha += (cd - behind_d) * gdep;
                        // Q2: the feature term does not enter dL_dalpha (backward.cu:575)
                        for (int ch = 0; ch < C; ch++) dL_dfeature[(size_t)g * C + ch] += w * dL_dfeat[ch * HW + pid];
                        dL_dalpha *= T;
                        prev_alpha = alpha;
                        float bgdot = 0.f;
                        for (int ch = 0; ch < 3; ch++) bgdot += bg[ch] * gpix[ch];
                        dL_dalpha += (-T_final / (1.f - alpha)) * bgdot;
                        // Q1: no gating by the 0.99 clamp
                        const float dL_dG = co[3] * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                        const float dG_ddely = -gdy * co[2] - gdx * co[1];
                        s.dL_dmean2D[3 * (size_t)g + 0] += dL_dG * dG_ddelx * ddelx_dx;
                        s.dL_dmean2D[3 * (size_t)g + 1] += dL_dG * dG_ddely * ddely_dy;
                        s.dL_dconic[4 * (size_t)g + 0] += -0.5f * gdx * dx * dL_dG;
                        s.dL_dconic[4 * (size_t)g + 1] += -0.5f * gdx * dy * dL_dG;
                        s.dL_dconic[4 * (size_t)g + 3] += -0.5f * gdy * dy * dL_dG;
                        s.dL_dopacity[g] += G * dL_dalpha;
                        s.dL_dz[g] += alpha * T * gdep;
                    }
                }
        }
}

// backward.cu:144-274 — gradient through conic = inverse(cov2D) and the EWA projection.
void cov2d_grad(const float* mean, const Cam& cam, const float* cov6, const float* dconic4, float* dmean3 /*assign*/,
                float* dcov6 /*assign*/) {
    Ewa e = ewa_setup(mean, cam, cov6);
    const M3& T = e.T;    // T(r,c) math; reference T[c][r]
    const M3& V = e.Vrk;
    float a, b, c;
    ewa_cov(e, a, b, c);
    const float dca = dconic4[0], dcb = dconic4[1], dcc = dconic4[3];
    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    // reference index translation: T[i][j] (GLM col i, row j) == T.v[j][i]
    auto t = [&](int i, int j) { return T.v[j][i]; };
    auto vr = [&](int i, int j) { return V.v[j][i]; };
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dca + 2 * b * c * dcb + (denom - a * c) * dcc);
        dL_dc = denom2inv * (-a * a * dcc + 2 * a * b * dcb + (denom - a * c) * dca);
        dL_db = denom2inv * 2 * (b * c * dca - (denom + 2 * b * b) * dcb + a * b * dcc);
        dcov6[0] = (t(0, 0) * t(0, 0) * dL_da + t(0, 0) * t(1, 0) * dL_db + t(1, 0) * t(1, 0) * dL_dc);
        dcov6[3] = (t(0, 1) * t(0, 1) * dL_da + t(0, 1) * t(1, 1) * dL_db + t(1, 1) * t(1, 1) * dL_dc);
        dcov6[5] = (t(0, 2) * t(0, 2) * dL_da + t(0, 2) * t(1, 2) * dL_db + t(1, 2) * t(1, 2) * dL_dc);
        dcov6[1] = 2 * t(0, 0) * t(0, 1) * dL_da + (t(0, 0) * t(1, 1) + t(0, 1) * t(1, 0)) * dL_db + 2 * t(1, 0) * t(1, 1) * dL_dc;
        dcov6[2] = 2 * t(0, 0) * t(0, 2) * dL_da + (t(0, 0) * t(1, 2) + t(0, 2) * t(1, 0)) * dL_db + 2 * t(1, 0) * t(1, 2) * dL_dc;
        dcov6[4] = 2 * t(0, 2) * t(0, 1) * dL_da + (t(0, 1) * t(1, 2) + t(0, 2) * t(1, 1)) * dL_db + 2 * t(1, 1) * t(1, 2) * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dcov6[i] = 0;
    }
    float dT00 = 2 * (t(0, 0) * vr(0, 0) + t(0, 1) * vr(0, 1) + t(0, 2) * vr(0, 2)) * dL_da +
                 (t(1, 0) * vr(0, 0) + t(1, 1) * vr(0, 1) + t(1, 2) * vr(0, 2)) * dL_db;
    float dT01 = 2 * (t(0, 0) * vr(1, 0) + t(0, 1) * vr(1, 1) + t(0, 2) * vr(1, 2)) * dL_da +
                 (t(1, 0) * vr(1, 0) + t(1, 1) * vr(1, 1) + t(1, 2) * vr(1, 2)) * dL_db;
    float dT02 = 2 * (t(0, 0) * vr(2, 0) + t(0, 1) * vr(2, 1) + t(0, 2) * vr(2, 2)) * dL_da +
                 (t(1, 0) * vr(2, 0) + t(1, 1) * vr(2, 1) + t(1, 2) * vr(2, 2)) * dL_db;
    float dT10 = 2 * (t(1, 0) * vr(0, 0) + t(1, 1) * vr(0, 1) + t(1, 2) * vr(0, 2)) * dL_dc +
                 (t(0, 0) * vr(0, 0) + t(0, 1) * vr(0, 1) + t(0, 2) * vr(0, 2)) * dL_db;
    float dT11 = 2 * (t(1, 0) * vr(1, 0) + t(1, 1) * vr(1, 1) + t(1, 2) * vr(1, 2)) * dL_dc +
                 (t(0, 0) * vr(1, 0) + t(0, 1) * vr(1, 1) + t(0, 2) * vr(1, 2)) * dL_db;
    float dT12 = 2 * (t(1, 0) * vr(2, 0) + t(1, 1) * vr(2, 1) + t(1, 2) * vr(2, 2)) * dL_dc +
                 (t(0, 0) * vr(2, 0) + t(0, 1) * vr(2, 1) + t(0, 2) * vr(2, 2)) * dL_db;
    auto w = [&](int i, int j) { return e.Wm.v[j][i]; };  // W[i][j] GLM
    float dJ00 = w(0, 0) * dT00 + w(0, 1) * dT01 + w(0, 2) * dT02;
    float dJ02 = w(2, 0) * dT00 + w(2, 1) * dT01 + w(2, 2) * dT02;
    float dJ11 = w(1, 0) * dT10 + w(1, 1) * dT11 + w(1, 2) * dT12;
    float dJ12 = w(2, 0) * dT10 + w(2, 1) * dT11 + w(2, 2) * dT12;
    float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float xm = e.clamp_x ? 0.f : 1.f, ym = e.clamp_y ? 0.f : 1.f;  // Q7
    float dtx = xm * -cam.fx * tz2 * dJ02;
    float dty = ym * -cam.fy * tz2 * dJ12;
    float dtz = -cam.fx * tz2 * dJ00 - cam.fy * tz2 * dJ11 + (2 * cam.fx * e.t[0]) * tz3 * dJ02 + (2 * cam.fy * e.t[1]) * tz3 * dJ12;
    const float* m = cam.view;  // transformVec4x3Transpose
    dmean3[0] = m[0] * dtx + m[1] * dty + m[2] * dtz;
    dmean3[1] = m[4] * dtx + m[5] * dty + m[6] * dtz;
    dmean3[2] = m[8] * dtx + m[9] * dty + m[10] * dtz;
}

// backward.cu:278-341
void cov3d_grad(const float* scale, float mod, const float* q, const float* dcov6, float* dscale3, float* drot4) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    M3 R = quat_to_rot(q);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    // Mm(k,i) = s_k R(i,k) as in cov3d_from_scale_rot
    M3 Mm;
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 3; i++) Mm.v[k][i] = s[k] * R.v[i][k];
    M3 dS;
    dS.v[0][0] = dcov6[0]; dS.v[0][1] = 0.5f * dcov6[1]; dS.v[0][2] = 0.5f * dcov6[2];
    dS.v[1][0] = 0.5f * dcov6[1]; dS.v[1][1] = dcov6[3]; dS.v[1][2] = 0.5f * dcov6[4];
    dS.v[2][0] = 0.5f * dcov6[2]; dS.v[2][1] = 0.5f * dcov6[4]; dS.v[2][2] = dcov6[5];
    // dL_dM = 2.0f * M * dL_dSigma (GLM: scalar*matrix first, then product)
    M3 M2;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) M2.v[a][b] = 2.0f * Mm.v[a][b];
    M3 dM = mul(M2, dS);
    // reference: Rt = transpose(R_glm), dL_dMt = transpose(dL_dM); X[i] is GLM column i.
    // Rt[i] (col i of R_glm^T) = row i of R_glm (math) = (R_glm(i,0..2)) = (R(0..2, i)) std.
    // dL_dMt[i] = col i of dM^T = row i of dM (math).
    float dMt[3][3];  // dMt[i][j] = GLM dL_dMt[i][j] = dM(i, j) math
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) dMt[i][j] = dM.v[i][j];
    for (int i = 0; i < 3; i++)
        dscale3[i] = R.v[0][i] * dMt[i][0] + R.v[1][i] * dMt[i][1] + R.v[2][i] * dMt[i][2];
    for (int j = 0; j < 3; j++) { dMt[0][j] *= s[0]; dMt[1][j] *= s[1]; dMt[2][j] *= s[2]; }
    drot4[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
    drot4[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
    drot4[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
    drot4[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
}

// backward.cu:20-139
void sh_grad(int deg, int M, const float* mean, const float* campos, const float* sh_all, const uint8_t* clamped,
             const float* dcolor3, float* dmean3 /* += */, float* dsh /* (M,3) assign for used coeffs */) {
    V3 dir_o = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
    float len = std::sqrt(dot3(dir_o, dir_o));
    V3 dir = {dir_o.x / len, dir_o.y / len, dir_o.z / len};
    const V3* sh = reinterpret_cast<const V3*>(sh_all);
    V3 g = {dcolor3[0] * (clamped[0] ? 0.f : 1.f), dcolor3[1] * (clamped[1] ? 0.f : 1.f), dcolor3[2] * (clamped[2] ? 0.f : 1.f)};
    V3 ddx = {0, 0, 0}, ddy = {0, 0, 0}, ddz = {0, 0, 0};
    float x = dir.x, y = dir.y, z = dir.z;
    V3* out = reinterpret_cast<V3*>(dsh);
    (void)M;
    out[0] = K0 * g;
    if (deg > 0) {
        out[1] = (-K1 * y) * g; out[2] = (K1 * z) * g; out[3] = (-K1 * x) * g;
        ddx = (-K1) * sh[3]; ddy = (-K1) * sh[1]; ddz = K1 * sh[2];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            out[4] = (K2[0] * xy) * g; out[5] = (K2[1] * yz) * g; out[6] = (K2[2] * (2.f * zz - xx - yy)) * g;
            out[7] = (K2[3] * xz) * g; out[8] = (K2[4] * (xx - yy)) * g;
            ddx = ddx + ((K2[0] * y) * sh[4] + (K2[2] * 2.f * -x) * sh[6] + (K2[3] * z) * sh[7] + (K2[4] * 2.f * x) * sh[8]);
            ddy = ddy + ((K2[0] * x) * sh[4] + (K2[1] * z) * sh[5] + (K2[2] * 2.f * -y) * sh[6] + (K2[4] * 2.f * -y) * sh[8]);
            ddz = ddz + ((K2[1] * y) * sh[5] + (K2[2] * 2.f * 2.f * z) * sh[6] + (K2[3] * x) * sh[7]);
            if (deg > 2) {
                out[9] = ((K3[0] * y) * (3.f * xx - yy)) * g;
                out[10] = ((K3[1] * xy) * z) * g;
                out[11] = ((K3[2] * y) * (4.f * zz - xx - yy)) * g;
                out[12] = ((K3[3] * z) * (2.f * zz - 3.f * xx - 3.f * yy)) * g;
                out[13] = ((K3[4] * x) * (4.f * zz - xx - yy)) * g;
                out[14] = ((K3[5] * z) * (xx - yy)) * g;
                out[15] = ((K3[6] * x) * (xx - 3.f * yy)) * g;
                ddx = ddx + ((((K3[0] * sh[9]) * 3.f) * 2.f) * xy + (K3[1] * sh[10]) * yz + ((K3[2] * sh[11]) * -2.f) * xy +
                             (((K3[3] * sh[12]) * -3.f) * 2.f) * xz + (K3[4] * sh[13]) * (-3.f * xx + 4.f * zz - yy) +
                             ((K3[5] * sh[14]) * 2.f) * xz + ((K3[6] * sh[15]) * 3.f) * (xx - yy));
                ddy = ddy + (((K3[0] * sh[9]) * 3.f) * (xx - yy) + (K3[1] * sh[10]) * xz + (K3[2] * sh[11]) * (-3.f * yy + 4.f * zz - xx) +
                             (((K3[3] * sh[12]) * -3.f) * 2.f) * yz + ((K3[4] * sh[13]) * -2.f) * xy + ((K3[5] * sh[14]) * -2.f) * yz +
                             (((K3[6] * sh[15]) * -3.f) * 2.f) * xy);
                ddz = ddz + ((K3[1] * sh[10]) * xy + (((K3[2] * sh[11]) * 4.f) * 2.f) * yz + ((K3[3] * sh[12]) * 3.f) * (2.f * zz - xx - yy) +
                             (((K3[4] * sh[13]) * 4.f) * 2.f) * xz + (K3[5] * sh[14]) * (xx - yy));
            }
        }
    }
    V3 ddir = {dot3(ddx, g), dot3(ddy, g), dot3(ddz, g)};
    // dnormvdv (auxiliary.h:107-118)
    V3 v = dir_o;
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float inv32 = 1.0f / std::sqrt(sum2 * sum2 * sum2);
    dmean3[0] += ((+sum2 - v.x * v.x) * ddir.x - v.y * v.x * ddir.y - v.z * v.x * ddir.z) * inv32;
    dmean3[1] += (-v.x * v.y * ddir.x + (sum2 - v.y * v.y) * ddir.y - v.z * v.y * ddir.z) * inv32;
    dmean3[2] += (-v.x * v.z * ddir.x - v.y * v.z * ddir.y + (sum2 - v.z * v.z) * ddir.z) * inv32;
}

}  // namespace

// ======================= C ABI (loaded with ctypes) ============================
extern "C" {

void* f3dgs_oracle_create() { return new State(); }
void f3dgs_oracle_destroy(void* h) { delete static_cast<State*>(h); }

// Mirrors f3dgs_forward (include/f3dgs.h) with HOST pointers; returns num_rendered.
int f3dgs_oracle_forward(void* h, int P, int D, int M, int C, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* colors_precomp,
                         const float* semantic_feature, const float* opacities, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                         float tan_fovy, float* out_color, float* out_feature_map, float* out_depth, int* radii) {
    State& s = *static_cast<State*>(h);
    s.P = P; s.D = D; s.M = M; s.C = C; s.W = width; s.H = height;
    s.gx = (width + TILE - 1) / TILE; s.gy = (height + TILE - 1) / TILE;
    Cam cam{viewmatrix, projmatrix, tan_fovx, tan_fovy, width / (2.0f * tan_fovx), height / (2.0f * tan_fovy),
            width, height, s.gx, s.gy};
    const size_t HW = (size_t)width * height;
    std::fill(out_color, out_color + 3 * HW, 0.f);
    std::fill(out_depth, out_depth + HW, 0.f);
    std::fill(out_feature_map, out_feature_map + (size_t)C * HW, 0.f);
    if (radii) std::fill(radii, radii + P, 0);
    if (P == 0) { s.N = 0; return 0; }  // rasterize_points.cu:84 skips everything
    stage_project(s, cam, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                  cam_pos);
    if (radii) std::copy(s.radii.begin(), s.radii.end(), radii);
    stage_bin(s);
    const float* colors = colors_precomp ? colors_precomp : s.rgb.data();
    stage_blend(s, background, colors, semantic_feature, out_color, out_feature_map, out_depth);
    return s.N;
}

// Mirrors f3dgs_backward with HOST pointers; uses the state of the last forward.
int f3dgs_oracle_backward(void* h, const float* background, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* scales, float scale_modifier,
                          const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                          const float* dL_dpix, const float* dL_dfeaturepix, const float* dL_depths,
                          float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                          float* dL_dsemantic_feature, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                          float* dL_dscale, float* dL_drot, float* dL_dz) {
    State& s = *static_cast<State*>(h);
    const int P = s.P, M = s.M;
    Cam cam{viewmatrix, projmatrix, tan_fovx, tan_fovy, s.W / (2.0f * tan_fovx), s.H / (2.0f * tan_fovy),
            s.W, s.H, s.gx, s.gy};
    std::fill(dL_dmean3D, dL_dmean3D + 3 * (size_t)P, 0.f);
    std::fill(dL_dcov3D, dL_dcov3D + 6 * (size_t)P, 0.f);
    if (dL_dsh) std::fill(dL_dsh, dL_dsh + 3 * (size_t)M * P, 0.f);
    if (dL_dscale) std::fill(dL_dscale, dL_dscale + 3 * (size_t)P, 0.f);
    if (dL_drot) std::fill(dL_drot, dL_drot + 4 * (size_t)P, 0.f);
    if (P == 0) return 0;
    const float* colors = colors_precomp ? colors_precomp : s.rgb.data();
    stage_blend_grad(s, background, colors, dL_dpix, dL_dfeaturepix, dL_depths, dL_dsemantic_feature);
    for (int i = 0; i < P; i++) {
        if (!(s.radii[i] > 0)) continue;
        const float* mean = means3D + 3 * (size_t)i;
        const float* cov6 = cov3D_precomp ? cov3D_precomp + 6 * (size_t)i : &s.cov3D[6 * (size_t)i];
        float* dmean = dL_dmean3D + 3 * (size_t)i;
        cov2d_grad(mean, cam, cov6, &s.dL_dconic[4 * (size_t)i], dmean, dL_dcov3D + 6 * (size_t)i);
        // backward.cu:372-395: screen-space mean gradient -> 3D mean
        const float* pr = projmatrix;
        float mh[4];
        xform_point4(pr, mean, mh);
        float mw = 1.0f / (mh[3] + 0.0000001f);
        float mul1 = (pr[0] * mean[0] + pr[4] * mean[1] + pr[8] * mean[2] + pr[12]) * mw * mw;
        float mul2 = (pr[1] * mean[0] + pr[5] * mean[1] + pr[9] * mean[2] + pr[13]) * mw * mw;
        const float gx = s.dL_dmean2D[3 * (size_t)i], gy = s.dL_dmean2D[3 * (size_t)i + 1];
        float dm[3];
        dm[0] = (pr[0] * mw - pr[3] * mul1) * gx + (pr[1] * mw - pr[3] * mul2) * gy;
        dm[1] = (pr[4] * mw - pr[7] * mul1) * gx + (pr[5] * mw - pr[7] * mul2) * gy;
        dm[2] = (pr[8] * mw - pr[11] * mul1) * gx + (pr[9] * mw - pr[11] * mul2) * gy;
        const float dz = s.dL_dz[i];
        dm[0] += dz * viewmatrix[2]; dm[1] += dz * viewmatrix[6]; dm[2] += dz * viewmatrix[10];
        dmean[0] += dm[0]; dmean[1] += dm[1]; dmean[2] += dm[2];
        if (shs)
            sh_grad(s.D, M, mean, campos, shs + 3 * (size_t)M * i, &s.clamped[3 * (size_t)i], &s.dL_dcolor[3 * (size_t)i],
                    dmean, dL_dsh + 3 * (size_t)M * i);
        if (scales)
            cov3d_grad(scales + 3 * (size_t)i, scale_modifier, rotations + 4 * (size_t)i, dL_dcov3D + 6 * (size_t)i,
                       dL_dscale + 3 * (size_t)i, dL_drot + 4 * (size_t)i);
    }
    std::copy(s.dL_dmean2D.begin(), s.dL_dmean2D.end(), dL_dmean2D);
    if (dL_dconic) std::copy(s.dL_dconic.begin(), s.dL_dconic.end(), dL_dconic);
    std::copy(s.dL_dopacity.begin(), s.dL_dopacity.end(), dL_dopacity);
    std::copy(s.dL_dcolor.begin(), s.dL_dcolor.end(), dL_dcolor);
    if (dL_dz) std::copy(s.dL_dz.begin(), s.dL_dz.end(), dL_dz);
    return 0;
}

void f3dgs_oracle_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
    for (int i = 0; i < P; i++) {
        float pv[3];
        xform_point(viewmatrix, means3D + 3 * (size_t)i, pv);
        present[i] = pv[2] > 0.2f;
    }
}

// Copy an intermediate of the last forward to `dst` (count = elements of the
// native type); returns elements available, or -1 for an unknown name.
long f3dgs_oracle_read(void* h, const char* what, void* dst, long count) {
    State& s = *static_cast<State*>(h);
    std::string w(what);
    auto put = [&](const void* src, size_t elems, size_t esz) -> long {
        if (dst) std::memcpy(dst, src, std::min((size_t)count, elems) * esz);
        return (long)elems;
    };
    if (w == "means2D") return put(s.means2D.data(), s.means2D.size(), 4);
    if (w == "depths") return put(s.depths.data(), s.depths.size(), 4);
    if (w == "cov3D") return put(s.cov3D.data(), s.cov3D.size(), 4);
    if (w == "conic_opacity") return put(s.conic_opacity.data(), s.conic_opacity.size(), 4);
    if (w == "rgb") return put(s.rgb.data(), s.rgb.size(), 4);
    if (w == "clamped") return put(s.clamped.data(), s.clamped.size(), 1);
    if (w == "radii") return put(s.radii.data(), s.radii.size(), 4);
    if (w == "tiles_touched") return put(s.tiles_touched.data(), s.tiles_touched.size(), 4);
    if (w == "point_list") return put(s.point_list.data(), s.point_list.size(), 4);
    if (w == "keys") return put(s.keys.data(), s.keys.size(), 8);
    if (w == "ranges") return put(s.ranges.data(), s.ranges.size(), 4);
    if (w == "final_T") return put(s.final_T.data(), s.final_T.size(), 4);
    if (w == "n_contrib") return put(s.n_contrib.data(), s.n_contrib.size(), 4);
    return -1;
}


// Work statistics of the last forward (development aid for sizing the HIP kernels): for every 8x8
// quadrant of every tile and every list entry the quadrant has to visit (position < the quadrant's
// deepest n_contrib), count visited entries, entries where at least one pixel blends, and blending pixels.
void f3dgs_oracle_work_stats(void* h, double* out /* [8] */) {
    State& s = *static_cast<State*>(h);
    double q_visit = 0, q_active = 0, px_blend = 0, t_visit = 0, t_active = 0, px_eval = 0, h_visit = 0, h_active = 0;
    for (int ty = 0; ty < s.gy; ty++)
        for (int tx = 0; tx < s.gx; tx++) {
            const uint32_t lo = s.ranges[2 * ((size_t)ty * s.gx + tx)];
            uint32_t tmax = 0, qmax[4] = {0, 0, 0, 0};
            for (int py = ty * TILE; py < std::min(s.H, (ty + 1) * TILE); py++)
                for (int px = tx * TILE; px < std::min(s.W, (tx + 1) * TILE); px++) {
                    const uint32_t n = s.n_contrib[(size_t)py * s.W + px];
                    const int q = ((py - ty * TILE) / 8) * 2 + ((px - tx * TILE) / 8);
                    qmax[q] = std::max(qmax[q], n);
                    tmax = std::max(tmax, n);
                }
            t_visit += tmax;
            h_visit += std::max(qmax[0], qmax[1]) + std::max(qmax[2], qmax[3]);
            for (uint32_t k = 0; k < tmax; k++) {
                const uint32_t g = s.point_list[lo + k];
                const float* co = &s.conic_opacity[4 * (size_t)g];
                int qcnt[4] = {0, 0, 0, 0};
                for (int py = ty * TILE; py < std::min(s.H, (ty + 1) * TILE); py++)
                    for (int px = tx * TILE; px < std::min(s.W, (tx + 1) * TILE); px++) {
                        const uint32_t n = s.n_contrib[(size_t)py * s.W + px];
                        if (k >= n) continue;
                        px_eval += 1;
                        const float dx = s.means2D[2 * (size_t)g] - (float)px, dy = s.means2D[2 * (size_t)g + 1] - (float)py;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float alpha = std::min(0.99f, co[3] * std::exp(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        qcnt[((py - ty * TILE) / 8) * 2 + ((px - tx * TILE) / 8)]++;
                    }
                int tot = 0;
                for (int q = 0; q < 4; q++) {
                    if (k < qmax[q]) { q_visit += 1; if (qcnt[q]) q_active += 1; }
                    tot += qcnt[q];
                }
                px_blend += tot;
                if (tot) t_active += 1;
                if (k < std::max(qmax[0], qmax[1]) && (qcnt[0] + qcnt[1])) h_active += 1;
                if (k < std::max(qmax[2], qmax[3]) && (qcnt[2] + qcnt[3])) h_active += 1;
            }
        }
    out[0] = q_visit; out[1] = q_active; out[2] = px_blend; out[3] = t_visit; out[4] = t_active; out[5] = px_eval;
    out[6] = h_visit; out[7] = h_active;
}

// Sub-step entry points so that the reference's own Python fall-backs can pin them
// (tests/golden/reference_fallbacks.npz).
void f3dgs_oracle_sh_to_rgb(int P, int deg, int M, const float* means3D, const float* campos, const float* shs,
                            float* rgb_out, uint8_t* clamped_out) {
    for (int i = 0; i < P; i++) {
        uint8_t cl[3];
        V3 c = sh_to_rgb(deg, M, means3D + 3 * (size_t)i, campos, shs + 3 * (size_t)M * i, cl);
        rgb_out[3 * (size_t)i] = c.x; rgb_out[3 * (size_t)i + 1] = c.y; rgb_out[3 * (size_t)i + 2] = c.z;
        if (clamped_out) std::memcpy(clamped_out + 3 * (size_t)i, cl, 3);
    }
}
void f3dgs_oracle_cov3d(int P, const float* scales, float mod, const float* rotations, float* cov6_out) {
    for (int i = 0; i < P; i++)
        cov3d_from_scale_rot(scales + 3 * (size_t)i, mod, rotations + 4 * (size_t)i, cov6_out + 6 * (size_t)i);
}

}  // extern "C"
