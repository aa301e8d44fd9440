// render_bwd_pl.hip — blend backward, PIXEL-LANE formulation (the default for feature_mfma = 1).
//
// Semantics: R/cuda_rasterizer/backward.cu:407-620 (R = submodules/diff-gaussian-rasterization-feature), quirks Q1, Q2,
// Q3, Q5, Q8 as in render_bwd.hip.
//
// The instance-lane kernel (render_bwd.hip) pays for the list-order dependence with two DPP prefix scans per pixel and
// chunk and runs every per-(pixel, Gaussian) multiply-add on the vector pipe.  Here the two halves of the problem are
// separated and each runs where it is cheap:
//
//   phase 1 (vector pipe, lane = PIXEL, one 8x8 quadrant per wave): the wave walks a chunk of 16 compacted list
//     entries back to front, one entry at a time; transmittance T and the "colour behind" sum S are per-lane running
//     scalars, so the recurrences are two multiply-adds - no scans.  Per entry and pixel it produces two numbers,
//         w = alpha T_front            (blend weight:   every dL/d{colour, depth, feature} sum is  sum_px w  x  dL/dpixel)
//         s = op G dL/dalpha           (every geometric sum is a moment  sum_px s x {1, x, y, x^2, xy, y^2}),
//     and stores them in two 16 x 64 LDS tiles (pixel-permuted columns, see PlLds).
//   phase 2 (matrix pipe): every per-Gaussian sum of the chunk is ONE contraction over the 64 pixels,
//         [16 entries x 64 px] x [64 px x (C + 4 + 6)]  ->  exact-fp32 v_mfma_f32_16x16x4_f32,
//     whose B operands (dL/dfeature, dL/dcolour, dL/ddepth of the wave's pixels and the pixel monomials) are loaded ONCE
//     per wave into registers in operand layout - the dL/dfeature tile never touches LDS (this is what capped the
//     instance-lane kernel at three waves per SIMD).  The A operands come back from the two LDS tiles with four
//     conflict-free 16-byte reads each.
//   flush: feature sums leave the accumulators as 64-byte runs (lane = channel) of coalesced atomics; the ten geometric
//     sums are rebuilt from the moments about the block centre (dx = (mean - centre) - u) by 16 lanes.
//
// Work per (entry, wave): ~28 vector instructions (instance-lane kernel: 63 per 64 pairs) and 4 matrix instructions
// (128 matrix-pipe cycles at C = 32); the two pipes overlap across the waves of a SIMD.

#include "render_common.h"

namespace f3dgs {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef F3DGS_DEV
#define PL_DEV_SKIP(bit) (a.dev & (bit))     // 1: no atomics  2: no phase 1/2  4: no matrix instructions  16: no geometric atomics  32: no feature atomics
#define PL_PHASE_BEGIN() unsigned long long ph_t0_ = (a.dev & 8) ? __builtin_readcyclecounter() : 0ull; unsigned long long cyc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PL_PHASE_END(K) do { if (a.dev & 8) { const unsigned long long t1_ = __builtin_readcyclecounter(); cyc_[K] += t1_ - ph_t0_; ph_t0_ = t1_; } } while (0)
#define PL_COUNT(K, N) do { if (a.dev & 8) cyc_[K] += (N); } while (0)
#else
#define PL_DEV_SKIP(bit) false
#define PL_PHASE_BEGIN() do {} while (0)
#define PL_PHASE_END(K) do {} while (0)
#define PL_COUNT(K, N) do {} while (0)
#endif

constexpr int PL_CAP = 16;         // list entries per chunk = rows of one 16x16x4 matrix instruction
constexpr int PL_ROW = 68;         // dwords per pixel block row of the A tiles (16 entries x 4 px + one 4-dword skew slot)

// One staged (compacted) list entry: three broadcast reads per entry in phase 1.
struct PlEnt {
    float4 q0;   // mean_x, mean_y, conic_a', conic_b'   (conic pre-scaled, see splat_power2)
    float4 q1;   // conic_c', opacity, list position (bits), Gaussian index (bits)
    float4 q2;   // red, green, blue, depth
};

// A tiles: element (entry i, column c) with c = 16 u + 4 k + m - the pixel that matrix step t = 4 u + m contracts at
// K index k - lives at dword (c >> 2) * 68 + (i + 1 - ((c >> 2) & 1)) * 4 + (c & 3): the 16 entries of one pixel block
// are 16 bytes apart, so the operand read of lane (i, k) - one 16-byte read per u - meets the other lanes of its
// 16-lane service group on sixteen different 4-bank groups (the one-slot skew between even and odd pixel blocks lines
// the two k values of a group up), and the phase-1 stores (64 pixels of one entry) are at most 2-way conflicted (free).
struct PlLds {
    PlEnt ent[PL_CAP];
    float wt[16 * PL_ROW];      // blend weights; reused as the 16 x 17 transpose tile of the geometric sums in the flush
    float st[16 * PL_ROW];      // s = op G dL/dalpha
};

template <int NCB, bool GEO>
__device__ __forceinline__ void render_backward_pl_body(const BwdArgs& a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PlLds& L = *reinterpret_cast<PlLds*>(smem);
    const int lane = threadIdx.x;
    PL_PHASE_BEGIN();     // [0] staging  [1] walk  [2] phase 1  [3] phase 2  [4] flush  [5] chunks  [6] entries

    const uint32_t wg = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t ntiles = gridDim.x / 4;
    const uint32_t tile = a.part_major ? wg % ntiles : wg / 4;
    const int part = a.part_major ? wg / ntiles : wg % 4;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const uint2 rg = a.ranges[tile];
    const uint32_t r_lo = __builtin_amdgcn_readfirstlane((int)rg.x);
    const size_t HW = (size_t)a.W * a.H;
    const int px0 = tx * TILE + (part & 1) * 8, py0 = ty * TILE + (part >> 1) * 8;

    // ---- per-pixel state (lane = pixel (lane & 7, lane >> 3) of the quadrant) ---------------------------------
    const int lx = lane & 7, ly = lane >> 3;
    const int x = px0 + lx, y = py0 + ly;
    const bool inside = x < a.W && y < a.H;
    const size_t pid = inside ? (size_t)y * a.W + x : 0;      // outside the image: a valid address, masked below
    uint32_t last = a.n_contrib[pid];
    if (!inside) last = 0;
    const uint32_t max_last = wave_max_u32(last);
    auto load_ids = [&](int k0w) -> uint32_t {
        const uint32_t pos = (uint32_t)(k0w + 63 - lane);
        return (k0w >= 0 && pos < max_last) ? a.point_list[r_lo + pos] : 0u;
    };
    const int k_top = (int)((max_last + 63) / 64) * 64 - 64;
    uint32_t ngid = load_ids(k_top), fgid = load_ids(k_top - 64);

    float T = a.final_T[pid];
    float dR = a.dL_dpix[pid], dG = a.dL_dpix[HW + pid], dB = a.dL_dpix[2 * HW + pid], dD = a.dL_ddepth[pid];
    if (!inside) { T = 0.f; dR = dG = dB = dD = 0.f; }
    float S = T * (a.bg[0] * dR + a.bg[1] * dG + a.bg[2] * dB);
    const float pxf = (float)x, pyf = (float)y;

    // ---- matrix-pipe B operands, resident for the life of the wave (lane = (column col, K index kk)) -------------
    // step t contracts the four pixels (4 (t & 1) + kk, t >> 1), kk = 0..3: 16-byte runs of the planar gradient images
    const int col = lane & 15, kk = lane >> 4;
    float Bf[NCB > 0 ? NCB : 1][16];
    float Bgw[GEO ? 16 : 1], Bgs[GEO ? 16 : 1];
    {
        const int bx0 = px0 + kk;
        const float* gsrc = col < 3 ? a.dL_dpix + (size_t)col * HW : a.dL_ddepth;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int bx = bx0 + 4 * (t & 1), by = py0 + (t >> 1);
            const bool in = bx < a.W && by < a.H;
            const size_t pb = in ? (size_t)by * a.W + bx : 0;
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                const int ch = 16 * cb + col;
                const float v = a.dL_dfeat[(size_t)(a.c0 + min(ch, a.nc - 1)) * HW + pb];
                Bf[cb][t] = (in && ch < a.nc) ? v : 0.f;
            }
            if constexpr (GEO) {
                float v = 0.f;
                if (col < 4) v = gsrc[pb];
                Bgw[t] = (in && col < 4) ? v : 0.f;
                // monomials of the pixel offset from the block centre, columns 4..9: 1, u, v, u^2, uv, v^2
                const float uu = (float)(4 * (t & 1) + kk) - 3.5f, vv = (float)(t >> 1) - 3.5f;
                Bgs[t] = col == 4 ? 1.f : col == 5 ? uu : col == 6 ? vv : col == 7 ? uu * uu : col == 8 ? uu * vv : col == 9 ? vv * vv : 0.f;
            }
        }
    }
    // LDS offsets (dwords): phase-1 store column of this lane's pixel, operand read base of this lane's (entry, K index)
    const int ccol = 16 * (ly >> 1) + 4 * (lx & 3) + 2 * (ly & 1) + (lx >> 2);
    const int wofs = (ccol >> 2) * PL_ROW + (1 - ((ccol >> 2) & 1)) * 4 + (ccol & 3);
    const int rofs = kk * PL_ROW + (col + 1 - (kk & 1)) * 4;
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
    const float cx0 = (float)px0 + 3.5f, cy0 = (float)py0 + 3.5f;

    // ---- one chunk of n <= 16 staged entries (L.ent[0..n), back to front) -------------------------------------------
    auto process = [&](const int n, const uint32_t pos_min) {
        PL_PHASE_END(1);
        PL_COUNT(5, 1); PL_COUNT(6, n);
        __builtin_amdgcn_wave_barrier();
        if (__ballot(last > pos_min) == 0) return;        // every pixel of the wave ended behind this chunk
        if (PL_DEV_SKIP(2)) return;
        // phase 1
        uint32_t tm = 0;      // entries that blended somewhere in the wave
#pragma unroll 2
        for (int i = 0; i < n; i++) {
            const float4 e0 = L.ent[i].q0, e1 = L.ent[i].q1;
            const float dx = e0.x - pxf, dy = e0.y - pyf;
            const float power2 = splat_power2(dx, dy, e0.z, e0.w, e1.x);
            const float v = e1.y * __builtin_amdgcn_exp2f(power2);                  // op G, not yet clamped
            const bool ok = (int)(__float_as_uint(e1.z) < last) & (int)!(power2 > 0.0f) & (int)!(v < ALPHA_MIN);
            const float au = ok ? v : 0.f;          // exp2 may be inf where power > 0: selected away, never multiplied
            const float al = fminf(ALPHA_MAX, au);
            const float f = __builtin_amdgcn_rcpf(1.f - al);                        // exactly 1 for skipped pairs
            const float Tb = T * f;                 // transmittance in front of this splat
            const float w = al * Tb;
            L.wt[wofs + 4 * i] = w;
            if constexpr (GEO) {
                const float4 e2 = L.ent[i].q2;
                const float q = fmaf(e2.x, dR, fmaf(e2.y, dG, fmaf(e2.z, dB, e2.w * dD)));
                const float dL_dalpha = fmaf(Tb, q, -(S * f));
                S = fmaf(w, q, S);
                L.st[wofs + 4 * i] = au * dL_dalpha;
            }
            T = Tb;
            if (__ballot(ok)) tm |= 1u << i;
        }
        __builtin_amdgcn_wave_barrier();
        PL_PHASE_END(2);
        if (tm == 0) return;

        // phase 2: all sums of the chunk on the matrix pipe
        f32x4 accf[NCB > 0 ? NCB : 1], accw, accs;
#pragma unroll
        for (int cb = 0; cb < (NCB > 0 ? NCB : 1); cb++) accf[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        accw = accs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float4 wa = *reinterpret_cast<const float4*>(&L.wt[rofs + u * 4 * PL_ROW]);
            float4 sa = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (GEO) sa = *reinterpret_cast<const float4*>(&L.st[rofs + u * 4 * PL_ROW]);
            const float wv[4] = {wa.x, wa.y, wa.z, wa.w}, sv[4] = {sa.x, sa.y, sa.z, sa.w};
            if (PL_DEV_SKIP(4)) { accw[0] += wv[0] + wv[1] + wv[2] + wv[3] + sv[0] + sv[1] + sv[2] + sv[3]; continue; }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int t = 4 * u + m;
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    accf[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[m], Bf[cb][t], accf[cb], 0, 0, 0);
                    if constexpr (GEO) {
                        if (cb == 0) accw = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[m], Bgw[t], accw, 0, 0, 0);
                        if (cb == NCB - 1) accs = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[m], Bgs[t], accs, 0, 0, 0);
                    }
                }
                if constexpr (GEO && NCB == 0) {
                    accw = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[m], Bgw[t], accw, 0, 0, 0);
                    accs = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[m], Bgs[t], accs, 0, 0, 0);
                }
            }
        }

        PL_PHASE_END(3);
        // flush.  D[i][j]: lane holds column j = lane & 15, register r holds row (entry) i = 4 (lane >> 4) + r.
        uint32_t gidr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) gidr[r] = __float_as_uint(L.ent[4 * kk + r].q1.w);
        if (PL_DEV_SKIP(1)) tm = 0;
        if constexpr (NCB > 0) if (!PL_DEV_SKIP(32)) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (!((tm >> (4 * kk + r)) & 1u)) continue;
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    const int ch = 16 * cb + col;
                    if (ch < a.nc) unsafeAtomicAdd(a.dL_dfeature + (size_t)gidr[r] * a.C + a.c0 + ch, accf[cb][r]);
                }
            }
        }
        if constexpr (GEO) {
            __builtin_amdgcn_wave_barrier();       // the operand reads of L.wt are done: reuse it as the 16 x 17 transpose tile
#pragma unroll
            for (int r = 0; r < 4; r++) L.wt[(4 * kk + r) * 17 + col] = accw[r] + accs[r];      // disjoint columns
            __builtin_amdgcn_wave_barrier();
            float o[10];
#pragma unroll
            for (int k = 0; k < 10; k++) o[k] = 0.f;
            if (lane < PL_CAP) {
                const float* g = &L.wt[lane * 17];
                const float4 e0 = L.ent[lane].q0, e1 = L.ent[lane].q1;
                const float ax = e0.x - cx0, ay = e0.y - cy0;       // mean relative to the block centre: dx = ax - u
                const float M0 = g[4], M1x = g[5], M1y = g[6], M2xx = g[7], M2xy = g[8], M2yy = g[9];
                const float m1 = fmaf(ax, M0, -M1x), m2 = fmaf(ay, M0, -M1y);                  // sum s dx, sum s dy
                const float sxx = fmaf(ax, fmaf(ax, M0, -2.f * M1x), M2xx);                    // sum s dx^2
                const float sxy = fmaf(ax, fmaf(ay, M0, -M1y), fmaf(-ay, M1x, M2xy));          // sum s dx dy
                const float syy = fmaf(ay, fmaf(ay, M0, -2.f * M1y), M2yy);                    // sum s dy^2
                const float ca = e0.z * CONIC_UNSCALE_AC, cb = e0.w * CONIC_UNSCALE_B, cc = e1.x * CONIC_UNSCALE_AC;
                // dG/ddelx = -G (a dx + b dy),  dG/da = -G dx^2 / 2, ...;  dL/dopacity = sum G dL/dalpha = M0 / op
                o[0] = -ddelx_dx * fmaf(ca, m1, cb * m2);
                o[1] = -ddely_dy * fmaf(cc, m2, cb * m1);
                o[2] = -0.5f * sxx; o[3] = -0.5f * sxy; o[4] = -0.5f * syy;
                o[5] = M0 * __builtin_amdgcn_rcpf(e1.y);     // only used where the entry blended somewhere => op > 0
                o[6] = g[0]; o[7] = g[1]; o[8] = g[2]; o[9] = g[3];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < PL_CAP) {
#pragma unroll
                for (int k = 0; k < 10; k++) L.wt[lane * 17 + k] = o[k];
            }
            __builtin_amdgcn_wave_barrier();
            // four entries per atomic instruction, ten consecutive floats of the gradient record each
            if (a.write_base && !PL_DEV_SKIP(16)) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int e = 4 * r + kk;
                    if (col < 10 && ((tm >> e) & 1u)) {
                        const uint32_t gg = __float_as_uint(L.ent[e].q1.w);
                        unsafeAtomicAdd(a.grec + (size_t)gg * GREC + col, L.wt[e * 17 + col]);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        PL_PHASE_END(4);
    };

    // ---- walk the list back to front in windows of 64 positions; entries whose 1/255 footprint misses this wave's pixel
    // block are dropped (rect_hit, exact-safe), the survivors are appended to the staged chunk ----------------------------
    const float wx0 = (float)px0, wx1 = (float)(px0 + 7), wy0 = (float)py0, wy1 = (float)(py0 + 7);
    int count = 0;
    uint32_t cur_min = 0;
    PL_PHASE_END(0);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // three-stage software pipeline over the windows: while window k0 is tested / appended / processed, the splat records of
    // window k0 - 64 and the list ids of window k0 - 128 are in flight
    auto load_recs = [&](int k0w, uint32_t gid, float (&f)[10], bool& have) {
        const uint32_t pos = (uint32_t)(k0w + 63 - lane);
        have = k0w >= 0 && pos < max_last;
#pragma unroll
        for (int k = 0; k < 10; k++) f[k] = 0.f;
        if (have) {
            const SplatRec* rp = a.rec + gid;
            const float4 q0 = rp->q0, q1 = rp->q1, q2 = rp->q2;
            f[0] = q0.x; f[1] = q0.y; f[2] = q0.z; f[3] = q0.w; f[4] = q1.x; f[5] = q1.y;
            f[6] = q1.z; f[7] = q1.w; f[8] = q2.x; f[9] = q2.y;
        }
    };
    float nf[10];
    bool nhave;
    load_recs(k_top, ngid, nf, nhave);
    for (int k0 = k_top;; k0 -= 64) {
        const bool drain = k0 < 0;                      // one extra trip flushes the last, partially filled chunk
        const uint32_t pos = (uint32_t)(k0 + 63 - lane);    // lane 0 = farthest back within the window
        float f[10];
#pragma unroll
        for (int k = 0; k < 10; k++) f[k] = nf[k];
        const uint32_t gid = ngid;
        const bool have = nhave && !drain;
        if (!drain) {
            ngid = fgid;
            load_recs(k0 - 64, ngid, nf, nhave);
            fgid = load_ids(k0 - 128);
        }
        const bool hit = have && (a.no_wave_cull || rect_hit(f[0], f[1], f[2], f[3], f[4], f[5], wx0, wx1, wy0, wy1));
        const unsigned long long hmask = __ballot(hit);
        const int c2 = __popcll(hmask);
        const int rank = __popcll(hmask & lt_mask);
        int first = 0;
        while (true) {
            if (count == PL_CAP || (drain && count > 0)) {
                process(count, cur_min);
                count = 0;
            }
            if (first >= c2) break;
            const int n = min(c2 - first, PL_CAP - count);
            if (hit && rank >= first && rank < first + n) {
                PlEnt en;
                en.q0 = make_float4(f[0], f[1], f[2] * CONIC_SCALE_AC, f[3] * CONIC_SCALE_B);   // see splat_power2
                en.q1 = make_float4(f[4] * CONIC_SCALE_AC, f[5], __uint_as_float(pos), __uint_as_float(gid));
                en.q2 = make_float4(f[6], f[7], f[8], f[9]);
                L.ent[count + rank - first] = en;
            }
            count += n;
            first += n;
            cur_min = (uint32_t)k0;      // every survivor of this window sits at a position >= k0
        }
        if (drain) break;
    }
    PL_PHASE_END(1);
#ifdef F3DGS_DEV
    if ((a.dev & 8) && lane == 0) {
        cyc_[7] = 1;
        for (int k = 0; k < 8; k++) a.dev_cycles[(size_t)blockIdx.x * 8 + k] = cyc_[k];     // private slots: summed on the host
    }
#endif
}

template <int NCB, bool GEO>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) render_backward_pl_kernel(BwdArgs a) {
    render_backward_pl_body<NCB, GEO>(a);
}

template <int NCB, bool GEO>
void launch_pl(const BwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((render_backward_pl_kernel<NCB, GEO>), dim3(a.gx * a.gy * 4), dim3(64), sizeof(PlLds), s, a);
}

}  // namespace

// Channel windows: the first carries the geometric sums and up to 32 channels, later ones up to 64 channels.
void launch_render_backward_pl(BwdArgs a, int C, hipStream_t s) {
    if (C == 0) {
        a.c0 = 0; a.nc = 0; a.write_base = 1;
        launch_pl<0, true>(a, s);
        return;
    }
    for (int c0 = 0; c0 < C;) {
        a.c0 = c0; a.write_base = (c0 == 0);
        if (c0 == 0) {
            a.nc = min(32, C);
            if (a.nc <= 16) launch_pl<1, true>(a, s); else launch_pl<2, true>(a, s);
        } else {
            a.nc = min(64, C - c0);
            if (a.nc <= 16) launch_pl<1, false>(a, s);
            else if (a.nc <= 32) launch_pl<2, false>(a, s);
            else launch_pl<4, false>(a, s);
        }
        c0 += a.nc;
    }
}

}  // namespace f3dgs
