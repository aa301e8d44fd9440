// feature_loss.hip - fused feature-map loss: bilinear resize (align_corners) -> optional 1x1-conv decoder -> L1,
// forward and backward.
//
// Replaces, around the rasterizer, the reference's (train.py:99-105, models/networks.py:107-119,
// utils/loss_utils.py:17-18)
//     feature_map = F.interpolate(feature_map[None], size=(Hg, Wg), mode='bilinear', align_corners=True)[0]
//     if speedup: feature_map = cnn_decoder(feature_map)          # nn.Conv2d(C, Cout = 4C, kernel_size=1)
//     Ll1_feature = torch.abs(feature_map - gt).mean()
// and their autograd backward.  Stages (N = Hg*Wg output pixels):
//   K1 resize      (C,H,W) -> X[N][C] pixel-major (the A operand layout of the contraction); without a decoder the
//                  L1 residual is taken right here and X never exists
//   K2 decoder     y = X W^T + b on the fp32 matrix pipe, 64 pixels x all Cout per workgroup; the residual against gt,
//                  the loss partial and g_y = sign(r)/(N Cout) are formed in the accumulator registers - the decoded
//                  (Cout,Hg,Wg) map (354 MB for LSeg at 360x480) is never written; g_x = g_y W on the matrix pipe
//                  from an LDS copy of g_y; only the SIGNS of r leave the kernel (1 byte per element)
//   K3 dW, db      dW = g_y^T X, db = sum g_y: split over pixel ranges, A operand expanded from the sign bytes
//   K4 resize^T    dL/dfeature_map (C,H,W) gathered from g_x (no atomics: every source pixel sums its own few outputs)
// The decoder is the one dense contraction next to the rasterizer (v_mfma_f32_32x32x2_f32: exact fp32, A/B one
// value per lane, A[i = l&31][k = l>>5], B[k = l>>5][n = l&31], D column l&31, rows (r&3)+8(r>>2)+4(l>>5)).

#include <hip/hip_fp16.h>

#include "common.h"
#include "resize_taps.h"

namespace f3dgs {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---- K1: resize into pixel-major X; without a decoder (gt != nullptr) the L1 residual is taken here -------------
// grid (ceil(N / 64), ceil(C / 32)); LDS tile [32 channels][64 pixels]
__global__ void __launch_bounds__(256)
fl_resize_kernel(ResizeGeom g, int C, const float* __restrict__ fm, float* __restrict__ X, const float* __restrict__ gt,
                 float inv_n, float* __restrict__ loss_partial) {
    __shared__ float t[32][65];
    __shared__ float lsum[4];
    const int N = g.Hg * g.Wg;
    const int p0 = blockIdx.x * 64, cb = blockIdx.y * 32;
    {
        const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
        const int p = p0 + px;
        float loss = 0.f;
        if (p < N) {
            const int yo = p / g.Wg, xo = p - yo * g.Wg;
            int y0, y1, x0, x1;
            float ly0, ly1, lx0, lx1;
            taps(yo, g.sy, g.H, y0, y1, ly0, ly1);
            taps(xo, g.sx, g.W, x0, x1, lx0, lx1);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int c = cb + cg * 8 + k;
                float v = 0.f;
                if (c < C) {
                    const float* pl = fm + (size_t)c * g.H * g.W;
                    v = ly0 * (lx0 * pl[y0 * g.W + x0] + lx1 * pl[y0 * g.W + x1]) +
                        ly1 * (lx0 * pl[y1 * g.W + x0] + lx1 * pl[y1 * g.W + x1]);
                    if (gt) {       // no decoder: residual, loss, gradient w.r.t. the resized map
                        const float r = v - gt[(size_t)c * N + p];
                        loss += fabsf(r);
                        v = r > 0.f ? inv_n : (r < 0.f ? -inv_n : 0.f);
                    }
                }
                t[cg * 8 + k][px] = v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) t[cg * 8 + k][px] = 0.f;
        }
        if (gt) {
            loss = wave_sum(loss);
            if (px == 0) lsum[cg] = loss;
        }
    }
    __syncthreads();
    {
        const int c = threadIdx.x & 31, pg = threadIdx.x >> 5;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int px = pg * 8 + k;
            if (p0 + px < N && cb + c < C) X[(size_t)(p0 + px) * C + cb + c] = t[c][px];
        }
    }
    if (gt && threadIdx.x == 0) loss_partial[blockIdx.y * gridDim.x + blockIdx.x] = (lsum[0] + lsum[1] + lsum[2] + lsum[3]) * inv_n;
}

// ---- K2: decoder forward + residual + g_x; a WAVE owns 32 pixels for all Cout, a workgroup = 4 waves = 128 pixels ----
// Everything per pixel stays in registers; the only shared data is the current 32-row tile of W (LDS, double-buffered).
// With lane l = (px = l & 31, h = l >> 5) and the K index of an MFMA step s defined as c = h C/2 + s (any bijection
// of K works as long as both operands use it):
//   phase A   y^T[co][px] = W[co][:] . X[px][:]      A = W tile row (b128 LDS reads: four steps per read),
//                                                    B = this lane's half row of X, loaded ONCE into C/2 registers
//   in place  r = y + b - gt,  loss += |r|,  sign byte out,  g = sign(r) / (N Cout)        (accumulator registers)
//   phase B   g_x^T[c][px] += W^T[c][co] g^T[co][px]  A = W tile column (b32 LDS reads), B = the accumulator register r
//             of phase A itself: D's layout (lane = column px, register r = row co(r, h)) IS the B layout when step r
//             contracts co(r, 0) with co(r, 1) - no transpose, no LDS round trip, no barrier between the phases.
// Per W tile a wave issues C MFMAs and 1.25 LDS reads per MFMA; the decoded (Cout, Hg, Wg) map is never written.
// (First version: 64 pixels x 128 Cout per workgroup through three LDS tiles of 132 KB - one workgroup of four waves per
// CU, every operand a 4-byte LDS read behind a barrier: 1.59 ms at 128 -> 512; `profiles/r02_notes.md`.)
template <int C>
struct GemmLds {
    static constexpr int WS = C + 4;          // row stride: 16-byte aligned rows, b128 reads of 32 rows hit all banks evenly
    float Ws[2][32][WS];
    float red[4];
};

__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // 32x32 D layout

template <int C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
fl_decoder_kernel(int N, int Np, int Cout, const float* __restrict__ X, const float* __restrict__ Wd, const float* __restrict__ bias,
                  const float* __restrict__ gt, float inv_n, float* __restrict__ GX, int8_t* __restrict__ S,
                  float* __restrict__ loss_partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GemmLds<C>& L = *reinterpret_cast<GemmLds<C>*>(smem);
    constexpr int NCB = C / 32, HC = C / 2;
    constexpr int LPT = (32 * C / 4) / 256;       // float4 loads per thread for one W tile (C = 32: 1, 64: 2, 128: 4)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int p = blockIdx.x * 128 + 32 * w + li;
    const bool p_ok = p < N;
    // this lane's half row of X
    float xr[HC];
    {
        const float4* src = reinterpret_cast<const float4*>(X + (size_t)(p_ok ? p : 0) * C + h * HC);
#pragma unroll
        for (int k = 0; k < HC / 4; k++) {
            const float4 v = src[k];
            xr[4 * k] = p_ok ? v.x : 0.f; xr[4 * k + 1] = p_ok ? v.y : 0.f; xr[4 * k + 2] = p_ok ? v.z : 0.f; xr[4 * k + 3] = p_ok ? v.w : 0.f;
        }
    }
    f32x16 gx[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) gx[cb][r] = 0.f;
    float loss = 0.f;
    // W tile staging: thread t moves float4 #(t + 256 k) of the 32 x C tile
    auto load_tile = [&](int co0, float4 (&v)[LPT]) {
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int e = (threadIdx.x + 256 * k) * 4, r = e / C, c = e - r * C;
            v[k] = co0 + r < Cout ? *reinterpret_cast<const float4*>(Wd + (size_t)(co0 + r) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf, const float4 (&v)[LPT]) {
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int e = (threadIdx.x + 256 * k) * 4, r = e / C, c = e - r * C;
            *reinterpret_cast<float4*>(&L.Ws[buf][r][c]) = v[k];
        }
    };
    float4 wnext[LPT];
    load_tile(0, wnext);
    store_tile(0, wnext);
    __syncthreads();
    const int ntiles = (Cout + 31) / 32;
    for (int t = 0; t < ntiles; t++) {
        const int co0 = 32 * t, buf = t & 1;
        // the next W tile: in flight during this tile's MFMAs.  C = 128: requested behind phase A instead (phase B alone is 4 k
        // matrix-pipe cycles), so that its sixteen registers never live next to the sixteen ground-truth values - together
        // they put the kernel 21 registers above the 256 of two waves per SIMD (spills inside the tile loop)
        constexpr bool LATE_W = C >= 128;
        if (!LATE_W && t + 1 < ntiles) load_tile(co0 + 32, wnext);
        // ground truth and bias of this tile's rows, requested before phase A
        float gtv[16];
        f32x16 acc;                                              // starts at the bias: y = W x + b
        // addresses as (wave-uniform row base) + (one 32-bit lane offset shared by the sixteen rows): row(r, h) = row(r, 0) + 4 h,
        // so the lane-dependent part, 4 h N + p, does not depend on r (sixteen 64-bit per-lane addresses - and sixteen more for
        // the sign bytes below - would otherwise be kept in registers across the tile)
        const uint32_t gt_lane = (uint32_t)(4 * h) * (uint32_t)N + (uint32_t)(p_ok ? p : 0);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + mfma_row(r, h);
            const bool ok = p_ok && co < Cout;
            const float* grow = gt + (size_t)(co0 + mfma_row(r, 0)) * N;      // uniform
            gtv[r] = ok ? grow[gt_lane] : 0.f;
            acc[r] = co < Cout ? bias[co] : 0.f;
        }
        // ---- phase A (scheduling fences: without them every LDS read of the unrolled loop is hoisted to the top and
        //      the kernel needs > 256 registers)
        const float* wrow = &L.Ws[buf][li][h * HC];
#pragma unroll
        for (int s4 = 0; s4 < HC / 4; s4++) {
            const float4 a = *reinterpret_cast<const float4*>(wrow + 4 * s4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[4 * s4 + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[4 * s4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[4 * s4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[4 * s4 + 3], acc, 0, 0, 0);
            if ((s4 & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- residual, loss, sign, g (in place)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + mfma_row(r, h);
            float gv = 0.f;
            if (p_ok && co < Cout) {
                const float res = acc[r] - gtv[r];
                loss += fabsf(res);
                gv = res > 0.f ? inv_n : (res < 0.f ? -inv_n : 0.f);
                int8_t* srow = S + (size_t)(co0 + mfma_row(r, 0)) * Np;            // uniform
                srow[(uint32_t)(4 * h) * (uint32_t)Np + (uint32_t)p] = res > 0.f ? 1 : (res < 0.f ? -1 : 0);     // co-major: 32 consecutive bytes per half-wave
            }
            acc[r] = gv;
        }
        // ---- phase B
        if (LATE_W && t + 1 < ntiles) load_tile(co0 + 32, wnext);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float* wr = &L.Ws[buf][mfma_row(r, h)][li];
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) gx[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[32 * cb], acc[r], gx[cb], 0, 0, 0);
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < ntiles) store_tile(buf ^ 1, wnext);          // the other buffer: its readers finished a tile ago
        __syncthreads();
    }
    // g_x^T[c][px]: this lane holds column px, register r of block cb = row c = 32 cb + row(r, h): four consecutive c per
    // register quad -> one 16-byte store
    if (p_ok) {
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4*>(GX + (size_t)p * C + 32 * cb + 8 * q + 4 * h) =
                    make_float4(gx[cb][4 * q], gx[cb][4 * q + 1], gx[cb][4 * q + 2], gx[cb][4 * q + 3]);
    }
    loss = wave_sum(loss);
    if (lane == 0) L.red[w] = loss;
    __syncthreads();
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = (L.red[0] + L.red[1] + L.red[2] + L.red[3]) * inv_n;
}

// ---- forward-only decode (inference side, render.py:169-171): phase A of the kernel above, the decoded map written out ----
// y^T[co][px] = W[co][:] . X[px][:] + b[co]; a lane holds column px and rows co(r, h): 32 consecutive pixels per half-wave
// and row -> 128-byte (fp32) or 64-byte (fp16) runs of the (Cout, Hg, Wg) output.
template <int C, bool HALF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
fl_decode_kernel(int N, int Cout, const float* __restrict__ X, const float* __restrict__ Wd, const float* __restrict__ bias,
                 void* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GemmLds<C>& L = *reinterpret_cast<GemmLds<C>*>(smem);
    constexpr int HC = C / 2;
    constexpr int LPT = (32 * C / 4) / 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int p = blockIdx.x * 128 + 32 * w + li;
    const bool p_ok = p < N;
    float xr[HC];
    {
        const float4* src = reinterpret_cast<const float4*>(X + (size_t)(p_ok ? p : 0) * C + h * HC);
#pragma unroll
        for (int k = 0; k < HC / 4; k++) {
            const float4 v = src[k];
            xr[4 * k] = v.x; xr[4 * k + 1] = v.y; xr[4 * k + 2] = v.z; xr[4 * k + 3] = v.w;
        }
    }
    auto load_tile = [&](int co0, float4 (&v)[LPT]) {
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int e = (threadIdx.x + 256 * k) * 4, r = e / C, c = e - r * C;
            v[k] = co0 + r < Cout ? *reinterpret_cast<const float4*>(Wd + (size_t)(co0 + r) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf, const float4 (&v)[LPT]) {
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int e = (threadIdx.x + 256 * k) * 4, r = e / C, c = e - r * C;
            *reinterpret_cast<float4*>(&L.Ws[buf][r][c]) = v[k];
        }
    };
    float4 wnext[LPT];
    load_tile(0, wnext);
    store_tile(0, wnext);
    __syncthreads();
    const int ntiles = (Cout + 31) / 32;
    for (int t = 0; t < ntiles; t++) {
        const int co0 = 32 * t, buf = t & 1;
        if (t + 1 < ntiles) load_tile(co0 + 32, wnext);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + mfma_row(r, h);
            acc[r] = co < Cout ? bias[co] : 0.f;
        }
        const float* wrow = &L.Ws[buf][li][h * HC];
#pragma unroll
        for (int s4 = 0; s4 < HC / 4; s4++) {
            const float4 a = *reinterpret_cast<const float4*>(wrow + 4 * s4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[4 * s4 + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[4 * s4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[4 * s4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[4 * s4 + 3], acc, 0, 0, 0);
            if ((s4 & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + mfma_row(r, h);
            if (p_ok && co < Cout) {
                if constexpr (HALF) reinterpret_cast<__half*>(out)[(size_t)co * N + p] = __float2half_rn(acc[r]);
                else reinterpret_cast<float*>(out)[(size_t)co * N + p] = acc[r];
            }
        }
        if (t + 1 < ntiles) store_tile(buf ^ 1, wnext);
        __syncthreads();
    }
}

// resize only (no decoder): pixel-major X -> (C, Hg, Wg), optionally fp16.  grid (ceil(N / 64), ceil(C / 32))
template <bool HALF>
__global__ void __launch_bounds__(256)
fl_resize_out_kernel(ResizeGeom g, int C, const float* __restrict__ fm, void* __restrict__ out) {
    const int N = g.Hg * g.Wg;
    const int p = blockIdx.x * 64 + (threadIdx.x & 63), cg = threadIdx.x >> 6;
    if (p >= N) return;
    const int yo = p / g.Wg, xo = p - yo * g.Wg;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    taps(yo, g.sy, g.H, y0, y1, ly0, ly1);
    taps(xo, g.sx, g.W, x0, x1, lx0, lx1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int c = blockIdx.y * 32 + cg * 8 + k;
        if (c >= C) continue;
        const float* pl = fm + (size_t)c * g.H * g.W;
        const float v = ly0 * (lx0 * pl[y0 * g.W + x0] + lx1 * pl[y0 * g.W + x1]) + ly1 * (lx0 * pl[y1 * g.W + x0] + lx1 * pl[y1 * g.W + x1]);
        if constexpr (HALF) reinterpret_cast<__half*>(out)[(size_t)c * N + p] = __float2half_rn(v);
        else reinterpret_cast<float*>(out)[(size_t)c * N + p] = v;
    }
}

// ---- K3: dW = g_y^T X, db = sum g_y over a range of pixel tiles; grid (ceil(Cout / 128), splits) ------------------
// A = g^T[co][px] comes straight from the co-major sign bytes: with the K index of step s defined as px = 32 h + s, a
// lane's 32 steps of a 64-pixel tile are 32 CONSECUTIVE bytes of its own row (two 16-byte loads, no LDS); B = X tile in
// LDS, shared by the four waves (four 32-row blocks of Cout), double-buffered with the next tile in flight in registers.
template <int C>
struct DwLds {
    static constexpr int XS = C + 4;
    float Xs[2][64][XS];
};

template <int C>
__global__ void __launch_bounds__(256)
fl_dweight_kernel(int N, int Np, int Cout, const float* __restrict__ X, const int8_t* __restrict__ S, float inv_n,
                  float* __restrict__ dW, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DwLds<C>& L = *reinterpret_cast<DwLds<C>*>(smem);
    constexpr int NCB = C / 32;
    constexpr int LPT = (64 * C / 4) / 256;       // float4 loads per thread for one X tile
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int co = blockIdx.x * 128 + 32 * w + li;
    const bool co_ok = co < Cout;
    const int tiles = (N + 63) / 64;
    const int per = (tiles + gridDim.y - 1) / gridDim.y;
    const int t_lo = blockIdx.y * per, t_hi = min(tiles, t_lo + per);
    if (t_lo >= t_hi) return;
    f32x16 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;
    float dbacc = 0.f;
    auto load_x = [&](int tI, float4 (&v)[LPT]) {
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int e = (threadIdx.x + 256 * k) * 4, px = e / C, c = e - px * C;
            const int p = tI * 64 + px;
            v[k] = p < N ? *reinterpret_cast<const float4*>(X + (size_t)p * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_x = [&](int buf, const float4 (&v)[LPT]) {
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int e = (threadIdx.x + 256 * k) * 4, px = e / C, c = e - px * C;
            *reinterpret_cast<float4*>(&L.Xs[buf][px][c]) = v[k];
        }
    };
    auto load_s = [&](int tI, uint4 (&v)[2]) {      // rows are padded to Np (a multiple of 64): aligned 16-byte loads
        const uint4* src = reinterpret_cast<const uint4*>(S + (size_t)(co_ok ? co : 0) * Np + (size_t)tI * 64 + 32 * h);
        v[0] = src[0]; v[1] = src[1];
    };
    float4 xn[LPT];
    uint4 sn[2];
    load_x(t_lo, xn);
    load_s(t_lo, sn);
    store_x(0, xn);
    __syncthreads();
    for (int tI = t_lo; tI < t_hi; tI++) {
        const int buf = (tI - t_lo) & 1;
        uint4 sc[2] = {sn[0], sn[1]};
        if (tI + 1 < t_hi) { load_x(tI + 1, xn); load_s(tI + 1, sn); }
        const uint32_t words[8] = {sc[0].x, sc[0].y, sc[0].z, sc[0].w, sc[1].x, sc[1].y, sc[1].z, sc[1].w};
        const int valid = co_ok ? N - (tI * 64 + 32 * h) : 0;     // steps s < valid are real pixels (the pad bytes are not)
#pragma unroll
        for (int s = 0; s < 32; s++) {
            float a = (float)(int)(int8_t)(words[s >> 2] >> (8 * (s & 3)));
            a = s < valid ? a : 0.f;
            dbacc += a;
            const float* xrow = &L.Xs[buf][32 * h + s][li];
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xrow[32 * cb], acc[cb], 0, 0, 0);
        }
        if (tI + 1 < t_hi) store_x(buf ^ 1, xn);
        __syncthreads();
    }
    // D: column l&31 = channel, rows = co of this wave's block
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int cor = blockIdx.x * 128 + 32 * w + mfma_row(r, h);
            if (cor < Cout && acc[cb][r] != 0.f) atomicAdd(&dW[(size_t)cor * C + 32 * cb + li], acc[cb][r] * inv_n);
        }
    dbacc += __shfl_xor(dbacc, 32, 64);
    if (h == 0 && co_ok && dbacc != 0.f) atomicAdd(&db[co], dbacc * inv_n);
}

// ---- K4: transpose of the resize: d_fm (C,H,W) gathered from g_x[N][C]; grid (ceil(W / 64), H, ceil(C / 32)) ----
// Which outputs touch source row y / source column x depends on the geometry only, not on the channel: the (output
// index, weight) lists are built once per workgroup (row list: every thread, it is uniform; column lists: one thread
// per column, in LDS) and the 32 channel lanes only multiply and add.  When shrinking (the training case: 1080p ->
// 360 x 480) two thirds of the source rows and columns have empty lists and their workgroups just store zeros - the
// kernel is then bound by the 4 C H W bytes it writes.  (The first version re-derived the taps per element and
// channel: 0.96 ms at C = 128 where the write takes 0.13 ms.)
constexpr int RB_MAXT = 6;     // list capacity: covers enlarging by up to 2.5x; larger factors take the general loop
__global__ void __launch_bounds__(256)
fl_resize_backward_kernel(ResizeGeom g, int C, const float* __restrict__ GX, float* __restrict__ dfm) {
    __shared__ float t[32][65];
    __shared__ int s_xo[64][RB_MAXT];
    __shared__ float s_wx[64][RB_MAXT];
    __shared__ int s_nx[64];
    const int x0 = blockIdx.x * 64, y = blockIdx.y, cb = blockIdx.z * 32;
    auto cand = [](int i, float scale, int out, int& lo, int& hi) { resize_cand(i, scale, out, lo, hi); };
    auto build = [](int i, float scale, int in, int out, int* oo, float* ww) { return resize_build<RB_MAXT>(i, scale, in, out, oo, ww); };
    int yo[RB_MAXT];
    float wy[RB_MAXT];
    const int ny = build(y, g.sy, g.H, g.Hg, yo, wy);           // uniform over the workgroup
    // 16-byte stores where a row segment allows them: this kernel is bound by the 4 C H W bytes it writes, and a 4-byte store
    // per lane (256 bytes per request) reaches 2.5 TB/s of them
    const bool vec4 = (g.W & 3) == 0 && (reinterpret_cast<uintptr_t>(dfm) & 15) == 0 && x0 + 64 <= g.W;
    if (ny == 0) {      // no output samples this source row (two rows of three when shrinking 3x): zeros, straight out
        if (vec4) {
            const int xq = threadIdx.x & 15, cg = threadIdx.x >> 4;       // 16 x 16 bytes per channel row, 16 channels per round
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int c = cb + cg + 16 * k;
                if (c < C) *reinterpret_cast<float4*>(dfm + (size_t)c * g.H * g.W + (size_t)y * g.W + x0 + 4 * xq) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
        const int xl = threadIdx.x & 63, cg = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = cb + cg * 8 + k;
            if (x0 + xl < g.W && c < C) dfm[(size_t)c * g.H * g.W + (size_t)y * g.W + x0 + xl] = 0.f;
        }
        return;
    }
    if (threadIdx.x < 64) {
        int xo[RB_MAXT];
        float wx[RB_MAXT];
        const int x = x0 + threadIdx.x;
        const int n = x < g.W ? build(x, g.sx, g.W, g.Wg, xo, wx) : 0;
        s_nx[threadIdx.x] = n;
#pragma unroll
        for (int k = 0; k < RB_MAXT; k++) { s_xo[threadIdx.x][k] = k < n ? xo[k] : 0; s_wx[threadIdx.x][k] = k < n ? wx[k] : 0.f; }
    }
    __syncthreads();
    {
        const int c = threadIdx.x & 31, xg = threadIdx.x >> 5;
        const bool c_ok = cb + c < C;
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            const int xl = xg * 8 + k, x = x0 + xl;
            const int nx = s_nx[xl];
            float acc = 0.f;
            if (x < g.W && c_ok && ny != 0 && nx != 0) {
                if (ny > 0 && nx > 0) {
                    for (int a = 0; a < ny; a++) {
                        const float* row = GX + (size_t)yo[a] * g.Wg * C + cb + c;
                        float r = 0.f;
                        for (int b = 0; b < nx; b++) r = fmaf(s_wx[xl][b], row[(size_t)s_xo[xl][b] * C], r);
                        acc = fmaf(wy[a], r, acc);
                    }
                } else {
                    // a list overflowed (enlarging by more than 2.5x): the general loop over the candidate ranges
                    int ylo, yhi, xlo, xhi;
                    cand(y, g.sy, g.Hg, ylo, yhi);
                    cand(x, g.sx, g.Wg, xlo, xhi);
                    for (int yy = ylo; yy <= yhi; yy++) {
                        int a0, a1;
                        float l0, l1;
                        taps(yy, g.sy, g.H, a0, a1, l0, l1);
                        const float wyv = (a0 == y ? l0 : 0.f) + (a1 == y ? l1 : 0.f);
                        if (wyv == 0.f) continue;
                        for (int xx = xlo; xx <= xhi; xx++) {
                            int b0, b1;
                            float m0, m1;
                            taps(xx, g.sx, g.W, b0, b1, m0, m1);
                            const float wxv = (b0 == x ? m0 : 0.f) + (b1 == x ? m1 : 0.f);
                            if (wxv != 0.f) acc += (wyv * wxv) * GX[(size_t)(yy * g.Wg + xx) * C + cb + c];
                        }
                    }
                }
            }
            t[c][xl] = acc;
        }
    }
    __syncthreads();
    if (vec4) {
        const int xq = threadIdx.x & 15, cg = threadIdx.x >> 4;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int cl = cg + 16 * k;
            if (cb + cl < C)
                *reinterpret_cast<float4*>(dfm + (size_t)(cb + cl) * g.H * g.W + (size_t)y * g.W + x0 + 4 * xq) =
                    make_float4(t[cl][4 * xq], t[cl][4 * xq + 1], t[cl][4 * xq + 2], t[cl][4 * xq + 3]);
        }
    } else {
        const int xl = threadIdx.x & 63, cg = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = cb + cg * 8 + k;
            if (x0 + xl < g.W && c < C) dfm[(size_t)c * g.H * g.W + (size_t)y * g.W + x0 + xl] = t[cg * 8 + k][xl];
        }
    }
}

__global__ void __launch_bounds__(256) fl_sum_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)partial[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

struct Scratch {
    float* X;
    float* GX;
    int8_t* S;
    float* loss_partial;
    size_t n_partial;
    static Scratch carve(char* base, int C, int Cout, int N, bool decoder, size_t* bytes) {
        Carver c(base);
        Scratch s;
        s.X = c.take<float>((size_t)N * C);
        s.GX = decoder ? c.take<float>((size_t)N * C) : s.X;     // without a decoder K1 writes g_x straight away
        s.S = c.take<int8_t>(decoder ? (size_t)((N + 63) / 64 * 64) * Cout : 0);      // [Cout][N padded to 64] sign bytes
        s.n_partial = decoder ? (size_t)(N + 127) / 128 : (size_t)((N + 63) / 64) * ((C + 31) / 32);
        s.loss_partial = c.take<float>(s.n_partial);
        if (bytes) *bytes = c.total();
        return s;
    }
};

template <int C>
hipError_t run_decoder(int N, int Cout, const Scratch& sc, const float* Wd, const float* bias, const float* gt, float inv_n,
                       float* dW, float* db, hipStream_t s) {
    const size_t lds2 = sizeof(GemmLds<C>), lds3 = sizeof(DwLds<C>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fl_decoder_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fl_dweight_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds3);
    if (e != hipSuccess) return e;
    const int Np = (N + 63) / 64 * 64;
    hipLaunchKernelGGL((fl_decoder_kernel<C>), dim3((N + 127) / 128), dim3(256), lds2, s, N, Np, Cout, sc.X, Wd, bias, gt, inv_n,
                       sc.GX, sc.S, sc.loss_partial);
    e = hipMemsetAsync(dW, 0, (size_t)Cout * C * sizeof(float), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), s);
    if (e != hipSuccess) return e;
    const int cblocks = (Cout + 127) / 128, tiles = (N + 63) / 64;
    const int splits = max(1, min(tiles, 512 / cblocks));
    hipLaunchKernelGGL((fl_dweight_kernel<C>), dim3(cblocks, splits), dim3(256), lds3, s, N, Np, Cout, sc.X, sc.S, inv_n, dW, db);
    return hipGetLastError();
}

}  // namespace

size_t feature_l1_scratch_bytes(int C, int Cout, int Hg, int Wg, bool decoder) {
    size_t b = 0;
    Scratch::carve(nullptr, C, Cout, Hg * Wg, decoder, &b);
    return b;
}

bool feature_l1_decoder_supported(int C) { return C == 32 || C == 64 || C == 128; }

const float* feature_l1_lowres_grad(char* scratch, int C, int Cout, int Hg, int Wg, bool decoder) {
    return Scratch::carve(scratch, C, Cout, Hg * Wg, decoder, nullptr).GX;
}

size_t feature_decode_scratch_bytes(int C, int Hg, int Wg, bool decoder) {
    return decoder ? (((size_t)Hg * Wg * C * sizeof(float) + ALIGN - 1) & ~(ALIGN - 1)) : 0;
}

template <int C>
static hipError_t run_decode(int N, int Cout, const float* X, const float* Wd, const float* bias, void* out, bool half, hipStream_t s) {
    const size_t lds = sizeof(GemmLds<C>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fl_decode_kernel<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fl_decode_kernel<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (half) hipLaunchKernelGGL((fl_decode_kernel<C, true>), dim3((N + 127) / 128), dim3(256), lds, s, N, Cout, X, Wd, bias, out);
    else hipLaunchKernelGGL((fl_decode_kernel<C, false>), dim3((N + 127) / 128), dim3(256), lds, s, N, Cout, X, Wd, bias, out);
    return hipGetLastError();
}

hipError_t launch_feature_decode(int C, int H, int W, int Cout, int Hg, int Wg, const float* feature_map, const float* weight,
                                 const float* bias, void* out, bool half, char* scratch, hipStream_t s) {
    const int N = Hg * Wg;
    const ResizeGeom g = make_resize_geom(H, W, Hg, Wg);
    const dim3 grid1((N + 63) / 64, (C + 31) / 32);
    if (!weight) {
        if (half) hipLaunchKernelGGL(fl_resize_out_kernel<true>, grid1, dim3(256), 0, s, g, C, feature_map, out);
        else hipLaunchKernelGGL(fl_resize_out_kernel<false>, grid1, dim3(256), 0, s, g, C, feature_map, out);
        return hipGetLastError();
    }
    float* X = reinterpret_cast<float*>(scratch);
    hipLaunchKernelGGL(fl_resize_kernel, grid1, dim3(256), 0, s, g, C, feature_map, X, nullptr, 0.f, nullptr);
    if (C == 32) return run_decode<32>(N, Cout, X, weight, bias, out, half, s);
    if (C == 64) return run_decode<64>(N, Cout, X, weight, bias, out, half, s);
    if (C == 128) return run_decode<128>(N, Cout, X, weight, bias, out, half, s);
    return hipErrorInvalidValue;
}

hipError_t launch_feature_l1(int C, int H, int W, int Cout, int Hg, int Wg, const float* feature_map, const float* weight,
                             const float* bias, const float* gt, float* loss, float* d_feature_map, float* d_weight,
                             float* d_bias, char* scratch, hipStream_t s) {
    const bool decoder = weight != nullptr;
    const int N = Hg * Wg;
    const Scratch sc = Scratch::carve(scratch, C, Cout, N, decoder, nullptr);
    const ResizeGeom g = make_resize_geom(H, W, Hg, Wg);
    const float inv_n = 1.0f / ((float)N * (float)Cout);
    const dim3 grid1((N + 63) / 64, (C + 31) / 32);
    hipLaunchKernelGGL(fl_resize_kernel, grid1, dim3(256), 0, s, g, C, feature_map, sc.X, decoder ? nullptr : gt, inv_n,
                       sc.loss_partial);
    if (decoder) {
        hipError_t e = hipErrorInvalidValue;
        if (C == 32) e = run_decoder<32>(N, Cout, sc, weight, bias, gt, inv_n, d_weight, d_bias, s);
        else if (C == 64) e = run_decoder<64>(N, Cout, sc, weight, bias, gt, inv_n, d_weight, d_bias, s);
        else if (C == 128) e = run_decoder<128>(N, Cout, sc, weight, bias, gt, inv_n, d_weight, d_bias, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fl_sum_kernel, dim3(1), dim3(256), 0, s, sc.loss_partial, (int)sc.n_partial, loss);
    // d_feature_map == nullptr: the caller hands the gradient at the loss's resolution (feature_l1_lowres_grad) to the blend
    // backward, which applies the transposed resize while it stages its tiles (f3dgs_set_feature_grad_lowres)
    if (d_feature_map)
        hipLaunchKernelGGL(fl_resize_backward_kernel, dim3((W + 63) / 64, H, (C + 31) / 32), dim3(256), 0, s, g, C, sc.GX,
                           d_feature_map);
    return hipGetLastError();
}

}  // namespace f3dgs
