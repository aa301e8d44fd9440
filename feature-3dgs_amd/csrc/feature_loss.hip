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

#include "common.h"

namespace f3dgs {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ResizeGeom {
    int H, W, Hg, Wg;
    float sy, sx;      // (in - 1) / (out - 1), 0 when out == 1 (PyTorch: area_pixel_compute_scale, align_corners)
};

// source taps of output index o (PyTorch upsample_bilinear2d, align_corners = true)
__device__ __forceinline__ void taps(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    const float src = scale * (float)o;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

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

// ---- K2: decoder forward + residual + g_x, one workgroup = 64 pixels x all Cout ---------------------------------------
template <int C>
struct GemmLds {
    float Xs[64][C + 1];       // pixel tile, row-major, odd stride: the A-operand reads (lane = row) are conflict-free
    float Ws[128][C + 1];      // decoder rows co0 .. co0+127 of the current iteration
    float Gs[64][129];         // g_y of the current iteration
    float red[4];
};

template <int C>
__global__ void __launch_bounds__(256)
fl_decoder_kernel(int N, int Cout, const float* __restrict__ X, const float* __restrict__ Wd, const float* __restrict__ bias,
                  const float* __restrict__ gt, float inv_n, float* __restrict__ GX, int8_t* __restrict__ S,
                  float* __restrict__ loss_partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GemmLds<C>& L = *reinterpret_cast<GemmLds<C>*>(smem);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int p0 = blockIdx.x * 64;
    constexpr int NCB = C / 32;                   // channel blocks of the g_x accumulation: wave w owns block w
    for (int e = threadIdx.x; e < 64 * C; e += 256) {
        const int px = e / C, c = e - px * C;
        L.Xs[px][c] = p0 + px < N ? X[(size_t)(p0 + px) * C + c] : 0.f;
    }
    f32x16 gx[2];
#pragma unroll
    for (int r = 0; r < 16; r++) { gx[0][r] = 0.f; gx[1][r] = 0.f; }
    float loss = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    for (int co0 = 0; co0 < Cout; co0 += 128) {
        __syncthreads();                          // previous iteration's readers of Ws / Gs are done
        for (int e = threadIdx.x; e < 128 * C; e += 256) {
            const int r = e / C, c = e - r * C;
            L.Ws[r][c] = co0 + r < Cout ? Wd[(size_t)(co0 + r) * C + c] : 0.f;
        }
        __syncthreads();
        // ---- phase A: y[64 px][32 co of this wave] = Xs Ws^T
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll 4
        for (int ks = 0; ks < C / 2; ks++) {
            const float b = L.Ws[32 * w + li][2 * ks + lk];              // B[k][n] = W[co = n][c = k]
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.Xs[li][2 * ks + lk], b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.Xs[32 + li][2 * ks + lk], b, acc[1], 0, 0, 0);
        }
        // ---- residual, loss, g_y (this lane: column co, rows = pixels)
        const int co = co0 + 32 * w + li;
        const bool co_ok = co < Cout;
        const float bv = co_ok ? bias[co] : 0.f;
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int px = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int p = p0 + px;
                float gv = 0.f;
                if (co_ok && p < N) {
                    const float res = acc[rb][r] + bv - gt[(size_t)co * N + p];
                    loss += fabsf(res);
                    gv = res > 0.f ? inv_n : (res < 0.f ? -inv_n : 0.f);
                    S[(size_t)p * Cout + co] = res > 0.f ? 1 : (res < 0.f ? -1 : 0);
                }
                L.Gs[px][32 * w + li] = gv;
            }
        __syncthreads();
        // ---- phase B: g_x[64 px][32 c of this wave] += Gs[64][128] Ws[128][c block]
        if (w < NCB) {
#pragma unroll 4
            for (int ks = 0; ks < 64; ks++) {
                const float b = L.Ws[2 * ks + lk][32 * w + li];          // B[k][n] = W[co = k][c = n]
                gx[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.Gs[li][2 * ks + lk], b, gx[0], 0, 0, 0);
                gx[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.Gs[32 + li][2 * ks + lk], b, gx[1], 0, 0, 0);
            }
        }
    }
    if (w < NCB) {
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int px = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (p0 + px < N) GX[(size_t)(p0 + px) * C + 32 * w + li] = gx[rb][r];
            }
    }
    loss = wave_sum(loss);
    if (lane == 0) L.red[w] = loss;
    __syncthreads();
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = (L.red[0] + L.red[1] + L.red[2] + L.red[3]) * inv_n;
}

// ---- K3: dW = g_y^T X, db = sum g_y over a range of pixel tiles; grid (ceil(Cout / 128), splits) ------------------
template <int C>
struct DwLds {
    float Xs[64][C + 1];
    float Ss[64][129];
};

template <int C>
__global__ void __launch_bounds__(256)
fl_dweight_kernel(int N, int Cout, const float* __restrict__ X, const int8_t* __restrict__ S, float inv_n,
                  float* __restrict__ dW, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DwLds<C>& L = *reinterpret_cast<DwLds<C>*>(smem);
    constexpr int NCB = C / 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int co0 = blockIdx.x * 128;
    const int tiles = (N + 63) / 64;
    const int per = (tiles + gridDim.y - 1) / gridDim.y;
    const int t_lo = blockIdx.y * per, t_hi = min(tiles, t_lo + per);
    f32x16 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;
    float dbacc = 0.f;
    for (int tI = t_lo; tI < t_hi; tI++) {
        const int p0 = tI * 64;
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * C; e += 256) {
            const int px = e / C, c = e - px * C;
            L.Xs[px][c] = p0 + px < N ? X[(size_t)(p0 + px) * C + c] : 0.f;
        }
        for (int e = threadIdx.x; e < 64 * 128; e += 256) {
            const int px = e >> 7, r = e & 127;
            L.Ss[px][r] = (p0 + px < N && co0 + r < Cout) ? (float)S[(size_t)(p0 + px) * Cout + co0 + r] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < 32; ks++) {
            const float a = L.Ss[2 * ks + lk][32 * w + li];              // A[i = co][k = pixel]
            dbacc += a;
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, L.Xs[2 * ks + lk][32 * cb + li], acc[cb], 0, 0, 0);
        }
    }
    // D: column l&31 = channel, rows = co
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (co < Cout && acc[cb][r] != 0.f) atomicAdd(&dW[(size_t)co * C + 32 * cb + li], acc[cb][r] * inv_n);
        }
    dbacc += __shfl_xor(dbacc, 32, 64);
    if (lk == 0 && co0 + 32 * w + li < Cout && dbacc != 0.f) atomicAdd(&db[co0 + 32 * w + li], dbacc * inv_n);
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
    // candidate outputs of a source index i: outputs o whose taps can include i
    auto cand = [](int i, float scale, int out, int& lo, int& hi) {
        if (scale <= 0.f) { lo = 0; hi = (i == 0) ? 0 : -1; return; }
        lo = max(0, (int)floorf((float)(i - 1) / scale) - 1);
        hi = min(out - 1, (int)ceilf((float)(i + 1) / scale) + 1);
    };
    // (output, weight) list of source index i along one axis; returns the count, or -1 when it exceeds the capacity
    auto build = [&](int i, float scale, int in, int out, int* oo, float* ww) {
        int lo, hi, n = 0;
        cand(i, scale, out, lo, hi);
        for (int o = lo; o <= hi; o++) {
            int a0, a1;
            float l0, l1;
            taps(o, scale, in, a0, a1, l0, l1);
            const float wv = (a0 == i ? l0 : 0.f) + (a1 == i ? l1 : 0.f);
            if (wv == 0.f) continue;
            if (n == RB_MAXT) return -1;
            oo[n] = o; ww[n] = wv; n++;
        }
        return n;
    };
    int yo[RB_MAXT];
    float wy[RB_MAXT];
    const int ny = build(y, g.sy, g.H, g.Hg, yo, wy);           // uniform over the workgroup
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
    {
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
        s.S = c.take<int8_t>(decoder ? (size_t)N * Cout : 0);
        s.n_partial = decoder ? (size_t)(N + 63) / 64 : (size_t)((N + 63) / 64) * ((C + 31) / 32);
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
    hipLaunchKernelGGL((fl_decoder_kernel<C>), dim3((N + 63) / 64), dim3(256), lds2, s, N, Cout, sc.X, Wd, bias, gt, inv_n, sc.GX,
                       sc.S, sc.loss_partial);
    e = hipMemsetAsync(dW, 0, (size_t)Cout * C * sizeof(float), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), s);
    if (e != hipSuccess) return e;
    const int cblocks = (Cout + 127) / 128, tiles = (N + 63) / 64;
    const int splits = max(1, min(tiles, 512 / cblocks));
    hipLaunchKernelGGL((fl_dweight_kernel<C>), dim3(cblocks, splits), dim3(256), lds3, s, N, Cout, sc.X, sc.S, inv_n, dW, db);
    return hipGetLastError();
}

}  // namespace

size_t feature_l1_scratch_bytes(int C, int Cout, int Hg, int Wg, bool decoder) {
    size_t b = 0;
    Scratch::carve(nullptr, C, Cout, Hg * Wg, decoder, &b);
    return b;
}

bool feature_l1_decoder_supported(int C) { return C == 32 || C == 64 || C == 128; }

hipError_t launch_feature_l1(int C, int H, int W, int Cout, int Hg, int Wg, const float* feature_map, const float* weight,
                             const float* bias, const float* gt, float* loss, float* d_feature_map, float* d_weight,
                             float* d_bias, char* scratch, hipStream_t s) {
    const bool decoder = weight != nullptr;
    const int N = Hg * Wg;
    const Scratch sc = Scratch::carve(scratch, C, Cout, N, decoder, nullptr);
    ResizeGeom g;
    g.H = H; g.W = W; g.Hg = Hg; g.Wg = Wg;
    g.sy = Hg > 1 ? (float)(H - 1) / (float)(Hg - 1) : 0.f;
    g.sx = Wg > 1 ? (float)(W - 1) / (float)(Wg - 1) : 0.f;
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
    hipLaunchKernelGGL(fl_resize_backward_kernel, dim3((W + 63) / 64, H, (C + 31) / 32), dim3(256), 0, s, g, C, sc.GX,
                       d_feature_map);
    return hipGetLastError();
}

}  // namespace f3dgs
