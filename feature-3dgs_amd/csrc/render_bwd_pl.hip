// render_bwd_pl.hip — blend backward, PIXEL-LANE formulation (the default for feature_mfma = 1).
//
// Semantics: R/cuda_rasterizer/backward.cu:407-620 (R = submodules/diff-gaussian-rasterization-feature), quirks Q1, Q2,
// Q3, Q5, Q8 as in render_bwd.hip.
//
// The instance-lane kernel (render_bwd.hip) pays for the list-order dependence with two DPP prefix scans per pixel and
// chunk, runs every per-(pixel, Gaussian) multiply-add on the vector pipe and flushes one partial sum per (8x8 quadrant,
// Gaussian) with global atomics.  Here the work is cut differently: one 16x16 tile per WORKGROUP of four waves; the
// tile's list is walked back to front in windows of 64 entries (splat records gathered ONCE per tile into LDS, a window
// ahead) and processed in chunks of 16 consecutive entries, every chunk in two phases:
//
//   phase 1 (vector pipe, lane = PIXEL, wave = 8x8 QUADRANT): the wave takes its hits of the chunk (rect_hit against its
//     quadrant, lane = entry, once per window) one after the other (NE at a time for instruction-level parallelism);
//     transmittance T and the "colour behind" sum S are per-lane running scalars, so the recurrences are two
//     multiply-adds - no scans.  Per entry and pixel it produces two numbers,
//         w = alpha T_front            (every dL/d{colour, depth, feature} sum is  sum_px w  x  dL/dpixel)
//         s = op G dL/dalpha           (every geometric sum is a moment  sum_px s x {1, u, v, u^2, uv, v^2}),
//     stored in the quadrant's two 16 x 64 LDS tiles (pixel-permuted columns, see PlShared); rows of entries that miss the
//     quadrant are zeroed.
//   phase 2 (matrix pipe, wave = 16 output COLUMNS): every per-Gaussian sum of the chunk is ONE contraction over the 256
//     pixels of the tile,  [16 entries x 256 px] x [256 px x (C + 4 + 6)]  ->  exact-fp32 v_mfma_f32_16x16x4_f32.  The
//     four waves split the OUTPUT COLUMNS, not the pixels: wave 0 / 1 take feature channels 0-15 / 16-31, wave 2 the
//     four colour / depth columns, wave 3 the six moment columns (later channel windows of a wide feature: sixteen
//     channels per wave).  Each wave reads the A tiles of all four quadrants and keeps its B operand - its sixteen
//     columns at all 256 pixels, 64 registers - resident for its whole life (the gradient images are staged once per tile
//     through LDS with full-row requests).  A wave's accumulators therefore hold TILE-level sums: no partial sums are
//     merged anywhere, and the tile leaves ONE coalesced atomic flush per (tile, entry) - 2.7x fewer global atomic
//     requests than the per-quadrant flushes of the instance-lane kernel at config c3 - after the geometric sums are
//     rebuilt from the moments about the tile centre (dx = (mean - centre) - u).
//
// Work per (entry, wave): ~35 vector instructions where the entry reaches the quadrant (instance-lane kernel: 63 per 64
// pairs) and 4 matrix instructions (128 matrix-pipe cycles); the two pipes overlap across the workgroups of a CU.

#include <cstdio>
#include <type_traits>

#include "render_common.h"
#include "pl_phase1.h"
#include "resize_taps.h"

namespace f3dgs {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- bf16 split operands (template parameter BF; option bwd_bf16) -------------------------------------------------------------
// An fp32 matrix instruction occupies its SIMD for its whole duration (it excludes the vector instructions of the other
// waves, tools/pipe_probe.hip) and runs at 1/16 of the bf16 rate.  With BF every contraction of phase 2 runs on
// v_mfma_f32_16x16x32_bf16 instead, fp32 accumulation, each fp32 operand split into TWO bf16 terms by round-to-nearest:
//     x = hi + mid + e,  hi = bf16(x),  mid = bf16(x - hi)  (x - hi is exact),  |e| <= 2^-18 |x|
// and a product w g evaluated as  w_hi g_hi + w_hi g_mid + w_mid g_hi  (the dropped w_mid g_mid is <= 2^-18 |w g| too): every
// term of a gradient sum carries a relative error of a few 1e-6 with random sign, against a bar of 1e-3 - the gradients'
// tolerance is what allows the two-term split (the forward's 1e-4 ABSOLUTE bar on O(1) features would not).  The monomials of
// the moment block (1, u, v, u^2, uv, v^2 with |u|, |v| <= 3.5 in steps of 1) have at most six significant bits: exact in bf16,
// one term.  Per chunk a wave issues 24 (moment wave: 16) matrix instructions of 16 cycles instead of 64 of 32 cycles.
// One v_cvt_pk_bf16_f32 converts two values (round to nearest even); the halves are stored with ds_write_b16 /
// ds_write_b16_d16_hi as they stand.
// (lo, hi) -> packed high terms `h` and packed middle terms `m`
__device__ __forceinline__ void split_bf16(float lo, float hi, uint32_t& h, uint32_t& m) {
    h = pack_bf16(lo, hi);
    m = pack_bf16(lo - __uint_as_float(h << 16), hi - __uint_as_float(h & 0xFFFF0000u));
}
__device__ __forceinline__ f32x4 mfma_bf16(const uint32_t (&a)[4], const uint32_t (&b)[4], f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, u32x4{a[0], a[1], a[2], a[3]}),
                                                   __builtin_bit_cast(bf16x8, u32x4{b[0], b[1], b[2], b[3]}), c, 0, 0, 0);
}
// BF tile layout (aliases PlShared::wt and ::st, 32 KB; constants in pl_phase1.h): quadrant q at byte 8192 q; term t (0: w high,
// 1: w middle, 2: s high, 3: s middle) at + 2048 t; row (entry) i at + 128 i; the row's eight 16-byte units = the eight pixel rows
// of the quadrant, unit y stored at slot y ^ (i >> 1); pixel x of the row at + 2 x.  Operand lane (i, kg) of K span ks reads unit
// 4 ks + kg of row i - sixteen bytes = the K slots 8 kg .. 8 kg + 7 - and the sixteen lanes of a ds_read_b128 service group meet
// sixteen different 16-byte slots of the 256-byte bank row (two rows per bank row x eight slots).

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global access of the
// wave - here the fire-and-forget atomics of the flush, microseconds under load - which nothing in the workgroup reads.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The lane index, recomputed where it is asked for: values derived from it inside a loop are rebuilt per iteration (two
// instructions) instead of living in registers across the whole walk - the B operand needs the room.
__device__ __forceinline__ int fresh_lane() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

#ifdef F3DGS_DEV
#define PL_DEV_SKIP(bit) (a.dev & (bit))     // 1: no global atomics  2: no phase 1  4: no matrix instructions  16: no flush
#define PL_PHASE_BEGIN() unsigned long long ph_t0_ = (a.dev & 8) ? __builtin_readcyclecounter() : 0ull; unsigned long long cyc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PL_PHASE_END(K) do { if (a.dev & 8) { const unsigned long long t1_ = __builtin_readcyclecounter(); cyc_[K] += t1_ - ph_t0_; ph_t0_ = t1_; } } while (0)
#define PL_COUNT(K, N) do { if (a.dev & 8) cyc_[K] += (N); } while (0)
#define PL_STAGE_MARK(K) do { if (a.dev & 32) PL_PHASE_END(K); } while (0)     // staging only (bit 32): finer phases
#else
#define PL_DEV_SKIP(bit) false
#define PL_PHASE_BEGIN() do {} while (0)
#define PL_PHASE_END(K) do {} while (0)
#define PL_COUNT(K, N) do {} while (0)
#define PL_STAGE_MARK(K) do {} while (0)
#endif

#ifndef PL_SCHED
#define PL_SCHED 1                 // who orders phase 1 of the bf16 shape, see pl_phase1.h (0: the compiler, as in round 5)
#endif
constexpr int PL_CAP = 16;         // list entries per chunk = rows of one 16x16x4 matrix instruction
constexpr int PL_ROW = 64;         // dwords per pixel-block row of an A tile (16 entries x 4 px, XOR-swizzled)
constexpr int PL_TILE = 16 * PL_ROW;
constexpr int PL_WIN = 64;         // list entries per window
constexpr int PL_FS = 68;          // dwords per entry of the flush tile: 32 (64) feature sums; 4 colour / depth sums + 4 x 6 moments; 8 of splat data
constexpr int PL_SP = 260;         // dwords per plane of the staging image (16 x 16 pixels + 4: plane offset of 4 banks)
constexpr int PL_STAGE_PLANES = 38;

// (PlRec - one staged list entry, shared by the four waves, three broadcast reads per entry in phase 1 - lives in pl_phase1.h)

// A tiles: element (row i, column c) of a quadrant's tile, with c = 16 u + 4 k + m - the pixel that matrix step
// t = 4 u + m contracts at K index k - lives in pixel block b = c >> 2 at dword 64 b + 4 (i ^ (b & 7)) + (c & 3): the 16
// rows of a pixel block are 16 bytes apart and rotated by an XOR with the block index, so that the operand read of lane
// (i, k) - one 16-byte read per u - meets the other lanes of its 16-lane service group ({0-3, 12-15, 20-27}, ...) on sixteen
// different 4-bank groups, and the phase-1 stores (the 64 pixels of one row: sixteen blocks) are 2-way conflicted at most.
// The XOR touches address bits 4..6 only: a row's address is (lane base) ^ (16 row), one instruction per entry.
constexpr int PL_FS8 = 136;        // eight-wave later windows: 128 feature sums + 8 of splat data
template <int FS>
struct PlSharedT {
    float wt[4][PL_TILE];       // per quadrant: blend weights
    float st[4][PL_TILE];       //               s = op G dL/dalpha
    PlRec rec[PL_WIN];          // one window of records: read by phase 1 only (the flush has its own copy in the flush tile)
    float ftile[PL_CAP * FS];   // the chunk's sums, entry-major, as they leave for global memory; the last 8 slots: splat data
    uint32_t wave_max[4];
    uint32_t touched[2];        // per chunk parity: rows that blended somewhere in the tile
    uint32_t pad[2];
};
typedef PlSharedT<PL_FS> PlShared;
typedef PlSharedT<PL_FS8> PlShared8;
static_assert(sizeof(PlShared) * 4 <= 160 * 1024, "four workgroups per CU");
static_assert(sizeof(PlShared8) * 2 <= 160 * 1024, "two eight-wave workgroups per CU");
static_assert(PL_STAGE_PLANES * PL_SP * 4 <= sizeof(PlShared), "staging image fits the aliased area");
constexpr int PL_TAPS_OFS = PL_STAGE_PLANES * PL_SP * 4;      // bytes: tap lists of the tile's rows and columns (low-resolution gradient)
static_assert(PL_TAPS_OFS + 32 * sizeof(float4) + 34 * sizeof(int) <= sizeof(PlShared), "tap lists fit behind the staging image");

// GEO = true:  first channel window (up to 32 channels) + the ten geometric sums; column blocks of the waves: feature
//              channels 0-15, 16-31, colour/depth, moments.
// GEO = false: a later channel window of up to 64 channels: sixteen per wave.
// P1_NE / P1_PREF: entries per phase-1 group, records of the next group prefetched.
// M44: the colour wave of the first window contracts on sixteen 4 x 4 blocks (v_mfma_f32_4x4x1_16B_f32).
// NWV = 8 (round 6; bf16 shape, later windows only): EIGHT waves per tile - a window of up to 128 channels, sixteen per wave.
//   Every later window re-evaluates the blend weights of the whole list (phase 1); with twice the columns per window half as many
//   windows do.  Waves 0..3 are the quadrant waves of phase 1 (and take channels 0..63 in phase 2), waves 4..7 only contract
//   (channels 64..127) and flush; two workgroups per CU (the same sixteen waves per CU as four workgroups of four).
// S32 (round 6; bf16 shape, first window): the HYBRID shape - feature and colour / depth blocks on bf16 matrix instructions as in the
//   bf16 shape, the MOMENT block (the sums the covariance chain amplifies) on exact-fp32 matrix instructions from an fp32 tile
//   of s: what a frame with needle-shaped Gaussians takes under option bwd_bf16 = -1.  Only the moment wave's SIMD is held by
//   fp32 matrix instructions (2048 cycles per chunk); in the exact shape all four are.
template <bool GEO, int P1_NE, bool P1_PREF, bool M44 = false, bool BF = false, int NWV = 4, bool S32 = false>
__device__ __forceinline__ void render_backward_pl_body(const BwdArgs& a) {
    static_assert(!BF || (P1_NE == 1 && !M44), "the bf16 shape takes its entries one at a time and has no 4 x 4 colour path");
    static_assert(!S32 || (BF && GEO && NWV == 4), "the hybrid shape is a first window of the bf16 shape");
    static_assert(NWV == 4 || (NWV == 8 && BF && !GEO), "eight waves: later windows of the bf16 shape only");
    constexpr int NTH = 64 * NWV;                    // threads per workgroup
    constexpr int FS = NWV == 8 ? PL_FS8 : PL_FS;    // dwords per row of the flush tile
    constexpr int FROWS = PL_CAP / NWV;              // rows of a chunk each wave flushes
    constexpr int GID_SLOT = GEO ? 66 : FS - 4;      // where a row's Gaussian index waits for the flush (GEO: behind its splat data)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PlSharedT<FS>& L = *reinterpret_cast<PlSharedT<FS>*>(smem);
    float* const stage = reinterpret_cast<float*>(smem);      // staging image: aliases everything, used before the walk only
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave: quadrant in phase 1 (waves 0..3), column block in phase 2
    const bool split16 = !BF && GEO && a.split16 && a.nc <= 16;      // wave-uniform: see the operand read-out below
    // later windows of up to 32 channels (the last window of 64 + 32 or 3 x 64 + 32 remaining channels: two waves would idle): wave q
    // takes channel block q & 1 at the quadrants 2 (q >> 1), 2 (q >> 1) + 1 - two partial sums per entry, added in the flush
    const bool split32 = !BF && !GEO && a.split16 && a.nc <= 32;
    PL_PHASE_BEGIN();     // [0] staging  [1] window + barriers  [2] phase 1  [3] phase 2  [4] flush  [5] chunks  [6] entries

    // longest walks first: the launch is ~8 rounds of workgroups whose lifetimes differ by an order of magnitude, and a long
    // one that starts late has the CU to itself at the end (c3: 0.708 -> 0.667 ms in a development build)
    const uint32_t tile = a.order ? a.order[blockIdx.x] : band_perm(xcd_remap(blockIdx.x, gridDim.x), gridDim.x, a.band_b0, a.band_tb);
    const int tx = tile % a.gx, ty = tile / a.gx;
    const uint2 rg = a.ranges[tile];
    if (rg.x >= rg.y) return;     // (workgroup-uniform) an empty list - every tile outside a listed band: nothing to stage, nothing to add
    const uint32_t r_lo = __builtin_amdgcn_readfirstlane((int)rg.x);
    const size_t HW = (size_t)a.W * a.H;
    const int tx0 = tx * TILE, ty0 = ty * TILE;
    const int qx = (q & 1) * 8, qy = ((q >> 1) & 1) * 8;
    // (lane-derived indices are rebuilt where they are used - staging read-out, walk set-up - instead of living across the staging)

    // ---- staging: every plane of the tile with full-row requests -> LDS [plane][16 rows][16 px]; then the per-pixel state
    // (lane = pixel of quadrant q) and the wave's B operand (lane = (column col, K index kk)) are read out of it.
    // B operand, resident for the life of the wave: Bop[qd][t] = column `col` of this wave's block at the pixel that step t
    // of quadrant qd contracts at K index kk, i.e. pixel (4 (t & 1) + kk, t >> 1) of the quadrant.
    // (BF: the same 64 registers hold the operand as bf16 pairs - Bbf[qd][ks][0..3] the high terms, [4..7] the middle terms
    // of this wave's column at the eight pixels (x = 0..7, y = 4 ks + kg) of quadrant qd, i.e. K slots 8 kg .. 8 kg + 7 of span ks)
    float Bop[BF ? 1 : 4][16];
    uint32_t Bbf[BF ? 4 : 1][2][8];
    uint32_t last = 0;
    float T = 0.f, dR = 0.f, dG = 0.f, dB = 0.f, dD = 0.f;
    {
        const bool vec = (a.W & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.dL_dfeat) | reinterpret_cast<uintptr_t>(a.dL_dpix) |
                                              reinterpret_cast<uintptr_t>(a.dL_ddepth) | reinterpret_cast<uintptr_t>(a.final_T) |
                                              reinterpret_cast<uintptr_t>(a.n_contrib)) & 15) == 0;
        const int rounds = GEO ? 1 : (a.nc + 31) / 32;
        for (int rd = 0; rd < rounds; rd++) {
            // planes of this round: 32 feature planes (channels 32 rd ..), then - first round only - the pixel planes:
            // GEO: dL/dR, dL/dG, dL/dB, dL/ddepth, final_T, n_contrib;  otherwise: final_T, n_contrib
            const int npix = rd == 0 ? (GEO ? 6 : 2) : 0;
            if (rd > 0) __syncthreads();
            // the thread index, rebuilt per round: what the requests below derive from it (plane, row, column of ten elements) is
            // then recomputed in the second round instead of living in registers across the first round's read-out
            const int tid = fresh_lane() + 64 * q;
            // all requests of a thread first (branch-free: a lane that has nothing to fetch reads a valid address and drops the
            // value), then the LDS stores: one memory latency per round instead of one per plane
            constexpr int NIT = (38 * 64 + NTH - 1) / NTH;
            float4 sv[NIT];
            if (vec) {
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int f = it * NTH + tid;
                    const int pl = f >> 6, row = (f >> 2) & 15, xq = f & 3;
                    const int y = ty0 + row, x = tx0 + 4 * xq;
                    const int ch = 32 * rd + pl;
                    const int pp = pl - 32 + (GEO ? 0 : 4);
                    const bool on = y < a.H && x < a.W && f < (32 + npix) * 64 && (pl >= 32 || (ch < a.nc && a.dL_dfeat));
                    const float* src = pl < 32 ? a.dL_dfeat + (size_t)(a.c0 + ch) * HW
                                     : pp < 3 ? a.dL_dpix + (size_t)pp * HW : pp == 3 ? a.dL_ddepth : pp == 4 ? a.final_T : reinterpret_cast<const float*>(a.n_contrib);
                    const float* p = on ? src + (size_t)y * a.W + x : a.final_T;
                    const float4 raw = *reinterpret_cast<const float4*>(p);
                    sv[it] = on ? raw : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else {
                for (int it = 0; it < NIT; it++) {          // images whose rows are not 16-byte aligned: element by element
                    const int f = it * NTH + tid;
                    const int pl = f >> 6, row = (f >> 2) & 15, xq = f & 3;
                    const int y = ty0 + row, x = tx0 + 4 * xq;
                    const int ch = 32 * rd + pl;
                    const int pp = pl - 32 + (GEO ? 0 : 4);
                    const bool on = y < a.H && f < (32 + npix) * 64 && (pl >= 32 || (ch < a.nc && a.dL_dfeat));
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (on) {
                        const float* src = pl < 32 ? a.dL_dfeat + (size_t)(a.c0 + ch) * HW
                                         : pp < 3 ? a.dL_dpix + (size_t)pp * HW : pp == 3 ? a.dL_ddepth : pp == 4 ? a.final_T : reinterpret_cast<const float*>(a.n_contrib);
                        const float* p = src + (size_t)y * a.W + x;
                        if (x + 0 < a.W) v.x = p[0];
                        if (x + 1 < a.W) v.y = p[1];
                        if (x + 2 < a.W) v.z = p[2];
                        if (x + 3 < a.W) v.w = p[3];
                    }
                    sv[it] = v;
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int f = it * NTH + tid;
                const int pl = f >> 6, row = (f >> 2) & 15, xq = f & 3;
                if (f < (32 + npix) * 64) *reinterpret_cast<float4*>(&stage[pl * PL_SP + row * 16 + 4 * xq]) = sv[it];
            }
            if (a.glow) {
                // The feature-map gradient arrives at the resolution of the loss (f3dgs_set_feature_grad_lowres), (gHg gWg, C)
                // pixel-major: apply the transposed bilinear resize to this tile - for every pixel the (at most 2 x 2: the image is
                // not smaller than the loss's map) output samples that read it, in the order and with the products of the loss's
                // own transposed-resize kernel (feature_loss.hip, fl_resize_backward_kernel), so the staged values are the ones the
                // dense path would have loaded (given both, they are added).  The rows and the columns of the tile that are
                // sampled at all (one in three at a 3x reduction) are compacted first; work item = (sampled row, sampled column),
                // lane = channel (one 128-byte request per sample), two items per pass with their eight requests in flight
                // together.  An absent second sample re-reads the first and is not added.
                // (Measured against this plain order - lists and samples behind the stores of the round: requesting the first pass
                // of samples beside the pixel planes, the lists built in front, is slower: c4 2.97 against 2.88 ms.)
                float4* const tl = reinterpret_cast<float4*>(smem + PL_TAPS_OFS);     // [0..15] rows, [16..31] columns: {o0, o1 (bits), w0, w1}
                int* const ti = reinterpret_cast<int*>(smem + PL_TAPS_OFS + 32 * sizeof(float4));   // their pixel row / column; [32], [33]: counts
                if (rd == 0 && tid < 64) {
                    const bool is_row = tid < 16;
                    const int k = tid & 15;
                    const int i = is_row ? ty0 + k : tx0 + k;
                    const int in = is_row ? a.H : a.W, out = is_row ? a.gHg : a.gWg;
                    int oo[2] = {0, 0};
                    float ww[2] = {0.f, 0.f};
                    int n = 0;
                    if (tid < 32 && i < in) n = resize_build<2>(i, is_row ? a.gsy : a.gsx, in, out, oo, ww);
                    const unsigned long long has = __ballot(n > 0);
                    const unsigned int mine = is_row ? (unsigned int)(has & 0xFFFFu) : (unsigned int)((has >> 16) & 0xFFFFu);
                    if (n > 0) {
                        const int slot = (is_row ? 0 : 16) + __popc(mine & ((1u << k) - 1u));
                        tl[slot] = make_float4(__int_as_float(oo[0]), __int_as_float(n > 1 ? oo[1] : oo[0]), ww[0], n > 1 ? ww[1] : 0.f);
                        ti[slot] = k;
                    }
                    if (tid == 0) { ti[32] = __popc((unsigned int)(has & 0xFFFFu)); ti[33] = __popc((unsigned int)((has >> 16) & 0xFFFFu)); }
                }
                __syncthreads();
                const int nr = ti[32], ncol = ti[33];
                const int items = nr * ncol;
                if (items > 0) {
                    const float gsc = a.gscale ? *a.gscale : 1.0f;
                    // (32-bit element offsets: the caller has checked gHg gWg C < 2^31)
                    const uint32_t ch_lane = (uint32_t)fresh_lane() & 31u;
                    const bool ch_ok = 32 * rd + (int)ch_lane < a.nc;
                    // a lane beyond the window's last channel re-reads channel 0 and drops the value: no read past the
                    // documented Hg Wg C floats of the caller's buffer when the last window is not a multiple of 32 channels
                    const uint32_t ch = ch_ok ? ch_lane : 0u;
                    const float* const gbase = a.glow + a.c0 + 32 * rd;
                    const uint32_t C_ = (uint32_t)a.C, Wg_ = (uint32_t)a.gWg;
                    const uint32_t inv = (65536u + (uint32_t)ncol - 1u) / (uint32_t)ncol;      // i / ncol = (i inv) >> 16 for i < 256
                    constexpr int HWV = NTH / 32;          // half waves of the workgroup: two items each per pass
#pragma unroll 1
                    for (int i0 = tid >> 5; i0 < items; i0 += 2 * HWV) {
                        float v[2][4], wy0[2], wy1[2], wx0[2], wx1[2];
                        int sp[2];
                        bool live[2];
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const int it = i0 + HWV * u;
                            live[u] = it < items && ch_ok;
                            const int itc = it < items ? it : i0;
                            const int ri = (int)(((uint32_t)itc * inv) >> 16), ci = itc - ri * ncol;
                            const float4 yl = tl[ri], xl = tl[16 + ci];
                            sp[u] = ti[ri] * 16 + ti[16 + ci];
                            const uint32_t y0o = (uint32_t)__float_as_int(yl.x) * Wg_, y1o = (uint32_t)__float_as_int(yl.y) * Wg_;
                            const uint32_t x0o = (uint32_t)__float_as_int(xl.x), x1o = (uint32_t)__float_as_int(xl.y);
                            wy0[u] = yl.z; wy1[u] = yl.w; wx0[u] = xl.z; wx1[u] = xl.w;
                            v[u][0] = gbase[(y0o + x0o) * C_ + ch]; v[u][1] = gbase[(y0o + x1o) * C_ + ch];
                            v[u][2] = gbase[(y1o + x0o) * C_ + ch]; v[u][3] = gbase[(y1o + x1o) * C_ + ch];
                        }
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            float rr = fmaf(wx0[u], v[u][0], 0.f);
                            rr = wx1[u] != 0.f ? fmaf(wx1[u], v[u][1], rr) : rr;
                            float acc = fmaf(wy0[u], rr, 0.f);
                            float r2 = fmaf(wx0[u], v[u][2], 0.f);
                            r2 = wx1[u] != 0.f ? fmaf(wx1[u], v[u][3], r2) : r2;
                            acc = wy1[u] != 0.f ? fmaf(wy1[u], r2, acc) : acc;
                            if (live[u]) stage[ch_lane * PL_SP + sp[u]] += a.gscale ? acc * gsc : acc;
                        }
                    }
                }
            }
            PL_STAGE_MARK(0);
            __syncthreads();
            PL_STAGE_MARK(1);
            if (rd == 0 && q < 4) {
                const int ln = fresh_lane();
                const int sp = (qy + (ln >> 3)) * 16 + qx + (ln & 7);
                constexpr int pb = 32 - (GEO ? 0 : 4);        // plane of dL/dR (GEO); final_T at pb + 4, n_contrib at pb + 5
                last = __float_as_uint(stage[(pb + 5) * PL_SP + sp]);     // 0 outside the image (zero-filled)
                T = stage[(pb + 4) * PL_SP + sp];
                if constexpr (GEO) {
                    dR = stage[(pb + 0) * PL_SP + sp]; dG = stage[(pb + 1) * PL_SP + sp]; dB = stage[(pb + 2) * PL_SP + sp];
                    dD = stage[(pb + 3) * PL_SP + sp];
                }
            }
            // which plane is this wave's column `col`?  (-1: a zero column)
            const int col = fresh_lane() & 15, kk = fresh_lane() >> 4;
            int plane = -1;
            bool mine = false;          // does this round hold this wave's block
            if constexpr (GEO) {
                mine = true;
                if (q < 2) plane = 16 * q + col;
                else if (q == 2) plane = col < 4 ? 32 + col : -1;
            } else {
                mine = (q >> 1) == rd;
                plane = 16 * (q & 1) + col;
            }
            if constexpr (BF) {
                if (GEO && q == 3) {
                    // the moment wave: monomials of the pixel offset from the quadrant centre, columns 0..5: 1, u, v, u^2, uv, v^2
                    // (at most six significant bits: exact in bf16); one operand for the four quadrants, kept in Bbf[0]
                    const float k0c = col == 0 ? 1.f : 0.f, k1c = col == 1 ? 1.f : 0.f, k2c = col == 2 ? 1.f : 0.f;
                    const float k3c = col == 3 ? 1.f : 0.f, k4c = col == 4 ? 1.f : 0.f, k5c = col == 5 ? 1.f : 0.f;
                    if constexpr (S32) {
                        // hybrid: the fp32 operand of the exact shape - step t of a quadrant's tile contracts pixel (4 (t & 1) + kk,
                        // t >> 1) at K index kk - kept (as bits) in the registers the bf16 operand would take
#pragma unroll
                        for (int t = 0; t < 16; t++) {
                            const float uu = (float)(4 * (t & 1) + kk) - 3.5f, vv = (float)(t >> 1) - 3.5f;
                            Bbf[0][t >> 3][t & 7] = __float_as_uint(fmaf(fmaf(k3c, uu, fmaf(k4c, vv, k1c)), uu, fmaf(fmaf(k5c, vv, k2c), vv, k0c)));
                        }
                    } else
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) {
                        const float vv = (float)(4 * ks + kk) - 3.5f;
                        float m[8];
#pragma unroll
                        for (int x = 0; x < 8; x++) {
                            const float uu = (float)x - 3.5f;
                            m[x] = fmaf(fmaf(k3c, uu, fmaf(k4c, vv, k1c)), uu, fmaf(fmaf(k5c, vv, k2c), vv, k0c));
                        }
#pragma unroll
                        for (int d2 = 0; d2 < 4; d2++) { Bbf[0][ks][d2] = pack_bf16(m[2 * d2], m[2 * d2 + 1]); Bbf[0][ks][4 + d2] = 0u; }
                    }
                } else if (mine) {
                    // feature-like blocks (GEO: channels 0-15, 16-31, colour / depth; later windows: sixteen channels per wave)
#pragma unroll
                    for (int qd = 0; qd < 4; qd++)
#pragma unroll
                        for (int ks = 0; ks < 2; ks++) {
                            const int py = (qd >> 1) * 8 + 4 * ks + kk, px0 = (qd & 1) * 8;
                            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                            if (plane >= 0) {
                                v0 = *reinterpret_cast<const float4*>(&stage[plane * PL_SP + py * 16 + px0]);
                                v1 = *reinterpret_cast<const float4*>(&stage[plane * PL_SP + py * 16 + px0 + 4]);
                            }
                            split_bf16(v0.x, v0.y, Bbf[qd][ks][0], Bbf[qd][ks][4]);
                            split_bf16(v0.z, v0.w, Bbf[qd][ks][1], Bbf[qd][ks][5]);
                            split_bf16(v1.x, v1.y, Bbf[qd][ks][2], Bbf[qd][ks][6]);
                            split_bf16(v1.z, v1.w, Bbf[qd][ks][3], Bbf[qd][ks][7]);
                        }
                }
            } else if (GEO && M44 && q == 2) {
                // the colour wave on sixteen 4 x 4 blocks (v_mfma_f32_4x4x1_16B_f32): lane = (block, i) with block = (entry group
                // rg = block & 3, pixel class kc = block >> 2) and i = its row in A / its COLUMN in B (dL/dR, dL/dG, dL/dB,
                // dL/ddepth).  A step (pixel block 4 t + kc of a quadrant tile, column m) contracts pixel (kc + 4 (m & 1),
                // 2 t + (m >> 1)) of the quadrant: Bop[qd][4 t + m] holds column i of the operand at that pixel.
                const int kc4 = lane >> 4, i4 = lane & 3;
#pragma unroll
                for (int qd = 0; qd < 4; qd++)
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        const int px = (qd & 1) * 8 + kc4 + 4 * (t & 1), py = (qd >> 1) * 8 + 2 * (t >> 2) + ((t >> 1) & 1);
                        Bop[qd][t] = stage[(32 + i4) * PL_SP + py * 16 + px];
                    }
            } else if (GEO && split16 && q != 2) {
                // Up to 16 channels (BwdArgs::split16): one feature block only - it is split over waves 0 and 1 by QUADRANTS (0, 1 | 2, 3:
                // two partial sums per entry, added in the flush), and the moment block over waves 0, 1 (quadrant 0 | 1) and 3
                // (quadrants 2, 3): 1536 / 1536 / 512 / 1024 matrix cycles per chunk instead of 2048 / 0 / 512 / 2048.
                // Operand rows: waves 0, 1: Bop[0], Bop[1] = the feature columns at their two quadrants, Bop[2] = the monomials;
                // wave 3: Bop[0] = the monomials.
                const float k0c = col == 0 ? 1.f : 0.f, k1c = col == 1 ? 1.f : 0.f, k2c = col == 2 ? 1.f : 0.f;
                const float k3c = col == 3 ? 1.f : 0.f, k4c = col == 4 ? 1.f : 0.f, k5c = col == 5 ? 1.f : 0.f;
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const float uu = (float)(4 * (t & 1) + kk) - 3.5f, vv = (float)(t >> 1) - 3.5f;
                    const float v = fmaf(fmaf(k3c, uu, fmaf(k4c, vv, k1c)), uu, fmaf(fmaf(k5c, vv, k2c), vv, k0c));
                    if (q == 3) Bop[0][t] = v; else Bop[2][t] = v;
                }
                if (q < 2) {
#pragma unroll
                    for (int sl = 0; sl < 2; sl++)
#pragma unroll
                        for (int t = 0; t < 16; t++) {
                            const int qd = 2 * q + sl;
                            const int px = (qd & 1) * 8 + 4 * (t & 1) + kk, py = (qd >> 1) * 8 + (t >> 1);
                            Bop[sl][t] = col < a.nc ? stage[col * PL_SP + py * 16 + px] : 0.f;
                        }
                }
            } else if (GEO && q == 3) {
                // the moment wave: monomials of the pixel offset from the QUADRANT centre (the moments of each quadrant are kept
                // apart and re-centred on the splat mean one by one: |u|, |v| <= 3.5), columns 0..5: 1, u, v, u^2, uv, v^2 - the
                // same operand for the four quadrants, evaluated as one polynomial with per-lane one-hot coefficients
                const float k0c = col == 0 ? 1.f : 0.f, k1c = col == 1 ? 1.f : 0.f, k2c = col == 2 ? 1.f : 0.f;
                const float k3c = col == 3 ? 1.f : 0.f, k4c = col == 4 ? 1.f : 0.f, k5c = col == 5 ? 1.f : 0.f;
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const float uu = (float)(4 * (t & 1) + kk) - 3.5f, vv = (float)(t >> 1) - 3.5f;
                    const float v = fmaf(fmaf(k3c, uu, fmaf(k4c, vv, k1c)), uu, fmaf(fmaf(k5c, vv, k2c), vv, k0c));
#pragma unroll
                    for (int qd = 0; qd < 4; qd++) Bop[qd][t] = v;
                }
            } else if (split32) {
#pragma unroll
                for (int sl = 0; sl < 2; sl++)
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        const int qd = 2 * (q >> 1) + sl;
                        const int px = (qd & 1) * 8 + 4 * (t & 1) + kk, py = (qd >> 1) * 8 + (t >> 1);
                        Bop[sl][t] = stage[plane * PL_SP + py * 16 + px];       // plane = 16 (q & 1) + col (zero-filled beyond nc)
                    }
            } else if (mine) {
#pragma unroll
                for (int qd = 0; qd < 4; qd++)
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        const int px = (qd & 1) * 8 + 4 * (t & 1) + kk, py = (qd >> 1) * 8 + (t >> 1);
                        Bop[qd][t] = plane >= 0 ? stage[plane * PL_SP + py * 16 + px] : 0.f;
                    }
            }
        }
    }
    PL_STAGE_MARK(2);
    float S = T * (a.bg[0] * dR + a.bg[1] * dG + a.bg[2] * dB);
    // wave-uniform floats of the flush come in as kernel arguments (scalar registers): computed here, on the vector pipe, they would
    // occupy vector registers across the whole walk
    const float neg_half_w = a.neg_half_w, neg_half_h = a.neg_half_h;
    const int lx = fresh_lane() & 7, ly = fresh_lane() >> 3;
    const float pxf = (float)(tx0 + qx + lx), pyf = (float)(ty0 + qy + ly);
    const uint32_t my_max = wave_max_u32(last);
    // does this wave hold a column block at all?
    const bool active = (GEO && q >= 2) || 16 * q < a.nc;
    const bool use_s = GEO && q == 3;
    __syncthreads();                                // every read of the staging image is done: the area is re-used from here on
    PL_STAGE_MARK(3);
    if (lane == 0 && q < 4) L.wave_max[q] = my_max;
    if (tid < 2) L.touched[tid] = 0;
    __syncthreads();
    PL_STAGE_MARK(4);
    const uint32_t tile_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)max(max(L.wave_max[0], L.wave_max[1]), max(L.wave_max[2], L.wave_max[3])));
    const int n_win = PL_DEV_SKIP(32) ? 0 : (int)((tile_max + PL_WIN - 1) / PL_WIN);

    // LDS offsets (dwords): phase-1 store column of this lane's pixel, operand read base of this lane's (row, K index)
    const int ccol = 16 * (ly >> 1) + 4 * (lx & 3) + 2 * (ly & 1) + (lx >> 2);
    const uint32_t wofs_b = (uint32_t)(((ccol >> 2) * PL_ROW + ((ccol >> 2) & 7) * 4 + (ccol & 3)) * 4);     // bytes, row 0
    char* const my_wt = reinterpret_cast<char*>(&L.wt[q][0]);
    constexpr int ST_OFS = 4 * PL_TILE;             // st[q] - wt[q], dwords
    // BF: the bf16 tiles alias wt / st (32 KB); this lane's pixel in its quadrant's rows: unit ly (XOR-ed per row), element lx
    char* const bf_tiles = reinterpret_cast<char*>(&L.wt[0][0]);
    const uint32_t bf_sofs = (uint32_t)(q * BF_QUAD + ly * 16 + lx * 2);
        if (!PL_DEV_SKIP(32)) PL_PHASE_END(0);

    // ---- record loader (waves 0..2: one 16-byte third of every record; lane = entry, entry 0 = farthest back) -------
    auto win_pos = [&](int w) -> uint32_t { return (uint32_t)(w * PL_WIN + PL_WIN - 1 - fresh_lane()); };
    auto load_id = [&](int w) -> uint32_t {
        const uint32_t pos = win_pos(w);
        return (w >= 0 && q < 3 && pos < tile_max) ? a.point_list[r_lo + pos] : 0u;
    };
    auto load_part = [&](int w, uint32_t gid) -> float4 {
        const uint32_t pos = win_pos(w);
        if (w >= 0 && q < 3 && pos < tile_max) return reinterpret_cast<const float4*>(a.rec + gid)[q];
        return make_float4(0.f, 0.f, 0.f, 0.f);       // opacity 0: never a hit
    };
    uint32_t gid_cur = load_id(n_win - 1), gid_nxt = load_id(n_win - 2);
    float4 part_cur = load_part(n_win - 1, gid_cur);
    int parity = 0;

    // records of window w -> LDS (by the three loader waves), the next window's requests go out
    auto store_window = [&](int w) {
        if (q < 3 && w >= 0) {
            // SplatRec thirds {mean, a, b | c, opacity, r, g | b, depth, ..} -> PlRec {mean, a', b' | c', opacity, id, - | r, g, b, depth}
            const float4 v = part_cur;
            PlRec& dst = L.rec[lane];
            if (q == 0) {
                dst.q0 = make_float4(v.x, v.y, v.z * CONIC_SCALE_AC, v.w * CONIC_SCALE_B);
            } else if (q == 1) {
                *reinterpret_cast<float2*>(&dst.q1.x) = make_float2(v.x * CONIC_SCALE_AC, v.y);
                *reinterpret_cast<float2*>(&dst.q2.x) = make_float2(v.z, v.w);
            } else {
                dst.q1.z = __uint_as_float(gid_cur);
                *reinterpret_cast<float2*>(&dst.q2.z) = make_float2(v.x, v.y);
            }
        }
        gid_cur = gid_nxt;
        part_cur = load_part(w - 1, gid_cur);
        gid_nxt = load_id(w - 2);
    };
    store_window(n_win - 1);
    lds_barrier();

    for (int w = n_win - 1; w >= 0; w--) {
        PlRec* const rec = L.rec;
        PL_PHASE_END(1);

        const uint32_t k0 = (uint32_t)w * PL_WIN;
        // entries of the window that exist: positions beyond the end of the list sit at the BACK (low entries) of the top window
        const int n_have = (int)min((uint32_t)PL_WIN, tile_max - k0);
        const int j_first = (PL_WIN - n_have) / PL_CAP;

        for (int j = j_first; j < PL_WIN / PL_CAP; j++, parity ^= 1) {
            PL_COUNT(5, 1);
            // ---- phase 1: rows of the chunk = entries 16 j .. 16 j + 15 of the window.  Every entry is evaluated (an entry that
            // misses the quadrant yields w = s = 0 at each of its pixels by itself): one straight-line block per chunk, records
            // of the next pair in flight while a pair is computed.  A quadrant whose pixels all ended behind the chunk skips it.
            const uint32_t pos_hi = k0 + (uint32_t)(PL_WIN - 1 - 16 * j);      // list position of row 0
            uint32_t tm = 0;      // rows that blended somewhere in this quadrant; bit 16 + q: the quadrant's tiles are live
            if (lane < FROWS) {
                // what the flush of this wave's rows needs of their splats, kept beside the sums: the record window may
                // be replaced while the chunk is flushed
                const PlRec& rr = rec[16 * j + FROWS * q + lane];
                float* const info = &L.ftile[(FROWS * q + lane) * FS];
                if constexpr (GEO) {
                    const float4 i0 = rr.q0, i1 = rr.q1;
                    *reinterpret_cast<float4*>(info + 60) = i0;
                    *reinterpret_cast<float4*>(info + 64) = make_float4(i1.x, i1.y, i1.z, 0.f);
                } else {
                    info[GID_SLOT] = rr.q1.z;
                }
            }
            if (q < 4 && pos_hi - 15 < my_max && !PL_DEV_SKIP(2)) {
                const PlRec* rc = &rec[16 * j];
                auto phase1 = [&](auto nec, auto prefc) {
                    constexpr int NE = decltype(nec)::value;
                    constexpr bool PREF = decltype(prefc)::value;
                    float4 n0[NE], n2[NE];
                    float2 n1[NE];
                    float w_even = 0.f;       // BF, later windows: the even entry's weight waits for its odd neighbour
                    (void)w_even;
                    if constexpr (PREF) {
#pragma unroll
                        for (int k = 0; k < NE; k++) {
                            n0[k] = rc[k].q0; n1[k] = *reinterpret_cast<const float2*>(&rc[k].q1);
                            if constexpr (GEO) n2[k] = rc[k].q2;
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 16 / NE; p++) {
                        float4 e0[NE], e2[NE];
                        float2 e1[NE];
                        if constexpr (PREF) {
#pragma unroll
                            for (int k = 0; k < NE; k++) { e0[k] = n0[k]; e1[k] = n1[k]; e2[k] = n2[k]; }
                            if (p < 16 / NE - 1) {
#pragma unroll
                                for (int k = 0; k < NE; k++) {
                                    n0[k] = rc[NE * (p + 1) + k].q0; n1[k] = *reinterpret_cast<const float2*>(&rc[NE * (p + 1) + k].q1);
                                    if constexpr (GEO) n2[k] = rc[NE * (p + 1) + k].q2;
                                }
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < NE; k++) {
                                e0[k] = rc[NE * p + k].q0; e1[k] = *reinterpret_cast<const float2*>(&rc[NE * p + k].q1);
                                if constexpr (GEO) e2[k] = rc[NE * p + k].q2;
                            }
                        }
                        PL_COUNT(6, NE);
                        float au[NE], al[NE], f[NE], qd[NE];
#pragma unroll
                        for (int k = 0; k < NE; k++) {
                            const float dx = e0[k].x - pxf, dy = e0[k].y - pyf;
                            const float power2 = splat_power2(dx, dy, e0[k].z, e0[k].w, e1[k].x);
                            const float v = e1[k].y * __builtin_amdgcn_exp2f(power2);                  // op G, not yet clamped
                            // the three tests as lane masks (scalar ANDs); the select takes the mask as it stands
                            const unsigned long long okm = __ballot(pos_hi - (uint32_t)(NE * p + k) < last) & __ballot(!(power2 > 0.0f)) & __ballot(!(v < ALPHA_MIN));
                            if (okm) tm |= 1u << (NE * p + k);
                            // exp2 may be inf where power > 0: selected away, never multiplied.  The clamp sits in the same block: behind
                            // an opaque value the compiler puts a canonicalising v_max in front of the fminf
                            static_assert(ALPHA_MAX == 0.99f, "literal below");
                            asm("v_cndmask_b32_e64 %0, 0, %2, %3\n\tv_min_f32_e32 %1, 0x3f7d70a4, %0" : "=&v"(au[k]), "=v"(al[k]) : "v"(v), "s"(okm));
                            f[k] = __builtin_amdgcn_rcpf(1.f - al[k]);                                 // exactly 1 for skipped pairs
                            if constexpr (GEO) qd[k] = fmaf(e2[k].x, dR, fmaf(e2[k].y, dG, fmaf(e2[k].z, dB, e2[k].w * dD)));
                        }
#pragma unroll
                        for (int k = 0; k < NE; k++) {
                            const float Tb = T * f[k];              // transmittance in front of this splat
                            const float wv = al[k] * Tb;
                            if constexpr (BF) {
                                // two bf16 terms per value; row e of the quadrant's tiles, this lane's pixel (see BF_QUAD)
                                const int e = NE * p + k;
                                char* const bp = bf_tiles + (bf_sofs ^ (uint32_t)(16 * (e >> 1))) + e * BF_ROWB;
                                if constexpr (GEO) {
                                    const float dL_dalpha = fmaf(Tb, qd[k], -(S * f[k]));
                                    S = fmaf(wv, qd[k], S);
                                    uint32_t h, m;
                                    split_bf16(wv, au[k] * dL_dalpha, h, m);      // low halves: w, high halves: s
                                    *reinterpret_cast<uint16_t*>(bp) = (uint16_t)h;
                                    *reinterpret_cast<uint16_t*>(bp + BF_TERM) = (uint16_t)m;
                                    *reinterpret_cast<uint16_t*>(bp + 2 * BF_TERM) = (uint16_t)(h >> 16);
                                    *reinterpret_cast<uint16_t*>(bp + 3 * BF_TERM) = (uint16_t)(m >> 16);
                                } else if ((e & 1) == 0) {
                                    w_even = wv;                                     // converted together with the next entry's
                                } else {
                                    uint32_t h, m;
                                    split_bf16(w_even, wv, h, m);                    // rows e - 1 and e share their XOR term
                                    *reinterpret_cast<uint16_t*>(bp - BF_ROWB) = (uint16_t)h;
                                    *reinterpret_cast<uint16_t*>(bp - BF_ROWB + BF_TERM) = (uint16_t)m;
                                    *reinterpret_cast<uint16_t*>(bp) = (uint16_t)(h >> 16);
                                    *reinterpret_cast<uint16_t*>(bp + BF_TERM) = (uint16_t)(m >> 16);
                                }
                            } else {
                            float* const wp = reinterpret_cast<float*>(my_wt + (wofs_b ^ (uint32_t)(16 * (NE * p + k))));
                            wp[0] = wv;
                            if constexpr (GEO) {
                                const float dL_dalpha = fmaf(Tb, qd[k], -(S * f[k]));
                                S = fmaf(wv, qd[k], S);
                                wp[ST_OFS] = au[k] * dL_dalpha;
                            }
                            }
                            T = Tb;
                        }
                    }
                };
                if constexpr (BF) {
                    const P1Pixel px{pxf, pyf, last, dR, dG, dB, dD};
                    if constexpr (S32) {
                        // this lane's slot in row 0 of the quadrant's fp32 s tile, rebuilt per chunk (see wofs_b: nothing lane-derived
                        // lives across the walk)
                        const int fl = fresh_lane(), sx = fl & 7, sy = fl >> 3;
                        const int sc = 16 * (sy >> 1) + 4 * (sx & 3) + 2 * (sy & 1) + (sx >> 2);
                        const uint32_t s_ofs = (uint32_t)(((sc >> 2) * PL_ROW + ((sc >> 2) & 7) * 4 + (sc & 3)) * 4);
                        tm = pl_phase1_bf16<GEO, 0, false, true>(rc, px, pos_hi, T, S, bf_tiles, bf_sofs,     // (schedule 0: the pinned pipeline costs this shape 12 spilled registers)
                                                                        bf_tiles + q * BF_QUAD + 2 * BF_TERM, s_ofs);
                    } else
                        tm = pl_phase1_bf16<GEO, PL_SCHED>(rc, px, pos_hi, T, S, bf_tiles, bf_sofs);
                } else {
                    phase1(std::integral_constant<int, P1_NE>{}, std::integral_constant<bool, P1_PREF>{});
                }
                tm |= 1u << (16 + q);
            }
            if (tm && lane == 0) atomicOr(&L.touched[parity], tm);
            PL_PHASE_END(2);
            lds_barrier();                                                  // B_a: the A tiles of the chunk are complete
            if (j == PL_WIN / PL_CAP - 1) {
                // phase 1 of the window's last chunk is done everywhere: the record buffer is free.  The loads of window w - 1 -
                // requested a window ago - are older than any atomic still in flight: no wait on the flush traffic
                store_window(w - 1);
            }
            PL_PHASE_END(1);

            // ---- phase 2: this wave's sixteen columns of every sum of the chunk
            const uint32_t tt = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.touched[parity]);
            if constexpr (BF) {
                if (active && (tt & 0xFFFFu) != 0) {
                    const int lane2 = fresh_lane();
                    const int col = lane2 & 15, kk = lane2 >> 4;
                    // operand read base of lane (row col, K group kk): unit 4 ks + kk at slot (4 ks + kk) ^ (row >> 1)
                    const uint32_t r0 = (uint32_t)(col * BF_ROWB + ((kk ^ (col >> 1)) & 7) * 16), r1 = r0 ^ 64u;
                    const int fs_row0 = (4 * kk) * FS;
                    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    auto a_read = [&](int qd, int term, int ks, uint32_t (&dst)[4]) {
                        const uint4 v = *reinterpret_cast<const uint4*>(bf_tiles + qd * BF_QUAD + term * BF_TERM + (ks ? r1 : r0));
                        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                    };
                    if (use_s) {
                        // the moment wave: s = s_hi + s_mid against the exact monomials, the four quadrants kept apart
#pragma unroll
                        for (int qd = 0; qd < 4; qd++) {
                            f32x4 acc0 = zero4, acc1 = zero4;
                            if constexpr (S32) {
                                // hybrid: s is an fp32 tile (the exact shape's layout) behind the quadrant's two w planes
                                if ((tt >> (16 + qd)) & 1u) {
                                    const float* const st = reinterpret_cast<const float*>(bf_tiles + qd * BF_QUAD + 2 * BF_TERM);
                                    const int rofs0 = kk * PL_ROW + (col ^ kk) * 4, rofs1 = rofs0 ^ 16;
#pragma unroll
                                    for (int u = 0; u < 4; u++) {
                                        const float4 av = *reinterpret_cast<const float4*>(st + ((u & 1) ? rofs1 : rofs0) + u * 4 * PL_ROW);
                                        auto B = [&](int t) { return __uint_as_float(Bbf[0][t >> 3][t & 7]); };
                                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, B(4 * u + 0), acc0, 0, 0, 0);
                                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, B(4 * u + 1), acc1, 0, 0, 0);
                                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, B(4 * u + 2), acc0, 0, 0, 0);
                                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, B(4 * u + 3), acc1, 0, 0, 0);
                                    }
                                }
                            } else
                            if ((tt >> (16 + qd)) & 1u) {
#pragma unroll
                                for (int ks = 0; ks < 2; ks++) {
                                    uint32_t ah[4], am[4];
                                    a_read(qd, 2, ks, ah); a_read(qd, 3, ks, am);
                                    const uint32_t (&bm)[4] = reinterpret_cast<const uint32_t (&)[4]>(Bbf[0][ks][0]);
                                    acc0 = mfma_bf16(ah, bm, acc0);
                                    acc1 = mfma_bf16(am, bm, acc1);
                                }
                            }
                            if (col < 6) {
#pragma unroll
                                for (int r = 0; r < 4; r++) L.ftile[fs_row0 + r * FS + 36 + 6 * qd + col] = acc0[r] + acc1[r];
                            }
                        }
                    } else {
                        // w g = w_hi g_hi + w_hi g_mid + w_mid g_hi: three accumulators, no back-to-back dependent instructions
                        f32x4 acc0 = zero4, acc1 = zero4, acc2 = zero4;
#pragma unroll
                        for (int qd = 0; qd < 4; qd++) {
                            if ((tt >> (16 + qd)) & 1u) {
#pragma unroll
                                for (int ks = 0; ks < 2; ks++) {
                                    uint32_t ah[4], am[4];
                                    a_read(qd, 0, ks, ah); a_read(qd, 1, ks, am);
                                    const uint32_t (&bh)[4] = reinterpret_cast<const uint32_t (&)[4]>(Bbf[qd][ks][0]);
                                    const uint32_t (&bm)[4] = reinterpret_cast<const uint32_t (&)[4]>(Bbf[qd][ks][4]);
                                    acc0 = mfma_bf16(ah, bh, acc0);
                                    acc1 = mfma_bf16(ah, bm, acc1);
                                    acc2 = mfma_bf16(am, bh, acc2);
                                }
                            }
                        }
                        const int slot = (GEO && q == 2) ? 32 + col : 16 * q + col;
                        if (!(GEO && q == 2) || col < 4) {
#pragma unroll
                            for (int r = 0; r < 4; r++) L.ftile[fs_row0 + r * FS + slot] = acc0[r] + (acc1[r] + acc2[r]);
                        }
                    }
                }
            } else if (GEO && M44 && q == 2 && (tt & 0xFFFFu) != 0) {
                // ---- colour wave, 4 x 4 blocks: 8 matrix-pipe cycles per step of 16 entries x 4 pixels x 4 columns instead of 32 for a
                // 16-column step of which 4 are used (c3: 0.645 -> 0.630 ms; the moment wave - 6 columns, two blocks, 128
                // instructions per chunk - was measured too and is slower in this form: 0.71 ms)
                const int lane2 = fresh_lane();
                const int i4 = lane2 & 3, rg = (lane2 >> 2) & 3, kc4 = lane2 >> 4;
                const int row = 4 * rg + i4;
                const float* abase = &L.wt[0][0];
                f32x4 accA = f32x4{0.f, 0.f, 0.f, 0.f}, accB = accA;
                // (pixel block 4 t + kc: the swizzle term (pb & 7) = (4 t + kc) & 7 alternates between kc and kc + 4 with t)
                const int ofs_e = (kc4 * PL_ROW + 4 * (row ^ kc4)), ofs_o = ((4 + kc4) * PL_ROW + 4 * (row ^ (4 + kc4)));
#pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    if ((tt >> (16 + qd)) & 1u) {
#pragma unroll
                        for (int th = 0; th < 2; th++) {      // pixel blocks 4 t + kc, t = 2 th, 2 th + 1: two tile-row reads in flight
                            const float4 av0 = *reinterpret_cast<const float4*>(abase + qd * PL_TILE + ofs_e + th * 8 * PL_ROW);
                            const float4 av1 = *reinterpret_cast<const float4*>(abase + qd * PL_TILE + ofs_o + th * 8 * PL_ROW);
                            accA = __builtin_amdgcn_mfma_f32_4x4x1f32(av0.x, Bop[qd][8 * th + 0], accA, 0, 0, 0);
                            accB = __builtin_amdgcn_mfma_f32_4x4x1f32(av0.y, Bop[qd][8 * th + 1], accB, 0, 0, 0);
                            accA = __builtin_amdgcn_mfma_f32_4x4x1f32(av0.z, Bop[qd][8 * th + 2], accA, 0, 0, 0);
                            accB = __builtin_amdgcn_mfma_f32_4x4x1f32(av0.w, Bop[qd][8 * th + 3], accB, 0, 0, 0);
                            accA = __builtin_amdgcn_mfma_f32_4x4x1f32(av1.x, Bop[qd][8 * th + 4], accA, 0, 0, 0);
                            accB = __builtin_amdgcn_mfma_f32_4x4x1f32(av1.y, Bop[qd][8 * th + 5], accB, 0, 0, 0);
                            accA = __builtin_amdgcn_mfma_f32_4x4x1f32(av1.z, Bop[qd][8 * th + 6], accA, 0, 0, 0);
                            accB = __builtin_amdgcn_mfma_f32_4x4x1f32(av1.w, Bop[qd][8 * th + 7], accB, 0, 0, 0);
                        }
                    }
                }
                // the four pixel classes hold partial sums of the same (entry, column): add them up (lanes 16 / 32 apart)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = accA[r] + accB[r];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (kc4 == 0) L.ftile[(4 * rg + r) * PL_FS + 32 + i4] = v;
                }
            } else if (GEO && split16 && q != 2 && (tt & 0xFFFFu) != 0) {
                // ---- up to 16 channels: quadrant-split feature block and moment block (see the operand read-out)
                const int lane2 = fresh_lane();
                const int col = lane2 & 15, kk = lane2 >> 4;
                const int rofs0 = kk * PL_ROW + (col ^ kk) * 4, rofs1 = rofs0 ^ 16;
                const int fs_row0 = (4 * kk) * PL_FS;
                const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
                auto contract = [&](const float* abase, int qd, const float (&B)[16], f32x4& c0, f32x4& c1) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float4 av = *reinterpret_cast<const float4*>(abase + ((u & 1) ? rofs1 : rofs0) + qd * PL_TILE + u * 4 * PL_ROW);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, B[4 * u + 0], c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, B[4 * u + 1], c1, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, B[4 * u + 2], c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, B[4 * u + 3], c1, 0, 0, 0);
                    }
                };
                auto live = [&](int qd) { return ((tt >> (16 + qd)) & 1u) != 0; };
                if (q < 2) {
                    f32x4 acc0 = zero4, acc1 = zero4;
                    if (live(2 * q)) contract(&L.wt[0][0], 2 * q, Bop[0], acc0, acc1);
                    if (live(2 * q + 1)) contract(&L.wt[0][0], 2 * q + 1, Bop[1], acc0, acc1);
#pragma unroll
                    for (int r = 0; r < 4; r++) L.ftile[fs_row0 + r * PL_FS + 16 * q + col] = acc0[r] + acc1[r];     // partial sum q of 2
                    acc0 = zero4; acc1 = zero4;
                    if (live(q)) contract(&L.st[0][0], q, Bop[2], acc0, acc1);
                    if (col < 6) {
#pragma unroll
                        for (int r = 0; r < 4; r++) L.ftile[fs_row0 + r * PL_FS + 36 + 6 * q + col] = acc0[r] + acc1[r];
                    }
                } else {
#pragma unroll
                    for (int qd = 2; qd < 4; qd++) {
                        f32x4 acc0 = zero4, acc1 = zero4;
                        if (live(qd)) contract(&L.st[0][0], qd, Bop[0], acc0, acc1);
                        if (col < 6) {
#pragma unroll
                            for (int r = 0; r < 4; r++) L.ftile[fs_row0 + r * PL_FS + 36 + 6 * qd + col] = acc0[r] + acc1[r];
                        }
                    }
                }
            } else if (split32 && (tt & 0xFFFFu) != 0) {
                const int lane2 = fresh_lane();
                const int col = lane2 & 15, kk = lane2 >> 4;
                const int rofs0 = kk * PL_ROW + (col ^ kk) * 4, rofs1 = rofs0 ^ 16;
                const int fs_row0 = (4 * kk) * PL_FS;
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
                for (int sl = 0; sl < 2; sl++) {
                    const int qd = 2 * (q >> 1) + sl;
                    if ((tt >> (16 + qd)) & 1u) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const float4 av = *reinterpret_cast<const float4*>(&L.wt[0][0] + ((u & 1) ? rofs1 : rofs0) + qd * PL_TILE + u * 4 * PL_ROW);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, Bop[sl][4 * u + 0], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, Bop[sl][4 * u + 1], acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, Bop[sl][4 * u + 2], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, Bop[sl][4 * u + 3], acc1, 0, 0, 0);
                        }
                    }
                }
                // partial sum q >> 1 of 2: slots 32 (q >> 1) + 16 (q & 1) + col
#pragma unroll
                for (int r = 0; r < 4; r++) L.ftile[fs_row0 + r * PL_FS + 32 * (q >> 1) + 16 * (q & 1) + col] = acc0[r] + acc1[r];
            } else if (active && (tt & 0xFFFFu) != 0) {
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                const float* abase = (use_s ? &L.st[0][0] : &L.wt[0][0]);
                const int lane2 = fresh_lane();
                const int col = lane2 & 15, kk = lane2 >> 4;
                const int rofs0 = kk * PL_ROW + (col ^ kk) * 4, rofs1 = rofs0 ^ 16;                          // even / odd u
                const int fs_row0 = (4 * kk) * PL_FS;
#pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    if ((tt >> (16 + qd)) & 1u) {              // the quadrant's tiles were written for this chunk
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const float4 av = *reinterpret_cast<const float4*>(abase + ((u & 1) ? rofs1 : rofs0) + qd * PL_TILE + u * 4 * PL_ROW);
                            if (PL_DEV_SKIP(4)) { acc0[0] += av.x + av.y + av.z + av.w; continue; }
                            // two accumulators alternate: no back-to-back dependent matrix instructions
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, Bop[qd][4 * u + 0], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, Bop[qd][4 * u + 1], acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, Bop[qd][4 * u + 2], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, Bop[qd][4 * u + 3], acc1, 0, 0, 0);
                        }
                    }
                    if (use_s) {
                        // the moment wave keeps the four quadrants apart (each is re-centred on the splat mean by itself)
                        if (col < 6) {
#pragma unroll
                            for (int r = 0; r < 4; r++) L.ftile[fs_row0 + r * PL_FS + 36 + 6 * qd + col] = acc0[r] + acc1[r];
                        }
                        acc0 = acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                // D[i][j]: lane holds column j = lane & 15, register r holds row (entry) i = 4 (lane >> 4) + r.
                // Flush-tile slots: features at their channel; colour / depth sums at 32..35, the 4 x 6 moments at 36..59.
                if (!use_s) {
                    const int slot = (GEO && q == 2) ? 32 + col : 16 * q + col;
                    if (!(GEO && q == 2) || col < 4) {
#pragma unroll
                        for (int r = 0; r < 4; r++) L.ftile[fs_row0 + r * PL_FS + slot] = acc0[r] + acc1[r];
                    }
                }
            }
            PL_PHASE_END(3);
            lds_barrier();                                                  // B_b: tiles read, flush tile written
            PL_PHASE_END(1);

            // ---- flush: wave q owns rows 4 q .. 4 q + 3 of the chunk
            if ((tt & 0xFFFFu) != 0 && !PL_DEV_SKIP(16)) {
                const uint32_t m4 = (tt >> (FROWS * q)) & ((1u << FROWS) - 1u);
                float* const F = &L.ftile[(FROWS * q) * FS];
                const int lane = fresh_lane();
                if (m4 != 0) {
                    if constexpr (GEO) {
                        // sixteen lanes: (row, quadrant) - the quadrant's moments re-centred on the splat mean (dx = ax - u), then summed
                        const int r = (lane >> 2) & 3, qd = lane & 3;
                        float* row = F + r * PL_FS;
                        const float4 i0 = *reinterpret_cast<const float4*>(row + 60);
                        const float2 i1 = *reinterpret_cast<const float2*>(row + 64);            // conic c', opacity
                        const float ax = i0.x - (float)(tx0 + (qd & 1) * 8) - 3.5f, ay = i0.y - (float)(ty0 + (qd >> 1) * 8) - 3.5f;
                        const float* mq = row + 36 + 6 * qd;
                        const float N0 = mq[0], N1x = mq[1], N1y = mq[2], N2xx = mq[3], N2xy = mq[4], N2yy = mq[5];
                        float M0 = N0;
                        float m1 = fmaf(ax, N0, -N1x), m2 = fmaf(ay, N0, -N1y);                      // sum s dx, sum s dy
                        float sxx = fmaf(ax, fmaf(ax, N0, -2.f * N1x), N2xx);                        // sum s dx^2
                        float sxy = fmaf(ax, fmaf(ay, N0, -N1y), fmaf(-ay, N1x, N2xy));              // sum s dx dy
                        float syy = fmaf(ay, fmaf(ay, N0, -2.f * N1y), N2yy);                        // sum s dy^2
                        M0 += dpp_get<0xB1>(M0); m1 += dpp_get<0xB1>(m1); m2 += dpp_get<0xB1>(m2);
                        sxx += dpp_get<0xB1>(sxx); sxy += dpp_get<0xB1>(sxy); syy += dpp_get<0xB1>(syy);
                        M0 += dpp_get<0x4E>(M0); m1 += dpp_get<0x4E>(m1); m2 += dpp_get<0x4E>(m2);
                        sxx += dpp_get<0x4E>(sxx); sxy += dpp_get<0x4E>(sxy); syy += dpp_get<0x4E>(syy);
                        const float c0 = row[32], c1 = row[33], c2 = row[34], c3 = row[35];
                        __builtin_amdgcn_wave_barrier();
                        if (lane < 16 && qd == 0) {
                            const float ca = i0.z * CONIC_UNSCALE_AC, cb = i0.w * CONIC_UNSCALE_B, cc = i1.x * CONIC_UNSCALE_AC;
                            // dG/ddelx = -G (a dx + b dy),  dG/da = -G dx^2 / 2, ...;  dL/dopacity = sum G dL/dalpha = M0 / op
                            float4 o0, o1;
                            o0.x = neg_half_w * fmaf(ca, m1, cb * m2);
                            o0.y = neg_half_h * fmaf(cc, m2, cb * m1);
                            o0.z = -0.5f * sxx; o0.w = -0.5f * sxy;
                            o1.x = -0.5f * syy;
                            o1.y = M0 * __builtin_amdgcn_rcpf(fmaxf(i1.y, 1e-30f));
                            o1.z = c0; o1.w = c1;
                            *reinterpret_cast<float4*>(row + 32) = o0;
                            *reinterpret_cast<float4*>(row + 36) = o1;
                            *reinterpret_cast<float2*>(row + 40) = make_float2(c2, c3);
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (!PL_DEV_SKIP(1)) {
                            // features: two rows per instruction, 32 channels each; gradient record: four rows per instruction
                            const int rr = lane >> 5, ch = lane & 31;
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                const int rw = 2 * h + rr;
                                float v = F[rw * PL_FS + ch];
                                if (split16) v += F[rw * PL_FS + 16 + (ch & 15)];       // the other pair of quadrants
                                const uint32_t gg = __float_as_uint(F[rw * PL_FS + GID_SLOT]);
                                if (((m4 >> rw) & 1u) && ch < a.nc) unsafeAtomicAdd(a.dL_dfeature + (size_t)gg * a.C + a.c0 + ch, v);
                            }
                            const int rw = lane >> 4, k16 = lane & 15;
                            const float v = F[rw * PL_FS + 32 + min(k16, 9)];
                            const uint32_t gg = __float_as_uint(F[rw * PL_FS + GID_SLOT]);
                            if (((m4 >> rw) & 1u) && k16 < 10) unsafeAtomicAdd(a.grec + (size_t)gg * GREC + k16, v);
                        }
                    } else if (!PL_DEV_SKIP(1)) {
#pragma unroll
                        for (int rw = 0; rw < FROWS; rw++) {
                            const uint32_t gg = __float_as_uint(F[rw * FS + GID_SLOT]);
#pragma unroll
                            for (int hf = 0; hf < NWV / 4; hf++) {          // 64 channels per instruction
                                const int ch = lane + 64 * hf;
                                float v = F[rw * FS + ch];
                                if (split32) v += F[rw * FS + 32 + (lane & 31)];     // the other pair of quadrants
                                if (((m4 >> rw) & 1u) && ch < a.nc) unsafeAtomicAdd(a.dL_dfeature + (size_t)gg * a.C + a.c0 + ch, v);
                            }
                        }
                    }
                }
            }
            if (tt != 0 && q == 0 && lane == 0) L.touched[parity] = 0;     // next used two chunks from now, behind two barriers
            PL_PHASE_END(4);
        }
    }
#ifdef F3DGS_DEV
    if ((a.dev & 8) && lane == 0) {
        cyc_[7] = 1;
        for (int k = 0; k < 8; k++) a.dev_cycles[((size_t)blockIdx.x * 4 + q) * 8 + k] = cyc_[k];     // private slots: summed on the host
    }
#endif
}

template <bool GEO, int P1_NE, bool P1_PREF, bool M44 = false, bool BF = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) render_backward_pl_kernel(BwdArgs a) {
    if constexpr (GEO && BF) {       // (BwdArgs::gate: one of two launches of a captured frame runs)
        if (a.gate && ((*a.gate != 0u) != (a.gate_want != 0))) return;
    }
    render_backward_pl_body<GEO, P1_NE, P1_PREF, M44, BF>(a);
}
// first window, hybrid shape (see S32 in the body)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) render_backward_pl_kernel_hyb(BwdArgs a) {
    if (a.gate && ((*a.gate != 0u) != (a.gate_want != 0))) return;
    render_backward_pl_body<true, 1, true, false, true, 4, true>(a);
}
// later windows of more than 64 channels, bf16 shape: eight waves per tile (see NWV in the body)
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) render_backward_pl_kernel8(BwdArgs a) {
    render_backward_pl_body<false, 1, true, false, true, 8>(a);
}

// Tiles by descending walk length, XCD by XCD: workgroup b runs on XCD b % 8 (xcd_remap) and every XCD keeps its contiguous
// run of tiles (neighbouring tiles share splat records and feature rows behind one L2) but takes it longest walk first.
// One workgroup per XCD: counting sort of its run on min(len, 4095) / 4 (1024 buckets); the k-th tile of XCD x goes to
// order[8 k + x].  The order inside a bucket is whatever the atomics make of it - only the launch order depends on it.
// (band_b0 / band_tb: the runs are runs of band_perm's VIRTUAL tile ids - a listed band is spread over all eight XCDs)
__global__ void __launch_bounds__(1024) tile_order_kernel(const uint32_t* __restrict__ tile_len, uint32_t tiles_all,
                                                          uint32_t* __restrict__ order, uint32_t band_b0, uint32_t band_tb) {
    __shared__ uint32_t bucket[1024];
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t x = blockIdx.x, q = tiles_all / 8, r = tiles_all % 8;
    const uint32_t first = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;      // as xcd_remap
    const uint32_t tiles = x < r ? q + 1 : q;
    auto real = [&](uint32_t t) { return band_perm(first + t, tiles_all, band_b0, band_tb); };
    bucket[tid] = 0;
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += 1024) atomicAdd(&bucket[1023u - (min(tile_len[real(t)], 4095u) >> 2)], 1u);
    __syncthreads();
    // exclusive scan of the 1024 bucket counts (bucket 0 = longest walks)
    const uint32_t c = bucket[tid];
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, d, 64);
        if ((int)lane >= d) inc += v;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < w; k++) base += wsum[k];
    __syncthreads();
    bucket[tid] = base + inc - c;
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += 1024) {
        const uint32_t rt = real(t);
        order[8u * atomicAdd(&bucket[1023u - (min(tile_len[rt], 4095u) >> 2)], 1u) + x] = rt;
    }
}

template <bool GEO>
void launch_pl(const BwdArgs& a, hipStream_t s) {
#ifdef F3DGS_DEV
    if (a.dev & 8) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_backward_pl_kernel<GEO, 1, true>, 256, sizeof(PlShared));
        fprintf(stderr, "[f3dgs dev] pixel-lane backward: %d workgroups per CU by the occupancy query, %zu bytes of LDS\n", nb, sizeof(PlShared));
    }
#endif
#ifdef F3DGS_DEV
    const int variant = (a.dev >> 8) & 7;
    if (variant == 4) { hipLaunchKernelGGL((render_backward_pl_kernel<GEO, 1, false>), dim3(a.gx * a.gy), dim3(256), sizeof(PlShared), s, a); return; }
    if (variant == 5) { hipLaunchKernelGGL((render_backward_pl_kernel<GEO, 2, true>), dim3(a.gx * a.gy), dim3(256), sizeof(PlShared), s, a); return; }
#endif
    if constexpr (GEO) {
        if (a.bf16 == 3) {       // a captured frame: the device's long-axis word picks the shape (both launched, one runs)
            BwdArgs b = a;
            b.bf16 = 1; b.gate_want = 0;
            hipLaunchKernelGGL((render_backward_pl_kernel<GEO, 1, true, false, true>), dim3(a.gx * a.gy), dim3(256), sizeof(PlShared), s, b);
            b.bf16 = 2; b.gate_want = 1;
            hipLaunchKernelGGL(render_backward_pl_kernel_hyb, dim3(a.gx * a.gy), dim3(256), sizeof(PlShared), s, b);
            return;
        }
        if (a.bf16 == 2) { hipLaunchKernelGGL(render_backward_pl_kernel_hyb, dim3(a.gx * a.gy), dim3(256), sizeof(PlShared), s, a); return; }
    }
    if (a.bf16) { hipLaunchKernelGGL((render_backward_pl_kernel<GEO, 1, true, false, true>), dim3(a.gx * a.gy), dim3(256), sizeof(PlShared), s, a); return; }
    if constexpr (GEO) {
        if (a.m44) { hipLaunchKernelGGL((render_backward_pl_kernel<GEO, 1, true, true>), dim3(a.gx * a.gy), dim3(256), sizeof(PlShared), s, a); return; }
    }
    hipLaunchKernelGGL((render_backward_pl_kernel<GEO, 1, true>), dim3(a.gx * a.gy), dim3(256), sizeof(PlShared), s, a);
}

}  // namespace

void launch_tile_order(const uint32_t* tile_len, size_t tiles, uint32_t* order, uint32_t band_b0, uint32_t band_tb, hipStream_t s) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(1024), 0, s, tile_len, (uint32_t)tiles, order, band_b0, band_tb);
}

// Channel windows: the first carries the geometric sums and up to 32 channels, later ones up to 64 channels.
void launch_render_backward_pl(BwdArgs a, int C, hipStream_t s) {
    a.neg_half_w = -(0.5f * a.W); a.neg_half_h = -(0.5f * a.H);
    // (a 16-channel first window in the quadrant-split shape, where that does not add a window, was measured: c4 2.675 -> 2.668 ms,
    // c5 4.65 -> 4.59 - the later window grows from 32 to 48 channels and takes a second staging round)
    a.c0 = 0; a.nc = min(32, C); a.write_base = 1;
    launch_pl<true>(a, s);
    // Later windows carry feature sums only (weights x gradient planes): nothing behind them amplifies an error of the sums, so
    // with option bwd_bf16 = -1 they contract on bf16 instructions whatever the frame's conditioning made of the FIRST window
    // (measured: every blend-level tensor within 0.3 of the gradient bound of the exact contraction at any axis ratio,
    // profiles/r06_ratio_sweep.txt); bwd_bf16 = 0 keeps them exact.
    if (options().bwd_bf16 < 0 || a.bf16 >= 2) a.bf16 = 1;
    a.gate = nullptr;
    // later windows: 64 channels on four waves; with the bf16 shape up to 128 channels on eight waves where more than 64 remain
    // (option bwd_wide8, default 1) - every window re-evaluates the blend weights of the whole list
    const bool wide8 = a.bf16 && options().bwd_wide8 != 0;
    for (int c0 = 32; c0 < C;) {
        const int left = C - c0;
        a.c0 = c0; a.write_base = 0;
        if (wide8 && left > 64) {
            a.nc = min(128, left);
            hipLaunchKernelGGL(render_backward_pl_kernel8, dim3(a.gx * a.gy), dim3(512), sizeof(PlShared8), s, a);
        } else {
            a.nc = min(64, left);
            launch_pl<false>(a, s);
        }
        c0 += a.nc;
    }
}

}  // namespace f3dgs
