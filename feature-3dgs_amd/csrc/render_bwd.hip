// render_bwd.hip — reverse-order (back-to-front) backward of the blend.
//
// Semantics: R/cuda_rasterizer/backward.cu:407-620 (R = submodules/diff-gaussian-rasterization-feature),
// with quirks Q1 (no gating by the 0.99 clamp), Q2 (feature loss never reaches alpha), Q3 (the
// reference's dead collected_semantic_feature buffer is not reproduced), Q5 and Q8.
//
// The reference keeps lane = pixel and issues 10 + C global fp32 atomics per (pixel, Gaussian) pair,
// every lane of a block on the same address.  This kernel TRANSPOSES the problem for wave64:
//
//   * one wave owns NPIX pixels of a tile and walks the tile's list back to front in chunks of 64
//     instances; lane l HOLDS instance hi-1-l of the chunk in registers (no LDS staging at all);
//   * the wave loops over its pixels; everything that belongs to the pixel (coordinates, upstream
//     gradients, running transmittance, running "colour behind") is wave-uniform and comes from LDS as a
//     broadcast; the 10 + C per-Gaussian gradient sums are LANE-PRIVATE accumulators, so the pixel
//     reduction that dominates the reference costs nothing;
//   * the sequential dependence along the list turns into two DPP prefix scans per pixel and chunk
//     (gfx9 row_shr / row_bcast forms): a product scan for the transmittance, T_g = T_in * prod 1/(1-a),
//     and a sum scan for S_g = sum_{j behind g} w_j (c_j . dL/dpix + depth_j dL/ddepth), with
//       dL/dalpha_g = T_g q_g - (S_g + T_final bg.dL/dpix) / (1 - alpha_g)
//     which is the reference's (c - accum_rec) T recurrence summed over channels first;
//   * pixels whose list ended before the chunk (early termination) are skipped through a scalar bit
//     mask, so the pixel loop visits exactly sum_px ceil(n_contrib/64) bodies;
//   * per chunk the accumulators are transposed through LDS and flushed with coalesced vector atomics:
//     one contiguous C-float run of dL_dsemantic_feature and one 48-byte gradient record per Gaussian.
//
// Numerics: T is rebuilt with reciprocals instead of the reference's chain of divisions and the colour
// recurrence is evaluated in closed form; both are algebraically identical and agree to fp32 round-off
// (the gradient tolerance of the path is 1e-3 relative).

#include <cstdio>
#include <vector>

#include "render_common.h"

namespace f3dgs {

namespace {


// ---- paired DPP prefix scans --------------------------------------------------------------------------
// Two independent inclusive scans over the 64 lanes run interleaved so that each DPP instruction's
// 2-wait-state read-after-write hazard is covered by the partner's instruction plus one s_nop.  The
// VOP2-DPP forms are fused (v = op(dpp(v), v)); lanes whose DPP source is out of range are disabled
// (bound_ctrl:0), i.e. keep their value = combine with the identity.  Measured on MI355X: 4.5 cycles per
// fused step vs 9.0 for the v_mov_b32_dpp + v_mul pair the compiler emits for update_dpp with identity 1.0.
#define F3DGS_SCAN2(OP)                                                                          \
    asm volatile("s_nop 1\n\t"                                                                   \
                 OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                       \
                 OP " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                       \
                 "s_nop 0\n\t"                                                                   \
                 OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                       \
                 OP " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                       \
                 "s_nop 0\n\t"                                                                   \
                 OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                       \
                 OP " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                       \
                 "s_nop 0\n\t"                                                                   \
                 OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                       \
                 OP " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                       \
                 "s_nop 0\n\t"                                                                   \
                 OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                    \
                 OP " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                    \
                 "s_nop 0\n\t"                                                                   \
                 OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                    \
                 OP " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                    \
                 "s_nop 1"                                                                       \
                 : "+v"(a), "+v"(b))
__device__ __forceinline__ void wave_incl_prod2(float& a, float& b) { F3DGS_SCAN2("v_mul_f32_dpp"); }
__device__ __forceinline__ void wave_incl_sum2(float& a, float& b) { F3DGS_SCAN2("v_add_f32_dpp"); }
#undef F3DGS_SCAN2
// Four interleaved chains: three partner instructions cover the hazard, no s_nop needed between steps.
#define F3DGS_STEP4(OP, CTRL)                                   \
    OP " %0, %0, %0 " CTRL " bank_mask:0xf\n\t"                 \
    OP " %1, %1, %1 " CTRL " bank_mask:0xf\n\t"                 \
    OP " %2, %2, %2 " CTRL " bank_mask:0xf\n\t"                 \
    OP " %3, %3, %3 " CTRL " bank_mask:0xf\n\t"
#define F3DGS_SCAN4(OP)                                                                                     \
    asm volatile("s_nop 1\n\t" F3DGS_STEP4(OP, "row_shr:1 row_mask:0xf") F3DGS_STEP4(OP, "row_shr:2 row_mask:0xf")  \
                 F3DGS_STEP4(OP, "row_shr:4 row_mask:0xf") F3DGS_STEP4(OP, "row_shr:8 row_mask:0xf")            \
                 F3DGS_STEP4(OP, "row_bcast:15 row_mask:0xa") F3DGS_STEP4(OP, "row_bcast:31 row_mask:0xc") "s_nop 1" \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
__device__ __forceinline__ void wave_incl_prod4(float& a, float& b, float& c, float& d) { F3DGS_SCAN4("v_mul_f32_dpp"); }
__device__ __forceinline__ void wave_incl_sum4(float& a, float& b, float& c, float& d) { F3DGS_SCAN4("v_add_f32_dpp"); }
#undef F3DGS_SCAN4
// Sums, out of place: the first step reads the inputs (an out-of-range DPP source reads as 0 with bound_ctrl, so
// the lane still gets its own value) and writes fresh registers - the callers need the inputs afterwards, and this
// saves the four copies an in-place scan would start with.
#define F3DGS_STEP4_FIRST(OP, CTRL)                             \
    OP " %0, %4, %4 " CTRL " bank_mask:0xf bound_ctrl:0\n\t"    \
    OP " %1, %5, %5 " CTRL " bank_mask:0xf bound_ctrl:0\n\t"    \
    OP " %2, %6, %6 " CTRL " bank_mask:0xf bound_ctrl:0\n\t"    \
    OP " %3, %7, %7 " CTRL " bank_mask:0xf bound_ctrl:0\n\t"
__device__ __forceinline__ void wave_incl_sum4_to(float (&o)[4], float a, float b, float c, float d) {
    asm volatile("s_nop 1\n\t" F3DGS_STEP4_FIRST("v_add_f32_dpp", "row_shr:1 row_mask:0xf")
                 F3DGS_STEP4("v_add_f32_dpp", "row_shr:2 row_mask:0xf") F3DGS_STEP4("v_add_f32_dpp", "row_shr:4 row_mask:0xf")
                 F3DGS_STEP4("v_add_f32_dpp", "row_shr:8 row_mask:0xf") F3DGS_STEP4("v_add_f32_dpp", "row_bcast:15 row_mask:0xa")
                 F3DGS_STEP4("v_add_f32_dpp", "row_bcast:31 row_mask:0xc") "s_nop 1"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                 : "v"(a), "v"(b), "v"(c), "v"(d));
}
// The same two scans over each HALF of the wave separately (lanes 0-31 and 32-63 hold the same 32 instances against
// two different pixels): the step across the half boundary (row_bcast:31) is simply not taken.
__device__ __forceinline__ void half_incl_prod4(float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 1\n\t" F3DGS_STEP4("v_mul_f32_dpp", "row_shr:1 row_mask:0xf") F3DGS_STEP4("v_mul_f32_dpp", "row_shr:2 row_mask:0xf")
                 F3DGS_STEP4("v_mul_f32_dpp", "row_shr:4 row_mask:0xf") F3DGS_STEP4("v_mul_f32_dpp", "row_shr:8 row_mask:0xf")
                 F3DGS_STEP4("v_mul_f32_dpp", "row_bcast:15 row_mask:0xa") "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void half_incl_sum4_to(float (&o)[4], float a, float b, float c, float d) {
    asm volatile("s_nop 1\n\t" F3DGS_STEP4_FIRST("v_add_f32_dpp", "row_shr:1 row_mask:0xf")
                 F3DGS_STEP4("v_add_f32_dpp", "row_shr:2 row_mask:0xf") F3DGS_STEP4("v_add_f32_dpp", "row_shr:4 row_mask:0xf")
                 F3DGS_STEP4("v_add_f32_dpp", "row_shr:8 row_mask:0xf") F3DGS_STEP4("v_add_f32_dpp", "row_bcast:15 row_mask:0xa")
                 "s_nop 1"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                 : "v"(a), "v"(b), "v"(c), "v"(d));
}
#undef F3DGS_STEP4_FIRST
#undef F3DGS_STEP4

// LDS image of one wave.  Per-pixel data is wave-uniform in the bodies and is fetched with broadcast
// ds_reads (the LDS pipe is idle; v_readlane costs ~8 VALU cycles apiece on gfx950):
//   pa[p] = {pixel x, pixel y, T_in, S_in}  (T/S are the running state, rewritten by lane 63 per chunk)
//   pb[p] = {dL/dR, dL/dG, dL/dB, dL/ddepth}     plast[p] = n_contrib
constexpr int FLUSH_GROUP = 16;               // accumulators transposed per flush round
constexpr int FLUSH_STRIDE = FLUSH_GROUP + 1; // odd: conflict-free lane-major writes and column reads
// MF = true: the feature-gradient contraction dF[g][c] += sum_px w[px][g] dO[px][c] runs on the matrix
// pipe (exact-fp32 v_mfma_f32_32x32x2_f32) concurrently with the VALU work; dO is then kept row-major per
// pixel with an odd row stride.  MF = false keeps it as float4 [c/4][pixel] for the VALU FMAs.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH, int NPIX, bool MF, bool HALF = false>
struct BwdLds {
    // Row stride (floats) of the MF image: unpadded, columns XOR-swizzled by the pixel index (staging writes and
    // matrix-pipe operand reads are both conflict-free).  Together with
    // the ids riding in the padding column of the flush tile and `touched` as a ballot this makes the NPIX = 64, CH = 32 image 13312 B, i.e. 12
    // waves (3 per SIMD) per CU instead of 11.
    static constexpr int GS = CH;
    float4 pa[NPIX];
    float4 pb[NPIX];
    uint32_t plast[NPIX];
    float4 gf[MF ? 1 : (CH > 0 ? CH / 4 : 1)][MF ? 1 : NPIX];
    float gfm[MF ? NPIX * GS : 4];
    static constexpr int FS = MF ? 11 : FLUSH_STRIDE;   // MF: only the 10 geometric sums travel through LDS
    float flush[(HALF ? 32 : 64) * FS];        // one row per instance of a chunk
};

#ifdef F3DGS_DEV
#define F3DGS_DEV_SKIP(bit) (a.dev & (bit))
// phase timing (dev builds, option dev bit 3): per-wave s_memtime deltas summed into a.dev_cycles[phase]
#define F3DGS_PHASE_BEGIN() unsigned long long ph_t0_ = (a.dev & 8) ? __builtin_readcyclecounter() : 0ull
#define F3DGS_PHASE_END(ACC) do { if (a.dev & 8) { const unsigned long long t1_ = __builtin_readcyclecounter(); ACC += t1_ - ph_t0_; ph_t0_ = t1_; } } while (0)
#else
#define F3DGS_DEV_SKIP(bit) false      // release builds: compiled out
#define F3DGS_PHASE_BEGIN() do {} while (0)
#define F3DGS_PHASE_END(ACC) do {} while (0)
#endif

struct SplatLane { // one chunk entry per lane
    float mx, my, ca, cb, cc, op, cr, cg, cbl, dep;
    uint32_t pos;
    bool have;
};

// GEO = false: a later channel window of a wide feature (C > 64 runs in windows of 64): only dL/dfeature of the window
// is produced - the colour/depth dot product, the sum scan, dL/dalpha and the ten geometric sums belong to the first
// window's launch and are not computed again (about half of a body's instructions).
// HALF = true: a chunk holds 32 instances and the two halves of the wave run them against two different pixels
// (lane l = instance l & 31, pixel half l >> 5; eight pixels per trip).  A body costs the same per (pixel, instance)
// pair, but the front-most, partially filled chunk of a wave wastes at most 31 lanes x its pixels instead of 63, a
// chunk's pixels are cut off at a finer depth, and the scans and the matrix-pipe operands need no step across the
// half boundary (`tools/pair_stats.py`: 12.6 % fewer body-equivalents at c3).
template <int CH, int NPIX, bool MF, int U, bool GEO, bool HALF>
__device__ __forceinline__ void render_backward_body(const BwdArgs& a) {
    static_assert(!HALF || (NPIX == 64 && U == 4), "half-wave chunks: 64-pixel blocks, four pixels per half and trip");
    constexpr int CAP = HALF ? 32 : 64;        // instances per chunk
    constexpr int MH = HALF ? 1 : 2;           // 32-row accumulator blocks per chunk on the matrix pipe
    constexpr int NB = MF ? CH / 32 : 0;       // 32-channel column blocks on the matrix pipe
    constexpr int CHV = CH / 4;
    constexpr int PARTS = 256 / NPIX;          // waves (workgroups) per tile
    constexpr int NV = NPIX / 64;              // pixel-state registers per lane
    using Lds = BwdLds<CH, NPIX, MF, HALF>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds& L = *reinterpret_cast<Lds*>(smem);
    const int lane = threadIdx.x;
#ifdef F3DGS_DEV
    unsigned long long cyc_stage = 0, cyc_walk = 0, cyc_trip = 0, cyc_flush = 0;
#endif
    F3DGS_PHASE_BEGIN();

    // the PARTS waves of one tile are scheduled back to back (L2 reuse of the tile's splat records)
    // a.order (option bwd_order): every XCD takes ITS contiguous run of tiles longest walk first (tile_order_kernel: the k-th
    // tile of XCD x sits at order[8 k + x]); workgroup b runs on XCD b % 8, the j-th workgroup of an XCD is part j % 4 of its
    // tile j / 4.  The grid is padded to 8 x 4 x ceil(tiles / 8): slots past an XCD's run have nothing to do.
    uint32_t tile;
    int part;
    if (a.order) {
        const uint32_t tiles = (uint32_t)(a.gx * a.gy);
        const uint32_t xcd = blockIdx.x % 8u, j = blockIdx.x / 8u, k = j / PARTS;
        const uint32_t mine = tiles / 8u + (xcd < tiles % 8u ? 1u : 0u);
        if (k >= mine) return;
        tile = a.order[8u * k + xcd];
        part = (int)(j % PARTS);
    } else {
        const uint32_t wg = xcd_remap(blockIdx.x, gridDim.x);
        tile = band_perm(wg / PARTS, (uint32_t)(a.gx * a.gy), a.band_b0, a.band_tb);
        part = (int)(wg % PARTS);
    }
    const int tx = tile % a.gx, ty = tile / a.gx;
    const uint2 rg = a.ranges[tile];
    if (rg.x >= rg.y) return;     // an empty list (every tile outside a listed band): nothing to stage, nothing to add
    const uint32_t r_lo = __builtin_amdgcn_readfirstlane((int)rg.x);
    const size_t HW = (size_t)a.W * a.H;
    // pixel block of this wave: one 8x8 quadrant of the tile
    static_assert(NPIX == 64, "one quadrant per wave");
    const int qd = part;
    const int px0 = tx * TILE + (qd & 1) * 8;
    const int py0 = ty * TILE + (qd >> 1) * 8;
    constexpr int PW = 8;    // pixels per row of this wave's block

    // ---- stage the per-pixel data into LDS (lane = pixel here) -----------------------------------------
    // Order of the requests (vector memory returns in order, so what is needed first is asked for first):
    //   n_contrib  ->  max_last  ->  list ids of the first two windows  ->  the 5 + CH per-pixel planes  ->  (ids arrive)
    //   splat records of the first window  ->  (planes arrive) LDS image  ->  the walk starts with its records in flight
    uint32_t v_last[NV];
    uint32_t max_last = 0;
#pragma unroll
    for (int it = 0; it < NV; it++) {
        const int p = it * 64 + lane;
        const int x = px0 + p % PW, y = py0 + p / PW;
        const bool inside = p < NPIX && x < a.W && y < a.H;
        v_last[it] = inside ? a.n_contrib[(size_t)y * a.W + x] : 0u;
        max_last = max(max_last, v_last[it]);
    }
    max_last = wave_max_u32(max_last);
    auto load_ids = [&](int k0w) -> uint32_t {
        const uint32_t pos = (uint32_t)(k0w + 63 - lane);
        return (k0w >= 0 && pos < max_last) ? a.point_list[r_lo + pos] : 0u;
    };
    const int k_top = (int)((max_last + 63) / 64) * 64 - 64;
    uint32_t ngid = load_ids(k_top), fgid = load_ids(k_top - 64);
#pragma unroll
    for (int it = 0; it < NV; it++) {
        const int p = it * 64 + lane;
        const int x = px0 + p % PW, y = py0 + p / PW;
        const bool inside = p < NPIX && x < a.W && y < a.H;
        const size_t pid = (size_t)y * a.W + x;
        // every lane reads a valid address (pixel 0 stands in outside the image) and the values are masked afterwards:
        // no branch between the loads, so all 6 + CH of them are in flight together - one memory latency per wave
        // instead of one per plane (measured with per-phase cycle counters: staging was 28 % of a wave's lifetime)
        const size_t pid_s = inside ? pid : 0;
        float4 g = make_float4(a.dL_dpix[pid_s], a.dL_dpix[HW + pid_s], a.dL_dpix[2 * HW + pid_s], a.dL_ddepth[pid_s]);
        float Tf = a.final_T[pid_s];
        float fv[CH > 0 ? CH : 1];
        if constexpr (CH > 0) {
            const int ncm1 = a.nc - 1;
#pragma unroll
            for (int c = 0; c < CH; c++) fv[c] = a.dL_dfeat[(size_t)(a.c0 + min(c, ncm1)) * HW + pid_s];   // min: wave-uniform
        }
        if (!inside) {
            g = make_float4(0.f, 0.f, 0.f, 0.f);
            Tf = 0.f;
        }
        if (p < NPIX) {
            L.pa[p] = make_float4((float)x, (float)y, Tf, Tf * (a.bg[0] * g.x + a.bg[1] * g.y + a.bg[2] * g.z));
            L.pb[p] = g;
            L.plast[p] = v_last[it];
        }
        if (CH > 0 && p < NPIX) {
#pragma unroll
            for (int v = 0; v < CHV; v++) {
                float4 f;
                f.x = (inside && 4 * v + 0 < a.nc) ? fv[4 * v + 0] : 0.f;
                f.y = (inside && 4 * v + 1 < a.nc) ? fv[4 * v + 1] : 0.f;
                f.z = (inside && 4 * v + 2 < a.nc) ? fv[4 * v + 2] : 0.f;
                f.w = (inside && 4 * v + 3 < a.nc) ? fv[4 * v + 3] : 0.f;
                if constexpr (MF) {
                    // XOR-swizzled columns (c ^ (p & 31)): with lane = pixel the 64 lanes of a store hit 32 different banks
                    // (unswizzled they all hit bank c: a 64-way conflict on each of the 32 stores, ~2000 LDS cycles per
                    // wave), and the matrix-pipe reads below (lane = channel, one pixel per half-wave) stay a permutation
                    // of one 32-float row.  No padding: the image stays 13312 B = 12 waves per CU.
                    const int sw = p & 31, row = p * Lds::GS + (4 * v & ~31);
                    L.gfm[row + (((4 * v + 0) & 31) ^ sw)] = f.x; L.gfm[row + (((4 * v + 1) & 31) ^ sw)] = f.y;
                    L.gfm[row + (((4 * v + 2) & 31) ^ sw)] = f.z; L.gfm[row + (((4 * v + 3) & 31) ^ sw)] = f.w;
                } else {
                    L.gf[v][p] = f;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    F3DGS_PHASE_END(cyc_stage);
    // (the pixel scales 0.5 W, 0.5 H of dL/dmean2D are formed where the flush uses them, see sgpr_opaque)

    // ---- one (compacted) chunk of up to 64 splats against all live pixels of the wave ---------------------
    auto process = [&](const SplatLane& sl_in, const uint32_t gid_in, const uint32_t pos_min, const int n_inst) {
        F3DGS_PHASE_END(cyc_walk);
        SplatLane sl = sl_in;
        uint32_t gid = gid_in;
        if constexpr (HALF) {      // both halves of the wave hold the chunk's 32 instances
            const int src = lane & 31;
            sl.mx = __shfl(sl_in.mx, src); sl.my = __shfl(sl_in.my, src); sl.ca = __shfl(sl_in.ca, src); sl.cb = __shfl(sl_in.cb, src);
            sl.cc = __shfl(sl_in.cc, src); sl.op = __shfl(sl_in.op, src); sl.cr = __shfl(sl_in.cr, src); sl.cg = __shfl(sl_in.cg, src);
            sl.cbl = __shfl(sl_in.cbl, src); sl.dep = __shfl(sl_in.dep, src);
            sl.pos = (uint32_t)__shfl((int)sl_in.pos, src);
            gid = (uint32_t)__shfl((int)gid_in, src);
            sl.have = src < n_inst;
        }
        float acc[10];
#pragma unroll
        for (int k = 0; k < 10; k++) acc[k] = 0.f;
        float fac[(CH > 0 && !MF) ? CH : 1];
#pragma unroll
        for (int c = 0; c < ((CH > 0 && !MF) ? CH : 1); c++) fac[c] = 0.f;
        f32x16 macc[NB > 0 ? NB : 1][MH];          // [column block][instances 0-31 | 32-63]
#pragma unroll
        for (int nb = 0; nb < (NB > 0 ? NB : 1); nb++)
#pragma unroll
            for (int hh = 0; hh < MH; hh++)
#pragma unroll
                for (int r = 0; r < 16; r++) macc[nb][hh][r] = 0.f;
        bool touched = false;

#pragma unroll
        for (int it = 0; it < NV; it++) {
            // pixels still alive at this depth (lane = pixel for the ballot); two bodies per trip: ILP for the
            // scans, and the K = 2 of the MFMA
            unsigned long long live = __ballot(v_last[it] > pos_min);
            if (F3DGS_DEV_SKIP(2)) { touched = sl.have; live = 0; }
            while (live) {
                // take up to U live pixels; missing ones repeat the first with n_contrib = 0 (inert bodies)
                int qs[U];
                bool acts[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    acts[u] = live != 0;
                    const int b = acts[u] ? __builtin_ctzll(live) : 0;
                    live &= live - 1;   // (0 & anything) stays 0
                    qs[u] = acts[u] ? it * 64 + b : qs[0];
                }
                // qi / act: this LANE's pixel of body u - wave-uniform, or (HALF) the upper half of the wave takes its
                // own U pixels from the top of the live mask
                int qi[U];
                bool act[U];
                if constexpr (HALF) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const bool ah = live != 0;
                        const int b = ah ? 63 - __builtin_clzll(live) : 0;
                        live &= ~((ah ? 1ull : 0ull) << b);
                        const int qh = ah ? b : qs[0];
                        qi[u] = lane < 32 ? qs[u] : qh;
                        act[u] = lane < 32 ? acts[u] : ah;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; u++) { qi[u] = qs[u]; act[u] = acts[u]; }
                }
                float4 pa[U], pb[U];
                uint32_t lastq[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    pa[u] = L.pa[qi[u]];
                    pb[u] = L.pb[qi[u]];
                    lastq[u] = L.plast[qi[u]];   // unconditional: keeps all broadcasts of the body in flight together
                }
                constexpr int NBV = HALF ? U : U / 2;       // B operands per trip (one per matrix-pipe step of K = 2 pixels)
                float Bv[MF ? NBV : 1][NB > 0 ? NB : 1];
                if constexpr (MF) {
                    // B[k][j] = dO[pixel k][channel j] of the matrix-pipe contraction below; requested here so that the
                    // LDS latency is covered by the alpha evaluation
#pragma unroll
                    for (int b = 0; b < NBV; b++) {
                        const int pq = HALF ? qi[b] : (lane < 32 ? qi[2 * b] : qi[2 * b + 1]);
                        const int prow = pq * Lds::GS + ((lane & 31) ^ (pq & 31));        // swizzled column, see the staging
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) Bv[b][nb] = L.gfm[prow + 32 * nb];
                    }
                }
                float dx[U], dy[U], au[U], al[U], f[U], q[U], D[U], P[U], Tb[U], w[U], Sinc[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    dx[u] = sl.mx - pa[u].x; dy[u] = sl.my - pa[u].y;
                    const float power2 = splat_power2(dx[u], dy[u], sl.ca, sl.cb, sl.cc);   // conic pre-scaled at staging
                    const float v = sl.op * __builtin_amdgcn_exp2f(power2);                 // op G, not yet clamped
                    // bitwise, not short-circuit: nothing here may turn into a branch that waits on one broadcast alone
                    // (min(0.99, v) < 1/255  <=>  v < 1/255)
                    ok[u] = (int)sl.have & (int)act[u] & (int)(sl.pos < lastq[u]) & (int)!(power2 > 0.0f) & (int)!(v < ALPHA_MIN);
                    au[u] = ok[u] ? v : 0.f;              // exp2 may be inf where power > 0: selected away, never multiplied
                    al[u] = fminf(ALPHA_MAX, au[u]);
                    f[u] = __builtin_amdgcn_rcpf(1.f - al[u]);   // 1/(1-alpha); exactly 1 for skipped lanes
                    P[u] = f[u];
                    if constexpr (GEO) q[u] = fmaf(sl.cr, pb[u].x, fmaf(sl.cg, pb[u].y, fmaf(sl.cbl, pb[u].z, sl.dep * pb[u].w)));
                    touched = touched || ok[u];
                }
                if constexpr (HALF) half_incl_prod4(P[0], P[1], P[2], P[3]);
                else if constexpr (U == 4) wave_incl_prod4(P[0], P[1], P[2], P[3]);
                else wave_incl_prod2(P[0], P[1]);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    Tb[u] = pa[u].z * P[u];           // transmittance in front of this splat
                    w[u] = al[u] * Tb[u];
                    if constexpr (GEO) D[u] = w[u] * q[u];
                }
                if constexpr (GEO) {
                    if constexpr (HALF) half_incl_sum4_to(Sinc, D[0], D[1], D[2], D[3]);
                    else if constexpr (U == 4) wave_incl_sum4_to(Sinc, D[0], D[1], D[2], D[3]);
                    else { Sinc[0] = D[0]; Sinc[1] = D[1]; wave_incl_sum2(Sinc[0], Sinc[1]); }
                }
                // the last lane (of each half) holds the chunk totals: it carries the pixel state to the next (nearer) chunk
                if ((lane & (CAP - 1)) == CAP - 1) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        if (!act[u]) continue;
                        if constexpr (GEO) *reinterpret_cast<float2*>(&L.pa[qi[u]].z) = make_float2(Tb[u], pa[u].w + Sinc[u]);
                        else L.pa[qi[u]].z = Tb[u];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if constexpr (GEO) {
                        const float Sbehind = pa[u].w + (Sinc[u] - D[u]);
                        // finite on skipped lanes too (f = 1 there), and every use below is multiplied by Gs = 0 or w = 0
                        const float dL_dalpha = fmaf(Tb[u], q[u], -(Sbehind * f[u]));
                        // raw moments of s = op G dL/dalpha over the pixels; the conic, the pixel scale and 1/op (for
                        // dL/dopacity = sum G dL/dalpha) are applied once per chunk, before the flush
                        const float sg = au[u] * dL_dalpha;
                        const float sdx = sg * dx[u], sdy = sg * dy[u];
                        acc[0] += sdx;
                        acc[1] += sdy;
                        acc[2] = fmaf(sdx, dx[u], acc[2]);
                        acc[3] = fmaf(sdx, dy[u], acc[3]);
                        acc[4] = fmaf(sdy, dy[u], acc[4]);
                        acc[5] += sg;
                        acc[6] = fmaf(w[u], pb[u].x, acc[6]);
                        acc[7] = fmaf(w[u], pb[u].y, acc[7]);
                        acc[8] = fmaf(w[u], pb[u].z, acc[8]);
                        acc[9] = fmaf(w[u], pb[u].w, acc[9]);
                    }
                    if constexpr (CH > 0 && !MF) {
#pragma unroll
                        for (int v = 0; v < CH / 4; v++) {
                            const float4 gf = L.gf[v][qi[u]];
                            fac[4 * v + 0] = fmaf(w[u], gf.x, fac[4 * v + 0]);
                            fac[4 * v + 1] = fmaf(w[u], gf.y, fac[4 * v + 1]);
                            fac[4 * v + 2] = fmaf(w[u], gf.z, fac[4 * v + 2]);
                            fac[4 * v + 3] = fmaf(w[u], gf.w, fac[4 * v + 3]);
                        }
                    }
                }
                if constexpr (MF) if (!F3DGS_DEV_SKIP(4)) {
                    // A = W^T block: rows = instances, k = two pixels.  One half-wave swap builds both 32-instance
                    // operands:  X = [w_a lanes 0-31 | w_b lanes 0-31],  Y = [w_a lanes 32-63 | w_b lanes 32-63].
                    if constexpr (HALF) {
                        // the halves already ARE the operand: row = instance l & 31, k = pixel half l >> 5
#pragma unroll
                        for (int u = 0; u < U; u++)
#pragma unroll
                            for (int nb = 0; nb < NB; nb++)
                                macc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u], Bv[u][nb], macc[nb][0], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int u = 0; u < U; u += 2) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(w[u]), __float_as_int(w[u + 1]), false, false);
                            const float X = __int_as_float(sw[0]), Y = __int_as_float(sw[1]);
#pragma unroll
                            for (int nb = 0; nb < NB; nb++) {
                                macc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(X, Bv[u / 2][nb], macc[nb][0], 0, 0, 0);
                                macc[nb][MH - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Y, Bv[u / 2][nb], macc[nb][MH - 1], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }

        // ---- flush this chunk: transpose through LDS in groups of 16 values, coalesced atomics -----------
        unsigned long long tmask = __ballot(touched);
        if constexpr (HALF) tmask = (tmask | (tmask >> 32)) & 0xFFFFFFFFull;      // an instance is touched in either half
        F3DGS_PHASE_END(cyc_trip);
        if (tmask == 0 || F3DGS_DEV_SKIP(1)) return;
        const int frow = lane & (CAP - 1);          // this lane's instance row in the flush tile
        L.flush[frow * Lds::FS + Lds::FS - 1] = __uint_as_float(gid);   // the stride's padding column carries the ids
        if constexpr (HALF) {
            // the two halves hold partial sums over their own pixels: add them up (both halves end with the total)
            if constexpr (GEO) {
#pragma unroll
                for (int k = 0; k < 10; k++) acc[k] += __shfl_xor(acc[k], 32);
            }
#pragma unroll
            for (int c = 0; c < ((CH > 0 && !MF) ? CH : 0); c++) fac[c] += __shfl_xor(fac[c], 32);
        }
        if constexpr (GEO) {
            // moments -> dL/d(mean2D), dL/d(conic), dL/d(opacity):  dG/ddelx = -G (a dx + b dy),  dG/da = -G dx^2 / 2, ...
            // (sl.ca.. are the staged, pre-scaled conic: undo the scale here, once per chunk)
            const float ca = sl.ca * CONIC_UNSCALE_AC, cb = sl.cb * CONIC_UNSCALE_B, cc = sl.cc * CONIC_UNSCALE_AC;
            const float m1 = acc[0], m2 = acc[1];
            acc[0] = -(0.5f * (float)sgpr_opaque(a.W)) * fmaf(ca, m1, cb * m2);
            acc[1] = -(0.5f * (float)sgpr_opaque(a.H)) * fmaf(cc, m2, cb * m1);
            acc[5] *= __builtin_amdgcn_rcpf(sl.op);      // touched => alpha >= 1/255 somewhere => op > 0
            acc[2] *= -0.5f; acc[3] *= -0.5f; acc[4] *= -0.5f;
        }
        constexpr int CHF = MF ? 0 : CH;            // feature channels that travel through the LDS transpose
        constexpr int NGEO = GEO ? 10 : 0;          // geometric sums ride with the first channel window only
        constexpr int NG = (CHF + NGEO + FLUSH_GROUP - 1) / FLUSH_GROUP;
        if constexpr (NG == 0) __builtin_amdgcn_wave_barrier();   // the id column above is read below
        const int fsub = lane >> 4, fk = lane & 15;   // 4 instances per atomic instruction, 16 values each
#pragma unroll
        for (int g = 0; g < NG; g++) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < FLUSH_GROUP; k++) {
                const int idx = g * FLUSH_GROUP + k;   // compile-time: feature channel or geometric slot
                if (k >= Lds::FS - 1) continue;        // MF tile holds 10 values per instance
                float v = 0.f;
                if (idx < CHF) v = fac[idx < CHF ? idx : 0];
                else if (idx < CHF + NGEO) v = acc[idx - CHF < 10 ? (idx - CHF >= 0 ? idx - CHF : 0) : 0];
                L.flush[frow * Lds::FS + k] = v;
            }
            __builtin_amdgcn_wave_barrier();
            const int idx = g * FLUSH_GROUP + fk;       // this lane's value index within the instance
#pragma unroll 4
            for (int i0 = 0; i0 < CAP; i0 += 4) {
                const int inst = i0 + fsub;
                // scalar shift by the compile-time part, per-lane shift by the small remainder (a 64-bit per-lane
                // mask per instance would be hoisted out of the chunk loop and cost two registers apiece)
                if (!(((uint32_t)(tmask >> i0) >> fsub) & 1u)) continue;
                const uint32_t gg = __float_as_uint(L.flush[inst * Lds::FS + Lds::FS - 1]);
                const float v = fk < Lds::FS - 1 ? L.flush[inst * Lds::FS + fk] : 0.f;
                if (idx < CHF) {
                    if (idx < a.nc) unsafeAtomicAdd(a.dL_dfeature + (size_t)gg * a.C + a.c0 + idx, v);
                } else if (idx < CHF + NGEO) {
                    if (a.write_base) unsafeAtomicAdd(a.grec + (size_t)gg * GREC + (idx - CHF), v);
                }
            }
        }
        if constexpr (MF) {
            // D[i][j]: lane holds column j = lane & 31 (channel); register r holds row
            // i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (instance = original lane index): every store is one
            // contiguous 32-float run per half-wave.
#pragma unroll
            for (int hh = 0; hh < MH; hh++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int inst = 32 * hh + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (!(((uint32_t)(tmask >> (32 * hh + (r & 3) + 8 * (r >> 2))) >> (4 * (lane >> 5))) & 1u)) continue;
                    const uint32_t gg = __float_as_uint(L.flush[inst * Lds::FS + Lds::FS - 1]);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const int ch = 32 * nb + (lane & 31);
                        if (ch < a.nc) unsafeAtomicAdd(a.dL_dfeature + (size_t)gg * a.C + a.c0 + ch, macc[nb][hh][r]);
                    }
                }
        }
        __builtin_amdgcn_wave_barrier();
        F3DGS_PHASE_END(cyc_flush);
    };

    // ---- walk the list back to front in windows of 64 positions; splats whose 1/255 footprint misses this
    // wave's pixel block are dropped (rect_hit, exact-safe) and the survivors of consecutive windows are packed
    // into full 64-lane chunks with ds_permute (lane 0 = farthest back stays true across windows) ----------------

    SplatLane cur;
    cur.mx = cur.my = cur.ca = cur.cb = cur.cc = cur.op = cur.cr = cur.cg = cur.cbl = cur.dep = 0.f;
    cur.pos = 0; cur.have = false;
    uint32_t cur_gid = 0;
    int count = 0;
    uint32_t cur_min = 0;
    // software pipeline: the ids + records of window k0 - 64 are requested before window k0 is tested,
    // compacted and (possibly) processed, so the two dependent gathers never sit on the critical path
    // three-stage software pipeline over the windows: while window k0 is tested / compacted / processed, the splat
    // records of window k0 - 64 and the list ids of window k0 - 128 are in flight - neither of the two dependent
    // gathers is waited for on the spot (per-phase counters: the walk was 10 % of a wave's lifetime with one stage)
    auto load_recs = [&](int k0w, uint32_t gid, float (&f)[10], bool& have) {
        const uint32_t pos = (uint32_t)(k0w + 63 - lane);
        have = k0w >= 0 && pos < max_last;
#pragma unroll
        for (int k = 0; k < 10; k++) f[k] = 0.f;
        if (have) {
            const SplatRec* rp = a.rec + gid;
            const float4 q0 = rp->q0, q1 = rp->q1;
            const float2 q2 = *reinterpret_cast<const float2*>(&rp->q2);      // blue, depth (the other half is not needed here)
            f[0] = q0.x; f[1] = q0.y; f[2] = q0.z; f[3] = q0.w; f[4] = q1.x; f[5] = q1.y;
            f[6] = q1.z; f[7] = q1.w; f[8] = q2.x; f[9] = q2.y;
        }
    };
    float nf[10];
    bool nhave;
    load_recs(k_top, ngid, nf, nhave);
    for (int k0 = k_top; k0 >= 0; k0 -= 64) {
        const uint32_t pos = (uint32_t)(k0 + 63 - lane);     // lane 0 = farthest back within the window
        float f[10];
#pragma unroll
        for (int k = 0; k < 10; k++) f[k] = nf[k];
        const uint32_t gid = ngid;
        const bool have = nhave;
        ngid = fgid;
        load_recs(k0 - 64, ngid, nf, nhave);
        fgid = load_ids(k0 - 128);
        // the wave's pixel rectangle, converted here from its scalar integers (sgpr_opaque)
        const float wx0 = (float)sgpr_opaque(px0), wy0 = (float)sgpr_opaque(py0);
        const bool hit = have && rect_hit(f[0], f[1], f[2], f[3], f[4], f[5], wx0, wx0 + (float)(PW - 1), wy0, wy0 + (float)(NPIX / PW - 1));
        const unsigned long long hmask = __ballot(hit);
        const int c2 = __popcll(hmask);
        if (c2 == 0) continue;
        // push survivors [first, first + n) of this window (in lane order = back to front) to lanes at .. at + n - 1
        // of the chunk; the other lanes aim at a lane whose result is unused
        // survivors in front of this lane: v_mbcnt counts the mask bits below the lane (no per-lane mask register)
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hmask, 0u));
        auto append = [&](int first, int n, int at) {
            const bool mine = hit && rank >= first && rank < first + n;
            const int dest = mine ? at + rank - first : (at > 0 ? 0 : n & 63);
            const bool recv = lane >= at && lane < at + n;
            auto push = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_permute(dest << 2, __float_as_int(v))); };
            const float m0 = push(f[0]), m1 = push(f[1]), m2 = push(f[2]), m3 = push(f[3]), m4 = push(f[4]);
            const float m5 = push(f[5]), m6 = push(f[6]), m7 = push(f[7]), m8 = push(f[8]), m9 = push(f[9]);
            const uint32_t mp = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)pos);
            const uint32_t mg = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)gid);
            if (recv) {
                cur.mx = m0; cur.my = m1; cur.op = m5;
                cur.ca = m2 * CONIC_SCALE_AC; cur.cb = m3 * CONIC_SCALE_B; cur.cc = m4 * CONIC_SCALE_AC;   // see splat_power2
                cur.cr = m6; cur.cg = m7; cur.cbl = m8; cur.dep = m9; cur.pos = mp; cur_gid = mg;
            }
        };
        // chunks are filled to exactly CAP lanes (every body of a chunk costs the same whatever the number of live
        // lanes): the farthest survivors of the window top the open chunk up, the rest opens the next one(s)
        int first = 0;
        while (first < c2) {
            if (count == CAP) {
                process(cur, cur_gid, cur_min, count);
                count = 0;
                // the chunk is consumed: an explicit reset ends the live range of its twelve registers at the top of
                // process() (lanes the next append does not fill would otherwise carry them through the pixel trips)
                cur.mx = cur.my = cur.ca = cur.cb = cur.cc = cur.op = cur.cr = cur.cg = cur.cbl = cur.dep = 0.f;
                cur.pos = 0; cur_gid = 0;
            }
            const int n = min(c2 - first, CAP - count);
            append(first, n, count);
            count += n;
            first += n;
            cur.have = lane < count;
            cur_min = (uint32_t)k0;      // every survivor of this window sits at a position >= k0
        }
    }
    if (count > 0) process(cur, cur_gid, cur_min, count);
    F3DGS_PHASE_END(cyc_walk);
#ifdef F3DGS_DEV
    if ((a.dev & 8) && lane == 0) {
        atomicAdd(&a.dev_cycles[0], cyc_stage); atomicAdd(&a.dev_cycles[1], cyc_walk); atomicAdd(&a.dev_cycles[2], cyc_trip);
        atomicAdd(&a.dev_cycles[3], cyc_flush); atomicAdd(&a.dev_cycles[4], 1ull);
    }
#endif
}

// Two entry points over one body: the half-wave shape sits a few registers above the three-waves-per-SIMD budget
// (156 + 16 against 168) and loses a third of its occupancy there; squeezed into the budget it keeps three waves.
template <int CH, int NPIX, bool MF, int U, bool GEO, bool HALF>
__global__ void __launch_bounds__(64) render_backward_kernel(BwdArgs a) {
    render_backward_body<CH, NPIX, MF, U, GEO, HALF>(a);
}
template <int CH, int NPIX, bool MF, int U, bool GEO, bool HALF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) render_backward_kernel_w3(BwdArgs a) {
    render_backward_body<CH, NPIX, MF, U, GEO, HALF>(a);
}

template <int CH, bool MF>
void launch_one(const BwdArgs& a, hipStream_t s) {
    constexpr int NPIX = 64;
    const size_t lds = a.half ? sizeof(BwdLds<CH, NPIX, MF, true>) : sizeof(BwdLds<CH, NPIX, MF, false>);
    const int tiles = a.gx * a.gy;
    const dim3 grid(a.order ? 8 * (256 / NPIX) * ((tiles + 7) / 8) : tiles * (256 / NPIX));
    // later channel windows skip the geometric half of the work
    if (a.half) {        // chunks of 32 instances against two pixel halves (default)
        if constexpr (CH <= 32) {
            if (CH > 0 && !a.write_base) hipLaunchKernelGGL((render_backward_kernel_w3<CH, NPIX, MF, 4, (CH == 0), true>), grid, dim3(64), lds, s, a);
            else hipLaunchKernelGGL((render_backward_kernel_w3<CH, NPIX, MF, 4, true, true>), grid, dim3(64), lds, s, a);
        } else {
            if (!a.write_base) hipLaunchKernelGGL((render_backward_kernel<CH, NPIX, MF, 4, false, true>), grid, dim3(64), lds, s, a);
            else hipLaunchKernelGGL((render_backward_kernel<CH, NPIX, MF, 4, true, true>), grid, dim3(64), lds, s, a);
        }
        return;
    }
    if (CH > 0 && !a.write_base) hipLaunchKernelGGL((render_backward_kernel<CH, NPIX, MF, 4, (CH == 0), false>), grid, dim3(64), lds, s, a);
    else hipLaunchKernelGGL((render_backward_kernel<CH, NPIX, MF, 4, true, false>), grid, dim3(64), lds, s, a);
}

}  // namespace

int launch_render_backward(const ViewParams& vp, int C, const uint2* ranges, const uint32_t* point_list,
                           const SplatRec* rec, const float* final_T, const uint32_t* n_contrib,
                           const float* dL_dpix, const float* dL_dfeat, const float* dL_ddepth, float* grec,
                           float* dL_dfeature, const uint32_t* tile_len, uint32_t* tile_order, const LowresGrad* lowres,
                           int contraction, const uint32_t* gate, hipStream_t s) {
    const bool bf16 = contraction != 0;
    BwdArgs a;
    a.order = nullptr;
    a.gate = contraction == 3 ? gate : nullptr; a.gate_want = 0;
    band_perm_params(vp.gx, vp.gy, vp.band0, vp.band1, &a.band_b0, &a.band_tb);
    a.glow = nullptr; a.gscale = nullptr; a.gHg = a.gWg = 0; a.gsy = a.gsx = 0.f;
    a.m44 = 0; a.split16 = 0; a.bf16 = 0; a.neg_half_w = a.neg_half_h = 0.f;
    const bool low = lowres && lowres->gx && C > 0;
    if (low) {
        a.glow = lowres->gx; a.gscale = lowres->scale; a.gHg = lowres->Hg; a.gWg = lowres->Wg;
        a.gsy = lowres->Hg > 1 ? (float)(vp.H - 1) / (float)(lowres->Hg - 1) : 0.f;
        a.gsx = lowres->Wg > 1 ? (float)(vp.W - 1) / (float)(lowres->Wg - 1) : 0.f;
    }
    a.ranges = ranges; a.point_list = point_list; a.rec = rec;
    a.bg = vp.bg;
    a.final_T = final_T; a.n_contrib = n_contrib; a.dL_dpix = dL_dpix; a.dL_dfeat = dL_dfeat;
    a.dL_ddepth = dL_ddepth; a.grec = grec; a.dL_dfeature = dL_dfeature;
    a.W = vp.W; a.H = vp.H; a.gx = vp.gx; a.gy = vp.gy; a.C = C;
    const Options& opt = options();
#ifdef F3DGS_DEV
    a.dev = opt.dev;
    static unsigned long long* dev_cycles = nullptr;
    if (!dev_cycles) (void)hipMalloc(&dev_cycles, (size_t)8 * sizeof(unsigned long long) * 262144 * 4);
    a.dev_cycles = dev_cycles;
    if (a.dev & 8) (void)hipMemsetAsync(dev_cycles, 0, 8 * sizeof(unsigned long long), s);
    struct Report {     // prints the phase totals of this launch when it goes out of scope (synchronises: dev only)
        unsigned long long* d; hipStream_t s; bool on;
        ~Report() {
            if (!on) return;
            unsigned long long h[8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            const double w = (double)(h[4] ? h[4] : 1);
            fprintf(stderr, "[f3dgs dev] blend backward, cycles per wave: staging %.0f  window walk %.0f  pixel trips %.0f  flush %.0f  (%llu waves)\n",
                    h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4]);
        }
    } report{dev_cycles, s, (a.dev & 8) != 0};
#endif
    // pixel-lane formulation (render_bwd_pl.hip): option bwd_pl = 1 always, 0 never, -1 (default) from 5 channels on (up to 16 its
    // feature and moment blocks are split over the waves by quadrants, option bwd_split16).  c2-sized scenes, ms per launch,
    // instance-lane / pixel-lane: C = 0 0.452 / 0.603, 3 0.568 / 0.603, 8 0.651 / 0.615, 12 0.66 / 0.619, 16 0.66 / 0.618
    // (a low-resolution feature-map gradient is taken by the pixel-lane kernel only: the caller has checked feature_mfma)
    // (round 5, with the bf16 contractions - option bwd_bf16 - the pixel-lane kernel wins from the first channel on: c2-sized scenes,
    // instance-lane / pixel-lane: C = 0 0.445 / 0.491, 3 0.567 / 0.509, 4 0.566 / 0.510, 8 0.650 / 0.519)
    if (low || ((opt.bwd_pl > 0 || (opt.bwd_pl < 0 && C > (bf16 ? 0 : 4))) && opt.feature_mfma)) {
        if (opt.bwd_order && tile_len && tile_order) {
            launch_tile_order(tile_len, (size_t)vp.gx * vp.gy, tile_order, a.band_b0, a.band_tb, s);
            a.order = tile_order;
        }
        a.half = 0;
        a.m44 = opt.bwd_m44;
        a.split16 = opt.bwd_split16;
        a.bf16 = contraction;            // of the first window (geometric sums + 32 channels); see launch_render_backward_pl for the later ones
        launch_render_backward_pl(a, C, s);
#ifdef F3DGS_DEV
        if (a.dev & 8) {
            unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipStreamSynchronize(s);
            const size_t nw = (size_t)a.gx * a.gy * 4;     // waves
            std::vector<unsigned long long> all(nw * 8);
            (void)hipMemcpy(all.data(), dev_cycles, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            for (size_t i = 0; i < nw; i++) for (int k = 0; k < 8; k++) h[k] += all[i * 8 + k];
            const double w = (double)(h[7] ? h[7] : 1);
            fprintf(stderr, "[f3dgs dev] pixel-lane backward, cycles per wave: staging %.0f  window %.0f  phase1 %.0f  phase2+merge %.0f  flush %.0f | chunks/wave %.2f entries/wave %.1f (%llu waves); per chunk: phase1 %.0f phase2 %.0f flush %.0f\n",
                    h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w, h[7], (double)h[2] / h[5], (double)h[3] / h[5], (double)h[4] / h[5]);
            report.on = false;
        }
#endif
        return a.bf16;
    }
    a.half = opt.bwd_half != 0;
    const bool mf = opt.feature_mfma != 0;
    // longest walks first, as in the pixel-lane kernel: the launch is ~10 rounds of single-wave workgroups whose lifetimes
    // differ by an order of magnitude (the order is built once per call, every channel window uses it)
    // (c2: 0.677 -> 0.660 ms including the order launch; below ~1000 tiles the launch costs more than the tail it removes)
    if (opt.bwd_order && tile_len && tile_order && (size_t)vp.gx * vp.gy >= 1024) {
        launch_tile_order(tile_len, (size_t)vp.gx * vp.gy, tile_order, a.band_b0, a.band_tb, s);
        a.order = tile_order;
    }
    if (C == 0) {
        a.c0 = 0; a.nc = 0; a.write_base = 1;
        launch_one<0, false>(a, s);
        return 0;
    }
    // channel windows of up to 64; the geometric sums ride along with the first window only
    for (int c0 = 0; c0 < C; c0 += 64) {
        a.c0 = c0; a.nc = min(64, C - c0); a.write_base = (c0 == 0);
        if (a.nc <= 4) launch_one<4, false>(a, s);
        else if (a.nc <= 16) launch_one<16, false>(a, s);
        else if (a.nc <= 32) { if (mf) launch_one<32, true>(a, s); else launch_one<32, false>(a, s); }
        else { if (mf) launch_one<64, true>(a, s); else launch_one<64, false>(a, s); }
    }
    return 0;
}

}  // namespace f3dgs
