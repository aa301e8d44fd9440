// render_fwd.hip — front-to-back alpha-composited RGB + N-dim feature + depth (forward blend).
//
// Semantics: R/cuda_rasterizer/forward.cu:261-396 (R = submodules/diff-gaussian-rasterization-feature),
// including quirks Q4 (no background on depth/feature) and Q5 (the Gaussian that would push T below
// 1e-4 is not blended and ends the pixel; n_contrib = list position of the last blended entry).
//
// CDNA4 design (see render_common.h for the decomposition):
//   * per-wave autonomous walk over 64-instance chunks; splat records (48 B) are gathered with three
//     16-byte loads per lane, the C-float feature vectors with coalesced 16-byte loads
//     (8 lanes x 16 B = one 128-B vector at C = 32), both staged in the wave's private LDS slice;
//   * the inner loop reads one instance's data as LDS broadcasts (same address in all lanes);
//   * PPL pixels per lane amortise every broadcast read over PPL * 64 pixels and keep the blend
//     VALU-bound instead of LDS-bound;
//   * wave-uniform ballots skip the feature FMAs for quadrants a splat does not reach and end the
//     wave as soon as all of its pixels are saturated;
//   * channels beyond CH are handled by re-walking the list per 64-channel window.

#include <atomic>
#include <cstddef>

#include "render_common.h"
#include "fwd_group.h"

namespace f3dgs {

namespace {

template <int CH>
struct FwdChunk {
    float4 geo[64];  // mean_x, mean_y, conic_a, conic_b
    float2 co[64];   // conic_c, opacity
    float4 cd[64];   // r, g, b, depth
    uint32_t id[64];
    float feat[64][CH > 0 ? CH : 4];
};

struct FwdArgs {
    const uint2* ranges_enc;   // BinState::ranges_enc, decoded into `ranges` by the launch that has write_base set
    uint2* ranges;
    const uint32_t* point_list;
    const SplatRec* rec;
    const float* feat;
    const float* bg;
    float* final_T;
    uint32_t* n_contrib;
    float* out_color;
    float* out_feat;
    float* out_depth;
    int W, H, gx, gy;
    int C;        // total feature channels (row stride of feat)
    int c0, nc;   // channel window handled by this launch
    int write_base;  // 1: also write colour / depth / final_T / n_contrib
    uint32_t* tile_len;   // per tile: max n_contrib of its pixels (first window's launch; zero-filled beforehand)
    int solo;        // one quadrant per wave: 64-thread workgroups, grid = 4 x tiles
    uint32_t band_b0, band_tb;   // band_perm (common.h): first tile / tiles of the listed band, (0, 0) = whole view
    int band_r0, band_r1;        // tile rows of the listed band; outside them a workgroup only records the empty range - the
                                 // pixels there are written by fill_outside_band_kernel (whole view: 0, gy)
    int dev;         // development builds: work-skipping bits (128: no feature gathers  256: no matrix instructions  512: no alpha evaluation skip)
};

#ifdef F3DGS_DEV
#define FW_DEV_SKIP(bit) (a.dev & (bit))
#else
#define FW_DEV_SKIP(bit) false
#endif

template <int CH, int PPL>
__global__ void __launch_bounds__(256 / PPL) render_forward_kernel(FwdArgs a) {
    constexpr int NW = 4 / PPL;
    constexpr int CHV = CH / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    // a.solo (one quadrant per wave): one 64-thread workgroup per quadrant, see render_forward_mfma_body
    const uint32_t vb = xcd_remap(blockIdx.x, gridDim.x);
    const bool solo = PPL == 1 && a.solo;
    const int wave = solo ? (int)(vb & 3u) : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    FwdChunk<CH>& ck = reinterpret_cast<FwdChunk<CH>*>(smem)[(NW > 1 && !solo) ? wave : 0];

    const uint32_t tile = band_perm(solo ? vb >> 2 : vb, (uint32_t)(a.gx * a.gy), a.band_b0, a.band_tb);
    const int tx = tile % a.gx, ty = tile / a.gx;
    if (ty < a.band_r0 || ty >= a.band_r1) {      // (workgroup-uniform) a tile outside the listed band: see FwdArgs::band_r0
        if (a.write_base && threadIdx.x == 0 && (!solo || wave == 0)) a.ranges[tile] = make_uint2(0u, 0u);
        return;
    }
    uint2 rg;
    if (a.write_base) {   // first channel window: decode {min, UINT_MAX - (max + 1)}; untouched = empty tile
        const uint2 e = a.ranges_enc[tile];
        rg = e.x == 0xFFFFFFFFu ? make_uint2(0u, 0u) : make_uint2(e.x, 0xFFFFFFFFu - e.y);
        if (threadIdx.x == 0) a.ranges[tile] = rg;
    } else {
        rg = a.ranges[tile];
    }
    const uint32_t r_lo = __builtin_amdgcn_readfirstlane((int)rg.x), r_hi = __builtin_amdgcn_readfirstlane((int)rg.y);

    const int lx = lane & 7, ly = lane >> 3;
    float pxf[PPL], pyf[PPL];
    int pix_id[PPL];
    bool inside[PPL], done[PPL];
    float T[PPL], col[PPL][3], dep[PPL];
    float sf[PPL][CH > 0 ? CH : 1];
    uint32_t last[PPL];
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const int q = wave * PPL + p;
        const int x = tx * TILE + (q & 1) * 8 + lx, y = ty * TILE + (q >> 1) * 8 + ly;
        pxf[p] = (float)x; pyf[p] = (float)y;
        inside[p] = x < a.W && y < a.H;
        pix_id[p] = y * a.W + x;
        done[p] = !inside[p];
        T[p] = 1.0f; dep[p] = 0.f; last[p] = 0;
        col[p][0] = col[p][1] = col[p][2] = 0.f;
#pragma unroll
        for (int c = 0; c < (CH > 0 ? CH : 1); c++) sf[p][c] = 0.f;
    }

    // software pipeline (as in the matrix-pipe variant below): records of chunk k+1 and list ids of chunk k+2 are
    // requested while chunk k is blended
    uint32_t n_id = 0, f_id = 0;
    float4 n_q0 = make_float4(0, 0, 0, 0), n_q1 = n_q0, n_q2 = n_q0;
    if (r_lo + lane < r_hi) n_id = a.point_list[r_lo + lane];
    if (r_lo + 64 + lane < r_hi) f_id = a.point_list[r_lo + 64 + lane];
    if (r_lo + lane < r_hi) {
        const SplatRec* rp = a.rec + n_id;
        n_q0 = rp->q0; n_q1 = rp->q1; n_q2 = rp->q2;
    }
    for (uint32_t base = r_lo; base < r_hi; base += 64) {
        bool alld = true;
#pragma unroll
        for (int p = 0; p < PPL; p++) alld = alld && done[p];
        if (__all(alld)) break;
        const int cnt = (int)min(64u, r_hi - base);
        // ---- stage one chunk: records (from the prefetch registers) ...
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            ck.geo[lane] = make_float4(n_q0.x, n_q0.y, n_q0.z * CONIC_SCALE_AC, n_q0.w * CONIC_SCALE_B);   // see splat_power2
            ck.co[lane] = make_float2(n_q1.x * CONIC_SCALE_AC, n_q1.y);
            ck.cd[lane] = make_float4(n_q1.z, n_q1.w, n_q2.x, n_q2.y);
            ck.id[lane] = n_id;
        }
        __builtin_amdgcn_wave_barrier();
        n_id = f_id;
        if (base + 64 + lane < r_hi) {
            const SplatRec* rp = a.rec + n_id;
            n_q0 = rp->q0; n_q1 = rp->q1; n_q2 = rp->q2;
        }
        if (base + 128 + lane < r_hi) f_id = a.point_list[base + 128 + lane];
        // ---- ... and feature vectors (coalesced: CHV lanes x 16 B per instance; LDS-direct loads when the window is full
        // width: lane l of a request lands at base + 16 l = the row-major image, all requests in flight together)
        if constexpr (CH > 0) {
            const bool vec_ok = (a.C & 3) == 0 && (a.c0 & 3) == 0;
            if (vec_ok && a.nc == CH) {
                using lds_ptr = __attribute__((address_space(3))) void*;
                using gbl_ptr = const __attribute__((address_space(1))) void*;
#pragma unroll
                for (int it = 0; it < (64 * CHV + 63) / 64; it++) {
                    const int e = it * 64 + lane;
                    if (e < cnt * CHV) {
                        const uint32_t g = ck.id[e / CHV];
                        const float* src = a.feat + (size_t)g * a.C + a.c0 + 4 * (e % CHV);
                        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(&ck.feat[0][0] + it * 256), 16, 0, 0);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                for (int e = lane; e < cnt * CHV; e += 64) {
                    const int inst = e / CHV, v = e % CHV;
                    const uint32_t g = ck.id[inst];
                    const float* src = a.feat + (size_t)g * a.C + a.c0 + 4 * v;
                    float4 f;
                    if (vec_ok && 4 * v + 3 < a.nc) {
                        f = *reinterpret_cast<const float4*>(src);
                    } else {
                        f.x = 4 * v + 0 < a.nc ? src[0] : 0.f;
                        f.y = 4 * v + 1 < a.nc ? src[1] : 0.f;
                        f.z = 4 * v + 2 < a.nc ? src[2] : 0.f;
                        f.w = 4 * v + 3 < a.nc ? src[3] : 0.f;
                    }
                    *reinterpret_cast<float4*>(&ck.feat[inst][4 * v]) = f;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- blend the chunk
        for (int j = 0; j < cnt; j++) {
            const float4 g0 = ck.geo[j];
            const float2 g1 = ck.co[j];
            float w[PPL];
            bool any_blend = false;
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                const float dx = g0.x - pxf[p], dy = g0.y - pyf[p];
                const float power = splat_power2(dx, dy, g0.z, g0.w, g1.x);
                const float alpha = fminf(ALPHA_MAX, g1.y * __builtin_amdgcn_exp2f(power));
                bool ok = !done[p] && !(power > 0.0f) && !(alpha < ALPHA_MIN);
                const float test_T = T[p] * (1.0f - alpha);
                if (ok && test_T < T_MIN) {
                    done[p] = true;
                    ok = false;
                }
                w[p] = ok ? alpha * T[p] : 0.0f;
                if (ok) {
                    T[p] = test_T;
                    last[p] = base - r_lo + j + 1;
                }
                any_blend = any_blend || ok;
            }
            if (__any(any_blend)) {
                const float4 cd = ck.cd[j];
#pragma unroll
                for (int p = 0; p < PPL; p++) {
                    col[p][0] = fmaf(cd.x, w[p], col[p][0]);
                    col[p][1] = fmaf(cd.y, w[p], col[p][1]);
                    col[p][2] = fmaf(cd.z, w[p], col[p][2]);
                    dep[p] = fmaf(cd.w, w[p], dep[p]);
                }
                if constexpr (CH > 0) {
#pragma unroll
                    for (int v = 0; v < CHV; v++) {
                        const float4 f = *reinterpret_cast<const float4*>(&ck.feat[j][4 * v]);
#pragma unroll
                        for (int p = 0; p < PPL; p++) {
                            sf[p][4 * v + 0] = fmaf(f.x, w[p], sf[p][4 * v + 0]);
                            sf[p][4 * v + 1] = fmaf(f.y, w[p], sf[p][4 * v + 1]);
                            sf[p][4 * v + 2] = fmaf(f.z, w[p], sf[p][4 * v + 2]);
                            sf[p][4 * v + 3] = fmaf(f.w, w[p], sf[p][4 * v + 3]);
                        }
                    }
                }
            }
        }
    }

    const size_t HW = (size_t)a.W * a.H;
    if (a.write_base) {     // the tile's longest walk: what the pixel-lane backward orders its workgroups by
        uint32_t m = 0;
#pragma unroll
        for (int p = 0; p < PPL; p++) m = max(m, last[p]);
        m = wave_max_u32(m);
        if (lane == 0 && m) atomicMax(&a.tile_len[tile], m);
    }
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        if (!inside[p]) continue;
        const size_t pid = (size_t)pix_id[p];
        if (a.write_base) {
            a.final_T[pid] = T[p];
            a.n_contrib[pid] = last[p];
            a.out_color[pid] = col[p][0] + T[p] * a.bg[0];
            a.out_color[HW + pid] = col[p][1] + T[p] * a.bg[1];
            a.out_color[2 * HW + pid] = col[p][2] + T[p] * a.bg[2];
            a.out_depth[pid] = dep[p];
        }
        if constexpr (CH > 0) {
#pragma unroll
            for (int c = 0; c < CH; c++)
                if (c < a.nc) a.out_feat[(size_t)(a.c0 + c) * HW + pid] = sf[p][c];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Variant with the feature blend on the matrix pipe.  out[px][c] += sum_g w[px][g] f[g][c] is a contraction
// over the instances of the list; measured on MI355X the blend kernels are VALU-issue bound (SQ counters in
// profiles/), not HBM bound, so the C FMAs per (pixel, instance) move to exact-fp32 v_mfma_f32_32x32x2_f32
// (same rounding as an fmaf chain) and run concurrently with the VALU alpha evaluation.  Instances are taken
// two at a time (K = 2): A[i][k] = w of pixel i for instance j+k, built from the two per-lane weights with
// one v_permlane32_swap; B[k][n] = feature n of instance j+k, one ds_read_b32 per lane.
// (FwdEntry - one staged, compacted list entry - and the group step of the blend loop live in fwd_group.h)

template <int CH, int CHK>
struct FwdChunkMF {
    FwdEntry ent[CHK];
    // epilogue transpose tile: [channel][65] for all 64 pixels at once.  ([channel][33], one half of the pixels at a time,
    // would cut the wave's LDS from 9.6 to 5.6 KB - and a fifth wave per SIMD fits the registers with 3-4 spilled - but the
    // half-width output stores cost more than the occupancy buys: c3 forward 0.363 -> 0.378 ms at four waves, 0.377 at five;
    // the code path stays for the record, TS = 33 selects it.)
    static constexpr int TS = 65;
    static constexpr int FT = (CHK * CH > 32 * TS) ? CHK * CH : 32 * TS;
    float feat[FT];        // row-major [instance][channel]; reused as the epilogue transpose tile
};

// The next chunk's ids and splat records are prefetched into registers while the current chunk is blended.
// (Fetching the B operand straight from global memory, prefetched two groups ahead, was measured slower
// than staging the chunk's feature rows in LDS: 0.62 vs 0.57 ms at config c3.)
// BASE = false: a later channel window of a wide feature (C > 64): colour, depth, final_T and n_contrib were written by
// the first window's launch; only the blend weights and the feature contraction are needed again.
// CH = 16 (windows of 5..16 channels): the 32-column B operand has room to spare, so on the base launch columns 16..19 carry
// the splat's colour and depth (CDB: read straight out of the staged entry) and their four multiply-adds per (pixel, entry)
// leave the vector pipe - the pipe this kernel is bound by.
template <int CH, int PPL, int CHK, int GI, bool BASE>
__device__ __forceinline__ void render_forward_mfma_body(const FwdArgs& a) {
    constexpr int NW = 4 / PPL;
    constexpr int NB = (CH + 31) / 32;
    constexpr bool CDB = BASE && CH == 16;
    constexpr int NP = GI / 2;        // instance pairs (MFMA K = 2) per group
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    // a.solo (one quadrant per wave only): every quadrant wave is its own 64-thread workgroup - the waves never
    // synchronise, so a workgroup of four only keeps the slots of its finished quadrants occupied until the slowest one
    // is done.  Four consecutive virtual ids = the quadrants of one tile (same XCD, see xcd_remap).
    const uint32_t vb = xcd_remap(blockIdx.x, gridDim.x);
    const bool solo = PPL == 1 && a.solo;
    const int wave = solo ? (int)(vb & 3u) : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    FwdChunkMF<CH, CHK>& ck = reinterpret_cast<FwdChunkMF<CH, CHK>*>(smem)[(NW > 1 && !solo) ? wave : 0];

    const uint32_t tile = band_perm(solo ? vb >> 2 : vb, (uint32_t)(a.gx * a.gy), a.band_b0, a.band_tb);
    const int tx = tile % a.gx, ty = tile / a.gx;
    if (ty < a.band_r0 || ty >= a.band_r1) {      // (workgroup-uniform) a tile outside the listed band: see FwdArgs::band_r0
        if (a.write_base && threadIdx.x == 0 && (!solo || wave == 0)) a.ranges[tile] = make_uint2(0u, 0u);
        return;
    }
    uint2 rg;
    if (a.write_base) {   // first channel window: decode {min, UINT_MAX - (max + 1)}; untouched = empty tile
        const uint2 e = a.ranges_enc[tile];
        rg = e.x == 0xFFFFFFFFu ? make_uint2(0u, 0u) : make_uint2(e.x, 0xFFFFFFFFu - e.y);
        if (threadIdx.x == 0) a.ranges[tile] = rg;
    } else {
        rg = a.ranges[tile];
    }
    const uint32_t r_lo = __builtin_amdgcn_readfirstlane((int)rg.x);
    uint32_t r_hi = __builtin_amdgcn_readfirstlane((int)rg.y);

    // pixel-centre rectangle covered by this wave (PPL quadrants): used by the wave-level footprint test
    // (kept as scalar integers; converted where the test runs, see sgpr_opaque)
    const int q0_ = wave * PPL, q1_ = wave * PPL + PPL - 1;
    const int iwx0 = tx * TILE + (q0_ & 1) * 8, iwx1 = tx * TILE + (q1_ & 1) * 8 + 7;
    const int iwy0 = ty * TILE + (q0_ >> 1) * 8, iwy1 = ty * TILE + (q1_ >> 1) * 8 + 7;
    const int lx = lane & 7, ly = lane >> 3;
    int pix_id[PPL];
    // T carries the pixel's "finished" flag in its sign bit (T > 0 while the pixel is still blending; -T_final
    // afterwards): a finished pixel then fails the test_T >= T_MIN check by itself and no per-lane boolean has
    // to live in a register.
    bool inside[PPL];
    FwdPixels<CH, PPL> px;
    auto& pxf = px.pxf; auto& pyf = px.pyf; auto& T = px.T; auto& col = px.col; auto& dep = px.dep; auto& last = px.last;
    auto& acc = px.acc;
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const int q = wave * PPL + p;
        const int x = tx * TILE + (q & 1) * 8 + lx, y = ty * TILE + (q >> 1) * 8 + ly;
        pxf[p] = (float)x; pyf[p] = (float)y;
        inside[p] = x < a.W && y < a.H;
        pix_id[p] = y * a.W + x;
        T[p] = inside[p] ? 1.0f : -1.0f; dep[p] = 0.f; last[p] = 0;
        if constexpr (!BASE) {
            // a later channel window knows where every pixel's walk ended (n_contrib of the first window: the position of its last
            // contributor): "pos <= n_contrib" replaces the transmittance test and the finished flag - the entries that pass are
            // the same, entry by entry (an entry fails the transmittance test only by ending the walk, and then lies behind the
            // last contributor) - and the quadrant stops at its deepest pixel instead of looking for live ones
            T[p] = 1.0f;
            last[p] = inside[p] ? a.n_contrib[pix_id[p]] : 0u;
        }
        col[p][0] = col[p][1] = col[p][2] = 0.f;
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[p][h][nb][r] = 0.f;
    }

    if constexpr (!BASE) {
        uint32_t m = 0;
#pragma unroll
        for (int p = 0; p < PPL; p++) m = max(m, last[p]);
        r_hi = min(r_hi, r_lo + (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(m)));
    }

    // CH = 16: where this lane's B column lives (dwords from the start of the chunk image, per staged entry): channels 0..15
    // in the feature rows; columns 16..19 = colour and depth inside the 12-dword entry; the remaining columns re-read those
    // (finite values; their sums are never looked at)
    int b_stride = 0, b_off = 0;
    if constexpr (CH == 16) {
        const int n = lane & 31;
        using Chunk = FwdChunkMF<CH, CHK>;
        constexpr int FEAT_OFS = (int)(offsetof(Chunk, feat) / sizeof(float));
        b_stride = n < 16 ? CH : (int)(sizeof(FwdEntry) / sizeof(float));
        b_off = n < 16 ? FEAT_OFS + n : (int)(offsetof(FwdEntry, cd) / sizeof(float)) + ((n - 16) & 3);
    }

    // software pipeline: the splat records of chunk k+1 and the list ids of chunk k+2 are requested while chunk k
    // is blended, so neither of the two dependent gathers (id, then record) is ever waited for on the spot
    uint32_t n_id = 0, f_id = 0;
    float4 n_q0 = make_float4(0, 0, 0, 0), n_q1 = n_q0;
    float2 n_q2 = make_float2(0, 0);          // blue, depth (the radius half of q2 is not needed here)
    if (lane < CHK && r_lo + lane < r_hi) n_id = a.point_list[r_lo + lane];
    if (lane < CHK && r_lo + CHK + lane < r_hi) f_id = a.point_list[r_lo + CHK + lane];
    if (lane < CHK && r_lo + lane < r_hi) {
        const SplatRec* rp = a.rec + n_id;
        n_q0 = rp->q0; n_q1 = rp->q1; n_q2 = *reinterpret_cast<const float2*>(&rp->q2);
    }

    for (uint32_t base = r_lo; base < r_hi; base += CHK) {
        if constexpr (BASE) {
            bool anylive = false;
#pragma unroll
            for (int p = 0; p < PPL; p++) anylive = anylive || T[p] > 0.0f;
            if (!__any(anylive)) break;
        }
        const int cnt_in = (int)min((uint32_t)CHK, r_hi - base);
        // wave-level culling: drop splats whose 1/255 footprint misses this wave's pixel block, compact the rest
        const bool hit = lane < cnt_in && rect_hit(n_q0.x, n_q0.y, n_q0.z, n_q0.w, n_q1.x, n_q1.y, (float)sgpr_opaque(iwx0),
                                                   (float)sgpr_opaque(iwx1), (float)sgpr_opaque(iwy0), (float)sgpr_opaque(iwy1));
        const unsigned long long hmask = __ballot(hit);
        const int cnt = __popcll(hmask);
        const int slot = __popcll(hmask & ((1ull << lane) - 1ull));
        __builtin_amdgcn_wave_barrier();
        if (hit) {
            FwdEntry en;
            en.geo = make_float4(n_q0.x, n_q0.y, n_q0.z * CONIC_SCALE_AC, n_q0.w * CONIC_SCALE_B);   // see splat_power2
            en.cd = make_float4(n_q1.z, n_q1.w, n_q2.x, n_q2.y);
            en.co_c = n_q1.x * CONIC_SCALE_AC; en.co_o = n_q1.y;
            en.pos = base - r_lo + lane + 1;
            en.id = n_id;
            ck.ent[slot] = en;
        }
        __builtin_amdgcn_wave_barrier();
        // prefetch: records of the next chunk (its ids arrived during the previous chunk), ids of the one after
        n_id = f_id;
        if (lane < CHK && base + CHK + lane < r_hi) {
            const SplatRec* rp = a.rec + n_id;
            n_q0 = rp->q0; n_q1 = rp->q1; n_q2 = *reinterpret_cast<const float2*>(&rp->q2);
        }
        if (lane < CHK && base + 2 * CHK + lane < r_hi) f_id = a.point_list[base + 2 * CHK + lane];

        // feature rows of the chunk -> LDS (coalesced: CH/4 lanes x 16 B per instance).  Full-width windows go through
        // LDS-direct loads (global_load_lds_dwordx4: lane l of a request lands at base + 16 l, which IS the row-major
        // image because element e = 64 it + lane sits at byte 16 e): all requests of the chunk are in flight together
        // and no data register is involved; one wait before the blend.
        constexpr int CHV = CH / 4;
        const bool vec_ok = (a.C & 3) == 0 && (a.c0 & 3) == 0;
        if (FW_DEV_SKIP(128)) {
        } else if (vec_ok && a.nc == CH) {
            using lds_ptr = __attribute__((address_space(3))) void*;
            using gbl_ptr = const __attribute__((address_space(1))) void*;
#pragma unroll
            for (int it = 0; it < CHK * CHV / 64; it++) {
                const int e = it * 64 + lane;
                if (e < cnt * CHV) {
                    const uint32_t g = ck.ent[e / CHV].id;
                    const float* src = a.feat + (size_t)g * a.C + a.c0 + 4 * (e % CHV);
                    __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)&ck.feat[it * 256], 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);    // one address pair live at a time (the requests are asynchronous anyway)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (int e = lane; e < cnt * CHV; e += 64) {
                const int inst = e / CHV, v = e % CHV;
                const uint32_t g = ck.ent[inst].id;
                const float* src = a.feat + (size_t)g * a.C + a.c0 + 4 * v;
                float4 f;
                if (vec_ok && 4 * v + 3 < a.nc) {
                    f = *reinterpret_cast<const float4*>(src);
                } else {
                    f.x = 4 * v + 0 < a.nc ? src[0] : 0.f;
                    f.y = 4 * v + 1 < a.nc ? src[1] : 0.f;
                    f.z = 4 * v + 2 < a.nc ? src[2] : 0.f;
                    f.w = 4 * v + 3 < a.nc ? src[3] : 0.f;
                }
                *reinterpret_cast<float4*>(&ck.feat[inst * CH + 4 * v]) = f;
            }
        }
        // an odd (not a multiple of GI) count is padded with null entries - opacity 0: alpha = 0 fails the 1/255 test at every
        // pixel; a zero feature row - so that the blend loop below needs no "is there a second entry" logic at all
        if ((cnt & (GI - 1)) != 0) {
            for (int q = cnt; q < ((cnt + GI - 1) & ~(GI - 1)); q++) {
                if (lane < 12) reinterpret_cast<float*>(&ck.ent[q])[lane] = 0.0f;
                for (int c = lane; c < CH; c += 64) ck.feat[q * CH + c] = 0.0f;
            }
        }
        __builtin_amdgcn_wave_barrier();

        for (int j = 0; j < cnt; j += GI)
            fwd_blend_group<CH, PPL, GI, BASE>(ck.ent, ck.feat, reinterpret_cast<const float*>(&ck), j, lane, b_stride, b_off, FW_DEV_SKIP(256), px);
    }

    const size_t HW = (size_t)a.W * a.H;
    if (BASE && a.write_base) {     // the tile's longest walk: what the pixel-lane backward orders its workgroups by
        uint32_t m = 0;
#pragma unroll
        for (int p = 0; p < PPL; p++) m = max(m, last[p]);
        m = wave_max_u32(m);
        if (lane == 0 && m) atomicMax(&a.tile_len[tile], m);
    }
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        if (BASE && !CDB && a.write_base && inside[p]) {
            const size_t pid = (size_t)pix_id[p];
            const float Tf = fabsf(T[p]);
            a.final_T[pid] = Tf;
            a.n_contrib[pid] = last[p];
            a.out_color[pid] = col[p][0] + Tf * a.bg[0];
            a.out_color[HW + pid] = col[p][1] + Tf * a.bg[1];
            a.out_color[2 * HW + pid] = col[p][2] + Tf * a.bg[2];
            a.out_depth[pid] = dep[p];
        }
        // D[i][n]: lane holds column n = lane & 31 (channel), register r holds row i = (r&3) + 8(r>>2) + 4(lane>>5)
        // (pixel 32h + i of this slot).  Transpose through LDS so that lanes are pixels again: [channel][65] in one go, or
        // (TS = 33) one half of the pixels at a time - the lanes of half h then write their own pixels.
        constexpr int TS = FwdChunkMF<CH, CHK>::TS;
        constexpr int NH = TS == 33 ? 2 : 1;          // transpose rounds per column block
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
#pragma unroll
            for (int hr = 0; hr < NH; hr++) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int h = (NH == 2 ? hr : 0); h < (NH == 2 ? hr + 1 : 2); h++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int i = (NH == 2 ? 0 : 32 * h) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        ck.feat[(lane & 31) * TS + i] = acc[p][h][nb][r];
                    }
                __builtin_amdgcn_wave_barrier();
                const int col = NH == 2 ? (lane & 31) : lane;
                if (inside[p] && (NH == 1 || (lane >> 5) == hr)) {
#pragma unroll 8
                    for (int n = 0; n < (CH < 32 ? CH : 32); n++)
                        if (32 * nb + n < a.nc)
                            a.out_feat[(size_t)(a.c0 + 32 * nb + n) * HW + (size_t)pix_id[p]] = ck.feat[n * TS + col];
                    if constexpr (CDB) {
                        if (a.write_base) {      // columns 16..19 of the contraction: red, green, blue, depth
                            const size_t pid = (size_t)pix_id[p];
                            const float Tf = fabsf(T[p]);
                            a.final_T[pid] = Tf;
                            a.n_contrib[pid] = last[p];
                            a.out_color[pid] = ck.feat[16 * TS + col] + Tf * a.bg[0];
                            a.out_color[HW + pid] = ck.feat[17 * TS + col] + Tf * a.bg[1];
                            a.out_color[2 * HW + pid] = ck.feat[18 * TS + col] + Tf * a.bg[2];
                            a.out_depth[pid] = ck.feat[19 * TS + col];
                        }
                    }
                }
            }
        }
    }
}

// Entry points by occupancy target over one body.  32 channels, one quadrant per wave, 32-instance chunks: 135 registers and
// 5.7 KB of LDS per wave - squeezed into the four-waves-per-SIMD budget (128) it is the fastest shape at c3 (0.39 ms; 0.45
// at three waves, 0.42-0.43 for two quadrants per wave with 64-instance chunks at three waves: 168 registers, 11.4 KB).
// 64 channels, same shape: 150 registers and 9.7 KB, three waves per SIMD (c4: 2.58 -> 2.23 ms with 64-instance chunks at
// two).  128 channels: 128 accumulator registers per quadrant, two waves per SIMD.
template <int CH, bool BASE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) render_forward_mfma_kernel_w4(FwdArgs a) {
    render_forward_mfma_body<CH, 1, 32, 2, BASE>(a);
}
template <int CH, bool BASE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) render_forward_mfma_kernel_w3(FwdArgs a) {
    render_forward_mfma_body<CH, 1, 32, 2, BASE>(a);
}
template <int CH, bool BASE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) render_forward_mfma_kernel_w2(FwdArgs a) {
    render_forward_mfma_body<CH, 1, 32, 2, BASE>(a);
}

// The 128-channel shape needs 70 KB of LDS per four-wave workgroup (above the 64 KB default limit): the limit is raised once
// per device; where that is refused the caller falls back to 64-channel windows.
template <bool BASE>
bool wide_shape_usable() {
    constexpr int MAX_DEV = 64;
    static std::atomic<int> state[MAX_DEV];      // 0: not asked yet, 1: usable, -1: refused (the ABI is re-entrant across threads)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return false;
    if (state[dev].load(std::memory_order_acquire) == 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_forward_mfma_kernel_w2<128, BASE>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * sizeof(FwdChunkMF<128, 32>)));
        if (e != hipSuccess) (void)hipGetLastError();     // not sticky: the narrower windows take over
        state[dev].store(e == hipSuccess ? 1 : -1, std::memory_order_release);
    }
    return state[dev].load(std::memory_order_acquire) > 0;
}

template <int CH, bool BASE>
void launch_shape(const FwdArgs& a, hipStream_t s) {
    const bool solo = a.solo != 0;
    const size_t lds = (solo ? 1 : 4) * sizeof(FwdChunkMF<CH, 32>);
    const dim3 grid(solo ? 4 * a.gx * a.gy : a.gx * a.gy), block(solo ? 64 : 256);
    if constexpr (CH <= 32) hipLaunchKernelGGL((render_forward_mfma_kernel_w4<CH, BASE>), grid, block, lds, s, a);
    else if constexpr (CH <= 64) hipLaunchKernelGGL((render_forward_mfma_kernel_w3<CH, BASE>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((render_forward_mfma_kernel_w2<CH, BASE>), grid, block, lds, s, a);
}
template <int CH>
void launch_one_mf(const FwdArgs& a, hipStream_t s) {
    // later channel windows of wide features run without the colour / depth half
    if (a.write_base) launch_shape<CH, true>(a, s); else launch_shape<CH, false>(a, s);
}

template <int CH, int PPL>
void launch_one(const FwdArgs& a, hipStream_t s) {
    constexpr int NW = 4 / PPL;
    const bool solo = PPL == 1 && a.solo;
    const size_t lds = (solo ? 1 : NW) * sizeof(FwdChunk<CH>);
    hipLaunchKernelGGL((render_forward_kernel<CH, PPL>), dim3(solo ? 4 * a.gx * a.gy : a.gx * a.gy), dim3(solo ? 64 : 256 / PPL), lds, s, a);
}

}  // namespace

// A call that lists a band of tile rows only (f3dgs_set_tile_band): every output pixel outside the band's rows - background
// colour, zero depth and features, final T = 1, no contributor - in whole-row runs (the blend kernels would write them 32 bytes
// at a time: at c5, C = 128, 2.3 ms for seven eighths of nothing).  Plane 0..2 colour, 3 depth, 4 final T, 5 n_contrib, 6.. features.
struct BandFill {
    float* color; float* depth; float* feat; float* final_T; uint32_t* n_contrib; const float* bg;
};
__global__ void __launch_bounds__(256) fill_outside_band_kernel(BandFill f, size_t HW, size_t lo, size_t hi) {
    const int plane = blockIdx.y;
    float* base;
    float v = 0.f;
    if (plane < 3) { base = f.color + (size_t)plane * HW; v = f.bg[plane]; }
    else if (plane == 3) base = f.depth;
    else if (plane == 4) { base = f.final_T; v = 1.f; }
    else if (plane == 5) base = reinterpret_cast<float*>(f.n_contrib);        // (0.f is the zero word)
    else base = f.feat + (size_t)(plane - 6) * HW;
    const size_t n = HW - (hi - lo), e1 = min(n, ((size_t)blockIdx.x + 1) * 2048);
    for (size_t e = (size_t)blockIdx.x * 2048 + threadIdx.x; e < e1; e += 256) base[e < lo ? e : e + (hi - lo)] = v;
}

void launch_render_forward(const ViewParams& vp, int C, const uint2* ranges_enc, uint2* ranges, const uint32_t* point_list,
                           const SplatRec* rec, const float* feat, float* final_T,
                           uint32_t* n_contrib, float* out_color, float* out_feat, float* out_depth, uint32_t* tile_len,
                           hipStream_t s) {
    FwdArgs a;
    a.ranges_enc = ranges_enc; a.ranges = ranges; a.point_list = point_list; a.rec = rec; a.feat = feat;
    a.bg = vp.bg;
    a.final_T = final_T; a.n_contrib = n_contrib; a.out_color = out_color; a.out_feat = out_feat;
    a.out_depth = out_depth;
    a.W = vp.W; a.H = vp.H; a.gx = vp.gx; a.gy = vp.gy; a.C = C;
    a.solo = options().fwd_solo;
    a.tile_len = tile_len;
    band_perm_params(vp.gx, vp.gy, vp.band0, vp.band1, &a.band_b0, &a.band_tb);
    a.band_r0 = vp.band0; a.band_r1 = vp.band1;
    if (vp.band0 > 0 || vp.band1 < vp.gy) {
        const size_t HW = (size_t)vp.W * vp.H;
        const size_t lo = (size_t)min(vp.H, vp.band0 * TILE) * vp.W, hi = (size_t)min(vp.H, vp.band1 * TILE) * vp.W;
        const size_t n = HW - (hi - lo);
        if (n) {
            const BandFill f = {out_color, out_depth, out_feat, final_T, n_contrib, vp.bg};
            hipLaunchKernelGGL(fill_outside_band_kernel, dim3((unsigned)((n + 2047) / 2048), (unsigned)(6 + C)), dim3(256), 0, s, f, HW, lo, hi);
        }
    }
#ifdef F3DGS_DEV
    a.dev = options().dev;
#else
    a.dev = 0;
#endif
    const bool mf = options().feature_mfma != 0;
    if (C == 0) {
        a.c0 = 0; a.nc = 0; a.write_base = 1;
        launch_one<0, 1>(a, s);      // one quadrant per wave, as every other width (four quadrants per wave: 0.448 vs the 0.287 ms of a 3-channel scene at c2's size)
        return;
    }
    // channel window: 64 channels, or 128 on the matrix pipe when more than 64 remain (every window re-evaluates the
    // blend weights of the whole list: fewer, wider windows - two waves per SIMD, the matrix pipe hides the rest)
    const bool wide_ok = mf && options().fwd_wide != 0 && C > 64 && wide_shape_usable<true>() && wide_shape_usable<false>();
    const int wide = wide_ok ? 128 : 64;
    for (int c0 = 0; c0 < C;) {
        const int win = (C - c0 > 64) ? wide : 64;
        a.c0 = c0; a.nc = min(win, C - c0); a.write_base = (c0 == 0);
        c0 += win;
        // up to four channels: vector pipe; 5..16: the 32-column matrix shape with colour and depth in its spare columns
        if (a.nc <= 4) launch_one<4, 1>(a, s);
        else if (a.nc <= 16) { if (mf) launch_one_mf<16>(a, s); else launch_one<16, 1>(a, s); }
        else if (a.nc <= 32) { if (mf) launch_one_mf<32>(a, s); else launch_one<32, 2>(a, s); }
        else if (a.nc <= 64) { if (mf) launch_one_mf<64>(a, s); else launch_one<64, 1>(a, s); }
        else launch_one_mf<128>(a, s);
    }
}

}  // namespace f3dgs
