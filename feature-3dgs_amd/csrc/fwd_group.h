// fwd_group.h — the blend forward's inner step on the matrix-pipe path (render_fwd.hip, render_forward_mfma_body): one group
// of GI consecutive staged list entries against the wave's PPL x 64 pixels - alpha evaluation, the transmittance walk, colour
// and depth on the vector pipe, the feature contraction on v_mfma_f32_32x32x2_f32.  Kept in a header of its own so that
// tools/ubench/blend_stream.hip times the very instruction stream the kernel runs.
//
// Semantics per (entry, pixel): R/cuda_rasterizer/forward.cu:336-377 (R = submodules/diff-gaussian-rasterization-feature),
// quirks Q4 and Q5 as in render_fwd.hip.
#pragma once

#include "render_common.h"

namespace f3dgs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One staged (compacted) list entry: a single LDS address per instance, three broadcast reads.
struct FwdEntry {
    float4 geo;        // mean_x, mean_y, conic_a, conic_b
    float4 cd;         // r, g, b, depth
    float co_c, co_o;  // conic_c, opacity
    uint32_t pos;      // 1-based list position: n_contrib bookkeeping
    uint32_t id;       // Gaussian index (feature row)
};
static_assert(sizeof(FwdEntry) == 48, "FwdEntry layout");

// Per-lane state of the wave's pixels (lane l = pixel (l & 7, l >> 3) of each of its PPL quadrants).
template <int CH, int PPL>
struct FwdPixels {
    static constexpr int NB = (CH + 31) / 32;
    float pxf[PPL], pyf[PPL];
    // T carries the pixel's "finished" flag in its sign bit (T > 0 while the pixel is still blending; -T_final afterwards)
    float T[PPL], col[PPL][3], dep[PPL];
    uint32_t last[PPL];
    f32x16 acc[PPL][2][NB];
};

// Entries ent[j .. j + GI) (a count that is not a multiple of GI is padded with null entries by the caller); feat: the chunk's
// feature rows, row-major [entry][CH]; chunk_base / b_stride / b_off: where a lane's B column lives when CH == 16 (see the caller).
template <int CH, int PPL, int GI, bool BASE>
__device__ __forceinline__ void fwd_blend_group(const FwdEntry* ent, const float* feat, const float* chunk_base, int j, int lane,
                                                int b_stride, int b_off, bool skip_mfma, FwdPixels<CH, PPL>& px) {
    constexpr int NB = (CH + 31) / 32;
    constexpr bool CDB = BASE && CH == 16;
    constexpr int NP = GI / 2;        // instance pairs (MFMA K = 2) per group
    auto& pxf = px.pxf; auto& pyf = px.pyf; auto& T = px.T; auto& col = px.col; auto& dep = px.dep; auto& last = px.last;
    auto& acc = px.acc;
    float4 g0[GI], cdv[GI];
    float2 g1[GI];
    uint32_t pos_e[GI];
    float Bv[NP][NB];
#pragma unroll
    for (int e = 0; e < GI; e++) {
        const int je = j + e;
        g0[e] = ent[je].geo;
        if constexpr (BASE && !CDB) cdv[e] = ent[je].cd;
        const float4 tail = *reinterpret_cast<const float4*>(&ent[je].co_c);
        g1[e] = make_float2(tail.x, tail.y);
        pos_e[e] = __float_as_uint(tail.z);
    }
#pragma unroll
    for (int k = 0; k < NP; k++) {
        // B rows: instance j+2k for lanes 0-31, j+2k+1 for lanes 32-63 (a missing instance is a null entry: w = 0
        // against a row of zeros)
        const int e0 = 2 * k;
        const int rsel = j + e0 + (lane >> 5);
        if constexpr (CH == 16) {
            Bv[k][0] = chunk_base[rsel * b_stride + b_off];
        } else {
#pragma unroll
            for (int nb = 0; nb < NB; nb++) Bv[k][nb] = feat[rsel * CH + (lane & 31) + 32 * nb];
        }
    }
    // a quadrant whose 64 pixels are all saturated is skipped as a whole (wave-uniform branch)
    bool slot_live[PPL];
#pragma unroll
    for (int p = 0; p < PPL; p++) slot_live[p] = BASE ? __any(T[p] > 0.0f) : true;
    float w[GI][PPL];
    unsigned long long blend_mask = 0ull;       // (a wave-uniform mask, not a per-lane flag: the flag would cross the
                                                //  slot_live branch as a 0/1 register and be compared again)
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        if (!slot_live[p]) {
#pragma unroll
            for (int e = 0; e < GI; e++) w[e][p] = 0.0f;
            continue;
        }
        float araw[GI];
        bool valid[GI];
#pragma unroll
        for (int e = 0; e < GI; e++) {
            const float dx = g0[e].x - pxf[p], dy = g0[e].y - pyf[p];
            const float power = splat_power2(dx, dy, g0[e].z, g0[e].w, g1[e].x);
            araw[e] = fminf(ALPHA_MAX, g1[e].y * __builtin_amdgcn_exp2f(power));
            valid[e] = !(power > 0.0f) && !(araw[e] < ALPHA_MIN);
            if constexpr (!BASE) valid[e] = valid[e] && pos_e[e] <= last[p];
        }
#pragma unroll
        for (int e = 0; e < GI; e++) {
            const float test_T = T[p] * (1.0f - araw[e]);      // negative once the pixel is finished
            const bool below = BASE && test_T < T_MIN;
            const bool ok = valid[e] & !below;
            const bool term = valid[e] & below;                 // (re-)marks finished pixels
            const float wv = ok ? araw[e] * T[p] : 0.0f;
            w[e][p] = wv;
            if constexpr (BASE) T[p] = ok ? test_T : (term ? -fabsf(T[p]) : T[p]);
            else T[p] = ok ? test_T : T[p];
            if constexpr (BASE) {
                last[p] = ok ? pos_e[e] : last[p];
                if constexpr (!CDB) {
                    col[p][0] = fmaf(cdv[e].x, wv, col[p][0]);
                    col[p][1] = fmaf(cdv[e].y, wv, col[p][1]);
                    col[p][2] = fmaf(cdv[e].z, wv, col[p][2]);
                    dep[p] = fmaf(cdv[e].w, wv, dep[p]);
                }
            }
        }
        // any weight of the pixel non-zero?  The weights are >= 0: one compare per entry pair (w0 != -w1), and a compare is
        // what the ballot wants to see - from a combination of lane masks it goes through a 0/1 register and a second compare
        bool nz = w[0][p] != -w[1][p];
#pragma unroll
        for (int e = 2; e < GI; e += 2) nz = nz || (w[e][p] != -w[e + 1][p]);
        blend_mask |= __builtin_amdgcn_ballot_w64(nz);
    }
    if (blend_mask != 0ull && !skip_mfma) {
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int e0 = 2 * k;
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(w[e0][p]),
                                                                 __float_as_int(w[e0 + 1][p]), false, false);
                const float X = __int_as_float(sw[0]), Y = __int_as_float(sw[1]);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    acc[p][0][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(X, Bv[k][nb], acc[p][0][nb], 0, 0, 0);
                    acc[p][1][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(Y, Bv[k][nb], acc[p][1][nb], 0, 0, 0);
                }
            }
        }
    }
}

}  // namespace f3dgs
