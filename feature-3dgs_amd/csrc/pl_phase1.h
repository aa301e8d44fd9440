// pl_phase1.h — phase 1 of the pixel-lane blend backward in its bf16 shape (render_bwd_pl.hip): sixteen list entries against
// the 64 pixels of a quadrant, lane = pixel.  Kept in a header of its own so that tools/ubench/blend_stream.hip times the very
// instruction stream the kernel runs.
//
// Semantics per (entry, pixel): R/cuda_rasterizer/backward.cu:520-617 (R = submodules/diff-gaussian-rasterization-feature) -
// alpha recomputed as in the forward (forward.cu:336-377), T walked back to front, dL/dalpha from the colour / depth sums behind.
//
// What the schedule is about (round 6; measured with tools/ubench/blend_stream.hip): one wave issues at most one instruction per
// ~5 cycles whatever the dependencies, and the compiler placed every broadcast read of a splat record right in front of its first
// use - an LDS latency (64+ cycles, more under load) exposed once per entry.  SCHED selects who orders the block:
//   0  the compiler (round 5: entry after entry, the next record requested "early" in the source, late in the ISA)
//   1  three-stage software pipeline, rows pinned with sched_barrier: while entry i's exponent is evaluated (stage B), entry
//      i - 1 takes its tests, clamp and reciprocal (stage C1) and entry i - 2 its transmittance / colour-behind recurrences, the
//      bf16 split and the stores (stage C2); records are requested a whole iteration ahead, colours eight rows ahead.
#pragma once

#include "render_common.h"

namespace f3dgs {

typedef float p1_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 p1_bf16x2 __attribute__((ext_vector_type(2)));

// One staged list entry (shared by the four waves of a tile).  Phase 1 reads q0 and q1.xy a whole entry ahead and q2 (needed
// last) in the entry's own iteration.
struct PlRec {
    float4 q0;   // mean_x, mean_y, conic_a', conic_b'   (conic pre-scaled, see splat_power2)
    float4 q1;   // conic_c', opacity, Gaussian index (bits), unused
    float4 q2;   // red, green, blue, depth
};

// BF tile layout (32 KB per tile): quadrant q at byte 8192 q; term t (0: w high, 1: w middle, 2: s high, 3: s middle) at
// + 2048 t; row (entry) i at + 128 i; the row's eight 16-byte units = the eight pixel rows of the quadrant, unit y stored at
// slot y ^ (i >> 1); pixel x of the row at + 2 x.
constexpr int BF_QUAD = 8192, BF_TERM = 2048, BF_ROWB = 128;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(p1_f32x2{lo, hi}, p1_bf16x2));
}

// Pixel state of a lane that phase 1 reads but does not change.
struct P1Pixel {
    float pxf, pyf;          // pixel centre
    uint32_t last;           // the pixel's last contributor (1-based list position; 0: none)
    float dR, dG, dB, dD;    // dL/d{red, green, blue, depth} of the pixel (first channel window only)
};

// Sixteen entries rc[0..15] (list positions pos_hi, pos_hi - 1, ...) against this lane's pixel.  T, S: transmittance and
// "colour behind" carried along the walk.  tiles + sofs: this lane's slot in row 0 of the quadrant's w-high plane.
// Returns the rows (bit e = entry e) that blended at some pixel of the wave.
// NOLDS (tools/ubench/blend_stream.hip only): four records read once per chunk stand in for all sixteen (one extra vector
// instruction per entry moves the mean) and the stores are dropped: the vector-pipe stream alone.
// S32 (the hybrid shape of the first window): s leaves as ONE fp32 value - into the quadrant's fp32 tile at s_tile + (s_ofs ^ 16 e),
// the layout of the exact-fp32 shape (render_bwd_pl.hip: PlShared) - instead of two bf16 terms; w keeps its two terms.
template <bool GEO, int SCHED, bool NOLDS = false, bool S32 = false>
__device__ __forceinline__ uint32_t pl_phase1_bf16(const PlRec* rc, const P1Pixel& px, uint32_t pos_hi, float& T, float& S,
                                                   char* tiles, uint32_t sofs, char* s_tile = nullptr, uint32_t s_ofs = 0) {
    static_assert(!S32 || (GEO && !NOLDS), "the fp32 s tile belongs to the first window");
    static_assert(ALPHA_MAX == 0.99f, "literal in the clamp below");
    uint32_t tm = 0;
    struct Ent {
        float4 g;          // mean_x, mean_y, a', b'
        float2 co;         // c', opacity
        float4 col;        // r, g, b, depth
        float dx, dy, t1, t2, pw, G, qd, v, au, al, f;
        unsigned long long m1, m2, m3;
    };
    Ent en[4];             // ring: entry e lives in en[e & 3]
    float Tb = 0.f, nSf = 0.f, wv = 0.f, dLa = 0.f, sv = 0.f, rw = 0.f, rs = 0.f, w_even = 0.f;
    uint32_t h = 0, hl = 0, hh = 0, m = 0;
    (void)nSf; (void)dLa; (void)sv; (void)rs; (void)w_even; (void)hh;

    uint32_t sink = 0;
    (void)sink;
    if constexpr (NOLDS) {
        // four records read once per chunk; an entry re-uses the record of entry e - 4 with its mean moved by one instruction
        // (so that nothing of the earlier evaluation can be re-used)
#pragma unroll
        for (int k = 0; k < 4; k++) { en[k].g = rc[k].q0; en[k].co = *reinterpret_cast<const float2*>(&rc[k].q1); en[k].col = rc[k].q2; }
    }
    auto load_geo = [&](int e) {
        Ent& x = en[e & 3];
        if constexpr (NOLDS) {
            x.g.x += 0.015625f;
        } else {
            x.g = rc[e].q0; x.co = *reinterpret_cast<const float2*>(&rc[e].q1);
        }
    };
    auto load_col = [&](int e) {
        if constexpr (GEO && !NOLDS) en[e & 3].col = rc[e].q2;
    };
    // stage B: exponent and exponential, colour dot product - nothing here depends on the walk
    auto stage_b = [&](int k, int e) {
        Ent& x = en[e & 3];
        switch (k) {
            case 0: x.dx = x.g.x - px.pxf; break;
            case 1: x.dy = x.g.y - px.pyf; break;
            case 2: x.t1 = x.g.w * x.dy; break;
            case 3: x.t2 = x.co.x * x.dy; break;
            case 4: x.t1 = fmaf(x.g.z, x.dx, x.t1); break;
            case 5: x.t2 = x.t2 * x.dy; break;
            case 6: x.pw = fmaf(x.t1, x.dx, x.t2); break;                  // = splat_power2(dx, dy, a', b', c')
            case 7: x.G = __builtin_amdgcn_exp2f(x.pw); break;
            case 8: if constexpr (GEO) x.qd = x.col.w * px.dD; break;
            case 9: if constexpr (GEO) x.qd = fmaf(x.col.z, px.dB, x.qd); break;
            case 10: if constexpr (GEO) x.qd = fmaf(x.col.y, px.dG, x.qd); break;
            case 11: if constexpr (GEO) x.qd = fmaf(x.col.x, px.dR, x.qd); break;
            default: break;
        }
    };
    // stage C1: the three tests as lane masks, clamp, reciprocal of 1 - alpha (exactly 1 for skipped pairs)
    auto stage_c1 = [&](int k, int e) {
        Ent& x = en[e & 3];
        switch (k) {
            case 0: x.v = x.co.y * x.G; break;                                   // op G, not yet clamped
            case 1: x.m1 = __ballot(pos_hi - (uint32_t)e < px.last); break;
            case 2: x.m2 = __ballot(!(x.pw > 0.0f)); break;
            case 3: x.m3 = __ballot(!(x.v < ALPHA_MIN)); break;
            case 5: {
                const unsigned long long okm = x.m1 & x.m2 & x.m3;
                if (okm) tm |= 1u << e;
                // exp2 may be inf where power > 0: selected away, never multiplied.  The clamp sits in the same block: behind an
                // opaque value the compiler puts a canonicalising v_max in front of the fminf
                asm("v_cndmask_b32_e64 %0, 0, %2, %3\n\tv_min_f32_e32 %1, 0x3f7d70a4, %0" : "=&v"(x.au), "=v"(x.al) : "v"(x.v), "s"(okm));
            } break;
            case 7: x.f = 1.f - x.al; break;
            case 8: x.f = __builtin_amdgcn_rcpf(x.f); break;
            default: break;
        }
    };
    // stage C2: the recurrences of the walk, the two-term bf16 split, the stores
    auto stage_c2 = [&](int k, int e) {
        Ent& x = en[e & 3];
        char* const bp = tiles + (sofs ^ (uint32_t)(16 * (e >> 1))) + e * BF_ROWB;
        if constexpr (GEO) {
            switch (k) {
                case 0: Tb = T * x.f; break;                              // transmittance in front of this splat
                case 1: nSf = S * -x.f; break;
                case 2: wv = x.al * Tb; break;
                case 3: dLa = fmaf(Tb, x.qd, nSf); break;                 // dL/dalpha
                case 4: S = fmaf(wv, x.qd, S); break;
                case 5: sv = x.au * dLa; break;
                case 6: h = pack_bf16(wv, S32 ? 0.f : sv); break;         // low half: w high term, high half: s high term
                case 7: hl = h << 16; break;
                case 8: if constexpr (!S32) hh = h & 0xFFFF0000u; break;
                case 9: rw = fmaf(x.al, Tb, -__uint_as_float(hl)); break; // residual of the unrounded product
                case 10: if constexpr (!S32) rs = fmaf(x.au, dLa, -__uint_as_float(hh)); break;
                case 11: m = pack_bf16(rw, S32 ? 0.f : rs); break;
                case 12:
                    if constexpr (S32) {
                        *reinterpret_cast<uint16_t*>(bp) = (uint16_t)h;
                        *reinterpret_cast<uint16_t*>(bp + BF_TERM) = (uint16_t)m;
                        *reinterpret_cast<float*>(s_tile + (s_ofs ^ (uint32_t)(16 * e))) = sv;
                    } else if constexpr (NOLDS) {
                        asm volatile("" : "+v"(h), "+v"(m));      // (the values stay alive; nothing is stored)
                        sink = h;
                    } else {
                        *reinterpret_cast<uint16_t*>(bp) = (uint16_t)h;
                        *reinterpret_cast<uint16_t*>(bp + BF_TERM) = (uint16_t)m;
                        *reinterpret_cast<uint16_t*>(bp + 2 * BF_TERM) = (uint16_t)(h >> 16);
                        *reinterpret_cast<uint16_t*>(bp + 3 * BF_TERM) = (uint16_t)(m >> 16);
                    }
                    T = Tb;
                    break;
                default: break;
            }
        } else {
            // later channel windows: the weight only; two entries share a conversion (rows e - 1 and e share their XOR term)
            switch (k) {
                case 0: Tb = T * x.f; break;
                case 2: wv = x.al * Tb; break;
                case 6: if (e & 1) h = pack_bf16(w_even, wv); break;
                case 7: if (e & 1) hl = h << 16; break;
                case 8: if (e & 1) hh = h & 0xFFFF0000u; break;
                case 9: if (e & 1) rw = w_even - __uint_as_float(hl); break;
                case 10: if (e & 1) rs = wv - __uint_as_float(hh); break;
                case 11: if (e & 1) m = pack_bf16(rw, rs); break;
                case 12:
                    if constexpr (NOLDS) {
                        if (e & 1) { asm volatile("" : "+v"(h), "+v"(m)); sink = h; } else w_even = wv;
                    } else if (e & 1) {
                        *reinterpret_cast<uint16_t*>(bp - BF_ROWB) = (uint16_t)h;
                        *reinterpret_cast<uint16_t*>(bp - BF_ROWB + BF_TERM) = (uint16_t)m;
                        *reinterpret_cast<uint16_t*>(bp) = (uint16_t)(h >> 16);
                        *reinterpret_cast<uint16_t*>(bp + BF_TERM) = (uint16_t)(m >> 16);
                    } else {
                        w_even = wv;
                    }
                    T = Tb;
                    break;
                default: break;
            }
        }
    };
    constexpr int ROWS = 13;

    if constexpr (SCHED == 0) {
        // the compiler's order: entry after entry, the next record requested at the top of the entry
        load_geo(0); load_col(0);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            if (e + 1 < 16) { load_geo(e + 1); load_col(e + 1); }
#pragma unroll
            for (int k = 0; k < ROWS; k++) stage_b(k, e);
#pragma unroll
            for (int k = 0; k < ROWS; k++) stage_c1(k, e);
#pragma unroll
            for (int k = 0; k < ROWS; k++) stage_c2(k, e);
        }
    } else {
        load_geo(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16 + 2; i++) {
            if (i + 1 < 16) load_geo(i + 1);
            if (i < 16) load_col(i);
#pragma unroll
            for (int k = 0; k < ROWS; k++) {
                if (i < 16) stage_b(k, i);
                if (i >= 1 && i - 1 < 16) stage_c1(k, i - 1);
                if (i >= 2) stage_c2(k, i - 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (NOLDS) tm ^= __builtin_amdgcn_readfirstlane((int)sink) & 0x10000;
    return tm;
}

}  // namespace f3dgs
