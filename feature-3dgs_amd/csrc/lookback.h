// lookback.h - workgroup scan and decoupled look-back helpers shared by binning.hip and preprocess.hip.
#pragma once

#include "common.h"

namespace f3dgs {

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread across a 256-thread workgroup. `sh` holds >= 8 words.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* sh, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t s = sh[k];
        if (k < w) base += s;
    }
    if (total) *total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + inc - v;
}

// ---- decoupled look-back ------------------------------------------------------------------------------
// A status word carries flag and value together (2 + 30 bits), so relaxed agent-scope loads and stores are
// enough: 0 = nothing yet, LB_AGG = this workgroup's own count, LB_INC = inclusive prefix up to and including it.
// Workgroups take their position from a ticket counter (arrival order), which makes "every predecessor is
// resident or finished" true whatever the hardware's dispatch order is - the spin below always terminates.
constexpr uint32_t LB_AGG = 1u << 30, LB_INC = 2u << 30, LB_MASK = (1u << 30) - 1u;

__device__ __forceinline__ uint32_t lb_load(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_store(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Exclusive prefix of `mine` over the virtual workgroups 0..vb-1 of one status column (`stride` words between
// consecutive workgroups' entries); publishes this workgroup's aggregate first, its inclusive prefix afterwards.
// One thread per column (the radix passes: thread d = digit d).  Eight predecessors are requested at a time: the
// loads are independent, only the walk is sequential, and a memory-side round trip costs a microsecond.
__device__ __forceinline__ uint32_t lookback(uint32_t* status, uint32_t stride, uint32_t vb, uint32_t mine) {
    uint32_t* my = status + (size_t)vb * stride;
    if (vb == 0) {
        lb_store(my, LB_INC | mine);
        return 0;
    }
    lb_store(my, LB_AGG | mine);
    uint32_t prefix = 0;
    int j = (int)vb - 1;
    bool done = false;
    while (!done && j >= 0) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = j - k >= 0 ? lb_load(status + (size_t)(j - k) * stride) : LB_INC;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (done) break;
            while ((v[k] >> 30) == 0) v[k] = lb_load(status + (size_t)(j - k) * stride);
            prefix += v[k] & LB_MASK;
            done = (v[k] >> 30) == 2;
        }
        j -= 8;
    }
    lb_store(my, LB_INC | (prefix + mine));
    return prefix;
}

// Single-column flavour for one whole wave (all 64 lanes must call it): lane l inspects predecessor vb-1-l-64w of
// window w, so a window of 64 predecessors costs one round trip.  Returns the exclusive prefix in every lane.
__device__ __forceinline__ uint32_t lookback_wave(uint32_t* status, uint32_t vb, uint32_t mine, int lane) {
    if (vb == 0) {
        if (lane == 0) lb_store(status, LB_INC | mine);
        return 0;
    }
    if (lane == 0) lb_store(status + vb, LB_AGG | mine);
    uint32_t prefix = 0;
    int top = (int)vb - 1;                 // nearest predecessor of the current window
    while (true) {
        const int j = top - lane;
        uint32_t v = j >= 0 ? lb_load(status + j) : LB_INC;       // before the first workgroup: an inclusive 0
        // every lane up to the first inclusive entry must be ready
        unsigned long long inc = __ballot((v >> 30) == 2);
        unsigned long long notready = __ballot((v >> 30) == 0);
        const int first_inc = inc ? __builtin_ctzll(inc) : 64;
        const unsigned long long need = first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1ull);
        if (notready & need) continue;     // poll the same window again
        uint32_t c = (lane <= first_inc) ? (v & LB_MASK) : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d, 64);
        prefix += c;
        if (inc) break;
        top -= 64;
    }
    if (lane == 0) lb_store(status + vb, LB_INC | (prefix + mine));
    return prefix;
}

}  // namespace f3dgs
