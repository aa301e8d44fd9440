// binning.hip — scan, stable LSD radix sort, instance emission and tile ranges.
//
// What the reference does (R/cuda_rasterizer/rasterizer_impl.cu:279-320): prefix-sum tiles_touched,
// emit one (tile<<32 | depth_bits, gaussian_id) pair per overlapped tile in Gaussian-index order,
// stable-sort all N pairs on the 64-bit key with cub (6 passes at 1080p), find per-tile ranges.
//
// What this file does instead (same resulting order, far less traffic):
//   1. stable sort of the P Gaussians by their 32-bit depth key (4 passes over P pairs);
//   2. prefix-sum of tiles_touched taken IN DEPTH ORDER;
//   3. emit instances (tile, id) in depth order;
//   4. stable sort of the N instances by tile id only (2 passes at <= 65536 tiles).
// An LSD radix sort is "sort by low key, then stably by high key": steps 1+4 are exactly the
// reference's sort on (tile | depth) with ties broken by ascending Gaussian id, because the depth
// sort starts from id order and both sorts are stable (Q6 in SURVEY.md section 8a).
//
// Radix pass = 3 launches: per-workgroup digit histogram, per-digit row scan (grid = 256),
// stable scatter (wave-level match with ballots, wave-ordered LDS counters).

#include "common.h"

namespace f3dgs {

namespace {

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

// ---------------- generic exclusive scan (reduce / spine / apply) ------------------------------------
__global__ void __launch_bounds__(256) scan_reduce_kernel(const uint32_t* __restrict__ in,
                                                          const uint32_t* __restrict__ gather, size_t n,
                                                          uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t sh[8];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        if (i < n) acc += gather ? in[gather[i]] : in[i];
    }
    uint32_t tot;
    block_excl_scan_256(acc, sh, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256) scan_spine_kernel(uint32_t* __restrict__ block_sums, size_t nb,
                                                         uint32_t* __restrict__ total,
                                                         const uint32_t* __restrict__ extra, size_t n_extra,
                                                         uint32_t* __restrict__ extra_total) {
    __shared__ uint32_t sh[8];
    if (extra) {
        uint32_t acc = 0;
        for (size_t i = threadIdx.x; i < n_extra; i += 256) acc += extra[i];
        uint32_t tot;
        block_excl_scan_256(acc, sh, &tot);
        if (threadIdx.x == 0) *extra_total = tot;
    }
    uint32_t carry = 0;
    for (size_t base = 0; base < nb; base += 256) {
        const size_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? block_sums[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256(v, sh, &tot);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ void __launch_bounds__(256) scan_apply_kernel(const uint32_t* __restrict__ in,
                                                         const uint32_t* __restrict__ gather, size_t n,
                                                         const uint32_t* __restrict__ block_sums,
                                                         uint32_t* __restrict__ out) {
    __shared__ uint32_t sh[8];
    // thread t owns 16 consecutive items so the running order is preserved
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * 16;
    uint32_t v[16];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const size_t i = base + k;
        v[k] = i < n ? (gather ? in[gather[i]] : in[i]) : 0;
        acc += v[k];
    }
    uint32_t ex = block_excl_scan_256(acc, sh, nullptr) + block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const size_t i = base + k;
        if (i < n) out[i] = ex;
        ex += v[k];
    }
}

// ---------------- radix sort pass -------------------------------------------------------------------
// Item order inside a workgroup chunk: wave w owns [w*1024, (w+1)*1024), visited in 16 steps of 64
// consecutive items (lane = item within the step).  BITS = digit width (8 for tile ids, 11 for the 32-bit
// depth keys: 3 passes instead of 4).  vals_in == nullptr means "value = item index" (first pass of the
// depth sort: saves an iota kernel and a key copy).
template <int BITS>
__global__ void __launch_bounds__(SORT_THREADS) radix_hist_kernel(const uint32_t* __restrict__ keys, size_t n,
                                                                  int shift, uint32_t nb,
                                                                  uint32_t* __restrict__ hist) {
    constexpr int BINS = 1 << BITS;
    __shared__ uint32_t h[BINS];
    for (int d = threadIdx.x; d < BINS; d += SORT_THREADS) h[d] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_CHUNK;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const size_t i = base + (size_t)k * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & (BINS - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < BINS; d += SORT_THREADS) hist[(size_t)d * nb + blockIdx.x] = h[d];  // digit-major
}

// One workgroup per digit: exclusive scan of that digit's per-workgroup counts, total to totals[d].
__global__ void __launch_bounds__(256) radix_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t nb,
                                                            uint32_t* __restrict__ totals) {
    __shared__ uint32_t sh[8];
    uint32_t* row = hist + (size_t)blockIdx.x * nb;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? row[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256(v, sh, &tot);
        if (i < nb) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <int BITS, bool REORDER>
__global__ void __launch_bounds__(SORT_THREADS)
radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n, int shift,
                     uint32_t nb, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals) {
    constexpr int BINS = 1 << BITS;
    constexpr int BPT = BINS / SORT_THREADS;   // bins per thread in the offset phase
    __shared__ uint32_t cnt[4][BINS];
    __shared__ uint32_t sh[8];
    // REORDER: the chunk is first sorted inside LDS so that the global stores of a wave run over consecutive
    // addresses per digit (16 items per digit on average) instead of 64 unrelated 4-byte scatters.
    __shared__ uint32_t gbase[REORDER ? BINS : 1];     // global offset of this workgroup's run of digit d
    __shared__ uint32_t lstart[REORDER ? BINS : 1];    // start of digit d inside the LDS-sorted chunk
    __shared__ uint32_t skey[REORDER ? SORT_CHUNK : 1];
    __shared__ uint32_t sval[REORDER ? SORT_CHUNK : 1];
    volatile uint32_t* vcnt = &cnt[0][0];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < 4 * BINS; d += SORT_THREADS) (&cnt[0][0])[d] = 0;
    __syncthreads();

    const size_t wbase = (size_t)blockIdx.x * SORT_CHUNK + (size_t)w * (SORT_ITEMS * 64);
    uint32_t key[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const size_t i = wbase + (size_t)k * 64 + lane;
        const bool valid = i < n;
        key[k] = valid ? keys_in[i] : 0xFFFFFFFFu;
        const uint32_t d = (key[k] >> shift) & (BINS - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = __popcll(peers & lt_mask);
        const uint32_t old = vcnt[w * BINS + d];
        if (valid && before == 0) vcnt[w * BINS + d] = old + (uint32_t)__popcll(peers);
        rank[k] = old + before;
    }
    __syncthreads();
    // thread t owns digits [t*BPT, (t+1)*BPT): turn per-wave counts into offsets
    {
        uint32_t tot[BPT], run = 0, blk[BPT], brun = 0;
#pragma unroll
        for (int q = 0; q < BPT; q++) {
            const uint32_t d = threadIdx.x * BPT + q;
            tot[q] = totals[d];
            run += tot[q];
            blk[q] = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
            brun += blk[q];
        }
        uint32_t digit_base = block_excl_scan_256(run, sh, nullptr);
        uint32_t local_base = REORDER ? block_excl_scan_256(brun, sh, nullptr) : 0;
#pragma unroll
        for (int q = 0; q < BPT; q++) {
            const uint32_t d = threadIdx.x * BPT + q;
            const uint32_t g = digit_base + hist[(size_t)d * nb + blockIdx.x];
            if (REORDER) { gbase[d] = g; lstart[d] = local_base; }
            uint32_t r2 = REORDER ? local_base : g;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t c = cnt[k][d];
                cnt[k][d] = r2;
                r2 += c;
            }
            digit_base += tot[q];
            local_base += blk[q];
        }
    }
    __syncthreads();
    if (REORDER) {
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const size_t i = wbase + (size_t)k * 64 + lane;
            if (i < n) {
                const uint32_t d = (key[k] >> shift) & (BINS - 1);
                const uint32_t lp = cnt[w][d] + rank[k];
                skey[lp] = key[k];
                sval[lp] = vals_in ? vals_in[i] : (uint32_t)i;
            }
        }
        __syncthreads();
        const size_t cbase = (size_t)blockIdx.x * SORT_CHUNK;
        const uint32_t have = (uint32_t)min((size_t)SORT_CHUNK, n - cbase);
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t j = (uint32_t)k * SORT_THREADS + threadIdx.x;
            if (j < have) {
                const uint32_t kk = skey[j];
                const uint32_t d = (kk >> shift) & (BINS - 1);
                const uint32_t pos = gbase[d] + (j - lstart[d]);
                keys_out[pos] = kk;
                vals_out[pos] = sval[j];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const size_t i = wbase + (size_t)k * 64 + lane;
            if (i < n) {
                const uint32_t d = (key[k] >> shift) & (BINS - 1);
                const uint32_t pos = cnt[w][d] + rank[k];
                keys_out[pos] = key[k];
                vals_out[pos] = vals_in ? vals_in[i] : (uint32_t)i;
            }
        }
    }
}

__global__ void __launch_bounds__(256) iota_kernel(uint32_t* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) tile_ranges_kernel(size_t N, const uint32_t* __restrict__ tile_sorted,
                                                          uint2* __restrict__ ranges) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t cur = tile_sorted[i];
    if (i == 0) {
        ranges[cur].x = 0;
    } else {
        const uint32_t prev = tile_sorted[i - 1];
        if (cur != prev) {
            ranges[prev].y = (uint32_t)i;
            ranges[cur].x = (uint32_t)i;
        }
    }
    if (i == N - 1) ranges[cur].y = (uint32_t)N;
}

}  // namespace

void launch_exclusive_scan(const uint32_t* in, const uint32_t* gather, uint32_t* out, uint32_t* total, size_t n,
                           uint32_t* tmp, const uint32_t* extra, size_t n_extra, uint32_t* extra_total, hipStream_t s) {
    const size_t nb = scan_blocks(n);
    if (nb == 0) return;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(256), 0, s, in, gather, n, tmp);
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(256), 0, s, tmp, nb, total, extra, n_extra, extra_total);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, s, in, gather, n, tmp, out);
}

template <int BITS>
static void radix_pass(const uint32_t* ki, const uint32_t* vi, uint32_t* ko, uint32_t* vo, size_t n, int shift,
                       uint32_t nb, uint32_t* hist, hipStream_t s) {
    constexpr int BINS = 1 << BITS;
    uint32_t* totals = hist + (size_t)BINS * nb;
    hipLaunchKernelGGL((radix_hist_kernel<BITS>), dim3(nb), dim3(SORT_THREADS), 0, s, ki, n, shift, nb, hist);
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(BINS), dim3(256), 0, s, hist, nb, totals);
    hipLaunchKernelGGL((radix_scatter_kernel<BITS, (BITS <= 8)>), dim3(nb), dim3(SORT_THREADS), 0, s, ki, vi, ko, vo, n,
                       shift, nb, hist, totals);
}

void launch_radix_sort_pairs(uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b, size_t n, int nbits,
                             uint32_t* hist, bool result_in_a, hipStream_t s) {
    // Input is expected in A when (#passes even) == result_in_a, else in B; the caller arranges that.
    const int passes = (nbits + RADIX_BITS - 1) / RADIX_BITS;
    if (n == 0 || passes == 0) return;
    const uint32_t nb = (uint32_t)sort_blocks(n);
    bool in_a = (passes % 2 == 0) ? result_in_a : !result_in_a;
    for (int p = 0; p < passes; p++) {
        uint32_t* ki = in_a ? key_a : key_b;
        uint32_t* vi = in_a ? val_a : val_b;
        uint32_t* ko = in_a ? key_b : key_a;
        uint32_t* vo = in_a ? val_b : val_a;
        radix_pass<RADIX_BITS>(ki, vi, ko, vo, n, p * RADIX_BITS, nb, hist, s);
        in_a = !in_a;
    }
}

// Depth sort of the Gaussians: 32-bit keys in `keys` (read-only), values = indices.  Three passes of 11/11/10
// bits; the sorted ids end in val_a (and the sorted keys in key_a).
void launch_depth_sort(const uint32_t* keys, uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b, size_t n,
                       uint32_t* hist, hipStream_t s) {
    if (n == 0) return;
    const uint32_t nb = (uint32_t)sort_blocks(n);
    if (n > 200000) {
        // large P: four 8-bit passes are faster than three 11-bit ones (measured at 1M: 0.105 vs 0.148 ms);
        // small P is launch-bound and prefers fewer passes.  Result must end in (key_a, val_a): A <- keys, then
        // A -> B -> A -> ... needs an odd number of remaining hops, so the first pass writes into B.
        radix_pass<RADIX_BITS>(keys, nullptr, key_b, val_b, n, 0, nb, hist, s);
        radix_pass<RADIX_BITS>(key_b, val_b, key_a, val_a, n, 8, nb, hist, s);
        radix_pass<RADIX_BITS>(key_a, val_a, key_b, val_b, n, 16, nb, hist, s);
        radix_pass<RADIX_BITS>(key_b, val_b, key_a, val_a, n, 24, nb, hist, s);
        return;
    }
    radix_pass<DEPTH_RADIX_BITS>(keys, nullptr, key_a, val_a, n, 0, nb, hist, s);
    radix_pass<DEPTH_RADIX_BITS>(key_a, val_a, key_b, val_b, n, DEPTH_RADIX_BITS, nb, hist, s);
    radix_pass<DEPTH_RADIX_BITS>(key_b, val_b, key_a, val_a, n, 2 * DEPTH_RADIX_BITS, nb, hist, s);
}

void launch_iota(uint32_t* dst, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, n);
}

void launch_tile_ranges(size_t N, const uint32_t* tile_sorted, uint2* ranges, size_t tiles, hipStream_t s) {
    (void)hipMemsetAsync(ranges, 0, tiles * sizeof(uint2), s);
    if (N == 0) return;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, tile_sorted, ranges);
}

}  // namespace f3dgs
