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

#include <atomic>
#include "common.h"
#include "lookback.h"

namespace f3dgs {

namespace {

// ---------------- list offsets in depth order: two-level sums ------------------------------------------
// Two levels are kept: the sum of every run of 64 consecutive items (`sub`, 64 per workgroup chunk - one run is what
// one wave of the emit kernel owns) and the chunk totals.  An emit wave rebuilds its offset from the two (the totals of
// the chunks in front of its own, one 256-byte read of its chunk's runs, a wave scan), so no per-item offset array is
// written or read and the scan is ONE launch.
__global__ void __launch_bounds__(256) scan_reduce_kernel(const uint32_t* __restrict__ in,
                                                          const uint32_t* __restrict__ gather, size_t n,
                                                          uint32_t* __restrict__ block_sums, uint32_t* __restrict__ sub) {
    __shared__ uint32_t sh[8];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        v[k] = i < n ? (gather ? in[gather[i]] : in[i]) : 0u;
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t r = v[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) r += (uint32_t)__shfl_xor((int)r, d, 64);
        if (lane == 0) sub[(size_t)blockIdx.x * 64 + 4 * k + w] = r;       // items base + 256 k + 64 w .. + 63
        acc += v[k];
    }
    uint32_t tot;
    block_excl_scan_256(acc, sh, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}


// ---------------- radix sort pass -------------------------------------------------------------------
// Item order inside a workgroup chunk: wave w owns [w*1024, (w+1)*1024), visited in 16 steps of 64
// consecutive items (lane = item within the step).  BITS = digit width (8 for tile ids, 11 for the 32-bit
// depth keys: 3 passes instead of 4).  vals_in == nullptr means "value = item index" (first pass of the
// depth sort: saves an iota kernel and a key copy).
// `tj` (first pass of the depth sort only): workgroup 0 also adds up the per-workgroup instance counts of
// preprocess_kernel - both totals are known right after that kernel - and stores them to the device counters and
// straight into the host's pinned read-back words: no totals launch and no copy launch in front of the sort.
// `tj`: the per-workgroup instance counts of preprocess_kernel added up by ONE workgroup (all its THREADS threads call this) and
// stored to the device counters and straight into the host's pinned read-back words.
template <int THREADS>
__device__ __forceinline__ void run_totals_job(const TotalsJob& tj) {
    __shared__ uint32_t shc[3][THREADS / 64];
    uint32_t v = 0, u = 0, ar = 0;
    for (int i = threadIdx.x; i < tj.n_partial; i += THREADS) {
        v += tj.partial[i]; u += tj.partial[tj.n_partial + i]; ar = max(ar, tj.partial[2 * tj.n_partial + i]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += (uint32_t)__shfl_xor((int)v, d, 64);
        u += (uint32_t)__shfl_xor((int)u, d, 64);
        ar = max(ar, (uint32_t)__shfl_xor((int)ar, d, 64));
    }
    if ((threadIdx.x & 63) == 0) { shc[0][threadIdx.x >> 6] = v; shc[1][threadIdx.x >> 6] = u; shc[2][threadIdx.x >> 6] = ar; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ref = 0, own = 0, arm = 0;
        for (int k = 0; k < THREADS / 64; k++) { ref += shc[0][k]; own += shc[1][k]; arm = max(arm, shc[2][k]); }
        // [0] entries of our lists, [1] the reference's count, [2] != 0: some visible Gaussian has a long axis (preprocess_kernel)
        tj.counters[0] = own; tj.counters[1] = ref; tj.counters[2] = arm; tj.counters[3] = 0u;    // ([3]: the emit kernel, later in the frame)
        if (tj.host) {
            __hip_atomic_store(&tj.host[0], own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&tj.host[1], ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&tj.host[2], arm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
        }
    }
}

template <int BITS, int ITEMS>
__global__ void __launch_bounds__(SORT_THREADS) radix_hist_kernel(const uint32_t* __restrict__ keys, size_t n,
                                                                  int shift, uint32_t nb,
                                                                  uint32_t* __restrict__ hist, TotalsJob tj,
                                                                  const uint32_t* __restrict__ n_dev) {
    constexpr int BINS = 1 << BITS;
    constexpr int CHUNK = SORT_THREADS * ITEMS;
    __shared__ uint32_t h[BINS];
    // n_dev (sync-free forward): the launch is sized by a CAPACITY n, the item count is the device's word; workgroups behind the
    // last item write their zero column of the histogram matrix and nothing else.  A count above the capacity reads as ZERO: the
    // emit kernel stored nothing for such a frame (the arrays hold what the allocator left there), the tile ranges stay empty
    if (n_dev) { const size_t nd = *n_dev; n = nd <= n ? nd : 0; }     // (a frame that does not fit is void: nothing was emitted)
    if (tj.partial && blockIdx.x == 0) run_totals_job<SORT_THREADS>(tj);
    for (int d = threadIdx.x; d < BINS; d += SORT_THREADS) h[d] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * CHUNK;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
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

// RANGES (final tile pass only, needs REORDER): the keys are whole tile ids and the output is fully sorted, so the
// range of a tile is [min, max + 1) of the output positions of its entries.  Inside the LDS-sorted chunk the entries
// of one tile are adjacent and so are their output positions: two atomicMin per (workgroup, tile) segment on the
// all-ones preset of BinState::ranges_enc - a few dozen per workgroup, spread over all tiles.  This replaces a
// separate pass over the sorted keys.
__device__ __forceinline__ void record_range(uint2* ranges_enc, uint32_t tile, bool first, bool last, uint32_t pos) {
    if (first) atomicMin(&ranges_enc[tile].x, pos);
    if (last) atomicMin(&ranges_enc[tile].y, 0xFFFFFFFFu - (pos + 1u));
}

template <int BITS, int ITEMS, bool REORDER, bool RANGES>
__global__ void __launch_bounds__(SORT_THREADS)
radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n, int shift,
                     uint32_t nb, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals,
                     uint2* __restrict__ ranges_enc, const uint32_t* __restrict__ n_dev) {
    constexpr int BINS = 1 << BITS;
    constexpr int CHUNK = SORT_THREADS * ITEMS;
    constexpr int BPT = BINS / SORT_THREADS;   // bins per thread in the offset phase
    if (n_dev) {        // (see radix_hist_kernel) a launch sized by a capacity: the workgroups behind the last item have nothing to move
        const size_t nd = *n_dev;
        n = nd <= n ? nd : 0;
        if ((size_t)blockIdx.x * CHUNK >= n) return;
    }
    __shared__ uint32_t cnt[4][BINS];
    __shared__ uint32_t sh[8];
    // REORDER: the chunk is first sorted inside LDS so that the global stores of a wave run over consecutive
    // addresses per digit (16 items per digit on average) instead of 64 unrelated 4-byte scatters.
    __shared__ uint32_t gbase[REORDER ? BINS : 1];     // global offset of this workgroup's run of digit d
    __shared__ uint32_t lstart[REORDER ? BINS : 1];    // start of digit d inside the LDS-sorted chunk
    __shared__ uint32_t skey[REORDER ? CHUNK : 1];
    __shared__ uint32_t sval[REORDER ? CHUNK : 1];
    volatile uint32_t* vcnt = &cnt[0][0];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < 4 * BINS; d += SORT_THREADS) (&cnt[0][0])[d] = 0;
    __syncthreads();

    const size_t wbase = (size_t)blockIdx.x * CHUNK + (size_t)w * (ITEMS * 64);
    uint32_t key[ITEMS];
    uint32_t val[REORDER ? ITEMS : 1];
    uint32_t rank[ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // every request of the chunk goes out first (the values are only needed by the LDS reorder: their latency hides
    // behind the ranking)
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const size_t i = wbase + (size_t)k * 64 + lane;
        key[k] = i < n ? keys_in[i] : 0xFFFFFFFFu;
        if constexpr (REORDER) val[k] = (vals_in && i < n) ? vals_in[i] : (uint32_t)i;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const size_t i = wbase + (size_t)k * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = (key[k] >> shift) & (BINS - 1);
        const unsigned long long vm = __ballot(valid);
        uint32_t p_lo = (uint32_t)vm, p_hi = (uint32_t)(vm >> 32);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            // x: all ones where the lane's digit has bit b; a peer's bit equals it: peers &= ~(ballot ^ x) = ~ballot ^ x
            const uint32_t x = 0u - ((d >> b) & 1u);
            const unsigned long long nm = ~__ballot(x != 0u);
            p_lo &= (uint32_t)nm ^ x;
            p_hi &= (uint32_t)(nm >> 32) ^ x;
        }
        const unsigned long long peers = ((unsigned long long)p_hi << 32) | p_lo;
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
        for (int k = 0; k < ITEMS; k++) {
            const size_t i = wbase + (size_t)k * 64 + lane;
            if (i < n) {
                const uint32_t d = (key[k] >> shift) & (BINS - 1);
                const uint32_t lp = cnt[w][d] + rank[k];
                skey[lp] = key[k];
                if constexpr (REORDER) sval[lp] = val[k];
            }
        }
        __syncthreads();
        const size_t cbase = (size_t)blockIdx.x * CHUNK;
        const uint32_t have = (uint32_t)min((size_t)CHUNK, n - cbase);
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t j = (uint32_t)k * SORT_THREADS + threadIdx.x;
            if (j < have) {
                const uint32_t kk = skey[j];
                const uint32_t d = (kk >> shift) & (BINS - 1);
                const uint32_t pos = gbase[d] + (j - lstart[d]);
                keys_out[pos] = kk;
                vals_out[pos] = sval[j];
                if (RANGES) record_range(ranges_enc, kk, j == 0 || skey[j - 1] != kk, j + 1 == have || skey[j + 1] != kk, pos);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
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


// =====================================================================================================
// Single-pass ("onesweep") radix passes with decoupled look-back.
//
// One launch per 8-bit digit.  The global digit histograms of ALL passes of a sort are produced up front in
// a single read of the keys (depth keys: sort_prologue_kernel; tile ids: derived from the per-tile counts by
// the last workgroup of emit_scan_kernel), so a pass is: take a ticket (virtual workgroup id = arrival order,
// which makes "every predecessor is resident or done" true whatever the hardware's dispatch order is), rank the
// chunk's items per digit, publish the chunk's 256 digit counts, look back over the predecessors' published
// counts (one thread per digit) until an inclusive prefix is found, scatter through an LDS-sorted image of
// the chunk (helpers: lookback.h).
template <int ITEMS, bool WRITE_KEYS, bool RANGES>
__global__ void __launch_bounds__(SORT_THREADS)
onesweep_pass_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n, int shift,
                     const uint32_t* __restrict__ digit_totals, uint32_t* __restrict__ status,
                     uint32_t* __restrict__ ticket, uint2* __restrict__ ranges) {
    constexpr int BINS = 256;
    constexpr int CHUNK = SORT_THREADS * ITEMS;
    __shared__ uint32_t cnt[4][BINS];
    __shared__ uint32_t sh[8];
    __shared__ uint32_t gbase[BINS];     // global offset of this workgroup's run of digit d
    __shared__ uint32_t lstart[BINS];    // start of digit d inside the LDS-sorted chunk
    __shared__ uint32_t skey[CHUNK];
    __shared__ uint32_t sval[CHUNK];
    __shared__ uint32_t s_vb;
    volatile uint32_t* vcnt = &cnt[0][0];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_vb = atomicAdd(ticket, 1u);
    for (int d = threadIdx.x; d < 4 * BINS; d += SORT_THREADS) (&cnt[0][0])[d] = 0;
    __syncthreads();
    const uint32_t vb = s_vb;

    const size_t cbase = (size_t)vb * CHUNK;
    const size_t wbase = cbase + (size_t)w * (ITEMS * 64);
    uint32_t key[ITEMS];
    uint32_t rank[ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const size_t i = wbase + (size_t)k * 64 + lane;
        const bool valid = i < n;
        key[k] = valid ? keys_in[i] : 0xFFFFFFFFu;
        const uint32_t d = (key[k] >> shift) & (BINS - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
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
    {   // thread d owns digit d
        const uint32_t d = threadIdx.x;
        const uint32_t blk = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
        const uint32_t total = digit_totals[d];
        const uint32_t digit_base = block_excl_scan_256(total, sh, nullptr);
        const uint32_t local_base = block_excl_scan_256(blk, sh, nullptr);
        const uint32_t before_me = lookback(status + d, BINS, vb, blk);
        gbase[d] = digit_base + before_me;
        lstart[d] = local_base;
        uint32_t r2 = local_base;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t c = cnt[k][d];
            cnt[k][d] = r2;
            r2 += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const size_t i = wbase + (size_t)k * 64 + lane;
        if (i < n) {
            const uint32_t d = (key[k] >> shift) & (BINS - 1);
            const uint32_t lp = cnt[w][d] + rank[k];
            skey[lp] = key[k];
            sval[lp] = vals_in ? vals_in[i] : (uint32_t)i;
        }
    }
    __syncthreads();
    const uint32_t have = (uint32_t)min((size_t)CHUNK, n - cbase);
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t j = (uint32_t)k * SORT_THREADS + threadIdx.x;
        if (j < have) {
            const uint32_t kk = skey[j];
            const uint32_t d = (kk >> shift) & (BINS - 1);
            const uint32_t pos = gbase[d] + (j - lstart[d]);
            if (WRITE_KEYS) keys_out[pos] = kk;
            vals_out[pos] = sval[j];
            if (RANGES) record_range(ranges, kk, j == 0 || skey[j - 1] != kk, j + 1 == have || skey[j + 1] != kk, pos);
        }
    }
}

// Before the depth sort: the four digit histograms of the depth keys in one read, the two instance totals
// (sum of the per-workgroup partials of preprocess_kernel; the host reads them back while the sort runs) and
// the zero-fill of every look-back status word the forward pass will use from the geometry buffer.
__global__ void __launch_bounds__(256)
sort_prologue_kernel(const uint32_t* __restrict__ keys, size_t n, uint32_t* __restrict__ depth_hist,
                     const uint32_t* __restrict__ partial, int n_partial, uint32_t* __restrict__ counters,
                     uint32_t* __restrict__ zero_words, size_t n_zero) {
    __shared__ uint32_t h[4][256];
    __shared__ uint32_t shc[2][4];
    for (int d = threadIdx.x; d < 1024; d += 256) (&h[0][0])[d] = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_zero; i += (size_t)gridDim.x * 256) zero_words[i] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    uint32_t culled = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        if (i < n) {
            const uint32_t kk = keys[i];
            if (kk == 0xFFFFFFFFu) { culled++; continue; }      // culled Gaussians: one shared bin per digit
            atomicAdd(&h[0][kk & 255], 1u);
            atomicAdd(&h[1][(kk >> 8) & 255], 1u);
            atomicAdd(&h[2][(kk >> 16) & 255], 1u);
            atomicAdd(&h[3][kk >> 24], 1u);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) culled += (uint32_t)__shfl_xor((int)culled, d, 64);
    if ((threadIdx.x & 63) == 0 && culled)
        for (int p = 0; p < 4; p++) atomicAdd(&h[p][255], culled);
    __syncthreads();
    for (int p = 0; p < 4; p++) {
        const uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&depth_hist[p * 256 + threadIdx.x], c);
    }
    if (blockIdx.x == 0) {   // counters[0] = instances in our lists, counters[1] = the reference's count
        uint32_t v = 0, u = 0;
        for (int i = threadIdx.x; i < n_partial; i += 256) { v += partial[i]; u += partial[n_partial + i]; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            v += (uint32_t)__shfl_xor((int)v, d, 64);
            u += (uint32_t)__shfl_xor((int)u, d, 64);
        }
        if ((threadIdx.x & 63) == 0) { shc[0][threadIdx.x >> 6] = v; shc[1][threadIdx.x >> 6] = u; }
        __syncthreads();
        if (threadIdx.x == 0) {
            counters[1] = shc[0][0] + shc[0][1] + shc[0][2] + shc[0][3];
            counters[0] = shc[1][0] + shc[1][1] + shc[1][2] + shc[1][3];
            counters[3] = 0u;      // "carved for exactly counters[0] entries" (f3dgs_debug_read; the three-kernel flavour: run_totals_job)
        }
    }
}

__global__ void __launch_bounds__(256) iota_kernel(uint32_t* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (uint32_t)i;
}

}  // namespace

void launch_offset_sums(const uint32_t* in, const uint32_t* gather, size_t n, uint32_t* chunk_sums, uint32_t* sub,
                        hipStream_t s) {
    const size_t nb = scan_blocks(n);
    if (nb == 0) return;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(256), 0, s, in, gather, n, chunk_sums, sub);
}

// =====================================================================================================
// Small problems (up to SMALL_SORT_MAX pairs: config c1, early training, inference on small scenes): the WHOLE stable sort in
// one launch of one workgroup, resident in LDS.  A multi-launch pass costs ~20 us of dependent launches whatever the size (c1:
// nine launches = 0.060 ms for 10,000 depth keys, three = 0.024 ms for 16,076 instances); here every 8-bit pass is: rank the
// thread's 16 items per wave (the ballot match of radix_scatter_kernel, wave-ordered LDS counters), turn the 16 x 256 counters
// into offsets, scatter the pairs into the LDS image, read them back in the new order - three barriers, no global round trip.
// The pairs live in registers between passes; the LDS image is written out once at the end (with the tile ranges, RANGES).
// (A first version of this idea - round 4, removed - ping-ponged the pairs through global memory behind fences: slower than the
// launches it replaced.)
// (SMALL_SORT_MAX = 16384: common.h)
constexpr int SMALL_SORT_THREADS = 1024;
constexpr int SMALL_SORT_ITEMS = SMALL_SORT_MAX / SMALL_SORT_THREADS;     // 16
struct SmallSortLds {
    uint32_t key[SMALL_SORT_MAX];
    uint32_t val[SMALL_SORT_MAX];
    uint32_t cnt[SMALL_SORT_THREADS / 64][256];
    uint32_t wsum[4];
};
template <bool RANGES>
__global__ void __launch_bounds__(SMALL_SORT_THREADS)
small_sort_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
                  uint32_t* __restrict__ vals_out, uint32_t n, int passes, TotalsJob tj, uint2* __restrict__ ranges_enc,
                  const uint32_t* __restrict__ n_dev, OffsetSumsJob os) {
    extern __shared__ __attribute__((aligned(16))) char small_sort_smem[];
    SmallSortLds& L = *reinterpret_cast<SmallSortLds*>(small_sort_smem);
    constexpr int ITEMS = SMALL_SORT_ITEMS, NW = SMALL_SORT_THREADS / 64;
    if (n_dev) { const uint32_t nd = *n_dev; n = nd <= n ? nd : 0u; }       // (sync-free forward) n is a capacity, the item count is the device's word; a frame that does not fit is void
    if (tj.partial) run_totals_job<SMALL_SORT_THREADS>(tj);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t* vcnt = &L.cnt[w][0];
    // item order: wave w owns [64 steps w, 64 steps (w + 1)), visited in `steps` <= 16 steps of 64 consecutive items (lane = item
    // within the step): the items are spread evenly over the sixteen waves, the unused steps of a small problem are skipped
    const int steps = (int)((n + SMALL_SORT_THREADS - 1) / SMALL_SORT_THREADS);
    const uint32_t wbase = (uint32_t)(w * steps * 64 + lane);
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = wbase + (uint32_t)(k * 64);
        const bool mine = k < steps && i < n;
        key[k] = mine ? keys_in[i] : 0xFFFFFFFFu;
        val[k] = (vals_in && mine) ? vals_in[i] : i;
    }
    for (int pass = 0; pass < passes; pass++) {
        const int shift = 8 * pass;
        for (int d = tid; d < NW * 256; d += SMALL_SORT_THREADS) (&L.cnt[0][0])[d] = 0;
        __syncthreads();       // counters cleared; the previous pass's read-back is complete
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            if (k >= steps) break;
            const uint32_t i = wbase + (uint32_t)(k * 64);
            const bool valid = i < n;
            const uint32_t d = (key[k] >> shift) & 255u;
            const unsigned long long vm = __ballot(valid);
            uint32_t p_lo = (uint32_t)vm, p_hi = (uint32_t)(vm >> 32);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const uint32_t x = 0u - ((d >> b) & 1u);
                const unsigned long long nm = ~__ballot(x != 0u);
                p_lo &= (uint32_t)nm ^ x;
                p_hi &= (uint32_t)(nm >> 32) ^ x;
            }
            const unsigned long long peers = ((unsigned long long)p_hi << 32) | p_lo;
            const uint32_t before = __popcll(peers & lt_mask);
            const uint32_t old = vcnt[d];
            if (valid && before == 0) vcnt[d] = old + (uint32_t)__popcll(peers);
            rank[k] = old + before;
        }
        __syncthreads();
        // digit d = tid (first 256 threads): total over the waves, exclusive scan over the digits, then per-wave offsets
        {
            uint32_t tot = 0;
            if (tid < 256) {
#pragma unroll
                for (int ww = 0; ww < NW; ww++) tot += L.cnt[ww][tid];
            }
            uint32_t inc = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)inc, d, 64);
                if (lane >= d) inc += v;
            }
            if (tid < 256 && lane == 63) L.wsum[w] = inc;
            __syncthreads();
            if (tid < 256) {
                uint32_t base = inc - tot;
                for (int ww = 0; ww < w; ww++) base += L.wsum[ww];
#pragma unroll
                for (int ww = 0; ww < NW; ww++) {
                    const uint32_t c = L.cnt[ww][tid];
                    L.cnt[ww][tid] = base;
                    base += c;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t i = wbase + (uint32_t)(k * 64);
            if (k < steps && i < n) {
                const uint32_t pos = L.cnt[w][(key[k] >> shift) & 255u] + rank[k];
                L.key[pos] = key[k];
                L.val[pos] = val[k];
            }
        }
        __syncthreads();
        if (pass + 1 < passes) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const uint32_t i = wbase + (uint32_t)(k * 64);
                if (k < steps && i < n) { key[k] = L.key[i]; val[k] = L.val[i]; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t j = (uint32_t)(k * SMALL_SORT_THREADS + tid);
        if (j < n) {
            const uint32_t kk = L.key[j];
            if (keys_out) keys_out[j] = kk;
            vals_out[j] = L.val[j];
            if (RANGES) record_range(ranges_enc, kk, j == 0 || L.key[j - 1] != kk, j + 1 == n || L.key[j + 1] != kk, j);
        }
    }
    if (os.tiles_touched) {
        // The list offsets in the NEW order (scan_reduce_kernel's two levels), from the sorted ids still in LDS: step k of wave w
        // covers items 1024 k + 64 w .. + 63 = run 16 k + w of the order, and a chunk of SCAN_CHUNK items is 64 consecutive runs.
        static_assert(SCAN_CHUNK == 64 * 64 && SMALL_SORT_THREADS == 1024, "run = item / 64, chunk = run / 64");
        const uint32_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
        uint32_t* const runs = &L.cnt[0][0];          // (the counters are done with: the last pass's barriers are behind us)
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t j = (uint32_t)(k * SMALL_SORT_THREADS + tid);
            uint32_t r = j < n ? os.tiles_touched[L.val[j]] : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) r += (uint32_t)__shfl_xor((int)r, d, 64);
            const uint32_t run = (uint32_t)(16 * k + w);
            if (lane == 0) {
                runs[run] = r;
                if (run < 64u * nchunks) os.sub[run] = r;
            }
        }
        __syncthreads();
        if ((uint32_t)tid < nchunks) {
            uint32_t t = 0;
            for (int q = 0; q < 64; q++) t += runs[64 * tid + q];
            os.chunk_sums[tid] = t;
        }
    }
}

__global__ void __launch_bounds__(256) totals_kernel(TotalsJob tj) { run_totals_job<256>(tj); }

// The LDS image is 144 KB: above the default limit, raised once per device (refused: the multi-launch passes take over).
template <bool RANGES>
bool small_sort_usable() {
    constexpr int MAX_DEV = 64;
    static std::atomic<int> state[MAX_DEV];      // 0: not asked yet, 1: usable, -1: refused (the ABI is re-entrant across threads)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return false;
    if (state[dev].load(std::memory_order_acquire) == 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_sort_kernel<RANGES>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmallSortLds));
        if (e != hipSuccess) (void)hipGetLastError();
        state[dev].store(e == hipSuccess ? 1 : -1, std::memory_order_release);
    }
    return state[dev].load(std::memory_order_acquire) > 0;
}

// ITEMS = keys per thread: 8 for the depth sort (2048-key workgroups: P = 1M gives 489 workgroups, about two per CU;
// with 16 there are fewer workgroups than CUs and every pass is one latency chain: 0.116 -> 0.097 ms at c3), 16 for
// the tile sort (twice the instances, and the LDS reorder pays more on longer runs: 0.119 vs 0.128 ms with 8).
template <int BITS, int ITEMS>
static hipError_t radix_pass(const uint32_t* ki, const uint32_t* vi, uint32_t* ko, uint32_t* vo, size_t n, int shift,
                             uint32_t* hist, hipStream_t s, uint2* ranges_enc = nullptr, const TotalsJob* tj = nullptr,
                             const uint32_t* n_dev = nullptr) {
    hipError_t rc = hipSuccess;
    constexpr int BINS = 1 << BITS;
    static_assert(ITEMS >= SORT_ITEMS, "the histogram buffers are sized for SORT_ITEMS keys per thread");
    const uint32_t nb = (uint32_t)((n + SORT_THREADS * ITEMS - 1) / (SORT_THREADS * ITEMS));
    uint32_t* totals = hist + (size_t)BINS * nb;
    hipLaunchKernelGGL((radix_hist_kernel<BITS, ITEMS>), dim3(nb), dim3(SORT_THREADS), 0, s, ki, n, shift, nb, hist,
                       tj ? *tj : TotalsJob{nullptr, 0, nullptr, nullptr}, n_dev);
    if (tj && tj->ready) rc = hipEventRecord(tj->ready, s);        // the totals are final behind this launch
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(BINS), dim3(256), 0, s, hist, nb, totals);
    if (ranges_enc && BITS <= 8)
        hipLaunchKernelGGL((radix_scatter_kernel<BITS, ITEMS, true, true>), dim3(nb), dim3(SORT_THREADS), 0, s, ki, vi, ko, vo, n,
                           shift, nb, hist, totals, ranges_enc, n_dev);
    else
        hipLaunchKernelGGL((radix_scatter_kernel<BITS, ITEMS, (BITS <= 8), false>), dim3(nb), dim3(SORT_THREADS), 0, s, ki, vi, ko,
                           vo, n, shift, nb, hist, totals, (uint2*)nullptr, n_dev);
    return rc;
}

void launch_radix_sort_pairs(uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b, size_t n, int nbits,
                             uint32_t* hist, bool result_in_a, uint2* ranges_enc, hipStream_t s, const uint32_t* n_dev) {
    // n_dev (sync-free forward, api.hip): `n` is the CAPACITY the buffers were carved for and every launch is sized by it; the
    // kernels take the item count from the device word (clamped to n)
    // Input is expected in A when (#passes even) == result_in_a, else in B; the caller arranges that.
    const int passes = (nbits + RADIX_BITS - 1) / RADIX_BITS;
    if (n == 0 || passes == 0) return;
    bool in_a = (passes % 2 == 0) ? result_in_a : !result_in_a;
    if (n <= (size_t)SMALL_SORT_MAX && (ranges_enc ? small_sort_usable<true>() : small_sort_usable<false>())) {
        // input where the caller put it for `passes` hops, result on the side it asked for - in one launch
        const uint32_t* ki = in_a ? key_a : key_b;
        const uint32_t* vi = in_a ? val_a : val_b;
        uint32_t* ko = result_in_a ? key_a : key_b;
        uint32_t* vo = result_in_a ? val_a : val_b;
        const TotalsJob none = {nullptr, 0, nullptr, nullptr};
        if (ranges_enc)
            hipLaunchKernelGGL(small_sort_kernel<true>, dim3(1), dim3(SMALL_SORT_THREADS), sizeof(SmallSortLds), s, ki, vi, ko, vo,
                               (uint32_t)n, passes, none, ranges_enc, n_dev, OffsetSumsJob{nullptr, nullptr, nullptr});
        else
            hipLaunchKernelGGL(small_sort_kernel<false>, dim3(1), dim3(SMALL_SORT_THREADS), sizeof(SmallSortLds), s, ki, vi, ko, vo,
                               (uint32_t)n, passes, none, (uint2*)nullptr, n_dev, OffsetSumsJob{nullptr, nullptr, nullptr});
        return;
    }
    for (int p = 0; p < passes; p++) {
        uint32_t* ki = in_a ? key_a : key_b;
        uint32_t* vi = in_a ? val_a : val_b;
        uint32_t* ko = in_a ? key_b : key_a;
        uint32_t* vo = in_a ? val_b : val_a;
        (void)radix_pass<RADIX_BITS, 16>(ki, vi, ko, vo, n, p * RADIX_BITS, hist, s, p == passes - 1 ? ranges_enc : nullptr, nullptr, n_dev);
        in_a = !in_a;
    }
}

// Depth sort of the Gaussians: 32-bit keys in `keys` (read-only), values = indices.  Three passes of 11/11/10
// bits; the sorted ids end in val_a (and the sorted keys in key_a).
hipError_t launch_depth_sort(const uint32_t* keys, uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b, size_t n,
                             uint32_t* hist, const TotalsJob* tj, hipStream_t s, const OffsetSumsJob* sums, bool* sums_done) {
    if (sums_done) *sums_done = false;
    if (n == 0) return hipSuccess;
    if (n <= (size_t)SMALL_SORT_MAX && small_sort_usable<false>()) {
        // the totals the host waits for go out in a launch of their own, in FRONT of the sort: behind it the host would sit out the
        // whole sort (0.04 ms) before it can enqueue the emit - and small scenes are bound by the host's enqueue time
        hipError_t rc = hipSuccess;
        // (no event to record = a call inside a graph capture: nobody waits on the host, the totals ride in the sort's launch -
        // one node less, ~4.5 us of a replayed c1 step)
        const bool apart = tj && tj->partial && tj->ready;
        if (apart) {
            hipLaunchKernelGGL(totals_kernel, dim3(1), dim3(256), 0, s, *tj);
            rc = hipEventRecord(tj->ready, s);
        }
        hipLaunchKernelGGL(small_sort_kernel<false>, dim3(1), dim3(SMALL_SORT_THREADS), sizeof(SmallSortLds), s, keys, (const uint32_t*)nullptr,
                           key_a, val_a, (uint32_t)n, 4, (tj && tj->partial && !apart) ? *tj : TotalsJob{nullptr, 0, nullptr, nullptr},
                           (uint2*)nullptr, (const uint32_t*)nullptr, sums ? *sums : OffsetSumsJob{nullptr, nullptr, nullptr});
        if (sums && sums_done) *sums_done = true;
        return rc;
    }
    if (n > 200000) {
        // large P: four 8-bit passes are faster than three 11-bit ones (measured at 1M: 0.105 vs 0.148 ms);
        // small P is launch-bound and prefers fewer passes.  Result must end in (key_a, val_a): A <- keys, then
        // A -> B -> A -> ... needs an odd number of remaining hops, so the first pass writes into B.
        const hipError_t rc = radix_pass<RADIX_BITS, SORT_ITEMS>(keys, nullptr, key_b, val_b, n, 0, hist, s, nullptr, tj);
        (void)radix_pass<RADIX_BITS, SORT_ITEMS>(key_b, val_b, key_a, val_a, n, 8, hist, s);
        (void)radix_pass<RADIX_BITS, SORT_ITEMS>(key_a, val_a, key_b, val_b, n, 16, hist, s);
        (void)radix_pass<RADIX_BITS, SORT_ITEMS>(key_b, val_b, key_a, val_a, n, 24, hist, s);
        return rc;
    }
    const hipError_t rc = radix_pass<DEPTH_RADIX_BITS, SORT_ITEMS>(keys, nullptr, key_a, val_a, n, 0, hist, s, nullptr, tj);
    (void)radix_pass<DEPTH_RADIX_BITS, SORT_ITEMS>(key_a, val_a, key_b, val_b, n, DEPTH_RADIX_BITS, hist, s);
    (void)radix_pass<DEPTH_RADIX_BITS, SORT_ITEMS>(key_b, val_b, key_a, val_a, n, 2 * DEPTH_RADIX_BITS, hist, s);
    return rc;
}


void launch_sort_prologue(const GeomState& g, size_t P, hipStream_t s) {
    const size_t nb = scan_blocks(P);
    if (nb == 0) return;
    hipLaunchKernelGGL(sort_prologue_kernel, dim3(nb), dim3(256), 0, s, g.depth_key, P, g.depth_hist, g.ref_partial,
                       (int)((P + 255) / 256), g.counters, g.lb_words, g.n_lb_words);
}

// Depth sort, onesweep flavour: four 8-bit passes, one launch each; ids end in val_a (keys are not written by the
// last pass: nothing reads them).
void launch_depth_sort_onesweep(const GeomState& g, size_t n, hipStream_t s) {
    if (n == 0) return;
    const uint32_t nb = (uint32_t)depth_sort_blocks(n);
    uint32_t* st = g.depth_status;
    const size_t st_stride = (size_t)nb * 256;
    uint2* const no_ranges = nullptr;
    hipLaunchKernelGGL((onesweep_pass_kernel<DEPTH_SORT_ITEMS, true, false>), dim3(nb), dim3(SORT_THREADS), 0, s, g.depth_key,
                       (const uint32_t*)nullptr, g.key_b, g.val_b, n, 0, g.depth_hist, st, g.tickets + 0, no_ranges);
    hipLaunchKernelGGL((onesweep_pass_kernel<DEPTH_SORT_ITEMS, true, false>), dim3(nb), dim3(SORT_THREADS), 0, s, g.key_b,
                       g.val_b, g.key_a, g.val_a, n, 8, g.depth_hist + 256, st + st_stride, g.tickets + 1, no_ranges);
    hipLaunchKernelGGL((onesweep_pass_kernel<DEPTH_SORT_ITEMS, true, false>), dim3(nb), dim3(SORT_THREADS), 0, s, g.key_a,
                       g.val_a, g.key_b, g.val_b, n, 16, g.depth_hist + 512, st + 2 * st_stride, g.tickets + 2, no_ranges);
    hipLaunchKernelGGL((onesweep_pass_kernel<DEPTH_SORT_ITEMS, false, false>), dim3(nb), dim3(SORT_THREADS), 0, s, g.key_b,
                       g.val_b, g.key_a, g.val_a, n, 24, g.depth_hist + 768, st + 3 * st_stride, g.tickets + 3, no_ranges);
}

// Tile sort, onesweep flavour (ids only travel with their tile ids; `passes` = 1 or 2 digits of 8 bits).
// Input in (tile_tmp, id_tmp) for an odd number of passes, in (tile_sorted, point_list) for an even one; the
// result lands in (tile_sorted, point_list).
void launch_tile_sort_onesweep(const GeomState& g, const BinState& b, size_t n, int passes, uint2* ranges, hipStream_t s) {
    if (n == 0) return;
    const uint32_t nb = (uint32_t)sort_blocks(n);
    const size_t st_stride = (size_t)nb * 256;
    if (passes == 1) {
        hipLaunchKernelGGL((onesweep_pass_kernel<SORT_ITEMS, true, true>), dim3(nb), dim3(SORT_THREADS), 0, s, b.tile_tmp,
                           b.id_tmp, b.tile_sorted, b.point_list, n, 0, g.tile_hist, b.tile_status, g.tickets + 5, ranges);
        return;
    }
    hipLaunchKernelGGL((onesweep_pass_kernel<SORT_ITEMS, true, false>), dim3(nb), dim3(SORT_THREADS), 0, s, b.tile_sorted,
                       b.point_list, b.tile_tmp, b.id_tmp, n, 0, g.tile_hist, b.tile_status, g.tickets + 5, (uint2*)nullptr);
    hipLaunchKernelGGL((onesweep_pass_kernel<SORT_ITEMS, true, true>), dim3(nb), dim3(SORT_THREADS), 0, s, b.tile_tmp,
                       b.id_tmp, b.tile_sorted, b.point_list, n, 8, g.tile_hist + 256, b.tile_status + st_stride,
                       g.tickets + 6, ranges);
}

// Stable sort of (keys[i], i) on key bits [0, nbits): the sorted indices end in val_a (8-bit passes; `keys` is only read).
void launch_radix_sort_keys_to_order(const uint32_t* keys, uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b,
                                     size_t n, int nbits, uint32_t* hist, hipStream_t s) {
    const int passes = (nbits + RADIX_BITS - 1) / RADIX_BITS;
    if (n == 0 || passes == 0) return;
    bool to_a = (passes % 2) == 1;        // the last pass must write the A side
    const uint32_t* ki = keys;
    const uint32_t* vi = nullptr;
    for (int p = 0; p < passes; p++) {
        uint32_t* ko = to_a ? key_a : key_b;
        uint32_t* vo = to_a ? val_a : val_b;
        (void)radix_pass<RADIX_BITS, SORT_ITEMS>(ki, vi, ko, vo, n, p * RADIX_BITS, hist, s);
        ki = ko; vi = vo;
        to_a = !to_a;
    }
}

void launch_iota(uint32_t* dst, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, n);
}

}  // namespace f3dgs
