// knn.hip - mean squared distance to the three nearest neighbours of every point (the initial Gaussian scales).
//
// Replaces `simple_knn._C.distCUDA2` (reference: submodules/simple-knn/simple_knn.cu:45-221, spatial.cu:15-25; caller
// scene/gaussian_model.py:146 `dist2 = torch.clamp_min(distCUDA2(points), 0.0000001)`).  The reference's result is
// the EXACT 3-NN answer (its pruning radius is an upper bound), so any exact search reproduces it up to the
// rounding of a squared distance; a missing neighbour (P < 4) counts as FLT_MAX like there.
//
// Design for MI355X (nothing on the host: the reference reads the bounding box back twice):
//   1. per-workgroup min/max of the points                                   (knn_bounds_kernel)
//   2. 30-bit Morton codes against the global box (every workgroup folds the partials itself), then the library's
//      radix sort on (code, index)                                           (knn_morton_kernel + binning.hip)
//   3. the points are GATHERED into Morton order once, as float4 {x, y, z, original index}: all later reads are
//      contiguous (the reference chases points[indices[i]] for every candidate)   (knn_gather_kernel)
//   4. one workgroup per leaf of 256 sorted points computes the leaf's box    (fused into the gather)
//   5. queries: one WAVE owns 64 consecutive sorted points (spatial neighbours).  It seeds the three best distances
//      from its own leaf, then walks all leaves: a leaf is visited if ANY lane's current third-best distance reaches
//      its box (wave-uniform ballot, so the leaf's points are read once per wave as broadcasts), and inside a visited
//      leaf every lane keeps its own running best three.                     (knn_query_kernel)

#include <float.h>

#include "common.h"

namespace f3dgs {

namespace {

constexpr int LEAF = 256;

struct Box {
    float lo[3], hi[3];
};

__device__ __forceinline__ uint32_t spread10(uint32_t x) {   // 10 bits -> every third bit
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

// min/max over one workgroup of 256 threads; result valid in thread 0
__device__ __forceinline__ void block_bounds(float (&lo)[3], float (&hi)[3], float (*sh)[6]) {
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; a++) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { sh[w][a] = lo[a]; sh[w][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(fminf(sh[0][a], sh[1][a]), fminf(sh[2][a], sh[3][a]));
            hi[a] = fmaxf(fmaxf(sh[0][3 + a], sh[1][3 + a]), fmaxf(sh[2][3 + a], sh[3][3 + a]));
        }
    }
}

__global__ void __launch_bounds__(256) knn_bounds_kernel(int P, const float* __restrict__ pts, Box* __restrict__ partial) {
    __shared__ float sh[4][6];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * (size_t)i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
    block_bounds(lo, hi, sh);
    if (threadIdx.x == 0) {
        Box b;
#pragma unroll
        for (int a = 0; a < 3; a++) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; }
        partial[blockIdx.x] = b;
    }
}

__global__ void __launch_bounds__(256) knn_morton_kernel(int P, const float* __restrict__ pts, const Box* __restrict__ partial,
                                                         int n_partial, uint32_t* __restrict__ codes) {
    __shared__ float sh[4][6];
    __shared__ float gb[6];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int k = threadIdx.x; k < n_partial; k += 256) {
        const Box b = partial[k];
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], b.lo[a]); hi[a] = fmaxf(hi[a], b.hi[a]); }
    }
    block_bounds(lo, hi, sh);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { gb[a] = lo[a]; gb[3 + a] = hi[a]; }
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float ext = gb[3 + a] - gb[a];
        const float t = ext > 0.f ? (pts[3 * (size_t)i + a] - gb[a]) / ext : 0.f;
        const uint32_t q = (uint32_t)fminf(1023.f, fmaxf(0.f, t * 1023.f));
        code |= spread10(q) << a;
    }
    codes[i] = code;
}

// sorted[j] = {point of rank j, its original index}; one workgroup = one leaf -> its box
__global__ void __launch_bounds__(LEAF) knn_gather_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                          float4* __restrict__ sorted, Box* __restrict__ leaf_box) {
    __shared__ float sh[4][6];
    const int j = blockIdx.x * LEAF + threadIdx.x;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (j < P) {
        const uint32_t i = order[j];
        const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
        sorted[j] = make_float4(x, y, z, __uint_as_float(i));
        lo[0] = hi[0] = x; lo[1] = hi[1] = y; lo[2] = hi[2] = z;
    }
    block_bounds(lo, hi, sh);
    if (threadIdx.x == 0) {
        Box b;
#pragma unroll
        for (int a = 0; a < 3; a++) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; }
        leaf_box[blockIdx.x] = b;
    }
}

__device__ __forceinline__ float box_dist2(const Box& b, float x, float y, float z) {
    const float dx = fmaxf(fmaxf(b.lo[0] - x, x - b.hi[0]), 0.f);
    const float dy = fmaxf(fmaxf(b.lo[1] - y, y - b.hi[1]), 0.f);
    const float dz = fmaxf(fmaxf(b.lo[2] - z, z - b.hi[2]), 0.f);
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ void keep3(float d, float& b0, float& b1, float& b2) {   // b0 <= b1 <= b2
    if (d < b2) {
        b2 = d;
        if (b2 < b1) { const float t = b1; b1 = b2; b2 = t; }
        if (b1 < b0) { const float t = b0; b0 = b1; b1 = t; }
    }
}

// The candidates of one leaf against the wave's 64 queries.  Candidate points are wave-uniform (scalar loads).
__device__ __forceinline__ void scan_leaf(const float4* __restrict__ sorted, int begin, int end, int self, bool want, float qx,
                                          float qy, float qz, float& b0, float& b1, float& b2) {
    for (int c = begin; c < end; c++) {
        const float4 p = sorted[c];                       // same address in every lane
        const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
        const float d = dx * dx + dy * dy + dz * dz;
        if (want && c != self) keep3(d, b0, b1, b2);
    }
}

__global__ void __launch_bounds__(64) knn_query_kernel(int P, const float4* __restrict__ sorted, const Box* __restrict__ leaf_box,
                                                       int n_leaf, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const int j = blockIdx.x * 64 + lane;
    const bool have = j < P;
    const float4 q = sorted[have ? j : P - 1];
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    const int home = (blockIdx.x * 64) / LEAF;
    // seed from the wave's own leaf (spatial neighbours), then every other leaf whose box some lane still reaches
    scan_leaf(sorted, home * LEAF, min(P, (home + 1) * LEAF), j, have, q.x, q.y, q.z, b0, b1, b2);
    for (int b = 0; b < n_leaf; b++) {
        if (b == home) continue;
        const Box bx = leaf_box[b];
        const bool want = have && box_dist2(bx, q.x, q.y, q.z) <= b2;
        if (__ballot(want) == 0) continue;
        scan_leaf(sorted, b * LEAF, min(P, (b + 1) * LEAF), j, want, q.x, q.y, q.z, b0, b1, b2);
    }
    if (have) out[__float_as_uint(q.w)] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

size_t knn_scratch_bytes(size_t P) {
    Carver c(nullptr);
    c.take<Box>(1024);
    c.take<uint32_t>(P);                 // codes
    c.take<uint32_t>(P); c.take<uint32_t>(P); c.take<uint32_t>(P); c.take<uint32_t>(P);   // sort ping-pong
    c.take<uint32_t>(RADIX_BINS * sort_blocks(P) + RADIX_BINS);
    c.take<float4>(P);
    c.take<Box>((P + LEAF - 1) / LEAF);
    return c.total();
}

void launch_knn_mean_dist2(int P, const float* points, float* out, char* scratch, hipStream_t s) {
    Carver c(scratch);
    Box* partial = c.take<Box>(1024);
    uint32_t* codes = c.take<uint32_t>(P);
    uint32_t* key_a = c.take<uint32_t>(P);
    uint32_t* val_a = c.take<uint32_t>(P);
    uint32_t* key_b = c.take<uint32_t>(P);
    uint32_t* val_b = c.take<uint32_t>(P);
    uint32_t* hist = c.take<uint32_t>(RADIX_BINS * sort_blocks(P) + RADIX_BINS);
    float4* sorted = c.take<float4>(P);
    const int n_leaf = (P + LEAF - 1) / LEAF;
    Box* leaf_box = c.take<Box>(n_leaf);
    const int nb = min(1024, (P + 255) / 256);
    hipLaunchKernelGGL(knn_bounds_kernel, dim3(nb), dim3(256), 0, s, P, points, partial);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, partial, nb, codes);
    // 30-bit codes: four 8-bit passes; ids start as the index order -> equal codes keep ascending index
    launch_radix_sort_keys_to_order(codes, key_a, val_a, key_b, val_b, (size_t)P, 30, hist, s);
    hipLaunchKernelGGL(knn_gather_kernel, dim3(n_leaf), dim3(LEAF), 0, s, P, points, val_a, sorted, leaf_box);
    hipLaunchKernelGGL(knn_query_kernel, dim3((P + 63) / 64), dim3(64), 0, s, P, sorted, leaf_box, n_leaf, out);
}

}  // namespace f3dgs
