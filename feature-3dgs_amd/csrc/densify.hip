// densify.hip - the row movement of one densify-and-prune (or prune) as ONE gather launch over all per-Gaussian
// tensors (densification part).
//
// Reference: scene/gaussian_model.py:300-431.  There, one densification is ~120 torch indexing / cat launches: for each
// of the 7 parameter tensors and their 14 Adam moments, `densify_and_clone` concatenates, `densify_and_split`
// concatenates again and then masks the split sources away, and `densify_and_prune` masks once more - every tensor is
// re-materialised four times and the allocator sees P-sized blocks come and go.  What happens to the ROWS is a single
// function  out_row -> (source row, kind)  that the host derives from the reference's masks (densify.py); given that
// plan every output tensor is one gather:
//
//     kind 0  kept original : the row and its optimizer moments are copied
//     kind 1  clone         : the row is copied, its moments start at zero (cat_tensors_to_optimizer, :337-357)
//     kind 2  split child   : like a clone, except that `xyz` and `scaling` take the child's freshly computed row
//                             (densify_and_split, :385-395) from an override array
//
// The outputs are capacity-sized buffers owned by the caller (stable shapes: nothing is reallocated while the count
// stays under the capacity), so the whole operation is (bytes of the model) read + written once.
#include "common.h"

namespace f3dgs {

namespace {

struct GatherTable {
    DensifyTensor t[DENSIFY_MAX_TENSORS];
};

// grid.y = tensor; a workgroup owns a block of consecutive OUTPUT rows of that tensor (rows_per_block(width): about
// 4096 floats), stages their plan entries in LDS once and then streams the block: writes are contiguous over the
// whole block, reads are contiguous within each source row (and kept rows of one block come from neighbouring
// source rows).  The row of an element is found with one float multiply (block-local indices stay below 2^24).
constexpr int GATHER_MAX_ROWS = 1024;
__host__ __device__ inline int rows_per_block(int width) {
    const int r = 4096 / width;
    return r < 64 ? 64 : (r > GATHER_MAX_ROWS ? GATHER_MAX_ROWS : r);
}

__global__ void __launch_bounds__(256)
densify_gather_kernel(size_t n_out, const int32_t* __restrict__ src_row, const uint8_t* __restrict__ kind,
                      const int32_t* __restrict__ override_row, GatherTable tab) {
    __shared__ int32_t s_row[GATHER_MAX_ROWS];       // source row; for an overridden child: -(override row) - 1
    __shared__ uint8_t s_zero[GATHER_MAX_ROWS];
    const DensifyTensor d = tab.t[blockIdx.y];
    const int R = rows_per_block(d.width);
    const size_t row0 = (size_t)blockIdx.x * R;
    if (row0 >= n_out) return;
    const int rows = (int)(n_out - row0 < (size_t)R ? n_out - row0 : (size_t)R);
    for (int r = threadIdx.x; r < rows; r += 256) {
        const uint8_t k = kind[row0 + r];
        const bool child = d.mode == DENSIFY_OVERRIDE_CHILD && k == 2;
        s_row[r] = child ? -override_row[row0 + r] - 1 : src_row[row0 + r];
        s_zero[r] = d.mode == DENSIFY_ZERO_NEW && k != 0;
    }
    __syncthreads();
    const uint32_t w = (uint32_t)d.width, total = (uint32_t)rows * w;
    const float inv_w = 1.0f / (float)w;
    float* __restrict__ out = d.dst + row0 * w;
    for (uint32_t e = threadIdx.x; e < total; e += 256) {
        uint32_t r = (uint32_t)(((float)e + 0.5f) * inv_w);
        r = r * w > e ? r - 1 : (r * w + w <= e ? r + 1 : r);      // the float estimate is off by at most one
        const uint32_t col = e - r * w;
        const int32_t sr = s_row[r];
        float v = 0.f;
        if (!s_zero[r]) v = sr >= 0 ? d.src[(size_t)sr * w + col] : d.override_src[(size_t)(-sr - 1) * w + col];
        out[e] = v;
    }
}

}  // namespace

void launch_densify_gather(size_t n_out, const int32_t* src_row, const uint8_t* kind, const int32_t* override_row, int n_tensors,
                           const DensifyTensor* tensors, hipStream_t s) {
    if (n_out == 0 || n_tensors == 0) return;
    GatherTable tab;
    size_t blocks = 1;
    for (int i = 0; i < n_tensors; i++) {
        tab.t[i] = tensors[i];
        const size_t b = (n_out + rows_per_block(tensors[i].width) - 1) / rows_per_block(tensors[i].width);
        blocks = b > blocks ? b : blocks;
    }
    // tensors with fewer row blocks than the widest grid leave their surplus workgroups at the first test
    hipLaunchKernelGGL(densify_gather_kernel, dim3((unsigned)blocks, n_tensors), dim3(256), 0, s, n_out, src_row, kind, override_row, tab);
}

}  // namespace f3dgs
