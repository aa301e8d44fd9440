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

// grid.y = tensor; grid.x strides over that tensor's n_out * width output floats: writes are contiguous, reads are
// contiguous within a row (rows of the same source block stay together, so they are nearly contiguous overall).
__global__ void __launch_bounds__(256)
densify_gather_kernel(size_t n_out, const int32_t* __restrict__ src_row, const uint8_t* __restrict__ kind,
                      const int32_t* __restrict__ override_row, GatherTable tab) {
    const DensifyTensor d = tab.t[blockIdx.y];
    const size_t total = n_out * (size_t)d.width;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / (size_t)d.width;
        const int col = (int)(e - row * (size_t)d.width);
        const uint8_t k = kind[row];
        float v;
        if (d.mode == DENSIFY_ZERO_NEW && k != 0) v = 0.f;
        else if (d.mode == DENSIFY_OVERRIDE_CHILD && k == 2) v = d.override_src[(size_t)override_row[row] * d.width + col];
        else v = d.src[(size_t)src_row[row] * d.width + col];
        d.dst[e] = v;
    }
}

}  // namespace

void launch_densify_gather(size_t n_out, const int32_t* src_row, const uint8_t* kind, const int32_t* override_row, int n_tensors,
                           const DensifyTensor* tensors, hipStream_t s) {
    if (n_out == 0 || n_tensors == 0) return;
    GatherTable tab;
    int wmax = 1;
    for (int i = 0; i < n_tensors; i++) {
        tab.t[i] = tensors[i];
        wmax = tensors[i].width > wmax ? tensors[i].width : wmax;
    }
    const size_t blocks = (n_out * (size_t)wmax + 255) / 256;
    const unsigned gx = (unsigned)(blocks < 8192 ? blocks : 8192);       // narrower tensors loop fewer times
    hipLaunchKernelGGL(densify_gather_kernel, dim3(gx, n_tensors), dim3(256), 0, s, n_out, src_row, kind, override_row, tab);
}

}  // namespace f3dgs
