// binding_knn.cpp - `simple_knn._C`: the reference's second native module (submodules/simple-knn/ext.cpp:15-17,
// spatial.cu:15-25) on top of the C ABI of libf3dgs_hip.so (include/f3dgs.h: f3dgs_knn_mean_dist2).
//   distCUDA2(points (P,3) float32 on the GPU) -> (P,) float32: mean squared distance to the 3 nearest neighbours.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <stdexcept>
#include <string>

#include "../../include/f3dgs.h"

namespace {

torch::Tensor distCUDA2(const torch::Tensor& points) {
    TORCH_CHECK(points.is_cuda(), "points must live on a HIP device: simple_knn has no CPU path");
    TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "points must have dimensions (num_points, 3)");
    TORCH_CHECK(points.scalar_type() == torch::kFloat32, "points must be float32");
    const int P = (int)points.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(points.device());
    auto pts = points.contiguous();
    torch::Tensor means = torch::empty({P}, points.options());
    if (P == 0) return means;
    torch::Tensor scratch = torch::empty({(long long)f3dgs_knn_scratch_bytes(P)}, points.options().dtype(torch::kByte));
    void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(points.device().index()).stream();
    const int rc = f3dgs_knn_mean_dist2(P, pts.data_ptr<float>(), means.data_ptr<float>(), scratch.data_ptr(), stream);
    if (rc != F3DGS_OK) throw std::runtime_error(std::string("distCUDA2: ") + f3dgs_last_error());
    return means;
}

}  // namespace

PYBIND11_MODULE(_C, m) { m.def("distCUDA2", &distCUDA2); }
