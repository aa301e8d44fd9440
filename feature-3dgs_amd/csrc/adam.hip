// adam.hip - one Adam step over one parameter tensor (optimizer part).
//
// Semantics: torch.optim.Adam as the reference configures it (scene/gaussian_model.py:163-178: eps = 1e-15, default
// betas, no weight decay, no amsgrad), i.e. per element
//     m = m + (1 - b1)(g - m);  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The reference steps seven tensors with torch's multi-kernel foreach path; this is ONE streaming pass per tensor:
// 16 bytes read + 12 written per element, float4-vectorised.
#include "common.h"

namespace f3dgs {

namespace {

__global__ void __launch_bounds__(256)
adam_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            float step_size, float b1, float b2, float omb1, float omb2, float inv_sqrt_bc2, float eps) {
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        float4 pp = *reinterpret_cast<float4*>(p + i4), mm = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
        const float4 gg = *reinterpret_cast<const float4*>(g + i4);
        float* pa = &pp.x; float* ma = &mm.x; float* va = &vv.x; const float* ga = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ma[k] = fmaf(omb1, ga[k] - ma[k], ma[k]);          // torch: exp_avg.lerp_(grad, 1 - beta1)
            va[k] = b2 * va[k] + omb2 * ga[k] * ga[k];
            pa[k] -= step_size * (ma[k] / (sqrtf(va[k]) * inv_sqrt_bc2 + eps));
        }
        *reinterpret_cast<float4*>(p + i4) = pp;
        *reinterpret_cast<float4*>(m + i4) = mm;
        *reinterpret_cast<float4*>(v + i4) = vv;
        return;
    }
    for (size_t i = i4; i < n && i < i4 + 4; i++) {
        const float gi = g[i];
        const float mi = fmaf(omb1, gi - m[i], m[i]);
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}

}  // namespace

void launch_adam_step(size_t n, float* p, const float* g, float* m, float* v, double lr, double b1, double b2, double eps, int step,
                      hipStream_t s) {
    if (n == 0) return;
    // every derived constant is formed in double from the caller's doubles and rounded once, like torch does
    // (1 - 0.999f in float is off by 1.3e-5 relative)
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    const size_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, n, p, g, m, v, (float)(lr / bc1), (float)b1,
                       (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)(1.0 / sqrt(bc2)), (float)eps);
}

}  // namespace f3dgs
