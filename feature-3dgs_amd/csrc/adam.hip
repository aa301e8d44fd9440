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

// One element's update; every loop below goes through it, so the vector path, the ragged tail and the masked
// variant round identically (explicit fmaf: no contraction choices left to the compiler).
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float step_size, float b2, float omb1,
                                            float omb2, float inv_sqrt_bc2, float eps) {
    m = fmaf(omb1, g - m, m);                  // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(b2, v, (omb2 * g) * g);
    p -= step_size * (m / fmaf(sqrtf(v), inv_sqrt_bc2, eps));
}

__global__ void __launch_bounds__(256)
adam_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            float step_size, float b1, float b2, float omb1, float omb2, float inv_sqrt_bc2, float eps,
            const uint8_t* __restrict__ row_mask, uint32_t width) {
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (row_mask) {
        // visibility-masked variant: rows whose mask byte is 0 keep parameter AND moments (no decay) - they cost one
        // byte of traffic instead of 28 per element
        const size_t last = (i4 + 4 <= n ? i4 + 4 : n) - 1;
        const size_t r0 = i4 / width, r1 = last / width;
        bool any = false;
        for (size_t r = r0; r <= r1; r++) any = any || row_mask[r] != 0;
        if (!any) return;
        for (size_t i = i4; i <= last; i++) {
            if (!row_mask[i / width]) continue;
            adam_update(p[i], g[i], m[i], v[i], step_size, b2, omb1, omb2, inv_sqrt_bc2, eps);
        }
        return;
    }
    if (i4 + 4 <= n && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        float4 pp = *reinterpret_cast<float4*>(p + i4), mm = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
        const float4 gg = *reinterpret_cast<const float4*>(g + i4);
        float* pa = &pp.x; float* ma = &mm.x; float* va = &vv.x; const float* ga = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            adam_update(pa[k], ga[k], ma[k], va[k], step_size, b2, omb1, omb2, inv_sqrt_bc2, eps);
        }
        *reinterpret_cast<float4*>(p + i4) = pp;
        *reinterpret_cast<float4*>(m + i4) = mm;
        *reinterpret_cast<float4*>(v + i4) = vv;
        return;
    }
    for (size_t i = i4; i < n && i < i4 + 4; i++) {
        adam_update(p[i], g[i], m[i], v[i], step_size, b2, omb1, omb2, inv_sqrt_bc2, eps);
    }
}

// ---- several tensors, one launch: block b belongs to the tensor whose block range contains it ------------------------------
struct AdamEntry {
    float* p; const float* g; float* m; float* v;
    size_t n;
    uint32_t block0;         // first workgroup of this tensor
    uint32_t width;          // floats per row (row mask variant), 0: the mask does not apply to this tensor
    float step_size, inv_sqrt_bc2;
};
struct AdamTable {
    AdamEntry t[F3DGS_ADAM_MAX_TENSORS];
    int count;
};

__global__ void __launch_bounds__(256)
adam_multi_kernel(AdamTable tab, float b2, float omb1, float omb2, float eps, const uint8_t* __restrict__ row_mask) {
    int k = 0;
#pragma unroll 1
    while (k + 1 < tab.count && blockIdx.x >= tab.t[k + 1].block0) k++;
    const AdamEntry e = tab.t[k];
    const size_t i4 = ((size_t)(blockIdx.x - e.block0) * 256 + threadIdx.x) * 4;
    if (i4 >= e.n) return;
    float* __restrict__ p = e.p; const float* __restrict__ g = e.g; float* __restrict__ m = e.m; float* __restrict__ v = e.v;
    if (row_mask && e.width) {
        const size_t last = (i4 + 4 <= e.n ? i4 + 4 : e.n) - 1;
        const size_t r0 = i4 / e.width, r1 = last / e.width;
        bool any = false;
        for (size_t r = r0; r <= r1; r++) any = any || row_mask[r] != 0;
        if (!any) return;
        for (size_t i = i4; i <= last; i++) {
            if (!row_mask[i / e.width]) continue;
            adam_update(p[i], g[i], m[i], v[i], e.step_size, b2, omb1, omb2, e.inv_sqrt_bc2, eps);
        }
        return;
    }
    if (i4 + 4 <= e.n && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        float4 pp = *reinterpret_cast<float4*>(p + i4), mm = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
        const float4 gg = *reinterpret_cast<const float4*>(g + i4);
        float* pa = &pp.x; float* ma = &mm.x; float* va = &vv.x; const float* ga = &gg.x;
#pragma unroll
        for (int q = 0; q < 4; q++) adam_update(pa[q], ga[q], ma[q], va[q], e.step_size, b2, omb1, omb2, e.inv_sqrt_bc2, eps);
        *reinterpret_cast<float4*>(p + i4) = pp;
        *reinterpret_cast<float4*>(m + i4) = mm;
        *reinterpret_cast<float4*>(v + i4) = vv;
        return;
    }
    for (size_t i = i4; i < e.n && i < i4 + 4; i++) adam_update(p[i], g[i], m[i], v[i], e.step_size, b2, omb1, omb2, e.inv_sqrt_bc2, eps);
}

}  // namespace

void launch_adam_step_multi(int count, const f3dgs_adam_tensor* tensors, double b1, double b2, double eps, const uint8_t* row_mask,
                            size_t rows, hipStream_t s) {
    AdamTable tab;
    tab.count = 0;
    uint32_t blocks = 0;
    for (int i = 0; i < count; i++) {
        const f3dgs_adam_tensor& t = tensors[i];
        if (t.n == 0) continue;
        AdamEntry& e = tab.t[tab.count++];
        e.p = t.param; e.g = t.grad; e.m = t.exp_avg; e.v = t.exp_avg_sq; e.n = t.n;
        e.block0 = blocks;
        e.width = (row_mask && rows && t.n % rows == 0) ? (uint32_t)(t.n / rows) : 0u;
        // every derived constant is formed in double and rounded once, as in launch_adam_step
        const double bc1 = 1.0 - pow(b1, (double)t.step), bc2 = 1.0 - pow(b2, (double)t.step);
        e.step_size = (float)(t.lr / bc1);
        e.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        blocks += (uint32_t)(((t.n + 3) / 4 + 255) / 256);
    }
    if (tab.count == 0) return;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, s, tab, (float)b2, (float)(1.0 - b1), (float)(1.0 - b2),
                       (float)eps, row_mask);
}

void launch_adam_step(size_t n, float* p, const float* g, float* m, float* v, double lr, double b1, double b2, double eps, int step,
                      const uint8_t* row_mask, size_t width, hipStream_t s) {
    if (n == 0) return;
    // every derived constant is formed in double from the caller's doubles and rounded once, like torch does
    // (1 - 0.999f in float is off by 1.3e-5 relative)
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    const size_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, n, p, g, m, v, (float)(lr / bc1), (float)b1,
                       (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)(1.0 / sqrt(bc2)), (float)eps, row_mask, (uint32_t)(width ? width : 1));
}

}  // namespace f3dgs
