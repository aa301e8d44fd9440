// Development aid: throughput of fp32 global atomic adds on MI355X by memory scope.
//   agent scope      (what unsafeAtomicAdd emits: sc1, performed at the memory side, coherent across the 8 XCD L2s)
//   workgroup scope  (no sc1: performed in the issuing XCD's own L2)
// Every wave adds RUN consecutive floats (one 128-B or 256-B run per instruction) at pseudo-random record
// positions of a large buffer, like the gradient flush of render_bwd.hip.  Also reports whether sums survive
// when all XCDs hit the same addresses (they must not, for the L2-local flavour - that is the point).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/atomic_scope.hip -o tools/ubench/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SCOPE, bool XCD_PRIVATE>
__global__ void __launch_bounds__(64) k_atomic(float* buf, uint32_t records, int iters, uint32_t* xcc_seen) {
    const uint32_t lane = threadIdx.x;
    uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) ;   // HW_REG_XCC_ID (id 20), bits 0..3
    xcc &= 7;
    if (lane == 0) atomicOr(&xcc_seen[blockIdx.x & 1023], 1u << xcc);
    uint32_t h = blockIdx.x * 2654435761u + 12345u;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        uint32_t r = (h >> 8) % records;
        if (XCD_PRIVATE) r = (r / 8) * 8 + xcc;      // records owned by this XCD only
        float* p = buf + (size_t)r * 64 + lane;
        __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, SCOPE);
    }
}

template <int SCOPE, bool XP>
double run(float* buf, uint32_t records, int blocks, int iters, uint32_t* seen, double* sum_out, const char* name) {
    hipMemset(buf, 0, (size_t)records * 64 * 4);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_atomic<SCOPE, XP>), dim3(blocks), dim3(64), 0, 0, buf, records, iters, seen);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<float> h((size_t)records * 64);
    hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost);
    double s = 0;
    for (float v : h) s += v;
    const double n = (double)blocks * iters * 64;
    printf("%-34s records %8u  %7.3f ms  %7.1f G dword-atomics/s  sum/expected = %.6f\n", name, records, ms, n / ms * 1e-6, s / n);
    if (sum_out) *sum_out = s / n;
    return ms;
}

int main() {
    const int blocks = 32768, iters = 64;
    float* buf;
    uint32_t* seen;
    const uint32_t big = 1u << 20;          // 1M records x 256 B = 256 MB
    hipMalloc(&buf, (size_t)big * 64 * 4);
    hipMalloc(&seen, 4096);
    hipMemset(seen, 0, 4096);
    for (uint32_t records : {big, 1u << 14, 64u}) {
        run<__HIP_MEMORY_SCOPE_AGENT, false>(buf, records, blocks, iters, seen, nullptr, "agent scope");
        run<__HIP_MEMORY_SCOPE_WORKGROUP, false>(buf, records, blocks, iters, seen, nullptr, "workgroup scope (any XCD)");
        run<__HIP_MEMORY_SCOPE_WORKGROUP, true>(buf, records, blocks, iters, seen, nullptr, "workgroup scope, XCD-private recs");
        run<__HIP_MEMORY_SCOPE_AGENT, true>(buf, records, blocks, iters, seen, nullptr, "agent scope, XCD-private recs");
    }
    std::vector<uint32_t> hs(1024);
    hipMemcpy(hs.data(), seen, 4096, hipMemcpyDeviceToHost);
    int consistent = 0;
    for (int i = 0; i < 1024; i++) consistent += (hs[i] == (1u << (i % 8)));
    printf("blockIdx %% 8 == XCC_ID for %d of 1024 residue classes (mask of class 0: 0x%x, class 1: 0x%x)\n", consistent, hs[0], hs[1]);
    return 0;
}
