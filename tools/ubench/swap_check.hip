// Development aid: checks the 4x4 row-block transpose (permlane32_swap + permlane16_swap) and the operand /
// result layout of v_mfma_f32_16x16x4_f32 that render_bwd.hip relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/swap_check.hip -o /tmp/swap_check && /tmp/swap_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(float* out_a, float* out_d, const float* s, const float* bmat) {
    const int l = threadIdx.x;
    // s[u][inst]: value of pixel u (k index) for instance inst
    const float s0 = s[l], s1 = s[64 + l], s2 = s[128 + l], s3 = s[192 + l];
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(s0), __float_as_uint(s2), false, false);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s3), false, false);
    auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    const float A[4] = {__uint_as_float(c[0]), __uint_as_float(c[1]), __uint_as_float(d[0]), __uint_as_float(d[1])};
    const float B = bmat[(l >> 4) * 16 + (l & 15)];   // B[k][n]
    for (int blk = 0; blk < 4; blk++) {
        out_a[blk * 64 + l] = A[blk];
        f32x4 acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[blk], B, acc, 0, 0, 0);
        for (int r = 0; r < 4; r++) out_d[(blk * 4 + r) * 64 + l] = acc[r];
    }
}

int main() {
    std::vector<float> s(256), bm(64), oa(256), od(1024);
    for (int u = 0; u < 4; u++) for (int i = 0; i < 64; i++) s[u * 64 + i] = 100.f * u + i;   // value encodes (pixel u, instance i)
    for (int kk = 0; kk < 4; kk++) for (int n = 0; n < 16; n++) bm[kk * 16 + n] = (kk + 1) * 0.001f + n * 7.f;  // asymmetric
    float *ds, *db, *da, *dd;
    hipMalloc(&ds, 1024); hipMalloc(&db, 256); hipMalloc(&da, 1024); hipMalloc(&dd, 4096);
    hipMemcpy(ds, s.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, bm.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dd, ds, db);
    hipMemcpy(oa.data(), da, 1024, hipMemcpyDeviceToHost); hipMemcpy(od.data(), dd, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    // expected A_blk lane (k*16+i) = s[k][16*blk + i]
    for (int blk = 0; blk < 4; blk++) for (int l = 0; l < 64; l++) {
        const float want = s[(l >> 4) * 64 + 16 * blk + (l & 15)];
        if (oa[blk * 64 + l] != want) { if (bad < 8) printf("A blk %d lane %d got %g want %g\n", blk, l, oa[blk * 64 + l], want); bad++; }
    }
    // expected D[i][n] = sum_k s[k][16 blk + i] * B[k][n], held by lane n + 16*(i/4), register i%4
    for (int blk = 0; blk < 4; blk++) for (int i = 0; i < 16; i++) for (int n = 0; n < 16; n++) {
        float want = 0;
        for (int kk = 0; kk < 4; kk++) want = fmaf(s[kk * 64 + 16 * blk + i], bm[kk * 16 + n], want);
        const float got = od[(blk * 4 + (i & 3)) * 64 + n + 16 * (i >> 2)];
        if (fabsf(got - want) > 1e-3f * fabsf(want) + 1e-3f) { if (bad < 16) printf("D blk %d i %d n %d got %g want %g\n", blk, i, n, got, want); bad++; }
    }
    printf(bad ? "FAILED: %d mismatches\n" : "swap/mfma layout OK (%d mismatches)\n", bad);
    return bad != 0;
}
