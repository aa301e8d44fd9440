// Micro-benchmark (development aid): VALU issue rate per SIMD on gfx950 for a few instruction kinds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND, int CHAINS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s) {
    float v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) v[c] = threadIdx.x * 0.001f + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (KIND == 0) v[c] = fmaf(v[c], s, 0.5f);                         // v_fma / v_fmac
            if (KIND == 1) v[c] = v[c] * s;                                    // v_mul
            if (KIND == 2) v[c] = __builtin_amdgcn_exp2f(v[c]);               // v_exp_f32
            if (KIND == 3) v[c] = __builtin_amdgcn_rcpf(v[c]);                // v_rcp_f32
            if (KIND == 4) v[c] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[c]), 0x111, 0xf, 0xf, true));  // add dpp
            if (KIND == 5) v[c] = v[c] > s ? v[c] * s : v[c] + s;             // cmp + cndmask-ish
            if (KIND == 6) v[c] = fminf(v[c] + s, 3.0f);                       // add + min
            if (KIND == 7) { float t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(1.0f), __float_as_int(v[c]), 0x111, 0xf, 0xf, false)); v[c] *= t; }  // mov_dpp(identity 1) + mul
            if (KIND == 8) asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[c]));   // fused dpp mul (with its own 2 wait states)
            if (KIND == 9) v[c] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[c]), i & 63));   // v_readlane (SGPR lane) + add
            if (KIND == 10) v[c] = (threadIdx.x & 63) == (i & 63) ? s : v[c];   // cmp_eq + cndmask (state write-back pattern)
            if (KIND == 12 && (c & 1) == 0) {   // v_pk_fma_f32 on register pairs (counts as ONE instruction per pair)
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 x = {v[c], v[c + 1]}; const f2 ss = {s, s}; const f2 h = {0.5f, 0.25f};
                x = __builtin_elementwise_fma(x, ss, h); v[c] = x.x; v[c + 1] = x.y;
            }
            if (KIND == 13 && (c & 1) == 0) {   // v_pk_mul_f32
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 x = {v[c], v[c + 1]}; const f2 ss = {s, s * 1.0001f};
                x = x * ss; v[c] = x.x; v[c + 1] = x.y;
            }
            if (KIND == 11) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v[c]), __float_as_int(v[(c + 1) % CHAINS]), false, false); v[c] = __int_as_float(r[0]) + 1.0f; }
        }
    }
    float acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc += v[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND, int CHAINS>
int run(const char* name, float* d, int blocks) {
    const int iters = 4096;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(blocks), dim3(256), 0, 0, d, 16, 1.0001f);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double waves = blocks * 4.0, insts = waves * iters * (double)CHAINS;
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024.0;
    printf("%-14s chains=%d blocks=%5d: %.3f ms, %.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, CHAINS, blocks, ms,
           simd_cycles / insts);
    return 0;
}

int main() {
    float* d; CHECK(hipMalloc(&d, 8192 * 256 * 4));
    for (int blocks : {1024, 2048, 8192}) {
        run<0, 1>("fma", d, blocks); run<0, 4>("fma", d, blocks); run<0, 8>("fma", d, blocks);
    }
    run<1, 8>("mul", d, 4096); run<2, 8>("exp2", d, 4096); run<3, 8>("rcp", d, 4096);
    run<4, 8>("add_dpp", d, 4096); run<5, 8>("cmp+sel", d, 4096); run<6, 8>("add+min", d, 4096);
    run<7, 8>("movdpp+mul", d, 4096); run<8, 8>("asm mul_dpp", d, 4096); run<8, 2>("asm mul_dpp", d, 4096); run<9, 8>("readlane+add", d, 4096); run<10, 8>("cmpeq+cnd", d, 4096); run<11, 8>("permswap+add", d, 4096);
    run<12, 8>("pk_fma (x2 work per inst; cycles per PAIR-inst = 2x shown)", d, 4096); run<13, 8>("pk_mul (same)", d, 4096);
    run<12, 16>("pk_fma 16", d, 4096);
    return 0;
}
