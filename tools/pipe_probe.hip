// pipe_probe.hip - does the fp32 matrix pipe of gfx950 run beside the vector pipe of the same SIMD, or instead of it?
// Development tool (not part of the product): hipcc --offload-arch=gfx950 -O3 tools/pipe_probe.hip -o gpurun_out/pipe_probe
//
// Every kernel runs `iters` rounds of a fixed instruction mix in workgroups of 4 or 8 waves (one or two waves per SIMD), one workgroup
// per CU.  Reported: SIMD cycles per round (wall time x clock / rounds).  If the two pipes are independent, mode "valu | mfma on
// two waves of a SIMD" costs max(valu, mfma); if they share the datapath it costs the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NV = 32;      // vector instructions per round
constexpr int NM = 8;       // matrix instructions per round

__device__ __forceinline__ void valu_round(float (&x)[8], float m, float c) {
#pragma unroll
    for (int k = 0; k < NV; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(m), "v"(c));     // (plain C gets packed in pairs)
}
__device__ __forceinline__ void exp_round(float (&x)[8]) {
#pragma unroll
    for (int k = 0; k < NV; k++) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k & 7]));
}
__device__ __forceinline__ void cnd_round(float (&x)[8], float m) {
#pragma unroll
    for (int k = 0; k < NV; k++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k & 7]) : "v"(m));
}
__device__ __forceinline__ void pk_round(f32x2 (&x)[8], f32x2 m, f32x2 c) {
#pragma unroll
    for (int k = 0; k < NV; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(m), "v"(c));
}
__device__ __forceinline__ void mfma16_round(f32x4 (&acc)[4], float a, float b) {
#pragma unroll
    for (int k = 0; k < NM; k++) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k & 3], 0, 0, 0);
}
__device__ __forceinline__ void mfma32_round(f32x16 (&acc)[2], float a, float b) {
#pragma unroll
    for (int k = 0; k < NM / 2; k++) acc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 1], 0, 0, 0);
}
__device__ __forceinline__ void mfma4_round(f32x4 (&acc)[4], float a, float b) {
#pragma unroll
    for (int k = 0; k < NM * 4; k++) acc[k & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k & 3], 0, 0, 0);
}
__device__ __forceinline__ void mfma_bf16_round(f32x4 (&acc)[4], bf16x8 a, bf16x8 b) {
#pragma unroll
    for (int k = 0; k < NM; k++) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 3], 0, 0, 0);
}

// what: 0 = vector, 1 = fp32 16x16x4, 2 = fp32 32x32x2, 3 = fp32 4x4x1, 4 = bf16 16x16x32, 5 = packed vector, 6 = nothing
template <int what>
__device__ __forceinline__ void run(int iters, float* out, unsigned long long* cyc) {
    float x[8]; f32x2 px[8]; f32x4 acc[4]; f32x16 acc32[2];
    const float seed = (float)threadIdx.x * 1e-3f;
    for (int k = 0; k < 8; k++) { x[k] = seed + k; px[k] = f32x2{seed, seed + k}; }
    for (int k = 0; k < 4; k++) acc[k] = f32x4{seed, 0, 0, 0};
    for (int k = 0; k < 2; k++) for (int r = 0; r < 16; r++) acc32[k][r] = seed;
    bf16x8 ba, bb;
    for (int k = 0; k < 8; k++) { ba[k] = (__bf16)seed; bb[k] = (__bf16)(seed + 1.f); }
    const float m = 0.999f + seed * 1e-6f, c = 1e-3f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if constexpr (what == 0) valu_round(x, m, c);
        if constexpr (what == 1) mfma16_round(acc, m, c);
        if constexpr (what == 2) mfma32_round(acc32, m, c);
        if constexpr (what == 3) mfma4_round(acc, m, c);
        if constexpr (what == 4) mfma_bf16_round(acc, ba, bb);
        if constexpr (what == 5) pk_round(px, f32x2{m, m}, f32x2{c, c});
        if constexpr (what == 6) asm volatile("s_nop 0");
        if constexpr (what == 7) exp_round(x);
        if constexpr (what == 8) cnd_round(x, m);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0.f;
    for (int k = 0; k < 8; k++) s += x[k] + px[k][0] + px[k][1];
    for (int k = 0; k < 4; k++) s += acc[k][0] + acc[k][3];
    for (int k = 0; k < 2; k++) s += acc32[k][0] + acc32[k][15];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// one role per SIMD slot: waves 0-3 run `w0`, waves 4-7 (the second wave of each SIMD) run `w1`
template <int W0, int W1>
__global__ void __launch_bounds__(512) probe(int iters, float* out, unsigned long long* cyc) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) run<W0>(iters, out, cyc); else run<W1>(iters, out, cyc);
}
// both mixes in ONE wave, round by round
__global__ void __launch_bounds__(256) probe_same(int iters, float* out, unsigned long long* cyc) {
    float x[8]; f32x4 acc[4];
    const float seed = (float)threadIdx.x * 1e-3f;
    for (int k = 0; k < 8; k++) x[k] = seed + k;
    for (int k = 0; k < 4; k++) acc[k] = f32x4{seed, 0, 0, 0};
    const float m = 0.999f + seed * 1e-6f, c = 1e-3f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < NM; k++) {
            acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(m, c, acc[k & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV / NM; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(4 * k + j) & 7]) : "v"(m), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0.f;
    for (int k = 0; k < 8; k++) s += x[k];
    for (int k = 0; k < 4; k++) s += acc[k][0] + acc[k][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

static double time_ms(void (*launch)(void*), void* ctx) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(ctx); hipDeviceSynchronize();
    hipEventRecord(a); launch(ctx); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
typedef void (*KernelFn)(int, float*, unsigned long long*);
struct Ctx { KernelFn fn; int iters, threads, same; float* out; int blocks; unsigned long long* cyc; };
static void do_launch(void* p) {
    Ctx* c = (Ctx*)p;
    if (c->same) hipLaunchKernelGGL(probe_same, dim3(c->blocks), dim3(256), 0, 0, c->iters, c->out, c->cyc);
    else hipLaunchKernelGGL(c->fn, dim3(c->blocks), dim3(c->threads), 0, 0, c->iters, c->out, c->cyc);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;     // kHz -> GHz
    const int cus = prop.multiProcessorCount;
    float* out; hipMalloc(&out, 4096);
    unsigned long long* cyc; hipMalloc(&cyc, 8 * 8 * 1024);
    std::vector<unsigned long long> hc(8 * 1024);
    const int iters = 20000;
    printf("device %s, %d CUs, %.2f GHz (nominal); %d vector / %d matrix instructions per round; one workgroup per CU\n", prop.name, cus, ghz, NV, NM);
    const char* names[] = {"v_fma_f32 x32", "mfma_f32_16x16x4 x8", "mfma_f32_32x32x2 x4", "mfma_f32_4x4x1 x32", "mfma_f32_16x16x32_bf16 x8", "v_pk_fma_f32 x32", "idle"};
    struct Case { KernelFn fn; int threads, same; const char* label; };
    std::vector<Case> cases = {
        {probe<0, 6>, 256, 0, "one wave per SIMD: vector"}, {probe<5, 6>, 256, 0, "one wave per SIMD: packed vector"},
        {probe<1, 6>, 256, 0, "one wave per SIMD: fp32 16x16x4"}, {probe<2, 6>, 256, 0, "one wave per SIMD: fp32 32x32x2"},
        {probe<3, 6>, 256, 0, "one wave per SIMD: fp32 4x4x1"}, {probe<4, 6>, 256, 0, "one wave per SIMD: bf16 16x16x32"},
        {probe<0, 0>, 512, 0, "two waves per SIMD: vector | vector"}, {probe<1, 1>, 512, 0, "two waves per SIMD: fp32 16x16x4 | same"},
        {probe<0, 1>, 512, 0, "two waves per SIMD: vector | fp32 16x16x4"}, {probe<0, 2>, 512, 0, "two waves per SIMD: vector | fp32 32x32x2"},
        {probe<0, 3>, 512, 0, "two waves per SIMD: vector | fp32 4x4x1"}, {probe<0, 4>, 512, 0, "two waves per SIMD: vector | bf16 16x16x32"},
        {probe<5, 1>, 512, 0, "two waves per SIMD: packed vector | fp32 16x16x4"}, {probe<5, 4>, 512, 0, "two waves per SIMD: packed vector | bf16 16x16x32"},
        {probe<7, 6>, 256, 0, "one wave per SIMD: v_exp_f32"}, {probe<8, 6>, 256, 0, "one wave per SIMD: v_cndmask_b32"},
        {probe<7, 7>, 512, 0, "two waves per SIMD: v_exp_f32 | same"}, {probe<8, 8>, 512, 0, "two waves per SIMD: v_cndmask_b32 | same"},
        {probe<7, 1>, 512, 0, "two waves per SIMD: v_exp_f32 | fp32 16x16x4"}, {probe<8, 1>, 512, 0, "two waves per SIMD: v_cndmask_b32 | fp32 16x16x4"},
        {nullptr, 256, 1, "one wave per SIMD: vector and fp32 16x16x4 interleaved in the wave"},
    };
    for (auto& cs : cases) {
        Ctx c{cs.fn, iters, cs.threads, cs.same, out, cus, cyc};
        hipMemset(cyc, 0, 8 * 8 * 1024);
        const double ms = time_ms(do_launch, &c);
        hipMemcpy(hc.data(), cyc, 8 * 8 * 1024, hipMemcpyDeviceToHost);
        // shader-clock cycles (s_memtime) of the slowest wave of each role
        unsigned long long m0 = 0, m1 = 0;
        for (int b = 0; b < cus; b++) for (int w = 0; w < 8; w++) { auto v = hc[b * 8 + w]; if (w < 4) m0 = v > m0 ? v : m0; else m1 = v > m1 ? v : m1; }
        printf("%-72s %8.3f ms  %8.1f | %8.1f counter ticks/round (first | second wave of a SIMD)\n", cs.label, ms, (double)m0 / iters, (double)m1 / iters);
    }
    (void)names;
    return 0;
}
