// blend_stream.hip - what the vector pipe allows the blend backward: phase 1 of the pixel-lane kernel - the code of
// csrc/pl_phase1.h as the kernel compiles it - run by itself: no staging, no barriers, no matrix phase, no flush, every wave
// of the SIMD in phase 1 all the time.  Development tool (not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I feature-3dgs_amd/csrc tools/ubench/blend_stream.hip -o tools/ubench/blend_stream
//   tools/ubench/blend_stream [json]
// Output per shape (first channel window GEO / later window) and schedule, at 1 .. 4 workgroups of four waves per CU (= waves
// per SIMD): counter ticks per (entry, wave) seen by a wave, and SIMD cycles per (entry, wave) from the wall clock at 2.4 GHz.
// bench.py turns the 4-waves-per-SIMD figure into `roofline_compute.floor_ms`: evaluations of a launch x cycles / (SIMDs x clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pl_phase1.h"
#include "fwd_group.h"

using namespace f3dgs;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool GEO, int SCHED, bool NOLDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) stream(int chunks, unsigned long long* cyc, float* out, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PlRec* const rec = reinterpret_cast<PlRec*>(smem);               // sixteen records
    char* const tiles = smem + 1024;                                    // four quadrants x 8 KB
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (threadIdx.x < 16) {
        const float k = (float)threadIdx.x;
        rec[threadIdx.x].q0 = make_float4(3.0f + 0.4f * k, 4.0f - 0.2f * k, 0.02f * CONIC_SCALE_AC, 0.004f * CONIC_SCALE_B);
        rec[threadIdx.x].q1 = make_float4(0.03f * CONIC_SCALE_AC, 0.5f + 0.02f * k, 0.f, 0.f);
        rec[threadIdx.x].q2 = make_float4(0.3f, 0.5f, 0.7f, 4.0f + k);
    }
    __syncthreads();
    P1Pixel px;
    px.pxf = (float)(lane & 7) + seed; px.pyf = (float)(lane >> 3); px.last = 1000000u - (uint32_t)lane;
    px.dR = 1e-6f * (float)(lane + 1); px.dG = -2e-6f; px.dB = 3e-6f; px.dD = seed * 1e-7f;
    const uint32_t sofs = (uint32_t)(q * BF_QUAD + (lane >> 3) * 16 + (lane & 7) * 2);
    float T = 0.f, S = 0.f;
    uint32_t tm = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int c = 0; c < chunks; c++) {
        T = 0.25f + seed; S = 1e-6f;                         // (bounded values: the walk of a real tile ends after ~10 chunks)
        tm |= pl_phase1_bf16<GEO, SCHED, NOLDS>(rec, px, 2000000u - 16u * (uint32_t)(c & 1023), T, S, tiles, sofs);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 4 + q] = t1 - t0;
    if (T + S == 12345.678f || tm == 0x12345u) out[threadIdx.x] = T + S;
}

// The blend forward's group step (csrc/fwd_group.h: two entries per step, 32 channels, one quadrant per wave - the c3 shape)
// over a chunk of 32 staged entries that all reach the quadrant and blend at every pixel, again and again.
template <bool MFMA>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) stream_fwd(int chunks, unsigned long long* cyc, float* out, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    struct Chunk { FwdEntry ent[32]; float feat[32 * 32]; };
    Chunk& ck = reinterpret_cast<Chunk*>(smem)[q];
    if (lane < 32) {
        const float k = (float)lane;
        FwdEntry e;
        e.geo = make_float4(3.0f + 0.1f * k, 4.0f - 0.05f * k, 0.02f * CONIC_SCALE_AC, 0.004f * CONIC_SCALE_B);
        e.cd = make_float4(0.3f, 0.5f, 0.7f, 4.0f + k);
        e.co_c = 0.03f * CONIC_SCALE_AC; e.co_o = 0.02f;
        e.pos = (uint32_t)lane + 1; e.id = (uint32_t)lane;
        ck.ent[lane] = e;
    }
    for (int i = lane; i < 32 * 32; i += 64) ck.feat[i] = 1e-3f * (float)(i & 63);
    __syncthreads();
    FwdPixels<32, 1> px;
    px.pxf[0] = (float)(lane & 7) + seed; px.pyf[0] = (float)(lane >> 3);
    px.T[0] = 1.f; px.col[0][0] = px.col[0][1] = px.col[0][2] = 0.f; px.dep[0] = 0.f; px.last[0] = 0;
    for (int h = 0; h < 2; h++) for (int r = 0; r < 16; r++) px.acc[0][h][0][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int c = 0; c < chunks; c++) {
        px.T[0] = 1.0f + seed;
#pragma unroll 1
        for (int j = 0; j < 32; j += 2) fwd_blend_group<32, 1, 2, true>(ck.ent, ck.feat, reinterpret_cast<const float*>(&ck), j, lane, 0, 0, !MFMA, px);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 4 + q] = t1 - t0;
    float sum = px.T[0] + px.col[0][0] + px.col[0][1] + px.col[0][2] + px.dep[0] + (float)px.last[0];
    for (int h = 0; h < 2; h++) for (int r = 0; r < 16; r++) sum += px.acc[0][h][0][r];
    if (sum == 12345.678f) out[threadIdx.x] = sum;
}

struct Res { double ticks, simd; };

template <bool MFMA>
Res run_fwd(int wps, int n_cu, unsigned long long* d_cyc, float* d_out) {
    const int chunks = 2000;
    const size_t lds = 160 * 1024 / wps - 256;
    auto kern = stream_fwd<MFMA>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(n_cu * wps), dim3(256), lds, 0, 20, d_cyc, d_out, 0.f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(n_cu * wps), dim3(256), lds, 0, chunks, d_cyc, d_out, 0.f);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h((size_t)n_cu * wps * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double tot = 0;
    for (auto v : h) tot += (double)v;
    Res r;
    r.ticks = tot / h.size() / ((double)chunks * 32);
    r.simd = ms * 1e-3 * 2.4e9 / ((double)chunks * 32 * wps);
    return r;
}

template <bool GEO, int SCHED, bool NOLDS>
Res run(int wps, int n_cu, unsigned long long* d_cyc, float* d_out) {
    const int chunks = 4000;
    const size_t lds = 160 * 1024 / wps - 256;                          // exactly `wps` workgroups fit a CU
    auto kern = stream<GEO, SCHED, NOLDS>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(n_cu * wps), dim3(256), lds, 0, 20, d_cyc, d_out, 0.f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(n_cu * wps), dim3(256), lds, 0, chunks, d_cyc, d_out, 0.f);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h((size_t)n_cu * wps * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double tot = 0;
    for (auto v : h) tot += (double)v;
    Res r;
    r.ticks = tot / h.size() / ((double)chunks * 16);
    r.simd = ms * 1e-3 * 2.4e9 / ((double)chunks * 16 * wps);
    return r;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool json = argc > 1 && !strcmp(argv[1], "json");
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long* d_cyc; float* d_out;
    CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 4 * 4 * 8)); CHECK(hipMalloc(&d_out, 4096));
    Res r[2][2][2][4];       // [GEO][SCHED][NOLDS][wps - 1]
    for (int wps = 1; wps <= 4; wps++) {
        r[1][0][0][wps - 1] = run<true, 0, false>(wps, n_cu, d_cyc, d_out);
        r[1][1][0][wps - 1] = run<true, 1, false>(wps, n_cu, d_cyc, d_out);
        r[1][0][1][wps - 1] = run<true, 0, true>(wps, n_cu, d_cyc, d_out);
        r[1][1][1][wps - 1] = run<true, 1, true>(wps, n_cu, d_cyc, d_out);
        r[0][0][0][wps - 1] = run<false, 0, false>(wps, n_cu, d_cyc, d_out);
        r[0][1][0][wps - 1] = run<false, 1, false>(wps, n_cu, d_cyc, d_out);
        r[0][0][1][wps - 1] = run<false, 0, true>(wps, n_cu, d_cyc, d_out);
        r[0][1][1][wps - 1] = run<false, 1, true>(wps, n_cu, d_cyc, d_out);
    }
    Res rf[2][4];
    for (int wps = 1; wps <= 4; wps++) { rf[0][wps - 1] = run_fwd<false>(wps, n_cu, d_cyc, d_out); rf[1][wps - 1] = run_fwd<true>(wps, n_cu, d_cyc, d_out); }
    if (json) {
        printf("{\"device\": \"%s\", \"cus\": %d, \"unit\": \"SIMD cycles per (entry, wave of 64 pixel lanes) at 2.4 GHz, wall clock\"", prop.name, n_cu);
        const char* gn[2] = {"later_window", "first_window"};
        for (int g = 0; g < 2; g++) for (int s = 0; s < 2; s++) for (int n = 0; n < 2; n++) {
            printf(", \"%s_sched%d%s\": [", gn[g], s, n ? "_nolds" : "");
            for (int w = 0; w < 4; w++) printf("%s%.2f", w ? ", " : "", r[g][s][n][w].simd);
            printf("]");
        }
        for (int mf = 0; mf < 2; mf++) {
            printf(", \"forward_c32_%s\": [", mf ? "with_matrix" : "vector_only");
            for (int w = 0; w < 4; w++) printf("%s%.2f", w ? ", " : "", rf[mf][w].simd);
            printf("]");
        }
        printf("}\n");
        return 0;
    }
    printf("%s, %d CUs.  Phase 1 of the pixel-lane blend backward alone; per (entry, wave): counter ticks seen by a wave | SIMD cycles (wall x 2.4 GHz / waves per SIMD)\n", prop.name, n_cu);
    const char* gn[2] = {"later window (weights only)", "first window (weights + dL/dalpha)"};
    for (int g = 1; g >= 0; g--) for (int n = 0; n < 2; n++) for (int s = 0; s < 2; s++) {
        printf("%-36s %-22s schedule %d:", gn[g], n ? "vector stream only" : "with its LDS traffic", s);
        for (int w = 0; w < 4; w++) printf("  %d/SIMD: %6.1f | %6.1f", w + 1, r[g][s][n][w].ticks, r[g][s][n][w].simd);
        printf("\n");
    }
    printf("Group step of the blend forward alone (32 channels, every entry blending at every pixel); per (entry, wave)\n");
    for (int mf = 0; mf < 2; mf++) {
        printf("%-36s %-22s            :", "forward, first window", mf ? "with the fp32 matrix instructions" : "vector pipe only");
        for (int w = 0; w < 4; w++) printf("  %d/SIMD: %6.1f | %6.1f", w + 1, rf[mf][w].ticks, rf[mf][w].simd);
        printf("\n");
    }
    return 0;
}
