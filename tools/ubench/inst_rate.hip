// inst_rate.hip - issue cost of the instructions the blend kernels are made of, on gfx950, at 1 .. 4 waves per SIMD.
// Development tool (not part of the product):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/inst_rate.hip -o gpurun_out/inst_rate && gpurun_out/inst_rate
//
// Every kernel runs ROUNDS rounds of an unrolled block of 32 instructions of one kind (eight independent destination registers,
// inline asm so that nothing is merged or reordered) in ONE workgroup per CU of 4 x WPS waves (the workgroup declares 160 KB of
// LDS: one per CU; its waves go to the four SIMDs in turn).  Reported per kind and WPS: cycles per instruction seen by one wave
// (s_memtime around the loop, mean over waves) and SIMD cycles per instruction (wall clock x 2.4 GHz / instructions per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NI = 32;   // instructions per block

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define B32(OP) R8(OP) R8(OP) R8(OP) R8(OP)

// kinds
enum Kind { FMA, MUL_S, EXP, RCP, CMP_VCC, CMP_SGPR, CND_VCC, CND_SGPR, CND_SGPR_FRESH, MIN_LIT, CVT_PK, LSHL, AND_LIT, SUB, XOR_, DS_W16, DS_W16HI, DS_W32,
            DS_R128, DS_R64, CMP3_CND, MED3, MAX_, SALU_AND, NOP_, MBCNT, PERM, CMPX, V_MOV, TRIPLE_ARITH, SLEEP_, PK_FMA, PK_MUL, V_MUL, V_FMAC, CND_E64_VCC, CND_VCC_WRITTEN, DS_ADD_F32, V_ADD_U32, NKIND };
static const char* kind_name[NKIND] = {"v_fma_f32", "v_mul_f32 (sgpr operand)", "v_exp_f32", "v_rcp_f32", "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32_e64 -> sgpr pair",
    "v_cndmask_b32 (vcc, never written)", "v_cndmask_b32_e64 (sgpr pair, never written)", "v_cmp -> s_and_b64 x2 -> v_cndmask (phase-1 masking, per cndmask: 3 cmp + 2 salu + 1 cnd)",
    "v_min_f32 (literal)", "v_cvt_pk_bf16_f32", "v_lshlrev_b32", "v_and_b32 (literal)", "v_sub_f32", "v_xor_b32", "ds_write_b16", "ds_write_b16_d16_hi", "ds_write_b32",
    "ds_read_b128 (broadcast)", "ds_read_b64 (broadcast)", "v_cmp x3 (vcc chain via v_cmp + s_and) + cndmask, 8 per block", "v_med3_f32", "v_max_f32", "s_and_b64", "s_nop 0",
    "v_mbcnt_lo", "v_perm_b32", "v_cmpx_lt_f32 (exec restored per block)", "v_mov_b32", "six plain vector instructions on one chain (per block of six)", "s_sleep 1 (calibration: 64 clocks each)", "v_pk_fma_f32", "v_pk_mul_f32", "v_mul_f32", "v_fmac_f32", "v_cndmask_b32_e64 (vcc as the pair, never written)",
    "v_cndmask_b32 (vcc written by s_mov_b64 in front of the block)", "ds_add_f32 (no return, distinct addresses)", "v_add_u32"};

template <int KIND>
__global__ void __launch_bounds__(1024) k(int rounds, float* out, unsigned long long* cyc, float s_in) {
    extern __shared__ char smem[];
    float x0 = threadIdx.x * 1e-3f + 0.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y = x0 * 0.37f, z = 0.25f;
    float4 r128 = make_float4(0, 0, 0, 0);
    const uint32_t lds_w = (threadIdx.x * 2) & 0x3FFF;                   // distinct halves, 2 lanes per dword
    const uint32_t lds_r = 16384 + (threadIdx.x >> 6) * 64;              // wave-uniform: a broadcast
    unsigned long long mask = 0x5555AAAA5555AAAAull ^ (unsigned long long)rounds;
    asm volatile("" : "+s"(mask));
    asm volatile("" : "+s"(s_in));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < rounds; it++) {
#define XR(i) x##i
        if constexpr (KIND == FMA) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(XR(i)) : "v"(y), "v"(z));
            B32(OP)
#undef OP
        } else if constexpr (KIND == MUL_S) {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(XR(i)) : "s"(s_in));
            B32(OP)
#undef OP
        } else if constexpr (KIND == EXP) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(XR(i)));
            B32(OP)
#undef OP
        } else if constexpr (KIND == RCP) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(XR(i)));
            B32(OP)
#undef OP
        } else if constexpr (KIND == CMP_VCC) {
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(XR(i)), "v"(y) : "vcc");
            B32(OP)
#undef OP
        } else if constexpr (KIND == CMP_SGPR) {
#define OP(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(XR(i)), "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == CND_VCC) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(XR(i)) : "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == CND_SGPR) {
#define OP(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(XR(i)) : "v"(y), "s"(mask));
            B32(OP)
#undef OP
        } else if constexpr (KIND == CND_SGPR_FRESH) {
            // the masking of phase 1, eight times per block: three compares into scalar pairs, two scalar ANDs, one select
#define OP(i) { unsigned long long m1, m2, m3; \
              asm volatile("v_cmp_lt_f32_e64 %0, %3, %4\n\tv_cmp_nlt_f32_e64 %1, 0, %3\n\tv_cmp_ngt_f32_e64 %2, %5, %3\n\ts_and_b64 %0, %0, %1\n\ts_and_b64 %0, %0, %2\n\tv_cndmask_b32_e64 %3, 0, %3, %0" \
                           : "=&s"(m1), "=&s"(m2), "=&s"(m3), "+v"(XR(i)) : "v"(y), "s"(s_in) : "scc"); }
            R8(OP)
#undef OP
        } else if constexpr (KIND == MIN_LIT) {
#define OP(i) asm volatile("v_min_f32 %0, 0x3f7d70a4, %0" : "+v"(XR(i)));
            B32(OP)
#undef OP
        } else if constexpr (KIND == CVT_PK) {
#define OP(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(XR(i)) : "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == LSHL) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(XR(i)));
            B32(OP)
#undef OP
        } else if constexpr (KIND == AND_LIT) {
#define OP(i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(XR(i)));
            B32(OP)
#undef OP
        } else if constexpr (KIND == SUB) {
#define OP(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(XR(i)) : "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == XOR_) {
#define OP(i) asm volatile("v_xor_b32 %0, 16, %0" : "+v"(XR(i)));
            B32(OP)
#undef OP
        } else if constexpr (KIND == DS_W16) {
#define OP(i) asm volatile("ds_write_b16 %0, %1 offset:" #i "*128" :: "v"(lds_w), "v"(XR(i)) : "memory");
            B32(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == DS_W16HI) {
#define OP(i) asm volatile("ds_write_b16_d16_hi %0, %1 offset:" #i "*128" :: "v"(lds_w), "v"(XR(i)) : "memory");
            B32(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == DS_W32) {
#define OP(i) asm volatile("ds_write_b32 %0, %1 offset:" #i "*256" :: "v"(lds_w * 2), "v"(XR(i)) : "memory");
            B32(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == DS_R128) {
#define OP(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*48" : "=v"(r128) : "v"(lds_r) : "memory");
            B32(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            x0 += r128.x;
        } else if constexpr (KIND == DS_R64) {
            float2 r64;
#define OP(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*48" : "=v"(r64) : "v"(lds_r) : "memory");
            B32(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            x0 += r64.x;
        } else if constexpr (KIND == CMP3_CND) {
            // same masking with the three compares through VCC-free e64 forms but only ONE scalar pair live (what a compiler may emit)
#define OP(i) { unsigned long long m1, m2; \
              asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\n\tv_cmp_nlt_f32_e64 %1, 0, %2\n\ts_and_b64 %0, %0, %1\n\tv_cmp_ngt_f32_e64 %1, %4, %2\n\ts_and_b64 %0, %0, %1\n\tv_cndmask_b32_e64 %2, 0, %2, %0" \
                           : "=&s"(m1), "=&s"(m2), "+v"(XR(i)) : "v"(y), "s"(s_in) : "scc"); }
            R8(OP)
#undef OP
        } else if constexpr (KIND == MED3) {
#define OP(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(XR(i)) : "v"(y), "v"(z));
            B32(OP)
#undef OP
        } else if constexpr (KIND == MAX_) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(XR(i)) : "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == SALU_AND) {
#define OP(i) asm volatile("s_and_b64 %0, %0, %0" : "+s"(mask) :: "scc");
            B32(OP)
#undef OP
        } else if constexpr (KIND == SLEEP_) {
#define OP(i) asm volatile("s_sleep 1");
            B32(OP)
#undef OP
        } else if constexpr (KIND == PK_FMA) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, yy = {y, y}, zz = {z, z};
#define OPP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##i) : "v"(yy), "v"(zz));
            OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3)
            OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3)
#undef OPP
            x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
        } else if constexpr (KIND == PK_MUL) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, yy = {y, y};
#define OPP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(yy));
            OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3)
            OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3) OPP(0) OPP(1) OPP(2) OPP(3)
#undef OPP
            x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
        } else if constexpr (KIND == V_MUL) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(XR(i)) : "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == V_FMAC) {
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(XR(i)) : "v"(y), "v"(z));
            B32(OP)
#undef OP
        } else if constexpr (KIND == CND_E64_VCC) {
#define OP(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(XR(i)) : "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == CND_VCC_WRITTEN) {
            asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc");
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(XR(i)) : "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == DS_ADD_F32) {
#define OP(i) asm volatile("ds_add_f32 %0, %1 offset:" #i "*256" :: "v"(lds_w * 2), "v"(XR(i)) : "memory");
            B32(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == V_ADD_U32) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(XR(i)) : "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == NOP_) {
#define OP(i) asm volatile("s_nop 0");
            B32(OP)
#undef OP
        } else if constexpr (KIND == MBCNT) {
#define OP(i) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(XR(i)));
            B32(OP)
#undef OP
        } else if constexpr (KIND == PERM) {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(XR(i)) : "v"(y), "v"(z));
            B32(OP)
#undef OP
        } else if constexpr (KIND == CMPX) {
#define OP(i) asm volatile("v_cmpx_lt_f32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(XR(i)), "v"(y) : "vcc");
            B32(OP)
#undef OP
        } else if constexpr (KIND == V_MOV) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(XR(i)) : "v"(y));
            B32(OP)
#undef OP
        } else if constexpr (KIND == TRIPLE_ARITH) {
            // the same decision without a select: (a) pos test folded into the opacity factor is not possible per lane, so:
            //   t = v_cmp-free form: keep = max(sign tests) ... here only the COST of six plain vector instructions is measured
#define OP(i) asm volatile("v_sub_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(XR(i)) : "v"(y));
            R8(OP)
#undef OP
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    const float sum = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(mask & 1);
    if (sum == 12345.678f) out[threadIdx.x] = sum + smem[threadIdx.x];
}

template <int KIND>
void run(int per_block, float* d_out, unsigned long long* d_cyc, int n_cu) {
    printf("%-100s", kind_name[KIND]);
    for (int wps = 1; wps <= 4; wps++) {
        const int threads = 256 * wps, rounds = 10000;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL((k<KIND>), dim3(n_cu), dim3(threads), 160 * 1024, 0, 50, d_out, d_cyc, 1.0001f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k<KIND>), dim3(n_cu), dim3(threads), 160 * 1024, 0, rounds, d_out, d_cyc, 1.0001f);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> h(n_cu * 16);
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double tot = 0; int nw = 0;
        for (int bI = 0; bI < n_cu; bI++) for (int w = 0; w < 4 * wps; w++) { tot += (double)h[bI * 16 + w]; nw++; }
        const double per_wave = tot / nw / ((double)rounds * per_block);          // counter ticks per instruction, one wave
        const double simd = ms * 1e-3 * 2.4e9 / ((double)rounds * per_block * wps); // SIMD cycles (at 2.4 GHz) per instruction
        printf(" | %d: %6.2f %6.2f", wps, per_wave, simd);
    }
    printf("\n");
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs; columns per waves-per-SIMD: counter ticks per instruction seen by one wave, SIMD cycles per instruction (wall x 2.4 GHz)\n", prop.name, n_cu);
    float* d_out; unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_out, 1024 * 4)); CHECK(hipMalloc(&d_cyc, n_cu * 16 * 8));
    CHECK(hipMemset(d_cyc, 0, n_cu * 16 * 8));
    run<SLEEP_>(NI, d_out, d_cyc, n_cu);
    run<FMA>(NI, d_out, d_cyc, n_cu);
    run<PK_FMA>(NI, d_out, d_cyc, n_cu);
    run<PK_MUL>(NI, d_out, d_cyc, n_cu);
    run<MUL_S>(NI, d_out, d_cyc, n_cu);
    run<SUB>(NI, d_out, d_cyc, n_cu);
    run<V_MOV>(NI, d_out, d_cyc, n_cu);
    run<EXP>(NI, d_out, d_cyc, n_cu);
    run<RCP>(NI, d_out, d_cyc, n_cu);
    run<CMP_VCC>(NI, d_out, d_cyc, n_cu);
    run<CMP_SGPR>(NI, d_out, d_cyc, n_cu);
    run<CND_VCC>(NI, d_out, d_cyc, n_cu);
    run<CND_SGPR>(NI, d_out, d_cyc, n_cu);
    run<CND_SGPR_FRESH>(8, d_out, d_cyc, n_cu);
    run<CMP3_CND>(8, d_out, d_cyc, n_cu);
    run<TRIPLE_ARITH>(8, d_out, d_cyc, n_cu);
    run<CMPX>(NI, d_out, d_cyc, n_cu);
    run<MIN_LIT>(NI, d_out, d_cyc, n_cu);
    run<MED3>(NI, d_out, d_cyc, n_cu);
    run<MAX_>(NI, d_out, d_cyc, n_cu);
    run<CVT_PK>(NI, d_out, d_cyc, n_cu);
    run<LSHL>(NI, d_out, d_cyc, n_cu);
    run<AND_LIT>(NI, d_out, d_cyc, n_cu);
    run<XOR_>(NI, d_out, d_cyc, n_cu);
    run<MBCNT>(NI, d_out, d_cyc, n_cu);
    run<PERM>(NI, d_out, d_cyc, n_cu);
    run<V_MUL>(NI, d_out, d_cyc, n_cu);
    run<V_FMAC>(NI, d_out, d_cyc, n_cu);
    run<V_ADD_U32>(NI, d_out, d_cyc, n_cu);
    run<CND_E64_VCC>(NI, d_out, d_cyc, n_cu);
    run<CND_VCC_WRITTEN>(NI, d_out, d_cyc, n_cu);
    run<SALU_AND>(NI, d_out, d_cyc, n_cu);
    run<NOP_>(NI, d_out, d_cyc, n_cu);
    run<DS_W16>(NI, d_out, d_cyc, n_cu);
    run<DS_W16HI>(NI, d_out, d_cyc, n_cu);
    run<DS_W32>(NI, d_out, d_cyc, n_cu);
    run<DS_ADD_F32>(NI, d_out, d_cyc, n_cu);
    run<DS_R128>(NI, d_out, d_cyc, n_cu);
    run<DS_R64>(NI, d_out, d_cyc, n_cu);
    return 0;
}
