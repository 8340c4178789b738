// How fast does ONE wave per SIMD issue vector-ALU instructions on gfx950? (round 6: the depthwise producer waves of dwproj_kernel ran at ~7 cycles per
// instruction.) Independent v_fma_f32 / v_pk_fma_f32 / v_and chains, 1..4 waves per SIMD, cycles per instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue.hip -o tools/microbench/valu_issue && tools/microbench/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    float a[16];
    f32x2 p[8];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
    for (int i = 0; i < 8; ++i) p[i] = f32x2{(float)threadIdx.x, (float)i};
    const float s = out[0], t = out[1];
    const f32x2 s2 = f32x2{s, t};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(s2));
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(0xfffffff0u));
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
        } else if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(0x10000u));
        } else if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(u[i]) : "v"(0x01000c0cu));
        } else if (MODE == 6) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(u[i]));
        } else if (MODE == 7) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
        } else if (MODE == 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        } else if (MODE == 9) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(u[i]));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0;
    for (int i = 0; i < 16; ++i) acc += a[i] + (float)u[i];
    for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y;
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, int ninst) {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
    hipMemset(out, 0, 1 << 24);
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int threads = 256 * waves_per_simd;       // one workgroup per CU
        const int iters = 2000;
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * ninst;
        printf("%-14s %d wave(s)/SIMD: %7.2f memtime ticks per instruction per wave | wall %.3f ms -> %.2f ns per instr per wave, %.2f ns per instr per SIMD\n", name,
               waves_per_simd, c / n, ms, ms * 1e6 / n, ms * 1e6 / n / waves_per_simd);
    }
}
int main() {
    run<0>("v_fma_f32", 64); run<1>("v_pk_fma_f32", 32); run<2>("v_and_b32", 64); run<3>("v_lshlrev_b32", 64);
    run<4>("v_mul_u32_u24", 64); run<5>("v_perm_b32", 64); run<6>("v_cvt_pk_bf16", 64); run<7>("v_med3_f32", 64); run<8>("v_mul_f32", 64);
    run<9>("v_mov_sdwa", 64); run<10>("v_dot2c_bf16", 64);
    return 0;
}
