// Micro-benchmark: how long does a kernel take whose waves do (almost) nothing, as a function of launch geometry,
// VGPR allocation and LDS allocation? Separates dispatch/teardown cost from the kernels' own work (r01 decode analysis).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int VG>
__global__ __launch_bounds__(512) void k_regs(float* out, int n) {
    float v[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < VG; ++i) asm volatile("" : "+v"(v[i]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += v[i];
    if (n < 0) out[threadIdx.x] = s;
}

__global__ void k_lds(float* out, int n) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (n < 0) out[threadIdx.x] = sm[(threadIdx.x + 1) % blockDim.x];
}

__global__ void k_memchain(const float* in, float* out, int hops, int stride) {
    // each thread does `hops` dependent loads (pointer-free: index from value), measures latency chains
    int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    float acc = 0.f;
    for (int h = 0; h < hops; ++h) {
        float v = in[idx];
        acc += v;
        idx = (idx + stride + (int)v) & ((1 << 24) - 1);
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename F>
float time_it(F f, int iters = 200) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / iters;
}

int main() {
    float *out, *in;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&in, (size_t)64 << 20);
    hipMemset(in, 0, (size_t)64 << 20);
    printf("geometry, us_per_launch (back-to-back on one stream)\n");
    for (int blocks : {1, 64, 224, 256, 512, 1024, 2048}) {
        printf("empty256  blocks=%4d : %.2f\n", blocks, time_it([&] { hipLaunchKernelGGL(k_regs<8>, dim3(blocks), dim3(256), 0, 0, out, 0); }));
    }
    for (int blocks : {224, 512, 1024}) {
        printf("regs64  256thr blocks=%4d : %.2f\n", blocks, time_it([&] { hipLaunchKernelGGL(k_regs<64>, dim3(blocks), dim3(256), 0, 0, out, 0); }));
        printf("regs160 256thr blocks=%4d : %.2f\n", blocks, time_it([&] { hipLaunchKernelGGL(k_regs<160>, dim3(blocks), dim3(256), 0, 0, out, 0); }));
        printf("regs100 512thr blocks=%4d : %.2f\n", blocks, time_it([&] { hipLaunchKernelGGL(k_regs<100>, dim3(blocks), dim3(512), 0, 0, out, 0); }));
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (int blocks : {224, 512}) {
        for (int kb : {1, 17, 72}) {
            printf("lds %2dKB 512thr blocks=%4d : %.2f\n", kb, blocks,
                   time_it([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(512), kb * 1024, 0, out, 0); }));
        }
    }
    for (int hops : {1, 2, 4, 8}) {
        printf("memchain hops=%d 512 blocks x256 (stride 1MB, HBM-ish): %.2f\n", hops,
               time_it([&] { hipLaunchKernelGGL(k_memchain, dim3(512), dim3(256), 0, 0, in, out, hops, 262144 + 64); }));
    }
    return 0;
}
