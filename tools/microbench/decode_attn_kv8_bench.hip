// Timing of decode_attn_kv8_kernel at the texify geometry (128 slots x 2 kv heads, head dim 128, G = 5; context given on the command
// line), with per-phase shader-cycle counters when built with -DSA_DA_TIMING:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSA_DA_TIMING -I include tools/microbench/decode_attn_kv8_bench.hip -o tools/microbench/kv8_bench
//   tools/microbench/kv8_bench 586 ; tools/microbench/kv8_bench 969
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../surya_amd/csrc/decode_attn_kv8.h"
using namespace sa;
int main(int argc, char** argv) {
    const int ctx = argc > 1 ? atoi(argv[1]) : 586;
    const int slots = 128, nq = 10, nkv = 2, D = 128, Tmax = 1002, T8 = 1024, S = 2, qkv_dim = (nq + 2 * nkv) * D, layers = 16;
    uint8_t *k8, *v8; float *ks, *vs, *part; bf16_t *out, *bias; int *active, *row_len; float2* rope;
    const size_t rows = (size_t)slots * nkv;
    hipMalloc(&k8, rows * Tmax * D * layers); hipMalloc(&v8, rows * D * T8 * layers);
    hipMalloc(&ks, rows * T8 * 4 * layers); hipMalloc(&vs, rows * T8 * 4 * layers);
    hipMalloc(&out, (size_t)slots * nq * D * 2); hipMalloc(&bias, qkv_dim * 2);
    hipMalloc(&part, (size_t)8 * slots * qkv_dim * 4); hipMalloc(&active, slots * 4); hipMalloc(&row_len, slots * 4);
    hipMalloc(&rope, (size_t)Tmax * (D / 2) * 8);
    hipMemset(k8, 0x38, rows * Tmax * D * layers); hipMemset(v8, 0x38, rows * D * T8 * layers);
    std::vector<float> ones(rows * T8 * layers, 1.0f);
    hipMemcpy(ks, ones.data(), ones.size() * 4, hipMemcpyHostToDevice); hipMemcpy(vs, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
    hipMemset(part, 0, (size_t)8 * slots * qkv_dim * 4); hipMemset(bias, 0, qkv_dim * 2); hipMemset(rope, 0, (size_t)Tmax * (D / 2) * 8);
    std::vector<int> a(slots), l(slots);
    for (int i = 0; i < slots; ++i) { a[i] = i; l[i] = ctx; }
    hipMemcpy(active, a.data(), slots * 4, hipMemcpyHostToDevice); hipMemcpy(row_len, l.data(), slots * 4, hipMemcpyHostToDevice);
    auto kern = decode_attn_kv8_kernel<128, 5>;
    const size_t lds = decode_attn_kv8_lds<128, 5>();
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    auto run = [&](int layer) {
        hipLaunchKernelGGL(kern, dim3(slots, nkv), dim3(KV8_THREADS), lds, 0, part, S, bias, out, k8 + (size_t)layer * rows * Tmax * D,
                           v8 + (size_t)layer * rows * D * T8, ks + (size_t)layer * rows * T8, vs + (size_t)layer * rows * T8, active, row_len, rope, nq,
                           nkv, Tmax, T8, 0.088f, (uint8_t*)nullptr, (uint8_t*)nullptr, 0);
    };
    for (int i = 0; i < 32; ++i) run(i % layers);
    hipDeviceSynchronize();
    printf("context %d, launch status: %s, lds %zu\n", ctx, hipGetErrorString(hipGetLastError()), lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#ifdef SA_DA_TIMING
    unsigned long long zero[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(sa_kv8_cycles), zero, sizeof(zero));
#endif
    hipEventRecord(e0);
    const int iters = 320;
    for (int i = 0; i < iters; ++i) run(i % layers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("decode_attn_kv8: %.2f us per launch\n", ms * 1000.f / iters);
#ifdef SA_DA_TIMING
    unsigned long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(sa_kv8_cycles), sizeof(st));
    const char* names[] = {"prologue (loads, reduce, RoPE, quantise new token)", "loop: s_waitcnt for the tile", "loop: barrier after the wait",
                           "loop: insert + scores + softmax + PV (wave 0)", "loop: barrier after compute", "loop: issue tile t + 2", "tile loop total",
                           "combine + store"};
    for (int i = 0; i < 8; ++i) printf("  %-55s %8.0f cycles per launch\n", names[i], (double)st[i] / iters);
#endif
    return 0;
}
