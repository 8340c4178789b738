// Timing of decode_attn_mfma_kernel at the bench's geometry (see decode_attn_bench.hip for the first version).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../surya_amd/csrc/decode_attn.h"
using namespace sa;
int main(int argc, char** argv) {
    const int slots = 256, nq = 10, nkv = 2, D = 128, Tmax = 144, S = 2, qkv_dim = (nq + 2 * nkv) * D;
    bf16_t *kc, *vc, *out, *bias; float* part; int *active, *row_len; float2* rope;
    size_t kvn = (size_t)slots * nkv * Tmax * D;
    hipMalloc(&kc, kvn * 2 * 16); hipMalloc(&vc, kvn * 2 * 16);
    hipMalloc(&out, (size_t)slots * nq * D * 2); hipMalloc(&bias, qkv_dim * 2);
    hipMalloc(&part, (size_t)8 * slots * qkv_dim * 4); hipMalloc(&active, slots * 4); hipMalloc(&row_len, slots * 4);
    hipMalloc(&rope, (size_t)Tmax * (D / 2) * 8);
    hipMemset(kc, 0, kvn * 2 * 16); hipMemset(vc, 0, kvn * 2 * 16); hipMemset(part, 0, (size_t)8 * slots * qkv_dim * 4);
    hipMemset(bias, 0, qkv_dim * 2); hipMemset(rope, 0, (size_t)Tmax * (D / 2) * 8);
    std::vector<int> a(slots), l(slots);
    for (int i = 0; i < slots; ++i) { a[i] = i; l[i] = 60 + (i * 7) % 50; }
    hipMemcpy(active, a.data(), slots * 4, hipMemcpyHostToDevice); hipMemcpy(row_len, l.data(), slots * 4, hipMemcpyHostToDevice);
    const bool v3 = argc > 1 && argv[1][0] == '3';
    auto kern = v3 ? decode_attn_flash_kernel<128, 5> : decode_attn_mfma_kernel<bf16_t, 128, 5>;
    const size_t lds = v3 ? decode_attn_flash_lds<128, 5>() : decode_attn_mfma_lds<bf16_t, 128, 5>();
    printf("kernel version %d\n", v3 ? 3 : 2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int layer) {
        hipLaunchKernelGGL(kern, dim3(slots, nkv), dim3(256), lds, 0, part, S, bias, out, kc + layer * kvn, vc + layer * kvn, active, row_len,
                           rope, nq, nkv, Tmax, 0.088f);
    };
    for (int i = 0; i < 32; ++i) run(i % 16);
    hipDeviceSynchronize();
    printf("launch status: %s, lds %zu\n", hipGetErrorString(hipGetLastError()), lds);
    hipEventRecord(e0);
    const int iters = 320;
    for (int i = 0; i < iters; ++i) run(i % 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("decode_attn_mfma: %.2f us per launch\n", ms * 1000.f / iters);
#ifdef SA_DA_TIMING
    unsigned long long st[16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(sa_da_stamps), sizeof(st));
    const char* names2[] = {"slot/len loads + K/V fetch issue", "qkv partial loads + reduce -> xrow", "stash + barrier", "RoPE + append + barrier",
                            "scores (MFMA) + barrier", "softmax + barrier", "PV + barrier", "normalise + store"};
    const char* names3[] = {"slot/len loads + K/V LDS-DMA issue", "qkv partial loads + reduce -> xrow", "vmcnt(0) + barrier", "RoPE + append + barrier",
                            "per-wave flash (scores, softmax, PV)", "combine records + barriers", "final combine + store", "-"};
    const char** names = v3 ? names3 : names2;
    for (int i = 0; i < 8; ++i) printf("  phase %d %-40s %6.2f us\n", i, names[i], (double)(st[i + 1] - st[i]) / 100.0);
    printf("  total inside the kernel %.2f us\n", (double)(st[8] - st[0]) / 100.0);
#endif
    return 0;
}
