// Standalone timing of the bf16 decode-attention kernels at the bench's geometry (256 slots x 2 kv heads, G = 5, D = 128, 2 qkv slabs).
//   decode_attn2_bench <version 2|3|4|5> [first context = 60] [context span = 50] [cold = 1] [slots = 256]
// version 2 = decode_attn_mfma_kernel (fp32-mode structure on bf16), 3 = decode_attn_flash_kernel (round 2-3), 4 = decode_attn_flash2_kernel
// (round 4), 5 = the same with two K/V tile buffers (long contexts; texify runs 128 slots). K / V / slabs hold a deterministic pattern and
// a checksum of the output is printed, so versions 4 and 5 can be compared bit for bit. Slot i holds ctx0 + (7 i) % span cached keys. `cold`: 16 layer-sized K/V caches are cycled (352 MB at 110 keys: past the
// 256 MB infinity cache), as in a real decode step; 0 re-reads one layer's cache (cache-warm).
// -DSA_DA_TIMING prints the in-kernel phase times of workgroup (64, 0) (100 MHz wall clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../surya_amd/csrc/decode_attn.h"
using namespace sa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int nq = 10, nkv = 2, D = 128, Tmax = 1024, S = 2, qkv_dim = (nq + 2 * nkv) * D;
    const int slots = argc > 5 ? atoi(argv[5]) : 256;
    const int ver = argc > 1 ? atoi(argv[1]) : 4;
    const int ctx0 = argc > 2 ? atoi(argv[2]) : 60, span = argc > 3 ? atoi(argv[3]) : 50, cold = argc > 4 ? atoi(argv[4]) : 1;
    const int layers = cold ? 16 : 1;
    bf16_t *kc, *vc, *out, *bias; float* part; int *active, *row_len; float2* rope;
    const size_t kvn = (size_t)slots * nkv * Tmax * D;
    CK(hipMalloc(&kc, kvn * 2 * layers)); CK(hipMalloc(&vc, kvn * 2 * layers));
    CK(hipMalloc(&out, (size_t)slots * nq * D * 2)); CK(hipMalloc(&bias, qkv_dim * 2));
    CK(hipMalloc(&part, (size_t)8 * slots * qkv_dim * 4)); CK(hipMalloc(&active, slots * 4)); CK(hipMalloc(&row_len, slots * 4));
    CK(hipMalloc(&rope, (size_t)Tmax * (D / 2) * 8));
    CK(hipMemset(kc, 0, kvn * 2 * layers)); CK(hipMemset(vc, 0, kvn * 2 * layers)); CK(hipMemset(part, 0, (size_t)8 * slots * qkv_dim * 4));
    CK(hipMemset(bias, 0, qkv_dim * 2)); CK(hipMemset(rope, 0, (size_t)Tmax * (D / 2) * 8));
    {   // deterministic, non-trivial operands (layer 0's caches, the slabs, the rotary table): bf16 values k / 64 with |k| <= 96
        std::vector<unsigned short> hk(kvn), hv(kvn);
        auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
        for (size_t i = 0; i < kvn; ++i) { hk[i] = bf((float)((int)((i * 2654435761u) >> 20 & 127) - 64) / 64.f); hv[i] = bf((float)((int)((i * 40503u) >> 7 & 127) - 64) / 64.f); }
        for (int ly = 0; ly < layers; ++ly) { CK(hipMemcpy(kc + (size_t)ly * kvn, hk.data(), kvn * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vc + (size_t)ly * kvn, hv.data(), kvn * 2, hipMemcpyHostToDevice)); }
        std::vector<float> hp((size_t)S * slots * qkv_dim);
        for (size_t i = 0; i < hp.size(); ++i) hp[i] = (float)((int)((i * 97u) % 61) - 30) / 32.f;
        CK(hipMemcpy(part, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hr((size_t)Tmax * (D / 2) * 2);
        for (int t = 0; t < Tmax; ++t) for (int i = 0; i < D / 2; ++i) { const float ang = t * powf(10000.f, -2.f * i / D); hr[((size_t)t * (D / 2) + i) * 2] = cosf(ang); hr[((size_t)t * (D / 2) + i) * 2 + 1] = sinf(ang); }
        CK(hipMemcpy(rope, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    }
    std::vector<int> a(slots), l(slots);
    for (int i = 0; i < slots; ++i) { a[i] = i; l[i] = ctx0 + (i * 7) % (span > 0 ? span : 1); }
    CK(hipMemcpy(active, a.data(), slots * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(row_len, l.data(), slots * 4, hipMemcpyHostToDevice));
    printf("kernel version %d, context %d..%d, %s K/V\n", ver, ctx0, ctx0 + (span > 0 ? span - 1 : 0), cold ? "cold" : "warm");
    size_t lds = 0;
    if (ver == 5) { lds = decode_attn_flash2_lds<128, 5, true>(); CK(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_attn_flash2_kernel<128, 5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
    else if (ver == 4) { lds = decode_attn_flash2_lds<128, 5>(); CK(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_attn_flash2_kernel<128, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
    else if (ver == 3) { lds = decode_attn_flash_lds<128, 5>(); CK(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_attn_flash_kernel<128, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
    else { lds = decode_attn_mfma_lds<bf16_t, 128, 5>(); CK(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_attn_mfma_kernel<bf16_t, 128, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
    auto run = [&](int layer) {
        bf16_t *k = kc + (size_t)layer * kvn, *v = vc + (size_t)layer * kvn;
        if (ver == 5) hipLaunchKernelGGL((decode_attn_flash2_kernel<128, 5, true>), dim3(slots, nkv), dim3(256), lds, 0, part, S, bias, out, k, v, active, row_len, rope, nq, nkv, Tmax, 0.088f, (uint8_t*)nullptr, (uint8_t*)nullptr, 0);
        else if (ver == 4) hipLaunchKernelGGL((decode_attn_flash2_kernel<128, 5>), dim3(slots, nkv), dim3(256), lds, 0, part, S, bias, out, k, v, active, row_len, rope, nq, nkv, Tmax, 0.088f, (uint8_t*)nullptr, (uint8_t*)nullptr, 0);
        else if (ver == 3) hipLaunchKernelGGL((decode_attn_flash_kernel<128, 5>), dim3(slots, nkv), dim3(256), lds, 0, part, S, bias, out, k, v, active, row_len, rope, nq, nkv, Tmax, 0.088f, (uint8_t*)nullptr, (uint8_t*)nullptr, 0);
        else hipLaunchKernelGGL((decode_attn_mfma_kernel<bf16_t, 128, 5>), dim3(slots, nkv), dim3(256), lds, 0, part, S, bias, out, k, v, active, row_len, rope, nq, nkv, Tmax, 0.088f);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    run(0);
    CK(hipDeviceSynchronize());
    {   // checksum of layer 0's output (the launch appends row `len`, which later launches of the same layer overwrite with the same values)
        std::vector<unsigned short> ho((size_t)slots * nq * D);
        CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
        unsigned long long h = 1469598103934665603ull; int nan = 0;
        for (unsigned short v : ho) { h = (h ^ v) * 1099511628211ull; nan += ((v & 0x7F80) == 0x7F80); }
        printf("output checksum %016llx (%d non-finite)\n", h, nan);
    }
    for (int i = 0; i < 32; ++i) run(i % layers);
    CK(hipDeviceSynchronize());
    printf("launch status: %s, lds %zu\n", hipGetErrorString(hipGetLastError()), lds);
    CK(hipEventRecord(e0));
    const int iters = 320;
    for (int i = 0; i < iters; ++i) run(i % layers);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("decode attention v%d: %.2f us per launch (back to back)\n", ver, ms * 1000.f / iters);
#ifdef SA_DA_TIMING
    unsigned long long st[16];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(sa_da_stamps), sizeof(st)));
    const char* names2[] = {"slot/len loads + K/V fetch issue", "qkv partial loads + reduce -> xrow", "stash + barrier", "RoPE + append + barrier",
                            "scores (MFMA) + barrier", "softmax + barrier", "PV + barrier", "normalise + store"};
    const char* names3[] = {"slot/len loads + K/V LDS-DMA issue", "qkv partial loads + reduce -> xrow", "vmcnt(0) + barrier", "RoPE + append + barrier",
                            "per-wave flash (scores, softmax, PV)", "combine records + barriers", "final combine + store", "-"};
    const char* names4[] = {"slot/len loads + K/V LDS-DMA issue", "slab / bias / rope loads issued, q pad rows zeroed", "wait for loads, reduce, RoPE, LDS writes",
                            "vmcnt(0) + barrier + cache append issue", "per-wave flash (scores, softmax, PV)", "records + barrier", "final combine + store", "-"};
    const char** names = ver >= 4 ? names4 : (ver == 3 ? names3 : names2);
    for (int i = 0; i < 8; ++i) printf("  phase %d %-52s %6.2f us\n", i, names[i], (double)(st[i + 1] - st[i]) / 100.0);
    printf("  total inside the kernel %.2f us\n", (double)(st[8] - st[0]) / 100.0);
#endif
    return 0;
}
