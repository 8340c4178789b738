// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds lds[i] = i (16-bit); every 16-lane group supplies the
// addresses of a [4 rows][16 cols] block (lane i: row i >> 2, cols (i & 3) * 4 .. + 3, row pitch PITCH elements) and
// the program prints what each lane receives. Expectation (guide T10): lane i gets column i: out[j] = block[j][i].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 64;
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const short* p = lds + (l >> 4) * 1024 + ((l & 15) >> 2) * PITCH + (l & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int expect = (l >> 4) * 1024 + j * PITCH + (l & 15);
            printf(" %4d%s", h[l * 4 + j], h[l * 4 + j] == expect ? "" : "!");
            bad += h[l * 4 + j] != expect;
        }
        printf("\n");
    }
    printf("mismatches vs column-of-[4][16]-block hypothesis: %d\n", bad);
    return 0;
}
