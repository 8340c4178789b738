#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Each block = one experiment. regs: per lane 8 dwords of A, 8 of B, scale a, scale b (all given per lane, raw), out 16 floats per lane.
template <int OPA, int OPB>
__global__ void run(const int* A, const int* B, const int* SA, const int* SB, float* D) {
    const int l = threadIdx.x, e = blockIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[(e * 64 + l) * 8 + i]; b[i] = B[(e * 64 + l) * 8 + i]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPA, SA[e * 64 + l], OPB, SB[e * 64 + l]);
    for (int r = 0; r < 16; ++r) D[(e * 64 + l) * 16 + r] = c[r];
}
static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}
struct Exp { std::vector<int> A, B, SA, SB; std::vector<float> D; int n;
    Exp(int n_) : A(n_ * 512), B(n_ * 512), SA(n_ * 64, 127), SB(n_ * 64, 127), D(n_ * 1024), n(n_) {}
    uint8_t* a8(int e, int l) { return reinterpret_cast<uint8_t*>(&A[(e * 64 + l) * 8]); }
    uint8_t* b8(int e, int l) { return reinterpret_cast<uint8_t*>(&B[(e * 64 + l) * 8]); }
    template <int OPA, int OPB> void go() {
        int *dA, *dB, *dSA, *dSB; float* dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dSA, SA.size() * 4); hipMalloc(&dSB, SB.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dSA, SA.data(), SA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), SB.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((run<OPA, OPB>), dim3(n), dim3(64), 0, 0, dA, dB, dSA, dSB, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        hipFree(dA); hipFree(dB); hipFree(dSA); hipFree(dSB); hipFree(dD);
    }
    // D as matrix [i][j]: standard 32x32 map col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
    float d(int e, int i, int j) const {
        for (int h = 0; h < 2; ++h) for (int r = 0; r < 16; ++r) if ((r & 3) + 8 * (r >> 2) + 4 * h == i) return D[(e * 64 + h * 32 + j) * 16 + r];
        return NAN;
    }
};
int main() {
    srand(5);
    {   // 1. data layout, scales = 1.0
        Exp x(1);
        auto rnd8 = [] { uint8_t v; do { v = rand() & 255; } while ((v & 0x7f) == 0x7f); return v; };
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { x.a8(0, l)[j] = rnd8(); x.b8(0, l)[j] = rnd8(); }
        x.go<0, 0>();
        double w = 0, wt = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int k = 0; k < 64; ++k) ref += (double)e4m3(x.a8(0, (k / 32) * 32 + i)[k % 32]) * e4m3(x.b8(0, (k / 32) * 32 + j)[k % 32]);
            w = fmax(w, fabs(x.d(0, i, j) - ref)); wt = fmax(wt, fabs(x.d(0, j, i) - ref));
        }
        printf("1. unscaled data layout: max abs err H1 %.4g  (transposed %.4g)  |D| sample %g\n", w, wt, x.d(0, 3, 5));
    }
    {   // 2. which lane's scale_a governs what: data all 1.0 (0x38), lane L has scale 2^4 in byte 0 (others 1.0)
        Exp x(64);
        for (int e = 0; e < 64; ++e) { for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { x.a8(e, l)[j] = 0x38; x.b8(e, l)[j] = 0x38; } x.SA[e * 64 + e] = 131; }
        x.go<0, 0>();
        printf("2. scale_a one-hot lane L (2^4), byte 0, opsel 0: rows whose D changed from 64 and (D-64)/15:\n");
        for (int e = 0; e < 64; ++e) { printf("   L=%2d:", e); for (int i = 0; i < 32; ++i) { float v = x.d(e, i, 0); if (v != 64.f) printf(" row %d +%g", i, (v - 64) / 15); } 
            bool colvar = false; for (int i = 0; i < 32; ++i) for (int j = 1; j < 32; ++j) if (x.d(e, i, j) != x.d(e, i, 0)) colvar = true; printf(colvar ? "  (varies with column!)\n" : "\n"); }
        Exp y(64);
        for (int e = 0; e < 64; ++e) { for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { y.a8(e, l)[j] = 0x38; y.b8(e, l)[j] = 0x38; } y.SB[e * 64 + e] = 131; }
        y.go<0, 0>();
        printf("2b. scale_b one-hot lane L: columns changed:\n");
        for (int e = 0; e < 64; e += 9) { printf("   L=%2d:", e); for (int j = 0; j < 32; ++j) { float v = y.d(e, 0, j); if (v != 64.f) printf(" col %d +%g", j, (v - 64) / 15); } printf("\n"); }
    }
    {   // 3. which BYTE of the scale register is used for opsel 0..3: all lanes scale word = bytes {2^1, 2^2, 2^3, 2^4} (128,129,130,131)
        Exp x(1);
        for (int l = 0; l < 64; ++l) { for (int j = 0; j < 32; ++j) { x.a8(0, l)[j] = 0x38; x.b8(0, l)[j] = 0x38; } x.SA[l] = 128 | (129 << 8) | (130 << 16) | (131 << 24); }
        x.go<0, 0>(); printf("3. opsel_a 0: D = 64 * %g\n", x.d(0, 0, 0) / 64);
        x.go<1, 0>(); printf("   opsel_a 1: D = 64 * %g\n", x.d(0, 0, 0) / 64);
        x.go<2, 0>(); printf("   opsel_a 2: D = 64 * %g\n", x.d(0, 0, 0) / 64);
        x.go<3, 0>(); printf("   opsel_a 3: D = 64 * %g\n", x.d(0, 0, 0) / 64);
    }
    {   // 4. k mapping inside a lane vs scale: A one-hot byte j of lane 0 (row 0, half 0) = 1.0, B all ones; scale_a lane 0 = 2^4 -> D[0][*] = 16 if that byte is governed by lane 0's scale
        Exp x(64);
        for (int e = 0; e < 64; ++e) { const int l = (e >> 5) * 32, j = e & 31; for (int q = 0; q < 64; ++q) for (int t = 0; t < 32; ++t) x.b8(e, q)[t] = 0x38; x.a8(e, l)[j] = 0x38; x.SA[e * 64 + 0] = 131; }
        x.go<0, 0>();
        printf("4. A one-hot (lane 0 or 32, byte j), scale_a lane0 = 16: D[0][0] per j:"); for (int e = 0; e < 64; ++e) printf(" %g", x.d(e, 0, 0)); printf("\n");
    }
    return 0;
}
