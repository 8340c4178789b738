// x / d for 32-bit unsigned x without a division instruction sequence (an integer division is ~40 VALU instructions on gfx950, a 64-bit one
// ~150 with branches). d a power of two -> shift; otherwise the 33-bit round-up reciprocal m = floor(2^(33 + k) / d) + 1, k = floor(log2 d),
// whose top bit is implicit: q = mulhi(x, m), x / d = (((x - q) >> 1) + q) >> k -- exact for every 32-bit x and every d in [1, 2^31].
// The reciprocal is made on the host (make_fastdiv) and travels in the kernel arguments. Host-compilable on purpose: tests/test_fastdiv_cpu.py
// builds this header with g++ and checks it against `/` (edge numerators for every divisor a launch can produce, random ones beyond).
#pragma once

#ifdef __HIPCC__
#define SA_FASTDIV_HD __host__ __device__ __forceinline__
#else
#define SA_FASTDIV_HD inline
#endif

namespace sa {

struct FastDiv { unsigned m = 0, s = 0; };

static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    if (d == 0) d = 1;
    unsigned k = 31 - (unsigned)__builtin_clz(d);
    if ((d & (d - 1)) == 0) { f.m = 0; f.s = k; return f; }
    f.m = (unsigned)((((unsigned __int128)1 << (33 + k)) / d + 1) & 0xffffffffu);
    f.s = k;
    return f;
}

SA_FASTDIV_HD unsigned fast_div(unsigned x, FastDiv f) {
    if (f.m == 0) return x >> f.s;
    const unsigned q = (unsigned)(((unsigned long long)x * f.m) >> 32);       // v_mul_hi_u32 / s_mul_hi_u32
    return (((x - q) >> 1) + q) >> f.s;
}

}  // namespace sa
