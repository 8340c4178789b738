// Pillow's 8-bit LANCZOS resampler on the device (detection pages: Image.thumbnail + Image.resize of
// surya/detection/__init__.py:50-57; algorithm = Pillow src/libImaging/Resample.c, restated in surya_amd/common/pil_resample.py,
// which also builds the fixed-point coefficient tables on the host with the same libm). Byte / integer work, HBM-bound:
//   pass(out position o) = clip8((2^21 + sum_{t < n(o)} src[first(o) + t] * k[o][t]) >> 22), horizontal pass first into a uint8
//   intermediate, then the vertical pass over it -- bit-identical to Pillow by construction (int32 accumulation, arithmetic
//   shift, same tables).
// One thread per output pixel (3 channels); a row of output pixels reads a contiguous span of the source row.
#pragma once
#include "common.h"

namespace sa {
namespace rs {

struct PassArgs {
    const unsigned char* src; int sw, sh, spix;      // source [sh][sw][spix]
    unsigned char* dst; int dw, dh, dpix;            // destination [dh][dw][dpix]
    const int* bounds;                               // [out][2] = first source index, tap count
    const int* kk; int ksize;                        // [out][ksize] 22-bit fixed point
};

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= 22;                                        // arithmetic shift, as Pillow's clip8 lookup index
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
__device__ __forceinline__ void put_px(unsigned char* d, int dpix, int s0, int s1, int s2) {
    if (dpix == 4) *reinterpret_cast<uint32_t*>(d) = (uint32_t)clip8(s0) | ((uint32_t)clip8(s1) << 8) | ((uint32_t)clip8(s2) << 16);
    else { d[0] = clip8(s0); d[1] = clip8(s1); d[2] = clip8(s2); }
}

__global__ __launch_bounds__(256) void resample_h_kernel(PassArgs a) {        // dh == sh
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xx >= a.dw) return;
    const int x0 = a.bounds[2 * xx], n = a.bounds[2 * xx + 1];
    const int* k = a.kk + (long)xx * a.ksize;
    const unsigned char* p = a.src + ((long)y * a.sw + x0) * a.spix;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int t = 0; t < n; ++t, p += a.spix) {
        const int kv = k[t];
        s0 += (int)p[0] * kv; s1 += (int)p[1] * kv; s2 += (int)p[2] * kv;
    }
    put_px(a.dst + ((long)y * a.dw + xx) * a.dpix, a.dpix, s0, s1, s2);
}

__global__ __launch_bounds__(256) void resample_v_kernel(PassArgs a) {        // dw == sw
    const int x = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y;
    if (x >= a.dw) return;
    const int y0 = a.bounds[2 * yy], n = a.bounds[2 * yy + 1];
    const int* k = a.kk + (long)yy * a.ksize;                                  // block-uniform: scalar loads
    const unsigned char* p = a.src + ((long)y0 * a.sw + x) * a.spix;
    const long stride = (long)a.sw * a.spix;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int t = 0; t < n; ++t, p += stride) {
        const int kv = k[t];
        s0 += (int)p[0] * kv; s1 += (int)p[1] * kv; s2 += (int)p[2] * kv;
    }
    put_px(a.dst + ((long)yy * a.dw + x) * a.dpix, a.dpix, s0, s1, s2);
}

__global__ __launch_bounds__(256) void repack_kernel(const unsigned char* src, int spix, unsigned char* dst, int dpix, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char* p = src + i * spix;
    unsigned char* d = dst + i * dpix;
    d[0] = p[0]; d[1] = p[1]; d[2] = p[2];
    if (dpix == 4) d[3] = 0;
}

// One ImagingResample: horizontal pass (if the width changes) into `tmp` ([sh][dw][4]), vertical pass (if the height changes).
static inline int resample_run(const unsigned char* src, int sw, int sh, int spix, unsigned char* dst, int dw, int dh, int dpix,
                               const int* bx, const int* kx, int ksx, const int* by, const int* ky, int ksy, unsigned char* tmp,
                               hipStream_t s) {
    const bool need_h = dw != sw, need_v = dh != sh;
    if (!need_h && !need_v) {
        hipLaunchKernelGGL(repack_kernel, dim3((unsigned)cdivl((long)sw * sh, 256)), dim3(256), 0, s, src, spix, dst, dpix, (long)sw * sh);
        return (int)hipGetLastError();
    }
    const unsigned char* cur = src;
    int cw = sw, cpix = spix;
    if (need_h) {
        if (!bx || !kx || (need_v && !tmp)) return SA_ERR_ARG;
        PassArgs a{src, sw, sh, spix, need_v ? tmp : dst, dw, sh, need_v ? 4 : dpix, bx, kx, ksx};
        hipLaunchKernelGGL(resample_h_kernel, dim3(cdiv(dw, 256), sh), dim3(256), 0, s, a);
        cur = tmp; cw = dw; cpix = 4;
    }
    if (need_v) {
        if (!by || !ky) return SA_ERR_ARG;
        PassArgs a{cur, cw, sh, cpix, dst, dw, dh, dpix, by, ky, ksy};
        hipLaunchKernelGGL(resample_v_kernel, dim3(cdiv(dw, 256), dh), dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
}

}  // namespace rs
}  // namespace sa
