"""CPU: the division-free index arithmetic the round-5 kernels rely on, checked against `/`.

  * surya_amd/csrc/fastdiv.h (FastDiv, make_fastdiv, fast_div): host-compilable on purpose -- built here with g++ and run over every
    divisor a launch can produce (1 ... 4096, image extents, row counts up to 2^31) x edge numerators (multiples of d and their
    neighbours, 2^31 - 1, 2^31, 2^32 - 1) + pseudo-random ones. The persistent convolution's per-row state, the depthwise kernels' thread
    index and conv_gemm_kernel's per-chunk tap arithmetic all go through it; a wrong quotient there is a silently wrong gather address.
  * the 16-bit reciprocals of the persistent convolution's wave-uniform tap math (csrc/det_kernels.h launch_conv_on_gemm: cv_m1, cv_m2;
    csrc/gemm.h SP_REQX): (x * ceil(2^16 / d)) >> 16 == x // d on the ranges the launcher admits (K-tile index < 4096 with d <= 16,
    tap index < 64 with d <= 7), and NOT beyond them -- the launcher's range check is what makes the shortcut legal.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r'''
#include <cstdio>
#include <cstdint>
#include <vector>
#include "fastdiv.h"
int main() {
    std::vector<uint32_t> ds;
    for (uint32_t d = 1; d <= 4096; ++d) ds.push_back(d);
    const uint32_t extra[] = {28224u, 65536u, 65537u, 76800u, 84672u, 262144u, 1048576u, 1048577u, 3u << 20, 5u << 24, 0x7fffffffu, 0x80000000u,
                              1000003u, 16777259u, 268435459u, 2147483629u, 641u, 6700417u};
    for (uint32_t d : extra) ds.push_back(d);
    for (int k = 1; k < 31; ++k) { ds.push_back((1u << k) - 1); ds.push_back((1u << k) + 1); }
    unsigned long long bad = 0, n = 0;
    uint32_t seed = 12345u;
    for (uint32_t d : ds) {
        if (d == 0) continue;
        const sa::FastDiv f = sa::make_fastdiv(d);
        std::vector<uint32_t> xs = {0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu, 0xffffffffu - d, 0xffffffffu - d + 1};
        const uint32_t qmax = 0xffffffffu / d;
        for (uint32_t q : {qmax, qmax - 1, qmax / 2, qmax / 3}) { xs.push_back(q * d); if (q * d) xs.push_back(q * d - 1); if ((uint64_t)q * d + 1 <= 0xffffffffull) xs.push_back(q * d + 1); }
        for (int i = 0; i < 400; ++i) { seed = seed * 1664525u + 1013904223u; xs.push_back(seed); xs.push_back(seed >> (i % 24)); }
        for (uint32_t x : xs) { ++n; if (sa::fast_div(x, f) != x / d) { if (bad < 5) std::printf("MISMATCH x=%u d=%u got=%u want=%u\n", x, d, sa::fast_div(x, f), x / d); ++bad; } }
    }
    std::printf("checked %llu bad %llu\n", n, bad);
    return bad != 0;
}
'''


def test_fastdiv_header_against_integer_division(tmp_path):
    src = tmp_path / "fd.cpp"
    src.write_text(HARNESS)
    exe = tmp_path / "fd"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "surya_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-600:]
    assert "bad 0" in out.stdout and int(out.stdout.split("checked ")[1].split()[0]) > 3_000_000, out.stdout


def test_sixteen_bit_reciprocals_on_the_launchers_ranges():
    for d in range(1, 17):                       # K-tiles per tap = Cin / 64 <= 16 (Cin <= 1024), K-tile index < 4096
        m = (65536 + d - 1) // d
        assert all((x * m) >> 16 == x // d for x in range(4096)), d
    for d in range(1, 8):                        # filter width <= 7, tap index < 64 (the virtual K-tile's taps included)
        m = (65536 + d - 1) // d
        assert all((x * m) >> 16 == x // d for x in range(64)), d
    # ... and the ranges are needed: the shortcut fails just outside them
    m = (65536 + 3 - 1) // 3
    assert any((x * m) >> 16 != x // 3 for x in range(4096, 70000))


def test_launcher_guards_match_the_reciprocal_ranges():
    """launch_conv_on_gemm admits the persistent loop only where the 16-bit reciprocals are exact."""
    src = open(os.path.join(ROOT, "surya_amd", "csrc", "det_kernels.h")).read()
    assert "a.KH <= 7 && a.KW <= 7 && nkc < 4096" in src and "a.Cin <= 1024" in src
