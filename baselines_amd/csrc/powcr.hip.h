// pow(x, a) for x > 0 in double-double arithmetic, rounded to nearest once at the end (round 6).
// The reference computes prioritized-replay leaves as `priority ** alpha` on Python floats (deepq/replay_buffer.py:169-191), i.e. with
// glibc's pow, which misrounds about 1 input in 1500 (measured: 0.065 % of 20,000 priorities against a 60-digit decimal evaluation);
// ROCm's device pow differs from glibc's in 15.7 % of the inputs (always by one ulp; scripts/pow_check.py).  A correctly rounded device
// pow agrees with the reference wherever glibc itself rounds correctly -- 99.9 % -- and is the same on every host (glibc's x86-64 pow picks
// an FMA or non-FMA variant at run time, so "the reference's last bit" is not even one number).
//   log x = e ln2 + log c + 2 atanh(s), m in [sqrt(1/2), sqrt(2)), c = the nearest sixteenth to m (13 tabulated logs), s = (m - c) / (m + c):
//           odd series in double-double (|s| <= 0.0233, 11 terms);
//   exp p  = 2^k exp(r), r = p - k ln2 reduced once more by 2^-4 (|r| <= 0.022: 14 Taylor terms), squared back four times;
// every step carries ~100 bits, the final hi + lo -> double is the only rounding.  Plain C++ (host + device) so that the CPU test suite
// can check it against a decimal reference (tests/test_powcr.py builds it with g++).
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MRL_PCR_HD __host__ __device__ __forceinline__
#else
#define MRL_PCR_HD inline
#endif
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace mrl {

struct pcr_dd { double hi, lo; };

MRL_PCR_HD pcr_dd pcr_two_sum(double a, double b) {          // exact a + b
    const double s = a + b, bb = s - a;
    return pcr_dd{s, (a - (s - bb)) + (b - bb)};
}
MRL_PCR_HD pcr_dd pcr_fast_two_sum(double a, double b) {     // |a| >= |b|
    const double s = a + b;
    return pcr_dd{s, b - (s - a)};
}
MRL_PCR_HD pcr_dd pcr_two_prod(double a, double b) {         // exact a * b
    const double p = a * b;
    return pcr_dd{p, __builtin_fma(a, b, -p)};
}
MRL_PCR_HD pcr_dd pcr_add(pcr_dd a, pcr_dd b) {
    pcr_dd s = pcr_two_sum(a.hi, b.hi);
    const pcr_dd t = pcr_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = pcr_fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return pcr_fast_two_sum(s.hi, s.lo);
}
MRL_PCR_HD pcr_dd pcr_add_d(pcr_dd a, double b) {
    pcr_dd s = pcr_two_sum(a.hi, b);
    s.lo += a.lo;
    return pcr_fast_two_sum(s.hi, s.lo);
}
MRL_PCR_HD pcr_dd pcr_mul(pcr_dd a, pcr_dd b) {
    pcr_dd p = pcr_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return pcr_fast_two_sum(p.hi, p.lo);
}
MRL_PCR_HD pcr_dd pcr_mul_d(pcr_dd a, double b) {
    pcr_dd p = pcr_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return pcr_fast_two_sum(p.hi, p.lo);
}
MRL_PCR_HD pcr_dd pcr_div(pcr_dd a, pcr_dd b) {              // three quotient digits
    const double q1 = a.hi / b.hi;
    pcr_dd r = pcr_add(a, pcr_mul_d(b, -q1));
    const double q2 = r.hi / b.hi;
    r = pcr_add(r, pcr_mul_d(b, -q2));
    const double q3 = r.hi / b.hi;
    pcr_dd q = pcr_fast_two_sum(q1, q2);
    return pcr_add_d(q, q3);
}

// The coefficients 1 / (2k + 1) and 1 / n! below are double-doubles from a 60-digit decimal evaluation, written out as literals in fully
// unrolled Horner chains: a local coefficient ARRAY ends up in scratch memory on the device (528 bytes in the sampling kernel -- and a
// kernel that needs scratch pays for its set-up at launch: the replay row of bench.py dropped from 10.7 M to 6.1 M transitions/s).

// log(k / 16), k = 11 .. 23, as double-doubles (80-digit decimal evaluation)
MRL_PCR_HD pcr_dd pcr_log_sixteenth(int k) {
    switch (k) {
        case 11: return pcr_dd{-0x1.7fafa3bd8151cp-2, 0x1.219024acd3b77p-58};
        case 12: return pcr_dd{-0x1.269621134db92p-2, -0x1.e0efadd9db02bp-56};
        case 13: return pcr_dd{-0x1.a93ed3c8ad9e3p-3, -0x1.bcafa9de97203p-57};
        case 14: return pcr_dd{-0x1.1178e8227e47cp-3, 0x1.0e63a5f01c691p-58};
        case 15: return pcr_dd{-0x1.08598b59e3a07p-4, 0x1.dd7009902bf32p-58};
        case 17: return pcr_dd{0x1.f0a30c01162a6p-5, 0x1.85f325c5bbacdp-59};
        case 18: return pcr_dd{0x1.e27076e2af2e6p-4, -0x1.61578001e0162p-60};
        case 19: return pcr_dd{0x1.5ff3070a793d4p-3, -0x1.bc60efafc6f6ep-58};
        case 20: return pcr_dd{0x1.c8ff7c79a9a22p-3, -0x1.4f689f8434012p-57};
        case 21: return pcr_dd{0x1.1675cababa60ep-2, 0x1.ce63eab883717p-61};
        case 22: return pcr_dd{0x1.4618bc21c5ec2p-2, 0x1.f42decdeccf1dp-56};
        case 23: return pcr_dd{0x1.739d7f6bbd007p-2, -0x1.8c76ceb014b04p-56};
        default: return pcr_dd{0.0, 0.0};                       // k == 16
    }
}

// log(x) for finite x > 0 (normal or subnormal) as a double-double
MRL_PCR_HD pcr_dd pcr_log(double x) {
    int e = 0;
    if (x < 0x1p-1022) { x *= 0x1p+54; e = -54; }          // subnormal
    uint64_t bits;
    memcpy(&bits, &x, 8);
    e += (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &bits, 8);                                   // [1, 2)
    if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; e += 1; }     // -> [sqrt(1/2), sqrt(2))
    const int k = (int)(m * 16.0 + 0.5);                    // 11 .. 23: the nearest sixteenth c = k / 16 (exact)
    const double c = (double)k * 0.0625;
    const pcr_dd s = pcr_div(pcr_two_sum(m, -c), pcr_two_sum(m, c));   // |s| <= 0.0233
    const pcr_dd t = pcr_mul(s, s);
    pcr_dd acc = {0x1.8618618618618p-5, 0x1.8618618618618p-59};        // 1 / 21: t^11 / 23 < 2^-124
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.af286bca1af28p-5, 0x1.af286bca1af28p-59});
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61});
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.1111111111111p-4, 0x1.1111111111111p-60});
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58});
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.745d1745d1746p-4, -0x1.745d1745d1746p-59});
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58});
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.2492492492492p-3, 0x1.2492492492492p-57});
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.999999999999ap-3, -0x1.999999999999ap-57});
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.5555555555555p-2, 0x1.5555555555555p-56});
    acc = pcr_add(pcr_mul(acc, t), pcr_dd{0x1.0000000000000p+0, 0x0.0p+0});
    acc = pcr_mul(acc, s);
    acc.hi *= 2.0; acc.lo *= 2.0;                           // 2 atanh(s)
    const pcr_dd ln2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
    return pcr_add(pcr_add(pcr_mul_d(ln2, (double)e), pcr_log_sixteenth(k)), acc);
}

// exp(p) rounded to nearest double; p a double-double with |p| < 700
MRL_PCR_HD double pcr_exp(pcr_dd p) {
    const pcr_dd ln2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
    const double kd = nearbyint(p.hi * 0x1.71547652b82fep+0);
    pcr_dd r = pcr_add(p, pcr_mul_d(ln2, -kd));
    r.hi *= 0x1p-4; r.lo *= 0x1p-4;                         // |r| <= 0.0217
    pcr_dd acc = {0x1.ae7f3e733b81fp-41, 0x1.1d8656b0ee8cbp-97};
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.93974a8c07c9dp-37, 0x1.05d6f8a2efd1fp-92});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.ae64567f544e4p-26, -0x1.c062e06d1f209p-80});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.27e4fb7789f5cp-22, 0x1.cbbc05b4fa99ap-76});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.1111111111111p-7, 0x1.1111111111111p-63});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.5555555555555p-5, 0x1.5555555555555p-59});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.5555555555555p-3, 0x1.5555555555555p-57});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.0000000000000p-1, 0x0.0p+0});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.0000000000000p+0, 0x0.0p+0});
    acc = pcr_add(pcr_mul(acc, r), pcr_dd{0x1.0000000000000p+0, 0x0.0p+0});
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = pcr_mul(acc, acc);
    return ldexp(acc.hi + acc.lo, (int)kd);                 // hi + lo: the one rounding (exact scaling unless the result is subnormal)
}

// x ** a for finite x > 0; the cases the callers can produce otherwise follow pow(): x == 0 -> 0 (a > 0), a == 0 -> 1
MRL_PCR_HD double pow_cr(double x, double a) {
    if (a == 0.0 || x == 1.0) return 1.0;
    if (!(x > 0.0) || !(x < INFINITY) || !(a == a)) return pow(x, a);
    if (a == 1.0) return x;
    const pcr_dd p = pcr_mul_d(pcr_log(x), a);
    if (!(fabs(p.hi) < 700.0)) return pow(x, a);            // overflow / underflow range: the library's handling
    return pcr_exp(p);
}

}  // namespace mrl
