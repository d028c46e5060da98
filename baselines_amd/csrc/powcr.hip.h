// pow(x, a) for x > 0 in double-double arithmetic, rounded to nearest once at the end (round 6).
// The reference computes prioritized-replay leaves as `priority ** alpha` on Python floats (deepq/replay_buffer.py:169-191), i.e. with
// glibc's pow, which misrounds about 1 input in 1500 (measured: 0.065 % of 20,000 priorities against a 60-digit decimal evaluation);
// ROCm's device pow differs from glibc's in 15.7 % of the inputs (always by one ulp; scripts/pow_check.py).  A correctly rounded device
// pow agrees with the reference wherever glibc itself rounds correctly -- 99.9 % -- and is the same on every host (glibc's x86-64 pow picks
// an FMA or non-FMA variant at run time, so "the reference's last bit" is not even one number).
//   log x = e ln2 + 2 atanh(s), s = (m - 1) / (m + 1), m in [sqrt(1/2), sqrt(2)): odd series in double-double (|s| <= 0.172, 24 terms);
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

// 1 / (2k + 1) and 1 / n! as double-doubles (60-digit decimal evaluation)
#define MRL_PCR_ODD_RECIP                                                                                                     \
    {0x1.0000000000000p+0, 0x0.0p+0}, {0x1.5555555555555p-2, 0x1.5555555555555p-56}, {0x1.999999999999ap-3, -0x1.999999999999ap-57},      \
    {0x1.2492492492492p-3, 0x1.2492492492492p-57}, {0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58}, {0x1.745d1745d1746p-4, -0x1.745d1745d1746p-59}, \
    {0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58}, {0x1.1111111111111p-4, 0x1.1111111111111p-60}, {0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61}, \
    {0x1.af286bca1af28p-5, 0x1.af286bca1af28p-59}, {0x1.8618618618618p-5, 0x1.8618618618618p-59}, {0x1.642c8590b2164p-5, 0x1.642c8590b2164p-60},  \
    {0x1.47ae147ae147bp-5, -0x1.eb851eb851eb8p-61}, {0x1.2f684bda12f68p-5, 0x1.2f684bda12f68p-59}, {0x1.1a7b9611a7b96p-5, 0x1.1a7b9611a7b96p-61}, \
    {0x1.0842108421084p-5, 0x1.0842108421084p-60}, {0x1.f07c1f07c1f08p-6, -0x1.f07c1f07c1f08p-61}, {0x1.d41d41d41d41dp-6, 0x1.0750750750750p-60}, \
    {0x1.bacf914c1bad0p-6, -0x1.bacf914c1bad0p-60}, {0x1.a41a41a41a41ap-6, 0x1.0690690690690p-60}, {0x1.8f9c18f9c18fap-6, -0x1.f3831f3831f38p-61}, \
    {0x1.7d05f417d05f4p-6, 0x1.7d05f417d05f4p-62}, {0x1.6c16c16c16c17p-6, -0x1.f49f49f49f49fp-61}, {0x1.5c9882b931057p-6, 0x1.310572620ae4cp-61}
#define MRL_PCR_INV_FACT                                                                                                      \
    {0x1.0000000000000p+0, 0x0.0p+0}, {0x1.0000000000000p+0, 0x0.0p+0}, {0x1.0000000000000p-1, 0x0.0p+0},                                 \
    {0x1.5555555555555p-3, 0x1.5555555555555p-57}, {0x1.5555555555555p-5, 0x1.5555555555555p-59}, {0x1.1111111111111p-7, 0x1.1111111111111p-63},  \
    {0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65}, {0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73}, {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76}, \
    {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73}, {0x1.27e4fb7789f5cp-22, 0x1.cbbc05b4fa99ap-76}, {0x1.ae64567f544e4p-26, -0x1.c062e06d1f209p-80}, \
    {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83}, {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87}, {0x1.93974a8c07c9dp-37, 0x1.05d6f8a2efd1fp-92}, \
    {0x1.ae7f3e733b81fp-41, 0x1.1d8656b0ee8cbp-97}

// log(x) for finite x > 0 (normal or subnormal) as a double-double
MRL_PCR_HD pcr_dd pcr_log(double x) {
    const pcr_dd odd[24] = {MRL_PCR_ODD_RECIP};
    int e = 0;
    if (x < 0x1p-1022) { x *= 0x1p+54; e = -54; }          // subnormal
    uint64_t bits;
    memcpy(&bits, &x, 8);
    e += (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &bits, 8);                                   // [1, 2)
    if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; e += 1; }     // -> [sqrt(1/2), sqrt(2))
    const pcr_dd s = pcr_div(pcr_two_sum(m, -1.0), pcr_two_sum(m, 1.0));
    const pcr_dd t = pcr_mul(s, s);
    pcr_dd acc = odd[23];
#pragma unroll
    for (int k = 22; k >= 0; --k) acc = pcr_add(pcr_mul(acc, t), odd[k]);
    acc = pcr_mul(acc, s);
    acc.hi *= 2.0; acc.lo *= 2.0;                           // 2 atanh(s)
    const pcr_dd ln2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
    return pcr_add(pcr_mul_d(ln2, (double)e), acc);
}

// exp(p) rounded to nearest double; p a double-double with |p| < 700
MRL_PCR_HD double pcr_exp(pcr_dd p) {
    const pcr_dd ifa[16] = {MRL_PCR_INV_FACT};
    const pcr_dd ln2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
    const double kd = nearbyint(p.hi * 0x1.71547652b82fep+0);
    pcr_dd r = pcr_add(p, pcr_mul_d(ln2, -kd));
    r.hi *= 0x1p-4; r.lo *= 0x1p-4;                         // |r| <= 0.0217
    pcr_dd acc = ifa[15];
#pragma unroll
    for (int n = 14; n >= 0; --n) acc = pcr_add(pcr_mul(acc, r), ifa[n]);
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
