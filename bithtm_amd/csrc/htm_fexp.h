// The documented float32 exponential of the boost factor (regularizations.py:16).
// Twin of oracle/fexp.py: the same IEEE-754 double operations in the same order, one final
// rounding to float32.  This translation unit must be compiled with -ffp-contract=off.
#pragma once

__device__ __forceinline__ float htm_exp_f32(float x32) {
    const double LOG2E = 0x1.71547652b82fep+0;
    const double LN2_HI = 0x1.62e42fee00000p-1;
    const double LN2_LO = 0x1.a39ef35793c76p-33;
    const double T[15] = {
        0x1.0000000000000p+0,  0x1.0000000000000p+0,  0x1.0000000000000p-1,  0x1.5555555555555p-3,
        0x1.5555555555555p-5,  0x1.1111111111111p-7,  0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-13,
        0x1.a01a01a01a01ap-16, 0x1.71de3a556c734p-19, 0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26,
        0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33, 0x1.93974a8c07c9dp-37};
    double x = (double)x32;
    double n = rint(x * LOG2E);
    double r = (x - n * LN2_HI) - n * LN2_LO;
    double p = T[14];
#pragma unroll
    for (int j = 13; j >= 0; --j) p = p * r + T[j];
    return (float)ldexp(p, (int)n);
}
