// device_math.h -- small __device__ helpers shared by the ROI kernels.
//
// Every function restates a piece of reference arithmetic whose exact rounding
// matters for parity (reference paths relative to /root/reference/src/nyx/).
// This translation unit is compiled with -ffp-contract=off: the reference is built
// with gcc -O2 and no -march (CMakeLists.txt:118), so it never fuses a*b+c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nyxhip {

// helpers/helpers.h:283-330 fast_log10(): float32 quadratic fit of log2 on the
// significand + integer exponent.  fast_log2f() is that function up to (and bit-for-bit
// including) its float `lg2`; the reference then returns lg2 * 0.30102999566.
__device__ __forceinline__ float fast_log2f(double _x)
{
    float x = (float)_x;
    const float a = -.6296735f;
    const float b = 1.466967f;
    unsigned int ui = __float_as_uint(x);
    int exp_ = (int)((ui & 0x7F800000u) >> 23);
    float signif, fexp;
    if (ui & 0x00400000u) {
        signif = __uint_as_float((ui & 0x007FFFFFu) | 0x3f000000u);
        fexp = (float)exp_ - 126.0f;
    } else {
        signif = __uint_as_float((ui & 0x007FFFFFu) | 0x3f800000u);
        fexp = (float)exp_ - 127.0f;
    }
    signif = signif - 1.0f;
    // fexp + a*signif*signif + b*signif, left to right, no FMA
    float t1 = __fmul_rn(__fmul_rn(a, signif), signif);
    float t2 = __fmul_rn(b, signif);
    return __fadd_rn(__fadd_rn(fexp, t1), t2);
}
__device__ __forceinline__ double fast_log10(double _x)
{
    return (double)fast_log2f(_x) * 0.30102999566;
}

// The GLCM code always uses the form  p * fast_log10(arg + EPSILON) / LOG10_2
// (EPSILON glcm.h:250, LOG10_2 = 0.30102999566 glcm.h:242), i.e. it scales lg2 by the
// constant and divides it out again.  (lg2*c)/c equals lg2 to within 1 ulp of double, so
// the product p * lg2 is used directly: one multiply instead of mul + mul + fp64 divide,
// at a relative deviation <= 2.3e-16 from the reference expression.
__device__ __forceinline__ double plogp(double p, double arg)
{
    return p * (double)fast_log2f(arg + 0.000000001);
}

// features/texture_feature.h:106-118 to_grayscale_radiomix
__device__ __forceinline__ uint32_t bin_radiomix(uint32_t x, uint32_t mn, uint32_t mx, int binCount)
{
    if (x) {
        double binW = (double)(mx - mn) / (double)binCount;
        uint32_t y = (uint32_t)((double)(x - mn) / binW + 1);
        if (y > (uint32_t)binCount)
            y = (uint32_t)binCount;
        return y;
    }
    return 0;
}

// features/texture_feature.h:138-197 matlab binning: slope = n / max, intercept 1,
// 0 -> 1, clip to [1, n].
__device__ __forceinline__ uint32_t bin_matlab(uint32_t x, double slope, int n_levels)
{
    if (x == 0)
        return 1;
    double scaled_real = floor(slope * (double)x + 1.0);
    uint32_t scaled = (uint32_t)scaled_real; // in [1, n+1]: x <= max by construction
    if (scaled > (uint32_t)n_levels)
        scaled = (uint32_t)n_levels;
    if (scaled < 1)
        scaled = 1;
    return scaled;
}

// features/texture_feature.h:77-98 bin_pixel: >0 matlab, <0 radiomics, 0 identity
__device__ __forceinline__ uint32_t bin_pixel(uint32_t x, uint32_t mn, uint32_t mx, int greybin_info)
{
    if (greybin_info < 0)
        return bin_radiomix(x, mn, mx, -greybin_info);
    if (greybin_info > 0)
        return bin_matlab(x, (double)greybin_info / ((double)mx - 0.), greybin_info);
    return x;
}

// helpers/helpers.h:337-345 to_grayscale (NaN -> 0 as the x86-64 reference binary does)
__device__ __forceinline__ uint32_t to_grayscale(uint32_t i, uint32_t min_i, uint32_t i_range, uint32_t n_levels)
{
    double pi = ((double)(i - min_i) / (double)i_range * (double)n_levels);
    if (pi != pi)
        return 0;
    return (uint32_t)pi;
}

// ---- wave64 / block reductions ------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v; // valid in lane 0
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

} // namespace nyxhip
