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

// ---- fast fp64 division for tolerance-class features ------------------------------------------
// a / b through v_rcp_f64 + two Newton steps + one residual correction (8 instructions instead of the
// ~35 of the IEEE expansion); result within 1 ulp of a / b.  Callers guarantee a finite, normal, non-zero
// b (pixel counts, 1 + k, sum_p ...).  Sites that feed bit-exact columns, or whose inf / NaN behaviour the
// reference relies on, keep the `/` operator.
__device__ __forceinline__ double fdiv(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

// 1 / b and 1 / sqrt(x) for tolerance-class features: hardware estimate + two Newton steps (within 1-2 ulp).  Callers
// guarantee a finite, normal, positive argument.
__device__ __forceinline__ double frcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
}
__device__ __forceinline__ double frsq(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-(x * y), y, 1.0);          // 1 - x y^2
    y = __builtin_fma(y * 0.5, e, y);
    e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * 0.5, e, y);
}

// ---- wave64 reductions on the DPP path -------------------------------------------------
// Cross-lane traffic goes through DPP (VALU data-parallel primitives: row_shr within a row
// of 16 lanes, row_bcast:15 / row_bcast:31 across rows on GFX9-family CDNA) instead of
// ds_bpermute, which would occupy the LDS pipe that the histograms need.
// After the six steps lane 63 holds the wave total; v_readlane broadcasts it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov0(uint32_t v)
{   // lanes without a source (or masked rows) read 0.  With every row enabled the zero comes from bound_ctrl and the
    // destination needs no initialisation (one v_mov less per move); masked rows keep `old`, so those steps pass an explicit 0.
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov0(double v)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    uint32_t lo = dpp_mov0<CTRL, ROW_MASK>((uint32_t)u), hi = dpp_mov0<CTRL, ROW_MASK>((uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_mov0(unsigned long long u)
{
    uint32_t lo = dpp_mov0<CTRL, ROW_MASK>((uint32_t)u), hi = dpp_mov0<CTRL, ROW_MASK>((uint32_t)(u >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double readlane63(double v)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ unsigned long long readlane63(unsigned long long u)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t readlane63(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

// DPP control words (GFX9 ISA): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
// Sum over the 64 lanes; the total is returned in EVERY lane (fixed, deterministic order).
template <typename T>
__device__ __forceinline__ T wave_sum_t(T v)
{
    v += dpp_mov0<0x111, 0xF>(v);
    v += dpp_mov0<0x112, 0xF>(v);
    v += dpp_mov0<0x114, 0xF>(v);
    v += dpp_mov0<0x118, 0xF>(v);   // lane 15 of each row = row total
    v += dpp_mov0<0x142, 0xA>(v);   // rows 1,3 += previous row's lane 15
    v += dpp_mov0<0x143, 0xC>(v);   // rows 2,3 += lane 31
    return readlane63(v);
}
// inclusive prefix sum over the 64 lanes (the same six DPP steps, without the broadcast)
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v)
{
    v += dpp_mov0<0x111, 0xF>(v);
    v += dpp_mov0<0x112, 0xF>(v);
    v += dpp_mov0<0x114, 0xF>(v);
    v += dpp_mov0<0x118, 0xF>(v);
    v += dpp_mov0<0x142, 0xA>(v);
    v += dpp_mov0<0x143, 0xC>(v);
    return v;
}
// inclusive prefix MINIMUM over the 64 lanes (lanes without a source in a step keep their own value)
// (lanes without a source read the identity of the minimum: the compiler then folds the move into v_min_u32_dpp -- one
//  instruction per step instead of copy + move + minimum)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_self(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ uint32_t wave_scan_min_u32(uint32_t v)
{
    v = min(v, dpp_self<0x111, 0xF>(v));
    v = min(v, dpp_self<0x112, 0xF>(v));
    v = min(v, dpp_self<0x114, 0xF>(v));
    v = min(v, dpp_self<0x118, 0xF>(v));
    v = min(v, dpp_self<0x142, 0xA>(v));
    v = min(v, dpp_self<0x143, 0xC>(v));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_t<double>(v); }
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) { return wave_sum_t<unsigned long long>(v); }

// max over the 64 lanes (values >= 0 or compared against the 0 fill: callers pass
// non-negative data or accept 0 as the floor), result in every lane
__device__ __forceinline__ double wave_max_nonneg(double v)
{
    double o;
    o = dpp_mov0<0x111, 0xF>(v); v = o > v ? o : v;
    o = dpp_mov0<0x112, 0xF>(v); v = o > v ? o : v;
    o = dpp_mov0<0x114, 0xF>(v); v = o > v ? o : v;
    o = dpp_mov0<0x118, 0xF>(v); v = o > v ? o : v;
    o = dpp_mov0<0x142, 0xA>(v); v = o > v ? o : v;
    o = dpp_mov0<0x143, 0xC>(v); v = o > v ? o : v;
    return readlane63(v);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    uint32_t o;
    o = dpp_mov0<0x111, 0xF>(v); v = o > v ? o : v;
    o = dpp_mov0<0x112, 0xF>(v); v = o > v ? o : v;
    o = dpp_mov0<0x114, 0xF>(v); v = o > v ? o : v;
    o = dpp_mov0<0x118, 0xF>(v); v = o > v ? o : v;
    o = dpp_mov0<0x142, 0xA>(v); v = o > v ? o : v;
    o = dpp_mov0<0x143, 0xC>(v); v = o > v ? o : v;
    return readlane63(v);
}
// ---- reductions inside one DPP row (16 lanes): butterfly over quad_perm / half-mirror / mirror -----
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140.  After the four
// steps every lane of the row holds the row total; the four rows of a wave reduce independently, which is
// how one wave serves four GLCM angles at once (roi_features.hip).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_perm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_perm(double v)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    uint32_t lo = dpp_perm<CTRL>((uint32_t)u), hi = dpp_perm<CTRL>((uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double row16_sum(double v)
{
    v += dpp_perm<0xB1>(v);
    v += dpp_perm<0x4E>(v);
    v += dpp_perm<0x141>(v);
    v += dpp_perm<0x140>(v);
    return v;
}
__device__ __forceinline__ uint32_t row16_sum(uint32_t v)
{
    v += dpp_perm<0xB1>(v);
    v += dpp_perm<0x4E>(v);
    v += dpp_perm<0x141>(v);
    v += dpp_perm<0x140>(v);
    return v;
}
// ---- one pass over a ROI's pixel cloud ---------------------------------------------------------------------------
// Calls f(i, intensity, x, y) for this thread's pixels i = tid, tid + BLK, ...  Four pixels per trip, read through buffer
// descriptors sized to the ROI (the bounds check belongs to the load: no compare / branch / 64-bit address per load) and
// all twelve loads of a trip issued before the first use -- one HBM round trip per 4 * BLK pixels of the workgroup instead
// of one per pixel and lane.
template <int BLK, typename F>
__device__ __forceinline__ void for_each_cloud_pixel(const uint32_t* inten, const uint16_t* x, const uint16_t* y, uint32_t n, int tid, F&& f)
{
    constexpr int kU = 4;
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)inten, 0, (int)(n * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(n * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)(n * 2u), 0x00020000);
    for (uint32_t base = 0; base < n; base += kU * BLK) {
        uint32_t v[kU], px[kU], py[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const uint32_t i = base + (uint32_t)(u * BLK) + (uint32_t)tid;
            v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_v, (int)(i * 4u), 0, 0);
            px[u] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_x, (int)(i * 2u), 0, 0);
            py[u] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_y, (int)(i * 2u), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const uint32_t i = base + (uint32_t)(u * BLK) + (uint32_t)tid;
            if (i < n)
                f(i, v[u], px[u], py[u]);
        }
    }
}

// 24-bit multiplies (both factors below 2^24; the low 32 bits of the product): full rate, where v_mul_lo_u32 -- what the
// compiler emits for a plain 32-bit product, and for __umul24 as well -- issues at a quarter of it
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---- (row, column) of a linear index that advances by a fixed stride -----------------------------------------------
// p = first, first + stride, ... over a plane of width w: one integer division per thread and loop instead of one per pixel
// (a u32 division is ~25 instructions, two of them quarter-rate multiplies).
struct RowCol {
    uint32_t row, col, step_r, step_c, w;
    __device__ __forceinline__ RowCol(uint32_t first, uint32_t stride, uint32_t width) : w(width)
    {
        row = first / width; col = first - row * width;
        step_r = stride / width; step_c = stride - step_r * width;
    }
    // the same for small operands (first, stride < 2^16, width <= 64): the two quotients through a float reciprocal --
    // (x + 1/2) / d is never closer than 1 / (2 d) to an integer, far beyond the rounding of the product
    struct small_t {};
    __device__ __forceinline__ RowCol(uint32_t first, uint32_t stride, uint32_t width, small_t) : w(width)
    {
        const float inv = __builtin_amdgcn_rcpf((float)width);
        row = (uint32_t)(((float)first + 0.5f) * inv); col = first - mul24(row, width);
        step_r = (uint32_t)(((float)stride + 0.5f) * inv); step_c = stride - mul24(step_r, width);
    }
    __device__ __forceinline__ void advance()
    {
        col += step_c; row += step_r;
        if (col >= w) { col -= w; row++; }
    }
};

// ---- transposed wave sum: 64 per-lane slots, lane L ends up with the wave total of slot L ------------------------
// Each step pairs the lanes and halves the slot list (one partner keeps the lower half, the other the upper half, each adds
// what the partner held of its half): 32 + 16 + 8 + 4 + 2 + 1 = 63 exchanges instead of 64 six-step butterflies.  The two
// cross-row steps are gfx950's v_permlane32_swap / v_permlane16_swap (swap the odd 32- / 16-lane rows of the first operand
// with the even rows of the second: no select needed), the four in-row steps are DPP moves.
__device__ __forceinline__ void permlane32_swap(double& a, double& b)
{
    unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)ua, (uint32_t)ub, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(ua >> 32), (uint32_t)(ub >> 32), false, false);
    a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
    b = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
__device__ __forceinline__ void permlane16_swap(double& a, double& b)
{
    unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)ua, (uint32_t)ub, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(ua >> 32), (uint32_t)(ub >> 32), false, false);
    a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
    b = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
template <int CTRL, int HALF>
__device__ __forceinline__ void transpose_sum_step(double* v, bool upper)
{
#pragma unroll
    for (int k = 0; k < HALF; k++) {
        const double keep = upper ? v[k + HALF] : v[k];
        const double send = upper ? v[k] : v[k + HALF];
        v[k] = keep + dpp_perm<CTRL>(send);
    }
}
// 16 slots: lane L returns the wave total of slot (L >> 2) & 15.
__device__ __forceinline__ double wave_transpose_sum16(double (&v)[16], int lane)
{
#pragma unroll
    for (int k = 0; k < 8; k++) { permlane32_swap(v[k], v[k + 8]); v[k] += v[k + 8]; }
#pragma unroll
    for (int k = 0; k < 4; k++) { permlane16_swap(v[k], v[k + 4]); v[k] += v[k + 4]; }
    transpose_sum_step<0x140, 2>(v, (lane & 8) != 0);
    transpose_sum_step<0x141, 1>(v, (lane & 4) != 0);
    double t = v[0];
    t += dpp_perm<0x4E>(t);
    t += dpp_perm<0xB1>(t);
    return t;
}
// 8 slots: lane L returns the wave total of slot (L >> 3) & 7 -- 3 halving exchanges, then a 3-step butterfly over the 8
// lanes that share a slot (34 instructions instead of 8 six-step butterflies).
__device__ __forceinline__ double wave_transpose_sum8(double (&v)[8], int lane)
{
#pragma unroll
    for (int k = 0; k < 4; k++) { permlane32_swap(v[k], v[k + 4]); v[k] += v[k + 4]; }
#pragma unroll
    for (int k = 0; k < 2; k++) { permlane16_swap(v[k], v[k + 2]); v[k] += v[k + 2]; }
    transpose_sum_step<0x140, 1>(v, (lane & 8) != 0);
    double t = v[0];
    t += dpp_perm<0x141>(t);
    t += dpp_perm<0x4E>(t);
    t += dpp_perm<0xB1>(t);
    return t;
}
// 8 slots of 32-bit integers: lane L returns the wave total of slot (L >> 3) & 7.
__device__ __forceinline__ uint32_t wave_transpose_sum8_u32(uint32_t (&v)[8], int lane)
{
#pragma unroll
    for (int k = 0; k < 4; k++) { auto r = __builtin_amdgcn_permlane32_swap(v[k], v[k + 4], false, false); v[k] = r[0] + r[1]; }
#pragma unroll
    for (int k = 0; k < 2; k++) { auto r = __builtin_amdgcn_permlane16_swap(v[k], v[k + 2], false, false); v[k] = r[0] + r[1]; }
    const bool upper = (lane & 8) != 0;
    const uint32_t keep = upper ? v[1] : v[0], send = upper ? v[0] : v[1];
    uint32_t t = keep + dpp_perm<0x140>(send);
    t += dpp_perm<0x141>(t);
    t += dpp_perm<0x4E>(t);
    t += dpp_perm<0xB1>(t);
    return t;
}
// 4 slots: lane L returns the wave total of slot (L >> 4) & 3.
__device__ __forceinline__ double wave_transpose_sum4(double (&v)[4])
{
#pragma unroll
    for (int k = 0; k < 2; k++) { permlane32_swap(v[k], v[k + 2]); v[k] += v[k + 2]; }
    permlane16_swap(v[0], v[1]);
    double t = v[0] + v[1];
    t += dpp_perm<0x140>(t);
    t += dpp_perm<0x141>(t);
    t += dpp_perm<0x4E>(t);
    t += dpp_perm<0xB1>(t);
    return t;
}
__device__ __forceinline__ double wave_transpose_sum64(double (&v)[64], int lane)
{
#pragma unroll
    for (int k = 0; k < 32; k++) { permlane32_swap(v[k], v[k + 32]); v[k] += v[k + 32]; }   // lanes >= 32 now hold slots 32..63
#pragma unroll
    for (int k = 0; k < 16; k++) { permlane16_swap(v[k], v[k + 16]); v[k] += v[k + 16]; }   // odd rows: + 16
    transpose_sum_step<0x140, 8>(v, (lane & 8) != 0);     // row_mirror: lane i <-> 15 - i
    transpose_sum_step<0x141, 4>(v, (lane & 4) != 0);     // row_half_mirror: i <-> 7 - i
    transpose_sum_step<0x4E, 2>(v, (lane & 2) != 0);      // quad_perm [2,3,0,1]
    transpose_sum_step<0xB1, 1>(v, (lane & 1) != 0);      // quad_perm [1,0,3,2]
    return v[0];
}
__device__ __forceinline__ double row16_max(double v)
{
    double o;
    o = dpp_perm<0xB1>(v); v = o > v ? o : v;
    o = dpp_perm<0x4E>(v); v = o > v ? o : v;
    o = dpp_perm<0x141>(v); v = o > v ? o : v;
    o = dpp_perm<0x140>(v); v = o > v ? o : v;
    return v;
}

// neighbour lanes through DPP wave shifts (GFX9: wave_shl:1 = 0x130, wave_shr:1 = 0x138);
// the lane without a source reads `fill`
// (a literal zero fill comes from bound_ctrl: the destination then needs no initialising move)
__device__ __forceinline__ uint32_t lane_plus1(uint32_t v, uint32_t fill)   // value of lane+1
{
    if (__builtin_constant_p(fill) && fill == 0)
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t lane_minus1(uint32_t v, uint32_t fill)  // value of lane-1
{
    if (__builtin_constant_p(fill) && fill == 0)
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false);
}


// 24-bit multiply that stays one: hipcc turns __umul24 of provably small operands into a plain 32-bit multiply, and
// v_mul_lo_u32 issues at a quarter of the rate.  `b` is wave-uniform (an SGPR operand).
__device__ __forceinline__ uint32_t mul_u24_su(uint32_t a, uint32_t b_uniform)
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(b_uniform), "v"(a));
    return r;
}

__device__ __forceinline__ int mul_i24(int a, int b)      // signed 24-bit factors, full rate
{
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// clamp of a per-lane value to wave-uniform bounds lo <= hi: the median of the three, one instruction
__device__ __forceinline__ uint32_t med3_u32_ss(uint32_t x, uint32_t lo_uniform, uint32_t hi_uniform)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo_uniform), "s"(hi_uniform));   // (one scalar operand per VOP3 on gfx9)
    return r;
}

// the same with zero fill through bound_ctrl (no initialisation of the destination)
// Word of a 16-bit counting table that holds entry ci (two entries per word): base + ((ci & ~1) << 1).  The mask is opaque to
// the compiler so that the shift and the base add stay one v_lshl_add_u32 (it otherwise canonicalises to shift, mask, add).
__device__ __forceinline__ uint32_t* cnt16_word(uint32_t* tab, uint32_t ci)
{
    uint32_t t;
    asm("v_and_b32 %0, -2, %1" : "=v"(t) : "v"(ci));
    return (uint32_t*)((char*)tab + (t << 1));
}

// rotations of the wave by one lane (DPP wave_rol:1 / wave_ror:1): every lane has a source
__device__ __forceinline__ uint32_t wave_rol1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x134, 0xF, 0xF, true); }   // value of lane + 1; lane 63 reads lane 0
__device__ __forceinline__ uint32_t wave_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C, 0xF, 0xF, true); }   // value of lane - 1; lane 0 reads lane 63
__device__ __forceinline__ uint32_t lane_plus1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t lane_minus1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true); }

// ---- synchronisation that also works when the "LDS" of a workgroup lives in the global workspace ----------
// GS = false: plain workgroup barrier / wave-level compiler fence (LDS instructions of a wave run in issue order).
// GS = true (workspace launches): the scratch is global memory, exchanged between the waves of ONE workgroup -- one CU, one
// vector L1 (write-through), one XCD's L2.  An exchange point waits for the wave's outstanding memory operations, passes the
// barrier, and invalidates the CU's L1 (acquire at agent scope, `buffer_inv sc1`: atomics are performed in L2, and a plain load
// must not meet a line cached before one).  Nothing is RELEASED at agent scope: that is `buffer_wbl2 sc1`, the write-back of the
// XCD's dirty L2 lines, which no other CU waits for here.  Rounds 1-3 did (seq_cst agent fences on both sides of every
// exchange): with hundreds of workspace workgroups in flight, each passing dozens of exchange points, the write-backs were a
// third of the workspace kernels' time (the 400 largest ROIs of the mixed batch with the config-4 families: 8.1 -> 5.1 ms).
// Builds for A/B runs: NYX_GS_FULL_FENCE (the old form), NYX_GS_WG_ONLY (workgroup scope only, no L1 invalidate: the AMDGPU
// memory model's rule for a workgroup on one CU; measured equal to the default within 1 %, suite and fuzzers green).
template <bool GS>
__device__ __forceinline__ void blk_sync()
{
#ifdef NYX_GS_FULL_FENCE
    if (GS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    __syncthreads();
    if (GS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
#elif defined(NYX_GS_WG_ONLY)
    __syncthreads();                                   // workgroup-scope release + acquire: s_waitcnt vmcnt(0) lgkmcnt(0), s_barrier
#else
    __syncthreads();
    if (GS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
template <bool GS>
__device__ __forceinline__ void wav_sync()
{
    if (GS) {
#ifdef NYX_GS_FULL_FENCE
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
#elif defined(NYX_GS_WG_ONLY)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // (the stores have reached L2 before the wave goes on: s_waitcnt vmcnt(0))
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

} // namespace nyxhip
