// roi_features.hip -- the fused per-ROI feature kernel for gfx950 (MI355X).
//
// One 256-thread workgroup (4 wave64) per ROI.  The ROI's pixel cloud is read from
// HBM exactly once (coalesced SoA streams x:u16, y:u16, inten:u32) and everything
// else happens in LDS:
//
//   * intensities are staged into an LDS array; order statistics come from one of two
//     LDS-resident engines chosen per ROI: a counting table over [min, max] (LDS
//     atomics + wave-scan prefix sums) when the intensity range fits, else an in-place
//     bitonic sort.  Exact median, exact mode, the 100-bin percentile histogram and the
//     n-bin histogram are all read off that engine by binary search; moments and the
//     robust / median absolute deviations are further passes over the resident copy
//       -> replaces PixelIntensityFeatures::calculate
//          (/root/reference/src/nyx/features/intensity.cpp:57-192), TrivialHistogram
//          (features/histogram.h:27-309) and Moments4 (features/moments.h:48-109);
//   * the binned bounding-box plane is scattered into LDS while loading, the
//     co-occurrence matrices of all angles are accumulated with LDS atomics and each
//     wave then derives the 30 Haralick features of one angle
//       -> replaces GLCMFeature::calculate / calculateCoocMatAtAngle / f_*
//          (features/glcm.cpp:16-100, 343-485, 487-1202).
//
// HBM bound by construction: algorithmic traffic is 8 B per ROI pixel in and
// 8 B x n_cols out; there are no re-reads.  No MFMA: nothing here is a contraction.
//
// Built with -ffp-contract=off (see device_math.h).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include "device_math.h"
#include "roi_kernel.h"
#include "launch_util.h"
#include "glcm_rows.h"
#include "glcm_w64.h"
#include "sort_lds.h"
#include "../../include/nyxhip.h"

namespace nyxhip {


// slots of the block-wide scalar array s_stat
enum { S_MEAN = 0, S_P10, S_P90, S_MEDIAN, S_MODE, S_NG };

// Lanes of one wave exchange data through LDS without a workgroup barrier: LDS
// instructions of a wave execute in issue order, so only the compiler has to be
// kept from reordering / caching across the exchange.
// (wav_sync<GS>() in device_math.h)

// Sum N doubles across the workgroup in a fixed order (deterministic): wave shuffle
// tree, then the four wave partials in wave order.  All threads get the totals.
// totals of slot k of the cross-wave exchange area (eight doubles per wave), in the two fixed orders the body uses
template <int NW> __device__ __forceinline__ double xw_chain(const double* s_red, int k)
{
    return NW == 1 ? s_red[k] : ((s_red[k] + s_red[8 + k]) + s_red[16 + k]) + s_red[24 + k];
}
template <int NW> __device__ __forceinline__ double xw_pairs(const double* s_red, int k)
{
    return NW == 1 ? s_red[k] : (s_red[k] + s_red[8 + k]) + (s_red[16 + k] + s_red[24 + k]);
}

template <int N, bool GS, int NW = 4>
__device__ __forceinline__ void block_sum(double (&v)[N], double* s_red, int tid)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (N > 4) {                                     // transposed wave sums: the lane group of slot k ends up with its total
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = k < N ? v[k] : 0.0;
        const double tot = wave_transpose_sum8(t, lane);
        if ((lane & 7) == 0 && (lane >> 3) < N)
            s_red[wave * 8 + (lane >> 3)] = tot;
    } else if (N > 2) {
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) t[k] = k < N ? v[k] : 0.0;
        const double tot = wave_transpose_sum4(t);
        if ((lane & 15) == 0 && (lane >> 4) < N)
            s_red[wave * 8 + (lane >> 4)] = tot;
    } else {
#pragma unroll
        for (int k = 0; k < N; k++)
            v[k] = wave_sum(v[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < N; k++)
                s_red[wave * 8 + k] = v[k];
        }
    }
    grp_sync<GS, NW>();
#pragma unroll
    for (int k = 0; k < N; k++) {
        double t = s_red[k];
#pragma unroll
        for (int w = 1; w < NW; w++)
            t += s_red[w * 8 + k];
        v[k] = t;
    }
    grp_sync<GS, NW>();
}

// In-place ascending bitonic sort of s[0..P), P a power of two.
template <bool GS, int NW = 4>
__device__ __forceinline__ void bitonic_sort(uint32_t* s, uint32_t P, int tid)
{
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (P >> 1); t += NW * 64) {
                uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                uint32_t l = i | j;
                uint32_t a = s[i], b = s[l];
                bool up = (i & k) == 0;
                if ((a > b) == up) {
                    s[i] = b;
                    s[l] = a;
                }
            }
            grp_sync<GS, NW>();
        }
    }
}

// ---- GLCM features of small matrices, one wave per ROI (glcm_features_kernel) ---------------------------------------------
// The kernel is bound by vector-instruction issue (every VALU operation occupies its SIMD for four cycles, whatever the
// type), so this routine is organised around the instruction count: level values are I[i] = i + 1 (matlab / IBSI), hence
//   * everything that only needs the marginal distributions is a sum over <= 16 lanes, one term each:  the means, CONTRAST,
//     DIS, VARIANCE, the correlation variances and JVAR from the row / column counts, the cluster sums and the sum features
//     from p_{x+y}, the difference features from p_{x-y} (integer numerators stay exact);
//   * only ASM, ACOR, ENTROPY, JMAX, the covariance term and the two HXY sums visit the cells;
//   * the partial sums of a phase are reduced together by a transposed exchange inside the 16-lane row (n values cost about
//     n + 3 exchanges instead of 4 n), land in LDS and are finished by one lane per angle.
// Same quantities as glcm_features_rows (features/glcm.cpp:487-1202), regrouped; deviations stay at the 1e-15 level.
__device__ __forceinline__ double row16_transpose_sum8(double (&v)[8], int l)
{   // lane l of the row returns the row total of slot (l >> 1) & 7
    transpose_sum_step<0x140, 4>(v, (l & 8) != 0);
    transpose_sum_step<0x141, 2>(v, (l & 4) != 0);
    transpose_sum_step<0x4E, 1>(v, (l & 2) != 0);
    double t = v[0];
    t += dpp_perm<0xB1>(t);
    return t;
}
__device__ __forceinline__ double row16_transpose_sum16(double (&v)[16], int l)
{   // lane l of the row returns the row total of slot l
    transpose_sum_step<0x140, 8>(v, (l & 8) != 0);
    transpose_sum_step<0x141, 4>(v, (l & 4) != 0);
    transpose_sum_step<0x4E, 2>(v, (l & 2) != 0);
    transpose_sum_step<0xB1, 1>(v, (l & 1) != 0);
    return v[0];
}

__device__ __forceinline__ void glcm_features_wave16(const uint32_t* Pslots, int n_slots, int Ng, double* scr_base, int scr_stride, double soft_nan,
                                                     double* fslots, double* sums, int lane)
{
    const int NN = Ng * Ng;
    const int slot_raw = lane >> 4, l = lane & 15;
    const bool live = slot_raw < n_slots;
    const int slot = live ? slot_raw : 0;              // idle rows shadow slot 0 and never store
    const uint32_t* P = Pslots + mul24((uint32_t)slot, (uint32_t)NN);
    double* pcol_s = scr_base + mul24((uint32_t)slot, (uint32_t)scr_stride);     // px[i] = sum_j xy(i,j)/sum_p   (glcm.cpp:523-525, :859-864)
    double* prow_s = pcol_s + Ng;                      // py[j] = sum_i xy(i,j)/sum_p
    double* Pxpy = pcol_s + 2 * Ng;                    // [2Ng]  glcm.cpp:503-508
    double* f = fslots + slot * 32;
    double* sm = sums + slot * 32;

    // ---- marginal counts: lane i < Ng owns column i, row i and the diagonal pair |x - y| = i ------------------------------
    // (indices advance by additions, products are 24-bit -- full-rate instructions -- and the loops stay rolled: their control is
    //  scalar or a mask update, while the unroller's prologue / remainder code is vector work)
    uint32_t cc = 0, rc = 0, dc = 0;
    if (l < Ng) {
        const uint32_t lN = mul24((uint32_t)l, (uint32_t)Ng);
        uint32_t ic = (uint32_t)l;
#pragma unroll 1
        for (int j = 0; j < Ng; j++, ic += (uint32_t)Ng) {
            cc += P[ic];                               // P[j * Ng + l]
            rc += P[lN + (uint32_t)j];                 // P[l * Ng + j]
        }
        uint32_t i1 = lN, i2 = (uint32_t)l;            // P[x * Ng + (x - l)], P[(x - l) * Ng + x] for x = l ..: both step by Ng + 1
#pragma unroll 1
        for (int x = l; x < Ng; x++, i1 += (uint32_t)Ng + 1u, i2 += (uint32_t)Ng + 1u) {
            dc += P[i1];
            if (l > 0)
                dc += P[i2];
        }
    }
    const uint32_t l1 = (uint32_t)l + 1u;
    const uint32_t csum = row16_sum(rc);               // sum_p (glcm.cpp:481-484)
    const uint32_t Sr_i = row16_sum(mul24(rc, l1)), Sc_i = row16_sum(mul24(cc, l1));          // f_corr mr :601, mc :608 (exact numerators)
    const uint32_t con_i = row16_sum(mul24(dc, (uint32_t)(l * l))), dis_i = row16_sum(mul24(dc, (uint32_t)l));   // f_contrast :579, f_GLCM_DIS :1052
    const bool empty = csum == 0;                      // glcm.cpp:260-295 -> soft NaN for this angle
    const double sum_p = empty ? 1.0 : (double)csum;
    const double inv_sum_p = fdiv(1.0, sum_p);
    const double mr = fdiv((double)Sr_i, sum_p), mc = fdiv((double)Sc_i, sum_p);
    const double pcol = fdiv((double)cc, sum_p), prow = fdiv((double)rc, sum_p), pxmy = fdiv((double)dc, sum_p);
    if (live && l < Ng) { pcol_s[l] = pcol; prow_s[l] = prow; }
    for (int k = l; k < 2 * Ng - 1; k += 16) {
        uint32_t c = 0;
        const int x0 = k - (Ng - 1) > 0 ? k - (Ng - 1) : 0, x1 = k < Ng - 1 ? k : Ng - 1;
        uint32_t ix = mad24((uint32_t)x0, (uint32_t)Ng, (uint32_t)(k - x0));     // P[x * Ng + (k - x)]: steps by Ng - 1
#pragma unroll 1
        for (int x = x0; x <= x1; x++, ix += (uint32_t)Ng - 1u)
            c += P[ix];
        if (live) Pxpy[k] = fdiv((double)c, sum_p);
    }
    wav_sync<false>();

    // ---- the cell pass ---------------------------------------------------------------------------------------------------
    // What a cell contributes on its own is integer work: ASM = sum cnt^2 / sum_p^2, ACOR and the covariance term from
    // sum cnt (r+1)(c+1) (the numerator  acor * sum_p - S_r * S_c  is an exact integer below 2^53), JMAX from the largest
    // count; only the entropy term p lg(p + eps) and the two HXY terms need the float log.
    double ent = 0, hxy1 = 0, hxy2 = 0, asm_d = 0;
    uint32_t acor_i = 0, asm_i = 0, cmax = 0;
    const bool big = csum >= 65536u;                                   // (symmetric matrices of ROIs beyond 32767 pixels)
    RowCol rcw((uint32_t)l, 16u, (uint32_t)Ng, RowCol::small_t{});
    for (int e = l; e < NN; e += 16, rcw.advance()) {
        const uint32_t r = rcw.row, c = rcw.col, cnt = P[e];
        const double p = (double)cnt * inv_sum_p;
        if (big) asm_d = __builtin_fma(p, p, asm_d);                 // f_asm :555 / f_energy :927-928
        else asm_i = mad24(cnt, cnt, asm_i);                         //   (sum cnt^2 <= (sum cnt)^2 < 2^32 while sum_p < 65536)
        acor_i = mad24(cnt, mul24(r + 1u, c + 1u), acor_i);          // f_GLCM_ACOR :961 (integer-exact; a count is far below 2^24)
        cmax = cnt > cmax ? cnt : cmax;                              // f_GLCM_JMAX :1178-1179
        ent = __builtin_fma(p, (double)fast_log2f(p + 0.000000001), ent);       // f_entropy :734-735, JE :1160-1161, HXY :868
        const double pp = pcol_s[c] * prow_s[r];                     // px[i]*py[j], i = column, j = row (:869, :909)
        const double lg = (double)fast_log2f(pp + 0.000000001);
        hxy1 = __builtin_fma(p, lg, hxy1);
        hxy2 = __builtin_fma(pp, lg, hxy2);
    }
    const double hx_t = l < Ng ? plogp(pcol, pcol) : 0.0;             // :873-874
    {
        double t4[4] = {ent, hxy1, hxy2, hx_t};
        transpose_sum_step<0x140, 2>(t4, (l & 8) != 0);              // lane l of the row ends with the row total of slot (l >> 2) & 3
        transpose_sum_step<0x141, 1>(t4, (l & 4) != 0);
        double tot = t4[0];
        tot += dpp_perm<0x4E>(tot);
        tot += dpp_perm<0xB1>(tot);
        if (live && (l & 3) == 0) sm[1 + (l >> 2)] = tot;            // sm[1] ent, [2] hxy1, [3] hxy2, [4] hx
    }
    acor_i = row16_sum(acor_i);
    asm_i = row16_sum(asm_i);
    if (big) asm_d = row16_sum(asm_d);
    {
        uint32_t o;
        o = dpp_perm<0xB1>(cmax); cmax = o > cmax ? o : cmax;
        o = dpp_perm<0x4E>(cmax); cmax = o > cmax ? o : cmax;
        o = dpp_perm<0x141>(cmax); cmax = o > cmax ? o : cmax;
        o = dpp_perm<0x140>(cmax); cmax = o > cmax ? o : cmax;
    }

    // ---- one term per lane: features of the marginal distributions ---------------------------------------------------------
    double t16[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t16[k] = 0.0;
    if (l < Ng) {
        const double dr = (double)l1 - mr, dcl = (double)l1 - mc, dr2 = dr * dr;
        t16[0] = prow * dr2;                                         // f_corr :617
        t16[1] = pcol * (dcl * dcl);                                 // :626
        t16[2] = (double)rc * dr2;                                   // f_var :672
        t16[3] = pcol * dr2;                                         // f_GLCM_JVAR :1196-1199 (x = column, +1 index)
        const double q = pxmy, kd = (double)l, Ngd = (double)Ng;
        t16[4] = fdiv(q, (double)(1 + l * l));                       // f_idm :685-687
        t16[5] = q != 0 ? plogp(q, q) : 0.0;                         // f_dentropy :778-781
        t16[6] = fdiv(q, 1.0 + fdiv(kd * kd, Ngd * Ngd));            // :1083-1084
        t16[7] = fdiv(q, 1.0 + kd);                                  // :1096-1097
        t16[8] = fdiv(q, 1.0 + fdiv(kd, Ngd));                       // :1110-1111
        t16[9] = l >= 1 ? q / (kd * kd) : 0.0;                       // :1123-1128
        t16[10] = kd * q;                                            // f_difference_avg :791-792
    }
    for (int k = l; k < 2 * Ng - 1; k += 16) {
        const double q = Pxpy[k], ks = (double)(k + 2);              // I[x] + I[k - x] = k + 2
        t16[11] += ks * q;                                           // f_savg :700-701
        t16[12] += plogp(q, q);                                      // f_sentropy :712-716
        const double m = ks - mc - mc, m2 = m * m;                   // by_row_mean (:531-536) = mc; CLUPROM :985, CLUSHADE :1007, CLUTEND :1034
        t16[13] += m2 * m2 * q;
        t16[14] += m2 * m * q;
        t16[15] += m2 * q;
    }
    {
        const double tot = row16_transpose_sum16(t16, l);
        if (live) sm[8 + l] = tot;
    }
    wav_sync<false>();
    // f_dvar (glcm.cpp:742-766): every k receives the same term Ng times and the total is divided by Ng
    const double davg = sm[8 + 10];
    double dv = 0;
    if (l < Ng) { const double dk = (double)l - davg; dv = dk * dk * pxmy; }
    dv = row16_sum(dv);

    if (live && l == 0) {
        const double ent_t = sm[1], hxy1_t = sm[2], hxy2_t = sm[3], hx = sm[4];
        const double asm_t = big ? asm_d : (double)asm_i * inv_sum_p * inv_sum_p;
        // (quotients by sum_p as products with its reciprocal; the correlation's sqrt(a) sqrt(b) denominator as one reciprocal root)
        const double cov_t = ((double)acor_i * sum_p - (double)Sr_i * (double)Sc_i) * (inv_sum_p * inv_sum_p);   // sum (r - mr)(c - mc) p, exact numerator
        f[G_ASM] = asm_t;
        f[G_ENERGY] = asm_t;
        f[G_CONTRAST] = (double)con_i * inv_sum_p;
        f[G_ACOR] = (double)acor_i * inv_sum_p;
        f[G_ENTROPY] = -ent_t;
        f[G_JE] = -ent_t;
        f[G_DIS] = (double)dis_i * inv_sum_p;
        f[G_JMAX] = (double)cmax * inv_sum_p;
        f[G_JAVE] = mr;
        f[G_VARIANCE] = sm[8 + 2] * inv_sum_p;
        f[G_CLUPROM] = sm[8 + 13];
        f[G_CLUSHADE] = sm[8 + 14];
        f[G_CLUTEND] = sm[8 + 15];
        f[G_SUMVARIANCE] = sm[8 + 15];                // glcm.cpp:323-326
        f[G_JVAR] = sm[8 + 3];
        const double vv = sm[8 + 0] * sm[8 + 1];                      // f_corr tail, glcm.cpp:619-643: cov / (sqrt(var_r) sqrt(var_c))
        f[G_CORRELATION] = !(sm[8 + 0] > 0.0 && sm[8 + 1] > 0.0) ? soft_nan : cov_t * frsq(vv);
        f[G_INFOMEAS2] = sqrt(fabs(1 - exp(-2 * (-hxy2_t + ent_t)))); // glcm.cpp:913 (HXY = ent)
        f[G_IDM] = sm[8 + 4];
        f[G_HOM2] = sm[8 + 4];                        // f_GLCM_HOM2 :1069 == f_idm over p_{x-y}
        f[G_HOM1] = sm[8 + 7];                        // f_homogeneity :942 == f_GLCM_ID over p_{x-y}
        f[G_SUMAVERAGE] = sm[8 + 11];
        f[G_SUMENTROPY] = -sm[8 + 12];
        f[G_DIFENTRO] = -sm[8 + 5];
        f[G_DIFAVE] = davg;
        f[G_DIFVAR] = dv;
        f[G_IDMN] = sm[8 + 6];
        f[G_ID] = sm[8 + 7];
        f[G_IDN] = sm[8 + 8];
        f[G_IV] = sm[8 + 9];
        const double r1 = (ent_t - hxy1_t) / hx;      // f_info_meas_corr1, glcm.cpp:880-883
        f[G_INFOMEAS1] = isfinite(r1) ? r1 : soft_nan;
        if (empty)                                    // blank matrix: all 30 values = soft NaN
            for (int k = 0; k < kGlcmAngled; k++)
                f[k] = soft_nan;
    }
    wav_sync<false>();
}

// ---- the same for matrices of up to 8 levels: EIGHT lanes per angle, two ROIs per wave ---------------------------------------------
// At eight levels (the metric's coarse grey depth) half of a 16-lane row idles in everything but the cell pass.  Here a group of
// eight lanes owns an angle; groups 0-3 take the angles of the wave's first ROI, groups 4-7 those of its second: the instruction
// stream is issued once for two ROIs (1086 -> ~600 vector instructions per ROI).  Same quantities, same formulas as
// glcm_features_wave16; only the grouping of the lane sums differs (1e-15 level).
__device__ __forceinline__ uint32_t grp8_sum(uint32_t v) { v += dpp_perm<0xB1>(v); v += dpp_perm<0x4E>(v); v += dpp_perm<0x141>(v); return v; }
__device__ __forceinline__ double grp8_sum(double v) { v += dpp_perm<0xB1>(v); v += dpp_perm<0x4E>(v); v += dpp_perm<0x141>(v); return v; }
__device__ __forceinline__ void glcm_features_wave8(const uint32_t* P_a, const uint32_t* P_b, int n_slots, int Ng_a, int Ng_b, double* scr_base, int scr_stride,
                                                    double soft_nan, double* fslots, size_t f_gap, double* sums, int lane)
{
    const int grp = lane >> 3, l = lane & 7, sel = grp >> 2, slot_raw = grp & 3;
    const int Ng = sel ? Ng_b : Ng_a;
    const int NN = Ng * Ng;
    const bool live = slot_raw < n_slots && Ng != 0;
    const int slot = slot_raw < n_slots ? slot_raw : 0;            // idle groups shadow slot 0 of their ROI and never store
    const uint32_t* P = (sel ? P_b : P_a) + mul24((uint32_t)slot, (uint32_t)NN);
    const int gslot = sel * kMaxAngles + slot;
    double* pcol_s = scr_base + mul24((uint32_t)gslot, (uint32_t)scr_stride);
    double* prow_s = pcol_s + 8;
    double* f = fslots + gslot * 32 + (sel ? f_gap : 0);          // (the kernel lays an ROI's features over its count block: written after the last read of a count)
    double* sm = sums + gslot * 32;

    // ---- marginal counts: lane i < Ng owns column i, row i and the diagonal pair |x - y| = i
    uint32_t cc = 0, rc = 0, dc = 0;
    if (l < Ng) {
        const uint32_t lN = mul24((uint32_t)l, (uint32_t)Ng);
        uint32_t ic = (uint32_t)l;
#pragma unroll 1
        for (int j = 0; j < Ng; j++, ic += (uint32_t)Ng) {
            cc += P[ic];
            rc += P[lN + (uint32_t)j];
        }
        uint32_t i1 = lN, i2 = (uint32_t)l;
#pragma unroll 1
        for (int x = l; x < Ng; x++, i1 += (uint32_t)Ng + 1u, i2 += (uint32_t)Ng + 1u) {
            dc += P[i1];
            if (l > 0) dc += P[i2];
        }
    }
    const uint32_t l1 = (uint32_t)l + 1u;
    const uint32_t csum = grp8_sum(rc);                            // sum_p (glcm.cpp:481-484)
    const uint32_t Sr_i = grp8_sum(mul24(rc, l1)), Sc_i = grp8_sum(mul24(cc, l1));
    const uint32_t con_i = grp8_sum(mul24(dc, (uint32_t)(l * l))), dis_i = grp8_sum(mul24(dc, (uint32_t)l));
    const bool empty = csum == 0;                                  // glcm.cpp:260-295 -> soft NaN for this angle
    const double sum_p = empty ? 1.0 : (double)csum;
    const double inv_sum_p = fdiv(1.0, sum_p);
    const double mr = fdiv((double)Sr_i, sum_p), mc = fdiv((double)Sc_i, sum_p);
    const double pcol = fdiv((double)cc, sum_p), prow = fdiv((double)rc, sum_p), pxmy = fdiv((double)dc, sum_p);
    if (live && l < Ng) { pcol_s[l] = pcol; prow_s[l] = prow; }
    double pxpy[2] = {0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int k = l + 8 * u;
        if (k < 2 * Ng - 1) {
            uint32_t c = 0;
            const int x0 = k - (Ng - 1) > 0 ? k - (Ng - 1) : 0, x1 = k < Ng - 1 ? k : Ng - 1;
            uint32_t ix = mad24((uint32_t)x0, (uint32_t)Ng, (uint32_t)(k - x0));
#pragma unroll 1
            for (int x = x0; x <= x1; x++, ix += (uint32_t)Ng - 1u) c += P[ix];
            pxpy[u] = fdiv((double)c, sum_p);
        }
    }
    wav_sync<false>();
#if defined(NYX_G8_EXIT) && NYX_G8_EXIT == 1     // diagnostic builds (tools/g8_exit_libs.sh): the routine ends here; results are wrong by design
    if (live && l < Ng) f[l] = pcol + prow + pxmy + pxpy[0] + pxpy[1] + mr + mc + (double)(con_i + dis_i);
    return;
#endif

    // ---- the cell pass
    double ent = 0, hxy1 = 0, hxy2 = 0, asm_d = 0;
    uint32_t acor_i = 0, asm_i = 0, cmax = 0;
    const bool big = csum >= 65536u;
    {
        RowCol rcw((uint32_t)l, 8u, (uint32_t)(Ng ? Ng : 1), RowCol::small_t{});
        for (int e = l; e < NN; e += 8, rcw.advance()) {
            const uint32_t r = rcw.row, c = rcw.col, cnt = P[e];
            const double p = (double)cnt * inv_sum_p;
            if (big) asm_d = __builtin_fma(p, p, asm_d);
            else asm_i = mad24(cnt, cnt, asm_i);
            acor_i = mad24(cnt, mul24(r + 1u, c + 1u), acor_i);
            cmax = cnt > cmax ? cnt : cmax;
            ent = __builtin_fma(p, (double)fast_log2f(p + 0.000000001), ent);
            const double pp = pcol_s[c] * prow_s[r];
            const double lg = (double)fast_log2f(pp + 0.000000001);
            hxy1 = __builtin_fma(p, lg, hxy1);
            hxy2 = __builtin_fma(pp, lg, hxy2);
        }
    }
#if defined(NYX_G8_EXIT) && NYX_G8_EXIT == 2
    if (live && l < Ng) f[l] = ent + hxy1 + hxy2 + asm_d + (double)(acor_i + asm_i + cmax);
    return;
#endif
    const double hx_t = l < Ng ? plogp(pcol, pcol) : 0.0;
    {
        double t4[4] = {ent, hxy1, hxy2, hx_t};
        transpose_sum_step<0x141, 2>(t4, (l & 4) != 0);              // lane l of the group ends with the group total of slot 2 * bit2 + bit1
        transpose_sum_step<0x4E, 1>(t4, (l & 2) != 0);
        double tot = t4[0];
        tot += dpp_perm<0xB1>(tot);
        if (live && (l & 1) == 0) sm[1 + (l >> 1)] = tot;            // sm[1] ent, [2] hxy1, [3] hxy2, [4] hx
    }
    acor_i = grp8_sum(acor_i);
    asm_i = grp8_sum(asm_i);
    if (big) asm_d = grp8_sum(asm_d);
    {
        uint32_t o;
        o = dpp_perm<0xB1>(cmax); cmax = o > cmax ? o : cmax;
        o = dpp_perm<0x4E>(cmax); cmax = o > cmax ? o : cmax;
        o = dpp_perm<0x141>(cmax); cmax = o > cmax ? o : cmax;
    }

    // ---- one term per lane: features of the marginal distributions, in two halves of eight slots (sixteen doubles live at once cost the
    // kernel its eighth wave per SIMD)
    {
        double t8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t8[k] = 0.0;
        if (l < Ng) {
            const double dr = (double)l1 - mr, dcl = (double)l1 - mc, dr2 = dr * dr;
            t8[0] = prow * dr2;                                          // f_corr :617
            t8[1] = pcol * (dcl * dcl);                                  // :626
            t8[2] = (double)rc * dr2;                                    // f_var :672
            t8[3] = pcol * dr2;                                          // f_GLCM_JVAR :1196-1199
            const double q = pxmy, kd = (double)l, Ngd = (double)Ng;
            t8[4] = fdiv(q, (double)(1 + l * l));                        // f_idm :685-687
            t8[5] = q != 0 ? plogp(q, q) : 0.0;                          // f_dentropy :778-781
            t8[6] = fdiv(q, 1.0 + fdiv(kd * kd, Ngd * Ngd));             // :1083-1084
            t8[7] = fdiv(q, 1.0 + kd);                                   // :1096-1097
        }
        transpose_sum_step<0x141, 4>(t8, (l & 4) != 0);                  // lane l of the group ends with the group total of slot l
        transpose_sum_step<0x4E, 2>(t8, (l & 2) != 0);
        transpose_sum_step<0xB1, 1>(t8, (l & 1) != 0);
        if (live) sm[8 + l] = t8[0];
    }
    {
        double t8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t8[k] = 0.0;
        if (l < Ng) {
            const double q = pxmy, kd = (double)l, Ngd = (double)Ng;
            t8[0] = fdiv(q, 1.0 + fdiv(kd, Ngd));                        // :1110-1111
            t8[1] = l >= 1 ? q / (kd * kd) : 0.0;                        // :1123-1128
            t8[2] = kd * q;                                              // f_difference_avg :791-792
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = l + 8 * u;
            if (k < 2 * Ng - 1) {
                const double q = pxpy[u], ks = (double)(k + 2);          // I[x] + I[k - x] = k + 2
                t8[3] += ks * q;                                         // f_savg :700-701
                t8[4] += plogp(q, q);                                    // f_sentropy :712-716
                const double m = ks - mc - mc, m2 = m * m;               // CLUPROM :985, CLUSHADE :1007, CLUTEND :1034
                t8[5] += m2 * m2 * q;
                t8[6] += m2 * m * q;
                t8[7] += m2 * q;
            }
        }
        transpose_sum_step<0x141, 4>(t8, (l & 4) != 0);
        transpose_sum_step<0x4E, 2>(t8, (l & 2) != 0);
        transpose_sum_step<0xB1, 1>(t8, (l & 1) != 0);
        if (live) sm[16 + l] = t8[0];
    }
    wav_sync<false>();
#if defined(NYX_G8_EXIT) && NYX_G8_EXIT == 3
    return;
#endif
    const double davg = sm[8 + 10];
    double dv = 0;                                                   // f_dvar (glcm.cpp:742-766)
    if (l < Ng) { const double dk = (double)l - davg; dv = dk * dk * pxmy; }
    dv = grp8_sum(dv);

    if (live && l == 0) {
        const double ent_t = sm[1], hxy1_t = sm[2], hxy2_t = sm[3], hx = sm[4];
        const double asm_t = big ? asm_d : (double)asm_i * inv_sum_p * inv_sum_p;
        const double cov_t = ((double)acor_i * sum_p - (double)Sr_i * (double)Sc_i) * (inv_sum_p * inv_sum_p);
        f[G_ASM] = asm_t;
        f[G_ENERGY] = asm_t;
        f[G_CONTRAST] = (double)con_i * inv_sum_p;
        f[G_ACOR] = (double)acor_i * inv_sum_p;
        f[G_ENTROPY] = -ent_t;
        f[G_JE] = -ent_t;
        f[G_DIS] = (double)dis_i * inv_sum_p;
        f[G_JMAX] = (double)cmax * inv_sum_p;
        f[G_JAVE] = mr;
        f[G_VARIANCE] = sm[8 + 2] * inv_sum_p;
        f[G_CLUPROM] = sm[8 + 13];
        f[G_CLUSHADE] = sm[8 + 14];
        f[G_CLUTEND] = sm[8 + 15];
        f[G_SUMVARIANCE] = sm[8 + 15];
        f[G_JVAR] = sm[8 + 3];
        const double vv = sm[8 + 0] * sm[8 + 1];
        f[G_CORRELATION] = !(sm[8 + 0] > 0.0 && sm[8 + 1] > 0.0) ? soft_nan : cov_t * frsq(vv);
        f[G_INFOMEAS2] = sqrt(fabs(1 - exp(-2 * (-hxy2_t + ent_t))));
        f[G_IDM] = sm[8 + 4];
        f[G_HOM2] = sm[8 + 4];
        f[G_HOM1] = sm[8 + 7];
        f[G_SUMAVERAGE] = sm[8 + 11];
        f[G_SUMENTROPY] = -sm[8 + 12];
        f[G_DIFENTRO] = -sm[8 + 5];
        f[G_DIFAVE] = davg;
        f[G_DIFVAR] = dv;
        f[G_IDMN] = sm[8 + 6];
        f[G_ID] = sm[8 + 7];
        f[G_IDN] = sm[8 + 8];
        f[G_IV] = sm[8 + 9];
        const double r1 = (ent_t - hxy1_t) / hx;
        f[G_INFOMEAS1] = isfinite(r1) ? r1 : soft_nan;
        if (empty)
            for (int k = 0; k < kGlcmAngled; k++) f[k] = soft_nan;
    }
    wav_sync<false>();
}

// ---- GLCM features of a matrix of up to 64 levels: one wave per angle (the reference's default grey depth) -------------------------
// Same organisation as glcm_features_wave16 (marginal sums carry everything they can, only ASM / ENTROPY / JMAX / HXY1 / HXY2 visit
// the cells), on the 16-bit matrices of the G16 launches.  Level values are I[i] = i + 1 (matlab binning).
//  * Marginal counts in ONE pass over the rows (lane = column): the column sums accumulate in place, and the two families of
//    diagonals ride along in registers that move one lane per row -- W(lane) holds the partial sum of the diagonal through (r, lane)
//    (index c - r, shifts right), X(lane) that of the anti-diagonal (index r + c, shifts left).  A diagonal is complete when it leaves
//    the matrix: upper diagonals (c - r = d >= 0) end in lane Ng - 1 at row Ng - 1 - d, anti-diagonals k <= Ng - 2 end in lane 0 at
//    row k.  What is still travelling after the last row -- the lower diagonals in W, the anti-diagonals k >= Ng - 1 in X -- is fetched
//    with one ds_bpermute each.
//  * A cell's own quantities are integer work (ASM = sum cnt^2 / sum_p^2, JMAX from the largest count; sum_p < 65536 in a G16 launch,
//    so every sum fits 32 bits); ACOR and the covariance numerator acor * sum_p - S_r * S_c come exactly from the diagonals; the
//    entropy term p lg(p + eps) depends on the count alone: small counts -- all of them on a textured ROI -- come from a table built
//    once per angle.
//  * HXY1 / HXY2: rows are visited in GROUPS of equal row marginal: the float log of p_x(i) p_y(j) + eps -- the reference's quadratic
//    -- depends on the row only through its marginal count, and the 64 rows of a textured ROI share ~25 distinct counts (Poisson around
//    n / 64).  One log per group and column; HXY2 takes the group's multiplicity, HXY1 the group's count per column against the
//    group's log (scaled by 1 / sum_p once at the end).  Rows with an empty marginal hold no pair and are skipped.  The group loop
//    runs on the scalar unit (ballot, s_ff1, bit clears).
// Round 6: matrices of pitch `pitch` (even) whose data cells are word-aligned pairs.
// Layout (roi_features_kernel_g16): u16 cells, rows 0..Ng (row = centre level, row 0 unused), element (centre a, neighbour b) at
// P[a * pitch + b - 1], pitch = (Ng + 3) & ~1: data columns 0..Ng-1 hold neighbour levels 1..Ng, a skipped neighbour (level 0)
// lands in the LAST column of the previous row (never read), and column Ng of every row stays zero.  A row starts on a word, so
// two cells travel in one register: row sums with v_pk_add_u16, the per-cell terms of two rows at once (lanes 0..31 one row, lanes
// 32..63 another; v_dot2_u32_u16 for sum cnt^2, v_pk_max_u16, the entropy terms of BOTH cells of a word from one read of an
// 8 x 8 table), completed diagonals leave the travelling accumulators through one store per row from every lane (the lanes that
// do not hold a complete diagonal write to a junk strip: no exec juggling).  Per angle-wave 3136 -> ~2300 vector instructions.
// blk: 256 words of LDS per wave: [f: 32 doubles | sm: 32 doubles | T2: 64 doubles], the diagonal picks lie over all of it
// (PX 64 words | PW 64 words | junk 127 words) while nothing else is live.
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
#ifdef NYX_G16_EXIT      // diagnostic builds (tools/g16_exit_libs.sh): the feature pass ends after phase NYX_G16_EXIT; results are wrong by design
#define G16_EXIT(k, keep) do { if ((k) == NYX_G16_EXIT) { ((double*)blk)[lane & 31] = (double)(keep); return; } } while (0)
#else
#define G16_EXIT(k, keep) do { } while (0)
#endif
// ROLE (the eight-wave launch, roi_features_kernel_g16w8): 0 = an angle's whole pass on one wave.  1 / 2 = the pass on TWO waves: wave 2 forms
// the row and column sums it needs itself and runs the HXY1 / HXY2 row groups -- half of the pass's instructions -- while wave 1 does
// everything else; wave 2's per-lane sums cross over through the angle's block (T2 | f, sm: dead between the two workgroup barriers) and
// wave 1 closes with them: the same per-lane values in the same reductions as ROLE 0, bit for bit.  Waves without an angle take the two
// barriers and nothing else.
template <int NG, int ROLE = 0>
__device__ __forceinline__ void glcm_features_wave64_v2(const uint16_t* P, int Ng_rt, uint32_t* blk, double soft_nan, int lane)
{
    const int Ng = NG ? NG : Ng_rt;                    // (NG = 64: the reference's default grey depth as a compile-time fact -- row offsets become immediates)
    typedef __attribute__((address_space(3))) const uint16_t lds_cu16_t;
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32_t;
    typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
    const int pitch = (Ng + 3) & ~1;
    double* const f = (double*)blk;
    double* const sm = f + 32;
    double* const T2 = f + 64;
    const bool act = lane < Ng;                        // lane l owns column l, row l and the diagonal pair |x - y| = l
    const uint16_t* const Pd = P + pitch;              // row of level 1
    G16_EXIT(0, lane);
    // ---- row sums: lane = row, the row's words (pitch / 2 apart per lane: no bank conflict) ------------------------------------
    uint32_t rc = 0;
    {
        const int nw = (Ng + 1) >> 1;
        lds_cu32_t* rw = (lds_cu32_t*)(uintptr_t)(Pd + (act ? lane : 0) * pitch);
        us2_t a2 = {0, 0};
#pragma unroll 8
        for (int j = 0; j < nw; j++) a2 = a2 + __builtin_bit_cast(us2_t, rw[j]);
        rc = act ? (uint32_t)a2.x + (uint32_t)a2.y : 0u;
    }
    G16_EXIT(1, rc);
    if constexpr (ROLE == 2) {
        // ---- the partner wave: column sums (a plain walk down the column), the two marginal probabilities, the row groups ------------
        const int cj = lane < Ng ? lane : Ng;
        lds_cu16_t* pc = (lds_cu16_t*)(uintptr_t)(Pd + cj);
        uint32_t cc2 = 0;
#pragma unroll 8
        for (int i = 0; i < Ng; i++) cc2 += pc[i * pitch];
        if (!act) cc2 = 0;
        const uint32_t csum2 = wave_sum_t<uint32_t>(rc);
        const double sum_p2 = csum2 == 0 ? 1.0 : (double)csum2;
        const double pcol = fdiv((double)cc2, sum_p2), prow = fdiv((double)rc, sum_p2);
        double hxy1c = 0, hxy2 = 0;
        unsigned long long rem = __ballot(act && rc != 0u);
        while (rem) {
            const int r0 = (int)__builtin_ctzll(rem);
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)rc, r0);
            const unsigned long long pbits = (unsigned long long)__double_as_longlong(prow);
            const double pr = __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pbits >> 32), r0) << 32) |
                                                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pbits, r0)));
            unsigned long long grp = __ballot(act && rc == v);
            rem &= ~grp;
            const double pp = pcol * pr;                                 // px[i]*py[j], i = column, j = row (:869, :909)
            const double lg = (double)fast_log2f(pp + 0.000000001);
            hxy2 = __builtin_fma(pp * (double)(uint32_t)__popcll(grp), lg, hxy2);
            uint32_t gsum = 0;
            while (grp) {
                const int r = (int)__builtin_ctzll(grp);
                grp &= grp - 1ull;
                gsum += pc[r * pitch];
            }
            hxy1c = __builtin_fma((double)gsum, lg, hxy1c);
        }
        blk_sync<false>();                             // wave 1 is through with the block's table
        T2[lane] = hxy1c;
        f[lane] = hxy2;                                // (f | sm: 64 doubles)
        blk_sync<false>();
        return;
    }
    // ---- column sums and the two families of diagonals in one pass over the rows (lane = column; see the header) --------------
    uint32_t cc = 0, dc = 0;
    uint32_t pxpy_c[2] = {0u, 0u};
    {
        uint32_t W = 0, X = 0;
        const int cj = lane < Ng ? lane : Ng;          // (lanes beyond the matrix read the zero column)
        lds_cu16_t* pc = (lds_cu16_t*)(uintptr_t)(Pd + cj);
        lds_u32_t* const b0 = (lds_u32_t*)(uintptr_t)blk;
        lds_u32_t* px = lane == 0 ? b0 : b0 + 128 + lane;           // [i]: anti-diagonal i, complete in lane 0 after row i
        lds_u32_t* pw = lane == Ng - 1 ? b0 + 64 : b0 + 128 + lane; // [i]: upper diagonal Ng - 1 - i, complete in lane Ng - 1 after row i
#pragma unroll 8
        for (int i = 0; i < Ng; i++) {
            const uint32_t cnt = pc[i * pitch];
            cc += cnt;
            X = lane_plus1_z(X) + cnt;
            W = lane_minus1_z(W) + cnt;
            px[i] = X;
            pw[i] = W;
        }
        wav_sync<false>();
        const uint32_t dlo = act ? blk[lane] : 0u;
        const uint32_t eup = act ? blk[64 + Ng - 1 - lane] : 0u;
        wav_sync<false>();
        // lower diagonal d (>= 1) waits in lane Ng - 1 - d of W; anti-diagonal k >= Ng - 1 in lane k - (Ng - 1) of X
        const uint32_t wrev = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (Ng - 1 - lane), (int)W);
        dc = act ? eup + (lane >= 1 ? wrev : 0u) : 0u;
        const int k1 = lane + 64;
        const uint32_t xa = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane - (Ng - 1)), (int)X);
        const uint32_t xb = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (k1 - (Ng - 1)), (int)X);
        pxpy_c[0] = lane < Ng ? dlo : (lane <= 2 * Ng - 2 ? xa : 0u);
        pxpy_c[1] = k1 <= 2 * Ng - 2 ? xb : 0u;
        if (!act) cc = 0;
    }
    G16_EXIT(2, rc + cc + dc + pxpy_c[0] + pxpy_c[1]);
    const uint32_t csum = wave_sum_t<uint32_t>(rc);    // sum_p (glcm.cpp:481-484)
    const bool empty = csum == 0;
    const double sum_p = empty ? 1.0 : (double)csum;
    const double inv_sum_p = fdiv(1.0, sum_p);
    const double pcol = fdiv((double)cc, sum_p), prow = fdiv((double)rc, sum_p), pxmy = fdiv((double)dc, sum_p);
    double pxpy[2] = {0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 2; u++)
        if (lane + 64 * u < 2 * Ng - 1) pxpy[u] = fdiv((double)pxpy_c[u], sum_p);

    // ---- entropy terms of a word's two cells: T2[a + 8 b] = T[a] + T[b], T[c] = (c / sum_p) lg(c / sum_p + eps) for c <= 6, T[7] = 0
    // (counts of 7 and more -- one ROI-angle in thirty on the benchmark's uniform data -- are added by a walk of their own below)
    {
        const int ca = lane & 7;
        const double pk = (double)ca * inv_sum_p;
        const double t = ca < 7 ? pk * (double)fast_log2f(pk + 0.000000001) : 0.0;
        if (lane < 8) sm[lane] = t;
        wav_sync<false>();
        T2[lane] = t + sm[lane >> 3];
        wav_sync<false>();
    }
    G16_EXIT(3, (uint32_t)(pcol + prow + pxmy + pxpy[0] + pxpy[1]));
    // ---- per-cell terms, two rows per step: lanes 0..31 row s, lanes 32..63 row s + half; a lane holds columns 2 l, 2 l + 1 -------
    // A cell's own quantities are integer work (ASM = sum cnt^2 / sum_p^2, JMAX from the largest count; sum_p < 65536 in a G16
    // launch, so every sum fits 32 bits); the entropy term p lg(p + eps) depends on the count alone.
    double ent = 0;
    uint32_t asm_i = 0, cmax = 0;
    {
        const int half = (Ng + 1) >> 1, l32 = lane & 31;
        const bool hi = lane >= 32, colp = 2 * l32 < Ng;
        lds_cu32_t* pwd = (lds_cu32_t*)(uintptr_t)(Pd + (hi ? half : 0) * pitch) + (colp ? l32 : 0);
        typedef __attribute__((address_space(3))) const double lds_cd_t;
        lds_cd_t* const T2l = (lds_cd_t*)(uintptr_t)T2;
        us2_t mx = {0, 0};
#pragma unroll 8
        for (int s = 0; s < half; s++) {
            uint32_t w = pwd[s * (pitch >> 1)];
            if (Ng & 1) w = (colp && (!hi || s + half < Ng)) ? w : 0u;   // (an even order: every lane holds two data columns of a row of the matrix)
            else if (Ng < 64) w = colp ? w : 0u;
            const us2_t v = __builtin_bit_cast(us2_t, w);
            asm_i = __builtin_amdgcn_udot2(v, v, asm_i, false);      // f_asm :555 / f_energy :927-928
            mx = __builtin_elementwise_max(mx, v);                   // f_GLCM_JMAX :1178-1179
            const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(v, (us2_t){7, 7}));
            ent += T2l[(m & 7u) | (m >> 13)];                        // f_entropy :734-735, JE :1160-1161, HXY :868
        }
        cmax = mx.x > mx.y ? (uint32_t)mx.x : (uint32_t)mx.y;
    }
    G16_EXIT(4, (uint32_t)(pcol + prow + pxmy + pxpy[0] + pxpy[1] + ent) + asm_i + cmax);
    // ---- HXY1 / HXY2: lane = column, rows visited in groups of equal row marginal (see the header) ---------------------------
    double hxy1c = 0, hxy2 = 0;
    if constexpr (ROLE == 0) {
        const int cj = lane < Ng ? lane : Ng;
        lds_cu16_t* pc = (lds_cu16_t*)(uintptr_t)(Pd + cj);
        unsigned long long rem = __ballot(act && rc != 0u);
        while (rem) {
            const int r0 = (int)__builtin_ctzll(rem);
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)rc, r0);
            const unsigned long long pbits = (unsigned long long)__double_as_longlong(prow);
            const double pr = __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pbits >> 32), r0) << 32) |
                                                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pbits, r0)));
            unsigned long long grp = __ballot(act && rc == v);
            rem &= ~grp;
            const double pp = pcol * pr;                                 // px[i]*py[j], i = column, j = row (:869, :909)
            const double lg = (double)fast_log2f(pp + 0.000000001);
            hxy2 = __builtin_fma(pp * (double)(uint32_t)__popcll(grp), lg, hxy2);
            uint32_t gsum = 0;
            while (grp) {
                const int r = (int)__builtin_ctzll(grp);
                grp &= grp - 1ull;
                gsum += pc[r * pitch];
            }
            hxy1c = __builtin_fma((double)gsum, lg, hxy1c);
        }
    }
    if (__ballot(cmax >= 7u)) {                                      // the entropy terms of the large counts, which the table left out
        const int cj = lane < Ng ? lane : Ng;
        lds_cu16_t* pc = (lds_cu16_t*)(uintptr_t)(Pd + cj);
        unsigned long long rem = __ballot(act && rc != 0u);
        while (rem) {
            const int r = (int)__builtin_ctzll(rem);
            rem &= rem - 1ull;
            const uint32_t cnt = pc[r * pitch];
            if (cnt >= 7u) { const double p = (double)cnt * inv_sum_p; ent += p * (double)fast_log2f(p + 0.000000001); }
        }
    }
    if constexpr (ROLE == 1) {
        blk_sync<false>();                             // (the partner wave writes its per-lane sums between the two barriers)
        blk_sync<false>();
        hxy1c = T2[lane];
        hxy2 = f[lane];
        wav_sync<false>();                             // (the tail writes sm)
    }
    G16_EXIT(5, (uint32_t)(pcol + prow + pxmy + pxpy[0] + pxpy[1] + ent + hxy1c + hxy2) + asm_i + cmax);
    glcm_w64_tail<NG>(Ng, lane, rc, cc, dc, pxpy_c, csum, sum_p, inv_sum_p, pcol, prow, pxmy, pxpy, ent, hxy1c, hxy2, asm_i, cmax, f);
}

// Diagnostic build (-DNYX_STAMP, tools/stamp_probe.py): wave 0 / lane 0 of every
// workgroup adds the cycles spent between consecutive stamps to A.stamps[phase].  The
// product build compiles the macro away.
#ifdef NYX_STAMP
#define STAMP(i)                                                                          \
    do {                                                                                  \
        unsigned long long t__ = __builtin_readcyclecounter();                            \
        if (tid == 0 && A.stamps) atomicAdd(&A.stamps[i], t__ - t_prev__);                \
        t_prev__ = __builtin_readcyclecounter();                                          \
    } while (0)
#elif defined(NYX_EXIT_AT)   // diagnostic build: the kernel ends at phase NYX_EXIT_AT (results are wrong by design; tools/pmc_exit.sh)
#define STAMP(i) do { if ((i) == NYX_EXIT_AT) return; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// ---- the fused kernel --------------------------------------------------------------
// GS: per-workgroup scratch in the global workspace instead of LDS (large-ROI launches).
// C16: the counting table holds 16-bit entries and the value buffer 16-bit offsets from the ROI minimum (every ROI of the
// launch has fewer than 65536 pixels and an intensity range the counting engine covers): half the LDS, so
// that -- in the build that asks the compiler for <= 96 VGPRs (roi_features_kernel_occ5) -- five workgroups instead of four
// share a CU.
// SPLIT: the launch exports the GLCM counts for glcm_features_kernel; the in-kernel feature code is compiled out.
// D8: the binned bounding-box plane holds 8-bit levels (grey depth <= 254); with C16 and SPLIT this brings the benchmark's
// carve-out under 20 KiB: eight workgroups per CU.
// FAM: family set known at compile time -- 0: read A.mask; 1: INTENSITY + GLCM under matlab binning with <= 16 levels (what
// make_layout's dense8 stands for); 2: INTENSITY alone.  The 64-VGPR tier is only launched for 1 and 2, so the run-time
// switches are compile-time facts there and their branches disappear from the pixel loops.
// WIN: where the pixels come from -- 0: a cloud (the window loader is compiled out), 1: the ROI's window of its tile (the cloud
// loader is compiled out), 2: decided at run time from A.win.  With both loaders in one kernel the 64-VGPR build carried ~90
// scalar-register spills through the whole body (every spill and reload a vector instruction: 1388 -> 1249 per wave without them).
// TIER: occupancy tier of the calling kernel; it only tags the kernel's private copy of glcm_features_rows (kRowsTag), so
// that caller and callee are always compiled for the same register budget.
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;     // a byte at an absolute LDS address

// the kernel's RoiArgs (its only parameter: offset 0 of the kernarg segment) behind a pointer that is opaque to the optimiser at the point of
// the call: loads through it stay scalar loads (constant address space) and stay BELOW the call
typedef const __attribute__((address_space(4))) RoiArgs KArgs;
__device__ __forceinline__ KArgs* late_kernarg()
{
    KArgs* p = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
// B->field: through the late view (LATE) or the kernel's by-value argument itself (the generic builds: exactly the code they had)
template <bool LATE>
struct KView {
    const RoiArgs& a;
    KArgs* p;
    __device__ __forceinline__ explicit KView(const RoiArgs& a_) : a(a_), p(LATE ? late_kernarg() : nullptr) {}
    __device__ __forceinline__ auto operator->() const
    {
        if constexpr (LATE) return p;
        else return &a;
    }
};
template <bool GS, bool C16, bool SPLIT, bool D8, int FAM = 0, int TIER = 4, bool G16 = false, int WIN = 2, int NW = 4>
__device__ __forceinline__ void roi_features_body(const RoiArgs& A, const uint64_t slot)
{
    // NW = waves per ROI.  Only 4 (one workgroup per ROI) is built.  A one-wave build -- four ROIs per workgroup, for the smallest
    // size class -- was measured and dropped: at 12 KB of LDS per wave (16-bit counting table) or 77 .. 120 VGPRs (sort engine) it
    // keeps 12 .. 24 ROIs in flight per CU against 8 here, but every ROI's chain of dependent LDS / HBM round trips gets longer:
    // 49-pixel ROIs 8.4 -> 7.3 ns, 253-pixel ROIs 8.7 -> 8.9 ns per ROI.  A two-wave build of the metric configuration
    // (round 4: what a wave pays once per phase whatever its share of the pixels is paid twice per ROI instead of four times, but
    // the same eight workgroups per CU are four waves per SIMD instead of eight) took 2.39 ms per 196 k ROIs against 2.16.
    // (round 6: EIGHT waves for the GLCM-only launch at 17..64 levels -- FAM 4, G16 -- whose 39.8 KB of matrices allow four workgroups per CU:
    //  the load, zeroing and co-occurrence sweep run on twice the waves; the feature pass keeps a wave per angle, the other four wait)
    static_assert((NW == 4 && NW == kMaxAngles) || (NW == 8 && G16 && FAM == 4), "one workgroup of four waves per ROI (and a wave per GLCM angle)");
    constexpr int BS = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char* const lds = GS ? A.sp.scratch + (size_t)blockIdx.x * A.sp.stride : lds_raw;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t roi;
    if (!roi_of_slot(A.sp, slot, A.n_roi, roi))
        return;

    // (workspace launches: regions below A.L.gs_lds_bytes live in the workgroup's LDS, the ROI-sized ones in the workspace)
    auto reg = [&](uint32_t o) -> unsigned char* { return (GS && o < A.L.gs_lds_bytes) ? lds_raw + o : lds + o; };
    double* s_red = (double*)reg(A.L.red);
    double* s_stat = (double*)reg(A.L.stat);
    uint32_t* s_lb100 = (uint32_t*)reg(A.L.lb100);
    uint32_t* s_lbc = (uint32_t*)reg(A.L.lbc);
    uint32_t* s_val = (uint32_t*)(lds + A.L.val);
    using dense_t = typename std::conditional<D8, uint8_t, uint16_t>::type;
    dense_t* s_dense = (dense_t*)(lds + (D8 ? 0u : A.L.dense));     // 8-bit plane launches: the plane opens the carve-out (make_layout)
    uint16_t* s_lvlmap = (uint16_t*)reg(A.L.lvlmap);
    uint32_t* s_P = (uint32_t*)reg(A.L.P);
    double* s_g = (double*)reg(A.L.gscr);

    const uint64_t off = A.px_offset[roi];
    const uint32_t n = (uint32_t)(A.px_offset[roi + 1] - off);
    const uint32_t w = A.bbox_w[roi], h = A.bbox_h[roi];
    const uint32_t area = w * h;
    const uint32_t vmin = A.min_inten[roi], vmax = A.max_inten[roi];
    if (!roi_in_launch(A.sp, n, w, h, vmax - vmin))
        return;                                       // another launch of this call serves the ROI's size class
    if (vmax - vmin < A.sp.min_range || (A.sp.max_range != 0u && vmax - vmin > A.sp.max_range))
        return;                                       // served by the histogram path (roi_large.hip) / the other launch of a wide-range class (run_class)
    constexpr bool FAST = FAM == 1 || FAM == 3;       // (3: GLCM alone under the same conditions -- BASELINE configs[2], and the GLCM columns of the 16-bit path)
    constexpr int kRowsTag = 16 + TIER * 4 + (C16 ? 2 : 0) + (D8 ? 1 : 0);
    const bool do_int = FAM == 1 || FAM == 2 || (FAM == 0 && (A.mask & NYXHIP_FAM_INTENSITY) != 0);
    const bool do_glcm = FAM == 1 || FAM == 3 || FAM == 4 || (FAM == 0 && (A.mask & NYXHIP_FAM_GLCM) != 0);   // (4: GLCM alone, any binning the generic code serves)
    double* const out_row = A.out + roi * A.ld;
#ifdef NYX_STAMP
    unsigned long long t_prev__ = __builtin_readcyclecounter();
#endif

    // order-statistics engine for this ROI: counting table when [vmin, vmax] fits
    const uint32_t range = vmax - vmin;
    const bool use_count = C16 ? do_int : (do_int && A.L.count_cap != 0 && range < A.L.count_cap);   // C16: every ROI of the launch counts
    if (C16 && do_int && range >= A.L.count_cap) {    // (a range beyond the table: the caller's statement about the batch was wrong)
        if (tid == 0) atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        return;
    }
    uint32_t* s_cnt = (uint32_t*)reg(A.L.cnt);
    // smallest power of two >= n (sort length of the fallback engine)
    uint32_t P2 = 1;
    while (P2 < n)
        P2 <<= 1;
    const bool radix = !C16 && A.L.radix != 0;        // launches whose ROIs all sort (wide intensity ranges): radix sort, no padding
    // ... on 16-bit keys (value - minimum, two bytes per pixel in both key buffers) when every range of the launch fits them: the
    // sorted values are then read as vmin + key (SV below) and the 32-bit value buffer does not exist
    const bool k16 = radix && A.L.radix_k16 != 0;
    if (n == 0 || (do_int && ((use_count || radix) ? n : P2) > A.L.sort_cap) || (do_glcm && area > A.L.dense_cap)) {
        if (SPLIT && A.glcm_ng && tid == 0)
            A.glcm_ng[roi] = 0;                       // nothing for glcm_features_kernel (a deferred ROI gets its features in the spill launch)
        if (n != 0 && A.sp.defer_large)
            return;                                   // handled by the spill launch that follows
        if (tid == 0 && n != 0)
            atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        for (int c = tid; c < A.n_cols; c += BS)
            out_row[c] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }

    // grey binning used by the co-occurrence scan (glcm.cpp:354,379-385)
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    if (FAST) __builtin_assume(greyInfo > 0 && greyInfo <= 16);
    if (G16) __builtin_assume(greyInfo > 16 && greyInfo <= 64);     // G16: matlab binning, 17..64 levels, 16-bit matrices (see the GLCM block)
    const double mslope = greyInfo > 0 ? (double)greyInfo / ((double)vmax - 0.) : 0.0;

    // ---- phase 0: clear LDS state; the output row is written in place (zeros first: features that are skipped stay 0; every
    //      later store to the row is separated from these by a workgroup barrier) ----------------------------------------
    for (int c = tid; c < A.n_cols; c += BS)
        out_row[c] = 0.0;
    if (do_glcm) {
        uint32_t* d32 = (uint32_t*)s_dense;       // region is 16-byte aligned and padded
        // (8-bit planes: 64 more zero bytes behind the last row -- the "row below" of the last row and the cell every lane beyond the
        //  box reads in the co-occurrence sweep; the cell that takes out-of-box coordinates lies behind them)
        for (uint32_t i = tid; i < (D8 ? (area + 64 + 3) / 4 : (area + 1) / 2); i += BS)
            d32[i] = 0;
        if (greyInfo < 0)
            for (uint32_t i = tid; i <= A.L.lvl_cap; i += BS)
                s_lvlmap[i] = 0;
    }
    if (do_int) {
        if (use_count) {
            uint4* c4 = (uint4*)s_cnt;             // count_cap is a multiple of 64 entries
            const uint32_t n16 = C16 ? (range + 8) / 8 : (range + 4) / 4;   // 16-byte stores
            for (uint32_t i = tid; i < n16; i += BS)
                c4[i] = make_uint4(0, 0, 0, 0);
        } else if (!radix) {
            for (uint32_t i = n + tid; i < P2; i += BS)
                s_val[i] = 0xFFFFFFFFu;
        }
    }
    if (tid < 16)
        s_stat[tid] = 0.0;
    grp_sync<GS, NW>();

    STAMP(0);
    // ---- phase 1: the only pass over HBM ------------------------------------------------
    unsigned long long sum = 0, sumsq = 0;
    uint32_t lvl_max = 0;
    // four pixels per thread per trip: all global loads of a trip are issued before the first
    // use, so one HBM round trip covers 1024 pixels of the workgroup
    // The cloud is read through buffer descriptors sized to this ROI (raw buffers, num_records in bytes): lanes past the last
    // pixel get 0 from the bounds check of the load itself -- no compare, branch or 64-bit address per load, and the twelve
    // loads of a trip issue back to back.
    constexpr int kU = 4;
    const bool small_v = vmax < (1u << 24);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)(A.inten + off), 0, (int)(n * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(A.x + off), 0, (int)(n * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(A.y + off), 0, (int)(n * 2u), 0x00020000);
    // Per-pixel work of the pass.  Three facts that hold for a whole trip (1024 pixels) or a whole ROI are compile-time
    // parameters of the body, so that the common case carries no per-pixel test for them:
    //   FULL  every lane of the trip has a pixel (all trips but the last);
    //   NZ    the ROI has no zero intensity (vmin > 0): the co-occurrence scan's "original intensity 0 is skipped" test vanishes;
    //   TINY  vmax < 2^15: the four values / squares of a trip sum in 32 bits, one 64-bit add per trip instead of per pixel.
    auto trip = [&](auto full_c, auto nz_c, auto tiny_c, uint32_t base) {
        constexpr bool FULL = decltype(full_c)::value, NZ = decltype(nz_c)::value, TINY = decltype(tiny_c)::value;
        uint32_t v[kU], px[kU], py[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const uint32_t i = base + u * BS + tid;
            v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_v, (int)(i * 4u), 0, 0);
            px[u] = do_glcm ? (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_x, (int)(i * 2u), 0, 0) : 0u;
            py[u] = do_glcm ? (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_y, (int)(i * 2u), 0, 0) : 0u;
        }
        uint32_t s32 = 0, q32 = 0;
        const uint32_t i0 = base + tid;
        uint16_t* const val16 = (uint16_t*)s_val + i0;
#pragma unroll
        for (int u = 0; u < kU; u++) {
            if (!FULL && i0 + u * BS >= n)
                continue;
            if (do_int) {
                if (C16) val16[u * BS] = (uint16_t)(v[u] - vmin);       // C16 launches: every ROI counts, range < 16384
                else if (k16) ((uint16_t*)s_val)[i0 + u * BS] = (uint16_t)(v[u] - vmin);
                else s_val[i0 + u * BS] = v[u];
                // unsigned-int product, wraps (intensity.cpp:90).  v_mul_lo_u32 issues at quarter rate; below 2^24 the 24-bit
                // multiply returns the same low 32 bits at full rate (uniform choice per ROI).
                const uint32_t sq = (TINY || small_v) ? (uint32_t)__umul24(v[u], v[u]) : (uint32_t)(v[u] * v[u]);
                if (TINY) { s32 += v[u]; q32 += sq; }
                else { sum += v[u]; sumsq += sq; }
                if (use_count)
                {
                    const uint32_t ci = v[u] - vmin;
                    if (C16) atomicAdd(cnt16_word(s_cnt, ci), 1u << (16 * (ci & 1u)));   // halves never carry: a count is < 65536
                    else atomicAdd(&s_cnt[ci], 1u);
                }
            }
            if (do_glcm) {
                uint32_t lvl = 0;
                if (NZ || v[u] != 0) { // original-intensity 0 is skipped by the scan (glcm.cpp:445)
                    if (FAST || G16) {  // matlab binning of a non-zero value: floor(slope v + 1) >= 1 already (the conversion truncates: floor of a positive value)
                        const uint32_t sc = (uint32_t)(mslope * (double)v[u] + 1.0);
                        lvl = sc > (uint32_t)greyInfo ? (uint32_t)greyInfo : sc;
                    } else
                    lvl = greyInfo > 0 ? bin_matlab(v[u], mslope, greyInfo)
                        : greyInfo < 0 ? bin_radiomix(v[u], vmin, vmax, -greyInfo) : v[u];
                    if (greyInfo < 0)
                        s_lvlmap[lvl] = 1;
                    if (greyInfo <= 0)                 // only the IBSI / radiomics paths derive the matrix order from the data
                        lvl_max = lvl > lvl_max ? lvl : lvl_max;
                }
                // a cell index inside the plane is all the store needs to be safe (coordinates beyond the box are the caller's
                // contract violation; they cannot leave the plane)
                const uint32_t cell = __umul24(py[u], w) + px[u];
                if (D8)                                    // the plane is padded: cell `area` takes whatever violates the contract
                    *(lds_u8_t*)(cell < area ? cell : area + 64) = (uint8_t)lvl;   // D8: matlab levels <= 64; the plane starts at LDS address 0 (launcher-checked)
                else if (cell < area)
                    s_dense[cell] = (dense_t)(lvl > 0xFFFFu ? 0xFFFFu : lvl);
            }
        }
        if (TINY) { sum += s32; sumsq += q32; }
    };
    bool from_window = false;
    if constexpr (WIN != 0 && !GS) {
    if (WIN == 1 || A.win.inten != nullptr) {
        from_window = true;
        // ---- window mode (fused tile path): the ROI's pixels are read straight from the tile -- the bounding-box window in
        // row-major order, a pixel belongs to the ROI when its label matches -- instead of from a materialised cloud
        // (roi_cloud_kernel wrote 8 B per ROI pixel that this pass then read back).  Every wave takes a contiguous quarter of
        // the window; a label-only pre-pass counts each quarter's members so that a pixel's index in s_val is its rank in
        // window order: exactly the index it has in the cloud, so all that follows is bit-identical to the batch path.
        const uint32_t L = A.win.label[roi], x0 = A.win.x0[roi];
        const uint64_t tile_px = (uint64_t)A.win.H * A.win.W;
        const bool nz_w = vmin > 0;
        // what one member pixel contributes (the body of `trip` above, with its cloud index i and its plane cell)
        auto member = [&](uint32_t v, uint32_t i, uint32_t cell) {
            if (do_int) {
                if (C16 || k16) ((uint16_t*)s_val)[i] = (uint16_t)(v - vmin);
                else s_val[i] = v;
                sum += v;
                sumsq += small_v ? mul24(v, v) : (uint32_t)(v * v);
                if (use_count) {
                    const uint32_t ci = v - vmin;
                    if (C16) atomicAdd(cnt16_word(s_cnt, ci), 1u << (16 * (ci & 1u)));
                    else atomicAdd(&s_cnt[ci], 1u);
                }
            }
            if (do_glcm) {
                uint32_t lvl = 0;
                if (nz_w || v != 0) {
                    if (FAST || G16) {
                        const uint32_t sc = (uint32_t)(mslope * (double)v + 1.0);
                        lvl = sc > (uint32_t)greyInfo ? (uint32_t)greyInfo : sc;
                    } else
                    lvl = greyInfo > 0 ? bin_matlab(v, mslope, greyInfo) : greyInfo < 0 ? bin_radiomix(v, vmin, vmax, -greyInfo) : v;
                    if (greyInfo < 0) s_lvlmap[lvl] = 1;
                    if (greyInfo <= 0) lvl_max = lvl > lvl_max ? lvl : lvl_max;
                }
                if (D8) *(lds_u8_t*)cell = (uint8_t)lvl;
                else s_dense[cell] = (dense_t)(lvl > 0xFFFFu ? 0xFFFFu : lvl);
            }
        };
        if (w <= 64 && tile_px * 4ull < (1ull << 31)) {
            // ---- boxes up to a wave wide: a row of the window per step, lane = column, every wave a contiguous block of rows
            // (window order = row-major order is kept: waves in order, rows in order, lanes in order).  The tile is read through
            // raw buffer descriptors (32-bit lane offsets, the row advance in the scalar offset: no 64-bit address arithmetic),
            // eight rows per wave and trip with every load of the trip issued before the first use, where a row-at-a-time loop left
            // the kernel waiting on ~30 dependent round trips per ROI (3.7 ms per 196 k ROIs against 2.0 ms from pre-assembled clouds).
            constexpr int UW = 4;                                                        // rows per wave and trip
            const int dtl = A.win.dt_label, dti = A.win.dt_inten;
            // The element sizes are launch constants, but a test per load put two scalar branches between any two loads of a trip
            // (1.6 k scalar instructions per wave, the loads trickling out behind taken branches): the loader body is instantiated
            // for the usual pairs (u32 + u32, u16 intensities + u8 / u16 labels) with the sizes as compile-time facts, and once
            // with run-time sizes for the rest.
            auto window_load = [&](auto DTL_c, auto DTI_c) {
            constexpr int DTLc = decltype(DTL_c)::value, DTIc = decltype(DTI_c)::value;   // 0: run-time size
            const int dtl = DTLc ? DTLc : A.win.dt_label, dti = DTIc ? DTIc : A.win.dt_inten;
            const uint32_t Wt = A.win.W;
            const uint64_t tile0 = (uint64_t)A.win.tile[roi] * tile_px;
            const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)A.win.lab + tile0 * (uint64_t)dtl), 0, (int)(tile_px * (uint64_t)dtl), 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)A.win.inten + tile0 * (uint64_t)dti), 0, (int)(tile_px * (uint64_t)dti), 0x00020000);
            auto ldb = [](const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, uint32_t soff, int dt) -> uint32_t {   // element `voff + soff` (bytes) of a tile
                return dt == 4 ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0)
                     : dt == 2 ? (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)voff, (int)soff, 0)
                               : (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)voff, (int)soff, 0);
            };
            // Rows are dealt to the waves round-robin (row r to wave r & 3), eight rows per wave and trip, every load of the trip issued
            // before the first use.  A member's index in s_val is its rank in window order; the ranks of a trip come from ONE
            // exchange of its 32 row populations through LDS (lanes 0 .. 31 pick them up in row order, a wave scan turns them into
            // offsets) -- where a label-only pre-pass over the whole window used to count each wave's block of rows first: a third
            // of the window's bytes, read twice.
            const uint32_t e0 = (A.win.y0[roi] + (uint32_t)wave) * Wt + x0;               // element offset of this wave's first row (scalar)
            const uint32_t vl = (uint32_t)lane * (uint32_t)dtl, vi = (uint32_t)lane * (uint32_t)dti;
            const bool incol = lane < (int)w;
            uint32_t* const s_rc = (uint32_t*)s_red;                                     // [2][32] row populations of a trip, double-buffered
            uint32_t base = 0, trip_i = 0;                                               // members in the rows of earlier trips (wave-uniform)
            for (uint32_t r0 = 0; r0 < h; r0 += NW * UW, trip_i++) {
                uint32_t lb[UW], vv[UW];
                if (incol) {
#pragma unroll
                    for (int u = 0; u < UW; u++)
                        if (r0 + (uint32_t)(u * NW) + (uint32_t)wave < h) {
                            const uint32_t eo = e0 + (r0 + (uint32_t)(u * NW)) * Wt;
                            lb[u] = ldb(rs_l, vl, eo * (uint32_t)dtl, dtl);
                            vv[u] = ldb(rs_i, vi, eo * (uint32_t)dti, dti);
                        }
                }
                unsigned long long bal[UW];
#pragma unroll
                for (int u = 0; u < UW; u++)
                    bal[u] = __ballot(incol && r0 + (uint32_t)(u * NW) + (uint32_t)wave < h && lb[u] == L);
                uint32_t* const rc = s_rc + (trip_i & 1u) * 32u;
                if (lane == 0) {
#pragma unroll
                    for (int u = 0; u < UW; u++) rc[u * NW + wave] = (uint32_t)__popcll(bal[u]);
                }
                grp_sync<GS, NW>();
                const uint32_t inc = wave_scan_u32(lane < NW * UW ? rc[lane & (NW * UW - 1)] : 0u);   // inclusive prefix over the trip's rows in row order
#pragma unroll
                for (int u = 0; u < UW; u++) {
                    const uint32_t upto = (uint32_t)__builtin_amdgcn_readlane((int)inc, u * NW + wave);
                    const uint32_t i = base + upto - (uint32_t)__popcll(bal[u])
                                     + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[u], 0u));
                    if (((bal[u] >> lane) & 1ull) && i < n)
                        member(vv[u], i, (r0 + (uint32_t)(u * NW) + (uint32_t)wave) * w + (uint32_t)lane);
                }
                base += (uint32_t)__builtin_amdgcn_readlane((int)inc, NW * UW - 1);
            }
            grp_sync<GS, NW>();                                                          // (the exchange words are the reduction scratch of what follows)
            };   // window_load
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
            if (dtl == 4 && dti == 4) window_load(I4{}, I4{});
            else if (dtl == 1 && dti == 2) window_load(I1{}, I2{});
            else if (dtl == 2 && dti == 2) window_load(I2{}, I2{});
            else window_load(I0{}, I0{});
        } else {
        const uint64_t row0 = (uint64_t)A.win.tile[roi] * A.win.H + A.win.y0[roi];
        const uint64_t org = row0 * A.win.W + x0;
        const int dtl = A.win.dt_label, dti = A.win.dt_inten;
        // element offsets inside the window stay below 2^32 (a tile has fewer pixels than that): one 64-bit base per ROI,
        // 32-bit offsets per lane, advanced by a wave-uniform stride plus the row wrap
        const char* const lab0 = (const char*)A.win.lab + org * (uint64_t)dtl;
        const char* const int0 = (const char*)A.win.inten + org * (uint64_t)dti;
        auto ld = [](const char* p, uint32_t i, int dt) -> uint32_t {
            return dt == 4 ? ((const uint32_t*)p)[i] : dt == 2 ? (uint32_t)((const uint16_t*)p)[i] : (uint32_t)((const uint8_t*)p)[i];
        };
        const uint32_t Wt = A.win.W;
        const uint32_t cw = ((area + BS - 1) / BS) * 64u;               // pixels per wave (a multiple of 64)
        const uint32_t p_begin = (uint32_t)wave * cw, p_end = p_begin + cw < area ? p_begin + cw : area;
        const uint32_t step_y = 64u / w, step_x = 64u - step_y * w;
        const uint32_t step_o = step_y * Wt + step_x, wrap_o = Wt - w;          // offset advance per 64 pixels, extra when the column wraps
        const uint32_t by0 = (p_begin + (uint32_t)lane) / w, bx0 = (p_begin + (uint32_t)lane) - by0 * w;
        const uint32_t off0 = by0 * Wt + bx0;
        uint32_t hits = 0;
        {
            uint32_t bx = bx0, o32 = off0;
            for (uint32_t p = p_begin; p < p_end; p += 64) {
                const bool in = p + (uint32_t)lane < p_end;
                const uint32_t lb = in ? ld(lab0, o32, dtl) : 0u;
                hits += (uint32_t)__popcll(__ballot(in && lb == L));
                bx += step_x; o32 += step_o;
                if (bx >= w) { bx -= w; o32 += wrap_o; }
            }
        }
        uint32_t* const s_hits = (uint32_t*)(s_stat + 12);                      // (s_stat is free until the sums)
        if (lane == 0) s_hits[wave] = hits;
        grp_sync<GS, NW>();
        uint32_t rank0 = 0;
        for (int wv = 0; wv < wave; wv++) rank0 += s_hits[wv];
        const bool nz = vmin > 0;
        uint32_t bx = bx0, by = by0, o32 = off0;
        for (uint32_t p = p_begin; p < p_end; p += 64) {
            const bool in = p + (uint32_t)lane < p_end;
            const uint32_t lb = in ? ld(lab0, o32, dtl) : 0u;
            const uint32_t v = in ? ld(int0, o32, dti) : 0u;
            const bool hit = in && lb == L;
            const unsigned long long bal = __ballot(hit);
            const uint32_t i = rank0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            rank0 += (uint32_t)__popcll(bal);
            if (hit && i < n) {
                if (do_int) {
                    if (C16 || k16) ((uint16_t*)s_val)[i] = (uint16_t)(v - vmin);
                    else s_val[i] = v;
                    sum += v;
                    sumsq += small_v ? (uint32_t)__umul24(v, v) : (uint32_t)(v * v);
                    if (use_count) {
                        const uint32_t ci = v - vmin;
                        if (C16) atomicAdd(cnt16_word(s_cnt, ci), 1u << (16 * (ci & 1u)));
                        else atomicAdd(&s_cnt[ci], 1u);
                    }
                }
                if (do_glcm) {
                    uint32_t lvl = 0;
                    if (nz || v != 0) {
                        if (FAST || G16) {
                            const uint32_t sc = (uint32_t)(mslope * (double)v + 1.0);
                            lvl = sc > (uint32_t)greyInfo ? (uint32_t)greyInfo : sc;
                        } else
                        lvl = greyInfo > 0 ? bin_matlab(v, mslope, greyInfo) : greyInfo < 0 ? bin_radiomix(v, vmin, vmax, -greyInfo) : v;
                        if (greyInfo < 0) s_lvlmap[lvl] = 1;
                        if (greyInfo <= 0) lvl_max = lvl > lvl_max ? lvl : lvl_max;
                    }
                    s_dense[__umul24(by, w) + bx] = D8 ? (dense_t)lvl : (dense_t)(lvl > 0xFFFFu ? 0xFFFFu : lvl);
                }
            }
            bx += step_x; by += step_y; o32 += step_o;
            if (bx >= w) { bx -= w; by++; o32 += wrap_o; }
        }
        }
    }
    }
    if constexpr (WIN != 1) {
    if (!from_window) {
        using T = std::true_type; using F = std::false_type;
        const bool nz = vmin > 0, tiny = vmax < (1u << 15);
        uint32_t base = 0;
        if (nz && tiny) {
            for (; base + kU * BS <= n; base += kU * BS) trip(T{}, T{}, T{}, base);
            if (base < n) trip(F{}, T{}, T{}, base);
        } else {
            for (; base < n; base += kU * BS) trip(F{}, F{}, F{}, base);
        }
    }
    }

    STAMP(1);
    // From here on the kernel arguments are read through a view of the kernarg segment whose loads the compiler cannot move above this
    // point.  It had merged them into the wide scalar loads of the entry block (max_inten | slide_min | slide_max | out in one
    // s_load_dwordx8, ...) and then SPILLED what the load pass had no scalar register for: 41 v_writelane_b32 and their v_readlane_b32
    // reloads -- vector instructions in a kernel whose limit is vector issue (80 scalar registers per wave at eight waves per SIMD).
    // (the compact and the 64-level builds only: the generic builds, 106 scalar registers at
    //  four to six waves per SIMD, came out with MORE spills under the late view and keep the plain one -- KView)
    const KView<FAM != 0 || G16> B(A);
    double* const out_row_l = B->out + roi * B->ld;
    // =====================================================================================
    // first-order intensity
    // =====================================================================================
    if (do_int) {
        double* o = out_row_l + B->col_intensity;
        const double dn = (double)n;
        // integer sums are exact in any order (the reference's double accumulation is
        // exact too while partial sums stay below 2^53)
        if ((unsigned long long)n * vmax < (1ull << 32))           // the whole ROI's sum fits 32 bits: one-instruction DPP steps
            sum = wave_sum_t<uint32_t>((uint32_t)sum);
        else
            sum = wave_sum_u64(sum);
        if (C16) {   // n < 65536: a lane's sum of squares is below 2^41 -- two 32-bit DPP sums (bits 0..23, bits 24..) instead of a 64-bit one
            const uint32_t lo24 = wave_sum_t<uint32_t>((uint32_t)sumsq & 0xFFFFFFu), hi = wave_sum_t<uint32_t>((uint32_t)(sumsq >> 24));
            sumsq = ((unsigned long long)hi << 24) + lo24;
        } else
            sumsq = wave_sum_u64(sumsq);
        if (lane == 0) {
            s_red[wave * 8 + 0] = (double)sum;
            s_red[wave * 8 + 1] = (double)sumsq;
        }
        grp_sync<GS, NW>(); // also: every s_val / s_cnt write of phase 1 is visible
        const bool blank = (vmin == 0 && vmax == 0); // intensity.cpp:121-122
        if (tid == 0) {                                // the sums' own outputs leave the registers right away
            double tot = 0, totsq = 0;
            for (int wv = 0; wv < NW; wv++) {
                tot += s_red[wv * 8 + 0];
                totsq += s_red[wv * 8 + 1];
            }
            const double mean = tot / dn;              // the one IEEE division; every other thread reads the mean after the barrier
            s_stat[S_MEAN] = mean;
            if (FAST)                                  // GLCM degenerate guard (glcm.cpp:27-95, on GLCM_GREYDEPTH): two binnings, once per ROI
                s_stat[S_NG] = bin_pixel(vmin, vmin, vmax, B->glcm_grey_depth) == bin_pixel(vmax, vmin, vmax, B->glcm_grey_depth) ? 1.0 : 0.0;
            o[I_MIN] = (double)vmin;                   // intensity.cpp:67-69
            o[I_MAX] = (double)vmax;
            o[I_RANGE] = (double)vmax - (double)vmin;
            if (B->slide_min && B->slide_max)            // intensity.cpp:72-77
                o[I_COVERED_IMAGE_INTENSITY_RANGE] = (double)(vmax - vmin) / (B->slide_max[roi] - B->slide_min[roi]);
            o[I_MEAN] = mean;                          // intensity.cpp:95-99
            o[I_ENERGY] = totsq;
            o[I_ROOT_MEAN_SQUARED] = sqrt(totsq / dn);
            o[I_INTEGRATED_INTENSITY] = tot;
            if (!blank)
                o[I_UNIFORMITY_PIU] = (1.0 - (double)(vmax - vmin) / (double)(uint32_t)(vmax + vmin)) * 100.0; // :162
        }
        grp_sync<GS, NW>();
        const double mean = s_stat[S_MEAN];
        STAMP(2);

        const double binW100 = (double)range / 100.;
        const uint32_t nb = (uint32_t)B->n_hist;
        // count of values strictly below the first value whose bin index reaches b; `pred`
        // is the bin index of a value, monotone in the value (histogram.h:55-66, :69-78)
        auto idx100 = [=](uint32_t v) -> int {
            double realIdx = (double)(v - vmin) / binW100;
            return (realIdx != realIdx) ? 0 : (int)realIdx;
        };

        // Q = entries of the counting table scanned per wave (multiple of 512: the 16-bit scan takes eight entries per lane and step)
        const uint32_t Q = ((range + 1 + NW * 512 - 1) / (NW * 512)) * 512;
        uint32_t* const s_woff = (uint32_t*)(s_stat + 8);   // [4] prefix offsets of the waves' table quarters (kept in LDS: as
                                                            // registers they were live across the whole intensity block and spilled)
        if (use_count) {
            // ---- counting engine: per-wave inclusive prefix sums over s_cnt, in place (each
            // lane owns 4 consecutive entries of a 256-entry tile: one 16-byte LDS access each
            // way and one 6-step shuffle scan per tile); the mode (largest count, smallest value
            // on ties: histogram.h:289-309) falls out of the same sweep
            uint32_t carry = 0, best_c = 0, best_i = 0;
            const uint32_t base = wave * Q;
            if (C16) {
                // 16-bit tables: a lane's four entries arrive as two words and stay packed.  No half ever carries into its
                // neighbour (every partial sum is a pixel count < 65536), so plain 32-bit adds work on both halves at once:
                //   A = x + (x << 16) = (c0, c0+c1),  B likewise for (c2, c3),  lane total T = (A + B) >> 16;
                //   the scanned totals are replicated into both halves and added to A and B in one instruction each.
                // Mode: key = count << 16 | (0xFFFF - index) -- the largest key is the largest count and, among equals, the smallest
                // index (an index is < 16384 here).  Eight running maxima, one per entry of the lane, share the lane's base index;
                // the 1 .. 7 they are off by are subtracted once after the loop.
                // Eight entries (four words, one 16-byte access) per lane and step: the six-step wave scan and the loop bookkeeping
                // are paid once per 512 entries.  Word k holds (c_2k, c_2k+1); A_k = w + (w << 16) = (c_2k, c_2k + c_2k+1); the
                // lane total is the high half of A_0 + .. + A_3; output word k = A_k + (everything before it, in both halves),
                // and "everything before word k" is the high half of output word k - 1, broadcast by one v_perm.
                uint32_t kk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                uint32_t inv = 0xFFFFu - (base + 8u * (uint32_t)lane);
                uint4* src = (uint4*)((uint16_t*)s_cnt + base + 8u * (uint32_t)lane);
                uint32_t i = base + 8u * (uint32_t)lane;
                for (uint32_t t = 0; t < Q; t += 512, i += 512, inv -= 512, src += 64) {
                    uint4 pk = make_uint4(0, 0, 0, 0);
                    const bool live = i <= range;
                    if (live) pk = *src;
                    kk[0] = max(kk[0], (pk.x << 16) | inv); kk[1] = max(kk[1], (pk.x & 0xFFFF0000u) | inv);
                    kk[2] = max(kk[2], (pk.y << 16) | inv); kk[3] = max(kk[3], (pk.y & 0xFFFF0000u) | inv);
                    kk[4] = max(kk[4], (pk.z << 16) | inv); kk[5] = max(kk[5], (pk.z & 0xFFFF0000u) | inv);
                    kk[6] = max(kk[6], (pk.w << 16) | inv); kk[7] = max(kk[7], (pk.w & 0xFFFF0000u) | inv);
                    const uint32_t A0 = pk.x + (pk.x << 16), A1 = pk.y + (pk.y << 16), A2 = pk.z + (pk.z << 16), A3 = pk.w + (pk.w << 16);
                    const uint32_t T = ((A0 + A1) + (A2 + A3)) >> 16;
                    const uint32_t sc = wave_scan_u32(T);
                    const uint32_t excl = carry + sc - T;
                    const uint32_t o0 = A0 + (excl | (excl << 16));
                    const uint32_t o1 = A1 + __builtin_amdgcn_perm(o0, o0, 0x03020302u);
                    const uint32_t o2 = A2 + __builtin_amdgcn_perm(o1, o1, 0x03020302u);
                    const uint32_t o3 = A3 + __builtin_amdgcn_perm(o2, o2, 0x03020302u);
                    if (live)
                        *src = make_uint4(o0, o1, o2, o3);
                    carry += readlane63(sc);
                }
#pragma unroll
                for (int q = 1; q < 8; q++) kk[q] -= (uint32_t)q;      // (a key with count 0 never wins: the ROI has a pixel)
                const uint32_t key = wave_max_u32(max(max(max(kk[0], kk[1]), max(kk[2], kk[3])), max(max(kk[4], kk[5]), max(kk[6], kk[7]))));
                best_c = key >> 16;
                best_i = 0xFFFFu - (key & 0xFFFFu);
            } else
            for (uint32_t t = 0; t < Q; t += 256) {
                uint32_t i = base + t + 4 * lane;
                uint4 c4 = make_uint4(0, 0, 0, 0);
                if (C16) {                                    // four 16-bit entries = one 8-byte access (padding is zero)
                    if (i <= range) {
                        const uint2 pk = *(const uint2*)((const uint16_t*)s_cnt + i);
                        c4 = make_uint4(pk.x & 0xFFFFu, pk.x >> 16, pk.y & 0xFFFFu, pk.y >> 16);
                    }
                } else if (i + 3 <= range)
                    c4 = *(const uint4*)&s_cnt[i];
                else {
                    if (i <= range) c4.x = s_cnt[i];
                    if (i + 1 <= range) c4.y = s_cnt[i + 1];
                    if (i + 2 <= range) c4.z = s_cnt[i + 2];
                }
                if (c4.x > best_c) { best_c = c4.x; best_i = i; }
                if (c4.y > best_c) { best_c = c4.y; best_i = i + 1; }
                if (c4.z > best_c) { best_c = c4.z; best_i = i + 2; }
                if (c4.w > best_c) { best_c = c4.w; best_i = i + 3; }
                c4.y += c4.x; c4.z += c4.y; c4.w += c4.z;
                const uint32_t sc = wave_scan_u32(c4.w);
                uint32_t excl = carry + sc - c4.w;
                c4.x += excl; c4.y += excl; c4.z += excl; c4.w += excl;
                if (C16) {
                    if (i <= range)
                        *(uint2*)((uint16_t*)s_cnt + i) = make_uint2(c4.x | (c4.y << 16), c4.z | (c4.w << 16));
                } else if (i + 3 <= range)
                    *(uint4*)&s_cnt[i] = c4;
                else {
                    if (i <= range) s_cnt[i] = c4.x;
                    if (i + 1 <= range) s_cnt[i + 1] = c4.y;
                    if (i + 2 <= range) s_cnt[i + 2] = c4.z;
                }
                carry += readlane63(sc);
            }
            // wave-level best (count desc, index asc)
            if (!C16) {   // (count desc, index asc) as one key: the largest count wins, ties go to the smallest index
                const uint32_t mc_w = wave_max_u32(best_c);
                const uint32_t cand = best_c == mc_w ? best_i : 0xFFFFFFFFu;
                best_i = ~wave_max_u32(~cand);                  // min over the lanes that hold the maximum
                best_c = mc_w;
            }
            if (lane == 0) {
                s_red[wave * 8 + 0] = (double)carry;
                s_red[wave * 8 + 1] = (double)best_c;
                s_red[wave * 8 + 2] = (double)best_i;
            }
            grp_sync<GS, NW>();
            if (tid == 0) {
                uint32_t mc = 0, mi = 0;
                for (int wv = 0; wv < NW; wv++) {
                    uint32_t c = (uint32_t)s_red[wv * 8 + 1], i = (uint32_t)s_red[wv * 8 + 2];
                    if (c > mc) { mc = c; mi = i; } // waves cover ascending value ranges
                }
                const uint32_t woff1 = (uint32_t)s_red[0], woff2 = woff1 + (NW > 1 ? (uint32_t)s_red[8] : 0u), woff3 = woff2 + (NW > 1 ? (uint32_t)s_red[16] : 0u);
                s_woff[0] = 0; s_woff[1] = woff1; s_woff[2] = woff2; s_woff[3] = woff3;
                s_stat[S_MODE] = (double)(vmin + mi);
            }
            grp_sync<GS, NW>();
        } else if (radix) {
            // second key buffer [sort_cap] | digit counts [NW * 256 + NW]
            if (k16) {                                               // (both key buffers in the value region, the counts in B->L.radix)
                uint16_t* const kb = (uint16_t*)s_val + ((B->L.sort_cap + 7u) & ~7u);
                s_val = (uint32_t*)radix_sort<GS, NW, uint16_t>((uint16_t*)s_val, kb, (uint32_t*)reg(B->L.radix), n, 0u, range, tid);
            } else {
                uint32_t* const s_rdx = (uint32_t*)reg(B->L.radix);
                s_val = radix_sort<GS, NW, uint32_t>(s_val, s_rdx, s_rdx + B->L.sort_cap, n, vmin, range, tid);
            }
        } else {
            bitonic_sort<GS, NW>(s_val, P2, tid);
        }
        STAMP(3);
        // sorted value i of the sort engines (16-bit radix keys are offsets from the minimum)
        auto SV = [=](uint32_t i) -> uint32_t { return k16 ? vmin + (uint32_t)((const uint16_t*)s_val)[i] : s_val[i]; };
        // C(i) = number of values <= vmin + i (counting engine)
        auto cum = [=](uint32_t i) -> uint32_t {
            const uint32_t wq = (uint32_t)(i >= Q) + (uint32_t)(i >= 2 * Q) + (uint32_t)(i >= 3 * Q);   // i / Q without the division
            return (C16 ? (uint32_t)((const uint16_t*)s_cnt)[i] : s_cnt[i]) + s_woff[wq];
        };

        // central sums over the LDS-resident values (intensity.cpp:102-109, :177-183; M2..M4 of moments.h:53-74 equal the
        // plain central sums).  C16 launches take them in ONE fused sweep together with the robust statistics, after the
        // percentiles (see below); a blank ROI (all zeros) has every sum equal to zero and needs no sweep at all.
#ifdef NYX_NO_FUSED      // diagnostic builds (A/B timing)
        constexpr bool FUSED = false;
#else
        constexpr bool FUSED = C16 && !GS;
#endif
        double acc[6] = {0, 0, 0, 0, 0, 0};
        auto central_outputs = [&](const double (&acc)[6]) {   // everything that depends only on the sums (single lane)
            // Tolerance-class outputs: the quotients and roots go through reciprocal / reciprocal-square-root estimates with two
            // Newton steps (1-2 ulp) and are shared -- 1/n, 1/(n-1), 1/sqrt(variance), 1/sqrt(M2), 1/sqrt(n) -- instead of ten
            // IEEE divisions and five IEEE roots on one lane (which the whole wave waits for): ~90 instead of ~250 instructions.
            const double var = acc[1];                 // intensity.cpp:110-118
            const double inv_n = frcp(dn);
            o[I_MEAN_ABSOLUTE_DEVIATION] = acc[0] * inv_n;
            const double variance = dn > 1 ? var * frcp(dn - 1) : 0.0;
            const double variance_b = dn > 1 ? var * inv_n : 0.0;
            const double rsd = variance > 0 ? frsq(variance) : 0.0;     // 1 / sd (0 stands for "sd == 0": every use below tests it)
            const double sd = variance * rsd;
            const double rs_n = frsq(dn);
            o[I_VARIANCE] = variance;
            o[I_VARIANCE_BIASED] = variance_b;
            o[I_STANDARD_DEVIATION] = sd;
            o[I_STANDARD_DEVIATION_BIASED] = variance_b > 0 ? variance_b * frsq(variance_b) : 0.0;
            o[I_COV] = sd / mean;                      // (IEEE: a zero mean must give the reference's inf / NaN)
            o[I_STANDARD_ERROR] = sd * rs_n;
            if (!blank) {
                const double M2 = acc[1], M3 = acc[2], M4 = acc[3]; // moments.h:79-109
                if (M2 != 0.0) {
                    const double r = frsq(M2), r2 = r * r;           // 1 / sqrt(M2), 1 / M2
                    const double kurt = n > 4 ? (dn * M4) * (r2 * r2) : 0.0;
                    o[I_SKEWNESS] = n > 3 ? ((dn * rs_n) * M3) * (r2 * r) : 0.0;   // sqrt(n) M3 / pow(M2, 1.5)
                    o[I_KURTOSIS] = kurt;
                    o[I_EXCESS_KURTOSIS] = n > 4 ? kurt - 3 : 0.0;
                }
                // n * pow(sd, 5), n * pow(sd, 6), intensity.cpp:186-191; a zero denominator gives 0
                const double rsd2 = rsd * rsd, t5 = inv_n * (rsd2 * rsd2 * rsd);
                o[I_HYPERSKEWNESS] = acc[4] * t5;
                o[I_HYPERFLATNESS] = acc[5] * (t5 * rsd);
            }
        };
        if (!FUSED || blank) {
            if (!blank) {
                for (uint32_t i = tid; i < n; i += BS) {
                    double d = (double)(C16 ? vmin + ((const uint16_t*)s_val)[i] : SV(i)) - mean;
                    double d2 = d * d;
                    acc[0] += fabs(d);
                    acc[1] += d2;
                    acc[2] += d2 * d;
                    acc[3] += d2 * d2;
                    acc[4] += d2 * d2 * d;
                    acc[5] += d2 * d2 * d2;
                }
                block_sum<6, GS, NW>(acc, s_red, tid);
            }
            if (tid == 0)
                central_outputs(acc);
        }
        STAMP(4);

        if (!blank) {
            // histogram bin boundaries: lower bounds found by binary search, over the value
            // domain (counting engine) or over the sorted array (sort engine)
            for (uint32_t t = tid; t < 100 + nb; t += BS) {
                const bool is100 = t < 100;
                const uint32_t b = is100 ? t : t - 100;
                uint32_t lo;
                if (use_count) {
                    // smallest offset d in [0, range+1] whose bin index reaches b
                    auto bin_of = [=](uint32_t dd) -> uint32_t {
                        return is100 ? (uint32_t)idx100(vmin + dd) : to_grayscale(vmin + dd, vmin, range, nb);
                    };
                    uint32_t d;
                    if (range < 65536u) {
                        // The real-valued boundary is P = b * (bin width).  Unless P lies within 1e-6 of an integer, the answer is
                        // floor(P) + 1 with certainty: the reference's bin function (one or two fp64 roundings of a value below
                        // 2^16) cannot move a value that is >= 1e-6 / range away from the boundary across it.  Next to an integer m
                        // the exact bin function decides between m and m + 1 (m - 1 is a whole bin width / range away): one
                        // evaluation, no search -- and a wave pays it only when one of its lanes is such a boundary.
                        const double Wl = is100 ? binW100 : (double)range / (double)nb;
                        const double P = (double)b * Wl;
                        const uint32_t m = (uint32_t)(P + 0.5);
                        d = (uint32_t)P + 1;
                        if (b == 0)
                            d = 0;
                        else if (fabs(P - (double)m) < 1e-6)
                            d = bin_of(m) >= b ? m : m + 1;
                    } else {
                        // wide ranges: start from the real-valued boundary and settle with the exact (reference) bin function
                        double edge = is100 ? (double)b * binW100 : (double)b * (double)range / (double)nb;
                        d = edge >= (double)range + 1.0 ? range + 1 : (uint32_t)edge;
                        while (d > 0 && bin_of(d - 1) >= b) d--;
                        while (d <= range && bin_of(d) < b) d++;
                    }
                    lo = d > 0 ? cum(d - 1) : 0u;
                } else {
                    uint32_t hi = n;
                    lo = 0;
                    while (lo < hi) {
                        uint32_t mid = (lo + hi) >> 1;
                        uint32_t v = SV(mid);
                        uint32_t idx = is100 ? (uint32_t)idx100(v) : to_grayscale(v, vmin, range, nb);
                        if (idx < b) lo = mid + 1; else hi = mid;
                    }
                }
                if (is100) s_lb100[b] = lo; else s_lbc[b] = lo;
            }
            STAMP(5);
            if (!use_count) {
                // mode on the sorted array: longest run, smallest value on ties; every thread
                // keeps its best run, then a wave / block reduction
                uint32_t best_c = 0, best_v = 0;
                for (uint32_t i = tid; i < n; i += BS) {
                    uint32_t v = SV(i);
                    if (i == n - 1 || SV(i + 1) != v) {
                        uint32_t lo = i, hi = i;
                        if (i > 0 && SV(i - 1) == v) {         // (a run of one -- nearly every run of 16-bit data -- needs no search)
                            lo = 0;
                            while (lo < hi) {
                                uint32_t mid = (lo + hi) >> 1;
                                if (SV(mid) < v) lo = mid + 1; else hi = mid;
                            }
                        }
                        uint32_t c = i - lo + 1;
                        if (c > best_c || (c == best_c && v < best_v)) { best_c = c; best_v = v; }
                    }
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    uint32_t oc = __shfl_down(best_c, d, 64), ov = __shfl_down(best_v, d, 64);
                    if (oc > best_c || (oc == best_c && ov < best_v)) { best_c = oc; best_v = ov; }
                }
                if (lane == 0) {
                    s_red[wave * 8 + 0] = (double)best_c;
                    s_red[wave * 8 + 1] = (double)best_v;
                }
                grp_sync<GS, NW>();
                if (tid == 0) {
                    uint32_t mc = 0, mv = 0;
                    for (int wv = 0; wv < NW; wv++) {
                        uint32_t c = (uint32_t)s_red[wv * 8 + 0], v = (uint32_t)s_red[wv * 8 + 1];
                        if (c > mc || (c == mc && v < mv)) { mc = c; mv = v; }
                    }
                    s_stat[S_MODE] = (double)mv;
                }
            }
            grp_sync<GS, NW>();
            STAMP(6);

            // three independent reductions run on three different waves
            if (NW == 1 || wave == 0) {
                // percentiles P01,P10,P25,P75,P90,P99 (histogram.h:214-243): the LAST bin i with
                // runSum_i <= cnt <= runSum_i + bins_i wins (every matching bin overwrites);
                // runSum_i is exactly the lower bound of bin i.  Lanes test bins i and i+64.
                const int i0 = lane, i1 = lane + 64;
                const uint32_t r0 = s_lb100[i0], e0 = (i0 < 99 ? s_lb100[i0 + 1] : n);
                const uint32_t r1 = i1 < 100 ? s_lb100[i1] : 0u, e1 = i1 < 100 ? (i1 < 99 ? s_lb100[i1 + 1] : n) : 0u;
                // the six winners are found with the whole wave (two ballots each); lane q then interpolates percentile q -- one
                // division per lane instead of six per wave -- and lane 0 collects the values
                int mywin = -1;
                double mycnt = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const double frac = q == 0 ? 0.01 : q == 1 ? 0.1 : q == 2 ? 0.25 : q == 3 ? 0.75 : q == 4 ? 0.9 : 0.99;
                    const double cnt_p = dn * frac;
                    bool m0 = (double)r0 <= cnt_p && cnt_p <= (double)e0;
                    bool m1 = i1 < 100 && (double)r1 <= cnt_p && cnt_p <= (double)e1;
                    unsigned long long b0 = __ballot(m0), b1 = __ballot(m1);
                    int win = b1 ? 64 + (63 - __clzll((long long)b1)) : (b0 ? 63 - __clzll((long long)b0) : -1);
                    if (lane == q) { mywin = win; mycnt = cnt_p; }
                }
                double pv = 0;
                if (mywin >= 0) {
                    uint32_t rs = s_lb100[mywin], bi = (mywin < 99 ? s_lb100[mywin + 1] : n) - rs;
                    pv = (mycnt - (double)rs) * binW100 / (double)bi + (double)vmin + binW100 * (double)mywin;
                }
                double pq[6];
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const unsigned long long u = (unsigned long long)__double_as_longlong(pv);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, q), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), q);
                    pq[q] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                }
                if (lane == 0) {
                    o[I_P01] = pq[0]; o[I_P10] = pq[1]; o[I_P25] = pq[2]; o[I_P75] = pq[3]; o[I_P90] = pq[4]; o[I_P99] = pq[5];
                    o[I_QCOD] = (pq[3] - pq[2]) / (pq[3] + pq[2]);
                    o[I_INTERQUARTILE_RANGE] = pq[3] - pq[2];
                    s_stat[S_P10] = pq[1];
                    s_stat[S_P90] = pq[4];
                    if (FUSED) {
                        // bounds of the robust statistics in the offset domain and the populations around them (see the fused sweep):
                        // derived once, here, instead of by every thread of the workgroup
                        const double p10 = pq[1], p90 = pq[4];
                        uint32_t lox = 0x80000000u, span = 0;                   // empty range unless the bounds say otherwise (NaN: empty)
                        if (p10 <= p90 && p90 >= (double)vmin && p10 <= (double)vmax) {
                            const double cl = ceil(p10), fl = floor(p90);
                            const uint32_t lo_v = cl <= (double)vmin ? vmin : (uint32_t)cl, hi_v = fl >= (double)vmax ? vmax : (uint32_t)fl;
                            if (lo_v <= hi_v) { lox = lo_v - vmin; span = hi_v - lo_v; }
                        }
                        const bool empty = span == 0 && lox == 0x80000000u;
                        uint32_t* const s_rob = (uint32_t*)(s_stat + 6);
                        s_rob[0] = lox;
                        s_rob[1] = span;
                        s_rob[2] = (empty || lox == 0) ? 0u : cum(lox - 1);       // values below the range
                        s_rob[3] = empty ? 0u : cum(lox + span);                  // values up to its upper end
                    }
                }
            }
            if (NW == 1 || wave == 1) {
                // entropy / uniformity over the n+1 slots (histogram.h:145-151): slot n is empty
                double e = 0, u = 0;
                for (uint32_t k = lane; k < nb; k += 64) {
                    uint32_t ck = (k < nb - 1 ? s_lbc[k + 1] : n) - s_lbc[k];
                    double p = fdiv((double)ck, dn);
                    e += p * log2(p + 2.2e-16);
                    u += p * p;
                }
                e = wave_sum(e);
                u = wave_sum(u);
                if (lane == 0) {
                    o[I_ENTROPY] = -e;
                    o[I_UNIFORMITY] = u;
                }
            }
            if (NW == 1 || wave == 2) {
                // median (histogram.h:268-287): order statistics n/2 and n/2-1
                uint32_t hi_v, lo_v;
                if (use_count) {
                    // smallest i with C(i) > k, for k = n/2 and n/2 - 1: a 64-way search -- every lane probes one position of the
                    // current interval, a ballot finds the first hit -- two rounds for ranges up to 4096, three up to 2^18
                    auto kth = [&](uint32_t k) -> uint32_t {
                        uint32_t lo = 0, span = range + 1;                  // the answer lies in [lo, lo + span)
                        while (span > 1) {
                            const uint32_t B = (span + 63) >> 6;
                            uint32_t i = lo + mul24((uint32_t)lane + 1, B) - 1;   // last position of this lane's block
                            if (i > range) i = range;
                            const unsigned long long hit = __ballot(cum(i) > k);
                            const uint32_t first = hit ? (uint32_t)__builtin_ctzll(hit) : 63u;
                            lo += first * B;
                            span = lo + B > range + 1 ? range + 1 - lo : B;
                        }
                        return lo;
                    };
                    hi_v = vmin + kth(n / 2);
                    lo_v = vmin + kth(n / 2 ? n / 2 - 1 : 0);
                } else {
                    hi_v = SV(n / 2);
                    lo_v = SV(n / 2 ? n / 2 - 1 : 0);
                }
                if (lane == 0) {
                    double median = (n & 1) ? (double)hi_v : (double)(uint32_t)(hi_v + lo_v) / 2.0;
                    o[I_MEDIAN] = median;
                    o[I_MODE] = s_stat[S_MODE];
                    s_stat[S_MEDIAN] = median;
                    if (FUSED) {   // 2 (median - vmin) as an integer, and #(x <= floor(median)) for the half-integer correction of the MAD
                        const uint32_t m2x = (uint32_t)((median - (double)vmin) * 2.0);
                        uint32_t* const s_med = (uint32_t*)(s_stat + 10);
                        s_med[0] = m2x;
                        s_med[1] = (m2x & 1u) ? cum(m2x >> 1) : 0u;
                    }
                }
            }
            grp_sync<GS, NW>();
            STAMP(7);

            // robust mean over p10..p90 (intensity.cpp:139-149 == histogram.h:90-101)
            const double p10 = s_stat[S_P10], p90 = s_stat[S_P90], median = s_stat[S_MEDIAN];
            if (FUSED) {
                // ---- one sweep over the resident values: central sums (fp64, fused multiply-adds: the sums are tolerance-class,
                // DESIGN 4.6) + the robust statistics in exact integer arithmetic on the 16-bit offsets x = v - vmin:
                //   a >= p10 && a <= p90  <=>  lox <= x <= hix  with  lox = ceil(p10) - vmin, hix = floor(p90) - vmin  (a is an integer)
                //   sum |a - median| = sum |2x - m2x| / 2,  m2x = 2 (median - vmin)  (an integer: the median is k or k + 1/2)
                // Every partial sum fits 32 bits: n < 65536 and x < 16384 in a C16 launch.
                const uint32_t* const s_rob = (const uint32_t*)(s_stat + 6);     // written by the percentile / median lanes above
                const uint32_t lox = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_rob[0]), span = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_rob[1]);
                const uint32_t hix = lox + span;
                const uint32_t m2x = (uint32_t)__builtin_amdgcn_readfirstlane((int)((const uint32_t*)(s_stat + 10))[0]);
                const uint32_t kmed = m2x >> 1;                 // floor(median) - vmin; the median is kmed or kmed + 1/2
                // The in-range tests cost nothing per value: x is CLAMPED to [lox, hix] (one v_med3_u32) and summed as it is; the
                // values below / above the range contribute lox / hix each, and how many there are is in the cumulative table:
                //   sum over [p10, p90] of x  =  sum of clamp(x)  -  #below * lox  -  #above * hix.
                // Likewise sum |x - median| = sum |x - kmed| (one v_sad_u32, accumulating) + the half-integer correction
                //   (#(x <= kmed) - #(x > kmed)) / 2  when the median is kmed + 1/2.
                uint32_t sx = 0, sad = 0;
                const double meanx = mean - (double)vmin;       // deviations are taken in the offset domain: d = x - (mean - vmin)
                auto px1 = [&](uint32_t x) {
                    const double d = (double)x - meanx;
                    const double d2 = d * d;
                    acc[0] += fabs(d);
                    acc[1] = __builtin_fma(d, d, acc[1]);
                    acc[2] = __builtin_fma(d2, d, acc[2]);
                    acc[3] = __builtin_fma(d2, d2, acc[3]);
                    const double d4 = d2 * d2;
                    acc[4] = __builtin_fma(d4, d, acc[4]);
                    acc[5] = __builtin_fma(d4, d2, acc[5]);
                    sx += med3_u32_ss(x, lox, hix);
                    asm("v_sad_u32 %0, %1, %2, %0" : "+v"(sad) : "v"(x), "s"(kmed));
                };
                // the trip count is wave-uniform (n / 256 full trips, then the lanes below n % 256 once more): the loop control
                // runs on the scalar unit and costs no vector instruction per value
                const uint16_t* const pv = (const uint16_t*)s_val + tid;
                const uint32_t n_full = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n / BS)), n_rem = n - n_full * BS;
                {   // (two values per trip: the second read sits at an immediate offset of the first, one address update per pair)
                    uint32_t k = 0;
                    for (; k + 1 < n_full; k += 2) { px1(pv[k * BS]); px1(pv[(k + 1) * BS]); }
                    if (k < n_full) px1(pv[k * BS]);
                }
                if ((uint32_t)tid < n_rem)
                    px1(pv[n_full * BS]);
                // workgroup totals: the two integer sums through 32-bit DPP wave sums (slots 6, 7 of the exchange area), the six
                // fp64 sums through the transposed wave sum (slots 0..5); one barrier pair for all eight
                {
                    const uint32_t wsx = wave_sum_t<uint32_t>(sx), wsad = wave_sum_t<uint32_t>(sad);
                    if (lane == 0) {
                        s_red[wave * 8 + 6] = (double)wsx;
                        s_red[wave * 8 + 7] = (double)wsad;
                    }
                }
                {
                    double t[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) t[k] = k < 6 ? acc[k] : 0.0;
                    const double tot = wave_transpose_sum8(t, lane);
                    if ((lane & 7) == 0 && (lane >> 3) < 6)
                        s_red[wave * 8 + (lane >> 3)] = tot;
                }
                grp_sync<GS, NW>();
                // every thread needs the in-range sum (sweep 2) and the population of [p10, p90] (read off the cumulative table);
                // the other totals are read by the one lane that derives the outputs -- no barrier follows: the next exchange
                // (sweep 2's) goes through its own scratch (the percentile bounds, dead by now)
                const uint32_t n_below = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_rob[2]);
                const uint32_t n_upto = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_rob[3]);
                const uint32_t K = n_upto - n_below, n_above = n - n_upto;                 // (wave-uniform: the products below run on the scalar unit)
                const uint32_t Sx_all = (uint32_t)xw_pairs<NW>(s_red, 6);
                const uint32_t Sxu = (uint32_t)__builtin_amdgcn_readfirstlane((int)(Sx_all - n_below * lox - n_above * hix));   // < 2^30
                const double Sx = (double)Sxu, dK = (double)K;
                if (tid == 0) {
                    double a6[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        a6[k] = xw_chain<NW>(s_red, k);
                        asm volatile("" : "+v"(a6[k]) :: "memory");   // one total at a time: 28 reads in flight (or their adds sunk into the output code) would spill
                    }
                    // sum |2x - m2x| = 2 sum |x - kmed| + (2 #(x <= kmed) - n  when m2x is odd)
                    const double sadk = xw_pairs<NW>(s_red, 7);
                    const double sadt = 2.0 * sadk + ((m2x & 1u) ? 2.0 * (double)((const uint32_t*)(s_stat + 10))[1] - dn : 0.0);
                    central_outputs(a6);
                    o[I_ROBUST_MEAN] = K ? (Sx + dK * (double)vmin) / dK : 0.0;   // exact integer sum / count, as the reference's
                    o[I_MEDIAN_ABSOLUTE_DEVIATION] = fdiv(sadt * 0.5, dn);
                }
                // sweep 2: robust MAD about mean1090 = S / K (histogram.h:102-112): sum |a - S/K| = sum |K x - Sx| / K, exact in integers.
                // The clamp again: sum over ALL values of |K clamp(x) - Sx|, minus what the values outside contribute
                // (#below |K lox - Sx| + #above |K hix - Sx|) -- three instructions per value.  32-bit lane sums need
                // trips * K * span < 2^32 (|K x - Sx| <= K span inside the range); otherwise the masked 64-bit form.
                const uint32_t Ku = K;
                const bool fast32 = (unsigned long long)(n_full + 1) * Ku * (span + 1ull) < (1ull << 32);
                double ad1[1];
                if (fast32) {
                    uint32_t ad = 0;
                    auto px2 = [&](uint32_t x) {
                        const uint32_t t = mul_u24_su(med3_u32_ss(x, lox, hix), Ku);      // < 2^30
                        asm("v_sad_u32 %0, %1, %2, %0" : "+v"(ad) : "v"(t), "s"(Sxu));
                    };
                    {
                        uint32_t k = 0;
                        for (; k + 1 < n_full; k += 2) { px2(pv[k * BS]); px2(pv[(k + 1) * BS]); }
                        if (k < n_full) px2(pv[k * BS]);
                    }
                    if ((uint32_t)tid < n_rem)
                        px2(pv[n_full * BS]);
                    ad1[0] = (double)ad;
                } else {
                    unsigned long long ad = 0;
                    auto px2 = [&](uint32_t x) {
                        const uint32_t t = (uint32_t)__umul24(Ku, x);            // < 2^30
                        uint32_t dlt;
                        asm("v_sad_u32 %0, %1, %2, 0" : "=v"(dlt) : "v"(t), "s"(Sxu));   // |t - Sx| in one instruction
                        ad += (x - lox) <= span ? dlt : 0u;
                    };
                    for (uint32_t k = 0; k < n_full; k++)
                        px2(pv[k * BS]);
                    if ((uint32_t)tid < n_rem)
                        px2(pv[n_full * BS]);
                    ad1[0] = (double)ad;
                }
                block_sum<1, GS, NW>(ad1, (double*)s_lb100, tid);
                if (tid == 0) {
                    double adin = ad1[0];
                    if (fast32) {                      // (exact: every term is an integer below 2^53)
                        const double klo = (double)Ku * (double)lox, khi = (double)Ku * (double)hix;
                        adin -= (double)n_below * fabs(klo - Sx) + (double)n_above * fabs(khi - Sx);
                    }
                    o[I_ROBUST_MEAN_ABSOLUTE_DEVIATION] = K ? fdiv(fdiv(adin, dK), dK) : 0.0;
                }
            } else {
            // sweep 1: sum and count inside [p10, p90], and the median absolute deviation (it only needs the median)
            double rb[3] = {0, 0, 0};
            for (uint32_t i = tid; i < n; i += BS) {
                double a = (double)(C16 ? vmin + ((const uint16_t*)s_val)[i] : SV(i));
                if (a >= p10 && a <= p90) {
                    rb[0] += a;
                    rb[1] += 1.0;
                }
                rb[2] += fabs(a - median);
            }
            block_sum<3, GS, NW>(rb, s_red, tid);
            const double mean1090 = rb[1] > 0 ? rb[0] / rb[1] : 0.0;
            // sweep 2: robust MAD about that mean (histogram.h:102-112)
            double ad[1] = {0};
            for (uint32_t i = tid; i < n; i += BS) {
                double a = (double)(C16 ? vmin + ((const uint16_t*)s_val)[i] : SV(i));
                if (a >= p10 && a <= p90)
                    ad[0] += fabs(a - mean1090);
            }
            block_sum<1, GS, NW>(ad, s_red, tid);
            if (tid == 0) {
                o[I_ROBUST_MEAN] = mean1090;
                o[I_ROBUST_MEAN_ABSOLUTE_DEVIATION] = rb[1] > 0 ? ad[0] / rb[1] : 0.0;
                o[I_MEDIAN_ABSOLUTE_DEVIATION] = fdiv(rb[2], dn);
            }
            }
        }

        STAMP(8);
    }

    STAMP(9);
    // =====================================================================================
    // GLCM
    // =====================================================================================
    if (FAST) {
        // ---- INTENSITY + GLCM under matlab binning with <= 16 levels, counts exported to glcm_features_kernel: the straight-line
        // form of the general block below (one pass over all angles, matrix order = grey depth, 8-bit plane at LDS address 0,
        // skip-column matrices).  Kept apart because the general block's loop structure and run-time cases cost this build
        // ~60 scalar-register spills per wave (each a vector instruction).
        double* o = out_row_l + B->col_glcm;
        const int na = B->glcm_na;
        const int Ng = greyInfo, NG1 = Ng + 1, NN = Ng * Ng, cells = NG1 * NG1;
        // (the guard of glcm.cpp:27-95 on GLCM_GREYDEPTH: taken beside the mean when the intensity block runs, here when it does not)
        const bool degenerate = FAM == 3 ? bin_pixel(vmin, vmin, vmax, B->glcm_grey_depth) == bin_pixel(vmax, vmin, vmax, B->glcm_grey_depth) : s_stat[S_NG] != 0.0;
        if (tid == 0)
            B->glcm_ng[roi] = degenerate ? 0u : (uint32_t)Ng;
        if (degenerate) {
            for (int c = tid; c < kGlcmAngled * na + kGlcmAve; c += BS)
                o[c] = B->soft_nan;
            return;
        }
        const bool symmetric = B->glcm_symmetric != 0;
        int slot0 = -1, slot1 = -1, slot2 = -1, slot3 = -1;
#pragma unroll
        for (int q = 0; q < kMaxAngles; q++)
            if (q < na) {
                const int ang = B->glcm_angles[q];
                if (ang == 0) slot0 = q; else if (ang == 45) slot1 = q; else if (ang == 90) slot2 = q; else slot3 = q;
            }
        const bool usual = slot0 >= 0 && slot1 >= 0 && slot2 >= 0 && slot3 >= 0 && !symmetric;   // four angles, asymmetric counts
        // lane-per-column sweeps (skip-column matrices): boxes up to a wave wide, and -- for the usual request -- boxes up to
        // two waves wide with two columns per lane
        const bool dpp2 = B->glcm_offset == 1 && w > 64 && w <= 128 && usual;
        const bool dpp = (B->glcm_offset == 1 && w <= 64) || dpp2;
        grp_sync<GS, NW>();
        for (int i = tid; i < na * (dpp ? cells : NN); i += BS)
            s_P[i] = 0;
        grp_sync<GS, NW>();
        STAMP(10);
        if (dpp2) {
            // ---- boxes 65 .. 128 wide: lane = column and column + 64.  What crosses the boundary is wave-uniform (v_readlane):
            // the E / SE neighbours of column 63 are column 64's, the SW neighbour of column 64 is column 63's.  Lanes whose second
            // column lies beyond the box read a zero byte behind the plane (stride 0), like the lanes beyond a narrow box; the row
            // below the last row is tested (the zero row behind the plane is 64 bytes, not 128).  (The per-pixel loop below
            // made the co-occurrence sweep of a 65-wide box three times as expensive as a 63-wide one.)
            const int rows_per_wave = ((int)h + NW - 1) / NW;
            const int r_begin = wave * rows_per_wave;
            const int r_end = (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h;
            const uint32_t ng1 = (uint32_t)NG1;
            char* const T0 = (char*)(s_P + slot0 * cells);
            char* const T1 = (char*)(s_P + slot1 * cells);
            char* const T2 = (char*)(s_P + slot2 * cells);
            char* const T3 = (char*)(s_P + slot3 * cells);
            const bool in1 = (uint32_t)lane + 64u < w;
            uint32_t adr0 = (uint32_t)r_begin * w + (uint32_t)lane;
            uint32_t adr1 = in1 ? adr0 + 64u : area + (uint32_t)lane;
            const uint32_t stride1 = in1 ? w : 0u;
            uint32_t c0 = 0, c1 = 0;
            if (r_begin < r_end) { c0 = (uint32_t)(*(const lds_u8_t*)adr0) << 2; c1 = (uint32_t)(*(const lds_u8_t*)adr1) << 2; }
            auto pairs2 = [&](uint32_t c4, uint32_t e, uint32_t se, uint32_t sth, uint32_t sw) {
                if (c4 != 0) {
                    const uint32_t rowb = mul_u24_su(c4, ng1);
                    atomicAdd((uint32_t*)(T0 + rowb + e), 1u);
                    atomicAdd((uint32_t*)(T1 + rowb + se), 1u);
                    atomicAdd((uint32_t*)(T2 + rowb + sth), 1u);
                    atomicAdd((uint32_t*)(T3 + rowb + sw), 1u);
                }
            };
            for (int row = r_begin; row < r_end; row++) {
                adr0 += w; adr1 += stride1;
                const bool below = row + 1 < (int)h;                                   // (wave-uniform)
                const uint32_t n0 = (uint32_t)(*(const lds_u8_t*)adr0) << 2;           // row h reads the zero row behind the plane
                const uint32_t r1 = (uint32_t)(*(const lds_u8_t*)adr1) << 2;
                const uint32_t n1 = below ? r1 : 0u;
                const uint32_t c1_0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c1), n1_0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n1);
                const uint32_t n0_63 = readlane63(n0);
                pairs2(c0, lane_plus1(c0, c1_0), lane_plus1(n0, n1_0), n0, lane_minus1_z(n0));
                pairs2(c1, lane_plus1_z(c1), lane_plus1_z(n1), n1, lane_minus1(n1, n0_63));
                c0 = n0; c1 = n1;
            }
        } else
        if (dpp) {
            const int rows_per_wave = ((int)h + NW - 1) / NW;
            const int r_begin = wave * rows_per_wave;
            const int r_end = (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h;
            const bool in_col = lane < (int)w;
            const uint32_t ng1 = (uint32_t)NG1;
            char* const T0 = (char*)(s_P + (slot0 >= 0 ? slot0 : 0) * cells);
            char* const T1 = (char*)(s_P + (slot1 >= 0 ? slot1 : 0) * cells);
            char* const T2 = (char*)(s_P + (slot2 >= 0 ? slot2 : 0) * cells);
            char* const T3 = (char*)(s_P + (slot3 >= 0 ? slot3 : 0) * cells);
            // a lane of the box walks its column of the plane (stride w), a lane beyond the box keeps reading one of the 64 zero
            // bytes behind the last row (stride 0), and the row below the last row is that zero row: the read needs no test
            uint32_t adr = in_col ? (uint32_t)r_begin * w + (uint32_t)lane : area + (uint32_t)lane - w;
            const uint32_t stride = in_col ? w : 0u;
            uint32_t cur4 = r_begin < r_end ? (uint32_t)(*(const lds_u8_t*)adr) << 2 : 0u;
            adr += stride;
            if (usual) {
                // the usual request -- four angles, asymmetric: nothing but the four adds per row
                auto pairs = [&](uint32_t c4, uint32_t n4) {
                    const uint32_t nb_e = lane_plus1_z(c4), nb_se = lane_plus1_z(n4), nb_sw = lane_minus1_z(n4);
                    if (c4 != 0) {        // skipped centres (a third of a disk's box) stay out: piled on one cell their adds serialise
                        const uint32_t rowb = mul_u24_su(c4, ng1);
                        atomicAdd((uint32_t*)(T0 + rowb + nb_e), 1u);
                        atomicAdd((uint32_t*)(T1 + rowb + nb_se), 1u);
                        atomicAdd((uint32_t*)(T2 + rowb + n4), 1u);
                        atomicAdd((uint32_t*)(T3 + rowb + nb_sw), 1u);
                    }
                };
                int row = r_begin;
                for (; row + 1 < r_end; row += 2) {   // two rows per trip: the row below becomes the centre row without a move
                    const uint32_t n4a = (uint32_t)(*(const lds_u8_t*)adr) << 2;
                    pairs(cur4, n4a);
                    const uint32_t n4b = (uint32_t)(*(const lds_u8_t*)(adr + stride)) << 2;
                    pairs(n4a, n4b);
                    cur4 = n4b;
                    adr += 2 * stride;
                }
                if (row < r_end)
                    pairs(cur4, (uint32_t)(*(const lds_u8_t*)adr) << 2);
            } else {
                const int has0 = slot0 >= 0, has1 = slot1 >= 0, has2 = slot2 >= 0, has3 = slot3 >= 0;
                for (int row = r_begin; row < r_end; row++, adr += stride) {
                    const uint32_t nxt4 = (uint32_t)(*(const lds_u8_t*)adr) << 2;
                    const uint32_t nb_e = lane_plus1_z(cur4), nb_se = lane_plus1_z(nxt4), nb_sw = lane_minus1_z(nxt4);
                    if (cur4 != 0) {
                        const uint32_t rowb = mul_u24_su(cur4, ng1);
                        if (has0) atomicAdd((uint32_t*)(T0 + rowb + nb_e), 1u);
                        if (has1) atomicAdd((uint32_t*)(T1 + rowb + nb_se), 1u);
                        if (has2) atomicAdd((uint32_t*)(T2 + rowb + nxt4), 1u);
                        if (has3) atomicAdd((uint32_t*)(T3 + rowb + nb_sw), 1u);
                        if (symmetric) {
                            if (has0) atomicAdd((uint32_t*)(T0 + mul_u24_su(nb_e, ng1) + cur4), 1u);
                            if (has1) atomicAdd((uint32_t*)(T1 + mul_u24_su(nb_se, ng1) + cur4), 1u);
                            if (has2) atomicAdd((uint32_t*)(T2 + mul_u24_su(nxt4, ng1) + cur4), 1u);
                            if (has3) atomicAdd((uint32_t*)(T3 + mul_u24_su(nb_sw, ng1) + cur4), 1u);
                        }
                    }
                    cur4 = nxt4;
                }
            }
        } else {
            // any offset / boxes wider than a wave: rows dealt to waves, columns to lanes, plain Ng x Ng matrices (glcm.cpp:431-478)
            for (int row = wave; row < (int)h; row += NW)
                for (int col = lane; col < (int)w; col += 64) {
                    const uint32_t lb = s_dense[(uint32_t)row * w + (uint32_t)col];
                    if (lb == 0)
                        continue;
#pragma unroll
                    for (int q = 0; q < kMaxAngles; q++) {
                        if (q >= na)
                            break;
                        const int ang = B->glcm_angles[q];                 // glcm.cpp:234-255
                        const int dx = ang == 90 ? 0 : ang == 135 ? -B->glcm_offset : B->glcm_offset, dy = ang == 0 ? 0 : B->glcm_offset;
                        const int r2 = row + dy, c2 = col + dx;
                        if (r2 < 0 || r2 >= (int)h || c2 < 0 || c2 >= (int)w)
                            continue;
                        const uint32_t la = s_dense[(uint32_t)r2 * w + (uint32_t)c2];
                        if (la == 0)
                            continue;
                        atomicAdd(&s_P[q * NN + ((int)lb - 1) * Ng + (int)la - 1], 1u);
                        if (symmetric)
                            atomicAdd(&s_P[q * NN + ((int)la - 1) * Ng + (int)lb - 1], 1u);
                    }
                }
        }
        grp_sync<GS, NW>();
        STAMP(11);
        {
            uint32_t* dst = B->glcm_ws + roi * B->glcm_ws_stride;
            if (dpp) {
                // dense cell i = (q, r, c) sits at q * (Ng+1)^2 + (r+1) * (Ng+1) + c + 1.  The two small divisions go through
                // float: (i + 1/2) / d is never closer than 1/(2d) to an integer, far beyond the error of the reciprocal.
                const float inv_nn = __builtin_amdgcn_rcpf((float)NN), inv_ng = __builtin_amdgcn_rcpf((float)Ng);
                for (int i = tid; i < na * NN; i += BS) {
                    const uint32_t q = (uint32_t)(((float)i + 0.5f) * inv_nn), rem = (uint32_t)i - mul24(q, (uint32_t)NN);
                    const uint32_t r = (uint32_t)(((float)rem + 0.5f) * inv_ng), c = rem - mul24(r, (uint32_t)Ng);
                    dst[i] = s_P[mad24(q, (uint32_t)cells, mad24(r + 1u, (uint32_t)NG1, c + 1u))];
                }
            } else
                for (int i = tid; i < na * NN; i += BS)
                    dst[i] = s_P[i];
        }
        STAMP(12);
    } else
    if (do_glcm) {
        double* o = out_row_l + B->col_glcm;
        const int na = B->glcm_na;
        const int ncol_g = kGlcmAngled * na + kGlcmAve;
        // degenerate guard (glcm.cpp:27-95) uses GLCM_GREYDEPTH
        // (the extrema pass through readfirstlane so that their conversions to double are redone here instead of being kept
        // -- spilled -- since the load phase)
        const uint32_t vmin_g = (uint32_t)__builtin_amdgcn_readfirstlane((int)vmin), vmax_g = (uint32_t)__builtin_amdgcn_readfirstlane((int)vmax);
        const bool degenerate = bin_pixel(vmin_g, vmin_g, vmax_g, B->glcm_grey_depth) == bin_pixel(vmax_g, vmin_g, vmax_g, B->glcm_grey_depth);

        // matrix order and level values (glcm.cpp:388-420)
        double* s_I = s_g;                       // [ng_cap] level values
        double* s_f = s_g + B->L.ng_cap;          // [kMaxAngles][32] per-angle features
        double* s_scr = s_f + kMaxAngles * 32;   // [kMaxAngles][6*ng_cap]
        if (FAST) {
            // matlab binning: the matrix order is the grey depth itself -- nothing to agree on, no barrier
        } else if (greyInfo > 0) {
            if (tid == 0)
                s_stat[S_NG] = (double)greyInfo;
        } else {
            lvl_max = wave_max_u32(lvl_max);
            if (lane == 0)
                s_red[wave * 8] = (double)lvl_max;
            grp_sync<GS, NW>(); // also orders the s_lvlmap writes of phase 1
            if (tid == 0) {
                int Ng;
                if (greyInfo == 0) {
                    double m = 0;
                    for (int wv = 0; wv < NW; wv++)
                        m = s_red[wv * 8] > m ? s_red[wv * 8] : m;
                    Ng = (int)m;
                } else {
                    // unique sorted non-zero levels -> compact indices (glcm.cpp:391-397)
                    int k = 0;
                    for (uint32_t l = 1; l <= B->L.lvl_cap; l++)
                        if (s_lvlmap[l]) {
                            s_lvlmap[l] = (uint16_t)(k + 1);
                            if ((uint32_t)k < B->L.ng_cap)
                                s_I[k] = (double)l;
                            k++;
                        }
                    Ng = k;
                }
                s_stat[S_NG] = (double)Ng;
            }
        }
        if (!FAST) grp_sync<GS, NW>();
        const int Ng = FAST ? greyInfo : (int)s_stat[S_NG];
        const bool too_big = !FAST && (uint32_t)Ng > B->L.ng_cap;      // (FAST: make_layout reserved exactly this order)
        if (too_big && tid == 0)
            atomicCAS(B->status, 0, NYXHIP_ERR_UNSUPPORTED);
        if (!SPLIT && !G16 && greyInfo >= 0 && !too_big)   // (G16: level values are i + 1 by construction, and its scratch lies over the live plane)
            for (int i = tid; i < Ng; i += BS)
                s_I[i] = (double)(i + 1);

        const bool split = SPLIT && !degenerate && !too_big;
        if (SPLIT && tid == 0)
            B->glcm_ng[roi] = split ? (uint32_t)Ng : 0u;
        if (degenerate) {
            for (int c = tid; c < ncol_g; c += BS)
                o[c] = B->soft_nan;
        } else if (too_big) {
            for (int c = tid; c < ncol_g; c += BS)
                o[c] = __longlong_as_double(0x7ff8000000000000LL);
        } else if (G16) {
            // ---- the reference's default grey depth (17..64 matlab levels) --------------------------------------------------------
            // Four 64 x 64 u32 matrices are 64 KiB of LDS: one workgroup per CU, four waves to hide every latency.  A cell count is
            // below 65536 here (the launch's ROIs have < 32768 pixels), so the matrices hold 16-bit cells, two per word, added to
            // with a shifted increment (halves never carry); order Ng + 1, indexed by the level itself -- column 0 takes the pairs
            // with a skipped neighbour, as in the split launches.  With the marginal-based feature routine (no 25 Ng doubles of
            // scratch per angle) the carve-out drops from ~87 to ~43 KiB: three workgroups per CU.
            // Round 6 layout (glcm_features_wave64_v2): rows 0..Ng of an even pitch, cell (centre a, neighbour b) at a * pitch + b - 1:
            // data cells are word-aligned pairs, a skipped neighbour lands in the previous row's last column, column Ng stays zero.
            const int NG1 = (Ng + 3) & ~1, cellsw = ((Ng + 1) * NG1) >> 1;    // NG1: the pitch; words per matrix
            const bool symmetric = B->glcm_symmetric != 0;
            uint32_t* const s_blk = (uint32_t*)(s_f);                 // [kMaxAngles][256 words]: per-wave scratch of the feature pass; a wave's f = its first 32 doubles
            grp_sync<GS, NW>();
            {   // 16 bytes per store (word stores were 34 trips of four instructions: 1.1 ns of every ROI); the matrices' region is 16-byte aligned and 16 cellsw bytes long
                uint4* const P4 = (uint4*)s_P;
                const int n4 = (na * cellsw + 3) >> 2;
                for (int i = tid; i < n4; i += BS)
                    P4[i] = uint4{0u, 0u, 0u, 0u};
            }
            grp_sync<GS, NW>();
            STAMP(10);
            auto bump16 = [&](uint32_t* M, uint32_t idx) { idx -= 1u; atomicAdd(&M[idx >> 1], 1u << ((idx & 1u) << 4)); };
            if (B->glcm_offset == 1 && w > 64 && w <= 128) {
                // boxes 65 .. 128 wide: lane = column and column + 64, the pairs across column 63 | 64 through v_readlane (see the
                // <= 16-level block above); any angle subset, symmetric counts included
                int slot0 = -1, slot1 = -1, slot2 = -1, slot3 = -1;
#pragma unroll
                for (int q = 0; q < kMaxAngles; q++)
                    if (q < na) {
                        const int ang = B->glcm_angles[q];
                        if (ang == 0) slot0 = q; else if (ang == 45) slot1 = q; else if (ang == 90) slot2 = q; else slot3 = q;
                    }
                const int rows_per_wave = ((int)h + NW - 1) / NW;
                const int r_begin = wave * rows_per_wave;
                const int r_end = (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h;
                const bool in1 = (uint32_t)lane + 64u < w;
                uint32_t adr0 = (uint32_t)r_begin * w + (uint32_t)lane;
                uint32_t adr1 = in1 ? adr0 + 64u : area + (uint32_t)lane;
                const uint32_t stride1 = in1 ? w : 0u;
                const uint32_t ng1 = (uint32_t)NG1;
                uint32_t c0 = 0, c1 = 0;
                if (r_begin < r_end) { c0 = (uint32_t)(*(const lds_u8_t*)adr0); c1 = (uint32_t)(*(const lds_u8_t*)adr1); }
                auto pairs2 = [&](uint32_t cur, uint32_t e, uint32_t se, uint32_t sth, uint32_t sw) {
                    if (cur != 0) {
                        const uint32_t rowi = mul24(cur, ng1);
                        if (slot0 >= 0) bump16(s_P + slot0 * cellsw, rowi + e);
                        if (slot1 >= 0) bump16(s_P + slot1 * cellsw, rowi + se);
                        if (slot2 >= 0) bump16(s_P + slot2 * cellsw, rowi + sth);
                        if (slot3 >= 0) bump16(s_P + slot3 * cellsw, rowi + sw);
                        if (symmetric) {
                            if (slot0 >= 0) bump16(s_P + slot0 * cellsw, mad24(e, ng1, cur));
                            if (slot1 >= 0) bump16(s_P + slot1 * cellsw, mad24(se, ng1, cur));
                            if (slot2 >= 0) bump16(s_P + slot2 * cellsw, mad24(sth, ng1, cur));
                            if (slot3 >= 0) bump16(s_P + slot3 * cellsw, mad24(sw, ng1, cur));
                        }
                    }
                };
                for (int row = r_begin; row < r_end; row++) {
                    adr0 += w; adr1 += stride1;
                    const bool below = row + 1 < (int)h;
                    const uint32_t n0 = (uint32_t)(*(const lds_u8_t*)adr0);            // row h reads the zero row behind the plane
                    const uint32_t r1 = (uint32_t)(*(const lds_u8_t*)adr1);
                    const uint32_t n1 = below ? r1 : 0u;                                // (that zero row is 64 bytes, not 128)
                    const uint32_t c1_0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c1), n1_0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n1);
                    const uint32_t n0_63 = readlane63(n0);
                    pairs2(c0, lane_plus1(c0, c1_0), lane_plus1(n0, n1_0), n0, lane_minus1_z(n0));
                    pairs2(c1, lane_plus1_z(c1), lane_plus1_z(n1), n1, lane_minus1(n1, n0_63));
                    c0 = n0; c1 = n1;
                }
            } else
            if (B->glcm_offset == 1 && w <= 64) {
                int slot0 = -1, slot1 = -1, slot2 = -1, slot3 = -1;
#pragma unroll
                for (int q = 0; q < kMaxAngles; q++)
                    if (q < na) {
                        const int ang = B->glcm_angles[q];
                        if (ang == 0) slot0 = q; else if (ang == 45) slot1 = q; else if (ang == 90) slot2 = q; else slot3 = q;
                    }
                const int rows_per_wave = ((int)h + NW - 1) / NW;
                const int r_begin = wave * rows_per_wave;
                const int r_end = (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h;
                const bool in_col = lane < (int)w;
                // (8-bit plane at LDS address 0 with 64 zero bytes behind its last row: a lane of the box walks its column, a lane
                //  beyond the box keeps reading a zero byte, the row below the last row is the zero row -- no test per read;
                //  24-bit products for the cell indices)
                uint32_t adr = in_col ? (uint32_t)r_begin * w + (uint32_t)lane : area + (uint32_t)lane - w;
                const uint32_t stride = in_col ? w : 0u;
                uint32_t cur = r_begin < r_end ? (uint32_t)(*(const lds_u8_t*)adr) : 0u;
                const uint32_t ng1 = (uint32_t)NG1;
                if (slot0 >= 0 && slot1 >= 0 && slot2 >= 0 && slot3 >= 0 && !symmetric) {
                    // the usual request -- four angles, asymmetric -- straight-line: levels travel doubled (the byte offset of a 16-bit
                    // cell in its row), the matrix bases are folded into the centre's row offset per direction, and a pair is
                    // add (cell's byte offset) / and (its word) / two shifts (1 << 16 * odd cell) / ds_add: four vector
                    // instructions instead of six, no scalar branch per pair
                    typedef __attribute__((address_space(3))) uint32_t lds_word_t;
                    lds_word_t* const Pw = (lds_word_t*)s_P;
                    const uint32_t B0 = (uint32_t)(uintptr_t)(Pw + slot0 * cellsw) - 2u, B1 = (uint32_t)(uintptr_t)(Pw + slot1 * cellsw) - 2u,     // (- 2: neighbour level b sits in column b - 1)
                                   B2 = (uint32_t)(uintptr_t)(Pw + slot2 * cellsw) - 2u, B3 = (uint32_t)(uintptr_t)(Pw + slot3 * cellsw) - 2u;
                    auto bump2 = [&](uint32_t a2) {        // a2: LDS byte address of the 16-bit cell
                        (void)__hip_atomic_fetch_add((lds_word_t*)(uintptr_t)(a2 & ~3u), 1u << ((a2 << 3) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    };
                    uint32_t cur2 = cur << 1;
                    const uint32_t ng1x = ng1;              // (row offset in bytes = 2 * level * NG1 = cur2 * NG1)
                    for (int row = r_begin; row < r_end; row++) {
                        adr += stride;
                        const uint32_t nxt2 = (uint32_t)(*(const lds_u8_t*)adr) << 1;
                        const uint32_t e2 = lane_plus1_z(cur2), se2 = lane_plus1_z(nxt2), sw2 = lane_minus1_z(nxt2);
                        if (cur2 != 0) {
                            const uint32_t rowb = mul24(cur2, ng1x);
                            bump2(B0 + rowb + e2);
                            bump2(B1 + rowb + se2);
                            bump2(B2 + rowb + nxt2);
                            bump2(B3 + rowb + sw2);
                        }
                        cur2 = nxt2;
                    }
                } else
                for (int row = r_begin; row < r_end; row++) {
                    adr += stride;
                    const uint32_t nxt = (uint32_t)(*(const lds_u8_t*)adr);
                    const uint32_t nb_e = lane_plus1_z(cur), nb_se = lane_plus1_z(nxt), nb_sw = lane_minus1_z(nxt);
                    if (cur != 0) {
                        const uint32_t rowi = mul24(cur, ng1);
                        if (slot0 >= 0) bump16(s_P + slot0 * cellsw, rowi + nb_e);
                        if (slot1 >= 0) bump16(s_P + slot1 * cellsw, rowi + nb_se);
                        if (slot2 >= 0) bump16(s_P + slot2 * cellsw, rowi + nxt);
                        if (slot3 >= 0) bump16(s_P + slot3 * cellsw, rowi + nb_sw);
                        if (symmetric) {
                            if (slot0 >= 0) bump16(s_P + slot0 * cellsw, mad24(nb_e, ng1, cur));
                            if (slot1 >= 0) bump16(s_P + slot1 * cellsw, mad24(nb_se, ng1, cur));
                            if (slot2 >= 0) bump16(s_P + slot2 * cellsw, mad24(nxt, ng1, cur));
                            if (slot3 >= 0) bump16(s_P + slot3 * cellsw, mad24(nb_sw, ng1, cur));
                        }
                    }
                    cur = nxt;
                }
            } else if (B->glcm_offset == 1) {
                // boxes wider than two waves: column strips of 62 centre columns between two halo lanes (see the general block below)
                int slot0 = -1, slot1 = -1, slot2 = -1, slot3 = -1;
#pragma unroll
                for (int q = 0; q < kMaxAngles; q++)
                    if (q < na) {
                        const int ang = B->glcm_angles[q];
                        if (ang == 0) slot0 = q; else if (ang == 45) slot1 = q; else if (ang == 90) slot2 = q; else slot3 = q;
                    }
                const int rows_per_wave = ((int)h + NW - 1) / NW;
                const int r_begin = wave * rows_per_wave;
                const int r_end = (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h;
                const uint32_t ng1 = (uint32_t)NG1;
                for (uint32_t c0 = 0; c0 < w; c0 += 62) {
                    const int c = (int)c0 - 1 + lane;
                    const bool in_col = c >= 0 && c < (int)w;
                    const bool centre = in_col && lane >= 1 && lane <= 62;
                    auto lvl = [&](int row) -> uint32_t { return (in_col && row < (int)h) ? (uint32_t)s_dense[(uint32_t)row * w + (uint32_t)c] : 0u; };
                    uint32_t cur = r_begin < r_end ? lvl(r_begin) : 0u;
                    for (int row = r_begin; row < r_end; row++) {
                        const uint32_t nxt = lvl(row + 1);
                        const uint32_t nb_e = lane_plus1(cur, 0), nb_se = lane_plus1(nxt, 0), nb_sw = lane_minus1(nxt, 0);
                        if (centre && cur != 0) {
                            const uint32_t rowi = mul24(cur, ng1);
                            if (slot0 >= 0) bump16(s_P + slot0 * cellsw, rowi + nb_e);
                            if (slot1 >= 0) bump16(s_P + slot1 * cellsw, rowi + nb_se);
                            if (slot2 >= 0) bump16(s_P + slot2 * cellsw, rowi + nxt);
                            if (slot3 >= 0) bump16(s_P + slot3 * cellsw, rowi + nb_sw);
                            if (symmetric) {
                                if (slot0 >= 0) bump16(s_P + slot0 * cellsw, mad24(nb_e, ng1, cur));
                                if (slot1 >= 0) bump16(s_P + slot1 * cellsw, mad24(nb_se, ng1, cur));
                                if (slot2 >= 0) bump16(s_P + slot2 * cellsw, mad24(nxt, ng1, cur));
                                if (slot3 >= 0) bump16(s_P + slot3 * cellsw, mad24(nb_sw, ng1, cur));
                            }
                        }
                        cur = nxt;
                    }
                }
            } else {
                for (int row = wave; row < (int)h; row += NW)
                    for (int col = lane; col < (int)w; col += 64) {
                        const uint32_t lb = s_dense[(uint32_t)row * w + (uint32_t)col];
                        if (lb == 0)
                            continue;
#pragma unroll
                        for (int q = 0; q < kMaxAngles; q++) {
                            if (q >= na)
                                break;
                            const int ang = B->glcm_angles[q];                 // glcm.cpp:234-255
                            const int dx = ang == 90 ? 0 : ang == 135 ? -B->glcm_offset : B->glcm_offset, dy = ang == 0 ? 0 : B->glcm_offset;
                            const int r2 = row + dy, c2 = col + dx;
                            if (r2 < 0 || r2 >= (int)h || c2 < 0 || c2 >= (int)w)
                                continue;
                            const uint32_t la = s_dense[(uint32_t)r2 * w + (uint32_t)c2];
                            if (la == 0)
                                continue;
                            bump16(s_P + q * cellsw, lb * (uint32_t)NG1 + la);
                            if (symmetric)
                                bump16(s_P + q * cellsw, la * (uint32_t)NG1 + lb);
                        }
                    }
            }
            grp_sync<GS, NW>();
            STAMP(11);
            if constexpr (NW == 8) {                     // two waves per angle: wave a and wave a + 4 (glcm_features_wave64_v2, ROLE)
                const int ang = wave & 3;
                if (ang < na) {
                    const uint16_t* const Pa = (const uint16_t*)(s_P + (size_t)ang * cellsw);
                    if (wave < 4) {
                        if (Ng == 64) glcm_features_wave64_v2<64, 1>(Pa, 64, s_blk + ang * 256, B->soft_nan, lane);
                        else glcm_features_wave64_v2<0, 1>(Pa, Ng, s_blk + ang * 256, B->soft_nan, lane);
                    } else {
                        if (Ng == 64) glcm_features_wave64_v2<64, 2>(Pa, 64, s_blk + ang * 256, B->soft_nan, lane);
                        else glcm_features_wave64_v2<0, 2>(Pa, Ng, s_blk + ang * 256, B->soft_nan, lane);
                    }
                } else { grp_sync<GS, NW>(); grp_sync<GS, NW>(); }
            } else
            if (wave < na) {                             // (NW = 4 = kMaxAngles: a wave per angle)
                if (Ng == 64) glcm_features_wave64_v2<64>((const uint16_t*)(s_P + (size_t)wave * cellsw), 64, s_blk + wave * 256, B->soft_nan, lane);
                else glcm_features_wave64_v2<0>((const uint16_t*)(s_P + (size_t)wave * cellsw), Ng, s_blk + wave * 256, B->soft_nan, lane);
            }
            grp_sync<GS, NW>();
            if (tid < na) glcm_features_final(s_blk + tid * 256, B->soft_nan);
            grp_sync<GS, NW>();
            STAMP(12);
            for (int c = tid; c < kGlcmAngled * na; c += BS) {
                int k = c / na, a = c - k * na;
                o[c] = s_f[a * 128 + k];
            }
            for (int j = tid; j < kGlcmAve; j += BS) {               // calc_ave (glcm.cpp:1205-1214): std::reduce folds four at a time
                int k = c_glcm_ave_order[j];
                double init = 0.0;
                int a = 0;
                for (; na - a >= 4; a += 4) {
                    double v1 = s_f[a * 128 + k] + s_f[(a + 1) * 128 + k];
                    double v2 = s_f[(a + 2) * 128 + k] + s_f[(a + 3) * 128 + k];
                    init = init + (v1 + v2);
                }
                for (; a < na; a++)
                    init = init + s_f[a * 128 + k];
                o[kGlcmAngled * na + j] = na ? init / (double)na : 0.0;
            }
        } else {
            const int NN = Ng * Ng;
            const bool symmetric = B->glcm_symmetric || greyInfo <= 0; // glcm.cpp:475
            const int app = FAST ? kMaxAngles : (int)B->L.app;         // (FAST launches hold every angle in one pass: build_args)
            for (int a0 = 0; a0 < na; a0 += app) {
                const int na_pass = (na - a0) < app ? (na - a0) : app;
                // Split launches on the lane-per-column path count into matrices of order Ng + 1 indexed by the level itself:
                // column 0 (row 0 under symmetry) collects the pairs whose NEIGHBOUR is skipped (level 0: background, zero
                // intensity, outside the box) -- the ROI's rim, spread over the centres' levels -- so the sweep has one test per
                // centre instead of one per pair; the export below drops the extra row / column.  (Skipped CENTRES are masked
                // out: sent to a cell of their own, a third of the box's lanes would pile on it and the LDS serialises same-
                // address adds -- measured: 2.3 -> 3.4 ms per 196 k ROIs.)
                const bool dpp = B->glcm_offset == 1 && w <= 64;
                const bool trash = SPLIT && dpp;
                const int NG1 = Ng + 1, cells = trash ? NG1 * NG1 : NN;
                grp_sync<GS, NW>();
                for (int i = tid; i < na_pass * cells; i += BS)
                    s_P[i] = 0;
                grp_sync<GS, NW>();
                STAMP(10);
                // co-occurrence scan (glcm.cpp:431-478): LDS atomics, all angles of the pass.
                // Rows are dealt to waves, columns to lanes (no integer division per pixel).
                int ddx[kMaxAngles], ddy[kMaxAngles];
#pragma unroll
                for (int q = 0; q < kMaxAngles; q++) {
                    int ang = B->glcm_angles[(a0 + q) < na ? (a0 + q) : 0]; // glcm.cpp:234-255
                    ddx[q] = ang == 90 ? 0 : ang == 135 ? -B->glcm_offset : B->glcm_offset;
                    ddy[q] = ang == 0 ? 0 : B->glcm_offset;
                }
                if (dpp) {
                    // one lane per column: horizontal neighbours come from DPP lane shifts, the
                    // row below is read once and becomes the next iteration's centre row; each
                    // wave owns a contiguous block of rows.  slot[d] = matrix of this pass that
                    // direction d (E, SE, S, SW = 0, 45, 90, 135 degrees) accumulates into, or -1.
                    int slot0 = -1, slot1 = -1, slot2 = -1, slot3 = -1;
#pragma unroll
                    for (int q = 0; q < kMaxAngles; q++)
                        if (q < na_pass) {
                            int ang = B->glcm_angles[a0 + q];
                            if (ang == 0) slot0 = q; else if (ang == 45) slot1 = q; else if (ang == 90) slot2 = q; else slot3 = q;
                        }
                    const int rows_per_wave = ((int)h + NW - 1) / NW;
                    const int r_begin = wave * rows_per_wave;
                    const int r_end = (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h;
                    const bool in_col = lane < (int)w;
                    const bool remap = greyInfo < 0;
                    // Levels travel through the loop as byte offsets (4 * level, 0 = skip): a cell address is then
                    // matrix base + row offset + neighbour offset -- one three-operand add per pair.
                    uint32_t cur4 = (in_col && r_begin < r_end) ? s_dense[(uint32_t)r_begin * w + lane] : 0u;
                    if (remap && cur4) cur4 = s_lvlmap[cur4];
                    cur4 <<= 2;
                    // one (centre, neighbour) pair into the matrix of its direction.  The centre's row offset is shared by the
                    // four directions; lanes beyond the last column hold level 0, so the right-hand neighbours of the last column
                    // are "skip" without a separate test.  B_q = matrix q minus one element (levels are 1-based).
                    if (trash) {
                        // branch-free sweep: address = matrix + 4 * (centre * (Ng + 1) + neighbour), every lane, every pair
                        // (matrix offsets and the "angle present" flags pass through readfirstlane: kept in SGPRs, the four tests
                        // of a row are scalar branches and cost no vector instruction)
                        const uint32_t ng1 = (uint32_t)__builtin_amdgcn_readfirstlane(NG1);
                        const int has0 = __builtin_amdgcn_readfirstlane(slot0 >= 0), has1 = __builtin_amdgcn_readfirstlane(slot1 >= 0),
                                  has2 = __builtin_amdgcn_readfirstlane(slot2 >= 0), has3 = __builtin_amdgcn_readfirstlane(slot3 >= 0),
                                  sym = __builtin_amdgcn_readfirstlane(symmetric ? 1 : 0);
                        char* const T0 = (char*)(s_P + (slot0 >= 0 ? slot0 : 0) * cells);
                        char* const T1 = (char*)(s_P + (slot1 >= 0 ? slot1 : 0) * cells);
                        char* const T2 = (char*)(s_P + (slot2 >= 0 ? slot2 : 0) * cells);
                        char* const T3 = (char*)(s_P + (slot3 >= 0 ? slot3 : 0) * cells);
                        if (has0 && has1 && has2 && has3 && !sym) {
                            // the usual request -- four angles, asymmetric: nothing but the four adds per row
                            auto pairs = [&](uint32_t c4, uint32_t n4) {
                                const uint32_t nb_e = lane_plus1_z(c4), nb_se = lane_plus1_z(n4), nb_sw = lane_minus1_z(n4);
                                if (c4 != 0) {        // skipped centres (a third of a disk's box) stay out: piled on one cell their adds serialise
                                    const uint32_t rowb = mul_u24_su(c4, ng1);
                                    atomicAdd((uint32_t*)(T0 + rowb + nb_e), 1u);
                                    atomicAdd((uint32_t*)(T1 + rowb + nb_se), 1u);
                                    atomicAdd((uint32_t*)(T2 + rowb + n4), 1u);
                                    atomicAdd((uint32_t*)(T3 + rowb + nb_sw), 1u);
                                }
                            };
                            if (D8) {
                                // 8-bit plane at LDS address 0 with 64 zero bytes behind its last row: a lane of the box walks its column
                                // (stride w), a lane beyond the box keeps reading one of the zero bytes (stride 0), and the row below
                                // the last row is the zero row -- the read needs no test at all
                                uint32_t adr = in_col ? (uint32_t)(r_begin + 1) * w + (uint32_t)lane : area + (uint32_t)lane - w;
                                const uint32_t stride = in_col ? w : 0u;
                                if (r_begin >= r_end) cur4 = 0;
                                int row = r_begin;
                                for (; row + 1 < r_end; row += 2) {   // two rows per trip: the row below becomes the centre row without a move
                                    const uint32_t n4a = (uint32_t)(*(const lds_u8_t*)adr) << 2;
                                    pairs(cur4, n4a);
                                    const uint32_t n4b = (uint32_t)(*(const lds_u8_t*)(adr + stride)) << 2;
                                    pairs(n4a, n4b);
                                    cur4 = n4b;
                                    adr += 2 * stride;
                                }
                                if (row < r_end)
                                    pairs(cur4, (uint32_t)(*(const lds_u8_t*)adr) << 2);
                            } else
                            for (int row = r_begin; row < r_end; row++) {
                                uint32_t nxt4 = (in_col && row + 1 < (int)h) ? s_dense[(uint32_t)(row + 1) * w + lane] : 0u;
                                nxt4 <<= 2;
                                pairs(cur4, nxt4);
                                cur4 = nxt4;
                            }
                        } else
                        for (int row = r_begin; row < r_end; row++) {
                            uint32_t nxt4 = (in_col && row + 1 < (int)h) ? s_dense[(uint32_t)(row + 1) * w + lane] : 0u;
                            nxt4 <<= 2;
                            const uint32_t nb_e = lane_plus1_z(cur4), nb_se = lane_plus1_z(nxt4), nb_sw = lane_minus1_z(nxt4);
                            const uint32_t rowb = mul_u24_su(cur4, ng1);
                            if (cur4 != 0) {
                            if (has0) atomicAdd((uint32_t*)(T0 + rowb + nb_e), 1u);
                            if (has1) atomicAdd((uint32_t*)(T1 + rowb + nb_se), 1u);
                            if (has2) atomicAdd((uint32_t*)(T2 + rowb + nxt4), 1u);
                            if (has3) atomicAdd((uint32_t*)(T3 + rowb + nb_sw), 1u);
                            }
                            if (sym && cur4 != 0) {
                                if (has0) atomicAdd((uint32_t*)(T0 + mul_u24_su(nb_e, ng1) + cur4), 1u);
                                if (has1) atomicAdd((uint32_t*)(T1 + mul_u24_su(nb_se, ng1) + cur4), 1u);
                                if (has2) atomicAdd((uint32_t*)(T2 + mul_u24_su(nxt4, ng1) + cur4), 1u);
                                if (has3) atomicAdd((uint32_t*)(T3 + mul_u24_su(nb_sw, ng1) + cur4), 1u);
                            }
                            cur4 = nxt4;
                        }
                    } else {
                    char* const B0 = slot0 >= 0 ? (char*)(s_P + slot0 * NN) - 4 : nullptr;
                    char* const B1 = slot1 >= 0 ? (char*)(s_P + slot1 * NN) - 4 : nullptr;
                    char* const B2 = slot2 >= 0 ? (char*)(s_P + slot2 * NN) - 4 : nullptr;
                    char* const B3 = slot3 >= 0 ? (char*)(s_P + slot3 * NN) - 4 : nullptr;
                    auto bump = [=](char* Bq, uint32_t rowb, uint32_t c4, uint32_t nb4) {
                        if (Bq != nullptr && nb4 != 0) {
                            atomicAdd((uint32_t*)(Bq + rowb + nb4), 1u);
                            if (symmetric)
                                atomicAdd((uint32_t*)(Bq + __umul24(nb4 - 4, (uint32_t)Ng) + c4), 1u);
                        }
                    };
                    // one row of pairs: `cur4` = this row's levels, `nxt4` = the row below, both as byte offsets (0 = skip)
                    auto pair_row = [&](uint32_t cur4, uint32_t nxt4) {
                        const uint32_t nb_e = lane_plus1(cur4, 0);   // (row,   col+1)  angle 0
                        const uint32_t nb_se = lane_plus1(nxt4, 0);  // (row+1, col+1)  angle 45
                        const uint32_t nb_sw = lane_minus1(nxt4, 0); // (row+1, col-1)  angle 135
                        if (cur4 != 0) {
                            const uint32_t rowb = __umul24(cur4 - 4, (uint32_t)Ng);   // 4 * (level - 1) * Ng; far below 2^24
                            bump(B0, rowb, cur4, nb_e);
                            bump(B1, rowb, cur4, nb_se);
                            bump(B2, rowb, cur4, nxt4);              // (row+1, col)    angle 90
                            bump(B3, rowb, cur4, nb_sw);
                        }
                    };
                    for (int row = r_begin; row < r_end; row++) {
                        uint32_t nxt4 = (in_col && row + 1 < (int)h) ? s_dense[(uint32_t)(row + 1) * w + lane] : 0u;
                        if (remap && nxt4) nxt4 = s_lvlmap[nxt4];   // compact index + 1 (0 stays "skip")
                        nxt4 <<= 2;
                        pair_row(cur4, nxt4);
                        cur4 = nxt4;
                    }
                    }
                } else if (B->glcm_offset == 1) {
                    // ---- boxes wider than a wave: column strips.  Lane L of a strip holds column c0 - 1 + L; lanes 1 .. 62 are the
                    // centres, lanes 0 and 63 only lend their values as the west / east neighbours of the strip's edge columns (a
                    // strip advances by 62 columns), so every neighbour is a DPP lane shift away as in the narrow-box sweep -- where
                    // the per-pixel loop below pays four bounds tests, four plane reads and the index arithmetic per pixel (two
                    // thirds of the kernel's time on 205-wide boxes).  Plain Ng x Ng matrices.
                    int slot0 = -1, slot1 = -1, slot2 = -1, slot3 = -1;
#pragma unroll
                    for (int q = 0; q < kMaxAngles; q++)
                        if (q < na_pass) {
                            int ang = B->glcm_angles[a0 + q];
                            if (ang == 0) slot0 = q; else if (ang == 45) slot1 = q; else if (ang == 90) slot2 = q; else slot3 = q;
                        }
                    const int rows_per_wave = ((int)h + NW - 1) / NW;
                    const int r_begin = wave * rows_per_wave;
                    const int r_end = (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h;
                    const bool remap = greyInfo < 0;
                    char* const B0 = slot0 >= 0 ? (char*)(s_P + slot0 * NN) - 4 : nullptr;
                    char* const B1 = slot1 >= 0 ? (char*)(s_P + slot1 * NN) - 4 : nullptr;
                    char* const B2 = slot2 >= 0 ? (char*)(s_P + slot2 * NN) - 4 : nullptr;
                    char* const B3 = slot3 >= 0 ? (char*)(s_P + slot3 * NN) - 4 : nullptr;
                    auto bump = [=](char* Bq, uint32_t rowb, uint32_t c4, uint32_t nb4) {
                        if (Bq != nullptr && nb4 != 0) {
                            atomicAdd((uint32_t*)(Bq + rowb + nb4), 1u);
                            if (symmetric)
                                atomicAdd((uint32_t*)(Bq + __umul24(nb4 - 4, (uint32_t)Ng) + c4), 1u);
                        }
                    };
                    for (uint32_t c0 = 0; c0 < w; c0 += 62) {
                        const int c = (int)c0 - 1 + lane;
                        const bool in_col = c >= 0 && c < (int)w;
                        const bool centre = in_col && lane >= 1 && lane <= 62;
                        auto lvl4 = [&](int row) -> uint32_t {      // level of (row, c) as a byte offset into a matrix row, 0 = skip
                            uint32_t v = (in_col && row < (int)h) ? (uint32_t)s_dense[(uint32_t)row * w + (uint32_t)c] : 0u;
                            if (remap && v) v = s_lvlmap[v];        // compact index + 1
                            return v << 2;
                        };
                        uint32_t cur4 = r_begin < r_end ? lvl4(r_begin) : 0u;
                        for (int row = r_begin; row < r_end; row++) {
                            const uint32_t nxt4 = lvl4(row + 1);
                            const uint32_t nb_e = lane_plus1(cur4, 0), nb_se = lane_plus1(nxt4, 0), nb_sw = lane_minus1(nxt4, 0);
                            if (centre && cur4 != 0) {
                                const uint32_t rowb = __umul24(cur4 - 4, (uint32_t)Ng);
                                bump(B0, rowb, cur4, nb_e);
                                bump(B1, rowb, cur4, nb_se);
                                bump(B2, rowb, cur4, nxt4);
                                bump(B3, rowb, cur4, nb_sw);
                            }
                            cur4 = nxt4;
                        }
                    }
                } else
                for (int row = wave; row < (int)h; row += NW) {
                    for (int col = lane; col < (int)w; col += 64) {
                        uint32_t lb = s_dense[(uint32_t)row * w + (uint32_t)col];
                        if (lb == 0)
                            continue;
                        int ib = greyInfo < 0 ? (int)s_lvlmap[lb] - 1 : (int)lb - 1;
#pragma unroll
                        for (int q = 0; q < kMaxAngles; q++) {
                            if (q >= na_pass)
                                break;
                            int r2 = row + ddy[q], c2 = col + ddx[q];
                            if (r2 < 0 || r2 >= (int)h || c2 < 0 || c2 >= (int)w)
                                continue;
                            uint32_t la = s_dense[(uint32_t)r2 * w + (uint32_t)c2];
                            if (la == 0)
                                continue;
                            int ia = greyInfo < 0 ? (int)s_lvlmap[la] - 1 : (int)la - 1;
                            atomicAdd(&s_P[q * NN + ib * Ng + ia], 1u);
                            if (symmetric)
                                atomicAdd(&s_P[q * NN + ia * Ng + ib], 1u);
                        }
                    }
                }
                grp_sync<GS, NW>();
                STAMP(11);
                if (SPLIT) {                         // the host sets this up only when every angle fits one pass
                    uint32_t* dst = B->glcm_ws + roi * B->glcm_ws_stride;
                    if (trash) {
                        // dense cell i = (q, r, c) sits at q * (Ng+1)^2 + (r+1) * (Ng+1) + c + 1.  The two small divisions go through
                        // float: (i + 1/2) / d is never closer than 1/(2d) to an integer, far beyond the rounding of the product.
                        const float inv_nn = __builtin_amdgcn_rcpf((float)NN), inv_ng = __builtin_amdgcn_rcpf((float)Ng);   // (1 ulp: far inside the margin)
                        for (int i = tid; i < na_pass * NN; i += BS) {
                            const uint32_t q = (uint32_t)(((float)i + 0.5f) * inv_nn), rem = (uint32_t)i - mul24(q, (uint32_t)NN);
                            const uint32_t r = (uint32_t)(((float)rem + 0.5f) * inv_ng), c = rem - mul24(r, (uint32_t)Ng);
                            dst[i] = s_P[mad24(q, (uint32_t)cells, mad24(r + 1u, (uint32_t)NG1, c + 1u))];
                        }
                    } else
                    for (int i = tid; i < na_pass * NN; i += BS)
                        dst[i] = s_P[i];
                } else if (SPLIT) {
                } else if (Ng <= 16) {               // small matrices: the four angles share one wave's instruction stream
                    if (wave == 0)
                        glcm_features_rows<GS, 16, kRowsTag>(s_P, na_pass, Ng, s_I, s_scr, 6 * (int)B->L.ng_cap, B->soft_nan, s_f + a0 * 32, lane);
                } else if (wave < na_pass)           // large matrices: a wave per angle, 64 lanes over the cells (NW = 4 = kMaxAngles)
                    glcm_features_rows<GS, 64, kRowsTag>(s_P + (size_t)wave * Ng * Ng, 1, Ng, s_I, s_scr + (size_t)wave * 6 * B->L.ng_cap, 6 * (int)B->L.ng_cap,
                                               B->soft_nan, s_f + (a0 + wave) * 32, lane);
            }
            grp_sync<GS, NW>();
            STAMP(12);
            // lay out: feature-major, angle-minor (output_2_buffer.cpp:336-346), then _AVE
            if (!SPLIT)
            for (int c = tid; c < kGlcmAngled * na; c += BS) {
                int k = c / na, a = c - k * na;
                o[c] = s_f[a * 32 + k];
            }
            if (!SPLIT)
            for (int j = tid; j < kGlcmAve; j += BS) {
                // calc_ave (glcm.cpp:1205-1214): libstdc++ std::reduce folds four at a time
                int k = c_glcm_ave_order[j];
                double init = 0.0;
                int a = 0;
                for (; na - a >= 4; a += 4) {
                    double v1 = s_f[a * 32 + k] + s_f[(a + 1) * 32 + k];
                    double v2 = s_f[(a + 2) * 32 + k] + s_f[(a + 3) * 32 + k];
                    init = init + (v1 + v2);
                }
                for (; a < na; a++)
                    init = init + s_f[a * 32 + k];
                o[kGlcmAngled * na + j] = na ? init / (double)na : 0.0;
            }
        }
    }

    STAMP(13);
    STAMP(14);
}

template <bool GS, bool C16, bool SPLIT, bool D8>
__global__ __launch_bounds__(kBlock, 4) void roi_features_kernel(const RoiArgs A)
{
    roi_features_body<GS, C16, SPLIT, D8>(A, blockIdx.x);
}

// the same body under tighter VGPR budgets (96 / 80): five or six workgroups per CU when their LDS fits (16-bit tables)
template <bool C16, bool SPLIT, bool D8>
__global__ __launch_bounds__(kBlock, 5) void roi_features_kernel_occ5(const RoiArgs A)
{
    roi_features_body<false, C16, SPLIT, D8, 0, 5>(A, blockIdx.x);
}
template <bool C16, bool SPLIT, bool D8>
__global__ __launch_bounds__(kBlock, 6) void roi_features_kernel_occ6(const RoiArgs A)
{
    roi_features_body<false, C16, SPLIT, D8, 0, 6>(A, blockIdx.x);
}
template <bool C16, bool SPLIT, bool D8>
__global__ __launch_bounds__(kBlock, 7) void roi_features_kernel_occ7(const RoiArgs A)
{
    roi_features_body<false, C16, SPLIT, D8, 0, 7>(A, blockIdx.x);
}
template <int FAM, int WIN>
__global__ __launch_bounds__(kBlock, 8) void roi_features_kernel_occ8(const RoiArgs A)   // 64 VGPRs: the fully compact build only
{
    roi_features_body<false, true, FAM == 1 || FAM == 3, true, FAM, 8, false, WIN>(A, (WIN == 1 && A.win.xcd_swz) ? xcd_slot(blockIdx.x, gridDim.x) : (uint64_t)blockIdx.x);
}
// The same compile-time family sets and loaders at seven and six workgroups per CU: batches whose largest ROI needs a bigger
// carve-out than the benchmark's (mixed-size data: the launch is sized by its largest ROI) keep the specialised body instead of
// falling back to the generic one with its run-time switches and scalar-register spills.
template <int FAM, int WIN>
__global__ __launch_bounds__(kBlock, 7) void roi_features_kernel_fam7(const RoiArgs A)
{
    roi_features_body<false, true, FAM == 1 || FAM == 3, true, FAM, 7, false, WIN>(A, (WIN == 1 && A.win.xcd_swz) ? xcd_slot(blockIdx.x, gridDim.x) : (uint64_t)blockIdx.x);
}
template <int FAM, int WIN>
__global__ __launch_bounds__(kBlock, 6) void roi_features_kernel_fam6(const RoiArgs A)
{
    roi_features_body<false, true, FAM == 1 || FAM == 3, true, FAM, 6, false, WIN>(A, (WIN == 1 && A.win.xcd_swz) ? xcd_slot(blockIdx.x, gridDim.x) : (uint64_t)blockIdx.x);
}

// the reference's default grey depth: 16-bit matrices, marginal-based features (three workgroups per CU)
template <int WIN>
__global__ __launch_bounds__(kBlock, 4) void roi_features_kernel_g16(const RoiArgs A)
{
    roi_features_body<false, true, false, true, 0, 9, true, WIN>(A, (WIN == 1 && A.win.xcd_swz) ? xcd_slot(blockIdx.x, gridDim.x) : (uint64_t)blockIdx.x);
}

// ... and its GLCM-only form on eight waves (512 threads) per ROI
template <int WIN>
__global__ __launch_bounds__(512, 8) void roi_features_kernel_g16w8(const RoiArgs A)
{
    roi_features_body<false, true, false, true, 4, 9, true, WIN, 8>(A, (WIN == 1 && A.win.xcd_swz) ? xcd_slot(blockIdx.x, gridDim.x) : (uint64_t)blockIdx.x);
}

// A launch that serves some classes only (SpillArgs::class_mask, whole-batch launches): is this ROI one of them?  (The feature kernels
// derive the class from what they load anyway; the two kernels below would otherwise redo the other launch group's ROIs.)
__device__ __forceinline__ bool glcm_roi_in_launch(const RoiArgs& A, uint64_t roi)
{
    if (A.sp.class_mask == 0) return true;
    return roi_in_launch(A.sp, (uint32_t)(A.px_offset[roi + 1] - A.px_offset[roi]), A.bbox_w[roi], A.bbox_h[roi], A.max_inten[roi] - A.min_inten[roi]);
}

// ---- GLCM features of small matrices as their own launch -------------------------------------------------------------
// One wave per ROI, the four angles in the wave's four DPP rows (glcm_features_rows<.., 16>), four ROIs per workgroup: every
// lane of every wave works, where the same code inside roi_features_kernel leaves three of four waves waiting.  Input: the
// co-occurrence counts roi_features_kernel exported (na * Ng^2 words per ROI, <= 4 KiB); output: the ROI's GLCM columns.
__global__ __launch_bounds__(kBlock, 4) void glcm_features_kernel(const RoiArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t roi;
    if (!roi_of_slot(A.sp, (uint64_t)blockIdx.x * kWaves + wave, A.n_roi, roi))
        return;
    const int Ng = glcm_roi_in_launch(A, roi) ? (int)A.glcm_ng[roi] : 0;
    if (Ng == 0)
        return;                                       // degenerate / skipped ROI: roi_features_kernel wrote the columns
    const int na = A.glcm_na, ngc = (int)A.L.ng_cap, NN = Ng * Ng;
    // per-wave carve-out: counts [na * ngc^2] u32 | (unused) [ngc] | marginals [4][6 * ngc] | features [4][32] | sums [4][32]  (doubles)
    const size_t per_wave = (((size_t)4 * kMaxAngles * ngc * ngc + 15) & ~(size_t)15) + 8ull * (ngc + kMaxAngles * 6 * ngc + 2 * kMaxAngles * 32);
    unsigned char* base = lds_raw + (size_t)wave * per_wave;
    uint32_t* s_P = (uint32_t*)base;
    double* s_scr = (double*)(base + (((size_t)4 * kMaxAngles * ngc * ngc + 15) & ~(size_t)15)) + ngc;
    double* s_f = s_scr + kMaxAngles * 6 * ngc;
    const uint32_t* src = A.glcm_ws + roi * A.glcm_ws_stride;
    // the ROI's counts: 16 bytes per lane and trip (na * Ng^2 = 256 words for four 8 x 8 matrices: one load per lane)
    const int nw = na * NN;
    if (((nw | (int)A.glcm_ws_stride) & 3) == 0) {
#pragma unroll 1
        for (int i = lane; i < (nw >> 2); i += 64) ((uint4*)s_P)[i] = ((const uint4*)src)[i];
    } else {
#pragma unroll 1
        for (int i = lane; i < nw; i += 64) s_P[i] = src[i];
    }
    wav_sync<false>();
    glcm_features_wave16(s_P, na, Ng, s_scr, 6 * ngc, A.soft_nan, s_f, s_f + kMaxAngles * 32, lane);   // level values I[i] = i + 1 (glcm.cpp:400-408)
    wav_sync<false>();
    double* o = A.out + roi * A.ld + A.col_glcm;
    const int sh = na == 4 ? 2 : na == 2 ? 1 : na == 1 ? 0 : -1;  // the usual angle counts split c = k * na + a by a shift
    for (int c = lane; c < kGlcmAngled * na; c += 64) {           // feature-major, angle-minor (output_2_buffer.cpp:336-346)
        const int k = sh >= 0 ? c >> sh : c / na, a = c - k * na;
        o[c] = s_f[a * 32 + k];
    }
    for (int j = lane; j < kGlcmAve; j += 64) {                   // calc_ave (glcm.cpp:1205-1214): std::reduce folds four at a time
        const int k = c_glcm_ave_order[j];
        double init = 0.0;
        int a = 0;
        for (; na - a >= 4; a += 4) {
            const double v1 = s_f[a * 32 + k] + s_f[(a + 1) * 32 + k];
            const double v2 = s_f[(a + 2) * 32 + k] + s_f[(a + 3) * 32 + k];
            init = init + (v1 + v2);
        }
        for (; a < na; a++)
            init = init + s_f[a * 32 + k];
        // (dividing by 4, 2 or 1 is a multiplication by the exact reciprocal: the same rounding, a tenth of the instructions)
        o[kGlcmAngled * na + j] = na == 4 ? init * 0.25 : na == 2 ? init * 0.5 : na ? init / (double)na : 0.0;
    }
}

__host__ __device__ inline size_t glcm8_cnt_bytes(uint32_t ngc)   // a count block, at least the 1 KiB the ROI's features take in its place
{
    const size_t c = ((size_t)4 * kMaxAngles * ngc * ngc + 15) & ~(size_t)15, fbytes = 8ull * kMaxAngles * 32;
    return c > fbytes ? c : fbytes;
}
// The same for matrices of up to 8 levels: two ROIs per wave (glcm_features_wave8), eight per workgroup.
__global__ __launch_bounds__(kBlock, 8) void glcm_features_kernel8(const RoiArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t slot0 = ((uint64_t)blockIdx.x * kWaves + wave) * 2;
    uint64_t roi[2] = {0, 0};
    int Ng[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; q++)
        if (roi_of_slot(A.sp, slot0 + q, A.n_roi, roi[q]) && glcm_roi_in_launch(A, roi[q])) Ng[q] = (int)A.glcm_ng[roi[q]];
    if (Ng[0] == 0 && Ng[1] == 0)
        return;                                       // degenerate / skipped ROIs: roi_features_kernel wrote the columns
    const int na = A.glcm_na, ngc = (int)A.L.ng_cap;
    // per-wave carve-out: count blocks [2] (an ROI's features are written over its dead counts: 4 angles x 32 doubles = the 1 KiB a block
    // holds at 8 levels) | per group (8): pcol [8] prow [8] | sums [8][32]  (doubles) -- 5 KiB per wave (round 6; 8 KiB before: five
    // workgroups per CU, now the register count's six waves per SIMD)
    const size_t cnt_bytes = glcm8_cnt_bytes((uint32_t)ngc);
    const size_t per_wave = 2 * cnt_bytes + 8ull * (2 * kMaxAngles * 16 + 2 * kMaxAngles * 32);
    unsigned char* base = lds_raw + (size_t)wave * per_wave;
    uint32_t* s_P[2] = {(uint32_t*)base, (uint32_t*)(base + cnt_bytes)};
    double* s_scr = (double*)(base + 2 * cnt_bytes);
    double* s_sum = s_scr + 2 * kMaxAngles * 16;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (Ng[q] == 0) continue;
        const uint32_t* src = A.glcm_ws + roi[q] * A.glcm_ws_stride;
        const int nw = na * Ng[q] * Ng[q];
        if (((nw | (int)A.glcm_ws_stride) & 3) == 0) {
#pragma unroll 1
            for (int i = lane; i < (nw >> 2); i += 64) ((uint4*)s_P[q])[i] = ((const uint4*)src)[i];
        } else {
#pragma unroll 1
            for (int i = lane; i < nw; i += 64) s_P[q][i] = src[i];
        }
    }
    wav_sync<false>();
    // (group g = ROI g >> 2, angle g & 3 writes f = fslots + g * 32 doubles: the ROI's own count block)
    glcm_features_wave8(s_P[0], s_P[1], na, Ng[0], Ng[1], s_scr, 16, A.soft_nan, (double*)base, (size_t)(cnt_bytes >> 3) - 4 * 32, s_sum, lane);
    wav_sync<false>();
    const int sh = na == 4 ? 2 : na == 2 ? 1 : na == 1 ? 0 : -1;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (Ng[q] == 0) continue;
        double* o = A.out + roi[q] * A.ld + A.col_glcm;
        const double* fq = (const double*)(base + (size_t)q * cnt_bytes);
        for (int c = lane; c < kGlcmAngled * na; c += 64) {           // feature-major, angle-minor (output_2_buffer.cpp:336-346)
            const int k = sh >= 0 ? c >> sh : c / na, a = c - k * na;
            o[c] = fq[a * 32 + k];
        }
        for (int j = lane; j < kGlcmAve; j += 64) {                   // calc_ave (glcm.cpp:1205-1214): std::reduce folds four at a time
            const int k = c_glcm_ave_order[j];
            double init = 0.0;
            int a = 0;
            for (; na - a >= 4; a += 4) {
                const double v1 = fq[a * 32 + k] + fq[(a + 1) * 32 + k];
                const double v2 = fq[(a + 2) * 32 + k] + fq[(a + 3) * 32 + k];
                init = init + (v1 + v2);
            }
            for (; a < na; a++) init = init + fq[a * 32 + k];
            o[kGlcmAngled * na + j] = na == 4 ? init * 0.25 : na == 2 ? init * 0.5 : na ? init / (double)na : 0.0;
        }
    }
}

size_t glcm_features8_lds(uint32_t ng_cap)
{
    return (2 * glcm8_cnt_bytes(ng_cap) + 8ull * (2 * kMaxAngles * 16 + 2 * kMaxAngles * 32)) * kWaves;
}

size_t glcm_features_lds(uint32_t ng_cap)
{
    const size_t per_wave = (((size_t)4 * kMaxAngles * ng_cap * ng_cap + 15) & ~(size_t)15) + 8ull * (ng_cap + kMaxAngles * 6 * ng_cap + 2 * kMaxAngles * 32);
    return per_wave * kWaves;
}

size_t roi_features_max_lds()
{
    return 160 * 1024; // gfx950: 160 KiB per CU, all of it usable by one workgroup
}

namespace {

// The 8-bit plane is addressed by absolute LDS addresses (it opens the carve-out, and the carve-out is expected at LDS address 0):
// true as long as the kernel owns no static LDS in front of its dynamic allocation.
int no_static_lds(const void* f)
{
    hipFuncAttributes at;
    if (hipError_t e = hipFuncGetAttributes(&at, f); e != hipSuccess)
        return (int)e;
    return at.sharedSizeBytes == 0 ? 0 : (int)hipErrorInvalidConfiguration;
}

template <bool C16, bool SPLIT, bool D8>
int launch_lds_variant(const RoiArgs& a, hipStream_t st, uint32_t grid)
{
    static DeviceOnce optin;
    if (int orc = optin.run([]() -> int {
        const void* fns[22] = {(const void*)roi_features_kernel<false, C16, SPLIT, D8>, (const void*)roi_features_kernel_occ5<C16, SPLIT, D8>,
                               (const void*)roi_features_kernel_occ6<C16, SPLIT, D8>, (const void*)roi_features_kernel_occ7<C16, SPLIT, D8>,
                               (const void*)roi_features_kernel_occ8<1, 0>, (const void*)roi_features_kernel_occ8<2, 0>,
                               (const void*)roi_features_kernel_occ8<1, 1>, (const void*)roi_features_kernel_occ8<2, 1>,
                               (const void*)roi_features_kernel_fam7<1, 0>, (const void*)roi_features_kernel_fam7<2, 0>,
                               (const void*)roi_features_kernel_fam7<1, 1>, (const void*)roi_features_kernel_fam7<2, 1>,
                               (const void*)roi_features_kernel_fam6<1, 0>, (const void*)roi_features_kernel_fam6<2, 0>,
                               (const void*)roi_features_kernel_fam6<1, 1>, (const void*)roi_features_kernel_fam6<2, 1>,
                               (const void*)roi_features_kernel_occ8<3, 0>, (const void*)roi_features_kernel_occ8<3, 1>,
                               (const void*)roi_features_kernel_fam7<3, 0>, (const void*)roi_features_kernel_fam7<3, 1>,
                               (const void*)roi_features_kernel_fam6<3, 0>, (const void*)roi_features_kernel_fam6<3, 1>};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)roi_features_max_lds());
            if (e != hipSuccess)
                return (int)e;
            if (int src = no_static_lds(f))
                return src;
        }
        return 0;
    }))
        return orc;
    // occupancy follows the carve-out: 8 / 7 / 6 / 5 / 4 workgroups per CU with builds held to 64 / 72 / 80 / 96 / 128 VGPRs
    const size_t lds = roi_features_max_lds();
    // the 64-VGPR tier exists for the two compile-time family sets only
    const bool int_only = !(a.mask & NYXHIP_FAM_GLCM) && (a.mask & NYXHIP_FAM_INTENSITY);
    const bool int_glcm = (a.mask & NYXHIP_FAM_GLCM) && (a.mask & NYXHIP_FAM_INTENSITY) && a.L.dense8 && a.glcm_ws != nullptr && !a.ibsi &&
                          a.grey_depth > 0 && a.grey_depth <= 16;
    const bool glcm_only = (a.mask & NYXHIP_FAM_GLCM) && !(a.mask & NYXHIP_FAM_INTENSITY) && a.L.dense8 && a.glcm_ws != nullptr && !a.ibsi &&
                           a.grey_depth > 0 && a.grey_depth <= 16;
    const bool fam_ok = C16 && SPLIT && D8 && (int_only || int_glcm || glcm_only);   // a compile-time family set of the compact builds applies
    int occ = 4;
    for (int o = fam_ok ? 8 : 7; o > 4; o--)
        if ((size_t)o * a.L.total <= lds) { occ = o; break; }
    if (const char* e = getenv("NYXHIP_MAX_OCC")) occ = occ < atoi(e) ? occ : (atoi(e) < 4 ? 4 : atoi(e));   // tuning knob (bench experiments)
    const bool win = a.win.inten != nullptr;
    if (win && a.win.xcd_swz) grid = (grid + 7u) & ~7u;               // (xcd_slot: every XCD walks its own eighth of the slots)
    if (fam_ok && glcm_only && occ >= 6) {
        if (occ == 8) { if (win) hipLaunchKernelGGL((roi_features_kernel_occ8<3, 1>), dim3(grid), dim3(kBlock), a.L.total, st, a);
                        else hipLaunchKernelGGL((roi_features_kernel_occ8<3, 0>), dim3(grid), dim3(kBlock), a.L.total, st, a); }
        else if (occ == 7) { if (win) hipLaunchKernelGGL((roi_features_kernel_fam7<3, 1>), dim3(grid), dim3(kBlock), a.L.total, st, a);
                             else hipLaunchKernelGGL((roi_features_kernel_fam7<3, 0>), dim3(grid), dim3(kBlock), a.L.total, st, a); }
        else { if (win) hipLaunchKernelGGL((roi_features_kernel_fam6<3, 1>), dim3(grid), dim3(kBlock), a.L.total, st, a);
               else hipLaunchKernelGGL((roi_features_kernel_fam6<3, 0>), dim3(grid), dim3(kBlock), a.L.total, st, a); }
    }
    else if (occ == 8 && int_only && win) hipLaunchKernelGGL((roi_features_kernel_occ8<2, 1>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (occ == 8 && int_only) hipLaunchKernelGGL((roi_features_kernel_occ8<2, 0>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (occ == 8 && win) hipLaunchKernelGGL((roi_features_kernel_occ8<1, 1>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (occ == 8) hipLaunchKernelGGL((roi_features_kernel_occ8<1, 0>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (fam_ok && occ == 7 && int_only && win) hipLaunchKernelGGL((roi_features_kernel_fam7<2, 1>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (fam_ok && occ == 7 && int_only) hipLaunchKernelGGL((roi_features_kernel_fam7<2, 0>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (fam_ok && occ == 7 && win) hipLaunchKernelGGL((roi_features_kernel_fam7<1, 1>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (fam_ok && occ == 7) hipLaunchKernelGGL((roi_features_kernel_fam7<1, 0>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (fam_ok && occ == 6 && int_only && win) hipLaunchKernelGGL((roi_features_kernel_fam6<2, 1>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (fam_ok && occ == 6 && int_only) hipLaunchKernelGGL((roi_features_kernel_fam6<2, 0>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (fam_ok && occ == 6 && win) hipLaunchKernelGGL((roi_features_kernel_fam6<1, 1>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (fam_ok && occ == 6) hipLaunchKernelGGL((roi_features_kernel_fam6<1, 0>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (occ == 7) hipLaunchKernelGGL((roi_features_kernel_occ7<C16, SPLIT, D8>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (occ == 6) hipLaunchKernelGGL((roi_features_kernel_occ6<C16, SPLIT, D8>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else if (occ == 5) hipLaunchKernelGGL((roi_features_kernel_occ5<C16, SPLIT, D8>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    else hipLaunchKernelGGL((roi_features_kernel<false, C16, SPLIT, D8>), dim3(grid), dim3(kBlock), a.L.total, st, a);
    return (int)hipGetLastError();
}

} // namespace

int launch_roi_features(const RoiArgs& a, void* stream, uint32_t grid)
{
    if (grid == 0)
        return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool c16 = a.L.cnt16 != 0;
    if (a.sp.scratch) {
        if (c16) hipLaunchKernelGGL((roi_features_kernel<true, true, false, false>), dim3(grid), dim3(kBlock), a.L.gs_lds_bytes, st, a);
        else hipLaunchKernelGGL((roi_features_kernel<true, false, false, false>), dim3(grid), dim3(kBlock), a.L.gs_lds_bytes, st, a);
        return (int)hipGetLastError();
    }
    if (getenv("NYXHIP_DEBUG")) fprintf(stderr, "[nyxhip] features launch: g16 %u dense8 %u cnt16 %u ng_cap %u app %u total %u mask %u gd %d\n", a.L.g16, a.L.dense8, a.L.cnt16, a.L.ng_cap, a.L.app, a.L.total, a.mask, a.grey_depth);
    // the smallest size class on its own kernel: a wave per ROI (roi_small.hip).  small_class: 1 = a list / filtered launch of class 0,
    // 2 = a whole-batch launch whose stated extrema promise class 0 only
    static const bool no_small = [] { const char* e = getenv("NYXHIP_NO_SMALL"); return e && *e && *e != '0'; }();   // A/B knob
    if (a.L.g16) {
        if (a.small_class && !no_small && roi_small_supported(a)) return launch_roi_small(a, st, grid, a.small_class == 2);
        static DeviceOnce optin;
        if (int orc = optin.run([]() -> int {
                for (const void* f : {(const void*)roi_features_kernel_g16<0>, (const void*)roi_features_kernel_g16<1>,
                                      (const void*)roi_features_kernel_g16w8<0>, (const void*)roi_features_kernel_g16w8<1>}) {
                    if (hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)roi_features_max_lds()); e != hipSuccess)
                        return (int)e;
                    if (int src = no_static_lds(f))
                        return src;
                }
                return 0;
            }))
            return orc;
        static const bool w4 = [] { const char* e = getenv("NYXHIP_G16_W4"); return e && *e && *e != '0'; }();   // A/B knob: four waves per ROI
        static const uint32_t g16_w8_min = [] { const char* e = getenv("NYXHIP_G16_W8_MIN"); return e && *e ? (uint32_t)atoi(e) : 48u * 48u; }();   // tuning knob: smallest class box for eight waves
        // (eight waves pay where the load and the sweep are long: 2821-px ROIs 23.0 -> 22.1 ns, 1009-px ROIs 18.5 -> 18.8: by the class's largest box)
        if (!(a.mask & NYXHIP_FAM_INTENSITY) && !w4 && a.L.dense_cap >= g16_w8_min) {
            if (a.win.inten != nullptr) hipLaunchKernelGGL(roi_features_kernel_g16w8<1>, dim3(a.win.xcd_swz ? (grid + 7u) & ~7u : grid), dim3(512), a.L.total, st, a);
            else hipLaunchKernelGGL(roi_features_kernel_g16w8<0>, dim3(grid), dim3(512), a.L.total, st, a);
        } else
        if (a.win.inten != nullptr) hipLaunchKernelGGL(roi_features_kernel_g16<1>, dim3(a.win.xcd_swz ? (grid + 7u) & ~7u : grid), dim3(kBlock), a.L.total, st, a);
        else hipLaunchKernelGGL(roi_features_kernel_g16<0>, dim3(grid), dim3(kBlock), a.L.total, st, a);
        return (int)hipGetLastError();
    }
    const bool split = a.glcm_ws != nullptr;
    // dense8 is set by the host only together with the 16-bit tables and the split; an intensity-only launch never touches
    // the GLCM code, so it can run the fully compact build (and its 64-VGPR tier) as well
    const bool d8 = a.L.dense8 != 0 || (c16 && !(a.mask & NYXHIP_FAM_GLCM));
    int rc;
    if (a.small_class && !no_small && (c16 || !(a.mask & NYXHIP_FAM_INTENSITY)) && roi_small_supported(a)) rc = launch_roi_small(a, st, grid, a.small_class == 2);
    else
    rc = d8 ? launch_lds_variant<true, true, true>(a, st, grid)
           : c16 ? (split ? launch_lds_variant<true, true, false>(a, st, grid) : launch_lds_variant<true, false, false>(a, st, grid))
                 : (split ? launch_lds_variant<false, true, false>(a, st, grid) : launch_lds_variant<false, false, false>(a, st, grid));
    if (rc == 0 && split && a.glcm_feats != 1) {
        static const bool no_pairs = [] { const char* e = getenv("NYXHIP_GLCM_NO_PAIRS"); return e && *e && *e != '0'; }();   // A/B knob
        RoiArgs af = a;
        if (a.glcm_feats == 2) af.sp.class_mask = 0;          // (the class-0 group of this call left its counts for this launch: everybody with a matrix order is derived)
        if (a.L.ng_cap <= 8 && !no_pairs)         // eight lanes per angle, two ROIs per wave
            hipLaunchKernelGGL(glcm_features_kernel8, dim3((grid + 2 * kWaves - 1) / (2 * kWaves)), dim3(kBlock), glcm_features8_lds(a.L.ng_cap), st, af);
        else
            hipLaunchKernelGGL(glcm_features_kernel, dim3((grid + kWaves - 1) / kWaves), dim3(kBlock), glcm_features_lds(a.L.ng_cap), st, af);
        rc = (int)hipGetLastError();
    }
    return rc;
}

} // namespace nyxhip
