// glcm_w64.h -- the part of the 17..64-level GLCM feature pass that starts from the MARGINAL counts of an angle (round 6): shared by
// roi_features.hip (glcm_features_wave64_v2: marginals from the dense 16-bit matrix) and roi_small.hip (marginals and per-cell terms from
// the pairs of a <= 256-pixel ROI).  Reference: /root/reference/src/nyx/features/glcm.cpp:487-1202.
#pragma once
#include <hip/hip_runtime.h>
#include "device_math.h"
#include "glcm_rows.h"

namespace nyxhip {
namespace {

// weights of the difference distribution at 64 levels, per |x - y| = l: 1 / (1 + l^2) (f_idm :685), 1 / (1 + l^2 / Ng^2) (:1083), 1 / (1 + l) (:1096),
// 1 / (1 + l / Ng) (:1110), 1 / l^2 (:1123) -- five divisions per lane and angle otherwise
struct GlcmW64 {
    double w[5][64];
    constexpr GlcmW64() : w{} {
        for (int l = 0; l < 64; l++) {
            w[0][l] = 1.0 / (double)(1 + l * l); w[1][l] = 1.0 / (1.0 + (double)(l * l) / 4096.0); w[2][l] = 1.0 / (1.0 + (double)l);
            w[3][l] = 1.0 / (1.0 + (double)l / 64.0); w[4][l] = l ? 1.0 / ((double)l * (double)l) : 0.0;
        }
    }
};
__device__ const GlcmW64 c_glcm_w64{};

// The sums of an angle from its marginal counts and per-cell totals; a wave per angle, lane l < Ng owns level l + 1 (row count rc, column
// count cc, difference count dc of |x - y| = l) and the sum-distribution counts pxpy_c[u] of x + y - 2 = l + 64 u.  Leaves in the
// angle's block f[32] | sm[32] what glcm_features_final needs.
template <int NG>
__device__ __forceinline__ void glcm_w64_tail(const int Ng, const int lane, const uint32_t rc, const uint32_t cc, const uint32_t dc, const uint32_t (&pxpy_c)[2],
                                              const uint32_t csum, const double sum_p, const double inv_sum_p, const double pcol, const double prow, const double pxmy,
                                              const double (&pxpy)[2], const double ent, const double hxy1c, const double hxy2, uint32_t asm_i, uint32_t cmax,
                                              double* const f)
{
    double* const sm = f + 32;
    const bool act = lane < Ng;
    const uint32_t l1 = (uint32_t)lane + 1u;
    const double hxy1 = hxy1c * inv_sum_p;
    const double hx_t = act ? plogp(pcol, pcol) : 0.0;               // :873-874
    wav_sync<false>();
    {
        double t4[4] = {ent, hxy1, hxy2, hx_t};
        const double tot = wave_transpose_sum4(t4);                  // lane L holds the total of slot (L >> 4) & 3
        if ((lane & 15) == 0) sm[1 + (lane >> 4)] = tot;             // sm[1] ent, [2] hxy1, [3] hxy2, [4] hx
    }
    // ---- the integer sums in ONE transposed reduction (six six-step butterflies before): S_r, S_c, the contrast and dissimilarity
    // numerators, 4 ACOR + contrast (f_GLCM_ACOR :961 from the two families of diagonals: I J = ((I + J)^2 - (I - J)^2) / 4, so
    // 4 ACOR = sum_k (k + 2)^2 n_{x+y}(k) - sum_d d^2 n_{x-y}(d); sum_p < 65536 and k + 2 <= 130 keep everything inside 32 bits), sum cnt^2
    uint32_t Sr_i, Sc_i, con_i, dis_i, acor_i;
    {
        const uint32_t k0 = (uint32_t)lane + 2u, k1 = (uint32_t)lane + 66u;
        uint32_t t8[8] = {mul24(rc, l1), mul24(cc, l1), mul24(dc, mul24((uint32_t)lane, (uint32_t)lane)), mul24(dc, (uint32_t)lane),
                          mad24(pxpy_c[1], mul24(k1, k1), mul24(pxpy_c[0], mul24(k0, k0))), asm_i, 0u, 0u};
        const uint32_t tot = wave_transpose_sum8_u32(t8, lane);      // lane L holds the total of slot (L >> 3) & 7
        Sr_i = (uint32_t)__builtin_amdgcn_readlane((int)tot, 0); Sc_i = (uint32_t)__builtin_amdgcn_readlane((int)tot, 8);
        con_i = (uint32_t)__builtin_amdgcn_readlane((int)tot, 16); dis_i = (uint32_t)__builtin_amdgcn_readlane((int)tot, 24);
        acor_i = ((uint32_t)__builtin_amdgcn_readlane((int)tot, 32) - con_i) >> 2;
        asm_i = (uint32_t)__builtin_amdgcn_readlane((int)tot, 40);
    }
    cmax = wave_max_u32(cmax);
    const double mr = fdiv((double)Sr_i, sum_p), mc = fdiv((double)Sc_i, sum_p);
    const double davg = fdiv((double)dis_i, sum_p);                  // f_difference_avg :791-792 = sum k p_{x-y}(k): the dissimilarity's exact numerator

    // ---- one term per lane: features of the marginal distributions -----------------------------------------------------------
    double t16[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t16[k] = 0.0;
    if (act) {
        const double dr = (double)l1 - mr, dr2 = dr * dr, dcl = (double)l1 - mc;
        t16[0] = prow * dr2;                                         // f_corr :617
        t16[1] = pcol * (dcl * dcl);                                 // :626
        t16[2] = (double)rc * dr2;                                   // f_var :672
        t16[3] = pcol * dr2;                                         // f_GLCM_JVAR :1196-1199
        const double q = pxmy, kd = (double)lane, Ngd = (double)Ng;
        t16[5] = q != 0 ? plogp(q, q) : 0.0;                         // f_dentropy :778-781
        if (NG == 64) {                                              // (reciprocal weights from the table: within an ulp of the quotients)
            t16[4] = q * c_glcm_w64.w[0][lane]; t16[6] = q * c_glcm_w64.w[1][lane]; t16[7] = q * c_glcm_w64.w[2][lane];
            t16[8] = q * c_glcm_w64.w[3][lane]; t16[9] = q * c_glcm_w64.w[4][lane];
        } else {
            t16[4] = fdiv(q, (double)(1 + lane * lane));             // f_idm :685-687
            t16[6] = fdiv(q, 1.0 + fdiv(kd * kd, Ngd * Ngd));        // :1083-1084
            t16[7] = fdiv(q, 1.0 + kd);                              // :1096-1097
            t16[8] = fdiv(q, 1.0 + fdiv(kd, Ngd));                   // :1110-1111
            t16[9] = lane >= 1 ? q / (kd * kd) : 0.0;                // :1123-1128
        }
        const double dk = kd - davg;
        t16[10] = dk * dk * q;                                       // f_dvar (glcm.cpp:742-766)
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int k = lane + 64 * u;
        if (k < 2 * Ng - 1) {
            const double q = pxpy[u], ks = (double)(k + 2);          // I[x] + I[k - x] = k + 2
            t16[11] += ks * q;                                       // f_savg :700-701
            t16[12] += plogp(q, q);                                  // f_sentropy :712-716
            const double m = ks - mc - mc, m2 = m * m;               // by_row_mean (:531-536) = mc; CLUPROM :985, CLUSHADE :1007, CLUTEND :1034
            t16[13] += m2 * m2 * q;
            t16[14] += m2 * m * q;
            t16[15] += m2 * q;
        }
    }
    {
        const double tot = wave_transpose_sum16(t16, lane);          // lane L holds the total of slot (L >> 2) & 15
        if ((lane & 3) == 0) sm[8 + (lane >> 2)] = tot;
    }
    // what the final formulas need beside the sums above (glcm_features_final: a LANE per angle, once per workgroup)
    if (lane == 0) {
        sm[0] = (double)csum; sm[5] = (double)Sr_i; sm[6] = (double)Sc_i; sm[7] = (double)con_i;
        sm[24] = (double)dis_i; sm[25] = (double)acor_i; sm[26] = (double)asm_i; sm[27] = (double)cmax;
    }
    wav_sync<false>();
}

// The closing formulas of an angle from the sums glcm_features_wave64_v2 left in its block: run by ONE lane per angle after the
// workgroup's barrier (as the last block of every angle's wave it was the same ~250 single-lane instructions four times over).
__device__ __forceinline__ void glcm_features_final(uint32_t* blk, double soft_nan)
{
    double* const f = (double*)blk;
    const double* const sm = f + 32;
    const double csum = sm[0], Sr = sm[5], Sc = sm[6], con = sm[7], dis = sm[24], acor = sm[25], asm_s = sm[26], cmax = sm[27];
    const bool empty = csum == 0.0;
    const double sum_p = empty ? 1.0 : csum, inv_sum_p = fdiv(1.0, sum_p);
    const double ent_t = sm[1], hxy1_t = sm[2], hxy2_t = sm[3], hx = sm[4];
    const double asm_t = asm_s * inv_sum_p * inv_sum_p;
    const double mr = fdiv(Sr, sum_p);
    const double cov_t = fdiv(acor * sum_p - Sr * Sc, sum_p * sum_p);   // sum (r - mr)(c - mc) p, exact numerator
    const double s8[16] = {sm[8], sm[9], sm[10], sm[11], sm[12], sm[13], sm[14], sm[15], sm[16], sm[17], sm[18], sm[19], sm[20], sm[21], sm[22], sm[23]};
    f[G_ASM] = asm_t;
    f[G_ENERGY] = asm_t;
    f[G_CONTRAST] = fdiv(con, sum_p);
    f[G_ACOR] = fdiv(acor, sum_p);
    f[G_ENTROPY] = -ent_t;
    f[G_JE] = -ent_t;
    f[G_DIS] = fdiv(dis, sum_p);
    f[G_JMAX] = cmax * inv_sum_p;
    f[G_JAVE] = mr;
    f[G_VARIANCE] = fdiv(s8[2], sum_p);
    f[G_CLUPROM] = s8[13];
    f[G_CLUSHADE] = s8[14];
    f[G_CLUTEND] = s8[15];
    f[G_SUMVARIANCE] = s8[15];                    // glcm.cpp:323-326
    f[G_JVAR] = s8[3];
    const double denom = sqrt(s8[0]) * sqrt(s8[1]);                  // f_corr tail, glcm.cpp:619-643
    f[G_CORRELATION] = !(denom > 0.0) ? soft_nan : cov_t / denom;
    f[G_INFOMEAS2] = sqrt(fabs(1 - exp(-2 * (-hxy2_t + ent_t))));    // glcm.cpp:913 (HXY = ent)
    f[G_IDM] = s8[4];
    f[G_HOM2] = s8[4];
    f[G_HOM1] = s8[7];
    f[G_SUMAVERAGE] = s8[11];
    f[G_SUMENTROPY] = -s8[12];
    f[G_DIFENTRO] = -s8[5];
    f[G_DIFAVE] = fdiv(dis, sum_p);
    f[G_DIFVAR] = s8[10];
    f[G_IDMN] = s8[6];
    f[G_ID] = s8[7];
    f[G_IDN] = s8[8];
    f[G_IV] = s8[9];
    const double r1 = (ent_t - hxy1_t) / hx;      // f_info_meas_corr1, glcm.cpp:880-883
    f[G_INFOMEAS1] = isfinite(r1) ? r1 : soft_nan;
    if (empty)                                    // blank matrix: all 30 values = soft NaN (glcm.cpp:260-295)
        for (int k = 0; k < kGlcmAngled; k++)
            f[k] = soft_nan;
}

} // namespace
} // namespace nyxhip
