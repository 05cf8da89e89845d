// glcm_rows.h -- column slots of the INTENSITY / GLCM blocks and the general GLCM feature routine (any matrix order, any level
// values), shared by roi_features.hip (LDS launches and the one-workgroup workspace launches) and roi_large.hip (the
// cooperative large-ROI path).  Reference: features/glcm.cpp:487-1202.
#pragma once
#include <hip/hip_runtime.h>
#include "device_math.h"
#include "roi_kernel.h"

namespace nyxhip {

// ---- column slots (Feature2D enum order, featureset.h) ---------------------------
enum {
    I_COV = 0, I_COVERED_IMAGE_INTENSITY_RANGE, I_ENERGY, I_ENTROPY, I_EXCESS_KURTOSIS,
    I_HYPERFLATNESS, I_HYPERSKEWNESS, I_INTEGRATED_INTENSITY, I_INTERQUARTILE_RANGE,
    I_KURTOSIS, I_MAX, I_MEAN, I_MEAN_ABSOLUTE_DEVIATION, I_MEDIAN,
    I_MEDIAN_ABSOLUTE_DEVIATION, I_MIN, I_MODE, I_P01, I_P10, I_P25, I_P75, I_P90, I_P99,
    I_QCOD, I_RANGE, I_ROBUST_MEAN, I_ROBUST_MEAN_ABSOLUTE_DEVIATION, I_ROOT_MEAN_SQUARED,
    I_SKEWNESS, I_STANDARD_DEVIATION, I_STANDARD_DEVIATION_BIASED, I_STANDARD_ERROR,
    I_VARIANCE, I_VARIANCE_BIASED, I_UNIFORMITY, I_UNIFORMITY_PIU
};
enum {
    G_ASM = 0, G_ACOR, G_CLUPROM, G_CLUSHADE, G_CLUTEND, G_CONTRAST, G_CORRELATION, G_DIFAVE,
    G_DIFENTRO, G_DIFVAR, G_DIS, G_ENERGY, G_ENTROPY, G_HOM1, G_HOM2, G_ID, G_IDN, G_IDM, G_IDMN,
    G_INFOMEAS1, G_INFOMEAS2, G_IV, G_JAVE, G_JE, G_JMAX, G_JVAR, G_SUMAVERAGE, G_SUMENTROPY,
    G_SUMVARIANCE, G_VARIANCE
};
// the 29 _AVE columns (featureset.h:205-233) as indices into the angled block
static __constant__ int c_glcm_ave_order[kGlcmAve] = {
    G_ASM, G_ACOR, G_CLUPROM, G_CLUSHADE, G_CLUTEND, G_CONTRAST, G_CORRELATION, G_DIFAVE,
    G_DIFENTRO, G_DIFVAR, G_DIS, G_ENERGY, G_ENTROPY, G_HOM1, G_ID, G_IDN, G_IDM, G_IDMN, G_IV,
    G_JAVE, G_JE, G_INFOMEAS1, G_INFOMEAS2, G_VARIANCE, G_JMAX, G_JVAR, G_SUMAVERAGE,
    G_SUMENTROPY, G_SUMVARIANCE};

// ---- GLCM features: one wave, one DPP row (16 lanes) per angle ------------------------------
// The four 16-lane rows of the wave work on four angles at once; every reduction is a 4-step butterfly
// inside the row (row16_sum), so the instruction stream is issued once for all angles.
// Pslots: matrices of this pass, Pslots[slot*NN + center*Ng + neighbour]  (== SimpleMatrix::xy(a,b)++ with
//    a = neighbour level, b = centre level, glcm.cpp:437-472; xy(x,y) = [y*W+x]).
// Iv: level values I[] (glcm.cpp:388-420).  scr: 5*Ng doubles per slot.  fslots: 32 doubles per slot.
//
// Numerics: marginals and the x+y / |x-y| distributions are formed from exact integer count sums and
// divided by sum_p once (the reference sums the already divided elements, glcm.cpp:503-508, :523-525: same
// value to ~1e-16 relative); matrix-wide sums are lane-strided partial sums combined in a fixed order.
// reductions over the LW lanes that share a matrix; every lane of the group gets the result
template <int LW> __device__ __forceinline__ double slot_sum(double v) { return LW == 16 ? row16_sum(v) : wave_sum(v); }
template <int LW> __device__ __forceinline__ uint32_t slot_sum(uint32_t v) { return LW == 16 ? row16_sum(v) : (uint32_t)wave_sum_u64(v); }
template <int LW> __device__ __forceinline__ double slot_max(double v) { return LW == 16 ? row16_max(v) : wave_max_nonneg(v); }

// TAG only separates instantiations: a non-inlined copy inherits the register budget of its loosest caller, and
// glcm_features_kernel (128 VGPRs) must not loosen the copy roi_features_kernel's 80-VGPR builds call.
template <bool GS, int LW, int TAG = 0>
__device__ void glcm_features_rows(const uint32_t* Pslots, int n_slots, int Ng, const double* Iv, double* scr_base, int scr_stride,
                                   double soft_nan, double* fslots, int lane)
{
    const int NN = Ng * Ng;
    const int slot_raw = lane / LW, l16 = lane % LW;   // LW lanes per matrix: 16 (one DPP row per angle) or 64 (a wave per angle)
    const bool live = slot_raw < n_slots;
    const int slot = live ? slot_raw : 0;              // idle rows shadow slot 0 and never store
    const uint32_t* P = Pslots + slot * NN;
    double* scr = scr_base + slot * scr_stride;
    double* f = fslots + slot * 32;

    // sum_p (glcm.cpp:481-484): integer counts, exact in any order
    uint32_t csum = 0;
    for (int e = l16; e < NN; e += LW)
        csum += P[e];
    csum = slot_sum<LW>(csum);
    const bool empty = csum == 0;                      // glcm.cpp:260-295 -> soft NaN for this angle
    const double sum_p = empty ? 1.0 : (double)csum;
    // per-element probabilities use one reciprocal (<= 1 ulp from cnt / sum_p); marginals and the
    // exact-numerator features below keep true divisions
    const double inv_sum_p = fdiv(1.0, sum_p);

    double* pcol = scr;           // px[i] = sum_j xy(i,j)/sum_p   (glcm.cpp:523-525, :859-864)
    double* prow = scr + Ng;      // py[j] = sum_i xy(i,j)/sum_p
    double* Pxpy = scr + 2 * Ng;  // [2Ng]  glcm.cpp:503-508
    double* Pxmy = scr + 4 * Ng;  // [Ng]

    uint32_t dis_cnt = 0;                              // sum |r - c| * count = sum_k k * count(|x - y| = k): f_GLCM_DIS numerator, exact
    for (int i = l16; i < Ng; i += LW) {
        uint32_t cc = 0, rc = 0, dc = 0;
        for (int j = 0; j < Ng; j++) {
            cc += P[j * Ng + i];
            rc += P[i * Ng + j];
        }
        for (int x = i; x < Ng; x++) { // |x-y| = i
            dc += P[x * Ng + (x - i)];
            if (i > 0)
                dc += P[(x - i) * Ng + x];
        }
        dis_cnt += (uint32_t)i * dc;
        if (live) {
            pcol[i] = fdiv((double)cc, sum_p);
            prow[i] = fdiv((double)rc, sum_p);
            Pxmy[i] = fdiv((double)dc, sum_p);
        }
    }
    for (int k = l16; k < 2 * Ng; k += LW) {
        uint32_t c = 0;
        int x0 = k - (Ng - 1) > 0 ? k - (Ng - 1) : 0, x1 = k < Ng - 1 ? k : Ng - 1;
        for (int x = x0; x <= x1; x++)
            c += P[x * Ng + (k - x)];
        if (live)
            Pxpy[k] = fdiv((double)c, sum_p);
    }
    wav_sync<GS>();

    // by_row_mean (glcm.cpp:531-536)
    double brm = 0;
    for (int i = l16; i < Ng; i += LW)
        brm += pcol[i] * Iv[i];
    brm = slot_sum<LW>(brm);

    // ---- pass 1 over matrix elements -------------------------------------------------
    // HOM1 = sum p / (1 + |r - c|) and HOM2 = sum p / (1 + |r - c|^2) are the sums ID and IDM take over the |x - y|
    // distribution below (same terms grouped by k; <= 1e-15 relative apart), so the two divisions per cell are not repeated here.
    double asm_ = 0, contrast_n = 0, S_r = 0, S_c = 0, acor_n = 0, ent = 0, jmax = -1;
    RowCol rc1((uint32_t)l16, (uint32_t)LW, (uint32_t)Ng);   // (row, column) of the cell without a division per cell
    for (int e = l16; e < NN; e += LW, rc1.advance()) {
        const int r = (int)rc1.row, c = (int)rc1.col;
        double cnt = (double)P[e];
        double p = cnt * inv_sum_p;
        double ir = Iv[r], ic = Iv[c];
        asm_ += p * p;                               // f_asm :555 / f_energy :927-928
        double d = ir - ic;
        contrast_n += cnt * d * d;                   // f_contrast :579 (integer-exact)
        S_r += cnt * ir;                             // f_corr mr :601, f_var mean :662, JAVE :1144
        S_c += cnt * ic;                             // f_corr mc :608
        acor_n += cnt * ir * ic;                     // f_GLCM_ACOR :961
        ent += plogp(p, p);                          // f_entropy :734-735, JE :1160-1161, HXY :868
        jmax = p > jmax ? p : jmax;                  // f_GLCM_JMAX :1178-1179
    }
    asm_ = slot_sum<LW>(asm_); contrast_n = slot_sum<LW>(contrast_n); S_r = slot_sum<LW>(S_r); S_c = slot_sum<LW>(S_c);
    acor_n = slot_sum<LW>(acor_n); ent = slot_sum<LW>(ent); jmax = slot_max<LW>(jmax);
    const double dis_n = (double)slot_sum<LW>(dis_cnt);   // f_GLCM_DIS :1052
    const double mr = fdiv(S_r, sum_p), mc = fdiv(S_c, sum_p); // mr == f_var's mean == JAVE (exact numerators)
    if (live && l16 == 0) { // results leave the registers as soon as they exist
        f[G_ASM] = asm_;
        f[G_ENERGY] = asm_;
        f[G_CONTRAST] = fdiv(contrast_n, sum_p);
        f[G_ACOR] = fdiv(acor_n, sum_p);
        f[G_ENTROPY] = -ent;
        f[G_JE] = -ent;
        f[G_DIS] = fdiv(dis_n, sum_p);
        f[G_JMAX] = jmax;
        f[G_JAVE] = mr;
    }

    // ---- pass 2: central quantities ---------------------------------------------------
    double s2r = 0, s2c = 0, tmp1 = 0, var_n = 0, cprom = 0, cshade = 0, ctend = 0, jvar = 0, hxy1 = 0, hxy2 = 0;
    RowCol rc2((uint32_t)l16, (uint32_t)LW, (uint32_t)Ng);
    for (int e = l16; e < NN; e += LW, rc2.advance()) {
        const int r = (int)rc2.row, c = (int)rc2.col;
        double cnt = (double)P[e];
        double p = cnt * inv_sum_p;
        double ir = Iv[r], ic = Iv[c];
        double dr = ir - mr, dc = ic - mc;
        s2r += p * dr * dr;                           // f_corr :617
        s2c += p * dc * dc;                           // :626
        tmp1 += dr * dc * p;                          // :633
        var_n += dr * dr * cnt;                       // f_var :672
        double m = ir + ic - brm - brm;               // CLUPROM :985, CLUSHADE :1007, CLUTEND :1034
        double m2 = m * m;
        cprom += m2 * m2 * p;
        cshade += m2 * m * p;
        ctend += m2 * p;
        double dj = (double)(c + 1) - mr;             // f_GLCM_JVAR :1196-1199 (x = column, +1 index)
        jvar += dj * dj * p;
        double pp = pcol[c] * prow[r];                // px[i]*py[j], i = column, j = row (:869, :909)
        double lg = (double)fast_log2f(pp + 0.000000001);
        hxy1 += p * lg;
        hxy2 += pp * lg;
    }
    s2r = slot_sum<LW>(s2r); s2c = slot_sum<LW>(s2c); tmp1 = slot_sum<LW>(tmp1); var_n = slot_sum<LW>(var_n);
    cprom = slot_sum<LW>(cprom); cshade = slot_sum<LW>(cshade); ctend = slot_sum<LW>(ctend); jvar = slot_sum<LW>(jvar);
    hxy1 = slot_sum<LW>(hxy1); hxy2 = slot_sum<LW>(hxy2);
    if (live && l16 == 0) {
        f[G_VARIANCE] = fdiv(var_n, sum_p);
        f[G_CLUPROM] = cprom;
        f[G_CLUSHADE] = cshade;
        f[G_CLUTEND] = ctend;
        f[G_SUMVARIANCE] = ctend;                     // glcm.cpp:323-326
        f[G_JVAR] = jvar;
        double denom = sqrt(s2r) * sqrt(s2c);         // f_corr tail, glcm.cpp:619-643
        f[G_CORRELATION] = !(denom > 0.0) ? soft_nan : tmp1 / denom;
        f[G_INFOMEAS2] = sqrt(fabs(1 - exp(-2 * (-hxy2 + ent)))); // glcm.cpp:913 (HXY = ent)
    }

    // ---- 1-D features over p_{x-y} (k < Ng) and p_{x+y} (k < 2Ng), lanes over k ----------
    // kValuesDiff[k] = |I[Ng-1] - I[Ng-1-k]| and kValuesSum[k] = I[min(k,Ng-1)] + I[k - min(k,Ng-1)]
    // are the last pairs calculatePxpmy writes (glcm.cpp:511-512).
    double idm = 0, dent = 0, idmn = 0, id = 0, idn = 0, iv = 0, hx = 0, davg = 0;
    const double Ng2 = (double)Ng * (double)Ng;
    for (int k = l16; k < Ng; k += LW) {
        double q = Pxmy[k];
        double kval = k == 0 ? 0.0 : fabs(Iv[Ng - 1] - Iv[Ng - 1 - k]);
        idm += fdiv(q, (double)(1 + (k * k)));                   // f_idm :685-687
        if (q != 0)
            dent += plogp(q, q);                                 // f_dentropy :778-781
        idmn += fdiv(q, 1.0 + fdiv((double)k * (double)k, Ng2)); // :1083-1084
        id += fdiv(q, 1.0 + (double)k);                          // :1096-1097
        idn += fdiv(q, 1.0 + fdiv((double)k, (double)Ng));       // :1110-1111
        if (k >= 1)
            iv += q / (kval * kval);                             // :1123-1128
        hx += plogp(pcol[k], pcol[k]);                           // :873-874
        davg += kval * q;                                        // f_difference_avg :791-792
    }
    idm = slot_sum<LW>(idm); dent = slot_sum<LW>(dent); idmn = slot_sum<LW>(idmn); id = slot_sum<LW>(id);
    idn = slot_sum<LW>(idn); iv = slot_sum<LW>(iv); hx = slot_sum<LW>(hx); davg = slot_sum<LW>(davg);
    const double diffAvg = davg;
    double savg = 0, sent = 0, dv = 0;
    for (int k = l16; k < 2 * Ng - 1; k += LW) {
        double q = Pxpy[k];
        int x = k < Ng - 1 ? k : Ng - 1;
        savg += (Iv[x] + Iv[k - x]) * q;                         // f_savg :700-701
        sent += plogp(q, q);                                     // f_sentropy :712-716
    }
    for (int k = l16; k < Ng; k += LW) {
        // f_dvar (glcm.cpp:742-766): var[k] receives the same term Ng times, total / Ng
        double dk = (double)k - diffAvg;
        double t = dk * dk * Pxmy[k], a = 0;
        for (int x = 0; x < Ng; x++)
            a += t;
        dv += a;
    }
    savg = slot_sum<LW>(savg); sent = slot_sum<LW>(sent); dv = slot_sum<LW>(dv);
    if (live && l16 == 0) {
        f[G_IDM] = idm;
        f[G_HOM2] = idm;                              // f_GLCM_HOM2 :1069 == f_idm over p_{x-y}
        f[G_HOM1] = id;                               // f_homogeneity :942 == f_GLCM_ID over p_{x-y}
        f[G_SUMAVERAGE] = savg;
        f[G_SUMENTROPY] = -sent;
        f[G_DIFENTRO] = -dent;
        f[G_DIFAVE] = diffAvg;
        f[G_DIFVAR] = fdiv(dv, (double)Ng);
        f[G_IDMN] = idmn;
        f[G_ID] = id;
        f[G_IDN] = idn;
        f[G_IV] = iv;
        double r1 = (ent - hxy1) / hx;                // f_info_meas_corr1, glcm.cpp:880-883
        f[G_INFOMEAS1] = isfinite(r1) ? r1 : soft_nan;
    }
    wav_sync<GS>();
    if (live && empty && l16 < 2) {                   // blank matrix: all 30 values = soft NaN (after the stores above)
        for (int k = l16; k < kGlcmAngled; k += 2)
            f[k] = soft_nan;
    }
    wav_sync<GS>();
}

} // namespace nyxhip
