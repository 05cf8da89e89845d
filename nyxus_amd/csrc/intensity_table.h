// intensity_table.h -- the 36 first-order features of an ROI from its intensity HISTOGRAM, by one 256-thread workgroup.
//
// Every first-order feature of the reference is a function of the multiset of intensities
// (/root/reference/src/nyx/features/intensity.cpp:57-192, histogram.h:27-309, moments.h:48-109): given the number of pixels
// at or below every occurring value, nothing else about the ROI is needed.  roi_large.hip arrives at a DENSE table over
// [min, max], counted by several workgroups per ROI with atomics (entry i = value min + i), and finishes through this routine.
// (Round 4 also tried it behind an LDS radix sort + run-length pass for LDS-sized ROIs of 16-bit data -- a COMPRESSED table of
// the distinct values: 4364 vector instructions per wave against 5403 for the sort engine inside roi_features.hip, 32.5 ns per
// 2821-pixel ROI + 10 ns for the GLCM-only build = the 43 ns it was to replace; not kept.  TAB keeps the interface general.)
// TAB describes the table:  m entries in ascending value order;  off(i) = value of entry i minus the ROI minimum;
// cum(i) = number of pixels with a value <= that of entry i (cum(m - 1) = n);  first_ge(d) = first entry whose offset is >= d
// (d: 64 bits; m when there is none).
//
// Integer-exact columns (MIN, MAX, RANGE, MEAN, ENERGY, INTEGRATED_INTENSITY, RMS, MEDIAN, MODE, the percentiles, IQR, QCOD,
// ROBUST_MEAN, PIU) follow the reference's expressions operation by operation; the floating-point sums run in a fixed order.
// Built with -ffp-contract=off (device_math.h).
#pragma once
#include <hip/hip_runtime.h>
#include "device_math.h"
#include "roi_kernel.h"
#include "glcm_rows.h"

namespace nyxhip {

struct IntensityScratch {            // LDS of the calling kernel (static or dynamic)
    double* x;                       // [32]   cross-wave sums
    unsigned long long* u;           // [8]
    double* stat;                    // [8]
    double* pq;                      // [8]
    uint32_t* w;                     // [16]
    uint32_t* lb100;                 // [104]  lower bounds of the 100 percentile bins
    uint32_t* lbc;                   // [n_hist + 8]  lower bounds of the custom bins
};

// fixed-order workgroup sums (wave DPP tree, then the four wave partials in wave order); every thread gets the totals
template <int N>
__device__ __forceinline__ void wg4_sum(double (&v)[N], double* s_x, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < N; k++) s_x[wave * 8 + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = ((s_x[k] + s_x[8 + k]) + s_x[16 + k]) + s_x[24 + k];
}
__device__ __forceinline__ unsigned long long wg4_sum_u64(unsigned long long v, unsigned long long* s_x, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    v = wave_sum_u64(v);
    __syncthreads();
    if (lane == 0) s_x[wave] = v;
    __syncthreads();
    return s_x[0] + s_x[1] + s_x[2] + s_x[3];
}

// the sums over the table's entries: entry i stands for count(i) pixels of value vmin + off(i)
template <class TAB>
struct TableSums {
    const TAB& tab; uint32_t vmin;
    __device__ __forceinline__ uint32_t count_at(uint32_t i) const { return tab.cum(i) - (i ? tab.cum(i - 1) : 0u); }
    __device__ __forceinline__ void central(double mean, bool blank, double (&acc)[6], uint32_t& mode_off, const IntensityScratch& S, int tid) const
    {
        const int lane = tid & 63, wave = tid >> 6;
        uint32_t best_c = 0, best_i = 0;
        for (uint32_t i = tid; i < tab.m; i += 256) {
            const uint32_t c = count_at(i);
            if (c > best_c) { best_c = c; best_i = i; }
            if (c && !blank) {
                const double cd = (double)c, d = (double)(vmin + tab.off(i)) - mean, d2 = d * d;
                acc[0] += cd * fabs(d);
                acc[1] += cd * d2;
                acc[2] += cd * (d2 * d);
                acc[3] += cd * (d2 * d2);
                acc[4] += cd * (d2 * d2 * d);
                acc[5] += cd * (d2 * d2 * d2);
            }
        }
        // mode: (count desc, index asc) over the workgroup
        const uint32_t mc_w = wave_max_u32(best_c);
        const uint32_t cand = best_c == mc_w ? best_i : 0xFFFFFFFFu;
        const uint32_t bi_w = ~wave_max_u32(~cand);
        __syncthreads();
        if (lane == 0) { S.w[4 + wave] = mc_w; S.w[8 + wave] = bi_w; }
        __syncthreads();
        uint32_t mc = 0, mi = 0;
        for (int wv = 0; wv < 4; wv++) {
            const uint32_t c = S.w[4 + wv], i = S.w[8 + wv];
            if (c > mc || (c == mc && i < mi)) { mc = c; mi = i; }
        }
        mode_off = tab.off(mi);
    }
    __device__ __forceinline__ void robust(uint32_t lo_off, uint32_t hi_off, bool some, double median, unsigned long long& sx, double& medad, int tid) const
    {
        for (uint32_t i = tid; i < tab.m; i += 256) {
            const uint32_t c = count_at(i);
            if (!c) continue;
            const uint32_t off = tab.off(i), v = vmin + off;
            if (some && off >= lo_off && off <= hi_off) sx += (unsigned long long)c * v;
            medad += (double)c * fabs((double)v - median);
        }
    }
    __device__ __forceinline__ void spread(uint32_t lo_off, uint32_t hi_off, double mean1090, double& ad, int tid) const
    {
        const uint32_t i0 = tab.first_ge((uint64_t)lo_off), i1 = tab.first_ge((uint64_t)hi_off + 1u);
        for (uint32_t i = i0 + tid; i < i1; i += 256) {
            const uint32_t c = count_at(i);
            if (c) ad += (double)c * fabs((double)(vmin + tab.off(i)) - mean1090);
        }
    }
};

// n pixels, extrema vmin / vmax, tot = sum of the intensities, totsq = sum of the (32-bit wrapping) squares, both exact.
// slide: slide_max - slide_min, or NaN when the slide extrema are not given.  o: the 36 columns (zeroed by the caller).
// Every thread of the 256-thread workgroup calls this; barriers inside.
// SUMS supplies the sums that run over the pixels' VALUES -- over the table's entries (TableSums below: entry i stands for
// count(i) pixels) or over the ROI's values themselves when the table is not worth walking (roi_wide.hip: 2821 keys against a
// 65536-entry domain):
//   central(mean, acc[6], mode_off)   per-thread partial sums of |d|, d^2 .. d^6 over the pixels (d = value - mean); mode_off: the
//                                     offset of the most frequent value (smallest on ties), the same on every thread
//   robust(lo_off, hi_off, some, median, sx, medad)   per-thread partials: exact sum of the values whose offset lies in
//                                     [lo_off, hi_off] (when `some`), sum of |value - median|
//   spread(lo_off, hi_off, mean1090)  per-thread partial of sum |value - mean1090| over the same pixels
template <class TAB, class SUMS>
__device__ __forceinline__ void intensity_from_table(const TAB& tab, const SUMS& sums, uint32_t n, uint32_t vmin, uint32_t vmax, double tot, double totsq,
                                                     bool have_slide, double slide_range, uint32_t n_hist, double* o, const IntensityScratch& S, int tid)
{
    constexpr int BS = 256, NW = 4;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t range = vmax - vmin, m = tab.m;
    const double dn = (double)n;
    const double mean = tot / dn;
    const bool blank = vmin == 0 && vmax == 0;                                // intensity.cpp:121-122
    if (tid == 0) {
        o[I_MIN] = (double)vmin;                                               // intensity.cpp:67-69
        o[I_MAX] = (double)vmax;
        o[I_RANGE] = (double)vmax - (double)vmin;
        if (have_slide)                                                        // intensity.cpp:72-77
            o[I_COVERED_IMAGE_INTENSITY_RANGE] = (double)(vmax - vmin) / slide_range;
        o[I_MEAN] = mean;                                                      // intensity.cpp:95-99
        o[I_ENERGY] = totsq;
        o[I_ROOT_MEAN_SQUARED] = sqrt(totsq / dn);
        o[I_INTEGRATED_INTENSITY] = tot;
        if (!blank)
            o[I_UNIFORMITY_PIU] = (1.0 - (double)(vmax - vmin) / (double)(uint32_t)(vmax + vmin)) * 100.0;   // :162
    }
    // ---- sweep 1: central sums (intensity.cpp:102-109, :177-183; M2..M4 of moments.h:53-74 equal the plain central sums) and
    // the mode (largest count, smallest value on ties: histogram.h:289-309)
    double acc[6] = {0, 0, 0, 0, 0, 0};
    uint32_t mode_off = 0;
    sums.central(mean, blank, acc, mode_off, S, tid);
    wg4_sum<6>(acc, S.x, tid);
    if (tid == 0) {
        if (!blank) o[I_MODE] = (double)(vmin + mode_off);
        // everything that depends only on the central sums (intensity.cpp:110-118, :166-191)
        o[I_MEAN_ABSOLUTE_DEVIATION] = acc[0] / dn;
        const double variance = dn > 1 ? acc[1] / (dn - 1) : 0.0, variance_b = dn > 1 ? acc[1] / dn : 0.0;
        const double sd = sqrt(variance);
        o[I_VARIANCE] = variance;
        o[I_VARIANCE_BIASED] = variance_b;
        o[I_STANDARD_DEVIATION] = sd;
        o[I_STANDARD_DEVIATION_BIASED] = sqrt(variance_b);
        o[I_COV] = sd / mean;
        o[I_STANDARD_ERROR] = sd / sqrt(dn);
        if (!blank) {
            const double M2 = acc[1], M3 = acc[2], M4 = acc[3];               // moments.h:79-109
            if (M2 != 0.0) {
                const double kurt = n > 4 ? (dn * M4) / (M2 * M2) : 0.0;
                o[I_SKEWNESS] = n > 3 ? (sqrt(dn) * M3) / (M2 * sqrt(M2)) : 0.0;
                o[I_KURTOSIS] = kurt;
                o[I_EXCESS_KURTOSIS] = n > 4 ? kurt - 3 : 0.0;
            }
            const double sd2 = sd * sd, d5 = dn * (sd2 * sd2 * sd), d6 = dn * (sd2 * sd2 * sd2);   // intensity.cpp:186-191
            o[I_HYPERSKEWNESS] = d5 == 0. ? 0. : acc[4] / d5;
            o[I_HYPERFLATNESS] = d6 == 0. ? 0. : acc[5] / d6;
        }
    }
    if (blank) { __syncthreads(); return; }
    // ---- histogram bin populations (histogram.h:55-78): lower bounds of the 100 percentile bins and the n custom bins -- the bin
    // index is monotone in the value, so a bin's population is a difference of two cumulative counts
    const uint32_t nb = n_hist;
    const double binW100 = (double)range / 100.;
    for (uint32_t t = tid; t < 100 + nb; t += BS) {
        const bool is100 = t < 100;
        const uint32_t b = is100 ? t : t - 100;
        auto bin_of = [=](uint32_t dd) -> uint32_t {
            if (is100) {
                const double realIdx = (double)dd / binW100;                  // (h - minVal) / binW100, histogram.h:57-60
                return (realIdx != realIdx) ? 0u : (uint32_t)(int)realIdx;
            }
            return to_grayscale(vmin + dd, vmin, range, nb);
        };
        // smallest offset d in [0, range + 1] whose bin index reaches b: start from the real-valued boundary and settle with the
        // exact (reference) bin function; the bin's lower bound is the number of pixels below that offset
        uint64_t d;
        if (range < 65536u) {
            // The real-valued boundary is P = b * (bin width).  Unless P lies within 1e-6 of an integer, the answer is floor(P) + 1
            // with certainty: the reference's bin function (one or two fp64 roundings of a value below 2^16) cannot move a value
            // that is >= 1e-6 / range away from the boundary across it.  Next to an integer m the exact bin function decides
            // between m and m + 1: one evaluation, no search (roi_features.hip takes its table bounds the same way).
            const double Wl = is100 ? binW100 : (double)range / (double)nb;
            const double P = (double)b * Wl;
            const uint32_t mi = (uint32_t)(P + 0.5);
            d = (uint64_t)(uint32_t)P + 1;
            if (b == 0) d = 0;
            else if (fabs(P - (double)mi) < 1e-6) d = bin_of(mi) >= b ? mi : mi + 1;
        } else {
            const double edge = is100 ? (double)b * binW100 : (double)b * (double)range / (double)nb;
            d = !(edge < (double)range + 1.0) ? (uint64_t)range + 1 : (uint64_t)edge;          // (64 bits: range + 1 of a full 32-bit range)
            while (d > 0 && bin_of((uint32_t)(d - 1)) >= b) d--;
            while (d <= range && bin_of((uint32_t)d) < b) d++;
        }
        const uint32_t idx = tab.first_ge(d);
        const uint32_t lo = idx > 0 ? tab.cum(idx - 1) : 0u;
        if (is100) S.lb100[b] = lo; else S.lbc[b] = lo;
    }
    __syncthreads();
    if (wave == 0) {
        // percentiles P01, P10, P25, P75, P90, P99 (histogram.h:214-243): the LAST bin i with runSum_i <= cnt <= runSum_i + bins_i
        // wins (every matching bin overwrites); runSum_i is the lower bound of bin i.  Lanes test bins i and i + 64.
        const int i0 = lane, i1 = lane + 64;
        const uint32_t r0 = S.lb100[i0], e0 = (i0 < 99 ? S.lb100[i0 + 1] : n);
        const uint32_t r1 = i1 < 100 ? S.lb100[i1] : 0u, e1 = i1 < 100 ? (i1 < 99 ? S.lb100[i1 + 1] : n) : 0u;
        int mywin = -1;
        double mycnt = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const double frac = q == 0 ? 0.01 : q == 1 ? 0.1 : q == 2 ? 0.25 : q == 3 ? 0.75 : q == 4 ? 0.9 : 0.99;
            const double cnt_p = dn * frac;
            const bool m0 = (double)r0 <= cnt_p && cnt_p <= (double)e0;
            const bool m1 = i1 < 100 && (double)r1 <= cnt_p && cnt_p <= (double)e1;
            const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1);
            const int win = b1 ? 64 + (63 - __clzll((long long)b1)) : (b0 ? 63 - __clzll((long long)b0) : -1);
            if (lane == q) { mywin = win; mycnt = cnt_p; }
        }
        double pv = 0;
        if (mywin >= 0) {
            const uint32_t rs = S.lb100[mywin], bi = (mywin < 99 ? S.lb100[mywin + 1] : n) - rs;
            pv = (mycnt - (double)rs) * binW100 / (double)bi + (double)vmin + binW100 * (double)mywin;
        }
        if (lane < 6) S.pq[lane] = pv;
        wav_sync<false>();
        if (lane == 0) {
            const double* const pq = S.pq;
            o[I_P01] = pq[0]; o[I_P10] = pq[1]; o[I_P25] = pq[2]; o[I_P75] = pq[3]; o[I_P90] = pq[4]; o[I_P99] = pq[5];
            o[I_QCOD] = (pq[3] - pq[2]) / (pq[3] + pq[2]);
            o[I_INTERQUARTILE_RANGE] = pq[3] - pq[2];
            S.stat[0] = pq[1];
            S.stat[1] = pq[4];
        }
    }
    if (wave == 1) {
        // entropy / uniformity over the n + 1 slots (histogram.h:145-151): slot n is empty
        double e = 0, u = 0;
        for (uint32_t k = lane; k < nb; k += 64) {
            const uint32_t ck = (k < nb - 1 ? S.lbc[k + 1] : n) - S.lbc[k];
            const double p = (double)ck / dn;
            e += p * log2(p + 2.2e-16);
            u += p * p;
        }
        e = wave_sum(e);
        u = wave_sum(u);
        if (lane == 0) { o[I_ENTROPY] = -e; o[I_UNIFORMITY] = u; }
    }
    if (wave == 2) {
        // median (histogram.h:268-287): order statistics n/2 and n/2 - 1 = smallest entry with cum > k, by a 64-way search
        auto kth = [&](uint32_t k) -> uint32_t {
            uint32_t lo = 0, span = m;                                         // the answer lies in [lo, lo + span)
            while (span > 1) {
                const uint32_t B = (span + 63) >> 6;
                const uint64_t i64 = (uint64_t)lo + (uint64_t)((uint32_t)lane + 1) * B - 1;   // last position of this lane's block
                const uint32_t i = i64 > m - 1 ? m - 1 : (uint32_t)i64;
                const unsigned long long hit = __ballot(tab.cum(i) > k);
                const uint32_t first = hit ? (uint32_t)__builtin_ctzll(hit) : 63u;
                const uint64_t lo64 = (uint64_t)lo + (uint64_t)first * B;
                lo = lo64 > m - 1 ? m - 1 : (uint32_t)lo64;                    // (a table that holds fewer than n values -- intensities outside the stated
                span = (uint64_t)lo + B > (uint64_t)m ? m - lo : B;            //  [min, max] -- must not walk off the table: the search always ends)
            }
            return lo;
        };
        const uint32_t hi_v = vmin + tab.off(kth(n / 2)), lo_v = vmin + tab.off(kth(n / 2 ? n / 2 - 1 : 0));
        if (lane == 0) {
            const double median = (n & 1) ? (double)hi_v : (double)(uint32_t)(hi_v + lo_v) / 2.0;
            o[I_MEDIAN] = median;
            S.stat[2] = median;
        }
    }
    __syncthreads();
    // ---- robust statistics over [p10, p90] (intensity.cpp:139-149, histogram.h:90-112) and the median deviation (:156-159)
    const double p10 = S.stat[0], p90 = S.stat[1], median = S.stat[2];
    uint32_t i0 = 1, i1 = 0;                                                   // entries inside the bounds: empty unless the bounds say otherwise (NaN: empty)
    if (p10 <= p90 && p90 >= (double)vmin && p10 <= (double)vmax) {
        const double cl = ceil(p10), fl = floor(p90);
        const uint32_t lo_v = cl <= (double)vmin ? vmin : (uint32_t)cl, hi_v = fl >= (double)vmax ? vmax : (uint32_t)fl;
        if (lo_v <= hi_v) {
            const uint32_t a = tab.first_ge((uint64_t)(lo_v - vmin)), z = tab.first_ge((uint64_t)(hi_v - vmin) + 1u);   // entries [a, z)
            if (a < z) { i0 = a; i1 = z - 1; }
        }
    }
    const bool some = i0 <= i1;
    const uint32_t K = some ? tab.cum(i1) - (i0 ? tab.cum(i0 - 1) : 0u) : 0u;
    const uint32_t lo_off = some ? tab.off(i0) : 1u, hi_off = some ? tab.off(i1) : 0u;
    unsigned long long sx = 0;                                                 // exact integer sum of the values inside the bounds
    double medad[1] = {0};
    sums.robust(lo_off, hi_off, some, median, sx, medad[0], tid);
    sx = wg4_sum_u64(sx, S.u, tid);
    wg4_sum<1>(medad, S.x, tid);
    const double mean1090 = K ? (double)sx / (double)K : 0.0;
    double ad[1] = {0};
    if (K) sums.spread(lo_off, hi_off, mean1090, ad[0], tid);
    wg4_sum<1>(ad, S.x, tid);
    if (tid == 0) {
        o[I_ROBUST_MEAN] = mean1090;
        o[I_ROBUST_MEAN_ABSOLUTE_DEVIATION] = K ? ad[0] / (double)K : 0.0;
        o[I_MEDIAN_ABSOLUTE_DEVIATION] = medad[0] / dn;
    }
    __syncthreads();
}

} // namespace nyxhip
