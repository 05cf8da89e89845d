// roi_small.hip -- INTENSITY + GLCM counts of the smallest size class (<= 256 pixels, box sides <= 32): a WAVE per ROI (round 6).
//
// The reference gives every ROI, whatever its size, to one worker thread (/root/reference/src/nyx/parallel.h:34-41;
// features/intensity.cpp:57-192, histogram.h:27-309, moments.h:48-109; glcm.cpp:343-485).  roi_features_kernel spends a 256-thread
// workgroup and ~1250 vector instructions per wave on an ROI whatever it holds -- a 49-pixel nucleus costs what a 2821-pixel disk costs
// minus a third (7.4 against 12.4 ns): every phase has a fixed part (the 4096-entry counting table is scanned even for 49 values, every
// exchange is a workgroup barrier) and a workgroup costs 1.1 ns of dispatch alone.  Here everything is proportional to the ROI:
//   * four ROIs per workgroup, one per wave, no workgroup barrier anywhere (a wave's LDS block is its own);
//   * the values (<= 4 per lane) are SORTED (bitonic network in registers: one key per lane up to 64 pixels, four packed 16-bit keys per
//     lane beyond) instead of counted into a table over [min, max]:
//     the sorted array IS the cumulative table -- "pixels below offset d" is a lower bound, the median two reads, the mode the
//     longest run;
//   * the 100 + n histogram-bin lower bounds (histogram.h:55-78) are three binary searches per lane; percentiles, entropy, robust
//     statistics follow intensity_table.h (intensity_from_table) line by line, wave sums instead of workgroup sums;
//   * GLCM (matlab binning, <= 16 levels): the plane and the four Ng x Ng matrices in the wave's block, counts exported for
//     glcm_features_kernel8 / glcm_features_kernel in the layout roi_features_kernel exports (angle-major, Ng^2 words per angle);
//   * GLCM at 17..64 levels (the reference's default depth): the features themselves, from the <= 255 PAIRS of an angle -- marginal
//     counts by atomics per pair, per-cell sums as sums over pairs, the rest through glcm_w64.h (the workgroup kernel's own tail).
// Integer-exact columns follow the reference operation by operation (MIN, MAX, RANGE, MEAN, ENERGY, INTEGRATED_INTENSITY, RMS, MEDIAN,
// MODE, percentiles, IQR, QCOD, ROBUST_MEAN, PIU); the floating-point sums run in a fixed order that is a function of the ROI.
// Which ROIs come here is a function of the ROI alone: roi_class(...) == 0 (the launcher routes that class; anybody else returns or
// -- in a launch that was promised class 0 only -- raises the error flag).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "device_math.h"
#include "roi_kernel.h"
#include "glcm_rows.h"
#include "glcm_w64.h"
#include "launch_util.h"
#include "../../include/nyxhip.h"

namespace nyxhip {
namespace {

#ifdef NYX_SMALL_EXIT   // diagnostic builds (tools/small_exit_libs.sh): the per-ROI routine ends after phase NYX_SMALL_EXIT; results are wrong by design
#define SMALL_EXIT(k) do { if ((k) == NYX_SMALL_EXIT) return; } while (0)
#else
#define SMALL_EXIT(k) do { } while (0)
#endif
constexpr int kSmallPx = 256;                // kClassPx[0]

// per-wave LDS block (bytes): sorted offsets u32[256] | lb100 u32[104] | lbc u32[nb + 8] | pq double[8] | plane u8[(32 + 2) x 33 + 6] | P u32[na * Ng * Ng]
// (the plane carries a zero column on either side and a zero row below: the four neighbours of the usual request need no bounds test)
struct SmallBlock { uint32_t S, lb100, lbc, pq, plane, P, marg, pp, fs, pos, total; };
// glcm: 0 none | 1: P = u32[na * Ng * Ng] counts for the feature launch | 2 (17..64 levels, features from the pairs): P = u8[64 x 64] counts of ONE angle,
// marg = u32[64 + 64 + 64 + 128] row / column / difference / sum counts, pp = double[64 + 64] row / column probabilities, fs = double[4][64] an
// angle's f | sm block (glcm_w64.h), pos = u16[256] plane address of pixel i (window mode ranks pixels by ballot).  10 160 bytes: four
// workgroups of four waves per CU.
__host__ __device__ inline SmallBlock small_block(uint32_t nb, uint32_t na, uint32_t ng, bool do_int, int glcm)
{
    SmallBlock b;
    uint32_t o = 0;
    b.S = o; o += do_int ? 4u * kSmallPx : 0u;
    b.lb100 = o; o += do_int ? 4u * 104 : 0u;
    b.lbc = o; o += do_int ? 4u * (nb + 8) : 0u;
    o = (o + 7u) & ~7u;
    b.pq = o; o += 8u * 8;
    b.plane = o; o += glcm ? 34 * 33 + 6 : 0;
    o = (o + 15u) & ~15u;
    b.P = o; o += glcm == 1 ? 4u * na * ng * ng : glcm == 2 ? 64u * 64u : 0u;
    b.marg = o; o += glcm == 2 ? 4u * 320u : 0u;
    b.pp = o; o += glcm == 2 ? 8u * 128u : 0u;
    b.fs = o; o += glcm == 2 ? 8u * 64u * kMaxAngles : 0u;
    b.pos = o; o += glcm == 2 ? 2u * kSmallPx : 0u;
    b.total = (o + 15u) & ~15u;
    return b;
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* S, uint32_t n, uint32_t key)   // first i in [0, n] with S[i] >= key
{
    uint32_t lo = 0, len = n;
    while (len > 0) {
        const uint32_t half = len >> 1, mid = lo + half;
        if (S[mid] < key) { lo = mid + 1; len -= half + 1; } else len = half;
    }
    return lo;
}

// Bitonic sort of P2 (a power of two, 64 <= P2 <= 64 NV) keys held NV per lane, position e = NV * lane + k, ascending in e.
// P2 < 64 NV: only lanes below P2 / NV take part meaningfully -- the caller loads the padding key into every position >= n, and the
// positions >= P2 form sorted-by-construction padding that the network never mixes with the first P2 (bit tests only reach below P2).
template <int NV>
__device__ __forceinline__ void small_sort(uint32_t (&x)[NV], int lane, uint32_t P2)
{
    constexpr int SH = NV == 4 ? 2 : 0;                   // log2(NV)
    for (uint32_t kk = 2; kk <= P2; kk <<= 1)
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            if (j >= (uint32_t)NV) {
                const int m = (int)(j >> SH);             // partner lane = lane ^ m, same register
                const bool lower = ((uint32_t)lane & (uint32_t)m) == 0;
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const uint32_t y = (uint32_t)__shfl_xor((int)x[k], m, 64);
                    const bool up = ((((uint32_t)lane << SH) | (uint32_t)k) & kk) == 0;
                    const uint32_t lo = x[k] < y ? x[k] : y, hi = x[k] < y ? y : x[k];
                    x[k] = (lower == up) ? lo : hi;
                }
            } else if constexpr (NV == 4) {
                // distance 1 or 2 inside the lane's four keys (compile-time register pairs: a run-time register index would live in scratch)
                auto cx = [&](uint32_t& a, uint32_t& b, int k) {
                    const bool up = ((((uint32_t)lane << SH) | (uint32_t)k) & kk) == 0;
                    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
                    a = up ? lo : hi;
                    b = up ? hi : lo;
                };
                if (j == 2) { cx(x[0], x[2], 0); cx(x[1], x[3], 1); }
                else { cx(x[0], x[1], 0); cx(x[2], x[3], 2); }
            }
        }
}


// The same network for four keys per lane held as TWO registers of packed 16-bit keys (offsets from the ROI minimum are below 2^14 in this
// class; padding 0xFFFF): r0 = k0 | k1 << 16, r1 = k2 | k3 << 16, position e = 4 lane + k.  A lane-crossing exchange is one shuffle and a
// v_pk_min_u16 / v_pk_max_u16 pair per register (both keys of a register share direction and side once kk >= 8), distance 2 is
// element-wise between the two registers, distance 1 a half-swap inside a register: ~300 vector instructions for 256 keys instead of ~750.
typedef unsigned short small_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmin16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(small_us2, a), __builtin_bit_cast(small_us2, b))); }
__device__ __forceinline__ uint32_t pkmax16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(small_us2, a), __builtin_bit_cast(small_us2, b))); }
__device__ __forceinline__ void small_sort4_packed(uint32_t& r0, uint32_t& r1, int lane, uint32_t P2)
{
    // distance 1 inside a register: (lo, hi) -> (min, max) when `up`, (max, min) otherwise
    auto cx1 = [](uint32_t r, bool up) -> uint32_t {
        const uint32_t sw = (r >> 16) | (r << 16);
        const uint32_t mn = pkmin16(r, sw), mx = pkmax16(r, sw);           // both halves hold the min / the max
        const uint32_t a = up ? mn : mx, b = up ? mx : mn;
        return (a & 0xFFFFu) | (b & 0xFFFF0000u);
    };
    // kk = 2: positions 0, 1 ascending, positions 2, 3 descending
    r0 = cx1(r0, true); r1 = cx1(r1, false);
    for (uint32_t kk = 4; kk <= P2; kk <<= 1) {
        const bool up = (((uint32_t)lane << 2) & kk) == 0;                 // (e & kk) == 0: a fact of the lane for kk >= 4
        for (uint32_t j = kk >> 1; j >= 4; j >>= 1) {
            const int m = (int)(j >> 2);                                   // partner lane = lane ^ m, same key slot
            const bool takemin = (((uint32_t)lane & (uint32_t)m) == 0) == up;
            const uint32_t y0 = (uint32_t)__shfl_xor((int)r0, m, 64), y1 = (uint32_t)__shfl_xor((int)r1, m, 64);
            r0 = takemin ? pkmin16(r0, y0) : pkmax16(r0, y0);
            r1 = takemin ? pkmin16(r1, y1) : pkmax16(r1, y1);
        }
        {   // distance 2: key k against key k + 2 = register 0 against register 1, element-wise
            const uint32_t mn = pkmin16(r0, r1), mx = pkmax16(r0, r1);
            r0 = up ? mn : mx; r1 = up ? mx : mn;
        }
        r0 = cx1(r0, up); r1 = cx1(r1, up);                                // distance 1
    }
}

template <bool DO_INT, int GLCM>      // GLCM: 0 none, 1 counts of <= 16-level matrices for the feature launch, 2 features of 17..64-level matrices from the pairs
__device__ __forceinline__ void small_one(const RoiArgs& A, const uint64_t slot, unsigned char* const blk, const int lane, const uint32_t promised)
{
    uint64_t roi;
    if (!roi_of_slot(A.sp, slot, A.n_roi, roi)) return;
    const uint64_t off = A.px_offset[roi];
    const uint32_t n = (uint32_t)(A.px_offset[roi + 1] - off);
    if (n > (uint32_t)kSmallPx && !promised) return;                       // (a filtered launch over a batch of larger ROIs: two loads and out)
    const uint32_t w = A.bbox_w[roi], h = A.bbox_h[roi];
    const uint32_t vmin = A.min_inten[roi], vmax = A.max_inten[roi];
    const uint32_t range = vmax - vmin;
    double* const out_row = A.out + roi * A.ld;
    if (roi_class(n, w, h, range) != 0) {
        // not of this launch's class: served by the other launch of the call -- unless the caller's statement about the batch promised
        // that there is no such ROI (whole-batch launch on stated extrema)
        if (promised && n != 0) {
            if (lane == 0) atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
            for (int c = lane; c < A.n_cols; c += 64) out_row[c] = __longlong_as_double(0x7ff8000000000000LL);
        }
        return;
    }
    if (!roi_in_launch(A.sp, n, w, h, range)) return;
    constexpr bool DO_GLCM = GLCM != 0;
    const int na = A.glcm_na, Ng = A.ibsi ? 0 : A.grey_depth;
    const uint32_t nb = (uint32_t)A.n_hist;
    const SmallBlock B = small_block(nb, (uint32_t)na, DO_GLCM ? (uint32_t)Ng : 0u, DO_INT, GLCM);
    uint32_t* const S = (uint32_t*)(blk + B.S);
    uint32_t* const lb100 = (uint32_t*)(blk + B.lb100);
    uint32_t* const lbc = (uint32_t*)(blk + B.lbc);
    double* const pq = (double*)(blk + B.pq);
    uint8_t* const plane = (uint8_t*)(blk + B.plane);
    uint32_t* const P = (uint32_t*)(blk + B.P);
    if (n == 0) {
        if (DO_GLCM && lane == 0 && A.glcm_ng) A.glcm_ng[roi] = 0;
        for (int c = lane; c < A.n_cols; c += 64) out_row[c] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    for (int c = lane; c < A.n_cols; c += 64) out_row[c] = 0.0;           // features that are skipped stay 0 (every later store follows in wave order)
    double* const o = out_row + (DO_INT ? A.col_intensity : 0);
    const double dn = (double)n;

    // ---- load: value k of a lane is pixel lane + 64 k ------------------------------------------------------------------------------
    uint32_t v[4];
    bool in[4];
    uint32_t padr[4] = {0u, 0u, 0u, 0u};                                   // GLCM == 2: plane address of the lane's pixel k (0 = a border cell: level 0)
    uint16_t* const pos = (uint16_t*)(blk + B.pos);
    const bool degenerate = DO_GLCM && bin_pixel(vmin, vmin, vmax, A.glcm_grey_depth) == bin_pixel(vmax, vmin, vmax, A.glcm_grey_depth);   // glcm.cpp:27-95
    const double mslope = Ng > 0 ? (double)Ng / ((double)vmax - 0.) : 0.0;
    if (DO_GLCM && !degenerate) {
        for (uint32_t i = lane; i < ((w + 2) * (h + 1) + 3) / 4; i += 64) ((uint32_t*)plane)[i] = 0;   // (the block's plane region is 4-byte aligned)
        if (GLCM == 1) for (int i = lane; i < na * Ng * Ng; i += 64) P[i] = 0;
        wav_sync<false>();
    }
    unsigned long long tot_i = 0, totsq_i = 0;
    const int rounds = (int)((n + 63u) >> 6);                              // rounds of 64 positions that hold a pixel (wave-uniform: the other trips of the k loops are skipped, not masked)
    // original-intensity 0 is skipped by the co-occurrence scan (glcm.cpp:445): level 0; else matlab binning (texture_feature.h:138-167)
    auto level_of = [&](uint32_t val) -> uint32_t { return val ? bin_matlab(val, mslope, Ng) : 0u; };
    const bool from_window = A.win.inten != nullptr;
    if (from_window) {
        // ---- window mode (fused tile path): the ROI's pixels are the cells of its bounding-box window of the tile whose label matches,
        // in row-major order -- the order roi_cloud_kernel gives the cloud, so position i here is position i there and everything
        // behind the load is bit-identical with the cloud path.  Two rows of the (<= 32 wide) window per step, ranks by ballot.
        const uint32_t L = A.win.label[roi], x0 = A.win.x0[roi], y0 = A.win.y0[roi];
        const int dtl = A.win.dt_label, dti = A.win.dt_inten;
        const uint64_t tile_px = (uint64_t)A.win.H * A.win.W, t0 = (uint64_t)A.win.tile[roi] * tile_px;
        const char* const labp = (const char*)A.win.lab + t0 * (uint64_t)dtl;
        const char* const intp = (const char*)A.win.inten + t0 * (uint64_t)dti;
        auto ld = [](const char* p, uint64_t i, int dt) -> uint32_t {
            return dt == 4 ? ((const uint32_t*)p)[i] : dt == 2 ? (uint32_t)((const uint16_t*)p)[i] : (uint32_t)((const uint8_t*)p)[i];
        };
        if (DO_INT) {
#pragma unroll
            for (int k = 0; k < 4; k++) S[(uint32_t)lane + 64u * k] = 0xFFFFFFFFu;
            wav_sync<false>();
        }
        const uint32_t rr = (uint32_t)lane >> 5, cc = (uint32_t)lane & 31u;
        uint32_t base = 0;
        for (uint32_t r = 0; r < h; r += 2) {
            const uint32_t row = r + rr;
            const bool valid = cc < w && row < h;
            const uint64_t e = (uint64_t)(y0 + row) * A.win.W + x0 + cc;
            const uint32_t lb = valid ? ld(labp, e, dtl) : 0u, vv = valid ? ld(intp, e, dti) : 0u;
            const bool mem = valid && lb == L;
            const unsigned long long bal = __ballot(mem);
            const uint32_t rank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (mem && rank < n) {
                if (DO_INT) S[rank] = vv - vmin;
                if (DO_GLCM && !degenerate) plane[row * (w + 2) + cc + 1] = (uint8_t)level_of(vv);
                if (GLCM == 2) pos[rank] = (uint16_t)(row * (w + 2) + cc + 1);
            }
            base += (uint32_t)__popcll(bal);
        }
        wav_sync<false>();
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = (uint32_t)lane + 64u * k;
        in[k] = false; v[k] = 0u;
        if (k >= rounds) {
            if (DO_INT && n > 64 && !from_window) S[i] = 0xFFFFFFFFu;      // (padding of the four-keys-per-lane sort)
            continue;
        }
        in[k] = i < n;
        if (from_window) { v[k] = (DO_INT && in[k]) ? vmin + S[i] : 0u; if (GLCM == 2 && in[k]) padr[k] = pos[i]; }
        else v[k] = in[k] ? A.inten[off + i] : 0u;
        if (in[k]) {
            const uint32_t sq = v[k] * v[k];                               // unsigned-int product, wraps (intensity.cpp:90)
            tot_i += v[k] - vmin;                                          // (offsets: < 2^14 each, four per lane)
            totsq_i += (unsigned long long)(sq & 0xFFFFu) | ((unsigned long long)(sq >> 16) << 32);   // low digits | high digits: < 2^18 each per lane
            if (DO_GLCM && !degenerate && !from_window) {
                const uint32_t px = A.x[off + i], py = A.y[off + i];
                if (px < w && py < h) { plane[py * (w + 2) + px + 1] = (uint8_t)level_of(v[k]); if (GLCM == 2) padr[k] = py * (w + 2) + px + 1; }
            }
        }
        if (DO_INT && !from_window) S[i] = in[k] ? v[k] - vmin : 0xFFFFFFFFu;   // (the sort network runs over a power of two)
    }
    wav_sync<false>();
    SMALL_EXIT(1);

    if (DO_INT) {
        // exact sums: sum v = n vmin + sum (v - vmin) (offsets below 2^14), sum of the wrapped squares as two 16-bit digit sums -- three
        // 32-bit slots of ONE transposed reduction instead of two 64-bit butterflies
        double tot, totsq;
        {
            uint32_t t8[8] = {(uint32_t)(tot_i), (uint32_t)(totsq_i & 0xFFFFFFFFull), (uint32_t)(totsq_i >> 32), 0u, 0u, 0u, 0u, 0u};
            // (tot_i here is the lane's sum of OFFSETS and totsq_i its digit sums: see the load loop)
            const uint32_t r = wave_transpose_sum8_u32(t8, lane);
            const unsigned long long so = (uint32_t)__builtin_amdgcn_readlane((int)r, 0), lo = (uint32_t)__builtin_amdgcn_readlane((int)r, 8), hi = (uint32_t)__builtin_amdgcn_readlane((int)r, 16);
            tot = (double)((unsigned long long)n * vmin + so);             // exact: < 2^41
            totsq = (double)(lo + (hi << 16));
        }
        const bool blank = vmin == 0 && vmax == 0;                         // intensity.cpp:121-122
        // The IEEE quotients of this block in ONE division, a lane each (an fp64 division is ~25 instructions whoever takes part):
        //   lane 0 tot / n (mean, :95)   1 totsq / n (:98)   2 range / (max + min as unsigned int) (:162)   3 range / 100 (histogram.h:55)
        //   4 range / (slide max - slide min) (:72-77)
        double qn = 0.0, qd = 1.0;
        if (lane == 0) { qn = tot; qd = dn; }
        else if (lane == 1) { qn = totsq; qd = dn; }
        else if (lane == 2) { qn = (double)(vmax - vmin); qd = (double)(uint32_t)(vmax + vmin); }
        else if (lane == 3) { qn = (double)range; qd = 100.; }
        else if (lane == 4 && A.slide_min && A.slide_max) { qn = (double)(vmax - vmin); qd = A.slide_max[roi] - A.slide_min[roi]; }
        const double qq = qn / qd;
        auto lane_d = [&](double x, int l) -> double {
            const unsigned long long u = (unsigned long long)__double_as_longlong(x);
            return __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l) << 32) |
                                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l)));
        };
        const double mean = lane_d(qq, 0);
        if (lane == 0) {
            o[I_MIN] = (double)vmin;                                       // intensity.cpp:67-69
            o[I_MAX] = (double)vmax;
            o[I_RANGE] = (double)vmax - (double)vmin;
            o[I_MEAN] = mean;                                              // intensity.cpp:95-99
            o[I_ENERGY] = totsq;
            o[I_INTEGRATED_INTENSITY] = tot;
        }
        if (lane == 1) o[I_ROOT_MEAN_SQUARED] = sqrt(qq);
        if (lane == 2 && !blank) o[I_UNIFORMITY_PIU] = (1.0 - qq) * 100.0;
        if (lane == 4 && A.slide_min && A.slide_max) o[I_COVERED_IMAGE_INTENSITY_RANGE] = qq;
        // ---- central sums (intensity.cpp:102-109, :177-183; M2..M4 of moments.h:53-74 equal the plain central sums) -------------------
        double acc[6] = {0, 0, 0, 0, 0, 0};
        if (!blank) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < rounds && in[k]) {
                    const double d = (double)v[k] - mean, d2 = d * d;
                    acc[0] += fabs(d); acc[1] += d2; acc[2] += d2 * d; acc[3] += d2 * d2; acc[4] += d2 * d2 * d; acc[5] += d2 * d2 * d2;
                }
            // one transposed wave reduction for the six sums (lane L ends up with slot (L >> 3) & 7), fetched by the lane that writes the outputs
            double t8[8] = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], 0.0, 0.0};
            const double tot8 = wave_transpose_sum8(t8, lane);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const unsigned long long u = (unsigned long long)__double_as_longlong(tot8);
                acc[k] = __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 8 * k) << 32) |
                                                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 8 * k)));
            }
        }
        if (lane == 0) {
            // everything that depends only on the central sums (intensity.cpp:110-118, :166-191, moments.h:79-109).  Tolerance-class outputs:
            // the quotients and roots go through reciprocal / reciprocal-square-root estimates with two Newton steps (1-2 ulp) and are
            // shared, as in roi_features_kernel's central_outputs -- ten IEEE divisions and five IEEE roots on one lane were a fifth of
            // this kernel's instructions
            const double var = acc[1];
            const double inv_n = frcp(dn);
            o[I_MEAN_ABSOLUTE_DEVIATION] = acc[0] * inv_n;
            const double variance = dn > 1 ? var * frcp(dn - 1) : 0.0;
            const double variance_b = dn > 1 ? var * inv_n : 0.0;
            const double rsd = variance > 0 ? frsq(variance) : 0.0;       // 1 / sd (0 stands for "sd == 0": every use below tests it)
            const double sd = variance * rsd;
            const double rs_n = frsq(dn);
            o[I_VARIANCE] = variance;
            o[I_VARIANCE_BIASED] = variance_b;
            o[I_STANDARD_DEVIATION] = sd;
            o[I_STANDARD_DEVIATION_BIASED] = variance_b > 0 ? variance_b * frsq(variance_b) : 0.0;
            o[I_COV] = mean != 0.0 ? fdiv(sd, mean) : sd / mean;           // (a zero mean must give the reference's inf / NaN: the IEEE quotient then)
            o[I_STANDARD_ERROR] = sd * rs_n;
            if (!blank) {
                const double M2 = acc[1], M3 = acc[2], M4 = acc[3];
                if (M2 != 0.0) {
                    const double r = frsq(M2), r2 = r * r;                 // 1 / sqrt(M2), 1 / M2
                    const double kurt = n > 4 ? (dn * M4) * (r2 * r2) : 0.0;
                    o[I_SKEWNESS] = n > 3 ? ((dn * rs_n) * M3) * (r2 * r) : 0.0;   // sqrt(n) M3 / pow(M2, 1.5)
                    o[I_KURTOSIS] = kurt;
                    o[I_EXCESS_KURTOSIS] = n > 4 ? kurt - 3 : 0.0;
                }
                const double rsd2 = rsd * rsd, t5 = inv_n * (rsd2 * rsd2 * rsd);   // 1 / (n sd^5); a zero denominator gives 0 (intensity.cpp:186-191)
                o[I_HYPERSKEWNESS] = acc[4] * t5;
                o[I_HYPERFLATNESS] = acc[5] * (t5 * rsd);
            }
        }
        SMALL_EXIT(2);
        if (!blank) {
            // ---- sort the offsets (padding = 0xFFFFFFFF sorts to the end): a bitonic network in REGISTERS -- position e = NV * lane + k,
            // so the exchanges at distance 1, 2 (NV = 4) stay inside a lane and the others are one wave shuffle per register
            if (n <= 64) {
                uint32_t x1[1] = {S[lane]};
                small_sort<1>(x1, lane, 64);
                S[lane] = x1[0];
            } else {
                const uint4 q4 = *(const uint4*)(S + 4 * lane);                             // position e = 4 lane + k holds S[e]: the first n are the ROI's
                // (offsets are below 2^14; the padding 0xFFFFFFFF becomes 0xFFFF and sorts to the end all the same)
                uint32_t r0 = (q4.x & 0xFFFFu) | (q4.y << 16), r1 = (q4.z & 0xFFFFu) | (q4.w << 16);
                small_sort4_packed(r0, r1, lane, n <= 128 ? 128 : 256);
                auto wide = [](uint32_t k16) -> uint32_t { return k16 == 0xFFFFu ? 0xFFFFFFFFu : k16; };
                *(uint4*)(S + 4 * lane) = make_uint4(wide(r0 & 0xFFFFu), wide(r0 >> 16), wide(r1 & 0xFFFFu), wide(r1 >> 16));
            }
            wav_sync<false>();
            SMALL_EXIT(3);
            // ---- histogram bin populations (histogram.h:55-78): lower bounds of the 100 percentile bins and the n custom bins ------------
            const double binW100 = lane_d(qq, 3);                          // (double)range / 100.
            if (n <= 64 && nb <= 64) {
                // one value per lane: the reference's own bin function per pixel (two IEEE divisions), LDS counters, a wave scan -- a third of
                // the three binary searches per lane below
                for (uint32_t t = lane; t < 104 + nb + 8; t += 64) lb100[t] = 0;        // (lb100 | lbc are neighbours in the block)
                wav_sync<false>();
                if ((uint32_t)lane < n) {
                    const uint32_t dd = S[lane];
                    const double realIdx = (double)dd / binW100;           // (h - minVal) / binW100, histogram.h:57-60
                    uint32_t i100 = (realIdx != realIdx) ? 0u : (uint32_t)(int)realIdx;
                    i100 = i100 > 99u ? 99u : i100;                        // slot 100 is folded into 99 (:64-66)
                    uint32_t ic = to_grayscale(vmin + dd, vmin, range, nb);
                    ic = ic > nb - 1 ? nb - 1 : ic;                        // slot n is folded into n - 1 (:76-78)
                    atomicAdd(&lb100[i100], 1u);
                    atomicAdd(&lbc[ic], 1u);
                }
                wav_sync<false>();
                const uint32_t c0 = lb100[lane], c1 = lane < 36 ? lb100[64 + lane] : 0u, cc = (uint32_t)lane < nb ? lbc[lane] : 0u;
                const uint32_t s0 = wave_scan_u32(c0), s1 = wave_scan_u32(c1), sc = wave_scan_u32(cc);
                const uint32_t tot0 = readlane63(s0);
                wav_sync<false>();
                lb100[lane] = s0 - c0;
                if (lane < 36) lb100[64 + lane] = tot0 + s1 - c1;
                if ((uint32_t)lane < nb) lbc[lane] = sc - cc;
            } else
            for (uint32_t t = lane; t < 100 + nb; t += 64) {
                const bool is100 = t < 100;
                const uint32_t b = is100 ? t : t - 100;
                auto bin_of = [=](uint32_t dd) -> uint32_t {
                    if (is100) {
                        const double realIdx = (double)dd / binW100;       // (h - minVal) / binW100, histogram.h:57-60
                        return (realIdx != realIdx) ? 0u : (uint32_t)(int)realIdx;
                    }
                    return to_grayscale(vmin + dd, vmin, range, nb);
                };
                // smallest offset d in [0, range + 1] whose bin index reaches b (intensity_table.h: the real-valued boundary settles it
                // unless it lies within 1e-6 of an integer, where the exact bin function decides) -- range < 16384 in this class
                const double Wl = is100 ? binW100 : (double)range / (double)nb;
                const double Pb = (double)b * Wl;
                const uint32_t mi = (uint32_t)(Pb + 0.5);
                uint32_t d = (uint32_t)Pb + 1;
                if (b == 0) d = 0;
                else if (fabs(Pb - (double)mi) < 1e-6) d = bin_of(mi) >= b ? mi : mi + 1;
                const uint32_t lo = lower_bound_u32(S, n, d);              // pixels with an offset below d
                if (is100) lb100[b] = lo; else lbc[b] = lo;
            }
            wav_sync<false>();
            SMALL_EXIT(4);
            {
                // percentiles P01, P10, P25, P75, P90, P99 (histogram.h:214-243): the LAST bin i with runSum_i <= cnt <= runSum_i + bins_i
                // wins (every matching bin overwrites); runSum_i is the lower bound of bin i.  Lanes test bins i and i + 64.
                const int i0 = lane, i1 = lane + 64;
                const uint32_t r0 = lb100[i0], e0 = (i0 < 99 ? lb100[i0 + 1] : n);
                const uint32_t r1 = i1 < 100 ? lb100[i1] : 0u, e1 = i1 < 100 ? (i1 < 99 ? lb100[i1 + 1] : n) : 0u;
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
                    const uint32_t rs = lb100[mywin], bi = (mywin < 99 ? lb100[mywin + 1] : n) - rs;
                    pv = (mycnt - (double)rs) * binW100 / (double)bi + (double)vmin + binW100 * (double)mywin;
                }
                if (lane < 6) pq[lane] = pv;
                wav_sync<false>();
            }
            const double p10 = pq[1], p90 = pq[4];
            if (lane == 0) {
                o[I_P01] = pq[0]; o[I_P10] = pq[1]; o[I_P25] = pq[2]; o[I_P75] = pq[3]; o[I_P90] = pq[4]; o[I_P99] = pq[5];
                o[I_QCOD] = (pq[3] - pq[2]) / (pq[3] + pq[2]);
                o[I_INTERQUARTILE_RANGE] = pq[3] - pq[2];
            }
            {
                // entropy / uniformity over the n + 1 slots (histogram.h:145-151): slot n is empty
                double e = 0, u = 0;
                for (uint32_t k = lane; k < nb; k += 64) {
                    const uint32_t ck = (k < nb - 1 ? lbc[k + 1] : n) - lbc[k];
                    const double p = fdiv((double)ck, dn);
                    e += p * log2(p + 2.2e-16);
                    u += p * p;
                }
                double t4[4] = {e, u, 0.0, 0.0};
                const double tt = wave_transpose_sum4(t4);                 // lane L holds the total of slot (L >> 4) & 3
                if (lane == 0) o[I_ENTROPY] = -tt;
                if (lane == 16) o[I_UNIFORMITY] = tt;
            }
            SMALL_EXIT(5);
            // median (histogram.h:268-287): order statistics n / 2 and n / 2 - 1 of the sorted array
            const uint32_t hi_v = vmin + S[n / 2], lo_v = vmin + S[n / 2 ? n / 2 - 1 : 0];
            const double median = (n & 1) ? (double)hi_v : (double)(uint32_t)(hi_v + lo_v) / 2.0;
            // mode (histogram.h:289-309): the longest run of the sorted array, the smallest value on ties.  A run starts where a value differs
            // from its predecessor; the start masks of the (up to four) rounds of 64 positions are ballots, and a start finds the next one
            // in its own mask (shift + count trailing zeros) or -- scalar -- in the first later round that has one.
            // key = length << 16 | (0xFFFF - position)
            uint32_t best = 0;
            {
                unsigned long long sm[4] = {0, 0, 0, 0};
                uint32_t xs[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (k < rounds) {
                        const uint32_t i = (uint32_t)lane + 64u * k;
                        const bool valid = i < n;
                        xs[k] = valid ? S[i] : 0u;
                        const uint32_t pv = (valid && i > 0) ? S[i - 1] : 0xFFFFFFFEu;
                        sm[k] = __ballot(valid && (i == 0 || pv != xs[k]));
                    }
                uint32_t nxt[5];                                           // nxt[k]: first start at or behind position 64 k (n when there is none)
                nxt[4] = n;
#pragma unroll
                for (int k = 3; k >= 0; k--) nxt[k] = sm[k] ? 64u * k + (uint32_t)__builtin_ctzll(sm[k]) : nxt[k + 1];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (k < rounds) {
                        const uint32_t i = (uint32_t)lane + 64u * k;
                        if ((sm[k] >> lane) & 1ull) {
                            const unsigned long long above = lane == 63 ? 0ull : sm[k] >> (lane + 1);
                            const uint32_t next = above ? i + 1u + (uint32_t)__builtin_ctzll(above) : nxt[k + 1];
                            const uint32_t key = ((next - i) << 16) | (0xFFFFu - i);
                            best = key > best ? key : best;
                        }
                    }
            }
            best = wave_max_u32(best);
            const uint32_t mode_v = vmin + S[0xFFFFu - (best & 0xFFFFu)];
            SMALL_EXIT(6);
            // ---- robust statistics over [p10, p90] (intensity.cpp:139-149, histogram.h:90-112) and the median deviation (:156-159) ------
            uint32_t a0 = 1, z0 = 0;                                       // positions [a0, z0) inside the bounds: empty unless the bounds say otherwise (NaN: empty)
            if (p10 <= p90 && p90 >= (double)vmin && p10 <= (double)vmax) {
                const double cl = ceil(p10), fl = floor(p90);
                const uint32_t lo_b = cl <= (double)vmin ? vmin : (uint32_t)cl, hi_b = fl >= (double)vmax ? vmax : (uint32_t)fl;
                if (lo_b <= hi_b) {
                    // pixels below a key = set bits of a ballot over the sorted keys (one compare per round instead of a binary search)
                    const uint32_t ka = lo_b - vmin, kz = hi_b - vmin + 1u;
                    a0 = 0; z0 = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (k < rounds) {
                            const uint32_t i = (uint32_t)lane + 64u * k;
                            const uint32_t key = i < n ? S[i] : 0xFFFFFFFFu;
                            a0 += (uint32_t)__popcll(__ballot(key < ka));
                            z0 += (uint32_t)__popcll(__ballot(key < kz));
                        }
                }
            }
            const uint32_t K = z0 > a0 ? z0 - a0 : 0u;
            unsigned long long sx = 0;
            double medad = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = (uint32_t)lane + 64u * k;
                if (k < rounds && i < n) {
                    const uint32_t val = vmin + S[i];
                    if (K && i >= a0 && i < z0) sx += val;
                    medad += fabs((double)val - median);
                }
            }
            sx = wave_sum_u64(sx);
            medad = wave_sum(medad);
            const double mean1090 = K ? (double)sx / (double)K : 0.0;
            double ad = 0;
            if (K) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t i = (uint32_t)lane + 64u * k;
                    if (k < rounds && i >= a0 && i < z0) ad += fabs((double)(vmin + S[i]) - mean1090);
                }
                ad = wave_sum(ad);
            }
            if (lane == 0) {
                o[I_MEDIAN] = median;
                o[I_MODE] = (double)mode_v;
                o[I_ROBUST_MEAN] = mean1090;
                o[I_ROBUST_MEAN_ABSOLUTE_DEVIATION] = K ? fdiv(ad, (double)K) : 0.0;
                o[I_MEDIAN_ABSOLUTE_DEVIATION] = fdiv(medad, dn);
            }
        }
    }

    SMALL_EXIT(7);
    if (GLCM == 2) {
        // ---- 17..64 levels (the reference's default depth): Haralick features straight from the <= 255 pairs of an angle ----------------
        // The dense passes of glcm_features_wave64_v2 (row / column / diagonal sums and the per-cell terms over 64 x 64 cells) cost the
        // same whatever the ROI holds.  Here the marginal counts are four LDS atomics per PAIR, and every per-cell sum is a sum over
        // pairs: a cell of count c is hit by c pairs, so sum_cells c g(c) = sum_pairs g(c(pair)) -- sum c^2 (ASM), the entropy terms
        // lg(c / sum_p + eps), HXY1's lg(px py + eps), JMAX.  Counts are kept asymmetric in 8-bit cells (an angle has at most n - 1
        // <= 255 pairs); a symmetric request reads cell and transposed cell.  HXY2 and everything behind the marginals: glcm_w64.h,
        // the code of the workgroup kernel.  One angle after the other over one 4 KiB matrix.
        double* const og = out_row + A.col_glcm;
        if (degenerate) {
            for (int c = lane; c < kGlcmAngled * na + kGlcmAve; c += 64) og[c] = A.soft_nan;
            return;
        }
        const uint32_t pitch = w + 2;
        const bool symmetric = A.glcm_symmetric != 0;
        const uint32_t inc = symmetric ? 2u : 1u;
        const uint8_t* const M8 = (const uint8_t*)P;
        uint32_t* const mR = (uint32_t*)(blk + B.marg);
        uint32_t* const mC = mR + 64, * const mD = mR + 128, * const mX = mR + 192;
        double* const prow_s = (double*)(blk + B.pp), * const pcol_s = prow_s + 64;
        double* const fs = (double*)(blk + B.fs);
        const bool act = lane < Ng;
        uint32_t lc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) lc[k] = k < rounds ? (uint32_t)plane[padr[k]] : 0u;
        for (int q = 0; q < na; q++) {
            const int ang = A.glcm_angles[q];                              // glcm.cpp:234-255: 0 E, 45 SE, 90 S, 135 SW at distance 1
            const uint32_t delta = ang == 0 ? 1u : ang == 45 ? pitch + 1u : ang == 90 ? pitch : pitch - 1u;
#pragma unroll
            for (int t = 0; t < 4; t++) ((uint4*)P)[lane + 64 * t] = uint4{0u, 0u, 0u, 0u};
            ((uint4*)mR)[lane] = uint4{0u, 0u, 0u, 0u};
            if (lane < 16) ((uint4*)mR)[64 + lane] = uint4{0u, 0u, 0u, 0u};
            wav_sync<false>();
            uint32_t ia[4], ib[4];
            bool pv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                pv[k] = false; ia[k] = 0; ib[k] = 0;
                if (k >= rounds) continue;
                const uint32_t nbv = (uint32_t)plane[padr[k] + delta];
                pv[k] = lc[k] != 0 && nbv != 0;
                if (pv[k]) {
                    const uint32_t a = lc[k] - 1u, b = nbv - 1u, idx = (a << 6) + b;
                    ia[k] = a; ib[k] = b;
                    atomicAdd(&P[idx >> 2], 1u << ((idx & 3u) << 3));
                    atomicAdd(&mR[a], 1u); atomicAdd(&mC[b], 1u);
                    atomicAdd(&mD[a > b ? a - b : b - a], inc); atomicAdd(&mX[a + b], inc);
                    if (symmetric) { atomicAdd(&mR[b], 1u); atomicAdd(&mC[a], 1u); }
                }
            }
            wav_sync<false>();
            const uint32_t rc = act ? mR[lane] : 0u, cc = act ? mC[lane] : 0u, dc = act ? mD[lane] : 0u;
            uint32_t pxpy_c[2] = {lane <= 2 * Ng - 2 ? mX[lane] : 0u, lane + 64 <= 2 * Ng - 2 ? mX[64 + lane] : 0u};
            const uint32_t csum = wave_sum_t<uint32_t>(rc);                // sum_p (glcm.cpp:481-484)
            const bool empty = csum == 0;
            const double sum_p = empty ? 1.0 : (double)csum;
            const double inv_sum_p = fdiv(1.0, sum_p);
            const double pcol = fdiv((double)cc, sum_p), prow = fdiv((double)rc, sum_p), pxmy = fdiv((double)dc, sum_p);
            double pxpy[2] = {0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (lane + 64 * u < 2 * Ng - 1) pxpy[u] = fdiv((double)pxpy_c[u], sum_p);
            prow_s[lane] = prow; pcol_s[lane] = pcol;
            wav_sync<false>();
            // ---- per-pair terms
            double ent = 0, hxy1c = 0, hxy2 = 0;
            uint32_t asm_i = 0, cmax = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k >= rounds) continue;
                if (pv[k]) {
                    uint32_t cnt = M8[(ia[k] << 6) + ib[k]];
                    if (symmetric) cnt += M8[(ib[k] << 6) + ia[k]];
                    asm_i += inc * cnt;                                    // f_asm :555 / f_energy :927-928
                    cmax = cmax > cnt ? cmax : cnt;                        // f_GLCM_JMAX :1178-1179
                    const double pk = (double)cnt * inv_sum_p;
                    ent += (double)inc * (double)fast_log2f(pk + 0.000000001);           // f_entropy :734-735, JE :1160-1161, HXY :868
                    const double pxy = pcol_s[ib[k]] * prow_s[ia[k]];                    // :869, :909
                    hxy1c += (double)inc * (double)fast_log2f(pxy + 0.000000001);
                }
            }
            ent *= inv_sum_p;
            // ---- HXY2 over all (row, column) products: rows in groups of equal row marginal (glcm_features_wave64_v2)
            {
                unsigned long long rem = __ballot(act && rc != 0u);
                while (rem) {
                    const int r0 = (int)__builtin_ctzll(rem);
                    const uint32_t vr = (uint32_t)__builtin_amdgcn_readlane((int)rc, r0);
                    const unsigned long long pbits = (unsigned long long)__double_as_longlong(prow);
                    const double pr = __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pbits >> 32), r0) << 32) |
                                                                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pbits, r0)));
                    const unsigned long long grp = __ballot(act && rc == vr);
                    rem &= ~grp;
                    const double ppr = pcol * pr;
                    const double lg = (double)fast_log2f(ppr + 0.000000001);
                    hxy2 = __builtin_fma(ppr * (double)(uint32_t)__popcll(grp), lg, hxy2);
                }
                if (!act) hxy2 = 0.0;
            }
            if (Ng == 64) glcm_w64_tail<64>(64, lane, rc, cc, dc, pxpy_c, csum, sum_p, inv_sum_p, pcol, prow, pxmy, pxpy, ent, hxy1c, hxy2, asm_i, cmax, fs + q * 64);
            else glcm_w64_tail<0>(Ng, lane, rc, cc, dc, pxpy_c, csum, sum_p, inv_sum_p, pcol, prow, pxmy, pxpy, ent, hxy1c, hxy2, asm_i, cmax, fs + q * 64);
        }
        wav_sync<false>();
        if (lane < na) glcm_features_final((uint32_t*)(fs + lane * 64), A.soft_nan);
        wav_sync<false>();
        for (int c = lane; c < kGlcmAngled * na; c += 64) {                // feature-major, angle-minor (output_2_buffer.cpp:336-346)
            const int k = c / na, a = c - k * na;
            og[c] = fs[a * 64 + k];
        }
        for (int j = lane; j < kGlcmAve; j += 64) {                        // calc_ave (glcm.cpp:1205-1214): std::reduce folds four at a time
            const int k = c_glcm_ave_order[j];
            double init = 0.0;
            int a = 0;
            for (; na - a >= 4; a += 4) {
                const double v1 = fs[a * 64 + k] + fs[(a + 1) * 64 + k];
                const double v2 = fs[(a + 2) * 64 + k] + fs[(a + 3) * 64 + k];
                init = init + (v1 + v2);
            }
            for (; a < na; a++) init = init + fs[a * 64 + k];
            og[kGlcmAngled * na + j] = na ? init / (double)na : 0.0;
        }
        return;
    }
    if (GLCM == 1) {
        double* const og = out_row + A.col_glcm;
        if (lane == 0 && A.glcm_ng) A.glcm_ng[roi] = degenerate ? 0u : (uint32_t)Ng;
        if (degenerate) {
            for (int c = lane; c < kGlcmAngled * na + kGlcmAve; c += 64) og[c] = A.soft_nan;
            return;
        }
        // co-occurrence counts (glcm.cpp:343-485): centre b at (row, col), neighbour a at (row + dy, col + dx); pairs with a level-0
        // member are skipped; the matrix is indexed (centre - 1, neighbour - 1); symmetric counts add the transposed cell too
        const uint32_t area = w * h, pitch = w + 2;
        const int d = A.glcm_offset;
        const bool symmetric = A.glcm_symmetric != 0;
        const bool usual = na == 4 && d == 1 && !symmetric && A.glcm_angles[0] == 0 && A.glcm_angles[1] == 45 && A.glcm_angles[2] == 90 && A.glcm_angles[3] == 135;
        if (usual) {
            // four angles at distance 1, asymmetric counts: E, SE, S, SW of every cell straight from the padded plane
            const int NN = Ng * Ng;
            for (uint32_t p = lane; p < area; p += 64) {
                const uint32_t row = p / w, col = p - row * w;
                const uint8_t* const c = plane + row * pitch + col + 1;
                const uint32_t lb = c[0];
                if (lb == 0) continue;
                const uint32_t e = c[1], se = c[pitch + 1], so = c[pitch], sw = c[pitch - 1];
                uint32_t* const Pr = P + (lb - 1) * (uint32_t)Ng - 1;
                if (e) atomicAdd(Pr + e, 1u);
                if (se) atomicAdd(Pr + NN + se, 1u);
                if (so) atomicAdd(Pr + 2 * NN + so, 1u);
                if (sw) atomicAdd(Pr + 3 * NN + sw, 1u);
            }
        } else
        for (uint32_t p = lane; p < area; p += 64) {
            const int row = (int)(p / w), col = (int)(p - (uint32_t)row * w);
            const uint32_t lb = plane[(uint32_t)row * pitch + (uint32_t)col + 1];
            if (lb == 0) continue;
            for (int q = 0; q < na; q++) {
                const int ang = A.glcm_angles[q];                          // glcm.cpp:234-255
                const int dx = ang == 90 ? 0 : ang == 135 ? -d : d, dy = ang == 0 ? 0 : d;
                const int r2 = row + dy, c2 = col + dx;
                if (r2 < 0 || r2 >= (int)h || c2 < 0 || c2 >= (int)w) continue;
                const uint32_t la = plane[(uint32_t)r2 * pitch + (uint32_t)c2 + 1];
                if (la == 0) continue;
                atomicAdd(&P[q * Ng * Ng + ((int)lb - 1) * Ng + (int)la - 1], 1u);
                if (symmetric) atomicAdd(&P[q * Ng * Ng + ((int)la - 1) * Ng + (int)lb - 1], 1u);
            }
        }
        wav_sync<false>();
        uint32_t* const dst = A.glcm_ws + roi * A.glcm_ws_stride;
        for (int i = lane; i < na * Ng * Ng; i += 64) dst[i] = P[i];
    }
}

// SCAN = false: every slot of the launch is an ROI of this class (its exact list, or a whole batch that is class 0 by the caller's statement):
// a wave per slot.  SCAN = true: a whole-batch launch FILTERED to class 0 -- over a batch that may hold none (the metric configuration:
// 196 000 ROIs of 2821 pixels), which must cost next to nothing: a wave takes `spw` (8 .. 63) consecutive slots, reads their CSR offsets in
// one load and is gone when none of them has <= 256 pixels (a wave per slot cost the headline 2 %, eight per wave 0.8 %); the ROIs it
// does find it serves one after the other (as a loop body the per-ROI code keeps every kernel argument alive across the loop: 116
// registers instead of 58 -- slower per ROI, which is why the unfiltered launches keep the other form).
template <bool DO_INT, int GLCM, bool SCAN>
__global__ __launch_bounds__(256, (SCAN && GLCM != 2) ? 8 : 1) void roi_small_kernel(const RoiArgs A, const uint32_t wave_bytes, const uint32_t promised, const uint32_t spw)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char small_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const blk = small_lds + (size_t)wave * wave_bytes;
    if (!SCAN) {
        small_one<DO_INT, GLCM>(A, (uint64_t)blockIdx.x * 4u + (uint32_t)wave, blk, lane, promised);
        return;
    }
    const uint64_t slot0 = ((uint64_t)blockIdx.x * 4u + (uint32_t)wave) * spw;             // spw <= 63 consecutive slots per wave (one load covers their CSR offsets)
    if (!A.sp.roi_index) {
        const uint64_t i = slot0 + (uint32_t)lane;
        const uint64_t po = ((uint32_t)lane <= spw && i <= A.n_roi) ? A.px_offset[i] : 0;
        const uint64_t nx = (uint64_t)__shfl_down((long long)po, 1, 64);
        const bool cand = (uint32_t)lane < spw && i < A.n_roi && nx - po <= (uint64_t)kSmallPx;
        const unsigned long long found = __ballot(cand);
        if (!found) return;
        if (A.census && lane == 0) atomicAdd(A.census, (uint32_t)__popcll(found));
    }
#pragma unroll 1
    for (uint32_t k = 0; k < spw; k++) {
        // (lane and block address made opaque per trip: what the per-ROI code derives from them is then computed inside the trip instead of
        //  being hoisted out of the loop and kept in registers across it -- the hoisting doubled the kernel's register count)
        int l2 = lane;
        uint32_t b2 = (uint32_t)wave * wave_bytes;
        asm volatile("" : "+v"(l2), "+s"(b2));
        small_one<DO_INT, GLCM>(A, slot0 + k, small_lds + b2, l2, promised);
        wav_sync<false>();                                               // (the block is the next ROI's)
    }
}

} // namespace

// Can the wave-per-ROI kernel serve this launch's family set and settings?  (INTENSITY with the 16-bit-table conditions of its class,
// GLCM only as the split launch of matlab binning with <= 16 levels -- what roi_features_kernel_occ8's compile-time family sets cover.)
bool roi_small_supported(const RoiArgs& a)
{
    if (a.sp.scratch) return false;
    const bool do_int = a.mask & NYXHIP_FAM_INTENSITY, do_glcm = a.mask & NYXHIP_FAM_GLCM;
    if (a.mask & ~(uint32_t)(NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM)) return false;
    if (!do_int && !do_glcm) return false;
    if (do_glcm && a.L.g16) {
        // 17..64 levels: the GLCM-only launch of the split (run_class), distance 1, the four directions
        if (do_int || a.ibsi || a.grey_depth < 17 || a.grey_depth > 64 || a.glcm_offset != 1 || a.glcm_na < 1 || a.glcm_na > kMaxAngles) return false;
        for (int q = 0; q < a.glcm_na; q++)
            if (a.glcm_angles[q] != 0 && a.glcm_angles[q] != 45 && a.glcm_angles[q] != 90 && a.glcm_angles[q] != 135) return false;
        return true;
    }
    if (do_glcm && !(a.glcm_ws != nullptr && a.glcm_ng != nullptr && !a.ibsi && a.grey_depth > 0 && a.grey_depth <= 16 && a.glcm_na >= 1 && a.glcm_na <= kMaxAngles))
        return false;
    if (do_int && (a.n_hist < 1 || a.n_hist > 1024)) return false;              // (the bin bounds sit in the wave's LDS block)
    return true;
}

int launch_roi_small(const RoiArgs& a, void* stream, uint32_t n_slots, bool promised)
{
    if (n_slots == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool do_int = a.mask & NYXHIP_FAM_INTENSITY, do_glcm = a.mask & NYXHIP_FAM_GLCM;
    const int gmode = !do_glcm ? 0 : a.L.g16 ? 2 : 1;
    const SmallBlock B = small_block((uint32_t)a.n_hist, (uint32_t)a.glcm_na, do_glcm ? (uint32_t)a.grey_depth : 0u, do_int, gmode);
    const uint32_t lds = 4u * B.total;
    const bool scan = !promised && a.sp.roi_index == nullptr;           // a filtered whole-batch launch: the batch may hold no ROI of the class
    // slots per wave of a scanning launch: enough waves to fill the chip twice (8192 wave slots), at most 63 slots each
    const uint32_t spw = scan ? std::min<uint32_t>(63u, std::max<uint32_t>(8u, n_slots / 16384u)) : 1u;
    const uint32_t per_wg = 4 * spw;
    const dim3 grid((n_slots + per_wg - 1) / per_wg);
    const uint32_t pr = promised ? 1u : 0u;
#define NYX_SMALL_LAUNCH(I, G)                                                                                              \
    do {                                                                                                                    \
        if (scan) hipLaunchKernelGGL((roi_small_kernel<I, G, true>), grid, dim3(256), lds, st, a, B.total, pr, spw);       \
        else hipLaunchKernelGGL((roi_small_kernel<I, G, false>), grid, dim3(256), lds, st, a, B.total, pr, spw);           \
    } while (0)
    if (gmode == 2) NYX_SMALL_LAUNCH(false, 2);
    else if (do_int && do_glcm) NYX_SMALL_LAUNCH(true, 1);
    else if (do_int) NYX_SMALL_LAUNCH(true, 0);
    else NYX_SMALL_LAUNCH(false, 1);
#undef NYX_SMALL_LAUNCH
    return (int)hipGetLastError();
}

} // namespace nyxhip
