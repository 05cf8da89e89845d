// roi_wide.hip -- first-order intensity features of LDS-sized ROIs of 16-bit data (gfx950).
//
// 16-bit microscopy data spreads a 2821-pixel ROI over 65536 values: the dense counting table of roi_features.hip (16384
// entries) does not apply, and its fallback -- a radix sort, then generic unfused sweeps and a binary search with two fp64
// divisions per step for each of the 108 histogram bounds -- cost 43 ns per ROI against 12.5 ns on 12-bit data (5403 vector
// instructions per wave, ~2000 of them the sort).  Nothing the reference computes needs the values SORTED
// (/root/reference/src/nyx/features/intensity.cpp:57-192, histogram.h:27-309): it needs "how many pixels are <= x", the most
// frequent value, and sums over the values.  So, one 256-thread workgroup per ROI:
//
//   bitmap    65536 presence bits (8 KB of LDS) set with one atomic OR per pixel; the atomic's return value says whether the value was
//             there already -- a DUPLICATE, appended to a short list (expected n^2 / 2R = 60 of 2821 pixels on spread-out data);
//   cum(x)    = set bits below or at x (a prefix-popcount word table + one popcount) + list entries <= x (the list is sorted by
//             counting: it is short): the cumulative distribution, exact, without a sort.  Everything order-related -- histogram
//             bin populations, percentiles, median, the robust range's population -- goes through the table routine of the
//             large-ROI path (intensity_table.h) with this virtual table;
//   mode      the list's most frequent value (count = occurrences + 1), or the minimum when nothing repeats;
//   sums      over the 2821 keys themselves (KeySums): central sums, robust sums, deviations.
//
// An ROI whose list overflows (heavily quantised data: few distinct values, each many times) takes the slow path in the same
// kernel: LDS radix sort (sort_lds.h) + run-length pass -> a compressed table of the distinct values through the same routine.
// The GLCM columns of these ROIs come from the GLCM-only build of roi_features.hip (launch_roi_features with mask = GLCM).
// Built with -ffp-contract=off (device_math.h).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "device_math.h"
#include "roi_kernel.h"
#include "launch_util.h"
#include "sort_lds.h"
#include "intensity_table.h"
#include "../../include/nyxhip.h"

namespace nyxhip {

namespace {

// the virtual table over the offsets 0 .. range: entry i = value vmin + i
struct BitTab {
    const uint32_t* B; const uint16_t* PW; const uint16_t* Ls; uint32_t nL, m;
    __device__ __forceinline__ uint32_t off(uint32_t i) const { return i; }
    __device__ __forceinline__ uint32_t first_ge(uint64_t d) const { return d < m ? (uint32_t)d : m; }
    __device__ __forceinline__ uint32_t dups_le(uint32_t x) const          // list entries <= x (the list is sorted)
    {
        uint32_t lo = 0, hi = nL;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)Ls[mid] <= x) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
    __device__ __forceinline__ uint32_t cum(uint32_t i) const
    {
        const uint32_t w = i >> 5;
        return (uint32_t)PW[w] + (uint32_t)__popc(B[w] & (0xFFFFFFFFu >> (31u - (i & 31u)))) + dups_le(i);
    }
};

// the sums over the ROI's own values (16-bit offsets from the minimum)
struct KeySums {
    const uint16_t* K; uint32_t n, vmin; const uint16_t* Ls; uint32_t nL;
    __device__ __forceinline__ void central(double mean, bool blank, double (&acc)[6], uint32_t& mode_off, const IntensityScratch& S, int tid) const
    {
        const int lane = tid & 63, wave = tid >> 6;
        if (!blank) {
            const double meanx = mean - (double)vmin;                        // deviations in the offset domain: d = x - (mean - vmin)
            for (uint32_t i = tid; i < n; i += 256) {
                const double d = (double)K[i] - meanx, d2 = d * d;
                acc[0] += fabs(d);
                acc[1] += d2;
                acc[2] += d2 * d;
                acc[3] += d2 * d2;
                acc[4] += d2 * d2 * d;
                acc[5] += d2 * d2 * d2;
            }
        }
        // mode (histogram.h:289-309): every value occurs once + its occurrences in the duplicate list; the longest run of the sorted
        // list wins, the smallest value on ties; an empty list leaves the minimum (offset 0: every count is 1)
        uint32_t key = 0xFFFFu;                                             // (run length << 16) | (0xFFFF - offset): larger = better
        if ((uint32_t)tid < nL && (tid == 0 || Ls[tid - 1] != Ls[tid])) {
            uint32_t len = 1;
            while ((uint32_t)tid + len < nL && Ls[tid + len] == Ls[tid]) len++;
            key = (len << 16) | (0xFFFFu - (uint32_t)Ls[tid]);
        }
        key = wave_max_u32(key);
        __syncthreads();
        if (lane == 0) S.w[4 + wave] = key;
        __syncthreads();
        const uint32_t best = max(max(S.w[4], S.w[5]), max(S.w[6], S.w[7]));
        mode_off = 0xFFFFu - (best & 0xFFFFu);
    }
    __device__ __forceinline__ void robust(uint32_t lo_off, uint32_t hi_off, bool some, double median, unsigned long long& sx, double& medad, int tid) const
    {
        const double medx = median - (double)vmin;
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t x = K[i];
            if (some && x - lo_off <= hi_off - lo_off) sx += (unsigned long long)vmin + x;
            medad += fabs((double)x - medx);
        }
    }
    __device__ __forceinline__ void spread(uint32_t lo_off, uint32_t hi_off, double mean1090, double& ad, int tid) const
    {
        const double mx = mean1090 - (double)vmin;
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t x = K[i];
            if (x - lo_off <= hi_off - lo_off) ad += fabs((double)x - mx);
        }
    }
};

// the compressed table of the slow path: entry i = value vmin + U[i]
struct SparseTab {
    const uint16_t* U; const uint16_t* C; uint32_t m;
    __device__ __forceinline__ uint32_t off(uint32_t i) const { return (uint32_t)U[i]; }
    __device__ __forceinline__ uint32_t cum(uint32_t i) const { return (uint32_t)C[i]; }
    __device__ __forceinline__ uint32_t first_ge(uint64_t d) const
    {
        if (d > 0xFFFFull) return m;
        uint32_t lo = 0, hi = m;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)U[mid] < (uint32_t)d) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
};

__global__ __launch_bounds__(256, 7) void roi_wide16_kernel(const WideArgs A)      // (72 VGPRs, no spills: seven workgroups per CU, what the 21.8 KB carve-out of a 2821-pixel ROI allows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int BS = 256, NW = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x >= A.n_list) return;
    const uint64_t roi = A.list[blockIdx.x];
    const uint64_t off = A.px_offset[roi];
    const uint32_t n = (uint32_t)(A.px_offset[roi + 1] - off);
    const uint32_t vmin = A.min_inten[roi], vmax = A.max_inten[roi], range = vmax - vmin;
    double* const o = A.out + roi * A.ld + A.col_intensity;
    __shared__ double s_x[32];
    __shared__ unsigned long long s_u[8];
    __shared__ double s_stat[8];
    __shared__ double s_pq[8];
    __shared__ uint32_t s_w[16];
    if (range > 0xFFFFu) return;                                              // a member beyond 16 bits: the sort engine of roi_features.hip serves it (run_class)
    for (int c = tid; c < kIntensityCols; c += BS) o[c] = 0.0;               // skipped features stay 0 (class members default to 0)
    if (n == 0 || n > A.key_cap) {                                            // (the host sized the launch from the class extrema: cannot happen)
        if (tid == 0 && n != 0) atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        if (n != 0)
            for (int c = tid; c < kIntensityCols; c += BS) o[c] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    uint16_t* const K = (uint16_t*)(lds_raw + A.o_keys);
    uint32_t* const B = (uint32_t*)(lds_raw + A.o_work);                      // [2048] presence bits
    uint16_t* const PW = (uint16_t*)(lds_raw + A.o_work + 8192);              // [2048] set bits in the words before a word
    uint16_t* const L = (uint16_t*)(lds_raw + A.o_work + 8192 + 4096);        // [kWideDupCap] duplicates as they arrive
    uint16_t* const Ls = L + kWideDupCap;                                     // [kWideDupCap] ... sorted
    IntensityScratch S{s_x, s_u, s_stat, s_pq, s_w, (uint32_t*)(lds_raw + A.o_lb), (uint32_t*)(lds_raw + A.o_lb) + 104};
    const bool have_slide = A.slide_min && A.slide_max;
    const double slide_range = have_slide ? A.slide_max[roi] - A.slide_min[roi] : 0.0;

    {
        uint4* const b4 = (uint4*)B;
        for (int i = tid; i < 512; i += BS) b4[i] = make_uint4(0, 0, 0, 0);
        if (tid == 0) s_w[0] = 0;                                             // duplicates so far
    }
    __syncthreads();
    // ---- load: offsets from the minimum, the two exact sums, the presence bits ------------------------------------------------
    unsigned long long sum = 0, sumsq = 0;
    const uint32_t* const gv = A.inten + off;
    for (uint32_t i0 = 0; i0 < n; i0 += 4 * BS) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * BS + tid; v[u] = i < n ? gv[i] : 0u; }   // every load of the trip before the first use
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * BS + tid;
            if (i >= n) continue;
            sum += v[u];
            sumsq += (uint32_t)(v[u] * v[u]);                                  // unsigned-int product, wraps (intensity.cpp:90)
            const uint32_t k = (v[u] - vmin) & 0xFFFFu;
            K[i] = (uint16_t)k;
            const uint32_t bit = 1u << (k & 31u);
            if (atomicOr(&B[k >> 5], bit) & bit) {                             // seen before: one more occurrence of k
                const uint32_t j = atomicAdd(&s_w[0], 1u);
                if (j < (uint32_t)kWideDupCap) L[j] = (uint16_t)k;
            }
        }
    }
    sum = wg4_sum_u64(sum, s_u, tid);
    __syncthreads();
    sumsq = wg4_sum_u64(sumsq, s_u, tid);                                      // (the barriers also publish the keys, the bits and the list)
    const uint32_t nL = s_w[0];
    __syncthreads();

    if (nL <= (uint32_t)kWideDupCap) {
        // ---- prefix popcounts of the bitmap's words: eight words per thread, wave scan, cross-wave carry ----------------------
        {
            uint32_t pc[8], tot = 0;
            const uint4 a = ((const uint4*)B)[2 * tid], c = ((const uint4*)B)[2 * tid + 1];
            pc[0] = __popc(a.x); pc[1] = __popc(a.y); pc[2] = __popc(a.z); pc[3] = __popc(a.w);
            pc[4] = __popc(c.x); pc[5] = __popc(c.y); pc[6] = __popc(c.z); pc[7] = __popc(c.w);
#pragma unroll
            for (int k = 0; k < 8; k++) tot += pc[k];
            const uint32_t sc = wave_scan_u32(tot);
            if (lane == 63) s_w[8 + wave] = sc;
            __syncthreads();
            uint32_t run = sc - tot;
            for (int wv = 0; wv < wave; wv++) run += s_w[8 + wv];
#pragma unroll
            for (int k = 0; k < 8; k++) { PW[8 * tid + k] = (uint16_t)run; run += pc[k]; }
        }
        // ---- the duplicate list, sorted by counting (it is short): rank = smaller entries + equal entries that arrived earlier ----
        if ((uint32_t)tid < nL) {
            const uint32_t x = L[tid];
            uint32_t r = 0;
            for (uint32_t j = 0; j < nL; j++) { const uint32_t y = L[j]; r += (y < x || (y == x && j < (uint32_t)tid)) ? 1u : 0u; }
            Ls[r] = (uint16_t)x;
        }
        __syncthreads();
        const BitTab tab{B, PW, Ls, nL, range + 1};
        const KeySums ks{K, n, vmin, Ls, nL};
        intensity_from_table(tab, ks, n, vmin, vmax, (double)sum, (double)sumsq, have_slide, slide_range, (uint32_t)A.n_hist, o, S, tid);
        return;
    }

    // ---- slow path (the list overflowed: few distinct values, each many times): sort, run-length pass, compressed table --------
    uint16_t* const kb = (uint16_t*)(lds_raw + A.o_work);
    uint16_t* const C = kb + ((A.key_cap + 7u) & ~7u);
    uint32_t* const hist = (uint32_t*)(lds_raw + A.o_work + A.slow_hist);
    uint16_t* const sorted = radix_sort<false, NW, uint16_t>(K, kb, hist, n, 0u, range, tid);
    uint16_t* const U = sorted == K ? kb : K;
    const uint32_t cs = (n + BS - 1) / BS, c0 = (uint32_t)tid * cs < n ? (uint32_t)tid * cs : n, c1 = c0 + cs < n ? c0 + cs : n;
    uint32_t ends = 0;                                                         // runs that end inside this thread's keys
    for (uint32_t i = c0; i < c1; i++) ends += (i + 1 == n || sorted[i + 1] != sorted[i]) ? 1u : 0u;
    const uint32_t sc = wave_scan_u32(ends);
    __syncthreads();
    if (lane == 63) s_w[8 + wave] = sc;
    __syncthreads();
    uint32_t r = sc - ends;
    for (int wv = 0; wv < wave; wv++) r += s_w[8 + wv];
    const uint32_t m = s_w[8] + s_w[9] + s_w[10] + s_w[11];
    for (uint32_t i = c0; i < c1; i++)
        if (i + 1 == n || sorted[i + 1] != sorted[i]) { U[r] = sorted[i]; C[r] = (uint16_t)(i + 1); r++; }
    __syncthreads();
    const SparseTab stab{U, C, m};
    const TableSums<SparseTab> ts{stab, vmin};
    intensity_from_table(stab, ts, n, vmin, vmax, (double)sum, (double)sumsq, have_slide, slide_range, (uint32_t)A.n_hist, o, S, tid);
}

} // namespace

// LDS carve-out for a class whose largest member has max_px pixels.  False: does not apply (too many pixels for 16-bit counts).
bool make_wide_layout(uint32_t max_px, uint32_t n_hist, WideArgs& a)
{
    const uint32_t cap = max_px ? max_px : 1u;
    auto al = [](uint32_t v) { return (v + 15u) & ~15u; };
    uint32_t o = 0;
    a.key_cap = cap;
    a.o_keys = o; o = al(o + 2u * cap);
    a.o_work = o;
    // fast path: bits 8192 + word prefixes 4096 + two lists; slow path: second key buffer + cumulative counts + digit counts
    const uint32_t fast = 8192u + 4096u + 4u * kWideDupCap;
    a.slow_hist = al(2u * ((cap + 7u) & ~7u) + 2u * cap);
    const uint32_t slow = a.slow_hist + 4u * (4u * 256u + 8u);
    o = al(o + (fast > slow ? fast : slow));
    a.o_lb = o; o = al(o + 4u * (112u + n_hist));
    a.lds_bytes = o;
    return cap < 65536u && o + 1024u <= 160u * 1024u;
}

int launch_roi_wide(const WideArgs& a, void* stream)
{
    if (a.n_list == 0) return 0;
    static DeviceOnce optin;
    if (int orc = optin.run([]() -> int {
            return (int)hipFuncSetAttribute((const void*)roi_wide16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        }))
        return orc;
    hipLaunchKernelGGL(roi_wide16_kernel, dim3(a.n_list), dim3(256), a.lds_bytes, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

} // namespace nyxhip
