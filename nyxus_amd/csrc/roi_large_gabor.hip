// roi_large_gabor.hip -- Gabor scores of ROIs whose bounding-box plane does not fit LDS: several workgroups per ROI (round 6).
//
// The reference hands any ROI to any worker (/root/reference/src/nyx/parallel.h:34-41) and convolves the whole bounding-box image
// with every filter of the bank (features/gabor.cpp:333-390 conv_dud, :452-510 GaborEnergy, :43-123 calculate).  Until round 6 an
// ROI beyond LDS ran roi_gabor_kernel<4, true>: ONE workgroup walking a plane of doubles in the global workspace, two global loads
// per tap -- 110 ms for the 400 largest ROIs (300..400-px boxes) of the heavy-tailed batch, nine tenths of an all-families call.
// Here the box is cut into tiles of 64 x 32 output pixels; a workgroup stages its tile's window (tile + n - 1 rows / columns of
// halo, zero outside the box) in LDS as doubles and produces its outputs with the reference's arithmetic: separate multiply and add
// per tap, taps in (j, i) order (a tap that reads a padding zero adds +-0, which leaves a sum that started at +0 as it is: the same
// bits as the reference's clipped loops -- the argument of roi_gabor_tiled_kernel MODE 0).  What crosses a workgroup is exact:
//   lgab_plane_kernel    cloud -> u32 plane of the ROI in the workspace (a workgroup per slab of the cloud; the plane is zeroed by
//                        the launcher's memset);
//   lgab_filter_kernel<0>  the low-pass filter: every tile leaves (max, min, number of pixels AT its min) of its energies;
//   lgab_filter_kernel<1>  folds the ROI's tile records into the filter's extrema (gabor.cpp:84-102: maxval, and the baseline
//                        count(e > min) = area - count(e == min)), then counts, per band-pass filter, the tile's pixels with
//                        e / maxval > threshold (:117) and adds the count to the ROI's counter (integer atomics);
//   lgab_finish_kernel   score = count / baseline (:121), the blank and flat cases (:53-57, :91-96).
// Counts are integers and every energy is computed by exactly one thread in a fixed order: rows are bit-identical to the
// one-workgroup kernels' whatever the tiling, the companions or the budget.
#include <hip/hip_runtime.h>
#include "device_math.h"
#include "roi_kernel.h"
#include "launch_util.h"
#include "../../include/nyxhip.h"

namespace nyxhip {
namespace {

constexpr int kTW = kLgabTileW, kTH = kLgabTileH, kT = 4;      // tile of output pixels; a thread owns kT consecutive outputs of a row
constexpr int kLgabBlk = 256;
typedef const double __attribute__((address_space(4))) * lgab_bank_ptr_t;

__device__ __forceinline__ bool lgab_roi(const LgabArgs& A, uint32_t slot, uint64_t& roi, uint32_t& w, uint32_t& h)
{
    if (!roi_of_slot(A.sp, slot, A.n_roi, roi)) return false;
    w = A.bbox_w[roi]; h = A.bbox_h[roi];
    return true;
}
// ROIs this path leaves to the one-workgroup kernels' conventions (nothing to convolve): written by the finish kernel
__device__ __forceinline__ bool lgab_trivial(const LgabArgs& A, uint64_t roi, uint32_t npx)
{
    return npx == 0 || A.max_inten[roi] == A.min_inten[roi];
}

__global__ __launch_bounds__(kLgabBlk) void lgab_plane_kernel(const LgabArgs A)
{
    uint64_t roi; uint32_t w, h;
    if (!lgab_roi(A, blockIdx.y, roi, w, h)) return;
    const uint64_t off = A.px_offset[roi];
    const uint32_t npx = (uint32_t)(A.px_offset[roi + 1] - off);
    if (lgab_trivial(A, roi, npx)) return;
    uint32_t* const plane = (uint32_t*)(A.ws + (size_t)blockIdx.y * A.stride);
    const uint32_t p0 = blockIdx.x * kLgabSlab, p1 = p0 + kLgabSlab < npx ? p0 + kLgabSlab : npx;
    for (uint32_t i = p0 + threadIdx.x; i < p1; i += kLgabBlk) {
        const uint32_t px = A.x[off + i], py = A.y[off + i];
        if (px < w && py < h) plane[(size_t)py * w + px] = A.inten[off + i];
    }
}

// PHASE 0: the low-pass filter (filter 0 of the bank) -> tile records.  PHASE 1: the band-pass filters -> counts.
template <int PHASE, bool N16>
__global__ __launch_bounds__(kLgabBlk, 2) void lgab_filter_kernel(const LgabArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lgab_lds[];
    __shared__ double s_red[3 * 4];
    __shared__ uint32_t s_cnt[4];
    uint64_t roi; uint32_t w, h;
    if (!lgab_roi(A, blockIdx.y, roi, w, h)) return;
    const uint32_t ntx = (w + kTW - 1) / kTW, nty = (h + kTH - 1) / kTH, ntiles = ntx * nty;
    if (blockIdx.x >= ntiles) return;
    const uint32_t npx = (uint32_t)(A.px_offset[roi + 1] - A.px_offset[roi]);
    if (lgab_trivial(A, roi, npx)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const wsb = A.ws + (size_t)blockIdx.y * A.stride;
    const uint32_t* const plane = (const uint32_t*)wsb;
    double* const rec = (double*)(wsb + A.off_rec);                  // [ntiles][3]: max, min, pixels at the min
    uint32_t* const cnt = (uint32_t*)(wsb + A.off_cnt);              // [nf]
    const int n = N16 ? 16 : A.n, c0 = (n + 1) / 2;                  // (int)ceil(n / 2.), gabor.cpp:492
    const uint32_t ty = blockIdx.x / ntx, tx = blockIdx.x - ty * ntx;
    const int b0 = (int)(ty * kTH), a0 = (int)(tx * kTW);
    const int WW = kTW + n - 1, WH = kTH + n - 1;                    // window of the tile: plane rows b0 + c0 - (n - 1) .. b0 + kTH - 1 + c0
    double* const s_win = (double*)lgab_lds;                         // [WH][WW], zero outside the box
    for (int i = tid; i < WW * WH; i += kLgabBlk) {
        const int r = i / WW, c = i - r * WW;
        const int R = b0 + c0 - (n - 1) + r, Cc = a0 + c0 - (n - 1) + c;
        s_win[i] = (R >= 0 && R < (int)h && Cc >= 0 && Cc < (int)w) ? (double)plane[(size_t)R * w + Cc] : 0.0;
    }
    double maxval = 0.0;
    if (PHASE == 1) {
        // the filter's extrema over the whole box from the tiles' records (every workgroup of the ROI folds them again: a few
        // hundred triples).  Only the maximum is needed here; a flat response (max == min) leaves the ROI to the finish kernel.
        double mx = -1.0, mn = 1.7976931348623157e308;
        for (uint32_t t = tid; t < ntiles; t += kLgabBlk) { const double a = rec[3 * t], b = rec[3 * t + 1]; mx = a > mx ? a : mx; mn = b < mn ? b : mn; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const double a = __shfl_xor(mx, o, 64), b = __shfl_xor(mn, o, 64); mx = a > mx ? a : mx; mn = b < mn ? b : mn; }
        if (lane == 0) { s_red[wave] = mx; s_red[4 + wave] = mn; }
        __syncthreads();
        mx = s_red[0]; mn = s_red[4];
        for (int wv = 1; wv < 4; wv++) { mx = s_red[wv] > mx ? s_red[wv] : mx; mn = s_red[4 + wv] < mn ? s_red[4 + wv] : mn; }
        if (mx == mn) return;                                        // gabor.cpp:91-96 (uniform over the workgroup)
        maxval = mx;
    }
    __syncthreads();
    const int f_begin = PHASE == 0 ? 0 : 1, f_end = PHASE == 0 ? 0 : A.nf;
    for (int f = f_begin; f <= f_end; f++) {
        const lgab_bank_ptr_t G = (lgab_bank_ptr_t)(A.bank + (size_t)f * n * n * 2);
        double tmax = -1.0, tmin = 1.7976931348623157e308;
        uint32_t at_min = 0, score = 0;
        // a thread's work items: kT consecutive outputs of one row of the tile
        for (int it = tid; it < (kTW / kT) * kTH; it += kLgabBlk) {
            const int rr = it / (kTW / kT), cc = (it - rr * (kTW / kT)) * kT;
            if (b0 + rr >= (int)h || a0 + cc >= (int)w) continue;
            double re[kT], im[kT];
#pragma unroll
            for (int t = 0; t < kT; t++) { re[t] = 0.0; im[t] = 0.0; }
            // tap (j, i) of output (rr, cc + t) reads window [rr + n - 1 - j][cc + t + n - 1 - i]; ascending j, then ascending i, as conv_dud
            if (N16) {
                // 16 x 16 banks (the reference's default): a tap row's kT + 15 window words are read ONCE into registers and the sixteen
                // taps run unrolled over them -- with a read per tap and output the loop was bound by LDS bandwidth (5.7 ms per
                // filter and 400 boxes of 300..400 px against 1.3 ms of arithmetic)
                for (int j = 0; j < 16; j++) {
                    const double* const wr = s_win + (rr + 15 - j) * WW + cc;           // window columns cc .. cc + kT + 14
                    double wv[kT + 15];
#pragma unroll
                    for (int q = 0; q < kT + 15; q++) wv[q] = wr[q];
                    const lgab_bank_ptr_t Gj = G + (size_t)j * 32;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const double gr = Gj[2 * i], gi = Gj[2 * i + 1];
#pragma unroll
                        for (int t = 0; t < kT; t++) {
                            const double av = wv[t + 15 - i];
                            re[t] += av * gr;                        // C[ip]   += a * wr   (gabor.cpp:374)
                            im[t] += av * gi;                        // C[ip+1] += a * wi   (:377)
                        }
                    }
                }
            } else
            for (int j = 0; j < n; j++) {
                const double* const wr = s_win + (rr + n - 1 - j) * WW + cc + (n - 1);
                const lgab_bank_ptr_t Gj = G + (size_t)j * n * 2;
                for (int i = 0; i < n; i++) {
                    const double gr = Gj[2 * i], gi = Gj[2 * i + 1];
#pragma unroll
                    for (int t = 0; t < kT; t++) {
                        const double av = wr[t - i];
                        re[t] += av * gr;                            // C[ip]   += a * wr   (gabor.cpp:374)
                        im[t] += av * gi;                            // C[ip+1] += a * wi   (:377)
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < kT; t++) {
                if (a0 + cc + t >= (int)w) continue;
                const double e = sqrt(re[t] * re[t] + im[t] * im[t]);   // :505
                if (PHASE == 0) {
                    tmax = e > tmax ? e : tmax;
                    if (e < tmin) { tmin = e; at_min = 1; } else if (e == tmin) at_min++;
                } else if (e / maxval > A.thr)                       // :117
                    score++;
            }
        }
        if (PHASE == 0) {
            // (max, min, count at the min) of the tile: wave butterflies that carry the count with the minimum, then the four waves
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double a = __shfl_xor(tmax, o, 64), b = __shfl_xor(tmin, o, 64);
                const uint32_t c = (uint32_t)__shfl_xor((int)at_min, o, 64);
                tmax = a > tmax ? a : tmax;
                if (b < tmin) { tmin = b; at_min = c; } else if (b == tmin) at_min += c;
            }
            if (lane == 0) { s_red[wave] = tmax; s_red[4 + wave] = tmin; s_cnt[wave] = at_min; }
            __syncthreads();
            if (tid == 0) {
                double mx = s_red[0], mn = s_red[4];
                uint32_t c = s_cnt[0];
                for (int wv = 1; wv < 4; wv++) {
                    mx = s_red[wv] > mx ? s_red[wv] : mx;
                    if (s_red[4 + wv] < mn) { mn = s_red[4 + wv]; c = s_cnt[wv]; } else if (s_red[4 + wv] == mn) c += s_cnt[wv];
                }
                rec[3 * blockIdx.x] = mx; rec[3 * blockIdx.x + 1] = mn; rec[3 * blockIdx.x + 2] = (double)c;
            }
        } else {
            score = (uint32_t)wave_sum_u64(score);
            if (lane == 0 && score) atomicAdd(&cnt[f - 1], score);
        }
    }
}

__global__ __launch_bounds__(64) void lgab_finish_kernel(const LgabArgs A)
{
    uint64_t roi; uint32_t w, h;
    if (!lgab_roi(A, blockIdx.x, roi, w, h)) return;
    const int lane = threadIdx.x;
    double* const o = A.out + roi * A.ld + A.col_gabor;
    const uint32_t npx = (uint32_t)(A.px_offset[roi + 1] - A.px_offset[roi]);
    if (npx == 0) {
        for (int c = lane; c < A.nf; c += 64) o[c] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    if (A.max_inten[roi] == A.min_inten[roi]) {                      // gabor.cpp:53-57: all zeros, not the soft NaN
        for (int c = lane; c < A.nf; c += 64) o[c] = 0.0;
        return;
    }
    unsigned char* const wsb = A.ws + (size_t)blockIdx.x * A.stride;
    const double* const rec = (const double*)(wsb + A.off_rec);
    const uint32_t* const cnt = (const uint32_t*)(wsb + A.off_cnt);
    const uint32_t ntiles = ((w + kTW - 1) / kTW) * ((h + kTH - 1) / kTH);
    double mx = -1.0, mn = 1.7976931348623157e308;
    for (uint32_t t = lane; t < ntiles; t += 64) { const double a = rec[3 * t], b = rec[3 * t + 1]; mx = a > mx ? a : mx; mn = b < mn ? b : mn; }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) { const double a = __shfl_xor(mx, k, 64), b = __shfl_xor(mn, k, 64); mx = a > mx ? a : mx; mn = b < mn ? b : mn; }
    if (mx == mn) {                                                  // gabor.cpp:91-96
        for (int c = lane; c < A.nf; c += 64) o[c] = A.soft_nan;
        return;
    }
    unsigned long long at_min = 0;
    for (uint32_t t = lane; t < ntiles; t += 64) if (rec[3 * t + 1] == mn) at_min += (unsigned long long)rec[3 * t + 2];
    at_min = wave_sum_u64(at_min);
    const double baseline = (double)((unsigned long long)w * h - at_min);   // count(e > min), :99-102
    for (int c = lane; c < A.nf; c += 64) o[c] = (double)cnt[c] / baseline;  // :121
}

} // namespace

size_t lgab_lds_bytes(int n) { return 8ull * (size_t)(kLgabTileW + n - 1) * (size_t)(kLgabTileH + n - 1); }

int launch_large_gabor(const LgabArgs& a, void* stream, uint32_t n_slots, uint32_t max_px)
{
    if (n_slots == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = lgab_lds_bytes(a.n);
    static DeviceOnce optin;
    if (int orc = optin.run([]() -> int {
        hipError_t e = hipSuccess;
        for (const void* f : {(const void*)lgab_filter_kernel<0, false>, (const void*)lgab_filter_kernel<1, false>, (const void*)lgab_filter_kernel<0, true>,
                              (const void*)lgab_filter_kernel<1, true>})
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        return (int)e;
    }))
        return orc;
    if (hipError_t e = hipMemsetAsync(a.ws, 0, a.stride * (size_t)n_slots, st); e != hipSuccess) return (int)e;   // planes (background 0), records, counters
    hipLaunchKernelGGL(lgab_plane_kernel, dim3((max_px + kLgabSlab - 1) / kLgabSlab, n_slots), dim3(kLgabBlk), 0, st, a);
    const dim3 grid(a.tiles_cap, n_slots);
    if (a.n == 16) {
        hipLaunchKernelGGL((lgab_filter_kernel<0, true>), grid, dim3(kLgabBlk), lds, st, a);
        hipLaunchKernelGGL((lgab_filter_kernel<1, true>), grid, dim3(kLgabBlk), lds, st, a);
    } else {
        hipLaunchKernelGGL((lgab_filter_kernel<0, false>), grid, dim3(kLgabBlk), lds, st, a);
        hipLaunchKernelGGL((lgab_filter_kernel<1, false>), grid, dim3(kLgabBlk), lds, st, a);
    }
    hipLaunchKernelGGL(lgab_finish_kernel, dim3(n_slots), dim3(64), 0, st, a);
    return (int)hipGetLastError();
}

} // namespace nyxhip
