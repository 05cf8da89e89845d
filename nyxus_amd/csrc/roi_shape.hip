// roi_shape.hip -- Gabor filter-bank scores and Zernike moment magnitudes for gfx950.
//
//   roi_gabor_kernel    one workgroup per ROI.  The ROI's bounding-box plane and the whole
//                       complex filter bank sit in LDS; every thread produces output pixels
//                       of the cropped full convolution, accumulating taps in exactly the
//                       reference's (j, i) order with separate multiply and add
//                       (/root/reference/src/nyx/features/gabor.cpp:333-390 conv_dud,
//                       :452-510 GaborEnergy, :43-123 calculate).  The feature is a ratio of
//                       threshold COUNTS, so one flipped pixel is a visible error: the
//                       arithmetic is therefore kept bit-identical instead of being routed
//                       through MFMA (fused, reordered; and on MI355X the fp64 matrix peak
//                       equals the fp64 vector peak, so there is nothing to win).
//                       The bank itself is built on the host with libm, like the reference
//                       (gabor.cpp:393-449), and uploaded once per settings.
//   roi_zernike_kernel  one workgroup per ROI, straight from the pixel cloud (background
//                       pixels carry zero weight, so the dense plane is never needed):
//                       integer-exact centroid moments, then each thread accumulates the
//                       30 complex moments of order <= 9 for its pixels in registers
//                       (features/zernike.cpp:176-343), wave/block reduction, magnitudes.
//
// Built with -ffp-contract=off (device_math.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "device_math.h"
#include "roi_kernel.h"
#include "launch_util.h"
#include "../../include/nyxhip.h"

namespace nyxhip {

__device__ __forceinline__ double wave_min_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// =========================================================================================
// Gabor
// =========================================================================================
template <int NW, bool GS>   // GS: planes in the global workspace (large-ROI launches)
__global__ __launch_bounds__(NW * 64) void roi_gabor_kernel(const ShapeArgs A)
{
    constexpr int kBlk = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char* const lds = GS ? A.sp.scratch + (size_t)blockIdx.x * A.sp.stride : lds_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t roi;
    if (!roi_of_slot(A.sp, blockIdx.x, A.n_roi, roi))
        return;
    double* s_red = (double*)(lds + A.L.red);
    double* s_plane = (double*)(lds + A.L.plane);   // [area] original intensities as double, 0 = background
    double* s_e = (double*)(lds + A.L.energy);      // [area] low-pass response magnitudes
    double* s_bank = (double*)(lds + A.L.bank);     // [(F+1)][n*n][2]

    const uint64_t off = A.px_offset[roi];
    const uint32_t npx = (uint32_t)(A.px_offset[roi + 1] - off);
    const uint32_t w = A.bbox_w[roi], h = A.bbox_h[roi];
    const uint32_t area = w * h;
    const int nF = A.gabor_nf, n = A.gabor_n;
    double* const o = A.out + roi * A.ld + A.col_gabor;
    if (!roi_in_launch(A.sp, npx, w, h, A.max_inten[roi] - A.min_inten[roi]))
        return;                                       // another launch of this call serves the ROI's size class
    if (npx == 0 || area > A.L.area_cap) {
        if (npx != 0 && A.sp.defer_large)
            return;                                   // handled by the spill launch that follows
        if (tid == 0 && npx != 0)
            atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        for (int c = tid; c < nF; c += kBlk)
            o[c] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    if (A.max_inten[roi] == A.min_inten[roi]) {     // gabor.cpp:53-57: all zeros, not the soft NaN
        for (int c = tid; c < nF; c += kBlk)
            o[c] = 0.0;
        return;
    }
    for (uint32_t i = tid; i < area; i += kBlk)
        s_plane[i] = 0.0;
    const int bank_len = (nF + 1) * n * n * 2;
    for (int i = tid; i < bank_len; i += kBlk)
        s_bank[i] = A.gabor_bank[i];
    blk_sync<GS>();
    for_each_cloud_pixel<kBlk>(A.inten + off, A.x + off, A.y + off, npx, tid, [&](uint32_t, uint32_t v, uint32_t px, uint32_t py) {
        if (px < w && py < h)
            s_plane[py * w + px] = (double)v;
    });
    blk_sync<GS>();

    const int c0 = (n + 1) / 2;                     // (int)ceil(n / 2.), gabor.cpp:492
    double maxval = 0, baseline = 0;
    for (int f = 0; f <= nF; f++) {
        const double* G = s_bank + (size_t)f * n * n * 2;
        double tmax = -1.0, tmin = 1.7976931348623157e308;
        uint32_t score = 0;
        for (uint32_t p = tid; p < area; p += kBlk) {
            const int b = (int)(p / w), a = (int)(p - (uint32_t)b * w);
            const int y = c0 + b, x = c0 + a;
            // taps (j, i) with 0 <= y-j < h and 0 <= x-i < w, ascending as in conv_dud
            const int j0 = y - ((int)h - 1) > 0 ? y - ((int)h - 1) : 0, j1 = y < n - 1 ? y : n - 1;
            const int i0 = x - ((int)w - 1) > 0 ? x - ((int)w - 1) : 0, i1 = x < n - 1 ? x : n - 1;
            double re = 0.0, im = 0.0;
            for (int j = j0; j <= j1; ++j) {
                const double* Arow = s_plane + (uint32_t)(y - j) * w;
                const double* Grow = G + (j * n) * 2;
                for (int i = i0; i <= i1; ++i) {
                    double av = Arow[x - i];
                    re += av * Grow[2 * i];        // C[ip]   += a * wr   (gabor.cpp:374)
                    im += av * Grow[2 * i + 1];    // C[ip+1] += a * wi   (:377)
                }
            }
            const double e = sqrt(re * re + im * im);   // :505
            if (f == 0) {
                s_e[p] = e;
                tmax = e > tmax ? e : tmax;
                tmin = e < tmin ? e : tmin;
            } else if (e / maxval > A.gabor_thr)         // :117
                score++;
        }
        if (f == 0) {
            tmax = wave_max_d(tmax);
            tmin = wave_min_d(tmin);
            if (lane == 0) { s_red[wave * 8] = tmax; s_red[wave * 8 + 1] = tmin; }
            blk_sync<GS>();
            double mx = s_red[0], mn = s_red[1];
            for (int wv = 1; wv < NW; wv++) {
                mx = s_red[wv * 8] > mx ? s_red[wv * 8] : mx;
                mn = s_red[wv * 8 + 1] < mn ? s_red[wv * 8 + 1] : mn;
            }
            blk_sync<GS>();
            if (mx == mn) {                               // gabor.cpp:91-96
                for (int c = tid; c < nF; c += kBlk)
                    o[c] = A.soft_nan;
                return;
            }
            maxval = mx;
            uint32_t cnt = 0;                             // baseline score, :99-102
            for (uint32_t p = tid; p < area; p += kBlk)
                cnt += s_e[p] > mn;
            cnt = (uint32_t)wave_sum_u64(cnt);
            if (lane == 0) s_red[wave * 8] = (double)cnt;
            blk_sync<GS>();
            baseline = 0;
            for (int wv = 0; wv < NW; wv++) baseline += s_red[wv * 8];
            blk_sync<GS>();
        } else {
            score = (uint32_t)wave_sum_u64(score);
            if (lane == 0) s_red[wave * 8] = (double)score;
            blk_sync<GS>();
            {
                double sc = 0;
                for (int wv = 0; wv < NW; wv++) sc += s_red[wv * 8];
                if (tid == 0) o[f - 1] = sc / baseline;   // :121
            }
            blk_sync<GS>();
        }
    }
}

// ---- register-tiled variant for the default 16 x 16 kernel ------------------------------------
// Every thread owns T consecutive output pixels of one row.  For one tap row j it loads the T + 16 image words its outputs
// touch ONCE (16-byte LDS reads from a zero-padded u32 plane) and keeps them in registers; the 16 complex taps of the row are
// wave-uniform and come through scalar loads, so the inner loop is nothing but the reference's separate multiplies and adds
// (4 VALU operations per tap and output, the floor for bit-identical arithmetic) instead of three LDS reads per tap.
// Out-of-box taps read a padding zero: adding a * 0 leaves every non-zero partial sum untouched and a zero sum zero, so the
// response is bit-identical to the reference's clipped loops (gabor.cpp:333-390), in the same (j, i) order.
// The low-pass energies are not kept: count(e > min) = area - count(e == min), carried through the min reduction.
typedef const double __attribute__((address_space(4))) * bank_ptr_t;
typedef const float __attribute__((address_space(4))) * bank32_ptr_t;
typedef float v2f __attribute__((ext_vector_type(2)));

// FUSED (opt-in, NYXHIP_GABOR_FUSED=1): each tap is one fused multiply-add -- 1.56x faster, responses a few ulp off the
// reference's.  On noisy or smooth intensity fields that never moved a feature (0 of 18 000 fuzzed ROIs), but on fields with
// exact ties (flat blocks: thousands of pixels share one energy, and which of them is the strict minimum is decided by the
// last bit) 0.6 % of the ROIs changed by up to 9 % (tools/gabor_fuzz.py).  The default therefore keeps the reference's
// separate multiply and add, which is bit-identical on every input.
// ZR: the bank has tap rows whose real or imaginary parts are all zero (ShapeArgs::gabor_zero_rows): those halves of the
// row's arithmetic are skipped.  Banks without such rows run the build without the tests (the three copies of the tap block
// cost the DSB-sized launches 6 %).
// MODE 2 (the default since round 3): fused taps WITH the reference's decisions.  A feature is a count of pixels whose energy
// ratio e / max exceeds the threshold (gabor.cpp:117); the fused response differs from the reference's by a rigorously bounded
// amount -- both are 256-term recursive sums of the same products, each within gamma_257 sum |a w| of the exact sum, and the
// bank is L1-normalised (sum |w| = 1, gabor.cpp:427-448), so |d re|, |d im| <= 2 gamma_257 a_max = 1.15e-13 a_max -- and a pixel
// whose ratio lies farther than that bound from the threshold is decided as the reference decides it.  The others (none on
// ordinary data, thousands on flat fields whose common energy sits on the threshold) go to a list and are recomputed with the
// reference's separate multiplies and adds in its (j, i) order.  The low-pass filter, whose strict minimum and maximum (and the
// number of pixels AT the minimum) the baseline hangs on, runs fused too in the 256-thread kernel: the true extrema are found
// among candidates -- pixels whose fused energy lies within the bound of a sampled lower bound of the maximum / upper bound of the
// minimum -- which are recomputed with the reference's arithmetic; a list overflow (ties on a flat field) runs the filter again
// unfused.  (A filter whose taps are one real constant is a box filter: exact integer sums, nothing to check.  The one-wave
// kernel of small ROIs keeps the unfused low-pass: two recomputation calls cost more than it saves there.)
// MODE 3 (the default since round 4): the same decisions from a SCREENING pass in packed fp32.  `v_pk_fma_f32` does two fp32 FMAs
// per lane and instruction -- a tap's real and imaginary products in one -- at twice the fp64 rate (tools/pipe_probe.hip: 132.7
// against 66.5 TFLOP/s; the fp64 matrix pipe does not overlap the vector pipe on gfx950 and adds nothing).  Intensities below
// 2^24 are exact in fp32, the taps are rounded once (relative 2^-24), and a 256-term FMA chain in fp32 stays within
// gamma_258 sum |a w| <= 1.54e-5 a_max of the exact sum per component (the bank is L1-normalised), so the screened energy is within
// 3.1e-5 a_max of the reference's: a pixel farther than 3.2e-5 a_max (+ 1e-15 (e + T)) from the threshold is decided as the
// reference decides it; the others -- a few per thousand on ordinary data -- are recomputed with the reference's fp64 arithmetic,
// and the low-pass extrema come from exactly recomputed candidates, as in MODE 2.  An ROI with an intensity of 2^24 or more runs
// every filter with the reference's arithmetic.
constexpr int kGaborRedoCap = 512;
// The response of one pixel with the reference's arithmetic: separate multiply and add, taps in (j, i) order (gabor.cpp:333-390; a
// zero tap adds +-0, which leaves a sum that started at +0 as it is: the same bits as the scans that skip zero rows).  Not inlined:
// it runs for a handful of pixels, and inlined into the unrolled output loop it cost the kernel 50 registers (two waves per SIMD
// instead of four).
// PF: the plane holds the intensities as fp32 bit patterns (MODE 3, every intensity of the ROI below 2^24: see the kernel) -- a
// compile-time fact of the copy: as a run-time flag it put a branch on every tap of this loop (46.6 against 38.9 ms per 196 k ROIs).
typedef __attribute__((address_space(3))) _Float16 lds_f16_t;
typedef _Float16 gabor_h8 __attribute__((ext_vector_type(8)));
typedef float gabor_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;  // (the plane through LDS instructions: behind a generic pointer this function read it with flat loads)
// UNI = false: the lanes of a call belong to different filters (the merged lists of the MFMA stage) -- taps through vector loads.
template <int PF, bool UNI = true>
__device__ __attribute__((noinline)) double gabor_exact_energy(const lds_u32_t* s_plane, uint32_t pitch, uint32_t a, uint32_t b, bank_ptr_t G, uint32_t words = 0)
{
    // (the bank pointer is the same in every lane; said so, the taps come through the scalar cache -- sixteen s_load per tap row
    //  instead of 512 vector loads per call -- and a row's sixteen window words are read before the first is used)
    const uint64_t gp = (uint64_t)(uintptr_t)G;
    const bank_ptr_t Gu = !UNI ? G : (bank_ptr_t)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gp >> 32)) << 32) |
                                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gp));
    double re = 0.0, im = 0.0;
    if constexpr (PF == 2) {
        // MODE 4 after the low-pass filter: the plane is two planes of f16 digits (same rows and pitch, `words` elements apart), whose
        // sum is the intensity -- both conversions and the sum are exact
        const lds_f16_t* rp = (const lds_f16_t*)s_plane + (b + 15) * pitch + a + 16;
#pragma unroll 1
        for (int j = 0; j < 16; j++, rp -= pitch) {
            _Float16 wh[16], wl[16];
#pragma unroll
            for (int i = 0; i < 16; i++) { wh[i] = rp[-i]; wl[i] = rp[(int)words - i]; }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const double av = (double)((float)wh[i] + (float)wl[i]);
                re += av * Gu[(j * 16 + i) * 2];
                im += av * Gu[(j * 16 + i) * 2 + 1];
            }
        }
        return sqrt(re * re + im * im);
    }
    const lds_u32_t* rp = s_plane + (b + 15) * pitch + a + 16;
#pragma unroll 1
    for (int j = 0; j < 16; j++, rp -= pitch) {
        uint32_t wv[16];
#pragma unroll
        for (int i = 0; i < 16; i++) wv[i] = rp[-i];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const double av = PF ? (double)__uint_as_float(wv[i]) : (double)wv[i];
            re += av * Gu[(j * 16 + i) * 2];
            im += av * Gu[(j * 16 + i) * 2 + 1];
        }
    }
    return sqrt(re * re + im * im);
}

// A tile's window of T + 16 words as 16-byte reads that stay 16-byte reads: word 0 of the window is never used (taps reach words
// 1 .. T + 15), and left alone the compiler drops it, which shifts the rest off their 16-byte alignment -- the window came in as
// twelve ds_read2_b32 (4 LDS cycles each, banks of a 128-byte period) instead of six ds_read_b128.  The empty asm behind the last
// load makes every word live (one statement for the whole window: the loads are all in flight before the first wait).
template <int W4>
__device__ __forceinline__ void lds_load_window(const uint32_t* row, uint4 (&u)[W4])
{
#pragma unroll
    for (int q = 0; q < W4; q++) u[q] = ((const uint4*)row)[q];
    static_assert(W4 == 5 || W4 == 6, "window of 20 or 24 words");
    if constexpr (W4 == 6)
        asm volatile("" : "+v"(u[0].x), "+v"(u[0].y), "+v"(u[0].z), "+v"(u[0].w), "+v"(u[1].x), "+v"(u[1].y), "+v"(u[1].z), "+v"(u[1].w),
                          "+v"(u[2].x), "+v"(u[2].y), "+v"(u[2].z), "+v"(u[2].w), "+v"(u[3].x), "+v"(u[3].y), "+v"(u[3].z), "+v"(u[3].w),
                          "+v"(u[4].x), "+v"(u[4].y), "+v"(u[4].z), "+v"(u[4].w), "+v"(u[5].x), "+v"(u[5].y), "+v"(u[5].z), "+v"(u[5].w));
    else
        asm volatile("" : "+v"(u[0].x), "+v"(u[0].y), "+v"(u[0].z), "+v"(u[0].w), "+v"(u[1].x), "+v"(u[1].y), "+v"(u[1].z), "+v"(u[1].w),
                          "+v"(u[2].x), "+v"(u[2].y), "+v"(u[2].z), "+v"(u[2].w), "+v"(u[3].x), "+v"(u[3].y), "+v"(u[3].z), "+v"(u[3].w),
                          "+v"(u[4].x), "+v"(u[4].y), "+v"(u[4].z), "+v"(u[4].w));
}

// acc += {a, a} * g with a = the low (HI = 0) or high (HI = 1) half of the register pair `pair`: the window stays in the registers the
// 16-byte loads delivered it to, and the half is picked by the instruction's op_sel bits.  (Written by the compiler, a window value in
// an odd register is first copied to an even one -- v_pk_fma_f32 takes 64-bit sources -- one v_mov per value and tap row.)
template <int HI>
__device__ __forceinline__ void pk_fma_bcast(v2f& acc, const v2f pair, const v2f g)
{
    if constexpr (HI == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(pair), "s"(g));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(pair), "s"(g));
}

// Diagnostic builds only (-DNYXHIP_GABOR_PHASE_EXITS, NYXHIP_DBG_PHASE=1..4 at run time): a workgroup leaves after the plane build /
// the low-pass filter / the digit planes / the MFMA screening -- the instruction counters of such runs give the phases' shares.
#ifdef NYXHIP_GABOR_PHASE_EXITS
#define NYX_GABOR_PHASE_EXIT(cond, stmt) do { if (cond) stmt; } while (0)
#else
#define NYX_GABOR_PHASE_EXIT(cond, stmt) do { } while (0)
#endif
// The value again, but opaque to the optimiser at this point: per-lane address arithmetic that depends only on the thread index is
// otherwise hoisted to the kernel's entry and -- alive across every phase -- spilled there (five scratch stores per wave: 1.5 GB of
// HBM writes per 196 k ROIs for 6 MB of results).
__device__ __forceinline__ int here(int v) { asm volatile("" : "+v"(v)); return v; }

// acc += pair * {b, b} with b the low (HI = 0) or high (HI = 1) half of the scalar register pair `bpair`
template <int HI>
__device__ __forceinline__ void pk_fma_sb(v2f& acc, const v2f pair, const v2f bpair)
{
    if constexpr (HI == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(pair), "s"(bpair));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(pair), "s"(bpair));
}

template <int T, int NW, int MODE, bool ZR = false>
__global__ __launch_bounds__(NW * 64, MODE >= 2 ? 4 : 1) void roi_gabor_tiled_kernel(const ShapeArgs A)   // (MODE 2 / 3: held to 128 registers, four waves per SIMD like the other two)
{
    constexpr int N = 16, kBlk = NW * 64, W4 = (T + 16) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t roi;
    if (!roi_of_slot(A.sp, blockIdx.x, A.n_roi, roi))
        return;
    double* s_red = (double*)(lds_raw + A.L.red);
#ifdef NYXHIP_GABOR_PHASE_EXITS
    uint32_t* const diag_cnt = (uint32_t*)(s_red + 12);  // (slots no reduction of the default bank touches)
#endif
    uint32_t* s_plane = (uint32_t*)(lds_raw + A.L.plane);   // [(h + 15)][pitch] original intensities, zero padding
    uint32_t* s_redo = (uint32_t*)(lds_raw + A.L.redo);     // MODE 2: [0] pixels to recompute, [1 ..] their indices b * w + a

    const uint64_t off = A.px_offset[roi];
    const uint32_t npx = (uint32_t)(A.px_offset[roi + 1] - off);
    const uint32_t w = A.bbox_w[roi], h = A.bbox_h[roi];
    const uint32_t area = w * h;
    const int nF = A.gabor_nf;
    double* const o = A.out + roi * A.ld + A.col_gabor;
    if (!roi_in_launch(A.sp, npx, w, h, A.max_inten[roi] - A.min_inten[roi]))
        return;                                       // another launch of this call serves the ROI's size class
    if (npx == 0 || area > A.L.area_cap || (w > h ? w : h) > A.L.side_cap) {
        if (npx != 0 && A.sp.defer_large)
            return;                                   // handled by the spill launch that follows
        if (tid == 0 && npx != 0)
            atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        for (int c = tid; c < nF; c += kBlk)
            o[c] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    if (A.max_inten[roi] == A.min_inten[roi]) {     // gabor.cpp:53-57: all zeros, not the soft NaN
        for (int c = tid; c < nF; c += kBlk)
            o[c] = 0.0;
        return;
    }
    // tap (j, i) of output (a, b) reads image (a + 8 - i, b + 8 - j): 7 padding rows above, 8 below, 8 padding columns
    // left, >= 8 right; padded column = image column + 8, padded row = image row + 7
    const uint32_t tpr = (w + T - 1) / T;             // tiles per row
    // (an ODD number of 16-byte units: a wave's 16-byte window reads then fall on different banks for lanes one row apart --
    //  with an even number the eight tiles of a row and those of the next row hit the same eight units of the 256-byte bank
    //  period, and the counters showed the LDS pipe 96 % busy at 28 cycles per read instead of 8)
    const uint32_t pitch = (tpr * T + 16) | 4u;       // multiple of 4 words
    const uint32_t words = pitch * (h + 15);
    {
        uint4* p4 = (uint4*)s_plane;
        for (uint32_t i = tid; i < words / 4; i += kBlk)
            p4[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    // MODE 3: the plane holds fp32 bit patterns when every intensity of the ROI is an fp32 integer (below 2^24) -- the screening pass
    // then reads its window without 23 conversions per tap row (26.7 k -> 24.2 k vector instructions per wave, 41.4 -> 38.9 ms per
    // 196 k ROIs); the copies that follow the reference's arithmetic convert fp32 -> fp64 instead of u32 -> fp64, the same value.
    // Zero padding is +0.0f.
    const bool pf = MODE >= 3 && A.max_inten[roi] < (1u << 24);
    for_each_cloud_pixel<kBlk>(A.inten + off, A.x + off, A.y + off, npx, tid, [&](uint32_t, uint32_t v, uint32_t px, uint32_t py) {
        if (px < w && py < h)
            s_plane[(py + 7) * pitch + px + 8] = pf ? __float_as_uint((float)v) : v;
    });
    __syncthreads();

    NYX_GABOR_PHASE_EXIT(MODE == 4 && A.dbg_phase == 1, return);
    const bank_ptr_t bank = (bank_ptr_t)(uintptr_t)A.gabor_bank;
    const bank32_ptr_t bank32 = (bank32_ptr_t)(uintptr_t)A.gabor_bank32;
    // Tiles are dealt to lanes COLUMN-major in blocks of 16 rows: the LDS serves a ds_read_b128 in four fixed groups of 16 lanes
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} of each half-wave; MI355X_MICROARCH.md, LDS), and a group whose 16 lanes read the
    // same tile column of 16 consecutive rows touches 16 different 16-byte units of the 256-byte bank period (the pitch is an odd
    // number of units) -- conflict-free for every box width.  Row-major dealing (lanes = consecutive tiles of a row, 32 bytes
    // apart) put two lanes of a group on one unit: SQ_LDS_BANK_CONFLICT 67 % of SQ_LDS_IDX_ACTIVE, the LDS pipe busy 40 of the
    // kernel's 48 ms.  vtid: the thread's index with the lanes of a hardware group made contiguous.
    // (Row blocks pad the box to a multiple of 16 rows: where that would add a trip to the tile loop -- boxes a little over 16 or 32
    //  rows -- and in the one-wave kernel of small ROIs, which does not saturate the LDS, the tiles stay row-major.)
    const bool colmajor = NW == 4 && (((h + 15u) & ~15u) * tpr + kBlk - 1) / kBlk == (h * tpr + kBlk - 1) / kBlk;
    auto vtid_of = [&](uint32_t t) -> uint32_t {        // (formed where a tile loop starts: see here())
        return !colmajor ? t : (t & ~31u) | ((0x73261540u >> ((t >> 2 & 7u) * 4u)) & 7u) << 2 | (t & 3u);
    };
    const uint32_t ntiles = (colmajor ? (h + 15u) & ~15u : h) * tpr;
    double maxval = 0;
    double tmax = -1.0, tmin = 1.7976931348623157e308;
    uint32_t n_min = 0;                                // pixels of this thread whose low-pass energy equals tmin

    auto exact_energy = [&](uint32_t a, uint32_t b, const bank_ptr_t G) -> double { const lds_u32_t* pl = (const lds_u32_t*)s_plane;
        return pf ? gabor_exact_energy<1>(pl, pitch, a, b, G) : gabor_exact_energy<0>(pl, pitch, a, b, G);
    };
    bool lp_overflow = false;                          // MODE 2: the fused low-pass pass found more candidates than its list holds
    const double amax = (double)A.max_inten[roi];

    // One filter over the whole box.  FUSE is a compile-time fact of the copy (the low-pass filter runs the copy with the
    // reference's arithmetic, the others -- MODE 1 / 2 -- the fused one): with both tap blocks behind a run-time flag in ONE tile
    // loop the MODE 2 build needed 164 registers against 120 for either alone.  Returns false when the ROI is finished.
    auto run_filter = [&](const int f, auto fuse_c) -> bool {
        constexpr bool fuse = decltype(fuse_c)::value;
        const bank_ptr_t G = bank + (size_t)f * N * N * 2;
        const uint32_t zero_rows = ZR ? A.gabor_zero_rows[f] : 0u;
        const bool box = ZR && ((A.gabor_box_mask >> f) & 1u) && A.max_inten[roi] < (1u << 24);
        const double box_c = box ? G[0] : 0.0;
        constexpr bool f32 = MODE >= 3 && fuse;             // this copy screens in packed fp32 (see the kernel's header)
        constexpr double kErr = f32 ? 3.2e-5 : 2.5e-13;     // |screened energy - reference's energy| <= kErr a_max
        const double thr_max = A.gabor_thr * maxval, thr_slack = __builtin_fma(kErr, amax, 1e-15 * thr_max);
        uint32_t sc = 0;
        // MODE 2, low-pass filter with fused taps: its strict maximum and minimum (and the number of pixels AT the minimum) carry
        // every feature, so they are the reference's -- found among CANDIDATES.  One pixel per thread, spread over the box, is
        // evaluated with the reference's arithmetic first: the largest of these energies, Bs, is a lower bound of the true maximum
        // (the smallest, bs, an upper bound of the true minimum), so the pixel that attains the true maximum has a fused energy of at
        // least Bs minus the error bound, and only such pixels go to the list (on a continuous field ~ area / 256 of them; a flat
        // field, where thousands tie, overflows the list and the filter runs again with the reference's arithmetic throughout).
        double Bs = 0.0, bs = 0.0;
        const bool lp_cand = MODE >= 2 && fuse && f == 0 && !box;     // (a box filter's sums are exact integers: nothing to check)
        // The low-pass filter is an outer product C_j B_i (ShapeArgs::gabor_lp_sep): its screening pass runs SEPARABLY -- a real
        // 16-tap filter along each plane row, then the complex 16-tap filter down the rows, a row's filtered values shared by the two
        // output rows a thread owns: 102 packed FMAs per output instead of 256.  Error against the exact sum, per component: the
        // factors rounded to fp32 (2 x 2^-24), two FMA chains of 16 (gamma_17 each; the row chain as two of 8 and an addition) and
        // the residual of the factorisation (<= 1e-12): <= 2.2e-6 sum |a w|, and as above sqrt of the two components' squares <= a_max.
        const bool lpsep = f32 && lp_cand && A.gabor_lp_sep != 0;
        const double kErrLp = lpsep ? 2.3e-6 : kErr;
        if (MODE >= 2 && fuse) {
            if (tid == 0) s_redo[0] = 0;
            if (lp_cand && !lpsep) {                      // (the separable pass takes its bounds from its own first trip: see there)
                if (wave == 0) {                          // (64 samples: one wave's worth of the reference's arithmetic)
                    const uint32_t p = (uint32_t)(((unsigned long long)(uint32_t)lane * area) / 64u), pb = p / w, pa = p - pb * w;
                    const double es = exact_energy(pa, pb, G);
                    const double wmx = wave_max_d(es), wmn = wave_min_d(es);
                    if (lane == 0) { s_red[0] = wmx; s_red[1] = wmn; }
                }
                __syncthreads();
                Bs = s_red[0]; bs = s_red[1];
            }
            __syncthreads();
        }
        // The screening copy decides in the SQUARED domain -- no square root per output (a correctly rounded fp64 sqrt is ~15
        // instructions).  S bounds the slack of the comparison below for every possible energy (e <= sqrt(2) a_max: the bank is
        // L1-normalised), so "e within S of the threshold" contains every pixel the e-domain test would recompute; s2 = fl(re^2 +
        // im^2) is within one rounding of e^2 (the squares of fp32 values are exact in fp64) and the bounds carry a factor
        // (1 +- 1e-15) for it and for their own rounding: s2 > thr_hi2 implies e > T + S, s2 < thr_lo2 implies e < T - S.
        const double S_thr = thr_slack + 3e-15 * amax;
        const double t_lo = thr_max - S_thr > 0.0 ? thr_max - S_thr : 0.0, t_hi = thr_max + S_thr;
        const double thr_lo2 = t_lo * t_lo * (1.0 - 1e-15), thr_hi2 = t_hi * t_hi * (1.0 + 1e-15);
        // (low-pass candidates, same construction: at least Bs - M, or at most bs + M)
        const double M_lp = kErrLp * amax + 3e-15 * amax + 1e-15 * (Bs > bs ? Bs : bs);
        const double b_lo = Bs - M_lp > 0.0 ? Bs - M_lp : 0.0, b_hi = bs + M_lp;
        double lp_hi2 = b_lo * b_lo * (1.0 - 1e-15), lp_lo2 = b_hi * b_hi * (1.0 + 1e-15);
        // The low-pass filter as a box filter over intensities below 2^16 (the reference's default bank on 8- to 16-bit images): the
        // 16 x 16 window sums stay below 2^24, so fp32 additions of the plane's fp32 patterns are exact in any order.  A thread owns
        // T columns of TWO rows: the column sums over the fourteen tap rows the two outputs share are formed once, the three rows
        // that differ (first, last of the upper output; last of the lower) go on top, and a sliding sum along the row finishes each
        // -- 29 additions per output instead of the 122 operations of the tile loop below, half its LDS reads.  Threads a row pair
        // apart walk the shared rows in opposite directions: with the row stride of a pair an even number of 16-byte units their
        // reads would otherwise meet on half of the banks.  Energies are c * sum exactly (c a power of two; the square is exact,
        // so is its root): extrema and the count at the minimum are taken on the sums.
        const bool fbox = MODE >= 3 && ZR && box && f == 0 && amax < 65536.0;
        if constexpr (f32) if (lpsep) {
            // a thread owns TL = 4 columns of RL rows (four in the 256-thread kernel, two in the one-wave kernel of small boxes: fewer
            // rows to filter for nothing); lanes run along the column blocks -- 16-byte window reads one unit apart: conflict-free
            constexpr int TL = 4, RL = NW == 4 ? 4 : 2, WL4 = (TL + 16) / 4, NV = TL * RL;
            const uint32_t n_cb = (w + (uint32_t)TL - 1u) / (uint32_t)TL, n_rp = (h + (uint32_t)RL - 1u) / (uint32_t)RL, n_it = n_rp * n_cb;
            v2f Bp[8];
#pragma unroll
            for (int k = 0; k < 8; k++) Bp[k] = v2f{A.gabor_lp_B[2 * k], A.gabor_lp_B[2 * k + 1]};
            // Candidates for the extrema without a single sample of the reference's arithmetic: the FIRST trip of the item loop
            // keeps its screened energies in registers; their largest and smallest over the workgroup bound the true maximum from
            // below (largest - M) and the true minimum from above (smallest + M), and a pixel is a candidate when its screened
            // energy lies within 2 M of them -- the pixel that attains the true maximum does, and so does every pixel AT the true
            // minimum.  Later trips (boxes of more items than threads) compare against the first trip's bounds.
            float keep[NV];                              // (fp32 copies: the bounds below carry 1e-6 relative for them)
            bool kept = false;
            uint32_t kb0 = 0, ka0 = 0;
            double lmax = -1.0, lmin = 1.7976931348623157e308;
            bool first = true;
            auto push_item = [&](const auto (&v)[NV], uint32_t b0, uint32_t a0) {        // v[RL * t + rr]: column a0 + t, row b0 + rr
#pragma unroll
                for (int q = 0; q < NV; q++)
                    if (v[q] >= 0 && ((double)v[q] >= lp_hi2 || (double)v[q] <= lp_lo2)) {
                        const uint32_t kk = atomicAdd(&s_redo[0], 1u);
                        if (kk < (uint32_t)kGaborRedoCap) s_redo[1 + kk] = (b0 + (uint32_t)(q % RL)) * w + a0 + (uint32_t)(q / RL);
                    }
            };
            auto return_item = [&](const double (&v)[NV], uint32_t b0, uint32_t a0) {
                if (first) {
#pragma unroll
                    for (int q = 0; q < NV; q++) {
                        keep[q] = (float)v[q];
                        if (v[q] >= 0.0) { lmax = v[q] > lmax ? v[q] : lmax; lmin = v[q] < lmin ? v[q] : lmin; }
                    }
                    kept = true; kb0 = b0; ka0 = a0;
                } else
                    push_item(v, b0, a0);
            };
            auto do_item = [&](uint32_t it) {
                const uint32_t rpi = it / n_cb, cb = it - rpi * n_cb, b0 = (uint32_t)RL * rpi, a0 = cb * (uint32_t)TL;
                // padded row b0 + k is tap row 15 + rr - k of output row b0 + rr.  (The last row group of a box whose height is not a
                // multiple of RL asks for padded rows that do not exist: they only feed output rows that do not exist either, and the
                // last row that does is read in their place.)
                const uint32_t k_last = h + 14u - b0;
                const uint32_t* const top = s_plane + b0 * pitch + a0;
                v2f o[RL][TL];
#pragma unroll
                for (int rr = 0; rr < RL; rr++)
#pragma unroll
                    for (int t = 0; t < TL; t++) o[rr][t] = v2f{0.0f, 0.0f};
#pragma unroll 1
                for (int k = 0; k < 16 + RL - 1; k++) {
                    const uint32_t* const row = top + ((uint32_t)k < k_last ? (uint32_t)k : k_last) * pitch;
                    v2f win2[(TL + 16) / 2];                     // window words 2 q, 2 q + 1
#pragma unroll
                    for (int q = 0; q < WL4; q++) {
                        const uint4 u = ((const uint4*)row)[q];
                        win2[2 * q] = v2f{__uint_as_float(u.x), __uint_as_float(u.y)};
                        win2[2 * q + 1] = v2f{__uint_as_float(u.z), __uint_as_float(u.w)};
                    }
                    // row filter: tap i of output t reads word t + 16 - i.  Even taps on the output pairs (t, t + 1), t even -- an
                    // aligned register pair of the window; odd taps on the pairs (t - 1, t): the same pairs serve them
                    v2f ae[TL / 2], ao[TL / 2 + 1];
#pragma unroll
                    for (int q = 0; q < TL / 2; q++) ae[q] = v2f{0.0f, 0.0f};
#pragma unroll
                    for (int q = 0; q <= TL / 2; q++) ao[q] = v2f{0.0f, 0.0f};
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
#pragma unroll
                        for (int t = 0; t < TL; t += 2) pk_fma_sb<0>(ae[t / 2], win2[(t + 16 - i) / 2], Bp[i / 2]);
#pragma unroll
                        for (int t = 0; t <= TL; t += 2) pk_fma_sb<1>(ao[t / 2], win2[(t + 14 - i) / 2], Bp[i / 2]);   // tap i + 1
                    }
                    v2f Hp[TL / 2];
#pragma unroll
                    for (int q = 0; q < TL / 2; q++) Hp[q] = v2f{ae[q].x + ao[q].y, ae[q].y + ao[q + 1].x};
#pragma unroll
                    for (int rr = 0; rr < RL; rr++) {
                        const int ci = 2 * (15 + rr - k + 3);                          // C_{15 + rr - k} (zero pairs outside 0 .. 15)
                        const v2f cj = v2f{A.gabor_lp_C[ci], A.gabor_lp_C[ci + 1]};
#pragma unroll
                        for (int t = 0; t < TL; t++) {
                            if ((t & 1) == 0) pk_fma_bcast<0>(o[rr][t], Hp[t / 2], cj);
                            else pk_fma_bcast<1>(o[rr][t], Hp[t / 2], cj);
                        }
                    }
                }
                double s2v[NV];                                                    // (-1: no such pixel)
#pragma unroll
                for (int t = 0; t < TL; t++)
#pragma unroll
                    for (int rr = 0; rr < RL; rr++) {
                        const double re_ = (double)o[rr][t].x, im_ = (double)o[rr][t].y;
                        s2v[RL * t + rr] = (a0 + (uint32_t)t < w && b0 + (uint32_t)rr < h) ? re_ * re_ + im_ * im_ : -1.0;
                    }
                return_item(s2v, b0, a0);
            };
            const uint32_t tid_h = (uint32_t)here(tid);
            if (tid_h < n_it) do_item(tid_h);
            {
                const double wmx = wave_max_d(lmax), wmn = wave_min_d(lmin);
                if (lane == 0) { s_red[wave * 8] = wmx; s_red[wave * 8 + 1] = wmn; }
                __syncthreads();
                double mx = s_red[0], mn = s_red[1];
                for (int wv = 1; wv < NW; wv++) { mx = s_red[wv * 8] > mx ? s_red[wv * 8] : mx; mn = s_red[wv * 8 + 1] < mn ? s_red[wv * 8 + 1] : mn; }
                __syncthreads();
                // (square roots within one rounding: the bounds below carry 1e-15 relative for them)
                const double e_hi = sqrt(mx), e_lo = sqrt(mn), M2 = 2.0 * (kErrLp * amax + 3e-15 * amax) + 4e-15 * e_hi;
                const double c_lo = e_hi - M2 > 0.0 ? e_hi - M2 : 0.0, c_hi = e_lo + M2;
                lp_hi2 = c_lo * c_lo * (1.0 - 1e-6); lp_lo2 = c_hi * c_hi * (1.0 + 1e-6);       // (1e-6: the first trip's energies wait as fp32)
            }
            first = false;
            if (kept) push_item(keep, kb0, ka0);
            for (uint32_t it = tid_h + kBlk; it < n_it; it += kBlk) do_item(it);
        }
        if (f32 && lpsep) {
        } else if (fbox) {
            const uint32_t n_rp = (h + 1u) / 2u, n_it = n_rp * tpr;
            float smax = -1.0f, smin = __builtin_inff();
            uint32_t cmin = 0;
            for (uint32_t it = (uint32_t)here(tid); it < n_it; it += kBlk) {
                const uint32_t cb = it / n_rp, rpi = it - cb * n_rp, b0 = 2u * rpi, a0 = cb * T;
                const bool down = (rpi & 1u) != 0;
                const uint32_t* const top = s_plane + b0 * pitch + a0;            // padded row b0: the first tap row of output row b0
                const uint32_t* rowp = top + (down ? 14u : 1u) * pitch;
                const int stride = down ? -(int)pitch : (int)pitch;
                float V[T + 16];
                {
                    uint4 uw[W4];
                    lds_load_window<W4>(rowp, uw);
#pragma unroll
                    for (int q = 0; q < W4; q++) {
                        V[4 * q] = __uint_as_float(uw[q].x); V[4 * q + 1] = __uint_as_float(uw[q].y);
                        V[4 * q + 2] = __uint_as_float(uw[q].z); V[4 * q + 3] = __uint_as_float(uw[q].w);
                    }
                }
#pragma unroll 1
                for (int j = 1; j < 14; j++) {
                    rowp += stride;
                    uint4 uw[W4];
                    lds_load_window<W4>(rowp, uw);
#pragma unroll
                    for (int q = 0; q < W4; q++) {
                        V[4 * q] += __uint_as_float(uw[q].x); V[4 * q + 1] += __uint_as_float(uw[q].y);
                        V[4 * q + 2] += __uint_as_float(uw[q].z); V[4 * q + 3] += __uint_as_float(uw[q].w);
                    }
                }
                const bool two_rows = b0 + 1u < h;                                 // (an odd box height: the last pair has one row; padded row b0 + 16 does not exist then)
                {
                    uint4 uz[W4];
                    lds_load_window<W4>(top + 15u * pitch, uz);
#pragma unroll
                    for (int q = 0; q < W4; q++) {
                        V[4 * q] += __uint_as_float(uz[q].x); V[4 * q + 1] += __uint_as_float(uz[q].y);
                        V[4 * q + 2] += __uint_as_float(uz[q].z); V[4 * q + 3] += __uint_as_float(uz[q].w);
                    }
                }
                auto finish = [&](const uint32_t* erow, bool on) {
                    uint4 ue[W4];
                    lds_load_window<W4>(erow, ue);
                    float wd[T + 16];
#pragma unroll
                    for (int q = 0; q < W4; q++) {
                        wd[4 * q] = V[4 * q] + __uint_as_float(ue[q].x); wd[4 * q + 1] = V[4 * q + 1] + __uint_as_float(ue[q].y);
                        wd[4 * q + 2] = V[4 * q + 2] + __uint_as_float(ue[q].z); wd[4 * q + 3] = V[4 * q + 3] + __uint_as_float(ue[q].w);
                    }
                    float sm = 0.0f;                                               // taps i = 0..15 of output t read words t + 1 .. t + 16
#pragma unroll
                    for (int k = 1; k <= 16; k++) sm += wd[k];
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        if (t) sm = (sm - wd[t]) + wd[t + 16];                        // (subtract first: both intermediates are window sums of integers <= 256 x 65535 < 2^24 -> exact; adding first can pass 2^24)
                        if (on && a0 + (uint32_t)t < w) {
                            smax = sm > smax ? sm : smax;
                            if (sm < smin) { smin = sm; cmin = 1; }
                            else if (sm == smin) cmin++;
                        }
                    }
                };
                finish(top, true);
                finish(top + (two_rows ? 16u : 15u) * pitch, two_rows);
            }
            if (smax >= 0.0f) { tmax = (double)smax * box_c; tmin = (double)smin * box_c; n_min = cmin; }
        } else
        for (uint32_t tile = vtid_of((uint32_t)here(tid)); tile < ntiles; tile += kBlk) {
            // column-major: (row block, tile column, row in block); row-major: (row, tile column)
            const uint32_t cb = colmajor ? tile >> 4 : tile, rb = cb / tpr, b = colmajor ? rb * 16u + (tile & 15u) : rb, a0 = (cb - rb * tpr) * T;
            if (b >= h)
                continue;
            double re[T], im[T];
#pragma unroll
            for (int t = 0; t < T; t++) { re[t] = 0.0; im[t] = 0.0; }
            const uint32_t* row = s_plane + (b + 15) * pitch + a0;      // tap row j reads padded row b + 15 - j
            if (ZR && box) {
                // box filter (ShapeArgs::gabor_box_mask): every tap is (c, 0) with c a power of two.  Each product a * c and every
                // partial sum of the reference's scan is then an exact multiple of c below 2^53 c -- no rounding anywhere, whatever
                // the order -- so the response is c times the integer sum of the 16 x 16 window: a sliding sum over the row's
                // words (two integer operations per output and tap row instead of sixteen multiplies and sixteen adds).  32-bit
                // sums hold because this ROI's intensities are below 2^24 (checked above).
                uint32_t S[T];
#pragma unroll
                for (int t = 0; t < T; t++) S[t] = 0;
#pragma unroll 1
                for (int j = 0; j < N; j++, row -= pitch) {
                    uint32_t wd[T + 16];
                    uint4 uw[W4];
                    lds_load_window<W4>(row, uw);
#pragma unroll
                    for (int q = 0; q < W4; q++) {
                        const uint4 u = uw[q];
                        if (MODE >= 3) {                           // (a box filter implies intensities below 2^24: fp32 patterns)
                            wd[4 * q + 0] = (uint32_t)__uint_as_float(u.x); wd[4 * q + 1] = (uint32_t)__uint_as_float(u.y);
                            wd[4 * q + 2] = (uint32_t)__uint_as_float(u.z); wd[4 * q + 3] = (uint32_t)__uint_as_float(u.w);
                        } else {
                            wd[4 * q + 0] = u.x; wd[4 * q + 1] = u.y; wd[4 * q + 2] = u.z; wd[4 * q + 3] = u.w;
                        }
                    }
                    uint32_t sm = 0;                             // taps i = 0..15 of output t read words t + 16 - i = t + 1 .. t + 16
#pragma unroll
                    for (int k = 1; k <= 16; k++) sm += wd[k];
                    S[0] += sm;
#pragma unroll
                    for (int t = 1; t < T; t++) { sm = sm + wd[t + 16] - wd[t]; S[t] += sm; }
                }
#pragma unroll
                for (int t = 0; t < T; t++) re[t] = (double)S[t] * box_c;
            } else if constexpr (f32) {
                // packed fp32: (re, im) of an output in one register pair, a tap's two products in one v_pk_fma_f32
                v2f acc[T];
#pragma unroll
                for (int t = 0; t < T; t++) acc[t] = v2f{0.0f, 0.0f};
                const bank32_ptr_t G32 = bank32 + (size_t)f * N * N * 2;
#pragma unroll 1
                for (int j = 0; j < N; j++, row -= pitch) {
                    if (((zero_rows >> j) & 1u) && ((zero_rows >> (16 + j)) & 1u))
                        continue;                                // (a row of +-0 taps adds nothing to either component)
                    v2f win2[(T + 16) / 2];                      // window words 2 k, 2 k + 1
#pragma unroll
                    for (int q = 0; q < W4; q++) {
                        const uint4 u = ((const uint4*)row)[q];  // (every pair is used whole by the asm below: nothing for the compiler to narrow)
                        win2[2 * q] = v2f{__uint_as_float(u.x), __uint_as_float(u.y)};      // (this copy runs with pf only)
                        win2[2 * q + 1] = v2f{__uint_as_float(u.z), __uint_as_float(u.w)};
                    }
                    const bank32_ptr_t Gj = G32 + j * N * 2;
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        const v2f g = v2f{Gj[2 * i], Gj[2 * i + 1]};
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            constexpr int kOdd = 1;
                            const int e = t + 16 - i;            // window word of output t and tap i (a constant once unrolled)
                            if ((e & kOdd) == 0) pk_fma_bcast<0>(acc[t], win2[e >> 1], g);
                            else pk_fma_bcast<1>(acc[t], win2[e >> 1], g);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < T; t++) { re[t] = (double)acc[t].x; im[t] = (double)acc[t].y; }
            } else
#pragma unroll 1
            for (int j = 0; j < N; j++, row -= pitch) {
                // tap rows whose real (imaginary) parts are all +-0 leave re (im) as it is: see ShapeArgs::gabor_zero_rows
                // (MODE 3 keeps the row tests out of this copy -- it runs for the odd ROI only, and three variants of the tap block
                //  next to the fp32 copy cost the kernel 45 spilled registers; a row of +-0 taps adds +-0: the same bits)
                constexpr bool ZRU = ZR && MODE < 3;
                const bool re0 = ZRU && ((zero_rows >> j) & 1u), im0 = ZRU && ((zero_rows >> (16 + j)) & 1u);
                if (re0 && im0)
                    continue;
                double win[T + 16];
                uint4 uw[W4];
                lds_load_window<W4>(row, uw);
#pragma unroll
                for (int q = 0; q < W4; q++) {
                    const uint4 u = uw[q];
                    if (MODE >= 3 && pf) {
                        win[4 * q + 0] = (double)__uint_as_float(u.x); win[4 * q + 1] = (double)__uint_as_float(u.y);
                        win[4 * q + 2] = (double)__uint_as_float(u.z); win[4 * q + 3] = (double)__uint_as_float(u.w);
                    } else {
                        win[4 * q + 0] = (double)u.x; win[4 * q + 1] = (double)u.y;
                        win[4 * q + 2] = (double)u.z; win[4 * q + 3] = (double)u.w;
                    }
                }
                const bank_ptr_t Gj = G + j * N * 2;
                auto taps = [&](auto do_re_c, auto do_im_c, auto fuse_c) {
                    constexpr bool DO_RE = decltype(do_re_c)::value, DO_IM = decltype(do_im_c)::value, FUSED = decltype(fuse_c)::value;
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        const double gr = DO_RE ? Gj[2 * i] : 0.0, gi = DO_IM ? Gj[2 * i + 1] : 0.0;
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            const double av = win[t + 16 - i];   // padded column a0 + t + 16 - i = image column a0 + t + 8 - i
                            if (FUSED) {
                                if (DO_RE) re[t] = __builtin_fma(av, gr, re[t]);
                                if (DO_IM) im[t] = __builtin_fma(av, gi, im[t]);
                            } else {
                                if (DO_RE) re[t] += av * gr;     // C[ip]   += a * wr   (gabor.cpp:374)
                                if (DO_IM) im[t] += av * gi;     // C[ip+1] += a * wi   (:377)
                            }
                        }
                    }
                };
                // (a row whose real parts alone vanish does not occur in a Gabor bank -- cos(x' f0) has no exact zeros on the tap
                //  grid -- and takes the full block)
                if (ZRU && im0) taps(std::true_type{}, std::false_type{}, fuse_c);
                else taps(std::true_type{}, std::true_type{}, fuse_c);
            }
#pragma unroll
            for (int t = 0; t < T; t++) {
                if (a0 + t >= w)
                    continue;
                const double s2 = re[t] * re[t] + im[t] * im[t];
                if (lp_cand) {
                    bool cand;
                    if constexpr (f32) cand = s2 >= lp_hi2 || s2 <= lp_lo2;
                    else {
                        // (the same bound as below, against Bs and bs instead of the threshold)
                        const double e = sqrt(s2), m0 = __builtin_fma(1e-15, e, kErr * amax);
                        cand = e >= Bs - __builtin_fma(1e-15, Bs, m0) || e <= bs + __builtin_fma(1e-15, bs, m0);
                    }
                    if (cand) {
                        const uint32_t k = atomicAdd(&s_redo[0], 1u);
                        if (k < (uint32_t)kGaborRedoCap) s_redo[1 + k] = b * w + a0 + (uint32_t)t;
                    }
                } else if (f == 0) {
                    const double e = sqrt(s2);                              // :505
                    tmax = e > tmax ? e : tmax;
                    if (e < tmin) { tmin = e; n_min = 1; }
                    else if (e == tmin) n_min++;
                } else if (MODE >= 2 && fuse) {
                    // The reference decides fl(e / max) > thr (:117).  With T = thr * max: an e above T (1 + 4 u) gives a quotient above
                    // thr, one below T (1 - 4 u) a quotient below it, whatever the rounding of the division; and the fused e is within
                    // |d re| + |d im| + 4 u e <= 0.9e-13 a_max + 4 u e of the reference's (see the kernel's header).  So outside
                    // |e - T| <= 2.5e-13 a_max + 1e-15 (e + T) the comparison of the fused e with T IS the reference's decision -- no
                    // division per pixel -- and inside it the pixel is recomputed.
                    bool redo, above;
                    if constexpr (f32) { above = s2 > thr_hi2; redo = !above && s2 >= thr_lo2; }
                    else {
                        const double e = sqrt(s2), dlt = e - thr_max;
                        redo = fabs(dlt) <= __builtin_fma(1e-15, e, thr_slack);
                        above = dlt > 0.0;
                    }
                    if (redo) {
                        const uint32_t k = atomicAdd(&s_redo[0], 1u);
                        if (k < (uint32_t)kGaborRedoCap) s_redo[1 + k] = b * w + a0 + (uint32_t)t;
                    } else if (above)
                        sc++;
                } else if (sqrt(s2) / maxval > A.gabor_thr)             // :117
                    sc++;
            }
        }
        if (lp_cand) {                                    // the candidates for maximum / minimum, with the reference's arithmetic
            __syncthreads();
            const uint32_t nr = s_redo[0];
            NYX_GABOR_PHASE_EXIT(A.dbg_phase == 6 && tid == 0, diag_cnt[0] = nr);
            lp_overflow = nr > (uint32_t)kGaborRedoCap;
            if (!lp_overflow)
                for (uint32_t k = (uint32_t)here(tid); k < nr; k += kBlk) {
                    const uint32_t p = s_redo[1 + k], b = p / w, a = p - b * w;
                    const double e = exact_energy(a, b, G);
                    tmax = e > tmax ? e : tmax;
                    if (e < tmin) { tmin = e; n_min = 1; }
                    else if (e == tmin) n_min++;
                }
            __syncthreads();
            if (lp_overflow) return true;                 // (block-uniform) the caller runs the filter again, unfused
        } else
        if (MODE >= 2 && fuse && f != 0) {                 // the pixels too close to the threshold, with the reference's arithmetic
            __syncthreads();
            const uint32_t nr = s_redo[0];
            if (nr > (uint32_t)kGaborRedoCap) {
                // more of them than the list holds (a flat field whose common energy IS the threshold): every pixel of the box is
                // decided by the reference's arithmetic -- slow, rare, and the same answer
                sc = 0;
                for (uint32_t p = (uint32_t)here(tid); p < area; p += kBlk) {
                    const uint32_t b = p / w, a = p - b * w;
                    if (exact_energy(a, b, G) / maxval > A.gabor_thr) sc++;
                }
            } else
                for (uint32_t k = (uint32_t)here(tid); k < nr; k += kBlk) {
                    const uint32_t p = s_redo[1 + k], b = p / w, a = p - b * w;
                    if (exact_energy(a, b, G) / maxval > A.gabor_thr) sc++;
                }
            __syncthreads();
        }
        if (f == 0) {
            tmax = wave_max_d(tmax);
            const double wmin = wave_min_d(tmin);
            const uint32_t wcnt = (uint32_t)wave_sum_u64(tmin == wmin ? n_min : 0u);
            if (lane == 0) { s_red[wave * 8] = tmax; s_red[wave * 8 + 1] = wmin; s_red[wave * 8 + 2] = (double)wcnt; }
            __syncthreads();
            double mx = s_red[0], mn = s_red[1];
            for (int wv = 1; wv < NW; wv++) {
                mx = s_red[wv * 8] > mx ? s_red[wv * 8] : mx;
                mn = s_red[wv * 8 + 1] < mn ? s_red[wv * 8 + 1] : mn;
            }
            uint32_t at_min = 0;
            for (int wv = 0; wv < NW; wv++)
                if (s_red[wv * 8 + 1] == mn) at_min += (uint32_t)s_red[wv * 8 + 2];
            __syncthreads();
            if (mx == mn) {                               // gabor.cpp:91-96
                for (int c = tid; c < nF; c += kBlk)
                    o[c] = A.soft_nan;
                return false;
            }
            maxval = mx;
            n_min = area - at_min;                        // baseline score, :99-102: pixels with e > min
        } else {
            const uint32_t tot = (uint32_t)wave_sum_u64(sc);
            if (lane == 0) s_red[wave * NYXHIP_MAX_GABOR_FILTERS + f - 1] = (double)tot;   // [wave][filter]; read after the final barrier
        }
        return true;
    };
    // MODE 4: the band-pass filters' screening pass on the matrix pipe (measured standalone in tools/gabor_mfma_probe.hip).  After the
    // low-pass filter the fp32 plane is rewritten IN PLACE as two planes of f16 digits -- intensity = main + rest, main the top eleven
    // significant bits (exact in f16 below 2^16), rest the bits under them (a zero plane, skipped, when the ROI's intensities stay below
    // 2^11) -- and four filters at a time run as v_mfma_f32_16x16x32_f16: M = 16 box rows at one column, K = two tap rows, N = (filter,
    // re / im) x (hi, lo part of the tap x 2^14: ensure_gabor_bank), both digits into one fp32 accumulator.  A lane's eight K-elements
    // are eight consecutive pixels of a plane row: a window of twelve pixels, three 8-byte LDS reads, serves four columns.
    // Error of a screened component against the exact sum: taps (hi + lo) within 2^-22 relative + 2^-28 absolute even if f16
    // subnormals were flushed (256 taps: <= 2.2e-6 a_max).  Accumulation, as measured (tools/mfma_rounding_probe.hip,
    // profiles/r05_mfma_rounding_probe.txt): the instruction sums its 32 products and the addend BEFORE it rounds, to nearest even,
    // each term first cut to a grid of 2^-24 of the largest term -- so one instruction errs by less than 33 * 2^-24 of the absolute sum
    // that has gone in so far (observed worst over 20 000 random instructions: 3.1 * 2^-24).  The rest plane's eight instructions run
    // first (absolute sum <= 2^-11 of the whole), then the main plane's eight: <= 8 * 33 * 2^-24 (1 + 2^-10) sum |a w|, and
    // sqrt((sum |w_re|)^2 + (sum |w_im|)^2) <= sum |w| = 1 (the bank is L1-normalised), so the screened energy is within 1.58e-5 a_max
    // + 2.2e-6 a_max (+ the epilogue's four fp32 roundings, 3e-7 a_max) of the reference's: kErr = 1.85e-5, half the band of the
    // packed-fp32 pass; the Gabor probe measures 1.0e-8 a_max.
    // Pixels inside the band go to the same list and are recomputed by the reference's arithmetic from the digit planes.
    auto run_bands_mfma = [&]() {
        // lists of band pixels, one per filter, for ALL groups: [16 counters][n filters x kSub entries]; they are recomputed in one run
        // behind the last group (per group it was one wave-long recomputation each: two of the three on an 8-filter bank's small ROIs)
        const uint32_t n_lists = (uint32_t)(4 * ((nF + 3) / 4)), kSub = ((uint32_t)kGaborRedoCap - 12u) / n_lists;
        const uint32_t maxi = A.max_inten[roi];
        const uint32_t low_bits = maxi < 2048u ? 0u : (uint32_t)(21 - __builtin_clz(maxi)), low_mask = (1u << low_bits) - 1u;   // (bits - 11)
        const bool two = low_bits != 0;
        // the rewrite holds the whole fp32 plane in registers across a barrier: 32 words per thread for the common boxes, 64 for the
        // largest LDS-resident ones (two builds of the loop: the wide one costs the metric boxes 0.7 ms per 196 k ROIs in idle trips)
        auto rewrite = [&](auto hold_c) {
            constexpr int kHold = decltype(hold_c)::value;
            uint32_t hold[kHold];
            const uint32_t tid_r = (uint32_t)here(tid);
#pragma unroll
            for (int q = 0; q < kHold; q++) { const uint32_t i = tid_r + (uint32_t)q * kBlk; hold[q] = i < words ? s_plane[i] : 0u; }
            __syncthreads();
            _Float16* const dp = (_Float16*)s_plane;
#pragma unroll
            for (int q = 0; q < kHold; q++) {
                const uint32_t i = tid_r + (uint32_t)q * kBlk;
                if (i < words) {
                    const uint32_t v = (uint32_t)__uint_as_float(hold[q]);
                    dp[i] = (_Float16)(float)(v & ~low_mask);
                    dp[words + i] = (_Float16)(float)(v & low_mask);
                }
            }
        };
        if (words <= 32u * kBlk) rewrite(std::integral_constant<int, 32>{});
        else rewrite(std::integral_constant<int, 64>{});
        NYX_GABOR_PHASE_EXIT(A.dbg_phase == 3, return);
        constexpr double kErr = 1.85e-5, kScale2 = kGaborTapScale * kGaborTapScale;
        const double thr_max = A.gabor_thr * maxval, S_thr = __builtin_fma(kErr, amax, 1e-15 * thr_max) + 3e-15 * amax;
        const double t_lo = thr_max - S_thr > 0.0 ? thr_max - S_thr : 0.0, t_hi = thr_max + S_thr;
        // (squared, in the accumulators' scale, rounded outwards: the conversion to fp32 moves a bound by 6e-8 relative at most)
        const float lo2f = (float)(t_lo * t_lo * kScale2 * (1.0 - 3e-7)), hi2f = (float)(t_hi * t_hi * kScale2 * (1.0 + 3e-7));
        const uint32_t n_x4 = w / 4u + 1u, n_rt = (h + 15u) / 16u, n_units = n_x4 * n_rt;   // columns 4 x4 - 1 .. 4 x4 + 2, rows 16 rt .. 16 rt + 15
        const int lane_h = here(lane), nn = lane_h & 15, kb = lane_h >> 4;
        const lds_f16_t* const dplane = (const lds_f16_t*)s_plane;
        const gabor_h8* const ops = (const gabor_h8*)A.gabor_bank16;
        for (int g = 0; 4 * g < nF; g++) {
            if (g == 0) { if (tid < 16) s_redo[tid] = 0; __syncthreads(); }
            gabor_h8 Bw[8];
#pragma unroll
            for (int jp = 0; jp < 8; jp++) Bw[jp] = ops[(g * 8 + jp) * 64 + lane_h];
            const int f_lane = 1 + 4 * g + ((nn & 7) >> 1);                        // this lane's filter in the epilogue
            const uint32_t fl_lane = 4u * (uint32_t)g + ((uint32_t)(nn & 7) >> 1);   // this lane's list
            const bool col_ok = (nn & 9) == 0 && f_lane <= nF;                     // (even column below 8: re^2 + im^2 of hi + lo lands there)
            uint32_t cnt = 0;
            // (the thresholds of a lane whose column carries no energy are infinite: its comparisons are false)
            const float inf = __builtin_inff(), hi_lane = col_ok ? hi2f : inf, lo_lane = col_ok ? lo2f : inf;
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(3))) u32x2_t* lds_u32x2_ptr;
            // four MFMAs of one tap-row pair and one digit plane: the twelve-pixel window at `rowp`, columns 4 x4 - 1 + (0 .. 3)
            auto pair_mfma = [&](const lds_f16_t* rowp, const gabor_h8 B, gabor_f4 (&C)[4], auto zero_c) {   // zero_c: the accumulators start here (a literal zero addend: no moves)
                // (the window's words 1 .. 4 are read a second time, into registers of their own: the operand of column + 2 must be an
                //  aligned register tuple, and assembling it from the first read costs four moves on the vector pipe per tap-row pair --
                //  the LDS has the cycles to spare, the vector pipe has not)
                typedef u32x2_t __attribute__((aligned(4))) u32x2_a4_t;
                typedef const __attribute__((address_space(3))) u32x2_a4_t* lds_u32x2_a4_ptr;
                const u32x2_t q0 = *(lds_u32x2_ptr)rowp, q1 = *(lds_u32x2_ptr)(rowp + 4);
                const u32x2_t p0 = *(lds_u32x2_a4_ptr)(rowp + 2), p1 = *(lds_u32x2_a4_ptr)(rowp + 6);
                const uint32_t w5 = *(const __attribute__((address_space(3))) uint32_t*)(rowp + 10);
                uint4 av[4];
                av[0] = uint4{q0.x, q0.y, q1.x, q1.y};
                av[2] = uint4{p0.x, p0.y, p1.x, p1.y};
                av[1] = uint4{__builtin_amdgcn_alignbit(p0.x, q0.x, 16), __builtin_amdgcn_alignbit(p0.y, q0.y, 16), __builtin_amdgcn_alignbit(p1.x, q1.x, 16), __builtin_amdgcn_alignbit(p1.y, q1.y, 16)};
                av[3] = uint4{__builtin_amdgcn_alignbit(q1.x, p0.x, 16), __builtin_amdgcn_alignbit(q1.y, p0.y, 16), __builtin_amdgcn_alignbit(p1.y, p1.x, 16), __builtin_amdgcn_alignbit(w5, p1.y, 16)};
#pragma unroll
                for (int xs = 0; xs < 4; xs++) {
                    gabor_h8 Av;
                    __builtin_memcpy(&Av, &av[xs], 16);
                    if constexpr (decltype(zero_c)::value) C[xs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Av, B, gabor_f4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                    else C[xs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Av, B, C[xs], 0, 0, 0);
                }
            };
            const uint32_t step = 2u * pitch;                                      // plane elements per tap-row pair
            for (uint32_t u = (uint32_t)wave; u < n_units; u += (uint32_t)NW) {
                const uint32_t rt = u / n_x4, x4 = u - rt * n_x4;
                // (rows beyond the box repeat its last row: their outputs are dropped, their reads stay inside the plane)
                const uint32_t brow = min(16u * rt + (uint32_t)nn, h - 1u);
                const lds_f16_t* const base = dplane + (brow + (uint32_t)(kb >> 1)) * pitch + 8u * (uint32_t)(kb & 1) + 4u * x4;
                gabor_f4 C[4];
                // The rest digits FIRST: an instruction's rounding is relative to what it adds up, and while only the rest plane has
                // gone in that is 2^-11 of the whole -- the bound above then counts the eight instructions of the main plane only.
                if (two) {
                    const lds_f16_t* rp = base + words;
                    pair_mfma(rp, Bw[0], C, std::true_type{});
                    rp += step;
#pragma unroll
                    for (int jp = 1; jp < 8; jp++, rp += step) pair_mfma(rp, Bw[jp], C, std::false_type{});
                    rp = base;
#pragma unroll
                    for (int jp = 0; jp < 8; jp++, rp += step) pair_mfma(rp, Bw[jp], C, std::false_type{});
                } else {
                    const lds_f16_t* rp = base;
                    pair_mfma(rp, Bw[0], C, std::true_type{});
                    rp += step;
#pragma unroll
                    for (int jp = 1; jp < 8; jp++, rp += step) pair_mfma(rp, Bw[jp], C, std::false_type{});
                }
                // epilogue: lane (column nn, rows 4 kb + r): hi-part column + lo-part column (eight lanes on), re^2 + im^2 (the lane beside)
                const uint32_t b0 = 16u * rt + 4u * (uint32_t)kb;
                float hi_r[4], lo_r[4];
#pragma unroll
                for (int r = 0; r < 4; r++) { const bool rv = b0 + (uint32_t)r < h; hi_r[r] = rv ? hi_lane : inf; lo_r[r] = rv ? lo_lane : inf; }
#pragma unroll
                for (int xs = 0; xs < 4; xs++) {
                    const uint32_t a = 4u * x4 + (uint32_t)xs - 1u;                // (column -1 of the first group: no such pixel)
                    if (a >= w) continue;                                          // (the same in every lane)
                    float e2[4];
                    unsigned long long band = 0;                                   // lanes with a pixel inside the band (kept as wave masks: scalar logic)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float c = C[xs][r];
                        c += __uint_as_float(dpp_perm<0x128>(__float_as_uint(c)));          // row_ror:8
                        const float sq = c * c;
                        e2[r] = sq + __uint_as_float(dpp_perm<0xB1>(__float_as_uint(sq)));  // quad_perm [1, 0, 3, 2]
                        const bool above = e2[r] > hi_r[r];
                        cnt += above ? 1u : 0u;
                        band |= __builtin_amdgcn_ballot_w64(e2[r] >= lo_r[r]) & ~__builtin_amdgcn_ballot_w64(above);
                    }
                    if (band) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (!(e2[r] > hi_r[r]) && e2[r] >= lo_r[r]) {
                                const uint32_t k = atomicAdd(&s_redo[fl_lane], 1u);
                                if (k < kSub) s_redo[16u + fl_lane * kSub + k] = (b0 + (uint32_t)r) * w + a;
                            }
                    }
                }
            }
            NYX_GABOR_PHASE_EXIT(A.dbg_phase == 4, continue);
            // the four 16-lane rows of the wave hold the same columns
            cnt += __shfl_xor(cnt, 16, 64);
            cnt += __shfl_xor(cnt, 32, 64);
            if (lane < 16 && col_ok) s_red[wave * NYXHIP_MAX_GABOR_FILTERS + f_lane - 1] = (double)cnt;   // [wave][filter]
        }
        __syncthreads();
        {
            const lds_u32_t* const pl = (const lds_u32_t*)s_plane;
            // All filters' lists as ONE run over the workgroup's threads (a handful of pixels each on ordinary data: as a run per filter
            // they cost a wave-long recomputation of a few live lanes each; the taps then differ from lane to lane and come through
            // vector loads).  A list that overflowed is left out here and its filter recomputed over the whole box below.
            uint32_t n_all = 0;
            for (int q = 0; q < nF; q++) { const uint32_t nr = s_redo[q]; n_all += nr > kSub ? 0u : nr; }
            NYX_GABOR_PHASE_EXIT(A.dbg_phase == 6 && tid < 4, diag_cnt[1 + tid] = s_redo[tid]);
            for (uint32_t k0 = (uint32_t)wave * 64u; k0 < n_all; k0 += kBlk) {          // (the same trips in every lane of a wave: the ballots below)
                uint32_t k = k0 + (uint32_t)lane, fl = 0;
                const bool live = k < n_all;
                for (int q = 0; q < nF - 1; q++) {
                    const uint32_t nr = s_redo[q], nq = nr > kSub ? 0u : nr;
                    if (live && fl == (uint32_t)q && k >= nq) { k -= nq; fl = (uint32_t)q + 1u; }
                }
                bool hit = false;
                if (live) {
                    const uint32_t p = s_redo[16u + fl * kSub + k], b = p / w, a = p - b * w;
                    hit = gabor_exact_energy<2, false>(pl, pitch, a, b, bank + (size_t)(1 + (int)fl) * N * N * 2, words) / maxval > A.gabor_thr;
                }
                for (uint32_t q = 0; q < (uint32_t)nF; q++) {
                    const uint32_t c = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit && fl == q));
                    if (lane == 0 && c) s_red[wave * NYXHIP_MAX_GABOR_FILTERS + (int)q] += (double)c;
                }
            }
            for (int fl = 0; fl < nF; fl++) {
                if (s_redo[fl] <= kSub)
                    continue;
                // more pixels in the band than the list holds: this filter over the whole box with the reference's arithmetic
                const bank_ptr_t G = bank + (size_t)(1 + fl) * N * N * 2;
                uint32_t sc = 0;
                for (uint32_t p = (uint32_t)here(tid); p < area; p += kBlk) {
                    const uint32_t b = p / w, a = p - b * w;
                    if (gabor_exact_energy<2>(pl, pitch, a, b, G, words) / maxval > A.gabor_thr) sc++;
                }
                const uint32_t tot = (uint32_t)wave_sum_u64(sc);
                if (lane == 0) s_red[wave * NYXHIP_MAX_GABOR_FILTERS + fl] = (double)tot;
            }
            __syncthreads();
        }
    };
    for (int f = 0; f <= nF; f++) {
        if constexpr (MODE == 4) {
            NYX_GABOR_PHASE_EXIT(f == 1 && A.dbg_phase == 2, return);
            if (f == 1 && pf && A.max_inten[roi] < 65536u && words <= 64u * kBlk && A.gabor_bank16) { run_bands_mfma(); break; }
        }
        bool go;
        if constexpr (MODE == 0) go = run_filter(f, std::false_type{});
        else if constexpr (MODE == 1) go = run_filter(f, std::true_type{});
        else if (MODE >= 3 && !(amax < 16777216.0)) go = run_filter(f, std::false_type{});   // an intensity that fp32 does not hold exactly: the reference's arithmetic throughout
        else if (f == 0 && NW == 1 && !A.gabor_lp_sep) go = run_filter(0, std::false_type{});   // one-wave launches (small ROIs): two recomputation calls cost more than the unfused low-pass (the separable pass needs none)
        else {
            go = run_filter(f, std::true_type{});
            if (f == 0 && lp_overflow) {                  // too many candidates (ties): the low-pass filter again, the reference's arithmetic throughout
                tmax = -1.0; tmin = 1.7976931348623157e308; n_min = 0;
                go = run_filter(0, std::false_type{});
            }
        }
        if (!go) return;
    }
    const double baseline = (double)n_min;
    __syncthreads();
#ifdef NYXHIP_GABOR_PHASE_EXITS
    if (A.dbg_phase == 6) {                              // the output row carries the list lengths: low-pass candidates, band pixels of filters 1 .. 3
        for (int k = tid; k < nF; k += kBlk) o[k] = (double)diag_cnt[k < 4 ? k : 0];
        return;
    }
#endif
    for (int k = tid; k < nF; k += kBlk) {
        double scv = 0;
        for (int wv = 0; wv < NW; wv++) scv += s_red[wv * NYXHIP_MAX_GABOR_FILTERS + k];
        o[k] = scv / baseline;                            // :121
    }
}

// =========================================================================================
// Zernike (order 9: 30 magnitudes)
// =========================================================================================
constexpr int kZL = 9;

// H1/H2/H3 of zernike.cpp:234-248, constant-folded per (n, m)
__device__ __forceinline__ constexpr double zH3(int n, int m) { return -(double)(4.0 * (m + 2.0) * (m + 1.0)) / (double)((n + m + 2.0) * (n - m)); }
__device__ __forceinline__ constexpr double zH2(int n, int m) { return ((double)(zH3(n, m) * (n + m + 4.0) * (n - m - 2.0)) / (double)(4.0 * (m + 3.0))) + (m + 2.0); }
__device__ __forceinline__ constexpr double zH1(int n, int m)
{
    return ((double)((m + 4.0) * (m + 3.0)) / 2.0) - ((m + 4.0) * zH2(n, m)) + ((double)(zH3(n, m) * (n + m + 6.0) * (n - m - 4.0)) / 8.0);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void roi_zernike_kernel(const ShapeArgs A)
{
    constexpr int kBlk = NW * 64;
    __shared__ double s_red[NW * 64];
    __shared__ unsigned long long s_mom[NW * 4];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t* const s_xy = (uint32_t*)lds_raw;                 // [zern_px_cap] x | y << 16 of the staged cloud
    uint32_t* const s_v = s_xy + A.L.zern_px_cap;              // [zern_px_cap] intensities
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t roi;                      // (no LDS-resident state that limits the ROI: never part of a workspace launch)
    if (!roi_of_slot(A.sp, blockIdx.x, A.n_roi, roi))
        return;
    const uint64_t off = A.px_offset[roi];
    const uint32_t npx = (uint32_t)(A.px_offset[roi + 1] - off);
    const uint32_t w = A.bbox_w[roi], h = A.bbox_h[roi];
    double* const o = A.out + roi * A.ld + A.col_zernike;
    if (!roi_in_launch(A.sp, npx, w, h, A.max_inten[roi] - A.min_inten[roi]))
        return;                                       // another launch of this call serves the ROI's size class
    if (npx == 0 || A.max_inten[roi] == A.min_inten[roi]) {   // zernike.cpp:348-356
        for (int c = tid; c < 30; c += kBlk)
            o[c] = npx == 0 ? __longlong_as_double(0x7ff8000000000000LL) : A.soft_nan;
        return;
    }
    // centroid moments (zernike.cpp:216-230): sums of integers, exact in any order
    // The same sweep stages the cloud in LDS (when it fits), so that the moment sweep below is not one HBM round trip per
    // pixel and lane.
    const bool staged = npx <= A.L.zern_px_cap;
    unsigned long long m00 = 0, m10 = 0, m01 = 0;
    for_each_cloud_pixel<kBlk>(A.inten + off, A.x + off, A.y + off, npx, tid, [&](uint32_t i, uint32_t vi, uint32_t xi, uint32_t yi) {
        if (staged) { s_xy[i] = xi | (yi << 16); s_v[i] = vi; }
        unsigned long long v = vi;
        m00 += v;
        m10 += ((unsigned long long)xi + 1) * v;
        m01 += ((unsigned long long)yi + 1) * v;
    });
    m00 = wave_sum_u64(m00); m10 = wave_sum_u64(m10); m01 = wave_sum_u64(m01);
    if (lane == 0) { s_mom[wave * 4] = m00; s_mom[wave * 4 + 1] = m10; s_mom[wave * 4 + 2] = m01; }
    __syncthreads();
    unsigned long long t00 = 0, t10 = 0, t01 = 0;
    for (int wv = 0; wv < NW; wv++) { t00 += s_mom[wv * 4]; t10 += s_mom[wv * 4 + 1]; t01 += s_mom[wv * 4 + 2]; }
    const double sum = (double)t00;
    const double m10_m00 = (double)t10 / sum, m01_m00 = (double)t01 / sum;
    const double rad = (double)(w < h ? w : h);                // N = min(width, height), :185-195
    // pixel weight / (sum * pi) as one multiplier (the reference divides by both per (n, pixel): :298; 1-2 ulp apart, the
    // moments carry a 1e-5 tolerance).  Everything that DECIDES something -- x, y, r and the unit-disc test -- keeps the
    // reference's exact divisions.
    const double inv_sum_pi = 1.0 / (sum * 3.14159265358979323846);
    const double inv_rad = 1.0 / rad;

    // The reference accumulates, per pixel, A_nm += (n + 1) f R_nm(r) e^{-i (m + 1) theta} for the 30 pairs (n, m) -- its COST[m] / SINT[m]
    // are cos / sin of (m + 1) theta (zernike.cpp:270-277) -- with R_nm from the H1 / H2 / H3 recurrence (:234-248, :283-296), which
    // yields the standard radial polynomial R_nm(r) = sum_s (-1)^s (n - s)! / (s! ((n + m) / 2 - s)! ((n - m) / 2 - s)!) r^(n - 2 s).
    // Term by term r^k e^{-i (m + 1) theta} = r^(k - m - 1) (x - i y)^(m + 1) with k - m even, so all 30 moments are fixed
    // combinations of the 30 complex sums  T[m][j] = sum_pixels f / r * (r^2)^j * (x - i y)^(m + 1),  j = 0 .. (9 - m) / 2:
    // per pixel one complex power recurrence (4 operations per m), five weights and two multiply-adds per sum -- ~125 vector
    // instructions instead of ~290 (the radial recurrence, a product and two multiply-adds per (n, m)).  The combination runs once
    // per ROI; its cancellation (coefficients up to 630) costs three of fp64's sixteen digits: 1e-13 against the per-pixel recurrence, the moments
    // carry a 1e-5 tolerance (the reference's regression vector: 1e-9 absolute).
    constexpr int kZT = 30;                             // pairs (m, j): m = 0 .. 9, j = 0 .. (9 - m) / 2
    double TR[kZT], TI[kZT];
#pragma unroll
    for (int k = 0; k < kZT; k++) { TR[k] = 0.0; TI[k] = 0.0; }

    for (uint32_t i = tid; i < npx; i += kBlk) {
        uint32_t xi, yi, vi;
        if (staged) { const uint32_t xy = s_xy[i]; xi = xy & 0xFFFFu; yi = xy >> 16; vi = s_v[i]; }
        else { xi = A.x[off + i]; yi = A.y[off + i]; vi = A.inten[off + i]; }
        // x, y, r (:254, :262) decide one thing: whether the pixel lies in the unit disc.  They are formed with a reciprocal of the
        // radius and a reciprocal square root (1-2 ulp from the reference's two divisions and its sqrt; the moments carry a 1e-5
        // tolerance); a pixel within 1e-12 of either edge of the test is redone with the reference's exact operations, so the
        // decision -- the only discontinuous step -- is the reference's in every case.
        const double dxp = (double)((int)xi + 1) - m10_m00, dyp = (double)((int)yi + 1) - m01_m00;
        double x = dxp * inv_rad, y = dyp * inv_rad;
        double r2 = x * x + y * y, inv_r = frsq(r2), r = r2 * inv_r;
        if (!(r >= 1e-12) || fabs(r - 1.0) < 1e-12) {
            x = dxp / rad; y = dyp / rad;
            r2 = x * x + y * y; r = sqrt(r2); inv_r = 1.0 / r;
        }
        if (r < 2.2204460492503131e-16 || r > 1.0)
            continue;
        {   // from here on nothing decides anything: products feed sums directly (contraction allowed, 1e-5 tolerance)
#pragma clang fp contract(fast)
        double P[5];                                   // f / r * (r^2)^j
        P[0] = (double)vi * inv_sum_pi * inv_r;
#pragma unroll
        for (int j = 1; j < 5; j++) P[j] = P[j - 1] * r2;
        double cr = x, ci = -y;                        // (x - i y)^(m + 1)
        int k = 0;
#pragma unroll
        for (int m = 0; m <= kZL; m++) {
            if (m) { const double nr = cr * x + ci * y, ni = ci * x - cr * y; cr = nr; ci = ni; }
#pragma unroll
            for (int j = 0; j <= (kZL - m) / 2; j++, k++) { TR[k] += P[j] * cr; TI[k] += P[j] * ci; }
        }
        }
    }
    // reduce the 30 complex sums over the workgroup: slot 2k = real, 2k + 1 = imaginary part of sum k, added up across the wave by
    // one transposed reduction (lane L receives the total of slot L)
    {
        double slots[64];
#pragma unroll
        for (int k = 0; k < kZT; k++) { slots[2 * k] = TR[k]; slots[2 * k + 1] = TI[k]; }
#pragma unroll
        for (int q = 60; q < 64; q++) slots[q] = 0.0;
        s_red[wave * 64 + lane] = wave_transpose_sum64(slots, lane);
    }
    __syncthreads();
    if (NW > 1) {
        double tot = 0.0;
        if (tid < 64) for (int wv = 0; wv < NW; wv++) tot += s_red[wv * 64 + tid];
        __syncthreads();
        if (tid < 64) s_red[tid] = tot;
        __syncthreads();
    }
    if (tid < 30) {
        // output tid = the (n, m) pair number tid in the reference's order (n ascending, m ascending, n - m even)
        int n = 0, m = 0;
        {
            int q = tid;
            for (n = 0; n <= kZL; n++) { const int cnt = n / 2 + 1; if (q < cnt) { m = (n & 1) + 2 * q; break; } q -= cnt; }
        }
        // first sum of column m: T[m][0] sits at index sum_{m' < m} ((9 - m') / 2 + 1)
        int base = 0;
        for (int mm = 0; mm < m; mm++) base += (kZL - mm) / 2 + 1;
        double vr = 0.0, vi = 0.0;
        // coefficient of r^k, k = n - 2 s: (-1)^s (n - s)! / (s! ((n + m) / 2 - s)! ((n - m) / 2 - s)!)   (exact integers below 2^53)
        for (int s_ = 0; s_ <= (n - m) / 2; s_++) {
            double c = 1.0;
            for (int q = 2; q <= n - s_; q++) c *= (double)q;
            for (int q = 2; q <= s_; q++) c /= (double)q;
            for (int q = 2; q <= (n + m) / 2 - s_; q++) c /= (double)q;
            for (int q = 2; q <= (n - m) / 2 - s_; q++) c /= (double)q;
            if (s_ & 1) c = -c;
            const int j = (n - 2 * s_ - m) / 2;
            vr += c * s_red[2 * (base + j)];
            vi += c * s_red[2 * (base + j) + 1];
        }
        vr *= (double)(n + 1); vi *= (double)(n + 1);
        o[tid] = fabs(sqrt(vr * vr + vi * vi));                // zernike.cpp:335-337
    }
}

// Small ROIs (the DSB2018-shaped ones of BASELINE.json configs[4]: 10x9 ... 16x14 px) get one wave per ROI;
// anything larger four waves.
int launch_roi_shape(const ShapeArgs& a, void* stream, uint32_t grid)
{
    static DeviceOnce optin;
    if (int orc = optin.run([]() -> int {
        hipError_t e = hipFuncSetAttribute((const void*)roi_gabor_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)roi_features_max_lds());
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)roi_gabor_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)roi_features_max_lds());
        const void* tiled[] = {(const void*)roi_gabor_tiled_kernel<8, 4, 0, false>, (const void*)roi_gabor_tiled_kernel<4, 1, 0, false>,
                               (const void*)roi_gabor_tiled_kernel<8, 4, 0, true>, (const void*)roi_gabor_tiled_kernel<4, 1, 0, true>,
                               (const void*)roi_gabor_tiled_kernel<8, 4, 1, false>, (const void*)roi_gabor_tiled_kernel<4, 1, 1, false>,
                               (const void*)roi_gabor_tiled_kernel<8, 4, 1, true>, (const void*)roi_gabor_tiled_kernel<4, 1, 1, true>,
                               (const void*)roi_gabor_tiled_kernel<8, 4, 2, false>, (const void*)roi_gabor_tiled_kernel<4, 1, 2, false>,
                               (const void*)roi_gabor_tiled_kernel<8, 4, 2, true>, (const void*)roi_gabor_tiled_kernel<4, 1, 2, true>,
                               (const void*)roi_gabor_tiled_kernel<8, 4, 3, false>, (const void*)roi_gabor_tiled_kernel<4, 1, 3, false>,
                               (const void*)roi_gabor_tiled_kernel<8, 4, 3, true>, (const void*)roi_gabor_tiled_kernel<4, 1, 3, true>,
                               (const void*)roi_gabor_tiled_kernel<8, 4, 4, false>, (const void*)roi_gabor_tiled_kernel<8, 4, 4, true>,
                               (const void*)roi_gabor_tiled_kernel<4, 1, 4, false>, (const void*)roi_gabor_tiled_kernel<4, 1, 4, true>};
        for (const void* fn : tiled)
            if (e == hipSuccess)
                e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)roi_features_max_lds());
        return (int)e;
    }))
        return orc;
    if (grid == 0)
        return 0;
    const bool small = a.small_rois != 0;
    hipStream_t st = (hipStream_t)stream;
    if (a.sp.scratch) {   // spill launch: Gabor only (Zernike has no LDS-resident state and ran with the first launch)
        if (a.mask & NYXHIP_FAM_GABOR)
            hipLaunchKernelGGL((roi_gabor_kernel<4, true>), dim3(grid), dim3(256), 0, st, a);
        return (int)hipGetLastError();
    }
    if ((a.mask & NYXHIP_FAM_GABOR) && a.L.tiled) {
        // 2 (default): fused taps with the reference's decisions; NYXHIP_GABOR_EXACT=1 -> 0: the reference's arithmetic throughout (A/B);
        // NYXHIP_GABOR_FUSED=1 -> 1: fused taps, decisions unchecked (changes results on tie-laden inputs: INTEGRATION.md)
        // 3 (default): packed-fp32 screening with the reference's decisions; NYXHIP_GABOR_MODE=2: the fp64 fused taps of round 3 (A/B)
        static const int mode = [] {
            const char* e = getenv("NYXHIP_GABOR_FUSED");
            if (e && *e && *e != '0') return 1;
            e = getenv("NYXHIP_GABOR_EXACT");
            if (e && *e && *e != '0') return 0;
            e = getenv("NYXHIP_GABOR_MODE");
            return (e && *e == '2') ? 2 : (e && *e == '3') ? 3 : 4;
        }();
        // worth its registers (the build with the row tests needs ~25 more: one wave per SIMD less) when at least 4 % of the
        // bank's arithmetic falls away: the reference's default bank (f0 = 0 in its first filter) saves 10.6 %, the 8-orientation
        // bank of BASELINE.json configs[4] one row in 288
        int zero_halves = 0;
        for (int f = 0; f <= a.gabor_nf && f <= NYXHIP_MAX_GABOR_FILTERS; f++) {
            const uint32_t z = a.gabor_zero_rows[f], im = z >> 16, both = z & im & 0xFFFFu;
            zero_halves += __builtin_popcount(im) + __builtin_popcount(both);
        }
        const bool zr = 25 * zero_halves >= 32 * (a.gabor_nf + 1);
        if (getenv("NYXHIP_DEBUG")) fprintf(stderr, "[nyxhip] gabor launch: nf %d box_mask %x zero_halves %d zr %d mode %d total %u small %d\n", a.gabor_nf, a.gabor_box_mask, zero_halves, (int)zr, mode, a.L.total, (int)small);
#define NYX_GABOR_LAUNCH(M, Z)                                                                                                         \
        do {                                                                                                                           \
            if (small) hipLaunchKernelGGL((roi_gabor_tiled_kernel<4, 1, M, Z>), dim3(grid), dim3(64), a.L.total, st, a);               \
            else hipLaunchKernelGGL((roi_gabor_tiled_kernel<8, 4, M, Z>), dim3(grid), dim3(256), a.L.total, st, a);                    \
        } while (0)
        if (mode == 4 && a.gabor_bank32 && a.gabor_bank16) { if (zr) NYX_GABOR_LAUNCH(4, true); else NYX_GABOR_LAUNCH(4, false); }
        else if (mode >= 3 && a.gabor_bank32) { if (zr) NYX_GABOR_LAUNCH(3, true); else NYX_GABOR_LAUNCH(3, false); }
        else if (mode >= 2) { if (zr) NYX_GABOR_LAUNCH(2, true); else NYX_GABOR_LAUNCH(2, false); }
        else if (mode == 1) { if (zr) NYX_GABOR_LAUNCH(1, true); else NYX_GABOR_LAUNCH(1, false); }
        else { if (zr) NYX_GABOR_LAUNCH(0, true); else NYX_GABOR_LAUNCH(0, false); }
#undef NYX_GABOR_LAUNCH
    } else if (a.mask & NYXHIP_FAM_GABOR) {
        if (small) hipLaunchKernelGGL((roi_gabor_kernel<1, false>), dim3(grid), dim3(64), a.L.total, st, a);
        else hipLaunchKernelGGL((roi_gabor_kernel<4, false>), dim3(grid), dim3(256), a.L.total, st, a);
    }
    if (a.mask & NYXHIP_FAM_ZERNIKE) {
        if (small) hipLaunchKernelGGL(roi_zernike_kernel<1>, dim3(grid), dim3(64), 8u * a.L.zern_px_cap, st, a);
        else hipLaunchKernelGGL(roi_zernike_kernel<4>, dim3(grid), dim3(256), 8u * a.L.zern_px_cap, st, a);
    }
    return (int)hipGetLastError();
}

} // namespace nyxhip
