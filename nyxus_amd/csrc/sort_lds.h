// sort_lds.h -- the in-LDS radix sort of roi_features.hip (ROIs whose intensity range rules the counting table out).
#pragma once
#include <hip/hip_runtime.h>
#include "device_math.h"

namespace nyxhip {

template <bool GS, int NW>
__device__ __forceinline__ void grp_sync()      // the waves of one ROI meet: a workgroup barrier, or -- one wave per ROI -- nothing but ordering
{
    if (NW == 1) wav_sync<GS>(); else blk_sync<GS>();   // (NW == 1: see roi_features_body)
}


// In-place ascending LSD radix sort of the n keys of `a` (8-bit digits of key - kmin, as many passes as the range has bytes),
// ping-ponging with `b`; returns the buffer that holds the result.  hist: [NW][256] words.
// For ROIs whose intensity range rules the counting table out -- 16-bit microscopy data -- the bitonic sort above costs 78
// barrier-separated stages over the padded array (134 ns per 2821-pixel ROI against 12.7 ns through the counting table).  A pass
// here: every wave counts the digits of ITS contiguous chunk, one thread per digit turns the [digit][wave] counts into
// offsets, and every wave scatters its chunk in order -- a key's rank among the equal digits of its 64-key step comes from
// eight ballots (lanes with my digit = AND over the digit's bits of ballot-or-its-complement), so the pass is stable and needs
// no atomics in the scatter.
template <bool GS, int NW, typename KEY>
__device__ __forceinline__ KEY* radix_sort(KEY* a, KEY* b, uint32_t* hist, uint32_t n, uint32_t kmin, uint32_t range, int tid)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t chunk = ((n + NW * 64 - 1) / (NW * 64)) * 64;           // keys per wave (a multiple of 64)
    const uint32_t c_begin = (uint32_t)wave * chunk, c_end = c_begin + chunk < n ? c_begin + chunk : n;
    uint32_t* const myh = hist + wave * 256;
    for (uint32_t shift = 0; shift < 32 && (range >> shift) != 0; shift += 8) {
        for (int i = tid; i < NW * 256; i += NW * 64) hist[i] = 0;
        grp_sync<GS, NW>();
        for (uint32_t i = c_begin + (uint32_t)lane; i < c_end; i += 64)
            atomicAdd(&myh[(((uint32_t)a[i] - kmin) >> shift) & 255u], 1u);
        grp_sync<GS, NW>();
        {   // thread d < 256 owns digit d: offsets of (d, wave) = keys with a smaller digit + keys of digit d in earlier waves
            uint32_t cnt[NW], tot = 0;
            if (tid < 256) {
#pragma unroll
                for (int w = 0; w < NW; w++) { cnt[w] = hist[w * 256 + tid]; tot += cnt[w]; }
            }
            const uint32_t inc = wave_scan_u32(tid < 256 ? tot : 0u);
            // cross-wave carry through the table's own spare row is not available: the four wave totals go through four words behind it
            uint32_t* const wtot = hist + NW * 256;
            if (lane == 63 && tid < 256) wtot[wave] = inc;
            grp_sync<GS, NW>();
            if (tid < 256) {
                uint32_t base = inc - tot;
                for (int w = 0; w < wave; w++) base += wtot[w];
#pragma unroll
                for (int w = 0; w < NW; w++) { hist[w * 256 + tid] = base; base += cnt[w]; }
            }
        }
        grp_sync<GS, NW>();
        for (uint32_t i0 = c_begin; i0 < c_end; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool live = i < c_end;
            const uint32_t key = live ? (uint32_t)a[i] : 0u;
            const uint32_t d = ((key - kmin) >> shift) & 255u;
            unsigned long long same = __ballot(live);
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const bool one = (d >> bit) & 1u;
                const unsigned long long bal = __ballot(live && one);
                same &= one ? bal : ~bal;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
            uint32_t pos = 0;
            if (live) pos = myh[d] + rank;
            wav_sync<GS>();                                               // every lane has read its digit's offset before the leaders advance it
            if (live) {
                b[pos] = (KEY)key;
                if (rank == 0) myh[d] = pos + (uint32_t)__popcll(same);   // the step's first key of digit d moves the offset past the step's keys
            }
            wav_sync<GS>();
        }
        grp_sync<GS, NW>();
        KEY* t = a; a = b; b = t;
    }
    return a;
}



} // namespace nyxhip
