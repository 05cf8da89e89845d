// roi_texture.hip -- GLRLM + GLSZM + NGTDM for gfx950: the second per-ROI kernel.
//
// One 256-thread workgroup per ROI; the binned bounding-box plane lives in LDS and all
// three matrices are built there with integer LDS atomics (exact, order-independent):
//
//   GLRLM  (/root/reference/src/nyx/features/glrlm.cpp:20-276, 357-885): one wave per
//          angle; a pixel starts a run when its predecessor along the angle differs, the
//          run is walked forward and counted into an LDS Ng x Nr matrix.
//   GLSZM  (features/glszm.cpp:56-395): the reference's zones are directed-reachability
//          sets (depth-first search through E/SE/S/SW neighbours from the raster-first
//          unvisited pixel).  zone(p) = min raster index s with p reachable from s, which
//          obeys  owner(p) = min(p, owner(W), owner(NW), owner(N), owner(NE))  over
//          equal-valued predecessors -- one raster sweep: rows in sequence, a segmented
//          prefix-min along each row.  Zone sizes by LDS atomics keyed by owner; the
//          (level, size) multiplicities P(i,j) through an LDS hash table (the reference's
//          dense Ng x (w*h) matrix is mostly zeros).
//   NGTDM  (features/ngtdm.cpp:33-345): 3x3 stencil; the per-level sums of |i - mean|
//          are accumulated exactly in units of 1/840 (lcm of the neighbour counts 1..8)
//          with 64-bit integer LDS atomics, hence deterministic.
//
// HBM traffic: the ROI cloud in (8 B/px), 101 doubles out; everything else is LDS.
// Built with -ffp-contract=off (device_math.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "device_math.h"
#include "roi_kernel.h"
#include "launch_util.h"
#include "texture_feats.h"
#include "../../include/nyxhip.h"

namespace nyxhip {

// diagnostic builds (-DNYX_TEX_EXIT_AT=k, tools/tex_phases.sh): the kernel ends at phase k (results are wrong by design)
#ifdef NYX_TEX_EXIT_AT
#define TSTAMP(k) do { if ((k) == NYX_TEX_EXIT_AT) return; } while (0)
#else
#define TSTAMP(k) do { } while (0)
#endif

// GS: scratch in the global workspace (large-ROI launches).  OCC: workgroups per CU the register budget is cut for -- the
// serial stretches of this kernel (the GLSZM row sweep, the hash probes) are latency-bound, so when six carve-outs fit a
// CU the 80-register build (a few spills) beats the 106-register one by 20-25 %.
// D8: the binned plane holds 8-bit levels (grey depth <= 254, LDS launches): with it the benchmark's carve-out fits eight times
// into a CU (the 64-register build).
// Row scans / sweeps in registers take boxes up to 64 * chunks wide: four chunks in the LDS builds (a wider box rarely fits LDS), eight in
// the global-workspace build -- the 300..400-px boxes of a heavy-tailed batch fell to the per-pixel walks there, dependent round trips
// to global memory: 17.5 ms for one 361 x 341 ROI (GLRLM 6.9, GLSZM sweep 4.5, NGTDM stencil 2.9 ms).
template <bool GS> struct TexChunks { static constexpr int value = GS ? 8 : 4; };

template <bool GS, int OCC, bool D8 = false>
__global__ __launch_bounds__(kBlock, OCC) void roi_texture_kernel(const TexArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int kRlmChunks = TexChunks<GS>::value;      // GLRLM row scans in registers: boxes up to 64 * kRlmChunks wide
    constexpr int kSzmChunks = TexChunks<GS>::value;      // GLSZM row sweep (and the NGTDM stencil): boxes up to 64 * kSzmChunks wide (TexLayout: owner labels only beyond that)
    unsigned char* const lds = GS ? A.sp.scratch + (size_t)blockIdx.x * A.sp.stride : lds_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: lets the line / row bookkeeping run on the scalar unit
    uint64_t roi;
    if (!roi_of_slot(A.sp, blockIdx.x, A.n_roi, roi))
        return;
    double* s_out = (double*)(lds + A.L.out);
    double* s_red = (double*)(lds + A.L.red);
    double* s_stat = (double*)(lds + A.L.stat);
    using dense_t = typename std::conditional<D8, uint8_t, uint16_t>::type;
    dense_t* s_dense = (dense_t*)(lds + A.L.dense);
    uint16_t* s_lvlmap = (uint16_t*)(lds + A.L.lvlmap);   // level -> row index + 1
    uint32_t* s_lv = (uint32_t*)(lds + A.L.lv);           // row index -> level value
    unsigned char* s_work = lds + A.L.work;                // per-family scratch (aliased)
    const double* const s_lvf = A.L.lvf ? (const double*)(lds + A.L.lvf) : nullptr;   // row index -> (level^2, 1 / level^2)

    const uint64_t off = A.px_offset[roi];
    const uint32_t n = (uint32_t)(A.px_offset[roi + 1] - off);
    const uint32_t w = A.bbox_w[roi], h = A.bbox_h[roi];
    if (GS && A.sp.skip_ltex && ltex_eligible(n, w, h))
        return;                                       // served by the several-workgroups-per-ROI path (roi_large_tex.hip)
    const uint32_t area = w * h;
    const uint32_t vmin = A.min_inten[roi], vmax = A.max_inten[roi];
    double* const out_row = A.out + roi * A.ld + A.col0;
    // local column -> column inside the row: other kernels' families (GLDZM; GLDM, NGLDM) interleave in Feature2D order
    const int end_rlm = (A.mask & NYXHIP_FAM_GLRLM) ? 80 : 0, end_szm = end_rlm + ((A.mask & NYXHIP_FAM_GLSZM) ? 16 : 0);
    auto gcol = [=](int c) { return c + (c >= end_rlm ? A.gap_after_glrlm : 0) + (c >= end_szm ? A.gap_after_glszm : 0); };
    const int solo = (int)(roi & 3u);                  // the wave that runs this ROI's single-wave stretches
    const bool do_rlm = (A.mask & NYXHIP_FAM_GLRLM) != 0, do_szm = (A.mask & NYXHIP_FAM_GLSZM) != 0,
               do_ngt = (A.mask & NYXHIP_FAM_NGTDM) != 0;
    const uint32_t side = w > h ? w : h;
    if (n == 0 || area > A.L.dense_cap || side > A.L.side_cap) {
        if (n != 0 && A.sp.defer_large)
            return;                                   // handled by the spill launch that follows
        if (tid == 0 && n != 0)
            atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        for (int c = tid; c < A.n_cols; c += kBlock)
            out_row[gcol(c)] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    const double mslope = greyInfo > 0 ? (double)greyInfo / ((double)vmax - 0.) : 0.0;
    const uint32_t Lcap = A.L.lvl_cap;

    // ---- phase 0: plane = background level (matlab binning sends 0 -> 1, texture_feature.h:150-154)
    for (int c = tid; c < A.n_cols; c += kBlock)
        s_out[c] = 0.0;
    {
        const uint32_t bg = greyInfo > 0 ? (D8 ? 0x01010101u : 0x00010001u) : 0u;
        uint32_t* d32 = (uint32_t*)s_dense;
        for (uint32_t i = tid; i < (D8 ? (area + 3) / 4 : (area + 1) / 2); i += kBlock)
            d32[i] = bg;
        for (uint32_t i = tid; i <= Lcap + 1; i += kBlock)
            s_lvlmap[i] = 0;
        if (tid >= 1 && tid <= 8)                       // 840 / nd for nd = 1..8 (NGTDM stencil): 840 420 280 210 168 140 120 105
            ((uint32_t*)(s_stat + 8))[tid] = (uint32_t)(840.0f / (float)tid + 0.5f);
    }
    blk_sync<GS>();

    // ---- phase 1: cloud -> binned plane -----------------------------------------------------------
    // The levels present and the number of non-zero binned pixels are taken on the way (a pixel list holds a position once): the
    // plane is not read again for them.  Under matlab binning the background is level 1 -- present when the box has a pixel
    // outside the list -- and every pixel of the box is non-zero.
    uint32_t nz_orig = 0, lvl_over = 0, nz_bin = 0;
    if (tid == 0 && greyInfo > 0 && n < area) s_lvlmap[1] = 1;
    for_each_cloud_pixel<kBlock>(A.inten + off, A.x + off, A.y + off, n, tid, [&](uint32_t, uint32_t v, uint32_t px, uint32_t py) {
        uint32_t lvl;
        if (greyInfo > 0) {     // bin_matlab: a non-zero value gives floor(slope v + 1) >= 1, and the conversion truncates (floor of a positive value)
            const uint32_t sc = (uint32_t)(mslope * (double)v + 1.0);         // (0 -> 1 as well)
            lvl = sc > (uint32_t)greyInfo ? (uint32_t)greyInfo : sc;
        } else
            lvl = greyInfo < 0 ? bin_radiomix(v, vmin, vmax, -greyInfo) : v;
        nz_orig += v != 0;
        if (greyInfo > 0) {                             // (matlab binning: 1 <= level <= grey depth <= Lcap, and every pixel of the box counts as non-zero)
            if (px < w && py < h) {
                s_dense[__umul24(py, w) + px] = (dense_t)lvl;
                s_lvlmap[lvl] = 1;
            }
        } else {
            if (lvl > Lcap) { lvl_over = 1; lvl = Lcap; }
            if (px < w && py < h) {
                s_dense[__umul24(py, w) + px] = (dense_t)lvl;
                if (lvl != 0) { s_lvlmap[lvl] = 1; nz_bin++; }
            }
        }
    });
    nz_orig = wave_sum_t<uint32_t>(nz_orig);
    nz_bin = wave_sum_t<uint32_t>(nz_bin);
    lvl_over = wave_max_u32(lvl_over);
    if (lane == 0) { s_red[wave * 8] = (double)nz_orig; s_red[wave * 8 + 1] = (double)lvl_over; s_red[wave * 8 + 2] = (double)nz_bin; }
    blk_sync<GS>();
    TSTAMP(0);
    // The two pixel counts are needed once each, much later: they wait in s_stat instead of occupying registers through
    // the whole kernel.
    bool over = false;
    {
        uint32_t Np_orig = 0;   // non-zero ORIGINAL pixels (glrlm.cpp:197-204)
        uint32_t Np_bin = 0;    // non-zero BINNED pixels (glszm.cpp:193-199)
        for (int wv = 0; wv < kWaves; wv++) { Np_orig += (uint32_t)s_red[wv * 8]; over |= s_red[wv * 8 + 1] != 0; Np_bin += (uint32_t)s_red[wv * 8 + 2]; }
        if (tid == 0) { s_stat[2] = (double)Np_orig; s_stat[3] = (double)(greyInfo > 0 ? area : Np_bin); }
    }
    if (over) { // IBSI level beyond the LDS-resident capacity
        if (tid == 0) atomicCAS(A.status, 0, NYXHIP_ERR_UNSUPPORTED);
        for (int c = tid; c < A.n_cols; c += kBlock)
            out_row[gcol(c)] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    // sorted unique non-zero levels (glrlm.cpp:101-105, glszm.cpp:97-101, ngtdm.cpp:53-67);
    // IBSI: I = 1..max (GLRLM/GLSZM) and 0..max (NGTDM)
    if (tid == 0) {
        int k = 0;
        uint32_t mx = 0;
        for (uint32_t l = 1; l <= Lcap; l++)
            if (s_lvlmap[l]) {
                mx = l;
                if (greyInfo != 0) { s_lvlmap[l] = (uint16_t)(k + 1); s_lv[k] = l; }
                k++;
            }
        if (greyInfo == 0)
            for (uint32_t l = 1; l <= mx; l++) { s_lvlmap[l] = (uint16_t)l; s_lv[l - 1] = l; }
        s_stat[0] = (double)(greyInfo == 0 ? (int)mx : k);   // Ng for GLRLM / GLSZM
        s_stat[1] = (double)k;                               // Ngp = unique non-zero levels (ngtdm.cpp:150)
    }
    blk_sync<GS>();
    const int Ng = __builtin_amdgcn_readfirstlane((int)s_stat[0]);    // (wave-uniform numbers go to scalar registers at once: kept as the doubles
                                                                      //  they were read as, two of them sat in scratch across the GLRLM section)
    if (A.L.lvf) {
        for (int i = tid; i < Ng; i += kBlock) {
            const double in2d = (double)(uint32_t)(s_lv[i] * s_lv[i]);       // (levels <= 4094: the unsigned product of glrlm.cpp and the double product of glszm.cpp are the same number)
            double* t = (double*)(lds + A.L.lvf) + 2 * i;
            t[0] = in2d; t[1] = frcp(in2d);
        }
        blk_sync<GS>();
    }
    TSTAMP(1);
    const int Nuniq = __builtin_amdgcn_readfirstlane((int)s_stat[1]);
    const bool blank = vmin == vmax;
    int col = 0;

    // =====================================================================================
    // GLRLM
    // =====================================================================================
    if (do_rlm) {
        double* o = s_out + col;
        col += 80;
        double* s_f = (double*)s_work;                       // [4][16]
        uint32_t* s_mat = (uint32_t*)(s_work + 4 * 16 * 8);  // slots of Ng*Nr + Ng + Nr words
        const int Nr = (int)side;
        const uint32_t slot_words = (uint32_t)(Ng * Nr + Ng + Nr + 4);
        const int nslot = (int)(A.L.work_bytes > 512 ? (A.L.work_bytes - 512) / (4ull * slot_words) : 0);
        if (blank) {                                         // glrlm.cpp:29-52
            for (int c = tid; c < 80; c += kBlock) o[c] = A.soft_nan;
        } else if (nslot < 1 || Ng < 1) {
            if (tid == 0 && Ng >= 1) atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
            for (int c = tid; c < 80; c += kBlock) o[c] = Ng < 1 ? 0.0 : __longlong_as_double(0x7ff8000000000000LL);
        } else {
            if (nslot >= 4 && side <= 64) {
                // One wave per direction, one lane per column, one step per row (h steps instead of one step per line of
                // the direction -- the two diagonals alone have 2 (w + h - 1) lines).
                //   E  (wave 0): the ballot of "equals the next pixel" turns every run of a row into a string of set bits; a
                //                run's first lane reads its length off the mask.
                //   SE, S, SW (waves 1-3): every lane carries the run (value, length) that ends on its line in the previous
                //                row; for the diagonals the pair moves one lane right / left per row through a DPP wave shift.
                //                A run is counted when its line ends or the value changes.  A pair that is shifted past the
                //                last column lands on a lane that always reads level 0 and is counted there; the pairs that
                //                would leave the wave (lane 0 for SW, lane 63 of a 64-wide box for SE) are counted before
                //                the shift.
                // Each wave fills and then reads its own matrix: no workgroup barrier between the scan and the features.
                blk_sync<GS>();
                for (uint32_t i = tid; i < 4u * slot_words; i += kBlock) s_mat[i] = 0;
                blk_sync<GS>();
                {
                    // A run travels as ONE number, its cell S = m * Nr + rl of the matrix (m = row + 1 of its level, rl = length so far;
                    // Pm = P - (Nr + 1) makes Pm[S] that cell): with c = m' * Nr + 1 the cell a run of the pixel's level m' starts in,
                    // "same level" is S - c < Nr (unsigned), "continue" is S + 1, "a run is open" is S > Nr.  Level 0 (outside the ROI /
                    // the box) counts as row 0: its S stays <= Nr (a line has at most Nr pixels) and is never written.  Lanes beyond
                    // the box read a zero of the level map through a pointer with stride 0 -- no per-row bounds arithmetic.
                    // ~11 vector instructions per row and direction (28 before: value, length and row travelled separately).
                    uint32_t* const P = s_mat + (uint32_t)wave * slot_words;
                    uint32_t* const Pm = P - (Nr + 1);
                    const uint32_t nr = (uint32_t)Nr;
                    // The E direction runs the same machine TRANSPOSED: its lanes are rows and its steps are columns, so the state never
                    // moves between lanes (the ballot formulation -- run starts and ends read off a mask -- cost 16 instructions per row
                    // against 10; byte reads a row pitch apart share banks only for pitches of 32 and 64, 8 and 16 lanes per bank).
                    const bool trans = wave == 0;
                    const bool in_col = (uint32_t)lane < (trans ? h : w);
                    const uint32_t n_steps = trans ? w : h;
                    // Two loads ahead: the level of row + 2 and the level-map entry of row + 1.  (The reads of the last two trips land
                    // up to two rows behind the plane -- in the regions that follow it, never used.  What a trip carries over is made
                    // of 32-bit results, the map address and the cell c, not of the loaded bytes: no re-extension per trip.)
                    const dense_t* ptr = in_col ? s_dense + (trans ? mul24((uint32_t)lane, w) : (uint32_t)lane) : (const dense_t*)s_lvlmap;   // (s_lvlmap[0] == 0)
                    const uint32_t col_stride = in_col ? (trans ? 1u : w) : 0u;
                    // (LDS launches form the map address as a 32-bit LDS address right behind the byte load and pin it there: sunk
                    //  into the block of its use, the byte costs a re-extension per trip)
                    typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
                    using map_h = typename std::conditional<GS, const uint16_t*, uint32_t>::type;
                    const uint32_t map0 = GS ? 0u : (uint32_t)(uintptr_t)(const lds_u16_t*)s_lvlmap;
                    auto map_of = [=](uint32_t v) -> map_h {
                        if constexpr (GS) return s_lvlmap + v;
                        else { uint32_t a = map0 + 2u * v; asm volatile("" : "+v"(a)); return a; }
                    };
                    auto map_read = [=](map_h hd) -> uint32_t {
                        if constexpr (GS) return (uint32_t)*hd;
                        else return (uint32_t)*(const lds_u16_t*)hd;
                    };
                    uint32_t c = mad24((uint32_t)s_lvlmap[*ptr], nr, 1u);                                     // row 0: the cell a run of the pixel's level starts in
                    ptr += col_stride;
                    map_h map_at = map_of((uint32_t)*ptr);                                                    // row 1
                    ptr += col_stride;
                    {
                        // E (transposed), SE, S, SW: the run that ends on a lane's line in the previous step; along the diagonals it moves one lane per
                        // row (a rotation: what leaves the box lands on a lane that reads level 0 and is written there; a 64-wide box
                        // has no such lane, its leaving run is written before the move).
                        auto scan = [&](auto dxc) {
                            constexpr int dx = decltype(dxc)::value;                         // glrlm.cpp:128-176
                            uint32_t S = 0;
                            // one row: the pixel's cell `cur`, the cell of the row after it into `nxt` (rows in pairs below: the two
                            // change roles instead of a copy per row)
                            auto step = [&](const uint32_t cur, uint32_t& nxt) {
                                nxt = mad24(map_read(map_at), nr, 1u);
                                map_at = map_of((uint32_t)*ptr);
                                ptr += col_stride;
                                if (dx != 0) {
                                    if (w == 64u) {
                                        if ((uint32_t)lane == (dx == 1 ? 63u : 0u)) { if (S > nr) atomicAdd(&Pm[S], 1u); S = 0; }
                                    }
                                    S = dx == 1 ? wave_ror1(S) : wave_rol1(S);
                                }
                                const bool cont = S - cur < nr;
                                if (!cont && S > nr) atomicAdd(&Pm[S], 1u);
                                S = cont ? S + 1u : cur;
                            };
                            uint32_t c2 = 0, row = 0;
                            for (; row + 1u < n_steps; row += 2u) { step(c, c2); step(c2, c); }
                            if (row < n_steps) step(c, c2);
                            if (S > nr) atomicAdd(&Pm[S], 1u);
                        };
                        if (wave == 1) scan(std::integral_constant<int, 1>());
                        else if (wave == 3) scan(std::integral_constant<int, -1>());
                        else scan(std::integral_constant<int, 0>());                 // wave 2: S; wave 0: E, transposed
                    }
                    wav_sync<GS>();
                    TSTAMP(2);
                    glrlm_features_wave<GS>(P, Ng, Nr, s_lv, s_lvf, P + Ng * Nr, P + Ng * Nr + Ng, (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s_stat[2]), s_f + wave * 16, s_red + wave * 8, lane);
                }
            } else if (nslot >= 4 && w > 64 && w <= 64u * kRlmChunks) {
                // Boxes 65 .. 256 wide: the same one-wave-per-direction row scans over up to four chunks of 64 columns (lane =
                // column + 64 c).  E: a run that reaches a chunk's column 63 continues with the following chunks' strings of set
                // bits (their lengths are wave-uniform).  SE / SW: the run state that leaves one chunk enters the next through
                // v_readlane, the state that leaves the box is counted before the shift.  (The per-pixel walk below made a 65-wide
                // box 2.8 times as expensive as a 63-wide one.)
                blk_sync<GS>();
                for (uint32_t i = tid; i < 4u * slot_words; i += kBlock) s_mat[i] = 0;
                blk_sync<GS>();
                {
                    uint32_t* const P = s_mat + (uint32_t)wave * slot_words;
                    const int nch = (int)((w + 63u) >> 6);           // 2 .. kRlmChunks, wave-uniform
                    bool inc[kRlmChunks];
#pragma unroll
                    for (int c = 0; c < kRlmChunks; c++) inc[c] = (uint32_t)lane + 64u * c < w;
                    uint32_t* const Pm = P - (Nr + 1);
                    // Workspace build: the matrix lives in global memory, where 64 lanes adding into the handful of (level, short run)
                    // cells are 64 serialised L2 atomics (4.1 ms of a 361 x 341 ROI's GLRLM).  Runs of up to kRlmLdsCols pixels -- nearly
                    // all of them on a textured image -- are counted in an LDS copy of those columns and added to the matrix before
                    // the feature pass.
                    const bool lds_cols = GS && A.L.gs_rlm_ok != 0;
                    uint32_t* const Lp = (uint32_t*)(lds_raw + A.L.gs_rlm) + (uint32_t)wave * (uint32_t)Ng * kRlmLdsCols;   // cell (m, rl) = Lp[(m - 1) * kRlmLdsCols + rl - 1]
                    if (lds_cols) {
                        for (uint32_t i = lane; i < (uint32_t)Ng * kRlmLdsCols; i += 64) Lp[i] = 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    }
                    auto count_at = [=](uint32_t rm, uint32_t rl) {
                        if (lds_cols && rl <= (uint32_t)kRlmLdsCols) atomicAdd(&Lp[mad24(rm, (uint32_t)kRlmLdsCols, rl) - (uint32_t)(kRlmLdsCols + 1)], 1u);
                        else atomicAdd(&Pm[mad24(rm, (uint32_t)Nr, rl)], 1u);
                    };
                    auto count_run = [=](uint32_t rv, uint32_t rl) { count_at((uint32_t)s_lvlmap[rv], rl); };
                    auto load = [=](uint32_t row, int c, bool in) -> uint32_t {
                        return (in && row < h) ? (uint32_t)s_dense[row * w + 64u * (uint32_t)c + (uint32_t)lane] : 0u;
                    };
                    uint32_t vn[kRlmChunks];
#pragma unroll
                    for (int c = 0; c < kRlmChunks; c++) vn[c] = c < nch ? load(0, c, inc[c]) : 0u;
                    if (wave == 0) {
                        for (uint32_t row = 0; row < h; row++) {
                            uint32_t v[kRlmChunks];
                            unsigned long long same[kRlmChunks];
#pragma unroll
                            for (int c = 0; c < kRlmChunks; c++) { v[c] = vn[c]; vn[c] = c < nch ? load(row + 1, c, inc[c]) : 0u; }
#pragma unroll
                            for (int c = 0; c < kRlmChunks; c++) {
                                const uint32_t fill = (c + 1 < kRlmChunks && c + 1 < nch) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v[c + 1 < kRlmChunks ? c + 1 : c]) : 0u;
                                const uint32_t nx = lane_plus1(v[c], fill);
                                same[c] = c < nch ? __ballot((uint32_t)lane + 64u * c + 1u < w && v[c] != 0 && v[c] == nx) : 0ull;   // bit 63: continues in the next chunk
                            }
                            // pixels a run gains beyond chunk c once it passes that chunk's column 63 (wave-uniform, last chunk first)
                            uint32_t ext[kRlmChunks];
#pragma unroll
                            for (int c = kRlmChunks - 1; c >= 0; c--) {
                                if (c + 1 < kRlmChunks) {
                                    const unsigned long long nxt = same[c + 1 < kRlmChunks ? c + 1 : c];
                                    const uint32_t head = ~nxt ? (uint32_t)__ffsll((long long)~nxt) - 1u : 64u;
                                    ext[c] = head + (head == 64u ? ext[c + 1 < kRlmChunks ? c + 1 : c] : 0u);
                                } else
                                    ext[c] = 0u;
                            }
#pragma unroll
                            for (int c = 0; c < kRlmChunks; c++) {
                                if (c < nch) {
                                    const bool cont = lane > 0 ? ((same[c] >> (lane - 1)) & 1ull) != 0 : (c > 0 ? (same[c > 0 ? c - 1 : 0] >> 63) != 0 : false);
                                    if (inc[c] && v[c] != 0 && !cont) {
                                        const unsigned long long rest = ~(same[c] >> lane);                 // (the shift zero-fills from the top)
                                        const uint32_t t = rest ? (uint32_t)__ffsll((long long)rest) - 1u : 64u;   // set bits from this lane on
                                        const bool through = t == 64u - (uint32_t)lane;                       // all of bits lane .. 63 set
                                        count_run(v[c], through ? t + 1u + ext[c] : t + 1u);
                                    }
                                }
                            }
                        }
                    } else {
                        const int dx = wave == 1 ? 1 : wave == 2 ? 0 : -1;                   // glrlm.cpp:128-176
                        uint32_t rv[kRlmChunks], rl[kRlmChunks], rm[kRlmChunks];              // runs of the chunks: level, length, matrix row + 1
#pragma unroll
                        for (int c = 0; c < kRlmChunks; c++) { rv[c] = 0; rl[c] = 0; rm[c] = 0; }
                        const bool full_last = w == 64u * (uint32_t)nch;                      // the last chunk's lane 63 is a column of the box
                        for (uint32_t row = 0; row < h; row++) {
                            if (dx == 1) {
#pragma unroll
                                for (int c = kRlmChunks - 1; c >= 0; c--) {
                                    if (c < nch) {
                                        if (c == nch - 1 && full_last && lane == 63 && rv[c] != 0) count_at(rm[c], rl[c]);
                                        const uint32_t a = c > 0 ? readlane63(rv[c > 0 ? c - 1 : 0]) : 0u, bq = c > 0 ? readlane63(rl[c > 0 ? c - 1 : 0]) : 0u,
                                                       cq = c > 0 ? readlane63(rm[c > 0 ? c - 1 : 0]) : 0u;
                                        rv[c] = lane_minus1(rv[c], a); rl[c] = lane_minus1(rl[c], bq); rm[c] = lane_minus1(rm[c], cq);
                                    }
                                }
                            } else if (dx == -1) {
                                if (lane == 0 && rv[0] != 0) count_at(rm[0], rl[0]);
#pragma unroll
                                for (int c = 0; c < kRlmChunks; c++) {
                                    if (c < nch) {
                                        const bool more = c + 1 < kRlmChunks && c + 1 < nch;
                                        const uint32_t a = more ? (uint32_t)__builtin_amdgcn_readfirstlane((int)rv[c + 1 < kRlmChunks ? c + 1 : c]) : 0u,
                                                       bq = more ? (uint32_t)__builtin_amdgcn_readfirstlane((int)rl[c + 1 < kRlmChunks ? c + 1 : c]) : 0u,
                                                       cq = more ? (uint32_t)__builtin_amdgcn_readfirstlane((int)rm[c + 1 < kRlmChunks ? c + 1 : c]) : 0u;
                                        rv[c] = lane_plus1(rv[c], a); rl[c] = lane_plus1(rl[c], bq); rm[c] = lane_plus1(rm[c], cq);
                                    }
                                }
                            }
#pragma unroll
                            for (int c = 0; c < kRlmChunks; c++) {
                                if (c < nch) {
                                    const uint32_t v = vn[c];
                                    vn[c] = load(row + 1, c, inc[c]);
                                    if (v != 0 && v == rv[c]) rl[c]++;
                                    else {
                                        if (rv[c] != 0) count_at(rm[c], rl[c]);
                                        rv[c] = v; rl[c] = v != 0 ? 1u : 0u;
                                        rm[c] = v != 0 ? (uint32_t)s_lvlmap[v] : 0u;
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int c = 0; c < kRlmChunks; c++)
                            if (c < nch && rv[c] != 0) count_at(rm[c], rl[c]);
                    }
                    wav_sync<GS>();
                    if (lds_cols) {                              // the short runs join the matrix (this wave's own: plain adds)
                        for (uint32_t i = lane; i < (uint32_t)Ng * kRlmLdsCols; i += 64) {
                            const uint32_t m0 = i / kRlmLdsCols, j = i - m0 * kRlmLdsCols, cnt = Lp[i];
                            if (cnt != 0 && j < (uint32_t)Nr) P[m0 * (uint32_t)Nr + j] += cnt;
                        }
                        wav_sync<GS>();
                    }
                    glrlm_features_wave<GS>(P, Ng, Nr, s_lv, s_lvf, P + Ng * Nr, P + Ng * Nr + Ng, (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s_stat[2]), s_f + wave * 16, s_red + wave * 8, lane);
                }
            } else {
            const int per = nslot >= 4 ? 4 : nslot;          // angles handled concurrently (one wave each)
                for (int a0 = 0; a0 < 4; a0 += per) {
                    blk_sync<GS>();
                    for (uint32_t i = tid; i < (uint32_t)per * slot_words; i += kBlock) s_mat[i] = 0;
                    blk_sync<GS>();
                    if (wave < per && a0 + wave < 4) {
                        const int ai = a0 + wave;
                        uint32_t* P = s_mat + wave * slot_words;
                        const int dx = ai == 2 ? 0 : ai == 3 ? -1 : 1, dy = ai == 0 ? 0 : 1; // glrlm.cpp:128-176
                        for (uint32_t p = lane; p < area; p += 64) {
                            uint32_t v = s_dense[p];
                            if (v == 0) continue;
                            int row = (int)(p / w), cl = (int)(p - (uint32_t)row * w);
                            int pr = row - dy, pc = cl - dx;
                            if (pr >= 0 && pc >= 0 && pc < (int)w && s_dense[(uint32_t)pr * w + pc] == v)
                                continue;                        // not the first pixel of its run
                            int len = 1, r2 = row + dy, c2 = cl + dx;
                            while (r2 < (int)h && c2 >= 0 && c2 < (int)w && s_dense[(uint32_t)r2 * w + c2] == v) {
                                len++; r2 += dy; c2 += dx;
                            }
                            atomicAdd(&P[((int)s_lvlmap[v] - 1) * Nr + (len - 1)], 1u);
                        }
                        wav_sync<GS>();
                        glrlm_features_wave<GS>(P, Ng, Nr, s_lv, s_lvf, P + Ng * Nr, P + Ng * Nr + Ng, (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s_stat[2]), s_f + ai * 16, s_red + wave * 8, lane);
                    }
                }
            }
            blk_sync<GS>();
            for (int c = tid; c < 64; c += kBlock) {         // feature-major, angle-minor
                int k = c >> 2, a = c & 3;
                o[c] = s_f[a * 16 + k];
            }
            for (int k = tid; k < 16; k += kBlock) {         // calc_ave :903-910 (std::reduce of 4)
                double v = 0.0 + ((s_f[0 * 16 + k] + s_f[1 * 16 + k]) + (s_f[2 * 16 + k] + s_f[3 * 16 + k]));
                o[64 + k] = v / 4.0;
            }
        }
        blk_sync<GS>();
    }

    TSTAMP(3);
    // ---- NGTDM stencil for boxes up to a wave wide (used from two places: see the GLSZM sweep) ------------------------------
    // IBSI: I = 0..max, row = level (ngtdm.cpp:56-61, :163-166); else rows = unique levels
    const int NgT = greyInfo == 0 ? (Nuniq ? Ng + 1 : 0) : Nuniq;
    const bool ngt_own = !GS && A.L.ngt_own != 0;
    // (workspace build: the accumulators sit in LDS all the same when they fit -- two atomics per pixel on Ng addresses of global
    //  memory were 1.6 ms of a 361 x 341 ROI)
    unsigned long long* const s_S = (unsigned long long*)((GS && A.L.gs_ngt_ok) ? lds_raw + A.L.gs_ngt : ngt_own ? lds + A.L.ngt_own : s_work);   // [NgT] sum |i - mean| in units of 1/840
    uint32_t* const s_N = (uint32_t*)(s_S + A.L.ng_cap + 2);                                           // [NgT]
    // Replicas of the accumulators (LDS launches with few levels): a lane adds into replica lane % R, so that the 64 lanes of an atomic
    // spread over R times as many addresses -- with eight levels every atomic of the stencil had eight lanes per address, and LDS
    // serialises those (the counters showed half of the kernel's LDS cycles as conflicts).  Replica 0 collects the others later.
    const uint32_t ngt_rep = A.L.ngt_rep, ngt_words = A.L.ngt_stride / 4u;                             // (stride in bytes, a multiple of 8)
    // One lane per column, rows r_begin .. r_end - 1; the rows above / below travel in registers and the horizontal neighbours
    // come through DPP lane shifts (level 0 = outside the box or not a pixel: skipped, like the bounds tests and the q != 0 test
    // of the reference's stencil).  A level travels with a "present" flag in bit 24 (levels are 16-bit), so ONE sum over the eight
    // neighbours yields both their level sum (bits 0..23) and their number (bits 24..27); |840 i - sum * (840 / nd)| is 32-bit
    // arithmetic (24-bit multiplies, one v_sad_u32), 840 / nd comes from a nine-entry LDS table, and while a level's sum cannot
    // reach 2^32 the accumulation is a 32-bit LDS atomic on the low word.  ~27 vector instructions per row (70 before).
    // (col0: first column of the 64-column chunk the wave's lanes stand for -- boxes wider than a wave take the stencil chunk by
    //  chunk; the columns just outside a chunk are wave-uniform plane reads that fill the lane shifts' open ends)
    auto ngtdm_rows = [&](int r_begin, int r_end, uint32_t col0 = 0u) {
        const bool in_col = col0 + (uint32_t)lane < w;
        const uint32_t* const s_q = (const uint32_t*)(s_stat + 8);
        const bool sum32 = (unsigned long long)area * 840ull * (unsigned long long)(greyInfo == 0 ? Ng + 1 : (int)s_lv[Ng > 0 ? Ng - 1 : 0]) < (1ull << 32);
        auto code_at = [=](int r, uint32_t cl, bool ok) -> uint32_t {
            const uint32_t v = (ok && r >= 0 && r < (int)h) ? (uint32_t)s_dense[(uint32_t)r * w + cl] : 0u;
            return v | (min(v, 1u) << 24);
        };
        auto load_row = [=](int r) -> uint32_t { return code_at(r, col0 + (uint32_t)lane, in_col); };
        const bool has_w = col0 > 0u, has_e = col0 + 64u < w;        // a column of the box to the chunk's left / right
        const uint32_t rep_off = mul24((uint32_t)lane & (ngt_rep - 1u), ngt_words);
        uint32_t* const r_N = s_N + rep_off;
        unsigned long long* const r_S = s_S + (rep_off >> 1);
        auto west = [=](int r) -> uint32_t { return has_w ? code_at(r, col0 - 1u, true) : 0u; };
        auto east = [=](int r) -> uint32_t { return has_e ? code_at(r, col0 + 64u, true) : 0u; };
        uint32_t prv = load_row(r_begin - 1), cur = load_row(r_begin);
        uint32_t prv_w = west(r_begin - 1), cur_w = west(r_begin), prv_e = east(r_begin - 1), cur_e = east(r_begin);
        for (int row = r_begin; row < r_end; row++) {
            const uint32_t nxt = load_row(row + 1);
            const uint32_t nxt_w = west(row + 1), nxt_e = east(row + 1);
            // the eight neighbours = the three-row sums of the columns to the left and right + this column's without the centre
            // (integer sums: any order) -- two lane shifts per row instead of six
            const uint32_t col3 = prv + cur + nxt;
            uint32_t tot;
            if (!has_w && !has_e) tot = lane_minus1(col3, 0u) + lane_plus1(col3, 0u) + (prv + nxt);
            else tot = lane_minus1(col3, prv_w + cur_w + nxt_w) + lane_plus1(col3, prv_e + cur_e + nxt_e) + (prv + nxt);
            if (cur != 0 && tot >= (1u << 24)) {
                const uint32_t lvl = cur & 0xFFFFFFu, sum = tot & 0xFFFFFFu, nd = tot >> 24;
                const uint32_t r = greyInfo == 0 ? lvl : (uint32_t)s_lvlmap[lvl] - 1u;
                uint32_t d;                             // |840 i - sum * (840 / nd)|
                asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(mul24(lvl, 840u)), "v"(mul24(sum, s_q[nd])));
                atomicAdd(&r_N[r], 1u);
                if (sum32) atomicAdd((uint32_t*)&r_S[r], d);
                else atomicAdd(&r_S[r], (unsigned long long)d);
            }
            prv = cur; cur = nxt;
            prv_w = cur_w; cur_w = nxt_w; prv_e = cur_e; cur_e = nxt_e;
        }
    };
    bool ngt_stencil_done = false;
    // =====================================================================================
    // GLSZM
    // =====================================================================================
    if (do_szm) {
        double* o = s_out + col;
        col += 16;
        uint32_t* s_count = (uint32_t*)(s_work + A.L.szm_count);   // [area + 1] zone size at the owner / later zones per size
        uint32_t* s_hkey = (uint32_t*)(s_work + A.L.szm_hkey);     // [hcap] (row << 20 | size), 0xFFFFFFFF = empty
        uint32_t* s_hval = s_hkey + A.L.hash_cap;            // [hcap] multiplicity P(i,j)
        uint32_t* s_si = s_hval + A.L.hash_cap;              // [Ng] zones per level
        uint32_t* s_label = (uint32_t*)(s_work + A.L.szm_label);   // [area] owner index of each pixel (boxes wider than 64 only)
        // the ROI's own table size (<= A.L.hash_cap, the same function of the launch's largest box): slot order, hence the order of
        // the floating-point sums over the cells, is a property of the ROI -- not of the batch it travels in
        const uint32_t hcap = szm_hash_cap(A.L.ng_cap, area);
        // P(i,j) cells: sizes <= S sit in a direct [level][size] table (one atomic per zone), larger ones in the ordered hash.
        // Every pass over the cells walks hash slots first, then the table: a fixed order on every launch.
        const uint32_t S = A.L.szm_small;                     // 0 or 32
        uint32_t* s_small = (uint32_t*)(s_work + A.L.szm_smalltab);
        const uint32_t n_cells = hcap + (uint32_t)Ng * S;
        auto cell = [=](uint32_t i, uint32_t& key, uint32_t& val) {
            if (i < hcap) { key = s_hkey[i]; val = s_hval[i]; }
            else {
                const uint32_t idx = i - hcap;
                val = s_small[idx];
                key = val ? ((idx >> 5) << 20) | ((idx & 31u) + 1u) : 0u;    // S == 32 whenever this branch exists
            }
        };
        // zone-size table: 16-bit entries packed two per word while every size fits (halves never carry: a count is <= area)
        const bool c16 = A.L.szm_c16 != 0;
        auto cnt_add = [=](uint32_t idx, uint32_t inc) {
            if (c16) atomicAdd(&s_count[idx >> 1], inc << (16u * (idx & 1u)));
            else atomicAdd(&s_count[idx], inc);
        };
        auto cnt_get = [=](uint32_t idx) -> uint32_t {
            return c16 ? (uint32_t)((const uint16_t*)s_count)[idx] : s_count[idx];
        };
        const uint32_t cnt_words = c16 ? (area + 2) / 2 : area + 1;
        if (blank) {                                         // glszm.cpp:61-65
            for (int c = tid; c < 16; c += kBlock) o[c] = A.soft_nan;
        } else if (A.L.szm_ok == 0 || area > (1u << 20) - 1) {
            if (tid == 0) atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
            for (int c = tid; c < 16; c += kBlock) o[c] = __longlong_as_double(0x7ff8000000000000LL);
        } else {
            for (uint32_t i = tid; i < cnt_words; i += kBlock) s_count[i] = 0;
            for (uint32_t i = tid; i < hcap; i += kBlock) { s_hkey[i] = 0; s_hval[i] = 0; }   // key 0 = empty (a size is >= 1)
            for (uint32_t i = tid; i < (uint32_t)Ng * S; i += kBlock) s_small[i] = 0;
            for (int i = tid; i < Ng; i += kBlock) s_si[i] = 0;
            uint32_t* const s_any_hashed = (uint32_t*)(s_stat + 5);   // a zone went to the hash (else its second sweep has nothing to do)
            if (tid == 0) *s_any_hashed = 0;
            // The row sweep below occupies ONE wave (a serial chain over the rows); with accumulators of its own the NGTDM stencil
            // runs on the other three meanwhile instead of adding its time afterwards.
            const bool ngt_here = ngt_own && do_ngt && NgT >= 2 && w <= 64u * kSzmChunks;   // (the widths the register sweep below takes)
            if (ngt_here)
                for (uint32_t i = tid; i < ngt_rep * ngt_words; i += kBlock) ((uint32_t*)s_S)[i] = 0;
            blk_sync<GS>();
            if (ngt_here) {
                ngt_stencil_done = true;
                if (wave != solo) {
                    const int k = (wave - solo - 1) & 3, per = ((int)h + 2) / 3;                 // k = 0, 1, 2
                    const int rb = k * per, re = rb + per < (int)h ? rb + per : (int)h;
                    for (uint32_t c0 = 0; c0 < w; c0 += 64u) ngtdm_rows(rb, re, c0);
                }
            }
            // owner labels and zone sizes: wave 0 sweeps the rows, lanes own columns
            // (the one-wave stretches of this kernel rotate over the four waves -- hence the four SIMDs -- by ROI: with six
            //  workgroups per CU a fixed wave 0 would pile all of them onto one SIMD)
            if (wave == solo && w <= 64) {
                // bounding boxes up to 64 wide: the previous row stays in registers, neighbours come through DPP lane
                // shifts, the W chain is a prefix-min in DPP steps -- no LDS on the critical path.  A pixel travels as
                // X = level << 20 | owner label (levels <= 4094, labels < 2^20; all ones: no zone pixel), so ONE lane shift fetches a predecessor and "same level ?
                // its label : nothing" is X' - (level << 20): a label (< 2^20) on a match, something >= 2^20 otherwise (unsigned
                // wrap) -- three subtractions and three minima for N / NW / NE.  Lanes beyond the box read a zero of the level map
                // through a pointer with stride 0.  ~40 vector instructions per row (75 before: value and label travelled separately).
                const bool in = (uint32_t)lane < w;
                const dense_t* const col_ptr = in ? s_dense + lane : (const dense_t*)s_lvlmap;       // (s_lvlmap[0] == 0)
                const uint32_t col_stride = in ? w : 0u, p0 = in ? (uint32_t)lane : 0u;
                uint32_t Xp = 0xFFFFFFFFu;
                uint32_t vn = (uint32_t)col_ptr[0];
                for (uint32_t row = 0; row < h; row++) {
                    const uint32_t v = vn;
                    vn = (uint32_t)col_ptr[mul_u24_su(col_stride, row + 1u < h ? row + 1u : row)];
                    const uint32_t p = p0 + mul_u24_su(col_stride, row), V20 = v << 20;
                    // N, NW, NE predecessors (final labels of the previous row).  (A lane without a source reads 0: on a zone pixel
                    // 0 - V20 wraps beyond 2^20; on any other lane the label is never used.)
                    uint32_t lab = min(min(p, Xp - V20), min(lane_minus1(Xp, 0u) - V20, lane_plus1(Xp, 0u) - V20));
                    // W chain: inclusive prefix-min over runs of equal level as a PLAIN prefix-min -- a label travels with 63 - (index
                    // of its run) in bits 20..25, so whatever comes from an earlier run compares larger than anything of the lane's own.
                    const bool zp = v != 0;
                    const unsigned long long nz = __builtin_amdgcn_ballot_w64(zp);
                    const unsigned long long smask = ~nz | __builtin_amdgcn_ballot_w64(lane_minus1(v, 0u) != v);   // run starts; a lane that is no zone pixel is a run of its own
                    const uint32_t ridx = __builtin_amdgcn_mbcnt_hi((uint32_t)(smask >> 33), __builtin_amdgcn_mbcnt_lo((uint32_t)(smask >> 1), 0u));   // starts in lanes 1 .. lane
                    uint32_t key;
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(key) : "v"(ridx), "s"(-(1 << 20)), "v"(63 << 20));
                    key = wave_scan_min_u32((lab & 0xFFFFFu) | key);
                    lab = key & 0xFFFFFu;
                    const uint32_t X = zp ? (V20 | lab) : 0xFFFFFFFFu;
                    // zone sizes: one atomic per string of equal X in the row, by the string's first lane (lane 0 compares with the 0
                    // fill: always first); the string ends before the next lane that differs from its left neighbour
                    const bool differs = lane_minus1(X, 0u) != X;
                    const unsigned long long ends = (__builtin_amdgcn_ballot_w64(differs) >> 1) | ~nz | (1ull << 63);
                    if (zp && differs)
                        cnt_add(lab, 1u + (uint32_t)__builtin_ctzll(ends >> lane));
                    Xp = X;
                }
            } else if (wave == solo && w <= 64u * kSzmChunks) {
                // bounding boxes 65 .. 256 wide: the same register sweep over up to four chunks of 64 columns per row (lane = column
                // + 64 c).  What crosses a chunk boundary is wave-uniform and travels through v_readlane: the NE predecessor of a
                // chunk's column 63 / the NW predecessor and the W chain of the next chunk's column 0.  A run that continues into
                // the next chunk hands over its (already final) label before that chunk's scan; a string of equal labels is
                // counted once per chunk.  (The chunked LDS sweep below made a 65-wide box 4.6 times as expensive as a 63-wide one.)
                const int nch = (int)((w + 63u) >> 6);           // 2 .. kSzmChunks, wave-uniform
                uint32_t vp[kSzmChunks], lp[kSzmChunks];
                bool inc[kSzmChunks];
#pragma unroll
                for (int c = 0; c < kSzmChunks; c++) { vp[c] = 0; lp[c] = 0xFFFFFFFFu; inc[c] = (uint32_t)lane + 64u * c < w; }
                // (the plane row is read one row ahead: in the workspace build it comes from global memory, a round trip per row)
                uint32_t vn[kSzmChunks];
#pragma unroll
                for (int c = 0; c < kSzmChunks; c++) vn[c] = (c < nch && inc[c]) ? (uint32_t)s_dense[(uint32_t)lane + 64u * c] : 0u;
                for (uint32_t row = 0; row < h; row++) {
                    uint32_t v[kSzmChunks], lab[kSzmChunks];
#pragma unroll
                    for (int c = 0; c < kSzmChunks; c++) {
                        const uint32_t p = row * w + (uint32_t)lane + 64u * c;
                        v[c] = vn[c];
                        vn[c] = (c < nch && inc[c] && row + 1u < h) ? (uint32_t)s_dense[p + w] : 0u;
                        lab[c] = p;
                    }
#pragma unroll
                    for (int c = 0; c < kSzmChunks; c++) {       // N, NW, NE predecessors (final labels of the previous row)
                        if (c < nch) {
                            uint32_t fvW = 0u, flW = 0xFFFFFFFFu, fvE = 0u, flE = 0xFFFFFFFFu;
                            if (c > 0) { fvW = readlane63(vp[c > 0 ? c - 1 : 0]); flW = readlane63(lp[c > 0 ? c - 1 : 0]); }
                            if (c + 1 < kSzmChunks && c + 1 < nch) {
                                fvE = (uint32_t)__builtin_amdgcn_readfirstlane((int)vp[c + 1 < kSzmChunks ? c + 1 : c]);
                                flE = (uint32_t)__builtin_amdgcn_readfirstlane((int)lp[c + 1 < kSzmChunks ? c + 1 : c]);
                            }
                            const uint32_t vW = lane_minus1(vp[c], fvW), lW = lane_minus1(lp[c], flW);
                            const uint32_t vE = lane_plus1(vp[c], fvE), lE = lane_plus1(lp[c], flE);
                            if (v[c] != 0) {
                                if (vp[c] == v[c]) lab[c] = min(lab[c], lp[c]);
                                if (vW == v[c]) lab[c] = min(lab[c], lW);
                                if (vE == v[c]) lab[c] = min(lab[c], lE);
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < kSzmChunks; c++) {       // W chain, chunk after chunk
                        if (c < nch) {
                            const uint32_t carry_v = c > 0 ? readlane63(v[c > 0 ? c - 1 : 0]) : 0u;
                            const uint32_t vl = lane_minus1(v[c], carry_v);
                            const bool start = v[c] == 0 || vl != v[c];
                            if (c > 0) {
                                const uint32_t carry_l = readlane63(lab[c > 0 ? c - 1 : 0]);     // (final: the previous chunk is scanned)
                                if (lane == 0 && !start) lab[c] = min(lab[c], carry_l);
                            }
                            const unsigned long long smk = __ballot(start);
                            const uint32_t ri = __builtin_amdgcn_mbcnt_hi((uint32_t)(smk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)smk, 0u)) + (start ? 1u : 0u);
                            lab[c] = wave_scan_min_u32(((64u - ri) << 20) | lab[c]) & 0xFFFFFu;   // (lanes before the first start carry bias 64: seven bits)
                        }
                    }
#pragma unroll
                    for (int c = 0; c < kSzmChunks; c++) {       // zone sizes: one atomic per string of equal labels and chunk
                        if (c < nch) {
                            const bool zp = inc[c] && v[c] != 0;
                            const uint32_t ln = lane_plus1(lab[c], 0xFFFFFFFFu);                   // (lane 63 reads the fill: never continues)
                            const unsigned long long same = __ballot(zp && (uint32_t)lane + 64u * c + 1u < w && ln == lab[c]);
                            if (zp && !(lane > 0 && ((same >> (lane - 1)) & 1ull)))
                                cnt_add(lab[c], (uint32_t)__ffsll((long long)~(same >> lane)));
                            vp[c] = v[c];
                            lp[c] = zp ? lab[c] : 0xFFFFFFFFu;
                        }
                    }
                }
            } else if (wave == solo) {
                for (uint32_t row = 0; row < h; row++) {
                    uint32_t carry_v = 0, carry_l = 0;       // right-most pixel of the previous chunk
                    for (uint32_t c0 = 0; c0 < w; c0 += 64) {
                        const uint32_t cl = c0 + lane;
                        const bool in = cl < w;
                        const uint32_t p = row * w + cl;
                        const uint32_t v = in ? s_dense[p] : 0u;
                        uint32_t lab = p;
                        if (v != 0 && row > 0) {             // N, NW, NE predecessors (final labels)
                            const uint32_t q = p - w;
                            if (s_dense[q] == v) lab = min(lab, s_label[q]);
                            if (cl > 0 && s_dense[q - 1] == v) lab = min(lab, s_label[q - 1]);
                            if (cl + 1 < w && s_dense[q + 1] == v) lab = min(lab, s_label[q + 1]);
                        }
                        // W chain: segmented inclusive prefix-min over runs of equal value
                        uint32_t vl = __shfl_up(v, 1, 64);
                        if (lane == 0) vl = carry_v;
                        const bool start = v == 0 || vl != v;   // run starts here (or not a zone pixel)
                        if (lane == 0 && !start) lab = min(lab, carry_l);
                        const unsigned long long smask = __ballot(start);
                        const int run0 = 63 - __clzll((long long)(smask & ((2ull << lane) - 1ull)) | 1ll);
                        // (bit 0 is forced so that a chunk continuing the previous chunk's run starts at lane 0)
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) {
                            uint32_t o2 = __shfl_up(lab, d, 64);
                            if (lane - d >= run0) lab = min(lab, o2);
                        }
                        if (in && v != 0) s_label[p] = lab;
                        carry_v = __shfl(v, 63, 64);
                        carry_l = __shfl(lab, 63, 64);
                    }
                    wav_sync<GS>();
                }
            }
            blk_sync<GS>();
            TSTAMP(4);
            // zone sizes at the owners (the DPP sweep counted on the way)
            if (w > 64u * kSzmChunks) {
                for (uint32_t p = tid; p < area; p += kBlock)
                    if (s_dense[p] != 0)
                        cnt_add(s_label[p], 1u);
                blk_sync<GS>();
            }
            // zones -> P(i,j) multiplicities (hash), zones per level; Nz.
            // The table is an ORDERED linear-probing hash (Amble & Knuth): a key is displaced only by a larger one, so the
            // final layout is a function of the key SET, not of the order in which the lanes win their atomics -- the
            // floating-point sums over the table below then run in the same order on every launch (bit-reproducible output).
            // Keys first (atomicMax carries a displaced key onward), multiplicities in a second sweep once the layout is final.
            uint32_t nzone = 0, sz_max = 0;
            for (uint32_t p = tid; p < area; p += kBlock) {
                uint32_t sz = cnt_get(p);
                if (sz == 0) continue;
                nzone++;
                sz_max = sz > sz_max ? sz : sz_max;
                uint32_t rowi = (uint32_t)s_lvlmap[s_dense[p]] - 1;
                if (sz <= S) { atomicAdd(&s_small[rowi * 32u + (sz - 1u)], 1u); continue; }
                *s_any_hashed = 1u;
                uint32_t k = (rowi << 20) | sz;
                uint32_t hsl = (k * 2654435761u) & (hcap - 1);
                for (;;) {
                    const uint32_t old = atomicMax(&s_hkey[hsl], k);
                    if (old == k || old == 0) break;     // already present / placed in an empty slot
                    if (old < k) k = old;                // placed here: the displaced key moves on
                    hsl = (hsl + 1) & (hcap - 1);
                }
            }
            blk_sync<GS>();
            if (*s_any_hashed)
            for (uint32_t p = tid; p < area; p += kBlock) {
                uint32_t sz = cnt_get(p);
                if (sz <= S) continue;                   // no zone here, or counted in the direct table
                const uint32_t key = (((uint32_t)s_lvlmap[s_dense[p]] - 1) << 20) | sz;
                uint32_t hsl = (key * 2654435761u) & (hcap - 1);
                while (s_hkey[hsl] != key)
                    hsl = (hsl + 1) & (hcap - 1);
                atomicAdd(&s_hval[hsl], 1u);
            }
            nzone = wave_sum_t<uint32_t>(nzone);
            sz_max = wave_max_u32(sz_max);
            blk_sync<GS>();
            TSTAMP(5);
            if (lane == 0) { s_red[wave * 8] = (double)nzone; s_red[wave * 8 + 1] = (double)sz_max; }
            blk_sync<GS>();
            double sum_p = 0;
            for (int wv = 0; wv < kWaves; wv++) { sum_p += s_red[wv * 8]; const uint32_t m = (uint32_t)s_red[wv * 8 + 1]; sz_max = wv == 0 || m > sz_max ? m : sz_max; }
            // zones per size (sj): reuse s_count, keyed by size -- only the sizes that occur (<= sz_max) are cleared and read back;
            // boxes of 65536 pixels and more keep the full range (their empty columns at multiples of 65536 matter, see below)
            const uint32_t j_max = area >= 65536u ? area : sz_max;
            for (uint32_t i = tid; i < (c16 ? (j_max + 2) / 2 : j_max + 1); i += kBlock) s_count[i] = 0;
            blk_sync<GS>();
            for (uint32_t i = tid; i < n_cells; i += kBlock) {
                uint32_t key, val;
                cell(i, key, val);
                if (key != 0) {
                    cnt_add(key & 0xFFFFFu, val);
                    atomicAdd(&s_si[key >> 20], val);    // zones per level, from the (few) cells instead of one atomic per zone on Ng hot addresses
                }
            }
            blk_sync<GS>();
            TSTAMP(6);
            if (sum_p == 0) {                                // glszm.cpp:229-233
                for (int c = tid; c < 16; c += kBlock) o[c] = A.soft_nan;
            } else {
                // calc_sums_of_P :342-395 over the non-zero cells.  (Vector-instruction diet: reciprocals of i^2 / j^2 instead of a
                // Newton division per quotient, p / sum_p as a product, wave totals through transposed reductions.)
                const double inv_p = frcp(sum_p);
                double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                // (the entropy term of a cell is a function of its multiplicity, a small number: lane k holds the term of
                //  multiplicity k and a cell fetches it through ds_bpermute -- whole waves, so the loop runs in block-sized steps)
                const double ztab = plog_tex((double)lane * inv_p);
                for (uint32_t i0 = 0; i0 < n_cells; i0 += kBlock) {
                    const uint32_t i = i0 + (uint32_t)tid;
                    uint32_t key = 0, val = 0;
                    if (i < n_cells) cell(i, key, val);
                    double ze = __shfl(ztab, (int)(val & 63u), 64);
                    if (__builtin_amdgcn_ballot_w64(val >= 64u))
                        ze = val >= 64u ? plog_tex((double)val * inv_p) : ze;
                    if (key == 0) continue;
                    const double p = (double)val;
                    const double inten = (double)s_lv[key >> 20], jd = (double)(key & 0xFFFFFu);
                    double i2, ri2;
                    if (s_lvf) { i2 = s_lvf[2 * (key >> 20)]; ri2 = s_lvf[2 * (key >> 20) + 1]; }
                    else { i2 = inten * inten; ri2 = frcp(i2); }
                    const double j2 = jd * jd, rj2 = frcp(j2);   // (levels and sizes are >= 1)
                    const double pj = p * j2, pr = p * rj2;
                    acc[0] = __builtin_fma(pj, i2, acc[0]);  // f_LAHGLE
                    acc[1] = __builtin_fma(pj, ri2, acc[1]); // f_LALGLE
                    acc[2] = __builtin_fma(pr, i2, acc[2]);  // f_SAHGLE
                    acc[3] = __builtin_fma(pr, ri2, acc[3]); // f_SALGLE
                    const double pn = p * inv_p;
                    acc[4] += ze;                            // f_ZE
                    acc[5] = __builtin_fma(pn, jd, acc[5]);  // mu_ZV
                    acc[6] = __builtin_fma(pn, inten, acc[6]); // mu_GLV
                }
                // block reduction (fixed order): transposed wave sum, lane 8 k holds total k
                {
                    const double tt = wave_transpose_sum8(acc, lane);
                    if ((lane & 7) == 0) s_red[wave * 8 + (lane >> 3)] = tt;
                }
                blk_sync<GS>();
                // every thread needs the two means; the other five totals only feed outputs, which one thread writes right here
                // (carried by all threads through the second sweep they cost ten registers -- scratch spills in the 64-register build)
                const double mu_ZV = ((s_red[5] + s_red[8 + 5]) + s_red[16 + 5]) + s_red[24 + 5];
                const double mu_GLV = ((s_red[6] + s_red[8 + 6]) + s_red[16 + 6]) + s_red[24 + 6];
                if (tid == 0) {
#pragma unroll
                    for (int k = 0; k < 5; k++) acc[k] = ((s_red[k] + s_red[8 + k]) + s_red[16 + k]) + s_red[24 + k];
                    o[Z_ZE] = -acc[4];
                    o[Z_SALGLE] = acc[3] * inv_p;
                    o[Z_SAHGLE] = acc[2] * inv_p;
                    o[Z_LALGLE] = acc[1] * inv_p;
                    o[Z_LAHGLE] = acc[0] * inv_p;
                }
                blk_sync<GS>();
                TSTAMP(7);
                double b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (uint32_t i = tid; i < n_cells; i += kBlock) {
                    uint32_t key, val;
                    cell(i, key, val);
                    if (key == 0) continue;
                    const double p = (double)val * inv_p;
                    const double dg = (double)s_lv[key >> 20] - mu_GLV, dz = (double)(key & 0xFFFFFu) - mu_ZV;
                    b[0] += p * (dg * dg);                   // calc_GLV :497-510
                    b[1] += p * (dz * dz);                   // calc_ZV :512-524
                }
                for (uint32_t j = 1 + tid; j <= j_max; j += kBlock) {
                    const uint32_t sji = cnt_get(j);
                    // j * j is an int product in the reference: it wraps for j >= 46341 and is exactly 0 at multiples of
                    // 65536, where the empty column contributes 0.0 / 0 = NaN to SAE (Ns = bbox area, glszm.cpp:212)
                    if (sji == 0 && (j & 0xFFFFu) != 0) continue;
                    const double sj = (double)sji;
                    if (j < 32768u) {                        // (the product is exact and positive: a reciprocal serves)
                        const double jj = (double)mul24(j, j);
                        b[2] += sj * frcp(jj);               // calc_SAE :419-428
                        b[3] += sj * jj;                     // calc_LAE :430-439
                    } else {
                        const int jj = (int)(j * j);
                        b[2] += sj / (double)jj;
                        b[3] += sj * (double)jj;
                    }
                    b[4] += sj * sj;                         // calc_SZN :464-474
                }
                for (int i = tid; i < Ng; i += kBlock) {
                    const double si = (double)s_si[i], inten = (double)s_lv[i];
                    double i2, ri2;
                    if (s_lvf) { i2 = s_lvf[2 * i]; ri2 = s_lvf[2 * i + 1]; }
                    else { i2 = inten * inten; ri2 = frcp(i2); }
                    b[5] += si * si;                         // calc_GLN :441-451
                    b[6] += si * ri2;                        // calc_LGLZE :531-541
                    b[7] += si * i2;                         // calc_HGLZE :543-553
                }
                {
                    const double tt = wave_transpose_sum8(b, lane);
                    if ((lane & 7) == 0) s_red[wave * 8 + (lane >> 3)] = tt;
                }
                blk_sync<GS>();
                if (tid == 0) {
                    for (int k = 0; k < 8; k++) b[k] = ((s_red[k] + s_red[8 + k]) + s_red[16 + k]) + s_red[24 + k];
                    const double inv_p2 = inv_p * inv_p;
                    o[Z_SAE] = b[2] * inv_p;
                    o[Z_LAE] = b[3] * inv_p;
                    o[Z_GLN] = b[5] * inv_p;
                    o[Z_GLNN] = b[5] * inv_p2;
                    o[Z_SZN] = b[4] * inv_p;
                    o[Z_SZNN] = b[4] * inv_p2;
                    o[Z_ZP] = fdiv(sum_p, (double)(int)(uint32_t)s_stat[3]);   // calc_ZP :491-495
                    o[Z_GLV] = b[0];
                    o[Z_ZV] = b[1];
                    o[Z_LGLZE] = b[6] * inv_p;
                    o[Z_HGLZE] = b[7] * inv_p;
                }
            }
        }
        blk_sync<GS>();
    }

    TSTAMP(8);
    // =====================================================================================
    // NGTDM
    // =====================================================================================
    if (do_ngt) {
        double* o = s_out + col;
        col += 5;
        double* s_P = (double*)(s_work + A.L.ngt_p);               // [NgT] (behind the S / N arrays when those are aliased)
        double* s_Sd = s_P + A.L.ng_cap + 2;                      // [NgT]
        if (NgT < 2) {                                            // ngtdm.cpp:70-78
            for (int c = tid; c < 5; c += kBlock) o[c] = A.soft_nan;
        } else {
            if (!ngt_stencil_done) {
                if (ngt_rep > 1u)
                    for (uint32_t i = tid; i < ngt_rep * ngt_words; i += kBlock) ((uint32_t*)s_S)[i] = 0;
                else
                    for (int i = tid; i < NgT; i += kBlock) { s_S[i] = 0; s_N[i] = 0; }
                blk_sync<GS>();
            }
            if (ngt_stencil_done) {
                // (the sums were taken during the GLSZM row sweep)
            } else if (w <= 64u * kSzmChunks) {
                const int rows_per_wave = ((int)h + kWaves - 1) / kWaves;
                const int r_begin = wave * rows_per_wave;
                for (uint32_t c0 = 0; c0 < w; c0 += 64u)
                    ngtdm_rows(r_begin, (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h, c0);
            } else {
            RowCol rc((uint32_t)tid, kBlock, w);
            for (uint32_t p = tid; p < area; p += kBlock, rc.advance()) {
                uint32_t pi = s_dense[p];
                if (pi == 0) continue;
                const int row = (int)rc.row, cl = (int)rc.col;
                uint32_t sum = 0, nd = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {                     // N,NE,E,SE,S,SW,W,NW (ngtdm.cpp:92-139)
                    const int oy = k == 0 || k == 1 || k == 7 ? -1 : (k == 2 || k == 6 ? 0 : 1);
                    const int ox = k == 1 || k == 2 || k == 3 ? 1 : (k == 0 || k == 4 ? 0 : -1);
                    int r2 = row + oy, c2 = cl + ox;
                    if (r2 >= 0 && r2 < (int)h && c2 >= 0 && c2 < (int)w) {
                        uint32_t q = s_dense[(uint32_t)r2 * w + c2];
                        if (q != 0) { sum += q; nd++; }
                    }
                }
                if (nd > 0) {
                    int r = greyInfo == 0 ? (int)pi : (int)s_lvlmap[pi] - 1;
                    // |pi - sum/nd| * 840 = |840*pi - sum*(840/nd)|, exact integers
                    long long t = 840ll * (long long)pi - (long long)sum * (long long)(840u / nd);
                    atomicAdd(&s_N[r], 1u);
                    atomicAdd(&s_S[r], (unsigned long long)(t < 0 ? -t : t));
                }
            }
            }
            if (ngt_rep > 1u) {                                   // replica 0 collects the others
                blk_sync<GS>();
                for (int i = tid; i < NgT; i += kBlock) {
                    unsigned long long sS = s_S[i];
                    uint32_t sN = s_N[i];
                    for (uint32_t r = 1; r < ngt_rep; r++) { sS += s_S[(size_t)r * (ngt_words >> 1) + i]; sN += s_N[r * ngt_words + i]; }
                    s_S[i] = sS; s_N[i] = sN;
                }
            }
            blk_sync<GS>();
            TSTAMP(9);
            // Nvc = Nvp = number of pixels with a neighbourhood (every mean is > 0), ngtdm.cpp:176-186
            uint32_t nvc_part = 0;
            for (int i = tid; i < NgT; i += kBlock) nvc_part += s_N[i];
            nvc_part = wave_sum_t<uint32_t>(nvc_part);
            if (lane == 0) s_red[wave * 8] = (double)nvc_part;
            blk_sync<GS>();
            const double Nvc = ((s_red[0] + s_red[8]) + s_red[16]) + s_red[24];
            {   // (quotients through reciprocals: every NGTDM value is tolerance-class, and this block runs on all four waves)
                const double inv_nvc = frcp(Nvc);
                for (int i = tid; i < NgT; i += kBlock) {
                    s_P[i] = (double)s_N[i] * inv_nvc;
                    s_Sd[i] = (double)s_S[i] * (1.0 / 840.0);
                }
            }
            blk_sync<GS>();
            if (wave == solo) {
                auto Iof = [=](int i) -> double { return greyInfo == 0 ? (double)i : (double)s_lv[i]; };
                // (one wave: the cell's (i, j) from a float reciprocal while Ng^2 stays small, the quotient of the complexity term by a
                //  reciprocal, the six totals through one transposed reduction)
                double t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};                        // ps, ssum, contrast, busyness, complexity, strength
                for (int i = lane; i < NgT; i += 64) { t8[0] += s_P[i] * s_Sd[i]; t8[1] += s_Sd[i]; }
                const bool small_ng = NgT <= 256;
                const float inv_ng = 1.0f / (float)NgT;
                for (int e = lane; e < NgT * NgT; e += 64) {
                    int i = small_ng ? (int)(((float)e + 0.5f) * inv_ng) : e / NgT;   // (e < 2^16: the product is off by < 1e-2 from (e + 0.5) / Ng, never across an integer)
                    const int j = e - i * NgT;
                    double pi_ = s_P[i], pj = s_P[j], iv = Iof(i), jv = Iof(j);
                    double d = iv - jv;
                    t8[2] += pi_ * pj * d * d;                                  // calc_Contrast :245-247
                    if (pi_ != 0 && pj != 0) {
                        t8[3] += fabs(pi_ * iv - pj * jv);                       // calc_Busyness :280-283
                        t8[4] += fdiv(fabs(d) * (pi_ * s_Sd[i] + pj * s_Sd[j]), pi_ + pj); // calc_Complexity :305
                        t8[5] += (pi_ + pj) * d * d;                             // calc_Strength :326
                    }
                }
                {
                    const double tt = wave_transpose_sum8(t8, lane);             // lane 8 k holds total k
                    if ((lane & 7) == 0) s_red[lane >> 3] = tt;
                }
                wav_sync<GS>();
                const double ps = s_red[0], ssum = s_red[1], c_sum = s_red[2], b_sum = s_red[3], x_sum = s_red[4], s_sum = s_red[5];
                if (lane == 0) {
                    int Ngp = Nuniq;
                    int Ngp_p2 = Ngp > 1 ? Ngp * (Ngp - 1) : Ngp;
                    o[0] = 1.0 / ps;                                             // calc_Coarseness :228-236
                    o[1] = (c_sum / (double)Ngp_p2) * (ssum / Nvc);              // calc_Contrast :251-261
                    o[2] = Ngp == 1 ? 0.0 : (b_sum == 0 ? 0.0 : ps / b_sum);     // calc_Busyness :266-291
                    o[3] = x_sum / (double)(int)Nvc;                             // calc_Complexity :310 (Nvp)
                    o[4] = s_sum / ssum;                                         // calc_Strength :331-335
                }
            }
        }
        blk_sync<GS>();
    }

    blk_sync<GS>();
    for (int c = tid; c < A.n_cols; c += kBlock)
        out_row[gcol(c)] = s_out[c];
}

static int tex_max_occ()   // diagnostic knob: NYXHIP_TEX_OCC=4 keeps the 106-register build
{
    static const int v = [] { const char* e = getenv("NYXHIP_TEX_OCC"); return e && *e ? atoi(e) : 8; }();
    return v;
}

int launch_roi_texture(const TexArgs& a, void* stream, uint32_t grid)
{
    static DeviceOnce optin;
    if (int orc = optin.run([]() -> int {
        hipError_t e = hipFuncSetAttribute((const void*)roi_texture_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)roi_features_max_lds());
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)roi_texture_kernel<false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)roi_features_max_lds());
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)roi_texture_kernel<false, 7>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)roi_features_max_lds());
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)roi_texture_kernel<false, 7, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)roi_features_max_lds());
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)roi_texture_kernel<false, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)roi_features_max_lds());
        return (int)e;
    }))
        return orc;
    if (grid == 0)
        return 0;
    if (getenv("NYXHIP_DEBUG")) fprintf(stderr, "[nyxhip] texture launch: dense8 %u ng_cap %u total %u work %u ngt_rep %u mask %u\n", a.L.dense8, a.L.ng_cap, a.L.total, a.L.work_bytes, a.L.ngt_rep, a.mask);
    if (a.sp.scratch)
        hipLaunchKernelGGL((roi_texture_kernel<true, 2>), dim3(grid), dim3(kBlock), a.L.gs_lds_bytes, (hipStream_t)stream, a);
    // (LDS is handed out in 1280-byte granules: k workgroups share a CU when k rounded-up carve-outs fit)
    else if (a.L.dense8 && tex_max_occ() >= 8 && 8ull * (((size_t)a.L.total + 1279) / 1280 * 1280) <= roi_features_max_lds())
        hipLaunchKernelGGL((roi_texture_kernel<false, 8, true>), dim3(grid), dim3(kBlock), a.L.total, (hipStream_t)stream, a);
    else if (a.L.dense8)
        hipLaunchKernelGGL((roi_texture_kernel<false, 7, true>), dim3(grid), dim3(kBlock), a.L.total, (hipStream_t)stream, a);
    else if (tex_max_occ() >= 7 && 7ull * (((size_t)a.L.total + 1279) / 1280 * 1280) <= roi_features_max_lds())
        hipLaunchKernelGGL((roi_texture_kernel<false, 7>), dim3(grid), dim3(kBlock), a.L.total, (hipStream_t)stream, a);
    else if (tex_max_occ() >= 6 && 6ull * (a.L.total + 256) <= roi_features_max_lds())
        hipLaunchKernelGGL((roi_texture_kernel<false, 6>), dim3(grid), dim3(kBlock), a.L.total, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((roi_texture_kernel<false, 4>), dim3(grid), dim3(kBlock), a.L.total, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

} // namespace nyxhip
