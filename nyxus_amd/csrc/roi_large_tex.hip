// roi_large_tex.hip -- GLRLM + GLSZM + NGTDM of ROIs beyond the LDS size classes, several workgroups per ROI (gfx950).
//
// The reference hands any ROI to any worker thread for every family (/root/reference/src/nyx/parallel.h:23-42;
// features/glrlm.cpp:20-276, glszm.cpp:56-340, ngtdm.cpp:33-226).  roi_texture.hip gives an ROI one workgroup; with its state in a
// global workspace a 120 k-cell box then costs milliseconds (ten-odd passes over the box by 256 threads).  Here the box is cut up:
//
//   ltex_prep_kernel    one thread per ROI: its block of the workspace, its slabs and strips in the two work maps
//   ltex_load_kernel    one workgroup per SLAB of the pixel cloud: the binned plane (texture_feature.h binning; 0 = not written,
//                       which matlab binning reads as its background level 1), the levels present, the two pixel counts
//   ltex_strip_kernel   one workgroup per STRIP of plane rows, staged in LDS with a one-row halo:
//                         NGTDM  3 x 3 stencil, N[level] and sum |i - mean| in units of 1/840 (ngtdm.cpp:83-183): integers
//                         GLRLM  0 degrees row by row; 45 / 90 / 135 degrees a lane per LINE of the direction.  Runs inside the
//                                strip are counted; a run that touches the strip's first or last row is RECORDED per column
//                                (level, length) and joined with its continuation by the finishing workgroup
//                       and, in the same launch, one workgroup per ROI for the
//                         GLSZM  owner sweep (glszm.cpp:108-185 as directed reachability, roi_texture.hip): a chain over the rows, run as a
//                                pipeline -- one wave per 64 columns, chunk j of row r at step 2 r + j; zone sizes by atomics at the
//                                owner pixel
//   ltex_post_kernel    one workgroup per strip: joins the recorded runs that start in its strip with their continuations and counts
//                       them; zones -> (level, size) multiplicities (direct table for sizes <= 32, a list for the few larger
//                       ones), number of zones, largest zone
//   ltex_finish_kernel  one workgroup per ROI: the 80 + 16 + 5 columns in a fixed order
//
// Everything that crosses a workgroup is an integer (atomic adds) or a record written once, so a row does not depend on the cut,
// the arrival order, the companions or the workspace budget.  Kernel boundaries are the only synchronisation.
// Built with -ffp-contract=off (device_math.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "device_math.h"
#include "roi_kernel.h"
#include "launch_util.h"
#include "texture_feats.h"
#include "../../include/nyxhip.h"

namespace nyxhip {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kLenMask = 0xFFFFFu;                   // a record / label: level << 20 | length (or owner index)

struct LtexRoi {
    uint64_t roi, off;
    uint32_t n, w, h, vmin, vmax, area, side;
    uint32_t ng;                                           // bound of the level count = capacity of the level map
    LtexWs L;
    unsigned char* base;
    uint32_t colmajor;                                     // the cloud runs down the columns (phase2_2d.cpp:655-656): the load kernel transposes through LDS
};

__device__ __forceinline__ bool ltex_roi(const LtexArgs& A, uint32_t j, LtexRoi& R, bool need_block)
{
    R.roi = A.list[j];
    R.off = A.px_offset[R.roi];
    R.n = (uint32_t)(A.px_offset[R.roi + 1] - R.off);
    R.w = A.bbox_w[R.roi]; R.h = A.bbox_h[R.roi];
    R.vmin = A.min_inten[R.roi]; R.vmax = A.max_inten[R.roi];
    R.side = R.w > R.h ? R.w : R.h;
    R.base = nullptr;
    if (!ltex_eligible(R.n, R.w, R.h)) return false;
    R.area = R.w * R.h;
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    R.ng = greyInfo > 0 ? (uint32_t)greyInfo : greyInfo < 0 ? (uint32_t)(-greyInfo) : R.vmax;
    if (R.ng > kLtexLevels) return false;                  // (the host sends such classes down the one-workgroup path, which reports them)
    R.L = ltex_ws_layout(R.w, R.h, R.ng, A.plane16 != 0, A.mask);
    if (need_block) {
        const uint64_t o = A.ws_off[j];
        if (o == ~0ull) return false;
        R.base = A.ws + (o & ~255ull);                      // (blocks are 256-byte aligned: bit 0 carries the scan order)
        R.colmajor = (uint32_t)(o & 1ull);
    }
    return true;
}

// ---- prep: one thread per member, one set of cursor adds per block (roi_large.hip: large_prep_kernel) --------------------------
__global__ __launch_bounds__(256) void ltex_prep_kernel(const LtexArgs A)
{
    __shared__ unsigned long long s_bytes[256];
    __shared__ uint32_t s_load[256], s_strip[256];
    __shared__ unsigned long long s_base_bytes;
    __shared__ uint32_t s_base_load, s_base_strip;
    const int tid = threadIdx.x;
    const uint32_t j = blockIdx.x * 256u + (uint32_t)tid;
    LtexRoi R;
    bool served = false;
    uint32_t g_load = 0, g_strip = 0;
    if (j < A.n_list) {
        served = ltex_roi(A, j, R, false);
        if (served) {
            g_load = (uint32_t)(((R.off & 3ull) + R.n + A.px_per_wg - 1) / A.px_per_wg);
            g_strip = R.L.K;
        }
    }
    s_bytes[tid] = served ? R.L.total : 0ull; s_load[tid] = g_load; s_strip[tid] = g_strip;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = 0; uint32_t l = 0, c = 0;
        for (int k = 0; k < 256; k++) {
            const unsigned long long tb = s_bytes[k]; const uint32_t tl = s_load[k], tc = s_strip[k];
            s_bytes[k] = b; s_load[k] = l; s_strip[k] = c;
            b += tb; l += tl; c += tc;
        }
        s_base_bytes = b ? atomicAdd((unsigned long long*)A.ctr, b) : 0ull;
        s_base_load = l ? atomicAdd(&A.ctr[2], l) : 0u;
        s_base_strip = c ? atomicAdd(&A.ctr[3], c) : 0u;
    }
    __syncthreads();
    if (j >= A.n_list) return;
    if (!served) { A.ws_off[j] = ~0ull; return; }
    const unsigned long long off = s_base_bytes + s_bytes[tid];
    const uint32_t b_load = s_base_load + s_load[tid], b_strip = s_base_strip + s_strip[tid];
    const bool fits = off + R.L.total <= A.ws_bytes && (uint64_t)b_load + g_load <= A.cap_load && (uint64_t)b_strip + g_strip <= A.cap_strip;
    if (!fits) {                                            // (the host sized all three from the class totals: cannot happen)
        A.ws_off[j] = ~0ull;
        atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        return;
    }
    // scan order of the cloud, read off a pair of pixels in its middle (the first column of a disk is a single pixel)
    const uint64_t m = R.off + (R.n >= 2 ? R.n / 2 - 1 : 0);
    const uint32_t cm = (R.n >= 2 && A.x[m + 1] == A.x[m] && (uint32_t)A.y[m + 1] == (uint32_t)A.y[m] + 1u) ? 1u : 0u;
    A.ws_off[j] = off | cm;
    for (uint32_t s = 0; s < g_load; s++) A.map_load[b_load + s] = make_uint2(j, s);
    for (uint32_t s = 0; s < g_strip; s++) A.map_strip[b_strip + s] = make_uint2(j, s);
}

// ---- load: one workgroup per slab of the cloud ---------------------------------------------------------------------------------
// The plane is row-major (the GLSZM sweep follows the raster order).  A cloud that runs down the columns (the in-memory workflow)
// would store every pixel of a wave to a cache line of its own -- the stores were nine tenths of this kernel's time -- so the
// slab's levels go to an LDS tile first (the slab covers a few whole columns of the box: [its first pixel's x, its last pixel's
// x] x all rows) and leave it row by row.  Cells that stayed 0 are not stored (the slabs that share a boundary column each write
// their own pixels; 0 is what the zeroed workspace holds anyway).  Pixels outside the tile -- a cloud in any other order -- are
// stored directly.
template <bool P16>
__global__ __launch_bounds__(256) void ltex_load_kernel(const LtexArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= A.ctr[2]) return;
    const uint2 job = A.map_load[blockIdx.x];
    LtexRoi R;
    if (!ltex_roi(A, job.x, R, true)) return;
    constexpr int BS = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    using plane_t = typename std::conditional<P16, uint16_t, uint8_t>::type;
    plane_t* const plane = (plane_t*)(R.base + R.L.plane);
    uint8_t* const flags = R.base + R.L.flags;
    uint32_t* const hdr = (uint32_t*)R.base;
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    const double mslope = greyInfo > 0 ? (double)greyInfo / ((double)R.vmax - 0.) : 0.0;
    const uint32_t Lcap = R.ng;
    // matlab binning: the background is level 1 -- present when the box has a cell outside the list (texture_feature.h:150-154)
    if (job.y == 0 && tid == 0 && greyInfo > 0 && R.n < R.area) flags[1] = 1;
    const uint64_t roi_lo = R.off, roi_hi = R.off + R.n;
    const uint64_t gs = (R.off & ~3ull) + (uint64_t)job.y * A.px_per_wg;      // slabs are cut in the batch's global pixel index (roi_large.hip)
    const uint64_t ge = gs + A.px_per_wg < roi_hi ? gs + A.px_per_wg : roi_hi;
    const uint32_t w = R.w, h = R.h;
    // the tile: columns [tx0, tx0 + tnc) of the box, all rows, column-major in LDS
    plane_t* const tile = (plane_t*)lds_raw;
    uint32_t tx0 = 0, tnc = 0;
    if (R.colmajor && ge > (gs > roi_lo ? gs : roi_lo)) {
        const uint64_t first = gs > roi_lo ? gs : roi_lo;
        const uint32_t xa = A.x[first], xb = A.x[ge - 1];
        const uint32_t fit = A.lds_load_bytes / ((uint32_t)sizeof(plane_t) * h);
        if (xb >= xa && xa < w && fit) { tx0 = xa; tnc = min(min(xb - xa + 1u, fit), w - xa); }
    }
    if (tnc) {
        uint32_t* const t32 = (uint32_t*)tile;
        for (uint32_t i = tid; i < (tnc * h * (uint32_t)sizeof(plane_t) + 3u) / 4u; i += BS) t32[i] = 0;
        __syncthreads();
    }
    uint32_t nz_orig = 0, nz_bin = 0;
    auto pixel = [&](uint32_t v, uint32_t px, uint32_t py) {
        uint32_t lvl;
        if (greyInfo > 0) {     // bin_matlab: floor(slope v + 1) >= 1, 0 -> 1 (the conversion truncates a positive value)
            const uint32_t sc = (uint32_t)(mslope * (double)v + 1.0);
            lvl = sc > (uint32_t)greyInfo ? (uint32_t)greyInfo : sc;
        } else
            lvl = greyInfo < 0 ? bin_radiomix(v, R.vmin, R.vmax, -greyInfo) : v;
        nz_orig += v != 0;
        if (lvl > Lcap) lvl = Lcap;
        if (px < w && py < h) {
            if (px - tx0 < tnc) tile[(px - tx0) * h + py] = (plane_t)lvl;
            else plane[py * w + px] = (plane_t)lvl;
            if (lvl != 0) { flags[lvl] = 1; nz_bin++; }
        }
    };
    constexpr int kU = 4;
    for (uint64_t gb = gs; gb < ge; gb += 4ull * kU * BS) {
        uint4 v4[kU]; uint2 x2[kU], y2[kU];
        bool whole[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const uint64_t g = gb + 4ull * ((uint64_t)u * BS + tid);
            whole[u] = A.vec_ok && g >= roi_lo && g + 4 <= ge;
            if (whole[u]) {
                v4[u] = *(const uint4*)(A.inten + g);
                x2[u] = *(const uint2*)(A.x + g); y2[u] = *(const uint2*)(A.y + g);
            }
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const uint64_t g = gb + 4ull * ((uint64_t)u * BS + tid);
            if (whole[u]) {
                pixel(v4[u].x, x2[u].x & 0xFFFFu, y2[u].x & 0xFFFFu);
                pixel(v4[u].y, x2[u].x >> 16, y2[u].x >> 16);
                pixel(v4[u].z, x2[u].y & 0xFFFFu, y2[u].y & 0xFFFFu);
                pixel(v4[u].w, x2[u].y >> 16, y2[u].y >> 16);
            } else {
                for (uint64_t i = g > roi_lo ? g : roi_lo; i < g + 4 && i < ge; i++)
                    pixel(A.inten[i], (uint32_t)A.x[i], (uint32_t)A.y[i]);
            }
        }
    }
    nz_orig = wave_sum_t<uint32_t>(nz_orig);
    nz_bin = wave_sum_t<uint32_t>(nz_bin);
    if (lane == 0) {
        if (nz_orig) atomicAdd(&hdr[LTEX_H_NP_ORIG], nz_orig);
        if (nz_bin) atomicAdd(&hdr[LTEX_H_NP_BIN], nz_bin);
    }
    if (tnc) {
        __syncthreads();
        RowCol rc((uint32_t)tid, BS, tnc);                  // row of the box, column of the tile
        for (uint32_t i = tid; i < tnc * h; i += BS, rc.advance()) {
            const plane_t v = tile[rc.col * h + rc.row];
            if (v != 0) plane[rc.row * w + tx0 + rc.col] = v;
        }
    }
}

// ---- the levels of an ROI: level -> row + 1, row -> level (glrlm.cpp:101-105, glszm.cpp:97-101, ngtdm.cpp:53-67) -----------------
// IBSI: rows are the levels 1 .. max themselves.  By wave 0; the caller's barrier publishes the result.
struct LtexLevels { int Ng, Nuniq; };
__device__ __forceinline__ void ltex_levels(const uint8_t* flags, uint32_t Lcap, int greyInfo, uint16_t* s_lvlmap, uint32_t* s_lv, int* s_res, int tid)
{
    if (tid >= 64) return;
    const int lane = tid;
    uint32_t k = 0, mx = 0;
    if (lane == 0) s_lvlmap[0] = 0;
    for (uint32_t l0 = 1; l0 <= Lcap; l0 += 64) {
        const uint32_t l = l0 + (uint32_t)lane;
        const bool on = l <= Lcap && flags[l] != 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (l <= Lcap && greyInfo != 0) {
            s_lvlmap[l] = on ? (uint16_t)(k + below + 1) : (uint16_t)0;
            if (on && s_lv) s_lv[k + below] = l;
        }
        k += (uint32_t)__builtin_popcountll(m);
        if (m) mx = l0 + 63u - (uint32_t)__builtin_clzll(m);
    }
    if (greyInfo == 0) {
        for (uint32_t l = 1 + (uint32_t)lane; l <= Lcap; l += 64) {
            s_lvlmap[l] = l <= mx ? (uint16_t)l : (uint16_t)0;
            if (l <= mx && s_lv) s_lv[l - 1] = l;
        }
    }
    if (lane == 0) { s_res[0] = greyInfo == 0 ? (int)mx : (int)k; s_res[1] = (int)k; }
}

__device__ __forceinline__ uint32_t align16u(uint32_t v) { return (v + 15u) & ~15u; }

// NGTDM accumulator replicas in LDS (roi_texture.hip): few levels mean few addresses under 64-lane atomics
__device__ __forceinline__ uint32_t ltex_ngt_rep(uint32_t ng1) { return ng1 <= 16 ? 8u : ng1 <= 32 ? 4u : ng1 <= 64 ? 2u : 1u; }
__device__ __forceinline__ uint32_t ltex_ngt_stride(uint32_t ng1) { return (((ng1 + 2) * 12u + 16u + 7u) & ~7u) | 8u; }

// GLRLM: runs of up to kLtexRlmLds pixels are counted in LDS first, in REPLICAS of a [direction][length][level] table: with a
// handful of levels and most runs one or two pixels long, the 64 lanes of an atomic would meet on a dozen addresses, which LDS
// serialises (two thirds of the strip kernel's time).  A lane adds into replica lane % R; levels are the minor index and the
// replicas lie Lcap words (mod 64) apart, so the lanes of one instruction spread over the banks.
__device__ __forceinline__ uint32_t ltex_rl_rep(uint32_t Lcap) { return Lcap <= 16 ? 8u : Lcap <= 32 ? 4u : Lcap <= 64 ? 2u : 1u; }
__device__ __forceinline__ uint32_t ltex_rl_words(uint32_t Lcap) { return 4u * kLtexRlmLds * Lcap + Lcap; }
__device__ __forceinline__ uint32_t ltex_rl_index(uint32_t Lcap, uint32_t dir, uint32_t m, uint32_t len) { return (dir * kLtexRlmLds + (len - 1u)) * Lcap + m; }
// the replicas of a table summed into the global matrices (integers: any order)
__device__ __forceinline__ void ltex_rl_flush(const uint32_t* s_short, uint32_t Lcap, uint32_t* gP, uint32_t slot_words, uint32_t Nr, int tid, int BS, uint32_t rep)
{
    const uint32_t words = ltex_rl_words(Lcap);
    for (uint32_t i = (uint32_t)tid; i < 4u * kLtexRlmLds * Lcap; i += (uint32_t)BS) {
        uint32_t cn = 0;
        for (uint32_t r = 0; r < rep; r++) cn += s_short[r * words + i];
        if (cn == 0) continue;
        const uint32_t dl = i / Lcap, m = i - dl * Lcap, dir = dl / kLtexRlmLds, j = dl - dir * kLtexRlmLds;
        if (j < Nr) atomicAdd(&gP[dir * slot_words + m * Nr + j], cn);
    }
}

// in place: a zero level becomes 1 (matlab binning's background), four bytes / two half-words at a time
template <bool P16>
__device__ __forceinline__ uint32_t zero_to_one(uint32_t v)
{
    uint32_t m;
    if (P16) { m = v | (v >> 8); m |= m >> 4; m |= m >> 2; m |= m >> 1; return v | (~m & 0x00010001u); }
    m = v | (v >> 4); m |= m >> 2; m |= m >> 1;
    return v | (~m & 0x01010101u);
}

__device__ __forceinline__ void lds_barrier()               // workgroup barrier that orders LDS only: loads from global memory stay in flight
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- strips: NGTDM + GLRLM; and the GLSZM owner sweep, one workgroup per ROI in front of them -------------------------------------
template <bool P16>
__global__ __launch_bounds__(1024) void ltex_strip_kernel(const LtexArgs A, uint32_t n_sweep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_res_all[8];
    int BS = (int)blockDim.x, NW = BS >> 6;
    int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    using plane_t = typename std::conditional<P16, uint16_t, uint8_t>::type;
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    const uint32_t bgmin = greyInfo > 0 ? 1u : 0u;         // an unwritten cell of the box: background, level 1 under matlab binning
    // The workgroup size follows the sweep (a wave per 64 columns), but a strip is four waves' work: a strip workgroup takes one strip
    // per GROUP of four waves, each group with its own part of the dynamic LDS (waves beyond the last whole group leave at once).
    // The groups share the workgroup's barriers and nothing else.
    unsigned char* lds_g = lds_raw;
    int grp = 0;
    if (blockIdx.x >= n_sweep && NW > 4) {
        grp = wave >> 2;
        if (grp >= (int)A.strip_groups) return;
        lds_g = lds_raw + (size_t)grp * A.lds_group_bytes;
        tid &= 255; wave &= 3; NW = 4; BS = 256;
    }

    if (blockIdx.x < n_sweep) {
        // =============================================================================================================
        // GLSZM owner sweep of member blockIdx.x: owner(p) = min(p, owner(W), owner(NW), owner(N), owner(NE)) over the predecessors of
        // p's level (roi_texture.hip).  A cell travels as X = level << 20 | owner; the X of the last two rows sit in LDS between
        // border entries.  The chain over the rows runs as a pipeline of 64-column chunks that DRIFT one column to the left per row
        // (chunk j of row r covers the columns 64 j - (r mod 64) + 0 .. 63; every 64 rows the drift starts over): then the three
        // predecessors of a chunk's cells in the previous row lie in the same chunk and its left neighbour, and chunk j of row r can
        // run at step r + floor(r / 64) + j -- one step per row and chunk diagonal instead of one per row and chunk (the start-over
        // costs one idle step per 64 rows: the NE cell of the last lane then belongs to the right neighbour).
        // =============================================================================================================
        LtexRoi R;
        if (blockIdx.x >= A.n_list || !ltex_roi(A, blockIdx.x, R, true)) return;
        if (R.vmin == R.vmax) return;                       // blank: no zones are asked for (glszm.cpp:61-65)
        const plane_t* const plane = (const plane_t*)(R.base + R.L.plane);
        uint32_t* const labp = (uint32_t*)(R.base + R.L.lab);
        const uint32_t w = R.w, h = R.h;
        const uint32_t nch = (w + 62u) / 64u + 1u;          // chunks 0 .. nch - 1 cover the columns -63 .. w - 1 at every drift
        // Plane rows come through an LDS ring filled by the workgroup's LAST wave, which requests the rows eight rows ahead and does
        // nothing else: the sweeping waves then have no loads from global memory in flight -- next to their stores those
        // made every wait a wait for everything (vmcnt counts both, out of order), one L2 round trip per step.
        // (Rows beyond 1008 bytes -- one 16-byte load per lane at any misalignment -- are read directly: every wave sweeps.)
        const uint32_t row_bytes = w * (uint32_t)sizeof(plane_t);
        const bool piped = row_bytes <= 1008u && NW >= 2;
        const int NWs = piped ? NW - 1 : NW;                // sweeping waves
        constexpr uint32_t kSlot = 1040u, kAhead = 8u;
        const uint32_t RR = kAhead + nch + 4u;              // ring slots (row r: slot r mod RR); slot RR: requests that carry no row
        uint32_t* const rowbuf = (uint32_t*)lds_raw;                                // [2][w + 2]
        uint32_t* const carry = rowbuf + 2 * (w + 2);                               // [2][nch + 1][2]: level and owner that enter chunk j from its left, by row parity
        unsigned char* const ring = lds_raw + align16u(4u * (2u * (w + 2) + 4u * (nch + 1)));
        for (uint32_t i = tid; i < 2 * (w + 2) + 4 * (nch + 1); i += (uint32_t)BS) rowbuf[i] = i < 2 * (w + 2) ? kNone : 0u;
        auto row_mis = [=](uint32_t row) -> uint32_t { return (uint32_t)((uintptr_t)(plane + (uint64_t)row * w) & 15u); };
        // the row whose chunk 0 runs at step tau, or kNone at the idle step that follows every 64 rows
        auto row_at = [](uint32_t tau) -> uint32_t { const uint32_t q = tau / 65u, m = tau - q * 65u; return m == 64u ? kNone : q * 64u + m; };
        // lane's 16 bytes of the row, from the 16-byte boundary below its first cell, straight into the row's ring slot (LDS-DMA: no
        // register in between, so nothing makes the compiler wait; completion is counted by hand below).  Always one request per
        // call -- a call without a row re-reads the last row into a slot of its own -- so that "at most kAhead requests outstanding"
        // means "everything older has landed".
        typedef __attribute__((address_space(3))) void lds_void_t;
        typedef const __attribute__((address_space(1))) void glb_void_t;
        auto row_request = [=](uint32_t row) {
            const uint32_t rr = row < h ? row : h - 1u, slot = row < h ? row % RR : RR;
            const uint32_t mis = row_mis(rr);
            if ((uint32_t)lane * 16u < mis + row_bytes)
                __builtin_amdgcn_global_load_lds((glb_void_t*)((const unsigned char*)(plane + (uint64_t)rr * w) - mis + (uint32_t)lane * 16u),
                                                 (lds_void_t*)(ring + slot * kSlot), 16, 0, 0);
        };
        const bool loader = piped && wave == NW - 1;
        if (loader) {
            for (uint32_t tau = 0; tau <= kAhead; tau++) row_request(row_at(tau));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        auto process = [&](uint32_t row, uint32_t j, uint32_t raw, int c) {
            const bool in = (uint32_t)c < w;
            const uint32_t ci = in ? (uint32_t)c : 0u;
            const uint32_t* const prev = rowbuf + ((row + 1u) & 1u) * (w + 2);
            uint32_t* const cur = rowbuf + (row & 1u) * (w + 2);
            const uint32_t XW = prev[ci], XN = prev[ci + 1], XE = prev[ci + 2];
            uint32_t* const cr = carry + (row & 1u) * 2u * (nch + 1);          // (chunk j - 1 of the NEXT row writes its hand-over in this very step)
            const uint32_t carry_v = cr[2 * j], carry_l = cr[2 * j + 1];
            const uint32_t v = in ? (raw > bgmin ? raw : bgmin) : 0u;
            const uint32_t p = row * w + (uint32_t)c, V20 = v << 20;
            // (X' - V20 is the owner (< 2^20) when the predecessor has this level, something >= 2^20 otherwise)
            uint32_t lab = v != 0 ? min(min(p, XN - V20), min(XW - V20, XE - V20)) : p;
            // W chain: segmented prefix-min over runs of equal level; the run index travels in bits 20.. so a plain prefix-min serves
            const uint32_t vl = lane_minus1(v, carry_v);
            const bool start = v == 0 || vl != v;
            if (lane == 0 && !start) lab = min(lab, carry_l);
            const unsigned long long smk = __builtin_amdgcn_ballot_w64(start);
            const uint32_t ri = __builtin_amdgcn_mbcnt_hi((uint32_t)(smk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)smk, 0u)) + (start ? 1u : 0u);
            lab = wave_scan_min_u32(((64u - ri) << 20) | (lab & kLenMask)) & kLenMask;
            const bool zp = in && v != 0;
            if (in) cur[ci + 1] = zp ? (V20 | lab) : kNone;
            if (lane == 63) { cr[2 * (j + 1)] = v; cr[2 * (j + 1) + 1] = lab; }
            // the owners go to the label plane (one coalesced store; the strips count the zones: a global atomic per pixel here --
            // zones of a textured image are a pixel or two -- was 40 % of the sweep's time and the chip's whole L2 atomic rate)
            if (in) labp[p] = zp ? lab : kNone;
        };
        const uint32_t n_steps = (h - 1u) + ((h - 1u) >> 6) + nch;
        // Three loops, one per role, with the same number of barriers each.  (One loop with the roles as branches made the compiler
        // merge their memory state: the loader then waited for every load it had in flight, the sweepers read the ring through
        // flat loads and waited for their atomics at every step.)
        if (loader) {
            for (uint32_t t = 0; t < n_steps; t++) {
                row_request(row_at(t + 1u + kAhead));       // the row that starts at step t + 1 must be in the ring after this step's barrier
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kAhead) : "memory");
                lds_barrier();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (piped) {
            for (uint32_t t = 0; t < n_steps; t++) {
                for (uint32_t j = (uint32_t)wave; j < nch; j += (uint32_t)NWs) {
                    const uint32_t row = t >= j ? row_at(t - j) : kNone;
                    if (row >= h) continue;                 // (kNone included)
                    const int c = (int)(64u * j + (uint32_t)lane) - (int)(row & 63u);
                    if ((int)(64u * j) - (int)(row & 63u) >= (int)w) continue;
                    const plane_t* const rrow = (const plane_t*)(ring + (row % RR) * kSlot + row_mis(row));
                    process(row, j, (uint32_t)rrow[(uint32_t)c < w ? (uint32_t)c : 0u], c);
                }
                lds_barrier();
            }
        } else {
            for (uint32_t t = 0; t < n_steps; t++) {
                for (uint32_t j = (uint32_t)wave; j < nch; j += (uint32_t)NWs) {
                    const uint32_t row = t >= j ? row_at(t - j) : kNone;
                    if (row >= h) continue;
                    const int c = (int)(64u * j + (uint32_t)lane) - (int)(row & 63u);
                    if ((int)(64u * j) - (int)(row & 63u) >= (int)w) continue;
                    process(row, j, (uint32_t)plane[row * w + ((uint32_t)c < w ? (uint32_t)c : 0u)], c);
                }
                lds_barrier();
            }
        }
        return;
    }

    // =================================================================================================================
    // strip role
    // =================================================================================================================
    int* const s_res_g = s_res_all + 2 * grp;
    const uint32_t sb = (blockIdx.x - n_sweep) * A.strip_groups + (uint32_t)grp;
    if (sb >= A.ctr[3]) return;
    const uint2 job = A.map_strip[sb];
    LtexRoi R;
    if (!ltex_roi(A, job.x, R, true)) return;
    const bool do_rlm = (A.mask & NYXHIP_FAM_GLRLM) != 0, do_ngt = (A.mask & NYXHIP_FAM_NGTDM) != 0;
    if (!do_rlm && !do_ngt) return;
    const uint32_t w = R.w, h = R.h;
    const uint32_t rows_full = R.L.rows;
    const uint32_t r0 = job.y * rows_full, r1 = r0 + rows_full < h ? r0 + rows_full : h;
    const uint32_t rows = r1 - r0;
    const uint32_t Lcap = R.ng;
    // staged rows [sa, sz): the strip with a row above and below where the box has them
    const uint32_t sa = r0 ? r0 - 1u : 0u, sz = r1 + 1u < h ? r1 + 1u : h;
    const plane_t* const plane = (const plane_t*)(R.base + R.L.plane);
    const plane_t* const src = plane + (uint64_t)sa * w;
    const uint32_t mis = (uint32_t)((uintptr_t)src & 15u);  // the block starts anywhere inside the plane: it is staged at the same misalignment
    const uint32_t st_bytes = (sz - sa) * w * (uint32_t)sizeof(plane_t);
    // ---- LDS: level map | strip | NGTDM accumulators | short-run table
    uint32_t o = 0;
    uint16_t* const s_lvlmap = (uint16_t*)(lds_g + o); o = align16u(o + 2u * (Lcap + 2));
    unsigned char* const s_stage = lds_g + o; o = align16u(o + mis + st_bytes + 16u);
    const uint32_t ng1 = Lcap + 1;                          // NGTDM rows: up to Ng + 1 (IBSI: row = level, 0 .. max)
    const bool ngt_lds = do_ngt && ng1 <= 1024;
    const uint32_t ngt_rep = ltex_ngt_rep(ng1), ngt_stride = ltex_ngt_stride(ng1), ngt_words = ngt_stride / 4u;
    unsigned long long* const s_S = (unsigned long long*)(lds_g + o); if (ngt_lds) o = align16u(o + ngt_rep * ngt_stride);
    uint32_t* const s_N = (uint32_t*)(s_S + ng1 + 2);
    const bool rlm_lds = do_rlm && Lcap <= 128;
    uint32_t* const s_short = (uint32_t*)(lds_g + o); if (rlm_lds) o = align16u(o + 4u * ltex_rl_rep(Lcap) * ltex_rl_words(Lcap));
    if (o > A.lds_group_bytes) {                            // (sized by the host from the class bounds: cannot happen)
        if (tid == 0) atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        return;
    }
    {
        // the rows are one contiguous block of the plane: 16-byte loads (reads up to 15 bytes around it: inside the ROI's padded block)
        const uint4* const src16 = (const uint4*)((const unsigned char*)src - mis);
        uint4* const dst16 = (uint4*)s_stage;
        const uint32_t nvec = (mis + st_bytes + 15u) / 16u;
        for (uint32_t i = tid; i < nvec; i += (uint32_t)BS) {
            uint4 q = src16[i];
            if (bgmin) { q.x = zero_to_one<P16>(q.x); q.y = zero_to_one<P16>(q.y); q.z = zero_to_one<P16>(q.z); q.w = zero_to_one<P16>(q.w); }
            dst16[i] = q;
        }
    }
    ltex_levels(R.base + R.L.flags, Lcap, greyInfo, s_lvlmap, (uint32_t*)nullptr, s_res_g, tid);
    if (ngt_lds) for (uint32_t i = tid; i < ngt_rep * ngt_words; i += (uint32_t)BS) ((uint32_t*)s_S)[i] = 0;
    if (rlm_lds) for (uint32_t i = tid; i < ltex_rl_rep(Lcap) * ltex_rl_words(Lcap); i += (uint32_t)BS) s_short[i] = 0;
    __syncthreads();
    const int Ng = s_res_g[0], Nuniq = s_res_g[1];
    const plane_t* const strip = (const plane_t*)(s_stage + mis);
    // cell of box row r, column c; 0 outside the box (the callers test the column, a missing halo row is handled per row)
    auto cell = [=](uint32_t r, uint32_t c) -> uint32_t { return (uint32_t)strip[(r - sa) * w + c]; };

    // ---- NGTDM (ngtdm.cpp:83-183): 62 centre columns per wave-chunk between two halo lanes; a level travels with a "present" flag in
    // bit 24, so one sum over the eight neighbours yields their level sum and their number (roi_texture.hip: ngtdm_rows)
    const int NgT = greyInfo == 0 ? (Nuniq ? Ng + 1 : 0) : Nuniq;
    unsigned long long* const g_S = (unsigned long long*)(R.base + R.L.ngt);
    uint32_t* const g_N = (uint32_t*)(g_S + R.ng + 2);
    if (do_ngt && NgT >= 2) {
        const bool sum32 = ngt_lds && (unsigned long long)rows * w * 840ull * Lcap < (1ull << 32);
        const uint32_t rep_off = mul24((uint32_t)lane & (ngt_rep - 1u), ngt_words);
        uint32_t* const r_N = ngt_lds ? s_N + rep_off : g_N;
        unsigned long long* const r_S = ngt_lds ? s_S + (rep_off >> 1) : g_S;
        const uint32_t ncs = (w + 61u) / 62u;
        const uint32_t nrb = ncs >= (uint32_t)NW ? 1u : (uint32_t)NW / ncs;
        const uint32_t rows_blk = (rows + nrb - 1) / nrb;
        for (uint32_t t = (uint32_t)wave; t < ncs * nrb; t += (uint32_t)NW) {
            const uint32_t cs = t % ncs, rbi = t / ncs;
            const uint32_t ra = r0 + rbi * rows_blk, rz = ra + rows_blk < r1 ? ra + rows_blk : r1;
            const int c = (int)(cs * 62u) - 1 + lane;
            const bool in_col = c >= 0 && c < (int)w;
            const bool centre = in_col && lane >= 1 && lane <= 62;
            auto code_at = [&](int r) -> uint32_t {
                const uint32_t v = (in_col && r >= (int)sa && r < (int)sz) ? cell((uint32_t)r, (uint32_t)c) : 0u;
                return v | (min(v, 1u) << 24);
            };
            if (ra >= rz) continue;
            uint32_t prv = code_at((int)ra - 1), cur = code_at((int)ra);
            for (uint32_t r = ra; r < rz; r++) {
                const uint32_t nxt = code_at((int)r + 1);
                const uint32_t col3 = prv + cur + nxt;
                const uint32_t tot = lane_minus1(col3, 0u) + lane_plus1(col3, 0u) + (prv + nxt);
                if (centre && cur != 0 && tot >= (1u << 24)) {
                    const uint32_t lvl = cur & 0xFFFFFFu, sum = tot & 0xFFFFFFu, nd = tot >> 24;
                    const uint32_t rr = greyInfo == 0 ? lvl : (uint32_t)s_lvlmap[lvl] - 1u;
                    const uint32_t q = (uint32_t)(840.0f * __builtin_amdgcn_rcpf((float)nd) + 0.5f);   // 840 / nd, exact: 840 = lcm(1 .. 8)
                    const uint32_t a = mul24(lvl, 840u), b2 = mul24(sum, q);
                    const uint32_t d = a > b2 ? a - b2 : b2 - a;                    // |840 i - sum * (840 / nd)|
                    atomicAdd(&r_N[rr], 1u);
                    if (sum32) atomicAdd((uint32_t*)&r_S[rr], d);
                    else atomicAdd(&r_S[rr], (unsigned long long)d);
                }
                prv = cur; cur = nxt;
            }
        }
    }

    // ---- GLRLM (glrlm.cpp:111-195) ------------------------------------------------------------------------------------------------
    if (do_rlm && R.vmin != R.vmax && Ng >= 1) {
        const uint32_t Nr = R.side;
        uint32_t* const gP = (uint32_t*)(R.base + R.L.rlm);
        const uint32_t slot_words = R.L.slot_words;
        uint32_t* const my_short = s_short + ((uint32_t)lane & (ltex_rl_rep(Lcap) - 1u)) * ltex_rl_words(Lcap);
        auto count_run = [&](uint32_t dir, uint32_t v, uint32_t len) {
            const uint32_t m = (uint32_t)s_lvlmap[v] - 1u;
            if (rlm_lds && len <= kLtexRlmLds) atomicAdd(&my_short[ltex_rl_index(Lcap, dir, m, len)], 1u);
            else atomicAdd(&gP[dir * slot_words + m * Nr + (len - 1u)], 1u);
        };
        // records of this strip: [K][2][3][w], level << 20 | length; 0 = none (the workspace was zeroed)
        uint32_t* const rec_top = (uint32_t*)(R.base + R.L.rec) + (uint64_t)job.y * 6u * w;
        uint32_t* const rec_bot = rec_top + 3u * w;
        // tasks: the strip's rows (0 degrees), then 64-line chunks of the three other directions
        const uint32_t nl_diag = w + rows - 1u;
        const uint32_t ch_s = (w + 63u) / 64u, ch_d = (nl_diag + 63u) / 64u;
        const uint32_t n_tasks = rows + ch_s + 2u * ch_d;
        for (uint32_t t = (uint32_t)wave; t < n_tasks; t += (uint32_t)NW) {
            if (t < rows) {
                // 0 degrees, one row: a run starts where a cell differs from its left neighbour; a start lane reads its run's length off
                // the ballot of starts; the run that is open at the end of a chunk is carried (wave-uniform) into the next one
                const uint32_t r = r0 + t;
                uint32_t cv = 0, cl = 0, last_v = 0;
                for (uint32_t c0 = 0; c0 < w; c0 += 64) {
                    const uint32_t c = c0 + (uint32_t)lane;
                    const uint32_t v = c < w ? cell(r, c) : 0u;
                    const uint32_t prev = lane_minus1(v, last_v);
                    const bool start = v != prev || c == 0;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(start);
                    if (m == 0) { cl += 64u; last_v = readlane63(v); continue; }
                    const uint32_t first = (uint32_t)__builtin_ctzll(m);
                    cl += first;
                    if (lane == 0 && cv != 0) count_run(0u, cv, cl);
                    if (start && v != 0) {
                        const unsigned long long above = lane < 63 ? (m >> (lane + 1)) : 0ull;
                        if (above) count_run(0u, v, (uint32_t)__builtin_ctzll(above) + 1u);
                    }
                    const uint32_t L = 63u - (uint32_t)__builtin_clzll(m);               // the chunk's last start: its run stays open
                    cv = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)L);
                    cl = 64u - L;
                    last_v = readlane63(v);
                }
                if (lane == 0 && cv != 0) count_run(0u, cv, cl);
            } else {
                // 45 / 90 / 135 degrees: lane = line, identified by its column in the strip's first row (lines that enter through a
                // side of the box have none: their first cell is a run start)
                uint32_t u = t - rows, dir;
                int dx;
                if (u < ch_d) { dir = 1u; dx = 1; }
                else if (u < ch_d + ch_s) { u -= ch_d; dir = 2u; dx = 0; }
                else { u -= ch_d + ch_s; dir = 3u; dx = -1; }
                const uint32_t li = u * 64u + (uint32_t)lane;
                const uint32_t nl = dx ? nl_diag : w;
                if (li >= nl) continue;
                const int c_top = (int)li - (dx == 1 ? (int)rows - 1 : 0);
                uint32_t rv = 0, rl = 0;
                bool top = false;
                int c = c_top;
                for (uint32_t tt = 0; tt < rows; tt++, c += dx) {
                    const uint32_t v = (uint32_t)c < w ? cell(r0 + tt, (uint32_t)c) : 0u;
                    if (v == rv) { rl += v != 0 ? 1u : 0u; continue; }
                    if (rv != 0) {
                        if (top) rec_top[(dir - 1u) * w + (uint32_t)c_top] = (rv << 20) | rl;
                        else count_run(dir, rv, rl);
                    }
                    rv = v; rl = v != 0 ? 1u : 0u; top = tt == 0;
                }
                if (rv != 0) {                               // alive in the strip's last row: open at the bottom
                    if (top) rec_top[(dir - 1u) * w + (uint32_t)c_top] = (rv << 20) | rl;         // (length == rows: it spans the strip)
                    else rec_bot[(dir - 1u) * w + (uint32_t)(c - dx)] = (rv << 20) | rl;
                }
            }
        }
    }
    __syncthreads();
    // ---- flush the LDS tables (integers: any order)
    if (ngt_lds && do_ngt && NgT >= 2) {
        for (int i = tid; i < NgT; i += BS) {
            unsigned long long sS = 0; uint32_t sN = 0;
            for (uint32_t r = 0; r < ngt_rep; r++) { sS += s_S[(size_t)r * (ngt_words >> 1) + i]; sN += s_N[r * ngt_words + i]; }
            if (sN) { atomicAdd(&g_N[i], sN); atomicAdd(&g_S[i], sS); }
        }
    }
    if (rlm_lds) ltex_rl_flush(s_short, Lcap, (uint32_t*)(R.base + R.L.rlm), R.L.slot_words, R.side, tid, BS, ltex_rl_rep(Lcap));
}

// ---- after the strips: one workgroup per strip ------------------------------------------------------------------------------------
// GLRLM: the recorded runs are pieces of runs that cross strips.  A piece is a HEAD when nothing of its level ends above it on its line
// -- every bottom record (its run started inside the strip), and a top record whose predecessor cell holds another level or lies
// outside the box.  The thread that finds a head follows its continuation down the strips (top records of its level that span their
// strip, then one that does not) and counts the run once.
// GLSZM: zones -> (level, size) multiplicities (direct table for sizes <= 32, a list for the few larger ones), zones, largest zone.
template <bool P16>
__global__ __launch_bounds__(256) void ltex_post_kernel(const LtexArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_res[2];
    __shared__ uint32_t s_nz[4], s_mx[4];
    constexpr int BS = 256;
    constexpr uint32_t kPostRep = 2;                       // replicas of the run table: the joined runs are a few hundred per strip -- LDS per workgroup matters more
    if (blockIdx.x >= A.ctr[3]) return;
    const uint2 job = A.map_strip[blockIdx.x];
    LtexRoi R;
    if (!ltex_roi(A, job.x, R, true)) return;
    if (R.vmin == R.vmax) return;                           // blank: neither runs nor zones are asked for (glrlm.cpp:29-52, glszm.cpp:61-65)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    using plane_t = typename std::conditional<P16, uint16_t, uint8_t>::type;
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    const uint32_t bgmin = greyInfo > 0 ? 1u : 0u;
    const uint32_t Lcap = R.ng, S = R.L.S, w = R.w, h = R.h, K = R.L.K, rows_full = R.L.rows;
    const bool do_rlm = (A.mask & NYXHIP_FAM_GLRLM) != 0, do_szm = (A.mask & NYXHIP_FAM_GLSZM) != 0;
    uint32_t o = 0;
    uint16_t* const s_lvlmap = (uint16_t*)(lds_raw + o); o = align16u(o + 2u * (Lcap + 2));
    uint32_t* const s_small = (uint32_t*)(lds_raw + o); if (do_szm) o = align16u(o + 4u * Lcap * S);
    const bool rlm_lds = do_rlm && Lcap <= 128;
    uint32_t* const s_short = (uint32_t*)(lds_raw + o);
    ltex_levels(R.base + R.L.flags, Lcap, greyInfo, s_lvlmap, (uint32_t*)nullptr, s_res, tid);
    if (do_szm) for (uint32_t i = tid; i < Lcap * S; i += BS) s_small[i] = 0;
    if (rlm_lds) for (uint32_t i = tid; i < kPostRep * ltex_rl_words(Lcap); i += BS) s_short[i] = 0;
    __syncthreads();

    if (do_rlm && s_res[0] >= 1) {
        const uint32_t Nr = R.side, slot_words = R.L.slot_words;
        uint32_t* const gP = (uint32_t*)(R.base + R.L.rlm);
        const uint32_t* const rec = (const uint32_t*)(R.base + R.L.rec);
        const uint32_t k = job.y;
        auto rows_of = [=](uint32_t kk) { return min(rows_full, h - kk * rows_full); };
        auto top_of = [=](uint32_t kk, uint32_t dir, uint32_t c) { return rec[((uint64_t)kk * 6u + (dir - 1u)) * w + c]; };
        auto bot_of = [=](uint32_t kk, uint32_t dir, uint32_t c) { return rec[((uint64_t)kk * 6u + 3u + (dir - 1u)) * w + c]; };
        uint32_t* const my_short = s_short + ((uint32_t)lane & (kPostRep - 1u)) * ltex_rl_words(Lcap);
        auto count_rec = [&](uint32_t dir, uint32_t lvl, uint32_t len) {
            const uint32_t m = (uint32_t)s_lvlmap[lvl] - 1u;
            if (rlm_lds && len <= kLtexRlmLds) atomicAdd(&my_short[ltex_rl_index(Lcap, dir, m, len)], 1u);
            else atomicAdd(&gP[dir * slot_words + m * Nr + (len - 1u)], 1u);
        };
        // the run (level lvl, len pixels so far) is alive in the last row of strip kk at column cb: follow it down
        auto follow = [&](uint32_t dir, int dx, uint32_t lvl, uint32_t len, uint32_t kk, uint32_t cb) {
            while (kk + 1u < K) {
                const int nc = (int)cb + dx;
                if ((uint32_t)nc >= w) break;
                const uint32_t T = top_of(kk + 1u, dir, (uint32_t)nc);
                if (T == 0 || (T >> 20) != lvl) break;
                len += T & kLenMask;
                const uint32_t rk = rows_of(kk + 1u);
                if ((T & kLenMask) != rk) break;             // it ends inside that strip
                kk++; cb = (uint32_t)(nc + dx * ((int)rk - 1));
            }
            count_rec(dir, lvl, len);
        };
        const uint32_t rows_k = rows_of(k);
        for (uint32_t i = tid; i < 3u * w; i += BS) {
            const uint32_t dir = 1u + i / w, c = i - (dir - 1u) * w;
            const int dx = dir == 1u ? 1 : dir == 2u ? 0 : -1;
            const uint32_t B = bot_of(k, dir, c);
            if (B != 0) follow(dir, dx, B >> 20, B & kLenMask, k, c);
            const uint32_t T = top_of(k, dir, c);
            if (T == 0) continue;
            bool head = true;
            const int pc = (int)c - dx;
            if (k > 0 && (uint32_t)pc < w) {                 // the run that is alive in the cell above on the line, if any
                uint32_t P = bot_of(k - 1u, dir, (uint32_t)pc);
                if (P == 0) {
                    const uint32_t rp = rows_of(k - 1u);
                    const int ct = pc - dx * ((int)rp - 1);
                    if ((uint32_t)ct < w) {
                        const uint32_t Tp = top_of(k - 1u, dir, (uint32_t)ct);
                        if (Tp != 0 && (Tp & kLenMask) == rp) P = Tp;
                    }
                }
                if (P != 0 && (P >> 20) == (T >> 20)) head = false;
            }
            if (!head) continue;
            if ((T & kLenMask) == rows_k) follow(dir, dx, T >> 20, rows_k, k, (uint32_t)((int)c + dx * ((int)rows_k - 1)));
            else count_rec(dir, T >> 20, T & kLenMask);
        }
        if (rlm_lds) {
            __syncthreads();
            ltex_rl_flush(s_short, Lcap, gP, slot_words, Nr, tid, BS, kPostRep);
        }
    }

    if (do_szm) {
        // Zones: the sweep left every cell's owner in the label plane.  A strip counts the cells of the owners that lie inside it in LDS
        // (16-bit counters: a strip has at most 8192 cells, or one row); cells whose owner lies in an earlier strip add to the owner's
        // global counter.  A zone none of whose cells sits in the strip's last row is complete here and is entered at once; the
        // others -- they may go on below -- add their local count to the global counter and are listed for the finishing kernel.
        const plane_t* const plane = (const plane_t*)(R.base + R.L.plane);
        const uint32_t* const labp = (const uint32_t*)(R.base + R.L.lab);
        uint32_t* const cnt = (uint32_t*)(R.base + R.L.cnt);
        uint32_t* const hdr = (uint32_t*)R.base;
        uint32_t* const big = (uint32_t*)(R.base + R.L.big);
        uint32_t* const openl = (uint32_t*)(R.base + R.L.open);
        const uint32_t p0 = job.y * rows_full * w, p1 = min(p0 + rows_full * w, R.area);
        const bool last_strip = job.y + 1u >= K;
        uint32_t* const s_loc = s_short + (rlm_lds ? kPostRep * ltex_rl_words(Lcap) : 0u);   // [(p1 - p0 + 1) / 2] words of two 16-bit counters
        __syncthreads();                                        // (the run tables above are flushed)
        for (uint32_t i = tid; i < (p1 - p0 + 2u) / 2u; i += BS) s_loc[i] = 0;
        __syncthreads();
        for (uint32_t p = p0 + (uint32_t)tid; p < p1; p += BS) {
            const uint32_t lab = labp[p];
            if (lab == kNone) continue;
            if (lab >= p0) { const uint32_t q = lab - p0; atomicAdd(&s_loc[q >> 1], 1u << (16u * (q & 1u))); }
        }
        // cells of zones owned above the strip: one atomic per string of equal owners among a wave's 64 consecutive cells (the
        // background corners of a box are ONE zone each: cell by cell, its millions of adds would all go to one address)
        for (uint32_t pb = p0 + (uint32_t)(wave * 64); pb < p1; pb += BS) {
            const uint32_t p = pb + (uint32_t)lane;
            const uint32_t lab = p < p1 ? labp[p] : kNone;
            const bool far = lab != kNone && lab < p0;
            const uint32_t ln = lane_plus1(lab, kNone);
            const unsigned long long same = __builtin_amdgcn_ballot_w64(far && lane < 63 && ln == lab);
            if (far && !(lane > 0 && ((same >> (lane - 1)) & 1ull)))
                atomicAdd(&cnt[lab], (uint32_t)__ffsll((long long)~(same >> lane)));
        }
        __syncthreads();
        if (!last_strip)
            for (uint32_t p = p1 - w + (uint32_t)tid; p < p1; p += BS) {
                const uint32_t lab = labp[p];
                if (lab != kNone && lab >= p0) { const uint32_t q = lab - p0; atomicOr(&s_loc[q >> 1], 0x8000u << (16u * (q & 1u))); }
            }
        __syncthreads();
        uint32_t nzone = 0, sz_max = 0;
        for (uint32_t p = p0 + (uint32_t)tid; p < p1; p += BS) {
            const uint32_t q = p - p0;
            const uint32_t c = (s_loc[q >> 1] >> (16u * (q & 1u))) & 0xFFFFu;
            if (c == 0) continue;
            const uint32_t sz = c & 0x7FFFu;
            if (c & 0x8000u) {                                  // may go on below: the finishing kernel enters it
                atomicAdd(&cnt[p], sz);
                // (one cursor add per wave: a thousand zones per ROI adding to the one word serialise in L2)
                const unsigned long long om = __builtin_amdgcn_ballot_w64(true);
                const int leader = (int)__builtin_ctzll(om);
                uint32_t base_slot = 0;
                if (lane == leader) base_slot = atomicAdd(&hdr[LTEX_H_NOPEN], (uint32_t)__builtin_popcountll(om));
                base_slot = (uint32_t)__builtin_amdgcn_readlane((int)base_slot, leader);
                const uint32_t slot = base_slot + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
                if (slot < R.L.open_cap) openl[slot] = p;
                continue;
            }
            nzone++;
            sz_max = sz > sz_max ? sz : sz_max;
            const uint32_t raw = (uint32_t)plane[p];
            const uint32_t rowi = (uint32_t)s_lvlmap[raw > bgmin ? raw : bgmin] - 1u;
            if (sz <= S) atomicAdd(&s_small[rowi * kLtexSmall + (sz - 1u)], 1u);
            else {
                const uint32_t slot = atomicAdd(&hdr[LTEX_H_NBIG], 1u);
                if (slot < R.L.big_cap) big[slot] = (rowi << 20) | sz;
            }
        }
        nzone = wave_sum_t<uint32_t>(nzone);
        sz_max = wave_max_u32(sz_max);
        if (lane == 0) { s_nz[wave] = nzone; s_mx[wave] = sz_max; }
        __syncthreads();
        if (tid == 0) {
            const uint32_t nz = s_nz[0] + s_nz[1] + s_nz[2] + s_nz[3];
            const uint32_t mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
            if (nz) { atomicAdd(&hdr[LTEX_H_NZONE], nz); atomicMax(&hdr[LTEX_H_SZMAX], mx); }
        }
        uint32_t* const g_small = (uint32_t*)(R.base + R.L.small);
        for (uint32_t i = tid; i < Lcap * S; i += BS) {
            const uint32_t v = s_small[i];
            if (v) atomicAdd(&g_small[i], v);
        }
    }
}

// ---- finish: one workgroup per ROI ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ltex_finish_kernel(const LtexArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_res[2];
    __shared__ double s_red[kWaves * 8];
    __shared__ double s_f[4 * 16];
    __shared__ uint32_t s_flag;
    constexpr int BS = 256;
    if (blockIdx.x >= A.n_list) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    LtexRoi R;
    const bool served = ltex_roi(A, blockIdx.x, R, true);
    const int end_rlm = (A.mask & NYXHIP_FAM_GLRLM) ? 80 : 0, end_szm = end_rlm + ((A.mask & NYXHIP_FAM_GLSZM) ? 16 : 0);
    auto gcol = [=](int c) { return c + (c >= end_rlm ? A.gap_after_glrlm : 0) + (c >= end_szm ? A.gap_after_glszm : 0); };
    double* const out_row = A.out + R.roi * A.ld + A.col0;
    if (!served) {
        if (R.n == 0)                                       // an empty ROI: the columns of roi_texture.hip's early exit
            for (int c = tid; c < A.n_cols; c += BS) out_row[gcol(c)] = __longlong_as_double(0x7ff8000000000000LL);
        return;                                             // (anybody else: the one-workgroup launch writes the row)
    }
    const bool do_rlm = (A.mask & NYXHIP_FAM_GLRLM) != 0, do_szm = (A.mask & NYXHIP_FAM_GLSZM) != 0, do_ngt = (A.mask & NYXHIP_FAM_NGTDM) != 0;
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    const uint32_t area = R.area, Lcap = R.ng;
    const uint32_t* const hdr = (const uint32_t*)R.base;
    // ---- LDS: output row | level map | levels | level squares | a region the families use one after the other
    uint32_t o = 0;
    double* const s_out = (double*)(lds_raw + o); o = align16u(o + 8u * (uint32_t)A.n_cols);
    uint16_t* const s_lvlmap = (uint16_t*)(lds_raw + o); o = align16u(o + 2u * (Lcap + 2));
    uint32_t* const s_lv = (uint32_t*)(lds_raw + o); o = align16u(o + 4u * (Lcap + 2));
    double* const s_lvf_mem = (double*)(lds_raw + o); if (Lcap <= 256) o = align16u(o + 16u * (Lcap + 2));
    unsigned char* const s_work = lds_raw + o;
    const uint32_t work_bytes = A.lds_fin_bytes > o ? A.lds_fin_bytes - o : 0u;
    ltex_levels(R.base + R.L.flags, Lcap, greyInfo, s_lvlmap, s_lv, s_res, tid);
    for (int c = tid; c < A.n_cols; c += BS) s_out[c] = 0.0;
    __syncthreads();
    const int Ng = s_res[0], Nuniq = s_res[1];
    const double* const s_lvf = Lcap <= 256 ? s_lvf_mem : nullptr;
    if (s_lvf) {
        for (int i = tid; i < Ng; i += BS) {
            const double in2d = (double)(uint32_t)(s_lv[i] * s_lv[i]);
            s_lvf_mem[2 * i] = in2d; s_lvf_mem[2 * i + 1] = frcp(in2d);
        }
    }
    __syncthreads();
    const bool blank = R.vmin == R.vmax;
    int col = 0;

    // =====================================================================================
    // GLRLM: the features of the four matrices (a wave each)
    // =====================================================================================
    if (do_rlm) {
        double* const oo = s_out + col;
        col += 80;
        const uint32_t Nr = R.side, slot_words = R.L.slot_words;
        uint32_t* const gP = (uint32_t*)(R.base + R.L.rlm);
        if (blank) {                                         // glrlm.cpp:29-52
            for (int c = tid; c < 80; c += BS) oo[c] = A.soft_nan;
        } else if (Ng < 1) {
            for (int c = tid; c < 80; c += BS) oo[c] = 0.0;
        } else {
            // the matrices from LDS when the four fit the work region, else in place
            const uint32_t Np = hdr[LTEX_H_NP_ORIG];
            if (16ull * slot_words <= work_bytes) {
                uint32_t* const s_mat = (uint32_t*)s_work;
                for (uint32_t i = tid; i < 4u * slot_words; i += BS) s_mat[i] = gP[i];
                __syncthreads();
                uint32_t* const P = s_mat + (uint32_t)wave * slot_words;
                glrlm_features_wave<false>(P, Ng, (int)Nr, s_lv, s_lvf, P + Ng * Nr, P + Ng * Nr + Ng, Np, s_f + wave * 16, s_red + wave * 8, lane);
            } else {
                uint32_t* const P = gP + (uint32_t)wave * slot_words;
                glrlm_features_wave<true>(P, Ng, (int)Nr, s_lv, s_lvf, P + Ng * Nr, P + Ng * Nr + Ng, Np, s_f + wave * 16, s_red + wave * 8, lane);
            }
            __syncthreads();
            for (int c = tid; c < 64; c += BS) {             // feature-major, angle-minor
                const int k = c >> 2, a = c & 3;
                oo[c] = s_f[a * 16 + k];
            }
            for (int k = tid; k < 16; k += BS) {             // calc_ave :903-910 (std::reduce of 4)
                const double v = 0.0 + ((s_f[0 * 16 + k] + s_f[1 * 16 + k]) + (s_f[2 * 16 + k] + s_f[3 * 16 + k]));
                oo[64 + k] = v / 4.0;
            }
        }
        __syncthreads();
    }

    // =====================================================================================
    // GLSZM (glszm.cpp:212-395 over the non-zero cells of P(i,j))
    // =====================================================================================
    if (do_szm) {
        double* const oo = s_out + col;
        col += 16;
        if (blank) {                                         // glszm.cpp:61-65
            for (int c = tid; c < 16; c += BS) oo[c] = A.soft_nan;
        } else {
            const uint32_t hcap = R.L.hcap, S = R.L.S;
            uint32_t* const hkey = (uint32_t*)(R.base + R.L.hash);
            uint32_t* const hval = hkey + hcap;
            uint32_t* const g_small = (uint32_t*)(R.base + R.L.small);
            uint32_t* const sj = (uint32_t*)(R.base + R.L.cnt);        // zones per size (all zero again once the open zones are read off below)
            const uint32_t* const big = (const uint32_t*)(R.base + R.L.big);
            uint32_t* const s_si = (uint32_t*)s_work;        // [Ng] zones per level
            // the zones that reached their strip's last row: their cells were counted by several strips into the owner's counter
            uint32_t* const bigw = (uint32_t*)(R.base + R.L.big);
            {
                const uint32_t n_open = min(hdr[LTEX_H_NOPEN], R.L.open_cap);
                const uint32_t* const openl = (const uint32_t*)(R.base + R.L.open);
                const bool P16r = A.plane16 != 0;
                const unsigned char* const pl = R.base + R.L.plane;
                const uint32_t bgmin = greyInfo > 0 ? 1u : 0u;
                uint32_t nz = 0, mx = 0;
                for (uint32_t i = tid; i < n_open; i += BS) {
                    const uint32_t p = openl[i];
                    const uint32_t sz = sj[p];
                    sj[p] = 0;                               // (the table serves as "zones per size" below)
                    nz++;
                    mx = sz > mx ? sz : mx;
                    const uint32_t raw = P16r ? (uint32_t)((const uint16_t*)pl)[p] : (uint32_t)pl[p];
                    const uint32_t rowi = (uint32_t)s_lvlmap[raw > bgmin ? raw : bgmin] - 1u;
                    if (sz <= S) atomicAdd(&g_small[rowi * kLtexSmall + (sz - 1u)], 1u);
                    else {
                        const uint32_t slot = atomicAdd((uint32_t*)&hdr[LTEX_H_NBIG], 1u);
                        if (slot < R.L.big_cap) bigw[slot] = (rowi << 20) | sz;
                    }
                }
                nz = wave_sum_t<uint32_t>(nz);
                mx = wave_max_u32(mx);
                if (lane == 0 && nz) { atomicAdd((uint32_t*)&hdr[LTEX_H_NZONE], nz); atomicMax((uint32_t*)&hdr[LTEX_H_SZMAX], mx); }
                if (tid == 0 && hdr[LTEX_H_NOPEN] > R.L.open_cap) atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);   // (sized for every boundary cell: cannot happen)
                blk_sync<true>();
            }
            const uint32_t n_bigr = __hip_atomic_load(&hdr[LTEX_H_NBIG], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t n_big = min(n_bigr, R.L.big_cap);
            if (tid == 0) s_flag = n_bigr > R.L.big_cap ? 1u : 0u;
            for (int i = tid; i < Ng; i += BS) s_si[i] = 0;
            // the larger zones: an ORDERED linear-probing hash (a key is displaced only by a larger one), so the layout -- hence the
            // order of the floating-point sums over it -- is a function of the key set (roi_texture.hip)
            for (uint32_t i = tid; i < n_big; i += BS) {
                uint32_t k = big[i];
                uint32_t hsl = (k * 2654435761u) & (hcap - 1);
                for (;;) {
                    const uint32_t old = atomicMax(&hkey[hsl], k);
                    if (old == k || old == 0) break;
                    if (old < k) k = old;
                    hsl = (hsl + 1) & (hcap - 1);
                }
            }
            blk_sync<true>();
            for (uint32_t i = tid; i < n_big; i += BS) {
                const uint32_t key = big[i];
                uint32_t hsl = (key * 2654435761u) & (hcap - 1);
                while (__hip_atomic_load(&hkey[hsl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != key) hsl = (hsl + 1) & (hcap - 1);
                atomicAdd(&hval[hsl], 1u);
            }
            blk_sync<true>();
            if (s_flag) {                                    // (the list is sized for every zone above S pixels: cannot happen)
                if (tid == 0) atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
            }
            const uint32_t n_cells = (n_big ? hcap : 0u) + (uint32_t)Ng * S;
            const uint32_t c_off = n_big ? hcap : 0u;
            auto cell = [=](uint32_t i, uint32_t& key, uint32_t& val) {
                if (i < c_off) { key = hkey[i]; val = hval[i]; }
                else {
                    const uint32_t idx = i - c_off;
                    val = g_small[idx];
                    key = val ? ((idx / kLtexSmall) << 20) | ((idx % kLtexSmall) + 1u) : 0u;
                }
            };
            const double sum_p = (double)__hip_atomic_load(&hdr[LTEX_H_NZONE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t sz_max = __hip_atomic_load(&hdr[LTEX_H_SZMAX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t i = tid; i < n_cells; i += BS) {
                uint32_t key, val;
                cell(i, key, val);
                if (key != 0) {
                    atomicAdd(&sj[key & kLenMask], val);
                    atomicAdd(&s_si[key >> 20], val);
                }
            }
            blk_sync<true>();
            if (sum_p == 0) {                                // glszm.cpp:229-233
                for (int c = tid; c < 16; c += BS) oo[c] = A.soft_nan;
            } else {
                const double inv_p = frcp(sum_p);
                double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const double ztab = plog_tex((double)lane * inv_p);
                for (uint32_t i0 = 0; i0 < n_cells; i0 += BS) {
                    const uint32_t i = i0 + (uint32_t)tid;
                    uint32_t key = 0, val = 0;
                    if (i < n_cells) cell(i, key, val);
                    double ze = __shfl(ztab, (int)(val & 63u), 64);
                    if (__builtin_amdgcn_ballot_w64(val >= 64u))
                        ze = val >= 64u ? plog_tex((double)val * inv_p) : ze;
                    if (key == 0) continue;
                    const double p = (double)val;
                    const double inten = (double)s_lv[key >> 20], jd = (double)(key & kLenMask);
                    double i2, ri2;
                    if (s_lvf) { i2 = s_lvf[2 * (key >> 20)]; ri2 = s_lvf[2 * (key >> 20) + 1]; }
                    else { i2 = inten * inten; ri2 = frcp(i2); }
                    const double j2 = jd * jd, rj2 = frcp(j2);
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
                {
                    const double tt = wave_transpose_sum8(acc, lane);
                    if ((lane & 7) == 0) s_red[wave * 8 + (lane >> 3)] = tt;
                }
                __syncthreads();
                const double mu_ZV = ((s_red[5] + s_red[8 + 5]) + s_red[16 + 5]) + s_red[24 + 5];
                const double mu_GLV = ((s_red[6] + s_red[8 + 6]) + s_red[16 + 6]) + s_red[24 + 6];
                if (tid == 0) {
#pragma unroll
                    for (int k = 0; k < 5; k++) acc[k] = ((s_red[k] + s_red[8 + k]) + s_red[16 + k]) + s_red[24 + k];
                    oo[Z_ZE] = -acc[4];
                    oo[Z_SALGLE] = acc[3] * inv_p;
                    oo[Z_SAHGLE] = acc[2] * inv_p;
                    oo[Z_LALGLE] = acc[1] * inv_p;
                    oo[Z_LAHGLE] = acc[0] * inv_p;
                }
                __syncthreads();
                double b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (uint32_t i = tid; i < n_cells; i += BS) {
                    uint32_t key, val;
                    cell(i, key, val);
                    if (key == 0) continue;
                    const double p = (double)val * inv_p;
                    const double dg = (double)s_lv[key >> 20] - mu_GLV, dz = (double)(key & kLenMask) - mu_ZV;
                    b[0] += p * (dg * dg);                   // calc_GLV :497-510
                    b[1] += p * (dz * dz);                   // calc_ZV :512-524
                }
                // zones per size.  Ns = the box's cell count in the reference (glszm.cpp:212): j * j is an int product there, which is
                // exactly 0 at the multiples of 65536, where an EMPTY column contributes 0.0 / 0 = NaN to SAE -- boxes of 65536
                // cells and more have such a column unless a zone of that very size exists (then it is sj / 0 = inf).
                for (uint32_t j = 1 + tid; j <= sz_max; j += BS) {
                    const uint32_t sji = sj[j];
                    if (sji == 0 && (j & 0xFFFFu) != 0) continue;
                    const double sjd = (double)sji;
                    if (j < 32768u) {
                        const double jj = (double)mul24(j, j);
                        b[2] += sjd * frcp(jj);              // calc_SAE :419-428
                        b[3] += sjd * jj;                    // calc_LAE :430-439
                    } else {
                        const int jj = (int)(j * j);
                        b[2] += sjd / (double)jj;
                        b[3] += sjd * (double)jj;
                    }
                    b[4] += sjd * sjd;                       // calc_SZN :464-474
                }
                if (tid == 0)
                    for (uint32_t j = 65536u; j <= area; j += 65536u)
                        if (j > sz_max) { const double zero = 0.0; b[2] += zero / (double)(int)(j * j); }
                for (int i = tid; i < Ng; i += BS) {
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
                __syncthreads();
                if (tid == 0) {
                    for (int k = 0; k < 8; k++) b[k] = ((s_red[k] + s_red[8 + k]) + s_red[16 + k]) + s_red[24 + k];
                    const double inv_p2 = inv_p * inv_p;
                    const uint32_t np_bin = greyInfo > 0 ? area : hdr[LTEX_H_NP_BIN];      // non-zero binned pixels (glszm.cpp:193-199)
                    oo[Z_SAE] = b[2] * inv_p;
                    oo[Z_LAE] = b[3] * inv_p;
                    oo[Z_GLN] = b[5] * inv_p;
                    oo[Z_GLNN] = b[5] * inv_p2;
                    oo[Z_SZN] = b[4] * inv_p;
                    oo[Z_SZNN] = b[4] * inv_p2;
                    oo[Z_ZP] = fdiv(sum_p, (double)(int)np_bin);               // calc_ZP :491-495
                    oo[Z_GLV] = b[0];
                    oo[Z_ZV] = b[1];
                    oo[Z_LGLZE] = b[6] * inv_p;
                    oo[Z_HGLZE] = b[7] * inv_p;
                }
            }
        }
        __syncthreads();
    }

    // =====================================================================================
    // NGTDM (ngtdm.cpp:228-345)
    // =====================================================================================
    if (do_ngt) {
        double* const oo = s_out + col;
        col += 5;
        const int NgT = greyInfo == 0 ? (Nuniq ? Ng + 1 : 0) : Nuniq;
        if (NgT < 2) {                                        // ngtdm.cpp:70-78
            for (int c = tid; c < 5; c += BS) oo[c] = A.soft_nan;
        } else {
            const unsigned long long* const g_S = (const unsigned long long*)(R.base + R.L.ngt);
            const uint32_t* const g_N = (const uint32_t*)(g_S + R.ng + 2);
            double* const s_P = (double*)s_work;              // [NgT]
            double* const s_Sd = s_P + (Lcap + 2);            // [NgT]
            uint32_t nvc_part = 0;
            for (int i = tid; i < NgT; i += BS) nvc_part += g_N[i];
            nvc_part = wave_sum_t<uint32_t>(nvc_part);
            __syncthreads();
            if (lane == 0) s_red[wave * 8] = (double)nvc_part;
            __syncthreads();
            const double Nvc = ((s_red[0] + s_red[8]) + s_red[16]) + s_red[24];
            {
                const double inv_nvc = frcp(Nvc);
                for (int i = tid; i < NgT; i += BS) {
                    s_P[i] = (double)g_N[i] * inv_nvc;
                    s_Sd[i] = (double)g_S[i] * (1.0 / 840.0);
                }
            }
            __syncthreads();
            if (wave == 0) {
                auto Iof = [=](int i) -> double { return greyInfo == 0 ? (double)i : (double)s_lv[i]; };
                double t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};                        // ps, ssum, contrast, busyness, complexity, strength
                for (int i = lane; i < NgT; i += 64) { t8[0] += s_P[i] * s_Sd[i]; t8[1] += s_Sd[i]; }
                const bool small_ng = NgT <= 256;
                const float inv_ng = 1.0f / (float)NgT;
                for (int e = lane; e < NgT * NgT; e += 64) {
                    const int i = small_ng ? (int)(((float)e + 0.5f) * inv_ng) : e / NgT;
                    const int j = e - i * NgT;
                    const double pi_ = s_P[i], pj = s_P[j], iv = Iof(i), jv = Iof(j);
                    const double d = iv - jv;
                    t8[2] += pi_ * pj * d * d;                                  // calc_Contrast :245-247
                    if (pi_ != 0 && pj != 0) {
                        t8[3] += fabs(pi_ * iv - pj * jv);                       // calc_Busyness :280-283
                        t8[4] += fdiv(fabs(d) * (pi_ * s_Sd[i] + pj * s_Sd[j]), pi_ + pj); // calc_Complexity :305
                        t8[5] += (pi_ + pj) * d * d;                             // calc_Strength :326
                    }
                }
                {
                    const double tt = wave_transpose_sum8(t8, lane);
                    if ((lane & 7) == 0) s_red[lane >> 3] = tt;
                }
                wav_sync<false>();
                const double ps = s_red[0], ssum = s_red[1], c_sum = s_red[2], b_sum = s_red[3], x_sum = s_red[4], s_sum = s_red[5];
                if (lane == 0) {
                    const int Ngp = Nuniq;
                    const int Ngp_p2 = Ngp > 1 ? Ngp * (Ngp - 1) : Ngp;
                    oo[0] = 1.0 / ps;                                            // calc_Coarseness :228-236
                    oo[1] = (c_sum / (double)Ngp_p2) * (ssum / Nvc);             // calc_Contrast :251-261
                    oo[2] = Ngp == 1 ? 0.0 : (b_sum == 0 ? 0.0 : ps / b_sum);    // calc_Busyness :266-291
                    oo[3] = x_sum / (double)(int)Nvc;                            // calc_Complexity :310 (Nvp)
                    oo[4] = s_sum / ssum;                                        // calc_Strength :331-335
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    for (int c = tid; c < A.n_cols; c += BS)
        out_row[gcol(c)] = s_out[c];
}

} // namespace

// Launches the kernels of one group.  The caller has zeroed a.ws[0 .. ws_bytes) and a.ctr on the stream.
int launch_large_texture(const LtexArgs& a, void* stream)
{
    if (a.n_list == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    static DeviceOnce optin;
    if (int orc = optin.run([]() -> int {
            const struct { const void* f; int bytes; } k[] = {
                {(const void*)ltex_load_kernel<false>, 64 * 1024}, {(const void*)ltex_load_kernel<true>, 64 * 1024},
                {(const void*)ltex_strip_kernel<false>, 144 * 1024}, {(const void*)ltex_strip_kernel<true>, 144 * 1024},
                {(const void*)ltex_post_kernel<false>, 128 * 1024}, {(const void*)ltex_post_kernel<true>, 128 * 1024},
                {(const void*)ltex_finish_kernel, 144 * 1024}};
            for (const auto& e : k)
                if (hipError_t rc = hipFuncSetAttribute(e.f, hipFuncAttributeMaxDynamicSharedMemorySize, e.bytes); rc != hipSuccess) return (int)rc;
            return 0;
        }))
        return orc;
    hipLaunchKernelGGL(ltex_prep_kernel, dim3((a.n_list + 255) / 256), dim3(256), 0, st, a);
    if (a.plane16) hipLaunchKernelGGL(ltex_load_kernel<true>, dim3(a.cap_load), dim3(256), a.lds_load_bytes, st, a);
    else hipLaunchKernelGGL(ltex_load_kernel<false>, dim3(a.cap_load), dim3(256), a.lds_load_bytes, st, a);
    const uint32_t n_sweep = (a.mask & NYXHIP_FAM_GLSZM) ? a.n_list : 0u;
    const bool strips = (a.mask & (NYXHIP_FAM_GLRLM | NYXHIP_FAM_NGTDM)) != 0;
    const uint32_t grid = n_sweep + (strips ? (a.cap_strip + a.strip_groups - 1) / a.strip_groups : 0u);
    if (grid) {
        if (a.plane16) hipLaunchKernelGGL(ltex_strip_kernel<true>, dim3(grid), dim3(a.strip_threads), a.lds_strip_bytes, st, a, n_sweep);
        else hipLaunchKernelGGL(ltex_strip_kernel<false>, dim3(grid), dim3(a.strip_threads), a.lds_strip_bytes, st, a, n_sweep);
    }
    if (a.mask & (NYXHIP_FAM_GLSZM | NYXHIP_FAM_GLRLM)) {
        if (a.plane16) hipLaunchKernelGGL(ltex_post_kernel<true>, dim3(a.cap_strip), dim3(256), a.lds_zone_bytes, st, a);
        else hipLaunchKernelGGL(ltex_post_kernel<false>, dim3(a.cap_strip), dim3(256), a.lds_zone_bytes, st, a);
    }
    hipLaunchKernelGGL(ltex_finish_kernel, dim3(a.n_list), dim3(256), a.lds_fin_bytes, st, a);
    return (int)hipGetLastError();
}

} // namespace nyxhip
