// roi_large.hip -- INTENSITY + GLCM of ROIs beyond the LDS size classes, several workgroups per ROI (gfx950).
//
// The reference hands any ROI to any worker thread (/root/reference/src/nyx/parallel.h:23-42, roi_cache.h:31-84); the LDS
// kernels of roi_features.hip give an ROI one workgroup, which leaves a 120 k-pixel ROI walking its cloud with 256 threads
// while the rest of the chip idles.  Here such an ROI is cut up:
//
//   large_prep_kernel    one thread per ROI: its block of the workspace, its slabs / strips in the two work maps
//   large_load_kernel    one workgroup per SLAB of the pixel cloud (the only pass over HBM: 8 B per pixel): exact sums, the
//                        intensity histogram over [min, max] (counted in LDS, flushed with contiguous atomic adds), the binned
//                        bounding-box plane (features/texture_feature.h binning) scattered to the workspace
//   large_cooc_kernel    one workgroup per STRIP of plane rows: co-occurrence counts of all angles (features/glcm.cpp:343-485)
//                        in LDS, flushed with atomic adds
//   large_finish_kernel  one workgroup per ROI: every first-order feature is a function of the histogram
//                        (features/intensity.cpp:57-192, histogram.h:27-309, moments.h:48-109), the Haralick features of the
//                        matrices (glcm_rows.h)
//
// Everything that crosses a workgroup is an integer added with atomics, so the result is independent of the cut and of the
// arrival order; the floating-point sums of the last kernel run in a fixed order.  Kernel boundaries are the only
// synchronisation (no tickets, no fences).
//
// Built with -ffp-contract=off (device_math.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "device_math.h"
#include "roi_kernel.h"
#include "launch_util.h"
#include "glcm_rows.h"
#include "intensity_table.h"
#include "../../include/nyxhip.h"

namespace nyxhip {

namespace {

// what the kernels agree on for one ROI
struct LargeRoi {
    uint64_t roi, off;
    uint32_t n, w, h, vmin, vmax, range;
    uint64_t area;
    uint32_t ng_bound, lvl_cap;
    LargeWs L;
    unsigned char* base;
    // The plane is laid out along the cloud: a cloud in column-major scan order (the in-memory workflow, phase2_2d.cpp:655-656)
    // gets a column-major plane (tr = 1: cell = x * h + y), any other order a row-major one -- consecutive pixels of a slab then
    // store consecutive bytes (scattered one byte per cache line, the stores were more than half of the load kernel's time).
    // pw x ph: the plane as the co-occurrence kernel walks it (rows of pw cells).
    uint32_t tr, pw, ph;
};
__device__ __forceinline__ void large_set_orientation(LargeRoi& R, uint32_t tr)
{
    R.tr = tr; R.pw = tr ? R.h : R.w; R.ph = tr ? R.w : R.h;
}

__device__ __forceinline__ bool large_roi(const LargeArgs& A, uint32_t j, LargeRoi& R, bool need_block)
{
    R.roi = A.list[j];
    R.off = A.px_offset[R.roi];
    R.n = (uint32_t)(A.px_offset[R.roi + 1] - R.off);
    R.w = A.bbox_w[R.roi]; R.h = A.bbox_h[R.roi];
    R.vmin = A.min_inten[R.roi]; R.vmax = A.max_inten[R.roi];
    R.range = R.vmax - R.vmin;
    R.area = (uint64_t)R.w * R.h;
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    R.ng_bound = greyInfo > 0 ? (uint32_t)greyInfo : greyInfo < 0 ? (uint32_t)(-greyInfo) : R.vmax;
    R.lvl_cap = greyInfo < 0 ? (uint32_t)(-greyInfo) : 0u;
    R.L = large_ws_layout(R.range, R.area, R.ng_bound, R.lvl_cap, (uint32_t)A.glcm_na, A.plane16 != 0, (A.mask & NYXHIP_FAM_INTENSITY) != 0,
                          (A.mask & NYXHIP_FAM_GLCM) != 0);
    R.base = nullptr;
    large_set_orientation(R, 0);
    if (need_block) {
        const uint64_t o = A.ws_off[j];
        if (o == ~0ull) return false;
        R.base = A.ws + (o & ~255ull);                                       // (blocks are 256-byte aligned: bit 0 carries the orientation)
        large_set_orientation(R, (uint32_t)(o & 1ull));
    }
    return true;
}

// served here: a non-empty ROI whose histogram the workspace can hold (wider ranges take the sort path of roi_features.hip)
__device__ __forceinline__ bool large_served(const LargeRoi& R) { return R.n != 0 && R.range < kLargeRangeMax; }

__device__ __forceinline__ uint32_t rows_per_strip(uint32_t w) { const uint32_t r = kLargeCells / (w ? w : 1u); return r ? r : 1u; }

// ---- prep: one thread per member, one set of cursor adds per block -------------------------------------------------------------
// (one wave per member with its own three adds on the shared cursors cost 11 ns per add: 130 us for 3700 members)
__global__ __launch_bounds__(256) void large_prep_kernel(const LargeArgs A)
{
    __shared__ unsigned long long s_bytes[256];
    __shared__ uint32_t s_load[256], s_cooc[256];
    __shared__ unsigned long long s_base_bytes;
    __shared__ uint32_t s_base_load, s_base_cooc;
    const int tid = threadIdx.x;
    const uint32_t j = blockIdx.x * 256u + (uint32_t)tid;
    LargeRoi R;
    bool served = false;
    uint32_t g_load = 0, g_cooc = 0;
    if (j < A.n_list) {
        large_roi(A, j, R, false);
        served = large_served(R);
        if (served) {
            g_load = (uint32_t)(((R.off & 3ull) + R.n + A.px_per_wg - 1) / A.px_per_wg);   // slabs start at the group of four that holds the first pixel
            // scan order of the cloud, read off a pair of pixels in its middle (the first column of a disk is a single pixel)
            const uint64_t m = R.off + (R.n >= 2 ? R.n / 2 - 1 : 0);
            large_set_orientation(R, (R.n >= 2 && A.x[m + 1] == A.x[m] && (uint32_t)A.y[m + 1] == (uint32_t)A.y[m] + 1u) ? 1u : 0u);
            const uint32_t rps = rows_per_strip(R.pw);
            g_cooc = (A.mask & NYXHIP_FAM_GLCM) ? (R.ph + rps - 1) / rps : 0u;
        }
    }
    s_bytes[tid] = served ? R.L.total : 0ull; s_load[tid] = g_load; s_cooc[tid] = g_cooc;
    __syncthreads();
    if (tid == 0) {                                         // exclusive prefixes over the block's members (256 short adds), then the cursors
        unsigned long long b = 0; uint32_t l = 0, c = 0;
        for (int k = 0; k < 256; k++) {
            const unsigned long long tb = s_bytes[k]; const uint32_t tl = s_load[k], tc = s_cooc[k];
            s_bytes[k] = b; s_load[k] = l; s_cooc[k] = c;
            b += tb; l += tl; c += tc;
        }
        s_base_bytes = b ? atomicAdd((unsigned long long*)A.ctr, b) : 0ull;
        s_base_load = l ? atomicAdd(&A.ctr[2], l) : 0u;
        s_base_cooc = c ? atomicAdd(&A.ctr[3], c) : 0u;
    }
    __syncthreads();
    if (j >= A.n_list) return;
    if (!served) { A.ws_off[j] = ~0ull; return; }
    const unsigned long long off = s_base_bytes + s_bytes[tid];
    const uint32_t b_load = s_base_load + s_load[tid], b_cooc = s_base_cooc + s_cooc[tid];
    const bool fits = off + R.L.total <= A.ws_bytes && (uint64_t)b_load + g_load <= A.cap_load && (uint64_t)b_cooc + g_cooc <= A.cap_cooc;
    if (!fits) {                                            // (the host sized all three from the class totals: cannot happen)
        A.ws_off[j] = ~0ull;
        atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        return;
    }
    A.ws_off[j] = off | R.tr;
    for (uint32_t s = 0; s < g_load; s++) A.map_load[b_load + s] = make_uint2(j, s);
    for (uint32_t s = 0; s < g_cooc; s++) A.map_cooc[b_cooc + s] = make_uint2(j, s);
}

// ---- load: one workgroup per slab of the cloud --------------------------------------------------------------------------------
template <bool P16>
__global__ __launch_bounds__(1024) void large_load_kernel(const LargeArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= A.ctr[2]) return;
    const uint2 job = A.map_load[blockIdx.x];
    LargeRoi R;
    if (!large_roi(A, job.x, R, true)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, BS = blockDim.x, NWV = BS >> 6;
    const bool do_int = (A.mask & NYXHIP_FAM_INTENSITY) != 0, do_glcm = (A.mask & NYXHIP_FAM_GLCM) != 0;
    uint32_t* const s_tab = (uint32_t*)lds_raw;                              // [tab_lds / 2] words of two 16-bit counters
    unsigned long long* const s_red = (unsigned long long*)(lds_raw + 2ull * A.tab_lds);   // [2 * 16]
    unsigned long long* const hdr = (unsigned long long*)R.base;
    uint32_t* const T = (uint32_t*)(R.base + R.L.tab);
    uint16_t* const flags = (uint16_t*)(R.base + R.L.lvl);
    using plane_t = typename std::conditional<P16, uint16_t, uint8_t>::type;
    plane_t* const plane = (plane_t*)(R.base + R.L.plane);
    const bool tab_in_lds = do_int && R.range < A.tab_lds;                   // (a slab has < 65536 pixels: 16-bit counters never carry)
    if (tab_in_lds) {
        for (uint32_t i = tid; i < (R.range + 2) / 2; i += BS) s_tab[i] = 0;
        __syncthreads();
    }
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    const double mslope = greyInfo > 0 ? (double)greyInfo / ((double)R.vmax - 0.) : 0.0;
    // Slabs are cut in the batch's GLOBAL pixel index, at multiples of four from the group of four that holds the ROI's first pixel:
    // a thread takes four consecutive pixels -- one 16-byte load of intensities, one 8-byte load each of x and y (the arrays of a
    // batch are 16 / 8-byte aligned: vec_ok) -- where a pixel per lane cost three loads per pixel.  The ROI's first and last group
    // (partly another ROI's pixels, or past the end of the arrays) go element by element.
    const uint64_t roi_lo = R.off, roi_hi = R.off + R.n;
    const uint64_t gs = (R.off & ~3ull) + (uint64_t)job.y * A.px_per_wg;      // first global index of the slab (a multiple of four)
    const uint64_t ge = gs + A.px_per_wg < roi_hi ? gs + A.px_per_wg : roi_hi;
    unsigned long long sum = 0, sumsq = 0;
    const uint32_t lvl_clip = P16 ? 0xFFFFu : 0xFFu;
    const uint32_t pw_ = R.tr ? R.h : R.w;                                    // cell = major * pw_ + minor: below 2^32 (box sides are 16-bit)
    auto pixel = [&](uint32_t v, uint32_t px, uint32_t py) {
        sum += v;
        sumsq += (uint32_t)(v * v);                                           // unsigned-int product, wraps (intensity.cpp:90)
        if (do_int && !(A.dbg & 2)) {
            const uint32_t ci = v - R.vmin;
            if (tab_in_lds) atomicAdd(&s_tab[ci >> 1], 1u << (16 * (ci & 1u)));
            else if (ci <= R.range) atomicAdd(&T[ci], 1u);
        }
        if (do_glcm) {
            uint32_t lvl = 0;
            if (v != 0) {                                                      // original intensity 0 is skipped by the scan (glcm.cpp:445)
                if (greyInfo > 0) {   // matlab binning of a non-zero value: floor(slope v + 1) >= 1 already (the conversion truncates a positive value)
                    const uint32_t sc = (uint32_t)(mslope * (double)v + 1.0);
                    lvl = sc > (uint32_t)greyInfo ? (uint32_t)greyInfo : sc;
                } else
                    lvl = greyInfo < 0 ? bin_radiomix(v, R.vmin, R.vmax, -greyInfo) : v;
                if (greyInfo < 0 && lvl <= R.lvl_cap) flags[lvl] = 1;
            }
            if (px < R.w && py < R.h && !(A.dbg & 1))
                plane[(R.tr ? px : py) * pw_ + (R.tr ? py : px)] = (plane_t)(lvl > lvl_clip ? lvl_clip : lvl);
        }
    };
    constexpr int kU = 4;
    for (uint64_t gb = gs; gb < ge; gb += 4ull * kU * BS) {
        uint4 v4[kU]; uint2 x2[kU], y2[kU];
        bool whole[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {                                        // every load of the trip before the first use
            const uint64_t g = gb + 4ull * ((uint64_t)u * BS + tid);
            whole[u] = A.vec_ok && g >= roi_lo && g + 4 <= ge;
            if (whole[u]) {
                v4[u] = *(const uint4*)(A.inten + g);
                if (do_glcm) { x2[u] = *(const uint2*)(A.x + g); y2[u] = *(const uint2*)(A.y + g); }
            }
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const uint64_t g = gb + 4ull * ((uint64_t)u * BS + tid);
            if (whole[u]) {
                const uint32_t xa = do_glcm ? x2[u].x : 0u, xb = do_glcm ? x2[u].y : 0u, ya = do_glcm ? y2[u].x : 0u, yb = do_glcm ? y2[u].y : 0u;
                pixel(v4[u].x, xa & 0xFFFFu, ya & 0xFFFFu);
                pixel(v4[u].y, xa >> 16, ya >> 16);
                pixel(v4[u].z, xb & 0xFFFFu, yb & 0xFFFFu);
                pixel(v4[u].w, xb >> 16, yb >> 16);
            } else {
                for (uint64_t i = g > roi_lo ? g : roi_lo; i < g + 4 && i < ge; i++)
                    pixel(A.inten[i], do_glcm ? (uint32_t)A.x[i] : 0u, do_glcm ? (uint32_t)A.y[i] : 0u);
            }
        }
    }
    sum = wave_sum_u64(sum);
    sumsq = wave_sum_u64(sumsq);
    if (lane == 0) { s_red[wave] = sum; s_red[16 + wave] = sumsq; }
    __syncthreads();                                                          // (also: every LDS count of the slab is in)
    if (tid == 0) {
        unsigned long long a = 0, b = 0;
        for (int wv = 0; wv < NWV; wv++) { a += s_red[wv]; b += s_red[16 + wv]; }
        atomicAdd(&hdr[0], a);
        atomicAdd(&hdr[1], b);
    }
    if (tab_in_lds && !(A.dbg & 4)) {
        const uint16_t* const t16 = (const uint16_t*)s_tab;
        for (uint32_t i = tid; i <= R.range; i += BS) {                       // contiguous adds: a wave covers 256 bytes of the table
            const uint32_t c = t16[i];
            if (c) atomicAdd(&T[i], c);
        }
    }
}

// ---- co-occurrence: one workgroup per strip of plane rows ---------------------------------------------------------------------
// matrix order and level -> matrix index of an ROI (glcm.cpp:388-420): s_map[level] = index + 1 (0 = skip) under radiomics binning
template <typename MAP>
__device__ __forceinline__ int large_matrix_order(const LargeArgs& A, const LargeRoi& R, MAP* s_map, double* s_I, int tid, int BS)
{
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    if (greyInfo > 0) return greyInfo;
    if (greyInfo == 0) return (int)R.vmax;                                   // IBSI: the largest level is the largest intensity
    const uint16_t* const flags = (const uint16_t*)(R.base + R.L.lvl);
    __shared__ int s_ng;
    if (tid == 0) {
        int k = 0;
        for (uint32_t l = 1; l <= R.lvl_cap; l++) {
            const bool on = flags[l] != 0;
            if (s_map) s_map[l] = on ? (MAP)(k + 1) : (MAP)0;
            if (on) { if (s_I) s_I[k] = (double)l; k++; }
        }
        s_ng = k;
    }
    __syncthreads();
    return s_ng;
}

template <bool P16>
__global__ __launch_bounds__(256) void large_cooc_kernel(const LargeArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= A.ctr[3]) return;
    const uint2 job = A.map_cooc[blockIdx.x];
    LargeRoi R;
    if (!large_roi(A, job.x, R, true)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int BS = 256, NW = 4;
    // degenerate guard (glcm.cpp:27-95, on GLCM_GREYDEPTH): nothing to count
    if (bin_pixel(R.vmin, R.vmin, R.vmax, A.glcm_grey_depth) == bin_pixel(R.vmax, R.vmin, R.vmax, A.glcm_grey_depth)) return;
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    uint16_t* const s_map = (uint16_t*)lds_raw;                               // [lvl_cap + 2] (radiomics only)
    const uint32_t map_bytes = greyInfo < 0 ? ((2u * (R.lvl_cap + 2) + 15u) & ~15u) : 0u;
    uint32_t* const s_P = (uint32_t*)(lds_raw + map_bytes);
    const int Ng = large_matrix_order<uint16_t>(A, R, greyInfo < 0 ? s_map : nullptr, (double*)nullptr, tid, BS);
    if (Ng <= 0) return;
    const int na = A.glcm_na;
    const uint64_t NN = (uint64_t)Ng * Ng;
    uint32_t* const gP = (uint32_t*)(R.base + R.L.P);
    // LDS matrices have order Ng + 1 and are indexed by the level itself: row 0 / column 0 collect the pairs whose partner is
    // skipped (level 0: background, zero intensity, outside the box), so the offset-1 sweep needs no test per pair; the flush
    // drops them.  Matrices that do not fit LDS are counted in place (plain Ng x Ng, one test per pair).
    const uint32_t NG1 = (uint32_t)Ng + 1u;
    const uint64_t CC = (uint64_t)NG1 * NG1;
    const bool in_lds = 4ull * na * CC + map_bytes <= A.lds_P_bytes;
    uint32_t* const P = in_lds ? s_P : gP;
    const uint32_t ldP = in_lds ? NG1 : (uint32_t)Ng, off1 = in_lds ? 0u : 1u;        // cell of (a, b) = q * cells + (a - off1) * ldP + (b - off1)
    const uint64_t cellsP = in_lds ? CC : NN;
    if (in_lds) {
        for (uint32_t i = tid; i < (uint32_t)(na * CC); i += BS) s_P[i] = 0;
        __syncthreads();
    }
    const bool symmetric = A.glcm_symmetric || greyInfo <= 0;                 // glcm.cpp:475
    int ddx[kMaxAngles], ddy[kMaxAngles];
#pragma unroll
    for (int q = 0; q < kMaxAngles; q++) {
        const int ang = A.glcm_angles[q < na ? q : 0];                        // glcm.cpp:234-255
        const int dx = ang == 90 ? 0 : ang == 135 ? -A.glcm_offset : A.glcm_offset, dy = ang == 0 ? 0 : A.glcm_offset;
        ddx[q] = R.tr ? dy : dx;                                               // (a column-major plane: its rows are the image's columns)
        ddy[q] = R.tr ? dx : dy;
    }
    using plane_t = typename std::conditional<P16, uint16_t, uint8_t>::type;
    const plane_t* const plane = (const plane_t*)(R.base + R.L.plane);
    const uint32_t rps = rows_per_strip(R.pw);
    const uint32_t r0 = job.y * rps, r1 = r0 + rps < R.ph ? r0 + rps : R.ph;
    const int w = (int)R.pw, h = (int)R.ph;
    // The strip and its halo rows (offset rows above and below: a column-major plane has its 135-degree neighbours above) are staged
    // in LDS with 16-byte loads; the pair loop then reads LDS bytes.  (Reading the plane from global memory left the kernel waiting
    // on one dependent L2 round trip per 64 cells.)  Strips whose rows do not fit -- boxes tens of thousands of cells wide -- read
    // the workspace directly.
    const int d = A.glcm_offset;
    const int s0 = (int)r0 - d > 0 ? (int)r0 - d : 0, s1 = (int)r1 + d < h ? (int)r1 + d : h;     // staged rows [s0, s1)
    const uint64_t st_cells = (uint64_t)(s1 - s0) * R.pw;
    const uint32_t p_bytes = in_lds ? (uint32_t)((4ull * na * CC + 15) & ~15ull) : 0u;
    const bool staged = map_bytes + p_bytes + st_cells * sizeof(plane_t) + 16 <= A.lds_P_bytes + A.lds_strip_bytes;
    plane_t* const s_strip = (plane_t*)(lds_raw + map_bytes + p_bytes);
    if (staged) {
        const plane_t* const src = plane + (uint64_t)s0 * R.pw;
        const uint64_t nbytes = st_cells * sizeof(plane_t);
        const uint32_t mis = (uint32_t)((uintptr_t)src & 15u);                 // (the strip starts anywhere inside the plane: align the vector loads down)
        const uint4* const src16 = (const uint4*)((const unsigned char*)src - mis);
        uint4* const dst16 = (uint4*)s_strip;                                  // staged at the same misalignment: cell k of the strip sits at byte mis + k
        const uint64_t nvec = (mis + nbytes + 15) / 16;
        for (uint64_t i = tid; i < nvec; i += BS) dst16[i] = src16[i];         // (reads up to 15 bytes around the strip: inside the ROI's block, which is padded)
        __syncthreads();
    }
    const plane_t* const strip = staged ? (const plane_t*)((const unsigned char*)s_strip + ((uintptr_t)(plane + (uint64_t)s0 * R.pw) & 15u)) : nullptr;
    int slot_of[4] = {-1, -1, -1, -1};                                        // matrix of the pass that angle 0 / 45 / 90 / 135 counts into
#pragma unroll
    for (int q = 0; q < kMaxAngles; q++)
        if (q < na) { const int ang = A.glcm_angles[q]; slot_of[ang == 0 ? 0 : ang == 45 ? 1 : ang == 90 ? 2 : 3] = q; }
    const bool usual = na == 4 && slot_of[0] >= 0 && slot_of[1] >= 0 && slot_of[2] >= 0 && slot_of[3] >= 0 && !symmetric && greyInfo >= 0;
    if (staged && d == 1 && in_lds && usual) {
        // ---- the usual request (four angles, offset 1, asymmetric counts, level = matrix index): the sweep below as straight-line
        // code -- levels travel as byte offsets (4 * level), a cell address is matrix base + row offset + neighbour offset, a skipped
        // partner lands in row / column 0 of the (Ng + 1)-order matrices, so a row of 62 centres costs one LDS read, three lane
        // shifts, one multiply and four add + atomic pairs.  (The general sweep spent ~48 vector instructions per row on 64-bit
        // index arithmetic and per-angle selects.)
        const uint32_t ccb = (uint32_t)CC * 4u;
        char* const T0 = (char*)s_P + (uint32_t)slot_of[0] * ccb;
        char* const T1 = (char*)s_P + (uint32_t)slot_of[1] * ccb;
        char* const T2 = (char*)s_P + (uint32_t)slot_of[2] * ccb;
        char* const T3 = (char*)s_P + (uint32_t)slot_of[3] * ccb;
        const uint32_t ncs = ((uint32_t)w + 61u) / 62u;
        const uint32_t nrb = ncs >= (uint32_t)NW ? 1u : (uint32_t)NW / ncs;
        const uint32_t rows_blk = (r1 - r0 + nrb - 1) / nrb;
        const uint32_t pw = R.pw, tr = R.tr;
        for (uint32_t t = (uint32_t)wave; t < ncs * nrb; t += NW) {
            const uint32_t cs = t % ncs, rbi = t / ncs;
            const uint32_t ra = r0 + rbi * rows_blk, rz = ra + rows_blk < r1 ? ra + rows_blk : r1;
            const int c = (int)(cs * 62u) - 1 + lane;
            const bool in_col = c >= 0 && c < w;
            const bool centre = in_col && lane >= 1 && lane <= 62;
            const plane_t* pp = strip + (uint32_t)((int)ra - s0) * pw + (uint32_t)(in_col ? c : 0);
            uint32_t cur4 = 0;
            if (in_col && ra < rz) cur4 = (uint32_t)pp[0] << 2;
            for (uint32_t row = ra; row < rz; row++) {
                pp += pw;
                uint32_t nxt4 = 0;
                if (in_col && (int)row + 1 < h) nxt4 = (uint32_t)pp[0] << 2;
                const uint32_t e4 = lane_plus1_z(cur4), se4 = lane_plus1_z(nxt4), sw4 = lane_minus1_z(nxt4);
                if (centre && cur4 != 0) {
                    const uint32_t rowb = mul_u24_su(cur4, NG1);
                    if (!tr) {
                        atomicAdd((uint32_t*)(T0 + rowb + e4), 1u);
                        atomicAdd((uint32_t*)(T1 + rowb + se4), 1u);
                        atomicAdd((uint32_t*)(T2 + rowb + nxt4), 1u);
                        atomicAdd((uint32_t*)(T3 + rowb + sw4), 1u);
                    } else {                               // column-major plane: 0 degrees is the cell below, 90 the cell to the right, and the
                        atomicAdd((uint32_t*)(T0 + rowb + nxt4), 1u);   // 135-degree pair is counted from its neighbour (this cell): its centre is below-left
                        atomicAdd((uint32_t*)(T1 + rowb + se4), 1u);
                        atomicAdd((uint32_t*)(T2 + rowb + e4), 1u);
                        atomicAdd((uint32_t*)(T3 + mul_u24_su(sw4, NG1) + cur4), 1u);
                    }
                }
                cur4 = nxt4;
            }
        }
    } else
    if (staged && d == 1) {
        // ---- offset 1 on a staged strip: lane = column, the horizontal neighbours through DPP lane shifts, the row below read once
        // and kept as the next centre row (the sweep of roi_features.hip).  A column strip is 62 centre columns between two halo
        // lanes; (column strip, row block) tasks are dealt to the four waves.  Every angle is one of E / SE / S / SW of the centre in
        // plane coordinates -- a column-major plane turns 135 degrees into "north-east", which is counted from the neighbour's
        // side instead (rev: the cell below-left is the centre of the pair).
        int dir[kMaxAngles], rev[kMaxAngles];
#pragma unroll
        for (int q = 0; q < kMaxAngles; q++) {
            int ex = ddx[q], ey = ddy[q];
            rev[q] = (ey < 0 || (ey == 0 && ex < 0)) ? 1 : 0;
            if (rev[q]) { ex = -ex; ey = -ey; }
            dir[q] = ey == 0 ? 0 : ex > 0 ? 1 : ex == 0 ? 2 : 3;
        }
        const uint32_t ncs = ((uint32_t)w + 61u) / 62u;
        const uint32_t nrb = ncs >= (uint32_t)NW ? 1u : (uint32_t)NW / ncs;
        const uint32_t rows_blk = (r1 - r0 + nrb - 1) / nrb;
        const bool remap = greyInfo < 0;
        for (uint32_t t = (uint32_t)wave; t < ncs * nrb; t += NW) {
            const uint32_t cs = t % ncs, rbi = t / ncs;
            const uint32_t ra = r0 + rbi * rows_blk, rz = ra + rows_blk < r1 ? ra + rows_blk : r1;
            const int c = (int)(cs * 62u) - 1 + lane;
            const bool in_col = c >= 0 && c < w;
            const bool centre = in_col && lane >= 1 && lane <= 62;
            auto lvl = [&](uint32_t row) -> uint32_t {
                uint32_t v = (in_col && (int)row < h) ? (uint32_t)strip[(uint64_t)((int)row - s0) * R.pw + (uint32_t)c] : 0u;
                if (remap && v) v = s_map[v];
                return v;
            };
            uint32_t cur = ra < rz ? lvl(ra) : 0u;
            for (uint32_t row = ra; row < rz; row++) {
                const uint32_t nxt = lvl(row + 1);
                const uint32_t nb_e = lane_plus1(cur, 0), nb_se = lane_plus1(nxt, 0), nb_sw = lane_minus1(nxt, 0);
                if (centre && cur != 0) {
#pragma unroll
                    for (int q = 0; q < kMaxAngles; q++) {
                        if (q >= na) break;
                        const uint32_t nb = dir[q] == 0 ? nb_e : dir[q] == 1 ? nb_se : dir[q] == 2 ? nxt : nb_sw;
                        if (!in_lds && nb == 0) continue;                                    // (LDS matrices: a skipped partner lands in row / column 0)
                        const uint32_t ca = rev[q] ? nb : cur, cb = rev[q] ? cur : nb;       // matrix row = the centre's level
                        uint32_t* const Pq = P + (uint64_t)q * cellsP;
                        atomicAdd(&Pq[(uint64_t)(ca - off1) * ldP + (cb - off1)], 1u);
                        if (symmetric) atomicAdd(&Pq[(uint64_t)(cb - off1) * ldP + (ca - off1)], 1u);
                    }
                }
                cur = nxt;
            }
        }
    } else
    for (uint32_t row = r0 + wave; row < r1; row += NW) {
        const plane_t* const prow = staged ? strip + (uint64_t)((int)row - s0) * R.pw : plane + (uint64_t)row * R.pw;
        for (int col = lane; col < w; col += 64) {
            const uint32_t lb = prow[col];
            if (lb == 0) continue;
            const int ib = greyInfo < 0 ? (int)s_map[lb] - 1 : (int)lb - 1;
#pragma unroll
            for (int q = 0; q < kMaxAngles; q++) {
                if (q >= na) break;
                const int r2 = (int)row + ddy[q], c2 = col + ddx[q];
                if (r2 < 0 || r2 >= h || c2 < 0 || c2 >= w) continue;
                const uint32_t la = staged ? strip[(uint64_t)(r2 - s0) * R.pw + (uint32_t)c2] : plane[(uint64_t)r2 * R.pw + (uint32_t)c2];
                if (la == 0) continue;
                const int ia = greyInfo < 0 ? (int)s_map[la] - 1 : (int)la - 1;
                uint32_t* const Pq = P + (uint64_t)q * cellsP;
                atomicAdd(&Pq[(uint64_t)(ib + 1 - (int)off1) * ldP + (uint32_t)(ia + 1 - (int)off1)], 1u);
                if (symmetric) atomicAdd(&Pq[(uint64_t)(ia + 1 - (int)off1) * ldP + (uint32_t)(ib + 1 - (int)off1)], 1u);
            }
        }
    }
    if (in_lds) {
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)(na * NN); i += BS) {          // cell (q, r, c) of the workspace matrices <- (q, r + 1, c + 1) here
            const uint32_t q = i / (uint32_t)NN, rem = i - q * (uint32_t)NN, r = rem / (uint32_t)Ng, c = rem - r * (uint32_t)Ng;
            const uint32_t v = s_P[(uint64_t)q * CC + (uint64_t)(r + 1) * NG1 + c + 1];
            if (v) atomicAdd(&gP[i], v);
        }
    }
}

// ---- finish: one workgroup per ROI ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void large_finish_kernel(const LargeArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    LargeRoi R;
    if (blockIdx.x >= A.n_list) return;
    if (!large_roi(A, blockIdx.x, R, true)) {
        // not served here.  An EMPTY member (a box with no pixel) is nobody else's either: the columns of the LDS kernels' early exit
        // (ranges beyond the histogram are the sort path's, which writes the row)
        if (R.n == 0)
            for (int c = threadIdx.x; c < A.n_cols; c += 256) A.out[R.roi * A.ld + c] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    constexpr int BS = 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool do_int = (A.mask & NYXHIP_FAM_INTENSITY) != 0, do_glcm = (A.mask & NYXHIP_FAM_GLCM) != 0;
    double* const out_row = A.out + R.roi * A.ld;
    for (int c = tid; c < A.n_cols; c += BS) out_row[c] = 0.0;               // skipped features stay 0 (class members default to 0)
    __shared__ double s_x[32];
    __shared__ unsigned long long s_u[8];
    __shared__ double s_stat[8];
    __shared__ double s_pq[8];
    __shared__ uint32_t s_w[16];
    const uint32_t n = R.n, vmin = R.vmin, vmax = R.vmax, range = R.range;
    __syncthreads();

    if (do_int) {
        double* const o = out_row + A.col_intensity;
        const unsigned long long* const hdr = (const unsigned long long*)R.base;
        // counts, then (in place) inclusive prefix sums: staged in LDS when the table fits (every later pass -- the prefix sums, the
        // order statistics' searches, the robust sweeps -- is then a chain of LDS instead of L2 round trips)
        uint32_t* T = (uint32_t*)(R.base + R.L.tab);
        if (4ull * ((uint64_t)range + 1) <= A.fin_tab_bytes) {
            uint32_t* const s_T = (uint32_t*)lds_raw;
            for (uint32_t i = tid; i <= range; i += BS) s_T[i] = T[i];
            T = s_T;
            __syncthreads();
        }
        // inclusive prefix sums in place: four entries per thread and step, wave scan, cross-wave carry
        uint32_t carry = 0;
        for (uint32_t i0 = 0; i0 <= range; i0 += BS * 4) {
            const uint32_t i = i0 + 4u * (uint32_t)tid;
            uint32_t c[4];
#pragma unroll
            for (int k = 0; k < 4; k++) c[k] = i + k <= range ? T[i + k] : 0u;
            c[1] += c[0]; c[2] += c[1]; c[3] += c[2];
            const uint32_t sc = wave_scan_u32(c[3]);
            __syncthreads();
            if (lane == 63) s_w[wave] = sc;
            __syncthreads();
            uint32_t excl = carry + sc - c[3];
            for (int wv = 0; wv < wave; wv++) excl += s_w[wv];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + k <= range) T[i + k] = c[k] + excl;
            carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        }
        __syncthreads();
        struct DenseTab {                                  // entry i = value vmin + i
            const uint32_t* T; uint32_t m;
            __device__ __forceinline__ uint32_t off(uint32_t i) const { return i; }
            __device__ __forceinline__ uint32_t cum(uint32_t i) const { return T[i]; }
            __device__ __forceinline__ uint32_t first_ge(uint64_t d) const { return d < m ? (uint32_t)d : m; }
        } tab{T, range + 1};
        IntensityScratch S{s_x, s_u, s_stat, s_pq, s_w, (uint32_t*)(lds_raw + A.fin_tab_bytes), (uint32_t*)(lds_raw + A.fin_tab_bytes) + 104};
        const bool have_slide = A.slide_min && A.slide_max;
        const TableSums<DenseTab> tsums{tab, vmin};
        intensity_from_table(tab, tsums, n, vmin, vmax, (double)hdr[0], (double)hdr[1], have_slide, have_slide ? A.slide_max[R.roi] - A.slide_min[R.roi] : 0.0,
                             (uint32_t)A.n_hist, o, S, tid);
        __syncthreads();
    }

    if (do_glcm) {
        double* const o = out_row + A.col_glcm;
        const int na = A.glcm_na, ncol_g = kGlcmAngled * na + kGlcmAve;
        const int greyInfo = A.ibsi ? 0 : A.grey_depth;
        if (bin_pixel(vmin, vmin, vmax, A.glcm_grey_depth) == bin_pixel(vmax, vmin, vmax, A.glcm_grey_depth)) {   // glcm.cpp:27-95
            for (int c = tid; c < ncol_g; c += BS) o[c] = A.soft_nan;
            return;
        }
        // level values | per-angle features | per-angle marginals: in LDS up to ~240 levels, else in the ROI's block (IBSI on wide data)
        const uint32_t ngb = R.ng_bound;
        double* const scratch = large_glcm_scratch_bytes(ngb) > kLargeScratchLds ? (double*)(R.base + R.L.scr) : (double*)lds_raw;
        double* const s_I = scratch;
        double* const s_f = s_I + ngb;
        double* const s_scr = s_f + kMaxAngles * 32;
        const int Ng = large_matrix_order<uint16_t>(A, R, (uint16_t*)nullptr, greyInfo < 0 ? s_I : (double*)nullptr, tid, BS);
        if (greyInfo >= 0)
            for (int i = tid; i < Ng; i += BS) s_I[i] = (double)(i + 1);
        const uint32_t* const gP = (const uint32_t*)(R.base + R.L.P);
        const uint64_t NN = (uint64_t)Ng * Ng;
        if (Ng <= 0) {
            for (int c = tid; c < ncol_g; c += BS) o[c] = A.soft_nan;
            return;
        }
        // the matrices: staged in LDS behind the scratch when they fit (the feature routine makes ~10 passes over them), else read in place
        const bool scr_lds = scratch == (double*)lds_raw;
        const bool p_lds = scr_lds && 4ull * na * NN <= A.fin_P_bytes;
        uint32_t* const s_Pm = (uint32_t*)(lds_raw + ((large_glcm_scratch_bytes(ngb) + 15) & ~15ull));
        if (p_lds)
            for (uint32_t i = tid; i < (uint32_t)(na * NN); i += BS) s_Pm[i] = gP[i];
        const uint32_t* const Pm = p_lds ? s_Pm : gP;
        __syncthreads();
        if (p_lds) {
            if (Ng <= 16) {                               // small matrices: the angles share one wave's instruction stream
                if (wave == 0) glcm_features_rows<false, 16, 900>(Pm, na, Ng, s_I, s_scr, 6 * (int)ngb, A.soft_nan, s_f, lane);
            } else if (wave < na)                         // a wave per angle, 64 lanes over the cells
                glcm_features_rows<false, 64, 901>(Pm + (size_t)wave * NN, 1, Ng, s_I, s_scr + (size_t)wave * 6 * ngb, 6 * (int)ngb, A.soft_nan, s_f + wave * 32, lane);
        } else {
            if (Ng <= 16) {
                if (wave == 0) glcm_features_rows<true, 16, 902>(Pm, na, Ng, s_I, s_scr, 6 * (int)ngb, A.soft_nan, s_f, lane);
            } else if (wave < na)
                glcm_features_rows<true, 64, 903>(Pm + (size_t)wave * NN, 1, Ng, s_I, s_scr + (size_t)wave * 6 * ngb, 6 * (int)ngb, A.soft_nan, s_f + wave * 32, lane);
        }
        __syncthreads();
        for (int c = tid; c < kGlcmAngled * na; c += BS) {   // feature-major, angle-minor (output_2_buffer.cpp:336-346)
            const int k = c / na, a = c - k * na;
            o[c] = s_f[a * 32 + k];
        }
        for (int j = tid; j < kGlcmAve; j += BS) {           // calc_ave (glcm.cpp:1205-1214): std::reduce folds four at a time
            const int k = c_glcm_ave_order[j];
            double init = 0.0;
            int a = 0;
            for (; na - a >= 4; a += 4) {
                const double v1 = s_f[a * 32 + k] + s_f[(a + 1) * 32 + k];
                const double v2 = s_f[(a + 2) * 32 + k] + s_f[(a + 3) * 32 + k];
                init = init + (v1 + v2);
            }
            for (; a < na; a++) init = init + s_f[a * 32 + k];
            o[kGlcmAngled * na + j] = na ? init / (double)na : 0.0;
        }
    }
}

} // namespace

// Launches the four kernels of one group.  The caller has zeroed a.ws[0 .. ws_bytes) and a.ctr on the stream.
int launch_large_features(const LargeArgs& a, void* stream)
{
    if (a.n_list == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    static DeviceOnce optin;
    if (int orc = optin.run([]() -> int {
            // (dynamic LDS beyond 64 KiB is an opt-in per kernel; the static words of a kernel count against the CU's 160 KiB too)
            const struct { const void* f; int bytes; } k[] = {
                {(const void*)large_load_kernel<false>, 136 * 1024}, {(const void*)large_load_kernel<true>, 136 * 1024},
                {(const void*)large_cooc_kernel<false>, 120 * 1024}, {(const void*)large_cooc_kernel<true>, 120 * 1024},
                {(const void*)large_finish_kernel, 64 * 1024}};
            for (const auto& e : k)
                if (hipError_t rc = hipFuncSetAttribute(e.f, hipFuncAttributeMaxDynamicSharedMemorySize, e.bytes); rc != hipSuccess) return (int)rc;
            return 0;
        }))
        return orc;
    hipLaunchKernelGGL(large_prep_kernel, dim3((a.n_list + 255) / 256), dim3(256), 0, st, a);
    const size_t lds_load = 2ull * a.tab_lds + 2 * 16 * 8;
    const uint32_t bs_load = a.px_per_wg / 32u;                              // 32 pixels per thread: 256 / 512 / 1024 threads
    if (a.plane16) hipLaunchKernelGGL(large_load_kernel<true>, dim3(a.cap_load), dim3(bs_load), lds_load, st, a);
    else hipLaunchKernelGGL(large_load_kernel<false>, dim3(a.cap_load), dim3(bs_load), lds_load, st, a);
    if (a.mask & NYXHIP_FAM_GLCM) {
        if (a.plane16) hipLaunchKernelGGL(large_cooc_kernel<true>, dim3(a.cap_cooc), dim3(256), a.lds_P_bytes + a.lds_strip_bytes, st, a);
        else hipLaunchKernelGGL(large_cooc_kernel<false>, dim3(a.cap_cooc), dim3(256), a.lds_P_bytes + a.lds_strip_bytes, st, a);
    }
    hipLaunchKernelGGL(large_finish_kernel, dim3(a.n_list), dim3(256), a.lds_fin_bytes, st, a);
    return (int)hipGetLastError();
}

} // namespace nyxhip
