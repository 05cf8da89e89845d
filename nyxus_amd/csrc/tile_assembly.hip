// tile_assembly.hip -- phases 1-2 of the in-memory workflow on the device.
//
// Replaces, for a stack of intensity/label tile pairs resident in HBM:
//   gatherRoisMetricsInMemory  /root/reference/src/nyx/phase1.cpp:373-409  + feed_pixel_2_metrics
//                              src/nyx/pixel_feed.cpp:19-43   -> per-label area, min, max, AABB
//   scanTrivialRoisInMemory    src/nyx/phase2_2d.cpp:637-684  -> per-ROI pixel clouds
// The reference does both with a serial scan and a hash-map lookup per pixel (`unordered_map<int, LR>`, roi_cache.h), so
// label VALUES are arbitrary 32-bit numbers there.  Here too: nothing is sized by the label magnitude.
//   tile_scan_kernel     one coalesced pass over the tiles in 256 x 32 blocks; per-column label runs are accumulated in
//                        registers, merged per block in an LDS hash table, and only that table's live entries reach the
//                        per-tile open-addressing tables in HBM (key = label, `cap` slots per tile);
//   tile_compact_kernel  occupied slots -> rows grouped by tile (slot order), per-tile row / pixel bases by prefix sums;
//   tile_rank_kernel     ascending label order inside every tile (the row order of save_features_2_buffer,
//                        output_2_buffer.cpp:305-306) by counting: rank = #labels of the tile below mine, CSR offset = pixels of
//                        those; the tile's prescan extrema (scan_slide_props, slideprops.cpp:456-...) fall out of the same loop;
//   roi_cloud_kernel     one workgroup per ROI scans its bounding-box window of the tile in row-major order and writes the
//                        ROI's SoA cloud with a ballot-ranked (deterministic) compaction.
// Tiles keep the caller's element type (8 / 16 / 32-bit unsigned): the kernels are instantiated per type pair, so H2D and the
// scan move the image's own bytes.
// HBM traffic per tile: (sizeof intensity + sizeof label) B/px read by the scan + the bbox windows (L2-resident re-read)
// + 8 B per ROI pixel written as clouds (only for the families that need clouds: see nyxhip_api.hip).
#include <hip/hip_runtime.h>
#include "device_math.h"
#include "roi_kernel.h"

namespace nyxhip {

__global__ void tile_init_tables_kernel(TileHash T, uint64_t n)
{
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        T.key[i] = 0; T.cnt[i] = 0; T.vmin[i] = 0xFFFFFFFFu; T.vmax[i] = 0;
        T.xmin[i] = 0xFFFFFFFFu; T.xmax[i] = 0; T.ymin[i] = 0xFFFFFFFFu; T.ymax[i] = 0;
    }
}

__device__ __forceinline__ uint32_t wave_max_u32_x(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// One workgroup owns a kScanCols x kScanRows block of one tile; a thread walks one column of the block and keeps
// the statistics of its current label run in registers (ROIs are compact: a column crosses a ROI once).  Runs
// are merged in an LDS hash table keyed by label, and only the table's few live entries go to the tile's global
// table: ~7 global atomics per (ROI, block) instead of per (ROI, row segment).
#ifndef NYX_SCAN_BATCH
#define NYX_SCAN_BATCH 8
#endif
#ifndef NYX_SCAN_ROWS
#define NYX_SCAN_ROWS 32
#endif
constexpr int kScanCols = 256, kScanRows = NYX_SCAN_ROWS, kScanCap = 512, kScanProbes = 32;

template <typename TI, typename TL>
__global__ __launch_bounds__(kScanCols) void tile_scan_kernel(const TI* __restrict__ inten, const TL* __restrict__ label, uint32_t W, uint32_t H,
                                                              uint32_t n_tiles, TileHash T, uint32_t* meta)
{
    __shared__ uint32_t s_key[kScanCap], s_cnt[kScanCap], s_vmin[kScanCap], s_vmax[kScanCap], s_xmin[kScanCap], s_xmax[kScanCap],
        s_ymin[kScanCap], s_ymax[kScanCap];
    const int tid = threadIdx.x;
    for (int i = tid; i < kScanCap; i += kScanCols) {
        s_key[i] = 0; s_cnt[i] = 0; s_vmin[i] = 0xFFFFFFFFu; s_vmax[i] = 0;
        s_xmin[i] = 0xFFFFFFFFu; s_xmax[i] = 0; s_ymin[i] = 0xFFFFFFFFu; s_ymax[i] = 0;
    }
    __syncthreads();
    const uint32_t tile = blockIdx.z;
    const uint32_t x = blockIdx.x * kScanCols + tid;
    const uint32_t y_begin = blockIdx.y * kScanRows;
    const uint32_t y_end = y_begin + kScanRows < H ? y_begin + kScanRows : H;
    const uint64_t tbase = (uint64_t)tile * T.cap;
    const uint32_t capm = T.cap - 1;

    // one run (or one LDS entry) into the tile's global table: linear probing, the slot is claimed by CAS on the label
    auto to_global = [&](uint32_t l, uint32_t cnt, uint32_t mn, uint32_t mx, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) {
        uint32_t h = (l * 2654435761u) >> T.shift;
        for (uint32_t probe = 0; probe <= capm; probe++) {
            const uint64_t g = tbase + h;
            const uint32_t prev = atomicCAS(&T.key[g], 0u, l);
            if (prev == 0u || prev == l) {
                atomicAdd(&T.cnt[g], cnt);
                atomicMin(&T.vmin[g], mn); atomicMax(&T.vmax[g], mx);
                atomicMin(&T.xmin[g], x0); atomicMax(&T.xmax[g], x1);
                atomicMin(&T.ymin[g], y0); atomicMax(&T.ymax[g], y1);
                return;
            }
            h = (h + 1) & capm;
        }
        atomicMax(&meta[7], 1u);                         // the tile holds more labels than its table has slots: the host retries with a larger one
    };
    auto flush = [&](uint32_t l, uint32_t cnt, uint32_t mn, uint32_t mx, uint32_t y0, uint32_t y1) {
        uint32_t h = (l * 2654435761u) >> 23;            // 9 bits
        for (int probe = 0; probe < kScanProbes; probe++) {
            const uint32_t prev = atomicCAS(&s_key[h], 0u, l);
            if (prev == 0u || prev == l) {
                atomicAdd(&s_cnt[h], cnt);
                atomicMin(&s_vmin[h], mn); atomicMax(&s_vmax[h], mx);
                atomicMin(&s_xmin[h], x); atomicMax(&s_xmax[h], x);
                atomicMin(&s_ymin[h], y0); atomicMax(&s_ymax[h], y1);
                return;
            }
            h = (h + 1) & (kScanCap - 1);
        }
        to_global(l, cnt, mn, mx, x, x, y0, y1);         // block table crowded (label confetti): straight to the global table
    };

    if (x < W) {
        const uint64_t col = ((uint64_t)tile * H) * W + x;
        uint32_t cur = 0, cnt = 0, mn = 0xFFFFFFFFu, mx = 0, y0 = 0, y1 = 0;
        constexpr int kB = NYX_SCAN_BATCH;
        for (uint32_t yb = y_begin; yb < y_end; yb += kB) {
            uint32_t l[kB], v[kB];
#pragma unroll
            for (int k = 0; k < kB; k++) {               // all loads of the batch in flight before first use
                const uint32_t y = yb + k;
                const bool in = y < y_end;
                l[k] = in ? (uint32_t)label[col + (uint64_t)y * W] : 0u;
                v[k] = in ? (uint32_t)inten[col + (uint64_t)y * W] : 0u;
            }
#pragma unroll
            for (int k = 0; k < kB; k++) {
                const uint32_t y = yb + k;
                if (l[k] != cur) {
                    if (cur) flush(cur, cnt, mn, mx, y0, y1);
                    cur = l[k]; cnt = 0; mn = 0xFFFFFFFFu; mx = 0; y0 = y;
                }
                if (cur) { cnt++; mn = v[k] < mn ? v[k] : mn; mx = v[k] > mx ? v[k] : mx; y1 = y; }
            }
        }
        if (cur) flush(cur, cnt, mn, mx, y0, y1);
    }
    __syncthreads();
    for (int i = tid; i < kScanCap; i += kScanCols) {
        const uint32_t l = s_key[i];
        if (l == 0) continue;
        to_global(l, s_cnt[i], s_vmin[i], s_vmax[i], s_xmin[i], s_xmax[i], s_ymin[i], s_ymax[i]);
    }
}

// Occupied slots -> rows grouped by tile (slot order inside a tile), the tiles' row / pixel bases.  Two launches over
// 1024-entry blocks of the tables: per-block (row count, pixel count), then every block sums its predecessors' partials and
// places its own rows.
// meta[0] = n_roi, meta[1..2] = total pixels (lo, hi), meta[3] = max area, meta[4] = max bbox area, meta[5] = max range,
// meta[6] = max side, meta[7] = status (1 = a tile's table overflowed, 2 = a bounding box wider / taller than 65535),
// meta[8] = largest label value present.
__global__ __launch_bounds__(1024) void tile_block_sums_kernel(TileHash T, uint64_t n_entries, uint32_t* blk_rows, unsigned long long* blk_px)
{
    __shared__ uint32_t s_w[16];
    __shared__ unsigned long long s_wpx[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t e = blockIdx.x * 1024ull + tid;
    const uint32_t cnt = (e < n_entries && T.key[e] != 0) ? T.cnt[e] : 0;
    const uint32_t rows = (uint32_t)__popcll(__ballot(cnt != 0));
    const unsigned long long px = wave_sum_u64(cnt);
    if (lane == 0) { s_w[wave] = rows; s_wpx[wave] = px; }
    __syncthreads();
    if (tid == 0) {
        uint32_t r = 0; unsigned long long q = 0;
        for (int w = 0; w < 16; w++) { r += s_w[w]; q += s_wpx[w]; }
        blk_rows[blockIdx.x] = r; blk_px[blockIdx.x] = q;
    }
}

__global__ __launch_bounds__(1024) void tile_compact_kernel(TileHash T, uint64_t n_entries, uint32_t n_tiles, TileRows U, uint32_t max_rows, uint32_t* meta,
                                                            const uint32_t* blk_rows, const unsigned long long* blk_px, uint32_t* tile_row_begin,
                                                            unsigned long long* tile_px_begin)
{
    __shared__ uint32_t s_w[16];
    __shared__ unsigned long long s_wpx[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // an empty block (nine in ten at the usual load) only has bookkeeping to do when a tile starts in it or it is the last one
    const bool has_tile_start = ((blockIdx.x * 1024ull) % T.cap == 0) || T.cap < 1024;
    if (blk_rows[blockIdx.x] == 0 && !has_tile_start && blockIdx.x != gridDim.x - 1)
        return;
    // rows / pixels of all preceding blocks
    uint32_t pre_r = 0; unsigned long long pre_p = 0;
    for (uint32_t b = tid; b < blockIdx.x; b += 1024) { pre_r += blk_rows[b]; pre_p += blk_px[b]; }
    pre_r = (uint32_t)wave_sum_u64(pre_r); pre_p = wave_sum_u64(pre_p);
    if (lane == 0) { s_w[wave] = pre_r; s_wpx[wave] = pre_p; }
    __syncthreads();
    uint32_t base_r = 0; unsigned long long base_p = 0;
    for (int w = 0; w < 16; w++) { base_r += s_w[w]; base_p += s_wpx[w]; }
    __syncthreads();

    const uint64_t e = blockIdx.x * 1024ull + tid;
    const bool in = e < n_entries;
    const uint32_t cnt = (in && T.key[e] != 0) ? T.cnt[e] : 0;
    const bool present = cnt != 0;
    const unsigned long long bal = __ballot(present);
    const uint32_t rank_in_wave = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    unsigned long long pxs = cnt;                       // inclusive scan of pixel counts inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { unsigned long long o = __shfl_up(pxs, d, 64); if (lane >= d) pxs += o; }
    if (lane == 63) { s_w[wave] = (uint32_t)__popcll(bal); s_wpx[wave] = pxs; }
    __syncthreads();
    uint32_t wbase = 0; unsigned long long wpx = 0;
    for (int w2 = 0; w2 < wave; w2++) { wbase += s_w[w2]; wpx += s_wpx[w2]; }
    const uint32_t row = base_r + wbase + rank_in_wave;
    const unsigned long long off = base_p + wpx + pxs - cnt;
    if (in && (e & (T.cap - 1)) == 0) {                 // first slot of a tile: its rows / pixels start here
        const uint32_t t = (uint32_t)(e / T.cap);
        tile_row_begin[t] = row; tile_px_begin[t] = off;
    }
    uint32_t mx_area = 0, mx_box = 0, mx_rng = 0, mx_side = 0, mx_lab = 0;
    if (present && row < max_rows) {
        const uint32_t w = T.xmax[e] - T.xmin[e] + 1, h = T.ymax[e] - T.ymin[e] + 1;
        U.tile[row] = (uint32_t)(e / T.cap); U.label[row] = T.key[e]; U.area[row] = cnt;
        U.bbox_x0[row] = T.xmin[e]; U.bbox_y0[row] = T.ymin[e]; U.bbox_w[row] = w; U.bbox_h[row] = h;
        U.vmin[row] = T.vmin[e]; U.vmax[row] = T.vmax[e];
        const unsigned long long box = (unsigned long long)w * h;
        mx_area = cnt; mx_box = box > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)box; mx_rng = T.vmax[e] - T.vmin[e]; mx_side = w > h ? w : h;
        mx_lab = T.key[e];
    }
    mx_area = wave_max_u32_x(mx_area); mx_box = wave_max_u32_x(mx_box); mx_rng = wave_max_u32_x(mx_rng); mx_side = wave_max_u32_x(mx_side);
    mx_lab = wave_max_u32_x(mx_lab);
    // the launch extrema: waves -> workgroup through LDS, then five global atomics per workgroup (one set per WAVE put ~10^5
    // same-address atomics in a row on one L2 channel: 0.9 ms per 1000 tiles, as much as the whole label scan of 600 tiles)
    __shared__ uint32_t s_mx[5];
    __syncthreads();                                    // (s_w / s_wpx reads above are done; s_mx is a separate array anyway)
    if (tid < 5) s_mx[tid] = 0;
    __syncthreads();
    if (lane == 0 && bal) {
        atomicMax(&s_mx[0], mx_area); atomicMax(&s_mx[1], mx_box); atomicMax(&s_mx[2], mx_rng); atomicMax(&s_mx[3], mx_side); atomicMax(&s_mx[4], mx_lab);
    }
    __syncthreads();
    if (tid == 0 && s_mx[0] != 0) {                     // (an area of 0 means the workgroup holds no ROI)
        atomicMax(&meta[3], s_mx[0]); atomicMax(&meta[4], s_mx[1]); atomicMax(&meta[5], s_mx[2]); atomicMax(&meta[6], s_mx[3]); atomicMax(&meta[8], s_mx[4]);
        if (s_mx[3] > 65535u) atomicMax(&meta[7], 2u);  // coordinates inside a bounding box are 16-bit
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 1023) {   // last thread of the last block knows the totals
        const uint32_t n_roi = row + (present ? 1u : 0u);
        const unsigned long long tot = off + cnt;
        meta[0] = n_roi; meta[1] = (uint32_t)tot; meta[2] = (uint32_t)(tot >> 32);
        tile_row_begin[n_tiles] = n_roi; tile_px_begin[n_tiles] = tot;
    }
}

// Rows of one tile, slot order -> ascending label order, by counting (a tile of the benchmark holds ~200 ROIs: one 256-thread
// block and 200 LDS broadcast reads per thread; a slide with 10^5 ROIs spreads over grid.y blocks of 256 rows each).
// slide_mode: 0 = montage semantics of the in-memory API (slide min / max stay at +/- DBL_MAX, slideprops.cpp:27-28,74-75, so
// COVERED_IMAGE_INTENSITY_RANGE = range / -inf = -0.0); 1 = one tile is one slide: min / max over the intensities under any mask
// (scan_slide_props, slideprops.cpp:456-...) = extrema over the tile's ROIs; 2 = given per tile by the caller.
__global__ __launch_bounds__(256) void tile_rank_kernel(TileRows U, const uint32_t* __restrict__ tile_row_begin,
                                                        const unsigned long long* __restrict__ tile_px_begin, TileRows R, uint32_t max_rows,
                                                        int slide_mode, const double* __restrict__ smin_in, const double* __restrict__ smax_in)
{
    __shared__ uint32_t s_l[256], s_c[256], s_mn[256], s_mx[256];
    const int tid = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const uint32_t b = tile_row_begin[tile], e = tile_row_begin[tile + 1];
    const uint32_t n_t = (e > max_rows ? max_rows : e) - (b > max_rows ? max_rows : b);
    // (a tile with more than 65535 x 256 rows: the blocks of grid.y stride over its row blocks -- the grid's y extent is capped)
    for (uint32_t yb = blockIdx.y; yb * 256ull < n_t; yb += gridDim.y) {
    const uint32_t i = yb * 256u + tid;
    const bool mine = i < n_t;
    const uint32_t my_l = mine ? U.label[b + i] : 0xFFFFFFFFu;
    uint32_t rank = 0, mn = 0xFFFFFFFFu, mx = 0;
    unsigned long long off = 0;
    for (uint32_t j0 = 0; j0 < n_t; j0 += 256) {
        __syncthreads();
        const bool has = j0 + tid < n_t;
        s_l[tid] = has ? U.label[b + j0 + tid] : 0xFFFFFFFFu;
        s_c[tid] = has ? U.area[b + j0 + tid] : 0u;
        s_mn[tid] = has ? U.vmin[b + j0 + tid] : 0xFFFFFFFFu;
        s_mx[tid] = has ? U.vmax[b + j0 + tid] : 0u;
        __syncthreads();
        const uint32_t m = n_t - j0 < 256u ? n_t - j0 : 256u;
        for (uint32_t j = 0; j < m; j++) {
            const bool lt = s_l[j] < my_l;
            rank += lt ? 1u : 0u;
            off += lt ? s_c[j] : 0u;
            mn = s_mn[j] < mn ? s_mn[j] : mn;
            mx = s_mx[j] > mx ? s_mx[j] : mx;
        }
    }
    if (!mine)
        continue;
    const uint32_t src = b + i, row = b + rank;
    R.tile[row] = tile; R.label[row] = my_l; R.area[row] = U.area[src];
    R.px_offset[row] = tile_px_begin[tile] + off;
    R.bbox_x0[row] = U.bbox_x0[src]; R.bbox_y0[row] = U.bbox_y0[src]; R.bbox_w[row] = U.bbox_w[src]; R.bbox_h[row] = U.bbox_h[src];
    R.vmin[row] = U.vmin[src]; R.vmax[row] = U.vmax[src];
    const double DBL_MAX_ = 1.7976931348623157e308;
    R.slide_min[row] = slide_mode == 0 ? DBL_MAX_ : slide_mode == 1 ? (double)mn : smin_in[tile];
    R.slide_max[row] = slide_mode == 0 ? -DBL_MAX_ : slide_mode == 1 ? (double)mx : smax_in[tile];
    if (row + 1 == tile_row_begin[gridDim.x])           // last row of the stack closes the CSR offsets
        R.px_offset[row + 1] = tile_px_begin[gridDim.x];
    }
}

// One workgroup per ROI: bbox window of the tile -> SoA cloud in row-major order (deterministic).
template <typename TI, typename TL>
__global__ __launch_bounds__(256) void roi_cloud_kernel(const TI* __restrict__ inten, const TL* __restrict__ label, uint32_t W, uint32_t H, TileRows R,
                                                        uint16_t* cx, uint16_t* cy, uint32_t* cv)
{
    // Four 256-pixel chunks of the window per trip: every label / intensity load of a trip is issued before the first
    // ballot (the intensity unconditionally -- one HBM round trip per trip instead of two per chunk), and one barrier
    // pair ranks 1024 pixels.  (bx, by) advance incrementally: one integer division per thread and ROI.
    constexpr int U = 4;
    __shared__ uint32_t s_cnt[U * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t row = blockIdx.x;
    const uint32_t tile = R.tile[row], L = R.label[row];
    const uint32_t x0 = R.bbox_x0[row], w = R.bbox_w[row], h = R.bbox_h[row];
    const uint64_t y0 = R.bbox_y0[row] + (uint64_t)tile * H;            // y0 is tile-relative
    const uint64_t area = (uint64_t)w * h;
    unsigned long long out = R.px_offset[row];
    const uint32_t step_y = 256u / w, step_x = 256u - step_y * w;     // 256 pixels further along the window
    uint32_t by = (uint32_t)tid / w, bx = (uint32_t)tid - by * w;
    const TL* const lab_base = label + y0 * W + x0;
    const TI* const int_base = inten + y0 * W + x0;
    for (uint64_t p0 = 0; p0 < area; p0 += 256 * U) {
        uint32_t lb[U], v[U], xs[U], ys[U];
        bool hit[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t p = p0 + (uint32_t)u * 256u + (uint32_t)tid;
            xs[u] = bx; ys[u] = by;
            const uint64_t g = (uint64_t)by * W + bx;
            const bool ok = p < area;
            lb[u] = ok ? (uint32_t)lab_base[g] : 0u;                   // 0 is never a ROI label
            v[u] = ok ? (uint32_t)int_base[g] : 0u;
            bx += step_x; by += step_y;
            if (bx >= w) { bx -= w; by++; }
        }
        unsigned long long bal[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            hit[u] = lb[u] == L;
            bal[u] = __ballot(hit[u]);
            if (lane == 0) s_cnt[u * 4 + wave] = (uint32_t)__popcll(bal[u]);
        }
        __syncthreads();
        uint32_t run = 0;                               // hits of this trip before the current (chunk, wave), in window order
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) {
                if (w2 == wave && hit[u]) {
                    const unsigned long long o = out + run + (uint32_t)__popcll(bal[u] & ((1ull << lane) - 1ull));
                    cx[o] = (uint16_t)xs[u]; cy[o] = (uint16_t)ys[u]; cv[o] = v[u];
                }
                run += s_cnt[u * 4 + w2];
            }
        }
        out += run;
        __syncthreads();
    }
}

namespace {
// element-type dispatch: dt = 1 / 2 / 4 bytes per element
template <typename F>
int with_types(int dt_inten, int dt_label, F&& f)
{
#define NYX_TL(TI)                                                                  \
    switch (dt_label) {                                                             \
    case 1: return f((const TI*)nullptr, (const uint8_t*)nullptr);                  \
    case 2: return f((const TI*)nullptr, (const uint16_t*)nullptr);                 \
    case 4: return f((const TI*)nullptr, (const uint32_t*)nullptr);                 \
    default: return (int)hipErrorInvalidValue;                                      \
    }
    switch (dt_inten) {
    case 1: NYX_TL(uint8_t)
    case 2: NYX_TL(uint16_t)
    case 4: NYX_TL(uint32_t)
    default: return (int)hipErrorInvalidValue;
    }
#undef NYX_TL
}
} // namespace

int launch_tile_assembly_scan(const void* inten, int dt_inten, const void* label, int dt_label, uint32_t W, uint32_t H, uint32_t n_tiles,
                              TileHash T, TileRows U, TileRows R, uint32_t max_rows, uint32_t* meta, uint32_t* blk_rows, unsigned long long* blk_px,
                              uint32_t* tile_row_begin, unsigned long long* tile_px_begin, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const uint64_t n = (uint64_t)T.cap * n_tiles;
    hipLaunchKernelGGL(tile_init_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, T, n);
    const dim3 grid((W + kScanCols - 1) / kScanCols, (H + kScanRows - 1) / kScanRows, n_tiles);
    int rc = with_types(dt_inten, dt_label, [&](auto* ti, auto* tl) -> int {
        using TI = std::remove_const_t<std::remove_pointer_t<decltype(ti)>>;
        using TL = std::remove_const_t<std::remove_pointer_t<decltype(tl)>>;
        hipLaunchKernelGGL((tile_scan_kernel<TI, TL>), grid, dim3(kScanCols), 0, st, (const TI*)inten, (const TL*)label, W, H, n_tiles, T, meta);
        return 0;
    });
    if (rc) return rc;
    const unsigned nb = (unsigned)((n + 1023) / 1024);
    hipLaunchKernelGGL(tile_block_sums_kernel, dim3(nb), dim3(1024), 0, st, T, n, blk_rows, blk_px);
    hipLaunchKernelGGL(tile_compact_kernel, dim3(nb), dim3(1024), 0, st, T, n, n_tiles, U, max_rows, meta, blk_rows, blk_px, tile_row_begin, tile_px_begin);
    return (int)hipGetLastError();
}

int launch_tile_rank(TileRows U, const uint32_t* tile_row_begin, const unsigned long long* tile_px_begin, TileRows R, uint32_t max_rows, uint32_t n_tiles,
                     uint32_t max_rows_per_tile, int slide_mode, const double* smin, const double* smax, void* stream)
{
    unsigned gy = (max_rows_per_tile + 255) / 256;
    if (gy > 65535u) gy = 65535u;                          // HIP's limit on grid.y; the kernel strides
    hipLaunchKernelGGL(tile_rank_kernel, dim3(n_tiles, gy ? gy : 1), dim3(256), 0, (hipStream_t)stream, U, tile_row_begin, tile_px_begin, R, max_rows,
                       slide_mode, smin, smax);
    return (int)hipGetLastError();
}

int launch_tile_clouds(const void* inten, int dt_inten, const void* label, int dt_label, uint32_t W, uint32_t H, TileRows R, uint32_t n_roi,
                       uint16_t* cx, uint16_t* cy, uint32_t* cv, void* stream)
{
    if (n_roi == 0) return 0;
    int rc = with_types(dt_inten, dt_label, [&](auto* ti, auto* tl) -> int {
        using TI = std::remove_const_t<std::remove_pointer_t<decltype(ti)>>;
        using TL = std::remove_const_t<std::remove_pointer_t<decltype(tl)>>;
        hipLaunchKernelGGL((roi_cloud_kernel<TI, TL>), dim3(n_roi), dim3(256), 0, (hipStream_t)stream, (const TI*)inten, (const TL*)label, W, H, R, cx, cy, cv);
        return 0;
    });
    return rc ? rc : (int)hipGetLastError();
}

} // namespace nyxhip
