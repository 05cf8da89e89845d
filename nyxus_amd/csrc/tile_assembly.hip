// tile_assembly.hip -- phases 1-2 of the in-memory workflow on the device.
//
// Replaces, for one intensity/label tile pair resident in HBM:
//   gatherRoisMetricsInMemory  /root/reference/src/nyx/phase1.cpp:373-409  + feed_pixel_2_metrics
//                              src/nyx/pixel_feed.cpp:19-43   -> per-label area, min, max, AABB
//   scanTrivialRoisInMemory    src/nyx/phase2_2d.cpp:637-684  -> per-ROI pixel clouds
// The reference does both with a serial scan and a hash-map lookup per pixel; here:
//   tile_scan_kernel     one coalesced pass over the tile in 256 x 32 blocks; per-column label runs are
//                        accumulated in registers, merged per block in an LDS hash table, and only the
//                        table's live entries reach the global [max_label+1] tables;
//   tile_compact_kernel  labels present -> rows in ascending label order (the row order of
//                        save_features_2_buffer, output_2_buffer.cpp:305-306), CSR offsets by prefix sum;
//   roi_cloud_kernel     one workgroup per ROI scans its bounding-box window of the tile in row-major
//                        order and writes the ROI's SoA cloud with a ballot-ranked (deterministic)
//                        compaction.
// HBM traffic per tile: 8 B/px read by the scan + the bbox windows (L2-resident re-read) + 8 B per ROI
// pixel written as clouds.
#include <hip/hip_runtime.h>
#include "device_math.h"
#include "roi_kernel.h"

namespace nyxhip {

__global__ void tile_init_tables_kernel(TileTables T, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        T.cnt[i] = 0; T.vmin[i] = 0xFFFFFFFFu; T.vmax[i] = 0;
        T.xmin[i] = 0xFFFFFFFFu; T.xmax[i] = 0; T.ymin[i] = 0xFFFFFFFFu; T.ymax[i] = 0;
    }
}

__device__ __forceinline__ uint32_t wave_min_u32_masked(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32_x(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// The input may be a stack of n_tiles tiles of H rows each (one tall image); labels are per tile, so the
// table index is tile * (max_label + 1) + label and coordinates are tile-relative.
//
// One workgroup owns a kScanCols x kScanRows block of one tile; a thread walks one column of the block and keeps
// the statistics of its current label run in registers (ROIs are compact: a column crosses a ROI once).  Runs
// are merged in an LDS hash table keyed by label, and only the table's few live entries go to the global
// [n_tiles * (max_label + 1)] tables: ~7 global atomics per (ROI, block) instead of per (ROI, row segment).
constexpr int kScanCols = 256, kScanRows = 32, kScanCap = 512, kScanProbes = 32;

__global__ __launch_bounds__(kScanCols) void tile_scan_kernel(const uint32_t* __restrict__ inten, const uint32_t* __restrict__ label,
                                                              uint32_t W, uint32_t H, uint32_t n_tiles, uint32_t max_label, TileTables T, int* status)
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
    const uint32_t tbase = tile * (max_label + 1);

    auto flush = [&](uint32_t l, uint32_t cnt, uint32_t mn, uint32_t mx, uint32_t y0, uint32_t y1) {
        if (l > max_label) { atomicCAS(status, 0, 1 /* NYXHIP_ERR_INVALID_ARG */); return; }
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
        const uint32_t g = tbase + l;                    // table crowded (label confetti): straight to the global tables
        atomicAdd(&T.cnt[g], cnt);
        atomicMin(&T.vmin[g], mn); atomicMax(&T.vmax[g], mx);
        atomicMin(&T.xmin[g], x); atomicMax(&T.xmax[g], x);
        atomicMin(&T.ymin[g], y0); atomicMax(&T.ymax[g], y1);
    };

    if (x < W) {
        const uint64_t col = ((uint64_t)tile * H) * W + x;
        uint32_t cur = 0, cnt = 0, mn = 0xFFFFFFFFu, mx = 0, y0 = 0, y1 = 0;
        for (uint32_t yb = y_begin; yb < y_end; yb += 8) {
            uint32_t l[8], v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {                // all loads of the batch in flight before first use
                const uint32_t y = yb + k;
                const bool in = y < y_end;
                l[k] = in ? label[col + (uint64_t)y * W] : 0u;
                v[k] = in ? inten[col + (uint64_t)y * W] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
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
        const uint32_t g = tbase + l;
        atomicAdd(&T.cnt[g], s_cnt[i]);
        atomicMin(&T.vmin[g], s_vmin[i]); atomicMax(&T.vmax[g], s_vmax[i]);
        atomicMin(&T.xmin[g], s_xmin[i]); atomicMax(&T.xmax[g], s_xmax[i]);
        atomicMin(&T.ymin[g], s_ymin[i]); atomicMax(&T.ymax[g], s_ymax[i]);
    }
}

// Labels present -> rows in ascending (tile, label) order; rows' CSR offsets.  Two launches over 1024-entry
// blocks of the tables: per-block (row count, pixel count), then every block sums its predecessors' partials
// (a few hundred values) and places its own rows.
// meta[0] = n_roi, meta[1..2] = total pixels (lo, hi), meta[3] = max area, meta[4] = max bbox area,
// meta[5] = max range, meta[6] = max side, meta[7] = scan status.
__global__ __launch_bounds__(1024) void tile_block_sums_kernel(TileTables T, uint32_t n_entries, uint32_t* blk_rows, unsigned long long* blk_px)
{
    __shared__ uint32_t s_w[16];
    __shared__ unsigned long long s_wpx[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t l = blockIdx.x * 1024u + tid;
    const uint32_t cnt = (l >= 1 && l < n_entries) ? T.cnt[l] : 0;
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

__global__ __launch_bounds__(1024) void tile_compact_kernel(TileTables T, uint32_t n_entries, TileRows R, uint32_t max_rows, uint32_t* meta,
                                                            const uint32_t* blk_rows, const unsigned long long* blk_px, const int* status)
{
    __shared__ uint32_t s_w[16];
    __shared__ unsigned long long s_wpx[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // rows / pixels of all preceding blocks
    uint32_t pre_r = 0; unsigned long long pre_p = 0;
    for (uint32_t b = tid; b < blockIdx.x; b += 1024) { pre_r += blk_rows[b]; pre_p += blk_px[b]; }
    pre_r = (uint32_t)wave_sum_u64(pre_r); pre_p = wave_sum_u64(pre_p);
    if (lane == 0) { s_w[wave] = pre_r; s_wpx[wave] = pre_p; }
    __syncthreads();
    uint32_t base_r = 0; unsigned long long base_p = 0;
    for (int w = 0; w < 16; w++) { base_r += s_w[w]; base_p += s_wpx[w]; }
    __syncthreads();

    const uint32_t l = blockIdx.x * 1024u + tid;
    const uint32_t cnt = (l >= 1 && l < n_entries) ? T.cnt[l] : 0;
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
    uint32_t mx_area = 0, mx_box = 0, mx_rng = 0, mx_side = 0;
    if (present && row < max_rows) {
        const uint32_t w = T.xmax[l] - T.xmin[l] + 1, h = T.ymax[l] - T.ymin[l] + 1;
        R.label[row] = l; R.px_offset[row] = off;
        R.bbox_x0[row] = T.xmin[l]; R.bbox_y0[row] = T.ymin[l]; R.bbox_w[row] = w; R.bbox_h[row] = h;
        R.vmin[row] = T.vmin[l]; R.vmax[row] = T.vmax[l];
        mx_area = cnt; mx_box = w * h; mx_rng = T.vmax[l] - T.vmin[l]; mx_side = w > h ? w : h;
    }
    mx_area = wave_max_u32_x(mx_area); mx_box = wave_max_u32_x(mx_box); mx_rng = wave_max_u32_x(mx_rng); mx_side = wave_max_u32_x(mx_side);
    if (lane == 0 && bal) { atomicMax(&meta[3], mx_area); atomicMax(&meta[4], mx_box); atomicMax(&meta[5], mx_rng); atomicMax(&meta[6], mx_side); }
    if (blockIdx.x == gridDim.x - 1 && tid == 1023) {   // last thread of the last block knows the totals
        const uint32_t n_roi = row + (present ? 1u : 0u);
        const unsigned long long tot = off + cnt;
        meta[0] = n_roi; meta[1] = (uint32_t)tot; meta[2] = (uint32_t)(tot >> 32); meta[7] = (uint32_t)*status;
        if (n_roi <= max_rows) R.px_offset[n_roi] = tot;
    }
}

// One workgroup per ROI: bbox window of the tile -> SoA cloud in row-major order (deterministic).
__global__ __launch_bounds__(256) void roi_cloud_kernel(const uint32_t* __restrict__ inten, const uint32_t* __restrict__ label, uint32_t W,
                                                        uint32_t H, uint32_t stride, TileRows R, uint16_t* cx, uint16_t* cy, uint32_t* cv)
{
    // Four 256-pixel chunks of the window per trip: every label / intensity load of a trip is issued before the first
    // ballot (the intensity unconditionally -- one HBM round trip per trip instead of two per chunk), and one barrier
    // pair ranks 1024 pixels.  (bx, by) advance incrementally: one integer division per thread and ROI.
    constexpr int U = 4;
    __shared__ uint32_t s_cnt[U * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t row = blockIdx.x;
    // R.label holds the table index tile * stride + label; y0 is tile-relative
    const uint32_t key = R.label[row], tile = key / stride, L = key - tile * stride;
    const uint32_t x0 = R.bbox_x0[row], y0 = R.bbox_y0[row] + tile * H, w = R.bbox_w[row], h = R.bbox_h[row];
    const uint32_t area = w * h;
    unsigned long long out = R.px_offset[row];
    const uint32_t step_y = 256u / w, step_x = 256u - step_y * w;     // 256 pixels further along the window
    uint32_t by = (uint32_t)tid / w, bx = (uint32_t)tid - by * w;
    const uint32_t* const lab_base = label + (uint64_t)y0 * W + x0;
    const uint32_t* const int_base = inten + (uint64_t)y0 * W + x0;
    for (uint32_t p0 = 0; p0 < area; p0 += 256 * U) {
        uint32_t lb[U], v[U], xs[U], ys[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t p = p0 + (uint32_t)u * 256u + (uint32_t)tid;
            xs[u] = bx; ys[u] = by;
            const uint64_t g = (uint64_t)by * W + bx;
            const bool ok = p < area;
            lb[u] = ok ? lab_base[g] : ~L;
            v[u] = ok ? int_base[g] : 0u;
            bx += step_x; by += step_y;
            if (bx >= w) { bx -= w; by++; }
        }
        unsigned long long bal[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            bal[u] = __ballot(lb[u] == L);
            if (lane == 0) s_cnt[u * 4 + wave] = (uint32_t)__popcll(bal[u]);
        }
        __syncthreads();
        uint32_t run = 0;                               // hits of this trip before the current (chunk, wave), in window order
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) {
                if (w2 == wave && lb[u] == L) {
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

// table key -> (label, tile index) for the caller; also the per-ROI slide extrema of the in-memory (montage)
// path: its prescan leaves slide min / max at +DBL_MAX / -DBL_MAX (slideprops.cpp:27-28,74-75), so
// COVERED_IMAGE_INTENSITY_RANGE = range / -inf = -0.0
__global__ void tile_split_keys_kernel(const uint32_t* key, uint32_t stride, uint32_t n, uint32_t* out_label, uint32_t* out_tile, double* slide_min,
                                       double* slide_max)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint32_t k = key[i], t = k / stride;
        out_label[i] = k - t * stride;
        if (out_tile) out_tile[i] = t;
        slide_min[i] = 1.7976931348623157e308;
        slide_max[i] = -1.7976931348623157e308;
    }
}

int launch_tile_assembly_scan(const uint32_t* inten, const uint32_t* label, uint32_t W, uint32_t H, uint32_t n_tiles, uint32_t max_label,
                              TileTables T, TileRows R, uint32_t max_rows, uint32_t* meta, uint32_t* blk_rows, unsigned long long* blk_px, int* status,
                              void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const uint32_t n = (max_label + 1) * n_tiles;
    hipLaunchKernelGGL(tile_init_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, st, T, n);
    const dim3 grid((W + kScanCols - 1) / kScanCols, (H + kScanRows - 1) / kScanRows, n_tiles);
    hipLaunchKernelGGL(tile_scan_kernel, grid, dim3(kScanCols), 0, st, inten, label, W, H, n_tiles, max_label, T, status);
    const unsigned nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(tile_block_sums_kernel, dim3(nb), dim3(1024), 0, st, T, n, blk_rows, blk_px);
    hipLaunchKernelGGL(tile_compact_kernel, dim3(nb), dim3(1024), 0, st, T, n, R, max_rows, meta, blk_rows, blk_px, status);
    return (int)hipGetLastError();
}

int launch_tile_clouds(const uint32_t* inten, const uint32_t* label, uint32_t W, uint32_t H, uint32_t stride, TileRows R, uint32_t n_roi,
                       uint16_t* cx, uint16_t* cy, uint32_t* cv, void* stream)
{
    if (n_roi == 0) return 0;
    hipLaunchKernelGGL(roi_cloud_kernel, dim3(n_roi), dim3(256), 0, (hipStream_t)stream, inten, label, W, H, stride, R, cx, cy, cv);
    return (int)hipGetLastError();
}

} // namespace nyxhip
