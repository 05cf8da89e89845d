// tile_assembly.hip -- phases 1-2 of the in-memory workflow on the device.
//
// Replaces, for one intensity/label tile pair resident in HBM:
//   gatherRoisMetricsInMemory  /root/reference/src/nyx/phase1.cpp:373-409  + feed_pixel_2_metrics
//                              src/nyx/pixel_feed.cpp:19-43   -> per-label area, min, max, AABB
//   scanTrivialRoisInMemory    src/nyx/phase2_2d.cpp:637-684  -> per-ROI pixel clouds
// The reference does both with a serial scan and a hash-map lookup per pixel; here:
//   tile_scan_kernel     one coalesced pass over the tile; a wave aggregates the pixels that share a
//                        label (ROIs are spatially compact, so usually one group per wave) and issues
//                        one atomic per statistic per group into the [max_label+1] tables;
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
__global__ __launch_bounds__(256) void tile_scan_kernel(const uint32_t* __restrict__ inten, const uint32_t* __restrict__ label,
                                                        uint32_t W, uint32_t H, uint32_t n_tiles, uint32_t max_label, TileTables T, int* status)
{
    const uint64_t npx = (uint64_t)W * H * n_tiles;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < npx; base += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = base + threadIdx.x;
        uint32_t l = 0, v = 0, x = 0, y = 0;
        if (p < npx) {
            l = label[p];
            if (l != 0) {
                v = inten[p];
                const uint64_t yy = p / W;
                x = (uint32_t)(p - yy * W);
                const uint32_t t = (uint32_t)(yy / H);
                y = (uint32_t)(yy - (uint64_t)t * H);
                if (l > max_label) { atomicCAS(status, 0, 1 /* NYXHIP_ERR_INVALID_ARG */); l = 0; }
                else l += t * (max_label + 1);
            }
        }
        unsigned long long todo = __ballot(l != 0);
        while (todo) {                                   // one iteration per distinct label in the wave
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t ll = __shfl(l, leader, 64);
            const bool mine = l == ll;
            const unsigned long long grp = __ballot(mine);
            const uint32_t c = (uint32_t)__popcll(grp);
            uint32_t mn = wave_min_u32_masked(mine ? v : 0xFFFFFFFFu), mx = wave_max_u32_x(mine ? v : 0u);
            uint32_t x0 = wave_min_u32_masked(mine ? x : 0xFFFFFFFFu), x1 = wave_max_u32_x(mine ? x : 0u);
            uint32_t y0 = wave_min_u32_masked(mine ? y : 0xFFFFFFFFu), y1 = wave_max_u32_x(mine ? y : 0u);
            if (lane == leader) {
                atomicAdd(&T.cnt[ll], c);
                atomicMin(&T.vmin[ll], mn); atomicMax(&T.vmax[ll], mx);
                atomicMin(&T.xmin[ll], x0); atomicMax(&T.xmax[ll], x1);
                atomicMin(&T.ymin[ll], y0); atomicMax(&T.ymax[ll], y1);
            }
            todo &= ~grp;
        }
    }
}

// One workgroup: ascending labels -> rows; rows' CSR offsets.  meta[0] = n_roi, meta[1..2] = total pixels
// (lo, hi), meta[3] = max area, meta[4] = max bbox area, meta[5] = max range, meta[6] = max side.
__global__ __launch_bounds__(1024) void tile_compact_kernel(TileTables T, uint32_t n_entries, TileRows R, uint32_t max_rows, uint32_t* meta)
{
    const uint32_t max_label = n_entries - 1;   // entry 0 (label 0 of tile 0) is never populated; other tiles' label 0 neither
    __shared__ uint32_t s_w[16];
    __shared__ unsigned long long s_wpx[16];
    __shared__ uint32_t s_base;
    __shared__ unsigned long long s_pxbase;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_base = 0; s_pxbase = 0; }
    uint32_t mx_area = 0, mx_box = 0, mx_rng = 0, mx_side = 0;
    __syncthreads();
    for (uint32_t c0 = 1; c0 <= max_label; c0 += 1024) {
        const uint32_t l = c0 + tid;
        const uint32_t cnt = l <= max_label ? T.cnt[l] : 0;
        const bool present = cnt != 0;
        const unsigned long long bal = __ballot(present);
        const uint32_t rank_in_wave = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        // inclusive scan of pixel counts inside the wave
        unsigned long long pxs = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { unsigned long long o = __shfl_up(pxs, d, 64); if (lane >= d) pxs += o; }
        if (lane == 63) { s_w[wave] = (uint32_t)__popcll(bal); s_wpx[wave] = pxs; }
        __syncthreads();
        uint32_t wbase = 0; unsigned long long wpx = 0;
        for (int w2 = 0; w2 < wave; w2++) { wbase += s_w[w2]; wpx += s_wpx[w2]; }
        const uint32_t row = s_base + wbase + rank_in_wave;
        const unsigned long long off = s_pxbase + wpx + pxs - cnt;
        if (present && row < max_rows) {
            const uint32_t w = T.xmax[l] - T.xmin[l] + 1, h = T.ymax[l] - T.ymin[l] + 1;
            R.label[row] = l; R.px_offset[row] = off;
            R.bbox_x0[row] = T.xmin[l]; R.bbox_y0[row] = T.ymin[l]; R.bbox_w[row] = w; R.bbox_h[row] = h;
            R.vmin[row] = T.vmin[l]; R.vmax[row] = T.vmax[l];
            mx_area = cnt > mx_area ? cnt : mx_area;
            mx_box = w * h > mx_box ? w * h : mx_box;
            mx_rng = T.vmax[l] - T.vmin[l] > mx_rng ? T.vmax[l] - T.vmin[l] : mx_rng;
            mx_side = (w > h ? w : h) > mx_side ? (w > h ? w : h) : mx_side;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0; unsigned long long tpx = 0;
            for (int w2 = 0; w2 < 16; w2++) { tot += s_w[w2]; tpx += s_wpx[w2]; }
            s_base += tot; s_pxbase += tpx;
        }
        __syncthreads();
    }
    mx_area = wave_max_u32_x(mx_area); mx_box = wave_max_u32_x(mx_box); mx_rng = wave_max_u32_x(mx_rng); mx_side = wave_max_u32_x(mx_side);
    if (lane == 0) { atomicMax(&meta[3], mx_area); atomicMax(&meta[4], mx_box); atomicMax(&meta[5], mx_rng); atomicMax(&meta[6], mx_side); }
    if (tid == 0) {
        meta[0] = s_base; meta[1] = (uint32_t)s_pxbase; meta[2] = (uint32_t)(s_pxbase >> 32);
        if (s_base <= max_rows) R.px_offset[s_base] = s_pxbase;
    }
}

// One workgroup per ROI: bbox window of the tile -> SoA cloud in row-major order (deterministic).
__global__ __launch_bounds__(256) void roi_cloud_kernel(const uint32_t* __restrict__ inten, const uint32_t* __restrict__ label, uint32_t W,
                                                        uint32_t H, uint32_t stride, TileRows R, uint16_t* cx, uint16_t* cy, uint32_t* cv)
{
    __shared__ uint32_t s_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t row = blockIdx.x;
    // R.label holds the table index tile * stride + label; y0 is tile-relative
    const uint32_t key = R.label[row], tile = key / stride, L = key - tile * stride;
    const uint32_t x0 = R.bbox_x0[row], y0 = R.bbox_y0[row] + tile * H, w = R.bbox_w[row], h = R.bbox_h[row];
    const uint32_t area = w * h;
    unsigned long long out = R.px_offset[row];
    for (uint32_t p0 = 0; p0 < area; p0 += 256) {
        const uint32_t p = p0 + tid;
        bool hit = false;
        uint32_t bx = 0, by = 0, v = 0;
        if (p < area) {
            by = p / w; bx = p - by * w;
            const uint64_t g = (uint64_t)(y0 + by) * W + (x0 + bx);
            hit = label[g] == L;
            if (hit) v = inten[g];
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w2 = 0; w2 < 4; w2++) { if (w2 < wave) before += s_cnt[w2]; total += s_cnt[w2]; }
        if (hit) {
            const unsigned long long o = out + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            cx[o] = (uint16_t)bx; cy[o] = (uint16_t)by; cv[o] = v;
        }
        out += total;
        __syncthreads();
    }
}

// table key -> (label, tile index) for the caller
__global__ void tile_split_keys_kernel(const uint32_t* key, uint32_t stride, uint32_t n, uint32_t* out_label, uint32_t* out_tile)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint32_t k = key[i], t = k / stride;
        out_label[i] = k - t * stride;
        if (out_tile) out_tile[i] = t;
    }
}

int launch_tile_assembly_scan(const uint32_t* inten, const uint32_t* label, uint32_t W, uint32_t H, uint32_t n_tiles, uint32_t max_label,
                              TileTables T, TileRows R, uint32_t max_rows, uint32_t* meta, int* status, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const uint32_t n = (max_label + 1) * n_tiles;
    hipLaunchKernelGGL(tile_init_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, st, T, n);
    const uint64_t npx = (uint64_t)W * H * n_tiles;
    unsigned blocks = (unsigned)((npx + 255) / 256);
    if (blocks > 256 * 32) blocks = 256 * 32;           // grid-stride: ~16 workgroups per CU
    hipLaunchKernelGGL(tile_scan_kernel, dim3(blocks), dim3(256), 0, st, inten, label, W, H, n_tiles, max_label, T, status);
    hipLaunchKernelGGL(tile_compact_kernel, dim3(1), dim3(1024), 0, st, T, n, R, max_rows, meta);
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
