// gabor_mfma_probe.hip -- the low-precision MFMA screening pass for the Gabor filter bank, measured (round-4 review, item 5).
//
// What it times: for R ROIs of 61 x 61 twelve-bit pixels, the responses (re, im) of FOUR 16 x 16 complex band-pass filters at
// every pixel -- the contraction  out[pixel][2 f + c] = sum_{j,i} image[y + j][x + i] * bank[f][j][i][c]  that the product's
// roi_gabor_tiled_kernel (MODE 3) runs as packed-fp32 FMAs on the vector ALU -- on the matrix pipe:
//   * the image as two 6-bit digits (v = 64 d1 + d0), each an exact f16; the taps, scaled by 2^10, as f16 hi + lo parts
//     (2^-22 relative).  The tile's 16 columns are the 8 real columns (4 filters x re / im) x {hi, lo}: one
//     v_mfma_f32_16x16x32_f16 per digit covers TWO tap rows (K = 32) and both tap parts; fp32 accumulation; the two halves and the
//     two digits are combined in the epilogue;
//   * M = 16 ROWS at one column x, K = horizontal taps: a lane's eight K-elements are eight consecutive pixels of a plane row, whose
//     misalignment (x mod 4) is the same in every lane -- a window of twelve pixels is read once (three 8-byte LDS reads) and
//     serves four columns x, odd shifts through v_perm;
//   * the bank lives in registers (8 tap-row pairs x 4 VGPRs), the padded digit planes in LDS (27 KB per workgroup);
//   * epilogue per tile: energy = re^2 + im^2 against a threshold, counted per lane (what the screening pass needs).
// Measured (MI355X, 16384 ROIs, scaled to 196 000): 16x16x16 with hi / lo as separate instructions 27.5 ms; hi / lo as columns 26.9 ms
// (SQ_INSTS_MFMA 2048, SQ_VALU_MFMA_BUSY_CYCLES 32768 = 16 cycles per instruction, SQ_INSTS_VALU 8179 per wave: operand assembly
// bound); 16x16x32 21.4 ms at pitch 80 / 88 (LDS bank conflicts), **15.2 ms at pitch 84** (616 -> 865 TFLOP/s of executed f16 MFMA).
// Response error 1.0e-8 a_max (the product's packed-fp32 screening: 1.54e-5 a_max per component), counts identical to fp64.
// The product spends ~27 of its 33.9 ms on the same four filters in v_pk_fma_f32: an MFMA screening stage is worth ~1.8 x on them.
// It checks itself against an fp64 evaluation of the same sums on the host (max error relative to a_max: the screening bound) and
// prints ms per 196 000 ROIs next to the product's 33.9 ms for the whole Gabor kernel (4 band-pass + low-pass, decisions, redo).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/gabor_mfma_probe tools/gabor_mfma_probe.hip && /tmp/gabor_mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef PITCH
#define PITCH 84     // f16 elements per plane row: 4 mod 8 -- the lanes' 8-byte window reads then spread over the banks (80 / 88 / 96: 21 / 21 / 36 ms; 84 / 92 / 100: 15.2 ms)
#endif
constexpr int kW = 61, kH = 61, kN = 16, kPitch = PITCH, kRows = 64 + kN;          // padded plane: 80 rows x 88 columns of f16 per digit (every window of a 64 x 64 output grid stays inside)
constexpr int kPlane = kRows * kPitch;                                           // f16 elements per digit plane
constexpr float kTapScale = 1024.0f;

__device__ __forceinline__ uint32_t alignbit16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }

// img: [R][61 * 61] u16; bank: [4 filters][16][16][2] float; thr2: squared energy threshold; counts: [R][4]
__global__ __launch_bounds__(256) void gabor_mfma_kernel(const uint16_t* __restrict__ img, const h8* __restrict__ bank_ops /* [8][64]: B operand of the tap-row pair jp, lane l */, float thr2,
                                                         uint32_t* __restrict__ counts, float* __restrict__ resp_out /* [61*61][8] of ROI 0, or null */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    _Float16* const plane = (_Float16*)lds_raw;                                  // [2 digits][kRows][kPitch]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t* const src = img + (size_t)blockIdx.x * (kW * kH);
    for (int i = tid; i < 2 * kPlane / 2; i += 256) ((uint32_t*)plane)[i] = 0;
    __syncthreads();
    for (int y = wave; y < kH; y += 4) {                                         // a row per wave and trip, lane = column
        if (lane < kW) {
            const uint32_t v = src[y * kW + lane];
            plane[y * kPitch + lane] = (_Float16)(float)(v >> 6);                // digit 1 (rows / columns beyond the box stay zero: the taps that reach them add 0)
            plane[kPlane + y * kPitch + lane] = (_Float16)(float)(v & 63u);       // digit 0
        }
    }
    // the bank as B operands: lane (n = lane % 16, kb = lane / 16) holds taps [j][i = 4 kb .. 4 kb + 3] of column n = 2 f + c (columns 8 .. 15: zero)
    // the bank as B operands, laid out by the host: columns 0 .. 7 the hi parts of (filter, re / im), columns 8 .. 15 the lo parts of
    // the same taps -- the tile's sixteen columns are all useful and ONE instruction covers both parts (added in the epilogue)
    h8 Bw[8];
#pragma unroll
    for (int jp = 0; jp < 8; jp++) Bw[jp] = bank_ops[jp * 64 + lane];
    __syncthreads();
    // wave = row tile (rows 16 wave .. 16 wave + 15); lane (m = lane % 16, kb = lane / 16)
    const int m = lane & 15, kb = lane >> 4;
    const int y0 = 16 * wave;
    const _Float16* const lane_base = plane + (y0 + m + (kb >> 1)) * kPitch + 8 * (kb & 1);   // the lane's window origin for tap-row pair 0, digit 1, x = 0
    uint32_t lane_cnt = 0;
    const float unscale = 1.0f / kTapScale;
    for (int x4 = 0; x4 < 16; x4++) {                                            // four columns x = 4 x4 + xs per trip
        f4 C[4][2];
#pragma unroll
        for (int xs = 0; xs < 4; xs++) { C[xs][0] = f4{0, 0, 0, 0}; C[xs][1] = f4{0, 0, 0, 0}; }
        // v_mfma_f32_16x16x32_f16: K = 32 = the 16 taps of tap rows 2 jp and 2 jp + 1; lane (m, kb): k = 8 kb .. 8 kb + 7, i.e. tap row
        // 2 jp + (kb >> 1), taps 8 (kb & 1) .. + 7 -- eight consecutive pixels of one plane row, read as a twelve-pixel window (three
        // 8-byte LDS reads at a fixed offset from the lane's base) that serves the four columns x = 4 x4 + xs
#pragma unroll
        for (int jp = 0; jp < 8; jp++) {                                       // (fully unrolled: the bank's registers need static names)
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const _Float16* const rowp = lane_base + d * kPlane + (2 * jp) * kPitch + 4 * x4;
                const uint2 q0 = *(const uint2*)rowp, q1 = *(const uint2*)(rowp + 4), q2 = *(const uint2*)(rowp + 8);
                const uint32_t w0 = q0.x, w1 = q0.y, w2 = q1.x, w3 = q1.y, w4 = q2.x, w5 = q2.y;
                uint4 a[4];
                a[0] = uint4{w0, w1, w2, w3};
                a[1] = uint4{alignbit16(w1, w0), alignbit16(w2, w1), alignbit16(w3, w2), alignbit16(w4, w3)};
                a[2] = uint4{w1, w2, w3, w4};
                a[3] = uint4{alignbit16(w2, w1), alignbit16(w3, w2), alignbit16(w4, w3), alignbit16(w5, w4)};
#pragma unroll
                for (int xs = 0; xs < 4; xs++) {
                    h8 A;
                    __builtin_memcpy(&A, &a[xs], 16);
                    C[xs][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bw[jp], C[xs][d], 0, 0, 0);
                }
            }
        }
        // epilogue: lane (n = lane % 16, rows 4 (lane / 16) + r): response = (64 C[high digit] + C[low digit]) / 2^10; energy^2 of filter n / 2 = re^2 + im^2
#pragma unroll
        for (int xs = 0; xs < 4; xs++) {
            const int x = 4 * x4 + xs;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float resp = (64.0f * C[xs][0][r] + C[xs][1][r]) * unscale;          // (d = 0 is the HIGH digit's plane)
                resp += __shfl_xor(resp, 8, 64);                                        // hi-part column n + lo-part column n + 8
                const int y = y0 + 4 * (lane >> 4) + r;
                if (resp_out && blockIdx.x == 0 && (lane & 15) < 8 && y < kH && x < kW) resp_out[(y * kW + x) * 8 + (lane & 15)] = resp;
                const float sq = resp * resp;
                const float e2 = sq + __shfl_xor(sq, 1, 64);                     // re (even column) + im (odd column)
                lane_cnt += (y < kH && x < kW && e2 > thr2) ? 1u : 0u;             // (a counter per lane; lanes n = 0, 2, 4, 6 of every 16-lane row carry filters 0 .. 3)
            }
        }
    }
    {
        uint32_t c = ((lane & 1) == 0 && (lane & 15) < 8) ? lane_cnt : 0u;
        c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);                    // the four 16-lane rows hold the same columns
        if (lane < 8 && (lane & 1) == 0) atomicAdd(&counts[(size_t)blockIdx.x * 4 + (lane >> 1)], c);
    }
}

int main()
{
    const int R = 16384, iters = 5;
    std::vector<uint16_t> h_img((size_t)R * kW * kH);
    srand(7);
    for (auto& v : h_img) v = (uint16_t)(rand() % 4095 + 1);
    // four complex band-pass filters with wide envelopes, L1-normalised per filter (sum |g| = 1), like the reference's bank
    std::vector<float> h_bank(4 * 256 * 2);
    for (int f = 0; f < 4; f++) {
        double tot = 0;
        std::vector<double> g(512);
        for (int j = 0; j < 16; j++)
            for (int i = 0; i < 16; i++) {
                const double xx = i - 7.5, yy = j - 7.5, th = M_PI * (f + 1) / 5.0, f0 = 0.2 + 0.1 * f;
                const double xr = xx * cos(th) + yy * sin(th), yr = -xx * sin(th) + yy * cos(th);
                const double env = exp(-(xr * xr + 0.5 * yr * yr) / 40.0);
                g[(j * 16 + i) * 2] = env * cos(2 * M_PI * f0 * xr);
                g[(j * 16 + i) * 2 + 1] = env * sin(2 * M_PI * f0 * xr);
                tot += sqrt(g[(j * 16 + i) * 2] * g[(j * 16 + i) * 2] + g[(j * 16 + i) * 2 + 1] * g[(j * 16 + i) * 2 + 1]);
            }
        for (int k = 0; k < 512; k++) h_bank[f * 512 + k] = (float)(g[k] / tot);
    }
    std::vector<h8> h_ops(8 * 64);
    for (int jp = 0; jp < 8; jp++)
        for (int l = 0; l < 64; l++) {
            const int n = l & 15, kb = l >> 4, nn = n & 7, j = 2 * jp + (kb >> 1);
            for (int t = 0; t < 8; t++) {
                const float wv = h_bank[(((nn >> 1) * 16 + j) * 16 + (8 * (kb & 1) + t)) * 2 + (nn & 1)] * kTapScale;
                const _Float16 hi = (_Float16)wv;
                h_ops[jp * 64 + l][t] = n < 8 ? hi : (_Float16)(wv - (float)hi);
            }
        }
    uint16_t* d_img; h8* d_bank; uint32_t* d_cnt; float* d_resp;
    hipMalloc(&d_img, h_img.size() * 2); hipMalloc(&d_bank, h_ops.size() * sizeof(h8)); hipMalloc(&d_cnt, (size_t)R * 16); hipMalloc(&d_resp, kW * kH * 8 * 4);
    hipMemcpy(d_img, h_img.data(), h_img.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_bank, h_ops.data(), h_ops.size() * sizeof(h8), hipMemcpyHostToDevice);
    hipMemset(d_cnt, 0, (size_t)R * 16);
    const size_t lds = (size_t)2 * kPlane * 2;
    const float thr2 = 30.0f * 30.0f;
    hipLaunchKernelGGL(gabor_mfma_kernel, dim3(R), dim3(256), lds, 0, d_img, d_bank, thr2, d_cnt, d_resp);
    hipDeviceSynchronize();
    // ---- check against fp64 on the host: ROI 0, every pixel, the eight responses; and its four counts
    std::vector<float> h_resp(kW * kH * 8);
    std::vector<uint32_t> h_cnt(4);
    hipMemcpy(h_resp.data(), d_resp, h_resp.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h_cnt.data(), d_cnt, 16, hipMemcpyDeviceToHost);
    double max_err = 0, a_max = 4095.0;
    uint32_t ref_cnt[4] = {0, 0, 0, 0};
    for (int y = 0; y < kH; y++)
        for (int x = 0; x < kW; x++)
            for (int f = 0; f < 4; f++) {
                double re = 0, im = 0;
                for (int j = 0; j < 16; j++)
                    for (int i = 0; i < 16; i++) {
                        const int yy = y + j, xx = x + i;
                        const double a = (yy < kH && xx < kW) ? (double)h_img[yy * kW + xx] : 0.0;
                        re += a * (double)h_bank[f * 512 + (j * 16 + i) * 2];
                        im += a * (double)h_bank[f * 512 + (j * 16 + i) * 2 + 1];
                    }
                max_err = fmax(max_err, fabs(re - (double)h_resp[(y * kW + x) * 8 + 2 * f]));
                max_err = fmax(max_err, fabs(im - (double)h_resp[(y * kW + x) * 8 + 2 * f + 1]));
                if (re * re + im * im > (double)thr2) ref_cnt[f]++;
            }
    printf("check (ROI 0): max |response error| = %.3e = %.3e a_max  (packed-fp32 screening bound of the product: 1.54e-5 a_max per component)\n", max_err, max_err / a_max);
    printf("counts above threshold, MFMA vs fp64: %u/%u %u/%u %u/%u %u/%u\n", h_cnt[0], ref_cnt[0], h_cnt[1], ref_cnt[1], h_cnt[2], ref_cnt[2], h_cnt[3], ref_cnt[3]);
    {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gabor_mfma_kernel, 256, lds);
        printf("occupancy: %d workgroups per CU (LDS %zu B per workgroup)\n", nb, lds);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL(gabor_mfma_kernel, dim3(R), dim3(256), lds, 0, d_img, d_bank, thr2, d_cnt, (float*)nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms / iters, per196 = per * 196000.0 / R;
    const double flop = 2.0 * 64 * 64 * 256 * 16 * 2;                             // executed: 64 columns x 64 rows (4 row tiles) x 256 taps x 16 columns (8 x hi / lo) x 2 digits
    printf("%d ROIs: %.3f ms per launch -> %.2f ms per 196 000 ROIs (MFMA screening of 4 band-pass filters); %.1f TFLOP/s executed f16 MFMA\n", R, per, per196,
           flop * R / (per * 1e-3) / 1e12);
    printf("product today (profiles/r04z): roi_gabor_tiled_kernel 33.9 ms per 196 000 ROIs for the whole bank (box low-pass + 4 band-pass, decisions, exact redo)\n");
    return 0;
}
