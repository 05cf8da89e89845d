// zernike_mfma_probe.hip -- round-5 review item 7: "Zernike's accumulation on v_mfma_f64_16x16x4_f64 ... measure against 4.2 ms and keep
// whichever wins".  The accumulation of roi_zernike_kernel is, per pixel, T[m][j] += (f / r) z^(m+1) (r^2)^j for ten complex powers
// (m = 0..9: twenty reals) and five radial weights (j = 0..4): an outer product with M = 20, N = 5, K = pixels.  Two register-resident
// forms of exactly that contraction, no memory traffic, every CU busy (1024 workgroups x 256 threads):
//   valu   lane = pixel: the complex power recurrence (10 complex multiplies) + the weights + 100 fp64 FMAs per pixel -- what the kernel
//          does today (the kernel adds the unit-disc test and the loads);
//   mfma   v_mfma_f64_16x16x4_f64: A = 16 of the twenty reals x 4 pixels, B = 4 pixels x (5 weights padded to 16 columns), two
//          M-tiles per four pixels.  Charged at its BEST: the operands are taken as they lie in registers -- the lane = pixel ->
//          (row, pixel) transposition of the twenty reals (a ds_bpermute each) and the powers themselves are NOT counted.
// Prints pixels per second per form and the ratio.  Build + run:  hipcc -O3 --offload-arch=gfx950 -o /tmp/zprobe tools/zernike_mfma_probe.hip && /tmp/zprobe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kPixPerLane = 512;                  // valu: pixels per lane; mfma: the wave covers the same 64 x 512 pixels, four per instruction pair

__global__ __launch_bounds__(256) void k_valu(double* out, double seed)
{
    double T[10][5][2];
#pragma unroll
    for (int m = 0; m < 10; m++)
#pragma unroll
        for (int j = 0; j < 5; j++) T[m][j][0] = T[m][j][1] = 0.0;
    double x = seed + 1e-3 * threadIdx.x, y = seed * 0.5 + 2e-3 * threadIdx.x;
    for (int p = 0; p < kPixPerLane; p++) {
        x = __builtin_fma(x, 0.999, 1e-4); y = __builtin_fma(y, 0.998, 2e-4);          // (a new pixel)
        const double r2 = x * x + y * y, f = 1.0 + x;                                // f / r folded into f here
        double w[5]; w[0] = f;
#pragma unroll
        for (int j = 1; j < 5; j++) w[j] = w[j - 1] * r2;
        double zr = x, zi = -y;                                                      // z = x - i y
        double pr = zr, pi = zi;
#pragma unroll
        for (int m = 0; m < 10; m++) {
#pragma unroll
            for (int j = 0; j < 5; j++) { T[m][j][0] = __builtin_fma(w[j], pr, T[m][j][0]); T[m][j][1] = __builtin_fma(w[j], pi, T[m][j][1]); }
            const double nr = pr * zr - pi * zi, ni = pr * zi + pi * zr;
            pr = nr; pi = ni;
        }
    }
    double s = 0;
#pragma unroll
    for (int m = 0; m < 10; m++)
#pragma unroll
        for (int j = 0; j < 5; j++) s += T[m][j][0] + T[m][j][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mfma(double* out, double seed)
{
    v4d c0 = v4d{0, 0, 0, 0}, c1 = v4d{0, 0, 0, 0};
    double a0 = seed + 1e-3 * threadIdx.x, a1 = seed * 0.5 + 2e-3 * threadIdx.x, b = 1.0 + 1e-6 * threadIdx.x;
    // the wave's 64 x kPixPerLane pixels, four per pair of instructions (rows 0..15 and 16..19 of the twenty reals)
    for (int q = 0; q < 64 * kPixPerLane / 4; q++) {
        a0 = __builtin_fma(a0, 0.999, 1e-4); a1 = __builtin_fma(a1, 0.998, 2e-4);      // (new operands: two instructions, as in k_valu)
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, c1, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
}

template <typename F>
static double time_ms(F&& launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0;
}

int main()
{
    const int grid = 1024;
    double* d;
    hipMalloc(&d, sizeof(double) * grid * 256);
    const double pixels = (double)grid * 256 * kPixPerLane;
    const double tv = time_ms([&] { hipLaunchKernelGGL(k_valu, dim3(grid), dim3(256), 0, 0, d, 0.25); });
    const double tm = time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, d, 0.25); });
    printf("valu  lane = pixel, 100 FMAs + recurrence          %8.3f ms  %7.2f G pixels/s  (196 k ROIs x 2821 px = 0.553 G pixels: %.2f ms)\n", tv, pixels / (tv * 1e-3) / 1e9, 0.553e9 / (pixels / (tv * 1e-3)) * 1e3);
    printf("mfma  2 x v_mfma_f64_16x16x4 per 4 pixels, operands free %8.3f ms  %7.2f G pixels/s  (%.2f ms)\n", tm, pixels / (tm * 1e-3) / 1e9, 0.553e9 / (pixels / (tm * 1e-3)) * 1e3);
    printf("ratio mfma / valu time: %.2f\n", tm / tv);
    hipFree(d);
    return 0;
}
