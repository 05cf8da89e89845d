// pipe_probe.hip -- what the idle pipes of gfx950 could give the Gabor filter bank (DESIGN 4.3 / round-3 review item 4).
//
// The tiled Gabor kernel is a stream of fp64 FMAs on the vector ALU (98.6 % of its issue slots).  Three alternatives for the
// same contraction -- out[pixel][filter] += image[pixel + tap] * bank[tap][filter] -- are timed here in isolation, each as a
// register-resident loop with no memory traffic, every CU busy (1024 workgroups x 256 threads, four waves per SIMD):
//   f64     v_fma_f64                                  (what the kernel does today)
//   pk32    v_pk_fma_f32: two fp32 FMAs per lane        (a screening pass in fp32 with a rigorous bound + exact redo)
//   mfma    v_mfma_f64_16x16x4_f64                      (im2col tile on the matrix pipe)
//   both    f64 FMAs and f64 MFMAs interleaved in every wave (do the two pipes overlap?)
// Prints TFLOP/s per variant.  Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/pipe_probe tools/pipe_probe.hip && /tmp/pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kIters = 4096;

__global__ __launch_bounds__(256) void k_f64(double* out, double w)
{
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = (double)(threadIdx.x + k);
    double x = (double)threadIdx.x * 1e-9;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) a[k] = __builtin_fma(x, w, a[k]);       // 16 independent fp64 FMAs
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk32(float* out, float w)
{
    v2f a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = v2f{(float)(threadIdx.x + k), (float)k};
    const v2f x = v2f{(float)threadIdx.x * 1e-6f, (float)threadIdx.x * 2e-6f}, ww = v2f{w, -w};
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) a[k] = __builtin_elementwise_fma(x, ww, a[k]);   // 16 independent packed fp32 FMAs (v_pk_fma_f32)
    }
    v2f s = v2f{0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

__global__ __launch_bounds__(256) void k_mfma(double* out, double w)
{
    v4d c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = v4d{0, 0, 0, 0};
    const double a = (double)threadIdx.x * 1e-9, b = w;
    for (int it = 0; it < kIters / 4; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);   // 4 independent tiles: 2048 flop each
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) s += c[k].x + c[k].y + c[k].z + c[k].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// per iteration: 4 MFMAs (4 x 2048 flop per wave) and NV x 16 vector FMAs (NV x 16 x 128 flop per wave)
template <int NV>
__global__ __launch_bounds__(256) void k_both(double* out, double w)
{
    v4d c[4];
    double v[16];
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = v4d{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = (double)k;
    const double a = (double)threadIdx.x * 1e-9, b = w;
    for (int it = 0; it < kIters / 4; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV * 4; q++) v[(k * 4 + q) & 15] = __builtin_fma(a, w, v[(k * 4 + q) & 15]);
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) s += c[k].x + c[k].y + c[k].z + c[k].w;
#pragma unroll
    for (int k = 0; k < 16; k++) s += v[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F&& launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main()
{
    const int grid = 4096;
    double* d; float* f;
    hipMalloc(&d, sizeof(double) * grid * 256); hipMalloc(&f, sizeof(float) * grid * 256);
    const double waves = (double)grid * 4;
    double t;
    t = time_ms([&] { hipLaunchKernelGGL(k_f64, dim3(grid), dim3(256), 0, 0, d, 1.0000001); });
    printf("f64   v_fma_f64            %8.3f ms  %7.1f TFLOP/s\n", t, waves * kIters * 16 * 128.0 / (t * 1e-3) / 1e12);
    t = time_ms([&] { hipLaunchKernelGGL(k_pk32, dim3(grid), dim3(256), 0, 0, f, 1.0000001f); });
    printf("pk32  v_pk_fma_f32         %8.3f ms  %7.1f TFLOP/s (fp32)\n", t, waves * kIters * 16 * 256.0 / (t * 1e-3) / 1e12);
    t = time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, d, 1.0000001); });
    printf("mfma  v_mfma_f64_16x16x4   %8.3f ms  %7.1f TFLOP/s\n", t, waves * kIters * 2048.0 / (t * 1e-3) / 1e12);
    t = time_ms([&] { hipLaunchKernelGGL(k_both<1>, dim3(grid), dim3(256), 0, 0, d, 1.0000001); });
    printf("both  4 mfma + 16 fma/iter %8.3f ms  %7.1f TFLOP/s (mfma %.1f + fma %.1f)\n", t, waves * (kIters / 4) * (4 * 2048.0 + 16 * 128.0) / (t * 1e-3) / 1e12,
           waves * (kIters / 4) * 4 * 2048.0 / (t * 1e-3) / 1e12, waves * (kIters / 4) * 16 * 128.0 / (t * 1e-3) / 1e12);
    t = time_ms([&] { hipLaunchKernelGGL(k_both<4>, dim3(grid), dim3(256), 0, 0, d, 1.0000001); });
    printf("both  4 mfma + 64 fma/iter %8.3f ms  %7.1f TFLOP/s (mfma %.1f + fma %.1f)\n", t, waves * (kIters / 4) * (4 * 2048.0 + 64 * 128.0) / (t * 1e-3) / 1e12,
           waves * (kIters / 4) * 4 * 2048.0 / (t * 1e-3) / 1e12, waves * (kIters / 4) * 64 * 128.0 / (t * 1e-3) / 1e12);
    hipFree(d); hipFree(f);
    return 0;
}
