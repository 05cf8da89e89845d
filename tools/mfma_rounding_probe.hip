// mfma_rounding_probe.hip -- how v_mfma_f32_16x16x32_f16 adds up, measured: the error model of the Gabor screening stage
// (roi_shape.hip, run_bands_mfma) rests on these answers.  One instruction computes D = C + sum_{k<32} a_k b_k (every row of A is
// the vector a, every column of B the vector b, so all 256 outputs are that one number); the cases below pick a, b, C so that
// different internal designs give different D:
//   1. absorption: C = 2^24 and 32 products of 1 -- a chain of fp32 additions rounds every one of them away (D = 2^24), an adder
//      that sums the products exactly before it rounds gives 2^24 + 32;
//   2. final rounding: C = 2^24 plus products that sum to 1, 3, 5, 7 -- round-to-nearest-even, truncation and round-up differ;
//   3. alignment width: +2^12, -2^12 and thirty products of 2^-s: the small products survive the cancellation only if the adder
//      keeps at least 12 + s bits below the largest term; the largest s with an exact answer is the width;
//   4. random vectors at the screening stage's magnitudes against an exact (integer) evaluation: the worst error in units of the
//      result's ulp.
// Build + run on the GPU box:  hipcc -O2 --offload-arch=gfx950 -o /tmp/mfma_rounding_probe tools/mfma_rounding_probe.hip && /tmp/mfma_rounding_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// cases: [n][32] a, [n][32] b, [n] c  ->  [n] d
__global__ void one_mfma(const _Float16* a, const _Float16* b, const float* c, float* d, int n)
{
    const int lane = threadIdx.x & 63, kb = lane >> 4;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        h8 A, B;
        for (int t = 0; t < 8; t++) { A[t] = a[i * 32 + 8 * kb + t]; B[t] = b[i * 32 + 8 * kb + t]; }
        f4 C = f4{c[i], c[i], c[i], c[i]};
        C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0);
        if (lane == 0) d[i] = C[0];
    }
}

struct Case { _Float16 a[32], b[32]; float c; };

static std::vector<float> run(const std::vector<Case>& cs)
{
    const int n = (int)cs.size();
    std::vector<_Float16> a((size_t)n * 32), b((size_t)n * 32);
    std::vector<float> c(n), d(n);
    for (int i = 0; i < n; i++) { memcpy(&a[(size_t)i * 32], cs[i].a, 64); memcpy(&b[(size_t)i * 32], cs[i].b, 64); c[i] = cs[i].c; }
    _Float16 *da, *db; float *dc, *dd;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dc, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_mfma, dim3(256), dim3(64), 0, 0, da, db, dc, dd, n);
    hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
    return d;
}

int main()
{
    // 1. absorption
    {
        Case k{}; for (int i = 0; i < 32; i++) { k.a[i] = (_Float16)1.0f; k.b[i] = (_Float16)1.0f; } k.c = 16777216.0f;
        const float d = run({k})[0];
        printf("1. C = 2^24, 32 products of 1: D - 2^24 = %.0f  (32: products summed before the rounding; 0: a chain of fp32 additions)\n", (double)d - 16777216.0);
    }
    // 2. final rounding
    {
        std::vector<Case> cs;
        for (int s : {1, 3, 5, 7, -1, -3}) { Case k{}; k.a[0] = (_Float16)(float)s; k.b[0] = (_Float16)1.0f; k.c = 16777216.0f; cs.push_back(k); }
        const std::vector<float> d = run(cs);
        printf("2. C = 2^24 + {1, 3, 5, 7, -1, -3}: D - 2^24 = %.0f %.0f %.0f %.0f %.0f %.0f  (nearest-even: 0 4 4 8 -1 -3; truncation: 0 2 4 6 -1 -3)\n",
               (double)d[0] - 16777216.0, (double)d[1] - 16777216.0, (double)d[2] - 16777216.0, (double)d[3] - 16777216.0, (double)d[4] - 16777216.0, (double)d[5] - 16777216.0);
    }
    // 3. alignment width under cancellation: 2^12 - 2^12 + 30 * 2^-s (as products 2^-(s/2) * 2^-(s - s/2), both f16-representable)
    {
        std::vector<Case> cs;
        for (int s = 0; s <= 24; s++) {
            Case k{}; k.a[0] = (_Float16)4096.0f; k.b[0] = (_Float16)1.0f; k.a[1] = (_Float16)-4096.0f; k.b[1] = (_Float16)1.0f;
            for (int i = 2; i < 32; i++) { k.a[i] = (_Float16)ldexpf(1.0f, -(s / 2)); k.b[i] = (_Float16)ldexpf(1.0f, -(s - s / 2)); }
            k.c = 0.0f; cs.push_back(k);
        }
        const std::vector<float> d = run(cs);
        int width = -1;
        for (int s = 0; s <= 24; s++) if ((double)d[s] == 30.0 * ldexp(1.0, -s)) width = s; else break;
        printf("3. 2^12 - 2^12 + 30 x 2^-s exact up to s = %d: the adder keeps >= %d bits below its largest term", width, 12 + width);
        if (width < 24) printf("  (s = %d gave %.10g, exact %.10g)", width + 1, (double)d[width + 1], 30.0 * ldexp(1.0, -(width + 1)));
        printf("\n");
        // the same with the large terms in the ADDEND: C = 2^24, product -2^12 * 2^12, and the small products
        std::vector<Case> c2;
        for (int s = 0; s <= 24; s++) {
            Case k{}; k.a[0] = (_Float16)-4096.0f; k.b[0] = (_Float16)4096.0f;
            for (int i = 2; i < 32; i++) { k.a[i] = (_Float16)ldexpf(1.0f, -(s / 2)); k.b[i] = (_Float16)ldexpf(1.0f, -(s - s / 2)); }
            k.c = 16777216.0f; c2.push_back(k);
        }
        const std::vector<float> d2 = run(c2);
        int w2 = -1;
        for (int s = 0; s <= 24; s++) if ((double)d2[s] == 30.0 * ldexp(1.0, -s)) w2 = s; else break;
        printf("   C = 2^24, product -2^24, + 30 x 2^-s exact up to s = %d: >= %d bits below the addend\n", w2, 24 + w2);
    }
    // 4. random vectors at the stage's magnitudes: digits 0..2047 (even, the main plane of 12-bit pixels), taps hi parts (11-bit
    //    significands, scaled by 2^14), an addend of the size the chain carries; exact value by integers (every product is an integer
    //    times 2^-e with e <= 24: sums in long double are exact here)
    {
        srand(3);
        std::vector<Case> cs(20000);
        std::vector<long double> exact(cs.size());
        for (size_t i = 0; i < cs.size(); i++) {
            Case& k = cs[i];
            long double ex = 0;
            for (int t = 0; t < 32; t++) {
                const float dig = (float)(2 * (rand() % 2048));
                const float tap = ldexpf((float)(rand() % 2048 - 1024), -(rand() % 12));       // up to 1024, down to 2^-11 steps
                k.a[t] = (_Float16)dig; k.b[t] = (_Float16)tap;
                ex += (long double)(float)k.a[t] * (long double)(float)k.b[t];
            }
            k.c = (float)((rand() % 2000001 - 1000000) * 64.0);
            exact[i] = ex + (long double)k.c;
        }
        const std::vector<float> d = run(cs);
        double worst_ulp = 0, worst_rel = 0;
        for (size_t i = 0; i < cs.size(); i++) {
            const double err = fabs((double)((long double)d[i] - exact[i]));
            int e; frexp((double)fabsl(exact[i]), &e);
            const double ulp = ldexp(1.0, e - 24);
            if (fabsl(exact[i]) > 0) { worst_ulp = fmax(worst_ulp, err / ulp); }
            long double mag = fabsl((long double)cs[i].c);
            for (int t = 0; t < 32; t++) mag += fabsl((long double)(float)cs[i].a[t] * (long double)(float)cs[i].b[t]);
            worst_rel = fmax(worst_rel, err / (double)mag);
        }
        printf("4. 20000 random instructions: worst |D - exact| = %.3f ulp of the result = %.3g of (|C| + sum |a b|)  (a single rounding to nearest: 0.5 ulp)\n", worst_ulp, worst_rel);
    }
    return 0;
}
