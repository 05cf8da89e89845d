// texture_feats.h -- what the texture kernels share: column slots of the GLRLM / GLSZM blocks, the entropy term, and the GLRLM
// feature routine of one angle (one wave).  Used by roi_texture.hip (one workgroup per ROI) and roi_large_tex.hip (several
// workgroups per ROI).  Reference: /root/reference/src/nyx/features/glrlm.cpp:357-885.
#pragma once
#include <hip/hip_runtime.h>
#include "device_math.h"
#include "roi_kernel.h"

namespace nyxhip {

enum { R_SRE = 0, R_LRE, R_GLN, R_GLNN, R_RLN, R_RLNN, R_RP, R_GLV, R_RV, R_RE, R_LGLRE, R_HGLRE,
       R_SRLGLE, R_SRHGLE, R_LRLGLE, R_LRHGLE };
enum { Z_SAE = 0, Z_LAE, Z_GLN, Z_GLNN, Z_SZN, Z_SZNN, Z_ZP, Z_GLV, Z_ZV, Z_ZE, Z_LGLZE, Z_HGLZE,
       Z_SALGLE, Z_SAHGLE, Z_LALGLE, Z_LAHGLE };


// p * fast_log10(p + EPS) / LOG10_2 with EPS = 2.2e-16 (glrlm.h:169-171, glszm.h:138-140);
// (lg2 * c) / c == lg2 to 1 ulp, see device_math.h plogp.
__device__ __forceinline__ double plog_tex(double p)
{
    return p * (double)fast_log2f(p + 2.2e-16);
}

// ---- GLRLM features of one angle from its LDS matrix, by one wave -------------------------
// P[row * Nr + (len-1)], rows = level indices; lv[row] = level value (PixIntens).
// ri / rj: scratch for row / column sums.
// Always inlined: as a real call (shared by kernels built for different register budgets, reached with SGPRs spilled to VGPR
// lanes) this function produced wrong GLRLM rows / faults in the global-workspace launches and a wrong GLSZM_ZP in the
// 80-register build (tools/spill_probe.py, tests/test_parity_gpu.py::test_glrlm_alone_on_spilled_rois).

// lvf: per level row (i^2, 1 / i^2) as doubles (TexLayout::lvf), or null (many levels: formed on the spot)
template <bool GS>
__device__ __forceinline__ void glrlm_features_wave(const uint32_t* P, int Ng, int Nr, const uint32_t* lv, const double* lvf, uint32_t* ri, uint32_t* rj,
                                    uint32_t Np, double* f, double* park, int lane)
{
    auto sq_of = [=](int i, double& in2d, double& ri2) {
        if (lvf) { in2d = lvf[2 * i]; ri2 = lvf[2 * i + 1]; }
        else { in2d = (double)(uint32_t)(lv[i] * lv[i]); ri2 = frcp(in2d); }   // unsigned-int product as in the reference (levels can be intensities: 32-bit wrap)
    };
    // The kernel is bound by vector-instruction issue, so this routine is organised around the instruction count:
    //   * row sums by 16-lane groups (four levels at a time, a 4-step DPP sum each) instead of one lane walking a whole row;
    //   * the fifteen wave totals through two transposed reductions (8 + 7 values: ~35 exchanges each) instead of fifteen
    //     six-step butterflies, parked in the output slot `f` between the stages;
    //   * every quotient by a per-lane or per-wave reciprocal (two Newton steps: 1-2 ulp; all of GLRLM is tolerance-class)
    //     instead of an IEEE division per term, and 24-bit multiplies for the small integer products.
    const int slot = lane >> 4, l = lane & 15;
    uint32_t part = 0;
    for (int i0 = 0; i0 < Ng; i0 += 4) {
        const int i = i0 + slot;
        uint32_t sm = 0;
        if (i < Ng) {
            uint32_t idx = mul24((uint32_t)i, (uint32_t)Nr) + (uint32_t)l;
            for (int j = l; j < Nr; j += 16, idx += 16) sm += P[idx];
        }
        part += sm;
        sm = row16_sum(sm);
        if (l == 0 && i < Ng) ri[i] = sm;
    }
    for (int j = lane; j < Nr; j += 64) {
        uint32_t sm = 0, idx = (uint32_t)j;
        for (int i = 0; i < Ng; i++, idx += (uint32_t)Nr) sm += P[idx];
        rj[j] = sm;
    }
    const uint32_t tot = wave_sum_t<uint32_t>(part);       // number of runs: at most the pixel count
    wav_sync<GS>();
    if (tot == 0) { // sum_p == 0 -> every feature 0.0 (glrlm.cpp:364-367 etc.)
        if (lane < 16) f[lane] = 0.0;
        wav_sync<GS>();
        return;
    }
    const double sum_p = (double)tot, inv_p = frcp(sum_p);
    // level-only and length-only sums from the marginals
    double t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // gln, mu_g, lgl, hgl | sre, lre, rln, mu_r
    for (int i = lane; i < Ng; i += 64) {
        const double r = (double)ri[i];
        double in2d, ri2;
        sq_of(i, in2d, ri2);
        t8[0] += r * r;                                    // calc_GLN :431-461
        t8[1] += (r * inv_p) * (double)lv[i];              // calc_GLV mu :602-609
        t8[2] += r * ri2;                                  // calc_LGLRE :712-741
        t8[3] += r * in2d;                                 // calc_HGLRE :744-773
    }
    for (int j = lane; j < Nr; j += 64) {
        const double c = (double)rj[j], jd = (double)(j + 1), j2 = jd * jd;
        t8[4] += c * frcp(j2);                             // calc_SRE :378-385
        t8[5] += c * j2;                                   // calc_LRE :411-418 (integer-exact)
        t8[6] += c * c;                                    // calc_RLN :499-529
        t8[7] += (c * inv_p) * jd;                         // calc_RV mu :649-655
    }
    {
        const double tt = wave_transpose_sum8(t8, lane);   // lane 8 k holds total k
        if ((lane & 7) == 0) f[lane >> 3] = tt;
    }
    wav_sync<GS>();
    const double mu_g = f[1], mu_r = f[7];
    // stage 1 waits in the wave's eight slots of the block exchange area (`park`; nobody else touches it while the waves are in
    // here) instead of in sixteen registers of lane 0 that every lane carried through the cell loop below -- the 64-register
    // build spilled them to scratch, which the counters showed as gigabytes of HBM writes per launch
    if ((lane & 7) == 0) park[lane >> 3] = f[lane >> 3];
    wav_sync<GS>();
    double u8[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // glv, rv, re, srl, srh, lrl, lrh
    for (int i = lane; i < Ng; i += 64) {
        const double d = (double)lv[i] - mu_g;
        u8[0] += ((double)ri[i] * inv_p) * (d * d);        // calc_GLV :611-620
    }
    for (int j = lane; j < Nr; j += 64) {
        const double d = (double)(j + 1) - mu_r;
        u8[1] += ((double)rj[j] * inv_p) * (d * d);        // calc_RV :657-665
    }
    // cell-level sums: lane = run length (a per-lane 1 / j^2), rows in sequence (a per-wave 1 / i^2).  The reference multiplies the
    // two squares as 32-bit unsigned integers; while (largest level)^2 * Nr^2 stays below 2^32 that product is exact and the
    // quotients factor into the two reciprocals -- otherwise the wrapped product is formed and divided cell by cell.
    const uint32_t lv_max = lv[Ng - 1];                    // (levels are sorted)
    const bool exact = (unsigned long long)lv_max * lv_max * (unsigned long long)Nr * (unsigned long long)Nr < (1ull << 32);
    // The entropy term of a cell is a function of its count alone, and counts are small: lane k holds the term of count k, a cell
    // fetches it through ds_bpermute (all lanes take part: a zero cell reads lane 0's -0.0, which adds nothing, like the +-0 the
    // reference adds for it) -- 64 evaluations of the float log per wave instead of one per cell.
    const double ptab = plog_tex((double)lane * inv_p);
    // (two loops, one per arithmetic: with the choice inside one loop the compiler shuffled all eight accumulators through copies
    //  on every trip -- sixteen 64-bit moves per cell)
    auto entropy_of = [=](uint32_t c, double cnt) -> double {
        double e = __shfl(ptab, (int)(c & 63u), 64);
        if (__builtin_amdgcn_ballot_w64(c >= 64u))
            e = c >= 64u ? plog_tex(cnt * inv_p) : e;
        return e;
    };
    if (exact) {
        for (int j0 = 0; j0 < Nr; j0 += 64) {
            const int j = j0 + lane;
            const bool live = j < Nr;
            const double jd = (double)(j + 1), j2d = jd * jd, rj2 = frcp(j2d);
            uint32_t idx = (uint32_t)j;
            for (int i = 0; i < Ng; i++, idx += (uint32_t)Nr) {
                const uint32_t c = live ? P[idx] : 0u;
                const double cnt = (double)c;
                u8[2] += entropy_of(c, cnt);               // calc_RE :693-699
                double in2d, ri2;
                sq_of(i, in2d, ri2);
                const double a = cnt * rj2, b = cnt * j2d; // (a zero cell adds +0 to every sum)
                u8[3] = __builtin_fma(a, ri2, u8[3]);      // calc_SRLGLE :790-797
                u8[4] = __builtin_fma(a, in2d, u8[4]);     // calc_SRHGLE :822-829
                u8[5] = __builtin_fma(b, ri2, u8[5]);      // calc_LRLGLE :855-862
                u8[6] = __builtin_fma(b, in2d, u8[6]);     // calc_LRHGLE :887-894
            }
        }
    } else {
        // the reference's matrix is as wide as the angle's LONGEST RUN (glrlm.cpp:206-214), not as the box: only those columns exist
        uint32_t nr_true = 0;
        for (int j = lane; j < Nr; j += 64) nr_true = rj[j] != 0 ? (uint32_t)(j + 1) : nr_true;
        nr_true = wave_max_u32(nr_true);
        for (int j0 = 0; j0 < Nr; j0 += 64) {
            const int j = j0 + lane;
            const bool live = j < Nr;
            const double jd = (double)(j + 1), j2d = jd * jd;
            const uint32_t j2 = mul24((uint32_t)(j + 1), (uint32_t)(j + 1));
            uint32_t idx = (uint32_t)j;
            for (int i = 0; i < Ng; i++, idx += (uint32_t)Nr) {
                const uint32_t c = live ? P[idx] : 0u;
                const double cnt = (double)c;
                u8[2] += entropy_of(c, cnt);
                const uint32_t in2 = lv[i] * lv[i];
                if (c == 0) {
                    // (an empty cell adds 0 / (product) in the reference: 0 / 0 = NaN where the wrapped product is 0, e.g. level 256
                    //  and run length 256)
                    if ((uint32_t)j < nr_true && (uint32_t)(in2 * j2) == 0u) u8[3] += __builtin_nan("");
                } else {
                    u8[3] += cnt / (double)(uint32_t)(in2 * j2);
                    u8[4] += fdiv(cnt * (double)in2, j2d);
                    u8[5] += fdiv(cnt * j2d, (double)in2);
                    u8[6] += cnt * (double)(uint32_t)(in2 * j2);
                }
            }
        }
    }
    {
        const double tt = wave_transpose_sum8(u8, lane);
        if ((lane & 7) == 0) f[8 + (lane >> 3)] = tt;
    }
    wav_sync<GS>();
    if (lane == 0) {
        const double glv = f[8], rv = f[9], re = f[10], srl = f[11], srh = f[12], lrl = f[13], lrh = f[14];
        const double inv_p2 = inv_p * inv_p;
        double s8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) s8[k] = park[k];
        f[R_SRE] = s8[4] * inv_p;
        f[R_LRE] = s8[5] * inv_p;
        f[R_GLN] = s8[0] * inv_p;
        f[R_GLNN] = s8[0] * inv_p2;
        f[R_RLN] = s8[6] * inv_p;
        f[R_RLNN] = s8[6] * inv_p2;
        f[R_RP] = fdiv(sum_p, (double)(int)Np);             // calc_RP :569-585
        f[R_GLV] = glv;
        f[R_RV] = rv;
        f[R_RE] = -re;
        f[R_LGLRE] = s8[2] * inv_p;
        f[R_HGLRE] = s8[3] * inv_p;
        f[R_SRLGLE] = srl * inv_p;
        f[R_SRHGLE] = srh * inv_p;
        f[R_LRLGLE] = lrl * inv_p;
        f[R_LRHGLE] = lrh * inv_p;
    }
    wav_sync<GS>();
}

} // namespace nyxhip
