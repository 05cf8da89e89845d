// roi_dependence.hip -- GLDZM + GLDM + NGLDM for a batch of ROIs, one 256-thread workgroup
// per ROI, all state in LDS (or in the global workspace for ROIs beyond the LDS carve-out, GS = true).
//
//   GLDZM  /root/reference/src/nyx/features/gldzm.cpp:53-236 (matrix kit), :238-285 (dist2border), :300-420 (features)
//          zones = 4-connected sets of equal binned level (the reference floods E,S,W,N with a parent stack: the full
//          component); zone metric = min over members of the 1-based distance to the nearest zero / bbox margin along
//          the row and the column.  Here: distances in closed form for the margins + an outward probe for zeros,
//          components by union-find in LDS (atomicMin), zone metric by atomicMin at the root, matrix by atomics.
//   GLDM   features/gldm.cpp:16-255: per pixel with ORIGINAL intensity != 0, dependence = 1 + # of 8-neighbours that
//          are ROI pixels (original != 0) of the same binned level; P[level][dependence]; 14 features :300-560.
//   NGLDM  features/ngldm.cpp:40-225: levels = to_grayscale(v, 0, max, GREYDEPTH) on the CLOUD (mask = cloud
//          membership, zero intensities included); P[level][# matching neighbours]; 19 features :246-340.
//
// Binned plane as in roi_texture.hip (bin_intensities: matlab binning sends background to level 1).
// Matrix sums are formed lane-strided by one wave and combined in a fixed order (deterministic); the reference's
// serial sums differ from these at the 1e-15 level.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "device_math.h"
#include "roi_kernel.h"
#include "launch_util.h"
#include "../../include/nyxhip.h"

namespace nyxhip {

namespace {

constexpr int kBlk = 256;

__device__ __forceinline__ double plog_dep(double p)   // p * fast_log10(p + EPS) / LOG10_2, gldm.cpp:440-452 (see device_math.h plogp)
{
    return p * (double)fast_log2f(p + 2.2e-16);
}

// flags and level field of the auxiliary plane; P8 = both planes hold bytes (levels <= 63: 6 bits + 2 flags)
template <bool P8> struct AuxBits { static constexpr uint32_t kInCloud = 0x4000, kOrigNZ = 0x8000, kLvlMask = 0x0FFF; };
template <> struct AuxBits<true> { static constexpr uint32_t kInCloud = 0x40, kOrigNZ = 0x80, kLvlMask = 0x3F; };

} // namespace

// P8: 8-bit planes (grey depth <= 63, LDS launches): 7.4 KB less LDS for the benchmark ROI -- six workgroups per CU instead of five.
template <bool GS, bool P8 = false>
__global__ __launch_bounds__(kBlk) void roi_dependence_kernel(const DepArgs A)
{
    constexpr uint32_t kInCloud = AuxBits<P8>::kInCloud, kOrigNZ = AuxBits<P8>::kOrigNZ, kLvlMask = AuxBits<P8>::kLvlMask;
    using plane_t = typename std::conditional<P8, uint8_t, uint16_t>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char* const lds = GS ? A.sp.scratch + (size_t)blockIdx.x * A.sp.stride : lds_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    uint64_t roi;
    if (!roi_of_slot(A.sp, blockIdx.x, A.n_roi, roi))
        return;
    const int solo = (int)(roi & 3u);    // the wave that runs this ROI's single-wave stretches (spreads them over the four SIMDs)
    double* s_red = (double*)(lds + A.L.red);
    double* s_stat = (double*)(lds + A.L.stat);
    plane_t* s_dense = (plane_t*)(lds + A.L.dense);
    plane_t* s_aux = (plane_t*)(lds + A.L.aux);
    uint16_t* s_lvlmap = (uint16_t*)(lds + A.L.lvlmap);    // binned level -> row + 1   (GLDM, GLDZM)
    uint32_t* s_lv = (uint32_t*)(lds + A.L.lv);            // row -> level
    uint16_t* s_lvlmap2 = (uint16_t*)(lds + A.L.lvlmap2);  // NGLDM level -> row + 1
    uint32_t* s_lv2 = (uint32_t*)(lds + A.L.lv2);
    unsigned char* s_work = lds + A.L.work;

    const uint64_t off = A.px_offset[roi];
    const uint32_t n = (uint32_t)(A.px_offset[roi + 1] - off);
    const uint32_t w = A.bbox_w[roi], h = A.bbox_h[roi];
    const uint32_t area = w * h;
    const uint32_t vmin = A.min_inten[roi], vmax = A.max_inten[roi];
    double* const row_out = A.out + roi * A.ld;
    const bool do_dzm = (A.mask & NYXHIP_FAM_GLDZM) != 0, do_dm = (A.mask & NYXHIP_FAM_GLDM) != 0, do_ng = (A.mask & NYXHIP_FAM_NGLDM) != 0;
    const uint32_t side = w > h ? w : h;
    auto fill_all = [&](double v) {
        if (do_dzm) for (int c = tid; c < kGldzmCols; c += kBlk) row_out[A.col_gldzm + c] = v;
        if (do_dm) for (int c = tid; c < kGldmCols; c += kBlk) row_out[A.col_gldm + c] = v;
        if (do_ng) for (int c = tid; c < kNgldmCols; c += kBlk) row_out[A.col_ngldm + c] = v;
    };
    if (n == 0 || area > A.L.dense_cap || side > A.L.side_cap) {
        if (n != 0 && A.sp.defer_large)
            return;                                   // handled by the spill launch that follows
        if (tid == 0 && n != 0)
            atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        fill_all(__longlong_as_double(0x7ff8000000000000LL));
        return;
    }
    if (vmin == vmax) {                               // blank ROI: gldzm.cpp:199-221, gldm.cpp:20-38, ngldm.cpp:152-174
        fill_all(A.soft_nan);
        return;
    }
    const int greyInfo = A.ibsi ? 0 : A.grey_depth;
    const double mslope = greyInfo > 0 ? (double)greyInfo / ((double)vmax - 0.) : 0.0;
    const uint32_t Lcap = A.L.lvl_cap;

    // ---- phase 0: planes ------------------------------------------------------------------------------------
    {
        const uint32_t bg = greyInfo > 0 ? (P8 ? 0x01010101u : 0x00010001u) : 0u;   // matlab binning: background -> level 1 (texture_feature.h:150-154)
        uint32_t* d32 = (uint32_t*)s_dense;
        uint32_t* a32 = (uint32_t*)s_aux;
        for (uint32_t i = tid; i < (P8 ? (area + 3) / 4 : (area + 1) / 2); i += kBlk) { d32[i] = bg; a32[i] = 0; }
        for (uint32_t i = tid; i <= Lcap + 1; i += kBlk) { s_lvlmap[i] = 0; s_lvlmap2[i] = 0; }
    }
    blk_sync<GS>();
    uint32_t lvl_over = 0;
    for_each_cloud_pixel<kBlk>(A.inten + off, A.x + off, A.y + off, n, tid, [&](uint32_t, uint32_t v, uint32_t px, uint32_t py) {
        uint32_t lvl = greyInfo > 0 ? bin_matlab(v, mslope, greyInfo) : greyInfo < 0 ? bin_radiomix(v, vmin, vmax, -greyInfo) : v;
        // NGLDM level: to_grayscale (p, 0, max, GREYDEPTH, ibsi), ngldm.cpp:44-47 / :118-121 (helpers.h:337-345)
        uint32_t nl = A.ibsi ? v : to_grayscale(v, 0u, vmax, (uint32_t)A.grey_depth);
        if (lvl > Lcap) { lvl_over = 1; lvl = Lcap; }
        if (nl > Lcap || nl > kLvlMask) { if (do_ng) lvl_over = 1; nl = 0; }
        if (px < w && py < h) {
            s_dense[py * w + px] = (plane_t)lvl;
            s_aux[py * w + px] = (plane_t)(kInCloud | (v != 0 ? kOrigNZ : 0) | nl);
            s_lvlmap2[nl] = 1;
        }
    });
    lvl_over = wave_max_u32(lvl_over);
    if (lane == 0) s_red[wave * 8] = (double)lvl_over;
    blk_sync<GS>();
    {
        bool over = false;
        for (int wv = 0; wv < kBlk / 64; wv++) over |= s_red[wv * 8] != 0;
        if (over) {                                   // level beyond the resident capacity
            if (tid == 0) atomicCAS(A.status, 0, NYXHIP_ERR_UNSUPPORTED);
            fill_all(__longlong_as_double(0x7ff8000000000000LL));
            return;
        }
    }
    for (uint32_t p = tid; p < area; p += kBlk) {
        const uint32_t l = s_dense[p];
        if (l) s_lvlmap[l] = 1;
    }
    blk_sync<GS>();
    // sorted unique levels.  GLDM / GLDZM: non-zero levels of the whole plane, 1..max in IBSI mode (gldm.cpp:54-68,
    // gldzm.cpp:73-87); NGLDM: levels of the cloud, zero included (ngldm.cpp:40-53)
    if (tid == 0) {
        int k = 0;
        uint32_t mx = 0;
        for (uint32_t l = 1; l <= Lcap; l++)
            if (s_lvlmap[l]) {
                mx = l;
                if (greyInfo != 0) { s_lvlmap[l] = (uint16_t)(k + 1); s_lv[k] = l; }
                k++;
            }
        if (greyInfo == 0)
            for (uint32_t l = 1; l <= mx; l++) { s_lvlmap[l] = (uint16_t)l; s_lv[l - 1] = l; }
        s_stat[0] = (double)(greyInfo == 0 ? (int)mx : k);
    }
    if (tid == 64) {
        int k = 0;
        for (uint32_t l = 0; l <= Lcap && l <= kLvlMask; l++)
            if (s_lvlmap2[l]) { s_lvlmap2[l] = (uint16_t)(k + 1); s_lv2[k] = l; k++; }
        s_stat[1] = (double)k;
    }
    blk_sync<GS>();
    const int Ng = (int)s_stat[0];
    const int Ng2 = (int)s_stat[1];

    // ---- feature tails: one wave each.  Organised around the instruction count (the kernel is bound by vector-instruction issue):
    //   * a cell's quotients by level^2 / distance^2 through two reciprocals (1-2 ulp; tolerance-class sums) instead of up to
    //     eight Newton divisions, the division by the zone count as a multiplication by its reciprocal;
    //   * the wave totals through ONE transposed reduction (wave_transpose_sum16: ~60 exchanges for sixteen values) instead of
    //     one six-step butterfly per value, parked in s_red-like scratch of the wave's own (`scr`, 16 doubles);
    //   * row sums by 16-lane groups, cell indices without integer division.
    const bool par = !GS && A.L.par != 0;
    int Nd_dzm = 0, Nd_dm = 0, Nr_ng = 0;
    auto tail_totals16 = [&](double (&t)[16], double* scr) {   // every lane gets all sixteen totals back (through LDS)
        const double tt = wave_transpose_sum16(t, lane);       // lane 4 k holds total k
        if ((lane & 3) == 0) scr[lane >> 2] = tt;
        wav_sync<GS>();
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = scr[k];
        wav_sync<GS>();
    };
    auto dzm_tail = [&](const uint32_t* P, uint32_t ndmax, int Nd, double* o) {
        double* const scr = (double*)(lds + A.L.stat);         // 16 doubles (s_stat is dead by now)
        const int NN = Ng * Nd;
        uint32_t ns_i = 0;
        {
            RowCol rc((uint32_t)lane, 64u, (uint32_t)Nd, RowCol::small_t{});
            for (int e = lane; e < NN; e += 64, rc.advance()) ns_i += P[mad24(rc.row, ndmax, rc.col)];
        }
        const double Ns = (double)wave_sum_t<uint32_t>(ns_i), invNs = frcp(Ns);
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = 0.0;               // 0..6: SDLGLE SDHGLE LDLGLE LDHGLE GLM ZDM ZDE | 7..9: SDE LDE ZDNU | 10..12: LGLZE HGLZE GLNU
        {
            RowCol rc((uint32_t)lane, 64u, (uint32_t)Nd, RowCol::small_t{});
            for (int e = lane; e < NN; e += 64, rc.advance()) {
                const uint32_t pc = P[mad24(rc.row, ndmax, rc.col)];
                if (pc == 0) continue;
                const double p = (double)pc, g_ = (double)s_lv[rc.row], d_ = (double)(rc.col + 1);
                const double g2 = g_ * g_, d2 = d_ * d_, rg2 = frcp(g2), rd2 = frcp(d2);    // (levels and distances are >= 1)
                const double pg = p * rg2, pG = p * g2;
                t[0] = __builtin_fma(pg, rd2, t[0]);
                t[1] = __builtin_fma(pG, rd2, t[1]);
                t[2] = __builtin_fma(pg, d2, t[2]);
                t[3] = __builtin_fma(pG, d2, t[3]);
                t[4] = __builtin_fma(g_, p, t[4]);
                t[5] = __builtin_fma(d_, p, t[5]);
                const double pn = p * invNs;
                t[6] += pn * log2(pn + 2.2e-16);
            }
        }
        for (int d = lane; d < Nd; d += 64) {
            uint32_t m = 0, idx = (uint32_t)d;
            for (int g = 0; g < Ng; g++, idx += ndmax) m += P[idx];
            const double md = (double)m, dd = (double)(d + 1), d2 = dd * dd;
            t[7] += md * frcp(d2); t[8] += d2 * md; t[9] += md * md;
        }
        {
            const int slot = lane >> 4, l = lane & 15;
            for (int g0 = 0; g0 < Ng; g0 += 4) {
                const int g = g0 + slot;
                uint32_t x = 0;
                if (g < Ng) {
                    uint32_t idx = mul24((uint32_t)g, ndmax) + (uint32_t)l;
                    for (int d = l; d < Nd; d += 16, idx += 16) x += P[idx];
                }
                x = row16_sum(x);
                if (l == 0 && g < Ng) {
                    const double xd = (double)x, g_ = (double)s_lv[g], g2 = g_ * g_;
                    t[10] += xd * frcp(g2); t[11] += g2 * xd; t[12] += xd * xd;
                }
            }
        }
        tail_totals16(t, scr);
        const double GLM = t[4] * invNs, ZDM = t[5] * invNs;
        double v4[4] = {0, 0, 0, 0};
        {
            RowCol rc((uint32_t)lane, 64u, (uint32_t)Nd, RowCol::small_t{});
            for (int e = lane; e < NN; e += 64, rc.advance()) {
                const double p = (double)P[mad24(rc.row, ndmax, rc.col)] * invNs;
                double dif = (double)s_lv[rc.row] - GLM;
                v4[0] += dif * dif * p;
                dif = (double)(rc.col + 1) - ZDM;
                v4[1] += dif * dif * p;
            }
        }
        const double vt = wave_transpose_sum4(v4);             // lane 16 k holds total k
        if (lane == 0 || lane == 16) scr[lane >> 4] = vt;
        wav_sync<GS>();
        if (lane == 0) {
            const double glv = scr[0], zdv = scr[1];
            const double zdnu = t[9] * invNs, glnu = t[12] * invNs;
            o[0] = t[7] * invNs; o[1] = t[8] * invNs; o[2] = t[10] * invNs; o[3] = t[11] * invNs;
            o[4] = t[0] * invNs; o[5] = t[1] * invNs; o[6] = t[2] * invNs; o[7] = t[3] * invNs;
            o[8] = glnu; o[9] = glnu * invNs; o[10] = zdnu; o[11] = zdnu * invNs;
            o[12] = fdiv(Ns, (double)n);                       // ZP = Ns / roi_area :399
            o[13] = GLM; o[14] = glv; o[15] = ZDM; o[16] = zdv; o[17] = -t[6];
        }
    };
    auto dm_tail = [&](const uint32_t* P, int Nd, double* o) {
        double* const scr = s_red;                             // 32 doubles: [0..15] this tail, [16..31] the NGLDM tail
        uint32_t nz_i = 0;
        for (int e = lane; e < Ng * 9; e += 64) nz_i += P[e];
        const uint32_t nz = wave_sum_t<uint32_t>(nz_i);
        if (nz == 0) {
            if (lane < kGldmCols) o[lane] = A.soft_nan;        // :216-234
            return;
        }
        const double Nz = (double)nz, invNz = frcp(Nz);
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = 0.0;               // 0..8: SDE LDE mu_g mu_d DE SDLGLE SDHGLE LDLGLE LDHGLE | 9..11: GLN LGLE HGLE | 12: DN
        {
            RowCol rc((uint32_t)lane, 64u, 9u, RowCol::small_t{});
            for (int e = lane; e < Ng * 9; e += 64, rc.advance()) {
                const uint32_t ci = P[e];
                const int j = (int)rc.col + 1;
                if (ci == 0 || j > Nd) continue;
                const double c = (double)ci, inten = (double)s_lv[rc.row], jj = (double)j;
                const double i2 = inten * inten, j2 = jj * jj, ri2 = frcp(i2), rj2 = frcp(j2);   // (levels and dependence counts are >= 1)
                const double cn = c * invNz, cj = c * rj2, cJ = c * j2;
                t[0] += cj;
                t[1] += cJ;
                t[2] = __builtin_fma(cn, inten, t[2]);
                t[3] = __builtin_fma(cn, jj, t[3]);
                t[4] += plog_dep(cn);
                t[5] = __builtin_fma(cj, ri2, t[5]);
                t[6] = __builtin_fma(cj, i2, t[6]);
                t[7] = __builtin_fma(cJ, ri2, t[7]);
                t[8] = __builtin_fma(cJ, i2, t[8]);
            }
        }
        for (int i = lane; i < Ng; i += 64) {
            uint32_t si_i = 0;
            for (int j = 0; j < Nd; j++) si_i += P[i * 9 + j];
            const double si = (double)si_i, inten = (double)s_lv[i], i2 = inten * inten;
            t[9] += si * si; t[10] += si * frcp(i2); t[11] += si * i2;
        }
        if (lane < Nd) {
            uint32_t sj_i = 0;
            for (int i = 0; i < Ng; i++) sj_i += P[i * 9 + lane];
            const double sj = (double)sj_i;
            t[12] = sj * sj;
        }
        tail_totals16(t, scr);
        const double mu_g = t[2], mu_d = t[3];
        double v4[4] = {0, 0, 0, 0};
        {
            RowCol rc((uint32_t)lane, 64u, 9u, RowCol::small_t{});
            for (int e = lane; e < Ng * 9; e += 64, rc.advance()) {
                const uint32_t ci = P[e];
                const int j = (int)rc.col + 1;
                if (ci == 0 || j > Nd) continue;
                const double cn = (double)ci * invNz, dg = (double)s_lv[rc.row] - mu_g, dd = (double)j - mu_d;
                v4[0] += cn * (dg * dg);
                v4[1] += cn * (dd * dd);
            }
        }
        const double vt = wave_transpose_sum4(v4);
        if (lane == 0 || lane == 16) scr[lane >> 4] = vt;
        wav_sync<GS>();
        if (lane == 0) {
            const double glv = scr[0], dv = scr[1];
            o[0] = t[0] * invNz; o[1] = t[1] * invNz; o[2] = t[9] * invNz; o[3] = t[12] * invNz; o[4] = t[12] * invNz * invNz;
            o[5] = glv; o[6] = dv; o[7] = -t[4]; o[8] = t[10] * invNz; o[9] = t[11] * invNz;
            o[10] = t[5] * invNz; o[11] = t[6] * invNz; o[12] = t[7] * invNz; o[13] = t[8] * invNz;
        }
    };
    auto ng_tail = [&](const uint32_t* M, int Nr, double* o) {
        double* const scr = s_red + 16;
        const double Ns = (double)n, invNs = frcp(Ns);         // every cloud pixel is counted once
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = 0.0;               // 0..11: LDE HDE LGLCE HGLCE LDLGLE LDHGLE HDLGLE HDHGLE GLM DCM DCENT DCENE | 12, 13: sum Sg^2, sum Sr^2
        {
            RowCol rc((uint32_t)lane, 64u, 9u, RowCol::small_t{});
            for (int e = lane; e < Ng2 * 9; e += 64, rc.advance()) {
                const uint32_t si = M[e];
                const int j = (int)rc.col;
                if (si == 0 || j >= Nr) continue;
                const double sij = (double)si, gl = (double)s_lv2[rc.row], dc = (double)(j + 1), pij = sij * invNs;
                const double d2 = dc * dc, g2 = gl * gl, rd2 = frcp(d2), sd = sij * rd2, sD = sij * d2;
                t[0] += sd;
                t[1] += sD;
                if (gl != 0.0) {                               // (a zero level has no reciprocal: its terms are skipped, ngldm.cpp)
                    const double rg2 = frcp(g2);
                    t[2] = __builtin_fma(sij, rg2, t[2]);
                    t[4] = __builtin_fma(sd, rg2, t[4]);
                    t[6] = __builtin_fma(sD, rg2, t[6]);
                }
                t[3] = __builtin_fma(sij, g2, t[3]);
                t[5] = __builtin_fma(sd, g2, t[5]);
                t[7] = __builtin_fma(sD, g2, t[7]);
                t[8] = __builtin_fma(gl, pij, t[8]);
                t[9] = __builtin_fma(dc, pij, t[9]);
                t[10] -= pij * (log(pij) * 1.4426950408889634);   // log2 through the natural log, as the reference's log(p) / log(2)
                t[11] = __builtin_fma(pij, pij, t[11]);
            }
        }
        for (int i = lane; i < Ng2; i += 64) {
            uint32_t sg = 0;
            for (int j = 0; j < Nr; j++) sg += M[i * 9 + j];
            t[12] += (double)sg * (double)sg;
        }
        if (lane < Nr) {
            uint32_t sr = 0;
            for (int i = 0; i < Ng2; i++) sr += M[i * 9 + lane];
            t[13] = (double)sr * (double)sr;
        }
        tail_totals16(t, scr);
        const double GLM = t[8], DCM = t[9];
        double v4[4] = {0, 0, 0, 0};
        {
            RowCol rc((uint32_t)lane, 64u, 9u, RowCol::small_t{});
            for (int e = lane; e < Ng2 * 9; e += 64, rc.advance()) {
                const uint32_t si = M[e];
                const int j = (int)rc.col;
                if (si == 0 || j >= Nr) continue;
                const double gl = (double)s_lv2[rc.row], dc = (double)(j + 1), pij = (double)si * invNs;
                v4[0] += (gl - GLM) * (gl - GLM) * pij;
                v4[1] += (dc - DCM) * (dc - DCM) * pij;
            }
        }
        const double vt = wave_transpose_sum4(v4);
        if (lane == 0 || lane == 16) scr[lane >> 4] = vt;
        wav_sync<GS>();
        if (lane == 0) {
            const double glv = scr[0], dcv = scr[1];
            o[0] = t[0] * invNs; o[1] = t[1] * invNs; o[2] = t[2] * invNs; o[3] = t[3] * invNs; o[4] = t[4] * invNs; o[5] = t[5] * invNs;
            o[6] = t[6] * invNs; o[7] = t[7] * invNs; o[8] = t[12] * invNs; o[9] = t[12] * invNs * invNs; o[10] = t[13] * invNs; o[11] = t[13] * invNs * invNs;
            o[12] = 1.0;                                       // DCP :339
            o[13] = GLM; o[14] = glv; o[15] = DCM; o[16] = dcv; o[17] = t[10]; o[18] = t[11];
        }
    };

    // =====================================================================================================
    // GLDZM
    // =====================================================================================================
    if (do_dzm) {
        double* o = row_out + A.col_gldzm;
        uint32_t* s_label = (uint32_t*)s_work;                       // [area]   union-find parents, then the zone metric at the roots
        uint32_t* s_P = s_label + A.L.dense_cap;                     // [Ng * ndmax]
        const uint32_t ndmax = ((w < h ? w : h) + 1) / 2 + 1;
        constexpr uint32_t kNone = 0xFFFFFFFFu, kRootTag = 0xC0000000u;
        // distance of a pixel = min over the four axis directions of (steps to the nearest zero or to the bbox margin) + 1
        // (dist2border :238-285); a pixel on the margin gets 1.  Closed form for the margins, outward probe for zeros
        // (only the IBSI plane holds zeros: matlab binning has none, radiomics binning is refused by the host).
        auto dist_of = [=](uint32_t p, uint32_t y, uint32_t x) {
            uint32_t best = x + 1;
            best = w - x < best ? w - x : best;
            best = y + 1 < best ? y + 1 : best;
            best = h - y < best ? h - y : best;
            if (greyInfo == 0)
                for (uint32_t k = 1; k + 1 < best; k++)
                    if (s_dense[p - k] == 0 || s_dense[p + k] == 0 || s_dense[p - k * w] == 0 || s_dense[p + k * w] == 0) { best = k + 1; break; }
            return best;
        };
        for (uint32_t i = tid; i < (uint32_t)Ng * ndmax; i += kBlk)
            s_P[i] = 0;
        // 4-connected components of equal level (zeros are not zone material in IBSI mode, gldzm.cpp:103-106).
        // (1) horizontal runs: one wave per row, 64 pixels per step; a ballot of the run-start flags gives every lane the
        //     start of its run (highest start bit at or below the lane), so a run is born with one parent: its first pixel.
        for (uint32_t y = wave; y < h; y += kBlk / 64) {
            uint32_t carry = 0;                                      // start (x) of the run that crosses into this chunk
            for (uint32_t x0 = 0; x0 < w; x0 += 64) {
                const uint32_t x = x0 + lane;
                const bool in = x < w;
                const uint32_t p = y * w + (in ? x : 0);
                const uint32_t v = in ? (uint32_t)s_dense[p] : 0u;
                const bool starts = in && (x == 0 || s_dense[p - 1] != v);
                const unsigned long long m = __ballot(starts);
                const unsigned long long below = m & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
                const uint32_t start = below ? x0 + 63u - (uint32_t)__clzll((long long)below) : carry;
                if (in) s_label[p] = (greyInfo > 0 || v != 0) ? y * w + start : kNone;
                carry = m ? x0 + 63u - (uint32_t)__clzll((long long)m) : carry;
            }
        }
        blk_sync<GS>();
        // (2) vertical adjacencies: union-find in place (larger root under the smaller, atomicMin).  A pixel merges with its
        //     north neighbour unless the pair to its west already joined the same two runs.
        auto find = [=](uint32_t v) { for (uint32_t l = s_label[v]; l != v; l = s_label[v]) v = l; return v; };
        auto unite = [=](uint32_t a, uint32_t b) {
            for (;;) {
                a = find(a); b = find(b);
                if (a == b) return;
                if (a < b) { const uint32_t t = a; a = b; b = t; }      // a > b: hang a under b
                const uint32_t old = atomicMin(&s_label[a], b);
                if (old == a) return;                                   // a was still a root: merged
                a = old;                                                // somebody re-parented a first: retry from there
            }
        };
        if (w <= 64) {
            // boxes up to a wave wide: lane = column, a block of rows per wave, the row above in a register and the west pair
            // through DPP lane shifts -- one plane read per pixel instead of a label read and up to four plane reads
            // (a pixel has a label unless it is an IBSI zero: s_label[p] != kNone <=> greyInfo > 0 || v != 0)
            const uint32_t rows_pw = (h + kBlk / 64 - 1) / (kBlk / 64);
            const uint32_t r_begin = (uint32_t)wave * rows_pw > 1u ? (uint32_t)wave * rows_pw : 1u;
            const uint32_t r_end = ((uint32_t)wave + 1u) * rows_pw < h ? ((uint32_t)wave + 1u) * rows_pw : h;
            const bool in_col = (uint32_t)lane < w;
            uint32_t up = (in_col && r_begin < r_end) ? (uint32_t)s_dense[(r_begin - 1) * w + (uint32_t)lane] : 0xFFFFFFFFu;
            for (uint32_t y = r_begin; y < r_end; y++) {
                const uint32_t p = y * w + (uint32_t)lane;
                const uint32_t v = in_col ? (uint32_t)s_dense[p] : 0xFFFFFFFEu;
                const uint32_t vw = lane_minus1(v, 0xFFFFFFFDu), upw = lane_minus1(up, 0xFFFFFFFCu);
                if (in_col && (greyInfo > 0 || v != 0) && up == v && !(vw == v && upw == v))
                    unite(p, p - w);
                up = v;
            }
        } else {
        RowCol rc_v(w + (uint32_t)tid, kBlk, w);
        for (uint32_t p = w + tid; p < area; p += kBlk, rc_v.advance()) {
            if (s_label[p] == kNone) continue;
            const uint32_t v = s_dense[p];
            if (s_dense[p - w] != v) continue;
            const uint32_t x = rc_v.col;
            if (x > 0 && s_dense[p - 1] == v && s_dense[p - w - 1] == v) continue;
            unite(p, p - w);
        }
        }
        blk_sync<GS>();
        for (uint32_t p = tid; p < area; p += kBlk)
            if (s_label[p] != kNone) {
                const uint32_t r = find(p);
                if (r != p) s_label[p] = r;                             // writes only shorten chains: concurrent finds stay valid
            }
        blk_sync<GS>();
        // (3) zone metric = min distance over the members, kept in the root's own slot as kRootTag | (0xFFFF - d) under
        //     atomicMax (a member reads its root from its own slot, which nobody else writes)
        RowCol rc_d((uint32_t)tid, kBlk, w);
        for (uint32_t p = tid; p < area; p += kBlk, rc_d.advance()) {
            const uint32_t l = s_label[p];
            if (l == kNone) continue;
            const uint32_t d = dist_of(p, rc_d.row, rc_d.col);
            atomicMax(&s_label[l >= kRootTag ? p : l], kRootTag | (0xFFFFu - d));
        }
        blk_sync<GS>();
        uint32_t nd_loc = 0;
        for (uint32_t p = tid; p < area; p += kBlk) {
            const uint32_t l = s_label[p];
            if (l != kNone && l >= kRootTag) {
                const uint32_t d = 0xFFFFu - (l & 0xFFFFu);
                const uint32_t rowi = (uint32_t)s_lvlmap[s_dense[p]] - 1;
                atomicAdd(&s_P[rowi * ndmax + (d - 1)], 1u);
                nd_loc = d > nd_loc ? d : nd_loc;
            }
        }
        nd_loc = wave_max_u32(nd_loc);
        if (lane == 0) s_red[wave * 8] = (double)nd_loc;
        blk_sync<GS>();
        int Nd = 0;
        for (int wv = 0; wv < kBlk / 64; wv++) Nd = (int)s_red[wv * 8] > Nd ? (int)s_red[wv * 8] : Nd;
        blk_sync<GS>();
        // features (calc_features :323-420): dzm_tail, on the solo wave now or next to the other tails at the end
        Nd_dzm = Nd;
        if (!par) {
            if (wave == solo) dzm_tail(s_P, ndmax, Nd, o);
            blk_sync<GS>();
        }
    }

    // =====================================================================================================
    // GLDM
    // =====================================================================================================
    if (do_dm) {
        double* o = row_out + A.col_gldm;
        uint32_t* s_P = (uint32_t*)(s_work + (par ? A.L.off_pdm : 0u));   // [Ng][9]
        for (int i = tid; i < Ng * 9; i += kBlk) s_P[i] = 0;
        blk_sync<GS>();
        uint32_t nd_loc = 0;
        if (w <= 64) {
            // one lane per column, a block of rows per wave: the rows above / below travel in registers, horizontal neighbours
            // come through DPP lane shifts.  code = level + 1 for a pixel that takes part, 0 otherwise (outside the box, or
            // ORIGINAL intensity 0): a neighbour depends on the centre iff the codes are equal.
            const int rows_per_wave = ((int)h + kBlk / 64 - 1) / (kBlk / 64);
            const int r_begin = wave * rows_per_wave;
            const int r_end = (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h;
            const bool in_col = (uint32_t)lane < w;
            auto code_row = [=](int r) -> uint32_t {
                if (!(in_col && r >= 0 && r < (int)h)) return 0u;
                const uint32_t q = (uint32_t)r * w + (uint32_t)lane;
                return (s_aux[q] & kOrigNZ) ? (uint32_t)s_dense[q] + 1u : 0u;
            };
            uint32_t prv = code_row(r_begin - 1), cur = code_row(r_begin);
            for (int row = r_begin; row < r_end; row++) {
                const uint32_t nxt = code_row(row + 1);
                const uint32_t nd = 1u + (uint32_t)(lane_minus1(prv, 0u) == cur) + (uint32_t)(prv == cur) + (uint32_t)(lane_plus1(prv, 0u) == cur) +
                                    (uint32_t)(lane_minus1(cur, 0u) == cur) + (uint32_t)(lane_plus1(cur, 0u) == cur) +
                                    (uint32_t)(lane_minus1(nxt, 0u) == cur) + (uint32_t)(nxt == cur) + (uint32_t)(lane_plus1(nxt, 0u) == cur);
                if (cur != 0) {
                    const uint32_t pi = cur - 1u;
                    const uint32_t rowi = greyInfo == 0 ? pi - 1 : (uint32_t)s_lvlmap[pi] - 1;
                    atomicAdd(&s_P[rowi * 9 + (nd - 1)], 1u);
                    nd_loc = nd > nd_loc ? nd : nd_loc;
                }
                prv = cur; cur = nxt;
            }
        } else {
        RowCol rc_dm((uint32_t)tid, kBlk, w);
        for (uint32_t p = tid; p < area; p += kBlk, rc_dm.advance()) {
            if (!(s_aux[p] & kOrigNZ)) continue;                     // skip by ORIGINAL intensity, gldm.cpp:83-84
            const uint32_t pi = s_dense[p];
            const int y = (int)rc_dm.row, x = (int)rc_dm.col;
            uint32_t nd = 1;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int dy = k < 3 ? -1 : (k == 3 || k == 7) ? 0 : 1;
                const int dx = (k == 0 || k == 6 || k == 7) ? -1 : (k == 1 || k == 5) ? 0 : 1;
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || xx < 0 || yy >= (int)h || xx >= (int)w) continue;
                const uint32_t q = (uint32_t)yy * w + (uint32_t)xx;
                nd += ((s_aux[q] & kOrigNZ) && s_dense[q] == pi) ? 1u : 0u;
            }
            const uint32_t rowi = greyInfo == 0 ? pi - 1 : (uint32_t)s_lvlmap[pi] - 1;
            atomicAdd(&s_P[rowi * 9 + (nd - 1)], 1u);
            nd_loc = nd > nd_loc ? nd : nd_loc;
        }
        }
        nd_loc = wave_max_u32(nd_loc);
        if (lane == 0) s_red[wave * 8] = (double)nd_loc;
        blk_sync<GS>();
        int Nd = 0;
        for (int wv = 0; wv < kBlk / 64; wv++) Nd = (int)s_red[wv * 8] > Nd ? (int)s_red[wv * 8] : Nd;
        if (greyInfo == 0) Nd = 9;                                   // gldm.cpp:173,208-209
        blk_sync<GS>();
        Nd_dm = Nd;
        if (!par) {
            if (wave == solo) dm_tail(s_P, Nd, o);
            blk_sync<GS>();
        }
    }

    // =====================================================================================================
    // NGLDM
    // =====================================================================================================
    if (do_ng) {
        double* o = row_out + A.col_ngldm;
        uint32_t* s_M = (uint32_t*)(s_work + (par ? A.L.off_m : 0u));    // [Ng2][9]
        for (int i = tid; i < Ng2 * 9; i += kBlk) s_M[i] = 0;
        blk_sync<GS>();
        uint32_t dep_loc = 0;
        if (w <= 64) {                                               // same row-stepped stencil; code = cloud level + 1
            const int rows_per_wave = ((int)h + kBlk / 64 - 1) / (kBlk / 64);
            const int r_begin = wave * rows_per_wave;
            const int r_end = (r_begin + rows_per_wave) < (int)h ? (r_begin + rows_per_wave) : (int)h;
            const bool in_col = (uint32_t)lane < w;
            auto code_row = [=](int r) -> uint32_t {
                if (!(in_col && r >= 0 && r < (int)h)) return 0u;
                const uint32_t ap = s_aux[(uint32_t)r * w + (uint32_t)lane];
                return (ap & kInCloud) ? (ap & kLvlMask) + 1u : 0u;
            };
            uint32_t prv = code_row(r_begin - 1), cur = code_row(r_begin);
            for (int row = r_begin; row < r_end; row++) {
                const uint32_t nxt = code_row(row + 1);
                const uint32_t nm = (uint32_t)(lane_minus1(prv, 0u) == cur) + (uint32_t)(prv == cur) + (uint32_t)(lane_plus1(prv, 0u) == cur) +
                                    (uint32_t)(lane_minus1(cur, 0u) == cur) + (uint32_t)(lane_plus1(cur, 0u) == cur) +
                                    (uint32_t)(lane_minus1(nxt, 0u) == cur) + (uint32_t)(nxt == cur) + (uint32_t)(lane_plus1(nxt, 0u) == cur);
                if (cur != 0) {
                    atomicAdd(&s_M[((uint32_t)s_lvlmap2[cur - 1u] - 1) * 9 + nm], 1u);
                    dep_loc = nm > dep_loc ? nm : dep_loc;
                }
                prv = cur; cur = nxt;
            }
        } else {
        RowCol rc_ng((uint32_t)tid, kBlk, w);
        for (uint32_t p = tid; p < area; p += kBlk, rc_ng.advance()) {
            const uint32_t ap = s_aux[p];
            if (!(ap & kInCloud)) continue;
            const uint32_t c = ap & kLvlMask;
            const int y = (int)rc_ng.row, x = (int)rc_ng.col;
            uint32_t nm = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int dy = k < 3 ? -1 : (k == 3 || k == 7) ? 0 : 1;
                const int dx = (k == 0 || k == 6 || k == 7) ? -1 : (k == 1 || k == 5) ? 0 : 1;
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || xx < 0 || yy >= (int)h || xx >= (int)w) continue;
                const uint32_t aq = s_aux[(uint32_t)yy * w + (uint32_t)xx];
                nm += ((aq & kInCloud) && (aq & kLvlMask) == c) ? 1u : 0u;
            }
            atomicAdd(&s_M[((uint32_t)s_lvlmap2[c] - 1) * 9 + nm], 1u);
            dep_loc = nm > dep_loc ? nm : dep_loc;
        }
        }
        dep_loc = wave_max_u32(dep_loc);
        if (lane == 0) s_red[wave * 8] = (double)dep_loc;
        blk_sync<GS>();
        int Nr = 0;
        for (int wv = 0; wv < kBlk / 64; wv++) Nr = (int)s_red[wv * 8] > Nr ? (int)s_red[wv * 8] : Nr;
        Nr += 1;                                                     // ngldm.cpp:142
        blk_sync<GS>();
        Nr_ng = Nr;
        if (!par) {
            if (wave == solo) ng_tail(s_M, Nr, o);
        }
    }
    // ---- the three feature tails side by side: one wave each (they are short serial chains of one wave; run one after the other
    //      they left three quarters of the workgroup waiting three times)
    if (par) {
        blk_sync<GS>();
        if (do_dzm && wave == solo)
            dzm_tail((const uint32_t*)s_work + A.L.dense_cap, ((w < h ? w : h) + 1) / 2 + 1, Nd_dzm, row_out + A.col_gldzm);
        else if (do_dm && wave == ((solo + 1) & 3))
            dm_tail((const uint32_t*)(s_work + A.L.off_pdm), Nd_dm, row_out + A.col_gldm);
        else if (do_ng && wave == ((solo + 2) & 3))
            ng_tail((const uint32_t*)(s_work + A.L.off_m), Nr_ng, row_out + A.col_ngldm);
    }
}

int launch_roi_dependence(const DepArgs& a, void* stream, uint32_t grid)
{
    static DeviceOnce optin;
    if (int orc = optin.run([]() -> int {
        hipError_t e = hipFuncSetAttribute((const void*)roi_dependence_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)roi_features_max_lds());
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)roi_dependence_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)roi_features_max_lds());
        return (int)e;
    }))
        return orc;
    if (grid == 0)
        return 0;
    if (getenv("NYXHIP_DEBUG")) fprintf(stderr, "[nyxhip] dependence launch: mask %u planes8 %u total %u work %u par %u\n", a.mask, a.L.planes8, a.L.total, a.L.work_bytes, a.L.par);
    if (a.sp.scratch)
        hipLaunchKernelGGL(roi_dependence_kernel<true>, dim3(grid), dim3(kBlk), 0, (hipStream_t)stream, a);
    else if (a.L.planes8)
        hipLaunchKernelGGL((roi_dependence_kernel<false, true>), dim3(grid), dim3(kBlk), a.L.total, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(roi_dependence_kernel<false>, dim3(grid), dim3(kBlk), a.L.total, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

} // namespace nyxhip
