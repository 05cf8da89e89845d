// roi_kernel.h -- host/device shared declarations of the fused per-ROI kernel.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/nyxhip.h"

namespace nyxhip {

constexpr int kBlock = 256;          // 4 wave64 per workgroup, one workgroup per ROI
constexpr int kWaves = kBlock / 64;
constexpr int kIntensityCols = 36;   // Feature2D COV..UNIFORMITY_PIU (featureset.h:12-46)
constexpr int kGlcmAngled = 30;      // Feature2D GLCM_ASM..GLCM_VARIANCE (featureset.h:174-203)
constexpr int kGlcmAve = 29;         // Feature2D GLCM_ASM_AVE..GLCM_SUMVARIANCE_AVE (:205-233)
constexpr int kMaxAngles = 4;

// ---- size classes (launch_device_all of nyxhip_api.hip) -------------------------------------------------------------------------
// Class of an ROI = 2 * size class + (1: its pixel count or intensity range rules the 16-bit tables out).  Size class k of
// 0 .. 3: n_px <= kClassPx[k] and both box sides <= kClassSide[k]; 4: everything larger.  A function of the ROI alone.
// Size classes 0 .. 2 run from LDS (their carve-outs at the CLASS BOUNDS fit a CU for every setting the LDS kernels serve -- checked
// per call on the host, run_class), classes 3 and 4 take the large-ROI path (roi_large.hip: several workgroups per ROI).  So that
// "LDS or large" is a function of the ROI and the settings alone -- never of the ROIs that share its call -- an ROI of size
// class 2 whose intensity range needs 32-bit sort keys (range >= 65536: two key buffers of 4 B per pixel do not fit next to a
// 128 x 128 plane) counts as size class 3, and so does any ROI whose IBSI level count (lvl = its largest intensity under IBSI, 0
// otherwise) exceeds kLdsLevels: the co-occurrence matrix of such an ROI is sized by its own intensities.
constexpr int kSizeClasses = 5, kClasses = 2 * kSizeClasses;
constexpr uint32_t kClassPx[kSizeClasses - 1] = {256, 4096, 16384, 32768};
constexpr uint32_t kClassSide[kSizeClasses - 1] = {32, 64, 128, 256};
constexpr uint32_t kLdsLevels = 128;          // largest IBSI matrix order the LDS launches of size classes 0 .. 2 are sized for
constexpr int kFirstLargeSizeClass = 3;
__host__ __device__ inline int roi_class(uint32_t n, uint32_t w, uint32_t h, uint32_t range, uint32_t lvl = 0)
{
    const uint32_t side = w > h ? w : h;
    int sc = (n <= 256u && side <= 32u) ? 0 : (n <= 4096u && side <= 64u) ? 1 : (n <= 16384u && side <= 128u) ? 2 : (n <= 32768u && side <= 256u) ? 3 : 4;
    if (sc == 2 && range >= 65536u) sc = 3;
    if (sc < 3 && lvl > kLdsLevels) sc = 3;
    const bool c16 = n < 65536u && range < 16384u;       // the 16-bit counting tables of roi_features.hip (LdsLayout::cnt16) can serve this ROI
    return 2 * sc + (c16 ? 0 : 1);
}
static_assert(kClassPx[0] == 256 && kClassPx[1] == 4096 && kClassPx[2] == 16384 && kClassPx[3] == 32768 && kClassSide[0] == 32 &&
              kClassSide[1] == 64 && kClassSide[2] == 128 && kClassSide[3] == 256, "roi_class spells the bounds out");

// Large-ROI ("spill") launches: the same kernels instantiated with their per-workgroup scratch in a global
// workspace instead of LDS, run over an index list of the ROIs that do not fit the LDS carve-out.
struct SpillArgs {
    const uint32_t* roi_index;   // NULL: workgroup b handles ROI b; else ROI roi_index[b]
    unsigned char* scratch;      // global scratch, `stride` bytes per workgroup (spill launches only)
    uint64_t stride;
    int32_t defer_large;         // LDS launch: silently skip ROIs beyond the caps, a workspace launch over them follows.  Set by the
                                 // contour launch of launch_moments only: the other kernels are launched per size class with
                                 // carve-outs of the class's own extrema and never meet an ROI beyond them
    uint32_t n_slots;            // entries of roi_index this launch may use (list launches; kernels that pack several ROIs per workgroup round their grids up)
    // launches over the whole batch (slot = ROI) that serve some classes only: bit c set = ROIs of class c are this launch's,
    // everybody else returns at once (the kernel derives the class from what it loads anyway: no list, no dependent load in
    // front of the ROI's own data).  0 = no filter.
    uint32_t class_mask;
    uint32_t min_range;          // launches of the feature kernel: serve only ROIs whose intensity range reaches this (0: all) -- the rest of
                                 // the class went through the histogram path of roi_large.hip / the 16-bit path of roi_wide.hip
    uint32_t max_range;          // ... and only ROIs whose range stays within this (0: no bound): the GLCM-only launch beside roi_wide.hip
    uint32_t skip_ltex;          // workspace launches of the texture kernel: 1 = leave out the ROIs the several-workgroups-per-ROI texture
                                 // path serves (ltex_eligible, roi_large_tex.hip)
};
__device__ __forceinline__ bool roi_in_launch(const SpillArgs& sp, uint32_t n, uint32_t w, uint32_t h, uint32_t range)
{
    return sp.class_mask == 0 || ((sp.class_mask >> roi_class(n, w, h, range)) & 1u) != 0;
}

// slot of a launch (workgroup, or wave of a wave-per-ROI launch) -> ROI; false: nothing to do for this slot
__device__ __forceinline__ bool roi_of_slot(const SpillArgs& sp, uint64_t slot, uint64_t n_roi, uint64_t& roi)
{
    if (sp.roi_index) {
        if (slot >= sp.n_slots) return false;
        roi = sp.roi_index[slot];
    } else
        roi = slot;
    return roi < n_roi;
}

// Byte offsets of the regions carved out of the workgroup's dynamic LDS; computed
// on the host per launch (size classes differ per batch).
struct LdsLayout {
    uint32_t out;      // double[n_cols]       staged output row
    uint32_t red;      // double[kWaves*8]     cross-wave reduction scratch
    uint32_t stat;     // double[16]           block-wide scalars
    uint32_t lb100;    // uint32[104]          lower bounds of the 100 percentile bins
    uint32_t lbc;      // uint32[n_hist+8]     lower bounds of the n-bin histogram
    uint32_t val;      // uint32[sort_cap]     intensities (sorted in place by the sort engine)
    uint32_t cnt;      // uint32[count_cap]    counting table over [min, max] -> prefix sums
    uint32_t dense;    // uint16[dense_cap]    binned bounding-box plane (0 = skip)
    uint32_t lvlmap;   // uint16[lvl_cap+8]    radiomics level -> compact index
    uint32_t P;        // uint32[app*ng_cap^2] co-occurrence counts
    uint32_t gscr;     // double[kMaxAngles*(6*ng_cap+40)] per-angle marginals + features
    uint32_t total;
    uint32_t sort_cap;   // >= max_px (a power of two when the sort engine may run)
    uint32_t count_cap;  // intensity ranges below this use the counting engine (0 = never)
    uint32_t cnt16;      // 1: 16-bit counting table (every ROI of the launch has < 65536 pixels)
    uint32_t dense8;     // 1: 8-bit binned plane (grey depth <= 254; only together with cnt16 and the split GLCM launch)
    uint32_t dense_cap;  // >= max bbox area
    uint32_t ng_cap;     // max GLCM matrix order held in LDS
    uint32_t lvl_cap;    // number of radiomics bins
    uint32_t app;        // angles per co-occurrence pass (4, 2 or 1)
    uint32_t g16;        // 1: matlab binning with 17..64 levels, 16-bit co-occurrence cells, marginal-based features (roi_features_kernel_g16)
    // workspace launches (state in global memory): regions whose offset lies below this bound still live in the workgroup's LDS
    // -- the small, atomics-heavy ones: reduction scratch, histogram bounds, co-occurrence matrices, the counting table when it
    // fits -- and only the ROI-sized buffers (values, binned plane) in the workspace
    uint32_t gs_lds_bytes;
    uint32_t radix;      // != 0: every ROI of the launch sorts (its size class is the wide-range one): offset of the radix sort's second key
                         // buffer [sort_cap] and digit counts [4 * 256 + 4]; no counting table, no power-of-two padding of the values
    uint32_t radix_k16;  // 1: every range of the launch fits 16 bits -- the keys are 16-bit offsets from the ROI minimum, BOTH key buffers
                         // live in the value region ([2][sort_cap rounded up to 8] u16) and `radix` holds the digit counts only
};

// Window source of the fused tile path: when `inten` is set the feature kernel reads an ROI's pixels from its bounding-box
// window of the tile stack (label match) instead of from x / y / inten clouds.
struct WindowSrc {
    const void* inten;           // tile stack [n_tiles][H][W], element size dt_inten
    const void* lab;             // label stack, element size dt_label
    int32_t dt_inten, dt_label;
    uint32_t W, H;
    const uint32_t* tile;        // [n_roi] tile of the ROI
    const uint32_t* label;       // [n_roi] its label value
    const uint32_t* x0;          // [n_roi] bounding-box origin inside the tile
    const uint32_t* y0;
    uint32_t xcd_swz;            // 1: workgroup b serves ROI xcd_slot(b): the eight XCDs (workgroups go to them round-robin) each take a
                                 // contiguous eighth of the rows, so the ROIs whose windows share cache lines of a tile meet in ONE L2
};
// Workgroup -> slot of a window-mode launch (grid rounded up to a multiple of 8): XCD b & 7 walks slots [x per, (x + 1) per).
__host__ __device__ inline uint64_t xcd_slot(uint32_t b, uint32_t grid) { const uint32_t per = (grid + 7u) >> 3; return (uint64_t)(b & 7u) * per + (b >> 3); }

struct RoiArgs {
    uint64_t n_roi;
    const uint64_t* px_offset;
    const uint16_t* x;
    const uint16_t* y;
    const uint32_t* inten;
    const uint32_t* bbox_w;
    const uint32_t* bbox_h;
    const uint32_t* min_inten;
    const uint32_t* max_inten;
    const double* slide_min;
    const double* slide_max;
    double* out;
    uint64_t ld;
    int* status;         // device word: first error code raised by any workgroup
    unsigned long long* stamps; // diagnostic build only (-DNYX_STAMP): per-phase cycle sums; NULL otherwise
    uint32_t mask;
    int32_t n_cols;
    int32_t col_intensity;   // first column of each family's block (-1 = absent)
    int32_t col_glcm;
    // settings (nyxhip_settings)
    double soft_nan;
    int32_t grey_depth, ibsi, glcm_grey_depth, glcm_offset, glcm_na, glcm_symmetric;
    int32_t glcm_angles[kMaxAngles];
    int32_t n_hist;          // |grey_depth| = intensity histogram bins
    // split GLCM (small matrices): this kernel exports the co-occurrence counts, glcm_features_kernel derives the features
    uint32_t* glcm_ws;       // [n_roi][glcm_ws_stride] counts, angle-major; NULL = features inside this kernel
    uint32_t* glcm_ng;       // [n_roi] matrix order of the ROI, 0 = nothing to derive (degenerate / skipped ROI)
    uint32_t glcm_ws_stride; // words per ROI = n_angles * ng_cap^2
    uint32_t* census;        // roi_small_kernel, scanning form: += ROIs of <= 256 pixels met (nullptr: not counted); read with the status flag
    uint32_t small_class;    // launch_roi_features: 1 = this launch serves class 0 (roi_class) through a list or a class filter, 2 = a whole-batch
                             // launch whose stated extrema promise class 0 only -> the wave-per-ROI kernel of roi_small.hip; 0 = roi_features_kernel
    uint32_t glcm_feats;     // split GLCM launches: 0 = glcm_features_kernel follows this launch over the same slots; 1 = not after this launch (the next
                             // launch group of the call derives this group's ROIs as well); 2 = it follows over ALL slots, class filter off
    SpillArgs sp;
    WindowSrc win;
    LdsLayout L;
};

// ---- second kernel: GLRLM + GLSZM + NGTDM (roi_texture.hip) -------------------------------
constexpr int kGlrlmCols = 16 * 4 + 16;   // Feature2D GLRLM_SRE..GLRLM_LRHGLE x4, then 16 _AVE (featureset.h:236-268)
constexpr int kGlszmCols = 16;            // Feature2D GLSZM_SAE..GLSZM_LAHGLE (featureset.h:291-306)
constexpr int kNgtdmCols = 5;             // Feature2D NGTDM_COARSENESS..NGTDM_STRENGTH (featureset.h:346-350)

constexpr int kRlmLdsCols = 16;           // GLRLM run lengths counted in LDS by the global-workspace launches (TexLayout::gs_rlm)
struct TexLayout {
    uint32_t out;       // double[n_cols]
    uint32_t red;       // double[kWaves*8]
    uint32_t stat;      // double[16]
    uint32_t dense;     // uint16[dense_cap]   binned bounding-box plane (background = level 1 under matlab binning)
    uint32_t lvlmap;    // uint16[lvl_cap+2]   level -> row index + 1
    uint32_t lv;        // uint32[ng_cap+2]    row index -> level value
    uint32_t lvf;       // double[2][ng_cap+2] row index -> (level^2, 1 / level^2) for the feature passes, 0 = none (more than 256 levels)
    uint32_t work;      // per-family scratch, families run one after the other
    uint32_t total;
    uint32_t dense_cap, side_cap, lvl_cap, ng_cap, hash_cap, work_bytes, szm_ok;
    uint32_t szm_c16;   // GLSZM zone-size table holds 16-bit entries (dense_cap < 65536, LDS launches): two per word
    uint32_t szm_small, szm_smalltab;   // GLSZM: zones of size <= szm_small (0 or 32) are counted in a direct [level][size] table at this offset of `work`
    uint32_t szm_count, szm_hkey, szm_label;   // offsets inside `work`: sizes, hash keys (values follow), owner labels (only when side_cap > 256)
    uint32_t dense8;    // 1: the binned plane holds 8-bit levels (LDS launches with a grey depth <= 254)
    uint32_t ngt_rep, ngt_stride;   // ngt_rep (a power of two) replicas of the NGTDM accumulators, ngt_stride bytes apart, at ngt_own or at the start of `work`
    uint32_t ngt_p;     // offset inside `work` of the NGTDM feature pass's double arrays
    // global-workspace launches: what stays in LDS all the same (small, atomics-heavy), offsets into the kernel's dynamic LDS
    uint32_t gs_lds_bytes;
    uint32_t gs_ngt, gs_ngt_ok;   // NGTDM accumulators (ngt_rep replicas, ngt_stride apart)
    uint32_t gs_rlm, gs_rlm_ok;   // GLRLM: columns 1 .. kRlmLdsCols of the four matrices, [4][ng_cap][kRlmLdsCols] u32
    uint32_t ngt_own;   // NGTDM accumulators (u64 S[ng_cap+2], u32 N[ng_cap+2]) outside `work`, 0 = none: lets the NGTDM stencil run on the
                        // three waves that would otherwise wait for the one-wave GLSZM row sweep
};

// Slots of the GLSZM (level, size) hash for a bounding box of `area` pixels at `ng` level rows: the distinct pairs number at most
// sqrt(2 ng area) + ng (the sizes of one level sum to <= area), kept at a load <= 2/3.  A function of the ROI and the settings
// alone: the kernel sizes the table of every ROI with it (the launch's carve-out, TexLayout::hash_cap, is the same function of
// the launch's largest box), so the slot order -- and with it the order of the floating-point sums over the cells -- does not
// depend on which other ROIs share the launch.  Integer arithmetic only: host and device agree on every value.
__host__ __device__ inline uint32_t szm_hash_cap(uint32_t ng, uint32_t area)
{
    const unsigned long long t = 2ull * ng * area;
    unsigned long long r = (unsigned long long)__builtin_sqrt((double)t);
    while (r * r > t) r--;
    while ((r + 1) * (r + 1) <= t) r++;                          // r = floor(sqrt(t)) whatever the rounding of the estimate
    const unsigned long long bound = r + 1 + ng;
    const uint32_t distinct = (uint32_t)(bound < area ? bound : area) + 1;
    uint32_t want = distinct + distinct / 2 + 8, p = 1;
    while (p < want) p <<= 1;
    return p;
}

struct TexArgs {
    uint64_t n_roi;
    const uint64_t* px_offset;
    const uint16_t* x;
    const uint16_t* y;
    const uint32_t* inten;
    const uint32_t* bbox_w;
    const uint32_t* bbox_h;
    const uint32_t* min_inten;
    const uint32_t* max_inten;
    double* out;
    uint64_t ld;
    int* status;
    uint32_t mask;        // subset of GLRLM | GLSZM | NGTDM
    int32_t n_cols;       // columns this kernel writes
    int32_t col0;         // first of them inside the output row
    int32_t gap_after_glrlm;  // columns of other kernels' families that sit between this kernel's blocks in Feature2D
    int32_t gap_after_glszm;  //   order: GLDZM after GLRLM; GLDM + NGLDM after GLSZM
    double soft_nan;
    int32_t grey_depth, ibsi;
    SpillArgs sp;
    TexLayout L;
};

// ---- GLDZM + GLDM + NGLDM (roi_dependence.hip) ---------------------------------------------
constexpr int kGldzmCols = 18;            // Feature2D GLDZM_SDE..GLDZM_ZDE (featureset.h:271-288)
constexpr int kGldmCols = 14;             // Feature2D GLDM_SDE..GLDM_LDHGLE (featureset.h:309-322)
constexpr int kNgldmCols = 19;            // Feature2D NGLDM_LDE..NGLDM_DCENE (featureset.h:325-343)

struct DepLayout {
    uint32_t red;       // double[kWaves*8]
    uint32_t stat;      // double[16]
    uint32_t dense;     // uint16[dense_cap]   binned plane (background = level 1 under matlab binning)
    uint32_t aux;       // uint16[dense_cap]   cloud membership | original != 0 | NGLDM level
    uint32_t lvlmap;    // uint16[lvl_cap+2]   binned level -> row + 1
    uint32_t lv;        // uint32[ng_cap+2]
    uint32_t lvlmap2;   // uint16[lvl_cap+2]   NGLDM level -> row + 1
    uint32_t lv2;       // uint32[ng_cap+2]
    uint32_t work;      // per-family scratch, families run one after the other
    uint32_t total;
    uint32_t dense_cap, side_cap, lvl_cap, ng_cap, nd_cap, work_bytes;
    uint32_t par;       // 1: the GLDM / NGLDM matrices have places of their own inside `work` (off_pdm, off_m), so the three feature
                        // tails run side by side on three waves at the end of the kernel instead of one after the other on one wave
    uint32_t off_pdm, off_m;
    uint32_t planes8;   // 1: both planes hold bytes (grey depth <= 63, LDS launches): roi_dependence_kernel<false, true>
};

struct DepArgs {
    uint64_t n_roi;
    const uint64_t* px_offset;
    const uint16_t* x;
    const uint16_t* y;
    const uint32_t* inten;
    const uint32_t* bbox_w;
    const uint32_t* bbox_h;
    const uint32_t* min_inten;
    const uint32_t* max_inten;
    double* out;
    uint64_t ld;
    int* status;
    uint32_t mask;        // subset of GLDZM | GLDM | NGLDM
    int32_t col_gldzm, col_gldm, col_ngldm;   // first column of each block inside the output row
    double soft_nan;
    int32_t grey_depth, ibsi;
    SpillArgs sp;
    DepLayout L;
};

// ---- contour + 2-D geometric moments (roi_moments.hip) -----------------------------------------
constexpr int kMomCols = 90;              // per family: RM 13, CM 16, NRM 16, NCM 7, HU 7, WRM 10, WCM 7, WNCM 7, WHU 7
constexpr int kMomStepTab = 2048;         // hill-descent step table (window width -> step): upper bound of MomArgs::step_cap
constexpr int kMomContourLds = 2048;      // upper bound of MomArgs::k_cap, the contour points the moments kernel keeps in LDS (longer contours are read from HBM)
constexpr int kMomPxLds = 3072;           // upper bound of MomArgs::px_cap, the ROI pixels (x | y << 16, intensity) the moments kernel keeps in LDS for its six sweeps

constexpr int kContourWaves = 4;     // ROIs (waves) per workgroup of roi_contour_kernel

struct MomArgs {
    uint32_t grid_rois;       // contour launch: ROI slots of the launch (kContourWaves per workgroup)
    uint64_t n_roi;
    const uint64_t* px_offset;
    const uint16_t* x;
    const uint16_t* y;
    const uint32_t* inten;
    const uint32_t* bbox_w;
    const uint32_t* bbox_h;
    double* out;
    uint64_t ld;
    int* status;
    uint32_t mask;            // subset of SMOMS | IMOMS
    int32_t col_smoms, col_imoms;
    uint32_t* ws_contour;     // [total pixels]  merged multicontour of every ROI at its CSR offset: x | y << 16, padded coordinates
    uint32_t* n_contour;      // [n_roi]         its length
    double* ws_L;             // [total pixels]  log(distance to contour + eps) per pixel (the contour kernel's walk stack before that)
    const double* log_tab;    // [log_tab_n]     log(sqrt(d) + 0.001) for the integer squared distances d < log_tab_n (context-owned, built once
    uint32_t log_tab_n;       //                 on the device by the same expression: bit-identical to evaluating it per pixel)
    uint32_t plane_cap;       // bytes of the padded flag plane one ROI may use
    uint32_t px_cap, k_cap, step_cap;   // moments kernel: pixels / contour points / step-table entries its dynamic LDS holds
    SpillArgs sp;
};

// ---- third kernel pair: Gabor + Zernike (roi_shape.hip) -------------------------------------
constexpr double kGaborTapScale = 16384.0;   // taps of the MFMA screening stage: f16 parts of tap x 2^14 (|tap| <= 1: the bank is L1-normalised)
constexpr int kZernikeCols = 30;          // ZernikeFeature::NUM_FEATURE_VALS (zernike.h:30)

struct ShapeLayout {
    uint32_t red;       // double[kWaves*8]
    uint32_t plane;     // double[area_cap]   original intensities (0 = background)
    uint32_t energy;    // double[area_cap]   low-pass response magnitudes
    uint32_t bank;      // double[(F+1)*n*n*2]
    uint32_t total;
    uint32_t area_cap;
    uint32_t zern_px_cap; // Zernike: pixels of a cloud staged in LDS (larger clouds are re-read from HBM)
    uint32_t side_cap;  // tiled layout only
    uint32_t tiled;     // 1: `plane` is the zero-padded u32 plane of roi_gabor_tiled_kernel (16 x 16 kernels, LDS launches);
                        //    no energy plane and no bank copy
    uint32_t redo;      // tiled layout: [1 + kGaborRedoCap] words -- the pixels whose decision the fused taps cannot make
};

struct ShapeArgs {
    uint64_t n_roi;
    const uint64_t* px_offset;
    const uint16_t* x;
    const uint16_t* y;
    const uint32_t* inten;
    const uint32_t* bbox_w;
    const uint32_t* bbox_h;
    const uint32_t* min_inten;
    const uint32_t* max_inten;
    double* out;
    uint64_t ld;
    int* status;
    uint32_t mask;            // subset of GABOR | ZERNIKE
    int32_t col_gabor, col_zernike;
    double soft_nan;
    const double* gabor_bank; // device: (F+1) filters (low-pass first), n*n complex taps each
    const float* gabor_bank32; // the same taps rounded to fp32 (the screening pass of roi_gabor_tiled_kernel, MODE 3)
    const void* gabor_bank16; // 16 x 16 banks: [ceil(F / 4)][8 tap-row pairs][64 lanes] x 8 f16 -- the band-pass filters as B operands of the MFMA
                              // screening stage (roi_gabor_tiled_kernel MODE 4; built by ensure_gabor_bank), or null
    int32_t gabor_nf, gabor_n;
    // 16 x 16 banks: per filter, bit j = every real part of tap row j is +-0, bit 16 + j = every imaginary part is.  Such a row adds
    // +-0 to the running sums, which leaves them as they are (they start at +0 and no sum of the scan is -0): the tiled kernel
    // skips that half of the row's arithmetic -- the reference's default bank has f0 = 0 in its first filter, i.e. sin(0) = 0
    // in every imaginary part
    uint32_t gabor_zero_rows[NYXHIP_MAX_GABOR_FILTERS + 1];
    // bit f: every tap of filter f is (c, +-0) with one power of two c -- a box filter, which the reference's default bank has
    // as its first filter (f0 = 0: infinite envelope, cos 0 = 1, normalised by 256).  Its response is exact integer arithmetic
    // (roi_gabor_tiled_kernel).
    uint32_t gabor_box_mask;
    double gabor_thr;
    int32_t small_rois;       // batch extrema say every ROI is small: one wave per ROI instead of four
    // 16 x 16 banks whose low-pass filter factors as tap(j, i) = C_j * B_i, C complex, B real >= 0 (the reference's always does: it is
    // built at theta = pi / 2, where the wave runs along the tap rows and the Gaussian envelope splits; ensure_gabor_bank checks the
    // taps it built, residual <= 1e-12 sum |tap|): fp32 factors for the separable screening pass of roi_gabor_tiled_kernel.
    uint32_t gabor_lp_sep;
    float gabor_lp_B[16];     // B_i
    float gabor_lp_C[44];     // (re, im) of C_j at [2 (j + 3)], three zero pairs at both ends
    int32_t dbg_phase;        // diagnostic builds of roi_shape.hip (NYXHIP_GABOR_PHASE_EXITS): leave after phase 1..4; 0 otherwise
    SpillArgs sp;
    ShapeLayout L;
};

// ---- device-side ROI assembly for a stack of tiles (tile_assembly.hip) -------------------------
struct TileHash {        // per-tile open-addressing tables, each [n_tiles * cap]; key = label, 0 = empty slot
    uint32_t *key, *cnt, *vmin, *vmax, *xmin, *xmax, *ymin, *ymax;
    uint32_t cap;        // slots per tile (a power of two)
    uint32_t shift;      // 32 - log2(cap)
};
struct TileRows {        // one entry per ROI ([max_rows], px_offset [max_rows + 1]); sorted rows: (tile, label) ascending
    uint32_t *tile, *label, *area;
    uint64_t* px_offset;
    uint32_t *bbox_x0, *bbox_y0, *bbox_w, *bbox_h, *vmin, *vmax;
    double *slide_min, *slide_max;
};
int launch_tile_assembly_scan(const void* inten, int dt_inten, const void* label, int dt_label, uint32_t W, uint32_t H, uint32_t n_tiles,
                              TileHash T, TileRows U, TileRows R, uint32_t max_rows, uint32_t* meta, uint32_t* blk_rows, unsigned long long* blk_px,
                              uint32_t* tile_row_begin, unsigned long long* tile_px_begin, void* stream);
int launch_tile_rank(TileRows U, const uint32_t* tile_row_begin, const unsigned long long* tile_px_begin, TileRows R, uint32_t max_rows, uint32_t n_tiles,
                     uint32_t max_rows_per_tile, int slide_mode, const double* smin, const double* smax, void* stream);
int launch_tile_clouds(const void* inten, int dt_inten, const void* label, int dt_label, uint32_t W, uint32_t H, TileRows R, uint32_t n_roi,
                       uint16_t* cx, uint16_t* cy, uint32_t* cv, void* stream);

// ---- large-ROI path: INTENSITY + GLCM of an ROI by several workgroups (roi_large.hip) ---------------------------------------------
// The reference gives any ROI to any worker (/root/reference/src/nyx/parallel.h:23-42); here an ROI beyond the LDS classes is cut
// into slabs of its pixel cloud (load pass) and strips of its bounding-box plane (co-occurrence pass), every slab / strip a
// workgroup of its own, and everything the slabs exchange is an INTEGER added with atomics into the ROI's block of a global
// workspace -- the intensity histogram over [min, max], the two exact sums, the co-occurrence counts -- so the result does not
// depend on how many workgroups shared the ROI or in which order they arrived.  A last kernel (one workgroup per ROI) derives the
// 36 + 30 n_angles + 29 columns from that state: every first-order feature is a function of the histogram alone.
constexpr uint32_t kLargeRangeMax = 1u << 22;     // histogram entries per ROI (wider ranges: the one-workgroup sort path of roi_features.hip)
constexpr uint32_t kLargeCells = 8192;            // plane cells per co-occurrence workgroup (whole rows)
struct LargeWs {                                  // byte offsets inside an ROI's workspace block
    uint64_t tab, lvl, plane, P, scr, total;
};
constexpr uint32_t kLargeScratchLds = 48 * 1024;  // GLCM feature scratch of the finishing kernel: in LDS up to this size, else in the ROI's block
__host__ __device__ inline uint64_t large_glcm_scratch_bytes(uint32_t ng) { return 8ull * ((uint64_t)ng + kMaxAngles * 32ull + kMaxAngles * 6ull * ng); }
// ng: bound of the ROI's matrix order (grey depth, or its largest intensity under IBSI); lvl_cap: radiomics bin count (0: none)
__host__ __device__ inline LargeWs large_ws_layout(uint32_t range, uint64_t area, uint32_t ng, uint32_t lvl_cap, uint32_t na, bool plane16, bool do_int, bool do_glcm)
{
    LargeWs L;
    uint64_t o = 256;                                                        // header: u64 sum, u64 sum of squares (mod 2^32 each)
    L.tab = o; if (do_int) o += (4ull * ((uint64_t)range + 1) + 255) & ~255ull;
    L.lvl = o; if (do_glcm && lvl_cap) o += (2ull * (lvl_cap + 8) + 255) & ~255ull;
    L.plane = o; if (do_glcm) o += ((plane16 ? 2ull : 1ull) * area + 64 + 255) & ~255ull;
    L.P = o; if (do_glcm) o += (4ull * na * ng * ng + 255) & ~255ull;
    L.scr = o; if (do_glcm && large_glcm_scratch_bytes(ng) > kLargeScratchLds) o += (large_glcm_scratch_bytes(ng) + 255) & ~255ull;
    L.total = o;
    return L;
}
struct LargeArgs {
    uint64_t n_roi;
    const uint64_t* px_offset;
    const uint16_t* x;
    const uint16_t* y;
    const uint32_t* inten;
    const uint32_t* bbox_w;
    const uint32_t* bbox_h;
    const uint32_t* min_inten;
    const uint32_t* max_inten;
    const double* slide_min;
    const double* slide_max;
    double* out;
    uint64_t ld;
    int* status;
    uint32_t mask;                 // subset of INTENSITY | GLCM
    int32_t n_cols, col_intensity, col_glcm;
    double soft_nan;
    int32_t grey_depth, ibsi, glcm_grey_depth, glcm_offset, glcm_na, glcm_symmetric;
    int32_t glcm_angles[kMaxAngles];
    int32_t n_hist;
    const uint32_t* list;          // the members of this launch group (ROI indices)
    uint32_t n_list;
    unsigned char* ws;             // workspace of the group, zeroed before the prep kernel
    uint64_t ws_bytes;
    uint64_t* ws_off;              // [n_list] byte offset of a member's block (prep kernel); ~0: not served by this path
    uint32_t* ctr;                 // [8] zeroed: 0-1 u64 cursor of workspace bytes, 2 load workgroups, 3 co-occurrence workgroups
    uint2* map_load;               // [cap_load] (member, slab) of a load workgroup
    uint2* map_cooc;               // [cap_cooc] (member, strip) of a co-occurrence workgroup
    uint32_t cap_load, cap_cooc;
    uint32_t px_per_wg;            // pixels of a slab
    uint32_t tab_lds;              // histogram entries the load kernel counts in LDS (16-bit counters): ROIs with range < tab_lds
    uint32_t plane16;              // 1: the plane holds 16-bit levels
    uint32_t lds_P_bytes;          // LDS the co-occurrence kernel may use for its matrices (+ the radiomics level map)
    uint32_t lds_strip_bytes;      // ... and for the staged strip of plane rows
    uint32_t lds_fin_bytes;        // dynamic LDS of the finishing kernel
    uint32_t fin_tab_bytes;        // ... of which the histogram may take this much (16-byte multiple; ROIs whose table fits are finished from LDS)
    uint32_t fin_P_bytes;          // ... of which the matrices may take this much (staged from the workspace; 0: read in place)
    uint32_t vec_ok;               // 1: inten is 16-byte and x / y are 8-byte aligned (groups of four pixels load as vectors)
    uint32_t dbg;                  // timing experiments only (NYXHIP_LARGE_DBG): 1 no plane stores, 2 no histogram counts, 4 no table flush, 8 no sums
};
int launch_large_features(const LargeArgs& a, void* stream);

// ---- large-ROI path, texture families: GLRLM + GLSZM + NGTDM of an ROI by several workgroups (roi_large_tex.hip) -------------------
// Same recipe as above: the binned plane of an ROI beyond the LDS classes goes to its block of a global workspace (load pass, slabs
// of the cloud), strips of plane rows are workgroups of their own, and what crosses a strip is an integer: the NGTDM
// accumulators (counts and sums of |i - mean| in units of 1/840), the run-length counts, the runs that touch a strip's first or
// last row (recorded per column and joined by the finishing workgroup), the zone sizes at their owner pixels.  The GLSZM owner
// sweep (a chain over the rows, features/glszm.cpp:108-185) is one wave per ROI beside the strips.
constexpr uint32_t kLtexCells = 8192;             // plane cells per strip workgroup (whole rows)
constexpr uint32_t kLtexMaxW = 8192;              // widest box the strip kernels stage
constexpr uint32_t kLtexLevels = 4094;            // largest level (grey depth, or intensity under IBSI): 12 bits next to a 20-bit length / label
constexpr uint32_t kLtexSmall = 32;               // GLSZM: zones of up to this many pixels are counted in a direct [level][size] table
constexpr uint32_t kLtexRlmLds = 16;              // GLRLM: runs of up to this many pixels are counted in LDS first
// served by this path: non-empty, box at most kLtexMaxW wide, fewer than 2^20 cells (owner labels travel in 20 bits).  A function
// of the ROI alone; everybody else takes the one-workgroup workspace launch of roi_texture.hip.
__host__ __device__ inline bool ltex_eligible(uint32_t n, uint32_t w, uint32_t h)
{
    return n != 0 && w != 0 && h != 0 && w <= kLtexMaxW && (uint64_t)w * h < (1ull << 20);
}
__host__ __device__ inline uint32_t ltex_rows_per_strip(uint32_t w, uint32_t h)
{
    uint32_t r = kLtexCells / (w ? w : 1u);
    if (r < 1u) r = 1u;
    return r < h ? r : (h ? h : 1u);
}
struct LtexWs {                                   // byte offsets inside an ROI's workspace block (header: 256 bytes of u32 words)
    uint64_t flags, plane, ngt, rlm, rec, cnt, small, hash, big, lab, open, total;
    uint32_t rows, K;                             // rows per strip, strips
    uint32_t slot_words;                          // words of one direction's run-length matrix + its marginals
    uint32_t hcap, S, big_cap;                    // GLSZM: hash slots, direct-table sizes (0 or kLtexSmall), entries of the list of larger zones
    uint32_t open_cap;                            // ... entries of the list of zones that reach their strip's last row
};
enum { LTEX_H_NP_ORIG = 0, LTEX_H_NP_BIN, LTEX_H_NZONE, LTEX_H_SZMAX, LTEX_H_NBIG, LTEX_H_NOPEN };   // header words
// ng: bound of the ROI's level count (grey depth, or its largest intensity under IBSI)
__host__ __device__ inline LtexWs ltex_ws_layout(uint32_t w, uint32_t h, uint32_t ng, bool plane16, uint32_t mask)
{
    LtexWs L;
    const uint64_t area = (uint64_t)w * h;
    const uint32_t side = w > h ? w : h;
    L.rows = ltex_rows_per_strip(w, h);
    L.K = (h + L.rows - 1) / L.rows;
    auto al = [](uint64_t v) { return (v + 255) & ~255ull; };
    uint64_t o = 256;
    L.flags = o; o += al((uint64_t)ng + 8);
    L.plane = o; o += al((plane16 ? 2ull : 1ull) * area + 64);
    L.ngt = o; if (mask & NYXHIP_FAM_NGTDM) o += al(12ull * ((uint64_t)ng + 2));            // u64 S[ng + 2] | u32 N[ng + 2]
    L.slot_words = ng * side + ng + side + 4;
    L.rlm = o; if (mask & NYXHIP_FAM_GLRLM) o += al(16ull * L.slot_words);
    L.rec = o; if (mask & NYXHIP_FAM_GLRLM) o += al(24ull * L.K * w);                      // [K][2][3][w] u32: runs touching a strip's first / last row
    L.hcap = 0; L.S = 0; L.big_cap = 0; L.open_cap = 0;
    L.cnt = o; L.small = o; L.hash = o; L.big = o; L.lab = o; L.open = o;
    if (mask & NYXHIP_FAM_GLSZM) {
        L.hcap = szm_hash_cap(ng + 1, (uint32_t)area);
        L.S = ng <= 256 ? kLtexSmall : 0u;
        L.big_cap = (uint32_t)(area / (L.S + 1)) + 8;
        L.cnt = o; o += al(4ull * (area + 2));
        L.small = o; o += al(4ull * ng * L.S);
        L.hash = o; o += al(8ull * L.hcap);
        L.big = o; o += al(4ull * L.big_cap);
        L.open_cap = L.K * w + 8;
        L.open = o; o += al(4ull * L.open_cap);
        L.lab = o; o += al(4ull * (area + 2));                                             // owner of every cell (written whole by the sweep)
    }
    L.total = o;
    return L;
}
struct LtexArgs {
    uint64_t n_roi;
    const uint64_t* px_offset;
    const uint16_t* x;
    const uint16_t* y;
    const uint32_t* inten;
    const uint32_t* bbox_w;
    const uint32_t* bbox_h;
    const uint32_t* min_inten;
    const uint32_t* max_inten;
    double* out;
    uint64_t ld;
    int* status;
    uint32_t mask;                 // subset of GLRLM | GLSZM | NGTDM
    int32_t n_cols, col0, gap_after_glrlm, gap_after_glszm;   // as TexArgs
    double soft_nan;
    int32_t grey_depth, ibsi;
    const uint32_t* list;          // the members of this launch group (ROI indices)
    uint32_t n_list;
    unsigned char* ws;             // workspace of the group, zeroed before the prep kernel
    uint64_t ws_bytes;
    uint64_t* ws_off;              // [n_list] byte offset of a member's block (prep kernel); ~0: not served by this path
    uint32_t* ctr;                 // [8] zeroed: 0-1 u64 cursor of workspace bytes, 2 load workgroups, 3 strip workgroups
    uint2* map_load;               // [cap_load] (member, slab)
    uint2* map_strip;              // [cap_strip] (member, strip)
    uint32_t cap_load, cap_strip;
    uint32_t px_per_wg;            // pixels of a slab
    uint32_t plane16;              // 1: the plane holds 16-bit levels
    uint32_t vec_ok;               // as LargeArgs
    uint32_t lds_load_bytes;       // dynamic LDS of the load kernel: its transposing tile
    uint32_t strip_threads;        // workgroup size of the strip / sweep kernel: a wave per 64 columns of the class's widest box, 4 .. 16 waves
    uint32_t strip_groups;         // strips per strip workgroup: one per group of four waves
    uint32_t lds_group_bytes;      // dynamic LDS of one such group
    uint32_t lds_strip_bytes;      // dynamic LDS of the strip / sweep kernel
    uint32_t lds_zone_bytes;       // ... of the join / zone kernel
    uint32_t lds_fin_bytes;        // dynamic LDS of the finishing kernel
};
int launch_large_texture(const LtexArgs& a, void* stream);

// ---- LDS-sized ROIs of 16-bit data: first-order features without a sort (roi_wide.hip) ----------------------------------------------
constexpr int kWideDupCap = 256;   // duplicate values the fast path lists per ROI (more: the sort-based slow path of the same kernel)
struct WideArgs {
    const uint64_t* px_offset;
    const uint32_t* inten;
    const uint32_t* min_inten;
    const uint32_t* max_inten;
    const double* slide_min;
    const double* slide_max;
    double* out;
    uint64_t ld;
    int* status;
    int32_t col_intensity, n_hist;
    const uint32_t* list;          // the members of the class (ROI indices)
    uint32_t n_list;
    uint32_t key_cap;              // keys the carve-out holds
    uint32_t o_keys, o_work, slow_hist, o_lb, lds_bytes;   // byte offsets: keys | bitmap + prefixes + lists (slow path: second keys, counts, digit counts at slow_hist) | bin bounds
};
bool make_wide_layout(uint32_t max_px, uint32_t n_hist, WideArgs& a);
// ---- Gabor of ROIs beyond LDS: several workgroups per ROI (roi_large_gabor.hip) ------------------------------------------------------
constexpr int kLgabTileW = 64, kLgabTileH = 32;       // output pixels of a tile (one workgroup)
constexpr uint32_t kLgabSlab = 8192;                  // cloud pixels per workgroup of the plane kernel
constexpr uint32_t kLgabMaxSide = 8192, kLgabMaxN = 32;
struct LgabArgs {
    uint64_t n_roi;
    const uint64_t* px_offset;
    const uint16_t* x;
    const uint16_t* y;
    const uint32_t* inten;
    const uint32_t* bbox_w;
    const uint32_t* bbox_h;
    const uint32_t* min_inten;
    const uint32_t* max_inten;
    double* out;
    uint64_t ld;
    int32_t col_gabor, nf, n;
    double thr, soft_nan;
    const double* bank;        // device: (nf + 1) filters (low-pass first), n * n complex taps each
    unsigned char* ws;         // per slot of the launch: u32 plane [area_cap] | tile records [tiles_cap][3] doubles | counters [nf]
    uint64_t stride, off_rec, off_cnt;
    uint32_t tiles_cap;        // tiles of the largest box of the launch (grid.x)
    SpillArgs sp;              // roi_index list + n_slots
};
size_t lgab_lds_bytes(int n);
int launch_large_gabor(const LgabArgs& a, void* stream, uint32_t n_slots, uint32_t max_px);

int launch_roi_wide(const WideArgs& a, void* stream);

// implemented in roi_features.hip / roi_texture.hip / roi_shape.hip
int launch_roi_features(const RoiArgs& a, void* stream, uint32_t grid);
bool roi_small_supported(const RoiArgs& a);
int launch_roi_small(const RoiArgs& a, void* stream, uint32_t n_slots, bool promised);
int launch_roi_texture(const TexArgs& a, void* stream, uint32_t grid);
int launch_roi_shape(const ShapeArgs& a, void* stream, uint32_t grid);
int launch_roi_dependence(const DepArgs& a, void* stream, uint32_t grid);
int launch_roi_contour(const MomArgs& a, void* stream, uint32_t grid);
int launch_roi_moments(const MomArgs& a, void* stream, uint32_t grid);
int launch_moments_logtab(double* tab, uint32_t n, void* stream);
size_t roi_features_max_lds();

} // namespace nyxhip
