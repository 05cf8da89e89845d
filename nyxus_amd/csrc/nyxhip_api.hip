// nyxhip_api.hip -- the C ABI of include/nyxhip.h: context, staging, launch, errors.
//
// Host-side counterpart of reduce_trivial_rois_manual()
// (/root/reference/src/nyx/reduce_trivial_rois.cpp:772-795): instead of fanning a
// label vector out over std::async threads per feature family (parallel.h:23-42),
// one fused kernel launch covers every requested family for the whole ROI batch.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <string>
#include <vector>
#include <algorithm>
#include <thread>
#include <atomic>

#include "../../include/nyxhip.h"
#include "roi_kernel.h"

using namespace nyxhip;

struct Extrema {
    uint32_t px, area, range, side;
    uint32_t vmax = 0;           // largest intensity (0: not known -- stated extrema carry none)
    bool wide_only = false;      // every ROI of the group has an intensity range beyond the counting tables (a wide-range size class)
};
struct ClassRun {              // one size class of one call, as launched (nyxhip_launch_report)
    int cls;                   // 2 * size class + (1: some ROI needs 32-bit tables); -1: the whole batch in one launch group
    uint32_t count;            // members (0xFFFFFFFF: counted on the device only)
    Extrema E;                 // extrema the carve-outs were sized for
    int workspace;             // kernel groups that ran from a global workspace instead of LDS: bit 0 INTENSITY + GLCM, 1 texture, 2 shape, 3 dependence
    hipEvent_t e0, e1;         // around the class's launches on the main stream (timing enabled), else NULL
    hipEvent_t e2 = nullptr;   // ... and the end of its launches on its workspace lane
    int cooperative = 0;       // bit 0: INTENSITY + GLCM by the several-workgroups-per-ROI kernels of roi_large.hip, bit 1: the texture families (roi_large_tex.hip)
};
struct ClassTotals {           // sums over the members of a class (class header): what the large-ROI path sizes its workspace from
    uint64_t px, area, range1; // pixels, bounding-box cells, histogram entries (range + 1 of the members whose range the path serves)
};
constexpr int NYXHIP_INTERNAL_NEEDS_CLOUDS = -1000;   // run_class: a launch group of a window-mode call needs the materialised clouds

struct nyxhip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t user_stream = nullptr;
    bool use_user_stream = false;
    int* d_status = nullptr;           // [0] error flag of the kernels | [1] census: ROIs of <= 256 px met by the scanning form of roi_small_kernel (read with the flag)
    // census of the recent calls: how many of the ROIs were of the smallest size class.  A batch on stated extrema that mixes that class
    // with the next one runs either as two filtered whole-batch launches (nothing counted, no host round trip: right when the class is
    // rare -- the metric configuration) or through the exact class lists (right when it is common: filtered launches spend a workgroup
    // on every slot they skip).  Both give the same rows; the census only picks the cheaper one.  Host batches are counted on the host.
    uint64_t census_small = 0, census_total = 0, census_pending = 0;
    // Gabor filter bank (host-built, gabor.cpp:393-449), re-uploaded when the settings change
    double* d_bank = nullptr;
    float* d_bank32 = nullptr;       // the bank rounded to fp32 (Gabor screening pass)
    void* d_bank16 = nullptr;        // 16 x 16 banks: the band-pass filters as f16 B operands of the MFMA screening stage (ShapeArgs::gabor_bank16)
    std::vector<double> bank_key;
    uint32_t bank_zero_rows[NYXHIP_MAX_GABOR_FILTERS + 1] = {};   // ShapeArgs::gabor_zero_rows of the uploaded bank (16 x 16 kernels)
    uint32_t bank_lp_sep = 0;          // ShapeArgs::gabor_lp_sep / _B / _C of the uploaded bank
    float bank_lp_B[16] = {}, bank_lp_C[44] = {};
    uint32_t bank_box_mask = 0;                                  // ShapeArgs::gabor_box_mask of the uploaded bank
    unsigned long long* d_stamps = nullptr; // diagnostic (NYXHIP_STAMPS=1 + -DNYX_STAMP build): [32] phase cycle sums
    std::string err;
    // grow-only device staging for host-memory batches
    void* d_stage = nullptr;
    size_t stage_bytes = 0;
    // split GLCM: exported co-occurrence counts + matrix orders (grow-only)
    uint32_t* d_glcm_ws = nullptr;
    uint32_t* d_glcm_ng = nullptr;   // [n_roi] matrix order of every ROI whose counts were exported in the CURRENT call (0: none) -- cleared per call
    size_t glcm_ng_bytes = 0;
    double* d_logtab = nullptr;      // moments: log(sqrt(d) + 0.001) per integer squared distance (roi_moments.hip)
    uint32_t logtab_n = 0;
    size_t glcm_ws_bytes = 0;
    // contour + moments workspace (grow-only): contour points, contour lengths, per-pixel log distances
    void* d_mom = nullptr;
    size_t mom_bytes = 0;
    // contour planes beyond LDS: index list (launch_moments); per-workgroup global scratch of every workspace launch
    uint32_t* d_spill_list = nullptr;
    size_t spill_list_bytes = 0;
    unsigned char* d_spill = nullptr;
    size_t spill_bytes = 0;
    // grow-only workspaces of the fused tile path: scan tables + rows | clouds | two staging slots for host tiles
    void* d_tile = nullptr;
    size_t tile_bytes = 0;
    void* d_cloud = nullptr;
    size_t cloud_bytes = 0;
    void* d_slot[2] = {nullptr, nullptr};
    size_t slot_bytes[2] = {0, 0};
    hipStream_t copy_stream = nullptr;         // H2D of the next chunk runs beside the kernels of the current one
    hipEvent_t slot_ready[2] = {nullptr, nullptr}, slot_free[2] = {nullptr, nullptr};
    // pinned staging ring of the host tile path (HostStager below): the library's own page-locked memory between a pageable
    // caller and the DMA engine
    static constexpr int kStageSlots = 4;
    static constexpr size_t kStageSlotBytes = (size_t)32 << 20;
    void* h_stage[kStageSlots] = {};
    hipEvent_t h_stage_done[kStageSlots] = {};
    bool h_stage_used[kStageSlots] = {};
    int h_stage_next = 0;
    WindowSrc win_next = {};                   // set by the tile path for its next launch_device call: read ROIs from their tile windows
    uint32_t tile_cap_hint = 0;                // per-tile table size that served the last call
    // result kept for nyxhip_fetch_result() (host-memory calls with out_table == NULL): device-resident, grow-only
    //   [res_cap x res_cols] doubles | [res_cap] labels | [res_cap] tile indices
    void* d_res = nullptr;
    size_t res_cap = 0, res_rows = 0, res_cols = 0;
    double* res_table() const { return (double*)d_res; }
    uint32_t* res_label() const { return (uint32_t*)((char*)d_res + (((size_t)res_cap * res_cols * 8 + 255) & ~(size_t)255)); }
    uint32_t* res_tile() const { return res_label() + res_cap; }
    // size classes of a call (launch_device_all): ROI indices grouped by class, class headers on the device and their pinned host copy
    uint32_t* d_cls_list = nullptr;
    size_t cls_list_bytes = 0;
    uint32_t* d_cls_hdr = nullptr;
    uint32_t* h_cls_hdr = nullptr;
    std::vector<ClassRun> runs;         // the classes of the last call as launched (nyxhip_launch_report)
    // large-ROI path (roi_large.hip): per-ROI blocks of histogram / plane / matrices, and the work maps + offsets + counters
    // Workspace lanes: the one-workgroup-per-ROI launches of a large class are a chain of dependent passes per ROI (milliseconds)
    // by a few hundred workgroups at most -- a fraction of the chip.  Each large class runs them on a stream of its own beside the
    // main stream (which goes on with the several-workgroups-per-ROI kernels and the LDS classes), with scratch of its own; the
    // lanes are forked from the main stream at the start of a call and joined into it at its end.
    static constexpr int kLanes = 12;              // 0-3: the large classes; 4-6: the LDS size classes of an exact call (run_class);
                                                   // 8, 10: contour + moments of a batch with boxes beyond LDS (the bulk | the big boxes);
                                                   // 9: the dependence trio of a large class; 11: Gabor of size class 2 beside the smaller classes
    static constexpr int kMomLane = 8, kDepLane = 9, kMomLaneBig = 10, kGaborLane = 11;
    hipStream_t lane_stream[kLanes] = {};
    hipEvent_t lane_done[kLanes] = {};
    hipEvent_t lane_fork = nullptr;
    unsigned char* lane_buf[kLanes] = {};
    size_t lane_bytes[kLanes] = {};
    bool lane_used[kLanes] = {};
    // ... and of the texture families (roi_large_tex.hip): one pair per lane (+ one for the main stream), the lanes run side by side
    void* ltex_buf[kLanes + 1] = {};
    size_t ltex_bytes[kLanes + 1] = {};
    void* ltex_aux[kLanes + 1] = {};
    size_t ltex_aux_bytes[kLanes + 1] = {};
    void* d_large = nullptr;
    size_t large_bytes = 0;
    void* d_large_aux = nullptr;
    size_t large_aux_bytes = 0;
    // timing
    int timing = 0;            // 0 off | 1 two events around every call (nyxhip_timing_get) | 2 also two events around every launch group (nyxhip_launch_report's ms)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t ev_used = 0;
    hipStream_t stream() const { return use_user_stream ? user_stream : own_stream; }
};

namespace {

thread_local std::string g_init_error;
std::atomic<int> g_ctx_on_device[64];          // live contexts per device (default memory budgets are shared among them)

int fail(nyxhip_ctx* ctx, int code, const std::string& msg)
{
    if (ctx)
        ctx->err = msg;
    else
        g_init_error = msg;
    return code;
}

#define HIP_TRY(ctx, call)                                                                    \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(ctx, NYXHIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// ---- column catalogue (Feature2D enum order; names = user-facing feature names,
// src/nyx/featureset.cpp UserFacingFeatureNames) -------------------------------------
const char* kIntensityNames[kIntensityCols] = {
    "COV", "COVERED_IMAGE_INTENSITY_RANGE", "ENERGY", "ENTROPY", "EXCESS_KURTOSIS", "HYPERFLATNESS",
    "HYPERSKEWNESS", "INTEGRATED_INTENSITY", "INTERQUARTILE_RANGE", "KURTOSIS", "MAX", "MEAN",
    "MEAN_ABSOLUTE_DEVIATION", "MEDIAN", "MEDIAN_ABSOLUTE_DEVIATION", "MIN", "MODE", "P01", "P10", "P25",
    "P75", "P90", "P99", "QCOD", "RANGE", "ROBUST_MEAN", "ROBUST_MEAN_ABSOLUTE_DEVIATION",
    "ROOT_MEAN_SQUARED", "SKEWNESS", "STANDARD_DEVIATION", "STANDARD_DEVIATION_BIASED", "STANDARD_ERROR",
    "VARIANCE", "VARIANCE_BIASED", "UNIFORMITY", "UNIFORMITY_PIU"};
const char* kGlcmNames[kGlcmAngled] = {
    "GLCM_ASM", "GLCM_ACOR", "GLCM_CLUPROM", "GLCM_CLUSHADE", "GLCM_CLUTEND", "GLCM_CONTRAST",
    "GLCM_CORRELATION", "GLCM_DIFAVE", "GLCM_DIFENTRO", "GLCM_DIFVAR", "GLCM_DIS", "GLCM_ENERGY",
    "GLCM_ENTROPY", "GLCM_HOM1", "GLCM_HOM2", "GLCM_ID", "GLCM_IDN", "GLCM_IDM", "GLCM_IDMN",
    "GLCM_INFOMEAS1", "GLCM_INFOMEAS2", "GLCM_IV", "GLCM_JAVE", "GLCM_JE", "GLCM_JMAX", "GLCM_JVAR",
    "GLCM_SUMAVERAGE", "GLCM_SUMENTROPY", "GLCM_SUMVARIANCE", "GLCM_VARIANCE"};
const char* kGlcmAveNames[kGlcmAve] = {
    "GLCM_ASM_AVE", "GLCM_ACOR_AVE", "GLCM_CLUPROM_AVE", "GLCM_CLUSHADE_AVE", "GLCM_CLUTEND_AVE",
    "GLCM_CONTRAST_AVE", "GLCM_CORRELATION_AVE", "GLCM_DIFAVE_AVE", "GLCM_DIFENTRO_AVE", "GLCM_DIFVAR_AVE",
    "GLCM_DIS_AVE", "GLCM_ENERGY_AVE", "GLCM_ENTROPY_AVE", "GLCM_HOM1_AVE", "GLCM_ID_AVE", "GLCM_IDN_AVE",
    "GLCM_IDM_AVE", "GLCM_IDMN_AVE", "GLCM_IV_AVE", "GLCM_JAVE_AVE", "GLCM_JE_AVE", "GLCM_INFOMEAS1_AVE",
    "GLCM_INFOMEAS2_AVE", "GLCM_VARIANCE_AVE", "GLCM_JMAX_AVE", "GLCM_JVAR_AVE", "GLCM_SUMAVERAGE_AVE",
    "GLCM_SUMENTROPY_AVE", "GLCM_SUMVARIANCE_AVE"};

const char* kGlrlmNames[16] = {"GLRLM_SRE", "GLRLM_LRE", "GLRLM_GLN", "GLRLM_GLNN", "GLRLM_RLN", "GLRLM_RLNN", "GLRLM_RP",
                               "GLRLM_GLV", "GLRLM_RV", "GLRLM_RE", "GLRLM_LGLRE", "GLRLM_HGLRE", "GLRLM_SRLGLE",
                               "GLRLM_SRHGLE", "GLRLM_LRLGLE", "GLRLM_LRHGLE"};
const char* kGlszmNames[16] = {"GLSZM_SAE", "GLSZM_LAE", "GLSZM_GLN", "GLSZM_GLNN", "GLSZM_SZN", "GLSZM_SZNN", "GLSZM_ZP",
                               "GLSZM_GLV", "GLSZM_ZV", "GLSZM_ZE", "GLSZM_LGLZE", "GLSZM_HGLZE", "GLSZM_SALGLE",
                               "GLSZM_SAHGLE", "GLSZM_LALGLE", "GLSZM_LAHGLE"};
const char* kGldzmNames[18] = {"GLDZM_SDE", "GLDZM_LDE", "GLDZM_LGLZE", "GLDZM_HGLZE", "GLDZM_SDLGLE", "GLDZM_SDHGLE", "GLDZM_LDLGLE",
                               "GLDZM_LDHGLE", "GLDZM_GLNU", "GLDZM_GLNUN", "GLDZM_ZDNU", "GLDZM_ZDNUN", "GLDZM_ZP", "GLDZM_GLM",
                               "GLDZM_GLV", "GLDZM_ZDM", "GLDZM_ZDV", "GLDZM_ZDE"};
const char* kGldmNames[14] = {"GLDM_SDE", "GLDM_LDE", "GLDM_GLN", "GLDM_DN", "GLDM_DNN", "GLDM_GLV", "GLDM_DV", "GLDM_DE", "GLDM_LGLE",
                              "GLDM_HGLE", "GLDM_SDLGLE", "GLDM_SDHGLE", "GLDM_LDLGLE", "GLDM_LDHGLE"};
const char* kNgldmNames[19] = {"NGLDM_LDE", "NGLDM_HDE", "NGLDM_LGLCE", "NGLDM_HGLCE", "NGLDM_LDLGLE", "NGLDM_LDHGLE", "NGLDM_HDLGLE",
                               "NGLDM_HDHGLE", "NGLDM_GLNU", "NGLDM_GLNUN", "NGLDM_DCNU", "NGLDM_DCNUN", "NGLDM_DCP", "NGLDM_GLM",
                               "NGLDM_GLV", "NGLDM_DCM", "NGLDM_DCV", "NGLDM_DCENT", "NGLDM_DCENE"};
const char* kNgtdmNames[5] = {"NGTDM_COARSENESS", "NGTDM_CONTRAST", "NGTDM_BUSYNESS", "NGTDM_COMPLEXITY", "NGTDM_STRENGTH"};
const int kGlrlmAngles[4] = {0, 45, 90, 135}; // GLRLMFeature::rotAngles, glrlm.h:134

// families the kernels cover so far
constexpr uint32_t kTexture = NYXHIP_FAM_GLRLM | NYXHIP_FAM_GLSZM | NYXHIP_FAM_NGTDM;
constexpr uint32_t kShape = NYXHIP_FAM_GABOR | NYXHIP_FAM_ZERNIKE;
constexpr uint32_t kDependence = NYXHIP_FAM_GLDZM | NYXHIP_FAM_GLDM | NYXHIP_FAM_NGLDM;
constexpr uint32_t kMoments = NYXHIP_FAM_SMOMS | NYXHIP_FAM_IMOMS;
constexpr uint32_t kImplemented = NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM | kTexture | kShape | kDependence | kMoments;

bool settings_ok(const nyxhip_settings* s, uint32_t mask, std::string& why)
{
    if (!s) { why = "settings is NULL"; return false; }
    if (mask & NYXHIP_FAM_INTENSITY) {
        if (s->grey_depth == 0) { why = "grey_depth must be non-zero (histogram bin count)"; return false; }
    }
    if (mask & NYXHIP_FAM_GLCM) {
        if (s->glcm_n_angles < 0 || s->glcm_n_angles > NYXHIP_MAX_GLCM_ANGLES) { why = "glcm_n_angles out of range"; return false; }
        for (int i = 0; i < s->glcm_n_angles; i++) {
            int a = s->glcm_angles[i];
            if (a != 0 && a != 45 && a != 90 && a != 135) { why = "unsupported GLCM angle (glcm.cpp:252-254)"; return false; }
        }
        if (s->glcm_offset < 0) { why = "glcm_offset must be >= 0"; return false; }
    }
    if ((mask & NYXHIP_FAM_NGLDM) && !s->ibsi && s->grey_depth < 0) {
        // ngldm.cpp:201 passes GREYDEPTH as unsigned: a negative depth becomes ~4.29e9 levels (one per intensity)
        why = "NGLDM with a negative (radiomics) grey depth is not supported";
        return false;
    }
    if ((mask & NYXHIP_FAM_GLDZM) && !s->ibsi && s->grey_depth < 0) {
        // radiomics binning leaves level-0 background zones: the reference writes them one row past its matrix
        // (gldzm.cpp:44-50) and its distances depend on the flood order (zeros turn VISITED, :111-116) -- undefined there
        why = "GLDZM with a negative (radiomics) grey depth is not supported (undefined in the reference)";
        return false;
    }
    if (mask & NYXHIP_FAM_GABOR) {
        if (s->gabor_n_filters < 0 || s->gabor_n_filters > NYXHIP_MAX_GABOR_FILTERS) { why = "gabor_n_filters out of range"; return false; }
        if (s->gabor_kersize < 1 || s->gabor_kersize > 64) { why = "gabor_kersize out of range (1..64)"; return false; }
    }
    return true;
}

std::vector<std::string> column_names(uint32_t mask, const nyxhip_settings* s)
{
    std::vector<std::string> v;
    if (mask & NYXHIP_FAM_INTENSITY)
        for (auto n : kIntensityNames) v.push_back(n);
    if (mask & NYXHIP_FAM_GLCM) {
        for (auto n : kGlcmNames)
            for (int a = 0; a < s->glcm_n_angles; a++)
                v.push_back(std::string(n) + "_" + std::to_string(s->glcm_angles[a])); // output_2_buffer.cpp:336-343
        for (auto n : kGlcmAveNames) v.push_back(n);
    }
    if (mask & NYXHIP_FAM_GLRLM) {
        for (auto n : kGlrlmNames)
            for (int a : kGlrlmAngles) v.push_back(std::string(n) + "_" + std::to_string(a)); // output_2_buffer.cpp:351-361
        for (auto n : kGlrlmNames) v.push_back(std::string(n) + "_AVE");
    }
    if (mask & NYXHIP_FAM_GLDZM)
        for (auto n : kGldzmNames) v.push_back(n);
    if (mask & NYXHIP_FAM_GLSZM)
        for (auto n : kGlszmNames) v.push_back(n);
    if (mask & NYXHIP_FAM_GLDM)
        for (auto n : kGldmNames) v.push_back(n);
    if (mask & NYXHIP_FAM_NGLDM)
        for (auto n : kNgldmNames) v.push_back(n);
    if (mask & NYXHIP_FAM_NGTDM)
        for (auto n : kNgtdmNames) v.push_back(n);
    if (mask & NYXHIP_FAM_GABOR)
        for (int i = 0; i < s->gabor_n_filters; i++) v.push_back("GABOR_" + std::to_string(i));       // output_2_buffer.cpp:364-373
    if (mask & NYXHIP_FAM_ZERNIKE)
        for (int i = 0; i < kZernikeCols; i++) v.push_back("ZERNIKE2D_Z" + std::to_string(i));        // :417-427
    if (mask & NYXHIP_FAM_SMOMS) {     // featureset.h:362-467
        const char* pq13[13] = {"00", "01", "02", "03", "10", "11", "12", "13", "20", "21", "22", "23", "30"};
        const char* pq7[7] = {"02", "03", "11", "12", "20", "21", "30"};
        const char* pq10[10] = {"00", "01", "02", "03", "10", "11", "12", "20", "21", "30"};
        for (auto k : pq13) v.push_back(std::string("SPAT_MOMENT_") + k);
        for (int p = 0; p < 4; p++) for (int q = 0; q < 4; q++) v.push_back("CENTRAL_MOMENT_" + std::to_string(p) + std::to_string(q));
        for (int p = 0; p < 4; p++) for (int q = 0; q < 4; q++) v.push_back("NORM_SPAT_MOMENT_" + std::to_string(p) + std::to_string(q));
        for (auto k : pq7) v.push_back(std::string("NORM_CENTRAL_MOMENT_") + k);
        for (int k = 1; k <= 7; k++) v.push_back("HU_M" + std::to_string(k));
        for (auto k : pq10) v.push_back(std::string("WEIGHTED_SPAT_MOMENT_") + k);
        for (auto k : pq7) v.push_back(std::string("WEIGHTED_CENTRAL_MOMENT_") + k);
        for (auto k : pq7) v.push_back(std::string("WT_NORM_CTR_MOM_") + k);
        for (int k = 1; k <= 7; k++) v.push_back("WEIGHTED_HU_M" + std::to_string(k));
    }
    if (mask & NYXHIP_FAM_IMOMS) {     // featureset.h:472-565
        const char* pq13[13] = {"00", "01", "02", "03", "10", "11", "12", "13", "20", "21", "22", "23", "30"};
        const char* pq7[7] = {"02", "03", "11", "12", "20", "21", "30"};
        const char* pq10[10] = {"00", "01", "02", "03", "10", "11", "12", "20", "21", "30"};
        for (auto k : pq13) v.push_back(std::string("IMOM_RM_") + k);
        for (int p = 0; p < 4; p++) for (int q = 0; q < 4; q++) v.push_back("IMOM_CM_" + std::to_string(p) + std::to_string(q));
        for (int p = 0; p < 4; p++) for (int q = 0; q < 4; q++) v.push_back("IMOM_NRM_" + std::to_string(p) + std::to_string(q));
        for (auto k : pq7) v.push_back(std::string("IMOM_NCM_") + k);
        for (int k = 1; k <= 7; k++) v.push_back("IMOM_HU" + std::to_string(k));
        for (auto k : pq10) v.push_back(std::string("IMOM_WRM_") + k);
        for (auto k : pq7) v.push_back(std::string("IMOM_WCM_") + k);
        for (auto k : pq7) v.push_back(std::string("IMOM_WNCM_") + k);
        for (int k = 1; k <= 7; k++) v.push_back("IMOM_WHU" + std::to_string(k));
    }
    return v;
}

uint32_t pow2ceil(uint32_t v)
{
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }

// Carves the workgroup's LDS for one launch.  Returns NYXHIP_OK, or
// NYXHIP_ERR_UNSUPPORTED when the grey depth alone cannot be held in LDS, or
// NYXHIP_ERR_ROI_TOO_LARGE when the batch extrema do not fit the 160 KiB of a CU.
int make_layout(uint32_t mask, const nyxhip_settings* s, int n_cols, uint32_t max_px, uint32_t max_area,
                uint32_t max_range, LdsLayout& L, std::string& why, size_t cap = 0, uint32_t vmax = 0, bool wide_only = false)
{
    memset(&L, 0, sizeof(L));
    const bool do_int = mask & NYXHIP_FAM_INTENSITY, do_glcm = mask & NYXHIP_FAM_GLCM;
    const bool spill = cap != 0;               // scratch in the global workspace: only the 2 GiB offset range limits it
    if (!spill) cap = roi_features_max_lds();
    uint32_t off = 0;
    L.out = off;                               // (the kernel writes its output row in place: no staging copy)
    L.red = off; off = align16(off + 8u * kWaves * 8);
    L.stat = off; off = align16(off + 8u * 16);
    L.lb100 = off; off = align16(off + 4u * 104);
    uint32_t n_hist = (uint32_t)abs(s->grey_depth);
    L.lbc = off; off = align16(off + 4u * (do_int ? n_hist + 8 : 8));
    const uint32_t fixed = off;               // everything that does not scale with the ROI
    // order-statistics engine (roi_features.hip): a counting table over [min, max] when the
    // batch's largest intensity range fits kCountCapMax entries -- then no ROI sorts and the
    // value buffer needs no power-of-two padding; otherwise ROIs with a small range still
    // count (table of kCountCapMixed) and the rest bitonic-sort a padded buffer.
    const uint32_t kCountCapMax = spill ? (1u << 22) : 16384u, kCountCapMixed = 4096;
    const bool radix = do_int && wide_only && !spill;   // every ROI sorts: LSD radix sort (roi_features.hip: radix_sort), no table, no padding
    if (radix) {
        L.count_cap = 0;
        L.sort_cap = max_px ? max_px : 1;
    } else if (do_int) {
        if ((uint64_t)max_range + 1 <= kCountCapMax) {
            L.count_cap = (max_range + 1 + 63u) & ~63u;
            L.sort_cap = max_px ? max_px : 1;
        } else {
            L.count_cap = kCountCapMixed;
            L.sort_cap = pow2ceil(max_px ? max_px : 1);
        }
    }
    // [val | cnt] is dead once the intensity block has finished, so the GLCM matrices and
    // their scratch alias the same bytes (the kernel separates the two uses by barriers);
    // the dense plane is written during the load phase and stays separate.
    L.dense_cap = do_glcm ? max_area : 0;
    L.dense = off;
    {   // 8-bit plane: matlab binning up to 16 levels in a launch that also gets the 16-bit tables and the split GLCM features
        // (build_args sets the split up under the same conditions) -- the carve-out of the benchmark ROI then fits 8 times per CU
        const int gi = s->ibsi ? 0 : s->grey_depth;
        const bool c16_pred = do_int && max_px < 65536u && (uint64_t)max_range + 1 <= kCountCapMax && max_range < 65536u;
        const bool split_pred = do_glcm && !spill && gi > 0 && gi <= 16 && s->glcm_n_angles > 0;
        L.dense8 = ((c16_pred || !do_int) && split_pred) ? 1u : 0u;           // (GLCM alone: nothing of the intensity block constrains the plane)
    }
    {   // the reference's default grey depth on LDS launches: 16-bit matrices + 8-bit plane (roi_features_kernel_g16)
        const int gi = s->ibsi ? 0 : s->grey_depth;
        const bool c16_pred = (!do_int || (uint64_t)max_range + 1 <= kCountCapMax) && max_range < 65536u;
        L.g16 = (do_glcm && !spill && gi > 16 && gi <= 64 && max_px < 32768u && c16_pred && s->glcm_n_angles > 0) ? 1u : 0u;
        if (L.g16) L.dense8 = 1;
    }
    if ((L.dense8 ? 1ull : 2ull) * L.dense_cap > cap) { why = "ROI bounding box of " + std::to_string(max_area) + " px exceeds the LDS-resident plane"; return NYXHIP_ERR_ROI_TOO_LARGE; }
    // (8-bit planes: + a zero row of 64 bytes + the out-of-box cell.  Grey-depth-64 launches: the plane is dead once the co-occurrence
    //  sweep is through, and the feature pass's scratch -- features, sums, row marginals: 4.6 KB -- takes its place: the carve-out
    //  of the benchmark ROI drops from 43.1 to 39.3 KB, four workgroups per CU instead of three)
    uint32_t plane_bytes = L.dense8 ? 1u * L.dense_cap + 64 + 8 : 2u * L.dense_cap + 8;
    const uint32_t g16_scratch = L.g16 ? 8u * ((uint32_t)s->grey_depth + kMaxAngles * 128u) : 0u;          // level values | a 1 KiB block per angle-wave (glcm_features_wave64_v2)
    if (plane_bytes < g16_scratch) plane_bytes = g16_scratch;
    off = align16(off + plane_bytes);
    if (do_glcm) {
        const int greyInfo = s->ibsi ? 0 : s->grey_depth;
        L.lvl_cap = greyInfo < 0 ? (uint32_t)(-greyInfo) : 0;
        L.lvlmap = off; off = align16(off + 2u * (L.lvl_cap + 8));
    }
    const uint32_t shared0 = off;
    L.val = off;
    // 16-bit tables: every ROI of the launch counts (range below the table) and has fewer than 65536 pixels, so counts,
    // per-wave prefix sums and the values' offsets from the ROI minimum all fit 16 bits
    L.cnt16 = (do_int && max_px < 65536u && (uint64_t)max_range + 1 <= kCountCapMax && max_range < 65536u) ? 1u : 0u;
    L.radix_k16 = (radix && max_range < 65536u) ? 1u : 0u;
    if ((L.cnt16 ? 2ull : 4ull) * L.sort_cap > cap) { why = "ROI pixel count " + std::to_string(max_px) + " exceeds the LDS-resident value buffer"; return NYXHIP_ERR_ROI_TOO_LARGE; }
    if (L.radix_k16) off = align16(off + 2u * 2u * ((L.sort_cap + 7u) & ~7u) + 16);      // two 16-bit key buffers
    else off = align16(off + (L.cnt16 ? 2u : 4u) * L.sort_cap + 16);
    L.cnt = off; off = align16(off + (L.cnt16 ? 2u : 4u) * L.count_cap + 16);
    if (radix) {                                          // (second 32-bit key buffer +) [4][256] digit counts + the four wave totals
        if (8ull * L.sort_cap > cap) { why = "ROI pixel count " + std::to_string(max_px) + " exceeds the LDS-resident sort buffers"; return NYXHIP_ERR_ROI_TOO_LARGE; }
        L.radix = off; off = align16(off + (L.radix_k16 ? 0u : 4u * L.sort_cap) + 4u * (kWaves * 256 + kWaves) + 16);
    }
    if (do_glcm && L.g16) {
        const uint32_t ng = (uint32_t)s->grey_depth, cellsw = ((ng + 1) * ((ng + 3) & ~1u)) / 2;     // rows 0..ng of an even pitch (roi_features.hip, G16 block)
        L.ng_cap = ng; L.app = 4;
        uint32_t goff = shared0;
        L.P = goff; goff = align16(goff + 4u * 4u * cellsw);
        L.gscr = L.dense;                                        // (unused) | features | sums | row marginals: over the dead plane
        if (goff > off) off = goff;
    } else if (do_glcm) {
        const int greyInfo = s->ibsi ? 0 : s->grey_depth;
        auto glcm_bytes = [&](uint32_t ng, uint32_t app) -> size_t {
            return (size_t)align16(4u * app * ng * ng) + 8ull * (25ull * ng + 128);
        };
        uint32_t ng;
        if (greyInfo != 0) {
            ng = (uint32_t)abs(greyInfo);
            if (fixed + 2ull * (L.lvl_cap + 8) + glcm_bytes(ng, 1) > cap) {
                why = "GLCM grey depth " + std::to_string(ng) + " too large for the LDS-resident co-occurrence matrix";
                return NYXHIP_ERR_UNSUPPORTED;
            }
        } else {
            // IBSI: matrix order = the ROI's largest intensity (glcm.cpp:400-419: the reference allocates max x max).  Known
            // (exact launch groups: the class header carries it): exactly that order -- and if it does not fit LDS next to the
            // ROIs the group goes to the global workspace like any grey depth beyond LDS.  Not known (stated extrema): the
            // largest order that fits next to this batch's ROIs, up to 128; a larger ROI raises the error flag.
            if (vmax != 0) {
                ng = vmax < 8 ? 8 : vmax;
                if (shared0 + glcm_bytes(ng, 1) > cap) {
                    why = "IBSI GLCM matrix order " + std::to_string(ng) + " too large for the LDS-resident co-occurrence matrix";
                    return NYXHIP_ERR_UNSUPPORTED;
                }
            } else {
                ng = 128;
                while (ng > 8 && shared0 + glcm_bytes(ng, 1) > cap) ng >>= 1;
            }
        }
        uint32_t app = 4;
        while (app > 1 && ((!spill && 4ull * app * ng * ng > 64 * 1024) || shared0 + glcm_bytes(ng, app) > cap)) app >>= 1;
        L.ng_cap = ng;
        L.app = app;
        uint32_t goff = shared0;
        L.P = goff; goff = align16(goff + 4u * app * (ng <= 16 ? (ng + 1) * (ng + 1) : ng * ng));   // split launches count with a skip row / column
        L.gscr = goff; goff = align16(goff + 8u * (25u * ng + 128));
        if (goff > off) off = goff;
    }
    if (L.dense8) {
        // 8-bit plane launches keep the plane at the START of the carve-out (the kernel then needs no base add per store):
        // [fixed | plane | ...] becomes [plane | fixed | ...], everything behind the two stays where it is
        const uint32_t dsz = align16(plane_bytes);            // the plane's bytes (dense8 implies GLCM; L.dense is 16-byte aligned)
        L.out += dsz; L.red += dsz; L.stat += dsz; L.lb100 += dsz; L.lbc += dsz;
        L.dense = 0;
        if (L.g16) L.gscr = 0;
    }
    if (spill) {
        // ---- workspace launches: the ROI-sized buffers (values, binned plane) live in global memory, but everything small and
        // atomics-heavy stays in LDS when it fits 60 KiB -- the fixed scratch always, then the co-occurrence matrices with their
        // feature scratch, the counting table, the level map.  (With all of it in the workspace a 96 k-pixel ROI spent two thirds of
        // its 4 ms in the global atomics of the co-occurrence sweep and the load pass.)  LDS-resident regions are exactly those at
        // offsets below L.gs_lds_bytes; nothing aliases.
        const uint32_t esz = L.cnt16 ? 2u : 4u;
        const uint64_t sz_val = align16(esz * L.sort_cap + 16), sz_cnt = do_int ? (uint64_t)esz * L.count_cap + 32 : 0;
        const uint64_t sz_dense = 2ull * L.dense_cap + 24, sz_lvl = do_glcm ? 2ull * (L.lvl_cap + 8) + 16 : 0;
        const uint64_t ngc = L.ng_cap, sz_P = do_glcm ? 4ull * L.app * (ngc <= 16 ? (ngc + 1) * (ngc + 1) : ngc * ngc) + 16 : 0;
        const uint64_t sz_g = do_glcm ? 8ull * (25ull * ngc + 128) + 16 : 0;
        const uint64_t kLdsMax = 60 * 1024;
        uint64_t o = fixed;
        const bool p_lds = do_glcm && o + sz_P + sz_g <= kLdsMax;
        if (p_lds) { L.P = (uint32_t)o; o = (o + sz_P + 15) & ~15ull; L.gscr = (uint32_t)o; o = (o + sz_g + 15) & ~15ull; }
        const bool c_lds = do_int && o + sz_cnt <= kLdsMax;
        if (c_lds) { L.cnt = (uint32_t)o; o = (o + sz_cnt + 15) & ~15ull; }
        const bool l_lds = do_glcm && o + sz_lvl <= kLdsMax;
        if (l_lds) { L.lvlmap = (uint32_t)o; o = (o + sz_lvl + 15) & ~15ull; }
        L.gs_lds_bytes = (uint32_t)o;
        if (do_glcm && !l_lds) { L.lvlmap = (uint32_t)o; o = (o + sz_lvl + 15) & ~15ull; }
        L.dense = (uint32_t)o; o = (o + sz_dense + 15) & ~15ull;
        L.val = (uint32_t)o; o = (o + sz_val + 15) & ~15ull;
        if (do_int && !c_lds) { L.cnt = (uint32_t)o; o = (o + sz_cnt + 15) & ~15ull; }
        if (do_glcm && !p_lds) { L.P = (uint32_t)o; o = (o + sz_P + 15) & ~15ull; L.gscr = (uint32_t)o; o = (o + sz_g + 15) & ~15ull; }
        if (o > cap) { why = "ROI too large for the global workspace (2 GiB of offsets per workgroup)"; return NYXHIP_ERR_ROI_TOO_LARGE; }
        off = (uint32_t)o;
    }
    L.total = off;
    if (L.total > cap) {
        why = "ROI too large for the LDS-resident path (" + std::to_string(L.total) + " B of LDS needed; max_px=" +
              std::to_string(max_px) + ", max_bbox_area=" + std::to_string(max_area) + ")";
        return NYXHIP_ERR_ROI_TOO_LARGE;
    }
    return NYXHIP_OK;
}

// One n x n complex Gabor kernel, interleaved re/im, L1-normalised by the sum of magnitudes:
// the formula and evaluation order of GaborFeature::Gabor (features/gabor.cpp:393-449), run
// on the host with libm exactly as the reference does.
void gabor_filter(double* Gex, double f0, double sig2lam, double gamma, double theta, double fi, int n)
{
    const double lambda = 2 * M_PI / f0, cos_theta = cos(theta), sin_theta = sin(theta), sig = sig2lam * lambda;
    std::vector<double> tx(n + 1), ty(n + 1);
    tx[0] = (n % 2 > 0) ? -((n - 1) / 2) : -(n / 2);
    for (int x = 1; x < n; x++) tx[x] = tx[x - 1] + 1;
    ty[0] = tx[0];
    for (int y = 1; y < n; y++) ty[y] = ty[y - 1] + 1;
    double sum = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            double xte = tx[x] * cos_theta + ty[y] * sin_theta;
            double yte = ty[y] * cos_theta - tx[x] * sin_theta;
            double rte = xte * xte + gamma * gamma * yte * yte;
            double ge = exp(-1 * rte / (2 * sig * sig));
            double argm = xte * f0 + fi;
            int idx = y * n * 2 + x * 2;
            Gex[idx] = ge * cos(argm);
            Gex[idx + 1] = ge * sin(argm);
            sum += sqrt(pow(Gex[idx], 2) + pow(Gex[idx + 1], 2));
        }
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n * 2; x++)
            Gex[y * n * 2 + x] /= sum;
}

// Low-pass baseline filter (f0LP at theta = pi/2, gabor.cpp:79) followed by the (f0, theta) pairs.
int ensure_gabor_bank(nyxhip_ctx* ctx, const nyxhip_settings* s)
{
    const int n = s->gabor_kersize, nF = s->gabor_n_filters;
    std::vector<double> key = {s->gabor_gamma, s->gabor_sig2lam, s->gabor_f0lp, (double)n, (double)nF};
    for (int i = 0; i < nF; i++) { key.push_back(s->gabor_f0[i]); key.push_back(s->gabor_theta[i]); }
    if (ctx->d_bank && key == ctx->bank_key)
        return NYXHIP_OK;
    std::vector<double> bank((size_t)(nF + 1) * n * n * 2);
    gabor_filter(bank.data(), s->gabor_f0lp, s->gabor_sig2lam, s->gabor_gamma, M_PI_2, 0, n);
    for (int f = 0; f < nF; f++)
        gabor_filter(bank.data() + (size_t)(f + 1) * n * n * 2, s->gabor_f0[f], s->gabor_sig2lam, s->gabor_gamma, s->gabor_theta[f], 0, n);
    for (int f = 0; f <= NYXHIP_MAX_GABOR_FILTERS; f++) ctx->bank_zero_rows[f] = 0;
    if (n == 16)
        for (int f = 0; f <= nF; f++)
            for (int j = 0; j < n; j++) {
                bool re0 = true, im0 = true;
                for (int i = 0; i < n; i++) {
                    const double* t = bank.data() + ((size_t)f * n * n + (size_t)j * n + i) * 2;
                    re0 = re0 && t[0] == 0.0;             // (+0 and -0 alike; a NaN or a denormal is not zero)
                    im0 = im0 && t[1] == 0.0;
                }
                ctx->bank_zero_rows[f] |= (re0 ? 1u << j : 0u) | (im0 ? 1u << (16 + j) : 0u);
            }
    ctx->bank_box_mask = 0;
    if (n == 16)
        for (int f = 0; f <= nF; f++) {
            const double* t = bank.data() + (size_t)f * n * n * 2;
            int e = 0;
            bool box = t[0] > 0.0 && std::frexp(t[0], &e) == 0.5 && t[0] >= 0x1p-64 && t[0] <= 1.0;   // a power of two (2^-8 in the default bank)
            for (int k = 0; k < n * n && box; k++)
                box = t[2 * k] == t[0] && t[2 * k + 1] == 0.0;
            if (box) ctx->bank_box_mask |= 1u << f;
        }
    // the low-pass filter as an outer product C_j B_i (ShapeArgs::gabor_lp_sep): pivot at the tap of largest magnitude
    ctx->bank_lp_sep = 0;
    if (n == 16) {
        const double* t = bank.data();
        int j0 = 0, i0 = 0;
        double best = -1.0, l1 = 0.0;
        for (int j = 0; j < 16; j++)
            for (int i = 0; i < 16; i++) {
                const double m = std::hypot(t[(j * 16 + i) * 2], t[(j * 16 + i) * 2 + 1]);
                l1 += m;
                if (m > best) { best = m; j0 = j; i0 = i; }
            }
        double B[16], resid = 0.0;
        bool ok = best > 0.0 && std::isfinite(l1);
        for (int i = 0; i < 16 && ok; i++) {
            // B_i = tap(j0, i) / tap(j0, i0), which must be real and non-negative
            const double ar = t[(j0 * 16 + i) * 2], ai = t[(j0 * 16 + i) * 2 + 1], pr = t[(j0 * 16 + i0) * 2], pi = t[(j0 * 16 + i0) * 2 + 1];
            B[i] = (ar * pr + ai * pi) / (pr * pr + pi * pi);
            ok = B[i] >= 0.0;
        }
        for (int j = 0; j < 16 && ok; j++)
            for (int i = 0; i < 16; i++) {
                const double cr = t[(j * 16 + i0) * 2], ci = t[(j * 16 + i0) * 2 + 1];
                resid += std::hypot(t[(j * 16 + i) * 2] - cr * B[i], t[(j * 16 + i) * 2 + 1] - ci * B[i]);
            }
        if (ok && resid <= 1e-12 * l1) {
            ctx->bank_lp_sep = 1;
            memset(ctx->bank_lp_C, 0, sizeof(ctx->bank_lp_C));
            for (int i = 0; i < 16; i++) ctx->bank_lp_B[i] = (float)B[i];
            for (int j = 0; j < 16; j++) { ctx->bank_lp_C[2 * (j + 3)] = (float)t[(j * 16 + i0) * 2]; ctx->bank_lp_C[2 * (j + 3) + 1] = (float)t[(j * 16 + i0) * 2 + 1]; }
        }
        if (getenv("NYXHIP_DEBUG")) fprintf(stderr, "[nyxhip] gabor low-pass: separable %u (residual %.3g of %.3g)\n", ctx->bank_lp_sep, resid, l1);
    }
    if (ctx->d_bank) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream())); HIP_TRY(ctx, hipFree(ctx->d_bank)); ctx->d_bank = nullptr; }
    if (ctx->d_bank32) { HIP_TRY(ctx, hipFree(ctx->d_bank32)); ctx->d_bank32 = nullptr; }
    if (ctx->d_bank16) { HIP_TRY(ctx, hipFree(ctx->d_bank16)); ctx->d_bank16 = nullptr; }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_bank, bank.size() * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_bank, bank.data(), bank.size() * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<float> b32(bank.size());
        for (size_t i = 0; i < bank.size(); i++) b32[i] = (float)bank[i];      // round to nearest: relative 2^-24 (the bound of the screening pass counts it)
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_bank32, b32.size() * sizeof(float)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_bank32, b32.data(), b32.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (n == 16 && nF > 0) {
        // MFMA screening stage of roi_gabor_tiled_kernel (MODE 4): the band-pass filters in groups of four as B operands of
        // v_mfma_f32_16x16x32_f16.  Operand (group g, tap-row pair jp), lane l = (column nn = l % 16, k block kb = l / 16), element t:
        // tap (j', i') = (2 jp + kb / 2, 8 (kb % 2) + t) of the FLIPPED kernel -- the convolution as a correlation over the padded
        // plane: out(a, b) = sum P[b + j'][a + 1 + i'] G[15 - j'][15 - i'] -- of filter 1 + 4 g + (nn % 8) / 2, component nn % 2,
        // scaled by 2^14; columns 0 .. 7 carry the f16 nearest to the scaled tap, columns 8 .. 15 the f16 nearest to the rest.
        const int groups = (nF + 3) / 4;
        std::vector<_Float16> ops((size_t)groups * 8 * 64 * 8);
        for (int g = 0; g < groups; g++)
            for (int jp = 0; jp < 8; jp++)
                for (int l = 0; l < 64; l++)
                    for (int t = 0; t < 8; t++) {
                        const int nn = l & 15, kb = l >> 4, f = 1 + 4 * g + ((nn & 7) >> 1), c = nn & 1, jq = 2 * jp + (kb >> 1), iq = 8 * (kb & 1) + t;
                        _Float16 v = (_Float16)0.0f;
                        if (f <= nF) {
                            const double tap = bank[(((size_t)f * 16 + (15 - jq)) * 16 + (15 - iq)) * 2 + c] * kGaborTapScale;
                            const _Float16 hi = (_Float16)tap;
                            v = nn < 8 ? hi : (_Float16)(tap - (double)hi);
                        }
                        ops[(((size_t)g * 8 + jp) * 64 + l) * 8 + t] = v;
                    }
        HIP_TRY(ctx, hipMalloc(&ctx->d_bank16, ops.size() * sizeof(_Float16)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_bank16, ops.data(), ops.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    }
    ctx->bank_key = key;
    return NYXHIP_OK;
}

int make_shape_layout(uint32_t mask, const nyxhip_settings* s, uint32_t max_area, uint32_t max_side, ShapeLayout& L, std::string& why, size_t cap = 0)
{
    memset(&L, 0, sizeof(L));
    const bool spill = cap != 0;
    if (cap == 0) cap = roi_features_max_lds();
    if (!(mask & NYXHIP_FAM_GABOR))
        return NYXHIP_OK;
    uint32_t off = 0;
    if (!spill && s->gabor_kersize == 16) {
        // register-tiled kernel: one zero-padded u32 plane, (roundup(w, 8) + 16) x (h + 15) words <= area + 38 side + 345
        L.tiled = 1;
        L.red = off; off = align16(off + 8u * kWaves * NYXHIP_MAX_GABOR_FILTERS);
        L.area_cap = max_area ? max_area : 1;
        L.side_cap = max_side ? max_side : 1;
        const uint64_t words = (uint64_t)L.area_cap + 42ull * L.side_cap + 405;   // (w + 27) (h + 15): tiles + padding, pitch made an odd number of 16-byte units
        if (4ull * words + off > cap) { why = "ROI bounding box too large for the LDS-resident Gabor plane"; return NYXHIP_ERR_ROI_TOO_LARGE; }
        L.plane = off; off = align16(off + 4u * (uint32_t)words);
        L.redo = off; off = align16(off + 4u * (512u + 4u));     // kGaborRedoCap of roi_shape.hip
        L.total = off;
        return NYXHIP_OK;
    }
    L.red = off; off = align16(off + 8u * kWaves * 8);
    L.area_cap = max_area ? max_area : 1;
    if (16ull * L.area_cap > cap) { why = "ROI bounding box too large for the LDS-resident Gabor planes"; return NYXHIP_ERR_ROI_TOO_LARGE; }
    L.plane = off; off = align16(off + 8u * L.area_cap);
    L.energy = off; off = align16(off + 8u * L.area_cap);
    L.bank = off; off = align16(off + 16u * (uint32_t)(s->gabor_n_filters + 1) * s->gabor_kersize * s->gabor_kersize);
    L.total = off;
    if (L.total > cap) { why = "ROI bounding box too large for the LDS-resident Gabor planes"; return NYXHIP_ERR_ROI_TOO_LARGE; }
    return NYXHIP_OK;
}

// LDS carve-out of the texture kernel (roi_texture.hip)
int make_tex_layout(uint32_t mask, const nyxhip_settings* s, int n_cols, uint32_t max_area, uint32_t max_side,
                    TexLayout& L, std::string& why, size_t cap = 0, uint32_t vmax = 0)
{
    memset(&L, 0, sizeof(L));
    const bool spill = cap != 0;
    if (!spill) cap = roi_features_max_lds();
    const int greyInfo = s->ibsi ? 0 : s->grey_depth;
    uint32_t off = 0;
    L.out = off; off = align16(off + 8u * (uint32_t)n_cols);
    L.red = off; off = align16(off + 8u * kWaves * 8);
    L.stat = off; off = align16(off + 8u * 16);
    L.dense_cap = max_area ? max_area : 1;
    L.side_cap = max_side ? max_side : 1;
    if (2ull * L.dense_cap > cap) { why = "ROI bounding box of " + std::to_string(max_area) + " px exceeds the LDS-resident plane"; return NYXHIP_ERR_ROI_TOO_LARGE; }
    // IBSI: levels are the intensities themselves -- up to the group's largest intensity when the class header gave it (exact launch
    // groups), else the 8-bit range
    L.lvl_cap = greyInfo != 0 ? (uint32_t)abs(greyInfo) : (vmax ? vmax : 255u);
    if (L.lvl_cap > 4094) { why = "grey depth (or IBSI intensity) above 4094 is not supported by the texture kernel"; return NYXHIP_ERR_UNSUPPORTED; }
    L.dense8 = (!spill && L.lvl_cap <= 254) ? 1u : 0u;            // 8-bit plane (roi_texture_kernel<.., true>)
    L.dense = off; off = align16(off + (L.dense8 ? 1u : 2u) * L.dense_cap + 4);
    L.ng_cap = L.lvl_cap + 1;
    L.lvlmap = off; off = align16(off + 2u * (L.lvl_cap + 4));
    L.lv = off; off = align16(off + 4u * (L.ng_cap + 4));
    if (L.ng_cap <= 256 && (mask & (NYXHIP_FAM_GLRLM | NYXHIP_FAM_GLSZM))) { L.lvf = off; off = align16(off + 16u * (L.ng_cap + 2)); }
    // NGTDM accumulators (u64 S[ng_cap + 2], u32 N[ng_cap + 2]).  Few levels mean few addresses under 64-lane atomics, which LDS
    // serialises: R replicas (lane % R picks one) keep the lanes per address near one; an odd multiple of 8 bytes apart so that the
    // replicas start in different banks.
    L.ngt_rep = 1; L.ngt_stride = ((L.ng_cap + 2) * 12u + 7u) & ~7u;
    if ((mask & NYXHIP_FAM_NGTDM) && !spill && L.ng_cap <= 64) {
        L.ngt_rep = L.ng_cap <= 16 ? 8u : L.ng_cap <= 32 ? 4u : 2u;
        L.ngt_stride = (((L.ng_cap + 2) * 12u + 16u + 7u) & ~7u) | 8u;
    }
    if ((mask & NYXHIP_FAM_NGTDM) && (mask & NYXHIP_FAM_GLSZM) && !spill && L.ng_cap <= 64) {
        // accumulators of its own: the stencil overlaps the GLSZM sweep
        L.ngt_own = off; off = align16(off + L.ngt_rep * L.ngt_stride);
    }
    L.work = off;
    size_t need = 0;
    if (mask & NYXHIP_FAM_NGTDM) {
        L.ngt_p = L.ngt_own ? 0u : (L.ngt_rep * L.ngt_stride + 15u) & ~15u;      // P[ng_cap + 2], S / 840 [ng_cap + 2] as doubles, behind the aliased accumulators
        need = std::max(need, (size_t)L.ngt_p + (size_t)(L.ng_cap + 2) * 16 + 64);
    }
    if (mask & NYXHIP_FAM_GLSZM) {
        // distinct (level, size) pairs <= sqrt(2 * Ng * area) (sizes of one level sum to <= its area)
        // (load <= 2/3 in the worst case; zones of up to 32 pixels bypass the hash altogether when the direct table exists.  With
        //  2 * distinct the benchmark's carve-out was 4 KiB larger: six instead of seven workgroups per CU.)  The kernel uses
        // szm_hash_cap of each ROI's OWN box, which this carve-out -- the same monotone function of the largest box -- covers.
        L.hash_cap = szm_hash_cap(L.ng_cap, L.dense_cap);
        // zone sizes (16-bit entries, two per word, while a size fits), hash, zones per level; the owner-label plane only
        // exists for boxes wider than one wave (the DPP sweep of narrower boxes keeps labels in registers)
        L.szm_c16 = (!spill && L.dense_cap < 65535u) ? 1 : 0;
        size_t szm = 0;
        L.szm_count = (uint32_t)szm; szm += ((L.szm_c16 ? 2ull : 4ull) * (L.dense_cap + 8) + 15) & ~15ull;
        L.szm_hkey = (uint32_t)szm; szm += 8ull * L.hash_cap + 4ull * (L.ng_cap + 4);
        szm = (szm + 15) & ~15ull;
        // direct [level][size] counters for the small zones (nearly all of them on textured images): one atomic, no probing
        L.szm_small = L.ng_cap <= 33 ? 32u : 0u;
        L.szm_smalltab = (uint32_t)szm; szm += 4ull * L.ng_cap * L.szm_small;
        szm = (szm + 15) & ~15ull;
        L.szm_label = (uint32_t)szm; if (L.side_cap > (spill ? 512u : 256u)) szm += 4ull * L.dense_cap;   // (boxes up to 256 wide keep their labels in registers: kSzmChunks of roi_texture.hip)
        L.szm_ok = (off + szm <= cap) ? 1 : 0;
        if (!L.szm_ok) { why = "ROI too large for the LDS-resident GLSZM zone tables"; return NYXHIP_ERR_ROI_TOO_LARGE; }
        need = std::max(need, szm);
    }
    if (mask & NYXHIP_FAM_GLRLM) {
        size_t slot = 4ull * ((size_t)L.ng_cap * L.side_cap + L.ng_cap + L.side_cap + 4);
        if (off + 512 + slot > cap) { why = "ROI too large for the LDS-resident run-length matrix"; return NYXHIP_ERR_ROI_TOO_LARGE; }
        size_t k = 4;
        while (k > 1 && off + 512 + k * slot > cap) k--;
        // keep the carve-out modest when four matrices would crowd out co-resident workgroups
        while (!spill && k > 1 && 512 + k * slot > 48 * 1024) k--;
        need = std::max(need, 512 + k * slot);
    }
    L.work_bytes = (uint32_t)need;
    off = align16(off + (uint32_t)need);
    L.total = off;
    if (L.total > cap) { why = "ROI too large for the LDS-resident texture path"; return NYXHIP_ERR_ROI_TOO_LARGE; }
    if (spill) {
        // what a workspace launch keeps in LDS all the same: the atomics-heavy small state (64 lanes adding into a handful of
        // addresses are 64 serialised L2 atomics in global memory)
        uint32_t o = 0;
        if ((mask & NYXHIP_FAM_NGTDM) && L.ng_cap <= 1024) {
            L.ngt_rep = L.ng_cap <= 16 ? 8u : L.ng_cap <= 32 ? 4u : L.ng_cap <= 64 ? 2u : 1u;
            L.ngt_stride = (((L.ng_cap + 2) * 12u + 16u + 7u) & ~7u) | 8u;
            L.gs_ngt = o; L.gs_ngt_ok = 1; o = align16(o + L.ngt_rep * L.ngt_stride);
        }
        if ((mask & NYXHIP_FAM_GLRLM) && 16ull * L.ng_cap * kRlmLdsCols <= 32768) {
            L.gs_rlm = o; L.gs_rlm_ok = 1; o = align16(o + 16u * L.ng_cap * kRlmLdsCols);
        }
        L.gs_lds_bytes = o;
    }
    return NYXHIP_OK;
}

// Carve-out of roi_dependence_kernel (GLDZM + GLDM + NGLDM).
int make_dep_layout(uint32_t mask, const nyxhip_settings* s, uint32_t max_area, uint32_t max_side, DepLayout& L, std::string& why, size_t cap = 0,
                    uint32_t vmax = 0)
{
    memset(&L, 0, sizeof(L));
    if (cap == 0) cap = roi_features_max_lds();
    const int greyInfo = s->ibsi ? 0 : s->grey_depth;
    uint32_t off = 0;
    L.red = off; off = align16(off + 8u * kWaves * 8);
    L.stat = off; off = align16(off + 8u * 16);
    L.dense_cap = max_area ? max_area : 1;
    L.side_cap = max_side ? max_side : 1;
    if (4ull * L.dense_cap > cap) { why = "ROI bounding box of " + std::to_string(max_area) + " px exceeds the LDS-resident planes"; return NYXHIP_ERR_ROI_TOO_LARGE; }
    L.lvl_cap = greyInfo != 0 ? (uint32_t)abs(greyInfo) : (vmax ? vmax : 255u);   // IBSI: levels are the intensities themselves
    if (L.lvl_cap > 4094) { why = "grey depth (or IBSI intensity) above 4094 is not supported by the dependence kernel"; return NYXHIP_ERR_UNSUPPORTED; }
    L.planes8 = (cap == roi_features_max_lds() && L.lvl_cap <= 63) ? 1u : 0u;   // byte planes: level + two flags fit 8 bits
    L.dense = off; off = align16(off + (L.planes8 ? 1u : 2u) * L.dense_cap + 4);
    L.aux = off; off = align16(off + (L.planes8 ? 1u : 2u) * L.dense_cap + 4);
    L.ng_cap = L.lvl_cap + 1;
    L.lvlmap = off; off = align16(off + 2u * (L.lvl_cap + 4));
    L.lv = off; off = align16(off + 4u * (L.ng_cap + 4));
    L.lvlmap2 = off; off = align16(off + 2u * (L.lvl_cap + 4));
    L.lv2 = off; off = align16(off + 4u * (L.ng_cap + 4));
    L.work = off;
    L.nd_cap = L.side_cap / 2 + 2;
    size_t need = (size_t)4 * 9 * (L.ng_cap + 1);                       // GLDM / NGLDM matrices
    if (mask & NYXHIP_FAM_GLDZM)                                     // union-find parents + matrix
        need = std::max<size_t>(need, 4ull * L.dense_cap + 4ull * (size_t)L.ng_cap * L.nd_cap + 64);
    if (cap == roi_features_max_lds() && L.ng_cap <= 65 && !getenv("NYXHIP_DEP_SEQ")) {   // (NYXHIP_DEP_SEQ: A/B knob, sequential tails)
        // small matrices: GLDM's and NGLDM's sit side by side at the start of `work` -- where GLDZM keeps its union-find parents,
        // which are dead once its own matrix (behind them) is built -- so the three tails can run in parallel at the end without
        // a byte of extra LDS (the carve-out of the benchmark ROI sits 400 B below the five-workgroups-per-CU line)
        const size_t mat = ((size_t)4 * 9 * (L.ng_cap + 1) + 15) & ~(size_t)15;
        size_t base = 0;
        if ((mask & NYXHIP_FAM_GLDZM) && 2 * mat > 4ull * L.dense_cap)       // small boxes, many levels: the pair would reach GLDZM's own
            base = (4ull * L.dense_cap + 4ull * (size_t)L.ng_cap * L.nd_cap + 64 + 15) & ~15ull;   // matrix -- it goes behind it instead
        L.par = 1; L.off_pdm = (uint32_t)base; L.off_m = (uint32_t)(base + mat);
        need = std::max<size_t>(need, base + 2 * mat);
    }
    if (off + need > cap) { why = "ROI too large for the LDS-resident dependence / distance-zone tables"; return NYXHIP_ERR_ROI_TOO_LARGE; }
    L.work_bytes = (uint32_t)need;
    off = align16(off + (uint32_t)need);
    L.total = off;
    return NYXHIP_OK;
}

int ensure_stage(nyxhip_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->stage_bytes)
        return NYXHIP_OK;
    if (ctx->d_stage)
        HIP_TRY(ctx, hipFree(ctx->d_stage));
    ctx->d_stage = nullptr;
    ctx->stage_bytes = 0;
    size_t want = bytes + bytes / 4 + (1 << 20);
    HIP_TRY(ctx, hipMalloc(&ctx->d_stage, want));
    ctx->stage_bytes = want;
    return NYXHIP_OK;
}

int check_status(nyxhip_ctx* ctx)
{
    int two[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpy(two, ctx->d_status, 2 * sizeof(int), hipMemcpyDeviceToHost));
    const int st = two[0];
    if (ctx->census_pending) {         // whole-batch launches on stated extrema ran since the last look: what their scan met
        ctx->census_small = (uint32_t)two[1]; ctx->census_total = ctx->census_pending; ctx->census_pending = 0;
    }
    if (two[1]) HIP_TRY(ctx, hipMemsetAsync(ctx->d_status + 1, 0, sizeof(int), ctx->stream()));
    if (st != 0) {
        int zero = 0;
        HIP_TRY(ctx, hipMemcpy(ctx->d_status, &zero, sizeof(int), hipMemcpyHostToDevice));
        if (st == NYXHIP_ERR_ROI_TOO_LARGE)
            return fail(ctx, st, "an ROI exceeds the LDS-resident capacity declared by the batch extrema");
        if (st == NYXHIP_ERR_UNSUPPORTED)
            return fail(ctx, st, "GLCM matrix order exceeds the LDS-resident capacity (IBSI mode with large intensities?)");
        return fail(ctx, st, "device-side error " + std::to_string(st));
    }
    return NYXHIP_OK;
}

__global__ void add_offset_kernel(const uint32_t* in, uint32_t add, uint32_t n, uint32_t* out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}


// Fills the three argument blocks for one set of extrema; `cap` = 0 -> LDS carve-outs, else spill layouts.
int build_args(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s, double* d_out, size_t ld,
               const Extrema& E, size_t cap, RoiArgs& a, TexArgs& t, ShapeArgs& g, DepArgs& d, std::string& why, uint32_t groups = 0xF)
{   // groups: bit 0 features (INTENSITY + GLCM), 1 texture, 2 shape, 3 dependence -- the kernel groups to build (columns always follow `mask`)
    const uint32_t mask1 = mask & (NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM), mask2 = mask & kTexture, mask3 = mask & kShape, mask4 = mask & kDependence;
    const int n_cols1 = nyxhip_n_columns(mask1, s), n_cols2 = nyxhip_n_columns(mask2, s), n_cols4 = nyxhip_n_columns(mask4, s);
    memset(&a, 0, sizeof(a));
    memset(&t, 0, sizeof(t));
    memset(&g, 0, sizeof(g));
    memset(&d, 0, sizeof(d));
    // Feature2D order inside the row: INTENSITY, GLCM, GLRLM, GLDZM, GLSZM, GLDM, NGLDM, NGTDM, GABOR, ZERNIKE
    int c_glrlm = n_cols1, c_gldzm = c_glrlm + ((mask & NYXHIP_FAM_GLRLM) ? kGlrlmCols : 0);
    int c_glszm = c_gldzm + ((mask & NYXHIP_FAM_GLDZM) ? kGldzmCols : 0), c_gldm = c_glszm + ((mask & NYXHIP_FAM_GLSZM) ? kGlszmCols : 0);
    int c_ngldm = c_gldm + ((mask & NYXHIP_FAM_GLDM) ? kGldmCols : 0);
    if (mask1 && (groups & 1)) {
        if (int lrc = make_layout(mask1, s, n_cols1, E.px, E.area, E.range, a.L, why, cap, E.vmax, E.wide_only))
            return lrc;
        a.n_roi = b->n_roi;
        a.px_offset = b->px_offset; a.x = b->x; a.y = b->y; a.inten = b->inten;
        a.bbox_w = b->bbox_w; a.bbox_h = b->bbox_h; a.min_inten = b->min_inten; a.max_inten = b->max_inten;
        a.slide_min = b->slide_min; a.slide_max = b->slide_max;
        a.out = d_out; a.ld = ld; a.status = ctx->d_status;
        a.stamps = ctx->d_stamps;
        if (cap == 0) a.win = ctx->win_next;   // LDS launches only (the tile path asks for windows only when everything fits LDS)
        a.mask = mask1; a.n_cols = n_cols1;
        int c = 0;
        a.col_intensity = a.col_glcm = -1;
        if (mask1 & NYXHIP_FAM_INTENSITY) { a.col_intensity = c; c += kIntensityCols; }
        if (mask1 & NYXHIP_FAM_GLCM) { a.col_glcm = c; c += kGlcmAngled * s->glcm_n_angles + kGlcmAve; }
        a.soft_nan = s->soft_nan;
        a.grey_depth = s->grey_depth; a.ibsi = s->ibsi; a.glcm_grey_depth = s->glcm_grey_depth;
        a.glcm_offset = s->glcm_offset; a.glcm_na = s->glcm_n_angles; a.glcm_symmetric = s->glcm_symmetric;
        for (int i = 0; i < kMaxAngles; i++) a.glcm_angles[i] = s->glcm_angles[i];
        a.n_hist = abs(s->grey_depth);
        // small matrices (order <= 16, all angles in one pass, matlab or IBSI level values 1..Ng): the features run as their
        // own launch with one wave per ROI (glcm_features_kernel); the counts travel through a context-owned workspace
        const int gi = s->ibsi ? 0 : s->grey_depth;
        if ((mask1 & NYXHIP_FAM_GLCM) && cap == 0 && gi >= 0 && a.L.ng_cap <= 16 && (int)a.L.app >= s->glcm_n_angles && s->glcm_n_angles > 0) {
            const size_t stride = (size_t)s->glcm_n_angles * a.L.ng_cap * a.L.ng_cap;
            const size_t need = 4 * stride * (size_t)b->n_roi + 256;
            if (need > ctx->glcm_ws_bytes) {
                if (ctx->d_glcm_ws) { (void)hipFree(ctx->d_glcm_ws); ctx->d_glcm_ws = nullptr; ctx->glcm_ws_bytes = 0; }
                if (hipMalloc((void**)&ctx->d_glcm_ws, need) == hipSuccess) ctx->glcm_ws_bytes = need;
            }
            if (!(ctx->d_glcm_ws && ctx->glcm_ws_bytes >= need) || !ctx->d_glcm_ng) {   // (d_glcm_ng: sized and cleared once per call by launch_device_all)
                why = "out of device memory for the GLCM count workspace";
                return NYXHIP_ERR_HIP;
            }
            {
                a.glcm_ng = ctx->d_glcm_ng;
                a.glcm_ws = ctx->d_glcm_ws;
                a.glcm_ws_stride = (uint32_t)stride;
            }
        }
    }
    if (mask2 && (groups & 2)) {
        if (int lrc = make_tex_layout(mask2, s, n_cols2, E.area, E.side, t.L, why, cap, E.vmax))
            return lrc;
        t.n_roi = b->n_roi;
        t.px_offset = b->px_offset; t.x = b->x; t.y = b->y; t.inten = b->inten;
        t.bbox_w = b->bbox_w; t.bbox_h = b->bbox_h; t.min_inten = b->min_inten; t.max_inten = b->max_inten;
        t.out = d_out; t.ld = ld; t.status = ctx->d_status;
        t.mask = mask2; t.n_cols = n_cols2; t.col0 = n_cols1;
        t.gap_after_glrlm = (mask & NYXHIP_FAM_GLDZM) ? kGldzmCols : 0;
        t.gap_after_glszm = ((mask & NYXHIP_FAM_GLDM) ? kGldmCols : 0) + ((mask & NYXHIP_FAM_NGLDM) ? kNgldmCols : 0);
        t.soft_nan = s->soft_nan; t.grey_depth = s->grey_depth; t.ibsi = s->ibsi;
    }
    if (mask4 && (groups & 8)) {
        if (int lrc = make_dep_layout(mask4, s, E.area, E.side, d.L, why, cap, E.vmax))
            return lrc;
        d.n_roi = b->n_roi;
        d.px_offset = b->px_offset; d.x = b->x; d.y = b->y; d.inten = b->inten;
        d.bbox_w = b->bbox_w; d.bbox_h = b->bbox_h; d.min_inten = b->min_inten; d.max_inten = b->max_inten;
        d.out = d_out; d.ld = ld; d.status = ctx->d_status;
        d.mask = mask4;
        d.col_gldzm = c_gldzm; d.col_gldm = c_gldm; d.col_ngldm = c_ngldm;
        d.soft_nan = s->soft_nan; d.grey_depth = s->grey_depth; d.ibsi = s->ibsi;
    }
    if (mask3 && (groups & 4)) {
        if (int lrc = make_shape_layout(mask3, s, E.area, E.side, g.L, why, cap))
            return lrc;
        g.L.zern_px_cap = std::min<uint32_t>((E.px + 3u) & ~3u, 4096);   // <= 32 KiB of dynamic LDS in the Zernike kernel
        g.n_roi = b->n_roi;
        g.px_offset = b->px_offset; g.x = b->x; g.y = b->y; g.inten = b->inten;
        g.bbox_w = b->bbox_w; g.bbox_h = b->bbox_h; g.min_inten = b->min_inten; g.max_inten = b->max_inten;
        g.out = d_out; g.ld = ld; g.status = ctx->d_status;
        g.mask = mask3;
        g.col_gabor = n_cols1 + n_cols2 + n_cols4;
        g.col_zernike = g.col_gabor + ((mask3 & NYXHIP_FAM_GABOR) ? s->gabor_n_filters : 0);
        g.soft_nan = s->soft_nan;
        g.small_rois = (E.px <= kClassPx[0] && E.side <= kClassSide[0]) ? 1 : 0;   // the smallest size class (a function of the ROI: roi_class)
        g.gabor_bank = ctx->d_bank; g.gabor_bank32 = ctx->d_bank32; g.gabor_bank16 = ctx->d_bank16; { static const int dbg_phase_env = [] { const char* e = getenv("NYXHIP_DBG_PHASE"); return e ? atoi(e) : 0; }(); g.dbg_phase = dbg_phase_env; } g.gabor_nf = s->gabor_n_filters; g.gabor_n = s->gabor_kersize; g.gabor_thr = s->gabor_graythr;
        for (int f = 0; f <= NYXHIP_MAX_GABOR_FILTERS; f++) g.gabor_zero_rows[f] = ctx->bank_zero_rows[f];
        g.gabor_box_mask = ctx->bank_box_mask;
        { static const bool no_lpsep = getenv("NYXHIP_GABOR_NO_LPSEP") != nullptr; g.gabor_lp_sep = ctx->bank_lp_sep && !no_lpsep; }
        memcpy(g.gabor_lp_B, ctx->bank_lp_B, sizeof(g.gabor_lp_B)); memcpy(g.gabor_lp_C, ctx->bank_lp_C, sizeof(g.gabor_lp_C));
    }
    return NYXHIP_OK;
}

// ROIs whose padded flag plane exceeds the LDS cap of the contour kernel -> index list
__global__ void classify_plane_kernel(uint64_t n_roi, const uint32_t* bw, const uint32_t* bh, uint32_t cap, uint32_t* list, uint32_t* n_out)
{
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n_roi) return;
    if ((uint64_t)(bw[i] + 2) * (bh[i] + 2) > cap)
        list[atomicAdd(n_out, 1u)] = (uint32_t)i;
}

// A workspace lane: a high-priority stream forked from the call's stream (nyxhip_ctx::lane_fork, recorded at the start of launch_device_all),
// joined into it at the end of the call (LaneJoin).  The lanes carry chains of short, latency-bound kernels -- a few hundred workgroups each --
// beside the main stream's chip-filling grids: at the device's highest priority their workgroups are placed first and the chain is not starved.
int use_lane(nyxhip_ctx* ctx, int lane, hipStream_t* st)
{
    if (!ctx->lane_stream[lane]) {
        static const bool no_prio = [] { const char* e = getenv("NYXHIP_NO_LANE_PRIORITY"); return e && *e && *e != '0'; }();   // A/B knob
        int lo = 0, hi = 0;
        if (no_prio || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->lane_stream[lane], hipStreamNonBlocking, hi));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->lane_done[lane], hipEventDisableTiming));
    }
    if (!ctx->lane_used[lane]) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->lane_stream[lane], ctx->lane_fork, 0));   // the batch and the class lists are complete on the main stream
        ctx->lane_used[lane] = true;
    }
    *st = ctx->lane_stream[lane];
    return NYXHIP_OK;
}

// Contour + 2-D geometric moments (roi_moments.hip).  The contour of every ROI goes to a context-owned workspace at the
// ROI's CSR offset (a contour never has more points than the ROI has pixels); the moments kernel reads it back.
int launch_moments(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s, double* d_out, size_t ld,
                   uint32_t max_px, uint32_t max_area, uint32_t max_side)
{
    hipStream_t st = ctx->stream();
    // A batch with boxes beyond the LDS plane sends those to a wave per ROI over a global workspace (a few hundred waves, ~10 ms of
    // latency for the heavy-tailed batch): with other families in the call the whole moments chain goes to a lane of its own and
    // runs beside them (enqueued last, dependent only on the batch).  Its scratch is the lane's, not the main stream's.
    static const bool no_mom_lane = [] { const char* e = getenv("NYXHIP_NO_MOM_LANE"); return e && *e && *e != '0'; }();   // A/B knob
    const bool big_boxes = (uint64_t)kContourWaves * (((uint64_t)max_area + 4ull * max_side + 4 + 15) & ~15ull) > (uint64_t)roi_features_max_lds();
    const bool on_lane = !no_mom_lane && big_boxes && (mask & ~kMoments) && ctx->lane_fork;
    if (on_lane)
        if (int lrc = use_lane(ctx, nyxhip_ctx::kMomLane, &st)) return lrc;
    unsigned char** const spillp = on_lane ? &ctx->lane_buf[nyxhip_ctx::kMomLane] : &ctx->d_spill;
    size_t* const spill_bytesp = on_lane ? &ctx->lane_bytes[nyxhip_ctx::kMomLane] : &ctx->spill_bytes;
    uint64_t total_px = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total_px, b->px_offset + b->n_roi, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_k = 0, o_n = al(4 * (size_t)total_px + 256), o_l = al(o_n + 4 * (size_t)b->n_roi + 256), need = al(o_l + 8 * (size_t)total_px + 256);
    if (need > ctx->mom_bytes) {
        if (ctx->d_mom) { HIP_TRY(ctx, hipFree(ctx->d_mom)); ctx->d_mom = nullptr; ctx->mom_bytes = 0; }
        HIP_TRY(ctx, hipMalloc(&ctx->d_mom, need + need / 8));
        ctx->mom_bytes = need + need / 8;
    }
    char* base = (char*)ctx->d_mom;
    MomArgs m;
    memset(&m, 0, sizeof(m));
    m.n_roi = b->n_roi;
    m.px_offset = b->px_offset; m.x = b->x; m.y = b->y; m.inten = b->inten; m.bbox_w = b->bbox_w; m.bbox_h = b->bbox_h;
    m.out = d_out; m.ld = ld; m.status = ctx->d_status;
    m.mask = mask & kMoments;
    m.col_smoms = nyxhip_n_columns(mask & ~kMoments, s);
    m.col_imoms = m.col_smoms + ((mask & NYXHIP_FAM_SMOMS) ? kMomCols : 0);
    m.ws_contour = (uint32_t*)(base + o_k); m.n_contour = (uint32_t*)(base + o_n); m.ws_L = (double*)(base + o_l);
    if (!ctx->d_logtab) {                                 // log(sqrt(d) + 0.001), d < 32768: boxes up to 128 x 128 never evaluate a logarithm
        constexpr uint32_t kLogTab = 32768;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_logtab, 8ull * kLogTab));
        if (launch_moments_logtab(ctx->d_logtab, kLogTab, st) != 0) return fail(ctx, NYXHIP_ERR_HIP, "moments log table: launch failed");
        ctx->logtab_n = kLogTab;
    }
    m.log_tab = ctx->d_logtab; m.log_tab_n = ctx->logtab_n;
    // LDS of the moments kernel from the batch extrema: every pixel of the largest ROI (up to kMomPxLds; larger ROIs sweep HBM),
    // a contour of up to the bounding box's perimeter (what a convex ROI can have; longer ones are read from HBM), its step table
    m.px_cap = std::min<uint32_t>((uint32_t)kMomPxLds, (std::max<uint32_t>(max_px ? max_px : max_area, 1u) + 7u) & ~7u);
    m.k_cap = std::min<uint32_t>((uint32_t)kMomContourLds, std::max<uint32_t>(256u, (4u * std::min<uint32_t>(max_side, 65536u) + 63u) & ~63u));
    m.step_cap = std::min<uint32_t>((uint32_t)kMomStepTab, m.k_cap);
    const uint64_t full_plane = (uint64_t)max_area + 4ull * max_side + 4;      // (w + 2)(h + 2) <= area + 2(w + h) + 4
    const uint32_t grid = (uint32_t)b->n_roi;
    const uint32_t lds_cap = (uint32_t)roi_features_max_lds();
    int rc;
    if ((uint64_t)kContourWaves * ((full_plane + 15) & ~15ull) <= lds_cap) {   // kContourWaves planes per workgroup
        m.plane_cap = (uint32_t)full_plane;
        rc = launch_roi_contour(m, st, grid);
        if (rc == 0) rc = launch_roi_moments(m, st, grid);
    } else {
        // the bulk of the batch from LDS (16 KiB planes keep ten waves per CU), the oversized ROIs from a global workspace: a wave per ROI,
        // a few hundred waves and ~10 ms of latency for the heavy-tailed batch.  Two independent chains -- big boxes: list, contour over the
        // workspace, moments of the list | bulk: contour from LDS (skipping the big boxes), moments of everybody else -- on two lanes when
        // the call has lanes (other families to run beside), one after the other on the call's stream otherwise.
        m.plane_cap = 16 * 1024;
        hipStream_t st_big = st;
        if (on_lane)
            if (int lrc = use_lane(ctx, nyxhip_ctx::kMomLaneBig, &st_big)) return lrc;
        const size_t list_bytes = 4ull * b->n_roi + 256;
        if (list_bytes > ctx->spill_list_bytes) {
            if (ctx->d_spill_list) { HIP_TRY(ctx, hipStreamSynchronize(st_big)); HIP_TRY(ctx, hipFree(ctx->d_spill_list)); ctx->d_spill_list = nullptr; }
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_spill_list, list_bytes));
            ctx->spill_list_bytes = list_bytes;
        }
        uint32_t* d_cnt = ctx->d_spill_list;
        uint32_t* d_list = ctx->d_spill_list + 64;
        HIP_TRY(ctx, hipMemsetAsync(d_cnt, 0, 4, st_big));
        hipLaunchKernelGGL(classify_plane_kernel, dim3((unsigned)((b->n_roi + 255) / 256)), dim3(256), 0, st_big, b->n_roi, b->bbox_w, b->bbox_h,
                           m.plane_cap, d_list, d_cnt);
        uint32_t n_large = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&n_large, d_cnt, 4, hipMemcpyDeviceToHost, st_big));
        HIP_TRY(ctx, hipStreamSynchronize(st_big));
        rc = 0;
        if (n_large) {
            if (full_plane > 0xFFFFFFF0ull) return fail(ctx, NYXHIP_ERR_ROI_TOO_LARGE, "bounding box too large for the contour plane");
            const size_t stride = al((size_t)full_plane);
            const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_large, ((size_t)4 << 30) / stride));
            const size_t sneed = stride * chunk;
            if (sneed > *spill_bytesp) {
                if (*spillp) { HIP_TRY(ctx, hipFree(*spillp)); *spillp = nullptr; *spill_bytesp = 0; }
                HIP_TRY(ctx, hipMalloc((void**)spillp, sneed));
                *spill_bytesp = sneed;
            }
            MomArgs m2 = m;
            m2.plane_cap = (uint32_t)full_plane;
            m2.sp.defer_large = 0;
            m2.sp.scratch = *spillp; m2.sp.stride = stride;
            for (uint32_t o = 0; o < n_large && rc == 0; o += chunk) {
                m2.sp.roi_index = d_list + o;
                rc = launch_roi_contour(m2, st_big, std::min(chunk, n_large - o));
            }
            if (rc == 0) {                                    // moments of the big boxes: the list
                MomArgs m3 = m;
                m3.sp.roi_index = d_list;
                rc = launch_roi_moments(m3, st_big, n_large);
            }
        }
        if (rc == 0) {
            m.sp.defer_large = 1;                             // both kernels skip the big boxes
            rc = launch_roi_contour(m, st, grid);
            if (rc == 0) rc = launch_roi_moments(m, st, grid);
        }
    }
    if (rc != 0)
        return fail(ctx, NYXHIP_ERR_HIP, std::string("moments kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
    return NYXHIP_OK;
}

// ---- size classes ------------------------------------------------------------------------------------------------------------
// The reference has no coupling between the ROIs of a batch: every worker thread takes ROIs of any size
// (/root/reference/src/nyx/parallel.h:23-42, roi_cache.h:31-84).  Here a launch carves its LDS for the largest ROI it holds, so
// a call is split into launches per SIZE CLASS: a classifier kernel sorts the ROI indices into five size classes x
// {16-bit tables possible, not possible} by each ROI's OWN pixel count, box and intensity range (never by its companions), and
// every class is launched over its index list with a carve-out -- hence kernel build and occupancy -- of its own.  Classes whose
// carve-out does not fit a CU's LDS run the same kernels with their scratch in a global workspace.
enum { H_COUNT = 0, H_OFFSET, H_PX, H_AREA, H_RANGE, H_SIDE, H_CURSOR, H_VMAX,
       H_SUMPX, H_SUMPX_HI, H_SUMAREA, H_SUMAREA_HI, H_SUMRANGE, H_SUMRANGE_HI, H_WORDS };   // header words per class (the three sums: 64 bits, even offsets)

// pass 1: members and extrema of every class (block-local in LDS first: ten hot words would serialise 5 n_roi global atomics)
// lvl_on: IBSI levels matter to the call (a texture family under IBSI): an ROI's largest intensity is its level count (roi_class)
__global__ void class_count_kernel(uint64_t n_roi, const uint64_t* px_offset, const uint32_t* bw, const uint32_t* bh, const uint32_t* mn,
                                   const uint32_t* mx, uint32_t* hdr, uint32_t lvl_on)
{
    __shared__ __attribute__((aligned(8))) uint32_t s_h[kClasses * H_WORDS];
    for (int i = threadIdx.x; i < kClasses * H_WORDS; i += blockDim.x) s_h[i] = 0;
    __syncthreads();
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n_roi) {
        const uint64_t n64 = px_offset[i + 1] - px_offset[i];
        const uint32_t n = n64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n64, w = bw[i], h = bh[i], r = mx[i] - mn[i];
        const uint64_t a64 = (uint64_t)w * h;
        uint32_t* c = s_h + roi_class(n, w, h, r, lvl_on ? mx[i] : 0u) * H_WORDS;
        atomicAdd(&c[H_COUNT], 1u);
        atomicAdd((unsigned long long*)&c[H_SUMPX], (unsigned long long)n);
        atomicAdd((unsigned long long*)&c[H_SUMAREA], (unsigned long long)a64);
        if (r < kLargeRangeMax) atomicAdd((unsigned long long*)&c[H_SUMRANGE], (unsigned long long)r + 1ull);
        atomicMax(&c[H_PX], n);
        atomicMax(&c[H_AREA], a64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a64);
        atomicMax(&c[H_RANGE], r);
        atomicMax(&c[H_SIDE], w > h ? w : h);
        atomicMax(&c[H_VMAX], mx[i]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kClasses * H_WORDS; k += blockDim.x) {
        const int f = k % H_WORDS;
        if (f == H_SUMPX || f == H_SUMAREA || f == H_SUMRANGE) {
            const unsigned long long v = *(const unsigned long long*)&s_h[k];
            if (v) atomicAdd((unsigned long long*)&hdr[k], v);
            continue;
        }
        if (s_h[k] == 0) continue;
        if (f == H_COUNT) atomicAdd(&hdr[k], s_h[k]);
        else if ((f >= H_PX && f <= H_SIDE) || f == H_VMAX) atomicMax(&hdr[k], s_h[k]);
    }
}

// pass 2: ROI indices grouped by class (class c occupies list[offset_c .. offset_c + count_c), offsets = prefix of the counts);
// blocks reserve their ranges in arrival order, so a class's list follows the batch order closely (neighbouring workgroups of a
// launch read neighbouring clouds) without being a function of it -- rows are addressed by ROI index, the order is free.
__global__ void class_scatter_kernel(uint64_t n_roi, const uint64_t* px_offset, const uint32_t* bw, const uint32_t* bh, const uint32_t* mn,
                                     const uint32_t* mx, uint32_t* hdr, uint32_t* list, uint32_t lvl_on)
{
    __shared__ uint32_t s_cnt[kClasses], s_base[kClasses];
    if (threadIdx.x < kClasses) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    int c = -1;
    uint32_t rank = 0;
    if (i < n_roi) {
        const uint64_t n64 = px_offset[i + 1] - px_offset[i];
        c = roi_class(n64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n64, bw[i], bh[i], mx[i] - mn[i], lvl_on ? mx[i] : 0u);
        rank = atomicAdd(&s_cnt[c], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kClasses) {
        uint32_t off = 0;
        for (int k = 0; k < (int)threadIdx.x; k++) off += hdr[k * H_WORDS + H_COUNT];
        if (blockIdx.x == 0) hdr[threadIdx.x * H_WORDS + H_OFFSET] = off;
        s_base[threadIdx.x] = off + (s_cnt[threadIdx.x] ? atomicAdd(&hdr[threadIdx.x * H_WORDS + H_CURSOR], s_cnt[threadIdx.x]) : 0u);
    }
    __syncthreads();
    if (c >= 0) list[s_base[c] + rank] = (uint32_t)i;
}

static void set_slots(SpillArgs& sp, const uint32_t* list, uint32_t n_slots)
{
    sp.roi_index = list; sp.n_slots = n_slots;
}

// Extrema every member of size class cls / 2 stays below (roi_class): what "does this class run from LDS?" is decided on, so that
// the answer is a function of the class -- i.e. of the ROI -- and the settings, never of the members a call happens to hold.
static Extrema class_bounds(int cls, const nyxhip_settings* s)
{
    const int sc = cls / 2;
    Extrema E{};
    E.px = kClassPx[sc]; E.side = kClassSide[sc]; E.area = kClassSide[sc] * kClassSide[sc];
    E.range = (cls & 1) ? (sc == 2 ? 65535u : 0xFFFFFFFFu) : 16383u;
    E.vmax = s->ibsi ? kLdsLevels : 0u;
    E.wide_only = (cls & 1) != 0;
    return E;
}

// INTENSITY + GLCM of one class by the several-workgroups-per-ROI kernels of roi_large.hip.  Members whose intensity range the
// histogram workspace does not hold (kLargeRangeMax) are left to the caller (the one-workgroup sort path).
static int run_large(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s, double* d_out, size_t ld, const Extrema& E,
                     const ClassTotals& tot, const uint32_t* list, uint32_t count, bool* served)
{   // *served = false: the settings do not fit the path's kernels (nothing launched; the one-workgroup workspace path takes the class)
    *served = true;
    const uint32_t mask1 = mask & (NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM);
    if (!mask1 || !count) return NYXHIP_OK;
    hipStream_t st = ctx->stream();
    LargeArgs a;
    memset(&a, 0, sizeof(a));
    a.n_roi = b->n_roi;
    a.px_offset = b->px_offset; a.x = b->x; a.y = b->y; a.inten = b->inten;
    a.bbox_w = b->bbox_w; a.bbox_h = b->bbox_h; a.min_inten = b->min_inten; a.max_inten = b->max_inten;
    a.slide_min = b->slide_min; a.slide_max = b->slide_max;
    a.out = d_out; a.ld = ld; a.status = ctx->d_status;
    a.mask = mask1; a.n_cols = nyxhip_n_columns(mask1, s);
    a.col_intensity = (mask1 & NYXHIP_FAM_INTENSITY) ? 0 : -1;
    a.col_glcm = (mask1 & NYXHIP_FAM_GLCM) ? ((mask1 & NYXHIP_FAM_INTENSITY) ? kIntensityCols : 0) : -1;
    a.soft_nan = s->soft_nan;
    a.grey_depth = s->grey_depth; a.ibsi = s->ibsi; a.glcm_grey_depth = s->glcm_grey_depth;
    a.glcm_offset = s->glcm_offset; a.glcm_na = s->glcm_n_angles; a.glcm_symmetric = s->glcm_symmetric;
    for (int i = 0; i < kMaxAngles; i++) a.glcm_angles[i] = s->glcm_angles[i];
    a.n_hist = abs(s->grey_depth);
    if (const char* e = getenv("NYXHIP_LARGE_DBG")) a.dbg = (uint32_t)atoi(e);
    a.vec_ok = (((uintptr_t)b->inten & 15u) == 0 && ((uintptr_t)b->x & 7u) == 0 && ((uintptr_t)b->y & 7u) == 0) ? 1u : 0u;
    const bool do_int = mask1 & NYXHIP_FAM_INTENSITY, do_glcm = mask1 & NYXHIP_FAM_GLCM;
    const int greyInfo = s->ibsi ? 0 : s->grey_depth;
    const uint32_t ng_max = greyInfo != 0 ? (uint32_t)abs(greyInfo) : E.vmax;       // largest matrix order of the class
    const uint32_t lvl_cap = greyInfo < 0 ? (uint32_t)(-greyInfo) : 0u;
    const uint32_t na = (uint32_t)s->glcm_n_angles;
    a.plane16 = ng_max > 255 ? 1u : 0u;
    // load kernel: histogram counted in LDS with 16-bit counters (a slab holds < 65536 pixels); up to 16384 entries at 256 threads and
    // 8192 pixels per slab, up to 65536 entries (128 KiB) at 1024 threads and 32768 pixels per slab (16-bit data: the flush of the
    // table -- one atomic add per non-empty entry -- must not outweigh the slab's pixels)
    const uint32_t r_max = std::min(E.range, kLargeRangeMax - 1);
    // Slab size: 32 pixels per thread at 256 / 512 / 1024 threads.  Large slabs flush the table less often (the flush of a 12-bit
    // table is 16 KB of atomic traffic per slab -- at 8192 pixels a quarter of what the slab reads); small slabs fill the chip when
    // the class holds few pixels: as large as leaves ~2000 workgroups.
    a.px_per_wg = tot.px / 2048 >= 32768 ? 32768u : tot.px / 2048 >= 16384 ? 16384u : 8192u;
    if (!do_int) a.tab_lds = 0;
    else if (r_max < 16384u) a.tab_lds = (r_max + 1 + 63u) & ~63u;
    else { a.tab_lds = 65536; a.px_per_wg = 32768; }
    a.lds_P_bytes = do_glcm ? (uint32_t)std::min<uint64_t>(72 * 1024, ((4ull * na * (ng_max + 1ull) * (ng_max + 1ull) + 15) & ~15ull) + ((2ull * (lvl_cap + 2) + 15) & ~15ull)) : 0u;
    // ... plus the strip of plane rows with its halo rows: kLargeCells cells + 2 * offset rows of the class's widest plane
    a.lds_strip_bytes = do_glcm ? (uint32_t)std::min<uint64_t>(40 * 1024, (a.plane16 ? 2ull : 1ull) * (kLargeCells + 2ull * (uint64_t)std::max(s->glcm_offset, 0) * std::min<uint32_t>(E.side, kLargeCells)) + 64) : 0u;
    // finishing kernel: histogram bin bounds, then (over the same bytes) the GLCM feature scratch and the matrices themselves
    {
        // (an ROI keeps its scratch in LDS when it needs at most kLargeScratchLds -- its own matrix order decides, roi_large.hip)
        const uint64_t scr = do_glcm ? ((std::min<uint64_t>(large_glcm_scratch_bytes(ng_max), kLargeScratchLds) + 15) & ~15ull) : 0ull;
        const uint64_t pm = 4ull * na * ng_max * ng_max;
        a.fin_P_bytes = (do_glcm && pm <= 32 * 1024) ? (uint32_t)pm : 0u;
        a.fin_tab_bytes = do_int ? (uint32_t)((4ull * (std::min<uint32_t>(r_max, 8191u) + 1) + 15) & ~15ull) : 0u;   // up to 32 KiB of histogram
        // (the bin bounds of the n-bin histogram sit behind the table: with thousands of bins the table shrinks, beyond ~16000 the
        //  finishing kernel's 64 KiB cannot hold the bounds at all -- round-4 advisor)
        const uint64_t bounds = 4ull * (112 + (uint64_t)abs(s->grey_depth));
        if (do_int && bounds + 16 > 64 * 1024) { *served = false; return NYXHIP_OK; }
        if (do_int && a.fin_tab_bytes + bounds > 64 * 1024) a.fin_tab_bytes = (uint32_t)((64 * 1024 - bounds) & ~15ull);
        a.lds_fin_bytes = (uint32_t)std::max<uint64_t>(a.fin_tab_bytes + bounds, scr + a.fin_P_bytes + 16);
    }
    // ---- workspace: the members' blocks back to back (offsets handed out by the prep kernel) when the class fits the budget, else
    // chunks of the list with room for the class's largest block each
    const char* const be = getenv("NYXHIP_LARGE_BUDGET_MB");                  // (tests: a small budget sends a class through the chunked form)
    const size_t budget = be && atoll(be) > 0 ? (size_t)atoll(be) << 20 : (size_t)8 << 30;
    const LargeWs Lmax = large_ws_layout(r_max, E.area, ng_max, lvl_cap, na, a.plane16 != 0, do_int, do_glcm);
    const LargeWs Lfix = large_ws_layout(0, 0, ng_max, lvl_cap, na, a.plane16 != 0, do_int, do_glcm);   // what every block holds whatever its ROI
    const uint64_t all_bytes = (uint64_t)count * (Lfix.total + 1024) + (do_int ? 4ull * tot.range1 : 0ull) + (do_glcm ? (a.plane16 ? 2ull : 1ull) * tot.area : 0ull);
    uint32_t chunk = count;
    uint64_t ws_need = all_bytes;
    if (all_bytes > budget) {
        if (Lmax.total > budget) return fail(ctx, NYXHIP_ERR_ROI_TOO_LARGE, "an ROI's workspace block (" + std::to_string(Lmax.total >> 20) + " MiB) exceeds the large-ROI budget");
        chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(count, budget / Lmax.total));
        ws_need = (uint64_t)chunk * Lmax.total;
    }
    const uint64_t slabs_max = ((uint64_t)E.px + 3 + a.px_per_wg - 1) / a.px_per_wg, strips_max = 2ull * E.area / kLargeCells + 1;
    uint64_t cap_load = chunk == count ? tot.px / a.px_per_wg + 2ull * count : (uint64_t)chunk * slabs_max;   // (a slab more per ROI: slabs start at a multiple of four pixels)
    uint64_t cap_cooc = do_glcm ? (chunk == count ? 2 * tot.area / kLargeCells + count : (uint64_t)chunk * strips_max) : 0;
    if (cap_load > 0x7FFFFFFFull || cap_cooc > 0x7FFFFFFFull) {          // (grid limit: smaller chunks)
        const uint64_t per = std::max(slabs_max, strips_max);
        chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(chunk, 0x7FFFFFFFull / per));
        cap_load = (uint64_t)chunk * slabs_max; cap_cooc = do_glcm ? (uint64_t)chunk * strips_max : 0;
        ws_need = std::min<uint64_t>(ws_need, (uint64_t)chunk * Lmax.total);
        if (slabs_max > 0x7FFFFFFFull || strips_max > 0x7FFFFFFFull) return fail(ctx, NYXHIP_ERR_ROI_TOO_LARGE, "ROI too large for the large-ROI launch grid");
    }
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_ctr = 0, o_off = 256, o_ml = al(o_off + 8ull * chunk), o_mc = al(o_ml + 8ull * cap_load), aux_need = al(o_mc + 8ull * cap_cooc);
    if (ws_need > ctx->large_bytes) {
        if (ctx->d_large) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(ctx->d_large)); ctx->d_large = nullptr; ctx->large_bytes = 0; }
        HIP_TRY(ctx, hipMalloc(&ctx->d_large, ws_need));
        ctx->large_bytes = ws_need;
    }
    if (aux_need > ctx->large_aux_bytes) {
        if (ctx->d_large_aux) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(ctx->d_large_aux)); ctx->d_large_aux = nullptr; ctx->large_aux_bytes = 0; }
        HIP_TRY(ctx, hipMalloc(&ctx->d_large_aux, aux_need + aux_need / 4));
        ctx->large_aux_bytes = aux_need + aux_need / 4;
    }
    char* const aux = (char*)ctx->d_large_aux;
    a.ws = (unsigned char*)ctx->d_large; a.ws_bytes = ws_need;
    a.ctr = (uint32_t*)(aux + o_ctr); a.ws_off = (uint64_t*)(aux + o_off);
    a.map_load = (uint2*)(aux + o_ml); a.map_cooc = (uint2*)(aux + o_mc);
    a.cap_load = (uint32_t)cap_load; a.cap_cooc = (uint32_t)cap_cooc;
    for (uint32_t o = 0; o < count; o += chunk) {
        a.list = list + o; a.n_list = std::min(chunk, count - o);
        HIP_TRY(ctx, hipMemsetAsync(a.ws, 0, ws_need, st));
        HIP_TRY(ctx, hipMemsetAsync(a.ctr, 0, 256, st));
        if (int rc = launch_large_features(a, st))
            return fail(ctx, NYXHIP_ERR_HIP, std::string("large-ROI kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
    }
    return NYXHIP_OK;
}

// Gabor of the ROIs `list[0 .. grid)` by several workgroups per ROI (roi_large_gabor.hip): tiles of 64 x 32 output pixels staged in LDS,
// the reference's arithmetic per pixel, integer counts across workgroups.  Boxes beyond 8192 px a side (or kernels beyond 32 taps) are
// left to the caller's one-workgroup kernel (*served stays false).  `g`: the shape arguments of the class (columns, bank, threshold).
static int run_large_gabor(nyxhip_ctx* ctx, const nyxhip_batch* b, const nyxhip_settings* s, double* d_out, size_t ld, const Extrema& E, const ShapeArgs& g,
                           const uint32_t* list, uint32_t grid, hipStream_t st, unsigned char** bufp, size_t* bytesp, bool* served)
{
    *served = false;
    const bool no_coop_gabor = [] { const char* e = getenv("NYXHIP_NO_COOP_GABOR"); return e && *e && *e != '0'; }();   // A/B knob (read per call: the tests compare both paths in one process)
    if (no_coop_gabor || !list || grid == 0 || E.side > kLgabMaxSide || (uint32_t)s->gabor_kersize > kLgabMaxN) return NYXHIP_OK;
    LgabArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.n_roi = b->n_roi; ga.px_offset = b->px_offset; ga.x = b->x; ga.y = b->y; ga.inten = b->inten;
    ga.bbox_w = b->bbox_w; ga.bbox_h = b->bbox_h; ga.min_inten = b->min_inten; ga.max_inten = b->max_inten;
    ga.out = d_out; ga.ld = ld; ga.col_gabor = g.col_gabor; ga.nf = s->gabor_n_filters; ga.n = s->gabor_kersize;
    ga.thr = g.gabor_thr; ga.soft_nan = s->soft_nan; ga.bank = g.gabor_bank;
    const uint64_t area_cap = std::max<uint64_t>(E.area, 1);
    // tiles of a w x h box: ceil(w / 64) ceil(h / 32) <= w h / 2048 + w / 64 + h / 32 + 1, and w, h <= side, w h <= area for every ROI of the class
    ga.tiles_cap = (uint32_t)std::min<uint64_t>((uint64_t)((E.side + kLgabTileW - 1) / kLgabTileW) * ((E.side + kLgabTileH - 1) / kLgabTileH),
                                                area_cap / (kLgabTileW * kLgabTileH) + E.side / kLgabTileW + E.side / kLgabTileH + 2);
    ga.off_rec = (4 * area_cap + 255) & ~255ull;
    ga.off_cnt = (ga.off_rec + 24ull * ga.tiles_cap + 255) & ~255ull;
    ga.stride = (ga.off_cnt + 4ull * (uint64_t)(ga.nf + 1) + 255) & ~255ull;
    const size_t gbudget = (size_t)2 << 30;
    const uint32_t gchunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min<uint32_t>(grid, 65535u), gbudget / ga.stride));
    const size_t gneed = ga.stride * gchunk;
    if (gneed > *bytesp) {
        if (*bufp) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(*bufp)); *bufp = nullptr; *bytesp = 0; }
        HIP_TRY(ctx, hipMalloc((void**)bufp, gneed));
        *bytesp = gneed;
    }
    ga.ws = *bufp;
    for (uint32_t o = 0; o < grid; o += gchunk) {
        const uint32_t nb = std::min(gchunk, grid - o);
        set_slots(ga.sp, list + o, nb);
        if (int rc = launch_large_gabor(ga, st, nb, E.px))
            return fail(ctx, NYXHIP_ERR_HIP, std::string("large-ROI Gabor launch failed: ") + hipGetErrorString((hipError_t)rc));
    }
    *served = true;
    return NYXHIP_OK;
}

// GLRLM + GLSZM + NGTDM of one class by the several-workgroups-per-ROI kernels of roi_large_tex.hip, on stream `st` with the
// workspace pair `slot`.  *served: 0 = the class does not qualify (nothing launched), 1 = every member was served, 2 = the members
// outside ltex_eligible are left to the caller (the one-workgroup launch with SpillArgs::skip_ltex).
static int run_large_tex(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t full, const nyxhip_settings* s, double* d_out, size_t ld, const Extrema& E,
                         const ClassTotals& tot, const uint32_t* list, uint32_t count, hipStream_t st, int slot, int* served)
{
    *served = 0;
    const uint32_t mask2 = full & kTexture;
    if (!mask2 || !count) return NYXHIP_OK;
    const int greyInfo = s->ibsi ? 0 : s->grey_depth;
    const uint32_t ng = greyInfo != 0 ? (uint32_t)abs(greyInfo) : E.vmax;         // bound of the class's level counts
    if (ng == 0 || ng > kLtexLevels) return NYXHIP_OK;
    LtexArgs a;
    memset(&a, 0, sizeof(a));
    a.n_roi = b->n_roi;
    a.px_offset = b->px_offset; a.x = b->x; a.y = b->y; a.inten = b->inten;
    a.bbox_w = b->bbox_w; a.bbox_h = b->bbox_h; a.min_inten = b->min_inten; a.max_inten = b->max_inten;
    a.out = d_out; a.ld = ld; a.status = ctx->d_status;
    a.mask = mask2; a.n_cols = nyxhip_n_columns(mask2, s);
    a.col0 = nyxhip_n_columns(full & (NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM), s);
    a.gap_after_glrlm = (full & NYXHIP_FAM_GLDZM) ? kGldzmCols : 0;
    a.gap_after_glszm = ((full & NYXHIP_FAM_GLDM) ? kGldmCols : 0) + ((full & NYXHIP_FAM_NGLDM) ? kNgldmCols : 0);
    a.soft_nan = s->soft_nan; a.grey_depth = s->grey_depth; a.ibsi = s->ibsi;
    a.vec_ok = (((uintptr_t)b->inten & 15u) == 0 && ((uintptr_t)b->x & 7u) == 0 && ((uintptr_t)b->y & 7u) == 0) ? 1u : 0u;
    a.plane16 = ng > 255 ? 1u : 0u;
    a.px_per_wg = 8192;
    const uint64_t cb = a.plane16 ? 2 : 1;
    auto al16 = [](uint64_t v) { return (v + 15) & ~15ull; };
    auto al256 = [](uint64_t v) { return (v + 255) & ~255ull; };
    // ---- dynamic LDS at the bounds of what the path serves in this class (box sides of eligible members: <= kLtexMaxW wide)
    const uint64_t side_w = std::min<uint32_t>(E.side, kLtexMaxW);
    const uint32_t ng1 = ng + 1;
    const uint64_t ngt_rep = ng1 <= 16 ? 8 : ng1 <= 32 ? 4 : ng1 <= 64 ? 2 : 1, ngt_stride = (((ng1 + 2) * 12ull + 16 + 7) & ~7ull) | 8;
    const uint64_t cells = std::max<uint64_t>(kLtexCells, side_w) + 2 * side_w + 64;       // a strip's rows and its two halo rows, staged as they lie in the plane
    // (under IBSI a member's level count is its own largest intensity, anything up to the class's: every term below covers the
    //  smaller counts as well)
    uint64_t strip = al16(2ull * (ng + 2)) + al16(cb * cells);
    if (mask2 & NYXHIP_FAM_NGTDM) strip += ng1 <= 1024 ? std::max<uint64_t>(2048, al16(ngt_rep * ngt_stride)) : al16(12ull * 1026 + 32);
    // (replicated short-run tables, roi_large_tex.hip: ltex_rl_rep x ltex_rl_words -- at most 8 x 65 x 16 words, reached at 16 levels)
    const uint64_t rl_bytes = (mask2 & NYXHIP_FAM_GLRLM) ? al16(4ull * 8 * 65 * std::min<uint32_t>(std::max<uint32_t>(ng, 1), 16)) : 0;
    strip += rl_bytes;
    // sweep: two rows of owners, the chunk hand-overs, the ring of plane rows (rows of up to 1008 bytes: 1040-byte slots)
    const uint64_t sweep = (mask2 & NYXHIP_FAM_GLSZM) ? al16(8 * (side_w + 2) + 16 * (side_w / 64 + 4)) + 1040 * (8 + side_w / 64 + 2 + 5) : 0;
    // a wave per 64 columns of the class's widest box, and the wave that feeds the ring
    a.strip_threads = 64u * (uint32_t)(std::min<uint64_t>(15, std::max<uint64_t>(3, (std::min<uint64_t>(side_w, 1024) + 62) / 64 + 1)) + 1);
    a.lds_load_bytes = 32 * 1024;
    if (!(mask2 & NYXHIP_FAM_GLSZM)) a.strip_threads = 256;            // no sweep in the launch
    a.strip_groups = std::max<uint32_t>(1, a.strip_threads / 256);
    a.lds_group_bytes = (uint32_t)((strip + 64 + 15) & ~15ull);
    while (a.strip_groups > 1 && (uint64_t)a.strip_groups * a.lds_group_bytes > 144 * 1024) a.strip_groups--;
    const uint64_t lds_strip = std::max<uint64_t>((uint64_t)a.strip_groups * a.lds_group_bytes, sweep + 64);
    const uint64_t S = ng <= 256 ? kLtexSmall : 0;
    // (tables that exist only below a level count: under IBSI a member may have any count up to the class's, else all have ng)
    const bool any_ng = greyInfo == 0;
    const uint64_t small_b = (mask2 & NYXHIP_FAM_GLSZM) ? (any_ng ? 4ull * std::min<uint32_t>(ng, 256) * kLtexSmall : ng <= 256 ? 4ull * ng * kLtexSmall : 0) : 0;
    const uint64_t rl2_b = (mask2 & NYXHIP_FAM_GLRLM) ? (any_ng ? 4ull * 2 * 65 * std::min<uint32_t>(ng, 128) : ng <= 128 ? 4ull * 2 * 65 * ng : 0) : 0;   // two replicas of the run table
    const uint64_t lds_zone = al16(2ull * (ng + 2)) + al16(small_b) + al16(rl2_b) +
                              ((mask2 & NYXHIP_FAM_GLSZM) ? al16(2 * std::max<uint64_t>(kLtexCells, side_w) + 8) : 0) + 64;   // ... and the strip's 16-bit zone counters
    const uint64_t side_e = std::min<uint64_t>(std::max<uint32_t>(E.side, 1), (1u << 20) - 1);     // an eligible box has fewer than 2^20 cells
    const uint64_t slot_max = (uint64_t)ng * side_e + ng + side_e + 4;
    const uint64_t fin_fixed = al16(8ull * a.n_cols) + al16(2ull * (ng + 2)) + al16(4ull * (ng + 2)) + al16(16ull * (std::min<uint32_t>(ng, 256) + 2));
    const uint64_t fin_work = std::max<uint64_t>(16ull * (ng + 2), std::min<uint64_t>(16 * slot_max, 64 * 1024));
    const uint64_t lds_fin = fin_fixed + al16(fin_work) + 64;
    if (getenv("NYXHIP_DEBUG")) fprintf(stderr, "[nyxhip] large texture: count %u ng %u side %u area %u lds strip %llu zone %llu fin %llu\n", count, ng, E.side, E.area,
                                        (unsigned long long)lds_strip, (unsigned long long)lds_zone, (unsigned long long)lds_fin);
    if (lds_strip > 144 * 1024 || lds_zone > 128 * 1024 || lds_fin > 144 * 1024) return NYXHIP_OK;
    a.lds_strip_bytes = (uint32_t)lds_strip; a.lds_zone_bytes = (uint32_t)lds_zone; a.lds_fin_bytes = (uint32_t)lds_fin;
    // ---- workspace: bounds of a member's block and of the class as a whole (ltex_ws_layout)
    const uint64_t area_e = std::min<uint64_t>(std::max<uint32_t>(E.area, 1), (1u << 20) - 1);
    const uint64_t rmin = std::max<uint64_t>(1, kLtexCells / std::max<uint64_t>(side_w, 1));
    const uint64_t fixed = 256 + al256(ng + 8) + al256(12ull * (ng + 2)) + al256(16 * slot_max) + al256(4ull * std::min<uint32_t>(ng, 256) * kLtexSmall) +
                           al256(8ull * szm_hash_cap(ng + 1, (uint32_t)area_e)) + 12 * 256;
    auto var_bytes = [&](uint64_t area, uint64_t members) {
        return cb * area + 64 * members + 8 * (area + 2 * members) + 4 * (area / (S + 1) + 8 * members) + 28 * (area / rmin + members * side_w) + 64 * members;
    };
    const char* const be = getenv("NYXHIP_LARGE_BUDGET_MB");                  // (tests: a small budget sends a class through the chunked form)
    const size_t budget = be && atoll(be) > 0 ? (size_t)atoll(be) << 20 : (size_t)2 << 30;   // per slot, kept until nyxhip_destroy (nine slots: 8 GiB each could pin 72 GiB of a context)
    const uint64_t all_bytes = (uint64_t)count * fixed + var_bytes(tot.area, count);
    const uint64_t one_max = fixed + var_bytes(area_e, 1);
    uint32_t chunk = count;
    uint64_t ws_need = all_bytes;
    if (all_bytes > budget) {
        if (one_max > budget) return NYXHIP_OK;              // (the one-workgroup path serves the class)
        chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(count, budget / one_max));
        ws_need = (uint64_t)chunk * one_max;
    }
    const uint64_t slabs_max = ((uint64_t)E.px + 3 + a.px_per_wg - 1) / a.px_per_wg, strips_max = area_e / 4096 + 2;
    uint64_t cap_load = chunk == count ? tot.px / a.px_per_wg + 2ull * count : (uint64_t)chunk * slabs_max;
    uint64_t cap_strip = chunk == count ? tot.area / 4096 + 2ull * count : (uint64_t)chunk * strips_max;
    if (cap_load > 0x3FFFFFFFull || cap_strip > 0x3FFFFFFFull) return NYXHIP_OK;
    if (ws_need > ctx->ltex_bytes[slot]) {
        if (ctx->ltex_buf[slot]) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(ctx->ltex_buf[slot])); ctx->ltex_buf[slot] = nullptr; ctx->ltex_bytes[slot] = 0; }
        // a device short of memory: smaller chunks (the form the budget already knows); when not even one member's block can be had the
        // class goes to the one-workgroup kernels (*served stays 0) instead of failing the call
        while (hipMalloc(&ctx->ltex_buf[slot], ws_need) != hipSuccess) {
            (void)hipGetLastError();
            ctx->ltex_buf[slot] = nullptr;
            if (chunk <= 1) return NYXHIP_OK;
            chunk = (chunk + 1) / 2;
            ws_need = (uint64_t)chunk * one_max;
            cap_load = (uint64_t)chunk * slabs_max; cap_strip = (uint64_t)chunk * strips_max;
        }
        ctx->ltex_bytes[slot] = ws_need;
    }
    const size_t o_ctr = 0, o_off = 256, o_ml = al256(o_off + 8ull * chunk), o_ms = al256(o_ml + 8ull * cap_load), aux_need = al256(o_ms + 8ull * cap_strip);
    if (aux_need > ctx->ltex_aux_bytes[slot]) {
        if (ctx->ltex_aux[slot]) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(ctx->ltex_aux[slot])); ctx->ltex_aux[slot] = nullptr; ctx->ltex_aux_bytes[slot] = 0; }
        HIP_TRY(ctx, hipMalloc(&ctx->ltex_aux[slot], aux_need + aux_need / 4));
        ctx->ltex_aux_bytes[slot] = aux_need + aux_need / 4;
    }
    char* const aux = (char*)ctx->ltex_aux[slot];
    a.ws = (unsigned char*)ctx->ltex_buf[slot]; a.ws_bytes = ws_need;
    a.ctr = (uint32_t*)(aux + o_ctr); a.ws_off = (uint64_t*)(aux + o_off);
    a.map_load = (uint2*)(aux + o_ml); a.map_strip = (uint2*)(aux + o_ms);
    a.cap_load = (uint32_t)cap_load; a.cap_strip = (uint32_t)cap_strip;
    for (uint32_t o = 0; o < count; o += chunk) {
        a.list = list + o; a.n_list = std::min(chunk, count - o);
        HIP_TRY(ctx, hipMemsetAsync(a.ws, 0, ws_need, st));
        HIP_TRY(ctx, hipMemsetAsync(a.ctr, 0, 256, st));
        if (int rc = launch_large_texture(a, st))
            return fail(ctx, NYXHIP_ERR_HIP, std::string("large-ROI texture kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
    }
    *served = (E.side <= kLtexMaxW && (uint64_t)E.area < (1ull << 20)) ? 1 : 2;
    if (getenv("NYXHIP_DEBUG")) fprintf(stderr, "[nyxhip] large texture: served %d, workspace %llu MiB, chunk %u, caps %llu / %llu\n", *served, (unsigned long long)(ws_need >> 20), chunk,
                                        (unsigned long long)cap_load, (unsigned long long)cap_strip);
    return NYXHIP_OK;
}

// One launch group.
//   list != NULL: the members of one class (cls), `grid` of them (exact launches).  Which kernel groups of the class run from LDS is
//   decided on the class BOUNDS (class_bounds); the carve-outs then follow the class's extrema E.  A group that does not fit runs
//   from a global workspace: INTENSITY + GLCM by the several-workgroups-per-ROI path (run_large), the others with one workgroup
//   per ROI.
//   list == NULL: the whole batch (slot = ROI, grid = n_roi); class_mask != 0 then restricts the launch to the classes of the
//   mask (SpillArgs::class_mask: everybody else returns at once) -- used when the host launches without knowing the member
//   counts; such a launch cannot fall back to the workspace (its chunks are sized by member counts): *needs_host is set instead
//   and nothing is launched.  dry: build the argument blocks only (do the carve-outs fit?).
//   group_sel: kernel groups to launch (bit 0 features, 1 texture, 2 shape, 3 dependence).
int run_class(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s, double* d_out, size_t ld, const Extrema& E,
              const uint32_t* list, uint32_t grid, bool dry, bool* needs_host, ClassRun* report, uint32_t class_mask = 0, uint32_t group_sel = 0xF,
              int cls = -1, const ClassTotals* tot = nullptr)
{
    std::string why;
    const uint32_t full = mask;                          // column positions follow the call's full mask: build_args always gets it
    if (!(group_sel & 1)) mask &= ~(uint32_t)(NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM);
    if (!(group_sel & 2)) mask &= ~kTexture;
    if (!(group_sel & 4)) mask &= ~kShape;
    if (!(group_sel & 8)) mask &= ~kDependence;
    const uint32_t fam_of_group[4] = {NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM, kTexture, kShape, kDependence};
    uint32_t want = 0;                                   // kernel groups this call has work for
    for (int k = 0; k < 4; k++) if (mask & fam_of_group[k]) want |= 1u << k;
    if (!want) return NYXHIP_OK;
    RoiArgs a; TexArgs t; ShapeArgs g; DepArgs d;
    uint32_t gs = 0;                                     // groups that run from a global workspace
    if (list) {
        const Extrema Eb = class_bounds(cls, s);
        for (int k = 0; k < 4; k++) {
            if (!(want & (1u << k))) continue;
            if (cls / 2 >= kFirstLargeSizeClass) { gs |= 1u << k; continue; }
            const int lrc = build_args(ctx, b, full, s, d_out, ld, Eb, 0, a, t, g, d, why, 1u << k);
            if (lrc == NYXHIP_ERR_ROI_TOO_LARGE || lrc == NYXHIP_ERR_UNSUPPORTED) gs |= 1u << k;
            else if (lrc) return fail(ctx, lrc, why);
        }
    }
    uint32_t lds = want & ~gs;
    if (lds) {
        int lrc = build_args(ctx, b, full, s, d_out, ld, E, 0, a, t, g, d, why, lds);
        if (lrc == NYXHIP_ERR_UNSUPPORTED && (lds & 1) && list == nullptr) {
            // whole-batch launches: a GLCM grey depth whose matrix does not fit LDS next to any ROI sends the feature group to the workspace
            gs |= 1; lds &= ~1u;
            lrc = lds ? build_args(ctx, b, full, s, d_out, ld, E, 0, a, t, g, d, why, lds) : NYXHIP_OK;
        }
        if (lrc == NYXHIP_ERR_UNSUPPORTED || lrc == NYXHIP_ERR_ROI_TOO_LARGE) {
            if (list == nullptr && lrc == NYXHIP_ERR_UNSUPPORTED) return fail(ctx, lrc, why);
            gs |= lds; lds = 0;                          // (exact launches: cannot happen -- the bounds fitted; served from the workspace all the same)
        } else if (lrc) return fail(ctx, lrc, why);
    }
    if (gs && needs_host) { *needs_host = true; return NYXHIP_OK; }
    if (gs && !list) return fail(ctx, NYXHIP_ERR_ROI_TOO_LARGE, "ROI too large for the LDS-resident path: " + why);
    if (dry) return NYXHIP_OK;
    if (gs && ctx->win_next.inten && !b->inten)          // window-mode call (no clouds materialised): the caller builds them and comes back
        return NYXHIP_INTERNAL_NEEDS_CLOUDS;
    if (report) report->workspace = (int)gs;

    hipStream_t st = ctx->stream();
    int rc = 0;
    auto enter_lane = [&](int lane) -> int { return use_lane(ctx, lane, &st); };   // the class's launches go to lane `lane`, forked from the main stream at the start of the call
    // The launch groups of the LDS size classes are independent of each other and could each take a stream of their own (the tail of
    // one class's grid beside the next class's launches).  Measured on the mixed batch with the config-4 families: 3.51 ms against
    // 3.28 ms on one stream -- the chip is busy either way, and the interleaved classes evict each other's L2 lines.  Off unless asked
    // for (NYXHIP_LDS_LANES=1, A/B knob).
    static const bool lds_lanes = [] { const char* e = getenv("NYXHIP_LDS_LANES"); return e && *e && *e != '0'; }();
    const bool on_lds_lane = lds && !gs && lds_lanes && list && cls >= 0 && cls / 2 < kFirstLargeSizeClass && !ctx->win_next.inten;
    if (on_lds_lane)
        if (int lrc = enter_lane(4 + cls / 2)) return lrc;
    // Size class 2 (boxes up to 128 x 128, up to 16384 pixels) fits LDS, but its texture kernel is one workgroup walking 16 k cells through
    // a dozen passes and a serial zone sweep: ~0.3 ms per ROI whatever the batch, and a class of two thousand such ROIs is a round and
    // a tail of them (0.7-0.9 ms of the mixed batch's 3.3).  Its GLRLM / GLSZM / NGTDM go through the several-workgroups-per-ROI path
    // as well (roi_large_tex.hip), on a lane beside the main stream.  A function of the class, i.e. of the ROI.
    static const bool no_sc2_tex = [] { const char* e = getenv("NYXHIP_NO_COOP_TEX_SC2"); const char* f = getenv("NYXHIP_NO_COOP_TEX"); const char* g0 = getenv("NYXHIP_NO_COOP");
                                        return (e && *e && *e != '0') || (f && *f && *f != '0') || (g0 && *g0 && *g0 != '0'); }();   // A/B knob
    if (list && cls / 2 == 2 && (lds & 2) && tot && !no_sc2_tex) {
        // (a function of the class alone: a window-mode chunk -- INTENSITY / GLCM only by construction, so never here with texture families --
        //  would come back with its clouds rather than take the other texture kernel)
        if (!b->inten) return NYXHIP_INTERNAL_NEEDS_CLOUDS;
        const hipStream_t main_st = st;
        int served2 = 0;
        // (the lane of size class 3 -- usually a handful of ROIs: streams beyond the device's four hardware queues share one, and a
        //  lane of its own landed on the queue of the largest class's lane, behind 2 ms of its kernels)
        const int lane2 = cls & 1;
        if (int lrc = enter_lane(lane2)) return lrc;
        if (int lrc = run_large_tex(ctx, b, full, s, d_out, ld, E, *tot, list, grid, st, lane2, &served2)) return lrc;
        if (served2 == 1) {
            lds &= ~2u;
            if (report) {
                report->cooperative |= 2;
                if (ctx->timing >= 2) { if (!report->e2) HIP_TRY(ctx, hipEventCreate(&report->e2)); HIP_TRY(ctx, hipEventRecord(report->e2, st)); }
            }
        }
        st = main_st;
    }
    // (Size class 2 and Gabor: a 128 x 128 box takes 87 KB of LDS in the tiled kernel -- one workgroup per CU, 10.9 ms for the 2 031 such
    //  ROIs of the heavy-tailed batch.  Cut into 64 x 32 tiles by run_large_gabor the same ROIs took 15.5 ms -- boxes of 65..127 px fill
    //  40 % of their tiles, and the strips compute every tap in fp64 where the tiled kernel screens on the matrix pipe: not routed there.)
    if (lds) {
        for (SpillArgs* sp : {&a.sp, &t.sp, &g.sp, &d.sp}) { set_slots(*sp, list, grid); sp->class_mask = list ? 0u : class_mask; }
        // the smallest size class (roi_class == 0) runs INTENSITY / GLCM a wave per ROI (roi_small.hip; launch_roi_features decides whether
        // the settings allow it): its exact list, a whole-batch launch filtered to it, or a whole batch that IS it by the stated extrema
        if (list ? cls == 0 : class_mask == 0x1u) a.small_class = 1;
        if (!list && class_mask == 0x1u) a.census = (uint32_t*)(ctx->d_status + 1);
        else if (!list && class_mask == 0 && E.px <= kClassPx[0] && E.side <= kClassSide[0] && E.range < 16384u) a.small_class = 2;
        // the two filtered feature launches of a call on stated extrema (launch_device_all): ONE GLCM feature launch, behind the second
        if (!list && class_mask == 0x1u) a.glcm_feats = 1;
        else if (!list && class_mask == 0x3FEu) a.glcm_feats = 2;
        // INTENSITY + GLCM at the reference's default grey depth (17..64 levels): two launches instead of one.  The 16-bit-matrix
        // kernel holds 43 KB of LDS per workgroup (three per CU); the intensity block inside it ran at that occupancy, 2.9 ms per
        // 196 k ROIs against 1.4 ms for the intensity-only build at eight workgroups per CU.  Each launch zeroes and fills its own
        // block of columns.
        auto launch_features_main = [&]() -> int {
            const uint32_t both = NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM;
            if ((a.mask & both) == both && a.L.g16 && !getenv("NYXHIP_G16_FUSED")) {
                RoiArgs ai = a, ag = a;
                std::string w2;
                const int ncol_g = a.n_cols - kIntensityCols;
                if (make_layout(NYXHIP_FAM_INTENSITY, s, kIntensityCols, E.px, E.area, E.range, ai.L, w2) == NYXHIP_OK &&
                    make_layout(NYXHIP_FAM_GLCM, s, ncol_g, E.px, E.area, E.range, ag.L, w2) == NYXHIP_OK && ag.L.g16) {
                    ai.mask = NYXHIP_FAM_INTENSITY; ai.n_cols = kIntensityCols; ai.col_intensity = 0; ai.col_glcm = -1;
                    ag.mask = NYXHIP_FAM_GLCM; ag.n_cols = ncol_g; ag.col_glcm = 0; ag.col_intensity = -1; ag.out = a.out + kIntensityCols;
                    ag.census = nullptr;                               // (the intensity launch counts)
                    if (int r1 = launch_roi_features(ag, st, grid)) return r1;
                    return launch_roi_features(ai, st, grid);
                }
            }
            return launch_roi_features(a, st, grid);
        };
        // Wide-range classes whose ranges fit 16 bits (16-bit microscopy data): the first-order features from a presence bitmap and a
        // duplicate list instead of a sort (roi_wide.hip), the GLCM columns from the GLCM-only build.  (The sort engine inside the
        // fused kernel cost 43 ns per 2821-pixel ROI against 12.5 ns on 12-bit data.)
        static const bool no_wide = [] { const char* e = getenv("NYXHIP_NO_WIDE"); return e && *e && *e != '0'; }();   // A/B knob
        // Which engine serves a member is decided by ITS range (<= 0xFFFF: roi_wide + the GLCM-only build; beyond: the fused sort
        // kernel), never by the class's observed extrema: both launches run over the class and each skips the other's members.
        auto launch_features_wide = [&](bool& done) -> int {
            done = false;
            if (no_wide || !list || !(cls & 1) || !(a.mask & NYXHIP_FAM_INTENSITY)) return 0;
            WideArgs wa;
            memset(&wa, 0, sizeof(wa));
            if (!make_wide_layout(E.px, (uint32_t)abs(s->grey_depth), wa)) return 0;
            RoiArgs ag = a;
            const bool with_glcm = (a.mask & NYXHIP_FAM_GLCM) != 0;
            if (with_glcm) {
                std::string w2;
                const int ncol_g = a.n_cols - kIntensityCols;
                if (make_layout(NYXHIP_FAM_GLCM, s, ncol_g, E.px, E.area, std::min(E.range, 0xFFFFu), ag.L, w2, 0, E.vmax) != NYXHIP_OK) return 0;
                ag.mask = NYXHIP_FAM_GLCM; ag.n_cols = ncol_g; ag.col_glcm = 0; ag.col_intensity = -1; ag.out = a.out + kIntensityCols;
                if (a.glcm_ws && ag.L.ng_cap != a.L.ng_cap) return 0;          // (the count workspace was sized for the fused layout)
            }
            if (!b->inten) return NYXHIP_INTERNAL_NEEDS_CLOUDS;               // window-mode chunk: this kernel reads the clouds
            wa.px_offset = b->px_offset; wa.inten = b->inten; wa.min_inten = b->min_inten; wa.max_inten = b->max_inten;
            wa.slide_min = b->slide_min; wa.slide_max = b->slide_max;
            wa.out = a.out; wa.ld = a.ld; wa.status = a.status;
            wa.col_intensity = a.col_intensity; wa.n_hist = a.n_hist;
            wa.list = list; wa.n_list = grid;
            if (int r1 = launch_roi_wide(wa, st)) return r1;
            done = true;
            ag.sp.max_range = 0xFFFFu;
            if (with_glcm)
                if (int r2 = launch_roi_features(ag, st, grid)) return r2;
            if (E.range > 0xFFFFu) {                                          // members beyond 16 bits: the fused kernel, as if they were alone
                a.sp.min_range = 0x10000u;
                return launch_features_main();
            }
            return 0;
        };
        if (lds & 1) {
            bool wide_done = false;
            rc = launch_features_wide(wide_done);
            if (rc == NYXHIP_INTERNAL_NEEDS_CLOUDS) return rc;
            if (rc == 0 && !wide_done) rc = launch_features_main();
        }
        if (rc == 0 && (lds & 2)) rc = launch_roi_texture(t, st, grid);
        if (rc == 0 && (lds & 8)) rc = launch_roi_dependence(d, st, grid);
        if (rc == 0 && (lds & 4)) {
            // Size class 2 and Gabor: boxes of 65..128 px hold 87 KB of LDS in the tiled kernel -- one workgroup per CU, ~11 ms for the
            // 2 031 such ROIs of the heavy-tailed batch, with three quarters of every CU's wave slots idle.  On a lane of its own the
            // smaller classes' launches (and this class's other families) run beside it instead of behind it.
            static const bool no_gabor_lane = [] { const char* e = getenv("NYXHIP_NO_GABOR_LANE"); return e && *e && *e != '0'; }();   // A/B knob
            hipStream_t gst = st;
            if (!no_gabor_lane && list && cls / 2 == 2 && (mask & NYXHIP_FAM_GABOR) && !on_lds_lane && !ctx->win_next.inten)
                if (int lrc = use_lane(ctx, nyxhip_ctx::kGaborLane, &gst)) return lrc;
            rc = launch_roi_shape(g, gst, grid);
        }
        if (rc != 0)
            return fail(ctx, NYXHIP_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
    }
    if (!gs) {
        if (on_lds_lane && report && ctx->timing >= 2) {
            if (!report->e2) HIP_TRY(ctx, hipEventCreate(&report->e2));      // (the size-class-2 texture branch may have made it already: that one is on its lane's stream)
            else return NYXHIP_OK;
            HIP_TRY(ctx, hipEventRecord(report->e2, st));
        }
        return NYXHIP_OK;
    }

    // ---- INTENSITY + GLCM of the class by several workgroups per ROI (every member whose intensity range the histogram holds) ------
    static const bool no_coop = [] { const char* e = getenv("NYXHIP_NO_COOP"); return e && *e && *e != '0'; }();   // A/B knob: the one-workgroup path
    bool coop = false;
    if ((gs & 1) && tot && !no_coop) {
        if (int lrc = run_large(ctx, b, full, s, d_out, ld, E, *tot, list, grid, &coop)) return lrc;
        if (coop) {
            if (report) report->cooperative = 1;
            if (E.range < kLargeRangeMax) gs &= ~1u;     // nobody left for the sort path below
            if (!gs) return NYXHIP_OK;
        }
    }

    // the lane of this class (nyxhip_ctx::lane_stream): large classes only -- the workspace fallback of an LDS class stays on the main stream
    static const bool no_lanes = [] { const char* e = getenv("NYXHIP_NO_LANES"); return e && *e && *e != '0'; }();   // A/B knob
    const int lane = (!no_lanes && cls / 2 >= kFirstLargeSizeClass) ? (cls - 2 * kFirstLargeSizeClass) % 4 : -1;
    if (lane >= 0)
        if (int lrc2 = enter_lane(lane)) return lrc2;
    auto lane_stamp = [&]() -> int {
        if (lane >= 0 && report && ctx->timing >= 2) {
            if (!report->e2) HIP_TRY(ctx, hipEventCreate(&report->e2));
            HIP_TRY(ctx, hipEventRecord(report->e2, st));
        }
        return NYXHIP_OK;
    };

    // ---- GLRLM + GLSZM + NGTDM of the class by several workgroups per ROI (roi_large_tex.hip) ---------------------------------------
    static const bool no_coop_tex = [] { const char* e = getenv("NYXHIP_NO_COOP_TEX"); return e && *e && *e != '0'; }();   // A/B knob
    int tex_served = 0;
    if ((gs & 2) && tot && !no_coop && !no_coop_tex && cls / 2 >= kFirstLargeSizeClass) {
        if (int lrc = run_large_tex(ctx, b, full, s, d_out, ld, E, *tot, list, grid, st, lane >= 0 ? lane : nyxhip_ctx::kLanes, &tex_served)) return lrc;
        if (tex_served && report) report->cooperative |= 2;
        if (tex_served == 1) gs &= ~2u;
        if (!gs) return lane_stamp();
    }

    // ---- one workgroup per ROI over a global workspace: the groups of `gs` ------------------------------------------------------
    RoiArgs a2; TexArgs t2; ShapeArgs g2; DepArgs d2;
    int lrc = build_args(ctx, b, full, s, d_out, ld, E, (size_t)1 << 31, a2, t2, g2, d2, why, gs);
    if (lrc) return fail(ctx, lrc, "large-ROI workspace: " + why);
    // ---- Gabor of the class by several workgroups per ROI (roi_large_gabor.hip) --------------------------------------------------------
    bool gabor_served = false;
    if ((gs & 4) && (mask & NYXHIP_FAM_GABOR) && !no_coop) {
        if (int grc = run_large_gabor(ctx, b, s, d_out, ld, E, g2, list, grid, st, lane >= 0 ? &ctx->lane_buf[lane] : &ctx->d_spill,
                                      lane >= 0 ? &ctx->lane_bytes[lane] : &ctx->spill_bytes, &gabor_served)) return grc;
        if (gabor_served && report) report->cooperative |= 4;
    }
    if (coop) a2.sp.min_range = kLargeRangeMax;          // the histogram path served everybody below
    if (tex_served == 2) t2.sp.skip_ltex = 1;            // ... and the strip path every box it takes
    size_t stride = 0;
    if (gs & 1) stride = std::max<size_t>(stride, a2.L.total);
    if (gs & 2) stride = std::max<size_t>(stride, t2.L.total);
    if ((gs & 4) && (mask & NYXHIP_FAM_GABOR) && !gabor_served) stride = std::max<size_t>(stride, g2.L.total);
    if (gs & 8) stride = std::max<size_t>(stride, d2.L.total);
    stride = (stride + 255) & ~(size_t)255;
    const size_t budget = (size_t)4 << 30;         // at most 4 GiB of scratch in flight
    const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(grid, budget / std::max<size_t>(stride, 1)));
    const size_t need = stride * chunk;
    unsigned char** const bufp = lane >= 0 ? &ctx->lane_buf[lane] : &ctx->d_spill;
    size_t* const bytesp = lane >= 0 ? &ctx->lane_bytes[lane] : &ctx->spill_bytes;
    if (need > *bytesp) {
        if (*bufp) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(*bufp)); *bufp = nullptr; *bytesp = 0; }
        HIP_TRY(ctx, hipMalloc((void**)bufp, need));
        *bytesp = need;
    }
    if ((gs & 4) && (mask & NYXHIP_FAM_ZERNIKE)) {   // Zernike keeps no ROI-sized state in LDS: one launch over the class, whatever its size
        ShapeArgs gz = g2;
        gz.mask = NYXHIP_FAM_ZERNIKE; gz.sp.scratch = nullptr; gz.small_rois = 0;
        set_slots(gz.sp, list, grid);
        rc = launch_roi_shape(gz, st, grid);
        if (rc != 0) return fail(ctx, NYXHIP_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
    }
    // The dependence trio of a large class (one workgroup per ROI over the workspace: a few hundred workgroups, ~11 ms for the heavy-tailed
    // batch) on a lane and a scratch buffer of its own: it runs beside the class's other chains instead of behind them.
    static const bool no_dep_lane = [] { const char* e = getenv("NYXHIP_NO_DEP_LANE"); return e && *e && *e != '0'; }();   // A/B knob
    if ((gs & 8) && lane >= 0 && !no_dep_lane && gs != 8u) {
        hipStream_t dst = st;
        if (int lrc2 = use_lane(ctx, nyxhip_ctx::kDepLane, &dst)) return lrc2;
        const size_t dstride = (d2.L.total + 255) & ~(size_t)255;
        const uint32_t dchunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(grid, budget / std::max<size_t>(dstride, 1)));
        unsigned char** const dbuf = &ctx->lane_buf[nyxhip_ctx::kDepLane];
        size_t* const dbytes = &ctx->lane_bytes[nyxhip_ctx::kDepLane];
        if (dstride * dchunk > *dbytes) {
            if (*dbuf) { HIP_TRY(ctx, hipStreamSynchronize(dst)); HIP_TRY(ctx, hipFree(*dbuf)); *dbuf = nullptr; *dbytes = 0; }
            HIP_TRY(ctx, hipMalloc((void**)dbuf, dstride * dchunk));
            *dbytes = dstride * dchunk;
        }
        for (uint32_t o = 0; o < grid; o += dchunk) {
            const uint32_t nb = std::min(dchunk, grid - o);
            set_slots(d2.sp, list + o, nb);
            d2.sp.scratch = *dbuf; d2.sp.stride = dstride;
            if (int drc = launch_roi_dependence(d2, dst, nb))
                return fail(ctx, NYXHIP_ERR_HIP, std::string("large-ROI kernel launch failed: ") + hipGetErrorString((hipError_t)drc));
        }
        gs &= ~8u;
    }
    for (uint32_t o = 0; o < grid; o += chunk) {
        const uint32_t nb = std::min(chunk, grid - o);
        set_slots(a2.sp, list + o, nb); set_slots(t2.sp, list + o, nb); set_slots(g2.sp, list + o, nb); set_slots(d2.sp, list + o, nb);
        a2.sp.scratch = t2.sp.scratch = g2.sp.scratch = d2.sp.scratch = *bufp;
        a2.sp.stride = t2.sp.stride = g2.sp.stride = d2.sp.stride = stride;
        rc = (gs & 1) ? launch_roi_features(a2, st, nb) : 0;
        if (rc == 0 && (gs & 2)) rc = launch_roi_texture(t2, st, nb);
        if (rc == 0 && (gs & 8)) rc = launch_roi_dependence(d2, st, nb);
        if (rc == 0 && (gs & 4) && (mask & NYXHIP_FAM_GABOR) && !gabor_served) rc = launch_roi_shape(g2, st, nb);
        if (rc != 0)
            return fail(ctx, NYXHIP_ERR_HIP, std::string("large-ROI kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
    }
    return lane_stamp();
}

static void clear_runs(nyxhip_ctx* ctx)
{
    for (ClassRun& r : ctx->runs) {
        if (r.e0) (void)hipEventDestroy(r.e0);
        if (r.e1) (void)hipEventDestroy(r.e1);
        if (r.e2) (void)hipEventDestroy(r.e2);
    }
    ctx->runs.clear();
}

// Launch on device-resident arrays.
//   hinted: the extrema are the caller's statement about the batch (or exact, computed by the caller of this function).  When
//   they rule out everything but the two smallest size classes, nothing has to be counted: the kernel builds of those classes
//   differ only in whether the 16-bit tables apply (feature kernels) and in the one-wave shape kernels of the smallest class, so
//   the call enqueues whole-batch launches -- filtered by class where the build follows the class -- and returns without a
//   host round trip (the metric configuration: a stream of back-to-back calls stays back to back).  Otherwise the classifier
//   runs, its class headers come to the host once (two small kernels + one 320-byte copy), and every class gets an exact
//   grid and a carve-out of its own extrema.  Either way the kernel build an ROI runs through follows from ITS class and the
//   settings, not from its companions.
int launch_device_all(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s, double* d_out,
                      size_t ld, uint32_t max_px, uint32_t max_area, uint32_t max_range, uint32_t max_side, bool hinted)
{
    if (mask & NYXHIP_FAM_GABOR)
        if (int brc = ensure_gabor_bank(ctx, s))
            return brc;
    hipStream_t st = ctx->stream();
    const uint32_t n_roi = (uint32_t)b->n_roi;
    clear_runs(ctx);
    // workspace lanes (run_class): forked from here, joined into the main stream on every way out
    if (!ctx->lane_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->lane_fork, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->lane_fork, st));
    struct LaneJoin {
        nyxhip_ctx* c; hipStream_t st;
        ~LaneJoin() {
            for (int k = 0; k < nyxhip_ctx::kLanes; k++)
                if (c->lane_used[k]) {
                    (void)hipEventRecord(c->lane_done[k], c->lane_stream[k]);
                    (void)hipStreamWaitEvent(st, c->lane_done[k], 0);
                    c->lane_used[k] = false;
                }
        }
    } lane_join{ctx, st};
    if (mask & NYXHIP_FAM_GLCM) {
        // matrix orders of the split GLCM launches (RoiArgs::glcm_ng).  A table of its own: the count workspace may be re-allocated
        // between the launch groups of a call.
        const size_t need = 4ull * n_roi + 256;
        if (need > ctx->glcm_ng_bytes) {
            if (ctx->d_glcm_ng) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(ctx->d_glcm_ng)); ctx->d_glcm_ng = nullptr; ctx->glcm_ng_bytes = 0; }
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_glcm_ng, need + need / 4));
            ctx->glcm_ng_bytes = need + need / 4;
        }
        // 0 = "nothing to derive": an ROI whose feature kernel returns before it states its matrix order (error paths) must not leave
        // glcm_features_kernel a stale order from an earlier call
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_glcm_ng, 0, 4ull * n_roi, st));
    }
    auto timed_class = [&](int cls, uint32_t count, const Extrema& E, const uint32_t* lp, uint32_t grid, uint32_t class_mask = 0, uint32_t group_sel = 0xF,
                           const ClassTotals* tot = nullptr) -> int {
        ClassRun r{cls, count, E, 0, nullptr, nullptr};
        if (ctx->timing >= 2) {
            HIP_TRY(ctx, hipEventCreate(&r.e0));
            HIP_TRY(ctx, hipEventCreate(&r.e1));
            HIP_TRY(ctx, hipEventRecord(r.e0, st));
        }
        const int rc = run_class(ctx, b, mask, s, d_out, ld, E, lp, grid, false, nullptr, &r, class_mask, group_sel, cls, tot);
        if (ctx->timing >= 2 && rc == 0) HIP_TRY(ctx, hipEventRecord(r.e1, st));
        ctx->runs.push_back(r);
        return rc;
    };
    if ((mask & ~kMoments) || !hinted) {               // (a batch without stated extrema gets them from the class headers)
        bool done = false;
        static const bool force_exact = [] { const char* e = getenv("NYXHIP_CLASS_SYNC"); return e && *e && *e != '0'; }();   // A/B knob
        // (IBSI co-occurrence matrices are as large as the largest intensity, which a statement about the batch does not carry)
        const bool need_vmax = s->ibsi && (mask & (NYXHIP_FAM_GLCM | kTexture | kDependence));
        // (a stated range that allows wide-range ROIs takes the exact path as well: a whole-batch launch per table width would
        //  carve 43 KB for a class that 16-bit data leaves empty -- 100 k workgroups that only return, three per CU)
        const bool wide_possible = max_range >= 16384u && (mask & NYXHIP_FAM_INTENSITY);
        const bool has_m1 = !(max_px <= kClassPx[0] && max_side <= kClassSide[0]);
        static const bool no_small = [] { const char* e = getenv("NYXHIP_NO_SMALL"); return e && *e && *e != '0'; }();   // A/B knob
        static const bool no_adapt = [] { const char* e = getenv("NYXHIP_NO_ADAPT"); return e && *e && *e != '0'; }();   // A/B knob
        const bool small_fits = !no_small && ((mask & NYXHIP_FAM_INTENSITY) || (!s->ibsi && s->grey_depth > 0 && s->grey_depth <= 64));
        const bool two_filtered = (mask & (NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM)) && has_m1 && small_fits;
        // (the recent calls held the smallest class in numbers: exact lists beat two launches that each skip the other's slots)
        const bool prefer_lists = two_filtered && !no_adapt && ctx->census_total != 0 && 4 * ctx->census_small >= ctx->census_total;
        if (hinted && !force_exact && !need_vmax && !wide_possible && !prefer_lists && max_px <= kClassPx[1] && max_side <= kClassSide[1]) {
            // ---- whole-batch launches, nothing counted -----------------------------------------------------------------------
            if (two_filtered) ctx->census_pending += n_roi;
            const Extrema Eall{max_px, max_area, max_range, max_side};
            struct Group { int cls; Extrema E; uint32_t class_mask, group_sel; };
            std::vector<Group> groups;
            // texture / dependence kernels: one build for both classes
            if (mask & (kTexture | kDependence)) groups.push_back({-1, Eall, 0u, 2u | 8u});
            // feature kernels: one launch, no filter (both size classes run the same build; a contradicting ROI raises the error flag
            // in the kernel)
            // ... except that the smallest size class has a kernel of its own (a wave per ROI, roi_small.hip) for INTENSITY and for
            // GLCM under matlab binning with <= 64 levels: then two launches, each filtered to its classes
            if (two_filtered) {
                const uint32_t sd0 = std::min(max_side, kClassSide[0]);
                groups.push_back({-2, Extrema{std::min(max_px, kClassPx[0]), std::min(max_area, sd0 * sd0), max_range, sd0}, 0x1u, 1u});
                groups.push_back({-3, Eall, 0x3FEu, 1u});
            } else
            if (mask & (NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM)) groups.push_back({-2, Eall, 0u, 1u});
            // shape kernels: one-wave builds for the smallest size class, four-wave builds for the other
            if (mask & kShape) {
                const uint32_t sd0 = std::min(max_side, kClassSide[0]);
                if (!has_m1) groups.push_back({-4, Eall, 0u, 4u});
                else {
                    groups.push_back({-4, Extrema{std::min(max_px, kClassPx[0]), std::min(max_area, sd0 * sd0), max_range, sd0}, 0x3u, 4u});
                    groups.push_back({-5, Eall, 0x3FCu, 4u});   // (every other class: an ROI beyond the stated extrema meets the kernel's cap check and raises the error flag)
                }
            }
            bool needs_host = false;
            for (const Group& g : groups)
                if (int rc = run_class(ctx, b, mask, s, d_out, ld, g.E, nullptr, n_roi, true, &needs_host, nullptr, g.class_mask, g.group_sel)) return rc;
            if (!needs_host) {                             // (else: the exact path, whose workspace chunks need member counts)
                for (const Group& g : groups)
                    if (int rc = timed_class(g.cls, n_roi, g.E, nullptr, n_roi, g.class_mask, g.group_sel)) return rc;
                done = true;
            }
        }
        if (!done) {
            // ---- classify, class headers to the host, one exact launch group per class ------------------------------------------
            const size_t list_bytes = 4ull * n_roi + 256;
            if (list_bytes > ctx->cls_list_bytes) {
                if (ctx->d_cls_list) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(ctx->d_cls_list)); ctx->d_cls_list = nullptr; ctx->cls_list_bytes = 0; }
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_cls_list, list_bytes + list_bytes / 4));
                ctx->cls_list_bytes = list_bytes + list_bytes / 4;
            }
            if (!ctx->d_cls_hdr) {
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_cls_hdr, sizeof(uint32_t) * kClasses * H_WORDS));
                HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_cls_hdr, sizeof(uint32_t) * kClasses * H_WORDS, hipHostMallocDefault));
            }
            uint32_t* const hdr = ctx->d_cls_hdr;
            uint32_t* const list = ctx->d_cls_list;
            HIP_TRY(ctx, hipMemsetAsync(hdr, 0, sizeof(uint32_t) * kClasses * H_WORDS, st));
            const unsigned blocks = (unsigned)((b->n_roi + 255) / 256);
            const uint32_t lvl_on = need_vmax ? 1u : 0u;
            hipLaunchKernelGGL(class_count_kernel, dim3(blocks), dim3(256), 0, st, b->n_roi, b->px_offset, b->bbox_w, b->bbox_h, b->min_inten, b->max_inten, hdr, lvl_on);
            hipLaunchKernelGGL(class_scatter_kernel, dim3(blocks), dim3(256), 0, st, b->n_roi, b->px_offset, b->bbox_w, b->bbox_h, b->min_inten, b->max_inten, hdr, list, lvl_on);
            if (hipError_t e = hipGetLastError(); e != hipSuccess)
                return fail(ctx, NYXHIP_ERR_HIP, std::string("classifier launch failed: ") + hipGetErrorString(e));
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_cls_hdr, hdr, sizeof(uint32_t) * kClasses * H_WORDS, hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
            const uint32_t* H = ctx->h_cls_hdr;
            if (!ctx->census_pending) { ctx->census_small = H[H_COUNT]; ctx->census_total = n_roi; }   // (class 0 = the smallest size class, 16-bit tables)
            if (!hinted) {
                max_px = max_area = max_range = max_side = 0;
                for (int cls = 0; cls < kClasses; cls++) {
                    const uint32_t* h = H + cls * H_WORDS;
                    max_px = std::max(max_px, h[H_PX]); max_area = std::max(max_area, h[H_AREA]);
                    max_range = std::max(max_range, h[H_RANGE]); max_side = std::max(max_side, h[H_SIDE]);
                }
            }
            // The size classes beyond LDS (3 and 4, either table width) go through the same several-workgroups-per-ROI kernels, which cut
            // every ROI by its own box: they run as ONE launch group (their lists are neighbours in the class order) -- four groups of
            // a handful of ROIs each cost four times the ten-odd launches and workspace clears of a group (~0.1 ms apiece on the mixed
            // batch).  Nothing an ROI's row depends on changes: the slab / strip cut is invisible by construction.
            // A window-mode chunk (no clouds materialised) with a class that reads clouds -- the classes beyond LDS, and the wide-range
            // classes the bitmap kernel serves -- goes back for them BEFORE anything is launched: raised from inside the class loop it
            // made the caller run the whole chunk again, every LDS class computed twice (16-bit tiles: on every chunk).
            if (ctx->win_next.inten && !b->inten && (mask & ~kMoments)) {
                static const bool no_wide_pre = [] { const char* e = getenv("NYXHIP_NO_WIDE"); return e && *e && *e != '0'; }();
                for (int cls = 0; cls < kClasses; cls++) {
                    if (H[cls * H_WORDS + H_COUNT] == 0) continue;
                    if (cls / 2 >= kFirstLargeSizeClass || ((cls & 1) && (mask & NYXHIP_FAM_INTENSITY) && !no_wide_pre))
                        return NYXHIP_INTERNAL_NEEDS_CLOUDS;
                }
            }
            static const bool no_merge = [] { const char* e = getenv("NYXHIP_NO_MERGE_LARGE"); return e && *e && *e != '0'; }();   // A/B knob
            int first_cls = kClasses - 1;
            if (!no_merge && (mask & ~kMoments)) {
                uint32_t cnt = 0; int top = -1;
                Extrema Em{0, 0, 0, 0, 0, false};
                ClassTotals tm{0, 0, 0};
                for (int cls = 2 * kFirstLargeSizeClass; cls < kClasses; cls++) {
                    const uint32_t* h = H + cls * H_WORDS;
                    if (h[H_COUNT] == 0) continue;
                    cnt += h[H_COUNT]; top = cls;
                    Em.px = std::max(Em.px, h[H_PX]); Em.area = std::max(Em.area, h[H_AREA]); Em.range = std::max(Em.range, h[H_RANGE]);
                    Em.side = std::max(Em.side, h[H_SIDE]); Em.vmax = std::max(Em.vmax, h[H_VMAX]);
                    tm.px += ((uint64_t)h[H_SUMPX_HI] << 32) | h[H_SUMPX]; tm.area += ((uint64_t)h[H_SUMAREA_HI] << 32) | h[H_SUMAREA];
                    tm.range1 += ((uint64_t)h[H_SUMRANGE_HI] << 32) | h[H_SUMRANGE];
                }
                if (top >= 0)
                    if (int rc = timed_class(top, cnt, Em, list + H[2 * kFirstLargeSizeClass * H_WORDS + H_OFFSET], cnt, 0, 0xF, &tm))
                        return rc;
                first_cls = 2 * kFirstLargeSizeClass - 1;
            }
            for (int cls = first_cls; cls >= 0 && (mask & ~kMoments); cls--) {   // largest ROIs first: their long workgroups start early
                const uint32_t* h = H + cls * H_WORDS;
                if (h[H_COUNT] == 0) continue;
                const Extrema E{h[H_PX], h[H_AREA], h[H_RANGE], h[H_SIDE], h[H_VMAX], (cls & 1) != 0 && cls / 2 < kSizeClasses - 1};
                const ClassTotals tot{((uint64_t)h[H_SUMPX_HI] << 32) | h[H_SUMPX], ((uint64_t)h[H_SUMAREA_HI] << 32) | h[H_SUMAREA],
                                      ((uint64_t)h[H_SUMRANGE_HI] << 32) | h[H_SUMRANGE]};
                if (int rc = timed_class(cls, h[H_COUNT], E, list + h[H_OFFSET], h[H_COUNT], 0, 0xF, &tot))
                    return rc;
            }
        }
    }
    if (mask & kMoments)
        if (int mrc = launch_moments(ctx, b, mask, s, d_out, ld, max_px, max_area, max_side))
            return mrc;
    return NYXHIP_OK;
}

// launch_device_all between two events on the launch stream: the timing hooks of include/nyxhip.h cover EVERY kernel the call
// enqueues (the LDS launch groups, the moments pair, the global-workspace and large-ROI passes), on every return path.
int launch_device(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s, double* d_out,
                  size_t ld, uint32_t max_px, uint32_t max_area, uint32_t max_range, uint32_t max_side, bool hinted = true)
{
    if (!ctx->timing)
        return launch_device_all(ctx, b, mask, s, d_out, ld, max_px, max_area, max_range, max_side, hinted);
    hipStream_t st = ctx->stream();
    if (ctx->ev_used == ctx->ev.size()) {
        hipEvent_t x, y;
        HIP_TRY(ctx, hipEventCreate(&x));
        HIP_TRY(ctx, hipEventCreate(&y));
        ctx->ev.push_back({x, y});
    }
    hipEvent_t e0 = ctx->ev[ctx->ev_used].first, e1 = ctx->ev[ctx->ev_used].second;
    HIP_TRY(ctx, hipEventRecord(e0, st));
    const int rc = launch_device_all(ctx, b, mask, s, d_out, ld, max_px, max_area, max_range, max_side, hinted);
    HIP_TRY(ctx, hipEventRecord(e1, st));
    ctx->ev_used++;
    return rc;
}

int validate(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s, double* out, size_t ld)
{
    if (!ctx) return NYXHIP_ERR_INVALID_ARG;
    if (!b || !s || !out) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "null batch / settings / out_table");
    if (mask == 0 || (mask & ~NYXHIP_FAM_ALL)) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "bad family mask");
    if (mask & ~kImplemented)
        return fail(ctx, NYXHIP_ERR_UNSUPPORTED, "requested feature family is not implemented by the HIP path yet "
                    "(all seven hot-path families are implemented; bad mask?)");
    std::string why;
    if (!settings_ok(s, mask, why)) return fail(ctx, NYXHIP_ERR_INVALID_ARG, why);
    if (b->n_roi && (!b->px_offset || !b->x || !b->y || !b->inten || !b->bbox_w || !b->bbox_h || !b->min_inten || !b->max_inten))
        return fail(ctx, NYXHIP_ERR_INVALID_ARG, "batch has null array pointers");
    if ((b->slide_min == nullptr) != (b->slide_max == nullptr))
        return fail(ctx, NYXHIP_ERR_INVALID_ARG, "slide_min and slide_max must both be given or both NULL");
    if ((int)ld < nyxhip_n_columns(mask, s)) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "out_ld smaller than the column count");
    if (b->n_roi > 0x7FFFFFFFull) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "too many ROIs in one batch");
    return NYXHIP_OK;
}

} // namespace

extern "C" {

int nyxhip_abi_version(void) { return NYXHIP_ABI_VERSION; }

void nyxhip_default_settings(nyxhip_settings* s)
{
    if (!s) return;
    memset(s, 0, sizeof(*s));
    s->soft_nan = 0.0;                 // cli_result_options.h:75
    s->tiny = 1e-10;
    s->grey_depth = 64;                // environment: coarse gray depth default
    s->ibsi = 0;
    s->glcm_grey_depth = 64;
    s->glcm_offset = 1;                // env_features.cpp:727
    s->glcm_n_angles = 4;              // glcm.cpp:9
    s->glcm_angles[0] = 0; s->glcm_angles[1] = 45; s->glcm_angles[2] = 90; s->glcm_angles[3] = 135;
    s->glcm_symmetric = 0;             // glcm.cpp:8
    s->gabor_gamma = 0.1; s->gabor_sig2lam = 0.8; s->gabor_kersize = 16; s->gabor_f0lp = 0.1; s->gabor_graythr = 0.025;
    s->gabor_n_filters = 4;            // gabor.cpp:19-25, consumed as (first = f0, second = theta) at :107-110
    const double pi4 = 0.78539816339744830962;
    const double f0[4] = {0.0, pi4, 2 * pi4, pi4 * 3.0}, th[4] = {4.0, 16.0, 32.0, 64.0};
    for (int i = 0; i < 4; i++) { s->gabor_f0[i] = f0[i]; s->gabor_theta[i] = th[i]; }
}

int nyxhip_init(int device, nyxhip_ctx** out_ctx)
{
    if (!out_ctx) return NYXHIP_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, NYXHIP_ERR_NO_DEVICE, "no HIP device available: the MI355X path has no CPU fallback");
    if (device < 0 || device >= n)
        return fail(nullptr, NYXHIP_ERR_INVALID_ARG, "device index out of range");
    nyxhip_ctx* ctx = new nyxhip_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&ctx->d_status, 4 * sizeof(int)) != hipSuccess || hipMemset(ctx->d_status, 0, 4 * sizeof(int)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, NYXHIP_ERR_HIP, "failed to create the device context");
    }
    if (getenv("NYXHIP_STAMPS")) { // diagnostic builds only; never set in production
        if (hipMalloc((void**)&ctx->d_stamps, 32 * sizeof(unsigned long long)) == hipSuccess)
            (void)hipMemset(ctx->d_stamps, 0, 32 * sizeof(unsigned long long));
    }
    g_ctx_on_device[device & 63].fetch_add(1);
    *out_ctx = ctx;
    return NYXHIP_OK;
}

void nyxhip_destroy(nyxhip_ctx* ctx)
{
    if (!ctx) return;
    g_ctx_on_device[ctx->device & 63].fetch_sub(1);
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->own_stream); (void)hipStreamDestroy(ctx->own_stream); }
    for (auto& p : ctx->ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    if (ctx->d_tile) (void)hipFree(ctx->d_tile);
    if (ctx->d_cloud) (void)hipFree(ctx->d_cloud);
    if (ctx->d_res) (void)hipFree(ctx->d_res);
    for (int k = 0; k < 2; k++) {
        if (ctx->d_slot[k]) (void)hipFree(ctx->d_slot[k]);
        if (ctx->slot_ready[k]) (void)hipEventDestroy(ctx->slot_ready[k]);
        if (ctx->slot_free[k]) (void)hipEventDestroy(ctx->slot_free[k]);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (int k = 0; k < nyxhip_ctx::kStageSlots; k++) {
        if (ctx->h_stage_done[k]) (void)hipEventDestroy(ctx->h_stage_done[k]);
        if (ctx->h_stage[k]) (void)hipHostFree(ctx->h_stage[k]);
    }
    if (ctx->d_spill) (void)hipFree(ctx->d_spill);
    if (ctx->d_mom) (void)hipFree(ctx->d_mom);
    if (ctx->d_glcm_ws) (void)hipFree(ctx->d_glcm_ws);
    if (ctx->d_glcm_ng) (void)hipFree(ctx->d_glcm_ng);
    if (ctx->d_logtab) (void)hipFree(ctx->d_logtab);
    if (ctx->d_spill_list) (void)hipFree(ctx->d_spill_list);
    for (int k = 0; k < nyxhip_ctx::kLanes; k++) {
        if (ctx->lane_stream[k]) { (void)hipStreamSynchronize(ctx->lane_stream[k]); (void)hipStreamDestroy(ctx->lane_stream[k]); }
        if (ctx->lane_done[k]) (void)hipEventDestroy(ctx->lane_done[k]);
        if (ctx->lane_buf[k]) (void)hipFree(ctx->lane_buf[k]);
    }
    if (ctx->lane_fork) (void)hipEventDestroy(ctx->lane_fork);
    for (int k = 0; k <= nyxhip_ctx::kLanes; k++) {
        if (ctx->ltex_buf[k]) (void)hipFree(ctx->ltex_buf[k]);
        if (ctx->ltex_aux[k]) (void)hipFree(ctx->ltex_aux[k]);
    }
    if (ctx->d_large) (void)hipFree(ctx->d_large);
    if (ctx->d_large_aux) (void)hipFree(ctx->d_large_aux);
    clear_runs(ctx);
    if (ctx->d_cls_list) (void)hipFree(ctx->d_cls_list);
    if (ctx->d_cls_hdr) (void)hipFree(ctx->d_cls_hdr);
    if (ctx->h_cls_hdr) (void)hipHostFree(ctx->h_cls_hdr);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_bank) (void)hipFree(ctx->d_bank);
    if (ctx->d_bank32) (void)hipFree(ctx->d_bank32);
    if (ctx->d_bank16) (void)hipFree(ctx->d_bank16);
    if (ctx->d_stamps) {
        unsigned long long h[32];
        if (hipMemcpy(h, ctx->d_stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long tot = 0;
            for (int i = 0; i < 32; i++) tot += h[i];
            for (int i = 0; i < 32; i++)
                if (h[i]) fprintf(stderr, "[nyxhip stamp] phase %2d: %14llu cycles  %5.1f %%\n", i, h[i], 100.0 * (double)h[i] / (double)tot);
        }
        (void)hipFree(ctx->d_stamps);
    }
    delete ctx;
}

const char* nyxhip_last_error(const nyxhip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int nyxhip_set_stream(nyxhip_ctx* ctx, void* hip_stream)
{
    if (!ctx) return NYXHIP_ERR_INVALID_ARG;
    ctx->user_stream = (hipStream_t)hip_stream;
    ctx->use_user_stream = true; // NULL is the legacy default stream, a valid choice
    return NYXHIP_OK;
}

int nyxhip_n_columns(uint32_t family_mask, const nyxhip_settings* s)
{
    if (!s) return 0;
    int n = 0;
    if (family_mask & NYXHIP_FAM_INTENSITY) n += kIntensityCols;
    if (family_mask & NYXHIP_FAM_GLCM) n += kGlcmAngled * s->glcm_n_angles + kGlcmAve;
    if (family_mask & NYXHIP_FAM_GLRLM) n += kGlrlmCols;
    if (family_mask & NYXHIP_FAM_GLDZM) n += kGldzmCols;
    if (family_mask & NYXHIP_FAM_GLSZM) n += kGlszmCols;
    if (family_mask & NYXHIP_FAM_GLDM) n += kGldmCols;
    if (family_mask & NYXHIP_FAM_NGLDM) n += kNgldmCols;
    if (family_mask & NYXHIP_FAM_NGTDM) n += kNgtdmCols;
    if (family_mask & NYXHIP_FAM_GABOR) n += s->gabor_n_filters;
    if (family_mask & NYXHIP_FAM_ZERNIKE) n += kZernikeCols;
    if (family_mask & NYXHIP_FAM_SMOMS) n += kMomCols;
    if (family_mask & NYXHIP_FAM_IMOMS) n += kMomCols;
    return n;
}

int nyxhip_column_name(uint32_t family_mask, const nyxhip_settings* s, int col, char* buf, size_t buf_len)
{
    if (!s || !buf || buf_len == 0) return NYXHIP_ERR_INVALID_ARG;
    auto v = column_names(family_mask & kImplemented, s);
    if (col < 0 || col >= (int)v.size()) return NYXHIP_ERR_INVALID_ARG;
    snprintf(buf, buf_len, "%s", v[col].c_str());
    return NYXHIP_OK;
}

int nyxhip_sync(nyxhip_ctx* ctx)
{
    if (!ctx) return NYXHIP_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream()));
    return check_status(ctx);
}

int nyxhip_featurize_batch_async(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s,
                                 double* out, size_t ld)
{
    int rc = validate(ctx, b, mask, s, out, ld);
    if (rc) return rc;
    if (b->memory != NYXHIP_MEM_DEVICE)
        return fail(ctx, NYXHIP_ERR_INVALID_ARG, "the async form takes device-resident batches only");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (b->n_roi == 0) return NYXHIP_OK;
    // batch extrema: the caller's statement when given (all of max_px / max_bbox_area / max_bbox_side non-zero), else the size
    // classifier of launch_device_all derives them on the device
    const bool hinted = b->max_px != 0 && b->max_bbox_area != 0 && b->max_bbox_side != 0;
    return launch_device(ctx, b, mask, s, out, ld, b->max_px, b->max_bbox_area, b->max_inten_range, b->max_bbox_side, hinted);
}

int nyxhip_featurize_batch(nyxhip_ctx* ctx, const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s,
                           double* out, size_t ld)
{
    int rc = validate(ctx, b, mask, s, out, ld);
    if (rc) return rc;
    if (b->memory == NYXHIP_MEM_DEVICE) {
        rc = nyxhip_featurize_batch_async(ctx, b, mask, s, out, ld);
        if (rc) return rc;
        return nyxhip_sync(ctx);
    }
    if (b->memory != NYXHIP_MEM_HOST) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "bad batch->memory");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (b->n_roi == 0) return NYXHIP_OK;

    // host batch: derive extrema, stage SoA arrays into one device slab, run, copy the table back
    const uint64_t nr = b->n_roi, npx = b->px_offset[nr];
    uint32_t max_px = 0, max_area = 0, max_range = 0, max_side = 0;
    uint64_t n_small = 0;                                  // census of the smallest size class (launch_device_all picks lists or filtered launches by it)
    for (uint64_t r = 0; r < nr; r++) {
        if (b->px_offset[r + 1] < b->px_offset[r]) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "px_offset is not monotone");
        if (b->px_offset[r + 1] - b->px_offset[r] <= kClassPx[0] && b->bbox_w[r] <= kClassSide[0] && b->bbox_h[r] <= kClassSide[0]) n_small++;
        uint64_t n = b->px_offset[r + 1] - b->px_offset[r];
        uint64_t a = (uint64_t)b->bbox_w[r] * b->bbox_h[r];
        if (n > 0xFFFFFFFFull || a > 0xFFFFFFFFull) return fail(ctx, NYXHIP_ERR_ROI_TOO_LARGE, "ROI exceeds 2^32 pixels");
        if (n > max_px) max_px = (uint32_t)n;
        if (a > max_area) max_area = (uint32_t)a;
        if (b->max_inten[r] - b->min_inten[r] > max_range) max_range = b->max_inten[r] - b->min_inten[r];
        if (b->bbox_w[r] > max_side) max_side = b->bbox_w[r];
        if (b->bbox_h[r] > max_side) max_side = b->bbox_h[r];
    }
    const int n_cols = nyxhip_n_columns(mask, s);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o_off = 0, o_x = al(o_off + 8 * (nr + 1)), o_y = al(o_x + 2 * npx), o_i = al(o_y + 2 * npx),
           o_bw = al(o_i + 4 * npx), o_bh = al(o_bw + 4 * nr), o_mn = al(o_bh + 4 * nr), o_mx = al(o_mn + 4 * nr),
           o_smin = al(o_mx + 4 * nr), o_smax = al(o_smin + 8 * nr), o_out = al(o_smax + 8 * nr),
           total = al(o_out + 8ull * nr * n_cols);
    rc = ensure_stage(ctx, total);
    if (rc) return rc;
    char* base = (char*)ctx->d_stage;
    hipStream_t st = ctx->stream();
    HIP_TRY(ctx, hipMemcpyAsync(base + o_off, b->px_offset, 8 * (nr + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x, b->x, 2 * npx, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_y, b->y, 2 * npx, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_i, b->inten, 4 * npx, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_bw, b->bbox_w, 4 * nr, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_bh, b->bbox_h, 4 * nr, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_mn, b->min_inten, 4 * nr, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_mx, b->max_inten, 4 * nr, hipMemcpyHostToDevice, st));
    if (b->slide_min) {
        HIP_TRY(ctx, hipMemcpyAsync(base + o_smin, b->slide_min, 8 * nr, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(base + o_smax, b->slide_max, 8 * nr, hipMemcpyHostToDevice, st));
    }
    nyxhip_batch d = *b;
    d.memory = NYXHIP_MEM_DEVICE;
    d.px_offset = (const uint64_t*)(base + o_off);
    d.x = (const uint16_t*)(base + o_x); d.y = (const uint16_t*)(base + o_y); d.inten = (const uint32_t*)(base + o_i);
    d.bbox_w = (const uint32_t*)(base + o_bw); d.bbox_h = (const uint32_t*)(base + o_bh);
    d.min_inten = (const uint32_t*)(base + o_mn); d.max_inten = (const uint32_t*)(base + o_mx);
    d.slide_min = b->slide_min ? (const double*)(base + o_smin) : nullptr;
    d.slide_max = b->slide_max ? (const double*)(base + o_smax) : nullptr;
    double* d_out = (double*)(base + o_out);
    ctx->census_small = n_small; ctx->census_total = nr; ctx->census_pending = 0;
    rc = launch_device(ctx, &d, mask, s, d_out, (size_t)n_cols, max_px, max_area, max_range, max_side);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy2DAsync(out, ld * sizeof(double), d_out, (size_t)n_cols * sizeof(double),
                                  (size_t)n_cols * sizeof(double), nr, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return check_status(ctx);
}

void nyxhip_finalize_table(double* table, size_t n_rows, size_t n_cols, size_t ld, double soft_nan)
{
    if (!table) return;
    for (size_t r = 0; r < n_rows; r++)
        for (size_t c = 0; c < n_cols; c++) {
            double& v = table[r * ld + c];
            if (isnan(v) || isinf(v)) v = soft_nan; // force_finite_number, helpers/helpers.h:376-382
        }
}

// ---- fused tile path -----------------------------------------------------------------------------------------------------
static int grow(nyxhip_ctx* ctx, void** p, size_t* have, size_t need, hipStream_t st)
{
    if (need <= *have) return NYXHIP_OK;
    if (*p) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(*p)); *p = nullptr; *have = 0; }
    const size_t want = need + need / 8 + (1 << 16);
    HIP_TRY(ctx, hipMalloc(p, want));
    *have = want;
    return NYXHIP_OK;
}

static uint32_t log2u(uint32_t v) { uint32_t k = 0; while ((1u << k) < v) k++; return k; }

// Per-tile table size to start with: one slot per 1024 pixels (a 1024 x 1024 tile: 1024 slots for ~200 ROIs); a tile
// with more labels than slots makes the scan raise the overflow flag and the chunk is rescanned with four times the slots.
static uint32_t first_tile_cap(uint64_t tile_px)
{
    uint64_t c = tile_px / 1024;
    if (c < 256) c = 256;
    if (c > (1u << 22)) c = 1u << 22;
    return pow2ceil((uint32_t)c);
}

// Device workspace of one chunk besides the staging slots and the clouds (which are sized after the scan).
static size_t chunk_table_bytes(uint32_t nt, uint32_t cap)
{
    const size_t ent = (size_t)nt * cap;
    // per slot: 8 table words + 10 + 10 row words (unsorted / sorted rows) + three 8-byte arrays (CSR offsets, slide min / max) -- the
    // carve-out of tiles_chunk, kept in step with it; per 1024 slots: block sums; per tile: row / pixel starts and given slide extrema
    return ent * (28 * 4 + 3 * 8) + (ent / 1024 + 2) * 12 + (size_t)nt * (12 + 16) + 24 * 256 + (1 << 16);
}

// One chunk of tiles resident on the device -> rows in d_lab / d_til / d_out (device).  *n_roi_out rows are produced; more than
// rows_cap -> nothing is written beyond rows_cap and the caller reports the shortage.
static int tiles_chunk(nyxhip_ctx* ctx, const void* d_inten, int dtI, const void* d_label, int dtL, uint32_t W, uint32_t H, uint32_t nt,
                       int slide_mode, const double* h_smin, const double* h_smax, uint32_t family_mask, const nyxhip_settings* s,
                       uint64_t rows_cap, uint32_t* d_lab, uint32_t* d_til, uint32_t tile_base, double* d_out, size_t d_ld, uint32_t label_limit,
                       uint64_t* n_roi_out, hipStream_t st)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    uint32_t cap = ctx->tile_cap_hint ? ctx->tile_cap_hint : first_tile_cap((uint64_t)W * H);
    const uint64_t tile_px = (uint64_t)W * H;
    const uint32_t cap_max = pow2ceil((uint32_t)std::min<uint64_t>(2 * tile_px, 1u << 30));
    if (cap > cap_max) cap = cap_max;
    uint32_t meta[16];
    TileRows R;
    char* base = nullptr;
    for (;;) {
        const uint64_t ent = (uint64_t)nt * cap;
        if (ent > (1ull << 31)) return fail(ctx, NYXHIP_ERR_UNSUPPORTED, "too many ROIs per tile for one chunk: lower max_device_bytes so that fewer tiles share a chunk");
        const uint64_t rc_rows = ent;                                    // every ROI occupies a slot: this many rows always suffice
        size_t o = 0;
        size_t o_h[8]; for (int i = 0; i < 8; i++) { o_h[i] = o; o = al(o + 4 * ent); }
        size_t o_u[10]; for (int i = 0; i < 10; i++) { o_u[i] = o; o = al(o + 4 * (rc_rows + 1)); }
        size_t o_r[10]; for (int i = 0; i < 10; i++) { o_r[i] = o; o = al(o + 4 * (rc_rows + 1)); }
        const size_t o_ro = o; o = al(o + 8 * (rc_rows + 2));
        const size_t o_smin = o; o = al(o + 8 * (rc_rows + 1));
        const size_t o_smax = o; o = al(o + 8 * (rc_rows + 1));
        const size_t o_meta = o; o = al(o + 64);
        const size_t n_blk = (size_t)((ent + 1023) / 1024);
        const size_t o_br = o; o = al(o + 4 * n_blk);
        const size_t o_bp = o; o = al(o + 8 * n_blk);
        const size_t o_trb = o; o = al(o + 4 * ((size_t)nt + 1));
        const size_t o_tpb = o; o = al(o + 8 * ((size_t)nt + 1));
        const size_t o_sin = o; o = al(o + 16 * (size_t)nt);
        if (int grc = grow(ctx, &ctx->d_tile, &ctx->tile_bytes, o, st)) return grc;
        base = (char*)ctx->d_tile;
        TileHash T{(uint32_t*)(base + o_h[0]), (uint32_t*)(base + o_h[1]), (uint32_t*)(base + o_h[2]), (uint32_t*)(base + o_h[3]),
                   (uint32_t*)(base + o_h[4]), (uint32_t*)(base + o_h[5]), (uint32_t*)(base + o_h[6]), (uint32_t*)(base + o_h[7]), cap, 32u - log2u(cap)};
        TileRows U{(uint32_t*)(base + o_u[0]), (uint32_t*)(base + o_u[1]), (uint32_t*)(base + o_u[2]), nullptr, (uint32_t*)(base + o_u[3]),
                   (uint32_t*)(base + o_u[4]), (uint32_t*)(base + o_u[5]), (uint32_t*)(base + o_u[6]), (uint32_t*)(base + o_u[7]), (uint32_t*)(base + o_u[8]),
                   nullptr, nullptr};
        R = TileRows{(uint32_t*)(base + o_r[0]), (uint32_t*)(base + o_r[1]), (uint32_t*)(base + o_r[2]), (uint64_t*)(base + o_ro), (uint32_t*)(base + o_r[3]),
                     (uint32_t*)(base + o_r[4]), (uint32_t*)(base + o_r[5]), (uint32_t*)(base + o_r[6]), (uint32_t*)(base + o_r[7]), (uint32_t*)(base + o_r[8]),
                     (double*)(base + o_smin), (double*)(base + o_smax)};
        uint32_t* d_meta = (uint32_t*)(base + o_meta);
        HIP_TRY(ctx, hipMemsetAsync(d_meta, 0, 64, st));
        const double* d_smin = nullptr; const double* d_smax = nullptr;
        if (slide_mode == NYXHIP_SLIDE_GIVEN) {
            HIP_TRY(ctx, hipMemcpyAsync(base + o_sin, h_smin, 8 * (size_t)nt, hipMemcpyHostToDevice, st));
            HIP_TRY(ctx, hipMemcpyAsync(base + o_sin + 8 * (size_t)nt, h_smax, 8 * (size_t)nt, hipMemcpyHostToDevice, st));
            d_smin = (const double*)(base + o_sin); d_smax = d_smin + nt;
        }
        int rc = launch_tile_assembly_scan(d_inten, dtI, d_label, dtL, W, H, nt, T, U, R, (uint32_t)std::min<uint64_t>(rc_rows, 0xFFFFFFFFu), d_meta,
                                           (uint32_t*)(base + o_br), (unsigned long long*)(base + o_bp), (uint32_t*)(base + o_trb),
                                           (unsigned long long*)(base + o_tpb), st);
        if (rc == 0)
            rc = launch_tile_rank(U, (const uint32_t*)(base + o_trb), (const unsigned long long*)(base + o_tpb), R, (uint32_t)std::min<uint64_t>(rc_rows, 0xFFFFFFFFu),
                                  nt, cap, slide_mode, d_smin, d_smax, st);
        if (rc) return fail(ctx, NYXHIP_ERR_HIP, std::string("tile scan launch failed: ") + hipGetErrorString((hipError_t)rc));
        HIP_TRY(ctx, hipMemcpyAsync(meta, d_meta, sizeof(meta), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (meta[7] == 1) {                                  // a tile holds more labels than its table has slots
            if (cap >= cap_max) return fail(ctx, NYXHIP_ERR_HIP, "tile table overflow at the maximum table size");
            cap = cap * 4 > cap_max ? cap_max : cap * 4;
            continue;
        }
        break;
    }
    ctx->tile_cap_hint = cap;
    if (meta[7] == 2)
        return fail(ctx, NYXHIP_ERR_ROI_TOO_LARGE, "an ROI's bounding box is wider or taller than 65535 pixels (coordinates inside a box are 16-bit)");
    if (meta[8] > label_limit)
        return fail(ctx, NYXHIP_ERR_INVALID_ARG, "the label tile holds a value above max_label");
    const uint64_t n_roi = meta[0];
    *n_roi_out = n_roi;
    if (n_roi == 0 || n_roi > rows_cap) return NYXHIP_OK;
    const uint64_t npx = ((uint64_t)meta[2] << 32) | meta[1];
    // INTENSITY / GLCM alone, every ROI LDS-sized: the feature kernel reads the ROIs' windows of the tiles itself and no cloud is
    // materialised (8 B per ROI pixel written and read back otherwise).  Any other family, or ROIs beyond LDS: clouds.
    static const bool no_window = [] { const char* e = getenv("NYXHIP_NO_WINDOW"); return e && *e && *e != '0'; }();   // A/B and tests
    bool window = !no_window && (family_mask & ~(uint32_t)(NYXHIP_FAM_INTENSITY | NYXHIP_FAM_GLCM)) == 0;
    if (window) {
        LdsLayout Lt; std::string why_t;
        window = make_layout(family_mask, s, nyxhip_n_columns(family_mask, s), meta[3], meta[4], meta[5], Lt, why_t) == NYXHIP_OK;
    }
    size_t c = 0;
    const size_t o_cx = c; c = al(c + 2 * npx);
    const size_t o_cy = c; c = al(c + 2 * npx);
    const size_t o_cv = c; c = al(c + 4 * npx);
    nyxhip_batch b;
    memset(&b, 0, sizeof(b));
    b.n_roi = n_roi; b.roi_label = R.label; b.px_offset = R.px_offset;
    b.bbox_w = R.bbox_w; b.bbox_h = R.bbox_h; b.min_inten = R.vmin; b.max_inten = R.vmax;
    b.slide_min = R.slide_min; b.slide_max = R.slide_max;
    b.memory = NYXHIP_MEM_DEVICE;
    auto make_clouds = [&]() -> int {
        if (int grc = grow(ctx, &ctx->d_cloud, &ctx->cloud_bytes, c, st)) return grc;
        char* const cb = (char*)ctx->d_cloud;
        const int rc = launch_tile_clouds(d_inten, dtI, d_label, dtL, W, H, R, (uint32_t)n_roi, (uint16_t*)(cb + o_cx), (uint16_t*)(cb + o_cy), (uint32_t*)(cb + o_cv), st);
        if (rc) return fail(ctx, NYXHIP_ERR_HIP, std::string("cloud kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
        b.x = (const uint16_t*)(cb + o_cx); b.y = (const uint16_t*)(cb + o_cy); b.inten = (const uint32_t*)(cb + o_cv);
        return NYXHIP_OK;
    };
    if (!window)
        if (int crc = make_clouds()) return crc;
    HIP_TRY(ctx, hipMemcpyAsync(d_lab, R.label, 4 * n_roi, hipMemcpyDeviceToDevice, st));
    if (d_til) {
        if (tile_base == 0) HIP_TRY(ctx, hipMemcpyAsync(d_til, R.tile, 4 * n_roi, hipMemcpyDeviceToDevice, st));
        else hipLaunchKernelGGL(add_offset_kernel, dim3((unsigned)((n_roi + 255) / 256)), dim3(256), 0, st, R.tile, tile_base, (uint32_t)n_roi, d_til);
    }
    if (window)
    {
        static const bool no_swz = [] { const char* e = getenv("NYXHIP_NO_XCD_SWIZZLE"); return e && *e && *e != '0'; }();   // A/B
        ctx->win_next = WindowSrc{d_inten, d_label, dtI, dtL, W, H, R.tile, R.label, R.bbox_x0, R.bbox_y0, no_swz ? 0u : 1u};
    }
    int lrc = launch_device(ctx, &b, family_mask, s, d_out, d_ld, meta[3], meta[4], meta[5], meta[6]);
    ctx->win_next = WindowSrc{};
    if (lrc == NYXHIP_INTERNAL_NEEDS_CLOUDS) {
        // a size class of this chunk does not run from LDS under these settings (the whole-chunk extrema above could not tell: classes
        // get layouts of their own -- IBSI matrix orders, radix sort buffers of the wide-range classes): the workspace paths read clouds
        if (int crc = make_clouds()) return crc;
        lrc = launch_device(ctx, &b, family_mask, s, d_out, d_ld, meta[3], meta[4], meta[5], meta[6]);
    }
    return lrc;
}

static int tiles_validate(nyxhip_ctx* ctx, const nyxhip_tiles* t, uint32_t family_mask, const nyxhip_settings* s, uint64_t* n_roi_out)
{
    if (!ctx) return NYXHIP_ERR_INVALID_ARG;
    if (!t || !s || !n_roi_out) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "null tiles / settings / n_roi_out");
    if (t->n_tiles == 0) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "n_tiles must be >= 1");
    if (!t->inten || !t->label || t->width == 0 || t->height == 0) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "null pointer or empty tile");
    auto dt_ok = [](int d) { return d == NYXHIP_U8 || d == NYXHIP_U16 || d == NYXHIP_U32; };
    if (!dt_ok(t->inten_dtype) || !dt_ok(t->label_dtype)) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "tile element types must be NYXHIP_U8 / U16 / U32");
    if (family_mask == 0 || (family_mask & ~kImplemented)) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "bad family mask");
    if (t->memory != NYXHIP_MEM_HOST && t->memory != NYXHIP_MEM_DEVICE && t->memory != NYXHIP_MEM_HOST_OWN_MAPPING) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "bad memory kind");
    if (t->slide_mode < NYXHIP_SLIDE_MONTAGE || t->slide_mode > NYXHIP_SLIDE_GIVEN) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "bad slide_mode");
    if (t->slide_mode == NYXHIP_SLIDE_GIVEN && (!t->slide_min || !t->slide_max)) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "NYXHIP_SLIDE_GIVEN needs slide_min and slide_max");
    std::string why;
    if (!settings_ok(s, family_mask, why)) return fail(ctx, NYXHIP_ERR_INVALID_ARG, why);
    return NYXHIP_OK;
}

// Room for `rows` result rows of n_cols columns in the context's device-resident result (rows already there are kept).
static int res_reserve(nyxhip_ctx* ctx, size_t rows, size_t n_cols, hipStream_t st)
{
    if (ctx->d_res && ctx->res_cols == n_cols && rows <= ctx->res_cap) return NYXHIP_OK;
    const bool carry = ctx->d_res && ctx->res_cols == n_cols && ctx->res_rows > 0;
    const size_t cap = std::max(rows, carry ? ctx->res_cap * 2 : (size_t)0);
    const size_t bytes = (((size_t)cap * n_cols * 8 + 255) & ~(size_t)255) + 8 * cap + 256;
    void* nb = nullptr;
    HIP_TRY(ctx, hipMalloc(&nb, bytes));
    void* const od = ctx->d_res;
    const double* o_tab = od ? ctx->res_table() : nullptr;
    const uint32_t* o_lab = od ? ctx->res_label() : nullptr;
    const uint32_t* o_til = od ? ctx->res_tile() : nullptr;
    const size_t o_rows = ctx->res_rows;
    ctx->d_res = nb; ctx->res_cap = cap; ctx->res_cols = n_cols;
    if (carry) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->res_table(), o_tab, o_rows * n_cols * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->res_label(), o_lab, o_rows * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->res_tile(), o_til, o_rows * 4, hipMemcpyDeviceToDevice, st));
    } else
        ctx->res_rows = 0;
    if (od) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipFree(od)); }
    return NYXHIP_OK;
}

// [src, src + bytes) of pageable host memory -> device through the context's pinned ring: per piece of at most kStageSlotBytes, wait for
// the slot's previous DMA, copy the piece into the slot with a few host threads (one thread moves ~10 GB/s, the link takes 50), enqueue
// the DMA, go on with the next slot.  Host copy of piece i + 1 and DMA of piece i overlap.
static void parallel_copy(void* dst, const void* src, size_t n)
{
    static const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>(std::min(8u, std::max(1u, hw / 2)), n >> 21);     // >= 2 MiB per thread
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = ((n / nt) + 4095) & ~(size_t)4095;
    for (unsigned t = 1; t < nt; t++) {
        const size_t o = (size_t)t * per;
        if (o >= n) break;
        th.emplace_back([=]() { memcpy((char*)dst + o, (const char*)src + o, std::min(per, n - o)); });
    }
    memcpy(dst, src, std::min(per, n));
    for (auto& t : th) t.join();
}
static hipError_t staged_h2d(nyxhip_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st)
{
    static const bool no_stage = [] { const char* e = getenv("NYXHIP_NO_STAGING"); return e && *e && *e != '0'; }();   // A/B knob: the runtime's own pageable path
    if (no_stage || bytes < ((size_t)1 << 20)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    for (size_t o = 0; o < bytes; o += nyxhip_ctx::kStageSlotBytes) {
        const size_t len = std::min(nyxhip_ctx::kStageSlotBytes, bytes - o);
        const int k = ctx->h_stage_next;
        ctx->h_stage_next = (k + 1) % nyxhip_ctx::kStageSlots;
        if (!ctx->h_stage[k]) {
            if (hipError_t e = hipHostMalloc(&ctx->h_stage[k], nyxhip_ctx::kStageSlotBytes, hipHostMallocDefault); e != hipSuccess) { ctx->h_stage[k] = nullptr; return e; }
            if (hipError_t e = hipEventCreateWithFlags(&ctx->h_stage_done[k], hipEventDisableTiming); e != hipSuccess) return e;
        }
        if (ctx->h_stage_used[k])
            if (hipError_t e = hipEventSynchronize(ctx->h_stage_done[k]); e != hipSuccess) return e;      // the slot's previous piece has left it
        parallel_copy(ctx->h_stage[k], (const char*)src + o, len);
        if (hipError_t e = hipMemcpyAsync((char*)dst + o, ctx->h_stage[k], len, hipMemcpyHostToDevice, st); e != hipSuccess) return e;
        if (hipError_t e = hipEventRecord(ctx->h_stage_done[k], st); e != hipSuccess) return e;
        ctx->h_stage_used[k] = true;
    }
    return hipSuccess;
}

// Host tiles reach the device in one of two ways (nyxhip_tiles::memory):
//   NYXHIP_MEM_HOST              any host memory.  The bytes go through the library's OWN pinned staging ring (HostStager: hipHostMalloc'ed
//                                slots; a few host threads copy a piece into a slot, the DMA engine takes it from there, the next piece is
//                                copied meanwhile).  Nothing is assumed about the caller's allocator.
//   NYXHIP_MEM_HOST_OWN_MAPPING  the caller states that both arrays are mappings of their own (mmap, a page-aligned allocation that is not
//                                handed back to an allocator's arena while the call runs): their whole pages are registered for the call
//                                (hipHostRegister) and copied by DMA in place -- no staging copy.
// Round 3-5 registered whatever looked like a mapping of its own in /proc/self/maps (a rule that knew glibc's malloc only): pages of a
// malloc arena, registered and released, left the driver's user-pointer bookkeeping in a state in which a LATER copy from those
// addresses faulted on the GPU.  The decision now lies with the one who knows -- the caller.
// Only WHOLE PAGES inside the array are registered (rounded inward to 4 KiB; what lies in front of and behind them travels through the
// staging ring): two arrays of a call that share a page never overlap in a registration.  One guard per ARRAY: the sharded entry pins
// the whole stack once, before its threads copy their shares.
struct HostPin {
    void* p[2] = {nullptr, nullptr};
    uintptr_t lo[2] = {0, 0}, hi[2] = {0, 0};          // registered byte range of array k (empty: lo == hi)
    static constexpr uintptr_t kPage = 4096;
    void pin(int k, const void* ptr, size_t bytes)
    {
        static const bool no_pin = [] { const char* e = getenv("NYXHIP_NO_PIN"); return e && *e && *e != '0'; }();   // A/B knob
        if (no_pin || !ptr) return;
        const uintptr_t a = ((uintptr_t)ptr + kPage - 1) & ~(kPage - 1), z = ((uintptr_t)ptr + bytes) & ~(kPage - 1);
        if (z <= a) return;                                   // no whole page inside the array
        if (hipHostRegister((void*)a, z - a, hipHostRegisterDefault) == hipSuccess) { p[k] = (void*)a; lo[k] = a; hi[k] = z; } else (void)hipGetLastError();
    }
    // host -> device copy of [src, src + bytes) of array k: the part inside the registered pages as one (DMA) copy, what lies in
    // front of and behind them as pageable copies
    hipError_t h2d(nyxhip_ctx* ctx, int k, void* dst, const void* src, size_t bytes, hipStream_t st) const
    {
        const uintptr_t b0 = (uintptr_t)src, b1 = b0 + bytes;
        const uintptr_t m0 = std::min(std::max(b0, lo[k]), b1), m1 = std::max(std::min(b1, hi[k]), m0);   // the registered middle [m0, m1)
        if (lo[k] == hi[k] || m0 == m1) return staged_h2d(ctx, dst, src, bytes, st);
        hipError_t e = hipSuccess;
        if (m0 > b0) e = staged_h2d(ctx, dst, src, m0 - b0, st);
        if (e == hipSuccess) e = hipMemcpyAsync((char*)dst + (m0 - b0), (const void*)m0, m1 - m0, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && b1 > m1) e = staged_h2d(ctx, (char*)dst + (m1 - b0), (const void*)m1, b1 - m1, st);
        return e;
    }
    ~HostPin()
    {
        for (void* q : p)
            if (q && hipHostUnregister(q) != hipSuccess) {
                (void)hipGetLastError();
                if (getenv("NYXHIP_DEBUG")) fprintf(stderr, "[nyxhip] hipHostUnregister(%p) failed\n", q);
            }
    }
};

// The whole stack in chunks.  label_limit: v1's max_label (validated only).  prepinned: the caller has pinned the arrays.
static int tiles_run(nyxhip_ctx* ctx, const nyxhip_tiles* t, uint32_t family_mask, const nyxhip_settings* s, uint32_t* out_labels, uint32_t* out_tile_index,
                     uint64_t max_rows, double* out_table, size_t out_ld, uint64_t* n_roi_out, uint32_t label_limit, uint32_t tile_index_base = 0,
                     const HostPin* prepinned = nullptr)
{
    if (int vrc = tiles_validate(ctx, t, family_mask, s, n_roi_out)) return vrc;
    const int n_cols = nyxhip_n_columns(family_mask, s);
    const bool host = t->memory == NYXHIP_MEM_HOST || t->memory == NYXHIP_MEM_HOST_OWN_MAPPING;
    const bool keep = host && out_table == nullptr;                  // result stays in the context (nyxhip_fetch_result)
    if (!keep && (!out_labels || !out_table)) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "null output pointers");
    if (!keep && (int)out_ld < n_cols) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "out_ld smaller than the column count");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *n_roi_out = 0;
    hipStream_t st = ctx->stream();
    const uint32_t W = t->width, H = t->height;
    const uint64_t tile_px = (uint64_t)W * H;
    const size_t tile_in_bytes = (size_t)tile_px * (size_t)(t->inten_dtype + t->label_dtype);
    // ---- chunking (the reference batches ROIs by ram_limit, phase2_2d.cpp:694-705): per tile the scan tables and rows, the
    // clouds (<= 8 B per pixel), the table rows and -- host input -- two staging copies of the tile
    size_t budget = (size_t)t->max_device_bytes;
    if (budget == 0) {
        size_t fr = 0, tot = 0;
        HIP_TRY(ctx, hipMemGetInfo(&fr, &tot));
        // half of what is free, shared with the other contexts living on this device (gpu_devices=[0, 0], two sharded contexts
        // on one GPU: each taking half of the free memory for itself would together claim all of it)
        const int sharers = std::max(1, g_ctx_on_device[ctx->device & 63].load());
        budget = (fr + ctx->tile_bytes + ctx->cloud_bytes + ctx->slot_bytes[0] + ctx->slot_bytes[1]) / 2 / (size_t)sharers;
    }
    const uint32_t cap0 = ctx->tile_cap_hint ? ctx->tile_cap_hint : first_tile_cap(tile_px);
    const size_t per_tile = chunk_table_bytes(1, cap0) + 8 * (size_t)tile_px + (size_t)cap0 * 8 * n_cols / 8 + (host ? 2 * tile_in_bytes : 0);
    uint64_t chunk = std::max<uint64_t>(1, budget / std::max<size_t>(per_tile, 1));
    if (host) chunk = std::min<uint64_t>(chunk, std::max<uint64_t>(1, ((size_t)512 << 20) / tile_in_bytes));   // <= 512 MiB per copy: the pipeline needs chunks
    if (chunk > t->n_tiles) chunk = t->n_tiles;
    if (host && t->n_tiles >= 4 && chunk > (t->n_tiles + 1) / 2) chunk = (t->n_tiles + 1) / 2;                 // at least two chunks to overlap
    while ((uint64_t)chunk * cap0 > (1ull << 30) && chunk > 1) chunk /= 2;
    if (chunk > 65535) chunk = 65535;                      // the scan kernel spends grid.z on the tiles of a chunk (HIP: z <= 65535)
    // the per-tile table may grow while the stack is processed (a tile with more labels than slots: x 4 and rescan); the chunks
    // after that are sized for the table that is then in force
    auto rechunk = [&](uint64_t cur) -> uint64_t {
        const uint32_t capn = ctx->tile_cap_hint ? ctx->tile_cap_hint : cap0;
        if (capn <= cap0) return cur;
        const size_t pt = chunk_table_bytes(1, capn) + 8 * (size_t)tile_px + (size_t)capn * 8 * n_cols / 8 + (host ? 2 * tile_in_bytes : 0);
        uint64_t c2 = std::max<uint64_t>(1, budget / std::max<size_t>(pt, 1));
        while ((uint64_t)c2 * capn > (1ull << 30) && c2 > 1) c2 /= 2;
        return std::min(cur, c2);
    };

    if (keep) ctx->res_rows = 0;
    uint64_t rows_done = 0;
    bool short_out = false;
    if (!host) {
        for (uint64_t t0 = 0; t0 < t->n_tiles; t0 += chunk) {
            chunk = rechunk(chunk);
            const uint32_t nt = (uint32_t)std::min<uint64_t>(chunk, t->n_tiles - t0);
            const char* di = (const char*)t->inten + (size_t)t0 * tile_px * t->inten_dtype;
            const char* dl = (const char*)t->label + (size_t)t0 * tile_px * t->label_dtype;
            const uint64_t room = rows_done < max_rows ? max_rows - rows_done : 0;
            uint64_t n = 0;
            int rc = tiles_chunk(ctx, di, t->inten_dtype, dl, t->label_dtype, W, H, nt, t->slide_mode, t->slide_min ? t->slide_min + t0 : nullptr,
                                 t->slide_max ? t->slide_max + t0 : nullptr, family_mask, s, short_out ? 0 : room, out_labels + rows_done,
                                 out_tile_index ? out_tile_index + rows_done : nullptr, (uint32_t)t0, out_table + rows_done * out_ld, out_ld, label_limit, &n, st);
            if (rc) return rc;
            if (n > room) short_out = true;
            rows_done += n;
        }
        HIP_TRY(ctx, hipStreamSynchronize(st));
        *n_roi_out = rows_done;
        if (short_out) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "max_rows is smaller than the number of ROIs in the stack (see *n_roi_out)");
        return check_status(ctx);
    }

    // ---- host tiles: copy chunk c + 1 while chunk c is reduced --------------------------------------------------------------
    if (!ctx->copy_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->slot_ready[k], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->slot_free[k], hipEventDisableTiming));
        }
    }
    const size_t slot_need = (size_t)chunk * tile_in_bytes + 512;
    const uint64_t n_chunks = (t->n_tiles + chunk - 1) / chunk;
    for (int k = 0; k < (n_chunks > 1 ? 2 : 1); k++)
        if (int grc = grow(ctx, &ctx->d_slot[k], &ctx->slot_bytes[k], slot_need, st)) return grc;
    auto slot_inten = [&](int k) { return (char*)ctx->d_slot[k]; };
    auto slot_label = [&](int k, uint32_t nt) { return (char*)ctx->d_slot[k] + (((size_t)nt * tile_px * t->inten_dtype + 255) & ~(size_t)255); };
    HostPin pin;                                        // (see HostPin: only on the caller's statement, unless the sharded entry pinned the stack)
    if (!prepinned && t->memory == NYXHIP_MEM_HOST_OWN_MAPPING) {
        pin.pin(0, t->inten, (size_t)t->n_tiles * tile_px * t->inten_dtype);
        pin.pin(1, t->label, (size_t)t->n_tiles * tile_px * t->label_dtype);
    }
    const HostPin* const pins = prepinned ? prepinned : &pin;
    auto upload = [&](uint64_t c) -> int {                                  // chunk c -> slot c & 1 on the copy stream
        const int k = (int)(c & 1);
        const uint64_t t0 = c * chunk;
        const uint32_t nt = (uint32_t)std::min<uint64_t>(chunk, t->n_tiles - t0);
        if (c >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->slot_free[k], 0));      // the kernels of chunk c - 2 have let go of the slot
        HIP_TRY(ctx, pins->h2d(ctx, 0, slot_inten(k), (const char*)t->inten + (size_t)t0 * tile_px * t->inten_dtype, (size_t)nt * tile_px * t->inten_dtype, ctx->copy_stream));
        HIP_TRY(ctx, pins->h2d(ctx, 1, slot_label(k, nt), (const char*)t->label + (size_t)t0 * tile_px * t->label_dtype, (size_t)nt * tile_px * t->label_dtype, ctx->copy_stream));
        HIP_TRY(ctx, hipEventRecord(ctx->slot_ready[k], ctx->copy_stream));
        return NYXHIP_OK;
    };
    // every exit below -- the error returns included -- first waits for the copies and kernels still in flight: the pin guard above
    // unregisters the caller's arrays, and the caller may free them the moment this function returns
    struct Drain {
        hipStream_t a, b;
        ~Drain() { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); }
    } drain{ctx->copy_stream, st};
    if (int urc = upload(0)) return urc;
    for (uint64_t c = 0; c < n_chunks; c++) {
        const int k = (int)(c & 1);
        const uint64_t t0 = c * chunk;
        const uint32_t nt = (uint32_t)std::min<uint64_t>(chunk, t->n_tiles - t0);
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->slot_ready[k], 0));
        uint64_t n = 0;
        // the chunk's rows are produced in a device block owned by the context (d_stage), then copied out.  Its size follows the
        // ROI density seen so far (first chunk: 256 per tile); a denser chunk is rescanned once with the room it asked for.
        const uint64_t est_rows = std::max<uint64_t>((uint64_t)nt * 256, t0 ? (rows_done * 5 / 4 / t0 + 1) * nt : 0);
        size_t need = (size_t)est_rows * (8 * (size_t)n_cols + 8) + 1024;
        int rc;
        for (;;) {
            uint64_t cap_rows;
            double* d_out; uint32_t *d_lab, *d_til;
            if (keep) {                                   // rows are appended to the context's device-resident result: no copy, no sync per chunk
                if (int grc = res_reserve(ctx, (size_t)(rows_done + std::max<uint64_t>(est_rows, n)), (size_t)n_cols, st)) return grc;
                cap_rows = ctx->res_cap - rows_done;
                d_out = ctx->res_table() + rows_done * (size_t)n_cols; d_lab = ctx->res_label() + rows_done; d_til = ctx->res_tile() + rows_done;
            } else {
                if (int grc = ensure_stage(ctx, need)) return grc;
                cap_rows = (ctx->stage_bytes - 1024) / (8 * (size_t)n_cols + 8);
                d_out = (double*)ctx->d_stage;
                d_lab = (uint32_t*)((char*)ctx->d_stage + (((size_t)cap_rows * 8 * n_cols + 255) & ~(size_t)255));
                d_til = d_lab + cap_rows;
            }
            rc = tiles_chunk(ctx, slot_inten(k), t->inten_dtype, slot_label(k, nt), t->label_dtype, W, H, nt, t->slide_mode,
                             t->slide_min ? t->slide_min + t0 : nullptr, t->slide_max ? t->slide_max + t0 : nullptr, family_mask, s, cap_rows, d_lab, d_til,
                             tile_index_base + (uint32_t)t0, d_out, (size_t)n_cols, label_limit, &n, st);
            if (rc) return rc;
            if (n > cap_rows) { HIP_TRY(ctx, hipStreamSynchronize(st)); need = (size_t)n * (8 * (size_t)n_cols + 8) + 4096; continue; }
            HIP_TRY(ctx, hipEventRecord(ctx->slot_free[k], st));
            if (c + 1 < n_chunks)
                if (int urc = upload(c + 1)) return urc;                    // the next chunk's DMA runs beside this chunk's kernels
            const uint64_t room = rows_done < max_rows ? max_rows - rows_done : 0;
            if (keep) {
                ctx->res_rows = (size_t)(rows_done + n);
            } else if (n <= room && !short_out) {
                if (n) {
                    HIP_TRY(ctx, hipMemcpy2DAsync(out_table + rows_done * out_ld, out_ld * sizeof(double), d_out, (size_t)n_cols * sizeof(double),
                                                  (size_t)n_cols * sizeof(double), n, hipMemcpyDeviceToHost, st));
                    HIP_TRY(ctx, hipMemcpyAsync(out_labels + rows_done, d_lab, 4 * n, hipMemcpyDeviceToHost, st));
                    if (out_tile_index) HIP_TRY(ctx, hipMemcpyAsync(out_tile_index + rows_done, d_til, 4 * n, hipMemcpyDeviceToHost, st));
                }
                HIP_TRY(ctx, hipStreamSynchronize(st));                     // the chunk's rows are on the host; d_stage is free for the next one
            } else {
                short_out = true;
                HIP_TRY(ctx, hipStreamSynchronize(st));
            }
            break;
        }
        rows_done += n;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *n_roi_out = rows_done;
    if (short_out) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "max_rows is smaller than the number of ROIs in the stack (see *n_roi_out)");
    return check_status(ctx);
}

int nyxhip_featurize_tiles_v2(nyxhip_ctx* ctx, const nyxhip_tiles* tiles, uint32_t family_mask, const nyxhip_settings* s, uint32_t* out_labels,
                              uint32_t* out_tile_index, uint64_t max_rows, double* out_table, size_t out_ld, uint64_t* n_roi_out)
{
    return tiles_run(ctx, tiles, family_mask, s, out_labels, out_tile_index, max_rows, out_table, out_ld, n_roi_out, 0xFFFFFFFFu);
}

int nyxhip_fetch_result(nyxhip_ctx* ctx, uint32_t* out_labels, uint32_t* out_tile_index, double* out_table, size_t out_ld)
{
    if (!ctx) return NYXHIP_ERR_INVALID_ARG;
    const size_t n = ctx->res_rows, nc = ctx->res_cols;
    if (n && (!out_labels || !out_table || out_ld < nc)) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "null output pointers or out_ld smaller than the column count");
    if (n) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream();
        HIP_TRY(ctx, hipMemcpy2DAsync(out_table, out_ld * sizeof(double), ctx->res_table(), nc * sizeof(double), nc * sizeof(double), n, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(out_labels, ctx->res_label(), 4 * n, hipMemcpyDeviceToHost, st));
        if (out_tile_index) HIP_TRY(ctx, hipMemcpyAsync(out_tile_index, ctx->res_tile(), 4 * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    ctx->res_rows = 0;                                 // (the device block is kept for the next call)
    return NYXHIP_OK;
}

int nyxhip_featurize_tiles_sharded(nyxhip_ctx* const* ctxs, int n_ctx, const nyxhip_tiles* tiles, uint32_t family_mask, const nyxhip_settings* s,
                                   uint32_t* out_labels, uint32_t* out_tile_index, uint64_t max_rows, double* out_table, size_t out_ld, uint64_t* n_roi_out)
{
    if (!ctxs || n_ctx < 1 || !ctxs[0]) return NYXHIP_ERR_INVALID_ARG;
    nyxhip_ctx* c0 = ctxs[0];
    if (!tiles || !n_roi_out) return fail(c0, NYXHIP_ERR_INVALID_ARG, "null tiles / n_roi_out");
    if (tiles->memory != NYXHIP_MEM_HOST && tiles->memory != NYXHIP_MEM_HOST_OWN_MAPPING) return fail(c0, NYXHIP_ERR_INVALID_ARG, "the sharded entry takes host-memory stacks (every context copies its own share)");
    const bool keep = out_table == nullptr;                              // results stay in the contexts (nyxhip_fetch_result_sharded)
    if (!keep && !out_labels) return fail(c0, NYXHIP_ERR_INVALID_ARG, "null output pointers");
    for (int g = 0; g < n_ctx; g++)
        if (!ctxs[g]) return fail(c0, NYXHIP_ERR_INVALID_ARG, "null context in the list");
    const int G = (int)std::min<uint64_t>((uint64_t)n_ctx, tiles->n_tiles ? tiles->n_tiles : 1);
    // contiguous block partition (the first n % G contexts get one tile more); every context keeps its rows, which are then
    // laid out back to back in context order = stack order
    std::vector<int> rcs(G, 0);
    std::vector<uint64_t> cnt(G, 0), lo(G + 1, 0);
    const uint64_t q = tiles->n_tiles / G, r = tiles->n_tiles % G;
    for (int g = 0; g < G; g++) lo[g + 1] = lo[g] + q + ((uint64_t)g < r ? 1 : 0);
    const uint64_t tile_px = (uint64_t)tiles->width * tiles->height;
    for (int g = 0; g < n_ctx; g++) ctxs[g]->res_rows = 0;
    HostPin pin;                                        // the whole stack, once: released after every share's copies have drained (join below)
    if (tiles->memory == NYXHIP_MEM_HOST_OWN_MAPPING && hipSetDevice(c0->device) == hipSuccess) {
        pin.pin(0, tiles->inten, (size_t)tiles->n_tiles * tile_px * tiles->inten_dtype);
        pin.pin(1, tiles->label, (size_t)tiles->n_tiles * tile_px * tiles->label_dtype);
    } else (void)hipGetLastError();
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++)
        th.emplace_back([&, g]() {
            nyxhip_tiles part = *tiles;
            part.n_tiles = (uint32_t)(lo[g + 1] - lo[g]);
            part.inten = (const char*)tiles->inten + (size_t)lo[g] * tile_px * tiles->inten_dtype;
            part.label = (const char*)tiles->label + (size_t)lo[g] * tile_px * tiles->label_dtype;
            if (tiles->slide_min) part.slide_min = tiles->slide_min + lo[g];
            if (tiles->slide_max) part.slide_max = tiles->slide_max + lo[g];
            if (part.n_tiles == 0) { rcs[g] = 0; return; }
            rcs[g] = tiles_run(ctxs[g], &part, family_mask, s, nullptr, nullptr, 0, nullptr, 0, &cnt[g], 0xFFFFFFFFu, (uint32_t)lo[g], &pin);   // tile indices of the whole stack
        });
    for (auto& t : th) t.join();
    for (int g = 0; g < G; g++)
        if (rcs[g]) return g == 0 ? rcs[g] : fail(c0, rcs[g], std::string("context ") + std::to_string(g) + ": " + ctxs[g]->err);
    uint64_t total = 0;
    for (int g = 0; g < G; g++) total += cnt[g];
    *n_roi_out = total;
    if (keep) return NYXHIP_OK;
    if (total > max_rows) {
        for (int g = 0; g < G; g++) ctxs[g]->res_rows = 0;
        return fail(c0, NYXHIP_ERR_INVALID_ARG, "max_rows is smaller than the number of ROIs in the stack (see *n_roi_out)");
    }
    return nyxhip_fetch_result_sharded(ctxs, n_ctx, out_labels, out_tile_index, out_table, out_ld);
}

int nyxhip_fetch_result_sharded(nyxhip_ctx* const* ctxs, int n_ctx, uint32_t* out_labels, uint32_t* out_tile_index, double* out_table, size_t out_ld)
{
    if (!ctxs || n_ctx < 1) return NYXHIP_ERR_INVALID_ARG;
    uint64_t row = 0;
    for (int g = 0; g < n_ctx; g++) {
        if (!ctxs[g]) return NYXHIP_ERR_INVALID_ARG;
        const uint64_t n = ctxs[g]->res_rows;
        if (n) {
            int rc = nyxhip_fetch_result(ctxs[g], out_labels + row, out_tile_index ? out_tile_index + row : nullptr, out_table + row * out_ld, out_ld);
            if (rc) return rc;
        }
        row += n;
    }
    return NYXHIP_OK;
}

int nyxhip_featurize_tile(nyxhip_ctx* ctx, const uint32_t* inten, const uint32_t* label, uint32_t width, uint32_t height,
                          int32_t memory, uint32_t max_label, uint32_t family_mask, const nyxhip_settings* s,
                          uint32_t* out_labels, uint64_t max_rows, double* out_table, size_t out_ld, uint64_t* n_roi_out)
{
    return nyxhip_featurize_tiles(ctx, inten, label, width, height, 1, memory, max_label, family_mask, s, out_labels, nullptr,
                                  max_rows, out_table, out_ld, n_roi_out);
}

int nyxhip_featurize_tiles(nyxhip_ctx* ctx, const uint32_t* inten, const uint32_t* label, uint32_t width, uint32_t height,
                           uint32_t n_tiles, int32_t memory, uint32_t max_label, uint32_t family_mask, const nyxhip_settings* s,
                           uint32_t* out_labels, uint32_t* out_tile_index, uint64_t max_rows, double* out_table, size_t out_ld,
                           uint64_t* n_roi_out)
{
    if (!ctx) return NYXHIP_ERR_INVALID_ARG;
    if (!out_labels || !out_table) return fail(ctx, NYXHIP_ERR_INVALID_ARG, "null pointer or empty tile");
    nyxhip_tiles t;
    memset(&t, 0, sizeof(t));
    t.inten = inten; t.label = label; t.inten_dtype = NYXHIP_U32; t.label_dtype = NYXHIP_U32;
    t.width = width; t.height = height; t.n_tiles = n_tiles; t.memory = memory; t.slide_mode = NYXHIP_SLIDE_MONTAGE;
    return tiles_run(ctx, &t, family_mask, s, out_labels, out_tile_index, max_rows, out_table, out_ld, n_roi_out, max_label);
}

int nyxhip_timing_enable(nyxhip_ctx* ctx, int on)
{
    if (!ctx) return NYXHIP_ERR_INVALID_ARG;
    ctx->timing = on < 0 ? 0 : on > 2 ? 2 : on;
    return NYXHIP_OK;
}

int nyxhip_timing_reset(nyxhip_ctx* ctx)
{
    if (!ctx) return NYXHIP_ERR_INVALID_ARG;
    ctx->ev_used = 0;
    return NYXHIP_OK;
}

int nyxhip_timing_get(nyxhip_ctx* ctx, double* avg_kernel_ms, uint64_t* n_launches)
{
    if (!ctx) return NYXHIP_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double tot = 0;
    for (size_t i = 0; i < ctx->ev_used; i++) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev[i].second));
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[i].first, ctx->ev[i].second));
        tot += ms;
    }
    if (avg_kernel_ms) *avg_kernel_ms = ctx->ev_used ? tot / (double)ctx->ev_used : 0.0;
    if (n_launches) *n_launches = ctx->ev_used;
    return NYXHIP_OK;
}

int nyxhip_launch_report(nyxhip_ctx* ctx, char* buf, size_t buf_len)
{
    if (!ctx) return -NYXHIP_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    std::string js = "[";
    for (size_t i = 0; i < ctx->runs.size(); i++) {
        const ClassRun& r = ctx->runs[i];
        char ms[48] = "null";
        if (r.e0 && r.e1 && hipEventSynchronize(r.e1) == hipSuccess) {
            float t = 0;
            if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) snprintf(ms, sizeof(ms), "%.6f", (double)t);
        }
        char lane_ms[48] = "null";                      // from the class's start on the main stream to the end of its workspace lane
        if (r.e0 && r.e2 && hipEventSynchronize(r.e2) == hipSuccess) {
            float t = 0;
            if (hipEventElapsedTime(&t, r.e0, r.e2) == hipSuccess) snprintf(lane_ms, sizeof(lane_ms), "%.6f", (double)t);
        }
        char one[448];
        snprintf(one, sizeof(one), "%s{\"class\": %d, \"size_class\": %d, \"wide_range\": %d, \"rois\": %u, \"max_px\": %u, \"max_bbox_area\": %u, "
                 "\"max_range\": %u, \"max_side\": %u, \"workspace\": %d, \"cooperative\": %d, \"ms\": %s, \"lane_ms\": %s}", i ? ", " : "", r.cls, r.cls < 0 ? -1 : r.cls / 2, r.cls < 0 ? -1 : r.cls & 1,
                 r.count, r.E.px, r.E.area, r.E.range, r.E.side, r.workspace, r.cooperative, ms, lane_ms);
        js += one;
    }
    js += "]";
    if (buf && buf_len) {
        const size_t n = std::min(js.size(), buf_len - 1);
        memcpy(buf, js.data(), n);
        buf[n] = 0;
    }
    return (int)js.size();
}

} // extern "C"
