// nyxhip_feature_method.hpp -- the reference's plugin surface, served by the HIP hot path.
//
// A header-only C++ adapter above the C ABI (nyxhip.h) that mirrors, name for name, the pieces of the
// reference a feature plugin touches, so that code written against Nyxus' `FeatureMethod` shape -- and the
// reference's own unit tests (tests/test_2d_*_common.h: build an LR from a pixel list, `f.calculate(r, s)`,
// `f.save_value(r.fvals)`) -- reads the same here:
//
//   Feature2D / FeatureSet      /root/reference/src/nyx/featureset.h:10-641, featureset.cpp:1198
//   NyxSetting / Fsettings      src/nyx/feature_settings.h:6-54
//   Pixel2, AABB, LR            src/nyx/features/pixel.h:52-61, features/aabb.h, roi_cache.h:31-84
//   SlideProps / Dataset        src/nyx/slideprops.h:6-89, dataset.h:6-26
//   FeatureMethod               src/nyx/feature_method.h:11-80   (provide_features, required, calculate, save_value)
//   functype / runParallel      src/nyx/parallel.h:13,23-42      (static F::reduce(start,end,labels,roiData,settings,dataset))
//   reduce_trivial_rois_manual  src/nyx/reduce_trivial_rois.cpp:772-795
//
// Differences, all forced by the boundary: feature codes cover only the seven hot-path families (the enum
// keeps the reference's identifiers and relative order); `calculate()` of ANY family runs the fused kernels
// for that family on the GPU; `osized_*` (out-of-core ROIs) throw -- they are outside SURVEY.md section 8.
// Errors follow the WITH_PYTHON_H convention of the reference: std::runtime_error.
#pragma once
#include <algorithm>
#include <cstdint>
#include <initializer_list>
#include <limits>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "nyxhip.h"

namespace NyxusHip {

// ---- feature codes (identifiers and order of Nyxus::Feature2D, hot-path families only) -----------
enum class Feature2D : int {
    COV = 0, COVERED_IMAGE_INTENSITY_RANGE, ENERGY, ENTROPY, EXCESS_KURTOSIS, HYPERFLATNESS, HYPERSKEWNESS,
    INTEGRATED_INTENSITY, INTERQUARTILE_RANGE, KURTOSIS, MAX, MEAN, MEAN_ABSOLUTE_DEVIATION, MEDIAN,
    MEDIAN_ABSOLUTE_DEVIATION, MIN, MODE, P01, P10, P25, P75, P90, P99, QCOD, RANGE, ROBUST_MEAN,
    ROBUST_MEAN_ABSOLUTE_DEVIATION, ROOT_MEAN_SQUARED, SKEWNESS, STANDARD_DEVIATION, STANDARD_DEVIATION_BIASED,
    STANDARD_ERROR, VARIANCE, VARIANCE_BIASED, UNIFORMITY, UNIFORMITY_PIU,
    GLCM_ASM, GLCM_ACOR, GLCM_CLUPROM, GLCM_CLUSHADE, GLCM_CLUTEND, GLCM_CONTRAST, GLCM_CORRELATION, GLCM_DIFAVE,
    GLCM_DIFENTRO, GLCM_DIFVAR, GLCM_DIS, GLCM_ENERGY, GLCM_ENTROPY, GLCM_HOM1, GLCM_HOM2, GLCM_ID, GLCM_IDN,
    GLCM_IDM, GLCM_IDMN, GLCM_INFOMEAS1, GLCM_INFOMEAS2, GLCM_IV, GLCM_JAVE, GLCM_JE, GLCM_JMAX, GLCM_JVAR,
    GLCM_SUMAVERAGE, GLCM_SUMENTROPY, GLCM_SUMVARIANCE, GLCM_VARIANCE,
    GLCM_ASM_AVE, GLCM_ACOR_AVE, GLCM_CLUPROM_AVE, GLCM_CLUSHADE_AVE, GLCM_CLUTEND_AVE, GLCM_CONTRAST_AVE,
    GLCM_CORRELATION_AVE, GLCM_DIFAVE_AVE, GLCM_DIFENTRO_AVE, GLCM_DIFVAR_AVE, GLCM_DIS_AVE, GLCM_ENERGY_AVE,
    GLCM_ENTROPY_AVE, GLCM_HOM1_AVE, GLCM_ID_AVE, GLCM_IDN_AVE, GLCM_IDM_AVE, GLCM_IDMN_AVE, GLCM_IV_AVE,
    GLCM_JAVE_AVE, GLCM_JE_AVE, GLCM_INFOMEAS1_AVE, GLCM_INFOMEAS2_AVE, GLCM_VARIANCE_AVE, GLCM_JMAX_AVE,
    GLCM_JVAR_AVE, GLCM_SUMAVERAGE_AVE, GLCM_SUMENTROPY_AVE, GLCM_SUMVARIANCE_AVE,
    GLRLM_SRE, GLRLM_LRE, GLRLM_GLN, GLRLM_GLNN, GLRLM_RLN, GLRLM_RLNN, GLRLM_RP, GLRLM_GLV, GLRLM_RV, GLRLM_RE,
    GLRLM_LGLRE, GLRLM_HGLRE, GLRLM_SRLGLE, GLRLM_SRHGLE, GLRLM_LRLGLE, GLRLM_LRHGLE,
    GLRLM_SRE_AVE, GLRLM_LRE_AVE, GLRLM_GLN_AVE, GLRLM_GLNN_AVE, GLRLM_RLN_AVE, GLRLM_RLNN_AVE, GLRLM_RP_AVE,
    GLRLM_GLV_AVE, GLRLM_RV_AVE, GLRLM_RE_AVE, GLRLM_LGLRE_AVE, GLRLM_HGLRE_AVE, GLRLM_SRLGLE_AVE,
    GLRLM_SRHGLE_AVE, GLRLM_LRLGLE_AVE, GLRLM_LRHGLE_AVE,
    GLDZM_SDE, GLDZM_LDE, GLDZM_LGLZE, GLDZM_HGLZE, GLDZM_SDLGLE, GLDZM_SDHGLE, GLDZM_LDLGLE, GLDZM_LDHGLE, GLDZM_GLNU,
    GLDZM_GLNUN, GLDZM_ZDNU, GLDZM_ZDNUN, GLDZM_ZP, GLDZM_GLM, GLDZM_GLV, GLDZM_ZDM, GLDZM_ZDV, GLDZM_ZDE,
    GLSZM_SAE, GLSZM_LAE, GLSZM_GLN, GLSZM_GLNN, GLSZM_SZN, GLSZM_SZNN, GLSZM_ZP, GLSZM_GLV, GLSZM_ZV, GLSZM_ZE,
    GLSZM_LGLZE, GLSZM_HGLZE, GLSZM_SALGLE, GLSZM_SAHGLE, GLSZM_LALGLE, GLSZM_LAHGLE,
    GLDM_SDE, GLDM_LDE, GLDM_GLN, GLDM_DN, GLDM_DNN, GLDM_GLV, GLDM_DV, GLDM_DE, GLDM_LGLE, GLDM_HGLE, GLDM_SDLGLE,
    GLDM_SDHGLE, GLDM_LDLGLE, GLDM_LDHGLE,
    NGLDM_LDE, NGLDM_HDE, NGLDM_LGLCE, NGLDM_HGLCE, NGLDM_LDLGLE, NGLDM_LDHGLE, NGLDM_HDLGLE, NGLDM_HDHGLE, NGLDM_GLNU,
    NGLDM_GLNUN, NGLDM_DCNU, NGLDM_DCNUN, NGLDM_DCP, NGLDM_GLM, NGLDM_GLV, NGLDM_DCM, NGLDM_DCV, NGLDM_DCENT, NGLDM_DCENE,
    NGTDM_COARSENESS, NGTDM_CONTRAST, NGTDM_BUSYNESS, NGTDM_COMPLEXITY, NGTDM_STRENGTH,
    GABOR, ZERNIKE2D,
    // shape / intensity geometric moments (featureset.h:362-565)
    SPAT_MOMENT_00, SPAT_MOMENT_01, SPAT_MOMENT_02, SPAT_MOMENT_03, SPAT_MOMENT_10, SPAT_MOMENT_11, SPAT_MOMENT_12,
    SPAT_MOMENT_13, SPAT_MOMENT_20, SPAT_MOMENT_21, SPAT_MOMENT_22, SPAT_MOMENT_23, SPAT_MOMENT_30, CENTRAL_MOMENT_00,
    CENTRAL_MOMENT_01, CENTRAL_MOMENT_02, CENTRAL_MOMENT_03, CENTRAL_MOMENT_10, CENTRAL_MOMENT_11, CENTRAL_MOMENT_12,
    CENTRAL_MOMENT_13, CENTRAL_MOMENT_20, CENTRAL_MOMENT_21, CENTRAL_MOMENT_22, CENTRAL_MOMENT_23, CENTRAL_MOMENT_30,
    CENTRAL_MOMENT_31, CENTRAL_MOMENT_32, CENTRAL_MOMENT_33, NORM_SPAT_MOMENT_00, NORM_SPAT_MOMENT_01,
    NORM_SPAT_MOMENT_02, NORM_SPAT_MOMENT_03, NORM_SPAT_MOMENT_10, NORM_SPAT_MOMENT_11, NORM_SPAT_MOMENT_12,
    NORM_SPAT_MOMENT_13, NORM_SPAT_MOMENT_20, NORM_SPAT_MOMENT_21, NORM_SPAT_MOMENT_22, NORM_SPAT_MOMENT_23,
    NORM_SPAT_MOMENT_30, NORM_SPAT_MOMENT_31, NORM_SPAT_MOMENT_32, NORM_SPAT_MOMENT_33, NORM_CENTRAL_MOMENT_02,
    NORM_CENTRAL_MOMENT_03, NORM_CENTRAL_MOMENT_11, NORM_CENTRAL_MOMENT_12, NORM_CENTRAL_MOMENT_20,
    NORM_CENTRAL_MOMENT_21, NORM_CENTRAL_MOMENT_30, HU_M1, HU_M2, HU_M3, HU_M4, HU_M5, HU_M6, HU_M7,
    WEIGHTED_SPAT_MOMENT_00, WEIGHTED_SPAT_MOMENT_01, WEIGHTED_SPAT_MOMENT_02, WEIGHTED_SPAT_MOMENT_03,
    WEIGHTED_SPAT_MOMENT_10, WEIGHTED_SPAT_MOMENT_11, WEIGHTED_SPAT_MOMENT_12, WEIGHTED_SPAT_MOMENT_20,
    WEIGHTED_SPAT_MOMENT_21, WEIGHTED_SPAT_MOMENT_30, WEIGHTED_CENTRAL_MOMENT_02, WEIGHTED_CENTRAL_MOMENT_03,
    WEIGHTED_CENTRAL_MOMENT_11, WEIGHTED_CENTRAL_MOMENT_12, WEIGHTED_CENTRAL_MOMENT_20, WEIGHTED_CENTRAL_MOMENT_21,
    WEIGHTED_CENTRAL_MOMENT_30, WT_NORM_CTR_MOM_02, WT_NORM_CTR_MOM_03, WT_NORM_CTR_MOM_11, WT_NORM_CTR_MOM_12,
    WT_NORM_CTR_MOM_20, WT_NORM_CTR_MOM_21, WT_NORM_CTR_MOM_30, WEIGHTED_HU_M1, WEIGHTED_HU_M2, WEIGHTED_HU_M3,
    WEIGHTED_HU_M4, WEIGHTED_HU_M5, WEIGHTED_HU_M6, WEIGHTED_HU_M7,
    IMOM_RM_00, IMOM_RM_01, IMOM_RM_02, IMOM_RM_03, IMOM_RM_10, IMOM_RM_11, IMOM_RM_12, IMOM_RM_13, IMOM_RM_20,
    IMOM_RM_21, IMOM_RM_22, IMOM_RM_23, IMOM_RM_30, IMOM_CM_00, IMOM_CM_01, IMOM_CM_02, IMOM_CM_03, IMOM_CM_10,
    IMOM_CM_11, IMOM_CM_12, IMOM_CM_13, IMOM_CM_20, IMOM_CM_21, IMOM_CM_22, IMOM_CM_23, IMOM_CM_30, IMOM_CM_31,
    IMOM_CM_32, IMOM_CM_33, IMOM_NRM_00, IMOM_NRM_01, IMOM_NRM_02, IMOM_NRM_03, IMOM_NRM_10, IMOM_NRM_11, IMOM_NRM_12,
    IMOM_NRM_13, IMOM_NRM_20, IMOM_NRM_21, IMOM_NRM_22, IMOM_NRM_23, IMOM_NRM_30, IMOM_NRM_31, IMOM_NRM_32,
    IMOM_NRM_33, IMOM_NCM_02, IMOM_NCM_03, IMOM_NCM_11, IMOM_NCM_12, IMOM_NCM_20, IMOM_NCM_21, IMOM_NCM_30, IMOM_HU1,
    IMOM_HU2, IMOM_HU3, IMOM_HU4, IMOM_HU5, IMOM_HU6, IMOM_HU7, IMOM_WRM_00, IMOM_WRM_01, IMOM_WRM_02, IMOM_WRM_03,
    IMOM_WRM_10, IMOM_WRM_11, IMOM_WRM_12, IMOM_WRM_20, IMOM_WRM_21, IMOM_WRM_30, IMOM_WCM_02, IMOM_WCM_03,
    IMOM_WCM_11, IMOM_WCM_12, IMOM_WCM_20, IMOM_WCM_21, IMOM_WCM_30, IMOM_WNCM_02, IMOM_WNCM_03, IMOM_WNCM_11,
    IMOM_WNCM_12, IMOM_WNCM_20, IMOM_WNCM_21, IMOM_WNCM_30, IMOM_WHU1, IMOM_WHU2, IMOM_WHU3, IMOM_WHU4, IMOM_WHU5,
    IMOM_WHU6, IMOM_WHU7,
    _COUNT_
};

// ---- settings (feature_settings.h:6-54) ----------------------------------------------------------
union FeatureSetting { bool bval; int ival; double rval; };
typedef std::vector<FeatureSetting> Fsettings;
enum class NyxSetting : int { SOFTNAN = 0, TINY, SINGLEROI, GREYDEPTH, PIXELSIZEUM, PIXELDISTANCE, XYRES, USEGPU, VERBOSLVL, IBSI,
                              GLCM_GREYDEPTH, GLCM_OFFSET, GLCM_NUMANG, __COUNT__ };
#define NYXHIP_STNGS_MISSING(obj) ((int)(obj).size() < (int)NyxusHip::NyxSetting::__COUNT__)

// ---- ROI record (roi_cache.h:31-84) ---------------------------------------------------------------
using PixIntens = unsigned int;
using StatsInt = long;
struct Pixel2 { StatsInt x, y; PixIntens inten; Pixel2(StatsInt x_ = 0, StatsInt y_ = 0, PixIntens i_ = 0) : x(x_), y(y_), inten(i_) {} };

class AABB {
public:
    void init_x(StatsInt x) { xmin = xmax = x; }
    void init_y(StatsInt y) { ymin = ymax = y; }
    void update_x(StatsInt x) { xmin = std::min(xmin, x); xmax = std::max(xmax, x); }
    void update_y(StatsInt y) { ymin = std::min(ymin, y); ymax = std::max(ymax, y); }
    StatsInt get_height() const { return ymax - ymin + 1; }
    StatsInt get_width() const { return xmax - xmin + 1; }
    StatsInt get_xmin() const { return xmin; }
    StatsInt get_ymin() const { return ymin; }
private:
    StatsInt xmin = std::numeric_limits<StatsInt>::max(), xmax = std::numeric_limits<StatsInt>::min(),
             ymin = std::numeric_limits<StatsInt>::max(), ymax = std::numeric_limits<StatsInt>::min();
};

class LR {
public:
    LR(int roi_label = -1) : label(roi_label) {}
    int label;
    int slide_idx = -1;
    std::vector<Pixel2> raw_pixels;
    unsigned int aux_area = 0;
    PixIntens aux_min = (std::numeric_limits<PixIntens>::max)(), aux_max = 0;
    AABB aabb;
    std::vector<std::vector<double>> fvals;
    void initialize_fvals() { fvals.assign((size_t)Feature2D::_COUNT_, std::vector<double>(1, 0.0)); }   // roi_cache.cpp:105-110
    // phase-1 bookkeeping of one pixel (feed_pixel_2_metrics, pixel_feed.cpp:19-43 + feed_pixel_2_cache :71-74)
    void feed_pixel(StatsInt x, StatsInt y, PixIntens i)
    {
        if (aux_area == 0) { aabb.init_x(x); aabb.init_y(y); } else { aabb.update_x(x); aabb.update_y(y); }
        aux_area++;
        aux_min = std::min(aux_min, i);
        aux_max = std::max(aux_max, i);
        raw_pixels.push_back(Pixel2(x, y, i));
    }
};

struct SlideProps { double min_preroi_inten = -1, max_preroi_inten = -1; };
struct Dataset { std::vector<SlideProps> dataset_props; };

// ---- feature-set bits (featureset.h FeatureSet) ------------------------------------------------------
class FeatureSet {
public:
    FeatureSet() : bits((size_t)Feature2D::_COUNT_, false) {}
    void enableFeature(Feature2D f) { bits[(size_t)f] = true; }
    void enableFeatures(const std::initializer_list<Feature2D>& F) { for (auto f : F) bits[(size_t)f] = true; }
    bool isEnabled(Feature2D f) const { return bits[(size_t)f]; }
    bool anyEnabled(const std::initializer_list<Feature2D>& F) const { for (auto f : F) if (bits[(size_t)f]) return true; return false; }
    bool anyEnabledInRange(Feature2D a, Feature2D b) const { for (int i = (int)a; i <= (int)b; i++) if (bits[(size_t)i]) return true; return false; }
private:
    std::vector<bool> bits;
};

// ---- one process-wide device context ------------------------------------------------------------------
inline nyxhip_ctx* context(int device = 0)
{
    static nyxhip_ctx* ctx = nullptr;
    if (!ctx && nyxhip_init(device, &ctx) != NYXHIP_OK)
        throw std::runtime_error(std::string("nyxhip_init: ") + nyxhip_last_error(nullptr));
    return ctx;
}

// Process-global knobs the reference keeps as class statics (glcm.cpp:8-9, gabor.cpp:14-25)
struct Knobs {
    std::vector<int> glcm_angles{0, 45, 90, 135};
    bool symmetric_glcm = false;
    double gabor_gamma = 0.1, gabor_sig2lam = 0.8, gabor_f0LP = 0.1, gabor_GRAYthr = 0.025;
    int gabor_n = 16;
    std::vector<std::pair<double, double>> f0_theta_pairs{{0, 4.0}, {0.78539816339744830962, 16.0}, {1.57079632679489661923, 32.0}, {0.78539816339744830962 * 3.0, 64.0}};
};
inline Knobs& knobs() { static Knobs k; return k; }

inline nyxhip_settings make_settings(const Fsettings& s)
{
    nyxhip_settings o;
    nyxhip_default_settings(&o);
    const Knobs& k = knobs();
    if (NYXHIP_STNGS_MISSING(s)) {
        o.grey_depth = 24;                                 // DEFAULT_NUM_HISTO_BINS when settings are missing (intensity.cpp:125)
        o.glcm_grey_depth = 24;
    } else {
        o.soft_nan = s[(int)NyxSetting::SOFTNAN].rval;
        o.tiny = s[(int)NyxSetting::TINY].rval;
        o.grey_depth = s[(int)NyxSetting::GREYDEPTH].ival;
        o.ibsi = s[(int)NyxSetting::IBSI].bval ? 1 : 0;
        o.glcm_grey_depth = s[(int)NyxSetting::GLCM_GREYDEPTH].ival;
        o.glcm_offset = s[(int)NyxSetting::GLCM_OFFSET].ival;
        if (o.ibsi && o.grey_depth == 0) o.grey_depth = 1; // histogram bins are irrelevant to the IBSI texture paths
    }
    o.glcm_n_angles = (int)k.glcm_angles.size();
    for (int i = 0; i < o.glcm_n_angles; i++) o.glcm_angles[i] = k.glcm_angles[i];
    o.glcm_symmetric = k.symmetric_glcm ? 1 : 0;
    o.gabor_gamma = k.gabor_gamma; o.gabor_sig2lam = k.gabor_sig2lam; o.gabor_f0lp = k.gabor_f0LP; o.gabor_graythr = k.gabor_GRAYthr;
    o.gabor_kersize = k.gabor_n; o.gabor_n_filters = (int)k.f0_theta_pairs.size();
    for (int i = 0; i < o.gabor_n_filters; i++) { o.gabor_f0[i] = k.f0_theta_pairs[i].first; o.gabor_theta[i] = k.f0_theta_pairs[i].second; }
    return o;
}

// Gathers labels[start,end) into the SoA batch, runs the families of `mask`, scatters into LR::fvals.
inline void reduce_range(uint32_t mask, size_t start, size_t end, std::vector<int>* labels, std::unordered_map<int, LR>* roiData,
                         const Fsettings& fst, const Dataset& ds)
{
    if (end <= start) return;
    nyxhip_settings s = make_settings(fst);
    const size_t n = end - start;
    std::vector<uint32_t> lab(n), bw(n), bh(n), mn(n), mx(n), inten;
    std::vector<uint64_t> off(n + 1, 0);
    std::vector<uint16_t> x, y;
    std::vector<double> smin(n), smax(n);
    bool have_slide = true;
    for (size_t i = 0; i < n; i++) {
        LR& r = (*roiData)[(*labels)[start + i]];
        lab[i] = (uint32_t)r.label; bw[i] = (uint32_t)r.aabb.get_width(); bh[i] = (uint32_t)r.aabb.get_height();
        mn[i] = r.aux_min; mx[i] = r.aux_max;
        for (const Pixel2& p : r.raw_pixels) { x.push_back((uint16_t)(p.x - r.aabb.get_xmin())); y.push_back((uint16_t)(p.y - r.aabb.get_ymin())); inten.push_back(p.inten); }
        off[i + 1] = inten.size();
        if (r.slide_idx >= 0 && (size_t)r.slide_idx < ds.dataset_props.size()) { smin[i] = ds.dataset_props[r.slide_idx].min_preroi_inten; smax[i] = ds.dataset_props[r.slide_idx].max_preroi_inten; }
        else have_slide = false;
        if (r.fvals.empty()) r.initialize_fvals();
    }
    nyxhip_batch b{};
    b.n_roi = n; b.roi_label = lab.data(); b.px_offset = off.data(); b.x = x.data(); b.y = y.data(); b.inten = inten.data();
    b.bbox_w = bw.data(); b.bbox_h = bh.data(); b.min_inten = mn.data(); b.max_inten = mx.data();
    if (have_slide) { b.slide_min = smin.data(); b.slide_max = smax.data(); }
    b.memory = NYXHIP_MEM_HOST;
    const int ncol = nyxhip_n_columns(mask, &s);
    std::vector<double> table(n * (size_t)ncol);
    nyxhip_ctx* ctx = context();
    if (nyxhip_featurize_batch(ctx, &b, mask, &s, table.data(), (size_t)ncol) != NYXHIP_OK)
        throw std::runtime_error(std::string("nyxhip_featurize_batch: ") + nyxhip_last_error(ctx));
    // table columns are Feature2D order with angle / index expansion (output_2_buffer.cpp:303-584)
    const int na = s.glcm_n_angles;
    for (size_t i = 0; i < n; i++) {
        LR& r = (*roiData)[(*labels)[start + i]];
        const double* p = table.data() + i * (size_t)ncol;
        auto put = [&](Feature2D first, Feature2D last, int width) {
            for (int f = (int)first; f <= (int)last; f++) { r.fvals[f].assign(p, p + width); p += width; }
        };
        if (mask & NYXHIP_FAM_INTENSITY) put(Feature2D::COV, Feature2D::UNIFORMITY_PIU, 1);
        if (mask & NYXHIP_FAM_GLCM) { put(Feature2D::GLCM_ASM, Feature2D::GLCM_VARIANCE, na); put(Feature2D::GLCM_ASM_AVE, Feature2D::GLCM_SUMVARIANCE_AVE, 1); }
        if (mask & NYXHIP_FAM_GLRLM) { put(Feature2D::GLRLM_SRE, Feature2D::GLRLM_LRHGLE, 4); put(Feature2D::GLRLM_SRE_AVE, Feature2D::GLRLM_LRHGLE_AVE, 1); }
        if (mask & NYXHIP_FAM_GLDZM) put(Feature2D::GLDZM_SDE, Feature2D::GLDZM_ZDE, 1);
        if (mask & NYXHIP_FAM_GLSZM) put(Feature2D::GLSZM_SAE, Feature2D::GLSZM_LAHGLE, 1);
        if (mask & NYXHIP_FAM_GLDM) put(Feature2D::GLDM_SDE, Feature2D::GLDM_LDHGLE, 1);
        if (mask & NYXHIP_FAM_NGLDM) put(Feature2D::NGLDM_LDE, Feature2D::NGLDM_DCENE, 1);
        if (mask & NYXHIP_FAM_NGTDM) put(Feature2D::NGTDM_COARSENESS, Feature2D::NGTDM_STRENGTH, 1);
        if (mask & NYXHIP_FAM_GABOR) put(Feature2D::GABOR, Feature2D::GABOR, s.gabor_n_filters);
        if (mask & NYXHIP_FAM_ZERNIKE) put(Feature2D::ZERNIKE2D, Feature2D::ZERNIKE2D, 30);
        if (mask & NYXHIP_FAM_SMOMS) put(Feature2D::SPAT_MOMENT_00, Feature2D::WEIGHTED_HU_M7, 1);
        if (mask & NYXHIP_FAM_IMOMS) put(Feature2D::IMOM_RM_00, Feature2D::IMOM_WHU7, 1);
    }
}

// ---- FeatureMethod (feature_method.h:11-80) ------------------------------------------------------------
class FeatureMethod {
public:
    std::string feature_info;
    explicit FeatureMethod(const std::string& info) : feature_info(info) {}
    virtual ~FeatureMethod() {}
    virtual void calculate(LR& roi, const Fsettings& settings) = 0;
    virtual void save_value(std::vector<std::vector<double>>& feature_vals) = 0;
    virtual void osized_add_online_pixel(size_t, size_t, uint32_t) { throw std::runtime_error(feature_info + ": out-of-core ROIs are outside the MI355X hot path"); }
    virtual void osized_calculate(LR&, const Fsettings&) { throw std::runtime_error(feature_info + ": out-of-core ROIs are outside the MI355X hot path"); }
    void provide_features(Feature2D first, Feature2D last) { for (int f = (int)first; f <= (int)last; f++) provided.push_back(f); }
    bool provides(int code) const { return std::find(provided.begin(), provided.end(), code) != provided.end(); }
protected:
    std::vector<int> provided;
    std::vector<std::vector<double>> held;   // values of the last calculate(), indexed by Feature2D
    void run_single(uint32_t mask, LR& r, const Fsettings& s, const Dataset& ds)
    {
        std::vector<int> L{r.label};
        std::unordered_map<int, LR> data;
        data[r.label] = r;
        reduce_range(mask, 0, 1, &L, &data, s, ds);
        held = data[r.label].fvals;
    }
    void save_range(std::vector<std::vector<double>>& fv, Feature2D a, Feature2D b) const
    {
        if (fv.size() < (size_t)Feature2D::_COUNT_) fv.resize((size_t)Feature2D::_COUNT_, std::vector<double>(1, 0.0));
        if (held.empty()) throw std::runtime_error(feature_info + ": save_value() before calculate()");
        for (int f = (int)a; f <= (int)b; f++) fv[f] = held[f];
    }
};

#define NYXHIP_FAMILY_CLASS(NAME, MASK, FIRST, LAST)                                                                          \
    class NAME : public FeatureMethod {                                                                                          \
    public:                                                                                                                      \
        NAME() : FeatureMethod(#NAME) { provide_features(Feature2D::FIRST, Feature2D::LAST); }                                   \
        static bool required(const FeatureSet& fs) { return fs.anyEnabledInRange(Feature2D::FIRST, Feature2D::LAST); }            \
        void calculate(LR& r, const Fsettings& s) override { run_single(MASK, r, s, Dataset()); }                                 \
        void calculate(LR& r, const Fsettings& s, const Dataset& ds) { run_single(MASK, r, s, ds); }                              \
        void save_value(std::vector<std::vector<double>>& fv) override { save_range(fv, Feature2D::FIRST, Feature2D::LAST); }     \
        static void extract(LR& r, const Fsettings& s, const Dataset& ds = Dataset()) { NAME f; f.calculate(r, s, ds); f.save_value(r.fvals); } \
        /* the `functype` of parallel.h:13 */                                                                                    \
        static void reduce(size_t start, size_t end, std::vector<int>* labels, std::unordered_map<int, LR>* roiData,              \
                           const Fsettings& s, const Dataset& ds) { reduce_range(MASK, start, end, labels, roiData, s, ds); }     \
        static void parallel_process_1_batch(size_t start, size_t end, std::vector<int>* labels, std::unordered_map<int, LR>* roiData, \
                                             const Fsettings& s, const Dataset& ds) { reduce_range(MASK, start, end, labels, roiData, s, ds); } \
    };

NYXHIP_FAMILY_CLASS(PixelIntensityFeatures, NYXHIP_FAM_INTENSITY, COV, UNIFORMITY_PIU)
NYXHIP_FAMILY_CLASS(GLCMFeature, NYXHIP_FAM_GLCM, GLCM_ASM, GLCM_SUMVARIANCE_AVE)
NYXHIP_FAMILY_CLASS(GLRLMFeature, NYXHIP_FAM_GLRLM, GLRLM_SRE, GLRLM_LRHGLE_AVE)
NYXHIP_FAMILY_CLASS(GLDZMFeature, NYXHIP_FAM_GLDZM, GLDZM_SDE, GLDZM_ZDE)
NYXHIP_FAMILY_CLASS(GLSZMFeature, NYXHIP_FAM_GLSZM, GLSZM_SAE, GLSZM_LAHGLE)
NYXHIP_FAMILY_CLASS(GLDMFeature, NYXHIP_FAM_GLDM, GLDM_SDE, GLDM_LDHGLE)
NYXHIP_FAMILY_CLASS(NGLDMfeature, NYXHIP_FAM_NGLDM, NGLDM_LDE, NGLDM_DCENE)
NYXHIP_FAMILY_CLASS(NGTDMFeature, NYXHIP_FAM_NGTDM, NGTDM_COARSENESS, NGTDM_STRENGTH)
NYXHIP_FAMILY_CLASS(GaborFeature, NYXHIP_FAM_GABOR, GABOR, GABOR)
NYXHIP_FAMILY_CLASS(ZernikeFeature, NYXHIP_FAM_ZERNIKE, ZERNIKE2D, ZERNIKE2D)
// the contour the weighted moments depend on (ContourFeature::reduce, reduce_trivial_rois.cpp:98-111) is built inside the call
NYXHIP_FAMILY_CLASS(Smoms2D_feature, NYXHIP_FAM_SMOMS, SPAT_MOMENT_00, WEIGHTED_HU_M7)
NYXHIP_FAMILY_CLASS(Imoms2D_feature, NYXHIP_FAM_IMOMS, IMOM_RM_00, IMOM_WHU7)

// runParallel (parallel.h:23-42): the GPU batch is the parallel unit, so the slices run back to back on the
// caller's thread -- same observable contract (every label of [0, datasetSize) reduced on return).
typedef void (*functype)(size_t, size_t, std::vector<int>*, std::unordered_map<int, LR>*, const Fsettings&, const Dataset&);
inline void runParallel(functype f, int /*nThr*/, size_t /*workPerThread*/, size_t datasetSize, std::vector<int>* labels,
                        std::unordered_map<int, LR>* roiData, const Fsettings& s, const Dataset& ds)
{
    f(0, datasetSize, labels, roiData, s, ds);
}

// reduce_trivial_rois_manual (reduce_trivial_rois.cpp:772-795): every required family in ONE fused batch call.
inline void reduce_trivial_rois_manual(std::vector<int>& PendingRoisLabels, std::unordered_map<int, LR>& roiData, const FeatureSet& fs,
                                       const Fsettings& s, const Dataset& ds)
{
    uint32_t mask = 0;
    if (PixelIntensityFeatures::required(fs)) mask |= NYXHIP_FAM_INTENSITY;
    if (GLCMFeature::required(fs)) mask |= NYXHIP_FAM_GLCM;
    if (GLRLMFeature::required(fs)) mask |= NYXHIP_FAM_GLRLM;
    if (GLDZMFeature::required(fs)) mask |= NYXHIP_FAM_GLDZM;   // reduce_trivial_rois.cpp:215-220
    if (GLSZMFeature::required(fs)) mask |= NYXHIP_FAM_GLSZM;
    if (GLDMFeature::required(fs)) mask |= NYXHIP_FAM_GLDM;     // :231-236
    if (NGLDMfeature::required(fs)) mask |= NYXHIP_FAM_NGLDM;   // :239-244
    if (NGTDMFeature::required(fs)) mask |= NYXHIP_FAM_NGTDM;
    if (GaborFeature::required(fs)) mask |= NYXHIP_FAM_GABOR;
    if (ZernikeFeature::required(fs)) mask |= NYXHIP_FAM_ZERNIKE;
    if (Smoms2D_feature::required(fs)) mask |= NYXHIP_FAM_SMOMS;   // reduce_trivial_rois.cpp:326-331
    if (Imoms2D_feature::required(fs)) mask |= NYXHIP_FAM_IMOMS;   // :320-325
    if (mask) reduce_range(mask, 0, PendingRoisLabels.size(), &PendingRoisLabels, &roiData, s, ds);
}

} // namespace NyxusHip
