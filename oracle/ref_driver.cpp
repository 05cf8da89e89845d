/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A thin driver (own code) around the REAL reference feature classes, compiled
 * from the sources where they lie under /root/reference (never copied) by
 * oracle/Makefile into oracle/_ref/libnyxref.so.  It converts an SoA
 * `nyxhip_batch` (host memory) into the reference's own containers
 * (`std::vector<int>` labels + `std::unordered_map<int, LR>`, parallel.h:13),
 * then dispatches exactly as reduce_trivial_2d does
 * (src/nyx/reduce_trivial_rois.cpp:62-413): one
 * `runParallel(F::reduce | F::parallel_process_1_batch, nThr, ...)` per family,
 * and finally lays `LR::fvals` out in the column order of
 * save_features_2_buffer (src/nyx/output_2_buffer.cpp:303-584).
 *
 * Uses: (a) pin oracle/nyx_oracle.c against the reference itself on arbitrary
 * inputs; (b) the `cpu_baseline` of bench.py (kind "reference": the reference's
 * own multithreaded CPU reduce, timed on the GPU box's host cores).
 */
#include <chrono>
#include <cstring>
#include <map>
#include <vector>
#include <unordered_map>
#include <unordered_set>
#include <algorithm>

#include "roi_cache.h"
#include "dataset.h"
#include "parallel.h"
#include "features/intensity.h"
#include "features/glcm.h"
#include "features/glrlm.h"
#include "features/glszm.h"
#include "features/ngtdm.h"
#include "features/gldzm.h"
#include "features/gldm.h"
#include "features/ngldm.h"
#include "features/gabor.h"
#include "features/zernike.h"
#include "features/contour.h"
#include "features/2d_geomoments.h"

#include "../include/nyxhip.h"

using namespace Nyxus;

namespace {

Fsettings make_settings(const nyxhip_settings* s)
{
    // Environment::compile_feature_settings, src/nyx/env_features.cpp:713-736
    Fsettings f;
    f.resize((int)NyxSetting::__COUNT__);
    f[(int)NyxSetting::SOFTNAN].rval = s->soft_nan;
    f[(int)NyxSetting::TINY].rval = s->tiny;
    f[(int)NyxSetting::SINGLEROI].bval = false;
    f[(int)NyxSetting::GREYDEPTH].ival = s->grey_depth;
    f[(int)NyxSetting::PIXELSIZEUM].rval = 1.0;
    f[(int)NyxSetting::PIXELDISTANCE].ival = 5;
    f[(int)NyxSetting::USEGPU].bval = false;
    f[(int)NyxSetting::VERBOSLVL].ival = 0;
    f[(int)NyxSetting::IBSI].bval = s->ibsi != 0;
    f[(int)NyxSetting::GLCM_OFFSET].ival = s->glcm_offset;
    f[(int)NyxSetting::GLCM_GREYDEPTH].ival = s->glcm_grey_depth;
    f[(int)NyxSetting::GLCM_NUMANG].ival = s->glcm_n_angles;
    return f;
}

void apply_statics(const nyxhip_settings* s)
{
    // process-global knobs of the reference (glcm.cpp:8-9, gabor.cpp:14-25)
    GLCMFeature::angles.assign(s->glcm_angles, s->glcm_angles + s->glcm_n_angles);
    GLCMFeature::symmetric_glcm = s->glcm_symmetric != 0;
    GaborFeature::gamma = s->gabor_gamma;
    GaborFeature::sig2lam = s->gabor_sig2lam;
    GaborFeature::n = s->gabor_kersize;
    GaborFeature::f0LP = s->gabor_f0lp;
    GaborFeature::GRAYthr = s->gabor_graythr;
    GaborFeature::f0_theta_pairs.clear();
    for (int i = 0; i < s->gabor_n_filters; i++)
        GaborFeature::f0_theta_pairs.push_back({s->gabor_f0[i], s->gabor_theta[i]});
}

const Feature2D kIntensity[36] = {
    Feature2D::COV, Feature2D::COVERED_IMAGE_INTENSITY_RANGE, Feature2D::ENERGY, Feature2D::ENTROPY,
    Feature2D::EXCESS_KURTOSIS, Feature2D::HYPERFLATNESS, Feature2D::HYPERSKEWNESS,
    Feature2D::INTEGRATED_INTENSITY, Feature2D::INTERQUARTILE_RANGE, Feature2D::KURTOSIS, Feature2D::MAX,
    Feature2D::MEAN, Feature2D::MEAN_ABSOLUTE_DEVIATION, Feature2D::MEDIAN,
    Feature2D::MEDIAN_ABSOLUTE_DEVIATION, Feature2D::MIN, Feature2D::MODE, Feature2D::P01, Feature2D::P10,
    Feature2D::P25, Feature2D::P75, Feature2D::P90, Feature2D::P99, Feature2D::QCOD, Feature2D::RANGE,
    Feature2D::ROBUST_MEAN, Feature2D::ROBUST_MEAN_ABSOLUTE_DEVIATION, Feature2D::ROOT_MEAN_SQUARED,
    Feature2D::SKEWNESS, Feature2D::STANDARD_DEVIATION, Feature2D::STANDARD_DEVIATION_BIASED,
    Feature2D::STANDARD_ERROR, Feature2D::VARIANCE, Feature2D::VARIANCE_BIASED, Feature2D::UNIFORMITY,
    Feature2D::UNIFORMITY_PIU};

inline void put_angled(double*& p, const std::vector<double>& v, size_t n)
{
    for (size_t i = 0; i < n; i++)
        *p++ = i < v.size() ? v[i] : 0.0;
}

// reduce_trivial_rois_manual -> reduce_trivial_2d (reduce_trivial_rois.cpp:772-777, :62-413): one runParallel per requested family
void reduce_ladder(uint32_t mask, int n_threads, std::vector<int>& L, std::unordered_map<int, LR>& roiData, const Fsettings& fst, const Dataset& ds)
{
    size_t jobSize = L.size(), workPerThread = jobSize / (size_t)n_threads;
    if (mask & NYXHIP_FAM_INTENSITY)
        runParallel(PixelIntensityFeatures::reduce, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & (NYXHIP_FAM_SMOMS | NYXHIP_FAM_IMOMS))   // the moments' dependency, reduce_trivial_rois.cpp:98-111
        runParallel(ContourFeature::reduce, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_GLCM)
        runParallel(GLCMFeature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_GLRLM)
        runParallel(GLRLMFeature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_GLDZM)   // reduce_trivial_rois.cpp:215-220
        runParallel(GLDZMFeature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_GLSZM)
        runParallel(GLSZMFeature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_GLDM)    // :231-236
        runParallel(GLDMFeature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_NGLDM)   // :239-244
        runParallel(NGLDMfeature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_NGTDM)
        runParallel(NGTDMFeature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_GABOR)
        runParallel(GaborFeature::reduce, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_ZERNIKE)
        runParallel(ZernikeFeature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_IMOMS)   // reduce_trivial_rois.cpp:320-325
        runParallel(Imoms2D_feature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
    if (mask & NYXHIP_FAM_SMOMS)   // :326-331
        runParallel(Smoms2D_feature::parallel_process_1_batch, n_threads, workPerThread, jobSize, &L, &roiData, fst, ds);
}

// one row in the table layout of save_features_2_buffer (enum order, angle expansion); raw values
void put_row(const LR& lr, uint32_t mask, const nyxhip_settings* s, double* p)
{
    if (mask & NYXHIP_FAM_INTENSITY)
        for (auto f : kIntensity) *p++ = lr.fvals[(int)f][0];
    if (mask & NYXHIP_FAM_GLCM) {
        size_t na = (size_t)s->glcm_n_angles;
        for (int f = (int)Feature2D::GLCM_ASM; f <= (int)Feature2D::GLCM_VARIANCE; f++)
            put_angled(p, lr.fvals[f], na);
        for (int f = (int)Feature2D::GLCM_ASM_AVE; f <= (int)Feature2D::GLCM_SUMVARIANCE_AVE; f++)
            *p++ = lr.fvals[f][0];
    }
    if (mask & NYXHIP_FAM_GLRLM) {
        for (int f = (int)Feature2D::GLRLM_SRE; f <= (int)Feature2D::GLRLM_LRHGLE; f++)
            put_angled(p, lr.fvals[f], 4);
        for (int f = (int)Feature2D::GLRLM_SRE_AVE; f <= (int)Feature2D::GLRLM_LRHGLE_AVE; f++)
            *p++ = lr.fvals[f][0];
    }
    if (mask & NYXHIP_FAM_GLDZM)
        for (int f = (int)Feature2D::GLDZM_SDE; f <= (int)Feature2D::GLDZM_ZDE; f++)
            *p++ = lr.fvals[f][0];
    if (mask & NYXHIP_FAM_GLSZM)
        for (int f = (int)Feature2D::GLSZM_SAE; f <= (int)Feature2D::GLSZM_LAHGLE; f++)
            *p++ = lr.fvals[f][0];
    if (mask & NYXHIP_FAM_GLDM)
        for (int f = (int)Feature2D::GLDM_SDE; f <= (int)Feature2D::GLDM_LDHGLE; f++)
            *p++ = lr.fvals[f][0];
    if (mask & NYXHIP_FAM_NGLDM)
        for (int f = (int)Feature2D::NGLDM_LDE; f <= (int)Feature2D::NGLDM_DCENE; f++)
            *p++ = lr.fvals[f][0];
    if (mask & NYXHIP_FAM_NGTDM)
        for (int f = (int)Feature2D::NGTDM_COARSENESS; f <= (int)Feature2D::NGTDM_STRENGTH; f++)
            *p++ = lr.fvals[f][0];
    if (mask & NYXHIP_FAM_GABOR)
        put_angled(p, lr.fvals[(int)Feature2D::GABOR], (size_t)s->gabor_n_filters);
    if (mask & NYXHIP_FAM_ZERNIKE)
        put_angled(p, lr.fvals[(int)Feature2D::ZERNIKE2D], 30);
    if (mask & NYXHIP_FAM_SMOMS)
        for (int f = (int)Feature2D::SPAT_MOMENT_00; f <= (int)Feature2D::WEIGHTED_HU_M7; f++)
            *p++ = lr.fvals[f][0];
    if (mask & NYXHIP_FAM_IMOMS)
        for (int f = (int)Feature2D::IMOM_RM_00; f <= (int)Feature2D::IMOM_WHU7; f++)
            *p++ = lr.fvals[f][0];
}

} // namespace

extern "C" {

int nyxref_n_columns(uint32_t mask, const nyxhip_settings* s)
{
    int n = 0;
    if (mask & NYXHIP_FAM_INTENSITY) n += 36;
    if (mask & NYXHIP_FAM_GLCM) n += 30 * s->glcm_n_angles + 29;
    if (mask & NYXHIP_FAM_GLRLM) n += 16 * 4 + 16;
    if (mask & NYXHIP_FAM_GLDZM) n += 18;
    if (mask & NYXHIP_FAM_GLSZM) n += 16;
    if (mask & NYXHIP_FAM_GLDM) n += 14;
    if (mask & NYXHIP_FAM_NGLDM) n += 19;
    if (mask & NYXHIP_FAM_NGTDM) n += 5;
    if (mask & NYXHIP_FAM_GABOR) n += s->gabor_n_filters;
    if (mask & NYXHIP_FAM_ZERNIKE) n += 30;
    if (mask & NYXHIP_FAM_SMOMS) n += 90;
    if (mask & NYXHIP_FAM_IMOMS) n += 90;
    return n;
}

/* Returns 0 on success.  *reduce_seconds (optional) receives the wall time of the
 * runParallel ladder alone (LR construction and table copy excluded) -- the same
 * span the reference brackets with STOPWATCH in reduce_trivial_2d. */
int nyxref_featurize_batch(const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s,
                           double* out, size_t ld, int n_threads, double* reduce_seconds)
{
    if (!b || !s || !out || b->memory != NYXHIP_MEM_HOST || n_threads < 1)
        return NYXHIP_ERR_INVALID_ARG;
    try {
        apply_statics(s);
        Fsettings fst = make_settings(s);
        Dataset ds;
        std::map<std::pair<double, double>, int> slide_of;

        std::vector<int> L;
        std::unordered_map<int, LR> roiData;
        L.reserve(b->n_roi);
        roiData.reserve(b->n_roi);
        for (uint64_t r = 0; r < b->n_roi; r++) {
            int lab = (int)r + 1;   // unique key; the caller's label is informational
            L.push_back(lab);
            LR& lr = roiData[lab];
            lr.label = lab;
            uint64_t o = b->px_offset[r], n = b->px_offset[r + 1] - o;
            lr.raw_pixels.reserve(n);
            for (uint64_t i = 0; i < n; i++)
                lr.raw_pixels.push_back(Pixel2((StatsInt)b->x[o + i], (StatsInt)b->y[o + i], (PixIntens)b->inten[o + i]));
            lr.aux_area = (unsigned int)n;
            lr.aux_min = b->min_inten[r];
            lr.aux_max = b->max_inten[r];
            lr.ph_aabb.init_x(0); lr.ph_aabb.update_x((StatsInt)b->bbox_w[r] - 1);
            lr.ph_aabb.init_y(0); lr.ph_aabb.update_y((StatsInt)b->bbox_h[r] - 1);
            lr.make_nonanisotropic_aabb();
            if (b->slide_min && b->slide_max) {
                auto key = std::make_pair(b->slide_min[r], b->slide_max[r]);
                auto it = slide_of.find(key);
                if (it == slide_of.end()) {
                    SlideProps p("", "");
                    p.min_preroi_inten = key.first;
                    p.max_preroi_inten = key.second;
                    ds.dataset_props.push_back(p);
                    it = slide_of.emplace(key, (int)ds.dataset_props.size() - 1).first;
                }
                lr.slide_idx = it->second;
            } else
                lr.slide_idx = -1;
            // allocateTrivialRoisBuffers, phase2_2d.cpp:427-465
            lr.aux_image_matrix.allocate((int)b->bbox_w[r], (int)b->bbox_h[r]);
            lr.aux_image_matrix.calculate_from_pixelcloud(lr.raw_pixels, lr.aabb);
            lr.initialize_fvals();
        }

        auto t0 = std::chrono::steady_clock::now();
        reduce_ladder(mask, n_threads, L, roiData, fst, ds);
        auto t1 = std::chrono::steady_clock::now();
        if (reduce_seconds)
            *reduce_seconds = std::chrono::duration<double>(t1 - t0).count();

        for (uint64_t r = 0; r < b->n_roi; r++)
            put_row(roiData[(int)r + 1], mask, s, out + r * ld);
    } catch (const std::exception& e) {
        fprintf(stderr, "nyxref_featurize_batch: %s\n", e.what());
        return NYXHIP_ERR_HIP;
    }
    return NYXHIP_OK;
}

/* The reference's in-memory workflow for a stack of uint32 tiles, end to end on the CPU (the tile-inclusive baseline of bench.py):
 * per image pair, as featurize_montage does (workflow_pythonapi.cpp:140-182),
 *   phase 1  gatherRoisMetricsInMemory (phase1.cpp:373-409): column-major scan, feed_pixel_2_metrics (pixel_feed.cpp:19-43:
 *            hash-set / hash-map lookup per pixel, init_label_record_3 / update_label_record_3, features_calc_workflow.cpp:65-105);
 *   phase 2  scanTrivialRoisInMemory (phase2_2d.cpp:637-684): second column-major scan, binary search in the sorted batch labels,
 *            feed_pixel_2_cache_LR (pixel_feed.cpp:71-74); allocateTrivialRoisBuffers (phase2_2d.cpp:427-465);
 *   reduce   reduce_trivial_rois_manual -> the runParallel ladder (the real feature classes).
 * The two scans are restated here on the reference's own LR / Pixel2 / AABB classes (their translation units need the whole
 * Environment); the reduce is the reference's code.  Rows: tiles in order, labels ascending.  Returns the row count in *n_rows
 * (rows beyond max_rows are computed but not stored); seconds[0] = scans + buffers (serial, as in the reference), seconds[1] = reduce. */
int nyxref_featurize_tiles(const uint32_t* inten, const uint32_t* label, uint32_t W, uint32_t H, uint32_t n_tiles, uint32_t mask,
                           const nyxhip_settings* s, int n_threads, uint32_t* out_labels, uint32_t* out_tiles, double* out, size_t ld,
                           uint64_t max_rows, uint64_t* n_rows, double* seconds)
{
    if (!inten || !label || !s || n_threads < 1 || !n_rows)
        return NYXHIP_ERR_INVALID_ARG;
    try {
        apply_statics(s);
        Fsettings fst = make_settings(s);
        Dataset ds;
        double t_scan = 0, t_red = 0;
        uint64_t rows = 0;
        for (uint32_t t = 0; t < n_tiles; t++) {
            const uint32_t* I = inten + (size_t)t * W * H;
            const uint32_t* Lb = label + (size_t)t * W * H;
            auto t0 = std::chrono::steady_clock::now();
            std::unordered_set<int> uniqueLabels;
            std::unordered_map<int, LR> roiData;
            for (size_t col = 0; col < W; col++)                   // phase 1
                for (size_t row = 0; row < H; row++) {
                    const int lab = (int)Lb[row * W + col];
                    if (!lab) continue;
                    const PixIntens v = I[row * W + col];
                    if (uniqueLabels.find(lab) == uniqueLabels.end()) {
                        uniqueLabels.insert(lab);
                        LR nr(lab);
                        nr.slide_idx = -1;                          // montage: no slide properties (slideprops.cpp:27-28)
                        nr.aux_area = 1; nr.aux_min = nr.aux_max = v; nr.init_aabb((StatsInt)col, (StatsInt)row);
                        roiData[lab] = nr;
                    } else {
                        LR& r = roiData[lab];
                        r.aux_area++; r.aux_min = std::min(r.aux_min, v); r.aux_max = std::max(r.aux_max, v); r.update_aabb((StatsInt)col, (StatsInt)row);
                    }
                }
            std::vector<int> L(uniqueLabels.begin(), uniqueLabels.end());
            std::sort(L.begin(), L.end());
            for (size_t col = 0; col < W; col++)                   // phase 2
                for (size_t row = 0; row < H; row++) {
                    const int lab = (int)Lb[row * W + col];
                    if (!lab) continue;
                    if (!std::binary_search(L.begin(), L.end(), lab)) continue;
                    roiData[lab].raw_pixels.push_back(Pixel2((StatsInt)col, (StatsInt)row, I[row * W + col]));
                }
            for (int lab : L) {                                    // allocateTrivialRoisBuffers
                LR& r = roiData[lab];
                r.make_nonanisotropic_aabb();
                r.aux_image_matrix.allocate(r.aabb.get_width(), r.aabb.get_height());
                r.aux_image_matrix.calculate_from_pixelcloud(r.raw_pixels, r.aabb);
                r.initialize_fvals();
            }
            auto t1 = std::chrono::steady_clock::now();
            reduce_ladder(mask, n_threads, L, roiData, fst, ds);
            auto t2 = std::chrono::steady_clock::now();
            t_scan += std::chrono::duration<double>(t1 - t0).count();
            t_red += std::chrono::duration<double>(t2 - t1).count();
            for (int lab : L) {
                if (rows < max_rows && out) {
                    put_row(roiData[lab], mask, s, out + rows * ld);
                    if (out_labels) out_labels[rows] = (uint32_t)lab;
                    if (out_tiles) out_tiles[rows] = t;
                }
                rows++;
            }
        }
        *n_rows = rows;
        if (seconds) { seconds[0] = t_scan; seconds[1] = t_red; }
    } catch (const std::exception& e) {
        fprintf(stderr, "nyxref_featurize_tiles: %s\n", e.what());
        return NYXHIP_ERR_HIP;
    }
    return NYXHIP_OK;
}

} // extern "C"
