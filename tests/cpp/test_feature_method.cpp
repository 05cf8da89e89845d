// C++ unit tests of the plugin adapter (include/nyxhip_feature_method.hpp) written the way the reference
// writes its own feature tests (/root/reference/tests/test_2d_firstorder_common.h:15-41,
// test_2d_glcm_regression.h:70-227, test_2d_gabor_skimage.cc:19-66): build an LR from a pixel list, run
// `F f; f.calculate(roi, settings); f.save_value(roi.fvals)`, compare with the golden tables through
// agrees_gt().  fixture.inc is generated from tests/golden/reference_tests.json by the pytest wrapper.
#include <cmath>
#include <cstdio>
#include <utility>
#include <vector>

#include "nyxhip_feature_method.hpp"

using namespace NyxusHip;

struct NyxusPixel { size_t x, y; unsigned int intensity; };
struct Golden { Feature2D f; const char* name; double v; };
struct ImageData { size_t x, y; std::vector<unsigned int> pixels; };
#include "fixture.inc"

static int failures = 0;
static bool agrees_gt(double fval, double ground_truth, double frac_tolerance = 1000.)
{   // tests/test_main_nyxus.h:13-24
    double diff = fval - ground_truth, tolerance = ground_truth / frac_tolerance;
    return std::abs(diff) <= std::abs(tolerance);
}
#define CHECK(cond, ...) do { if (!(cond)) { failures++; std::printf("FAIL %s:%d ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static void load_test_roi_data(LR& r, const NyxusPixel* px, size_t n) { for (size_t i = 0; i < n; i++) r.feed_pixel((StatsInt)px[i].x, (StatsInt)px[i].y, px[i].intensity); }
static void load_masked_test_roi_data(LR& r, const NyxusPixel* I, const NyxusPixel* M, size_t n)
{   // tests/test_main_nyxus.h:48-88
    for (size_t i = 0; i < n; i++) if (M[i].intensity != 0) r.feed_pixel((StatsInt)I[i].x, (StatsInt)I[i].y, I[i].intensity);
}

static void test_firstorder_matlab()
{
    Dataset ds; ds.dataset_props.push_back(SlideProps());
    ds.dataset_props[0].min_preroi_inten = 0.0; ds.dataset_props[0].max_preroi_inten = 65535.0;
    LR roidata(100); roidata.slide_idx = 0;
    load_test_roi_data(roidata, pixelIntensityFeaturesTestData, sizeof(pixelIntensityFeaturesTestData) / sizeof(NyxusPixel));
    PixelIntensityFeatures f;
    f.calculate(roidata, Fsettings(), ds);          // default settings -> 24 histogram bins (intensity.cpp:125)
    roidata.initialize_fvals();
    f.save_value(roidata.fvals);
    for (const Golden& g : firstorder_2d_matlab_ref_vals) {
        if (g.f == Feature2D::UNIFORMITY) continue;   // matched at GREYDEPTH=20 below
        CHECK(agrees_gt(roidata.fvals[(int)g.f][0], g.v), "%s got %.17g want %.17g", g.name, roidata.fvals[(int)g.f][0], g.v);
    }
    Fsettings s; s.resize((int)NyxSetting::__COUNT__);
    s[(int)NyxSetting::GREYDEPTH].ival = 20; s[(int)NyxSetting::GLCM_GREYDEPTH].ival = 20; s[(int)NyxSetting::GLCM_OFFSET].ival = 1;
    LR r2(100); load_test_roi_data(r2, pixelIntensityFeaturesTestData, sizeof(pixelIntensityFeaturesTestData) / sizeof(NyxusPixel));
    PixelIntensityFeatures f2; f2.calculate(r2, s); r2.initialize_fvals(); f2.save_value(r2.fvals);
    for (const Golden& g : firstorder_2d_matlab_ref_vals)
        if (g.f == Feature2D::UNIFORMITY) CHECK(agrees_gt(r2.fvals[(int)g.f][0], g.v, 100.), "UNIFORMITY got %.17g want %.17g", r2.fvals[(int)g.f][0], g.v);
}

static void test_glcm_regression()
{   // tests/test_2d_glcm_regression.h:70-227: matlab binning, 100 levels, asymmetric, mean over 4 slices x 4 angles
    Fsettings s; s.resize((int)NyxSetting::__COUNT__);
    s[(int)NyxSetting::SOFTNAN].rval = 0.0; s[(int)NyxSetting::GREYDEPTH].ival = 100; s[(int)NyxSetting::IBSI].bval = false;
    s[(int)NyxSetting::GLCM_GREYDEPTH].ival = 100; s[(int)NyxSetting::GLCM_OFFSET].ival = 1;
    knobs().symmetric_glcm = false; knobs().glcm_angles = {0, 45, 90, 135};
    const NyxusPixel* I[4] = {ibsi_phantom_z1_intensity, ibsi_phantom_z2_intensity, ibsi_phantom_z3_intensity, ibsi_phantom_z4_intensity};
    const NyxusPixel* M[4] = {ibsi_phantom_z1_mask, ibsi_phantom_z2_mask, ibsi_phantom_z3_mask, ibsi_phantom_z4_mask};
    std::vector<LR> rois(4);
    for (int z = 0; z < 4; z++) {
        load_masked_test_roi_data(rois[z], I[z], M[z], 20);
        GLCMFeature f; f.calculate(rois[z], s); rois[z].initialize_fvals(); f.save_value(rois[z].fvals);
    }
    for (const Golden& g : glcm_2d_regression_ref_vals) {
        double total = 0;
        for (int z = 0; z < 4; z++) for (int a = 0; a < 4; a++) total += rois[z].fvals[(int)g.f][a];
        CHECK(agrees_gt(total / 16.0, g.v, 100.), "%s got %.17g want %.17g", g.name, total / 16.0, g.v);
    }
}

static void test_gabor_truth()
{   // tests/test_2d_gabor_skimage.cc:19-66
    for (size_t i = 0; i < dsb_data.size(); ++i) {
        LR roidata; size_t w = dsb_data[i].x;
        for (size_t k = 0; k < dsb_data[i].pixels.size(); k++) roidata.feed_pixel((StatsInt)(k % w), (StatsInt)(k / w), dsb_data[i].pixels[k]);
        roidata.initialize_fvals();
        GaborFeature f; f.calculate(roidata, Fsettings()); f.save_value(roidata.fvals);
        CHECK(roidata.fvals[(int)Feature2D::GABOR].size() == gabor_truth[i].size(), "gabor width");
        for (size_t j = 0; j < gabor_truth[i].size(); ++j)
            CHECK(agrees_gt(gabor_truth[i][j], roidata.fvals[(int)Feature2D::GABOR][j]), "GABOR roi %zu filter %zu got %.17g want %.17g", i, j,
                  roidata.fvals[(int)Feature2D::GABOR][j], gabor_truth[i][j]);
    }
}

template <class F>
static void phantom_average_check(const std::vector<Golden>& gold, double frac_tol, const char* tag)
{   // tests/test_2d_ngldm_common.h:47-139, test_2d_gldm_ibsi.h:34-119, test_2d_gldzm_ibsi.h:92-183: the four IBSI phantom
    // slices featurised one at a time (GREYDEPTH 128, IBSI mode) and averaged
    Fsettings s; s.resize((int)NyxSetting::__COUNT__);
    s[(int)NyxSetting::SOFTNAN].rval = 0.0; s[(int)NyxSetting::GREYDEPTH].ival = 128; s[(int)NyxSetting::IBSI].bval = true;
    s[(int)NyxSetting::GLCM_GREYDEPTH].ival = 128; s[(int)NyxSetting::GLCM_OFFSET].ival = 1;
    const NyxusPixel* I[4] = {ibsi_phantom_z1_intensity, ibsi_phantom_z2_intensity, ibsi_phantom_z3_intensity, ibsi_phantom_z4_intensity};
    const NyxusPixel* M[4] = {ibsi_phantom_z1_mask, ibsi_phantom_z2_mask, ibsi_phantom_z3_mask, ibsi_phantom_z4_mask};
    std::vector<LR> rois(4);
    for (int z = 0; z < 4; z++) {
        load_masked_test_roi_data(rois[z], I[z], M[z], 20);
        F f; f.calculate(rois[z], s); rois[z].initialize_fvals(); f.save_value(rois[z].fvals);
    }
    for (const Golden& g : gold) {
        double total = 0;
        for (int z = 0; z < 4; z++) total += rois[z].fvals[(int)g.f][0];
        CHECK(agrees_gt(total / 4.0, g.v, frac_tol), "%s %s got %.17g want %.17g", tag, g.name, total / 4.0, g.v);
    }
}

static void test_dependence_families()
{
    phantom_average_check<GLDMFeature>(gldm_2d_ibsi_ref_vals, 100., "GLDM/IBSI");
    phantom_average_check<NGLDMfeature>(ngldm_2d_ibsi_ref_vals, 100., "NGLDM/IBSI");
    phantom_average_check<NGLDMfeature>(ngldm_2d_mirp_ref_vals, 1e9, "NGLDM/mirp");          // test_2d_ngldm_mirp.h:50-58
    phantom_average_check<NGLDMfeature>(ngldm_2d_regression_ref_vals, 1e9, "NGLDM/regression");
    phantom_average_check<GLDZMFeature>(gldzm_2d_ibsi_ref_vals, 2., "GLDZM/IBSI");           // test_2d_gldzm_ibsi.h:183
}

static void test_reduce_trivial_rois_manual()
{   // the boundary itself: labels + roiData + FeatureSet in, fvals filled for every required family in one call
    std::unordered_map<int, LR> roiData;
    std::vector<int> L{7, 3};
    for (int l : L) { roiData[l] = LR(l); load_masked_test_roi_data(roiData[l], ibsi_phantom_z1_intensity, l == 7 ? ibsi_phantom_z1_mask : ibsi_phantom_z3_mask, 20); roiData[l].initialize_fvals(); }
    FeatureSet fs; fs.enableFeatures({Feature2D::MEAN, Feature2D::GLCM_ASM, Feature2D::NGTDM_COARSENESS});
    Fsettings s; s.resize((int)NyxSetting::__COUNT__);
    s[(int)NyxSetting::GREYDEPTH].ival = 8; s[(int)NyxSetting::GLCM_GREYDEPTH].ival = 8; s[(int)NyxSetting::GLCM_OFFSET].ival = 1;
    reduce_trivial_rois_manual(L, roiData, fs, s, Dataset());
    for (int l : L) {
        LR single(l); load_masked_test_roi_data(single, ibsi_phantom_z1_intensity, l == 7 ? ibsi_phantom_z1_mask : ibsi_phantom_z3_mask, 20);
        PixelIntensityFeatures::extract(single, s); GLCMFeature::extract(single, s); NGTDMFeature::extract(single, s);
        CHECK(roiData[l].fvals[(int)Feature2D::MEAN][0] == single.fvals[(int)Feature2D::MEAN][0], "MEAN batch vs single");
        CHECK(roiData[l].fvals[(int)Feature2D::GLCM_ASM] == single.fvals[(int)Feature2D::GLCM_ASM], "GLCM_ASM batch vs single");
        CHECK(roiData[l].fvals[(int)Feature2D::GLCM_ASM].size() == 4, "angled width");
        CHECK(roiData[l].fvals[(int)Feature2D::NGTDM_COARSENESS] == single.fvals[(int)Feature2D::NGTDM_COARSENESS], "NGTDM batch vs single");
        CHECK(roiData[l].fvals[(int)Feature2D::GLRLM_SRE][0] == 0.0, "family not required stays untouched");
    }
    // runParallel with the functype signature
    runParallel(GLRLMFeature::parallel_process_1_batch, 4, L.size() / 4, L.size(), &L, &roiData, s, Dataset());
    CHECK(roiData[7].fvals[(int)Feature2D::GLRLM_SRE].size() == 4 && roiData[7].fvals[(int)Feature2D::GLRLM_SRE][0] > 0, "runParallel(GLRLM)");
    bool threw = false;
    try { ZernikeFeature z; z.osized_calculate(roiData[7], s); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw, "osized_calculate must throw");
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::string(argv[1]) == "--compile-check") { std::printf("compiled\n"); return 0; }
    try {
        test_firstorder_matlab();
        test_glcm_regression();
        test_gabor_truth();
        test_dependence_families();
        test_reduce_trivial_rois_manual();
    } catch (const std::exception& e) { std::printf("EXCEPTION: %s\n", e.what()); return 2; }
    std::printf(failures ? "FAILED %d checks\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
