/*
 * nyx_oracle.h -- TEST INFRASTRUCTURE (see nyx_oracle.c header).
 * Entry points of the plain-C CPU restatement of the reference hot path.
 */
#ifndef NYX_ORACLE_H
#define NYX_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/nyxhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* features/intensity.cpp:57-192 -> out[36] in Feature2D order */
void nyxo_intensity(const uint32_t* inten, uint64_t n_px, uint32_t aux_min, uint32_t aux_max,
                    int has_slide, double slide_min, double slide_max, int n_grey_bins,
                    double* out);

/* features/texture_feature.h:77-98 bin_pixel */
uint32_t nyxo_bin_pixel(uint32_t x, uint32_t mn, uint32_t mx, int greybin_info);

/* features/image_matrix.h:284-304; caller frees */
uint32_t* nyxo_dense_from_cloud(const uint16_t* x, const uint16_t* y, const uint32_t* inten,
                                uint64_t n_px, uint32_t w, uint32_t h);

/* features/glcm.cpp:16-100,143-208 -> out[30*n_angles + 29] */
void nyxo_glcm(const uint32_t* im, uint32_t w, uint32_t h, uint32_t aux_min, uint32_t aux_max,
               const nyxhip_settings* s, double* out);

/* features/glrlm.cpp:20-276 -> out[16*4 + 16]; glszm.cpp:56-340 -> out[16]; ngtdm.cpp:33-345 -> out[5] */
void nyxo_glrlm(const uint32_t* im, uint32_t w, uint32_t h, uint32_t aux_min, uint32_t aux_max,
                const nyxhip_settings* s, double* out);
void nyxo_glszm(const uint32_t* im, uint32_t w, uint32_t h, uint32_t aux_min, uint32_t aux_max,
                const nyxhip_settings* s, double* out);
void nyxo_ngtdm(const uint32_t* im, uint32_t w, uint32_t h, uint32_t aux_min, uint32_t aux_max,
                const nyxhip_settings* s, double* out);

/* features/gabor.cpp:43-123, :333-510 -> out[gabor_n_filters]; zernike.cpp:176-363 -> out[30] */
void nyxo_gabor_kernel(double* Gex, double f0, double sig2lam, double gamma, double theta, double fi, int n);
void nyxo_gabor(const uint32_t* im, uint32_t w, uint32_t h, uint32_t aux_min, uint32_t aux_max,
                const nyxhip_settings* s, double* out);
void nyxo_zernike(const uint32_t* im, uint32_t w, uint32_t h, uint32_t aux_min, uint32_t aux_max,
                  const nyxhip_settings* s, double* out);

/* features/gldm.cpp:16-255 -> out[14]; ngldm.cpp:40-340 -> out[19] (mask: 1 = pixel of the ROI cloud);
 * gldzm.cpp:53-420 -> out[18] (roi_area = LR::aux_area) */
void nyxo_gldm(const uint32_t* im, uint32_t w, uint32_t h, uint32_t aux_min, uint32_t aux_max,
               const nyxhip_settings* s, double* out);
void nyxo_ngldm(const uint32_t* im, const uint8_t* mask, uint32_t w, uint32_t h, uint32_t aux_min, uint32_t aux_max,
                const nyxhip_settings* s, double* out);
void nyxo_gldzm(const uint32_t* im, uint32_t w, uint32_t h, uint32_t aux_min, uint32_t aux_max, uint32_t roi_area,
                const nyxhip_settings* s, double* out);

/* contour.cpp:306-678 + 2d_geomoments_basic.cpp:32-376: shape (Smoms2D) and intensity (Imoms2D) moments, 90 each */
void nyxo_geomoments(const uint16_t* x, const uint16_t* y, const uint32_t* inten, uint64_t n, uint32_t w, uint32_t h,
                     double* out_shape, double* out_inten);
int nyxo_contour(const uint16_t* x, const uint16_t* y, const uint32_t* inten, uint64_t n, uint32_t w, uint32_t h, int32_t* out_xy, int cap);

int nyxo_n_columns(uint32_t mask, const nyxhip_settings* s);

/* Host-memory batch, same argument meaning as nyxhip_featurize_batch. */
int nyxo_featurize_batch(const nyxhip_batch* b, uint32_t mask, const nyxhip_settings* s,
                         double* out, size_t ld);

#ifdef __cplusplus
}
#endif
#endif
