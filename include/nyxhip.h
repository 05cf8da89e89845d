/*
 * nyxhip.h -- C ABI of the MI355X-native per-ROI feature reducer.
 *
 * This is the drop-in boundary for the reference's ROI-batch reduce step:
 *
 *   reduce_trivial_rois_manual(std::vector<int>& Pending, Environment& env)
 *       /root/reference/src/nyx/reduce_trivial_rois.cpp:772-795
 *   reduce_trivial_2d(...)                       reduce_trivial_rois.cpp:62-413
 *   runParallel(F::reduce, nThr, ...)            src/nyx/parallel.h:23-42
 *
 * i.e. "given a batch of in-RAM ROIs (pixel cloud + bounding box + min/max),
 * fill every ROI's feature values for the enabled feature families".
 * The reference keeps ROIs as `LR` records (src/nyx/roi_cache.h:31-84) holding
 * an AoS `Pixel2{long x; long y; uint inten}` cloud (features/pixel.h:52-61) and a
 * dense bounding-box matrix (features/image_matrix.h:284-304); here the same
 * information crosses the boundary as flat SoA arrays (plain pointers + sizes).
 *
 * Everything below is plain C: no C++ types, no torch types.  All calls return
 * an int status (NYXHIP_OK == 0); nyxhip_last_error() gives the message.
 * Degenerate ROIs are not errors: they produce the reference's soft-NaN
 * sentinel values (reference: glcm.cpp:27-95, intensity.cpp:121-122).
 */
#ifndef NYXHIP_H
#define NYXHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NYXHIP_ABI_VERSION 2   /* 2: tile ABI v2 (nyxhip_tiles), sharded entry; every v1 symbol is kept */

/* ---- status codes ------------------------------------------------------- */
enum {
    NYXHIP_OK = 0,
    NYXHIP_ERR_INVALID_ARG = 1,   /* null pointer, bad size, bad mask          */
    NYXHIP_ERR_NO_DEVICE = 2,     /* HIP runtime / GPU not available           */
    NYXHIP_ERR_HIP = 3,           /* a HIP call failed (message has detail)    */
    NYXHIP_ERR_UNSUPPORTED = 4,   /* setting outside what the kernels cover    */
    NYXHIP_ERR_ROI_TOO_LARGE = 5  /* ROI exceeds the LDS-resident capacity     */
};

/* ---- feature families (bitmask) -----------------------------------------
 * One bit per reference FeatureMethod class on the hot path; replaces the
 * `if (F::required(fs)) runParallel(F::reduce ...)` ladder of
 * reduce_trivial_rois.cpp:65-70 (intensity), :199-204 (GLCM), :207-212 (GLRLM),
 * :223-228 (GLSZM), :247-252 (NGTDM), :364-369 (Gabor), :373-378 (Zernike). */
enum {
    NYXHIP_FAM_INTENSITY = 1u << 0, /* PixelIntensityFeatures, 36 columns      */
    NYXHIP_FAM_GLCM      = 1u << 1, /* GLCMFeature, 30*n_angles + 29 columns   */
    NYXHIP_FAM_GLRLM     = 1u << 2, /* GLRLMFeature, 16*4 + 16 columns         */
    NYXHIP_FAM_GLSZM     = 1u << 3, /* GLSZMFeature, 16 columns                */
    NYXHIP_FAM_NGTDM     = 1u << 4, /* NGTDMFeature, 5 columns                 */
    NYXHIP_FAM_GABOR     = 1u << 5, /* GaborFeature, n_gabor_filters columns   */
    NYXHIP_FAM_ZERNIKE   = 1u << 6, /* ZernikeFeature, 30 columns              */
    /* the remaining dependence / distance-zone texture families */
    NYXHIP_FAM_GLDZM     = 1u << 7, /* GLDZMFeature, 18 columns (features/gldzm.h:18-38)  */
    NYXHIP_FAM_GLDM      = 1u << 8, /* GLDMFeature, 14 columns  (features/gldm.h:30-46)   */
    NYXHIP_FAM_NGLDM     = 1u << 9, /* NGLDMfeature, 19 columns (features/ngldm.h:17-38)  */
    NYXHIP_FAM_SMOMS     = 1u << 10, /* Smoms2D_feature, 90 columns: shape moments (features/2d_geomoments.h:247-340)      */
    NYXHIP_FAM_IMOMS     = 1u << 11, /* Imoms2D_feature, 90 columns: intensity moments (features/2d_geomoments.h:99-193)   */
    NYXHIP_FAM_NORTH_STAR = 0x7Fu,  /* the seven families of BASELINE.json's north_star */
    NYXHIP_FAM_ALL       = 0xFFFu
};

#define NYXHIP_MAX_GLCM_ANGLES 4
#define NYXHIP_MAX_GABOR_FILTERS 16

/* ---- settings ------------------------------------------------------------
 * Mirrors the slots of `Fsettings` indexed by `NyxSetting`
 * (src/nyx/feature_settings.h:17-54) as filled by
 * Environment::compile_feature_settings (src/nyx/env_features.cpp:713-736),
 * plus the class-static knobs the reference keeps process-global:
 * GLCMFeature::angles / symmetric_glcm (features/glcm.cpp:8-9) and the Gabor
 * bank (features/gabor.cpp:14-25). */
typedef struct nyxhip_settings {
    double soft_nan;          /* NyxSetting::SOFTNAN  (default 0.0)            */
    double tiny;              /* NyxSetting::TINY     (default 1e-10)          */
    int32_t grey_depth;       /* NyxSetting::GREYDEPTH: >0 matlab binning with
                                 that many levels, <0 radiomics binning with
                                 |n| bins (texture_feature.h:100-102); also
                                 the intensity-histogram bin count
                                 (intensity.cpp:125)                           */
    int32_t ibsi;             /* NyxSetting::IBSI: no binning when non-zero    */
    int32_t glcm_grey_depth;  /* NyxSetting::GLCM_GREYDEPTH (degeneracy guard
                                 only, glcm.cpp:23-29)                         */
    int32_t glcm_offset;      /* NyxSetting::GLCM_OFFSET (default 1)           */
    int32_t glcm_n_angles;    /* GLCMFeature::angles.size()                    */
    int32_t glcm_angles[NYXHIP_MAX_GLCM_ANGLES]; /* subset of {0,45,90,135}    */
    int32_t glcm_symmetric;   /* GLCMFeature::symmetric_glcm                   */
    /* Gabor bank (features/gabor.cpp:14-25) */
    double gabor_gamma, gabor_sig2lam, gabor_f0lp, gabor_graythr;
    int32_t gabor_kersize;
    int32_t gabor_n_filters;
    double gabor_f0[NYXHIP_MAX_GABOR_FILTERS];
    double gabor_theta[NYXHIP_MAX_GABOR_FILTERS];   /* radians */
} nyxhip_settings;

/* Fills `s` with the reference defaults (environment.cpp / env_features.cpp:
 * SOFTNAN 0.0, TINY 1e-10, grey depth 64, IBSI off, GLCM offset 1, angles
 * {0,45,90,135}, asymmetric; Gabor gamma .1, sig2lam .8, n 16, f0LP .1,
 * thr .025, bank {(4,0),(16,pi/4),(32,pi/2),(64,3pi/4)}). */
void nyxhip_default_settings(nyxhip_settings* s);

/* ---- a batch of ROIs -------------------------------------------------------
 * SoA restatement of `std::vector<int> labels` + `unordered_map<int,LR>`
 * (the two containers `functype` receives, parallel.h:13).  ROI r owns pixels
 * [px_offset[r], px_offset[r+1]).  Coordinates are relative to the ROI's
 * bounding-box origin (LR::aabb), so they fit uint16 for every in-RAM
 * ("trivial") ROI.  Pixel order inside an ROI is the scan order of the caller
 * (column-major for the in-memory API, phase2_2d.cpp:655-656); no kernel
 * result depends on it beyond floating-point summation order.
 *
 * `memory` says where ALL pointers of the struct live. */
enum { NYXHIP_MEM_HOST = 0, NYXHIP_MEM_DEVICE = 1,
       /* nyxhip_tiles only: host memory AND the caller's statement that both tile arrays are mappings of their own (mmap, a
        * page-aligned allocation that is not handed back to an allocator arena while the call runs).  Their whole pages are
        * then registered for the call (hipHostRegister) and copied by DMA in place; plain NYXHIP_MEM_HOST goes through the
        * library's own pinned staging ring and assumes nothing about the caller's allocator ("Host memory"). */
       NYXHIP_MEM_HOST_OWN_MAPPING = 2 };

typedef struct nyxhip_batch {
    uint64_t n_roi;
    const uint32_t* roi_label;   /* [n_roi]   LR::label (informational)        */
    const uint64_t* px_offset;   /* [n_roi+1] CSR offsets into x/y/inten       */
    const uint16_t* x;           /* [n_px]    Pixel2::x - aabb.xmin            */
    const uint16_t* y;           /* [n_px]    Pixel2::y - aabb.ymin            */
    const uint32_t* inten;       /* [n_px]    Pixel2::inten                    */
    const uint32_t* bbox_w;      /* [n_roi]   aabb width                       */
    const uint32_t* bbox_h;      /* [n_roi]   aabb height                      */
    const uint32_t* min_inten;   /* [n_roi]   LR::aux_min                      */
    const uint32_t* max_inten;   /* [n_roi]   LR::aux_max                      */
    const double* slide_min;     /* [n_roi] or NULL: SlideProps::min_preroi_inten
                                    of the ROI's slide (intensity.cpp:72-77);
                                    NULL = slide_idx < 0 (feature left at 0)   */
    const double* slide_max;     /* [n_roi] or NULL                            */
    int32_t memory;              /* NYXHIP_MEM_HOST | NYXHIP_MEM_DEVICE        */
    /* Batch extrema, used to size the per-workgroup LDS carve-out.  The reference
     * tracks the same quantities per dataset (Dataset::dataset_max_roi_area / _w /
     * _h, src/nyx/dataset.h:12-18).  0 = unknown: the library derives them (for a
     * device batch that costs one small device->host copy and a stream sync). */
    uint32_t max_px;             /* max over ROIs of px_offset[r+1]-px_offset[r] */
    uint32_t max_bbox_area;      /* max over ROIs of bbox_w[r]*bbox_h[r]        */
    uint32_t max_inten_range;    /* max over ROIs of max_inten[r]-min_inten[r]     */
    uint32_t max_bbox_side;      /* max over ROIs of max(bbox_w[r], bbox_h[r])     */
                                 /* (all four are derived when max_px == 0)        */
} nyxhip_batch;

typedef struct nyxhip_ctx nyxhip_ctx;

/* ---- lifecycle ------------------------------------------------------------ */
int nyxhip_abi_version(void);

/* Binds a context to one GPU (one context per GPU / per process rank).  Fails
 * with NYXHIP_ERR_NO_DEVICE when no HIP device exists: there is no CPU
 * fallback behind this ABI. */
int nyxhip_init(int device, nyxhip_ctx** out_ctx);
void nyxhip_destroy(nyxhip_ctx* ctx);
const char* nyxhip_last_error(const nyxhip_ctx* ctx);

/* Launch on a caller-owned hipStream_t (e.g. torch's current stream) instead of
 * the context's own stream.  NULL restores the context stream. */
int nyxhip_set_stream(nyxhip_ctx* ctx, void* hip_stream);

/* ---- output-table layout ----------------------------------------------------
 * Columns follow `Feature2D` enum order with the angle / index expansion of
 * save_features_2_buffer (src/nyx/output_2_buffer.cpp:303-584): angled GLCM and
 * GLRLM features expand to `_0,_45,_90,_135`, GABOR to `_i`, ZERNIKE2D to `_Z i`.
 * Column names are the reference's user-facing feature names. */
int nyxhip_n_columns(uint32_t family_mask, const nyxhip_settings* s);
/* Writes the NUL-terminated name of column `col` into buf; returns NYXHIP_OK or
 * NYXHIP_ERR_INVALID_ARG. */
int nyxhip_column_name(uint32_t family_mask, const nyxhip_settings* s, int col,
                       char* buf, size_t buf_len);

/* ---- the hot path -----------------------------------------------------------
 * Replaces reduce_trivial_rois_manual() for the families in `family_mask`.
 * out_table is row-major [n_roi x out_ld] doubles (out_ld >= n_columns), in host
 * memory when batch->memory == NYXHIP_MEM_HOST, in device memory otherwise.
 * Synchronous: results are complete on return.  NaN/inf are NOT replaced here;
 * that is the table writer's job in the reference too (force_finite_number,
 * helpers/helpers.h:376-382) -- see nyxhip_finalize_table(). */
int nyxhip_featurize_batch(nyxhip_ctx* ctx, const nyxhip_batch* batch,
                           uint32_t family_mask, const nyxhip_settings* s,
                           double* out_table, size_t out_ld);

/* Asynchronous form for device-resident batches: enqueues the kernels on the
 * context's stream and returns; call nyxhip_sync() (or synchronise the stream
 * given to nyxhip_set_stream) before reading out_table. */
int nyxhip_featurize_batch_async(nyxhip_ctx* ctx, const nyxhip_batch* batch,
                                 uint32_t family_mask, const nyxhip_settings* s,
                                 double* out_table, size_t out_ld);
int nyxhip_sync(nyxhip_ctx* ctx);

/* Host-side NaN/inf -> soft_nan replacement over a host table, as
 * save_features_2_buffer does per value (output_2_buffer.cpp:296,...). */
void nyxhip_finalize_table(double* table, size_t n_rows, size_t n_cols, size_t ld,
                           double soft_nan);

/* ---- fused tile path ("next" row: phases 1-2 + reduce on the device) ---------
 * Replaces gatherRoisMetricsInMemory (src/nyx/phase1.cpp:373-409) +
 * scanTrivialRoisInMemory (src/nyx/phase2_2d.cpp:637-684) +
 * allocateTrivialRoisBuffers (:427-465) + the RAM batching of
 * processTrivialRoisInMemory (:686-770) + reduce_trivial_rois_manual for a stack
 * of equally sized intensity/label tile pairs in device or host memory.
 *
 * Tile t occupies rows [t*height, (t+1)*height) of one tall image.  Labels are
 * per tile and may be ANY non-zero 32-bit values (the reference keys ROIs through
 * hash containers, roi_cache.h / phase1.cpp:373-409): nothing is sized by label
 * magnitude.  Rows come back ordered by (tile, label) -- images in input order,
 * labels ascending, the row order of the reference's table
 * (workflow_pythonapi.cpp:140-182 + output_2_buffer.cpp:305-306).
 *
 * Tiles keep the caller's unsigned element type (the reference casts everything to
 * uint32 on the host, nyxus.py:486-489; here the cast happens in the kernels, so
 * H2D carries the image's own bytes). */
enum { NYXHIP_U8 = 1, NYXHIP_U16 = 2, NYXHIP_U32 = 4 };   /* element size in bytes */

/* Slide min / max behind COVERED_IMAGE_INTENSITY_RANGE (intensity.cpp:72-77):
 * MONTAGE  = the in-memory API: the montage prescan leaves them at +/-DBL_MAX
 *            (slideprops.cpp:27-28,74-75), the feature is range / -inf = -0.0;
 * PER_TILE = one tile is one slide: min / max of the intensities under any mask
 *            (scan_slide_props, slideprops.cpp:456-...), computed on the device;
 * GIVEN    = per-tile values from the caller (a slide cut into several tiles). */
enum { NYXHIP_SLIDE_MONTAGE = 0, NYXHIP_SLIDE_PER_TILE = 1, NYXHIP_SLIDE_GIVEN = 2 };

typedef struct nyxhip_tiles {
    const void* inten;          /* [n_tiles][height][width] of inten_dtype             */
    const void* label;          /* [n_tiles][height][width] of label_dtype, 0 = no ROI */
    int32_t inten_dtype;        /* NYXHIP_U8 | NYXHIP_U16 | NYXHIP_U32                 */
    int32_t label_dtype;
    uint32_t width, height, n_tiles;
    int32_t memory;             /* NYXHIP_MEM_HOST | NYXHIP_MEM_DEVICE (tiles AND outputs) */
    int32_t slide_mode;         /* NYXHIP_SLIDE_*                                      */
    const double* slide_min;    /* host [n_tiles], NYXHIP_SLIDE_GIVEN only             */
    const double* slide_max;
    uint64_t max_device_bytes;  /* device workspace budget of the call -- the counterpart of the
                                   reference's ram_limit batching (phase2_2d.cpp:694-705): the stack
                                   is processed in chunks of tiles that fit; 0 = half of the free
                                   device memory                                       */
} nyxhip_tiles;

/* Host tiles: chunk c+1 is copied to the device while chunk c is being reduced (two device
 * staging slots, copies and kernels on different streams).
 * Outputs (host memory when tiles->memory == NYXHIP_MEM_HOST, else device memory):
 * out_labels / out_tile_index (may be NULL) / out_table hold up to max_rows rows.  When the
 * stack has more ROIs, NYXHIP_ERR_INVALID_ARG is returned and *n_roi_out holds the count.
 * Host memory only: out_table == NULL asks the library to keep the result in the context;
 * *n_roi_out then sizes the caller's buffers for nyxhip_fetch_result(). */
int nyxhip_featurize_tiles_v2(nyxhip_ctx* ctx, const nyxhip_tiles* tiles, uint32_t family_mask,
                              const nyxhip_settings* s, uint32_t* out_labels, uint32_t* out_tile_index,
                              uint64_t max_rows, double* out_table, size_t out_ld, uint64_t* n_roi_out);
/* Copies the result kept by the last nyxhip_featurize_tiles_v2(out_table == NULL) call and frees it. */
int nyxhip_fetch_result(nyxhip_ctx* ctx, uint32_t* out_labels, uint32_t* out_tile_index,
                        double* out_table, size_t out_ld);

/* One node, several GPUs: the stack (HOST memory) is block-partitioned over the contexts
 * (context g gets tiles [g*n/G, (g+1)*n/G) -- ROIs are independent, the label vector is
 * already sliced like this by the reference's runParallel, parallel.h:34-41), each context
 * is driven by its own host thread, and the rows are returned in stack order as above.
 * Contexts may sit on the same or on different devices.  out_table == NULL: see
 * nyxhip_fetch_result_sharded(). */
int nyxhip_featurize_tiles_sharded(nyxhip_ctx* const* ctxs, int n_ctx, const nyxhip_tiles* tiles,
                                   uint32_t family_mask, const nyxhip_settings* s,
                                   uint32_t* out_labels, uint32_t* out_tile_index, uint64_t max_rows,
                                   double* out_table, size_t out_ld, uint64_t* n_roi_out);
/* out_table == NULL above keeps every context's rows in the context; this copies them out in
 * stack order (sum of the counts = the *n_roi_out of that call) and frees them. */
int nyxhip_fetch_result_sharded(nyxhip_ctx* const* ctxs, int n_ctx, uint32_t* out_labels,
                                uint32_t* out_tile_index, double* out_table, size_t out_ld);

/* v1 entry points (uint32 tiles, montage slide semantics).  `max_label` is only validated:
 * a label above it is NYXHIP_ERR_INVALID_ARG; pass 0xFFFFFFFF to accept every value. */
int nyxhip_featurize_tile(nyxhip_ctx* ctx, const uint32_t* inten, const uint32_t* label,
                          uint32_t width, uint32_t height, int32_t memory,
                          uint32_t max_label, uint32_t family_mask,
                          const nyxhip_settings* s,
                          uint32_t* out_labels, uint64_t max_rows,
                          double* out_table, size_t out_ld, uint64_t* n_roi_out);
int nyxhip_featurize_tiles(nyxhip_ctx* ctx, const uint32_t* inten, const uint32_t* label,
                           uint32_t width, uint32_t height, uint32_t n_tiles, int32_t memory,
                           uint32_t max_label, uint32_t family_mask, const nyxhip_settings* s,
                           uint32_t* out_labels, uint32_t* out_tile_index, uint64_t max_rows,
                           double* out_table, size_t out_ld, uint64_t* n_roi_out);

/* ---- measurement hooks -------------------------------------------------------
 * Average device time (ms) per featurize call since the last nyxhip_timing_reset(),
 * measured with hipEvents recorded on the launch stream around EVERYTHING the call
 * enqueues (all kernel groups, the moments pair, the large-ROI passes).  Used by
 * bench.py for `roofline.achieved`.  on = 1: those two events per call; on = 2: also two
 * events around every launch group of the call (the "ms" of nyxhip_launch_report) -- each
 * recorded event is a marker packet on the stream, a handful of microseconds of device
 * time, so the per-group level is for diagnosis, not for the timed region of a benchmark. */
int nyxhip_timing_enable(nyxhip_ctx* ctx, int on);
int nyxhip_timing_reset(nyxhip_ctx* ctx);
int nyxhip_timing_get(nyxhip_ctx* ctx, double* avg_kernel_ms, uint64_t* n_launches);

/* The size classes of the LAST nyxhip_featurize_batch[_async] call on this context, as launched: a JSON array written to
 * buf (NUL-terminated, truncated to buf_len), one object per launch group:
 *   {"class": c, "size_class": 0..4, "wide_range": 0|1, "rois": n, "max_px": .., "max_bbox_area": .., "max_range": ..,
 *    "max_side": .., "workspace": <bit mask>, "cooperative": <bit mask>, "ms": t, "lane_ms": t2}
 * "class" < 0 = a launch over the whole batch, enqueued without counting anything because the stated batch extrema rule out
 * all but the two smallest size classes and wide intensity ranges (-1: texture + dependence kernels, -2: feature kernels,
 * -4 / -5: one-wave / four-wave shape kernels; "rois" is then the batch size); "workspace" 0 = kernels
 * with their state in LDS, else the kernel groups that run from the global workspace (bit 0 INTENSITY + GLCM, 1 texture,
 * 2 shape, 3 dependence); "cooperative" = families of the class served by several workgroups per ROI (ROIs beyond LDS): bit 0 INTENSITY +
 * GLCM, bit 1 GLRLM + GLSZM + NGTDM;
 * "ms" = device time of the class's launches on the call's stream and "lane_ms" = fork-to-end time of the class's workspace
 * launches on their own stream (large classes run them beside the main stream; null otherwise) when
 * nyxhip_timing_enable(ctx, 2) was in force for the call (waits for them), else null.  Counterpart in the
 * reference: none -- its worker threads take ROIs of any size (parallel.h:23-42); here a launch is sized by its largest
 * ROI, so a call is split by ROI size.  Returns the number of bytes the full text needs (excluding the NUL), or < 0. */
int nyxhip_launch_report(nyxhip_ctx* ctx, char* buf, size_t buf_len);

#ifdef __cplusplus
}
#endif
#endif /* NYXHIP_H */
