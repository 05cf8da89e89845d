/*
 * nyxtiff.h -- C ABI of the TIFF tile ingest behind featurize_directory (a "next" row).
 *
 * Counterpart of the reference's tile / strip loaders:
 *   NyxusGrayscaleTiffTileLoader  ::loadTileFromFile   /root/reference/src/nyx/grayscale_tiff.h:104-200 (TIFFReadTile per tile)
 *   NyxusGrayscaleTiffStripLoader ::loadTileFromFile   /root/reference/src/nyx/grayscale_tiff.h:473-560 (TIFFReadScanline per row)
 *   ImageLoader::open / load_tile                      /root/reference/src/nyx/image_loader.cpp:14-...
 * The file is decoded tile by tile (or strip by strip) with libtiff and written straight into the caller's image buffer in the
 * element type the device path takes natively (8 / 16 / 32-bit unsigned) -- the reference casts every sample to uint32 on the
 * host; here the cast happens in the kernels.  Sample handling follows the reference's loadTile<FileType>
 * (grayscale_tiff.h:257-312): unsigned samples are copied, negative signed samples are clamped to 0 (the #373 fix), 64-bit
 * samples are truncated to 32 bits.  Floating-point files need the reference's fpimage rescaling options and are refused.
 * First directory (page), first sample per pixel.  Library: nyxus_amd/libnyxtiff.so (host-only, links libtiff).
 */
#ifndef NYXTIFF_H
#define NYXTIFF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct nyxtiff_info_t {
    uint32_t width, height;
    uint32_t bits_per_sample;    /* 8 / 16 / 32 / 64 */
    uint32_t sample_format;      /* libtiff SAMPLEFORMAT_*: 1 unsigned, 2 signed, 3 IEEE float */
    uint32_t tile_width;         /* 0 = the file is stored in strips */
    uint32_t tile_height;        /* rows per strip when tile_width == 0 */
    uint32_t samples_per_pixel;
} nyxtiff_info_t;

/* 0 on success; a message in err (when given) otherwise. */
int nyxtiff_info(const char* path, nyxtiff_info_t* info, char* err, size_t err_len);

/* Decodes the first page into dst[height][width] of dst_bytes-wide unsigned elements (1, 2 or 4; at least the file's sample
 * width, 4 for 64-bit samples).  width / height must be the values nyxtiff_info reported. */
int nyxtiff_read(const char* path, void* dst, int dst_bytes, uint32_t width, uint32_t height, char* err, size_t err_len);

#ifdef __cplusplus
}
#endif
#endif /* NYXTIFF_H */
