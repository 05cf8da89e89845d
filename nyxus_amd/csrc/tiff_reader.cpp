// tiff_reader.cpp -- include/nyxtiff.h: tile-by-tile / strip-by-strip TIFF decode into a native-width image buffer.
// Host-only (g++ + libtiff); the device path starts at nyxhip_featurize_tiles_v2.
#include <tiffio.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <type_traits>

#include "../../include/nyxtiff.h"

namespace {

int fail(char* err, size_t n, const std::string& m)
{
    if (err && n) snprintf(err, n, "%s", m.c_str());
    return 1;
}

void quiet(const char*, const char*, va_list) {}

struct Tif {
    TIFF* t = nullptr;
    explicit Tif(const char* path)
    {
        TIFFSetWarningHandler(quiet);      // OME-TIFF private tags are not errors
        t = TIFFOpen(path, "r");
    }
    ~Tif() { if (t) TIFFClose(t); }
};

int read_info(TIFF* t, nyxtiff_info_t* o)
{
    uint16_t bits = 1, fmt = SAMPLEFORMAT_UINT, spp = 1;
    uint32_t w = 0, h = 0, tw = 0, th = 0, rps = 0;
    if (!TIFFGetField(t, TIFFTAG_IMAGEWIDTH, &w) || !TIFFGetField(t, TIFFTAG_IMAGELENGTH, &h)) return 1;
    TIFFGetFieldDefaulted(t, TIFFTAG_BITSPERSAMPLE, &bits);
    TIFFGetFieldDefaulted(t, TIFFTAG_SAMPLEFORMAT, &fmt);
    TIFFGetFieldDefaulted(t, TIFFTAG_SAMPLESPERPIXEL, &spp);
    if (fmt < 1 || fmt > 3) fmt = 1;      // grayscale_tiff.h:446-449: unknown formats are read as unsigned
    if (TIFFIsTiled(t)) { TIFFGetField(t, TIFFTAG_TILEWIDTH, &tw); TIFFGetField(t, TIFFTAG_TILELENGTH, &th); }
    else { TIFFGetFieldDefaulted(t, TIFFTAG_ROWSPERSTRIP, &rps); th = rps > h ? h : rps; }
    o->width = w; o->height = h; o->bits_per_sample = bits; o->sample_format = fmt; o->tile_width = tw; o->tile_height = th; o->samples_per_pixel = spp;
    return 0;
}

// loadTile<FileType> of the reference (grayscale_tiff.h:257-312): native cast, negatives of a signed file type clamp to 0
template <typename F, typename D>
inline D cast_sample(F v)
{
    if (std::is_signed<F>::value && v < 0) v = 0;
    return (D)v;
}

template <typename F, typename D>
void copy_block(const void* src, size_t src_row_samples, uint32_t spp, D* dst, size_t dst_w, uint32_t x0, uint32_t y0, uint32_t bw, uint32_t bh)
{
    const F* s = (const F*)src;
    for (uint32_t r = 0; r < bh; r++) {
        const F* sr = s + (size_t)r * src_row_samples;
        D* dr = dst + (size_t)(y0 + r) * dst_w + x0;
        if (spp == 1)
            for (uint32_t c = 0; c < bw; c++) dr[c] = cast_sample<F, D>(sr[c]);
        else
            for (uint32_t c = 0; c < bw; c++) dr[c] = cast_sample<F, D>(sr[(size_t)c * spp]);
    }
}

template <typename D>
int copy_any(uint32_t fmt, uint32_t bits, const void* src, size_t src_row_samples, uint32_t spp, D* dst, size_t dst_w, uint32_t x0, uint32_t y0, uint32_t bw,
             uint32_t bh)
{
#define NYX_CP(F) copy_block<F, D>(src, src_row_samples, spp, dst, dst_w, x0, y0, bw, bh); return 0
    if (fmt == SAMPLEFORMAT_UINT) {
        switch (bits) { case 8: NYX_CP(uint8_t); case 16: NYX_CP(uint16_t); case 32: NYX_CP(uint32_t); case 64: NYX_CP(uint64_t); }
    } else if (fmt == SAMPLEFORMAT_INT) {
        switch (bits) { case 8: NYX_CP(int8_t); case 16: NYX_CP(int16_t); case 32: NYX_CP(int32_t); case 64: NYX_CP(int64_t); }
    }
#undef NYX_CP
    return 1;
}

template <typename D>
int read_as(TIFF* t, const nyxtiff_info_t& I, D* dst, std::string& why)
{
    const uint32_t W = I.width, H = I.height, spp = I.samples_per_pixel;
    if (I.tile_width) {
        const uint32_t tw = I.tile_width, th = I.tile_height;
        std::vector<unsigned char> buf((size_t)TIFFTileSize(t));
        for (uint32_t y = 0; y < H; y += th)
            for (uint32_t x = 0; x < W; x += tw) {
                if (TIFFReadTile(t, buf.data(), x, y, 0, 0) < 0) { why = "TIFFReadTile failed"; return 1; }
                const uint32_t bw = x + tw > W ? W - x : tw, bh = y + th > H ? H - y : th;      // edge tiles are clipped to the image
                if (copy_any<D>(I.sample_format, I.bits_per_sample, buf.data(), (size_t)tw * spp, spp, dst, W, x, y, bw, bh)) { why = "unsupported sample type"; return 1; }
            }
    } else {
        std::vector<unsigned char> buf((size_t)TIFFScanlineSize(t));
        for (uint32_t y = 0; y < H; y++) {
            if (TIFFReadScanline(t, buf.data(), y, 0) < 0) { why = "TIFFReadScanline failed"; return 1; }
            if (copy_any<D>(I.sample_format, I.bits_per_sample, buf.data(), (size_t)W * spp, spp, dst, W, 0, y, W, 1)) { why = "unsupported sample type"; return 1; }
        }
    }
    return 0;
}

} // namespace

extern "C" {

int nyxtiff_info(const char* path, nyxtiff_info_t* info, char* err, size_t err_len)
{
    if (!path || !info) return fail(err, err_len, "null argument");
    Tif f(path);
    if (!f.t) return fail(err, err_len, std::string("cannot open ") + path);
    if (read_info(f.t, info)) return fail(err, err_len, std::string(path) + ": missing image dimensions");
    return 0;
}

int nyxtiff_read(const char* path, void* dst, int dst_bytes, uint32_t width, uint32_t height, char* err, size_t err_len)
{
    if (!path || !dst) return fail(err, err_len, "null argument");
    Tif f(path);
    if (!f.t) return fail(err, err_len, std::string("cannot open ") + path);
    nyxtiff_info_t I;
    if (read_info(f.t, &I)) return fail(err, err_len, std::string(path) + ": missing image dimensions");
    if (I.width != width || I.height != height) return fail(err, err_len, std::string(path) + ": image size differs from the buffer's");
    if (I.sample_format == SAMPLEFORMAT_IEEEFP)
        return fail(err, err_len, std::string(path) + ": floating-point TIFFs need the reference's fpimage rescaling options (outside the hot path)");
    uint16_t planar = PLANARCONFIG_CONTIG;
    TIFFGetFieldDefaulted(f.t, TIFFTAG_PLANARCONFIG, &planar);
    if (planar != PLANARCONFIG_CONTIG && I.samples_per_pixel > 1) I.samples_per_pixel = 1;   // separate planes: plane 0 is read as it is
    const uint32_t need = I.bits_per_sample >= 32 ? 4 : I.bits_per_sample / 8;
    if (need == 0 || (uint32_t)dst_bytes < need || (dst_bytes != 1 && dst_bytes != 2 && dst_bytes != 4))
        return fail(err, err_len, std::string(path) + ": destination elements narrower than the file's samples");
    std::string why;
    int rc = dst_bytes == 1 ? read_as<uint8_t>(f.t, I, (uint8_t*)dst, why) : dst_bytes == 2 ? read_as<uint16_t>(f.t, I, (uint16_t*)dst, why)
                                                                                             : read_as<uint32_t>(f.t, I, (uint32_t*)dst, why);
    if (rc) return fail(err, err_len, std::string(path) + ": " + why);
    return 0;
}

} // extern "C"
