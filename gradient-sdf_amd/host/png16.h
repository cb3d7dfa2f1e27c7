/* Minimal PNG reader (zlib inflate + scanline unfiltering) for the dataset formats the reference
 * loads through cv::imread(IMREAD_ANYDEPTH): 16-bit grayscale depth (ImageLoader.h:159-175), plus
 * 8-bit gray/RGB/RGBA decoded to their first channel.  No libpng in this image; zlib only. */
#ifndef GSDF_HOST_PNG16_H_
#define GSDF_HOST_PNG16_H_

#include <cstdint>
#include <string>
#include <vector>

struct PngImage {
    int width = 0, height = 0, bit_depth = 0, channels = 0;
    std::vector<uint16_t> first_channel;      /* row-major, host byte order */
};

bool png_read(const std::string& path, PngImage& out, std::string* err = nullptr);
/* 16-bit grayscale writer (stored-deflate friendly): used to build synthetic datasets in tests */
bool png_write_gray16(const std::string& path, int width, int height, const uint16_t* pixels);

#endif
