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

/* looks the inflate library up now (dlopen, a few ms the first time) instead of inside the first decode */
void png_warm_up();
bool png_read(const std::string& path, PngImage& out, std::string* err = nullptr);
/* the decoder threads' path: first channel of a width x height image as float * unit (cv::Mat::convertTo(CV_32FC1, unit),
 * ImageLoader.h:167-172) straight into dst, without the intermediate image */
bool png_read_scaled(const std::string& path, float* dst, int width, int height, float unit, std::string* err = nullptr);
/* 16-bit grayscale writer (stored-deflate friendly): used to build synthetic datasets in tests */
/* filter_mode: scanline filter type of every row (0 = None ... 4 = Paeth), 5 = row y uses filter y % 5 and the data is
 * split into two IDAT chunks (the reader's self-test) */
bool png_write_gray16(const std::string& path, int width, int height, const uint16_t* pixels, int filter_mode = 0);

#endif
