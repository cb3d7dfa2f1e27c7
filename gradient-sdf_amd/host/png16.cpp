#include "png16.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
bool fail(std::string* err, const char* m) { if (err) *err = m; return false; }
}

bool png_read(const std::string& path, PngImage& out, std::string* err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return fail(err, "cannot open file");
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (buf.size() < 8 + 25 || std::memcmp(buf.data(), sig, 8) != 0) return fail(err, "not a PNG");
    size_t pos = 8;
    int color_type = -1, interlace = 0;
    std::vector<unsigned char> idat;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const char* type = (const char*)&buf[pos + 4];
        if (pos + 12 + len > buf.size()) return fail(err, "truncated chunk");
        const unsigned char* d = &buf[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            out.width = (int)be32(d); out.height = (int)be32(d + 4);
            out.bit_depth = d[8]; color_type = d[9]; interlace = d[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (interlace) return fail(err, "interlaced PNG not supported");
    if (out.bit_depth != 8 && out.bit_depth != 16) return fail(err, "bit depth must be 8 or 16");
    switch (color_type) {
        case 0: out.channels = 1; break;
        case 2: out.channels = 3; break;
        case 4: out.channels = 2; break;
        case 6: out.channels = 4; break;
        default: return fail(err, "palette PNG not supported");
    }
    const int bpp = out.channels * out.bit_depth / 8;
    const size_t stride = (size_t)out.width * bpp;
    std::vector<unsigned char> raw((stride + 1) * out.height);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        return fail(err, "zlib inflate failed");
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    out.first_channel.assign((size_t)out.width * out.height, 0);
    for (int y = 0; y < out.height; ++y) {
        const unsigned char* line = &raw[(stride + 1) * y];
        const int ft = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v = line[1 + i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: return fail(err, "bad filter type");
            }
            cur[i] = (unsigned char)v;
        }
        for (int x = 0; x < out.width; ++x) {
            const unsigned char* px = &cur[(size_t)x * bpp];
            out.first_channel[(size_t)y * out.width + x] = out.bit_depth == 16 ? (uint16_t)((px[0] << 8) | px[1]) : px[0];
        }
        prev.swap(cur);
    }
    return true;
}

bool png_write_gray16(const std::string& path, int width, int height, const uint16_t* pixels) {
    std::vector<unsigned char> raw((size_t)(2 * width + 1) * height);
    for (int y = 0; y < height; ++y) {
        unsigned char* line = &raw[(size_t)(2 * width + 1) * y];
        line[0] = 0;
        for (int x = 0; x < width; ++x) {
            const uint16_t v = pixels[(size_t)y * width + x];
            line[1 + 2 * x] = (unsigned char)(v >> 8);
            line[2 + 2 * x] = (unsigned char)(v & 0xff);
        }
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto chunk = [&](const char* type, const unsigned char* d, uint32_t len) {
        unsigned char hdr[8] = { (unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len,
                                 (unsigned char)type[0], (unsigned char)type[1], (unsigned char)type[2], (unsigned char)type[3] };
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(d, 1, len, f);
        uLong crc = crc32(0L, hdr + 4, 4);
        if (len) crc = crc32(crc, d, len);
        unsigned char c[4] = { (unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc };
        std::fwrite(c, 1, 4, f);
    };
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::fwrite(sig, 1, 8, f);
    unsigned char ihdr[13] = { (unsigned char)(width >> 24), (unsigned char)(width >> 16), (unsigned char)(width >> 8), (unsigned char)width,
                               (unsigned char)(height >> 24), (unsigned char)(height >> 16), (unsigned char)(height >> 8), (unsigned char)height,
                               16, 0, 0, 0, 0 };
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    std::fclose(f);
    return true;
}
