#include "png16.h"

#include <dlfcn.h>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>

namespace {
uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
bool fail(std::string* err, const char* m) { if (err) *err = m; return false; }

/* The inflate of a frame is what a decoder thread spends its time on.  libdeflate (a runtime library of this image, no
 * header: the three entry points are declared here) inflates a depth image 3x faster than zlib's uncompress; it is looked
 * up once with dlopen, every thread keeps its own decompressor, and zlib does the job where the library is missing
 * (or GSDF_NO_LIBDEFLATE is set: the self-test compares the two). */
struct Deflate {
    void* (*alloc)() = nullptr;
    int (*zlib_decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_)(void*) = nullptr;
    Deflate() {
        if (std::getenv("GSDF_NO_LIBDEFLATE")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        zlib_decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_zlib_decompress");
        free_ = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !zlib_decompress || !free_) { alloc = nullptr; zlib_decompress = nullptr; free_ = nullptr; }
    }
};
const Deflate& deflate_lib() { static const Deflate d; return d; }
struct ThreadDecompressor {
    void* d = nullptr;
    ~ThreadDecompressor() { if (d) deflate_lib().free_(d); }
};
bool inflate_all(const unsigned char* in, size_t n_in, unsigned char* out, size_t n_out) {
    const Deflate& L = deflate_lib();
    if (L.alloc) {
        thread_local ThreadDecompressor td;
        if (!td.d) td.d = L.alloc();
        size_t got = 0;
        if (td.d && L.zlib_decompress(td.d, in, n_in, out, n_out, &got) == 0 && got == n_out) return true;
        /* anything unusual (trailing data, a stream libdeflate rejects): let zlib have the last word */
    }
    uLongf len = (uLongf)n_out;
    return uncompress(out, &len, in, (uLong)n_in) == Z_OK && len == n_out;
}

/* per-thread scratch: file bytes, concatenated IDAT (only when there are several), inflated scanlines */
struct Scratch { std::vector<unsigned char> file, idat, raw; };
Scratch& scratch() { thread_local Scratch s; return s; }

struct Header { int width = 0, height = 0, bit_depth = 0, channels = 0; };

/* file -> unfiltered scanlines in scratch().raw: row y starts at (stride + 1) * y + 1 */
/* want_w / want_h > 0: the size the caller's buffer was made for -- a file of another size is rejected right behind its IHDR,
 * before anything is sized by the header's numbers */
bool decode_rows_impl(const std::string& path, Header& hd, size_t& stride, int& bpp, std::string* err, int want_w, int want_h) {
    Scratch& S = scratch();
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return fail(err, "cannot open file");
    size_t n = 0;
    if (std::fseek(f, 0, SEEK_END) == 0) {
        const long sz = std::ftell(f);
        std::rewind(f);
        if (sz > 0) {
            if (S.file.size() < (size_t)sz) S.file.resize((size_t)sz);
            n = std::fread(S.file.data(), 1, (size_t)sz, f);
        }
    }
    std::fclose(f);
    const unsigned char* buf = S.file.data();
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (n < 8 + 25 || std::memcmp(buf, sig, 8) != 0) return fail(err, "not a PNG");
    size_t pos = 8;
    int color_type = -1, interlace = 0, n_idat = 0;
    const unsigned char* idat = nullptr;
    size_t idat_len = 0;
    while (pos + 12 <= n) {
        const uint32_t len = be32(&buf[pos]);
        const char* type = (const char*)&buf[pos + 4];
        if (pos + 12 + (size_t)len > n) return fail(err, "truncated chunk");
        const unsigned char* d = &buf[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return fail(err, "bad IHDR");
            hd.width = (int)be32(d); hd.height = (int)be32(d + 4);
            hd.bit_depth = d[8]; color_type = d[9]; interlace = d[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            if (n_idat == 0) { idat = d; idat_len = len; }
            else {
                if (n_idat == 1) S.idat.assign(idat, idat + idat_len);
                S.idat.insert(S.idat.end(), d, d + len);
            }
            ++n_idat;
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (n_idat > 1) { idat = S.idat.data(); idat_len = S.idat.size(); }
    if (interlace) return fail(err, "interlaced PNG not supported");
    if (hd.bit_depth != 8 && hd.bit_depth != 16) return fail(err, "bit depth must be 8 or 16");
    switch (color_type) {
        case 0: hd.channels = 1; break;
        case 2: hd.channels = 3; break;
        case 4: hd.channels = 2; break;
        case 6: hd.channels = 4; break;
        default: return fail(err, "palette PNG not supported");
    }
    if (hd.width <= 0 || hd.height <= 0 || !idat) return fail(err, "no image data");
    if (want_w > 0 && (hd.width != want_w || hd.height != want_h)) return fail(err, "frame size differs from --width/--height");
    bpp = hd.channels * hd.bit_depth / 8;
    /* the header's numbers are untrusted: bound what gets allocated (a corrupt IHDR must not overflow size_t or ask for gigabytes) */
    /* (factors bounded one by one: the 64-bit product of three 31-bit numbers can wrap back below the limit) */
    if (hd.width > 65535 || hd.height > 65535 ||
        (unsigned long long)hd.width * (unsigned long long)hd.height * (unsigned long long)bpp > (1ull << 31)) return fail(err, "image too large");
    stride = (size_t)hd.width * bpp;
    const size_t raw_len = (stride + 1) * (size_t)hd.height;
    if (S.raw.size() < raw_len) S.raw.resize(raw_len);
    if (!inflate_all(idat, idat_len, S.raw.data(), raw_len)) return fail(err, "zlib inflate failed");
    /* undo the scanline filters in place, one loop per filter type */
    unsigned char* raw = S.raw.data();
    for (int y = 0; y < hd.height; ++y) {
        unsigned char* cur = raw + (stride + 1) * (size_t)y + 1;
        const unsigned char* prev = y ? cur - (stride + 1) : nullptr;
        const size_t B = (size_t)bpp;
        switch (cur[-1]) {
            case 0: break;
            case 1:
                for (size_t i = B; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + cur[i - B]);
                break;
            case 2:
                if (prev) for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]);
                break;
            case 3:
                for (size_t i = 0; i < stride; ++i) {
                    const int a = i >= B ? cur[i - B] : 0, b = prev ? prev[i] : 0;
                    cur[i] = (unsigned char)(cur[i] + (a + b) / 2);
                }
                break;
            case 4:
                for (size_t i = 0; i < stride; ++i) {
                    const int a = i >= B ? cur[i - B] : 0, b = prev ? prev[i] : 0, c = (prev && i >= B) ? prev[i - B] : 0;
                    cur[i] = (unsigned char)(cur[i] + paeth(a, b, c));
                }
                break;
            default: return fail(err, "bad filter type");
        }
    }
    return true;
}
/* decoder threads must not terminate the process: an allocation failure (file buffer, IDAT, scanlines) becomes an error string */
bool decode_rows(const std::string& path, Header& hd, size_t& stride, int& bpp, std::string* err, int want_w = 0, int want_h = 0) {
    try {
        return decode_rows_impl(path, hd, stride, bpp, err, want_w, want_h);
    } catch (const std::bad_alloc&) {
        return fail(err, "out of memory while decoding");
    }
}
}

void png_warm_up() { (void)deflate_lib(); }

bool png_read(const std::string& path, PngImage& out, std::string* err) {
    Header hd;
    size_t stride = 0;
    int bpp = 0;
    if (!decode_rows(path, hd, stride, bpp, err)) return false;
    out.width = hd.width; out.height = hd.height; out.bit_depth = hd.bit_depth; out.channels = hd.channels;
    out.first_channel.resize((size_t)hd.width * hd.height);
    const unsigned char* raw = scratch().raw.data();
    for (int y = 0; y < hd.height; ++y) {
        const unsigned char* line = raw + (stride + 1) * (size_t)y + 1;
        uint16_t* o = &out.first_channel[(size_t)y * hd.width];
        if (hd.bit_depth == 16) for (int x = 0; x < hd.width; ++x) { const unsigned char* px = line + (size_t)x * bpp; o[x] = (uint16_t)((px[0] << 8) | px[1]); }
        else for (int x = 0; x < hd.width; ++x) o[x] = line[(size_t)x * bpp];
    }
    return true;
}

bool png_read_scaled(const std::string& path, float* dst, int width, int height, float unit, std::string* err) {
    Header hd;
    size_t stride = 0;
    int bpp = 0;
    if (width <= 0 || height <= 0) return fail(err, "bad frame size");
    if (!decode_rows(path, hd, stride, bpp, err, width, height)) return false;
    const unsigned char* raw = scratch().raw.data();
    for (int y = 0; y < height; ++y) {
        const unsigned char* line = raw + (stride + 1) * (size_t)y + 1;
        float* o = dst + (size_t)y * width;
        if (hd.bit_depth == 16) for (int x = 0; x < width; ++x) { const unsigned char* px = line + (size_t)x * bpp; o[x] = (float)(uint16_t)((px[0] << 8) | px[1]) * unit; }
        else for (int x = 0; x < width; ++x) o[x] = (float)line[(size_t)x * bpp] * unit;
    }
    return true;
}

bool png_write_gray16(const std::string& path, int width, int height, const uint16_t* pixels, int filter_mode) {
    const size_t stride = 2 * (size_t)width;
    std::vector<unsigned char> raw((stride + 1) * height);
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    for (int y = 0; y < height; ++y) {
        unsigned char* line = &raw[(stride + 1) * y];
        for (int x = 0; x < width; ++x) {
            const uint16_t v = pixels[(size_t)y * width + x];
            cur[2 * x] = (unsigned char)(v >> 8);
            cur[2 * x + 1] = (unsigned char)(v & 0xff);
        }
        const int ft = filter_mode == 5 ? y % 5 : filter_mode;          /* scanline filter of this row (PNG spec 6.3) */
        line[0] = (unsigned char)ft;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= 2 ? cur[i - 2] : 0, b = prev[i], c = i >= 2 ? prev[i - 2] : 0;
            const int pred = ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) / 2 : ft == 4 ? paeth(a, b, c) : 0;
            line[1 + i] = (unsigned char)(cur[i] - pred);
        }
        prev = cur;
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
    if (filter_mode == 5) {                                            /* the mixed-filter test image also splits its data */
        const uint32_t half = (uint32_t)clen / 2;
        chunk("IDAT", comp.data(), half);
        chunk("IDAT", comp.data() + half, (uint32_t)clen - half);
    } else chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    std::fclose(f);
    return true;
}
