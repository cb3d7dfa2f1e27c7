/* Dataset readers mirroring the reference loaders (cpp/include/img_loader): 16-bit PNG depth x unit -> float metres
 * (ImageLoader.h:159-175), intrinsics.txt (ImageLoader.h:138-157), TUM pose files
 * (ImageLoader.h:231-259), the synthetic layout depth/%03d.png 1-based (SynthLoader.h:64-83) and
 * TUM associated.txt (TumrgbdLoader.h:80-104).  Colour images are not decoded: update() ignores them. */
#ifndef GSDF_HOST_IMG_LOADER_H_
#define GSDF_HOST_IMG_LOADER_H_

#include <fstream>
#include <string>
#include <vector>

#include "Image.h"

class ImageLoader {
protected:
    Mat3f K_;
    const float unit_;
    const std::string path_;
    std::string timestamp_rgb_, timestamp_depth_;
public:
    ImageLoader(float unit, const std::string& path) : unit_(unit), path_(path) {}
    virtual ~ImageLoader() {}
    Mat3f K() const { return K_; }
    std::string depth_timestamp() const { return timestamp_depth_; }
    std::string rgb_timestamp() const { return timestamp_rgb_; }
    bool load_intrinsics(const std::string& filename = "intrinsics.txt");
    bool load_depth(const std::string& filename, DepthImage& depth);
    /* the same decode into a caller's buffer of W x H floats; no member is modified: callable from several threads */
    bool decode_depth(const std::string& filename, float* dst, int W, int H, std::string* err) const;
    /* advance to the next frame WITHOUT decoding it: its depth file (relative to the directory) and depth timestamp */
    virtual bool next_entry(std::string& depth_file, std::string& timestamp) = 0;
    static bool load_pose(const std::string& filename, std::vector<Mat4f>& poses);
    virtual bool load_next(ColorImage& color, DepthImage& depth) = 0;
    virtual void reset() = 0;
};

class SynthLoader : public ImageLoader {       /* unit 1/1000, files depth/001.png ... */
    size_t counter_ = 1;
public:
    explicit SynthLoader(const std::string& path) : ImageLoader(1.f / 1000, path) {}
    bool load_next(ColorImage& color, DepthImage& depth) override;
    bool next_entry(std::string& depth_file, std::string& timestamp) override;
    void reset() override { counter_ = 1; }
};

class TumrgbdLoader : public ImageLoader {     /* unit 1/5000, associated.txt */
    std::ifstream assoc_;
public:
    explicit TumrgbdLoader(const std::string& path) : ImageLoader(1.f / 5000, path) { assoc_.open(path_ + "associated.txt"); }
    bool load_next(ColorImage& color, DepthImage& depth) override;
    bool next_entry(std::string& depth_file, std::string& timestamp) override;
    void reset() override { assoc_.close(); assoc_.open(path_ + "associated.txt"); }
};

#endif
