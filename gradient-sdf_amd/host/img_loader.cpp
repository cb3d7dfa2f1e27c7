#include "img_loader.h"

#include <iomanip>
#include <iostream>
#include <sstream>

#include "png16.h"

bool ImageLoader::load_intrinsics(const std::string& filename) {
    if (filename.empty()) return false;
    std::ifstream in(path_ + filename);
    if (!in.is_open()) return false;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float v = 0; in >> v; K_(i, j) = v; }
    return true;
}

bool ImageLoader::load_depth(const std::string& filename, DepthImage& depth) {
    if (filename.empty()) { std::cerr << "Error: missing filename" << std::endl; return false; }
    PngImage img;
    std::string err;
    if (!png_read(path_ + filename, img, &err)) {
        std::cerr << "Error: empty depth image " << path_ + filename << " (" << err << ")" << std::endl;
        return false;
    }
    depth.rows = img.height; depth.cols = img.width;
    depth.buf.resize(img.first_channel.size());
    for (size_t i = 0; i < depth.buf.size(); ++i) depth.buf[i] = (float)img.first_channel[i] * unit_;   /* convertTo(CV_32FC1, unit_) */
    return true;
}

bool ImageLoader::decode_depth(const std::string& filename, float* dst, int W, int H, std::string* err) const {
    std::string e;
    if (filename.empty() || !png_read_scaled(path_ + filename, dst, W, H, unit_, &e)) {   /* convertTo(CV_32FC1, unit_) */
        if (err) *err = e.find("differs") != std::string::npos ? "frame size of " + filename + " differs from --width/--height"
                                                               : "empty depth image " + path_ + filename + " (" + e + ")";
        return false;
    }
    return true;
}

bool ImageLoader::load_pose(const std::string& filename, std::vector<Mat4f>& poses) {
    std::ifstream file(filename.c_str());
    if (!file.is_open()) { std::cout << "can't load poses!" << std::endl; return false; }
    std::string line;
    while (std::getline(file, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::stringstream s(line);
        float ts, t[3], q[4];                                  /* ts tx ty tz qx qy qz qw */
        if (!(s >> ts >> t[0] >> t[1] >> t[2] >> q[0] >> q[1] >> q[2] >> q[3])) continue;
        if (q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2] < 0.99f)
            std::cerr << "pose " << ts << " has invalid rotation" << std::endl;
        float R[9];
        gsdf_quat_to_R(q, R);                                  /* q.toRotationMatrix(), not normalised first */
        Mat4f T;
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T(r, c) = R[3 * r + c]; T(r, 3) = t[r]; }
        poses.push_back(T);
    }
    return true;
}

bool SynthLoader::load_next(ColorImage&, DepthImage& depth) {
    std::stringstream ss;
    ss << std::setfill('0') << std::setw(3) << counter_;
    timestamp_rgb_ = ss.str();
    timestamp_depth_ = timestamp_rgb_;
    if (!load_depth("depth/" + timestamp_rgb_ + ".png", depth)) return false;
    ++counter_;
    return true;
}

bool SynthLoader::next_entry(std::string& depth_file, std::string& timestamp) {
    std::stringstream ss;
    ss << std::setfill('0') << std::setw(3) << counter_;
    timestamp_rgb_ = ss.str();
    timestamp_depth_ = timestamp_rgb_;
    depth_file = "depth/" + timestamp_rgb_ + ".png";
    timestamp = timestamp_depth_;
    std::ifstream probe(path_ + depth_file);
    if (!probe.is_open()) return false;
    ++counter_;
    return true;
}

bool TumrgbdLoader::next_entry(std::string& depth_file, std::string& timestamp) {
    std::string line = "#", rgb_file;
    while (line.empty() || line.at(0) == '#')
        if (!std::getline(assoc_, line)) return false;
    std::istringstream ss(line);
    ss >> timestamp_rgb_ >> rgb_file >> timestamp_depth_ >> depth_file;
    timestamp = timestamp_depth_;
    return !depth_file.empty();
}

bool TumrgbdLoader::load_next(ColorImage&, DepthImage& depth) {
    std::string line = "#", rgb_file, depth_file;
    while (line.empty() || line.at(0) == '#')
        if (!std::getline(assoc_, line)) return false;
    std::istringstream ss(line);
    ss >> timestamp_rgb_ >> rgb_file >> timestamp_depth_ >> depth_file;
    std::cout << "load image " << timestamp_rgb_ << std::endl;
    return load_depth(depth_file, depth);
}
