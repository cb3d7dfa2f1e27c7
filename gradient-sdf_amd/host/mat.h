/*
 * mat.h -- the small linear-algebra vocabulary the reference takes from Eigen and Sophus
 * (cpp/include/mat.h:47-66: Vec3f, Mat3f, Vec3i, Vec6f, SE3), dependency-free.  Only what the
 * facade needs: the arithmetic of the hot path itself runs on the GPU behind include/gsdf.h.
 * SE3 keeps Sophus' state (unit quaternion + translation) and uses csrc/gsdf_math.h for
 * rotationMatrix() / Quaternion(Matrix3) / exp so that host and device agree bit for bit.
 */
#ifndef GSDF_HOST_MAT_H_
#define GSDF_HOST_MAT_H_

#include <array>
#include <cmath>
#include <cstdint>

#include "../csrc/gsdf_math.h"

struct Vec3f {
    float v[3] = { 0.f, 0.f, 0.f };
    Vec3f() {}
    Vec3f(float x, float y, float z) { v[0] = x; v[1] = y; v[2] = z; }
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
    const float* data() const { return v; }
    float* data() { return v; }
    float norm() const { return std::sqrt(v[0] * v[0] + (v[1] * v[1] + v[2] * v[2])); }
    Vec3f normalized() const {
        const gsdf_v3 n = gsdf_normalized3(gsdf_v3{ v[0], v[1], v[2] });
        return Vec3f(n.x, n.y, n.z);
    }
};
inline Vec3f operator+(const Vec3f& a, const Vec3f& b) { return Vec3f(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3f operator-(const Vec3f& a, const Vec3f& b) { return Vec3f(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vec3f operator-(const Vec3f& a) { return Vec3f(-a[0], -a[1], -a[2]); }
inline Vec3f operator*(float s, const Vec3f& a) { return Vec3f(s * a[0], s * a[1], s * a[2]); }

struct Vec3i {
    int32_t v[3] = { 0, 0, 0 };
    Vec3i() {}
    Vec3i(int x, int y, int z) { v[0] = x; v[1] = y; v[2] = z; }
    int32_t& operator[](int i) { return v[i]; }
    int32_t operator[](int i) const { return v[i]; }
    bool operator==(const Vec3i& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
};
struct Vec3iHash {
    size_t operator()(const Vec3i& k) const {
        uint64_t h = (uint64_t)(uint32_t)k[0] * 0x9E3779B97F4A7C15ull;
        h ^= (uint64_t)(uint32_t)k[1] * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
        h ^= (uint64_t)(uint32_t)k[2] * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
        return (size_t)(h ^ (h >> 29));
    }
};

/* row-major 3x3 */
struct Mat3f {
    float m[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f };
    float& operator()(int r, int c) { return m[3 * r + c]; }
    float operator()(int r, int c) const { return m[3 * r + c]; }
    const float* data() const { return m; }
    float* data() { return m; }
    static Mat3f Identity() { return Mat3f(); }
};
inline Vec3f operator*(const Mat3f& R, const Vec3f& p) {
    const gsdf_v3 r = gsdf_matvec(R.m, gsdf_v3{ p[0], p[1], p[2] });
    return Vec3f(r.x, r.y, r.z);
}

/* row-major 4x4 homogeneous transform (the loaders' pose type, ImageLoader.h:231-259) */
struct Mat4f {
    float m[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    float& operator()(int r, int c) { return m[4 * r + c]; }
    float operator()(int r, int c) const { return m[4 * r + c]; }
    static Mat4f Identity() { return Mat4f(); }
};

/* Sophus::SE3<float>: unit quaternion (x y z w) + translation; camera->world */
class SE3 {
    float p_[7] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };    /* tx ty tz qx qy qz qw */
public:
    SE3() {}
    /* SE3(Matrix4f): rotation -> quaternion like Eigen::Quaternion(Matrix3) (main_scan_3d.cpp:242,252) */
    explicit SE3(const Mat4f& T) {
        float R[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = T(r, c);
        gsdf_R_to_quat(R, p_ + 3);           /* Sophus SO3(Matrix3) stores Eigen's quaternion as is */
        p_[0] = T(0, 3); p_[1] = T(1, 3); p_[2] = T(2, 3);
    }
    static SE3 from_pose7(const float p[7]) { SE3 s; for (int i = 0; i < 7; ++i) s.p_[i] = p[i]; return s; }
    const float* pose7() const { return p_; }
    float* pose7() { return p_; }
    Mat3f rotationMatrix() const { Mat3f R; gsdf_quat_to_R(p_ + 3, R.m); return R; }
    Vec3f translation() const { return Vec3f(p_[0], p_[1], p_[2]); }
    Mat4f matrix() const {
        Mat4f T; const Mat3f R = rotationMatrix();
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T(r, c) = R(r, c); T(r, 3) = p_[r]; }
        return T;
    }
    /* SE3::exp(xi) * (*this) */
    SE3 left_exp(const float xi[6]) const { SE3 s = *this; gsdf_se3_exp_mul(xi, s.p_); return s; }
};

#endif
