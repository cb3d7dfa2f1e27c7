#include "MarchingCubes.h"
#include "../../include/gsdf_mc_tables.h"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <limits>
#include <string>
#include <thread>
#include <unordered_map>

namespace {

/* corner c -> (dx, dy, dz), numbering of LayeredMarchingCubesNoColor::computeLutIndex (:599-606) */
const int CORNER[8][3] = { { 1, 1, 0 }, { 1, 0, 0 }, { 0, 0, 0 }, { 0, 1, 0 }, { 1, 1, 1 }, { 1, 0, 1 }, { 0, 0, 1 }, { 0, 1, 1 } };
/* edge e -> its two corners (the classic numbering: 0-3 bottom ring, 4-7 top ring, 8-11 verticals) */
const int EDGE[12][2] = { { 0, 1 }, { 1, 2 }, { 2, 3 }, { 3, 0 }, { 4, 5 }, { 5, 6 }, { 6, 7 }, { 7, 4 }, { 0, 4 }, { 1, 5 }, { 2, 6 }, { 3, 7 } };
/* The classic case tables (constant data, include/gsdf_mc_tables.h = LayeredMarchingCubesNoColor.cpp:67-352) as
 * per-case edge lists: the mesh must be the reference's, triangle for triangle. */
struct Tables {
    std::vector<int> tri[256];
    int mask[256];
    Tables() {
        for (int c = 0; c < 256; ++c) {
            mask[c] = GSDF_MC_EDGE_TABLE[c];
            for (int k = 0; k < 16 && GSDF_MC_TRI_TABLE[16 * c + k] >= 0; ++k) tri[c].push_back(GSDF_MC_TRI_TABLE[16 * c + k]);
        }
    }
};
const Tables& tables() { static Tables t; return t; }

/* LayeredMarchingCubesNoColor::interpolate (:642-662) */
Vec3f interpolate(float t0, float t1, const Vec3f& v0, const Vec3f& v1, float iso) {
    if (std::fabs(iso - t0) < 1e-7) return v0;
    if (std::fabs(iso - t1) < 1e-7) return v1;
    if (std::fabs(t0 - t1) < 1e-7) return v0;
    double mu = (iso - t0) / (t1 - t0);
    if (mu > 1.0) mu = 1.0; else if (mu < 0) mu = 0.0;
    Vec3f v;
    for (int k = 0; k < 3; ++k) v[k] = (float)(v0[k] + mu * (v1[k] - v0[k]));
    return v;
}

} // namespace

const std::vector<int>& MarchingCubes::triangles(int cube_index) { return tables().tri[cube_index & 255]; }
int MarchingCubes::edge_mask(int cube_index) { return tables().mask[cube_index & 255]; }

bool MarchingCubes::computeIsoSurface(const std::vector<int32_t>& keys, const std::vector<float>& payload, float iso) {
    vertices_.clear();
    faces_.clear();
    const size_t n = keys.size() / 3;
    if (!n) return false;
    int mn[3] = { std::numeric_limits<int>::max(), std::numeric_limits<int>::max(), std::numeric_limits<int>::max() };
    int mx[3] = { std::numeric_limits<int>::min(), std::numeric_limits<int>::min(), std::numeric_limits<int>::min() };
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], keys[3 * i + a]); mx[a] = std::max(mx[a], keys[3 * i + a]); }
    const int dim[3] = { mx[0] - mn[0] + 1, mx[1] - mn[1] + 1, mx[2] - mn[2] + 1 };
    const size_t area = (size_t)dim[0] * dim[1];
    /* bucket the voxels by z-layer so that two layers can be rasterised at a time (:393-561) */
    std::vector<std::vector<uint32_t>> by_layer((size_t)dim[2]);
    for (size_t i = 0; i < n; ++i) by_layer[(size_t)(keys[3 * i + 2] - mn[2])].push_back((uint32_t)i);
    std::vector<float> tsdf(2 * area, 0.f), wgt(2 * area, 0.f);
    auto copy_layer = [&](int z) {                                      /* copyLayer (:565-587) */
        float* w = &wgt[(size_t)(z & 1) * area];
        float* d = &tsdf[(size_t)(z & 1) * area];
        std::fill(w, w + area, 0.f);
        for (uint32_t i : by_layer[(size_t)z]) {
            const size_t off = (size_t)(keys[3 * i + 1] - mn[1]) * dim[0] + (size_t)(keys[3 * i] - mn[0]);
            w[off] = payload[5 * i + 4];
            d[off] = payload[5 * i];
        }
    };
    const Vec3f origin(-(float)mn[0] * vs_, -(float)mn[1] * vs_, -(float)mn[2] * vs_);      /* origin_ (:377) */
    auto world = [&](int i, int j, int k) { return Vec3f((float)i * vs_ - origin[0], (float)j * vs_ - origin[1], (float)k * vs_ - origin[2]); };
    copy_layer(0);
    for (int z = 0; z < dim[2] - 1; ++z) {
        copy_layer(z + 1);
        for (int y = 0; y < dim[1] - 1; ++y)
            for (int x = 0; x < dim[0] - 1; ++x) {
                size_t off[8];
                bool ok = true;
                int idx = 0;
                for (int c = 0; c < 8; ++c) {
                    off[c] = (size_t)((z + CORNER[c][2]) & 1) * area + (size_t)(y + CORNER[c][1]) * dim[0] + (size_t)(x + CORNER[c][0]);
                    if (wgt[off[c]] == 0.0f) ok = false;                 /* computeLutIndex (:611-618) */
                }
                if (!ok) continue;
                for (int c = 0; c < 8; ++c) if (tsdf[off[c]] > iso) idx |= 1 << c;
                if (idx == 0 || idx == 255) continue;
                Vec3f ep[12];
                const int mask = edge_mask(idx);
                for (int e = 0; e < 12; ++e) {
                    if (!((mask >> e) & 1)) continue;
                    const int a = EDGE[e][0], b = EDGE[e][1];
                    ep[e] = interpolate(tsdf[off[a]], tsdf[off[b]], world(x + CORNER[a][0], y + CORNER[a][1], z + CORNER[a][2]),
                                        world(x + CORNER[b][0], y + CORNER[b][1], z + CORNER[b][2]), iso);
                }
                const std::vector<int>& t = triangles(idx);
                for (size_t i = 0; i + 2 < t.size(); i += 3) {
                    const Vec3f& p1 = ep[t[i]]; const Vec3f& p2 = ep[t[i + 1]]; const Vec3f& p3 = ep[t[i + 2]];
                    auto same = [](const Vec3f& a, const Vec3f& b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; };
                    if (same(p1, p2) || same(p1, p3) || same(p2, p3)) continue;     /* computeTriangles (:686-712) */
                    const int v0 = (int)vertices_.size();
                    vertices_.push_back(p1); vertices_.push_back(p2); vertices_.push_back(p3);
                    faces_.push_back({ v0, v0 + 1, v0 + 2 });
                }
            }
    }
    return true;
}

void MarchingCubes::setTriangles(const float* tris, size_t n_tris) {
    vertices_.clear();
    faces_.clear();
    vertices_.reserve(3 * n_tris);
    faces_.reserve(n_tris);
    for (size_t i = 0; i < n_tris; ++i) {
        const int v0 = (int)vertices_.size();
        for (int v = 0; v < 3; ++v) vertices_.push_back(Vec3f(tris[9 * i + 3 * v], tris[9 * i + 3 * v + 1], tris[9 * i + 3 * v + 2]));
        faces_.push_back({ v0, v0 + 1, v0 + 2 });
    }
}

void MarchingCubes::fill_table(int8_t out[256 * 16]) {
    for (int i = 0; i < 256 * 16; ++i) out[i] = GSDF_MC_TRI_TABLE[i];
}

/* LayeredMarchingCubesNoColor::savePly (:721-757): ASCII PLY, vertices then faces, numbers as `ofstream << float` prints them
 * (6 significant digits, %g).  The text is the same; it is produced with std::to_chars(general, 6) -- specified to give what
 * printf("%.6g") gives -- by a few threads into memory and written with one call each (a 10^5-face mesh is ~10^6 numbers:
 * 120 ms through operator<<, most of config C4's export time; ~10 ms this way). */
namespace {
inline char* put_float(char* p, float v) {
    auto r = std::to_chars(p, p + 32, v, std::chars_format::general, 6);
    return r.ptr;
}
inline char* put_int(char* p, int v) {
    auto r = std::to_chars(p, p + 16, v);
    return r.ptr;
}
}

bool MarchingCubes::savePly(const std::string& filename) const {
    if (vertices_.empty()) return false;
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "ply\nformat ascii 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\n"
                    "element face %d\nproperty list uchar int vertex_indices\nend_header\n", vertices_.size(), (int)faces_.size());
    const size_t nv = vertices_.size(), nf = faces_.size();
    const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    const size_t n_chunks = nv + nf < 4096 ? 1 : hw;
    std::vector<std::string> vtxt(n_chunks), ftxt(n_chunks);
    std::vector<std::thread> th;
    auto work = [&](size_t c) {
        const size_t v0 = nv * c / n_chunks, v1 = nv * (c + 1) / n_chunks;
        std::string& a = vtxt[c];
        a.resize((v1 - v0) * 3 * 16 + 16);
        char* p = &a[0];
        for (size_t i = v0; i < v1; ++i) {
            const Vec3f& v = vertices_[i];
            p = put_float(p, v[0]); *p++ = ' ';
            p = put_float(p, v[1]); *p++ = ' ';
            p = put_float(p, v[2]); *p++ = '\n';
        }
        a.resize((size_t)(p - &a[0]));
        const size_t f0 = nf * c / n_chunks, f1 = nf * (c + 1) / n_chunks;
        std::string& b = ftxt[c];
        b.resize((f1 - f0) * (2 + 3 * 12) + 16);
        p = &b[0];
        for (size_t i = f0; i < f1; ++i) {
            *p++ = '3';
            for (int k = 0; k < 3; ++k) { *p++ = ' '; p = put_int(p, faces_[i][k]); }
            *p++ = '\n';
        }
        b.resize((size_t)(p - &b[0]));
    };
    for (size_t c = 1; c < n_chunks; ++c) th.emplace_back(work, c);
    work(0);
    for (std::thread& t : th) t.join();
    bool ok = true;
    for (size_t c = 0; c < n_chunks; ++c) ok = ok && std::fwrite(vtxt[c].data(), 1, vtxt[c].size(), f) == vtxt[c].size();
    for (size_t c = 0; c < n_chunks; ++c) ok = ok && std::fwrite(ftxt[c].data(), 1, ftxt[c].size(), f) == ftxt[c].size();
    return std::fclose(f) == 0 && ok;
}
