#include "MarchingCubes.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <limits>
#include <unordered_map>

namespace {

/* corner c -> (dx, dy, dz), numbering of LayeredMarchingCubesNoColor::computeLutIndex (:599-606) */
const int CORNER[8][3] = { { 1, 1, 0 }, { 1, 0, 0 }, { 0, 0, 0 }, { 0, 1, 0 }, { 1, 1, 1 }, { 1, 0, 1 }, { 0, 0, 1 }, { 0, 1, 1 } };
/* edge e -> its two corners (the classic numbering: 0-3 bottom ring, 4-7 top ring, 8-11 verticals) */
const int EDGE[12][2] = { { 0, 1 }, { 1, 2 }, { 2, 3 }, { 3, 0 }, { 4, 5 }, { 5, 6 }, { 6, 7 }, { 7, 4 }, { 0, 4 }, { 1, 5 }, { 2, 6 }, { 3, 7 } };
/* faces: 4 corners in cyclic order */
const int FACE[6][4] = { { 0, 1, 2, 3 }, { 4, 5, 6, 7 }, { 0, 1, 5, 4 }, { 1, 2, 6, 5 }, { 2, 3, 7, 6 }, { 3, 0, 4, 7 } };

int edge_between(int a, int b) {
    for (int e = 0; e < 12; ++e)
        if ((EDGE[e][0] == a && EDGE[e][1] == b) || (EDGE[e][0] == b && EDGE[e][1] == a)) return e;
    return -1;
}

struct Tables {
    std::vector<int> tri[256];
    int mask[256];
    Tables() {
        for (int c = 0; c < 256; ++c) {
            mask[c] = 0;
            for (int e = 0; e < 12; ++e)
                if (((c >> EDGE[e][0]) & 1) != ((c >> EDGE[e][1]) & 1)) mask[c] |= 1 << e;
            /* contour segments on every face; each crossing edge gets exactly two neighbours */
            int nb[12][2];
            for (int e = 0; e < 12; ++e) nb[e][0] = nb[e][1] = -1;
            auto link = [&](int a, int b) {
                nb[a][nb[a][0] < 0 ? 0 : 1] = b;
                nb[b][nb[b][0] < 0 ? 0 : 1] = a;
            };
            for (int f = 0; f < 6; ++f) {
                int s[4], cross[4], n = 0;
                for (int i = 0; i < 4; ++i) s[i] = (c >> FACE[f][i]) & 1;
                for (int i = 0; i < 4; ++i) cross[i] = s[i] != s[(i + 1) & 3];     /* edge corner i -> i+1 */
                for (int i = 0; i < 4; ++i) n += cross[i];
                auto E = [&](int i) { return edge_between(FACE[f][i & 3], FACE[f][(i + 1) & 3]); };
                if (n == 2) {
                    int a = -1, b = -1;
                    for (int i = 0; i < 4; ++i) if (cross[i]) { if (a < 0) a = i; else b = i; }
                    link(E(a), E(b));
                } else if (n == 4) {
                    /* ambiguous face: cut off the corners that are set (depends on the face's signs only,
                     * so both cubes sharing the face agree) */
                    const int p = s[0] ? 0 : 1;                                     /* a set corner */
                    link(E(p + 3), E(p));
                    link(E(p + 1), E(p + 2));
                }
            }
            bool used[12] = { false };
            for (int e0 = 0; e0 < 12; ++e0) {
                if (!((mask[c] >> e0) & 1) || used[e0]) continue;
                std::vector<int> loop;
                int prev = -1, cur = e0;
                while (cur >= 0 && !used[cur]) {
                    used[cur] = true;
                    loop.push_back(cur);
                    const int nxt = nb[cur][0] != prev ? nb[cur][0] : nb[cur][1];
                    prev = cur;
                    cur = nxt;
                }
                if (loop.size() < 3) continue;
                /* orient the loop: normal towards the side of the UNSET corners (tsdf <= iso) */
                double P[12][3], g[3] = { 0, 0, 0 }, nrm[3] = { 0, 0, 0 };
                for (size_t i = 0; i < loop.size(); ++i) {
                    const int a = EDGE[loop[i]][0], b = EDGE[loop[i]][1];
                    const int set = ((c >> a) & 1) ? a : b, unset = set == a ? b : a;
                    for (int k = 0; k < 3; ++k) {
                        P[i][k] = 0.5 * (CORNER[a][k] + CORNER[b][k]);
                        g[k] += CORNER[unset][k] - CORNER[set][k];
                    }
                }
                for (size_t i = 0; i < loop.size(); ++i) {                          /* Newell */
                    const double* p = P[i];
                    const double* q = P[(i + 1) % loop.size()];
                    nrm[0] += (p[1] - q[1]) * (p[2] + q[2]);
                    nrm[1] += (p[2] - q[2]) * (p[0] + q[0]);
                    nrm[2] += (p[0] - q[0]) * (p[1] + q[1]);
                }
                if (nrm[0] * g[0] + nrm[1] * g[1] + nrm[2] * g[2] < 0) std::reverse(loop.begin(), loop.end());
                for (size_t i = 1; i + 1 < loop.size(); ++i) {
                    tri[c].push_back(loop[0]); tri[c].push_back(loop[i]); tri[c].push_back(loop[i + 1]);
                }
            }
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
    for (int c = 0; c < 256; ++c) {
        const std::vector<int>& t = triangles(c);
        for (int k = 0; k < 16; ++k) out[16 * c + k] = k < (int)t.size() && k < 15 ? (int8_t)t[k] : (int8_t)-1;
    }
}

bool MarchingCubes::savePly(const std::string& filename) const {
    if (vertices_.empty()) return false;
    std::ofstream f(filename.c_str());
    if (!f.is_open()) return false;
    f << "ply\nformat ascii 1.0\nelement vertex " << vertices_.size() << "\n"
      << "property float x\nproperty float y\nproperty float z\n"
      << "element face " << (int)faces_.size() << "\nproperty list uchar int vertex_indices\nend_header\n";
    for (const Vec3f& v : vertices_) f << v[0] << " " << v[1] << " " << v[2] << "\n";
    for (const auto& t : faces_) f << "3 " << t[0] << " " << t[1] << " " << t[2] << "\n";
    return true;
}
