/* CPU-only self test of the host-side pieces that need no GPU: PNG round trip, pose parsing,
 * SE3 helpers and the marching-cubes case tables (watertight sphere).  Run by tests/. */
#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <vector>

#include "MarchingCubes.h"
#include "img_loader.h"
#include "png16.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    {   /* PNG 16-bit round trip */
        const int W = 37, H = 23;
        std::vector<uint16_t> px(W * H);
        for (int i = 0; i < W * H; ++i) px[i] = (uint16_t)((i * 2654435761u) >> 13);
        CHECK(png_write_gray16(dir + "/t.png", W, H, px.data()));
        PngImage im;
        CHECK(png_read(dir + "/t.png", im));
        CHECK(im.width == W && im.height == H && im.bit_depth == 16 && im.first_channel == px);
    }
    {   /* pose file */
        std::ofstream f(dir + "/pose.txt");
        f << "001 0.1275 -2.0035 -0.125 0.022285 -0.706756 -0.706756 0.022285\n";
        f.close();
        std::vector<Mat4f> poses;
        CHECK(ImageLoader::load_pose(dir + "/pose.txt", poses) && poses.size() == 1);
        const SE3 s(poses[0]);
        const Mat3f R = s.rotationMatrix();
        double err = 0;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) err = std::fmax(err, std::fabs(R(r, c) - poses[0](r, c)));
        CHECK(err < 1e-5);
        CHECK(std::fabs(s.translation()[1] + 2.0035f) < 1e-6f);
    }
    {   /* marching cubes: every case's triangles use exactly the sign-changing edges; sphere is watertight */
        for (int c = 1; c < 255; ++c) {
            int used = 0;
            for (int e : MarchingCubes::triangles(c)) used |= 1 << e;
            CHECK(used == MarchingCubes::edge_mask(c));
            CHECK(MarchingCubes::triangles(c).size() % 3 == 0 && MarchingCubes::triangles(c).size() <= 15);
        }
        const int N = 24;
        const float vs = 0.05f;
        std::vector<int32_t> keys;
        std::vector<float> pay;
        for (int z = -N; z <= N; ++z) for (int y = -N; y <= N; ++y) for (int x = -N; x <= N; ++x) {
            const float d = std::sqrt((float)(x * x + y * y + z * z)) * vs - 0.8f;
            keys.push_back(x); keys.push_back(y); keys.push_back(z);
            pay.push_back(d); pay.push_back(0); pay.push_back(0); pay.push_back(1); pay.push_back(1.f);
        }
        MarchingCubes mc(vs);
        CHECK(mc.computeIsoSurface(keys, pay, 0.f));
        CHECK(mc.faces().size() > 1000);
        std::map<std::pair<std::array<int, 3>, std::array<int, 3>>, int> edges;    /* undirected edge -> count */
        auto q = [&](const Vec3f& v) { return std::array<int, 3>{ (int)std::lround(v[0] * 1e5), (int)std::lround(v[1] * 1e5), (int)std::lround(v[2] * 1e5) }; };
        double rmax = 0, rmin = 1e9;
        for (const auto& t : mc.faces())
            for (int k = 0; k < 3; ++k) {
                auto a = q(mc.vertices()[t[k]]), b = q(mc.vertices()[t[(k + 1) % 3]]);
                if (b < a) std::swap(a, b);
                edges[{ a, b }]++;
                const Vec3f& v = mc.vertices()[t[k]];
                const double r = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                rmax = std::fmax(rmax, r); rmin = std::fmin(rmin, r);
            }
        int open = 0;
        for (const auto& e : edges) if (e.second != 2) ++open;
        CHECK(open == 0);
        CHECK(rmax < 0.81 && rmin > 0.79);
        CHECK(mc.savePly(dir + "/sphere.ply"));
    }
    std::printf(fails ? "host_selftest: %d FAILED\n" : "host_selftest: OK\n", fails);
    return fails ? 1 : 0;
}
