/* CPU-only self test of the host-side pieces that need no GPU: PNG round trip, pose parsing,
 * SE3 helpers, the marching-cubes case tables (watertight sphere), the PLY writer's text, and the abortable rendezvous of
 * `Scan3D --gpus N`.  Run by tests/. */
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <thread>
#include <vector>

#include "MarchingCubes.h"
#include "img_loader.h"
#include "png16.h"
#include "shm_collective.h"

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
        /* every scanline filter, mixed filters + two IDAT chunks, and the decoder threads' direct-to-float path */
        for (int mode = 1; mode <= 5; ++mode) {
            CHECK(png_write_gray16(dir + "/tf.png", W, H, px.data(), mode));
            PngImage im2;
            CHECK(png_read(dir + "/tf.png", im2));
            CHECK(im2.width == W && im2.height == H && im2.first_channel == px);
            std::vector<float> fl((size_t)W * H, -1.f);
            CHECK(png_read_scaled(dir + "/tf.png", fl.data(), W, H, 0.0002f));
            bool same = true;
            for (int i = 0; i < W * H; ++i) same = same && fl[i] == (float)px[i] * 0.0002f;
            CHECK(same);
            std::string err;
            CHECK(!png_read_scaled(dir + "/tf.png", fl.data(), W + 1, H, 0.0002f, &err) && err.find("differs") != std::string::npos);
        }
        /* a corrupt header (width and height patched to 2^31 - 1 / 2^30): rejected by its numbers -- against the caller's frame
         * size right behind the IHDR, against the allocation bound otherwise -- before anything is sized by them */
        {
            CHECK(png_write_gray16(dir + "/bad.png", W, H, px.data()));
            std::fstream f(dir + "/bad.png", std::ios::in | std::ios::out | std::ios::binary);
            const unsigned char huge[8] = { 0x7f, 0xff, 0xff, 0xff, 0x40, 0x00, 0x00, 0x00 };
            f.seekp(16);                                    /* IHDR data: width, height */
            f.write((const char*)huge, 8);
            f.close();
            std::vector<float> fl((size_t)W * H, -1.f);
            std::string err;
            CHECK(!png_read_scaled(dir + "/bad.png", fl.data(), W, H, 0.0002f, &err) && err.find("differs") != std::string::npos);
            PngImage im3;
            err.clear();
            CHECK(!png_read(dir + "/bad.png", im3, &err) && err.find("too large") != std::string::npos);
        }
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
        /* the PLY text is what `ofstream << float` prints (LayeredMarchingCubesNoColor::savePly, :721-757) */
        std::ostringstream want;
        want << "ply\nformat ascii 1.0\nelement vertex " << mc.vertices().size() << "\nproperty float x\nproperty float y\nproperty float z\n"
             << "element face " << (int)mc.faces().size() << "\nproperty list uchar int vertex_indices\nend_header\n";
        for (const Vec3f& v : mc.vertices()) want << v[0] << " " << v[1] << " " << v[2] << "\n";
        for (const auto& t : mc.faces()) want << "3 " << t[0] << " " << t[1] << " " << t[2] << "\n";
        std::ifstream got(dir + "/sphere.ply", std::ios::binary);
        std::stringstream gs;
        gs << got.rdbuf();
        CHECK(gs.str() == want.str());
    }
    {   /* the rendezvous of `Scan3D --gpus N`: a barrier that a failing rank can abort (here: two threads as the two ranks) */
        const std::string name = "gsdf_selftest_" + std::to_string((long)getpid());
        CHECK(ShmCollective::create(name, 2));
        {
            ShmCollective a(name, 2, 0), b(name, 2, 1);
            CHECK(a.ok() && b.ok());
            bool rb = false;
            std::thread t([&] { rb = b.barrier(); });
            CHECK(a.barrier());
            t.join();
            CHECK(rb);
            /* all-gather through the segment */
            const int va = 11, vb = 22;
            int ra[2] = { 0, 0 }, rb2[2] = { 0, 0 };
            gsdf_collective ca = a.ops(), cb = b.ops();
            int rc_b = 1;
            std::thread t2([&] { rc_b = cb.allgather(cb.user, &vb, rb2, sizeof(int)); });
            CHECK(ca.allgather(ca.user, &va, ra, sizeof(int)) == 0);
            t2.join();
            CHECK(rc_b == 0 && ra[0] == 11 && ra[1] == 22 && rb2[0] == 11 && rb2[1] == 22);
            /* rank 1 gives up: rank 0's barrier returns false instead of waiting for it */
            bool r0 = true;
            std::thread t3([&] { r0 = a.barrier(); });
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            b.abort();
            t3.join();
            CHECK(!r0 && a.aborted());
            CHECK(!a.barrier());                                   /* and stays aborted */
        }
        ShmCollective::destroy(name);
    }
    std::printf(fails ? "host_selftest: %d FAILED\n" : "host_selftest: OK\n", fails);
    return fails ? 1 : 0;
}
