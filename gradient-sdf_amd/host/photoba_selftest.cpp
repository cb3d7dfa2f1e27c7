/*
 * photoba_selftest -- runs PhotometricOptimizer (host/PhotometricOptimizer.h, the facade for
 * cpp/include/ps_optimizer/PhotometricOptimizer.h:68-186) once, from C++, through every public method: the header was only
 * compile-checked before (every C5 test drives gsdf_ba_* from ctypes).  Needs a GPU; tests/test_photoba.py writes the inputs
 * and compares what this prints with the same steps driven through the C-ABI directly.
 *
 *   photoba_selftest <dir> W H n voxel_size trunc_voxels
 *   <dir>/K.bin (9 f32)  depth.bin (n*H*W f32)  images.bin (n*H*W*3 f32, BGR)  poses_true.bin / poses_start.bin (n*16 f32)
 */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "Image.h"
#include "MapGradPixelSdf.h"
#include "PhotometricOptimizer.h"

static bool read_bin(const std::string& path, std::vector<float>& v, size_t n) {
    std::ifstream f(path, std::ios::binary);
    v.resize(n);
    return f.good() && f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(float))).good();
}

int main(int argc, char** argv) {
    if (argc < 7) { std::cerr << "usage: photoba_selftest <dir> W H n voxel_size trunc_voxels" << std::endl; return 2; }
    const std::string dir = std::string(argv[1]) + "/";
    const int W = atoi(argv[2]), H = atoi(argv[3]), n = atoi(argv[4]);
    const float vs = (float)atof(argv[5]), trunc = (float)atof(argv[6]);
    const size_t N = (size_t)W * H;
    std::vector<float> Kb, depth, images, Pt, Ps;
    if (!read_bin(dir + "K.bin", Kb, 9) || !read_bin(dir + "depth.bin", depth, n * N) || !read_bin(dir + "images.bin", images, n * N * 3) ||
        !read_bin(dir + "poses_true.bin", Pt, (size_t)n * 16) || !read_bin(dir + "poses_start.bin", Ps, (size_t)n * 16)) {
        std::cerr << "photoba_selftest: cannot read the inputs in " << dir << std::endl;
        return 2;
    }
    try {
        Mat3f K;
        for (int i = 0; i < 9; ++i) K.m[i] = Kb[i];
        NormalEstimator NEst(W, H, K, 2 * 5 + 1);
        MapGradPixelSdf map(vs, trunc * vs, 20, 0, 20);
        map.enable_vis(64);                                            /* vis_ -- MapGradPixelSdf.cpp:113-115 */
        ColorImage color;
        std::vector<Mat4f> truth((size_t)n), start((size_t)n);
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < 16; ++k) { truth[i].m[k] = Pt[(size_t)i * 16 + k]; start[i].m[k] = Ps[(size_t)i * 16 + k]; }
            DepthImage d;
            d.rows = H; d.cols = W;
            d.buf.assign(depth.begin() + (long)(i * N), depth.begin() + (long)((i + 1) * N));
            map.update(color, d, K, SE3(truth[i]), &NEst);             /* main_photo_ba.cpp:237-243 */
        }
        std::vector<std::shared_ptr<ColorImageF>> imgs;
        std::vector<int> keyframes;
        for (int i = 0; i < n; ++i) {
            auto im = std::make_shared<ColorImageF>();
            im->rows = H; im->cols = W;
            im->bgr.assign(images.begin() + (long)(i * N * 3), images.begin() + (long)((i + 1) * N * 3));
            imgs.push_back(im);
            keyframes.push_back(i);                                    /* main_photo_ba.cpp:249: keyframe id = sequence index */
        }
        OptSettings settings;
        settings.max_it = 4;
        PhotometricOptimizer opt(&map, settings);                      /* main_photo_ba.cpp:300-306 */
        opt.setImages(imgs);
        opt.setPoses(start);
        opt.setKeyframes(keyframes);
        const float E0 = opt.getEnergy();
        opt.solvePose(settings.damping);
        const float E1 = opt.getEnergy();
        opt.solveDist(settings.damping);
        const float E2 = opt.getEnergy();
        std::printf("voxels %lld frames %lld\n", (long long)map.size(), (long long)map.frame_counter());
        std::printf("steps %.9g %.9g %.9g\n", E0, E1, E2);
        for (int i = 0; i < n; ++i) {
            std::printf("pose_after_step %d", i);
            for (int k = 0; k < 16; ++k) std::printf(" %.9g", opt.poses()[(size_t)i].m[k]);
            std::printf("\n");
        }
        std::vector<float> energies;
        const bool conv = opt.optimize(&energies);
        std::printf("optimize %d", conv ? 1 : 0);
        for (float e : energies) std::printf(" %.9g", e);
        std::printf("\n");
        for (int i = 0; i < n; ++i) {
            std::printf("pose_final %d", i);
            for (int k = 0; k < 16; ++k) std::printf(" %.9g", opt.poses()[(size_t)i].m[k]);
            std::printf("\n");
        }
    } catch (const std::exception& e) {
        std::cerr << "photoba_selftest: " << e.what() << std::endl;
        return 1;
    }
    std::printf("photoba_selftest: OK\n");
    return 0;
}
