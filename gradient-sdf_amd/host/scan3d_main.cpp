/*
 * Scan3D -- the depth-scanning CLI of the reference (cpp/depth_scanning/src/main_scan_3d.cpp) on
 * top of the MI355X engine.  Same flags (--input --results --pose-file --first --last --scan-type
 * --data-type --voxel-size --trunc --save-sdf), same frame loop (GT-pose fusion or track+fuse,
 * :208-281), same outputs (<results>_poses.txt in TUM format, <results>gradient_sdf_mesh_final.ply,
 * _cloud_final.ply, optional sdf text files, :285-311) and the Timer labels.  New flags:
 * --width/--height (the reference hard-codes 640x480, :183), --hash-capacity (log2 slots), --device.
 * Only --scan-type grad-sdf exists here: base-sdf is the comparison method, out of scope.
 */
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <string>

#include "MapGradPixelSdf.h"
#include "RigidOptimizer.h"
#include "Timer.h"
#include "img_loader.h"

namespace {
struct Options {
    std::string input, output = "../results/", pose_file = "pose.txt", stype = "map-gp", dtype;
    size_t first = 0, last = std::numeric_limits<size_t>::max();
    float voxel_size = 0.01f, trunc = 5.f;
    bool save_sdf = false;
    int width = 640, height = 480, capacity_log2 = 22, device = 0;
};

bool parse(int argc, char** argv, Options& o) {
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](std::string& dst) { if (i + 1 >= argc) return false; dst = argv[++i]; return true; };
        std::string v;
        if (a == "--save-sdf") { o.save_sdf = true; continue; }
        if (a == "-h" || a == "--help") {
            std::cout << "Hash Table-Based 3D Scanning (MI355X)\n  --input --results --pose-file --first --last --scan-type"
                         " --data-type --voxel-size --trunc --save-sdf --width --height --hash-capacity --device\n";
            std::exit(0);
        }
        if (!val(v)) { std::cerr << "missing value for " << a << std::endl; return false; }
        if (a == "--input") o.input = v;
        else if (a == "--results") o.output = v;
        else if (a == "--pose-file") o.pose_file = v;
        else if (a == "--first") o.first = std::stoul(v);
        else if (a == "--last") o.last = std::stoul(v);
        else if (a == "--scan-type") o.stype = v;
        else if (a == "--data-type") o.dtype = v;
        else if (a == "--voxel-size") o.voxel_size = std::stof(v);
        else if (a == "--trunc") o.trunc = std::stof(v);
        else if (a == "--width") o.width = std::stoi(v);
        else if (a == "--height") o.height = std::stoi(v);
        else if (a == "--hash-capacity") o.capacity_log2 = std::stoi(v);
        else if (a == "--device") o.device = std::stoi(v);
        else { std::cerr << "unknown option " << a << std::endl; return false; }
    }
    return true;
}
} // namespace

int main(int argc, char* argv[]) {
    Timer T;
    Options opt;
    if (!parse(argc, argv, opt)) return 1;

    if (opt.stype != "grad-sdf") {           /* the default "map-gp" is rejected like in the reference (:105-114) */
        std::cerr << "Your specified scan type is not supported (yet)." << std::endl;
        return 1;
    }
    std::unique_ptr<ImageLoader> loader;
    if (opt.dtype == "tum") loader.reset(new TumrgbdLoader(opt.input));
    else if (opt.dtype == "synth") loader.reset(new SynthLoader(opt.input));
    else {                                   /* printed / redwood loaders: formats not part of this build */
        std::cerr << "Your specified dataset type is not supported (yet)." << std::endl;
        return 1;
    }
    if (!loader->load_intrinsics("intrinsics.txt")) {
        std::cerr << "No intrinsics file found in " << opt.input << "!" << std::endl;
        return 1;
    }
    const Mat3f K = loader->K();
    std::cout << "K: " << std::endl;
    for (int r = 0; r < 3; ++r) std::cout << K(r, 0) << " " << K(r, 1) << " " << K(r, 2) << std::endl;

    std::vector<Mat4f> poses;
    bool GT_pose = false;
    if (!ImageLoader::load_pose(opt.input + opt.pose_file, poses)) std::cerr << "No GT poses are avaible!" << std::endl;
    else { std::cout << poses.size() << " GT poses are loaded!" << std::endl; GT_pose = true; }

    T.tic();
    NormalEstimator NEst(opt.width, opt.height, K, 2 * 5 + 1);                         /* :183 */
    T.toc("Init normal estimation");

    const float truncation = opt.trunc * opt.voxel_size;                              /* :191 */
    std::unique_ptr<MapGradPixelSdf> tSDF;
    std::unique_ptr<RigidPointOptimizer> pOpt;
    std::ofstream pose_file(opt.output + "_poses.txt");

    ColorImage color;
    DepthImage depth;
    for (size_t i = 0; i < opt.first; ++i) loader->load_next(color, depth);           /* :203-205 */

    for (size_t i = opt.first; i <= opt.last; ++i) {
        std::cout << "Working on frame: " << i << std::endl;
        T.tic();
        const bool loaded = loader->load_next(color, depth);
        if (!loaded) std::cerr << " -> Frame " << i << " could not be loaded!" << std::endl;
        T.toc("Load data");
        if (!loaded) break;
        if (depth.cols != opt.width || depth.rows != opt.height) {
            std::cerr << "frame size " << depth.cols << "x" << depth.rows << " differs from --width/--height" << std::endl;
            return 1;
        }
        if (i == opt.first) {
            T.tic();
            tSDF.reset(new MapGradPixelSdf(opt.voxel_size, truncation, opt.capacity_log2, opt.device));
            T.toc("Create Sdf");
            T.tic();
            if (GT_pose) tSDF->update(color, depth, K, SE3(poses[0]), &NEst);          /* poses[0] even if --first > 0 (:242) */
            else tSDF->setup(color, depth, K, &NEst);
            T.toc("Integrate depth data into Sdf");
            T.tic();
            pOpt.reset(new RigidPointOptimizer(tSDF.get()));
            T.toc("Create RigidOptimizer");
        } else if (GT_pose) {
            if (i >= poses.size()) break;
            T.tic();
            tSDF->update(color, depth, K, SE3(poses[i]), &NEst);
            T.toc("Integrate depth data into Sdf");
        } else {
            T.tic();
            const bool conv = pOpt->optimize(depth, K);
            T.toc("Point optimization");
            if (conv) {                                                                /* :261-265 */
                T.tic();
                tSDF->update(color, depth, K, pOpt->pose(), &NEst);
                T.toc("Integrate depth data into Sdf");
            }
        }
        /* timestamp tx ty tz qx qy qz qw (:268-280); the quaternion is Eigen::Quaternion(R) of the pose matrix */
        const Mat4f p = (GT_pose && i < poses.size()) ? poses[i] : pOpt->pose().matrix();
        float R[9], q[4];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = p(r, c);
        gsdf_R_to_quat(R, q);
        pose_file << loader->depth_timestamp() << " " << p(0, 3) << " " << p(1, 3) << " " << p(2, 3) << " "
                  << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << "\n";
    }
    pose_file.close();
    if (!tSDF) { std::cerr << "no frame was processed" << std::endl; return 1; }

    const std::string prefix = "gradient_sdf";
    T.tic();
    std::string filename = opt.output + prefix + "_mesh_final.ply";
    if (!tSDF->extract_mesh(filename)) std::cerr << "Could not save mesh to " << filename << "!" << std::endl;
    T.toc("Save mesh to disk");
    T.tic();
    filename = opt.output + prefix + "_cloud_final.ply";
    if (!tSDF->extract_pc(filename)) std::cerr << "Could not save point cloud to " << filename << "!" << std::endl;
    T.toc("Save point cloud to disk");
    if (opt.save_sdf) {
        T.tic();
        if (!tSDF->save_sdf(opt.output + prefix)) std::cerr << "could not save voxel grid info file " << opt.output + prefix << "!" << std::endl;
        T.toc("Save sdf txt files to disk");
    }
    return 0;
}
