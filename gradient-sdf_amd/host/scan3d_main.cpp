/*
 * Scan3D -- the depth-scanning CLI of the reference (cpp/depth_scanning/src/main_scan_3d.cpp) on
 * top of the MI355X engine.  Same flags (--input --results --pose-file --first --last --scan-type
 * --data-type --voxel-size --trunc --save-sdf), same frame loop semantics (GT-pose fusion or track+fuse,
 * :208-281), same outputs (<results>_poses.txt in TUM format, <results>gradient_sdf_mesh_final.ply,
 * _cloud_final.ply, optional sdf text files, :285-311) and the Timer labels.  New flags:
 *   --width/--height (the reference hard-codes 640x480, :183), --hash-capacity (log2 voxel records to start with),
 *   --hash-max-capacity (default 28: the table doubles by itself up to this, like the reference's map grows), --device,
 *   --sync            the reference's call structure literally: one blocking optimize() / update() per frame through the
 *                     facade classes (host-pointer entries; every call copies the frame and waits for the GPU);
 *   (default)         the device-resident loop: PNG decode on host threads into page-locked buffers, asynchronous copy
 *                     to HBM, gsdf_track_and_fuse_dev / gsdf_update_dev enqueued per frame, poses read back once at
 *                     the end (frame_pipeline.h) -- same poses and map as --sync;
 *   --decode-threads  host threads of the PNG decode (default: half the cores, at most 8);
 *   --gpus N          GT-pose fusion sharded over N ranks, one process per GPU (main_scan_3d.cpp:250-254 has no frame-to-
 *                     frame dependence): contiguous frame ranges, ONE exchange (gsdf_merge_allreduce: RCCL all-reduce of
 *                     the per-voxel sums over the union of blocks), then rank 0 writes the outputs.  Tracked mode does
 *                     not shard (frame i needs the map of all frames < i);
 *   --transport shm   the exchange through host shared memory instead of RCCL: N ranks on ONE device (test boxes).
 * Only --scan-type grad-sdf exists here: base-sdf is the comparison method, out of scope.
 */
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <thread>

#include "MapGradPixelSdf.h"
#include "RigidOptimizer.h"
#include "Timer.h"
#include "frame_pipeline.h"
#include "img_loader.h"
#include "shm_collective.h"

namespace {
struct Options {
    std::string input, output = "../results/", pose_file = "pose.txt", stype = "map-gp", dtype;
    size_t first = 0, last = std::numeric_limits<size_t>::max();
    float voxel_size = 0.01f, trunc = 5.f;
    bool save_sdf = false, sync = false;
    int width = 640, height = 480, capacity_log2 = 22, device = 0;
    int max_capacity_log2 = 28;          /* the table doubles by itself up to this (the reference's map has no capacity) */
    int gpus = 1, rank = -1, decode_threads = 0;
    std::string transport = "rccl", rendezvous;
};

bool parse(int argc, char** argv, Options& o) {
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](std::string& dst) { if (i + 1 >= argc) return false; dst = argv[++i]; return true; };
        std::string v;
        if (a == "--save-sdf") { o.save_sdf = true; continue; }
        if (a == "--sync") { o.sync = true; continue; }
        if (a == "-h" || a == "--help") {
            std::cout << "Hash Table-Based 3D Scanning (MI355X)\n  --input --results --pose-file --first --last --scan-type"
                         " --data-type --voxel-size --trunc --save-sdf --width --height --hash-capacity --hash-max-capacity --device"
                         " --sync --decode-threads --gpus --transport rccl|shm\n";
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
        else if (a == "--hash-max-capacity") o.max_capacity_log2 = std::stoi(v);
        else if (a == "--device") o.device = std::stoi(v);
        else if (a == "--gpus") o.gpus = std::stoi(v);
        else if (a == "--rank") o.rank = std::stoi(v);                 /* set by the launcher */
        else if (a == "--rendezvous") o.rendezvous = v;                /* set by the launcher */
        else if (a == "--transport") o.transport = v;
        else if (a == "--decode-threads") o.decode_threads = std::stoi(v);
        else { std::cerr << "unknown option " << a << std::endl; return false; }
    }
    return true;
}

/* --gpus N: one process per rank (started before anything touches the GPU); the ranks meet in a /dev/shm segment */
int launch_ranks(int argc, char** argv, const Options& opt) {
    const std::string name = "gsdf_scan3d_" + std::to_string((long)getpid());
    if (!ShmCollective::create(name, opt.gpus)) { std::cerr << "cannot create the rendezvous segment" << std::endl; return 1; }
    std::remove(("/dev/shm/" + name + ".id").c_str());
    std::vector<pid_t> kids;
    for (int r = 0; r < opt.gpus; ++r) {
        const pid_t pid = fork();
        if (pid == 0) {
            std::vector<std::string> args(argv, argv + argc);
            args.push_back("--rank"); args.push_back(std::to_string(r));
            args.push_back("--rendezvous"); args.push_back(name);
            std::vector<char*> av;
            for (std::string& a : args) av.push_back(&a[0]);
            av.push_back(nullptr);
            execv("/proc/self/exe", av.data());
            std::perror("execv");
            _exit(127);
        }
        if (pid < 0) {
            /* the ranks already started wait for peers that will never come: release, stop and reap them, remove the segment */
            std::perror("fork");
            ShmCollective::abort_all(name);
            for (pid_t k : kids) kill(k, SIGKILL);                             /* exact pids */
            for (pid_t k : kids) { int st = 0; (void)waitpid(k, &st, 0); }
            ShmCollective::destroy(name);
            std::remove(("/dev/shm/" + name + ".id").c_str());
            return 1;
        }
        kids.push_back(pid);
    }
    /* Wait for the ranks.  The first one that fails takes the others with it: the abort flag releases ranks that wait in a
     * shared-memory barrier, and ranks that are stuck inside an RCCL collective (whose peer is gone) are killed after a short
     * grace period -- by their exact pids. */
    int worst = 0;
    size_t left = kids.size();
    bool stopping = false;                 /* the abort / grace period / kill sequence runs once; afterwards the ranks are only reaped */
    while (left > 0) {
        int st = 0;
        const pid_t pid = waitpid(-1, &st, 0);
        if (pid < 0) break;
        for (pid_t& k : kids) if (k == pid) { k = -1; --left; }
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128;
        if (code > worst && !(stopping && !WIFEXITED(st))) worst = code;      /* ranks this launcher killed do not set the exit code */
        if (code != 0 && left > 0 && !stopping) {
            stopping = true;
            std::cerr << "a rank exited with code " << code << ": stopping the other ranks" << std::endl;
            ShmCollective::abort_all(name);
            for (int t = 0; t < 40 && left > 0; ++t) {                       /* 2 s to leave on their own */
                const pid_t q = waitpid(-1, &st, WNOHANG);
                if (q > 0) { for (pid_t& k : kids) if (k == q) { k = -1; --left; } }
                else std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
            for (pid_t k : kids) if (k > 0) kill(k, SIGKILL);
        }
    }
    ShmCollective::destroy(name);
    std::remove(("/dev/shm/" + name + ".id").c_str());
    return worst;
}

/* [lo, hi) of `rank` among `world` contiguous shards of n items */
void shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
    const size_t base = n / (size_t)world, rem = n % (size_t)world;
    *lo = (size_t)rank * base + std::min<size_t>((size_t)rank, rem);
    *hi = *lo + base + ((size_t)rank < rem ? 1 : 0);
}

void write_pose_line(std::ofstream& f, const std::string& ts, const Mat4f& p) {
    /* timestamp tx ty tz qx qy qz qw (main_scan_3d.cpp:268-280); the quaternion is Eigen::Quaternion(R) of the pose matrix */
    float R[9], q[4];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = p(r, c);
    gsdf_R_to_quat(R, q);
    f << ts << " " << p(0, 3) << " " << p(1, 3) << " " << p(2, 3) << " " << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << "\n";
}

/* --gpus N: what the exchange needs, set up BEFORE the frame loop (outside anything timed, and so that a rank that cannot
 * join fails before its peers have fused their shards): the rendezvous segment, and for RCCL the communicator -- rank 0
 * publishes the id as a file next to the segment first thing, the other ranks poll for it (giving up when the abort flag
 * is raised or after the barrier timeout). */
struct Exchange {
    std::unique_ptr<ShmCollective> seg;
    void* comm = nullptr;
    ~Exchange() { if (comm) gsdf_rccl_comm_destroy(comm); }
};
bool exchange_prepare(Exchange& ex, const Options& opt, int device) {
    ex.seg.reset(new ShmCollective(opt.rendezvous, opt.gpus, opt.rank));
    if (!ex.seg->ok()) { std::cerr << "rank " << opt.rank << ": rendezvous segment missing" << std::endl; return false; }
    if (opt.transport == "shm") return true;
    const std::string idf = "/dev/shm/" + opt.rendezvous + ".id";
    char id[128];
    if (opt.rank == 0) {
        if (gsdf_rccl_unique_id(id) != GSDF_OK) { std::cerr << "RCCL: " << gsdf_last_error() << std::endl; return false; }
        { std::ofstream f(idf + ".tmp", std::ios::binary); f.write(id, 128); }
        std::rename((idf + ".tmp").c_str(), idf.c_str());
    } else {
        bool got = false;
        const auto t0 = std::chrono::steady_clock::now();
        while (!got && !ex.seg->aborted() && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 300.0) {
            std::ifstream f(idf, std::ios::binary);
            got = (bool)f.read(id, 128);
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (!got) { std::cerr << "rank " << opt.rank << ": no RCCL id from rank 0" << std::endl; return false; }
    }
    if (gsdf_rccl_comm_init(&ex.comm, opt.gpus, id, opt.rank, device) != GSDF_OK) { std::cerr << "RCCL: " << gsdf_last_error() << std::endl; return false; }
    return true;
}

/* the exchange step of --gpus N */
bool exchange(Exchange& ex, gsdf_ctx* ctx, const Options& opt) {
    int64_t n_blocks = 0, bytes = 0;
    if (opt.transport == "shm") {
        gsdf_collective ops = ex.seg->ops();
        if (gsdf_merge_allreduce_with(ctx, &ops, &n_blocks, &bytes) != GSDF_OK) { std::cerr << "exchange: " << gsdf_last_error() << std::endl; return false; }
    } else {
        Timer T;
        T.tic();
        const int rc = gsdf_merge_allreduce(ctx, ex.comm, &n_blocks, &bytes);
        if (opt.rank == 0) T.toc("Exchange without communicator set-up");
        if (rc != GSDF_OK) { std::cerr << "exchange: " << gsdf_last_error() << std::endl; return false; }
    }
    if (opt.rank == 0)
        std::cout << "Exchanged " << n_blocks << " voxel blocks (" << bytes / 1048576.0 << " MiB all-reduced) among " << opt.gpus << " ranks" << std::endl;
    return true;
}

int run(int argc, char* argv[], Options& opt);
} // namespace

int main(int argc, char* argv[]) {
    Options opt;
    if (!parse(argc, argv, opt)) return 1;
    if (opt.gpus > 1 && opt.rank < 0) return launch_ranks(argc, argv, opt);      /* the launcher itself never touches the GPU */
    const int rc = run(argc, argv, opt);
    /* a rank that leaves early must not leave its peers waiting for it (barriers see the flag; the launcher stops ranks
     * that hang inside RCCL) */
    if (rc != 0 && opt.gpus > 1 && !opt.rendezvous.empty()) ShmCollective::abort_all(opt.rendezvous);
    return rc;
}

namespace {
int run(int, char**, Options& opt) {
    Timer T;
    const bool sharded = opt.gpus > 1;
    const bool lead = !sharded || opt.rank == 0;                                  /* writes the outputs */
    if (sharded && opt.transport != "shm") opt.device = opt.rank;                 /* one GPU per rank */

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
    if (lead) {
        std::cout << "K: " << std::endl;
        for (int r = 0; r < 3; ++r) std::cout << K(r, 0) << " " << K(r, 1) << " " << K(r, 2) << std::endl;
    }

    std::vector<Mat4f> poses;
    bool GT_pose = false;
    if (!ImageLoader::load_pose(opt.input + opt.pose_file, poses)) std::cerr << "No GT poses are avaible!" << std::endl;
    else { if (lead) std::cout << poses.size() << " GT poses are loaded!" << std::endl; GT_pose = true; }
    if (sharded && !GT_pose) {
        std::cerr << "--gpus needs GT poses: tracked frames depend on the map of all earlier frames and cannot be sharded" << std::endl;
        return 1;
    }

    T.tic();
    NormalEstimator NEst(opt.width, opt.height, K, 2 * 5 + 1);                         /* :183 */
    if (lead) T.toc("Init normal estimation");

    const float truncation = opt.trunc * opt.voxel_size;                              /* :191 */
    std::unique_ptr<MapGradPixelSdf> tSDF;
    std::unique_ptr<RigidPointOptimizer> pOpt;

    if (opt.sync && !sharded) {
        /* ---- the reference's call structure: blocking optimize() / update() per frame through the facade ---- */
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
                tSDF.reset(new MapGradPixelSdf(opt.voxel_size, truncation, opt.capacity_log2, opt.device, std::max(opt.capacity_log2, opt.max_capacity_log2)));
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
            write_pose_line(pose_file, loader->depth_timestamp(), (GT_pose && i < poses.size()) ? poses[i] : pOpt->pose().matrix());
        }
        pose_file.close();
        if (!tSDF) { std::cerr << "no frame was processed" << std::endl; return 1; }
    } else {
        /* ---- the device-resident loop ------------------------------------------------------------------------
         * The frame list first (the loaders advance through their index files without decoding), then: decode on host
         * threads -> page-locked buffer -> asynchronous copy to HBM -> the frame's kernels, all enqueued; the host
         * never waits for a frame.  Poses of tracked frames come back in ONE read of the device's frame log. */
        std::vector<FrameEntry> all;
        {
            std::string f, ts;
            for (size_t i = 0; i < opt.first; ++i) if (!loader->next_entry(f, ts)) break;  /* :203-205 */
            for (size_t i = opt.first; i <= opt.last; ++i) {
                if (GT_pose && i > opt.first && i >= poses.size()) break;                  /* :251 */
                if (!loader->next_entry(f, ts)) break;
                all.push_back(FrameEntry{ f, ts });
            }
        }
        if (all.empty()) { std::cerr << " -> Frame " << opt.first << " could not be loaded!" << std::endl << "no frame was processed" << std::endl; return 1; }
        size_t lo = 0, hi = all.size();
        if (sharded) shard_range(all.size(), opt.rank, opt.gpus, &lo, &hi);
        Exchange ex;
        if (sharded && !exchange_prepare(ex, opt, opt.device)) return 1;
        T.tic();
        tSDF.reset(new MapGradPixelSdf(opt.voxel_size, truncation, opt.capacity_log2, opt.device, std::max(opt.capacity_log2, opt.max_capacity_log2)));
        if (lead) T.toc("Create Sdf");
        tSDF->prepare(opt.width, opt.height, K, &NEst);
        gsdf_ctx* ctx = tSDF->handle();
        if (sharded) (void)gsdf_merge_prepare(ctx, opt.gpus);         /* the exchange's scratch and sort kernels: set-up, like the communicator */
        T.tic();
        pOpt.reset(new RigidPointOptimizer(tSDF.get()));
        if (lead) T.toc("Create RigidOptimizer");
        double el = 0., el_all = 0.;
        {
            /* Two clocks.  `el_all` starts BEFORE the staging buffers are created and the decoder threads start (they decode up to
             * `slots` frames ahead at once), so every frame's load is inside it, as in the reference's loop (main_scan_3d.cpp:213);
             * it is the number to compare.  `el` starts behind that set-up (page-locked allocations: 15-20 ms) and is the steady
             * state of the loop; on a short stream the frames decoded ahead make it optimistic.  Both stop before the buffers are
             * released. */
            const auto t_all = std::chrono::steady_clock::now();
            FramePipeline pipe(ctx, loader.get(), std::vector<FrameEntry>(all.begin() + (long)lo, all.begin() + (long)hi), opt.width,
                               opt.height, opt.decode_threads);
            const auto t_loop = std::chrono::steady_clock::now();
            for (size_t j = lo; j < hi; ++j) {
                const size_t i = opt.first + j;                       /* frame number as in the reference's loop */
                if (lead) std::cout << "Working on frame: " << i << "\n";
                T.tic();
                size_t got = 0;
                const float* d = pipe.next(&got);
                if (lead) T.toc("Load data");
                if (!d) { std::cerr << " -> Frame " << i << " could not be loaded! " << pipe.error() << std::endl; break; }
                T.tic();
                int rc;
                if (j == 0 || GT_pose) {
                    /* first frame: poses[0] (even if --first > 0, :242) or setup() at the identity (Sdf.h:119-121) */
                    const SE3 p = GT_pose ? SE3(poses[j == 0 ? 0 : i]) : SE3();
                    const Mat3f R = p.rotationMatrix();
                    const Vec3f t = p.translation();
                    rc = gsdf_update_dev(ctx, d, R.data(), t.data());                                  /* :242-243, :252 */
                    if (lead) T.toc("Integrate depth data into Sdf");
                } else {
                    /* :258-265; the successor is named when its image is already in HBM (normals in this frame's fusion tail) */
                    rc = gsdf_track_and_fuse_ahead_dev(ctx, d, pipe.next_ready(), K.data(), pOpt->num_iterations(), pOpt->conv_threshold(), pOpt->damping());
                    if (lead) T.toc("Point optimization + integration (enqueued)");
                }
                if (rc != GSDF_OK || !pipe.submitted()) { std::cerr << "frame " << i << ": " << gsdf_last_error() << std::endl; return 1; }
            }
            if (gsdf_sync(ctx) != GSDF_OK) { std::cerr << "engine: " << gsdf_last_error() << std::endl; return 1; }
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
            el_all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
        }
        if (lead) {
            std::cout << "Current frame counter: " << tSDF->frame_counter() << std::endl;
            std::cout << "---------- " << (hi - lo) << " frames loaded + " << (GT_pose ? "fused" : "tracked + fused") << ": " << el_all << "s ("
                      << (double)(hi - lo) / el_all << " frames per second" << (sharded ? ", this rank" : "") << "); behind the set-up of the "
                      << "staging buffers and decoders: " << el << "s (" << (double)(hi - lo) / el << " frames per second)." << std::endl;
        }
        if (sharded) {
            T.tic();
            if (!exchange(ex, ctx, opt)) return 1;
            if (lead) T.toc("Exchange voxel sums between the ranks");
        }
        if (lead) {
            std::ofstream pose_file(opt.output + "_poses.txt");
            if (GT_pose) {
                for (size_t j = 0; j < all.size(); ++j) write_pose_line(pose_file, all[j].timestamp, poses[opt.first + j < poses.size() ? opt.first + j : poses.size() - 1]);
            } else {
                std::vector<float> log(all.size() * 10);
                int64_t n = 0;
                if (gsdf_read_frame_log(ctx, log.data(), (int64_t)all.size(), &n) != GSDF_OK) { std::cerr << gsdf_last_error() << std::endl; return 1; }
                write_pose_line(pose_file, all[0].timestamp, SE3().matrix());                          /* pOpt->pose() before the first optimize() */
                for (int64_t r = 0; r < n && (size_t)(r + 1) < all.size(); ++r) {
                    write_pose_line(pose_file, all[(size_t)r + 1].timestamp, SE3::from_pose7(&log[(size_t)r * 10]).matrix());
                    if (log[(size_t)r * 10 + 7] > 0.f) std::cout << "frame " << opt.first + (size_t)r + 1 << " ... Convergence after " << (int)log[(size_t)r * 10 + 8] - 1 << " iterations!" << std::endl;
                }
            }
        }
    }
    if (!lead) return 0;

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
} // namespace
