/*
 * gsdf.h -- C-ABI of the MI355X-native Gradient-SDF engine (libgsdf.so).
 *
 * This is the drop-in boundary for the hot path of c-sommer/gradient-sdf
 * (cpp/depth_scanning): per-voxel TSDF+gradient fusion, the voxel-hash point
 * query and the 6-DoF SDF-tracking normal-equation reduction.  The reference has
 * no FFI / plugin layer: its seam is two C++ virtual interfaces and one POD
 * (SURVEY.md 8b).  Every entry point below names the reference interface it
 * replaces (paths relative to /root/reference/cpp/include/).  The C++ facade in
 * gradient-sdf_amd/host/ (MapGradPixelSdf, RigidPointOptimizer, SdfVoxel) calls
 * only these functions; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions (identical to the reference's):
 *   depth  : float32, row-major H x W, continuous, metres, 0 = invalid
 *            (cv::Mat CV_32FC1, img_loader/ImageLoader.h:159-175)
 *   K      : row-major 3x3 float  fx 0 cx; 0 fy cy; 0 0 1
 *   R, t   : camera->world, p_w = R p_c + t, R row-major (MapGradPixelSdf.cpp:62-64,103)
 *   pose7  : tx ty tz qx qy qz qw  (Sophus SE3 state; TUM file order, main_scan_3d.cpp:274-280)
 *   keys   : int32 x,y,z voxel indices (Eigen::Vector3i, MapGradPixelSdf.h:65-68)
 *   payload: float dist, gx, gy, gz, weight (struct SdfVoxel, sdf_voxel/SdfVoxel.h:45-57)
 *
 * Plain C types only, caller-owned buffers, no exceptions across the ABI.
 * One context = one GPU; a context is not thread-safe (the reference has a
 * single caller thread).  All functions return GSDF_OK (0) or a GSDF_ERR_* code;
 * gsdf_last_error() gives the message.  *_dev variants take DEVICE pointers and
 * only enqueue work on the context's HIP stream (no host synchronisation).
 */
#ifndef GSDF_H_
#define GSDF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSDF_OK               0
#define GSDF_ERR_TABLE_FULL   1   /* the block table (or the deferred list) ran out of room (new failure mode) */
#define GSDF_ERR_KEY_RANGE    2   /* voxel index outside the packable +-2^20 range */
#define GSDF_ERR_INVALID      3   /* bad argument / call order */
#define GSDF_ERR_HIP          4   /* HIP runtime error (message in gsdf_last_error) */
#define GSDF_ERR_NO_DEVICE    5   /* no gfx950 device visible: the engine has NO CPU fallback */

typedef struct gsdf_ctx gsdf_ctx;

/* counters of the fusion / tracking launches (for algorithmic-bytes accounting: callers take differences) */
typedef struct gsdf_stats {
    int64_t n_upd;        /* (pixel,k) samples with w>0, summed over every update() since create/reset   */
    int64_t n_valid;      /* pixels that passed the z-range and both normal gates, summed likewise       */
    int64_t n_hit;        /* sum over executed tracker passes of pixels with w0>0       */
    int32_t track_passes; /* tracker reduction passes executed in the last optimize()   */
    int32_t converged;    /* result of the last optimize()                              */
    int64_t frames;       /* Sdf::counter_ (Sdf.h:65)                                   */
    int64_t n_deferred;   /* voxel contributions that took the deferred (float-atomic) route of the fusion flush
                             since create/reset: near tiles, LDS overflow, timed-out waits (0 in steady state)  */
    int64_t fuse_timeouts;/* fusion tiles whose bounded wait for a neighbouring tile expired (they deferred)     */
} gsdf_stats;

const char* gsdf_last_error(void);
const char* gsdf_version(void);

/* MapGradPixelSdf(voxel_size, T) + table allocation -- MapGradPixelSdf.h:99-103, Sdf.h:103-107.
 * capacity_log2 in [10, 30]: 2^capacity_log2 32-byte voxel records, organised as 2^(capacity_log2-6) blocks of
 * 4x4x4 voxels (BASELINE configs: 22 and 25).  A surface map fills its blocks to ~70 %, so it holds about
 * 0.6 * 2^capacity_log2 voxels before GSDF_ERR_TABLE_FULL.
 * trunc_dist (T) must be below 2 m: the fusion kernel accumulates a tile's w * sdf sums in 51-bit fixed point (2^-40).
 * device: HIP device ordinal.  Fails with GSDF_ERR_NO_DEVICE when no GPU is present. */
int gsdf_create(gsdf_ctx** out, float voxel_size, float trunc_dist, int capacity_log2, int device);
/* delete tSDF -- main_scan_3d.cpp:314 */
void gsdf_destroy(gsdf_ctx* c);
/* drop all voxels, frame counter := 0 (new: lets one context be reused by bench/tests) */
int gsdf_reset(gsdf_ctx* c);
/* The reference's tsdf_ grows without bound (a node hash map, MapGradPixelSdf.h:65-68); this table has a capacity.  gsdf_grow
 * moves every block into a table of 2^new_capacity_log2 records (synchronous; the map -- voxels, vis_ bit-vectors, frame counter
 * -- is unchanged, only the room differs; on failure the old table stays; a sticky GSDF_ERR_TABLE_FULL from an earlier fusion
 * does not stop it, but stays reported: that fusion's samples were dropped).  gsdf_set_auto_grow lets the frame entries
 * (gsdf_update_dev / gsdf_update / gsdf_track_and_fuse_dev) do that by themselves: the table is doubled when ~45 % of its block
 * entries are in use, up to max_capacity_log2; 0 switches it off, which is the default -- then GSDF_ERR_TABLE_FULL reports a map
 * that outgrew its table, as before.  The load is counted on the device behind every frame and read without waiting; the
 * library knows how old the count it sees is and how fast the map grew lately, and an entry waits by itself (for the device to
 * come within 8 frames, or for an exact count) when count + lag x growth comes near the limit -- callers need not
 * synchronise.  What auto-grow cannot absorb: ONE frame that needs more new blocks than the table has left (more than half
 * its entries): that frame still ends in GSDF_ERR_TABLE_FULL.  Size the initial table for at least two frames' blocks.
 * gsdf_capacity reports the present capacity_log2. */
int gsdf_grow(gsdf_ctx* c, int new_capacity_log2);
int gsdf_set_auto_grow(gsdf_ctx* c, int max_capacity_log2);
int gsdf_capacity(gsdf_ctx* c, int* capacity_log2);

/* Sdf::set_zmin / Sdf::set_zmax -- Sdf.h:123-129 (defaults 0.5 / 3.5) */
int gsdf_set_zrange(gsdf_ctx* c, float zmin, float zmax);

/* new cv::NormalEstimator<float>(W, H, K, Size(win,win)) -> cache() -- normals/NormalEstimator.h:81-165,
 * main_scan_3d.cpp:183.  Also fixes the frame size all later calls must use. */
int gsdf_normals_init(gsdf_ctx* c, int W, int H, const float K[9], int win);
/* the 11 cached planes (x0,y0,x0/n2,y0/n2,1/n2,Q11,Q12,Q13,Q22,Q23,Q33) copied to the host */
int gsdf_normals_cache(gsdf_ctx* c, float* planes11_host);
/* NormalEstimator::compute(depth, nx, ny, nz) -- NormalEstimator.h:179-204 (host in, host out) */
int gsdf_normals_compute(gsdf_ctx* c, const float* depth_host, float* nx, float* ny, float* nz);

/* Sdf::update(color, depth, K, pose, NEst) -- Sdf.h:117, MapGradPixelSdf.cpp:43-122.
 * color is ignored by the reference and is not part of this ABI.  Synchronous; returns
 * GSDF_ERR_TABLE_FULL / GSDF_ERR_KEY_RANGE if the launch reported one. */
int gsdf_update(gsdf_ctx* c, const float* depth_host, const float R[9], const float t[3]);
/* same with depth already resident in HBM; enqueue only.
 * Runs of this call are pipelined: the fusion of a frame is launched when the NEXT frame arrives (that launch's last workgroups
 * compute the next frame's normals in its otherwise idle tail) or when any other entry point of this header that touches the
 * map, the stream or the frame is called on the context -- every one of them launches a waiting fusion first, so results never
 * depend on it.  The staging entries (gsdf_dev_upload, gsdf_dev_upload_async) do so when their destination overlaps the waiting
 * fusion's depth image, so one staging buffer may be reused frame after frame (upload -> update_dev -> upload -> ...) exactly as
 * before; gsdf_dev_alloc / gsdf_host_alloc / gsdf_host_free / gsdf_mark_wait / gsdf_mark_reached / gsdf_upload_wait do not
 * touch it.  The caller's contract is the old one: depth_dev must stay valid and unchanged until the fusion has RUN, i.e. until
 * a mark recorded after this call (gsdf_mark) has been reached, or gsdf_sync has returned.
 * Error reporting lags with the launch: a launch error of frame i's fusion is returned by the call that launches it -- the
 * gsdf_update_dev of frame i + 1 (which is then NOT queued: call it again after handling the error) or whichever entry point
 * flushed it; device-side failures (GSDF_ERR_TABLE_FULL, GSDF_ERR_KEY_RANGE) are sticky and reported by gsdf_sync as before. */
int gsdf_update_dev(gsdf_ctx* c, const float* depth_dev, const float R[9], const float t[3]);

/* RigidOptimizer::optimize(depth, K) -- RigidOptimizer.h:106, RigidPointOptimizer.cpp:40-99.
 * pose7 in/out replaces RigidOptimizer::pose()/set_pose (RigidOptimizer.h:99-103);
 * num_iterations/conv_threshold/damping replace the setters (RigidOptimizer.h:85-97).
 * *converged receives the bool the reference returns; *passes the reduction passes executed. */
int gsdf_track(gsdf_ctx* c, const float* depth_host, const float K[9], float pose7[7],
               int num_iterations, float conv_threshold, float damping,
               int* converged, int* passes);

/* RigidPointOptimizer::optimize_sampled(depth, K, sampling) -- RigidPointOptimizer.h:65, the public member optimize() forwards to
 * with sampling 1 (.h:69-72).  sampling >= 1 is the pixel stride of RigidPointOptimizer.cpp:62 (`y += sampling`, `x += sampling`:
 * the pixels (i * sampling, j * sampling)); a stride beyond the image leaves pixel (0, 0), as there; sampling < 1 (the
 * reference's size_t 0 never leaves its loop) is GSDF_ERR_INVALID.  gsdf_track == gsdf_track_sampled(..., 1, ...).  The frame
 * loop entry below has no such argument because the reference's loop calls optimize() (main_scan_3d.cpp:258). */
int gsdf_track_sampled(gsdf_ctx* c, const float* depth_host, const float K[9], float pose7[7],
                       int num_iterations, float conv_threshold, float damping, int sampling,
                       int* converged, int* passes);

/* The Scan3D loop body without host round trips -- main_scan_3d.cpp:255-266:
 *   conv = pOpt->optimize(depth, K);  if (conv) tSDF->update(color, depth, K, pOpt->pose(), NEst);
 * The pose persists on the device between frames (RigidOptimizer::pose_, RigidOptimizer.h:64);
 * set it with gsdf_set_pose.  Enqueue only; per-frame results are appended to a device log that
 * gsdf_read_frame_log returns (pose7 + converged + passes per frame). */
int gsdf_set_pose(gsdf_ctx* c, const float pose7[7]);
int gsdf_get_pose(gsdf_ctx* c, float pose7[7]);          /* synchronises */
int gsdf_track_and_fuse_dev(gsdf_ctx* c, const float* depth_dev, const float K[9],
                            int num_iterations, float conv_threshold, float damping);
/* Optional, time only: names the frame the NEXT gsdf_track_and_fuse_dev will be called with, before the call for the CURRENT
 * frame.  The current frame's fusion launch then computes NormalEstimator::compute of that next frame in its tail (its last
 * workgroups: the fusion leaves a third of the chip's workgroup slots idle there) -- when that fusion runs, i.e. the current
 * frame converged; it leaves a token, and the next frame's tracker launches, which carry the normals tiles as ever, skip them
 * when they find it.  The same launch also performs the closing head of the current frame's optimize() (reduce, solve, stop
 * test of the first batch's last pass) instead of a tracker launch of its own.
 * The normals depend on the depth image alone (MapGradPixelSdf.cpp:60), so results do not: same poses, same map
 * (tests/test_gpu_parity.py::test_next_depth_hint_is_invisible_except_in_time).  Contract: next_depth_dev already holds the
 * next frame when the CURRENT frame's gsdf_track_and_fuse_dev is called, and stays unchanged until the next frame's call; a copy
 * into it through gsdf_dev_upload* in between withdraws the hint, as does any other frame entry; a hint that does not match the
 * next call's depth_dev is ignored.  One hint per frame; NULL withdraws it.  No device work, never an error for a mismatch. */
int gsdf_hint_next_depth_dev(gsdf_ctx* c, const float* next_depth_dev);
/* both in one call: gsdf_hint_next_depth_dev(c, next_depth_dev) + gsdf_track_and_fuse_dev(c, depth_dev, ...) -- for hosts whose
 * calls are not free (a ctypes call is ~1.5 us, 1 % of a frame, and the host is in the loop while a frame needs further batches
 * of passes).  next_depth_dev may be NULL. */
int gsdf_track_and_fuse_ahead_dev(gsdf_ctx* c, const float* depth_dev, const float* next_depth_dev, const float K[9],
                                  int num_iterations, float conv_threshold, float damping);
/* log rows = float[10]: pose7, converged, passes, n_hit (of the last pass) */
int gsdf_read_frame_log(gsdf_ctx* c, float* rows10, int64_t max_rows, int64_t* n_rows);

/* wait for all enqueued work; returns the sticky launch status (table full / key range) */
int gsdf_sync(gsdf_ctx* c);
int gsdf_get_stats(gsdf_ctx* c, gsdf_stats* out);        /* synchronises */

/* tsdf_.size() -- number of occupied voxels */
int gsdf_count(gsdf_ctx* c, int64_t* n);
/* get_tsdf() -- MapGradPixelSdf.h:133-138: all (key, SdfVoxel) pairs.  sorted!=0 orders rows by
 * (z,y,x) so exports are diffable.  raw_sums!=0 returns the additive accumulators
 * (sum w*d, sum w*Rn, sum w) instead of (dist, grad, weight) -- the merge wire format. */
int gsdf_export(gsdf_ctx* c, int32_t* keys, float* payload, int64_t max_n, int64_t* n,
                int sorted, int raw_sums);
/* vis_ -- MapGradPixelSdf.h:70, MapGradPixelSdf.cpp:113-115: per voxel, bit f set <=> the voxel was updated by
 * integrated frame f (f = Sdf::counter_ at the time).  The reference always maintains it (PhotoBA reads it
 * through get_vis(), MapGradPixelSdf.h:140-142); here it is opt-in because it costs one more atomic per
 * touched voxel: call gsdf_enable_vis once (before fusing) with the number of frames to keep.
 * gsdf_export_vis returns ceil(max_frames/32) uint32 words per voxel, rows in the (z,y,x) order of
 * gsdf_export(sorted=1); keys may be NULL. */
int gsdf_enable_vis(gsdf_ctx* c, int max_frames);
int gsdf_export_vis(gsdf_ctx* c, int32_t* keys, uint32_t* words, int words_per_voxel, int64_t max_n, int64_t* n);

/* PhotoBA -- class PhotometricOptimizer (ps_optimizer/PhotometricOptimizer.h:68-186, .cpp), coarse photometric
 * bundle adjustment of keyframe poses and voxel distances on the fused map.  Needs gsdf_enable_vis.
 * images: float32 BGR in [0,1], n x H x W x 3 (setImages, cv::Mat CV_32FC3); poses16: n row-major 4x4
 * camera->world (setPoses); frame_idx: integrated-frame index of each keyframe (setKeyframes).
 * LIMIT: 1 <= n <= 64 keyframes per set (GSDF_ERR_INVALID beyond).  The reference's frame_idx_ / poses_ / images_ are
 * unbounded std::vectors (PhotometricOptimizer.h:68-186); here the pose sweep keeps one visibility bit per keyframe in a 64-bit
 * word per voxel and its per-wave LDS accumulators are sized by n.  BASELINE configs[4] (50 keyframes) fits; a larger keyframe
 * set has to be optimised in windows of <= 64 keyframes (the energy is a sum over voxels of terms that couple only the keyframes
 * that see the voxel, so windows of neighbouring keyframes are the usual bundle-adjustment practice -- but it is NOT what the
 * reference computes for n > 64, and no such windowing is done inside the library).
 * gsdf_ba_energy = getEnergy (:273-321), gsdf_ba_solve_pose = solvePose (:499-590), gsdf_ba_solve_dist =
 * solveDist (:326-388), gsdf_ba_optimize = optimize (:611-662): energies receives E0 and E after every pose
 * and distance step (<= 2*max_it+1 values), *converged = relative change < 5e-4. */
int gsdf_ba_setup(gsdf_ctx* c, int n, const float* images_bgr_host, const float* poses16_host, const int* frame_idx,
                  float reg_weight);
/* OptSettings::loss and OptSettings::lambda (PhotometricOptimizer.h:54-57; LossFunction, loss.h:39-46: 0 L2, 1 CAUCHY -- the
 * default --, 2 HUBER, 3 TUKEY, 4 TRUNC_L2).  As in the reference only TRUNC_L2 changes the computation: solveDist (:364) and
 * solvePose (:542) then leave out a keyframe whose intensity residual exceeds lambda^2 in any channel.  Default: CAUCHY, 0.5. */
int gsdf_ba_set_loss(gsdf_ctx* c, int loss, float lambda);
int gsdf_ba_energy(gsdf_ctx* c, float* E);
int gsdf_ba_solve_pose(gsdf_ctx* c, float damping);
int gsdf_ba_solve_dist(gsdf_ctx* c, float damping);
int gsdf_ba_optimize(gsdf_ctx* c, int max_it, float* energies, int* n_energies, int* converged);
int gsdf_ba_get_poses(gsdf_ctx* c, float* poses16_host);
/* what the last gsdf_ba_energy / gsdf_ba_solve_dist call (or the last energy sweep of gsdf_ba_optimize) counted: voxels that took
 * part (energy: |dist| <= voxel size, :285, seen by at least one keyframe; distance sweep: every voxel with an observation) and
 * observations (voxel x keyframe pairs that project into the image, :238-260) -- the units of the sweeps' algorithmic bytes
 * (measurement only; the reference has no counterpart) */
int gsdf_ba_counters(gsdf_ctx* c, int64_t* voxels, int64_t* observations);

/* additive merge of raw sums into this table (frame-sharded fusion) */
int gsdf_merge_raw(gsdf_ctx* c, const int32_t* keys, const float* payload_raw, int64_t n);

/* dst += src for two contexts on the SAME device ("G logical shards on 1 GPU"; the GT-pose branch,
 * main_scan_3d.cpp:250-254): a GPU can fuse two frame shards at once, each context on its own stream -- a fusion launch leaves
 * a third of the chip's workgroup slots idle in its tail, which the other context's launch fills -- and adds them up locally
 * before the exchange between devices.  Voxel sums are added, Sdf::counter_ becomes the frames of both, vis_ bit-vectors of the
 * source (if enabled, alike on both) are OR-ed in shifted by dst's frame count (dst's frames come first).  Synchronous; src is
 * left unchanged.  Room: the union holds at most blocks(dst) + blocks(src); if that passes 45 % of dst's block entries dst is
 * doubled first when gsdf_set_auto_grow allows it, and if it would pass 90 % of them (no growth allowed, or not enough) the call
 * returns GSDF_ERR_TABLE_FULL with dst UNCHANGED.  Should the kernel itself still report a full table, dst holds part of src,
 * Sdf::counter_ is not advanced and dst is to be reset (gsdf_reset). */
int gsdf_merge_from(gsdf_ctx* dst, gsdf_ctx* src);
/* n (1..4) contexts for n frame shards on ONE device, like gsdf_create each -- but their streams are guaranteed to sit in n
 * different hardware queues (created at the device's highest stream priority: the runtime pools its queues per priority, and
 * nothing else uses that pool), so that their launches overlap.  Streams of the default priority share queues as soon as the
 * process holds more streams than the runtime has queues (4), and then two contexts fuse no faster than one.
 * Side effects, stated: the shard streams run at the HIGHEST priority, so their launches are scheduled ahead of default-priority
 * work on the same GPU (other gsdf contexts, PyTorch) -- meant for a process whose GPU work at that moment IS the shard fusion;
 * and the guarantee holds only while nothing else in the process creates highest-priority streams.  A device without a stream
 * priority range makes the call fail (GSDF_ERR_HIP) instead of handing out default-priority streams. */
int gsdf_create_shards(gsdf_ctx** out, int n, float voxel_size, float trunc_dist, int capacity_log2, int device);

/* device-buffer variants for the multi-GPU exchange (RCCL works on device memory): unsorted
 * compaction of (key, raw sums) into caller-provided DEVICE buffers / additive merge from them.
 * These are the pack / unpack steps either side of the all-gather. */
int gsdf_export_raw_dev(gsdf_ctx* c, int32_t* keys_dev, float* payload_raw_dev, int64_t max_n, int64_t* n);
int gsdf_merge_raw_dev(gsdf_ctx* c, const int32_t* keys_dev, const float* payload_raw_dev, int64_t n);

/* Dense variant of the exchange: the all-reduce of per-voxel (weight, weighted distance, weighted gradient) named in
 * BASELINE.json's north_star.  The map is a hash of 4x4x4 voxel blocks; block ids are opaque 64-bit words, identical
 * across contexts.  (1) gsdf_block_keys_dev lists this map's block ids (*n = count, also when > max_n); the ranks
 * all-gather and unique them; (2) gsdf_pack_blocks_dev writes 64 x 5 raw sums (w, s, gx, gy, gz; zeros for voxels or
 * blocks this map lacks) per listed block into a DEVICE buffer, which RCCL all-reduces (sum); (3)
 * gsdf_unpack_blocks_dev stores the reduced sums (inserting missing blocks).  All pointers are device pointers. */
int gsdf_block_keys_dev(gsdf_ctx* c, uint64_t* block_keys_dev, int64_t max_n, int64_t* n);
int gsdf_pack_blocks_dev(gsdf_ctx* c, const uint64_t* block_keys_dev, int64_t n, float* dense_dev);
int gsdf_unpack_blocks_dev(gsdf_ctx* c, const uint64_t* block_keys_dev, int64_t n, const float* dense_dev);

/* The exchange step as ONE call, for C++ hosts (BASELINE.json north_star: "frames shard naturally across the
 * 8 GPUs of one node with a RCCL-over-xGMI all-reduce of per-voxel (weight, weighted-distance, weighted-gradient) before
 * mesh extraction").  Collective: every rank of the communicator calls it with the map it fused from its own frames (the
 * GT-pose branch, main_scan_3d.cpp:250-254); on return every rank's map is the sum of all maps (and every rank's table has the
 * capacity of the largest one: ranks whose tables are smaller grow first, gsdf_grow).  Steps: all-gather of the block-key arrays, sorted union on the device, gsdf pack, ONE ncclAllReduce (sum, float32, 1280 B per block of the union) on the context's own
 * stream, gsdf unpack.  nccl_comm: an ncclComm_t of the RCCL already in the process (RCCL is resolved at run time, it is
 * not a link dependency of libgsdf.so).  n_blocks / bytes (nullable): size of the union / of the all-reduced buffers.
 * Also exchanged: Sdf::counter_ (Sdf.h:65) -- afterwards the frames integrated by ALL ranks -- and, when gsdf_enable_vis was
 * called (on every rank, with the frame count of the WHOLE job), the vis_ bit-vectors (MapGradPixelSdf.cpp:113-115): the
 * frame shards are contiguous in rank order, so frame f of rank r becomes integrated frame (frames of the ranks < r) + f and
 * the ranks' shifted vectors are OR-ed (a second all-reduce, unsigned words): the merged map is what PhotoBA needs (C4 -> C5).
 * ONE-SHOT: after the call every map IS the sum, so a second exchange over more than one rank is refused (GSDF_ERR_INVALID)
 * until gsdf_reset.  Failures are collective: a rank that cannot prepare its part reports that through the first all-gather
 * and every rank returns the error, none is left waiting inside a collective -- with ONE exception: the two header buffers
 * of a few hundred bytes are needed to talk at all, and a rank whose hipMalloc fails for THEM returns before the first
 * all-gather; its peers then wait in it (treat that as fatal for the job, as an out-of-memory device is).
 * Memory: the key arrays are gathered WHOLE (8 B per block entry of the table, occupied or not, from every rank) and sorted on
 * the device: three buffers of nranks x 2^(capacity_log2 - 6) x 8 B plus the sort's scratch -- 3 x 4 MB at 8 ranks and 2^22
 * records, 3 x 268 MB at 2^28 (auto-grow's CLI limit), 3 x 1 GB at 2^30.  Growing to the largest rank's capacity also happens
 * inside this call (and inside a timed exchange). */
int gsdf_merge_allreduce(gsdf_ctx* c, void* nccl_comm, int64_t* n_blocks, int64_t* bytes);
/* Optional set-up of the exchange outside a timed region (like the communicator): the scratch buffers for `nranks` ranks and one
 * run of the union's sort kernels, whose code is loaded on first use (~10 ms in the first exchange of a process otherwise).
 * Not collective. */
int gsdf_merge_prepare(gsdf_ctx* c, int nranks);
/* communicator plumbing for hosts that do not link RCCL themselves (the Scan3D CLI): ncclGetUniqueId / ncclCommInitRank
 * (on `device`) / ncclCommDestroy.  Create the communicator once, outside any timed region. */
int gsdf_rccl_unique_id(char id128[128]);
int gsdf_rccl_comm_init(void** nccl_comm, int nranks, const char id128[128], int rank, int device);
int gsdf_rccl_comm_count(void* nccl_comm, int* nranks);       /* ncclCommCount: ranks of the communicator */
int gsdf_rccl_comm_destroy(void* nccl_comm);
/* The same exchange over a caller-provided transport: two collectives on HOST buffers (libgsdf stages the device data).
 * Used where RCCL cannot run (two ranks on one GPU in the tests) or where the host has its own communication layer.
 * Both callbacks return 0 on success. */
typedef struct gsdf_collective {
    /* recv[r * bytes .. (r + 1) * bytes) := rank r's `send` (bytes per rank are equal on all ranks) */
    int (*allgather)(void* user, const void* send, void* recv, int64_t bytes);
    /* buf[i] := sum over the ranks of buf[i], i < n */
    int (*allreduce_sum_f32)(void* user, float* buf, int64_t n);
    void* user;
    int nranks;
} gsdf_collective;
int gsdf_merge_allreduce_with(gsdf_ctx* c, const gsdf_collective* ops, int64_t* n_blocks, int64_t* bytes);

/* Sdf::weights(point) and Sdf::tsdf(point, &grad) at n points -- MapGradPixelSdf.h:109-125.
 * w[i]==0 marks a missing voxel (dist/grad are then 0; the reference's .at() would throw). */
int gsdf_query(gsdf_ctx* c, const float* pts_host, int64_t n, float* dist, float* grad, float* w);

/* MapGradPixelSdf::getSdf(idx) = tsdf_.at(idx) -- MapGradPixelSdf.h:127-129, for n voxel indices at once: the STORED voxel,
 * payload = dist, gx, gy, gz (the raw weighted gradient sum, not the normalised one gsdf_query returns), weight.
 * found[i] == 0 marks a missing voxel (payload zeros; the reference's .at() would throw). */
int gsdf_get_voxels(gsdf_ctx* c, const int32_t* keys_host, int64_t n, float* payload, int32_t* found);

/* Voxel-hash raycaster: depth (camera z, 0 = no hit) and camera-frame normals (3 planes, nullable) of the
 * fused map seen from pose (R, t) through K.  Named in BASELINE.json's north_star but ABSENT from the
 * reference: it is defined on top of Sdf::weights / Sdf::tsdf (MapGradPixelSdf.h:109-125) and the
 * tracker's back-projection (RigidPointOptimizer.cpp:46-47,67-70); DESIGN.md states the definition. */
int gsdf_raycast(gsdf_ctx* c, const float K[9], const float R[9], const float t[3], int W, int H, float zmin, float zmax,
                 float* depth_out, float* normals_out);

/* the same with DEVICE output buffers (depth W*H floats, normals 3*W*H floats or NULL); enqueue only */
int gsdf_raycast_dev(gsdf_ctx* c, const float K[9], const float R[9], const float t[3], int W, int H, float zmin, float zmax,
                     float* depth_dev, float* normals_dev);
/* what the raycasts since the last reset did: samples their definition evaluated (one block-key probe each) and voxel records
 * read -- the algorithmic bytes of the raycaster's roofline entry are 8 B per sample + 32 B per record + 16 B per pixel.
 * Synchronises; reset != 0 clears the counters. */
int gsdf_raycast_counters(gsdf_ctx* c, int64_t* samples, int64_t* records, int reset);

/* Iso-surface of the map on the device -- LayeredMarchingCubesNoColor::computeIsoSurface + computeLutIndex + interpolate +
 * computeTriangles (mesh/LayeredMarchingCubesNoColor.cpp:354-712), called by MapGradPixelSdf::extract_mesh
 * (MapGradPixelSdf.cpp:124-175).  tri_table: NULL = the reference's triTable (:96-352, the classic marching-cubes
 * cases, constant data in include/gsdf_mc_tables.h), which gives the reference's mesh triangle for triangle; or a
 * caller's 256 x 16 table (edge ids, 3 per triangle, -1 ends a row; corner / edge numbering of :599-606, :410-549).
 * triangles_out: 9 floats per triangle, in the reference's z-y-x sweep order, no vertex de-duplication, degenerate
 * triangles dropped.  *n_tris = triangles found; call with max_tris = 0 to size the buffer. */
int gsdf_extract_mesh(gsdf_ctx* c, float iso, const int8_t tri_table[256 * 16], float* triangles_out, int64_t max_tris,
                      int64_t* n_tris);

/* device-memory plumbing so callers can stage frames in HBM without another runtime */
int gsdf_dev_alloc(gsdf_ctx* c, void** dev_ptr, int64_t bytes);
int gsdf_dev_free(gsdf_ctx* c, void* dev_ptr);
int gsdf_dev_upload(gsdf_ctx* c, void* dev_dst, const void* host_src, int64_t bytes);
int gsdf_dev_download(gsdf_ctx* c, void* host_dst, const void* dev_src, int64_t bytes);
/* Asynchronous frame staging for a host loop that keeps the GPU fed (the Scan3D CLI): page-locked host
 * buffers, an upload that is only ENQUEUED on the context's stream (host_src must stay untouched until a later mark is
 * reached), and marks: gsdf_mark records a point in the stream, gsdf_mark_wait blocks until the stream has passed it,
 * gsdf_mark_reached polls.  Marks complete in the order they were recorded.  A reached mark orders STREAM PROGRESS only (the
 * kernels and copies queued before it have run: a staging slot may be reused); its event carries no system-scope fence, so it
 * does NOT make device writes visible to the host -- read results through gsdf_sync or the download / export entries. */
int gsdf_host_alloc(gsdf_ctx* c, void** host_ptr, int64_t bytes);
int gsdf_host_free(gsdf_ctx* c, void* host_ptr);
int gsdf_dev_upload_async(gsdf_ctx* c, void* dev_dst, const void* host_src, int64_t bytes);
int gsdf_mark(gsdf_ctx* c, int64_t* mark);
int gsdf_mark_wait(gsdf_ctx* c, int64_t mark);
int gsdf_mark_reached(gsdf_ctx* c, int64_t mark, int* reached);
/* Staging AHEAD of the stream: the copy runs on the context's copy stream -- a DMA engine beside the kernels of the frames
 * before it, where gsdf_dev_upload_async queues behind them and makes the stream change engines twice per frame --
 * and gsdf_upload_wait blocks the HOST until that copy has arrived; after it returned, dev_dst may be handed to any _dev entry.
 * The caller keeps dev_dst out of the stream's reach meanwhile (a buffer whose last reader passed a mark) and host_src
 * untouched until the wait returned.  Uploads complete in the order they were started. */
int gsdf_dev_upload_ahead(gsdf_ctx* c, void* dev_dst, const void* host_src, int64_t bytes, int64_t* upload);
int gsdf_upload_wait(gsdf_ctx* c, int64_t upload);
/* HIP-event timing on the context's stream: t0/t1 bracket whatever is enqueued between them */
int gsdf_timer_start(gsdf_ctx* c);
int gsdf_timer_stop_ms(gsdf_ctx* c, float* ms);          /* synchronises */
/* accumulated per-kernel HIP-event time (ms) and launch counts since the last reset:
 * index 0 normals, 1 fusion, 2 tracking pass.  Enabled by gsdf_profile(c, 1). */
int gsdf_profile(gsdf_ctx* c, int enable);
int gsdf_profile_read(gsdf_ctx* c, double ms[3], int64_t launches[3]);
/* the first n <= 5 slots: 0 normals, 1 fusion, 2 tracking pass launches, 3 raycast, 4 reserved */
int gsdf_profile_read_n(gsdf_ctx* c, int n, double* ms, int64_t* launches);
/* every launch of one slot by itself, in launch order (ms each; at most 2^20 kept): *n = launches recorded since the last
 * gsdf_profile(c, 1), the first min(max_n, *n) durations in ms[].  Lets a caller tell the tracker launches that ran a pass from
 * the head-only ones and those that found optimize() already ended (bench.py: roofline.tracker per executed pass). */
int gsdf_profile_read_launches(gsdf_ctx* c, int slot, float* ms, int64_t max_n, int64_t* n);

#ifdef __cplusplus
}
#endif
#endif /* GSDF_H_ */
