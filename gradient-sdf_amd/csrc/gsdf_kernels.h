/*
 * gsdf_kernels.h -- launch interface between the C-ABI host code (gsdf_capi.hip)
 * and the gfx950 kernels (gsdf_kernels.hip).
 */
#ifndef GSDF_KERNELS_H_
#define GSDF_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gsdf_table.h"

#define GSDF_STATUS_TABLE_FULL 1
#define GSDF_STATUS_KEY_RANGE  2
#define GSDF_STATUS_TRACK_ABORT 4   /* k_track_all: the workgroups of the one-launch optimize() were not co-resident (GPU shared with another process) */

#ifndef GSDF_TRACK_BLOCK
#define GSDF_TRACK_BLOCK   512
#endif
#ifndef GSDF_TRACK_MAXBLK
#define GSDF_TRACK_MAXBLK  256   /* workgroups of a tracker pass at >= 2^18 pixels: every workgroup re-reduces all
                                    partial rows at the head of the next pass, so fewer, larger workgroups win
                                    (measured: 512 x 256 thr 16.3 us per pass, 256 x 512 thr 13.8, 1024 x 256 thr 21.4) */
#endif
#define GSDF_TRACK_NSUM    29     /* E, g[6], H upper triangle[21], count */
#ifndef GSDF_TRACK_GROUPS
#define GSDF_TRACK_GROUPS  16     /* partial-sum groups of a tracker pass (workgroup b -> group b % 16); 8..32 measured within 0.4 us, 64 slower */
#endif
#define GSDF_TRACK_ROWSET  (GSDF_TRACK_GROUPS * 32)   /* doubles per buffer; three buffers rotate */

/* Device-resident engine state: the tracker's pose (RigidOptimizer::pose_, RigidOptimizer.h:64),
 * per-optimize flags, sticky launch status and the counters the stats API reports. */
/* tracker state handed from the head of one pass launch to the next (double-buffered by pass parity) */
struct gsdf_trk_buf {
    float pose7[7];
    int done, converged, passes;
};

struct gsdf_dev_state {
    float pose7[7];               /* tx ty tz qx qy qz qw */
    float R[9];                   /* rotationMatrix() of pose7, kept in step by whoever writes pose7 */
    int done;                     /* the running / last optimize() has ended (published with `converged` by the pass heads) */
    int converged;
    int passes;
    int max_passes;
    unsigned int fuse_timeouts;   /* fusion tiles whose bounded wait for a neighbour expired (they deferred instead) */
    unsigned int far_tiles;       /* tiles of the running fusion whose voxels do not fit the small LDS table in one band (reset by its last workgroup) */
    unsigned int last_deferred;   /* length of the deferred list of the last fusion launch */
    float last_hits;
    float conv_sq;
    float damping;
    int status;                   /* GSDF_STATUS_* bits, sticky */
    unsigned long long n_upd, n_valid, n_hit, n_occupied;
    unsigned long long n_deferred; /* contributions added by k_fuse_resolve so far */
    long long frames;             /* Sdf::counter_ */
    long long log_rows;
    long long frame_cur;          /* counter_ snapshot for the running update (k_normals -> k_fuse) */
    gsdf_trk_buf trk[2];
    unsigned long long dbg[24];   /* experiment counters (gsdf_debug_flags & 128), see tools/go_count.py */
    /* The closing head of optimize() when a fusion launch performs it (k_fuse<.., HEAD>): three 16-byte chunks that workgroup 0
     * writes and every other workgroup polls, each carrying the launch's tag, so a chunk with the right tag holds the right
     * values and no ordering between data and flag is needed: {tag, done | converged << 1, tx, ty} {tag, tz, qx, qy}
     * {tag, qz, qw, passes} */
    unsigned int fh[12] __attribute__((aligned(16)));
    unsigned int nrm_token;       /* gsdf_hint_next_depth_dev: token of the frame whose normals a fusion launch computed in its tail */
};

struct gsdf_frame_geom {
    int W, H;
    float fx, fy, cx, cy;
    float vs, inv_vs, T, inv_T, zmin, zmax;
    int factor;
};

/* cached planes of the normal estimator, each W*H floats (NormalEstimator.h:60-64) */
struct gsdf_ncache {
    const float *x0, *y0, *x0n, *y0n, *ninv, *q11, *q12, *q13, *q22, *q23, *q33;
};

struct gsdf_pose_arg { float R[9]; float t[3]; };

/* a contribution to a voxel that another tile owns in this fusion launch (added by k_fuse_resolve) */
struct __attribute__((aligned(16))) gsdf_deferred {
    gsdf_payload* p;
    float w, s, gx, gy, gz;
    uint32_t pad;
};

void gsdf_launch_table_clear(hipStream_t s, gsdf_table tab, size_t n_slots);
void gsdf_launch_occ_rebuild(hipStream_t s, gsdf_table tab);      /* block / cell filters of the raycaster from the key array */
/* scratch: gsdf_normals_cache_scratch_bytes(W, H) of device memory, free again once the stream has run the two kernels */
size_t gsdf_normals_cache_scratch_bytes(int W, int H);
void gsdf_launch_normals_cache(hipStream_t s, int W, int H, const float* K, int win, float* planes11, double* scratch);
void gsdf_launch_normals(hipStream_t s, const gsdf_frame_geom& g, int win, const gsdf_ncache& nc,
                         const float* depth, float* nx, float* ny, float* nz,
                         unsigned int* deferred_count /* nullable: cleared for the k_fuse that follows */,
                         gsdf_dev_state* st_rw /* nullable: snapshot of the frame counter for k_fuse */,
                         uint32_t* tile_stats /* nullable: [gsdf_fuse_grid_blocks][4], the frame's tile statistics for its k_fuse */);
/* The head of tracker launch k performed by a fusion launch instead (the frame's first gated fusion stands in for the last launch
 * of the first batch): k = 0 none */
struct gsdf_fuse_head {
    int k;                        /* the tracker launch whose head this is: finishes pass k - 1 */
    unsigned int rot_prev;        /* which of the three sum buffers pass k - 1 accumulated into */
    float conv_sq, damping;
    int max_passes;
    unsigned int serial;
    unsigned int* progress;       /* as gsdf_track_params::progress */
    const double* rows;           /* the partial-sum buffers (3 x GSDF_TRACK_ROWSET) */
    int debug;
};
/* use_dev_pose: take R,t from st->R / st->pose7 and skip the launch unless st->converged */
void gsdf_launch_fuse(hipStream_t s, const gsdf_frame_geom& g, const gsdf_ncache& nc, const float* depth,
                      const float* nx, const float* ny, const float* nz, const gsdf_pose_arg& pose,
                      int use_dev_pose, gsdf_table tab, gsdf_dev_state* st,
                      unsigned long long* blk_counters /* [gsdf_fuse_grid_blocks][4] */,
                      gsdf_deferred* deferred, unsigned int* deferred_count, unsigned int deferred_cap,
                      unsigned int tag /* serial of the fusion launch, never 0 */,
                      unsigned int* tile_flags /* [gsdf_fuse_grid_blocks] hand-off flags, zeroed once */,
                      const uint32_t* tile_order /* [gsdf_fuse_grid_blocks] from gsdf_fuse_tile_order, on the device */,
                      float* log_rows /* nullable: frame log, written when use_dev_pose */, long long max_rows,
                      uint32_t* vis /* nullable: per-voxel frame bit-vectors */, int vis_words,
                      int debug /* path-forcing / measurement switches, honoured by -DGSDF_EXPERIMENTS builds only */,
                      unsigned int* ticket /* two device words, zeroed once: arrivals of finished workgroups of k_fuse / of k_fuse_resolve */,
                      int resolve_follows /* also queue k_fuse_resolve (long deferred lists) */,
                      unsigned int* host_note /* nullable, 2 pinned host words: length of the deferred list, tiles too big for the small LDS table */,
                      int far_table /* use the kernel with the larger LDS table */,
                      const gsdf_fuse_head* head /* nullable: also perform the closing head of optimize() (use_dev_pose only) */,
                      const float* next_depth /* nullable: the launch also computes the normals of this (the next) frame ... */,
                      float* next_nx, float* next_ny, float* next_nz /* ... into these planes */, int win,
                      const uint32_t* tile_stats /* this frame's tile statistics (written with its normals) */,
                      uint32_t* next_tile_stats /* the next frame's, written by the launch's normals workgroups */,
                      unsigned int next_token /* tracked frames: left in st->nrm_token by the normals role when it ran */);
int  gsdf_fuse_grid_blocks(int W, int H);
void gsdf_fuse_tile_order(int W, int H, uint32_t* order_host /* [gsdf_fuse_grid_blocks] */);
/* per-launch parameters of one Gauss-Newton pass (RigidOptimizer.h:57-62) */
struct gsdf_track_params {
    int pass_index, max_passes;
    float conv_sq, damping;
    unsigned int serial;          /* optimize() call number, for the host progress words */
    unsigned int* progress;       /* pinned host word: serial << 16 | done << 15 | passes, written by every pass head; nullable */
    int debug;                    /* experiment switches (gsdf_debug_flags >> 8); 0 in production */
    unsigned int rot;             /* tracker launches issued on this context so far, mod 3: selects the sum buffers */
    int n_track_blocks;           /* workgroups of the pass itself (set by the launcher); further ones compute normals tiles */
    int sampling;                 /* optimize_sampled's stride (RigidPointOptimizer.h:65); > 1: the geometry is the sampled grid, the
                                     depth image its compaction (gsdf_launch_subsample), no normals riders */
    int head_done;                /* the head of this launch (finish pass pass_index - 1) was performed by the fusion launch queued in
                                     front of it (gsdf_fuse_head): start from st->trk[pass_index & 1] */
};
/* NormalEstimator::compute of the frame being tracked, run by extra workgroups of its first pass (Scan3D loop) */
struct gsdf_normals_job {
    gsdf_ncache nc;
    float *nx, *ny, *nz;
    unsigned int* deferred_count; /* cleared for the k_fuse of this frame */
    uint32_t* stats;              /* the frame's tile statistics for its k_fuse ([tiles of 16 x 16 pixels][4], see gsdf_kernels.hip) */
    int r, ntx;                   /* window radius; tiles per image row (set by the launcher) */
    int tile_first, tile_count;   /* the tiles this launch computes: [tile_first, tile_first + tile_count); count 0 = all the rest */
    unsigned int token;           /* != 0: leave at once if st->nrm_token carries it (the frame's normals were computed ahead) */
};
void gsdf_launch_subsample(hipStream_t s, const float* depth, int W, int H, int sampling, float* out /* ceil(W/s) * ceil(H/s) */);
int  gsdf_normals_tiles(int W, int H);       /* normals tiles of a frame (workgroups of k_normals / of the normals role) */
void gsdf_launch_track_none(hipStream_t s, gsdf_dev_state* st);
void gsdf_launch_track_pass(hipStream_t s, const gsdf_frame_geom& g, const float* depth, gsdf_table tab,
                            gsdf_dev_state* st, double* partials /* 3 * GSDF_TRACK_ROWSET, zeroed */, int n_blocks,
                            const gsdf_track_params& tp, const gsdf_normals_job* normals /* nullable */);
/* optimize() as ONE launch (k_track_all): n_blocks co-resident workgroups exchange their sums through `rows`
 * (gsdf_track_all_rows_bytes(n_blocks), zeroed once); n_blocks <= 2 * GSDF_TRACK_MAXBLK */
void gsdf_launch_track_all(hipStream_t s, const gsdf_frame_geom& g, const float* depth, gsdf_table tab, gsdf_dev_state* st,
                           void* rows, unsigned int* abort_word /* device word, zeroed once */, int n_blocks,
                           const gsdf_track_params& tp, const gsdf_normals_job* normals /* nullable */);
size_t gsdf_track_all_rows_bytes(int n_blocks);
void gsdf_launch_set_pose(hipStream_t s, gsdf_dev_state* st, const float* pose7_dev_or_null,
                          const float pose7_host[7]);
void gsdf_launch_export(hipStream_t s, gsdf_table tab, size_t n_slots, unsigned long long* keys_out,
                        float* payload_out, unsigned long long* counter, long long max_n, int raw,
                        const uint32_t* vis, int vis_words, uint32_t* vis_out);
void gsdf_launch_export_raw(hipStream_t s, gsdf_table tab, size_t n_slots, int32_t* keys_out, float* payload_out,
                            unsigned long long* counter, long long max_n);
void gsdf_launch_merge_raw(hipStream_t s, gsdf_table tab, const int32_t* keys, const float* payload,
                           long long n, gsdf_dev_state* st);
void gsdf_launch_query(hipStream_t s, gsdf_table tab, float vs, float inv_vs, const float* pts, long long n,
                       float* dist, float* grad, float* w);
void gsdf_launch_get_voxels(hipStream_t s, gsdf_table tab, const int32_t* keys, long long n, float* payload, int32_t* found);

void gsdf_launch_raycast(hipStream_t s, gsdf_table tab, float vs, float inv_vs, int factor /* band half-width in voxels */, int W, int H, const float K[9],
                         const gsdf_pose_arg& pose, float zmin, float zmax, float* depth_dev, float* normals_dev_or_null,
                         unsigned long long* wg_counts /* nullable: [workgroups of the 16x16-pixel grid][8]: samples, records, fast / slow loop iterations of wave 0 (added to); start / end tick of the workgroup */,
                         int debug /* test build: 16384 = the sample-at-a-time kernel */);

/* iso-surface: bounding-box minimum (mn_dev preset to INT_MAX x3), then triangles + sort keys appended through `counter` */
void gsdf_launch_mesh(hipStream_t s, gsdf_table tab, size_t n_slots, float vs, float iso, int* mn_dev, const signed char* tri_table_dev,
                      float* tris_dev, unsigned long long* keys_dev, unsigned long long* counter, long long max_tris);

/* mesh export: device radix sort of (64-bit sweep key, triangle index) pairs (rocPRIM; tmp == nullptr: only *tmp_bytes is set),
 * index fill, and the gather of 9-float triangles into sorted order (gsdf_sort.hip) */
hipError_t gsdf_sort_keys_u64(void* tmp, size_t* tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, size_t n,
                              hipStream_t s);
hipError_t gsdf_unique_u64(void* tmp, size_t* tmp_bytes, const unsigned long long* sorted_in, unsigned long long* out,
                           unsigned long long* count_out, size_t n, hipStream_t s);
hipError_t gsdf_sort_pairs_u64(void* tmp, size_t* tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                               const uint32_t* vals_in, uint32_t* vals_out, size_t n, hipStream_t s);
void gsdf_launch_iota(hipStream_t s, uint32_t* idx, size_t n);
void gsdf_launch_gather_tris(hipStream_t s, const float* tris, const uint32_t* order, float* sorted, size_t n);

/* dense block exchange (frame-sharded fusion): list of block keys; pack / unpack of 64 x 5 raw sums per listed block */
void gsdf_launch_block_keys(hipStream_t s, gsdf_table tab, size_t n_blocks, unsigned long long* out_dev, unsigned long long* counter,
                            long long max_n);
void gsdf_launch_pack_blocks(hipStream_t s, gsdf_table tab, const unsigned long long* keys_dev, long long n, float* dense_dev);
void gsdf_launch_unpack_blocks(hipStream_t s, gsdf_table tab, const unsigned long long* keys_dev, long long n, const float* dense_dev,
                               gsdf_dev_state* st);

/* vis_ bit-vectors of the exchange: this rank's vectors shifted by `bit_offset` frames into a dense buffer (vw words per voxel of
 * every listed block) / the combined vectors stored; Sdf::counter_ := frames of all ranks */
void gsdf_launch_pack_vis(hipStream_t s, gsdf_table tab, const uint32_t* vis, int vw, long long bit_offset,
                          const unsigned long long* keys_dev, long long n, uint32_t* dense_dev);
void gsdf_launch_unpack_vis(hipStream_t s, gsdf_table tab, uint32_t* vis, int vw, const unsigned long long* keys_dev, long long n,
                            const uint32_t* dense_dev);
void gsdf_launch_set_frames(hipStream_t s, gsdf_dev_state* st, long long frames);

/* PhotoBA (gsdf_ba.hip): device-side problem description, same layout as the kernels' ba_args */
struct gsdf_ba_dev {
    gsdf_table tab;
    size_t n_slots;
    const uint32_t* vis;
    int vis_words;
    int n, W, H;
    const float* images;
    const float* R;
    const float* t;
    const int* frame_idx;
    float fx, fy, cx, cy, vs, reg_weight;
    float trunc_sq;               /* TRUNC_L2 (PhotometricOptimizer.cpp:364,:542): lambda^2, or < 0 for every other loss */
    /* nullable: the slots of the voxels inside the |dist| <= voxel size gate of getEnergy / solvePose (:285, :509), in slot order,
     * and their number (device word) -- gsdf_ba_compact.  With it the two gated sweeps visit only those voxels (8 % of a
     * surface map: without it 92 % of their lanes look at a record and leave) */
    const uint32_t* gate_list;
    const unsigned long long* gate_count;
    /* nullable: per entry of gate_list what getEnergy's first loop found for that voxel -- mean intensity over its keyframes, their
     * number, their set (24 B) -- written by the energy sweep, read by the pose sweep that follows it at the same state instead
     * of repeating that loop (solvePose :520-560 computes exactly getEnergy's :287-311 unless the loss truncates) */
    void* mean_cache;
};
/* the list above: ordered compaction of the table's slots (rocPRIM select over a counting iterator; tmp == nullptr: only
 * *tmp_bytes is set) */
hipError_t gsdf_ba_compact(hipStream_t s, const gsdf_ba_dev& d, uint32_t* list_out, unsigned long long* count_out, void* tmp, size_t* tmp_bytes);
void gsdf_launch_ba_energy(hipStream_t s, const gsdf_ba_dev& d, double* block_E, bool write_mean_cache = false /* needs gate_list and mean_cache */);
void gsdf_launch_ba_dist(hipStream_t s, const gsdf_ba_dev& d, float damping, double* block_cnt /* nullable: [2][gsdf_ba_blocks()] voxels, observations */);
void gsdf_launch_ba_pose(hipStream_t s, const gsdf_ba_dev& d, float* block_part, float* out, bool use_mean_cache = false /* written by an energy sweep at this very state */);
int  gsdf_ba_blocks(void);

#endif /* GSDF_KERNELS_H_ */
