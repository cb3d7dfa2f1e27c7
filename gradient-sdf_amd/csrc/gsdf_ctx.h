/*
 * gsdf_ctx.h -- the context behind the C-ABI handle (private to libgsdf.so: gsdf_capi.hip, gsdf_merge.hip).
 */
#ifndef GSDF_CTX_H_
#define GSDF_CTX_H_

#include "../../include/gsdf.h"
#include "gsdf_kernels.h"

#include <hip/hip_runtime.h>

#include <deque>
#include <string>
#include <utility>
#include <vector>

#define GSDF_SCRATCH_BYTES (256 * 1024)
#define GSDF_MX_BUFS 8
#define GSDF_GROW_MAX_LAG_DEFAULT 8       /* auto-grow: frame entries the host may be ahead of the newest finished block count */
#define GSDF_PROF_SLOTS 5     /* gsdf_profile: 0 normals, 1 fusion, 2 tracking launches, 3 raycast, 4 tracker (whole optimize) */

inline thread_local std::string g_gsdf_err;

struct gsdf_ctx;
int gsdf_flush_pending(gsdf_ctx* c);               /* gsdf_capi.hip: launch the deferred GT-pose fusion, if one waits */
int gsdf_grow_impl(gsdf_ctx* c, int new_capacity_log2);      /* gsdf_merge.hip: rehash into a larger table */
void gsdf_enqueue_block_count(gsdf_ctx* c, unsigned int tag);   /* gsdf_merge.hip: existing blocks | tag << 32 -> pinned words progress[4..5] */

inline int gsdf_fail(int code, const std::string& msg) {
    g_gsdf_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return gsdf_fail(GSDF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

struct gsdf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    /* a GT-pose fusion whose launch waits for the next gsdf_update_dev (its launch then also computes that frame's normals) */
    struct pending_fuse { bool valid = false; const float* depth = nullptr; gsdf_pose_arg pose; int set = 0; } pending;
    int defer = 1;                                 /* pipeline runs of gsdf_update_dev that way (GSDF_DEFER=0: launch at once) */
    int nrm_split = 30, nrm_split2 = 40;           /* per cent of a tracked frame's normals tiles computed in its first / second tracker launch (rest: third) */
    int nrm_parity = 0;                            /* which set of normal planes the next GT-pose fusion uses (0 / 1; set 2: tracked frames) */
    /* MapGradPixelSdf / Sdf members */
    float voxel_size = 0, voxel_size_inv = 0, T = 0, inv_T = 0;
    float zmin = 0.5f, zmax = 3.5f;                /* Sdf.h:67-68 */
    int factor = 0;
    /* table */
    int capacity_log2 = 0;
    size_t n_slots = 0;
    gsdf_table tab{ nullptr, nullptr, 0, nullptr };
    /* normal estimator + frame scratch */
    int W = 0, H = 0, win = 0;
    float K[9] = { 0 };
    float* planes = nullptr;                       /* 11 planes */
    float* depth_stage = nullptr;                  /* H2D staging for host-pointer entry points */
    float* normals = nullptr;                      /* 3 sets of 3 planes */
    uint32_t* tile_stats = nullptr;                /* per set: [fuse_blocks][4] statistics of the frame's fusion tiles (gsdf_kernels.hip: gsdf_tile_stats) */
    /* tracker */
    gsdf_dev_state* st = nullptr;
    double* partials = nullptr;                    /* 3 rotating buffers of tracker partial sums */
    unsigned int track_rot = 0;                    /* tracker launches issued so far, mod 3 (selects the sum buffers) */
    int track_blocks = 0;
    void* track_rows = nullptr;                    /* k_track_all: the workgroups' rows of sums, two buffers (pass parity) */
    unsigned int* track_abort = nullptr;           /* k_track_all: abort word */
    int persist = 0;                               /* optimize() as one launch (k_track_all) instead of one launch per pass */
    unsigned long long* blk_counters = nullptr;
    int fuse_blocks = 0;                           /* tiles of a frame */
    gsdf_deferred* deferred = nullptr;
    unsigned int* deferred_count = nullptr;
    unsigned int* fuse_ticket = nullptr;           /* arrivals of finished k_fuse workgroups (reset by the last one) */
    unsigned int deferred_cap = 0;
    unsigned int fuse_tag = 0;                     /* serial of the last fusion launch */
    unsigned int* tile_flags = nullptr;            /* per-tile hand-off flags of k_fuse */
    uint32_t* tile_order = nullptr;                /* launch order of the fusion tiles (gsdf_fuse_tile_order) */
    uint32_t* vis = nullptr;                       /* optional vis_ bit-vectors, n_slots x vis_words */
    int vis_words = 0;
    unsigned long long* rc_counts = nullptr;       /* raycaster: per-workgroup rows of (samples, records, fast / slow iterations of wave 0) */
    size_t rc_rows = 0;
    long long rc_iters[2] = { 0, 0 };              /* loop iterations of the workgroups' wave 0 as of the last gsdf_raycast_counters */
    /* gsdf_hint_next_depth_dev: the frame the NEXT gsdf_track_and_fuse_dev will be called with (set by the caller, consumed by the
     * next frame entry), and the frame whose normals a fusion launch has already computed into set nrm_ready_set (0 / 1) */
    bool fuse_head = true;                         /* the frame's first fusion launch performs the head of the first batch's last tracker launch (GSDF_FUSE_HEAD) */
    const float* hint_next = nullptr;
    const float* nrm_ready_depth = nullptr;
    int nrm_ready_set = -1;
    unsigned int prev_track_serial = 0; int prev_first_last = 0; bool prev_slow = false;   /* the last tracked frame: did it need more than its first batch (as far as the host knows)? */
    unsigned int nrm_ready_token = 0, nrm_token_ctr = 0;   /* what that fusion launch leaves in st->nrm_token when its gate was open */
    float* depth_sampled = nullptr;                /* the compacted pixels of gsdf_track_sampled (sampling > 1), lazily allocated */
    void* scratch = nullptr;                       /* device scratch of gsdf_query / gsdf_get_voxels for small batches (GSDF_SCRATCH_BYTES) */
    bool occ_dirty = false;                        /* blocks may have been inserted since the raycaster's filters (gsdf_table::occ) were built */
    /* the reference's map grows without bound (MapGradPixelSdf.h:65-68); here: gsdf_grow, or by itself when gsdf_set_auto_grow
     * named a limit -- every few fusions the number of existing blocks is counted into a pinned word, and a frame entry that
     * finds the key array more than GSDF_GROW_LOAD full doubles the table first */
    int auto_grow_max = 0;                         /* largest capacity_log2 auto-grow may reach; 0 = off */
    unsigned int grow_seq = 0;                     /* frame entries since auto-grow was switched on / the map was reset or grown */
    unsigned int grow_last_enq = 0;                /* entry whose count was enqueued last */
    unsigned int grow_prev_seq = 0, grow_prev_cnt = 0;   /* the newest finished count the growth rate was updated from */
    unsigned int grow_rate = 0;                    /* blocks a frame added lately (max over recent counts, decaying) */
    int grow_counts_seen = 0;                      /* finished counts seen since grow_seq restarted (the rate needs two) */
    bool grow_forget = false;                      /* set by gsdf_grow / gsdf_reset: restart the bookkeeping above */
    int grow_max_lag = GSDF_GROW_MAX_LAG_DEFAULT;
    long long grow_syncs = 0;                      /* entries that had to wait for an exact count (statistics for the tests) */
    unsigned int* grow_scratch = nullptr;          /* two device words of k_count_blocks */
    bool merged = false;                           /* gsdf_merge_allreduce has run: the map is the sum of all ranks (one-shot) */
    struct mx_buf { void* p = nullptr; size_t bytes = 0; } mx[GSDF_MX_BUFS];   /* scratch of the exchange (gsdf_merge.hip): grows, never shrinks */
    /* PhotoBA (PhotometricOptimizer) */
    int ba_n = 0;
    float ba_reg = 10.f;
    float ba_trunc_sq = -1.f;                      /* OptSettings::lambda_sq when loss == TRUNC_L2, else < 0 */
    float* ba_images = nullptr;
    float* ba_Rt = nullptr;                        /* device: n x 9 rotations then n x 3 translations */
    int* ba_frame_idx = nullptr;
    double* ba_block_E = nullptr;
    float* ba_block_part = nullptr;
    float* ba_Hb = nullptr;
    std::vector<float> ba_R, ba_t;                 /* host copies of the keyframe poses being optimised */
    uint32_t* ba_gate_list = nullptr;              /* slots of the voxels with |dist| <= voxel size (the gate of getEnergy / solvePose), slot order */
    void* ba_gate_tmp = nullptr;                   /* rocPRIM select scratch */
    size_t ba_gate_tmp_bytes = 0;
    unsigned long long* counter2 = nullptr;        /* device word: entries of ba_gate_list */
    bool ba_gate_fresh = false;                    /* the list matches the distances in the table */
    void* ba_mean = nullptr;                       /* per entry of ba_gate_list: what the last energy sweep's first loop found (24 B each; gsdf_ba_dev::mean_cache) */
    int ba_mean_on = 1;                            /* GSDF_BA_MEAN_CACHE (read by gsdf_ba_setup) */
    bool ba_mean_valid = false;                    /* ... at the very state (poses, distances, gate list) the next pose sweep will see */
    long long ba_last_voxels = 0, ba_last_obs = 0; /* what the last energy sweep read back counted (gsdf_ba_counters) */
    unsigned int track_serial = 0;                 /* optimize() call counter */
    volatile unsigned int* progress = nullptr;     /* pinned host words written by the tracker epilogue */
    unsigned int* progress_dev = nullptr;
    int adaptive = 1;                              /* issue tracker passes in batches, following the device (see enqueue_track) */
    int far_table = -1;                            /* fusion kernel's LDS table: -1 chosen per launch from the previous fusions, 0 / 1 pinned */
    int first_batch = 5, next_batch = 8;           /* launches per batch: 5 cover the usual <= 4 passes + their last head; a frame that needs
                                                      more is most likely one that runs all 25 (pass counts on the bench stream: 142 x <= 6, 7 x 7..22,
                                                      51 x 25) -- batches of 8 behind the first: 6 656 -> 6 815 frames/s on the default window (4 / 12 / 21: 6 656 / 6 780 / 6 760) */
    int lazy_fuse = 1;                             /* the frame's fusion is queued behind the first and the last batch of passes only; in between once optimize() has ended */
    unsigned long long* trace = nullptr;           /* test build: per-workgroup time stamps of k_fuse (gsdf_debug_flags & 64) */
    int debug = 0;                                 /* path-forcing / measurement switches (gsdf_debug_flags; test build only) */
    float* frame_log = nullptr;
    long long frame_log_cap = 0;
    /* misc */
    unsigned long long* counter = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events[GSDF_PROF_SLOTS];
    std::vector<hipEvent_t> event_pool;
    /* gsdf_mark: events recorded on the stream, retired in order */
    std::deque<std::pair<long long, hipEvent_t>> marks;
    std::vector<hipEvent_t> mark_pool, upload_pool;
    long long mark_serial = 0;
    hipStream_t copy_stream = nullptr;             /* gsdf_dev_upload_ahead: created on first use */
    std::deque<std::pair<long long, hipEvent_t>> uploads;
    long long upload_serial = 0;
    double prof_ms[GSDF_PROF_SLOTS] = { 0 };
    std::vector<float> prof_each[GSDF_PROF_SLOTS];  /* every launch's duration since the last gsdf_profile(c, 1) (gsdf_profile_read_launches) */
    long long prof_n[GSDF_PROF_SLOTS] = { 0 };

    gsdf_frame_geom geom() const {
        gsdf_frame_geom g;
        g.W = W; g.H = H;
        g.fx = K[0]; g.fy = K[4]; g.cx = K[2]; g.cy = K[5];
        g.vs = voxel_size; g.inv_vs = voxel_size_inv; g.T = T; g.inv_T = inv_T;
        g.zmin = zmin; g.zmax = zmax; g.factor = factor;
        return g;
    }
    gsdf_ncache ncache() const {
        const size_t N = (size_t)W * H;
        gsdf_ncache nc;
        nc.x0 = planes; nc.y0 = planes + N; nc.x0n = planes + 2 * N; nc.y0n = planes + 3 * N;
        nc.ninv = planes + 4 * N; nc.q11 = planes + 5 * N; nc.q12 = planes + 6 * N; nc.q13 = planes + 7 * N;
        nc.q22 = planes + 8 * N; nc.q23 = planes + 9 * N; nc.q33 = planes + 10 * N;
        return nc;
    }
};

#endif /* GSDF_CTX_H_ */
