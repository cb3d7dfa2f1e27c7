/*
 * gsdf_capi.hip -- host side of libgsdf.so: the C-ABI declared in include/gsdf.h.
 *
 * Owns the HBM-resident state (voxel hash table, cached normal-estimator planes, frame
 * scratch, tracker state) and enqueues the gfx950 kernels on one HIP stream.  There is no
 * CPU implementation behind these entry points: without a GPU gsdf_create fails with
 * GSDF_ERR_NO_DEVICE.
 */
#include "../../include/gsdf.h"
#include "../../include/gsdf_mc_tables.h"
#include "gsdf_ctx.h"
#include "gsdf_kernels.h"
#include "gsdf_math.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

static int fail(int code, const std::string& msg) { return gsdf_fail(code, msg); }

namespace {

hipEvent_t take_event(gsdf_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    /* timing events only: no system-scope fence when they are recorded (the cache write-back and invalidation it implies would
     * be charged to the kernels they bracket, and slow the ones behind them) */
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
    return e;
}

struct prof_scope {
    gsdf_ctx* c; int which; hipEvent_t a = nullptr, b = nullptr;
    prof_scope(gsdf_ctx* c_, int w) : c(c_), which(w) {
        if (c->profiling) { a = take_event(c); b = take_event(c); (void)hipEventRecord(a, c->stream); }
    }
    ~prof_scope() {
        if (c->profiling) { (void)hipEventRecord(b, c->stream); c->prof_events[which].push_back({ a, b }); }
    }
};

void prof_collect(gsdf_ctx* c) {
    for (int k = 0; k < GSDF_PROF_SLOTS; ++k) {
        for (auto& pr : c->prof_events[k]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                c->prof_ms[k] += ms; c->prof_n[k] += 1;
                if (c->prof_each[k].size() < (size_t)1 << 20) c->prof_each[k].push_back(ms);
            }
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
        c->prof_events[k].clear();
    }
}

int status_to_code(int status) {
    if (status & GSDF_STATUS_TABLE_FULL) return fail(GSDF_ERR_TABLE_FULL, "voxel hash table full (probe budget exhausted)");
    if (status & GSDF_STATUS_KEY_RANGE) return fail(GSDF_ERR_KEY_RANGE, "voxel index outside the packable +-2^20 range");
    if (status & GSDF_STATUS_TRACK_ABORT)
        return fail(GSDF_ERR_HIP, "tracking: the workgroups of the one-launch optimize() were not co-resident (GPU shared with another "
                                  "process?); that optimize() was abandoned -- set GSDF_PERSIST=0 to use one launch per pass");
    return GSDF_OK;
}

int require_frame(gsdf_ctx* c) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    if (!c->planes) return fail(GSDF_ERR_INVALID, "gsdf_normals_init must be called first (NEst == nullptr, MapGradPixelSdf.cpp:55-58)");
    return GSDF_OK;
}

/* Which size of the fusion kernel's LDS table (gsdf_kernels.hip, FUSE_LCAP): the larger one while many tiles did not fit the
 * small one lately (far geometry).  The count comes from a pinned word the last workgroup of every fusion writes; it lags by a
 * launch or two -- a hint, never a condition for correctness (either kernel fuses any tile). */
static int fuse_far_table(const gsdf_ctx* c) {
#ifdef GSDF_EXPERIMENTS
    if (c->debug & 1024) return 1;                         /* tests: pin the larger table */
    if (c->debug & 2048) return 0;
#endif
    if (c->far_table >= 0) return c->far_table;
    return c->progress && c->progress[3] * 16u > (unsigned int)c->fuse_blocks ? 1 : 0;
}

/* Auto-grow (gsdf_set_auto_grow): called at the top of the frame entries (gsdf_update_dev, gsdf_track_and_fuse_dev).
 *
 * The table is doubled once 45 % of its block entries are in use (the probe budget runs out near 95 %).  The load is the
 * count a 64-workgroup kernel leaves in a pinned word; it is enqueued at the top of every entry (every fourth while the
 * estimate is below 25 %), tagged with the entry's
 * number, and read without waiting -- so the number the host sees is some entries old, and the host may run ahead of the
 * device (an unsynchronised gsdf_update_dev loop).  What keeps the doubling in time:
 *  - the age of the count is known (the tag), and so is how fast the map grew lately (max over the recent counts, decaying):
 *    while  count + 2 x (entries not covered) x (blocks per frame)  stays below the 45 % the host does not wait;
 *  - otherwise, and whenever the count is more than grow_max_lag (8) entries old, the entry waits: for the lag it polls the
 *    pinned word (the device keeps running, no bubble); for the load it flushes, counts and synchronises -- the EXACT load
 *    decides the doubling;
 *  - until two counts have been seen (the first entries of a scan, after a reset or a doubling) every entry is exact: a
 *    first frame that fills half the table is seen before the second one is queued.
 * What remains possible: ONE frame that adds more blocks than the table has room for (more than half its entries beyond the
 * 45 %) still sets the sticky GSDF_STATUS_TABLE_FULL -- its samples are dropped where the probes run out, which no later
 * doubling can undo.  Size the initial table for at least two frames' blocks (the CLI's 2^22 holds 65 536 blocks, a
 * 640x480 frame at 1 cm touches ~9 000). */
static int auto_grow_step(gsdf_ctx* c) {
    if (!c->auto_grow_max || !c->progress) return GSDF_OK;
    volatile unsigned long long* word = reinterpret_cast<volatile unsigned long long*>(c->progress + 4);
    if (c->grow_forget) {                                   /* a reset or a doubling: the counts so far describe another table */
        c->grow_forget = false;
        c->grow_seq = 0; c->grow_prev_seq = 0; c->grow_prev_cnt = 0; c->grow_rate = 0; c->grow_counts_seen = 0; c->grow_last_enq = 0;
        HIP_TRY(hipStreamSynchronize(c->stream));           /* no count kernel of the old numbering is in flight any more */
        *word = 0ull;
    }
    const unsigned int k = ++c->grow_seq;
    const size_t cap_blocks = c->n_slots / GSDF_BLOCK_VOX;
    auto look = [&](unsigned int& cnt, unsigned int& seen) {
        const unsigned long long w = *word;
        cnt = (unsigned int)(w & 0xFFFFFFFFull); seen = (unsigned int)(w >> 32);
        if (seen > c->grow_prev_seq) {                       /* a newer finished count: update the growth per frame */
            if (c->grow_counts_seen > 0) {
                const unsigned int d = cnt > c->grow_prev_cnt ? cnt - c->grow_prev_cnt : 0u, n = seen - c->grow_prev_seq;
                const unsigned int rate = (d + n - 1) / n;
                c->grow_rate = std::max(rate, c->grow_rate - c->grow_rate / 16u);
            }
            c->grow_prev_seq = seen; c->grow_prev_cnt = cnt; ++c->grow_counts_seen;
        }
    };
    unsigned int cnt = 0, seen = 0;
    look(cnt, seen);
    /* the count tagged `seen` was enqueued at the top of entry `seen`: it holds the fusions of the entries before it, except
     * one that a pipelined gsdf_update_dev still kept back => entries seen - 1 ... k - 1 are not in it */
    unsigned int uncovered = k - seen + 1;
    if (c->grow_counts_seen >= 2 && (int)uncovered > c->grow_max_lag + 1) {
        /* far ahead of the device: wait for it to come within grow_max_lag / 2 entries (it is working; nothing is drained) */
        /* ... but never for a count that was not enqueued: counts are tagged with the entry that queued them (every fourth
         * entry while the estimate is low), so `seen` cannot pass grow_last_enq (with a small GSDF_GROW_MAX_LAG the
         * old target k - seen <= lag / 2 was unreachable and the loop spun out its whole timeout on every entry) */
        const int target = std::max(c->grow_max_lag / 2, (int)(k - c->grow_last_enq));
        const auto t0 = std::chrono::steady_clock::now();
        for (long spin = 0; (int)(k - seen) > target; ++spin) {
            look(cnt, seen);
            if (spin < 4096) { __builtin_ia32_pause(); continue; }
            if ((spin & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            std::this_thread::yield();
        }
        uncovered = k - seen + 1;
    }
    const bool safe = c->grow_counts_seen >= 2 && (int)uncovered <= c->grow_max_lag + 1 &&
                      ((size_t)cnt + 2u * (size_t)uncovered * std::max(c->grow_rate, 1u)) * 100u <= cap_blocks * 45u;
    if (!safe && c->capacity_log2 < c->auto_grow_max) {
        int rc = gsdf_flush_pending(c);                      /* a fusion kept back by the pipelined path counts too */
        if (rc) return rc;
        gsdf_enqueue_block_count(c, k);
        c->grow_last_enq = k;
        HIP_TRY(hipStreamSynchronize(c->stream));
        ++c->grow_syncs;
        look(cnt, seen);
        if ((size_t)cnt * 100u > cap_blocks * 45u) {
            rc = gsdf_grow_impl(c, c->capacity_log2 + 1);    /* (sets grow_forget: the next entry starts the bookkeeping afresh) */
            if (rc) return rc;
        }
        return GSDF_OK;
    }
    /* far below the limit (estimate under 25 % of the entries) a count every fourth entry will do: the 3 us kernel and its
     * launch boundary are ~4 % of a tracked frame */
    const bool near = ((size_t)cnt + 2u * (size_t)uncovered * std::max(c->grow_rate, 1u)) * 100u > cap_blocks * 25u;
    if (near || k - c->grow_last_enq >= 4u) { gsdf_enqueue_block_count(c, k); c->grow_last_enq = k; }
    return GSDF_OK;
}

/* the tile statistics that belong to a set of normal planes (written by whatever computes the normals, read by the set's k_fuse) */
static uint32_t* stats_of(const gsdf_ctx* c, const float* nrm) {
    const size_t N = (size_t)c->W * c->H;
    const size_t set = (size_t)(nrm - c->normals) / (3 * N);
    return c->tile_stats + set * (size_t)c->fuse_blocks * 4;
}

/* one k_fuse launch: depth + its normal planes `nrm` (3 x N floats) -> the map.  next_depth (nullable): the launch's extra
 * workgroups compute the normals of that frame into next_nrm */
int launch_fuse(gsdf_ctx* c, const float* depth_dev, const float* nrm, const gsdf_pose_arg& pose, int use_dev_pose,
                const float* next_depth, float* next_nrm, const gsdf_fuse_head* head = nullptr, unsigned int next_token = 0u) {
    const size_t N = (size_t)c->W * c->H;
    c->occ_dirty = true;                                      /* new blocks: the raycaster's filters are rebuilt when it next runs */
    {
        prof_scope ps(c, 1);
        c->fuse_tag += 1;
        if (c->fuse_tag == 0) {                                /* wrapped: no stale flag may equal a new tag */
            c->fuse_tag = 1;
            HIP_TRY(hipMemsetAsync(c->tile_flags, 0, (size_t)c->fuse_blocks * sizeof(unsigned int), c->stream));
        }
        gsdf_launch_fuse(c->stream, c->geom(), c->ncache(), depth_dev, nrm, nrm + N, nrm + 2 * N, pose,
                         use_dev_pose, c->tab, c->st, c->blk_counters, c->deferred, c->deferred_count,
                         c->deferred_cap, c->fuse_tag, c->tile_flags, c->tile_order, c->frame_log, c->frame_log_cap, c->vis, c->vis_words,
                         c->debug & 0xFFFF, c->fuse_ticket,
                         /* long deferred lists lately (the note lags by a launch or two: a hint, not a condition) */
                         c->progress && c->progress[2] > 8192u ? 1 : 0, c->progress ? c->progress_dev + 2 : nullptr,
                         /* many tiles did not fit the small LDS table lately (far geometry): the kernel with the larger one.
                          * Like the note above a hint that lags by a launch or two, never a condition for correctness. */
                         fuse_far_table(c), head,
                         next_depth, next_nrm, next_nrm ? next_nrm + N : nullptr, next_nrm ? next_nrm + 2 * N : nullptr, c->win,
                         stats_of(c, nrm), next_nrm ? stats_of(c, next_nrm) : nullptr, next_token);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GSDF_ERR_HIP, std::string("fusion launch: ") + hipGetErrorString(e));
    return GSDF_OK;
}

/* normals (unless the frame's were computed beside its first tracker pass: normals_done) + fusion, in stream order */
int enqueue_fuse(gsdf_ctx* c, const float* depth_dev, const gsdf_pose_arg& pose, int use_dev_pose, bool normals_done, int set,
                 const float* next_depth, const gsdf_fuse_head* head, int next_set, unsigned int next_token) {
    const size_t N = (size_t)c->W * c->H;
    /* tracked frames: set 2, filled beside the first tracker passes -- or the set the PREVIOUS frame's fusion filled in its
     * tail (gsdf_hint_next_depth_dev) */
    float* nrm = c->normals + (size_t)set * 3 * N;
    if (!normals_done) {
        nrm = c->normals + (size_t)c->nrm_parity * 3 * N;
        c->nrm_parity ^= 1;
        c->nrm_ready_depth = nullptr;                  /* (sets 0 / 1 are also where a hinted frame's normals wait) */
        prof_scope ps(c, 0);
        gsdf_launch_normals(c->stream, c->geom(), c->win, c->ncache(), depth_dev, nrm, nrm + N, nrm + 2 * N, nullptr, nullptr, stats_of(c, nrm));
        next_depth = nullptr;
    }
    /* gsdf_hint_next_depth_dev: the launch's last workgroups compute NormalEstimator::compute of the NEXT frame into set
     * next_set -- if the launch's gate is open -- and leave next_token in st->nrm_token */
    float* next_nrm = next_depth ? c->normals + (size_t)next_set * 3 * N : nullptr;
    return launch_fuse(c, depth_dev, nrm, pose, use_dev_pose, next_depth, next_nrm, head, next_token);
}

/* RigidPointOptimizer::optimize_sampled as a chain of per-pass launches.  The convergence test, the pose update
 * and the done/converged flags live on the device; launches after `done` return immediately.  Launch k > 0
 * first finishes pass k-1 (reduce, solve, update: the "head"), launch `iters` is head-only.
 *
 * The host issues the launches in BATCHES and never waits in the common case: the first batch covers the usual
 * 3-4 passes; behind every batch it queues the frame's fusion (fuse_after: the Scan3D loop body), which is gated
 * on the device by done && converged, so it runs exactly once -- behind the batch in which optimize() ended.
 * While that batch (and the fusion) execute, the host follows one pinned word the pass heads write (serial,
 * done, passes) and only issues a further batch if the optimisation has not ended by the last head of the
 * previous one.  Correctness never depends on what the host sees: a late or lost observation only costs empty
 * launches. */
int enqueue_fuse(gsdf_ctx* c, const float* depth_dev, const gsdf_pose_arg& pose, int use_dev_pose, bool normals_done, int set = 2,
                 const float* next_depth = nullptr, const gsdf_fuse_head* head = nullptr, int next_set = 0, unsigned int next_token = 0u);

/* 1 = optimize() ended, 0 = the head of launch `last` ran and it has not ended, -1 = gave up waiting */
int follow_progress(gsdf_ctx* c, unsigned int serial, int last) {
    /* a batch takes ~50 us: spin briefly, then yield the core between looks; bounded by TIME (2 s), not by iterations */
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        const unsigned int w = c->progress[0];
        if ((w >> 16) == serial) {
            if (w & 0x8000u) return 1;
            if ((int)(w & 0x7FFFu) >= last) return 0;
        }
        if (spin < 4096) { __builtin_ia32_pause(); continue; }
        if ((spin & 63) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) return -1;
        std::this_thread::yield();
    }
}

int enqueue_track(gsdf_ctx* c, const float* depth_dev, int iters, float conv, float damping, bool fuse_after, int sampling = 1) {
    gsdf_frame_geom g = c->geom();
    if (sampling > 1) {
        /* optimize_sampled(depth, K, sampling) -- RigidPointOptimizer.cpp:62: the pixels (x, y) = (i * s, j * s).  They are
         * compacted into their own image (so that the pass kernel's coalesced pixel -> lane mapping serves them unchanged) and the
         * passes run on the grid of sampled pixels; k_track_pass<true> turns a grid index back into the pixel coordinate. */
        if (fuse_after) return fail(GSDF_ERR_INVALID, "the frame loop tracks every pixel (RigidPointOptimizer.h:69-72)");
        const int Ws = (c->W + sampling - 1) / sampling, Hs = (c->H + sampling - 1) / sampling;
        if (!c->depth_sampled) HIP_TRY(hipMalloc((void**)&c->depth_sampled, (size_t)((c->W + 1) / 2) * ((c->H + 1) / 2) * sizeof(float)));
        gsdf_launch_subsample(c->stream, depth_dev, c->W, c->H, sampling, c->depth_sampled);
        depth_dev = c->depth_sampled;
        g.W = Ws; g.H = Hs;
    }
    gsdf_pose_arg unused;
    std::memset(&unused, 0, sizeof(unused));
    /* gsdf_hint_next_depth_dev: (1) this frame's normals may already lie in set 0 / 1 -- the previous frame's fusion computed them
     * in its tail IF IT RAN (a fusion whose gate stays closed does not: its tail is not idle, the tiles would run in the open);
     * this frame's riders are queued as ever and leave at once when they find the frame's token in st->nrm_token; (2) every
     * fusion launch of this frame carries the normals role for the hinted NEXT frame (at most one of them passes the gate).
     * Not in event-timed replays (gsdf_profile): there the fusion launch is the plain one, so that what bench.py quotes as
     * k_fuse's duration is the fusion work alone. */
    const bool pre = fuse_after && !c->profiling && c->nrm_ready_depth == depth_dev && c->nrm_ready_set >= 0 && iters > 0;
    const int fuse_set = pre ? c->nrm_ready_set : 2;
    const unsigned int ride_token = pre ? c->nrm_ready_token : 0u;
    /* Both round-6 steps -- the next frame's normals in the fusion's tail, the fusion launch standing in for the first batch's last
     * tracker launch -- pay where optimize() ends within the first batch and the fusion RUNS behind it.  Frames that converge late or
     * never come in stretches (the reference's tracker cycles on them), and there the extra workgroups only look at a closed gate
     * and the stand-in's head is followed by a head-less launch anyway (measured: -1 ... -2 % on the bench's default window): the
     * host takes the old route while the last tracked frame whose end it KNOWS needed more than the first batch (a hint for the
     * hint: the progress word of that frame's heads, read without waiting -- in such a stretch the host has just waited for it). */
    if (fuse_after && c->progress && c->prev_track_serial) {
        const unsigned int w = c->progress[0];
        if ((w >> 16) == c->prev_track_serial && (w & 0x8000u)) c->prev_slow = (int)(w & 0x7FFFu) > c->prev_first_last;
    }
    /* ... and not while the fusion uses the larger LDS table (far geometry): that instantiation with the normals role is at the
     * register limit and moves its records unpaired, with the head it spills -- both cost more than the step saves */
    const bool new_route = fuse_after && !c->profiling && iters > 0 && !c->prev_slow && !fuse_far_table(c);
    const float* hint = new_route ? c->hint_next : nullptr;
    c->nrm_ready_depth = nullptr;        /* consumed (or not ours) */
    c->hint_next = nullptr;
    if (hint == depth_dev) hint = nullptr;
    int next_set = 0;
    unsigned int next_token = 0u;
    if (hint) {
        next_set = fuse_set == 0 ? 1 : 0;                     /* whichever of sets 0 / 1 this frame's fusion does not read */
        if (++c->nrm_token_ctr == 0u) c->nrm_token_ctr = 1u;
        next_token = c->nrm_token_ctr;
        c->nrm_ready_depth = hint; c->nrm_ready_set = next_set; c->nrm_ready_token = next_token;
    }
    if (iters <= 0) {
        gsdf_launch_track_none(c->stream, c->st);
        return fuse_after ? enqueue_fuse(c, depth_dev, unused, 1, false) : GSDF_OK;
    }
    if (iters > 0x7FFF) return fail(GSDF_ERR_INVALID, "num_iterations must be <= 32767");
    gsdf_normals_job nj;
    std::memset(&nj, 0, sizeof(nj));
    if (fuse_after) {                    /* the frame's normals ride along with its first passes (unless computed ahead: ride_token) */
        const size_t N = (size_t)c->W * c->H;
        nj.nc = c->ncache();
        nj.nx = c->normals + (size_t)(3 * fuse_set) * N; nj.ny = nj.nx + N; nj.nz = nj.nx + 2 * N;    /* set 2, or the hinted one */
        nj.token = ride_token;
        nj.deferred_count = c->deferred_count;
        nj.stats = stats_of(c, nj.nx);
        nj.r = c->win / 2; nj.ntx = 0;
        nj.tile_first = 0; nj.tile_count = 0;
    }
    gsdf_track_params tp;
    tp.max_passes = iters;
    tp.conv_sq = conv * conv;                                /* RigidOptimizer.h:72 */
    tp.damping = damping;
    c->track_serial = (c->track_serial + 1u) & 0xFFFFu;
    if (c->track_serial == 0) c->track_serial = 1;
    tp.serial = c->track_serial;
    const bool adaptive = c->adaptive && c->progress;
    if (fuse_after) {                                        /* (the last launch index of this frame's first batch) */
        c->prev_track_serial = tp.serial;
        c->prev_first_last = std::min(iters, (adaptive ? c->first_batch : iters + 1) - 1);
    }
    tp.progress = adaptive ? c->progress_dev : nullptr;
    tp.debug = c->debug >> 16;
    tp.n_track_blocks = c->track_blocks;
    tp.sampling = sampling;
    tp.head_done = 0;
    if (sampling == 1 && c->persist && c->track_rows && c->track_blocks <= 2 * GSDF_TRACK_MAXBLK) {
        /* the whole optimize() as one launch; the frame's fusion, gated on the device by done && converged, right behind it */
        tp.pass_index = 0;
        tp.rot = 0;
        tp.progress = c->progress_dev;
        {
            prof_scope ps(c, 2);
            gsdf_launch_track_all(c->stream, g, depth_dev, c->tab, c->st, c->track_rows, c->track_abort, c->track_blocks, tp,
                                  fuse_after ? &nj : nullptr);
        }
        if (fuse_after) {
            const int rc = enqueue_fuse(c, depth_dev, unused, 1, true, fuse_set, hint, nullptr, next_set, next_token);   /* main_scan_3d.cpp:261-265 */
            if (rc) return rc;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(GSDF_ERR_HIP, std::string("tracking launch: ") + hipGetErrorString(e));
        return GSDF_OK;
    }
    int k = 0, batch_no = 0;
    int batch = adaptive ? c->first_batch : iters + 1;
    /* The frame's first fusion launch stands in for the LAST tracker launch of the first batch (k_fuse<.., HEAD>): that launch's
     * head -- for the usual frame the closing one of optimize() -- runs in workgroup 0 of the fusion launch, which every other
     * workgroup of it follows; a launch and a kernel boundary less per frame.  If optimize() has not ended with that head the
     * next batch starts with a tracker launch of the same index that skips its head (head_done).  Not in event-timed replays:
     * what bench.py quotes as k_fuse's duration is the fusion work alone. */
    const bool fuse_head = new_route && adaptive && c->fuse_head;
    bool headless = false;                                   /* the next launch's head was performed by a fusion launch */
    while (k <= iters) {
        const int last = std::min(iters, k + batch - 1);
        const bool stand_in = fuse_head && batch_no == 0 && last >= 1;
        for (; k <= last; ++k) {
            if (stand_in && k == last) continue;             /* its head is the fusion launch's */
            tp.pass_index = k;
            tp.head_done = headless ? 1 : 0;
            headless = false;
            tp.rot = c->track_rot;
            c->track_rot = (c->track_rot + 1u) % 3u;          /* kept in [0, 3): no discontinuity at wrap-around */
            /* The frame's normals tiles ride in its first launches: nrm_split per cent in launch 0, nrm_split2 in launch 1, the rest
             * in launch 2.  Launch 1 always exists (it is at least the head of pass 0), launch 2 when the first batch has three
             * launches and optimize() may take two passes.  Next to the one tracker workgroup per CU a launch hides a share of the
             * tiles; launch 0 has no head and ends early, so it takes the smallest one.  Tile 0 (it resets the frame's deferred list)
             * stays in launch 0. */
            const gsdf_normals_job* job = nullptr;
            if (fuse_after && k <= 2) {
                const int tiles = gsdf_normals_tiles(c->W, c->H);
                /* tracker launches of the first batch (the fusion launch may stand in for its last one: no riders there) */
                const int nl = batch_no == 0 ? (stand_in ? last : last + 1) : 3;
                const bool three = iters >= 2 && batch >= 3 && nl >= 3;
                const int b1 = nl >= 2 ? std::max(1, tiles * c->nrm_split / 100) : tiles;
                const int b2 = three ? std::min(tiles, std::max(b1, tiles * (c->nrm_split + c->nrm_split2) / 100)) : tiles;
                const int lo = k == 0 ? 0 : k == 1 ? b1 : b2, hi = k == 0 ? b1 : k == 1 ? b2 : tiles;
                if (hi > lo && (k < 2 || three)) { nj.tile_first = lo; nj.tile_count = hi - lo; job = &nj; }
            }
            prof_scope ps(c, 2);
            gsdf_launch_track_pass(c->stream, g, depth_dev, c->tab, c->st, c->partials, c->track_blocks, tp, job);
        }
        /* The fusion is queued behind the FIRST batch unseen (the usual frame ends there and must not wait for the host) and
         * behind the last one (nothing follows it).  Behind the batches in between it would nearly always be a gated launch
         * that finds optimize() still running -- 3 us each, 3 per frame that never converges -- so there it is queued only
         * once the progress word says that optimize() has ended (the frame that converges late pays the host's look). */
        bool fuse_queued = false;
        if (fuse_after && (batch_no == 0 || last == iters || !c->lazy_fuse)) {
            gsdf_fuse_head hd;
            std::memset(&hd, 0, sizeof(hd));
            if (stand_in) {
                hd.k = last;
                hd.rot_prev = (c->track_rot + 2u) % 3u;      /* the buffer the last launch issued (pass last - 1) accumulated into */
                hd.conv_sq = tp.conv_sq; hd.damping = tp.damping; hd.max_passes = tp.max_passes;
                hd.serial = tp.serial; hd.progress = tp.progress; hd.rows = c->partials; hd.debug = tp.debug;
            }
            const int rc = enqueue_fuse(c, depth_dev, unused, 1, true, fuse_set, hint, stand_in ? &hd : nullptr, next_set, next_token);   /* main_scan_3d.cpp:261-265 */
            if (rc) return rc;
            fuse_queued = true;
        }
        ++batch_no;
        if (last == iters) break;                            /* the head-only launch always ends optimize() */
        int ended = follow_progress(c, tp.serial, last);
        if (ended < 0) {                                     /* the device is far behind: wait for it properly */
            gsdf_dev_state s;
            if (hipMemcpyAsync(&s, c->st, sizeof(s), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess)
                return fail(GSDF_ERR_HIP, "tracking: device state read failed");
            ended = s.done;
        }
        if (ended) {
            if (fuse_after && !fuse_queued) {
                const int rc = enqueue_fuse(c, depth_dev, unused, 1, true, fuse_set, hint, nullptr, next_set, next_token);
                if (rc) return rc;
            }
            break;
        }
        if (stand_in) { k = last; headless = true; }         /* launch `last` comes now, without the head the fusion launch performed */
        batch = c->next_batch;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GSDF_ERR_HIP, std::string("tracking launch: ") + hipGetErrorString(e));
    return GSDF_OK;
}

int read_state(gsdf_ctx* c, gsdf_dev_state* out) {
    HIP_TRY(hipMemcpyAsync(out, c->st, sizeof(gsdf_dev_state), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSDF_OK;
}

} // namespace

/* launches the GT-pose fusion that waits for its successor (gsdf_update_dev), if any */
int gsdf_flush_pending(gsdf_ctx* c) {
    if (!c || !c->pending.valid) return GSDF_OK;
    c->pending.valid = false;
    if (hipSetDevice(c->device) != hipSuccess) return fail(GSDF_ERR_HIP, "hipSetDevice");
    const size_t N = (size_t)c->W * c->H;
    return launch_fuse(c, c->pending.depth, c->normals + (size_t)c->pending.set * 3 * N, c->pending.pose, 0, nullptr, nullptr);
}
#define GSDF_FLUSH(c) do { if ((c) && (c)->pending.valid) { const int rc_ = gsdf_flush_pending(c); if (rc_) return rc_; } } while (0)
/* The staging entries (gsdf_dev_upload*) write device memory the caller names.  A GT-pose fusion that still waits for its
 * successor reads its depth image when it is LAUNCHED, so a copy into that image has to come behind the launch: the pattern
 * upload(buf) -> update_dev(buf) -> upload(buf) -> update_dev(buf) on ONE staging buffer is correct by stream order only if
 * the waiting fusion is launched before the second copy is queued.  Copies elsewhere leave the pipelining alone. */
static int flush_if_overlaps(gsdf_ctx* c, const void* dst, int64_t bytes) {
    if (c && (c->nrm_ready_depth || c->hint_next)) {
        /* a copy into a frame whose normals were computed ahead (gsdf_hint_next_depth_dev), or that is hinted: those belong to the
         * old contents -- forget them, the frame's own tracker launches compute its normals then */
        const uintptr_t a0 = (uintptr_t)dst, a1 = a0 + (uintptr_t)std::max<int64_t>(bytes, 0);
        const size_t fb = (size_t)c->W * c->H * sizeof(float);
        for (const float** q : { &c->nrm_ready_depth, &c->hint_next })
            if (*q) { const uintptr_t b0 = (uintptr_t)*q; if (a0 < b0 + fb && b0 < a1) *q = nullptr; }
    }
    if (!c || !c->pending.valid) return GSDF_OK;
    const uintptr_t a0 = (uintptr_t)dst, a1 = a0 + (uintptr_t)std::max<int64_t>(bytes, 0);
    const uintptr_t b0 = (uintptr_t)c->pending.depth, b1 = b0 + (size_t)c->W * c->H * sizeof(float);
    if (a0 < b1 && b0 < a1) return gsdf_flush_pending(c);
    return GSDF_OK;
}

extern "C" {

const char* gsdf_last_error(void) { return g_gsdf_err.c_str(); }
#ifdef GSDF_EXPERIMENTS
/* Test / measurement build only (libgsdf_test.so, make EXPERIMENTS=1); not part of include/gsdf.h and absent from the
 * production library.  Per context.  Bits 0-15 go to k_fuse: 4 every tile defers, 256 single band, 512 four bands,
 * 8192 every hand-off wait expires at once, 1024 / 2048 pin the larger / smaller LDS table, 1/2/16/32/64/128/4096 measurement
 * switches of tools/; bits 16+ go to the tracker. */
#define GSDF_TRACE_WG 8192
#define GSDF_TRACE_COLS 16
int gsdf_debug_flags(gsdf_ctx* c, int flags) {
    GSDF_FLUSH(c);
    if (!c) return GSDF_ERR_INVALID;
    c->debug = flags;
    if (flags & 64) {                                  /* k_fuse trace: GSDF_TRACE_COLS time stamps per workgroup, pointer in dbg[23] */
        if (hipStreamSynchronize(c->stream) != hipSuccess) return GSDF_ERR_HIP;
        if (!c->trace && hipMalloc((void**)&c->trace, (size_t)GSDF_TRACE_WG * GSDF_TRACE_COLS * 8) != hipSuccess) return GSDF_ERR_HIP;
        if (hipMemset(c->trace, 0, (size_t)GSDF_TRACE_WG * GSDF_TRACE_COLS * 8) != hipSuccess) return GSDF_ERR_HIP;
        const unsigned long long ptr = (unsigned long long)(uintptr_t)c->trace;
        if (hipMemcpy(&c->st->dbg[23], &ptr, 8, hipMemcpyHostToDevice) != hipSuccess) return GSDF_ERR_HIP;
    }
    return GSDF_OK;
}
/* the trace of the last k_fuse launch: n_wg rows of GSDF_TRACE_COLS (16) values */
int gsdf_debug_trace(gsdf_ctx* c, unsigned long long* out, int n_wg) {
    GSDF_FLUSH(c);
    if (!c || !out || !c->trace || n_wg > GSDF_TRACE_WG) return GSDF_ERR_INVALID;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return GSDF_ERR_HIP;
    if (hipMemcpy(out, c->trace, (size_t)n_wg * GSDF_TRACE_COLS * 8, hipMemcpyDeviceToHost) != hipSuccess) return GSDF_ERR_HIP;
    return GSDF_OK;
}
/* the raycaster's per-workgroup rows (8 values each, see gsdf_launch_raycast), tools/raycast_bench.py */
int gsdf_debug_raycast_rows(gsdf_ctx* c, unsigned long long* out, int n_rows) {
    GSDF_FLUSH(c);
    if (!c || !out || !c->rc_counts || (size_t)n_rows > c->rc_rows) return GSDF_ERR_INVALID;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return GSDF_ERR_HIP;
    if (hipMemcpy(out, c->rc_counts, (size_t)n_rows * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return GSDF_ERR_HIP;
    return GSDF_OK;
}
/* experiments with the launch order of the fusion tiles (tools/fuse_order.py): the per-tile counter rows (last n_upd, last
 * n_valid, totals) and a caller-made order (same encoding as gsdf_fuse_tile_order; n = workgroups) */
int gsdf_debug_tile_counters(gsdf_ctx* c, unsigned long long* out, int n_rows) {
    GSDF_FLUSH(c);
    if (!c || !out || !c->blk_counters || n_rows > c->fuse_blocks) return GSDF_ERR_INVALID;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return GSDF_ERR_HIP;
    if (hipMemcpy(out, c->blk_counters, (size_t)n_rows * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return GSDF_ERR_HIP;
    return GSDF_OK;
}
int gsdf_debug_set_tile_order(gsdf_ctx* c, const uint32_t* order, int n) {
    GSDF_FLUSH(c);
    if (!c || !order || !c->tile_order || n != c->fuse_blocks) return GSDF_ERR_INVALID;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return GSDF_ERR_HIP;
    if (hipMemcpy(c->tile_order, order, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return GSDF_ERR_HIP;
    return GSDF_OK;
}
int gsdf_debug_get_tile_order(gsdf_ctx* c, uint32_t* order, int* n) {
    if (!c || !order || !n || !c->tile_order) return GSDF_ERR_INVALID;
    if (hipMemcpy(order, c->tile_order, (size_t)c->fuse_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return GSDF_ERR_HIP;
    *n = c->fuse_blocks;
    return GSDF_OK;
}
int gsdf_debug_read(gsdf_ctx* c, unsigned long long out[24]) {
    GSDF_FLUSH(c);
    if (!c || !out) return GSDF_ERR_INVALID;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return GSDF_ERR_HIP;
    gsdf_dev_state h;
    if (hipMemcpy(&h, c->st, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return GSDF_ERR_HIP;
    for (int i = 0; i < 24; ++i) out[i] = h.dbg[i];
    return GSDF_OK;
}
const char* gsdf_version(void) { return "gsdf-mi355x 0.2 (gfx950) +experiments"; }
#else
const char* gsdf_version(void) { return "gsdf-mi355x 0.2 (gfx950)"; }
#endif

static int create_impl(gsdf_ctx** out, float voxel_size, float trunc_dist, int capacity_log2, int device, bool other_queue);
int gsdf_create(gsdf_ctx** out, float voxel_size, float trunc_dist, int capacity_log2, int device) {
    return create_impl(out, voxel_size, trunc_dist, capacity_log2, device, false);
}
/* n contexts for n frame shards on ONE device (gsdf_merge_from adds them up): their streams are created with the device's
 * HIGHEST stream priority.  The HIP runtime keeps one pool of hardware queues per priority and hands a new stream the
 * least-used queue of its pool; the normal pool is shared with every other stream of the process (the contexts' copy streams,
 * PyTorch's, ...), and two streams that land in one queue do not overlap -- measured inside bench.py: two normal-priority
 * contexts fused 24 500 frames/s where two on separate queues reach 31 600.  Nothing else uses the high-priority pool, so up to
 * four shard contexts get a queue each, at EQUAL priority. */
int gsdf_create_shards(gsdf_ctx** out, int n, float voxel_size, float trunc_dist, int capacity_log2, int device) {
    if (!out || n < 1 || n > 4) return fail(GSDF_ERR_INVALID, "gsdf_create_shards: 1..4 contexts (one hardware queue each)");
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    for (int i = 0; i < n; ++i) {
        const int rc = create_impl(&out[i], voxel_size, trunc_dist, capacity_log2, device, true);
        if (rc) { for (int j = 0; j < i; ++j) { gsdf_destroy(out[j]); out[j] = nullptr; } return rc; }
    }
    return GSDF_OK;
}
static int create_impl(gsdf_ctx** out, float voxel_size, float trunc_dist, int capacity_log2, int device, bool other_queue) {
    if (!out) return fail(GSDF_ERR_INVALID, "out == NULL");
    *out = nullptr;
    if (!(voxel_size > 0.f) || !(trunc_dist > 0.f)) return fail(GSDF_ERR_INVALID, "voxel_size and trunc_dist must be > 0");
    if (capacity_log2 < 10 || capacity_log2 > 30) return fail(GSDF_ERR_INVALID, "capacity_log2 must be in [10, 30]");
    if (!(trunc_dist < 2.f)) return fail(GSDF_ERR_INVALID, "trunc_dist must be < 2 m (fixed-point range of the fusion kernel's accumulators)");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        return fail(GSDF_ERR_NO_DEVICE, "no HIP device visible: libgsdf has no CPU fallback");
    }
    if (device < 0 || device >= n_dev) return fail(GSDF_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    gsdf_ctx* c = new gsdf_ctx();
    c->device = device;
    c->voxel_size = voxel_size;
    c->voxel_size_inv = (float)(1. / voxel_size);            /* MapGradPixelSdf.h:99-103 */
    c->T = trunc_dist;
    c->inv_T = (float)(1. / trunc_dist);                     /* Sdf.h:103-107 */
    c->factor = (int)std::floor(c->T / c->voxel_size);       /* MapGradPixelSdf.cpp:79 */
    c->capacity_log2 = capacity_log2;
    c->n_slots = (size_t)1 << capacity_log2;
    hipError_t e;
    int prio_least = 0, prio_greatest = 0;
    if (other_queue) {
        /* the separate-queue guarantee of gsdf_create_shards rests on a priority pool of its own: without one it is
         * not claimed -- an error, not a silent default-priority stream */
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess || prio_greatest == prio_least) {
            (void)hipGetLastError();
            delete c;
            return fail(GSDF_ERR_HIP, "gsdf_create_shards: the device offers no stream priority range, so separate hardware queues cannot be guaranteed (use gsdf_create)");
        }
    }
    if ((e = (other_queue ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest)
                          : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking))) != hipSuccess ||
        (e = hipMalloc((void**)&c->tab.vox, c->n_slots * sizeof(gsdf_payload))) != hipSuccess ||
        (e = hipMalloc((void**)&c->tab.bkeys, (c->n_slots / GSDF_BLOCK_VOX) * sizeof(unsigned long long))) != hipSuccess ||
        /* block filter (64 bits per block entry) followed by the cell filter (1 bit per block entry, at least one word) */
        (e = hipMalloc((void**)&c->tab.occ, c->n_slots / 8 + std::max<size_t>(c->n_slots / GSDF_BLOCK_VOX / 8, 4))) != hipSuccess ||
        (e = hipMalloc((void**)&c->st, sizeof(gsdf_dev_state))) != hipSuccess ||
        (e = hipMalloc((void**)&c->counter, sizeof(unsigned long long))) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        std::string m = std::string("gsdf_create: ") + hipGetErrorString(e);
        gsdf_destroy(c);
        return fail(GSDF_ERR_HIP, m);
    }
    {
        const char* env = getenv("GSDF_DEFER");                    /* 0: every gsdf_update_dev launches its normals and its fusion at once */
        if (env) c->defer = atoi(env);
    }
    c->tab.block_mask = (uint32_t)(c->n_slots / GSDF_BLOCK_VOX - 1);
    {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
            std::memset(hp, 0, 64);
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) { c->progress = (volatile unsigned int*)hp; c->progress_dev = (unsigned int*)dp; }
            else (void)hipHostFree(hp);
        }
        (void)hipGetLastError();
        const char* env = getenv("GSDF_ADAPTIVE");
        if (env) c->adaptive = atoi(env);
        if ((env = getenv("GSDF_FIRST_BATCH")) && atoi(env) >= 2) c->first_batch = atoi(env);
        if ((env = getenv("GSDF_FUSE_HEAD"))) c->fuse_head = atoi(env) != 0;   /* 0: the first batch's last launch is a tracker launch again */
        if ((env = getenv("GSDF_NRM_SPLIT"))) {            /* experiments: "a" or "a,b" = per cent of the normals tiles in launch 0 (and 1) */
            int a = -1, b = -1;
            const int n = sscanf(env, "%d,%d", &a, &b);
            if (n >= 1 && a >= 0 && a <= 100) { c->nrm_split = a; c->nrm_split2 = 100 - a; }
            if (n >= 2 && b >= 0 && a + b <= 100) c->nrm_split2 = b;
        }
        if ((env = getenv("GSDF_NEXT_BATCH")) && atoi(env) >= 1) c->next_batch = atoi(env);
        if ((env = getenv("GSDF_LAZY_FUSE"))) c->lazy_fuse = atoi(env);
        if ((env = getenv("GSDF_PERSIST"))) c->persist = atoi(env);
        if ((env = getenv("GSDF_FAR_TABLE"))) c->far_table = atoi(env);       /* experiments: 0 / 1 pin the fusion kernel's table size */
        /* auto-grow: entries the host may be ahead of the newest count.  Counts are queued every fourth entry while the load
         * is low, so a lag below 8 could never be "safe" between two of them: smaller values are raised to 8 */
        if ((env = getenv("GSDF_GROW_MAX_LAG")) && atoi(env) >= 1) c->grow_max_lag = std::max(8, atoi(env));
    }
    int rc = gsdf_reset(c);
    if (rc != GSDF_OK) { gsdf_destroy(c); return rc; }
    *out = c;
    return GSDF_OK;
}

void gsdf_destroy(gsdf_ctx* c) {
    if (!c) return;
    (void)gsdf_flush_pending(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->trace) { (void)hipFree(c->trace); c->trace = nullptr; }      /* after the sync: a running kernel may still write stamps */
    prof_collect(c);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    for (auto& u : c->uploads) { (void)hipEventSynchronize(u.second); (void)hipEventDestroy(u.second); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (hipEvent_t e : c->mark_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->upload_pool) (void)hipEventDestroy(e);
    for (auto& m : c->marks) (void)hipEventDestroy(m.second);
    void* ptrs[] = { c->depth_sampled, c->tile_stats, c->grow_scratch, c->scratch, c->track_rows, c->track_abort, c->rc_counts, c->tab.vox, c->tab.bkeys, c->tab.occ, c->st, c->counter, c->planes, c->depth_stage, c->normals, c->partials,
                     c->blk_counters, c->frame_log, c->deferred, c->deferred_count, c->fuse_ticket, c->tile_flags, c->tile_order, c->vis, c->ba_images, c->ba_Rt,
                     c->ba_frame_idx, c->ba_block_E, c->ba_block_part, c->ba_Hb, c->ba_gate_list, c->ba_gate_tmp, c->counter2, c->ba_mean };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& b : c->mx) if (b.p) (void)hipFree(b.p);
    if (c->progress) (void)hipHostFree((void*)c->progress);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int gsdf_reset(gsdf_ctx* c) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    c->pending.valid = false;                              /* a fusion that was never launched is dropped with the map */
    c->hint_next = nullptr; c->nrm_ready_depth = nullptr;  /* (gsdf_hint_next_depth_dev: a new scan starts without them) */
    c->prev_track_serial = 0; c->prev_slow = false;
    if (c->deferred_count) HIP_TRY(hipMemsetAsync(c->deferred_count, 0, sizeof(unsigned int), c->stream));
    gsdf_launch_table_clear(c->stream, c->tab, c->n_slots);
    if (c->vis) HIP_TRY(hipMemsetAsync(c->vis, 0, c->n_slots * (size_t)c->vis_words * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(c->st, 0, sizeof(gsdf_dev_state), c->stream));
    if (c->trace) {                                          /* test build: the memset cleared the trace pointer kept in dbg[23] */
        const unsigned long long ptr = (unsigned long long)(uintptr_t)c->trace;
        HIP_TRY(hipMemcpyAsync(&c->st->dbg[23], &ptr, 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));            /* `ptr` is a stack variable */
    }
    const float ident[7] = { 0, 0, 0, 0, 0, 0, 1 };          /* pose_ = SE3() -- RigidOptimizer.h:64 */
    gsdf_launch_set_pose(c->stream, c->st, nullptr, ident);
    if (c->blk_counters)
        HIP_TRY(hipMemsetAsync(c->blk_counters, 0, (size_t)c->fuse_blocks * 4 * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->merged = false;
    c->grow_forget = true;                                   /* auto-grow's block counts describe the old map (a grown table keeps its size) */
    c->occ_dirty = false;                                    /* the table clear zeroed the filters as well */
    return GSDF_OK;
}

int gsdf_set_auto_grow(gsdf_ctx* c, int max_capacity_log2) {
    GSDF_FLUSH(c);
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    if (max_capacity_log2 != 0 && (max_capacity_log2 < c->capacity_log2 || max_capacity_log2 > 30))
        return fail(GSDF_ERR_INVALID, "gsdf_set_auto_grow: 0 (off) or a capacity_log2 between the present one and 30");
    HIP_TRY(hipSetDevice(c->device));
    if (max_capacity_log2 && !c->grow_scratch) {
        HIP_TRY(hipMalloc((void**)&c->grow_scratch, 2 * sizeof(unsigned int)));
        HIP_TRY(hipMemsetAsync(c->grow_scratch, 0, 2 * sizeof(unsigned int), c->stream));
    }
    if (max_capacity_log2 && !c->progress) return fail(GSDF_ERR_HIP, "gsdf_set_auto_grow: no pinned progress words on this context");
    c->auto_grow_max = max_capacity_log2;
    c->grow_forget = true;
    return GSDF_OK;
}

int gsdf_capacity(gsdf_ctx* c, int* capacity_log2) {
    if (!c || !capacity_log2) return fail(GSDF_ERR_INVALID, "null argument");
    *capacity_log2 = c->capacity_log2;
    return GSDF_OK;
}

int gsdf_set_zrange(gsdf_ctx* c, float zmin, float zmax) {
    GSDF_FLUSH(c);
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    c->zmin = zmin; c->zmax = zmax;
    return GSDF_OK;
}

int gsdf_normals_init(gsdf_ctx* c, int W, int H, const float K[9], int win) {
    GSDF_FLUSH(c);
    if (!c || !K) return fail(GSDF_ERR_INVALID, "null argument");
    if (W <= 0 || H <= 0 || win <= 0 || !(win & 1) || win / 2 > 7)
        return fail(GSDF_ERR_INVALID, "W,H > 0 and odd window <= 15 required");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    void* old[] = { c->depth_sampled, c->tile_stats, c->planes, c->depth_stage, c->normals, c->partials, c->blk_counters, c->frame_log, c->deferred,
                    c->deferred_count, c->tile_flags, c->tile_order, c->fuse_ticket, c->track_rows, c->track_abort };
    for (void* p : old) if (p) (void)hipFree(p);
    c->track_rows = nullptr; c->track_abort = nullptr;
    c->tile_flags = nullptr; c->tile_order = nullptr; c->tile_stats = nullptr;
    c->planes = c->depth_stage = c->normals = nullptr; c->partials = nullptr; c->depth_sampled = nullptr;
    c->hint_next = nullptr; c->nrm_ready_depth = nullptr;
    c->blk_counters = nullptr; c->frame_log = nullptr; c->deferred = nullptr; c->deferred_count = nullptr; c->fuse_ticket = nullptr;
    c->W = W; c->H = H; c->win = win;
    std::memcpy(c->K, K, 9 * sizeof(float));
    const size_t N = (size_t)W * H;
    HIP_TRY(hipMalloc((void**)&c->planes, 11 * N * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c->depth_stage, N * sizeof(float)));
    /* three sets of 3 planes: sets 0 / 1 alternate between GT-pose fusions (the normals of frame i + 1 are computed on a second
     * stream while frame i is being fused), set 2 belongs to the main stream (tracked frames, gsdf_normals_compute) */
    HIP_TRY(hipMalloc((void**)&c->normals, 3 * 3 * N * sizeof(float)));
    /* tracker grid: a multiple of the 256 CUs when the frame is large enough, 1-4 pixels per lane */
    c->track_blocks = N >= (size_t)1 << 20 ? 2 * GSDF_TRACK_MAXBLK : N >= (size_t)1 << 18 ? GSDF_TRACK_MAXBLK
                                                : (int)std::max<size_t>(1, (N + 511) / 512);
    HIP_TRY(hipMalloc((void**)&c->partials, (size_t)3 * GSDF_TRACK_ROWSET * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->partials, 0, (size_t)3 * GSDF_TRACK_ROWSET * sizeof(double), c->stream));
    c->track_rot = 0;
    if (c->track_blocks <= 2 * GSDF_TRACK_MAXBLK) {
        HIP_TRY(hipMalloc(&c->track_rows, gsdf_track_all_rows_bytes(c->track_blocks)));
        HIP_TRY(hipMemsetAsync(c->track_rows, 0, gsdf_track_all_rows_bytes(c->track_blocks), c->stream));
        HIP_TRY(hipMalloc((void**)&c->track_abort, sizeof(unsigned int)));
        HIP_TRY(hipMemsetAsync(c->track_abort, 0, sizeof(unsigned int), c->stream));
    }
    c->fuse_blocks = gsdf_fuse_grid_blocks(W, H);
    /* per set of normal planes: the statistics of the frame's fusion tiles (depth range, valid pixels), written with the normals */
    HIP_TRY(hipMalloc((void**)&c->tile_stats, (size_t)3 * c->fuse_blocks * 4 * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(c->tile_stats, 0, (size_t)3 * c->fuse_blocks * 4 * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMalloc((void**)&c->blk_counters, (size_t)c->fuse_blocks * 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->blk_counters, 0, (size_t)c->fuse_blocks * 4 * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipMalloc((void**)&c->tile_flags, (size_t)c->fuse_blocks * sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(c->tile_flags, 0, (size_t)c->fuse_blocks * sizeof(unsigned int), c->stream));
    {
        std::vector<uint32_t> order((size_t)c->fuse_blocks);
        gsdf_fuse_tile_order(W, H, order.data());
        HIP_TRY(hipMalloc((void**)&c->tile_order, order.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(c->tile_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    /* deferred list: contributions of near tiles and LDS overflow; bounded by the samples of a frame */
    c->deferred_cap = (unsigned int)std::min<size_t>((size_t)1 << 26, std::max<size_t>((size_t)1 << 18,
                                                     N * (size_t)(2 * c->factor + 1)));   /* every sample of a frame */
    HIP_TRY(hipMalloc((void**)&c->deferred, (size_t)c->deferred_cap * sizeof(gsdf_deferred)));
    HIP_TRY(hipMalloc((void**)&c->deferred_count, sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(c->deferred_count, 0, sizeof(unsigned int), c->stream));
    HIP_TRY(hipMalloc((void**)&c->fuse_ticket, 2 * sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(c->fuse_ticket, 0, 2 * sizeof(unsigned int), c->stream));
    c->frame_log_cap = 1 << 16;
    HIP_TRY(hipMalloc((void**)&c->frame_log, (size_t)c->frame_log_cap * 10 * sizeof(float)));
    {
        double* scratch = nullptr;                           /* the row sums of the six moment planes, in double */
        HIP_TRY(hipMalloc((void**)&scratch, gsdf_normals_cache_scratch_bytes(W, H)));
        gsdf_launch_normals_cache(c->stream, W, H, c->K, win, c->planes, scratch);
        const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(c->stream);
        (void)hipFree(scratch);
        HIP_TRY(e1);
        HIP_TRY(e2);
    }
    return GSDF_OK;
}

int gsdf_normals_cache(gsdf_ctx* c, float* planes11_host) {
    GSDF_FLUSH(c);
    int rc = require_frame(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(planes11_host, c->planes, 11 * (size_t)c->W * c->H * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSDF_OK;
}

int gsdf_normals_compute(gsdf_ctx* c, const float* depth_host, float* nx, float* ny, float* nz) {
    GSDF_FLUSH(c);
    int rc = require_frame(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const size_t N = (size_t)c->W * c->H;
    HIP_TRY(hipMemcpyAsync(c->depth_stage, depth_host, N * sizeof(float), hipMemcpyHostToDevice, c->stream));
    float* set2 = c->normals + 6 * N;
    gsdf_launch_normals(c->stream, c->geom(), c->win, c->ncache(), c->depth_stage, set2, set2 + N, set2 + 2 * N, nullptr, nullptr, stats_of(c, set2));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(nx, set2, N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(ny, set2 + N, N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(nz, set2 + 2 * N, N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSDF_OK;
}

int gsdf_update_dev(gsdf_ctx* c, const float* depth_dev, const float R[9], const float t[3]) {
    int rc = require_frame(c);
    if (rc) return rc;
    if (!depth_dev || !R || !t) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = auto_grow_step(c))) return rc;
    gsdf_pose_arg pose;
    std::memcpy(pose.R, R, sizeof(pose.R));
    std::memcpy(pose.t, t, sizeof(pose.t));
    if (!c->defer || c->profiling) {                       /* event-timed replays: normals and fusion as two launches, in order */
        int rc = gsdf_flush_pending(c);
        if (rc) return rc;
        return enqueue_fuse(c, depth_dev, pose, 0, false);
    }
    /* MapGradPixelSdf.cpp:60: the normals of a frame depend on its depth only.  A run of GT-pose fusions (the branch
     * main_scan_3d.cpp:250-254; frame-sharded fusion) is pipelined by ONE call: the k_fuse of frame i is launched when frame
     * i + 1 arrives, and that launch's last workgroups compute the normals of frame i + 1 -- in the tail of the fusion,
     * where two thirds of the CUs idle, instead of as a launch of their own in front of the next fusion (10 us + a gap on
     * the critical path of every frame before).  Anything else the caller does with the context launches the waiting
     * fusion first (gsdf_flush_pending at the top of every other entry), so the deferral is invisible except in time.
     * The two sets of normal planes alternate. */
    const size_t N = (size_t)c->W * c->H;
    const int set = c->nrm_parity;
    c->nrm_parity ^= 1;
    c->nrm_ready_depth = nullptr;                          /* sets 0 / 1 are this path's: a hinted tracked frame's normals are gone */
    float* nrm = c->normals + (size_t)set * 3 * N;
    if (!c->pending.valid) {
        gsdf_launch_normals(c->stream, c->geom(), c->win, c->ncache(), depth_dev, nrm, nrm + N, nrm + 2 * N, nullptr, nullptr, stats_of(c, nrm));
        HIP_TRY(hipGetLastError());
    } else {
        const float* pn = c->normals + (size_t)c->pending.set * 3 * N;
        int rc = launch_fuse(c, c->pending.depth, pn, c->pending.pose, 0, depth_dev, nrm);
        c->pending.valid = false;
        if (rc) return rc;
    }
    c->pending.valid = true; c->pending.depth = depth_dev; c->pending.pose = pose; c->pending.set = set;
    return GSDF_OK;
}

int gsdf_update(gsdf_ctx* c, const float* depth_host, const float R[9], const float t[3]) {
    GSDF_FLUSH(c);
    int rc = require_frame(c);
    if (rc) return rc;
    if (!depth_host) return fail(GSDF_ERR_INVALID, "null depth");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->depth_stage, depth_host, (size_t)c->W * c->H * sizeof(float), hipMemcpyHostToDevice, c->stream));
    rc = gsdf_update_dev(c, c->depth_stage, R, t);
    if (rc) return rc;
    return gsdf_sync(c);
}

int gsdf_set_pose(gsdf_ctx* c, const float pose7[7]) {
    GSDF_FLUSH(c);
    if (!c || !pose7) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    gsdf_launch_set_pose(c->stream, c->st, nullptr, pose7);
    HIP_TRY(hipGetLastError());
    return GSDF_OK;
}

int gsdf_get_pose(gsdf_ctx* c, float pose7[7]) {
    GSDF_FLUSH(c);
    if (!c || !pose7) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    gsdf_dev_state s;
    int rc = read_state(c, &s);
    if (rc) return rc;
    std::memcpy(pose7, s.pose7, 7 * sizeof(float));
    return GSDF_OK;
}

int gsdf_track(gsdf_ctx* c, const float* depth_host, const float K[9], float pose7[7], int num_iterations,
               float conv_threshold, float damping, int* converged, int* passes) {
    return gsdf_track_sampled(c, depth_host, K, pose7, num_iterations, conv_threshold, damping, 1, converged, passes);
}

int gsdf_track_sampled(gsdf_ctx* c, const float* depth_host, const float K[9], float pose7[7], int num_iterations,
                       float conv_threshold, float damping, int sampling, int* converged, int* passes) {
    GSDF_FLUSH(c);
    int rc = require_frame(c);
    if (rc) return rc;
    if (!depth_host || !K || !pose7) return fail(GSDF_ERR_INVALID, "null argument");
    /* size_t sampling of the reference: 0 never ends its loops (y += 0); a stride beyond the image is the single pixel (0, 0) */
    if (sampling < 1) return fail(GSDF_ERR_INVALID, "sampling must be >= 1");
    if (sampling > std::max(c->W, c->H)) sampling = std::max(c->W, c->H);
    if (std::memcmp(K, c->K, 9 * sizeof(float)) != 0)
        return fail(GSDF_ERR_INVALID, "K differs from the intrinsics given to gsdf_normals_init");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->depth_stage, depth_host, (size_t)c->W * c->H * sizeof(float), hipMemcpyHostToDevice, c->stream));
    gsdf_launch_set_pose(c->stream, c->st, nullptr, pose7);
    rc = enqueue_track(c, c->depth_stage, num_iterations, conv_threshold, damping, false, sampling);
    if (rc) return rc;
    gsdf_dev_state s;
    rc = read_state(c, &s);
    if (rc) return rc;
    std::memcpy(pose7, s.pose7, 7 * sizeof(float));
    if (converged) *converged = s.converged;
    if (passes) *passes = s.passes;
    return status_to_code(s.status);
}

int gsdf_hint_next_depth_dev(gsdf_ctx* c, const float* next_depth_dev) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    c->hint_next = next_depth_dev;                       /* nullptr withdraws it; no device work, no flush */
    return GSDF_OK;
}

int gsdf_track_and_fuse_ahead_dev(gsdf_ctx* c, const float* depth_dev, const float* next_depth_dev, const float K[9],
                                  int num_iterations, float conv_threshold, float damping) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    c->hint_next = next_depth_dev;
    return gsdf_track_and_fuse_dev(c, depth_dev, K, num_iterations, conv_threshold, damping);
}

int gsdf_track_and_fuse_dev(gsdf_ctx* c, const float* depth_dev, const float K[9], int num_iterations,
                            float conv_threshold, float damping) {
    GSDF_FLUSH(c);
    int rc = require_frame(c);
    if (rc) return rc;
    if (!depth_dev || !K) return fail(GSDF_ERR_INVALID, "null argument");
    if (std::memcmp(K, c->K, 9 * sizeof(float)) != 0)
        return fail(GSDF_ERR_INVALID, "K differs from the intrinsics given to gsdf_normals_init");
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = auto_grow_step(c))) return rc;
    rc = enqueue_track(c, depth_dev, num_iterations, conv_threshold, damping, true);   /* main_scan_3d.cpp:258-265 */
    if (rc) return rc;
    HIP_TRY(hipGetLastError());                  /* the frame's log row is written on the device: by the last workgroup of the
                                                    k_fuse that fused it, or by the gated k_fuse that found it not converged */
    return GSDF_OK;
}

int gsdf_read_frame_log(gsdf_ctx* c, float* rows10, int64_t max_rows, int64_t* n_rows) {
    GSDF_FLUSH(c);
    int rc = require_frame(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    gsdf_dev_state s;
    rc = read_state(c, &s);
    if (rc) return rc;
    int64_t n = std::min<int64_t>(s.log_rows, c->frame_log_cap);
    if (n_rows) *n_rows = n;
    n = std::min<int64_t>(n, max_rows);
    if (n > 0 && rows10) {
        HIP_TRY(hipMemcpyAsync(rows10, c->frame_log, (size_t)n * 10 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return GSDF_OK;
}

int gsdf_sync(gsdf_ctx* c) {
    GSDF_FLUSH(c);
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    gsdf_dev_state s;
    int rc = read_state(c, &s);
    if (rc) return rc;
    prof_collect(c);
    return status_to_code(s.status);
}

int gsdf_get_stats(gsdf_ctx* c, gsdf_stats* out) {
    GSDF_FLUSH(c);
    if (!c || !out) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    gsdf_dev_state s;
    int rc = read_state(c, &s);
    if (rc) return rc;
    std::memset(out, 0, sizeof(*out));
    if (c->blk_counters) {
        std::vector<unsigned long long> h((size_t)c->fuse_blocks * 4);
        HIP_TRY(hipMemcpy(h.data(), c->blk_counters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long cu = 0, cv = 0;
        for (int b = 0; b < c->fuse_blocks; ++b) { cu += h[4 * b + 2]; cv += h[4 * b + 3]; }
        out->n_upd = (int64_t)cu;
        out->n_valid = (int64_t)cv;
    }
    out->n_hit = (int64_t)s.n_hit;
    out->track_passes = s.passes;
    out->converged = s.converged;
    out->frames = s.frames;
    out->n_deferred = (int64_t)s.n_deferred;
    out->fuse_timeouts = (int64_t)s.fuse_timeouts;
    return GSDF_OK;
}

int gsdf_count(gsdf_ctx* c, int64_t* n) {
    GSDF_FLUSH(c);
    if (!c || !n) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
    gsdf_launch_export(c->stream, c->tab, c->n_slots, nullptr, nullptr, c->counter, 0, 0, nullptr, 0, nullptr);
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, c->counter, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *n = (int64_t)h;
    return GSDF_OK;
}

static int export_impl(gsdf_ctx* c, int32_t* keys, float* payload, uint32_t* vis_words_out, int64_t max_n, int64_t* n_out,
                       int sorted, int raw_sums);

int gsdf_export(gsdf_ctx* c, int32_t* keys, float* payload, int64_t max_n, int64_t* n_out, int sorted, int raw_sums) {
    GSDF_FLUSH(c);
    return export_impl(c, keys, payload, nullptr, max_n, n_out, sorted, raw_sums);
}

int gsdf_enable_vis(gsdf_ctx* c, int max_frames) {
    GSDF_FLUSH(c);
    if (!c || max_frames <= 0) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->vis) { (void)hipFree(c->vis); c->vis = nullptr; }
    c->vis_words = (max_frames + 31) / 32;
    const size_t bytes = c->n_slots * (size_t)c->vis_words * sizeof(uint32_t);
    HIP_TRY(hipMalloc((void**)&c->vis, bytes));
    HIP_TRY(hipMemsetAsync(c->vis, 0, bytes, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSDF_OK;
}

int gsdf_export_vis(gsdf_ctx* c, int32_t* keys, uint32_t* words, int words_per_voxel, int64_t max_n, int64_t* n_out) {
    GSDF_FLUSH(c);
    if (!c || !words) return fail(GSDF_ERR_INVALID, "null argument");
    if (!c->vis) return fail(GSDF_ERR_INVALID, "gsdf_enable_vis was not called");
    if (words_per_voxel != c->vis_words) return fail(GSDF_ERR_INVALID, "words_per_voxel differs from gsdf_enable_vis");
    return export_impl(c, keys, nullptr, words, max_n, n_out, 1, 0);
}

static int export_impl(gsdf_ctx* c, int32_t* keys, float* payload, uint32_t* vis_words_out, int64_t max_n, int64_t* n_out,
                       int sorted, int raw_sums) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    int64_t n = 0;
    int rc = gsdf_count(c, &n);
    if (rc) return rc;
    if (n_out) *n_out = n;
    if (n == 0 || max_n <= 0 || (!keys && !payload && !vis_words_out)) return GSDF_OK;
    if (max_n < n) return fail(GSDF_ERR_INVALID, "export buffer too small (call gsdf_count first)");
    unsigned long long* dkeys = nullptr;
    float* dpay = nullptr;
    HIP_TRY(hipMalloc((void**)&dkeys, (size_t)n * sizeof(unsigned long long)));
    hipError_t e = hipMalloc((void**)&dpay, (size_t)n * 5 * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(dkeys); return fail(GSDF_ERR_HIP, hipGetErrorString(e)); }
    uint32_t* dvis = nullptr;
    const int vw = vis_words_out ? c->vis_words : 0;
    if (vw) {
        e = hipMalloc((void**)&dvis, (size_t)n * vw * sizeof(uint32_t));
        if (e != hipSuccess) { (void)hipFree(dkeys); (void)hipFree(dpay); return fail(GSDF_ERR_HIP, hipGetErrorString(e)); }
    }
    (void)hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream);
    gsdf_launch_export(c->stream, c->tab, c->n_slots, dkeys, dpay, c->counter, n, raw_sums, c->vis, vw, dvis);
    std::vector<unsigned long long> hk((size_t)n);
    std::vector<float> hp((size_t)n * 5);
    std::vector<uint32_t> hv((size_t)n * vw);
    e = hipMemcpyAsync(hk.data(), dkeys, hk.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hp.data(), dpay, hp.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && vw) e = hipMemcpyAsync(hv.data(), dvis, hv.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dkeys);
    (void)hipFree(dpay);
    if (dvis) (void)hipFree(dvis);
    if (e != hipSuccess) return fail(GSDF_ERR_HIP, hipGetErrorString(e));
    std::vector<size_t> order((size_t)n);
    std::iota(order.begin(), order.end(), (size_t)0);
    if (sorted)   /* packed keys order like (z, y, x) */
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return hk[a] < hk[b]; });
    for (int64_t i = 0; i < n; ++i) {
        const size_t j = order[(size_t)i];
        if (keys) {
            int x, y, z;
            gsdf_key_unpack(hk[j], &x, &y, &z);
            keys[3 * i] = x; keys[3 * i + 1] = y; keys[3 * i + 2] = z;
        }
        if (payload) std::memcpy(payload + 5 * i, hp.data() + 5 * j, 5 * sizeof(float));
        if (vis_words_out) std::memcpy(vis_words_out + (size_t)vw * i, hv.data() + (size_t)vw * j, vw * sizeof(uint32_t));
    }
    return GSDF_OK;
}

int gsdf_merge_raw(gsdf_ctx* c, const int32_t* keys, const float* payload_raw, int64_t n) {
    GSDF_FLUSH(c);
    if (!c || (n > 0 && (!keys || !payload_raw))) return fail(GSDF_ERR_INVALID, "null argument");
    if (n <= 0) return GSDF_OK;
    HIP_TRY(hipSetDevice(c->device));
    int32_t* dk = nullptr;
    float* dp = nullptr;
    HIP_TRY(hipMalloc((void**)&dk, (size_t)n * 3 * sizeof(int32_t)));
    hipError_t e = hipMalloc((void**)&dp, (size_t)n * 5 * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(dk); return fail(GSDF_ERR_HIP, hipGetErrorString(e)); }
    e = hipMemcpyAsync(dk, keys, (size_t)n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dp, payload_raw, (size_t)n * 5 * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        c->occ_dirty = true;
        gsdf_launch_merge_raw(c->stream, c->tab, dk, dp, n, c->st);
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(dk);
    (void)hipFree(dp);
    if (e != hipSuccess) return fail(GSDF_ERR_HIP, hipGetErrorString(e));
    return gsdf_sync(c);
}

int gsdf_export_raw_dev(gsdf_ctx* c, int32_t* keys_dev, float* payload_dev, int64_t max_n, int64_t* n) {
    GSDF_FLUSH(c);
    if (!c || !keys_dev || !payload_dev || max_n < 0) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
    gsdf_launch_export_raw(c->stream, c->tab, c->n_slots, keys_dev, payload_dev, c->counter, max_n);
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, c->counter, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n) *n = (int64_t)h;
    if ((int64_t)h > max_n) return fail(GSDF_ERR_INVALID, "export buffer too small (call gsdf_count first)");
    return GSDF_OK;
}

int gsdf_merge_raw_dev(gsdf_ctx* c, const int32_t* keys_dev, const float* payload_raw_dev, int64_t n) {
    GSDF_FLUSH(c);
    if (!c || (n > 0 && (!keys_dev || !payload_raw_dev))) return fail(GSDF_ERR_INVALID, "null argument");
    if (n <= 0) return GSDF_OK;
    HIP_TRY(hipSetDevice(c->device));
    c->occ_dirty = true;
    gsdf_launch_merge_raw(c->stream, c->tab, keys_dev, payload_raw_dev, n, c->st);
    HIP_TRY(hipGetLastError());
    return gsdf_sync(c);
}

/* ---- PhotoBA: PhotometricOptimizer (ps_optimizer/PhotometricOptimizer.cpp) ------------------------------- */
static gsdf_ba_dev ba_dev(gsdf_ctx* c) {
    gsdf_ba_dev d;
    d.tab = c->tab; d.n_slots = c->n_slots; d.vis = c->vis; d.vis_words = c->vis_words;
    d.n = c->ba_n; d.W = c->W; d.H = c->H;
    d.images = c->ba_images; d.R = c->ba_Rt; d.t = c->ba_Rt + 9 * (size_t)c->ba_n; d.frame_idx = c->ba_frame_idx;
    d.fx = c->K[0]; d.fy = c->K[4]; d.cx = c->K[2]; d.cy = c->K[5]; d.vs = c->voxel_size; d.reg_weight = c->ba_reg;
    d.trunc_sq = c->ba_trunc_sq;
    d.gate_list = c->ba_gate_fresh ? c->ba_gate_list : nullptr;
    d.gate_count = c->counter2;
    d.mean_cache = c->ba_mean;
    return d;
}
/* (re)builds the list of the voxels inside the |dist| <= vs gate, if the distances may have changed since it was made.  Enqueue
 * only; failures leave the sweeps on their whole-table form (gate_list == nullptr). */
static void ba_refresh_gate(gsdf_ctx* c) {
    if (c->ba_gate_fresh || !c->ba_gate_list || !c->counter2) return;
    gsdf_ba_dev d = ba_dev(c);
    size_t bytes = c->ba_gate_tmp_bytes;
    if (gsdf_ba_compact(c->stream, d, c->ba_gate_list, c->counter2, c->ba_gate_tmp, &bytes) == hipSuccess) c->ba_gate_fresh = true;
    else (void)hipGetLastError();
}
static int ba_upload_poses(gsdf_ctx* c) {
    HIP_TRY(hipMemcpyAsync(c->ba_Rt, c->ba_R.data(), c->ba_R.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ba_Rt + 9 * (size_t)c->ba_n, c->ba_t.data(), c->ba_t.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSDF_OK;
}
static int ba_require(gsdf_ctx* c) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    if (!c->ba_n) return fail(GSDF_ERR_INVALID, "gsdf_ba_setup was not called");
    return GSDF_OK;
}

int gsdf_ba_set_loss(gsdf_ctx* c, int loss, float lambda) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    if (loss < 0 || loss > 4 || !(lambda >= 0.f)) return fail(GSDF_ERR_INVALID, "loss must be a LossFunction value (0..4), lambda >= 0");
    c->ba_trunc_sq = loss == 4 ? lambda * lambda : -1.f;          /* the reference's code only distinguishes TRUNC_L2 */
    return GSDF_OK;
}

int gsdf_ba_setup(gsdf_ctx* c, int n, const float* images_bgr_host, const float* poses16_host, const int* frame_idx,
                  float reg_weight) {
    GSDF_FLUSH(c);
    int rc = require_frame(c);
    if (rc) return rc;
    if (!c->vis) return fail(GSDF_ERR_INVALID, "PhotoBA needs the vis_ bit-vectors: call gsdf_enable_vis before fusing");
    if (n <= 0 || n > 64 || !images_bgr_host || !poses16_host || !frame_idx) return fail(GSDF_ERR_INVALID, "bad argument (1..64 keyframes)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    void* old[] = { c->ba_images, c->ba_Rt, c->ba_frame_idx, c->ba_block_E, c->ba_block_part, c->ba_Hb, c->ba_gate_list, c->ba_gate_tmp, c->counter2, c->ba_mean };
    for (void* p : old) if (p) (void)hipFree(p);
    c->ba_gate_list = nullptr; c->ba_gate_tmp = nullptr; c->counter2 = nullptr; c->ba_gate_fresh = false;
    c->ba_mean = nullptr; c->ba_mean_valid = false;
    c->ba_images = nullptr; c->ba_Rt = nullptr; c->ba_frame_idx = nullptr; c->ba_block_E = nullptr; c->ba_block_part = nullptr; c->ba_Hb = nullptr;
    c->ba_n = n; c->ba_reg = reg_weight;
    const size_t img_bytes = (size_t)n * c->W * c->H * 3 * sizeof(float);
    HIP_TRY(hipMalloc((void**)&c->ba_images, img_bytes));
    HIP_TRY(hipMalloc((void**)&c->ba_Rt, (size_t)n * 12 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c->ba_frame_idx, (size_t)n * sizeof(int)));
    /* two sets of 3 x blocks doubles (energy, voxels, observations per workgroup): gsdf_ba_optimize keeps the sweep behind the
     * pose step and the one behind the distance step apart and reads both with one synchronisation */
    HIP_TRY(hipMalloc((void**)&c->ba_block_E, (size_t)2 * 3 * gsdf_ba_blocks() * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&c->ba_block_part, (size_t)gsdf_ba_blocks() * n * 27 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c->ba_Hb, (size_t)n * 27 * sizeof(float)));
    {   /* the gate list of the energy / pose sweeps: optional (without it they sweep the whole table) */
        gsdf_ba_dev d = ba_dev(c);
        size_t bytes = 0;
        if (gsdf_ba_compact(c->stream, d, nullptr, nullptr, nullptr, &bytes) == hipSuccess &&
            hipMalloc((void**)&c->ba_gate_list, c->n_slots * sizeof(uint32_t)) == hipSuccess &&
            hipMalloc(&c->ba_gate_tmp, bytes ? bytes : 8) == hipSuccess && hipMalloc((void**)&c->counter2, sizeof(unsigned long long)) == hipSuccess) {
            c->ba_gate_tmp_bytes = bytes;
            /* optional on top: 24 B per possible list entry for what the energy sweep hands to the pose sweep (100 MB at 2^22 records) */
            /* GSDF_BA_MEAN_CACHE (read here, so that a test can switch within one context): 0 the pose sweep computes its means
             * itself; 2 (tests only) the stand-alone gsdf_ba_solve_pose trusts that nothing changed since the last energy sweep */
            const char* env = getenv("GSDF_BA_MEAN_CACHE");
            c->ba_mean_on = env ? atoi(env) : 1;
            if (!c->ba_mean_on || hipMalloc(&c->ba_mean, c->n_slots * 24) != hipSuccess) { (void)hipGetLastError(); c->ba_mean = nullptr; }
        } else {
            (void)hipGetLastError();
            if (c->ba_gate_list) { (void)hipFree(c->ba_gate_list); c->ba_gate_list = nullptr; }
        }
        c->ba_gate_fresh = false;
    }
    HIP_TRY(hipMemcpyAsync(c->ba_images, images_bgr_host, img_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ba_frame_idx, frame_idx, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->ba_R.resize(9 * (size_t)n); c->ba_t.resize(3 * (size_t)n);
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) c->ba_R[9 * i + 3 * r + k] = poses16_host[16 * i + 4 * r + k];
            c->ba_t[3 * i + r] = poses16_host[16 * i + 4 * r + 3];
        }
    c->ba_mean_valid = false;
    return ba_upload_poses(c);
}

/* the energy sweep enqueued into set `which` of the per-workgroup sums; ba_energy_sum adds a set up in the fixed order */
static int ba_energy_enqueue(gsdf_ctx* c, int which, bool pose_sweep_follows = true) {
    ba_refresh_gate(c);
    const gsdf_ba_dev d = ba_dev(c);
    /* the sweep leaves every gated voxel's mean intensity / keyframe set behind for a pose sweep at this same state */
    const bool write_mean = pose_sweep_follows && d.gate_list && d.mean_cache && c->ba_trunc_sq < 0.f;
    gsdf_launch_ba_energy(c->stream, d, c->ba_block_E + (size_t)which * 3 * gsdf_ba_blocks(), write_mean);
    c->ba_mean_valid = write_mean;
    HIP_TRY(hipGetLastError());
    return GSDF_OK;
}
static float ba_energy_sum(gsdf_ctx* c, const double* h) {
    const size_t nb = (size_t)gsdf_ba_blocks();
    double s = 0.0, a = 0.0, o = 0.0;
    for (size_t i = 0; i < nb; ++i) { s += h[i]; a += h[nb + i]; o += h[2 * nb + i]; }
    c->ba_last_voxels = (long long)a; c->ba_last_obs = (long long)o;
    return (float)s;
}

int gsdf_ba_energy(gsdf_ctx* c, float* E) {
    GSDF_FLUSH(c);
    int rc = ba_require(c);
    if (rc) return rc;
    c->ba_gate_fresh = false;                                 /* the map may have changed since the last BA call */
    if (!E) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = ba_energy_enqueue(c, 0))) return rc;
    std::vector<double> h((size_t)3 * gsdf_ba_blocks());
    HIP_TRY(hipMemcpyAsync(h.data(), c->ba_block_E, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *E = ba_energy_sum(c, h.data());
    return GSDF_OK;
}

int gsdf_ba_counters(gsdf_ctx* c, int64_t* voxels, int64_t* observations) {
    if (!c || !voxels || !observations) return fail(GSDF_ERR_INVALID, "null argument");
    *voxels = c->ba_last_voxels; *observations = c->ba_last_obs;
    return GSDF_OK;
}

static int ba_dist_enqueue(gsdf_ctx* c, float damping, double* block_cnt = nullptr) {
    gsdf_launch_ba_dist(c->stream, ba_dev(c), damping, block_cnt);
    c->ba_gate_fresh = false;                                 /* the distances moved: voxels may have crossed the gate */
    c->ba_mean_valid = false;
    HIP_TRY(hipGetLastError());
    return GSDF_OK;
}
int gsdf_ba_solve_dist(gsdf_ctx* c, float damping) {
    GSDF_FLUSH(c);
    int rc = ba_require(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    /* the stand-alone entry also counts what the sweep visited (gsdf_ba_counters); gsdf_ba_optimize's sweeps do not */
    double* cnt = c->ba_block_E + (size_t)3 * gsdf_ba_blocks();
    if ((rc = ba_dist_enqueue(c, damping, cnt))) return rc;
    std::vector<double> h((size_t)2 * gsdf_ba_blocks());
    HIP_TRY(hipMemcpyAsync(h.data(), cnt, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    double a = 0.0, o = 0.0;
    for (size_t i = 0; i < (size_t)gsdf_ba_blocks(); ++i) { a += h[i]; o += h[(size_t)gsdf_ba_blocks() + i]; }
    c->ba_last_voxels = (long long)a; c->ba_last_obs = (long long)o;
    return GSDF_OK;
}

/* the pose sweep + the per-keyframe 6x6 solves; wait_upload: block until the new poses are on the device (the public entry)
 * or leave the copy queued in front of whatever the caller enqueues next (gsdf_ba_optimize) */
static int ba_solve_pose(gsdf_ctx* c, bool wait_upload) {
    const int n = c->ba_n;
    const bool list_was_fresh = c->ba_gate_fresh;
    ba_refresh_gate(c);
    /* (the cache belongs to the list the energy sweep walked: a list rebuilt since then has other entries) */
    gsdf_launch_ba_pose(c->stream, ba_dev(c), c->ba_block_part, c->ba_Hb, c->ba_mean_valid && list_was_fresh);
    c->ba_mean_valid = false;                                 /* the poses move */
    std::vector<float> hb((size_t)n * 27);
    HIP_TRY(hipMemcpyAsync(hb.data(), c->ba_Hb, hb.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                 /* the 6x6 systems are solved on the host (:577-589) */
    for (int i = 0; i < n; ++i) {                             /* per-keyframe 6x6 LDLT + pose update (:577-589) */
        const float* v = &hb[27 * (size_t)i];
        float H[36], dp[6];
        int q = 6;
        for (int a1 = 0; a1 < 6; ++a1) for (int a2 = a1; a2 < 6; ++a2) { H[6 * a1 + a2] = v[q]; H[6 * a2 + a1] = v[q]; ++q; }
        gsdf_ldlt_solve6(H, v, dp);
        bool nan = false;
        for (int k = 0; k < 6; ++k) nan = nan || std::isnan(dp[k]);
        if (nan) continue;
        for (int k = 0; k < 3; ++k) c->ba_t[3 * i + k] -= dp[k];
        float pose[7] = { 0, 0, 0, 0, 0, 0, 1 };
        const float xi[6] = { 0, 0, 0, -dp[3], -dp[4], -dp[5] };
        gsdf_se3_exp_mul(xi, pose);                           /* SO3::exp(-omega) */
        float Ex[9], Rn[9];
        gsdf_quat_to_R(pose + 3, Ex);
        const float* Ri = &c->ba_R[9 * (size_t)i];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) Rn[3 * r + k] = gsdf_sum3(Ri[3 * r] * Ex[k], Ri[3 * r + 1] * Ex[3 + k], Ri[3 * r + 2] * Ex[6 + k]);
        std::memcpy(&c->ba_R[9 * (size_t)i], Rn, sizeof(Rn));
    }
    if (wait_upload) return ba_upload_poses(c);
    /* c->ba_R / ba_t are members: they stay untouched until the caller's next synchronisation, which is behind these copies */
    HIP_TRY(hipMemcpyAsync(c->ba_Rt, c->ba_R.data(), c->ba_R.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ba_Rt + 9 * (size_t)c->ba_n, c->ba_t.data(), c->ba_t.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    return GSDF_OK;
}
int gsdf_ba_solve_pose(gsdf_ctx* c, float damping) {
    GSDF_FLUSH(c);
    (void)damping;                                            /* unused by the reference as well (:499) */
    int rc = ba_require(c);
    if (rc) return rc;
    if (c->ba_mean_on != 2) {
        c->ba_gate_fresh = false;                             /* the map may have changed since the last BA call */
        c->ba_mean_valid = false;
    }
    HIP_TRY(hipSetDevice(c->device));
    return ba_solve_pose(c, true);
}

/* optimize (:611-662).  Per iteration the host needs two things from the device: the 27 sums per keyframe of the pose sweep
 * (it solves the 6x6 systems) and, for the stop rule, the two energies.  So an iteration synchronises twice: behind the pose
 * sweep, and ONCE behind [energy -> distance sweep -> energy], both energies in one read (they used to be four waits). */
int gsdf_ba_optimize(gsdf_ctx* c, int max_it, float* energies, int* n_energies, int* converged) {
    GSDF_FLUSH(c);
    int rc = ba_require(c);
    if (rc) return rc;
    c->ba_gate_fresh = false;                                 /* the map may have changed since the last BA call */
    if (!energies || !n_energies || !converged || max_it < 0) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int ne = 0;
    float E = 0.f, E_pose = 0.f;
    *converged = 0;
    if ((rc = gsdf_ba_energy(c, &E))) return rc;
    energies[ne++] = E;
    const size_t set = (size_t)3 * gsdf_ba_blocks();
    std::vector<double> h(2 * set);
    for (int iter = 0; iter < max_it; ++iter) {               /* :621-657 */
        if ((rc = ba_solve_pose(c, false))) return rc;
        if ((rc = ba_energy_enqueue(c, 0, false))) return rc;  /* (the distance sweep follows: nobody reads its means) */
        if ((rc = ba_dist_enqueue(c, 1.0f))) return rc;
        if ((rc = ba_energy_enqueue(c, 1))) return rc;
        HIP_TRY(hipMemcpyAsync(h.data(), c->ba_block_E, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        E_pose = ba_energy_sum(c, h.data());
        E = ba_energy_sum(c, h.data() + set);
        energies[ne++] = E_pose;
        energies[ne++] = E;
        const float rel = std::fabs(E_pose - E) / E_pose;
        if (rel < 0.0005f) { *converged = 1; break; }
        if (E_pose < E) break;                                /* diverged */
    }
    *n_energies = ne;
    return GSDF_OK;
}

int gsdf_ba_get_poses(gsdf_ctx* c, float* poses16_host) {
    int rc = ba_require(c);
    if (rc) return rc;
    if (!poses16_host) return fail(GSDF_ERR_INVALID, "null argument");
    for (int i = 0; i < c->ba_n; ++i) {
        float* P = poses16_host + 16 * (size_t)i;
        for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) P[4 * r + k] = c->ba_R[9 * (size_t)i + 3 * r + k]; P[4 * r + 3] = c->ba_t[3 * (size_t)i + r]; }
        P[12] = P[13] = P[14] = 0.f; P[15] = 1.f;
    }
    return GSDF_OK;
}

int gsdf_query(gsdf_ctx* c, const float* pts_host, int64_t n, float* dist, float* grad, float* w) {
    GSDF_FLUSH(c);
    if (!c || (n > 0 && (!pts_host || !dist || !grad || !w))) return fail(GSDF_ERR_INVALID, "null argument");
    if (n <= 0) return GSDF_OK;
    HIP_TRY(hipSetDevice(c->device));
    float* d = nullptr;
    const bool small = (size_t)n * 8 * sizeof(float) <= GSDF_SCRATCH_BYTES;       /* single points / small batches: no allocation per call */
    if (small) {
        if (!c->scratch) HIP_TRY(hipMalloc(&c->scratch, GSDF_SCRATCH_BYTES));
        d = (float*)c->scratch;
    } else HIP_TRY(hipMalloc((void**)&d, (size_t)n * 8 * sizeof(float)));
    float *dp = d, *dd = d + 3 * n, *dg = d + 4 * n, *dw = d + 7 * n;
    hipError_t e = hipMemcpyAsync(dp, pts_host, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        gsdf_launch_query(c->stream, c->tab, c->voxel_size, c->voxel_size_inv, dp, n, dd, dg, dw);
        e = hipMemcpyAsync(dist, dd, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(grad, dg, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(w, dw, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (!small) (void)hipFree(d);
    if (e != hipSuccess) return fail(GSDF_ERR_HIP, hipGetErrorString(e));
    return GSDF_OK;
}

int gsdf_get_voxels(gsdf_ctx* c, const int32_t* keys_host, int64_t n, float* payload, int32_t* found) {
    GSDF_FLUSH(c);
    if (!c || (n > 0 && (!keys_host || !payload || !found))) return fail(GSDF_ERR_INVALID, "null argument");
    if (n <= 0) return GSDF_OK;
    HIP_TRY(hipSetDevice(c->device));
    char* d = nullptr;
    const bool small = (size_t)n * 36 <= GSDF_SCRATCH_BYTES;
    if (small) {
        if (!c->scratch) HIP_TRY(hipMalloc(&c->scratch, GSDF_SCRATCH_BYTES));
        d = (char*)c->scratch;
    } else HIP_TRY(hipMalloc((void**)&d, (size_t)n * (3 * sizeof(int32_t) + 5 * sizeof(float) + sizeof(int32_t))));
    int32_t* dk = (int32_t*)d;
    float* dp = (float*)(d + (size_t)n * 3 * sizeof(int32_t));
    int32_t* df = (int32_t*)(d + (size_t)n * (3 * sizeof(int32_t) + 5 * sizeof(float)));
    hipError_t e = hipMemcpyAsync(dk, keys_host, (size_t)n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        gsdf_launch_get_voxels(c->stream, c->tab, dk, n, dp, df);
        e = hipMemcpyAsync(payload, dp, (size_t)n * 5 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(found, df, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (!small) (void)hipFree(d);
    if (e != hipSuccess) return fail(GSDF_ERR_HIP, hipGetErrorString(e));
    return GSDF_OK;
}

static int raycast_enqueue(gsdf_ctx* c, const float K[9], const float R[9], const float t[3], int W, int H, float zmin, float zmax,
                           float* depth_dev, float* normals_dev) {
    gsdf_pose_arg pose;
    std::memcpy(pose.R, R, sizeof(pose.R));
    std::memcpy(pose.t, t, sizeof(pose.t));
    /* per-workgroup counter rows (samples, records): sized for the largest grid seen, kept until reset */
    const size_t n_wg = (size_t)((W + 15) / 16) * (size_t)((H + 15) / 16);
    if (n_wg > c->rc_rows) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->rc_counts) (void)hipFree(c->rc_counts);
        c->rc_counts = nullptr; c->rc_rows = 0;
        HIP_TRY(hipMalloc((void**)&c->rc_counts, n_wg * 8 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(c->rc_counts, 0, n_wg * 8 * sizeof(unsigned long long), c->stream));
        c->rc_rows = n_wg;
    }
    if (c->occ_dirty) {                                       /* the map changed since the filters were built */
        gsdf_launch_occ_rebuild(c->stream, c->tab);
        c->occ_dirty = false;
    }
    prof_scope ps(c, 3);
    gsdf_launch_raycast(c->stream, c->tab, c->voxel_size, c->voxel_size_inv, c->factor, W, H, K, pose, zmin, zmax, depth_dev,
                        normals_dev, c->rc_counts, c->debug & 0xFFFF);
    return GSDF_OK;
}

int gsdf_raycast_dev(gsdf_ctx* c, const float K[9], const float R[9], const float t[3], int W, int H, float zmin, float zmax,
                     float* depth_dev, float* normals_dev) {
    GSDF_FLUSH(c);
    if (!c || !K || !R || !t || !depth_dev) return fail(GSDF_ERR_INVALID, "null argument");
    if (W <= 0 || H <= 0 || !(zmax > zmin) || !(zmin > 0.f)) return fail(GSDF_ERR_INVALID, "W,H > 0 and 0 < zmin < zmax required");
    HIP_TRY(hipSetDevice(c->device));
    int rc = raycast_enqueue(c, K, R, t, W, H, zmin, zmax, depth_dev, normals_dev);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return GSDF_OK;
}

int gsdf_raycast(gsdf_ctx* c, const float K[9], const float R[9], const float t[3], int W, int H, float zmin, float zmax,
                 float* depth_out, float* normals_out) {
    GSDF_FLUSH(c);
    if (!c || !K || !R || !t || !depth_out) return fail(GSDF_ERR_INVALID, "null argument");
    if (W <= 0 || H <= 0 || !(zmax > zmin) || !(zmin > 0.f)) return fail(GSDF_ERR_INVALID, "W,H > 0 and 0 < zmin < zmax required");
    HIP_TRY(hipSetDevice(c->device));
    const size_t N = (size_t)W * H;
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 4 * N * sizeof(float)));
    if (int rc = raycast_enqueue(c, K, R, t, W, H, zmin, zmax, d, normals_out ? d + N : nullptr)) { (void)hipFree(d); return rc; }
    hipError_t e = hipMemcpyAsync(depth_out, d, N * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && normals_out)
        e = hipMemcpyAsync(normals_out, d + N, 3 * N * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(GSDF_ERR_HIP, hipGetErrorString(e));
    return GSDF_OK;
}

int gsdf_raycast_counters(gsdf_ctx* c, int64_t* samples, int64_t* records, int reset) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long tot[4] = { 0, 0, 0, 0 };
    if (c->rc_counts) {
        std::vector<unsigned long long> h(c->rc_rows * 8);
        HIP_TRY(hipMemcpyAsync(h.data(), c->rc_counts, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < c->rc_rows; ++i) for (int k = 0; k < 4; ++k) tot[k] += h[8 * i + k];
        if (reset) HIP_TRY(hipMemsetAsync(c->rc_counts, 0, h.size() * sizeof(unsigned long long), c->stream));
    }
    if (samples) *samples = (int64_t)tot[0];
    if (records) *records = (int64_t)tot[1];
    c->rc_iters[0] = (long long)tot[2]; c->rc_iters[1] = (long long)tot[3];
    return GSDF_OK;
}

int gsdf_extract_mesh(gsdf_ctx* c, float iso, const int8_t tri_table[256 * 16], float* triangles_out, int64_t max_tris,
                      int64_t* n_tris) {
    GSDF_FLUSH(c);
    if (!c || !n_tris || (max_tris > 0 && !triangles_out)) return fail(GSDF_ERR_INVALID, "null argument");
    if (!tri_table) tri_table = GSDF_MC_TRI_TABLE;            /* the reference's triTable (LayeredMarchingCubesNoColor.cpp:96-352) */
    HIP_TRY(hipSetDevice(c->device));
    *n_tris = 0;
    /* upper bound of the output: 5 triangles per cube, one cube per voxel; sized by the caller through max_tris */
    const long long cap = max_tris > 0 ? (long long)max_tris : 0;
    int* d_mn = nullptr; signed char* d_tab = nullptr; float* d_tris = nullptr; unsigned long long* d_keys = nullptr;
    hipError_t e = hipMalloc((void**)&d_mn, 3 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&d_tab, 256 * 16);
    if (e == hipSuccess && cap) e = hipMalloc((void**)&d_tris, (size_t)cap * 9 * sizeof(float));
    if (e == hipSuccess && cap) e = hipMalloc((void**)&d_keys, (size_t)cap * sizeof(unsigned long long));
    const int big[3] = { 2147483647, 2147483647, 2147483647 };
    if (e == hipSuccess) e = hipMemcpyAsync(d_mn, big, sizeof(big), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tab, tri_table, 256 * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream);
    unsigned long long n = 0;
    if (e == hipSuccess) {
        gsdf_launch_mesh(c->stream, c->tab, c->n_slots, c->voxel_size, iso, d_mn, d_tab, d_tris, d_keys, c->counter, cap);
        e = hipMemcpyAsync(&n, c->counter, sizeof(n), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    const size_t got = (size_t)std::min<unsigned long long>(n, (unsigned long long)cap);
    /* the reference's order (z-y-x sweep, triangles of a cube in table order) = ascending sort key: radix sort of (key, index)
     * on the device, the triangles gathered into that order on the device, ONE copy of the sorted list to the caller */
    unsigned long long* d_keys2 = nullptr; uint32_t *d_idx = nullptr, *d_idx2 = nullptr; float* d_sorted = nullptr; void* d_tmp = nullptr;
    if (e == hipSuccess && got && n <= (unsigned long long)cap) {
        size_t tmp_bytes = 0;
        e = hipMalloc((void**)&d_keys2, got * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc((void**)&d_idx, got * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d_idx2, got * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d_sorted, got * 9 * sizeof(float));
        if (e == hipSuccess) e = gsdf_sort_pairs_u64(nullptr, &tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, got, c->stream);
        if (e == hipSuccess) e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8);
        if (e == hipSuccess) {
            gsdf_launch_iota(c->stream, d_idx, got);
            e = gsdf_sort_pairs_u64(d_tmp, &tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, got, c->stream);
        }
        if (e == hipSuccess) {
            gsdf_launch_gather_tris(c->stream, d_tris, d_idx2, d_sorted, got);
            e = hipMemcpyAsync(triangles_out, d_sorted, got * 9 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d_mn); (void)hipFree(d_tab); (void)hipFree(d_tris); (void)hipFree(d_keys);
    (void)hipFree(d_keys2); (void)hipFree(d_idx); (void)hipFree(d_idx2); (void)hipFree(d_sorted); (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail(GSDF_ERR_HIP, hipGetErrorString(e));
    *n_tris = (int64_t)n;                                    /* total found, also when it exceeds max_tris */
    if (n > (unsigned long long)cap) return cap ? fail(GSDF_ERR_INVALID, "gsdf_extract_mesh: max_tris too small (n_tris holds the need)") : GSDF_OK;
    return GSDF_OK;
}

int gsdf_block_keys_dev(gsdf_ctx* c, uint64_t* keys_dev, int64_t max_n, int64_t* n) {
    GSDF_FLUSH(c);
    if (!c || !n || (max_n > 0 && !keys_dev)) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
    gsdf_launch_block_keys(c->stream, c->tab, c->n_slots / GSDF_BLOCK_VOX, (unsigned long long*)keys_dev, c->counter, max_n);
    unsigned long long cnt = 0;
    HIP_TRY(hipMemcpyAsync(&cnt, c->counter, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *n = (int64_t)cnt;
    return GSDF_OK;
}
int gsdf_pack_blocks_dev(gsdf_ctx* c, const uint64_t* block_keys_dev, int64_t n, float* dense_dev) {
    GSDF_FLUSH(c);
    if (!c || (n > 0 && (!block_keys_dev || !dense_dev))) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    gsdf_launch_pack_blocks(c->stream, c->tab, (const unsigned long long*)block_keys_dev, n, dense_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSDF_OK;
}
int gsdf_unpack_blocks_dev(gsdf_ctx* c, const uint64_t* block_keys_dev, int64_t n, const float* dense_dev) {
    GSDF_FLUSH(c);
    if (!c || (n > 0 && (!block_keys_dev || !dense_dev))) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    c->occ_dirty = true;
    gsdf_launch_unpack_blocks(c->stream, c->tab, (const unsigned long long*)block_keys_dev, n, dense_dev, c->st);
    HIP_TRY(hipGetLastError());
    return gsdf_sync(c);
}

int gsdf_dev_alloc(gsdf_ctx* c, void** dev_ptr, int64_t bytes) {
    if (!c || !dev_ptr || bytes <= 0) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(dev_ptr, (size_t)bytes));
    return GSDF_OK;
}
int gsdf_dev_free(gsdf_ctx* c, void* dev_ptr) {
    GSDF_FLUSH(c);
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(dev_ptr));
    return GSDF_OK;
}
int gsdf_dev_download(gsdf_ctx* c, void* host_dst, const void* dev_src, int64_t bytes) {
    GSDF_FLUSH(c);
    if (!c || !host_dst || !dev_src || bytes < 0) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(host_dst, dev_src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSDF_OK;
}
int gsdf_dev_upload(gsdf_ctx* c, void* dev_dst, const void* host_src, int64_t bytes) {
    if (!c || !dev_dst || !host_src || bytes < 0) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = flush_if_overlaps(c, dev_dst, bytes); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(dev_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSDF_OK;
}

/* ---- asynchronous frame staging (Scan3D: decode on host threads -> pinned buffer -> HBM, overlapped with the GPU) ---- */
int gsdf_host_alloc(gsdf_ctx* c, void** host_ptr, int64_t bytes) {
    if (!c || !host_ptr || bytes <= 0) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostMalloc(host_ptr, (size_t)bytes, hipHostMallocDefault));
    return GSDF_OK;
}
int gsdf_host_free(gsdf_ctx* c, void* host_ptr) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostFree(host_ptr));
    return GSDF_OK;
}
int gsdf_dev_upload_async(gsdf_ctx* c, void* dev_dst, const void* host_src, int64_t bytes) {
    if (!c || !dev_dst || !host_src || bytes < 0) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = flush_if_overlaps(c, dev_dst, bytes); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(dev_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    return GSDF_OK;
}
int gsdf_mark(gsdf_ctx* c, int64_t* mark) {
    GSDF_FLUSH(c);
    if (!c || !mark) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    /* a mark says "the stream has passed this point" -- what a staging slot's recycling needs -- and nothing about device
     * memory being visible to the host (the download entries synchronise for that): no system-scope fence, which would write
     * back and invalidate the caches between the frame's fusion and the next frame's first tracker pass */
    hipEvent_t e = nullptr;
    if (!c->mark_pool.empty()) { e = c->mark_pool.back(); c->mark_pool.pop_back(); }
    else HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    HIP_TRY(hipEventRecord(e, c->stream));
    c->marks.push_back({ ++c->mark_serial, e });
    *mark = c->mark_serial;
    return GSDF_OK;
}
/* retire the marks up to `mark` (they complete in stream order); wait != 0 blocks until then */
static int marks_reached(gsdf_ctx* c, int64_t mark, int wait, int* reached) {
    *reached = 1;
    while (!c->marks.empty() && c->marks.front().first <= mark) {
        hipEvent_t e = c->marks.front().second;
        hipError_t q = wait ? hipEventSynchronize(e) : hipEventQuery(e);
        if (q == hipErrorNotReady) { *reached = 0; return GSDF_OK; }
        if (q != hipSuccess) return fail(GSDF_ERR_HIP, std::string("gsdf mark: ") + hipGetErrorString(q));
        c->mark_pool.push_back(e);
        c->marks.pop_front();
    }
    return GSDF_OK;
}
int gsdf_mark_wait(gsdf_ctx* c, int64_t mark) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    int r = 0;
    return marks_reached(c, mark, 1, &r);
}
int gsdf_mark_reached(gsdf_ctx* c, int64_t mark, int* reached) {
    if (!c || !reached) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    return marks_reached(c, mark, 0, reached);
}

int gsdf_dev_upload_ahead(gsdf_ctx* c, void* dev_dst, const void* host_src, int64_t bytes, int64_t* upload) {
    if (!c || !dev_dst || !host_src || bytes < 0 || !upload) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    /* (a caller that keeps to the contract -- dev_dst is out of the stream's reach: its last reader passed a gsdf_mark, which
     * launched any waiting fusion -- never gets here with an overlap; the copy stream is not ordered behind the kernels) */
    if (c->pending.valid) {
        const uintptr_t a0 = (uintptr_t)dev_dst, b0 = (uintptr_t)c->pending.depth;
        if (a0 < b0 + (size_t)c->W * c->H * sizeof(float) && b0 < a0 + (uintptr_t)bytes)
            return fail(GSDF_ERR_INVALID, "gsdf_dev_upload_ahead into the depth image of a fusion that has not run yet (record a gsdf_mark "
                                          "behind its gsdf_update_dev and wait for it first)");
    }
    for (const float** q : { &c->nrm_ready_depth, &c->hint_next })      /* new contents: normals computed ahead are the old frame's */
        if (*q) {
            const uintptr_t a0 = (uintptr_t)dev_dst, b0 = (uintptr_t)*q;
            if (a0 < b0 + (size_t)c->W * c->H * sizeof(float) && b0 < a0 + (uintptr_t)bytes) *q = nullptr;
        }
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipMemcpyAsync(dev_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice, c->copy_stream));
    hipEvent_t e = nullptr;                                      /* (a pool of their own: these keep the default fences) */
    if (!c->upload_pool.empty()) { e = c->upload_pool.back(); c->upload_pool.pop_back(); }
    else HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e, c->copy_stream));
    c->uploads.push_back({ ++c->upload_serial, e });
    *upload = c->upload_serial;
    return GSDF_OK;
}
int gsdf_upload_wait(gsdf_ctx* c, int64_t upload) {
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    while (!c->uploads.empty() && c->uploads.front().first <= upload) {
        hipEvent_t e = c->uploads.front().second;
        const hipError_t q = hipEventSynchronize(e);
        if (q != hipSuccess) return fail(GSDF_ERR_HIP, std::string("gsdf upload: ") + hipGetErrorString(q));
        c->upload_pool.push_back(e);
        c->uploads.pop_front();
    }
    return GSDF_OK;
}

int gsdf_timer_start(gsdf_ctx* c) {
    GSDF_FLUSH(c);
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return GSDF_OK;
}
int gsdf_timer_stop_ms(gsdf_ctx* c, float* ms) {
    GSDF_FLUSH(c);
    if (!c || !ms) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return GSDF_OK;
}

int gsdf_profile(gsdf_ctx* c, int enable) {
    GSDF_FLUSH(c);
    if (!c) return fail(GSDF_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    prof_collect(c);
    c->profiling = enable != 0;
    if (enable) for (int k = 0; k < GSDF_PROF_SLOTS; ++k) { c->prof_ms[k] = 0; c->prof_n[k] = 0; c->prof_each[k].clear(); }
    return GSDF_OK;
}
int gsdf_profile_read_launches(gsdf_ctx* c, int slot, float* ms, int64_t max_n, int64_t* n) {
    if (!c || !n || slot < 0 || slot >= GSDF_PROF_SLOTS || max_n < 0 || (max_n > 0 && !ms)) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    prof_collect(c);
    const std::vector<float>& v = c->prof_each[slot];
    *n = (int64_t)v.size();
    for (int64_t i = 0; i < std::min<int64_t>(max_n, *n); ++i) ms[i] = v[i];
    return GSDF_OK;
}
int gsdf_profile_read(gsdf_ctx* c, double ms[3], int64_t launches[3]) {
    if (!c || !ms || !launches) return fail(GSDF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    prof_collect(c);
    for (int k = 0; k < 3; ++k) { ms[k] = c->prof_ms[k]; launches[k] = c->prof_n[k]; }
    return GSDF_OK;
}
int gsdf_profile_read_n(gsdf_ctx* c, int n, double* ms, int64_t* launches) {
    if (!c || !ms || !launches || n < 0 || n > GSDF_PROF_SLOTS) return fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    prof_collect(c);
    for (int k = 0; k < n; ++k) { ms[k] = c->prof_ms[k]; launches[k] = c->prof_n[k]; }
    return GSDF_OK;
}

} // extern "C"
