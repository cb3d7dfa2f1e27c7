/*
 * gsdf_merge.hip -- the exchange step of frame-sharded fusion (BASELINE.json north_star: "frames shard
 * naturally across the 8 GPUs of one node with a RCCL-over-xGMI all-reduce of per-voxel (weight, weighted-distance,
 * weighted-gradient) before mesh extraction").
 *
 * MapGradPixelSdf::update with a known pose (the GT-pose branch, main_scan_3d.cpp:250-254) depends only on (depth, pose)
 * and changes the map additively, so every rank (one process per GPU) fuses its own frames into its own map and ONE
 * exchange makes every map the sum of all:
 *   1. all-gather of the ranks' block-key arrays (the ids of their 4x4x4 blocks, 8 B per table entry, as they lie in HBM),
 *      sorted union on the device (rocPRIM radix sort + unique; identical on every rank);
 *   2. gsdf pack:    64 x 5 raw sums (w, s, gx, gy, gz) per block of the union into ONE dense device buffer
 *                    (zeros where this rank has nothing): 1280 B per block, independent of the number of ranks;
 *   3. all-reduce (sum, float32) of that buffer -- RCCL over xGMI; RCCL picks ring / direct by size;
 *   4. gsdf unpack:  the reduced sums become the map (missing blocks are inserted).
 * Everything runs on the context's own HIP stream (the RCCL calls take it), so there is no cross-stream hand-over.
 *
 * RCCL is NOT a link dependency of libgsdf.so: its entry points are resolved at run time from the RCCL the process
 * already holds (PyTorch's bundled copy, or librccl.so of the ROCm installation), so a process never has two.
 * gsdf_merge_allreduce_with takes the two collectives as callbacks on host buffers instead -- the transport-agnostic
 * form used by the world-size-2 tests on a one-GPU box (two ranks cannot share a device under RCCL) and by hosts that
 * bring their own communication layer.
 */
#include "gsdf_ctx.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <random>

namespace {

/* ---- RCCL, resolved at run time ------------------------------------------------------------------------------- */
struct rccl_api {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};

const rccl_api& rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;                                    /* first: whatever RCCL the process already has */
        if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
            for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" })
                if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
            if (!h) { api.why = "RCCL not found (librccl.so): " + std::string(dlerror() ? dlerror() : ""); return; }
        }
        auto sym = [&](const char* n) { void* p = h ? dlsym(h, n) : nullptr; return p ? p : dlsym(RTLD_DEFAULT, n); };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommCount && api.AllGather && api.AllReduce &&
                 api.GetErrorString;
        if (!api.ok) api.why = "RCCL library lacks an entry point";
    });
    return api;
}
#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return gsdf_fail(GSDF_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

/* ---- the two collectives the exchange needs, on DEVICE buffers ------------------------------------------------- */
struct transport {
    int nranks = 1;
    /* recv[r * bytes .. (r + 1) * bytes) = rank r's send buffer; on the context's stream */
    virtual int allgather(gsdf_ctx* c, const void* send_dev, void* recv_dev, size_t bytes) = 0;
    virtual int allreduce_sum_f32(gsdf_ctx* c, float* buf_dev, size_t n) = 0;
    /* bitwise OR of unsigned words whose set bits are disjoint between the ranks (so a sum does it too) */
    virtual int allreduce_or_u32(gsdf_ctx* c, uint32_t* buf_dev, size_t n) = 0;
    virtual ~transport() {}
};

struct rccl_transport : transport {
    ncclComm_t comm;
    int allgather(gsdf_ctx* c, const void* send_dev, void* recv_dev, size_t bytes) override {
        RCCL_TRY(rccl().AllGather(send_dev, recv_dev, bytes, ncclInt8, comm, c->stream));
        return GSDF_OK;
    }
    int allreduce_sum_f32(gsdf_ctx* c, float* buf_dev, size_t n) override {
        RCCL_TRY(rccl().AllReduce(buf_dev, buf_dev, n, ncclFloat32, ncclSum, comm, c->stream));
        return GSDF_OK;
    }
    int allreduce_or_u32(gsdf_ctx* c, uint32_t* buf_dev, size_t n) override {
        RCCL_TRY(rccl().AllReduce(buf_dev, buf_dev, n, ncclUint32, ncclSum, comm, c->stream));   /* disjoint bits: sum == or */
        return GSDF_OK;
    }
};

/* callbacks on HOST buffers: staged through host memory around every call */
struct host_transport : transport {
    const gsdf_collective* ops;
    std::vector<char> a, b;
    int allgather(gsdf_ctx* c, const void* send_dev, void* recv_dev, size_t bytes) override {
        a.resize(bytes); b.resize(bytes * (size_t)nranks);
        HIP_TRY(hipMemcpyAsync(a.data(), send_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (ops->allgather(ops->user, a.data(), b.data(), (int64_t)bytes) != 0) return gsdf_fail(GSDF_ERR_INVALID, "gsdf_collective.allgather failed");
        HIP_TRY(hipMemcpyAsync(recv_dev, b.data(), b.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return GSDF_OK;
    }
    int allreduce_sum_f32(gsdf_ctx* c, float* buf_dev, size_t n) override {
        a.resize(n * sizeof(float));
        HIP_TRY(hipMemcpyAsync(a.data(), buf_dev, a.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (ops->allreduce_sum_f32(ops->user, (float*)a.data(), (int64_t)n) != 0) return gsdf_fail(GSDF_ERR_INVALID, "gsdf_collective.allreduce_sum_f32 failed");
        HIP_TRY(hipMemcpyAsync(buf_dev, a.data(), a.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return GSDF_OK;
    }
    /* the callback interface has no integer reduction: gather every rank's words and OR them here (the host transport is
     * for tests and small worlds) */
    int allreduce_or_u32(gsdf_ctx* c, uint32_t* buf_dev, size_t n) override {
        const size_t bytes = n * sizeof(uint32_t);
        a.resize(bytes); b.resize(bytes * (size_t)nranks);
        HIP_TRY(hipMemcpyAsync(a.data(), buf_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (ops->allgather(ops->user, a.data(), b.data(), (int64_t)bytes) != 0) return gsdf_fail(GSDF_ERR_INVALID, "gsdf_collective.allgather failed");
        uint32_t* acc = (uint32_t*)a.data();
        std::memset(acc, 0, bytes);
        for (int r = 0; r < nranks; ++r) {
            const uint32_t* src = (const uint32_t*)(b.data() + (size_t)r * bytes);
            for (size_t i = 0; i < n; ++i) acc[i] |= src[i];
        }
        HIP_TRY(hipMemcpyAsync(buf_dev, a.data(), bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return GSDF_OK;
    }
};

/* Scratch of the exchange lives in the context and only grows (gsdf_ctx::mx): the first exchange of a context allocates,
 * later ones find their buffers -- like the communicator, they are set-up that stays outside the exchange itself. */
enum { MX_HDR = 0, MX_AGREE, MX_KEYS_ALL, MX_SORTED, MX_UNION, MX_TMP, MX_DENSE, MX_VIS, MX_COUNT_ };
static_assert(MX_COUNT_ <= GSDF_MX_BUFS, "gsdf_ctx::mx is too small");
int mx_ensure(gsdf_ctx* c, int i, size_t bytes, bool spare) {
    if (c->mx[i].bytes >= bytes && c->mx[i].p) return GSDF_OK;
    if (c->mx[i].p) { (void)hipFree(c->mx[i].p); c->mx[i].p = nullptr; c->mx[i].bytes = 0; }      /* no exchange is in flight: every one ends with a sync */
    const size_t want = std::max<size_t>(spare ? bytes + bytes / 4 : bytes, 256);
    HIP_TRY(hipMalloc(&c->mx[i].p, want));
    c->mx[i].bytes = want;
    return GSDF_OK;
}

int read_status(gsdf_ctx* c) {
    gsdf_dev_state s;
    HIP_TRY(hipMemcpyAsync(&s, c->st, sizeof(s), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (s.status & GSDF_STATUS_TABLE_FULL) return gsdf_fail(GSDF_ERR_TABLE_FULL, "voxel hash table full (probe budget exhausted)");
    if (s.status & GSDF_STATUS_KEY_RANGE) return gsdf_fail(GSDF_ERR_KEY_RANGE, "voxel index outside the packable +-2^20 range");
    return GSDF_OK;
}

/* What every rank tells the others before anything sized is exchanged.  `err` makes local failures collective: a rank that
 * could not prepare (allocation, launch) still takes part in this all-gather, and then ALL ranks return an error together
 * instead of one leaving its peers inside a collective. */
struct merge_hdr {
    long long cap_blocks;        /* entries of this rank's block-key array: must be equal on all ranks (same capacity_log2) */
    long long frames;            /* Sdf::counter_ of this rank: frames it integrated (filled in on the device) */
    long long vis_words;         /* words per voxel of its vis_ bit-vectors, 0 = not enabled */
    long long err;               /* GSDF_ERR_* of its preparation, 0 = fine */
    unsigned long long token;    /* random: a rank finds its own position in the gathered list by it (the callback transport
                                    does not tell a rank its number) */
    long long dense_blocks;      /* blocks its pack / all-reduce buffers can hold as they are (so that every rank knows whether
                                    anybody has to allocate once the size of the union is known) */
    long long pad[2];
};
static_assert(sizeof(merge_hdr) == 64, "header layout");

/* the header of this rank, completed on the device (Sdf::counter_ lives there): no host read before the first collective */
__global__ void k_merge_hdr(merge_hdr* out, merge_hdr mine, const gsdf_dev_state* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { mine.frames = st->frames; *out = mine; }
}
/* size of the union: the sorted unique list ends with the EMPTY marker whenever any rank's key array had a free entry */
__global__ void k_union_size(const unsigned long long* uni, unsigned long long* count) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long n = *count;
        *count = n - ((n > 0 && uni[n - 1] == GSDF_KEY_EMPTY) ? 1ull : 0ull);
    }
}

/* agreement point: every rank reports whether its buffers for the next stage exist */
int agree(gsdf_ctx* c, transport& tr, int local_rc, const char* what) {
    const int R = tr.nranks;
    long long* scratch_dev = (long long*)c->mx[MX_AGREE].p;          /* R + 1 words */
    const long long mine = local_rc;
    HIP_TRY(hipMemcpyAsync(scratch_dev + R, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
    int rc = tr.allgather(c, scratch_dev + R, scratch_dev, sizeof(long long));
    if (rc) return rc;
    std::vector<long long> all((size_t)R);
    HIP_TRY(hipMemcpyAsync(all.data(), scratch_dev, (size_t)R * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (local_rc) return local_rc;                                    /* g_gsdf_err already says why */
    for (int r = 0; r < R; ++r)
        if (all[(size_t)r]) return gsdf_fail((int)all[(size_t)r], std::string("gsdf_merge_allreduce: rank ") + std::to_string(r) + " failed to " + what);
    return GSDF_OK;
}

unsigned long long random_token(const gsdf_ctx* c) {
    std::random_device rd;
    unsigned long long t = ((unsigned long long)rd() << 32) ^ (unsigned long long)rd();
    t ^= gsdf_hash64((unsigned long long)(uintptr_t)c ^ ((unsigned long long)getpid() << 32));
    return t ? t : 1ull;
}

/* The exchange.  Host synchronisations: (1) the gathered headers -- nothing sized by another rank's numbers is exchanged before
 * they were checked --, (2) the size of the union, which the all-reduce's element count needs, (3) the sticky status at the
 * end; a fourth one only when some rank has to allocate its pack buffers (first exchange of a context, or a larger union than
 * ever before).  The union itself is made on the device: the ranks' block-key ARRAYS are all-gathered as they are (8 B per
 * table entry: 0.5 MB at 2^22 records -- no compaction pass, no count to wait for), sorted (rocPRIM radix sort) and
 * de-duplicated there; empty entries sort to the end.  Every rank computes the same list. */
int merge_impl(gsdf_ctx* c, transport& tr, int64_t* n_blocks_out, int64_t* bytes_out) {
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = gsdf_flush_pending(c)) return rc;            /* the last frame's fusion may still wait for a successor (gsdf_update_dev) */
    if (c->merged && tr.nranks > 1)
        return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: this map already holds the sum of all ranks (the exchange is one-shot: "
                                           "a second one would count every rank's frames again); gsdf_reset starts over");
    size_t cap = c->n_slots / GSDF_BLOCK_VOX;
    const int R = tr.nranks;
    size_t n_all = cap * (size_t)R;
    /* two small buffers are needed to talk at all: without them this rank cannot even report its failure */
    if (int rc = mx_ensure(c, MX_HDR, (size_t)(R + 1) * sizeof(merge_hdr), false)) return rc;
    if (int rc = mx_ensure(c, MX_AGREE, (size_t)(R + 1) * sizeof(long long), false)) return rc;
    /* 1. everything whose size follows from (ranks, capacity) alone is prepared BEFORE the header goes out: a failure travels
     *    in the header */
    size_t tmp_sort = 0, tmp_uniq = 0;
    auto prepare = [&]() -> int {
        if (int rc = mx_ensure(c, MX_KEYS_ALL, n_all * sizeof(unsigned long long), false)) return rc;
        if (int rc = mx_ensure(c, MX_SORTED, n_all * sizeof(unsigned long long), false)) return rc;
        if (int rc = mx_ensure(c, MX_UNION, n_all * sizeof(unsigned long long), false)) return rc;
        HIP_TRY(gsdf_sort_keys_u64(nullptr, &tmp_sort, nullptr, nullptr, n_all, c->stream));
        HIP_TRY(gsdf_unique_u64(nullptr, &tmp_uniq, nullptr, nullptr, nullptr, n_all, c->stream));
        return mx_ensure(c, MX_TMP, std::max(tmp_sort, tmp_uniq), false);
    };
    const int prep = prepare();
    const int vw_mine = c->vis ? c->vis_words : 0;
    auto dense_capacity = [&]() -> long long {      /* blocks the pack buffers hold as they are */
        long long b = c->mx[MX_DENSE].p ? (long long)(c->mx[MX_DENSE].bytes / (GSDF_BLOCK_VOX * 5 * sizeof(float))) : 0;
        if (vw_mine) b = std::min(b, c->mx[MX_VIS].p ? (long long)(c->mx[MX_VIS].bytes / (GSDF_BLOCK_VOX * (size_t)vw_mine * sizeof(uint32_t))) : 0ll);
        return b;
    };
    merge_hdr* hd = (merge_hdr*)c->mx[MX_HDR].p;
    merge_hdr mine;
    std::memset(&mine, 0, sizeof(mine));
    mine.cap_blocks = (long long)cap; mine.vis_words = vw_mine; mine.err = prep; mine.token = random_token(c);
    mine.dense_blocks = dense_capacity();
    hipLaunchKernelGGL(k_merge_hdr, dim3(1), dim3(64), 0, c->stream, hd + R, mine, c->st);
    HIP_TRY(hipGetLastError());
    int rc = tr.allgather(c, hd + R, hd, sizeof(merge_hdr));
    if (rc) return rc;
    std::vector<merge_hdr> hdr((size_t)R);
    HIP_TRY(hipMemcpyAsync(hdr.data(), hd, (size_t)R * sizeof(merge_hdr), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                                                          /* (1) */
    if (prep) return prep;
    /* from here on every decision is taken from the gathered headers, i.e. identically on all ranks */
    long long frames_total = 0, frames_before = 0, dense_min = hdr[0].dense_blocks;
    int me = -1, dup = 0;
    for (int r = 0; r < R; ++r) {
        const merge_hdr& h = hdr[(size_t)r];
        if (h.err) return gsdf_fail((int)h.err, "gsdf_merge_allreduce: rank " + std::to_string(r) + " failed to prepare its buffers");
        if (h.cap_blocks <= 0 || (h.cap_blocks & (h.cap_blocks - 1)) || h.cap_blocks > (1ll << 24) || h.frames < 0 || h.dense_blocks < 0)
            return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: inconsistent header from rank " + std::to_string(r));
        if (h.vis_words != hdr[0].vis_words)
            return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: gsdf_enable_vis must be called with the same frame count on every rank (or on none)");
        for (int q = 0; q < r; ++q) dup |= hdr[(size_t)q].token == h.token;
        if (h.token == mine.token) me = r;
        frames_total += h.frames;
        dense_min = std::min(dense_min, h.dense_blocks);
    }
    if (dup || me < 0) return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: rank tokens collide (retry)");   /* seen by all ranks */
    for (int r = 0; r < me; ++r) frames_before += hdr[(size_t)r].frames;
    const int vw = (int)hdr[0].vis_words;
    if (vw && frames_total > 32ll * vw)
        return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: the ranks integrated " + std::to_string(frames_total) +
                                           " frames, gsdf_enable_vis reserved " + std::to_string(32ll * vw) + " bits per voxel");
    /* The ranks exchange their block-key ARRAYS, so the tables must be of one size.  They usually are; where they are not (one
     * rank's table grew during its scan: gsdf_set_auto_grow), the smaller ones grow to the largest capacity first -- every rank
     * knows from the headers that this happens, and they meet once more to agree that it worked. */
    long long cap_max = 0;
    bool any_grow = false;
    for (int r = 0; r < R; ++r) { cap_max = std::max(cap_max, hdr[(size_t)r].cap_blocks); any_grow |= hdr[(size_t)r].cap_blocks != hdr[0].cap_blocks; }
    if (any_grow) {
        int grc = GSDF_OK;
        if ((long long)cap < cap_max) {
            int lg = 0;
            while (((size_t)1 << lg) < (size_t)cap_max * GSDF_BLOCK_VOX) ++lg;
            grc = gsdf_grow_impl(c, lg);
        }
        if (!grc) {
            cap = c->n_slots / GSDF_BLOCK_VOX;
            n_all = cap * (size_t)R;
            grc = prepare();
        }
        rc = agree(c, tr, grc, "grow its table to the largest rank's capacity");
        if (rc) return rc;
    }
    /* 2. the ranks' key arrays -> sorted union, on the device */
    unsigned long long* keys_all = (unsigned long long*)c->mx[MX_KEYS_ALL].p;
    unsigned long long* sorted = (unsigned long long*)c->mx[MX_SORTED].p;
    unsigned long long* uk = (unsigned long long*)c->mx[MX_UNION].p;
    rc = tr.allgather(c, c->tab.bkeys, keys_all, cap * sizeof(unsigned long long));
    if (rc) return rc;
    size_t tmp_bytes = c->mx[MX_TMP].bytes;
    HIP_TRY(gsdf_sort_keys_u64(c->mx[MX_TMP].p, &tmp_bytes, keys_all, sorted, n_all, c->stream));
    tmp_bytes = c->mx[MX_TMP].bytes;
    HIP_TRY(gsdf_unique_u64(c->mx[MX_TMP].p, &tmp_bytes, sorted, uk, c->counter, n_all, c->stream));
    hipLaunchKernelGGL(k_union_size, dim3(1), dim3(64), 0, c->stream, uk, c->counter);
    HIP_TRY(hipGetLastError());
    unsigned long long nu64 = 0;
    HIP_TRY(hipMemcpyAsync(&nu64, c->counter, sizeof(nu64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                                                          /* (2) */
    const size_t nu = (size_t)nu64;
    const size_t vis_count = nu * GSDF_BLOCK_VOX * (size_t)vw;
    if (n_blocks_out) *n_blocks_out = (int64_t)nu;
    if (bytes_out) *bytes_out = (int64_t)(nu * GSDF_BLOCK_VOX * 5 * sizeof(float) + vis_count * sizeof(uint32_t));
    /* 3. pack, all-reduce, unpack -- the sums, then the vis_ bit-vectors.  Whether ANY rank has to allocate is known to all of
     *    them (the headers carry the capacities): only then do they meet once more to agree that it worked */
    if ((long long)nu > dense_min) {
        auto alloc_dense = [&]() -> int {
            if (int r2 = mx_ensure(c, MX_DENSE, nu * GSDF_BLOCK_VOX * 5 * sizeof(float), true)) return r2;
            if (vw) return mx_ensure(c, MX_VIS, vis_count * sizeof(uint32_t), true);
            return GSDF_OK;
        };
        rc = agree(c, tr, alloc_dense(), "allocate the exchange buffers");
        if (rc) return rc;
    }
    if (nu) {
        float* dense = (float*)c->mx[MX_DENSE].p;
        uint32_t* dense_vis = (uint32_t*)c->mx[MX_VIS].p;
        gsdf_launch_pack_blocks(c->stream, c->tab, uk, (long long)nu, dense);
        if (vw) gsdf_launch_pack_vis(c->stream, c->tab, c->vis, vw, frames_before, uk, (long long)nu, dense_vis);
        HIP_TRY(hipGetLastError());
        rc = tr.allreduce_sum_f32(c, dense, nu * GSDF_BLOCK_VOX * 5);
        if (rc) return rc;
        if (vw) {
            rc = tr.allreduce_or_u32(c, dense_vis, vis_count);
            if (rc) return rc;
        }
        c->occ_dirty = true;
        gsdf_launch_unpack_blocks(c->stream, c->tab, uk, (long long)nu, dense, c->st);
        if (vw) gsdf_launch_unpack_vis(c->stream, c->tab, c->vis, vw, uk, (long long)nu, dense_vis);
    }
    /* Sdf::counter_ of the merged map: the frames of all ranks (frame f of rank r is integrated frame frames_before(r) + f) */
    gsdf_launch_set_frames(c->stream, c->st, frames_total);
    HIP_TRY(hipGetLastError());
    c->merged = R > 1;
    return read_status(c);                                                                             /* (3) */
}

/* ---- growing the map (gsdf_grow): every block of the old table moves into a table of twice (or more) the entries ---- */
/* one wave per old entry: lane 0 claims the block's entry in the new key array, the 64 lanes copy its 64 records (and their
 * vis_ words); the new table is empty and larger, so the probe budget cannot run out */
__global__ __launch_bounds__(256) void k_rehash(gsdf_table from, gsdf_table to, const uint32_t* vis_from, uint32_t* vis_to, int vw,
                                                size_t n_blocks_from, gsdf_dev_state* st) {
    const size_t b = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_blocks_from) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long bk = from.bkeys[b];
    if (bk == GSDF_KEY_EMPTY) return;
    int nb = -1;
    if (lane == 0) {
        const uint32_t h = gsdf_hash(bk) & to.block_mask;
        nb = gsdf_block_find_or_insert(to, bk, h, to.bkeys[h]);
        if (nb < 0) atomicOr(&st->status, GSDF_STATUS_TABLE_FULL);
    }
    nb = __shfl(nb, 0);
    if (nb < 0) return;
    const size_t src = b * GSDF_BLOCK_VOX + lane, dst = (size_t)nb * GSDF_BLOCK_VOX + lane;
    const uint4* ps = reinterpret_cast<const uint4*>(from.vox + src);
    uint4* pd = reinterpret_cast<uint4*>(to.vox + dst);
    pd[0] = ps[0]; pd[1] = ps[1];
    for (int w = 0; w < vw; ++w) vis_to[dst * vw + w] = vis_from[src * vw + w];
}
/* gsdf_merge_from: the map of ANOTHER context on the same device added into this one -- one wave per block entry of the source:
 * lane 0 finds or claims the block in the destination, the 64 lanes add their records (plain read-modify-write: a source block
 * has one wave, a destination block one source block) and OR the vis_ words in, shifted by the destination's frame count so
 * that frame f of the source becomes integrated frame (frames of the destination) + f, as in the exchange between ranks. */
__global__ __launch_bounds__(256) void k_merge_from(gsdf_table from, gsdf_table to, const uint32_t* vis_from, uint32_t* vis_to, int vw,
                                                    long long bit_offset, size_t n_blocks_from, gsdf_dev_state* st) {
    const size_t b = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_blocks_from) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long bk = from.bkeys[b];
    if (bk == GSDF_KEY_EMPTY) return;
    int nb = -1;
    if (lane == 0) {
        const uint32_t h = gsdf_hash(bk) & to.block_mask;
        nb = gsdf_block_find_or_insert(to, bk, h, to.bkeys[h]);
        if (nb < 0) atomicOr(&st->status, GSDF_STATUS_TABLE_FULL);
    }
    nb = __shfl(nb, 0);
    if (nb < 0) return;
    const size_t src = b * GSDF_BLOCK_VOX + lane, dst = (size_t)nb * GSDF_BLOCK_VOX + lane;
    const gsdf_payload s = from.vox[src];
    if (!(s.w > 0.f)) return;                                  /* the voxel does not exist in the source */
    gsdf_payload d = to.vox[dst];
    d.w += s.w; d.s += s.s; d.gx += s.gx; d.gy += s.gy; d.gz += s.gz;
    to.vox[dst] = d;
    const int wsh = (int)(bit_offset >> 5), bsh = (int)(bit_offset & 31);
    for (int w = 0; w < vw; ++w) {
        const uint32_t v = vis_from[src * vw + w];
        if (!v) continue;
        if (w + wsh < vw) vis_to[dst * vw + w + wsh] |= v << bsh;
        if (bsh && w + wsh + 1 < vw) vis_to[dst * vw + w + wsh + 1] |= v >> (32 - bsh);
    }
}
/* occupied entries of the key array -> one pinned 64-bit host word, count | tag << 32 (auto-grow: the host looks at it without
 * waiting; the tag is the number of the frame entry that enqueued this count, so the host knows how old the number is) */
__global__ __launch_bounds__(256) void k_count_blocks(const unsigned long long* bkeys, size_t n, unsigned int* scratch,
                                                      unsigned long long* host_word, unsigned int tag) {
    __shared__ unsigned int part[4];
    unsigned int c = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c += bkeys[i] != GSDF_KEY_EMPTY ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&scratch[0], part[0] + part[1] + part[2] + part[3]);
        __threadfence();
        if (atomicAdd(&scratch[1], 1u) + 1u == gridDim.x) {               /* the last workgroup publishes and resets */
            const unsigned long long cnt = __hip_atomic_load(&scratch[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_word, cnt | ((unsigned long long)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&scratch[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&scratch[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

} // namespace

/* enqueue the count of existing blocks into the pinned words progress[4..5] (gsdf_capi.hip: at the top of every frame entry
 * while auto-grow is on), tagged with the entry's number */
/* the same count into a device word (zeroed by the caller) */
__global__ __launch_bounds__(256) void k_count_blocks_dev(const unsigned long long* bkeys, size_t n, unsigned long long* out) {
    unsigned int c = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c += bkeys[i] != GSDF_KEY_EMPTY ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

void gsdf_enqueue_block_count(gsdf_ctx* c, unsigned int tag) {
    if (!c->progress_dev || !c->grow_scratch) return;
    const size_t cap = c->n_slots / GSDF_BLOCK_VOX;
    hipLaunchKernelGGL(k_count_blocks, dim3(64), dim3(256), 0, c->stream, c->tab.bkeys, cap, c->grow_scratch,
                       reinterpret_cast<unsigned long long*>(c->progress_dev + 4), tag);
    (void)hipGetLastError();
}

int gsdf_grow_impl(gsdf_ctx* c, int new_capacity_log2) {
    if (!c) return gsdf_fail(GSDF_ERR_INVALID, "null context");
    if (new_capacity_log2 <= c->capacity_log2 || new_capacity_log2 > 30)
        return gsdf_fail(GSDF_ERR_INVALID, "gsdf_grow: the new capacity_log2 must be larger than the present one and <= 30");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = gsdf_flush_pending(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t n_new = (size_t)1 << new_capacity_log2;
    gsdf_table nt{ nullptr, nullptr, 0, nullptr };
    uint32_t* nvis = nullptr;
    auto release = [&]() { if (nt.vox) (void)hipFree(nt.vox); if (nt.bkeys) (void)hipFree(nt.bkeys); if (nt.occ) (void)hipFree(nt.occ); if (nvis) (void)hipFree(nvis); };
    hipError_t e;
    if ((e = hipMalloc((void**)&nt.vox, n_new * sizeof(gsdf_payload))) != hipSuccess ||
        (e = hipMalloc((void**)&nt.bkeys, (n_new / GSDF_BLOCK_VOX) * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc((void**)&nt.occ, n_new / 8 + std::max<size_t>(n_new / GSDF_BLOCK_VOX / 8, 4))) != hipSuccess ||
        (c->vis && (e = hipMalloc((void**)&nvis, n_new * (size_t)c->vis_words * sizeof(uint32_t))) != hipSuccess)) {
        release();
        (void)hipGetLastError();
        return gsdf_fail(GSDF_ERR_HIP, std::string("gsdf_grow: ") + hipGetErrorString(e));      /* the map is untouched */
    }
    nt.block_mask = (uint32_t)(n_new / GSDF_BLOCK_VOX - 1);
    /* A sticky GSDF_STATUS_TABLE_FULL from an earlier fusion must not stop the map from growing: the rehash into an
     * empty, larger table reports its own failure through the same bit, so the bit is looked at before and after.  (The old
     * error stays sticky -- samples WERE dropped -- and gsdf_sync keeps reporting it until gsdf_reset.)  Every failure path
     * below releases the new table. */
    gsdf_dev_state s0, s1;
    auto bail = [&](hipError_t err, const char* what) {
        (void)hipStreamSynchronize(c->stream);                 /* nothing may still write into the buffers that are freed */
        release();
        (void)hipGetLastError();
        return gsdf_fail(GSDF_ERR_HIP, std::string("gsdf_grow: ") + what + ": " + hipGetErrorString(err));
    };
    if ((e = hipMemcpy(&s0, c->st, sizeof(s0), hipMemcpyDeviceToHost)) != hipSuccess) return bail(e, "reading the status");
    const int full_before = s0.status & GSDF_STATUS_TABLE_FULL;
    if (full_before) {                                         /* cleared for the rehash, restored below */
        const int cleared = s0.status & ~GSDF_STATUS_TABLE_FULL;
        if ((e = hipMemcpy(&c->st->status, &cleared, sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "clearing the status");
    }
    gsdf_launch_table_clear(c->stream, nt, n_new);
    if (nvis && (e = hipMemsetAsync(nvis, 0, n_new * (size_t)c->vis_words * sizeof(uint32_t), c->stream)) != hipSuccess) return bail(e, "clearing vis_");
    const size_t nb_old = c->n_slots / GSDF_BLOCK_VOX;
    hipLaunchKernelGGL(k_rehash, dim3((unsigned int)((nb_old + 3) / 4)), dim3(256), 0, c->stream, c->tab, nt, c->vis, nvis, c->vis ? c->vis_words : 0,
                       nb_old, c->st);
    if ((e = hipGetLastError()) != hipSuccess) return bail(e, "launching the rehash");
    if ((e = hipMemcpyAsync(&s1, c->st, sizeof(s1), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return bail(e, "reading the status");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(e, "the rehash");
    if (full_before) {
        const int restored = s1.status | GSDF_STATUS_TABLE_FULL;
        (void)hipMemcpy(&c->st->status, &restored, sizeof(int), hipMemcpyHostToDevice);
    }
    if (s1.status & GSDF_STATUS_TABLE_FULL) {                  /* cannot happen: the new table is empty and larger */
        release();
        return gsdf_fail(GSDF_ERR_TABLE_FULL, "gsdf_grow: the rehash ran out of probes");
    }
    (void)hipFree(c->tab.vox); (void)hipFree(c->tab.bkeys); (void)hipFree(c->tab.occ);
    if (c->vis) (void)hipFree(c->vis);
    c->tab = nt; c->vis = nvis;
    c->n_slots = n_new; c->capacity_log2 = new_capacity_log2;
    c->occ_dirty = true;                                       /* the raycaster's filters are rebuilt from the keys when it next runs */
    /* PhotoBA's gate list was sized for the old table: the sweeps fall back to the whole table until the next gsdf_ba_setup */
    if (c->ba_gate_list) { (void)hipFree(c->ba_gate_list); c->ba_gate_list = nullptr; }
    if (c->ba_mean) { (void)hipFree(c->ba_mean); c->ba_mean = nullptr; }
    c->ba_gate_fresh = false; c->ba_mean_valid = false;
    c->grow_forget = true;                                     /* auto-grow: the counts in flight describe the old table */
    return GSDF_OK;
}

namespace {
} // namespace

extern "C" {

int gsdf_rccl_unique_id(char id128[128]) {
    if (!id128) return gsdf_fail(GSDF_ERR_INVALID, "null argument");
    if (!rccl().ok) return gsdf_fail(GSDF_ERR_INVALID, rccl().why);
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId id;
    RCCL_TRY(rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return GSDF_OK;
}

int gsdf_rccl_comm_init(void** comm, int nranks, const char id128[128], int rank, int device) {
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return gsdf_fail(GSDF_ERR_INVALID, "bad argument");
    if (!rccl().ok) return gsdf_fail(GSDF_ERR_INVALID, rccl().why);
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t cm = nullptr;
    RCCL_TRY(rccl().CommInitRank(&cm, nranks, id, rank));
    *comm = (void*)cm;
    return GSDF_OK;
}

int gsdf_rccl_comm_count(void* comm, int* nranks) {
    if (!comm || !nranks) return gsdf_fail(GSDF_ERR_INVALID, "null argument");
    if (!rccl().ok) return gsdf_fail(GSDF_ERR_INVALID, rccl().why);
    RCCL_TRY(rccl().CommCount((ncclComm_t)comm, nranks));
    return GSDF_OK;
}

int gsdf_rccl_comm_destroy(void* comm) {
    if (!comm) return GSDF_OK;
    if (!rccl().ok) return gsdf_fail(GSDF_ERR_INVALID, rccl().why);
    RCCL_TRY(rccl().CommDestroy((ncclComm_t)comm));
    return GSDF_OK;
}

/* Two shard contexts on ONE device ("G logical shards on 1 GPU + local merge"): dst += src.  Both fused frames of the
 * same job with known poses, each on its own stream; the sums are additive, the frame shards contiguous (dst's frames first). */
int gsdf_merge_from(gsdf_ctx* dst, gsdf_ctx* src) {
    if (!dst || !src || dst == src) return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_from: two different contexts are needed");
    if (dst->device != src->device) return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_from: the contexts live on different devices (use gsdf_merge_allreduce between devices)");
    if (dst->voxel_size != src->voxel_size || dst->T != src->T) return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_from: voxel size / truncation differ");
    if ((dst->vis != nullptr) != (src->vis != nullptr) || (dst->vis && dst->vis_words != src->vis_words))
        return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_from: gsdf_enable_vis must have been called alike on both contexts");
    HIP_TRY(hipSetDevice(dst->device));
    if (int rc = gsdf_flush_pending(src)) return rc;
    if (int rc = gsdf_flush_pending(dst)) return rc;
    /* the blocks both maps hold, counted behind everything queued on their streams and read with the state words */
    gsdf_dev_state ss, ds;
    unsigned long long n_src = 0, n_dst = 0;
    const size_t nb_src = src->n_slots / GSDF_BLOCK_VOX;
    HIP_TRY(hipMemsetAsync(src->counter, 0, sizeof(unsigned long long), src->stream));
    hipLaunchKernelGGL(k_count_blocks_dev, dim3(64), dim3(256), 0, src->stream, src->tab.bkeys, nb_src, src->counter);
    HIP_TRY(hipMemcpyAsync(&n_src, src->counter, sizeof(n_src), hipMemcpyDeviceToHost, src->stream));
    HIP_TRY(hipMemcpyAsync(&ss, src->st, sizeof(ss), hipMemcpyDeviceToHost, src->stream));
    HIP_TRY(hipStreamSynchronize(src->stream));              /* the source map is complete (and stays untouched) */
    if (ss.status & GSDF_STATUS_TABLE_FULL) return gsdf_fail(GSDF_ERR_TABLE_FULL, "gsdf_merge_from: the source map reported a full table");
    HIP_TRY(hipMemsetAsync(dst->counter, 0, sizeof(unsigned long long), dst->stream));
    hipLaunchKernelGGL(k_count_blocks_dev, dim3(64), dim3(256), 0, dst->stream, dst->tab.bkeys, dst->n_slots / GSDF_BLOCK_VOX, dst->counter);
    HIP_TRY(hipMemcpyAsync(&n_dst, dst->counter, sizeof(n_dst), hipMemcpyDeviceToHost, dst->stream));
    HIP_TRY(hipMemcpyAsync(&ds, dst->st, sizeof(ds), hipMemcpyDeviceToHost, dst->stream));
    HIP_TRY(hipStreamSynchronize(dst->stream));
    /* Room in dst: the union holds at most n_dst + n_src blocks.  Beyond the 45 % at which the frame entries double
     * the table, dst is doubled here too when gsdf_set_auto_grow allows it; where it does not and the worst case would pass
     * 90 % of the entries (the probe budget runs out near 95 %), the call fails BEFORE dst is touched. */
    {
        const size_t worst = (size_t)n_dst + (size_t)n_src;
        int need = dst->capacity_log2;
        while (need < 30 && worst * 100u > (((size_t)1 << need) / GSDF_BLOCK_VOX) * 45u) ++need;
        if (need > dst->capacity_log2 && dst->auto_grow_max > dst->capacity_log2) {
            if (int rc = gsdf_grow_impl(dst, std::min(need, dst->auto_grow_max))) return rc;
        }
        if (worst * 100u > (dst->n_slots / GSDF_BLOCK_VOX) * 90u)
            return gsdf_fail(GSDF_ERR_TABLE_FULL, "gsdf_merge_from: dst cannot hold the blocks of both maps (" + std::to_string(n_dst) + " + " +
                             std::to_string(n_src) + " of " + std::to_string(dst->n_slots / GSDF_BLOCK_VOX) + " entries); dst is unchanged -- gsdf_grow it or allow it with gsdf_set_auto_grow");
    }
    hipLaunchKernelGGL(k_merge_from, dim3((unsigned int)((nb_src + 3) / 4)), dim3(256), 0, dst->stream, src->tab, dst->tab, src->vis, dst->vis,
                       dst->vis ? dst->vis_words : 0, (long long)ds.frames, nb_src, dst->st);
    HIP_TRY(hipGetLastError());
    dst->occ_dirty = true;
    dst->ba_gate_fresh = false;
    dst->grow_forget = true;
    /* synchronises: the source may be reset or destroyed afterwards.  Sdf::counter_ only advances when every block found its
     * entry: behind a GSDF_ERR_TABLE_FULL here (possible between 45 % and 90 % only in theory: a probe chain that runs out early)
     * dst holds PART of src and is to be reset. */
    if (int rc = read_status(dst)) return rc;
    gsdf_launch_set_frames(dst->stream, dst->st, (long long)(ds.frames + ss.frames));   /* Sdf::counter_ = frames of both shards */
    HIP_TRY(hipStreamSynchronize(dst->stream));
    return GSDF_OK;
}

int gsdf_grow(gsdf_ctx* c, int new_capacity_log2) { return gsdf_grow_impl(c, new_capacity_log2); }

/* Set-up of the exchange that does not depend on the other ranks' data, for hosts that keep it out of a timed region (like the
 * communicator): the scratch sized by (ranks, capacity), the pack buffers for twice the blocks this map holds now, and one run of
 * the sort / unique kernels (their code is loaded on first use: ~10 ms).  Optional -- the exchange does all of it itself. */
int gsdf_merge_prepare(gsdf_ctx* c, int nranks) {
    if (!c || nranks < 1) return gsdf_fail(GSDF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t cap = c->n_slots / GSDF_BLOCK_VOX, n_all = cap * (size_t)nranks;
    if (int rc = mx_ensure(c, MX_HDR, (size_t)(nranks + 1) * sizeof(merge_hdr), false)) return rc;
    if (int rc = mx_ensure(c, MX_AGREE, (size_t)(nranks + 1) * sizeof(long long), false)) return rc;
    if (int rc = mx_ensure(c, MX_KEYS_ALL, n_all * sizeof(unsigned long long), false)) return rc;
    if (int rc = mx_ensure(c, MX_SORTED, n_all * sizeof(unsigned long long), false)) return rc;
    if (int rc = mx_ensure(c, MX_UNION, n_all * sizeof(unsigned long long), false)) return rc;
    size_t tmp_sort = 0, tmp_uniq = 0;
    HIP_TRY(gsdf_sort_keys_u64(nullptr, &tmp_sort, nullptr, nullptr, n_all, c->stream));
    HIP_TRY(gsdf_unique_u64(nullptr, &tmp_uniq, nullptr, nullptr, nullptr, n_all, c->stream));
    if (int rc = mx_ensure(c, MX_TMP, std::max(tmp_sort, tmp_uniq), false)) return rc;
    /* one run on this map's own keys (the other ranks' parts of the buffer: empty entries) */
    unsigned long long* keys_all = (unsigned long long*)c->mx[MX_KEYS_ALL].p;
    HIP_TRY(hipMemsetAsync(keys_all, 0xFF, n_all * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipMemcpyAsync(keys_all, c->tab.bkeys, cap * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
    size_t tb = c->mx[MX_TMP].bytes;
    HIP_TRY(gsdf_sort_keys_u64(c->mx[MX_TMP].p, &tb, keys_all, (unsigned long long*)c->mx[MX_SORTED].p, n_all, c->stream));
    tb = c->mx[MX_TMP].bytes;
    HIP_TRY(gsdf_unique_u64(c->mx[MX_TMP].p, &tb, (unsigned long long*)c->mx[MX_SORTED].p, (unsigned long long*)c->mx[MX_UNION].p, c->counter, n_all, c->stream));
    hipLaunchKernelGGL(k_union_size, dim3(1), dim3(64), 0, c->stream, (unsigned long long*)c->mx[MX_UNION].p, c->counter);
    unsigned long long n_mine = 0;
    HIP_TRY(hipMemcpyAsync(&n_mine, c->counter, sizeof(n_mine), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t guess = std::min<size_t>(cap, std::max<size_t>(2 * (size_t)n_mine, 1024));
    if (int rc = mx_ensure(c, MX_DENSE, guess * GSDF_BLOCK_VOX * 5 * sizeof(float), false)) return rc;
    if (c->vis) if (int rc = mx_ensure(c, MX_VIS, guess * GSDF_BLOCK_VOX * (size_t)c->vis_words * sizeof(uint32_t), false)) return rc;
    return GSDF_OK;
}

int gsdf_merge_allreduce(gsdf_ctx* c, void* nccl_comm, int64_t* n_blocks, int64_t* bytes) {
    if (!c || !nccl_comm) return gsdf_fail(GSDF_ERR_INVALID, "null argument");
    if (!rccl().ok) return gsdf_fail(GSDF_ERR_INVALID, rccl().why);
    rccl_transport tr;
    tr.comm = (ncclComm_t)nccl_comm;
    RCCL_TRY(rccl().CommCount(tr.comm, &tr.nranks));
    return merge_impl(c, tr, n_blocks, bytes);
}

int gsdf_merge_allreduce_with(gsdf_ctx* c, const gsdf_collective* ops, int64_t* n_blocks, int64_t* bytes) {
    if (!c || !ops || !ops->allgather || !ops->allreduce_sum_f32 || ops->nranks < 1) return gsdf_fail(GSDF_ERR_INVALID, "bad argument");
    host_transport tr;
    tr.ops = ops;
    tr.nranks = ops->nranks;
    return merge_impl(c, tr, n_blocks, bytes);
}

} // extern "C"
