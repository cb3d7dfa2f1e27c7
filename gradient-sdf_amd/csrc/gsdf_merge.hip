/*
 * gsdf_merge.hip -- the exchange step of frame-sharded fusion (BASELINE.json north_star: "frames shard
 * naturally across the 8 GPUs of one node with a RCCL-over-xGMI all-reduce of per-voxel (weight, weighted-distance,
 * weighted-gradient) before mesh extraction").
 *
 * MapGradPixelSdf::update with a known pose (the GT-pose branch, main_scan_3d.cpp:250-254) depends only on (depth, pose)
 * and changes the map additively, so every rank (one process per GPU) fuses its own frames into its own map and ONE
 * exchange makes every map the sum of all:
 *   1. all-gather of the ranks' 4x4x4-block ids, sorted union (identical on every rank);
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

struct dev_buf {
    void* p = nullptr;
    ~dev_buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
};

int read_status(gsdf_ctx* c) {
    gsdf_dev_state s;
    HIP_TRY(hipMemcpyAsync(&s, c->st, sizeof(s), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (s.status & GSDF_STATUS_TABLE_FULL) return gsdf_fail(GSDF_ERR_TABLE_FULL, "voxel hash table full (probe budget exhausted)");
    if (s.status & GSDF_STATUS_KEY_RANGE) return gsdf_fail(GSDF_ERR_KEY_RANGE, "voxel index outside the packable +-2^20 range");
    return GSDF_OK;
}

/* What every rank tells the others before anything is exchanged.  `err` makes local failures collective: a rank that could
 * not prepare (allocation, launch) still takes part in this all-gather, and then ALL ranks return an error together instead
 * of one leaving its peers inside a collective. */
struct merge_hdr {
    long long n_blocks;          /* blocks of this rank's map */
    long long frames;            /* Sdf::counter_ of this rank: frames it integrated */
    long long vis_words;         /* words per voxel of its vis_ bit-vectors, 0 = not enabled */
    long long err;               /* GSDF_ERR_* of its preparation, 0 = fine */
    unsigned long long token;    /* random: a rank finds its own position in the gathered list by it (the callback transport
                                    does not tell a rank its number) */
};

/* second agreement point: every rank reports whether its buffers for the exchange exist */
int agree(gsdf_ctx* c, transport& tr, long long* scratch_dev /* R + 1 words */, int local_rc, const char* what) {
    const int R = tr.nranks;
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

int merge_impl(gsdf_ctx* c, transport& tr, int64_t* n_blocks_out, int64_t* bytes_out) {
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = gsdf_flush_pending(c)) return rc;            /* the last frame's fusion may still wait for a successor (gsdf_update_dev) */
    if (c->merged && tr.nranks > 1)
        return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: this map already holds the sum of all ranks (the exchange is one-shot: "
                                           "a second one would count every rank's frames again); gsdf_reset starts over");
    const size_t cap = c->n_slots / GSDF_BLOCK_VOX;
    const int R = tr.nranks;
    /* 1. this rank's block ids and frame count.  Failures up to the first all-gather travel in the header. */
    dev_buf local, hdr_dev, scratch;
    unsigned long long n_local = 0;
    gsdf_dev_state st_host;
    std::memset(&st_host, 0, sizeof(st_host));
    auto prepare = [&]() -> int {
        HIP_TRY(local.alloc(cap * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
        gsdf_launch_block_keys(c->stream, c->tab, cap, (unsigned long long*)local.p, c->counter, (long long)cap);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&n_local, c->counter, sizeof(n_local), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&st_host, c->st, sizeof(st_host), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return GSDF_OK;
    };
    const int prep = prepare();
    if (prep) n_local = 0;
    /* two small buffers are needed to talk at all: without them this rank cannot even report its failure */
    HIP_TRY(hdr_dev.alloc((size_t)(R + 1) * sizeof(merge_hdr)));
    HIP_TRY(scratch.alloc((size_t)(R + 1) * sizeof(long long)));
    merge_hdr* hd = (merge_hdr*)hdr_dev.p;
    const merge_hdr mine = { (long long)n_local, (long long)st_host.frames, (long long)(c->vis ? c->vis_words : 0), (long long)prep,
                             random_token(c) };
    HIP_TRY(hipMemcpyAsync(hd + R, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
    int rc = tr.allgather(c, hd + R, hd, sizeof(merge_hdr));
    if (rc) return rc;
    std::vector<merge_hdr> hdr((size_t)R);
    HIP_TRY(hipMemcpyAsync(hdr.data(), hd, (size_t)R * sizeof(merge_hdr), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (prep) return prep;
    /* from here on every decision is taken from the gathered headers, i.e. identically on all ranks */
    long long m = 1, frames_total = 0, frames_before = 0;
    int me = -1, dup = 0;
    for (int r = 0; r < R; ++r) {
        const merge_hdr& h = hdr[(size_t)r];
        if (h.err) return gsdf_fail((int)h.err, "gsdf_merge_allreduce: rank " + std::to_string(r) + " failed to list its blocks");
        if (h.n_blocks < 0 || (size_t)h.n_blocks > ((size_t)1 << 40) || h.frames < 0)
            return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: inconsistent header from rank " + std::to_string(r));
        if (h.vis_words != hdr[0].vis_words)
            return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: gsdf_enable_vis must be called with the same frame count on every rank (or on none)");
        for (int q = 0; q < r; ++q) dup |= hdr[(size_t)q].token == h.token;
        if (h.token == mine.token) me = r;
        m = std::max(m, h.n_blocks);
        frames_total += h.frames;
    }
    if (dup || me < 0) return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: rank tokens collide (retry)");   /* seen by all ranks */
    for (int r = 0; r < me; ++r) frames_before += hdr[(size_t)r].frames;
    const int vw = (int)hdr[0].vis_words;
    if (vw && frames_total > 32ll * vw)
        return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: the ranks integrated " + std::to_string(frames_total) +
                                           " frames, gsdf_enable_vis reserved " + std::to_string(32ll * vw) + " bits per voxel");
    /* 2. the ranks' id lists, padded to the longest one */
    dev_buf padded, all;
    auto alloc_lists = [&]() -> int {
        HIP_TRY(padded.alloc((size_t)m * sizeof(unsigned long long)));
        HIP_TRY(all.alloc((size_t)m * (size_t)R * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(padded.p, 0xFF, (size_t)m * sizeof(unsigned long long), c->stream));
        if (n_local) HIP_TRY(hipMemcpyAsync(padded.p, local.p, (size_t)n_local * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
        return GSDF_OK;
    };
    rc = agree(c, tr, (long long*)scratch.p, alloc_lists(), "allocate its id list");
    if (rc) return rc;
    rc = tr.allgather(c, padded.p, all.p, (size_t)m * sizeof(unsigned long long));
    if (rc) return rc;
    /* sorted union: a few 10^4 .. 10^5 ids (8 B each), host sort; every rank computes the same list */
    std::vector<unsigned long long> ids((size_t)m * (size_t)R);
    HIP_TRY(hipMemcpyAsync(ids.data(), all.p, ids.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> uni;
    uni.reserve(ids.size());
    for (int r = 0; r < R; ++r) uni.insert(uni.end(), ids.begin() + (size_t)r * m, ids.begin() + (size_t)r * m + (size_t)hdr[(size_t)r].n_blocks);
    std::sort(uni.begin(), uni.end());
    uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
    const size_t nu = uni.size();
    const size_t vis_count = nu * GSDF_BLOCK_VOX * (size_t)vw;
    if (n_blocks_out) *n_blocks_out = (int64_t)nu;
    if (bytes_out) *bytes_out = (int64_t)(nu * GSDF_BLOCK_VOX * 5 * sizeof(float) + vis_count * sizeof(uint32_t));
    /* 3. pack, all-reduce, unpack -- the sums, then the vis_ bit-vectors */
    dev_buf union_dev, dense, dense_vis;
    auto alloc_dense = [&]() -> int {
        HIP_TRY(union_dev.alloc(nu * sizeof(unsigned long long)));
        HIP_TRY(dense.alloc(nu * GSDF_BLOCK_VOX * 5 * sizeof(float)));
        if (vw) HIP_TRY(dense_vis.alloc(vis_count * sizeof(uint32_t)));
        if (nu) HIP_TRY(hipMemcpyAsync(union_dev.p, uni.data(), nu * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
        return GSDF_OK;
    };
    rc = agree(c, tr, (long long*)scratch.p, alloc_dense(), "allocate the exchange buffers");
    if (rc) return rc;
    if (nu) {
        const unsigned long long* uk = (const unsigned long long*)union_dev.p;
        gsdf_launch_pack_blocks(c->stream, c->tab, uk, (long long)nu, (float*)dense.p);
        if (vw) gsdf_launch_pack_vis(c->stream, c->tab, c->vis, vw, frames_before, uk, (long long)nu, (uint32_t*)dense_vis.p);
        HIP_TRY(hipGetLastError());
        rc = tr.allreduce_sum_f32(c, (float*)dense.p, nu * GSDF_BLOCK_VOX * 5);
        if (rc) return rc;
        if (vw) {
            rc = tr.allreduce_or_u32(c, (uint32_t*)dense_vis.p, vis_count);
            if (rc) return rc;
        }
        c->occ_dirty = true;
        gsdf_launch_unpack_blocks(c->stream, c->tab, uk, (long long)nu, (const float*)dense.p, c->st);
        if (vw) gsdf_launch_unpack_vis(c->stream, c->tab, c->vis, vw, uk, (long long)nu, (const uint32_t*)dense_vis.p);
    }
    /* Sdf::counter_ of the merged map: the frames of all ranks (frame f of rank r is integrated frame frames_before(r) + f) */
    gsdf_launch_set_frames(c->stream, c->st, frames_total);
    HIP_TRY(hipGetLastError());
    c->merged = R > 1;
    return read_status(c);                                    /* synchronises: the buffers above may go */
}

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
