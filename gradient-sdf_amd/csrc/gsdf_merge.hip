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

#include <algorithm>
#include <cstring>
#include <mutex>

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

int merge_impl(gsdf_ctx* c, transport& tr, int64_t* n_blocks_out, int64_t* bytes_out) {
    HIP_TRY(hipSetDevice(c->device));
    const size_t cap = c->n_slots / GSDF_BLOCK_VOX;
    const int R = tr.nranks;
    /* 1. this rank's block ids */
    dev_buf local, counts_dev;
    HIP_TRY(local.alloc(cap * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
    gsdf_launch_block_keys(c->stream, c->tab, cap, (unsigned long long*)local.p, c->counter, (long long)cap);
    unsigned long long n_local = 0;
    HIP_TRY(hipMemcpyAsync(&n_local, c->counter, sizeof(n_local), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    /* the ranks' counts, then the id lists padded to the longest one */
    HIP_TRY(counts_dev.alloc((size_t)(R + 1) * sizeof(long long)));
    long long* cd = (long long*)counts_dev.p;
    const long long mine = (long long)n_local;
    HIP_TRY(hipMemcpyAsync(cd + R, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
    int rc = tr.allgather(c, cd + R, cd, sizeof(long long));
    if (rc) return rc;
    std::vector<long long> counts((size_t)R);
    HIP_TRY(hipMemcpyAsync(counts.data(), cd, (size_t)R * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    long long m = 1;
    for (long long v : counts) { if (v < 0 || (size_t)v > ((size_t)1 << 40)) return gsdf_fail(GSDF_ERR_INVALID, "gsdf_merge_allreduce: inconsistent block count from a rank"); m = std::max(m, v); }
    dev_buf padded, all;
    HIP_TRY(padded.alloc((size_t)m * sizeof(unsigned long long)));
    HIP_TRY(all.alloc((size_t)m * (size_t)R * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(padded.p, 0xFF, (size_t)m * sizeof(unsigned long long), c->stream));
    if (n_local) HIP_TRY(hipMemcpyAsync(padded.p, local.p, (size_t)n_local * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
    rc = tr.allgather(c, padded.p, all.p, (size_t)m * sizeof(unsigned long long));
    if (rc) return rc;
    /* sorted union: a few 10^4 .. 10^5 ids (8 B each), host sort; every rank computes the same list */
    std::vector<unsigned long long> ids((size_t)m * (size_t)R);
    HIP_TRY(hipMemcpyAsync(ids.data(), all.p, ids.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> uni;
    uni.reserve(ids.size());
    for (int r = 0; r < R; ++r) uni.insert(uni.end(), ids.begin() + (size_t)r * m, ids.begin() + (size_t)r * m + (size_t)counts[(size_t)r]);
    std::sort(uni.begin(), uni.end());
    uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
    const size_t nu = uni.size();
    if (n_blocks_out) *n_blocks_out = (int64_t)nu;
    if (bytes_out) *bytes_out = (int64_t)(nu * GSDF_BLOCK_VOX * 5 * sizeof(float));
    if (nu == 0) return GSDF_OK;
    /* 2.-4. pack, all-reduce, unpack */
    dev_buf union_dev, dense;
    HIP_TRY(union_dev.alloc(nu * sizeof(unsigned long long)));
    HIP_TRY(dense.alloc(nu * GSDF_BLOCK_VOX * 5 * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(union_dev.p, uni.data(), nu * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    gsdf_launch_pack_blocks(c->stream, c->tab, (const unsigned long long*)union_dev.p, (long long)nu, (float*)dense.p);
    HIP_TRY(hipGetLastError());
    rc = tr.allreduce_sum_f32(c, (float*)dense.p, nu * GSDF_BLOCK_VOX * 5);
    if (rc) return rc;
    gsdf_launch_unpack_blocks(c->stream, c->tab, (const unsigned long long*)union_dev.p, (long long)nu, (const float*)dense.p, c->st);
    HIP_TRY(hipGetLastError());
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
