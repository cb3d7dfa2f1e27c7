/*
 * fake_rccl.c -- TEST DOUBLE for RCCL (test infrastructure only; never shipped, never loaded by the product by itself).
 *
 * The pool this repository is tested on offers ONE GPU per call, and RCCL refuses two ranks on one device -- so
 * gsdf_merge_allreduce's RCCL-typed path (csrc/gsdf_merge.hip: rccl_transport -- ncclAllGather of ncclInt8, ncclAllReduce of
 * ncclFloat32 and ncclUint32 on the context's stream, the header / token / agree() scheme around them) never ran with peers.
 * libgsdf.so resolves the nccl* entry points with dlsym(RTLD_DEFAULT, ...) from whatever the process already holds
 * (gsdf_merge.hip:50-73); with LD_PRELOAD=tests/libfake_rccl.so that is THIS file: the seven entry points libgsdf uses,
 * implemented for N processes that share device 0.  tests/test_parallel.py starts 2 and 8 such processes which call
 * gsdf_merge_allreduce(ctx, comm) ITSELF.
 *
 * Mechanism: a POSIX shared-memory segment per communicator (name carried in the ncclUniqueId): a header with a sense-reversing
 * barrier and one staging slot per rank.  A collective copies the caller's DEVICE buffer to its slot on the stream it was given
 * (hipMemcpyAsync + hipStreamSynchronize: a synchronous RCCL is a valid RCCL), meets the peers at the barrier, combines the
 * slots in rank order (every rank computes the same bytes) and copies the result back to the device.  Buffers larger than a
 * slot go through in chunks.  Barriers time out (60 s) into ncclSystemError instead of hanging a test.
 *
 * Build (tests/test_parallel.py does it): gcc -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.c -o libfake_rccl.so -lrt
 * -- WITHOUT a DT_NEEDED on libamdhip64: like libgsdf.so it binds to the HIP runtime the process already holds.
 */
#define _GNU_SOURCE
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define FAKE_SLOT_BYTES ((size_t)8 << 20)          /* staging per rank; larger buffers are chunked */
#define FAKE_MAGIC 0x4652434Cu                       /* "FRCL" */
#define FAKE_TIMEOUT_S 60.0

typedef struct {
    volatile uint32_t magic;
    volatile uint32_t nranks;
    volatile uint32_t arrived;
    volatile uint32_t generation;
    volatile uint32_t abort_flag;
    /* what the ranks asked for in the collective in flight: a mismatch is a bug in the caller and is reported, not reduced */
    volatile uint64_t want_bytes[64];
    volatile uint32_t want_kind[64];
    /* statistics (rank 0's calls): all-gathers by Int8 bytes, all-reduces by type */
    volatile uint64_t n_allgather, n_allreduce_f32, n_allreduce_u32, bytes_total;
} fake_hdr;

struct ncclComm {
    int rank, nranks;
    char name[64];
    size_t map_bytes;
    fake_hdr* h;
    unsigned char* slots;
    void* tmp;                                        /* FAKE_SLOT_BYTES of host scratch for the reduced chunk */
};

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static ncclResult_t barrier(struct ncclComm* c) {
    fake_hdr* h = c->h;
    const uint32_t gen = __atomic_load_n(&h->generation, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&h->arrived, 1u, __ATOMIC_ACQ_REL) == (uint32_t)c->nranks) {
        __atomic_store_n(&h->arrived, 0u, __ATOMIC_RELAXED);
        __atomic_add_fetch(&h->generation, 1u, __ATOMIC_RELEASE);
        return ncclSuccess;
    }
    const double t0 = now_s();
    while (__atomic_load_n(&h->generation, __ATOMIC_ACQUIRE) == gen) {
        if (__atomic_load_n(&h->abort_flag, __ATOMIC_RELAXED)) return ncclSystemError;
        if (now_s() - t0 > FAKE_TIMEOUT_S) { __atomic_store_n(&h->abort_flag, 1u, __ATOMIC_RELAXED); return ncclSystemError; }
        sched_yield();
    }
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: HIP error";
    case ncclSystemError: return "fake_rccl: a peer did not arrive (timeout or abort)";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    case ncclInvalidUsage: return "fake_rccl: the ranks disagree about a collective";
    default: return "fake_rccl: error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static unsigned counter = 0;
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/gsdf_fake_rccl_%d_%u_%lx", (int)getpid(), counter++, (unsigned long)(now_s() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof(*c));
    if (!c) return ncclSystemError;
    c->rank = rank; c->nranks = nranks;
    memcpy(c->name, id.internal, sizeof(c->name) - 1);
    c->map_bytes = 4096 + (size_t)nranks * FAKE_SLOT_BYTES;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { if (fd >= 0) close(fd); free(c); return ncclSystemError; }
    void* p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { free(c); return ncclSystemError; }
    c->h = (fake_hdr*)p;
    c->slots = (unsigned char*)p + 4096;
    c->tmp = malloc(FAKE_SLOT_BYTES);
    if (!c->tmp) { munmap(p, c->map_bytes); free(c); return ncclSystemError; }
    if (rank == 0) { c->h->nranks = (uint32_t)nranks; __atomic_store_n(&c->h->magic, FAKE_MAGIC, __ATOMIC_RELEASE); }
    const double t0 = now_s();
    while (__atomic_load_n(&c->h->magic, __ATOMIC_ACQUIRE) != FAKE_MAGIC) {      /* a fresh segment is zero-filled */
        if (now_s() - t0 > FAKE_TIMEOUT_S) { munmap(p, c->map_bytes); free(c->tmp); free(c); return ncclSystemError; }
        sched_yield();
    }
    if (c->h->nranks != (uint32_t)nranks) { munmap(p, c->map_bytes); free(c->tmp); free(c); return ncclInvalidArgument; }
    ncclResult_t r = barrier(c);
    if (r != ncclSuccess) { munmap(p, c->map_bytes); free(c->tmp); free(c); return r; }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    (void)barrier(comm);                                /* nobody unlinks a segment a peer still reads */
    if (comm->rank == 0) shm_unlink(comm->name);
    munmap((void*)comm->h, comm->map_bytes);
    free(comm->tmp);
    free(comm);
    return ncclSuccess;
}

static size_t type_size(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

/* every rank announces (kind, bytes) of the collective it entered; all must agree */
static ncclResult_t agree(struct ncclComm* c, uint32_t kind, uint64_t bytes) {
    c->h->want_kind[c->rank] = kind;
    c->h->want_bytes[c->rank] = bytes;
    ncclResult_t r = barrier(c);
    if (r != ncclSuccess) return r;
    int bad = 0;
    for (int i = 0; i < c->nranks; ++i) bad |= c->h->want_kind[i] != kind || c->h->want_bytes[i] != bytes;
    r = barrier(c);                                     /* nobody overwrites its announcement before everybody has looked */
    if (r != ncclSuccess) return r;
    return bad ? ncclInvalidUsage : ncclSuccess;
}

#define HIPCHK(e) do { if ((e) != hipSuccess) { __atomic_store_n(&comm->h->abort_flag, 1u, __ATOMIC_RELAXED); return ncclUnhandledCudaError; } } while (0)

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    const size_t ts = type_size(datatype);
    if (!comm || !ts || (sendcount && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
    const size_t bytes = sendcount * ts;
    ncclResult_t r = agree(comm, 0x100u | (uint32_t)datatype, bytes);
    if (r != ncclSuccess) return r;
    if (comm->rank == 0) { comm->h->n_allgather++; comm->h->bytes_total += bytes; }
    for (size_t off = 0; off < bytes; off += FAKE_SLOT_BYTES) {
        const size_t n = bytes - off < FAKE_SLOT_BYTES ? bytes - off : FAKE_SLOT_BYTES;
        HIPCHK(hipMemcpyAsync(comm->slots + (size_t)comm->rank * FAKE_SLOT_BYTES, (const char*)sendbuff + off, n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if ((r = barrier(comm)) != ncclSuccess) return r;
        for (int p = 0; p < comm->nranks; ++p)
            HIPCHK(hipMemcpyAsync((char*)recvbuff + (size_t)p * bytes + off, comm->slots + (size_t)p * FAKE_SLOT_BYTES, n, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if ((r = barrier(comm)) != ncclSuccess) return r;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
    if (!comm || (count && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
    if (op != ncclSum || (datatype != ncclFloat32 && datatype != ncclUint32)) return ncclInvalidArgument;   /* what libgsdf uses */
    const size_t bytes = count * 4;
    ncclResult_t r = agree(comm, 0x200u | (uint32_t)datatype, bytes);
    if (r != ncclSuccess) return r;
    if (comm->rank == 0) { if (datatype == ncclFloat32) comm->h->n_allreduce_f32++; else comm->h->n_allreduce_u32++; comm->h->bytes_total += bytes; }
    for (size_t off = 0; off < bytes; off += FAKE_SLOT_BYTES) {
        const size_t n = bytes - off < FAKE_SLOT_BYTES ? bytes - off : FAKE_SLOT_BYTES, ne = n / 4;
        HIPCHK(hipMemcpyAsync(comm->slots + (size_t)comm->rank * FAKE_SLOT_BYTES, (const char*)sendbuff + off, n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if ((r = barrier(comm)) != ncclSuccess) return r;
        if (datatype == ncclFloat32) {                  /* rank order: every rank computes the same floats */
            float* acc = (float*)comm->tmp;
            memcpy(acc, comm->slots, n);
            for (int p = 1; p < comm->nranks; ++p) {
                const float* s = (const float*)(comm->slots + (size_t)p * FAKE_SLOT_BYTES);
                for (size_t i = 0; i < ne; ++i) acc[i] += s[i];
            }
        } else {
            uint32_t* acc = (uint32_t*)comm->tmp;
            memcpy(acc, comm->slots, n);
            for (int p = 1; p < comm->nranks; ++p) {
                const uint32_t* s = (const uint32_t*)(comm->slots + (size_t)p * FAKE_SLOT_BYTES);
                for (size_t i = 0; i < ne; ++i) acc[i] += s[i];
            }
        }
        HIPCHK(hipMemcpyAsync((char*)recvbuff + off, comm->tmp, n, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if ((r = barrier(comm)) != ncclSuccess) return r;
    }
    return ncclSuccess;
}

/* test hook (not an RCCL entry): the collectives rank 0 has seen on this communicator */
void fake_rccl_stats(ncclComm_t comm, uint64_t out[4]) {
    out[0] = comm->h->n_allgather; out[1] = comm->h->n_allreduce_f32; out[2] = comm->h->n_allreduce_u32; out[3] = comm->h->bytes_total;
}
