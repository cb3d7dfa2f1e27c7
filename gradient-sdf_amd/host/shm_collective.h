/*
 * ShmCollective -- the two collectives gsdf_merge_allreduce_with needs (all-gather of bytes, all-reduce sum of floats)
 * between the processes of one node through files in /dev/shm and a process-shared barrier.  It is the transport of
 * `Scan3D --gpus N --transport shm`: N ranks that share ONE GPU (a one-GPU test box, where RCCL refuses two ranks on a
 * device), or a node without RCCL.  With one GPU per rank the transport is RCCL (gsdf_merge_allreduce).
 */
#ifndef GSDF_HOST_SHM_COLLECTIVE_H_
#define GSDF_HOST_SHM_COLLECTIVE_H_

#include <cstdint>
#include <string>

#include "../../include/gsdf.h"

class ShmCollective {
public:
    /* the launcher creates the rendezvous segment BEFORE it starts the ranks ... */
    static bool create(const std::string& name, int nranks);
    static void destroy(const std::string& name);
    /* ... and every rank attaches to it */
    ShmCollective(const std::string& name, int nranks, int rank);
    ~ShmCollective();
    bool ok() const { return seg_ != nullptr; }
    gsdf_collective ops();                       /* callbacks bound to this object */
    /* false = aborted: a rank raised the flag (it failed), or a peer did not arrive within GSDF_SHM_TIMEOUT_S (default 300 s) */
    bool barrier();
    void abort();                                /* this rank gives up: release every rank that waits in a barrier */
    bool aborted() const;
    static void abort_all(const std::string& name);   /* the same from the launcher */

private:
    static int allgather_cb(void* user, const void* send, void* recv, int64_t bytes);
    static int allreduce_cb(void* user, float* buf, int64_t n);
    std::string file(long seq, int rank) const;
    std::string name_;
    int nranks_, rank_;
    long seq_ = 0;
    void* seg_ = nullptr;
};

#endif
