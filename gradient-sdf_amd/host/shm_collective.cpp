#include "shm_collective.h"

#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

namespace {
struct Segment { pthread_barrier_t barrier; };
std::string seg_name(const std::string& name) { return "/" + name + ".seg"; }
}

bool ShmCollective::create(const std::string& name, int nranks) {
    const int fd = shm_open(seg_name(name).c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0) return false;
    if (ftruncate(fd, sizeof(Segment)) != 0) { close(fd); return false; }
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return false;
    pthread_barrierattr_t at;
    pthread_barrierattr_init(&at);
    pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
    const int rc = pthread_barrier_init(&((Segment*)p)->barrier, &at, (unsigned)nranks);
    pthread_barrierattr_destroy(&at);
    munmap(p, sizeof(Segment));
    return rc == 0;
}

void ShmCollective::destroy(const std::string& name) { shm_unlink(seg_name(name).c_str()); }

ShmCollective::ShmCollective(const std::string& name, int nranks, int rank) : name_(name), nranks_(nranks), rank_(rank) {
    const int fd = shm_open(seg_name(name).c_str(), O_RDWR, 0600);
    if (fd < 0) return;
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p != MAP_FAILED) seg_ = p;
}

ShmCollective::~ShmCollective() {
    if (seg_) munmap(seg_, sizeof(Segment));
}

bool ShmCollective::barrier() {
    if (!seg_) return false;
    const int rc = pthread_barrier_wait(&((Segment*)seg_)->barrier);
    return rc == 0 || rc == PTHREAD_BARRIER_SERIAL_THREAD;
}

std::string ShmCollective::file(long seq, int rank) const {
    return "/dev/shm/" + name_ + "." + std::to_string(seq) + "." + std::to_string(rank);
}

/* every rank publishes its buffer as a file, barrier, every rank reads what it needs, barrier, files go */
int ShmCollective::allgather_cb(void* user, const void* send, void* recv, int64_t bytes) {
    ShmCollective* s = (ShmCollective*)user;
    const long seq = s->seq_++;
    {
        std::ofstream f(s->file(seq, s->rank_), std::ios::binary);
        if (!f.write((const char*)send, bytes)) return 1;
    }
    if (!s->barrier()) return 1;
    for (int r = 0; r < s->nranks_; ++r) {
        std::ifstream f(s->file(seq, r), std::ios::binary);
        if (!f.read((char*)recv + (size_t)r * (size_t)bytes, bytes)) return 1;
    }
    if (!s->barrier()) return 1;
    std::remove(s->file(seq, s->rank_).c_str());
    return 0;
}

int ShmCollective::allreduce_cb(void* user, float* buf, int64_t n) {
    ShmCollective* s = (ShmCollective*)user;
    const long seq = s->seq_++;
    {
        std::ofstream f(s->file(seq, s->rank_), std::ios::binary);
        if (!f.write((const char*)buf, n * (int64_t)sizeof(float))) return 1;
    }
    if (!s->barrier()) return 1;
    std::vector<float> other((size_t)n);
    std::memset(buf, 0, (size_t)n * sizeof(float));
    for (int r = 0; r < s->nranks_; ++r) {                    /* rank order: every rank computes the same sums */
        std::ifstream f(s->file(seq, r), std::ios::binary);
        if (!f.read((char*)other.data(), n * (int64_t)sizeof(float))) return 1;
        for (int64_t i = 0; i < n; ++i) buf[i] += other[(size_t)i];
    }
    if (!s->barrier()) return 1;
    std::remove(s->file(seq, s->rank_).c_str());
    return 0;
}

gsdf_collective ShmCollective::ops() {
    gsdf_collective c;
    c.allgather = &ShmCollective::allgather_cb;
    c.allreduce_sum_f32 = &ShmCollective::allreduce_cb;
    c.user = this;
    c.nranks = nranks_;
    return c;
}
