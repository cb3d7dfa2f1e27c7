#include "shm_collective.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <thread>
#include <vector>

namespace {
/* the rendezvous segment: a counting barrier that can be ABORTED.  A rank that fails (or a barrier that times out) raises
 * `abort`; every rank polling in a barrier then returns false instead of waiting for a peer that will never arrive. */
struct Segment {
    std::atomic<int> abort;
    std::atomic<unsigned> count;
    std::atomic<unsigned> gen;
};
static_assert(std::atomic<unsigned>::is_always_lock_free, "process-shared atomics must be lock-free");
std::string seg_name(const std::string& name) { return "/" + name + ".seg"; }
double barrier_timeout_s() {
    const char* e = getenv("GSDF_SHM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 300.0;
}
}

bool ShmCollective::create(const std::string& name, int nranks) {
    (void)nranks;
    const int fd = shm_open(seg_name(name).c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0) return false;
    if (ftruncate(fd, sizeof(Segment)) != 0) { close(fd); return false; }
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return false;
    Segment* s = new (p) Segment;
    s->abort.store(0); s->count.store(0); s->gen.store(0);
    munmap(p, sizeof(Segment));
    return true;
}

void ShmCollective::destroy(const std::string& name) { shm_unlink(seg_name(name).c_str()); }

/* raise the abort flag of a segment from outside a rank (the launcher, when a rank died) */
void ShmCollective::abort_all(const std::string& name) {
    const int fd = shm_open(seg_name(name).c_str(), O_RDWR, 0600);
    if (fd < 0) return;
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return;
    ((Segment*)p)->abort.store(1);
    munmap(p, sizeof(Segment));
}

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

void ShmCollective::abort() { if (seg_) ((Segment*)seg_)->abort.store(1); }
bool ShmCollective::aborted() const { return seg_ && ((Segment*)seg_)->abort.load() != 0; }

bool ShmCollective::barrier() {
    if (!seg_) return false;
    Segment* s = (Segment*)seg_;
    if (s->abort.load()) return false;
    const unsigned gen = s->gen.load();
    if (s->count.fetch_add(1) + 1 == (unsigned)nranks_) {
        s->count.store(0);
        s->gen.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = barrier_timeout_s();
    while (s->gen.load() == gen) {
        if (s->abort.load()) return false;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            s->abort.store(1);                           /* a peer never arrived: release everybody else as well */
            return false;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
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
