#include "frame_pipeline.h"
#include "png16.h"

#include <algorithm>

FramePipeline::FramePipeline(gsdf_ctx* ctx, const ImageLoader* loader, std::vector<FrameEntry> entries, int W, int H, int threads,
                             int slots)
    : ctx_(ctx), loader_(loader), entries_(std::move(entries)), W_(W), H_(H) {
    /* a decoder thread delivers ~1000 frames/s (libdeflate inflate 0.3-0.7 ms + conversion); the tracked loop consumes 5-9 k */
    if (threads <= 0) threads = (int)std::min<unsigned>(8u, std::max(2u, std::thread::hardware_concurrency() / 2));
    if (slots <= 0) slots = std::min(2 * threads + 4, 20);      /* each slot is a page-locked + a device buffer: ~0.4 ms to create */
    slots = (int)std::min<size_t>((size_t)slots, std::max<size_t>(entries_.size(), 1));
    slots_.resize((size_t)slots);
    const int64_t bytes = (int64_t)W_ * H_ * (int64_t)sizeof(float);
    for (Slot& s : slots_) {
        void *h = nullptr, *d = nullptr;
        if (gsdf_host_alloc(ctx_, &h, bytes) != GSDF_OK || gsdf_dev_alloc(ctx_, &d, bytes) != GSDF_OK) {
            error_ = std::string("frame staging buffers: ") + gsdf_last_error();
            if (h) gsdf_host_free(ctx_, h);
            break;
        }
        s.host = (float*)h; s.dev = (float*)d;
    }
    png_warm_up();                                     /* the inflate library's dlopen belongs to the set-up, not to the first frame */
    /* ... and so does the creation of the copy stream and its first transfer (the DMA queue's one-time set-up: 15-20 ms that
     * the first next() used to wait for) */
    if (error_.empty() && !slots_.empty() && slots_[0].host && slots_[0].dev) {
        int64_t id = 0;
        slots_[0].host[0] = 0.f;
        if (gsdf_dev_upload_ahead(ctx_, slots_[0].dev, slots_[0].host, (int64_t)sizeof(float), &id) == GSDF_OK) (void)gsdf_upload_wait(ctx_, id);
    }
    if (error_.empty())
        for (int t = 0; t < threads; ++t) threads_.emplace_back(&FramePipeline::worker, this);
}

FramePipeline::~FramePipeline() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_free_.notify_all();
    cv_filled_.notify_all();
    for (std::thread& t : threads_) t.join();
    if (last_upload_ > 0) gsdf_upload_wait(ctx_, last_upload_);   /* copies started ahead and never delivered (early exit) */
    gsdf_sync(ctx_);                                   /* nothing on the stream reads the buffers any more */
    for (Slot& s : slots_) {
        if (s.host) gsdf_host_free(ctx_, s.host);
        if (s.dev) gsdf_dev_free(ctx_, s.dev);
    }
}

/* decoder: claim the next frame whose slot (frame % slots) is free, decode into the slot's pinned buffer */
void FramePipeline::worker() {
    for (;;) {
        size_t f;
        Slot* s;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_free_.wait(lk, [&] { return stop_ || next_decode_ >= entries_.size() || slots_[next_decode_ % slots_.size()].state == FREE; });
            if (stop_ || next_decode_ >= entries_.size()) { cv_free_.notify_all(); return; }
            f = next_decode_++;
            s = &slots_[f % slots_.size()];
            s->state = DECODING;
            s->frame = f;
        }
        cv_free_.notify_one();                         /* the frame after this one may have a free slot too: pass the baton */
        std::string err;
        const bool ok = loader_->decode_depth(entries_[f].depth_file, s->host, W_, H_, &err);
        {
            std::lock_guard<std::mutex> lk(mu_);
            s->state = ok ? FILLED : FAILED;
            if (!ok) s->error = err;                   /* reported when THIS frame is delivered, not before (the frames ahead of it are fine) */
        }
        cv_filled_.notify_one();                       /* one consumer; the other decoders have nothing to learn from this */
    }
}

/* slots whose frame the stream has finished with go back to the decoders (called with mu_ NOT held) */
void FramePipeline::reclaim(bool block_oldest) {
    int freed = 0;
    for (;;) {
        Slot* oldest = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (Slot& s : slots_)
                if (s.state == INFLIGHT && (!oldest || s.mark < oldest->mark)) oldest = &s;
        }
        if (!oldest) break;
        int reached = 0;
        if (block_oldest) { if (gsdf_mark_wait(ctx_, oldest->mark) != GSDF_OK) break; reached = 1; block_oldest = false; }
        else if (gsdf_mark_reached(ctx_, oldest->mark, &reached) != GSDF_OK || !reached) break;
        {
            std::lock_guard<std::mutex> lk(mu_);
            oldest->state = FREE;
        }
        ++freed;
    }
    for (int i = 0; i < freed; ++i) cv_free_.notify_one();      /* one decoder per slot that came back (they pass the baton on) */
}

/* The copies run AHEAD of the stream on the context's copy stream (gsdf_dev_upload_ahead): frames are copied in order as soon as
 * they are decoded, up to UPLOAD_AHEAD beyond the one being delivered, so a frame's depth image is in HBM before the kernels
 * of the frame before it have finished -- the stream itself never changes from kernels to a copy and back. */
bool FramePipeline::start_uploads() {
    constexpr size_t UPLOAD_AHEAD = 4;
    for (;;) {
        Slot* s = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (next_upload_ >= entries_.size() || next_upload_ > next_deliver_ + UPLOAD_AHEAD) return true;
            Slot& c = slots_[next_upload_ % slots_.size()];
            if (c.state != FILLED || c.frame != next_upload_) return true;       /* not decoded yet (or failed: reported at delivery) */
            s = &c;
        }
        int64_t id = 0;
        if (gsdf_dev_upload_ahead(ctx_, s->dev, s->host, (int64_t)W_ * H_ * (int64_t)sizeof(float), &id) != GSDF_OK) {
            error_ = std::string("upload: ") + gsdf_last_error();
            return false;
        }
        std::lock_guard<std::mutex> lk(mu_);
        s->upload = id;
        last_upload_ = id;
        s->state = UPLOADING;
        ++next_upload_;
    }
}

const float* FramePipeline::next(size_t* index) {
    if (!error_.empty() || next_deliver_ >= entries_.size()) return nullptr;
    const size_t f = next_deliver_;
    Slot& s = slots_[f % slots_.size()];
    for (;;) {
        reclaim(false);
        if (!start_uploads()) return nullptr;
        std::unique_lock<std::mutex> lk(mu_);
        if ((s.state == UPLOADING || s.state == FAILED) && s.frame == f) break;
        if (s.state == FILLED && s.frame == f) continue;   /* decoded a moment ago: start its copy */
        if (s.state == INFLIGHT) {                     /* the ring is full of frames the GPU still owns: wait for the oldest */
            lk.unlock();
            reclaim(true);
            continue;
        }
        cv_filled_.wait_for(lk, std::chrono::milliseconds(2));
    }
    if (s.state == FAILED) {
        std::lock_guard<std::mutex> lk(mu_);
        error_ = s.error;
        next_deliver_ = entries_.size();
        return nullptr;
    }
    if (gsdf_upload_wait(ctx_, s.upload) != GSDF_OK) {   /* normally long done: it was started frames ago */
        error_ = std::string("upload: ") + gsdf_last_error();
        return nullptr;
    }
    last_slot_ = (long)(f % slots_.size());
    ++next_deliver_;
    if (index) *index = f;
    /* the successor, if its copy has been started (it then is at most microseconds from done: the copy stream runs ahead) */
    next_ready_ = nullptr;
    if (f + 1 < entries_.size()) {
        int64_t up = 0;
        const float* d = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu_);
            const Slot& n = slots_[(f + 1) % slots_.size()];
            if (n.state == UPLOADING && n.frame == f + 1) { up = n.upload; d = n.dev; }
        }
        if (d && gsdf_upload_wait(ctx_, up) == GSDF_OK) next_ready_ = d;
    }
    return s.dev;
}

bool FramePipeline::submitted() {
    if (last_slot_ < 0) return false;
    Slot& s = slots_[(size_t)last_slot_];
    int64_t m = 0;
    if (gsdf_mark(ctx_, &m) != GSDF_OK) { error_ = std::string("mark: ") + gsdf_last_error(); return false; }
    {
        std::lock_guard<std::mutex> lk(mu_);
        s.mark = m;
        s.state = INFLIGHT;
    }
    last_slot_ = -1;
    return true;
}
