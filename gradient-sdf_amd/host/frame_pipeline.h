/*
 * FramePipeline -- keeps the GPU fed from a directory of depth PNGs ("async H2D staging of frames").
 *
 * The reference's loop is load -> optimize -> update, all on one thread (main_scan_3d.cpp:208-281); there the CPU
 * fusion (~0.6 s per frame) hides the PNG decode.  Here a frame costs ~130 us on the GPU, so the decode (zlib inflate
 * + unfilter + scale to metres, ~1-2 ms per 640x480 frame) runs on a pool of host threads, each frame into a
 * page-locked buffer; the consumer thread enqueues the host->HBM copy and the frame's kernels on the context's stream
 * and never waits for the GPU (slots are recycled through gsdf_mark).  Frames are delivered strictly in order.
 */
#ifndef GSDF_HOST_FRAME_PIPELINE_H_
#define GSDF_HOST_FRAME_PIPELINE_H_

#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gsdf.h"
#include "img_loader.h"

struct FrameEntry {
    std::string depth_file;      /* relative to the loader's directory */
    std::string timestamp;       /* the loader's depth_timestamp() of this frame */
};

class FramePipeline {
public:
    /* entries: the frames to deliver, in order.  slots: frames in flight (decoded or on the GPU); threads: decoders. */
    FramePipeline(gsdf_ctx* ctx, const ImageLoader* loader, std::vector<FrameEntry> entries, int W, int H, int threads = 0,
                  int slots = 0);
    ~FramePipeline();
    FramePipeline(const FramePipeline&) = delete;
    FramePipeline& operator=(const FramePipeline&) = delete;

    /* Next frame: waits for its decode, enqueues the host->HBM copy, returns the DEVICE pointer of its depth image
     * (nullptr: no more frames, or the frame could not be read -- error() tells).  index = position in `entries`. */
    const float* next(size_t* index);
    /* The frame AFTER the one the last next() returned, if its copy into HBM is known to be over (it was started frames ago; a
     * host-side look, no wait) -- else nullptr.  What gsdf_hint_next_depth_dev / gsdf_track_and_fuse_ahead_dev want to be told:
     * the current frame's fusion launch then computes that frame's normals in its tail.  Stays valid until the next next(). */
    const float* next_ready() const { return next_ready_; }
    /* The kernels that read the frame returned by the last next() have been enqueued: its slot may be recycled once the
     * stream has passed this point. */
    bool submitted();
    const std::string& error() const { return error_; }
    size_t size() const { return entries_.size(); }
    const FrameEntry& entry(size_t i) const { return entries_[i]; }

private:
    enum State { FREE, DECODING, FILLED, FAILED, UPLOADING, INFLIGHT };
    struct Slot { float* host = nullptr; float* dev = nullptr; State state = FREE; size_t frame = 0; int64_t mark = 0; int64_t upload = 0; std::string error; };
    bool start_uploads();        /* host->HBM copies of the next few decoded frames, on the copy stream, in frame order */
    void worker();
    void reclaim(bool block_oldest);

    gsdf_ctx* ctx_;
    const ImageLoader* loader_;
    std::vector<FrameEntry> entries_;
    int W_, H_;
    std::vector<Slot> slots_;
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_free_;     /* decoders wait here for the slot of the next frame to come back from the GPU */
    std::condition_variable cv_filled_;   /* the consumer waits here for its frame's decode */
    size_t next_decode_ = 0;     /* next frame a decoder may claim */
    size_t next_deliver_ = 0;    /* next frame next() returns */
    size_t next_upload_ = 0;     /* next frame whose copy has not been started */
    int64_t last_upload_ = 0;    /* id of the copy started last */
    long last_slot_ = -1;
    const float* next_ready_ = nullptr;
    bool stop_ = false;
    std::string error_;          /* written and read by the consumer thread only (decode errors wait in their slot) */
};

#endif
