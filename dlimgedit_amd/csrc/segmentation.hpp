// SegmentationImpl: per-image state of the Segment-Anything path.
// Counterpart of /root/reference/src/segmentation.{hpp,cpp} (SegmentationImpl, ResizeLongestSide):
// same life cycle (process once, query many masks), but the embedding lives in HBM.
#pragma once

#include "common.hpp"
#include "environment.hpp"

#include <dlimgedit/dlimgedit.h>

namespace dlimg {

struct Extent { int width = 0, height = 0; };
struct Point { int x = 0, y = 0; };
struct Region { Point top_left, bottom_right; };

int channel_bytes(int channels);          // 4 for bgra/argb (reference: dlimgedit.impl.hpp:15)
int scale_coord(int coord, float scale);  // reference: segmentation.cpp:26

// Longest-side-to-1024 geometry (reference: segmentation.cpp:58-74).  The pixel resampling itself is
// a device kernel here; this struct only keeps the numbers needed later for prompts and masks.
struct ResizeLongestSide {
    Extent original;
    Extent resized;
    float scale = 1.f;

    explicit ResizeLongestSide(int max_side = kImageSize) : max_side_(max_side) {}
    void set(Extent image);
    Point transform(Point p) const { return Point{scale_coord(p.x, scale), scale_coord(p.y, scale)}; }

  private:
    int max_side_;
};

// Packs one prompt the way SegmentationImpl::compute_mask does (reference: segmentation.cpp:135-152).
void pack_prompt(ResizeLongestSide const& rs, Point const* point, Region const* region, float coords[4],
                 float labels[2]);

class SegmentationImpl {
  public:
    explicit SegmentationImpl(EnvironmentImpl& env);

    void process(dlimg_ImageView const& image);
    // Batched variant: segs[i] receives the embedding of images[i].
    static void process_batch(EnvironmentImpl& env, SegmentationImpl* const* segs, dlimg_ImageView const* images,
                              int count);

    void compute_mask(Point const* point, Region const* region, uint8_t* const out_masks[3],
                      float out_accuracy[3]) const;
    static void compute_mask_batch(SegmentationImpl const* const* segs, int count, int const* points,
                                   int const* regions, uint8_t* const* out_masks);

    // Device-output form of compute_mask_batch: mask i is produced on the GPU that holds segs[i]'s embedding
    // and lands at dev_out + offset_i in the memory of HIP device `root_device` (offset_i = sum of width*height of the
    // entries before it; also returned in out_offsets when given).  Returns when every mask is in place.
    static void compute_mask_batch_device(SegmentationImpl const* const* segs, int count, int const* points,
                                          int const* regions, int root_device, uint8_t* dev_out, size_t* out_offsets);

    Extent extent() const { return image_size_.original; }
    ResizeLongestSide const& geometry() const { return image_size_; }
    float const* embedding() const { settle(); return embedding_; }       // complete when this returns
    ~SegmentationImpl();
    SegmentationImpl(SegmentationImpl const&) = delete;
    SegmentationImpl& operator=(SegmentationImpl const&) = delete;
    EnvironmentImpl& environment() const { return env_; }
    int replica() const { return replica_; }      // which entry of the environment's device list holds the embedding
    void set_geometry(Extent e) { image_size_.set(e); }
    float* embedding_storage(int replica);

  private:
    EnvironmentImpl& env_;
    int replica_ = 0;
    ResizeLongestSide image_size_;
    float* embedding_ = nullptr;        // [4096][256] fp32, resident on the replica's GPU
    std::shared_ptr<EmbeddingPool> pool_;   // where embedding_ came from and goes back to

    // process() of ONE image returns once its encoder pass is enqueued (the pixels have been copied by then; argument
    // errors have been reported).  The pass is waited for by the first call that needs its result: compute_mask queues its
    // decoder on the same lane behind the encoder and waits once, for both; everything else settles first.  A pass that
    // reported non-finite values makes every query of this handle fail with the message process() would have thrown.
    // DLIMGEDIT_SYNC_PROCESS=1: process() waits itself, as in the reference (Ort::Session::Run).
    mutable std::mutex pending_mutex_;
    mutable std::shared_ptr<SamModel::DeferredPass> pending_;
    mutable std::atomic<bool> invalid_{false};
    std::shared_ptr<SamModel::DeferredPass> pending() const;
    void settle() const;                    // waits for a deferred pass; throws when the embedding is not usable
    void forget_pending() noexcept;         // the same without the verdict: the embedding is about to be replaced or freed
};

// Validates an image view the way the entry points need it; throws on nonsense.
void check_image(dlimg_ImageView const& image);

}  // namespace dlimg
