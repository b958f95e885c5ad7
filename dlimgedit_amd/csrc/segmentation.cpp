#include "segmentation.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace dlimg {

int channel_bytes(int channels) { return channels > 4 ? 4 : channels; }

int scale_coord(int coord, float scale) { return int(float(coord) * scale + 0.5f); }

void ResizeLongestSide::set(Extent image) {
    original = image;
    scale = float(max_side_) / float(std::max(image.width, image.height));
    resized = image;
    if (scale != 1) resized = Extent{scale_coord(image.width, scale), scale_coord(image.height, scale)};
}

void pack_prompt(ResizeLongestSide const& rs, Point const* point, Region const* region, float coords[4],
                 float labels[2]) {
    DLIMG_ASSERT((point != nullptr) != (region != nullptr));
    auto set = [&](int index, Point p, int label) {
        Point t = rs.transform(p);
        coords[index * 2 + 0] = float(t.x);
        coords[index * 2 + 1] = float(t.y);
        labels[index] = float(label);
    };
    if (point) {
        set(0, *point, 1);
        set(1, Point{0, 0}, -1);     // padding point of the exported decoder graph
    } else {
        set(0, region->top_left, 2);
        set(1, region->bottom_right, 3);
    }
}

void check_image(dlimg_ImageView const& image) {
    if (!image.pixels) throw Exception("Image has no pixel data");
    if (image.width <= 0 || image.height <= 0) throw Exception("Image extent must be positive");
    const int c = image.channels;
    if (!(c == 1 || c == 3 || c == 4 || c == 5 || c == 6))
        throw Exception("Unsupported channel order [" + std::to_string(c) + "]");
    DLIMG_ASSERT(image.stride >= image.width * channel_bytes(c));
}

SegmentationImpl::SegmentationImpl(EnvironmentImpl& env) : env_(env) {
    (void)env.lane_count();     // loads the model (and reports a missing weight file) at the same point as the reference
}

float* SegmentationImpl::embedding_storage() {
    embedding_.reserve((size_t)kTokens * kEmbedDim);
    return embedding_.get();
}

namespace {

// Brings one host image into slot `slot` of the model's patch matrix (resizing on the device when
// the longest side is not 1024; reference: ResizeLongestSide::resize, segmentation.cpp:60-70).
void stage_image(SamModel& model, int slot, int batch, dlimg_ImageView const& image, ResizeLongestSide const& rs) {
    if (rs.scale != 1) {
        model.upload_and_resize_image(slot, batch, image.pixels, image.width, image.height, image.stride,
                                      image.channels, rs.resized.width, rs.resized.height);
    } else {
        model.upload_image(slot, batch, image.pixels, image.width, image.height, image.stride, image.channels);
    }
}

}  // namespace

void SegmentationImpl::process(dlimg_ImageView const& image) {
    SegmentationImpl* self = this;
    process_batch(env_, &self, &image, 1);
}

void SegmentationImpl::process_batch(EnvironmentImpl& env, SegmentationImpl* const* segs, dlimg_ImageView const* images,
                                     int count) {
    if (count <= 0) return;
    SamModel& model = env.sam_model();
    for (int i = 0; i < count; ++i) {
        check_image(images[i]);
        segs[i]->image_size_.set(Extent{images[i].width, images[i].height});
    }
    std::lock_guard<std::mutex> lock(model.mutex());
    HIP_CHECK(hipSetDevice(model.device()));
    for (int i = 0; i < count; ++i) stage_image(model, i, count, images[i], segs[i]->image_size_);
    model.encode(count);
    const size_t n = (size_t)kTokens * kEmbedDim;
    for (int i = 0; i < count; ++i) {
        HIP_CHECK(hipMemcpyAsync(segs[i]->embedding_storage(), model.embeddings() + i * n, n * sizeof(float),
                                 hipMemcpyDeviceToDevice, model.stream()));
    }
    // process() is synchronous in the reference (Ort::Session::Run returns when the result is
    // there); errors of this call must surface in this call.
    model.synchronize();
}

void SegmentationImpl::compute_mask(Point const* point, Region const* region, uint8_t* const out_masks[3],
                                    float out_accuracy[3]) const {
    DLIMG_ASSERT(point || region);
    DLIMG_ASSERT(embedding_.get() != nullptr);
    float coords[4], labels[2];
    pack_prompt(image_size_, point, region, coords, labels);
    const bool is_single_mask = out_masks[1] == nullptr;
    if (is_single_mask) {
        DLIMG_ASSERT(out_masks[0] != nullptr);
    } else {
        for (int i = 0; i < 3; ++i) DLIMG_ASSERT(out_masks[i] != nullptr);
    }

    SamModel& model_ = env_.sam_model();
    std::lock_guard<std::mutex> lock(model_.mutex());
    HIP_CHECK(hipSetDevice(model_.device()));
    float const* emb = embedding_.get();
    model_.decode(&emb, coords, labels, 1);

    const Extent o = image_size_.original, r = image_size_.resized;
    k::PostJob jobs[3];
    int n_jobs = 0;
    if (is_single_mask) {
        // single-mask decoder: best of the four by SamOnnxModel.select_masks, chosen on the device
        jobs[n_jobs++] = k::PostJob{model_.logits(), model_.iou(), out_masks[0], o.width, o.height, r.width, r.height};
    } else {
        // multi-mask decoder: outputs 1..3 (reference: segmentation.cpp:167-172)
        for (int i = 0; i < 3; ++i)
            jobs[n_jobs++] = k::PostJob{model_.logits() + (size_t)(i + 1) * kLowRes * kLowRes, nullptr, out_masks[i],
                                        o.width, o.height, r.width, r.height};
    }
    float iou[4];
    if (!is_single_mask)
        HIP_CHECK(hipMemcpyAsync(iou, model_.iou(), sizeof(iou), hipMemcpyDeviceToHost, model_.stream()));
    model_.masks_to_host(jobs, n_jobs);      // synchronises the stream
    if (!is_single_mask)
        for (int i = 0; i < 3; ++i) out_accuracy[i] = iou[i + 1];
}

void SegmentationImpl::compute_mask_batch(SegmentationImpl const* const* segs, int count, int const* points,
                                          int const* regions, uint8_t* const* out_masks) {
    if (count <= 0) return;
    DLIMG_ASSERT((points != nullptr) != (regions != nullptr));
    SamModel& model = segs[0]->env_.sam_model();
    std::vector<float> coords((size_t)count * 4), labels((size_t)count * 2);
    std::vector<float const*> emb(count);
    for (int i = 0; i < count; ++i) {
        DLIMG_ASSERT(&segs[i]->env_ == &segs[0]->env_);
        DLIMG_ASSERT(segs[i]->embedding_.get() != nullptr && out_masks[i] != nullptr);
        if (points) {
            Point p{points[i * 2], points[i * 2 + 1]};
            pack_prompt(segs[i]->image_size_, &p, nullptr, &coords[i * 4], &labels[i * 2]);
        } else {
            Region r{Point{regions[i * 4], regions[i * 4 + 1]}, Point{regions[i * 4 + 2], regions[i * 4 + 3]}};
            pack_prompt(segs[i]->image_size_, nullptr, &r, &coords[i * 4], &labels[i * 2]);
        }
        emb[i] = segs[i]->embedding_.get();
    }
    std::lock_guard<std::mutex> lock(model.mutex());
    HIP_CHECK(hipSetDevice(model.device()));
    model.decode(emb.data(), coords.data(), labels.data(), count);
    std::vector<k::PostJob> jobs(count);
    for (int i = 0; i < count; ++i) {
        const Extent o = segs[i]->image_size_.original, r = segs[i]->image_size_.resized;
        jobs[i] = k::PostJob{model.logits() + (size_t)i * 4 * kLowRes * kLowRes, model.iou() + (size_t)i * 4,
                             out_masks[i], o.width, o.height, r.width, r.height};
    }
    model.masks_to_host(jobs.data(), count);
}

}  // namespace dlimg
