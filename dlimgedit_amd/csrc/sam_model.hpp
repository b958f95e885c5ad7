// SamModel: device-resident weights + the HIP executor for the Segment-Anything path.
// Takes the place of the reference's SegmentAnythingModel, i.e. of its three onnxruntime Sessions
// (/root/reference/src/segmentation.hpp:17-32, /root/reference/src/session.cpp:57-136):
//   encode()  == image_embedder.run(...)            (segmentation.cpp:126-128)
//   decode()  == single/multi_mask_decoder()(...)   (segmentation.cpp:154-158)
// Tensors stay in HBM between the two; nothing round-trips through host memory as it does in the
// reference (environment.cpp:142 binds every Ort::Value to CPU memory).
#pragma once

#include "common.hpp"

#include <dlimgedit/dlimgedit.h>
#include "kernels/kernels.hpp"
#include "resize_tables.hpp"
#include "weights.hpp"

#include <array>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace dlimg {

constexpr int kTokens = 4096;       // 64 x 64 embedding grid
constexpr int kEmbedDim = 256;      // channels of the image embedding
constexpr int kPatchK = 768;        // 3 * 16 * 16
constexpr int kImageSize = 1024;    // /root/reference/src/segmentation.cpp:17
constexpr int kDecTokens = 7;
constexpr int kLowRes = 256;

struct LinearH {                    // f16 weight for MFMA GEMMs, fp32 bias
    DeviceBuffer<half_t> w;
    DeviceBuffer<float> b;
    int out = 0, in = 0;
    bool has_bias = false;
    // set when a preceding LayerNorm is folded in: w = W * diag(gamma), b = b + W.beta and
    // colsum[n] = sum_k w[n][k] (of the f16-rounded values) for the mean correction in the GEMM epilogue
    DeviceBuffer<float> colsum;
};
struct LinearF {                    // fp32 weight for the token-side kernels
    DeviceBuffer<float> w, b;
    int out = 0, in = 0;
};
struct NormW { DeviceBuffer<float> w, b; };

struct EncoderLayer {
    bool global = false;
    NormW ln1, ln2;
    LinearH qkv, proj, fc1, fc2;
    DeviceBuffer<half_t> qkv_pad;    // q|k|v of a window's zero-padding token = the plain qkv bias (windowed layers)
    DeviceBuffer<half_t> rel_h16, rel_w16;    // the attention kernels take the tables as f16 (global layers: pre-scaled)
};

struct TokenAttention { LinearF q, k, v, o; };

struct DecoderLayer {
    TokenAttention self_attn;
    NormW ln1, ln2, ln3, ln4;
    LinearF t2i_q, t2i_o;            // token side of token->image attention
    DeviceBuffer<float> t2i_o_t;     // t2i_o.w transposed: [128][256] (kernels/decoder.hip, out_projection_columns)
    LinearH img_kqv;                 // [t2i.k ; i2t.q ; t2i.v] fused: one GEMM over the keys
    DeviceBuffer<float> pos_kqv;     // [4096, 384] pos . [Wk ; Wq]^T, zeros for the v columns (which take no pos)
    LinearF mlp1, mlp2;
    LinearF i2t_k, i2t_v;            // token side of image->token attention
    LinearH i2t_o;                   // image side output projection (128 -> 256)
};

// Device-resident weights of one SAM variant; immutable after loading and shared by every execution lane.
struct SamWeights {
    explicit SamWeights(std::string const& weight_path, int device);
    SamWeights(SamWeights const&) = delete;
    SamWeights& operator=(SamWeights const&) = delete;

    int device = 0;
    SamGeometry geom_;
    bool fused_ln_ = true;            // encoder LayerNorms folded into the qkv / fc1 GEMMs
    LinearH patch_;                       // [D, 768] + bias
    DeviceBuffer<float> pos_embed_;       // [4096, D]
    std::vector<EncoderLayer> layers_;
    LinearH neck1_, neck2_;               // 1x1 conv [256, D]; 3x3 conv as [256, 9*256] (tap-major columns)
    NormW neck_ln1_, neck_ln2_;
    DeviceBuffer<float> pe_gauss_, pe_point_, pe_not_a_point_, pe_no_mask_;
    DeviceBuffer<float> iou_token_, mask_tokens_;
    std::array<DecoderLayer, 2> dec_;
    LinearF final_q_, final_o_;
    DeviceBuffer<float> final_o_t_;       // final_o_.w transposed: [128][256]
    LinearH final_kv_;                    // [final.k ; final.v]
    DeviceBuffer<float> final_pos_kv_;    // [4096, 256] pos . Wk^T | 0
    NormW ln_final_;
    LinearH up1_, up2_;                   // transposed-conv weights as GEMM operands (sub-pixel-major rows)
    NormW up_ln_;
    std::array<std::array<LinearF, 3>, 5> heads_;   // 4 hyper MLPs + IoU head

};

// Stage clock for roofline accounting (HIP events on the executor's own stream).
// ST_GEMM is the sum over all GEMM launches of the encoder; ST_GEMM_* split the same launches by kernel flavour (what
// the by-grid table of a kernel trace tells apart): residual-stream writers with row statistics (patch / proj / fc2),
// LayerNorm-folded consumers without / with GELU (qkv / fc1), everything else (neck).
enum Stage { ST_PRE = 0, ST_GEMM, ST_LAYERNORM, ST_ATTN_WINDOW, ST_ATTN_GLOBAL, ST_ENC_OTHER, ST_DECODER, ST_POST,
             ST_GEMM_STATS, ST_GEMM_NORM, ST_GEMM_NORM_GELU, ST_GEMM_OTHER,
             // r06: the stream writers (ST_GEMM_STATS) once more by shape -- they share a kernel and a grid, so no profiler
             // table can tell them apart, and proj (K = D: 152 FLOP per byte at ViT-B) sits on the other side of the ridge
             // from fc2 (K = 4 D)
             ST_GEMM_PATCH, ST_GEMM_PROJ, ST_GEMM_FC2, ST_COUNT };

struct StageStats {
    double ms[ST_COUNT] = {0};
    double work[ST_COUNT] = {0};     // algorithmic FLOPs (MFMA stages) or bytes (HBM stages)
    long launches[ST_COUNT] = {0};
};

// Which execution lanes of a GPU have work in flight: one marker event per lane, re-recorded behind everything the lane
// enqueues (encode, decode, mask transfer).  A lane that starts an encoder pass asks whether every OTHER lane's marker has
// been reached: then the pass has the GPU to itself -- the situation of a synchronous caller of slots 3 / 4, which is what
// every existing user of the reference is (/root/reference/src/include/dlimgedit/detail/dlimgedit.impl.hpp:70-116) -- and
// its one-image GEMMs may trade CU time for latency (kernels/gemm.hip, tile 11).  hipEventQuery on an event another
// thread is re-recording is allowed; a stale answer only costs or gains a tile choice, never a result (same bits).
class LaneBoard {
  public:
    LaneBoard(int device, int lanes);
    ~LaneBoard();
    LaneBoard(LaneBoard const&) = delete;
    LaneBoard& operator=(LaneBoard const&) = delete;
    void begin(int lane);                         // the lane starts enqueuing a pass: busy until the next mark() / end()
    void end(int lane) noexcept;                  // (no event: for the error path of an enqueue)
    void mark(int lane, hipStream_t stream);      // behind what the lane has just enqueued
    bool others_idle(int lane) const;
    // diagnostics: encoder passes of one image enqueued so far, and how many of them found every other lane idle
    void count_pass(bool alone) { passes_.fetch_add(1, std::memory_order_relaxed); if (alone) alone_.fetch_add(1, std::memory_order_relaxed); }
    long passes() const { return passes_.load(std::memory_order_relaxed); }
    long alone_passes() const { return alone_.load(std::memory_order_relaxed); }

  private:
    std::atomic<long> passes_{0}, alone_{0};
    std::vector<hipEvent_t> marker_;
    std::unique_ptr<std::atomic<bool>[]> armed_;
    std::unique_ptr<std::atomic<bool>[]> enqueuing_;
};

class SamModel {
  public:
    // One execution lane: own stream, workspaces and staging buffers over shared weights.  Several lanes
    // let independent images overlap on the GPU (tails and small kernels of one image hide behind the
    // large kernels of another) -- the serving counterpart of the reference's "Environment is
    // thread-safe" contract (reference: src/include/dlimgedit/dlimgedit.hpp:98-101).
    explicit SamModel(std::shared_ptr<SamWeights const> weights, int lane_index = 0, int lane_count = 1,
                      std::shared_ptr<LaneBoard> board = nullptr);
    ~SamModel();
    SamModel(SamModel const&) = delete;
    SamModel& operator=(SamModel const&) = delete;

    SamGeometry const& geometry() const { return weights_->geom_; }
    hipStream_t stream() const { return stream_; }
    std::mutex& mutex() { return mutex_; }
    int device() const { return device_; }
    int lane_index() const { return lane_index_; }
    LaneBoard const* board() const { return board_.get(); }

    // All methods below require mutex() to be held by the caller, unless stated otherwise.  The mutex covers the
    // ENQUEUE of a request (host-side state, staging areas), not its execution: workspaces are re-used in stream
    // order, so a caller records completion(), releases the mutex and waits for its own event while the next
    // request is already being enqueued behind it.

    // Host image (already at its encoder resolution: longest side 1024) -> slot `slot` of the patch
    // matrix.  Copies through pinned staging and runs the pre-processing kernel.
    // Pixels in pinned image memory of the library with packed rows are sent from where they lie; the caller waits for
    // that copy (wait_caller_copies) before it hands the pixels back to their owner.
    void upload_image(int slot, int batch, uint8_t const* pixels, int w, int h, int stride, int channels);
    void wait_caller_copies();           // mutex() held or not: the event is this lane's own, re-recorded under mutex()
    // Host image whose longest side is not 1024: uploaded at its own size and resampled on the device to
    // rw x rh (reference: dlimg::resize through stb, /root/reference/src/image.cpp:37-51).
    void upload_and_resize_image(int slot, int batch, uint8_t const* pixels, int w, int h, int stride, int channels,
                                 int rw, int rh);
    // Device-resident image variant (used by the batch benchmark so PCIe is outside the timed region).
    void preprocess_device_image(int slot, int batch, uint8_t const* dev_pixels, int w, int h, int stride, int channels);
    // All `batch` slots from device-resident images in ONE launch (views carry device pixel pointers).
    void preprocess_device_images(dlimg_ImageView const* views, int batch);
    // Runs the encoder on `batch` uploaded images; embeddings [batch][4096][256] fp32 in embeddings() and, where
    // emb_dst[i] is given, in that device buffer too (batch 1: written there directly).
    void encode(int batch, float* const* emb_dst = nullptr);
    float const* embeddings() const { return emb_.get(); }

    // Decoder for `count` prompts. emb[i]: device embedding of prompt i's image; coords [count][2][2],
    // labels [count][2] host arrays. Results stay on device: logits() [count][4][256][256], iou() [count][4].
    void decode(float const* const* emb, float const* coords, float const* labels, int count);
    float const* logits() const { return logits_.get(); }
    // Diagnostic: the token-side workspaces as the last decode of ONE prompt left them (after synchronize()), one after
    // the other; names/sizes in decoder_state_layout().  What a parity or race hunt compares stage by stage.
    static std::vector<std::pair<const char*, size_t>> decoder_state_layout();
    void decoder_state(float* out) const;
    float const* iou() const { return iou_.get(); }

    // Post-process to host masks in two steps so that the wait happens outside mutex():
    //   slot = acquire_mask_slot()               no mutex needed; a free staging slot (or a new one: never blocks),
    //                                            the caller's until release_mask_slot
    //   enqueue_masks(slot, jobs, n, n_iou)      under mutex(): kernel -> device staging -> pinned staging (async),
    //                                            plus the first n_iou IoU predictions of the last decode()
    //   finish_masks(slot, jobs, n, iou_out)     no mutex: waits for the slot's event, copies into jobs[i].dst (HOST
    //                                            pointers of out_w*out_h bytes) and iou_out
    struct MaskSlot {
        DeviceBuffer<uint8_t> dev;
        PinnedBuffer pin;
        hipEvent_t done = nullptr;
        size_t iou_offset = 0;
        // masks whose destination is pinned image memory of the library (csrc/image_memory.hpp) and were written there by
        // the kernel itself: finish_masks has nothing to copy for them
        std::vector<char> in_place;
        // the staging area travels to the host in a few pieces, each with its own event, so that the host copies piece i
        // to the caller's buffers while piece i + 1 is still on the bus (enqueue_masks / finish_masks)
        std::vector<hipEvent_t> piece_done;
        std::vector<size_t> piece_end;
    };
    MaskSlot& acquire_mask_slot();
    void release_mask_slot(MaskSlot& s);
    void enqueue_masks(MaskSlot& slot, k::PostJob const* jobs, int count, int iou_count);
    void finish_masks(MaskSlot& slot, k::PostJob const* jobs, int count, float* iou_out, int iou_count);
    // Masks to DEVICE memory that may belong to ANOTHER GPU ("all masks on one device"): jobs[i].dst are
    // pointers valid on HIP device dst_device.  Same device as this lane's: the kernel writes them directly.  Another
    // device: the kernel writes the slot's staging memory and one hipMemcpyPeerAsync per mask moves it over xGMI.
    // slot.done is recorded behind the last write; wait_masks() (no mutex needed) waits for it.
    void enqueue_masks_device(MaskSlot& slot, k::PostJob const* jobs, int count, int dst_device);
    void wait_masks(MaskSlot& slot);
    // Blocking convenience form of the three calls above (mutex() held throughout).
    void masks_to_host(k::PostJob const* jobs, int count);
    // Same kernel, but jobs[i].dst are DEVICE pointers and nothing is copied or waited for.
    void masks_on_device(k::PostJob const* jobs, int count);

    void synchronize();
    // Event recorded behind everything enqueued so far; wait for it WITHOUT mutex(), then give it back.
    hipEvent_t completion();
    // Overflow report of the encoder pass enqueued last (valid under mutex(), right after encode()): an int in pinned host
    // memory that the pass sets to 1 when an activation left the f16 range somewhere in the image (an infinity or a NaN
    // reached the last LayerNorm of the neck).  Read it after the pass's completion event; slots are reused after
    // kPassFlags further passes of this lane.
    static constexpr int kPassFlags = 64;
    const volatile int* last_pass_flag() const { return pass_flag_; }
    // An encoder pass nobody has waited for yet: SegmentationImpl::process hands the caller's thread back once the pass is
    // enqueued, and the first request that needs the embedding is queued on the SAME lane behind it (stream order), so the
    // GPU goes from the encoder's last kernel to the decoder's first without a round trip through the host.  Whoever gets
    // there first settles it -- the handle (first mask query, re-use, destruction) or this lane, before it re-uses the
    // pass's flag; settle() is idempotent and callable from any thread without mutex().
    struct DeferredPass {
        SamModel* lane = nullptr;
        bool settle();                    // waits for the pass; true when it reported non-finite values
      private:
        friend class SamModel;
        std::mutex mutex_;
        hipEvent_t done_ = nullptr;
        const volatile int* flag_ = nullptr;
        bool settled_ = false, overflowed_ = false;
    };
    // Under mutex(), right after encode(): completion() + last_pass_flag() of that pass as one object.
    std::shared_ptr<DeferredPass> defer_last_pass();
    void wait_and_recycle(hipEvent_t e);      // no mutex needed
    bool poll_and_recycle(hipEvent_t e);      // no mutex needed: true (and the event is taken back) once it has completed

    void set_profiling(bool on);
    StageStats take_stats();

  private:
    void reserve_encoder(int batch);
    void reserve_decoder(int count);
    void decode_chunk(float const* const* emb, float const* coords, float const* labels, int count, int first);
    void gemm(k::GemmArgs const& a, Stage shape = ST_COUNT);     // shape: ST_GEMM_PATCH / _PROJ / _FC2 for the stage clocks
    template <typename F> void timed(Stage st, double work, F&& launch);
    void flush_events();
    hipEvent_t take_event();

    int device_ = 0;
    bool shared_gpu_ = false;            // other lanes run on this device too (GEMM tile choice, kernels/gemm.hip)
    std::shared_ptr<LaneBoard> board_;   // activity of the sibling lanes (null: a lane on its own)
    int lane_index_ = 0;
    bool alone_ = false;                 // the encoder pass being enqueued found every other lane idle (set by encode())
    void mark_activity() { if (board_) board_->mark(lane_index_, stream_); }
    void begin_activity() { if (board_) board_->begin(lane_index_); }
    hipStream_t stream_ = nullptr;
    std::mutex mutex_;

    std::shared_ptr<SamWeights const> weights_;

    // ---- encoder workspace (sized for enc_batch_ images)
    int enc_batch_ = 0;
    DeviceBuffer<uint8_t> img_dev_;
    // pinned staging for host images: a ring, each entry guarded by the event of the copy that last read it
    static constexpr int kStageRing = 4;
    struct ImageStage { PinnedBuffer pin; hipEvent_t copied = nullptr; };
    ImageStage stage_[kStageRing];
    unsigned stage_seq_ = 0;
    hipEvent_t caller_copied_ = nullptr;     // behind the last copy that read a caller's pinned pixels directly
    bool caller_copy_pending_ = false;
    uint8_t* stage_rows(uint8_t const* pixels, size_t row_bytes, int rows, int stride, hipEvent_t* copied);
    DeviceBuffer<half_t> patches_, xn_, xlo_, qkv_, att_, hid_;
    DeviceBuffer<float> x_, xstat_, neck_f32_, emb_;
    // The residual stream as an f16 pair (xn_ = hi, which is also the consumers' A operand; xlo_ = lo) instead of fp32 x_ +
    // its f16 copy xn_: 8 instead of 10 bytes per element through every stream writer (kernels.hpp, GemmArgs::out_l).
    // Needs the ping-pong epilogue for every stream writer: folded LayerNorms, several lanes (the shared-GPU tile choice) and
    // an embedding width that is a multiple of 256 (ViT-B / L / H; the reduced test variants keep the fp32 stream).
    bool split_stream_ = false;
    static bool split_stream_allowed();

    // ---- longest-side resize (images whose longest side is not 1024)
    struct AxisDev {
        int in_size = 0, out_size = 0, taps = 0;
        DeviceBuffer<int> first, count;
        DeviceBuffer<float> coef;
    };
    std::shared_ptr<AxisDev const> axis_table(int in_size, int out_size);
    static constexpr size_t kAxisCacheEntries = 64;
    std::vector<std::shared_ptr<AxisDev const>> axis_cache_;     // least recently used first
    DeviceBuffer<float> srgb_decode_;
    DeviceBuffer<uint32_t> srgb_encode_;
    DeviceBuffer<uint8_t> resize_src_;
    DeviceBuffer<float> resize_tmp_;

    // ---- decoder workspace (sized for dec_count_ prompts)
    int dec_count_ = 0;
    DeviceBuffer<float> keys_, logits_, iou_, hyper_;
    DeviceBuffer<half_t> keys_h_, kqv_h_;
    DeviceBuffer<float> tokens_, queries_, tk_, tv_, sq_, sk_, sv_, tsa_, tt2i_, tmlp_, t2i_part_;
    std::vector<std::unique_ptr<MaskSlot>> mask_slots_;     // all ever made (owned), guarded by done_mutex_
    std::vector<MaskSlot*> mask_free_;                      // those not handed out, guarded by done_mutex_

    // ---- profiling
    bool profiling_ = false;
    struct Pending { hipEvent_t a, b; Stage st; double work; Stage also = ST_COUNT; Stage shape = ST_COUNT; };
    std::vector<Pending> pending_;
    std::vector<hipEvent_t> event_pool_;
    StageStats stats_;
    std::mutex done_mutex_;
    int* pass_flags_ = nullptr;          // [kPassFlags] pinned, host-visible; pass_flag_ = the slot of the pass enqueued last
    int* pass_flag_ = nullptr;
    unsigned pass_counter_ = 0;
    std::shared_ptr<DeferredPass> flag_owner_[kPassFlags];   // deferred passes by the flag they report through
    std::vector<hipEvent_t> done_pool_;   // completion() events, guarded by done_mutex_ (taken without mutex_)
};

}  // namespace dlimg
