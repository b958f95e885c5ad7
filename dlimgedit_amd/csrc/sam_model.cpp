#include "sam_model.hpp"
#include "image_memory.hpp"
#include "mask_pieces.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>

namespace dlimg {

namespace {

constexpr float kLnEps = 1e-6f;       // encoder blocks and every LayerNorm2d
// norm1..4 of the two-way blocks and norm_final_attn: nn.LayerNorm's default in Meta's mask decoder, which the
// reference's decoder graphs are exports of (/root/reference/script/export_models.py:29-43)
constexpr float kDecLnEps = 1e-5f;

struct Loader {
    WeightFile const& file;
    hipStream_t stream;
    DeviceBuffer<float> staging;    // fp32 staging for device-side f16 conversion

    void f32(std::string const& name, std::vector<int64_t> const& dims, DeviceBuffer<float>& dst) {
        HostTensor const& t = file.get(name, dims);
        dst.reserve(t.numel());
        HIP_CHECK(hipMemcpy(dst.get(), t.data, t.numel() * 4, hipMemcpyHostToDevice));
    }
    void f32_host(std::vector<float> const& v, DeviceBuffer<float>& dst) {
        dst.reserve(v.size());
        HIP_CHECK(hipMemcpy(dst.get(), v.data(), v.size() * 4, hipMemcpyHostToDevice));
    }
    // Everything that becomes an f16 MFMA operand passes here: a value beyond the f16 range would be an infinity on the
    // device and every mask a NaN pattern, silently -- refused when the model is loaded instead (what: the tensor's name,
    // or what it was folded from)
    void f16_host(float const* src, size_t n, DeviceBuffer<half_t>& dst, std::string const& what) {
        for (size_t i = 0; i < n; ++i)
            if (!(std::fabs(src[i]) <= 65504.0f))
                throw Exception("'" + file.path() + "': " + what + " holds " + std::to_string(src[i]) + " (element " +
                                std::to_string(i) + "), outside the f16 range of this build's MFMA operands");
        staging.reserve(n);
        dst.reserve(n);
        HIP_CHECK(hipMemcpy(staging.get(), src, n * 4, hipMemcpyHostToDevice));
        k::cast_f16(staging.get(), dst.get(), n, stream);
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    // head_rows / head_scale: the first head_rows output rows (weight rows and bias entries) are multiplied by head_scale
    // before anything else happens to them -- the q rows of a global-attention block's qkv (kernels.hpp, attention_global)
    void linear_h(std::string const& prefix, int out, int in, bool bias, LinearH& l, int head_rows = 0, float head_scale = 1.f) {
        HostTensor const& w = file.get(prefix + ".w", {out, in});
        if (head_rows > 0) {
            std::vector<float> ws(w.data, w.data + w.numel());
            for (size_t i = 0; i < (size_t)head_rows * in; ++i) ws[i] *= head_scale;
            f16_host(ws.data(), ws.size(), l.w, prefix + ".w");
        } else {
            f16_host(w.data, w.numel(), l.w, prefix + ".w");
        }
        l.out = out;
        l.in = in;
        l.has_bias = bias;
        if (bias && head_rows > 0) {
            HostTensor const& b = file.get(prefix + ".b", {out});
            std::vector<float> bs(b.data, b.data + out);
            for (int i = 0; i < head_rows; ++i) bs[i] *= head_scale;
            f32_host(bs, l.b);
        } else if (bias) {
            f32(prefix + ".b", {out}, l.b);
        }
    }
    // Linear layer behind a LayerNorm, with the norm folded in: y = W (g*(x-mu)*rstd + beta) + b
    //   = rstd * ((W g) x - mu * rowsum(W g)) + (b + W beta).  The GEMM multiplies the raw x by W g and applies
    // the rest per output element; rowsum is taken over the f16 values the GEMM really multiplies with.
    void linear_ln_h(std::string const& prefix, std::string const& norm, int out, int in, LinearH& l, int head_rows = 0,
                     float head_scale = 1.f) {
        HostTensor const& w = file.get(prefix + ".w", {out, in});
        HostTensor const& b = file.get(prefix + ".b", {out});
        HostTensor const& gamma = file.get(norm + ".w", {in});
        HostTensor const& beta = file.get(norm + ".b", {in});
        std::vector<float> wg((size_t)out * in), colsum(out), bias(out);
        for (int n = 0; n < out; ++n) {
            const float rs = n < head_rows ? head_scale : 1.f;      // see linear_h
            double sum = 0, shift = 0;
            for (int i = 0; i < in; ++i) {
                const float v = rs * w.data[(size_t)n * in + i] * gamma.data[i];
                wg[(size_t)n * in + i] = v;
                sum += (double)(float)(half_t)v;
                shift += (double)rs * w.data[(size_t)n * in + i] * beta.data[i];
            }
            colsum[n] = (float)sum;
            bias[n] = (float)((double)rs * b.data[n] + shift);
        }
        f16_host(wg.data(), wg.size(), l.w, prefix + ".w scaled by " + norm + ".w");
        f32_host(bias, l.b);
        f32_host(colsum, l.colsum);
        l.out = out;
        l.in = in;
        l.has_bias = true;
    }
    void linear_f(std::string const& prefix, int out, int in, LinearF& l) {
        f32(prefix + ".w", {out, in}, l.w);
        f32(prefix + ".b", {out}, l.b);
        l.out = out;
        l.in = in;
    }
    // [out][in] weight as [in][out]
    void transposed_f(std::string const& prefix, int out, int in, DeviceBuffer<float>& dst) {
        HostTensor const& w = file.get(prefix + ".w", {out, in});
        std::vector<float> t((size_t)out * in);
        for (int n = 0; n < out; ++n)
            for (int i = 0; i < in; ++i) t[(size_t)i * out + n] = w.data[(size_t)n * in + i];
        f32_host(t, dst);
    }
    void norm(std::string const& prefix, int dim, NormW& n) {
        f32(prefix + ".w", {dim}, n.w);
        f32(prefix + ".b", {dim}, n.b);
    }
    void attention(std::string const& prefix, int dim, int inner, TokenAttention& a) {
        linear_f(prefix + ".q", inner, dim, a.q);
        linear_f(prefix + ".k", inner, dim, a.k);
        linear_f(prefix + ".v", inner, dim, a.v);
        linear_f(prefix + ".o", dim, inner, a.o);
    }
    // rows of the named linears one after the other -> one f16 GEMM weight with concatenated bias
    void fused_h(std::vector<std::string> const& parts, int out_each, int in, LinearH& l) {
        const size_t n = parts.size();
        std::vector<float> w(n * (size_t)out_each * in), b(n * (size_t)out_each);
        for (size_t i = 0; i < n; ++i) {
            HostTensor const& wi = file.get(parts[i] + ".w", {out_each, in});
            HostTensor const& bi = file.get(parts[i] + ".b", {out_each});
            std::memcpy(w.data() + i * wi.numel(), wi.data, wi.numel() * 4);
            std::memcpy(b.data() + i * out_each, bi.data, bi.numel() * 4);
        }
        f16_host(w.data(), w.size(), l.w, parts[0] + ".w (fused with its siblings)");
        f32_host(b, l.b);
        l.out = int(n) * out_each;
        l.in = in;
        l.has_bias = true;
    }
    // ConvTranspose2d(k=2, s=2) weight [ci, co, 2, 2] -> GEMM weight [n = (dy*2+dx)*co_n + co][k = ci]
    void conv_transpose_h(std::string const& prefix, int ci_n, int co_n, LinearH& l) {
        HostTensor const& w = file.get(prefix + ".w", {ci_n, co_n, 2, 2});
        HostTensor const& b = file.get(prefix + ".b", {co_n});
        std::vector<float> g((size_t)4 * co_n * ci_n), gb((size_t)4 * co_n);
        for (int s = 0; s < 4; ++s)
            for (int co = 0; co < co_n; ++co) {
                gb[(size_t)s * co_n + co] = b.data[co];
                for (int ci = 0; ci < ci_n; ++ci)
                    g[((size_t)s * co_n + co) * ci_n + ci] = w.data[((size_t)ci * co_n + co) * 4 + s];
            }
        f16_host(g.data(), g.size(), l.w, prefix + ".w");
        f32_host(gb, l.b);
        l.out = 4 * co_n;
        l.in = ci_n;
        l.has_bias = true;
    }
};

}  // namespace

SamWeights::SamWeights(std::string const& weight_path, int device_index) : device(device_index) {
    WeightFile file(weight_path);
    geom_ = file.geometry();
    const int D = geom_.embed_dim, hd = geom_.head_dim();
    if (D % 64 || geom_.mlp_dim % 64) throw Exception("SAM encoder width must be a multiple of 64");
    if (D != hd * geom_.num_heads || (hd != 64 && hd != 80))
        throw Exception("SAM encoder head dimension must be 64 or 80");
    // The encoder's LayerNorms run inside the GEMMs around them (see encode()); DLIMGEDIT_FUSED_LN=0 keeps
    // them as separate kernels for A/B measurements.
    if (const char* e = std::getenv("DLIMGEDIT_FUSED_LN")) fused_ln_ = std::atoi(e) != 0;

    HIP_CHECK(hipSetDevice(device));
    hipStream_t stream_ = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    Loader ld{file, stream_, {}};

    ld.linear_h("enc.patch", D, kPatchK, true, patch_);
    ld.f32("enc.pos", {kTokens, D}, pos_embed_);
    layers_.resize(geom_.depth);
    for (int i = 0; i < geom_.depth; ++i) {
        EncoderLayer& L = layers_[i];
        const std::string p = "enc.L" + std::to_string(i);
        L.global = geom_.is_global(i);
        const int span = L.global ? 64 : 14;
        if (!L.global) {
            HostTensor const& qb = file.get(p + ".qkv.b", {3 * D});
            ld.f16_host(qb.data, qb.numel(), L.qkv_pad, p + ".qkv.b");
        }
        // a global block's attention kernel works in units of log2 on pre-scaled operands (kernels.hpp): q rows of the
        // qkv weight and bias times log2(e) / sqrt(hd), rel-pos tables times sqrt(hd); L.qkv_pad (the windowed
        // kernel's padding bias) is not used by global blocks
        const int q_rows = L.global ? D : 0;
        const float q_scale = L.global ? k::attention_global_q_scale(hd) : 1.f;
        if (fused_ln_) {
            ld.linear_ln_h(p + ".qkv", p + ".ln1", 3 * D, D, L.qkv, q_rows, q_scale);
            ld.linear_ln_h(p + ".fc1", p + ".ln2", geom_.mlp_dim, D, L.fc1);
        } else {
            ld.norm(p + ".ln1", D, L.ln1);
            ld.norm(p + ".ln2", D, L.ln2);
            ld.linear_h(p + ".qkv", 3 * D, D, true, L.qkv, q_rows, q_scale);
            ld.linear_h(p + ".fc1", geom_.mlp_dim, D, true, L.fc1);
        }
        HostTensor const& rh = file.get(p + ".rel_h", {2 * span - 1, hd});
        HostTensor const& rw = file.get(p + ".rel_w", {2 * span - 1, hd});
        if (L.global) {
            std::vector<float> rhs(rh.data, rh.data + rh.numel()), rws(rw.data, rw.data + rw.numel());
            const float rel_scale = k::attention_global_rel_scale(hd);
            for (auto& v : rhs) v *= rel_scale;
            for (auto& v : rws) v *= rel_scale;
            ld.f16_host(rhs.data(), rhs.size(), L.rel_h16, p + ".rel_h");
            ld.f16_host(rws.data(), rws.size(), L.rel_w16, p + ".rel_w");
        } else {
            ld.f16_host(rh.data, rh.numel(), L.rel_h16, p + ".rel_h");
            ld.f16_host(rw.data, rw.numel(), L.rel_w16, p + ".rel_w");
        }
        ld.linear_h(p + ".proj", D, D, true, L.proj);
        ld.linear_h(p + ".fc2", D, geom_.mlp_dim, true, L.fc2);
    }
    ld.linear_h("enc.neck.conv1", kEmbedDim, D, false, neck1_);
    ld.norm("enc.neck.ln1", kEmbedDim, neck_ln1_);
    {   // 3x3 conv [co, ci, ky, kx] -> [co][(ky*3+kx)*256 + ci], matching im2col3x3's column order
        HostTensor const& w = file.get("enc.neck.conv2.w", {kEmbedDim, kEmbedDim, 3, 3});
        std::vector<float> g(w.numel());
        for (int co = 0; co < kEmbedDim; ++co)
            for (int ci = 0; ci < kEmbedDim; ++ci)
                for (int t = 0; t < 9; ++t)
                    g[((size_t)co * 9 + t) * kEmbedDim + ci] = w.data[((size_t)co * kEmbedDim + ci) * 9 + t];
        ld.f16_host(g.data(), g.size(), neck2_.w, "enc.neck.conv2.w");
        neck2_.out = kEmbedDim;
        neck2_.in = 9 * kEmbedDim;
    }
    ld.norm("enc.neck.ln2", kEmbedDim, neck_ln2_);

    DeviceBuffer<half_t> pe_h;      // dense positional encoding as a GEMM operand, only needed below
    ld.f32("pe.gauss", {2, 128}, pe_gauss_);
    ld.f32("pe.point", {4, 256}, pe_point_);
    ld.f32("pe.not_a_point", {256}, pe_not_a_point_);
    ld.f32("pe.no_mask", {256}, pe_no_mask_);
    {   // dense positional encoding of the 64x64 grid (PositionEmbeddingRandom.forward), constant
        HostTensor const& g = file.get("pe.gauss", {2, 128});
        std::vector<float> pe((size_t)kTokens * 256);
        for (int y = 0; y < 64; ++y)
            for (int x = 0; x < 64; ++x) {
                const float cx = 2.0f * ((x + 0.5f) / 64.0f) - 1.0f, cy = 2.0f * ((y + 0.5f) / 64.0f) - 1.0f;
                float* row = pe.data() + ((size_t)y * 64 + x) * 256;
                for (int kf = 0; kf < 128; ++kf) {
                    const float v = 6.283185307179586f * (cx * g.data[kf] + cy * g.data[128 + kf]);
                    row[kf] = std::sin(v);
                    row[128 + kf] = std::cos(v);
                }
            }
        ld.f16_host(pe.data(), pe.size(), pe_h, "the dense positional encoding");
    }
    // (keys + pos) W = keys W + pos W: the second term is a constant of the model, computed here once (same GEMM
    // kernel, f16 pos like the sum it replaces) and added by the image-side projections as an fp32 addend.  Columns
    // past `with_pos` (the value projection, which takes the keys without pos) stay zero.
    auto pos_term = [&](LinearH const& l, int with_pos, DeviceBuffer<float>& dst) {
        dst.reserve((size_t)kTokens * l.out);
        HIP_CHECK(hipMemsetAsync(dst.get(), 0, (size_t)kTokens * l.out * sizeof(float), stream_));
        k::GemmArgs g;
        g.A = pe_h.get(); g.lda = 256; g.W = l.w.get(); g.ldw = 256;
        g.out_f32 = dst.get(); g.ldc32 = l.out; g.M = kTokens; g.N = with_pos; g.K = 256;
        g.unit_rows = kTokens;
        k::gemm(g, stream_);
    };
    ld.f32("dec.iou_token", {256}, iou_token_);
    ld.f32("dec.mask_tokens", {4, 256}, mask_tokens_);
    for (int i = 0; i < 2; ++i) {
        DecoderLayer& L = dec_[i];
        const std::string p = "dec.L" + std::to_string(i);
        ld.attention(p + ".self", 256, 256, L.self_attn);
        ld.norm(p + ".ln1", 256, L.ln1);
        ld.norm(p + ".ln2", 256, L.ln2);
        ld.norm(p + ".ln3", 256, L.ln3);
        ld.norm(p + ".ln4", 256, L.ln4);
        ld.linear_f(p + ".t2i.q", 128, 256, L.t2i_q);
        ld.linear_f(p + ".t2i.o", 256, 128, L.t2i_o);
        ld.transposed_f(p + ".t2i.o", 256, 128, L.t2i_o_t);
        ld.fused_h({p + ".t2i.k", p + ".i2t.q", p + ".t2i.v"}, 128, 256, L.img_kqv);
        pos_term(L.img_kqv, 256, L.pos_kqv);
        ld.linear_f(p + ".mlp.fc1", 2048, 256, L.mlp1);
        ld.linear_f(p + ".mlp.fc2", 256, 2048, L.mlp2);
        ld.linear_f(p + ".i2t.k", 128, 256, L.i2t_k);
        ld.linear_f(p + ".i2t.v", 128, 256, L.i2t_v);
        ld.linear_h(p + ".i2t.o", 256, 128, true, L.i2t_o);
    }
    ld.linear_f("dec.final.q", 128, 256, final_q_);
    ld.linear_f("dec.final.o", 256, 128, final_o_);
    ld.transposed_f("dec.final.o", 256, 128, final_o_t_);
    ld.fused_h({"dec.final.k", "dec.final.v"}, 128, 256, final_kv_);
    pos_term(final_kv_, 128, final_pos_kv_);
    ld.norm("dec.ln_final", 256, ln_final_);
    ld.conv_transpose_h("dec.up1", 256, 64, up1_);
    ld.norm("dec.up_ln", 64, up_ln_);
    ld.conv_transpose_h("dec.up2", 64, 32, up2_);
    for (int m = 0; m < 5; ++m) {
        const std::string p = m < 4 ? "dec.hyper" + std::to_string(m) : std::string("dec.iou");
        const int last = m < 4 ? 32 : 4;
        ld.linear_f(p + ".0", 256, 256, heads_[m][0]);
        ld.linear_f(p + ".1", 256, 256, heads_[m][1]);
        ld.linear_f(p + ".2", last, 256, heads_[m][2]);
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    HIP_CHECK(hipStreamDestroy(stream_));
}

LaneBoard::LaneBoard(int device, int lanes)
    : armed_(new std::atomic<bool>[std::max(1, lanes)]), enqueuing_(new std::atomic<bool>[std::max(1, lanes)]) {
    HIP_CHECK(hipSetDevice(device));
    for (int i = 0; i < lanes; ++i) {
        hipEvent_t e = nullptr;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        marker_.push_back(e);
        armed_[i].store(false);
        enqueuing_[i].store(false);
    }
}

LaneBoard::~LaneBoard() {
    for (hipEvent_t e : marker_) (void)hipEventDestroy(e);
}

void LaneBoard::begin(int lane) {
    if (lane >= 0 && lane < (int)marker_.size()) enqueuing_[lane].store(true, std::memory_order_release);
}

void LaneBoard::end(int lane) noexcept {
    if (lane >= 0 && lane < (int)marker_.size()) enqueuing_[lane].store(false, std::memory_order_release);
}

void LaneBoard::mark(int lane, hipStream_t stream) {
    if (lane < 0 || lane >= (int)marker_.size()) return;
    HIP_CHECK(hipEventRecord(marker_[lane], stream));
    armed_[lane].store(true, std::memory_order_release);
    enqueuing_[lane].store(false, std::memory_order_release);
}

bool LaneBoard::others_idle(int lane) const {
    for (int i = 0; i < (int)marker_.size(); ++i) {
        if (i == lane) continue;
        if (enqueuing_[i].load(std::memory_order_acquire)) return false;       // a pass is being enqueued there right now
        if (!armed_[i].load(std::memory_order_acquire)) continue;
        const hipError_t st = hipEventQuery(marker_[i]);
        if (st == hipErrorNotReady) return false;
        if (st != hipSuccess) (void)hipGetLastError();      // not this call's problem: treated as "busy" is the safe answer
        if (st != hipSuccess) return false;
    }
    return true;
}

SamModel::SamModel(std::shared_ptr<SamWeights const> weights, int lane_index, int lane_count, std::shared_ptr<LaneBoard> board)
    : device_(weights->device), shared_gpu_(lane_count > 1), board_(std::move(board)), lane_index_(lane_index),
      weights_(std::move(weights)) {
    HIP_CHECK(hipSetDevice(device_));
    {
        // The runtime multiplexes streams of one priority onto its hardware queues, shared with the host's other streams.
        // With the default four queues a fourth lane of the same priority ends up behind another lane's kernels and costs
        // 15 %; each priority level has its own queues, so the lanes are then spread over the three levels (no lane is
        // favoured for long because requests are dealt round-robin).  With eight queues plain streams are better: the
        // priority levels make four host threads wait on each other's lanes (ABI, config 2 from four threads: 479 against
        // 569-596 images/s; one prompt per call from four threads: 4900 against 5500 masks/s).  The library asks for eight
        // queues when it is loaded (environment.cpp) and remembers whether the runtime can have seen that request: plain
        // streams are the default only when it can (hardware_queues_trusted(): the host set >= 8 itself, or the library set
        // it before the runtime initialised); a host that initialised HIP first keeps the three-priority layout.
        // DLIMGEDIT_PLAIN_STREAMS=0/1 overrides the detection.
        static const bool plain = [] {
            if (const char* e = std::getenv("DLIMGEDIT_PLAIN_STREAMS")) return std::atoi(e) != 0;
            return hardware_queues_trusted();
        }();
        if (plain) {
            HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        } else {
            int least = 0, greatest = 0;
            HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            const int prio = least + (greatest - least) * (lane_index % 3) / 2;
            HIP_CHECK(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, prio));
        }
    }
    for (auto& st : stage_) HIP_CHECK(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&caller_copied_, hipEventDisableTiming));
}

SamModel::~SamModel() {
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& d : flag_owner_) {                // handles that outlive their lane find their pass settled
        try {
            if (d) d->settle();
        } catch (...) {
        }
    }
    for (auto& p : pending_) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto e : event_pool_) (void)hipEventDestroy(e);
    for (auto e : done_pool_) (void)hipEventDestroy(e);
    for (auto& st : stage_)
        if (st.copied) (void)hipEventDestroy(st.copied);
    if (caller_copied_) (void)hipEventDestroy(caller_copied_);
    for (auto& m : mask_slots_) {
        if (m->done) (void)hipEventDestroy(m->done);
        for (auto e : m->piece_done) (void)hipEventDestroy(e);
    }
    if (stream_) (void)hipStreamDestroy(stream_);
    if (pass_flags_) (void)hipHostFree(pass_flags_);
}

// ---------------------------------------------------------------------------------------------
// profiling

void SamModel::set_profiling(bool on) {
    flush_events();
    profiling_ = on;
}

void SamModel::flush_events() {
    if (pending_.empty()) return;
    HIP_CHECK(hipStreamSynchronize(stream_));
    for (auto& p : pending_) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        stats_.ms[p.st] += ms;
        stats_.work[p.st] += p.work;
        stats_.launches[p.st] += 1;
        for (Stage extra : {p.also, p.shape}) {
            if (extra == ST_COUNT) continue;
            stats_.ms[extra] += ms;
            stats_.work[extra] += p.work;
            stats_.launches[extra] += 1;
        }
        event_pool_.push_back(p.a);
        event_pool_.push_back(p.b);
    }
    pending_.clear();
}

StageStats SamModel::take_stats() {
    flush_events();
    StageStats s = stats_;
    stats_ = StageStats{};
    return s;
}

hipEvent_t SamModel::take_event() {
    hipEvent_t e;
    if (!event_pool_.empty()) {
        e = event_pool_.back();
        event_pool_.pop_back();
    } else {
        HIP_CHECK(hipEventCreate(&e));
    }
    return e;
}

template <typename F> void SamModel::timed(Stage st, double work, F&& launch) {
    if (!profiling_) {
        launch();
        return;
    }
    Pending p{take_event(), take_event(), st, work};
    HIP_CHECK(hipEventRecord(p.a, stream_));
    launch();
    HIP_CHECK(hipEventRecord(p.b, stream_));
    pending_.push_back(p);
    if (pending_.size() > 8192) flush_events();
}

void SamModel::gemm(k::GemmArgs const& args, Stage shape) {
    k::GemmArgs a = args;
    a.shared_gpu = shared_gpu_;
    a.alone = alone_;
    a.unit_rows = kTokens;
    if (!profiling_) {
        k::gemm(a, stream_);
        return;
    }
    // the clock of a GEMM launch is the kernel's own dispatch-to-completion time (events attached to the dispatch)
    const Stage flavour = a.stats_out ? ST_GEMM_STATS : a.ln_stats ? (a.act == k::ACT_GELU ? ST_GEMM_NORM_GELU : ST_GEMM_NORM) : ST_GEMM_OTHER;
    Pending p{take_event(), take_event(), ST_GEMM, 2.0 * a.M * a.N * a.K, flavour, shape};
    k::gemm(a, stream_, p.a, p.b);
    pending_.push_back(p);
    if (pending_.size() > 8192) flush_events();
}

void SamModel::synchronize() { HIP_CHECK(hipStreamSynchronize(stream_)); }

hipEvent_t SamModel::completion() {
    hipEvent_t e = nullptr;
    {
        std::lock_guard<std::mutex> lock(done_mutex_);
        if (!done_pool_.empty()) {
            e = done_pool_.back();
            done_pool_.pop_back();
        }
    }
    if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(e, stream_));
    return e;
}

void SamModel::wait_and_recycle(hipEvent_t e) {
    const hipError_t err = hipEventSynchronize(e);
    {
        std::lock_guard<std::mutex> lock(done_mutex_);
        done_pool_.push_back(e);
    }
    HIP_CHECK(err);
}

std::shared_ptr<SamModel::DeferredPass> SamModel::defer_last_pass() {
    DLIMG_ASSERT(pass_flag_ != nullptr);
    auto d = std::make_shared<DeferredPass>();
    d->lane = this;
    d->done_ = completion();
    d->flag_ = pass_flag_;
    flag_owner_[pass_flag_ - pass_flags_] = d;
    return d;
}

bool SamModel::DeferredPass::settle() {
    std::lock_guard<std::mutex> lock(mutex_);
    if (!settled_) {
        settled_ = true;                         // the event goes back to the lane whatever the wait reports
        lane->wait_and_recycle(done_);
        overflowed_ = *flag_ != 0;
        *const_cast<volatile int*>(flag_) = 0;
    }
    return overflowed_;
}

bool SamModel::poll_and_recycle(hipEvent_t e) {
    const hipError_t err = hipEventQuery(e);
    if (err == hipErrorNotReady) return false;
    {
        std::lock_guard<std::mutex> lock(done_mutex_);
        done_pool_.push_back(e);
    }
    HIP_CHECK(err);
    return true;
}

// ---------------------------------------------------------------------------------------------
// encoder

void SamModel::reserve_encoder(int batch) {
    SamWeights const& W = *weights_;
    if (batch <= enc_batch_) return;
    HIP_CHECK(hipStreamSynchronize(stream_));
    const size_t M = (size_t)batch * kTokens;
    const size_t D = W.geom_.embed_dim;
    const size_t wide = std::max<size_t>(W.geom_.mlp_dim, 9 * kEmbedDim);
    img_dev_.reserve((size_t)batch * kImageSize * kImageSize * 4);
    patches_.reserve(M * kPatchK);
    split_stream_ = W.fused_ln_ && shared_gpu_ && D % 256 == 0 && split_stream_allowed();
    if (split_stream_) xlo_.reserve(M * D);
    else x_.reserve(M * D);
    xn_.reserve(M * D);
    xstat_.reserve(M * 24 * 2);              // at most 24 tile column blocks per row (kernels/gemm.hip)
    qkv_.reserve(M * 3 * D);
    att_.reserve(M * std::max<size_t>(D, kEmbedDim));
    hid_.reserve(M * wide);
    neck_f32_.reserve(M * kEmbedDim);
    emb_.reserve(M * kEmbedDim);
    enc_batch_ = batch;
}

bool SamModel::split_stream_allowed() {
    // measurement aid like DLIMGEDIT_FUSED_LN: =0 keeps the fp32 stream (same-box A/B of the two representations)
    static const bool on = [] { const char* e = std::getenv("DLIMGEDIT_SPLIT_STREAM"); return !e || std::atoi(e) != 0; }();
    return on;
}

void SamModel::preprocess_device_image(int slot, int batch, uint8_t const* dev_pixels, int w, int h, int stride,
                                       int channels) {
    DLIMG_ASSERT(slot >= 0 && slot < batch);
    reserve_encoder(batch);
    const int bytes = channels > 4 ? 4 : channels;
    timed(ST_PRE, (double)w * h * bytes + (double)kTokens * kPatchK * 2, [&] {
        k::preprocess(dev_pixels, w, h, stride, channels, patches_.get() + (size_t)slot * kTokens * kPatchK, stream_);
    });
}

void SamModel::preprocess_device_images(dlimg_ImageView const* views, int batch) {
    DLIMG_ASSERT(views != nullptr && batch > 0);
    reserve_encoder(batch);
    std::vector<k::PreImage> images(batch);
    double bytes = 0;
    for (int i = 0; i < batch; ++i) {
        const int px = views[i].channels > 4 ? 4 : views[i].channels;
        images[i] = k::PreImage{views[i].pixels, views[i].width, views[i].height, views[i].stride, views[i].channels,
                                patches_.get() + (size_t)i * kTokens * kPatchK};
        bytes += (double)views[i].width * views[i].height * px + (double)kTokens * kPatchK * 2;
    }
    timed(ST_PRE, bytes, [&] { k::preprocess_batch(images.data(), batch, stream_); });
}

// Packs `rows` rows of `row_bytes` bytes into the next entry of the pinned staging ring and returns it; *copied is the
// event the caller records behind its copy out of the entry (the entry is not touched again before that event).
uint8_t* SamModel::stage_rows(uint8_t const* pixels, size_t row_bytes, int rows, int stride, hipEvent_t* copied) {
    ImageStage& st = stage_[stage_seq_++ % kStageRing];
    HIP_CHECK(hipEventSynchronize(st.copied));       // the copy that last read this entry has run
    st.pin.reserve(row_bytes * rows);                // (re-allocation is safe for the same reason)
    uint8_t* pin = static_cast<uint8_t*>(st.pin.get());
    if ((size_t)stride == row_bytes) {
        std::memcpy(pin, pixels, row_bytes * rows);
    } else {
        for (int y = 0; y < rows; ++y) std::memcpy(pin + y * row_bytes, pixels + (size_t)y * stride, row_bytes);
    }
    *copied = st.copied;
    return pin;
}

void SamModel::upload_image(int slot, int batch, uint8_t const* pixels, int w, int h, int stride, int channels) {
    DLIMG_ASSERT(slot >= 0 && slot < batch);
    DLIMG_ASSERT(w > 0 && h > 0 && w <= kImageSize && h <= kImageSize);
    reserve_encoder(batch);
    const int bytes = channels > 4 ? 4 : channels;
    const size_t row = (size_t)w * bytes;
    const size_t slot_bytes = (size_t)kImageSize * kImageSize * 4;
    // Rows are packed on the way (the reference's create_image_tensor assumes packed rows when no
    // resize happens, segmentation.cpp:81-106; honouring the stride is identical for packed views).
    // In pieces of ~1 MiB: piece i crosses PCIe while the host packs piece i + 1 into the pinned area (a 4 MiB image:
    // 0.30 -> ~0.2 ms of a synchronous caller's 2.6 ms per image; one piece for small images)
    uint8_t* dev = img_dev_.get() + slot * slot_bytes;
    if ((size_t)stride == row && image_memory_is_pinned(pixels, row * h)) {
        // pixels the library allocated itself (an Image of the consumer: load_image / create_image) are pinned: one copy
        // command from where they lie, no packing pass
        // (the pre-processing kernel reading the pinned pixels itself, no copy command: 489 -> 476-485 images/s; not kept)
        HIP_CHECK(hipMemcpyAsync(dev, pixels, row * h, hipMemcpyHostToDevice, stream_));
        HIP_CHECK(hipEventRecord(caller_copied_, stream_));
        caller_copy_pending_ = true;
        preprocess_device_image(slot, batch, dev, w, h, (int)row, channels);
        return;
    }
    ImageStage& st = stage_[stage_seq_++ % kStageRing];
    HIP_CHECK(hipEventSynchronize(st.copied));       // the copy that last read this entry has run
    st.pin.reserve(row * h);                         // (re-allocation is safe for the same reason)
    uint8_t* pin = static_cast<uint8_t*>(st.pin.get());
    // (r06, the wrapper's own loop with a 4 MiB image: pieces of 4 / 2 / 1 / 0.5 / 0.25 MiB = 468 / 478 / 478 / 464 / 448 images/s
    // -- a copy command costs ~9 us of its own; from pinned image memory, one command and no packing: 493)
    const int pieces = (int)std::min<size_t>(8, std::max<size_t>(1, row * h / (1u << 20)));
    for (int p = 0; p < pieces; ++p) {
        const int y0 = (int)((long)h * p / pieces), y1 = (int)((long)h * (p + 1) / pieces);
        if ((size_t)stride == row) {
            std::memcpy(pin + (size_t)y0 * row, pixels + (size_t)y0 * stride, row * (size_t)(y1 - y0));
        } else {
            for (int y = y0; y < y1; ++y) std::memcpy(pin + (size_t)y * row, pixels + (size_t)y * stride, row);
        }
        HIP_CHECK(hipMemcpyAsync(dev + (size_t)y0 * row, pin + (size_t)y0 * row, row * (size_t)(y1 - y0), hipMemcpyHostToDevice, stream_));
    }
    HIP_CHECK(hipEventRecord(st.copied, stream_));
    preprocess_device_image(slot, batch, dev, w, h, (int)row, channels);
}

void SamModel::wait_caller_copies() {
    if (!caller_copy_pending_) return;
    caller_copy_pending_ = false;
    HIP_CHECK(hipEventSynchronize(caller_copied_));
}

std::shared_ptr<SamModel::AxisDev const> SamModel::axis_table(int in_size, int out_size) {
    for (size_t i = 0; i < axis_cache_.size(); ++i)
        if (axis_cache_[i]->in_size == in_size && axis_cache_[i]->out_size == out_size) {
            // most recently used entry last; eviction takes from the front
            std::rotate(axis_cache_.begin() + i, axis_cache_.begin() + i + 1, axis_cache_.end());
            return axis_cache_.back();
        }
    AxisTable t = make_axis_table(in_size, out_size);
    auto a = std::make_shared<AxisDev>();
    a->in_size = in_size;
    a->out_size = out_size;
    a->taps = t.taps;
    a->first.reserve(t.first.size());
    a->count.reserve(t.count.size());
    a->coef.reserve(t.coef.size());
    HIP_CHECK(hipMemcpy(a->first.get(), t.first.data(), t.first.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(a->count.get(), t.count.data(), t.count.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(a->coef.get(), t.coef.data(), t.coef.size() * sizeof(float), hipMemcpyHostToDevice));
    if (axis_cache_.size() >= kAxisCacheEntries) {
        // The entry's device tables may still be read by a resize kernel queued on this lane's stream; callers hold
        // their own reference for the duration of the call, but the kernel outlives the call.
        HIP_CHECK(hipStreamSynchronize(stream_));
        axis_cache_.erase(axis_cache_.begin());
    }
    axis_cache_.push_back(a);
    return a;
}

void SamModel::upload_and_resize_image(int slot, int batch, uint8_t const* pixels, int w, int h, int stride, int channels,
                                       int rw, int rh) {
    DLIMG_ASSERT(slot >= 0 && slot < batch);
    DLIMG_ASSERT(w > 0 && h > 0 && rw > 0 && rh > 0 && rw <= kImageSize && rh <= kImageSize);
    reserve_encoder(batch);
    const int bytes = channels > 4 ? 4 : channels;
    const size_t row = (size_t)w * bytes;
    if (!srgb_decode_.get()) {
        float lut[256];
        srgb_decode_table(lut);
        srgb_decode_.reserve(256);
        srgb_encode_.reserve(104);
        HIP_CHECK(hipMemcpy(srgb_decode_.get(), lut, sizeof(lut), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(srgb_encode_.get(), kSrgbEncodeTab4, sizeof(kSrgbEncodeTab4), hipMemcpyHostToDevice));
    }
    // both tables are held by value: looking up the second axis may evict the entry of the first
    const std::shared_ptr<AxisDev const> axp = axis_table(w, rw), ayp = axis_table(h, rh);
    AxisDev const& ax = *axp;
    AxisDev const& ay = *ayp;
    // the source image and its fp32 intermediate are re-used by the next resize in stream order; growing them
    // frees memory that queued kernels may still read, so everything queued runs first
    if (row * h > resize_src_.capacity() || (size_t)h * rw * bytes > resize_tmp_.capacity())
        HIP_CHECK(hipStreamSynchronize(stream_));
    resize_src_.reserve(row * h);
    resize_tmp_.reserve((size_t)h * rw * bytes);
    if ((size_t)stride == row && image_memory_is_pinned(pixels, row * h)) {
        HIP_CHECK(hipMemcpyAsync(resize_src_.get(), pixels, row * h, hipMemcpyHostToDevice, stream_));     // as upload_image
        HIP_CHECK(hipEventRecord(caller_copied_, stream_));
        caller_copy_pending_ = true;
    } else {
        hipEvent_t copied = nullptr;
        uint8_t* pin = stage_rows(pixels, row, h, stride, &copied);
        HIP_CHECK(hipMemcpyAsync(resize_src_.get(), pin, row * h, hipMemcpyHostToDevice, stream_));
        HIP_CHECK(hipEventRecord(copied, stream_));
    }
    uint8_t* dev = img_dev_.get() + (size_t)slot * kImageSize * kImageSize * 4;
    k::ResizeAxis kx{ax.first.get(), ax.count.get(), ax.coef.get(), ax.taps, rw};
    k::ResizeAxis ky{ay.first.get(), ay.count.get(), ay.coef.get(), ay.taps, rh};
    timed(ST_PRE, (double)row * h + (double)rw * rh * bytes, [&] {
        k::resize_srgb(resize_src_.get(), w, h, (int)row, bytes, kx, ky, srgb_decode_.get(), srgb_encode_.get(),
                       resize_tmp_.get(), dev, stream_);
    });
    preprocess_device_image(slot, batch, dev, rw, rh, rw * bytes, channels);
}

void SamModel::encode(int batch, float* const* emb_dst) {
    SamWeights const& W = *weights_;
    DLIMG_ASSERT(batch > 0 && batch <= enc_batch_);
    // one image, and no other lane of this GPU has anything in flight: the pass may trade CU time for latency
    alone_ = batch == 1 && shared_gpu_ && board_ && board_->others_idle(lane_index_);
    if (board_ && batch == 1) board_->count_pass(alone_);
    begin_activity();                            // from here on this lane counts as busy for its siblings
    struct EndOnExit {                           // (an enqueue that throws must not leave the lane "busy" for ever)
        LaneBoard* board; int lane;
        ~EndOnExit() { if (board) board->end(lane); }
    } end_on_exit{board_.get(), lane_index_};
    const int D = W.geom_.embed_dim, H = W.geom_.num_heads, hd = W.geom_.head_dim(), mlp = W.geom_.mlp_dim;
    const int M = batch * kTokens;

    // Block structure: x += proj(attn(qkv(LN1(x)))); x += fc2(gelu(fc1(LN2(x)))).  With folded LayerNorms every
    // GEMM that writes the residual stream x (fp32) also leaves its f16 copy and, per tile column block, the row
    // statistics of what it wrote; the next GEMM multiplies the f16 copy by the gamma-scaled weight, merges the
    // statistics of its rows and normalises in its epilogue: the stream is read once (as f16) per consumer instead
    // of LN read + LN write + GEMM read.
    const bool fused = W.fused_ln_;
    int stat_groups = 0;                         // tile column blocks of the GEMM that last wrote the stream
    const bool split = split_stream_;
    auto stream_residual = [&](k::GemmArgs& a) {                 // += the stream itself, in place
        if (split) { a.resid_h = xn_.get(); a.resid_l = xlo_.get(); a.ldrs = D; }
        else { a.resid = x_.get(); a.ldr = D; }
        a.resid_mod = M;
    };
    auto writes_stream = [&](k::GemmArgs& a) {
        a.M = M; a.N = D;
        if (split) { a.out_l = xlo_.get(); }
        else { a.out_f32 = x_.get(); a.ldc32 = D; }
        if (fused) {
            a.out_h = xn_.get(); a.ldc16 = D; a.stats_out = xstat_.get();
            a.shared_gpu = shared_gpu_;
            a.alone = alone_;
            a.unit_rows = kTokens;
            stat_groups = D / k::gemm_choose_tile(a);    // the launch below uses exactly this tile (a.tile)
        }
    };
    auto reads_stream = [&](k::GemmArgs& a, LinearH const& lin, NormW const& norm) {
        if (fused) {
            a.ln_stats = xstat_.get(); a.ln_groups = stat_groups; a.ln_colsum = lin.colsum.get(); a.ln_eps = kLnEps;
        } else {
            timed(ST_LAYERNORM, (double)M * D * 6, [&] {
                k::layernorm(x_.get(), norm.w.get(), norm.b.get(), kLnEps, M, D, k::ACT_NONE, nullptr, xn_.get(), stream_);
            });
        }
        a.A = xn_.get(); a.lda = D; a.W = lin.w.get(); a.ldw = D; a.bias = lin.b.get(); a.M = M; a.N = lin.out; a.K = D;
    };

    k::GemmArgs g;
    g.A = patches_.get(); g.lda = kPatchK; g.W = W.patch_.w.get(); g.ldw = kPatchK; g.bias = W.patch_.b.get();
    g.resid = W.pos_embed_.get(); g.ldr = D; g.resid_mod = kTokens; g.K = kPatchK;
    writes_stream(g);
    gemm(g, ST_GEMM_PATCH);

    for (EncoderLayer const& L : W.layers_) {
        g = k::GemmArgs{};
        reads_stream(g, L.qkv, L.ln1);
        g.out_h = qkv_.get(); g.ldc16 = 3 * D;
        gemm(g);
        if (L.global) {
            const double fl = (double)batch * (4.0 * kTokens * (double)kTokens * D + 4.0 * kTokens * 64.0 * hd * H);
            timed(ST_ATTN_GLOBAL, fl, [&] {
                k::attention_global(qkv_.get(), L.rel_h16.get(), L.rel_w16.get(), att_.get(), batch, H, hd, stream_);
            });
        } else {
            const double fl = (double)batch * 25.0 * (4.0 * 196.0 * 196.0 * D + 4.0 * 196.0 * 14.0 * hd * H);
            timed(ST_ATTN_WINDOW, fl, [&] {
                k::attention_window(qkv_.get(), L.qkv_pad.get(), L.rel_h16.get(), L.rel_w16.get(), att_.get(), batch, H, hd,
                                    stream_);
            });
        }
        g = k::GemmArgs{};
        g.A = att_.get(); g.lda = D; g.W = L.proj.w.get(); g.ldw = D; g.bias = L.proj.b.get(); g.K = D;
        stream_residual(g);
        writes_stream(g);
        gemm(g, ST_GEMM_PROJ);
        g = k::GemmArgs{};
        reads_stream(g, L.fc1, L.ln2);
        g.act = k::ACT_GELU; g.out_h = hid_.get(); g.ldc16 = mlp;
        gemm(g);
        g = k::GemmArgs{};
        g.A = hid_.get(); g.lda = mlp; g.W = L.fc2.w.get(); g.ldw = mlp; g.bias = L.fc2.b.get(); g.K = mlp;
        stream_residual(g);
        writes_stream(g);
        gemm(g, ST_GEMM_FC2);
    }

    // neck: 1x1 conv -> LayerNorm2d -> 3x3 conv (pad 1) -> LayerNorm2d, all channel-last
    if (!fused) {
        timed(ST_ENC_OTHER, (double)M * D * 6, [&] {
            k::add_cast(x_.get(), nullptr, 0, (size_t)M * D, nullptr, xn_.get(), stream_);
        });
    }
    g = k::GemmArgs{};
    g.A = xn_.get(); g.lda = D; g.W = W.neck1_.w.get(); g.ldw = D;
    g.out_f32 = neck_f32_.get(); g.ldc32 = kEmbedDim; g.M = M; g.N = kEmbedDim; g.K = D;
    gemm(g);
    timed(ST_LAYERNORM, (double)M * kEmbedDim * 6, [&] {
        k::layernorm(neck_f32_.get(), W.neck_ln1_.w.get(), W.neck_ln1_.b.get(), kLnEps, M, kEmbedDim, k::ACT_NONE, nullptr,
                     att_.get(), stream_);
    });
    timed(ST_ENC_OTHER, (double)M * kEmbedDim * 2 * 10, [&] {
        k::im2col3x3(att_.get(), batch, kEmbedDim, hid_.get(), stream_);
    });
    g = k::GemmArgs{};
    g.A = hid_.get(); g.lda = 9 * kEmbedDim; g.W = W.neck2_.w.get(); g.ldw = 9 * kEmbedDim;
    g.out_f32 = neck_f32_.get(); g.ldc32 = kEmbedDim; g.M = M; g.N = kEmbedDim; g.K = 9 * kEmbedDim;
    gemm(g);
    // the embedding goes straight into the handle's storage when there is one image; a batch is copied out per image
    float* direct = (emb_dst && batch == 1 && emb_dst[0]) ? emb_dst[0] : nullptr;
    // f16 operands and the f16 residual pair overflow to infinity beyond 65504; an infinity anywhere in an image turns
    // into NaNs that reach this LayerNorm's input, which reports it (last_pass_flag) so that the caller refuses the
    // embedding instead of decoding masks from it
    if (!pass_flags_) {
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&pass_flags_), kPassFlags * sizeof(int), hipHostMallocDefault));
        std::memset(pass_flags_, 0, kPassFlags * sizeof(int));
    }
    const unsigned flag_slot = pass_counter_++ % kPassFlags;
    if (flag_owner_[flag_slot]) {                // a deferred pass of kPassFlags passes ago still reports through this flag
        flag_owner_[flag_slot]->settle();
        flag_owner_[flag_slot].reset();
    }
    pass_flag_ = pass_flags_ + flag_slot;
    *pass_flag_ = 0;
    timed(ST_LAYERNORM, (double)M * kEmbedDim * 8, [&] {
        k::layernorm(neck_f32_.get(), W.neck_ln2_.w.get(), W.neck_ln2_.b.get(), kLnEps, M, kEmbedDim, k::ACT_NONE,
                     direct ? direct : emb_.get(), nullptr, stream_, pass_flag_);
    });
    if (emb_dst && !direct) {
        const size_t n = (size_t)kTokens * kEmbedDim;
        for (int i = 0; i < batch; ++i)
            if (emb_dst[i])
                HIP_CHECK(hipMemcpyAsync(emb_dst[i], emb_.get() + i * n, n * sizeof(float), hipMemcpyDeviceToDevice, stream_));
    }
    HIP_CHECK(hipGetLastError());     // a refused launch (bad grid, LDS size) is reported here, not at a later sync
    mark_activity();                  // behind the last kernel of the pass
}

// ---------------------------------------------------------------------------------------------
// prompt encoder + mask decoder

void SamModel::reserve_decoder(int count) {
    if (count <= dec_count_) return;
    HIP_CHECK(hipStreamSynchronize(stream_));
    const size_t P = count, M = P * kTokens;
    keys_.reserve(M * 256);
    keys_h_.reserve(M * 256);
    kqv_h_.reserve(M * 384);
    logits_.reserve(P * 4 * kLowRes * kLowRes);
    iou_.reserve(P * 4);
    hyper_.reserve(P * 4 * 32);
    const size_t T = P * kDecTokens;
    tokens_.reserve(T * 256);
    queries_.reserve(T * 256);
    tk_.reserve(T * 256);
    tv_.reserve(T * 256);
    sq_.reserve(T * 256);
    sk_.reserve(T * 256);
    sv_.reserve(T * 256);
    tsa_.reserve(T * 256);
    tt2i_.reserve(T * 256);
    t2i_part_.reserve(k::token_to_image_scratch_floats((int)P));
    tmlp_.reserve(T * 2048);
    dec_count_ = count;
}

void SamModel::decode(float const* const* emb, float const* coords, float const* labels, int count) {
    DLIMG_ASSERT(count > 0);
    reserve_decoder(count);
    // the token-side kernels take at most 16 prompts (112 rows) per launch: larger requests run in chunks that share
    // the workspaces (stream order) and write their own part of logits() / iou()
    constexpr int kChunk = 16;
    for (int c0 = 0; c0 < count; c0 += kChunk)
        decode_chunk(emb + c0, coords + (size_t)c0 * 4, labels + (size_t)c0 * 2, std::min(kChunk, count - c0), c0);
    mark_activity();
}

void SamModel::decode_chunk(float const* const* emb, float const* coords, float const* labels, int count, int first) {
    SamWeights const& W = *weights_;
    const int P = count, M = P * kTokens, T = P * kDecTokens;
    hipStream_t s = stream_;
    float* logits_out = logits_.get() + (size_t)first * 4 * kLowRes * kLowRes;
    float* iou_out = iou_.get() + (size_t)first * 4;

    auto body = [&] {
        // prompts travel as kernel arguments of the first launch
        k::DecoderPrompts prompts{};
        std::memcpy(prompts.coords, coords, (size_t)P * 4 * sizeof(float));
        std::memcpy(prompts.labels, labels, (size_t)P * 2 * sizeof(float));
        for (int i = 0; i < P; ++i) prompts.emb[i] = emb[i];

        // Token side.  `cur` is the running token matrix as its consumers read it: un-normalised rows plus the
        // LayerNorm that belongs in front of them (applied on the fly by whoever reads, kernels/decoder.hip).
        float const* qpe = tokens_.get();
        k::TokenRows cur;
        auto rows_of = [&](k::TokenRows r, bool with_pe) { if (with_pe) r.add = qpe; return r; };
        auto lin = [&](k::TokenRows in, int K, LinearF const& l, k::TokenRows resid, float* Y, int relu) {
            k::TokenLinear op;
            op.in = in; op.K = K; op.W = l.w.get(); op.b = l.b.get(); op.resid = resid; op.Y = Y; op.N = l.out; op.relu = relu;
            return op;
        };
        auto plain = [&](float const* x) { k::TokenRows r; r.x = x; return r; };
        auto normed = [&](float const* x, NormW const& n) {
            k::TokenRows r; r.x = x; r.ln_w = n.w.get(); r.ln_b = n.b.get(); r.eps = kDecLnEps; return r;
        };
        // image side of the attentions: all projections of the keys in one MFMA GEMM, the positional part of
        // (keys + pos) W as the constant addend SamWeights prepared
        auto img_gemm_args = [&](LinearH const& l, DeviceBuffer<float> const& pos) {
            k::GemmArgs g;
            g.A = keys_h_.get(); g.lda = 256; g.W = l.w.get(); g.ldw = 256; g.bias = l.b.get();
            g.resid = pos.get(); g.ldr = l.out; g.resid_mod = kTokens;
            g.out_h = kqv_h_.get(); g.ldc16 = l.out; g.M = M; g.N = l.out; g.K = 256;
            g.shared_gpu = shared_gpu_;
            g.unit_rows = kTokens;
            return g;
        };
        auto img_gemm = [&](LinearH const& l, DeviceBuffer<float> const& pos) { k::gemm(img_gemm_args(l, pos), s); };

        // First launch: prompt tokens (= the positional part `qpe` of every later query), keys = image_embedding +
        // no_mask_embed (has_mask_input == 0, segmentation.cpp:43-45), and the q / k / v projections of the first
        // self-attention, whose input ARE the prompt tokens (no PE, no LayerNorm in front of the first block).
        {
            DecoderLayer const& L = W.dec_[0];
            k::TokenLinear qkv[3] = {lin({}, 256, L.self_attn.q, {}, sq_.get(), 0), lin({}, 256, L.self_attn.k, {}, sk_.get(), 0),
                                     lin({}, 256, L.self_attn.v, {}, sv_.get(), 0)};
            k::decoder_start(prompts, W.pe_gauss_.get(), W.pe_point_.get(), W.pe_not_a_point_.get(), W.iou_token_.get(),
                             W.mask_tokens_.get(), tokens_.get(), qkv, 3, W.pe_no_mask_.get(), keys_.get(), keys_h_.get(), P, s);
        }
        for (int i = 0; i < 2; ++i) {
            DecoderLayer const& L = W.dec_[i];
            // (1) self attention of the tokens; the first layer has no residual.  The q / k / v projections were computed
            // by the launch that produced their input rows (decoder_start, or the first layer's step (4)).
            // (2) tokens -> image: [K | Q of step 4 | V] = [(keys + pos) Wk | (keys + pos) Wq | keys Wv]; the query
            // projection runs inside the attention launch, the fold + output projection inside the MLP's first launch.
            // The projection of the keys depends on (1) as little as (1) on it: its tiles ride in (1)'s launch (r06)
            const k::TokenLinear self_out = lin({}, 256, L.self_attn.o, i == 0 ? k::TokenRows{} : cur, tsa_.get(), 0);
            // (DLIMGEDIT_DECODER_RIDE=0: measurement aid, the two launches on their own)
            static const bool ride = [] { const char* e = std::getenv("DLIMGEDIT_DECODER_RIDE"); return !e || std::atoi(e) != 0; }();
            if (!ride || !k::token_self_attention_out_with_gemm(sq_.get(), sk_.get(), sv_.get(), self_out, P, img_gemm_args(L.img_kqv, L.pos_kqv), s)) {
                k::token_self_attention_out(sq_.get(), sk_.get(), sv_.get(), self_out, P, s);
                img_gemm(L.img_kqv, L.pos_kqv);
            }
            const k::TokenRows q1 = normed(tsa_.get(), L.ln1);
            const k::TokenLinear tq = lin(rows_of(q1, true), 256, L.t2i_q, {}, nullptr, 0);
            k::token_to_image_partials(nullptr, &tq, kqv_h_.get(), 384, kqv_h_.get() + 256, 384, t2i_part_.get(), P, s);
            const k::TokenRows q2 = normed(tt2i_.get(), L.ln2);
            // (3) token MLP
            k::token_merge_linear(t2i_part_.get(), lin({}, 128, L.t2i_o, q1, tt2i_.get(), 0), L.t2i_o_t.get(),
                                  lin(q2, 256, L.mlp1, {}, tmlp_.get(), 1), P, s);
            k::TokenLinear m2 = lin(plain(tmlp_.get()), 2048, L.mlp2, q2, queries_.get(), 0);
            k::token_linears(&m2, 1, T, s);
            const k::TokenRows q3 = normed(queries_.get(), L.ln3);
            // (4) image -> tokens.  Everything else that reads the same rows q3 rides in this launch: the next layer's
            // self-attention projections, or (last layer) the query projection of the final token -> image attention.
            k::TokenLinear kv[5] = {lin(rows_of(q3, true), 256, L.i2t_k, {}, tk_.get(), 0),
                                    lin(q3, 256, L.i2t_v, {}, tv_.get(), 0)};
            int n_ops = 2;
            if (i == 0) {
                DecoderLayer const& N = W.dec_[1];
                kv[n_ops++] = lin(rows_of(q3, true), 256, N.self_attn.q, {}, sq_.get(), 0);
                kv[n_ops++] = lin(rows_of(q3, true), 256, N.self_attn.k, {}, sk_.get(), 0);
                kv[n_ops++] = lin(q3, 256, N.self_attn.v, {}, sv_.get(), 0);
            } else {
                kv[n_ops++] = lin(rows_of(q3, true), 256, W.final_q_, {}, sq_.get(), 0);
            }
            k::token_linears(kv, n_ops, T, s);
            k::image_update(kqv_h_.get() + 128, 384, tk_.get(), tv_.get(), L.i2t_o.w.get(), L.i2t_o.b.get(), L.ln4.w.get(),
                            L.ln4.b.get(), kDecLnEps, keys_.get(), keys_h_.get(), P, s);
            cur = q3;
        }
        // final token -> image attention; its fold + output projection + norm_final_attn happen in output_heads
        img_gemm(W.final_kv_, W.final_pos_kv_);
        k::token_to_image_partials(sq_.get(), nullptr, kqv_h_.get(), 256, kqv_h_.get() + 128, 256, t2i_part_.get(), P, s);

        k::HeadWeights hw;
        for (int m = 0; m < 5; ++m)
            for (int j = 0; j < 3; ++j) {
                hw.w[m][j] = W.heads_[m][j].w.get();
                hw.b[m][j] = W.heads_[m][j].b.get();
            }
        k::output_heads(t2i_part_.get(), lin({}, 128, W.final_o_, cur, nullptr, 0), W.final_o_t_.get(), normed(nullptr, W.ln_final_),
                        hw, hyper_.get(), iou_out, P, s);
        // upscaling ConvT(256->64) -> LN2d -> GELU -> ConvT(64->32) -> GELU and the product with the hyper vectors
        k::upscale_logits(keys_h_.get(), W.up1_.w.get(), W.up1_.b.get(), W.up_ln_.w.get(), W.up_ln_.b.get(), kLnEps,
                          W.up2_.w.get(), W.up2_.b.get(), hyper_.get(), logits_out, P, s);
    };
    timed(ST_DECODER, 3.62e9 * P, body);
    HIP_CHECK(hipGetLastError());
}

std::vector<std::pair<const char*, size_t>> SamModel::decoder_state_layout() {
    const size_t T = kDecTokens;
    return {{"tokens", T * 256}, {"final_q", T * 128}, {"self_k", T * 256}, {"self_v", T * 256}, {"self_out", T * 256},
            {"t2i_out", T * 256}, {"mlp_hidden", T * 2048}, {"queries", T * 256}, {"i2t_k", T * 128}, {"i2t_v", T * 128},
            {"final_partials", k::token_to_image_scratch_floats(1)}, {"hyper", 4 * 32}, {"iou", 4}, {"keys_head", 4096}};
}

void SamModel::decoder_state(float* out) const {
    float const* src[] = {tokens_.get(), sq_.get(), sk_.get(), sv_.get(), tsa_.get(), tt2i_.get(), tmlp_.get(), queries_.get(),
                          tk_.get(), tv_.get(), t2i_part_.get(), hyper_.get(), iou_.get(), keys_.get()};
    size_t off = 0, i = 0;
    for (auto const& part : decoder_state_layout()) {
        HIP_CHECK(hipMemcpy(out + off, src[i++], part.second * sizeof(float), hipMemcpyDeviceToHost));
        off += part.second;
    }
}

void SamModel::masks_on_device(k::PostJob const* jobs, int count) {
    if (count <= 0) return;
    double bytes = 0;
    for (int i = 0; i < count; ++i) bytes += (double)kLowRes * kLowRes * 4 + (double)jobs[i].out_w * jobs[i].out_h;
    timed(ST_POST, bytes, [&] { k::postprocess_masks(jobs, count, stream_); });
    mark_activity();
}

SamModel::MaskSlot& SamModel::acquire_mask_slot() {
    {
        std::lock_guard<std::mutex> lock(done_mutex_);
        if (!mask_free_.empty()) {
            MaskSlot* s = mask_free_.back();
            mask_free_.pop_back();
            return *s;
        }
    }
    // as many slots come into being as there are mask requests in flight on this lane at once
    auto fresh = std::make_unique<MaskSlot>();
    HIP_CHECK(hipSetDevice(device_));
    HIP_CHECK(hipEventCreateWithFlags(&fresh->done, hipEventDisableTiming));
    std::lock_guard<std::mutex> lock(done_mutex_);
    mask_slots_.push_back(std::move(fresh));
    return *mask_slots_.back();
}

void SamModel::release_mask_slot(MaskSlot& s) {
    std::lock_guard<std::mutex> lock(done_mutex_);
    mask_free_.push_back(&s);
}

static size_t mask_bytes(k::PostJob const& j) { return padded_mask_bytes((size_t)j.out_w * j.out_h); }

void SamModel::enqueue_masks(MaskSlot& slot, k::PostJob const* jobs, int count, int iou_count) {
    if (count <= 0) return;
    size_t total = 0;
    for (int i = 0; i < count; ++i) total += mask_bytes(jobs[i]);
    slot.iou_offset = total;
    const size_t with_iou = total + (size_t)iou_count * sizeof(float);
    // the slot is ours, and its previous user waited for the slot's event before letting go of it
    slot.dev.reserve(with_iou);
    slot.pin.reserve(with_iou);
    std::vector<k::PostJob> dev_jobs(jobs, jobs + count);
    size_t off = 0;
    double bytes = 0;
    // One mask, or up to six while no other lane of this GPU has work in flight: the post-processing kernel writes each
    // mask STRAIGHT into the pinned host staging memory, one launch and one event per mask -- the stores leave over PCIe
    // while the kernel runs, and the host copies mask i to the caller's buffer while mask i + 1 is being written.
    // [r04: one mask this way instead of a device buffer plus a copy command, 0.330 -> 0.309 ms per compute_mask call.
    // r06: several masks too -- by the kernel trace a five-prompt call spent 265 us between its last kernel and the next
    // call, 42 % of the call, on five ~1 MiB copy commands and their events: one caller 7299 -> 8106 prompts/s.  But a
    // kernel that waits for PCIe holds its lane's stream and its CUs meanwhile, where a copy command runs beside the next
    // kernels: with FOUR callers the same change cost 21 900 -> 18 300 prompts/s, hence the idle-lanes condition -- the
    // LaneBoard's answer, as for the encoder's tiles.]  Otherwise: device buffer + piecewise copy.
    // DLIMGEDIT_DIRECT_MASKS=0: measurement aid (always the copy path).
    static const bool direct_allowed = [] { const char* e = std::getenv("DLIMGEDIT_DIRECT_MASKS"); return !e || std::atoi(e) != 0; }();
    constexpr int kDirectMasks = 6;
    const bool direct = direct_allowed && (count == 1 || (count <= kDirectMasks && (!board_ || board_->others_idle(lane_index_))));
    slot.in_place.assign(count, 0);
    if (direct) {
        uint8_t* pin = static_cast<uint8_t*>(slot.pin.get());
        slot.piece_end.clear();
        for (int i = 0; i < count; ++i) {
            if ((int)slot.piece_done.size() <= i) {
                hipEvent_t e = nullptr;
                HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                slot.piece_done.push_back(e);
            }
            // a destination in pinned image memory of the library (the Image the reference's wrapper allocates for the
            // result through create_image): written where the consumer reads it
            slot.in_place[i] = image_memory_is_pinned(jobs[i].dst, (size_t)jobs[i].out_w * jobs[i].out_h);
            dev_jobs[i].dst = slot.in_place[i] ? jobs[i].dst : pin + off;
            off += mask_bytes(jobs[i]);
            bytes = (double)kLowRes * kLowRes * 4 + (double)jobs[i].out_w * jobs[i].out_h;
            timed(ST_POST, bytes, [&] { k::postprocess_masks(&dev_jobs[i], 1, stream_); });
            if (i + 1 == count && iou_count > 0)
                HIP_CHECK(hipMemcpyAsync(pin + total, iou_.get(), (size_t)iou_count * sizeof(float), hipMemcpyDeviceToHost, stream_));
            HIP_CHECK(hipEventRecord(slot.piece_done[i], stream_));
            slot.piece_end.push_back(i + 1 == count ? with_iou : off);
        }
        HIP_CHECK(hipEventRecord(slot.done, stream_));
        mark_activity();
        return;
    }
    for (int i = 0; i < count; ++i) {
        dev_jobs[i].dst = slot.dev.get() + off;
        off += mask_bytes(jobs[i]);
        bytes += (double)kLowRes * kLowRes * 4 + (double)jobs[i].out_w * jobs[i].out_h;
    }
    timed(ST_POST, bytes, [&] { k::postprocess_masks(dev_jobs.data(), count, stream_); });
    if (iou_count > 0)
        HIP_CHECK(hipMemcpyAsync(slot.dev.get() + total, iou_.get(), (size_t)iou_count * sizeof(float),
                                 hipMemcpyDeviceToDevice, stream_));
    // every destination is pinned image memory of the library (Images of the reference's wrapper): one copy command per mask
    // from the device buffer to where the consumer reads it, nothing for the host to copy afterwards
    bool all_pinned = true;
    for (int i = 0; i < count && all_pinned; ++i)
        all_pinned = image_memory_is_pinned(jobs[i].dst, (size_t)jobs[i].out_w * jobs[i].out_h);
    if (all_pinned) {
        slot.in_place.assign(count, 1);
        size_t from = 0;
        for (int i = 0; i < count; ++i) {
            HIP_CHECK(hipMemcpyAsync(jobs[i].dst, slot.dev.get() + from, (size_t)jobs[i].out_w * jobs[i].out_h, hipMemcpyDeviceToHost, stream_));
            from += mask_bytes(jobs[i]);
        }
        if (iou_count > 0)
            HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t*>(slot.pin.get()) + total, slot.dev.get() + total,
                                     (size_t)iou_count * sizeof(float), hipMemcpyDeviceToHost, stream_));
        if (slot.piece_done.empty()) {
            hipEvent_t e = nullptr;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            slot.piece_done.push_back(e);
        }
        slot.piece_end.assign(1, with_iou);
        HIP_CHECK(hipEventRecord(slot.piece_done[0], stream_));
        HIP_CHECK(hipEventRecord(slot.done, stream_));
        mark_activity();
        return;
    }
    // device -> pinned host in pieces, each with its own event (mask_pieces.hpp)
    slot.piece_end = mask_piece_ends(with_iou);
    size_t a = 0;
    for (size_t i = 0; i < slot.piece_end.size(); ++i) {
        const size_t b = slot.piece_end[i];
        if (slot.piece_done.size() <= i) {
            hipEvent_t e = nullptr;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            slot.piece_done.push_back(e);
        }
        HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t*>(slot.pin.get()) + a, slot.dev.get() + a, b - a, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipEventRecord(slot.piece_done[i], stream_));
        a = b;
    }
    HIP_CHECK(hipEventRecord(slot.done, stream_));
    mark_activity();
}

void SamModel::finish_masks(MaskSlot& slot, k::PostJob const* jobs, int count, float* iou_out, int iou_count) {
    if (count <= 0) return;
    static const bool trace = std::getenv("DLIMGEDIT_TIMING") != nullptr;     // diagnostic: host time of the two phases
    const auto t0 = std::chrono::steady_clock::now();
    uint8_t const* pin = static_cast<uint8_t const*>(slot.pin.get());
    // piece by piece: what has arrived is copied to the callers' buffers while the rest is still on its way
    double waited_us = 0;
    std::vector<size_t> sizes(count);
    for (int i = 0; i < count; ++i) sizes[i] = (size_t)jobs[i].out_w * jobs[i].out_h;
    MaskCursor cursor;
    size_t begin = 0;
    for (size_t i = 0; i < slot.piece_end.size(); ++i) {
        const auto w0 = std::chrono::steady_clock::now();
        HIP_CHECK(hipEventSynchronize(slot.piece_done[i]));
        waited_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        for (MaskCopy const& c : mask_copies_in_piece(sizes, begin, slot.piece_end[i], cursor))
            if (!slot.in_place[c.mask]) std::memcpy(jobs[c.mask].dst + c.mask_offset, pin + c.staging_offset, c.bytes);
        begin = slot.piece_end[i];
    }
    HIP_CHECK(hipEventSynchronize(slot.done));
    const auto t1 = std::chrono::steady_clock::now();
    size_t off = 0;
    for (int i = 0; i < count; ++i) off += mask_bytes(jobs[i]);
    if (iou_out && iou_count > 0) std::memcpy(iou_out, pin + slot.iou_offset, (size_t)iou_count * sizeof(float));
    if (trace)
        std::fprintf(stderr, "finish_masks: %.1f us in all, %.1f us of them waiting for the %zu pieces (%zu bytes)\n",
                     std::chrono::duration<double, std::micro>(t1 - t0).count(), waited_us, slot.piece_end.size(), off);
}

namespace {
// Direct xGMI copies between two GPUs need peer access switched on once per direction (without it the runtime stages
// the copy through host memory: still correct, slower).  Failures are not errors: the copy falls back by itself.
void enable_peer_access(int from_device, int to_device) {
    static std::mutex m;
    static std::vector<std::pair<int, int>> done;
    std::lock_guard<std::mutex> lock(m);
    for (auto& d : done)
        if (d.first == from_device && d.second == to_device) return;
    done.emplace_back(from_device, to_device);
    if (from_device == to_device) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, from_device, to_device) == hipSuccess && can) {
        (void)hipSetDevice(from_device);
        if (hipDeviceEnablePeerAccess(to_device, 0) != hipSuccess) (void)hipGetLastError();
    }
    if (hipDeviceCanAccessPeer(&can, to_device, from_device) == hipSuccess && can) {
        (void)hipSetDevice(to_device);
        if (hipDeviceEnablePeerAccess(from_device, 0) != hipSuccess) (void)hipGetLastError();
    }
    (void)hipSetDevice(prev);
}
}  // namespace

void SamModel::enqueue_masks_device(MaskSlot& slot, k::PostJob const* jobs, int count, int dst_device) {
    if (count <= 0) return;
    // test hook: take the staging + peer-copy path even when the destination is this lane's own GPU (a one-GPU box
    // has no second device to copy to; the path is the same code, the copy degenerates to device-to-device)
    const char* fp = std::getenv("DLIMGEDIT_FORCE_PEER_COPY");       // read per call: the tests switch it on and off
    const bool force_peer = fp && std::atoi(fp) != 0;
    double bytes = 0;
    for (int i = 0; i < count; ++i) bytes += (double)kLowRes * kLowRes * 4 + (double)jobs[i].out_w * jobs[i].out_h;
    if (dst_device == device_ && !force_peer) {
        timed(ST_POST, bytes, [&] { k::postprocess_masks(jobs, count, stream_); });
    } else {
        size_t total = 0;
        for (int i = 0; i < count; ++i) total += mask_bytes(jobs[i]);
        slot.dev.reserve(total);             // the slot is ours; its previous user waited for slot.done
        std::vector<k::PostJob> staged(jobs, jobs + count);
        size_t off = 0;
        for (int i = 0; i < count; ++i) {
            staged[i].dst = slot.dev.get() + off;
            off += mask_bytes(jobs[i]);
        }
        timed(ST_POST, bytes, [&] { k::postprocess_masks(staged.data(), count, stream_); });
        enable_peer_access(device_, dst_device);
        for (int i = 0; i < count; ++i)
            HIP_CHECK(hipMemcpyPeerAsync(jobs[i].dst, dst_device, staged[i].dst, device_,
                                         (size_t)jobs[i].out_w * jobs[i].out_h, stream_));
    }
    HIP_CHECK(hipEventRecord(slot.done, stream_));
}

void SamModel::wait_masks(MaskSlot& slot) { HIP_CHECK(hipEventSynchronize(slot.done)); }

void SamModel::masks_to_host(k::PostJob const* jobs, int count) {
    MaskSlot& slot = acquire_mask_slot();
    try {
        enqueue_masks(slot, jobs, count, 0);
        finish_masks(slot, jobs, count, nullptr, 0);
    } catch (...) {
        release_mask_slot(slot);
        throw;
    }
    release_mask_slot(slot);
}

}  // namespace dlimg
