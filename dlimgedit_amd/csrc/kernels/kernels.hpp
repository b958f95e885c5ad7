// Host-callable launchers of the gfx950 kernels.  All pointers are device pointers unless noted;
// every launcher validates the shapes its grid assumes and throws dlimg::Exception on mismatch
// (a faulting kernel can take the whole node down, so nothing is launched on unchecked shapes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dlimg {

typedef _Float16 half_t;

[[noreturn]] void throw_error(const char* msg);   // defined in common.cpp

namespace k {

// Opt-in of one kernel to more than 64 KB of dynamic LDS.  hipFuncAttributeMaxDynamicSharedMemorySize belongs to the
// function ON THE CURRENT DEVICE, and an environment may hold one replica per GPU (DLIMGEDIT_DEVICES), so the state is
// kept per device: the first launcher on each GPU sets the attribute, every later one only reads the outcome.  A refused
// opt-in is remembered and reported by every caller (launching anyway would fail with an invalid-configuration error
// or, worse, run with less LDS than the kernel addresses).  One static instance per kernel at its launch site.
class LdsOptIn {
  public:
    static constexpr int kMaxDevices = 64;
    void ensure(const void* kernel, size_t bytes, const char* refused_message) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) throw_error("kernel launch: no current HIP device");
        int st = __atomic_load_n(&state_[dev], __ATOMIC_ACQUIRE);
        if (st == 0) {
            // several lanes may arrive together: setting the same value twice is harmless, the outcome is the same
            st = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 1 : -1;
            if (st < 0) (void)hipGetLastError();
            __atomic_store_n(&state_[dev], st, __ATOMIC_RELEASE);
        }
        if (st < 0) throw_error(refused_message);
    }

  private:
    int state_[kMaxDevices] = {};        // 0 = not tried on this device, 1 = granted, -1 = refused
};

enum Act { ACT_NONE = 0, ACT_GELU = 1 };

// ---- GEMM ------------------------------------------------------------------------------------
struct GemmArgs {
    const half_t* A = nullptr;  int lda = 0;     // [M,K] row-major f16
    const half_t* W = nullptr;  int ldw = 0;     // [N,K] row-major f16 (nn.Linear layout)
    const float* bias = nullptr;                 // [N] or null
    const float* resid = nullptr; int ldr = 0; int resid_mod = 1;   // added after activation: resid[m % resid_mod][n]
    float* out_f32 = nullptr;   int ldc32 = 0;
    half_t* out_h = nullptr;    int ldc16 = 0;
    // The residual stream as an f16 PAIR (r04): value = hi + lo, hi = f16(value) -- which is also the A operand of the GEMM
    // that consumes the stream -- and lo = f16(value - hi): 22 significant bits in 4 bytes.  A stream writer then moves 8 bytes
    // per element (pair in, pair out) instead of 10 (fp32 in, fp32 + f16 copy out).  resid_h / resid_l: the residual as a
    // pair (instead of `resid`; rows m % resid_mod, leading dimension ldrs); out_l with out_h: the result as a pair (instead
    // of out_f32; both with leading dimension ldc16).  Ping-pong tiles (9, 10, 11) only.
    const half_t* resid_h = nullptr; const half_t* resid_l = nullptr; int ldrs = 0;
    half_t* out_l = nullptr;
    int M = 0, N = 0, K = 0;
    int act = ACT_NONE;
    // LayerNorm folded into the GEMMs around it (gemm.hip).  Producer: stats_out receives, per tile column block and
    // row, the (sum, sum of squared deviations) of the fp32 result -- [N / BN][M][2] (BN from gemm_choose_tile).  Consumer: A is
    // the raw, un-normalised input, W carries the LayerNorm scale (W * diag(gamma)), bias carries b + W.beta; ln_stats
    // are the producer's partials ([ln_groups][M][2], groups of K / ln_groups columns) and the epilogue applies
    // rstd_m * (acc - mean_m * ln_colsum[n]) + bias[n].
    float* stats_out = nullptr;
    const float* ln_stats = nullptr;
    int ln_groups = 0;
    const float* ln_colsum = nullptr;            // [N] sum over k of the (f16-rounded) scaled weight
    float ln_eps = 0.f;
    // Tile selection.  shared_gpu: kernels of several execution lanes share the device (prefer tiles that can share a
    // CU).  tile: -1 = let gemm() choose; gemm_choose_tile() fixes the choice in the arguments so that a caller who
    // needs the tile width beforehand (stats_out layout) and the launch agree by construction.
    bool shared_gpu = false;
    // alone: the pass this GEMM belongs to has the GPU to itself (no other lane has work in flight: a synchronous caller).
    // A one-image launch may then use tiles that double its workgroups at a worse cost per FLOP (gemm.hip, tile 11);
    // the bits do not depend on it.
    bool alone = false;
    int tile = -1;
    // Rows of ONE independent unit (an image: 4096 tokens) when M stacks several of them.  The tile is chosen for the
    // unit's shape, so a batch runs the same tiles -- and produces the same bits -- as its images one at a time.
    int unit_rows = 0;
    // Diagnostics (tuning builds of the benchmark hook only; null in the product): per workgroup 4 x u64 =
    // { shader cycles, 100 MHz ticks } of the main loop and of the whole kernel, written by wave 0.
    unsigned long long* stamps = nullptr;
};
const char* gemm_check(const GemmArgs&);
bool gemm_tile_fits(const GemmArgs&, int tile);   // may `tile` (index into gemm.hip's table) run this problem?
int gemm_choose_tile(GemmArgs&);              // sets a.tile; returns BN (columns per tile) of that configuration
// start/stop (both or neither): events attached to the kernel's own dispatch (hipExtLaunchKernelGGL): their elapsed time
// is the kernel's execution time as a profiler reports it, without the marker packets an hipEventRecord pair adds.
void gemm(const GemmArgs&, hipStream_t, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);

// ---- row LayerNorm ---------------------------------------------------------------------------
// y = (x-mean)/sqrt(var+eps)*w+b over rows of length D (<= 1280); optional GELU; f32 and/or f16 out.
// x and out_f32 may alias.  nonfinite (optional, D a multiple of 256): an int in memory the kernel can write (pinned host
// memory) that is set to 1 when a row holds an infinity or a NaN.
void layernorm(const float* x, const float* w, const float* b, float eps, int rows, int D, int act,
               float* out_f32, half_t* out_h, hipStream_t, int* nonfinite = nullptr);

// ---- pixel pre-processing (K1) ---------------------------------------------------------------
// u8 image [h,w,C] (row stride `stride` bytes; dlimg::Channels code `channels`) -> f16 patch-major
// matrix [4096, 768]: row = patch (py*64+px), column = c*256 + iy*16 + ix, value
// (float(u8) - mean[c]) / std[c]; pixels outside h x w (zero padding of the graph) are 0.
void preprocess(const uint8_t* img, int w, int h, int stride, int channels, half_t* patches, hipStream_t);
// the same for several images in one launch (the images of a batched pass)
struct PreImage { const uint8_t* img; int w, h, stride, channels; half_t* patches; };
void preprocess_batch(const PreImage* images, int count, hipStream_t);

// ---- K17 longest-side resize (stb_image_resize equivalent; tables from csrc/resize_tables.cpp) ------------
struct ResizeAxis { const int* first; const int* count; const float* coef; int taps; int out; };   // device pointers
// src u8 [h][stride] with C bytes per pixel -> dst u8 [ay.out][ax.out*C] packed; tmp: fp32 [h][ax.out][C] scratch
void resize_srgb(const uint8_t* src, int w, int h, int stride, int C, const ResizeAxis& ax, const ResizeAxis& ay,
                 const float* decode_lut, const uint32_t* encode_tab, float* tmp, uint8_t* dst, hipStream_t);

// ---- elementwise -----------------------------------------------------------------------------
// out_h[i] = f16(a[i] + (b ? b[i % b_mod] : 0)); out_f32 likewise (either may be null); n % 4 == 0
void add_cast(const float* a, const float* b, size_t b_mod, size_t n, float* out_f32, half_t* out_h, hipStream_t);
// 3x3 im2col over a [B,64,64,C] f16 map (zero pad 1): out [B*4096, 9*C], column = (ky*3+kx)*C + c
void im2col3x3(const half_t* in, int B, int C, half_t* out, hipStream_t);
// f32 -> f16 conversion of a weight tensor
void cast_f16(const float* in, half_t* out, size_t n, hipStream_t);

// ---- encoder attention -----------------------------------------------------------------------
// qkv: [B*4096, 3*D] f16 token-major (q | k | v, head-major inside each), out: [B*4096, D] f16.
// qkv_pad: f16 [3*D], the qkv bias (keys / values of a window's zero-padding tokens equal it).
// rel_h/rel_w: f16 [2*S-1, hd] with S = 14 (windowed) or 64 (global); converted once when the weights are loaded.
void attention_window(const half_t* qkv, const half_t* qkv_pad, const half_t* rel_h, const half_t* rel_w,
                      half_t* out, int B, int heads, int hd, hipStream_t);
// attention_global works in units of log2 and leaves the scaling to whoever produces its operands: the q columns of qkv
// must arrive multiplied by attention_global_q_scale(hd) = log2(e) / sqrt(hd) and both rel-pos tables by
// attention_global_rel_scale(hd) = sqrt(hd), so that q'.k is the scaled score and q'.R' the bias q.R, both times
// log2(e).  SamModel folds the factors into the qkv weights / bias and the tables of the global blocks when it loads them.
float attention_global_q_scale(int hd);
float attention_global_rel_scale(int hd);
void attention_global(const half_t* qkv, const half_t* rel_h, const half_t* rel_w, half_t* out, int B,
                      int heads, int hd, hipStream_t);

// ---- mask decoder (token side is tiny: fp32 VALU kernels) ------------------------------------
// floats of workspace for the per-key-group partial results of token_to_image_partials
size_t token_to_image_scratch_floats(int P);
// A [rows][256] fp32 token matrix as a consumer sees it: optionally LayerNorm'ed (over the 256 columns) and with another
// matrix added behind the LayerNorm (the query positional encoding), both applied while the rows are read.
struct TokenRows {
    const float* x = nullptr;
    const float* ln_w = nullptr;
    const float* ln_b = nullptr;
    float eps = 0.f;
    const float* add = nullptr;
};
// Y[r][n] = act(in[r] . W[n] + b[n]) + resid[r][n]; W is [N][K] fp32.  K = 256: `in` with TokenRows semantics; any other
// K: in.x is a plain [rows][K] matrix.  resid.x == nullptr: no residual.
struct TokenLinear {
    TokenRows in;
    int K = 0;
    const float* W = nullptr;
    const float* b = nullptr;
    TokenRows resid;
    float* Y = nullptr;
    int N = 0;
    int relu = 0;
};
// The prompts of one decode (at most 16 per launch), passed to the first kernel by value: coords [P][2][2] in the
// 1024-pixel frame, labels [P][2], and per prompt the DEVICE address of its image embedding ([4096][256] fp32).
constexpr int kDecoderMaxPrompts = 16;
struct DecoderPrompts {
    float coords[kDecoderMaxPrompts * 4];
    float labels[kDecoderMaxPrompts * 2];
    const float* emb[kDecoderMaxPrompts];
};
// First launch of a decode.  tokens [P,7,256]: iou token, 4 mask tokens, 2 prompt tokens (the positional part later steps
// add); `first` (n_first <= 5 layers, K = 256, no LayerNorm / residual; their `in` is ignored) are applied to those same
// rows in this launch.  Image side: keys = emb[p] + no_mask (fp32 + f16) for all prompts.
void decoder_start(const DecoderPrompts& prompts, const float* gauss, const float* point_embed, const float* not_a_point,
                   const float* iou_token, const float* mask_tokens, float* tokens, const TokenLinear* first, int n_first,
                   const float* no_mask, float* keys, half_t* keys_h, int P, hipStream_t s);
// up to 5 layers over the same rows (<= 112) in one launch
void token_linears(const TokenLinear* ops, int count, int rows, hipStream_t);
// self-attention among the 7 tokens of each prompt + its output projection `out` (K = 256) in one launch
void token_self_attention_out(const float* q, const float* k, const float* v, const TokenLinear& out, int P, hipStream_t);
// The same launch carrying a plain GEMM (no activation, no folded LayerNorm, fp32 bias / residual, f16 or fp32 result) that
// neither depends on it nor it on the GEMM: its 64 x 64 tiles are extra workgroups of the launch (decoder.hip).  Returns false
// without launching anything when `g` is not of that kind (the caller then launches both on their own).
bool token_self_attention_out_with_gemm(const float* q, const float* k, const float* v, const TokenLinear& out, int P,
                                        const GemmArgs& g, hipStream_t);
// token-to-image attention.  partials: per key group (max, sum, output) of every (prompt, head, token); the queries
// are q [P,7,128], or are computed in the launch as q_proj (256 -> 128, LayerNorm / positional part on the fly).  The
// fold of the partials + output projection `out` (128 -> 256, out_wt = its weight transposed [128][256]) + residual is
// done by the consumer's launch: token_merge_linear (writes out.Y and applies `next`, a K = 256 layer whose input is
// next.in's LayerNorm of out.Y) or output_heads.
void token_to_image_partials(const float* q, const TokenLinear* q_proj, const half_t* K, int ldk, const half_t* V, int ldv,
                             float* scratch, int P, hipStream_t);
void token_merge_linear(const float* scratch, const TokenLinear& out, const float* out_wt, const TokenLinear& next, int P,
                        hipStream_t);
// The image positions' half of a two-way block in one launch (kernels/decoder_image.hip):
// keys <- LayerNorm(keys + attention(q -> token k / v) Wo + bias), fp32 in place + f16 copy.  q: f16 [P*4096][ldq] (128 wide),
// tk / tv: fp32 [P][7][128], W: f16 [256][128].
void image_update(const half_t* q, int ldq, const float* tk, const float* tv, const half_t* W, const float* bias,
                  const float* ln_w, const float* ln_b, float eps, float* keys, half_t* keys_h, int P, hipStream_t);
// Up-scaling path + mask product in one launch (kernels/decoder_image.hip): logits [P,4,256,256] from the f16 keys, the two
// transposed convolutions as GEMM weights (W1 [256][256], rows = sub-pixel * 64 + channel; W2 [128][64], rows = sub-pixel * 32
// + channel), the LayerNorm2d between them and the hyper vectors [P,4,32].
void upscale_logits(const half_t* keys_h, const half_t* W1, const float* b1, const float* ln_w, const float* ln_b, float eps,
                    const half_t* W2, const float* b2, const float* hyper, float* logits, int P, hipStream_t);
// hyper-network MLPs (4 x 256->256->256->32) and IoU head (256->256->256->4) on the output tokens: the launch finishes the
// final token-to-image attention (scratch, out, out_wt as above) for the five tokens it needs and applies `norm` to them
struct HeadWeights { const float* w[5][3]; const float* b[5][3]; };
void output_heads(const float* scratch, const TokenLinear& out, const float* out_wt, const TokenRows& norm /*ln_w, ln_b, eps*/,
                  const HeadWeights& hw, float* hyper /*[P,4,32]*/, float* iou /*[P,4]*/, int P, hipStream_t);

// ---- mask post-processing (K16) --------------------------------------------------------------
// For each of `count` jobs: logits plane job.src (256x256 f32, device) -> two-stage bilinear
// (256->1024, crop to prepadded ph x pw, -> out_h x out_w), threshold > 0 -> 255/0 into job.dst (device, packed rows).
// If job.select_iou != nullptr the plane is chosen on device as argmax over planes 1..3 of
// select_iou[0..3] (SamOnnxModel.select_masks with 2 prompt points) starting from job.src as plane 0.
// BiRefNet pre/post (kernels/objects.hip): channels 0..2 of an HWC u8 image -> normalised f32 planes; logits -> u8
void birefnet_prepare_image(const uint8_t* pixels, int w, int h, int stride, int bytes_pp, const float mean[3],
                            const float std[3], float* out, hipStream_t s);
void birefnet_process_mask(const float* logits, int w, int h, uint8_t* out, hipStream_t s);

struct PostJob { const float* src; const float* select_iou; uint8_t* dst; int out_w, out_h, pre_w, pre_h; };
void postprocess_masks(const PostJob* jobs_host, int count, hipStream_t);

}  // namespace k
}  // namespace dlimg
