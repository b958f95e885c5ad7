/* dlimgedit_amd.h -- extension entry points of the MI355X build (plain C, exported next to
 * dlimg_init from the same shared library).  None of these exist in the reference; they serve
 *   (1) the throughput benchmark: device-resident inputs/outputs, asynchronous steps, stage clocks;
 *   (2) the parity tests: intermediate results (embedding, mask logits) and single-kernel hooks so
 *       every HIP kernel can be compared with the CPU oracle in isolation.
 * All functions return 0 on success and non-zero on failure with the message available through
 * dlimg_init()->last_error() unless stated otherwise.  f16 tensors cross the boundary as uint16_t
 * bit patterns (IEEE binary16).
 */
#ifndef DLIMGEDIT_AMD_H_
#define DLIMGEDIT_AMD_H_

#include "dlimgedit.h"

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Number of usable HIP devices (0 if the runtime cannot be initialised).  Never fails. */
DLIMG_API int dlimg_amd_device_count(void);

/* ---- model facts ---------------------------------------------------------------------------- */
/* out[0..3] = embed_dim, depth, num_heads, mlp_dim of the SAM encoder loaded by `env` (loads it). */
DLIMG_API int dlimg_amd_model_geometry(dlimg_Environment env, int* out);

/* ---- intermediates for parity tests (Segmentation::process / compute_mask internals) -------- */
/* Copies the cached image embedding, token-major [4096][256] fp32 (reference layout 1x256x64x64 is
 * its transpose), replacing what the reference keeps in SegmentationImpl::image_embedding_
 * (reference: src/segmentation.hpp:61). */
DLIMG_API int dlimg_amd_get_embedding(dlimg_Segmentation seg, float* out);
/* Runs prompt encoder + mask decoder for one prompt and returns the decoder's raw outputs:
 * out_logits [4][256][256] (low_res_masks), out_iou [4] (iou_predictions). */
DLIMG_API int dlimg_amd_get_logits(dlimg_Segmentation seg, int const* point, int const* region, float* out_logits,
                                   float* out_iou);

/* Diagnostic: decodes one point prompt and copies out the decoder's token-side workspaces as that decode left them
 * (tokens, projections, attention outputs, MLP hidden layer, final partials, hyper vectors, IoU, the first 4096 keys
 * values), one after the other; out_layout receives "name:floats,name:floats,...".  out == NULL: layout only. */
DLIMG_API int dlimg_amd_decoder_state(dlimg_Segmentation seg, int const* point, float* out, int capacity, char* out_layout,
                                      int layout_capacity);

/* Host logic of the mask transfer (csrc/mask_pieces.hpp), callable without a GPU (tests): for `count` masks of the given sizes
 * (+ extra_bytes behind them) the end offsets of the pieces the staging area travels in, and per piece the copies the host
 * makes once it has arrived, five numbers each: piece, mask, staging offset, offset inside the mask, bytes.  Returns the
 * number of copies (-1: error). */
DLIMG_API int dlimg_amd_test_mask_pieces(int count, long long const* mask_bytes, long long extra_bytes, long long* out_piece_end,
                                         int piece_capacity, long long* out_copies, int copy_capacity, int* out_pieces);
/* Host logic of the device-step queue behind dlimg_amd_encode_and_mask, callable without a GPU (tests): plans the passes
 * for `pending` waiting requests given the per-lane passes / images in flight and the lane cursor, updates those as if the
 * passes had been launched, writes (lane, images) per pass in launch order and returns the number of passes (-1: error). */
DLIMG_API int dlimg_amd_test_plan_steps(int lanes, int* passes_in_flight, int* images_in_flight, int* cursor, int pending, int width,
                                        int depth, int all, int* out_lane, int* out_images, int capacity);

/* ---- benchmark path: everything device-resident, stream-ordered, no host synchronisation ----- */
/* Device memory helpers (hipMalloc/hipFree/hipMemcpy on the environment's device). */
DLIMG_API int dlimg_amd_device_alloc(dlimg_Environment env, size_t bytes, void** out_ptr);
DLIMG_API int dlimg_amd_device_free(dlimg_Environment env, void* ptr);
DLIMG_API int dlimg_amd_copy_to_device(dlimg_Environment env, void* dst_dev, void const* src_host, size_t bytes);
DLIMG_API int dlimg_amd_copy_to_host(dlimg_Environment env, void* dst_host, void const* src_dev, size_t bytes);
/* Host logic of the lanes' enqueue threads (csrc/environment.hpp, LaneWorker), callable without a GPU (tests): `tasks` tasks
 * sleeping sleep_us each; drain must wait for all of them, they run in posting order, and a task posted after the drain has
 * run when the worker is destroyed.  out_order: tasks + 1 entries.  Returns the number of tasks that ran, -1 on error. */
DLIMG_API int dlimg_amd_test_lane_worker(int tasks, int sleep_us, int* out_order);
/* One pass of the hot path over `count` images already in HBM: pre-process, encode (one batched
 * pass), decode one point prompt per image (single-mask mode) and write the 0/255 masks to
 * dev_masks[i] (width*height bytes each, device memory).  Views carry DEVICE pixel pointers.
 * points: count x {x,y}.  Asynchronous: returns once the request is accepted; call dlimg_amd_synchronize to wait.
 * The passes are put on the lanes' streams by one host thread per lane (DLIMGEDIT_STEP_WORKERS=0: by the calling thread);
 * a pass that cannot be enqueued drops its own requests only, and dlimg_amd_synchronize reports how many and why.
 * Independent single-image requests are coalesced into batched passes of DLIMGEDIT_COALESCE images (default 2, 1 = off;
 * dynamic batching -- the results are bit-identical to single-image passes); a request that is still waiting for a
 * partner is launched by the next request or by dlimg_amd_synchronize (which deals what is left evenly over the lanes). */
DLIMG_API int dlimg_amd_encode_and_mask(dlimg_Environment env, dlimg_ImageView const* dev_images, int count,
                                        int const* points, uint8_t* const* dev_masks);
/* Encode only / decode only variants of the above, for per-stage rates. */
DLIMG_API int dlimg_amd_encode_only(dlimg_Environment env, dlimg_ImageView const* dev_images, int count);
/* Waits for every execution lane of the environment. */
DLIMG_API int dlimg_amd_synchronize(dlimg_Environment env);
/* Number of execution lanes (independent streams + workspaces over shared weights; DLIMGEDIT_LANES, default 3).
 * Successive requests are spread over the lanes round-robin so independent images overlap on the GPU. */
DLIMG_API int dlimg_amd_lane_count(dlimg_Environment env);

/* What the step queue behind dlimg_amd_encode_and_mask really uses (the library clamps DLIMGEDIT_COALESCE / _STEP_DEPTH /
 * _LANES and has its own defaults): out[0] = requests coalesced per pass, out[1] = passes that may wait on a lane,
 * out[2] = execution lanes of replica 0, out[3] = lanes requests are currently spread over (1 while per-kernel clocks run),
 * out[4] = one-image encoder passes enqueued on replica 0 so far, out[5] = how many of them found every other lane idle and
 * ran in the "GPU to itself" tile configuration (six ints). */
DLIMG_API int dlimg_amd_queue_config(dlimg_Environment env, int* out);

/* Multi-GPU: number of replicas of the environment (entries of DLIMGEDIT_DEVICES; 1 by default) and, for a
 * segmentation handle, the replica / HIP device index that holds its embedding (either pointer may be null). */
DLIMG_API int dlimg_amd_replica_count(dlimg_Environment env);
DLIMG_API int dlimg_amd_segmentation_device(dlimg_Segmentation seg, int* out_replica, int* out_device);

/* Device-output form of get_segmentation_masks (table slot 14; reference: there is no multi-device or device-output
 * path at all, src/session.cpp:63-66 uses the default device and src/environment.cpp:142 binds every tensor to host
 * memory): one single-mask query per entry, points / regions as in slot 14.  Mask i is produced on the GPU that holds
 * segs[i]'s embedding and is delivered into the memory of HIP device `root_device` at dev_out + offset_i, where
 * offset_i = sum over the entries before i of width*height (tightly packed, 0 / 255 bytes); the offsets are also written
 * to out_offsets[count] when it is non-null.  Masks of other GPUs cross xGMI as peer-to-peer copies (hipMemcpyPeerAsync);
 * no host memory is involved.  Returns when every mask is in place.  This is the "gather of mask outputs" of the
 * multi-GPU design. */
DLIMG_API int dlimg_amd_get_segmentation_masks_device(dlimg_Segmentation const* segs, int count, int const* points,
                                                      int const* regions, int root_device, uint8_t* dev_out,
                                                      size_t* out_offsets);

/* ---- stage clocks (HIP events on the executor's stream) -------------------------------------- */
#define DLIMG_AMD_STAGE_COUNT 12
/* stage ids: 0 pre, 1 gemm (all MFMA GEMMs of the encoder), 2 layernorm, 3 attention_window,
 * 4 attention_global, 5 encoder_other, 6 decoder (whole prompt+mask decoder), 7 post;
 * 8-11 split the launches of stage 1 by kernel flavour: 8 residual-stream writers with row statistics (patch, proj, fc2),
 * 9 LayerNorm-folded consumer (qkv), 10 LayerNorm-folded consumer + GELU (fc1), 11 other (neck) */
/* enabled = 1: all requests run on lane 0, so every kernel is clocked alone on the chip; enabled = 2: the lanes run as
 * usual and every lane clocks its own launches (the regime the throughput figure is measured in); 0 = off.
 * dlimg_amd_take_stage_stats sums over the lanes. */
DLIMG_API int dlimg_amd_set_profiling(dlimg_Environment env, int enabled);
/* Accumulated since the previous call: milliseconds, algorithmic work (FLOPs for stages 1,3,4,6;
 * bytes for the others) and launch counts; arrays of DLIMG_AMD_STAGE_COUNT. Resets the counters. */
DLIMG_API int dlimg_amd_take_stage_stats(dlimg_Environment env, double* out_ms, double* out_work, long* out_launches);

/* ---- single-kernel hooks (host buffers in and out; device memory handled inside) ------------- */
/* K1: pixels -> patch matrix [4096][768] f16. */
DLIMG_API int dlimg_amd_test_preprocess(uint8_t const* pixels, int width, int height, int stride, int channels,
                                        uint16_t* out_patches);
/* K16: planes [n_planes][256][256] fp32; iou (4 floats) non-null selects the plane as the single-mask
 * decoder does, otherwise plane 0 is used.  out_mask: out_w*out_h bytes. */
DLIMG_API int dlimg_amd_test_postprocess(float const* planes, int n_planes, float const* iou, int out_w, int out_h,
                                         uint8_t* out_mask);
/* K16, several masks in ONE launch (the form slot 14 uses for a chunk of prompts): plane i -> out_masks + i * out_w * out_h. */
DLIMG_API int dlimg_amd_test_postprocess_batch(float const* planes, int n_masks, int out_w, int out_h, uint8_t* out_masks);
/* Forces tile configuration `tile` (index into kernels/gemm.hip's table; < 0: off) in the GEMM test hooks below wherever it
 * fits the problem.  Affects only dlimg_amd_test_gemm / dlimg_amd_test_gemm_ln / the bench hooks, never the product path. */
DLIMG_API int dlimg_amd_test_force_gemm_tile(int tile);
/* A separate tile for the LayerNorm-folded consumer GEMM of dlimg_amd_test_gemm_ln (< 0: the one forced above); reset by
 * every dlimg_amd_test_force_gemm_tile call. */
DLIMG_API int dlimg_amd_test_force_gemm_consumer_tile(int tile);
/* C = epilogue(A[M,K] . W[N,K]^T): bias[N], resid[resid_rows][N] (row m % resid_rows), act 0/1(GELU). */
DLIMG_API int dlimg_amd_test_gemm(int M, int N, int K, uint16_t const* A, uint16_t const* W, float const* bias,
                                  float const* resid, int resid_rows, int act, float* out_f32, uint16_t* out_f16);
/* One stream-writing GEMM of the encoder, x = A[M,K] . W[D,K]^T + bias + (resid_hi + resid_lo), with its per-tile row
 * statistics (out_stats: M * 24 * 2 floats), in either representation of the residual stream: pair == 0 writes fp32 out_x and
 * its f16 copy out_hi, pair == 1 the f16 pair out_hi / out_lo (hi = f16(x), lo = f16(x - hi)); resid_hi / resid_lo may both be
 * NULL.  The two must agree exactly: hi, the statistics, and lo recomputed from out_x (tests/test_gpu_kernels.py). */
DLIMG_API int dlimg_amd_test_gemm_stream(int M, int D, int K, uint16_t const* A, uint16_t const* W, float const* bias,
                                         uint16_t const* resid_hi, uint16_t const* resid_lo, int pair, float* out_x,
                                         uint16_t* out_hi, uint16_t* out_lo, float* out_stats);
/* Two chained GEMMs with the LayerNorm between them folded in ("LayerNorm inside the GEMMs"):
 *   x = A1[M,K1] . W1[D,K1]^T + bias1 + resid[M,D]        -> out_x [M,D] fp32 and out_xh [M,D] f16
 *   y = act(rstd * (xh . Wg[N,D]^T - mean * colsum[N]) + bias2[N])   -> out_y [M,N]; mean / rstd of the rows of x,
 *       merged from the per-tile statistics the first GEMM leaves
 * Wg = W * diag(gamma) in f16, colsum = row sums of Wg, bias2 = b + W.beta: prepared by the caller. */
DLIMG_API int dlimg_amd_test_gemm_ln(int M, int D, int K1, int N, uint16_t const* A1, uint16_t const* W1,
                                     float const* bias1, float const* resid, uint16_t const* Wg, float const* colsum,
                                     float const* bias2, float eps, int act, float* out_x, uint16_t* out_xh,
                                     float* out_y);
DLIMG_API int dlimg_amd_test_layernorm(float const* x, float const* w, float const* b, float eps, int rows, int dim,
                                       int act, float* out_f32, uint16_t* out_f16);
/* Encoder attention on qkv [B*4096][3*heads*hd] f16 -> out [B*4096][heads*hd] f16.
 * global != 0: rel tables are [127][hd]; else windowed 14x14 with [27][hd] tables and qkv_bias [3*D]. */
DLIMG_API int dlimg_amd_test_attention(int global, uint16_t const* qkv, float const* qkv_bias, float const* rel_h,
                                       float const* rel_w, int batch, int heads, int hd, uint16_t* out);
/* K17: the stb_image_resize-equivalent longest-side resampler (default filter, sRGB, clamp):
 * pixels [height][stride] -> out_pixels [out_h][out_w * bytes_per_pixel] packed. */
DLIMG_API int dlimg_amd_test_resize(uint8_t const* pixels, int width, int height, int stride, int channels, int out_w,
                                    int out_h, uint8_t* out_pixels);
/* Pre- and post-processing of dlimg_segment_objects (BiRefNet), the parts the reference library computes itself;
 * the network is an ONNX graph there and is not part of this build.  Host buffers in and out.
 *   prepare_image  replaces BiRefNet::prepare_image   (/root/reference/src/segmentation.cpp:244-256):
 *                  channels 0..2 of an HWC u8 image -> out_nchw [3][height][width] f32 = (x / 255 - mean) / std
 *   process_mask   replaces BiRefNet::process_mask    (/root/reference/src/segmentation.cpp:258-270):
 *                  logits [height][width] f32 -> u8 = uint8_t(sigmoid(x) * 255.f)
 *   resize_mask    replaces dlimg::resize_mask        (/root/reference/src/image.cpp:53-62):
 *                  stb_image_resize, 1 channel, box filter, linear colour space, clamped edges */
DLIMG_API int dlimg_amd_birefnet_prepare_image(uint8_t const* pixels, int width, int height, int stride, int channels,
                                               float const* mean, float const* std, float* out_nchw);
DLIMG_API int dlimg_amd_birefnet_process_mask(float const* logits, int width, int height, uint8_t* out_mask);
DLIMG_API int dlimg_amd_resize_mask(uint8_t const* mask, int width, int height, int stride, int out_w, int out_h,
                                    uint8_t* out_mask);
/* Times the two pixel kernels of the path alone on `batch` (1..16) device-resident 1024x1024 RGBA images / mask requests in
 * ONE launch each: K1 pre-processing (4 MiB u8 in, 6 MiB f16 patch matrix out per image) and K16 post-processing (256 KiB
 * of fp32 logits in, 1 MiB u8 mask out per mask); average ms per launch.  Successive launches rotate over distinct input and
 * output sets with a footprint of `working_set_mb` MB (0: 768, three times the 256 MB Infinity Cache), so that the bytes a
 * launch moves come from and go to HBM.  At one image the kernels sit on the launch floor; at 16 they move 168 MB / 21 MB
 * per launch and can be read against the HBM rate. */
DLIMG_API int dlimg_amd_bench_prepost(int batch, int iters, int working_set_mb, double* out_pre_ms, double* out_post_ms);
/* Times `iters` back-to-back launches of an encoder attention kernel (global != 0: the 4096-token kernel, else the 14x14
 * windowed one) on device-resident random data of `batch` images; returns the average ms per launch. */
DLIMG_API int dlimg_amd_bench_attention(int global, int batch, int heads, int hd, int iters, double* out_ms);
/* Times `iters` launches of the GEMM on device-resident random operands; returns average ms per launch.
 * flavour 0: f16 output; 1: LayerNorm folded in; 2: bias + fp32 residual in place; 3: 2 + f16 copy of the result + row statistics;
 * 4: f16 output with bias. */
DLIMG_API int dlimg_amd_bench_gemm(int M, int N, int K, int act, int flavour, int iters, double* out_ms);
/* The same with the tile configuration forced (tile >= 0, index into kernels/gemm.hip's table; -1: chosen as in the
 * product, `shared` = the shared-GPU hint) and `streams` (1..8) concurrent copies of the problem launched round-robin;
 * out_ms = wall time per GEMM over all streams. */
DLIMG_API int dlimg_amd_bench_gemm_streams(int M, int N, int K, int act, int flavour, int tile, int shared, int streams,
                                           int iters, double* out_ms);

/* The same; additionally returns in-kernel time stamps of the LAST launch on stream 0 for kernels that record them
 * (the ping-pong GEMM): per workgroup 4 x u64 = {shader cycles, 100 MHz ticks} of the main loop and of the whole
 * kernel.  max_groups must be >= the grid size. */
DLIMG_API int dlimg_amd_bench_gemm_stamps(int M, int N, int K, int act, int flavour, int tile, int shared, int streams,
                                          int iters, double* out_ms, unsigned long long* out_stamps, int max_groups);

#ifdef __cplusplus
}
#endif

#endif /* DLIMGEDIT_AMD_H_ */
