/* dlimgedit_amd.h -- extension entry points of the MI355X build (plain C, exported next to
 * dlimg_init from the same shared library).  None of these exist in the reference; they serve
 *   (1) the throughput path: device-resident inputs/outputs, asynchronous steps, the device gather, stage clocks;
 *   (2) parity checks on the product build: intermediate results (embedding, mask logits, decoder workspaces);
 *   (3) the library-side kernels of segment_objects.
 * The single-kernel test hooks and the kernels-alone benchmark hooks (dlimg_amd_test_*, dlimg_amd_bench_*) are NOT in
 * the product library: dlimgedit_amd_test.h declares them, lib/libdlimgedit_test.so and the tuning library export them.
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

/* ---- benchmark path: everything device-resident, stream-ordered, no host synchronisation ----- */
/* Device memory helpers (hipMalloc/hipFree/hipMemcpy on the environment's device). */
DLIMG_API int dlimg_amd_device_alloc(dlimg_Environment env, size_t bytes, void** out_ptr);
DLIMG_API int dlimg_amd_device_free(dlimg_Environment env, void* ptr);
DLIMG_API int dlimg_amd_copy_to_device(dlimg_Environment env, void* dst_dev, void const* src_host, size_t bytes);
DLIMG_API int dlimg_amd_copy_to_host(dlimg_Environment env, void* dst_host, void const* src_dev, size_t bytes);
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

/* Image memory of the table (load_image / create_image, slots 8 / 10; reference: new[] in src/dlimgedit.cpp:95-118): once a
 * GPU environment exists in the process such memory is pinned, process_image_for_segmentation reads an image that lies in it
 * from where it lies and get_segmentation_mask writes a mask whose buffer lies in it in place (csrc/image_memory.hpp).
 * *out_pinned = 1 when [pixels, pixels + bytes) is inside one live block of that kind, i.e. takes those paths; 0 otherwise
 * (the program's own buffers: staged through the library's pinned rings as before). */
DLIMG_API int dlimg_amd_image_memory(void const* pixels, size_t bytes, int* out_pinned);

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
#define DLIMG_AMD_STAGE_COUNT 15
/* stage ids: 0 pre, 1 gemm (all MFMA GEMMs of the encoder), 2 layernorm, 3 attention_window,
 * 4 attention_global, 5 encoder_other, 6 decoder (whole prompt+mask decoder), 7 post;
 * 8-11 split the launches of stage 1 by kernel flavour: 8 residual-stream writers with row statistics (patch, proj, fc2),
 * 9 LayerNorm-folded consumer (qkv), 10 LayerNorm-folded consumer + GELU (fc1), 11 other (neck);
 * 12-14 split stage 8 once more by shape: 12 patch embedding, 13 proj (K = D), 14 fc2 (K = 4 D) */
/* enabled = 1: all requests run on lane 0, so every kernel is clocked alone on the chip; enabled = 2: the lanes run as
 * usual and every lane clocks its own launches (the regime the throughput figure is measured in); 0 = off.
 * dlimg_amd_take_stage_stats sums over the lanes. */
DLIMG_API int dlimg_amd_set_profiling(dlimg_Environment env, int enabled);
/* Accumulated since the previous call: milliseconds, algorithmic work (FLOPs for stages 1,3,4,6;
 * bytes for the others) and launch counts; arrays of DLIMG_AMD_STAGE_COUNT. Resets the counters. */
DLIMG_API int dlimg_amd_take_stage_stats(dlimg_Environment env, double* out_ms, double* out_work, long* out_launches);

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

#ifdef __cplusplus
}
#endif

#endif /* DLIMGEDIT_AMD_H_ */
