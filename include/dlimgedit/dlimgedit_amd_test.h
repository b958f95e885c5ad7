/* dlimgedit_amd_test.h -- test and benchmark hooks of the MI355X build.  They are exported by lib/libdlimgedit_test.so
 * (the product's objects plus csrc/test_hooks.cpp) and by the tuning library, NOT by the product libdlimgedit.so, whose
 * dynamic symbols are dlimg_init and the entry points of dlimgedit_amd.h only (reference:
 * /root/reference/src/CMakeLists.txt:11).  Single kernels behind plain host buffers so that every HIP kernel can be
 * compared with the CPU oracle in isolation, host-logic checks that need no GPU, and kernels-alone timing loops.
 * All functions return 0 on success and non-zero on failure (message: dlimg_init()->last_error() of the SAME library)
 * unless stated otherwise.  f16 tensors cross the boundary as uint16_t bit patterns (IEEE binary16).
 */
#ifndef DLIMGEDIT_AMD_TEST_H_
#define DLIMGEDIT_AMD_TEST_H_

#include "dlimgedit_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- host logic, callable without a GPU ------------------------------------------------------ */
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
/* Host logic of the lanes' enqueue threads (csrc/environment.hpp, LaneWorker), callable without a GPU (tests): `tasks` tasks
 * sleeping sleep_us each; drain must wait for all of them, they run in posting order, and a task posted after the drain has
 * run when the worker is destroyed.  out_order: tasks + 1 entries.  Returns the number of tasks that ran, -1 on error. */
DLIMG_API int dlimg_amd_test_lane_worker(int tasks, int sleep_us, int* out_order);

/* Host logic of the multi-GPU helper threads' CPU binding (csrc/environment.cpp, bind_thread_near_device): the sysfs cpulist
 * parser, "0-3,8,10-11" -> indices; returns their number, -1 on error. */
DLIMG_API int dlimg_amd_test_parse_cpu_list(char const* text, int* out_cpus, int capacity);

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
/* ---- single-kernel hooks (host buffers in and out; device memory handled inside) ------------- */
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

/* ---- kernels alone, timed on device-resident data --------------------------------------------- */
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

#endif /* DLIMGEDIT_AMD_TEST_H_ */
