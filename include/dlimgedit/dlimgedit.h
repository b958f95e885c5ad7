/* dlimgedit C ABI -- MI355X (gfx950) build.
 *
 * Binary-compatible replacement for the table exported by the reference library
 * (reference: src/include/dlimgedit/detail/dlimgedit.h:27-70, filled in src/dlimgedit.cpp:100-117).
 * The first 13 slots of dlimg_Api have the reference's order, signatures and POD layouts, so a
 * consumer compiled against the reference header (or its header-only C++ wrapper) can load this
 * library unchanged.  Slots after `last_error` are additions of this build (batch entry points,
 * SURVEY.md D5); a consumer that does not know them simply never reads past slot 13.
 *
 * Behaviour that differs from the reference, by design:
 *   - dlimg_gpu means "HIP device (MI355X)"; dlimg_cpu is reported as unsupported: this build has
 *     no CPU execution path (the onnxruntime CPU provider is not reproduced).
 *   - last_error() is per calling thread (the reference's global string is unsynchronised).
 *   - segment_objects returns dlimg_error (the BiRefNet graph is not part of this build); load_image reads PNG and
 *     JPEG files on the host, save_image writes PNG.
 *   - unlike the reference header this file is valid C (the struct tag is typedef'ed).
 */
#ifndef DLIMGEDIT_H_
#define DLIMGEDIT_H_

#include <stdint.h>

#if defined(_MSC_VER)
#    if defined(DLIMGEDIT_EXPORTS)
#        define DLIMG_API __declspec(dllexport)
#    else
#        define DLIMG_API __declspec(dllimport)
#    endif
#elif defined(DLIMGEDIT_EXPORTS)
#    define DLIMG_API __attribute__((visibility("default")))
#else
#    define DLIMG_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque handles.  An environment must outlive every segmentation created from it. */
typedef struct dlimg_Environment_* dlimg_Environment;
typedef struct dlimg_Segmentation_* dlimg_Segmentation;

/* Borrowed view of 8-bit pixels, rows top to bottom.  24 bytes:
 * width@0 height@4 channels@8 stride@12 pixels@16.
 * channels: 1 mask, 3 rgb, 4 rgba, 5 bgra, 6 argb (5 and 6 are 4-byte pixels). stride is in bytes. */
typedef struct dlimg_ImageView {
    int width;
    int height;
    int channels;
    int stride;
    uint8_t* pixels;
} dlimg_ImageView;

typedef enum dlimg_Backend { dlimg_cpu = 0, dlimg_gpu = 1 } dlimg_Backend;

/* 16 bytes: backend@0, model_directory@8.  Weights are looked up as
 * <model_directory>/segmentation/sam_<variant>.dlw */
typedef struct dlimg_Options {
    dlimg_Backend backend;
    char const* model_directory;
} dlimg_Options;

typedef enum dlimg_Result { dlimg_success = 0, dlimg_error = 1 } dlimg_Result;

typedef struct dlimg_Api {
    /* ---- slots 0..12: identical to the reference ------------------------------------------ */

    /* 1 if the backend can be used on this machine; never fails. */
    int (*is_backend_supported)(dlimg_Backend);

    dlimg_Result (*create_environment)(dlimg_Environment* out_env, dlimg_Options const* options);
    void (*destroy_environment)(dlimg_Environment);

    /* Encodes one image.  *out_seg is assigned before encoding starts, so on dlimg_error the
     * caller still owns a handle and must destroy it.  Pixels are only read during the call.
     * The call returns once the pixels have been taken and the encoder pass is enqueued; the first call that needs
     * the embedding (get_segmentation_mask, ...) waits for it, and reports a pass that produced non-finite values
     * (DLIMGEDIT_SYNC_PROCESS=1: this call waits itself, as the reference's does). */
    dlimg_Result (*process_image_for_segmentation)(dlimg_Segmentation* out_seg, dlimg_ImageView const* image,
                                                   dlimg_Environment env);

    /* Exactly one of point {x,y} / region {x0,y0,x1,y1} is non-null (original image pixels).
     * out_masks[1] == NULL: single-mask mode, writes out_masks[0] only, out_accuracy untouched.
     * otherwise: writes three masks (decoder outputs 1..3) and their predicted IoU.
     * Each mask buffer is caller-allocated, width*height bytes, values 0 / 255. */
    dlimg_Result (*get_segmentation_mask)(dlimg_Segmentation seg, int const* point, int const* region,
                                          uint8_t** out_masks, float* out_accuracy);

    /* out_extent[0] = width, out_extent[1] = height of the image given to process. */
    void (*get_segmentation_extent)(dlimg_Segmentation seg, int* out_extent);
    void (*destroy_segmentation)(dlimg_Segmentation);

    dlimg_Result (*segment_objects)(dlimg_ImageView const* image, uint8_t* out_mask, dlimg_Environment env);

    dlimg_Result (*load_image)(char const* filepath, int* out_extent, int* out_channels, uint8_t** out_pixels);
    dlimg_Result (*save_image)(dlimg_ImageView const* image, char const* filepath);

    /* w*h*channels uninitialised bytes; release with destroy_image.  Once a GPU environment exists in the process this
     * memory (and load_image's) is pinned: images that lie in it are uploaded from where they lie, masks whose buffer
     * lies in it are written in place by the GPU.  destroy_image ignores foreign pointers. */
    uint8_t* (*create_image)(int width, int height, int channels);
    void (*destroy_image)(uint8_t const* pixels);

    /* Message of the most recent dlimg_error on the calling thread; valid until its next error. */
    char const* (*last_error)(void);

    /* ---- slots 13..: additions of the MI355X build ---------------------------------------- */

    /* Encodes `count` independent images in one batched pass.  out_segs[i] is assigned for every i
     * before any work starts (same ownership rule as the single-image call). */
    dlimg_Result (*process_images_for_segmentation)(dlimg_Segmentation* out_segs, dlimg_ImageView const* images,
                                                    int count, dlimg_Environment env);

    /* One single-mask query per entry, decoded as one batch.  points: count x {x,y} or NULL;
     * regions: count x {x0,y0,x1,y1} or NULL (exactly one of them non-null).  The same handle may
     * appear several times (several prompts on one cached embedding). out_masks[i]: width*height bytes. */
    dlimg_Result (*get_segmentation_masks)(dlimg_Segmentation const* segs, int count, int const* points,
                                           int const* regions, uint8_t** out_masks);
} dlimg_Api;

/* The only exported symbol of the drop-in ABI.  Returns a process-lifetime table; idempotent. */
DLIMG_API dlimg_Api const* dlimg_init(void);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* DLIMGEDIT_H_ */
