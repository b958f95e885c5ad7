// Pixel memory the library hands out through its table: load_image (slot 8) and create_image (slot 10) allocate,
// destroy_image (slot 11) releases -- reference: /root/reference/src/dlimgedit.cpp:95-118 (new[] / delete[] there).
// The reference's own C++ wrapper allocates EVERY Image through these slots (dlimgedit.impl.hpp:139,144,165): the pixels a
// consumer loaded from a file and the Image that Segmentation::compute_mask(point) returns both live in memory this library
// chose.  Once a GPU environment exists in the process that memory is pinned (hipHostMalloc, kept in a free list by size):
// process() then sends such an image to the GPU from where it lies, and the post-processing kernel writes such a mask where
// the consumer will read it -- no staging copy on either side (csrc/sam_model.cpp: upload_image, enqueue_masks).
// Memory that came from anywhere else (a caller's own buffer) takes the staged path as before.
#pragma once

#include <cstddef>
#include <cstdint>

namespace dlimg {

uint8_t* image_alloc(size_t bytes) noexcept;               // nullptr when there is no memory
void image_free(uint8_t const* pixels) noexcept;           // memory from image_alloc (nullptr: nothing)
// [p, p + bytes) lies inside one live pinned block from image_alloc: a GPU may read / write it in place
bool image_memory_is_pinned(void const* p, size_t bytes) noexcept;
// From now on blocks are pinned (EnvironmentImpl, GPU backend).  DLIMGEDIT_PINNED_IMAGES=0 keeps them pageable.
void image_memory_use_pinned() noexcept;
struct ImageMemoryStats { size_t live_blocks, live_pinned_blocks, cached_blocks, cached_bytes; };
ImageMemoryStats image_memory_stats() noexcept;

}  // namespace dlimg
