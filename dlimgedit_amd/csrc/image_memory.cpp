#include "image_memory.hpp"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace dlimg {
namespace {

struct Block { size_t capacity; bool pinned; };

struct Registry {
    std::mutex mutex;
    std::map<uintptr_t, Block> live;                              // by first byte
    std::unordered_map<size_t, std::vector<void*>> cached;        // released pinned blocks by capacity
    size_t cached_bytes = 0, cached_blocks = 0;
    size_t live_pinned_bytes = 0;
};

// lives for the whole process (never destroyed: images may be released from static destructors of the host program)
Registry& registry() {
    static Registry* r = new Registry;
    return *r;
}

std::atomic<bool> g_pinned{false};

// Pinning costs ~0.1-0.3 ms per MiB, so released blocks are kept for the next image of that size; a consumer cycles through
// a handful of sizes (its images, their masks).  Beyond this much cached memory a released block goes back to the system.
constexpr size_t kCacheLimitBytes = size_t(512) << 20;
// Pinned pages cannot be swapped or moved: a consumer that keeps thousands of masks alive gets pageable memory (and the
// staged path) for what goes beyond this much live pinned image memory.
constexpr size_t kLivePinnedLimitBytes = size_t(4) << 30;
constexpr size_t kGranule = 64 * 1024;         // pinned capacities are multiples of this: masks of nearby sizes share blocks

size_t pinned_capacity(size_t bytes) { return (bytes + kGranule - 1) / kGranule * kGranule; }

}  // namespace

void image_memory_use_pinned() noexcept {
    static const bool allowed = [] {
        const char* e = std::getenv("DLIMGEDIT_PINNED_IMAGES");
        return !(e && std::atoi(e) == 0);
    }();
    if (allowed) g_pinned.store(true, std::memory_order_release);
}

uint8_t* image_alloc(size_t bytes) noexcept {
    if (bytes == 0) bytes = 1;
    Registry& r = registry();
    try {
        if (g_pinned.load(std::memory_order_acquire)) {
            const size_t cap = pinned_capacity(bytes);
            void* p = nullptr;
            bool room = false;
            {
                std::lock_guard<std::mutex> lock(r.mutex);
                room = r.live_pinned_bytes + cap <= kLivePinnedLimitBytes;
                auto it = r.cached.find(cap);
                if (room && it != r.cached.end() && !it->second.empty()) {
                    p = it->second.back();
                    it->second.pop_back();
                    r.cached_bytes -= cap;
                    --r.cached_blocks;
                }
            }
            // portable: every GPU of a multi-replica environment reads and writes it
            if (room && !p && hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess) {
                (void)hipGetLastError();
                p = nullptr;                         // no pinned memory to be had: a pageable block serves as well
            }
            if (p) {
                std::lock_guard<std::mutex> lock(r.mutex);
                r.live[reinterpret_cast<uintptr_t>(p)] = Block{cap, true};
                r.live_pinned_bytes += cap;
                return static_cast<uint8_t*>(p);
            }
        }
        void* p = std::malloc(bytes);
        if (!p) return nullptr;
        std::lock_guard<std::mutex> lock(r.mutex);
        r.live[reinterpret_cast<uintptr_t>(p)] = Block{bytes, false};
        return static_cast<uint8_t*>(p);
    } catch (...) {
        return nullptr;
    }
}

void image_free(uint8_t const* pixels) noexcept {
    if (!pixels) return;
    Registry& r = registry();
    void* p = const_cast<uint8_t*>(pixels);
    Block b{0, false};
    bool keep = false;
    try {
        std::lock_guard<std::mutex> lock(r.mutex);
        auto it = r.live.find(reinterpret_cast<uintptr_t>(p));
        if (it == r.live.end()) return;              // not ours (or released twice): the reference would corrupt its heap here
        b = it->second;
        r.live.erase(it);
        if (b.pinned) r.live_pinned_bytes -= b.capacity;
        if (b.pinned && r.cached_bytes + b.capacity <= kCacheLimitBytes) {
            r.cached[b.capacity].push_back(p);
            r.cached_bytes += b.capacity;
            ++r.cached_blocks;
            keep = true;
        }
    } catch (...) {
        keep = false;
    }
    if (keep) return;
    if (b.pinned) {
        if (hipHostFree(p) != hipSuccess) (void)hipGetLastError();
    } else {
        std::free(p);
    }
}

bool image_memory_is_pinned(void const* p, size_t bytes) noexcept {
    if (!p || !g_pinned.load(std::memory_order_acquire)) return false;
    Registry& r = registry();
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> lock(r.mutex);
    auto it = r.live.upper_bound(a);
    if (it == r.live.begin()) return false;
    --it;
    return it->second.pinned && a >= it->first && a + bytes <= it->first + it->second.capacity;
}

ImageMemoryStats image_memory_stats() noexcept {
    Registry& r = registry();
    std::lock_guard<std::mutex> lock(r.mutex);
    ImageMemoryStats s{r.live.size(), 0, r.cached_blocks, r.cached_bytes};
    for (auto const& kv : r.live) s.live_pinned_blocks += kv.second.pinned;
    return s;
}

}  // namespace dlimg
