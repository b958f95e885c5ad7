// Host-side utilities shared by the runtime: error type, assertion macro, HIP error mapping and
// RAII device memory.  Mirrors the error convention of the reference (exceptions inside, mapped to
// dlimg_error + last_error() at the C boundary: /root/reference/src/dlimgedit.cpp:26-40,
// /root/reference/src/assert.hpp:12-28).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <exception>
#include <mutex>
#include <optional>
#include <string>
#include <vector>
#include <tuple>
#include <utility>

namespace dlimg {

class Exception : public std::exception {
  public:
    explicit Exception(std::string msg) : msg_(std::move(msg)) {}
    char const* what() const noexcept override { return msg_.c_str(); }

  private:
    std::string msg_;
};

[[noreturn]] void throw_error(const char* msg);
[[noreturn]] void assertion_failed(const char* file, int line, const char* expr);
[[noreturn]] void hip_failed(const char* file, int line, const char* expr, hipError_t err);

// Did the HIP runtime read a GPU_MAX_HW_QUEUES of at least 8 (environment.cpp: set by the host, or by this library's load-time
// constructor while the runtime was demonstrably not yet initialised)?  Decides the lanes' stream layout (sam_model.cpp).
bool hardware_queues_trusted();

// Multi-GPU host hygiene (r06): binds the CALLING thread to the CPUs of the NUMA node HIP device `device` hangs off (its PCI
// function's numa_node in sysfs), intersected with the CPUs the process may use; returns the number of CPUs it was bound
// to, 0 when nothing was changed (no NUMA information, a single node, DLIMGEDIT_NUMA_AFFINITY=0, or any failure: never an
// error).  Used by the helper threads that feed the replicas of a multi-GPU environment, never on a one-GPU environment.
int bind_thread_near_device(int device) noexcept;
// "0-3,8,10-11" -> {0,1,2,3,8,10,11} (sysfs cpulist format); malformed pieces are skipped
std::vector<int> parse_cpu_list(std::string const& text);

#define DLIMG_ASSERT(cond)                                           \
    do {                                                             \
        if (!(cond)) ::dlimg::assertion_failed(__FILE__, __LINE__, #cond); \
    } while (0)

#define HIP_CHECK(expr)                                                      \
    do {                                                                     \
        hipError_t err__ = (expr);                                           \
        if (err__ != hipSuccess) ::dlimg::hip_failed(__FILE__, __LINE__, #expr, err__); \
    } while (0)

// Owning device allocation; grows on demand, never shrinks.
template <typename T> class DeviceBuffer {
  public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) { reserve(n); }
    DeviceBuffer(DeviceBuffer const&) = delete;
    DeviceBuffer& operator=(DeviceBuffer const&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept : ptr_(o.ptr_), cap_(o.cap_) { o.ptr_ = nullptr; o.cap_ = 0; }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
        std::swap(ptr_, o.ptr_);
        std::swap(cap_, o.cap_);
        return *this;
    }
    ~DeviceBuffer() { release(); }

    void reserve(size_t n) {
        if (n <= cap_) return;
        release();
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ptr_), n * sizeof(T)));
        cap_ = n;
    }
    void release() noexcept {
        if (ptr_) (void)hipFree(ptr_);
        ptr_ = nullptr;
        cap_ = 0;
    }
    T* get() const noexcept { return ptr_; }
    size_t capacity() const noexcept { return cap_; }

  private:
    T* ptr_ = nullptr;
    size_t cap_ = 0;
};

// Pinned host staging memory.
class PinnedBuffer {
  public:
    PinnedBuffer() = default;
    PinnedBuffer(PinnedBuffer const&) = delete;
    PinnedBuffer& operator=(PinnedBuffer const&) = delete;
    ~PinnedBuffer() { if (ptr_) (void)hipHostFree(ptr_); }
    void reserve(size_t bytes) {
        if (bytes <= cap_) return;
        if (ptr_) (void)hipHostFree(ptr_);
        ptr_ = nullptr;
        cap_ = 0;
        HIP_CHECK(hipHostMalloc(&ptr_, bytes, hipHostMallocDefault));
        cap_ = bytes;
    }
    void* get() const noexcept { return ptr_; }

  private:
    void* ptr_ = nullptr;
    size_t cap_ = 0;
};

// Thread-safe create-on-first-use slot (same contract as the reference's Lazy<T>,
// /root/reference/src/lazy.hpp:8-18: concurrent first callers block until construction is done).
template <typename T> class Lazy {
  public:
    template <typename... Args> T& get_or_create(Args&&... args) {
        std::call_once(flag_, [&] { obj_.emplace(std::forward<Args>(args)...); });
        return *obj_;
    }
    // `make_args()` returns a tuple of constructor arguments and only runs for the creating caller.
    template <typename F> T& get_or_make(F&& make_args) {
        std::call_once(flag_, [&] {
            std::apply([&](auto&&... a) { obj_.emplace(std::forward<decltype(a)>(a)...); }, make_args());
        });
        return *obj_;
    }

  private:
    std::once_flag flag_;
    std::optional<T> obj_;
};

}  // namespace dlimg
