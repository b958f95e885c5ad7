// roctx ranges around the stages of the hot path (pre / encode / decode / post), visible in rocprofv3 --marker-trace.
// The marker library is looked up at run time (librocprofiler-sdk-roctx, else libroctx64): it is used when the process
// already has it (the host or the profiler brought it in) or when DLIMGEDIT_ROCTX=1 asks for it to be loaded; a consumer
// of libdlimgedit.so does not have to link it, and without it the ranges cost one predictable branch.
#pragma once

#include <dlfcn.h>

#include <cstdlib>

namespace dlimg {
namespace roctx {

struct Api {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Api() {
        const char* want = std::getenv("DLIMGEDIT_ROCTX");
        const int flags = RTLD_NOW | RTLD_GLOBAL | ((want && std::atoi(want) != 0) ? 0 : RTLD_NOLOAD);
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(name, flags);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};

inline Api const& api() {
    static const Api a;
    return a;
}

class Range {
  public:
    explicit Range(const char* name) : active_(api().push != nullptr) {
        if (active_) api().push(name);
    }
    ~Range() {
        if (active_) api().pop();
    }
    Range(Range const&) = delete;
    Range& operator=(Range const&) = delete;

  private:
    bool active_;
};

}  // namespace roctx
}  // namespace dlimg
