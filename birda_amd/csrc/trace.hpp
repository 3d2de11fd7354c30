// Optional ROCTx ranges around the stages of a forward pass (tracing): off unless BIRDA_HIP_ROCTX=1.
// The reference logs stage boundaries through `tracing` (`info!` / `debug!` lines, src/lib.rs:1102-1127,
// src/pipeline/processor.rs:445,537,585,683,785); here `rocprofv3 --marker-trace --kernel-trace` shows the same boundaries on
// the GPU timeline.  The ROCTx library is opened at run time (the product has no link-time dependency on a profiler).
#pragma once
#include <dlfcn.h>

#include <cstdlib>

namespace bh {

struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
};

inline const Roctx &roctx() {
    static const Roctx r = [] {
        Roctx x;
        const char *e = getenv("BIRDA_HIP_ROCTX");
        if (!e || e[0] != '1') return x;
        void *h = nullptr;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return x;
        x.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        x.on = x.push && x.pop;
        return x;
    }();
    return r;
}

// one nested range on the calling thread, closed when the object leaves scope
struct TraceRange {
    bool on;
    explicit TraceRange(const char *name) : on(roctx().on) {
        if (on) roctx().push(name);
    }
    ~TraceRange() {
        if (on) roctx().pop();
    }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
};

}  // namespace bh
