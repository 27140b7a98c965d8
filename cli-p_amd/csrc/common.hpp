// common.hpp — shared host-side helpers for libclipmi.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include "../../include/clipmi.h"

namespace clipmi {

// thread-local last-error string behind clipmi_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
    char* base;
    size_t cap, off;
    Arena(void* p, size_t bytes) : base(static_cast<char*>(p)), cap(bytes), off(0) {}
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* r = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap; }
};

#define CLIPMI_CHECK_LAUNCH(what)                                                        \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess)                                                           \
            return clipmi::set_err(CLIPMI_EHIP, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

// Development knobs. The CLIPMI_* environment variables that A/B runs and tools/ use exist only in the -DCLIPMI_DEV build
// (libclipmi_dev.so: cli-p_amd/build.py --dev); the product library reads no environment variable on any path, and the
// laboratory kernels (live-threshold scan, gemm2w, the scan ablations) are not compiled into it.
#ifdef CLIPMI_DEV
inline long long dev_knob(const char* name, long long dflt) { const char* e = getenv(name); return e && *e ? atoll(e) : dflt; }
inline bool dev_knob_set(const char* name) { return getenv(name) != nullptr; }
#else
constexpr long long dev_knob(const char*, long long dflt) { return dflt; }
constexpr bool dev_knob_set(const char*) { return false; }
#endif

constexpr int NUM_CU = 256;           // MI355X
constexpr int LDS_BYTES = 160 * 1024; // per CU

}  // namespace clipmi
