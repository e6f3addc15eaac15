// Shared helpers of libemphases_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/emphases_hip.h"

namespace emph {

constexpr int kWave = 64;          // CDNA wavefront width
constexpr int kHop = 160;          // emphases/config/defaults.py:53
constexpr int kFft = 1024;         // defaults.py:59
constexpr int kBins = 513;
constexpr int kMels = 80;          // defaults.py:62
constexpr int kPad = 432;          // (1024 - 160) / 2, core.py:357, mels.py:31

void set_error(const char* format, ...);

// Launch + check_launch report errors of THIS launch only: hipGetLastError is a
// per-thread sticky slot that other users of the runtime in the same process
// (torch probing a host pointer, say) may leave set.
//
// Launch timer (emph_launch_timer_*, measurement only): while the calling
// thread has one armed, every launch carries its own pair of events, bound to
// the kernel's dispatch packet itself (hipExtLaunchKernel), so that their
// distance is the kernel's own begin -> end - the timestamps rocprofv3 reads -
// and not the command processor's dispatch around it, which a pair of recorded
// events brackets as well.
struct LaunchTimer {
    hipEvent_t* begin;
    hipEvent_t* end;
    int capacity;
    int count;        // launches seen (those beyond `capacity` are not timed)
};
// (one per thread, one definition for the library and for the micro-benchmarks that
// include a kernel's source: a C++17 inline variable)
inline thread_local LaunchTimer* t_launch_timer = nullptr;

#define EMPH_LAUNCH(kernel, grid, block, lds, stream, ...)                               \
    do {                                                                                 \
        (void)hipGetLastError();                                                         \
        ::emph::LaunchTimer* timer_ = ::emph::t_launch_timer;                            \
        if (timer_ == nullptr || timer_->count >= timer_->capacity) {                    \
            if (timer_ != nullptr) ++timer_->count;                                      \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, ##__VA_ARGS__);         \
        } else {                                                                         \
            const int slot_ = timer_->count++;                                           \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, timer_->begin[slot_], \
                                  timer_->end[slot_], 0, ##__VA_ARGS__);                 \
        }                                                                                \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t status = hipGetLastError();
    if (status != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(status));
        return static_cast<int>(status);
    }
    return EMPH_OK;
}

// Dynamic-LDS limit of a kernel above the 64 KiB default.  The attribute
// belongs to each DEVICE's function object, so the "already raised" cache is
// per (call site = kernel instantiation, device); atomics because two host
// threads may launch at once.  hipFuncSetAttribute is not a stream operation:
// it runs on a kernel's first launch per device only, which keeps steady-state
// launches hipGraph-capturable.
constexpr int kMaxDevices = 32;
struct LdsReservation {
    std::atomic<size_t> bytes[kMaxDevices];
};

inline int reserve_lds(LdsReservation& cache, const void* kernel, size_t lds,
                       const char* what) {
    if (lds <= 64 * 1024) return EMPH_OK;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = -1;
    const bool cached = device >= 0 && device < kMaxDevices;
    if (cached && lds <= cache.bytes[device].load(std::memory_order_acquire))
        return EMPH_OK;
    hipError_t status = hipFuncSetAttribute(
        kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (status != hipSuccess) {
        set_error("%s: cannot reserve %zu bytes of LDS: %s", what, lds,
                  hipGetErrorString(status));
        return static_cast<int>(status);
    }
    if (cached) {
        size_t seen = cache.bytes[device].load(std::memory_order_relaxed);
        while (seen < lds && !cache.bytes[device].compare_exchange_weak(
                                 seen, lds, std::memory_order_release)) {
        }
    }
    return EMPH_OK;
}

#define EMPH_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            ::emph::set_error(__VA_ARGS__);      \
            return (code);                       \
        }                                        \
    } while (0)

// Per-axis view of one row of the segment table.
struct Span {
    int64_t offset;   // first column on the packed axis
    int32_t count;    // valid positions
};

__device__ __forceinline__ Span load_span(const int64_t* seg, int segment,
                                          int axis) {
    const int64_t* row = seg + static_cast<int64_t>(segment) * EMPH_SEG_FIELDS;
    Span span;
    if (axis == EMPH_AXIS_FRAMES) {
        span.offset = row[EMPH_SEG_FRAME_OFF];
        span.count = static_cast<int32_t>(row[EMPH_SEG_FRAMES]);
    } else {
        span.offset = row[EMPH_SEG_WORD_OFF];
        span.count = static_cast<int32_t>(row[EMPH_SEG_WORDS]);
    }
    return span;
}

// One row of a tile table: int32 {segment, first position, first column of the
// segment on the walked axis, positions in the segment}.
struct Tile {
    int segment;
    int first;
    int offset;
    int count;
};

__device__ __forceinline__ Tile load_tile(const int32_t* tiles, int index) {
    const int4 row = reinterpret_cast<const int4*>(tiles)[index];
    return Tile{row.x, row.y, row.z, row.w};
}

// LDS traffic between lanes of ONE wave: DS operations of a wave execute in
// program order, so only the compiler has to be kept from reordering them.
// (A wavefront-scope release/acquire fence would also do that, but hipcc lowers
// it to s_waitcnt vmcnt(0): every pending global store would have to retire
// first — 2 us per use in a store epilogue.)
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace emph
