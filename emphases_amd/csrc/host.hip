// Host-side helpers of the batch API (no device code).
//
// emph_host_gather copies many host buffers back to back into one destination
// (the pinned staging buffer of a batch) with a small persistent pool of
// threads.  The reference moves one utterance at a time with `audio.to(device)`
// (emphases/data/preprocess/core.py:74); the batch API stages a whole ragged
// batch for ONE DMA, and at 41 MB per batch the gather, not PCIe, is what the
// host spends its time on: a Python thread pool over numpy copies reaches
// ~58 GB/s and loses to its own dispatch beyond 16 threads.
#include <emmintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>

#include "common.h"

namespace {

// A source is copied in pieces of at most kPiece bytes, so that a batch of a
// few long utterances (BASELINE configs[4]: eight of 19 MB) spreads over the
// threads like one of many short ones.
constexpr int64_t kPiece = 1 << 20;

struct Piece {
    const char* source;
    char* destination;
    size_t bytes;
};

// The staging buffer is written once and read by the DMA engine only: streaming
// (non-temporal) stores skip the read-for-ownership of every destination line,
// a third of the memory traffic of a plain memcpy.
void copy_streaming(char* destination, const char* source, size_t bytes) {
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(destination) & 15)) & 15;
    if (bytes < 4096 + head) {
        memcpy(destination, source, bytes);
        return;
    }
    memcpy(destination, source, head);
    destination += head, source += head, bytes -= head;
    const size_t blocks = bytes / 64;
    for (size_t i = 0; i < blocks; ++i) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(source) + 0);
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(source) + 1);
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(source) + 2);
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(source) + 3);
        _mm_stream_si128(reinterpret_cast<__m128i*>(destination) + 0, a);
        _mm_stream_si128(reinterpret_cast<__m128i*>(destination) + 1, b);
        _mm_stream_si128(reinterpret_cast<__m128i*>(destination) + 2, c);
        _mm_stream_si128(reinterpret_cast<__m128i*>(destination) + 3, d);
        source += 64, destination += 64;
    }
    memcpy(destination, source, bytes - blocks * 64);
    _mm_sfence();
}

struct Job {
    std::vector<Piece> pieces;
    int count = 0;
    int workers = 0;                  // pool threads that take part (the rest sleep on)
    int active = 0;                   // ... and are inside drain() (guarded by the pool mutex)
    bool streaming = true;
    std::atomic<int> next{0};
    std::atomic<int> done{0};
};

class Pool {
  public:
    explicit Pool(int threads) { grow(threads); }
    // (the pool only grows: a call that asks for fewer threads leaves the others asleep)
    // (Pinning the workers to cores 8 apart - one per core complex of the host -
    // made the gather another 20 % faster at the median and the 99th percentile of
    // a call six times slower, 1.7 -> 11.7 ms: a pinned worker waits for its core.)
    void grow(int threads) {
        while (size() < threads) {
            const int index = size();
            workers_.emplace_back([this, index] { work(index); });
        }
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto& worker : workers_) worker.join();
    }
    int size() const { return static_cast<int>(workers_.size()); }
    void run(Job* job) {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            job_ = job;
            ++generation_;
        }
        wake_.notify_all();
        drain(job);                               // the caller copies too
        std::unique_lock<std::mutex> lock(mutex_);
        // (the job lives on the caller's stack: nobody may still hold it)
        finished_.wait(lock, [&] { return job->done.load() >= job->count && job->active == 0; });
        job_ = nullptr;
    }

  private:
    static void drain(Job* job) {
        for (;;) {
            const int index = job->next.fetch_add(1);
            if (index >= job->count) return;
            const Piece& piece = job->pieces[index];
            if (job->streaming) copy_streaming(piece.destination, piece.source, piece.bytes);
            else memcpy(piece.destination, piece.source, piece.bytes);
            job->done.fetch_add(1);
        }
    }
    void work(int index) {
        uint64_t seen = generation_at_start();
        for (;;) {
            Job* job;
            {
                std::unique_lock<std::mutex> lock(mutex_);
                wake_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                job = job_;
                if (job == nullptr || index >= job->workers) continue;
                ++job->active;
            }
            drain(job);
            std::lock_guard<std::mutex> lock(mutex_);
            --job->active;
            finished_.notify_all();
        }
    }
    // a thread added by grow() must not take the generation that was current
    // before it existed for a new job
    uint64_t generation_at_start() {
        std::lock_guard<std::mutex> lock(mutex_);
        return generation_;
    }
    std::vector<std::thread> workers_;
    std::mutex mutex_;
    std::condition_variable wake_, finished_;
    Job* job_ = nullptr;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

std::mutex* g_pool_mutex = new std::mutex;
Pool* g_pool = nullptr;

// fork(): the child has this thread only - the pool's workers do not exist
// there and its mutexes may have been held by one of them.  The child forgets
// the parent's pool (and lock) without touching them and builds its own on its
// first gather.
void forget_pool_in_child() {
    g_pool = nullptr;
    g_pool_mutex = new std::mutex;
}
const int g_atfork = pthread_atfork(nullptr, nullptr, forget_pool_in_child);

}  // namespace

namespace {
void release_timer(emph::LaunchTimer* timer) {
    for (int i = 0; i < timer->capacity; ++i) {
        if (timer->begin[i]) (void)hipEventDestroy(timer->begin[i]);
        if (timer->end[i]) (void)hipEventDestroy(timer->end[i]);
    }
    delete[] timer->begin;
    delete[] timer->end;
    delete timer;
}
}  // namespace

extern "C" {

int emph_launch_timer_begin(int32_t capacity) {
    EMPH_REQUIRE(capacity >= 1 && capacity <= (1 << 20), EMPH_EINVAL,
                 "emph_launch_timer_begin: capacity %d (1 .. 2^20)", capacity);
    EMPH_REQUIRE(emph::t_launch_timer == nullptr, EMPH_EINVAL,
                 "emph_launch_timer_begin: this thread's timer is already armed");
    auto* timer = new emph::LaunchTimer{new hipEvent_t[capacity](), new hipEvent_t[capacity](),
                                        capacity, 0};
    for (int i = 0; i < capacity; ++i) {
        hipError_t status = hipEventCreate(&timer->begin[i]);
        if (status == hipSuccess) status = hipEventCreate(&timer->end[i]);
        if (status != hipSuccess) {
            release_timer(timer);
            emph::set_error("emph_launch_timer_begin: %s", hipGetErrorString(status));
            return static_cast<int>(status);
        }
    }
    emph::t_launch_timer = timer;
    return EMPH_OK;
}

int32_t emph_launch_timer_count(void) {
    return emph::t_launch_timer ? emph::t_launch_timer->count : -1;
}

int emph_launch_timer_end(float* host_microseconds, int32_t capacity, int32_t* host_count) {
    emph::LaunchTimer* timer = emph::t_launch_timer;
    EMPH_REQUIRE(timer != nullptr, EMPH_EINVAL,
                 "emph_launch_timer_end: no timer armed on this thread");
    emph::t_launch_timer = nullptr;
    const int timed = timer->count < timer->capacity ? timer->count : timer->capacity;
    if (host_count) *host_count = timer->count;
    int result = EMPH_OK;
    for (int i = 0; i < timed && i < capacity && host_microseconds; ++i) {
        float ms = 0.f;
        hipError_t status = hipEventSynchronize(timer->end[i]);
        if (status == hipSuccess) status = hipEventElapsedTime(&ms, timer->begin[i], timer->end[i]);
        if (status != hipSuccess) {
            emph::set_error("emph_launch_timer_end: launch %d: %s", i, hipGetErrorString(status));
            result = static_cast<int>(status);
            break;
        }
        host_microseconds[i] = ms * 1e3f;
    }
    release_timer(timer);
    return result;
}

int emph_host_gather(const void* const* host_sources, const int64_t* host_bytes,
                     const int64_t* host_offsets, int32_t count, void* host_destination,
                     int32_t threads) {
    if (count == 0) return EMPH_OK;
    EMPH_REQUIRE(host_sources && host_bytes && host_offsets && host_destination, EMPH_EINVAL,
                 "emph_host_gather: null pointer");
    EMPH_REQUIRE(threads >= 1 && threads <= 64, EMPH_EINVAL,
                 "emph_host_gather: %d threads (1 .. 64)", threads);
    std::lock_guard<std::mutex> guard(*g_pool_mutex);      // one gather at a time
    if (g_pool == nullptr) g_pool = new Pool(threads - 1);
    g_pool->grow(threads - 1);
    Job job;
    for (int32_t i = 0; i < count; ++i) {
        EMPH_REQUIRE(host_bytes[i] >= 0 && host_offsets[i] >= 0, EMPH_EINVAL,
                     "emph_host_gather: source %d has a negative size or offset", i);
        const char* source = static_cast<const char*>(host_sources[i]);
        char* destination = static_cast<char*>(host_destination) + host_offsets[i];
        for (int64_t at = 0; at < host_bytes[i]; at += kPiece) {
            const int64_t bytes = host_bytes[i] - at < kPiece ? host_bytes[i] - at : kPiece;
            job.pieces.push_back({source + at, destination + at, static_cast<size_t>(bytes)});
        }
    }
    job.count = static_cast<int>(job.pieces.size());
    job.workers = threads - 1;
    job.streaming = true;
    if (job.count == 0) return EMPH_OK;
    g_pool->run(&job);
    return EMPH_OK;
}

}  // extern "C"
