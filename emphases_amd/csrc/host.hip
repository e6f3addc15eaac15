// Host-side helpers of the batch API (no device code).
//
// emph_host_gather copies many host buffers back to back into one destination
// (the pinned staging buffer of a batch) with a small persistent pool of
// threads.  The reference moves one utterance at a time with `audio.to(device)`
// (emphases/data/preprocess/core.py:74); the batch API stages a whole ragged
// batch for ONE DMA, and at 41 MB per batch the gather, not PCIe, is what the
// host spends its time on: a Python thread pool over numpy copies reaches
// ~58 GB/s and loses to its own dispatch beyond 16 threads.
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace {

struct Job {
    const void* const* sources = nullptr;
    const int64_t* bytes = nullptr;
    const int64_t* offsets = nullptr;
    char* destination = nullptr;
    int count = 0;
    std::atomic<int> next{0};
    std::atomic<int> done{0};
};

class Pool {
  public:
    explicit Pool(int threads) {
        for (int i = 0; i < threads; ++i) workers_.emplace_back([this] { work(); });
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
        finished_.wait(lock, [&] { return job->done.load() >= job->count; });
        job_ = nullptr;
    }

  private:
    static void drain(Job* job) {
        for (;;) {
            const int index = job->next.fetch_add(1);
            if (index >= job->count) return;
            memcpy(job->destination + job->offsets[index], job->sources[index],
                   static_cast<size_t>(job->bytes[index]));
            job->done.fetch_add(1);
        }
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            Job* job;
            {
                std::unique_lock<std::mutex> lock(mutex_);
                wake_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                job = job_;
            }
            if (job == nullptr) continue;
            drain(job);
            std::lock_guard<std::mutex> lock(mutex_);
            finished_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mutex_;
    std::condition_variable wake_, finished_;
    Job* job_ = nullptr;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

std::mutex g_pool_mutex;
Pool* g_pool = nullptr;

}  // namespace

extern "C" {

int emph_host_gather(const void* const* host_sources, const int64_t* host_bytes,
                     const int64_t* host_offsets, int32_t count, void* host_destination,
                     int32_t threads) {
    if (count == 0) return EMPH_OK;
    EMPH_REQUIRE(host_sources && host_bytes && host_offsets && host_destination, EMPH_EINVAL,
                 "emph_host_gather: null pointer");
    EMPH_REQUIRE(threads >= 1 && threads <= 64, EMPH_EINVAL,
                 "emph_host_gather: %d threads (1 .. 64)", threads);
    std::lock_guard<std::mutex> guard(g_pool_mutex);       // one gather at a time
    if (g_pool == nullptr || g_pool->size() != threads - 1) {
        delete g_pool;
        g_pool = new Pool(threads - 1);
    }
    Job job;
    job.sources = host_sources;
    job.bytes = host_bytes;
    job.offsets = host_offsets;
    job.destination = static_cast<char*>(host_destination);
    job.count = count;
    g_pool->run(&job);
    return EMPH_OK;
}

}  // extern "C"
