// Standalone timeline of emph_logmel on the C2 layout (64 x 10 s).
// NOTE: the front-end kernels of the product carry no EMPH_STAMP hooks any more (they distorted
// the kernels they timed, EXPERIMENTS.md, rounds 1-4 section 6): this file times whole launches; the stamp
// machinery below is inert.  In-kernel timelines: tools/micro/stack_bench.hip (STACK_STAMP).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <math.h>
__device__ unsigned long long* g_stamps = nullptr;
#ifdef NO_STAMPS
#define EMPH_STAMP(slot)
#endif
#ifdef STAMP_FIRST   // the first block of every workgroup instead of its last
#define EMPH_STAMP_KEEP(p) (*(p) == 0)
#else
#define EMPH_STAMP_KEEP(p) true
#endif
#ifndef EMPH_STAMP
#define EMPH_STAMP(slot)                                                          \
    do {                                                                          \
        if (g_stamps != nullptr && (threadIdx.x & 63) == 0) {                     \
            unsigned long long* p_ = g_stamps + (static_cast<size_t>(blockIdx.x) * 4 + \
                     (threadIdx.x >> 6)) * 16 + (slot);                           \
            if (EMPH_STAMP_KEEP(p_)) *p_ = __builtin_amdgcn_s_memrealtime();      \
        }                                                                         \
    } while (0)
#endif
#include "../../emphases_amd/csrc/frontend.hip"
#ifdef WITH_PIPELINED
#include "logmel_pipelined.inc"
#endif
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
    const int segments = 64, frames = 1000, samples = 160000;
    const int64_t ld = 16 + static_cast<int64_t>(segments) * 1008 + 128;
    std::vector<float> haudio(static_cast<size_t>(segments) * samples);
    for (size_t i = 0; i < haudio.size(); ++i) haudio[i] = 0.1f * sinf(0.01f * (i % 100000)) + 1e-3f * ((i * 2654435761u) % 1000) / 1000.f;
    std::vector<int64_t> hseg(segments * 8, 0);
    std::vector<int32_t> tiles;
    for (int s = 0; s < segments; ++s) {
        hseg[s * 8 + 0] = static_cast<int64_t>(s) * samples; hseg[s * 8 + 1] = samples;
        hseg[s * 8 + 2] = 0; hseg[s * 8 + 3] = samples;
        hseg[s * 8 + 4] = 16 + s * 1008; hseg[s * 8 + 5] = frames;
        for (int t = 0; t < frames; t += emph_frontend_block()) { tiles.push_back(s); tiles.push_back(t); tiles.push_back(16 + s * 1008); tiles.push_back(frames); }
    }
    // a plausible sparse basis: 80 rows, runs like the real one
    std::vector<int32_t> start(80), count(80), offset(80);
    std::vector<float> values;
    int bin = 1;
    for (int m = 0; m < 80; ++m) {
        count[m] = m < 64 ? 4 + (16 * m) / 63 : 21 + (m - 64);
        start[m] = std::min(bin, 512 - count[m]); offset[m] = values.size();
        for (int j = 0; j < count[m]; ++j) values.push_back(0.01f);
        bin += std::max(1, count[m] / 2);
    }
    std::vector<float> table(emph_frontend_table_size());
    emph_frontend_table_fill(table.data());
    float *audio, *dtable, *dvalues, *out; int64_t* seg; int32_t *dtiles, *dstart, *dcount, *doffset;
    CHECK(hipMalloc(&audio, haudio.size() * 4)); CHECK(hipMemcpy(audio, haudio.data(), haudio.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dtable, table.size() * 4)); CHECK(hipMemcpy(dtable, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dvalues, values.size() * 4)); CHECK(hipMemcpy(dvalues, values.data(), values.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 80 * ld * 4));
    CHECK(hipMalloc(&seg, hseg.size() * 8)); CHECK(hipMemcpy(seg, hseg.data(), hseg.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dtiles, tiles.size() * 4)); CHECK(hipMemcpy(dtiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dstart, 320)); CHECK(hipMemcpy(dstart, start.data(), 320, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dcount, 320)); CHECK(hipMemcpy(dcount, count.data(), 320, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&doffset, 320)); CHECK(hipMemcpy(doffset, offset.data(), 320, hipMemcpyHostToDevice));
    const int n_tiles = tiles.size() / 4;
    int32_t* work; CHECK(hipMalloc(&work, 8)); CHECK(hipMemset(work, 0, 8));
    auto launch = [&]() {
        int status = emph_logmel(audio, 0, seg, dtiles, n_tiles, dtable, dstart, dcount, doffset, dvalues,
                                 (int)values.size(), out, ld, 0, -1, nullptr, nullptr, 0, nullptr);
        if (status) { printf("launch failed %d %s\n", status, emph_last_error()); exit(1); }
    };
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("frontend: %d blocks, %.2f us/launch (%.1f ns/frame)\n", n_tiles, ms * 1e3 / 20, ms * 1e6 / 20 / (segments * frames));
#ifdef WITH_PIPELINED
    {
        // the two-frames-in-flight experiment against the shipped kernel
        std::vector<float> plain(80 * ld), piped(80 * ld);
        CHECK(hipMemcpy(plain.data(), out, plain.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemset(out, 0, 80 * ld * 4));
        const size_t bytes = 4 * emph::kPipeWaveFloats * sizeof(float);
        auto kernel = emph::logmel_pipelined_kernel<false>;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        auto piped_launch = [&]() {
            hipLaunchKernelGGL(kernel, dim3(emph::frontend_grid(n_tiles)), dim3(256), bytes, 0, audio, seg, dtiles, dtable,
                               dstart, dcount, doffset, dvalues, (int)values.size(), out, ld, 0, 0, n_tiles);
        };
        for (int i = 0; i < 3; ++i) piped_launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(piped.data(), out, piped.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) piped_launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        size_t differ = 0;
        for (size_t i = 0; i < plain.size(); ++i) differ += plain[i] != piped[i];
        printf("two frames in flight per wave: %.2f us/launch, %zu of %zu values differ from the shipped kernel\n",
               ms * 1e3 / 20, differ, plain.size());
    }
#endif
    const size_t slots = static_cast<size_t>(n_tiles) * 4 * 16;
    unsigned long long* stamps; CHECK(hipMalloc(&stamps, slots * 8)); CHECK(hipMemset(stamps, 0, slots * 8));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
    launch(); CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> host(slots);
    CHECK(hipMemcpy(host.data(), stamps, slots * 8, hipMemcpyDeviceToHost));
#ifdef FX_CLOCK
    {
        double best = 0, sum = 0; size_t n = 0; double longest = 0;
        for (size_t i = 0; i < slots; i += 16) {
            if (!host[i + 3] || host[i + 3] == host[i + 2]) continue;
            const double mhz = double(host[i + 1] - host[i]) / double(host[i + 3] - host[i + 2]) * 100.;
            sum += mhz; ++n; best = std::max(best, mhz);
            longest = std::max(longest, double(host[i + 3] - host[i + 2]) * 0.01);
        }
        printf("   shader clock during the kernel: mean %.0f MHz, max %.0f MHz over %zu waves; longest wave %.2f us\n", sum / n, best, n, longest);
        return 0;
    }
#endif
    const char* names[16] = {"start", "tile", "top", "windowed", "nextload", "A T1", "B T1", "A T2", "B T2", "A hi", "B hi", "power", "mag wr", "A mel", "B mel", ""};
    for (int slot = 1; slot < 15; ++slot) {
        std::vector<double> values2;
        for (size_t i = 0; i < slots; i += 16) if (host[i] && host[i + slot]) values2.push_back((host[i + slot] - host[i]) * 0.01);
        if (values2.empty()) continue;
        std::sort(values2.begin(), values2.end());
        printf("   %-10s waves=%5zu  min %7.2f  median %7.2f  max %7.2f us after the wave's start\n", names[slot], values2.size(), values2.front(), values2[values2.size() / 2], values2.back());
    }
    return 0;
}
