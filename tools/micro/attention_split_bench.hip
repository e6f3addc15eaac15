// attention_split_kernel (emphases_amd/csrc/attention_split.hip) with an in-kernel timeline:
// where do a wave's cycles go?  s_memtime at the phase boundaries of every block of 32 keys
// and of every stage, summed per wave, averaged over the waves that work.
//   phases of a block: 0->1 fragments asked for + S^T issued | 1->2 S^T complete, maximum |
//                      2->3 exp2 + split of the probabilities | 3->4 O^T issued
//   phases of a stage: 5->6 wait for the next stage's DMA | 6->7 the workgroup's barrier
// BASELINE configs[2]: 64 segments of 1000 positions, 2 heads of 40.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Iemphases_amd/csrc \
//        tools/micro/attention_split_bench.hip -o tools/micro/bin/attention_split_bench [-DBENCH_PIECES=3]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define SPLIT_STAMP(slot)                                                        \
    do {                                                                         \
        const unsigned long long now = __builtin_amdgcn_s_memtime();             \
        if (stamp_last_slot >= 0) stamp_sum[(slot)] += now - stamp_last;         \
        stamp_last = now;                                                        \
        stamp_last_slot = (slot);                                                \
    } while (0)
#define SPLIT_STAMP_ARGUMENT , unsigned long long* __restrict__ stamp_out
#define SPLIT_STAMP_PASS , nullptr
#define SPLIT_STAMP_FINISH                                                       \
    if (stamp_out != nullptr && (threadIdx.x & 63) == 0)                                             \
        for (int i = 0; i < 10; ++i)                                             \
            stamp_out[(static_cast<size_t>(blockIdx.x + gridDim.x * blockIdx.y) * 8 + (threadIdx.x >> 6)) * 10 + i] = \
                !working ? 0ull : i < 8 ? stamp_sum[i] : i == 8 ? __builtin_amdgcn_s_memtime() - stamp_begin  \
                                                                : __builtin_amdgcn_s_memrealtime() - stamp_real;
// (the variables the stamps use live at kernel scope)
#define SPLIT_STAMP_DECLARE unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = 0; int stamp_last_slot = -1; \
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memtime(), stamp_real = __builtin_amdgcn_s_memrealtime();
#include "attention_split.hip"
#include <stdarg.h>
// (what csrc/frontend.hip gives the library)
namespace emph {
static char bench_error[512];
void set_error(const char* format, ...) {
    va_list args;
    va_start(args, format);
    vsnprintf(bench_error, sizeof(bench_error), format, args);
    va_end(args);
}
}  // namespace emph

#ifndef BENCH_PIECES
#define BENCH_PIECES 2
#endif
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
    const int segments = 64, frames = 1000, channels = 80, heads = 2;
    // packed axis as emphases_amd.batch lays it out: 16 lead columns, segments 16-aligned
    std::vector<int> offsets(segments);
    int ld = 16;
    for (int i = 0; i < segments; ++i) { offsets[i] = ld; ld += (frames + 15) / 16 * 16; }
    ld += 128;
    std::vector<int32_t> tiles64, tiles256;
    for (int i = 0; i < segments; ++i) {
        for (int first = 0; first < frames; first += 64) { tiles64.insert(tiles64.end(), {i, first, offsets[i], frames}); }
        for (int first = 0; first < frames; first += 256) { tiles256.insert(tiles256.end(), {i, first, offsets[i], frames}); }
    }
    std::vector<float> qk(size_t(2) * channels * ld), v(size_t(ld) * channels);
    unsigned state = 12345;
    auto uniform = [&] { state = state * 1664525u + 1013904223u; return float(state >> 8) / float(1 << 24) - 0.5f; };
    for (auto& x : qk) x = 3.f * uniform();
    for (auto& x : v) x = 2.f * uniform();
    float *d_qk, *d_v, *d_out;
    int32_t *d_t64, *d_t256;
    unsigned char* images;
    unsigned long long* stamps;
    const int64_t image_bytes = emph_split_kv_bytes(ld, segments, channels, heads, BENCH_PIECES);
    const int n256 = int(tiles256.size() / 4);
    CHECK(hipMalloc(&d_qk, qk.size() * 4)); CHECK(hipMalloc(&d_v, v.size() * 4));
    CHECK(hipMalloc(&d_out, size_t(channels) * ld * 4));
    CHECK(hipMalloc(&d_t64, tiles64.size() * 4)); CHECK(hipMalloc(&d_t256, tiles256.size() * 4));
    CHECK(hipMalloc(&images, image_bytes));
    const size_t stamp_count = size_t(n256) * heads * 8 * 10;
    CHECK(hipMalloc(&stamps, stamp_count * 8));
    CHECK(hipMemcpy(d_qk, qk.data(), qk.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_v, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_t64, tiles64.data(), tiles64.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_t256, tiles256.data(), tiles256.size() * 4, hipMemcpyHostToDevice));
    if (emph_split_kv(d_qk, d_v, ld, channels, heads, d_t64, int(tiles64.size() / 4), 64, BENCH_PIECES, images, nullptr)) {
        printf("%s\n", emph::bench_error);
        return 1;
    }
    auto kernel = emph::attention_split_kernel<40, (BENCH_PIECES == 32 ? 3 : BENCH_PIECES), (BENCH_PIECES == 32 ? 2 : BENCH_PIECES)>;
    const size_t lds = emph::kSplitRing * emph::SplitImages<40, (BENCH_PIECES == 32 ? 3 : BENCH_PIECES), (BENCH_PIECES == 32 ? 2 : BENCH_PIECES)>::kStageBytes;
    CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&] {
        hipLaunchKernelGGL(kernel, dim3(n256, heads), dim3(512), lds, 0, d_qk, images, d_out, (int64_t)ld, channels,
                           d_t256, (const int32_t*)nullptr, stamps);
    };
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> host(stamp_count);
    CHECK(hipMemcpy(host.data(), stamps, stamp_count * 8, hipMemcpyDeviceToHost));
    double sum[10] = {0};
    size_t waves = 0;
    for (size_t w = 0; w < stamp_count / 10; ++w) {
        unsigned long long total = 0;
        for (int i = 0; i < 8; ++i) total += host[w * 10 + i];
        if (!total) continue;
        ++waves;
        for (int i = 0; i < 10; ++i) sum[i] += double(host[w * 10 + i]);
    }
    printf("a wave lives %.0f cycles = %.2f us by the 100 MHz counter: the kernel's clock is %.2f GHz\n", sum[8] / waves,
           sum[9] / waves / 100., sum[8] / sum[9] * 0.1);
    const double blocks = 32., stages = 16.;
    printf("attention_split_kernel<40, %d> with stamps: %.1f us per launch; %zu working waves; cycles per wave:\n", BENCH_PIECES,
           ms * 1e3 / 20, waves);
    const char* what[8] = {"(between blocks: loop, stage switch)", "fragments asked for + S^T issued", "S^T complete + maximum",
                           "exp2 + split of P", "O^T issued", "(blocks -> stage end)", "wait for the next stage's DMA",
                           "workgroup barrier"};
    double total = 0;
    for (int i = 0; i < 8; ++i) total += sum[i] / waves;
    for (int i = 0; i < 8; ++i)
        printf("  %-40s %9.0f  (%5.1f %%)  %7.1f per %s\n", what[i], sum[i] / waves, 100. * sum[i] / waves / total,
               sum[i] / waves / (i >= 5 || i == 0 ? stages : blocks), i >= 5 || i == 0 ? "stage" : "block");
    printf("  total %.0f cycles per wave = %.1f us at 2.1 GHz; matrix pipe alone: %d MFMAs x 32 cycles x 2 waves = %d per block pair\n",
           total, total / 2.1e3, (BENCH_PIECES == 2 ? 21 : BENCH_PIECES == 3 ? 42 : 30), (BENCH_PIECES == 2 ? 21 : BENCH_PIECES == 3 ? 42 : 30) * 64);
    return 0;
}
