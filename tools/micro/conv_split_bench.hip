// conv1d_split_kernel (emphases_amd/csrc/conv_split.hip) with an in-kernel timeline: where do an
// MFMA wave's cycles go?  s_memtime at the phase boundaries, summed per wave, averaged.
//   0 input staged | 1 a tap's MFMAs issued (up to the next barrier) | 2 waited at the chunk's
//   barrier | 3 waited for the layer's last reader | 4 output split + written | 5 output stored
// BASELINE configs[1]: 64 segments of 1000 positions; LAYERS layers in the launch (default 4).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Iemphases_amd/csrc \
//        tools/micro/conv_split_bench.hip -o tools/micro/bin/conv_split_bench [-DLAYERS=3]
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CONV_STAMP(slot)                                                         \
    do {                                                                         \
        const unsigned long long now = __builtin_amdgcn_s_memtime();             \
        stamp_sum[(slot)] += now - stamp_last;                                   \
        stamp_last = now;                                                        \
    } while (0)
#define CONV_STAMP_ARGUMENT , unsigned long long* __restrict__ stamp_out
#define CONV_STAMP_PASS , nullptr
#define CONV_STAMP_DECLARE                                                       \
    unsigned long long stamp_sum[6] = {0, 0, 0, 0, 0, 0};                        \
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memtime(),         \
                             stamp_real = __builtin_amdgcn_s_memrealtime();      \
    unsigned long long stamp_last = stamp_begin;
#define CONV_STAMP_FINISH                                                        \
    if (stamp_out != nullptr && (threadIdx.x & 63) == 0)                         \
        for (int i = 0; i < 8; ++i)                                              \
            stamp_out[(static_cast<size_t>(blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8 + i] =           \
                i < 6 ? stamp_sum[i] : i == 6 ? __builtin_amdgcn_s_memtime() - stamp_begin            \
                                              : __builtin_amdgcn_s_memrealtime() - stamp_real;
#include "conv_split.hip"
namespace emph {
static char bench_error[512];
void set_error(const char* format, ...) {
    va_list args;
    va_start(args, format);
    vsnprintf(bench_error, sizeof(bench_error), format, args);
    va_end(args);
}
}  // namespace emph
extern "C" int32_t emph_conv_stack_spans(const int64_t*, const int64_t*, int32_t, int32_t*) __attribute__((weak));

#ifndef LAYERS
#define LAYERS 4
#endif
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
    const int segments = 64, frames = 1000;
    std::vector<int32_t> spans;
    int ld = 16;
    for (int i = 0; i < segments; ++i) {
        // the span table of emph_conv_stack_spans for 1000 positions: 252 + 248 + 248 + 252
        const int first[4] = {0, 252, 500, 748}, own[4] = {252, 248, 248, 252};
        for (int k = 0; k < 4; ++k)
            spans.insert(spans.end(), {i, first[k], ld, frames, own[k], first[k] ? first[k] - 4 : 0, 0, 0});
        ld += (frames + 15) / 16 * 16;
    }
    ld += 128;
    const int n_spans = int(spans.size() / 8);
    std::vector<float> x(size_t(80) * ld), weight(80 * 80 * 3), bias(LAYERS * 80, 0.1f);
    unsigned state = 777;
    auto uniform = [&] { state = state * 1664525u + 1013904223u; return float(state >> 8) / float(1 << 24) - 0.5f; };
    for (auto& v : x) v = 2.f * uniform();
    std::vector<unsigned char> packs(size_t(LAYERS) * emph_conv_split_pack_size());
    for (int l = 0; l < LAYERS; ++l) {
        for (auto& v : weight) v = 0.25f * uniform();
        emph_conv_split_pack(weight.data(), packs.data() + size_t(l) * emph_conv_split_pack_size());
    }
    float *d_x, *d_y, *d_bias;
    unsigned char* d_packs;
    int32_t* d_spans;
    unsigned long long* stamps;
    CHECK(hipMalloc(&d_x, x.size() * 4)); CHECK(hipMalloc(&d_y, x.size() * 4));
    CHECK(hipMalloc(&d_bias, bias.size() * 4)); CHECK(hipMalloc(&d_packs, packs.size()));
    CHECK(hipMalloc(&d_spans, spans.size() * 4));
    const size_t stamp_count = size_t(n_spans) * 8 * 8;
    CHECK(hipMalloc(&stamps, stamp_count * 8));
    CHECK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_packs, packs.data(), packs.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_spans, spans.data(), spans.size() * 4, hipMemcpyHostToDevice));
    auto kernel = emph::conv1d_split_kernel<false>;
    CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, emph::kSplitLdsBytes));
    auto launch = [&] {
        hipLaunchKernelGGL(kernel, dim3(n_spans), dim3(emph::kSplitThreads), emph::kSplitLdsBytes, 0, d_x, (int64_t)ld, d_y,
                           (int64_t)ld, d_packs, d_bias, LAYERS, (1 << LAYERS) - 1, d_spans, (const int32_t*)nullptr, stamps);
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
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
    double sum[8] = {0};
    const size_t waves = stamp_count / 8;
    for (size_t w = 0; w < waves; ++w)
        for (int i = 0; i < 8; ++i) sum[i] += double(host[w * 8 + i]);
    printf("conv1d_split_kernel, %d layers, %d spans: %.1f us per launch; an MFMA wave lives %.0f cycles = %.2f us: %.2f GHz\n",
           LAYERS, n_spans, ms * 1e3 / 20, sum[6] / waves, sum[7] / waves / 100., sum[6] / sum[7] * 0.1);
    const char* what[6] = {"input staged", "MFMAs of the taps issued", "waited at a chunk's barrier",
                           "waited for the layer's last reader", "output split + written", "output stored"};
    for (int i = 0; i < 6; ++i)
        printf("  %-36s %9.0f cycles (%5.1f %%)\n", what[i], sum[i] / waves, 100. * sum[i] / sum[6]);
    printf("  matrix pipe alone: %d layers x 135 MFMAs x 32 cycles x 2 waves = %d cycles\n", LAYERS, LAYERS * 135 * 64);
    return 0;
}
