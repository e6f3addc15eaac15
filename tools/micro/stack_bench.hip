// Standalone micro-benchmark of emph_conv1d_stack on the configs[1] layout (64 x 1000
// frames = 256 spans), against the same layers as emph_conv1d_winograd4 launches.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Iemphases_amd/csrc \
//            tools/micro/stack_bench.hip emphases_amd/csrc/frontend.hip -o tools/micro/bin/stack_bench
//        (frontend.hip supplies set_error; -DSTACK_STAMPS adds the in-kernel timeline)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#ifdef STACK_STAMPS
__device__ unsigned long long* g_stack_stamps = nullptr;
// slot per (block, wave): realtime stamps (100 MHz)
// (and, in the second half of the table, the shader clock's own counter: cycles per stamp
// interval over its microseconds is the clock the kernel ran at)
#define STACK_STAMP(slot)                                                                   \
    do {                                                                                    \
        if (g_stack_stamps != nullptr && (threadIdx.x & 63) == 0) {                         \
            const size_t at = (static_cast<size_t>(blockIdx.x) * 12 + (threadIdx.x >> 6)) * 32 + (slot); \
            g_stack_stamps[at] = __builtin_amdgcn_s_memrealtime();                          \
            g_stack_stamps[at + static_cast<size_t>(gridDim.x) * 12 * 32] = __builtin_amdgcn_s_memtime(); \
        }                                                                                   \
    } while (0)
#endif
#include "../../emphases_amd/csrc/conv_w4.hip"
#include "../../emphases_amd/csrc/conv_stack.hip"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int segments = 64, frames = 1000, c = 80;
    const int layers = argc > 1 ? atoi(argv[1]) : 3;
    const int64_t ld = 16 + static_cast<int64_t>(segments) * 1008 + 128;
    std::vector<float> hx(c * ld);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    std::vector<float> hw(static_cast<size_t>(layers) * c * c * 3);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 1000) / 5000.f - 0.1f;
    const int64_t pack_floats = emph_conv_winograd4_pack_size(c, c);
    std::vector<float> hp(static_cast<size_t>(layers) * pack_floats);
    for (int l = 0; l < layers; ++l)
        emph_conv_winograd4_pack(hw.data() + static_cast<size_t>(l) * c * c * 3, c, c,
                                 hp.data() + static_cast<size_t>(l) * pack_floats);
    std::vector<float> hb(layers * c, 0.01f);
    std::vector<int64_t> counts(segments, frames), offsets(segments);
    for (int s = 0; s < segments; ++s) offsets[s] = 16 + 1008 * s;
    const int n_spans = emph_conv_stack_spans(counts.data(), offsets.data(), segments, nullptr);
    std::vector<int32_t> hs(static_cast<size_t>(n_spans) * 8);
    emph_conv_stack_spans(counts.data(), offsets.data(), segments, hs.data());
    std::vector<int32_t> ht;
    for (int s = 0; s < segments; ++s)
        for (int t = 0; t < frames; t += 64) {
            ht.push_back(s), ht.push_back(t), ht.push_back((int)offsets[s]), ht.push_back(frames);
        }
    const int n_tiles = static_cast<int>(ht.size() / 4);
    float *x, *y, *z, *packs, *biases;
    int32_t *spans, *tiles;
    CHECK(hipMalloc(&x, hx.size() * 4)); CHECK(hipMalloc(&y, hx.size() * 4)); CHECK(hipMalloc(&z, hx.size() * 4));
    CHECK(hipMalloc(&packs, hp.size() * 4)); CHECK(hipMalloc(&biases, hb.size() * 4));
    CHECK(hipMalloc(&spans, hs.size() * 4)); CHECK(hipMalloc(&tiles, ht.size() * 4));
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(packs, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(biases, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(spans, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(tiles, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    const int relu = 0b110;
    auto stack = [&] {
        if (emph_conv1d_stack(x, ld, y, ld, packs, biases, layers, relu, spans, n_spans, nullptr, nullptr)) {
            printf("stack failed: %s\n", emph_last_error());
            exit(1);
        }
    };
    auto layered = [&] {
        const float* in = x;
        float* out[2] = {y, z};
        for (int l = 0; l < layers; ++l) {
            emph_conv1d_winograd4(in, ld, out[l & 1], ld, packs + l * pack_floats, biases + l * c, c, c,
                                  (relu >> l) & 1, tiles, n_tiles, nullptr);
            in = out[l & 1];
        }
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto&& run) {
        for (int i = 0; i < 20; ++i) run();
        CHECK(hipDeviceSynchronize());
        std::vector<float> laps;
        for (int rep = 0; rep < 7; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < 50; ++i) run();
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            laps.push_back(ms / 50 * 1e3f);
        }
        std::sort(laps.begin(), laps.end());
        printf("%-28s %d layers: %7.2f us per pass (%.2f per layer), %d spans / %d tiles\n", name, layers, laps[3],
               laps[3] / layers, n_spans, n_tiles);
    };
    time("emph_conv1d_stack", stack);
    time("emph_conv1d_winograd4 x L", layered);
#ifdef STACK_STAMPS
    unsigned long long* stamps;
    const size_t count = static_cast<size_t>(n_spans) * 12 * 32;
    CHECK(hipMalloc(&stamps, 2 * count * 8)); CHECK(hipMemset(stamps, 0, 2 * count * 8));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stack_stamps), &stamps, sizeof(stamps)));
    stack(); CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs2(2 * count);
    CHECK(hipMemcpy(hs2.data(), stamps, 2 * count * 8, hipMemcpyDeviceToHost));
    unsigned long long first = ~0ull;
    for (size_t i = 0; i < count; i += 32) if (hs2[i]) first = std::min(first, hs2[i]);
    // mean over blocks of wave 0 (MFMA) and wave 8 (loader) stamps, in us from the first stamp
    for (int wave : {0, 4, 8}) {
        printf("wave %2d:", wave);
        for (int slot = 0; slot < 32; ++slot) {
            double total = 0; int n = 0;
            for (int b = 0; b < n_spans; ++b) {
                const unsigned long long v = hs2[(static_cast<size_t>(b) * 12 + wave) * 32 + slot];
                if (v) total += double(v - first) / 100., ++n;
            }
            if (n) printf(" [%d] %.2f", slot, total / n);
        }
        printf("\n");
    }
    // the clock between consecutive stamps of wave 0 (mean over blocks)
    printf("GHz  0:");
    int before = -1;
    for (int slot = 0; slot < 32; ++slot) {
        double cycles = 0, micros = 0;
        bool have = false;
        for (int b = 0; b < n_spans; ++b) {
            const size_t at = (static_cast<size_t>(b) * 12) * 32;
            if (!hs2[at + slot]) continue;
            have = true;
            if (before >= 0) {
                micros += double(hs2[at + slot] - hs2[at + before]) / 100.;
                cycles += double(hs2[count + at + slot] - hs2[count + at + before]);
            }
        }
        if (have) {
            if (before >= 0 && micros > 0) printf(" [%d] %.2f", slot, cycles / micros / 1e3);
            before = slot;
        }
    }
    printf("\n");
#endif
    return 0;
}
