// Standalone timeline of emph_word_decoder on the C2 layout (64 x 1000 frames, `words` words each).
// NOTE: the decoder kernels of the product carry no EMPH_STAMP hooks any more (they distorted
// the kernels they timed, EXPERIMENTS.md, rounds 1-4 section 6): this file times whole launches; the stamp
// machinery below is inert.  In-kernel timelines: tools/micro/stack_bench.hip (STACK_STAMP).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
__device__ unsigned long long* g_stamps = nullptr;
#define EMPH_STAMP(slot)                                                          \
    do {                                                                          \
        if (g_stamps != nullptr && (threadIdx.x & 63) == 0)                       \
            g_stamps[(static_cast<size_t>(blockIdx.x) * 16 + (threadIdx.x >> 6)) * 16 + \
                     (slot)] = __builtin_amdgcn_s_memrealtime();                  \
    } while (0)
#include "../../emphases_amd/csrc/decoder.hip"
#include "../../emphases_amd/csrc/conv.hip"
// stub for the library's error slot (lives in frontend.hip)
namespace emph { void set_error(const char*, ...) {} }
extern "C" const char* emph_last_error(void) { return ""; }
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int segments = 64, frames = 1000, c = 80, ks = 3, layers = 6;
    const int words = argc > 1 ? atoi(argv[1]) : 197;      // per utterance (C2: ~197)
    const int wstride = (words + 15) / 16 * 16;
    const int64_t ld = 16 + static_cast<int64_t>(segments) * 1008 + 128;
    const int64_t ldw = 16 + segments * wstride + 128;
    std::vector<float> hx(c * ld, 0.25f);
    std::vector<int32_t> hbounds(2 * ldw, 0);
    std::vector<int64_t> hseg(segments * 8, 0);
    std::vector<int32_t> tiles;
    const int block = emph_word_decoder_block(layers, ks, ks);
    for (int s = 0; s < segments; ++s) {
        hseg[s * 8 + 4] = 16 + s * 1008; hseg[s * 8 + 5] = frames;
        hseg[s * 8 + 6] = 16 + s * wstride; hseg[s * 8 + 7] = words;
        for (int w = 0; w < words; ++w) {
            hbounds[16 + s * wstride + w] = w * frames / words;
            hbounds[ldw + 16 + s * wstride + w] = (w + 1) * frames / words;
        }
        for (int t = 0; t < words; t += block) {
            tiles.push_back(s); tiles.push_back(t); tiles.push_back(16 + s * wstride); tiles.push_back(words);
        }
    }
    std::vector<float> hw(c * c * ks, 0.01f), hpack(emph_word_decoder_pack_size(c, ks));
    emph_word_decoder_pack(hw.data(), c, ks, hpack.data());
    std::vector<float> hpacks;
    for (int l = 0; l < layers; ++l) hpacks.insert(hpacks.end(), hpack.begin(), hpack.end());
    std::vector<float> hbias(layers * c, 0.1f), how(c * ks, 0.01f), hob(1, 0.f);
    float *x, *packs, *biases, *ow, *ob, *logits, *scores; int32_t *bounds, *dtiles; int64_t* seg;
    CHECK(hipMalloc(&x, hx.size() * 4)); CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&packs, hpacks.size() * 4)); CHECK(hipMemcpy(packs, hpacks.data(), hpacks.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&biases, hbias.size() * 4)); CHECK(hipMemcpy(biases, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&ow, how.size() * 4)); CHECK(hipMemcpy(ow, how.data(), how.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&ob, 4)); CHECK(hipMemcpy(ob, hob.data(), 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&logits, ldw * 4)); CHECK(hipMalloc(&scores, ldw * 4));
    CHECK(hipMalloc(&bounds, hbounds.size() * 4)); CHECK(hipMemcpy(bounds, hbounds.data(), hbounds.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dtiles, tiles.size() * 4)); CHECK(hipMemcpy(dtiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&seg, hseg.size() * 8)); CHECK(hipMemcpy(seg, hseg.data(), hseg.size() * 8, hipMemcpyHostToDevice));
    const int n_tiles = tiles.size() / 4;
    auto launch = [&]() {
        int status = emph_word_decoder(x, ld, dtiles, n_tiles, c, packs, biases,
                                       layers, ks, 1, ow, ob, ks, 1, logits, scores, nullptr);
        if (status) { printf("launch failed %d %s\n", status, emph_last_error()); exit(1); }
    };
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    hipEvent_t start, stop; CHECK(hipEventCreate(&start)); CHECK(hipEventCreate(&stop));
    CHECK(hipEventRecord(start));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(stop)); CHECK(hipEventSynchronize(stop));
    float ms; CHECK(hipEventElapsedTime(&ms, start, stop));
    printf("word decoder: %d tiles, %.2f us/launch\n", n_tiles, ms * 1e3 / 20);
    const size_t slots = static_cast<size_t>(n_tiles) * 16 * 16;
    unsigned long long* stamps; CHECK(hipMalloc(&stamps, slots * 8)); CHECK(hipMemset(stamps, 0, slots * 8));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
    launch(); CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> host(slots);
    CHECK(hipMemcpy(host.data(), stamps, slots * 8, hipMemcpyDeviceToHost));
    unsigned long long first = ~0ull;
    for (size_t i = 0; i < slots; i += 16) if (host[i]) first = std::min(first, host[i]);
    const char* names[16] = {"start", "tables", "-", "layer1", "layer2", "layer3", "layer4", "layer5", "layer6", "output", "L2 open", "L2 trip0", "L2 trip1", "L2 trip2", "L2 trip3", "L2 trip4"};
    for (int slot = 0; slot < 16; ++slot) {
        std::vector<double> values;
        for (size_t i = 0; i < slots; i += 16) if (host[i] && host[i + slot]) values.push_back((host[i + slot] - first) * 0.01);
        if (values.empty()) continue;
        std::sort(values.begin(), values.end());
        printf("   %-8s waves=%4zu  min %7.2f  median %7.2f  max %7.2f us\n", names[slot], values.size(), values.front(), values[values.size() / 2], values.back());
    }
    return 0;
}
