// Standalone micro-benchmark of emph_conv1d on the C2 layout (64 x 1000 frames).
// NOTE: the conv kernels of the product carry no EMPH_STAMP hooks any more (they distorted
// the kernels they timed, EXPERIMENTS.md, rounds 1-4 section 6): this file times whole launches; the stamp
// machinery below is inert.  In-kernel timelines: tools/micro/stack_bench.hip (STACK_STAMP).
// Build: hipcc -O3 --offload-arch=gfx950 -DEMPH_STAMPS -I. tools/micro/conv_bench.hip \
//            emphases_amd/csrc/frontend.hip -o gpurun_out/conv_bench   (frontend.hip supplies set_error)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#ifdef EMPH_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define EMPH_STAMP(slot)                                                          \
    do {                                                                          \
        if (g_stamps != nullptr && (threadIdx.x & 63) == 0) {                     \
            unsigned long long now = __builtin_amdgcn_s_memrealtime();           \
            g_stamps[(static_cast<size_t>(blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8 + \
                     (slot)] = now;                                               \
            if ((slot) == 2 || (slot) == 3)                                       \
                g_stamps[(static_cast<size_t>(blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8 + \
                         (slot) + 3] = __builtin_amdgcn_s_memtime();              \
        }                                                                         \
    } while (0)
#endif
#include "../../emphases_amd/csrc/conv.hip"
#include "../../emphases_amd/csrc/conv_w4.hip"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// WARM=1: before every timed launch a kernel reads the whole pack on every CU, so
// that the conv kernel's LDS-DMA finds it in its XCD's L2 (is the 2.4 us of the
// pack transfer L2-miss latency?)
__global__ void warm_kernel(const float* pack, int floats, float* sink) {
    float total = 0.f;
    for (int index = threadIdx.x; index < floats; index += blockDim.x) total += pack[index];
    if (total == 123.456f) sink[0] = total;
}
static bool g_wino = false, g_w4 = false;
static float* g_wino_pack = nullptr;
static float* g_w4_pack = nullptr;
static int conv(const float* x, int64_t ld, float* y, const float* pack, const float* bias, int c,
                int ks, const int32_t* tiles, int n_tiles, int tile_n) {
    if (g_w4)
        return emph_conv1d_winograd4(x, ld, y, ld, g_w4_pack, bias, c, c, 1, tiles, n_tiles,
                                     nullptr);
    if (g_wino)
        return emph_conv1d_winograd(x, ld, y, ld, g_wino_pack, bias, c, c, 1, tiles, n_tiles,
                                    tile_n, nullptr);
    return emph_conv1d(x, ld, y, ld, pack, bias, c, c, ks, 1, tiles, n_tiles, tile_n, 0, nullptr);
}

int main(int argc, char** argv) {
    g_wino = getenv("WINO") != nullptr;
    g_w4 = getenv("W4") != nullptr;
    const int segments = 64, frames = 1000, c = 80;
    const int ks = getenv("KS") ? atoi(getenv("KS")) : 3;
    const int64_t ld = 16 + static_cast<int64_t>(segments) * 1008 + 64;
    std::vector<float> hx(c * ld);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    std::vector<float> hw(c * c * ks);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 1000) / 5000.f - 0.1f;
    std::vector<float> hpack(emph_conv_pack_size(c, c, ks));
    emph_conv_pack(hw.data(), c, c, ks, hpack.data());
    std::vector<float> hwino(emph_conv_winograd_pack_size(c, c));
    emph_conv_winograd_pack(hw.data(), c, c, hwino.data());
    std::vector<float> hw4(emph_conv_winograd4_pack_size(c, c));
    emph_conv_winograd4_pack(hw.data(), c, c, hw4.data());
    CHECK(hipMalloc(&g_w4_pack, hw4.size() * 4));
    CHECK(hipMemcpy(g_w4_pack, hw4.data(), hw4.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&g_wino_pack, hwino.size() * 4));
    CHECK(hipMemcpy(g_wino_pack, hwino.data(), hwino.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hbias(c, 0.1f);
    float *x, *y, *pack, *bias;
    CHECK(hipMalloc(&x, hx.size() * 4)); CHECK(hipMalloc(&y, hx.size() * 4));
    CHECK(hipMalloc(&pack, hpack.size() * 4)); CHECK(hipMalloc(&bias, c * 4));
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(pack, hpack.data(), hpack.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(bias, hbias.data(), c * 4, hipMemcpyHostToDevice));
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start)); CHECK(hipEventCreate(&stop));
    for (int tile_n : {64, 32, 16}) {
        if (g_wino && tile_n == 16) continue;
        if (g_w4 && tile_n != 64) continue;
        std::vector<int32_t> tiles;
        for (int s = 0; s < segments; ++s)
            for (int t = 0; t < frames; t += tile_n) {
                tiles.push_back(s); tiles.push_back(t);
                tiles.push_back(16 + s * 1008); tiles.push_back(frames);
            }
        const int n_tiles = tiles.size() / 4;
        int32_t* dtiles;
        CHECK(hipMalloc(&dtiles, tiles.size() * 4));
        CHECK(hipMemcpy(dtiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 5; ++rep)
            conv(x, ld, y, pack, bias, c, ks, dtiles, n_tiles, tile_n);
        CHECK(hipDeviceSynchronize());
        const int reps = 50;
        float ms = 0;
        if (getenv("WARM") || getenv("COLD")) {
            // one event pair per launch, the warming kernel outside it
            for (int rep = 0; rep < reps; ++rep) {
                if (getenv("WARM"))
                    hipLaunchKernelGGL(warm_kernel, dim3(256), dim3(256), 0, 0,
                                       g_w4 ? g_w4_pack : pack,
                                       g_w4 ? (int)hw4.size() : (int)hpack.size(), y);
                CHECK(hipEventRecord(start));
                conv(x, ld, y, pack, bias, c, ks, dtiles, n_tiles, tile_n);
                CHECK(hipEventRecord(stop));
                CHECK(hipEventSynchronize(stop));
                float one = 0;
                CHECK(hipEventElapsedTime(&one, start, stop));
                ms += one;
            }
        } else {
        CHECK(hipEventRecord(start));
        for (int rep = 0; rep < reps; ++rep)
            conv(x, ld, y, pack, bias, c, ks, dtiles, n_tiles, tile_n);
        CHECK(hipEventRecord(stop));
        CHECK(hipEventSynchronize(stop));
        CHECK(hipEventElapsedTime(&ms, start, stop));
        }
        const double us = ms * 1e3 / reps;
        printf("tile %2d: %7.2f us/launch  %6.1f TFLOP/s\n", tile_n, us,
               2.0 * c * c * ks * segments * frames / us * 1e-6);
#ifdef EMPH_STAMPS
        {
            const size_t slots = 512 * 8 * 8;
            unsigned long long* stamps;
            CHECK(hipMalloc(&stamps, slots * 8));
            CHECK(hipMemset(stamps, 0, slots * 8));
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
            conv(x, ld, y, pack, bias, c, ks, dtiles, n_tiles, tile_n);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> host(slots);
            CHECK(hipMemcpy(host.data(), stamps, slots * 8, hipMemcpyDeviceToHost));
            unsigned long long* null_ptr = nullptr;
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &null_ptr, sizeof(null_ptr)));
            unsigned long long first = ~0ull;
            for (size_t i = 0; i < slots; i += 8) if (host[i]) first = std::min(first, host[i]);
            // s_memrealtime ticks at 100 MHz: 10 ns each
            const char* names[8] = {"start", "staged", "setup", "loop", "store", "", "", ""};
            for (int slot = 0; slot < 5; ++slot) {
                std::vector<double> values;
                for (size_t i = 0; i < slots; i += 8)
                    if (host[i] && host[i + slot]) values.push_back((host[i + slot] - first) * 0.01);
                if (values.empty()) continue;
                std::sort(values.begin(), values.end());
                printf("   %-7s waves=%4zu  min %6.2f  median %6.2f  max %6.2f us since first wave start\n",
                       names[slot], values.size(), values.front(), values[values.size() / 2], values.back());
            }
            {
                std::vector<double> clocks;
                for (size_t i = 0; i < slots; i += 8)
                    if (host[i] && host[i + 3] > host[i + 2])
                        clocks.push_back(double(host[i + 6] - host[i + 5]) /
                                         (double(host[i + 3] - host[i + 2]) * 10.0));
                std::sort(clocks.begin(), clocks.end());
                if (!clocks.empty())
                    printf("   shader clock during the K loop: median %.2f GHz (min %.2f max %.2f)\n",
                           clocks[clocks.size() / 2], clocks.front(), clocks.back());
            }
            CHECK(hipFree(stamps));
        }
#endif
        CHECK(hipFree(dtiles));
    }
    return 0;
}
