// Micro-benchmark: emph_conv1d (k=3) vs emph_conv1d_winograd on the C2 layout,
// with a max-abs-difference check between the two.
// Build: hipcc -O3 --offload-arch=gfx950 -Iinclude tools/micro/wino_bench.hip \
//            emphases_amd/csrc/frontend.hip -o tools/micro/bin/wino_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../emphases_amd/csrc/conv.hip"
#undef EMPH_STAMP
#include "../../emphases_amd/csrc/conv_w4.hip"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int segments = 64, ks = 3;
    const int c = argc > 1 ? atoi(argv[1]) : 80;
    const int frames = argc > 2 ? atoi(argv[2]) : 1000;
    const int stride = (frames + 15) / 16 * 16 + 16 - (frames % 16 == 0 ? 8 : 0);
    const int seg_stride = (stride + 15) / 16 * 16;
    const int64_t ld = 16 + static_cast<int64_t>(segments) * seg_stride + 128;
    std::vector<float> hx(c * ld);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    std::vector<float> hw(c * c * ks);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 1000) / 5000.f - 0.1f;
    std::vector<float> hpack(emph_conv_pack_size(c, c, ks));
    emph_conv_pack(hw.data(), c, c, ks, hpack.data());
    std::vector<float> hwino(emph_conv_winograd_pack_size(c, c));
    emph_conv_winograd_pack(hw.data(), c, c, hwino.data());
    std::vector<float> hwino4(emph_conv_winograd4_pack_size(c, c));
    emph_conv_winograd4_pack(hw.data(), c, c, hwino4.data());
    float* wino4;
    CHECK(hipMalloc(&wino4, hwino4.size() * 4));
    CHECK(hipMemcpy(wino4, hwino4.data(), hwino4.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hbias(c, 0.1f);
    float *x, *y, *z, *pack, *wino, *bias;
    CHECK(hipMalloc(&x, hx.size() * 4)); CHECK(hipMalloc(&y, hx.size() * 4));
    CHECK(hipMalloc(&z, hx.size() * 4));
    CHECK(hipMalloc(&pack, hpack.size() * 4)); CHECK(hipMalloc(&bias, c * 4));
    CHECK(hipMalloc(&wino, hwino.size() * 4));
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(pack, hpack.data(), hpack.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(wino, hwino.data(), hwino.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(bias, hbias.data(), c * 4, hipMemcpyHostToDevice));
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start)); CHECK(hipEventCreate(&stop));
    for (int tile_n : {64, 32}) {
        std::vector<int32_t> tiles;
        for (int s = 0; s < segments; ++s)
            for (int t = 0; t < frames; t += tile_n) {
                tiles.push_back(s); tiles.push_back(t);
                tiles.push_back(16 + s * seg_stride); tiles.push_back(frames);
            }
        const int n_tiles = tiles.size() / 4;
        int32_t* dtiles;
        CHECK(hipMalloc(&dtiles, tiles.size() * 4));
        CHECK(hipMemcpy(dtiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(y, 0, hx.size() * 4)); CHECK(hipMemset(z, 0, hx.size() * 4));
        for (int which = 0; which < 3; ++which) {
            if (which == 2 && tile_n != 64) continue;
            auto run = [&]() {
                int status = which == 0
                    ? emph_conv1d(x, ld, y, ld, pack, bias, c, c, ks, 1, dtiles, n_tiles, tile_n, 0, nullptr)
                    : which == 1
                    ? emph_conv1d_winograd(x, ld, z, ld, wino, bias, c, c, 1, dtiles, n_tiles, tile_n, nullptr)
                    : emph_conv1d_winograd4(x, ld, z, ld, wino4, bias, c, c, 1, dtiles, n_tiles, nullptr);
                if (status) { printf("status %d: %s\n", status, emph_last_error()); exit(1); }
            };
            for (int rep = 0; rep < 5; ++rep) run();
            CHECK(hipDeviceSynchronize());
            const int reps = 50;
            CHECK(hipEventRecord(start));
            for (int rep = 0; rep < reps; ++rep) run();
            CHECK(hipEventRecord(stop));
            CHECK(hipEventSynchronize(stop));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, start, stop));
            const double us = ms * 1e3 / reps;
            printf("%s tile %2d: %7.2f us/launch  %6.1f direct-equivalent TFLOP/s\n",
                   which == 2 ? "F(4,3)  " : which ? "winograd" : "direct  ", tile_n, us,
                   2.0 * c * c * ks * segments * frames / us * 1e-6);
        }
        std::vector<float> hy(hx.size()), hz(hx.size());
        CHECK(hipMemcpy(hy.data(), y, hx.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hz.data(), z, hx.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (int ch = 0; ch < c; ++ch)
            for (int s = 0; s < segments; ++s)
                for (int t = 0; t < frames; ++t) {
                    const size_t i = ch * ld + 16 + s * seg_stride + t;
                    worst = fmax(worst, fabs((double)hy[i] - hz[i]));
                    scale = fmax(scale, fabs((double)hy[i]));
                }
        printf("   max |direct - winograd| = %.3g (output scale %.3g)\n", worst, scale);
        CHECK(hipFree(dtiles));
    }
    return 0;
}
