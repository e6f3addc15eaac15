// Can the front-end run UNDER the split (bf16x3) conv stack?  VALU issue is free beside bf16
// MFMAs (profiles/r5_coexec.txt), but conv1d_split_kernel as shipped takes 156 KB of LDS and
// three waves of 164 VGPRs per SIMD: no front-end wave fits on its CU.  This builds the kernel
// 128 positions wide (-DCONV_SPLIT_WIDTH=128: four MFMA waves + four loader waves = two waves a
// SIMD, 111 KB of LDS) so that ONE four-wave front-end workgroup (34 KB, 162 VGPRs) fits beside
// it, and times on BASELINE configs[1] (64 x 1000 frames, four layers in the launch):
//   the conv launch alone (128 and, built separately, 256 wide); the front-end alone at full
//   occupancy and held to one workgroup per CU (LDS padded); the two on two streams at once;
//   and the pipeline's shape - two lanes of front-end -> conv.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Iemphases_amd/csrc \
//        -DCONV_SPLIT_WIDTH=128 tools/micro/corun_split.hip -o tools/micro/bin/corun_split128
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <algorithm>
#include <vector>

#include "conv_split.hip"
#include "frontend.hip"
// (frontend.hip defines emph::set_error for the library)

#ifndef LAYERS
#define LAYERS 4
#endif
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int segments = 64, frames = 1000, samples = 160000;
    const int width = emph::kSplitWidth;
    // ---- conv: spans of `width` computed positions, a quad of halo at every inner end
    std::vector<int32_t> spans;
    int ldc = 16;
    for (int i = 0; i < segments; ++i) {
        for (int first = 0; first < frames;) {
            const int c0 = first ? first - 4 : 0;
            const int reach = c0 + width >= frames ? frames : c0 + width - 4;     // own up to here
            spans.insert(spans.end(), {i, first, ldc, frames, reach - first, c0, 0, 0});
            first = reach;
        }
        ldc += (frames + 15) / 16 * 16;
    }
    ldc += 128;
    const int n_spans = int(spans.size() / 8);
    std::vector<float> x(size_t(80) * ldc), weight(80 * 80 * 3), bias(LAYERS * 80, 0.1f);
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
    CHECK(hipMalloc(&d_x, x.size() * 4)); CHECK(hipMalloc(&d_y, x.size() * 4));
    CHECK(hipMalloc(&d_bias, bias.size() * 4)); CHECK(hipMalloc(&d_packs, packs.size()));
    CHECK(hipMalloc(&d_spans, spans.size() * 4));
    CHECK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_packs, packs.data(), packs.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_spans, spans.data(), spans.size() * 4, hipMemcpyHostToDevice));
    auto conv_kernel = emph::conv1d_split_kernel<false>;
    CHECK(hipFuncSetAttribute((const void*)conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, emph::kSplitLdsBytes));
    // ---- the front-end on the same layout (frontend_bench.hip)
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
    const size_t fe_lds = frontend_lds_bytes(false);
    const size_t fe_padded = 100 * 1024;          // one workgroup per CU
    auto fe_kernel = frontend_kernel<0, false>;
    CHECK(hipFuncSetAttribute((const void*)fe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fe_padded));
    const int fe_grid = frontend_grid(n_tiles);
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
    auto conv_on = [&](hipStream_t stream) {
        hipLaunchKernelGGL(conv_kernel, dim3(n_spans), dim3(emph::kSplitThreads), emph::kSplitLdsBytes, stream, d_x,
                           (int64_t)ldc, d_y, (int64_t)ldc, d_packs, d_bias, LAYERS, (1 << LAYERS) - 1, d_spans,
                           (const int32_t*)nullptr);
    };
    auto fe_on = [&](hipStream_t stream, size_t lds) {
        fe_kernel<<<fe_grid, 256, lds, stream>>>(audio, seg, dtiles, dtable, dstart, dcount, doffset, dvalues, (int)values.size(),
                                                 out, ld, 0, -1, nullptr, nullptr, 0, n_tiles);
    };
    auto wall = [&](auto&& run) {
        std::vector<double> laps;
        for (int rep = 0; rep < 9; ++rep) {
            CHECK(hipDeviceSynchronize());
            timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (int i = 0; i < 20; ++i) run();
            CHECK(hipDeviceSynchronize());
            clock_gettime(CLOCK_MONOTONIC, &t1);
            laps.push_back(((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 20e3);
        }
        std::sort(laps.begin(), laps.end());
        return laps[4];
    };
    for (int i = 0; i < 5; ++i) { conv_on(sa); fe_on(sb, fe_lds); fe_on(sb, fe_padded); }
    const double conv_alone = wall([&] { conv_on(sa); });
    const double fe_full = wall([&] { fe_on(sb, fe_lds); });
    const double fe_thin = wall([&] { fe_on(sb, fe_padded); });
    const double both = wall([&] { conv_on(sa); fe_on(sb, fe_lds); });
    const double both_fe_first = wall([&] { fe_on(sb, fe_lds); conv_on(sa); });
    auto lanes = [&](int count_) {
        std::vector<double> laps;
        for (int rep = 0; rep < 7; ++rep) {
            CHECK(hipDeviceSynchronize());
            timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (int i = 0; i < 40; ++i) {
                hipStream_t stream = (count_ == 2 && (i & 1)) ? sb : sa;
                fe_on(stream, fe_lds);
                conv_on(stream);
            }
            CHECK(hipDeviceSynchronize());
            clock_gettime(CLOCK_MONOTONIC, &t1);
            laps.push_back(((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 40e3);
        }
        std::sort(laps.begin(), laps.end());
        return laps[3];
    };
    const double one_lane = lanes(1), two_lanes = lanes(2);
    printf("conv1d_split_kernel %d positions wide (%d threads, %d KB LDS, %d spans, %d layers): %.1f us per launch alone\n",
           width, emph::kSplitThreads, emph::kSplitLdsBytes / 1024, n_spans, LAYERS, conv_alone);
    printf("front-end (%d workgroups, %zu KB LDS): %.1f us alone; held to one workgroup per CU (LDS padded to %zu KB): %.1f us\n",
           fe_grid, fe_lds / 1024, fe_full, fe_padded / 1024, fe_thin);
    printf("conv + front-end on two streams at once: %.1f us per pair (conv launched first), %.1f (front-end first); "
           "one after the other %.1f\n", both, both_fe_first, conv_alone + fe_full);
    printf("front-end -> conv, over and over: %.1f us per batch on one stream, %.1f on two\n", one_lane, two_lanes);
    return 0;
}
