// Can front-end waves live in the idle issue slots of a conv workgroup?  The conv K loop
// (kloop.hip: eight MFMA waves, 168 VGPRs, weight chunks of two k-steps so that the workgroup
// holds 114 KB of LDS) on one stream, the product's front-end kernel on another with ONE
// workgroup per CU (four waves, one per SIMD, 34 KB): alone and together.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Iemphases_amd/csrc tools/micro/corun.hip -o tools/micro/bin/corun
#define KLOOP_NO_MAIN
#define KLOOP_CHUNK_STEPS 2
#define BARRIER
#include "kloop.hip"
#include <math.h>
#include "../../emphases_amd/csrc/frontend.hip"

int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 100;          // of two k-steps: ~1.1 us each
    const int fe_grid = argc > 2 ? atoi(argv[2]) : 256;
    // ---- conv stand-in
    std::vector<float> seed(4096);
    for (int i = 0; i < 4096; ++i) seed[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    float *d_seed, *kout;
    unsigned long long* clocks;
    CHECK(hipMalloc(&d_seed, 4096 * 4)); CHECK(hipMalloc(&kout, 256 * 768 * 4)); CHECK(hipMalloc(&clocks, 256 * 16 * 8));
    CHECK(hipMemcpy(d_seed, seed.data(), 4096 * 4, hipMemcpyHostToDevice));
    const size_t conv_lds = (kChannels * kStride + 2 * kChunkFloats) * 4;
    CHECK(hipFuncSetAttribute((const void*)kloop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_lds));
    // ---- the front-end on the configs[1] layout (frontend_bench.hip)
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
    auto fe_kernel = frontend_kernel<0, false>;
    CHECK(hipFuncSetAttribute((const void*)fe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fe_lds));
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
    auto conv = [&] { kloop_kernel<<<256, 512, conv_lds, sa>>>(d_seed, chunks, kout, clocks); };
    auto fe = [&](int grid) {
        fe_kernel<<<grid, 256, fe_lds, sb>>>(audio, seg, dtiles, dtable, dstart, dcount, doffset, dvalues, (int)values.size(),
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
    for (int i = 0; i < 5; ++i) { conv(); fe(768); fe(fe_grid); }
    const double conv_alone = wall(conv);
    const double fe_full = wall([&] { fe(768); });
    const double fe_thin = wall([&] { fe(fe_grid); });
    const double both_thin = wall([&] { conv(); fe(fe_grid); });
    const double both_full = wall([&] { conv(); fe(768); });
    // two lanes, each front-end -> conv loop, over and over (the pipeline's shape)
    auto conv_on = [&](hipStream_t stream) { kloop_kernel<<<256, 512, conv_lds, stream>>>(d_seed, chunks, kout, clocks); };
    auto fe_on = [&](hipStream_t stream) {
        fe_kernel<<<768, 256, fe_lds, stream>>>(audio, seg, dtiles, dtable, dstart, dcount, doffset, dvalues, (int)values.size(),
                                               out, ld, 0, -1, nullptr, nullptr, 0, n_tiles);
    };
    auto lanes = [&](int count) {
        std::vector<double> laps;
        for (int rep = 0; rep < 7; ++rep) {
            CHECK(hipDeviceSynchronize());
            timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (int i = 0; i < 40; ++i) {
                hipStream_t stream = (count == 2 && (i & 1)) ? sb : sa;
                fe_on(stream);
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
    printf("front-end -> conv loop, over and over: %.1f us per batch on one stream, %.1f on two\n", one_lane, two_lanes);
    printf("conv loop (%d chunks of 2 k-steps, %zu KB LDS) alone %.1f us per launch\n", chunks, conv_lds / 1024, conv_alone);
    printf("front-end, 768 workgroups alone %.1f us; %d workgroups alone %.1f us\n", fe_full, fe_grid, fe_thin);
    printf("conv loop + front-end of %d workgroups on two streams: %.1f us per pair (sum of the two alone %.1f)\n", fe_grid,
           both_thin, conv_alone + fe_thin);
    printf("conv loop + front-end of 768 workgroups on two streams: %.1f us per pair (sum %.1f)\n", both_full,
           conv_alone + fe_full);
    return 0;
}
