// Front-end + seven F(4,3) conv layers per batch, two batches in flight, as HIP
// graphs: the layer as ONE launch (what shipped until round 3) against the layer as
// two independent half launches on two streams (emph_conv1d_winograd4_half).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Iemphases_amd/csrc \
//            tools/micro/split_bench.hip -o tools/micro/bin/split_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include <chrono>
#include "../../emphases_amd/csrc/frontend.hip"
#include "../../emphases_amd/csrc/conv_w4.hip"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define OK(x) do { int st = (x); if (st) { printf("%s failed: %d %s\n", #x, st, emph_last_error()); exit(1); } } while (0)

struct Lane {
    hipStream_t main, side;
    hipEvent_t fork, join_a, join_b;
    float *audio_out, *a, *b;
    hipGraphExec_t graph;
};

int main(int argc, char** argv) {
    const int segments = 64, frames = 1000, samples = 160000, c = 80, layers = 7;
    const int64_t ld = 16 + static_cast<int64_t>(segments) * 1008 + 128;
    const bool with_frontend = !(argc > 1 && atoi(argv[1]) == 0);
    // ---- front-end inputs (as tools/micro/frontend_bench.hip)
    std::vector<float> haudio(static_cast<size_t>(segments) * samples);
    for (size_t i = 0; i < haudio.size(); ++i) haudio[i] = 0.1f * sinf(0.01f * (i % 100000)) + 1e-3f * ((i * 2654435761u) % 1000) / 1000.f;
    std::vector<int64_t> hseg(segments * 8, 0);
    std::vector<int32_t> fe_tiles, conv_tiles;
    for (int s = 0; s < segments; ++s) {
        hseg[s * 8 + 0] = static_cast<int64_t>(s) * samples; hseg[s * 8 + 1] = samples;
        hseg[s * 8 + 2] = 0; hseg[s * 8 + 3] = samples;
        hseg[s * 8 + 4] = 16 + s * 1008; hseg[s * 8 + 5] = frames;
        for (int t = 0; t < frames; t += emph_frontend_block()) { fe_tiles.push_back(s); fe_tiles.push_back(t); fe_tiles.push_back(16 + s * 1008); fe_tiles.push_back(frames); }
        for (int t = 0; t < frames; t += 64) { conv_tiles.push_back(s); conv_tiles.push_back(t); conv_tiles.push_back(16 + s * 1008); conv_tiles.push_back(frames); }
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
    float *audio, *dtable, *dvalues; int64_t* seg; int32_t *dfe, *dconv, *dstart, *dcount, *doffset;
    CHECK(hipMalloc(&audio, haudio.size() * 4)); CHECK(hipMemcpy(audio, haudio.data(), haudio.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dtable, table.size() * 4)); CHECK(hipMemcpy(dtable, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dvalues, values.size() * 4)); CHECK(hipMemcpy(dvalues, values.data(), values.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&seg, hseg.size() * 8)); CHECK(hipMemcpy(seg, hseg.data(), hseg.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dfe, fe_tiles.size() * 4)); CHECK(hipMemcpy(dfe, fe_tiles.data(), fe_tiles.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dconv, conv_tiles.size() * 4)); CHECK(hipMemcpy(dconv, conv_tiles.data(), conv_tiles.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dstart, 320)); CHECK(hipMemcpy(dstart, start.data(), 320, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dcount, 320)); CHECK(hipMemcpy(dcount, count.data(), 320, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&doffset, 320)); CHECK(hipMemcpy(doffset, offset.data(), 320, hipMemcpyHostToDevice));
    const int n_fe = fe_tiles.size() / 4, n_conv = conv_tiles.size() / 4;
    // ---- conv weights
    std::vector<float> hw(c * c * 3);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 1000) / 5000.f - 0.1f;
    std::vector<float> whole(emph_conv_winograd4_pack_size(c, c)), split(whole.size());
    OK(emph_conv_winograd4_pack(hw.data(), c, c, whole.data()));
    OK(emph_conv_winograd4_split_pack(hw.data(), c, c, split.data()));
    std::vector<float> hbias(c, 0.01f);
    float *dwhole, *dsplit, *dbias;
    CHECK(hipMalloc(&dwhole, whole.size() * 4)); CHECK(hipMemcpy(dwhole, whole.data(), whole.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dsplit, split.size() * 4)); CHECK(hipMemcpy(dsplit, split.data(), split.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dbias, c * 4)); CHECK(hipMemcpy(dbias, hbias.data(), c * 4, hipMemcpyHostToDevice));

    // ---- correctness: the two halves give the whole layer's bits
    {
        float *x, *y0, *y1;
        CHECK(hipMalloc(&x, c * ld * 4)); CHECK(hipMalloc(&y0, c * ld * 4)); CHECK(hipMalloc(&y1, c * ld * 4));
        std::vector<float> hx(c * ld);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
        CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(y0, 0, c * ld * 4)); CHECK(hipMemset(y1, 0, c * ld * 4));
        OK(emph_conv1d_winograd4(x, ld, y0, ld, dwhole, dbias, c, c, 1, dconv, n_conv, nullptr));
        OK(emph_conv1d_winograd4_half(x, ld, y1, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 0, nullptr));
        OK(emph_conv1d_winograd4_half(x, ld, y1, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 1, nullptr));
        CHECK(hipDeviceSynchronize());
        std::vector<float> h0(c * ld), h1(c * ld);
        CHECK(hipMemcpy(h0.data(), y0, h0.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h1.data(), y1, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; double sum = 0;
        for (size_t i = 0; i < h0.size(); ++i) { bad += h0[i] != h1[i]; sum += h0[i]; }
        printf("halves vs whole: %zu of %zu values differ (checksum %.6f)\n", bad, h0.size(), sum);
        CHECK(hipFree(x)); CHECK(hipFree(y0)); CHECK(hipFree(y1));
    }

    // ---- the halves by themselves, and side by side without any dependency
    {
        float *x, *y;
        CHECK(hipMalloc(&x, c * ld * 4)); CHECK(hipMalloc(&y, c * ld * 4));
        CHECK(hipMemset(x, 0, c * ld * 4));
        hipStream_t s0, s1; CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
        auto timed = [&](const char* what, auto body) {
            for (int i = 0; i < 10; ++i) body();
            CHECK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < 100; ++i) body();
            CHECK(hipDeviceSynchronize());
            auto t1 = std::chrono::high_resolution_clock::now();
            printf("%-46s %.2f us per layer\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count() / 100);
        };
        timed("whole layer, one stream", [&] { OK(emph_conv1d_winograd4(x, ld, y, ld, dwhole, dbias, c, c, 1, dconv, n_conv, s0)); });
        // inputs that every XCD's L2 holds (rows 1024 floats apart alias into 0.6 MB): what
        // would the layer gain if its activations were always an L2 hit?
        timed("whole layer, inputs L2-resident (aliased rows)", [&] { OK(emph_conv1d_winograd4(x, 1024, y, ld, dwhole, dbias, c, c, 1, dconv, n_conv, s0)); });
        timed("half 0 alone", [&] { OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 0, s0)); });
        timed("half 1 alone", [&] { OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 1, s0)); });
        timed("half 0 and half 1, two streams, no dependency", [&] {
            OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 0, s0));
            OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 1, s1)); });
        timed("half 0 twice, two streams", [&] {
            OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 0, s0));
            OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 0, s1)); });
        timed("half 1 twice, two streams", [&] {
            OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 1, s0));
            OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 1, s1)); });
        // ---- does a front-end workgroup run BESIDE a conv workgroup on the same CU?
        float* mel; CHECK(hipMalloc(&mel, 80 * ld * 4));
        auto frontend = [&](hipStream_t stream) {
            OK(emph_logmel(audio, 0, seg, dfe, n_fe, dtable, dstart, dcount, doffset, dvalues, (int)values.size(),
                           mel, ld, 0, -1, nullptr, nullptr, 0, stream));
        };
        auto pair = [&](const char* what, int convs, int fronts, auto conv_body) {
            for (int warm = 0; warm < 2; ++warm) { for (int i = 0; i < convs; ++i) conv_body(s0); for (int i = 0; i < fronts; ++i) frontend(s1); }
            CHECK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < convs; ++i) conv_body(s0);
            CHECK(hipDeviceSynchronize());
            auto t1 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < fronts; ++i) frontend(s1);
            CHECK(hipDeviceSynchronize());
            auto t2 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < std::max(convs, fronts); ++i) { if (i < convs) conv_body(s0); if (i < fronts) frontend(s1); }
            CHECK(hipDeviceSynchronize());
            auto t3 = std::chrono::high_resolution_clock::now();
            auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            printf("%-34s %d convs alone %.0f us, %d front-ends alone %.0f us, together on two streams %.0f us\n", what, convs,
                   us(t0, t1), fronts, us(t1, t2), us(t2, t3));
        };
        pair("whole layer (153.6 KB) + front-end", 30, 10, [&](hipStream_t queue) { OK(emph_conv1d_winograd4(x, ld, y, ld, dwhole, dbias, c, c, 1, dconv, n_conv, queue)); });
        pair("half 0 (92 KB) + front-end", 30, 10, [&](hipStream_t queue) { OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 0, queue)); });
        pair("half 1 (61 KB) + front-end", 30, 10, [&](hipStream_t queue) { OK(emph_conv1d_winograd4_half(x, ld, y, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 1, queue)); });
        CHECK(hipFree(mel));
        CHECK(hipFree(x)); CHECK(hipFree(y));
    }

    for (int mode = 0; mode < 2; ++mode) {           // 0: whole layers, 1: halves on two streams
        Lane lanes[2];
        for (Lane& lane : lanes) {
            CHECK(hipStreamCreate(&lane.main)); CHECK(hipStreamCreate(&lane.side));
            CHECK(hipEventCreateWithFlags(&lane.fork, hipEventDisableTiming));
            CHECK(hipEventCreateWithFlags(&lane.join_a, hipEventDisableTiming));
            CHECK(hipEventCreateWithFlags(&lane.join_b, hipEventDisableTiming));
            CHECK(hipMalloc(&lane.audio_out, 80 * ld * 4)); CHECK(hipMalloc(&lane.a, c * ld * 4)); CHECK(hipMalloc(&lane.b, c * ld * 4));
            auto enqueue = [&]() {
                if (with_frontend)
                    OK(emph_logmel(audio, 0, seg, dfe, n_fe, dtable, dstart, dcount, doffset, dvalues, (int)values.size(),
                                   lane.audio_out, ld, 0, -1, nullptr, nullptr, 0, lane.main));
                const float* in = lane.audio_out; float* out = lane.a;
                for (int layer = 0; layer < layers; ++layer) {
                    if (mode == 0) {
                        OK(emph_conv1d_winograd4(in, ld, out, ld, dwhole, dbias, c, c, 1, dconv, n_conv, lane.main));
                    } else {
                        // fork: the side stream takes half 1 once everything before is done
                        CHECK(hipEventRecord(lane.fork, lane.main));
                        CHECK(hipStreamWaitEvent(lane.side, lane.fork, 0));
                        OK(emph_conv1d_winograd4_half(in, ld, out, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 0, lane.main));
                        OK(emph_conv1d_winograd4_half(in, ld, out, ld, dsplit, dbias, c, c, 1, dconv, n_conv, nullptr, 0, 1, lane.side));
                        CHECK(hipEventRecord(lane.join_b, lane.side));
                        CHECK(hipStreamWaitEvent(lane.main, lane.join_b, 0));
                    }
                    in = out; out = (out == lane.a) ? lane.b : lane.a;
                }
            };
            enqueue();                                // first launches set the LDS attributes
            CHECK(hipDeviceSynchronize());
            hipGraph_t graph;
            CHECK(hipStreamBeginCapture(lane.main, hipStreamCaptureModeThreadLocal));
            enqueue();
            CHECK(hipStreamEndCapture(lane.main, &graph));
            CHECK(hipGraphInstantiate(&lane.graph, graph, nullptr, nullptr, 0));
        }
        for (int in_flight = 1; in_flight <= 2; ++in_flight) {
            for (int rep = 0; rep < 20; ++rep) CHECK(hipGraphLaunch(lanes[rep % in_flight].graph, lanes[rep % in_flight].main));
            CHECK(hipDeviceSynchronize());
            double best = 1e9;
            for (int trial = 0; trial < 5; ++trial) {
                const int reps = 200;
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                CHECK(hipDeviceSynchronize());
                auto t0 = std::chrono::high_resolution_clock::now();
                for (int rep = 0; rep < reps; ++rep) CHECK(hipGraphLaunch(lanes[rep % in_flight].graph, lanes[rep % in_flight].main));
                CHECK(hipDeviceSynchronize());
                auto t1 = std::chrono::high_resolution_clock::now();
                best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
            }
            printf("%s, %d batch(es) in flight: %.1f us per batch (%s + %d layers)\n", mode ? "two halves per layer" : "one launch per layer ",
                   in_flight, best, with_frontend ? "front-end" : "no front-end", layers);
        }
    }
    return 0;
}
