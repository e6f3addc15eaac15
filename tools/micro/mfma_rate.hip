// Sustained fp32 MFMA throughput of the whole chip by instruction shape: a bare stream of
// independent MFMAs from W waves per SIMD on every CU.  Is the 32x32x2 shape (twice the flops
// per operand read) cheaper in power, i.e. faster under the package limit, than 16x16x4?
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_rate.hip -o tools/micro/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// VARIED: eight different pseudo-random operand pairs per lane instead of one constant pair
// (the multipliers' inputs toggle from one MFMA to the next, as they do on real data)
template <int SHAPE, bool VARIED>
__global__ void stream_kernel(int rounds, const float* seed, float* out, unsigned long long* ticks) {
    float as[8], bs[8];
    for (int i = 0; i < 8; ++i) {
        as[i] = VARIED ? seed[(threadIdx.x * 8 + i) % 4096] : threadIdx.x * 1e-3f;
        bs[i] = VARIED ? seed[(threadIdx.x * 8 + i + 2048) % 4096] : 1.0001f;
    }
    const unsigned long long start = __builtin_amdgcn_s_memtime();
    float result = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rounds; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[i], bs[i], acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) result += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
        for (int r = 0; r < rounds; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[i], bs[i + 4], acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) result += acc[i][0] + acc[i][15];
    }
    const unsigned long long stop = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = result;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 16 + (threadIdx.x >> 6)] = stop - start;
}

template <int SHAPE, bool VARIED>
void run(int waves_per_simd, int rounds, const float* seed, float* out, unsigned long long* ticks) {
    const int threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) stream_kernel<SHAPE, VARIED><<<256, threads>>>(rounds, seed, out, ticks);
    CHECK(hipDeviceSynchronize());
    std::vector<float> laps;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        stream_kernel<SHAPE, VARIED><<<256, threads>>>(rounds, seed, out, ticks);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        laps.push_back(ms);
    }
    std::sort(laps.begin(), laps.end());
    // flops: 16x16x4 = 2048 per MFMA, 8 per round; 32x32x2 = 4096 per MFMA, 4 per round
    const double flops = 256. * 4 * waves_per_simd * rounds * 16384.;
    std::vector<unsigned long long> h(256 * 16);
    CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    printf("%dx%d %s operands, %d wave(s) per SIMD: %.1f us, %.1f TFLOP/s, %.2f ticks per us in wave 0\n", SHAPE, SHAPE,
           VARIED ? "varied" : "constant", waves_per_simd, laps[2] * 1e3, flops / (laps[2] * 1e-3) / 1e12, double(h[0]) / (laps[2] * 1e3));
}

int main() {
    float* out; unsigned long long* ticks;
    CHECK(hipMalloc(&out, 256 * 1024 * 4)); CHECK(hipMalloc(&ticks, 256 * 16 * 8));
    std::vector<float> host(4096);
    for (int i = 0; i < 4096; ++i) host[i] = (float)((i * 2654435761u) % 100003) / 50000.f - 1.f;
    float* seed;
    CHECK(hipMalloc(&seed, 4096 * 4));
    CHECK(hipMemcpy(seed, host.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int waves : {1, 2}) {
        run<16, false>(waves, 20000 / waves, seed, out, ticks);
        run<16, true>(waves, 20000 / waves, seed, out, ticks);
        run<32, false>(waves, 20000 / waves, seed, out, ticks);
        run<32, true>(waves, 20000 / waves, seed, out, ticks);
    }
    return 0;
}
