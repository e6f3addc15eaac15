// What lowers the clock of an MFMA loop?  A stream of independent 16x16x4 fp32 MFMAs on every
// SIMD (two waves per SIMD), plus, by variant: a large LDS allocation that is never touched, LDS
// reads of the operands (one per MFMA or one per four), vector instructions between the MFMAs.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/clock_probe.hip -o tools/micro/bin/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// READS: LDS reads per 8 MFMAs (0, 2, 8); VALU: dependent fmas per 8 MFMAs (0, 8)
template <int READS, int VALU>
__global__ __launch_bounds__(512) void probe_kernel(int rounds, const float* seed, float* out,
                                                    unsigned long long* ticks) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = seed[i % 4096];
    __syncthreads();
    float as[8], bs[8];
    for (int i = 0; i < 8; ++i) as[i] = seed[(threadIdx.x * 8 + i) % 4096], bs[i] = seed[(threadIdx.x + 64 * i) % 4096];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float chain = as[0];
    const float* mine = lds + (threadIdx.x & 63);
    const unsigned long long start = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        const float* row = mine + 64 * ((r & 7) * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (READS == 8 || (READS == 2 && (i & 3) == 0)) bs[i] = row[64 * i];
            if (VALU && i < VALU) chain = fmaf(chain, 0.999f, bs[i]);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[i], VALU ? bs[i] + 0.f * chain : bs[i], acc[i], 0, 0, 0);
        }
    }
    const unsigned long long stop = __builtin_amdgcn_s_memtime();
    float result = chain;
    for (int i = 0; i < 8; ++i) result += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = result;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 16 + (threadIdx.x >> 6)] = stop - start;
}

template <int READS, int VALU>
void run(const char* name, size_t lds_bytes, int rounds, const float* seed, float* out, unsigned long long* ticks) {
    CHECK(hipFuncSetAttribute((const void*)probe_kernel<READS, VALU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) probe_kernel<READS, VALU><<<256, 512, lds_bytes>>>(rounds, seed, out, ticks);
    CHECK(hipDeviceSynchronize());
    std::vector<float> laps;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        probe_kernel<READS, VALU><<<256, 512, lds_bytes>>>(rounds, seed, out, ticks);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        laps.push_back(ms);
    }
    std::sort(laps.begin(), laps.end());
    std::vector<unsigned long long> h(256 * 16);
    CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    double longest = 0;
    for (int b = 0; b < 256; ++b) {
        unsigned long long most = 0;
        for (int w = 0; w < 8; ++w) most = std::max(most, h[b * 16 + w]);
        longest += most;
    }
    longest /= 256;
    const double mfmas = 2. * rounds * 8;                // per SIMD
    printf("%-44s %8.1f us  %5.0f MHz  %5.1f cycles per MFMA  %6.1f TFLOP/s\n", name, laps[2] * 1e3,
           longest / (laps[2] * 1e3), longest / mfmas, 256. * 4 * mfmas * 2048 / (laps[2] * 1e-3) / 1e12);
}

int main() {
    float *out, *seed; unsigned long long* ticks;
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&ticks, 256 * 16 * 8)); CHECK(hipMalloc(&seed, 4096 * 4));
    std::vector<float> host(4096);
    for (int i = 0; i < 4096; ++i) host[i] = (float)((i * 2654435761u) % 100003) / 50000.f - 1.f;
    CHECK(hipMemcpy(seed, host.data(), 4096 * 4, hipMemcpyHostToDevice));
    const int rounds = 10000;
    run<0, 0>("MFMAs only, 32 KB of LDS", 32768, rounds, seed, out, ticks);
    run<0, 0>("MFMAs only, 144 KB of LDS allocated", 147456, rounds, seed, out, ticks);
    run<2, 0>("an LDS read per four MFMAs", 32768, rounds, seed, out, ticks);
    run<8, 0>("an LDS read per MFMA", 32768, rounds, seed, out, ticks);
    run<0, 8>("a dependent fma per MFMA", 32768, rounds, seed, out, ticks);
    run<8, 8>("an LDS read and an fma per MFMA", 32768, rounds, seed, out, ticks);
    run<8, 8>("the same, 144 KB of LDS allocated", 147456, rounds, seed, out, ticks);
    return 0;
}
