// The K loop of conv1d_stack_kernel by itself: eight MFMA waves (two per SIMD, 3 + 2 m-tiles)
// run `chunks` chunks of four k-steps over LDS-resident data - no weight stream, no loader
// waves - to see what the loop costs without the rest of the kernel.
//   -DBARRIER      a workgroup barrier per chunk (as in the kernel)
//   -DNO_TRANSFORM v = d (no vector work)
//   -DPACKED       the input transform as six v_pk_fma_f32
//   -DNO_A_READS   the A fragments stay in registers (no LDS reads but the six B floats)
//   -DONE_WAVE     one wave per SIMD owning all five m-tiles (four waves)
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/micro/kloop.hip -o tools/micro/bin/kloop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef KLOOP_CHUNK_STEPS
#define KLOOP_CHUNK_STEPS 4
#endif
constexpr int kStride = 260, kChannels = 80, kMTiles = 5, kChunkSteps = KLOOP_CHUNK_STEPS;
constexpr int kStepFloats = 6 * kMTiles * 64;
constexpr int kChunkFloats = kChunkSteps * kStepFloats;

template <int COUNT>
__device__ __forceinline__ void loop(const float* act, const float* ring, int chunks, int m_begin,
                                     int tile, float* out, unsigned long long* clocks) {
    const int lane = threadIdx.x & 63;
    const int kk = lane >> 4, col = lane & 15;
    const float* lane_rows = act + kk * kStride + 64 * tile + 4 * col;
    f32x4 acc[6][COUNT];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int m = 0; m < COUNT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[6][COUNT], d[6], v[6];
    auto load_b = [&](int step) {
        const float* source = lane_rows + 4 * step * kStride;
        const f32x4 first = *reinterpret_cast<const f32x4*>(source);
        const f32x2 second = *reinterpret_cast<const f32x2*>(source + 4);
        d[0] = first[0], d[1] = first[1], d[2] = first[2], d[3] = first[3];
        d[4] = second[0], d[5] = second[1];
    };
    load_b(0);
#ifdef NO_A_READS
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int m = 0; m < COUNT; ++m) a[j][m] = ring[((j * kMTiles + m) << 6) + lane];
#endif
    const unsigned long long start = __builtin_amdgcn_s_memtime();
    for (int chunk = 0; chunk < chunks; ++chunk) {
#ifdef BARRIER
        __syncthreads();
#endif
        const int within = chunk % (20 / kChunkSteps);
        if (within == 0) load_b(0);
        const float* weights = ring + (chunk & 1) * kChunkFloats + (m_begin << 6) + lane;
#pragma unroll
        for (int ks = 0; ks < kChunkSteps; ++ks) {
            const int step = within * kChunkSteps + ks;
#ifndef NO_A_READS
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int m = 0; m < COUNT; ++m)
                    a[j][m] = weights[(ks * 6 * kMTiles + j * kMTiles + m) << 6];
#endif
#ifdef NO_TRANSFORM
            for (int j = 0; j < 6; ++j) v[j] = d[j];
#elif defined(PACKED)
            // the same operations on register pairs: {p, c}, {q, e/2}, {v1, v3}, {v2, v4},
            // {t0, t5}, {v0, v5} - six v_pk_fma_f32 for fourteen scalar instructions
            {
                const f32x2 p0 = {d[0], d[1]}, p1 = {d[2], d[3]}, p2 = {d[4], d[5]};
                const f32x2 m41 = {-4.f, -1.f}, k12 = {1.f, 2.f}, n12 = {-1.f, -2.f};
                const f32x2 m55 = {-5.f, -5.f}, k44 = {4.f, 4.f};
                f32x2 pc, qe, v13, v24, t05, v05;
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]"
                    : "=v"(pc) : "v"(m41), "v"(p1), "v"(p2));           // {d4 - 4 d2, d4 - d2}
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]"
                    : "=v"(qe) : "v"(m41), "v"(p0), "v"(p1));           // {d3 - 4 d1, d3 - d1}
                asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v13) : "v"(k12), "v"(qe), "v"(pc));
                asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v24) : "v"(n12), "v"(qe), "v"(pc));
                asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t05) : "v"(m55), "v"(p1), "v"(p2));
                asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v05) : "v"(k44), "v"(p0), "v"(t05));
                v[0] = v05[0], v[5] = v05[1];
                v[1] = v13[0], v[3] = v13[1];
                v[2] = v24[0], v[4] = v24[1];
            }
#else
            const float p = fmaf(-4.f, d[2], d[4]);
            const float q = fmaf(-4.f, d[1], d[3]);
            const float c = d[4] - d[2];
            const float e = 2.f * (d[3] - d[1]);
            v[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
            v[1] = p + q;
            v[2] = p - q;
            v[3] = c + e;
            v[4] = c - e;
            v[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
#endif
            if (step + 1 < 20) load_b(step + 1);
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int m = 0; m < COUNT; ++m)
                    acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][m], v[j], acc[j][m], 0, 0, 0);
        }
    }
    const unsigned long long stop = __builtin_amdgcn_s_memtime();
    f32x4 total = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int m = 0; m < COUNT; ++m) total += acc[j][m];
    out[blockIdx.x * blockDim.x + threadIdx.x] = total[0] + total[1] + total[2] + total[3];
    if (lane == 0) clocks[blockIdx.x * 16 + (threadIdx.x >> 6)] = stop - start;
}

__global__ __launch_bounds__(768) void kloop_kernel(const float* seed, int chunks, float* out,
                                                    unsigned long long* clocks) {
    extern __shared__ __align__(16) float lds[];
    float* act = lds;
    float* ring = act + kChannels * kStride;
    for (int i = threadIdx.x; i < kChannels * kStride + 2 * kChunkFloats; i += blockDim.x)
        lds[i] = seed[i % 4096];
    __syncthreads();
    const int wave = threadIdx.x >> 6;
#ifdef ONE_WAVE
    loop<5>(act, ring, chunks, 0, wave & 3, out, clocks);
#else
    if (wave >= 8) {
#ifdef BARRIER
        for (int chunk = 0; chunk < chunks; ++chunk) __syncthreads();
#endif
        return;
    }
    if (wave >> 2) loop<2>(act, ring, chunks, 3, wave & 3, out, clocks);
    else loop<3>(act, ring, chunks, 0, wave & 3, out, clocks);
#endif
}

#ifndef KLOOP_NO_MAIN
int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 150;
#ifdef ONE_WAVE
    const int threads = 256;
#elif defined(LOADERS)
    const int threads = 768;
#else
    const int threads = 512;
#endif
    std::vector<float> seed(4096);
    for (int i = 0; i < 4096; ++i) seed[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    float *d_seed, *out;
    unsigned long long* clocks;
    CHECK(hipMalloc(&d_seed, 4096 * 4)); CHECK(hipMalloc(&out, 256 * 768 * 4)); CHECK(hipMalloc(&clocks, 256 * 16 * 8));
    CHECK(hipMemcpy(d_seed, seed.data(), 4096 * 4, hipMemcpyHostToDevice));
    const size_t lds = (kChannels * kStride + 2 * kChunkFloats) * 4;
    CHECK(hipFuncSetAttribute((const void*)kloop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) kloop_kernel<<<256, threads, lds>>>(d_seed, chunks, out, clocks);
    CHECK(hipDeviceSynchronize());
    std::vector<float> laps;
    for (int rep = 0; rep < 7; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        kloop_kernel<<<256, threads, lds>>>(d_seed, chunks, out, clocks);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        laps.push_back(ms * 1e3f);
    }
    std::sort(laps.begin(), laps.end());
    std::vector<unsigned long long> h(256 * 16);
    CHECK(hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost));
    // the workgroup's LONGEST wave (waves of a SIMD do not progress evenly without a barrier:
    // the mean over the waves is not the loop's duration), mean over the workgroups
    double cycles = 0, mean = 0;
    const int waves = threads >= 512 ? 8 : 4;
    for (int b = 0; b < 256; ++b) {
        unsigned long long longest = 0;
        for (int w = 0; w < waves; ++w) longest = std::max(longest, h[b * 16 + w]), mean += h[b * 16 + w];
        cycles += longest;
    }
    cycles /= 256, mean /= 256 * waves;
    // per chunk and SIMD: 120 MFMAs of 32 cycles = 3840 cycles of matrix pipe
    printf("%d chunks: kernel %.1f us = %.3f us per chunk; %.0f cycles per chunk in the longest wave (mean of the "
           "waves %.0f) = %.2f GHz; matrix pipe %.0f %% of them\n", chunks, laps[3], laps[3] / chunks, cycles / chunks,
           mean / chunks, cycles / (laps[3] * 1e3), 100. * 3840 * chunks / cycles);
    return 0;
}
#endif
