// What hides under an fp32 MFMA on gfx950?  (round-4 review, weak #3: tools/micro/coexec.hip only
// ever put v_pk_fma_f32 beside the MFMAs - the one class MI355X_MICROARCH.md flags as NOT hiding.)
//
// One instruction stream per wave: V independent FILLER instructions behind every MFMA, for
//   matrix   0 v_mfma_f32_16x16x4_f32   1 v_mfma_f32_32x32x2_f32
//            2 v_mfma_f32_16x16x32_bf16 3 v_mfma_f32_32x32x16_bf16      (the guide's controls)
//   filler   0 v_fma_f32  1 v_add_f32  2 v_exp_f32  3 ds_read_b64  4 ds_read_b128
//            5 v_pk_fma_f32 (round 4's filler, as the control)  6 ds_read_b32
//            7 the conv K loop's own mix per MFMA: 1 ds_read2st64_b32 per two MFMAs + 4 VALU per
//              six MFMAs (V scales it: V = 3 is the loop's density x 3)
//   V        1 .. 6, on one and on two waves per SIMD
// and the PARTNERS experiment with scalar instructions: waves 0-3 MFMAs only, waves 4-7
// v_fma_f32 only (they share a SIMD pairwise), sized to last equally long alone.
//
// Output: cycles per MFMA (s_memtime of the workgroup's longest wave, mean over the workgroups) per row.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/micro/fillers.hip -o tools/micro/bin/fillers
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int FILLER>
__device__ __forceinline__ void filler(float& s, f2& p, f4& wide, unsigned address) {
    if (FILLER == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s) : "v"(1.0001f), "v"(1e-4f));
    if (FILLER == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(1e-4f));
    if (FILLER == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(s));
    if (FILLER == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(p) : "v"(address));
    if (FILLER == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(wide) : "v"(address));
    if (FILLER == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(f2{1.0001f, 0.9999f}), "v"(f2{1e-4f, -1e-4f}));
    if (FILLER == 6) asm volatile("ds_read_b32 %0, %1" : "=v"(s) : "v"(address));
}

template <int MATRIX, int FILLER, int V>
__global__ __launch_bounds__(512) void stream_kernel(float* out, unsigned long long* cycles, int iterations) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-3f;
    f4 small[8];
    f16 big[4];
    for (int i = 0; i < 8; ++i) small[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    float s[8];
    f2 p[8];
    f4 wide[4];
    for (int i = 0; i < 8; ++i) s[i] = threadIdx.x * 1e-3f + i, p[i] = f2{threadIdx.x * 1e-3f + i, 1.f - i};
    for (int i = 0; i < 4; ++i) wide[i] = f4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-4f, b = 1.f - threadIdx.x * 1e-4f;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) ah[i] = __bf16(1.f + threadIdx.x * 1e-3f), bh[i] = __bf16(0.5f + i);
    // conflict-free LDS addresses: 8 B (b32 / b64) or 16 B (b128) per lane
    const unsigned address = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds +
                             (threadIdx.x & 63) * (FILLER == 4 ? 16 : 8);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MATRIX == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(small[i]) : "v"(a), "v"(b));
            if (MATRIX == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(a), "v"(b));
            if (MATRIX == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(small[i]) : "v"(ah), "v"(bh));
            if (MATRIX == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(ah), "v"(bh));
            if (FILLER == 7) {
                // the K loop's density x V: per MFMA V/2 two-dword LDS reads and 2V/3 scalar VALU
#pragma unroll
                for (int j = 0; j < ((i + 1) * V) / 2 - (i * V) / 2; ++j)
                    asm volatile("ds_read2st64_b32 %0, %1 offset1:1" : "=v"(p[(i + j) & 7]) : "v"(address));
#pragma unroll
                for (int j = 0; j < ((i + 1) * 2 * V) / 3 - (i * 2 * V) / 3; ++j)
                    filler<(0)>(s[(i + j) & 7], p[0], wide[0], address);
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    filler<FILLER>(s[(i * V + j) & 7], p[(i * V + j) & 7], wide[(i * V + j) & 3], address);
            }
        }
        if (FILLER == 3 || FILLER == 4 || FILLER == 6 || FILLER == 7) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += small[i].x + small[i].w + s[i] + p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) sum += big[i][0] + big[i][15] + wide[i].x + wide[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

// partners: waves 0-3 one kind of MFMA only, waves 4-7 one kind of scalar filler only
template <int MATRIX, int FILLER>
__global__ __launch_bounds__(512) void partner_kernel(float* out, unsigned long long* cycles, int matrix_iterations,
                                                      int filler_iterations, int mode) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-3f;
    const int role = mode == 0 ? 0 : mode == 1 ? 1 : wave >= 4;
    f4 small[8];
    f16 big[4];
    for (int i = 0; i < 8; ++i) small[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    float s[8];
    f2 p[8];
    f4 wide[4];
    for (int i = 0; i < 8; ++i) s[i] = threadIdx.x * 1e-3f + i, p[i] = f2{threadIdx.x * 1e-3f + i, 1.f - i};
    for (int i = 0; i < 4; ++i) wide[i] = f4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-4f, b = 1.f - threadIdx.x * 1e-4f;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) ah[i] = __bf16(1.f + threadIdx.x * 1e-3f), bh[i] = __bf16(0.5f + i);
    const unsigned address = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds +
                             (threadIdx.x & 63) * (FILLER == 4 ? 16 : 8);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) {
        for (int it = 0; it < matrix_iterations; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MATRIX == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(small[i]) : "v"(a), "v"(b));
                if (MATRIX == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(a), "v"(b));
                if (MATRIX == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(small[i]) : "v"(ah), "v"(bh));
                if (MATRIX == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(ah), "v"(bh));
            }
    } else {
        for (int it = 0; it < filler_iterations; ++it) {
#pragma unroll
            for (int j = 0; j < 32; ++j) filler<FILLER>(s[j & 7], p[j & 7], wide[j & 3], address);
            if (FILLER == 3 || FILLER == 4 || FILLER == 6) asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += small[i].x + small[i].w + s[i] + p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) sum += big[i][0] + big[i][15] + wide[i].x + wide[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

// blocks: B fp32 MFMAs, then G scalar fmas (the shape of a K-loop step with its input transform)
template <int B, int G>
__global__ __launch_bounds__(512) void phased_kernel(float* out, unsigned long long* cycles, int iterations) {
    const int wave = threadIdx.x >> 6;
    f4 small[8];
    float s[8];
    for (int i = 0; i < 8; ++i) small[i] = f4{0.f, 0.f, 0.f, 0.f}, s[i] = threadIdx.x * 1e-3f + i;
    const float a = threadIdx.x * 1e-4f, b = 1.f - threadIdx.x * 1e-4f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int i = 0; i < B; ++i)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(small[i & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < G; ++j)
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[j & 7]) : "v"(1.0001f), "v"(1e-4f));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += small[i].x + small[i].w + s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

static float* out;
static unsigned long long* cycles;
static unsigned long long host[256 * 8];
static const char* matrix_name[4] = {"f32 16x16x4", "f32 32x32x2", "bf16 16x16x32", "bf16 32x32x16"};
static const char* filler_name[8] = {"v_fma_f32", "v_add_f32", "v_exp_f32", "ds_read_b64", "ds_read_b128",
                                     "v_pk_fma_f32", "ds_read_b32", "conv K-loop mix"};

// cycles of the LONGEST of waves first .. first + waves - 1, mean over the workgroups (two waves of
// a SIMD are not served evenly - the older one runs ahead - so the mean of the waves is not the
// duration of the stream; wave w and w + 4 share a SIMD)
template <typename Launch>
static double timed(Launch launch, int first, int waves, float* us = nullptr) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (us) *us = ms * 1e3f;
    CHECK(hipMemcpy(host, cycles, sizeof(host), hipMemcpyDeviceToHost));
    double total = 0;
    for (int block = 0; block < 256; ++block) {
        unsigned long long longest = 0;
        for (int w = first; w < first + waves; ++w) longest = host[block * 8 + w] > longest ? host[block * 8 + w] : longest;
        total += double(longest);
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return total / 256.;
}

constexpr int kIterations = 2000;          // 16 000 MFMAs per wave

template <int MATRIX, int FILLER, int V>
static void row(double* result) {
    for (int threads = 256; threads <= 512; threads += 256) {
        float us;
        const double c = timed([&] { hipLaunchKernelGGL((stream_kernel<MATRIX, FILLER, V>), dim3(256), dim3(threads), 0, 0, out, cycles, kIterations); },
                               0, threads / 64, &us);
        result[threads / 256 - 1] = c / (kIterations * 8.);
    }
}

template <int MATRIX, int FILLER>
static void block() {
    double r[7][2];
    row<MATRIX, FILLER, 0>(r[0]);
    row<MATRIX, FILLER, 1>(r[1]);
    row<MATRIX, FILLER, 2>(r[2]);
    row<MATRIX, FILLER, 3>(r[3]);
    row<MATRIX, FILLER, 4>(r[4]);
    row<MATRIX, FILLER, 5>(r[5]);
    row<MATRIX, FILLER, 6>(r[6]);
    for (int waves = 1; waves <= 2; ++waves) {
        printf("%-14s + V x %-15s %d wave%s/SIMD: cycles per MFMA%s at V = 0..6:", matrix_name[MATRIX], filler_name[FILLER],
               waves, waves == 1 ? " " : "s", waves == 2 ? " and wave" : "");
        for (int v = 0; v <= 6; ++v) printf(" %6.2f", r[v][waves - 1]);
        printf("   (+ per filler at V = 1, 3, 6: %+.2f %+.2f %+.2f)\n", r[1][waves - 1] - r[0][waves - 1],
               (r[3][waves - 1] - r[0][waves - 1]) / 3, (r[6][waves - 1] - r[0][waves - 1]) / 6);
    }
}

template <int MATRIX, int FILLER>
static void partners() {
    const int mi = kIterations, fi = kIterations;
    const double matrix_alone = timed([&] { hipLaunchKernelGGL((partner_kernel<MATRIX, FILLER>), dim3(256), dim3(256), 0, 0, out, cycles, mi, fi, 0); }, 0, 4);
    const double filler_alone = timed([&] { hipLaunchKernelGGL((partner_kernel<MATRIX, FILLER>), dim3(256), dim3(256), 0, 0, out, cycles, mi, fi, 1); }, 0, 4);
    const int matched = int(fi * matrix_alone / filler_alone);
    float us;
    const double m = timed([&] { hipLaunchKernelGGL((partner_kernel<MATRIX, FILLER>), dim3(256), dim3(512), 0, 0, out, cycles, mi, matched, 2); }, 0, 4, &us);
    double f = 0;
    for (int block = 0; block < 256; ++block) {
        unsigned long long longest = 0;
        for (int w = 4; w < 8; ++w) longest = host[block * 8 + w] > longest ? host[block * 8 + w] : longest;
        f += double(longest);
    }
    f /= 256.;
    const double f_alone = filler_alone * matched / fi;
    printf("partners %-14s | %-13s: matrix waves x%.2f of alone (%.2f cycles per MFMA), filler waves x%.2f of alone "
           "(%.2f cycles per %s alone, %.2f beside the matrix wave)   [no overlap: the filler waves run after the matrix waves, x2.00; full overlap: x1.00]\n",
           matrix_name[MATRIX], filler_name[FILLER], m / matrix_alone, m / (mi * 8.), f / f_alone,
           filler_alone / (fi * 32.), filler_name[FILLER], f / (matched * 32.));
}

template <int B, int G>
static void phased() {
    for (int threads = 256; threads <= 512; threads += 256) {
        const int iterations = 16000 / B;
        const double c = timed([&] { hipLaunchKernelGGL((phased_kernel<B, G>), dim3(256), dim3(threads), 0, 0, out, cycles, iterations); },
                               0, threads / 64);
        const double per_block = c / iterations - 32. * B * (threads / 256);
        printf("phased f32 16x16x4: %2d MFMAs then %2d v_fma_f32, %d wave%s/SIMD: %7.1f cycles per block = the MFMAs' %d + %.1f"
               " (%.2f per fma%s)\n", B, G, threads / 256, threads == 256 ? " " : "s", c / iterations, 32 * B * (threads / 256),
               per_block, per_block / (G * (threads / 256)), threads == 512 ? " of either wave" : "");
    }
}

int main() {
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cycles, sizeof(host)));
    phased<6, 1>(); phased<6, 4>(); phased<6, 12>(); phased<18, 12>(); phased<18, 24>(); phased<36, 24>(); phased<72, 48>();
    block<0, 0>(); block<0, 1>(); block<0, 2>(); block<0, 3>(); block<0, 4>(); block<0, 6>(); block<0, 5>(); block<0, 7>();
    block<1, 0>(); block<1, 1>(); block<1, 2>(); block<1, 3>(); block<1, 4>(); block<1, 6>(); block<1, 5>();
    block<2, 0>(); block<2, 2>(); block<2, 4>(); block<2, 5>();
    block<3, 0>(); block<3, 2>(); block<3, 4>(); block<3, 5>();
    partners<0, 0>(); partners<0, 1>(); partners<0, 2>(); partners<0, 3>(); partners<0, 5>();
    partners<1, 0>(); partners<1, 3>();
    partners<2, 0>(); partners<2, 5>();
    partners<3, 0>(); partners<3, 5>();
    return 0;
}
