// Do fp32 MFMAs and vector instructions of ONE SIMD overlap on gfx950?
// (round-3 DESIGN inferred "no" from SQ_VALU_MFMA_COEXEC_CYCLES = 0 in the conv and
// attention kernels; this measures it.)
//
//   mfma_only      1 wave per SIMD, v_mfma_f32_16x16x4_f32 on 8 independent accumulators
//   valu_only      1 wave per SIMD, v_pk_fma_f32 on 8 independent register pairs
//   partners       2 waves per SIMD: waves 0-3 matrix only, waves 4-7 vector only
//                  (wave w runs on SIMD w % 4, so w and w + 4 are partners)
//   neighbours     2 waves per SIMD: SIMDs 0, 2 matrix only (both waves), SIMDs 1, 3 vector only
//   interleaved<V> 1 wave per SIMD, ONE instruction stream: V vector instructions behind every MFMA
//   interleaved2<V> the same stream on 2 waves per SIMD
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/coexec.hip -o tools/micro/bin/coexec
// Counters: rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- tools/micro/bin/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void matrix_block(f4 (&acc)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
}

__device__ __forceinline__ void vector_block(f2 (&v)[8], f2 k, f2 c) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(c));
}

// role of a wave: 0 = matrix only (8 MFMAs per iteration), 1 = vector only (32 v_pk_fma per iteration)
template <int MODE>
__global__ __launch_bounds__(512) void roles_kernel(float* out, unsigned long long* cycles, int matrix_iterations,
                                                    int vector_iterations) {
    const int wave = threadIdx.x >> 6;
    int role;
    if (MODE == 0) role = 0;
    else if (MODE == 1) role = 1;
    else if (MODE == 2) role = wave >= 4;
    else role = wave & 1;
    f4 acc[8];
    f2 v[8];
    for (int i = 0; i < 8; ++i) {
        acc[i] = {0.f, 0.f, 0.f, 0.f};
        v[i] = {threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    }
    const float a = threadIdx.x * 1e-4f, b = 1.f - threadIdx.x * 1e-4f;
    const f2 k = {1.0001f, 0.9999f}, c = {1e-4f, -1e-4f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0)
        for (int it = 0; it < matrix_iterations; ++it) matrix_block(acc, a, b);
    else
        for (int it = 0; it < vector_iterations; ++it) vector_block(v, k, c);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += acc[i].x + acc[i].y + acc[i].z + acc[i].w + v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

// control: the same partner experiment with a bf16 MFMA (v_mfma_f32_16x16x32_bf16: its own
// multipliers) as the matrix role - if THIS pair overlaps, the harness sees overlap when there is one
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void bf16_partners_kernel(float* out, unsigned long long* cycles, int matrix_iterations,
                                                            int vector_iterations, int mode) {
    const int wave = threadIdx.x >> 6;
    const int role = mode == 0 ? 0 : mode == 1 ? 1 : wave >= 4;
    f4 acc[8];
    f2 v[8];
    for (int i = 0; i < 8; ++i) {
        acc[i] = {0.f, 0.f, 0.f, 0.f};
        v[i] = {threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = __bf16(1.f + threadIdx.x * 1e-3f); b[i] = __bf16(0.5f + i); }
    const f2 k = {1.0001f, 0.9999f}, c = {1e-4f, -1e-4f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0)
        for (int it = 0; it < matrix_iterations; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
    else
        for (int it = 0; it < vector_iterations; ++it) vector_block(v, k, c);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += acc[i].x + acc[i].y + acc[i].z + acc[i].w + v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

// one stream: D ds_read_b64 (D > 0) or -D global_load_dwordx2 behind every fp32 MFMA
template <int D>
__global__ __launch_bounds__(512) void memory_interleaved_kernel(float* out, unsigned long long* cycles, int iterations,
                                                                 const float* source) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    f4 acc[8];
    f2 v[8];
    for (int i = 0; i < 8; ++i) {
        acc[i] = {0.f, 0.f, 0.f, 0.f};
        v[i] = {0.f, 0.f};
    }
    const float a = threadIdx.x * 1e-4f, b = 1.f - threadIdx.x * 1e-4f;
    const unsigned address = (threadIdx.x & 511) * 8;
    const float* pointer = source + (threadIdx.x & 511) * 2;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < (D > 0 ? D : -D); ++j) {
                if (D > 0) asm volatile("ds_read_b64 %0, %1" : "=v"(v[(i + j) & 7]) : "v"(address));
                else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v[(i + j) & 7]) : "v"(pointer));
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += acc[i].x + acc[i].y + acc[i].z + acc[i].w + v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

// one stream: V vector instructions behind every MFMA (all independent of one another)
template <int V>
__global__ __launch_bounds__(512) void interleaved_kernel(float* out, unsigned long long* cycles, int iterations,
                                                          int delay_second_half) {
    const int wave = threadIdx.x >> 6;
    f4 acc[8];
    f2 v[8];
    for (int i = 0; i < 8; ++i) {
        acc[i] = {0.f, 0.f, 0.f, 0.f};
        v[i] = {threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    }
    const float a = threadIdx.x * 1e-4f, b = 1.f - threadIdx.x * 1e-4f;
    const f2 k = {1.0001f, 0.9999f}, c = {1e-4f, -1e-4f};
    __syncthreads();
    if (delay_second_half && wave >= 4) __builtin_amdgcn_s_sleep(8);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < V; ++j)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * V + j) & 7]) : "v"(k), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += acc[i].x + acc[i].y + acc[i].z + acc[i].w + v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

// blocks of B MFMAs then B*V vector instructions (what a K loop with a transform phase looks like)
template <int V, int B>
__global__ __launch_bounds__(512) void phased_kernel(float* out, unsigned long long* cycles, int iterations, int stagger) {
    const int wave = threadIdx.x >> 6;
    f4 acc[8];
    f2 v[8];
    for (int i = 0; i < 8; ++i) {
        acc[i] = {0.f, 0.f, 0.f, 0.f};
        v[i] = {threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    }
    const float a = threadIdx.x * 1e-4f, b = 1.f - threadIdx.x * 1e-4f;
    const f2 k = {1.0001f, 0.9999f}, c = {1e-4f, -1e-4f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // stagger: waves 4-7 start with the vector phase, so a SIMD's partners are half a block apart
    if (stagger && wave >= 4) {
#pragma unroll
        for (int j = 0; j < B * V; ++j)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(k), "v"(c));
    }
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int i = 0; i < B; ++i)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < B * V; ++j)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(k), "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += acc[i].x + acc[i].y + acc[i].z + acc[i].w + v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

static float* out;
static unsigned long long* cycles;
static unsigned long long host[256 * 8];

template <typename Launch>
static void run(const char* name, int waves, Launch launch, double* per_wave) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(host, cycles, sizeof(host), hipMemcpyDeviceToHost));
    for (int w = 0; w < 8; ++w) per_wave[w] = 0;
    for (int block = 0; block < 256; ++block)
        for (int w = 0; w < waves; ++w) per_wave[w] += double(host[block * 8 + w]) / 256;
    printf("%-34s %8.1f us |", name, ms * 1e3);
}

int main() {
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cycles, sizeof(host)));
    const int mi = 4000, vi = 4000;    // 32 000 MFMAs; 128 000 packed fmas per wave
    double w[8];
    double matrix_alone, vector_alone;

    run("mfma_only (1 wave/SIMD)", 4, [&] { hipLaunchKernelGGL(roles_kernel<0>, dim3(256), dim3(256), 0, 0, out, cycles, mi, vi); }, w);
    matrix_alone = (w[0] + w[1] + w[2] + w[3]) / 4;
    printf(" %.2f cycles per MFMA\n", matrix_alone / (mi * 8.));
    run("valu_only (1 wave/SIMD)", 4, [&] { hipLaunchKernelGGL(roles_kernel<1>, dim3(256), dim3(256), 0, 0, out, cycles, mi, vi); }, w);
    vector_alone = (w[0] + w[1] + w[2] + w[3]) / 4;
    printf(" %.2f cycles per v_pk_fma_f32\n", vector_alone / (vi * 32.));
    run("mfma_only (2 waves/SIMD)", 8, [&] { hipLaunchKernelGGL(roles_kernel<0>, dim3(256), dim3(512), 0, 0, out, cycles, mi, vi); }, w);
    printf(" %.2f cycles per MFMA and wave\n", (w[0] + w[4]) / 2 / (mi * 8.));
    run("valu_only (2 waves/SIMD)", 8, [&] { hipLaunchKernelGGL(roles_kernel<1>, dim3(256), dim3(512), 0, 0, out, cycles, mi, vi); }, w);
    printf(" %.2f cycles per v_pk_fma_f32 and wave\n", (w[0] + w[4]) / 2 / (vi * 32.));

    // partners on one SIMD: vector work sized to last as long as the matrix work does alone
    const int vi_matched = int(vi * matrix_alone / vector_alone);
    run("partners (MFMA w0-3 | VALU w4-7)", 8, [&] { hipLaunchKernelGGL(roles_kernel<2>, dim3(256), dim3(512), 0, 0, out, cycles, mi, vi_matched); }, w);
    {
        const double m = (w[0] + w[1] + w[2] + w[3]) / 4, v = (w[4] + w[5] + w[6] + w[7]) / 4;
        const double v_alone = vector_alone * vi_matched / vi;
        printf(" matrix waves %.0f cycles (alone %.0f: x%.2f), vector waves %.0f (alone %.0f: x%.2f)  [sum would be x2.00]\n",
               m, matrix_alone, m / matrix_alone, v, v_alone, v / v_alone);
    }
    run("neighbours (MFMA SIMD0,2 | VALU 1,3)", 8, [&] { hipLaunchKernelGGL(roles_kernel<3>, dim3(256), dim3(512), 0, 0, out, cycles, mi, vi_matched); }, w);
    printf(" matrix waves %.0f cycles, vector waves %.0f\n", (w[0] + w[2] + w[4] + w[6]) / 4, (w[1] + w[3] + w[5] + w[7]) / 4);

#define INTER(V, THREADS, DELAY, label) \
    run(label, THREADS / 64, [&] { hipLaunchKernelGGL(interleaved_kernel<V>, dim3(256), dim3(THREADS), 0, 0, out, cycles, mi, DELAY); }, w); \
    printf(" %.2f cycles per MFMA (+%d v_pk_fma each)\n", (THREADS == 256 ? (w[0] + w[1] + w[2] + w[3]) / 4 : (w[0] + w[4]) / 2) / (mi * 8.), V);
    INTER(0, 256, 0, "interleaved V=0 (1 wave/SIMD)")
    INTER(1, 256, 0, "interleaved V=1 (1 wave/SIMD)")
    INTER(2, 256, 0, "interleaved V=2 (1 wave/SIMD)")
    INTER(3, 256, 0, "interleaved V=3 (1 wave/SIMD)")
    INTER(4, 256, 0, "interleaved V=4 (1 wave/SIMD)")
    INTER(6, 256, 0, "interleaved V=6 (1 wave/SIMD)")
    INTER(8, 256, 0, "interleaved V=8 (1 wave/SIMD)")
    INTER(0, 512, 0, "interleaved V=0 (2 waves/SIMD)")
    INTER(1, 512, 0, "interleaved V=1 (2 waves/SIMD)")
    INTER(2, 512, 0, "interleaved V=2 (2 waves/SIMD)")
    INTER(4, 512, 0, "interleaved V=4 (2 waves/SIMD)")

#define PHASED(V, B, THREADS, STAGGER, label) \
    run(label, THREADS / 64, [&] { hipLaunchKernelGGL((phased_kernel<V, B>), dim3(256), dim3(THREADS), 0, 0, out, cycles, mi * 8 / B, STAGGER); }, w); \
    printf(" %.2f cycles per MFMA and wave (blocks of %d MFMAs then %d v_pk_fma)\n", (THREADS == 256 ? (w[0] + w[1] + w[2] + w[3]) / 4 : (w[0] + w[4]) / 2) / (mi * 8.), B, B * V);
    PHASED(1, 16, 256, 0, "phased V=1 B=16 (1 wave/SIMD)")
    PHASED(1, 16, 512, 0, "phased V=1 B=16 (2 waves, lockstep)")
    PHASED(1, 16, 512, 1, "phased V=1 B=16 (2 waves, stagger)")
    PHASED(2, 16, 512, 0, "phased V=2 B=16 (2 waves, lockstep)")
    PHASED(2, 16, 512, 1, "phased V=2 B=16 (2 waves, stagger)")

    // control: bf16 MFMA as the matrix role
    double bf_alone, bv_alone;
    run("bf16 mfma_only (1 wave/SIMD)", 4, [&] { hipLaunchKernelGGL(bf16_partners_kernel, dim3(256), dim3(256), 0, 0, out, cycles, mi, vi, 0); }, w);
    bf_alone = (w[0] + w[1] + w[2] + w[3]) / 4;
    printf(" %.2f cycles per v_mfma_f32_16x16x32_bf16\n", bf_alone / (mi * 8.));
    run("valu_only again (1 wave/SIMD)", 4, [&] { hipLaunchKernelGGL(bf16_partners_kernel, dim3(256), dim3(256), 0, 0, out, cycles, mi, vi, 1); }, w);
    bv_alone = (w[0] + w[1] + w[2] + w[3]) / 4;
    printf(" %.2f cycles per v_pk_fma_f32\n", bv_alone / (vi * 32.));
    const int bvi = int(vi * bf_alone / bv_alone);
    run("bf16 partners (MFMA w0-3 | VALU w4-7)", 8, [&] { hipLaunchKernelGGL(bf16_partners_kernel, dim3(256), dim3(512), 0, 0, out, cycles, mi, bvi, 2); }, w);
    {
        const double m = (w[0] + w[1] + w[2] + w[3]) / 4, v = (w[4] + w[5] + w[6] + w[7]) / 4;
        const double v_alone = bv_alone * bvi / vi;
        printf(" matrix waves %.0f cycles (alone %.0f: x%.2f), vector waves %.0f (alone %.0f: x%.2f)\n", m, bf_alone,
               m / bf_alone, v, v_alone, v / v_alone);
    }

    // memory instructions between fp32 MFMAs
    float* source; CHECK(hipMalloc(&source, 4096 * 4)); CHECK(hipMemset(source, 0, 4096 * 4));
#define MEMORY(D, THREADS, label) \
    run(label, THREADS / 64, [&] { hipLaunchKernelGGL(memory_interleaved_kernel<D>, dim3(256), dim3(THREADS), 0, 0, out, cycles, mi, source); }, w); \
    printf(" %.2f cycles per MFMA and wave\n", (THREADS == 256 ? (w[0] + w[1] + w[2] + w[3]) / 4 : (w[0] + w[4]) / 2) / (mi * 8.));
    MEMORY(1, 256, "MFMA + 1 ds_read_b64 (1 wave/SIMD)")
    MEMORY(2, 256, "MFMA + 2 ds_read_b64 (1 wave/SIMD)")
    MEMORY(1, 512, "MFMA + 1 ds_read_b64 (2 waves/SIMD)")
    MEMORY(-1, 256, "MFMA + 1 global_load_x2 (1 wave)")
    MEMORY(-1, 512, "MFMA + 1 global_load_x2 (2 waves)")
    return 0;
}
