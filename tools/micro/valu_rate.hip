// Issue / execute rate of packed vs scalar f32 vector instructions on gfx950, by
// waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float cf __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, unsigned long long* cycles, int iterations) {
    cf a[8];
    for (int i = 0; i < 8; ++i) a[i] = {threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    const cf k = {1.0001f, 0.9999f}, c = {1e-4f, -1e-4f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(c));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(k.x), "v"(c.x));
                if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
                if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
                if (KIND == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x));
                if (KIND == 7) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(k.x), "v"(c.x));
                                 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(k.y), "v"(c.y)); }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cycles;
    CHECK(hipMalloc(&out, 256 * 1024 * 4)); CHECK(hipMalloc(&cycles, 256 * 16 * 8));
    const char* names[8] = {"v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_fma_f32", "v_add_f32", "v_pk_add_f32 op_sel/neg", "v_sqrt_f32", "2 x v_fma_f32"};
    const int iterations = 2000;
    for (int kind = 0; kind < 8; ++kind)
        for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
            const int threads = 256 * waves_per_simd;   // one workgroup per CU: its waves spread over the 4 SIMDs
            auto launch = [&]() {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(threads), 0, 0, out, cycles, iterations); break;
                    case 1: hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(threads), 0, 0, out, cycles, iterations); break;
                    case 2: hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(threads), 0, 0, out, cycles, iterations); break;
                    case 3: hipLaunchKernelGGL(rate_kernel<3>, dim3(256), dim3(threads), 0, 0, out, cycles, iterations); break;
                    case 4: hipLaunchKernelGGL(rate_kernel<4>, dim3(256), dim3(threads), 0, 0, out, cycles, iterations); break;
                    case 5: hipLaunchKernelGGL(rate_kernel<5>, dim3(256), dim3(threads), 0, 0, out, cycles, iterations); break;
                    case 6: hipLaunchKernelGGL(rate_kernel<6>, dim3(256), dim3(threads), 0, 0, out, cycles, iterations); break;
                    default: hipLaunchKernelGGL(rate_kernel<7>, dim3(256), dim3(threads), 0, 0, out, cycles, iterations); break;
                }
            };
            launch(); CHECK(hipDeviceSynchronize());
            launch(); CHECK(hipDeviceSynchronize());
            unsigned long long host[64];
            CHECK(hipMemcpy(host, cycles, sizeof(host), hipMemcpyDeviceToHost));
            double mean = 0; const int n = threads / 64;
            for (int i = 0; i < n; ++i) mean += double(host[i]);
            mean /= n;
            const double per = mean / (double(iterations) * 32 * (kind == 7 ? 1 : 1));
            printf("%-26s %d wave(s)/SIMD: %6.2f cycles per wave-instruction%s, %6.2f per SIMD\n", names[kind], waves_per_simd,
                   per, kind == 7 ? " pair" : "", per / waves_per_simd);
        }
    return 0;
}
