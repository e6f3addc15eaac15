// How fast can every workgroup pull the same 76.8 KB weight pack into LDS?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kQuads = 4800;   // 76.8 KB / 16

// MODE 0: shared pack, linear order; 1: shared pack, rotated start; 2: private
// copy per workgroup; 3: LDS-DMA shared; 4: shared, but only 1 KB per request wave
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void stage_kernel(const float4* pack, float* out,
                                                        unsigned long long* stamps) {
    extern __shared__ __align__(16) float lds[];
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    const float4* source = pack;
    if (MODE == 2) source += static_cast<size_t>(blockIdx.x) * kQuads;
    const int rotate = MODE == 1 ? (blockIdx.x * 2654435761u) % kQuads : 0;
    if (MODE == 3) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int base = wave * 64; base < kQuads; base += THREADS)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(source + base + lane),
                (__attribute__((address_space(3))) void*)(lds + 4 * base), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        constexpr int kBatch = (kQuads + THREADS - 1) / THREADS;
        float4 staged[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            int index = j * THREADS + threadIdx.x;
            if (index < kQuads) {
                index += rotate;
                if (index >= kQuads) index -= kQuads;
                staged[j] = source[index];
            }
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            int index = j * THREADS + threadIdx.x;
            if (index < kQuads) {
                index += rotate;
                if (index >= kQuads) index -= kQuads;
                reinterpret_cast<float4*>(lds)[index] = staged[j];
            }
        }
    }
    __syncthreads();
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t_start;
        stamps[2 * blockIdx.x + 1] = t_end;
        out[blockIdx.x] = lds[blockIdx.x % (4 * kQuads)];
    }
}

template <int MODE, int THREADS>
void run(const char* name, const float4* pack, float* out, unsigned long long* stamps, int blocks) {
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start)); CHECK(hipEventCreate(&stop));
    auto kernel = stage_kernel<MODE, THREADS>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(THREADS), 96 * 1024, 0, pack, out, stamps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(start));
    for (int i = 0; i < 20; ++i)
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(THREADS), 96 * 1024, 0, pack, out, stamps);
    CHECK(hipEventRecord(stop));
    CHECK(hipEventSynchronize(stop));
    float ms; CHECK(hipEventElapsedTime(&ms, start, stop));
    std::vector<unsigned long long> host(2 * blocks);
    CHECK(hipMemcpy(host.data(), stamps, host.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> spans;
    for (int b = 0; b < blocks; ++b) spans.push_back((host[2 * b + 1] - host[2 * b]) * 0.01);
    std::sort(spans.begin(), spans.end());
    printf("%-44s %d thr x %d blocks: %6.2f us/launch; in-kernel stage median %5.2f max %5.2f us\n",
           name, THREADS, blocks, ms * 1e3 / 20, spans[blocks / 2], spans.back());
}

int main() {
    const int blocks = 250;
    float4* pack; float* out; unsigned long long* stamps;
    CHECK(hipMalloc(&pack, static_cast<size_t>(blocks) * kQuads * 16));
    CHECK(hipMemset(pack, 0, static_cast<size_t>(blocks) * kQuads * 16));
    CHECK(hipMalloc(&out, blocks * 4)); CHECK(hipMalloc(&stamps, blocks * 16));
    run<0, 256>("shared pack, linear", pack, out, stamps, blocks);
    run<1, 256>("shared pack, rotated start", pack, out, stamps, blocks);
    run<2, 256>("private pack per workgroup", pack, out, stamps, blocks);
    run<3, 256>("shared pack, LDS-DMA", pack, out, stamps, blocks);
    run<0, 512>("shared pack, linear", pack, out, stamps, blocks);
    run<3, 512>("shared pack, LDS-DMA", pack, out, stamps, blocks);
    run<0, 1024>("shared pack, linear", pack, out, stamps, blocks);
    run<3, 1024>("shared pack, LDS-DMA", pack, out, stamps, blocks);
    run<0, 256>("shared pack, linear, 64 blocks", pack, out, stamps, 64);
    run<0, 256>("shared pack, linear, 8 blocks", pack, out, stamps, 8);
    return 0;
}
