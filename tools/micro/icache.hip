// Is a kernel's code cold in the instruction cache at every launch?  A straight-line block
// of N vector instructions (8 bytes each) runs three times per launch from one wave per CU;
// the kernel is launched back to back.  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/icache.hip -o tools/micro/bin/icache
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))

template <int KB>
__global__ void block_kernel(unsigned long long* stamps, float* out) {
    float a = threadIdx.x, b = 1.0001f;
    unsigned long long t[4];
#pragma nounroll
    for (int pass = 0; pass < 3; ++pass) {
        t[pass] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // 256 x v_fma_f32 (VOP3, 8 bytes) = 2 KB per R256
        for (int k = 0; k < 1; ++k) {
            if (KB >= 2) { asm volatile(R256("v_fma_f32 %0, %0, %1, %1\n\t") : "+v"(a) : "v"(b)); }
            if (KB >= 4) { asm volatile(R256("v_fma_f32 %0, %0, %1, %1\n\t") : "+v"(a) : "v"(b)); }
            if (KB >= 8) { asm volatile(R256("v_fma_f32 %0, %0, %1, %1\n\t") R256("v_fma_f32 %0, %0, %1, %1\n\t") : "+v"(a) : "v"(b)); }
            if (KB >= 16) { asm volatile(R256("v_fma_f32 %0, %0, %1, %1\n\t") R256("v_fma_f32 %0, %0, %1, %1\n\t") R256("v_fma_f32 %0, %0, %1, %1\n\t") R256("v_fma_f32 %0, %0, %1, %1\n\t") : "+v"(a) : "v"(b)); }
        }
    }
    t[3] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0)
        for (int i = 0; i < 4; ++i) stamps[blockIdx.x * 4 + i] = t[i];
    out[blockIdx.x * 64 + threadIdx.x] = a;
}

template <int KB>
void run(unsigned long long* stamps, float* out, bool traffic, float* big, size_t big_floats) {
    const int blocks = 256;
    for (int i = 0; i < 5; ++i) {
        if (traffic) CHECK(hipMemsetAsync(big, i, big_floats * 4, 0));   // 64 MB through the L2s between launches
        block_kernel<KB><<<blocks, 64>>>(stamps, out);
    }
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double pass[3] = {0, 0, 0};
    for (int b = 0; b < blocks; ++b)
        for (int p = 0; p < 3; ++p) pass[p] += double(h[b * 4 + p + 1] - h[b * 4 + p]) / 100.;
    printf("%2d KB block%s: pass 0 %.2f us, pass 1 %.2f us, pass 2 %.2f us (mean over %d CUs)\n", KB,
           traffic ? ", 64 MB written between launches" : "", pass[0] / blocks, pass[1] / blocks,
           pass[2] / blocks, blocks);
}

int main() {
    unsigned long long* stamps; float* out; float* big;
    const size_t big_floats = 16u << 20;
    CHECK(hipMalloc(&stamps, 256 * 4 * 8)); CHECK(hipMalloc(&out, 256 * 64 * 4)); CHECK(hipMalloc(&big, big_floats * 4));
    for (int traffic = 0; traffic < 2; ++traffic) {
        run<2>(stamps, out, traffic, big, big_floats);
        run<4>(stamps, out, traffic, big, big_floats);
        run<8>(stamps, out, traffic, big, big_floats);
        run<16>(stamps, out, traffic, big, big_floats);
    }
    return 0;
}
