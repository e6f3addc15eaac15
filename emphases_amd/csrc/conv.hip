// Conv1d(padding='same') + bias + activation over ragged segments on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, bit-for-bit an fmaf chain).
//
// Replaces torch.nn.Conv1d + activation of the frame encoder / word decoder
// (emphases/model/core.py:17-21,93; model/layers/convolution.py:25-30) and, at
// kernel_size 1, the nn.Linear layers inside nn.TransformerEncoderLayer
// (model/layers/transformer.py:18-23).
//
// Formulation: implicit GEMM  Y[c_out, n] = W[c_out, (c_in, tap)] X[(c_in, tap), n]
// with M = c_out (16-row MFMA tiles), N = positions, K = c_in * taps.
//   * A workgroup is four waves, one per SIMD; each wave owns an
//     [MB*16 x NB*16] output tile: MB*NB independent fp32 accumulators keep the
//     32-cycle MFMA issue slot full (40-cycle dependent latency).
//   * The layer's weights, pre-packed on the host in A-fragment order
//     [k-step][m-tile][lane], are staged ONCE per workgroup into LDS (76.8 KB
//     for 80x80x3) and shared by the four waves; workgroups are persistent
//     (grid-stride over groups of four tiles), so weight traffic out of L2 is
//     per CU, not per tile.  An A fragment is one conflict-free ds_read_b32.
//     Layers whose pack exceeds the LDS budget are staged in K chunks.
//   * Activations go straight from L2/HBM to registers, no LDS stage: a B
//     fragment (4 input rows x 16 positions) is one global_load_dword per lane,
//     four 64-byte segments.  The K loop is software-pipelined through a
//     ping-pong pair of register sets so that every load has one full
//     iteration (U*MB*NB MFMAs, ~2000 cycles) to land.  Each segment's own
//     zero halo is applied by loop-invariant per-lane masks, so ragged batches
//     keep the reference's B=1 edge semantics.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// N consecutive floats of one input row, loaded with the widest instructions.
// The pieces are carried through the K loop AS vectors: when hipcc merges scalar
// loads into a dwordx3 on its own, the loop-carried registers stay scalars and
// it copies the tuple apart right after the load — a wait on a load that was
// just issued, in every iteration.
template <int N>
struct Run;
template <>
struct Run<1> {
    float a;
    __device__ __forceinline__ void load(const float* p) { a = p[0]; }
    __device__ __forceinline__ float get(int) const { return a; }
};
template <>
struct Run<3> {
    f32x3u a;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x3u*>(p);
    }
    __device__ __forceinline__ float get(int i) const { return a[i]; }
};
template <>
struct Run<4> {
    f32x4u a;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4u*>(p);
    }
    __device__ __forceinline__ float get(int i) const { return a[i]; }
};
template <>
struct Run<5> {
    f32x4u a;
    float b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4u*>(p);
        b = p[4];
    }
    __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b; }
};
template <>
struct Run<7> {
    f32x4u a;
    f32x3u b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4u*>(p);
        b = *reinterpret_cast<const f32x3u*>(p + 4);
    }
    __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
};

__host__ __device__ inline int round_up(int value, int multiple) {
    return (value + multiple - 1) / multiple * multiple;
}

template <int ACT>
__device__ __forceinline__ float activate(float x) {
    if (ACT == EMPH_ACT_RELU) return fmaxf(x, 0.f);
    if (ACT == EMPH_ACT_GELU)
        return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    if (ACT == EMPH_ACT_SILU) return x / (1.f + expf(-x));
    if (ACT == EMPH_ACT_LEAKY_RELU) return x > 0.f ? x : 0.01f * x;
    return x;
}

__device__ __forceinline__ float activate(float x, int act) {
    switch (act) {
        case EMPH_ACT_RELU: return activate<EMPH_ACT_RELU>(x);
        case EMPH_ACT_GELU: return activate<EMPH_ACT_GELU>(x);
        case EMPH_ACT_SILU: return activate<EMPH_ACT_SILU>(x);
        case EMPH_ACT_LEAKY_RELU: return activate<EMPH_ACT_LEAKY_RELU>(x);
        default: return x;
    }
}

// K is walked in iterations of U k-steps (4 input rows each): the taps of one
// row group for KS >= 3, two row groups for KS == 1.  Rows are padded with
// zeros so that the iteration count is even (the loop is a ping-pong pair).
__host__ __device__ constexpr int steps_per_iteration(int ks) { return ks == 1 ? 2 : ks; }
__host__ __device__ inline int padded_rows(int c_in, int ks) {
    return round_up(c_in, ks == 1 ? 16 : 8);
}

constexpr int kConvLdsBudget = 136 * 1024;            // weights only
// Waves per workgroup (one or two per SIMD), chosen at launch.
__host__ __device__ constexpr int conv_waves(int nb) { return nb == 4 ? 4 : 8; }
// floats per row of the epilogue patch; 16-byte aligned rows, and the two
// 16-lane k-groups of a 32-lane ds_write phase land on disjoint banks
__host__ __device__ constexpr int patch_stride(int nb) { return 16 * nb + 4; }

// Bias + activation + store of one [16 channels x NB*16 positions] patch held
// in LDS (row stride patch_stride(NB) floats), by one wave.  Channel-major
// output: a lane owns four consecutive positions of one channel (16-byte
// store, 256-byte runs per row).  Position-major output (transpose): a lane
// owns one position and writes its 16 channels as four 16-byte stores.
template <int NB>
__device__ __forceinline__ void store_patch(float* patch, float* __restrict__ y,
                                         int64_t ldy, const float* bias /* LDS, 16 */,
                                         int channel0, int c_out, int act, Tile span,
                                         int transpose) {
    constexpr int stride = patch_stride(NB);
    const int lane = threadIdx.x & 63;
    const int t0 = span.first;
    if (transpose) {
        const int t = t0 + lane;
        if (lane >= NB * 16 || t >= span.count) return;
        float* out = y + (static_cast<int64_t>(span.offset) + t) * ldy + channel0;
        const bool vector_ok = (ldy & 3) == 0 &&
                               (reinterpret_cast<uintptr_t>(y) & 15) == 0;
#pragma unroll 1
        for (int quad = 0; quad < 4; ++quad) {
            float value[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                value[e] = activate(
                    patch[(4 * quad + e) * stride + lane] + bias[4 * quad + e], act);
            }
            if (vector_ok && channel0 + 4 * quad + 3 < c_out) {
                *reinterpret_cast<float4*>(out + 4 * quad) =
                    make_float4(value[0], value[1], value[2], value[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (channel0 + 4 * quad + e < c_out) out[4 * quad + e] = value[e];
            }
        }
        return;
    }
    const int kk = lane >> 4;
    const int col = lane & 15;
    const bool vector_ok = (ldy & 3) == 0 && (span.offset & 3) == 0 &&
                           (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    // This function is inlined once per m-tile of the caller's unrolled
    // epilogue, so the straight-line part below must stay small: identity and
    // ReLU are a compare + select there; the transcendental activations are
    // applied in place on the LDS patch by a rolled loop first.
    const bool plain = act <= EMPH_ACT_RELU;
    if (!plain) {
#pragma unroll 1
        for (int index = lane; index < 16 * NB * 16; index += 64) {
            const int row = index / (NB * 16);
            float* cell = patch + row * stride + (index - row * (NB * 16));
            *cell = activate(*cell + bias[row], act);
        }
        wave_lds_fence();
    }
    const bool relu = act == EMPH_ACT_RELU;
    // all LDS reads of the patch first (independent), then the stores
    constexpr int kQuads = (NB * 4 + 15) / 16;
    float4 value[4][kQuads];
    float b[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int row = 4 * pass + kk;
        b[pass] = plain ? bias[row] : 0.f;
#pragma unroll
        for (int q = 0; q < kQuads; ++q) {
            const int quad = min(col + 16 * q, NB * 4 - 1);
            value[pass][q] =
                *reinterpret_cast<const float4*>(patch + row * stride + 4 * quad);
        }
    }
    auto finish = [&](float v, float add) {
        v += add;
        return (relu && v < 0.f) ? 0.f : v;       // NaN stays NaN, as torch.relu
    };
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int row = 4 * pass + kk;
        const int channel = channel0 + row;
#pragma unroll
        for (int q = 0; q < kQuads; ++q) {
            const int quad = col + 16 * q;
            const int t = t0 + 4 * quad;
            float4 v = value[pass][q];
            v.x = finish(v.x, b[pass]);
            v.y = finish(v.y, b[pass]);
            v.z = finish(v.z, b[pass]);
            v.w = finish(v.w, b[pass]);
            if (channel >= c_out || quad >= NB * 4 || t >= span.count) continue;
            float* out = y + static_cast<int64_t>(channel) * ldy + span.offset + t;
            if (vector_ok && t + 3 < span.count) {
                *reinterpret_cast<float4*>(out) = v;
            } else {
                out[0] = v.x;
                if (t + 1 < span.count) out[1] = v.y;
                if (t + 2 < span.count) out[2] = v.z;
                if (t + 3 < span.count) out[3] = v.w;
            }
        }
    }
}

// grid = (persistent workgroups, m_tiles_padded / MB); block = 64*WAVES threads
//
// Timeline of a workgroup (measured: a dependent global access costs 1-2 us on
// a busy chip, so nothing that can be requested early waits for anything):
//   tile descriptor -> request the weight pack (registers) and the first B
//   fragments -> commit pack + bias to LDS -> barrier -> K loop -> epilogue out
//   of LDS (no global reads).
template <int KS, int MB, int NB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv1d_kernel(
    const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
    const float* __restrict__ pack, const float* __restrict__ bias, int c_in,
    int c_out, int act, const int32_t* __restrict__ tiles, int n_tiles,
    int chunk_iterations, int patch_offset, int transpose_out) {
    constexpr int THREADS = 64 * WAVES;
    constexpr int kPatchStride = patch_stride(NB);
    constexpr int HALO = (KS - 1) / 2;
    constexpr int U = steps_per_iteration(KS);
    constexpr int GROUPS = KS == 1 ? U : 1;    // 4-row groups per iteration
    constexpr int kBatch = 10;                 // 16-byte loads in flight per lane
    extern __shared__ __align__(16) float weights[];   // [steps][MB][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4;      // k index inside the k-step
    const int col = lane & 15;     // A: row inside the m-tile, B: position
    const int rows = padded_rows(c_in, KS);
    const int m_tiles = (c_out + 15) >> 4;
    const int m_padded = gridDim.y * MB;
    const int m_first = blockIdx.y * MB;
    const int iterations = rows / (4 * GROUPS);
    const int chunks = (iterations + chunk_iterations - 1) / chunk_iterations;
    const int last_row = c_in - 1;
    float* patch = weights + patch_offset + wave * (16 * kPatchStride);
    float* bias_lds = weights + patch_offset + WAVES * (16 * kPatchStride);   // [MB*16]

    // ---- weight staging: pack[step][m_padded][64] -> weights[step][MB][64].
    // When the workgroup owns every m-tile the source is one contiguous run and
    // goes through LDS-DMA (global_load_lds: 1 KiB per wave instruction, no
    // VGPRs, every piece in flight at once — measured 1.1 us for 76.8 KB against
    // 4.4 us through registers).  Otherwise rows are gathered through
    // registers, kBatch requests a lane.
    float4 staged[kBatch];
    auto stage_issue = [&](int first, int count, int pass) {
        const int quads = count * U * MB * 16;
        if (m_padded == MB) {
            if (pass) return;
            const float* source = pack + static_cast<int64_t>(first) * U * MB * 64;
            for (int base = wave * 64; base < quads; base += THREADS) {
                if (base + 64 <= quads) {
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(
                            source + 4 * (base + lane)),
                        (__attribute__((address_space(3))) void*)(weights + 4 * base),
                        16, 0, 0);
                } else if (base + lane < quads) {
                    reinterpret_cast<float4*>(weights)[base + lane] =
                        reinterpret_cast<const float4*>(source)[base + lane];
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int index = (pass * kBatch + j) * THREADS + static_cast<int>(threadIdx.x);
            if (index < quads) {
                const int step = index / (MB * 16);
                const int rest = index - step * (MB * 16);
                staged[j] = reinterpret_cast<const float4*>(
                    pack + ((static_cast<int64_t>(first) * U + step) * m_padded +
                            m_first) * 64)[rest];
            }
        }
    };
    auto stage_commit = [&](int count, int pass) {
        if (m_padded == MB) {
            __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): LDS-DMA landed
            return;
        }
        const int quads = count * U * MB * 16;
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int index = (pass * kBatch + j) * THREADS + static_cast<int>(threadIdx.x);
            if (index < quads) reinterpret_cast<float4*>(weights)[index] = staged[j];
        }
    };
    auto stage_passes = [&](int count) {
        if (m_padded == MB) return 1;
        return (count * U * MB * 16 + THREADS * kBatch - 1) / (THREADS * kBatch);
    };

    // ---- per-tile state
    Tile span;
    int t0 = 0;
    bool active = false;
    bool inside[U][NB];
    const float* lane_base = x;
    auto open_tile = [&](int group) {
        const int tile = group * WAVES + wave;
        active = tile < n_tiles;                      // wave-uniform
        span = load_tile(tiles, active ? tile : 0);
        t0 = span.first;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int t = t0 + 16 * n + col + (KS == 1 ? 0 : u - HALO);
                inside[u][n] = t >= 0 && t < span.count;
            }
        // B fragments are read at compile-time offsets from this address; what
        // lies outside the segment (its halo, or a neighbour's columns) is
        // valid memory by the layout contract and is masked to zero below
        lane_base = x + span.offset + t0 + col - HALO;
    };
    // rows 4*group .. 4*group+3 of x; rows past c_in are clamped to a valid
    // address here and zeroed by `row_inside` in select_b()
    constexpr int RUN = KS == 1 ? 1 : KS;     // consecutive floats per fragment row
    auto load_b = [&](Run<RUN> (&b)[GROUPS][NB], int iteration) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int first = min(4 * (iteration * GROUPS + g), last_row & ~3);
            const float* source = lane_base +
                                  static_cast<int64_t>(first + min(kk, last_row - first)) * ldx;
#pragma unroll
            for (int n = 0; n < NB; ++n) b[g][n].load(source + 16 * n);
        }
    };
    // A fragments of one iteration: U*MB conflict-free LDS reads
    auto load_a = [&](float (&a)[U][MB], int local) {
        const float* fragment = weights + (local * U * MB << 6) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int m = 0; m < MB; ++m) a[u][m] = fragment[(u * MB + m) << 6];
    };

    float a0[U][MB];
    Run<RUN> b0[GROUPS][NB];
    const int groups_of_tiles = (n_tiles + WAVES - 1) / WAVES;
    open_tile(blockIdx.x);
    if (chunks == 1) {
        stage_issue(0, iterations, 0);
        if (active) load_b(b0, 0);
        for (int index = threadIdx.x; index < MB * 16; index += THREADS) {
            const int channel = m_first * 16 + index;
            bias_lds[index] = (bias != nullptr && channel < c_out) ? bias[channel] : 0.f;
        }
        stage_commit(iterations, 0);
        for (int pass = 1; pass < stage_passes(iterations); ++pass) {
            stage_issue(0, iterations, pass);
            stage_commit(iterations, pass);
        }
        __syncthreads();
    } else {
        for (int index = threadIdx.x; index < MB * 16; index += THREADS) {
            const int channel = m_first * 16 + index;
            bias_lds[index] = (bias != nullptr && channel < c_out) ? bias[channel] : 0.f;
        }
    }

    for (int group = blockIdx.x; group < groups_of_tiles; group += gridDim.x) {
        if (group != static_cast<int>(blockIdx.x)) {
            open_tile(group);
            if (chunks == 1 && active) load_b(b0, 0);
        }
        f32x4 acc[MB][NB];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        // zero what lies outside the segment / past the last input row; this
        // is also the point where the wave waits for the fragments to land
        auto select_b = [&](float (&bv)[U][NB], const Run<RUN> (&b)[GROUPS][NB],
                            int iteration) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rows_group = iteration * GROUPS + (KS == 1 ? u : 0);
                const bool row_inside = 4 * rows_group + kk < c_in;
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    bv[u][n] = (inside[u][n] && row_inside)
                                   ? b[KS == 1 ? u : 0][n].get(KS == 1 ? 0 : u)
                                   : 0.f;
            }
        };
        auto compute = [&](const float (&a)[U][MB], const float (&bv)[U][NB]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            a[u][m], bv[u][n], acc[m][n], 0, 0, 0);
        };

        for (int chunk = 0; chunk < chunks; ++chunk) {
            const int first = chunk * chunk_iterations;
            const int count = min(chunk_iterations, iterations - first);  // even
            if (chunks > 1) {
                __syncthreads();           // everyone is done with the old chunk
                for (int pass = 0; pass < stage_passes(count); ++pass) {
                    stage_issue(first, count, pass);
                    stage_commit(count, pass);
                }
                __syncthreads();
                if (active) load_b(b0, first);
            }
            if (!active) continue;
            // Software pipeline, one iteration deep: move the operands that
            // have landed into their MFMA registers (select_b masks B; A is
            // copied), immediately re-request the SAME registers for the next
            // iteration (B from global memory, A from LDS), then run this
            // iteration's MFMAs while those are in flight.  One load site per
            // operand keeps the loop-carried registers stable (a two-site
            // ping-pong made hipcc copy freshly loaded values at the loop end,
            // i.e. wait for them there), and the waits sit where nothing newer
            // is outstanding.  sched_barrier pins the three phases.
            float av[U][MB], bv[U][NB];
            load_a(a0, 0);
#pragma unroll 1
            for (int local = 0; local < count; ++local) {
                select_b(bv, b0, first + local);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int m = 0; m < MB; ++m) av[u][m] = a0[u][m];
                __builtin_amdgcn_sched_barrier(0);
                // the last iteration re-reads itself instead of branching
                const int next = min(local + 1, count - 1);
                load_b(b0, first + next);
                load_a(a0, next);
                compute(av, bv);
                // Issue order inside the block: the requests (and the address
                // arithmetic in front of them) are threaded between the MFMAs —
                // global loads first, they have the longest way to go — instead
                // of running ahead of them with the matrix pipe idle.
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);   // VALU/SALU
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
                }
#pragma unroll
                for (int k = 0; k < U * MB; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // MFMA
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!active) continue;

        // ---- epilogue.  The MFMA result layout is D[row = 4*(lane>>4) + r]
        // [col = lane&15].  Identity / ReLU with channel-major output store it
        // as it is - one dword per lane, 64-byte row pieces that pair up in L2 -
        // which beats the LDS round trip below (two fences and 0.45 us per
        // m-tile).  Transcendental activations and position-major output turn
        // each 16-row m-tile through a wave-private LDS patch and store_patch().
        if (act <= EMPH_ACT_RELU && !(transpose_out & 1)) {
            const bool relu = act == EMPH_ACT_RELU;
            float* row_base = y + static_cast<int64_t>(m_first * 16 + 4 * kk) * ldy +
                              span.offset + t0 + col;
            bool live[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) live[n] = t0 + 16 * n + col < span.count;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m_first + m >= m_tiles) break;
                const f32x4 add = *reinterpret_cast<const f32x4*>(bias_lds + 16 * m + 4 * kk);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool row_ok = (m_first + m) * 16 + 4 * kk + r < c_out;
                    float* out = row_base + static_cast<int64_t>(16 * m + r) * ldy;
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        float value = acc[m][n][r] + add[r];
                        value = (relu && value < 0.f) ? 0.f : value;
                        if (row_ok && live[n]) out[16 * n] = value;
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m_first + m >= m_tiles) break;
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        patch[(4 * kk + r) * kPatchStride + 16 * n + col] = acc[m][n][r];
                wave_lds_fence();
                store_patch<NB>(patch, y, ldy, bias_lds + 16 * m, (m_first + m) * 16, c_out,
                                act, span, transpose_out & 1);
                wave_lds_fence();
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Winograd F(2,3) variant of the kernel_size-3 convolution.
//
// Two adjacent outputs share four inputs d0..d3 = x[2p-1 .. 2p+2]:
//   m0 = (d0 - d2) g0              m1 = (d1 + d2) (g0 + g1 + g2)/2
//   m2 = (d2 - d1) (g0 - g1 + g2)/2   m3 = (d1 - d3) g2
//   y[2p] = m0 + m1 + m2           y[2p+1] = m1 - m2 - m3
// i.e. four [c_out x c_in] GEMMs over PAIRS of positions instead of three over
// positions: 2/3 of the MFMA work.  The input transform is two packed adds per
// fragment, the output transform four adds per accumulator, the transformed
// weights G_j are formed in float64 on the host.  In fp32 the result differs
// from the direct form by ~1e-7 of the output scale (per-word scores move by
// ~1e-7, measured against the reference goldens), far inside the 1e-4 budget.
//
// Structure as conv1d_kernel: 4 waves, the (c_in/4 x 4 x MB x 64)-float pack in
// LDS by LDS-DMA, one wave = MB m-tiles x NB pair-tiles (NB*32 positions), B
// fragments as one unaligned 16-byte run per lane, one-iteration pipeline.
// pack[m_block][group][j][m][lane] = G_j[(m_block*MB + m)*16 + (lane & 15)]
//                                       [4*group + (lane >> 4)]
template <int MB, int NB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv1d_winograd_kernel(
    const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
    const float* __restrict__ pack, const float* __restrict__ bias, int c_in,
    int c_out, int act, const int32_t* __restrict__ tiles, int n_tiles,
    int patch_offset) {
    constexpr int THREADS = 64 * WAVES;
    constexpr int kPatchStride = patch_stride(2 * NB);
    extern __shared__ __align__(16) float weights[];   // [groups][4][MB][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4;
    const int col = lane & 15;
    const int iterations = (c_in + 3) >> 2;            // 4-row groups
    const int m_tiles = (c_out + 15) >> 4;
    const int m_first = blockIdx.y * MB;
    const int last_row = c_in - 1;
    float* patch = weights + patch_offset + wave * (16 * kPatchStride);
    float* bias_lds = weights + patch_offset + WAVES * (16 * kPatchStride);

    // ---- per-tile state
    Tile span;
    int t0 = 0;
    bool active = false;
    bool inside[4][NB];
    const float* lane_base = x;
    auto open_tile = [&](int group) {
        const int tile = group * WAVES + wave;
        active = tile < n_tiles;
        span = load_tile(tiles, active ? tile : 0);
        t0 = span.first;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int t = t0 + 32 * n + 2 * col - 1 + i;
                inside[i][n] = t >= 0 && t < span.count;
            }
        lane_base = x + span.offset + t0 + 2 * col - 1;
    };
    auto load_b = [&](Run<4> (&b)[NB], int iteration) {
        const int first = min(4 * iteration, last_row & ~3);
        const float* source =
            lane_base + static_cast<int64_t>(first + min(kk, last_row - first)) * ldx;
#pragma unroll
        for (int n = 0; n < NB; ++n) b[n].load(source + 32 * n);
    };
    auto load_a = [&](float (&a)[4][MB], int iteration) {
        const float* fragment = weights + (iteration * 4 * MB << 6) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m) a[j][m] = fragment[(j * MB + m) << 6];
    };

    Run<4> b0[NB], b1[NB];
    float a0[4][MB];
    const int groups_of_tiles = (n_tiles + WAVES - 1) / WAVES;
    open_tile(blockIdx.x);
    {
        const int quads = iterations * 4 * MB * 16;
        const float* source =
            pack + static_cast<int64_t>(blockIdx.y) * iterations * 4 * MB * 64;
        for (int base = wave * 64; base < quads; base += THREADS)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(source + 4 * (base + lane)),
                (__attribute__((address_space(3))) void*)(weights + 4 * base), 16, 0, 0);
        if (active) load_b(b0, 0);
        for (int index = threadIdx.x; index < MB * 16; index += THREADS) {
            const int channel = m_first * 16 + index;
            bias_lds[index] = (bias != nullptr && channel < c_out) ? bias[channel] : 0.f;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): LDS-DMA landed
        __syncthreads();
    }

    for (int group = blockIdx.x; group < groups_of_tiles; group += gridDim.x) {
        if (group != static_cast<int>(blockIdx.x)) {
            // drain the previous tile's stores: with stores and loads both in
            // flight hipcc waits for vmcnt(0) at every use of a loaded value,
            // which would serialise the two requests the K loop keeps in flight
            __builtin_amdgcn_s_waitcnt(0x0F70);
            open_tile(group);
            if (active) load_b(b0, 0);
        }
        if (!active) continue;
        f32x4 acc[4][MB][NB];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        float av[4][MB], v[4][NB];
        load_a(a0, 0);
        // One K iteration (4 input channels): transform the B fragments that
        // have landed, request the ones TWO iterations ahead into the registers
        // just freed, then the 4 MB NB MFMAs with the next A fragments' LDS
        // reads between them.  One iteration of MFMAs (0.6 us for the two
        // waves of a SIMD) does not cover an L2 round trip on a busy chip
        // (0.7-1 us): with a single request in flight every iteration began
        // with a stall on vmcnt(0) (37.6 cycles per MFMA instead of 32).
        auto step = [&](Run<4> (&b)[NB], int iteration, int ahead) {
            const bool row_inside = 4 * iteration + kk < c_in;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                float d[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    d[i] = (inside[i][n] && row_inside) ? b[n].get(i) : 0.f;
                v[0][n] = d[0] - d[2];
                v[1][n] = d[1] + d[2];
                v[2][n] = d[2] - d[1];
                v[3][n] = d[1] - d[3];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MB; ++m) av[j][m] = a0[j][m];
            __builtin_amdgcn_sched_barrier(0);
            load_b(b, min(iteration + ahead, iterations - 1));
            load_a(a0, min(iteration + 1, iterations - 1));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[j][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            av[j][m], v[j][n], acc[j][m][n], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);   // VALU/SALU
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
            }
#pragma unroll
            for (int k = 0; k < 4 * MB; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (NB == 1) {
            // (whole pairs in the loop, the odd iteration after it: a
            // conditional second half would merge into a vmcnt(0) at the loop
            // head)
            load_b(b1, min(1, iterations - 1));
            int iteration = 0;
#pragma unroll 1
            for (; iteration + 1 < iterations; iteration += 2) {
                step(b0, iteration, 2);
                step(b1, iteration + 1, 2);
            }
            if (iteration < iterations) step(b0, iteration, 2);
        } else {
            // 64-position tiles: 40 MFMAs per iteration cover the round trip,
            // and a second fragment set does not fit the register file
#pragma unroll 1
            for (int iteration = 0; iteration < iterations; ++iteration)
                step(b0, iteration, 1);
        }

        // ---- output transform + epilogue.  Lane (kk, col) holds rows 4 kk + r
        // of pair `col` of pair-tile n, i.e. positions 32 n + 2 col (+1): for
        // identity / ReLU it stores them straight from registers as 8-byte
        // pairs (16 lanes = one 128-byte run per row; measured: the LDS patch
        // round trip below costs 0.45 us per m-tile of fences and latency).
        const bool plain = act <= EMPH_ACT_RELU && (ldy & 1) == 0 &&
                           (reinterpret_cast<uintptr_t>(y) & 7) == 0;
        if (plain) {
            const bool relu = act == EMPH_ACT_RELU;
            float add[MB][4];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) add[m][r] = bias_lds[16 * m + 4 * kk + r];
            float* row_base = y + static_cast<int64_t>(m_first * 16 + 4 * kk) * ldy +
                              span.offset + t0 + 2 * col;
            bool both[NB], first[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int t = t0 + 32 * n + 2 * col;
                both[n] = t + 1 < span.count;
                first[n] = t < span.count;
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m_first + m >= m_tiles) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool row_ok = (m_first + m) * 16 + 4 * kk + r < c_out;
                    float* out = row_base + static_cast<int64_t>(16 * m + r) * ldy;
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        const float m1 = acc[1][m][n][r], m2 = acc[2][m][n][r];
                        float even = acc[0][m][n][r] + m1 + m2 + add[m][r];
                        float odd = m1 - m2 - acc[3][m][n][r] + add[m][r];
                        even = (relu && even < 0.f) ? 0.f : even;
                        odd = (relu && odd < 0.f) ? 0.f : odd;
                        if (row_ok && both[n])
                            *reinterpret_cast<float2*>(out + 32 * n) = make_float2(even, odd);
                        else if (row_ok && first[n])
                            out[32 * n] = even;
                    }
                }
            }
        } else {
            // through the LDS patch (transcendental activations, odd strides)
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m_first + m >= m_tiles) break;
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m1 = acc[1][m][n][r], m2 = acc[2][m][n][r];
                        float2 pair;
                        pair.x = acc[0][m][n][r] + m1 + m2;
                        pair.y = m1 - m2 - acc[3][m][n][r];
                        *reinterpret_cast<float2*>(
                            patch + (4 * kk + r) * kPatchStride + 32 * n + 2 * col) = pair;
                    }
                wave_lds_fence();
                store_patch<2 * NB>(patch, y, ldy, bias_lds + 16 * m, (m_first + m) * 16,
                                    c_out, act, span, 0);
                wave_lds_fence();
            }
        }
    }
}

template <int KS, int MB, int NB, int WAVES>
int launch_conv_waves(int n_tiles, int m_blocks, size_t weight_bytes, hipStream_t s,
                      const float* x, int64_t ldx, float* y, int64_t ldy,
                      const float* pack, const float* bias, int c_in, int c_out,
                      int act, const int32_t* tiles, int chunk_iterations,
                      int transpose_out) {
    const size_t lds = weight_bytes +
                       (WAVES * 16 * patch_stride(NB) + MB * 16) * sizeof(float);
    const int patch_offset = static_cast<int>(weight_bytes / sizeof(float));
    auto kernel = conv1d_kernel<KS, MB, NB, WAVES>;
    static LdsReservation reserved;
    if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,
                                 "emph_conv1d"))
        return status;
    // persistent workgroups: LDS admits one (two for small packs) per CU
    const int groups_of_tiles = (n_tiles + WAVES - 1) / WAVES;
    const int resident = 256 * (lds > 80 * 1024 ? 1 : 2);
    dim3 grid(groups_of_tiles < resident ? groups_of_tiles : resident, m_blocks);
    EMPH_LAUNCH(kernel, grid, dim3(64 * WAVES), lds, s, x, ldx, y, ldy, pack,
                       bias, c_in, c_out, act, tiles, n_tiles, chunk_iterations,
                       patch_offset, transpose_out);
    return check_launch("emph_conv1d");
}

template <int KS, int MB, int NB>
int launch_conv(int n_tiles, int m_blocks, size_t weight_bytes, hipStream_t s,
                const float* x, int64_t ldx, float* y, int64_t ldy,
                const float* pack, const float* bias, int c_in, int c_out, int act,
                const int32_t* tiles, int chunk_iterations, int transpose_out) {
    if (conv_waves(NB) == 4)
        return launch_conv_waves<KS, MB, NB, 4>(n_tiles, m_blocks, weight_bytes, s, x,
                                                ldx, y, ldy, pack, bias, c_in, c_out,
                                                act, tiles, chunk_iterations,
                                                transpose_out);
    return launch_conv_waves<KS, MB, NB, 8>(n_tiles, m_blocks, weight_bytes, s, x, ldx,
                                            y, ldy, pack, bias, c_in, c_out, act, tiles,
                                            chunk_iterations, transpose_out);
}

template <int KS, int MB>
int launch_conv_nb(int tile_n, int n_tiles, int m_blocks, size_t weight_bytes,
                   hipStream_t s, const float* x, int64_t ldx, float* y,
                   int64_t ldy, const float* pack, const float* bias, int c_in,
                   int c_out, int act, const int32_t* tiles, int chunk_iterations,
                   int transpose_out) {
    switch (tile_n) {
        case 16:
            return launch_conv<KS, MB, 1>(n_tiles, m_blocks, weight_bytes, s, x, ldx, y,
                                          ldy, pack, bias, c_in, c_out, act, tiles,
                                          chunk_iterations, transpose_out);
        case 32:
            return launch_conv<KS, MB, 2>(n_tiles, m_blocks, weight_bytes, s, x, ldx, y,
                                          ldy, pack, bias, c_in, c_out, act, tiles,
                                          chunk_iterations, transpose_out);
        default:
            return launch_conv<KS, MB, 4>(n_tiles, m_blocks, weight_bytes, s, x, ldx, y,
                                          ldy, pack, bias, c_in, c_out, act, tiles,
                                          chunk_iterations, transpose_out);
    }
}

template <int MB, int NB, int WAVES>
static int launch_winograd(const float* x, int64_t ldx, float* y, int64_t ldy,
                           const float* pack, const float* bias, int c_in, int c_out,
                           int act, const int32_t* tiles, int n_tiles, int m_blocks,
                           hipStream_t s) {
    const size_t weight_bytes = static_cast<size_t>((c_in + 3) / 4) * 4 * MB * 64 * sizeof(float);
    const size_t lds =
        weight_bytes + (WAVES * 16 * patch_stride(2 * NB) + MB * 16) * sizeof(float);
    EMPH_REQUIRE(lds <= 160 * 1024, EMPH_ERANGE,
                 "emph_conv1d_winograd: %zu bytes of LDS needed (c_in too large)", lds);
    auto kernel = conv1d_winograd_kernel<MB, NB, WAVES>;
    static LdsReservation reserved;
    if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,
                                 "emph_conv1d"))
        return status;
    const int groups_of_tiles = (n_tiles + WAVES - 1) / WAVES;
    dim3 grid(groups_of_tiles < 256 ? groups_of_tiles : 256, m_blocks);
    EMPH_LAUNCH(kernel, grid, dim3(64 * WAVES), lds, s, x, ldx, y, ldy, pack, bias, c_in,
                       c_out, act, tiles, n_tiles,
                       static_cast<int>(weight_bytes / sizeof(float)));
    return check_launch("emph_conv1d_winograd");
}

}  // namespace emph

using namespace emph;

extern "C" {

static int conv_m_block(int c_out) {
    const int m_tiles = (c_out + 15) / 16;
    return (m_tiles % 4 == 0) ? 4 : 5;
}

int64_t emph_conv_pack_size(int32_t c_out, int32_t c_in, int32_t kernel_size) {
    const int mb = conv_m_block(c_out);
    const int64_t m_padded = round_up((c_out + 15) / 16, mb);
    const int64_t k_steps =
        static_cast<int64_t>(kernel_size) * (padded_rows(c_in, kernel_size) / 4);
    return k_steps * m_padded * 64;
}

int emph_conv_pack(const float* host_weight, int32_t c_out, int32_t c_in,
                   int32_t kernel_size, float* host_pack) {
    EMPH_REQUIRE(host_weight && host_pack, EMPH_EINVAL, "emph_conv_pack: null pointer");
    EMPH_REQUIRE(c_out > 0 && c_in > 0 && kernel_size > 0, EMPH_EINVAL,
                 "emph_conv_pack: bad shape");
    const int m_padded = round_up((c_out + 15) / 16, conv_m_block(c_out));
    const int row_groups = padded_rows(c_in, kernel_size) / 4;
    // pack[step][m][lane] = W[m*16 + (lane & 15)][4*group + (lane >> 4)][tap]
    // with step = group * k + tap (A fragment: row = lane & 15, k = lane >> 4);
    // rows/channels beyond the real shape are zero.
    for (int group = 0; group < row_groups; ++group)
        for (int tap = 0; tap < kernel_size; ++tap)
            for (int m = 0; m < m_padded; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = m * 16 + (lane & 15);
                    const int ci = 4 * group + (lane >> 4);
                    const int64_t step =
                        static_cast<int64_t>(group) * kernel_size + tap;
                    float value = 0.f;
                    if (co < c_out && ci < c_in)
                        value = host_weight[(static_cast<int64_t>(co) * c_in + ci) *
                                                kernel_size + tap];
                    host_pack[(step * m_padded + m) * 64 + lane] = value;
                }
    return EMPH_OK;
}

int64_t emph_conv_winograd_pack_size(int32_t c_out, int32_t c_in) {
    const int mb = conv_m_block(c_out);
    const int64_t m_padded = round_up((c_out + 15) / 16, mb);
    return static_cast<int64_t>((c_in + 3) / 4) * 4 * m_padded * 64;
}

int64_t emph_conv_winograd_lds_bytes(int32_t c_out, int32_t c_in) {
    const int mb = conv_m_block(c_out);
    return (static_cast<int64_t>((c_in + 3) / 4) * 4 * mb * 64 +
            4 * 16 * patch_stride(4) + mb * 16) * static_cast<int64_t>(sizeof(float));
}

int emph_conv_winograd_pack(const float* host_weight, int32_t c_out, int32_t c_in,
                            float* host_pack) {
    EMPH_REQUIRE(host_weight && host_pack, EMPH_EINVAL,
                 "emph_conv_winograd_pack: null pointer");
    EMPH_REQUIRE(c_out > 0 && c_in > 0, EMPH_EINVAL, "emph_conv_winograd_pack: bad shape");
    const int mb = conv_m_block(c_out);
    const int m_padded = round_up((c_out + 15) / 16, mb);
    const int blocks = m_padded / mb;
    const int groups = (c_in + 3) / 4;
    for (int block = 0; block < blocks; ++block)
        for (int group = 0; group < groups; ++group)
            for (int j = 0; j < 4; ++j)
                for (int m = 0; m < mb; ++m)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = (block * mb + m) * 16 + (lane & 15);
                        const int ci = 4 * group + (lane >> 4);
                        double value = 0.;
                        if (co < c_out && ci < c_in) {
                            const float* w = host_weight + (static_cast<int64_t>(co) * c_in + ci) * 3;
                            const double g0 = w[0], g1 = w[1], g2 = w[2];
                            value = j == 0   ? g0
                                    : j == 1 ? (g0 + g1 + g2) / 2
                                    : j == 2 ? (g0 - g1 + g2) / 2
                                             : g2;
                        }
                        host_pack[((((static_cast<int64_t>(block) * groups + group) * 4 + j) * mb +
                                    m) << 6) + lane] = static_cast<float>(value);
                    }
    return EMPH_OK;
}

int emph_conv1d_winograd(const float* x, int64_t ldx, float* y, int64_t ldy,
                         const float* pack, const float* bias, int32_t c_in,
                         int32_t c_out, int32_t activation, const int32_t* tiles,
                         int32_t n_tiles, int32_t tile_n, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(x && y && pack && tiles, EMPH_EINVAL, "emph_conv1d_winograd: null pointer");
    EMPH_REQUIRE(tile_n == 32 || tile_n == 64, EMPH_ERANGE,
                 "emph_conv1d_winograd: tile_n %d not in {32,64}", tile_n);
    EMPH_REQUIRE(c_in >= 1 && c_in <= 256 && c_out >= 1 && c_out <= 1024, EMPH_ERANGE,
                 "emph_conv1d_winograd: channels %d -> %d out of range", c_in, c_out);
    EMPH_REQUIRE(activation >= EMPH_ACT_NONE && activation <= EMPH_ACT_LEAKY_RELU,
                 EMPH_EINVAL, "emph_conv1d_winograd: unknown activation %d", activation);
    const int m_tiles = (c_out + 15) / 16;
    const int mb = conv_m_block(c_out);
    const int m_blocks = (m_tiles + mb - 1) / mb;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // 32-position tiles run two waves per SIMD (the second hides the first's
    // LDS / global latencies: 24.5 us vs 26.7 us with one 64-position wave)
#define EMPH_WINOGRAD(MB, NB, WAVES)                                                      \
    launch_winograd<MB, NB, WAVES>(x, ldx, y, ldy, pack, bias, c_in, c_out, activation, \
                                   tiles, n_tiles, m_blocks, s)
    if (mb == 5) {
        if (tile_n == 64) return EMPH_WINOGRAD(5, 2, 4);
        return EMPH_WINOGRAD(5, 1, 8);
    }
    if (tile_n == 64) return EMPH_WINOGRAD(4, 2, 4);
    return EMPH_WINOGRAD(4, 1, 8);
#undef EMPH_WINOGRAD
}

int emph_conv1d(const float* x, int64_t ldx, float* y, int64_t ldy,
                const float* pack, const float* bias, int32_t c_in,
                int32_t c_out, int32_t kernel_size, int32_t activation,
                const int32_t* tiles, int32_t n_tiles, int32_t tile_n,
                int32_t transpose_out, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(x && y && pack && tiles, EMPH_EINVAL,
                 "emph_conv1d: null pointer");
    EMPH_REQUIRE(kernel_size == 1 || kernel_size == 3 || kernel_size == 5 ||
                     kernel_size == 7,
                 EMPH_ERANGE, "emph_conv1d: kernel_size %d not in {1,3,5,7}",
                 kernel_size);
    EMPH_REQUIRE(tile_n == 16 || tile_n == 32 || tile_n == 64, EMPH_ERANGE,
                 "emph_conv1d: tile_n %d not in {16,32,64}", tile_n);
    EMPH_REQUIRE(c_in >= 1 && c_in <= 256 && c_out >= 1 && c_out <= 1024,
                 EMPH_ERANGE, "emph_conv1d: channels %d -> %d out of range", c_in,
                 c_out);
    EMPH_REQUIRE(activation >= EMPH_ACT_NONE && activation <= EMPH_ACT_LEAKY_RELU,
                 EMPH_EINVAL, "emph_conv1d: unknown activation %d", activation);
    EMPH_REQUIRE(ldx > 0 && ldx < (int64_t{1} << 28), EMPH_ERANGE,
                 "emph_conv1d: ldx out of range");
    const int m_tiles = (c_out + 15) / 16;
    const int mb = conv_m_block(c_out);
    // K iterations resident in LDS at a time (even, see the ping-pong loop)
    const int u = steps_per_iteration(kernel_size);
    const int groups = kernel_size == 1 ? u : 1;
    const int iterations = padded_rows(c_in, kernel_size) / (4 * groups);
    const size_t iteration_bytes = static_cast<size_t>(u) * mb * 64 * sizeof(float);
    int chunk_iterations = static_cast<int>(kConvLdsBudget / iteration_bytes) & ~1;
    if (chunk_iterations > iterations) chunk_iterations = iterations;
    const size_t weight_bytes = chunk_iterations * iteration_bytes;
    const int m_blocks = (m_tiles + mb - 1) / mb;
    hipStream_t s = static_cast<hipStream_t>(stream);
    transpose_out = transpose_out ? 1 : 0;
#define EMPH_CONV(KS)                                                             \
    return mb == 5                                                                \
               ? launch_conv_nb<KS, 5>(tile_n, n_tiles, m_blocks, weight_bytes, s, x, \
                                       ldx, y, ldy, pack, bias, c_in, c_out,      \
                                       activation, tiles, chunk_iterations,       \
                                       transpose_out)                             \
               : launch_conv_nb<KS, 4>(tile_n, n_tiles, m_blocks, weight_bytes, s, x, \
                                       ldx, y, ldy, pack, bias, c_in, c_out,      \
                                       activation, tiles, chunk_iterations,       \
                                       transpose_out)
    switch (kernel_size) {
        case 1: EMPH_CONV(1);
        case 3: EMPH_CONV(3);
        case 5: EMPH_CONV(5);
        default: EMPH_CONV(7);
    }
#undef EMPH_CONV
}

}  // extern "C"
