// Conv1d(padding='same') + bias + activation over ragged segments on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, bit-for-bit an fmaf chain).
//
// Replaces torch.nn.Conv1d + activation of the frame encoder / word decoder
// (emphases/model/core.py:17-21,93; model/layers/convolution.py:25-30) and, at
// kernel_size 1, the nn.Linear layers inside nn.TransformerEncoderLayer
// (model/layers/transformer.py:18-23).
//
// Formulation: implicit GEMM  Y[c_out, n] = W[c_out, (tap, c_in)] X[(tap, c_in), n]
// with M = c_out (16-row MFMA tiles), N = positions, K = taps * c_in.
//   * One wave owns an [MB*16 x NB*16] output tile: MB*NB independent fp32
//     accumulators keep the 32-cycle MFMA issue slot full (40-cycle dependent
//     latency) from a single wave per SIMD.
//   * The input tile (all c_in rows, NB*16 positions + halo) is staged once in
//     LDS with 16-byte loads; each segment's own zero halo is applied there, so
//     ragged batches keep the reference's B=1 edge semantics.  Row stride is
//     16 (mod 32) floats, which makes the four k-rows of a B fragment hit
//     disjoint banks (conflict-free ds_read_b32).
//   * Weights are pre-packed on the host in A-fragment order
//     [k-step][m-tile][lane]; a fragment load is one coalesced 256-byte
//     global_load_dword that every wave on the chip shares through L2/L1.
#include <math.h>

#include "common.h"

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLead = 4;   // positions kept left of the tile (16-byte aligned halo)

__host__ __device__ inline int round_up(int value, int multiple) {
    return (value + multiple - 1) / multiple * multiple;
}

// floats per staged row: tile + 2*kLead, rounded to 16 (mod 32)
__host__ __device__ inline int stage_stride(int tile_n) {
    int stride = tile_n + 2 * kLead;
    while (stride % 32 != 16) ++stride;
    return stride;
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

// grid = (n_tiles, m_tiles_padded / MB); block = 64 threads (one wave)
template <int KS, int MB, int NB>
__global__ __launch_bounds__(64) void conv1d_kernel(
    const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
    const float* __restrict__ pack, const float* __restrict__ bias, int c_in,
    int c_out, int act, const int64_t* __restrict__ seg, int axis,
    const int32_t* __restrict__ tiles, int transpose_out) {
    constexpr int TN = NB * 16;
    constexpr int HALO = (KS - 1) / 2;
    constexpr int U = steps_per_iteration(KS);
    extern __shared__ __align__(16) float xs[];
    const int stride = stage_stride(TN);
    const int lane = threadIdx.x;
    const int rows = padded_rows(c_in, KS);
    const int m_tiles = (c_out + 15) >> 4;
    const int m_padded = gridDim.y * MB;
    const int m_first = blockIdx.y * MB;

    const int segment = tiles[2 * blockIdx.x];
    const int t0 = tiles[2 * blockIdx.x + 1];
    const Span span = load_span(seg, segment, axis);

    // ---- stage x[:, t0 - kLead : t0 + TN + kLead) with the segment's zero halo
    {
        constexpr int quads = (TN + 2 * kLead) / 4;
        const int total = rows * quads;
        const float* base = x + span.offset + t0 - kLead;
        for (int index = lane; index < total; index += 64) {
            const int row = index / quads;
            const int quad = index - row * quads;
            const int t = t0 - kLead + 4 * quad;       // segment-relative
            float4 value = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < c_in && t + 3 >= 0 && t < span.count) {
                value = *reinterpret_cast<const float4*>(
                    base + static_cast<int64_t>(row) * ldx + 4 * quad);
                if (t < 0 || t + 3 >= span.count) {
                    if (t < 0 || t >= span.count) value.x = 0.f;
                    if (t + 1 < 0 || t + 1 >= span.count) value.y = 0.f;
                    if (t + 2 < 0 || t + 2 >= span.count) value.z = 0.f;
                    if (t + 3 < 0 || t + 3 >= span.count) value.w = 0.f;
                }
            }
            *reinterpret_cast<float4*>(xs + row * stride + 4 * quad) = value;
        }
    }
    __syncthreads();

    f32x4 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kk = lane >> 4;      // k index inside the k-step
    const int col = lane & 15;     // A: row inside the m-tile, B: position
    const int iterations = KS == 1 ? rows >> 3 : rows >> 2;
    const int64_t step_stride = static_cast<int64_t>(m_padded) << 6;
    const int64_t iteration_stride = step_stride * U;
    const float* fragment = pack + (static_cast<int64_t>(m_first) << 6) + lane;
    const float* b_base = xs + kk * stride + kLead + col - HALO;

    // A fragments are prefetched one iteration (U*MB*NB MFMAs) ahead into the
    // other register set of a ping-pong pair; sched_barrier pins the issue point
    float a0[U][MB], a1[U][MB];
    auto load_a = [&](float (&a)[U][MB]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int m = 0; m < MB; ++m) a[u][m] = fragment[u * step_stride + (m << 6)];
    };
    auto compute = [&](const float (&a)[U][MB]) {
        float b[U][NB];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int n = 0; n < NB; ++n)
                b[u][n] = KS == 1 ? b_base[4 * u * stride + n * 16]
                                  : b_base[n * 16 + u];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        a[u][m], b[u][n], acc[m][n], 0, 0, 0);
        b_base += 4 * (KS == 1 ? U : 1) * stride;
    };
    load_a(a0);
    for (int iteration = 0; iteration < iterations; iteration += 2) {
        fragment += iteration_stride;
        load_a(a1);
        __builtin_amdgcn_sched_barrier(0);
        compute(a0);
        __builtin_amdgcn_sched_barrier(0);
        // the final pair re-reads its own fragments instead of branching
        if (iteration + 2 < iterations) fragment += iteration_stride;
        load_a(a0);
        __builtin_amdgcn_sched_barrier(0);
        compute(a1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: bias + activation; D[row = 4*(lane>>4) + r][col = lane&15]
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        if (m_first + m >= m_tiles) break;
        const int channel = (m_first + m) * 16 + 4 * kk;
        float b4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            b4[r] = (bias != nullptr && channel + r < c_out) ? bias[channel + r] : 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int t = t0 + n * 16 + col;
            if (t >= span.count) continue;
            const int64_t position = span.offset + t;
            if (transpose_out) {
                // y[position][channel .. channel+3]
                float4 value;
                value.x = activate(acc[m][n][0] + b4[0], act);
                value.y = activate(acc[m][n][1] + b4[1], act);
                value.z = activate(acc[m][n][2] + b4[2], act);
                value.w = activate(acc[m][n][3] + b4[3], act);
                float* out = y + position * ldy + channel;
                if (channel + 3 < c_out) {
                    *reinterpret_cast<float4*>(out) = value;
                } else {
                    if (channel < c_out) out[0] = value.x;
                    if (channel + 1 < c_out) out[1] = value.y;
                    if (channel + 2 < c_out) out[2] = value.z;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (channel + r < c_out)
                        y[static_cast<int64_t>(channel + r) * ldy + position] =
                            activate(acc[m][n][r] + b4[r], act);
            }
        }
    }
}


template <int KS, int MB>
int launch_conv_nb(int tile_n, dim3 grid, size_t lds, hipStream_t s,
                   const float* x, int64_t ldx, float* y, int64_t ldy,
                   const float* pack, const float* bias, int c_in, int c_out,
                   int act, const int64_t* seg, int axis, const int32_t* tiles,
                   int transpose_out) {
    switch (tile_n) {
        case 16:
            hipLaunchKernelGGL((conv1d_kernel<KS, MB, 1>), grid, dim3(64), lds, s,
                               x, ldx, y, ldy, pack, bias, c_in, c_out, act, seg,
                               axis, tiles, transpose_out);
            break;
        case 32:
            hipLaunchKernelGGL((conv1d_kernel<KS, MB, 2>), grid, dim3(64), lds, s,
                               x, ldx, y, ldy, pack, bias, c_in, c_out, act, seg,
                               axis, tiles, transpose_out);
            break;
        default:
            hipLaunchKernelGGL((conv1d_kernel<KS, MB, 4>), grid, dim3(64), lds, s,
                               x, ldx, y, ldy, pack, bias, c_in, c_out, act, seg,
                               axis, tiles, transpose_out);
            break;
    }
    return check_launch("emph_conv1d");
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

int emph_conv1d(const float* x, int64_t ldx, float* y, int64_t ldy,
                const float* pack, const float* bias, int32_t c_in,
                int32_t c_out, int32_t kernel_size, int32_t activation,
                const int64_t* seg, int32_t axis, const int32_t* tiles,
                int32_t n_tiles, int32_t tile_n, int32_t transpose_out,
                void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(x && y && pack && seg && tiles, EMPH_EINVAL,
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
    EMPH_REQUIRE((ldx & 3) == 0, EMPH_EINVAL,
                 "emph_conv1d: ldx must be a multiple of 4");
    const int m_tiles = (c_out + 15) / 16;
    const int mb = conv_m_block(c_out);
    const size_t lds = static_cast<size_t>(padded_rows(c_in, kernel_size)) *
                       stage_stride(tile_n) * sizeof(float);
    dim3 grid(n_tiles, (m_tiles + mb - 1) / mb);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define EMPH_CONV(KS)                                                             \
    return mb == 5 ? launch_conv_nb<KS, 5>(tile_n, grid, lds, s, x, ldx, y, ldy,  \
                                           pack, bias, c_in, c_out, activation,  \
                                           seg, axis, tiles, transpose_out)      \
                   : launch_conv_nb<KS, 4>(tile_n, grid, lds, s, x, ldx, y, ldy,  \
                                           pack, bias, c_in, c_out, activation,  \
                                           seg, axis, tiles, transpose_out)
    switch (kernel_size) {
        case 1: EMPH_CONV(1);
        case 3: EMPH_CONV(3);
        case 5: EMPH_CONV(5);
        default: EMPH_CONV(7);
    }
#undef EMPH_CONV
}

}  // extern "C"
