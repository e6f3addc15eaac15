// The position-wise half of a post-LN Transformer encoder layer in ONE launch:
//   y = LayerNorm1(x + W_o a + b_o)
//   x <- LayerNorm2(y + W_2 relu(W_1 y + b_1) + b_2)
// (a = attention output), i.e. out_proj, residual, norm1, linear1, activation,
// linear2, residual, norm2 of nn.TransformerEncoderLayer
// (emphases/model/layers/transformer.py:18-23).  As separate launches these
// are three kernel_size-1 convs and two residual LayerNorms: 240 MB of HBM
// traffic and five launch overheads per layer on 64 x 10 s; fused, the three
// GEMMs run out of registers and 60 MB move.
//
// A wave owns NB*16 positions and ALL channels, which makes the chain possible:
// the MFMA result layout D[row = 4*(lane>>4) + r][col] of m-tile m holds
// channel 16 m + 4 (lane>>4) + r of position col, and that is a valid B
// fragment (k = lane>>4) of the next GEMM if its k-step s = 4 m + r multiplies
// input channels {16 m + 4 k + r}: the next layer's weights are packed in that
// order (emph_linear_chain_pack) and the activations never leave the registers
// they were accumulated in.  LayerNorm statistics are 4 MB in-lane values plus a
// reduction over the four 16-lane rows (v_permlane swaps).
#include <math.h>

#include "common.h"

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rows_sum4(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// grid = workgroups of 8 waves over the tile table; MB = channels / 16.
// LDS: three weight packs [4 MB steps][MB][64] + 7 vectors of `channels`.
//
// QKV: the NEXT layer's Q, K, V projections (in_proj of nn.MultiheadAttention) in
// the same launch - the layer's output is in the accumulator layout, i.e. already
// the B fragments of a GEMM whose weights are packed in chain order, so the
// three projections read it from the registers it was normalised in instead of
// from memory in a launch of their own.  LDS then holds six packs and ten vectors
// (156.8 KB for 80 channels).
template <int MB, int NB, bool QKV>
__global__ __launch_bounds__(512) void transformer_block_kernel(
    const float* __restrict__ attended, float* __restrict__ x, int64_t ld,
    const float* __restrict__ packs /* out | linear1 | linear2 [| q | k | v] */,
    const float* __restrict__ vectors /* b_o g1 be1 b_1 b_2 g2 be2 [b_q b_k b_v] */, float eps,
    int act, const int32_t* __restrict__ tiles, int n_tiles, float* __restrict__ qk,
    float* __restrict__ v_out) {
    constexpr int C = 16 * MB;
    constexpr int STEPS = 4 * MB;                 // k-steps of one GEMM
    constexpr int PACK = STEPS * MB * 64;         // floats per pack
    constexpr int PACKS = QKV ? 6 : 3;
    constexpr int VECTORS = QKV ? 10 : 7;
    extern __shared__ __align__(16) float lds[];
    float* vec = lds + PACKS * PACK;              // [VECTORS][C]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4;
    const int col = lane & 15;

    // stage the packs by LDS-DMA and the vectors by plain loads
    for (int base = wave * 64; base < PACKS * PACK / 4; base += 512)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(packs + 4 * (base + lane)),
            (__attribute__((address_space(3))) void*)(lds + 4 * base), 16, 0, 0);
    for (int index = threadIdx.x; index < VECTORS * C; index += 512)
        vec[index] = vectors[index];
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    const float* b_o = vec;
    const float* g1 = vec + C;
    const float* be1 = vec + 2 * C;
    const float* b_1 = vec + 3 * C;
    const float* b_2 = vec + 4 * C;
    const float* g2 = vec + 5 * C;
    const float* be2 = vec + 6 * C;

    for (int tile = blockIdx.x * 8 + wave; tile < n_tiles; tile += gridDim.x * 8) {
        const Tile span = load_tile(tiles, tile);
        const int t0 = span.first;
        // row pointers are wave-uniform (scalar registers); a lane adds one
        // 32-bit offset: its k row (4 kk rows down) and its position
        bool live[NB];
        int64_t column_of[NB];
        uint32_t lane_offset[NB];      // accumulator layout: row 4 kk (+ 16 m + r)
        uint32_t operand_offset[NB];   // B-fragment layout: row kk (+ 4 g)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            live[n] = t0 + 16 * n + col < span.count;
            const int64_t column = span.offset + min(t0 + 16 * n + col, span.count - 1);
            column_of[n] = column;
            lane_offset[n] = static_cast<uint32_t>(4 * kk * ld + column);
            operand_offset[n] = static_cast<uint32_t>(kk * ld + column);
        }
        // B fragments of the attention output (natural channel order) and the
        // residual stream in the accumulator layout, all requested together
        // (the accumulators start from the residual stream, so only three
        // [C x positions] register tiles are ever live: operand, accumulator,
        // and the value carried to the second residual)
        float b0[STEPS][NB];
        f32x4 y[MB][NB];
#pragma unroll
        for (int g = 0; g < STEPS; ++g)
#pragma unroll
            for (int n = 0; n < NB; ++n)
                b0[g][n] = (attended + static_cast<int64_t>(4 * g) * ld)[operand_offset[n]];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    y[m][n][r] = (x + static_cast<int64_t>(16 * m + r) * ld)[lane_offset[n]];

        // one GEMM: acc[m][n] += sum_s A[s][m] * B(s)[n], A from the LDS pack, read
        // one k-step ahead of its MFMAs (the scheduling barriers keep hipcc from
        // hoisting all 4 MB steps' reads to the top: 1.6 KB of scratch per lane)
        auto gemm = [&](const float* pack, f32x4 (&acc)[MB][NB], auto fragment) {
            float a[2][MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) a[0][m] = pack[(m << 6) + lane];
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < STEPS) {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
                        a[(s + 1) & 1][m] = pack[(((s + 1) * MB + m) << 6) + lane];
                }
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            a[s & 1][m], fragment(s, n), acc[m][n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // v <- LayerNorm(v) over the channels of each position (two passes,
        // as torch.nn.LayerNorm)
        auto layernorm = [&](f32x4 (&v)[MB][NB], const float* gamma, const float* beta) {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                float sum = 0.f;
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum += v[m][n][r];
                const float mean = rows_sum4(sum) / static_cast<float>(C);
                float square = 0.f;
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[m][n][r] -= mean;
                        square = fmaf(v[m][n][r], v[m][n][r], square);
                    }
                const float rstd =
                    1.f / sqrtf(rows_sum4(square) / static_cast<float>(C) + eps);
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const f32x4 scale = *reinterpret_cast<const f32x4*>(gamma + 16 * m + 4 * kk);
                    const f32x4 shift = *reinterpret_cast<const f32x4*>(beta + 16 * m + 4 * kk);
                    v[m][n] = v[m][n] * rstd * scale + shift;
                }
            }
        };
        auto add_bias = [&](f32x4 (&v)[MB][NB], const float* bias) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const f32x4 add = *reinterpret_cast<const f32x4*>(bias + 16 * m + 4 * kk);
#pragma unroll
                for (int n = 0; n < NB; ++n) v[m][n] += add;
            }
        };

        // y = LayerNorm1(x + b_o + W_o a)
        add_bias(y, b_o);
        gemm(lds, y, [&](int s, int n) { return b0[s][n]; });
        layernorm(y, g1, be1);
        // h = relu(b_1 + W_1 y)
        f32x4 h[MB][NB];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const f32x4 add = *reinterpret_cast<const f32x4*>(b_1 + 16 * m + 4 * kk);
#pragma unroll
            for (int n = 0; n < NB; ++n) h[m][n] = add;
        }
        gemm(lds + PACK, h, [&](int s, int n) { return y[s >> 2][n][s & 3]; });
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = h[m][n][r];
                    h[m][n][r] = (act == EMPH_ACT_RELU && v < 0.f) ? 0.f : v;
                }
        // z = LayerNorm2(y + b_2 + W_2 h), accumulated in y's registers
        f32x4 (&z)[MB][NB] = y;
        add_bias(z, b_2);
        gemm(lds + 2 * PACK, z, [&](int s, int n) { return h[s >> 2][n][s & 3]; });
        layernorm(z, g2, be2);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (live[n])
                        (x + static_cast<int64_t>(16 * m + r) * ld)[lane_offset[n]] = z[m][n][r];
        if constexpr (QKV) {
            // the next layer's Q | K (channel-major) and V (position-major)
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                f32x4 acc[MB][NB];
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const f32x4 add = *reinterpret_cast<const f32x4*>(
                        vec + (7 + part) * C + 16 * m + 4 * kk);
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc[m][n] = add;
                }
                gemm(lds + (3 + part) * PACK, acc,
                     [&](int s, int n) { return z[s >> 2][n][s & 3]; });
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        if (!live[n]) continue;
                        if (part < 2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                (qk + static_cast<int64_t>(part * C + 16 * m + r) *
                                          ld)[lane_offset[n]] = acc[m][n][r];
                        } else {
                            *reinterpret_cast<f32x4*>(v_out + column_of[n] * C + 16 * m +
                                                      4 * kk) = acc[m][n];
                        }
                    }
            }
        }
    }
}

// Q, K and V projections of self-attention (in_proj of nn.MultiheadAttention,
// transformer.py:18-23) in one launch: x is read once as B fragments, Q and K
// are written channel-major for emph_attention's S^T = K Q^T, V position-major
// (a lane's accumulator register is four consecutive channels of one position:
// one 16-byte store).  As three kernel_size-1 convs this was 65 us per layer on
// 64 x 10 s.
template <int MB, int NB>
__global__ __launch_bounds__(512) void qkv_kernel(
    const float* __restrict__ x, int64_t ld, float* __restrict__ qk, float* __restrict__ v,
    const float* __restrict__ packs /* q | k | v, natural order */,
    const float* __restrict__ bias /* [3][C] */, const int32_t* __restrict__ tiles,
    int n_tiles) {
    constexpr int C = 16 * MB;
    constexpr int STEPS = 4 * MB;
    constexpr int PACK = STEPS * MB * 64;
    extern __shared__ __align__(16) float lds[];
    float* vec = lds + 3 * PACK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4;
    const int col = lane & 15;
    for (int base = wave * 64; base < 3 * PACK / 4; base += 512)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(packs + 4 * (base + lane)),
            (__attribute__((address_space(3))) void*)(lds + 4 * base), 16, 0, 0);
    for (int index = threadIdx.x; index < 3 * C; index += 512) vec[index] = bias[index];
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    for (int tile = blockIdx.x * 8 + wave; tile < n_tiles; tile += gridDim.x * 8) {
        const Tile span = load_tile(tiles, tile);
        const int t0 = span.first;
        bool live[NB];
        int64_t column[NB];
        uint32_t lane_offset[NB], operand_offset[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            live[n] = t0 + 16 * n + col < span.count;
            column[n] = span.offset + min(t0 + 16 * n + col, span.count - 1);
            lane_offset[n] = static_cast<uint32_t>(4 * kk * ld + column[n]);
            operand_offset[n] = static_cast<uint32_t>(kk * ld + column[n]);
        }
        float b0[STEPS][NB];
#pragma unroll
        for (int g = 0; g < STEPS; ++g)
#pragma unroll
            for (int n = 0; n < NB; ++n)
                b0[g][n] = (x + static_cast<int64_t>(4 * g) * ld)[operand_offset[n]];
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            f32x4 acc[MB][NB];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const f32x4 add =
                    *reinterpret_cast<const f32x4*>(vec + part * C + 16 * m + 4 * kk);
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[m][n] = add;
            }
            const float* pack = lds + part * PACK;
            float a[2][MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) a[0][m] = pack[(m << 6) + lane];
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < STEPS) {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
                        a[(s + 1) & 1][m] = pack[(((s + 1) * MB + m) << 6) + lane];
                }
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            a[s & 1][m], b0[s][n], acc[m][n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    if (!live[n]) continue;
                    if (part < 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            (qk + static_cast<int64_t>(part * C + 16 * m + r) *
                                      ld)[lane_offset[n]] = acc[m][n][r];
                    } else {
                        *reinterpret_cast<f32x4*>(v + column[n] * C + 16 * m + 4 * kk) =
                            acc[m][n];
                    }
                }
        }
    }
}

}  // namespace emph

using namespace emph;

extern "C" {

int64_t emph_linear_chain_pack_size(int32_t channels) {
    return static_cast<int64_t>(channels) * channels;
}

// natural = 1: k-step g multiplies input channels 4 g .. 4 g + 3 (B fragments
// loaded from memory); natural = 0: k-step s = 4 m + r multiplies channels
// {16 m + 4 k + r} (B fragments = the previous GEMM's accumulators).
int emph_linear_chain_pack(const float* host_weight, int32_t channels, int32_t natural,
                           float* host_pack) {
    EMPH_REQUIRE(host_weight && host_pack, EMPH_EINVAL,
                 "emph_linear_chain_pack: null pointer");
    EMPH_REQUIRE(channels >= 16 && channels <= 128 && channels % 16 == 0, EMPH_ERANGE,
                 "emph_linear_chain_pack: channels %d not a multiple of 16 in 16..128",
                 channels);
    const int mb = channels / 16;
    for (int s = 0; s < 4 * mb; ++s)
        for (int m = 0; m < mb; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const int k = lane >> 4;
                const int ci = natural ? 4 * s + k : 16 * (s >> 2) + 4 * k + (s & 3);
                const int co = 16 * m + (lane & 15);
                host_pack[((static_cast<int64_t>(s) * mb + m) << 6) + lane] =
                    host_weight[static_cast<int64_t>(co) * channels + ci];
            }
    return EMPH_OK;
}

static int launch_block(const float* attended, float* x, int64_t ld, int32_t channels,
                        const float* packs, const float* vectors, float eps,
                        int32_t activation, const int32_t* tiles, int32_t n_tiles,
                        int32_t tile_n, float* qk, float* v, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    const bool qkv = qk != nullptr;
    EMPH_REQUIRE(attended && x && packs && vectors && tiles, EMPH_EINVAL,
                 "emph_transformer_block: null pointer");
    EMPH_REQUIRE(channels == 64 || channels == 80, EMPH_ERANGE,
                 "emph_transformer_block: channels %d not in {64, 80} (three %d x %d packs "
                 "must fit in LDS)", channels, channels, channels);
    EMPH_REQUIRE(tile_n == 16 || tile_n == 32, EMPH_ERANGE,
                 "emph_transformer_block: tile_n %d not in {16, 32}", tile_n);
    EMPH_REQUIRE(ld > 0 && ld < (int64_t{1} << 28), EMPH_ERANGE,
                 "emph_transformer_block: ld %lld outside the 32-bit lane offsets",
                 static_cast<long long>(ld));
    EMPH_REQUIRE(activation == EMPH_ACT_RELU || activation == EMPH_ACT_NONE, EMPH_ERANGE,
                 "emph_transformer_block: activation %d (ReLU or none)", activation);
    const size_t lds = ((qkv ? 6 : 3) * static_cast<size_t>(channels) * channels +
                        (qkv ? 10 : 7) * channels) * sizeof(float);
    EMPH_REQUIRE(lds <= 160 * 1024, EMPH_ERANGE, "emph_transformer_block: %zu bytes of LDS", lds);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int groups = (n_tiles + 7) / 8;
    dim3 grid(groups < 256 ? groups : 256);
#define EMPH_BLOCK_LAUNCH(MB, NB, QKV)                                                         \
    do {                                                                                       \
        auto kernel = transformer_block_kernel<MB, NB, QKV>;                                   \
        static LdsReservation reserved;                                                        \
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,     \
                                     "emph_transformer_block"))                                \
            return status;                                                                     \
        EMPH_LAUNCH(kernel, grid, dim3(512), lds, s, attended, x, ld, packs, vectors, eps,     \
                    activation, tiles, n_tiles, qk, v);                                        \
    } while (0)
#define EMPH_BLOCK(MB, NB)                                                                     \
    do {                                                                                       \
        if (qkv) EMPH_BLOCK_LAUNCH(MB, NB, true);                                              \
        else EMPH_BLOCK_LAUNCH(MB, NB, false);                                                 \
    } while (0)
    if (channels == 80) {
        if (tile_n == 32) EMPH_BLOCK(5, 2); else EMPH_BLOCK(5, 1);
    } else {
        if (tile_n == 32) EMPH_BLOCK(4, 2); else EMPH_BLOCK(4, 1);
    }
#undef EMPH_BLOCK
#undef EMPH_BLOCK_LAUNCH
    return check_launch("emph_transformer_block");
}

int emph_transformer_block(const float* attended, float* x, int64_t ld, int32_t channels,
                           const float* packs, const float* vectors, float eps,
                           int32_t activation, const int32_t* tiles, int32_t n_tiles,
                           int32_t tile_n, void* stream) {
    return launch_block(attended, x, ld, channels, packs, vectors, eps, activation, tiles,
                        n_tiles, tile_n, nullptr, nullptr, stream);
}

int emph_transformer_block_qkv(const float* attended, float* x, int64_t ld, int32_t channels,
                               const float* packs, const float* vectors, float eps,
                               int32_t activation, const int32_t* tiles, int32_t n_tiles,
                               int32_t tile_n, float* qk, float* v, void* stream) {
    EMPH_REQUIRE(qk && v, EMPH_EINVAL, "emph_transformer_block_qkv: null output");
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(v) & 15) == 0, EMPH_EINVAL,
                 "emph_transformer_block_qkv: v must be 16-byte aligned");
    return launch_block(attended, x, ld, channels, packs, vectors, eps, activation, tiles,
                        n_tiles, tile_n, qk, v, stream);
}

int emph_qkv_projection(const float* x, int64_t ld, float* qk, float* v, int32_t channels,
                        const float* packs, const float* bias, const int32_t* tiles,
                        int32_t n_tiles, int32_t tile_n, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(x && qk && v && packs && bias && tiles, EMPH_EINVAL,
                 "emph_qkv_projection: null pointer");
    EMPH_REQUIRE(channels == 64 || channels == 80, EMPH_ERANGE,
                 "emph_qkv_projection: channels %d not in {64, 80}", channels);
    EMPH_REQUIRE(tile_n == 16 || tile_n == 32, EMPH_ERANGE,
                 "emph_qkv_projection: tile_n %d not in {16, 32}", tile_n);
    EMPH_REQUIRE(ld > 0 && ld < (int64_t{1} << 28), EMPH_ERANGE,
                 "emph_qkv_projection: ld %lld outside the 32-bit lane offsets",
                 static_cast<long long>(ld));
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(v) & 15) == 0, EMPH_EINVAL,
                 "emph_qkv_projection: v must be 16-byte aligned");
    const size_t lds = (3 * static_cast<size_t>(channels) * channels + 3 * channels) *
                       sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int groups = (n_tiles + 7) / 8;
    dim3 grid(groups < 256 ? groups : 256);
#define EMPH_QKV(MB, NB)                                                                \
    do {                                                                                \
        auto kernel = qkv_kernel<MB, NB>;                                               \
        static LdsReservation reserved;                                                        \
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,     \
                                     "emph_transformer_block"))                                \
            return status;                                                                     \
        EMPH_LAUNCH(kernel, grid, dim3(512), lds, s, x, ld, qk, v, packs, bias,  \
                           tiles, n_tiles);                                             \
    } while (0)
    if (channels == 80) {
        if (tile_n == 32) EMPH_QKV(5, 2); else EMPH_QKV(5, 1);
    } else {
        if (tile_n == 32) EMPH_QKV(4, 2); else EMPH_QKV(4, 1);
    }
#undef EMPH_QKV
    return check_launch("emph_qkv_projection");
}

}  // extern "C"
