// The position-wise half of a post-LN Transformer encoder layer, and the Q / K / V
// projections in front of the attention, on the bf16 matrix pipe with every fp32
// operand split into bf16 pieces (split.h: two pieces and three products per term,
// "bf16x3", or three and six, "bf16x6"; fp32 accumulation) - the opt-in precisions of
// the engine for 80 channels; the default stays the fp32 kernels of block.hip.
//   transformer_block_split_kernel   y = LayerNorm1(x + W_o a + b_o)
//                                    x <- LayerNorm2(y + W_2 relu(W_1 y + b_1) + b_2)
//   qkv_split_kernel                 Q | K (channel-major), V (position-major), or
//                                    Q and the split images of K and V that
//                                    attention_split_kernel stages (no emph_split_kv)
// (out_proj, residual, norm1, linear1, activation, linear2, residual, norm2 and
// in_proj of nn.TransformerEncoderLayer, emphases/model/layers/transformer.py:18-23).
//
// As in block.hip a wave owns a tile of positions (32 here) and ALL channels, and the
// chain of GEMMs runs out of registers: the result layout of
// v_mfma_f32_32x32x16_bf16 - lane (position, half) holds rows 8 b + 4 half + i of an
// m-tile - IS the B operand of the next GEMM once split, if that GEMM's k-step j
// multiplies input channels 16 j + 8 (e / 4) + 4 half + e % 4 (e = 0 .. 7): the
// weights are packed in that order (emph_linear_split_pack), for the first GEMM of a
// kernel the operand is simply loaded in it.  80 channels: three m-tiles (rows 80 ..
// 95 are zero weights), five k-steps, 45 (90) MFMAs per GEMM and 32 positions against
// 200 fp32 MFMAs of the same duration.  A pack is 30 (45) KB, so the three GEMMs of each
// kernel fit in LDS but not all six: the block and the next layer's projections are two
// launches here.
//
// Four waves per workgroup, one per SIMD, 512 registers each: what hides the memory
// latency is the wave's own next tile, requested while the current one is multiplied.
#include <stdlib.h>
#include <string.h>

#include "split.h"

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kLinearChannels = 80;
constexpr int kLinearSteps = kLinearChannels / 16;             // k-steps
constexpr int kLinearMTiles = 3;                               // 96 rows
constexpr int kLinearTile = 32;                                // positions per wave
constexpr int kLinearWaves = 4;
constexpr int kLinearThreads = 64 * kLinearWaves;
constexpr int linear_pack_bytes(int pieces) { return kLinearSteps * kLinearMTiles * pieces * 1024; }

// A lane's column of channel-major rows: the row's address is wave-uniform (scalar
// registers), what varies with the lane - its column and its half's four channels - is
// ONE 32-bit offset, so that eighty loads of a tile hold no eighty addresses in vector
// registers (the entry points keep 5 ld below 2^32).
__device__ __forceinline__ uint32_t linear_lane_offset(int64_t ld, int64_t column, int half) {
    return static_cast<uint32_t>(column) + static_cast<uint32_t>(4 * half) * static_cast<uint32_t>(ld);
}
__device__ __forceinline__ const float& linear_at(const float* base, int64_t ld, int channel,
                                                  uint32_t lane_offset) {
    return (base + static_cast<int64_t>(channel) * ld)[lane_offset];
}
__device__ __forceinline__ float& linear_at(float* base, int64_t ld, int channel,
                                            uint32_t lane_offset) {
    return (base + static_cast<int64_t>(channel) * ld)[lane_offset];
}

// register r of m-tile m holds channel 32 m + 8 (r / 4) + 4 half + r % 4; the third
// m-tile's registers 8 .. 15 are rows 80 .. 95 (nothing)
#define EMPH_LINEAR_VALID(m, r) (32 * (m) + 8 * ((r) >> 2) < kLinearChannels)

// acc[m] += W (pack, in LDS) x B.  `fragment(j, b)`: the B operand of k-step j, split.
// SWAP: the operands trade places (the result is transposed: lane = (output channel,
// half), registers = positions).  The weights of k-step j + 1 are read from LDS while
// the MFMAs of k-step j run, and nothing moves further than that: left alone, hipcc
// hoists the reads of a whole GEMM (and spills hundreds of registers doing it).
template <int P, bool SWAP, typename Fragment>
__device__ __forceinline__ void linear_gemm(const unsigned char* pack, int lane,
                                            f32x16 (&acc)[kLinearMTiles], Fragment fragment) {
    u32x4 a[2][kLinearMTiles][P];
    auto weights = [&](int j, u32x4 (&out)[kLinearMTiles][P]) {
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int piece = 0; piece < P; ++piece)
                out[m][piece] = *reinterpret_cast<const u32x4*>(
                    pack + ((j * kLinearMTiles + m) * P + piece) * 1024 + 16 * lane);
    };
    weights(0, a[0]);
#pragma unroll
    for (int j = 0; j < kLinearSteps; ++j) {
        if (j + 1 < kLinearSteps) weights(j + 1, a[(j + 1) & 1]);
        u32x4 b[P];
        fragment(j, b);
        // the small products first; three independent accumulators between two MFMAs
        // on the same one
#pragma unroll
        for (int order = P - 1; order >= 0; --order)
#pragma unroll
            for (int i = 0; i <= order; ++i)
#pragma unroll
                for (int m = 0; m < kLinearMTiles; ++m)
                    acc[m] = SWAP ? mfma_bf16(b[order - i], a[j & 1][m][i], acc[m])
                                  : mfma_bf16(a[j & 1][m][i], b[order - i], acc[m]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ... with the operand taken from the registers of the GEMM in front: `value(j, e)`
template <int P, typename Value>
__device__ __forceinline__ void linear_chain(const unsigned char* pack, int lane,
                                             f32x16 (&acc)[kLinearMTiles], Value value) {
    linear_gemm<P, false>(pack, lane, acc, [&](int j, u32x4 (&b)[P]) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = value(j, e);
        split_eight<P>(v, b);
    });
}

// v <- LayerNorm(v) over the channels of each position (two passes, as
// torch.nn.LayerNorm): 40 values in the lane, 40 in the lane of the other half
__device__ __forceinline__ void linear_layernorm(f32x16 (&v)[kLinearMTiles], const float* gamma,
                                                 const float* beta, float eps, int half) {
    constexpr int C = kLinearChannels;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (EMPH_LINEAR_VALID(m, r)) sum += v[m][r];
    const float mean = (sum + __shfl_xor(sum, 32)) / static_cast<float>(C);
    float square = 0.f;
#pragma unroll
    for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (EMPH_LINEAR_VALID(m, r)) {
                v[m][r] -= mean;
                square = fmaf(v[m][r], v[m][r], square);
            }
    const float rstd =
        1.f / sqrtf((square + __shfl_xor(square, 32)) / static_cast<float>(C) + eps);
#pragma unroll
    for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (32 * m + 8 * b >= C) continue;
            const f32x4 scale = *reinterpret_cast<const f32x4*>(gamma + 32 * m + 8 * b + 4 * half);
            const f32x4 shift = *reinterpret_cast<const f32x4*>(beta + 32 * m + 8 * b + 4 * half);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[m][4 * b + i] = v[m][4 * b + i] * rstd * scale[i] + shift[i];
        }
}

// The packs (and FLOATS floats of vectors behind them) into LDS: requested by the
// constructor, landed behind landed() - the first tile's inputs are requested in between,
// so that one trip to memory covers both.
template <int P, int FLOATS>
struct LinearWeights {
    static constexpr int kHeld = (FLOATS + kLinearThreads - 1) / kLinearThreads;
    float held[kHeld];
    float* vec;
    __device__ __forceinline__ LinearWeights(unsigned char* lds, const unsigned char* packs,
                                             const float* vectors) {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        vec = reinterpret_cast<float*>(lds + 3 * linear_pack_bytes(P));
        // (ahead of the LDS-DMA requests: what waits for these must not wait for those)
#pragma unroll
        for (int i = 0; i < kHeld; ++i)
            held[i] = vectors[min(static_cast<int>(threadIdx.x) + i * kLinearThreads, FLOATS - 1)];
        for (int base = wave * 64; base < 3 * linear_pack_bytes(P) / 16; base += kLinearThreads)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(packs + 16 * (base + lane)),
                (__attribute__((address_space(3))) void*)(lds + 16 * base), 16, 0, 0);
    }
    __device__ __forceinline__ void landed() {
#pragma unroll
        for (int i = 0; i < kHeld; ++i)
            if (static_cast<int>(threadIdx.x) + i * kLinearThreads < FLOATS)
                vec[threadIdx.x + i * kLinearThreads] = held[i];
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
};

// grid = workgroups of four waves over the tile table (tiles of 32 positions)
template <int P>
__global__ __launch_bounds__(kLinearThreads) void transformer_block_split_kernel(
    const float* __restrict__ attended, float* __restrict__ x, int64_t ld,
    const unsigned char* __restrict__ packs /* out | linear1 | linear2 */,
    const float* __restrict__ vectors /* b_o g1 be1 b_1 b_2 g2 be2 */, float eps, int act,
    const int32_t* __restrict__ tiles, int n_tiles) {
    constexpr int C = kLinearChannels;
    constexpr int PACK = linear_pack_bytes(P);
    extern __shared__ __align__(16) unsigned char block_split_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;
    const int half = lane >> 5;
    LinearWeights<P, 7 * C> weights(block_split_lds, packs, vectors);
    const float* vec = weights.vec;
    const float* b_o = vec;
    const float* g1 = vec + C;
    const float* be1 = vec + 2 * C;
    const float* b_1 = vec + 3 * C;
    const float* b_2 = vec + 4 * C;
    const float* g2 = vec + 5 * C;
    const float* be2 = vec + 6 * C;

    auto layernorm = [&](f32x16 (&v)[kLinearMTiles], const float* gamma, const float* beta) {
        linear_layernorm(v, gamma, beta, eps, half);
    };
    auto bias_of = [&](const float* bias, int m, int r) {
        return bias[32 * m + 8 * (r >> 2) + 4 * half + (r & 3)];
    };
    // a tile's inputs: the attention output in operand order, the residual stream in
    // the accumulator layout (columns beyond the segment read its last one)
    auto request = [&](int tile, float (&operand)[kLinearSteps][8], f32x16 (&residual)[kLinearMTiles]) {
        const Tile span = load_tile(tiles, tile);
        const uint32_t at = linear_lane_offset(
            ld, span.offset + min(span.first + col, span.count - 1), half);
#pragma unroll
        for (int j = 0; j < kLinearSteps; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                operand[j][e] = linear_at(attended, ld, 16 * j + 8 * (e >> 2) + (e & 3), at);
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                residual[m][r] = EMPH_LINEAR_VALID(m, r)
                                     ? linear_at(x, ld, 32 * m + 8 * (r >> 2) + (r & 3), at)
                                     : 0.f;
    };

    const int stride = gridDim.x * kLinearWaves;
    int tile = blockIdx.x * kLinearWaves + wave;
    float operand[kLinearSteps][8];
    f32x16 residual[kLinearMTiles];
    if (tile < n_tiles) request(tile, operand, residual);
    weights.landed();
    for (; tile < n_tiles; tile += stride) {
        const Tile span = load_tile(tiles, tile);
        const bool live = span.first + col < span.count;
        const int64_t column = span.offset + min(span.first + col, span.count - 1);
        // this tile's inputs leave their registers (split, biased) ...
        u32x4 a_frag[kLinearSteps][P];
#pragma unroll
        for (int j = 0; j < kLinearSteps; ++j) split_eight<P>(operand[j], a_frag[j]);
        f32x16 y[kLinearMTiles];
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                y[m][r] = EMPH_LINEAR_VALID(m, r) ? residual[m][r] + bias_of(b_o, m, r) : 0.f;
        // ... and the next tile's are requested into them, in flight during the MFMAs
        if (tile + stride < n_tiles) request(tile + stride, operand, residual);
        // y = LayerNorm1(x + b_o + W_o a)
        linear_gemm<P, false>(block_split_lds, lane, y, [&](int j, u32x4 (&b)[P]) {
#pragma unroll
            for (int piece = 0; piece < P; ++piece) b[piece] = a_frag[j][piece];
        });
        layernorm(y, g1, be1);
        // h = relu(b_1 + W_1 y)
        f32x16 h[kLinearMTiles];
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[m][r] = EMPH_LINEAR_VALID(m, r) ? bias_of(b_1, m, r) : 0.f;
        linear_chain<P>(block_split_lds + PACK, lane, h,
                       [&](int j, int e) { return y[j >> 1][8 * (j & 1) + e]; });
        if (act == EMPH_ACT_RELU) {
#pragma unroll
            for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(h[m][r], 0.f);
        }
        // z = LayerNorm2(y + b_2 + W_2 h), accumulated in y's registers
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (EMPH_LINEAR_VALID(m, r)) y[m][r] += bias_of(b_2, m, r);
        linear_chain<P>(block_split_lds + 2 * PACK, lane, y,
                       [&](int j, int e) { return h[j >> 1][8 * (j & 1) + e]; });
        layernorm(y, g2, be2);
        if (live) {
#pragma unroll
            for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (EMPH_LINEAR_VALID(m, r))
                        linear_at(x, ld, 32 * m + 8 * (r >> 2) + (r & 3),
                                  linear_lane_offset(ld, column, half)) = y[m][r];
        }
    }
}

// The three projections of a tile whose operand is split already (`b_frag`), and what
// becomes of them.  IMAGES = false: qk float32 [2 C][ld], v float32 [ld][C]
// (emph_qkv_projection's).  IMAGES = true: Q into qk's first C rows; K and V as the
// stages of bf16 pieces attention_split_kernel<40, PK, PV> reads (split.h,
// SplitImages): K's accumulators are (key, eight consecutive d) already; V is multiplied
// with the operands swapped, so that a lane holds eight keys of one d in the permuted
// order of the image.  `pack_of(part)`: the pack of W_q / W_k / W_v in LDS (called once
// per part, in order, by every wave that runs this).  `vec`: bias [3][C] in LDS.
template <int P, bool IMAGES, int PK, int PV, typename PackOf>
__device__ __forceinline__ void qkv_parts(const u32x4 (&b_frag)[kLinearSteps][P], const Tile& span,
                                          bool live, int64_t column, int64_t ld,
                                          float* __restrict__ qk, float* __restrict__ v,
                                          unsigned char* __restrict__ images, const float* vec,
                                          int lane, PackOf pack_of) {
    constexpr int C = kLinearChannels;
    constexpr int D = 40, HEADS = 2;
    typedef SplitImages<D, PK, PV> Images;
    const int col = lane & 31;
    const int half = lane >> 5;
    unsigned char* stage = nullptr;       // of head 0
    int key0 = 0;
    if (IMAGES) {
        const int slot = (span.offset >> 6) + span.segment + (span.first >> 6);
        stage = images + static_cast<int64_t>(slot) * HEADS * Images::kStageBytes;
        key0 = span.first & 63;
    }
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const bool swapped = IMAGES && part == 2;
        f32x16 acc[kLinearMTiles];
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[m][r] = swapped ? vec[part * C + min(32 * m + col, C - 1)]
                            : EMPH_LINEAR_VALID(m, r)
                                ? vec[part * C + 32 * m + 8 * (r >> 2) + 4 * half + (r & 3)]
                                : 0.f;
        const unsigned char* pack = pack_of(part);
        auto fragment = [&](int j, u32x4 (&b)[P]) {
#pragma unroll
            for (int piece = 0; piece < P; ++piece) b[piece] = b_frag[j][piece];
        };
        if (swapped) linear_gemm<P, true>(pack, lane, acc, fragment);
        else linear_gemm<P, false>(pack, lane, acc, fragment);
        if (!IMAGES || part == 0) {
            if (!live) continue;
#pragma unroll
            for (int m = 0; m < kLinearMTiles; ++m) {
                if (part < 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (EMPH_LINEAR_VALID(m, r))
                            linear_at(qk, ld, part * C + 32 * m + 8 * (r >> 2) + (r & 3),
                                      linear_lane_offset(ld, column, half)) = acc[m][r];
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (32 * m + 8 * b < C)
                            *reinterpret_cast<f32x4*>(v + column * C + 32 * m + 8 * b + 4 * half) =
                                f32x4{acc[m][4 * b], acc[m][4 * b + 1], acc[m][4 * b + 2],
                                      acc[m][4 * b + 3]};
                }
            }
        } else if (part == 1) {
            // K image: [d / 8][key][8 d], the lane's four consecutive d of every octet
#pragma unroll
            for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (32 * m + 8 * b >= C) continue;
                    constexpr int OCTETS = D / 8;
                    const int head = (4 * m + b) / OCTETS, octet = (4 * m + b) % OCTETS;
                    uint32_t low[PK], high[PK];
                    split_pair<PK>(live ? acc[m][4 * b] : 0.f, live ? acc[m][4 * b + 1] : 0.f, low);
                    split_pair<PK>(live ? acc[m][4 * b + 2] : 0.f, live ? acc[m][4 * b + 3] : 0.f,
                                   high);
#pragma unroll
                    for (int piece = 0; piece < PK; ++piece)
                        *reinterpret_cast<u32x2*>(stage + head * Images::kStageBytes +
                                                  Images::key_piece(piece) +
                                                  (octet * kSplitStage + key0 + col) * 16 + 8 * half) =
                            u32x2{low[piece], high[piece]};
                }
        } else {
            // V image: [key / 8][d][8 keys]; registers 8 G .. 8 G + 7 are the eight
            // keys of chunk 2 G + half in the image's order
#pragma unroll
            for (int m = 0; m < kLinearMTiles; ++m) {
                const int c = 32 * m + col;
                if (c >= C) continue;
                const int head = c / D, d = c % D;
#pragma unroll
                for (int group = 0; group < 2; ++group) {
                    float keys[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        keys[e] = span.first + 16 * group + 8 * (e >> 2) + 4 * half + (e & 3) <
                                          span.count
                                      ? acc[m][8 * group + e]
                                      : 0.f;
                    u32x4 parts[PV];
                    split_eight<PV>(keys, parts);
#pragma unroll
                    for (int piece = 0; piece < PV; ++piece)
                        *reinterpret_cast<u32x4*>(
                            stage + head * Images::kStageBytes + Images::value_piece(piece) +
                            (((key0 >> 3) + 2 * group + half) * Images::kRows + d) * 16) = parts[piece];
                }
            }
        }
    }
    if (IMAGES && key0 == 0) {
        // the tile that opens a stage writes what no projection produces: K's octets
        // from D / 8 on (ones at d = D in piece 0), V's row of ones and row of zeros -
        // and, when the segment ends inside the first half, the zeros of the second
        constexpr int PAD_OCTETS = Images::kOctets - D / 8;
        constexpr int K_FILL = HEADS * PK * PAD_OCTETS * kSplitStage;
        constexpr int V_FILL = HEADS * PV * (kSplitStage / 8) * 2;
        for (int index = lane; index < K_FILL + V_FILL; index += 64) {
            int head, byte;
            u32x4 fill = {0u, 0u, 0u, 0u};
            if (index < K_FILL) {
                const int key = index % kSplitStage, octet = D / 8 + index / kSplitStage % PAD_OCTETS;
                const int piece = index / kSplitStage / PAD_OCTETS % PK;
                head = index / kSplitStage / PAD_OCTETS / PK;
                if (octet == D / 8 && piece == 0) fill[0] = 0x3f80u;
                byte = Images::key_piece(piece) + (octet * kSplitStage + key) * 16;
            } else {
                const int rest = index - K_FILL;
                const int chunk = rest % (kSplitStage / 8), row = D + rest / (kSplitStage / 8) % 2;
                const int piece = rest / (kSplitStage / 8) / 2 % PV;
                head = rest / (kSplitStage / 8) / 2 / PV;
                if (row == D && piece == 0)
                    fill = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                byte = Images::value_piece(piece) + (chunk * Images::kRows + row) * 16;
            }
            *reinterpret_cast<u32x4*>(stage + head * Images::kStageBytes + byte) = fill;
        }
        if (span.first + 32 >= span.count) {          // wave-uniform
            constexpr int K_ZERO = HEADS * PK * (D / 8) * 32;
            constexpr int V_ZERO = HEADS * PV * 4 * D;
            const u32x4 zero = {0u, 0u, 0u, 0u};
            for (int index = lane; index < K_ZERO + V_ZERO; index += 64) {
                int head, byte;
                if (index < K_ZERO) {
                    const int key = 32 + index % 32, octet = index / 32 % (D / 8);
                    const int piece = index / 32 / (D / 8) % PK;
                    head = index / 32 / (D / 8) / PK;
                    byte = Images::key_piece(piece) + (octet * kSplitStage + key) * 16;
                } else {
                    const int rest = index - K_ZERO;
                    const int d = rest % D, chunk = 4 + rest / D % 4;
                    const int piece = rest / D / 4 % PV;
                    head = rest / D / 4 / PV;
                    byte = Images::value_piece(piece) + (chunk * Images::kRows + d) * 16;
                }
                *reinterpret_cast<u32x4*>(stage + head * Images::kStageBytes + byte) = zero;
            }
        }
    }
}

// IMAGES = false: qk float32 [2 C][ld], v float32 [ld][C] (emph_qkv_projection's).
// IMAGES = true: Q into qk's first C rows; K and V as the stages of bf16 pieces
// attention_split_kernel<40, PK, PV> reads (split.h, SplitImages): K's accumulators are
// (key, eight consecutive d) already; V is multiplied with the operands swapped, so that
// a lane holds eight keys of one d in the permuted order of the image.
template <int P, bool IMAGES, int PK, int PV>
__global__ __launch_bounds__(kLinearThreads) void qkv_split_kernel(
    const float* __restrict__ x, int64_t ld, float* __restrict__ qk, float* __restrict__ v,
    unsigned char* __restrict__ images,
    const unsigned char* __restrict__ packs /* q | k | v */, const float* __restrict__ bias /* [3][C] */,
    const int32_t* __restrict__ tiles, int n_tiles) {
    constexpr int C = kLinearChannels;
    constexpr int PACK = linear_pack_bytes(P);
    extern __shared__ __align__(16) unsigned char block_split_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;
    const int half = lane >> 5;
    LinearWeights<P, 3 * C> weights(block_split_lds, packs, bias);
    const float* vec = weights.vec;

    auto request = [&](int tile, float (&operand)[kLinearSteps][8]) {
        const Tile span = load_tile(tiles, tile);
        const uint32_t at = linear_lane_offset(
            ld, span.offset + min(span.first + col, span.count - 1), half);
#pragma unroll
        for (int j = 0; j < kLinearSteps; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                operand[j][e] = linear_at(x, ld, 16 * j + 8 * (e >> 2) + (e & 3), at);
    };
    const int stride = gridDim.x * kLinearWaves;
    int tile = blockIdx.x * kLinearWaves + wave;
    float operand[kLinearSteps][8];
    if (tile < n_tiles) request(tile, operand);
    weights.landed();
    for (; tile < n_tiles; tile += stride) {
        const Tile span = load_tile(tiles, tile);
        const bool live = span.first + col < span.count;
        const int64_t column = span.offset + min(span.first + col, span.count - 1);
        // (the operand is split once for the three projections; the next tile's is
        // requested into its registers)
        u32x4 b_frag[kLinearSteps][P];
#pragma unroll
        for (int j = 0; j < kLinearSteps; ++j) split_eight<P>(operand[j], b_frag[j]);
        if (tile + stride < n_tiles) request(tile + stride, operand);

        qkv_parts<P, IMAGES, PK, PV>(b_frag, span, live, column, ld, qk, v, images, vec, lane,
                                     [&](int part) { return block_split_lds + part * PACK; });
    }
}

// transformer_block_split_kernel and the NEXT layer's qkv_split_kernel as one launch
// (what block.hip's transformer_block_kernel<.., true> is to the fp32 path): the layer's
// output is split in the registers it is normalised in and multiplied by W_q, W_k, W_v
// at once - x is written for the residual of the next block but not read again, and a
// layer is one position-wise launch instead of two.  Same arithmetic, instruction for
// instruction, as the two kernels in a row: the results are theirs bit for bit.
//
// Six packs (45 KB each with three pieces) do not fit in LDS, so they STREAM: a ring of
// three slots, the pack of GEMM n + 2 requested (LDS-DMA, every wave its quarter) when
// GEMM n starts - behind the barrier that says every wave has left GEMM n - 1, whose slot
// it takes - and waited for two GEMMs later.  All waves of a workgroup run the same
// number of rounds (a wave without a tile in the last round takes part in the ring only).
template <int P>
struct PackRing {
    static constexpr int PACK = linear_pack_bytes(P);
    unsigned char* lds;
    const unsigned char* block_packs;
    const unsigned char* qkv_packs;
    int total;        // GEMMs of this workgroup (6 per round)
    int next;         // the GEMM acquire() hands out next
    int lane, wave;
    __device__ __forceinline__ void request(int n) {
        const int g = n % 6;
        const unsigned char* source =
            g < 3 ? block_packs + g * PACK : qkv_packs + (g - 3) * PACK;
        unsigned char* slot = lds + (n % 3) * PACK;
        for (int base = wave * 64; base < PACK / 16; base += kLinearThreads)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(source + 16 * (base + lane)),
                (__attribute__((address_space(3))) void*)(slot + 16 * base), 16, 0, 0);
    }
    // the pack of the next GEMM, landed and visible to every wave
    __device__ __forceinline__ const unsigned char* acquire() {
        const int n = next++;
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): this wave's quarters
        __syncthreads();
        if (n + 2 < total) request(n + 2);
        return lds + (n % 3) * PACK;
    }
};

template <int P, bool IMAGES, int PK, int PV>
__global__ __launch_bounds__(kLinearThreads) void transformer_block_qkv_split_kernel(
    const float* __restrict__ attended, float* __restrict__ x, int64_t ld,
    const unsigned char* __restrict__ block_packs /* out | linear1 | linear2 */,
    const unsigned char* __restrict__ qkv_packs /* q | k | v of the next layer */,
    const float* __restrict__ vectors /* b_o g1 be1 b_1 b_2 g2 be2 */,
    const float* __restrict__ qkv_bias /* [3][C] */, float eps, int act,
    float* __restrict__ qk, float* __restrict__ v, unsigned char* __restrict__ images,
    const int32_t* __restrict__ tiles, int n_tiles) {
    constexpr int C = kLinearChannels;
    constexpr int PACK = linear_pack_bytes(P);
    constexpr int FLOATS = 10 * C;
    constexpr int HELD = (FLOATS + kLinearThreads - 1) / kLinearThreads;
    extern __shared__ __align__(16) unsigned char block_split_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;
    const int half = lane >> 5;
    float* vec = reinterpret_cast<float*>(block_split_lds + 3 * PACK);
    const float* b_o = vec;
    const float* g1 = vec + C;
    const float* be1 = vec + 2 * C;
    const float* b_1 = vec + 3 * C;
    const float* b_2 = vec + 4 * C;
    const float* g2 = vec + 5 * C;
    const float* be2 = vec + 6 * C;
    const float* b_qkv = vec + 7 * C;

    const int stride = gridDim.x * kLinearWaves;
    const int first = blockIdx.x * kLinearWaves;
    const int rounds = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    // (ahead of the LDS-DMA requests: what waits for these must not wait for those)
    float held[HELD];
#pragma unroll
    for (int i = 0; i < HELD; ++i) {
        const int index = min(static_cast<int>(threadIdx.x) + i * kLinearThreads, FLOATS - 1);
        held[i] = index < 7 * C ? vectors[index] : qkv_bias[index - 7 * C];
    }
    PackRing<P> ring{block_split_lds, block_packs, qkv_packs, 6 * rounds, 0, lane, wave};
    if (ring.total > 0) ring.request(0);
    if (ring.total > 1) ring.request(1);

    auto bias_of = [&](const float* bias, int m, int r) {
        return bias[32 * m + 8 * (r >> 2) + 4 * half + (r & 3)];
    };
    auto request = [&](int tile, float (&operand)[kLinearSteps][8], f32x16 (&residual)[kLinearMTiles]) {
        const Tile span = load_tile(tiles, tile);
        const uint32_t at = linear_lane_offset(
            ld, span.offset + min(span.first + col, span.count - 1), half);
#pragma unroll
        for (int j = 0; j < kLinearSteps; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                operand[j][e] = linear_at(attended, ld, 16 * j + 8 * (e >> 2) + (e & 3), at);
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                residual[m][r] = EMPH_LINEAR_VALID(m, r)
                                     ? linear_at(x, ld, 32 * m + 8 * (r >> 2) + (r & 3), at)
                                     : 0.f;
    };

    int tile = first + wave;
    float operand[kLinearSteps][8];
    f32x16 residual[kLinearMTiles];
    if (tile < n_tiles) request(tile, operand, residual);
#pragma unroll
    for (int i = 0; i < HELD; ++i)
        if (static_cast<int>(threadIdx.x) + i * kLinearThreads < FLOATS)
            vec[threadIdx.x + i * kLinearThreads] = held[i];
    for (int round = 0; round < rounds; ++round, tile += stride) {
        if (tile >= n_tiles) {          // wave-uniform: the ring needs every wave
#pragma unroll 1
            for (int g = 0; g < 6; ++g) ring.acquire();
            continue;
        }
        const Tile span = load_tile(tiles, tile);
        const bool live = span.first + col < span.count;
        const int64_t column = span.offset + min(span.first + col, span.count - 1);
        u32x4 a_frag[kLinearSteps][P];
#pragma unroll
        for (int j = 0; j < kLinearSteps; ++j) split_eight<P>(operand[j], a_frag[j]);
        f32x16 y[kLinearMTiles];
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[m][r] = residual[m][r];
        const unsigned char* pack = ring.acquire();
        // (behind the barrier: the vectors are in LDS; the next tile's inputs have a whole
        // GEMM to arrive in before the next barrier waits for them)
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                y[m][r] = EMPH_LINEAR_VALID(m, r) ? y[m][r] + bias_of(b_o, m, r) : 0.f;
        if (tile + stride < n_tiles) request(tile + stride, operand, residual);
        // y = LayerNorm1(x + b_o + W_o a)
        linear_gemm<P, false>(pack, lane, y, [&](int j, u32x4 (&b)[P]) {
#pragma unroll
            for (int piece = 0; piece < P; ++piece) b[piece] = a_frag[j][piece];
        });
        linear_layernorm(y, g1, be1, eps, half);
        // h = relu(b_1 + W_1 y)
        f32x16 h[kLinearMTiles];
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[m][r] = EMPH_LINEAR_VALID(m, r) ? bias_of(b_1, m, r) : 0.f;
        pack = ring.acquire();
        linear_chain<P>(pack, lane, h, [&](int j, int e) { return y[j >> 1][8 * (j & 1) + e]; });
        if (act == EMPH_ACT_RELU) {
#pragma unroll
            for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(h[m][r], 0.f);
        }
        // z = LayerNorm2(y + b_2 + W_2 h), accumulated in y's registers
#pragma unroll
        for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (EMPH_LINEAR_VALID(m, r)) y[m][r] += bias_of(b_2, m, r);
        pack = ring.acquire();
        linear_chain<P>(pack, lane, y, [&](int j, int e) { return h[j >> 1][8 * (j & 1) + e]; });
        linear_layernorm(y, g2, be2, eps, half);
        if (live) {
#pragma unroll
            for (int m = 0; m < kLinearMTiles; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (EMPH_LINEAR_VALID(m, r))
                        linear_at(x, ld, 32 * m + 8 * (r >> 2) + (r & 3),
                                  linear_lane_offset(ld, column, half)) = y[m][r];
        }
        // the next layer's projections of what was just stored: its registers ARE the
        // operand (k-step j = channels 16 j + 8 (e / 4) + 4 half + e % 4)
        u32x4 b_frag[kLinearSteps][P];
#pragma unroll
        for (int j = 0; j < kLinearSteps; ++j) {
            float values[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) values[e] = y[j >> 1][8 * (j & 1) + e];
            split_eight<P>(values, b_frag[j]);
        }
        qkv_parts<P, IMAGES, PK, PV>(b_frag, span, live, column, ld, qk, v, images, b_qkv, lane,
                                     [&](int) { return ring.acquire(); });
    }
}

}  // namespace emph

using namespace emph;

namespace {

// the pieces of split_pair (split.h) on the host: two, rounded to nearest; three, truncated
void linear_host_pieces(float value, int pieces, uint16_t (&out)[3]) {
    for (int piece = 0; piece < pieces; ++piece) {
        uint32_t bits;
        memcpy(&bits, &value, 4);
        if (pieces == 2) bits += 0x7fffu + ((bits >> 16) & 1u);
        bits &= 0xffff0000u;
        out[piece] = static_cast<uint16_t>(bits >> 16);
        float kept;
        memcpy(&kept, &bits, 4);
        value -= kept;
    }
}

}  // namespace

extern "C" {

int64_t emph_linear_split_pack_size(int32_t pieces) {
    return pieces == 2 || pieces == 3 ? linear_pack_bytes(pieces) : 0;
}

// weight float32 [80][80] (HOST; out x in, as nn.Linear stores it) -> [k-step][m-tile]
// [piece][lane][8 bf16], lane = (output channel 32 m + lane % 32; input channels 16 j +
// 8 (e / 4) + 4 (lane / 32) + e % 4 for e = 0 .. 7: the order in which the result of
// one GEMM of the chain lies in the registers of the next); rows 80 .. 95 are zeros.
int emph_linear_split_pack(const float* host_weight, int32_t pieces, void* host_pack) {
    EMPH_REQUIRE(host_weight && host_pack, EMPH_EINVAL, "emph_linear_split_pack: null pointer");
    EMPH_REQUIRE(pieces == 2 || pieces == 3, EMPH_ERANGE, "emph_linear_split_pack: %d pieces",
                 pieces);
    uint16_t* out = static_cast<uint16_t*>(host_pack);
    for (int j = 0; j < kLinearSteps; ++j)
        for (int m = 0; m < kLinearMTiles; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int row = 32 * m + (lane & 31);
                    const int channel = 16 * j + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
                    const float weight =
                        row < kLinearChannels ? host_weight[row * kLinearChannels + channel] : 0.f;
                    uint16_t parts[3];
                    linear_host_pieces(weight, pieces, parts);
                    const size_t base = (static_cast<size_t>(j) * kLinearMTiles + m) * pieces * 512;
                    for (int piece = 0; piece < pieces; ++piece)
                        out[base + piece * 512 + lane * 8 + e] = parts[piece];
                }
    return EMPH_OK;
}

// emph_transformer_block (block.hip) with the three GEMMs on the bf16 pipe: packs =
// out_proj | linear1 | linear2 (emph_linear_split_pack each, `pieces` pieces), vectors =
// b_o g1 be1 b_1 b_2 g2 be2.
int emph_transformer_block_split(const float* attended, float* x, int64_t ld, int32_t channels,
                                 const void* packs, int32_t pieces, const float* vectors,
                                 float eps, int32_t activation, const int32_t* tiles,
                                 int32_t n_tiles, int32_t tile_n, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(attended && x && packs && vectors && tiles, EMPH_EINVAL,
                 "emph_transformer_block_split: null pointer");
    EMPH_REQUIRE(channels == kLinearChannels && tile_n == kLinearTile, EMPH_ERANGE,
                 "emph_transformer_block_split: %d channels, tiles of %d (built for 80 and 32)",
                 channels, tile_n);
    EMPH_REQUIRE(pieces == 2 || pieces == 3, EMPH_ERANGE,
                 "emph_transformer_block_split: %d pieces (2 or 3)", pieces);
    EMPH_REQUIRE(ld > 0 && ld < (int64_t{1} << 29), EMPH_ERANGE,
                 "emph_transformer_block_split: ld %lld outside the 32-bit lane offsets",
                 static_cast<long long>(ld));
    EMPH_REQUIRE(activation == EMPH_ACT_RELU || activation == EMPH_ACT_NONE, EMPH_ERANGE,
                 "emph_transformer_block_split: activation %d", activation);
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(packs) & 15) == 0, EMPH_EINVAL,
                 "emph_transformer_block_split: the packs must be 16-byte aligned");
    const size_t lds = 3 * linear_pack_bytes(pieces) + 7 * kLinearChannels * sizeof(float);
    const unsigned groups =
        static_cast<unsigned>(min((n_tiles + kLinearWaves - 1) / kLinearWaves, 256));
#define EMPH_BLOCK_SPLIT(P)                                                                    \
    do {                                                                                       \
        auto kernel = transformer_block_split_kernel<P>;                                       \
        static LdsReservation reserved;                                                        \
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,     \
                                     "emph_transformer_block_split"))                          \
            return status;                                                                     \
        EMPH_LAUNCH(kernel, dim3(groups), dim3(kLinearThreads), lds,                           \
                    static_cast<hipStream_t>(stream), attended, x, ld,                         \
                    static_cast<const unsigned char*>(packs), vectors, eps, activation, tiles, \
                    n_tiles);                                                                  \
    } while (0)
    if (pieces == 2) EMPH_BLOCK_SPLIT(2); else EMPH_BLOCK_SPLIT(3);
#undef EMPH_BLOCK_SPLIT
    return check_launch("emph_transformer_block_split");
}

#define EMPH_QKV_SPLIT(P, IMAGES, PK, PV, what)                                                \
    do {                                                                                       \
        auto kernel = qkv_split_kernel<P, IMAGES, PK, PV>;                                     \
        static LdsReservation reserved;                                                        \
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,     \
                                     what))                                                    \
            return status;                                                                     \
        EMPH_LAUNCH(kernel, dim3(groups), dim3(kLinearThreads), lds,                           \
                    static_cast<hipStream_t>(stream), x, ld, qk, v,                            \
                    static_cast<unsigned char*>(images),                                       \
                    static_cast<const unsigned char*>(packs), bias, tiles, n_tiles);           \
    } while (0)

// emph_qkv_projection (block.hip) on the bf16 pipe: packs = q | k | v rows of
// in_proj_weight (emph_linear_split_pack each), bias [3][80].
int emph_qkv_projection_split(const float* x, int64_t ld, float* qk, float* v, int32_t channels,
                              const void* packs, int32_t pieces, const float* bias,
                              const int32_t* tiles, int32_t n_tiles, int32_t tile_n,
                              void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(x && qk && v && packs && bias && tiles, EMPH_EINVAL,
                 "emph_qkv_projection_split: null pointer");
    EMPH_REQUIRE(channels == kLinearChannels && tile_n == kLinearTile, EMPH_ERANGE,
                 "emph_qkv_projection_split: %d channels, tiles of %d (built for 80 and 32)",
                 channels, tile_n);
    EMPH_REQUIRE(pieces == 2 || pieces == 3, EMPH_ERANGE,
                 "emph_qkv_projection_split: %d pieces (2 or 3)", pieces);
    EMPH_REQUIRE(ld > 0 && ld < (int64_t{1} << 29), EMPH_ERANGE,
                 "emph_qkv_projection_split: ld %lld outside the 32-bit lane offsets",
                 static_cast<long long>(ld));
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(packs) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(v) & 15) == 0,
                 EMPH_EINVAL, "emph_qkv_projection_split: packs and v must be 16-byte aligned");
    const size_t lds = 3 * linear_pack_bytes(pieces) + 3 * kLinearChannels * sizeof(float);
    const unsigned groups =
        static_cast<unsigned>(min((n_tiles + kLinearWaves - 1) / kLinearWaves, 256));
    void* images = nullptr;
    if (pieces == 2) EMPH_QKV_SPLIT(2, false, 2, 2, "emph_qkv_projection_split");
    else EMPH_QKV_SPLIT(3, false, 2, 2, "emph_qkv_projection_split");
    return check_launch("emph_qkv_projection_split");
}

// ... with K and V written as the images of emph_split_kv (attention_split.hip): Q into
// qk's first 80 rows (the other 80 are not touched), `images` as emph_split_kv_bytes
// sizes them for `attention_pieces` (2, 3 or 32), two heads of 40.  EVERY stage of every
// segment of the tile table is written whole; emph_attention_split reads them as they are.
int emph_qkv_projection_split_images(const float* x, int64_t ld, float* qk, void* images,
                                     int32_t channels, int32_t heads, const void* packs,
                                     int32_t pieces, int32_t attention_pieces, const float* bias,
                                     const int32_t* tiles, int32_t n_tiles, int32_t tile_n,
                                     void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(x && qk && images && packs && bias && tiles, EMPH_EINVAL,
                 "emph_qkv_projection_split_images: null pointer");
    EMPH_REQUIRE(channels == kLinearChannels && tile_n == kLinearTile && heads == 2, EMPH_ERANGE,
                 "emph_qkv_projection_split_images: %d channels, %d heads, tiles of %d (built "
                 "for 80, 2 and 32)", channels, heads, tile_n);
    EMPH_REQUIRE(pieces == 2 || pieces == 3, EMPH_ERANGE,
                 "emph_qkv_projection_split_images: %d pieces (2 or 3)", pieces);
    EMPH_REQUIRE(ld > 0 && ld < (int64_t{1} << 29), EMPH_ERANGE,
                 "emph_qkv_projection_split_images: ld %lld outside the 32-bit lane offsets",
                 static_cast<long long>(ld));
    EMPH_REQUIRE(attention_pieces == 2 || attention_pieces == 3 || attention_pieces == 32,
                 EMPH_ERANGE, "emph_qkv_projection_split_images: attention pieces %d (2, 3 or 32)",
                 attention_pieces);
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(packs) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(images) & 15) == 0,
                 EMPH_EINVAL,
                 "emph_qkv_projection_split_images: packs and images must be 16-byte aligned");
    const size_t lds = 3 * linear_pack_bytes(pieces) + 3 * kLinearChannels * sizeof(float);
    const unsigned groups =
        static_cast<unsigned>(min((n_tiles + kLinearWaves - 1) / kLinearWaves, 256));
    float* v = nullptr;
    const char* what = "emph_qkv_projection_split_images";
    if (pieces == 2) {
        if (attention_pieces == 2) EMPH_QKV_SPLIT(2, true, 2, 2, what);
        else if (attention_pieces == 3) EMPH_QKV_SPLIT(2, true, 3, 3, what);
        else EMPH_QKV_SPLIT(2, true, 3, 2, what);
    } else {
        if (attention_pieces == 2) EMPH_QKV_SPLIT(3, true, 2, 2, what);
        else if (attention_pieces == 3) EMPH_QKV_SPLIT(3, true, 3, 3, what);
        else EMPH_QKV_SPLIT(3, true, 3, 2, what);
    }
    return check_launch(what);
}

// emph_transformer_block_split and the NEXT layer's emph_qkv_projection_split (IMAGES:
// emph_qkv_projection_split_images) as ONE launch; bit for bit the two in a row.
// block_packs / vectors: as emph_transformer_block_split; qkv_packs / qkv_bias: the next
// layer's, as emph_qkv_projection_split; images != NULL: Q into qk's first 80 rows and
// the K / V images for `attention_pieces` (v is not used); images == NULL: qk and v.
int emph_transformer_block_qkv_split(const float* attended, float* x, int64_t ld,
                                     int32_t channels, int32_t heads, const void* block_packs,
                                     const void* qkv_packs, int32_t pieces,
                                     int32_t attention_pieces, const float* vectors,
                                     const float* qkv_bias, float eps, int32_t activation,
                                     const int32_t* tiles, int32_t n_tiles, int32_t tile_n,
                                     float* qk, float* v, void* images, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    const char* what = "emph_transformer_block_qkv_split";
    EMPH_REQUIRE(attended && x && block_packs && qkv_packs && vectors && qkv_bias && tiles && qk &&
                     (images || v),
                 EMPH_EINVAL, "%s: null pointer", what);
    EMPH_REQUIRE(channels == kLinearChannels && tile_n == kLinearTile && heads == 2, EMPH_ERANGE,
                 "%s: %d channels, %d heads, tiles of %d (built for 80, 2 and 32)", what, channels,
                 heads, tile_n);
    EMPH_REQUIRE(pieces == 2 || pieces == 3, EMPH_ERANGE, "%s: %d pieces (2 or 3)", what, pieces);
    EMPH_REQUIRE(ld > 0 && ld < (int64_t{1} << 29), EMPH_ERANGE,
                 "%s: ld %lld outside the 32-bit lane offsets", what, static_cast<long long>(ld));
    EMPH_REQUIRE(activation == EMPH_ACT_RELU || activation == EMPH_ACT_NONE, EMPH_ERANGE,
                 "%s: activation %d", what, activation);
    EMPH_REQUIRE(!images || attention_pieces == 2 || attention_pieces == 3 || attention_pieces == 32,
                 EMPH_ERANGE, "%s: attention pieces %d (2, 3 or 32)", what, attention_pieces);
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(block_packs) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(qkv_packs) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(images) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(v) & 15) == 0,
                 EMPH_EINVAL, "%s: packs, images and v must be 16-byte aligned", what);
    const size_t lds = 3 * linear_pack_bytes(pieces) + 10 * kLinearChannels * sizeof(float);
    const unsigned groups =
        static_cast<unsigned>(min((n_tiles + kLinearWaves - 1) / kLinearWaves, 256));
#define EMPH_BLOCK_QKV_SPLIT(P, IMAGES, PK, PV)                                                \
    do {                                                                                       \
        auto kernel = transformer_block_qkv_split_kernel<P, IMAGES, PK, PV>;                   \
        static LdsReservation reserved;                                                        \
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,     \
                                     what))                                                    \
            return status;                                                                     \
        EMPH_LAUNCH(kernel, dim3(groups), dim3(kLinearThreads), lds,                           \
                    static_cast<hipStream_t>(stream), attended, x, ld,                         \
                    static_cast<const unsigned char*>(block_packs),                            \
                    static_cast<const unsigned char*>(qkv_packs), vectors, qkv_bias, eps,      \
                    activation, qk, v, static_cast<unsigned char*>(images), tiles, n_tiles);   \
    } while (0)
    if (images == nullptr) {
        if (pieces == 2) EMPH_BLOCK_QKV_SPLIT(2, false, 2, 2);
        else EMPH_BLOCK_QKV_SPLIT(3, false, 2, 2);
    } else if (pieces == 2) {
        if (attention_pieces == 2) EMPH_BLOCK_QKV_SPLIT(2, true, 2, 2);
        else if (attention_pieces == 3) EMPH_BLOCK_QKV_SPLIT(2, true, 3, 3);
        else EMPH_BLOCK_QKV_SPLIT(2, true, 3, 2);
    } else {
        if (attention_pieces == 2) EMPH_BLOCK_QKV_SPLIT(3, true, 2, 2);
        else if (attention_pieces == 3) EMPH_BLOCK_QKV_SPLIT(3, true, 3, 3);
        else EMPH_BLOCK_QKV_SPLIT(3, true, 3, 2);
    }
#undef EMPH_BLOCK_QKV_SPLIT
    return check_launch(what);
}

#undef EMPH_QKV_SPLIT

}  // extern "C"
