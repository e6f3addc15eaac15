// The position-wise kernels of block_split.hip - the block of a post-LN Transformer
// encoder layer, the Q / K / V projections in front of the attention, or both in one
// launch - rebuilt around tiles of SIXTEEN positions on v_mfma_f32_16x16x32_bf16, so that
// a wave needs 160-230 registers instead of 450 and TWO waves share a SIMD.  The chain of a
// tile (GEMM, LayerNorm, split, GEMM ...) is serial by construction: at one wave per SIMD
// the matrix pipe idled through every vector phase, every LDS fragment read and every trip
// to memory (block_split.hip's kernels: pipe 29-37 % busy, 51 us per layer whether as one
// launch or two); the second wave fills part of that (42.7 us per layer for block + next
// projections in ONE launch, profiles/r6_transformer_bf16x3_kernel_stats_1stream.csv; what
// still bounds it is the chain, EXPERIMENTS.md round 6).
//   emph_position_wise_split     y = LayerNorm1(x + W_o a + b_o)
//                                x <- LayerNorm2(y + W_2 relu(W_1 y + b_1) + b_2)
//                                and / or  Q | K | V of the NEXT layer, as fp32 rows or
//                                as Q + the split images attention_split_kernel stages
// (out_proj, residual, norm1, linear1, activation, linear2, residual, norm2 and in_proj of
// nn.TransformerEncoderLayer, emphases/model/layers/transformer.py:18-23.)
//
// Layout.  A wave owns 16 positions and ALL 80 channels.  Lane l = (position l % 16,
// group g = l / 16); the accumulators of m-tile m (16 output channels, five of them: no
// padded rows) hold channels 16 m + 4 g + i, i = 0 .. 3 - which IS the B operand of the
// next GEMM once split, if that GEMM's k-step j (32 input channels, three of them: the
// last one half zeros) multiplies, in group g, channels 32 j + 16 (e / 4) + 4 g + e % 4
// (e = 0 .. 7): the weights are packed in that order (emph_linear_split_pack16), the first
// operand of a kernel is loaded in it.  90 (45) MFMAs of 16 cycles per GEMM and 16
// positions with six (three) products per term: the matrix work per position of the
// 32-position kernels.
//
// Weights.  A pack is 45 (30) KB.  Three GEMMs' packs sit in LDS for the whole launch;
// the six of the fused launch STREAM through a ring of three slots (the pack of GEMM n + 2
// requested by LDS-DMA when GEMM n starts, behind the barrier that says every wave has
// left GEMM n - 1, whose slot it takes).  Eight waves per workgroup share the ring.
#include <stdlib.h>
#include <string.h>

#include "split.h"

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kP16Channels = 80;
constexpr int kP16Steps = 3;                 // k-steps of 32 input channels (80 -> 96)
constexpr int kP16MTiles = 5;                // m-tiles of 16 output channels
constexpr int kP16Tile = 16;                 // positions per wave and round
constexpr int kP16Waves = 8;
constexpr int kP16Threads = 64 * kP16Waves;
constexpr int p16_pack_bytes(int pieces) { return kP16Steps * kP16MTiles * pieces * 1024; }

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Channel-major rows through BUFFER instructions: the array is one buffer resource (four
// scalar registers), a lane's column and its group's four channels ONE 32-bit byte offset in
// a vector register, the row of an access a scalar offset (channel x ld x 4, scalar
// arithmetic) - no vector instruction per access.  With plain pointers every one of a tile's
// 150 accesses cost a 64-bit vector add and the 80 row addresses spilled scalar registers
// into vector lanes (v_writelane / v_readlane): 350 of the kernel's 1 840 vector
// instructions per tile, in a kernel bound by vector issue.  (The entry point keeps every
// byte offset below 2^32.)
typedef __amdgpu_buffer_rsrc_t p16_rows;
__device__ __forceinline__ p16_rows p16_buffer(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ uint32_t p16_lane_offset(uint32_t ld4, int64_t column, int group) {
    return 4u * static_cast<uint32_t>(column) + static_cast<uint32_t>(4 * group) * ld4;
}
__device__ __forceinline__ float p16_load(p16_rows rows, uint32_t ld4, int channel,
                                          uint32_t lane_offset) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
        rows, lane_offset, static_cast<uint32_t>(channel) * ld4, 0));
}
__device__ __forceinline__ void p16_store(p16_rows rows, uint32_t ld4, int channel,
                                          uint32_t lane_offset, float value) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(value), rows, lane_offset,
                                          static_cast<uint32_t>(channel) * ld4, 0);
}

// acc[m] += W (pack, in LDS) x B.  `fragment(j, b)`: the B operand of k-step j, split.
// SWAP: the operands trade places (the result is transposed: lane = (output channel
// 16 m + l % 16, group), registers = positions 4 g + i).  Per k-step the five m-tiles'
// fragments are read, then the products run small to large with the m-tiles innermost:
// five independent accumulators between two MFMAs on the same one.
template <int P, bool SWAP, typename Fragment>
__device__ __forceinline__ void p16_gemm(const unsigned char* pack, int lane,
                                         f32x4 (&acc)[kP16MTiles], Fragment fragment) {
#pragma unroll
    for (int j = 0; j < kP16Steps; ++j) {
        u32x4 a[kP16MTiles][P];
#pragma unroll
        for (int m = 0; m < kP16MTiles; ++m)
#pragma unroll
            for (int piece = 0; piece < P; ++piece)
                a[m][piece] = *reinterpret_cast<const u32x4*>(
                    pack + ((j * kP16MTiles + m) * P + piece) * 1024 + 16 * lane);
        u32x4 b[P];
        fragment(j, b);
#pragma unroll
        for (int order = P - 1; order >= 0; --order)
#pragma unroll
            for (int i = 0; i <= order; ++i)
#pragma unroll
                for (int m = 0; m < kP16MTiles; ++m)
                    acc[m] = SWAP ? mfma16(b[order - i], a[m][i], acc[m])
                                  : mfma16(a[m][i], b[order - i], acc[m]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the B operand of k-step j out of the accumulators of the GEMM in front
template <int P>
__device__ __forceinline__ void p16_fragment(const f32x4 (&source)[kP16MTiles], int j, u32x4 (&b)[P]) {
    float values[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        values[e] = source[2 * j][e];
        values[4 + e] = 2 * j + 1 < kP16MTiles ? source[min(2 * j + 1, kP16MTiles - 1)][e] : 0.f;
    }
    split_eight<P>(values, b);
}

// v <- LayerNorm(v) over the channels of each position (two passes, as
// torch.nn.LayerNorm): 20 values in the lane, 20 in each of the three other groups' lanes
__device__ __forceinline__ void p16_layernorm(f32x4 (&v)[kP16MTiles], const float* gamma,
                                              const float* beta, float eps, int group) {
    constexpr int C = kP16Channels;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < kP16MTiles; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += v[m][i];
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum / static_cast<float>(C);
    float square = 0.f;
#pragma unroll
    for (int m = 0; m < kP16MTiles; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[m][i] -= mean;
            square = fmaf(v[m][i], v[m][i], square);
        }
    square += __shfl_xor(square, 16);
    square += __shfl_xor(square, 32);
    const float rstd = 1.f / sqrtf(square / static_cast<float>(C) + eps);
#pragma unroll
    for (int m = 0; m < kP16MTiles; ++m) {
        const f32x4 scale = *reinterpret_cast<const f32x4*>(gamma + 16 * m + 4 * group);
        const f32x4 shift = *reinterpret_cast<const f32x4*>(beta + 16 * m + 4 * group);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[m][i] = v[m][i] * rstd * scale[i] + shift[i];
    }
}

// The packs of a launch in LDS.  GEMMS == 3: the three packs, requested once, stay.
// GEMMS == 6: a ring of three slots (see the head of the file).  acquire() hands out
// the pack of the workgroup's next GEMM; every wave calls it the same number of times.
template <int P, int GEMMS>
struct P16Packs {
    static constexpr int PACK = p16_pack_bytes(P);
    unsigned char* lds;
    const unsigned char* first;       // packs of GEMMs 0 .. 2 of a round
    const unsigned char* second;      // ... of GEMMs 3 .. 5 (GEMMS == 6)
    int total;                        // GEMMs of this workgroup: GEMMS per round
    int next;
    int lane, wave;
    __device__ __forceinline__ void request(int n) {
        const int g = n % GEMMS;
        const unsigned char* source = g < 3 ? first + g * PACK : second + (g - 3) * PACK;
        unsigned char* slot = lds + (n % 3) * PACK;
        for (int base = wave * 64; base < PACK / 16; base += kP16Threads)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(source + 16 * (base + lane)),
                (__attribute__((address_space(3))) void*)(slot + 16 * base), 16, 0, 0);
    }
    __device__ __forceinline__ void start() {
        for (int n = 0; n < (GEMMS == 3 ? 3 : 2) && n < total; ++n) request(n);
    }
    __device__ __forceinline__ const unsigned char* acquire() {
        const int n = next++;
        if (GEMMS == 3 && n >= 3) return lds + (n % 3) * PACK;     // (resident)
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): this wave's shares
        __syncthreads();
        if (GEMMS != 3 && n + 2 < total) request(n + 2);
        return lds + (n % 3) * PACK;
    }
};

// grid = workgroups of eight waves over the tile table (tiles of 16 positions)
template <int P, bool BLOCK, bool QKV, bool IMAGES, int PK, int PV>
__global__ __launch_bounds__(kP16Threads) void position_wise16_kernel(
    const float* __restrict__ attended, float* __restrict__ x, int64_t ld,
    const unsigned char* __restrict__ block_packs /* out | linear1 | linear2 */,
    const unsigned char* __restrict__ qkv_packs /* q | k | v */,
    const float* __restrict__ vectors /* b_o g1 be1 b_1 b_2 g2 be2 */,
    const float* __restrict__ qkv_bias /* [3][C] */, float eps, int act,
    float* __restrict__ qk, float* __restrict__ v, unsigned char* __restrict__ images,
    const int32_t* __restrict__ tiles, int n_tiles) {
    static_assert(BLOCK || QKV, "something to do");
    constexpr int C = kP16Channels;
    constexpr int GEMMS = (BLOCK ? 3 : 0) + (QKV ? 3 : 0);
    constexpr int PACK = p16_pack_bytes(P);
    constexpr int FLOATS = 10 * C;
    constexpr int HELD = (FLOATS + kP16Threads - 1) / kP16Threads;
    constexpr int D = 40, HEADS = 2;
    typedef SplitImages<D, PK, PV> Images;
    extern __shared__ __align__(16) unsigned char p16_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15;
    const int group = lane >> 4;
    float* vec = reinterpret_cast<float*>(p16_lds + 3 * PACK);
    const float* b_o = vec;
    const float* g1 = vec + C;
    const float* be1 = vec + 2 * C;
    const float* b_1 = vec + 3 * C;
    const float* b_2 = vec + 4 * C;
    const float* g2 = vec + 5 * C;
    const float* be2 = vec + 6 * C;
    const float* b_qkv = vec + 7 * C;

    const int stride = gridDim.x * kP16Waves;
    const int first = blockIdx.x * kP16Waves;
    const int rounds = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    // (ahead of the LDS-DMA requests: what waits for these must not wait for those)
    float held[HELD];
#pragma unroll
    for (int i = 0; i < HELD; ++i) {
        const int index = min(static_cast<int>(threadIdx.x) + i * kP16Threads, FLOATS - 1);
        held[i] = index < 7 * C ? (BLOCK ? vectors[index] : 0.f)
                                : (QKV ? qkv_bias[index - 7 * C] : 0.f);
    }
    P16Packs<P, GEMMS> packs{p16_lds, BLOCK ? block_packs : qkv_packs, qkv_packs,
                             GEMMS * rounds, 0, lane, wave};
    packs.start();

    // a tile's inputs: the first GEMM's operand in operand order (the attention's output,
    // or x itself without the block), the residual stream in the accumulator layout
    // (columns beyond the segment read its last one)
    const uint32_t ld4 = 4u * static_cast<uint32_t>(ld);
    const p16_rows operand_rows = p16_buffer(BLOCK ? attended : x);
    const p16_rows x_rows = p16_buffer(x);
    const p16_rows qk_rows = p16_buffer(qk);
    const p16_rows v_rows = p16_buffer(v);
    auto request = [&](int tile, float (&operand)[kP16Steps][8], f32x4 (&residual)[kP16MTiles]) {
        const Tile span = load_tile(tiles, tile);
        const uint32_t at =
            p16_lane_offset(ld4, span.offset + min(span.first + p, span.count - 1), group);
#pragma unroll
        for (int j = 0; j < kP16Steps; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                operand[j][e] = 32 * j + 16 * (e >> 2) < C
                                    ? p16_load(operand_rows, ld4, 32 * j + 16 * (e >> 2) + (e & 3), at)
                                    : 0.f;
        if (BLOCK) {
#pragma unroll
            for (int m = 0; m < kP16MTiles; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) residual[m][i] = p16_load(x_rows, ld4, 16 * m + i, at);
        }
    };
    auto bias_of = [&](const float* bias, int m) {
        return *reinterpret_cast<const f32x4*>(bias + 16 * m + 4 * group);
    };

    int tile = first + wave;
    float operand[kP16Steps][8];
    f32x4 residual[kP16MTiles];
    if (tile < n_tiles) request(tile, operand, residual);
#pragma unroll
    for (int i = 0; i < HELD; ++i)
        if (static_cast<int>(threadIdx.x) + i * kP16Threads < FLOATS)
            vec[threadIdx.x + i * kP16Threads] = held[i];
    for (int round = 0; round < rounds; ++round, tile += stride) {
        if (tile >= n_tiles) {          // wave-uniform: the packs need every wave
#pragma unroll 1
            for (int g = 0; g < GEMMS; ++g) packs.acquire();
            continue;
        }
        const Tile span = load_tile(tiles, tile);
        const bool live = span.first + p < span.count;
        const int64_t column = span.offset + min(span.first + p, span.count - 1);
        const uint32_t at = p16_lane_offset(ld4, column, group);
        // this tile's inputs leave their registers (split) ...
        u32x4 b_frag[kP16Steps][P];
#pragma unroll
        for (int j = 0; j < kP16Steps; ++j) split_eight<P>(operand[j], b_frag[j]);
        f32x4 y[kP16MTiles];
        if (BLOCK) {
#pragma unroll
            for (int m = 0; m < kP16MTiles; ++m) y[m] = residual[m];
        }
        const unsigned char* pack = packs.acquire();
        // ... and the next tile's are requested into them (behind the barrier: they have
        // a whole GEMM to arrive in before the next barrier waits for them)
        if (tile + stride < n_tiles) request(tile + stride, operand, residual);
        if (BLOCK) {
            // y = LayerNorm1(x + b_o + W_o a)
#pragma unroll
            for (int m = 0; m < kP16MTiles; ++m) y[m] += bias_of(b_o, m);
            p16_gemm<P, false>(pack, lane, y, [&](int j, u32x4 (&b)[P]) {
#pragma unroll
                for (int piece = 0; piece < P; ++piece) b[piece] = b_frag[j][piece];
            });
            p16_layernorm(y, g1, be1, eps, group);
            // h = relu(b_1 + W_1 y)
            f32x4 h[kP16MTiles];
#pragma unroll
            for (int m = 0; m < kP16MTiles; ++m) h[m] = bias_of(b_1, m);
            pack = packs.acquire();
            p16_gemm<P, false>(pack, lane, h,
                               [&](int j, u32x4 (&b)[P]) { p16_fragment<P>(y, j, b); });
            if (act == EMPH_ACT_RELU) {
#pragma unroll
                for (int m = 0; m < kP16MTiles; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[m][i] = fmaxf(h[m][i], 0.f);
            }
            // z = LayerNorm2(y + b_2 + W_2 h), accumulated in y's registers
#pragma unroll
            for (int m = 0; m < kP16MTiles; ++m) y[m] += bias_of(b_2, m);
            pack = packs.acquire();
            p16_gemm<P, false>(pack, lane, y,
                               [&](int j, u32x4 (&b)[P]) { p16_fragment<P>(h, j, b); });
            p16_layernorm(y, g2, be2, eps, group);
            if (live) {
#pragma unroll
                for (int m = 0; m < kP16MTiles; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) p16_store(x_rows, ld4, 16 * m + i, at, y[m][i]);
            }
            if (!QKV) continue;
            // the next layer's projections of what was just stored: its registers ARE
            // the operand
#pragma unroll
            for (int j = 0; j < kP16Steps; ++j) p16_fragment<P>(y, j, b_frag[j]);
            pack = packs.acquire();
        }
        if (!QKV) continue;

        // ---- Q | K | V of the tile (`pack`: W_q's)
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
            if (part > 0) pack = packs.acquire();
            f32x4 acc[kP16MTiles];
#pragma unroll
            for (int m = 0; m < kP16MTiles; ++m) {
                if (swapped) {
                    const float bias = b_qkv[part * C + 16 * m + p];
                    acc[m] = f32x4{bias, bias, bias, bias};
                } else {
                    acc[m] = bias_of(b_qkv + part * C, m);
                }
            }
            auto fragment = [&](int j, u32x4 (&b)[P]) {
#pragma unroll
                for (int piece = 0; piece < P; ++piece) b[piece] = b_frag[j][piece];
            };
            if (swapped) p16_gemm<P, true>(pack, lane, acc, fragment);
            else p16_gemm<P, false>(pack, lane, acc, fragment);
            if (!IMAGES || part == 0) {
                if (live && part < 2) {
#pragma unroll
                    for (int m = 0; m < kP16MTiles; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            p16_store(qk_rows, ld4, part * C + 16 * m + i, at, acc[m][i]);
                } else if (live) {
#pragma unroll
                    for (int m = 0; m < kP16MTiles; ++m)
                        __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(u32x4, acc[m]), v_rows,
                            4u * (static_cast<uint32_t>(column) * C + 4 * group), 64 * m, 0);
                }
            } else if (part == 1) {
                // K image: [d / 8][key][8 d]; the lane's four consecutive d are half an
                // octet.  (Buffer stores again: the stage is the resource, a lane's key and
                // half one vector offset, the m-tile's (head, octet) - one of two constants
                // by the lane's group - a select, the piece a scalar offset.)
                const p16_rows stage_bytes = p16_buffer(stage);
                const uint32_t lane_key = 16u * static_cast<uint32_t>(key0 + p) + 8u * (group & 1);
#pragma unroll
                for (int m = 0; m < kP16MTiles; ++m) {
                    constexpr int OCTETS = D / 8;
                    // channel / 8 = 2 m + group / 2
                    const uint32_t even = (2 * m / OCTETS) * Images::kStageBytes +
                                          (2 * m % OCTETS) * kSplitStage * 16;
                    const uint32_t odd = ((2 * m + 1) / OCTETS) * Images::kStageBytes +
                                         ((2 * m + 1) % OCTETS) * kSplitStage * 16;
                    const uint32_t where = lane_key + ((group >> 1) ? odd : even);
                    uint32_t low[PK], high[PK];
                    split_pair<PK>(live ? acc[m][0] : 0.f, live ? acc[m][1] : 0.f, low);
                    split_pair<PK>(live ? acc[m][2] : 0.f, live ? acc[m][3] : 0.f, high);
#pragma unroll
                    for (int piece = 0; piece < PK; ++piece)
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{low[piece], high[piece]}, stage_bytes,
                                                              where, Images::key_piece(piece), 0);
                }
            } else {
                // V image: [key / 8][d][8 keys], the keys of each 16 permuted (position
                // 8 h + 4 a + i holds key 8 a + 4 h + i): the lane's keys 4 g + i are half
                // of chunk g % 2, at offset 4 (g / 2)
                const p16_rows stage_bytes = p16_buffer(stage);
                const uint32_t lane_chunk =
                    16u * static_cast<uint32_t>(((key0 >> 3) + (group & 1)) * Images::kRows) +
                    8u * (group >> 1);
#pragma unroll
                for (int m = 0; m < kP16MTiles; ++m) {
                    // channel 16 m + p = head x D + d (only the m-tile that holds channel D
                    // straddles the heads)
                    const int c = 16 * m + p;
                    const int head = c >= D ? 1 : 0;
                    const int d = c - D * head;
                    const uint32_t where =
                        lane_chunk + 16u * static_cast<uint32_t>(d) + head * Images::kStageBytes;
                    float keys[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        keys[i] = span.first + 4 * group + i < span.count ? acc[m][i] : 0.f;
                    uint32_t low[PV], high[PV];
                    split_pair<PV>(keys[0], keys[1], low);
                    split_pair<PV>(keys[2], keys[3], high);
#pragma unroll
                    for (int piece = 0; piece < PV; ++piece)
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{low[piece], high[piece]}, stage_bytes,
                                                              where, Images::value_piece(piece), 0);
                }
            }
        }
        if (IMAGES && key0 == 0) {
            // the tile that opens a stage writes what no projection produces: K's octets
            // from D / 8 on (ones at d = D in piece 0), V's row of ones and row of zeros -
            // and the zeros of the sixteens of keys the segment does not reach
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
            // (wave-uniform) first key no tile of the stage writes
            const int reached =
                min(kSplitStage, (min(span.count - span.first, kSplitStage) + 15) / 16 * 16);
            if (reached < kSplitStage) {
                const int keys = kSplitStage - reached, chunks = keys / 8;
                const int k_zero = HEADS * PK * (D / 8) * keys;
                const int v_zero = HEADS * PV * chunks * D;
                const u32x4 zero = {0u, 0u, 0u, 0u};
                for (int index = lane; index < k_zero + v_zero; index += 64) {
                    int head, byte;
                    if (index < k_zero) {
                        const int key = reached + index % keys, octet = index / keys % (D / 8);
                        const int piece = index / keys / (D / 8) % PK;
                        head = index / keys / (D / 8) / PK;
                        byte = Images::key_piece(piece) + (octet * kSplitStage + key) * 16;
                    } else {
                        const int rest = index - k_zero;
                        const int d = rest % D, chunk = reached / 8 + rest / D % chunks;
                        const int piece = rest / D / chunks % PV;
                        head = rest / D / chunks / PV;
                        byte = Images::value_piece(piece) + (chunk * Images::kRows + d) * 16;
                    }
                    *reinterpret_cast<u32x4*>(stage + head * Images::kStageBytes + byte) = zero;
                }
            }
        }
    }
}

}  // namespace emph

using namespace emph;

namespace {

// the pieces of split_pair (split.h) on the host: two, rounded to nearest; three, truncated
void p16_host_pieces(float value, int pieces, uint16_t (&out)[3]) {
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

int64_t emph_linear_split_pack16_size(int32_t pieces) {
    return pieces == 2 || pieces == 3 ? p16_pack_bytes(pieces) : 0;
}

// weight float32 [80][80] (HOST; out x in, as nn.Linear stores it) -> [k-step j][m-tile m]
// [piece][lane][8 bf16], lane = (output channel 16 m + lane % 16; input channels 32 j +
// 16 (e / 4) + 4 (lane / 16) + e % 4 for e = 0 .. 7: the order in which the result of
// one GEMM of the chain lies in the registers of the next); input channels 80 .. 95 are
// zeros.
int emph_linear_split_pack16(const float* host_weight, int32_t pieces, void* host_pack) {
    EMPH_REQUIRE(host_weight && host_pack, EMPH_EINVAL, "emph_linear_split_pack16: null pointer");
    EMPH_REQUIRE(pieces == 2 || pieces == 3, EMPH_ERANGE, "emph_linear_split_pack16: %d pieces",
                 pieces);
    uint16_t* out = static_cast<uint16_t*>(host_pack);
    for (int j = 0; j < kP16Steps; ++j)
        for (int m = 0; m < kP16MTiles; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int row = 16 * m + (lane & 15);
                    const int channel = 32 * j + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
                    const float weight =
                        channel < kP16Channels ? host_weight[row * kP16Channels + channel] : 0.f;
                    uint16_t parts[3];
                    p16_host_pieces(weight, pieces, parts);
                    const size_t base = (static_cast<size_t>(j) * kP16MTiles + m) * pieces * 512;
                    for (int piece = 0; piece < pieces; ++piece)
                        out[base + piece * 512 + lane * 8 + e] = parts[piece];
                }
    return EMPH_OK;
}

// The position-wise half of a layer and / or the next layer's projections, tiles of 16.
//   attended != NULL  the block: x <- LayerNorm2(...) as emph_transformer_block_split
//                     (block_packs = emph_linear_split_pack16 of out_proj | linear1 |
//                     linear2, vectors = b_o g1 be1 b_1 b_2 g2 be2)
//   qkv_packs != NULL the projections of (the new) x: emph_linear_split_pack16 of the q |
//                     k | v rows of in_proj_weight, qkv_bias [3][80]; images == NULL: qk
//                     float32 [160][ld] and v float32 [ld][80]; images != NULL: Q into
//                     qk's first 80 rows and the K / V images for `attention_pieces`
// Both: one launch, the six packs streamed through LDS; the results are bit for bit those
// of the two launches.
int emph_position_wise_split(const float* attended, float* x, int64_t ld, int32_t channels,
                             int32_t heads, const void* block_packs, const float* vectors,
                             const void* qkv_packs, const float* qkv_bias, int32_t pieces,
                             int32_t attention_pieces, float eps, int32_t activation,
                             const int32_t* tiles, int32_t n_tiles, int32_t tile_n, float* qk,
                             float* v, void* images, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    const char* what = "emph_position_wise_split";
    const bool block = attended != nullptr, qkv = qkv_packs != nullptr;
    EMPH_REQUIRE(block || qkv, EMPH_EINVAL, "%s: neither a block nor projections", what);
    EMPH_REQUIRE(x && tiles, EMPH_EINVAL, "%s: null pointer", what);
    EMPH_REQUIRE(!block || (block_packs && vectors), EMPH_EINVAL, "%s: the block's packs / vectors",
                 what);
    EMPH_REQUIRE(!qkv || (qkv_bias && qk && (images || v)), EMPH_EINVAL,
                 "%s: the projections' bias / outputs", what);
    EMPH_REQUIRE(channels == kP16Channels && tile_n == kP16Tile && heads == 2, EMPH_ERANGE,
                 "%s: %d channels, %d heads, tiles of %d (built for 80, 2 and 16)", what, channels,
                 heads, tile_n);
    EMPH_REQUIRE(pieces == 2 || pieces == 3, EMPH_ERANGE, "%s: %d pieces (2 or 3)", what, pieces);
    // (byte offsets of 32 bits: 160 rows of qk, v's ld x 80 floats)
    EMPH_REQUIRE(ld > 0 && ld < (int64_t{1} << 22), EMPH_ERANGE,
                 "%s: ld %lld outside the 32-bit byte offsets (2^22 columns: 11 hours of frames)",
                 what, static_cast<long long>(ld));
    EMPH_REQUIRE(activation == EMPH_ACT_RELU || activation == EMPH_ACT_NONE, EMPH_ERANGE,
                 "%s: activation %d", what, activation);
    EMPH_REQUIRE(!images || attention_pieces == 2 || attention_pieces == 3 || attention_pieces == 32,
                 EMPH_ERANGE, "%s: attention pieces %d (2, 3 or 32)", what, attention_pieces);
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(block_packs) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(qkv_packs) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(images) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(v) & 15) == 0,
                 EMPH_EINVAL, "%s: packs, images and v must be 16-byte aligned", what);
    const size_t lds = 3 * p16_pack_bytes(pieces) + 10 * kP16Channels * sizeof(float);
    const unsigned groups =
        static_cast<unsigned>(min((n_tiles + kP16Waves - 1) / kP16Waves, 256));
#define EMPH_P16(P, BLOCK, QKV, IMAGES, PK, PV)                                                \
    do {                                                                                       \
        auto kernel = position_wise16_kernel<P, BLOCK, QKV, IMAGES, PK, PV>;                   \
        static LdsReservation reserved;                                                        \
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,     \
                                     what))                                                    \
            return status;                                                                     \
        EMPH_LAUNCH(kernel, dim3(groups), dim3(kP16Threads), lds,                              \
                    static_cast<hipStream_t>(stream), attended, x, ld,                         \
                    static_cast<const unsigned char*>(block_packs),                            \
                    static_cast<const unsigned char*>(qkv_packs), vectors, qkv_bias, eps,      \
                    activation, qk, v, static_cast<unsigned char*>(images), tiles, n_tiles);   \
    } while (0)
#define EMPH_P16_MODES(P, BLOCK)                                                               \
    do {                                                                                       \
        if (!qkv) EMPH_P16(P, true, false, false, 2, 2);                                       \
        else if (images == nullptr) EMPH_P16(P, BLOCK, true, false, 2, 2);                     \
        else if (attention_pieces == 2) EMPH_P16(P, BLOCK, true, true, 2, 2);                  \
        else if (attention_pieces == 3) EMPH_P16(P, BLOCK, true, true, 3, 3);                  \
        else EMPH_P16(P, BLOCK, true, true, 3, 2);                                             \
    } while (0)
    if (pieces == 2) {
        if (block) EMPH_P16_MODES(2, true); else EMPH_P16_MODES(2, false);
    } else {
        if (block) EMPH_P16_MODES(3, true); else EMPH_P16_MODES(3, false);
    }
#undef EMPH_P16_MODES
#undef EMPH_P16
    return check_launch(what);
}

}  // extern "C"
