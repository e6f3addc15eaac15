// Up to THREE consecutive Conv1d(80, 80, 3, 'same') + activation layers of the frame
// encoder (emphases/model/core.py:24-31,96-100 over model/layers/convolution.py:25-37)
// in one launch on the bf16 matrix pipe, every fp32 operand split into TWO bf16 pieces
// ("bf16x3": hi.hi + hi.lo + lo.hi, fp32 accumulation; pieces rounded to nearest, each
// product within 2^-16) - the opt-in precision='bf16x3' of the engine; the default
// stays the fp32 kernel of conv_stack.hip.
//
// Why direct form and not Winograd here: v_mfma_f32_16x16x4_f32 shares the vector
// ALU's multipliers (profiles/r5_coexec.txt), which is what made F(4,3) pay in fp32 -
// half the matrix work.  On the bf16 pipe three products of the DIRECT form cost what
// six of the Winograd form would (the transform's extra bits want three pieces), need
// no input transform at all (nothing but LDS reads between the MFMAs), and the packed
// weights are a third of the bytes.  7 x 3 x 80 x 240 x 64 000 x 2 = 51.6 GFLOP of
// v_mfma_f32_32x32x16_bf16 against 8.8 GFLOP of fp32 MFMA at 1/16 of the rate.
//
// A workgroup owns a SPAN of one segment for all the layers of the launch - the span
// table of emph_conv_stack_spans: 256 computed positions, of which a quad at an inner
// end is halo.  In the direct form layer l of a launch spoils l - 1 positions per inner
// end (the column beside the computed region is exact for the first layer only), so
// that quad covers FIVE layers: the seven frame-rate layers are two launches (4 + 3),
// where the F(4,3) kernel needs three.  As in conv_stack.hip:
//   * activations live in LDS as bf16 pieces, POSITION-major: X[piece][row][88]
//     (80 channels + 8 of padding; row i is position c0 - 1 + i, 258 rows used), so
//     that the B fragment of a k-step - eight consecutive input channels of one
//     position and tap - is one 16-byte read and rows 176 bytes apart are conflict-free;
//     positions outside the segment hold zeros ('same' padding by construction);
//   * the contraction is (tap, channel): k-step s = 5 tap + block multiplies channels
//     16 block .. + 15 of position p + tap - 1; a tap's five k-steps of packed weights
//     (3 m-tiles of 32 output channels x 2 pieces x 1 KB = 30 KB) are one CHUNK of the
//     two-slot weight ring, streamed by four loader waves with LDS-DMA while the eight
//     MFMA waves (one 32-position column tile each, all three m-tiles: 9 MFMAs per
//     k-step) consume the previous one.  One barrier per chunk, one more per layer;
//   * a layer's output is split in registers and written over its input behind that
//     barrier; the launch's last layer stores the span's own positions as fp32.
#include <string.h>

#include "common.h"

// (tools/micro/conv_split_bench.hip defines CONV_STAMP for an in-kernel timeline)
#ifndef CONV_STAMP
#define CONV_STAMP(slot)
#define CONV_STAMP_ARGUMENT
#define CONV_STAMP_DECLARE
#define CONV_STAMP_FINISH
#define CONV_STAMP_PASS
#endif

namespace emph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// (tools/micro/corun_split.hip builds a 128-position variant - four MFMA waves, 111 KB of
// LDS - to put a front-end workgroup beside it on a CU; the product is 256)
#ifndef CONV_SPLIT_WIDTH
#define CONV_SPLIT_WIDTH 256
#endif

constexpr int kSplitChannels = 80;
constexpr int kSplitWidth = CONV_SPLIT_WIDTH;       // computed positions
constexpr int kSplitMfmaWaves = kSplitWidth / 32;   // one 32-position column tile each
constexpr int kSplitMfmaThreads = 64 * kSplitMfmaWaves;
constexpr int kSplitRows = kSplitWidth + 2;         // + the columns beside them
constexpr int kSplitRowBytes = 176;                 // 88 bf16
constexpr int kSplitImageBytes = (kSplitWidth + 8) * kSplitRowBytes;         // one piece
constexpr int kSplitMTiles = 3;                     // 96 rows, 80 used
constexpr int kSplitBlocks = kSplitChannels / 16;   // k-steps per tap
constexpr int kSplitChunkBytes = kSplitBlocks * kSplitMTiles * 2 * 1024;     // one tap
constexpr int kSplitLayerBytes = 3 * kSplitChunkBytes;
constexpr int kSplitThreads = kSplitMfmaThreads + 256;      // 8 MFMA waves + 4 loader waves
constexpr int kSplitMaxLayers = 5;
constexpr int kSplitLdsBytes = 2 * kSplitImageBytes + 2 * kSplitChunkBytes +
                               kSplitMaxLayers * kSplitChannels * 4;

// two floats -> two dwords of two bf16 (element 0 in the low half), rounded to nearest
__device__ __forceinline__ void split_two(float a, float b, uint32_t& high, uint32_t& low) {
    const bf16x2 first = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
    uint32_t bits = __builtin_bit_cast(uint32_t, first);
    asm("" : "+v"(bits));       // (attention_split.hip: keep hipcc from converting twice)
    const float ra = a - __uint_as_float(bits << 16);
    const float rb = b - __uint_as_float(bits & 0xffff0000u);
    const bf16x2 second = {static_cast<__bf16>(ra), static_cast<__bf16>(rb)};
    high = bits;
    low = __builtin_bit_cast(uint32_t, second);
}

__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// spans: int32 [n][8] of emph_conv_stack_spans
// WORD_SUMS: the launch's last layer is the one in front of the per-word sum
// (emphases/core.py:438-454): instead of its output it leaves, in y = sums[slot][ldy],
// the running sum of the span's own positions - restarting at the span's first own
// position and at every wave's column tile of 32 - at the frames `slot_map` marks
// (`Plan.word_sum_tables` with restarts every 32 computed positions); emph_word_sums
// adds a word's handful of signed terms (as conv_stack.hip does for the fp32 kernel).
template <bool WORD_SUMS>
__global__ __launch_bounds__(kSplitThreads) void conv1d_split_kernel(
    const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
    const unsigned char* __restrict__ packs, const float* __restrict__ biases, int layers,
    int relu_mask, const int32_t* __restrict__ spans,
    const int32_t* __restrict__ slot_map CONV_STAMP_ARGUMENT) {
    CONV_STAMP_DECLARE
    extern __shared__ __align__(16) unsigned char conv_split_lds[];
    unsigned char* image = conv_split_lds;                         // [2 pieces][264 rows][176 B]
    unsigned char* ring = image + 2 * kSplitImageBytes;            // [2 slots][30 KB]
    float* bias_lds = reinterpret_cast<float*>(ring + 2 * kSplitChunkBytes);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= kSplitMfmaWaves;

    const int4 span_a = reinterpret_cast<const int4*>(spans)[2 * blockIdx.x];
    const int4 span_b = reinterpret_cast<const int4*>(spans)[2 * blockIdx.x + 1];
    const int owned_first = span_a.y;
    const int64_t column = span_a.z;          // frame column of the segment's position 0
    const int count = span_a.w;               // positions of the segment
    const int owned = span_b.x;
    const int c0 = span_b.y;                  // first computed position

    const int total_chunks = 3 * layers;
    // chunk g (layer g / 3, tap g % 3) -> ring[g & 1]: 30 requests of 1 KB; every loader
    // wave issues eight (the last two of the fourth wave repeat the chunk's last KB)
    auto request = [&](int g) {
        const unsigned char* source = packs + static_cast<int64_t>(g) * kSplitChunkBytes;
        unsigned char* target = ring + (g & 1) * kSplitChunkBytes;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int from = min((wave - kSplitMfmaWaves) + 4 * k, kSplitChunkBytes / 1024 - 1) * 1024;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(source + from + 16 * lane),
                (__attribute__((address_space(3))) void*)(target + from), 16, 0, 0);
        }
    };

    if (loader) {
        request(0);
        __syncthreads();                               // (the MFMA waves' "biases staged")
        for (int g = 0; g < total_chunks; ++g) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // chunk g starts
            if (g + 1 < total_chunks) request(g + 1);  // (its slot held chunk g - 1: done)
            // (the MFMA waves' barrier in front of a layer's in-place update)
            if ((g + 1) % 3 == 0 && g + 1 < total_chunks) __syncthreads();
        }
        return;
    }

    // ---- the launch's input: fp32 [80][ld] -> bf16 pieces, position-major.  A thread
    // takes four channels of one position (loads coalesced over positions), rows of
    // positions outside the segment become zeros.
    {
        const int tid = threadIdx.x;                   // 0 .. 511
        for (int i = tid; i < layers * kSplitChannels; i += kSplitMfmaThreads)
            bias_lds[i] = biases[i];
        // (every load of the thread is in flight before the first is used: a load is
        // 1-2 us away when the whole chip starts a launch)
        constexpr int kTasks = (kSplitChannels / 4) * kSplitRows;
        constexpr int kRounds = (kTasks + kSplitMfmaThreads - 1) / kSplitMfmaThreads;
        float raw[kRounds][4];
#pragma unroll
        for (int round = 0; round < kRounds; ++round) {
            const int task = min(tid + kSplitMfmaThreads * round, kTasks - 1);
            const int quad = task / kSplitRows, row = task - quad * kSplitRows;
            const float* source = x + static_cast<int64_t>(4 * quad) * ldx + column +
                                  min(max(c0 - 1 + row, 0), count - 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[round][e] = source[e * ldx];
        }
#pragma unroll
        for (int round = 0; round < kRounds; ++round) {
            const int task = tid + kSplitMfmaThreads * round;
            if (task >= kTasks) break;
            const int quad = task / kSplitRows, row = task - quad * kSplitRows;
            const int p = c0 - 1 + row;
            const float gate = (p >= 0 && p < count) ? 1.f : 0.f;
            uint32_t h0, h1, l0, l1;
            split_two(raw[round][0] * gate, raw[round][1] * gate, h0, l0);
            split_two(raw[round][2] * gate, raw[round][3] * gate, h1, l1);
            unsigned char* target = image + row * kSplitRowBytes + quad * 8;
            *reinterpret_cast<u32x2*>(target) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(target + kSplitImageBytes) = u32x2{l0, l1};
        }
    }

    CONV_STAMP(0);                                      // input staged
    // ---- MFMA waves: column tile `wave` (32 positions), all three m-tiles
    const int col = lane & 31;
    const int half = lane >> 5;
    const int p_mine = c0 + 32 * wave + col;            // the lane's output position
    // B fragment of (tap, block): row 32 wave + col + tap, channels 16 block + 8 half ..
    const unsigned char* lane_rows = image + (32 * wave + col) * kSplitRowBytes + 16 * half;
    // (wave-uniform: the whole column tile lies inside the segment - no masks needed)
    const bool all_inside = __builtin_amdgcn_readfirstlane(
        c0 + 32 * wave >= 0 && c0 + 32 * wave + 31 < count);
    for (int layer = 0; layer < layers; ++layer) {
        // the accumulators start at the bias (rows 80 .. 95 of the last tile: zero
        // weights, never stored)
        f32x16 acc[kSplitMTiles];
        const float* bias_row = bias_lds + layer * kSplitChannels;
        if (layer == 0) __syncthreads();               // (the biases are in LDS)
#pragma unroll
        for (int m = 0; m < kSplitMTiles; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[m][r] = 32 * m + 8 * (r >> 2) < kSplitChannels
                                ? bias_row[32 * m + 8 * (r >> 2) + 4 * half + (r & 3)]
                                : 0.f;
        for (int tap = 0; tap < 3; ++tap) {
            // the chunk has landed (tap 0: and every wave has written its part of this
            // layer's input)
            CONV_STAMP(1);                              // (a tap's MFMAs issued)
            __syncthreads();
            CONV_STAMP(2);                              // waited for the chunk
            const unsigned char* weights = ring + ((3 * layer + tap) & 1) * kSplitChunkBytes +
                                           16 * lane;
            const unsigned char* rows = lane_rows + tap * kSplitRowBytes;
#pragma unroll
            for (int block = 0; block < kSplitBlocks; ++block) {
                const u32x4 b_high = *reinterpret_cast<const u32x4*>(rows + 32 * block);
                const u32x4 b_low =
                    *reinterpret_cast<const u32x4*>(rows + 32 * block + kSplitImageBytes);
#pragma unroll
                for (int m = 0; m < kSplitMTiles; ++m) {
                    const u32x4 a_high = *reinterpret_cast<const u32x4*>(
                        weights + ((block * kSplitMTiles + m) * 2) * 1024);
                    const u32x4 a_low = *reinterpret_cast<const u32x4*>(
                        weights + ((block * kSplitMTiles + m) * 2 + 1) * 1024);
                    // the small products first
                    acc[m] = mfma32(a_low, b_high, acc[m]);
                    acc[m] = mfma32(a_high, b_low, acc[m]);
                    acc[m] = mfma32(a_high, b_high, acc[m]);
                }
            }
        }
        // ---- bias, activation; rows of the 32 x 32 tile: channel 32 m + 8 (r / 4) +
        // 4 half + r % 4
        const bool relu = (relu_mask >> layer) & 1;
        const bool last = layer == layers - 1;
        const bool inside = p_mine >= 0 && p_mine < count;
        if (!last) {
            // every wave is done reading this layer's input: its output takes the rows'
            // place (zeros outside the segment)
            CONV_STAMP(1);
            __syncthreads();
            CONV_STAMP(3);                              // waited for the layer's last reader
            unsigned char* target = image + (32 * wave + col + 1) * kSplitRowBytes;
#pragma unroll
            for (int m = 0; m < kSplitMTiles; ++m)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (32 * m + 8 * b >= kSplitChannels) continue;  // (m = 2: rows 80 ..)
                    const int channel = 32 * m + 8 * b + 4 * half;
                    float value[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float out = acc[m][4 * b + i];
                        if (relu) out = fmaxf(out, 0.f);
                        value[i] = (all_inside || inside) ? out : 0.f;
                    }
                    uint32_t h0, h1, l0, l1;
                    split_two(value[0], value[1], h0, l0);
                    split_two(value[2], value[3], h1, l1);
                    *reinterpret_cast<u32x2*>(target + 2 * channel) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(target + 2 * channel + kSplitImageBytes) =
                        u32x2{l0, l1};
                }
            CONV_STAMP(4);                              // output split and written
            continue;
        }
        // ---- the launch's last layer: the span's own positions leave the chip as fp32
        const bool mine = inside && p_mine >= owned_first && p_mine < owned_first + owned;
        if (WORD_SUMS) {
            // running sums over the wave's 32 positions, channel by channel: an inclusive
            // scan over the 32 lanes of a half (two DPP rows of 16: row_shr 1, 2, 4, 8
            // inside a row, then lane 15 of the lower row into the upper one), a fixed
            // order; positions that are not the span's own count as zero
            const int slot = mine ? slot_map[column + p_mine] : -1;
#pragma unroll
            for (int m = 0; m < kSplitMTiles; ++m)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (32 * m + 8 * b >= kSplitChannels) continue;
                    const int channel = 32 * m + 8 * b + 4 * half;
                    float sum[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float out = acc[m][4 * b + i];
                        if (relu) out = fmaxf(out, 0.f);
                        float scan = mine ? out : 0.f;
                        scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, scan), 0x111, 0xf, 0xf, true));
                        scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, scan), 0x112, 0xf, 0xf, true));
                        scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, scan), 0x114, 0xf, 0xf, true));
                        scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, scan), 0x118, 0xf, 0xf, true));
                        // row_bcast15 into rows 1 and 3: the lower row's total
                        scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, scan), 0x142, 0xa, 0xf, true));
                        sum[i] = scan;
                    }
                    if (slot >= 0)
                        *reinterpret_cast<float4*>(y + static_cast<int64_t>(slot) * ldy + channel) =
                            make_float4(sum[0], sum[1], sum[2], sum[3]);
                }
            CONV_STAMP(5);
            continue;
        }
        if (mine) {
#pragma unroll
            for (int m = 0; m < kSplitMTiles; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (32 * m + 8 * (r >> 2) >= kSplitChannels) continue;
                    const int channel = 32 * m + 8 * (r >> 2) + 4 * half + (r & 3);
                    float out = acc[m][r];
                    if (relu) out = fmaxf(out, 0.f);
                    y[static_cast<int64_t>(channel) * ldy + column + p_mine] = out;
                }
        }
        CONV_STAMP(5);                                  // output stored
    }
    CONV_STAMP_FINISH
}

}  // namespace emph

using namespace emph;

namespace {

// fp32 -> bf16 bits, round to nearest even (finite inputs)
uint16_t bf16_bits(float value) {
    uint32_t bits;
    memcpy(&bits, &value, 4);
    bits += 0x7fffu + ((bits >> 16) & 1u);
    return static_cast<uint16_t>(bits >> 16);
}

float bf16_value(uint16_t bits) {
    const uint32_t wide = static_cast<uint32_t>(bits) << 16;
    float value;
    memcpy(&value, &wide, 4);
    return value;
}

}  // namespace

extern "C" {

int64_t emph_conv_split_pack_size(void) { return kSplitLayerBytes; }

// weight float32 [80][80][3] (HOST) -> the layer's pack: [tap][block of 16 input
// channels][m-tile][piece][lane][8 bf16], lane = (output channel 32 m + lane % 32, input
// channels 16 block + 8 (lane / 32) ..); two pieces per weight, rounded to nearest;
// rows 80 .. 95 of the third m-tile are zeros.
int emph_conv_split_pack(const float* host_weight, void* host_pack) {
    EMPH_REQUIRE(host_weight && host_pack, EMPH_EINVAL, "emph_conv_split_pack: null pointer");
    uint16_t* out = static_cast<uint16_t*>(host_pack);
    for (int tap = 0; tap < 3; ++tap)
        for (int block = 0; block < kSplitBlocks; ++block)
            for (int m = 0; m < kSplitMTiles; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int row = 32 * m + (lane & 31);
                        const int channel = 16 * block + 8 * (lane >> 5) + e;
                        const float weight =
                            row < kSplitChannels
                                ? host_weight[(row * kSplitChannels + channel) * 3 + tap]
                                : 0.f;
                        const uint16_t high = bf16_bits(weight);
                        const uint16_t low = bf16_bits(weight - bf16_value(high));
                        const size_t base =
                            (((static_cast<size_t>(tap) * kSplitBlocks + block) * kSplitMTiles + m) * 2) *
                            512;
                        out[base + lane * 8 + e] = high;
                        out[base + 512 + lane * 8 + e] = low;
                    }
    return EMPH_OK;
}

// `layers` (1 .. 5) consecutive Conv1d(80, 80, 3, 'same') layers in one launch, products
// as bf16x3.
//   packs   emph_conv_split_pack of every layer, back to back (device, 16-byte aligned)
//   biases  float32 [layers][80]
//   relu_mask  bit l: layer l is followed by ReLU (else identity)
//   spans   int32 [n_spans][8] from emph_conv_stack_spans (device copy)
//   slot_map != NULL: the last layer leaves running sums in y = sums[slot][ldy]
//   (restarts at a span's first own position and every 32 computed positions:
//   `Plan.word_sum_tables(Plan.sum_restarts(spans, step=32))`), for emph_word_sums
int emph_conv1d_split(const float* x, int64_t ldx, float* y, int64_t ldy, const void* packs,
                      const float* biases, int32_t layers, int32_t relu_mask,
                      const int32_t* spans, int32_t n_spans, const int32_t* slot_map,
                      void* stream) {
    if (n_spans == 0) return EMPH_OK;
    EMPH_REQUIRE(x && y && packs && biases && spans, EMPH_EINVAL, "emph_conv1d_split: null pointer");
    EMPH_REQUIRE(layers >= 1 && layers <= kSplitMaxLayers, EMPH_ERANGE,
                 "emph_conv1d_split: %d layers (1 .. %d)", layers, kSplitMaxLayers);
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(packs) & 15) == 0, EMPH_EINVAL,
                 "emph_conv1d_split: the packs must be 16-byte aligned");
    EMPH_REQUIRE(slot_map == nullptr ||
                     ((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (ldy & 3) == 0 &&
                      ldy >= kSplitChannels),
                 EMPH_EINVAL, "emph_conv1d_split: bad sums buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned char* bytes = static_cast<const unsigned char*>(packs);
    if (slot_map != nullptr) {
        auto kernel = conv1d_split_kernel<true>;
        static LdsReservation reserved;
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel),
                                     kSplitLdsBytes, "emph_conv1d_split"))
            return status;
        EMPH_LAUNCH(kernel, dim3(n_spans), dim3(kSplitThreads), kSplitLdsBytes, s, x, ldx, y, ldy,
                    bytes, biases, layers, relu_mask, spans, slot_map CONV_STAMP_PASS);
    } else {
        auto kernel = conv1d_split_kernel<false>;
        static LdsReservation reserved;
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel),
                                     kSplitLdsBytes, "emph_conv1d_split"))
            return status;
        EMPH_LAUNCH(kernel, dim3(n_spans), dim3(kSplitThreads), kSplitLdsBytes, s, x, ldx, y, ldy,
                    bytes, biases, layers, relu_mask, spans, slot_map CONV_STAMP_PASS);
    }
    return check_launch("emph_conv1d_split");
}

}  // extern "C"
