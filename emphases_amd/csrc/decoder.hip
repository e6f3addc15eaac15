// Word-rate stage of the convolutional model in ONE launch:
//   word embeddings -> N x [Conv1d 'same' + act] -> Conv1d(channels, 1) ->
//   sigmoid / clamp.
//
// Replaces the word_decoder stack (emphases/model/core.py:105-107 over
// model/layers/convolution.py:25-30), output_layer (model/core.py:33-37,138)
// and emphases.postprocess (core.py:335-342).  As separate kernels these are
// eight launches of a few microseconds of work each (a 10 s utterance has ~30
// words), so launch floors and cold L2 round trips (1-2 us per dependent
// access) dominate; fused, the word activations never leave LDS and the only
// global traffic inside the layer loop is the LDS-DMA weight stream.
//
// One workgroup owns a window of kWindow = 64 consecutive word positions of
// one segment: `block` output words plus a halo of `halo` words on either side
// (the receptive field of the remaining layers), recomputed per tile; a segment
// of at most 64 words is one tile with no halo.  Per layer the [C x 64] output
// is C/16 x 4 MFMA tiles (fp32 v_mfma_f32_16x16x4_f32): wave (m, h) owns m-tile
// m and n-tiles {2h, 2h+1}.  The packed weights stream through a two-slot LDS
// ring by LDS-DMA, one chunk (about half a layer) ahead of the MFMAs, so each
// weight is read from L2 once per workgroup and never waited for; B fragments
// come from the LDS-resident activations, whose row stride of 80 floats keeps
// the four k-rows of a fragment on disjoint banks.
// Every layer re-applies the segment's own zero halo, so ragged batches keep
// the reference's B=1 edge semantics.
#include <math.h>
#include <stdint.h>

#include "common.h"

#ifndef EMPH_STAMP
#define EMPH_STAMP(slot)   // in-kernel timeline stamps: tools/micro only
#endif

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWindow = 64;        // word positions per workgroup
constexpr int kLeadCols = 4;       // zero columns left of the window in LDS
constexpr int kActStride = 80;     // floats per LDS activation row (16 mod 32)

__device__ __forceinline__ float activate_word(float x, int act) {
    switch (act) {
        case EMPH_ACT_RELU: return fmaxf(x, 0.f);
        case EMPH_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
        case EMPH_ACT_SILU: return x / (1.f + expf(-x));
        case EMPH_ACT_LEAKY_RELU: return x > 0.f ? x : 0.01f * x;
        default: return x;
    }
}

// block = 128 * m_tiles threads (wave = (m-tile, n-half)); grid = n_tiles.
// MAX_M (5 or 8 m-tiles) bounds the block size and with it the register budget.
template <int KS, int MAX_M>
__global__ __launch_bounds__(128 * MAX_M) void word_decoder_kernel(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ tiles,
    int block, int halo, int channels, const float* __restrict__ packs,
    const float* __restrict__ biases, int layers, int act, int chunk_steps,
    const float* __restrict__ out_weight, const float* __restrict__ out_bias,
    int out_kernel, int post, float* __restrict__ logits, float* __restrict__ scores) {
    extern __shared__ __align__(16) float lds[];
    EMPH_STAMP(0);
    const int m_tiles = channels >> 4;
    const int threads = blockDim.x;
    const int waves = threads >> 6;
    // (the two activation buffers are addressed as lds + index * size: a runtime-
    // indexed array of pointers would decay to flat addressing)
    const int buffer_floats = channels * kActStride;
    float* partial = lds + 2 * channels * kActStride;   // [waves][kWindow]
    float* bias_lds = partial + waves * kWindow;        // [layers][channels]
    float* out_lds = bias_lds + layers * channels;      // [channels][out_kernel] + 1
    // two weight chunks of chunk_steps k-steps each, 16-byte aligned
    float* ring = lds + ((2 * channels * kActStride + waves * kWindow + layers * channels +
                          channels * out_kernel + 1 + 3) & ~3);
    const int chunk_floats = chunk_steps * m_tiles * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4;
    const int col = lane & 15;

    const Tile span = load_tile(tiles, blockIdx.x);
    const bool single = span.count <= kWindow;        // whole segment, no halo
    if (single && span.first != 0) return;            // covered by the first tile
    const int first_out = span.first;
    const int last_out = single ? span.count : min(span.first + block, span.count);
    const int start = single ? 0 : span.first - halo; // word index of window col 0
    // a segment of at most 32 words needs only two of the four 16-column tiles:
    // every wave then owns ONE of them and the per-layer MFMA work halves
    const bool narrow = single && span.count <= 32;

    // Weight stream: the decoder's packs, layer after layer, cut into chunks of
    // chunk_steps k-steps.  Chunk q is copied by LDS-DMA into ring[q & 1] while
    // chunk q-1 is being consumed.
    constexpr int HALO = (KS - 1) / 2;
    const int groups_k = KS == 1 ? (((channels + 15) & ~15) >> 2) : (((channels + 7) & ~7) >> 2);
    const int steps = groups_k * KS;                  // k-steps per layer
    const int chunks_per_layer = (steps + chunk_steps - 1) / chunk_steps;
    const int total_chunks = layers * chunks_per_layer;
    auto request = [&](int q) {
        const int layer = q / chunks_per_layer;
        const int first = (q - layer * chunks_per_layer) * chunk_steps;
        const int quads = min(chunk_steps, steps - first) * m_tiles * 16;
        const float* source =
            packs + (static_cast<int64_t>(layer) * steps + first) * m_tiles * 64;
        float* target = ring + (q & 1) * chunk_floats;
        for (int base = wave * 64; base < quads; base += threads) {
            if (base + 64 <= quads) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(source + 4 * (base + lane)),
                    (__attribute__((address_space(3))) void*)(target + 4 * base), 16, 0, 0);
            } else if (base + lane < quads) {
                reinterpret_cast<float4*>(target)[base + lane] =
                    reinterpret_cast<const float4*>(source)[base + lane];
            }
        }
    };
    if (total_chunks > 0) request(0);

    // Everything else the stages need comes into LDS in the same round trip
    // (a dependent global load costs 1-2 us on this chip): the window of word
    // embeddings with the segment's zero halo, biases, the output projection.
    for (int index = threadIdx.x; index < channels * (kActStride / 4); index += threads) {
        const int c = index / (kActStride / 4);
        const int quad = index - c * (kActStride / 4);
        float4 value = make_float4(0.f, 0.f, 0.f, 0.f);
        const int p = 4 * quad - kLeadCols;           // window position of .x
        if (p >= 0 && p < kWindow) {
            const int word = start + p;
            const float* source = x + static_cast<int64_t>(c) * ldx + span.offset + word;
            if (word >= 0 && word + 3 < span.count &&
                (reinterpret_cast<uintptr_t>(source) & 15) == 0) {
                value = *reinterpret_cast<const float4*>(source);
            } else {
                if (word >= 0 && word < span.count) value.x = source[0];
                if (word + 1 >= 0 && word + 1 < span.count) value.y = source[1];
                if (word + 2 >= 0 && word + 2 < span.count) value.z = source[2];
                if (word + 3 >= 0 && word + 3 < span.count) value.w = source[3];
            }
        }
        reinterpret_cast<float4*>(lds)[index] = value;
        reinterpret_cast<float4*>(lds + buffer_floats)[index] =
            make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int index = threadIdx.x; index < layers * channels; index += threads)
        bias_lds[index] = biases[index];
    for (int index = threadIdx.x; index <= channels * out_kernel; index += threads)
        out_lds[index] = index < channels * out_kernel ? out_weight[index] : out_bias[0];
    EMPH_STAMP(1);

    // ---- decoder layers: wave (m, half) owns m-tile m, n-tiles 2*half, 2*half+1
    const int m = wave % m_tiles;
    const int half = wave / m_tiles;
    int current = 0;
    for (int layer = 0; layer < layers; ++layer) {
        const float* source = lds + current * buffer_floats;
        float* target = lds + (current ^ 1) * buffer_floats;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (int piece = 0; piece < chunks_per_layer; ++piece) {
            const int q = layer * chunks_per_layer + piece;
            __builtin_amdgcn_s_waitcnt(0x0F70);       // my DMA pieces of chunk q landed
            __syncthreads();                          // everyone's did; ring[(q+1)&1] is free
#ifdef EMPH_DECODER_SKIP
            if (!(EMPH_DECODER_SKIP & 1))
#endif
            if (q + 1 < total_chunks) request(q + 1);
            const float* fragment = ring + (q & 1) * chunk_floats + (m << 6) + lane;
            const int first = piece * chunk_steps;            // multiple of KS
            const int groups_here = min(chunk_steps, steps - first) / KS;
            const float* b_base = source + (4 * (first / KS) + kk) * kActStride +
                                  kLeadCols + (narrow ? 16 : 32) * half + col - HALO;
            // kTrip 4-row groups per trip: their 3*KS*kTrip LDS reads are issued
            // together (one exposed LDS latency per trip instead of one per
            // group), then the MFMAs run back to back
            constexpr int kTrip = 4;
            for (int g = 0; g < groups_here; g += kTrip) {
                float a[kTrip][KS], b[kTrip][KS][2];
#pragma unroll
                for (int i = 0; i < kTrip; ++i) {
                    const int gi = min(g + i, groups_here - 1);
#pragma unroll
                    for (int tap = 0; tap < KS; ++tap) {
                        a[i][tap] = fragment[(gi * KS + tap) * m_tiles * 64];
                        b[i][tap][0] = b_base[4 * gi * kActStride + tap];
                        b[i][tap][1] = b_base[4 * gi * kActStride + tap + 16];
                    }
                }
#pragma unroll
                for (int i = 0; i < kTrip; ++i) {
                    if (g + i >= groups_here) break;          // wave-uniform
#pragma unroll
                    for (int tap = 0; tap < KS; ++tap) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            a[i][tap], b[i][tap][0], acc[0], 0, 0, 0);
                        if (!narrow)
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                a[i][tap], b[i][tap][1], acc[1], 0, 0, 0);
                    }
                }
            }
        }
        // bias + activation; re-apply the segment's zero halo
        const float* bias = bias_lds + layer * channels;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (narrow && n == 1) break;
            const int p = (narrow ? 16 : 32) * half + 16 * n + col;
            const int word = start + p;
            const bool inside = word >= 0 && word < span.count;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * m + 4 * kk + r;
                target[c * kActStride + kLeadCols + p] =
                    inside ? activate_word(acc[n][r] + bias[c], act) : 0.f;
            }
        }
        EMPH_STAMP(3 + layer);
        current ^= 1;
    }
    __syncthreads();

    // ---- output projection: channels are split over the waves, then summed
    {
        const float* source = lds + current * buffer_floats;
        const int halo_out = (out_kernel - 1) / 2;
        float sum = 0.f;
        for (int c = wave; c < channels; c += waves)
            for (int tap = 0; tap < out_kernel; ++tap)
                sum = fmaf(out_lds[c * out_kernel + tap],
                           source[c * kActStride + kLeadCols + lane + tap - halo_out], sum);
        partial[wave * kWindow + lane] = sum;
        __syncthreads();
        if (wave == 0) {
            float total = out_lds[channels * out_kernel];
            for (int w = 0; w < waves; ++w) total += partial[w * kWindow + lane];
            const int word = start + lane;
            if (word >= first_out && word < last_out) {
                const int64_t column = span.offset + word;
                if (logits != nullptr) logits[column] = total;
                if (scores != nullptr) {
                    float value = total;
                    if (post == EMPH_POST_SIGMOID) value = 1.f / (1.f + expf(-total));
                    if (post == EMPH_POST_CLAMP01) value = fminf(fmaxf(total, 0.f), 1.f);
                    scores[column] = value;
                }
            }
        }
        EMPH_STAMP(9);
    }
}

}  // namespace emph

using namespace emph;

extern "C" {

int32_t emph_word_decoder_block(int32_t layers, int32_t kernel_size,
                                int32_t out_kernel_size) {
    const int halo = layers * ((kernel_size - 1) / 2) + (out_kernel_size - 1) / 2;
    return kWindow - 2 * halo;
}

int emph_word_decoder(const float* x, int64_t ldx, const int32_t* tiles,
                      int32_t n_tiles, int32_t channels, const float* packs,
                      const float* biases, int32_t layers, int32_t kernel_size,
                      int32_t activation, const float* out_weight,
                      const float* out_bias, int32_t out_kernel_size, int32_t post,
                      float* logits, float* scores, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(x && tiles && out_weight && out_bias, EMPH_EINVAL,
                 "emph_word_decoder: null pointer");
    EMPH_REQUIRE(layers == 0 || (packs && biases), EMPH_EINVAL,
                 "emph_word_decoder: decoder weights are null");
    EMPH_REQUIRE(layers >= 0 && layers <= 16, EMPH_ERANGE,
                 "emph_word_decoder: %d layers", layers);
    EMPH_REQUIRE(channels >= 16 && channels <= 128 && channels % 16 == 0, EMPH_ERANGE,
                 "emph_word_decoder: channels %d not a multiple of 16 in 16..128",
                 channels);
    EMPH_REQUIRE(kernel_size == 1 || kernel_size == 3 || kernel_size == 5, EMPH_ERANGE,
                 "emph_word_decoder: kernel_size %d not in {1,3,5}", kernel_size);
    EMPH_REQUIRE(out_kernel_size >= 1 && out_kernel_size <= 7 && (out_kernel_size & 1),
                 EMPH_ERANGE, "emph_word_decoder: out_kernel_size %d", out_kernel_size);
    const int block = emph_word_decoder_block(layers, kernel_size, out_kernel_size);
    EMPH_REQUIRE(block >= 16, EMPH_ERANGE,
                 "emph_word_decoder: receptive field too wide for a 64-word window");
    const int halo = (kWindow - block) / 2;
    const int m_tiles = channels / 16;
    const int threads = 128 * m_tiles;
    // two weight chunks of about 36 KB each, less when the activations of a
    // wide model leave less LDS
    const size_t fixed_bytes = ((2 * channels * kActStride + (threads / 64) * kWindow +
                                 layers * channels + channels * out_kernel_size + 1 + 3) & ~3) *
                               sizeof(float);
    size_t ring_budget = 156 * 1024 - fixed_bytes;
    if (ring_budget > 72 * 1024) ring_budget = 72 * 1024;
    const int groups_k = kernel_size == 1 ? (((channels + 15) & ~15) >> 2)
                                          : (((channels + 7) & ~7) >> 2);
    // whole 4-row groups per chunk, split evenly over the layer
    int chunk_groups = static_cast<int>(ring_budget / 2) / (m_tiles * 256) / kernel_size;
    if (chunk_groups < 1) chunk_groups = 1;
    if (chunk_groups > groups_k) chunk_groups = groups_k;
    const int pieces = (groups_k + chunk_groups - 1) / chunk_groups;
    chunk_groups = (groups_k + pieces - 1) / pieces;
    const int chunk_steps = chunk_groups * kernel_size;
    const size_t floats = ((2 * channels * kActStride + (threads / 64) * kWindow +
                            layers * channels + channels * out_kernel_size + 1 + 3) & ~3) +
                          2 * static_cast<size_t>(chunk_steps) * m_tiles * 64;
    const size_t lds = floats * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define EMPH_WORDS(KS)                                                              \
    do {                                                                            \
        auto kernel = m_tiles <= 5 ? word_decoder_kernel<KS, 5>                     \
                                   : word_decoder_kernel<KS, 8>;                    \
        static size_t reserved[2] = {64 * 1024, 64 * 1024};                         \
        if (lds > reserved[m_tiles <= 5]) {                                         \
            hipError_t status = hipFuncSetAttribute(                                \
                reinterpret_cast<const void*>(kernel),                              \
                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
            if (status != hipSuccess) {                                             \
                set_error("emph_word_decoder: cannot reserve %zu bytes of LDS", lds); \
                return static_cast<int>(status);                                    \
            }                                                                       \
            reserved[m_tiles <= 5] = lds;                                           \
        }                                                                           \
        hipLaunchKernelGGL(kernel, dim3(n_tiles), dim3(threads), lds, s, x, ldx,     \
                           tiles, block, halo, channels, packs, biases, layers,      \
                           activation, chunk_steps, out_weight, out_bias,            \
                           out_kernel_size, post, logits, scores);                   \
    } while (0)
    switch (kernel_size) {
        case 1: EMPH_WORDS(1); break;
        case 3: EMPH_WORDS(3); break;
        default: EMPH_WORDS(5); break;
    }
#undef EMPH_WORDS
    return check_launch("emph_word_decoder");
}

}  // extern "C"
