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
// is C/16 x 4 MFMA tiles (fp32 v_mfma_f32_16x16x4_f32) shared by eight waves:
// wave w owns n-tile w & 3 and the lower (w < 4) or upper half of the m-tiles,
// so the two waves of a SIMD carry C/16 tiles between them whatever C is (a
// wave per m-tile put 3 waves on two SIMDs and 2 on the others: +20%).  The
// packed weights stream through a two-slot LDS ring by LDS-DMA, requested by a
// ninth wave one chunk (a third of a layer) ahead of the MFMAs, so each weight
// is read from L2 once per workgroup and never waited for; B fragments
// come from the LDS-resident activations, whose row stride of 80 floats keeps
// the four k-rows of a fragment on disjoint banks.
// Every layer re-applies the segment's own zero halo, so ragged batches keep
// the reference's B=1 edge semantics.
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "common.h"

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

// block = 64 (8 + kLoaderWaves) threads: eight MFMA waves + the loader waves; grid = n_tiles.
// MT = m-tiles per MFMA wave (half of C/16, rounded up).
//
// The loader wave: hipcc puts s_waitcnt vmcnt(0) in front of every LDS read
// that follows an LDS-DMA in program order (it cannot tell which LDS bytes the
// DMA writes), so a wave that both requests chunk q+1 and reads chunk q from
// LDS waits for q+1 to land first - the DMA latency (1.1 us per chunk,
// measured) ends up in series with the MFMAs instead of under them.  The MFMA
// waves never have a DMA in flight, so their waits are free.
// A single wave's LDS-DMA requests complete one after the other (conv_stack.hip:
// four waves keep a chunk ahead): a chunk is 46 KB here; with ONE loader wave the
// decoder took 32.4 us, with four 28.5 (round 5).  A third ring slot with chunks
// requested two ahead was tried and is slower (three smaller chunks per layer, one
// more barrier each: 34 us).
#ifndef EMPH_DECODER_LOADERS
#define EMPH_DECODER_LOADERS 4
#endif
constexpr int kLoaderWaves = EMPH_DECODER_LOADERS;

template <int KS, int MT>
__global__ __launch_bounds__(64 * (8 + kLoaderWaves)) void word_decoder_kernel(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ tiles,
    int block, int halo, int channels, const float* __restrict__ packs,
    const float* __restrict__ biases, int layers, int act, int chunk_trips,
    const float* __restrict__ out_weight, const float* __restrict__ out_bias,
    int out_kernel, int post, float* __restrict__ logits, float* __restrict__ scores) {
    extern __shared__ __align__(16) float lds[];
    const int m_tiles = channels >> 4;
    const int threads = blockDim.x;
    const int waves = threads >> 6;
    const int compute_waves = waves - kLoaderWaves;
    // (the two activation buffers are addressed as lds + index * size: a runtime-
    // indexed array of pointers would decay to flat addressing)
    const int buffer_floats = channels * kActStride;
    float* partial = lds + 2 * channels * kActStride;   // [waves][kWindow]
    float* bias_lds = partial + waves * kWindow;        // [layers][channels]
    float* out_lds = bias_lds + layers * channels;      // [channels][out_kernel] + 1
    // two weight chunks of chunk_trips trips each, 16-byte aligned
    float* ring = lds + ((2 * channels * kActStride + waves * kWindow + layers * channels +
                          channels * out_kernel + 1 + 3) & ~3);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4;
    const int col = lane & 15;

    const Tile span = load_tile(tiles, blockIdx.x);
    // A segment of 33 .. 2 (32 - halo) words is TWO tiles in the table
    // (emph_word_decoder_tiles: first = 0 and ceil(count / 2)), each a 32-position
    // window at one end of the segment that computes its half of the words with
    // the other half's nearest `halo` words as halo: two workgroups of two
    // 16-column tiles instead of one workgroup of four (a 10 s utterance has
    // 24-36 words: the ones past 32 used to set the kernel's time, 41 vs 32 us).
    const int reach = 32 - halo;                       // outputs next to ONE halo
    const bool halves = span.count > 32 && span.count <= 2 * reach;
    const bool second = halves && span.first != 0;
    const bool single = span.count <= kWindow && !halves;   // whole segment, no halo
    if (single && span.first != 0) return;            // covered by the first tile
    const int half = (span.count + 1) >> 1;
    const int first_out = halves ? (second ? half : 0) : span.first;
    const int last_out = halves   ? (second ? span.count : half)
                         : single ? span.count
                                  : min(span.first + block, span.count);
    // word index of window column 0
    const int start = halves ? (second ? span.count - 32 : 0) : single ? 0 : span.first - halo;
    // a window of at most 32 positions needs only two of the four 16-column tiles:
    // every wave then owns ONE of them and the per-layer MFMA work halves
    const bool narrow = halves || (single && span.count <= 32);

    // Weight stream: the decoder's packs (emph_word_decoder_pack), layer after
    // layer, cut into chunks of chunk_trips trips (a trip = four 4-row groups of
    // input channels x KS taps).  Chunk q is copied by LDS-DMA into ring[q & 1]
    // while chunk q-1 is being consumed.
    constexpr int HALO = (KS - 1) / 2;
    constexpr int kTripFloats = 4 * KS;               // per lane, m-tile and trip
    const int trips = channels >> 4;                  // per layer: C/4 groups / 4
    const int trip_floats = m_tiles * 64 * kTripFloats;
    const int chunks_per_layer = (trips + chunk_trips - 1) / chunk_trips;
    const int total_chunks = layers * chunks_per_layer;
    const int chunk_floats = chunk_trips * trip_floats;
    auto request = [&](int q) {                       // loader waves only
        const int layer = q / chunks_per_layer;
        const int first = (q - layer * chunks_per_layer) * chunk_trips;
        const int quads = min(chunk_trips, trips - first) * (trip_floats >> 2);
        const float* source =
            packs + (static_cast<int64_t>(layer) * trips + first) * trip_floats;
        float* target = ring + (q & 1) * chunk_floats;
        // (1 KB requests dealt over the loader waves)
        for (int base = 64 * (wave - compute_waves); base < quads;
             base += 64 * kLoaderWaves)               // trip_floats % 256 == 0
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(source + 4 * (base + lane)),
                (__attribute__((address_space(3))) void*)(target + 4 * base), 16, 0, 0);
    };
    const bool loader = wave >= compute_waves;
    if (total_chunks > 0 && loader) request(0);

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

    if (loader) {
        // one barrier per chunk, in step with the MFMA waves below: chunk q has
        // landed -> barrier (everyone is also done with chunk q-1, whose slot
        // chunk q+1 overwrites) -> request chunk q+1
        for (int q = 0; q < total_chunks; ++q) {
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            if (q + 1 < total_chunks) request(q + 1);
        }
    }

    // ---- decoder layers
    const int n_tile = narrow ? (wave & 1) : (wave & 3);
    const int part = narrow ? (wave >> 1) : (wave >> 2);
    const int split = (m_tiles + 1) >> 1;
    const int m_begin = narrow ? (m_tiles * part + 3) >> 2 : (part ? split : 0);
    const int m_end = narrow ? (m_tiles * (part + 1) + 3) >> 2 : (part ? m_tiles : split);
    const int m_count = m_end - m_begin;              // wave-uniform, <= MT
    // One instantiation of the layer loop per number of m-tiles a wave can own:
    // the accumulators stay plain registers (a wave-uniform `if (i < m_count)`
    // around each MFMA made hipcc copy them through temporaries with s_nop
    // padding: 19 us per layer instead of 5).
    auto run_layers = [&](auto count_tag) {
        constexpr int COUNT = decltype(count_tag)::value;
        int buffer = 0;
        for (int layer = 0; layer < layers; ++layer) {
            const float* source = lds + buffer * buffer_floats;
            float* target = lds + (buffer ^ 1) * buffer_floats;
            f32x4 acc[COUNT];
#pragma unroll
            for (int i = 0; i < COUNT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            // The layer is a sequence of trips (16 input channels x KS taps, i.e.
            // 4 KS MFMAs per m-tile).  The LDS reads of trip i+1 - KS 16-byte
            // reads per m-tile for A, the taps of four activation rows for B -
            // are threaded between the MFMAs of trip i: a wave can have only 15
            // LDS instructions in flight, so a block of reads in front of a
            // block of MFMAs leaves the matrix pipe idle while the reads drain
            // (measured 3.2 us of LDS time + 4.6 us of MFMA time per layer, in
            // series), and every wave of the workgroup leaves the chunk barrier
            // in the same phase.
            // Two register sets, read and consumed alternately (copying one set
            // into the other cost 48 v_mov per trip, in lockstep on both waves
            // of a SIMD, with the matrix pipe idle meanwhile).
            // (kernel_size 5 with three or four m-tiles per wave does not fit two
            // sets in 168 registers: it reads each trip right before using it)
            constexpr bool kTwoSets = KS * COUNT <= 12;
            f32x4 a_set[kTwoSets ? 2 : 1][COUNT][KS];
            float b_set[kTwoSets ? 2 : 1][4][KS];
            auto issue = [&](f32x4 (&a_out)[COUNT][KS], float (&b_out)[4][KS], int trip) {
                const int piece = trip / chunk_trips;
                const int q = layer * chunks_per_layer + piece;
                const float* fragment = ring + (q & 1) * chunk_floats +
                                        (trip - piece * chunk_trips) * trip_floats +
                                        (m_begin * 64 + lane) * kTripFloats;
                const float* b_base = source + (16 * trip + kk) * kActStride + kLeadCols +
                                      16 * n_tile + col - HALO;
#pragma unroll
                for (int i = 0; i < COUNT; ++i)
#pragma unroll
                    for (int v = 0; v < KS; ++v)
                        a_out[i][v] = *reinterpret_cast<const f32x4*>(
                            fragment + i * 64 * kTripFloats + 4 * v);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int tap = 0; tap < KS; ++tap)
                        b_out[t][tap] = b_base[4 * t * kActStride + tap];
            };
            auto step = [&](f32x4 (&a)[COUNT][KS], float (&b)[4][KS],
                            f32x4 (&a_out)[COUNT][KS], float (&b_out)[4][KS], int trip) {
                if (kTwoSets) {
                    const int next = min(trip + 1, trips - 1);
                    if (trip + 1 < trips && (trip + 1) % chunk_trips == 0)
                        __syncthreads();              // the next chunk has landed
                    __builtin_amdgcn_sched_barrier(0);
                    issue(a_out, b_out, next);        // (the last trip re-reads itself)
                } else {
                    if (trip > 0 && trip % chunk_trips == 0) __syncthreads();
                    issue(a_out, b_out, trip);        // a_out aliases a
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int tap = 0; tap < KS; ++tap)
#pragma unroll
                        for (int i = 0; i < COUNT; ++i) {
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                a[i][(t * KS + tap) >> 2][(t * KS + tap) & 3], b[t][tap],
                                acc[i], 0, 0, 0);
                        }
                // (COUNT + 4) KS reads behind the first MFMAs, one per MFMA, so
                // that the last of them has the rest of the trip to land
#pragma unroll
                for (int k = 0; k < (COUNT + 4) * KS; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            __syncthreads();                          // chunk (layer, 0) is in the ring
            if (kTwoSets) {
                issue(a_set[0], b_set[0], 0);
                for (int trip = 0; trip < trips; trip += 2) {
                    step(a_set[0], b_set[0], a_set[kTwoSets], b_set[kTwoSets], trip);
                    if (trip + 1 < trips)
                        step(a_set[kTwoSets], b_set[kTwoSets], a_set[0], b_set[0], trip + 1);
                }
            } else {
                for (int trip = 0; trip < trips; ++trip)
                    step(a_set[0], b_set[0], a_set[0], b_set[0], trip);
            }
            // bias + activation; re-apply the segment's zero halo.  All bias
            // reads first, the activation switch outside the element loops (an
            // LDS read, a wait and a branch ladder per element took 1.2 us)
            const float* bias = bias_lds + layer * channels + 16 * m_begin + 4 * kk;
            const int p = 16 * n_tile + col;
            const int word = start + p;
            const bool inside = word >= 0 && word < span.count;
            f32x4 value[COUNT];
#pragma unroll
            for (int i = 0; i < COUNT; ++i)
                value[i] = acc[i] + *reinterpret_cast<const f32x4*>(bias + 16 * i);
#define EMPH_APPLY(EXPRESSION)                                   \
    _Pragma("unroll") for (int i = 0; i < COUNT; ++i)            \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {          \
            const float v = value[i][r];                         \
            value[i][r] = (EXPRESSION);                          \
        }
            switch (act) {
                case EMPH_ACT_RELU: EMPH_APPLY(v < 0.f ? 0.f : v); break;
                case EMPH_ACT_GELU:
                    EMPH_APPLY(0.5f * v * (1.f + erff(v * 0.70710678118654752440f)));
                    break;
                case EMPH_ACT_SILU: EMPH_APPLY(v / (1.f + expf(-v))); break;
                case EMPH_ACT_LEAKY_RELU: EMPH_APPLY(v > 0.f ? v : 0.01f * v); break;
                default: break;
            }
#undef EMPH_APPLY
#pragma unroll
            for (int i = 0; i < COUNT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * (m_begin + i) + 4 * kk + r;
                    target[c * kActStride + kLeadCols + p] = inside ? value[i][r] : 0.f;
                }
            buffer ^= 1;
        }
    };
    if (!loader) {
        // (a wave with no m-tile still has to keep step with the barriers)
        if (m_count >= 4 && MT >= 4) run_layers(std::integral_constant<int, MT >= 4 ? 4 : 1>{});
        else if (m_count == 3) run_layers(std::integral_constant<int, 3>{});
        else if (m_count == 2) run_layers(std::integral_constant<int, 2>{});
        else if (m_count == 1) run_layers(std::integral_constant<int, 1>{});
        else
            for (int q = 0; q < total_chunks; ++q) __syncthreads();
    }
    int current = layers & 1;
    __syncthreads();

    // ---- output projection: channels are split over the waves, then summed
    {
        const float* source = lds + current * buffer_floats;
        const int halo_out = (out_kernel - 1) / 2;
        float sum = 0.f;
        for (int c = wave; c < channels && !loader; c += compute_waves)
            for (int tap = 0; tap < out_kernel; ++tap)
                sum = fmaf(out_lds[c * out_kernel + tap],
                           source[c * kActStride + kLeadCols + lane + tap - halo_out], sum);
        partial[wave * kWindow + lane] = sum;
        __syncthreads();
        if (wave == 0) {
            float total = out_lds[channels * out_kernel];
            for (int w = 0; w < compute_waves; ++w) total += partial[w * kWindow + lane];
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
    }
}

}  // namespace emph

using namespace emph;

extern "C" {

int64_t emph_word_decoder_pack_size(int32_t channels, int32_t kernel_size) {
    return static_cast<int64_t>(channels) * channels * kernel_size;
}

// pack[trip][m][lane][4 t + tap... as (t * KS + tap)] =
//     weight[16 m + (lane & 15)][16 trip + 4 t + (lane >> 4)][tap]
int emph_word_decoder_pack(const float* host_weight, int32_t channels, int32_t kernel_size,
                           float* host_pack) {
    EMPH_REQUIRE(host_weight && host_pack, EMPH_EINVAL,
                 "emph_word_decoder_pack: null pointer");
    EMPH_REQUIRE(channels >= 16 && channels <= 128 && channels % 16 == 0, EMPH_ERANGE,
                 "emph_word_decoder_pack: channels %d not a multiple of 16 in 16..128",
                 channels);
    EMPH_REQUIRE(kernel_size == 1 || kernel_size == 3 || kernel_size == 5, EMPH_ERANGE,
                 "emph_word_decoder_pack: kernel_size %d not in {1,3,5}", kernel_size);
    const int m_tiles = channels / 16, trips = channels / 16;
    for (int trip = 0; trip < trips; ++trip)
        for (int m = 0; m < m_tiles; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < 4; ++t)
                    for (int tap = 0; tap < kernel_size; ++tap) {
                        const int co = 16 * m + (lane & 15);
                        const int ci = 16 * trip + 4 * t + (lane >> 4);
                        host_pack[((static_cast<int64_t>(trip) * m_tiles + m) * 64 + lane) *
                                      (4 * kernel_size) +
                                  t * kernel_size + tap] =
                            host_weight[(static_cast<int64_t>(co) * channels + ci) * kernel_size +
                                        tap];
                    }
    return EMPH_OK;
}

int32_t emph_word_decoder_block(int32_t layers, int32_t kernel_size,
                                int32_t out_kernel_size) {
    const int halo = layers * ((kernel_size - 1) / 2) + (out_kernel_size - 1) / 2;
    return kWindow - 2 * halo;
}

// The decoder's tile table (host): int32 [n][4] = (segment, first word, segment's
// first column, segment's words).  A segment of at most 32 words, or of 2 (32 -
// halo) + 1 .. 64, is one tile; 33 .. 2 (32 - halo) words are two (first = 0 and
// ceil(count / 2)); longer segments one tile per `block` words.  Returns the number
// of tiles (host_tiles may be NULL to count them).
int32_t emph_word_decoder_tiles(const int64_t* host_counts, const int64_t* host_offsets,
                                int32_t segments, int32_t layers, int32_t kernel_size,
                                int32_t out_kernel_size, int32_t* host_tiles) {
    const int block = emph_word_decoder_block(layers, kernel_size, out_kernel_size);
    if (host_counts == nullptr || host_offsets == nullptr || block < 16) return -1;
    const int halo = (kWindow - block) / 2;
    int32_t total = 0;
    auto emit = [&](int segment, int64_t first) {
        if (host_tiles != nullptr) {
            int32_t* row = host_tiles + 4 * static_cast<int64_t>(total);
            row[0] = segment;
            row[1] = static_cast<int32_t>(first);
            row[2] = static_cast<int32_t>(host_offsets[segment]);
            row[3] = static_cast<int32_t>(host_counts[segment]);
        }
        ++total;
    };
    for (int segment = 0; segment < segments; ++segment) {
        const int64_t count = host_counts[segment];
        if (count <= 0) continue;
        if (count > 32 && count <= 2 * (32 - halo)) {
            emit(segment, 0);
            emit(segment, (count + 1) / 2);
        } else if (count <= kWindow) {
            emit(segment, 0);
        } else {
            for (int64_t first = 0; first < count; first += block) emit(segment, first);
        }
    }
    return total;
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
    const int threads = 64 * (8 + kLoaderWaves);      // 8 MFMA waves + the loader waves
    // two weight chunks of whole trips in whatever LDS the activations leave
    // (80 channels: 3 + 2 trips per layer, i.e. one chunk barrier per layer;
    // each barrier costs about 0.3 us)
    const size_t fixed_floats = (2 * channels * kActStride + (threads / 64) * kWindow +
                                 layers * channels + channels * out_kernel_size + 1 + 3) & ~3;
    size_t ring_budget = 156 * 1024 - fixed_floats * sizeof(float);
    const int trips = channels / 16;
    const size_t trip_bytes = static_cast<size_t>(m_tiles) * 64 * 4 * kernel_size * sizeof(float);
    int chunk_trips = static_cast<int>(ring_budget / 2 / trip_bytes);
    EMPH_REQUIRE(chunk_trips >= 1, EMPH_ERANGE,
                 "emph_word_decoder: no LDS left for the weight ring (%d channels, %d layers)",
                 channels, layers);
    if (chunk_trips > trips) chunk_trips = trips;
    const int pieces = (trips + chunk_trips - 1) / chunk_trips;
    chunk_trips = (trips + pieces - 1) / pieces;
    const size_t lds = fixed_floats * sizeof(float) + 2 * chunk_trips * trip_bytes;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define EMPH_WORDS(KS)                                                              \
    do {                                                                            \
        auto kernel = m_tiles <= 6 ? word_decoder_kernel<KS, 3>                     \
                                   : word_decoder_kernel<KS, 4>;                    \
        static LdsReservation reserved[2];                                                     \
        if (int status = reserve_lds(reserved[m_tiles <= 6], reinterpret_cast<const void*>(kernel), lds,\
                                     "emph_word_decoder"))                                     \
            return status;                                                                     \
        EMPH_LAUNCH(kernel, dim3(n_tiles), dim3(threads), lds, s, x, ldx,     \
                           tiles, block, halo, channels, packs, biases, layers,      \
                           activation, chunk_trips, out_weight, out_bias,            \
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
