// Frame -> word resampling and the scalar output projection.
//
// emph_segment_reduce replaces the Python double loop of emphases.downsample
// (emphases/core.py:426-469: one slice + reduce + copy, i.e. >= 3 kernel
// launches and a host sync, per word).  It is HBM-bound: every frame embedding
// is read exactly once (320 B per frame for 80 channels) in 64-byte row
// segments; one wave owns one word, lanes are 16 frames x 4 channels, every
// load of a 32-frame round is in flight at once, and the frame axis is folded
// with a 16-lane butterfly.
//
// emph_output_layer replaces Conv1d(channels, 1, k, 'same')
// (emphases/model/core.py:33-37,138) fused with emphases.postprocess
// (emphases/core.py:335-342).
#include <math.h>

#include "common.h"

namespace emph {

// grid.x = blocks of 4 words over the packed word axis, grid.y = slices of
// CGROUPS channel groups of 4; block = 256 (one wave per word and slice).  A
// 10 s utterance has ~30 words, so a wave per word alone leaves most of the
// chip without a wave: the channel slices give 4-5x as many.
template <int CGROUPS>
__global__ __launch_bounds__(256) void segment_reduce_kernel(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ bounds,
    float* __restrict__ out, int64_t ldw, int channels,
    const int64_t* __restrict__ seg, const int32_t* __restrict__ word_segment,
    int64_t total_words, int mode) {
    const int lane = threadIdx.x & 63;
    const int64_t word = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (word >= total_words) return;
    const int segment = word_segment[word];
    if (segment < 0) return;                      // alignment padding column
    const int raw_start = bounds[word];
    const int raw_end = bounds[ldw + word];
    const int64_t* row = seg + static_cast<int64_t>(segment) * EMPH_SEG_FIELDS;
    const int64_t frame_off = row[EMPH_SEG_FRAME_OFF];
    const int frames = static_cast<int>(row[EMPH_SEG_FRAMES]);
    // Python slice semantics: clamp to the chunk (core.py:446-454)
    const int start = max(0, min(raw_start, frames));
    const int end = max(start, min(raw_end, frames));

    const int fr = lane & 15;
    const int cg = lane >> 4;
    const int group0 = blockIdx.y * CGROUPS;
    if (mode == EMPH_REDUCE_CENTER) {
        // gather at (start + end) // 2 of the UNclamped bounds (core.py:459-466)
        const int center = (raw_start + raw_end) >> 1;
        for (int c = 4 * group0 + lane; c < min(channels, 4 * (group0 + CGROUPS)); c += 64)
            out[static_cast<int64_t>(c) * ldw + word] =
                (center >= 0 && center < frames)
                    ? x[static_cast<int64_t>(c) * ldx + frame_off + center]
                    : 0.f;
        return;
    }
    const int groups = (channels + 3) >> 2;
    const float identity = mode == EMPH_REDUCE_MAX ? -INFINITY : 0.f;
    const float* base = x + frame_off;
    float acc[CGROUPS];
#pragma unroll
    for (int g = 0; g < CGROUPS; ++g) acc[g] = identity;
    // 32 frames x every channel group per round, all loads unconditional
    // (clamped address, masked value) so that they are all in flight together:
    // a dependent L2/HBM round trip costs 1-2 us here
    for (int t0 = start; t0 < end; t0 += 32) {
        float value[2][CGROUPS];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t = min(t0 + 16 * half + fr, max(end - 1, 0));
#pragma unroll
            for (int g = 0; g < CGROUPS; ++g) {
                const int c = min(4 * min(group0 + g, groups - 1) + cg, channels - 1);
                value[half][g] = base[static_cast<int64_t>(c) * ldx + t];
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const bool live = t0 + 16 * half + fr < end;
#pragma unroll
            for (int g = 0; g < CGROUPS; ++g) {
                const float v = live ? value[half][g] : identity;
                acc[g] = mode == EMPH_REDUCE_MAX ? fmaxf(acc[g], v) : acc[g] + v;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < CGROUPS; ++g) {
        float value = acc[g];
#pragma unroll
        for (int offset = 8; offset > 0; offset >>= 1) {
            const float other = __shfl_xor(value, offset);
            value = mode == EMPH_REDUCE_MAX ? fmaxf(value, other) : value + other;
        }
        if (mode == EMPH_REDUCE_AVERAGE)
            value = value / static_cast<float>(end - start);   // 0/0 = NaN
        const int c = 4 * (group0 + g) + cg;
        if (fr == 0 && group0 + g < groups && c < channels)
            out[static_cast<int64_t>(c) * ldw + word] = value;
    }
}

// The per-word sums from the running sums the last frame-rate layer left behind
// (conv1d_winograd4_kernel<..., WORD_SUMS>): word w = a handful of signed terms,
// terms[first[w] .. first[w + 1]): a slot s >= 0 adds sums[s][:], ~s < 0
// subtracts it (the running sum at the frame in front of the word's part of a
// 64-frame tile), in the order of the table - a fixed order, so the result does
// not depend on the launch.  A thread owns four channels of a word (one 16-byte
// load per term); out is the [channels][ldw] layout emph_segment_reduce writes.
__global__ __launch_bounds__(256) void word_sums_kernel(
    const float* __restrict__ sums, int64_t ld_sums, const int32_t* __restrict__ terms,
    const int32_t* __restrict__ first, const int32_t* __restrict__ lengths,
    float* __restrict__ out, int64_t ldw, int channels, int64_t columns, int mode) {
    const int groups = channels >> 2;
    const int64_t index = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t word = index / groups;
    const int group = static_cast<int>(index - word * groups);
    if (word >= columns) return;
    const int begin = first[word], end = first[word + 1];
    const int frames = lengths[word];
    if (frames < 0) return;                       // alignment padding column
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = begin; k < end; ++k) {
        const int term = terms[k];
        const int slot = term < 0 ? ~term : term;
        const float4 v = *reinterpret_cast<const float4*>(
            sums + static_cast<int64_t>(slot) * ld_sums + 4 * group);
        if (term < 0) {
            acc.x -= v.x, acc.y -= v.y, acc.z -= v.z, acc.w -= v.w;
        } else {
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
    }
    if (mode == EMPH_REDUCE_AVERAGE) {
        const float count = static_cast<float>(frames);    // 0 / 0 = NaN, as torch.mean
        acc.x /= count, acc.y /= count, acc.z /= count, acc.w /= count;
    }
    float* target = out + static_cast<int64_t>(4 * group) * ldw + word;
    target[0] = acc.x;
    target[ldw] = acc.y;
    target[2 * ldw] = acc.z;
    target[3 * ldw] = acc.w;
}

// One workgroup per piece: copy `length` columns of every row, zero the rest
// of the piece's `padded` columns.
__global__ __launch_bounds__(256) void gather_columns_kernel(
    const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
    int channels, const int64_t* __restrict__ pieces) {
    const int64_t* piece = pieces + static_cast<int64_t>(blockIdx.x) * 4;
    const int64_t source = piece[0], target = piece[2];
    const int length = static_cast<int>(piece[1]);
    const int padded = static_cast<int>(piece[3]);
    for (int index = threadIdx.x; index < channels * padded; index += 256) {
        const int c = index / padded;
        const int t = index - c * padded;
        y[static_cast<int64_t>(c) * ldy + target + t] =
            t < length ? x[static_cast<int64_t>(c) * ldx + source + t] : 0.f;
    }
}

// One wave per position of the packed axis (a thread per position walked 240
// strided loads one after the other: 60 us for 1 882 words); lanes split the
// channels, then a butterfly sum.
__global__ __launch_bounds__(256) void output_layer_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ weight,
    const float* __restrict__ bias, int channels, int kernel_size,
    const int64_t* __restrict__ seg, const int32_t* __restrict__ position_segment,
    int64_t total, int axis, int post, float* __restrict__ logits,
    float* __restrict__ scores) {
    extern __shared__ float w[];
    for (int index = threadIdx.x; index < channels * kernel_size; index += 256)
        w[index] = weight[index];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t position = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (position >= total) return;
    const int segment = position_segment[position];
    if (segment < 0) return;
    const Span span = load_span(seg, segment, axis);
    const int t = static_cast<int>(position - span.offset);
    const int halo = (kernel_size - 1) / 2;
    float acc = 0.f;
    for (int c = lane; c < channels; c += 64) {
        const float* src = x + static_cast<int64_t>(c) * ldx + position;
        for (int tap = 0; tap < kernel_size; ++tap) {
            const int u = t + tap - halo;
            if (u >= 0 && u < span.count)
                acc = fmaf(w[c * kernel_size + tap], src[tap - halo], acc);
        }
    }
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1) acc += __shfl_xor(acc, offset);
    if (lane != 0) return;
    acc += bias[0];
    if (logits != nullptr) logits[position] = acc;
    if (scores != nullptr) {
        float value = acc;
        if (post == EMPH_POST_SIGMOID) value = 1.f / (1.f + expf(-acc));
        if (post == EMPH_POST_CLAMP01) value = fminf(fmaxf(acc, 0.f), 1.f);
        scores[position] = value;
    }
}

}  // namespace emph

using namespace emph;

extern "C" {

int emph_segment_reduce(const float* x, int64_t ldx, const int32_t* bounds,
                        float* out, int64_t ldw, int32_t channels,
                        const int64_t* seg, const int32_t* word_segment,
                        int64_t total_words, int32_t mode, void* stream) {
    if (total_words == 0) return EMPH_OK;
    EMPH_REQUIRE(x && bounds && out && seg && word_segment, EMPH_EINVAL,
                 "emph_segment_reduce: null pointer");
    EMPH_REQUIRE(mode >= EMPH_REDUCE_SUM && mode <= EMPH_REDUCE_CENTER, EMPH_EINVAL,
                 "emph_segment_reduce: unknown mode %d", mode);
    EMPH_REQUIRE(channels > 0 && total_words <= ldw, EMPH_EINVAL,
                 "emph_segment_reduce: bad shape");
    EMPH_REQUIRE(channels <= 128, EMPH_ERANGE,
                 "emph_segment_reduce: channels %d > 128", channels);
    const unsigned blocks = static_cast<unsigned>((total_words + 3) / 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned slices = static_cast<unsigned>(((channels + 3) / 4 + 4) / 5);
    EMPH_LAUNCH(segment_reduce_kernel<5>, dim3(blocks, slices), dim3(256), 0, s, x, ldx,
                       bounds, out, ldw, channels, seg, word_segment, total_words, mode);
    return check_launch("emph_segment_reduce");
}

int emph_word_sums(const float* sums, int64_t ld_sums, const int32_t* terms,
                   const int32_t* first, const int32_t* lengths, float* out, int64_t ldw,
                   int32_t channels, int64_t columns, int32_t mode, void* stream) {
    if (columns == 0) return EMPH_OK;
    EMPH_REQUIRE(sums && terms && first && lengths && out, EMPH_EINVAL,
                 "emph_word_sums: null pointer");
    EMPH_REQUIRE(mode == EMPH_REDUCE_SUM || mode == EMPH_REDUCE_AVERAGE, EMPH_EINVAL,
                 "emph_word_sums: mode %d (sum or average)", mode);
    EMPH_REQUIRE(channels > 0 && channels % 4 == 0 && ld_sums >= channels &&
                     ld_sums % 4 == 0 && columns <= ldw &&
                     (reinterpret_cast<uintptr_t>(sums) & 15) == 0,
                 EMPH_EINVAL, "emph_word_sums: bad shape");
    const int64_t threads = columns * (channels / 4);
    EMPH_LAUNCH(word_sums_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)),
                dim3(256), 0, static_cast<hipStream_t>(stream), sums, ld_sums, terms, first,
                lengths, out, ldw, channels, columns, mode);
    return check_launch("emph_word_sums");
}

int emph_gather_columns(const float* x, int64_t ldx, float* y, int64_t ldy,
                        int32_t channels, const int64_t* pieces, int32_t n_pieces,
                        void* stream) {
    if (n_pieces == 0) return EMPH_OK;
    EMPH_REQUIRE(x && y && pieces, EMPH_EINVAL, "emph_gather_columns: null pointer");
    EMPH_REQUIRE(channels > 0 && n_pieces > 0, EMPH_EINVAL,
                 "emph_gather_columns: bad shape");
    EMPH_LAUNCH(gather_columns_kernel, dim3(n_pieces), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, y, ldy, channels, pieces);
    return check_launch("emph_gather_columns");
}

int emph_output_layer(const float* x, int64_t ldx, const float* weight,
                      const float* bias, int32_t channels, int32_t kernel_size,
                      const int64_t* seg, const int32_t* position_segment,
                      int64_t total, int32_t axis, int32_t post, float* logits,
                      float* scores, void* stream) {
    if (total == 0) return EMPH_OK;
    EMPH_REQUIRE(x && weight && bias && seg && position_segment, EMPH_EINVAL,
                 "emph_output_layer: null pointer");
    EMPH_REQUIRE(kernel_size >= 1 && kernel_size <= 7 && (kernel_size & 1),
                 EMPH_ERANGE, "emph_output_layer: kernel_size %d", kernel_size);
    EMPH_REQUIRE(channels > 0 && channels * kernel_size <= 8192, EMPH_ERANGE,
                 "emph_output_layer: channels %d", channels);
    const unsigned blocks = static_cast<unsigned>((total + 3) / 4);
    EMPH_LAUNCH(output_layer_kernel, dim3(blocks), dim3(256),
                       channels * kernel_size * sizeof(float),
                       static_cast<hipStream_t>(stream), x, ldx, weight, bias,
                       channels, kernel_size, seg, position_segment, total, axis,
                       post, logits, scores);
    return check_launch("emph_output_layer");
}

}  // extern "C"
