// Feature rows that do not come from the STFT front-end.
//
// emph_pitch_rows writes the pitch / periodicity rows of the feature matrix
// (emphases/data/preprocess/core.py:83-113): the pitch tracker itself (`penn`,
// a third-party neural network) runs outside this library and hands over its
// per-frame outputs; what the reference does with them —
//     log2(pitch)                                   (core.py:103)
//     (log2(pitch) - LOGFMIN) / (LOGFMAX - LOGFMIN) (core.py:98-101, NORMALIZE)
//     periodicity                                   (core.py:105-106)
// and the row concatenation of core.py:123 — happens here, in place in the
// [rows, ld] feature matrix the convolutions read.  HBM-bound elementwise work:
// 4-8 B read + 4-8 B written per frame, one 16-byte access per lane.
#include <math.h>

#include "common.h"

namespace emph {

__global__ __launch_bounds__(256) void pitch_rows_kernel(
    const float* __restrict__ pitch, const float* __restrict__ periodicity,
    float* __restrict__ pitch_out, float* __restrict__ periodicity_out, int64_t columns,
    int normalize, float logfmin, float span) {
    const int64_t first = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (first >= columns) return;
    const bool whole = first + 4 <= columns;
    auto row = [&](const float* in, float* out, bool is_pitch) {
        float value[4];
        if (whole) {
            const float4 v = *reinterpret_cast<const float4*>(in + first);
            value[0] = v.x; value[1] = v.y; value[2] = v.z; value[3] = v.w;
        } else {
            for (int i = 0; i < 4; ++i) value[i] = first + i < columns ? in[first + i] : 1.f;
        }
        if (is_pitch) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float l = log2f(value[i]);
                value[i] = normalize ? (l - logfmin) / span : l;
            }
        }
        if (whole) {
            *reinterpret_cast<float4*>(out + first) =
                float4{value[0], value[1], value[2], value[3]};
        } else {
            for (int i = 0; i < 4; ++i)
                if (first + i < columns) out[first + i] = value[i];
        }
    };
    if (pitch_out != nullptr) row(pitch, pitch_out, true);
    if (periodicity_out != nullptr) row(periodicity, periodicity_out, false);
}

// Polyphase windowed-sinc resampling of every utterance of a batch
// (emphases/core.py:613-619 -> torchaudio.transforms.Resample, whose strided
// conv1d this restates): output sample m = i * new_rate + phase of an utterance
// is sum_k kernel[phase][k] * x[i * orig + k - width], zero outside the
// utterance.  One thread per output sample, the taps of a phase streamed from
// L2 (a warp's lanes hold consecutive phases: the kernel table is read once per
// 64 outputs, the audio window is shared).  grid = (blocks, utterances).
template <bool PCM>
__global__ __launch_bounds__(256) void resample_kernel(
    const void* __restrict__ audio, const int64_t* __restrict__ table,
    const float* __restrict__ kernel, int orig, int fresh, int width, int taps,
    int n_utterances, float* __restrict__ out) {
  // (grid.y is capped at 65 535: a block row walks the utterances with that stride)
  for (int utterance = blockIdx.y; utterance < n_utterances; utterance += gridDim.y) {
    const int64_t* row = table + static_cast<int64_t>(utterance) * 4;
    const int64_t source = row[0], length = row[1], target = row[2], count = row[3];
    for (int64_t m = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; m < count;
         m += static_cast<int64_t>(gridDim.x) * 256) {
        const int64_t i = m / fresh;
        const int phase = static_cast<int>(m - i * fresh);
        const int64_t base = i * orig - width;
        const float* weights = kernel + static_cast<int64_t>(phase) * taps;
        float acc = 0.f;
        for (int k = 0; k < taps; ++k) {
            const int64_t at = base + k;
            if (at < 0 || at >= length) continue;
            const float x = PCM ? static_cast<float>(
                                      static_cast<const int16_t*>(audio)[source + at]) *
                                      (1.f / 32768.f)
                                : static_cast<const float*>(audio)[source + at];
            acc = fmaf(weights[k], x, acc);
        }
        out[target + m] = acc;
    }
  }
}

}  // namespace emph

using namespace emph;

extern "C" {

int emph_pitch_rows(const float* pitch, const float* periodicity, float* out, int64_t ld,
                    int32_t pitch_row, int32_t periodicity_row, int32_t normalize,
                    float logfmin, float logfmax, void* stream) {
    if (ld == 0 || (pitch_row < 0 && periodicity_row < 0)) return EMPH_OK;
    EMPH_REQUIRE(out != nullptr, EMPH_EINVAL, "emph_pitch_rows: null output");
    EMPH_REQUIRE(pitch_row < 0 || pitch != nullptr, EMPH_EINVAL,
                 "emph_pitch_rows: pitch row requested without pitch input");
    EMPH_REQUIRE(periodicity_row < 0 || periodicity != nullptr, EMPH_EINVAL,
                 "emph_pitch_rows: periodicity row requested without periodicity input");
    EMPH_REQUIRE((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(pitch) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(periodicity) & 15) == 0,
                 EMPH_EINVAL, "emph_pitch_rows: rows must be 16-byte aligned (ld %% 4 == 0)");
    EMPH_REQUIRE(!normalize || logfmax > logfmin, EMPH_EINVAL,
                 "emph_pitch_rows: LOGFMAX must exceed LOGFMIN");
    const unsigned blocks = static_cast<unsigned>((ld / 4 + 255) / 256);
    EMPH_LAUNCH(
        pitch_rows_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pitch,
        periodicity, pitch_row < 0 ? nullptr : out + static_cast<int64_t>(pitch_row) * ld,
        periodicity_row < 0 ? nullptr : out + static_cast<int64_t>(periodicity_row) * ld, ld,
        normalize, logfmin, logfmax - logfmin);
    return check_launch("emph_pitch_rows");
}

int emph_resample(const void* audio, int32_t audio_format, const int64_t* table,
                  int32_t n_utterances, int64_t most_samples, const float* kernel,
                  int32_t orig, int32_t fresh, int32_t width, float* out, void* stream) {
    if (n_utterances == 0 || most_samples == 0) return EMPH_OK;
    EMPH_REQUIRE(audio && table && kernel && out, EMPH_EINVAL, "emph_resample: null pointer");
    EMPH_REQUIRE(audio_format == EMPH_AUDIO_F32 || audio_format == EMPH_AUDIO_PCM16,
                 EMPH_EINVAL, "emph_resample: unknown audio format %d", audio_format);
    EMPH_REQUIRE(orig > 0 && fresh > 0 && width >= 0, EMPH_EINVAL,
                 "emph_resample: rates %d -> %d, width %d", orig, fresh, width);
    const int taps = 2 * width + orig;
    const int64_t blocks = (most_samples + 255) / 256;
    dim3 grid(static_cast<unsigned>(blocks < 4096 ? blocks : 4096),
              static_cast<unsigned>(n_utterances < 65535 ? n_utterances : 65535));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (audio_format == EMPH_AUDIO_PCM16)
        EMPH_LAUNCH(resample_kernel<true>, grid, dim3(256), 0, s, audio, table, kernel, orig,
                    fresh, width, taps, n_utterances, out);
    else
        EMPH_LAUNCH(resample_kernel<false>, grid, dim3(256), 0, s, audio, table, kernel, orig,
                    fresh, width, taps, n_utterances, out);
    return check_launch("emph_resample");
}

}  // extern "C"
