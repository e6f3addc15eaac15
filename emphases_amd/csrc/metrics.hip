// Evaluation metrics at word resolution: the masked
// reductions of emphases/evaluate/metrics.py:12-110 -
//   Metrics.update          mask_from_lengths + boolean indexing (metrics.py:36-38)
//   BinaryCrossEntropy      binary_cross_entropy_with_logits, or the clamped
//                           log form under LOSS = 'mse' (metrics.py:59-76)
//   MeanSquaredError        mse_loss(postprocess(logits), targets) (metrics.py:80-92)
//   PearsonCorrelation      sum((p - mean_p) (t - mean_t)) and the count
//                           (torchutil.metrics.PearsonCorrelation.update)
//   Statistics              count / sum / sum of squares of values
//                           (metrics.py:101-110 over torchutil.metrics.MeanStd)
// - over the packed word axis of a whole batch in one launch: the mask is the
// packed layout's own `word_segment` table (>= 0 on real words), so nothing is
// gathered.  HBM-bound: 12 bytes read per word column, no write but eight
// double-precision atomics per workgroup.  Sums are accumulated in float64 (the
// reference adds float32 batch sums into Python floats).
#include <math.h>

#include "common.h"

namespace emph {

__device__ __forceinline__ double wave_sum(double value) {
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1) value += __shfl_xor(value, offset);
    return value;
}

// accumulators: double [EMPH_METRIC_FIELDS]
__global__ __launch_bounds__(256) void word_metrics_kernel(
    const float* __restrict__ logits, const float* __restrict__ targets,
    const int32_t* __restrict__ word_segment, int64_t total, int post, float predicted_mean,
    float target_mean, double* __restrict__ accumulators) {
    double local[EMPH_METRIC_FIELDS];
#pragma unroll
    for (int i = 0; i < EMPH_METRIC_FIELDS; ++i) local[i] = 0.;
    for (int64_t index = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; index < total;
         index += static_cast<int64_t>(gridDim.x) * 256) {
        if (word_segment[index] < 0) continue;          // alignment padding column
        const float x = logits[index];
        const float y = targets[index];
        float score, bce;
        if (post == EMPH_POST_SIGMOID) {
            score = 1.f / (1.f + expf(-x));
            // binary_cross_entropy_with_logits: max(x, 0) - x y + log1p(exp(-|x|))
            bce = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        } else {
            score = post == EMPH_POST_CLAMP01 ? fminf(fmaxf(x, 0.f), 1.f) : x;
            const float c = fminf(fmaxf(x, 0.f), 1.f);
            bce = -(y * logf(c + 1e-6f) + (1.f - y) * logf(1.f - c + 1e-6f));
        }
        const float error = score - y;
        local[EMPH_METRIC_COUNT] += 1.;
        local[EMPH_METRIC_BCE] += bce;
        local[EMPH_METRIC_SQUARED_ERROR] += error * error;
        local[EMPH_METRIC_COVARIANCE] += (score - predicted_mean) * (y - target_mean);
        local[EMPH_METRIC_SUM_PREDICTED] += score;
        local[EMPH_METRIC_SUMSQ_PREDICTED] += static_cast<double>(score) * score;
        local[EMPH_METRIC_SUM_TARGET] += y;
        local[EMPH_METRIC_SUMSQ_TARGET] += static_cast<double>(y) * y;
    }
    __shared__ double partial[4][EMPH_METRIC_FIELDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < EMPH_METRIC_FIELDS; ++i) {
        const double sum = wave_sum(local[i]);
        if (lane == 0) partial[wave][i] = sum;
    }
    __syncthreads();
    if (threadIdx.x < EMPH_METRIC_FIELDS) {
        const double sum = partial[0][threadIdx.x] + partial[1][threadIdx.x] +
                           partial[2][threadIdx.x] + partial[3][threadIdx.x];
        if (sum != 0.) atomicAdd(accumulators + threadIdx.x, sum);
    }
}

}  // namespace emph

using namespace emph;

extern "C" {

int emph_word_metrics(const float* logits, const float* targets, const int32_t* word_segment,
                      int64_t total, int32_t post, float predicted_mean, float target_mean,
                      double* accumulators, void* stream) {
    if (total == 0) return EMPH_OK;
    EMPH_REQUIRE(logits && targets && word_segment && accumulators, EMPH_EINVAL,
                 "emph_word_metrics: null pointer");
    EMPH_REQUIRE(post >= EMPH_POST_NONE && post <= EMPH_POST_CLAMP01, EMPH_EINVAL,
                 "emph_word_metrics: unknown postprocess %d", post);
    EMPH_REQUIRE(total > 0, EMPH_EINVAL, "emph_word_metrics: negative size");
    const int64_t blocks = (total + 255) / 256;
    EMPH_LAUNCH(word_metrics_kernel, dim3(static_cast<unsigned>(blocks < 1024 ? blocks : 1024)),
                dim3(256), 0, static_cast<hipStream_t>(stream), logits, targets, word_segment,
                total, post, predicted_mean, target_mean, accumulators);
    return check_launch("emph_word_metrics");
}

}  // extern "C"
