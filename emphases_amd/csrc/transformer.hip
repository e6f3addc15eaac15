// Transformer encoder pieces (emphases/model/layers/transformer.py:13-52):
// positional encoding, the attention core and the post-LN residual.  The
// linear layers (in_proj, out_proj, linear1, linear2) are kernel_size-1 calls
// of emph_conv1d, so activations stay in the packed [channels, positions]
// layout end to end (the reference permutes to [T, B, C] and back).
//
// Attention (fp32 MFMA, scores never materialised; the reference's bmm writes a
// 2 x T x T fp32 matrix per utterance and layer):
//   S^T = K Q^T   A = K[key][d]  (keys on the lane's low bits -> coalesced
//                 reads of the d-major K rows), B = Q^T pre-scaled by 1/sqrt(d)
//   O^T = V^T P^T A = V^T[d][key] read from the position-major V buffer,
//                 B = P^T = the S^T accumulator registers as they stand: with
//                 k-step r taking keys {4g + r}, the MFMA D-layout (row =
//                 4*(lane>>4) + r) IS the B-operand layout, so probabilities
//                 never move between lanes or through LDS.
//   Online softmax statistics live per lane because the query is the MFMA
//   column (lane & 15) in both products.
#include <math.h>

#include <type_traits>

#include "common.h"

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid = n_tiles (blocks of `tile_n` positions); block = 256
__global__ __launch_bounds__(256) void add_position_kernel(
    float* __restrict__ x, int64_t ldx, const float* __restrict__ table,
    int channels, int max_positions, const int32_t* __restrict__ tiles,
    int tile_n) {
    const Tile span = load_tile(tiles, blockIdx.x);
    const int t0 = span.first;
    const int count = min(tile_n, span.count - t0);
    for (int index = threadIdx.x; index < channels * tile_n; index += 256) {
        const int c = index / tile_n;
        const int i = index - c * tile_n;
        if (i < count && t0 + i < max_positions)
            x[static_cast<int64_t>(c) * ldx + span.offset + t0 + i] +=
                table[static_cast<int64_t>(t0 + i) * channels + c];
    }
}

// y = LayerNorm(x + r) over channels.  A workgroup is 64 columns x 4 channel
// slices (wave w owns channels w, w + 4, ...: 256-byte row segments); the two
// reductions (mean, then centred squares, as torch does) cross the waves
// through LDS.  One thread per column left a 64 x 1000-frame batch with one
// wave per SIMD and 160 loads per lane: 33 us for 61 MB.
template <int CMAX>
__global__ __launch_bounds__(256) void add_layernorm_kernel(
    const float* x, const float* __restrict__ r, float* y,
    int64_t ld, int channels, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, int64_t first, int64_t count) {
    constexpr int PER = (CMAX + 3) / 4;
    __shared__ float partial[2][4][64];
    const int lane = threadIdx.x & 63;
    const int slice = threadIdx.x >> 6;
    const int64_t index = static_cast<int64_t>(blockIdx.x) * 64 + lane;
    const bool live = index < count;
    const int64_t column = first + (live ? index : count - 1);
    float v[PER];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = 4 * j + slice;
        v[j] = 0.f;
        if (c < channels) {
            v[j] = x[static_cast<int64_t>(c) * ld + column] +
                   r[static_cast<int64_t>(c) * ld + column];
            sum += v[j];
        }
    }
    partial[0][slice][lane] = sum;
    __syncthreads();
    const float mean = (partial[0][0][lane] + partial[0][1][lane] + partial[0][2][lane] +
                        partial[0][3][lane]) / static_cast<float>(channels);
    float square = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (4 * j + slice < channels) square = fmaf(v[j] - mean, v[j] - mean, square);
    partial[1][slice][lane] = square;
    __syncthreads();
    const float variance = (partial[1][0][lane] + partial[1][1][lane] + partial[1][2][lane] +
                            partial[1][3][lane]) / static_cast<float>(channels);
    const float rstd = 1.f / sqrtf(variance + eps);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = 4 * j + slice;
        if (c < channels)
            y[static_cast<int64_t>(c) * ld + column] =
                (v[j] - mean) * rstd * gamma[c] + beta[c];
    }
}

// Reductions over the four 16-lane rows of a wave (the k index of an MFMA
// fragment) on gfx950's v_permlane16_swap / v_permlane32_swap: with both
// operands equal, the swap leaves (row 0, row 0, row 2, row 2) in one register
// and (row 1, row 1, row 3, row 3) in the other, so one VALU op combines rows
// pairwise - no trip through the LDS crossbar (ds_bpermute) and its wait.
__device__ __forceinline__ float rows_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// grid = (n_tiles, heads); block = 64 (one wave = 64 queries of one head)
template <int D>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_kernel(
    const float* __restrict__ qk, const float* __restrict__ v,
    float* __restrict__ out, int64_t ld, int channels,
    const int32_t* __restrict__ tiles, const int32_t* __restrict__ key_counts) {
    constexpr int QT = 4;                 // 16-query tiles per wave
    constexpr int KSTEPS = D / 4;         // k-steps of the QK^T product
    constexpr int MT = (D + 15) / 16;     // 16-row tiles of O^T
    const int lane = threadIdx.x;
    const int col = lane & 15;
    const int kk = lane >> 4;
    const int head = blockIdx.y;
    const Tile span = load_tile(tiles, blockIdx.x);
    const int q0 = span.first;
    const int queries = span.count;
    // keys: all positions of the segment, or its leading `key_counts[segment]`
    // when the rest is padding hidden by src_key_padding_mask (transformer.py:
    // 26-29); padded positions are still computed as queries
    const int length = key_counts != nullptr ? min(key_counts[span.segment], span.count)
                                             : span.count;
    // softmax in base 2: log2(e) rides on the query scale, exp is v_exp_f32
    const float scale = 1.44269504088896340736f / sqrtf(static_cast<float>(D));

    const float* q_rows = qk + static_cast<int64_t>(head * D) * ld + span.offset;
    const float* k_rows =
        qk + static_cast<int64_t>(channels + head * D) * ld + span.offset;
    const float* v_rows = v + static_cast<int64_t>(span.offset) * channels + head * D;

    float bq[QT][KSTEPS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int query = q0 + 16 * t + col;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            bq[t][s] = query < queries
                           ? q_rows[static_cast<int64_t>(4 * s + kk) * ld + query] * scale
                           : 0.f;
    }

    f32x4 o[QT][MT];
    float row_max[QT], row_sum[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        row_max[t] = -INFINITY;
        row_sum[t] = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) o[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // K fragment: A[i = key][k = d]; V fragment for k-step r: A[i = d][k] =
    // V[key0 + 4*kk + r][d].  The fragments of block i+1 are requested before
    // the MFMAs of block i (clamped addresses, masked values).
    float ak_next[KSTEPS], av_next[4][MT];
    auto request = [&](int key0) {
        const int key = min(key0 + col, length - 1);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            ak_next[s] = k_rows[static_cast<int64_t>(4 * s + kk) * ld + key];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int vkey = min(key0 + 4 * kk + r, length - 1);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                av_next[r][m] =
                    v_rows[static_cast<int64_t>(vkey) * channels + min(16 * m + col, D - 1)];
        }
    };
    if (length > 0) request(0);
    // One block of 16 keys.  The fragments need no masks: the loads are
    // clamped to valid rows, a key past the segment only feeds score rows that
    // are set to -inf below (probability exactly 0, times a finite V), and the
    // rows d >= D of a partial V^T tile only feed output rows that are never
    // stored.  Only the last, partial block masks its scores.
    auto block = [&](int key0, auto masked_tag) {
        constexpr bool kMasked = decltype(masked_tag)::value;
        float ak[KSTEPS];
        float av[4][MT];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) ak[s] = ak_next[s];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MT; ++m) av[r][m] = av_next[r][m];
        __builtin_amdgcn_sched_barrier(0);
        request(min(key0 + 16, length - 1));
        __builtin_amdgcn_sched_barrier(0);

        // Three phases over the four query tiles rather than tile by tile: the
        // ten QK^T MFMAs of a tile form a dependent chain, so the four chains
        // are issued interleaved; the softmax of all tiles then runs under the
        // tail of those MFMAs, and the PV products follow back to back.
        f32x4 s4[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) s4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int t = 0; t < QT; ++t)
                s4[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[s], bq[t][s], s4[t], 0, 0, 0);
        float alpha[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            // s4[t][r] = score(key0 + 4*kk + r, query col)
            float local = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kMasked && key0 + 4 * kk + r >= length) s4[t][r] = -INFINITY;
                local = fmaxf(local, s4[t][r]);
            }
            local = rows_max(local);
            const float new_max = fmaxf(row_max[t], local);
            alpha[t] = __builtin_amdgcn_exp2f(row_max[t] - new_max);
            float partial = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s4[t][r] = __builtin_amdgcn_exp2f(s4[t][r] - new_max);
                partial += s4[t][r];
            }
            partial = rows_sum(partial);
            row_sum[t] = row_sum[t] * alpha[t] + partial;
            row_max[t] = new_max;
        }
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) o[t][m] *= alpha[t];
        // key sub-step outermost: consecutive MFMAs then write twelve different
        // accumulators instead of the same one four times in a row
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    o[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        av[r][m], s4[t][r], o[t][m], 0, 0, 0);
    };
    const int full = length & ~15;
    for (int key0 = 0; key0 < full; key0 += 16) block(key0, std::false_type{});
    if (full < length) block(full, std::true_type{});

    // O^T[d = 16 m + 4 kk + r][query col]
    float* o_rows = out + static_cast<int64_t>(head * D) * ld + span.offset;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int query = q0 + 16 * t + col;
        if (query >= queries) continue;
        // no key at all: softmax over an empty set is NaN, as in torch
        const float inverse = length > 0 ? 1.f / row_sum[t] : NAN;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * m + 4 * kk + r;
                if (d < D)
                    o_rows[static_cast<int64_t>(d) * ld + query] = o[t][m][r] * inverse;
            }
    }
}

}  // namespace emph

using namespace emph;

extern "C" {

int emph_add_position(float* x, int64_t ldx, const float* table, int32_t channels,
                      int32_t max_positions, const int32_t* tiles,
                      int32_t n_tiles, int32_t tile_n, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(x && table && tiles, EMPH_EINVAL,
                 "emph_add_position: null pointer");
    EMPH_REQUIRE(tile_n > 0 && channels > 0, EMPH_EINVAL, "emph_add_position: bad shape");
    EMPH_LAUNCH(add_position_kernel, dim3(n_tiles), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, table, channels,
                       max_positions, tiles, tile_n);
    return check_launch("emph_add_position");
}

int emph_add_layernorm(const float* x, const float* r, float* y, int64_t ld,
                       int32_t channels, const float* gamma, const float* beta,
                       float eps, int64_t first_column, int64_t columns,
                       void* stream) {
    if (columns == 0) return EMPH_OK;
    EMPH_REQUIRE(x && r && y && gamma && beta, EMPH_EINVAL,
                 "emph_add_layernorm: null pointer");
    EMPH_REQUIRE(channels > 0 && channels <= 128, EMPH_ERANGE,
                 "emph_add_layernorm: channels %d not in 1..128", channels);
    const unsigned blocks = static_cast<unsigned>((columns + 63) / 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels <= 80)
        EMPH_LAUNCH(add_layernorm_kernel<80>, dim3(blocks), dim3(256), 0, s, x,
                           r, y, ld, channels, gamma, beta, eps, first_column, columns);
    else
        EMPH_LAUNCH(add_layernorm_kernel<128>, dim3(blocks), dim3(256), 0, s, x,
                           r, y, ld, channels, gamma, beta, eps, first_column, columns);
    return check_launch("emph_add_layernorm");
}

int emph_attention(const float* qk, const float* v, float* out, int64_t ld,
                   int32_t channels, int32_t heads, const int32_t* tiles,
                   int32_t n_tiles, const int32_t* key_counts, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(qk && v && out && tiles, EMPH_EINVAL,
                 "emph_attention: null pointer");
    EMPH_REQUIRE(heads > 0 && channels % heads == 0, EMPH_EINVAL,
                 "emph_attention: channels %d not divisible by heads %d", channels,
                 heads);
    const int d = channels / heads;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(n_tiles, heads);
    switch (d) {
        case 32:
            EMPH_LAUNCH(attention_kernel<32>, grid, dim3(64), 0, s, qk, v, out, ld,
                               channels, tiles, key_counts);
            break;
        case 40:
            EMPH_LAUNCH(attention_kernel<40>, grid, dim3(64), 0, s, qk, v, out, ld,
                               channels, tiles, key_counts);
            break;
        case 64:
            EMPH_LAUNCH(attention_kernel<64>, grid, dim3(64), 0, s, qk, v, out, ld,
                               channels, tiles, key_counts);
            break;
        default:
            set_error("emph_attention: head dimension %d not in {32, 40, 64}", d);
            return EMPH_ERANGE;
    }
    return check_launch("emph_attention");
}

}  // extern "C"
