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

// Online softmax with a LAZY reference (scores in log2 units).  The textbook
// form subtracts the running row maximum and, whenever it moves, rescales the
// output accumulators - 24 packed multiplies per query tile and key block, plus
// two cross-lane reductions: 280 vector instructions per 16 keys against 88
// MFMAs, and on this chip fp32 MFMAs and vector instructions of a SIMD do not
// overlap.  Softmax is invariant to the reference, so any r with
// max - 100 < r works in fp32 (exp2(s - r) <= 2^100; what falls below 2^-126
// relative to the row's total is below fp32 resolution of the result anyway).
// So r = the first block's maximum, and it moves (with the rescale) only when
// some score of some lane of the wave exceeds it by more than 2^64: a
// wave-uniform, rare branch.  The denominator stays a per-lane partial sum and
// is reduced over the four key rows once, at the end.
template <int QT, int MT>
__device__ __forceinline__ void soften(f32x4 (&s4)[QT], f32x4 (&o)[QT][MT],
                                       float (&reference)[QT], float (&partial)[QT],
                                       bool masked, int first_key, int length) {
    bool moved = false;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (masked) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (first_key + r >= length) s4[t][r] = -INFINITY;
        }
        const float top = fmaxf(fmaxf(s4[t][0], s4[t][1]), fmaxf(s4[t][2], s4[t][3]));
        moved |= top > reference[t] + 64.f;       // also true while reference = -inf
    }
    if (__builtin_amdgcn_ballot_w64(moved)) {     // wave-uniform
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float top = fmaxf(fmaxf(s4[t][0], s4[t][1]), fmaxf(s4[t][2], s4[t][3]));
            top = fmaxf(rows_max(top), reference[t]);
            // all -inf so far (a fully masked block): keep the sums at zero
            const float alpha =
                top == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(reference[t] - top);
            partial[t] *= alpha;
#pragma unroll
            for (int m = 0; m < MT; ++m) o[t][m] *= alpha;
            reference[t] = top;
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s4[t][r] = __builtin_amdgcn_exp2f(s4[t][r] - reference[t]);
            partial[t] += s4[t][r];
        }
    }
}

// The same with the scores arriving already shifted (the QK^T accumulators
// start at -reference) and, with SUMMED_BY_MFMA, no denominator bookkeeping
// (a ones row of V^T accumulates it).
template <int QT, int MT, bool SUMMED_BY_MFMA>
__device__ __forceinline__ void soften_shifted(f32x4 (&s4)[QT], f32x4 (&o)[QT][MT],
                                               float (&reference)[QT], float (&partial)[QT],
                                               bool masked, int first_key, int length) {
    bool moved = false;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (masked) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (first_key + r >= length) s4[t][r] = -INFINITY;
        }
        const float top = fmaxf(fmaxf(s4[t][0], s4[t][1]), fmaxf(s4[t][2], s4[t][3]));
        moved |= reference[t] == -INFINITY ? top > -INFINITY : top > 64.f;
    }
    if (__builtin_amdgcn_ballot_w64(moved)) {     // wave-uniform, rare
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float top = fmaxf(fmaxf(s4[t][0], s4[t][1]), fmaxf(s4[t][2], s4[t][3]));
            // shifted scores: the reference moves by max(top, 0) (or, from -inf,
            // to the unshifted maximum itself)
            top = rows_max(top);
            const bool first = reference[t] == -INFINITY;
            const float shift = first ? top : fmaxf(top, 0.f);
            if (shift == -INFINITY) continue;      // nothing but masked keys so far
            const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-shift);
            partial[t] *= alpha;
#pragma unroll
            for (int m = 0; m < MT; ++m) o[t][m] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) s4[t][r] -= shift;
            reference[t] = first ? shift : reference[t] + shift;
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s4[t][r] = __builtin_amdgcn_exp2f(s4[t][r]);
            if (!SUMMED_BY_MFMA) partial[t] += s4[t][r];
        }
    }
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
        // s4[t][r] = score(key0 + 4*kk + r, query col), in log2 units
        soften<QT, MT>(s4, o, row_max, row_sum, kMasked, key0 + 4 * kk, length);
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
        const float inverse = length > 0 ? 1.f / rows_sum(row_sum[t]) : NAN;
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


// The same attention for LONG segments: a workgroup is eight waves of 32 queries
// (126 VGPRs: four waves per SIMD; four waves of 64 queries at two per SIMD are
// 6 % slower) = 256 consecutive queries of one segment and head, and the key / value blocks every
// one of them needs travel L2 -> registers -> LDS ONCE per workgroup (the
// one-wave kernel above reads them from L2 once per wave: 1.3 GB per layer on
// 64 x 1000 frames).  Stages of 64 keys, double buffered: while the MFMAs run
// on stage i, the 80 bytes per thread of stage i + 1 are in flight to registers;
// they are written to the other buffer behind ONE workgroup barrier per stage.
// LDS images (strides chosen so that the fragment reads of a 32-lane group cover
// 32 distinct banks):
//     K[d][64 keys]   row stride 80 floats  (A fragment: K[4 s + kk][key0 + col])
//     V[key][D]       row stride D + 4      (A fragment: V[key0 + 4 kk + r][16 m + col])
//
// WAVES = 8: 256 queries per workgroup, two workgroups per CU; WAVES = 16: 512
// queries, one workgroup per CU - the same sixteen waves per CU, but a segment's
// keys and values are staged once per 512 queries instead of once per 256.
constexpr int kQueryTiles = 2;     // 16-query tiles per wave: waves of 32 queries
template <int D, int WAVES>
__global__ __launch_bounds__(64 * WAVES)
__attribute__((amdgpu_waves_per_eu(8 / kQueryTiles, 8 / kQueryTiles)))
void attention_group_kernel(
    const float* __restrict__ qk, const float* __restrict__ v, float* __restrict__ out,
    int64_t ld, int channels, const int32_t* __restrict__ tiles,
    const int32_t* __restrict__ key_counts) {
    constexpr int QT = kQueryTiles;       // 16-query tiles per wave
    constexpr int THREADS = 64 * WAVES;
    constexpr int KSTEPS = D / 4;
    constexpr int MT = (D + 15) / 16;
    constexpr int STAGE = 64;             // keys per stage
    constexpr int KROW = 80;              // floats per K row (64 keys + pad)
    constexpr int VROW = D + 4;           // floats per V row
    constexpr int KFLOATS = D * KROW, VFLOATS = STAGE * VROW;
    constexpr int PIECES = (D * STAGE / 4 + THREADS - 1) / THREADS;     // 16-byte pieces per thread
    __shared__ __align__(16) float stage_k[2][KFLOATS];
    __shared__ __align__(16) float stage_v[2][VFLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15;
    const int kk = lane >> 4;
    // XCD-aware order: the hardware deals workgroups round-robin over the eight
    // XCDs (linear id L goes to XCD L % 8), each with an L2 of its own, and the
    // four tiles of a segment stream the same keys and values.  Walking the
    // (head, tile) space so that XCD i takes the i-th EIGHTH of it - consecutive
    // tiles in consecutive slots of one XCD - lets the second to fourth reader of
    // a segment's keys and values hit in that XCD's L2 (in dispatch order, tiles
    // t .. t + 3 of a segment sat on four different XCDs: four HBM fetches).
    const int linear = blockIdx.x + gridDim.x * blockIdx.y;
    const int total = gridDim.x * gridDim.y;
    const int per_xcd = total >> 3;
    const int logical = linear < 8 * per_xcd ? (linear & 7) * per_xcd + (linear >> 3) : linear;
    const int head = logical / static_cast<int>(gridDim.x);
    const Tile span = load_tile(tiles, logical - head * static_cast<int>(gridDim.x));
    const int q0 = span.first + 16 * QT * wave;
    const int queries = span.count;
    const int length = key_counts != nullptr ? min(key_counts[span.segment], span.count)
                                             : span.count;
    const bool working = q0 < queries;            // wave-uniform: this wave has queries
    const float scale = 1.44269504088896340736f / sqrtf(static_cast<float>(D));

    const float* q_rows = qk + static_cast<int64_t>(head * D) * ld + span.offset;
    const float* k_rows = qk + static_cast<int64_t>(channels + head * D) * ld + span.offset;
    const float* v_rows = v + static_cast<int64_t>(span.offset) * channels + head * D;

    // ---- staging: thread -> 16-byte pieces of the K and V images
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    f32x4 hold_k[PIECES], hold_v[PIECES];
    auto fetch = [&](int key0) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int piece = tid + THREADS * i;
            if (piece < D * STAGE / 4) {
                // K: piece = (d, 4 keys); clamped keys (masked or never used)
                const int d = piece / (STAGE / 4), quad = piece - d * (STAGE / 4);
                const int key = min(key0 + 4 * quad, max(length - 4, 0));
                const float* source = k_rows + static_cast<int64_t>(d) * ld + key;
                f32x4 value = *reinterpret_cast<const f32x4u*>(source);
                if (key0 + 4 * quad + 3 >= length) {
                    // a quad that straddles the end: element-wise, clamped
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        value[e] = k_rows[static_cast<int64_t>(d) * ld +
                                          min(key0 + 4 * quad + e, max(length - 1, 0))];
                }
                hold_k[i] = value;
                // V: piece = (key, 4 channels)
                const int key_v = piece / (D / 4), part = piece - key_v * (D / 4);
                const int row = min(key0 + key_v, max(length - 1, 0));
                hold_v[i] = *reinterpret_cast<const f32x4u*>(
                    v_rows + static_cast<int64_t>(row) * channels + 4 * part);
            }
        }
    };
    auto deposit = [&](int buffer) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int piece = tid + THREADS * i;
            if (piece < D * STAGE / 4) {
                const int d = piece / (STAGE / 4), quad = piece - d * (STAGE / 4);
                *reinterpret_cast<f32x4*>(&stage_k[buffer][d * KROW + 4 * quad]) = hold_k[i];
                const int key_v = piece / (D / 4), part = piece - key_v * (D / 4);
                *reinterpret_cast<f32x4*>(&stage_v[buffer][key_v * VROW + 4 * part]) =
                    hold_v[i];
            }
        }
    };

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

    // Rows D .. of the last V^T tile are spare when D is not a multiple of 16:
    // row D is fed ONES (column D of the staged V image), so its output row
    // accumulates sum(P) - the softmax denominator - inside the PV products and
    // no vector instruction adds probabilities up.
    constexpr bool kOnes = D % 16 != 0;
    if (kOnes) {
        for (int index = tid; index < 2 * STAGE; index += THREADS)
            stage_v[index / STAGE][(index % STAGE) * VROW + D] = 1.f;
    }

    // fragments of one block of 16 keys out of the staged images (no masks:
    // clamped keys only feed score rows that are set to -inf)
    float ak[1][KSTEPS], av[1][4][MT];
    auto fragments = [&](int set, const float* image_k, const float* image_v, int local) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            ak[set][s] = image_k[(4 * s + kk) * KROW + local + col];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                av[set][r][m] = image_v[(local + 4 * kk + r) * VROW +
                                        min(16 * m + col, kOnes ? D : D - 1)];
    };
    // the QK^T accumulators start at -reference: the scores land shifted
    auto block = [&](int set, int key0, bool masked) {
        f32x4 s4[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
            s4[t] = f32x4{-row_max[t], -row_max[t], -row_max[t], -row_max[t]};
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int t = 0; t < QT; ++t)
                s4[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[set][s], bq[t][s], s4[t], 0, 0,
                                                             0);
        if (masked) {                     // wave-uniform: a segment's last block only
            asm volatile("" ::: "memory");    // (a real branch, not sixteen selects)
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key0 + 4 * kk + r >= length) s4[t][r] = -INFINITY;
        }
        // s4 = score - reference.  Lazy reference: it moves (and the
        // accumulators are rescaled) only when a score of some lane of the wave
        // is more than 2^64 above it - one max3 chain over the sixteen scores
        float top = fmaxf(s4[0][0], s4[0][1]);
#pragma unroll
        for (int t = 0; t < QT; ++t)
            top = t == 0 ? __builtin_fmaxf(__builtin_fmaxf(top, s4[0][2]), s4[0][3])
                         : __builtin_fmaxf(
                               __builtin_fmaxf(__builtin_fmaxf(top, s4[t][0]), s4[t][1]),
                               __builtin_fmaxf(s4[t][2], s4[t][3]));
        if (__builtin_amdgcn_ballot_w64(top > 64.f)) {          // wave-uniform, rare
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float tile_top =
                    fmaxf(fmaxf(s4[t][0], s4[t][1]), fmaxf(s4[t][2], s4[t][3]));
                const float shift = fmaxf(rows_max(tile_top), 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-shift);
                row_sum[t] *= alpha;
#pragma unroll
                for (int m = 0; m < MT; ++m) o[t][m] *= alpha;
#pragma unroll
                for (int r = 0; r < 4; ++r) s4[t][r] -= shift;
                row_max[t] += shift;
            }
        }
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s4[t][r] = __builtin_amdgcn_exp2f(s4[t][r]);
                if (!kOnes) row_sum[t] += s4[t][r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    o[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][r][m], s4[t][r],
                                                                   o[t][m], 0, 0, 0);
    };

    const int stages = (length + STAGE - 1) / STAGE;
    if (stages > 0) {
        fetch(0);
        deposit(0);
    }
    __syncthreads();
    if (working && stages > 0) {
        // the reference starts at the maximum of the first 16 keys' scores
        fragments(0, stage_k[0], stage_v[0], 0);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
                s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[0][s], bq[t][s], s4, 0, 0, 0);
            float top = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * kk + r < length) top = fmaxf(top, s4[r]);
            row_max[t] = rows_max(top);               // finite: key 0 exists
        }
    }
    for (int stage = 0; stage < stages; ++stage) {
        const int buffer = stage & 1;
        const bool more = stage + 1 < stages;
        if (more) fetch((stage + 1) * STAGE);          // in flight during the MFMAs
        if (working) {
            const int key_base = stage * STAGE;
            const int keys = min(STAGE, length - key_base);
            // only a segment's last block can hold keys past its end
#pragma unroll 1
            for (int local = 0; local < keys; local += 16) {
                fragments(0, stage_k[buffer], stage_v[buffer], local);
                block(0, key_base + local, local + 16 > keys);
            }
        }
        if (more) deposit(buffer ^ 1);                 // last read two barriers ago
        __syncthreads();
    }

    if (!working) return;
    float* o_rows = out + static_cast<int64_t>(head * D) * ld + span.offset;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int query = q0 + 16 * t + col;
        // the denominator: output row D (tile MT - 1, row D % 16, held by the
        // lanes with kk = (D % 16) / 4 in register (D % 16) % 4), or the
        // per-lane partial sums reduced over the four key rows
        float total;
        if (kOnes) {
            const float mine = o[t][MT - 1][(D % 16) % 4];
            total = __shfl(mine, 16 * ((D % 16) / 4) + col);
        } else {
            total = rows_sum(row_sum[t]);
        }
        if (query >= queries) continue;
        const float inverse = length > 0 ? 1.f / total : NAN;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * m + 4 * kk + r;
                if (d < D) o_rows[static_cast<int64_t>(d) * ld + query] = o[t][m][r] * inverse;
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
                   int32_t n_tiles, int32_t tile_n, const int32_t* key_counts,
                   void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(tile_n == 64 || tile_n == 256 || tile_n == 512, EMPH_EINVAL,
                 "emph_attention: tile_n %d (64: one wave per tile; 256 / 512: a workgroup "
                 "of 8 / 16 waves per tile with keys and values staged in LDS)", tile_n);
    EMPH_REQUIRE(qk && v && out && tiles, EMPH_EINVAL,
                 "emph_attention: null pointer");
    EMPH_REQUIRE(heads > 0 && channels % heads == 0, EMPH_EINVAL,
                 "emph_attention: channels %d not divisible by heads %d", channels,
                 heads);
    const int d = channels / heads;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(n_tiles, heads);
    if (tile_n >= 256) {
#define EMPH_GROUP(D)                                                                     \
    do {                                                                                  \
        if (tile_n == 256)                                                                \
            EMPH_LAUNCH((attention_group_kernel<D, 8>), grid, dim3(512), 0, s, qk, v, out, \
                        ld, channels, tiles, key_counts);                                 \
        else                                                                              \
            EMPH_LAUNCH((attention_group_kernel<D, 16>), grid, dim3(1024), 0, s, qk, v,   \
                        out, ld, channels, tiles, key_counts);                            \
    } while (0)
        switch (d) {
            case 32: EMPH_GROUP(32); break;
            case 40: EMPH_GROUP(40); break;
            case 64: EMPH_GROUP(64); break;
            default:
                set_error("emph_attention: head dimension %d not in {32, 40, 64}", d);
                return EMPH_ERANGE;
        }
#undef EMPH_GROUP
        return check_launch("emph_attention");
    }
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
