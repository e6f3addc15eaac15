// The word-rate Transformer decoder in ONE launch: positional encoding and all
// `layers` post-LN encoder layers (emphases/model/layers/transformer.py:13-52,
// used as the word decoder by emphases/model/core.py:26-30,105-107) for
// segments of at most 64 words - every utterance and chunk of the path has a
// few dozen words, so a segment is ONE workgroup and its residual stream never
// leaves the registers between layers.
//
// As separate launches a layer is emph_qkv_projection + emph_attention +
// emph_transformer_block: three launches of 13-15 us for 1 882 words on sixteen
// workgroups each (launch ramp, three weight packs into LDS, a few microseconds
// of work): 18 launches, 256 us per batch.  Here a layer is about 12 us of MFMAs
// per wave and the weights stream through LDS behind them.
//
// Workgroup = four MFMA waves (wave w owns words 16 w .. 16 w + 15 of the
// segment, all channels) + one loader wave (hipcc orders every LDS read of a
// wave behind that wave's own LDS-DMAs, so the wave that requests weights must
// not be one that reads them; decoder.hip).  Per layer:
//   1. Q, K, V = W x + b from the residual stream IN REGISTERS: the MFMA result
//      layout (channel 16 m + 4 (lane >> 4) + r of word lane & 15) is a valid B
//      fragment of a GEMM whose k-step 4 m + r multiplies exactly those
//      channels (emph_linear_chain_pack, natural = 0).  Q (pre-scaled), K go to
//      LDS channel-major, V word-major.
//   2. attention per head over the <= 64 keys of the segment: all scores of a
//      query tile first (<= 4 blocks of 16 keys), exact two-pass softmax, PV.
//   3. out_proj straight from the PV accumulators (its k-steps are packed in
//      "attention order": head, V^T tile, register), residual, LayerNorm,
//      linear1, ReLU, linear2, residual, LayerNorm - as in block.hip.
// The weights of a layer are two groups that share one LDS region:
//   A = [W_q | W_k | W_v | b_q b_k b_v]   B = [W_o | W_1 | W_2 | 7 vectors]
// B is requested as soon as every wave is done with A's fragments and lands
// under the attention; A of the next layer is requested when B is done with.
#include <math.h>

#include <type_traits>

#include "common.h"

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace wt {

__device__ __forceinline__ float rows_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false,
                                              false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

constexpr int kWords = 64;        // words per workgroup (= one segment)
constexpr int kQStride = 68;      // floats per channel row of Q / K in LDS (4 mod 8)

// floats of one layer's pack, C = 16 MB channels, H heads of D = C / H
__host__ __device__ constexpr int pack_plain(int mb) { return 4 * mb * mb * 64; }
__host__ __device__ constexpr int attention_steps(int mb, int heads) {
    return heads * ((16 * mb / heads + 15) / 16) * 4;
}
__host__ __device__ constexpr int group_a(int mb) { return 3 * pack_plain(mb) + 3 * 16 * mb; }
__host__ __device__ constexpr int group_b(int mb, int heads) {
    return attention_steps(mb, heads) * mb * 64 + 2 * pack_plain(mb) + 7 * 16 * mb;
}
__host__ __device__ constexpr int region(int mb, int heads) {
    return group_a(mb) > group_b(mb, heads) ? group_a(mb) : group_b(mb, heads);
}

}  // namespace wt

// grid = word-axis tiles of 64 (one per segment); block = 320
template <int MB, int HEADS>
__global__ __launch_bounds__(320) void word_transformer_kernel(
    float* __restrict__ x, int64_t ld, const float* __restrict__ position, int max_positions,
    const float* __restrict__ packs, int layers, float eps,
    const int32_t* __restrict__ tiles) {
    constexpr int C = 16 * MB;
    constexpr int D = C / HEADS;
    constexpr int MT = (D + 15) / 16;             // V^T tiles per head
    constexpr int KSTEPS = D / 4;
    constexpr int STEPS = 4 * MB;
    constexpr int PACK = wt::pack_plain(MB);
    constexpr int OSTEPS = wt::attention_steps(MB, HEADS);
    constexpr int GA = wt::group_a(MB), GB = wt::group_b(MB, HEADS);
    constexpr int REGION = (wt::region(MB, HEADS) + 3) & ~3;
    constexpr int VS = C + 4;                     // floats per word row of V
    extern __shared__ __align__(16) float lds[];
    float* weights = lds;                                     // [REGION]
    float* qs = weights + REGION;                             // [C][kQStride]
    float* ks = qs + C * wt::kQStride;                        // [C][kQStride]
    float* vs = ks + C * wt::kQStride;                        // [64][VS]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4;
    const int col = lane & 15;
    const bool loader = wave == 4;
    const Tile span = load_tile(tiles, blockIdx.x);
    if (span.first != 0) return;                  // (segments of <= 64 words: one tile)
    const int count = span.count;
    const int word = 16 * wave + col;             // the lane's word (compute waves)
    const bool live = !loader && word < count;
    const int64_t column = span.offset + min(word, count - 1);

    auto request = [&](const float* source, int floats) {     // loader wave only
        for (int base = 0; base < floats / 4; base += 64)
            if (base + lane < floats / 4)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(source + 4 * (base + lane)),
                    (__attribute__((address_space(3))) void*)(weights + 4 * base), 16, 0, 0);
    };
    auto landed = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };   // vmcnt(0)
    constexpr int LAYER = ((GA + 3) & ~3) + ((GB + 3) & ~3);
    if (loader) request(packs, GA);

    // residual stream in the accumulator layout, + positional encoding
    f32x4 xr[MB];
    if (!loader) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * m + 4 * kk + r;
                const int t = min(word, min(count, max_positions) - 1);
                xr[m][r] = x[static_cast<int64_t>(c) * ld + column] +
                           position[static_cast<int64_t>(t) * C + c];
            }
    }

    // acc[m] += sum_s A[s][m] * B(s), A from an LDS pack
    auto gemm = [&](const float* pack, auto steps_tag, f32x4 (&acc)[MB], auto fragment) {
        constexpr int steps = decltype(steps_tag)::value;
        float a[2][MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) a[0][m] = pack[(m << 6) + lane];
#pragma unroll
        for (int s = 0; s < steps; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < steps) {
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    a[(s + 1) & 1][m] = pack[(((s + 1) * MB + m) << 6) + lane];
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][m], fragment(s), acc[m],
                                                              0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto with_bias = [&](f32x4 (&v)[MB], const float* bias) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
            v[m] = *reinterpret_cast<const f32x4*>(bias + 16 * m + 4 * kk);
    };
    auto add_bias = [&](f32x4 (&v)[MB], const float* bias) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
            v[m] += *reinterpret_cast<const f32x4*>(bias + 16 * m + 4 * kk);
    };
    auto layernorm = [&](f32x4 (&v)[MB], const float* gamma, const float* beta) {
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += v[m][r];
        const float mean = wt::rows_sum(sum) / static_cast<float>(C);
        float square = 0.f;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[m][r] -= mean;
                square = fmaf(v[m][r], v[m][r], square);
            }
        const float rstd = 1.f / sqrtf(wt::rows_sum(square) / static_cast<float>(C) + eps);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const f32x4 scale = *reinterpret_cast<const f32x4*>(gamma + 16 * m + 4 * kk);
            const f32x4 shift = *reinterpret_cast<const f32x4*>(beta + 16 * m + 4 * kk);
            v[m] = v[m] * rstd * scale + shift;
        }
    };

    const std::integral_constant<int, STEPS> plain{};
    const float scale = 1.44269504088896340736f / sqrtf(static_cast<float>(D));
    const int key_blocks = (count + 15) >> 4;
    for (int layer = 0; layer < layers; ++layer) {
        const float* group = packs + static_cast<int64_t>(layer) * LAYER;
        if (loader) landed();
        __syncthreads();                          // (a) group A is in LDS
        if (!loader) {
            // ---- 1. Q, K, V
            const float* bias = weights + 3 * PACK;
            f32x4 acc[MB];
            auto chain = [&](int s) { return xr[s >> 2][s & 3]; };
            with_bias(acc, bias);
            gemm(weights, plain, acc, chain);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    qs[(16 * m + 4 * kk + r) * wt::kQStride + word] = acc[m][r] * scale;
            with_bias(acc, bias + C);
            gemm(weights + PACK, plain, acc, chain);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ks[(16 * m + 4 * kk + r) * wt::kQStride + word] = acc[m][r];
            with_bias(acc, bias + 2 * C);
            gemm(weights + 2 * PACK, plain, acc, chain);
#pragma unroll
            for (int m = 0; m < MB; ++m)
                *reinterpret_cast<f32x4*>(vs + word * VS + 16 * m + 4 * kk) = acc[m];
        }
        __syncthreads();                          // (b) A's fragments are read; Q K V written
        if (loader) request(group + ((GA + 3) & ~3), GB);
        f32x4 o[HEADS][MT];
        if (!loader) {
            // ---- 2. attention: scores of all keys of the segment, then softmax, then PV
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                float bq[KSTEPS];
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s)
                    bq[s] = qs[(D * h + 4 * s + kk) * wt::kQStride + word];
                f32x4 s4[4];
                float top = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    s4[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (kb < key_blocks) {        // wave-uniform
#pragma unroll
                        for (int s = 0; s < KSTEPS; ++s)
                            s4[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                ks[(D * h + 4 * s + kk) * wt::kQStride + 16 * kb + col], bq[s],
                                s4[kb], 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // s4[kb][r] = score(key 16 kb + 4 kk + r, query col)
                        if (16 * kb + 4 * kk + r >= count) s4[kb][r] = -INFINITY;
                        top = fmaxf(top, s4[kb][r]);
                    }
                }
                top = wt::rows_max(top);          // finite: key 0 exists
                float sum = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s4[kb][r] = __builtin_amdgcn_exp2f(s4[kb][r] - top);
                        sum += s4[kb][r];
                    }
                const float inverse = 1.f / wt::rows_sum(sum);
#pragma unroll
                for (int m = 0; m < MT; ++m) o[h][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb >= key_blocks) break;  // wave-uniform
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            o[h][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                vs[(16 * kb + 4 * kk + r) * VS + D * h + min(16 * m + col, D - 1)],
                                s4[kb][r], o[h][m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) o[h][m] *= inverse;
            }
        }
        if (loader) landed();
        __syncthreads();                          // (c) group B is in LDS; K, V are read
        if (!loader) {
            // ---- 3. out_proj, residual, LayerNorm, feed-forward, residual, LayerNorm
            const float* pack_o = weights;
            const float* pack_1 = pack_o + OSTEPS * MB * 64;
            const float* pack_2 = pack_1 + PACK;
            const float* vec = pack_2 + PACK;     // b_o g1 be1 b_1 b_2 g2 be2
            add_bias(xr, vec);
            gemm(pack_o, std::integral_constant<int, OSTEPS>{}, xr, [&](int s) {
                return o[s / (4 * MT)][(s / 4) % MT][s & 3];
            });
            layernorm(xr, vec + C, vec + 2 * C);
            f32x4 hidden[MB];
            with_bias(hidden, vec + 3 * C);
            gemm(pack_1, plain, hidden, [&](int s) { return xr[s >> 2][s & 3]; });
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) hidden[m][r] = fmaxf(hidden[m][r], 0.f);
            add_bias(xr, vec + 4 * C);
            gemm(pack_2, plain, xr, [&](int s) { return hidden[s >> 2][s & 3]; });
            layernorm(xr, vec + 5 * C, vec + 6 * C);
        }
        __syncthreads();                          // (d) B's fragments are read
        if (loader && layer + 1 < layers) request(group + LAYER, GA);
    }
    if (live) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[static_cast<int64_t>(16 * m + 4 * kk + r) * ld + column] = xr[m][r];
    }
}

}  // namespace emph

using namespace emph;

namespace {

template <int MB, int HEADS>
int launch_word_transformer(float* x, int64_t ld, const float* position, int max_positions,
                            const float* packs, int layers, float eps, const int32_t* tiles,
                            int n_tiles, hipStream_t s) {
    constexpr int C = 16 * MB;
    const size_t lds = (((wt::region(MB, HEADS) + 3) & ~3) + 2 * C * wt::kQStride +
                        wt::kWords * (C + 4)) * sizeof(float);
    auto kernel = word_transformer_kernel<MB, HEADS>;
    static LdsReservation reserved;
    if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,
                                 "emph_word_transformer"))
        return status;
    EMPH_LAUNCH(kernel, dim3(n_tiles), dim3(320), lds, s, x, ld, position, max_positions,
                packs, layers, eps, tiles);
    return check_launch("emph_word_transformer");
}

}  // namespace

extern "C" {

int64_t emph_word_transformer_pack_size(int32_t channels, int32_t heads) {
    const int mb = channels / 16;
    return ((wt::group_a(mb) + 3) & ~3) + ((wt::group_b(mb, heads) + 3) & ~3);
}

int emph_word_transformer_pack(const float* in_proj_weight, const float* in_proj_bias,
                               const float* out_weight, const float* out_bias,
                               const float* linear1_weight, const float* linear1_bias,
                               const float* linear2_weight, const float* linear2_bias,
                               const float* norm1_weight, const float* norm1_bias,
                               const float* norm2_weight, const float* norm2_bias,
                               int32_t channels, int32_t heads, float* host_pack) {
    EMPH_REQUIRE(in_proj_weight && in_proj_bias && out_weight && out_bias && linear1_weight &&
                     linear1_bias && linear2_weight && linear2_bias && norm1_weight &&
                     norm1_bias && norm2_weight && norm2_bias && host_pack,
                 EMPH_EINVAL, "emph_word_transformer_pack: null pointer");
    EMPH_REQUIRE((channels == 64 || channels == 80) && heads == 2, EMPH_ERANGE,
                 "emph_word_transformer_pack: %d channels / %d heads (64 or 80 channels, 2 "
                 "heads)", channels, heads);
    const int mb = channels / 16, c = channels, d = channels / heads;
    const int mt = (d + 15) / 16;
    const int pack = wt::pack_plain(mb);
    for (int64_t i = 0; i < emph_word_transformer_pack_size(channels, heads); ++i)
        host_pack[i] = 0.f;
    // group A: W_q | W_k | W_v in chain order (k-step s = 4 m + r multiplies input
    // channels 16 m + 4 k + r), then the three biases
    float* a = host_pack;
    for (int part = 0; part < 3; ++part) {
        int status = emph_linear_chain_pack(in_proj_weight + static_cast<int64_t>(part) * c * c,
                                            channels, 0, a + part * pack);
        if (status) return status;
    }
    for (int i = 0; i < 3 * c; ++i) a[3 * pack + i] = in_proj_bias[i];
    // group B: W_o in attention order - k-step (h, m', r) multiplies input channels
    // d h + 16 m' + 4 k + r (zero where 16 m' + 4 k + r >= d) - then W_1, W_2
    // (chain order) and the seven vectors
    float* b = host_pack + ((wt::group_a(mb) + 3) & ~3);
    for (int h = 0; h < heads; ++h)
        for (int mp = 0; mp < mt; ++mp)
            for (int r = 0; r < 4; ++r) {
                const int step = (h * mt + mp) * 4 + r;
                for (int m = 0; m < mb; ++m)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int k = lane >> 4, co = 16 * m + (lane & 15);
                        const int within = 16 * mp + 4 * k + r;
                        b[((static_cast<int64_t>(step) * mb + m) << 6) + lane] =
                            within < d ? out_weight[static_cast<int64_t>(co) * c + d * h + within]
                                       : 0.f;
                    }
            }
    float* b1 = b + wt::attention_steps(mb, heads) * mb * 64;
    int status = emph_linear_chain_pack(linear1_weight, channels, 0, b1);
    if (status) return status;
    status = emph_linear_chain_pack(linear2_weight, channels, 0, b1 + pack);
    if (status) return status;
    float* vec = b1 + 2 * pack;
    const float* vectors[7] = {out_bias,     norm1_weight, norm1_bias, linear1_bias,
                               linear2_bias, norm2_weight, norm2_bias};
    for (int v = 0; v < 7; ++v)
        for (int i = 0; i < c; ++i) vec[v * c + i] = vectors[v][i];
    return EMPH_OK;
}

int emph_word_transformer(float* x, int64_t ld, const float* position, int32_t max_positions,
                          int32_t channels, int32_t heads, const float* packs, int32_t layers,
                          float eps, const int32_t* tiles, int32_t n_tiles, void* stream) {
    if (n_tiles == 0 || layers == 0) return EMPH_OK;
    EMPH_REQUIRE(x && position && packs && tiles, EMPH_EINVAL,
                 "emph_word_transformer: null pointer");
    EMPH_REQUIRE((channels == 64 || channels == 80) && heads == 2, EMPH_ERANGE,
                 "emph_word_transformer: %d channels / %d heads (64 or 80 channels, 2 heads)",
                 channels, heads);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels == 80)
        return launch_word_transformer<5, 2>(x, ld, position, max_positions, packs, layers, eps,
                                             tiles, n_tiles, s);
    return launch_word_transformer<4, 2>(x, ld, position, max_positions, packs, layers, eps,
                                         tiles, n_tiles, s);
}

}  // extern "C"
