// Winograd F(4,3) form of Conv1d(kernel_size 3, 'same'): four adjacent outputs
// share six inputs d0..d5 = x[4q-1 .. 4q+4], so a layer is SIX [c_out x c_in]
// GEMMs over QUADS of positions instead of three over positions: half of the
// direct form's MFMA work (F(2,3), conv.hip, does two thirds).
//
//   v = B^T d   (13 VALU ops per fragment and iteration, see transform())
//   m_j = U_j v_j,  U = G w formed in float64 on the host
//   y0 = m0 + m1 + m2 + m3 + m4      y1 = (m1 - m2) + 2 (m3 - m4)
//   y2 = (m1 + m2) + 4 (m3 + m4)     y3 = (m1 - m2) + 8 (m3 - m4) + m5
//
// In fp32 the encoder output of the trained model differs from the direct form
// by 2.5e-7 (scale 0.64) and the input layer by 9.5e-6 (scale 27): the same
// level as F(2,3) (numpy emulation, then the GPU parity tests).
//
// A workgroup is eight waves around one 6 x (c_in/4) x (c_out/16) x 256-byte
// pack in LDS (153.6 KB for 80 x 80: identity / ReLU epilogues only, there is no
// room for an LDS patch).  Four tiles of 64 positions per workgroup; waves w and
// w + 4 - the two waves of a SIMD - share a tile and split its m-tiles 3 + 2, so
// every SIMD carries c_out/16 tiles.  A lane holds one quad: its six inputs are
// one 16-byte and one 8-byte load, its four outputs one 16-byte store (256-byte
// runs per row).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct Run6 {
    f32x4u a;
    f32x2u b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4u*>(p);
        b = *reinterpret_cast<const f32x2u*>(p + 4);
    }
    __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
};

// M_TILES = ceil(c_out / 16) (compile time: the LDS offsets of the 6 M_TILES
// fragments of an iteration are immediates)
// POSITION: the layer is followed by PositionalEncoding (transformer.py:45-52) -
// `position[c][t]`, channel-major with `max_positions` columns, is added to the
// output at position t of its segment (after the activation).
//
// HALF: the workgroup is ONE of the two halves - four waves (one per tile) that
// own the upper (`half` = 0: ceil(m_tiles / 2)) or lower m-tiles, around THEIR
// rows of the pack only (`pack` = that half's slice of emph_conv_winograd4_
// split_pack: 92 KB and 61 KB for 80 x 80).  The two halves of a layer are
// separate launches on two streams: nothing couples them (they read the same
// input and write different output rows), so a compute unit holds the two halves
// of DIFFERENT layers - of the two batches in flight - side by side, or one half
// beside front-end workgroups, and one's weight transfer and store burst run
// under the other's MFMAs instead of idling the matrix pipe.
//
// WORD_SUMS: the layer is the LAST frame-rate layer in front of the per-word sum
// (emphases/core.py:438-454, DOWNSAMPLE_METHOD 'sum' / 'average').  Its output is
// never written: a tile forms the running sum of its 64 positions per channel
// (in-lane over the quad, then a 16-lane row scan on DPP) and stores it only at
// the positions `slot_map` marks - the last frame of a word (or of a word's part
// in this tile) and the frame in front of a word's first - as one 16-byte run of
// four channels into `y` = sums[slot][ldy].  A word's sum is then a handful of
// signed terms (emph_word_sums): 1.2 MB leave the chip instead of 20.5 MB, and
// the 320 B / frame pass of emph_segment_reduce is gone.
template <int M_TILES, bool POSITION, bool HALF, bool WORD_SUMS = false>
__global__ __launch_bounds__(HALF ? 256 : 512) void conv1d_winograd4_kernel(
    const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
    const float* __restrict__ pack, const float* __restrict__ bias, int c_in, int c_out,
    int act, const int32_t* __restrict__ tiles, int n_tiles, int bias_offset,
    const float* __restrict__ position, int max_positions, int half,
    const int32_t* __restrict__ slot_map = nullptr) {
    extern __shared__ __align__(16) float weights[];   // [groups][6][m_tiles][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4;
    const int col = lane & 15;
    const int iterations = (c_in + 3) >> 2;
    constexpr int m_tiles = M_TILES;
    constexpr int split = (m_tiles + 1) >> 1;
    constexpr int MT = split;
    constexpr int kThreads = HALF ? 256 : 512;
    const int part = HALF ? half : wave >> 2;
    const int m_begin = part ? split : 0;
    const int m_count = part ? m_tiles - split : split;     // wave-uniform, <= MT
    float* bias_lds = weights + bias_offset;

    Tile span;
    int t0 = 0;
    bool active = false;
    bool inside[6];
    bool edge = false;              // wave-uniform: the tile touches its segment's ends
    const float* lane_rows = x;     // row kk of the lane's quad
    int4 slots = {-1, -1, -1, -1};  // WORD_SUMS: where this lane's positions report
    auto open_tile = [&](int group) {
        const int tile = group * 4 + (wave & 3);
        active = tile < n_tiles && m_count > 0;
        span = load_tile(tiles, tile < n_tiles ? tile : 0);
        t0 = span.first;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int t = t0 + 4 * col - 1 + i;
            inside[i] = t >= 0 && t < span.count;
        }
        edge = t0 == 0 || t0 + 65 > span.count;
        lane_rows = x + span.offset + t0 + 4 * col - 1 + static_cast<int64_t>(kk) * ldx;
        // (requested a whole K loop ahead of its use; columns past the segment's
        // end are padding of the same buffer, never reported)
        if (WORD_SUMS && active)
            slots = *reinterpret_cast<const int4*>(slot_map + span.offset + t0 + 4 * col);
    };
    // c_in is a multiple of 4 (checked by the launcher), so rows 4 it + kk always
    // exist: the address is a wave-uniform offset on a per-tile lane pointer.
    auto load_b = [&](Run6& b, int iteration) {
        const int it = min(iteration, iterations - 1);
        b.load(lane_rows + static_cast<int64_t>(4 * it) * ldx);
    };

    const int groups_of_tiles = (n_tiles + 3) / 4;
    Run6 b0, b1;
    open_tile(blockIdx.x);
    {
        const int quads = iterations * 6 * (HALF ? m_count : m_tiles) * 16;
        // (480 quads per iteration: an odd iteration count ends in half a wave)
        for (int base = wave * 64; base < quads; base += kThreads)
            if (base + lane < quads)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(pack + 4 * (base + lane)),
                    (__attribute__((address_space(3))) void*)(weights + 4 * base), 16, 0, 0);
        if (active) load_b(b0, 0);
        // (HALF: the half's own channels, bias_lds[i] = bias of channel 16 m_begin + i)
        const int bias_first = HALF ? 16 * m_begin : 0;
        for (int index = threadIdx.x; index < (HALF ? m_count : m_tiles) * 16;
             index += kThreads)
            bias_lds[index] = (bias != nullptr && bias_first + index < c_out)
                                  ? bias[bias_first + index]
                                  : 0.f;
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): LDS-DMA landed
        __syncthreads();
    }

    for (int group = blockIdx.x; group < groups_of_tiles; group += gridDim.x) {
        if (group != static_cast<int>(blockIdx.x)) {
            __builtin_amdgcn_s_waitcnt(0x0F70);    // drain the previous tile's stores
            open_tile(group);
            if (active) load_b(b0, 0);
        }
        if (!active) continue;
        // one instantiation per number of m-tiles a wave can own (a wave-uniform
        // `if (m < m_count)` around the MFMAs makes hipcc shuffle accumulators)
        auto run = [&](auto count_tag) {
            constexpr int COUNT = decltype(count_tag)::value;
        f32x4 acc[6][COUNT];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int m = 0; m < COUNT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

        float a0[6][COUNT], av[6][COUNT], v[6];
        auto load_a = [&](int iteration) {
            // (HALF: the image holds this half's COUNT m-tiles only)
            constexpr int row = HALF ? COUNT : m_tiles;
            const float* fragment =
                weights + ((iteration * 6 * row + (HALF ? 0 : m_begin)) << 6) + lane;
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int m = 0; m < COUNT; ++m) a0[j][m] = fragment[(j * row + m) << 6];
        };
        auto step = [&](Run6& b, int iteration) {
            float d[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = b.get(i);
            if (edge) {                                    // wave-uniform
#pragma unroll
                for (int i = 0; i < 6; ++i) d[i] = inside[i] ? d[i] : 0.f;
            }
            // v = B^T d
            const float p = fmaf(-4.f, d[2], d[4]);
            const float q = fmaf(-4.f, d[1], d[3]);
            const float c = d[4] - d[2];
            const float e = 2.f * (d[3] - d[1]);
            v[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
            v[1] = p + q;
            v[2] = p - q;
            v[3] = c + e;
            v[4] = c - e;
            v[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int m = 0; m < COUNT; ++m) av[j][m] = a0[j][m];
            __builtin_amdgcn_sched_barrier(0);
            load_b(b, iteration + 2);
            load_a(min(iteration + 1, iterations - 1));
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int m = 0; m < COUNT; ++m)
                    acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        av[j][m], v[j], acc[j][m], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);   // VALU/SALU
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM read
#pragma unroll
            for (int k = 0; k < 6 * COUNT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        load_a(0);
        load_b(b1, 1);
        int iteration = 0;
#pragma unroll 1
        for (; iteration + 1 < iterations; iteration += 2) {
            step(b0, iteration);
            step(b1, iteration + 1);
        }
        if (iteration < iterations) step(b0, iteration);

        // ---- output transform, bias, ReLU, one 16-byte store per row and quad
        const bool relu = act == EMPH_ACT_RELU;
        const int t = t0 + 4 * col;
        if (WORD_SUMS) {
            // `slots`: where this lane's four positions report (-1: nobody needs
            // the running sum there)
#pragma unroll
            for (int m = 0; m < COUNT; ++m) {
                const int channel0 = 16 * (m_begin + m) + 4 * kk;
                const f32x4 add = *reinterpret_cast<const f32x4*>(bias_lds + channel0);
                f32x4 sum[4];       // [position of the quad][channel r]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m1 = acc[1][m][r], m2 = acc[2][m][r];
                    const float m3 = acc[3][m][r], m4 = acc[4][m][r];
                    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                    float o0 = acc[0][m][r] + s12 + s34 + add[r];
                    float o1 = fmaf(2.f, d34, d12) + add[r];
                    float o2 = fmaf(4.f, s34, s12) + add[r];
                    float o3 = fmaf(8.f, d34, d12) + acc[5][m][r] + add[r];
                    if (relu) {
                        o0 = o0 < 0.f ? 0.f : o0;
                        o1 = o1 < 0.f ? 0.f : o1;
                        o2 = o2 < 0.f ? 0.f : o2;
                        o3 = o3 < 0.f ? 0.f : o3;
                    }
                    // running sum over the tile: the quad in the lane, then the 16
                    // quads of the row (lanes of one kk) on DPP, in a fixed order
                    o1 += o0;
                    o2 += o1;
                    o3 += o2;
                    float scan = o3;
                    scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, scan), 0x111, 0xf, 0xf, true));  // row_shr:1
                    scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, scan), 0x112, 0xf, 0xf, true));  // row_shr:2
                    scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, scan), 0x114, 0xf, 0xf, true));  // row_shr:4
                    scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, scan), 0x118, 0xf, 0xf, true));  // row_shr:8
                    const float before = scan - o3;     // the quads to the left
                    sum[0][r] = o0 + before;
                    sum[1][r] = o1 + before;
                    sum[2][r] = o2 + before;
                    sum[3][r] = o3 + before;
                }
                if (channel0 >= c_out) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int slot = j == 0 ? slots.x : j == 1 ? slots.y : j == 2 ? slots.z : slots.w;
                    if (slot >= 0 && t + j < span.count)
                        *reinterpret_cast<f32x4*>(y + static_cast<int64_t>(slot) * ldy + channel0) =
                            sum[j];
                }
            }
            return;
        }
        const bool vector_ok = (ldy & 3) == 0 && (span.offset & 3) == 0 &&
                               (reinterpret_cast<uintptr_t>(y) & 15) == 0;
#pragma unroll
        for (int m = 0; m < COUNT; ++m) {
            const int channel0 = 16 * (m_begin + m) + 4 * kk;
            const f32x4 add = *reinterpret_cast<const f32x4*>(
                bias_lds + channel0 - (HALF ? 16 * m_begin : 0));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m1 = acc[1][m][r], m2 = acc[2][m][r];
                const float m3 = acc[3][m][r], m4 = acc[4][m][r];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                float4 out;
                out.x = acc[0][m][r] + s12 + s34 + add[r];
                out.y = fmaf(2.f, d34, d12) + add[r];
                out.z = fmaf(4.f, s34, s12) + add[r];
                out.w = fmaf(8.f, d34, d12) + acc[5][m][r] + add[r];
                if (relu) {
                    out.x = out.x < 0.f ? 0.f : out.x;
                    out.y = out.y < 0.f ? 0.f : out.y;
                    out.z = out.z < 0.f ? 0.f : out.z;
                    out.w = out.w < 0.f ? 0.f : out.w;
                }
                if (channel0 + r >= c_out || t >= span.count) continue;
                if (POSITION) {
                    const float* encoding =
                        position + static_cast<int64_t>(channel0 + r) * max_positions + t;
                    if (t + 3 < max_positions) {
                        const f32x4u e = *reinterpret_cast<const f32x4u*>(encoding);
                        out.x += e[0], out.y += e[1], out.z += e[2], out.w += e[3];
                    } else {
                        if (t < max_positions) out.x += encoding[0];
                        if (t + 1 < max_positions) out.y += encoding[1];
                        if (t + 2 < max_positions) out.z += encoding[2];
                    }
                }
                float* target = y + static_cast<int64_t>(channel0 + r) * ldy + span.offset + t;
                if (vector_ok && t + 3 < span.count) {
                    *reinterpret_cast<float4*>(target) = out;
                } else {
                    target[0] = out.x;
                    if (t + 1 < span.count) target[1] = out.y;
                    if (t + 2 < span.count) target[2] = out.z;
                    if (t + 3 < span.count) target[3] = out.w;
                }
            }
        }
        };
        if (m_count == MT) run(std::integral_constant<int, MT>{});
        else run(std::integral_constant<int, (MT > 1 ? MT - 1 : 1)>{});
    }
}

}  // namespace emph

using namespace emph;

extern "C" {

int64_t emph_conv_winograd4_pack_size(int32_t c_out, int32_t c_in) {
    return static_cast<int64_t>((c_in + 3) / 4) * 6 * ((c_out + 15) / 16) * 64;
}

int64_t emph_conv_winograd4_lds_bytes(int32_t c_out, int32_t c_in) {
    return (emph_conv_winograd4_pack_size(c_out, c_in) + ((c_out + 15) / 16) * 16) *
           static_cast<int64_t>(sizeof(float));
}

// pack[group][j][m][lane] = U_j[16 m + (lane & 15)][4 group + (lane >> 4)],
// U_j = sum_k G[j][k] w[:, :, k]
int emph_conv_winograd4_pack(const float* host_weight, int32_t c_out, int32_t c_in,
                             float* host_pack) {
    EMPH_REQUIRE(host_weight && host_pack, EMPH_EINVAL,
                 "emph_conv_winograd4_pack: null pointer");
    EMPH_REQUIRE(c_out > 0 && c_in > 0, EMPH_EINVAL, "emph_conv_winograd4_pack: bad shape");
    static const double G[6][3] = {{1. / 4, 0., 0.},           {-1. / 6, -1. / 6, -1. / 6},
                                   {-1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6},
                                   {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    const int groups = (c_in + 3) / 4, m_tiles = (c_out + 15) / 16;
    for (int group = 0; group < groups; ++group)
        for (int j = 0; j < 6; ++j)
            for (int m = 0; m < m_tiles; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = 16 * m + (lane & 15);
                    const int ci = 4 * group + (lane >> 4);
                    double value = 0.;
                    if (co < c_out && ci < c_in) {
                        const float* w = host_weight + (static_cast<int64_t>(co) * c_in + ci) * 3;
                        value = G[j][0] * w[0] + G[j][1] * w[1] + G[j][2] * w[2];
                    }
                    host_pack[(((static_cast<int64_t>(group) * 6 + j) * m_tiles + m) << 6) +
                              lane] = static_cast<float>(value);
                }
    return EMPH_OK;
}

// The same pack cut into the two halves a split layer launches: half 0 = m-tiles
// 0 .. ceil(m_tiles / 2) - 1 as [group][j][m][lane], then half 1 the same way.
int emph_conv_winograd4_split_pack(const float* host_weight, int32_t c_out, int32_t c_in,
                                   float* host_pack) {
    EMPH_REQUIRE(host_weight && host_pack, EMPH_EINVAL,
                 "emph_conv_winograd4_split_pack: null pointer");
    const int64_t size = emph_conv_winograd4_pack_size(c_out, c_in);
    float* whole = static_cast<float*>(malloc(static_cast<size_t>(size) * sizeof(float)));
    EMPH_REQUIRE(whole != nullptr, EMPH_EINVAL, "emph_conv_winograd4_split_pack: out of memory");
    const int status = emph_conv_winograd4_pack(host_weight, c_out, c_in, whole);
    if (status == EMPH_OK) {
        const int groups = (c_in + 3) / 4, m_tiles = (c_out + 15) / 16;
        const int split = (m_tiles + 1) / 2;
        float* cursor = host_pack;
        for (int half = 0; half < 2; ++half) {
            const int first = half ? split : 0, count = half ? m_tiles - split : split;
            for (int group = 0; group < groups; ++group)
                for (int j = 0; j < 6; ++j)
                    for (int m = 0; m < count; ++m)
                        for (int lane = 0; lane < 64; ++lane)
                            *cursor++ = whole[(((static_cast<int64_t>(group) * 6 + j) * m_tiles +
                                                first + m) << 6) + lane];
        }
    }
    free(whole);
    return status;
}

static int launch_winograd4(const float* x, int64_t ldx, float* y, int64_t ldy,
                            const float* pack, const float* bias, int32_t c_in, int32_t c_out,
                            int32_t activation, const int32_t* tiles, int32_t n_tiles,
                            const float* position, int32_t max_positions, int32_t half,
                            void* stream, const int32_t* slot_map = nullptr) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(x && y && pack && tiles, EMPH_EINVAL, "emph_conv1d_winograd4: null pointer");
    EMPH_REQUIRE(activation == EMPH_ACT_NONE || activation == EMPH_ACT_RELU, EMPH_ERANGE,
                 "emph_conv1d_winograd4: activation %d (identity or ReLU only)", activation);
    EMPH_REQUIRE(c_in >= 4 && c_in % 4 == 0 && c_out >= 1 && c_out <= 96, EMPH_ERANGE,
                 "emph_conv1d_winograd4: channels %d -> %d (c_in a multiple of 4, c_out <= 96)",
                 c_in, c_out);
    EMPH_REQUIRE(position == nullptr || max_positions > 0, EMPH_EINVAL,
                 "emph_conv1d_winograd4_position: %d positions in the table", max_positions);
    const int m_tiles = (c_out + 15) / 16;
    EMPH_REQUIRE(half >= -1 && half <= 1 && (half < 0 || m_tiles >= 2), EMPH_ERANGE,
                 "emph_conv1d_winograd4: half %d of %d m-tiles", half, m_tiles);
    // (half < 0: the whole layer in one launch)
    const int split = (m_tiles + 1) / 2;
    const int own_tiles = half < 0 ? m_tiles : half ? m_tiles - split : split;
    const int64_t per_tile = static_cast<int64_t>((c_in + 3) / 4) * 6 * 64;   // floats
    const int bias_offset = static_cast<int>(per_tile * own_tiles);
    const size_t lds = static_cast<size_t>(bias_offset + own_tiles * 16) * sizeof(float);
    EMPH_REQUIRE(lds <= 160 * 1024, EMPH_ERANGE,
                 "emph_conv1d_winograd4: %zu bytes of LDS needed", lds);
    if (half == 1) pack += per_tile * split;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int groups = (n_tiles + 3) / 4;
    dim3 grid(groups < 256 || half >= 0 ? groups : 256);
#define EMPH_W4_LAUNCH(M_TILES, POSITION, HALF, ...)                                           \
    do {                                                                                       \
        auto kernel = conv1d_winograd4_kernel<M_TILES, POSITION, HALF, ##__VA_ARGS__>;         \
        static LdsReservation reserved;                                                        \
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,     \
                                     "emph_conv1d_winograd4"))                                 \
            return status;                                                                     \
        EMPH_LAUNCH(kernel, grid, dim3(HALF ? 256 : 512), lds, s, x, ldx, y, ldy, pack, bias,  \
                    c_in, c_out, activation, tiles, n_tiles, bias_offset, position,            \
                    max_positions, half, slot_map);                                            \
    } while (0)
#define EMPH_W4(M_TILES)                                                                       \
    do {                                                                                       \
        if (slot_map != nullptr) EMPH_W4_LAUNCH(M_TILES, false, false, true);                  \
        else if (half >= 0) {                                                                  \
            if (position != nullptr) EMPH_W4_LAUNCH(M_TILES, true, true);                      \
            else EMPH_W4_LAUNCH(M_TILES, false, true);                                         \
        } else if (position != nullptr) EMPH_W4_LAUNCH(M_TILES, true, false);                  \
        else EMPH_W4_LAUNCH(M_TILES, false, false);                                            \
    } while (0)
    switch (m_tiles) {
        case 1: EMPH_W4(1); break;
        case 2: EMPH_W4(2); break;
        case 3: EMPH_W4(3); break;
        case 4: EMPH_W4(4); break;
        case 5: EMPH_W4(5); break;
        case 6: EMPH_W4(6); break;
        default:
            set_error("emph_conv1d_winograd4: %d output channels (at most 96)", c_out);
            return EMPH_ERANGE;
    }
#undef EMPH_W4
#undef EMPH_W4_LAUNCH
    return check_launch("emph_conv1d_winograd4");
}

int emph_conv1d_winograd4(const float* x, int64_t ldx, float* y, int64_t ldy,
                          const float* pack, const float* bias, int32_t c_in, int32_t c_out,
                          int32_t activation, const int32_t* tiles, int32_t n_tiles,
                          void* stream) {
    return launch_winograd4(x, ldx, y, ldy, pack, bias, c_in, c_out, activation, tiles, n_tiles,
                            nullptr, 0, -1, stream);
}

int emph_conv1d_winograd4_word_sums(const float* x, int64_t ldx, float* sums, int64_t ld_sums,
                                    const float* pack, const float* bias, int32_t c_in,
                                    int32_t c_out, int32_t activation, const int32_t* tiles,
                                    int32_t n_tiles, const int32_t* slot_map, void* stream) {
    EMPH_REQUIRE(slot_map != nullptr && sums != nullptr, EMPH_EINVAL,
                 "emph_conv1d_winograd4_word_sums: null pointer");
    EMPH_REQUIRE(c_out % 4 == 0 && ld_sums >= c_out && ld_sums % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(sums) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(slot_map) & 15) == 0,
                 EMPH_EINVAL,
                 "emph_conv1d_winograd4_word_sums: %d channels in rows of %lld floats (both "
                 "multiples of 4, 16-byte aligned tables)",
                 c_out, static_cast<long long>(ld_sums));
    return launch_winograd4(x, ldx, sums, ld_sums, pack, bias, c_in, c_out, activation, tiles,
                            n_tiles, nullptr, 0, -1, stream, slot_map);
}

int emph_conv1d_winograd4_half(const float* x, int64_t ldx, float* y, int64_t ldy,
                               const float* split_pack, const float* bias, int32_t c_in,
                               int32_t c_out, int32_t activation, const int32_t* tiles,
                               int32_t n_tiles, const float* position, int32_t max_positions,
                               int32_t half, void* stream) {
    EMPH_REQUIRE(half == 0 || half == 1, EMPH_EINVAL, "emph_conv1d_winograd4_half: half %d",
                 half);
    return launch_winograd4(x, ldx, y, ldy, split_pack, bias, c_in, c_out, activation, tiles,
                            n_tiles, position, max_positions, half, stream);
}

int emph_conv1d_winograd4_position(const float* x, int64_t ldx, float* y, int64_t ldy,
                                   const float* pack, const float* bias, int32_t c_in,
                                   int32_t c_out, int32_t activation, const int32_t* tiles,
                                   int32_t n_tiles, const float* position,
                                   int32_t max_positions, void* stream) {
    EMPH_REQUIRE(position != nullptr, EMPH_EINVAL,
                 "emph_conv1d_winograd4_position: null position table");
    return launch_winograd4(x, ldx, y, ldy, pack, bias, c_in, c_out, activation, tiles, n_tiles,
                            position, max_positions, -1, stream);
}

}  // extern "C"
