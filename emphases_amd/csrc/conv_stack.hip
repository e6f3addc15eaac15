// Up to THREE consecutive Conv1d(80, 80, 3, 'same') + activation layers of the
// frame encoder (emphases/model/core.py:24-31,96-100 over
// model/layers/convolution.py:25-37) in ONE launch, Winograd F(4,3) on
// v_mfma_f32_16x16x4_f32 like conv_w4.hip - the same arithmetic per output, so the
// results are bit for bit those of the layer-by-layer kernel - but with the
// activations resident in LDS from layer to layer and the weights streamed
// through a two-slot LDS ring by loader waves.
//
// Why: a layer-by-layer launch spends 9.3 us of its 18.6 us in the matrix pipe; the
// rest is launch ramp, the 153.6 KB pack transfer that nothing can run under
// (the pack fills the LDS), and the store burst of 20.5 MB per layer
// (EXPERIMENTS.md, rounds 1-4 section 6).  Here a workgroup owns a SPAN of up to 252 consecutive
// positions of one segment for all the layers of the launch:
//
//   * 256 computed positions = 4 MFMA column tiles of 16 quads = the span plus a
//     halo of one quad on each side that continues inside the segment.  The
//     halo is recomputed, at most 2.4 % more matrix work, instead of exchanged: a
//     cross-workgroup hand-off per layer costs more than the layer's ramp
//     (MI355X_MICROARCH.md, inter-workgroup visibility).  Why three layers and
//     not four: the launch's first layer is exact everywhere (the columns beside
//     the computed region are loaded with it); in the SECOND the stale column
//     reaches only ONE output of the edge quad (in F(4,3) d0 enters v0, m0, y0
//     alone; d5 only y3), in the third every output of that quad - two of them
//     only through rounding, the mathematically cancelling terms - and in a
//     fourth, through d0 / d5 of the NEXT quad, the span's own first / last
//     position: 6.6e-7 off the layer-by-layer result (measured).  Three layers
//     are exact with one quad of halo; four would need two (240 own positions:
//     a 10 s utterance would no longer be four spans);
//   * activations [80][260] floats (83.2 KB) are updated in place: accumulators
//     live in registers, a barrier separates the last read of a layer from the
//     first write of its output.  Index i of a row is position c0 - 1 + i, so the
//     six inputs x[4q-1 .. 4q+4] of a quad are one aligned 16-byte and one 8-byte
//     LDS read; positions outside the segment hold zeros ('same' padding by
//     construction: no masks in the K loop);
//   * weights: the layer's pack is k-major, so it streams in chunks of four
//     k-steps (30.7 KB) through ring[2]; four loader waves request chunk g + 1 by
//     LDS-DMA while the eight MFMA waves consume chunk g (one wave's LDS-DMA
//     requests complete one after the other: four waves are what keeps a chunk
//     ahead).  One barrier per chunk, one more per layer.
//
// The last layer of the launch writes the span to global memory, or - WORD_SUMS,
// when it is the layer in front of the per-word sum (emphases/core.py:438-454) -
// only the running sums the words need, exactly like
// conv1d_winograd4_kernel<..., WORD_SUMS> (conv_w4.hip), over the span's own
// positions.
#include <type_traits>

#include "common.h"

// (tools/micro/stack_bench.hip defines STACK_STAMP for an in-kernel timeline)
#ifndef STACK_STAMP
#define STACK_STAMP(slot)
#endif

namespace emph {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kStackChannels = 80;
constexpr int kStackMTiles = 5;
constexpr int kStackSteps = kStackChannels / 4;           // k-steps per layer
constexpr int kStackChunkSteps = 4;
constexpr int kStackChunks = kStackSteps / kStackChunkSteps;
constexpr int kStackStepFloats = 6 * kStackMTiles * 64;   // one k-step of the pack
constexpr int kStackChunkFloats = kStackChunkSteps * kStackStepFloats;
constexpr int kStackWidth = 256;                          // computed positions
constexpr int kStackStride = 260;                         // floats per activation row
constexpr int kStackThreads = 768;                        // 8 MFMA waves + 4 loader waves
constexpr int kStackMaxLayers = 3;
constexpr int kSpanFields = 8;

__host__ __device__ constexpr int stack_lds_floats() {
    return kStackChannels * kStackStride + 2 * kStackChunkFloats +
           kStackMaxLayers * kStackChannels;
}

// spans: int32 [n][8] = {segment, first owned position, frame column of the
// segment, positions of the segment, owned positions, first computed position,
// 0, 0}
template <bool WORD_SUMS>
__global__ __launch_bounds__(kStackThreads) void conv1d_stack_kernel(
    const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
    const float* __restrict__ packs, const float* __restrict__ biases, int layers,
    int relu_mask, const int32_t* __restrict__ spans, const int32_t* __restrict__ slot_map) {
    extern __shared__ __align__(16) float lds[];
    float* act = lds;
    float* ring = act + kStackChannels * kStackStride;
    float* bias_lds = ring + 2 * kStackChunkFloats;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= 8;

    const int4 span_a = reinterpret_cast<const int4*>(spans)[2 * blockIdx.x];
    const int4 span_b = reinterpret_cast<const int4*>(spans)[2 * blockIdx.x + 1];
    const int owned_first = span_a.y;
    const int64_t column = span_a.z;          // frame column of the segment's position 0
    const int count = span_a.w;               // positions of the segment
    const int owned = span_b.x;
    const int c0 = span_b.y;                  // first computed position (a multiple of 4)

    constexpr int kPackFloats = kStackSteps * kStackStepFloats;
    const int total_chunks = layers * kStackChunks;
    // chunk g -> ring[g & 1]: 1920 16-byte quads, i.e. 30 wave requests, dealt over
    // waves first .. first + n - 1 (n = 4: 8 each, n = 8: 4 each)
    auto request = [&](int g, int first, int n) {
        const int layer = g / kStackChunks;
        const float* source = packs + static_cast<int64_t>(layer) * kPackFloats +
                              (g - layer * kStackChunks) * kStackChunkFloats;
        float* target = ring + (g & 1) * kStackChunkFloats;
        // (every wave the same number of requests, the waits below count them: the
        // last of some waves repeats the chunk's last 1 KB)
        constexpr int kQuads = kStackChunkFloats / 4;
        const int each = (kQuads / 64 + n - 1) / n;
        for (int k = 0; k < each; ++k) {
            const int from = min(((wave - first) + k * n) * 64, kQuads - 64);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(source + 4 * (from + lane)),
                (__attribute__((address_space(3))) void*)(target + 4 * from), 16, 0, 0);
        }
    };
    STACK_STAMP(0);

    // ---- the layer-0 input of the computed positions, zeros outside the segment.
    // One 16-byte run per (row, quad), always ONE load from a readable address
    // (clamped to the segment's last quad) and masked by position - so a thread's
    // loads are all requested before the first is used (a global load is 1-2 us
    // away on this chip) and their number is a constant.  All CUs pulling their
    // 82 KB + two weight chunks at once is a 4 us burst at HBM speed, and the K
    // loop needs only the first weight chunk and the 16 rows it multiplies to
    // start: the eight MFMA waves request exactly those, FIRST (requests are
    // served roughly in the order the CUs issue them); the loader waves request
    // chunk 1 and rows 16 .. 79 behind them and hand the rows over in the order
    // the chunks need them - rows 16 .. 31 under chunk 0, the rest under chunk 1.
    const int last_quad = (count - 1) & ~3;
    auto fetch = [&](int c, int q) {
        const int p = c0 + 4 * q;
#ifdef STACK_NO_ROWS               // (micro-benchmark only: the weight stream alone)
        return make_float4(float(p), 0.f, 0.f, 0.f);
#endif
        return *reinterpret_cast<const float4*>(
            x + static_cast<int64_t>(c) * ldx + column + min(p, last_quad));
    };
    auto masked = [&](int q, const float4& v) {
        const int p = c0 + 4 * q;
        return make_float4(p < count ? v.x : 0.f, p + 1 < count ? v.y : 0.f,
                           p + 2 < count ? v.z : 0.f, p + 3 < count ? v.w : 0.f);
    };
    auto deposit = [&](int c, int q, const float4& raw) {
        const float4 v = masked(q, raw);
        float* target = act + c * kStackStride + 4 * q;
        target[1] = v.x;
        *reinterpret_cast<f32x2*>(target + 2) = f32x2{v.y, v.z};
        target[4] = v.w;
    };
    constexpr int kEarlyRows = 16;
    if (loader) {
        // The loader waves' row loads and LDS writes are inline asm with explicit
        // s_waitcnt: hipcc waits for vmcnt(0) at the first use of ANY load result
        // while an LDS-DMA is outstanding (two kinds of events on one counter) and
        // in front of every LDS access it can see behind one, which would turn the
        // staged hand-over below into "wait for everything".  Straight-line code
        // with a register set per stage: a loop would carry the registers of loads
        // in flight around its back edge, where hipcc is free to copy them.
        // vmcnt counts in order, so a wait is a position in the wave's issue order:
        //   chunk 1 (8 requests) | rows of chunk 1 (5 loads) | rows of chunk 2 |
        //   chunk 2 | rows of chunk 3 | chunk 3 | rows of chunk 4 | chunk 4
        // Stage c (under chunk c - 1 of layer 0): request chunk c of the weights,
        // ask for the rows chunk c + 1 multiplies, hand over the rows of chunk c,
        // see chunk c land.  The barriers pair with the MFMA waves' (one per chunk,
        // plus one per layer between the last read and the first write of the
        // activations); layer 0's are the bare instruction (__syncthreads() waits
        // for every load).
        constexpr int kStage = 16 * 64 / 256;                    // 4 loads: the rows of a chunk
        static_assert(kStage == 4, "the s_waitcnt below count these loads");
        f32x4 rows[kStackChunks - 1][kStage];
        float side[kStackChunks - 1];
        const int mine = threadIdx.x - 512;
        const int q = mine & 63;
        const int p = c0 + 4 * q;
        const float* source = x + static_cast<int64_t>(mine >> 6) * ldx + column + min(p, last_quad);
        const uint32_t target = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
            (__attribute__((address_space(3))) float*)(act + (mine >> 6) * kStackStride + 4 * q)));
        // the columns beside the computed positions: threads 0 .. 63, a (row, kind)
        // each; the others repeat kind 3, the row's zero padding
        const int side_kind = mine < 64 ? (mine & 3) : 3;
        const int side_p = side_kind == 0 ? c0 - 1 : c0 + kStackWidth + side_kind - 1;
        const bool side_real = side_kind < 3 && side_p >= 0 && side_p < count;
        const float* side_source = x + static_cast<int64_t>((mine >> 2) & 15) * ldx + column +
                                   (side_real ? side_p : 0);
        const uint32_t side_target = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
            (__attribute__((address_space(3))) float*)(
                act + ((mine >> 2) & 15) * kStackStride +
                (side_kind == 0 ? 0 : kStackWidth + side_kind))));
        auto ask = [&](int chunk) {                              // rows 16 chunk + 4 round + (mine >> 6)
#pragma unroll
            for (int round = 0; round < kStage; ++round) {
                const float* address = source + static_cast<int64_t>(16 * chunk + 4 * round) * ldx;
                asm volatile("global_load_dwordx4 %0, %1, off"
                             : "=v"(rows[chunk - 1][round]) : "v"(address) : "memory");
            }
            const float* address = side_source + static_cast<int64_t>(16 * chunk) * ldx;
            asm volatile("global_load_dword %0, %1, off" : "=v"(side[chunk - 1]) : "v"(address) : "memory");
        };
        auto hand_over = [&](int chunk) {
#pragma unroll
            for (int round = 0; round < kStage; ++round) {
                const f32x4 raw = rows[chunk - 1][round];
                const float first = p < count ? raw[0] : 0.f;
                const f32x2 middle = {p + 1 < count ? raw[1] : 0.f, p + 2 < count ? raw[2] : 0.f};
                const float last = p + 3 < count ? raw[3] : 0.f;
                const uint32_t address = target + (16 * chunk + 4 * round) * (kStackStride * 4);
                asm volatile(
                    "ds_write_b32 %0, %1 offset:4\n\t"
                    "ds_write_b64 %0, %2 offset:8\n\t"
                    "ds_write_b32 %0, %3 offset:16"
                    :
                    : "v"(address), "v"(first), "v"(middle), "v"(last)
                    : "memory");
            }
            const float beside = side_real ? side[chunk - 1] : 0.f;
            const uint32_t address = side_target + 16 * chunk * (kStackStride * 4);
            asm volatile("ds_write_b32 %0, %1" : : "v"(address), "v"(beside) : "memory");
        };
#define EMPH_ROWS_HERE(COUNT, STAGE)                                                        \
    asm volatile("s_waitcnt vmcnt(" #COUNT ")"                                              \
                 : "+v"(rows[STAGE][0]), "+v"(rows[STAGE][1]), "+v"(rows[STAGE][2]),        \
                   "+v"(rows[STAGE][3]), "+v"(side[STAGE])::"memory")
        request(1, 8, 4);
        ask(1);
        __builtin_amdgcn_s_barrier();                            // chunk 0 starts
        ask(2);
        EMPH_ROWS_HERE(5, 0);                                    // (and chunk 1, in front of them)
        hand_over(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // chunk 1 starts
        request(2, 8, 4);
        ask(3);
        EMPH_ROWS_HERE(13, 1);
        hand_over(2);
        asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // chunk 2 starts
        request(3, 8, 4);
        ask(4);
        EMPH_ROWS_HERE(13, 2);
        hand_over(3);
        asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // chunk 3 starts
        request(4, 8, 4);
        EMPH_ROWS_HERE(8, 3);
        hand_over(4);
#undef EMPH_ROWS_HERE
        for (int g = 4; g < total_chunks; ++g) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            STACK_STAMP(2 + 8 * (g / kStackChunks) + g % kStackChunks);
            __syncthreads();
            if (g + 1 < total_chunks) request(g + 1, 8, 4);
            // (the MFMA waves' barrier in front of a layer's in-place update: every
            // layer but the launch's last)
            if ((g + 1) % kStackChunks == 0 && g + 1 < total_chunks) __syncthreads();
        }
        return;
    }
    request(0, 0, 8);
    {
        // (every load in front of the first LDS write: that write waits for the DMA)
        static_assert(kStackMaxLayers * kStackChannels <= 512, "one bias per thread");
        const bool has_bias = threadIdx.x < layers * kStackChannels;
        const float bias_value = has_bias ? biases[threadIdx.x] : 0.f;
        constexpr int kEarly = kEarlyRows * 64 / 512;                    // 2 per thread
        float4 early[kEarly];
#pragma unroll
        for (int round = 0; round < kEarly; ++round) {
            const int index = threadIdx.x + 512 * round;
            early[round] = fetch(index >> 6, index & 63);
        }
        // the columns beside the computed positions, these sixteen rows
        float side = 0.f;
        const int side_row = threadIdx.x >> 2, side_kind = threadIdx.x & 3;
        if (side_row < kEarlyRows) {
            const float* row = x + static_cast<int64_t>(side_row) * ldx + column;
            const int p = side_kind == 0 ? c0 - 1 : c0 + kStackWidth + side_kind - 1;
            if (side_kind < 3 && p >= 0 && p < count) side = row[p];
        }
#pragma unroll
        for (int round = 0; round < kEarly; ++round) {
            const int index = threadIdx.x + 512 * round;
            deposit(index >> 6, index & 63, early[round]);
        }
        if (side_row < kEarlyRows)
            act[side_row * kStackStride + (side_kind == 0 ? 0 : kStackWidth + side_kind)] = side;
        if (has_bias) bias_lds[threadIdx.x] = bias_value;
    }
    STACK_STAMP(1);

    // ---- MFMA waves: column tile `tile` of 16 quads, m-tiles split 3 + 2 between
    // the two waves of a SIMD (conv_w4.hip)
    const int kk = lane >> 4;
    const int col = lane & 15;
    const int tile = wave & 3;
    const int part = wave >> 2;
    constexpr int split = (kStackMTiles + 1) >> 1;
    const int m_begin = part ? split : 0;
    // the lane's quad: LDS index of x[4q - 1] in row kk
    const float* lane_rows = act + kk * kStackStride + 64 * tile + 4 * col;
    const int p_quad = c0 + 64 * tile + 4 * col;            // position of the quad's first output

    auto run = [&](auto count_tag) {
        constexpr int COUNT = decltype(count_tag)::value;
        for (int layer = 0; layer < layers; ++layer) {
            f32x4 acc[6][COUNT];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int m = 0; m < COUNT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            float a[6][COUNT], d[6], v[6];
            auto load_b = [&](int step) {
                const float* source = lane_rows + 4 * step * kStackStride;
                const f32x4 first = *reinterpret_cast<const f32x4*>(source);
                const f32x2 second = *reinterpret_cast<const f32x2*>(source + 4);
                d[0] = first[0], d[1] = first[1], d[2] = first[2], d[3] = first[3];
                d[4] = second[0], d[5] = second[1];
            };
            for (int chunk = 0; chunk < kStackChunks; ++chunk) {
                // the chunk has landed (and, chunk 0: every wave has written its
                // part of this layer's input)
                __syncthreads();
                STACK_STAMP(2 + 8 * layer + chunk);
                // (the launch's input rows 16 c .. 16 c + 15 arrive under chunk c - 1,
                // whose last k-step prefetched what may predate them - read it again)
                if (chunk == 0 || layer == 0) load_b(chunk * kStackChunkSteps);
                const float* weights =
                    ring + ((layer * kStackChunks + chunk) & 1) * kStackChunkFloats +
                    (m_begin << 6) + lane;
#pragma unroll
                for (int ks = 0; ks < kStackChunkSteps; ++ks) {
                    const int step = chunk * kStackChunkSteps + ks;
#pragma unroll
                    for (int j = 0; j < 6; ++j)
#pragma unroll
                        for (int m = 0; m < COUNT; ++m)
#ifdef STACK_FEWER_A_READS         // (micro-benchmark only: what the LDS reads between MFMAs cost)
                            a[j][m] = (j & 1) ? a[j - 1][m]
                                              : weights[(ks * 6 * kStackMTiles + j * kStackMTiles + m) << 6];
#else
                            a[j][m] = weights[(ks * 6 * kStackMTiles + j * kStackMTiles + m) << 6];
#endif
#ifdef STACK_NO_TRANSFORM          // (micro-benchmark only: what the vector work costs)
                    for (int j = 0; j < 6; ++j) v[j] = d[j];
#else
                    // v = B^T d (conv_w4.hip: the same operations in the same order)
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
#endif
                    if (step + 1 < kStackSteps) load_b(step + 1);
#pragma unroll
                    for (int j = 0; j < 6; ++j)
#pragma unroll
                        for (int m = 0; m < COUNT; ++m)
                            acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                a[j][m], v[j], acc[j][m], 0, 0, 0);
                }
            }
            // ---- output transform, bias, activation
            const bool relu = (relu_mask >> layer) & 1;
            const bool last = layer == layers - 1;
            const float* bias_row = bias_lds + layer * kStackChannels;
            STACK_STAMP(2 + 8 * layer + 5);
            if (!last) {
                // every wave is done reading this layer's input: its output may
                // take the rows' place (zeros outside the segment: 'same' padding)
                __syncthreads();
#pragma unroll
                for (int m = 0; m < COUNT; ++m) {
                    const int channel0 = 16 * (m_begin + m) + 4 * kk;
                    const f32x4 add = *reinterpret_cast<const f32x4*>(bias_row + channel0);
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
                        o0 = p_quad < count ? o0 : 0.f;
                        o1 = p_quad + 1 < count ? o1 : 0.f;
                        o2 = p_quad + 2 < count ? o2 : 0.f;
                        o3 = p_quad + 3 < count ? o3 : 0.f;
                        float* target =
                            act + (channel0 + r) * kStackStride + 64 * tile + 4 * col + 1;
                        target[0] = o0;
                        *reinterpret_cast<f32x2*>(target + 1) = f32x2{o1, o2};
                        target[3] = o3;
                    }
                }
                STACK_STAMP(2 + 8 * layer + 6);
                continue;
            }
            // ---- the launch's last layer: the span's own positions leave the chip
            const int t = p_quad;
            const int owned_end = owned_first + owned;
            if (WORD_SUMS) {
                int4 slots = {-1, -1, -1, -1};
                if (t < count) slots = *reinterpret_cast<const int4*>(slot_map + column + t);
#pragma unroll
                for (int m = 0; m < COUNT; ++m) {
                    const int channel0 = 16 * (m_begin + m) + 4 * kk;
                    const f32x4 add = *reinterpret_cast<const f32x4*>(bias_row + channel0);
                    f32x4 sum[4];
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
                        // only the span's own positions count (a halo position
                        // belongs to the neighbouring span's sums)
                        o0 = (t >= owned_first && t < owned_end) ? o0 : 0.f;
                        o1 = (t + 1 >= owned_first && t + 1 < owned_end) ? o1 : 0.f;
                        o2 = (t + 2 >= owned_first && t + 2 < owned_end) ? o2 : 0.f;
                        o3 = (t + 3 >= owned_first && t + 3 < owned_end) ? o3 : 0.f;
                        o1 += o0;
                        o2 += o1;
                        o3 += o2;
                        float scan = o3;
                        scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, scan), 0x111, 0xf, 0xf, true));
                        scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, scan), 0x112, 0xf, 0xf, true));
                        scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, scan), 0x114, 0xf, 0xf, true));
                        scan += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, scan), 0x118, 0xf, 0xf, true));
                        const float before = scan - o3;
                        sum[0][r] = o0 + before;
                        sum[1][r] = o1 + before;
                        sum[2][r] = o2 + before;
                        sum[3][r] = o3 + before;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int slot = j == 0 ? slots.x : j == 1 ? slots.y : j == 2 ? slots.z : slots.w;
                        if (slot >= 0 && t + j >= owned_first && t + j < owned_end)
                            *reinterpret_cast<f32x4*>(y + static_cast<int64_t>(slot) * ldy +
                                                      channel0) = sum[j];
                    }
                }
                continue;
            }
            const bool vector_ok = (ldy & 3) == 0 && (column & 3) == 0 &&
                                   (reinterpret_cast<uintptr_t>(y) & 15) == 0;
#pragma unroll
            for (int m = 0; m < COUNT; ++m) {
                const int channel0 = 16 * (m_begin + m) + 4 * kk;
                const f32x4 add = *reinterpret_cast<const f32x4*>(bias_row + channel0);
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
                    float* target = y + static_cast<int64_t>(channel0 + r) * ldy + column + t;
                    if (vector_ok && t >= owned_first && t + 3 < owned_end) {
                        *reinterpret_cast<float4*>(target) = out;
                    } else {
                        if (t >= owned_first && t < owned_end) target[0] = out.x;
                        if (t + 1 >= owned_first && t + 1 < owned_end) target[1] = out.y;
                        if (t + 2 >= owned_first && t + 2 < owned_end) target[2] = out.z;
                        if (t + 3 >= owned_first && t + 3 < owned_end) target[3] = out.w;
                    }
                }
            }
            STACK_STAMP(2 + 8 * layer + 6);
        }
    };
    if (part == 0) run(std::integral_constant<int, split>{});
    else run(std::integral_constant<int, kStackMTiles - split>{});
}

}  // namespace emph

using namespace emph;

extern "C" {

int32_t emph_conv_stack_max_layers(void) { return kStackMaxLayers; }

// Spans of a packed axis for emph_conv1d_stack: every segment of `counts[i]`
// positions at frame column `offsets[i]` is cut into the fewest spans a
// workgroup can own - 256 positions for a whole segment, 252 for a span at one
// end (a halo of 4 recomputed positions on the other side), 248 in between -
// of even size, quads (4 positions) never split.  host_spans == NULL: returns
// the number of spans.
int32_t emph_conv_stack_spans(const int64_t* host_counts, const int64_t* host_offsets,
                              int32_t n_segments, int32_t* host_spans) {
    int32_t total = 0;
    for (int32_t segment = 0; segment < n_segments; ++segment) {
        const int64_t count = host_counts[segment];
        if (count <= 0) continue;
        const int64_t quads = (count + 3) / 4;
        int64_t pieces = 1;
        if (quads > 64) {
            pieces = 2;
            while (2 * 63 + (pieces - 2) * 62 < quads) ++pieces;
        }
        // even shares of the quads; what does not divide goes to the ends first
        // (they may hold 63 quads), then to the spans in between
        const int64_t base = quads / pieces;
        int64_t spare = quads - base * pieces;
        int64_t extra[3] = {0, 0, 0};         // first, last, in between (count)
        if (pieces > 1) {
            if (spare > 0 && base + 1 <= 63) extra[0] = 1, --spare;
            if (spare > 0 && base + 1 <= 63) extra[1] = 1, --spare;
            extra[2] = spare;
        }
        int64_t done = 0;
        for (int64_t k = 0; k < pieces; ++k) {
            int64_t share = base;
            if (pieces == 1) share = quads;
            else if (k == 0) share += extra[0];
            else if (k == pieces - 1) share += extra[1];
            else if (k <= extra[2]) share += 1;
            if (k == pieces - 1) share = quads - done;
            if (host_spans != nullptr) {
                int32_t* row = host_spans + static_cast<int64_t>(total) * kSpanFields;
                const int64_t first = 4 * done;
                int64_t owned = 4 * share;
                if (first + owned > count) owned = count - first;
                row[0] = segment;
                row[1] = static_cast<int32_t>(first);
                row[2] = static_cast<int32_t>(host_offsets[segment]);
                row[3] = static_cast<int32_t>(count);
                row[4] = static_cast<int32_t>(owned);
                row[5] = static_cast<int32_t>(first > 0 ? first - 4 : 0);
                row[6] = row[7] = 0;
            }
            done += share;
            ++total;
        }
    }
    return total;
}

// `layers` (1 .. emph_conv_stack_max_layers()) consecutive Conv1d(80, 80, 3, 'same') layers in
// one launch.
//   packs   float32: emph_conv_winograd4_pack of every layer, back to back
//   biases  float32 [layers][80]
//   relu_mask  bit l: layer l is followed by ReLU (else identity)
//   spans   int32 [n_spans][8] from emph_conv_stack_spans (device copy)
//   slot_map != NULL: the last layer leaves running sums in y = sums[slot][ldy]
//   (emph_conv1d_winograd4_word_sums; the running sum restarts at every span's
//   first own position and every 64 computed positions: `Plan.word_sum_tables`
//   with the spans' restart columns)
int emph_conv1d_stack(const float* x, int64_t ldx, float* y, int64_t ldy, const float* packs,
                      const float* biases, int32_t layers, int32_t relu_mask,
                      const int32_t* spans, int32_t n_spans, const int32_t* slot_map,
                      void* stream) {
    if (n_spans == 0) return EMPH_OK;
    EMPH_REQUIRE(x && y && packs && biases && spans, EMPH_EINVAL,
                 "emph_conv1d_stack: null pointer");
    EMPH_REQUIRE(layers >= 1 && layers <= kStackMaxLayers, EMPH_ERANGE,
                 "emph_conv1d_stack: %d layers (1 .. %d)", layers, kStackMaxLayers);
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (ldx & 3) == 0, EMPH_EINVAL,
                 "emph_conv1d_stack: the input must be 16-byte aligned with ldx a multiple of 4");
    EMPH_REQUIRE(slot_map == nullptr ||
                     ((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (ldy & 3) == 0 &&
                      ldy >= kStackChannels && (reinterpret_cast<uintptr_t>(slot_map) & 15) == 0),
                 EMPH_EINVAL, "emph_conv1d_stack: bad sums buffer or slot map");
    const size_t lds = static_cast<size_t>(stack_lds_floats()) * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (slot_map != nullptr) {
        auto kernel = conv1d_stack_kernel<true>;
        static LdsReservation reserved;
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,
                                     "emph_conv1d_stack"))
            return status;
        EMPH_LAUNCH(kernel, dim3(n_spans), dim3(kStackThreads), lds, s, x, ldx, y, ldy, packs,
                    biases, layers, relu_mask, spans, slot_map);
    } else {
        auto kernel = conv1d_stack_kernel<false>;
        static LdsReservation reserved;
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,
                                     "emph_conv1d_stack"))
            return status;
        EMPH_LAUNCH(kernel, dim3(n_spans), dim3(kStackThreads), lds, s, x, ldx, y, ldy, packs,
                    biases, layers, relu_mask, spans, slot_map);
    }
    return check_launch("emph_conv1d_stack");
}

}  // extern "C"
