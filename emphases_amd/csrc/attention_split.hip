// The attention core of the Transformer encoder (emphases/model/layers/
// transformer.py:18-30; torch's scaled_dot_product over nhead = 2) for LONG
// segments on the bf16 matrix pipe, with every fp32 operand SPLIT into bf16
// pieces so that the result stays at fp32 grade - an opt-in (`precision=` of
// the engine; the default stays the fp32-MFMA kernel of transformer.hip).
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (1/16 of the bf16
// matrix rate) and shares the vector ALU's multipliers - profiles/r5_coexec.txt:
// nothing a wave's VALU does hides beside it, and 190 us per layer of BASELINE
// configs[2] is matrix time.  v_mfma_f32_32x32x16_bf16 multiplies bf16 exactly
// (8 x 8 bits fit fp32's 24) and accumulates in fp32, and the softmax's vector
// work runs BESIDE it.  So with x = x0 + x1 (+ x2), pieces of 8 significant bits,
//     PIECES = 2 ("bf16x3"):  a.b ~ a0.b0 + a0.b1 + a1.b0            3 MFMAs,
//         pieces rounded to nearest: each product within 2^-16 of exact
//     PIECES = 3 ("bf16x6"):  + a0.b2 + a1.b1 + a2.b0                6 MFMAs,
//         pieces by truncation, x0 + x1 + x2 == x EXACTLY; the three dropped
//         products are below 2^-23 of the result: the error of one fp32 rounding
// against 16 units of pipe time for the fp32 instruction.
//
// Shapes.  S^T = K Q^T (keys are MFMA rows, queries columns) in 32 x 32 tiles, the
// head dimension (40 -> 48) in three k-steps; O^T = V^T P^T with rows d (40 ->
// 64: row 40 multiplies ONES, so it accumulates the softmax denominator; rows
// 41 .. 63 read one shared row of zeros) and 32 keys in two k-steps.  As in
// transformer.hip the probabilities never move: the accumulator registers of
// S^T ARE the B operand of the second product once split (register r of lane
// half h holds key 8 (r / 4) + 4 h + r % 4; the V image is stored with the keys
// of each 16 permuted to match, so a lane's eight keys are one 16-byte read).
// Online softmax with the lazy reference of transformer.hip (scores in log2
// units, accumulators start at -reference, a wave-uniform rare rescale).
//
// A workgroup is eight waves of 32 queries = 256 consecutive queries of one
// segment and head (the tile table of attention_group_kernel<D, 8>); stages of
// 64 keys, double buffered.  The fp32 keys / values of the next stage are in
// flight to registers during the MFMAs and are split + transposed on their way
// into LDS - once per workgroup, not per wave:
//     K image  [piece][d / 8][key][8 d]  bf16   (A fragment of a k-step: 16 B)
//     V image  [piece][d][64 keys, permuted]  bf16, row stride 144 B
#include <type_traits>

#include "common.h"

// (tools/micro/attention_split_bench.hip defines SPLIT_STAMP for an in-kernel timeline)
#ifndef SPLIT_STAMP
#define SPLIT_STAMP(slot)
#define SPLIT_STAMP_ARGUMENT
#define SPLIT_STAMP_FINISH
#define SPLIT_STAMP_DECLARE
#define SPLIT_STAMP_PASS
#endif

#include "split.h"

namespace emph {

constexpr int kSplitWaves = 8;
constexpr int kSplitQueries = 32 * kSplitWaves;       // per workgroup
constexpr int kSplitRing = 4;                         // stages in LDS

// grid = n_tiles (blocks of 64 positions of the frame axis); block = 256
template <int D, int PK, int PV>
__global__ __launch_bounds__(256) void split_kv_kernel(
    const float* __restrict__ qk, const float* __restrict__ v, int64_t ld, int channels,
    int heads, const int32_t* __restrict__ tiles, unsigned char* __restrict__ images) {
    typedef SplitImages<D, PK, PV> Images;
    constexpr int STAGE = kSplitStage;
    const Tile tile = load_tile(tiles, blockIdx.x);
    const int slot = (tile.offset >> 6) + tile.segment + (tile.first >> 6);
    const int valid = tile.count - tile.first;            // keys of this stage that exist
    const int k_tasks = heads * (D / 8) * STAGE;
    const int v_tasks = heads * (STAGE / 8) * D;
    for (int task = threadIdx.x; task < k_tasks + v_tasks; task += blockDim.x) {
        float values[8];
        int head;
        int byte;
        const bool is_key = task < k_tasks;
        if (is_key) {
            const int key = task % STAGE, octet = task / STAGE % (D / 8);
            head = task / STAGE / (D / 8);
            const float* source = qk + static_cast<int64_t>(channels + head * D + 8 * octet) * ld +
                                  tile.offset + tile.first + min(key, max(valid - 1, 0));
#pragma unroll
            for (int e = 0; e < 8; ++e)
                values[e] = key < valid ? source[static_cast<int64_t>(e) * ld] : 0.f;
            byte = (octet * STAGE + key) * 16;
        } else {
            const int index = task - k_tasks;
            const int d = index % D, chunk = index / D % (STAGE / 8);
            head = index / D / (STAGE / 8);
            const int group = chunk >> 1, h = chunk & 1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = 16 * group + 8 * (e >> 2) + 4 * h + (e & 3);
                values[e] = key < valid
                                ? v[static_cast<int64_t>(tile.offset + tile.first + key) * channels +
                                    head * D + d]
                                : 0.f;
            }
            byte = (chunk * Images::kRows + d) * 16;
        }
        unsigned char* stage =
            images + (static_cast<int64_t>(slot) * heads + head) * Images::kStageBytes;
        if (is_key) {
            u32x4 parts[PK];
            split_eight<PK>(values, parts);
#pragma unroll
            for (int piece = 0; piece < PK; ++piece)
                *reinterpret_cast<u32x4*>(stage + Images::key_piece(piece) + byte) = parts[piece];
        } else {
            u32x4 parts[PV];
            split_eight<PV>(values, parts);
#pragma unroll
            for (int piece = 0; piece < PV; ++piece)
                *reinterpret_cast<u32x4*>(stage + Images::value_piece(piece) + byte) = parts[piece];
        }
    }
    // the constant parts: K's octets from D / 8 on (ones at d = D in piece 0), V's ones
    // and zero rows
    constexpr int PAD_OCTETS = Images::kOctets - D / 8;
    const int k_fill = heads * PK * PAD_OCTETS * STAGE;
    const int v_fill = heads * PV * (STAGE / 8) * 2;
    for (int index = threadIdx.x; index < k_fill + v_fill; index += blockDim.x) {
        int head, byte;
        u32x4 fill = {0u, 0u, 0u, 0u};
        if (index < k_fill) {
            const int key = index % STAGE, octet = D / 8 + index / STAGE % PAD_OCTETS;
            const int piece = index / STAGE / PAD_OCTETS % PK;
            head = index / STAGE / PAD_OCTETS / PK;
            if (octet == D / 8 && piece == 0) fill[0] = 0x3f80u;        // bf16 1.0 at d = D
            byte = Images::key_piece(piece) + (octet * STAGE + key) * 16;
        } else {
            const int rest = index - k_fill;
            const int chunk = rest % (STAGE / 8), row = D + rest / (STAGE / 8) % 2;
            const int piece = rest / (STAGE / 8) / 2 % PV;
            head = rest / (STAGE / 8) / 2 / PV;
            if (row == D && piece == 0) fill = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            byte = Images::value_piece(piece) + (chunk * Images::kRows + row) * 16;
        }
        *reinterpret_cast<u32x4*>(
            images + (static_cast<int64_t>(slot) * heads + head) * Images::kStageBytes + byte) = fill;
    }
}

// grid = (n_tiles, heads); block = 512
template <int D, int PK, int PV>
__global__ __launch_bounds__(64 * kSplitWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attention_split_kernel(const float* __restrict__ qk, const unsigned char* __restrict__ images,
                            float* __restrict__ out, int64_t ld, int channels,
                            const int32_t* __restrict__ tiles,
                            const int32_t* __restrict__ key_counts SPLIT_STAMP_ARGUMENT) {
    typedef SplitImages<D, PK, PV> Images;
    SPLIT_STAMP_DECLARE
    constexpr int THREADS = 64 * kSplitWaves;
    constexpr int KSTEPS = (D + 15) / 16;                 // of the S^T product
    constexpr int STAGE = kSplitStage;
    constexpr int UNITS = Images::kStageBytes / 16;       // 16-byte units of a stage
    constexpr int PASSES = (UNITS + THREADS - 1) / THREADS;
    static_assert(PASSES >= 3 && PASSES <= 5, "the s_waitcnt below count the requests");
    static_assert(D % 8 == 0 && D % 16 != 0 && D > 32 && D + 2 <= 48,
                  "an m-tile of 32 rows and a tail of 16 with a spare row for the ones, a spare d "
                  "for the reference");
    extern __shared__ __align__(16) unsigned char split_lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int column = lane & 31;            // the lane's query inside the wave's 32
    const int half = lane >> 5;
    // XCD-aware order of the (head, tile) space: transformer.hip, attention_group_kernel
    const int linear = blockIdx.x + gridDim.x * blockIdx.y;
    const int total = gridDim.x * gridDim.y;
    const int per_xcd = total >> 3;
    const int logical = linear < 8 * per_xcd ? (linear & 7) * per_xcd + (linear >> 3) : linear;
    const int head = logical / static_cast<int>(gridDim.x);
    const int heads = gridDim.y;
    const Tile span = load_tile(tiles, logical - head * static_cast<int>(gridDim.x));
    const int q0 = span.first + 32 * wave;
    const int queries = span.count;
    const int length = key_counts != nullptr ? min(key_counts[span.segment], span.count)
                                             : span.count;
    const bool working = q0 < queries;            // wave-uniform
    const float scale = 1.44269504088896340736f / sqrtf(static_cast<float>(D));
    const float* q_rows = qk + static_cast<int64_t>(head * D) * ld + span.offset;
    // (wave-uniform, and known to be: the stage's address stays in scalar registers)
    const int first_slot = __builtin_amdgcn_readfirstlane((span.offset >> 6) + span.segment);
    const int stages = __builtin_amdgcn_readfirstlane((length + STAGE - 1) / STAGE);

    // ---- staging: a stage's image, as it lies in memory, by LDS-DMA (1 KB per wave
    // instruction; the last pass of a stage is partial)
    // Every wave issues exactly PASSES requests per stage (what lies beyond the image
    // repeats its last 1 KB; a stage beyond the last repeats the last stage), so that
    // "all but the newest stage have landed" is the constant s_waitcnt vmcnt(PASSES).
    static_assert(UNITS >= 64, "whole 1 KB wave requests (the last one overlaps its predecessor)");
    auto request = [&](int stage) {
        const int buffer = stage & (kSplitRing - 1);
        const unsigned char* source =
            images + (static_cast<int64_t>(first_slot + max(min(stage, stages - 1), 0)) * heads + head) *
                         Images::kStageBytes;
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            const int first_unit = min(THREADS * pass + 64 * wave, UNITS - 64);     // wave-uniform
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(source + 16 * (first_unit + lane)),
                (__attribute__((address_space(3))) void*)(split_lds + buffer * Images::kStageBytes +
                                                         16 * first_unit),
                16, 0, 0);
        }
    };
    auto key_image = [&](int buffer, int piece) {
        return split_lds + buffer * Images::kStageBytes + Images::key_piece(piece);
    };
    auto value_image = [&](int buffer, int piece) {
        return split_lds + buffer * Images::kStageBytes + Images::value_piece(piece);
    };

    // ---- the wave's queries: B fragments of Q^T, scaled, split once.  k-step
    // KSTEPS - 1 of the upper half covers d = D .. : slot d = D carries -reference.
    // All loads are issued before anything waits (a clamped address and a factor of
    // zero instead of a branch per element: hipcc sinks a conditional load into its
    // branch and waits for each in turn - 24 trips to memory, a fifth of the kernel).
    u32x4 bq[KSTEPS][PK];
    {
        const int query = q0 + column;
        const float* source = q_rows + min(query, queries - 1);
        float raw[KSTEPS][8];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                raw[s][e] = source[static_cast<int64_t>(min(16 * s + 8 * half + e, D - 1)) * ld];
        request(0);
        request(1);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            float values[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                values[e] = raw[s][e] *
                            ((16 * s + 8 * half + e < D && query < queries) ? scale : 0.f);
            split_eight<PK>(values, bq[s]);
        }
    }
    auto set_reference = [&](float reference) {
        // (lanes of the upper half: element 0 of the last k-step is d = D)
        uint32_t parts[PK];
        split_pair<PK>(-reference, 0.f, parts);
#pragma unroll
        for (int piece = 0; piece < PK; ++piece)
            if (half == (D % 16) / 8) bq[KSTEPS - 1][piece][(D % 8) / 2] = parts[piece];
    };
    // O^T: rows 0 .. 31 in the 32 x 32 layout; rows 32 .. 47 as two 16 x 16 tiles (queries
    // 0 .. 15 and 16 .. 31 of the wave: lane l holds query 16 q + l % 16, rows 32 + 4 (l / 16) + i)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    f32x4 o_tail[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float reference = 0.f;

    // S^T - reference of one block of 32 keys: A fragments of K (lane = key, half), the
    // accumulators start from zero (the reference rides in Q^T's spare slot)
    auto scores = [&](int buffer, int local) {
        f32x16 s16;
#pragma unroll
        for (int r = 0; r < 16; ++r) s16[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            u32x4 ak[PK];
#pragma unroll
            for (int piece = 0; piece < PK; ++piece)
                ak[piece] = *reinterpret_cast<const u32x4*>(
                    key_image(buffer, piece) + ((2 * s + half) * STAGE + local + column) * 16);
            s16 = split_product<PK>(ak, bq[s], s16);
        }
        return s16;
    };
    // V fragments of a block.  Rows 0 .. 31 of V^T: row `column`, the eight keys 16 ks +
    // 8 half .. of k-step ks (v_mfma_f32_32x32x16_bf16).  Rows 32 .. 47 - d 32 .. D - 1, the
    // row of ones, the shared row of zeros - are ONE 16-row tile of v_mfma_f32_16x16x32_bf16
    // (all 32 keys of the block per instruction, 16 queries): lane (row 32 + l % 16, group
    // g = l / 16) reads the eight keys of (k-step g % 2, half g / 2), which is where the
    // swapped probabilities of `attend` put them.
    auto values = [&](int buffer, int local, u32x4 (&av)[2][PV], u32x4 (&tail)[PV]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int piece = 0; piece < PV; ++piece)
                av[ks][piece] = *reinterpret_cast<const u32x4*>(
                    value_image(buffer, piece) +
                    (((local >> 3) + 2 * ks + half) * Images::kRows + column) * 16);
        const int group = lane >> 4;
        const int row = min(32 + (lane & 15), D + 1);
#pragma unroll
        for (int piece = 0; piece < PV; ++piece)
            tail[piece] = *reinterpret_cast<const u32x4*>(
                value_image(buffer, piece) +
                (((local >> 3) + 2 * (group & 1) + (group >> 1)) * Images::kRows + row) * 16);
    };
    // softmax numerators of a block's scores and their product with the values
    // (`pending`: the scores of the NEXT block, already issued against the reference as
    // it stands - when the reference moves they move with it)
    auto attend = [&](f32x16 s16, f32x16& pending, const u32x4 (&av)[2][PV],
                      const u32x4 (&tail)[PV], int key0, bool masked) {
        SPLIT_STAMP(1);
        if (masked) {                     // wave-uniform: a segment's last block only
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (key0 + 8 * (r >> 2) + 4 * half + (r & 3) >= length) s16[r] = -INFINITY;
        }
        // s16 = score - reference; the reference moves only when some score of the wave
        // is more than 2^64 above it
        float top = fmaxf(s16[0], s16[1]);
#pragma unroll
        for (int r = 2; r < 16; ++r) top = fmaxf(top, s16[r]);
        const bool moved = __builtin_amdgcn_ballot_w64(top > 64.f) != 0;
        SPLIT_STAMP(2);                   // the block's S^T complete, maximum known
        if (moved) {                      // wave-uniform, rare
            const float shift = fmaxf(fmaxf(top, __shfl_xor(top, 32)), 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
            // (the tail tiles hold other queries than the lane's own: 16 q + lane % 16)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float alpha_q = __shfl(alpha, 16 * q + (lane & 15));
#pragma unroll
                for (int i = 0; i < 4; ++i) o_tail[q][i] *= alpha_q;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s16[r] -= shift;
#pragma unroll
            for (int r = 0; r < 16; ++r) pending[r] -= shift;
            reference += shift;
            set_reference(reference);
        }
        u32x4 bp[2][PV];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float p[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) p[e] = __builtin_amdgcn_exp2f(s16[8 * ks + e]);
            split_eight<PV>(p, bp[ks]);
        }
        SPLIT_STAMP(3);                   // probabilities split
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) o = split_product<PV>(av[ks], bp[ks], o);
        // the tail tile's B operands: v_permlane16_swap trades lanes 16 .. 31 (48 .. 63) of
        // k-step 0's fragment with lanes 0 .. 15 (32 .. 47) of k-step 1's - afterwards the
        // first holds queries 0 .. 15, the second queries 16 .. 31, each with the keys of
        // (k-step g % 2, half g / 2) in lane group g
        u32x4 bt[2][PV];
#pragma unroll
        for (int piece = 0; piece < PV; ++piece)
#pragma unroll
            for (int word = 0; word < 4; ++word) {
                const auto pair = __builtin_amdgcn_permlane16_swap(bp[0][piece][word],
                                                                   bp[1][piece][word], false, false);
                bt[0][piece][word] = pair[0];
                bt[1][piece][word] = pair[1];
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int order = PV - 1; order >= 0; --order)       // the smallest products first
#pragma unroll
                for (int i = 0; i <= order; ++i)
                    o_tail[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, tail[i]),
                        __builtin_bit_cast(bf16x8, bt[q][order - i]), o_tail[q], 0, 0, 0);
        SPLIT_STAMP(4);                   // O^T issued
    };
    // the next stage for every wave: this wave's share has landed (all but the newest
    // request group), then the workgroup's barrier
    auto next_stage = [&] {
        SPLIT_STAMP(5);
        if (PASSES == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (PASSES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        SPLIT_STAMP(6);
        __syncthreads();
        SPLIT_STAMP(7);
    };

    // (Q's loads are older than the requests: they have landed too)
    next_stage();
    f32x16 current;
#pragma unroll
    for (int r = 0; r < 16; ++r) current[r] = 0.f;
    if (working && stages > 0) {
        // the reference starts near the maximum of the first 32 keys' scores (the leading
        // pieces only: any value within 2^64 of the maximum serves)
        f32x16 s16;
#pragma unroll
        for (int r = 0; r < 16; ++r) s16[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(
                key_image(0, 0) + ((2 * s + half) * STAGE + column) * 16);
            s16 = mfma_bf16(a, bq[s][0], s16);
        }
        float top = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (8 * (r >> 2) + 4 * half + (r & 3) < length) top = fmaxf(top, s16[r]);
        reference = fmaxf(top, __shfl_xor(top, 32));      // finite: key 0 exists
        set_reference(reference);
        current = scores(0, 0);
    }
    // Software pipeline: the scores of block b + 1 are ISSUED before the vector work of
    // block b (maximum, exp2, split) - the bf16 matrix pipe takes them while the wave's
    // VALU instructions issue, which the fp32 MFMA does not allow (profiles/
    // r5_coexec.txt).  A stage is two blocks; the first block of stage s + 1 is issued
    // from the last block of stage s, behind the barrier that opens stage s + 1 - so a
    // wave may still read stage s while others are in s + 1: the ring holds four stages
    // and a stage is asked for TWO ahead (its buffer was last read in stage s - 2).
#pragma unroll 1
    for (int stage = 0; stage < stages; ++stage) {
        const int buffer = stage & (kSplitRing - 1);
        const int following = (stage + 1) & (kSplitRing - 1);
        const bool more = stage + 1 < stages;
        request(stage + 2);
        if (!working) {
            if (more) next_stage();
            continue;
        }
        const int key_base = stage * STAGE;
        const int keys = min(STAGE, length - key_base);
        {
            SPLIT_STAMP(0);
            u32x4 av[2][PV], tail[PV];
            values(buffer, 0, av, tail);
            f32x16 next = current;
            if (keys > 32) {
                next = scores(buffer, 32);
            } else if (more) {
                next_stage();
                next = scores(following, 0);
            }
            attend(current, next, av, tail, key_base, keys < 32);
            current = next;
        }
        if (keys > 32) {
            SPLIT_STAMP(0);
            u32x4 av[2][PV], tail[PV];
            values(buffer, 32, av, tail);
            f32x16 next = current;
            if (more) {
                next_stage();
                next = scores(following, 0);
            }
            attend(current, next, av, tail, key_base + 32, keys < 64);
            current = next;
        }
    }
    SPLIT_STAMP_FINISH
    // (requests beyond the last stage are still in flight into LDS nobody reads:
    // they must land before the workgroup's LDS is given to another)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (!working) return;
    float* o_rows = out + static_cast<int64_t>(head * D) * ld + span.offset;
    const int query = q0 + column;
    // the denominator: output row D = row D - 32 of the tail tiles, register (D - 32) % 4
    // of lane group (D - 32) / 4: query 16 q + n's is in lane 16 ((D - 32) / 4) + n of tile q
    constexpr int kSumGroup = (D - 32) / 4, kSumRegister = (D - 32) % 4;
    const float total_low = __shfl(o_tail[0][kSumRegister], 16 * kSumGroup + (lane & 15));
    const float total_high = __shfl(o_tail[1][kSumRegister], 16 * kSumGroup + (lane & 15));
    // rows 0 .. 31: the lane's query is `column`
    if (query < queries) {
        const float total_p = column < 16 ? total_low : total_high;
        const float inverse = length > 0 ? 1.f / total_p : NAN;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = 8 * (r >> 2) + 4 * half + (r & 3);
            o_rows[static_cast<int64_t>(d) * ld + query] = o[r] * inverse;
        }
    }
    // rows 32 .. D - 1: lane l holds rows 32 + 4 (l / 16) + i of queries l % 16 and 16 + l % 16
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int mine = q0 + 16 * q + (lane & 15);
        if (mine >= queries) continue;
        const float inverse = length > 0 ? 1.f / (q == 0 ? total_low : total_high) : NAN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = 32 + 4 * (lane >> 4) + i;
            if (d < D) o_rows[static_cast<int64_t>(d) * ld + mine] = o_tail[q][i] * inverse;
        }
    }
}

}  // namespace emph

using namespace emph;

namespace {

// `pieces` of the C ABI -> (pieces of keys / queries, pieces of values / probabilities)
//   2   two and two: three products per term everywhere ("bf16x3"; the scores' error,
//       2^-17 of sum |q k|, sits in front of the exponential and grows with their range)
//   3   three and three: six products per term, fp32 grade ("bf16x6")
//   32  three for the scores, two behind the softmax: six products where the error is
//       amplified, three where it is not
template <typename F>
int with_pieces(int pieces, const char* what, F&& call) {
    switch (pieces) {
        case 2: return call(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
        case 3: return call(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});
        case 32: return call(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
        default:
            set_error("%s: pieces %d (2: two bf16 pieces per operand; 3: three; 32: three for "
                      "the scores, two for the values)", what, pieces);
            return EMPH_EINVAL;
    }
}

}  // namespace

extern "C" {

/* bytes of scratch emph_split_kv fills for a packed frame axis of `ld` columns
 * and `n_segments` segments */
int64_t emph_split_kv_bytes(int64_t ld, int32_t n_segments, int32_t channels, int32_t heads,
                            int32_t pieces) {
    if (heads <= 0 || channels / heads != 40) return -1;
    const int64_t slots = ld / kSplitStage + n_segments + 1;
    int64_t stage = -1;
    with_pieces(pieces, "emph_split_kv_bytes", [&](auto pk, auto pv) {
        stage = SplitImages<40, decltype(pk)::value, decltype(pv)::value>::kStageBytes;
        return EMPH_OK;
    });
    return stage < 0 ? -1 : slots * heads * stage;
}

int emph_split_kv(const float* qk, const float* v, int64_t ld, int32_t channels, int32_t heads,
                  const int32_t* tiles, int32_t n_tiles, int32_t tile_n, int32_t pieces,
                  void* images, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(qk && v && tiles && images, EMPH_EINVAL, "emph_split_kv: null pointer");
    EMPH_REQUIRE(tile_n == kSplitStage, EMPH_EINVAL,
                 "emph_split_kv: tile_n %d (a stage is %d keys)", tile_n, kSplitStage);
    EMPH_REQUIRE(heads > 0 && channels % heads == 0 && channels / heads == 40, EMPH_ERANGE,
                 "emph_split_kv: head dimension %d (built for 40 = 80 channels, 2 heads)",
                 heads > 0 ? channels / heads : 0);
    EMPH_REQUIRE((reinterpret_cast<uintptr_t>(images) & 15) == 0, EMPH_EINVAL,
                 "emph_split_kv: images must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return with_pieces(pieces, "emph_split_kv", [&](auto pk, auto pv) {
        EMPH_LAUNCH((split_kv_kernel<40, decltype(pk)::value, decltype(pv)::value>), dim3(n_tiles),
                    dim3(256), 0, s, qk, v, ld, channels, heads, tiles,
                    static_cast<unsigned char*>(images));
        return check_launch("emph_split_kv");
    });
}

int emph_attention_split(const float* qk, const void* images, float* out, int64_t ld,
                         int32_t channels, int32_t heads, const int32_t* tiles,
                         int32_t n_tiles, int32_t tile_n, const int32_t* key_counts,
                         int32_t pieces, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(qk && images && out && tiles, EMPH_EINVAL, "emph_attention_split: null pointer");
    EMPH_REQUIRE(tile_n == kSplitQueries, EMPH_EINVAL,
                 "emph_attention_split: tile_n %d (a workgroup owns %d queries)", tile_n,
                 kSplitQueries);
    EMPH_REQUIRE(heads > 0 && channels % heads == 0 && channels / heads == 40, EMPH_ERANGE,
                 "emph_attention_split: head dimension %d (built for 40 = 80 channels, 2 heads)",
                 heads > 0 ? channels / heads : 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(n_tiles, heads);
    const unsigned char* bytes = static_cast<const unsigned char*>(images);
    return with_pieces(pieces, "emph_attention_split", [&](auto pk, auto pv) {
        constexpr int PK = decltype(pk)::value, PV = decltype(pv)::value;
        auto kernel = attention_split_kernel<40, PK, PV>;
        const size_t lds = kSplitRing * SplitImages<40, PK, PV>::kStageBytes;
        static LdsReservation reserved;
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel), lds,
                                     "emph_attention_split"))
            return status;
        EMPH_LAUNCH(kernel, grid, dim3(64 * kSplitWaves), lds, s, qk, bytes, out, ld, channels,
                    tiles, key_counts SPLIT_STAMP_PASS);
        return check_launch("emph_attention_split");
    });
}

}  // extern "C"
