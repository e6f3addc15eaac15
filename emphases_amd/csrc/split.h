// Shared by attention_split.hip and block_split.hip: fp32 values as bf16 pieces, the
// bf16 matrix instruction over them, and the layout of a stage of split keys / values.
#pragma once

#include "common.h"

namespace emph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// two floats -> PIECES dwords of two bf16 each (element 0 in the low half)
template <int PIECES>
__device__ __forceinline__ void split_pair(float a, float b, uint32_t (&out)[PIECES]) {
    if (PIECES == 2) {
        // round to nearest, twice: |x - x0 - x1| <= 2^-17 |x|
        const bf16x2 high = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
        uint32_t bits = __builtin_bit_cast(uint32_t, high);
        // (opaque from here on: seeing the two casts behind `bits`, hipcc converts the
        // low element a second time to shift it.  The conversion itself stays a real
        // instruction - the hazard recogniser does not look inside inline asm, and a
        // v_cvt_pk written in asm in front of an MFMA gave wrong products)
        asm("" : "+v"(bits));
        const float ra = a - __uint_as_float(bits << 16);
        const float rb = b - __uint_as_float(bits & 0xffff0000u);
        const bf16x2 low = {static_cast<__bf16>(ra), static_cast<__bf16>(rb)};
        out[0] = bits;
        out[1] = __builtin_bit_cast(uint32_t, low);
    } else {
        // truncation, three times: exact (8 + 8 + 8 bits)
        float ra = a, rb = b;
#pragma unroll
        for (int piece = 0; piece < PIECES; ++piece) {
            const uint32_t ua = __float_as_uint(ra), ub = __float_as_uint(rb);
            out[piece] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
            if (piece + 1 < PIECES) {
                ra -= __uint_as_float(ua & 0xffff0000u);
                rb -= __uint_as_float(ub & 0xffff0000u);
            }
        }
    }
}

// eight floats -> PIECES fragments of eight bf16
template <int PIECES>
__device__ __forceinline__ void split_eight(const float (&x)[8], u32x4 (&out)[PIECES]) {
#pragma unroll
    for (int pair = 0; pair < 4; ++pair) {
        uint32_t parts[PIECES];
        split_pair<PIECES>(x[2 * pair], x[2 * pair + 1], parts);
#pragma unroll
        for (int piece = 0; piece < PIECES; ++piece) out[piece][pair] = parts[piece];
    }
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// products of pieces (i of the A operand, j of the B operand) that are kept
template <int PIECES>
__device__ __forceinline__ f32x16 split_product(const u32x4 (&a)[PIECES], const u32x4 (&b)[PIECES],
                                                f32x16 c) {
    // the smallest products first
#pragma unroll
    for (int order = PIECES - 1; order >= 0; --order)
#pragma unroll
        for (int i = 0; i <= order; ++i) c = mfma_bf16(a[i], b[order - i], c);
    return c;
}

constexpr int kSplitStage = 64;                       // keys per stage

// One stage (64 keys of one segment and head) of split keys and values as it lies in
// LDS - and, written once per layer by split_kv_kernel, in global memory, so that
// staging is a copy (LDS-DMA), not a conversion per query tile:
//     K part  [d / 8][key][8 d]  bf16, 16 bytes per (octet, key): the A fragment of
//             a k-step of S^T = K Q^T is one conflict-free 16-byte read.  d = D holds
//             ONES in piece 0 (zeros up to the next multiple of 16): with -reference
//             in that slot of Q^T the scores come out of the matrix pipe shifted.
//     V part  [key / 8][row][8 keys]  bf16 with the keys of each 16 permuted
//             (position 8 h + 4 a + i holds key 8 a + 4 h + i), rows 0 .. D - 1 = d,
//             row D = ONES (piece 0: accumulates the softmax denominator), row
//             D + 1 = zeros (what the rows up to 63 of the second m-tile read).
// Keys beyond the segment are ZEROS (their probabilities are zero; padding of the
// packed axis may hold NaN).  Images are indexed by SLOT = offset / 64 + segment +
// stage: distinct for all stages of all segments of a packed axis.
// PK pieces of the keys (and queries), PV of the values (and probabilities): a stage is
// [K piece 0 .. PK - 1][V piece 0 .. PV - 1].
template <int D, int PK, int PV>
struct SplitImages {
    static constexpr int kOctets = (D + 15) / 16 * 2;               // k-steps of 16, two octets each
    static constexpr int kKeyBytes = kOctets * kSplitStage * 16;
    static constexpr int kRows = D + 2;
    static constexpr int kValueBytes = (kSplitStage / 8) * kRows * 16;
    static constexpr int kStageBytes = PK * kKeyBytes + PV * kValueBytes;   // a buffer of LDS
    static constexpr int key_piece(int piece) { return piece * kKeyBytes; }
    static constexpr int value_piece(int piece) { return PK * kKeyBytes + piece * kValueBytes; }
};

}  // namespace emph
