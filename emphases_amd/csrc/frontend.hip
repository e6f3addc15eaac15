// Front-end of the prominence path: framed log-mel (and the optional
// A-weighted loudness row) straight from packed utterance audio.
//
// Replaces, per chunk (paths relative to the reference repository):
//   F.pad(audio, (432, 432)) + word-boundary slice    emphases/core.py:357-401
//   reflect pad 432                                   data/preprocess/mels.py:31-36
//   torch.stft(1024, hop 160, periodic Hann)          mels.py:39-48
//   sqrt(re^2 + im^2 + 1e-6)                          mels.py:51
//   mel basis matmul, log(clamp(., 1e-5))             mels.py:94-109
//   librosa loudness (always on CPU in the reference) data/preprocess/loudness.py:59-107
//
// Design (gfx950): a wave owns tiles of 8 consecutive frames (kWaveFrames) and
// transforms them one at a time (kPair), three waves per SIMD covering each
// other's LDS round trips.  A frame's 1024 samples come straight from the packed
// audio as eight coalesced 8-byte loads per lane (float32; 4-byte loads for
// 16-bit PCM) - consecutive frames overlap by 864 samples, so all but the first
// touch of a sample is an L1/L2 hit and HBM sees each sample about once; the
// zero-pad / slice / reflect index arithmetic runs only for the frames at a
// chunk's ends.  The 1024-point real FFT is a 512-point complex FFT held 8
// points per lane - three radix-8 passes in registers with two wave-private LDS
// transposes - followed by the real-FFT split on bin pairs (only the high half
// of the spectrum crosses lanes), magnitudes, the 1001-non-zero sparse mel
// projection and the log.  The wave's [80 x 8] result tile is staged in LDS and
// written as 32-byte row segments.  There is no workgroup barrier anywhere, and
// nothing of the 513 x F complex spectrogram ever reaches HBM.
#include <math.h>
#include <stdarg.h>

#include "common.h"

namespace emph {

namespace {
thread_local char g_error[512] = "";
}

void set_error(const char* format, ...) {
    va_list args;
    va_start(args, format);
    vsnprintf(g_error, sizeof(g_error), format, args);
    va_end(args);
}

// Exchange between lanes of one wave through LDS.  A wave's DS instructions are
// executed by the LDS unit in program order, so a read issued after a write
// sees it: only the COMPILER has to be kept from reordering them; the wait for
// the data is the s_waitcnt hipcc places before the first use of what was read.
__device__ __forceinline__ void frontend_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// End of one frame's section of a stage: hipcc's scheduler may not move the
// other frame's arithmetic across it (left alone it hoists both frames'
// arithmetic in front of both exchanges, and each exchange's latency is then
// exposed instead of running under the other frame's work).
__device__ __forceinline__ void frontend_section() { __builtin_amdgcn_sched_barrier(0); }

constexpr float kLn2 = 0.69314718055994530942f;
constexpr int kWaveFrames = 8;     // consecutive frames per wave = tile-table block
constexpr int kPair = 1;           // frames a wave transforms at a time
constexpr int kWavesPerSimd = 3;   // 168 VGPRs
constexpr int kTileStride = kWaveFrames + 1;
constexpr int kExRow = 72;                                     // complex per exchange row
// Second exchange, C[r][t][p0] at r * kExRowB + t * kExStepB + p0: written by
// lane (r, p0) for each t and read by lane (r, t) for each p0, 8 bytes each.
// 2 * 88 = 48 (mod 64) puts the four r of a 32-lane group 16 banks apart and the
// odd step 9 spreads t (or p0) over those 16: both directions are conflict free
// (with rows of 8 the reads were 4-way conflicts: SQ_LDS_BANK_CONFLICT was 42 %
// of the LDS-active cycles of this kernel).
constexpr int kExRowB = 88;
constexpr int kExStepB = 9;
constexpr int kExFloats = 2 * 8 * kExRowB;                     // floats per frame slot
// natural-order spectrum: 4 complex of padding after every 32 so that the
// stride-8 writes of the last pass (lanes t and t + 4 used to collide) and the
// contiguous reads of the real-FFT split are both conflict free
__device__ __forceinline__ int spectrum_slot(int k) { return k + 4 * (k >> 5); }
constexpr int kHighBase = 256 + 4 * (256 >> 5);   // slot of bin 256: only 256..511 are stored
constexpr int kMagFloats = 560;      // 513 magnitudes + zero tail for the runs, laid over
                                     // the frame's exchange buffer
// floats of LDS per wave: exchange buffers, output tile, loudness tile
constexpr int kWaveFloats = kPair * kExFloats + kMels * kTileStride + kWaveFrames;
// Filterbank runs are read from LDS as 16-byte pieces from a start rounded down
// to a multiple of four bins: rows 0..63 (at most 20 bins + 3 of alignment) as
// six pieces by one lane each, rows 64..79 (at most 40 + 3) as three pieces by
// each of four lanes.  (Dword reads from per-lane starts cost twice the LDS
// cycles per byte and collide on banks: 30 % of this kernel's LDS-active
// cycles were still bank conflicts after the exchange layouts were fixed.)
constexpr int kRunA = 24;
constexpr int kRunB = 12;

// Table layout (floats)
constexpr int kTabWindow = 0;                  // [1024]
constexpr int kTabTw1 = 1024;                  // [8][64] complex  W512^(p r)
constexpr int kTabTw2 = kTabTw1 + 2 * 512;     // [8][8] complex   W64^(p0 t)
constexpr int kTabTw3 = kTabTw2 + 2 * 64;      // [513] complex    W1024^k
constexpr int kTabSize = kTabTw3 + 2 * 514;

// A complex number is one 64-bit register pair, and every butterfly below is a
// packed instruction on it: hipcc's own pairing of the scalar form mixed halves
// of different values and spent 70 v_mov per frame splicing them (this kernel
// is bound by VALU issue).  Rotations by -i ride on the operand-select / negate
// modifiers of the add that consumes them.
typedef float cf __attribute__((ext_vector_type(2)));

// Eight 8-byte LDS reads at base + k * STRIDE bytes, as EIGHT ds_read_b64:
// hipcc merges such reads pairwise into ds_read2_b64, which the LDS serves at
// half the rate (8 cycles per 1 KiB against 2 x 2; MI355X_MICROARCH.md, LDS
// table) - and this kernel is bound by LDS cycles plus vector issue.
template <int STRIDE>
__device__ __forceinline__ void lds_read8(cf (&v)[8], const cf* base) {
    const uint32_t address = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (const __attribute__((address_space(3))) cf*)base));
    asm volatile(
        "ds_read_b64 %0, %8\n\t"
        "ds_read_b64 %1, %8 offset:%9\n\t"
        "ds_read_b64 %2, %8 offset:%10\n\t"
        "ds_read_b64 %3, %8 offset:%11\n\t"
        "ds_read_b64 %4, %8 offset:%12\n\t"
        "ds_read_b64 %5, %8 offset:%13\n\t"
        "ds_read_b64 %6, %8 offset:%14\n\t"
        "ds_read_b64 %7, %8 offset:%15\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]),
          "=&v"(v[6]), "=&v"(v[7])
        : "v"(address), "n"(STRIDE), "n"(2 * STRIDE), "n"(3 * STRIDE), "n"(4 * STRIDE),
          "n"(5 * STRIDE), "n"(6 * STRIDE), "n"(7 * STRIDE)
        : "memory");
}


// a b: t = (a.y b.y, a.y b.x), then (a.x b.x - t.x, a.x b.y + t.y)
__device__ __forceinline__ cf cmul(cf a, cf b) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : "=v"(r)
        : "v"(a), "v"(b), "v"(t));
    return r;
}
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ cf add_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"
        : "=v"(r)
        : "v"(a), "v"(b));
    return r;
}
// a - (-i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cf sub_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        : "=v"(r)
        : "v"(a), "v"(b));
    return r;
}
// a + conj(b) and a - conj(b)
__device__ __forceinline__ cf add_conj(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf sub_conj(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// c + k (-i a) = (c.x + k a.y, c.y - k a.x) and c - k (-i a), k a real scalar
// given as the pair kk = (k, -k): one packed fma each (the rotation rides on
// the operand selects, the sign of the second form on the negate modifiers)
__device__ __forceinline__ cf fma_mi(cf a, cf kk, cf c) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]"
        : "=v"(r)
        : "v"(a), "v"(kk), "v"(c));
    return r;
}
__device__ __forceinline__ cf fms_mi(cf a, cf kk, cf c) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0] "
        "neg_hi:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(kk), "v"(c));
    return r;
}

// In-place forward DFT of 8 points, natural order in and out:
// v[r] <- sum_q v[q] exp(-2 pi i q r / 8).  26 packed instructions: the two
// multiplications by W8 = (1 - i) / sqrt(2) and W8^3 are deferred into the fmas
// of the last stage.
__device__ __forceinline__ void dft8(cf v[8]) {
    constexpr float kH = 0.70710678118654752440f;
    const cf a0 = v[0] + v[4], b0 = v[0] - v[4];
    const cf a1 = v[1] + v[5], d1 = v[1] - v[5];
    const cf a2 = v[2] + v[6], b2 = v[2] - v[6];    // (b2 is used as -i b2 below)
    const cf a3 = v[3] + v[7], d3 = v[3] - v[7];
    // b1 = W8 d1 = kH (d1.x + d1.y, d1.y - d1.x) = kH p1; b3 = W8^3 d3 =
    // -kH (d3.x - d3.y, d3.x + d3.y) = -kH p3
    const cf p1 = add_mi(d1, d1);
    const cf p3 = sub_mi(d3, d3);
    // DFT-4 of a -> even outputs
    cf e0 = a0 + a2, e1 = a0 - a2;
    cf o0 = a1 + a3, t = a1 - a3;
    v[0] = e0 + o0;
    v[4] = e0 - o0;
    v[2] = add_mi(e1, t);
    v[6] = sub_mi(e1, t);
    // DFT-4 of (b0, b1, -i b2, b3) -> odd outputs, with b1 + b3 = kH (p1 - p3)
    // and b1 - b3 = kH (p1 + p3)
    e0 = add_mi(b0, b2);
    e1 = sub_mi(b0, b2);
    o0 = p1 - p3;
    t = p1 + p3;
    const cf scale = {kH, kH}, rotate = {kH, -kH};
    v[1] = o0 * scale + e0;
    v[5] = e0 - o0 * scale;
    v[3] = fma_mi(t, rotate, e1);
    v[7] = fms_mi(t, rotate, e1);
}

// Sum over the four lanes of a quad on the DPP path (quad_perm [1,0,3,2] and
// [2,3,0,1]): __shfl_xor goes through ds_bpermute, an LDS round trip each.
__device__ __forceinline__ float quad_sum(float x) {
    x += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(x), 0xB1, 0xF, 0xF, true));
    x += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(x), 0x4E, 0xF, 0xF, true));
    return x;
}

__device__ __forceinline__ float wave_max(float value) {
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1)
        value = fmaxf(value, __shfl_xor(value, offset));
    return value;
}

// One frame's 1024 samples as the 512 complex points z[n] = y[2n] + i y[2n+1]
// the lane transforms: s[q] = z[p + 64 q].  Interior frames (all but the first
// and last few of a chunk) are eight coalesced 8-byte loads straight from the
// packed audio (4-byte loads for 16-bit PCM); frames that touch a chunk's or
// the utterance's ends apply the zero-pad / slice / reflect index arithmetic of
// emphases/core.py:357-401 and mels.py:31-36 per sample.  Consecutive frames of
// a wave overlap by 864 of 1024 samples, so most of these loads hit in L1/L2.
typedef float cf_u __attribute__((ext_vector_type(2), aligned(4)));

// Where a chunk's samples live, in 32-bit chunk positions (a chunk is far
// shorter than 2^31 samples): position r of the chunk is audio sample
// `origin[r]` when r_lo <= r < r_hi and zero otherwise (the 432 zeros that
// emphases/core.py:357-358 pads the utterance with, or beyond the audio).
struct Chunk {
    const void* origin;   // address of chunk position 0 (may lie before the buffer)
    int length;           // samples in the chunk
    int r_lo, r_hi;       // chunk positions backed by audio
};

template <bool PCM>
__device__ __forceinline__ Chunk open_chunk(const void* audio, int64_t audio_off,
                                            int64_t audio_len, int64_t start, int64_t length) {
    Chunk chunk;
    const int64_t shift = audio_off + start - kPad;     // audio index of position 0
    chunk.origin = PCM ? static_cast<const void*>(static_cast<const int16_t*>(audio) + shift)
                       : static_cast<const void*>(static_cast<const float*>(audio) + shift);
    chunk.length = static_cast<int>(length);
    const int64_t lo = kPad - start, hi = audio_len + kPad - start;
    chunk.r_lo = static_cast<int>(lo < 0 ? 0 : (lo > length ? length : lo));
    chunk.r_hi = static_cast<int>(hi < 0 ? 0 : (hi > length ? length : hi));
    return chunk;
}

// Samples of an INTERIOR frame (its 1024 samples all lie inside the audio that
// backs the chunk): eight coalesced 8-byte loads (4-byte for 16-bit PCM) from
// `source` = address of the frame's first sample.
template <bool PCM>
__device__ __forceinline__ void load_interior(const void* source, int p, cf (&s)[8]) {
    if (PCM) {
        const int16_t* samples = static_cast<const int16_t*>(source);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            typedef uint32_t u32_u __attribute__((aligned(2)));
            const uint32_t two = *reinterpret_cast<const u32_u*>(samples + 2 * (p + 64 * q));
            s[q] = {static_cast<float>(static_cast<int16_t>(two & 0xffffu)),
                    static_cast<float>(static_cast<int16_t>(two >> 16))};
        }
    } else {
        const float* samples = static_cast<const float*>(source);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            s[q] = *reinterpret_cast<const cf_u*>(samples + 2 * (p + 64 * q));
    }
}

// Is frame `frame` of the chunk interior, and where does it start?
__device__ __forceinline__ bool frame_is_interior(const Chunk& chunk, int frame) {
    const int first = frame * kHop - kPad;              // chunk position of sample 0
    return first >= chunk.r_lo && first + kFft <= chunk.r_hi;
}

// Address the NEXT frame is requested from, one frame ahead and without a
// branch: an interior frame's own samples; for a frame at a chunk's end (whose
// samples `load_edge` fetches when its turn comes) any 1024 readable samples -
// the nearest interior frame's, or the constant table when the chunk has none
// (the table holds more than 1024 floats).  The prefetch is a single
// unconditional definition of the sample registers: with the interior / edge
// choice inside it, hipcc merged the two paths through 32 v_mov per frame.
template <bool PCM>
__device__ __forceinline__ const void* frame_source(const Chunk& chunk, int frame,
                                                    const float* table) {
    int first = frame * kHop - kPad;
    const int highest = chunk.r_hi - kFft;
    if (highest < chunk.r_lo) return table;
    first = min(max(first, chunk.r_lo), highest);
    return PCM ? static_cast<const void*>(static_cast<const int16_t*>(chunk.origin) + first)
               : static_cast<const void*>(static_cast<const float*>(chunk.origin) + first);
}

// A frame at a chunk's (or the utterance's) end: reflect / zero per sample
// (emphases/core.py:357-401 and mels.py:31-36).
template <bool PCM>
__device__ __forceinline__ void load_edge(const Chunk& chunk, int frame, int p, cf (&s)[8]) {
    const int first = frame * kHop - kPad;
    const int safe = chunk.r_hi > chunk.r_lo ? chunk.r_lo : 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float pair[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int r = first + 2 * (p + 64 * q) + h;          // position in the chunk
            if (r < 0) r = -r;                             // reflect (no edge repeat)
            if (r >= chunk.length) r = 2 * (chunk.length - 1) - r;
            const bool live = r >= chunk.r_lo && r < chunk.r_hi;
            const int at = live ? r : safe;
            float value = 0.f;
            if (chunk.r_hi > chunk.r_lo)
                value = PCM ? static_cast<float>(static_cast<const int16_t*>(chunk.origin)[at])
                            : static_cast<const float*>(chunk.origin)[at];
            pair[h] = live ? value : 0.f;
        }
        s[q] = {pair[0], pair[1]};
    }
}

// MODE 0: mel rows.  MODE 1: per-chunk peak power only.  MODE 2: mel rows and
// loudness row.  MODE 3: loudness row only.  PCM: audio is int16 (x / 32768).
//
// A workgroup is four INDEPENDENT waves (no workgroup barrier anywhere): each
// owns tiles of kWaveFrames consecutive frames and transforms them kPair at a
// time (kPair = 2 lets one frame's arithmetic cover the other's LDS round trips
// at 253 VGPRs; kPair = 1 at 164 VGPRs puts three waves on a SIMD instead and is
// what ships: 66.5 vs 68.7 us).
template <int MODE, bool PCM>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(kWavesPerSimd, kWavesPerSimd))) void frontend_kernel(
    const void* __restrict__ audio, const int64_t* __restrict__ seg,
    const int32_t* __restrict__ tiles, const float* __restrict__ table,
    const int32_t* __restrict__ mel_start, const int32_t* __restrict__ mel_count,
    const int32_t* __restrict__ mel_offset, const float* __restrict__ mel_values,
    int mel_nnz, float* __restrict__ out, int64_t ld, int mel_row, int loud_row,
    float* __restrict__ seg_peak, const float* __restrict__ a_weights,
    int normalize, int n_tiles) {
    constexpr bool kMel = MODE == 0 || MODE == 2;
    constexpr bool kLoud = MODE == 2 || MODE == 3;
    constexpr bool kPeak = MODE == 1;
    // what `power` carries on top of |X|^2: the magnitude's 1e-6 when nothing
    // but the mel rows reads it
    constexpr float kPowerFloor = MODE == 0 ? 1e-6f : 0.f;

    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* mine = lds + wave * kWaveFloats;
    float* exchange = mine;                                  // [kPair][kExFloats]
    float* tile = exchange + kPair * kExFloats;              // [80][kTileStride]
    float* loud_tile = tile + kMels * kTileStride;           // [kWaveFrames]
    float* weights = lds + 4 * kWaveFloats;                  // [513+] (loudness only)

    if (kLoud) {
        for (int index = tid; index < kBins; index += 256) weights[index] = a_weights[index];
        __syncthreads();
    }

    // ---- per-lane constants
    const int p = lane;              // pass-1 position
    const int r1 = lane >> 3;        // pass-2/3 residue r
    const int p0 = lane & 7;         // pass-2 position / pass-3 output t
    float window[16];
    cf tw1[8], tw2[8], tw3[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        // 16-bit PCM: the 1/32768 rides on the window (an exact power of two)
        const float unit = PCM ? 1.f / 32768.f : 1.f;
        window[2 * q] = unit * table[kTabWindow + 2 * (p + 64 * q)];
        window[2 * q + 1] = unit * table[kTabWindow + 2 * (p + 64 * q) + 1];
        tw1[q] = {table[kTabTw1 + 2 * (q * 64 + p)],
                  table[kTabTw1 + 2 * (q * 64 + p) + 1]};
        tw2[q] = {table[kTabTw2 + 2 * (q * 8 + p0)],
                  table[kTabTw2 + 2 * (q * 8 + p0) + 1]};
    }
    // -i W1024^k for the lane's four low bins k = r + 8 t + 64 u: the real-FFT
    // split multiplies it with Z[k] - conj(Z[512 - k])
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = r1 + 8 * p0 + 64 * u;
        tw3[u] = {table[kTabTw3 + 2 * k + 1], -table[kTabTw3 + 2 * k]};
    }
    // spectrum bin of power[j]: the low bins, their partners 512 - k, and bin 256
    // (lane 0 only)
    auto bin_of = [&](int j) {
        const int k = r1 + 8 * p0 + 64 * (j & 3);
        return j == 8 ? 256 : (j < 4 ? k : 512 - k);
    };

    // Sparse mel projection: every filterbank row is a contiguous run of bins.
    // Rows 0..63 (runs of at most kRunA bins) get one lane each; rows 64..79
    // (runs of at most 4*kRunB) get four lanes each.  The run weights live in
    // registers for the kernel's whole life, zero-padded to a fixed length so
    // that the per-frame loops are fully unrolled with every LDS read independent.
    float weight_a[kRunA], weight_b[kRunB];
    int start_a = 0, start_b = 0;
    if (kMel) {
        const int first_a = mel_start[lane];
        const int count_a = mel_count[lane];
        const int offset_a = mel_offset[lane];
        start_a = first_a & ~3;
#pragma unroll
        for (int j = 0; j < kRunA; ++j) {
            const int index = start_a + j - first_a;        // position in the run
            weight_a[j] = (index >= 0 && index < count_a)
                              ? mel_values[min(max(offset_a + index, 0), mel_nnz - 1)]
                              : 0.f;
        }
        const int row_b = 64 + (lane >> 2);
        const int first_b = mel_start[row_b];
        const int count_b = mel_count[row_b];
        const int offset_b = mel_offset[row_b];
        start_b = (first_b & ~3) + kRunB * (lane & 3);
#pragma unroll
        for (int j = 0; j < kRunB; ++j) {
            const int index = start_b + j - first_b;
            weight_b[j] = (index >= 0 && index < count_b)
                              ? mel_values[min(max(offset_b + index, 0), mel_nnz - 1)]
                              : 0.f;
        }
    }

    // ---- persistent loop over this WAVE's tiles of kWaveFrames frames
    float peak = 0.f;
    int peak_segment = -1;
    const int wave_id = blockIdx.x * 4 + wave;
    const int wave_stride = gridDim.x * 4;
    // (handing tiles out through an atomic counter was measured: 8 000 requests to
    // one address serialise in L2 and the launch takes 2.7x as long)
    for (int block = wave_id; block < n_tiles; block += wave_stride) {
    const Tile span = load_tile(tiles, block);
    const int segment = __builtin_amdgcn_readfirstlane(span.segment);
    const int frame0 = __builtin_amdgcn_readfirstlane(span.first);
    const int64_t* row = seg + static_cast<int64_t>(segment) * EMPH_SEG_FIELDS;
    const int64_t audio_off = row[EMPH_SEG_AUDIO_OFF];
    const int64_t audio_len = row[EMPH_SEG_AUDIO_LEN];
    const int64_t start = row[EMPH_SEG_START];
    const int64_t length = row[EMPH_SEG_LENGTH];
    const int64_t frame_off = row[EMPH_SEG_FRAME_OFF];
    const int frames = static_cast<int>(row[EMPH_SEG_FRAMES]);
    const int valid = min(kWaveFrames, frames - frame0);
    const Chunk chunk = open_chunk<PCM>(audio, audio_off, audio_len, start, length);
    if (kPeak && segment != peak_segment) {
        // a new chunk: publish the running peak of the previous one
        if (peak_segment >= 0) {
            const float total = wave_max(peak);
            if (lane == 0)
                atomicMax(reinterpret_cast<unsigned int*>(seg_peak + peak_segment),
                          __float_as_uint(total));
        }
        peak = 0.f;
        peak_segment = segment;
    }

    float floor_db = 0.f;
    if (kLoud) {
        // librosa.amplitude_to_db: max(D) - top_db with D = 10 log10(max(amin^2, S^2))
        const float top = 10.f * log10f(fmaxf(1e-10f, seg_peak[segment]));
        floor_db = top - 80.f;
    }

    // The samples of a frame are requested one frame ahead (a global load takes
    // 1-2 us under load: far longer than the transform's own LDS round trips
    // could cover), the first frame of a tile in front of the loop.
    // (`transform` is called with the same set as `cur` and `next`: the frame is
    // windowed out of it before the next frame's samples are requested into it)
    cf raw[kPair][8];
#pragma unroll
    for (int f = 0; f < kPair; ++f)
        // an odd tail repeats the last frame (its result is written twice)
        load_interior<PCM>(frame_source<PCM>(chunk, frame0 + min(f, valid - 1), table), p,
                           raw[f]);
    auto transform = [&](cf (&cur)[kPair][8], cf (&next)[kPair][8], int local) {
        cf v[kPair][8];
        cf* ex[kPair];
#pragma unroll
        for (int f = 0; f < kPair; ++f) {
            const int frame = frame0 + min(local + f, valid - 1);
            if (!frame_is_interior(chunk, frame))          // wave-uniform, rare
                load_edge<PCM>(chunk, frame, p, cur[f]);
            ex[f] = reinterpret_cast<cf*>(exchange + f * kExFloats);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[f][q] = cur[f][q] * cf{window[2 * q], window[2 * q + 1]};
        }
#pragma unroll
        for (int f = 0; f < kPair; ++f)
            load_interior<PCM>(
                frame_source<PCM>(chunk, frame0 + min(local + kPair + f, valid - 1), table), p,
                next[f]);
        // Every exchange below is written frame by frame as
        //     compute(f); write(f); fence; read(f)
        // so that a frame's LDS round trip runs under the OTHER frame's
        // arithmetic (a wave's LDS operations execute in program order: the
        // fence only keeps the compiler from reordering them).
        //
        // pass 1: radix-8 over q of z[p + 64 q], z[n] = y[2n] + i y[2n+1];
        // pass 2: lane (r, p0) takes A[p0 + 8 p1][r], radix-8 over p1
#pragma unroll
        for (int f = 0; f < kPair; ++f) {
            dft8(v[f]);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r) v[f][r] = cmul(v[f][r], tw1[r]);
                ex[f][r * kExRow + p] = v[f][r];
            }
            frontend_fence();
            lds_read8<64>(v[f], ex[f] + r1 * kExRow + p0);        // p1 = 0 .. 7
            frontend_fence();
            frontend_section();
        }
        // pass 3: lane (r, t) takes C[r][p0][t], radix-8 over p0
#pragma unroll
        for (int f = 0; f < kPair; ++f) {
            dft8(v[f]);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t) v[f][t] = cmul(v[f][t], tw2[t]);
                ex[f][r1 * kExRowB + t * kExStepB + p0] = v[f][t];     // C[r][t][p0]
            }
            frontend_fence();
            lds_read8<8>(v[f], ex[f] + r1 * kExRowB + p0 * kExStepB);     // q = 0 .. 7
            frontend_fence();
            frontend_section();
        }
        // Z[r + 8 t + 64 u] = v[u].  Real-FFT split, two bins per pair: with
        // E = Z[k] + conj(Z[512-k]) and O' = (-i W^k)(Z[k] - conj(Z[512-k])),
        //     X[k] = E + O'        conj(X[512-k]) = E - O'
        // (the 1/2 rides on the window table).  A lane owns the pairs of its
        // four low bins k = r + 8 t + 64 u, u < 4, so only the HIGH half of the
        // spectrum (u >= 4) crosses lanes: four 8-byte LDS writes and four reads
        // per lane instead of eight and sixteen for a natural-order round trip.
        cf zm[kPair][4];
#pragma unroll
        for (int f = 0; f < kPair; ++f) {
            dft8(v[f]);
#pragma unroll
            for (int u = 4; u < 8; ++u)
                ex[f][spectrum_slot(r1 + 8 * p0 + 64 * u) - kHighBase] = v[f][u];
            frontend_fence();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = r1 + 8 * p0 + 64 * u;
                // k = 0 pairs with itself: bins 0 and 512 come out of the same formulas
                zm[f][u] = ex[f][spectrum_slot((512 - k) & 511 ? (512 - k) & 511 : 256) -
                                 kHighBase];
            }
            frontend_fence();
            frontend_section();
        }
        float power[kPair][9];
#pragma unroll
        for (int f = 0; f < kPair; ++f) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cf z = zm[f][u];
                if (u == 0) z = lane == 0 ? v[f][0] : z;
                const cf e = add_conj(v[f][u], z);
                const cf o = cmul(tw3[u], sub_conj(v[f][u], z));
                // (|X[k]|^2, |X[512 - k]|^2) with X[k] = e + o, conj(X[512 - k]) =
                // e - o, as real parts squared plus imaginary parts squared of the
                // two bins side by side: four packed instructions for two powers.
                // Where only the mel rows are wanted, the 1e-6 of the magnitude
                // (mels.py:51) is the addend of the first of them.
                cf real, imag;
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]"
                    : "=v"(real) : "v"(e), "v"(o));
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]"
                    : "=v"(imag) : "v"(e), "v"(o));
                const cf floor = {kPowerFloor, kPowerFloor};
                const cf both = imag * imag + (real * real + floor);
                power[f][u] = both.x;
                power[f][4 + u] = both.y;
            }
            // k = 256 pairs with itself: X[256] = conj(Z[256]) (lane 0 holds it, u = 4);
            // the window's 1/2 has to be undone there
            power[f][8] = 4.f * (v[f][4].x * v[f][4].x + v[f][4].y * v[f][4].y) + kPowerFloor;
        }
        if (kPeak) {
#pragma unroll
            for (int f = 0; f < kPair; ++f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) peak = fmaxf(peak, power[f][j]);
                if (lane == 0) peak = fmaxf(peak, power[f][8]);
            }
            return;
        }

        if (kLoud) {
            // 10 log10(max(1e-10, |X|^2)) floored at peak - 80, + A-weight,
            // clamped at MIN_DB = -100, mean over the 513 bins
#pragma unroll
            for (int f = 0; f < kPair; ++f) {
                double total = 0.;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    if (j == 8 && lane != 0) break;
                    const int k = bin_of(j);
                    float db = 10.f * log10f(fmaxf(1e-10f, power[f][j]));
                    db = fmaxf(db, floor_db) + weights[k];
                    total += static_cast<double>(fmaxf(db, -100.f));
                }
#pragma unroll
                for (int offset = 32; offset > 0; offset >>= 1)
                    total += __shfl_xor(total, offset);
                if (lane == 0) {
                    float value = static_cast<float>(total / 513.);
                    if (normalize) value = (value + 100.f) / 100.f;
                    loud_tile[min(local + f, valid - 1)] = value;
                }
            }
        }

        if (kMel) {
            // magnitudes over the frame's own exchange buffer (its last reads are
            // behind us: a wave's LDS operations execute in program order)
#pragma unroll
            for (int f = 0; f < kPair; ++f) {
                float* mag = exchange + f * kExFloats;
                // v_sqrt_f32 (1 ulp; the argument is >= 1e-6, never denormal): the
                // correctly rounded sqrtf is ten more instructions per bin
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    mag[bin_of(j)] = __builtin_amdgcn_sqrtf(power[f][j] + (1e-6f - kPowerFloor));
                if (lane == 0)
                    mag[256] = __builtin_amdgcn_sqrtf(power[f][8] + (1e-6f - kPowerFloor));
                // zero tail: the fixed-length runs read past bin 512 (weight 0, but
                // the buffer holds spectrum bits there: 0 * NaN would poison the sum)
                if (lane < kMagFloats - kBins) mag[kBins + lane] = 0.f;
            }
            frontend_fence();
#pragma unroll
            for (int f = 0; f < kPair; ++f) {
                const float* mag = exchange + f * kExFloats;
                const int column = min(local + f, valid - 1);
                float acc = 0.f;
#pragma unroll
                for (int piece = 0; piece < kRunA / 4; ++piece) {
                    const float4 four =
                        *reinterpret_cast<const float4*>(mag + start_a + 4 * piece);
                    acc = fmaf(weight_a[4 * piece], four.x, acc);
                    acc = fmaf(weight_a[4 * piece + 1], four.y, acc);
                    acc = fmaf(weight_a[4 * piece + 2], four.z, acc);
                    acc = fmaf(weight_a[4 * piece + 3], four.w, acc);
                }
                // natural log on v_log_f32 (log2, 1 ulp; the argument is >= 1e-5);
                // (x + 10) / 10 as one fma: both within 1e-7 of the exact forms
                float value = kLn2 * __builtin_amdgcn_logf(fmaxf(acc, 1e-5f));
                if (normalize) value = fmaf(value, 0.1f, 1.f);
                tile[lane * kTileStride + column] = value;
                acc = 0.f;
#pragma unroll
                for (int piece = 0; piece < kRunB / 4; ++piece) {
                    const float4 four =
                        *reinterpret_cast<const float4*>(mag + start_b + 4 * piece);
                    acc = fmaf(weight_b[4 * piece], four.x, acc);
                    acc = fmaf(weight_b[4 * piece + 1], four.y, acc);
                    acc = fmaf(weight_b[4 * piece + 2], four.z, acc);
                    acc = fmaf(weight_b[4 * piece + 3], four.w, acc);
                }
                acc = quad_sum(acc);
                if ((lane & 3) == 0) {
                    value = kLn2 * __builtin_amdgcn_logf(fmaxf(acc, 1e-5f));
                    if (normalize) value = fmaf(value, 0.1f, 1.f);
                    tile[(64 + (lane >> 2)) * kTileStride + column] = value;
                }
            }
            frontend_fence();
        }
        };
    for (int local = 0; local < valid; local += kPair) transform(raw, raw, local);

    if (!kPeak) {
        if (kMel) {
            // 80 rows x kWaveFrames frames: 4 * kWaveFrames-byte row segments
            const int column = lane & (kWaveFrames - 1);
#pragma unroll 4
            for (int m = lane / kWaveFrames; m < kMels; m += 64 / kWaveFrames) {
                const float value = tile[m * kTileStride + column];
                if (column < valid)
                    out[static_cast<int64_t>(mel_row + m) * ld + frame_off + frame0 + column] =
                        value;
            }
        }
        if (kLoud && lane < valid)
            out[static_cast<int64_t>(loud_row) * ld + frame_off + frame0 + lane] =
                loud_tile[lane];
        frontend_fence();      // the next tile's writes come after these reads
    }
    }   // tiles
    if (kPeak && peak_segment >= 0) {
        peak = wave_max(peak);
        // non-negative floats order like their bit patterns
        if (lane == 0)
            atomicMax(reinterpret_cast<unsigned int*>(seg_peak + peak_segment),
                      __float_as_uint(peak));
    }
}

// three workgroups per CU (registers), each wave looping over its share of the tiles
inline int frontend_grid(int n_tiles) {
    const int groups = (n_tiles + 3) / 4;
    constexpr int resident = 256 * kWavesPerSimd;
    return groups < resident ? groups : resident;
}

size_t frontend_lds_bytes(bool loud) {
    size_t floats = 4 * kWaveFloats;
    if (loud) floats += 516;
    return floats * sizeof(float);
}

}  // namespace emph

using namespace emph;

extern "C" {

int emph_abi_version(void) { return EMPH_ABI_VERSION; }

const char* emph_last_error(void) { return g_error; }

int64_t emph_frontend_table_size(void) { return kTabSize; }

int32_t emph_frontend_block(void) { return kWaveFrames; }

int emph_frontend_table_fill(float* host_table) {
    EMPH_REQUIRE(host_table != nullptr, EMPH_EINVAL, "table is null");
    const double two_pi = 6.283185307179586476925286766559;
    // periodic Hann (torch.hann_window default), rounded to float32 like torch's,
    // then halved: the exact power of two carries the 1/2 of the real-FFT split
    // X[k] = (E + W^k O) / 2 through the whole transform
    for (int n = 0; n < kFft; ++n)
        host_table[kTabWindow + n] =
            0.5f * static_cast<float>(0.5 - 0.5 * cos(two_pi * n / kFft));
    for (int r = 0; r < 8; ++r)
        for (int p = 0; p < 64; ++p) {
            const double angle = -two_pi * (p * r) / 512.;
            host_table[kTabTw1 + 2 * (r * 64 + p)] = static_cast<float>(cos(angle));
            host_table[kTabTw1 + 2 * (r * 64 + p) + 1] =
                static_cast<float>(sin(angle));
        }
    for (int t = 0; t < 8; ++t)
        for (int p0 = 0; p0 < 8; ++p0) {
            const double angle = -two_pi * (p0 * t) / 64.;
            host_table[kTabTw2 + 2 * (t * 8 + p0)] = static_cast<float>(cos(angle));
            host_table[kTabTw2 + 2 * (t * 8 + p0) + 1] =
                static_cast<float>(sin(angle));
        }
    for (int k = 0; k < 514; ++k) {
        const double angle = -two_pi * k / 1024.;
        host_table[kTabTw3 + 2 * k] = static_cast<float>(cos(angle));
        host_table[kTabTw3 + 2 * k + 1] = static_cast<float>(sin(angle));
    }
    return EMPH_OK;
}

int emph_logmel(const void* audio, int32_t audio_format, const int64_t* seg,
                const int32_t* tiles, int32_t n_tiles, const float* table,
                const int32_t* mel_start, const int32_t* mel_count,
                const int32_t* mel_offset, const float* mel_values, int32_t mel_nnz,
                float* out, int64_t ld, int32_t mel_row, int32_t loud_row,
                const float* seg_peak, const float* a_weights, int32_t normalize,
                void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    const bool mel = mel_row >= 0, loud = loud_row >= 0;
    EMPH_REQUIRE(audio && seg && tiles && table && out, EMPH_EINVAL,
                 "emph_logmel: null pointer");
    EMPH_REQUIRE(audio_format == EMPH_AUDIO_F32 || audio_format == EMPH_AUDIO_PCM16,
                 EMPH_EINVAL, "emph_logmel: unknown audio format %d", audio_format);
    EMPH_REQUIRE(mel || loud, EMPH_EINVAL, "emph_logmel: no output row selected");
    EMPH_REQUIRE(!mel || (mel_start && mel_count && mel_offset && mel_values),
                 EMPH_EINVAL, "emph_logmel: mel basis is null");
    EMPH_REQUIRE(!mel || (mel_nnz > 0 && mel_nnz <= 8192), EMPH_ERANGE,
                 "emph_logmel: mel_nnz %d out of range", mel_nnz);
    EMPH_REQUIRE(!loud || (seg_peak && a_weights), EMPH_EINVAL,
                 "emph_logmel: loudness needs seg_peak and a_weights");
    const size_t lds = frontend_lds_bytes(loud);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* peak = const_cast<float*>(seg_peak);
    const bool pcm = audio_format == EMPH_AUDIO_PCM16;
#define EMPH_FRONTEND(MODE, PCM)                                                          \
    do {                                                                                  \
        auto kernel = frontend_kernel<MODE, PCM>;                                         \
        static LdsReservation reserved;                                                   \
        if (int status = reserve_lds(reserved, reinterpret_cast<const void*>(kernel),     \
                                     lds, "emph_logmel"))                                 \
            return status;                                                                \
        EMPH_LAUNCH(kernel, dim3(frontend_grid(n_tiles)), dim3(256), lds, s, audio, seg,  \
                    tiles, table, mel_start, mel_count, mel_offset, mel_values, mel_nnz,  \
                    out, ld, mel_row, loud_row, peak, a_weights, normalize, n_tiles);     \
    } while (0)
    if (mel && loud) {
        if (pcm) EMPH_FRONTEND(2, true); else EMPH_FRONTEND(2, false);
    } else if (mel) {
        if (pcm) EMPH_FRONTEND(0, true); else EMPH_FRONTEND(0, false);
    } else {
        if (pcm) EMPH_FRONTEND(3, true); else EMPH_FRONTEND(3, false);
    }
    return check_launch("emph_logmel");
}

int emph_frontend_peak(const void* audio, int32_t audio_format, const int64_t* seg,
                       const int32_t* tiles, int32_t n_tiles, const float* table,
                       float* seg_peak, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(audio && seg && tiles && table && seg_peak, EMPH_EINVAL,
                 "emph_frontend_peak: null pointer");
    EMPH_REQUIRE(audio_format == EMPH_AUDIO_F32 || audio_format == EMPH_AUDIO_PCM16,
                 EMPH_EINVAL, "emph_frontend_peak: unknown audio format %d", audio_format);
    const size_t lds = frontend_lds_bytes(false);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int32_t* mel_start = nullptr;
    const int32_t* mel_count = nullptr;
    const int32_t* mel_offset = nullptr;
    const float* mel_values = nullptr;
    const float* a_weights = nullptr;
    float* out = nullptr;
    float* peak = seg_peak;
    const int64_t ld = 0;
    const int mel_nnz = 0, mel_row = -1, loud_row = -1, normalize = 0;
    if (audio_format == EMPH_AUDIO_PCM16) EMPH_FRONTEND(1, true); else EMPH_FRONTEND(1, false);
    return check_launch("emph_frontend_peak");
}
#undef EMPH_FRONTEND

}  // extern "C"
