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
// Design (gfx950): one 256-thread workgroup owns a block of 32 frames of one
// chunk.  The 5984 contiguous samples those frames cover are staged ONCE into
// LDS with the zero-pad / slice / reflect index arithmetic applied per sample,
// so HBM sees each audio sample about once (adjacent blocks share 864).  Each
// wave then transforms one frame at a time: a 1024-point real FFT as a
// 512-point complex FFT held 8 points per lane — three radix-8 passes in
// registers with two wave-private LDS transposes — followed by the real-FFT
// split, magnitudes, the 1001-non-zero sparse mel projection and the log.
// The [80 x 32] result tile is staged in LDS and written as 128-byte row
// segments.  Nothing of the 513 x F complex spectrogram ever reaches HBM.
#include <math.h>
#include <stdarg.h>

#include "common.h"

#ifndef EMPH_STAMP
#define EMPH_STAMP(slot)   // in-kernel timeline stamps: tools/micro only
#endif

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
#ifdef EMPH_FE_FENCE_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
    __builtin_amdgcn_wave_barrier();
}

constexpr float kLn2 = 0.69314718055994530942f;
constexpr int kBlockFrames = 32;                               // frames per workgroup
constexpr int kStage = kHop * (kBlockFrames - 1) + kFft;       // 5984 samples
constexpr int kExRow = 72;                                     // complex per exchange row
// Second exchange, C[r][t][p0] at r * kExRowB + t * kExStepB + p0: written by
// lane (r, p0) for each t and read by lane (r, t) for each p0, 8 bytes each.
// 2 * 88 = 48 (mod 64) puts the four r of a 32-lane group 16 banks apart and the
// odd step 9 spreads t (or p0) over those 16: both directions are conflict free
// (with rows of 8 the reads were 4-way conflicts: SQ_LDS_BANK_CONFLICT was 42 %
// of the LDS-active cycles of this kernel).
constexpr int kExRowB = 88;
constexpr int kExStepB = 9;
constexpr int kExFloats = 2 * 8 * kExRowB;                     // 1408 floats per wave
// natural-order spectrum: 4 complex of padding after every 32 so that the
// stride-8 writes of the last pass (lanes t and t + 4 used to collide) and the
// contiguous reads of the real-FFT split are both conflict free
__device__ __forceinline__ int spectrum_slot(int k) { return k + 4 * (k >> 5); }
constexpr int kHighBase = 256 + 4 * (256 >> 5);   // slot of bin 256: only 256..511 are stored
constexpr int kOutStride = kBlockFrames + 1;
constexpr int kMagFloats = 560;      // 513 magnitudes per wave + zero tail for the runs
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

// In-place forward DFT of 8 points, natural order in and out:
// v[r] <- sum_q v[q] exp(-2 pi i q r / 8).  28 packed instructions.
__device__ __forceinline__ void dft8(cf v[8]) {
    constexpr float kH = 0.70710678118654752440f;
    const cf a0 = v[0] + v[4], b0 = v[0] - v[4];
    const cf a1 = v[1] + v[5], d1 = v[1] - v[5];
    const cf a2 = v[2] + v[6], b2 = v[2] - v[6];    // (b2 is used as -i b2 below)
    const cf a3 = v[3] + v[7], d3 = v[3] - v[7];
    // b1 = W8 d1 = kH (d1.x + d1.y, d1.y - d1.x); b3 = W8^3 d3 = -kH (d3.x - d3.y,
    // d3.x + d3.y)
    const cf b1 = add_mi(d1, d1) * kH;
    const cf b3 = sub_mi(d3, d3) * -kH;
    // DFT-4 of a -> even outputs
    cf e0 = a0 + a2, e1 = a0 - a2;
    cf o0 = a1 + a3, t = a1 - a3;
    v[0] = e0 + o0;
    v[4] = e0 - o0;
    v[2] = add_mi(e1, t);
    v[6] = sub_mi(e1, t);
    // DFT-4 of (b0, b1, -i b2, b3) -> odd outputs
    e0 = add_mi(b0, b2);
    e1 = sub_mi(b0, b2);
    o0 = b1 + b3;
    t = b1 - b3;
    v[1] = e0 + o0;
    v[5] = e0 - o0;
    v[3] = add_mi(e1, t);
    v[7] = sub_mi(e1, t);
}

__device__ __forceinline__ float wave_max(float value) {
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1)
        value = fmaxf(value, __shfl_xor(value, offset));
    return value;
}

// MODE 0: mel rows.  MODE 1: per-chunk peak power only.  MODE 2: mel rows and
// loudness row.  MODE 3: loudness row only.
template <int MODE>
__global__ __launch_bounds__(256) void frontend_kernel(
    const float* __restrict__ audio, const int64_t* __restrict__ seg,
    const int32_t* __restrict__ tiles, const float* __restrict__ table,
    const int32_t* __restrict__ mel_start, const int32_t* __restrict__ mel_count,
    const int32_t* __restrict__ mel_offset, const float* __restrict__ mel_values,
    int mel_nnz, float* __restrict__ out, int64_t ld, int mel_row, int loud_row,
    float* __restrict__ seg_peak, const float* __restrict__ a_weights,
    int normalize, int n_tiles) {
    constexpr bool kMel = MODE == 0 || MODE == 2;
    constexpr bool kLoud = MODE == 2 || MODE == 3;
    constexpr bool kPeak = MODE == 1;

    extern __shared__ __align__(16) float lds[];
    EMPH_STAMP(0);
    float* stage = lds;                                  // [kStage]
    float* exchange = stage + kStage;                    // [4][kExFloats]
    float* tile = exchange + 4 * kExFloats;              // [80][kOutStride]
    float* loud_tile = tile + kMels * kOutStride;        // [kBlockFrames]
    float* mel_sums = loud_tile + kBlockFrames;          // [4][kMagFloats] magnitudes
    float* weights = mel_sums + 4 * kMagFloats;          // [513+]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    if (kLoud)
        for (int index = tid; index < kBins; index += 256)
            weights[index] = a_weights[index];
    if (kMel)   // zero tail of each wave's magnitude row: the fixed-length mel
                // runs read up to kMagPad bins past the last one (weight 0)
        for (int index = tid; index < 4 * (kMagFloats - kBins); index += 256)
            mel_sums[(index / (kMagFloats - kBins)) * kMagFloats + kBins +
                     index % (kMagFloats - kBins)] = 0.f;

    // ---- per-lane constants
    const int p = lane;              // pass-1 position
    const int r1 = lane >> 3;        // pass-2/3 residue r
    const int p0 = lane & 7;         // pass-2 position / pass-3 output t
    float window[16];
    cf tw1[8], tw2[8], tw3[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        window[2 * q] = table[kTabWindow + 2 * (p + 64 * q)];
        window[2 * q + 1] = table[kTabWindow + 2 * (p + 64 * q) + 1];
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
    // registers for the whole block, zero-padded to a fixed length so that the
    // per-frame loops are fully unrolled with every LDS read independent.
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

    // ---- persistent loop over this workgroup's blocks of 32 frames: the
    // per-lane tables above are loaded once (96 global loads per lane)
    cf* ex = reinterpret_cast<cf*>(exchange + wave * kExFloats);
    float peak = 0.f;
    int peak_segment = -1;
    for (int block = blockIdx.x; block < n_tiles; block += gridDim.x) {
    const int segment = tiles[EMPH_TILE_FIELDS * block];
    const int frame0 = tiles[EMPH_TILE_FIELDS * block + 1];
    const int64_t* row = seg + static_cast<int64_t>(segment) * EMPH_SEG_FIELDS;
    const int64_t audio_off = row[EMPH_SEG_AUDIO_OFF];
    const int64_t audio_len = row[EMPH_SEG_AUDIO_LEN];
    const int64_t start = row[EMPH_SEG_START];
    const int64_t length = row[EMPH_SEG_LENGTH];
    const int64_t frame_off = row[EMPH_SEG_FRAME_OFF];
    const int frames = static_cast<int>(row[EMPH_SEG_FRAMES]);
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

    // ---- stage the block's samples: zero pad + slice + reflect in one pass.
    // Every load is unconditional (clamped address, masked value) and the loop
    // is fully unrolled, so a thread's 24 requests are in flight together; a
    // predicated, rolled loop pays one 1-2 us memory round trip per sample.
    const int64_t first = static_cast<int64_t>(frame0) * kHop - kPad;
    __syncthreads();          // the previous block is done with `stage` and `tile`
    {
        constexpr int kPerThread = (kStage + 255) / 256;
        float value[kPerThread];
        const float* source = audio + audio_off;
        const int64_t base = start + first - kPad;      // audio index of stage[0]
        if (first >= 0 && first + kStage <= length && base >= 0 &&
            base + kStage <= audio_len) {
            // interior block (all but the first and last of an utterance): a
            // straight copy, no per-sample index arithmetic
            const float* from = source + base + tid;
#pragma unroll
            for (int j = 0; j < kPerThread; ++j)
                value[j] = from[tid + 256 * j < kStage ? 256 * j : 0];
#pragma unroll
            for (int j = 0; j < kPerThread; ++j)
                if (tid + 256 * j < kStage) stage[tid + 256 * j] = value[j];
        } else {
            bool live[kPerThread];
#pragma unroll
            for (int j = 0; j < kPerThread; ++j) {
                int64_t r = first + tid + 256 * j;     // position in the chunk
                if (r < 0) r = -r;                     // reflect (no edge repeat)
                if (r >= length) r = 2 * (length - 1) - r;
                const int64_t a = start + r - kPad;    // undo the 432 zero pad
                live[j] = r >= 0 && r < length && a >= 0 && a < audio_len;
                const int64_t clamped = a < 0 ? 0 : (a >= audio_len ? audio_len - 1 : a);
                value[j] = source[audio_len > 0 ? clamped : 0];
            }
#pragma unroll
            for (int j = 0; j < kPerThread; ++j) {
                const int index = tid + 256 * j;
                if (index < kStage) stage[index] = live[j] ? value[j] : 0.f;
            }
        }
    }
    __syncthreads();
    EMPH_STAMP(1);

    float floor_db = 0.f;
    if (kLoud) {
        // librosa.amplitude_to_db: max(D) - top_db with D = 10 log10(max(amin^2, S^2))
        const float top = 10.f * log10f(fmaxf(1e-10f, seg_peak[segment]));
        floor_db = top - 80.f;
    }

    for (int local = wave; local < kBlockFrames; local += 4) {
        if (frame0 + local >= frames) break;   // wave-uniform
        const float* samples = stage + local * kHop;

        // pass 1: radix-8 over q of z[p + 64 q], z[n] = y[2n] + i y[2n+1]
        cf v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[q] = *reinterpret_cast<const cf*>(samples + 2 * (p + 64 * q)) *
                   cf{window[2 * q], window[2 * q + 1]};
        }
        if (local == wave) EMPH_STAMP(2);
        dft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r) v[r] = cmul(v[r], tw1[r]);
            ex[r * kExRow + p] = v[r];
        }
        frontend_fence();

        if (local == wave) EMPH_STAMP(3);
        // pass 2: lane (r, p0) takes A[p0 + 8 p1][r], radix-8 over p1
#pragma unroll
        for (int p1 = 0; p1 < 8; ++p1) v[p1] = ex[r1 * kExRow + p0 + 8 * p1];
        frontend_fence();
        dft8(v);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t) v[t] = cmul(v[t], tw2[t]);
            ex[r1 * kExRowB + t * kExStepB + p0] = v[t];     // C[r][t][p0]
        }
        frontend_fence();

        if (local == wave) EMPH_STAMP(4);
        // pass 3: lane (r, t) takes C[r][p0][t], radix-8 over p0
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ex[r1 * kExRowB + p0 * kExStepB + q];
        frontend_fence();
        dft8(v);
        // Z[r + 8 t + 64 u] = v[u].  Real-FFT split, two bins per pair: with
        // E = Z[k] + conj(Z[512-k]) and O' = (-i W^k)(Z[k] - conj(Z[512-k])),
        //     X[k] = E + O'        conj(X[512-k]) = E - O'
        // (the 1/2 rides on the window table).  A lane owns the pairs of its
        // four low bins k = r + 8 t + 64 u, u < 4, so only the HIGH half of the
        // spectrum (u >= 4) crosses lanes: four 8-byte LDS writes and four reads
        // per lane instead of eight and sixteen for a natural-order round trip.
#pragma unroll
        for (int u = 4; u < 8; ++u) ex[spectrum_slot(r1 + 8 * p0 + 64 * u) - kHighBase] = v[u];
        frontend_fence();

        if (local == wave) EMPH_STAMP(5);
        float power[9];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = r1 + 8 * p0 + 64 * u;
            // k = 0 pairs with itself: bins 0 and 512 come out of the same formulas
            cf zm = ex[spectrum_slot((512 - k) & 511 ? (512 - k) & 511 : 256) - kHighBase];
            if (u == 0) zm = lane == 0 ? v[0] : zm;
            const cf e = add_conj(v[u], zm);
            const cf o = cmul(tw3[u], sub_conj(v[u], zm));
            const cf low = e + o, high = e - o;
            power[u] = low.x * low.x + low.y * low.y;            // |X[k]|^2
            power[4 + u] = high.x * high.x + high.y * high.y;    // |X[512 - k]|^2
        }
        // k = 256 pairs with itself: X[256] = conj(Z[256]) (lane 0 holds it, u = 4);
        // the window's 1/2 has to be undone there
        power[8] = 4.f * (v[4].x * v[4].x + v[4].y * v[4].y);
        frontend_fence();

        if (local == wave) EMPH_STAMP(6);
        if (kPeak) {
#pragma unroll
            for (int j = 0; j < 8; ++j) peak = fmaxf(peak, power[j]);
            if (lane == 0) peak = fmaxf(peak, power[8]);
            continue;
        }

        if (kLoud) {
            // 10 log10(max(1e-10, |X|^2)) floored at peak - 80, + A-weight,
            // clamped at MIN_DB = -100, mean over the 513 bins
            double total = 0.;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                if (j == 8 && lane != 0) break;
                const int k = bin_of(j);
                float db = 10.f * log10f(fmaxf(1e-10f, power[j]));
                db = fmaxf(db, floor_db) + weights[k];
                total += static_cast<double>(fmaxf(db, -100.f));
            }
#pragma unroll
            for (int offset = 32; offset > 0; offset >>= 1)
                total += __shfl_xor(total, offset);
            if (lane == 0) {
                float value = static_cast<float>(total / 513.);
                if (normalize) value = (value + 100.f) / 100.f;
                loud_tile[local] = value;
            }
        }

        if (kMel) {
            float* mag = mel_sums + wave * kMagFloats;
#pragma unroll
            // v_sqrt_f32 (1 ulp; the argument is >= 1e-6, never denormal): the
            // correctly rounded sqrtf is ten more instructions per bin
            for (int j = 0; j < 8; ++j)
                mag[bin_of(j)] = __builtin_amdgcn_sqrtf(power[j] + 1e-6f);
            if (lane == 0) mag[256] = __builtin_amdgcn_sqrtf(power[8] + 1e-6f);
            frontend_fence();
            float acc = 0.f;
#pragma unroll
            for (int piece = 0; piece < kRunA / 4; ++piece) {
                const float4 four = *reinterpret_cast<const float4*>(mag + start_a + 4 * piece);
                acc = fmaf(weight_a[4 * piece], four.x, acc);
                acc = fmaf(weight_a[4 * piece + 1], four.y, acc);
                acc = fmaf(weight_a[4 * piece + 2], four.z, acc);
                acc = fmaf(weight_a[4 * piece + 3], four.w, acc);
            }
            // natural log on v_log_f32 (log2, 1 ulp; the argument is >= 1e-5);
            // (x + 10) / 10 as one fma: both within 1e-7 of the exact forms
            float value = kLn2 * __builtin_amdgcn_logf(fmaxf(acc, 1e-5f));
            if (normalize) value = fmaf(value, 0.1f, 1.f);
            tile[lane * kOutStride + local] = value;
            acc = 0.f;
#pragma unroll
            for (int piece = 0; piece < kRunB / 4; ++piece) {
                const float4 four = *reinterpret_cast<const float4*>(mag + start_b + 4 * piece);
                acc = fmaf(weight_b[4 * piece], four.x, acc);
                acc = fmaf(weight_b[4 * piece + 1], four.y, acc);
                acc = fmaf(weight_b[4 * piece + 2], four.z, acc);
                acc = fmaf(weight_b[4 * piece + 3], four.w, acc);
            }
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if ((lane & 3) == 0) {
                value = kLn2 * __builtin_amdgcn_logf(fmaxf(acc, 1e-5f));
                if (normalize) value = fmaf(value, 0.1f, 1.f);
                tile[(64 + (lane >> 2)) * kOutStride + local] = value;
            }
            frontend_fence();
        }
        if (local == wave) EMPH_STAMP(7);
    }
    EMPH_STAMP(8);

    if (!kPeak) {
        __syncthreads();
        const int valid = min(kBlockFrames, frames - frame0);
        if (kMel) {
            // 80 rows x 32 frames: 8 rows per pass, 128-byte segments
            const int column = tid & 31;
            for (int m = tid >> 5; m < kMels; m += 8)
                if (column < valid)
                    out[static_cast<int64_t>(mel_row + m) * ld + frame_off + frame0 +
                        column] = tile[m * kOutStride + column];
        }
        if (kLoud && tid < valid)
            out[static_cast<int64_t>(loud_row) * ld + frame_off + frame0 + tid] =
                loud_tile[tid];
    }
    EMPH_STAMP(9);
    }   // blocks

    if (kPeak && peak_segment >= 0) {
        peak = wave_max(peak);
        // non-negative floats order like their bit patterns
        if (lane == 0)
            atomicMax(reinterpret_cast<unsigned int*>(seg_peak + peak_segment),
                      __float_as_uint(peak));
    }
}

// two workgroups per CU (LDS), each looping over its share of the blocks
inline int frontend_grid(int n_tiles) { return n_tiles < 512 ? n_tiles : 512; }

size_t frontend_lds_bytes(bool loud) {
    size_t floats = kStage + 4 * kExFloats + kMels * kOutStride + kBlockFrames +
                    4 * kMagFloats;
    if (loud) floats += 516;
    return floats * sizeof(float);
}

}  // namespace emph

using namespace emph;

extern "C" {

int emph_abi_version(void) { return EMPH_ABI_VERSION; }

const char* emph_last_error(void) { return g_error; }

int64_t emph_frontend_table_size(void) { return kTabSize; }

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

int emph_logmel(const float* audio, const int64_t* seg, const int32_t* tiles,
                int32_t n_tiles, const float* table, const int32_t* mel_start,
                const int32_t* mel_count, const int32_t* mel_offset,
                const float* mel_values, int32_t mel_nnz, float* out,
                int64_t ld, int32_t mel_row, int32_t loud_row,
                const float* seg_peak, const float* a_weights,
                int32_t normalize, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    const bool mel = mel_row >= 0, loud = loud_row >= 0;
    EMPH_REQUIRE(audio && seg && tiles && table && out, EMPH_EINVAL,
                 "emph_logmel: null pointer");
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
#define EMPH_FRONTEND(MODE)                                                    \
    EMPH_LAUNCH(frontend_kernel<MODE>, dim3(frontend_grid(n_tiles)),    \
                       dim3(256), lds, s, audio, seg, tiles, table, mel_start, \
                       mel_count, mel_offset, mel_values, mel_nnz, out, ld,    \
                       mel_row, loud_row, peak, a_weights, normalize, n_tiles)
    if (mel && loud) {
        EMPH_FRONTEND(2);
    } else if (mel) {
        EMPH_FRONTEND(0);
    } else {
        EMPH_FRONTEND(3);
    }
    return check_launch("emph_logmel");
}

int emph_frontend_peak(const float* audio, const int64_t* seg,
                       const int32_t* tiles, int32_t n_tiles,
                       const float* table, float* seg_peak, void* stream) {
    if (n_tiles == 0) return EMPH_OK;
    EMPH_REQUIRE(audio && seg && tiles && table && seg_peak, EMPH_EINVAL,
                 "emph_frontend_peak: null pointer");
    const size_t lds = frontend_lds_bytes(false);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int32_t* none_i = nullptr;
    const float* none_f = nullptr;
    float* none_o = nullptr;
    EMPH_LAUNCH(frontend_kernel<1>, dim3(frontend_grid(n_tiles)), dim3(256), lds,
                       s, audio, seg, tiles, table, none_i, none_i, none_i, none_f, 0,
                       none_o, int64_t{0}, -1, -1, seg_peak, none_f, 0, n_tiles);
    return check_launch("emph_frontend_peak");
}
#undef EMPH_FRONTEND

}  // extern "C"
