/*
 * emphases_hip.h — C ABI of the MI355X (gfx950) prominence-inference hot path.
 *
 * libemphases_hip.so replaces the ATen calls that interactiveaudiolab/emphases
 * issues on its inference path (emphases/core.py:223-265 and below).  Every
 * entry point is a plain `extern "C"` function over raw device pointers and
 * sizes: no torch types cross this boundary.  The reference interface each one
 * replaces is cited as file:line relative to the reference repository.
 *
 * Conventions
 *   - All pointers are DEVICE pointers on the current HIP device unless the
 *     parameter name starts with `host_`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     work is enqueued asynchronously; nothing here synchronises the device
 *     or allocates memory, so every call is hipGraph-capturable.
 *   - Return value: 0 on success, a negative EMPH_E* code on a contract
 *     violation, or a positive hipError_t if a launch failed.
 *     `emph_last_error()` returns a message for the calling thread.
 *   - Activations live in "packed ragged" form: a [channels, ld] row-major
 *     float32 matrix whose column axis is the concatenation of all segments
 *     (utterances or word-boundary chunks, emphases/core.py:361-418).  Each
 *     segment keeps its OWN zero 'same' halo — results equal the reference's
 *     one-utterance-at-a-time (B=1) semantics, not its padded-batch semantics
 *     (model/layers/convolution.py:35-37 ignores lengths).
 *
 * Segment table: int64 [num_segments][EMPH_SEG_FIELDS], one row per segment.
 */
#ifndef EMPHASES_HIP_H
#define EMPHASES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMPH_ABI_VERSION 30

/* Segment-table fields */
enum {
    EMPH_SEG_AUDIO_OFF = 0, /* first sample of the utterance in `audio`        */
    EMPH_SEG_AUDIO_LEN = 1, /* samples in the utterance                        */
    EMPH_SEG_START = 2,     /* chunk start, in the 432-zero-padded signal
                               (emphases/core.py:357-358,395-401)              */
    EMPH_SEG_LENGTH = 3,    /* chunk length in samples                         */
    EMPH_SEG_FRAME_OFF = 4, /* first column of the chunk on the frame axis     */
    EMPH_SEG_FRAMES = 5,    /* frames in the chunk                             */
    EMPH_SEG_WORD_OFF = 6,  /* first column of the chunk on the word axis      */
    EMPH_SEG_WORDS = 7,     /* words in the chunk                              */
    EMPH_SEG_FIELDS = 8
};

/* Which axis of the segment table a ragged op walks */
enum { EMPH_AXIS_FRAMES = 0, EMPH_AXIS_WORDS = 1 };

/* Tile table: int32 [num_tiles][EMPH_TILE_FIELDS], 16-byte aligned.  Ragged
 * ops work on fixed-width blocks of positions of one segment on one axis; the
 * row carries everything a block needs so that no kernel chases pointers. */
enum {
    EMPH_TILE_SEGMENT = 0, /* row of the segment table                        */
    EMPH_TILE_FIRST = 1,   /* first position of the block inside the segment  */
    EMPH_TILE_OFFSET = 2,  /* first column of the segment on the packed axis  */
    EMPH_TILE_COUNT = 3,   /* positions in the segment                        */
    EMPH_TILE_FIELDS = 4
};

/* Activations (emphases/config/defaults.py:181 and config/hparam-search/) */
enum {
    EMPH_ACT_NONE = 0,
    EMPH_ACT_RELU = 1,
    EMPH_ACT_GELU = 2,      /* exact erf form, torch.nn.GELU default          */
    EMPH_ACT_SILU = 3,
    EMPH_ACT_LEAKY_RELU = 4 /* slope 0.01, torch.nn.LeakyReLU default         */
};

/* Word-boundary reductions (emphases/core.py:426-469) */
enum {
    EMPH_REDUCE_SUM = 0,
    EMPH_REDUCE_AVERAGE = 1,
    EMPH_REDUCE_MAX = 2,
    EMPH_REDUCE_CENTER = 3
};

/* Sample formats of the packed audio buffer.  The reference converts 16-bit
 * PCM to float32 on the host when it loads a file (torchaudio.load,
 * emphases/load.py:11-17); EMPH_AUDIO_PCM16 moves that x / 32768 into the
 * front-end so that half the bytes cross PCIe and HBM.  Results are identical
 * bit for bit (the scale is an exact power of two). */
enum { EMPH_AUDIO_F32 = 0, EMPH_AUDIO_PCM16 = 1 };

/* Postprocess (emphases/core.py:335-342) */
enum { EMPH_POST_NONE = 0, EMPH_POST_SIGMOID = 1, EMPH_POST_CLAMP01 = 2 };

/* Error codes */
enum {
    EMPH_OK = 0,
    EMPH_EINVAL = -1,   /* bad argument (null pointer, unsupported size)       */
    EMPH_ERANGE = -2    /* size outside what the kernels are built for         */
};

int emph_abi_version(void);
const char* emph_last_error(void);

/* ------------------------------------------------------------------------ */
/* Host staging                                                              */
/* ------------------------------------------------------------------------ */

/* Copy `count` host buffers into one host destination (the pinned staging
 * buffer of a ragged batch) with `threads` threads (a persistent pool; the
 * caller's thread copies too).  All pointers are HOST pointers.  Replaces the
 * per-utterance `audio.to(device)` of emphases/data/preprocess/core.py:74 on
 * the way to ONE host-to-device copy per batch.
 *   host_offsets[i]  byte offset of buffer i in the destination */
int emph_host_gather(const void* const* host_sources, const int64_t* host_bytes,
                     const int64_t* host_offsets, int32_t count,
                     void* host_destination, int32_t threads);

/* ------------------------------------------------------------------------ */
/* The file boundary of from_files_to_files, a batch at a time               */
/* ------------------------------------------------------------------------ */

/* emphases.from_files_to_files (emphases/core.py:115-179) loads one alignment
 * (`pypar.Alignment(text_file)`, core.py:49,107) and one audio file
 * (`emphases.load.audio`, load.py:11-17) at a time on the Python thread and
 * writes `<prefix>.TextGrid` + `<prefix>.pt` (core.py:111-112) the same way.
 * These entry points do that for a BATCH of files on a pool of host threads
 * (no device code, no Python per file): Praat TextGrids (long or short text
 * format; UTF-8 with or without BOM, UTF-16) are parsed and the RIFF/WAVE
 * chunk lists walked in parallel; the samples of mono 16-bit PCM / float32
 * files are read straight into the batch's (pinned) staging buffer; outputs
 * are written in parallel.  Same grammar, same gap filling, same walker and
 * the same written bytes as emphases_amd/alignment.py / load.py. */
typedef struct emph_file_batch emph_file_batch;

/* The file pool's own threads onto the given CPUs (those next to the GPU whose
 * DMA engine reads what they copy); the caller's threads are not touched. */
int emph_files_affinity(const int32_t* cpus, int32_t count);

/* Parse text_paths[i] (.TextGrid) and walk the headers of audio_paths[i]
 * (.wav), i < count.  A file that fails is reported per file (emph_files_sizes
 * status, emph_files_error), not as an error of the call. */
int emph_files_open(const char* const* text_paths,
                    const char* const* audio_paths, int32_t count,
                    int32_t threads, emph_file_batch** batch);
void emph_files_close(emph_file_batch* batch);
const char* emph_files_error(const emph_file_batch* batch, int32_t index);

/* sizes int64 [count][12] = {status (bit 0: alignment failed, bit 1: audio
 * failed), words (gaps filled), phonemes (-1: no phoneme tier), bytes of word
 * labels, bytes of phoneme labels, WAVE format code, channels, sample rate,
 * bits per sample, data offset, data bytes, phoneme tier first?} */
int emph_files_sizes(const emph_file_batch* batch, int64_t* sizes);

/* All alignments back to back (any pointer may be NULL): word_times float64
 * [words][2] seconds, word labels as UTF-8 bytes with END offsets per word;
 * the same for the phonemes plus the index (within its file) of the word each
 * belongs to; per file "word tier\nphoneme tier" with END offsets. */
int emph_files_alignments(const emph_file_batch* batch, double* word_times,
                          char* word_text, int64_t* word_text_end,
                          double* phone_times, char* phone_text,
                          int64_t* phone_text_end, int32_t* phone_word,
                          char* tier_names, int64_t* tier_names_end);
int64_t emph_files_tier_name_bytes(const emph_file_batch* batch);

/* bytes[k] bytes of the data chunk of file which[k] to destination +
 * where[k] (a HOST pointer: the pinned staging buffer). */
int emph_files_read_audio(const emph_file_batch* batch, const int32_t* which,
                          const int64_t* where, const int64_t* bytes,
                          int32_t count, void* destination, int32_t threads);

/* <prefixes[k]>.TextGrid = the alignment of file which[k] as loaded (both
 * tiers, original tier names and order) and <prefixes[k]>.pt = torch.save of
 * the float32 CPU tensor [1, W] = scores[first[k] .. first[k + 1]) (HOST
 * pointer), readable by torch.load. */
int emph_files_write(const emph_file_batch* batch, const int32_t* which,
                     const char* const* prefixes, const float* scores,
                     const int64_t* first, int32_t count, int32_t threads);

/* Integer tables of a batch plan built on the host (what emphases_amd/batch.py
 * builds with numpy; host arithmetic on the segment lengths and word bounds
 * of emphases.preprocess, emphases/core.py:345-418).
 * emph_plan_tiles: the tile table int32 [rows][4] of an axis for `block`-wide
 * tiles of the segments with least <= count <= most (host_tiles NULL: returns
 * the number of rows).
 * emph_plan_word_sums: the tables of the folded per-word sum for running sums
 * that restart at the sorted frame columns `restarts` (see
 * emph_conv1d_stack / emph_word_sums); returns the number of terms, or
 * -(needed) when `capacity` is too small. */
int64_t emph_plan_tiles(const int64_t* host_counts, const int64_t* host_offsets,
                        int32_t n_segments, int32_t block, int64_t least,
                        int64_t most, int32_t* host_tiles);
/* emph_plan_batch: the chunk of every utterance of a batch when none needs more
 * than one - emphases.preprocess (emphases/core.py:345-418) with the default
 * batch_size=None, the reference's float64 operations in the reference's order.
 * Utterance u: counts[u] words, whose (start, end) seconds are the next counts[u]
 * rows of `times`, and lengths[u] samples at `sample_rate`.  One output row per
 * utterance that yields a chunk (*n_segments; core.py:414-415 drops the others):
 * utterance, start_sample, length, frames, words; bounds int64 [2][capacity]
 * (capacity >= sum(counts)): the first *n_words columns are those chunks' words'
 * chunk-relative (start, end) frames.  Returns 0; 1 when the batch has to be
 * planned one utterance at a time (several chunks, a negative duration, a time
 * that is not finite or whose frame index reaches 2^52: nothing is written
 * then); < 0 on bad arguments. */
int emph_plan_batch(const double* times, const int64_t* counts,
                    const int64_t* lengths, int32_t n_utterances,
                    int64_t sample_rate, int64_t hopsize, int64_t padding,
                    int64_t num_fft, int64_t* utterance, int64_t* start_sample,
                    int64_t* length, int64_t* frames, int64_t* words,
                    int64_t* bounds, int64_t capacity, int64_t* n_segments,
                    int64_t* n_words);
int64_t emph_plan_word_sums(const int64_t* frames, const int64_t* frame_off,
                            const int64_t* words, int32_t n_segments,
                            const int64_t* word_columns, const int64_t* bounds,
                            int64_t total_words, const int64_t* restarts,
                            int64_t n_restarts, int64_t ld_frames,
                            int64_t ld_words, int32_t* slot_map, int32_t* first,
                            int32_t* lengths, int32_t* terms, int64_t capacity,
                            int32_t* n_slots);

/* ------------------------------------------------------------------------ */
/* Front-end: framed log-mel (+ optional A-weighted loudness row)            */
/* ------------------------------------------------------------------------ */

/* Number of floats in the constant table consumed by emph_logmel():
 * periodic Hann window [1024], FFT twiddles.  Layout is private to the
 * library; fill it on the host once and upload it. */
int64_t emph_frontend_table_size(void);

/* Frames per tile of the frame-axis tile table emph_logmel / emph_frontend_peak
 * walk (one wave transforms one tile at a time). */
int32_t emph_frontend_block(void);

/* Fill `host_table` (emph_frontend_table_size() floats), computed in double
 * precision.  Replaces torch.hann_window (emphases/data/preprocess/mels.py:
 * 19-29) and the FFT plan inside torch.stft (mels.py:39-47). */
int emph_frontend_table_fill(float* host_table);

/* Log-mel features of every segment.
 *
 * Replaces, per chunk: F.pad(audio,(432,432)) + slice (emphases/core.py:
 * 357-358,401), reflect pad (mels.py:31-36), torch.stft(1024, hop 160,
 * periodic Hann, center=False) (mels.py:39-48), sqrt(re^2+im^2+1e-6)
 * (mels.py:51), mel projection + log(clamp(.,1e-5)) (mels.py:94-109) and the
 * optional (x+10)/10 (mels.py:57-58) — none of it materialised.
 *
 *   audio        float32 / int16 [*] all utterances back to back (audio_format)
 *   seg          int64 [n_seg][8]   segment table
 *   tiles        int32 [n_tiles][4] tile table of the frame axis, blocks of
 *                                   emph_frontend_block() frames
 *   table        float32            from emph_frontend_table_fill
 *   mel_start/mel_count/mel_offset  int32 [80] run of each filterbank row
 *   mel_values   float32 [nnz]      run values (librosa.filters.mel restated);
 *                                   runs of rows 0..63 may hold at most 20 bins,
 *                                   runs of rows 64..79 at most 40
 *   out          float32 [rows, ld] rows 0..79 receive the mel rows when
 *                                   `mel_row >= 0` (row index of the first)
 *   loud_row     row that receives A-weighted loudness, or -1
 *   seg_peak     float32 [n_seg]    per-chunk max |X|^2 from
 *                                   emph_frontend_peak (loudness only)
 *   a_weights    float32 [513]      A-weighting minus REF_DB (loudness only)
 *   normalize    0/1                emphases NORMALIZE switch
 */
int emph_logmel(const void* audio, int32_t audio_format, const int64_t* seg,
                const int32_t* tiles, int32_t n_tiles, const float* table,
                const int32_t* mel_start, const int32_t* mel_count,
                const int32_t* mel_offset, const float* mel_values,
                int32_t mel_nnz, float* out, int64_t ld, int32_t mel_row,
                int32_t loud_row, const float* seg_peak,
                const float* a_weights, int32_t normalize, void* stream);

/* Per-chunk max power spectrum value, needed by librosa.amplitude_to_db's
 * top_db=80 floor, which is relative to the max over the WHOLE chunk
 * (emphases/data/preprocess/loudness.py:84-93).  `seg_peak` float32 [n_seg]
 * must be zeroed by the caller before the launch. */
int emph_frontend_peak(const void* audio, int32_t audio_format,
                       const int64_t* seg, const int32_t* tiles,
                       int32_t n_tiles, const float* table, float* seg_peak,
                       void* stream);

/* Sample-rate conversion of a whole batch on the device.
 *
 * Replaces emphases.resample (emphases/core.py:613-619): torchaudio.transforms.
 * Resample(sample_rate, 16000) = a strided conv1d with a polyphase windowed-sinc
 * kernel (Hann window, lowpass_filter_width 6, rolloff 0.99).  The kernel table
 * is built on the host in float64 (emphases_amd/load.py::resample_kernel).
 *
 *   audio   float32 / int16 [*]     utterances at the ORIGINAL rate, back to back
 *   table   int64 [n][4]            (source offset, source samples, target offset,
 *                                   target samples = ceil(new * samples / orig))
 *   kernel  float32 [new][2 width + orig]
 *   orig, fresh                     the two rates divided by their gcd
 *   out     float32 [*]             utterances at the new rate
 *   most_samples                    the largest target count (sizes the grid) */
int emph_resample(const void* audio, int32_t audio_format, const int64_t* table,
                  int32_t n_utterances, int64_t most_samples,
                  const float* kernel, int32_t orig, int32_t fresh,
                  int32_t width, float* out, void* stream);

/* Pitch / periodicity rows of the feature matrix.
 *
 * Replaces what emphases/data/preprocess/core.py:94-106,123 does with the
 * outputs of the pitch tracker (`penn.from_audio`, core.py:84-92 — a third-party
 * neural network that stays outside this library): torch.log2(pitch), or
 * (log2(pitch) - LOGFMIN) / (LOGFMAX - LOGFMIN) under NORMALIZE
 * (config/static.py:33-36), the periodicity row as it is, and the torch.cat
 * that puts them under the mel rows.
 *
 *   pitch, periodicity  float32 [ld]  tracker outputs on the packed frame axis
 *                                     (Hz; columns outside segments are ignored
 *                                     downstream and may hold anything)
 *   out                 float32 [rows, ld]  feature matrix
 *   pitch_row, periodicity_row        destination rows, or -1 to skip
 *   logfmin, logfmax    float32 log2(FMIN), log2(FMAX)
 * ld must be a multiple of 4 and the rows 16-byte aligned. */
int emph_pitch_rows(const float* pitch, const float* periodicity, float* out,
                    int64_t ld, int32_t pitch_row, int32_t periodicity_row,
                    int32_t normalize, float logfmin, float logfmax,
                    void* stream);

/* ------------------------------------------------------------------------ */
/* Conv1d 'same' (+ bias + activation) over ragged segments, fp32 MFMA       */
/* ------------------------------------------------------------------------ */

/* Number of floats of the MFMA-fragment-ordered weight pack for a Conv1d /
 * Linear weight [c_out, c_in, k]. */
int64_t emph_conv_pack_size(int32_t c_out, int32_t c_in, int32_t kernel_size);

/* Reorder a torch-layout weight [c_out][c_in][k] (host) into the pack. */
int emph_conv_pack(const float* host_weight, int32_t c_out, int32_t c_in,
                   int32_t kernel_size, float* host_pack);

/* y = act(conv1d(x, w, padding='same') + b) independently per segment.
 *
 * Replaces torch.nn.Conv1d(padding='same') + activation (emphases/model/
 * core.py:17-21,93; model/layers/convolution.py:25-30) and, with
 * kernel_size 1, torch.nn.Linear inside nn.TransformerEncoderLayer
 * (model/layers/transformer.py:18-23).
 *
 *   x     float32 [c_in, ldx]   y  float32 [c_out, ldy]
 *   tiles int32 [n_tiles][4]    tile table of the walked axis, block = tile_n
 *   tile_n 16, 32 or 64 positions per tile (one wave each)
 *   transpose_out  0: y is [c_out, ldy];  1: y is [ldy, c_out] (position-major)
 */
int emph_conv1d(const float* x, int64_t ldx, float* y, int64_t ldy,
                const float* pack, const float* bias, int32_t c_in,
                int32_t c_out, int32_t kernel_size, int32_t activation,
                const int32_t* tiles, int32_t n_tiles, int32_t tile_n,
                int32_t transpose_out, void* stream);

/* Winograd F(2,3) form of the kernel_size-3 convolution: the same contract and
 * result as emph_conv1d(kernel_size = 3) up to fp32 rounding (~1e-7 of the
 * output scale), with two thirds of the matrix-core work.  Weights are packed
 * by emph_conv_winograd_pack (transformed in float64 on the host); tile_n is 32
 * or 64 and the whole pack must fit in LDS (c_in * 1.25 KiB per 5 m-tiles). */
int64_t emph_conv_winograd_pack_size(int32_t c_out, int32_t c_in);
/* LDS bytes one workgroup needs (tile_n = 64); must be <= 160 KiB. */
int64_t emph_conv_winograd_lds_bytes(int32_t c_out, int32_t c_in);
int emph_conv_winograd_pack(const float* host_weight, int32_t c_out,
                            int32_t c_in, float* host_pack);
int emph_conv1d_winograd(const float* x, int64_t ldx, float* y, int64_t ldy,
                         const float* pack, const float* bias, int32_t c_in,
                         int32_t c_out, int32_t activation,
                         const int32_t* tiles, int32_t n_tiles, int32_t tile_n,
                         void* stream);

/* Winograd F(4,3) form: six GEMMs over quads of positions, half of the direct
 * form's matrix-core work; the same contract as emph_conv1d(kernel_size = 3)
 * for identity / ReLU, with 64-position tiles (the tile table's block), c_in a
 * multiple of 4, c_out <= 96 and the whole pack in LDS
 * (emph_conv_winograd4_lds_bytes <= 160 KiB: up to 80 x 80). */
int64_t emph_conv_winograd4_pack_size(int32_t c_out, int32_t c_in);
int64_t emph_conv_winograd4_lds_bytes(int32_t c_out, int32_t c_in);
int emph_conv_winograd4_pack(const float* host_weight, int32_t c_out,
                             int32_t c_in, float* host_pack);
int emph_conv1d_winograd4(const float* x, int64_t ldx, float* y, int64_t ldy,
                          const float* pack, const float* bias, int32_t c_in,
                          int32_t c_out, int32_t activation,
                          const int32_t* tiles, int32_t n_tiles, void* stream);
/* ... followed by PositionalEncoding (emphases/model/layers/transformer.py:45-52:
 * `x + pe[:T]`, dropout the identity in eval) in the same launch, for the input
 * layer in front of a Transformer encoder (model/core.py:88-99).  `position`:
 * the sin / cos table CHANNEL-major, float32 [c_out][max_positions]; column t
 * is added to position t of every segment after the activation. */
int emph_conv1d_winograd4_position(const float* x, int64_t ldx, float* y,
                                   int64_t ldy, const float* pack,
                                   const float* bias, int32_t c_in,
                                   int32_t c_out, int32_t activation,
                                   const int32_t* tiles, int32_t n_tiles,
                                   const float* position,
                                   int32_t max_positions, void* stream);

/* The F(4,3) layer as the LAST frame-rate layer in front of the per-word sum
 * of emphases.downsample (emphases/core.py:438-454; DOWNSAMPLE_METHOD 'sum' /
 * 'average'): the layer's output is never written.  Each 64-position tile forms
 * the running sum of its positions per channel (fixed order) and stores it only
 * where `slot_map` asks: slot_map int32 [ld frames] (16-byte aligned, indexed
 * by packed frame column) holds the row of `sums` a column reports to, or -1.
 * sums float32 [n_slots][ld_sums], ld_sums >= c_out, both multiples of 4.
 * emph_word_sums turns the rows into per-word sums.  Replaces
 * emph_conv1d_winograd4 + emph_segment_reduce for that layer (same values up
 * to summation order: a word's sum is a difference of running sums). */
int emph_conv1d_winograd4_word_sums(const float* x, int64_t ldx, float* sums,
                                    int64_t ld_sums, const float* pack,
                                    const float* bias, int32_t c_in,
                                    int32_t c_out, int32_t activation,
                                    const int32_t* tiles, int32_t n_tiles,
                                    const int32_t* slot_map, void* stream);

/* Up to emph_conv_stack_max_layers() (3) consecutive Conv1d(80, 80, 3, 'same')
 * + activation layers of the frame encoder (emphases/model/core.py:24-31,
 * 96-100; model/layers/convolution.py:25-37) in ONE launch: a workgroup owns
 * a span of up to 252 positions of one segment through all the layers, the
 * activations stay in LDS from layer to layer, the weights stream through an
 * LDS ring, one recomputed quad of halo on each side that continues inside
 * the segment.  Bit for bit the values of `layers` emph_conv1d_winograd4
 * launches (same arithmetic per output).
 *   packs      emph_conv_winograd4_pack of every layer, back to back
 *   biases     float32 [layers][80]
 *   relu_mask  bit l set: layer l is followed by ReLU (else identity)
 *   spans      int32 [n_spans][8], emph_conv_stack_spans (device copy)
 *   slot_map   NULL: y float32 [80][ldy] receives the last layer's output;
 *              else the last layer leaves running sums in y = sums[slot][ldy]
 *              like emph_conv1d_winograd4_word_sums, a running sum restarting
 *              at a span's first own position and every 64 computed positions
 * emph_conv_stack_spans cuts every segment (counts[i] positions at frame
 * column offsets[i]) into the fewest spans of even size (host arithmetic;
 * host_spans NULL: returns the number of spans). */
int32_t emph_conv_stack_max_layers(void);
int32_t emph_conv_stack_spans(const int64_t* host_counts,
                              const int64_t* host_offsets, int32_t n_segments,
                              int32_t* host_spans);
int emph_conv1d_stack(const float* x, int64_t ldx, float* y, int64_t ldy,
                      const float* packs, const float* biases, int32_t layers,
                      int32_t relu_mask, const int32_t* spans, int32_t n_spans,
                      const int32_t* slot_map, void* stream);

/* The same group of layers on the bf16 matrix pipe, direct form, every fp32
 * operand split into two bf16 pieces (three products per term, fp32
 * accumulation; emphases_amd/csrc/conv_split.hip): the opt-in
 * precision='bf16x3' of the host side - the reference runs these convolutions
 * under bf16 / fp16 autocast (emphases/core.py:594-607).  Same spans as
 * emph_conv1d_stack; `packs`: emph_conv_split_pack of every layer back to back
 * (emph_conv_split_pack_size() bytes each, device, 16-byte aligned); up to
 * FIVE layers per launch (the direct form spoils one position less per layer
 * than F(4,3)).  `slot_map` != NULL: the last layer leaves running sums in
 * y = sums[slot][ldy] for emph_word_sums, restarting at a span's first own
 * position and every 32 computed positions. */
int64_t emph_conv_split_pack_size(void);
int emph_conv_split_pack(const float* host_weight, void* host_pack);
int emph_conv1d_split(const float* x, int64_t ldx, float* y, int64_t ldy,
                      const void* packs, const float* biases, int32_t layers,
                      int32_t relu_mask, const int32_t* spans, int32_t n_spans,
                      const int32_t* slot_map, void* stream);

/* EXPERIMENTAL (measured in EXPERIMENTS.md, rounds 1-4 section 6, not used by the engine):
 * the F(4,3) layer as TWO independent launches ("halves"): half 0 computes the
 * output channels of the first ceil(m_tiles / 2) 16-channel tiles, half 1 the
 * rest, each around its own rows of the pack only (92 KB + 61 KB of LDS for
 * 80 x 80, four waves per workgroup).  Same values, bit for bit, as
 * emph_conv1d_winograd4[_position]; issued on two streams, the halves of
 * different layers / batches share a compute unit, so one's weight transfer and
 * store burst run under the other's matrix work.  `split_pack`: the pack cut
 * by emph_conv_winograd4_split_pack (same size as emph_conv_winograd4_pack's;
 * half 0's rows first); `position` may be NULL. */
int emph_conv_winograd4_split_pack(const float* host_weight, int32_t c_out,
                                   int32_t c_in, float* host_pack);
int emph_conv1d_winograd4_half(const float* x, int64_t ldx, float* y,
                               int64_t ldy, const float* split_pack,
                               const float* bias, int32_t c_in, int32_t c_out,
                               int32_t activation, const int32_t* tiles,
                               int32_t n_tiles, const float* position,
                               int32_t max_positions, int32_t half,
                               void* stream);

/* ------------------------------------------------------------------------ */
/* Frame -> word resampling                                                  */
/* ------------------------------------------------------------------------ */

/* out[:, word] = reduce(x[:, start:end]) for every word of every segment.
 *
 * Replaces the per-word Python loop of emphases.downsample (emphases/core.py:
 * 426-469): one slice + reduce + copy per word, plus a host sync.
 *
 *   x            float32 [channels, ldx]  frame axis
 *   bounds       int32 [2][ldw]           chunk-relative (start, end) frames on
 *                                         the word axis (row 0 starts, row 1 ends)
 *   out          float32 [channels, ldw]  word axis
 *   word_segment int32 [total_words]      segment of each word column, -1 for
 *                                         alignment padding columns
 * Empty words give 0 (sum) or NaN (average) like the reference; `end` beyond
 * the chunk is truncated like a Python slice.  The reference raises for an
 * empty word under 'max' and an out-of-range 'center'; callers check that on
 * the host (the kernel writes -inf / 0 there).
 */
int emph_segment_reduce(const float* x, int64_t ldx, const int32_t* bounds,
                        float* out, int64_t ldw, int32_t channels,
                        const int64_t* seg, const int32_t* word_segment,
                        int64_t total_words, int32_t mode, void* stream);

/* Per-word sums from the running sums emph_conv1d_winograd4_word_sums left in
 * `sums`: word column w = the signed terms terms[first[w] .. first[w + 1]) in
 * table order (s >= 0: + sums[s][:]; s < 0: - sums[~s][:]), divided by
 * lengths[w] for EMPH_REDUCE_AVERAGE (0 / 0 = NaN like torch.mean).
 *   first    int32 [columns + 1]
 *   lengths  int32 [columns]   frames the word covers inside its chunk (the
 *                              Python-slice clamp of core.py:446-454), -1 for
 *                              alignment padding columns (left untouched)
 *   out      float32 [channels][ldw]   what emph_segment_reduce writes
 * The tables are host arithmetic on the word bounds (emphases_amd/batch.py,
 * `Plan.word_sum_tables`). */
int emph_word_sums(const float* sums, int64_t ld_sums, const int32_t* terms,
                   const int32_t* first, const int32_t* lengths, float* out,
                   int64_t ldw, int32_t channels, int64_t columns, int32_t mode,
                   void* stream);

/* ------------------------------------------------------------------------ */
/* Output projection + postprocess                                           */
/* ------------------------------------------------------------------------ */

/* logits = conv1d(x, w[1, c, k], 'same') + b; scores = post(logits), per
 * column of the packed axis.  Replaces output_layer (emphases/model/core.py:
 * 33-37,138) and emphases.postprocess (emphases/core.py:335-342).  `weight`
 * is the plain torch layout [1][c][k] on the device; `logits` / `scores` may
 * be NULL; `position_segment` as `word_segment` above. */
int emph_output_layer(const float* x, int64_t ldx, const float* weight,
                      const float* bias, int32_t channels,
                      int32_t kernel_size, const int64_t* seg,
                      const int32_t* position_segment, int64_t total,
                      int32_t axis, int32_t post, float* logits,
                      float* scores, void* stream);

/* Word pieces for DOWNSAMPLE_LOCATION = 'input' (emphases/model/core.py:41-87
 * with emphases/core.py:552-586): every word becomes its own zero-padded
 * sequence.  Piece p copies `length` columns of every row of x starting at
 * column `source` to columns target .. target + length of y and zeroes columns
 * target + length .. target + padded.
 *   pieces  int64 [n_pieces][4] = (source, length, target, padded) */
int emph_gather_columns(const float* x, int64_t ldx, float* y, int64_t ldy,
                        int32_t channels, const int64_t* pieces,
                        int32_t n_pieces, void* stream);

/* ------------------------------------------------------------------------ */
/* Fused word stage of the convolutional model                               */
/* ------------------------------------------------------------------------ */

/* Output words per workgroup of emph_word_decoder for a decoder of `layers`
 * convolutions of `kernel_size` followed by an output convolution of
 * `out_kernel_size`: 64 minus the receptive-field halo on both sides (what a
 * workgroup computes of a segment longer than 64 words). */
int32_t emph_word_decoder_block(int32_t layers, int32_t kernel_size,
                                int32_t out_kernel_size);

/* The word-axis tile table emph_word_decoder takes (HOST arrays in, host table
 * out): rows (segment, first word, segment's first column, segment's words).
 * A segment of at most 32 words is one tile (a workgroup then computes two
 * 16-word MFMA tiles per layer instead of four); 33 .. 2 (32 - halo) words are
 * TWO tiles, first = 0 and ceil(words / 2), each a 32-position window at one end
 * of the segment with the other half's nearest words as halo; up to 64 words one
 * tile; longer segments one tile per emph_word_decoder_block words.
 *   host_counts, host_offsets  int64 [segments]  words and first column per segment
 *   host_tiles                 int32 [n][4] or NULL (count only)
 * Returns the number of tiles, or -1 for bad arguments. */
int32_t emph_word_decoder_tiles(const int64_t* host_counts,
                                const int64_t* host_offsets, int32_t segments,
                                int32_t layers, int32_t kernel_size,
                                int32_t out_kernel_size, int32_t* host_tiles);

/* Host-side repack of one decoder layer's Conv1d weight [channels][channels]
 * [kernel_size] for emph_word_decoder: channels^2 * kernel_size floats, ordered
 * so that a lane reads the weights of sixteen input channels x kernel_size taps
 * as kernel_size 16-byte LDS reads. */
int64_t emph_word_decoder_pack_size(int32_t channels, int32_t kernel_size);
int emph_word_decoder_pack(const float* host_weight, int32_t channels,
                           int32_t kernel_size, float* host_pack);

/* `layers` x [Conv1d 'same' + activation] at word rate -> Conv1d(channels, 1)
 * -> postprocess, in one launch with the word activations resident in LDS and
 * the weights streamed through LDS by LDS-DMA.
 *
 * Replaces word_decoder (emphases/model/core.py:105-107; model/layers/
 * convolution.py:25-30), output_layer (model/core.py:33-37,138) and
 * emphases.postprocess (core.py:335-342).  `layers` = 0 gives the
 * DOWNSAMPLE_LOCATION 'inference' / 'loss' models (model/core.py:109-130).
 *
 *   x           float32 [channels, ldx]  word embeddings (word axis), e.g. the
 *                                        output of emph_segment_reduce
 *   tiles       int32 [n_tiles][4]       the table of emph_word_decoder_tiles
 *                                        (device copy)
 *   packs       float32                  emph_word_decoder_pack of every
 *                                        decoder layer, back to back
 *   biases      float32 [layers][channels]
 *   out_weight  float32 [1][channels][out_kernel_size], out_bias float32 [1]
 *   logits, scores  float32 [ldx] (either may be NULL)
 * channels must be a multiple of 16, at most 128.
 */
int emph_word_decoder(const float* x, int64_t ldx, const int32_t* tiles,
                      int32_t n_tiles, int32_t channels, const float* packs,
                      const float* biases, int32_t layers, int32_t kernel_size,
                      int32_t activation, const float* out_weight,
                      const float* out_bias, int32_t out_kernel_size,
                      int32_t post, float* logits, float* scores, void* stream);

/* ------------------------------------------------------------------------ */
/* Transformer blocks (emphases/model/layers/transformer.py:13-52)           */
/* ------------------------------------------------------------------------ */

/* x[c, pos] += table[pos_in_segment][c]  (PositionalEncoding.forward,
 * transformer.py:51-52; dropout is the identity at inference).  `table` is
 * float32 [max_positions][channels]; tiles are blocks of `tile_n` positions. */
int emph_add_position(float* x, int64_t ldx, const float* table,
                      int32_t channels, int32_t max_positions,
                      const int32_t* tiles, int32_t n_tiles, int32_t tile_n,
                      void* stream);

/* Multi-head self-attention core: softmax(Q K^T / sqrt(d)) V per segment and
 * head, never materialising the score matrix.  Replaces the attention inside
 * nn.MultiheadAttention (transformer.py:18-23; key-padding mask all-false at
 * inference, core.py:321-328).
 *
 *   qk   float32 [2*channels, ld]   rows 0..c-1 = Q, c..2c-1 = K
 *   v    float32 [ld, channels]     position-major V
 *   out  float32 [channels, ld]
 *   tiles int32 [n_tiles][4]        tile table, block = tile_n queries
 *   tile_n 64: one wave per tile, keys / values read from L2 by every wave
 *              (short segments: the word axis);
 *          256: a workgroup of eight waves per tile with the key / value blocks
 *              staged ONCE per workgroup in LDS (long segments: the frame axis);
 *              workgroups walk the (head, tile) space in XCD order, so that
 *              the tiles of a segment share one XCD's L2;
 *          512 (EXPERIMENTAL: measured, not used by the engine, DESIGN.md
 *          section 4): the same with sixteen waves (keys / values staged once per 512
 *              queries; one workgroup per CU: 1 % faster alone, 1 % slower
 *              with two batches in flight - the engine uses 256)
 *   key_counts int32 [n_seg] or NULL  src_key_padding_mask (transformer.py:
 *                                   26-29) as the number of leading positions
 *                                   of each segment that are real keys; the
 *                                   rest (zero padding of the word pieces of
 *                                   DOWNSAMPLE_LOCATION 'input', model/core.py:
 *                                   41-87) is hidden as keys but still computed
 *                                   as queries.  NULL: every position is a key.
 * Head dimension (channels / heads) must be 32, 40 or 64.
 */
int emph_attention(const float* qk, const float* v, float* out, int64_t ld,
                   int32_t channels, int32_t heads, const int32_t* tiles,
                   int32_t n_tiles, int32_t tile_n, const int32_t* key_counts,
                   void* stream);

/* The same attention for segments of >= 128 positions (tile_n 256: the tile
 * table of emph_attention's grouped kernel) on the bf16 matrix pipe, every
 * fp32 operand split into `pieces` bf16 pieces (emphases_amd/csrc/
 * attention_split.hip): pieces = 2 keeps three products per term ("bf16x3",
 * each product within 2^-16), pieces = 3 six ("bf16x6": the pieces represent
 * the fp32 operands exactly and what is dropped is below one fp32 rounding).
 * fp32 in, fp32 accumulation, fp32 out; head dimension 40.  An opt-in of the
 * host side (`precision=`): the reference's own inference path runs its
 * matmuls under bf16 / fp16 autocast (emphases/core.py:594-607).
 *
 * Two calls per layer: emph_split_kv splits this layer's keys and values once
 * (tile_n 64 tile table of the frame axis; `images` = scratch of
 * emph_split_kv_bytes(ld, n_segments, ...) bytes, 16-byte aligned), then
 * emph_attention_split reads Q from `qk` and the keys / values from `images`. */
int64_t emph_split_kv_bytes(int64_t ld, int32_t n_segments, int32_t channels,
                            int32_t heads, int32_t pieces);
int emph_split_kv(const float* qk, const float* v, int64_t ld, int32_t channels,
                  int32_t heads, const int32_t* tiles, int32_t n_tiles,
                  int32_t tile_n, int32_t pieces, void* images, void* stream);
int emph_attention_split(const float* qk, const void* images, float* out,
                         int64_t ld, int32_t channels, int32_t heads,
                         const int32_t* tiles, int32_t n_tiles, int32_t tile_n,
                         const int32_t* key_counts, int32_t pieces, void* stream);

/* y = LayerNorm(x + r) over channels for columns [first_column,
 * first_column + columns) (post-LN residual of nn.TransformerEncoderLayer,
 * transformer.py:18-23; eps 1e-5).  channels <= 128. */
int emph_add_layernorm(const float* x, const float* r, float* y, int64_t ld,
                       int32_t channels, const float* gamma,
                       const float* beta, float eps, int64_t first_column,
                       int64_t columns, void* stream);

/* The position-wise half of a post-LN Transformer encoder layer in one launch
 * (nn.TransformerEncoderLayer, emphases/model/layers/transformer.py:18-23):
 *     y = LayerNorm1(x + W_o attended + b_o)
 *     x <- LayerNorm2(y + W_2 act(W_1 y + b_1) + b_2)
 * for square layers (dim_feedforward == channels, as the reference builds them:
 * transformer.py:20) of 64 or 80 channels (the three packs share the LDS); the
 * three GEMMs chain through registers.
 *   attended float32 [channels, ld]  output of emph_attention
 *   x        float32 [channels, ld]  residual stream, updated in place
 *   packs    three emph_linear_chain_pack images back to back: out_proj
 *            (natural = 1), linear1 (natural = 0), linear2 (natural = 0)
 *   vectors  float32 [7][channels]: b_o, gamma1, beta1, b_1, b_2, gamma2, beta2
 *   activation  EMPH_ACT_RELU or EMPH_ACT_NONE
 *   tiles    tile table of the walked axis, block = tile_n (16 or 32) */
int64_t emph_linear_chain_pack_size(int32_t channels);
int emph_linear_chain_pack(const float* host_weight, int32_t channels,
                           int32_t natural, float* host_pack);
int emph_transformer_block(const float* attended, float* x, int64_t ld,
                           int32_t channels, const float* packs,
                           const float* vectors, float eps, int32_t activation,
                           const int32_t* tiles, int32_t n_tiles,
                           int32_t tile_n, void* stream);

/* ... and the NEXT layer's Q, K, V projections in the same launch (the layer's
 * output is still in registers, already in the operand layout of a chain-packed
 * GEMM: no second pass over x, one launch less per layer).  `packs` = out |
 * linear1 | linear2 | W_q | W_k | W_v, the last three emph_linear_chain_pack(
 * natural = 0) images of the next layer's in_proj; `vectors` float32 [10][channels]:
 * the seven above, then b_q, b_k, b_v; qk / v as emph_qkv_projection writes them. */
int emph_transformer_block_qkv(const float* attended, float* x, int64_t ld,
                               int32_t channels, const float* packs,
                               const float* vectors, float eps,
                               int32_t activation, const int32_t* tiles,
                               int32_t n_tiles, int32_t tile_n, float* qk,
                               float* v, void* stream);

/* Q, K, V projections of self-attention in one launch (in_proj of
 * nn.MultiheadAttention, transformer.py:18-23) in the layouts emph_attention
 * takes: qk float32 [2*channels, ld] (rows 0..c-1 = Q), v float32 [ld, channels]
 * position-major.  packs = three emph_linear_chain_pack(natural = 1) images
 * (W_q, W_k, W_v), bias float32 [3][channels]; channels 64 or 80; tile_n 16/32. */
int emph_qkv_projection(const float* x, int64_t ld, float* qk, float* v,
                        int32_t channels, const float* packs,
                        const float* bias, const int32_t* tiles,
                        int32_t n_tiles, int32_t tile_n, void* stream);

/* emph_transformer_block and emph_qkv_projection on the bf16 matrix pipe
 * (v_mfma_f32_32x32x16_bf16), every fp32 operand split into `pieces` bf16 pieces:
 *   pieces 2  hi.hi + hi.lo + lo.hi, pieces rounded to nearest: 2^-16 per product
 *   pieces 3  six products of exact (truncated) pieces: one fp32 rounding per product
 * with fp32 accumulation; LayerNorm, bias, residual and ReLU in fp32.  The opt-in
 * precisions 'bf16x3' / 'bf16x6' of the engine, never its default.  80 channels,
 * tiles of 32 positions; the same transformer.py:18-23 arithmetic.
 *   packs (block)  emph_linear_split_pack of out_proj.weight | linear1.weight |
 *                  linear2.weight, back to back (device, 16-byte aligned)
 *   packs (qkv)    emph_linear_split_pack of the q | k | v rows of in_proj_weight
 *   vectors, bias, attended, x, qk, v, tiles: as the fp32 entries above */
int64_t emph_linear_split_pack_size(int32_t pieces);
int emph_linear_split_pack(const float* host_weight /* [80][80] */, int32_t pieces,
                           void* host_pack);
int emph_transformer_block_split(const float* attended, float* x, int64_t ld,
                                 int32_t channels, const void* packs,
                                 int32_t pieces, const float* vectors, float eps,
                                 int32_t activation, const int32_t* tiles,
                                 int32_t n_tiles, int32_t tile_n, void* stream);
int emph_qkv_projection_split(const float* x, int64_t ld, float* qk, float* v,
                              int32_t channels, const void* packs,
                              int32_t pieces, const float* bias,
                              const int32_t* tiles, int32_t n_tiles,
                              int32_t tile_n, void* stream);
/* emph_transformer_block_split and the NEXT layer's emph_qkv_projection_split
 * (images == NULL: qk and v) or emph_qkv_projection_split_images (images != NULL: Q
 * into qk's first 80 rows and the K / V images for `attention_pieces`, v unused) as
 * ONE launch - the split counterpart of emph_transformer_block_qkv.  The layer's
 * output is split in the registers it is normalised in: x is written (the next
 * block's residual) but not read again; the six weight packs stream through a ring
 * of three LDS slots.  Results are bit for bit those of the two entries in a row.
 * Replaces the tail of one nn.TransformerEncoderLayer and the in_proj of the next
 * (emphases/model/layers/transformer.py:18-30).
 *   block_packs / vectors   as emph_transformer_block_split
 *   qkv_packs / qkv_bias    the next layer's, as emph_qkv_projection_split */
int emph_transformer_block_qkv_split(const float* attended, float* x, int64_t ld,
                                     int32_t channels, int32_t heads,
                                     const void* block_packs, const void* qkv_packs,
                                     int32_t pieces, int32_t attention_pieces,
                                     const float* vectors, const float* qkv_bias,
                                     float eps, int32_t activation,
                                     const int32_t* tiles, int32_t n_tiles,
                                     int32_t tile_n, float* qk, float* v,
                                     void* images, void* stream);
/* The same three operations on tiles of SIXTEEN positions (csrc/block_split16.hip:
 * v_mfma_f32_16x16x32_bf16, 230 registers a wave, two waves a SIMD - one tile's
 * vector work under the other's MFMAs; eight waves share the packs in LDS): what the
 * engine's opt-in precisions run.  One entry point, three shapes:
 *   attended != NULL, qkv_packs == NULL   emph_transformer_block_split
 *   attended == NULL, qkv_packs != NULL   emph_qkv_projection_split (images == NULL)
 *                                         / _images (images != NULL) of x
 *   both                                  emph_transformer_block_qkv_split: one launch,
 *                                         the six packs streamed through three LDS
 *                                         slots; bit for bit the two launches
 * Packs: emph_linear_split_pack16 (k-step j, group g, element e = input channel
 * 32 j + 16 (e / 4) + 4 g + e % 4: the order in which a GEMM's result lies in the
 * registers of the next); `tiles`: blocks of 16 (`tile_n` = 16); vectors, bias, qk, v,
 * images as above.  emphases/model/layers/transformer.py:18-30. */
int64_t emph_linear_split_pack16_size(int32_t pieces);
int emph_linear_split_pack16(const float* host_weight /* [80][80] */, int32_t pieces,
                             void* host_pack);
int emph_position_wise_split(const float* attended, float* x, int64_t ld,
                             int32_t channels, int32_t heads, const void* block_packs,
                             const float* vectors, const void* qkv_packs,
                             const float* qkv_bias, int32_t pieces,
                             int32_t attention_pieces, float eps, int32_t activation,
                             const int32_t* tiles, int32_t n_tiles, int32_t tile_n,
                             float* qk, float* v, void* images, void* stream);
/* ... with K and V written straight as the images emph_attention_split stages
 * (what emph_split_kv would make of qk / v: no fp32 K and V, no second pass): Q
 * into qk's first 80 rows (the K rows are left alone), `images` sized by
 * emph_split_kv_bytes for `attention_pieces` (2, 3 or 32); two heads.  `tiles`
 * (blocks of 32) must cover every segment whose images the attention will read. */
int emph_qkv_projection_split_images(const float* x, int64_t ld, float* qk,
                                     void* images, int32_t channels,
                                     int32_t heads, const void* packs,
                                     int32_t pieces, int32_t attention_pieces,
                                     const float* bias, const int32_t* tiles,
                                     int32_t n_tiles, int32_t tile_n,
                                     void* stream);

/* The word-rate Transformer decoder in ONE launch: positional encoding + all
 * `layers` post-LN encoder layers (transformer.py:13-52 as the word decoder,
 * emphases/model/core.py:26-30,105-107) for segments of at most 64 words (a
 * segment is one workgroup; the residual stream stays in registers between
 * layers).  Replaces emph_add_position + layers x (emph_qkv_projection +
 * emph_attention + emph_transformer_block) on the word axis.
 *
 *   x        float32 [channels, ld]   word embeddings, updated in place
 *   position float32 [max_positions][channels] sinusoidal table
 *   packs    `layers` images of emph_word_transformer_pack, back to back
 *   tiles    word-axis tile table with block 64; every segment must have at most
 *            64 words (the caller checks; longer segments take the three-kernel
 *            path)
 * channels 64 or 80, 2 heads, dim_feedforward = channels, ReLU. */
int64_t emph_word_transformer_pack_size(int32_t channels, int32_t heads);
int emph_word_transformer_pack(
    const float* in_proj_weight, const float* in_proj_bias,
    const float* out_weight, const float* out_bias,
    const float* linear1_weight, const float* linear1_bias,
    const float* linear2_weight, const float* linear2_bias,
    const float* norm1_weight, const float* norm1_bias,
    const float* norm2_weight, const float* norm2_bias, int32_t channels,
    int32_t heads, float* host_pack);
int emph_word_transformer(float* x, int64_t ld, const float* position,
                          int32_t max_positions, int32_t channels,
                          int32_t heads, const float* packs, int32_t layers,
                          float eps, const int32_t* tiles, int32_t n_tiles,
                          void* stream);

/* ------------------------------------------------------------------------ */
/* Evaluation metrics at word resolution                                     */
/* ------------------------------------------------------------------------ */

/* Accumulator fields of emph_word_metrics (float64 each) */
enum {
    EMPH_METRIC_COUNT = 0,           /* words seen                              */
    EMPH_METRIC_BCE = 1,             /* sum of per-word binary cross entropy    */
    EMPH_METRIC_SQUARED_ERROR = 2,   /* sum of (score - target)^2               */
    EMPH_METRIC_COVARIANCE = 3,      /* sum of (score - mean_p)(target - mean_t) */
    EMPH_METRIC_SUM_PREDICTED = 4,   /* sum / sum of squares of the scores ...  */
    EMPH_METRIC_SUMSQ_PREDICTED = 5,
    EMPH_METRIC_SUM_TARGET = 6,      /* ... and of the targets (Statistics)     */
    EMPH_METRIC_SUMSQ_TARGET = 7,
    EMPH_METRIC_FIELDS = 8
};

/* accumulators[i] += the masked sums over every word of the batch.
 *
 * Replaces Metrics.update of emphases/evaluate/metrics.py:27-46 (the
 * mask_from_lengths + boolean indexing, BinaryCrossEntropy.update 59-76,
 * MeanSquaredError.update 80-92, torchutil's PearsonCorrelation.update) and the
 * sums behind Statistics (metrics.py:101-110), for one packed batch.
 *
 *   logits, targets  float32 [total]  packed word axis (emph_word_decoder's
 *                                     `logits`; targets laid out the same way)
 *   word_segment     int32 [total]    >= 0 on real words, -1 on padding columns
 *   post             EMPH_POST_*      emphases.postprocess / the LOSS switch:
 *                                     SIGMOID = 'bce' (BCE from logits), CLAMP01 =
 *                                     'mse' (BCE from clamped probabilities)
 *   predicted_mean, target_mean       the dataset statistics the reference hands
 *                                     to PearsonCorrelation (metrics.py:15-17)
 *   accumulators     float64 [EMPH_METRIC_FIELDS], zeroed by the caller before
 *                                     the first batch
 */
int emph_word_metrics(const float* logits, const float* targets,
                      const int32_t* word_segment, int64_t total, int32_t post,
                      float predicted_mean, float target_mean,
                      double* accumulators, void* stream);

/* ------------------------------------------------------------------------ */
/* The whole convolutional path in one call                                  */
/* ------------------------------------------------------------------------ */

/* Device-side description of a convolutional model (every pointer is device
 * memory prepared once per checkpoint). */
typedef struct emph_conv_model {
    int32_t channels;             /* multiple of 16, <= 128                    */
    int32_t features;             /* input rows: the 80 mel rows               */
    int32_t encoder_layers;       /* frame-rate layers after the input layer   */
    int32_t decoder_layers;       /* word-rate layers (0: no decoder)          */
    int32_t decoder_kernel_size;  /* also the output layer's kernel size       */
    int32_t activation;           /* EMPH_ACT_*                                */
    int32_t reduction;            /* EMPH_REDUCE_*                             */
    int32_t post;                 /* EMPH_POST_*                               */
    int32_t normalize;            /* emphases NORMALIZE                        */
    int32_t mel_nnz;
    int32_t conv_variant;         /* 0: Winograd F(2,3) packs, 1: F(4,3) packs */
    const float* table;           /* emph_frontend_table_fill                  */
    const int32_t* mel_start;
    const int32_t* mel_count;
    const int32_t* mel_offset;
    const float* mel_values;
    const float* input_pack;      /* emph_conv_winograd[4]_pack [C][features][3] */
    const float* input_bias;      /* [C]                                       */
    const float* encoder_packs;   /* encoder_layers packs [C][C][3], same variant */
    const float* encoder_biases;  /* [encoder_layers][C]                       */
    const float* decoder_packs;   /* emph_word_decoder_pack per layer          */
    const float* decoder_biases;  /* [decoder_layers][C]                       */
    const float* out_weight;      /* [1][C][decoder_kernel_size]               */
    const float* out_bias;        /* [1]                                       */
} emph_conv_model;

/* Tables of the fused per-word sum (emph_conv1d_winograd4_word_sums +
 * emph_word_sums), device pointers. */
typedef struct emph_word_sum_tables {
    const int32_t* slot_map;      /* [ld_frames]                               */
    const int32_t* terms;         /* signed slots                              */
    const int32_t* first;         /* [ld_words + 1]                            */
    const int32_t* lengths;       /* [ld_words]                                */
    int32_t n_slots;
} emph_word_sum_tables;

/* Floats of scratch emph_prominence_forward needs (features, two activation
 * buffers on the frame axis, the word embeddings, `n_slots` rows of running
 * sums). */
int64_t emph_prominence_workspace_floats(int32_t features, int32_t channels,
                                         int64_t ld_frames, int64_t ld_words,
                                         int32_t n_slots);

/* log-mel -> input conv -> encoder convs -> per-word reduce -> word decoder ->
 * scores for a whole ragged batch: emphases.infer + emphases.postprocess
 * (emphases/core.py:295-342) over emphases.Model.forward (emphases/model/
 * core.py:89-138) of the convolutional configurations with encoder
 * kernel_size 3 and mel features.  The tables are those of the individual
 * entry points: `frontend_tiles` with blocks of emph_frontend_block() frames, `frame_tiles` with
 * block `tile_n` (32 or 64; 64 for conv_variant 1), `word_tiles` from
 * emph_word_decoder_tiles(...).  `word_sums` (conv_variant 1, reduction sum /
 * average; else NULL): the last encoder layer leaves running sums instead of
 * its output and emph_word_sums replaces emph_segment_reduce.  `conv_spans`
 * (emph_conv_stack_spans, device copy; else NULL): the frame-rate layers run
 * as groups of up to three layers per launch (emph_conv1d_stack) when the model
 * is the 80 -> 80 family with `input_pack` / `encoder_packs` and `input_bias`
 * / `encoder_biases` back to back in memory; `word_sums` must then follow the
 * spans' restarts.  Enqueues on `stream`; allocates nothing. */
int emph_prominence_forward(const emph_conv_model* model, const void* audio,
                            int32_t audio_format, const int64_t* seg,
                            const int32_t* frontend_tiles,
                            int32_t n_frontend_tiles,
                            const int32_t* frame_tiles, int32_t n_frame_tiles,
                            int32_t tile_n, const int32_t* word_tiles,
                            int32_t n_word_tiles, const int32_t* bounds,
                            const int32_t* word_segment, int64_t ld_frames,
                            int64_t ld_words, float* workspace, float* logits,
                            float* scores,
                            const emph_word_sum_tables* word_sums,
                            const int32_t* conv_spans, int32_t n_conv_spans,
                            void* stream);

/* ------------------------------------------------------------------------ */
/* Measurement                                                               */
/* ------------------------------------------------------------------------ */

/* Enqueues one empty kernel (one wave, no memory access) on `stream`.  A pair
 * of HIP events around a launch brackets the command processor's dispatch of
 * the kernel as well as the kernel; the same pair around this launch is that
 * overhead alone, which `bench.py` subtracts from its live per-kernel
 * durations so that they can be held against rocprofv3's (which timestamps
 * the kernel itself).  No reference counterpart: the reference measures
 * nothing per kernel. */
int emph_launch_probe(void* stream);

/* Kernel-exact launch timer of the CALLING thread (measurement only; not to be
 * armed while a stream is being captured).  Between `begin` and `end` every
 * kernel this library launches from the thread carries its own pair of events
 * bound to the kernel's dispatch packet (hipExtLaunchKernel): their distance is
 * the kernel's own begin -> end, the timestamps rocprofv3's kernel trace reads,
 * without the command processor's dispatch that a pair of RECORDED events
 * brackets as well.  `emph_launch_timer_count` = launches since `begin` (-1:
 * no timer armed); `emph_launch_timer_end` waits for the timed launches and
 * writes their durations in microseconds, in launch order, to
 * `host_microseconds[0 .. min(count, capacity))`, the number of launches seen
 * to `host_count`, and disarms.  Launches beyond `capacity` run untimed.  This
 * is what `bench.py`'s `roofline.avg_launch_us` is measured with.  No
 * reference counterpart. */
int emph_launch_timer_begin(int32_t capacity);
int32_t emph_launch_timer_count(void);
int emph_launch_timer_end(float* host_microseconds, int32_t capacity, int32_t* host_count);

#ifdef __cplusplus
}
#endif
#endif /* EMPHASES_HIP_H */
