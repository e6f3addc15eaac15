// The whole convolutional path behind one C call (SURVEY.md 8b's optional
// `prominence_forward`): log-mel -> input conv -> encoder convs -> per-word
// reduce -> word decoder -> scores, i.e. emphases/core.py:295-342 over
// emphases/model/core.py:89-138 for the default configuration family
// (ARCHITECTURE 'convolution', DOWNSAMPLE_LOCATION 'intermediate' /
// 'inference' / 'loss', mel features, encoder kernel_size 3).
//
// Host code only: it enqueues the library's own kernels on `stream` in order,
// allocates nothing and never synchronises, so it is hipGraph-capturable like
// the entry points it calls.
#include "common.h"

using namespace emph;

namespace emph {
__global__ __launch_bounds__(64) void launch_probe_kernel() {}
}  // namespace emph

extern "C" {

int emph_launch_probe(void* stream) {
    EMPH_LAUNCH(launch_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream));
    return check_launch("emph_launch_probe");
}

int64_t emph_prominence_workspace_floats(int32_t features, int32_t channels,
                                         int64_t ld_frames, int64_t ld_words,
                                         int32_t n_slots) {
    return (static_cast<int64_t>(features) + 2 * channels) * ld_frames +
           static_cast<int64_t>(channels) * (ld_words + n_slots);
}

int emph_prominence_forward(const emph_conv_model* model, const void* audio,
                            int32_t audio_format, const int64_t* seg,
                            const int32_t* frontend_tiles,
                            int32_t n_frontend_tiles, const int32_t* frame_tiles,
                            int32_t n_frame_tiles, int32_t tile_n,
                            const int32_t* word_tiles, int32_t n_word_tiles,
                            const int32_t* bounds, const int32_t* word_segment,
                            int64_t ld_frames, int64_t ld_words, float* workspace,
                            float* logits, float* scores,
                            const emph_word_sum_tables* word_sums,
                            const int32_t* conv_spans, int32_t n_conv_spans, void* stream) {
    EMPH_REQUIRE(model != nullptr, EMPH_EINVAL, "emph_prominence_forward: model is null");
    const emph_conv_model& m = *model;
    EMPH_REQUIRE(audio && seg && workspace, EMPH_EINVAL,
                 "emph_prominence_forward: null pointer");
    EMPH_REQUIRE(m.features == kMels, EMPH_ERANGE,
                 "emph_prominence_forward: %d feature rows (the fused path takes the 80 mel "
                 "rows only)", m.features);
    EMPH_REQUIRE(m.encoder_layers >= 0 && m.encoder_layers <= 16, EMPH_ERANGE,
                 "emph_prominence_forward: %d encoder layers", m.encoder_layers);
    const bool quad = m.conv_variant == 1;        // Winograd F(4,3)
    EMPH_REQUIRE(m.conv_variant == 0 || quad, EMPH_EINVAL,
                 "emph_prominence_forward: conv_variant %d", m.conv_variant);
    EMPH_REQUIRE(!quad || tile_n == 64, EMPH_ERANGE,
                 "emph_prominence_forward: F(4,3) takes 64-position tiles, not %d", tile_n);
    const int64_t lds_square = quad ? emph_conv_winograd4_lds_bytes(m.channels, m.channels)
                                    : emph_conv_winograd_lds_bytes(m.channels, m.channels);
    const int64_t lds_input = quad ? emph_conv_winograd4_lds_bytes(m.channels, m.features)
                                   : emph_conv_winograd_lds_bytes(m.channels, m.features);
    EMPH_REQUIRE(lds_square <= 160 * 1024 && lds_input <= 160 * 1024, EMPH_ERANGE,
                 "emph_prominence_forward: %d channels do not fit the Winograd kernel's LDS",
                 m.channels);
    const int c = m.channels;
    float* features = workspace;
    float* current = features + static_cast<int64_t>(m.features) * ld_frames;
    float* other = current + static_cast<int64_t>(c) * ld_frames;
    float* words = other + static_cast<int64_t>(c) * ld_frames;
    float* sums = words + static_cast<int64_t>(c) * ld_words;
    const bool fold = word_sums != nullptr && m.encoder_layers > 0;
    EMPH_REQUIRE(!fold || (quad && (m.reduction == EMPH_REDUCE_SUM ||
                                    m.reduction == EMPH_REDUCE_AVERAGE)),
                 EMPH_EINVAL,
                 "emph_prominence_forward: word_sums needs conv_variant 1 and a sum / average "
                 "reduction");

    int status = emph_logmel(audio, audio_format, seg, frontend_tiles, n_frontend_tiles, m.table,
                             m.mel_start, m.mel_count, m.mel_offset, m.mel_values, m.mel_nnz,
                             features, ld_frames, 0, -1, nullptr, nullptr, m.normalize, stream);
    if (status) return status;
    auto conv = [&](const float* in, float* out, const float* pack, const float* bias, int c_in,
                    int activation) {
        return quad ? emph_conv1d_winograd4(in, ld_frames, out, ld_frames, pack, bias, c_in, c,
                                            activation, frame_tiles, n_frame_tiles, stream)
                    : emph_conv1d_winograd(in, ld_frames, out, ld_frames, pack, bias, c_in, c,
                                           activation, frame_tiles, n_frame_tiles, tile_n,
                                           stream);
    };
    const int64_t pack_floats =
        quad ? emph_conv_winograd4_pack_size(c, c) : emph_conv_winograd_pack_size(c, c);
    // The frame-rate layers as a few launches of several layers each
    // (emph_conv1d_stack: activations resident in LDS from layer to layer) when the
    // caller brought the span table and the model is the 80 -> 80 family with its
    // packs and biases back to back (input layer first).
    const bool stack = conv_spans != nullptr && quad && c == 80 && m.features == 80 &&
                       m.encoder_packs == m.input_pack + pack_floats &&
                       m.encoder_biases == m.input_bias + c &&
                       (m.activation == EMPH_ACT_RELU || m.activation == EMPH_ACT_NONE);
    if (stack) {
        const int total = 1 + m.encoder_layers;
        const int most = emph_conv_stack_max_layers();
        const int groups = (total + most - 1) / most;
        float* buffers[2] = {current, other};
        const float* in = features;
        int done = 0;
        for (int group = 0; group < groups; ++group) {
            // as even as possible, the larger groups first
            const int size = (total - done + (groups - group) - 1) / (groups - group);
            int relu = 0;
            for (int l = 0; l < size; ++l)
                if (done + l >= 1 && m.activation == EMPH_ACT_RELU) relu |= 1 << l;
            // the closing group leaves running sums when the per-word sum is folded
            const bool to_sums = group == groups - 1 && fold;
            float* out = to_sums ? sums : buffers[group & 1];
            status = emph_conv1d_stack(in, ld_frames, out, to_sums ? c : ld_frames,
                                       m.input_pack + done * pack_floats,
                                       m.input_bias + static_cast<int64_t>(done) * c, size, relu,
                                       conv_spans, n_conv_spans,
                                       to_sums ? word_sums->slot_map : nullptr, stream);
            if (status) return status;
            in = out;
            done += size;
        }
        current = const_cast<float*>(in);        // (the encoder's output when not folded)
        status = fold ? emph_word_sums(sums, c, word_sums->terms, word_sums->first,
                                       word_sums->lengths, words, ld_words, c, ld_words,
                                       m.reduction, stream)
                      : emph_segment_reduce(current, ld_frames, bounds, words, ld_words, c, seg,
                                            word_segment, ld_words, m.reduction, stream);
        if (status) return status;
        return emph_word_decoder(words, ld_words, word_tiles, n_word_tiles, c, m.decoder_packs,
                                 m.decoder_biases, m.decoder_layers, m.decoder_kernel_size,
                                 m.activation, m.out_weight, m.out_bias, m.decoder_kernel_size,
                                 m.post, logits, scores, stream);
    }
    status = conv(features, current, m.input_pack, m.input_bias, m.features, EMPH_ACT_NONE);
    if (status) return status;
    for (int layer = 0; layer < m.encoder_layers; ++layer) {
        if (fold && layer == m.encoder_layers - 1) {
            // the last frame-rate layer: running sums at the marked frames only
            status = emph_conv1d_winograd4_word_sums(
                current, ld_frames, sums, c, m.encoder_packs + layer * pack_floats,
                m.encoder_biases + static_cast<int64_t>(layer) * c, c, c, m.activation,
                frame_tiles, n_frame_tiles, word_sums->slot_map, stream);
            if (status) return status;
            break;
        }
        status = conv(current, other, m.encoder_packs + layer * pack_floats,
                      m.encoder_biases + static_cast<int64_t>(layer) * c, c, m.activation);
        if (status) return status;
        float* swap = current;
        current = other;
        other = swap;
    }
    status = fold ? emph_word_sums(sums, c, word_sums->terms, word_sums->first,
                                   word_sums->lengths, words, ld_words, c, ld_words,
                                   m.reduction, stream)
                  : emph_segment_reduce(current, ld_frames, bounds, words, ld_words, c, seg,
                                        word_segment, ld_words, m.reduction, stream);
    if (status) return status;
    return emph_word_decoder(words, ld_words, word_tiles, n_word_tiles, c, m.decoder_packs,
                             m.decoder_biases, m.decoder_layers, m.decoder_kernel_size,
                             m.activation, m.out_weight, m.out_bias, m.decoder_kernel_size,
                             m.post, logits, scores, stream);
}

}  // extern "C"
