// The file boundary of emphases.from_files_to_files (emphases/core.py:115-179) for a
// whole batch of files, on a pool of host threads (no device code):
//
//   pypar.Alignment(text_file)              core.py:49,107    Praat TextGrid -> words
//   emphases.load.audio(audio_file)         load.py:11-17     RIFF/WAVE -> samples
//   alignment.save(prefix.TextGrid)         core.py:111
//   torch.save(scores, prefix.pt)           core.py:112
//
// The reference does this one file at a time on the Python thread; the device path
// behind it takes a few microseconds per utterance, so at corpus scale the files,
// not the kernels, set the rate.  Here the TextGrids of a batch are parsed and the
// WAVE headers walked in parallel, the samples are read straight into the
// (pinned) staging buffer of the batch - no intermediate bytes object, no
// Python per file - and the outputs are written in parallel.
//
// This is the same grammar and the same walker as emphases_amd/alignment.py and
// emphases_amd/load.py (which stay the readers of the one-file API and of JSON
// alignments); tests/test_host.py holds the two against each other, byte for byte
// on the written files.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <charconv>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>

#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// A pool that runs `count` independent items of several callers' jobs at once
// (two Python threads pipeline consecutive batches through the library).
// ---------------------------------------------------------------------------

struct Task {
    std::function<void(int)> run;
    int count = 0;
    int helpers = 0;
    std::atomic<int> next{0};
    std::atomic<int> done{0};
    int inside = 0;                   // pool threads inside drain() (pool mutex)
};

class Workers {
  public:
    void grow(int threads) {
        std::lock_guard<std::mutex> lock(mutex_);
        while (static_cast<int>(threads_.size()) < threads) {
            threads_.emplace_back([this] { loop(); });
            if (pinned_)
                pthread_setaffinity_np(threads_.back().native_handle(), sizeof(cpus_), &cpus_);
        }
    }
    // the pool's own threads (the ones there are and the ones to come) onto `cpus`
    void pin(const cpu_set_t& cpus) {
        std::lock_guard<std::mutex> lock(mutex_);
        cpus_ = cpus;
        pinned_ = true;
        for (auto& thread : threads_)
            pthread_setaffinity_np(thread.native_handle(), sizeof(cpus_), &cpus_);
    }
    void run(Task& task) {
        if (task.count <= 0) return;
        if (task.helpers > 0) {
            grow(task.helpers);
            {
                std::lock_guard<std::mutex> lock(mutex_);
                queue_.push_back(&task);
            }
            wake_.notify_all();
        }
        drain(task);                                  // the caller works too
        if (task.helpers > 0) {
            std::unique_lock<std::mutex> lock(mutex_);
            for (auto it = queue_.begin(); it != queue_.end(); ++it)
                if (*it == &task) {
                    queue_.erase(it);
                    break;
                }
            finished_.wait(lock, [&] {
                return task.done.load() >= task.count && task.inside == 0;
            });
        }
    }

  private:
    static void drain(Task& task) {
        for (;;) {
            const int index = task.next.fetch_add(1);
            if (index >= task.count) return;
            task.run(index);
            task.done.fetch_add(1);
        }
    }
    void loop() {
        for (;;) {
            Task* task = nullptr;
            {
                std::unique_lock<std::mutex> lock(mutex_);
                wake_.wait(lock, [&] {
                    for (Task* candidate : queue_)
                        if (candidate->next.load() < candidate->count &&
                            candidate->inside < candidate->helpers)
                            return true;
                    return false;
                });
                for (Task* candidate : queue_)
                    if (candidate->next.load() < candidate->count &&
                        candidate->inside < candidate->helpers) {
                        task = candidate;
                        break;
                    }
                ++task->inside;
            }
            drain(*task);
            {
                std::lock_guard<std::mutex> lock(mutex_);
                --task->inside;
            }
            finished_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::deque<Task*> queue_;
    std::mutex mutex_;
    std::condition_variable wake_, finished_;
    cpu_set_t cpus_;
    bool pinned_ = false;
};

Workers* g_workers = new Workers;
// fork(): the child has the calling thread only; it builds its own pool
void forget_workers_in_child() { g_workers = new Workers; }
const int g_atfork = pthread_atfork(nullptr, nullptr, forget_workers_in_child);

void parallel(int count, int threads, std::function<void(int)> run) {
    Task task;
    task.run = std::move(run);
    task.count = count;
    task.helpers = count > 1 ? (threads - 1 < count - 1 ? threads - 1 : count - 1) : 0;
    if (task.helpers < 0) task.helpers = 0;
    g_workers->run(task);
}

// ---------------------------------------------------------------------------
// Files
// ---------------------------------------------------------------------------

bool read_whole(const char* path, std::string* data, std::string* error) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        *error = std::string(path) + ": " + strerror(errno);
        return false;
    }
    struct stat info;
    if (fstat(fd, &info) != 0) {
        *error = std::string(path) + ": " + strerror(errno);
        close(fd);
        return false;
    }
    data->resize(static_cast<size_t>(info.st_size));
    size_t got = 0;
    while (got < data->size()) {
        const ssize_t n = pread(fd, &(*data)[got], data->size() - got, static_cast<off_t>(got));
        if (n <= 0) break;
        got += static_cast<size_t>(n);
    }
    close(fd);
    data->resize(got);
    return true;
}

bool write_whole(const std::string& path, const std::string& data, std::string* error) {
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) {
        *error = path + ": " + strerror(errno);
        return false;
    }
    size_t put = 0;
    while (put < data.size()) {
        const ssize_t n = write(fd, data.data() + put, data.size() - put);
        if (n <= 0) {
            *error = path + ": " + strerror(errno);
            close(fd);
            return false;
        }
        put += static_cast<size_t>(n);
    }
    close(fd);
    return true;
}

// ---------------------------------------------------------------------------
// RIFF/WAVE headers: emphases_amd/load.py `_walk`, line for line
// ---------------------------------------------------------------------------

struct Wave {
    int64_t code = 0, channels = 0, rate = 0, bits = 0, offset = 0, bytes = 0;
};

uint32_t le32(const unsigned char* p) {
    return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24);
}
uint32_t le16(const unsigned char* p) { return p[0] | (p[1] << 8); }

bool walk_wave(const char* path, Wave* wave, std::string* error) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        *error = std::string(path) + ": " + strerror(errno);
        return false;
    }
    struct stat info;
    unsigned char head[12];
    if (fstat(fd, &info) != 0 || pread(fd, head, 12, 0) != 12 || memcmp(head, "RIFF", 4) ||
        memcmp(head + 8, "WAVE", 4)) {
        close(fd);
        *error = std::string(path) + " is not a RIFF/WAVE file";
        return false;
    }
    const int64_t end = info.st_size;
    int64_t cursor = 12;
    bool have_fmt = false, have_data = false;
    while (cursor + 8 <= end) {
        unsigned char header[8];
        if (pread(fd, header, 8, cursor) != 8) break;
        const int64_t size = le32(header + 4);
        if (!memcmp(header, "fmt ", 4)) {
            unsigned char body[40];
            const int64_t want = size < 40 ? size : 40;
            const ssize_t got = pread(fd, body, static_cast<size_t>(want), cursor + 8);
            if (got < 16) {
                close(fd);
                *error = std::string(path) + ": fmt chunk of " +
                         std::to_string(got < 0 ? 0 : got) + " bytes";
                return false;
            }
            wave->code = le16(body);
            wave->channels = le16(body + 2);
            wave->rate = le32(body + 4);
            wave->bits = le16(body + 14);
            if (wave->code == 0xFFFE && got >= 26) wave->code = le16(body + 24);
            have_fmt = true;
        } else if (!memcmp(header, "data", 4)) {
            wave->offset = cursor + 8;
            wave->bytes = size < end - cursor - 8 ? size : end - cursor - 8;
            have_data = true;
        }
        cursor += 8 + size + (size & 1);
    }
    close(fd);
    if (!have_fmt || !have_data) {
        *error = std::string(path) + " has no fmt/data chunk";
        return false;
    }
    const int64_t c = wave->code, b = wave->bits;
    const bool known = (c == 1 && (b == 8 || b == 16 || b == 24 || b == 32)) ||
                       (c == 3 && (b == 32 || b == 64));
    if (!known || wave->channels < 1) {
        *error = std::string(path) + ": unsupported WAVE format " + std::to_string(c) + "/" +
                 std::to_string(b) + " (" + std::to_string(wave->channels) + " channels)";
        return false;
    }
    return true;
}

// ---------------------------------------------------------------------------
// Praat TextGrid: emphases_amd/alignment.py (decode, _values, _tiers_from_textgrid,
// _words_from_textgrid, _fill_gaps, _textgrid)
// ---------------------------------------------------------------------------

const char kSilence[] = "<silent>";

void append_utf8(std::string* out, uint32_t code) {
    if (code < 0x80) {
        out->push_back(static_cast<char>(code));
    } else if (code < 0x800) {
        out->push_back(static_cast<char>(0xC0 | (code >> 6)));
        out->push_back(static_cast<char>(0x80 | (code & 0x3F)));
    } else if (code < 0x10000) {
        out->push_back(static_cast<char>(0xE0 | (code >> 12)));
        out->push_back(static_cast<char>(0x80 | ((code >> 6) & 0x3F)));
        out->push_back(static_cast<char>(0x80 | (code & 0x3F)));
    } else {
        out->push_back(static_cast<char>(0xF0 | (code >> 18)));
        out->push_back(static_cast<char>(0x80 | ((code >> 12) & 0x3F)));
        out->push_back(static_cast<char>(0x80 | ((code >> 6) & 0x3F)));
        out->push_back(static_cast<char>(0x80 | (code & 0x3F)));
    }
}

// (strict, like bytes.decode: an odd byte count, a lone surrogate -> false)
bool utf16_to_utf8(const unsigned char* data, size_t bytes, bool big, std::string* out) {
    if (bytes & 1) return false;
    out->reserve(bytes / 2);
    auto unit = [&](size_t i) -> uint32_t {
        return big ? (data[i] << 8) | data[i + 1] : data[i] | (data[i + 1] << 8);
    };
    for (size_t i = 0; i + 1 < bytes; i += 2) {
        uint32_t code = unit(i);
        if (code >= 0xDC00 && code < 0xE000) return false;
        if (code >= 0xD800 && code < 0xDC00) {
            if (i + 3 >= bytes) return false;
            const uint32_t low = unit(i + 2);
            if (low < 0xDC00 || low >= 0xE000) return false;
            code = 0x10000 + ((code - 0xD800) << 10) + (low - 0xDC00);
            i += 2;
        }
        append_utf8(out, code);
    }
    return true;
}

// what bytes.decode('utf-8') accepts: no overlong forms, no surrogates, <= U+10FFFF
bool valid_utf8(const unsigned char* d, size_t n) {
    size_t i = 0;
    while (i < n) {
        const unsigned char c = d[i];
        if (c < 0x80) {
            ++i;
            continue;
        }
        int extra;
        uint32_t code, least;
        if (c >= 0xC2 && c <= 0xDF) extra = 1, code = c & 0x1Fu, least = 0x80;
        else if (c >= 0xE0 && c <= 0xEF) extra = 2, code = c & 0x0Fu, least = 0x800;
        else if (c >= 0xF0 && c <= 0xF4) extra = 3, code = c & 0x07u, least = 0x10000;
        else return false;
        if (i + static_cast<size_t>(extra) >= n) return false;
        for (int k = 1; k <= extra; ++k) {
            if ((d[i + k] & 0xC0) != 0x80) return false;
            code = (code << 6) | (d[i + k] & 0x3Fu);
        }
        if (code < least || code > 0x10FFFF || (code >= 0xD800 && code < 0xE000)) return false;
        i += static_cast<size_t>(extra) + 1;
    }
    return true;
}

// alignment.decode: UTF-16 by BOM or by the zero bytes of the ASCII header,
// UTF-8 with or without BOM; false where Python's decoder raises
bool decode(const std::string& raw, std::string* text) {
    const unsigned char* d = reinterpret_cast<const unsigned char*>(raw.data());
    const size_t n = raw.size();
    if (n >= 2 && d[0] == 0xFF && d[1] == 0xFE) return utf16_to_utf8(d + 2, n - 2, false, text);
    if (n >= 2 && d[0] == 0xFE && d[1] == 0xFF) return utf16_to_utf8(d + 2, n - 2, true, text);
    if (n >= 3 && d[0] == 0xEF && d[1] == 0xBB && d[2] == 0xBF) {
        *text = raw.substr(3);
        return valid_utf8(d + 3, n - 3);
    }
    if (n >= 4 && d[1] == 0 && d[0] != 0) return utf16_to_utf8(d, n, false, text);
    if (n >= 4 && d[0] == 0 && d[1] != 0) return utf16_to_utf8(d, n, true, text);
    *text = raw;
    return valid_utf8(d, n);
}

struct Value {
    enum Kind { kString, kFlag, kNumber } kind;
    std::string text;
    double number = 0.;
};

bool is_word(unsigned char c) {
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') ||
           c == '_';
}
bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }

// alignment._VALUE, as a scanner: "strings" (a quote inside is doubled), <flags>,
// numbers that neither follow [\w.\[] nor run into [\w\]].  Returns false where this
// scanner cannot vouch for what the regex would do - a character beyond ASCII outside a
// string (Python's \w and \d know Unicode) - and the file goes to the Python reader.
bool scan_values(const std::string& text, std::vector<Value>* values) {
    const size_t n = text.size();
    size_t i = 0;
    while (i < n) {
        const unsigned char c = static_cast<unsigned char>(text[i]);
        if (c >= 0x80) return false;
        if (c == '"') {
            std::string body;
            size_t j = i + 1;
            bool closed = false;
            // the regex backtracks: a string that never closes ends at the FIRST
            // quote of the last doubled pair it swallowed, if there was one
            size_t last_pair = 0, body_at_pair = 0;
            bool pair = false;
            while (j < n) {
                if (text[j] == '"') {
                    if (j + 1 < n && text[j + 1] == '"') {
                        pair = true, last_pair = j, body_at_pair = body.size();
                        body.push_back('"');
                        j += 2;
                        continue;
                    }
                    closed = true;
                    break;
                }
                body.push_back(text[j]);
                ++j;
            }
            if (!closed && pair) {
                body.resize(body_at_pair);
                j = last_pair;
                closed = true;
            }
            if (closed) {
                values->push_back({Value::kString, std::move(body), 0.});
                i = j + 1;
                continue;
            }
            ++i;
            continue;
        }
        if (c == '<') {
            size_t j = i + 1;
            while (j < n && is_word(static_cast<unsigned char>(text[j]))) ++j;
            if (j < n && static_cast<unsigned char>(text[j]) >= 0x80) return false;
            if (j > i + 1 && j < n && text[j] == '>') {
                values->push_back({Value::kFlag, text.substr(i + 1, j - i - 1), 0.});
                i = j + 1;
                continue;
            }
            ++i;
            continue;
        }
        if (is_digit(c) || c == '-' || c == '+' || c == '.') {
            const unsigned char before = i ? static_cast<unsigned char>(text[i - 1]) : ' ';
            if (!(i && (is_word(before) || before == '.' || before == '['))) {
                size_t j = i;
                if (text[j] == '-' || text[j] == '+') ++j;
                size_t digits = j;
                while (j < n && is_digit(static_cast<unsigned char>(text[j]))) ++j;
                const size_t whole = j;                  // end of \d+
                bool mantissa = j > digits;
                if (mantissa) {                          // \d+\.?\d*
                    if (j < n && text[j] == '.') {
                        ++j;
                        while (j < n && is_digit(static_cast<unsigned char>(text[j]))) ++j;
                    }
                } else if (j < n && text[j] == '.') {    // \.\d+
                    size_t k = j + 1;
                    while (k < n && is_digit(static_cast<unsigned char>(text[k]))) ++k;
                    if (k > j + 1) {
                        mantissa = true;
                        j = k;
                    }
                }
                if (mantissa) {
                    if (j < n && (text[j] == 'e' || text[j] == 'E')) {
                        size_t k = j + 1;
                        if (k < n && (text[k] == '-' || text[k] == '+')) ++k;
                        size_t exponent = k;
                        while (k < n && is_digit(static_cast<unsigned char>(text[k]))) ++k;
                        if (k > exponent) j = k;
                    }
                    const unsigned char after = j < n ? static_cast<unsigned char>(text[j]) : ' ';
                    if (j < n && after >= 0x80) return false;
                    if (j < n && (is_word(after) || after == ']')) {
                        // the lookahead fails; the regex backtracks over the digits
                        // (each shorter end runs into a digit) down to ONE end that
                        // holds: the whole part in front of a '.'
                        if (whole > digits && whole < j && text[whole] == '.') j = whole;
                        else {
                            i = j;
                            continue;
                        }
                    }
                    Value value{Value::kNumber, std::string(), 0.};
                    // (std::from_chars: strtod follows the process's LC_NUMERIC - in a
                    // host application under a comma-decimal locale "0.5" would read
                    // as 0 - and Python's float(), the oracle, never does.  A value
                    // that does not fit a double - 1e999 - is not vouched for: the
                    // file goes to the Python reader, which raises as it always did)
                    size_t from = i;
                    if (text[from] == '+') ++from;
                    const auto parsed = std::from_chars(text.data() + from, text.data() + j,
                                                        value.number);
                    if (parsed.ec != std::errc() || parsed.ptr != text.data() + j ||
                        !std::isfinite(value.number))
                        return false;
                    values->push_back(std::move(value));
                    i = j;
                    continue;
                }
            }
        }
        ++i;
    }
    return true;
}

struct Item {
    double start = 0., end = 0.;
    std::string text;
    int word = -1;                    // phonemes: index of the word (after gap filling)
    bool filler = false;              // a silence inserted between two words
};

struct Grid {
    std::vector<Item> words;          // gap-free, silences named kSilence
    std::vector<Item> phones;
    std::string word_tier = "words", phone_tier = "phones";
    bool phones_first = false;
    bool has_phones = false;
};

// alignment._is_silence: `not text.strip()` (str.strip knows Unicode whitespace) or one of
// the two silence labels
bool is_silence(const std::string& text) {
    bool blank = true;
    const unsigned char* d = reinterpret_cast<const unsigned char*>(text.data());
    const size_t n = text.size();
    for (size_t i = 0; i < n && blank;) {
        uint32_t code = d[i];
        if (code < 0x80) {
            ++i;
        } else if (code < 0xE0 && i + 1 < n) {
            code = ((code & 0x1F) << 6) | (d[i + 1] & 0x3F);
            i += 2;
        } else if (code < 0xF0 && i + 2 < n) {
            code = ((code & 0x0F) << 12) | ((d[i + 1] & 0x3Fu) << 6) | (d[i + 2] & 0x3F);
            i += 3;
        } else {
            code = 0x10000;                       // nothing beyond the BMP is whitespace
            i += 4;
        }
        const bool space =
            code == ' ' || (code >= 9 && code <= 13) || (code >= 0x1c && code <= 0x1f) ||
            code == 0x85 || code == 0xA0 || code == 0x1680 || (code >= 0x2000 && code <= 0x200A) ||
            code == 0x2028 || code == 0x2029 || code == 0x202F || code == 0x205F || code == 0x3000;
        if (!space) blank = false;
    }
    return blank || text == "sp" || text == kSilence;
}

std::string lower(const std::string& text) {
    std::string out = text;
    for (char& c : out)
        if (c >= 'A' && c <= 'Z') c = static_cast<char>(c - 'A' + 'a');
    return out;
}

bool parse_grid(const std::string& text, Grid* grid, std::string* error) {
    std::vector<Value> values;
    if (!scan_values(text, &values)) {
        *error = "TextGrid: text beyond ASCII outside a string";
        return false;
    }
    size_t at = 0;
    bool short_of = false;
    auto take = [&](Value::Kind kind, const char* name) -> const Value* {
        if (at >= values.size()) {
            short_of = true;
            return nullptr;
        }
        if (values[at].kind != kind) {
            *error = std::string("TextGrid: expected ") + name;
            return nullptr;
        }
        return &values[at++];
    };
    auto fail = [&]() {
        if (short_of) *error = "TextGrid ends in the middle of a tier";
        return false;
    };
    const Value* v;
    if (!(v = take(Value::kString, "str")) || v->text != "ooTextFile") {
        if (v) *error = "not a TextGrid text file";
        return fail();
    }
    if (!(v = take(Value::kString, "str")) || v->text != "TextGrid") {
        if (v) *error = "not a TextGrid text file";
        return fail();
    }
    if (!take(Value::kNumber, "float") || !take(Value::kNumber, "float")) return fail();
    if (!(v = take(Value::kFlag, "tuple"))) return fail();
    struct Tier {
        std::string name;
        std::vector<Item> items;
    };
    std::vector<Tier> tiers;              // interval tiers only
    if (v->text == "exists") {
        if (!(v = take(Value::kNumber, "float"))) return fail();
        // (int(float) of Python: nan / inf raise, anything else truncates - a count no
        // int holds runs out of values, which the loop would only find out by walking)
        auto whole = [&](double number, int* out) {
            if (!(number > -2147483648. && number < 2147483647.)) {
                *error = "TextGrid: a count out of range";
                return false;
            }
            *out = static_cast<int>(number);
            return true;
        };
        int count = 0;
        if (!whole(v->number, &count)) return false;
        for (int t = 0; t < count; ++t) {
            const Value* kind = take(Value::kString, "str");
            if (!kind) return fail();
            const Value* name = take(Value::kString, "str");
            if (!name) return fail();
            if (!take(Value::kNumber, "float") || !take(Value::kNumber, "float")) return fail();
            const Value* size = take(Value::kNumber, "float");
            if (!size) return fail();
            int items = 0;
            if (!whole(size->number, &items)) return false;
            const bool interval = kind->text == "IntervalTier";
            Tier tier;
            tier.name = name->text;
            for (int k = 0; k < items; ++k) {
                Item item;
                const Value* a = take(Value::kNumber, "float");
                if (!a) return fail();
                item.start = item.end = a->number;
                if (interval) {
                    const Value* b = take(Value::kNumber, "float");
                    if (!b) return fail();
                    item.end = b->number;
                }
                const Value* label = take(Value::kString, "str");
                if (!label) return fail();
                item.text = label->text;
                if (interval) tier.items.push_back(std::move(item));
            }
            if (interval) tiers.push_back(std::move(tier));
        }
    }
    if (tiers.empty()) {
        *error = "TextGrid holds no interval tiers";
        return false;
    }
    // which tier holds the words, which the phonemes (alignment._words_from_textgrid)
    const int count = static_cast<int>(tiers.size());
    int word_index = -1;
    for (int i = 0; i < count && word_index < 0; ++i) {
        const std::string name = lower(tiers[i].name);
        if (name == "words" || name == "word") word_index = i;
    }
    if (word_index < 0) {
        word_index = 0;
        for (int i = 1; i < count; ++i)
            if (tiers[i].items.size() < tiers[word_index].items.size()) word_index = i;
    }
    int phone_index = -1;
    for (int i = 0; i < count && phone_index < 0; ++i) {
        if (i == word_index) continue;
        const std::string name = lower(tiers[i].name);
        if (name == "phones" || name == "phone" || name == "phonemes" || name == "phoneme")
            phone_index = i;
    }
    if (phone_index < 0 && count > 1) {
        int finest = -1;
        for (int i = 0; i < count; ++i) {
            if (i == word_index) continue;
            if (finest < 0 || tiers[i].items.size() > tiers[finest].items.size()) finest = i;
        }
        if (tiers[finest].items.size() >= tiers[word_index].items.size()) phone_index = finest;
    }
    std::vector<Item> words = std::move(tiers[word_index].items);
    for (Item& word : words)
        if (is_silence(word.text)) word.text = kSilence;
    grid->word_tier = tiers[word_index].name;
    if (phone_index >= 0) {
        grid->has_phones = true;
        grid->phone_tier = tiers[phone_index].name;
        grid->phones_first = phone_index < word_index;
        grid->phones = std::move(tiers[phone_index].items);
        size_t cursor = 0;
        for (Item& phone : grid->phones) {
            if (is_silence(phone.text)) phone.text = kSilence;
            const double middle = 0.5 * (phone.start + phone.end);
            while (cursor + 1 < words.size() && middle >= words[cursor].end) ++cursor;
            phone.word = static_cast<int>(cursor);        // index BEFORE gap filling
        }
        if (words.empty()) grid->phones.clear();
    }
    // alignment._fill_gaps: silences so that consecutive words touch
    std::vector<int> moved(words.size(), 0);
    for (size_t i = 0; i < words.size(); ++i) {
        if (!grid->words.empty() && words[i].start > grid->words.back().end) {
            Item filler;
            filler.start = grid->words.back().end;
            filler.end = words[i].start;
            filler.text = kSilence;
            filler.filler = true;
            grid->words.push_back(std::move(filler));
        }
        moved[i] = static_cast<int>(grid->words.size());
        grid->words.push_back(std::move(words[i]));
    }
    for (Item& phone : grid->phones) phone.word = moved[phone.word];
    // (the two tier names travel to the host side as one line each)
    if (grid->word_tier.find('\n') != std::string::npos ||
        grid->phone_tier.find('\n') != std::string::npos) {
        *error = "TextGrid: a line break in a tier name";
        return false;
    }
    return true;
}

// repr(float) of CPython: shortest digits that read back, positional notation
// for 1e-4 <= |x| < 1e16, else d.ddde+XX
std::string python_repr(double value) {
    if (value == 0.) return std::signbit(value) ? "-0.0" : "0.0";
    if (std::isnan(value)) return "nan";
    if (std::isinf(value)) return value < 0 ? "-inf" : "inf";
    char buffer[64];
    auto result = std::to_chars(buffer, buffer + sizeof(buffer), value, std::chars_format::scientific);
    std::string text(buffer, result.ptr);
    std::string sign;
    if (text[0] == '-') {
        sign = "-";
        text = text.substr(1);
    }
    const size_t e = text.find('e');
    std::string digits = text.substr(0, e);
    const int exponent = atoi(text.c_str() + e + 1);
    const size_t dot = digits.find('.');
    if (dot != std::string::npos) digits.erase(dot, 1);
    const int count = static_cast<int>(digits.size());
    std::string out;
    if (exponent >= -4 && exponent < 16) {
        if (exponent < 0) {
            out = "0." + std::string(static_cast<size_t>(-exponent - 1), '0') + digits;
        } else if (count <= exponent + 1) {
            out = digits + std::string(static_cast<size_t>(exponent + 1 - count), '0') + ".0";
        } else {
            out = digits.substr(0, static_cast<size_t>(exponent + 1)) + "." +
                  digits.substr(static_cast<size_t>(exponent + 1));
        }
    } else {
        out = digits.substr(0, 1);
        if (count > 1) out += "." + digits.substr(1);
        char tail[16];
        snprintf(tail, sizeof(tail), "e%c%02d", exponent < 0 ? '-' : '+', abs(exponent));
        out += tail;
    }
    return sign + out;
}

// alignment._number: integers without a fraction
std::string number(double value) {
    // (the range check first: a cast of inf or of |x| >= 2^63 is undefined)
    if (std::isfinite(value) && (value < 0 ? -value : value) < 1e15 &&
        value == static_cast<double>(static_cast<long long>(value)))
        return std::to_string(static_cast<long long>(value));
    return python_repr(value);
}

void tier_lines(std::string* out, int index, const std::string& name,
                const std::vector<const Item*>& items, double xmax) {
    *out += "    item [" + std::to_string(index) + "]:\n";
    *out += "        class = \"IntervalTier\"\n";
    *out += "        name = \"" + name + "\"\n";
    *out += "        xmin = 0\n";
    *out += "        xmax = " + number(xmax) + "\n";
    *out += "        intervals: size = " + std::to_string(items.size()) + "\n";
    int i = 0;
    for (const Item* item : items) {
        std::string label;
        if (item->text != kSilence)
            for (char c : item->text) {
                label.push_back(c);
                if (c == '"') label.push_back('"');
            }
        *out += "        intervals [" + std::to_string(++i) + "]:\n";
        *out += "            xmin = " + number(item->start) + "\n";
        *out += "            xmax = " + number(item->end) + "\n";
        *out += "            text = \"" + label + "\"\n";
    }
}

// alignment._textgrid
std::string grid_text(const Grid& grid) {
    const double xmax = grid.words.empty() ? 0. : grid.words.back().end;
    std::vector<const Item*> words, phones;
    for (const Item& word : grid.words) words.push_back(&word);
    for (const Item& phone : grid.phones) phones.push_back(&phone);
    const bool two = !phones.empty();
    std::string out = "File type = \"ooTextFile\"\nObject class = \"TextGrid\"\n\nxmin = 0\n";
    out += "xmax = " + number(xmax) + "\ntiers? <exists>\n";
    out += std::string("size = ") + (two ? "2" : "1") + "\nitem []:\n";
    int index = 0;
    if (two && grid.phones_first) tier_lines(&out, ++index, grid.phone_tier, phones, xmax);
    tier_lines(&out, ++index, grid.word_tier, words, xmax);
    if (two && !grid.phones_first) tier_lines(&out, ++index, grid.phone_tier, phones, xmax);
    return out;
}

// ---------------------------------------------------------------------------
// torch.save(tensor float32 [1, W] on the CPU): the zip container
// torch.serialization writes (records <stem>/data.pkl, byteorder, data/0,
// version, ...; stored, payloads 64-byte aligned), readable by torch.load
// ---------------------------------------------------------------------------

uint32_t crc32_of(const unsigned char* data, size_t bytes) {
    static uint32_t table[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
    });
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < bytes; ++i) crc = table[(crc ^ data[i]) & 0xFF] ^ (crc >> 8);
    return crc ^ 0xFFFFFFFFu;
}

void put16(std::string* out, uint32_t v) {
    out->push_back(static_cast<char>(v & 0xFF));
    out->push_back(static_cast<char>((v >> 8) & 0xFF));
}
void put32(std::string* out, uint32_t v) {
    put16(out, v & 0xFFFF);
    put16(out, v >> 16);
}

void pickle_int(std::string* out, int64_t value) {
    if (value < 256) {
        out->push_back('K');
        out->push_back(static_cast<char>(value));
    } else if (value < 65536) {
        out->push_back('M');
        put16(out, static_cast<uint32_t>(value));
    } else {
        out->push_back('J');
        put32(out, static_cast<uint32_t>(value));
    }
}

// pickle protocol 2 of torch._utils._rebuild_tensor_v2(FloatStorage '0' on 'cpu'
// of `count` elements, offset 0, size (1, count), stride (count, 1), no grad)
std::string tensor_pickle(int64_t count) {
    std::string p;
    p += "\x80\x02" "ctorch._utils\n_rebuild_tensor_v2\nq";
    p.push_back('\0');
    p += "((X\x07";
    p.append(3, '\0');
    p += "storageq\x01" "ctorch\nFloatStorage\nq\x02X\x01";
    p.append(3, '\0');
    p += "0q\x03X\x03";
    p.append(3, '\0');
    p += "cpuq\x04";
    pickle_int(&p, count);
    p += "tq\x05QK";
    p.push_back('\0');
    p += "K\x01";
    pickle_int(&p, count);
    p += "\x86q\x06";
    pickle_int(&p, count);
    p += "K\x01\x86q\x07\x89" "ccollections\nOrderedDict\nq\x08)Rq\ttq\nRq\x0b.";
    return p;
}

struct Record {
    std::string name;
    uint32_t crc, size, offset;
};

void zip_record(std::string* out, std::vector<Record>* records, const std::string& name,
                const unsigned char* data, size_t bytes) {
    // an extra field "FB" + padding puts the payload on a 64-byte boundary
    const size_t header = 30 + name.size() + 4;
    const size_t start = out->size() + header;
    const size_t padding = (64 - start % 64) % 64;
    Record record{name, crc32_of(data, bytes), static_cast<uint32_t>(bytes),
                  static_cast<uint32_t>(out->size())};
    put32(out, 0x04034b50u);
    put16(out, 20);                 // version needed
    put16(out, 0x0800);             // UTF-8 names
    put16(out, 0);                  // stored
    put16(out, 0);
    put16(out, 0x21);               // 1980-01-01
    put32(out, record.crc);
    put32(out, record.size);
    put32(out, record.size);
    put16(out, static_cast<uint32_t>(name.size()));
    put16(out, static_cast<uint32_t>(4 + padding));
    *out += name;
    *out += "FB";
    put16(out, static_cast<uint32_t>(padding));
    out->append(padding, 'Z');
    out->append(reinterpret_cast<const char*>(data), bytes);
    records->push_back(std::move(record));
}

std::string tensor_file(const std::string& stem, const float* scores, int64_t count) {
    std::string out;
    std::vector<Record> records;
    const std::string pickle = tensor_pickle(count);
    auto text = [&](const std::string& name, const std::string& body) {
        zip_record(&out, &records, stem + "/" + name,
                   reinterpret_cast<const unsigned char*>(body.data()), body.size());
    };
    text("data.pkl", pickle);
    text(".format_version", "1");
    text(".storage_alignment", "64");
    text("byteorder", "little");
    zip_record(&out, &records, stem + "/data/0", reinterpret_cast<const unsigned char*>(scores),
               static_cast<size_t>(count) * sizeof(float));
    text("version", "3\n");
    text(".data/serialization_id", "0000000000000000000000000000000000000000");
    const size_t directory = out.size();
    for (const Record& record : records) {
        put32(&out, 0x02014b50u);
        put16(&out, 20);
        put16(&out, 20);
        put16(&out, 0x0800);
        put16(&out, 0);
        put16(&out, 0);
        put16(&out, 0x21);
        put32(&out, record.crc);
        put32(&out, record.size);
        put32(&out, record.size);
        put16(&out, static_cast<uint32_t>(record.name.size()));
        put16(&out, 0);
        put16(&out, 0);
        put16(&out, 0);
        put16(&out, 0);
        put32(&out, 0);
        put32(&out, record.offset);
        out += record.name;
    }
    const size_t directory_bytes = out.size() - directory;
    put32(&out, 0x06054b50u);
    put16(&out, 0);
    put16(&out, 0);
    put16(&out, static_cast<uint32_t>(records.size()));
    put16(&out, static_cast<uint32_t>(records.size()));
    put32(&out, static_cast<uint32_t>(directory_bytes));
    put32(&out, static_cast<uint32_t>(directory));
    put16(&out, 0);
    return out;
}

std::string stem_of(const std::string& path) {
    const size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot > 0) name = name.substr(0, dot);
    return name;
}

}  // namespace

// One batch of (alignment file, audio file) pairs.
struct emph_file_batch {
    struct File {
        std::string text_path, audio_path;
        Grid grid;
        Wave wave;
        int status = 0;               // bit 0: alignment failed, bit 1: audio failed
        std::string error;
    };
    std::vector<File> files;
};

using namespace emph;

extern "C" {

// The file pool's OWN threads onto the given CPUs (the ones next to the GPU: a batch's
// samples are copied into pinned memory by these threads and read from there by the
// GPU's DMA engine; on a two-socket host the far socket costs 10 % of the file API).
// Threads of the caller are not touched.
int emph_files_affinity(const int32_t* cpus, int32_t count) {
    EMPH_REQUIRE(cpus && count > 0, EMPH_EINVAL, "emph_files_affinity: no CPUs");
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int32_t i = 0; i < count; ++i) {
        EMPH_REQUIRE(cpus[i] >= 0 && cpus[i] < CPU_SETSIZE, EMPH_ERANGE,
                     "emph_files_affinity: CPU %d", cpus[i]);
        CPU_SET(cpus[i], &set);
    }
    g_workers->pin(set);
    return EMPH_OK;
}

int emph_files_open(const char* const* text_paths, const char* const* audio_paths, int32_t count,
                    int32_t threads, emph_file_batch** batch) {
    EMPH_REQUIRE(batch != nullptr && count >= 0 && threads >= 1 && threads <= 64 &&
                     (count == 0 || (text_paths && audio_paths)),
                 EMPH_EINVAL, "emph_files_open: bad arguments");
    emph_file_batch* opened = new emph_file_batch;
    opened->files.resize(static_cast<size_t>(count));
    parallel(count, threads, [&](int i) {
        emph_file_batch::File& file = opened->files[static_cast<size_t>(i)];
        file.text_path = text_paths[i];
        file.audio_path = audio_paths[i];
        std::string raw, error;
        std::string text;
        bool parsed = read_whole(text_paths[i], &raw, &error);
        if (parsed && !decode(raw, &text)) {
            error = "TextGrid: not valid UTF-8 / UTF-16";
            parsed = false;
        }
        if (!parsed || !parse_grid(text, &file.grid, &error)) {
            file.status |= 1;
            file.error = error;
        }
        if (!walk_wave(audio_paths[i], &file.wave, &error)) {
            file.status |= 2;
            if (file.error.empty()) file.error = error;
        }
    });
    *batch = opened;
    return EMPH_OK;
}

void emph_files_close(emph_file_batch* batch) { delete batch; }

const char* emph_files_error(const emph_file_batch* batch, int32_t index) {
    if (batch == nullptr || index < 0 || index >= static_cast<int32_t>(batch->files.size()))
        return "";
    return batch->files[static_cast<size_t>(index)].error.c_str();
}

// sizes[i] = {status, words, phonemes, bytes of word labels, bytes of phoneme
// labels, WAVE format code, channels, sample rate, bits per sample, data offset,
// data bytes, phonemes-first}
int emph_files_sizes(const emph_file_batch* batch, int64_t* sizes) {
    EMPH_REQUIRE(batch && sizes, EMPH_EINVAL, "emph_files_sizes: null pointer");
    for (size_t i = 0; i < batch->files.size(); ++i) {
        const emph_file_batch::File& file = batch->files[i];
        int64_t* row = sizes + 12 * i;
        int64_t word_bytes = 0, phone_bytes = 0;
        for (const Item& word : file.grid.words) word_bytes += static_cast<int64_t>(word.text.size());
        for (const Item& phone : file.grid.phones)
            phone_bytes += static_cast<int64_t>(phone.text.size());
        row[0] = file.status;
        row[1] = static_cast<int64_t>(file.grid.words.size());
        row[2] = file.grid.has_phones ? static_cast<int64_t>(file.grid.phones.size()) : -1;
        row[3] = word_bytes;
        row[4] = phone_bytes;
        row[5] = file.wave.code;
        row[6] = file.wave.channels;
        row[7] = file.wave.rate;
        row[8] = file.wave.bits;
        row[9] = file.wave.offset;
        row[10] = file.wave.bytes;
        row[11] = file.grid.phones_first ? 1 : 0;
    }
    return EMPH_OK;
}

// The alignments of all files back to back: times [sum words][2] seconds (gaps
// filled with silences), labels as UTF-8 bytes with their END offsets, the same
// for the phonemes plus the (gap-filled) index of the word each belongs to, and
// per file the two tier names separated by '\n' in `tier_names` with END offsets
// (any output pointer may be NULL).
int emph_files_alignments(const emph_file_batch* batch, double* word_times, char* word_text,
                          int64_t* word_text_end, double* phone_times, char* phone_text,
                          int64_t* phone_text_end, int32_t* phone_word, char* tier_names,
                          int64_t* tier_names_end) {
    EMPH_REQUIRE(batch != nullptr, EMPH_EINVAL, "emph_files_alignments: null pointer");
    int64_t w = 0, wb = 0, p = 0, pb = 0, tb = 0;
    for (size_t i = 0; i < batch->files.size(); ++i) {
        const Grid& grid = batch->files[i].grid;
        for (const Item& word : grid.words) {
            if (word_times) {
                word_times[2 * w] = word.start;
                word_times[2 * w + 1] = word.end;
            }
            if (word_text) memcpy(word_text + wb, word.text.data(), word.text.size());
            wb += static_cast<int64_t>(word.text.size());
            if (word_text_end) word_text_end[w] = wb;
            ++w;
        }
        for (const Item& phone : grid.phones) {
            if (phone_times) {
                phone_times[2 * p] = phone.start;
                phone_times[2 * p + 1] = phone.end;
            }
            if (phone_text) memcpy(phone_text + pb, phone.text.data(), phone.text.size());
            pb += static_cast<int64_t>(phone.text.size());
            if (phone_text_end) phone_text_end[p] = pb;
            if (phone_word) phone_word[p] = phone.word;
            ++p;
        }
        const std::string names = grid.word_tier + "\n" + grid.phone_tier;
        if (tier_names) memcpy(tier_names + tb, names.data(), names.size());
        tb += static_cast<int64_t>(names.size());
        if (tier_names_end) tier_names_end[i] = tb;
    }
    return EMPH_OK;
}

int64_t emph_files_tier_name_bytes(const emph_file_batch* batch) {
    if (batch == nullptr) return 0;
    int64_t total = 0;
    for (const emph_file_batch::File& file : batch->files)
        total += static_cast<int64_t>(file.grid.word_tier.size() + 1 + file.grid.phone_tier.size());
    return total;
}

// The data chunks of files which[0 .. count) (mono 16-bit PCM or float32: their
// bytes ARE the packed staging layout), file which[k] at destination + where[k],
// bytes[k] of them (at most the chunk holds).
int emph_files_read_audio(const emph_file_batch* batch, const int32_t* which,
                          const int64_t* where, const int64_t* bytes, int32_t count,
                          void* destination, int32_t threads) {
    if (count == 0) return EMPH_OK;
    EMPH_REQUIRE(batch && which && where && bytes && destination && threads >= 1 &&
                     threads <= 64,
                 EMPH_EINVAL, "emph_files_read_audio: bad arguments");
    std::atomic<int> failed{-1};
    // (a 5-minute file is split into pieces so that a few long files spread over
    // the threads like many short ones)
    constexpr int64_t kPiece = 4 << 20;
    struct Piece {
        int file;
        int64_t from, bytes, to;
    };
    std::vector<Piece> pieces;
    for (int32_t k = 0; k < count; ++k) {
        EMPH_REQUIRE(which[k] >= 0 && which[k] < static_cast<int32_t>(batch->files.size()),
                     EMPH_EINVAL, "emph_files_read_audio: file %d", which[k]);
        const Wave& wave = batch->files[static_cast<size_t>(which[k])].wave;
        EMPH_REQUIRE(bytes[k] >= 0 && bytes[k] <= wave.bytes, EMPH_EINVAL,
                     "emph_files_read_audio: %lld bytes of a %lld-byte data chunk",
                     static_cast<long long>(bytes[k]), static_cast<long long>(wave.bytes));
        for (int64_t at = 0; at < bytes[k]; at += kPiece)
            pieces.push_back({which[k], wave.offset + at,
                              bytes[k] - at < kPiece ? bytes[k] - at : kPiece, where[k] + at});
    }
    parallel(static_cast<int>(pieces.size()), threads, [&](int index) {
        const Piece& piece = pieces[static_cast<size_t>(index)];
        const std::string& path = batch->files[static_cast<size_t>(piece.file)].audio_path;
        const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
        if (fd < 0) {
            failed.store(piece.file);
            return;
        }
        char* target = static_cast<char*>(destination) + piece.to;
        int64_t got = 0;
        while (got < piece.bytes) {
            const ssize_t n = pread(fd, target + got, static_cast<size_t>(piece.bytes - got),
                                    static_cast<off_t>(piece.from + got));
            if (n <= 0) break;
            got += n;
        }
        close(fd);
        if (got < piece.bytes) failed.store(piece.file);
    });
    EMPH_REQUIRE(failed.load() < 0, EMPH_EINVAL, "emph_files_read_audio: could not read %s",
                 batch->files[static_cast<size_t>(failed.load())].audio_path.c_str());
    return EMPH_OK;
}

// What the reference leaves behind for file which[k] (core.py:111-112):
// <prefix>.TextGrid = the alignment it loaded, <prefix>.pt = torch.save of the
// float32 scores [1, W] = scores[first[k] .. first[k + 1]).
int emph_files_write(const emph_file_batch* batch, const int32_t* which,
                     const char* const* prefixes, const float* scores, const int64_t* first,
                     int32_t count, int32_t threads) {
    if (count == 0) return EMPH_OK;
    EMPH_REQUIRE(batch && which && prefixes && scores && first && threads >= 1 && threads <= 64,
                 EMPH_EINVAL, "emph_files_write: bad arguments");
    std::mutex guard;
    std::string problem;
    parallel(count, threads, [&](int k) {
        std::string error;
        const std::string prefix = prefixes[k];
        bool ok = which[k] >= 0 && which[k] < static_cast<int32_t>(batch->files.size());
        if (ok) {
            const emph_file_batch::File& file = batch->files[static_cast<size_t>(which[k])];
            ok = write_whole(prefix + ".TextGrid", grid_text(file.grid), &error) &&
                 write_whole(prefix + ".pt",
                             tensor_file(stem_of(prefix + ".pt"), scores + first[k],
                                         first[k + 1] - first[k]),
                             &error);
        } else {
            error = "no such file in the batch";
        }
        if (!ok) {
            std::lock_guard<std::mutex> lock(guard);
            if (problem.empty()) problem = error;
        }
    });
    EMPH_REQUIRE(problem.empty(), EMPH_EINVAL, "emph_files_write: %s", problem.c_str());
    return EMPH_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Integer tables of a batch plan (emphases_amd/batch.py builds the same arrays
// with numpy; at a few hundred utterances per batch their construction was the
// largest piece of host time of a call that brings a layout never seen before)
// ---------------------------------------------------------------------------

namespace {

// Python's (and numpy's) float floor division a // b for b > 0 and quotients below
// 2^52: the interpreter derives it from fmod (floatobject.c, float_divmod) and arrives
// at the exact floor of the exact quotient.  So does this, without the fmod: a / b
// rounded can only land ON the integer above the exact quotient, never beyond it, and
// the sign of the fused a - q b says when it did.
double plan_floor_divide(double a, double b) {
    double quotient = floor(a / b);
    if (fma(-quotient, b, a) < 0.0) quotient -= 1.0;
    return quotient;
}

}  // namespace

extern "C" {

// Tile table int32 [n][4] = (segment, first position, segment's first column,
// segment's positions) of every `block`-wide tile of every segment whose count
// lies in [least, most]; host_tiles == NULL: returns the number of rows.
int64_t emph_plan_tiles(const int64_t* host_counts, const int64_t* host_offsets,
                        int32_t n_segments, int32_t block, int64_t least, int64_t most,
                        int32_t* host_tiles) {
    int64_t rows = 0;
    for (int32_t segment = 0; segment < n_segments; ++segment) {
        const int64_t count = host_counts[segment];
        if (count < least || count > most) continue;
        for (int64_t first = 0; first < count; first += block) {
            if (host_tiles != nullptr) {
                int32_t* row = host_tiles + 4 * rows;
                row[0] = segment;
                row[1] = static_cast<int32_t>(first);
                row[2] = static_cast<int32_t>(host_offsets[segment]);
                row[3] = static_cast<int32_t>(count);
            }
            ++rows;
        }
    }
    return rows;
}

// The chunk of every utterance of a batch when no utterance needs more than one
// (batch.plan_batch's vectorised pass; emphases/core.py:345-418 with the default
// batch_size=None): utterance u has counts[u] words whose (start, end) seconds are the
// next counts[u] rows of `times`, and lengths[u] samples.  Same float64 operations in
// the same order as the numpy code (and the reference's Python floats).
// Outputs, one row per utterance that yields a chunk (n_segments of them): utterance,
// start_sample, length, frames, words; bounds int64 [2][capacity >= sum(counts)]: the
// first *n_words columns are the chunk-relative (start, end) frames of those chunks'
// words.  Returns 0, or 1 when the batch must be planned one utterance at a time (an
// utterance that needs several chunks, a negative duration, a time that is not finite).
constexpr double kPlanExact = 4503599627370496.0;      // 2^52

int emph_plan_batch(const double* times, const int64_t* counts, const int64_t* lengths,
                    int32_t n_utterances, int64_t sample_rate, int64_t hopsize,
                    int64_t padding, int64_t num_fft, int64_t* utterance,
                    int64_t* start_sample, int64_t* length, int64_t* frames, int64_t* words,
                    int64_t* bounds, int64_t capacity, int64_t* n_segments,
                    int64_t* n_words) {
    EMPH_REQUIRE(counts && lengths && utterance && start_sample && length && frames && words &&
                     bounds && n_segments && n_words,
                 EMPH_EINVAL, "emph_plan_batch: null pointer");
    const double rate = static_cast<double>(sample_rate), hop = static_cast<double>(hopsize);
    int64_t total = 0;
    for (int32_t u = 0; u < n_utterances; ++u) total += counts[u];
    EMPH_REQUIRE(total == 0 || times, EMPH_EINVAL, "emph_plan_batch: null pointer");
    EMPH_REQUIRE(capacity >= total, EMPH_ERANGE, "emph_plan_batch: bounds hold %lld of %lld words",
                 static_cast<long long>(capacity), static_cast<long long>(total));
    int64_t segments = 0, kept = 0, word = 0;
    for (int32_t u = 0; u < n_utterances; word += counts[u], ++u) {
        const int64_t count = counts[u];
        if (count <= 0) continue;
        const double* rows = times + 2 * word;
        // the running frame count of the words in front of the last one against the
        // frames of the padded audio (core.py:359,369-381)
        double running = 0.0;
        for (int64_t w = 0; w < count; ++w) {
            const double start = rows[2 * w], end = rows[2 * w + 1];
            if (!std::isfinite(start) || !std::isfinite(end)) return 1;
            // a time whose frame index leaves the exactly representable integers (a
            // corrupt file: 1e300 s) would overflow the casts below - undefined in C++;
            // the one-utterance-at-a-time path reports it the way Python does
            if (std::fabs(start * rate / hop) >= kPlanExact ||
                std::fabs(end * rate / hop) >= kPlanExact)
                return 1;
            const double duration = plan_floor_divide((end - start) * rate, hop);
            if (duration < 0.0) return 1;
            if (w + 1 < count) running += duration;
        }
        const int64_t padded = lengths[u] + 2 * padding;
        const int64_t limit = static_cast<int64_t>(static_cast<double>(padded) / hop);
        if (static_cast<int64_t>(running) > limit) return 1;
        int64_t first_sample =
            static_cast<int64_t>(plan_floor_divide(rows[0] * rate, hop)) * hopsize;      // core.py:395
        int64_t last_sample =
            static_cast<int64_t>(plan_floor_divide(rows[2 * count - 1] * rate, hop)) * hopsize;
        first_sample = std::min(std::max<int64_t>(first_sample, 0), padded);             // (slices clamp)
        last_sample = std::min(std::max<int64_t>(last_sample, 0), padded);
        const int64_t samples = std::max<int64_t>(0, last_sample - first_sample);
        if (samples <= padding) continue;         // reflect padding needs more (mels.py:31-36)
        const int64_t origin = static_cast<int64_t>(rows[0] * rate / hop);
        for (int64_t w = 0; w < count; ++w) {
            bounds[kept + w] = static_cast<int64_t>(rows[2 * w] * rate / hop) - origin;
            bounds[capacity + kept + w] = static_cast<int64_t>(rows[2 * w + 1] * rate / hop) - origin;
        }
        utterance[segments] = u;
        start_sample[segments] = first_sample;
        length[segments] = samples;
        // (floor division of a value that is positive here: samples > padding)
        frames[segments] = 1 + (samples + 2 * padding - num_fft) / hopsize;
        words[segments] = count;
        kept += count;
        ++segments;
    }
    *n_segments = segments;
    *n_words = kept;
    return 0;
}

// The tables of the folded per-word sum (batch.Plan.word_sum_tables):
//   frames, frame_off  per segment; words per segment; word_columns[total_words]
//   the packed word column of every word; bounds int64 [2][total_words]
//   (chunk-relative, unclamped); restarts: sorted packed frame columns at which
//   the running sum restarts (every segment's first column among them)
// Outputs (host): slot_map int32 [ld_frames] (filled here, -1 elsewhere),
//   first int32 [ld_words + 1], lengths int32 [ld_words] (-1 on padding columns),
//   terms int32 [capacity]; returns the number of terms (or -(needed) when
//   `capacity` is too small - two terms per part at most), *n_slots the rows of
//   the running-sum buffer.
int64_t emph_plan_word_sums(const int64_t* frames, const int64_t* frame_off,
                            const int64_t* words, int32_t n_segments,
                            const int64_t* word_columns, const int64_t* bounds,
                            int64_t total_words, const int64_t* restarts, int64_t n_restarts,
                            int64_t ld_frames, int64_t ld_words, int32_t* slot_map,
                            int32_t* first, int32_t* lengths, int32_t* terms, int64_t capacity,
                            int32_t* n_slots) {
    for (int64_t i = 0; i < ld_frames; ++i) slot_map[i] = -1;
    for (int64_t i = 0; i < ld_words; ++i) lengths[i] = -1;
    for (int64_t i = 0; i <= ld_words; ++i) first[i] = 0;
    // pass 1: the frames whose running sum somebody needs (0 marks, for now)
    int64_t needed = 0;
    auto walk = [&](auto&& emit) {
        int64_t word = 0, cut = 0;
        for (int32_t segment = 0; segment < n_segments; ++segment) {
            const int64_t limit = frames[segment], column = frame_off[segment];
            for (int64_t k = 0; k < words[segment]; ++k, ++word) {
                int64_t start = bounds[word], end = bounds[total_words + word];
                start = start < 0 ? 0 : (start > limit ? limit : start);
                end = end < 0 ? 0 : (end > limit ? limit : end);
                if (end < start) end = start;
                emit(word, -1, static_cast<int32_t>(end - start), 0);
                if (end <= start) continue;
                const int64_t begin = column + start, stop = column + end;
                // the first restart strictly behind `begin` (restarts are sorted;
                // words of a segment mostly move forward, but need not)
                while (cut < n_restarts && restarts[cut] <= begin) ++cut;
                while (cut > 0 && restarts[cut - 1] > begin) --cut;
                int64_t a = begin;
                bool at_restart = cut > 0 && restarts[cut - 1] == begin;
                int64_t next = cut;
                for (;;) {
                    const bool last = next >= n_restarts || restarts[next] >= stop;
                    const int64_t b = last ? stop : restarts[next];
                    emit(word, b - 1, 0, +1);
                    if (!at_restart) emit(word, a - 1, 0, -1);
                    if (last) break;
                    a = b;
                    at_restart = true;
                    ++next;
                }
            }
        }
    };
    walk([&](int64_t word, int64_t frame_column, int32_t length, int sign) {
        if (sign == 0) {
            lengths[word_columns[word]] = length;
            return;
        }
        slot_map[frame_column] = 0;
        first[word_columns[word] + 1] += 1;
        ++needed;
    });
    int32_t slots = 0;
    for (int64_t i = 0; i < ld_frames; ++i)
        if (slot_map[i] == 0) slot_map[i] = slots++;
    *n_slots = slots;
    for (int64_t i = 0; i < ld_words; ++i) first[i + 1] += first[i];
    if (needed > capacity) return -needed;
    // pass 2: the terms, a word's parts in order, plus before minus
    std::vector<int32_t> cursor(first, first + ld_words);
    walk([&](int64_t word, int64_t frame_column, int32_t, int sign) {
        if (sign == 0) return;
        const int32_t slot = slot_map[frame_column];
        terms[cursor[word_columns[word]]++] = sign > 0 ? slot : ~slot;
    });
    return needed;
}

}  // extern "C"
