// The file boundary's thread pool under ThreadSanitizer: 64 (TextGrid, WAVE) pairs opened, read
// and written forty times with 1 .. 8 threads.  Files: any 64 pairs named u<i>.TextGrid / u<i>.wav
// in /tmp/tsan/files (e.g. Alignment.from_frames(...).save + load.save_wav).  Build (host code
// of files.hip and frontend.hip with -Xarch_host -fsanitize=thread, this file with clang++):
//   hipcc -O1 -g --offload-arch=gfx950 -fPIC -std=c++17 -Xarch_host -fsanitize=thread -Iinclude \
//       -Iemphases_amd/csrc -c emphases_amd/csrc/files.hip -o files.o        (and frontend.hip)
//   /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=thread -Iinclude -c \
//       tools/micro/files_tsan.cpp -o harness.o
//   hipcc --offload-arch=gfx950 -fsanitize=thread harness.o files.o frontend.o -lpthread -o harness
// Round 4: no report.
#include <stdio.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "emphases_hip.h"
int main() {
    std::vector<std::string> texts, waves;
    for (int i = 0; i < 64; ++i) {
        texts.push_back("/tmp/tsan/files/u" + std::to_string(i) + ".TextGrid");
        waves.push_back("/tmp/tsan/files/u" + std::to_string(i) + ".wav");
    }
    std::vector<const char*> t, w;
    for (int i = 0; i < 64; ++i) t.push_back(texts[i].c_str()), w.push_back(waves[i].c_str());
    long total = 0;
    for (int rep = 0; rep < 40; ++rep) {
        emph_file_batch* batch = nullptr;
        if (emph_files_open(t.data(), w.data(), 64, 1 + rep % 8, &batch)) { printf("open failed\n"); return 1; }
        std::vector<int64_t> sizes(64 * 12);
        emph_files_sizes(batch, sizes.data());
        for (int i = 0; i < 64; ++i) total += sizes[i * 12 + 1];
        // read the audio of every file into one buffer, in parallel
        std::vector<int32_t> index(64);
        std::vector<int64_t> offset(64), bytes(64);
        int64_t at = 0;
        for (int i = 0; i < 64; ++i) { index[i] = i; offset[i] = at; bytes[i] = sizes[i * 12 + 10]; at += bytes[i]; }
        std::vector<char> buffer(at + 64);
        if (emph_files_read_audio(batch, index.data(), offset.data(), bytes.data(), 64, buffer.data(), 1 + rep % 8)) { printf("read failed\n"); return 1; }
        // write every file's outputs, in parallel
        std::vector<std::string> prefixes;
        std::vector<const char*> p;
        std::vector<int64_t> first(65, 0);
        for (int i = 0; i < 64; ++i) { prefixes.push_back("/tmp/tsan/files/out" + std::to_string(i)); first[i + 1] = first[i] + sizes[i * 12 + 1]; }
        for (int i = 0; i < 64; ++i) p.push_back(prefixes[i].c_str());
        std::vector<float> scores(first[64] + 1, 0.5f);
        if (emph_files_write(batch, index.data(), p.data(), scores.data(), first.data(), 64, 1 + rep % 8)) { printf("write failed: %s\n", emph_last_error()); return 1; }
        emph_files_close(batch);
    }
    printf("words seen %ld\n", total);
    return 0;
}
