// Sanitizer harness for the library's image file readers (csrc/image_io.cpp, csrc/jpeg_decode.cpp): built by
// tests/test_sanitizers.py with g++ -fsanitize=address,undefined from the library's own sources (no GPU, no HIP
// runtime: only the error helpers of common.hpp are provided here) and run over well-formed and damaged files.  A file
// may be refused (exception -> "refused"); it must never trip a sanitizer or crash.
#include "common.hpp"

#include <cstdint>
#include <cstdio>
#include <exception>
#include <string>

namespace dlimg {
uint8_t* load_image_file(char const* filepath, int* out_extent, int* out_channels);

void throw_error(const char* msg) { throw Exception(msg); }
void assertion_failed(const char* file, int line, const char* expr) {
    throw Exception(std::string("Assertion failed at ") + file + ":" + std::to_string(line) + ": " + expr);
}
void hip_failed(const char*, int, const char* expr, hipError_t) { throw Exception(std::string("HIP call in a CPU harness: ") + expr); }
bool hardware_queues_trusted() { return false; }
}  // namespace dlimg

int main(int argc, char** argv) {
    int loaded = 0, refused = 0;
    for (int i = 1; i < argc; ++i) {
        try {
            int extent[2] = {0, 0}, channels = 0;
            uint8_t* px = dlimg::load_image_file(argv[i], extent, &channels);
            unsigned long sum = 0;
            const size_t n = (size_t)extent[0] * extent[1] * channels;
            for (size_t k = 0; k < n; ++k) sum += px[k];      // every byte of the result is readable
            delete[] px;
            std::printf("loaded %s %dx%dx%d %lu\n", argv[i], extent[0], extent[1], channels, sum);
            ++loaded;
        } catch (std::exception const& e) {
            std::printf("refused %s: %s\n", argv[i], e.what());
            ++refused;
        }
    }
    std::printf("done: %d loaded, %d refused\n", loaded, refused);
    return 0;
}
