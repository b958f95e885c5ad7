// ASan + UBSan harness for the weight file parser (csrc/weights.cpp; tests/test_sanitizers.py): argv[1] = a text file of
// tensor names, the rest = DLW files, intact or damaged.  A file may be refused; of one that is accepted every listed
// tensor's first and last element must be readable.
#include "weights.hpp"

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

namespace dlimg {
void throw_error(const char* msg) { throw Exception(msg); }
void assertion_failed(const char* file, int line, const char* expr) {
    throw Exception(std::string("Assertion failed at ") + file + ":" + std::to_string(line) + ": " + expr);
}
void hip_failed(const char*, int, const char* expr, hipError_t) { throw Exception(std::string("HIP call in a CPU harness: ") + expr); }
bool hardware_queues_trusted() { return false; }
}  // namespace dlimg

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<std::string> names;
    {
        std::ifstream f(argv[1]);
        for (std::string line; std::getline(f, line);)
            if (!line.empty()) names.push_back(line);
    }
    int loaded = 0, refused = 0;
    for (int i = 2; i < argc; ++i) {
        try {
            dlimg::WeightFile w(argv[i]);
            double sum = 0;
            int present = 0;
            for (auto const& n : names) {
                if (!w.has(n)) continue;
                dlimg::HostTensor const& t = w.get(n);
                const size_t numel = t.numel();
                if (numel) sum += (double)t.data[0] + (double)t.data[numel - 1];
                ++present;
            }
            std::printf("loaded %s dim %d depth %d heads %d: %d tensors %g\n", argv[i], w.geometry().embed_dim, w.geometry().depth,
                        w.geometry().num_heads, present, sum);
            ++loaded;
        } catch (std::exception const& e) {
            std::printf("refused %s: %s\n", argv[i], e.what());
            ++refused;
        }
    }
    std::printf("done: %d loaded, %d refused\n", loaded, refused);
    return 0;
}
