// ASan + UBSan harness for the pure host planners (tests/test_sanitizers.py): csrc/step_queue.hpp (which lane takes which
// requests), csrc/mask_pieces.hpp (how a batch of masks is cut into transfer pieces) and csrc/resize_tables.cpp (the
// contributor tables the resize kernels index with) on a few hundred thousand random inputs, with the invariants the
// callers rely on checked on every one.
#include "mask_pieces.hpp"
#include "resize_tables.hpp"
#include "step_queue.hpp"

#include <cstdint>
#include <cmath>
#include <cstdio>
#include <numeric>

using namespace dlimg;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd(uint32_t n) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)((rng_state >> 11) % (n ? n : 1));
}

int main() {
    for (int iter = 0; iter < 200000; ++iter) {
        StepQueueState st;
        const int lanes = 1 + (int)rnd(8);
        for (int l = 0; l < lanes; ++l) {
            const int passes = (int)rnd(4);
            st.passes_in_flight.push_back(passes);
            st.images_in_flight.push_back(passes * (1 + (int)rnd(4)));
        }
        st.cursor = (int)rnd(lanes);
        const int pending = (int)rnd(40), width = (int)rnd(9), depth = 1 + (int)rnd(4);
        const bool all = rnd(2) != 0;
        const std::vector<int> before = st.images_in_flight;
        const std::vector<StepPlanPass> plan = plan_device_steps(st, pending, width, depth, all);
        int planned = 0;
        for (StepPlanPass const& p : plan) {
            if (p.lane < 0 || p.lane >= lanes || p.images <= 0 || p.images > std::max(1, width)) { std::printf("bad pass\n"); return 1; }
            planned += p.images;
        }
        if (planned > pending || (all && planned != pending)) { std::printf("planned %d of %d (all %d)\n", planned, pending, (int)all); return 1; }
        int now = std::accumulate(st.images_in_flight.begin(), st.images_in_flight.end(), 0);
        if (now != std::accumulate(before.begin(), before.end(), 0) + planned) { std::printf("state out of step\n"); return 1; }
        if (st.cursor < 0 || st.cursor >= lanes) { std::printf("cursor out of range\n"); return 1; }
    }
    for (int iter = 0; iter < 20000; ++iter) {
        const int count = (int)rnd(40);
        std::vector<size_t> sizes(count);
        size_t total = 0;
        for (auto& s : sizes) { s = 1 + rnd(rnd(4) ? 3000000 : 70); total += padded_mask_bytes(s); }
        total += rnd(2) ? rnd(64) * 4 : 0;
        const std::vector<size_t> ends = mask_piece_ends(total);
        if (total && (ends.empty() || ends.back() != total)) { std::printf("pieces do not cover the staging area\n"); return 1; }
        MaskCursor cursor;
        size_t begin = 0;
        std::vector<size_t> copied(count, 0);
        for (size_t end : ends) {
            if (end <= begin) { std::printf("empty piece\n"); return 1; }
            for (MaskCopy const& c : mask_copies_in_piece(sizes, begin, end, cursor)) {
                if (c.mask < 0 || c.mask >= count || c.bytes == 0 || c.mask_offset + c.bytes > sizes[c.mask] ||
                    c.staging_offset < begin || c.staging_offset + c.bytes > end) { std::printf("copy outside its piece or mask\n"); return 1; }
                if (c.mask_offset != copied[c.mask]) { std::printf("mask bytes out of order\n"); return 1; }
                copied[c.mask] += c.bytes;
            }
            begin = end;
        }
        for (int i = 0; i < count; ++i)
            if (copied[i] != sizes[i]) { std::printf("mask %d: %zu of %zu bytes copied\n", i, copied[i], sizes[i]); return 1; }
    }
    // resize tables: any axis from 1 pixel to far beyond what an image has, both filters; the kernels read coef[o * taps + k]
    // for k < count[o] and clamp first[o] + k into the source
    for (int iter = 0; iter < 1500; ++iter) {
        const int shape = (int)rnd(12);
        const int in_size = shape == 0 ? 1 + (int)rnd(4) : shape == 1 ? 20000 + (int)rnd(20000) : 1 + (int)rnd(4000);
        const int out_size = shape == 2 ? 1 + (int)rnd(4) : shape == 3 ? 1024 : 1 + (int)rnd(2048);
        const ResizeFilter filter = rnd(2) ? ResizeFilter::default_ : ResizeFilter::box;
        const AxisTable t = make_axis_table(in_size, out_size, filter);
        if (t.in_size != in_size || t.out_size != out_size || t.taps <= 0 || (int)t.first.size() != out_size ||
            (int)t.count.size() != out_size || t.coef.size() != (size_t)out_size * t.taps) { std::printf("table shape %d -> %d\n", in_size, out_size); return 1; }
        for (int o = 0; o < out_size; ++o) {
            if (t.count[o] <= 0 || t.count[o] > t.taps) { std::printf("count %d -> %d at %d\n", in_size, out_size, o); return 1; }
            // a contributor may lie outside the source (edge clamp), but never further than the filter reaches
            if (t.first[o] < -t.taps || t.first[o] + t.count[o] > in_size + t.taps) { std::printf("first %d -> %d at %d\n", in_size, out_size, o); return 1; }
            double sum = 0;
            for (int k = 0; k < t.taps; ++k) {
                const float c = t.coef[(size_t)o * t.taps + k];
                if (!std::isfinite(c) || (k >= t.count[o] && c != 0.0f)) { std::printf("coef %d -> %d at %d\n", in_size, out_size, o); return 1; }
                sum += c;
            }
            if (std::fabs(sum - 1.0) > 1e-3) { std::printf("weights of %d -> %d at %d sum to %g\n", in_size, out_size, o, sum); return 1; }
        }
    }
    std::printf("ok\n");
    return 0;
}
