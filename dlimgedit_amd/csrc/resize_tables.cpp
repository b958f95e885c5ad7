#include "resize_tables.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>

namespace dlimg {

namespace {

// STBIR_FILTER_CATMULLROM (default when an axis is upsampled)
float catmullrom(float x) {
    x = std::fabs(x);
    if (x < 1.0f) return 1 - x * x * (2.5f - 1.5f * x);
    if (x < 2.0f) return 2 - x * (4 + x * (0.5f * x - 2.5f));
    return 0.0f;
}

// STBIR_FILTER_MITCHELL, B = C = 1/3 (default otherwise)
float mitchell(float x) {
    x = std::fabs(x);
    if (x < 1.0f) return (16 + x * x * (21 * x - 36)) / 18;
    if (x < 2.0f) return (32 + x * (-60 + x * (36 - 7 * x))) / 18;
    return 0.0f;
}

// STBIR_FILTER_BOX: the trapezoid that is a box widened by the scale (s <= 1)
float trapezoid(float x, float s) {
    const float half = s / 2;
    const float t = 0.5f + half;
    x = std::fabs(x);
    if (x >= t) return 0.0f;
    const float r = 0.5f - half;
    if (x <= r) return 1.0f;
    return (t - x) / s;
}

}  // namespace

AxisTable make_axis_table(int in_size, int out_size, ResizeFilter filter) {
    AxisTable t;
    t.in_size = in_size;
    t.out_size = out_size;
    const float scale = float(out_size) / float(in_size);
    const bool box = filter == ResizeFilter::box;
    const float filter_scale = scale > 1 ? 1 / scale : scale;      // the kernel's second argument
    const float support = box ? 0.5f + filter_scale / 2 : 2.0f;
    auto up_kernel = [&](float x) { return box ? trapezoid(x, filter_scale) : catmullrom(x); };
    auto down_kernel = [&](float x) { return box ? trapezoid(x, filter_scale) : mitchell(x); };
    std::vector<std::vector<std::pair<int, float>>> lists(out_size);

    if (scale > 1) {  // gather: coefficients of the input pixels under each output pixel, normalised to 1
        const float radius = support * scale;
        for (int n = 0; n < out_size; ++n) {
            const float center = float(n) + 0.5f;
            const float lo = (center - radius) / scale;
            const float hi = (center + radius) / scale;
            const float in_center = center / scale;
            int first = int(std::floor(lo + 0.5));
            const int last = int(std::floor(hi - 0.5));
            std::vector<float> cs;
            for (int i = 0; i <= last - first; ++i) {
                const float c = up_kernel(in_center - (float(i + first) + 0.5f));
                if (i == 0 && c == 0) {      // a leading zero drops the pixel
                    ++first;
                    --i;
                    continue;
                }
                cs.push_back(c);
            }
            float total = 0;
            for (float c : cs) total += c;
            const float fs = 1 / total;
            for (float& c : cs) c *= fs;
            while (!cs.empty() && cs.back() == 0) cs.pop_back();
            for (size_t k = 0; k < cs.size(); ++k) lists[n].push_back({first + int(k), cs[k]});
        }
    } else {          // scatter kernel(x)*scale from every input pixel incl. the clamped margin, normalise per output
        const float in_radius = support / scale;
        const int margin = int(std::ceil(support * 2 / scale)) / 2;
        struct Scatter { int j, first; std::vector<float> cs; };
        std::vector<Scatter> scat;
        for (int j = -margin; j < in_size + margin; ++j) {
            const float center = float(j) + 0.5f;
            const float lo = (center - in_radius) * scale;
            const float hi = (center + in_radius) * scale;
            const float out_center = center * scale;
            Scatter s;
            s.j = j;
            s.first = int(std::floor(lo + 0.5));
            const int last = int(std::floor(hi - 0.5));
            for (int i = s.first; i <= last; ++i) s.cs.push_back(down_kernel((float(i) + 0.5f) - out_center) * scale);
            scat.push_back(std::move(s));
        }
        std::vector<float> totals(out_size, 0.0f);
        for (auto const& s : scat)
            for (size_t k = 0; k < s.cs.size(); ++k) {
                const int i = s.first + int(k);
                if (i >= 0 && i < out_size) totals[i] += s.cs[k];
            }
        for (auto const& s : scat)
            for (size_t k = 0; k < s.cs.size(); ++k) {
                const int i = s.first + int(k);
                if (i >= 0 && i < out_size) {
                    const float w = s.cs[k] * (1 / totals[i]);
                    if (w != 0) lists[i].push_back({s.j, w});
                }
            }
    }

    for (auto const& l : lists) t.taps = std::max<int>(t.taps, int(l.size()));
    t.first.assign(out_size, 0);
    t.count.assign(out_size, 0);
    t.coef.assign(size_t(out_size) * t.taps, 0.0f);
    for (int n = 0; n < out_size; ++n) {
        auto const& l = lists[n];
        if (l.empty()) continue;
        t.first[n] = l[0].first;
        t.count[n] = int(l.size());
        for (size_t k = 0; k < l.size(); ++k) t.coef[size_t(n) * t.taps + k] = l[k].second;
    }
    return t;
}

void srgb_decode_table(float out[256]) {
    // the literal table of stb holds the exact sRGB curve printed with six decimals
    for (int i = 0; i < 256; ++i) {
        const double c = i / 255.0;
        const double v = c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
        char buf[32];
        std::snprintf(buf, sizeof(buf), "%.6f", v);
        out[i] = float(std::atof(buf));
    }
}

const unsigned kSrgbEncodeTab4[104] = {
    0x0073000d, 0x007a000d, 0x0080000d, 0x0087000d, 0x008d000d, 0x0094000d, 0x009a000d, 0x00a1000d,
    0x00a7001a, 0x00b4001a, 0x00c1001a, 0x00ce001a, 0x00da001a, 0x00e7001a, 0x00f4001a, 0x0101001a,
    0x010e0033, 0x01280033, 0x01410033, 0x015b0033, 0x01750033, 0x018f0033, 0x01a80033, 0x01c20033,
    0x01dc0067, 0x020f0067, 0x02430067, 0x02760067, 0x02aa0067, 0x02dd0067, 0x03110067, 0x03440067,
    0x037800ce, 0x03df00ce, 0x044600ce, 0x04ad00ce, 0x051400ce, 0x057b00c5, 0x05dd00bc, 0x063b00b5,
    0x06970158, 0x07420142, 0x07e30130, 0x087b0120, 0x090b0112, 0x09940106, 0x0a1700fc, 0x0a9500f2,
    0x0b0f01cb, 0x0bf401ae, 0x0ccb0195, 0x0d950180, 0x0e56016e, 0x0f0d015e, 0x0fbc0150, 0x10630143,
    0x11070264, 0x1238023e, 0x1357021d, 0x14660201, 0x156601e9, 0x165a01d3, 0x174401c0, 0x182401af,
    0x18fe0331, 0x1a9602fe, 0x1c1502d2, 0x1d7e02ad, 0x1ed4028d, 0x201a0270, 0x21520256, 0x227d0240,
    0x239f0443, 0x25c003fe, 0x27bf03c4, 0x29a10392, 0x2b6a0367, 0x2d1d0341, 0x2ebe031f, 0x304d0300,
    0x31d105b0, 0x34a80555, 0x37520507, 0x39d504c5, 0x3c37048b, 0x3e7c0458, 0x40a8042a, 0x42bd0401,
    0x44c20798, 0x488e071e, 0x4c1c06b6, 0x4f76065d, 0x52a50610, 0x55ac05cc, 0x5892058f, 0x5b590559,
    0x5e0c0a23, 0x631c0980, 0x67db08f6, 0x6c55087f, 0x70940818, 0x74a007bd, 0x787d076c, 0x7c330723,
};

}  // namespace dlimg
