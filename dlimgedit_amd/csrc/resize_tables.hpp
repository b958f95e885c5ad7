// Per-axis contributor tables of the stb_image_resize-equivalent resampler (a-1 of SURVEY.md §8).
// The reference calls stbir_resize_uint8_generic with the default filter, sRGB colour space and edge
// clamping (/root/reference/src/image.cpp:37-51); stb is an un-vendored dependency, its published
// algorithm (v0.97) is restated here and in oracle/stb_resize.py with identical float arithmetic.
#pragma once

#include <vector>

namespace dlimg {

struct AxisTable {
    int in_size = 0, out_size = 0, taps = 0;
    std::vector<int> first;      // [out] first source index (may be negative / past the end: clamps)
    std::vector<int> count;      // [out] number of contributors
    std::vector<float> coef;     // [out][taps], zero padded
};

// default_: STBIR_FILTER_DEFAULT (Catmull-Rom when the axis grows, Mitchell otherwise); box: STBIR_FILTER_BOX
enum class ResizeFilter { default_, box };

AxisTable make_axis_table(int in_size, int out_size, ResizeFilter filter = ResizeFilter::default_);

// stbir__srgb_uchar_to_linear_float (256 entries) and fp32_to_srgb8_tab4 (104 entries)
void srgb_decode_table(float out[256]);
extern const unsigned kSrgbEncodeTab4[104];

}  // namespace dlimg
