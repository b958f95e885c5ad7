// Image file IO behind table slots 8 and 9 (load_image / save_image).
// Counterpart of /root/reference/src/image.cpp:11-35, which hands both to stb_image / stb_image_write.  Neither is
// on the Segment-Anything path; the slots exist so that a consumer calling Image::load / Image::save on this library
// keeps working.  Host code only (file IO has no business on the GPU):
//   save_image  PNG, 8 bits per channel, grey / RGB / RGBA (the three channel orders the reference accepts), zlib
//               deflate, per-row filter chosen by the minimum-sum-of-absolute-differences heuristic
//   load_image  PNG: grey, grey+alpha, RGB, RGBA, palette; 1/2/4/8/16 bits; non-interlaced and Adam7; tRNS as alpha.
//               Channel counts come out as stb_image reports them with req_comp = 0 (file's own count; palette = 3, or
//               4 with tRNS), and 2-channel files are rejected with the reference's message.  16-bit samples keep
//               their high byte (stb_image's conversion).  JPEG: csrc/jpeg_decode.cpp (baseline and progressive, grey and
//               YCbCr / RGB; stb_image's integer IDCT, up-sampling and colour conversion).  The other formats stb_image
//               reads (BMP, GIF, PSD, TGA, HDR, PIC, PNM) are not decoded.
#include "common.hpp"
#include "segmentation.hpp"

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace dlimg {

uint8_t* decode_jpeg(uint8_t const* data, size_t size, char const* filepath, int* out_extent, int* out_channels);

namespace {

struct File {
    FILE* f = nullptr;
    File(char const* path, char const* mode) : f(std::fopen(path, mode)) {}
    ~File() { if (f) std::fclose(f); }
};

uint32_t be32(uint8_t const* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

void write_chunk(std::vector<uint8_t>& out, char const type[4], uint8_t const* data, size_t n) {
    put_be32(out, uint32_t(n));
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data, data + n);
    put_be32(out, uint32_t(crc32(0L, out.data() + start, uInt(n + 4))));
}

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Reverses the PNG row filters in place: `rows` rows of `stride` bytes, each preceded by its filter byte.
void unfilter(uint8_t* data, int rows, size_t stride, int bpp, char const* path) {
    std::vector<uint8_t> zero(stride, 0);
    uint8_t const* prev = zero.data();
    for (int y = 0; y < rows; ++y) {
        uint8_t* row = data + (size_t)y * (stride + 1);
        const int ft = row[0];
        uint8_t* cur = row + 1;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int add;
            switch (ft) {
            case 0: add = 0; break;
            case 1: add = a; break;
            case 2: add = b; break;
            case 3: add = (a + b) >> 1; break;
            case 4: add = paeth(a, b, c); break;
            default: throw Exception(std::string("Failed to load image ") + path + ": invalid filter");
            }
            cur[i] = uint8_t(cur[i] + add);
        }
        prev = cur;
    }
}

}  // namespace

void save_image_file(dlimg_ImageView const& img, char const* filepath) {
    if (!(img.channels == 1 || img.channels == 3 || img.channels == 4))
        throw Exception("Unsupported channel order [" + std::to_string(img.channels) + "]");
    if (!img.pixels || img.width <= 0 || img.height <= 0) throw Exception(std::string("Failed to save image ") + filepath);
    const int comp = img.channels, w = img.width, h = img.height;
    const size_t stride = (size_t)w * comp;
    // the reference writes packed rows (stride = width * channels, image.cpp:31); a view's own stride is honoured here
    const size_t src_stride = img.stride > 0 ? (size_t)img.stride : stride;
    std::vector<uint8_t> raw((stride + 1) * h), best(stride), trial(stride);
    std::vector<uint8_t> zero(stride, 0);
    for (int y = 0; y < h; ++y) {
        uint8_t const* cur = img.pixels + (size_t)y * src_stride;
        uint8_t const* prev = y ? img.pixels + (size_t)(y - 1) * src_stride : zero.data();
        long best_sum = -1;
        int best_ft = 0;
        for (int ft = 0; ft < 5; ++ft) {
            long sum = 0;
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)comp ? cur[i - comp] : 0, b = prev[i], c = i >= (size_t)comp ? prev[i - comp] : 0;
                const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? ((a + b) >> 1) : paeth(a, b, c);
                trial[i] = uint8_t(cur[i] - pred);
                sum += std::abs((int)(int8_t)trial[i]);
            }
            if (best_sum < 0 || sum < best_sum) { best_sum = sum; best_ft = ft; best.swap(trial); }
        }
        raw[(size_t)y * (stride + 1)] = uint8_t(best_ft);
        std::memcpy(raw.data() + (size_t)y * (stride + 1) + 1, best.data(), stride);
    }
    uLongf zn = compressBound(uLong(raw.size()));
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), uLong(raw.size()), 8) != Z_OK)
        throw Exception(std::string("Failed to save image ") + filepath);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, uint32_t(w));
    put_be32(ihdr, uint32_t(h));
    ihdr.push_back(8);
    ihdr.push_back(comp == 1 ? 0 : comp == 3 ? 2 : 6);
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    write_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    write_chunk(out, "IDAT", z.data(), zn);
    write_chunk(out, "IEND", nullptr, 0);
    File f(filepath, "wb");
    if (!f.f || std::fwrite(out.data(), 1, out.size(), f.f) != out.size())
        throw Exception(std::string("Failed to save image ") + filepath);
}

uint8_t* load_image_file(char const* filepath, int* out_extent, int* out_channels) {
    auto fail = [&](char const* why) { return Exception(std::string("Failed to load image ") + filepath + ": " + why); };
    std::vector<uint8_t> data;
    {
        File f(filepath, "rb");
        if (!f.f) throw fail("can't fopen");
        std::fseek(f.f, 0, SEEK_END);
        const long n = std::ftell(f.f);
        std::fseek(f.f, 0, SEEK_SET);
        if (n <= 0) throw fail("unknown image type");
        data.resize(size_t(n));
        if (std::fread(data.data(), 1, data.size(), f.f) != data.size()) throw fail("can't fopen");
    }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (data.size() < 8 || std::memcmp(data.data(), sig, 8) != 0) {
        if (data.size() > 2 && data[0] == 0xff && data[1] == 0xd8) {
            uint8_t* pixels = decode_jpeg(data.data(), data.size(), filepath, out_extent, out_channels);     // csrc/jpeg_decode.cpp
            if (*out_channels != 1 && *out_channels != 3 && *out_channels != 4) {
                delete[] pixels;
                throw Exception("Unsupported number of channels (" + std::to_string(*out_channels) + ") in " + filepath);
            }
            return pixels;
        }
        throw fail("unknown image type");
    }
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool seen_ihdr = false, seen_iend = false;
    for (size_t pos = 8; pos + 12 <= data.size() && !seen_iend;) {
        const uint32_t len = be32(&data[pos]);
        if (pos + 12 + (size_t)len > data.size()) throw fail("corrupt PNG");
        uint8_t const* type = &data[pos + 4];
        uint8_t const* body = &data[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) throw fail("bad IHDR len");
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] || body[11]) throw fail("bad comp/filter method");
            if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) throw fail("too large");
            if (interlace > 1) throw fail("bad interlace method");
            seen_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            seen_iend = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!seen_ihdr || idat.empty()) throw fail("no IDAT");
    int samples;       // samples per pixel in the file
    switch (ctype) {
    case 0: samples = 1; break;
    case 2: samples = 3; break;
    case 3: samples = 1; break;
    case 4: samples = 2; break;
    case 6: samples = 4; break;
    default: throw fail("bad ctype");
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) throw fail("1/2/4/8/16-bit only");
    if (ctype == 3 && (depth == 16 || palette.empty())) throw fail("bad palette");
    if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) throw fail("bad bit depth");
    // channels as stb_image reports them
    int out_ch = ctype == 3 ? (trns.empty() ? 3 : 4) : samples + ((ctype == 0 || ctype == 2) && !trns.empty() ? 1 : 0);
    if (out_ch != 1 && out_ch != 3 && out_ch != 4)
        throw Exception("Unsupported number of channels (" + std::to_string(out_ch) + ") in " + filepath);

    // ---- inflate
    const int bits_pp = samples * depth;
    const int bpp = (bits_pp + 7) / 8;                          // filter distance in bytes
    auto row_bytes = [&](uint32_t pw) { return ((size_t)pw * bits_pp + 7) / 8; };
    // pass geometry (Adam7) or the single pass of a non-interlaced file
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1},
                     dy[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t expect = 0;
    const int passes = interlace ? 7 : 1;
    uint32_t pw[7], ph[7];
    for (int p = 0; p < passes; ++p) {
        pw[p] = interlace ? (w - xs[p] + dx[p] - 1) / dx[p] : w;
        ph[p] = interlace ? (h - ys[p] + dy[p] - 1) / dy[p] : h;
        if (pw[p] && ph[p]) expect += (row_bytes(pw[p]) + 1) * ph[p];
    }
    // stb_image refuses images whose byte counts do not fit an int (stbi__mad3sizes_valid); and deflate cannot expand
    // by more than 1032:1, so a header that promises more than its IDAT can hold is refused BEFORE anything of that
    // size is allocated and zero-filled (a 100-byte file must not cost gigabytes)
    constexpr size_t kIntMax = 0x7fffffff;
    if ((size_t)w * h > kIntMax / (size_t)out_ch || expect > kIntMax) throw fail("too large");
    if (expect > idat.size() * 1032 + 1024) throw fail("bad zlib data");
    std::vector<uint8_t> raw(expect);
    uLongf got = uLongf(expect);
    const int zr = uncompress(raw.data(), &got, idat.data(), uLong(idat.size()));
    if ((zr != Z_OK && zr != Z_BUF_ERROR) || got != expect) throw fail("bad zlib data");

    // ---- samples -> 8-bit output pixels
    uint8_t* pixels = new uint8_t[(size_t)w * h * out_ch];
    try {
        auto sample = [&](uint8_t const* row, size_t index) -> int {      // index-th sample of a row, scaled to 8 bits
            if (depth == 8) return row[index];
            if (depth == 16) return row[index * 2];                       // high byte (stb_image)
            const int per = 8 / depth;
            const int v = (row[index / per] >> ((per - 1 - int(index % per)) * depth)) & ((1 << depth) - 1);
            return ctype == 3 ? v : v * (255 / ((1 << depth) - 1));
        };
        auto raw16 = [&](uint8_t const* row, size_t index) -> int {       // un-scaled sample for tRNS comparison
            if (depth == 16) return (row[index * 2] << 8) | row[index * 2 + 1];
            if (depth == 8) return row[index];
            const int per = 8 / depth;
            return (row[index / per] >> ((per - 1 - int(index % per)) * depth)) & ((1 << depth) - 1);
        };
        size_t off = 0;
        for (int p = 0; p < passes; ++p) {
            if (!pw[p] || !ph[p]) continue;
            const size_t rb = row_bytes(pw[p]);
            unfilter(raw.data() + off, int(ph[p]), rb, bpp, filepath);
            for (uint32_t y = 0; y < ph[p]; ++y) {
                uint8_t const* row = raw.data() + off + (size_t)y * (rb + 1) + 1;
                const uint32_t oy = interlace ? ys[p] + y * dy[p] : y;
                for (uint32_t x = 0; x < pw[p]; ++x) {
                    const uint32_t ox = interlace ? xs[p] + x * dx[p] : x;
                    uint8_t* dst = pixels + ((size_t)oy * w + ox) * out_ch;
                    if (ctype == 3) {
                        const size_t idx = size_t(sample(row, x));
                        if (idx * 3 + 2 >= palette.size()) throw fail("invalid palette index");
                        dst[0] = palette[idx * 3]; dst[1] = palette[idx * 3 + 1]; dst[2] = palette[idx * 3 + 2];
                        if (out_ch == 4) dst[3] = idx < trns.size() ? trns[idx] : 255;
                    } else {
                        for (int c = 0; c < samples; ++c) dst[c] = uint8_t(sample(row, (size_t)x * samples + c));
                        if (out_ch == samples + 1) {                    // tRNS colour key of a grey / RGB file
                            bool key = trns.size() >= (size_t)samples * 2;
                            for (int c = 0; key && c < samples; ++c)
                                key = raw16(row, (size_t)x * samples + c) == ((trns[c * 2] << 8) | trns[c * 2 + 1]);
                            dst[samples] = key ? 0 : 255;
                        }
                    }
                }
            }
            off += (rb + 1) * ph[p];
        }
    } catch (...) {
        delete[] pixels;
        throw;
    }
    out_extent[0] = int(w);
    out_extent[1] = int(h);
    *out_channels = out_ch;
    return pixels;      // released by destroy_image (delete[]): one allocator for both, unlike the reference
}

}  // namespace dlimg
