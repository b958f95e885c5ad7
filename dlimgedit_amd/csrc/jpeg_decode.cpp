// JPEG decoding behind table slot 8 (load_image).  The reference hands every file to stb_image
// (/root/reference/src/image.cpp:11-23; stb is an un-vendored dependency pinned at nothings/stb@5736b15f,
// /root/reference/depend/stb/CMakeLists.txt:3-7), which reads baseline and progressive Huffman JPEGs.  This file restates
// that decoder from the published algorithm (ITU-T T.81) and uses stb_image's integer pipeline wherever the choice shows in
// the pixels, so that a consumer sees the bytes it would have seen:
//   * inverse DCT: the Loeffler-Ligtenberg-Moschytz integer transform with 12-bit constants, columns first (>> 10 after
//     + 512), then rows (>> 17 after + 65536 + (128 << 17)), clamped to 0..255
//   * chroma up-sampling: the 3:1 triangle filter -- vertical (3 near + far + 2) >> 2, horizontal likewise, both at once
//     (3 t0 + t1 + 8) >> 4 on t = 3 near + far -- and pixel replication for other ratios
//   * YCbCr -> RGB in 20-bit fixed point with the constants 1.40200 / 0.71414 / 0.34414 / 1.77200 rounded to 12 bits
// Supported: SOF0 / SOF1 (8-bit sequential), SOF2 (progressive: spectral selection and successive approximation), 1 or 3
// components with sampling factors 1..4, restart intervals, RGB-coded files (component ids 'R' 'G' 'B' or Adobe transform
// 0).  Refused with a message: arithmetic coding, 12-bit samples, lossless and hierarchical processes, 4-component (CMYK /
// YCCK) files.  Output: 1 channel for grey files, 3 for colour -- the file's own count, as stb_image reports with
// req_comp = 0.  Pinned by tests/test_image_io.py against Pillow (libjpeg-turbo) on files written by Pillow: baseline and
// progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, optimised tables, restart markers, odd sizes (max abs difference <= 4 levels:
// the two libraries round their IDCT and colour conversion differently).
#include "common.hpp"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace dlimg {
namespace {

struct JpegError { std::string why; };
[[noreturn]] void bad(char const* why) { throw JpegError{why}; }

constexpr int kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // a corrupt run may step past the block: the extra entries keep such writes inside it
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {
    bool defined = false;
    uint8_t fast[512];            // 9-bit prefix -> symbol index, 255 = longer code
    uint16_t code[256];
    uint8_t value[256], size[257];
    unsigned maxcode[18];
    int delta[17];

    void build(uint8_t const counts[16], uint8_t const* symbols, int total) {
        int k = 0;
        for (int len = 1; len <= 16; ++len)
            for (int i = 0; i < counts[len - 1]; ++i) {
                if (k >= 256) bad("bad code lengths");
                size[k++] = uint8_t(len);
            }
        size[k] = 0;
        if (k != total) bad("bad code lengths");
        unsigned c = 0;
        k = 0;
        for (int len = 1; len <= 16; ++len) {
            delta[len] = k - int(c);
            if (size[k] == len) {
                while (size[k] == len) code[k++] = uint16_t(c++);
                if (c - 1 >= (1u << len)) bad("bad code lengths");
            }
            maxcode[len] = c << (16 - len);
            c <<= 1;
        }
        maxcode[17] = 0xffffffffu;
        std::memset(fast, 255, sizeof(fast));
        for (int i = 0; i < k; ++i) {
            value[i] = symbols[i];
            if (size[i] <= 9) {
                const int first = code[i] << (9 - size[i]), n = 1 << (9 - size[i]);
                for (int j = 0; j < n; ++j) fast[first + j] = uint8_t(i);
            }
        }
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    int blocks_w = 0, blocks_h = 0;           // blocks covering the padded (whole-MCU) plane
    int plane_w = 0, plane_h = 0;             // samples of that plane
    int real_blocks_w = 0, real_blocks_h = 0; // blocks a non-interleaved scan covers (ceil of the component's own size)
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;               // progressive: every coefficient of the padded plane
};

class Decoder {
  public:
    Decoder(uint8_t const* data, size_t size) : p_(data), end_(data + size) {}

    uint8_t* run(int* out_w, int* out_h, int* out_channels) {
        if (end_ - p_ < 4 || p_[0] != 0xff || p_[1] != 0xd8) bad("no SOI");
        p_ += 2;
        for (;;) {
            const int m = next_marker();
            if (m == 0xd9) break;                                 // EOI
            if (m == 0xda) {                                      // start of scan
                if (!frame_) bad("no SOF");
                read_scan_header();
                decode_scan();
                continue;
            }
            read_segment(m);
        }
        if (!frame_) bad("no SOF");
        if (progressive_) finish_progressive();
        return to_pixels(out_w, out_h, out_channels);
    }

  private:
    // ---- byte / marker level ------------------------------------------------------------------------------------
    int byte() { return p_ < end_ ? *p_++ : 0; }
    int word() { const int a = byte(); return (a << 8) | byte(); }

    int next_marker() {
        if (pending_marker_ >= 0) {
            const int m = pending_marker_;
            pending_marker_ = -1;
            return m;
        }
        while (p_ < end_) {
            if (*p_++ != 0xff) continue;                          // (garbage between segments is skipped)
            while (p_ < end_ && *p_ == 0xff) ++p_;
            if (p_ >= end_) break;
            const int m = *p_++;
            if (m != 0) return m;
        }
        return 0xd9;                                              // truncated file: what has been decoded is returned
    }

    void read_segment(int m) {
        if (m == 0xdd) {                                          // DRI
            if (word() != 4) bad("bad DRI len");
            restart_interval_ = word();
            return;
        }
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) return;        // markers without a body
        const int len = word();
        if (len < 2 || p_ + (len - 2) > end_) bad("bad segment length");
        uint8_t const* body = p_;
        uint8_t const* stop = p_ + (len - 2);
        p_ = stop;
        switch (m) {
        case 0xdb:                                                // DQT
            while (body < stop) {
                const int pq = *body >> 4, tq = *body & 15;
                ++body;
                if (pq > 1 || tq > 3) bad("bad DQT");
                if (body + 64 * (pq + 1) > stop) bad("bad DQT");
                for (int i = 0; i < 64; ++i) {
                    quant_[tq][kZigzag[i]] = uint16_t(pq ? (body[0] << 8) | body[1] : body[0]);
                    body += pq + 1;
                }
            }
            break;
        case 0xc4:                                                // DHT
            while (body < stop) {
                if (body + 17 > stop) bad("bad DHT");
                const int tc = *body >> 4, th = *body & 15;
                if (tc > 1 || th > 3) bad("bad DHT header");
                int total = 0;
                for (int i = 0; i < 16; ++i) total += body[1 + i];
                if (total > 256 || body + 17 + total > stop) bad("bad DHT");
                (tc ? ac_ : dc_)[th].build(body + 1, body + 17, total);
                body += 17 + total;
            }
            break;
        case 0xc0: case 0xc1: case 0xc2:
            read_frame(m, body, stop);
            break;
        case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xcb: case 0xcd: case 0xce: case 0xcf:
            bad("lossless and hierarchical JPEG are not supported");
        case 0xc9: case 0xca:
            bad("arithmetic-coded JPEG is not supported");
        case 0xee:                                                // APP14 "Adobe": colour transform flag
            if (stop - body >= 12 && !std::memcmp(body, "Adobe", 5)) adobe_transform_ = body[11];
            break;
        default:
            break;                                                // APPn, COM, ...: skipped
        }
    }

    void read_frame(int m, uint8_t const* b, uint8_t const* stop) {
        if (frame_) bad("multiple SOF");
        if (stop - b < 6) bad("bad SOF len");
        if (b[0] != 8) bad("only 8-bit");
        h_ = (b[1] << 8) | b[2];
        w_ = (b[3] << 8) | b[4];
        const int n = b[5];
        if (h_ == 0) bad("no header height");
        if (w_ == 0) bad("0 width");
        if (n == 4) bad("4-component (CMYK) JPEG is not supported");
        if (n != 1 && n != 3) bad("bad component count");
        if (stop - b < 6 + 3 * n) bad("bad SOF len");
        if ((size_t)w_ * h_ > 0x7fffffffull / 3) bad("too large");
        comp_.resize(n);
        for (int i = 0; i < n; ++i) {
            Component& c = comp_[i];
            c.id = b[6 + 3 * i];
            c.h = b[7 + 3 * i] >> 4;
            c.v = b[7 + 3 * i] & 15;
            c.tq = b[8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) bad("bad H/V");
            if (c.tq > 3) bad("bad TQ");
            hmax_ = std::max(hmax_, c.h);
            vmax_ = std::max(vmax_, c.v);
        }
        for (Component const& c : comp_)
            if (hmax_ % c.h || vmax_ % c.v) bad("bad H/V");       // (stb_image refuses fractional ratios as well)
        progressive_ = m == 0xc2;
        mcu_w_ = 8 * hmax_;
        mcu_h_ = 8 * vmax_;
        mcus_x_ = (w_ + mcu_w_ - 1) / mcu_w_;
        mcus_y_ = (h_ + mcu_h_ - 1) / mcu_h_;
        // every block costs at least one bit of entropy-coded data (its DC symbol): a header that promises more blocks than
        // the file has bits is refused before planes of that size are allocated
        size_t total_blocks = 0;
        for (Component const& c : comp_) total_blocks += (size_t)mcus_x_ * c.h * mcus_y_ * c.v;
        if (total_blocks > (size_t)(end_ - p_) * 8 + 64) bad("corrupt JPEG");
        for (Component& c : comp_) {
            c.blocks_w = mcus_x_ * c.h;
            c.blocks_h = mcus_y_ * c.v;
            c.plane_w = c.blocks_w * 8;
            c.plane_h = c.blocks_h * 8;
            const int cw = (w_ * c.h + hmax_ - 1) / hmax_, ch = (h_ * c.v + vmax_ - 1) / vmax_;
            c.real_blocks_w = (cw + 7) / 8;
            c.real_blocks_h = (ch + 7) / 8;
            c.plane.assign((size_t)c.plane_w * c.plane_h, 0);
            if (progressive_) c.coeff.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
        frame_ = true;
    }

    void read_scan_header() {
        const int len = word();
        const int n = byte();
        if (n < 1 || n > (int)comp_.size() || len != 6 + 2 * n) bad("bad SOS");
        scan_.clear();
        for (int i = 0; i < n; ++i) {
            const int id = byte(), t = byte();
            int which = -1;
            for (size_t k = 0; k < comp_.size(); ++k)
                if (comp_[k].id == id) which = int(k);
            if (which < 0) bad("bad SOS component");
            comp_[which].td = t >> 4;
            comp_[which].ta = t & 15;
            if (comp_[which].td > 3 || comp_[which].ta > 3) bad("bad SOS table");
            scan_.push_back(which);
        }
        ss_ = byte();
        se_ = byte();
        const int a = byte();
        ah_ = a >> 4;
        al_ = a & 15;
        if (progressive_) {
            if (ss_ > 63 || se_ > 63 || ss_ > se_ || ah_ > 13 || al_ > 13) bad("bad SOS");
            if (ss_ == 0 && se_ != 0) bad("bad SOS");
            if (ss_ > 0 && n != 1) bad("bad SOS");
        } else {
            if (ss_ != 0 || ah_ != 0 || al_ != 0) bad("bad SOS");
            se_ = 63;
        }
    }

    // ---- entropy-coded segment: bit reader ----------------------------------------------------------------------------
    void reset_bits() {
        bits_ = 0;
        nbits_ = 0;
        hit_marker_ = false;
        eobrun_ = 0;
        for (Component& c : comp_) c.dc_pred = 0;
    }
    void fill() {
        while (nbits_ <= 24) {
            int b = 0;
            if (!hit_marker_ && p_ < end_) {
                b = *p_++;
                if (b == 0xff) {
                    int c = p_ < end_ ? *p_++ : 0xd9;
                    while (c == 0xff) c = p_ < end_ ? *p_++ : 0xd9;
                    if (c != 0) {                                 // a marker ends the segment: zeros from here on
                        pending_marker_ = c;
                        hit_marker_ = true;
                        b = 0;
                    }
                }
            }
            bits_ |= (uint32_t)b << (24 - nbits_);
            nbits_ += 8;
        }
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (nbits_ < n) fill();
        const int v = int(bits_ >> (32 - n));
        bits_ <<= n;
        nbits_ -= n;
        return v;
    }
    int get_bit() { return get_bits(1); }
    int receive_extend(int n) {                                   // T.81 F.2.2.1: n-bit two's-complement-like value
        if (n == 0) return 0;
        const int v = get_bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int decode(Huffman const& h) {
        if (!h.defined) bad("bad huffman code");
        if (nbits_ < 16) fill();
        const int k = h.fast[bits_ >> 23];
        if (k < 255) {
            const int s = h.size[k];
            bits_ <<= s;
            nbits_ -= s;
            return h.value[k];
        }
        const unsigned top = bits_ >> 16;
        int len = 10;
        while (len <= 16 && top >= h.maxcode[len]) ++len;
        if (len > 16) bad("bad huffman code");
        const int idx = int(bits_ >> (32 - len)) + h.delta[len];
        if (idx < 0 || idx >= 256) bad("bad huffman code");
        bits_ <<= len;
        nbits_ -= len;
        return h.value[idx];
    }

    // ---- blocks --------------------------------------------------------------------------------------------------------
    void baseline_block(Component& c, int16_t* blk) {
        std::memset(blk, 0, 64 * sizeof(int16_t));
        const int t = decode(dc_[c.td]);
        if (t > 15) bad("bad huffman code");
        c.dc_pred = int(unsigned(c.dc_pred) + unsigned(receive_extend(t)));
        uint16_t const* q = quant_[c.tq];
        blk[0] = int16_t(unsigned(c.dc_pred) * q[0]);
        for (int k = 1; k < 64;) {
            const int rs = decode(ac_[c.ta]);
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (rs != 0xf0) break;                            // end of block
                k += 16;
            } else {
                k += r;
                const int z = kZigzag[k++];
                blk[z] = int16_t(unsigned(receive_extend(s)) * q[z]);
            }
        }
    }
    void progressive_dc(Component& c, int16_t* blk) {
        if (ah_ == 0) {
            const int t = decode(dc_[c.td]);
            if (t > 15) bad("bad huffman code");
            c.dc_pred = int(unsigned(c.dc_pred) + unsigned(receive_extend(t)));
            blk[0] = int16_t(unsigned(c.dc_pred) << al_);
        } else if (get_bit()) {
            blk[0] = int16_t(unsigned(int(blk[0])) + (1u << al_));
        }
    }
    void progressive_ac(Component& c, int16_t* blk) {
        Huffman const& h = ac_[c.ta];
        if (ah_ == 0) {                                           // first pass over this band
            if (eobrun_) {
                --eobrun_;
                return;
            }
            for (int k = ss_; k <= se_;) {
                const int rs = decode(h);
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) {
                        eobrun_ = (1 << r) - 1;
                        if (r) eobrun_ += get_bits(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    blk[kZigzag[k++]] = int16_t(unsigned(receive_extend(s)) << al_);
                }
            }
            return;
        }
        const int16_t bit = int16_t(1 << al_);                    // refinement of a band (T.81 G.1.2.3)
        auto refine = [&](int16_t& v) {
            if (v != 0 && get_bit() && (v & bit) == 0) v = int16_t(v > 0 ? v + bit : v - bit);
        };
        if (eobrun_) {
            --eobrun_;
            for (int k = ss_; k <= se_; ++k) refine(blk[kZigzag[k]]);
            return;
        }
        for (int k = ss_; k <= se_;) {
            const int rs = decode(h);
            int r = rs >> 4;
            const int s = rs & 15;
            int16_t fresh = 0;
            if (s == 0) {
                if (r < 15) {
                    eobrun_ = (1 << r) - 1;
                    if (r) eobrun_ += get_bits(r);
                    r = 64;                                       // run to the end of the band, refining on the way
                }
            } else {
                if (s != 1) bad("bad huffman code");
                fresh = get_bit() ? bit : int16_t(-bit);
            }
            while (k <= se_) {
                int16_t& v = blk[kZigzag[k++]];
                if (v != 0) {
                    refine(v);
                } else {
                    if (r == 0) {
                        v = fresh;
                        break;
                    }
                    --r;
                }
            }
        }
    }

    void idct_to_plane(Component& c, int bx, int by, int16_t const* d) {
        // All arithmetic modulo 2^32 (unsigned), read back as signed for the shifts: identical to signed arithmetic for every
        // coefficient a real encoder can produce, and defined -- garbage in, garbage out, no overflow trap -- for corrupt files.
        typedef uint32_t U;
        constexpr auto f2f = [](double x) { return U(int(x * 4096 + 0.5)); };      // (+ 0.5 and truncation for negative constants too: stb_image)
        auto sar = [](U x, int n) { return int32_t(x) >> n; };
        int32_t val[64];
        auto pass = [&](U s0, U s1, U s2, U s3, U s4, U s5, U s6, U s7, U& x0, U& x1, U& x2, U& x3, U& t0, U& t1, U& t2, U& t3) {
            U p2 = s2, p3 = s6;
            U p1 = (p2 + p3) * f2f(0.5411961);
            t2 = p1 + p3 * f2f(-1.847759065);
            t3 = p1 + p2 * f2f(0.765366865);
            p2 = s0;
            p3 = s4;
            t0 = (p2 + p3) * 4096u;
            t1 = (p2 - p3) * 4096u;
            x0 = t0 + t3;
            x3 = t0 - t3;
            x1 = t1 + t2;
            x2 = t1 - t2;
            t0 = s7;
            t1 = s5;
            t2 = s3;
            t3 = s1;
            p3 = t0 + t2;
            U p4 = t1 + t3;
            p1 = t0 + t3;
            p2 = t1 + t2;
            const U p5 = (p3 + p4) * f2f(1.175875602);
            t0 = t0 * f2f(0.298631336);
            t1 = t1 * f2f(2.053119869);
            t2 = t2 * f2f(3.072711026);
            t3 = t3 * f2f(1.501321110);
            p1 = p5 + p1 * f2f(-0.899976223);
            p2 = p5 + p2 * f2f(-2.562915447);
            p3 = p3 * f2f(-1.961570560);
            p4 = p4 * f2f(-0.390180644);
            t3 += p1 + p4;
            t2 += p2 + p3;
            t1 += p2 + p4;
            t0 += p1 + p3;
        };
        auto u = [](int v) { return U(int32_t(v)); };
        for (int i = 0; i < 8; ++i) {                             // columns
            int16_t const* s = d + i;
            int32_t* v = val + i;
            if (s[8] == 0 && s[16] == 0 && s[24] == 0 && s[32] == 0 && s[40] == 0 && s[48] == 0 && s[56] == 0) {
                const int32_t dc = int32_t(u(s[0]) * 4u);
                for (int r = 0; r < 8; ++r) v[r * 8] = dc;
                continue;
            }
            U x0, x1, x2, x3, t0, t1, t2, t3;
            pass(u(s[0]), u(s[8]), u(s[16]), u(s[24]), u(s[32]), u(s[40]), u(s[48]), u(s[56]), x0, x1, x2, x3, t0, t1, t2, t3);
            x0 += 512u; x1 += 512u; x2 += 512u; x3 += 512u;
            v[0] = sar(x0 + t3, 10);
            v[56] = sar(x0 - t3, 10);
            v[8] = sar(x1 + t2, 10);
            v[48] = sar(x1 - t2, 10);
            v[16] = sar(x2 + t1, 10);
            v[40] = sar(x2 - t1, 10);
            v[24] = sar(x3 + t0, 10);
            v[32] = sar(x3 - t0, 10);
        }
        auto clamp = [](int x) { return uint8_t(x < 0 ? 0 : (x > 255 ? 255 : x)); };
        uint8_t* out = c.plane.data() + (size_t)by * 8 * c.plane_w + (size_t)bx * 8;
        for (int r = 0; r < 8; ++r, out += c.plane_w) {           // rows
            int32_t const* v = val + r * 8;
            U x0, x1, x2, x3, t0, t1, t2, t3;
            pass(u(v[0]), u(v[1]), u(v[2]), u(v[3]), u(v[4]), u(v[5]), u(v[6]), u(v[7]), x0, x1, x2, x3, t0, t1, t2, t3);
            const U bias = 65536u + (128u << 17);
            x0 += bias; x1 += bias; x2 += bias; x3 += bias;
            out[0] = clamp(sar(x0 + t3, 17));
            out[7] = clamp(sar(x0 - t3, 17));
            out[1] = clamp(sar(x1 + t2, 17));
            out[6] = clamp(sar(x1 - t2, 17));
            out[2] = clamp(sar(x2 + t1, 17));
            out[5] = clamp(sar(x2 - t1, 17));
            out[3] = clamp(sar(x3 + t0, 17));
            out[4] = clamp(sar(x3 - t0, 17));
        }
    }

    // ---- one scan --------------------------------------------------------------------------------------------------------
    void restart_if_due(int& todo) {
        if (restart_interval_ == 0 || --todo > 0) return;
        // the segment ends in RSTn: byte-align, swallow the marker, reset the predictors
        if (!hit_marker_) {
            nbits_ = 0;
            bits_ = 0;
            fill();                                               // runs into the marker
        }
        if (pending_marker_ >= 0xd0 && pending_marker_ <= 0xd7) pending_marker_ = -1;
        reset_bits();
        todo = restart_interval_;
    }

    void decode_scan() {
        reset_bits();
        int todo = restart_interval_ ? restart_interval_ : 0x7fffffff;
        int16_t blk[64];
        auto one_block = [&](Component& c, int bx, int by) {
            if (!progressive_) {
                baseline_block(c, blk);
                idct_to_plane(c, bx, by, blk);
                return;
            }
            int16_t* stored = c.coeff.data() + ((size_t)by * c.blocks_w + bx) * 64;
            if (ss_ == 0) progressive_dc(c, stored);
            else progressive_ac(c, stored);
        };
        if (scan_.size() == 1) {                                  // non-interleaved: the component's own blocks, row by row
            Component& c = comp_[scan_[0]];
            for (int by = 0; by < c.real_blocks_h; ++by)
                for (int bx = 0; bx < c.real_blocks_w; ++bx) {
                    one_block(c, bx, by);
                    restart_if_due(todo);
                }
        } else {
            for (int my = 0; my < mcus_y_; ++my)
                for (int mx = 0; mx < mcus_x_; ++mx) {
                    for (int which : scan_) {
                        Component& c = comp_[which];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) one_block(c, mx * c.h + x, my * c.v + y);
                    }
                    restart_if_due(todo);
                }
        }
        // whatever is left of the segment up to the next marker belongs to nobody
        if (!hit_marker_) {
            while (p_ < end_) {
                if (*p_++ != 0xff) continue;
                while (p_ < end_ && *p_ == 0xff) ++p_;
                if (p_ >= end_) break;
                const int m = *p_++;
                if (m != 0 && !(m >= 0xd0 && m <= 0xd7)) {
                    pending_marker_ = m;
                    break;
                }
            }
        }
        nbits_ = 0;
        bits_ = 0;
    }

    void finish_progressive() {
        int16_t blk[64];
        for (Component& c : comp_) {
            uint16_t const* q = quant_[c.tq];
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx) {
                    int16_t const* stored = c.coeff.data() + ((size_t)by * c.blocks_w + bx) * 64;
                    for (int i = 0; i < 64; ++i) blk[i] = int16_t(unsigned(int(stored[i])) * q[i]);
                    idct_to_plane(c, bx, by, blk);
                }
        }
    }

    // ---- planes -> pixels ----------------------------------------------------------------------------------------------
    // one output row of component c at image row y, up-sampled to the image width (w_ + 3 bytes of slack in `line`)
    void upsampled_row(Component const& c, int y, std::vector<uint8_t>& line) const {
        const int hs = hmax_ / c.h, vs = vmax_ / c.v;
        const int cw = (w_ + hs - 1) / hs;                        // samples of this component that the image uses per row
        auto row = [&](int r) { return c.plane.data() + (size_t)std::min(r, c.plane_h - 1) * c.plane_w; };
        const int ch = (h_ + vs - 1) / vs;
        if (hs == 1 && vs == 1) {
            std::memcpy(line.data(), row(y), (size_t)w_);
            return;
        }
        if (vs == 2 && (hs == 1 || hs == 2)) {
            // the two source rows around image row y: its own ("near") and the neighbour on the side y leans to ("far")
            const int own = y >> 1;
            const int other = (y & 1) ? std::min(own + 1, ch - 1) : std::max(own - 1, 0);
            uint8_t const* near = row(own);
            uint8_t const* far = row(other);
            if (hs == 1) {
                for (int i = 0; i < w_; ++i) line[i] = uint8_t((3 * near[i] + far[i] + 2) >> 2);
                return;
            }
            if (cw == 1) {
                line[0] = line[1] = uint8_t((3 * near[0] + far[0] + 2) >> 2);
                return;
            }
            int t1 = 3 * near[0] + far[0];
            line[0] = uint8_t((t1 + 2) >> 2);
            for (int i = 1; i < cw; ++i) {
                const int t0 = t1;
                t1 = 3 * near[i] + far[i];
                line[i * 2 - 1] = uint8_t((3 * t0 + t1 + 8) >> 4);
                line[i * 2] = uint8_t((3 * t1 + t0 + 8) >> 4);
            }
            line[cw * 2 - 1] = uint8_t((t1 + 2) >> 2);
            return;
        }
        if (vs == 1 && hs == 2) {
            uint8_t const* in = row(y);
            if (cw == 1) {
                line[0] = line[1] = in[0];
                return;
            }
            line[0] = in[0];
            line[1] = uint8_t((in[0] * 3 + in[1] + 2) >> 2);
            for (int i = 1; i < cw - 1; ++i) {
                const int n = 3 * in[i] + 2;
                line[i * 2] = uint8_t((n + in[i - 1]) >> 2);
                line[i * 2 + 1] = uint8_t((n + in[i + 1]) >> 2);
            }
            line[(cw - 1) * 2] = uint8_t((in[cw - 2] * 3 + in[cw - 1] + 2) >> 2);
            line[(cw - 1) * 2 + 1] = in[cw - 1];
            return;
        }
        uint8_t const* in = row(y / vs);                          // any other ratio: replication
        for (int i = 0; i < w_; ++i) line[i] = in[i / hs];
    }

    uint8_t* to_pixels(int* out_w, int* out_h, int* out_channels) {
        const int n = (int)comp_.size();
        uint8_t* pixels = new uint8_t[(size_t)w_ * h_ * n];
        if (n == 1) {
            for (int y = 0; y < h_; ++y) std::memcpy(pixels + (size_t)y * w_, comp_[0].plane.data() + (size_t)y * comp_[0].plane_w, (size_t)w_);
        } else {
            const bool rgb = (comp_[0].id == 'R' && comp_[1].id == 'G' && comp_[2].id == 'B') || adobe_transform_ == 0;
            std::vector<uint8_t> l0((size_t)w_ + 8), l1((size_t)w_ + 8), l2((size_t)w_ + 8);
            constexpr auto fixed = [](double x) { return int(x * 4096.0 + 0.5) << 8; };
            auto clamp = [](int x) { return uint8_t(x < 0 ? 0 : (x > 255 ? 255 : x)); };
            for (int y = 0; y < h_; ++y) {
                upsampled_row(comp_[0], y, l0);
                upsampled_row(comp_[1], y, l1);
                upsampled_row(comp_[2], y, l2);
                uint8_t* out = pixels + (size_t)y * w_ * 3;
                for (int i = 0; i < w_; ++i, out += 3) {
                    if (rgb) {
                        out[0] = l0[i]; out[1] = l1[i]; out[2] = l2[i];
                        continue;
                    }
                    const int yf = (l0[i] << 20) + (1 << 19);
                    const int cb = l1[i] - 128, cr = l2[i] - 128;
                    const int r = yf + cr * fixed(1.40200);
                    const int g = yf + cr * -fixed(0.71414) + int((unsigned)(cb * -fixed(0.34414)) & 0xffff0000u);
                    const int b = yf + cb * fixed(1.77200);
                    out[0] = clamp(r >> 20);
                    out[1] = clamp(g >> 20);
                    out[2] = clamp(b >> 20);
                }
            }
        }
        *out_w = w_;
        *out_h = h_;
        *out_channels = n;
        return pixels;
    }

    uint8_t const* p_;
    uint8_t const* end_;
    int pending_marker_ = -1;
    bool frame_ = false, progressive_ = false;
    int w_ = 0, h_ = 0, hmax_ = 1, vmax_ = 1, mcu_w_ = 8, mcu_h_ = 8, mcus_x_ = 0, mcus_y_ = 0;
    int restart_interval_ = 0, adobe_transform_ = -1;
    std::vector<Component> comp_;
    std::vector<int> scan_;
    int ss_ = 0, se_ = 63, ah_ = 0, al_ = 0, eobrun_ = 0;
    uint16_t quant_[4][64] = {};
    Huffman dc_[4], ac_[4];
    uint32_t bits_ = 0;
    int nbits_ = 0;
    bool hit_marker_ = false;
};

}  // namespace

// Decodes a JPEG file image; pixels are allocated with new[] (released by destroy_image like every image of this
// library).  Throws Exception with the reference's "Failed to load image <path>: <reason>" wording.
uint8_t* decode_jpeg(uint8_t const* data, size_t size, char const* filepath, int* out_extent, int* out_channels) {
    try {
        Decoder d(data, size);
        int w = 0, h = 0;
        uint8_t* pixels = d.run(&w, &h, out_channels);
        out_extent[0] = w;
        out_extent[1] = h;
        return pixels;
    } catch (JpegError const& e) {
        throw Exception(std::string("Failed to load image ") + filepath + ": " + e.why);
    }
}

}  // namespace dlimg
