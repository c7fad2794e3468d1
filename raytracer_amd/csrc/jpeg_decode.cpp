// jpeg_decode.cpp — JPEG -> 8-bit RGB texels for base-colour textures.
//
// The reference gets its texels from the gltf crate, which hands JPEG files to the image crate 0.25.1 and that to
// zune-jpeg 0.4.11 (Cargo.lock); `load_image` (src/scene.rs:87-114) accepts the 8-bit grey and RGB images that come
// back and `Image::value_at` (src/texture.rs:115-136) reads channels 0..2.  JPEG is lossy and the standard leaves
// the inverse transform, the chroma upsampling and the colour conversion to the decoder, so which texels a file
// yields depends on the decoder.  zune-jpeg's sources are not in the reference tree; the stages below follow its
// published scalar algorithms as far as they are known to the author:
//   * inverse DCT: the 12-bit fixed-point integer transform of stb_image (column pass rounded at 2^-10, row pass at
//     2^-17 with the level shift folded in), on dequantised coefficients;
//   * chroma upsampling by 2: linear, weights 3:1 towards the nearer sample, rounded to an integer after each
//     direction (vertical first), edge samples repeated;
//   * YCbCr -> RGB in 16-bit integers: r = y + (45 cr >> 5), g = y - ((11 cb + 23 cr) >> 5), b = y + (113 cb >> 6)
//     with cb, cr centred on 0.
// No fixture of the reference covers JPEG textures, so the texels are "parity unpinned": grey and 4:4:4 files depend
// only on the transform and the conversion above; for subsampled files the treatment of the image borders is an
// assumption.  Against libjpeg-turbo the result differs by a few code values at most (tests/test_host.py).
#include "jpeg_decode.h"

#include <algorithm>
#include <cstring>

namespace rt {

namespace {

const uint8_t kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // a run that overshoots the block lands here instead of outside the array
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {
    bool present = false;
    uint8_t fast[512];     // 9-bit lookahead: index into `vals`, 255 = longer code
    uint16_t code[256];
    uint8_t size[257];
    uint8_t vals[256];
    uint32_t maxcode[18];  // first code of length j, left-aligned to 16 bits, that is too large
    int delta[17];         // index of the first symbol of length j minus its code

    bool build(const uint8_t counts[16], const uint8_t *symbols, int n_symbols) {
        int k = 0;
        for (int len = 1; len <= 16; ++len)
            for (int i = 0; i < counts[len - 1]; ++i) size[k++] = (uint8_t)len;
        if (k != n_symbols || k > 256) return false;
        size[k] = 0;
        memcpy(vals, symbols, (size_t)k);
        uint32_t c = 0;
        k = 0;
        for (int len = 1; len <= 16; ++len) {
            delta[len] = k - (int)c;
            if (size[k] == len) {
                while (size[k] == len) code[k++] = (uint16_t)c++;
                if (c - 1 >= (1u << len)) return false;
            }
            maxcode[len] = c << (16 - len);
            c <<= 1;
        }
        maxcode[17] = 0xFFFFFFFFu;
        memset(fast, 255, sizeof fast);
        for (int i = 0; i < k; ++i) {
            const int s = size[i];
            if (s <= 9) {
                const int first = code[i] << (9 - s), count = 1 << (9 - s);
                for (int j = 0; j < count; ++j) fast[first + j] = (uint8_t)i;
            }
        }
        present = true;
        return true;
    }
};

// The entropy-coded bytes of one scan: 0xFF 0x00 is a data byte 0xFF, any other 0xFF xx ends the data (the marker is
// kept and zeros are fed from then on, so a truncated scan decodes to something instead of reading past the end).
struct BitReader {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int count = 0;
    bool at_marker = false;
    uint8_t marker = 0;

    void fill() {
        while (count <= 24) {
            uint32_t b = 0;
            if (!at_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    while (p < end && *p == 0xFF) ++p; // fill bytes
                    const uint8_t c = p < end ? *p++ : 0xD9;
                    if (c != 0) {
                        marker = c;
                        at_marker = true;
                        b = 0;
                    }
                }
            } else {
                at_marker = true;
                if (p >= end && marker == 0) marker = 0xD9;
            }
            buf |= b << (24 - count);
            count += 8;
        }
    }
    int bits(int n) {
        if (n == 0) return 0;
        if (count < n) fill();
        const int v = (int)(buf >> (32 - n));
        buf <<= n;
        count -= n;
        return v;
    }
    int bit() { return bits(1); }
    // the value of an n-bit magnitude category: the lower half of the codes stands for negative numbers
    int extend(int n) {
        if (n == 0) return 0;
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int symbol(const Huffman &h) {
        if (count < 16) fill();
        const int look = (int)(buf >> 23);
        int k = h.fast[look];
        if (k < 255) {
            const int s = h.size[k];
            buf <<= s;
            count -= s;
            return h.vals[k];
        }
        const uint32_t top = buf >> 16;
        int len = 10;
        while (len <= 16 && top >= h.maxcode[len]) ++len;
        if (len > 16) return -1;
        const int c = (int)(buf >> (32 - len)) + h.delta[len];
        if (c < 0 || c >= 256 || h.size[c] != len) return -1;
        buf <<= len;
        count -= len;
        return h.vals[c];
    }
    void reset() {
        buf = 0;
        count = 0;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int dc_table = 0, ac_table = 0;
    int dc_pred = 0;
    int width = 0, height = 0;       // samples that belong to the image
    int blocks_w = 0, blocks_h = 0;  // stored blocks: whole MCUs
    std::vector<int16_t> coef;       // blocks_w * blocks_h * 64, natural order
    std::vector<uint8_t> plane;      // (blocks_w * 8) x (blocks_h * 8) after the inverse transform
};

inline uint8_t clamp_u8(long long x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// One 8x8 block: dequantised coefficients in natural order -> samples.  Constants are round(x * 4096).
void inverse_dct(const int16_t *coef, const uint16_t *q, uint8_t *out, int stride) {
    typedef long long wide; // a malformed file can hold coefficients that overflow 32 bits here
    wide tmp[64];
    for (int c = 0; c < 8; ++c) {
        const int16_t *d = coef + c;
        const uint16_t *dq = q + c;
        wide *o = tmp + c;
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            const wide dc = (wide)d[0] * dq[0] * 4;
            for (int r = 0; r < 8; ++r) o[r * 8] = dc;
            continue;
        }
        const wide s0 = d[0] * dq[0], s1 = d[8] * dq[8], s2 = d[16] * dq[16], s3 = d[24] * dq[24];
        const wide s4 = d[32] * dq[32], s5 = d[40] * dq[40], s6 = d[48] * dq[48], s7 = d[56] * dq[56];
        wide p2 = s2, p3 = s6;
        wide p1 = (p2 + p3) * 2217;
        wide t2 = p1 + p3 * -7567, t3 = p1 + p2 * 3135;
        p2 = s0;
        p3 = s4;
        wide t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
        wide x0 = t0 + t3, x3 = t0 - t3, x1 = t1 + t2, x2 = t1 - t2;
        t0 = s7;
        t1 = s5;
        t2 = s3;
        t3 = s1;
        p3 = t0 + t2;
        wide p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        const wide p5 = (p3 + p4) * 4816;
        t0 *= 1223;
        t1 *= 8410;
        t2 *= 12586;
        t3 *= 6149;
        p1 = p5 + p1 * -3685;
        p2 = p5 + p2 * -10497;
        p3 *= -8034;
        p4 *= -1597;
        t3 += p1 + p4;
        t2 += p2 + p3;
        t1 += p2 + p4;
        t0 += p1 + p3;
        x0 += 512;
        x1 += 512;
        x2 += 512;
        x3 += 512;
        o[0] = (x0 + t3) >> 10;
        o[56] = (x0 - t3) >> 10;
        o[8] = (x1 + t2) >> 10;
        o[48] = (x1 - t2) >> 10;
        o[16] = (x2 + t1) >> 10;
        o[40] = (x2 - t1) >> 10;
        o[24] = (x3 + t0) >> 10;
        o[32] = (x3 - t0) >> 10;
    }
    for (int r = 0; r < 8; ++r) {
        const wide *d = tmp + r * 8;
        uint8_t *o = out + (size_t)r * stride;
        wide p2 = d[2], p3 = d[6];
        wide p1 = (p2 + p3) * 2217;
        wide t2 = p1 + p3 * -7567, t3 = p1 + p2 * 3135;
        p2 = d[0];
        p3 = d[4];
        wide t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
        wide x0 = t0 + t3, x3 = t0 - t3, x1 = t1 + t2, x2 = t1 - t2;
        t0 = d[7];
        t1 = d[5];
        t2 = d[3];
        t3 = d[1];
        p3 = t0 + t2;
        wide p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        const wide p5 = (p3 + p4) * 4816;
        t0 *= 1223;
        t1 *= 8410;
        t2 *= 12586;
        t3 *= 6149;
        p1 = p5 + p1 * -3685;
        p2 = p5 + p2 * -10497;
        p3 *= -8034;
        p4 *= -1597;
        t3 += p1 + p4;
        t2 += p2 + p3;
        t1 += p2 + p4;
        t0 += p1 + p3;
        const wide bias = 65536 + (128 << 17); // rounding and the +128 level shift
        x0 += bias;
        x1 += bias;
        x2 += bias;
        x3 += bias;
        o[0] = clamp_u8((x0 + t3) >> 17);
        o[7] = clamp_u8((x0 - t3) >> 17);
        o[1] = clamp_u8((x1 + t2) >> 17);
        o[6] = clamp_u8((x1 - t2) >> 17);
        o[2] = clamp_u8((x2 + t1) >> 17);
        o[5] = clamp_u8((x2 - t1) >> 17);
        o[3] = clamp_u8((x3 + t0) >> 17);
        o[4] = clamp_u8((x3 - t0) >> 17);
    }
}

// Doubles the rows of a w x h plane: each output row is 3/4 of the nearer input row and 1/4 of the next nearer one.
std::vector<uint8_t> upsample_rows(const std::vector<uint8_t> &in, int w, int h) {
    std::vector<uint8_t> out((size_t)w * h * 2);
    for (int y = 0; y < h; ++y) {
        const uint8_t *cur = &in[(size_t)y * w];
        const uint8_t *above = &in[(size_t)std::max(y - 1, 0) * w];
        const uint8_t *below = &in[(size_t)std::min(y + 1, h - 1) * w];
        uint8_t *o0 = &out[(size_t)(2 * y) * w], *o1 = o0 + w;
        for (int x = 0; x < w; ++x) {
            o0[x] = (uint8_t)((3 * cur[x] + above[x] + 2) >> 2);
            o1[x] = (uint8_t)((3 * cur[x] + below[x] + 2) >> 2);
        }
    }
    return out;
}

// Doubles the columns of a w x h plane in the same way; the first and last output samples are copies.
std::vector<uint8_t> upsample_columns(const std::vector<uint8_t> &in, int w, int h) {
    std::vector<uint8_t> out((size_t)w * 2 * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t *i = &in[(size_t)y * w];
        uint8_t *o = &out[(size_t)y * w * 2];
        o[0] = i[0];
        for (int x = 0; x < w; ++x) {
            if (x > 0) o[2 * x] = (uint8_t)((3 * i[x] + i[x - 1] + 2) >> 2);
            if (x + 1 < w) o[2 * x + 1] = (uint8_t)((3 * i[x] + i[x + 1] + 2) >> 2);
        }
        o[2 * w - 1] = i[w - 1];
    }
    return out;
}

struct Decoder {
    const uint8_t *data;
    size_t n;
    size_t pos = 2;
    std::string &err;
    int width = 0, height = 0, n_comp = 0;
    bool progressive = false, have_frame = false;
    int h_max = 1, v_max = 1, mcus_x = 0, mcus_y = 0;
    int restart_interval = 0;
    int adobe_transform = -1;
    uint16_t quant[4][64];
    bool quant_present[4] = {false, false, false, false};
    Huffman dc_tables[4], ac_tables[4];
    Component comp[4];

    Decoder(const uint8_t *d, size_t len, std::string &e) : data(d), n(len), err(e) {}

    bool fail(const char *what) {
        err = std::string("JPEG: ") + what;
        return false;
    }
    int be16(size_t at) const { return (data[at] << 8) | data[at + 1]; }

    bool read_quant(size_t at, size_t end) {
        while (at < end) {
            const int pq = data[at] >> 4, tq = data[at] & 15;
            ++at;
            if (tq > 3 || pq > 1) return fail("bad quantisation table header");
            const size_t need = pq ? 128 : 64;
            if (at + need > end) return fail("quantisation table runs past its segment");
            for (int i = 0; i < 64; ++i) {
                quant[tq][kZigzag[i]] = (uint16_t)(pq ? be16(at + 2 * (size_t)i) : data[at + (size_t)i]);
            }
            quant_present[tq] = true;
            at += need;
        }
        return true;
    }

    bool read_huffman(size_t at, size_t end) {
        while (at < end) {
            if (at + 17 > end) return fail("Huffman table runs past its segment");
            const int tc = data[at] >> 4, th = data[at] & 15;
            if (tc > 1 || th > 3) return fail("bad Huffman table header");
            const uint8_t *counts = data + at + 1;
            int total = 0;
            for (int i = 0; i < 16; ++i) total += counts[i];
            at += 17;
            if (total > 256 || at + (size_t)total > end) return fail("Huffman table runs past its segment");
            Huffman &h = tc ? ac_tables[th] : dc_tables[th];
            if (!h.build(counts, data + at, total)) return fail("inconsistent Huffman table");
            at += (size_t)total;
        }
        return true;
    }

    bool read_frame(size_t at, size_t end, int marker) {
        if (have_frame) return fail("more than one frame");
        if (end - at < 6) return fail("short frame header");
        if (data[at] != 8) return fail("only 8-bit samples are supported");
        height = be16(at + 1);
        width = be16(at + 3);
        n_comp = data[at + 5];
        if (width == 0 || height == 0) return fail("empty image (or a height given by a DNL segment)");
        if (n_comp == 4) return fail("four-component (CMYK / YCCK) files are not supported");
        if (n_comp != 1 && n_comp != 3) return fail("bad component count");
        if (end - at < 6 + 3 * (size_t)n_comp) return fail("short frame header");
        // Planes are allocated from these two numbers before a single coded byte is looked at: refuse sizes a texture
        // does not have (64 M pixels) and sizes the file cannot hold (a block of 64 pixels takes at least one coded
        // bit per component, so more than 512 pixels per byte of file is not a picture but a header).
        if ((uint64_t)width * (uint64_t)height > (1ull << 26)) return fail("image too large");
        if ((uint64_t)width * (uint64_t)height > 512ull * (uint64_t)n + 65536ull) return fail("image size implausible for the size of the file");
        progressive = marker == 0xC2;
        for (int i = 0; i < n_comp; ++i) {
            Component &c = comp[i];
            c.id = data[at + 6 + 3 * (size_t)i];
            c.h = data[at + 7 + 3 * (size_t)i] >> 4;
            c.v = data[at + 7 + 3 * (size_t)i] & 15;
            c.tq = data[at + 8 + 3 * (size_t)i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component description");
            h_max = std::max(h_max, c.h);
            v_max = std::max(v_max, c.v);
        }
        if (n_comp == 1) comp[0].h = comp[0].v = h_max = v_max = 1; // one component: blocks come in raster order
        mcus_x = (width + 8 * h_max - 1) / (8 * h_max);
        mcus_y = (height + 8 * v_max - 1) / (8 * v_max);
        for (int i = 0; i < n_comp; ++i) {
            Component &c = comp[i];
            if (h_max % c.h || v_max % c.v || h_max / c.h > 2 || v_max / c.v > 2)
                return fail("only chroma subsampled by 1 or 2 per direction is supported");
            c.width = (width * c.h + h_max - 1) / h_max;
            c.height = (height * c.v + v_max - 1) / v_max;
            c.blocks_w = mcus_x * c.h;
            c.blocks_h = mcus_y * c.v;
            c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
        have_frame = true;
        return true;
    }

    // --- block decoders -------------------------------------------------------------------------------------
    bool block_sequential(BitReader &br, Component &c, int16_t *b) {
        const Huffman &dc = dc_tables[c.dc_table], &ac = ac_tables[c.ac_table];
        const int t = br.symbol(dc);
        if (t < 0 || t > 15) return fail("bad DC code");
        c.dc_pred = (int16_t)(c.dc_pred + br.extend(t));
        b[0] = (int16_t)c.dc_pred;
        int k = 1;
        while (k < 64) {
            const int rs = br.symbol(ac);
            if (rs < 0) return fail("bad AC code");
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (rs != 0xF0) break;
                k += 16;
            } else {
                k += r;
                b[kZigzag[k++]] = (int16_t)br.extend(s);
            }
        }
        return true;
    }

    bool block_dc_progressive(BitReader &br, Component &c, int16_t *b, int ah, int al) {
        if (ah == 0) {
            const int t = br.symbol(dc_tables[c.dc_table]);
            if (t < 0 || t > 15) return fail("bad DC code");
            c.dc_pred = (int16_t)(c.dc_pred + br.extend(t));
            b[0] = (int16_t)(c.dc_pred * (1 << al));
        } else if (br.bit()) {
            b[0] = (int16_t)(b[0] + (1 << al));
        }
        return true;
    }

    bool block_ac_progressive(BitReader &br, Component &c, int16_t *b, int ss, int se, int ah, int al, int &eob_run) {
        const Huffman &ac = ac_tables[c.ac_table];
        if (ah == 0) {
            if (eob_run) {
                --eob_run;
                return true;
            }
            int k = ss;
            do {
                const int rs = br.symbol(ac);
                if (rs < 0) return fail("bad AC code");
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = (1 << r) - 1;
                        if (r) eob_run += br.bits(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    b[kZigzag[k++]] = (int16_t)(br.extend(s) * (1 << al));
                }
            } while (k <= se);
            return true;
        }
        const int bit = 1 << al;
        auto refine = [&](int16_t &x) {
            if (br.bit() && (x & bit) == 0) x = (int16_t)(x + (x > 0 ? bit : -bit));
        };
        if (eob_run) {
            --eob_run;
            for (int k = ss; k <= se; ++k) {
                int16_t &x = b[kZigzag[k]];
                if (x != 0) refine(x);
            }
            return true;
        }
        int k = ss;
        do {
            const int rs = br.symbol(ac);
            if (rs < 0) return fail("bad AC code");
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r) eob_run += br.bits(r);
                    r = 64; // the rest of the band only gets correction bits
                }
            } else {
                if (s != 1) return fail("bad refinement code");
                s = br.bit() ? bit : -bit;
            }
            while (k <= se) {
                int16_t &x = b[kZigzag[k++]];
                if (x != 0) {
                    refine(x);
                } else {
                    if (r == 0) {
                        x = (int16_t)s;
                        break;
                    }
                    --r;
                }
            }
        } while (k <= se);
        return true;
    }

    bool read_scan(size_t at, size_t end) {
        if (!have_frame) return fail("scan before frame header");
        if (end - at < 1) return fail("short scan header");
        const int ns = data[at];
        if (ns < 1 || ns > n_comp || end - at < 4 + 2 * (size_t)ns) return fail("bad scan header");
        Component *sc[4];
        for (int i = 0; i < ns; ++i) {
            const int id = data[at + 1 + 2 * (size_t)i], tables = data[at + 2 + 2 * (size_t)i];
            Component *found = nullptr;
            for (int j = 0; j < n_comp; ++j)
                if (comp[j].id == id) found = &comp[j];
            if (!found) return fail("scan names an unknown component");
            found->dc_table = tables >> 4;
            found->ac_table = tables & 15;
            if (found->dc_table > 3 || found->ac_table > 3) return fail("bad table selector");
            sc[i] = found;
        }
        int ss = data[at + 1 + 2 * (size_t)ns], se = data[at + 2 + 2 * (size_t)ns];
        const int ah = data[at + 3 + 2 * (size_t)ns] >> 4, al = data[at + 3 + 2 * (size_t)ns] & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return fail("bad spectral selection");
            if (ss == 0 && se != 0) return fail("a progressive scan mixes DC and AC coefficients");
            if (ss > 0 && ns != 1) return fail("an AC scan must have one component");
        } else {
            ss = 0;
            se = 63;
        }
        const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss > 0;
        for (int i = 0; i < ns; ++i) {
            if (need_dc && !dc_tables[sc[i]->dc_table].present) return fail("scan uses an undefined DC table");
            if (need_ac && !ac_tables[sc[i]->ac_table].present) return fail("scan uses an undefined AC table");
        }

        BitReader br;
        br.p = data + end;
        br.end = data + n;
        int eob_run = 0;
        int until_restart = restart_interval ? restart_interval : -1;
        auto restart = [&]() -> bool { // false: the data ends here
            br.reset();
            for (int i = 0; i < n_comp; ++i) comp[i].dc_pred = 0;
            eob_run = 0;
            if (!br.at_marker) br.fill(), br.reset(); // the marker follows the last byte used; find it
            if (!br.at_marker || br.marker < 0xD0 || br.marker > 0xD7) return false;
            br.at_marker = false;
            br.marker = 0;
            return true;
        };
        auto one_block = [&](Component &c, int bx, int by) -> bool {
            int16_t *b = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
            if (!progressive) return block_sequential(br, c, b);
            if (ss == 0) return block_dc_progressive(br, c, b, ah, al);
            return block_ac_progressive(br, c, b, ss, se, ah, al, eob_run);
        };
        for (int i = 0; i < n_comp; ++i) comp[i].dc_pred = 0;
        bool more = true;
        if (ns == 1) {
            Component &c = *sc[0];
            const int bw = (c.width + 7) >> 3, bh = (c.height + 7) >> 3;
            for (int by = 0; by < bh && more; ++by)
                for (int bx = 0; bx < bw && more; ++bx) {
                    if (!one_block(c, bx, by)) return false;
                    if (until_restart > 0 && --until_restart == 0) {
                        until_restart = restart_interval;
                        if (!(by == bh - 1 && bx == bw - 1)) more = restart();
                    }
                }
        } else {
            for (int my = 0; my < mcus_y && more; ++my)
                for (int mx = 0; mx < mcus_x && more; ++mx) {
                    for (int i = 0; i < ns; ++i) {
                        Component &c = *sc[i];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x)
                                if (!one_block(c, mx * c.h + x, my * c.v + y)) return false;
                    }
                    if (until_restart > 0 && --until_restart == 0) {
                        until_restart = restart_interval;
                        if (!(my == mcus_y - 1 && mx == mcus_x - 1)) more = restart();
                    }
                }
        }
        // where the next segment starts: at the marker the reader ran into, or after the bytes it used
        if (br.at_marker && br.marker != 0) {
            pos = (size_t)(br.p - data) - 2;
            if (br.p - data < 2 || data[pos] != 0xFF) pos = (size_t)(br.p - data); // synthetic end-of-data marker
        } else {
            pos = (size_t)(br.p - data);
        }
        return true;
    }

    bool parse() {
        if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG file");
        bool seen_scan = false;
        while (true) {
            // next marker: skip anything that is not 0xFF <code>
            while (pos + 1 < n && !(data[pos] == 0xFF && data[pos + 1] != 0x00 && data[pos + 1] != 0xFF)) ++pos;
            if (pos + 1 >= n) break;
            const int marker = data[pos + 1];
            pos += 2;
            if (marker == 0xD9) break;
            if (marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01) continue;
            if (pos + 2 > n) return fail("truncated segment");
            const size_t len = (size_t)be16(pos);
            if (len < 2 || pos + len > n) return fail("segment runs past the end of the file");
            const size_t at = pos + 2, end = pos + len;
            pos = end;
            switch (marker) {
            case 0xDB:
                if (!read_quant(at, end)) return false;
                break;
            case 0xC4:
                if (!read_huffman(at, end)) return false;
                break;
            case 0xC0:
            case 0xC1:
            case 0xC2:
                if (!read_frame(at, end, marker)) return false;
                break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return fail("lossless, hierarchical and arithmetic-coded files are not supported");
            case 0xDD:
                if (end - at < 2) return fail("short restart interval segment");
                restart_interval = be16(at);
                break;
            case 0xEE:
                if (end - at >= 12 && memcmp(data + at, "Adobe", 5) == 0) adobe_transform = data[at + 11];
                break;
            case 0xDA:
                if (!read_scan(at, end)) return false;
                seen_scan = true;
                break;
            case 0xDC:
                return fail("DNL segments are not supported");
            default:
                break; // APPn, COM and the rest carry nothing the texels depend on
            }
        }
        if (!have_frame || !seen_scan) return fail("no image data");
        return true;
    }

    bool reconstruct(std::vector<uint8_t> &rgb) {
        for (int i = 0; i < n_comp; ++i) {
            Component &c = comp[i];
            if (!quant_present[c.tq]) return fail("component uses an undefined quantisation table");
            const int pw = c.blocks_w * 8, ph = c.blocks_h * 8;
            c.plane.assign((size_t)pw * ph, 0);
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx)
                    inverse_dct(&c.coef[((size_t)by * c.blocks_w + bx) * 64], quant[c.tq], &c.plane[(size_t)by * 8 * pw + (size_t)bx * 8], pw);
            std::vector<int16_t>().swap(c.coef);
            int w = pw, h = ph;
            if (v_max / c.v == 2) {
                c.plane = upsample_rows(c.plane, w, h);
                h *= 2;
            }
            if (h_max / c.h == 2) {
                c.plane = upsample_columns(c.plane, w, h);
                w *= 2;
            }
        }
        const int stride = mcus_x * h_max * 8;
        rgb.resize((size_t)width * height * 3);
        const bool is_rgb = n_comp == 3 && adobe_transform == 0;
        for (int y = 0; y < height; ++y) {
            uint8_t *o = &rgb[(size_t)y * width * 3];
            const uint8_t *p0 = &comp[0].plane[(size_t)y * stride];
            if (n_comp == 1) {
                for (int x = 0; x < width; ++x) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = p0[x];
                continue;
            }
            const uint8_t *p1 = &comp[1].plane[(size_t)y * stride], *p2 = &comp[2].plane[(size_t)y * stride];
            for (int x = 0; x < width; ++x) {
                if (is_rgb) {
                    o[3 * x] = p0[x];
                    o[3 * x + 1] = p1[x];
                    o[3 * x + 2] = p2[x];
                } else {
                    const int yy = p0[x], cb = p1[x] - 128, cr = p2[x] - 128;
                    o[3 * x] = clamp_u8(yy + ((45 * cr) >> 5));
                    o[3 * x + 1] = clamp_u8(yy - ((11 * cb + 23 * cr) >> 5));
                    o[3 * x + 2] = clamp_u8(yy + ((113 * cb) >> 6));
                }
            }
        }
        return true;
    }
};

} // namespace

bool decode_jpeg_rgb8(const uint8_t *data, size_t n, uint32_t &width, uint32_t &height, std::vector<uint8_t> &rgb,
                      std::string &err) {
    Decoder d(data, n, err);
    if (!d.parse() || !d.reconstruct(rgb)) return false;
    width = (uint32_t)d.width;
    height = (uint32_t)d.height;
    return true;
}

} // namespace rt
