// jpeg_decode.cpp -- baseline JPEG decoder for embedded glTF images (host prep, S1).
//
// The reference decodes images with its vendored stb_image (Texture.cu:23, stbi_load_from_memory(..., 0)), and a lossy
// format leaves the decoder a lot of freedom: which inverse DCT, how chroma is upsampled, how YCbCr becomes RGB.  Texels
// feed the shading, so this decoder makes the same choices and its output equals the reference's byte for byte
// (tests/golden/jpeg_ref.json was produced by the reference's own decoder; tests/test_host_prep.py compares hashes):
//   * the integer LLM inverse DCT (Loeffler/Ligtenberg/Moschytz, as in the IJG library's jidctint) with 12-bit constants,
//     column pass rounded to 10 bits, row pass to 17, +128 level shift folded into the row pass, dequantised
//     coefficients held in 16 bits;
//   * chroma upsampled by the 3:1 triangle filter (h, v, or both), nearest-neighbour for other factors;
//   * YCbCr -> RGB in 20-bit fixed point with the green chroma term truncated to 16 bits.
// Supported: SOF0 / SOF1 (8-bit, Huffman), 1 or 3 components, any sampling factors, restart intervals, JFIF or Adobe
// (RGB without transform).  Not supported (error): progressive, arithmetic coding, 12-bit, CMYK.
#include "png_decode.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace drt {

namespace {

const uint8_t kDezigzag[64 + 15] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63 };        // run-off entries for corrupt run lengths

[[noreturn]] void bad(const char *what) { throw std::runtime_error(std::string("JPEG: ") + what); }

struct Huffman {
    bool present = false;
    int maxcode[18];          // largest code of each length, left-aligned to 16 bits (+1 => exclusive bound)
    int delta[17];            // values index = code + delta[len]
    uint8_t values[256];
    void build(const uint8_t counts[16], const uint8_t *vals, int n_vals) {
        std::memcpy(values, vals, (size_t)n_vals);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            delta[len] = k - code;
            code += counts[len - 1];
            k += counts[len - 1];
            if (counts[len - 1] && code - 1 >= (1 << len)) bad("bad code lengths");
            maxcode[len] = code << (16 - len);     // exclusive, left-aligned
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;       // size in samples; allocated size (whole MCUs)
    int dc_pred = 0;
    std::vector<uint8_t> plane;
};

// Entropy-coded segment reader: removes FF00 stuffing, stops at a marker and feeds zero bits from there on.
struct BitReader {
    const uint8_t *p, *end;
    uint32_t buffer = 0;
    int bits = 0;
    int marker = 0;            // marker met in the bit stream (0 = none)
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    void fill() {
        while (bits <= 24) {
            int byte = 0;
            if (!marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    int c = p < end ? *p++ : 0xD9;
                    while (c == 0xFF) c = p < end ? *p++ : 0xD9;
                    if (c != 0) { marker = c; byte = 0; }
                }
            }
            buffer |= (uint32_t)byte << (24 - bits);
            bits += 8;
        }
    }
    int decode(const Huffman &h) {
        if (bits < 16) fill();
        const int top = (int)(buffer >> 16);
        int len = 1;
        while (top >= h.maxcode[len]) len++;
        if (len > 16) bad("bad Huffman code");
        const int code = top >> (16 - len);
        buffer <<= len; bits -= len;
        return h.values[(code + h.delta[len]) & 255];
    }
    int receive_extend(int n) {       // n magnitude bits -> signed value (ITU T.81 F.2.2.1 EXTEND)
        if (n == 0) return 0;
        if (bits < n) fill();
        const int v = (int)(buffer >> (32 - n));
        buffer <<= n; bits -= n;
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    void restart() {                  // byte-align, drop the RSTn marker
        buffer = 0; bits = 0; marker = 0;
    }
};

inline uint8_t clamp255(int x) { return (unsigned)x > 255 ? (x < 0 ? 0 : 255) : (uint8_t)x; }

// One 1-D pass of the LLM inverse DCT (Loeffler, Ligtenberg, Moschytz 1989; the IJG library's "islow" factorisation) on
// eight values, constants in 12-bit fixed point.  The eight outputs are even[k] + odd[k] (k = 0..3) and, mirrored,
// even[k] - odd[k]; the caller adds its rounding bias to the even part and shifts.
constexpr int fix12(double c) { return (int)(c * 4096 + 0.5); }       // C truncation, also for the negative constants
struct Butterfly { int even[4], odd[4]; };
inline Butterfly idct8(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
    Butterfly r;
    // even part: one rotation for (s2, s6), one butterfly for (s0, s4)
    const int rot = (s2 + s6) * fix12(0.5411961f);
    const int lo = rot + s6 * fix12(-1.847759065f);
    const int hi = rot + s2 * fix12(0.765366865f);
    const int sum = (s0 + s4) * 4096, diff = (s0 - s4) * 4096;
    r.even[0] = sum + hi; r.even[3] = sum - hi;
    r.even[1] = diff + lo; r.even[2] = diff - lo;
    // odd part: (s7, s5, s3, s1) through the shared term of the four rotations
    const int a = s7 + s3, b = s5 + s1, c = s7 + s1, d = s5 + s3;
    const int shared = (a + b) * fix12(1.175875602f);
    const int c_term = shared + c * fix12(-0.899976223f);
    const int d_term = shared + d * fix12(-2.562915447f);
    const int a_term = a * fix12(-1.961570560f);
    const int b_term = b * fix12(-0.390180644f);
    r.odd[3] = s7 * fix12(0.298631336f) + c_term + a_term;       // pairs with even[3]
    r.odd[2] = s5 * fix12(2.053119869f) + d_term + b_term;
    r.odd[1] = s3 * fix12(3.072711026f) + d_term + a_term;
    r.odd[0] = s1 * fix12(1.501321110f) + c_term + b_term;
    return r;
}

void idct_block(uint8_t *out, int stride, const short coef[64]) {
    int mid[64];
    for (int col = 0; col < 8; col++) {              // columns -> 10 fractional bits dropped
        const short *c = coef + col;
        int *m = mid + col;
        if (!(c[8] | c[16] | c[24] | c[32] | c[40] | c[48] | c[56])) {      // DC-only column: every row gets 4 * DC
            const int flat = c[0] * 4;
            for (int row = 0; row < 8; row++) m[8 * row] = flat;
            continue;
        }
        const Butterfly b = idct8(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
        for (int k = 0; k < 4; k++) {
            const int e = b.even[k] + 512;
            m[8 * k] = (e + b.odd[k]) >> 10;
            m[8 * (7 - k)] = (e - b.odd[k]) >> 10;
        }
    }
    for (int row = 0; row < 8; row++) {              // rows -> 17 bits dropped, +128 level shift and rounding in the bias
        const int *m = mid + 8 * row;
        uint8_t *o = out + (size_t)row * stride;
        const Butterfly b = idct8(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
        for (int k = 0; k < 4; k++) {
            const int e = b.even[k] + 65536 + (128 << 17);
            o[k] = clamp255((e + b.odd[k]) >> 17);
            o[7 - k] = clamp255((e - b.odd[k]) >> 17);
        }
    }
}

// ---- chroma upsampling, one output row at a time ----
const uint8_t *row_1(uint8_t *, const uint8_t *near, const uint8_t *, int, int) { return near; }
const uint8_t *row_v2(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int) {
    for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
    return out;
}
const uint8_t *row_h2(uint8_t *out, const uint8_t *in, const uint8_t *, int w, int) {
    if (w == 1) { out[0] = out[1] = in[0]; return out; }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    int i;
    for (i = 1; i < w - 1; i++) {
        const int n = 3 * in[i] + 2;
        out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2);
        out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
    }
    out[i * 2 + 0] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
    out[i * 2 + 1] = in[w - 1];
    return out;
}
const uint8_t *row_hv2(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int) {
    if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return out; }
    int t1 = 3 * near[0] + far[0];
    out[0] = (uint8_t)((t1 + 2) >> 2);
    for (int i = 1; i < w; i++) {
        const int t0 = t1;
        t1 = 3 * near[i] + far[i];
        out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
        out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    return out;
}
const uint8_t *row_generic(uint8_t *out, const uint8_t *near, const uint8_t *, int w, int hs) {
    for (int i = 0; i < w; i++)
        for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
    return out;
}
typedef const uint8_t *(*RowFn)(uint8_t *, const uint8_t *, const uint8_t *, int, int);

inline int fixed20(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

void ycbcr_row(uint8_t *out, const uint8_t *y, const uint8_t *pcb, const uint8_t *pcr, int count) {
    for (int i = 0; i < count; i++) {
        const int y_fixed = (y[i] << 20) + (1 << 19);
        const int cr = pcr[i] - 128, cb = pcb[i] - 128;
        int r = y_fixed + cr * fixed20(1.40200f);
        // the green chroma term of Cb keeps only its upper 16 bits (as a 16-bit SIMD multiply-high would)
        int g = (int)((uint32_t)y_fixed + (uint32_t)(cr * -fixed20(0.71414f)) + ((uint32_t)(cb * -fixed20(0.34414f)) & 0xffff0000u));
        int b = y_fixed + cb * fixed20(1.77200f);
        r >>= 20; g >>= 20; b >>= 20;
        out[0] = clamp255(r); out[1] = clamp255(g); out[2] = clamp255(b);
        out += 3;
    }
}

uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

}  // namespace

bool looks_like_jpeg(const uint8_t *data, size_t size) { return size >= 3 && data[0] == 0xFF && data[1] == 0xD8 && data[2] == 0xFF; }

DecodedImage decode_jpeg(const uint8_t *data, size_t size) {
    if (!looks_like_jpeg(data, size)) bad("not a JPEG stream");
    const uint8_t *p = data + 2, *const end = data + size;
    uint16_t dequant[4][64];
    bool have_q[4] = { false, false, false, false };
    Huffman dc_tab[4], ac_tab[4];
    Component comp[3];
    int n_comp = 0, img_w = 0, img_h = 0, h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    int restart_interval = 0;
    bool jfif = false, seen_sof = false, done = false;
    int adobe_transform = -1, rgb_ids = 0;

    while (!done) {
        // next marker (fill bytes FF are allowed before it)
        while (p < end && *p != 0xFF) p++;
        while (p < end && *p == 0xFF) p++;
        if (p >= end) bad("truncated (no EOI)");
        const int m = *p++;
        if (m == 0xD9) break;                                       // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;        // TEM / stray RSTn: no payload
        if (end - p < 2) bad("truncated segment");
        const int len = be16(p);
        if (len < 2 || end - p < len) bad("bad segment length");
        const uint8_t *s = p + 2, *const s_end = p + len;
        p = s_end;
        switch (m) {
        case 0xDB:                                                  // DQT
            while (s < s_end) {
                const int pq = *s >> 4, tq = *s & 15;
                s++;
                if (tq > 3 || pq > 1) bad("bad DQT");
                if (s_end - s < (pq ? 128 : 64)) bad("short DQT");
                for (int i = 0; i < 64; i++) {
                    dequant[tq][kDezigzag[i]] = pq ? be16(s) : *s;
                    s += pq ? 2 : 1;
                }
                have_q[tq] = true;
            }
            break;
        case 0xC4:                                                  // DHT
            while (s < s_end) {
                const int tc = *s >> 4, th = *s & 15;
                s++;
                if (tc > 1 || th > 3 || s_end - s < 16) bad("bad DHT");
                int n = 0;
                for (int i = 0; i < 16; i++) n += s[i];
                if (n > 256 || s_end - s < 16 + n) bad("bad DHT");
                (tc ? ac_tab : dc_tab)[th].build(s, s + 16, n);
                s += 16 + n;
            }
            break;
        case 0xC0: case 0xC1: {                                     // SOF0 / SOF1: baseline / extended sequential, Huffman
            if (seen_sof) bad("more than one frame");
            if (s_end - s < 6) bad("short SOF");
            if (s[0] != 8) bad("only 8-bit samples are supported");
            img_h = be16(s + 1); img_w = be16(s + 3); n_comp = s[5];
            if (img_w == 0 || img_h == 0) bad("empty image");
            if (n_comp != 1 && n_comp != 3) bad("only 1- or 3-component images are supported (no CMYK)");
            if (s_end - s < 6 + 3 * n_comp) bad("short SOF");
            for (int i = 0; i < n_comp; i++) {
                Component &c = comp[i];
                c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) bad("bad component");
                if (c.id == "RGB"[i]) rgb_ids++;
                h_max = std::max(h_max, c.h); v_max = std::max(v_max, c.v);
            }
            for (int i = 0; i < n_comp; i++)
                if (h_max % comp[i].h || v_max % comp[i].v) bad("fractional sampling ratio");
            mcu_x = (img_w + 8 * h_max - 1) / (8 * h_max);
            mcu_y = (img_h + 8 * v_max - 1) / (8 * v_max);
            for (int i = 0; i < n_comp; i++) {
                Component &c = comp[i];
                c.x = (img_w * c.h + h_max - 1) / h_max;
                c.y = (img_h * c.v + v_max - 1) / v_max;
                c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
                c.plane.assign((size_t)c.w2 * c.h2, 0);
            }
            seen_sof = true;
            break;
        }
        case 0xC2: bad("progressive JPEG is not supported");
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            bad("lossless / hierarchical / arithmetic-coded JPEG is not supported");
        case 0xDD:                                                  // DRI
            if (s_end - s < 2) bad("short DRI");
            restart_interval = be16(s);
            break;
        case 0xE0:                                                  // APP0: JFIF?
            if (s_end - s >= 5 && std::memcmp(s, "JFIF\0", 5) == 0) jfif = true;
            break;
        case 0xEE:                                                  // APP14: Adobe colour transform flag
            if (s_end - s >= 12 && std::memcmp(s, "Adobe\0", 6) == 0) adobe_transform = s[11];
            break;
        case 0xDA: {                                                // SOS + entropy-coded data
            if (!seen_sof) bad("scan before frame header");
            if (s_end - s < 1) bad("short SOS");
            const int ns = s[0];
            if (ns < 1 || ns > n_comp || s_end - s < 1 + 2 * ns + 3) bad("bad SOS");
            int order[3];
            for (int i = 0; i < ns; i++) {
                int which = -1;
                for (int k = 0; k < n_comp; k++) if (comp[k].id == s[1 + 2 * i]) which = k;
                if (which < 0) bad("scan names an unknown component");
                comp[which].td = s[2 + 2 * i] >> 4; comp[which].ta = s[2 + 2 * i] & 15;
                if (comp[which].td > 3 || comp[which].ta > 3) bad("bad table index");
                order[i] = which;
            }
            if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) bad("spectral selection in a sequential scan");
            for (int i = 0; i < ns; i++) {
                const Component &c = comp[order[i]];
                if (!dc_tab[c.td].present || !ac_tab[c.ta].present || !have_q[c.tq]) bad("scan uses a table that was not defined");
                comp[order[i]].dc_pred = 0;
            }
            BitReader br(p, end);
            int todo = restart_interval ? restart_interval : 0x7fffffff;
            short block[64];
            auto decode_block = [&](Component &c, int bx, int by) {
                std::memset(block, 0, sizeof block);
                const int t = br.decode(dc_tab[c.td]);
                if (t > 15) bad("bad DC size");
                c.dc_pred += br.receive_extend(t);
                block[0] = (short)(c.dc_pred * dequant[c.tq][0]);
                for (int k = 1; k < 64;) {
                    const int rs = br.decode(ac_tab[c.ta]);
                    const int ssss = rs & 15, run = rs >> 4;
                    if (ssss == 0) {
                        if (rs != 0xF0) break;                      // end of block
                        k += 16;
                    } else {
                        k += run;
                        const int zig = kDezigzag[k++];
                        block[zig] = (short)(br.receive_extend(ssss) * dequant[c.tq][zig]);
                    }
                }
                idct_block(c.plane.data() + (size_t)by * 8 * c.w2 + (size_t)bx * 8, c.w2, block);
            };
            auto after_mcu = [&]() -> bool {                       // false: the scan ends here
                if (--todo > 0) return true;
                if (br.bits < 24) br.fill();
                if (!(br.marker >= 0xD0 && br.marker <= 0xD7)) return false;
                br.restart();
                for (int k = 0; k < n_comp; k++) comp[k].dc_pred = 0;
                todo = restart_interval ? restart_interval : 0x7fffffff;
                return true;
            };
            bool more = true;
            if (ns == 1) {                                          // non-interleaved: blocks of that component, row by row
                Component &c = comp[order[0]];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                for (int j = 0; j < h && more; j++)
                    for (int i = 0; i < w && more; i++) { decode_block(c, i, j); more = after_mcu(); }
            } else {
                for (int j = 0; j < mcu_y && more; j++)
                    for (int i = 0; i < mcu_x && more; i++) {
                        for (int k = 0; k < ns; k++) {
                            Component &c = comp[order[k]];
                            for (int y = 0; y < c.v; y++)
                                for (int x = 0; x < c.h; x++) decode_block(c, i * c.h + x, j * c.v + y);
                        }
                        more = after_mcu();
                    }
            }
            // continue after the entropy-coded data: at the marker the reader stopped on, or search for the next one
            p = br.p;
            if (br.marker) {
                if (br.marker == 0xD9) done = true;
                else if (!(br.marker >= 0xD0 && br.marker <= 0xD7)) p -= 2;     // re-read that marker in the main loop
            }
            break;
        }
        default:
            break;                                                  // APPn, COM, ...: skipped
        }
    }
    if (!seen_sof) bad("no frame header");

    DecodedImage img;
    img.width = img_w; img.height = img_h; img.components = n_comp >= 3 ? 3 : 1;
    img.texels.resize((size_t)img_w * img_h * img.components);
    const bool is_rgb = n_comp == 3 && (rgb_ids == 3 || (adobe_transform == 0 && !jfif));

    struct Resample { RowFn fn; int hs, vs, w_lores, ystep, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; };
    Resample rs[3];
    for (int k = 0; k < n_comp; k++) {
        Resample &r = rs[k];
        r.hs = h_max / comp[k].h; r.vs = v_max / comp[k].v;
        r.ystep = r.vs >> 1; r.ypos = 0;
        r.w_lores = (img_w + r.hs - 1) / r.hs;
        r.line0 = r.line1 = comp[k].plane.data();
        r.buf.assign((size_t)img_w + 3 + 2 * (size_t)r.hs, 0);
        r.fn = (r.hs == 1 && r.vs == 1) ? row_1 : (r.hs == 1 && r.vs == 2) ? row_v2 : (r.hs == 2 && r.vs == 1) ? row_h2
             : (r.hs == 2 && r.vs == 2) ? row_hv2 : row_generic;
    }
    for (int j = 0; j < img_h; j++) {
        const uint8_t *rows[3] = { nullptr, nullptr, nullptr };
        for (int k = 0; k < n_comp; k++) {
            Resample &r = rs[k];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            rows[k] = r.fn(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
            }
        }
        uint8_t *out = img.texels.data() + (size_t)j * img_w * img.components;
        if (n_comp == 1) std::memcpy(out, rows[0], (size_t)img_w);
        else if (is_rgb) for (int i = 0; i < img_w; i++) { out[3 * i] = rows[0][i]; out[3 * i + 1] = rows[1][i]; out[3 * i + 2] = rows[2][i]; }
        else ycbcr_row(out, rows[0], rows[1], rows[2], img_w);
    }
    return img;
}

}  // namespace drt
