// png_decode.cpp -- PNG -> 8-bit texels with the file's native channel count.
//
// Replaces stbi_load_from_memory(data, size, &w, &h, &comp, 0) as used by
// Texture::Texture (Core/Scene/Texture.cu:21-30).  PNG is lossless, so any conforming decoder
// yields the texels stb_image yields; what must match is stb's CHANNEL COUNT convention:
// grey 1, grey+alpha 2, RGB 3, RGBA 4, palette 3 (4 with tRNS), tRNS colour key adds alpha,
// 16-bit samples keep the high byte, sub-byte grey samples are scaled to 0..255.
// Written from the PNG (ISO/IEC 15948) and DEFLATE (RFC 1951) / zlib (RFC 1950) specifications.
#include "png_decode.hpp"

#include <cstring>
#include <stdexcept>

namespace drt {
namespace {

[[noreturn]] void bad(const char *what) { throw std::runtime_error(std::string("PNG: ") + what); }

// ---- RFC 1951 inflate ----
struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    uint32_t bits(int n) {
        while (nbits < n) {
            if (p >= end) bad("deflate stream truncated");
            acc |= (uint32_t)(*p++) << nbits;
            nbits += 8;
        }
        uint32_t v = acc & ((n == 32) ? 0xFFFFFFFFu : ((1u << n) - 1u));
        acc = (n == 32) ? 0 : (acc >> n);
        nbits -= n;
        return v;
    }
    void align_byte() { acc = 0; nbits = 0; }
};

// Canonical Huffman decoder driven by code-length counts (decode one bit at a time).
struct Huffman {
    uint16_t count[16];
    uint16_t symbol[288];
    void build(const uint8_t *lengths, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; i++)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= (int)br.bits(1);
            int cnt = count[len];
            if (code - cnt < first) return symbol[index + (code - first)];
            index += cnt;
            first += cnt;
            first <<= 1;
            code <<= 1;
        }
        bad("invalid Huffman code");
    }
};

const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

void inflate_block(BitReader &br, const Huffman &lit, const Huffman &dist, std::vector<uint8_t> &out) {
    for (;;) {
        int sym = lit.decode(br);
        if (sym < 256) { out.push_back((uint8_t)sym); continue; }
        if (sym == 256) return;
        sym -= 257;
        if (sym >= 29) bad("bad length symbol");
        size_t len = kLenBase[sym] + br.bits(kLenExtra[sym]);
        int ds = dist.decode(br);
        if (ds >= 30) bad("bad distance symbol");
        size_t d = kDistBase[ds] + br.bits(kDistExtra[ds]);
        if (d > out.size()) bad("distance beyond window");
        size_t from = out.size() - d;
        for (size_t i = 0; i < len; i++) out.push_back(out[from + i]);   // may overlap: byte by byte
    }
}

std::vector<uint8_t> zlib_inflate(const uint8_t *data, size_t n, size_t expect) {
    if (n < 6) bad("zlib stream too short");
    if ((data[0] & 0x0F) != 8 || ((data[0] << 8 | data[1]) % 31) != 0 || (data[1] & 0x20)) bad("bad zlib header");
    BitReader br{ data + 2, data + n };
    std::vector<uint8_t> out;
    out.reserve(expect);
    int final_block;
    do {
        final_block = (int)br.bits(1);
        int type = (int)br.bits(2);
        if (type == 0) {
            br.align_byte();
            if (br.end - br.p < 4) bad("stored block truncated");
            uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xFFFF) != nlen) bad("stored block length mismatch");
            if ((size_t)(br.end - br.p) < len) bad("stored block truncated");
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1) {
            uint8_t l[288], d[30];
            for (int i = 0; i < 144; i++) l[i] = 8;
            for (int i = 144; i < 256; i++) l[i] = 9;
            for (int i = 256; i < 280; i++) l[i] = 7;
            for (int i = 280; i < 288; i++) l[i] = 8;
            for (int i = 0; i < 30; i++) d[i] = 5;
            Huffman lit, dist;
            lit.build(l, 288);
            dist.build(d, 30);
            inflate_block(br, lit, dist, out);
        } else if (type == 2) {
            int hlit = (int)br.bits(5) + 257, hdist = (int)br.bits(5) + 1, hclen = (int)br.bits(4) + 4;
            if (hlit > 286 || hdist > 30) bad("bad dynamic header");
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            uint8_t cl[19] = { 0 };
            for (int i = 0; i < hclen; i++) cl[order[i]] = (uint8_t)br.bits(3);
            Huffman clh;
            clh.build(cl, 19);
            uint8_t lengths[320] = { 0 };
            int i = 0;
            while (i < hlit + hdist) {
                int sym = clh.decode(br);
                if (sym < 16) { lengths[i++] = (uint8_t)sym; continue; }
                int rep; uint8_t val = 0;
                if (sym == 16) { if (i == 0) bad("repeat with no previous length"); val = lengths[i - 1]; rep = 3 + (int)br.bits(2); }
                else if (sym == 17) rep = 3 + (int)br.bits(3);
                else rep = 11 + (int)br.bits(7);
                if (i + rep > hlit + hdist) bad("code lengths overflow");
                while (rep--) lengths[i++] = val;
            }
            if (lengths[256] == 0) bad("no end-of-block code");
            Huffman lit, dist;
            lit.build(lengths, hlit);
            dist.build(lengths + hlit, hdist);
            inflate_block(br, lit, dist, out);
        } else {
            bad("reserved block type");
        }
    } while (!final_block);
    return out;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

}  // namespace

bool looks_like_png(const uint8_t *data, size_t size) {
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    return size >= 8 && std::memcmp(data, sig, 8) == 0;
}

DecodedImage decode_png(const uint8_t *data, size_t size) {
    if (!looks_like_png(data, size)) bad("signature missing");
    size_t off = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool seen_iend = false;
    while (off + 12 <= size && !seen_iend) {
        uint32_t len = be32(data + off);
        const uint8_t *tag = data + off + 4, *body = data + off + 8;
        if ((size_t)len > size - off - 12) bad("chunk runs past end of file");
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len != 13) bad("bad IHDR");
            w = be32(body); h = be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0) bad("unknown compression/filter method");
        } else if (!std::memcmp(tag, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(tag, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(tag, "IEND", 4)) seen_iend = true;
        off += 12 + (size_t)len;
    }
    if (ctype < 0 || w == 0 || h == 0) bad("missing IHDR");
    if (interlace) throw std::runtime_error("PNG: interlaced images are outside the supported subset");
    int channels;
    switch (ctype) {
    case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break;
    case 4: channels = 2; break; case 6: channels = 4; break;
    default: bad("bad colour type");
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) bad("bad bit depth");
    if (ctype == 3 && depth == 16) bad("bad palette depth");
    if ((uint64_t)w * h > (1ull << 28)) bad("image too large");

    size_t bpp_bits = (size_t)channels * depth;
    size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    size_t bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;        // filter distance in bytes
    std::vector<uint8_t> raw = zlib_inflate(idat.data(), idat.size(), (stride + 1) * h);
    if (raw.size() < (stride + 1) * (size_t)h) bad("not enough image data");

    // un-filter in place (row r lives at raw[r*(stride+1)+1 ..])
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t r = 0; r < h; r++) {
        uint8_t *cur = raw.data() + (size_t)r * (stride + 1) + 1;
        const uint8_t *up = r ? cur - (stride + 1) : zero.data();
        int ft = cur[-1];
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int x = cur[i];
            switch (ft) {
            case 0: break;
            case 1: x += a; break;
            case 2: x += b; break;
            case 3: x += (a + b) >> 1; break;
            case 4: x += paeth(a, b, c); break;
            default: bad("bad filter type");
            }
            cur[i] = (uint8_t)x;
        }
    }

    // expand to 8-bit samples with stb_image's channel-count convention
    DecodedImage img;
    img.width = (int)w; img.height = (int)h;
    const bool pal = ctype == 3;
    bool key = false;
    uint16_t key_rgb[3] = { 0, 0, 0 };
    if (pal) {
        if (plte.empty() || plte.size() % 3) bad("palette missing");
        img.components = trns.empty() ? 3 : 4;
    } else if (!trns.empty() && (ctype == 0 || ctype == 2)) {
        size_t need = ctype == 0 ? 2 : 6;
        if (trns.size() < need) bad("short tRNS");
        for (size_t k = 0; k < need / 2; k++) key_rgb[k] = (uint16_t)((trns[2 * k] << 8) | trns[2 * k + 1]);
        key = true;
        img.components = channels + 1;
    } else {
        img.components = channels;
    }
    img.texels.resize((size_t)w * h * img.components);
    static const uint8_t scale[9] = { 0, 0xFF, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
    for (uint32_t r = 0; r < h; r++) {
        const uint8_t *row = raw.data() + (size_t)r * (stride + 1) + 1;
        uint8_t *dst = img.texels.data() + (size_t)r * w * img.components;
        for (uint32_t x = 0; x < w; x++) {
            uint16_t s[4] = { 0, 0, 0, 0 };
            for (int ch = 0; ch < channels; ch++) {
                if (depth == 8) s[ch] = row[(size_t)x * channels + ch];
                else if (depth == 16) s[ch] = (uint16_t)((row[((size_t)x * channels + ch) * 2] << 8) | row[((size_t)x * channels + ch) * 2 + 1]);
                else {
                    size_t bit = (size_t)x * depth;
                    s[ch] = (uint16_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1));
                }
            }
            if (pal) {
                size_t idx = s[0];
                if (idx * 3 + 2 >= plte.size()) idx = 0;   // out-of-range index: stb reads whatever follows; keep it defined
                dst[0] = plte[idx * 3]; dst[1] = plte[idx * 3 + 1]; dst[2] = plte[idx * 3 + 2];
                if (img.components == 4) dst[3] = idx < trns.size() ? trns[idx] : 255;
            } else {
                bool transparent = key;
                for (int ch = 0; ch < channels; ch++) {
                    if (key && s[ch] != key_rgb[ch]) transparent = false;
                    dst[ch] = depth == 16 ? (uint8_t)(s[ch] >> 8) : (depth == 8 ? (uint8_t)s[ch] : (uint8_t)(s[ch] * scale[depth]));
                }
                if (key) dst[channels] = transparent ? 0 : 255;
            }
            dst += img.components;
        }
    }
    return img;
}

}  // namespace drt
