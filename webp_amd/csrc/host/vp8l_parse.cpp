// Host half of the lossless path: VP8L header, transform descriptors and the serial
// Huffman / LZ77 / colour-cache pixel decode.  Mirrors vp8l/Vp8LDecoder.hx (decodeHeader :347-360,
// decodeTransform :78-122, decodeHuffmanGroups :185-220, decodePix :222-335) and vp8l/HTree.hx.
// The inverse transforms themselves (vp8l/Transform.hx) run on the GPU; this file only records
// them, in bitstream order, together with their tile / palette data.
#include "host.h"
#include "vp8_data.h"
#include <cstring>

namespace wp {
namespace {

struct Fail { int code; const char* msg; };   // internal unwinding, never crosses the C ABI

// LSB-first bit reader over one or two concatenated buffers (Vp8LDecoder.read :66-76, ConcatInput).
class BitSource {
public:
    BitSource(const uint8_t* a, size_t alen, const uint8_t* b, size_t blen) : a_(a), alen_(alen), b_(b), blen_(blen) {}
    // returns false on end of input
    bool next_byte(int& v) {
        if (pos_ < alen_) { v = a_[pos_++]; return true; }
        const size_t k = pos_ - alen_;
        if (k < blen_) { v = b_[k]; ++pos_; return true; }
        return false;
    }
    uint32_t read(int n) {
        while (nbits < n) {
            int c;
            if (!next_byte(c)) throw Fail{WP_ERR_EOF, "Eof"};
            bits |= int32_t(c & 0xFF) << nbits;
            nbits += 8;
        }
        const uint32_t u = uint32_t(bits) & ((1u << n) - 1);
        bits >>= n;
        nbits -= n;
        return u;
    }
    int32_t bits = 0;
    int nbits = 0;
private:
    const uint8_t* a_; size_t alen_; const uint8_t* b_; size_t blen_; size_t pos_ = 0;
};

// vp8l/HTree.hx: binary tree + 7-bit lookup table.  Kept structurally identical so that even
// degenerate code-length sets decode to the same symbols as the reference.
class Huffman {
public:
    Huffman() : nodes_(1, Node{0, 0}) { std::memset(lut_, 0, sizeof lut_); }

    void build(const std::vector<int>& lengths) {                        // HTree.hx:91-115
        int n_sym = 0, last = 0;
        for (size_t s = 0; s < lengths.size(); ++s) if (lengths[s]) { ++n_sym; last = int(s); }
        if (n_sym == 0) throw Fail{WP_ERR_FORMAT, "Invalid Huffman tree"};
        nodes_.assign(1, Node{0, 0});
        if (n_sym == 1) { insert(last, 0, 0); return; }
        int max_len = 0;
        for (int l : lengths) if (l > max_len) max_len = l;
        if (lengths.empty() || max_len > 15) throw Fail{WP_ERR_FORMAT, "Invalid Huffman tree"};
        int hist[16] = {0}, next[16] = {0};
        for (int l : lengths) ++hist[l];
        int code = 0;
        for (int l = 1; l < 16; ++l) { code = (code + hist[l - 1]) << 1; next[l] = code; }   // HTree.hx:188-191, hist[0] included
        std::vector<int> codes(lengths.size(), 0);
        for (size_t s = 0; s < lengths.size(); ++s) if (lengths[s] > 0) codes[s] = next[lengths[s]]++;
        for (size_t s = 0; s < lengths.size(); ++s) if (lengths[s] > 0) insert(int(s), codes[s], lengths[s]);
    }
    void build_simple(int n_sym, const int* syms, int alphabet) {        // HTree.hx:118-128
        nodes_.assign(1, Node{0, 0});
        for (int i = 0; i < n_sym; ++i) {
            if (syms[i] >= alphabet) throw Fail{WP_ERR_FORMAT, "Invalid huffman tree"};
            insert(syms[i], i, n_sym - 1);
        }
    }
    int next(BitSource& d) const {                                       // HTree.hx:130-169
        if (d.nbits < 7) {
            int c;
            if (!d.next_byte(c)) return walk(d, 0);
            d.bits |= c << d.nbits;
            d.nbits += 8;
        }
        int n = lut_[d.bits & 127];
        if ((n & 0xFF) != 0) {
            const int b = (n & 0xFF) - 1;
            d.bits >>= b;
            d.nbits -= b;
            return n >> 8;
        }
        n >>= 8;
        d.bits >>= 7;
        d.nbits -= 7;
        return walk(d, n);
    }

private:
    struct Node { int symbol; int children; };   // children: 0 = unset, -1 = leaf
    static int rev8(int v) { int r = 0; for (int i = 0; i < 8; ++i) if (v & (1 << i)) r |= 0x80 >> i; return r; }

    void insert(int symbol, int code, int len) {                         // HTree.hx:45-89
        if (symbol > 0xFFFF || len > 0xFE) throw Fail{WP_ERR_FORMAT, "Invalid Huffman tree"};
        int base;
        if (len > 7) {
            base = rev8((code >> (len - 7)) & 0xFF) >> 1;
        } else {
            base = rev8(code & 0xFF) >> (8 - len);
            for (int i = 0; i < (1 << (7 - len)); ++i) lut_[base | (i << len)] = (symbol << 8) | (len + 1);
        }
        int n = 0, jump = 7;
        while (len > 0) {
            --len;
            if (n >= int(nodes_.size())) throw Fail{WP_ERR_FORMAT, "Invalid Huffman tree"};
            if (nodes_[n].children == -1) throw Fail{WP_ERR_FORMAT, "Invalid Huffman tree"};
            if (nodes_[n].children == 0) {
                nodes_[n].children = int(nodes_.size());
                nodes_.push_back(Node{0, 0});
                nodes_.push_back(Node{0, 0});
            }
            n = nodes_[n].children + ((code >> len) & 1);
            if (--jump == 0 && lut_[base] == 0) lut_[base] = n << 8;
        }
        if (nodes_[n].children == 0) nodes_[n].children = -1;
        else if (nodes_[n].children != -1) throw Fail{WP_ERR_FORMAT, "Invalid Huffman tree"};
        nodes_[n].symbol = symbol;
    }
    int walk(BitSource& d, int n) const {
        while (nodes_[n].children != -1) {
            if (d.nbits == 0) {
                int c;
                if (!d.next_byte(c)) throw Fail{WP_ERR_EOF, "Eof"};
                d.bits = c;
                d.nbits = 8;
            }
            n = nodes_[n].children + (d.bits & 1);
            d.bits >>= 1;
            --d.nbits;
        }
        return nodes_[n].symbol;
    }
    std::vector<Node> nodes_;
    int lut_[128];
};

inline int n_tiles(int dim, int bits) { return (dim + (1 << bits) - 1) >> bits; }   // Transform.hx:23-25

class Vp8lParser {
public:
    Vp8lParser(const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, bool tokens) : d_(a, alen, b, blen), tokens_(tokens) {}

    void run(Vp8lPacked& out) {                                          // Vp8LDecoder.hx:347-393
        if (d_.read(8) != 0x2f) throw Fail{WP_ERR_FORMAT, "vp8l: invalid header"};
        int w = int(d_.read(14)) + 1;
        const int h = int(d_.read(14)) + 1;
        d_.read(1);
        if (d_.read(3) != 0) throw Fail{WP_ERR_FORMAT, "vp8l: invalid version"};
        out.width = w;
        out.height = h;
        bool seen[4] = {false, false, false, false};
        while (d_.read(1)) {
            Vp8lTransform t;
            transform(w, h, t);
            if (seen[t.type]) throw Fail{WP_ERR_FORMAT, "vp8l: repeated transform"};
            seen[t.type] = true;
            out.transforms.push_back(std::move(t));
        }
        out.coded_width = w;
        if (tokens_) {
            out.tokenized = true;
            tokenize(w, h, out);
        } else {
            pixels(w, h, true, out.pix);
        }
    }

private:
    void transform(int& w, int h, Vp8lTransform& t) {                    // Vp8LDecoder.hx:78-122
        t.old_width = w;
        t.type = int(d_.read(2));
        switch (t.type) {
        case 0: case 1:
            t.bits = int(d_.read(3)) + 2;
            pixels(n_tiles(w, t.bits), n_tiles(h, t.bits), false, t.pix);
            break;
        case 2:
            break;
        default: {
            const int n_colors = int(d_.read(8)) + 1;
            t.bits = n_colors <= 2 ? 3 : n_colors <= 4 ? 2 : n_colors <= 16 ? 1 : 0;
            const int full = 1 << (8 >> t.bits);
            w = n_tiles(w, t.bits);
            std::vector<uint8_t> delta;
            pixels(n_colors, 1, false, delta);
            t.pix.assign(size_t(4 * full), 0);
            for (int p = 0; p < 4 * n_colors; ++p)
                t.pix[size_t(p)] = uint8_t(delta[size_t(p)] + (p >= 4 ? t.pix[size_t(p - 4)] : 0));
            break; }
        }
    }

    void code_lengths(std::vector<int>& dst, const std::vector<int>& clcl) {   // Vp8LDecoder.hx:124-158
        static const int kRepBits[3] = {2, 3, 7}, kRepOff[3] = {3, 3, 11};
        Huffman h;
        h.build(clcl);
        int max_symbol = int(dst.size());
        if (d_.read(1)) {
            const int n = 2 + 2 * int(d_.read(3));
            max_symbol = int(d_.read(n)) + 2;
            if (max_symbol > int(dst.size())) throw Fail{WP_ERR_FORMAT, "Invalid code lengths"};
        }
        int prev = 8;
        size_t sym = 0;
        while (sym < dst.size()) {
            if (max_symbol == 0) break;
            --max_symbol;
            const int cl = h.next(d_);
            if (cl < 16) {
                dst[sym++] = cl;
                if (cl) prev = cl;
                continue;
            }
            const int rep = int(d_.read(kRepBits[cl - 16])) + kRepOff[cl - 16];
            if (sym + size_t(rep) > dst.size()) throw Fail{WP_ERR_FORMAT, "Invalid code lengths"};
            const int v = cl == 16 ? prev : 0;
            for (int i = 0; i < rep; ++i) dst[sym++] = v;
        }
    }

    void tree(Huffman& h, int alphabet) {                                // Vp8LDecoder.hx:160-183
        static const uint8_t kOrder[19] = {17, 18, 0, 1, 2, 3, 4, 5, 16, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
        if (d_.read(1)) {
            const int n_sym = int(d_.read(1)) + 1;
            const int first_len = int(d_.read(1)) * 7 + 1;
            int syms[2] = {0, 0};
            syms[0] = int(d_.read(first_len));
            if (n_sym == 2) syms[1] = int(d_.read(8));
            h.build_simple(n_sym, syms, alphabet);
            return;
        }
        const int n_codes = int(d_.read(4)) + 4;
        if (n_codes > 19) throw Fail{WP_ERR_FORMAT, "Invalid Huffman tree"};
        std::vector<int> clcl(19, 0);
        for (int i = 0; i < n_codes; ++i) clcl[kOrder[i]] = int(d_.read(3));
        std::vector<int> lengths(size_t(alphabet), 0);
        code_lengths(lengths, clcl);
        h.build(lengths);
    }

    uint32_t lz77(uint32_t sym) {                                        // Vp8LDecoder.hx:337-345
        if (sym < 4) return sym + 1;
        const uint32_t extra = (sym - 2) >> 1;
        const uint32_t off = (2 + (sym & 1)) << extra;
        return off + d_.read(int(extra)) + 1;
    }
    static int distance(int w, int code) {                               // Vp8LDecoder.hx:50-59
        if (code > 120) return code - 120;
        const int dc = wpd_DistanceMap[code - 1];
        const int d = (dc >> 4) * w + (8 - (dc & 0xF));
        return d >= 1 ? d : 1;
    }

    struct Group { Huffman t[5]; };

    // What decodePix sets up before its pixel loop (Vp8LDecoder.hx:185-252): colour-cache size, the meta image that picks
    // a Huffman group per tile, the groups themselves.
    struct PixelCodes {
        int cc_bits = 0;
        int h_bits = 0, h_mask = 0, tiles_per_row = 0;
        std::vector<uint8_t> meta;
        std::vector<Group> groups;
        const Group* at(int x, int y) const {
            const size_t i = size_t(4) * size_t(tiles_per_row * (y >> h_bits) + (x >> h_bits));
            return &groups[size_t((meta[i] << 8) | meta[i + 1])];
        }
    };
    void pixel_codes(int w, int h, bool top, PixelCodes& c) {
        static const int kAlphabet[5] = {256 + 24, 256, 256, 256, 40};
        if (d_.read(1)) {
            c.cc_bits = int(d_.read(4));
            if (c.cc_bits < 1 || c.cc_bits > 11) throw Fail{WP_ERR_FORMAT, "Invalid color cache parameters"};
        }
        int max_group = 0;
        if (top && d_.read(1)) {
            c.h_bits = int(d_.read(3)) + 2;
            pixels(n_tiles(w, c.h_bits), n_tiles(h, c.h_bits), false, c.meta);
            for (size_t p = 0; p < c.meta.size(); p += 4) {
                const int g = (c.meta[p] << 8) | c.meta[p + 1];
                if (g > max_group) max_group = g;
            }
        }
        c.groups.resize(size_t(max_group) + 1);
        for (Group& g : c.groups)
            for (int j = 0; j < 5; ++j) tree(g.t[j], kAlphabet[j] + ((j == 0 && c.cc_bits > 0) ? (1 << c.cc_bits) : 0));
        c.h_mask = c.h_bits ? (1 << c.h_bits) - 1 : 0;
        c.tiles_per_row = c.h_bits ? n_tiles(w, c.h_bits) : 0;
    }

    void pixels(int w, int h, bool top, std::vector<uint8_t>& pix) {     // Vp8LDecoder.hx:185-335
        PixelCodes c;
        pixel_codes(w, h, top, c);
        const int cc_shift = 32 - c.cc_bits;
        std::vector<uint32_t> cache;
        if (c.cc_bits) cache.assign(size_t(1) << c.cc_bits, 0);
        const int h_mask = c.h_mask;
        const int64_t total = int64_t(4) * w * h;
        pix.assign(size_t(total), 0);
        int64_t p = 0, cached = 0;
        int x = 0, y = 0;
        const Group* hg = &c.groups[0];
        bool lookup = h_mask != 0;
        while (p < total) {
            if (lookup) hg = c.at(x, y);
            int green = hg->t[0].next(d_);
            if (green < 256) {
                const int red = hg->t[1].next(d_);
                const int blue = hg->t[2].next(d_);
                const int alpha = hg->t[3].next(d_);
                pix[size_t(p)] = uint8_t(red); pix[size_t(p + 1)] = uint8_t(green);
                pix[size_t(p + 2)] = uint8_t(blue); pix[size_t(p + 3)] = uint8_t(alpha);
                p += 4;
                if (++x == w) { x = 0; ++y; }
                lookup = h_mask != 0 && (x & h_mask) == 0;
            } else if (green < 256 + 24) {
                const int length = int(lz77(uint32_t(green - 256)));
                const int dist_code = int(lz77(uint32_t(hg->t[4].next(d_))));
                const int dist = distance(w, dist_code);
                const int64_t p_end = p + int64_t(4) * length;
                int64_t q = p - int64_t(4) * dist;
                const int64_t q_end = p_end - int64_t(4) * dist;
                if (p < 0 || q < 0 || p_end > total || q_end > total) throw Fail{WP_ERR_FORMAT, "Invalid LZ77 parameters"};
                while (p < p_end) pix[size_t(p++)] = pix[size_t(q++)];
                x += length;
                while (x >= w) { x -= w; ++y; }
                lookup = h_mask != 0;
            } else {
                if (cache.empty()) throw Fail{WP_ERR_NULL, "null color cache"};
                for (; cached < p; cached += 4) {
                    const uint32_t argb = uint32_t(pix[size_t(cached)]) << 16 | uint32_t(pix[size_t(cached + 1)]) << 8 |
                                          uint32_t(pix[size_t(cached + 2)]) | uint32_t(pix[size_t(cached + 3)]) << 24;
                    cache[(argb * 0x1e35a7bdu) >> cc_shift] = argb;
                }
                green -= 256 + 24;
                if (green >= int(cache.size())) throw Fail{WP_ERR_FORMAT, "Invalid color cache index"};
                const uint32_t argb = cache[size_t(green)];
                pix[size_t(p)] = uint8_t(argb >> 16); pix[size_t(p + 1)] = uint8_t(argb >> 8);
                pix[size_t(p + 2)] = uint8_t(argb); pix[size_t(p + 3)] = uint8_t(argb >> 24);
                p += 4;
                if (++x == w) { x = 0; ++y; }
                lookup = h_mask != 0 && (x & h_mask) == 0;
            }
        }
    }

    // The same walk over the top-level image with the symbols written down instead of acted on: literal runs, copies
    // (length, distance in pixels) and colour-cache indices, in the token format of webp_amd.h.  Every check decodePix makes
    // on the way (Vp8LDecoder.hx:283-289, 308-316) depends on positions and indices only, never on pixel values, so the
    // same inputs fail here with the same messages.
    void tokenize(int w, int h, Vp8lPacked& out) {
        PixelCodes c;
        pixel_codes(w, h, true, c);
        out.cc_bits = c.cc_bits;
        std::vector<uint32_t>& tok = out.tokens;
        const int h_mask = c.h_mask;
        const int64_t total = int64_t(w) * h;
        tok.reserve(size_t(total / 2) + 16);
        int64_t p = 0;
        size_t run = SIZE_MAX;                   // index of the open literal run's header word
        int x = 0, y = 0;
        const Group* hg = &c.groups[0];
        bool lookup = h_mask != 0;
        while (p < total) {
            if (lookup) hg = c.at(x, y);
            const int green = hg->t[0].next(d_);
            if (green < 256) {
                const uint32_t red = uint32_t(hg->t[1].next(d_)) & 0xFF;
                const uint32_t blue = uint32_t(hg->t[2].next(d_)) & 0xFF;
                const uint32_t alpha = uint32_t(hg->t[3].next(d_)) & 0xFF;
                if (run == SIZE_MAX) { run = tok.size(); tok.push_back(WP_TOK_LITERALS << 30); }
                tok.push_back(red | uint32_t(green) << 8 | blue << 16 | alpha << 24);     // R,G,B,A in memory order
                ++tok[run];
                ++p;
                if (++x == w) { x = 0; ++y; }
                lookup = h_mask != 0 && (x & h_mask) == 0;
            } else if (green < 256 + 24) {
                const int length = int(lz77(uint32_t(green - 256)));
                const int dist_code = int(lz77(uint32_t(hg->t[4].next(d_))));
                const int dist = distance(w, dist_code);
                if (int64_t(dist) > p || p + length > total) throw Fail{WP_ERR_FORMAT, "Invalid LZ77 parameters"};
                run = SIZE_MAX;
                tok.push_back(WP_TOK_COPY << 30 | uint32_t(length));
                tok.push_back(uint32_t(dist));
                p += length;
                x += length;
                while (x >= w) { x -= w; ++y; }
                lookup = h_mask != 0;
            } else {
                if (c.cc_bits == 0) throw Fail{WP_ERR_NULL, "null color cache"};
                const int index = green - (256 + 24);
                if (index >= (1 << c.cc_bits)) throw Fail{WP_ERR_FORMAT, "Invalid color cache index"};
                run = SIZE_MAX;
                tok.push_back(WP_TOK_CACHE << 30 | uint32_t(index));
                ++p;
                if (++x == w) { x = 0; ++y; }
                lookup = h_mask != 0 && (x & h_mask) == 0;
            }
        }
    }

    BitSource d_;
    bool tokens_;
};

}  // namespace

int parse_vp8l(const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, Vp8lPacked& out, std::string& err, bool tokens) {
    try {
        Vp8lParser p(a, alen, b, blen, tokens);
        p.run(out);
    } catch (const Fail& f) {
        err = f.msg;
        return f.code;
    } catch (const std::bad_alloc&) {
        err = "out of memory";
        return WP_ERR_NOMEM;
    }
    return WP_OK;
}

}  // namespace wp
