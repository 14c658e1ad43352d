// Host half of the lossy path: VP8 frame-header parse + boolean-coder entropy decode, emitting the
// packed per-macroblock mode / coefficient / loop-filter buffers declared in include/webp_amd.h.
//
// Mirrors the *parse* half of the reference's vp8/Vp8Decoder.hx (decodeFrameHeader :126-184,
// parseOtherHeaders :289-324, reconstruct's prologue :772-808, parsePredMode* :838-903,
// parseResiduals/parseResiduals4 :579-714) and vp8/Partition.hx.  Pixel work (prepareYBR,
// predFunc*, inverseDCT*, filters) is NOT done here: it runs on the GPU.  Reference behaviours that
// change results are kept on purpose and marked Q1..Q3.
#include "host.h"
#include "vp8_data.h"
#include <cstring>

namespace wp {
namespace {

// vp8/Partition.hx:7-113.  `eof` reproduces unexpectedEOF: a read past the end yields false.
// Partition.hx (the boolean decoder).  Same arithmetic and the same end-of-data behaviour as the reference (a bit that
// needs a byte past the end sets the EOF flag and reads as 0, leaving the state untouched), but with a 64-bit window:
// bytes are shifted in up to five at a time instead of one per refill, and the normalisation tables are extended to
// 256 entries (shift 0 above 126) so the per-bit path has no second branch.  `v` holds `nb` not yet consumed bits,
// right-aligned; the decoder's 8-bit window is v >> (nb - 8).
// One degenerate case needs care: a partition whose first byte is 0xFF starts with value == range, which no later
// step repairs; every bit then reads 1 and the reference's 32-bit `bits` grows until it wraps.  The comparison below
// is therefore made on the same 32-bit quantity (window << 8, wrapped, signed), and the refill stops at 48 bits so that
// the 24 window bits the reference keeps always lie inside `v`.
struct BoolReader {
    const uint8_t* p = nullptr;
    const uint8_t* end = nullptr;
    uint64_t v = 0;
    int32_t range_m1 = 254;
    int nb = 0;
    bool eof = false;

    void open(const uint8_t* b, size_t n) { p = b; end = b + n; range_m1 = 254; v = 0; nb = 0; eof = false; }

    struct Norm { uint16_t e[256]; };        // shift | new range-1 << 8
    static const Norm& norm() {
        static const Norm t = [] {
            Norm n{};
            for (int i = 0; i < 256; ++i) n.e[i] = uint16_t(i < 127 ? (wpd_BoolShift[i] | (wpd_BoolRange[i] << 8)) : (i << 8));
            return n;
        }();
        return t;
    }

    inline bool bit(int prob) {
        if (__builtin_expect(nb < 8, 0)) {
            if (end - p >= 4) {                       // 4 bytes at once: nb < 8 before, so at most 39 bits afterwards
                uint32_t w;
                std::memcpy(&w, p, 4);
                v = (v << 32) | __builtin_bswap32(w);
                p += 4; nb += 32;
            } else {
                while (nb <= 40 && p < end) { v = (v << 8) | *p++; nb += 8; }
                if (nb < 8) { eof = true; return false; }
            }
        }
        const uint32_t split = ((uint32_t(range_m1) * uint32_t(prob)) >> 8) + 1;
        const int pos = nb - 8;
        const bool b = int32_t(uint32_t(v >> pos) << 8) >= int32_t(split << 8);
        if (b) { range_m1 -= int32_t(split); v -= uint64_t(split) << pos; } else { range_m1 = int32_t(split) - 1; }
        const uint32_t e = norm_->e[range_m1];
        nb -= int(e & 0xff);
        range_m1 = int32_t(e >> 8);
        return b;
    }
    inline int uint(int prob, int n) { int u = 0; for (int i = n - 1; i >= 0; --i) if (bit(prob)) u |= 1 << i; return u; }
    inline int sint(int prob, int n) { const int u = uint(prob, n); return bit(prob) ? -u : u; }
    inline int opt_int(int prob, int n) { return bit(prob) ? sint(prob, n) : 0; }

    const Norm* norm_ = &norm();
};

constexpr int kUniform = 128;
enum { kDC = 0, kTM, kVE, kHE, kRD, kVR, kLD, kVL, kHD, kHU };
enum { kPlaneY1WithY2 = 0, kPlaneY2 = 1, kPlaneUV = 2, kPlaneY1SansY2 = 3 };

const uint8_t kBand[17] = {0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 0};
const uint8_t kZig[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
const uint8_t kCat[4][12] = {{173, 148, 140, 0}, {176, 155, 140, 135, 0}, {180, 157, 141, 134, 130, 0},
                             {254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129, 0}};

struct Dequant { int y1[2], y2[2], uv[2]; };
struct LoopParam { int level = 0, ilevel = 0, hlevel = 0; bool inner = false; };
struct NzCtx { uint8_t pred[4] = {0, 0, 0, 0}; uint8_t nz_mask = 0; uint8_t nz_y16 = 0; };

class Vp8Parser {
public:
    Vp8Parser(const uint8_t* d, size_t n, Vp8Packed& o) : data_(d), len_(n), out_(o) {}

    int run(std::string& err) {
        int rc = frame_header(err);
        if (rc) return rc;
        rc = other_headers(err);
        if (rc) return rc;
        const int nmb = out_.mbw * out_.mbh;
        out_.mb.assign(size_t(nmb) * 4, 0);
        out_.filt.assign(size_t(nmb), 0);
        out_.coeff.clear();
        out_.coeff.reserve(size_t(nmb) * 16 * 14);
        std::vector<uint8_t> which(size_t(nmb), 0);   // (segment << 1) | i4, resolved after the frame (Q1)
        up_.assign(size_t(out_.mbw), NzCtx());
        for (int mby = 0; mby < out_.mbh; ++mby) {
            left_ = NzCtx();
            for (int mbx = 0; mbx < out_.mbw; ++mbx) {
                const int idx = mby * out_.mbw + mbx;
                bool i16 = false;
                const bool skip = macroblock(mbx, mby, &out_.mb[size_t(idx) * 4], &i16);
                // Vp8Decoder.hx:350-352: the record is SHARED per (segment, i16/i4); `inner` is sticky.
                LoopParam& fs = loop_[segment_][i16 ? 0 : 1];
                fs.inner = fs.inner || !skip;
                which[idx] = uint8_t((segment_ << 1) | (i16 ? 0 : 1));
                out_.n_i4 += i16 ? 0 : 1;
            }
        }
        if (first_.eof) { err = "vp8: decodeFrame returned null"; return WP_ERR_NULL; }
        for (int i = 0; i < n_parts_; ++i)
            if (part_[i].eof) { err = "vp8: decodeFrame returned null"; return WP_ERR_NULL; }
        for (int i = 0; i < nmb; ++i) {
            const LoopParam& f = loop_[which[i] >> 1][which[i] & 1];
            out_.filt[i] = uint32_t(f.level & 0xff) | uint32_t(f.ilevel & 0xff) << 8 | uint32_t(f.hlevel & 0xff) << 16 |
                           uint32_t(f.inner ? 1 : 0) << 24;
        }
        out_.filter_type = filter_level_ == 0 ? 0 : (filter_simple_ ? 1 : 2);
        return WP_OK;
    }

private:
    // Vp8Decoder.hx:126-184
    int frame_header(std::string& err) {
        if (len_ - pos_ < 3) { err = "Eof"; return WP_ERR_EOF; }
        const uint8_t* b = data_ + pos_; pos_ += 3;
        out_.keyFrame = (b[0] & 1) == 0;
        out_.versionNumber = (b[0] >> 1) & 7;
        out_.showFrame = ((b[0] >> 4) & 1) == 1;
        out_.firstPartitionLen = (b[0] >> 5) | (b[1] << 3) | (b[2] << 11);
        out_.width = out_.height = out_.xScale = out_.yScale = 0;
        out_.mbw = out_.mbh = 0;
        if (!out_.keyFrame) return WP_OK;
        if (len_ - pos_ < 7) { err = "Eof"; return WP_ERR_EOF; }
        b = data_ + pos_; pos_ += 7;
        if (b[0] != 0x9d || b[1] != 0x01 || b[2] != 0x2a) { err = "vp8: invalid format"; return WP_ERR_FORMAT; }
        out_.width = ((b[4] & 0x3f) << 8) | b[3];
        out_.height = ((b[6] & 0x3f) << 8) | b[5];
        out_.xScale = b[4] >> 6;
        out_.yScale = b[6] >> 6;
        out_.mbw = (out_.width + 15) >> 4;
        out_.mbh = (out_.height + 15) >> 4;
        std::memcpy(prob_, wpd_TokenProbDefault, sizeof prob_);
        return WP_OK;
    }

    // Vp8Decoder.hx:203-227.  The Int8Vector fields read back as unsigned bytes (Q2).
    void segment_header() {
        use_segment_ = first_.bit(kUniform);
        if (!use_segment_) { update_map_ = false; return; }
        update_map_ = first_.bit(kUniform);
        if (first_.bit(kUniform)) {
            relative_delta_ = !first_.bit(kUniform);
            for (int i = 0; i < 4; ++i) seg_quant_[i] = uint8_t(first_.opt_int(kUniform, 7));
            for (int i = 0; i < 4; ++i) seg_strength_[i] = uint8_t(first_.opt_int(kUniform, 6));
        }
        if (!update_map_) return;
        for (int i = 0; i < 3; ++i) seg_prob_[i] = first_.bit(kUniform) ? uint8_t(first_.uint(kUniform, 8)) : 0xff;
    }

    // Vp8Decoder.hx:229-258 + computeFilterParams :488-539
    void filter_header() {
        filter_simple_ = first_.bit(kUniform);
        filter_level_ = first_.uint(kUniform, 6);
        sharpness_ = first_.uint(kUniform, 3);
        use_lf_delta_ = first_.bit(kUniform);
        if (use_lf_delta_ && first_.bit(kUniform)) {
            for (int i = 0; i < 4; ++i) ref_delta_[i] = uint8_t(first_.opt_int(kUniform, 6));
            for (int i = 0; i < 4; ++i) mode_delta_[i] = uint8_t(first_.opt_int(kUniform, 6));
        }
        if (filter_level_ == 0) return;
        for (int s = 0; s < 4; ++s) {
            int base = filter_level_;
            if (use_segment_) {
                base = seg_strength_[s];
                if (relative_delta_) base += filter_level_;
            }
            for (int j = 0; j < 2; ++j) {
                LoopParam& p = loop_[s][j];
                p.inner = j != 0;
                int level = base;
                if (use_lf_delta_) {
                    level += ref_delta_[0];
                    if (j) level += mode_delta_[0];
                }
                if (level <= 0) { p.level = 0; continue; }
                if (level > 63) level = 63;
                int il = level;
                if (sharpness_ > 0) {
                    il >>= sharpness_ > 4 ? 2 : 1;
                    if (il > 9 - sharpness_) il = 9 - sharpness_;
                }
                if (il < 1) il = 1;
                p.ilevel = il;
                p.level = 2 * level + il;
                p.hlevel = out_.keyFrame ? (level < 15 ? 0 : level < 40 ? 1 : 2)
                                         : (level < 15 ? 0 : level < 20 ? 1 : level < 40 ? 2 : 3);
            }
        }
    }

    // Vp8Decoder.hx:260-287.  Every DCT partition starts at offset 0 of the remaining data (Q3).
    bool token_partitions() {
        int part_len[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        n_parts_ = 1 << first_.uint(kUniform, 2);
        const int n = 3 * (n_parts_ - 1);
        part_len[n_parts_ - 1] = int(len_ - pos_) - n;
        if (part_len[n_parts_ - 1] < 0) return false;
        if (n > 0) {
            const uint8_t* b = data_ + pos_; pos_ += size_t(n);
            for (int i = 0; i < n_parts_ - 1; ++i) {
                const int pl = b[3 * i] | (b[3 * i + 1] << 8) | (b[3 * i + 2] << 16);
                if (pl > part_len[n_parts_ - 1]) return false;
                part_len[i] = pl;
                part_len[n_parts_ - 1] -= pl;
            }
        }
        if ((1 << 24) <= part_len[n_parts_ - 1]) return false;
        const uint8_t* rest = data_ + pos_;
        pos_ = len_;
        for (int i = 0; i < n_parts_; ++i) part_[i].open(rest, size_t(part_len[i]));
        return true;
    }

    static int clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

    // Vp8Decoder.hx:382-415
    void quant_header() {
        const int base = first_.uint(kUniform, 7);
        const int dy1dc = first_.opt_int(kUniform, 4);
        const int dy2dc = first_.opt_int(kUniform, 4);
        const int dy2ac = first_.opt_int(kUniform, 4);
        const int duvdc = first_.opt_int(kUniform, 4);
        const int duvac = first_.opt_int(kUniform, 4);
        for (int s = 0; s < 4; ++s) {
            int q = base;
            if (use_segment_) q = relative_delta_ ? q + seg_quant_[s] : seg_quant_[s];
            Dequant& d = dq_[s];
            d.y1[0] = wpd_DequantDC[clamp(q + dy1dc, 0, 127)];
            d.y1[1] = wpd_DequantAC[clamp(q, 0, 127)];
            d.y2[0] = wpd_DequantDC[clamp(q + dy2dc, 0, 127)] * 2;
            d.y2[1] = wpd_DequantAC[clamp(q + dy2ac, 0, 127)] * 155 / 100;
            if (d.y2[1] < 8) d.y2[1] = 8;
            d.uv[0] = wpd_DequantDC[clamp(q + duvdc, 0, 117)];
            d.uv[1] = wpd_DequantAC[clamp(q + duvac, 0, 127)];
        }
    }

    // Vp8Decoder.hx:289-324
    int other_headers(std::string& err) {
        if (len_ - pos_ < size_t(out_.firstPartitionLen)) { err = "Eof"; return WP_ERR_EOF; }
        first_.open(data_ + pos_, size_t(out_.firstPartitionLen));
        pos_ += size_t(out_.firstPartitionLen);
        if (out_.keyFrame) { first_.bit(kUniform); first_.bit(kUniform); }
        segment_header();
        filter_header();
        const char* null_msg = "vp8: decodeFrame returned null";
        if (!token_partitions()) { err = null_msg; return WP_ERR_NULL; }
        quant_header();
        if (!out_.keyFrame) { err = null_msg; return WP_ERR_NULL; }
        first_.bit(kUniform);
        for (size_t i = 0; i < sizeof prob_; ++i)      // parseTokenProb :326-333
            if (first_.bit(wpd_TokenProbUpdate[i])) (&prob_[0][0][0][0])[i] = uint8_t(first_.uint(kUniform, 8));
        use_skip_prob_ = first_.bit(kUniform);
        if (use_skip_prob_) skip_prob_ = first_.uint(kUniform, 8);
        if (first_.eof) { err = null_msg; return WP_ERR_NULL; }
        return WP_OK;
    }

    // parseResiduals4, Vp8Decoder.hx:579-637.  Writes dequantised, 16-bit-truncated values in
    // natural order into blk[16]; returns the "non-zero" context flag (first EOB test).
    // *stored: OR of the values written (0: the block is still all zero -- a product can truncate to 0).
    int tokens(BoolReader& reader, int plane, int ctx, const int* q, bool skip_first, int16_t* blk, int* stored) {
        const uint8_t (*prob)[3][11] = prob_[plane];
        *stored = 0;
        int n = skip_first ? 1 : 0;
        const uint8_t* p = prob[kBand[n]][ctx];
        if (!reader.bit(p[0])) return 0;
        // The decoder state works in registers from here on: through the reference its fields would be re-read and written
        // back around every store to blk / *stored (an int* may alias them).
        BoolReader r = reader;
        int any = 0;
        const int q0 = q[0], q1 = q[1];
        while (n != 16) {
            ++n;
            if (!r.bit(p[1])) { p = prob[kBand[n]][0]; continue; }
            int v;
            if (!r.bit(p[2])) {
                v = 1;
                p = prob[kBand[n]][1];
            } else {
                if (!r.bit(p[3])) {
                    if (!r.bit(p[4])) v = 2; else v = 3 + r.uint(p[5], 1);
                } else if (!r.bit(p[6])) {
                    if (!r.bit(p[7])) v = 5 + r.uint(159, 1);
                    else { v = 7 + 2 * r.uint(165, 1); v += r.uint(145, 1); }
                } else {
                    const int b1 = r.uint(p[8], 1);
                    const int b0 = r.uint(p[9 + b1], 1);
                    const int cat = 2 * b1 + b0;
                    v = 0;
                    for (const uint8_t* t = kCat[cat]; *t; ++t) v = 2 * v + r.uint(*t, 1);
                    v += 3 + (8 << cat);
                }
                p = prob[kBand[n]][2];
            }
            const int z = kZig[n - 1];
            const int32_t c = v * (z > 0 ? q1 : q0);
            const int32_t sign = -int32_t(r.bit(kUniform));       // 0 or -1: the sign is a coin flip, no branch on it
            const int16_t w = int16_t(uint16_t(uint32_t((c ^ sign) - sign)));          // (c << 16) >> 16
            blk[z] = w;
            any |= w;
            if (n == 16 || !r.bit(p[0])) break;
        }
        reader = r;
        *stored = any;
        return 1;
    }

    // stores a block that holds something and hands the scratch block back zeroed
    void push_block(int16_t* b) { out_.coeff.insert(out_.coeff.end(), b, b + 16); std::memset(b, 0, 16 * sizeof(int16_t)); }

    // inverseWHT16 (Vp8Decoder.hx:960-989) restricted to the question the host needs answered:
    // which of the 16 luma DCs are non-zero (feeds `skip`, hence FilterParam.inner).  int32, untruncated (Q5).
    static uint32_t wht_nonzero_mask(const int16_t* c) {
        int32_t m[16];
        for (int i = 0; i < 4; ++i) {
            const int32_t a0 = c[i] + c[12 + i], a1 = c[4 + i] + c[8 + i], a2 = c[4 + i] - c[8 + i], a3 = c[i] - c[12 + i];
            m[i] = a0 + a1; m[8 + i] = a0 - a1; m[4 + i] = a3 + a2; m[12 + i] = a3 - a2;
        }
        uint32_t mask = 0;
        for (int i = 0; i < 4; ++i) {
            const int32_t dc = m[4 * i] + 3;
            const int32_t a0 = dc + m[4 * i + 3], a1 = m[4 * i + 1] + m[4 * i + 2], a2 = m[4 * i + 1] - m[4 * i + 2], a3 = dc - m[4 * i + 3];
            if ((a0 + a1) >> 3) mask |= 1u << (4 * i + 0);
            if ((a3 + a2) >> 3) mask |= 1u << (4 * i + 1);
            if ((a0 - a1) >> 3) mask |= 1u << (4 * i + 2);
            if ((a3 - a2) >> 3) mask |= 1u << (4 * i + 3);
        }
        return mask;
    }

    // reconstruct()'s parse half, Vp8Decoder.hx:772-808 (+ parsePredMode* :838-903, parseResiduals :639-714).
    // Returns the reference's `skip` value.
    bool macroblock(int mbx, int mby, uint32_t* rec, bool* is_i16) {
        if (update_map_)
            segment_ = !first_.bit(seg_prob_[0]) ? first_.uint(seg_prob_[1], 1) : first_.uint(seg_prob_[2], 1) + 2;
        bool skip = use_skip_prob_ ? first_.bit(skip_prob_) : false;
        NzCtx& up = up_[size_t(mbx)];
        const bool i16 = first_.bit(145);
        *is_i16 = i16;
        uint32_t w0 = 0, w1 = 0, w2 = 0;
        if (i16) {
            int p;
            if (!first_.bit(156)) p = !first_.bit(163) ? kDC : kVE;
            else p = !first_.bit(128) ? kHE : kTM;
            for (int i = 0; i < 4; ++i) { up.pred[i] = uint8_t(p); left_.pred[i] = uint8_t(p); }
            w0 = uint32_t(p);
            w2 |= 1u << WP_MB_I16_BIT;
        } else {
            for (int j = 0; j < 4; ++j) {
                int p = left_.pred[j];
                for (int i = 0; i < 4; ++i) {
                    const uint8_t* pr = &wpd_PredProb[(up.pred[i] * 10 + p) * 9];
                    if (!first_.bit(pr[0])) p = kDC;
                    else if (!first_.bit(pr[1])) p = kTM;
                    else if (!first_.bit(pr[2])) p = kVE;
                    else if (!first_.bit(pr[3])) p = !first_.bit(pr[4]) ? kHE : (!first_.bit(pr[5]) ? kRD : kVR);
                    else if (!first_.bit(pr[6])) p = kLD;
                    else if (!first_.bit(pr[7])) p = kVL;
                    else if (!first_.bit(pr[8])) p = kHD;
                    else p = kHU;
                    const int n = 4 * j + i;
                    if (n < 8) w0 |= uint32_t(p) << (4 * n); else w1 |= uint32_t(p) << (4 * (n - 8));
                    up.pred[i] = uint8_t(p);
                }
                left_.pred[j] = uint8_t(p);
            }
        }
        int c8;
        if (!first_.bit(142)) c8 = kDC;
        else if (!first_.bit(114)) c8 = kVE;
        else if (!first_.bit(183)) c8 = kHE;
        else c8 = kTM;
        w2 |= uint32_t(c8) << WP_MB_C8_SHIFT;

        const size_t first_block = out_.coeff.size() / 16;
        if (!skip) {
            BoolReader& r = part_[mby & (n_parts_ - 1)];
            const Dequant& dq = dq_[segment_];
            int plane = kPlaneY1SansY2;
            uint32_t dc_nz = 0, ac_nz = 0;
            int16_t blk[16] = {0};            // all zero whenever tokens() is entered
            int stored;
            if (i16) {
                const int nz = tokens(r, kPlaneY2, left_.nz_y16 + up.nz_y16, dq.y2, false, blk, &stored);
                left_.nz_y16 = up.nz_y16 = uint8_t(nz);
                if (stored) { dc_nz = wht_nonzero_mask(blk); push_block(blk); w2 |= 1u << WP_MB_WHT_BIT; }
                plane = kPlaneY1WithY2;
            }
            int lnz[4], unz[4];
            for (int i = 0; i < 4; ++i) { lnz[i] = (left_.nz_mask >> i) & 1; unz[i] = (up.nz_mask >> i) & 1; }
            for (int y = 0; y < 4; ++y) {
                int nz = lnz[y];
                for (int x = 0; x < 4; ++x) {
                    nz = tokens(r, plane, nz + unz[x], dq.y1, i16, blk, &stored);
                    unz[x] = nz;
                    const int n = 4 * y + x;
                    if (nz) ac_nz |= 1u << n;
                    if (!i16 && blk[0] != 0) dc_nz |= 1u << n;
                    if (stored) { push_block(blk); w2 |= 1u << n; }
                }
                lnz[y] = nz;
            }
            uint32_t lmask = uint32_t(lnz[0] | lnz[1] << 1 | lnz[2] << 2 | lnz[3] << 3);
            uint32_t umask = uint32_t(unz[0] | unz[1] << 1 | unz[2] << 2 | unz[3] << 3);
            for (int i = 0; i < 4; ++i) { lnz[i] = (left_.nz_mask >> (4 + i)) & 1; unz[i] = (up.nz_mask >> (4 + i)) & 1; }
            for (int c = 0; c < 4; c += 2) {
                for (int y = 0; y < 2; ++y) {
                    int nz = lnz[y + c];
                    for (int x = 0; x < 2; ++x) {
                        nz = tokens(r, kPlaneUV, nz + unz[x + c], dq.uv, false, blk, &stored);
                        unz[x + c] = nz;
                        const int n = 16 + c * 2 + y * 2 + x;
                        if (nz) ac_nz |= 1u << n;
                        if (blk[0] != 0) dc_nz |= 1u << n;
                        if (stored) { push_block(blk); w2 |= 1u << n; }
                    }
                    lnz[y + c] = nz;
                }
            }
            lmask |= uint32_t(lnz[0] | lnz[1] << 1 | lnz[2] << 2 | lnz[3] << 3) << 4;
            umask |= uint32_t(unz[0] | unz[1] << 1 | unz[2] << 2 | unz[3] << 3) << 4;
            left_.nz_mask = uint8_t(lmask);
            up.nz_mask = uint8_t(umask);
            skip = dc_nz == 0 && ac_nz == 0;
        } else {
            if (i16) { left_.nz_y16 = 0; up.nz_y16 = 0; }
            left_.nz_mask = 0;
            up.nz_mask = 0;
        }
        rec[0] = w0; rec[1] = w1; rec[2] = w2; rec[3] = uint32_t(first_block);
        return skip;
    }

    const uint8_t* data_; size_t len_; size_t pos_ = 0;
    Vp8Packed& out_;
    BoolReader first_, part_[8];
    int n_parts_ = 1;
    bool use_segment_ = false, update_map_ = false, relative_delta_ = false;
    uint8_t seg_quant_[4] = {0, 0, 0, 0}, seg_strength_[4] = {0, 0, 0, 0}, seg_prob_[3] = {255, 255, 255};
    bool filter_simple_ = false, use_lf_delta_ = false;
    int filter_level_ = 0, sharpness_ = 0;
    uint8_t ref_delta_[4] = {0, 0, 0, 0}, mode_delta_[4] = {0, 0, 0, 0};
    Dequant dq_[4];
    uint8_t prob_[4][8][3][11];
    bool use_skip_prob_ = false; int skip_prob_ = 0;
    LoopParam loop_[4][2];
    int segment_ = 0;
    NzCtx left_; std::vector<NzCtx> up_;
};

}  // namespace

int parse_vp8(const uint8_t* data, size_t len, Vp8Packed& out, std::string& err) {
    Vp8Parser p(data, len, out);
    return p.run(err);
}

}  // namespace wp
