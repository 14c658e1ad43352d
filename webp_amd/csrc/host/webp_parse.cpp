// RIFF container walk: mirrors WebPDecoder.decode / decodeChunks / readRiffChunk / readVp8x /
// readAlph / readAnmf / readFrameData (WebPDecoder.hx:37-309) and the host half of readAlpha
// (:311-355).  It stops where the reference starts touching pixels: every VP8 / VP8L / ALPH payload
// is turned into the packed buffers that the HIP kernels consume.
#include "host.h"
#include <cstring>

namespace wp {
namespace {

inline uint32_t le32(const uint8_t* p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; }
inline int le24(const uint8_t* p) { return int(p[0]) | int(p[1]) << 8 | int(p[2]) << 16; }
inline bool tag(const uint8_t* p, const char* t) { return std::memcmp(p, t, 4) == 0; }

// readAlpha's host half (WebPDecoder.hx:311-355): raw bytes, or the VP8L entropy decode of the
// alpha image behind a synthesised 5-byte header (:330-342).  Green extraction + unfilterAlpha
// run on the device.
int parse_alpha(const uint8_t* d, size_t n, int wm1, int hm1, int flags, AlphaPacked& out, std::string& err) {
    const int compression = flags & 3;
    out.width = wm1 + 1;
    out.height = hm1 + 1;
    out.filter = (flags >> 2) & 3;
    if (compression == 0) {
        const size_t need = size_t(out.width) * size_t(out.height);
        if (n < need) { err = "Failed to read alpha values: Eof"; return WP_ERR_FORMAT; }
        out.raw.assign(d, d + need);
        out.vp8l = false;
    } else if (compression == 1) {
        if (wm1 > 0x3fff || hm1 > 0x3fff) { err = "Invalid format: dimensions too large"; return WP_ERR_FORMAT; }
        const uint8_t hdr[5] = {0x2f, uint8_t(wm1 & 0xff), uint8_t((wm1 >> 8) | ((hm1 & 0x3f) << 6)),
                                uint8_t((hm1 >> 2) & 0xff), uint8_t((hm1 >> 10) & 0xff)};
        const int rc = parse_vp8l(hdr, 5, d, n, out.coded, err);
        if (rc) return rc;
        out.vp8l = true;
    } else {
        err = "Invalid format: unsupported compression type";
        return WP_ERR_FORMAT;
    }
    out.present = true;
    return WP_OK;
}

// readFrameData, WebPDecoder.hx:258-309
int parse_frame_data(const uint8_t* d, size_t len, size_t pos, int wm1, int hm1, ParsedFrame& fr, std::string& err, int flags) {
    bool lossy = false, lossless = false;
    while (pos < len) {
        if (len - pos < 8) break;
        const uint8_t* id = d + pos;
        const int32_t n = int32_t(le32(d + pos + 4));
        pos += 8;
        if (n < 0 || size_t(n) > len - pos) { err = "Invalid ANMF subchunk length"; return WP_ERR_FORMAT; }
        const uint8_t* cd = d + pos;
        pos += size_t(n);
        if ((n & 1) == 1 && pos < len) ++pos;
        if (tag(id, "ALPH")) {
            fr.alpha = AlphaPacked();
            if (n >= 1) {
                std::string ignored;        // failures fall back to an opaque frame (:283-290)
                AlphaPacked a;
                if (parse_alpha(cd + 1, size_t(n) - 1, wm1, hm1, cd[0], a, ignored) == WP_OK) fr.alpha = std::move(a);
            }
        } else if (tag(id, "VP8 ")) {
            fr.vp8 = Vp8Packed();
            const int rc = parse_vp8(cd, size_t(n), fr.vp8, err);
            if (rc) return rc;
            lossy = true;
        } else if (tag(id, "VP8L")) {
            fr.vp8l = Vp8lPacked();
            const int rc = parse_vp8l(cd, size_t(n), nullptr, 0, fr.vp8l, err, (flags & WP_PARSE_VP8L_TOKENS) != 0);
            if (rc) return rc;
            lossless = true;
        }
    }
    if (lossless) { fr.type = 1; fr.alpha = AlphaPacked(); return WP_OK; }
    if (lossy) { fr.type = 0; return WP_OK; }
    err = "Invalid format: no VP8/VP8L chunk found in ANMF";
    return WP_ERR_FORMAT;
}

}  // namespace

int parse_webp(const uint8_t* data, size_t len, wp_parsed& out, std::string& err, int flags) {
    // readRiffHeader :37-45
    if (len < 4) { err = "Eof"; return WP_ERR_EOF; }
    if (!tag(data, "RIFF")) { err = "Missing RIFF chunk header"; return WP_ERR_FORMAT; }
    if (len < 12) { err = "Eof"; return WP_ERR_EOF; }
    const int32_t riff_len = int32_t(le32(data + 4));
    if (riff_len < 4) { err = "Short chunk data"; return WP_ERR_FORMAT; }
    int64_t remaining = int64_t(riff_len) - 4;
    size_t pos = 12;
    int wm1 = 0, hm1 = 0;
    bool lossy = false, lossless = false;
    ParsedFrame still;
    std::vector<ParsedFrame> frames;
    // decodeChunks (:246-256) collects every chunk before any payload is decoded: a container error anywhere in the file
    // comes before a payload's own error.  The same walk, without the payloads, first.
    {
        size_t p2 = pos;
        int64_t r2 = remaining;
        while (r2 > 0) {
            if (len - p2 < 8) { err = "Eof"; return WP_ERR_EOF; }
            const int32_t n = int32_t(le32(data + p2 + 4));
            p2 += 8;
            if (n < 0 || size_t(n) > len - p2) { err = "Eof"; return WP_ERR_EOF; }
            p2 += size_t(n);
            if (n & 1) {
                if (p2 >= len) { err = "Eof"; return WP_ERR_EOF; }
                ++p2;
                r2 -= 1;
            }
            r2 -= int64_t(n) + 8;
        }
    }
    while (remaining > 0) {                                     // readRiffChunk :47-76
        if (len - pos < 8) { err = "Eof"; return WP_ERR_EOF; }
        const uint8_t* id = data + pos;
        const int32_t n = int32_t(le32(data + pos + 4));
        pos += 8;
        if (n < 0 || size_t(n) > len - pos) { err = "Eof"; return WP_ERR_EOF; }
        const uint8_t* cd = data + pos;
        pos += size_t(n);
        if (n & 1) {
            if (pos >= len) { err = "Eof"; return WP_ERR_EOF; }
            ++pos;
            remaining -= 1;
        }
        remaining -= int64_t(n) + 8;
        if (tag(id, "VP8X")) {                                  // readVp8x :133-141
            if (n < 10) { err = "Eof"; return WP_ERR_EOF; }
            wm1 = le24(cd + 4);
            hm1 = le24(cd + 7);
        } else if (tag(id, "ALPH")) {                           // readAlph :78-88
            if (n < 1) { err = "Eof"; return WP_ERR_EOF; }
            AlphaPacked a;
            const int rc = parse_alpha(cd + 1, size_t(n) - 1, wm1, hm1, cd[0], a, err);
            if (rc) return rc;
            still.alpha = std::move(a);
        } else if (tag(id, "VP8 ")) {                           // readVp8 :90-103
            still.vp8 = Vp8Packed();
            const int rc = parse_vp8(cd, size_t(n), still.vp8, err);
            if (rc) return rc;
            lossy = true;
        } else if (tag(id, "VP8L")) {                           // readVp8l :105-131
            still.vp8l = Vp8lPacked();
            const int rc = parse_vp8l(cd, size_t(n), nullptr, 0, still.vp8l, err, (flags & WP_PARSE_VP8L_TOKENS) != 0);
            if (rc) return rc;
            lossless = true;
        } else if (tag(id, "ANIM")) {                           // readAnim :143-148
            if (n < 6) { err = "Eof"; return WP_ERR_EOF; }
        } else if (tag(id, "ANMF")) {                           // readAnmf :150-177
            if (n < 16) { err = "Eof"; return WP_ERR_EOF; }
            ParsedFrame fr;
            const int fwm1 = le24(cd + 6), fhm1 = le24(cd + 9);
            const int rc = parse_frame_data(cd, size_t(n), 16, fwm1, fhm1, fr, err, flags);
            if (rc) return rc;
            fr.x = le24(cd); fr.y = le24(cd + 3);
            fr.width = fwm1 + 1; fr.height = fhm1 + 1;
            fr.duration = le24(cd + 12);
            fr.dispose = (cd[15] & 1) != 0;
            fr.blend = (cd[15] & 2) == 0;
            frames.push_back(std::move(fr));
        }
    }
    if (!frames.empty()) {                                      // :203-217
        out.kind = 2; out.width = wm1 + 1; out.height = hm1 + 1;
        out.frames = std::move(frames);
        return WP_OK;
    }
    if (lossless) {                                             // :219-234
        still.type = 1; still.alpha = AlphaPacked();
        still.width = still.vp8l.width; still.height = still.vp8l.height;
        out.kind = 1; out.width = still.width; out.height = still.height;
        out.frames.clear(); out.frames.push_back(std::move(still));
        return WP_OK;
    }
    if (lossy) {                                                // :236-241
        still.type = 0;
        still.width = still.vp8.width; still.height = still.vp8.height;
        out.kind = 0; out.width = still.width; out.height = still.height;
        out.frames.clear(); out.frames.push_back(std::move(still));
        return WP_OK;
    }
    err = "Invalid format: no VP8/VP8L chunk found";
    return WP_ERR_FORMAT;
}

}  // namespace wp
