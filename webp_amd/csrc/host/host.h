// Internal C++ declarations of the host side (container walk + entropy decode).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "../../../include/webp_amd.h"

namespace wp {

// Result of the VP8 parse half: webp.Image.FrameHeader fields + the packed device buffers.
struct Vp8Packed {
    int keyFrame = 0, versionNumber = 0, showFrame = 0, firstPartitionLen = 0;
    int width = 0, height = 0, xScale = 0, yScale = 0;
    int mbw = 0, mbh = 0;
    int filter_type = 0;                 // 0 none, 1 simple, 2 normal
    int n_i4 = 0;                        // macroblocks predicted per 4x4 block (they cost the device ten dependent mini-steps)
    std::vector<uint32_t> mb;            // WP_MB_WORDS per macroblock
    std::vector<uint32_t> filt;          // 1 per macroblock
    std::vector<int16_t> coeff;          // 16 per stored block
};
int parse_vp8(const uint8_t* data, size_t len, Vp8Packed& out, std::string& err);

// Result of the VP8L parse half (Vp8LDecoder.decode up to and including decodePix).
struct Vp8lTransform { int type = 0, bits = 0, old_width = 0; std::vector<uint8_t> pix; };
struct Vp8lPacked {
    int width = 0, height = 0;           // originalW, h
    int coded_width = 0;                 // width of `pix` rows
    std::vector<Vp8lTransform> transforms;   // bitstream order
    std::vector<uint8_t> pix;            // RGBA, 4*coded_width*height (empty when the image was tokenized instead)
    // WP_PARSE_VP8L_TOKENS: the entropy-decoded symbols of the top-level image, NOT resolved into pixels (the LZ77 copies
    // and colour-cache lookups of decodePix, Vp8LDecoder.hx:254-333, are left to wp_vp8l_resolve_dev); format in webp_amd.h
    bool tokenized = false;
    int cc_bits = 0;                     // colour-cache bits of the top-level image (0 = none)
    std::vector<uint32_t> tokens;
};
// Input may be the concatenation of two buffers (types/ConcatInput.hx, used by readAlpha).
int parse_vp8l(const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, Vp8lPacked& out, std::string& err, bool tokens = false);

// ALPH chunk payload after the host half of readAlpha (WebPDecoder.hx:311-355).
struct AlphaPacked {
    bool present = false;
    int width = 0, height = 0, filter = 0;
    bool vp8l = false;
    std::vector<uint8_t> raw;            // compression 0: width*height bytes
    Vp8lPacked coded;                    // compression 1
};

struct ParsedFrame {
    int type = 0;                        // 0 lossy (Yuv420), 1 lossless (Argb)
    int width = 0, height = 0;           // size used for ARGB output (frame header for ANMF, image header otherwise)
    int x = 0, y = 0, duration = 0; bool blend = false, dispose = false;   // AnimFrameHeader
    Vp8Packed vp8;
    Vp8lPacked vp8l;
    AlphaPacked alpha;
};

}  // namespace wp

// The opaque object behind the C ABI's wp_parsed.
struct wp_parsed {
    int kind = 0;                        // 0 Yuv420, 1 Argb, 2 Anim
    int width = 0, height = 0;           // Image.header.width/height
    std::vector<wp::ParsedFrame> frames;
};

namespace wp {
int parse_webp(const uint8_t* data, size_t len, wp_parsed& out, std::string& err, int flags = 0);
void set_error(const std::string& msg);
}
