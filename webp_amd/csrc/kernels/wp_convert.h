// Per-pixel YUV -> ARGB arithmetic shared by the streaming conversion kernels (Tools.hx:25-31).
//
// The reference converts in double precision, full range:  R = Y + 1.402 Cr,  G = Y - 0.344136 Cb - 0.714136 Cr,
// B = Y + 1.772 Cb  (Cb, Cr centred on 128), then Math.round(clamp(x, 0, 255)) with Math.round = floor(x + 0.5)
// (SURVEY Q7).  Every chroma term is an integer divided by 1000 (R, B) or 10^6 (G), so x is within ~1e-13 of a
// rational whose distance to the next rounding boundary is either 0 (an exact .5 tie) or >= 1e-6: only at exact
// ties can fp64 rounding error change the result.  Enumerating them:
//   R: 1402 Cr = 500 (mod 1000) has no solution in -128..127.
//   B: Cb = +-125; fp64 rounds both like round-half-up.
//   G: (Cb, Cr) = (-50, 50) and (50, -50); fp64 rounds like round-half-up except (-50, 50) with 111 <= Y <= 146,
//      where (Y - a) - b lands one ulp below the tie (result Y - 19 instead of Y - 18).
// Hence the integer form below is bit-identical to the reference's fp64 arithmetic for all 2^24 (Y, Cb, Cr)
// triples; the GPU test checks every one of them against the fp64 oracle.  round(clamp(x)) = clamp(round(x)).
#pragma once
#include "wp_device.h"

namespace wpk {

// rounded chroma contributions of one (Cb, Cr) sample, shared by the 2x2 luma samples it covers
struct ChromaTerms { int r, g, b; bool tie; };
__device__ __forceinline__ ChromaTerms chroma_terms(int cbv, int crv) {
    // Each term is floor(rational + 1/2) of an integer in -128..127 (two of them for G).  The multiply-and-shift forms below
    // give exactly those floors for every argument (tools/convert_constants.py searches the constants and checks all 256 /
    // 65536 arguments against the exact rational); they need no division (v_mul_hi_u32 runs at a quarter of the rate of the
    // 24-bit multiplies used here, and the finishing pass is bound by instruction issue when it overlaps reconstruction).
    const int cb = cbv - 128, cr = crv - 128;
    ChromaTerms t;
    t.r = (__mul24(5743, cr) + 2048) >> 12;                                  // floor(1.402 cr + 0.5)
    t.b = (__mul24(3629, cb) + 1031) >> 11;                                  // floor(1.772 cb + 0.5)
    t.g = (1048616 - __mul24(721705, cb) - __mul24(1497652, cr)) >> 21;      // floor(0.5 - 0.344136 cb - 0.714136 cr)
    t.tie = cb == -50 && cr == 50;
    return t;
}

// one pixel -> bytes A,R,G,B in memory order (little-endian dword: A | R<<8 | G<<16 | B<<24).  FIX = apply the single
// Y-dependent tie correction; callers take the FIX = false path when no lane of the wavefront holds the tie pair.
template <bool FIX>
__device__ __forceinline__ uint32_t convert_px(int yv, const ChromaTerms& t, uint32_t a) {
    const int fix = (FIX && t.tie && yv >= 111 && yv <= 146) ? 1 : 0;
    return a | uint32_t(clampi(yv + t.r, 0, 255)) << 8 | uint32_t(clampi(yv + t.g - fix, 0, 255)) << 16 | uint32_t(clampi(yv + t.b, 0, 255)) << 24;
}

// four horizontally adjacent pixels: luma dword, the chroma terms of its two chroma samples, four alpha bytes
template <bool FIX>
__device__ __forceinline__ uint4 convert_row4(uint32_t y4, const ChromaTerms& t0, const ChromaTerms& t1, uint32_t a4) {
    return make_uint4(convert_px<FIX>(int(y4 & 0xff), t0, a4 & 0xff), convert_px<FIX>(int((y4 >> 8) & 0xff), t0, (a4 >> 8) & 0xff),
                      convert_px<FIX>(int((y4 >> 16) & 0xff), t1, (a4 >> 16) & 0xff), convert_px<FIX>(int(y4 >> 24), t1, a4 >> 24));
}
// The same without the tie correction, in packed 16-bit arithmetic: pixels (0, 2) and (1, 3) of the row share the term
// pair (t0, t1), so each channel is two packed adds and a saturating pack (clip_pack: 0..255, bytes in pixel order),
// and a 4x4 byte transpose turns the four channel dwords into four A,R,G,B pixels: 26 instructions for four pixels
// where the scalar form above takes about 44.
__device__ __forceinline__ uint4 convert_row4_packed(uint32_t y4, const ChromaTerms& t0, const ChromaTerms& t1, uint32_t a4) {
    const short2_t y02 = as_s2(y4 & 0x00ff00ffu), y13 = as_s2((y4 >> 8) & 0x00ff00ffu);
    const short2_t tr = as_s2(__builtin_amdgcn_perm(uint32_t(t1.r), uint32_t(t0.r), 0x05040100u));      // (t0, t1) as an int16 pair
    const short2_t tg = as_s2(__builtin_amdgcn_perm(uint32_t(t1.g), uint32_t(t0.g), 0x05040100u));
    const short2_t tb = as_s2(__builtin_amdgcn_perm(uint32_t(t1.b), uint32_t(t0.b), 0x05040100u));
    const uint32_t r4 = clip_pack(y02 + tr, y13 + tr), g4 = clip_pack(y02 + tg, y13 + tg), b4 = clip_pack(y02 + tb, y13 + tb);
    const uint32_t ar_lo = __builtin_amdgcn_perm(r4, a4, 0x05010400u), ar_hi = __builtin_amdgcn_perm(r4, a4, 0x07030602u);   // A0 R0 A1 R1 | A2 R2 A3 R3
    const uint32_t gb_lo = __builtin_amdgcn_perm(b4, g4, 0x05010400u), gb_hi = __builtin_amdgcn_perm(b4, g4, 0x07030602u);   // G0 B0 G1 B1 | G2 B2 G3 B3
    return make_uint4(__builtin_amdgcn_perm(gb_lo, ar_lo, 0x05040100u), __builtin_amdgcn_perm(gb_lo, ar_lo, 0x07060302u),
                      __builtin_amdgcn_perm(gb_hi, ar_hi, 0x05040100u), __builtin_amdgcn_perm(gb_hi, ar_hi, 0x07060302u));
}
// Opaque pixels, no tie: pixel-major.  A chroma sample's terms are kept as two int16 pairs, (255, r) and (g, b); a pixel is
// two packed adds of its luma value -- broadcast to both halves by the instruction's operand selects, from whichever half of
// y02 / y13 holds it -- and two saturating packs, the second writing the upper half of the result (255 + Y saturates to the
// opaque alpha byte): 8 four-cycle instructions per pixel and no byte transposes, 39 half-cost units per 4 pixels against 49.
struct ChromaPairs { uint32_t ar, gb; };
__device__ __forceinline__ ChromaPairs chroma_pairs(const ChromaTerms& t) {
    ChromaPairs p;
    p.ar = (uint32_t(t.r) << 16) | 255u;
    p.gb = __builtin_amdgcn_perm(uint32_t(t.b), uint32_t(t.g), 0x05040100u);
    return p;
}
template <int HALF>      // which half of `ypair` holds this pixel's luma value
__device__ __forceinline__ uint32_t convert_px_opaque(uint32_t ypair, const ChromaPairs& c) {
    uint32_t lo, hi;
    if (HALF == 0) {
        asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(lo) : "v"(ypair), "v"(c.ar));
        asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(hi) : "v"(ypair), "v"(c.gb));
    } else {
        asm("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(lo) : "v"(ypair), "v"(c.ar));
        asm("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(hi) : "v"(ypair), "v"(c.gb));
    }
    return clip_pack_seq(as_s2(lo), as_s2(hi));
}
// The tie correction applied afterwards (taken when some lane of the wavefront holds the tie pair): one off the green byte of a
// pixel whose chroma sample is the tie pair and whose luma value lies in 111 .. 146.  Green is Y - 18 there (93 .. 128 before
// the correction), so the byte can simply be decremented: no clamp is in reach.
__device__ __forceinline__ void tie_patch_row4(uint4& o, uint32_t y4, bool tie0, bool tie1) {
    o.x -= (tie0 && (y4 & 0xffu) - 111u <= 35u) ? 0x00010000u : 0u;
    o.y -= (tie0 && ((y4 >> 8) & 0xffu) - 111u <= 35u) ? 0x00010000u : 0u;
    o.z -= (tie1 && ((y4 >> 16) & 0xffu) - 111u <= 35u) ? 0x00010000u : 0u;
    o.w -= (tie1 && (y4 >> 24) - 111u <= 35u) ? 0x00010000u : 0u;
}
// Frames with an alpha plane: the four alpha bytes of the row replace the opaque ones (one v_bfi_b32 per pixel)
__device__ __forceinline__ void alpha_patch_row4(uint4& o, uint32_t a4) {
    o.x = (o.x & 0xffffff00u) | (a4 & 0xffu);
    o.y = (o.y & 0xffffff00u) | ((a4 >> 8) & 0xffu);
    o.z = (o.z & 0xffffff00u) | ((a4 >> 16) & 0xffu);
    o.w = (o.w & 0xffffff00u) | (a4 >> 24);
}
__device__ __forceinline__ uint4 convert_row4_opaque(uint32_t y4, const ChromaPairs& c0, const ChromaPairs& c1) {
    const uint32_t y02 = y4 & 0x00ff00ffu, y13 = (y4 >> 8) & 0x00ff00ffu;
    return make_uint4(convert_px_opaque<0>(y02, c0), convert_px_opaque<0>(y13, c0), convert_px_opaque<1>(y02, c1), convert_px_opaque<1>(y13, c1));
}
__device__ __forceinline__ uint4 convert_row4(uint32_t y4, const ChromaTerms& t0, const ChromaTerms& t1, uint32_t a4, bool any_tie) {
    return any_tie ? convert_row4<true>(y4, t0, t1, a4) : convert_row4_packed(y4, t0, t1, a4);
}

// luma dword + two chroma samples each (u2, v2: two bytes) -> 16 bytes of A,R,G,B
__device__ __forceinline__ uint4 convert_px4(uint32_t y4, uint32_t u2, uint32_t v2, uint32_t a4) {
    const ChromaTerms t0 = chroma_terms(int(u2 & 0xff), int(v2 & 0xff));
    const ChromaTerms t1 = chroma_terms(int((u2 >> 8) & 0xff), int((v2 >> 8) & 0xff));
    const bool any_tie = __builtin_amdgcn_ballot_w64(t0.tie || t1.tie) != 0;      // wavefront-uniform
    return convert_row4(y4, t0, t1, a4, any_tie);
}

}  // namespace wpk
