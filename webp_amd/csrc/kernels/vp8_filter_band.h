// Loop filter of one band (4 macroblock rows) of one frame by one wavefront: device code shared by the stand-alone loop
// filter kernel (vp8_filter.hip) and the fused lossy kernel (vp8_lossy.hip).  See vp8_filter.hip for the design.
#pragma once
#include "wp_device.h"

namespace wpk {

struct alignas(16) FilterTile {
    uint8_t y[24][48];          // rows 0..7: the 8 rows above, 8..23: MB rows; cols: MB slot (x & 1) * 16; pitch 48 = 12 dwords:
                                // the 16-byte row accesses (lane = row) of 8 consecutive lanes cover all 32 banks once
    uint8_t c[2][12][16];       // rows 0..3: above, 4..11: MB rows; cols: slot (x & 1) * 8
    uint8_t pad[16];            // 1552 B = 388 dwords: the four groups' tiles start 4 banks apart
};
struct alignas(16) FilterWaveLds {
    FilterTile tile[kBandRows];
    alignas(16) uint8_t ring_y[kBandRows][8][64];      // ring_y[g][row][abs_col & 63]: 8 bottom luma rows of the MB row above group g
    uint8_t ring_c[kBandRows][2][4][32];   // same for Cb / Cr (4 rows)
};

__device__ __forceinline__ uint2& t64(uint8_t* p) { return *reinterpret_cast<uint2*>(p); }
__device__ __forceinline__ uint4& t128(uint8_t* p) { return *reinterpret_cast<uint4*>(p); }
__device__ __forceinline__ void unpack4(uint32_t w, int* v) { v[0] = w & 0xff; v[1] = (w >> 8) & 0xff; v[2] = (w >> 16) & 0xff; v[3] = w >> 24; }
__device__ __forceinline__ uint32_t pack4i(const int* v) { return uint32_t(v[0]) | uint32_t(v[1]) << 8 | uint32_t(v[2]) << 16 | uint32_t(v[3]) << 24; }
__device__ __forceinline__ uint32_t& t32(uint8_t* p) { return *reinterpret_cast<uint32_t*>(p); }
// |a - b| for non-negative a, b in ONE instruction (v_sad_u32 with a zero accumulator) instead of sub, sub, max:
// both kernels are bound by VALU issue, and most integer VALU instructions cost 4 cycles per wavefront
// (tools/ubench/valu_rate.hip), so instruction count is what matters.
__device__ __forceinline__ int absdiff(int a, int b) {
    int d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ int clamp127(int v) { return clampi(v, -127, 127); }   // Filter.hx:16-18 (symmetric)
__device__ __forceinline__ int clamp15(int v) { return clampi(v, -15, 15); }      // Filter.hx:20-22
__device__ __forceinline__ int clamp255(int v) { return clampi(v, 0, 255); }

// The edge filters are written branch-free (every lane of the wavefront handles a different pixel
// position, so all branches would be executed anyway) and take / return values, never references:
// the sample arrays then stay in registers.  "Not filtered" is expressed as zero deltas, which
// leaves the 0..255 samples unchanged through clamp255.

// "Not filtered" is folded into the filter value itself: with a = 0 every delta below comes out 0 ((0 + 4) >> 3,
// (0 + 3) >> 3, (0 + 1) >> 1 and (k * 0 + 63) >> 7 are all 0), so one select replaces one per delta.  absdiff() is
// always called with its arguments in position order, so that the difference an edge shares with the next edge of the
// line (its |q2 - q3| is the next one's |p1 - p0|) is computed once.

// filter2, one position (Filter.hx:29-50)
__device__ __forceinline__ void simple_edge(int p1, int& p0, int& q0, int q1, int level, bool use) {
    const bool on = use & ((absdiff(p0, q0) << 1) + (absdiff(p1, q1) >> 1) <= level);
    const int a = on ? __mul24(q0 - p0, 3) + clamp127(p1 - q1) : 0;
    const int a1 = clamp15((a + 4) >> 3);
    const int a2 = clamp15((a + 3) >> 3);
    p0 = clamp255(p0 + a2);
    q0 = clamp255(q0 - a1);
}

// filter246, one position (Filter.hx:53-111).  FOUR = inner edge (4-pixel filter), else MB edge (6-pixel).
template <bool FOUR>
__device__ __forceinline__ void normal_edge(int p3, int& p2, int& p1, int& p0, int& q0, int& q1, int& q2, int q3,
                                            int level, int ilevel, int hlevel, bool use) {
    const int m10 = max(absdiff(p1, p0), absdiff(q0, q1));
    const int worst = max(max(max(absdiff(p3, p2), absdiff(p2, p1)), m10), max(absdiff(q1, q2), absdiff(q2, q3)));
    const bool on = use & ((absdiff(p0, q0) << 1) + (absdiff(p1, q1) >> 1) <= level) & (worst <= ilevel);      // & not &&: no branch around the mask terms
    const bool hev = m10 > hlevel;
    const int base = __mul24(q0 - p0, 3);
    const int outer = clamp127(p1 - q1);
    if (FOUR) {
        const int a = on ? base + (hev ? outer : 0) : 0;
        const int a1 = clamp15((a + 4) >> 3);
        const int a2 = clamp15((a + 3) >> 3);
        const int a3 = hev ? 0 : (a1 + 1) >> 1;
        p1 = clamp255(p1 + a3);
        p0 = clamp255(p0 + a2);
        q0 = clamp255(q0 - a1);
        q1 = clamp255(q1 - a3);
    } else {
        const int w = on ? base + outer : 0;
        const int a = clamp127(w);
        const int c1 = (__mul24(a, 27) + 63) >> 7, c2 = (__mul24(a, 18) + 63) >> 7, c3 = (__mul24(a, 9) + 63) >> 7;
        const int dp0 = hev ? clamp15((w + 3) >> 3) : c1;
        const int dq0 = hev ? clamp15((w + 4) >> 3) : c1;
        const int d1 = hev ? 0 : c2;
        const int d2 = hev ? 0 : c3;
        p2 = clamp255(p2 + d2);
        p1 = clamp255(p1 + d1);
        p0 = clamp255(p0 + dp0);
        q0 = clamp255(q0 - dq0);
        q1 = clamp255(q1 - d1);
        q2 = clamp255(q2 - d2);
    }
}

// One line of N = 4 + 16 (luma) or 4 + 8 (chroma) samples crossing the MB edge at index 4 and the
// inner edges at 8 (and 12, 16 for luma).  Order as in the reference: MB edge first, then inner edges.
// Edges are written out with constant indices so that the sample array stays in registers.  "This lane's macroblock
// has no MB edge / no inner edges" is one more term of the edge's filter mask, not a branch: a branch whose condition
// differs between the lanes of a wavefront is executed by the wavefront anyway, and merging its results with the
// untouched samples of the other lanes cost ~30 register moves per line.  SIMPLE is a property of the frame.
template <int N, bool SIMPLE>
__device__ __forceinline__ void filter_line(int (&v)[N], bool mb_edge, bool inner, int level, int ilevel, int hlevel) {
    if constexpr (SIMPLE) {
        simple_edge(v[2], v[3], v[4], v[5], level + 4, mb_edge);
        simple_edge(v[6], v[7], v[8], v[9], level, inner);
        if constexpr (N == 20) {
            simple_edge(v[10], v[11], v[12], v[13], level, inner);
            simple_edge(v[14], v[15], v[16], v[17], level, inner);
        }
    } else {
        normal_edge<false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], level + 4, ilevel, hlevel, mb_edge);
        normal_edge<true>(v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], level, ilevel, hlevel, inner);
        if constexpr (N == 20) {
            normal_edge<true>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], level, ilevel, hlevel, inner);
            normal_edge<true>(v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19], level, ilevel, hlevel, inner);
        }
    }
}

// ---- the normal filter with its mask and filter value computed in fp32 (same integers, fewer issue cycles) ----
// A sample x travels as the bit pattern of the float 2^23 + x (0x4B000000 | x).  Differences of such floats are exact
// integers; |.| is a free source modifier, so every |a - b| is ONE 2-cycle subtraction (v_sad_u32 takes 4), the two
// maxima over five differences are two v_max3_f32, and 2|p0 - q0| + (|p1 - q1| >> 1) <= level becomes one fused
// multiply-add and a comparison against 2 * level + 1 (4A + B <= 2L + 1: the left side of the original is the even number
// 4A + 2 floor(B / 2), which is <= 2L exactly when 4A + B <= 2L + 1).  The filter value 3 (q0 - p0) + clamp127(p1 - q1)
// is one v_med3_f32 and one fma; adding 1.5 * 2^23 turns it into an integer in the mantissa, and from there the deltas are
// the integer code of normal_edge, applied to the samples' bit patterns with integer adds and clamped between the bit
// patterns of 2^23 + 0 and 2^23 + 255.  Measured against normal_edge: 50 against 61 half-cost units per inner edge.
#ifndef WP_EDGE_SKIP
#define WP_EDGE_SKIP 0      // measured: q90 stream 7.27 -> 6.94 ms, q75 4.98 -> 5.05, mixed pipelined step unchanged within noise: off
#endif
constexpr uint32_t kFBias = 0x4B000000u;                 // 2^23
constexpr uint32_t kFMagicBits = 0x4B400000u;            // 1.5 * 2^23
__device__ __forceinline__ float fsample(uint32_t bits) { return __uint_as_float(bits); }
__device__ __forceinline__ float fmax_abs2(float a, float b) {
    float r;
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fmax_abs3(float a, float b, float c) {      // max(|a|, |b|, c), c >= 0
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float fma_abs4(float a, float b) {                // 4 |a| + |b|
    float r;
    asm("v_fma_f32 %0, |%1|, 4.0, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// DEN = the denormal domain: a sample x is the float whose bit pattern is the integer x itself, x * 2^-149.  Kernels are
// compiled with fp32 denormals preserved (.amdhsa_float_denorm_mode_32 3; add / fma / max / med3 / compare run at full rate
// on them), differences and 4A + B are still exact, thresholds are the record's integers as they are, and samples need no bias
// going in (byte loads, unpacked dwords) or out.  Signed results (the filter value) are sign-magnitude bit patterns: adding
// 2048 * 2^-149 first makes them positive, and the integer is the bit pattern minus 2048.
constexpr uint32_t kDenMagicBits = 0x00000800u;
template <bool DEN>
__device__ __forceinline__ int f_to_int(float a) {                           // a: an integer (times 2^-149 if DEN) of magnitude < 2048
    constexpr uint32_t m = DEN ? kDenMagicBits : kFMagicBits;
    return int(__float_as_uint(a + __uint_as_float(m)) - m);
}
template <bool DEN>
__device__ __forceinline__ uint32_t fclamp255(uint32_t bits) {
    constexpr uint32_t b = DEN ? 0u : kFBias;
    return uint32_t(clampi(int(bits), int(b), int(b + 255u)));
}

// Differences are written in position order, so that the one an edge shares with the next edge of the line (its q2 - q3 is
// the next one's p1 - p0 when neither sample changed) is computed once by common-subexpression elimination.
template <bool FOUR, bool DEN = false>
__device__ __forceinline__ void normal_edge_f(uint32_t p3, uint32_t& p2, uint32_t& p1, uint32_t& p0, uint32_t& q0, uint32_t& q1, uint32_t& q2, uint32_t q3,
                                              float thr, float ilevel, float hlevel, bool use) {
    const float d32 = fsample(p3) - fsample(p2), d21 = fsample(p2) - fsample(p1), d10 = fsample(p1) - fsample(p0);
    const float e01 = fsample(q0) - fsample(q1), e12 = fsample(q1) - fsample(q2), e23 = fsample(q2) - fsample(q3);
    const float e0 = fsample(q0) - fsample(p0), d1 = fsample(p1) - fsample(q1);
    const float m10 = fmax_abs2(d10, e01);
    const float worst = fmax_abs3(e12, e23, fmax_abs3(d32, d21, m10));
    const bool on = use & (fma_abs4(e0, d1) <= thr) & (worst <= ilevel);
#if WP_EDGE_SKIP
    if (__builtin_amdgcn_ballot_w64(on) == 0) return;      // no lane of the wavefront filters this edge
#endif
    const bool hev = m10 > hlevel;
    const float lim = DEN ? __uint_as_float(127u) : 127.0f;
    const float outer = __builtin_amdgcn_fmed3f(d1, -lim, lim);
    if (FOUR) {
        const float af = __builtin_fmaf(e0, 3.0f, hev ? outer : 0.0f);
        const int a = f_to_int<DEN>(on ? af : 0.0f);
        const int a1 = clamp15((a + 4) >> 3);
        const int a2 = clamp15((a + 3) >> 3);
        const int a3 = hev ? 0 : (a1 + 1) >> 1;
        p1 = fclamp255<DEN>(p1 + uint32_t(a3));
        p0 = fclamp255<DEN>(p0 + uint32_t(a2));
        q0 = fclamp255<DEN>(q0 - uint32_t(a1));
        q1 = fclamp255<DEN>(q1 - uint32_t(a3));
    } else {
        const float wf = __builtin_fmaf(e0, 3.0f, outer);
        const int w = f_to_int<DEN>(on ? wf : 0.0f);
        const int a = clamp127(w);
        const int c1 = (__mul24(a, 27) + 63) >> 7, c2 = (__mul24(a, 18) + 63) >> 7, c3 = (__mul24(a, 9) + 63) >> 7;
        const int dp0 = hev ? clamp15((w + 3) >> 3) : c1;
        const int dq0 = hev ? clamp15((w + 4) >> 3) : c1;
        const int dd1 = hev ? 0 : c2;
        const int dd2 = hev ? 0 : c3;
        p2 = fclamp255<DEN>(p2 + uint32_t(dd2));
        p1 = fclamp255<DEN>(p1 + uint32_t(dd1));
        p0 = fclamp255<DEN>(p0 + uint32_t(dp0));
        q0 = fclamp255<DEN>(q0 - uint32_t(dq0));
        q1 = fclamp255<DEN>(q1 - uint32_t(dd1));
        q2 = fclamp255<DEN>(q2 - uint32_t(dd2));
    }
}

// filter_line<N, false> on biased-float samples; level_f / ilevel_f / hlevel_f are the record's bytes as floats
template <int N>
__device__ __forceinline__ void filter_line_f(uint32_t (&v)[N], bool mb_edge, bool inner, float level_f, float ilevel_f, float hlevel_f) {
    const float thr = __builtin_fmaf(level_f, 2.0f, 1.0f);
    normal_edge_f<false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], thr + 8.0f, ilevel_f, hlevel_f, mb_edge);
    normal_edge_f<true>(v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], thr, ilevel_f, hlevel_f, inner);
    if constexpr (N == 20) {
        normal_edge_f<true>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], thr, ilevel_f, hlevel_f, inner);
        normal_edge_f<true>(v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19], thr, ilevel_f, hlevel_f, inner);
    }
}
// the same in the denormal domain: v[] holds plain integers 0..255, level / ilevel / hlevel are the record's bytes
template <int N>
__device__ __forceinline__ void filter_line_d(uint32_t (&v)[N], bool mb_edge, bool inner, int level, int ilevel, int hlevel) {
    const float thr = __uint_as_float(uint32_t(2 * level + 1)), thr_mb = __uint_as_float(uint32_t(2 * level + 9));
    const float il = __uint_as_float(uint32_t(ilevel)), hl = __uint_as_float(uint32_t(hlevel));
    normal_edge_f<false, true>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], thr_mb, il, hl, mb_edge);
    normal_edge_f<true, true>(v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], thr, il, hl, inner);
    if constexpr (N == 20) {
        normal_edge_f<true, true>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], thr, il, hl, inner);
        normal_edge_f<true, true>(v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19], thr, il, hl, inner);
    }
}
// the simple filter (filter2, Filter.hx:29-50) likewise: luma only, one edge position
__device__ __forceinline__ void simple_edge_d(uint32_t p1, uint32_t& p0, uint32_t& q0, uint32_t q1, float thr, bool use) {
    const float e0 = fsample(q0) - fsample(p0), d1 = fsample(p1) - fsample(q1);
    const bool on = use & (fma_abs4(e0, d1) <= thr);
    const float lim = __uint_as_float(127u);
    const float wf = __builtin_fmaf(e0, 3.0f, __builtin_amdgcn_fmed3f(d1, -lim, lim));
    const int a = f_to_int<true>(on ? wf : 0.0f);
    p0 = fclamp255<true>(p0 + uint32_t(clamp15((a + 3) >> 3)));
    q0 = fclamp255<true>(q0 - uint32_t(clamp15((a + 4) >> 3)));
}
__device__ __forceinline__ void simple_line_d(uint32_t (&v)[20], bool mb_edge, bool inner, int level) {
    const float thr = __uint_as_float(uint32_t(2 * level + 1)), thr_mb = __uint_as_float(uint32_t(2 * level + 9));
    simple_edge_d(v[2], v[3], v[4], v[5], thr_mb, mb_edge);
    simple_edge_d(v[6], v[7], v[8], v[9], thr, inner);
    simple_edge_d(v[10], v[11], v[12], v[13], thr, inner);
    simple_edge_d(v[14], v[15], v[16], v[17], thr, inner);
}
__device__ __forceinline__ void unpack4u(uint32_t w, uint32_t* v) { v[0] = w & 0xff; v[1] = (w >> 8) & 0xff; v[2] = (w >> 16) & 0xff; v[3] = w >> 24; }
__device__ __forceinline__ uint32_t pack4u(const uint32_t* v) { return v[0] | v[1] << 8 | v[2] << 16 | v[3] << 24; }
// bytes of a dword -> four biased samples, and back (the clamped bit patterns carry the sample in their low byte)
__device__ __forceinline__ void unpack4f(uint32_t w, uint32_t* v) {
    v[0] = __builtin_amdgcn_perm(kFBias, w, 0x070c0c00u); v[1] = __builtin_amdgcn_perm(kFBias, w, 0x070c0c01u);
    v[2] = __builtin_amdgcn_perm(kFBias, w, 0x070c0c02u); v[3] = __builtin_amdgcn_perm(kFBias, w, 0x070c0c03u);
}
__device__ __forceinline__ uint32_t pack4f(const uint32_t* v) {
    const uint32_t lo = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0400u), hi = __builtin_amdgcn_perm(v[3], v[2], 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// One band.  `q` is the band's sequence number in this workgroup, `q_prev` that of the band above (band > 0), `progress`
// the workgroup's LDS progress words (one per wavefront, W of them).  The caller has made sure that the band's R tiles
// are complete and visible.  off: the frame's loop filter is off (filterHeader.level == 0): every record counts as level 0,
// so the pass only re-tiles the frame (R tiles -> shifted S tiles, which is what the finishing pass reads).
template <int W, bool SIMPLE>
__device__ __forceinline__ void filter_band(const wp_vp8_frame& fr, int band, int n_bands, uint32_t q, uint32_t q_prev, int wave, uint32_t* progress,
                                            FilterWaveLds& lds, const uint32_t* __restrict__ filt, const uint8_t* __restrict__ Rp, uint8_t* Sp, bool off = false) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4;
    const int l = lane & 15;
    FilterTile& T = lds.tile[g];
    // lane roles in the row-wise phases
    const int lp = l >> 3, lr = l & 7;                    // chroma plane / chroma row
    const int hr = l >> 1, hh = l & 1;                    // hand-over: luma row 0..7, 8-byte half
    const int hcr = (l >> 1) & 3;                         // hand-over: chroma row 0..3 (plane = lp, 4-byte half = hh)
    const int mbw = fr.mbw, mbh = fr.mbh;
    constexpr bool simple = SIMPLE;                          // fr.filter_type == 1, resolved by the caller
    const uint8_t* const R = Rp + size_t(fr.mb_base) * kTileBytes;
    uint8_t* const S = Sp + size_t(fr.stile_base) * kTileBytes;
    const int spitch = mbw + 1;
    WP_PROF_DECL;
    const int y = band * kBandRows + g;
    const bool row_ok = y < mbh;
    const bool has_above = band > 0;
    const bool has_below = band + 1 < n_bands;
    const int prev_wave = int(q_prev % uint32_t(W));        // the wavefront that runs the band above
    const int rows_here = min(kBandRows, mbh - band * kBandRows);
    const int n_steps = mbw + 2 * (rows_here - 1);
    const bool last_row = g == rows_here - 1;             // its bottom rows leave the band through the S tiles

    // Everything a step reads from global memory is fetched one step ahead (registers), so no global
    // round trip sits on a step's critical path: the MB's own tile and filter record, and -- for the
    // band's first row -- the eight rows above (which makes this band trail the one above by 3 MBs).
    const int yy = row_ok ? y : 0;
    const uint8_t* rrow = R + size_t(yy) * mbw * kTileBytes;                 // R tiles of this MB row
    uint8_t* srow = S + size_t(yy) * spitch * kTileBytes;                    // S tiles of tile row y
    const uint32_t* frow = filt + size_t(fr.mb_base) + size_t(yy) * mbw;
    const bool top_lane = has_above && g == 0;
    // above rows of MB x: S(x, y) bytes 8..15 of rows 0..7 (cols 0..7) and S(x+1, y) bytes 0..7 (cols 8..15)
    const uint8_t* trow_y = srow + size_t(hh) * kTileBytes + 16 * hr + 8 * (1 - hh);
    const uint8_t* trow_c = srow + size_t(hh) * kTileBytes + kTileCb + 64 * lp + 8 * hcr + 4 * (1 - hh);
    uint4 nx_y = make_uint4(0, 0, 0, 0);
    uint2 nx_c = make_uint2(0, 0), nx_ty = make_uint2(0, 0);
    uint32_t nx_fp = 0, nx_tc = 0;
    if (has_above) wait_progress(&progress[prev_wave], (q_prev << kSeqShift) | uint32_t(min(2, mbw)));
    if (row_ok && g == 0) {
        nx_y = ld_nt(reinterpret_cast<const uint4*>(rrow + 16 * l));
        nx_c = ld_nt(reinterpret_cast<const uint2*>(rrow + kTileCb + 8 * l));
        nx_fp = off ? 0u : frow[0];
        if (top_lane) { nx_ty = *reinterpret_cast<const uint2*>(trow_y); nx_tc = *reinterpret_cast<const uint32_t*>(trow_c); }
    }
    settle(nx_y); settle(nx_ty); settle(nx_c); settle(nx_tc); settle(nx_fp);
    uint32_t pending_publish = 0;

    WP_PROF_MARK(9);
    for (int s = 0; s < n_steps; ++s) {
        WP_PROF_MARK(8);
        const int x = s - 2 * g;
        const bool act = row_ok && x >= 0 && x < mbw;
        const uint4 cur_y = nx_y;
        const uint2 cur_c = nx_c, cur_ty = nx_ty;
        const uint32_t fp = nx_fp, cur_tc = nx_tc;
        if (has_above && s + 1 < mbw)
            wait_progress(&progress[prev_wave], (q_prev << kSeqShift) | uint32_t(min(s + 3, mbw)));
        if (row_ok && x + 1 >= 0 && x + 1 < mbw) {
            const uint32_t o = uint32_t(x + 1) * uint32_t(kTileBytes);      // 32-bit offset inside the row (mbw <= 1024)
            nx_y = ld_nt(reinterpret_cast<const uint4*>(rrow + o + 16 * l));
            nx_c = ld_nt(reinterpret_cast<const uint2*>(rrow + o + kTileCb + 8 * l));
            nx_fp = off ? 0u : frow[x + 1];
            if (top_lane) { nx_ty = *reinterpret_cast<const uint2*>(trow_y + o); nx_tc = *reinterpret_cast<const uint32_t*>(trow_c + o); }
        }

        WP_PROF_MARK(0);
        const int level = int(fp & 0xff), ilevel = int((fp >> 8) & 0xff), hlevel = int((fp >> 16) & 0xff);
        const bool inner = (fp >> 24) & 1;
        const bool do_filter = act && level != 0;                   // `if (f.level == 0) continue` (:421, :449)
        const int cb = 16 * (x & 1), pb = 16 - cb;                  // luma column base of this MB's slot / the previous MB's
        const int cb8 = cb >> 1, pb8 = pb >> 1;
        if (act) {
            // ---- 1. own macroblock (nobody has touched it yet) and the rows above it ----
            t128(&T.y[8 + l][cb]) = cur_y;
            t64(&T.c[lp][4 + lr][cb8]) = cur_c;
            if (y > 0) {
                if (g > 0) {
                    t64(&T.y[hr][cb + 8 * hh]) = t64(&lds.ring_y[g][hr][(16 * x + 8 * hh) & 63]);
                    t32(&T.c[lp][hcr][cb8 + 4 * hh]) = t32(&lds.ring_c[g][lp][hcr][(8 * x + 4 * hh) & 31]);
                } else {
                    t64(&T.y[hr][cb + 8 * hh]) = cur_ty;
                    t32(&T.c[lp][hcr][cb8 + 4 * hh]) = cur_tc;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        WP_PROF_MARK(1);
        __builtin_amdgcn_s_setprio(1);      // issue priority rises along the step (a wavefront close to publishing goes first): 6 % faster
        if (do_filter) {
            // ---- 2a. vertical edges: lane = pixel row; columns -4..15 relative to the MB ----
            {
                int v[20];
                unpack4(t32(&T.y[8 + l][pb + 12]), v);
                const uint4 w = t128(&T.y[8 + l][cb]);
                unpack4(w.x, v + 4); unpack4(w.y, v + 8); unpack4(w.z, v + 12); unpack4(w.w, v + 16);
                filter_line<20, SIMPLE>(v, x > 0, inner, level, ilevel, hlevel);
                t32(&T.y[8 + l][pb + 12]) = pack4i(v);
                t128(&T.y[8 + l][cb]) = make_uint4(pack4i(v + 4), pack4i(v + 8), pack4i(v + 12), pack4i(v + 16));
            }
            if (!simple) {
                int v[12];
                unpack4(t32(&T.c[lp][4 + lr][pb8 + 4]), v);
                const uint2 w = t64(&T.c[lp][4 + lr][cb8]);
                unpack4(w.x, v + 4); unpack4(w.y, v + 8);
                filter_line<12, false>(v, x > 0, inner, level, ilevel, hlevel);
                t32(&T.c[lp][4 + lr][pb8 + 4]) = pack4i(v);
                t64(&T.c[lp][4 + lr][cb8]) = make_uint2(pack4i(v + 4), pack4i(v + 8));
            }
        }
        __builtin_amdgcn_wave_barrier();
        WP_PROF_MARK(2);
        __builtin_amdgcn_s_setprio(2);
        if (do_filter) {
            // ---- 2b. horizontal edges: lane = pixel column; rows -4..15 relative to the MB ----
            {
                int v[20];
#pragma unroll
                for (int k = 0; k < 20; ++k) v[k] = T.y[4 + k][cb + l];
                filter_line<20, SIMPLE>(v, y > 0, inner, level, ilevel, hlevel);
#pragma unroll
                for (int k = 1; k < 19; ++k) T.y[4 + k][cb + l] = uint8_t(v[k]);
            }
            if (!simple) {
                int v[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) v[k] = T.c[lp][k][cb8 + lr];
                filter_line<12, false>(v, y > 0, inner, level, ilevel, hlevel);
#pragma unroll
                for (int k = 1; k < 11; ++k) T.c[lp][k][cb8 + lr] = uint8_t(v[k]);
            }
        }
        __builtin_amdgcn_wave_barrier();

        WP_PROF_MARK(3);
        __builtin_amdgcn_s_setprio(0);
        // ---- tell the band below how far this band's last row has got: the value of the PREVIOUS step,
        //      whose stores were issued a whole step ago (the release fence then does not wait for acks) ----
        settle(nx_y); settle(nx_ty); settle(nx_c); settle(nx_tc); settle(nx_fp);   // prefetch landed long ago; wait before the stores go out
        if (has_below) {
            if (pending_publish) publish_progress(&progress[wave], pending_publish);
            const int done = s - 2 * (kBandRows - 1) + 1;
            pending_publish = done >= 1 ? ((q << kSeqShift) | uint32_t(done)) : 0u;
        }

        WP_PROF_MARK(4);
        if (act) {
            // ---- 3. emit S tile (x, y): rows -8..7, cols -8..7 relative to the MB, all final ----
            uint8_t* st = srow + uint32_t(x) * uint32_t(kTileBytes);
            const uint2 a = t64(&T.y[l][pb + 8]), b = t64(&T.y[l][cb]);
            st_nt(reinterpret_cast<uint4*>(st + 16 * l), make_uint4(a.x, a.y, b.x, b.y));
            st_nt(reinterpret_cast<uint2*>(st + kTileCb + 8 * l), make_uint2(t32(&T.c[lp][lr][pb8 + 4]), t32(&T.c[lp][lr][cb8])));
            if (x == mbw - 1) {        // right-most tile column: cols 8..15 (left half of tile mbw)
                *reinterpret_cast<uint2*>(st + kTileBytes + 16 * l) = t64(&T.y[l][cb + 8]);
                *reinterpret_cast<uint32_t*>(st + kTileBytes + kTileCb + 8 * l) = t32(&T.c[lp][lr][cb8 + 4]);
            }
            // ---- 4. hand the eight bottom rows (cols -8..7; state after this MB) down ----
            const uint2 vy = t64(&T.y[16 + hr][hh ? cb : pb + 8]);
            const uint32_t vc = t32(&T.c[lp][8 + hcr][hh ? cb8 : pb8 + 4]);
            if (!last_row) {
                t64(&lds.ring_y[g + 1][hr][(16 * x - 8 + 8 * hh) & 63]) = vy;
                t32(&lds.ring_c[g + 1][lp][hcr][(8 * x - 4 + 4 * hh) & 31]) = vc;
                if (x == mbw - 1 && hh) {   // cols 8..15 of the last MB
                    t64(&lds.ring_y[g + 1][hr][(16 * x + 8) & 63]) = t64(&T.y[16 + hr][cb + 8]);
                    t32(&lds.ring_c[g + 1][lp][hcr][(8 * x + 4) & 31]) = t32(&T.c[lp][8 + hcr][cb8 + 4]);
                }
            } else {
                uint8_t* sn = st + uint32_t(spitch) * uint32_t(kTileBytes);       // S tile (x, y+1); tile row mbh exists
                *reinterpret_cast<uint2*>(sn + 16 * hr + 8 * hh) = vy;
                *reinterpret_cast<uint32_t*>(sn + kTileCb + 64 * lp + 8 * hcr + 4 * hh) = vc;
                if (x == mbw - 1 && hh) {
                    *reinterpret_cast<uint2*>(sn + kTileBytes + 16 * hr) = t64(&T.y[16 + hr][cb + 8]);
                    *reinterpret_cast<uint32_t*>(sn + kTileBytes + kTileCb + 64 * lp + 8 * hcr) = t32(&T.c[lp][8 + hcr][cb8 + 4]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        WP_PROF_MARK(5);
    }
    if (has_below && pending_publish) publish_progress(&progress[wave], pending_publish);
    WP_PROF_FLUSH(0);
}

}  // namespace wpk
