// Reconstruction of one band (4 macroblock rows) of one frame by one wavefront: device code shared by the stand-alone
// reconstruction kernel (vp8_recon.hip) and the fused lossy kernel (vp8_lossy.hip).  See vp8_recon.hip for the design.
#pragma once
#include "wp_device.h"

namespace wpk {

struct ReconTile {              // bytes; row pitch 36 = 9 dwords: 4-pixel groups stay dword aligned, and both the column
                                // accesses (lane = row) and the i4 block accesses (lane = block) hit 16 different banks
    uint8_t y[17][36];          // row 0: [3] corner, [4..19] top, [20..23] top-right; rows 1..16: [3] left, [4..19] pixels
    uint8_t c[9][36];           // Cb at cols 3 | 4..11, Cr at cols 19 | 20..27; row 0 is the top border
    uint8_t pad[24];            // 960 B = 240 dwords: the four groups' tiles start 16 banks apart
};
struct ReconWaveLds {
    ReconTile tile[kBandRows];
    uint32_t ring[kBandRows][4][8];   // ring[g][col & 3]: bottom row of the MB above group g: Y (4 dwords), U (2), V (2)
};

// Inputs are read through buffer resources (wave-uniform base in scalar registers + a 32-bit byte offset per lane): a lane
// with nothing to fetch uses an out-of-range offset and gets zeros back, without a branch and without memory traffic, and
// the hardware range check keeps a malformed block index inside the window.
typedef int v4i_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kOob = 0xFFFFFF00u;                   // beyond every buffer resource below: the load returns zeros
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void* p, uint32_t bytes) {      // raw buffer of `bytes` bytes from p
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, int(bytes), 0x00020000);
}
// the frame's own coefficient blocks when the descriptor bounds them, else a 2 GB window (offsets are 32-bit; the largest
// frame, 1024 x 1024 macroblocks x 25 blocks, stays below it)
__device__ __forceinline__ uint32_t coeff_window_bytes(const wp_vp8_frame& fr) {
    return (fr.flags & WP_FRAME_COEFF_BOUNDED) ? min(fr.coeff_blocks, 0x03ffffffu) * 32u : 0x80000000u;
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t off, bool nt) {
    const v4i_t v = nt ? __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2) : __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(uint32_t(v.x), uint32_t(v.y), uint32_t(v.z), uint32_t(v.w));
}

__device__ __forceinline__ uint32_t& tile32(uint8_t* row, int col) { return *reinterpret_cast<uint32_t*>(row + col); }

__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) { return uint32_t(a) | uint32_t(b) << 8 | uint32_t(c) << 16 | uint32_t(d) << 24; }
// byte 0 of v in all four bytes (one v_perm_b32)
__device__ __forceinline__ uint32_t splat4(int v) { return __builtin_amdgcn_perm(uint32_t(v), uint32_t(v), 0u); }

// Wrapping 32-bit multiply, arithmetic >> 16 (Vp8Decoder.hx:913-914).  The operands of every call fit in 24
// signed bits (coefficients are int16, the WHT DC < 2^18, first-pass outputs < 2^18, the constants < 2^17), so the low
// 32 bits of the full-rate 24-bit multiply equal the wrapping 32-bit product (v_mul_lo_u32 is quarter rate).
__device__ __forceinline__ int mul16(int a, int c) { return __mul24(a, c) >> 16; }


// Residual of one 4x4 block: inverseDCT4 without the "+ prediction, clip8" tail
// (Vp8Decoder.hx:905-935).  Row j comes back as two packed int16 pairs in pixel order:
// even[j] = (r[j][0], r[j][1]), odd[j] = (r[j][2], r[j][3]) (the names are historical), each saturated to int16
// (clip8(pred + r), computed with a saturating add, cannot tell larger magnitudes apart, pred being in 0..255).
__device__ __forceinline__ void idct_residual(const int (&c)[16], uint32_t (&even)[4], uint32_t (&odd)[4]) {
    const int c1 = 85627, c2 = 35468;
    int m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = c[i] + c[8 + i];
        const int b = c[i] - c[8 + i];
        const int cc = mul16(c[4 + i], c2) - mul16(c[12 + i], c1);
        const int d = mul16(c[4 + i], c1) + mul16(c[12 + i], c2);
        m[i][0] = a + d; m[i][1] = b + cc; m[i][2] = b - cc; m[i][3] = a - d;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dc = m[0][j] + 4;
        const int a = dc + m[2][j];
        const int b = dc - m[2][j];
        const int cc = mul16(m[1][j], c2) - mul16(m[3][j], c1);
        const int d = mul16(m[1][j], c1) + mul16(m[3][j], c2);
        // v_cvt_pk_i16_i32 saturates to int16 and packs in one instruction
        even[j] = as_u32(__builtin_amdgcn_cvt_pk_i16((a + d) >> 3, (b + cc) >> 3));
        odd[j] = as_u32(__builtin_amdgcn_cvt_pk_i16((b - cc) >> 3, (a - d) >> 3));
    }
}

// Unpacks a stored coefficient block (16 x int16, natural order; two 16-byte halves) into ints.
__device__ __forceinline__ void unpack_block(const uint4& a, const uint4& b, int (&c)[16]) {
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) { c[2 * i] = int(int16_t(w[i] & 0xffff)); c[2 * i + 1] = int(w[i]) >> 16; }
}

// inverseWHT16 (Vp8Decoder.hx:960-989) spread over the 16 lanes of a group: lane (r, c) holds input
// coefficient 4r+c and ends up with the DC of luma block 4r+c.  int32 throughout, no truncation.
__device__ __forceinline__ int wht_dc_lanes(int v, int l) {
    const int r = l >> 2, c = l & 3;
    const int i0 = __shfl(v, c, 16), i1 = __shfl(v, 4 + c, 16), i2 = __shfl(v, 8 + c, 16), i3 = __shfl(v, 12 + c, 16);
    const int a0 = i0 + i3, a1 = i1 + i2, a2 = i1 - i2, a3 = i0 - i3;
    const int m = r == 0 ? a0 + a1 : r == 1 ? a3 + a2 : r == 2 ? a0 - a1 : a3 - a2;      // m[4r + c]
    // second pass: the four values of a row sit in one quad of lanes: DPP quad broadcasts, no LDS round trip
    const int m0 = __builtin_amdgcn_mov_dpp(m, 0x00, 0xf, 0xf, false), m1 = __builtin_amdgcn_mov_dpp(m, 0x55, 0xf, 0xf, false),
              m2 = __builtin_amdgcn_mov_dpp(m, 0xAA, 0xf, 0xf, false), m3 = __builtin_amdgcn_mov_dpp(m, 0xFF, 0xf, 0xf, false);
    const int dc = m0 + 3;
    const int b0 = dc + m3, b1 = m1 + m2, b2 = m1 - m2, b3 = dc - m3;
    const int o = c == 0 ? b0 + b1 : c == 1 ? b3 + b2 : c == 2 ? b0 - b1 : b3 - b2;
    return o >> 3;
}

// ---- packed-byte helpers for the predictors: four pixels per dword, one instruction per four averages ----
// v_lerp_u8: per byte (a + b + (c & 1)) >> 1.  avg2(a, b) = (a + b + 1) >> 1 is the rounding-up form, and
// avg3(a, b, c) = (a + 2b + c + 2) >> 2 == up(down(a, c), b):  with a + c = 2m + r (r = 0 or 1) the left side is
// floor((m + b + 1) / 2 + r / 4), and r / 4 never carries past the half-integer grid of (m + b + 1) / 2.
__device__ __forceinline__ uint32_t avg2p(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }
__device__ __forceinline__ uint32_t avg3p(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_lerp(__builtin_amdgcn_lerp(a, c, 0u), b, 0x01010101u);
}
// bytes n..n+3 of the 8-byte value {hi, lo}
#define WP_ALB(hi, lo, n) __builtin_amdgcn_alignbyte((hi), (lo), (n))
// byte selector: selector bytes 0..3 pick from `lo`, 4..7 from `hi`
#define WP_PERM(hi, lo, sel) __builtin_amdgcn_perm((hi), (lo), (sel))

// clip8(pred + residual) for one row of four pixels (Vp8Decoder.hx:930-933, 991-996).
__device__ __forceinline__ uint32_t add_clip_row(uint32_t pred, uint32_t even, uint32_t odd) {
    return clip_pack_seq(__builtin_elementwise_add_sat(as_s2(WP_PERM(0u, pred, 0x0c010c00u)), as_s2(even)),     // pixels 0, 1 as int16
                         __builtin_elementwise_add_sat(as_s2(WP_PERM(0u, pred, 0x0c030c02u)), as_s2(odd)));     // pixels 2, 3
}

// TrueMotion (PredFunc.hx TM variants): row r = clip(top + left[r] - corner), in packed 16-bit arithmetic
__device__ __forceinline__ void pred_tm(uint32_t top4, uint32_t left4, int corner, uint32_t (&row)[4]) {
    const short2_t c2 = as_s2(uint32_t(corner) * 0x00010001u);
    const short2_t t01 = as_s2(WP_PERM(0u, top4, 0x0c010c00u)) - c2, t23 = as_s2(WP_PERM(0u, top4, 0x0c030c02u)) - c2;
    const uint32_t sel[4] = {0x0c000c00u, 0x0c010c01u, 0x0c020c02u, 0x0c030c03u};      // (left[r], 0, left[r], 0)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const short2_t lr = as_s2(WP_PERM(0u, left4, sel[r]));
        row[r] = clip_pack_seq(t01 + lr, t23 + lr);
    }
}

// The ten 4x4 luma predictors (PredFunc.hx:118-418).  top_lo / top_hi = the 8 pixels above (4 above the block + 4
// above-right), left4 = the 4 pixels to the left (byte r = row r), corner = above-left.  The eight directional modes
// read windows of the edge vector X = [L3 L2 L1 L0 | C T0 T1 T2 | T3 T4 T5 T6 | T7 T7 T7 T7] (L3 repeated below, T7 to
// the right) through its 3-tap averages a[i] = avg3(X[i-1], X[i], X[i+1]) and 2-tap averages b[i] = avg2(X[i], X[i+1]).
// The lanes of a wavefront hold different modes, so the averages are computed once for all of them (four pixels per
// dword) and each mode only assembles its rows from them: the divergent part is a handful of byte permutes per mode.
__device__ __forceinline__ void pred4x4(int mode, uint32_t top_lo, uint32_t top_hi, uint32_t left4, int corner, uint32_t (&row)[4]) {
    if (mode == 0) {            // DC
        const uint32_t sum = __builtin_amdgcn_sad_u8(top_lo, 0u, __builtin_amdgcn_sad_u8(left4, 0u, 4u));
        row[0] = row[1] = row[2] = row[3] = splat4(int(sum >> 3));
        return;
    }
    if (mode == 1) {            // TM
        pred_tm(top_lo, left4, corner, row);
        return;
    }
    const uint32_t X0 = WP_PERM(left4, left4, 0x00010203u);               // L3 L2 L1 L0
    const uint32_t X1 = (top_lo << 8) | uint32_t(corner);                 // C  T0 T1 T2
    const uint32_t X2 = WP_ALB(top_hi, top_lo, 3);                        // T3 T4 T5 T6
    const uint32_t X3 = WP_PERM(top_hi, top_hi, 0x03030303u);             // T7 T7 T7 T7
    const uint32_t N0 = WP_ALB(X1, X0, 1), N1 = WP_ALB(X2, X1, 1), N2 = WP_ALB(X3, X2, 1);     // X shifted by +1
    const uint32_t A0 = avg3p(WP_PERM(X0, X0, 0x02010000u), X0, N0);      // a0..a3
    const uint32_t A1 = avg3p(WP_ALB(X1, X0, 3), X1, N1);                 // a4..a7
    const uint32_t A2 = avg3p(WP_ALB(X2, X1, 3), X2, N2);                 // a8..a11
    const uint32_t B0 = avg2p(X0, N0), B1 = avg2p(X1, N1), B2 = avg2p(X2, N2);                 // b0..b3, b4..b7, b8..b11
    switch (mode) {
    case 2:     // VE (smoothed): a5..a8 in every row
        row[0] = row[1] = row[2] = row[3] = WP_ALB(A2, A1, 1);
        break;
    case 3:     // HE (smoothed): row r = a[3-r] four times
        row[0] = WP_PERM(A0, A0, 0x03030303u);
        row[1] = WP_PERM(A0, A0, 0x02020202u);
        row[2] = WP_PERM(A0, A0, 0x01010101u);
        row[3] = WP_PERM(A0, A0, 0x00000000u);
        break;
    case 4:     // RD: row r = a[4-r .. 7-r]
        row[0] = A1;
        row[1] = WP_ALB(A1, A0, 3);
        row[2] = WP_ALB(A1, A0, 2);
        row[3] = WP_ALB(A1, A0, 1);
        break;
    case 5:     // VR
        row[0] = B1;                                                       // b4 b5 b6 b7
        row[1] = A1;                                                       // a4 a5 a6 a7
        row[2] = WP_ALB(B1, A0, 3);                                        // a3 b4 b5 b6
        row[3] = WP_PERM(A1, A0, 0x06050402u);                             // a2 a4 a5 a6
        break;
    case 6: {   // LD: row r = a[6+r .. 9+r]
        const uint32_t A3 = avg3p(WP_ALB(X3, X2, 3), X3, X3);              // a12 in byte 0
        row[0] = WP_ALB(A2, A1, 2);
        row[1] = WP_ALB(A2, A1, 3);
        row[2] = A2;
        row[3] = WP_ALB(A3, A2, 1);
        break; }
    case 7:     // VL
        row[0] = WP_ALB(B2, B1, 1);                                        // b5 b6 b7 b8
        row[1] = WP_ALB(A2, A1, 2);                                        // a6 a7 a8 a9
        row[2] = WP_PERM(A2, WP_ALB(B2, B1, 2), 0x06020100u);              // b6 b7 b8 a10
        row[3] = WP_PERM(A2, WP_ALB(A2, A1, 3), 0x07020100u);              // a7 a8 a9 a11
        break;
    case 8: {   // HD
        const uint32_t As = WP_ALB(A1, A0, 1);                             // a1 a2 a3 a4
        row[0] = WP_PERM(A1, B0, 0x06050403u);                             // b3 a4 a5 a6
        row[1] = WP_PERM(As, B0, 0x07030602u);                             // b2 a3 b3 a4
        row[2] = WP_PERM(As, B0, 0x06020501u);                             // b1 a2 b2 a3
        row[3] = WP_PERM(As, B0, 0x05010400u);                             // b0 a1 b1 a2
        break; }
    default:    // 9: HU
        row[0] = WP_PERM(A0, B0, 0x05010602u);                             // b2 a2 b1 a1
        row[1] = WP_PERM(A0, B0, 0x04000501u);                             // b1 a1 b0 a0
        row[2] = WP_PERM(X0, WP_PERM(A0, B0, 0x00000400u), 0x04040100u);   // b0 a0 L3 L3
        row[3] = WP_PERM(X0, X0, 0x00000000u);                             // L3 L3 L3 L3
        break;
    }
}

// 8x8 / 16x16 predictors restricted to one 4x4 sub-block (PredFunc.hx:420-572).  `mode` has
// already been through checkTopLeftPred (PredFunc.hx:57-68); dcval is the block-wide DC value.
__device__ __forceinline__ void pred_large(int mode, uint32_t top4, uint32_t left4, int corner, int dcval, uint32_t (&row)[4]) {
    if (mode == 1) {            // TM
        pred_tm(top4, left4, corner, row);
    } else if (mode == 2) {     // VE
        row[0] = row[1] = row[2] = row[3] = top4;
    } else if (mode == 3) {     // HE
        row[0] = WP_PERM(left4, left4, 0x00000000u);
        row[1] = WP_PERM(left4, left4, 0x01010101u);
        row[2] = WP_PERM(left4, left4, 0x02020202u);
        row[3] = WP_PERM(left4, left4, 0x03030303u);
    } else {                    // DC and its top/left variants
        row[0] = row[1] = row[2] = row[3] = splat4(dcval);
    }
}

// Cross-lane sums with DPP (VALU modifiers, no LDS round trip).  A group of 16 lanes is one DPP row.
template <int CTRL> __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E, kDppHalfMirror = 0x141, kDppRowRor = 0x120;
// Sum over the 16 lanes of a group, result in every lane (rotations by 8, 4, 2, 1).
__device__ __forceinline__ int group_sum16(int v) {
    v += dpp<kDppRowRor + 8>(v);
    v += dpp<kDppRowRor + 4>(v);
    v += dpp<kDppRowRor + 2>(v);
    v += dpp<kDppRowRor + 1>(v);
    return v;
}
// Sum over each half (8 lanes) of a group, result in every lane of the half.
__device__ __forceinline__ int group_sum8(int v) {
    v += dpp<kDppQuadXor1>(v);
    v += dpp<kDppQuadXor2>(v);
    v += dpp<kDppHalfMirror>(v);        // lane i <-> 7 - i: the other quad of the half
    return v;
}

// One band.  `q` is the band's sequence number in this workgroup, `q_prev` that of the band above (band > 0), `progress`
// the workgroup's LDS progress words (one per wavefront, W of them).  mark_done: also publish kBandDone when the band's
// last store has been issued (the fused kernel's loop-filter bands wait for that).
template <int W>
__device__ __forceinline__ void recon_band(const wp_vp8_frame& fr, int band, int n_bands, uint32_t q, uint32_t q_prev, int wave, uint32_t* progress,
                                           ReconWaveLds& lds, const uint4* __restrict__ mb, const int16_t* __restrict__ coeff,
                                           uint8_t* Rp, bool mark_done) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4;            // macroblock row within the band
    const int l = lane & 15;            // 4x4 luma block owned by this lane
    const int bj = l >> 2, bi = l & 3;
    ReconTile& T = lds.tile[g];
    const int mbw = fr.mbw, mbh = fr.mbh;
    uint8_t* const R = Rp + size_t(fr.mb_base) * kTileBytes;      // this frame's R tiles
    WP_PROF_DECL;
    const int y = band * kBandRows + g;
    const bool row_ok = y < mbh;
    const bool has_above = band > 0;
    const bool has_below = band + 1 < n_bands;
    const int prev_wave = int(q_prev % uint32_t(W));        // the wavefront that runs the band above
    const int rows_here = min(kBandRows, mbh - band * kBandRows);
    const int n_steps = mbw + 2 * (rows_here - 1);

    // No global round trip sits on a step's critical path: the MB record is fetched two steps ahead, and with
    // it the lane's luma coefficient block and WHT coefficient one step ahead (registers).  prefetch(s) is what
    // step s issues: the record of MB x+2 and, from the record of MB x+1 (fetched by the previous call), the
    // coefficients of MB x+1.
    uint8_t* const Rrow = R + size_t(row_ok ? y : 0) * mbw * kTileBytes;      // R tiles of this lane's MB row
    const __amdgpu_buffer_rsrc_t mb_rs = buffer_of(mb + size_t(fr.mb_base), uint32_t(mbw) * uint32_t(mbh) * 16u);
    const __amdgpu_buffer_rsrc_t co_rs = buffer_of(coeff + size_t(fr.coeff_base) * 16, coeff_window_bytes(fr));
    uint4 rec_n1 = make_uint4(0, 0, 0, 0), rec_n2 = make_uint4(0, 0, 0, 0);     // records of MB x+1, x+2
    uint4 nl0 = make_uint4(0, 0, 0, 0), nl1 = make_uint4(0, 0, 0, 0);           // luma block l of MB x+1
    uint32_t nwv = 0;                                                           // WHT coefficient l of MB x+1
    uint4 nc0 = make_uint4(0, 0, 0, 0), nc1 = make_uint4(0, 0, 0, 0);           // chroma block of MB x+1 (chroma lanes)
    const int cn = 16 + (l & 3) + ((l & 8) ? 4 : 0);                            // lanes 0-3: Cb blocks, 8-11: Cr blocks
    const bool chroma_lane = (l & 4) == 0;
    // Every lane issues every load of the prefetch, every step: a lane that has nothing to fetch (outside the frame, no
    // stored block, no WHT) reads at an out-of-range offset and gets zeros.  No branches, nothing to clear beforehand, and
    // -- the count of loads per step being fixed -- the compiler can wait for the top-row loads issued before them without
    // also waiting for these.
    auto prefetch = [&](int s_) {
        const int x_ = s_ - 2 * g;
        rec_n1 = rec_n2;                                            // all zero for a lane whose MB x+1 is outside the frame
        const bool in2 = row_ok && x_ + 2 >= 0 && x_ + 2 < mbw;
        rec_n2 = buf_load16(mb_rs, in2 ? (uint32_t(y) * uint32_t(mbw) + uint32_t(x_ + 2)) * 16u : kOob, false);
        const uint32_t w2n = rec_n1.z;
        const uint32_t whtn = (w2n >> WP_MB_WHT_BIT) & 1u;
        const uint32_t blk0 = rec_n1.w + whtn;                                  // first non-WHT block of MB x+1
        const uint32_t lo = ((w2n >> l) & 1) ? (blk0 + uint32_t(__popc(w2n & ((1u << l) - 1u)))) * 32u : kOob;
        nl0 = buf_load16(co_rs, lo, true); nl1 = buf_load16(co_rs, lo + 16u, true);
        nwv = uint32_t(int(int16_t(__builtin_amdgcn_raw_buffer_load_b16(co_rs, whtn ? rec_n1.w * 32u + 2u * uint32_t(l) : kOob, 0, 0))));
        const uint32_t cofs = (chroma_lane && ((w2n >> cn) & 1)) ? (blk0 + uint32_t(__popc(w2n & ((1u << cn) - 1u)))) * 32u : kOob;
        nc0 = buf_load16(co_rs, cofs, true); nc1 = buf_load16(co_rs, cofs + 16u, true);
    };
    prefetch(-2);
    prefetch(-1);
    // Top context of group 0: the bottom rows of the band above, read from its output tiles one step ahead (like the
    // coefficient blocks, so that no load result is needed in the middle of a step).  Lanes 0-3 fetch the last luma row,
    // 4-5 the last Cb row, 6-7 the last Cr row of MB (column, y-1); top_n holds column s+1 at the top of step s.
    const bool top_lane = has_above && g == 0 && l < 8;
    const uint8_t* const top_src = R + size_t(has_above ? y - 1 : 0) * mbw * kTileBytes
                                 + (l < 4 ? 240 + 4 * l : l < 6 ? kTileCb + 56 + 4 * (l - 4) : kTileCr + 56 + 4 * (l - 6));
    uint32_t top_n = 0;
    if (has_above) {
        wait_progress(&progress[prev_wave], (q_prev << kSeqShift) | uint32_t(min(2, mbw)));
        if (top_lane) {
            lds.ring[0][0][l] = *reinterpret_cast<const uint32_t*>(top_src);
            if (mbw > 1) top_n = *reinterpret_cast<const uint32_t*>(top_src + kTileBytes);
        }
    }
    settle(rec_n1); settle(rec_n2); settle(nl0); settle(nl1); settle(nwv); settle(nc0); settle(nc1); settle(top_n);
    uint32_t pending_publish = 0;

    WP_PROF_MARK(9);
    for (int s = 0; s < n_steps; ++s) {
        WP_PROF_MARK(8);
        const int x = s - 2 * g;
        const bool act = row_ok && x >= 0 && x < mbw;
        const uint4 rec = act ? rec_n1 : make_uint4(0, 0, 0, 0);
        const uint4 cl0 = nl0, cl1 = nl1, cc0 = nc0, cc1 = nc1;
        const int wv_cur = int(nwv);

        // ---- top context of group 0: park the column fetched during the previous step in the LDS ring, fetch the next ----
        if (has_above && s + 1 < mbw) {
            if (top_lane) lds.ring[0][(s + 1) & 3][l] = top_n;
            if (s + 2 < mbw) {
                wait_progress(&progress[prev_wave], (q_prev << kSeqShift) | uint32_t(min(s + 3, mbw)));
                if (top_lane) top_n = *reinterpret_cast<const uint32_t*>(top_src + uint32_t(s + 2) * uint32_t(kTileBytes));
            }
        }

        prefetch(s);           // nothing loaded in this step is used before the step's end

        WP_PROF_MARK(0);
        // ---- residuals (independent of the neighbours) ----
        const uint32_t w2 = rec.z;
        const bool i16 = (w2 >> WP_MB_I16_BIT) & 1;
        const int has_wht = (w2 >> WP_MB_WHT_BIT) & 1;

        uint32_t ev[4] = {0, 0, 0, 0}, od[4] = {0, 0, 0, 0};          // luma residual of block l
        {
            // No per-lane branches here: a lane without a stored block holds zeros from the prefetch, and zero
            // coefficients give zero residuals ((0 + 4) >> 3), so every lane runs the same code; only whole-wavefront
            // conditions (nobody has a WHT / nobody has anything to transform) skip work.
            int c[16];
            unpack_block(cl0, cl1, c);
            const bool present = act && ((w2 >> l) & 1);
            int dc16 = 0;
            if (__builtin_amdgcn_ballot_w64(act && has_wht) != 0) {
                const int wv = (act && has_wht) ? wv_cur : 0;
                dc16 = wht_dc_lanes(wv, l);
                if (act && has_wht) c[0] = dc16; else dc16 = 0;
            }
            if (__builtin_amdgcn_ballot_w64(present || dc16 != 0) != 0) idct_residual(c, ev, od);
        }
        uint32_t cev[4] = {0, 0, 0, 0}, cod[4] = {0, 0, 0, 0};
        {
            // inverseDCT8 / inverseDCT8DCOnly (Vp8Decoder.hx:937-958): a block whose AC coefficients are all zero adds
            // (dc + 4) >> 3 to every pixel, which is also what the full transform gives for it; the full transform
            // is only run when some lane of the wavefront has a chroma block with AC (rare below ~q85).
            const bool cpres = act && chroma_lane && ((w2 >> cn) & 1);
            const bool c_ac = cpres && (((cc0.x >> 16) | cc0.y | cc0.z | cc0.w | cc1.x | cc1.y | cc1.z | cc1.w) != 0);
            if (__builtin_amdgcn_ballot_w64(c_ac) != 0) {           // lanes without a chroma block hold zeros: zero residuals
                int c[16];
                unpack_block(cc0, cc1, c);
                idct_residual(c, cev, cod);
            } else {
                const int v = (int(int16_t(cc0.x & 0xffff)) + 4) >> 3;
                const uint32_t p = as_u32(__builtin_amdgcn_cvt_pk_i16(v, v));
#pragma unroll
                for (int r = 0; r < 4; ++r) { cev[r] = p; cod[r] = p; }
            }
        }
        WP_PROF_MARK(1);
        __builtin_amdgcn_wave_barrier();
        // Issue priority rises along the step: the prediction chain (and most of all the i4 mini-steps, which keep
        // few lanes busy) is what the next step and the band below wait for, the residuals above are not.
        __builtin_amdgcn_s_setprio(1);

        if (act) {
            // ---- prepareYBR (Vp8Decoder.hx:541-577): left columns, top rows, the rows-4/8/12 copy of the top-right
            //      (:572-576).  Every lane has a fixed set of small copy jobs and none of them reads what another one
            //      writes, so all LDS reads are issued first and all writes after one wait (the literal order --
            //      left column, then top row, then the copies -- is three dependent LDS round trips) ----
            {
                const int lc = min(l, 8);
                uint32_t va = T.y[l][19], vb = T.y[16][19], vc1 = T.c[lc][11], vc2 = T.c[lc][27];     // last column of the previous MB
                // top border from the ring: lanes 0-3 luma dwords, 6-9 chroma dwords; lanes 4 and 10-12 the top-right dword
                // for rows 0, 4, 8 and 12
                const bool tr_job = l == 4 || (l >= 10 && l <= 12);
                const uint32_t* slot = lds.ring[g][x & 3];
                uint32_t rv = slot[l < 4 ? l : (l >= 6 && l <= 9) ? l - 2 : 3];
                const uint32_t nxt = lds.ring[g][(x + 1) & 3][0];
                if (tr_job) rv = x == mbw - 1 ? splat4(int(rv >> 24)) : nxt;
                if (x == 0) { va = 0x81; vb = 0x81; vc1 = 0x81; vc2 = 0x81; }
                if (y == 0) {
                    rv = 0x7f7f7f7fu;
                    if (l == 0) { va = 0x7f; vc1 = 0x7f; vc2 = 0x7f; }                          // the corners belong to the top row
                }
                T.y[l][3] = uint8_t(va);
                if (l == 0) T.y[16][3] = uint8_t(vb);
                if (l < 9) { T.c[l][3] = uint8_t(vc1); T.c[l][19] = uint8_t(vc2); }
                uint8_t* dst = l < 4 ? &T.y[0][4 + 4 * l] : l == 4 ? &T.y[0][20] : l < 8 ? &T.c[0][4 + 4 * (l - 6)]
                             : l < 10 ? &T.c[0][20 + 4 * (l - 8)] : &T.y[4 * (l - 9)][20];
                if (l < 5 || (l >= 6 && l <= 12)) tile32(dst, 0) = rv;
            }
            __builtin_amdgcn_wave_barrier();

            WP_PROF_MARK(2);
            // ---- luma ----
            if (i16) {
                int mode = int(rec.x & 15);
                // checkTopLeftPred (PredFunc.hx:57-68): 10 = DCTop (mby==0), 11 = DCLeft (mbx==0), 12 = DCTopLeft
                const int tsum = group_sum16(int(T.y[0][4 + l]));
                const int lsum = group_sum16(int(T.y[1 + l][3]));
                int dcv = (tsum + lsum + 16) >> 5;
                if (mode == 0) {
                    if (x == 0) dcv = y == 0 ? 0x80 : (tsum + 8) >> 4;     // "DCLeft" sums the top row
                    else if (y == 0) dcv = (lsum + 8) >> 4;                // "DCTop" sums the left column
                }
                const uint32_t top4 = tile32(T.y[0], 4 + 4 * bi);
                const uint32_t left4 = pack4(T.y[4 * bj + 1][3], T.y[4 * bj + 2][3], T.y[4 * bj + 3][3], T.y[4 * bj + 4][3]);
                uint32_t pr[4];
                pred_large(mode, top4, left4, int(T.y[0][3]), dcv, pr);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tile32(T.y[4 * bj + 1 + r], 4 + 4 * bi) = add_clip_row(pr[r], ev[r], od[r]);
                }
            }
        }
        WP_PROF_MARK(3);
        // i4: blocks in raster order predict from already reconstructed neighbours; block (bj,bi)
        // is ready at mini-step bi + 2*bj (it needs (bj,bi-1), (bj-1,bi) and (bj-1,bi+1)).
        if (__builtin_amdgcn_ballot_w64(act && !i16) != 0) {
            __builtin_amdgcn_s_setprio(2);
            const int my_t = bi + 2 * bj;
            const int mode4 = int(((l < 8 ? rec.x : rec.y) >> (4 * (l & 7))) & 15);
            for (int t = 0; t < 10; ++t) {
                if (act && !i16 && t == my_t) {
                    uint8_t* above = T.y[4 * bj];
                    const uint32_t top_lo = tile32(above, 4 + 4 * bi);
                    const uint32_t top_hi = tile32(above, 8 + 4 * bi);
                    const int corner = above[3 + 4 * bi];
                    const uint32_t left4 = pack4(T.y[4 * bj + 1][3 + 4 * bi], T.y[4 * bj + 2][3 + 4 * bi],
                                                 T.y[4 * bj + 3][3 + 4 * bi], T.y[4 * bj + 4][3 + 4 * bi]);
                    uint32_t pr[4];
                    pred4x4(mode4, top_lo, top_hi, left4, corner, pr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tile32(T.y[4 * bj + 1 + r], 4 + 4 * bi) = add_clip_row(pr[r], ev[r], od[r]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            __builtin_amdgcn_s_setprio(1);
        }
        WP_PROF_MARK(4);
        if (act) {
            // ---- chroma: one 8x8 predictor per plane, four 4x4 residual blocks each ----
            const int c8 = int((w2 >> WP_MB_C8_SHIFT) & 3);
            const int plane = l >> 3;                 // lanes 0..7 sum Cb borders, 8..15 Cr borders
            const int cb = plane ? 20 : 4;
            int tsum = int(T.c[0][cb + (l & 7)]), lsum = int(T.c[1 + (l & 7)][cb - 1]);
            tsum = group_sum8(tsum); lsum = group_sum8(lsum);
            int dcv = (tsum + lsum + 8) >> 4;
            if (c8 == 0) {
                if (x == 0) dcv = y == 0 ? 0x80 : (tsum + 4) >> 3;
                else if (y == 0) dcv = (lsum + 4) >> 3;
            }
            if (chroma_lane) {
                const int cj = (l >> 1) & 1, ci = l & 1;
                const uint32_t top4 = tile32(T.c[0], cb + 4 * ci);
                const uint32_t left4 = pack4(T.c[4 * cj + 1][cb - 1], T.c[4 * cj + 2][cb - 1], T.c[4 * cj + 3][cb - 1], T.c[4 * cj + 4][cb - 1]);
                uint32_t pr[4];
                pred_large(c8, top4, left4, int(T.c[0][cb - 1]), dcv, pr);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tile32(T.c[4 * cj + 1 + r], cb + 4 * ci) = add_clip_row(pr[r], cev[r], cod[r]);
                }
            }
        }
        // ---- tell the band below how far this band's last row has got.  The value published here is the
        //      one of the PREVIOUS step: its stores were issued a whole step ago, so the release fence
        //      (s_waitcnt vmcnt(0)) does not stall on store acknowledgements ----
        WP_PROF_MARK(5);
        __builtin_amdgcn_s_setprio(0);
        settle(rec_n2); settle(nl0); settle(nl1); settle(nwv); settle(nc0); settle(nc1); settle(top_n);   // the prefetches have had the whole step to land; wait for them before this step's stores go out
        if (has_below) {
            if (pending_publish) publish_progress(&progress[wave], pending_publish);
            const int done = s - 2 * (kBandRows - 1) + 1;
            pending_publish = done >= 1 ? ((q << kSeqShift) | uint32_t(done)) : 0u;
        }
        __builtin_amdgcn_wave_barrier();
        if (act) {
            // ---- write back (Vp8Decoder.hx:812-821): the MB sits complete in the LDS tile; lane l stores
            //      luma row l and one chroma row, so the group emits the 384-byte tile as three full lines ----
            uint8_t* tile = Rrow + uint32_t(x) * uint32_t(kTileBytes);     // 32-bit offset inside the row (mbw <= 1024)
            const uint8_t* yr = T.y[1 + l];
            st_nt(reinterpret_cast<uint4*>(tile + 16 * l), make_uint4(tile32(const_cast<uint8_t*>(yr), 4), tile32(const_cast<uint8_t*>(yr), 8),
                                                                  tile32(const_cast<uint8_t*>(yr), 12), tile32(const_cast<uint8_t*>(yr), 16)));
            uint8_t* cr = T.c[1 + (l & 7)];
            const int cc = (l >> 3) ? 20 : 4;
            st_nt(reinterpret_cast<uint2*>(tile + kTileCb + 8 * l), make_uint2(tile32(cr, cc), tile32(cr, cc + 4)));
        }
        __builtin_amdgcn_wave_barrier();
        WP_PROF_MARK(6);
        // ---- hand the bottom row to the group below (same wavefront, two steps later) ----
        if (act && g + 1 < kBandRows && l < 8) {
            const uint32_t v = l < 4 ? tile32(T.y[16], 4 + 4 * l)
                             : l < 6 ? tile32(T.c[8], 4 + 4 * (l - 4))
                                     : tile32(T.c[8], 20 + 4 * (l - 6));
            lds.ring[g + 1][x & 3][l] = v;
        }
    }
    if (has_below && pending_publish) publish_progress(&progress[wave], pending_publish);
    WP_PROF_MARK(7);
    if (mark_done) publish_progress(&progress[wave], (q << kSeqShift) | kBandDone);
    WP_PROF_FLUSH(0);
}

}  // namespace wpk
