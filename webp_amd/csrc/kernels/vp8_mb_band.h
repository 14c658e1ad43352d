// Reconstruction AND loop filter of one band (4 macroblock rows) of one frame by one wavefront, macroblock by macroblock:
// a group reconstructs MB (x, y) into its LDS workspace and filters it in the same step, so the reconstructed pixels
// never leave the CU (the two-kernel path writes them to HBM as R tiles and reads them back: 768 bytes per MB).
//
// Both halves have the same partial order -- MB (x, y) after (x-1, y), (x, y-1) and (x+1, y-1) -- so one 2:1 skewed
// schedule serves both: group g of the wavefront works on row 4*band+g at column s-2g in step s.  Reconstruction is
// vp8_recon_band.h's (prepareYBR / inverseWHT16 / inverseDCT4 / PredFunc, Vp8Decoder.hx:541-577, 716-770, 905-996),
// the loop filter vp8_filter_band.h's (simpleFilter / normalFilter, Vp8Decoder.hx:417-486; Filter.hx); intra prediction
// keeps reading UNFILTERED neighbours (Vp8Decoder.hx:561-569) because the two halves keep separate LDS state:
//   ReconTile      the reference's ybr workspace (unfiltered MB + its borders), plus a 4-slot ring with the unfiltered
//                  bottom row of the MB row above;
//   MbFilterTile   the group's own 16 luma / 8 chroma rows in two MB slots (slot = x & 1: the previous MB's slot is the
//                  left neighbour), filtered in place;
//   ring_y/ring_c  the 4 rows above the MB row (the loop filter reads 4 and rewrites 3 of them), 32 luma / 16 chroma
//                  columns addressed by absolute column: handed down by the group above, filtered in place by this one.
// Output: S tiles (wp_device.h) for EVERY frame -- with the loop filter off the shifted tiles simply hold the
// reconstruction.  Tile (x, y) is written in three pieces, each byte exactly once per producer: luma rows -8..-5 by the
// group above when it hands its bottom rows down (they are final then), rows -4..7 and chroma by the group itself.
// Between bands (different wavefronts of the workgroup, same CU): the unfiltered bottom row of a band goes through a
// 32-byte-per-MB edge buffer (the head of the workspace's R region), its 8 filtered bottom rows through the top half of
// the S tiles of the next tile row, both announced by the LDS progress word (workgroup-scope release / acquire).
#pragma once
#include "vp8_recon_band.h"
#include "vp8_filter_band.h"

namespace wpk {

constexpr int kEdgeBytes = 32;      // per MB of a band's last row: 16 Y + 8 Cb + 8 Cr bytes of its unfiltered bottom row

struct alignas(16) MbFilterTile {
    uint8_t y[16][48];          // own MB rows; cols: MB slot (x & 1) * 16; pitch 48 = 12 dwords (see FilterTile)
    uint8_t c[2][8][16];        // own chroma rows; cols: slot (x & 1) * 8
    uint8_t pad[16];            // 1040 B = 260 dwords: the four groups' tiles start 4 banks apart
};
struct alignas(16) MbGroupLds {                          // everything one group (one MB row of the band) keeps in LDS: one base address per lane
    ReconTile rt;
    uint32_t rring[4][8];                                // rring[col & 3]: unfiltered bottom row of the MB above: Y (4 dwords), U (2), V (2)
    MbFilterTile ft;
    alignas(16) uint8_t ring_y[4][32];                   // ring_y[r][abs_col & 31]: luma rows -4..-1 above the group's MB row
    uint8_t ring_c[2][4][16];                            // the same for Cb / Cr (abs_col & 15)
    uint8_t pad[16];                                     // 2400 B = 600 dwords: groups start 24 banks apart
};
struct alignas(16) MbWaveLds {
    MbGroupLds grp[kBandRows];
};

template <int W>
__device__ __forceinline__ void mb_band(const wp_vp8_frame& fr, int band, int n_bands, uint32_t q, uint32_t q_prev, int wave, uint32_t* progress,
                                        MbWaveLds& lds, const uint4* __restrict__ mb, const int16_t* __restrict__ coeff,
                                        const uint32_t* __restrict__ filt, uint8_t* Ep, uint8_t* Sp) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4;            // macroblock row within the band
    const int l = lane & 15;            // reconstruction: 4x4 luma block owned by this lane; loop filter: pixel row / column
    const int bj = l >> 2, bi = l & 3;
    const int lp = l >> 3, lr = l & 7;                    // chroma plane / chroma row (or column)
    const int hcr = (l >> 1) & 3, hh = l & 1;             // chroma hand-over: row 0..3, 4-byte half
    const int qr = l >> 2, qd = l & 3;                    // luma hand-over: row 0..3, 4-byte quarter of the 16 columns
    MbGroupLds& G = lds.grp[g];
    MbGroupLds& GN = lds.grp[(g + 1) & 3];                  // the group below (never written by the band's last group)
    ReconTile& T = G.rt;
    MbFilterTile& F = G.ft;
    const int mbw = fr.mbw, mbh = fr.mbh;
    const int ftype = fr.filter_type;                     // wave-uniform: 0 off, 1 simple, 2 normal
    uint8_t* const E = Ep + size_t(fr.mb_base) * kEdgeBytes;
    uint8_t* const S = Sp + size_t(fr.stile_base) * kTileBytes;
    const int spitch = mbw + 1;
    const int y = band * kBandRows + g;
    const bool row_ok = y < mbh;
    const bool has_above = band > 0;
    const bool has_below = band + 1 < n_bands;
    const int prev_wave = int(q_prev % uint32_t(W));        // the wavefront that runs the band above
    const int rows_here = min(kBandRows, mbh - band * kBandRows);
    const int n_steps = mbw + 2 * (rows_here - 1);
    const bool last_row = g == rows_here - 1;               // its bottom rows leave the band through memory

    // Global addresses are a wave-uniform base (scalar registers) plus a 32-bit per-lane byte offset: a frame's records,
    // coefficient blocks (<= 25 x 32 B per MB) and tiles all stay below 4 GB for the largest frame (1024 x 1024 MBs).
    const int yy = row_ok ? y : 0;
    const uint4* const mbf = mb + size_t(fr.mb_base);
    const uint32_t* const ff = filt + size_t(fr.mb_base);
    const uint32_t mrow = uint32_t(yy) * uint32_t(mbw);
    const char* const cfr = reinterpret_cast<const char*>(coeff + size_t(fr.coeff_base) * 16);
    const uint32_t srow = uint32_t(yy) * uint32_t(spitch) * uint32_t(kTileBytes);      // S tiles of tile row y
    const uint32_t erow = uint32_t(band) * uint32_t(mbw) * uint32_t(kEdgeBytes);       // edge rows this band leaves behind
    // ---- what a step consumes from global memory is fetched one step ahead (registers) ----
    uint4 rec_n1 = make_uint4(0, 0, 0, 0), rec_n2 = make_uint4(0, 0, 0, 0);     // records of MB x+1, x+2
    uint4 nl0 = make_uint4(0, 0, 0, 0), nl1 = make_uint4(0, 0, 0, 0);           // luma block l of MB x+1
    uint32_t nwv = 0;                                                           // WHT coefficient l of MB x+1
    uint4 nc0 = make_uint4(0, 0, 0, 0), nc1 = make_uint4(0, 0, 0, 0);           // chroma block of MB x+1 (chroma lanes)
    uint32_t nfp = 0, nfp2 = 0;                                                 // loop-filter records of MB x+1, x+2
    const int cn = 16 + (l & 3) + ((l & 8) ? 4 : 0);                            // lanes 0-3: Cb blocks, 8-11: Cr blocks
    const bool chroma_lane = (l & 4) == 0;
    auto prefetch = [&](int s_) {
        const int x_ = s_ - 2 * g;
        rec_n1 = rec_n2;
        nfp = nfp2;
        // a lane whose MB x+2 is outside the frame reads the frame's first record instead: whatever it then fetches is
        // ignored (everything below is guarded by `act`), and the addresses stay inside the frame's buffers
        const bool in2 = row_ok && x_ + 2 >= 0 && x_ + 2 < mbw;
        const uint32_t idx = in2 ? mrow + uint32_t(x_ + 2) : 0u;
        rec_n2 = mbf[idx];
        nfp2 = ff[idx];
        const uint32_t w2n = rec_n1.z;
        const uint32_t whtn = (w2n >> WP_MB_WHT_BIT) & 1u;
        const uint32_t blk0 = rec_n1.w + whtn;                                  // first non-WHT block of MB x+1
        nl0 = make_uint4(0, 0, 0, 0); nl1 = nl0; nc0 = nl0; nc1 = nl0; nwv = 0;
        if ((w2n >> l) & 1) {
            const uint4* p = reinterpret_cast<const uint4*>(cfr + size_t((blk0 + uint32_t(__popc(w2n & ((1u << l) - 1u)))) * 32u));
            nl0 = ld_nt(p); nl1 = ld_nt(p + 1);
        }
        if (whtn) nwv = uint32_t(int(*reinterpret_cast<const int16_t*>(cfr + size_t(rec_n1.w * 32u + 2u * uint32_t(l)))));
        if (chroma_lane && ((w2n >> cn) & 1)) {
            const uint4* p = reinterpret_cast<const uint4*>(cfr + size_t((blk0 + uint32_t(__popc(w2n & ((1u << cn) - 1u)))) * 32u));
            nc0 = ld_nt(p); nc1 = ld_nt(p + 1);
        }
    };
    prefetch(-2);
    prefetch(-1);
    // Band boundary (group 0 only): per column the unfiltered bottom row of the band above (lanes 0-7: one dword each of
    // its edge record) and the 4 filtered rows above (luma: lane = (row, 4-column quarter), chroma: lane = (plane, row,
    // half); columns 0..7 of MB x come from S tile x, columns 8..15 from S tile x+1).
    const bool top_lane = has_above && g == 0;
    const uint32_t top_src = uint32_t(has_above ? band - 1 : 0) * uint32_t(mbw) * uint32_t(kEdgeBytes) + 4u * uint32_t(l & 7);
    const uint32_t trow_y = srow + uint32_t(qd >> 1) * uint32_t(kTileBytes) + uint32_t(16 * (4 + qr) + (qd < 2 ? 8 + 4 * qd : 4 * (qd - 2)));
    const uint32_t trow_c = srow + uint32_t(hh) * uint32_t(kTileBytes) + uint32_t(kTileCb + 64 * lp + 8 * hcr + 4 * (1 - hh));
    uint32_t top_n = 0, nx_ty = 0, nx_tc = 0;
    if (has_above) {
        wait_progress(&progress[prev_wave], (q_prev << kSeqShift) | uint32_t(min(2, mbw)));
        if (top_lane) {
            if (l < 8) {
                lds.grp[0].rring[0][l] = *reinterpret_cast<const uint32_t*>(E + size_t(top_src));
                if (mbw > 1) top_n = *reinterpret_cast<const uint32_t*>(E + size_t(top_src + uint32_t(kEdgeBytes)));
            }
            nx_ty = *reinterpret_cast<const uint32_t*>(S + size_t(trow_y));
            nx_tc = *reinterpret_cast<const uint32_t*>(S + size_t(trow_c));
        }
    }
    settle(rec_n1); settle(rec_n2); settle(nl0); settle(nl1); settle(nwv); settle(nc0); settle(nc1); settle(top_n);
    settle(nfp); settle(nfp2); settle(nx_ty); settle(nx_tc);
    uint32_t pending_publish = 0;

    for (int s = 0; s < n_steps; ++s) {
        const int x = s - 2 * g;
        const bool act = row_ok && x >= 0 && x < mbw;
        const uint4 rec = act ? rec_n1 : make_uint4(0, 0, 0, 0);
        const uint32_t fp = (act && ftype != 0) ? nfp : 0u;
        const uint32_t cur_ty = nx_ty, cur_tc = nx_tc;

        // ---- band boundary: park the column fetched during the previous step, fetch the next ----
        if (has_above && s + 1 < mbw) {
            if (top_lane && l < 8) lds.grp[0].rring[(s + 1) & 3][l] = top_n;
            wait_progress(&progress[prev_wave], (q_prev << kSeqShift) | uint32_t(min(s + 3, mbw)));
            if (top_lane) {
                const uint32_t o = uint32_t(s + 1) * uint32_t(kTileBytes);
                if (l < 8 && s + 2 < mbw) top_n = *reinterpret_cast<const uint32_t*>(E + size_t(top_src + uint32_t(s + 2) * uint32_t(kEdgeBytes)));
                nx_ty = *reinterpret_cast<const uint32_t*>(S + size_t(trow_y + o));
                nx_tc = *reinterpret_cast<const uint32_t*>(S + size_t(trow_c + o));
            }
        }

        // ---- residuals (independent of the neighbours) ----
        const uint32_t w2 = rec.z;
        const bool i16 = (w2 >> WP_MB_I16_BIT) & 1;
        const int has_wht = (w2 >> WP_MB_WHT_BIT) & 1;
        uint32_t ev[4] = {0, 0, 0, 0}, od[4] = {0, 0, 0, 0};          // luma residual of block l
        {
            int c[16];
            unpack_block(nl0, nl1, c);
            const bool present = act && ((w2 >> l) & 1);
            int dc16 = 0;
            if (__builtin_amdgcn_ballot_w64(act && has_wht) != 0) {
                const int wv = (act && has_wht) ? int(nwv) : 0;
                dc16 = wht_dc_lanes(wv, l);
                if (act && has_wht) c[0] = dc16; else dc16 = 0;
            }
            if (__builtin_amdgcn_ballot_w64(present || dc16 != 0) != 0) idct_residual(c, ev, od);
        }
        uint32_t cev[4] = {0, 0, 0, 0}, cod[4] = {0, 0, 0, 0};
        {
            const bool cpres = act && chroma_lane && ((w2 >> cn) & 1);
            const bool c_ac = cpres && (((nc0.x >> 16) | nc0.y | nc0.z | nc0.w | nc1.x | nc1.y | nc1.z | nc1.w) != 0);
            if (__builtin_amdgcn_ballot_w64(c_ac) != 0) {
                int c[16];
                unpack_block(nc0, nc1, c);
                idct_residual(c, cev, cod);
            } else {
                const int v = (int(int16_t(nc0.x & 0xffff)) + 4) >> 3;
                const uint32_t p = as_u32(__builtin_amdgcn_cvt_pk_i16(v, v));
#pragma unroll
                for (int r = 0; r < 4; ++r) { cev[r] = p; cod[r] = p; }
            }
        }
        // The coefficients of this step are consumed: fetch the next step's into the same registers.  They have the
        // prediction and the whole loop filter of this step to land.
        prefetch(s);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(1);

        if (act) {
            // ---- prepareYBR (Vp8Decoder.hx:541-577); see vp8_recon_band.h ----
            {
                const int lc = min(l, 8);
                uint32_t va = T.y[l][19], vb = T.y[16][19], vc1 = T.c[lc][11], vc2 = T.c[lc][27];     // last column of the previous MB
                const bool tr_job = l == 4 || (l >= 10 && l <= 12);
                const uint32_t* slot = G.rring[x & 3];
                uint32_t rv = slot[l < 4 ? l : (l >= 6 && l <= 9) ? l - 2 : 3];
                const uint32_t nxt = G.rring[(x + 1) & 3][0];
                if (tr_job) rv = x == mbw - 1 ? splat4(int(rv >> 24)) : nxt;
                if (x == 0) { va = 0x81; vb = 0x81; vc1 = 0x81; vc2 = 0x81; }
                if (y == 0) {
                    rv = 0x7f7f7f7fu;
                    if (l == 0) { va = 0x7f; vc1 = 0x7f; vc2 = 0x7f; }
                }
                T.y[l][3] = uint8_t(va);
                if (l == 0) T.y[16][3] = uint8_t(vb);
                if (l < 9) { T.c[l][3] = uint8_t(vc1); T.c[l][19] = uint8_t(vc2); }
                uint8_t* dst = l < 4 ? &T.y[0][4 + 4 * l] : l == 4 ? &T.y[0][20] : l < 8 ? &T.c[0][4 + 4 * (l - 6)]
                             : l < 10 ? &T.c[0][20 + 4 * (l - 8)] : &T.y[4 * (l - 9)][20];
                if (l < 5 || (l >= 6 && l <= 12)) tile32(dst, 0) = rv;
            }
            __builtin_amdgcn_wave_barrier();

            // ---- luma, 16x16 ----
            if (i16) {
                int mode = int(rec.x & 15);
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
                for (int r = 0; r < 4; ++r) tile32(T.y[4 * bj + 1 + r], 4 + 4 * bi) = add_clip_row(pr[r], ev[r], od[r]);
            }
        }
        // ---- luma, 4x4: block (bj, bi) is ready at mini-step bi + 2*bj ----
#ifndef X_NOI4
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
                    for (int r = 0; r < 4; ++r) tile32(T.y[4 * bj + 1 + r], 4 + 4 * bi) = add_clip_row(pr[r], ev[r], od[r]);
                }
                __builtin_amdgcn_wave_barrier();
            }
            __builtin_amdgcn_s_setprio(1);
        }
#endif
        if (act) {
            // ---- chroma: one 8x8 predictor per plane, four 4x4 residual blocks each ----
            const int c8 = int((w2 >> WP_MB_C8_SHIFT) & 3);
            const int ccol = lp ? 20 : 4;
            int tsum = int(T.c[0][ccol + lr]), lsum = int(T.c[1 + lr][ccol - 1]);
            tsum = group_sum8(tsum); lsum = group_sum8(lsum);
            int dcv = (tsum + lsum + 8) >> 4;
            if (c8 == 0) {
                if (x == 0) dcv = y == 0 ? 0x80 : (tsum + 4) >> 3;
                else if (y == 0) dcv = (lsum + 4) >> 3;
            }
            if (chroma_lane) {
                const int cj = (l >> 1) & 1, ci = l & 1;
                const uint32_t top4 = tile32(T.c[0], ccol + 4 * ci);
                const uint32_t left4 = pack4(T.c[4 * cj + 1][ccol - 1], T.c[4 * cj + 2][ccol - 1], T.c[4 * cj + 3][ccol - 1], T.c[4 * cj + 4][ccol - 1]);
                uint32_t pr[4];
                pred_large(c8, top4, left4, int(T.c[0][ccol - 1]), dcv, pr);
#pragma unroll
                for (int r = 0; r < 4; ++r) tile32(T.c[4 * cj + 1 + r], ccol + 4 * ci) = add_clip_row(pr[r], cev[r], cod[r]);
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ================= the macroblock is reconstructed (unfiltered) in T =================
        const int level = int(fp & 0xff), ilevel = int((fp >> 8) & 0xff), hlevel = int((fp >> 16) & 0xff);
        const bool inner = (fp >> 24) & 1;
        const bool do_filter = act && level != 0;                   // `if (f.level == 0) continue` (:421, :449)
        const int cb = 16 * (x & 1), pb = 16 - cb;                  // luma column base of this MB's slot / the previous MB's
        const int cb8 = cb >> 1, pb8 = pb >> 1;
        const int rc0 = (16 * x) & 31, rc8 = (8 * x) & 15;          // this MB's first column in the rings
        uint8_t (*const RY)[32] = G.ring_y;
        uint8_t (*const RC)[16] = G.ring_c[lp];
        if (act) {
            // ---- its unfiltered bottom row goes to the group below (Vp8Decoder.hx:561-569 reads Y/Cb/Cr before the
            //      loop filter has run) or, from the band's last row, to the edge buffer ----
            if (l < 8) {
                const uint32_t v = l < 4 ? tile32(T.y[16], 4 + 4 * l) : l < 6 ? tile32(T.c[8], 4 + 4 * (l - 4)) : tile32(T.c[8], 20 + 4 * (l - 6));
                if (g + 1 < kBandRows) GN.rring[x & 3][l] = v;
                if (last_row && has_below) *reinterpret_cast<uint32_t*>(E + size_t(erow + uint32_t(x) * uint32_t(kEdgeBytes) + 4u * uint32_t(l))) = v;
            }
            // ---- copy into the loop filter's tile (lane = pixel row), and the 4 rows above at a band boundary ----
            {
                uint8_t* yr = T.y[1 + l];
                t128(&F.y[l][cb]) = make_uint4(tile32(yr, 4), tile32(yr, 8), tile32(yr, 12), tile32(yr, 16));
                uint8_t* cr = T.c[1 + lr];
                const int cc = lp ? 20 : 4;
                t64(&F.c[lp][lr][cb8]) = make_uint2(tile32(cr, cc), tile32(cr, cc + 4));
            }
            if (g == 0 && y > 0) {
                t32(&RY[qr][(rc0 + 4 * qd) & 31]) = cur_ty;
                t32(&RC[hcr][(rc8 + 4 * hh) & 15]) = cur_tc;
            }
        }
        __builtin_amdgcn_wave_barrier();

#ifndef X_NOFILTER
        if (ftype != 0) {
            if (do_filter) {
                // ---- vertical edges: lane = pixel row; columns -4..15 relative to the MB ----
                {
                    int v[20];
                    unpack4(t32(&F.y[l][pb + 12]), v);
                    const uint4 w = t128(&F.y[l][cb]);
                    unpack4(w.x, v + 4); unpack4(w.y, v + 8); unpack4(w.z, v + 12); unpack4(w.w, v + 16);
                    if (ftype == 1) filter_line<20, true>(v, x > 0, inner, level, ilevel, hlevel);
                    else filter_line<20, false>(v, x > 0, inner, level, ilevel, hlevel);
                    t32(&F.y[l][pb + 12]) = pack4i(v);
                    t128(&F.y[l][cb]) = make_uint4(pack4i(v + 4), pack4i(v + 8), pack4i(v + 12), pack4i(v + 16));
                }
                if (ftype == 2) {
                    int v[12];
                    unpack4(t32(&F.c[lp][lr][pb8 + 4]), v);
                    const uint2 w = t64(&F.c[lp][lr][cb8]);
                    unpack4(w.x, v + 4); unpack4(w.y, v + 8);
                    filter_line<12, false>(v, x > 0, inner, level, ilevel, hlevel);
                    t32(&F.c[lp][lr][pb8 + 4]) = pack4i(v);
                    t64(&F.c[lp][lr][cb8]) = make_uint2(pack4i(v + 4), pack4i(v + 8));
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_setprio(2);
            if (do_filter) {
                // ---- horizontal edges: lane = pixel column; rows -4..-1 live in the ring, 0..15 in the tile ----
                {
                    int v[20];
                    const int rc = (rc0 + l) & 31;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = RY[k][rc];
#pragma unroll
                    for (int k = 4; k < 20; ++k) v[k] = F.y[k - 4][cb + l];
                    if (ftype == 1) filter_line<20, true>(v, y > 0, inner, level, ilevel, hlevel);
                    else filter_line<20, false>(v, y > 0, inner, level, ilevel, hlevel);
#pragma unroll
                    for (int k = 1; k < 4; ++k) RY[k][rc] = uint8_t(v[k]);
#pragma unroll
                    for (int k = 4; k < 19; ++k) F.y[k - 4][cb + l] = uint8_t(v[k]);
                }
                if (ftype == 2) {
                    int v[12];
                    const int rc = (rc8 + lr) & 15;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = RC[k][rc];
#pragma unroll
                    for (int k = 4; k < 12; ++k) v[k] = F.c[lp][k - 4][cb8 + lr];
                    filter_line<12, false>(v, y > 0, inner, level, ilevel, hlevel);
#pragma unroll
                    for (int k = 1; k < 4; ++k) RC[k][rc] = uint8_t(v[k]);
#pragma unroll
                    for (int k = 4; k < 11; ++k) F.c[lp][k - 4][cb8 + lr] = uint8_t(v[k]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
#endif

        __builtin_amdgcn_s_setprio(0);
        // ---- tell the band below how far this band's last row has got: the value of the PREVIOUS step, whose stores
        //      were issued a whole step ago (the release fence then does not wait for their acknowledgements) ----
        settle(rec_n2); settle(nl0); settle(nl1); settle(nwv); settle(nc0); settle(nc1); settle(top_n);
        settle(nfp2); settle(nx_ty); settle(nx_tc);      // the prefetches landed long ago; wait before the stores go out
        if (has_below) {
            if (pending_publish) publish_progress(&progress[wave], pending_publish);
            const int done = s - 2 * (kBandRows - 1) + 1;
            pending_publish = done >= 1 ? ((q << kSeqShift) | uint32_t(done)) : 0u;
        }

        if (act) {
            // ---- S tile (x, y): luma rows -4..7 (lanes 4..15; rows -8..-5 came from the group above), chroma rows -4..3;
            //      cols -8..7 (chroma -4..3) relative to the MB: all final ----
            uint8_t* st = S + size_t(srow + uint32_t(x) * uint32_t(kTileBytes));
            if (l >= 4) {
                const uint8_t* pa = l < 8 ? &RY[l - 4][(rc0 - 8) & 31] : &F.y[l - 8][pb + 8];
                const uint8_t* pq = l < 8 ? &RY[l - 4][rc0] : &F.y[l - 8][cb];
                const uint2 a = *reinterpret_cast<const uint2*>(pa), b = *reinterpret_cast<const uint2*>(pq);
                st_nt(reinterpret_cast<uint4*>(st + 16 * l), make_uint4(a.x, a.y, b.x, b.y));
                if (x == mbw - 1) {        // right-most tile column: cols 8..15 (left half of tile mbw)
                    const uint8_t* pe = l < 8 ? &RY[l - 4][(rc0 + 8) & 31] : &F.y[l - 8][cb + 8];
                    *reinterpret_cast<uint2*>(st + kTileBytes + 16 * l) = *reinterpret_cast<const uint2*>(pe);
                }
            }
            {
                const uint8_t* pa = lr < 4 ? &RC[lr][(rc8 - 4) & 15] : &F.c[lp][lr - 4][pb8 + 4];
                const uint8_t* pq = lr < 4 ? &RC[lr][rc8] : &F.c[lp][lr - 4][cb8];
                st_nt(reinterpret_cast<uint2*>(st + kTileCb + 8 * l), make_uint2(*reinterpret_cast<const uint32_t*>(pa), *reinterpret_cast<const uint32_t*>(pq)));
                if (x == mbw - 1) {
                    const uint8_t* pe = lr < 4 ? &RC[lr][(rc8 + 4) & 15] : &F.c[lp][lr - 4][cb8 + 4];
                    *reinterpret_cast<uint32_t*>(st + kTileBytes + kTileCb + 8 * l) = *reinterpret_cast<const uint32_t*>(pe);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();      // the ring reads above come before the ring writes below (same columns mod 32 / 16)
        if (act) {
            // ---- hand the eight bottom luma rows (cols -8..7; state after this MB) down: rows 8..11 are final and go
            //      to S tile (x, y+1) rows -8..-5; rows 12..15 to the ring of the group below, or -- from the band's last
            //      row -- to the same S tile, where the next band (or the finishing pass) picks them up.  Chroma: 4 rows. ----
            uint8_t* sn = S + size_t(srow + (uint32_t(x) + uint32_t(spitch)) * uint32_t(kTileBytes));       // S tile (x, y+1); tile row mbh exists
            const uint8_t* fsrc = qd < 2 ? &F.y[8 + qr][pb + 8 + 4 * qd] : &F.y[8 + qr][cb + 4 * (qd - 2)];
            const uint32_t v_fin = *reinterpret_cast<const uint32_t*>(fsrc);                     // rows 8..11
            const uint32_t v_low = *reinterpret_cast<const uint32_t*>(fsrc + 4 * 48);            // rows 12..15
            const uint32_t vc = t32(&F.c[lp][4 + hcr][hh ? cb8 : pb8 + 4]);
            *reinterpret_cast<uint32_t*>(sn + 16 * qr + 4 * qd) = v_fin;
            if (!last_row) {
                uint8_t (*const NY)[32] = GN.ring_y;
                uint8_t (*const NC)[16] = GN.ring_c[lp];
                t32(&NY[qr][(rc0 - 8 + 4 * qd) & 31]) = v_low;
                t32(&NC[hcr][(rc8 - 4 + 4 * hh) & 15]) = vc;
            } else {
                *reinterpret_cast<uint32_t*>(sn + 64 + 16 * qr + 4 * qd) = v_low;
                *reinterpret_cast<uint32_t*>(sn + kTileCb + 64 * lp + 8 * hcr + 4 * hh) = vc;
                if (x == mbw - 1) {
                    if (qd < 2) *reinterpret_cast<uint32_t*>(sn + kTileBytes + 64 + 16 * qr + 4 * qd) = t32(&F.y[12 + qr][cb + 8 + 4 * qd]);
                    if (hh) *reinterpret_cast<uint32_t*>(sn + kTileBytes + kTileCb + 64 * lp + 8 * hcr) = t32(&F.c[lp][4 + hcr][cb8 + 4]);
                }
            }
            if (x == mbw - 1 && qd < 2) *reinterpret_cast<uint32_t*>(sn + kTileBytes + 16 * qr + 4 * qd) = t32(&F.y[8 + qr][cb + 8 + 4 * qd]);
        } else if (row_ok && x == mbw && !last_row) {
            // cols 8..15 of the row's last MB reach the ring one step late: a step earlier they would land on columns
            // (mod 32 / mod 16) the group below still reads.  Relative to the virtual MB `mbw` they are its cols -8..-1.
            if (qd < 2) t32(&GN.ring_y[qr][(rc0 - 8 + 4 * qd) & 31]) = t32(&F.y[12 + qr][pb + 8 + 4 * qd]);
            if (!hh) t32(&GN.ring_c[lp][hcr][(rc8 - 4) & 15]) = t32(&F.c[lp][4 + hcr][pb8 + 4]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (has_below && pending_publish) publish_progress(&progress[wave], pending_publish);
}

}  // namespace wpk
