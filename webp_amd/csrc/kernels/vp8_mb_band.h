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
//                  bottom row of the MB row above (kept in the pad columns of MbFilterTile's luma rows);
//   MbFilterTile   4 rows above + the group's own 16 luma rows (chroma: 4 + 8) in two MB slots, slot = x & 1: the
//                  previous MB's slot is the left neighbour.  Columns are absolute columns mod 32 (chroma: mod 16), so
//                  the group above hands its 4 bottom rows down by writing them straight into rows 0..3 of this tile;
//                  everything is filtered in place.
// Output: S tiles (wp_device.h) for EVERY frame -- with the loop filter off the shifted tiles simply hold the
// reconstruction.  Tile (x, y) is written in pieces, each byte once per producer: luma rows -8..-5 by the group above
// when it hands its bottom rows down (they are final then), rows -4..7 and chroma by the group itself.
// Between bands (different wavefronts of the workgroup, same CU): the unfiltered bottom row of a band goes through a
// 32-byte-per-MB edge buffer (the head of the workspace's R region), its 8 filtered bottom rows through the top half of
// the S tiles of the next tile row, both announced by the LDS progress word (workgroup-scope release / acquire).
// Inputs are read through a buffer resource: a lane with nothing to fetch (outside the frame, no stored block) uses an
// out-of-range offset and gets zeros back, without a branch and without memory traffic.
#pragma once
#include "vp8_recon_band.h"
#include "vp8_filter_band.h"

namespace wpk {

// Issue priorities along a step (a wavefront close to publishing goes first): residuals + emit / hand-down, prediction +
// vertical edges, i4 mini-steps, horizontal edges.  The finishing pass that shares the GPU with this kernel runs at 1.
#ifndef WP_MB_PRIO_R
#define WP_MB_PRIO_R 0
#endif
#ifndef WP_MB_PRIO_P
#define WP_MB_PRIO_P 1
#endif
#ifndef WP_MB_PRIO_I4
#define WP_MB_PRIO_I4 2
#endif
#ifndef WP_MB_PRIO_H
#define WP_MB_PRIO_H 2
#endif

#ifndef WP_COEFF_NT
#define WP_COEFF_NT 0
#endif
// Coefficient blocks WITHOUT the non-temporal hint: a block is used once, but the 64-byte sector it shares with the next
// macroblock's first block is wanted again two steps later, and with the hint that second half came from HBM again
// (PMC: FETCH_SIZE 5.03 -> 4.35 GB per 1024 x 1080p, the step 0.4 % shorter).
constexpr bool kCoeffNt = WP_COEFF_NT != 0;
#ifndef WP_FILTER_F32
#define WP_FILTER_F32 2      // 0: integer edges, 1: fp32 on 2^23 + x, 2: fp32 on denormals (vp8_filter_band.h); 5.45 / 5.15 / 4.99 ms per 1024 x 1080p alone
#endif
constexpr bool kFilterDen = WP_FILTER_F32 == 2;      // ... with the samples as fp32 denormals (plain integers as bit patterns) instead of 2^23 + x
constexpr bool kFilterF32 = WP_FILTER_F32 != 0;      // the normal loop filter's mask and filter value in fp32 (vp8_filter_band.h: normal_edge_f)
constexpr int kEdgeBytes = 32;      // per MB of a band's last row: 16 Y + 8 Cb + 8 Cr bytes of its unfiltered bottom row

struct alignas(16) MbFilterTile {
    uint8_t y[20][48];          // rows 0..3: the 4 rows above, 4..19: own MB rows; cols: abs_col & 31 = MB slot (x & 1) * 16 + col; pitch 48 = 12 dwords (see FilterTile)
    uint8_t c[2][12][16];       // rows 0..3 above, 4..11 own; cols: abs_col & 15 = slot (x & 1) * 8 + col
};
struct alignas(16) MbGroupLds {                          // everything one group (one MB row of the band) keeps in LDS: one base address per lane
    ReconTile rt;
    MbFilterTile ft;
    // The ring with the unfiltered bottom row of the MB row above (slot = column & 3; Y 4 dwords, U 2, V 2) lives in the
    // 16 pad bytes of the filter tile's luma rows (columns 32..47, which only exist for the bank layout): slot k = rows 2k
    // and 2k + 1.  That is 128 bytes per group the workgroup does not have to own: 73.7 KB instead of 77.8, so that two
    // workgroups leave a CU 16 KB of LDS for the finishing pass beside them.
    __device__ __forceinline__ uint32_t& rring(int slot, int i) { return *reinterpret_cast<uint32_t*>(&ft.y[2 * slot + (i >> 2)][32 + 4 * (i & 3)]); }
};                                                       // 960 + 1344 = 2304 B = 576 dwords: 9216 B per wavefront, 73.7 KB per workgroup
struct alignas(16) MbWaveLds {
    MbGroupLds grp[kBandRows];
};

// The lane's roles are re-derived from an opaque copy of the lane id at the start of the two halves of a step: derived
// values (group, block, row / column roles, LDS addresses, lane masks) would otherwise be hoisted out of the step loop
// and each hold a register -- vector or, for the masks, a scalar pair -- for the whole band (the first version kept ~40
// such values and spilt 62 vector and 39 scalar registers).  Recomputing them costs a few cheap instructions per step.
__device__ __forceinline__ int opaque_lane() { int v = int(threadIdx.x & 63); asm volatile("" : "+v"(v)); return v; }

template <int W>
__device__ __forceinline__ void mb_band(const wp_vp8_frame& fr, int band, int n_bands, uint32_t q, uint32_t q_prev, int wave, uint32_t* progress,
                                        MbWaveLds& lds, const uint4* __restrict__ mb, const int16_t* __restrict__ coeff,
                                        const uint32_t* __restrict__ filt, uint8_t* Ep, uint8_t* Sp) {
    const int mbw = fr.mbw, mbh = fr.mbh;
    const int ftype = fr.filter_type;                     // wave-uniform: 0 off, 1 simple, 2 normal
    uint8_t* const E = Ep + size_t(fr.mb_base) * kEdgeBytes;
    uint8_t* const S = Sp + size_t(fr.stile_base) * kTileBytes;
    const int spitch = mbw + 1;
    const bool has_above = band > 0;
    const bool has_below = band + 1 < n_bands;
    const int prev_wave = int(q_prev % uint32_t(W));        // the wavefront that runs the band above
    const int rows_here = min(kBandRows, mbh - band * kBandRows);
    const int n_steps = mbw + 2 * (rows_here - 1);
    // Inputs: wave-uniform buffer resources (scalar registers) + 32-bit per-lane byte offsets; a frame's records and
    // coefficient blocks (<= 25 x 32 B per MB) stay below 2 GB for the largest frame (1024 x 1024 MBs).
    // The record arrays are as long as the frame has macroblocks; the coefficient window is the frame's own blocks when the
    // descriptor says how many there are (WP_FRAME_COEFF_BOUNDED): a malformed block index then reads zeros, not foreign memory.
    const __amdgpu_buffer_rsrc_t mb_rs = buffer_of(mb + size_t(fr.mb_base), uint32_t(mbw) * uint32_t(mbh) * 16u);
    const __amdgpu_buffer_rsrc_t fl_rs = buffer_of(filt + size_t(fr.mb_base), uint32_t(mbw) * uint32_t(mbh) * 4u);
    const __amdgpu_buffer_rsrc_t co_rs = buffer_of(coeff + size_t(fr.coeff_base) * 16, coeff_window_bytes(fr));
    const uint32_t erow = uint32_t(band) * uint32_t(mbw) * uint32_t(kEdgeBytes);                       // edge rows this band leaves behind
    const uint32_t etop = uint32_t(has_above ? band - 1 : 0) * uint32_t(mbw) * uint32_t(kEdgeBytes);   // ... and those of the band above

    // ---- what a step consumes from global memory is fetched one step ahead (registers) ----
    uint4 rec_n1 = make_uint4(0, 0, 0, 0), rec_n2 = make_uint4(0, 0, 0, 0);     // records of MB x+1, x+2 (zeros outside the frame)
    uint4 nl0 = make_uint4(0, 0, 0, 0), nl1 = make_uint4(0, 0, 0, 0);           // luma block l of MB x+1
    uint32_t nwv = 0;                                                           // WHT coefficient l of MB x+1
    uint4 nc0 = make_uint4(0, 0, 0, 0), nc1 = make_uint4(0, 0, 0, 0);           // chroma block of MB x+1 (chroma lanes)
    uint32_t nfp = 0, nfp2 = 0;                                                 // loop-filter records of MB x+1, x+2
    uint32_t top_n = 0, nx_ty = 0, nx_tc = 0;                                   // band boundary: next column's rows above
    auto prefetch = [&](int s_, int ln) {
        const int g = ln >> 4, l = ln & 15;
        const int y = band * kBandRows + g;
        const int x2 = s_ - 2 * g + 2;
        const int cn = 16 + (l & 3) + ((l & 8) ? 4 : 0);                        // lanes 0-3: Cb blocks, 8-11: Cr blocks
        rec_n1 = rec_n2;
        nfp = nfp2;
        const bool in2 = y < mbh && x2 >= 0 && x2 < mbw;
        const uint32_t idx = uint32_t(y) * uint32_t(mbw) + uint32_t(x2);
        rec_n2 = buf_load16(mb_rs, in2 ? idx * 16u : kOob, false);
        nfp2 = uint32_t(__builtin_amdgcn_raw_buffer_load_b32(fl_rs, in2 ? idx * 4u : kOob, 0, 0));
        const uint32_t w2n = rec_n1.z;                                          // 0 outside the frame: nothing is fetched
        const uint32_t whtn = (w2n >> WP_MB_WHT_BIT) & 1u;
        const uint32_t blk0 = rec_n1.w + whtn;                                  // first non-WHT block of MB x+1
        const uint32_t lo = ((w2n >> l) & 1) ? (blk0 + uint32_t(__popc(w2n & ((1u << l) - 1u)))) * 32u : kOob;
        nl0 = buf_load16(co_rs, lo, kCoeffNt); nl1 = buf_load16(co_rs, lo + 16u, kCoeffNt);
        nwv = uint32_t(int(int16_t(__builtin_amdgcn_raw_buffer_load_b16(co_rs, whtn ? rec_n1.w * 32u + 2u * uint32_t(l) : kOob, 0, 0))));   // the builtin's result is unsigned
        const uint32_t cofs = ((l & 4) == 0 && ((w2n >> cn) & 1)) ? (blk0 + uint32_t(__popc(w2n & ((1u << cn) - 1u)))) * 32u : kOob;
#ifdef WP_KO_CHROMA_AC      // timing-only build: chroma DC only (what a register budget without the chroma AC prefetch would buy)
        nc0.x = uint32_t(__builtin_amdgcn_raw_buffer_load_b16(co_rs, cofs, 0, 0));
#else
        nc0 = buf_load16(co_rs, cofs, kCoeffNt); nc1 = buf_load16(co_rs, cofs + 16u, kCoeffNt);
#endif
    };
    // Band boundary (group 0 only), per column: the unfiltered bottom row of the band above (lanes 0-7: one dword each of
    // its edge record) and the 4 filtered rows above (luma: lane = (row, 4-column quarter), chroma: lane = (plane, row,
    // half); columns 0..7 of MB x come from S tile x, columns 8..15 from S tile x+1).  col: the MB column to fetch.
    // (the lane's share of the three addresses is computed once per band and kept in registers: since the fp32 edges the
    // kernel has ~20 registers to spare under its cap, and re-deriving them cost 25 instructions per step)
    uint32_t fa_e, fa_y, fa_c;
    {
        const int l = int(threadIdx.x) & 15;
        const int lp = l >> 3, hcr = (l >> 1) & 3, hh = l & 1, qr = l >> 2, qd = l & 3;
        const uint32_t srow0 = uint32_t(band * kBandRows) * uint32_t(spitch) * uint32_t(kTileBytes);
        fa_e = etop + uint32_t(kEdgeBytes) + 4u * uint32_t(l);
        fa_y = srow0 + uint32_t(qd >> 1) * uint32_t(kTileBytes) + uint32_t(16 * (4 + qr) + (qd < 2 ? 8 + 4 * qd : 4 * (qd - 2)));
        fa_c = srow0 + uint32_t(hh) * uint32_t(kTileBytes) + uint32_t(kTileCb + 64 * lp + 8 * hcr + 4 * (1 - hh));
    }
    // tile preparation (prepareYBR): the lane's LDS addresses, likewise
    uint8_t *py_left, *pc_left, *p_top, *p_rr;
    {
        const int ln = int(threadIdx.x) & 63, g = ln >> 4, l = ln & 15;
        MbGroupLds& G = lds.grp[g];
        ReconTile& T = G.rt;
        py_left = &T.y[l][3];                                                  // + 16: column 19, the previous MB's last
        pc_left = &T.c[min(l, 8)][3];                                          // + 8 / + 24: columns 11 / 27; + 16: column 19
        p_top = l < 4 ? &T.y[0][4 + 4 * l] : l == 4 ? &T.y[0][20] : l < 8 ? &T.c[0][4 + 4 * (l - 6)]
              : l < 10 ? &T.c[0][20 + 4 * (l - 8)] : &T.y[4 * (l - 9)][20];  // where this lane's dword of the row above goes
        p_rr = reinterpret_cast<uint8_t*>(&G.rring(0, l < 4 ? l : (l >= 6 && l <= 9) ? l - 2 : 3));      // + 96 per ring slot
    }
    auto fetch_above = [&](int col, int ln, bool edge_too) {
        const int l = ln & 15;
        if (l < 8 && edge_too) top_n = *reinterpret_cast<const uint32_t*>(E + size_t(fa_e + uint32_t(col) * uint32_t(kEdgeBytes)));
        nx_ty = *reinterpret_cast<const uint32_t*>(S + size_t(fa_y + uint32_t(col) * uint32_t(kTileBytes)));
        nx_tc = *reinterpret_cast<const uint32_t*>(S + size_t(fa_c + uint32_t(col) * uint32_t(kTileBytes)));
    };
    {
        const int ln = opaque_lane();
        prefetch(-2, ln);
        prefetch(-1, ln);
        if (has_above) {
            wait_progress(&progress[prev_wave], (q_prev << kSeqShift) | uint32_t(min(2, mbw)));
            if (ln < 16) {                                  // group 0
                if (ln < 8) lds.grp[0].rring(0, ln) = *reinterpret_cast<const uint32_t*>(E + size_t(etop + 4u * uint32_t(ln)));
                fetch_above(0, ln, mbw > 1);
            }
        }
    }
    settle(rec_n1); settle(rec_n2); settle(nl0); settle(nl1); settle(nwv); settle(nc0); settle(nc1); settle(top_n);
    settle(nfp); settle(nfp2); settle(nx_ty); settle(nx_tc);
    uint32_t pending_publish = 0;
    WP_PROF_DECL;
    WP_PROF_MARK(9);

    for (int s = 0; s < n_steps; ++s) {
        WP_PROF_MARK(8);
        const uint32_t cur_ty = nx_ty, cur_tc = nx_tc;
        uint32_t fp;
        // ============================== first half: reconstruction (vp8_recon_band.h) ==============================
        {
            const int ln = opaque_lane();
            const int g = ln >> 4, l = ln & 15;
            const int bj = l >> 2, bi = l & 3, lp = l >> 3, lr = l & 7;
            MbGroupLds& G = lds.grp[g];
            ReconTile& T = G.rt;
            const int y = band * kBandRows + g;
            const int x = s - 2 * g;
            const bool act = y < mbh && x >= 0 && x < mbw;
            const uint4 rec = rec_n1;                                   // zeros when the MB is outside the frame
            fp = ftype != 0 ? nfp : 0u;
            // ---- band boundary: park the column fetched during the previous step, fetch the next ----
            if (has_above && s + 1 < mbw) {
                if (ln < 8) lds.grp[0].rring((s + 1) & 3, ln) = top_n;
                wait_progress(&progress[prev_wave], (q_prev << kSeqShift) | uint32_t(min(s + 3, mbw)));
                if (ln < 16) fetch_above(s + 1, ln, s + 2 < mbw);
            }
            WP_PROF_MARK(0);
            // ---- residuals (independent of the neighbours) ----
            const uint32_t w2 = rec.z;
            const bool i16 = (w2 >> WP_MB_I16_BIT) & 1;
            const int has_wht = (w2 >> WP_MB_WHT_BIT) & 1;
            uint32_t ev[4] = {0, 0, 0, 0}, od[4] = {0, 0, 0, 0};          // luma residual of block l
            uint32_t cev[4] = {0, 0, 0, 0}, cod[4] = {0, 0, 0, 0};        // chroma residual (chroma lanes)
            {
                int c[16];
                unpack_block(nl0, nl1, c);
                const bool present = (w2 >> l) & 1;
                int dc16 = 0;
                if (__builtin_amdgcn_ballot_w64(has_wht) != 0) {
                    dc16 = wht_dc_lanes(int(nwv), l);                   // nwv is 0 without a WHT: all-zero input, zero output
                    if (has_wht) c[0] = dc16;
                }
                if (__builtin_amdgcn_ballot_w64(present || dc16 != 0) != 0) idct_residual(c, ev, od);
            }
            {
#ifdef WP_KO_CHROMA_AC
                const bool c_ac = false;
#else
                const bool c_ac = ((nc0.x >> 16) | nc0.y | nc0.z | nc0.w | nc1.x | nc1.y | nc1.z | nc1.w) != 0;
#endif
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
            prefetch(s, ln);
            WP_PROF_MARK(1);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_setprio(WP_MB_PRIO_P);

            if (act) {
                // ---- prepareYBR (Vp8Decoder.hx:541-577) ----
                {
                    uint32_t va = py_left[16], vb = T.y[16][19], vc1 = pc_left[8], vc2 = pc_left[24];     // last column of the previous MB: T.y[l][19], T.c[min(l, 8)][11 / 27]
                    const bool tr_job = l == 4 || (l >= 10 && l <= 12);
                    uint32_t rv = *reinterpret_cast<uint32_t*>(p_rr + 96 * (x & 3));                       // G.rring(x & 3, l < 4 ? l : 6 <= l <= 9 ? l - 2 : 3)
                    const uint32_t nxt = G.rring((x + 1) & 3, 0);
                    if (tr_job) rv = x == mbw - 1 ? splat4(int(rv >> 24)) : nxt;
                    if (x == 0) { va = 0x81; vb = 0x81; vc1 = 0x81; vc2 = 0x81; }
                    if (y == 0) {
                        rv = 0x7f7f7f7fu;
                        if (l == 0) { va = 0x7f; vc1 = 0x7f; vc2 = 0x7f; }
                    }
                    py_left[0] = uint8_t(va);                                                               // T.y[l][3]
                    if (l == 0) T.y[16][3] = uint8_t(vb);
                    if (l < 9) { pc_left[0] = uint8_t(vc1); pc_left[16] = uint8_t(vc2); }                  // T.c[l][3], T.c[l][19]
                    if (l < 5 || (l >= 6 && l <= 12)) tile32(p_top, 0) = rv;
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
#ifdef WP_KO_I4             // timing-only build: no 4x4 luma prediction (what the ten mini-steps cost)
            if (false) {
#else
            if (__builtin_amdgcn_ballot_w64(act && !i16) != 0) {
#endif
                __builtin_amdgcn_s_setprio(WP_MB_PRIO_I4);
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
                __builtin_amdgcn_s_setprio(WP_MB_PRIO_P);
            }
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
                if ((l & 4) == 0) {
                    const int cj = (l >> 1) & 1, ci = l & 1;
                    const uint32_t top4 = tile32(T.c[0], ccol + 4 * ci);
                    const uint32_t left4 = pack4(T.c[4 * cj + 1][ccol - 1], T.c[4 * cj + 2][ccol - 1], T.c[4 * cj + 3][ccol - 1], T.c[4 * cj + 4][ccol - 1]);
                    uint32_t pr[4];
                    pred_large(c8, top4, left4, int(T.c[0][ccol - 1]), dcv, pr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) tile32(T.c[4 * cj + 1 + r], ccol + 4 * ci) = add_clip_row(pr[r], cev[r], cod[r]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        WP_PROF_MARK(2);

        // ============================== second half: the MB is reconstructed (unfiltered) in T: hand its bottom row on,
        //                                filter it, write the S tile, hand the filtered bottom rows down ==============================
        {
            const int ln = opaque_lane();
            const int g = ln >> 4, l = ln & 15;
            const int lp = l >> 3, lr = l & 7, hcr = (l >> 1) & 3, hh = l & 1, qr = l >> 2, qd = l & 3;
            MbGroupLds& G = lds.grp[g];
            MbGroupLds& GN = lds.grp[(g + 1) & 3];                  // the group below (never written by the band's last group)
            ReconTile& T = G.rt;
            MbFilterTile& F = G.ft;
            const int y = band * kBandRows + g;
            const int x = s - 2 * g;
            const bool row_ok = y < mbh;
            const bool act = row_ok && x >= 0 && x < mbw;
            const bool last_row = g == rows_here - 1;               // its bottom rows leave the band through memory
            const int level = int(fp & 0xff), ilevel = int((fp >> 8) & 0xff), hlevel = int((fp >> 16) & 0xff);
            const bool inner = (fp >> 24) & 1;
            const float level_f = float(fp & 0xff), ilevel_f = float((fp >> 8) & 0xff), hlevel_f = float((fp >> 16) & 0xff);   // v_cvt_f32_ubyteN
            const bool do_filter = act && level != 0;                   // `if (f.level == 0) continue` (:421, :449)
            const int cb = 16 * (x & 1), pb = 16 - cb;                  // luma column base of this MB's slot / the previous MB's
            const int cb8 = cb >> 1, pb8 = pb >> 1;
            if (act) {
                // ---- its unfiltered bottom row goes to the group below (Vp8Decoder.hx:561-569 reads Y/Cb/Cr before the
                //      loop filter has run) or, from the band's last row, to the edge buffer ----
                if (l < 8) {
                    const uint32_t v = l < 4 ? tile32(T.y[16], 4 + 4 * l) : l < 6 ? tile32(T.c[8], 4 + 4 * (l - 4)) : tile32(T.c[8], 20 + 4 * (l - 6));
                    if (g + 1 < kBandRows) GN.rring(x & 3, l) = v;
                    if (last_row && has_below) *reinterpret_cast<uint32_t*>(E + size_t(erow + uint32_t(x) * uint32_t(kEdgeBytes) + 4u * uint32_t(l))) = v;
                }
                // ---- the 4 rows above at a band boundary ----
                if (g == 0 && y > 0) {
                    t32(&F.y[qr][cb + 4 * qd]) = cur_ty;
                    t32(&F.c[lp][hcr][cb8 + 4 * hh]) = cur_tc;
                }
            }
            WP_PROF_MARK(3);
            // ---- the MB moves from the reconstruction tile into the loop filter's tile (lane = pixel row) THROUGH the vertical
            //      edges: each lane takes its luma row and its chroma row from T, filters them in registers (columns -4..-1 come
            //      from the previous MB's slot of F) and stores them into F -- one LDS round trip instead of copy + reload ----
            if (act) {
                uint8_t* yr = T.y[1 + l];
                uint4 w = make_uint4(tile32(yr, 4), tile32(yr, 8), tile32(yr, 12), tile32(yr, 16));
                uint8_t* cr = T.c[1 + lr];
                const int cc = lp ? 20 : 4;
                uint2 wc = make_uint2(tile32(cr, cc), tile32(cr, cc + 4));
                if (kFilterF32 && ftype == 2 && do_filter) {
                    {
                        uint32_t v[20];
                        if constexpr (kFilterDen) {
                            unpack4u(t32(&F.y[4 + l][pb + 12]), v);
                            unpack4u(w.x, v + 4); unpack4u(w.y, v + 8); unpack4u(w.z, v + 12); unpack4u(w.w, v + 16);
                            filter_line_d<20>(v, x > 0, inner, level, ilevel, hlevel);
                            t32(&F.y[4 + l][pb + 12]) = pack4u(v);
                            w = make_uint4(pack4u(v + 4), pack4u(v + 8), pack4u(v + 12), pack4u(v + 16));
                        } else {
                            unpack4f(t32(&F.y[4 + l][pb + 12]), v);
                            unpack4f(w.x, v + 4); unpack4f(w.y, v + 8); unpack4f(w.z, v + 12); unpack4f(w.w, v + 16);
                            filter_line_f<20>(v, x > 0, inner, level_f, ilevel_f, hlevel_f);
                            t32(&F.y[4 + l][pb + 12]) = pack4f(v);
                            w = make_uint4(pack4f(v + 4), pack4f(v + 8), pack4f(v + 12), pack4f(v + 16));
                        }
                    }
                    {
                        uint32_t v[12];
                        if constexpr (kFilterDen) {
                            unpack4u(t32(&F.c[lp][4 + lr][pb8 + 4]), v);
                            unpack4u(wc.x, v + 4); unpack4u(wc.y, v + 8);
                            filter_line_d<12>(v, x > 0, inner, level, ilevel, hlevel);
                            t32(&F.c[lp][4 + lr][pb8 + 4]) = pack4u(v);
                            wc = make_uint2(pack4u(v + 4), pack4u(v + 8));
                        } else {
                            unpack4f(t32(&F.c[lp][4 + lr][pb8 + 4]), v);
                            unpack4f(wc.x, v + 4); unpack4f(wc.y, v + 8);
                            filter_line_f<12>(v, x > 0, inner, level_f, ilevel_f, hlevel_f);
                            t32(&F.c[lp][4 + lr][pb8 + 4]) = pack4f(v);
                            wc = make_uint2(pack4f(v + 4), pack4f(v + 8));
                        }
                    }
                } else if (kFilterDen && ftype == 1 && do_filter) {
                    uint32_t v[20];
                    unpack4u(t32(&F.y[4 + l][pb + 12]), v);
                    unpack4u(w.x, v + 4); unpack4u(w.y, v + 8); unpack4u(w.z, v + 12); unpack4u(w.w, v + 16);
                    simple_line_d(v, x > 0, inner, level);
                    t32(&F.y[4 + l][pb + 12]) = pack4u(v);
                    w = make_uint4(pack4u(v + 4), pack4u(v + 8), pack4u(v + 12), pack4u(v + 16));
                } else if (!kFilterDen && ftype != 0 && do_filter) {
                    {
                        int v[20];
                        unpack4(t32(&F.y[4 + l][pb + 12]), v);
                        unpack4(w.x, v + 4); unpack4(w.y, v + 8); unpack4(w.z, v + 12); unpack4(w.w, v + 16);
                        if (kFilterF32 || ftype == 1) filter_line<20, true>(v, x > 0, inner, level, ilevel, hlevel);      // (with the fp32 edges only the simple filter gets here)
                        else filter_line<20, false>(v, x > 0, inner, level, ilevel, hlevel);
                        t32(&F.y[4 + l][pb + 12]) = pack4i(v);
                        w = make_uint4(pack4i(v + 4), pack4i(v + 8), pack4i(v + 12), pack4i(v + 16));
                    }
                    if (!kFilterF32 && ftype == 2) {
                        int v[12];
                        unpack4(t32(&F.c[lp][4 + lr][pb8 + 4]), v);
                        unpack4(wc.x, v + 4); unpack4(wc.y, v + 8);
                        filter_line<12, false>(v, x > 0, inner, level, ilevel, hlevel);
                        t32(&F.c[lp][4 + lr][pb8 + 4]) = pack4i(v);
                        wc = make_uint2(pack4i(v + 4), pack4i(v + 8));
                    }
                }
                t128(&F.y[4 + l][cb]) = w;
                t64(&F.c[lp][4 + lr][cb8]) = wc;
            }
            __builtin_amdgcn_wave_barrier();

            if (ftype != 0) {
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_setprio(WP_MB_PRIO_H);
                if (kFilterF32 && ftype == 2 && do_filter) {
                    // ---- horizontal edges: lane = pixel column; rows -4..15 relative to the MB ----
                    {
                        uint32_t v[20];
#pragma unroll
                        for (int k = 0; k < 20; ++k) v[k] = uint32_t(F.y[k][cb + l]) | (kFilterDen ? 0u : kFBias);
                        if constexpr (kFilterDen) filter_line_d<20>(v, y > 0, inner, level, ilevel, hlevel);
                        else filter_line_f<20>(v, y > 0, inner, level_f, ilevel_f, hlevel_f);
#pragma unroll
                        for (int k = 1; k < 19; ++k) F.y[k][cb + l] = uint8_t(v[k]);
                    }
                    {
                        uint32_t v[12];
#pragma unroll
                        for (int k = 0; k < 12; ++k) v[k] = uint32_t(F.c[lp][k][cb8 + lr]) | (kFilterDen ? 0u : kFBias);
                        if constexpr (kFilterDen) filter_line_d<12>(v, y > 0, inner, level, ilevel, hlevel);
                        else filter_line_f<12>(v, y > 0, inner, level_f, ilevel_f, hlevel_f);
#pragma unroll
                        for (int k = 1; k < 11; ++k) F.c[lp][k][cb8 + lr] = uint8_t(v[k]);
                    }
                } else if (kFilterDen && ftype == 1 && do_filter) {
                    uint32_t v[20];
#pragma unroll
                    for (int k = 0; k < 20; ++k) v[k] = F.y[k][cb + l];
                    simple_line_d(v, y > 0, inner, level);
#pragma unroll
                    for (int k = 3; k < 17; ++k) F.y[k][cb + l] = uint8_t(v[k]);      // (the simple filter changes p0 and q0 only: rows 3, 4, 7, 8, ...)
                } else if (!kFilterDen && do_filter) {
                    // ---- horizontal edges: lane = pixel column; rows -4..15 relative to the MB ----
                    {
                        int v[20];
#pragma unroll
                        for (int k = 0; k < 20; ++k) v[k] = F.y[k][cb + l];
                        if (kFilterF32 || ftype == 1) filter_line<20, true>(v, y > 0, inner, level, ilevel, hlevel);
                        else filter_line<20, false>(v, y > 0, inner, level, ilevel, hlevel);
#pragma unroll
                        for (int k = 1; k < 19; ++k) F.y[k][cb + l] = uint8_t(v[k]);
                    }
                    if (!kFilterF32 && ftype == 2) {
                        int v[12];
#pragma unroll
                        for (int k = 0; k < 12; ++k) v[k] = F.c[lp][k][cb8 + lr];
                        filter_line<12, false>(v, y > 0, inner, level, ilevel, hlevel);
#pragma unroll
                        for (int k = 1; k < 11; ++k) F.c[lp][k][cb8 + lr] = uint8_t(v[k]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            WP_PROF_MARK(4);

            __builtin_amdgcn_s_setprio(WP_MB_PRIO_R);
            // ---- tell the band below how far this band's last row has got: the value of the PREVIOUS step, whose stores
            //      were issued a whole step ago (the release fence then does not wait for their acknowledgements) ----
            settle(rec_n2); settle(nl0); settle(nl1); settle(nwv); settle(nc0); settle(nc1); settle(top_n);
            settle(nfp2); settle(nx_ty); settle(nx_tc);      // the prefetches landed long ago; wait before the stores go out
            if (has_below) {
                if (pending_publish) publish_progress(&progress[wave], pending_publish);
                const int done = s - 2 * (kBandRows - 1) + 1;
                pending_publish = done >= 1 ? ((q << kSeqShift) | uint32_t(done)) : 0u;
            }
            WP_PROF_MARK(5);

            const uint32_t srow = uint32_t(row_ok ? y : 0) * uint32_t(spitch) * uint32_t(kTileBytes);      // S tiles of tile row y
            uint32_t v_fin = 0, v_low = 0, vc = 0;
            if (act) {
                // ---- S tile (x, y): luma rows -4..7 (lanes 4..15; rows -8..-5 came from the group above), chroma rows -4..3;
                //      cols -8..7 (chroma -4..3) relative to the MB: all final ----
                uint8_t* st = S + size_t(srow + uint32_t(x) * uint32_t(kTileBytes));
                if (l >= 4) {
                    const uint2 a = t64(&F.y[l - 4][pb + 8]), b = t64(&F.y[l - 4][cb]);
                    st_nt(reinterpret_cast<uint4*>(st + 16 * l), make_uint4(a.x, a.y, b.x, b.y));
                    if (x == mbw - 1) *reinterpret_cast<uint2*>(st + kTileBytes + 16 * l) = t64(&F.y[l - 4][cb + 8]);      // right-most tile column: cols 8..15
                }
                st_nt(reinterpret_cast<uint2*>(st + kTileCb + 8 * l), make_uint2(t32(&F.c[lp][lr][pb8 + 4]), t32(&F.c[lp][lr][cb8])));
                if (x == mbw - 1) *reinterpret_cast<uint32_t*>(st + kTileBytes + kTileCb + 8 * l) = t32(&F.c[lp][lr][cb8 + 4]);
                // the eight bottom luma rows (cols -8..7; state after this MB) and the four bottom chroma rows, for the hand-down below
                const uint8_t* fsrc = qd < 2 ? &F.y[12 + qr][pb + 8 + 4 * qd] : &F.y[12 + qr][cb + 4 * (qd - 2)];
                v_fin = *reinterpret_cast<const uint32_t*>(fsrc);                     // rows 8..11
                v_low = *reinterpret_cast<const uint32_t*>(fsrc + 4 * 48);            // rows 12..15
                vc = t32(&F.c[lp][8 + hcr][hh ? cb8 : pb8 + 4]);
            }
            __builtin_amdgcn_wave_barrier();      // the reads of rows 0..3 above come before the group above overwrites them below (same columns mod 32 / 16)
            if (act) {
                // ---- hand-down: rows 8..11 are final and go to S tile (x, y+1) rows -8..-5; rows 12..15 to rows 0..3 of the tile
                //      of the group below, or -- from the band's last row -- to the same S tile, where the next band (or the
                //      finishing pass) picks them up.  Chroma: 4 rows. ----
                uint8_t* sn = S + size_t(srow + (uint32_t(x) + uint32_t(spitch)) * uint32_t(kTileBytes));       // S tile (x, y+1); tile row mbh exists
                *reinterpret_cast<uint32_t*>(sn + 16 * qr + 4 * qd) = v_fin;
                if (!last_row) {
                    t32(&GN.ft.y[qr][(cb - 8 + 4 * qd) & 31]) = v_low;
                    t32(&GN.ft.c[lp][hcr][(cb8 - 4 + 4 * hh) & 15]) = vc;
                } else {
                    *reinterpret_cast<uint32_t*>(sn + 64 + 16 * qr + 4 * qd) = v_low;
                    *reinterpret_cast<uint32_t*>(sn + kTileCb + 64 * lp + 8 * hcr + 4 * hh) = vc;
                    if (x == mbw - 1) {
                        if (qd < 2) *reinterpret_cast<uint32_t*>(sn + kTileBytes + 64 + 16 * qr + 4 * qd) = t32(&F.y[16 + qr][cb + 8 + 4 * qd]);
                        if (hh) *reinterpret_cast<uint32_t*>(sn + kTileBytes + kTileCb + 64 * lp + 8 * hcr) = t32(&F.c[lp][8 + hcr][cb8 + 4]);
                    }
                }
                if (x == mbw - 1 && qd < 2) *reinterpret_cast<uint32_t*>(sn + kTileBytes + 16 * qr + 4 * qd) = t32(&F.y[12 + qr][cb + 8 + 4 * qd]);
            } else if (row_ok && x == mbw && !last_row) {
                // cols 8..15 of the row's last MB reach the tile below one step late: a step earlier they would land on columns
                // (mod 32 / mod 16) the group below still reads.  Relative to the virtual MB `mbw` they are its cols -8..-1.
                if (qd < 2) t32(&GN.ft.y[qr][pb + 8 + 4 * qd]) = t32(&F.y[16 + qr][pb + 8 + 4 * qd]);
                if (!hh) t32(&GN.ft.c[lp][hcr][pb8 + 4]) = t32(&F.c[lp][8 + hcr][pb8 + 4]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        WP_PROF_MARK(6);
    }
    if (has_below && pending_publish) publish_progress(&progress[wave], pending_publish);
    WP_PROF_FLUSH(0);
}

}  // namespace wpk
