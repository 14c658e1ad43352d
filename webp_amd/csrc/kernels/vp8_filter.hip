// VP8 loop filter (simple and normal), in place on the reconstructed planes, for a batch of frames.
//
// Replaces simpleFilter / normalFilter (vp8/Vp8Decoder.hx:417-486) and vp8/Filter.hx.
//
// The reference filters macroblocks in raster order; each MB does: left MB edge, inner vertical
// edges 4/8/12, top MB edge, inner horizontal edges 4/8/12, and reaches 4 pixels into its left and
// upper neighbours.  MB (x,y) therefore has to run after (x-1,y), (x,y-1) and (x+1,y-1) -- the same
// 2:1 anti-diagonal skew as reconstruction -- and any schedule that respects it is bit-identical to
// raster order.  As in vp8_recon.hip one wavefront owns a band of 4 MB rows (16 lanes per row,
// lockstep skew).  A group keeps a 20x20 luma tile (+ two 12x12 chroma tiles) in LDS: vertical
// edges are filtered with lane = pixel row, horizontal edges with lane = pixel column.  The four
// right-most columns stay in the tile for the next MB; the four bottom rows go through a small LDS
// ring to the group below; band-to-band hand-over goes through the planes themselves
// (workgroup-scope release/acquire, same CU).
#include "wp_device.h"

namespace wpk {

struct FilterTile {
    uint8_t y[20][24];          // rows 0..3: above, 4..19: MB; cols 0..3: left, 4..19: MB
    uint8_t c[2][12][12];       // same shape at chroma scale (4 + 8)
    uint8_t pad[32];            // 800 B = 200 dwords: the four groups' tiles start 8 banks apart
};
struct FilterWaveLds {
    FilterTile tile[kBandRows];
    uint8_t ring_y[kBandRows][4][64];      // ring_y[g][row][abs_col & 63]: 4 bottom luma rows of the MB row above group g
    uint8_t ring_c[kBandRows][2][4][32];   // same for Cb / Cr
};

__device__ __forceinline__ uint32_t& t32(uint8_t* p) { return *reinterpret_cast<uint32_t*>(p); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clamp127(int v) { return clampi(v, -127, 127); }   // Filter.hx:16-18 (symmetric)
__device__ __forceinline__ int clamp15(int v) { return clampi(v, -15, 15); }      // Filter.hx:20-22
__device__ __forceinline__ int clamp255(int v) { return clampi(v, 0, 255); }

// The edge filters are written branch-free (every lane of the wavefront handles a different pixel
// position, so all branches would be executed anyway) and take / return values, never references:
// the sample arrays then stay in registers.  "Not filtered" is expressed as zero deltas, which
// leaves the 0..255 samples unchanged through clamp255.

// filter2, one position (Filter.hx:29-50): returns the deltas for p0 (added) and q0 (subtracted)
__device__ __forceinline__ void simple_edge(int p1, int& p0, int& q0, int q1, int level) {
    const bool on = (iabs(p0 - q0) << 1) + (iabs(p1 - q1) >> 1) <= level;
    const int a = 3 * (q0 - p0) + clamp127(p1 - q1);
    const int a1 = on ? clamp15((a + 4) >> 3) : 0;
    const int a2 = on ? clamp15((a + 3) >> 3) : 0;
    p0 = clamp255(p0 + a2);
    q0 = clamp255(q0 - a1);
}

// filter246, one position (Filter.hx:53-111).  FOUR = inner edge (4-pixel filter), else MB edge (6-pixel).
template <bool FOUR>
__device__ __forceinline__ void normal_edge(int p3, int& p2, int& p1, int& p0, int& q0, int& q1, int& q2, int q3,
                                            int level, int ilevel, int hlevel) {
    const int d10 = iabs(p1 - p0), e10 = iabs(q1 - q0);
    const int worst = max(max(max(iabs(p3 - p2), iabs(p2 - p1)), max(d10, e10)), max(iabs(q2 - q1), iabs(q3 - q2)));
    const bool on = (iabs(p0 - q0) << 1) + (iabs(p1 - q1) >> 1) <= level && worst <= ilevel;
    const bool hev = d10 > hlevel || e10 > hlevel;
    const int base = 3 * (q0 - p0);
    const int outer = clamp127(p1 - q1);
    if (FOUR) {
        const int a = base + (hev ? outer : 0);
        const int a1 = on ? clamp15((a + 4) >> 3) : 0;
        const int a2 = on ? clamp15((a + 3) >> 3) : 0;
        const int a3 = hev ? 0 : (a1 + 1) >> 1;            // a1 == 0 when !on
        p1 = clamp255(p1 + a3);
        p0 = clamp255(p0 + a2);
        q0 = clamp255(q0 - a1);
        q1 = clamp255(q1 - a3);
    } else {
        const int w = base + outer;
        const int a = clamp127(w);
        const int c1 = (27 * a + 63) >> 7, c2 = (18 * a + 63) >> 7, c3 = (9 * a + 63) >> 7;
        const int dp0 = on ? (hev ? clamp15((w + 3) >> 3) : c1) : 0;
        const int dq0 = on ? (hev ? clamp15((w + 4) >> 3) : c1) : 0;
        const int d1 = (on && !hev) ? c2 : 0;
        const int d2 = (on && !hev) ? c3 : 0;
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
// Edges are written out with constant indices so that the sample array stays in registers.
template <int N>
__device__ __forceinline__ void filter_line(int (&v)[N], bool simple, bool mb_edge, bool inner, int level, int ilevel, int hlevel) {
    if (simple) {
        if (mb_edge) simple_edge(v[2], v[3], v[4], v[5], level + 4);
        if (inner) {
            simple_edge(v[6], v[7], v[8], v[9], level);
            if constexpr (N == 20) {
                simple_edge(v[10], v[11], v[12], v[13], level);
                simple_edge(v[14], v[15], v[16], v[17], level);
            }
        }
    } else {
        if (mb_edge) normal_edge<false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], level + 4, ilevel, hlevel);
        if (inner) {
            normal_edge<true>(v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], level, ilevel, hlevel);
            if constexpr (N == 20) {
                normal_edge<true>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], level, ilevel, hlevel);
                normal_edge<true>(v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19], level, ilevel, hlevel);
            }
        }
    }
}

__global__ __launch_bounds__(kWavesPerWG * kWave)
void vp8_filter_kernel(int n_frames, const wp_vp8_frame* __restrict__ frames, const uint32_t* __restrict__ filt,
                       uint8_t* Yp, uint8_t* Up, uint8_t* Vp) {
    __shared__ FilterWaveLds lds_all[kWavesPerWG];
    __shared__ uint32_t progress[kWavesPerWG];

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4;
    const int l = lane & 15;
    FilterWaveLds& lds = lds_all[wave];
    FilterTile& T = lds.tile[g];

    if (threadIdx.x < kWavesPerWG) progress[threadIdx.x] = 0;
    __syncthreads();

    uint32_t seq = 1;
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const wp_vp8_frame fr = frames[f];
        const int mbw = fr.mbw, mbh = fr.mbh;
        const int n_bands = (mbh + kBandRows - 1) / kBandRows;
        if (fr.filter_type == 0) { seq += uint32_t(n_bands); continue; }      // filterHeader.level == 0 (Vp8Decoder.hx:365)
        const bool simple = fr.filter_type == 1;
        const int ystride = 16 * mbw, cstride = 8 * mbw;
        uint8_t* const Y = Yp + fr.y_off;
        uint8_t* const U = Up + fr.u_off;
        uint8_t* const V = Vp + fr.v_off;

        for (int band = 0; band < n_bands; ++band) {
            const uint32_t q = seq + uint32_t(band);
            if (int(q % kWavesPerWG) != wave) continue;
            const int y = band * kBandRows + g;
            const bool row_ok = y < mbh;
            const bool has_above = band > 0;
            const bool has_below = band + 1 < n_bands;
            const int prev_wave = (wave + kWavesPerWG - 1) % kWavesPerWG;
            const int rows_here = min(kBandRows, mbh - band * kBandRows);
            const int n_steps = mbw + 2 * (rows_here - 1);

            // Everything a step reads from global memory is fetched one step ahead (registers), so no global
            // round trip sits on a step's critical path: the MB's own pixels and filter record, and -- for the
            // band's first row -- the four rows above (which makes this band trail the one above by 3 MBs).
            const int lp = l >> 3, lr = l & 7;                    // chroma: plane / row handled by this lane in the row-wise phases
            const uint8_t* yrow = Y + size_t(16 * (row_ok ? y : 0) + l) * ystride;
            const uint8_t* crow = (lp ? V : U) + size_t(8 * (row_ok ? y : 0) + lr) * cstride;
            const uint32_t* frow = filt + size_t(fr.mb_base) + size_t(row_ok ? y : 0) * mbw;
            const bool top_lane_y = has_above && g == 0 && l < 4;
            const bool top_lane_c = has_above && g == 0 && !simple && l >= 4 && l < 12;
            const uint8_t* trow_y = Y + size_t(16 * (row_ok ? y : 0) - (y > 0 ? 4 : 0) + (l & 3)) * ystride;
            const uint8_t* trow_c = (((l - 4) >> 2) & 1 ? V : U) + size_t(8 * (row_ok ? y : 0) - (y > 0 ? 4 : 0) + ((l - 4) & 3)) * cstride;
            uint4 nx_y = make_uint4(0, 0, 0, 0), nx_ty = make_uint4(0, 0, 0, 0);
            uint2 nx_c = make_uint2(0, 0), nx_tc = make_uint2(0, 0);
            uint32_t nx_fp = 0;
            if (has_above) wait_progress(&progress[prev_wave], ((q - 1) << kSeqShift) | uint32_t(min(2, mbw)));
            if (row_ok && g == 0) {
                nx_y = *reinterpret_cast<const uint4*>(yrow);
                if (!simple) nx_c = *reinterpret_cast<const uint2*>(crow);
                nx_fp = frow[0];
                if (top_lane_y) nx_ty = *reinterpret_cast<const uint4*>(trow_y);
                if (top_lane_c) nx_tc = *reinterpret_cast<const uint2*>(trow_c);
            }
            settle(nx_y); settle(nx_ty); settle(nx_c); settle(nx_tc); settle(nx_fp);
            uint32_t pending_publish = 0;

            for (int s = 0; s < n_steps; ++s) {
                const int x = s - 2 * g;
                const bool act = row_ok && x >= 0 && x < mbw;
                const uint4 cur_y = nx_y, cur_ty = nx_ty;
                const uint2 cur_c = nx_c, cur_tc = nx_tc;
                const uint32_t fp = nx_fp;
                if (has_above && s + 1 < mbw)
                    wait_progress(&progress[prev_wave], ((q - 1) << kSeqShift) | uint32_t(min(s + 3, mbw)));
                if (row_ok && x + 1 >= 0 && x + 1 < mbw) {
                    nx_y = *reinterpret_cast<const uint4*>(yrow + 16 * (x + 1));
                    if (!simple) nx_c = *reinterpret_cast<const uint2*>(crow + 8 * (x + 1));
                    nx_fp = frow[x + 1];
                    if (top_lane_y) nx_ty = *reinterpret_cast<const uint4*>(trow_y + 16 * (x + 1));
                    if (top_lane_c) nx_tc = *reinterpret_cast<const uint2*>(trow_c + 8 * (x + 1));
                }

                const int level = int(fp & 0xff), ilevel = int((fp >> 8) & 0xff), hlevel = int((fp >> 16) & 0xff);
                const bool inner = (fp >> 24) & 1;
                const bool do_filter = act && level != 0;                   // `if (f.level == 0) continue` (:421, :449)
                if (act) {
                    // ---- 1. slide the tile: right-most 4 columns become the left strip ----
                    if (x > 0) {
                        t32(&T.y[4 + l][0]) = t32(&T.y[4 + l][16]);
                        if (!simple) t32(&T.c[lp][4 + lr][0]) = t32(&T.c[lp][4 + lr][8]);
                    }
                    __builtin_amdgcn_wave_barrier();
                    // own macroblock, straight from the reconstructed planes (nobody has touched it yet)
                    t32(&T.y[4 + l][4]) = cur_y.x; t32(&T.y[4 + l][8]) = cur_y.y; t32(&T.y[4 + l][12]) = cur_y.z; t32(&T.y[4 + l][16]) = cur_y.w;
                    if (!simple) { t32(&T.c[lp][4 + lr][4]) = cur_c.x; t32(&T.c[lp][4 + lr][8]) = cur_c.y; }
                    // four rows above: from the group above (LDS ring) or, for the band's first row, from the planes
                    if (y > 0) {
                        if (g > 0) {
                            if (l < 4) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) t32(&T.y[l][4 + 4 * k]) = t32(&lds.ring_y[g][l][(16 * x + 4 * k) & 63]);
                            } else if (!simple && l < 12) {
                                const int p = (l - 4) >> 2, r = (l - 4) & 3;
                                t32(&T.c[p][r][4]) = t32(&lds.ring_c[g][p][r][(8 * x) & 31]);
                                t32(&T.c[p][r][8]) = t32(&lds.ring_c[g][p][r][(8 * x + 4) & 31]);
                            }
                        } else {
                            if (l < 4) {
                                t32(&T.y[l][4]) = cur_ty.x; t32(&T.y[l][8]) = cur_ty.y; t32(&T.y[l][12]) = cur_ty.z; t32(&T.y[l][16]) = cur_ty.w;
                            } else if (!simple && l < 12) {
                                const int p = (l - 4) >> 2, r = (l - 4) & 3;
                                t32(&T.c[p][r][4]) = cur_tc.x; t32(&T.c[p][r][8]) = cur_tc.y;
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();

                if (do_filter) {
                    // ---- 2a. vertical edges: lane = pixel row ----
                    {
                        int v[20];
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const uint32_t w = t32(&T.y[4 + l][4 * k]);
                            v[4 * k] = w & 0xff; v[4 * k + 1] = (w >> 8) & 0xff; v[4 * k + 2] = (w >> 16) & 0xff; v[4 * k + 3] = w >> 24;
                        }
                        filter_line<20>(v, simple, x > 0, inner, level, ilevel, hlevel);
#pragma unroll
                        for (int k = 0; k < 5; ++k)
                            t32(&T.y[4 + l][4 * k]) = uint32_t(v[4 * k]) | uint32_t(v[4 * k + 1]) << 8 | uint32_t(v[4 * k + 2]) << 16 | uint32_t(v[4 * k + 3]) << 24;
                    }
                    if (!simple) {
                        int v[12];
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const uint32_t w = t32(&T.c[lp][4 + lr][4 * k]);
                            v[4 * k] = w & 0xff; v[4 * k + 1] = (w >> 8) & 0xff; v[4 * k + 2] = (w >> 16) & 0xff; v[4 * k + 3] = w >> 24;
                        }
                        filter_line<12>(v, false, x > 0, inner, level, ilevel, hlevel);
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            t32(&T.c[lp][4 + lr][4 * k]) = uint32_t(v[4 * k]) | uint32_t(v[4 * k + 1]) << 8 | uint32_t(v[4 * k + 2]) << 16 | uint32_t(v[4 * k + 3]) << 24;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (do_filter) {
                    // ---- 2b. horizontal edges: lane = pixel column ----
                    {
                        int v[20];
#pragma unroll
                        for (int k = 0; k < 20; ++k) v[k] = T.y[k][4 + l];
                        filter_line<20>(v, simple, y > 0, inner, level, ilevel, hlevel);
#pragma unroll
                        for (int k = 1; k < 19; ++k) T.y[k][4 + l] = uint8_t(v[k]);
                    }
                    if (!simple) {
                        int v[12];
#pragma unroll
                        for (int k = 0; k < 12; ++k) v[k] = T.c[lp][k][4 + lr];
                        filter_line<12>(v, false, y > 0, inner, level, ilevel, hlevel);
#pragma unroll
                        for (int k = 1; k < 11; ++k) T.c[lp][k][4 + lr] = uint8_t(v[k]);
                    }
                }
                __builtin_amdgcn_wave_barrier();

                // ---- tell the band below how far this band's last row has got: the value of the PREVIOUS step,
                //      whose stores were issued a whole step ago (the release fence then does not wait for acks) ----
                settle(nx_y); settle(nx_ty); settle(nx_c); settle(nx_tc); settle(nx_fp);   // prefetch landed long ago; wait before the stores go out
                if (has_below) {
                    if (pending_publish) publish_progress(&progress[wave], pending_publish);
                    const int done = s - 2 * (kBandRows - 1) + 1;
                    pending_publish = done >= 1 ? ((q << kSeqShift) | uint32_t(done)) : 0u;
                }

                if (do_filter) {
                    // ---- 3. write back everything this macroblock may have modified ----
                    {
                        uint8_t* dst = Y + size_t(16 * y + l) * ystride + 16 * x;
                        if (x > 0) *reinterpret_cast<uint32_t*>(dst - 4) = t32(&T.y[4 + l][0]);
                        *reinterpret_cast<uint4*>(dst) = make_uint4(t32(&T.y[4 + l][4]), t32(&T.y[4 + l][8]), t32(&T.y[4 + l][12]), t32(&T.y[4 + l][16]));
                        if (y > 0 && l >= 1 && l < 4) {
                            uint8_t* up = Y + size_t(16 * y - 4 + l) * ystride + 16 * x;
                            *reinterpret_cast<uint4*>(up) = make_uint4(t32(&T.y[l][4]), t32(&T.y[l][8]), t32(&T.y[l][12]), t32(&T.y[l][16]));
                        }
                    }
                    if (!simple) {
                        uint8_t* dst = (lp ? V : U) + size_t(8 * y + lr) * cstride + 8 * x;
                        if (x > 0) *reinterpret_cast<uint32_t*>(dst - 4) = t32(&T.c[lp][4 + lr][0]);
                        *reinterpret_cast<uint2*>(dst) = make_uint2(t32(&T.c[lp][4 + lr][4]), t32(&T.c[lp][4 + lr][8]));
                        if (y > 0 && (l & 3) != 0 && l < 8) {
                            const int pp = l >> 2, rr = l & 3;
                            uint8_t* up = (pp ? V : U) + size_t(8 * y - 4 + rr) * cstride + 8 * x;
                            *reinterpret_cast<uint2*>(up) = make_uint2(t32(&T.c[pp][rr][4]), t32(&T.c[pp][rr][8]));
                        }
                    }
                }
                // ---- 4. hand the four bottom rows (current state) to the group below ----
                if (act && g + 1 < kBandRows) {
                    const int r = l >> 2, k = l & 3;           // 4 rows x 4 dwords of the MB itself
                    t32(&lds.ring_y[g + 1][r][(16 * x + 4 * k) & 63]) = t32(&T.y[16 + r][4 + 4 * k]);
                    if (x > 0 && l < 4) t32(&lds.ring_y[g + 1][l][(16 * x - 4) & 63]) = t32(&T.y[16 + l][0]);
                    if (!simple) {
                        const int rr = (l >> 1) & 3, kk = l & 1;   // 2 planes x 4 rows x 2 dwords
                        t32(&lds.ring_c[g + 1][lp][rr][(8 * x + 4 * kk) & 31]) = t32(&T.c[lp][8 + rr][4 + 4 * kk]);
                        if (x > 0 && l < 8) t32(&lds.ring_c[g + 1][l >> 2][l & 3][(8 * x - 4) & 31]) = t32(&T.c[l >> 2][8 + (l & 3)][0]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (has_below && pending_publish) publish_progress(&progress[wave], pending_publish);
        }
        seq += uint32_t(n_bands);
    }
}

}  // namespace wpk

extern "C" int wpk_launch_vp8_filter(int n, int grid, const wp_vp8_frame* d_frames, const uint32_t* d_filt,
                                     uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, hipStream_t stream) {
    hipLaunchKernelGGL(wpk::vp8_filter_kernel, dim3(grid), dim3(wpk::kWavesPerWG * wpk::kWave), 0, stream,
                       n, d_frames, d_filt, d_y, d_u, d_v);
    return int(hipGetLastError());
}
