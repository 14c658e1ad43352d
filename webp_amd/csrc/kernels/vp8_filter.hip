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

// filter2, one position (Filter.hx:29-50)
__device__ __forceinline__ void simple_edge(int p1, int& p0, int& q0, int q1, int level) {
    if ((iabs(p0 - q0) << 1) + (iabs(p1 - q1) >> 1) > level) return;
    const int a = 3 * (q0 - p0) + clamp127(p1 - q1);
    const int a1 = clamp15((a + 4) >> 3);
    const int a2 = clamp15((a + 3) >> 3);
    p0 = clamp255(p0 + a2);
    q0 = clamp255(q0 - a1);
}

// filter246, one position (Filter.hx:53-111)
__device__ __forceinline__ void normal_edge(int p3, int& p2, int& p1, int& p0, int& q0, int& q1, int& q2, int q3,
                                            int level, int ilevel, int hlevel, bool four_not_six) {
    if ((iabs(p0 - q0) << 1) + (iabs(p1 - q1) >> 1) > level || iabs(p3 - p2) > ilevel || iabs(p2 - p1) > ilevel ||
        iabs(p1 - p0) > ilevel || iabs(q1 - q0) > ilevel || iabs(q2 - q1) > ilevel || iabs(q3 - q2) > ilevel)
        return;
    if (iabs(p1 - p0) > hlevel || iabs(q1 - q0) > hlevel) {
        const int a = 3 * (q0 - p0) + clamp127(p1 - q1);
        const int a1 = clamp15((a + 4) >> 3);
        const int a2 = clamp15((a + 3) >> 3);
        p0 = clamp255(p0 + a2);
        q0 = clamp255(q0 - a1);
    } else if (four_not_six) {
        const int a = 3 * (q0 - p0);
        const int a1 = clamp15((a + 4) >> 3);
        const int a2 = clamp15((a + 3) >> 3);
        const int a3 = (a1 + 1) >> 1;
        p1 = clamp255(p1 + a3);
        p0 = clamp255(p0 + a2);
        q0 = clamp255(q0 - a1);
        q1 = clamp255(q1 - a3);
    } else {
        const int a = clamp127(3 * (q0 - p0) + clamp127(p1 - q1));
        const int a1 = (27 * a + 63) >> 7;
        const int a2 = (18 * a + 63) >> 7;
        const int a3 = (9 * a + 63) >> 7;
        p2 = clamp255(p2 + a3);
        p1 = clamp255(p1 + a2);
        p0 = clamp255(p0 + a1);
        q0 = clamp255(q0 - a1);
        q1 = clamp255(q1 - a2);
        q2 = clamp255(q2 - a3);
    }
}

// One line of N = 4 + 16 (luma) or 4 + 8 (chroma) samples crossing the MB edge at index 4 and the
// inner edges at 8 (and 12, 16 for luma).  Order as in the reference: MB edge first, then inner edges.
template <int N>
__device__ __forceinline__ void filter_line(int (&v)[N], bool simple, bool mb_edge, bool inner, int level, int ilevel, int hlevel) {
    if (simple) {
        if (mb_edge) simple_edge(v[2], v[3], v[4], v[5], level + 4);
        if (inner) {
#pragma unroll
            for (int e = 8; e < N; e += 4) simple_edge(v[e - 2], v[e - 1], v[e], v[e + 1], level);
        }
    } else {
        if (mb_edge) normal_edge(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], level + 4, ilevel, hlevel, false);
        if (inner) {
#pragma unroll
            for (int e = 8; e < N; e += 4)
                normal_edge(v[e - 4], v[e - 3], v[e - 2], v[e - 1], v[e], v[e + 1], v[e + 2], v[e + 3], level, ilevel, hlevel, true);
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

            for (int s = 0; s < n_steps; ++s) {
                const int x = s - 2 * g;
                const bool act = row_ok && x >= 0 && x < mbw;
                if (has_above && s < mbw)
                    wait_progress(&progress[prev_wave], ((q - 1) << kSeqShift) | uint32_t(min(s + 2, mbw)));

                if (act) {
                    const uint32_t fp = filt[size_t(fr.mb_base) + size_t(y) * mbw + x];
                    const int level = int(fp & 0xff), ilevel = int((fp >> 8) & 0xff), hlevel = int((fp >> 16) & 0xff);
                    const bool inner = (fp >> 24) & 1;
                    const bool do_filter = level != 0;                          // `if (f.level == 0) continue` (:421, :449)

                    // ---- 1. slide the tile: right-most 4 columns become the left strip ----
                    if (x > 0) {
                        t32(&T.y[4 + l][0]) = t32(&T.y[4 + l][16]);
                        if (!simple) { const int p = l >> 3, r = l & 7; t32(&T.c[p][4 + r][0]) = t32(&T.c[p][4 + r][8]); }
                    }
                    __builtin_amdgcn_wave_barrier();
                    // own macroblock, straight from the reconstructed planes (nobody has touched it yet)
                    {
                        const uint4 v = *reinterpret_cast<const uint4*>(Y + size_t(16 * y + l) * ystride + 16 * x);
                        t32(&T.y[4 + l][4]) = v.x; t32(&T.y[4 + l][8]) = v.y; t32(&T.y[4 + l][12]) = v.z; t32(&T.y[4 + l][16]) = v.w;
                        if (!simple) {
                            const int p = l >> 3, r = l & 7;
                            const uint2 c = *reinterpret_cast<const uint2*>((p ? V : U) + size_t(8 * y + r) * cstride + 8 * x);
                            t32(&T.c[p][4 + r][4]) = c.x; t32(&T.c[p][4 + r][8]) = c.y;
                        }
                    }
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
                                const uint4 v = *reinterpret_cast<const uint4*>(Y + size_t(16 * y - 4 + l) * ystride + 16 * x);
                                t32(&T.y[l][4]) = v.x; t32(&T.y[l][8]) = v.y; t32(&T.y[l][12]) = v.z; t32(&T.y[l][16]) = v.w;
                            } else if (!simple && l < 12) {
                                const int p = (l - 4) >> 2, r = (l - 4) & 3;
                                const uint2 c = *reinterpret_cast<const uint2*>((p ? V : U) + size_t(8 * y - 4 + r) * cstride + 8 * x);
                                t32(&T.c[p][r][4]) = c.x; t32(&T.c[p][r][8]) = c.y;
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
                            const int p = l >> 3, r = l & 7;
                            int v[12];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const uint32_t w = t32(&T.c[p][4 + r][4 * k]);
                                v[4 * k] = w & 0xff; v[4 * k + 1] = (w >> 8) & 0xff; v[4 * k + 2] = (w >> 16) & 0xff; v[4 * k + 3] = w >> 24;
                            }
                            filter_line<12>(v, false, x > 0, inner, level, ilevel, hlevel);
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                t32(&T.c[p][4 + r][4 * k]) = uint32_t(v[4 * k]) | uint32_t(v[4 * k + 1]) << 8 | uint32_t(v[4 * k + 2]) << 16 | uint32_t(v[4 * k + 3]) << 24;
                        }
                        __builtin_amdgcn_wave_barrier();
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
                            const int p = l >> 3, cidx = l & 7;
                            int v[12];
#pragma unroll
                            for (int k = 0; k < 12; ++k) v[k] = T.c[p][k][4 + cidx];
                            filter_line<12>(v, false, y > 0, inner, level, ilevel, hlevel);
#pragma unroll
                            for (int k = 1; k < 11; ++k) T.c[p][k][4 + cidx] = uint8_t(v[k]);
                        }
                        __builtin_amdgcn_wave_barrier();

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
                            const int p = l >> 3, r = l & 7;
                            uint8_t* dst = (p ? V : U) + size_t(8 * y + r) * cstride + 8 * x;
                            if (x > 0) *reinterpret_cast<uint32_t*>(dst - 4) = t32(&T.c[p][4 + r][0]);
                            *reinterpret_cast<uint2*>(dst) = make_uint2(t32(&T.c[p][4 + r][4]), t32(&T.c[p][4 + r][8]));
                            if (y > 0 && (l & 3) != 0 && l < 8) {
                                const int pp = l >> 2, rr = l & 3;
                                uint8_t* up = (pp ? V : U) + size_t(8 * y - 4 + rr) * cstride + 8 * x;
                                *reinterpret_cast<uint2*>(up) = make_uint2(t32(&T.c[pp][rr][4]), t32(&T.c[pp][rr][8]));
                            }
                        }
                    }

                    // ---- 4. hand the four bottom rows (current state) to the group below ----
                    if (g + 1 < kBandRows) {
                        const int r = l >> 2, k = l & 3;           // 4 rows x 4 dwords of the MB itself
                        t32(&lds.ring_y[g + 1][r][(16 * x + 4 * k) & 63]) = t32(&T.y[16 + r][4 + 4 * k]);
                        if (x > 0 && l < 4) t32(&lds.ring_y[g + 1][l][(16 * x - 4) & 63]) = t32(&T.y[16 + l][0]);
                        if (!simple) {
                            const int p = l >> 3, rr = (l >> 1) & 3, kk = l & 1;   // 2 planes x 4 rows x 2 dwords
                            t32(&lds.ring_c[g + 1][p][rr][(8 * x + 4 * kk) & 31]) = t32(&T.c[p][8 + rr][4 + 4 * kk]);
                            if (x > 0 && l < 8) t32(&lds.ring_c[g + 1][l >> 2][l & 3][(8 * x - 4) & 31]) = t32(&T.c[l >> 2][8 + (l & 3)][0]);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (has_below) {
                    const int done = s - 2 * (kBandRows - 1) + 1;
                    if (done >= 1) publish_progress(&progress[wave], (q << kSeqShift) | uint32_t(done));
                }
            }
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
