// VP8 loop filter (simple and normal) for a batch of frames: reads the reconstructed macroblock tiles (R),
// writes the filtered image as shifted tiles (S) -- see wp_device.h for the two layouts.
//
// Replaces simpleFilter / normalFilter (vp8/Vp8Decoder.hx:417-486) and vp8/Filter.hx.
//
// The reference filters macroblocks in raster order; each MB does: left MB edge, inner vertical
// edges 4/8/12, top MB edge, inner horizontal edges 4/8/12, and reaches 4 pixels into its left and
// upper neighbours.  MB (x,y) therefore has to run after (x-1,y), (x,y-1) and (x+1,y-1) -- the same
// 2:1 anti-diagonal skew as reconstruction -- and any schedule that respects it is bit-identical to
// raster order.  As in vp8_recon.hip one wavefront owns a band of 4 MB rows (16 lanes per row,
// lockstep skew).  A group keeps a 24-row luma tile with two 16-column MB slots (+ two 12-row chroma
// tiles with two 8-column slots) in LDS: vertical edges are filtered with lane = pixel row, horizontal
// edges with lane = pixel column.  After MB (x,y) everything up to row 16y+12 / column 16x+12 is final,
// so the group emits S tile (x,y) = rows 16y-8..16y+7, cols 16x-8..16x+7 (chroma: 8y-4.., 8x-4..): every
// output byte is written exactly once, 384 contiguous bytes per MB.  The eight bottom rows (state after
// this MB row's filtering) go to the group below through an LDS ring; the last row of a band writes
// them as the top half of the S tiles of the next tile row, from where the next band (or, for the
// frame's last row, the finishing pass) picks them up (workgroup-scope release/acquire, same CU).
#include "vp8_filter_band.h"

namespace wpk {

// workgroup shape and occupancy target: waves per workgroup, workgroups per CU (=> waves per SIMD = product / 4)
#ifndef WP_FILTER_WAVES
#define WP_FILTER_WAVES 8
#endif
#ifndef WP_FILTER_WGS_PER_CU
#define WP_FILTER_WGS_PER_CU 2
#endif
constexpr int kFilterWaves = WP_FILTER_WAVES;
constexpr int kFilterWGsPerCU = WP_FILTER_WGS_PER_CU;

__global__ __launch_bounds__(kFilterWaves * kWave, (kFilterWaves * kFilterWGsPerCU + 3) / 4)
void vp8_filter_kernel(int n_frames, const wp_vp8_frame* __restrict__ frames, const uint32_t* __restrict__ filt,
                       const uint8_t* __restrict__ Rp, uint8_t* Sp) {
    __shared__ FilterWaveLds lds_all[kFilterWaves];
    __shared__ uint32_t progress[kFilterWaves];
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < kFilterWaves) progress[threadIdx.x] = 0;
    __syncthreads();

    uint32_t seq = 1;
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const wp_vp8_frame fr = frames[f];
        const int n_bands = (fr.mbh + kBandRows - 1) / kBandRows;
        for (int band = 0; band < n_bands; ++band) {
            const uint32_t q = seq + uint32_t(band);
            if (int(q % kFilterWaves) != wave) continue;
            // filterHeader.level == 0 (Vp8Decoder.hx:365): no filtering, the frame is only re-tiled
            if (fr.filter_type != 2) filter_band<kFilterWaves, true>(fr, band, n_bands, q, q - 1, wave, progress, lds_all[wave], filt, Rp, Sp, fr.filter_type == 0);
            else filter_band<kFilterWaves, false>(fr, band, n_bands, q, q - 1, wave, progress, lds_all[wave], filt, Rp, Sp);
        }
        seq += uint32_t(n_bands);
    }
}

}  // namespace wpk

#ifdef WP_PROFILE
extern "C" int wp_debug_profile_filter(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(wpk::wp_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(wpk::wp_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

extern "C" int wpk_launch_vp8_filter(int n, int cus, const wp_vp8_frame* d_frames, const uint32_t* d_filt,
                                     const uint8_t* d_rtiles, uint8_t* d_stiles, hipStream_t stream) {
    const int grid = n < cus * wpk::kFilterWGsPerCU ? n : cus * wpk::kFilterWGsPerCU;
    hipLaunchKernelGGL(wpk::vp8_filter_kernel, dim3(grid), dim3(wpk::kFilterWaves * wpk::kWave), 0, stream,
                       n, d_frames, d_filt, d_rtiles, d_stiles);
    return int(hipGetLastError());
}
