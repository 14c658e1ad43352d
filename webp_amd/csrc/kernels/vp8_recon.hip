// VP8 intra prediction + inverse WHT/DCT for a batch of frames (gfx950).
//
// Replaces prepareYBR / inverseWHT16 / inverseDCT4* / reconstructMacroblock / copy of the reference
// (vp8/Vp8Decoder.hx:541-577, 716-770, 810-836, 905-996) and vp8/PredFunc.hx.
//
// Schedule: macroblock (x, y) needs (x-1,y), (x,y-1), (x-1,y-1) and (x+1,y-1), so it becomes ready
// at anti-diagonal step s = x + 2y.  One wavefront owns a band of 4 macroblock rows: lane group
// g = lane>>4 works on row 4*band+g at column s-2g, so the four groups run the 2:1 skew in
// lockstep and need no synchronisation among themselves.  Inside a group, lane l owns 4x4 luma
// block l (and lanes 0-3 / 8-11 a chroma block).  Neighbour pixels live in a per-group LDS tile
// with the same geometry as the reference's `ybr` workspace (Reconstruct.hx:3-12), which keeps the
// border rules (0x7f / 0x81 fills, top-right replication, the rows-4/8/12 top-right copy) literal.
// Bands of one frame are pipelined across the wavefronts of a workgroup: the band below waits on
// an LDS progress word and reads the bottom row of the band above from the output tiles
// (workgroup-scope release/acquire, same CU).  Output goes to the macroblock-tiled workspace
// (wp_device.h): one 16-byte luma row + 8-byte chroma row per lane, 384 contiguous bytes per group.
#include "vp8_recon_band.h"

namespace wpk {

// workgroup shape and occupancy target: waves per workgroup, workgroups per CU (=> waves per SIMD = product / 4)
#ifndef WP_RECON_WAVES
#define WP_RECON_WAVES 8
#endif
#ifndef WP_RECON_WGS_PER_CU
#define WP_RECON_WGS_PER_CU 2
#endif
constexpr int kReconWaves = WP_RECON_WAVES;
constexpr int kReconWGsPerCU = WP_RECON_WGS_PER_CU;

__global__ __launch_bounds__(kReconWaves * kWave, (kReconWaves * kReconWGsPerCU + 3) / 4)
void vp8_recon_kernel(int n_frames, const wp_vp8_frame* __restrict__ frames, const uint4* __restrict__ mb,
                      const int16_t* __restrict__ coeff, uint8_t* Rp) {
    __shared__ ReconWaveLds lds_all[kReconWaves];
    __shared__ uint32_t progress[kReconWaves];
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < kReconWaves) progress[threadIdx.x] = 0;
    __syncthreads();

    uint32_t seq = 1;                   // running band number of this workgroup (identical in all its waves)
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const wp_vp8_frame fr = frames[f];
        const int n_bands = (fr.mbh + kBandRows - 1) / kBandRows;
        for (int band = 0; band < n_bands; ++band) {
            const uint32_t q = seq + uint32_t(band);
            if (int(q % kReconWaves) != wave) continue;
            recon_band<kReconWaves>(fr, band, n_bands, q, q - 1, wave, progress, lds_all[wave], mb, coeff, Rp, false);
        }
        seq += uint32_t(n_bands);
    }
}

}  // namespace wpk

#ifdef WP_PROFILE
extern "C" int wp_debug_profile(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(wpk::wp_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(wpk::wp_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

extern "C" int wpk_launch_vp8_recon(int n, int cus, const wp_vp8_frame* d_frames, const void* d_mb, const int16_t* d_coeff,
                                    uint8_t* d_rtiles, hipStream_t stream) {
    const int grid = n < cus * wpk::kReconWGsPerCU ? n : cus * wpk::kReconWGsPerCU;
    hipLaunchKernelGGL(wpk::vp8_recon_kernel, dim3(grid), dim3(wpk::kReconWaves * wpk::kWave), 0, stream,
                       n, d_frames, reinterpret_cast<const uint4*>(d_mb), d_coeff, d_rtiles);
    return int(hipGetLastError());
}
