// Reconstruction and loop filter of a batch of frames in ONE launch: every persistent workgroup first reconstructs its
// frames (vp8_recon_band.h), then filters the same frames (vp8_filter_band.h).  No grid-wide dependency exists between
// the two -- a frame's loop filter needs that frame's R tiles only -- so workgroups whose frames were cheap to
// reconstruct start filtering while the ones holding an expensive frame are still reconstructing: the latency-bound
// tail of one phase overlaps the issue-bound other phase ("kernel times versus batch size"): 1024 x 1080p
// 5.95 ms against 3.5 + 2.8 ms for the two launches.
#include "vp8_recon_band.h"
#include "vp8_filter_band.h"

namespace wpk {

constexpr int kLossyWaves = 8;
constexpr int kLossyWGsPerCU = 2;

__global__ __launch_bounds__(kLossyWaves * kWave, (kLossyWaves * kLossyWGsPerCU + 3) / 4)
void vp8_recon_filter_kernel(int n_frames, const wp_vp8_frame* __restrict__ frames, const uint4* __restrict__ mb,
                             const int16_t* __restrict__ coeff, const uint32_t* __restrict__ filt, uint8_t* Rp, uint8_t* Sp) {
    constexpr size_t kLdsBytes = sizeof(FilterWaveLds) > sizeof(ReconWaveLds) ? sizeof(FilterWaveLds) : sizeof(ReconWaveLds);
    __shared__ __attribute__((aligned(16))) uint8_t lds_raw[kLossyWaves][kLdsBytes];      // one phase at a time uses it
    ReconWaveLds& lds_recon = *reinterpret_cast<ReconWaveLds*>(lds_raw[threadIdx.x >> 6]);
    FilterWaveLds& lds_filter = *reinterpret_cast<FilterWaveLds*>(lds_raw[threadIdx.x >> 6]);
    __shared__ uint32_t progress[kLossyWaves];
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < kLossyWaves) progress[threadIdx.x] = 0;
    __syncthreads();

    uint32_t seq = 1;                   // running band number of this workgroup, across both phases (progress words only grow)
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const wp_vp8_frame fr = frames[f];
        const int n_bands = (fr.mbh + kBandRows - 1) / kBandRows;
        for (int band = 0; band < n_bands; ++band) {
            const uint32_t q = seq + uint32_t(band);
            if (int(q % kLossyWaves) != wave) continue;
            recon_band<kLossyWaves>(fr, band, n_bands, q, q - 1, wave, progress, lds_recon, mb, coeff, Rp, false);
        }
        seq += uint32_t(n_bands);
    }
    // this workgroup's R tiles are complete and visible to all of its wavefronts; the LDS changes hands
    __threadfence_block();
    __syncthreads();
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const wp_vp8_frame fr = frames[f];
        const int n_bands = (fr.mbh + kBandRows - 1) / kBandRows;
        if (fr.filter_type == 0) { seq += uint32_t(n_bands); continue; }      // filterHeader.level == 0 (Vp8Decoder.hx:365)
        for (int band = 0; band < n_bands; ++band) {
            const uint32_t q = seq + uint32_t(band);
            if (int(q % kLossyWaves) != wave) continue;
            if (fr.filter_type == 1) filter_band<kLossyWaves, true>(fr, band, n_bands, q, q - 1, wave, progress, lds_filter, filt, Rp, Sp);
            else filter_band<kLossyWaves, false>(fr, band, n_bands, q, q - 1, wave, progress, lds_filter, filt, Rp, Sp);
        }
        seq += uint32_t(n_bands);
    }
}

}  // namespace wpk

extern "C" int wpk_launch_vp8_recon_filter(int n, int cus, const wp_vp8_frame* d_frames, const void* d_mb, const int16_t* d_coeff,
                                           const uint32_t* d_filt, uint8_t* d_rtiles, uint8_t* d_stiles, hipStream_t stream) {
    const int grid = n < cus * wpk::kLossyWGsPerCU ? n : cus * wpk::kLossyWGsPerCU;
    hipLaunchKernelGGL(wpk::vp8_recon_filter_kernel, dim3(grid), dim3(wpk::kLossyWaves * wpk::kWave), 0, stream,
                       n, d_frames, reinterpret_cast<const uint4*>(d_mb), d_coeff, d_filt, d_rtiles, d_stiles);
    return int(hipGetLastError());
}
