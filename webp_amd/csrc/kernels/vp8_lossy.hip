// Reconstruction and loop filter of a batch of frames as ONE kernel, macroblock by macroblock (vp8_mb_band.h): packed
// macroblock records + coefficient blocks + filter records in, S tiles (the filtered frame in shifted macroblock tiles,
// wp_device.h) out.  The reconstructed, unfiltered pixels stay in LDS; the workspace's R region only carries the 32-byte
// edge records that bands of a frame exchange.  Persistent workgroups, one frame at a time per workgroup, the bands of
// the frame pipelined over its wavefronts (LDS progress words); frames are handed out in index order, grid-strided.
// (Running the finishing pass inside this kernel too -- each wavefront streaming out the tile rows of the band it has
// just filtered -- was measured and dropped: 9.7 ms per 1024 x 1080p against 9.2 ms for the two launches; a single
// wavefront cannot keep enough loads in flight, and the call it takes cost the step loop registers.)
#include <cstdlib>
#include "vp8_mb_band.h"

namespace wpk {

#ifndef WP_MB_WAVES
#define WP_MB_WAVES 8
#endif
#ifndef WP_MB_WGS_PER_CU
#define WP_MB_WGS_PER_CU 2
#endif
constexpr int kMbWaves = WP_MB_WAVES;
constexpr int kMbWGsPerCU = WP_MB_WGS_PER_CU;

// Register budget: 104 VGPRs.  Two workgroups of this kernel then leave a SIMD 512 - 4 x 104 = 96 registers per lane: three
// waves of the finishing pass (32 each; the LDS left beside this kernel holds three of its workgroups per CU), which is what
// a pipelined step runs beside it -- its issue priorities are tuned for that (vp8_finish.hip).  The register allocator only
// sees that LDS caps this kernel at 4 waves per SIMD anyway and would use up to 128 freely: at 105..112 two finishing waves
// fit, at 113..128 one, and the step gets longer although this kernel alone gets faster (round 3, integer edges, 111 -> 115
// registers: 6.38 -> 6.94 ms).  (The attribute counts in units of two registers on gfx90a and later.)
// History: with the integer loop-filter edges the kernel needed 112 and spilt below; the fp32 edges on denormals need 92, and
// the 12 registers between are spent on per-lane addresses that are no longer re-derived in every step (vp8_mb_band.h).
#ifndef WP_MB_NUM_VGPR
#define WP_MB_NUM_VGPR 52
#endif
// W wavefronts per workgroup pipeline the bands of one frame: 8, two workgroups per CU.  Four-wave workgroups (four per CU,
// WP_MB_FORCE_WAVES=4) were tried for batches with fewer frames than workgroup slots, where frames would spread more evenly
// over the CUs (cfg 4: 300 x 720p put two eight-wave workgroups on 44 of the 256 CUs): slower, 2.11 against 1.94 ms -- a
// frame's twelve bands then take three rounds of four instead of two of eight.
template <int W>
__global__ __attribute__((amdgpu_num_vgpr(WP_MB_NUM_VGPR))) __launch_bounds__(W * kWave, (kMbWaves * kMbWGsPerCU + 3) / 4)
void vp8_mb_kernel(int n_frames, const wp_vp8_frame* __restrict__ frames, int use_turns, const uint4* __restrict__ mb,
                   const int16_t* __restrict__ coeff, const uint32_t* __restrict__ filt, uint8_t* Ep, uint8_t* Sp, uint32_t* started) {
    __shared__ MbWaveLds lds_all[W];
    __shared__ uint32_t progress[W];
    const int wave = threadIdx.x >> 6;
    // "this workgroup has been placed": what a pipelining caller's other stream waits for before it lets a streaming kernel
    // onto the CUs (pipeline.cpp; a count of workgroups, not a time)
    if (started != nullptr && threadIdx.x == 0) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < W) progress[threadIdx.x] = 0;
    __syncthreads();

    uint32_t seq = 1;                   // running band number of this workgroup (identical in all its waves)
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        // slot f works on the frame the host's schedule names (wp_vp8_frame.turn: heavy frames paired with light ones)
        const wp_vp8_frame fr = frames[use_turns ? frames[f].turn : uint32_t(f)];
        const int n_bands = (fr.mbh + kBandRows - 1) / kBandRows;
        for (int band = 0; band < n_bands; ++band) {
            const uint32_t q = seq + uint32_t(band);
            if (int(q % W) != wave) continue;
            mb_band<W>(fr, band, n_bands, q, q - 1, wave, progress, lds_all[wave], mb, coeff, filt, Ep, Sp);
        }
        seq += uint32_t(n_bands);
    }
}

}  // namespace wpk

extern "C" int wpk_launch_vp8_recon_filter(int n, int cus, const wp_vp8_frame* d_frames, int use_turns, const void* d_mb, const int16_t* d_coeff,
                                           const uint32_t* d_filt, uint8_t* d_rtiles, uint8_t* d_stiles, hipStream_t stream,
                                           uint32_t* d_started, uint32_t* workgroups) {
    static const int force_w = std::getenv("WP_MB_FORCE_WAVES") ? std::atoi(std::getenv("WP_MB_FORCE_WAVES")) : 0;      // tuning switch
    const bool small = force_w == 4;
    if (small) {        // four-wave workgroups, four per CU: measured on cfg 4 (300 x 720p, fewer frames than workgroup slots), not the default
        const int slots = cus * wpk::kMbWGsPerCU * 2;
        if (workgroups) *workgroups += uint32_t(n < slots ? n : slots);
        hipLaunchKernelGGL(wpk::vp8_mb_kernel<wpk::kMbWaves / 2>, dim3(n < slots ? n : slots), dim3(wpk::kMbWaves / 2 * wpk::kWave), 0, stream,
                           n, d_frames, use_turns, reinterpret_cast<const uint4*>(d_mb), d_coeff, d_filt, d_rtiles, d_stiles, d_started);
    } else {
        const int grid = n < cus * wpk::kMbWGsPerCU ? n : cus * wpk::kMbWGsPerCU;
        if (workgroups) *workgroups += uint32_t(grid);
        hipLaunchKernelGGL(wpk::vp8_mb_kernel<wpk::kMbWaves>, dim3(grid), dim3(wpk::kMbWaves * wpk::kWave), 0, stream,
                           n, d_frames, use_turns, reinterpret_cast<const uint4*>(d_mb), d_coeff, d_filt, d_rtiles, d_stiles, d_started);
    }
    return int(hipGetLastError());
}

#ifdef WP_PROFILE
extern "C" int wp_debug_profile_mb(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(wpk::wp_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(wpk::wp_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

extern "C" int wpk_mb_wgs_per_cu(void) { return wpk::kMbWGsPerCU; }

// Tuning aid: resident workgroups per CU the runtime computes for the kernel (registers, LDS).
extern "C" int wp_debug_mb_occupancy(void) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wpk::vp8_mb_kernel<wpk::kMbWaves>, wpk::kMbWaves * wpk::kWave, 0) != hipSuccess) return -1;
    return n;
}
