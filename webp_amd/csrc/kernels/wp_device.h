// Shared device-side helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/webp_amd.h"

namespace wpk {

constexpr int kWave = 64;             // CDNA wavefront
constexpr int kBandRows = 4;          // macroblock rows handled by one wavefront (16 lanes each)
constexpr int kWavesPerWG = 8;        // wavefronts per workgroup in the wavefront-scheduled kernels
constexpr uint32_t kSeqShift = 12;    // progress word = (band sequence << 12) | macroblocks finished
constexpr uint32_t kBandDone = 0xFFF;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }   // -> v_med3_i32

// Spin until the LDS progress word of another wavefront of this workgroup reaches `need`
// (workgroup-scope acquire: the producer's global stores are visible afterwards).
__device__ __forceinline__ void wait_progress(const uint32_t* word, uint32_t need) {
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void publish_progress(uint32_t* word, uint32_t value) {
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace wpk
