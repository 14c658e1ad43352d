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

// Cross-wavefront hand-over inside a workgroup (all its wavefronts run on one CU and share its L1):
// the producer finishes its global stores, executes a workgroup-scope release FENCE (all address
// spaces: s_waitcnt vmcnt(0) lgkmcnt(0)) and then stores the LDS progress word; the consumer polls the
// word and executes a workgroup-scope acquire fence before touching the data.  Fences are explicit
// because an atomic on an LDS address alone only orders LDS accesses.
__device__ __forceinline__ void wait_progress(const uint32_t* word, uint32_t need) {
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void publish_progress(uint32_t* word, uint32_t value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Forces the wait for previously issued loads to happen HERE (loads and stores share one in-order
// counter on gfx9-class hardware: waiting for a load that was issued before a batch of stores, but
// only after those stores have been issued, would also wait for the stores' acknowledgements).
__device__ __forceinline__ void settle(uint32_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void settle(uint4& v) { settle(v.x); settle(v.y); settle(v.z); settle(v.w); }
__device__ __forceinline__ void settle(uint2& v) { settle(v.x); settle(v.y); }

}  // namespace wpk
