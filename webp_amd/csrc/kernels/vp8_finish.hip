// Finishing pass of the lossy path: macroblock tiles (wp_device.h) -> row-major Y/U/V planes (what
// ImageData.Yuv420 carries, Vp8Decoder.ensureImg geometry) and tightly packed A,R,G,B (Tools.toArgb,
// Tools.hx:6-48), plus the inverse helper that tiles row-major planes for the stand-alone loop filter.
//
// Pure streaming.  A workgroup produces 256 output columns x 16 rows: it parks the 16 (or, for shifted tiles,
// 17) horizontally adjacent source tiles (6.4 KB, contiguous in memory) in LDS and writes rows: a wavefront emits
// 1 KB of contiguous ARGB (256 pixels of one row), 256 B of luma or 128 B of chroma per instruction, all starting
// on 128-byte boundaries of the row (no cache line is shared between workgroups).  Frames are read from the S tiles (shift 8 luma /
// 4 chroma: what the loop filter and the one-kernel path leave, also for frames whose loop filter is off) or, on request,
// from the R tiles (no shift: the unfiltered planes wp_vp8_recon_dev can return).
#ifndef WP_FINISH_PRIO
#define WP_FINISH_PRIO 1
#endif
// Interior chunks drop to priority 0 after their first row of pixels (WP_FINISH_LO_AT 1).  The two kernels of a pipelined
// step share the SIMDs' issue slots, their stretched durations add up to what those allow (~11 ms per 1024 x 1080p), and a
// step is as long as the longer of the two: at 1 throughout the pass takes 4.6 ms and vp8_mb_kernel 6.4; at 0 throughout 5.9 /
// 5.1; dropping after parking (LO_AT 0) 5.85 / 5.2, after the second row (2) 5.35 / 5.8; after the first row 5.44 / 5.51 ms.
#ifndef WP_FINISH_PRIO_LO
#define WP_FINISH_PRIO_LO 0
#endif
#ifndef WP_FINISH_LO_AT
#define WP_FINISH_LO_AT 1
#endif
#include "vp8_finish_chunk.h"

namespace wpk {

#ifndef WP_FINISH_PER_WG
#define WP_FINISH_PER_WG 4
#endif
constexpr int kFinishPerWgShared = WP_FINISH_PER_WG;      // chunks per workgroup when the pass shares the GPU (see the kernel)
#ifndef WP_FINISH_FAST
#define WP_FINISH_FAST 1
#endif
#ifndef WP_FINISH_PRIO
#define WP_FINISH_PRIO 1
#endif

// 32 registers: two wavefronts per SIMD fit beside four of vp8_mb_kernel (112 each); the attribute counts in units of 2
#ifndef WP_FINISH_NUM_VGPR
#define WP_FINISH_NUM_VGPR 16
#endif
template <int kFinishPerWg>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(WP_FINISH_NUM_VGPR)))
void vp8_finish_kernel(const wp_vp8_frame* __restrict__ frames, const uint8_t* __restrict__ Rp, const uint8_t* __restrict__ Sp,
                       int force_unshifted, uint8_t* Yp, uint8_t* Up, uint8_t* Vp, const uint8_t* __restrict__ Ap, uint8_t* out) {
    __shared__ uint4 tiles4[kFinishLdsBytes / 16];
#ifdef WP_FINISH_PAD_LDS      // tuning experiment: cap the pass's residency (workgroups per CU) as if another kernel held the rest of the LDS
    __shared__ uint32_t pad_lds[WP_FINISH_PAD_LDS / 4];
    if (Yp == nullptr && out == nullptr && Up == nullptr) pad_lds[threadIdx.x] = 1;
#endif
    const wp_vp8_frame& fr = frames[blockIdx.z];
    const bool shifted = !force_unshifted;
    const int t = int(threadIdx.x);
    // Issue priority 1: above the residual phase of the one-kernel reconstruction + loop filter (0), level with its
    // prediction (1), below its horizontal edges and i4 mini-steps (2).  At 0 the pass only got what that kernel left over and
    // was what a pipelined step waited for (7.07 ms per 1024 x 1080p); at 1: 6.64, at 2: 6.72, at 3: 7.03 (the persistent
    // kernel then pays more than the pass gains).  Alone on the GPU the priority changes nothing.
    __builtin_amdgcn_s_setprio(WP_FINISH_PRIO);
    const int ty = blockIdx.y, c_first = blockIdx.x * (kFinishPerWg * kFinishCols);
    const int c_end = 16 * fr.mbw;
    if (ty >= fr.mbh + (shifted ? 1 : 0) || c_first >= c_end) return;
    // Two loops, chosen per workgroup (its chunks are neighbours in one tile row, so only the one at the ragged right edge
    // differs from the rest): all chunks interior -> the predicate-free path (vp8_finish_chunk.h), else the general path for all
    // of them.  One loop that chose per chunk held both paths' state at once: 38 registers.
    // Measured and not kept: interior chunks without workgroup barriers (every wavefront fetching and parking just the 54
    // 16-byte pieces of the tiles that its own four luma rows and two chroma rows need).  The pass gets faster beside the
    // reconstruction kernel (5.63 -> 4.56 ms stretched) and slows that kernel by as much (5.69 -> 6.86 ms; at issue priority 0:
    // 5.52 / 6.26 ms): the sum of the two is what the SIMDs' issue slots allow, and a step is 6.4 - 7.0 ms against 5.8.
    const int n_here = min(kFinishPerWg, (c_end - c_first + kFinishCols - 1) / kFinishCols);
    if (WP_FINISH_FAST && finish_is_interior(fr, ty, c_first + (n_here - 1) * kFinishCols, shifted, Yp, Up, Vp, Ap, out)) {
        // an interior chunk's 9 source tiles all exist, and the next chunk's start 8 tiles further on: the source is one pointer
        const u32x4* src = reinterpret_cast<const u32x4*>(finish_src(fr, ty, c_first, Rp, Sp, true).src);
        FinishBases fb = finish_bases(fr, ty, c_first, Yp, Up, Vp, Ap, out);
        u32x4 nx = {0u, 0u, 0u, 0u};
        if (t < kFinishTiles * (kTileBytes / 16)) nx = WP_NT_LOAD(src + t);
#pragma unroll
        for (int i = 0; i < kFinishPerWg; ++i) {
            if (i >= n_here) break;
            if (i) __syncthreads();                           // the chunk before has been read out of LDS
            int tt = t;
            asm volatile("" : "+v"(tt));                      // (thread roles re-derived per chunk from an opaque copy: see below)
#if WP_FINISH_PRIO_LO != WP_FINISH_PRIO
            if (i) __builtin_amdgcn_s_setprio(WP_FINISH_PRIO);
#endif
            if (tt < kFinishTiles * (kTileBytes / 16)) tiles4[tt] = make_uint4(nx.x, nx.y, nx.z, nx.w);
            __syncthreads();
            src += (kFinishTiles - 1) * (kTileBytes / 16);
            if (i + 1 < n_here && tt < kFinishTiles * (kTileBytes / 16)) nx = WP_NT_LOAD(src + tt);
            __builtin_amdgcn_sched_barrier(0);
            finish_emit_interior(fb, tt, tiles4);
            fb.next_chunk();
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    // kFinishPerWg adjacent chunks per workgroup, the next chunk's tiles fetched into registers while the current one is
    // converted.  Built to test whether the pass, of which a CU holds only two workgroups beside the one-kernel
    // reconstruction + loop filter, is bound there by the latency of its one load per workgroup: hardly -- at issue priority 0
    // 2, 3, 4 or 8 chunks per workgroup (32 registers, same residency) give 7.07 / 7.04 / 7.10 / 6.94 ms per pipelined step
    // against 7.10 with one; at priority 1 (below) 6.60 / 6.52 / 6.48 for 2 / 4 / 8 against 6.64.  The pass alone: 2.93 - 3.14
    // ms against 2.84.  So: 4 when the caller runs the pass on its own (wp_vp8_finish[_range]_dev: what a pipelining caller
    // uses), 1 inside wp_vp8_decode_to_argb_dev, where the pass follows the persistent kernel on the same stream.
    FinishSrc fs = finish_src(fr, ty, c_first, Rp, Sp, shifted);
    u32x4 nx = finish_fetch(fs, t);
    int nx_nt = fs.nt;
#pragma unroll
    for (int i = 0; i < kFinishPerWg; ++i) {
        const int c0 = c_first + i * kFinishCols;
        if (c0 >= c_end) break;
        if (i) __syncthreads();                               // the chunk before has been read out of LDS
        const int nt = nx_nt;
        // everything derived from the thread index is re-derived per chunk from an opaque copy: shared across the unrolled
        // chunks it would sit in registers for the whole kernel, and 32 registers is the budget (two waves per SIMD beside
        // vp8_mb_kernel)
        int tt = t;
        asm volatile("" : "+v"(tt));
        if (tt < nt * (kTileBytes / 16)) tiles4[tt] = make_uint4(nx.x, nx.y, nx.z, nx.w);
        __syncthreads();
        if (i + 1 < kFinishPerWg && c0 + kFinishCols < c_end) {
            fs = finish_src(fr, ty, c0 + kFinishCols, Rp, Sp, shifted);
            nx = finish_fetch(fs, tt);
            nx_nt = fs.nt;
        }
        __builtin_amdgcn_sched_barrier(0);
        finish_emit(fr, ty, c0, tt, nt, tiles4, shifted, Yp, Up, Vp, Ap, out);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Row-major planes -> R tiles (only used by the stand-alone loop-filter entry point).
__global__ __launch_bounds__(256)
void vp8_tile_kernel(const wp_vp8_frame* __restrict__ frames, const uint8_t* __restrict__ Yp, const uint8_t* __restrict__ Up,
                     const uint8_t* __restrict__ Vp, uint8_t* Rp) {
    const wp_vp8_frame& fr = frames[blockIdx.y];
    const int mbw = fr.mbw, n_mb = fr.mbw * fr.mbh;
    const int t = threadIdx.x, j = t & 31;
    for (int m = blockIdx.x * 8 + (t >> 5); m < n_mb; m += gridDim.x * 8) {
        const int x = m % mbw, y = m / mbw;
        uint8_t* tile = Rp + (size_t(fr.mb_base) + m) * kTileBytes;
        if (j < 16) {
            *reinterpret_cast<uint4*>(tile + 16 * j) = *reinterpret_cast<const uint4*>(Yp + fr.y_off + size_t(16 * y + j) * (16 * mbw) + 16 * x);
        } else {
            const int p = (j - 16) >> 3, r = (j - 16) & 7;
            const uint8_t* P = p ? Vp + fr.v_off : Up + fr.u_off;
            *reinterpret_cast<uint2*>(tile + kTileCb + 64 * p + 8 * r) = *reinterpret_cast<const uint2*>(P + size_t(8 * y + r) * (8 * mbw) + 8 * x);
        }
    }
}

}  // namespace wpk

extern "C" int wpk_launch_vp8_finish(int n, int chunks, int tile_rows, int alone, const wp_vp8_frame* d_frames, const uint8_t* d_rtiles,
                                     const uint8_t* d_stiles, int force_unshifted, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                                     const uint8_t* d_alpha, uint8_t* d_argb, hipStream_t stream) {
    if (alone)
        hipLaunchKernelGGL(wpk::vp8_finish_kernel<1>, dim3(chunks, tile_rows, n), dim3(256), 0, stream,
                           d_frames, d_rtiles, d_stiles, force_unshifted, d_y, d_u, d_v, d_alpha, d_argb);
    else
        hipLaunchKernelGGL(wpk::vp8_finish_kernel<wpk::kFinishPerWgShared>, dim3((chunks + wpk::kFinishPerWgShared - 1) / wpk::kFinishPerWgShared, tile_rows, n),
                           dim3(256), 0, stream, d_frames, d_rtiles, d_stiles, force_unshifted, d_y, d_u, d_v, d_alpha, d_argb);
    return int(hipGetLastError());
}

extern "C" int wpk_launch_vp8_tile(int n, int grid_x, const wp_vp8_frame* d_frames, const uint8_t* d_y, const uint8_t* d_u,
                                   const uint8_t* d_v, uint8_t* d_rtiles, hipStream_t stream) {
    hipLaunchKernelGGL(wpk::vp8_tile_kernel, dim3(grid_x, n), dim3(256), 0, stream, d_frames, d_y, d_u, d_v, d_rtiles);
    return int(hipGetLastError());
}
