// Canvas compositing of an animation: what a viewer shows after each frame.  The reference parses
// AnimFrameHeader (Image.hx:21-34, WebPDecoder.hx:150-177) and stops; the rules here are the WebP container
// specification's, in the integer arithmetic of libwebp's WebPAnimDecoder (non-premultiplied blend), which is what the
// golden canvases under tests/golden/anim/ were produced with.
//
// Frames depend on each other, pixels do not: one thread owns four adjacent canvas pixels and walks the frames in order,
// keeping them in registers -- every canvas is written once, coalesced, and every frame pixel is read once.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include "../../../include/webp_amd.h"

namespace wpk {

// pixels are A,R,G,B in memory: as a little-endian dword A is the low byte
__device__ __forceinline__ uint32_t blend_non_premult(uint32_t src, uint32_t dst) {
    const uint32_t src_a = src & 0xffu;
    if (src_a == 0xffu) return src;
    if (src_a == 0u) return dst;
    const uint32_t dst_a = dst & 0xffu;
    const uint32_t dst_factor_a = (dst_a * (256u - src_a)) >> 8;       // libwebp's approximation of dst_a * (255 - src_a) / 255
    const uint32_t blend_a = src_a + dst_factor_a;
    const uint32_t scale = (1u << 24) / blend_a;
    uint32_t out = blend_a;
#pragma unroll
    for (int shift = 8; shift < 32; shift += 8) {
        const uint32_t c = (((src >> shift) & 0xffu) * src_a + ((dst >> shift) & 0xffu) * dst_factor_a) * scale >> 24;
        out |= c << shift;
    }
    return out;
}

// A key frame does not depend on the canvas before it, and libwebp does not blend it (which shows where a pixel has
// alpha 0 and colour: blended over the empty canvas it would become 0).  libwebp's rule (anim_decode.c IsKeyFrame), without
// its "frame has no alpha" shortcut -- such a frame is opaque everywhere, so blending returns its pixels unchanged and the
// only frame whose status the shortcut could change further is a full-canvas one, which the third clause treats alike.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Four horizontally adjacent canvas pixels per thread: 16-byte loads and stores wherever the four lie in one row, inside
// (or outside) the frame together and the addresses are 16-byte aligned; pixel by pixel otherwise.
__global__ __launch_bounds__(256)
void anim_compose_kernel(int n, const wp_anim_frame* __restrict__ frames, int cw, int ch, const uint8_t* __restrict__ argb, uint32_t* canvases) {
    const long long total = (long long)cw * ch;
    const int quads = (cw + 3) >> 2;
    const long long n_quads = (long long)quads * ch;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += (long long)gridDim.x * blockDim.x) {
        const int py = int(q / quads), px0 = int(q - (long long)py * quads) * 4;
        const int npx = min(4, cw - px0);
        const long long i0 = (long long)py * cw + px0;
        uint32_t cur[4] = {0u, 0u, 0u, 0u};                   // transparent black (the ANIM background colour is ignored)
        bool in_prev[4] = {false, false, false, false};
        bool prev_dispose = false, prev_full = false, prev_key = false;
        for (int k = 0; k < n; ++k) {
            const wp_anim_frame& f = frames[k];
            const bool full = f.x == 0 && f.y == 0 && f.width == cw && f.height == ch;
            const bool key = k == 0 || (!f.blend && full) || (prev_dispose && (prev_full || prev_key));
            const bool do_blend = f.blend && !key;
            const int fy = py - 2 * f.y, fx0 = px0 - 2 * f.x;
            const bool row_in = fy >= 0 && fy < f.height;
            const uint8_t* srow = argb + f.argb_off + 4 * ((long long)fy * f.width + fx0);
            uint32_t src[4] = {0u, 0u, 0u, 0u};
            const bool all_in = row_in && fx0 >= 0 && fx0 + 4 <= f.width;
            if (all_in && (reinterpret_cast<uintptr_t>(srow) & 15) == 0) {
                const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(srow));
                src[0] = v.x; src[1] = v.y; src[2] = v.z; src[3] = v.w;
            } else if (row_in) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (fx0 + j >= 0 && fx0 + j < f.width) src[j] = *reinterpret_cast<const uint32_t*>(srow + 4 * j);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool inside = row_in && fx0 + j >= 0 && fx0 + j < f.width;
                // the frame before asked for its rectangle to be cleared; libwebp does not blend there either (nothing is under
                // the new pixel), which again shows for alpha 0 with colour and in the rounding of the blend
                const bool cleared = prev_dispose && in_prev[j];
                if (cleared) cur[j] = 0u;
                if (inside) cur[j] = (do_blend && !cleared) ? blend_non_premult(src[j], cur[j]) : src[j];
                in_prev[j] = inside;
            }
            uint32_t* dst = canvases + (long long)k * total + i0;
            if (npx == 4 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                const u32x4 v = {cur[0], cur[1], cur[2], cur[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j < npx) dst[j] = cur[j];
            }
            prev_dispose = f.dispose != 0; prev_full = full; prev_key = key;
        }
    }
}

}  // namespace wpk

extern "C" int wpk_launch_anim_compose(int n, const wp_anim_frame* d_frames, int cw, int ch, const uint8_t* d_argb,
                                       uint8_t* d_canvases, hipStream_t stream) {
    const int grid = int(std::min<long long>((((long long)((cw + 3) >> 2) * ch) + 255) / 256, 1 << 20));
    hipLaunchKernelGGL(wpk::anim_compose_kernel, dim3(grid), dim3(256), 0, stream, n, d_frames, cw, ch, d_argb,
                       reinterpret_cast<uint32_t*>(d_canvases));
    return int(hipGetLastError());
}
