// Canvas compositing of an animation: what a viewer shows after each frame.  The reference parses
// AnimFrameHeader (Image.hx:21-34, WebPDecoder.hx:150-177) and stops; the rules here are the WebP container
// specification's, in the integer arithmetic of libwebp's WebPAnimDecoder (non-premultiplied blend), which is what the
// golden canvases under tests/golden/anim/ were produced with.
//
// Frames depend on each other, pixels do not: one thread owns one canvas pixel and walks the frames in order, keeping the
// pixel in a register -- every canvas is written once, coalesced, and every frame pixel is read once.
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
__global__ __launch_bounds__(256)
void anim_compose_kernel(int n, const wp_anim_frame* __restrict__ frames, int cw, int ch, const uint8_t* __restrict__ argb, uint32_t* canvases) {
    const long long total = (long long)cw * ch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int py = int(i / cw), px = int(i - (long long)py * cw);
        uint32_t cur = 0u;                                    // transparent black (the ANIM background colour is ignored)
        bool in_prev = false, prev_dispose = false, prev_full = false, prev_key = false;
        for (int k = 0; k < n; ++k) {
            const wp_anim_frame& f = frames[k];
            const bool full = f.x == 0 && f.y == 0 && f.width == cw && f.height == ch;
            const bool key = k == 0 || (!f.blend && full) || (prev_dispose && (prev_full || prev_key));
            // the frame before asked for its rectangle to be cleared; libwebp does not blend there either (nothing is under the
            // new pixel), which again shows for alpha 0 with colour and in the rounding of the blend
            const bool cleared = prev_dispose && in_prev;
            if (cleared) cur = 0u;
            const int fx = px - 2 * f.x, fy = py - 2 * f.y;
            const bool inside = fx >= 0 && fy >= 0 && fx < f.width && fy < f.height;
            if (inside) {
                const uint32_t src = *reinterpret_cast<const uint32_t*>(argb + f.argb_off + 4 * ((long long)fy * f.width + fx));
                cur = (f.blend && !key && !cleared) ? blend_non_premult(src, cur) : src;
            }
            canvases[(long long)k * total + i] = cur;
            in_prev = inside;
            prev_dispose = f.dispose != 0; prev_full = full; prev_key = key;
        }
    }
}

}  // namespace wpk

extern "C" int wpk_launch_anim_compose(int n, const wp_anim_frame* d_frames, int cw, int ch, const uint8_t* d_argb,
                                       uint8_t* d_canvases, hipStream_t stream) {
    const long long total = (long long)cw * ch;
    const int grid = int(std::min<long long>((total + 255) / 256, 1 << 20));
    hipLaunchKernelGGL(wpk::anim_compose_kernel, dim3(grid), dim3(256), 0, stream, n, d_frames, cw, ch, d_argb,
                       reinterpret_cast<uint32_t*>(d_canvases));
    return int(hipGetLastError());
}
