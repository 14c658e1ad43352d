// Per-pixel YUV -> ARGB arithmetic shared by the streaming conversion kernels (Tools.hx:25-31).
#pragma once
#include "wp_device.h"

namespace wpk {

__device__ __forceinline__ uint32_t round255(double x) {
    x = fmax(0.0, fmin(255.0, x));
    return uint32_t(int(floor(__dadd_rn(x, 0.5))));
}

// one pixel -> bytes A,R,G,B in memory order (little-endian dword: A | R<<8 | G<<16 | B<<24)
__device__ __forceinline__ uint32_t convert_px(int yv, int cb, int cr, uint32_t a) {
    const double Yd = double(yv), Cb = double(cb - 128), Cr = double(cr - 128);
    const double R = __dadd_rn(Yd, __dmul_rn(1.402, Cr));
    const double G = __dsub_rn(__dsub_rn(Yd, __dmul_rn(0.344136, Cb)), __dmul_rn(0.714136, Cr));
    const double B = __dadd_rn(Yd, __dmul_rn(1.772, Cb));
    return a | round255(R) << 8 | round255(G) << 16 | round255(B) << 24;
}

// four horizontally adjacent pixels: luma dword, two chroma samples each, four alpha bytes -> 16 bytes of A,R,G,B
__device__ __forceinline__ uint4 convert_px4(uint32_t y4, uint32_t u2, uint32_t v2, uint32_t a4) {
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        o[k] = convert_px(int((y4 >> (8 * k)) & 0xff), int((u2 >> (8 * (k >> 1))) & 0xff), int((v2 >> (8 * (k >> 1))) & 0xff), (a4 >> (8 * k)) & 0xff);
    return make_uint4(o[0], o[1], o[2], o[3]);
}

}  // namespace wpk
