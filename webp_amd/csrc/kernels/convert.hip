// YUV420 (macroblock-padded planes) -> tightly packed A,R,G,B.  Replaces Tools.toArgb (Tools.hx:6-48).
//
// The arithmetic (an integer form that is bit-identical to the reference's fp64 expressions for every
// (Y, Cb, Cr) triple) is in wp_convert.h.  Pure streaming: each thread converts 4 horizontally adjacent
// pixels (one dword of luma, two chroma samples) and stores 16 bytes.
#include "wp_device.h"
#include "wp_convert.h"

namespace wpk {

__global__ __launch_bounds__(256)
void yuv420_to_argb_kernel(const wp_vp8_frame* __restrict__ frames, const uint8_t* __restrict__ Yp, const uint8_t* __restrict__ Up,
                           const uint8_t* __restrict__ Vp, const uint8_t* __restrict__ Ap, uint8_t* __restrict__ out) {
    const wp_vp8_frame fr = frames[blockIdx.y];
    const int width = fr.width, height = fr.height;
    const int ystride = 16 * fr.mbw, cstride = 8 * fr.mbw;
    const int quads = (width + 3) >> 2;
    const long long total = (long long)quads * height;
    const uint8_t* Y = Yp + fr.y_off;
    const uint8_t* U = Up + fr.u_off;
    const uint8_t* V = Vp + fr.v_off;
    const uint8_t* A = (Ap != nullptr && fr.has_alpha) ? Ap + fr.alpha_off : nullptr;
    uint8_t* dst = out + fr.argb_off;

    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = int(i / quads);
        const int col = int(i - (long long)row * quads) * 4;
        const uint32_t y4 = *reinterpret_cast<const uint32_t*>(Y + size_t(row) * ystride + col);     // planes are padded to 16
        const uint32_t u2 = *reinterpret_cast<const uint16_t*>(U + size_t(row >> 1) * cstride + (col >> 1));
        const uint32_t v2 = *reinterpret_cast<const uint16_t*>(V + size_t(row >> 1) * cstride + (col >> 1));
        const int npx = min(4, width - col);
        uint32_t a4 = 0xffffffffu;
        const size_t pix = size_t(row) * width + col;
        if (A) {
            a4 = 0;
            for (int k = 0; k < npx; ++k) a4 |= uint32_t(A[pix + k]) << (8 * k);
        }
        const uint4 o4 = convert_px4(y4, u2, v2, a4);
        const uint32_t o[4] = {o4.x, o4.y, o4.z, o4.w};
        uint8_t* p = dst + pix * 4;
        if (npx == 4 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
            *reinterpret_cast<uint4*>(p) = o4;
        } else {
            for (int k = 0; k < npx; ++k) reinterpret_cast<uint32_t*>(p)[k] = o[k];
        }
    }
}

}  // namespace wpk

extern "C" int wpk_launch_yuv420_to_argb(int n, int grid_x, const wp_vp8_frame* d_frames, const uint8_t* d_y, const uint8_t* d_u,
                                         const uint8_t* d_v, const uint8_t* d_alpha, uint8_t* d_argb, hipStream_t stream) {
    hipLaunchKernelGGL(wpk::yuv420_to_argb_kernel, dim3(grid_x, n), dim3(256), 0, stream, d_frames, d_y, d_u, d_v, d_alpha, d_argb);
    return int(hipGetLastError());
}
