// Micro-benchmark: cost of row-strided partial stores on MI355X.  Each wavefront owns a band of 64 rows of a
// 1920-byte-pitch plane and walks left to right, writing CHUNK bytes per row per visit (16, 32 or 64),
// the way the wavefront-scheduled kernels emit macroblock rows.  Total bytes written are identical.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CHUNK>
__global__ __launch_bounds__(512) void k(uint8_t* base, int pitch, int rows_per_img, int nimg) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int bands = rows_per_img / 64;
    const int img = wave / bands, band = wave % bands;
    if (img >= nimg) return;
    uint8_t* row = base + (size_t(img) * rows_per_img + band * 64 + lane) * pitch;
    for (int x = 0; x < pitch; x += CHUNK) {
        const uint4 v = make_uint4(x, lane, img, band);
#pragma unroll
        for (int k2 = 0; k2 < CHUNK / 16; ++k2) *reinterpret_cast<uint4*>(row + x + 16 * k2) = v;
        __builtin_amdgcn_s_sleep(20);      // spread the visits out in time like a compute step would
    }
}
int main() {
    const int pitch = 1920, rows = 1088 + 64 * 0, nimg = 1024;   // 17 bands of 64 rows
    const int bands = rows / 64;
    size_t bytes = size_t(nimg) * rows * pitch;
    uint8_t* d; hipMalloc(&d, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int waves = nimg * bands, blocks = (waves * 64 + 511) / 512;
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(512), 0, 0, d, pitch, rows, nimg); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("chunk16: %.3f ms  %.1f GB/s\n", ms, bytes / ms / 1e6);
        hipEventRecord(e0); hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(512), 0, 0, d, pitch, rows, nimg); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("chunk32: %.3f ms  %.1f GB/s\n", ms, bytes / ms / 1e6);
        hipEventRecord(e0); hipLaunchKernelGGL(k<64>, dim3(blocks), dim3(512), 0, 0, d, pitch, rows, nimg); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("chunk64: %.3f ms  %.1f GB/s\n", ms, bytes / ms / 1e6);
    }
    return 0;
}
