/* webpgpu.hdll -- HashLink primitives over libwebp_amd.so (include/webp_amd.h).
 *
 * Build where HashLink is installed (hl.h is not in the MI355X build image, so this file is not compiled by
 * __graft_entry__.build(); tests/c_caller/c_caller.c is the C-side caller that IS built and run):
 *     gcc -O2 -shared -fPIC -I<hashlink>/src -I../../include -o webpgpu.hdll webpgpu_hl.c -lhl -L../../webp_amd -lwebp_amd
 * Every primitive returns the library's int status (0 ok, < 0 wp_status); the Haxe side (bindings/haxe/webp/gpu/Gpu.hx)
 * turns a failure into `throw lastError()`, the reference's error convention.  Buffers are hl.Bytes owned by the
 * Haxe GC; the host-pointer entry points copy in and out before they return, so nothing is retained. */
#define HL_NAME(n) webpgpu_##n
#include <hl.h>
#include "webp_amd.h"

HL_PRIM int HL_NAME(init)(int device) { return wp_init(device); }
HL_PRIM int HL_NAME(shutdown)(void) { return wp_shutdown(); }
HL_PRIM int HL_NAME(device_count)(void) { return wp_device_count(); }
HL_PRIM int HL_NAME(last_error)(vbyte* buf, int cap) { return wp_last_error((char*)buf, cap); }

/* descriptors are packed by the Haxe side exactly as the C structs (little-endian, natural alignment):
 * wp_vp8_frame = 96 bytes, wp_vp8l_frame = 168 bytes, wp_alpha_frame = 32 bytes */
HL_PRIM int HL_NAME(vp8_decode_to_argb)(int n, vbyte* frames, vbyte* mb, int n_mb, vbyte* filt, vbyte* coeff, int n_blocks,
                                        vbyte* y, vbyte* u, vbyte* v, vbyte* alpha, vbyte* argb) {
    return wp_vp8_decode_to_argb(n, (const wp_vp8_frame*)frames, (const uint32_t*)mb, (size_t)n_mb, (const uint32_t*)filt,
                                 (const int16_t*)coeff, (size_t)n_blocks, y, u, v, alpha, argb);
}
HL_PRIM int HL_NAME(vp8_recon)(int n, vbyte* frames, vbyte* mb, int n_mb, vbyte* coeff, int n_blocks, vbyte* y, vbyte* u, vbyte* v) {
    return wp_vp8_recon(n, (const wp_vp8_frame*)frames, (const uint32_t*)mb, (size_t)n_mb, (const int16_t*)coeff, (size_t)n_blocks, y, u, v);
}
HL_PRIM int HL_NAME(vp8_loopfilter)(int n, vbyte* frames, vbyte* filt, int n_mb, vbyte* y, vbyte* u, vbyte* v) {
    return wp_vp8_loopfilter(n, (const wp_vp8_frame*)frames, (const uint32_t*)filt, (size_t)n_mb, y, u, v);
}
HL_PRIM int HL_NAME(yuv420_to_argb)(int n, vbyte* frames, vbyte* y, vbyte* u, vbyte* v, vbyte* alpha, vbyte* argb) {
    return wp_yuv420_to_argb(n, (const wp_vp8_frame*)frames, y, u, v, alpha, argb);
}
HL_PRIM int HL_NAME(vp8l_inverse)(int n, vbyte* frames, vbyte* pix, int pix_len, vbyte* aux, int aux_len, vbyte* out, int out_len, int swizzle) {
    return wp_vp8l_inverse(n, (const wp_vp8l_frame*)frames, pix, (size_t)pix_len, aux, (size_t)aux_len, out, (size_t)out_len, swizzle);
}
HL_PRIM int HL_NAME(alpha_unfilter)(int n, vbyte* frames, vbyte* src, int src_len, vbyte* alpha, int alpha_len) {
    return wp_alpha_unfilter(n, (const wp_alpha_frame*)frames, src, (size_t)src_len, alpha, (size_t)alpha_len);
}

/* decodePix's pixel loop (Vp8LDecoder.hx:283-333) over the token stream a `-D webp_gpu` Vp8LDecoder records instead of
 * acting on its Huffman reads: wp_vp8l_tokens_frame = 40 bytes, tokens = 32-bit words */
HL_PRIM int HL_NAME(vp8l_resolve)(int n, vbyte* frames, vbyte* tokens, int n_words, vbyte* pix, int pix_len) {
    return wp_vp8l_resolve(n, (const wp_vp8l_tokens_frame*)frames, (const uint32_t*)tokens, (size_t)n_words, pix, (size_t)pix_len);
}

/* canvas compositing (not in the reference, which composes nothing): wp_anim_frame = 32 bytes */
HL_PRIM int HL_NAME(anim_compose)(int n, vbyte* frames, int canvas_w, int canvas_h, vbyte* argb, int argb_len, vbyte* canvases, int canvases_len) {
    return wp_anim_compose(n, (const wp_anim_frame*)frames, canvas_w, canvas_h, argb, (size_t)argb_len, canvases, (size_t)canvases_len);
}

DEFINE_PRIM(_I32, init, _I32);
DEFINE_PRIM(_I32, shutdown, _NO_ARG);
DEFINE_PRIM(_I32, device_count, _NO_ARG);
DEFINE_PRIM(_I32, last_error, _BYTES _I32);
DEFINE_PRIM(_I32, vp8_decode_to_argb, _I32 _BYTES _BYTES _I32 _BYTES _BYTES _I32 _BYTES _BYTES _BYTES _BYTES _BYTES);
DEFINE_PRIM(_I32, vp8_recon, _I32 _BYTES _BYTES _I32 _BYTES _I32 _BYTES _BYTES _BYTES);
DEFINE_PRIM(_I32, vp8_loopfilter, _I32 _BYTES _BYTES _I32 _BYTES _BYTES _BYTES);
DEFINE_PRIM(_I32, yuv420_to_argb, _I32 _BYTES _BYTES _BYTES _BYTES _BYTES _BYTES);
DEFINE_PRIM(_I32, vp8l_inverse, _I32 _BYTES _BYTES _I32 _BYTES _I32 _BYTES _I32 _I32);
DEFINE_PRIM(_I32, alpha_unfilter, _I32 _BYTES _BYTES _I32 _BYTES _I32);
DEFINE_PRIM(_I32, vp8l_resolve, _I32 _BYTES _BYTES _I32 _BYTES _I32);
DEFINE_PRIM(_I32, anim_compose, _I32 _BYTES _I32 _I32 _BYTES _I32 _BYTES _I32);
