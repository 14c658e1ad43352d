/*
 * webp_amd.h -- C ABI of the MI355X (gfx950) WebP reconstruction backend.
 *
 * This is the drop-in boundary for the hot path of fourst4r/webp (pure Haxe).  The reference has
 * no FFI of its own; the functions below are what a HashLink `.hdll` / hxcpp `@:native` extern
 * would bind.  Each entry point names the reference code it replaces
 * (paths relative to the reference's src/webp/).
 *
 * Conventions
 *   - plain `extern "C"`, no exceptions across the boundary; every function returns 0 on success
 *     and a negative wp_status on failure; wp_last_error() returns the message (the Haxe wrapper
 *     re-throws it as a String, as the reference does).
 *   - `*_dev` functions take DEVICE pointers and a hipStream_t (as void*, NULL = default stream);
 *     they enqueue work and return without synchronising.  The non-`_dev` variants take HOST
 *     pointers, stage through the library's own buffers and return when the result is in host
 *     memory; the library never keeps a caller pointer after returning.
 *   - all batched functions treat their `n` items as independent (a bad descriptor fails the call
 *     before anything is launched).
 */
#ifndef WEBP_AMD_H
#define WEBP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum wp_status {
    WP_OK = 0,
    WP_ERR_INVALID = -1,     /* bad argument / descriptor */
    WP_ERR_HIP = -2,         /* HIP runtime error (message holds hipGetErrorString) */
    WP_ERR_FORMAT = -3,      /* malformed stream: the reference's `throw "<String>"` */
    WP_ERR_EOF = -4,         /* haxe.io.Eof */
    WP_ERR_NULL = -5,        /* Vp8Decoder.decodeFrame returned null */
    WP_ERR_NOMEM = -6,
    WP_ERR_UNSUPPORTED = -7
} wp_status;

/* ------------------------------------------------------------------------------------------ */
/* Packed buffer formats (host entropy decode -> device)                                       */
/* ------------------------------------------------------------------------------------------ */

/* One lossy frame.  Planes are macroblock-padded exactly as Vp8Decoder.ensureImg allocates them
 * (vp8/Vp8Decoder.hx:186-201): ystride = 16*mbw, uvstride = 8*mbw, 16*mbh / 8*mbh rows. */
typedef struct wp_vp8_frame {
    int32_t width, height;      /* FrameHeader.width / height (visible size) */
    int32_t mbw, mbh;           /* (width+15)>>4, (height+15)>>4 */
    int32_t filter_type;        /* 0: filterHeader.level == 0, 1: simple, 2: normal (Vp8Decoder.hx:365-371) */
    int32_t has_alpha;          /* alpha_off is valid */
    uint32_t mb_base;           /* index of this frame's first record in the MB / filter record arrays */
    uint32_t stile_base;        /* index of this frame's first shifted tile in the workspace; set by wp_vp8_plan */
    uint64_t coeff_base;        /* index (in 32-byte blocks) of this frame's first coefficient block */
    uint64_t y_off, u_off, v_off;   /* byte offsets of the frame's planes inside the Y/U/V slabs */
    uint64_t argb_off;          /* byte offset inside the ARGB slab; tight stride 4*width */
    uint64_t alpha_off;         /* byte offset inside the alpha slab; tight stride width */
    uint32_t cost;              /* in: relative cost of the frame, any unit (0 = unknown: the macroblock count is used);
                                 * wp_vp8_batch_layout fills it from the parse (macroblocks, stored blocks, i4 macroblocks) */
    uint32_t turn;              /* set by wp_vp8_plan / wp_vp8_batch_layout: SCHEDULE, not data.  The one-kernel path hands frames
                                 * to its persistent workgroups by slot (slot s -> workgroup s mod grid); slot i works on frame
                                 * `turn` of its launch (index relative to the launch's first frame), a permutation that pairs
                                 * heavy frames with light ones.  No counterpart in the reference (one image per call). */
    uint32_t coeff_blocks;      /* coefficient blocks this frame owns, counted from coeff_base (valid when WP_FRAME_COEFF_BOUNDED is set) */
    uint32_t flags;             /* WP_FRAME_*; 0 from callers that build descriptors as before */
} wp_vp8_frame;
/* wp_vp8_frame.flags.  WP_FRAME_COEFF_BOUNDED: coeff_blocks is valid and the device bounds every coefficient fetch of the
 * frame by it -- a block index w[3] (+ block number) at or beyond coeff_blocks reads zeros through the hardware range
 * check of the buffer descriptor, never memory outside the frame's own blocks.  This is how the *_dev entry points, which
 * cannot inspect device-resident records, stay inside the slab on malformed records (the host-pointer entry points check
 * the indices on the host and always bound the frames).  wp_vp8_batch_layout sets it.  Without the flag the device trusts
 * w[3] (a 2 GB window from the frame's first block, as in rounds 1-2).  No counterpart in the reference, whose coeff
 * array is per macroblock. */
#define WP_FRAME_COEFF_BOUNDED 1u

/* Per-macroblock record, 4 x uint32 (what reconstruct()'s parse half produces, Vp8Decoder.hx:772-808):
 *   w[0], w[1]  i4 MBs: predY4[j][i] as 4-bit fields, block n = 4*j+i at bits 4*(n&7) of w[n>>3]
 *               i16 MBs: w[0] bits 0..3 = predY16
 *   w[2]        bits 0..23  block n has a stored coefficient block (0..15 luma, 16..19 Cb, 20..23 Cr)
 *               bit 24      a WHT (Y2) block is stored (i16 only)      bit 25  usePredY16
 *               bits 26..27 predC8
 *   w[3]        index of the MB's first coefficient block relative to wp_vp8_frame.coeff_base
 * Coefficient blocks are 16 x int16 in the natural 4x4 order of Vp8Decoder.coeff (dequantised,
 * sign-truncated to 16 bits, Vp8Decoder.hx:627-631); only blocks with a non-zero value are stored,
 * in the order WHT, luma 0..15, Cb 0..3, Cr 0..3.  For i16 MBs the luma blocks carry DC = 0; the
 * device computes the WHT in int32. */
#define WP_MB_WORDS 4
#define WP_MB_I16_BIT 25
#define WP_MB_WHT_BIT 24
#define WP_MB_C8_SHIFT 26

/* Per-macroblock loop-filter record (Filter.hx:5-10 FilterParam, final state after the whole frame
 * has been parsed -- SURVEY Q1):  level | ilevel << 8 | hlevel << 16 | inner << 24. */

/* One VP8L transform, in BITSTREAM order (vp8l/Transform.hx:13-19). */
typedef struct wp_vp8l_transform {
    int32_t type;               /* 0 Predictor, 1 CrossColor, 2 SubtractGreen, 3 ColorIndexing */
    int32_t bits;
    int32_t old_width;
    int32_t reserved;
    uint64_t data_off;          /* byte offset of Transform.pix (tile image / expanded palette) in the aux slab */
    uint64_t data_len;
} wp_vp8l_transform;

#define WP_VP8L_MAX_TRANSFORMS 4
typedef struct wp_vp8l_frame {
    int32_t width, height;      /* final image size (originalW x h, Vp8LDecoder.hx:403-408) */
    int32_t coded_width;        /* width of the decodePix output (after colour-indexing packing) */
    int32_t n_transforms;
    wp_vp8l_transform t[WP_VP8L_MAX_TRANSFORMS];
    uint64_t pix_off;           /* byte offset of the decodePix output (RGBA, 4*coded_width*height) in the pix slab */
    uint64_t out_off;           /* byte offset of the result (4*width*height) in the out slab */
    uint64_t tmp_off;           /* byte offset of a 4*width*height scratch area in the tmp slab */
} wp_vp8l_frame;

/* Token stream of one VP8L image: what decodePix's Huffman reads yield (Vp8LDecoder.hx:254-281, 291-306, 317-319) before
 * its pixel loop acts on them (:283-290 literal store, :308-316 LZ77 copy, :320-333 colour-cache fill + lookup).  A stream
 * is a sequence of 32-bit words; the top two bits of a token's first word give its kind, the low 30 its argument:
 *   WP_TOK_LITERALS  n   followed by n words, one pixel each, bytes R,G,B,A in memory order (what decodePix stores)
 *   WP_TOK_COPY      len followed by one word: the distance in pixels, already mapped through distanceMap (:50-59)
 *   WP_TOK_CACHE     key the colour-cache index read from the bitstream
 * The host emits it (wp_parse_ex with WP_PARSE_VP8L_TOKENS) and checks everything decodePix checks (copy source and end
 * inside the image, cache index in range): those checks depend on positions only, never on pixel values. */
#define WP_TOK_LITERALS 0u
#define WP_TOK_COPY 1u
#define WP_TOK_CACHE 2u
typedef struct wp_vp8l_tokens_frame {
    int32_t coded_width, height; /* size of the decodePix output */
    int32_t cc_bits;            /* colour-cache bits (Vp8LDecoder.hx:226-234): 0 = no cache, else 1..11 */
    int32_t reserved;
    uint64_t tok_off;           /* byte offset of the stream in the token slab (multiple of 4) */
    uint64_t n_words;           /* its length in 32-bit words */
    uint64_t pix_off;           /* byte offset of the result (RGBA, 4*coded_width*height) in the pix slab (multiple of 16) */
} wp_vp8l_tokens_frame;

/* One alpha plane (WebPDecoder.hx:357-411). */
typedef struct wp_alpha_frame {
    int32_t width, height;      /* alphaStride = width */
    int32_t filter;             /* 0 none, 1 horizontal, 2 vertical, 3 gradient */
    int32_t src_is_vp8l_rgba;   /* 1: source is a VP8L RGBA image, take byte 4*i+1 (WebPDecoder.hx:344-349) */
    uint64_t src_off;           /* byte offset in the source slab */
    uint64_t dst_off;           /* byte offset in the alpha slab */
} wp_alpha_frame;

/* One frame of an animation for canvas compositing.  The reference stops at independent frames
 * (Image.hx:21-34 AnimFrameHeader, WebPDecoder.hx:150-177 readAnmf: x, y, width, height, duration, blend, dispose are
 * parsed and handed to the caller, nothing is composed); the rules applied here are the WebP container specification's
 * ("Assembling the Canvas from Frames") in the integer arithmetic of libwebp's WebPAnimDecoder, non-premultiplied. */
typedef struct wp_anim_frame {
    int32_t x, y;               /* AnimFrameHeader.x / .y as stored in the file: units of 2 pixels (offset = 2x, 2y) */
    int32_t width, height;      /* AnimFrameHeader.width / .height */
    int32_t blend;              /* AnimFrameHeader.blend: 1 = alpha-blend over the canvas, 0 = overwrite the rectangle */
    int32_t dispose;            /* AnimFrameHeader.dispose: 1 = clear the rectangle to transparent before the NEXT frame */
    uint64_t argb_off;          /* byte offset of the frame's pixels (A,R,G,B, tight stride 4*width) in the frame slab */
} wp_anim_frame;

/* ------------------------------------------------------------------------------------------ */
/* Lifecycle                                                                                    */
/* ------------------------------------------------------------------------------------------ */
int wp_init(int device);                 /* selects the HIP device for the calling thread; fails if none */
int wp_shutdown(void);                   /* destroys the streams / events wp_init created (all devices); the library re-initialises on demand */
int wp_last_error(char* buf, int cap);   /* copies the calling thread's last error message */
int wp_device_count(void);
/* Test hook (no counterpart in the reference): caps the number of frames the wavefront-scheduled lossy kernels take per
 * launch, so that the split the library makes for batches beyond its band-sequence range (2^19 bands per persistent
 * workgroup, i.e. > 500 000 frames) can be exercised with a small batch.  0 restores the default. */
int wp_debug_set_frames_per_launch(int frames);
/* Test / tuning hook (no counterpart in the reference): the LDS ring of wp_vp8l_resolve_dev's kernel -- 4096, 8192 or 16384
 * pixels; 0 restores the choice by batch size.  Results are identical for every ring (copies that reach further back than
 * the ring re-read the output slab). */
int wp_debug_set_resolve_ring(int pixels);
/* Tuning hint (no counterpart in the reference): will wp_vp8_finish_dev / wp_vp8_finish_range_dev share the GPU with another
 * kernel of the caller (a pipelining caller runs them beside the next batch's wp_vp8_recon_filter_dev: the default, 1), or
 * run on their own (0)?  Selects how much work a workgroup of the pass takes; results are identical either way.
 * wp_vp8_decode_to_argb_dev, whose finishing pass follows its own kernel on one stream, always uses the second form. */
int wp_hint_finish_overlap(int on);

/* ------------------------------------------------------------------------------------------ */
/* Device-resident batch entry points                                                           */
/* ------------------------------------------------------------------------------------------ */

/* The lossy kernels exchange pixels through a macroblock-tiled device workspace (384 contiguous bytes
 * per macroblock, so that the wavefront-scheduled kernels only ever touch whole cache lines):
 *   R tiles  one per MB record slot (indexed by mb_base): reconstruction output, loop-filter input;
 *   S tiles  (mbw+1) x (mbh+1) per frame (indexed by stile_base): loop-filter output, shifted by 8 luma
 *            / 4 chroma pixels so that every byte is final when it is written.
 * wp_vp8_plan fills stile_base in the (host) descriptors and returns the workspace size for the batch;
 * call it before copying the descriptors to the device.  The workspace must be 256-byte aligned. */
int wp_vp8_plan(int n, wp_vp8_frame* frames, uint64_t* work_bytes);   /* also fills `turn` (from `cost`) */
/* Per-item validity of a descriptor array (no counterpart in the reference, which decodes one image per call: there "a bad
 * image fails alone" is automatic).  status[i] = WP_OK or WP_ERR_INVALID; returns the number of bad items.  The batched
 * entry points refuse a batch that contains a bad descriptor before anything is launched; a caller that wants the good
 * items decoded drops the bad ones first. */
int wp_vp8_check_frames(int n, const wp_vp8_frame* frames, int* status);

/* Intra prediction + inverse WHT/DCT for n frames.
 * Replaces prepareYBR / inverseWHT16 / inverseDCT4* / reconstructMacroblock / copy
 * (vp8/Vp8Decoder.hx:541-577, 716-770, 810-836, 905-996) and vp8/PredFunc.hx.
 * frames_host: host copy of the descriptors (used for launch geometry and bounds checks); d_frames: same
 * array on the device.  d_mb: WP_MB_WORDS uint32 per MB; d_coeff: int16 blocks.  Output: R tiles in the
 * workspace; if d_y/d_u/d_v are non-NULL the unfiltered planes are also written row-major. */
int wp_vp8_recon_dev(int n, const wp_vp8_frame* frames_host, const wp_vp8_frame* d_frames,
                     const uint32_t* d_mb, const int16_t* d_coeff, void* d_work, size_t work_bytes,
                     uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, void* stream);

/* Loop filter for n frames.  Replaces simpleFilter / normalFilter (vp8/Vp8Decoder.hx:417-486) and
 * vp8/Filter.hx.  d_filt: one filter record per MB (same indexing as d_mb).
 * d_y/d_u/d_v NULL: filters the R tiles left in the workspace by wp_vp8_recon_dev into S tiles (all frames; frames with
 * filter_type == 0 are re-tiled unchanged).
 * d_y/d_u/d_v given: filters those row-major planes in place (tile -> filter -> untile). */
int wp_vp8_loopfilter_dev(int n, const wp_vp8_frame* frames_host, const wp_vp8_frame* d_frames,
                          const uint32_t* d_filt, void* d_work, size_t work_bytes,
                          uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, void* stream);

/* wp_vp8_recon_dev followed by wp_vp8_loopfilter_dev (workspace variants) as ONE kernel that reconstructs and
 * filters macroblock by macroblock: the reconstructed pixels stay on chip (no R tiles are written or read back;
 * the R region of the workspace only carries 32-byte edge records between bands and holds no tiles afterwards).
 * Same results, same S tiles afterwards. */
int wp_vp8_recon_filter_dev(int n, const wp_vp8_frame* frames_host, const wp_vp8_frame* d_frames,
                            const uint32_t* d_mb, const uint32_t* d_filt, const int16_t* d_coeff,
                            void* d_work, size_t work_bytes, void* stream);

/* Finishing pass: workspace tiles -> row-major MB-padded planes (ImageData.Yuv420, Vp8Decoder.ensureImg
 * geometry; optional) and tightly packed A,R,G,B (Tools.toArgb, Tools.hx:6-48; optional).  Reads the S tiles, which
 * wp_vp8_loopfilter_dev and wp_vp8_recon_filter_dev write for every frame of the batch (frames whose loop filter is off
 * are re-tiled unchanged). */
int wp_vp8_finish_dev(int n, const wp_vp8_frame* frames_host, const wp_vp8_frame* d_frames,
                      const void* d_work, size_t work_bytes, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                      const uint8_t* d_alpha, uint8_t* d_argb, void* stream);

/* The same for frames [first, first + count) of the batch only (the descriptors and the workspace are
 * still those of the whole batch).  Lets a caller spread the memory-bound finishing pass of one batch
 * over the reconstruction and the loop filter of the next ("pipelining steps"). */
int wp_vp8_finish_range_dev(int n, const wp_vp8_frame* frames_host, const wp_vp8_frame* d_frames,
                            int first, int count, const void* d_work, size_t work_bytes,
                            uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                            const uint8_t* d_alpha, uint8_t* d_argb, void* stream);

/* YUV420 (row-major, MB-padded) -> tightly packed A,R,G,B.  Replaces Tools.toArgb (Tools.hx:6-48).
 * d_alpha may be NULL (all frames opaque); frames with has_alpha == 0 get A = 0xFF. */
int wp_yuv420_to_argb_dev(int n, const wp_vp8_frame* frames_host, const wp_vp8_frame* d_frames,
                          const uint8_t* d_y, const uint8_t* d_u, const uint8_t* d_v,
                          const uint8_t* d_alpha, uint8_t* d_argb, void* stream);

/* recon -> loop filter -> finishing pass without leaving the device.  Replaces Vp8Decoder.decodeFrame's
 * pixel half + Tools.toArgb.  d_y/d_u/d_v (the filtered planes, what ImageData.Yuv420 carries) and
 * d_argb are each optional. */
int wp_vp8_decode_to_argb_dev(int n, const wp_vp8_frame* frames_host, const wp_vp8_frame* d_frames,
                              const uint32_t* d_mb, const uint32_t* d_filt, const int16_t* d_coeff,
                              void* d_work, size_t work_bytes,
                              uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                              const uint8_t* d_alpha, uint8_t* d_argb, void* stream);

/* wp_vp8_decode_to_argb_dev for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc) and therefore cannot bring a workspace.  No counterpart in the reference.  The
 * R / S tiles live in a static device array of WP_STATIC_WORK_BYTES that every GPU the library runs on holds from the
 * moment the library is loaded until the process ends; the call allocates and frees nothing, on the device or through
 * any library.  stile_base in the descriptors must come from wp_vp8_plan as usual; a batch whose workspace (what
 * wp_vp8_plan reports: ~768 bytes per macroblock, 6.4 MB per 1080p frame) exceeds WP_STATIC_WORK_BYTES is refused with
 * WP_ERR_UNSUPPORTED.  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams,
 * two threads) race -- serialise them.  Results are identical to wp_vp8_decode_to_argb_dev's. */
#define WP_STATIC_WORK_BYTES (64u << 20)
int wp_vp8_decode_to_argb_static_dev(int n, const wp_vp8_frame* frames_host, const wp_vp8_frame* d_frames,
                                     const uint32_t* d_mb, const uint32_t* d_filt, const int16_t* d_coeff,
                                     uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                                     const uint8_t* d_alpha, uint8_t* d_argb, void* stream);

/* Software pipeline over successive lossy batches (no counterpart in the reference, which decodes one image per call): the
 * memory-bound finishing pass of batch k runs on a second stream of the pipeline beside the issue-bound reconstruction +
 * loop filter kernel of batch k+1, which is worth a quarter of a step (8.4 -> 6.4 ms per 1024 x 1080p).  The
 * pipeline owns two HIP streams of default priority, `n_workspaces` (2..8) tile workspaces of `work_bytes` each (what
 * wp_vp8_plan reports for the largest batch that will be submitted) and the hold-off that dispatches the persistent kernel
 * ahead of the streaming one -- everything a caller would otherwise rebuild from INTEGRATION.md prose.
 *   create   on the calling thread's current device.  flags: WP_PIPELINE_TIMING keeps HIP timing events around both kernels
 *            of every step (wp_vp8_pipeline_stats).
 *   submit   enqueues the kernel of this batch and, beside it, the finishing pass of the batch submitted before; returns
 *            without waiting (it blocks only while three earlier steps are still incomplete).  Arguments as for
 *            wp_vp8_decode_to_argb_dev, minus the workspace; the descriptors must have been through wp_vp8_plan /
 *            wp_vp8_batch_layout.  frames_host is copied; every device buffer must stay valid, and its contents unchanged,
 *            until the step is complete.  inputs_ready: a hipEvent_t the kernel waits for (the caller's upload), or NULL.
 *            alpha_ready: a hipEvent_t only the finishing pass waits for, or NULL -- the alpha planes are not needed before
 *            it, so a caller can produce them (wp_vp8l_inverse_dev + wp_alpha_unfilter_dev) on a stream of its own beside
 *            the reconstruction + loop filter kernel; both events must stay alive until the step is complete.
 *            A batch's outputs are complete once a LATER flush / drain has returned (or the event logic of flush has run).
 *   flush    enqueues the finishing pass of the last batch and makes `stream` (hipStream_t; NULL = default stream) wait
 *            for everything submitted so far, without blocking the host.
 *   drain    the same, and blocks until everything submitted so far is complete.
 *   destroy  drains and frees.
 * One thread at a time per pipeline. */
typedef struct wp_vp8_pipeline wp_vp8_pipeline;
#define WP_PIPELINE_TIMING 1
int wp_vp8_pipeline_create(int n_workspaces, size_t work_bytes, int flags, wp_vp8_pipeline** out);
int wp_vp8_pipeline_submit(wp_vp8_pipeline* p, int n, const wp_vp8_frame* frames_host, const wp_vp8_frame* d_frames,
                           const uint32_t* d_mb, const uint32_t* d_filt, const int16_t* d_coeff,
                           uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, const uint8_t* d_alpha, uint8_t* d_argb,
                           void* inputs_ready, void* alpha_ready);
int wp_vp8_pipeline_flush(wp_vp8_pipeline* p, void* stream);
int wp_vp8_pipeline_drain(wp_vp8_pipeline* p);
/* Accumulated duration (ms) of the reconstruction + loop filter launches and of the finishing launches of the steps that
 * have completed, each measured with HIP events on the stream the kernel runs on, and their number. */
int wp_vp8_pipeline_stats(wp_vp8_pipeline* p, double* ms_recon_filter, double* ms_finish, long* steps, int reset);
int wp_vp8_pipeline_set_hold_us(wp_vp8_pipeline* p, int us);      /* tuning: least time (us) the finishing pass is held back, on top of the wait for the persistent kernel to be placed; default 0 */
int wp_vp8_pipeline_destroy(wp_vp8_pipeline* p);

/* VP8L inverse transform chain + RGBA->ARGB swizzle.  Replaces Transform.hx inversePredictor /
 * inverseCrossColor / inverseSubtractGreen / inverseColorIndexing, the driver loop
 * Vp8LDecoder.hx:395-401 and the swizzle WebPDecoder.hx:109-123.
 * d_pix is consumed (transforms run in place where the reference does).  `swizzle` selects what the
 * last pass writes at out_off: WP_SWIZZLE_NONE R,G,B,A as decoded; WP_SWIZZLE_ARGB the rotated
 * A,R,G,B of WebPDecoder.hx:109-123; WP_SWIZZLE_GREEN one byte per pixel, the green channel, which
 * is all readAlpha keeps of a VP8L-coded ALPH image (WebPDecoder.hx:330-336) -- feed it to
 * wp_alpha_unfilter with src_is_vp8l_rgba = 0, or use it as the alpha plane directly when the
 * ALPH filter is 0. */
#define WP_SWIZZLE_NONE 0
#define WP_SWIZZLE_ARGB 1
#define WP_SWIZZLE_GREEN 2
int wp_vp8l_inverse_dev(int n, const wp_vp8l_frame* frames_host, const wp_vp8l_frame* d_frames,
                        uint8_t* d_pix, const uint8_t* d_aux, uint8_t* d_tmp, uint8_t* d_out,
                        int swizzle, void* stream);

/* The pixel loop of decodePix on the device (first slice): literal stores, LZ77 copies and the lazily
 * filled colour cache (Vp8LDecoder.hx:283-290, 308-316, 320-333) for n token streams, one wavefront per image.  Writes at
 * pix_off exactly the bytes decodePix returns, i.e. what wp_vp8l_inverse_dev consumes as d_pix.  frames_host gives the
 * launch geometry; the streams themselves are device memory the library cannot inspect: they must come from
 * wp_parse_ex(WP_PARSE_VP8L_TOKENS) or have passed wp_vp8l_check_tokens.  (The kernel still never reads or writes outside
 * a frame's stream and pixel extents: it stops at the first token that would.) */
int wp_vp8l_resolve_dev(int n, const wp_vp8l_tokens_frame* frames_host, const wp_vp8l_tokens_frame* d_frames,
                        const uint32_t* d_tokens, uint8_t* d_pix, void* stream);
/* Host-side validity of one stream against its descriptor: WP_OK, or WP_ERR_FORMAT with decodePix's message
 * ("Invalid LZ77 parameters", "Invalid color cache index", "null color cache") or a note on what is malformed. */
int wp_vp8l_check_tokens(const wp_vp8l_tokens_frame* frame, const uint32_t* tokens /* the frame's stream, host memory */);

/* Canvas compositing of one animation: canvas k (k = 0..n-1, each 4*canvas_w*canvas_h bytes, A,R,G,B, laid one after the
 * other in d_canvases) is what a viewer shows after frame k.  No counterpart in the reference (see wp_anim_frame).  Frame
 * rectangles are clipped to the canvas; the background is transparent black (the ANIM chunk's colour is ignored, as
 * libwebp's decoder does). */
int wp_anim_compose_dev(int n, const wp_anim_frame* frames_host, const wp_anim_frame* d_frames, int canvas_w, int canvas_h,
                        const uint8_t* d_argb, uint8_t* d_canvases, void* stream);

/* Alpha plane: optional green-channel extraction + unfilterAlpha (WebPDecoder.hx:311-411). */
int wp_alpha_unfilter_dev(int n, const wp_alpha_frame* frames_host, const wp_alpha_frame* d_frames,
                          const uint8_t* d_src, uint8_t* d_alpha, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Host-pointer entry points (what the Haxe externs call)                                      */
/* ------------------------------------------------------------------------------------------ */
/* Same semantics as the _dev variants; all pointers are host memory.  Slab sizes are taken from
 * the descriptors (max offset + extent). out_y/out_u/out_v receive the filtered planes and may be
 * NULL when only ARGB is wanted; out_argb may be NULL when only YUV is wanted.  The workspace is
 * planned and allocated internally (stile_base in the caller's descriptors is ignored). */
int wp_vp8_decode_to_argb(int n, const wp_vp8_frame* frames,
                          const uint32_t* mb, size_t n_mb, const uint32_t* filt,
                          const int16_t* coeff, size_t n_coeff_blocks,
                          uint8_t* out_y, uint8_t* out_u, uint8_t* out_v,
                          const uint8_t* alpha, uint8_t* out_argb);
/* The two halves on their own.  wp_vp8_recon: intra prediction + inverse WHT/DCT only -- the
 * UNFILTERED planes, i.e. Y/Cb/Cr as Vp8Decoder.decodeFrame holds them before simpleFilter / normalFilter run
 * (Vp8Decoder.hx:345-354).  wp_vp8_loopfilter: simpleFilter / normalFilter (Vp8Decoder.hx:417-486) in place on row-major,
 * MB-padded planes; filt holds one record per MB, indexed by mb_base like the MB records. */
int wp_vp8_recon(int n, const wp_vp8_frame* frames, const uint32_t* mb, size_t n_mb, const int16_t* coeff, size_t n_coeff_blocks,
                 uint8_t* out_y, uint8_t* out_u, uint8_t* out_v);
int wp_vp8_loopfilter(int n, const wp_vp8_frame* frames, const uint32_t* filt, size_t n_mb, uint8_t* y, uint8_t* u, uint8_t* v);
int wp_yuv420_to_argb(int n, const wp_vp8_frame* frames, const uint8_t* y, const uint8_t* u, const uint8_t* v,
                      const uint8_t* alpha, uint8_t* out_argb);
int wp_vp8l_inverse(int n, const wp_vp8l_frame* frames, const uint8_t* pix, size_t pix_len,
                    const uint8_t* aux, size_t aux_len, uint8_t* out, size_t out_len, int swizzle);
int wp_alpha_unfilter(int n, const wp_alpha_frame* frames, const uint8_t* src, size_t src_len,
                      uint8_t* alpha, size_t alpha_len);
int wp_anim_compose(int n, const wp_anim_frame* frames, int canvas_w, int canvas_h, const uint8_t* argb, size_t argb_len,
                    uint8_t* canvases, size_t canvases_len);
/* wp_vp8l_resolve_dev on host memory; every stream is checked (wp_vp8l_check_tokens) before anything is uploaded. */
int wp_vp8l_resolve(int n, const wp_vp8l_tokens_frame* frames, const uint32_t* tokens, size_t n_words,
                    uint8_t* pix, size_t pix_len);

/* ------------------------------------------------------------------------------------------ */
/* Host side of the path (C++ mirror of the reference's Haxe host code; the reference toolchain  */
/* is absent in the build image).  RIFF/VP8/VP8L parsing and entropy decode on the CPU.         */
/* ------------------------------------------------------------------------------------------ */
typedef struct wp_parsed wp_parsed;      /* opaque: result of parsing one .webp file */

/* Mirrors WebPDecoder.decode's container walk (WebPDecoder.hx:179-309) and the parse half of
 * Vp8Decoder / Vp8LDecoder; produces the packed buffers above for every frame in the file. */
int wp_parse(const uint8_t* data, size_t len, wp_parsed** out);
/* flags: WP_PARSE_VP8L_TOKENS -- the top-level image of every VP8L still / ANMF frame is left as a token stream
 * (wp_parsed_frame_buf which = 6, colour-cache bits in wp_parsed_frame_vp8l_meta[3]) instead of being resolved into
 * pixels on the host (which = 3 is then empty); transform data and VP8L-coded alpha images are resolved as before. */
#define WP_PARSE_VP8L_TOKENS 1
int wp_parse_ex(const uint8_t* data, size_t len, int flags, wp_parsed** out);
void wp_parsed_free(wp_parsed* p);
/* n independent files on up to n_threads host threads (0 = one per hardware thread).  The reference is single-threaded
 *  but keeps all mutable state in its decoder objects, so one decoder per host thread is what a Haxe caller
 * would do with `sys.thread`; this is that loop in C++ (the host entropy decode caps end-to-end
 * throughput long before the kernels do).  out[i] receives the parsed object or NULL, status[i] its wp_status;
 * returns 0 if every file parsed, else the first failing status (wp_last_error holds that file's message). */
int wp_parse_batch(int n, const uint8_t* const* data, const size_t* len, int n_threads, wp_parsed** out, int* status);
int wp_parse_batch_ex(int n, const uint8_t* const* data, const size_t* len, int n_threads, int flags, wp_parsed** out, int* status);
/* The same loop as wp_parse_batch, but running in the background so that a caller can consume the files in index order
 * while later ones are still being parsed (one work queue for the whole set: no thread idles while another finishes a
 * large file, whatever the size of the chunks the caller consumes).  At most `window` files beyond the last index waited
 * for are started (0 = no limit): that bounds the memory held by parsed files nobody has consumed yet, and lets the
 * allocator reuse what the consumer has freed.  data / len / out / status must stay valid until
 * wp_parse_job_finish.  wp_parse_job_wait blocks until files [0, upto) are done (parsed or failed) and returns 0, or the
 * first failing status among them.  wp_parse_job_finish stops handing out files, joins the threads and frees the job
 * (files not yet started stay unparsed; parsed objects in out[] stay the caller's, to be released with wp_parsed_free). */
typedef struct wp_parse_job wp_parse_job;
int wp_parse_job_start(int n, const uint8_t* const* data, const size_t* len, int n_threads, int window, wp_parsed** out,
                       int* status, wp_parse_job** job);
int wp_parse_job_wait(wp_parse_job* job, int upto);
int wp_parse_job_finish(wp_parse_job* job);

/* Batch assembly for the lossy path: what `reconstruct`'s prologue and the per-MB capture of the filter parameters
 * leave behind for one frame (Vp8Decoder.hx:772-808, 348-353), laid side by side for n frames so that one
 * device call handles them.  files[i] / frame_index[i] (NULL = frame 0 everywhere) name lossy frames of parsed files.
 * wp_vp8_batch_layout fills frames[i] (sizes, filter type, has_alpha = with_alpha[i] (NULL = none), mb_base / coeff_base
 * and output offsets rounded up to 256 bytes, in the order given; stile_base as wp_vp8_plan would) and reports the slab
 * sizes; wp_vp8_batch_gather then copies the records into the caller's slabs (host memory, pinned if they are to be
 * uploaded asynchronously; capacities in MB records / coefficient blocks are checked) on up to n_threads threads. */
typedef struct wp_vp8_batch_sizes {
    uint64_t n_mb, n_coeff_blocks;                  /* MB / filter records, coefficient blocks (32 bytes each) */
    uint64_t y_bytes, u_bytes, v_bytes, argb_bytes, alpha_bytes, work_bytes;
} wp_vp8_batch_sizes;
int wp_vp8_batch_layout(int n, const wp_parsed* const* files, const int* frame_index, const int* with_alpha,
                        wp_vp8_frame* frames, wp_vp8_batch_sizes* sizes);
int wp_vp8_batch_gather(int n, const wp_parsed* const* files, const int* frame_index, const wp_vp8_frame* frames,
                        uint32_t* mb, uint32_t* filt, size_t n_mb_capacity, int16_t* coeff, size_t n_coeff_blocks_capacity,
                        int n_threads);
/* kind: 0 Yuv420, 1 Argb, 2 Anim (Image.hx:36-46) */
int wp_parsed_kind(const wp_parsed* p);
int wp_parsed_width(const wp_parsed* p);
int wp_parsed_height(const wp_parsed* p);
int wp_parsed_nframes(const wp_parsed* p);     /* 1 for stills */
/* per frame i: type 0 lossy / 1 lossless; sizes; pointers into the parsed object's own storage */
int wp_parsed_frame_info(const wp_parsed* p, int i, int32_t* info /* [16] */);
const void* wp_parsed_frame_buf(const wp_parsed* p, int i, int which, size_t* len);
/* VP8L metadata of frame i (and of its VP8L-coded alpha image): sizes and the transform list, 32 ints */
int wp_parsed_frame_vp8l_meta(const wp_parsed* p, int i, int32_t* meta /* [32] */);
/* webp.Image.FrameHeader of a lossy frame (Image.hx:10-19), 8 ints */
int wp_parsed_frame_vp8_header(const wp_parsed* p, int i, int32_t* hdr /* [8] */);

#ifdef __cplusplus
}
#endif
#endif /* WEBP_AMD_H */
