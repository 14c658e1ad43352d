"""ctypes binding of libwebp_amd.so (the C ABI in include/webp_amd.h).  There is no CPU fallback:
if the library is missing, or no HIP device is present when a compute entry point is called,
the call fails loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("WEBP_AMD_LIB") or os.path.join(_HERE, "libwebp_amd.so")   # override: tuning variants only

c_u8p = ctypes.POINTER(ctypes.c_uint8)


class Vp8Frame(ctypes.Structure):          # wp_vp8_frame
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("mbw", ctypes.c_int32), ("mbh", ctypes.c_int32),
                ("filter_type", ctypes.c_int32), ("has_alpha", ctypes.c_int32), ("mb_base", ctypes.c_uint32),
                ("stile_base", ctypes.c_uint32), ("coeff_base", ctypes.c_uint64), ("y_off", ctypes.c_uint64),
                ("u_off", ctypes.c_uint64), ("v_off", ctypes.c_uint64), ("argb_off", ctypes.c_uint64),
                ("alpha_off", ctypes.c_uint64), ("cost", ctypes.c_uint32), ("turn", ctypes.c_uint32),
                ("coeff_blocks", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


FRAME_COEFF_BOUNDED = 1                    # WP_FRAME_COEFF_BOUNDED


class Vp8lTransform(ctypes.Structure):     # wp_vp8l_transform
    _fields_ = [("type", ctypes.c_int32), ("bits", ctypes.c_int32), ("old_width", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("data_off", ctypes.c_uint64), ("data_len", ctypes.c_uint64)]


class Vp8lFrame(ctypes.Structure):         # wp_vp8l_frame
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("coded_width", ctypes.c_int32),
                ("n_transforms", ctypes.c_int32), ("t", Vp8lTransform * 4), ("pix_off", ctypes.c_uint64),
                ("out_off", ctypes.c_uint64), ("tmp_off", ctypes.c_uint64)]


class AlphaFrame(ctypes.Structure):        # wp_alpha_frame
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("filter", ctypes.c_int32),
                ("src_is_vp8l_rgba", ctypes.c_int32), ("src_off", ctypes.c_uint64), ("dst_off", ctypes.c_uint64)]


class Vp8lTokensFrame(ctypes.Structure):   # wp_vp8l_tokens_frame
    _fields_ = [("coded_width", ctypes.c_int32), ("height", ctypes.c_int32), ("cc_bits", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("tok_off", ctypes.c_uint64), ("n_words", ctypes.c_uint64), ("pix_off", ctypes.c_uint64)]


class AnimFrame(ctypes.Structure):         # wp_anim_frame
    _fields_ = [("x", ctypes.c_int32), ("y", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("blend", ctypes.c_int32), ("dispose", ctypes.c_int32), ("argb_off", ctypes.c_uint64)]


class Vp8BatchSizes(ctypes.Structure):   # wp_vp8_batch_sizes
    _fields_ = [(k, ctypes.c_uint64) for k in ("n_mb", "n_coeff_blocks", "y_bytes", "u_bytes", "v_bytes", "argb_bytes",
                                               "alpha_bytes", "work_bytes")]


class WebPError(Exception):
    """Raised with the message the reference would `throw` (plus the native status code)."""
    def __init__(self, code, msg):
        super().__init__("%s (status %d)" % (msg, code))
        self.code = code
        self.msg = msg


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError("libwebp_amd.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C webp_amd/csrc`. There is no CPU fallback." % SO_PATH)
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64; if torch is going to be
    # used in this process (device memory, streams, torch.distributed) it has to be loaded first so that
    # this library binds to the same runtime instead of bringing /opt/rocm's copy in beside it.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(SO_PATH)
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.wp_init.argtypes = [i32]
    L.wp_debug_set_frames_per_launch.argtypes = [i32]
    L.wp_debug_set_resolve_ring.argtypes = [i32]
    L.wp_hint_finish_overlap.argtypes = [i32]
    L.wp_last_error.argtypes = [ctypes.c_char_p, i32]
    L.wp_vp8_plan.argtypes = [i32, vp, ctypes.POINTER(ctypes.c_uint64)]
    L.wp_vp8_check_frames.argtypes = [i32, vp, vp]
    L.wp_vp8_recon_dev.argtypes = [i32, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]
    L.wp_vp8_loopfilter_dev.argtypes = [i32, vp, vp, vp, vp, sz, vp, vp, vp, vp]
    L.wp_vp8_finish_dev.argtypes = [i32, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp]
    L.wp_vp8_recon_filter_dev.argtypes = [i32, vp, vp, vp, vp, vp, vp, sz, vp]
    L.wp_vp8_finish_range_dev.argtypes = [i32, vp, vp, i32, i32, vp, sz, vp, vp, vp, vp, vp, vp]
    L.wp_yuv420_to_argb_dev.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.wp_vp8_decode_to_argb_dev.argtypes = [i32, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp]
    L.wp_vp8_decode_to_argb_static_dev.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.wp_vp8_pipeline_create.argtypes = [i32, sz, i32, ctypes.POINTER(vp)]
    L.wp_vp8_pipeline_submit.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.wp_vp8_pipeline_flush.argtypes = [vp, vp]
    L.wp_vp8_pipeline_drain.argtypes = [vp]
    L.wp_vp8_pipeline_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), i32]
    L.wp_vp8_pipeline_set_hold_us.argtypes = [vp, i32]
    L.wp_vp8_pipeline_destroy.argtypes = [vp]
    L.wp_vp8l_inverse_dev.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, vp]
    L.wp_alpha_unfilter_dev.argtypes = [i32, vp, vp, vp, vp, vp]
    L.wp_vp8_decode_to_argb.argtypes = [i32, vp, vp, sz, vp, vp, sz, vp, vp, vp, vp, vp]
    L.wp_yuv420_to_argb.argtypes = [i32, vp, vp, vp, vp, vp, vp]
    L.wp_vp8_recon.argtypes = [i32, vp, vp, sz, vp, sz, vp, vp, vp]
    L.wp_vp8_loopfilter.argtypes = [i32, vp, vp, sz, vp, vp, vp]
    L.wp_vp8l_inverse.argtypes = [i32, vp, vp, sz, vp, sz, vp, sz, i32]
    L.wp_alpha_unfilter.argtypes = [i32, vp, vp, sz, vp, sz]
    L.wp_anim_compose_dev.argtypes = [i32, vp, vp, i32, i32, vp, vp, vp]
    L.wp_anim_compose.argtypes = [i32, vp, i32, i32, vp, sz, vp, sz]
    L.wp_vp8l_resolve_dev.argtypes = [i32, vp, vp, vp, vp, vp]
    L.wp_vp8l_check_tokens.argtypes = [vp, vp]
    L.wp_vp8l_resolve.argtypes = [i32, vp, vp, sz, vp, sz]
    L.wp_parse.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.wp_parse_ex.argtypes = [vp, sz, i32, ctypes.POINTER(vp)]
    L.wp_parse_batch_ex.argtypes = [i32, vp, vp, i32, i32, vp, vp]
    L.wp_parsed_free.argtypes = [vp]
    L.wp_parse_batch.argtypes = [i32, vp, vp, i32, vp, vp]
    L.wp_parsed_free.restype = None
    L.wp_parse_job_start.argtypes = [i32, vp, vp, i32, i32, vp, vp, ctypes.POINTER(vp)]
    L.wp_parse_job_wait.argtypes = [vp, i32]
    L.wp_parse_job_finish.argtypes = [vp]
    L.wp_vp8_batch_layout.argtypes = [i32, vp, vp, vp, vp, ctypes.POINTER(Vp8BatchSizes)]
    L.wp_vp8_batch_gather.argtypes = [i32, vp, vp, vp, vp, vp, sz, vp, sz, i32]
    for f in ("wp_parsed_kind", "wp_parsed_width", "wp_parsed_height", "wp_parsed_nframes"):
        getattr(L, f).argtypes = [vp]
    L.wp_parsed_frame_info.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_int32)]
    L.wp_parsed_frame_buf.argtypes = [vp, i32, i32, ctypes.POINTER(sz)]
    L.wp_parsed_frame_buf.restype = vp
    L.wp_parsed_frame_vp8l_meta.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_int32)]
    L.wp_parsed_frame_vp8_header.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_int32)]
    _lib = L
    return L


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib().wp_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise WebPError(rc, last_error())
