"""ctypes binding for the CPU oracle (oracle/webp_oracle.c). TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libwebp_oracle.so")

def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])

_lib = None
def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.wo_last_error.restype = ctypes.c_char_p
        L.wo_decode.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.wo_free.argtypes = [ctypes.c_void_p]
        for f in ("wo_kind", "wo_width", "wo_height", "wo_nframes"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.wo_frame.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.wo_frame.restype = ctypes.c_void_p
        L.wo_plane.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.wo_plane.restype = ctypes.c_void_p
        L.wo_info.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.wo_to_argb.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.wo_set_keep_unfiltered.argtypes = [ctypes.c_int]
        _lib = L
    return _lib

class OracleError(Exception):
    pass

def _arr(ptr, n):
    if not ptr or n == 0:
        return None
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()

def _image(L, h):
    kind = L.wo_kind(h)
    d = {"kind": ("yuv420", "argb", "anim")[kind], "width": L.wo_width(h), "height": L.wo_height(h)}
    if kind == 0:
        ys, cs, mbw, mbh = (L.wo_info(h, i) for i in range(4))
        d.update(ystride=ys, uvstride=cs, mbw=mbw, mbh=mbh)
        d["y"] = _arr(L.wo_plane(h, 0), ys * 16 * mbh)
        d["u"] = _arr(L.wo_plane(h, 1), cs * 8 * mbh)
        d["v"] = _arr(L.wo_plane(h, 2), cs * 8 * mbh)
        d["a"] = _arr(L.wo_plane(h, 3), L.wo_info(h, 5))
        d["y0"] = _arr(L.wo_plane(h, 5), ys * 16 * mbh)
        d["u0"] = _arr(L.wo_plane(h, 6), cs * 8 * mbh)
        d["v0"] = _arr(L.wo_plane(h, 7), cs * 8 * mbh)
    elif kind == 1:
        d["stride"] = L.wo_info(h, 4)
        d["argb"] = _arr(L.wo_plane(h, 4), L.wo_info(h, 6))
    if kind != 2:
        out = np.empty(4 * d["width"] * d["height"], dtype=np.uint8)
        L.wo_to_argb(h, out.ctypes.data)
        d["to_argb"] = out
        d.update(x=L.wo_info(h, 7), y_off=L.wo_info(h, 8), duration=L.wo_info(h, 9),
                 blend=L.wo_info(h, 10), dispose=L.wo_info(h, 11))
    else:
        d["frames"] = [_image(L, L.wo_frame(h, i)) for i in range(L.wo_nframes(h))]
    return d

def decode(data: bytes, keep_unfiltered=False):
    """WebPDecoder.decode + Tools.toArgb (per frame for Anim, as test/Visual.hx does)."""
    L = lib()
    L.wo_set_keep_unfiltered(1 if keep_unfiltered else 0)
    h = ctypes.c_void_p()
    rc = L.wo_decode(data, len(data), ctypes.byref(h))
    if rc != 0:
        raise OracleError("%d: %s" % (rc, L.wo_last_error().decode()))
    try:
        return _image(L, h)
    finally:
        L.wo_free(h)

def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None

def yuv420_to_argb(y, ystride, u, v, uvstride, a, width, height):
    out = np.empty(4 * width * height, dtype=np.uint8)
    lib().wo_yuv420_to_argb(_p(y), ystride, _p(u), _p(v), uvstride, _p(a), width, height, _p(out))
    return out

def vp8_from_packed(mbw, mbh, mb, coeff, filt, filter_type):
    """Reconstruction + loop filter from the packed buffers of include/webp_amd.h -> dict of filtered and unfiltered planes."""
    n = mbw * mbh
    out = {k: np.zeros(256 * n if k[0] == "y" else 64 * n, np.uint8) for k in ("y", "u", "v", "y0", "u0", "v0")}
    mb = np.ascontiguousarray(mb, np.uint32); coeff = np.ascontiguousarray(coeff, np.int16); filt = np.ascontiguousarray(filt, np.uint32)
    rc = lib().wo_vp8_from_packed(mbw, mbh, _p(mb), _p(coeff), _p(filt), filter_type,
                                  _p(out["y"]), _p(out["u"]), _p(out["v"]), _p(out["y0"]), _p(out["u0"]), _p(out["v0"]))
    assert rc == 0
    return out


def inverse_predictor(old_width, bits, tpix, pix, h):
    pix = pix.copy(); lib().wo_inverse_predictor(old_width, bits, _p(tpix), _p(pix), h); return pix
def inverse_cross_color(old_width, bits, tpix, pix, h):
    pix = pix.copy(); lib().wo_inverse_cross_color(old_width, bits, _p(tpix), _p(pix), h); return pix
def inverse_subtract_green(pix):
    pix = pix.copy(); lib().wo_inverse_subtract_green(_p(pix), pix.size); return pix
def inverse_color_indexing(old_width, bits, tpix, pix, h):
    dst = np.zeros(pix.size if bits == 0 else 4 * old_width * h, dtype=np.uint8)
    lib().wo_inverse_color_indexing(old_width, bits, _p(tpix), _p(pix), pix.size, _p(dst), h); return dst
def unfilter_alpha(alpha, stride, filt):
    a = alpha.copy(); lib().wo_unfilter_alpha(_p(a), a.size, stride, filt); return a

def resolve_tokens(w, h, cc_bits, tokens):
    """decodePix's pixel loop (Vp8LDecoder.hx:254-333) over a token stream: the RGBA bytes decodePix returns."""
    tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
    pix = np.zeros(4 * w * h, np.uint8)
    L = lib()
    L.wo_resolve_tokens.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    rc = L.wo_resolve_tokens(w, h, cc_bits, _p(tokens), tokens.size, _p(pix))
    if rc:
        raise OracleError(L.wo_last_error().decode())
    return pix

def anim_compose(frames, canvas_w, canvas_h):
    """Canvas compositing (WebP container spec, libwebp WebPAnimDecoder arithmetic): frames = [(x, y, width, height, blend,
    dispose, argb_bytes)], x / y in the file's units of 2 pixels; returns uint8 [n, canvas_h, canvas_w, 4] (A,R,G,B)."""
    n = len(frames)
    desc = np.array([[f[0], f[1], f[2], f[3], int(f[4]), int(f[5])] for f in frames], np.int32).reshape(-1)
    bufs = [np.ascontiguousarray(f[6], dtype=np.uint8) for f in frames]
    ptrs = (ctypes.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    out = np.zeros((n, canvas_h, canvas_w, 4), np.uint8)
    L = lib()
    L.wo_anim_compose.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.wo_anim_compose(n, _p(desc), canvas_w, canvas_h, ptrs, _p(out))
    if rc:
        raise OracleError(L.wo_last_error().decode())
    return out
