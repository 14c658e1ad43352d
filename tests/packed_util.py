"""Test helper: lays random packed frames (tests/test_gpu_packed_random.make_frame) side by side as one batch with the
attributes webp_amd.device.DeviceLossyBatch expects of a LossyBatch, so that the device entry points can be driven with
inputs no encoder produces."""
import numpy as np


class RawBatch:
    def __init__(self, shapes, frames, filter_types, visible=None, bounded=True):
        """shapes: [(mbw, mbh)], frames: [(mb, coeff, filt)] as make_frame returns them, filter_types: one per frame."""
        from webp_amd._lib import Vp8Frame
        n = len(frames)
        self.n = n
        self.desc = (Vp8Frame * max(n, 1))()
        al = lambda v: (v + 255) // 256 * 256
        mb_base = cbase = yo = co = ao = 0
        for i, ((w, h), (mb, coeff, filt)) in enumerate(zip(shapes, frames)):
            d = self.desc[i]
            d.mbw, d.mbh = w, h
            d.width, d.height = visible[i] if visible else (16 * w - (i % 5 if w > 1 else 0), 16 * h - (i % 7 if h > 1 else 0))
            d.filter_type = filter_types[i]
            d.mb_base, d.coeff_base = mb_base, cbase
            if bounded:
                d.coeff_blocks, d.flags = coeff.size // 16, 1      # WP_FRAME_COEFF_BOUNDED: what wp_vp8_batch_layout writes
            d.y_off, d.u_off, d.v_off, d.argb_off = yo, co, co, ao
            mb_base += w * h
            cbase += coeff.size // 16
            yo += al(256 * w * h)
            co += al(64 * w * h)
            ao += al(4 * d.width * d.height)
        self.mb = np.concatenate([f[0] for f in frames]).astype(np.uint32)
        self.coeff = np.concatenate([f[1] for f in frames]).astype(np.int16)
        self.filt = np.concatenate([f[2] for f in frames]).astype(np.uint32)
        self.n_mb, self.n_coeff_blocks = mb_base, cbase
        self.y_bytes, self.u_bytes, self.v_bytes, self.argb_bytes, self.alpha_bytes = yo, co, co, ao, 0
        self.shapes, self.frames, self.filter_types = shapes, frames, filter_types

    def expected(self, i):
        """Oracle planes (filtered + unfiltered) and ARGB of frame i."""
        from oracle import oracle
        (w, h), (mb, coeff, filt) = self.shapes[i], self.frames[i]
        exp = oracle.vp8_from_packed(w, h, mb, coeff, filt, self.filter_types[i])
        d = self.desc[i]
        exp["argb"] = oracle.yuv420_to_argb(exp["y"], 16 * w, exp["u"], exp["v"], 8 * w, None, d.width, d.height)
        return exp


def random_batch(n, seed, extreme=False, max_w=9, max_h=6, filter_types=(0, 1, 2)):
    from test_gpu_packed_random import make_frame
    rng = np.random.default_rng(seed)
    shapes = [(int(rng.integers(1, max_w + 1)), int(rng.integers(1, max_h + 1))) for _ in range(n)]
    frames = [make_frame(rng, w, h, extreme) for (w, h) in shapes]
    ft = [int(filter_types[i % len(filter_types)]) for i in range(n)]
    return RawBatch(shapes, frames, ft)
