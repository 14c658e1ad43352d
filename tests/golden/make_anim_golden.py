"""Generates tests/golden/anim/*: small animated WebP files (lossless frames, so that every decoder yields the same frame
pixels) and the canvases libwebp's WebPAnimDecoder composes from them (through PIL, in the build container: the GPU box and
the test suite only read the committed files).  The reference does no compositing (Image.hx:21-34: AnimFrameHeader is
handed to the caller), so for SURVEY 8f item 4 libwebp's decoder is the pin: same container rules, same integer blend.

Files: PIL's animation encoder chooses sub-rectangles, offsets, blend and dispose itself; further variants are made by
rewriting the blend / dispose bits of every ANMF chunk at random (any combination is a valid file).
    python tests/golden/make_anim_golden.py"""
import io
import os
import struct

import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "anim")


def frames_rgba(rng, w, h, n, style):
    out = []
    base = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    for k in range(n):
        a = base.copy() if style != 2 else rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        # moving boxes with different alpha classes: opaque, half, fully transparent WITH colour underneath
        for b in range(3):
            x0, y0 = int(rng.integers(0, w - 2)), int(rng.integers(0, h - 2))
            x1, y1 = int(rng.integers(x0 + 1, w + 1)), int(rng.integers(y0 + 1, h + 1))
            a[y0:y1, x0:x1, :3] = rng.integers(0, 256, 3, dtype=np.uint8)
            a[y0:y1, x0:x1, 3] = (255, int(rng.integers(1, 255)), 0)[b]
        if style == 1:
            a[..., 3] = np.where(rng.random((h, w)) < 0.5, a[..., 3], 255)
        if style == 3:
            a[..., 3] = 255
        base = a if style == 0 else base
        out.append(Image.fromarray(a, "RGBA"))
    return out


def anmf_chunks(data):
    """(offset of the flags byte) of every ANMF chunk of a RIFF/WEBP file."""
    pos, out = 12, []
    while pos + 8 <= len(data):
        tag, n = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        if tag == b"ANMF":
            out.append(pos + 8 + 15)
        pos += 8 + n + (n & 1)
    return out


def canvases(data):
    im = Image.open(io.BytesIO(data))
    res = []
    for k in range(im.n_frames):
        im.seek(k)
        res.append(np.array(im.convert("RGBA")))
    return np.stack(res)


def main():
    os.makedirs(HERE, exist_ok=True)
    rng = np.random.default_rng(20261111)
    k = 0
    for style, (w, h, n) in enumerate(((56, 40, 6), (33, 47, 5), (64, 64, 4), (40, 30, 5))):
        fr = frames_rgba(rng, w, h, n, style)
        buf = io.BytesIO()
        fr[0].save(buf, "WEBP", save_all=True, append_images=fr[1:], lossless=True, exact=True, duration=40,
                   minimize_size=(style != 2), allow_mixed=False)
        data = bytearray(buf.getvalue())
        variants = [bytes(data)]
        for _ in range(3):
            v = bytearray(data)
            for off in anmf_chunks(v):
                v[off] = (v[off] & ~3) | int(rng.integers(0, 4))          # bit 1: no blend, bit 0: dispose to background
            variants.append(bytes(v))
        for v in variants:
            name = "anim%02d" % k
            k += 1
            with open(os.path.join(HERE, name + ".webp"), "wb") as f:
                f.write(v)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=canvases(v))
    print("wrote %d animations to %s" % (k, HERE))


if __name__ == "__main__":
    main()
