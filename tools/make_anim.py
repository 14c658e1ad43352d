"""Builds an animated WebP (VP8X + ANIM + n x ANMF{ALPH, VP8 }) in memory from lossy stills that carry an ALPH chunk, by
re-wrapping their chunks: no encoder involved, so it runs on the GPU box too.  Layout as WebPDecoder.readAnmf /
readFrameData read it (WebPDecoder.hx:150-177, 258-309).  Used for BASELINE.json configs[3] (300 frames of 1280x720
with alpha) by bench.py and the tests."""
import struct


def chunks(data: bytes):
    assert data[:4] == b"RIFF" and data[8:12] == b"WEBP"
    pos, out = 12, []
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        out.append((tag, data[pos + 8:pos + 8 + size]))
        pos += 8 + size + (size & 1)
    return out


def _chunk(tag: bytes, body: bytes) -> bytes:
    return tag + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")


def u24(v):
    return bytes((v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff))


def alph_with_filter(alph: bytes, filter_id: int) -> bytes:
    """The same ALPH chunk body with another prediction filter in its header byte (pre-processing | filter | compression,
    WebPDecoder.hx:78-83).  The coded data stay what they are, so the plane that comes out is no longer the encoder's --
    but it is a valid chunk, and unfilterAlpha (WebPDecoder.hx:357-411) runs the chosen filter over it."""
    return bytes(((alph[0] & ~0x0c) | ((filter_id & 3) << 2),)) + alph[1:]


def raw_alph(plane: bytes, filter_id: int) -> bytes:
    """An uncompressed ALPH chunk body (compression 0, WebPDecoder.hx:313-322): header byte + width * height bytes."""
    return bytes(((filter_id & 3) << 2,)) + bytes(plane)


def build_anim(stills, n_frames: int, duration_ms: int = 33, alph_for=None) -> bytes:
    """stills: list of .webp byte strings (VP8X stills with ALPH + 'VP8 ', all of one size); frame k re-uses still k % len.
    alph_for(k, alph, w, h) -> ALPH chunk body of frame k (default: the still's own)."""
    parts = []
    for s in stills:
        c = dict(chunks(s))
        assert b"VP8 " in c and b"VP8X" in c, "expected a VP8X lossy still"
        x = c[b"VP8X"]
        w = 1 + int.from_bytes(x[4:7], "little")
        h = 1 + int.from_bytes(x[7:10], "little")
        parts.append((w, h, c.get(b"ALPH"), c[b"VP8 "]))
    w, h = parts[0][0], parts[0][1]
    assert all(p[0] == w and p[1] == h for p in parts)
    body = _chunk(b"VP8X", bytes((0x02 | 0x10, 0, 0, 0)) + u24(w - 1) + u24(h - 1))
    body += _chunk(b"ANIM", struct.pack("<IH", 0, 0))
    for k in range(n_frames):
        _, _, alph, vp8 = parts[k % len(parts)]
        frame = u24(0) + u24(0) + u24(w - 1) + u24(h - 1) + u24(duration_ms) + bytes((0x02,))      # no blend, no dispose
        if alph_for is not None:
            alph = alph_for(k, alph, w, h)
        if alph is not None:
            frame += _chunk(b"ALPH", alph)
        frame += _chunk(b"VP8 ", vp8)
        body += _chunk(b"ANMF", frame)
    return b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WEBP" + body
