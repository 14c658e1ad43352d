"""Test helper: hand-authored VP8 key frames.  A boolean ENCODER (RFC 6386 section 7.3, the mirror of
Partition.readBit) writes the frame-header fields of the first partition with chosen values -- segment quantiser /
filter-strength values and loop-filter deltas of either sign, the number of DCT partitions -- in the order
Vp8Decoder.parseOtherHeaders reads them (Vp8Decoder.hx:289-324: colour space, clamp, parseSegmentHeader :203-227,
parseFilterHeader :229-258, the partition count :263, parseQuant :382-388).  Everything after those fields (token
probability updates, skip probability, the per-macroblock modes, and the DCT partitions themselves) is seeded random bits:
every bit pattern is a valid VP8 stream short of running out of data, so the decoders read random segment ids, skip flags,
i16 / i4 modes and coefficient tokens -- what matters to the tests is that two decoders read the SAME thing from it."""
import struct

import numpy as np


class BoolEncoder:
    def __init__(self):
        self.out = bytearray()
        self.range = 255
        self.bottom = 0
        self.bit_count = 24

    def _carry(self):
        i = len(self.out) - 1
        while i >= 0 and self.out[i] == 255:
            self.out[i] = 0
            i -= 1
        if i >= 0:
            self.out[i] += 1

    def put(self, bit, prob=128):
        split = 1 + (((self.range - 1) * prob) >> 8)
        if bit:
            self.bottom += split
            self.range -= split
        else:
            self.range = split
        while self.range < 128:
            self.range <<= 1
            if self.bottom & (1 << 31):
                self._carry()
            self.bottom = (self.bottom << 1) & 0xFFFFFFFF
            self.bit_count -= 1
            if self.bit_count == 0:
                self.out.append(self.bottom >> 24)
                self.bottom &= (1 << 24) - 1
                self.bit_count = 8

    def uint(self, v, n):                       # Partition.readUint: most significant bit first
        for i in range(n):
            self.put((v >> (n - 1 - i)) & 1)

    def optional_int(self, v, n):               # Partition.readOptionalInt: flag, magnitude, sign
        if v == 0:
            self.put(0)
            return
        self.put(1)
        self.uint(abs(v), n)
        self.put(1 if v < 0 else 0)

    def flush(self):
        c = self.bit_count
        v = self.bottom
        if v & (1 << (32 - c)):
            self._carry()
        v = (v << (c & 7)) & 0xFFFFFFFF
        c >>= 3
        while c > 0:
            v = (v << 8) & 0xFFFFFFFF
            c -= 1
        for _ in range(4):
            self.out.append(v >> 24)
            v = (v << 8) & 0xFFFFFFFF
        return bytes(self.out)


def forge(width, height, seed, segments=None, filt=None, log2_parts=0, quant=None, part_bytes=None, tail_bits=60000):
    """One VP8 key frame in a RIFF container.
    segments: None, or dict(update_map, absolute, quantizer=[4 signed], strength=[4 signed], probs=[3 or None])
    filt:     dict(simple, level, sharpness, ref_delta=[4] or None, mode_delta=[4] or None)
    quant:    dict(base, y1dc, y2dc, y2ac, uvdc, uvac)
    part_bytes: length of each DCT partition (default: generous for the frame size)"""
    rng = np.random.default_rng(seed)
    filt = filt or dict(simple=0, level=20, sharpness=0, ref_delta=None, mode_delta=None)
    quant = quant or dict(base=40, y1dc=0, y2dc=0, y2ac=0, uvdc=0, uvac=0)
    e = BoolEncoder()
    e.put(0); e.put(0)                                           # colour space, clamping type
    if segments is None:
        e.put(0)
    else:
        e.put(1)
        e.put(1 if segments.get("update_map", True) else 0)
        if "quantizer" in segments:
            e.put(1)
            e.put(1 if segments.get("absolute", False) else 0)   # relativeDelta = !bit
            for v in segments["quantizer"]:
                e.optional_int(v, 7)
            for v in segments["strength"]:
                e.optional_int(v, 6)
        else:
            e.put(0)
        if segments.get("update_map", True):
            for p in segments.get("probs", [None, None, None]):
                if p is None:
                    e.put(0)
                else:
                    e.put(1); e.uint(p, 8)
    e.put(filt["simple"]); e.uint(filt["level"], 6); e.uint(filt["sharpness"], 3)
    if filt.get("ref_delta") is None and filt.get("mode_delta") is None:
        e.put(0)
    else:
        e.put(1); e.put(1)
        for v in filt.get("ref_delta") or [0, 0, 0, 0]:
            e.optional_int(v, 6)
        for v in filt.get("mode_delta") or [0, 0, 0, 0]:
            e.optional_int(v, 6)
    e.uint(log2_parts, 2)
    e.uint(quant["base"], 7)
    for k in ("y1dc", "y2dc", "y2ac", "uvdc", "uvac"):
        e.optional_int(quant[k], 4)
    for b in rng.integers(0, 2, tail_bits):                      # the rest of the first partition: random bits
        e.put(int(b), int(rng.integers(1, 256)) if (b & 0) else 128)
    first = e.flush()
    n_parts = 1 << log2_parts
    mbs = ((width + 15) >> 4) * ((height + 15) >> 4)
    part_bytes = part_bytes or [max(4096, 1400 * mbs // n_parts + 2048)] * n_parts
    parts = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in part_bytes]
    sizes = b"".join(struct.pack("<I", len(p))[:3] for p in parts[:-1])
    tag = (0 | (0 << 1) | (1 << 4) | (len(first) << 5))          # key frame, version 0, show
    frame = struct.pack("<I", tag)[:3] + b"\x9d\x01\x2a" + struct.pack("<HH", width, height) + first + sizes + b"".join(parts)
    if len(frame) & 1:
        frame += b"\0"
    return b"RIFF" + struct.pack("<I", 4 + 8 + len(frame)) + b"WEBP" + b"VP8 " + struct.pack("<I", len(frame)) + frame


# the cases the tests share: (name, kwargs).  Negative segment values and loop-filter deltas read back as 256 + v through
# the reference's unsigned Int8Vector; more than one DCT partition makes every partition alias the start of the
# first (Q3).
CASES = [
    ("rel_neg_quant", dict(width=96, height=64, seed=1, segments=dict(quantizer=[-20, 13, -64, 0], strength=[0, 0, 0, 0], probs=[128, 100, 150]))),
    ("rel_neg_strength", dict(width=96, height=64, seed=2, segments=dict(quantizer=[0, 0, 0, 0], strength=[-10, 5, -63, 20], probs=[128, 128, 128]),
                              filt=dict(simple=0, level=30, sharpness=2, ref_delta=None, mode_delta=None))),
    ("abs_neg_both", dict(width=80, height=48, seed=3, segments=dict(absolute=True, quantizer=[-5, 100, -127, 30], strength=[-1, 63, -33, 8], probs=[90, 128, 200]),
                          filt=dict(simple=1, level=11, sharpness=0, ref_delta=None, mode_delta=None))),
    ("lf_deltas_neg", dict(width=96, height=64, seed=4, filt=dict(simple=0, level=25, sharpness=5, ref_delta=[-12, 3, 0, -1], mode_delta=[-30, 7, 0, 0]))),
    ("lf_deltas_pos_segments", dict(width=64, height=64, seed=5, segments=dict(quantizer=[10, -10, 0, 3], strength=[3, -3, 9, 0], probs=[128, 128, 128]),
                                    filt=dict(simple=0, level=40, sharpness=0, ref_delta=[20, 0, 0, 0], mode_delta=[40, 0, 0, 0]))),
    ("two_partitions", dict(width=96, height=80, seed=6, log2_parts=1)),
    ("four_partitions_neg", dict(width=96, height=96, seed=7, log2_parts=2, segments=dict(quantizer=[-3, 4, -5, 6], strength=[-7, 8, -9, 10], probs=[128, 128, 128]),
                                 filt=dict(simple=0, level=33, sharpness=1, ref_delta=[-2, 2, 0, 0], mode_delta=[-4, 0, 0, 0]))),
    ("eight_partitions", dict(width=64, height=144, seed=8, log2_parts=3)),
    ("quant_deltas", dict(width=64, height=48, seed=9, quant=dict(base=120, y1dc=-15, y2dc=15, y2ac=-8, uvdc=9, uvac=-15),
                          segments=dict(update_map=False, quantizer=[-100, 0, 0, 0], strength=[0, 0, 0, 0]))),
    ("short_partitions", dict(width=64, height=64, seed=10, log2_parts=2, part_bytes=[40, 4000, 60, 3000])),
    ("partition_runs_dry", dict(width=64, height=64, seed=11, log2_parts=1, part_bytes=[3, 9000])),      # decodeFrame returns null (Q11)
]
