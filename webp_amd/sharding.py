"""Multi-GPU policy for this path: independent images / animation frames are split into contiguous index ranges, one per
rank (one process per GPU), weighted by pixel count so that ranks get equal work rather than equal file counts.  There is
no exchange step, hence no collective on the data path; ranks only meet for timing, and the caller gathers
results by index.

    lo, hi = shard_by_weight(weights, rank, world)          # policy
    lo, hi, images = decode_files(datas, rank, world)        # product path: parse + decode this rank's shard on its GPU
"""
import struct
from typing import List, Sequence


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) of `n_items` owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_by_weight(weights: Sequence[int], rank: int, world: int):
    """Contiguous [lo, hi) such that every rank's total weight is as close to total / world as contiguity allows: item i
    goes to the rank in whose share its midpoint falls.  Uniform weights reduce to ceil/floor(n / world) items each."""
    total = sum(weights)
    if total <= 0:
        return shard_range(len(weights), rank, world)
    bounds = [0] * (world + 1)
    acc, r = 0, 1
    for i, w in enumerate(weights):
        mid = acc + w / 2.0
        while r < world and mid >= total * r / world:
            bounds[r] = i
            r += 1
        acc += w
    for k in range(r, world + 1):
        bounds[k] = len(weights)
    return bounds[rank], bounds[rank + 1]


def peek_pixels(data: bytes) -> int:
    """Pixels a file decodes to (canvas size x frame count for animations), from the container headers alone
    (WebPDecoder.hx:37-76, 133-148; Vp8Decoder.hx:126-160; Vp8LDecoder.hx:347-360).  0 if the headers do not parse: such a
    file costs nothing to decode and fails on the rank that owns it."""
    try:
        if len(data) < 20 or data[:4] != b"RIFF" or data[8:12] != b"WEBP":
            return 0
        pos, canvas, frames, first = 12, 0, 0, 0
        while pos + 8 <= len(data):
            tag, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
            body = pos + 8
            if tag == b"VP8X" and size >= 10:
                w = 1 + int.from_bytes(data[body + 4:body + 7], "little")
                h = 1 + int.from_bytes(data[body + 7:body + 10], "little")
                canvas = w * h
            elif tag == b"VP8 " and size >= 10 and not first:
                w, h = struct.unpack_from("<HH", data, body + 6)
                first = (w & 0x3fff) * (h & 0x3fff)
            elif tag == b"VP8L" and size >= 5 and not first:
                bits = struct.unpack_from("<I", data, body + 1)[0]
                first = ((bits & 0x3fff) + 1) * (((bits >> 14) & 0x3fff) + 1)
            elif tag == b"ANMF":
                frames += 1
            pos = body + size + (size & 1)
        if frames:
            return canvas * frames
        return first or canvas
    except Exception:
        return 0


def decode_files(datas: List[bytes], rank: int, world: int, threads: int = 0):
    """The multi-GPU decode path, one call per rank (one process per GPU, the device already selected): takes the
    contiguous, pixel-weighted shard of the global file list that belongs to `rank`, parses it on host threads and decodes
    it as one device batch.  Returns (lo, hi, images): images[i] is the Image of file lo + i, or the WebPError it failed
    with (a bad file fails alone).  No communication: the caller owns gathering by index."""
    from .decoder import decode_many
    lo, hi = shard_by_weight([peek_pixels(d) for d in datas], rank, world)
    return lo, hi, decode_many(datas[lo:hi], threads)
