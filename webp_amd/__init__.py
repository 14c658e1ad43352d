"""webp_amd: MI355X (gfx950) reconstruction backend for the fourst4r/webp decode path.

Host side mirrors the reference's public surface (WebPDecoder.decode / Tools.toArgb / ImageData);
pixel work runs in hand-written HIP kernels behind the C ABI declared in include/webp_amd.h."""
from ._lib import WebPError, lib, SO_PATH  # noqa: F401
from .decoder import (Image, ImageData, Yuv420, Argb, Anim, AnimImage, AnimFrameHeader, FrameHeader,  # noqa: F401
                      WebPDecoder, Tools, parse, parse_many, ParsedFile, decode_batch, decode_many)
