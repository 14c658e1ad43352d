"""Generates bench_data/ll1080_mixed_m{4,6}.webp with PIL's libwebp: a photo-like 1080p image (smooth waves, flat
patches with hard edges, 4x4-blocky texture) whose lossless encoding mixes predictor modes per tile (select and
clamp-add-subtract about half each), unlike the noise images of cfg 3 whose tiles all use select; and
bench_data/ll1080_smooth_m4.webp, a smooth image whose tiles nearly all use mode 10 (reads TR).  Used by
tools/lossless_times.py --pattern to time the predictor's mixed-mode path; not part of bench.py's cfg 3."""
import io, os
import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_data")
rng = np.random.default_rng(5)
H, W = 1080, 1920
yy, xx = np.mgrid[0:H, 0:W]
img = np.stack([128 + 100 * np.sin(xx / 97.0) * np.cos(yy / 61.0), 128 + 90 * np.sin((xx + yy) / 140.0),
                128 + 80 * np.cos(xx / 33.0 + yy / 270.0)], -1)
img += rng.normal(0, 1, (H // 4 + 1, W // 4 + 1, 3)).repeat(4, 0).repeat(4, 1)[:H, :W] * 6
img[200:500, 300:900] += 60
img[600:900, 1000:1700] -= 70
img = np.clip(img, 0, 255).astype(np.uint8)
for m in (4, 6):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "WEBP", lossless=True, method=m, quality=75)
    open(os.path.join(OUT, "ll1080_mixed_m%d.webp" % m), "wb").write(b.getvalue())
    print("ll1080_mixed_m%d.webp" % m, len(b.getvalue()))
# a smooth image with fine noise (sky-like): libwebp takes mode 10 (average of averages, reads TR) for nearly every tile
rng = np.random.default_rng(1)
sm = np.clip(np.stack([xx / 7.5 + yy / 13.5, 200 - yy / 6.0, 120 + np.sin(xx / 50.0) * 60], -1) + rng.normal(0, 2, (H, W, 3)), 0, 255).astype(np.uint8)
b = io.BytesIO()
Image.fromarray(sm).save(b, "WEBP", lossless=True, method=4, quality=75)
open(os.path.join(OUT, "ll1080_smooth_m4.webp"), "wb").write(b.getvalue())
print("ll1080_smooth_m4.webp", len(b.getvalue()))
