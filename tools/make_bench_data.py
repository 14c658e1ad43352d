"""Generates bench_data/*.webp (run in the build container only; needs libwebp + the helper built from
tools/enc_webp.c at /tmp/enc_webp).  Seeded synthetic images as BASELINE.md section 3 prescribes:
RGB gradient + 8x8-blocky N(0,18) + per-pixel N(0,6) noise, clipped to u8."""
import os, subprocess, sys
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_data")
ENC = "/tmp/enc_webp"

def synth(w, h, seed, block_sigma=18.0, px_sigma=6.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = np.stack([255 * xx / w, 255 * yy / h, 255 * (1 - (xx + yy) / (w + h))], axis=-1)
    bh, bw = (h + 7) // 8, (w + 7) // 8
    blocky = rng.normal(0, block_sigma, (bh, bw, 3)).repeat(8, axis=0).repeat(8, axis=1)[:h, :w]
    noise = rng.normal(0, px_sigma, (h, w, 3))
    return np.clip(base + blocky + noise, 0, 255).astype(np.uint8)

def enc(arr, name, quality=75, method=4, filter_type=1, strength=60, sharp=0, lossless=0, alpha_q=100, partitions=0, segments=4):
    h, w, ch = arr.shape
    raw = "/tmp/_enc_in.raw"
    arr.tofile(raw)
    out = os.path.join(OUT, name)
    subprocess.check_call([ENC, raw, str(w), str(h), str(ch), out, str(quality), str(method), str(filter_type), str(strength),
                           str(sharp), str(lossless), str(alpha_q), str(partitions), str(segments)])

def main():
    os.makedirs(OUT, exist_ok=True)
    W, H = 1920, 1080
    for k in range(8):                                   # cfg 2: 1080p lossy, q75 m4 (normal filter)
        enc(synth(W, H, 1234 + k), "lossy1080_q75_%d.webp" % k)
    enc(synth(W, H, 1234), "lossy1080_q90_0.webp", quality=90)
    enc(synth(W, H, 1235), "lossy1080_q75_simple_1.webp", filter_type=0)
    enc(synth(W, H, 1236), "lossy1080_q50_sharp3_2.webp", quality=50, sharp=3)
    for k in range(2):                                   # cfg 3: 1080p lossless (predictor + cross-colour + subtract-green)
        enc(synth(W, H, 2234 + k, block_sigma=6.0, px_sigma=1.0), "lossless1080_%d.webp" % k, lossless=1, method=4)
    rng = np.random.default_rng(99)                      # <= 16 colours: colour-indexing with pixel packing
    pal = rng.integers(0, 256, (12, 3), dtype=np.uint8)
    idx = (rng.integers(0, 12, ((H + 15) // 16, (W + 15) // 16))).repeat(16, axis=0).repeat(16, axis=1)[:H, :W]
    enc(pal[idx], "lossless1080_palette.webp", lossless=1, method=4)
    for k in range(2):                                   # cfg 4: 720p lossy + alpha (soft disc)
        w, h = 1280, 720
        rgb = synth(w, h, 3234 + k)
        yy, xx = np.mgrid[0:h, 0:w]
        r = np.hypot(xx - w / 2 - 40 * k, yy - h / 2)
        a = np.clip((300 - r) / 40.0, 0, 1) * 255
        enc(np.dstack([rgb, a.astype(np.uint8)]), "lossy720_alpha_%d.webp" % k, alpha_q=100)
    enc(synth(3840, 2160, 4234), "lossy2160_q75_0.webp")  # cfg 5: 4K lossy
    enc(synth(333, 217, 5234), "lossy_odd_333x217.webp", quality=60)   # odd size, partial macroblocks
    enc(synth(16, 16, 5235), "lossy_16x16.webp")
    enc(synth(1, 1, 5236), "lossy_1x1.webp")

if __name__ == "__main__":
    main()
