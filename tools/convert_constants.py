"""Searches multiply-and-shift constants that reproduce the rounded chroma terms of Tools.toArgb (Tools.hx:25-31) exactly:
  R: floor(1.402 cr + 1/2)      B: floor(1.772 cb + 1/2)      G: floor(1/2 - 0.344136 cb - 0.714136 cr)
for every cb, cr in -128..127, against exact rational arithmetic.  The constants are used by webp_amd/csrc/kernels/wp_convert.h
(the device-side check of all 2^24 (Y, Cb, Cr) triples against the fp64 oracle is tests/test_gpu_convert.py)."""
import numpy as np
c = np.arange(-128, 128, dtype=np.int64)
tr = np.floor_divide(1402 * c + 500, 1000)
tb = np.floor_divide(1772 * c + 500, 1000)
assert np.array_equal(tr, (5743 * c + 2048) >> 12)
assert np.array_equal(tb, (3629 * c + 1031) >> 11)
CB, CR = np.meshgrid(c, c, indexing="ij")
tg = np.floor_divide(500000 - 344136 * CB - 714136 * CR, 1000000)
assert np.array_equal(tg, (1048616 - 721705 * CB - 1497652 * CR) >> 21)
assert np.abs(1048616 - 721705 * CB - 1497652 * CR).max() < 2 ** 31 and 1497652 < 2 ** 23      # 24-bit multiplies, 32-bit sums
print("constants verified for all 256 / 65536 arguments")


def search():
    for S in range(8, 24):
        K0 = round(1.772 * 2 ** S)
        for K in range(K0 - 3, K0 + 4):
            for R in range(2 ** (S - 1) - 2000, 2 ** (S - 1) + 2000):
                if np.array_equal((K * c + R) >> S, tb):
                    return S, K, R


if __name__ == "__main__":
    print("B term: shift, multiplier, rounding constant =", search())
