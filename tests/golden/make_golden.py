"""Regenerates tests/golden/ from the reference's own test data (run in the build container only;
needs /root/reference and PIL).  inputs/*.webp are the reference's test inputs (test/*.webp);
expected/<name>.npz holds the reference's golden PNGs (test/expected/*.png) converted to the raw
A,R,G,B byte order that test/Visual.hx handed to format.png (build32ARGB), one array per frame."""
import glob, os, shutil, sys, hashlib, json
import numpy as np
from PIL import Image

REF = "/root/reference/test"
HERE = os.path.dirname(os.path.abspath(__file__))

def argb_of(png):
    im = Image.open(png)
    assert im.mode == "RGBA", (png, im.mode)
    a = np.asarray(im)
    return np.ascontiguousarray(np.concatenate([a[..., 3:4], a[..., 0:3]], axis=-1)), im.size

def main():
    for w in sorted(glob.glob(REF + "/*.webp")):
        shutil.copy(w, os.path.join(HERE, "inputs", os.path.basename(w)))
    names = ["1", "2", "3", "4", "5", "1ll", "2ll", "3ll", "4ll", "5ll", "1a", "2a"]
    index = {}
    for n in names:
        single = os.path.join(REF, "expected", n + ".png")
        arrays = {}
        if os.path.exists(single):
            arr, _ = argb_of(single)
            arrays["f0"] = arr
        else:
            i = 0
            while os.path.exists(os.path.join(REF, "expected", "%s_%d.png" % (n, i))):
                arrays["f%d" % i], _ = argb_of(os.path.join(REF, "expected", "%s_%d.png" % (n, i)))
                i += 1
        np.savez_compressed(os.path.join(HERE, "expected", n + ".npz"), **arrays)
        index[n] = {k: {"shape": list(v.shape), "sha256": hashlib.sha256(v.tobytes()).hexdigest()} for k, v in arrays.items()}
    json.dump(index, open(os.path.join(HERE, "expected", "index.json"), "w"), indent=0, sort_keys=True)

if __name__ == "__main__":
    main()
