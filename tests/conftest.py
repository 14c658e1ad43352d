import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")

def golden_input(name):
    with open(os.path.join(GOLDEN, "inputs", name + ".webp"), "rb") as f:
        return f.read()

def golden_expected(name):
    import numpy as np
    z = np.load(os.path.join(GOLDEN, "expected", name + ".npz"))
    return [z["f%d" % i] for i in range(len(z.files))]

GOLDEN_NAMES = ["1", "2", "3", "4", "5", "1ll", "2ll", "3ll", "4ll", "5ll", "1a", "2a"]
