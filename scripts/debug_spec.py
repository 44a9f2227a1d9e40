import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
from test_gpu_parity import _run_batch
d = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_list = [5000, 37, 1, 12345, 64, 65]
seq = _run_batch(32 * 512, (0, 0, 0), d, n_list, 4, 100)
for blk, warm in ((32, (1, 1, 1)), (64, (2, 4, 2)), (256, (2, 8, 4)), (32, (0, 0, 0)), (32,(8,8,8))):
    spec = _run_batch(blk, warm, d, n_list, 4, 100)
    print("== block", blk, "warm", warm, spec["stats"])
    for key, val in seq.items():
        if key == "stats": continue
        a, b = np.asarray(val, np.float64), np.asarray(spec[key], np.float64)
        diff = np.abs(a - b)
        if diff.size and diff.max() > 0:
            idx = np.unravel_index(np.argmax(diff), diff.shape)
            first = np.argwhere(diff.reshape(diff.shape[0], -1).max(axis=1) > 0)[:3].ravel()
            print(key, "maxdiff", diff.max(), "at", idx, "nbad", int((diff > 0).sum()), "first bad rows", first, "seq", a[idx], "spec", b[idx])
