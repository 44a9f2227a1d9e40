import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np
from test_gpu_parity import _run_batch
n_list = [5000]
seq = _run_batch(32 * 512, (0, 0, 0), 2, n_list, 4, 100)
for blk, warm in ((32, (0, 0, 0)), (64, (0, 0, 0)), (128, (0,0,0)), (32, (1,1,1)), (32,(2,2,2))):
    spec = _run_batch(blk, warm, 2, n_list, 4, 100)
    print("== block", blk, "warm", warm, spec["stats"])
    for name in ("Pf", "xf", "xs", "Ps"):
        a, b = seq[(0, name)], spec[(0, name)]
        bad = np.any((a != b).reshape(a.shape[0], -1), axis=1)
        blocks = np.unique(np.nonzero(bad)[0] // blk)
        print(name, "bad blocks:", blocks[:40], "count", len(blocks), "of", (5000 + blk - 1) // blk)
