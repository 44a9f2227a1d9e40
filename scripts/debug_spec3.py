import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np
from test_gpu_parity import _run_batch
n_list = [5000]
seq = _run_batch(32 * 512, (0, 0, 0), 2, n_list, 4, 100)
os.environ["CONSENRICH_AMD_DEBUG"] = "1"
spec = _run_batch(32, (0,0,0), 2, n_list, 4, 100)
for name in ("Pf",):
    a, b = seq[(0, name)], spec[(0, name)]
    bad = np.any((a != b).reshape(a.shape[0], -1), axis=1)
    print(name, "bad blocks:", np.unique(np.nonzero(bad)[0] // 32)[:40])
