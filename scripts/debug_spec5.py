import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np
from test_gpu_parity import _run_batch
n_list = [5000]
seq = _run_batch(32 * 512, (0, 0, 0), 2, n_list, 4, 100)
def trial(tag):
    for rep in range(3):
        spec = _run_batch(32, (0,0,0), 2, n_list, 4, 100)
        out = []
        for name in ("Pf","xf","xs"):
            a, b = seq[(0, name)], spec[(0, name)]
            bad = np.any((a != b).reshape(a.shape[0], -1), axis=1)
            out.append(len(np.unique(np.nonzero(bad)[0] // 32)))
        print(tag, rep, "bad block counts Pf/xf/xs", out, {k: spec["stats"][k] for k in ("reruns_p","reruns_x","reruns_b","fix_launches")})
trial("plain")
os.environ["CONSENRICH_AMD_POISON"] = "1"; trial("poison"); del os.environ["CONSENRICH_AMD_POISON"]
os.environ["CONSENRICH_AMD_FENCE"] = "1"; trial("fence")
