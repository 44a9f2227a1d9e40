import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np
from test_gpu_parity import _full_chain
from consenrich_amd import cconsenrich as product
from oracle import oracle
o = _full_chain(oracle, 2, 1000000, 4)
for pre in (False, True):
    if pre:
        _full_chain(product, 1, 1000000, 4)
    for xt in (0, 2, 0):
        product.set_validation(xt)
        g = _full_chain(product, 2, 1000000, 4)
        msg = []
        for name in ("xf", "Pf", "D", "xs", "Ps"):
            bad = np.any((g[name] != o[name]).reshape(g[name].shape[0], -1), axis=1)
            msg.append(f"{name}:{int(bad.sum())}" + (f"@{np.nonzero(bad)[0][0]}" if bad.any() else ""))
        print("pre-level", pre, "xtol", xt, " ".join(msg), flush=True)
