"""CPU: the writer oracle (the reference's own pandas call, consenrich.py:9797-9805) against an independent statement of
the format with Python's correctly rounded '%.4f'."""
import numpy as np

from oracle import writers as ow


def edge_values():
    rng = np.random.default_rng(0)
    v = np.concatenate([
        np.asarray([0.03125, -0.03125, 0.09375, -0.0, 0.0, 1e-9, -1e-9, 12345.67891, np.nan, np.inf, -np.inf, 3.4e38,
                    -3.4028235e38, 1e15, 9.0e14, 9.3e14, 1.8e15, 2.5e19, 0.00005, 0.00015, 2.5, 16777216.0, 1e-45, 0.99995,
                    0.99996, 9.99995, 99999.99, 1e7, 123456789.0], np.float32),
        rng.normal(0, 3, 3000).astype(np.float32),
        (rng.integers(-2 ** 20, 2 ** 20, 500) / 32.0 + 1 / 64.0).astype(np.float32),      # exact ties at 4 decimals
        np.exp(rng.uniform(-30, 80, 1500)).astype(np.float32) * rng.choice([-1, 1], 1500),
    ]).astype(np.float32)
    return v


def test_pandas_call_equals_python_percent_formatting():
    v = edge_values()
    n = len(v)
    s = np.arange(n, dtype=np.int64) * 200 + 10_000
    e = s + 200
    for tr in (None, "round4", "sqrt"):
        assert ow.bedgraph_bytes("chr1", s, e, v, tr) == ow.bedgraph_bytes_python("chr1", s, e, v, tr)
    txt = ow.bedgraph_bytes("chrX_alt", [0, 5], [5, 9], np.asarray([0.03125, np.nan], np.float32))
    assert txt == b"chrX_alt\t0\t5\t0.0312\nchrX_alt\t5\t9\t\n"       # round-half-even on the exact value; NaN -> empty
