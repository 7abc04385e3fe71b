"""Child process of test_coupling_product_by_tile_groups: the sparse C = B X^T with whatever GMRF_BXT_GROUPS / GMRF_BXT_TILES its
environment sets (both are read once per process).  Prints one JSON line: digests of a few coupling blocks of a darcy256 batch."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
w = pkg.workloads.make(sys.argv[1] if len(sys.argv) > 1 else "darcy256")
vals = np.stack([w.Q.data, 1.5 * w.Q.data])
F = pkg.TridiagonalCholeskyFactor(batch=2).factor(w.Q, w.n_blocks, values=vals)
out = {}
for p in (0, 1):
    F.select_problem(p)
    for i in (0, w.n_blocks // 2, w.n_blocks - 2):
        out[f"C{p}_{i}"] = hashlib.sha256(np.ascontiguousarray(F.Cs[i]).tobytes()).hexdigest()
print(json.dumps(out))
