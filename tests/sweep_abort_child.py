"""Child process of test_persistent_sweep_abort_falls_back (GMRF_SWEEP_SPIN_MS=0 in its environment: the first look at an input
panel inside a persistent sweep that finds a sentinel gives up at once and raises the abort words; the factorisation's persistent
launches keep their own bound).  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

pkg = g.load_package()
out = {}
w = pkg.workloads.make("burgers512x64")
rhs = torch.from_numpy(w.rhs).cuda()
F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks)
keys = ("persist_aborts", "persist_cus", "sweep_persist")
out["after_factor"] = {k: F.stats()[k] for k in keys}
mu = pkg.ldiv(F, rhs)                                   # the forward sweep gives up; the solve is repeated with a launch per product
out["after_solve"] = {k: F.stats()[k] for k in keys}
X = F.sample(16, mean=mu, seed=11, like=rhs)            # the handle has left the persistent forms: no second abort
out["after_sample"] = {k: F.stats()[k] for k in keys}
F.refactor(w.Q.data)                                    # ... and its factorisations take the launch-per-step route from now on
out["route_after_refactor"] = int(F.stats()["persist_route"])
mu2 = pkg.ldiv(F, rhs)
G = pkg.TridiagonalCholeskyFactor()
G.set_eager(65536)                                      # a launch per product up front
G.factor(w.Q, w.n_blocks)
mu_g = pkg.ldiv(G, rhs)
X_g = G.sample(16, mean=mu_g, seed=11, like=rhs)
out["solve_equal"] = bool(torch.equal(mu, mu_g) and torch.equal(mu2, mu_g))
out["sample_equal"] = bool(torch.equal(X, X_g))
out["aborts_of_the_per_product_form"] = int(G.stats()["persist_aborts"])
# a fresh handle whose FIRST persistent sweep is a sample's (the abort is seen behind the sample's own synchronisation)
del F, G                                                # (the chip is theirs until they let go of it)
import gc; gc.collect()
H = pkg.tridiagonal_cholesky(w.Q, w.n_blocks)
Xh = H.sample(16, mean=mu_g, seed=11, like=rhs)
out["sample_first"] = {k: H.stats()[k] for k in keys}
out["sample_first_equal"] = bool(torch.equal(Xh, X_g))
print(json.dumps(out))
