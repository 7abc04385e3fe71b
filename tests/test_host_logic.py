"""CPU tests of the host side: C-ABI surface, argument checking, sharding helpers, workloads."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg, lib):
    hdr = open(os.path.join(ROOT, "include", "gmrf_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(gmrf_[a-z0-9_]+)\(", hdr)) - {"gmrf_status"})
    assert declared, "no declarations parsed"
    assert sorted(pkg._cabi.EXPORTS) == declared                 # binding list == header
    nm = subprocess.run(["nm", "-D", "--defined-only", pkg._cabi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (gmrf_[a-z0-9_]+)", nm))
    assert set(declared) <= exported
    for s in declared:
        assert hasattr(lib, s)
    assert lib.gmrf_version() >= 100


def test_no_gpu_means_loud_failure_not_fallback(pkg, lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    st = lib.gmrf_bt_create(0, None, C.byref(h))
    assert st == pkg._cabi.ERR_NO_DEVICE
    with pytest.raises(pkg.GmrfError) as e:
        pkg.tridiagonal_cholesky(sp.identity(8, format="csc"), 2)
    assert e.value.status == pkg._cabi.ERR_NO_DEVICE
    w = pkg.workloads.random_block_tridiagonal(2, 8)
    with pytest.raises(pkg.GmrfError):
        pkg.CsrMatrix(w.Q)


def test_storage_size_query_needs_no_gpu(pkg, lib):
    a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    assert lib.gmrf_bt_storage_bytes(65536, 64, 1, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert a.value == 64 * 1024 * 1024 * 8 and b.value == 63 * 1024 * 1024 * 8 and c.value == a.value
    assert lib.gmrf_bt_storage_bytes(65536, 64, 3, C.byref(a), C.byref(b), C.byref(c)) == 0      # a batch scales every array
    assert a.value == 3 * 64 * 1024 * 1024 * 8 and b.value == 3 * 63 * 1024 * 1024 * 8
    assert lib.gmrf_bt_storage_bytes(600, 2, 1, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert a.value == 2 * 512 * 512 * 8                            # 300 -> padded to 64 * 8 = 512
    assert lib.gmrf_bt_storage_bytes(10, 3, 1, C.byref(a), C.byref(b), C.byref(c)) == pkg._cabi.ERR_BAD_SHAPE
    assert lib.gmrf_bt_storage_bytes(64, 1, 0, C.byref(a), C.byref(b), C.byref(c)) == pkg._cabi.ERR_BAD_SHAPE


def test_persistent_launch_budget_refuses_a_fifth_batched_handle(pkg, lib):
    """Round 5: every workgroup of a persistent launch must be resident, so the handles of a device share a
    budget of its CUs (gmrf_handle::persist_cus; persist_plan in csrc/gmrf_hip.hip).  Four StreamSet handles of batch 8 ask for
    4 x 8 x 7 = 224 of 256 CUs and get them; a fifth is refused UP FRONT (it takes potrf_diag128 + GEMM) instead of meeting the
    200 ms bound of a wait; two one-problem handles of 135 workgroups (darcy256) do not fit together either."""
    import ctypes as C

    def grant(cus, demands):
        d = (C.c_int32 * len(demands))(*demands); g = (C.c_int32 * len(demands))()
        pkg._cabi.check(lib.gmrf_test_persist_budget(cus, len(demands), d, g))
        return list(g)
    assert grant(256, [56] * 5) == [1, 1, 1, 1, 0]
    assert grant(256, [135, 135]) == [1, 0]
    assert grant(256, [135, 56, 56, 56]) == [1, 1, 1, 0]
    assert grant(256, [0, 256]) == [0, 1] and grant(256, [257]) == [0]


def test_argument_validation_in_python_layer(pkg):
    with pytest.raises(ValueError):
        pkg.tridiagonal_cholesky(sp.identity(10, format="csc"), 3)
    x = np.arange(10.0)
    assert [len(c) for c in pkg.make_chunks(x, 4)] == [2, 2, 2, 4]


def test_shard_helpers(pkg):
    from importlib import import_module
    import __graft_entry__ as g
    post = import_module(g.PKG_NAME + ".posterior")
    for total, world in [(64, 8), (65, 8), (7, 3), (0, 2)]:
        parts = [post.shard_range(total, world, r) for r in range(world)]
        assert sum(c for _, c in parts) == total
        assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
    assert post.block_groups(10, 4) == [(0, 4), (4, 8), (8, 10)]


@pytest.mark.parametrize("name,n,N", [("darcy64", 4096, 16), ("burgers64x8", 512, 8), ("elliptic32", 1024, 16)])
def test_workloads_are_spd_block_tridiagonal(pkg, name, n, N):
    w = pkg.workloads.make(name)
    assert w.n == n and w.n_blocks == N
    assert pkg.workloads.block_bandwidth_ok(w.Q, N)
    assert abs(w.Q - w.Q.T).max() == 0.0
    if n <= 1024:
        assert np.linalg.eigvalsh(w.Q.toarray()).min() > 0


def test_baseline_config_shapes(pkg):
    # sizes of SURVEY.md section 8a, without building the big ones
    w = pkg.workloads.darcy(16)
    assert (w.n, w.n_blocks, w.block_size) == (256, 4, 64)
    assert set(pkg.workloads.CONFIGS) >= {"burgers512x64", "darcy64", "darcy256", "elliptic512", "burgers4096x512"}


def test_posterior_assembler_symbolic_phase_without_gpu(pkg, lib):
    """pattern of Q + noise J'J and the product lists, host side only."""
    from oracle import bt_oracle as O
    gn = pkg.workloads.burgers_gauss_newton(32, 6)
    J = gn["jacobian"](gn["x_prior"])
    asm = pkg.PosteriorAssembler(gn["Q"], J, device=-1)
    A = O.assemble_posterior(gn["Q"], J, 1.0)
    assert asm.nnz_out == A.nnz
    assert np.array_equal(asm.pattern.indptr, A.indptr) and np.array_equal(asm.pattern.indices, A.indices)
    assert asm.n_products == int((np.diff(J.indptr) ** 2).sum())
    assert pkg.workloads.block_bandwidth_ok(asm.pattern, gn["n_blocks"])
    with pytest.raises(pkg.GmrfError) as e:           # numeric phase needs the GPU: no CPU fallback
        asm.precision(gn["Q"].data, J.data, 1.0)
    assert e.value.status == pkg._cabi.ERR_NO_DEVICE
    with pytest.raises(ValueError):
        pkg.PosteriorAssembler(gn["Q"], J[:, :-1], device=-1)


def _check_julia_ccalls(src, hdr, min_calls):
    calls = re.findall(r"ccall\(\(:(\w+),\s*libgmrf\),\s*(\w+),\s*\(([^)]*)\)", src, flags=re.S)
    protos = {m.group(1): m.group(2) for m in re.finditer(r"gmrf_status\s+(gmrf_\w+)\s*\(([^;]*?)\);", hdr, flags=re.S)}
    assert len(calls) >= min_calls

    def c_kind(a):
        a = a.strip()
        if "*" in a:
            return "ptr"
        return {"int64_t": "Int64", "int32_t": "Int32", "uint64_t": "UInt64", "double": "Float64"}[a.split()[0]]

    def j_kind(a):
        a = a.strip()
        return "ptr" if a.startswith(("Ptr{", "Ref{")) else a

    for name, ret, args in calls:
        if name == "gmrf_last_error":
            assert ret == "Cstring"
            continue
        if name == "gmrf_version":
            assert ret == "Int32" and not args.strip()
            continue
        assert name in protos, name
        assert ret == "Int32"                                  # gmrf_status
        jt = [j_kind(a) for a in args.split(",") if a.strip()]
        ct = [c_kind(a) for a in protos[name].split(",") if a.strip() and a.strip() != "void"]
        assert jt == ct, (name, jt, ct)
    return {name for name, _, _ in calls}


def test_julia_shim_ccalls_match_the_header():
    """The Julia shim cannot be executed here (no Julia in the image): check statically that every
    `ccall` names an exported function with the same number of arguments and compatible C types, and
    that EVERY non-test export of the header is bound by the shim."""
    src = open(os.path.join(ROOT, "julia", "DiffEqGMRFsHIP.jl")).read()
    hdr = open(os.path.join(ROOT, "include", "gmrf_hip.h")).read()
    bound = _check_julia_ccalls(src, hdr, 40)
    declared = set(re.findall(r"\b(gmrf_[a-z0-9_]+)\(", hdr)) - {"gmrf_status"}
    missing = sorted(s for s in declared if not s.startswith("gmrf_test_") and s not in bound)
    assert not missing, missing
    # struct mirrors: same number of fields as the C structs
    stats_fields = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} gmrf_stats;", hdr, flags=re.S).group(1)
    stats_fields = re.sub(r"/\*.*?\*/", "", stats_fields, flags=re.S)
    n_c = len(re.findall(r"^\s*(?:double|int64_t|int32_t)\s+[^;]+;", stats_fields, flags=re.M))
    jl = re.search(r"struct GmrfStats(.*?)\nend", src, flags=re.S).group(1)
    c_names = [n.strip().split("[")[0] for line in re.findall(r"^\s*(?:double|int64_t|int32_t)\s+([^;]+);", stats_fields, flags=re.M) for n in line.split(",")]
    j_names = re.findall(r"(\w+)::", jl)
    assert c_names == j_names, (c_names, j_names)
    assert n_c >= 6


def test_julia_solver_blueprint_ccalls_match_the_shim():
    """julia/BlockTridiagonalSolver.jl (the GaussianMarkovRandomFields.jl solver blueprint)
    goes through the shim only: no ccall of its own, and every shim function it uses exists."""
    src = open(os.path.join(ROOT, "julia", "BlockTridiagonalSolver.jl")).read()
    shim = open(os.path.join(ROOT, "julia", "DiffEqGMRFsHIP.jl")).read()
    assert "ccall(" not in src
    used = set(re.findall(r"\bHIP\.(\w+!?)", src)) - {"jl"}          # ("DiffEqGMRFsHIP.jl" in comments)
    assert used, "the adapter must call the shim"
    defined = set(re.findall(r"^(?:function\s+)?(\w+!?)\(", shim, flags=re.M)) | set(re.findall(r"^(?:mutable\s+)?struct\s+(\w+)", shim, flags=re.M)) \
        | set(re.findall(r"^const\s+([\w, ]+)=", shim, flags=re.M)) | set(re.findall(r"^(\w+!?)\(.*\)\s*=", shim, flags=re.M))
    consts = {c.strip() for grp in re.findall(r"^const\s+([\w, ]+?)\s*=", shim, flags=re.M) for c in grp.split(",")}
    missing = sorted(u for u in used if u not in defined and u not in consts)
    assert not missing, missing


def test_darcy_p1_pattern_and_oracle_assembly_without_gpu(pkg):
    """SURVEY 8f rank 4, first piece: the 7-point CSR pattern of the structured P1 Darcy stiffness (library,
    device = -1) equals the oracle's; the oracle's restatement of assemble_darcy_diff_matrix
    (src/problems/darcy.jl:5-63) agrees with the independent vectorised generator behind the packaged
    workloads; the numeric phase needs the GPU (no CPU fallback)."""
    from oracle import bt_oracle as O
    n = 32
    coeff = pkg.workloads.darcy_coefficient(523802340)
    gq = np.linspace(0.0, 1.0, 241)
    GX, GY = np.meshgrid(gq, gq, indexing="ij")
    table = coeff(GX.ravel(), GY.ravel()).reshape(241, 241)
    G, f = O.assemble_darcy_diff_matrix(n, n, gq, gq, table, 1.0)
    d = pkg.DarcyP1Assembler(n, n, device=-1)
    assert d.nnz == G.nnz and np.array_equal(d.pattern.indptr, G.indptr) and np.array_equal(d.pattern.indices, G.indices)
    _, obs, _ = pkg.workloads.darcy_conditioning(n, seeds=(523802340,))
    A, y = obs[0]
    assert abs(G - A).max() / abs(A).max() < 1e-14 and np.max(np.abs(f - y)) < 1e-16
    # Ferrite's apply!: constrained rows / columns are unit-scale diagonal only, f vanishes there
    bnd = np.flatnonzero((np.arange(n * n) % n == 0) | (np.arange(n * n) // n == 0))
    Gd = G.toarray()
    assert np.count_nonzero(Gd[bnd]) == bnd.size and np.all(f[bnd] == 0.0)
    with pytest.raises(pkg.GmrfError) as e:
        d.assemble(table)
    assert e.value.status == pkg._cabi.ERR_NO_DEVICE


def test_burgers_p1_pattern_and_oracle_tangent_without_gpu(pkg):
    """SURVEY 8f rank 4, second piece: the CSR pattern of the Burgers space-time tangent J (library, device = -1)
    equals the pattern of the oracle's f_and_J (scripts/burgers/solve_burgers_gmrf-fem.jl:118-149); the oracle's
    restatement of assemble_burgers_advection_matrix (src/problems/burgers.jl:5-59) is pinned by what the function
    computes: its matrix is the derivative of its residual vector (finite differences), the residual of a constant
    state vanishes, mass / diffusion have their closed forms; the numeric phase needs the GPU (no CPU fallback)."""
    from oracle import bt_oracle as O
    ns, nt, dt, nu = 16, 5, 0.05, 0.01 / np.pi
    rng = np.random.default_rng(4)
    w = rng.standard_normal(ns * nt)
    f, J = O.burgers_f_and_J(ns, nt, dt, nu, w)
    b = pkg.BurgersP1Tangent(ns, nt, dt, nu, device=-1)
    assert b.pattern.shape == J.shape and b.nnz == J.nnz == 6 * (nt - 1) * ns
    assert np.array_equal(b.pattern.indptr, J.indptr) and np.array_equal(b.pattern.indices, J.indices)
    # the oracle itself
    h = 1.0 / ns
    M, G = O.assemble_burgers_mass_diffusion_matrices(ns)
    assert abs(M[3, 3] - 4 * h / 6) < 1e-16 and abs(M[3, 4] - h / 6) < 1e-16 and abs(M[0, ns - 1] - h / 6) < 1e-16
    assert abs(G[3, 3] - 2 / h) < 1e-12 and abs(G[3, 2] + 1 / h) < 1e-12 and abs(G[ns - 1, 0] + 1 / h) < 1e-12
    u = w[:ns]
    Gt, vt = O.assemble_burgers_advection_matrix(ns, u)
    num = np.zeros((ns, ns))
    for j in range(ns):
        e = np.zeros(ns); e[j] = 1e-6
        num[:, j] = (O.assemble_burgers_advection_matrix(ns, u + e)[1] - O.assemble_burgers_advection_matrix(ns, u - e)[1]) / 2e-6
    assert np.max(np.abs(Gt.toarray() - num)) < 1e-8
    assert np.max(np.abs(O.assemble_burgers_advection_matrix(ns, np.full(ns, 0.7))[1])) < 1e-15     # u u_x = 0
    assert abs(vt.sum()) < 1e-13                               # int u u_x over the periodic line
    # a steady state of the scheme: constant in space and time -> f = 0 (M (u_t+1 - u_t) + dt (nu G u + adv) = 0)
    f0, _ = O.burgers_f_and_J(ns, nt, dt, nu, np.full(ns * nt, -0.3))
    assert np.max(np.abs(f0)) < 1e-15
    with pytest.raises(pkg.GmrfError) as e:
        b.tangent(w)
    assert e.value.status == pkg._cabi.ERR_NO_DEVICE


def test_burgers_p2_line_pattern_and_oracle_without_gpu(pkg):
    """The quadratic periodic line of the reference's Burgers scripts (src/utils.jl:42-49): the library's J pattern (device = -1)
    equals the oracle's (vertex rows 10 entries, midpoint rows 6); the P2 restatement is pinned by what it computes: its
    matrix is the derivative of its residual, constants have zero residual, the P2 element mass matrix has its closed form
    h / 30 [4 -1 2; -1 4 2; 2 2 16], the stiffness annihilates constants and reproduces int (x^2)' phi_i' exactly."""
    from oracle import bt_oracle as O
    ns, nt, dt, nu = 12, 4, 0.05, 0.01 / np.pi
    rng = np.random.default_rng(2)
    w = rng.standard_normal(ns * nt)
    f, J = O.burgers_f_and_J(ns, nt, dt, nu, w, order=2)
    b = pkg.BurgersP1Tangent(ns, nt, dt, nu, device=-1, order=2)
    assert b.pattern.shape == J.shape and b.nnz == J.nnz == 8 * (nt - 1) * ns
    assert np.array_equal(b.pattern.indptr, J.indptr) and np.array_equal(b.pattern.indices, J.indices)
    h = 2.0 / ns
    M, G = O.assemble_burgers_mass_diffusion_matrices(ns, 2)
    assert abs(M[2, 2] - 8 * h / 30) < 1e-16 and abs(M[3, 3] - 16 * h / 30) < 1e-16 and abs(M[2, 3] - 2 * h / 30) < 1e-16
    assert abs(M[2, 4] + h / 30) < 1e-16 and abs(M[0, ns - 2] + h / 30) < 1e-16 and abs(M.sum() - 1.0) < 1e-14
    assert abs(G @ np.ones(ns)).max() < 1e-12
    u = w[:ns]
    Gt, vt = O.assemble_burgers_advection_matrix(ns, u, 2)
    num = np.zeros((ns, ns))
    for j in range(ns):
        e = np.zeros(ns); e[j] = 1e-6
        num[:, j] = (O.assemble_burgers_advection_matrix(ns, u + e, 2)[1] - O.assemble_burgers_advection_matrix(ns, u - e, 2)[1]) / 2e-6
    assert np.max(np.abs(Gt.toarray() - num)) < 1e-8
    assert np.max(np.abs(O.assemble_burgers_advection_matrix(ns, np.full(ns, 0.7), 2)[1])) < 1e-15
    f0, _ = O.burgers_f_and_J(ns, nt, dt, nu, np.full(ns * nt, -0.3), order=2)
    assert np.max(np.abs(f0)) < 1e-15
    with pytest.raises(pkg.GmrfError):
        pkg.BurgersP1Tangent(7, nt, dt, nu, device=-1, order=2)            # the quadratic line has two dofs per cell


def test_shallow_water_p1_patterns_and_oracle_without_gpu(pkg):
    """The shallow-water restatement (oracle/bt_oracle.py, /root/reference/src/spdes/shallow_water.jl:17-122, :170-217) is
    unpinned against Ferrite (absent); it is pinned by what the operators are: lumped mass integrates 1 per field, the
    stiffness annihilates constants, the h-u / h-v coupling is minus the weak gradient weighted by H (exact for linear
    fields on P1), u-u = k times the consistent mass, u-v = -f mass, K's u-h block = -g (weak gradient)'; the library's
    patterns (device = -1: no GPU) are the oracle's, its quadrature points the oracle's; Q_matern = J'J is symmetric
    positive definite and equals ratio K_m' M~^-1 K_m."""
    from oracle import bt_oracle as O
    nx, ny = 7, 6
    sw = pkg.ShallowWaterP1(nx, ny, device=-1)
    qp = O.shallow_water_qpoints(nx, ny)
    assert np.max(np.abs(sw.qpoints - qp)) < 1e-16
    H = 2.0 + 0.5 * qp[:, :, 0] - 0.25 * qp[:, :, 1]
    kk, ff, gg = 0.3, 0.7, 9.81
    K, M, S = O.assemble_shallow_water_system(nx, ny, H, kk, ff, gg)
    assert np.array_equal(sw.pattern_K.indptr, K.indptr) and np.array_equal(sw.pattern_K.indices, K.indices)
    assert np.array_equal(sw.pattern_S.indptr, S.indptr) and np.array_equal(sw.pattern_S.indices, S.indices)
    nn = nx * ny
    one = lambda fld: np.where(np.arange(3 * nn) % 3 == fld, 1.0, 0.0)
    assert abs(M.sum() - 3.0) < 1e-13 and abs(S @ np.ones(3 * nn)).max() < 1e-12
    # a constant velocity field is divergence free in the weak sense up to the boundary term: sum_i of (K e_u)[h_i] = -int H dx(sum phi_i) = 0
    assert abs((K @ one(1))[0::3].sum()) < 1e-12 and abs((K @ one(2))[0::3].sum()) < 1e-12
    assert np.max(np.abs((K @ one(1))[1::3] - kk * M[1::3])) < 1e-14          # u-u rows: k * mass (row sums of the consistent mass)
    assert np.max(np.abs((K @ one(2))[1::3] + ff * M[1::3])) < 1e-14          # u-v: -f * mass
    assert np.max(np.abs((K @ one(1))[2::3] - ff * M[2::3])) < 1e-14          # v-u: +f * mass
    # u-h block: -g int dx(phi_i) phi_j; applied to the nodal values of x it gives -g int dx(phi_i) x: summed over i zero again,
    # and for H = 1 the h-u block is (1 / g) times the u-h block
    K1, _, _ = O.assemble_shallow_water_system(nx, ny, np.ones_like(H), kk, ff, gg)
    Kd = K1.toarray()
    assert np.max(np.abs(Kd[0::3, 1::3] - Kd[1::3, 0::3] / gg)) < 1e-13
    # constraints: apply! zeroes rows and columns and puts meandiag on the diagonal
    pres = np.zeros(3 * nn, dtype=bool); pres[[1, 2, 3 * nn - 1, 40]] = True
    Kc, Mc, Sc = O.assemble_shallow_water_system(nx, ny, H, kk, ff, gg, prescribed=pres)
    assert np.all(Sc[pres].toarray()[:, ~pres] == 0) and np.allclose(Sc.diagonal()[pres], np.abs(S.diagonal()).sum() / (3 * nn))
    ops = O.shallow_water_operators(Kc, Mc, Sc, pres, kappa_matern=2.0, tau=0.5, dt=0.1)
    Q = (ops["J"].T @ ops["J"]).toarray()
    Km = ops["K_matern"].toarray()
    assert np.max(np.abs(Q - ops["ratio"] * Km.T @ np.diag(1.0 / ops["M_tilde"]) @ Km)) < 1e-12 * np.max(np.abs(Q))
    assert np.min(np.linalg.eigvalsh(0.5 * (Q + Q.T))) > 0
    assert np.allclose(ops["beta"][pres], np.sqrt(0.1) * 1e-2) and np.allclose(ops["beta"][~pres], np.sqrt(0.1) * 0.5)


def test_host_side_under_address_sanitizer(pkg, tmp_path):
    """SURVEY section 5 ("Race detection / sanitizers"): the host side of the library -- symbolic phase of the factor (block
    split, band check, staircase, tile plans), posterior-assembler symbolic phase, FEM patterns, argument validation -- is
    compiled with -fsanitize=address,undefined (device code as usual) and driven through the entry points that need no GPU,
    in a child process with the sanitizer runtime preloaded.  A report aborts the child."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "diffeqgmrfs.jl_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "libgmrf_hip_asan.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    rt = subprocess.run(["make", "-s", "-C", csrc, "asan-runtime"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=77", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py"), os.path.join(csrc, "libgmrf_hip_asan.so")],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and "asan driver ok" in p.stdout and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, \
        (p.returncode, p.stdout[-500:], p.stderr[-3000:])


def test_bench_refuses_a_rank_count_that_differs_from_the_launcher():
    """bench.py under a launcher (WORLD_SIZE set): --gpus that differs from it is a configuration error, exit code 2,
    before anything touches a GPU; so is a launcher-less environment asking for ranks it is not given (no WORLD_SIZE
    and --gpus 1 is the plain single-GPU run and is not exercised here: it needs the GPU)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr and p.stdout.strip() == ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 2          # default --gpus 1 under a 2-rank launcher


def test_darcy_p2_triangle_pattern_and_oracle_without_gpu(pkg):
    """Quadratic triangles (the reference's own element, src/utils.jl:32-33): the restatement
    oracle/bt_oracle.py `assemble_darcy_diff_matrix_p2` of /root/reference/src/problems/darcy.jl:5-63 is unpinned against
    Ferrite (absent) and pinned by what G is: symmetric, constants in its null space, u' G u = int a |grad u|^2 exactly for
    quadratic u and piecewise-constant a that the quadrature points resolve (the degree-3 rule integrates the degree-2
    integrand), f = beta int N_i (vertex functions integrate to 0, edge functions to |T| / 3); the library's pattern
    (device = -1: no GPU) is the oracle's; `apply!` semantics on the boundary lattice points."""
    from oracle import bt_oracle as O
    nx, ny = 6, 5
    W, H = 2 * nx - 1, 2 * ny - 1
    ng = 11
    xc = np.linspace(0.0, 1.0, ng)
    G, f = O.assemble_darcy_diff_matrix_p2(nx, ny, xc, xc, np.ones((ng, ng)), beta=2.0, constrain=False)
    d = pkg.DarcyP1Assembler(nx, ny, device=-1, order=2)
    assert d.n == W * H and np.array_equal(d.pattern.indptr, G.indptr) and np.array_equal(d.pattern.indices, G.indices)
    assert abs(G - G.T).max() < 1e-14 and abs(G @ np.ones(W * H)).max() < 1e-13
    I, J = np.arange(W * H) % W, np.arange(W * H) // W
    x, y = I / (W - 1), J / (H - 1)
    for u, energy in ((x, 1.0), (y, 1.0), (x * x, 4.0 / 3.0), (x * y, 2.0 / 3.0), (x * x - 2.0 * y * y + x * y, 19.0 / 3.0)):
        assert abs(u @ (G @ u) - energy) < 1e-12
    assert abs(f.sum() - 2.0) < 1e-13
    vertex = (I % 2 == 0) & (J % 2 == 0)
    assert np.max(np.abs(f[vertex])) < 1e-15                       # int N_vertex = 0 on every cell for the quadratic Lagrange basis
    # a coefficient that is 3 on the left half and 12 on the right (the reference's two-phase medium): energy of u = x
    # (table fine enough that the nearest-grid-point lookup of no quadrature point crosses x = 1/2: they keep 0.05 from it)
    xf = np.linspace(0.0, 1.0, 41)
    tab = np.where(xf[:, None] < 0.5, 3.0, 12.0) * np.ones((1, 41))
    Ga, _ = O.assemble_darcy_diff_matrix_p2(5, 5, xf, xf, tab, constrain=False)       # cell edges at multiples of 1/4: x = 1/2 is a cell boundary
    Ia = np.arange(81) % 9
    assert abs((Ia / 8.0) @ (Ga @ (Ia / 8.0)) - 7.5) < 1e-12
    # constraints: boundary lattice points
    Gc, fc = O.assemble_darcy_diff_matrix_p2(nx, ny, xc, xc, np.ones((ng, ng)), beta=2.0)
    bnd = (I == 0) | (J == 0) | (I == W - 1) | (J == H - 1)
    assert np.all(fc[bnd] == 0) and np.all(Gc[bnd].toarray()[:, ~bnd] == 0)
    assert np.allclose(Gc.diagonal()[bnd], np.abs(G.diagonal()).sum() / (W * H))
