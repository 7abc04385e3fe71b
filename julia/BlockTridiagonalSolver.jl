# BlockTridiagonalSolver.jl -- solver blueprint that plugs the MI355X block-tridiagonal path into
# GaussianMarkovRandomFields.jl (SURVEY 8f rank 3), so that the reference's scripts reach it by
# swapping ONE constructor:
#
#     scripts/darcy/solve_darcy_gmrf-fem.jl:100,174
#         CholeskySolverBlueprint(var_strategy = RBMCStrategy(50; rng), perm = p)
#      -> BlockTridiagonalSolverBlueprint(N_blocks; var_strategy = :rbmc, n_var_samples = 50)
#     scripts/burgers/solve_burgers_gmrf-fem.jl:169-170
#         GNCholeskySolverBlueprint(p)
#      -> BlockTridiagonalSolverBlueprint(N_t)          (time-major ordering: one block per time step, :133)
#
# `N_blocks` plays the role the fill-reducing permutation `p` plays for CHOLMOD
# (scripts/darcy/solve_darcy_gmrf-fem.jl:166-174): it is chosen once per mesh and re-used for every
# problem; the library validates that the posterior precision is block tridiagonal in that partition
# (GMRF_ERR_BAND otherwise).  The sparsity pattern is analysed once per solver chain: a solver built
# from a previous one (`reuse = s`) re-factors VALUES ONLY (gmrf_bt_refactor_values), which is what the
# per-problem loop (:176-198) and the Gauss-Newton loop (scripts/solve_burger.jl:143-149) need; the reused solver is
# CONSUMED (its handle moves to the new solver; using it afterwards raises).
#
# STATUS.  GaussianMarkovRandomFields.jl is not vendored in the reference (Project.toml:19, installed
# from a GitHub URL, unpinned) and there is no Julia in the build image: the solver interface below
# (AbstractSolver / AbstractSolverBlueprint, construct_solver, construct_conditional_solver,
# compute_mean, compute_variance, compute_rand!, gmrf_precision) is written from the call sites in
# this repository and from recollection of that package, and has never been executed.  It contains
# no `ccall`: everything goes through julia/DiffEqGMRFsHIP.jl, whose calls are checked statically
# against include/gmrf_hip.h (tests/test_host_logic.py).  The executed twin of this file is
# `api.ConditionedGMRF` / `api.gn_step` (diffeqgmrfs.jl_amd/api.py, tests/test_gpu_parity.py).
module BlockTridiagonalSolvers      # (plural: the solver type below is exported as `BlockTridiagonalSolver`)

using SparseArrays, LinearAlgebra, Random
import ..DiffEqGMRFsHIP as HIP
import GaussianMarkovRandomFields
import GaussianMarkovRandomFields: AbstractSolver, AbstractSolverBlueprint, construct_solver,
                                   construct_conditional_solver, compute_mean, compute_variance, compute_rand!,
                                   gmrf_precision, to_matrix

export BlockTridiagonalSolverBlueprint, BlockTridiagonalSolver, gn_step!, sqmahal, nll

"""
    BlockTridiagonalSolverBlueprint(N_blocks; var_strategy = :exact, n_var_samples = 50, seed = 0x5EED, device = 0, keep_l = false)

`var_strategy`: `:exact` (block-tridiagonal selected inversion -- deterministic, same O(N bs^3) work as the
factor), `:rbmc` (the reference's RBMCStrategy(n_var_samples)), `:mc`.
"""
struct BlockTridiagonalSolverBlueprint <: AbstractSolverBlueprint
    n_blocks::Int
    var_strategy::Symbol
    n_var_samples::Int
    seed::UInt64
    device::Int
    keep_l::Bool
end
BlockTridiagonalSolverBlueprint(n_blocks::Integer; var_strategy::Symbol = :exact, n_var_samples::Integer = 50,
                                seed::Integer = 0x5EED, device::Integer = 0, keep_l::Bool = false) =
    BlockTridiagonalSolverBlueprint(Int(n_blocks), var_strategy, Int(n_var_samples), UInt64(seed), Int(device), keep_l)

mutable struct BlockTridiagonalSolverState <: AbstractSolver
    bp::BlockTridiagonalSolverBlueprint
    mean::Vector{Float64}                              # posterior mean (computed in the constructor: one ldiv)
    precision::SparseMatrixCSC{Float64,Int}
    precision_chol::HIP.TridiagonalCholeskyFactor      # what the scripts reach through x.solver_ref[].precision_chol
    csr::Union{Nothing,HIP.GmrfCsr}                    # device copy of the precision for RBMC (K6), built on demand
    computed_var::Union{Nothing,Vector{Float64}}
    n_draws::Int                                       # sample ids handed out so far (Philox stream position)
    consumed::Bool                                     # a later solver re-factored this one's handle (`reuse = s`): s is spent
end
const BlockTridiagonalSolver = BlockTridiagonalSolverState
_live(s::BlockTridiagonalSolverState) =
    s.consumed ? error("this solver was passed as `reuse` to a later one: its factor handle now holds the later problem's values") : s

_sparse(Q) = Q isa SparseMatrixCSC{Float64,Int} ? Q : SparseMatrixCSC{Float64,Int}(to_matrix(Q))

function _factor(bp::BlockTridiagonalSolverBlueprint, Q::SparseMatrixCSC{Float64,Int}, reuse)
    if reuse !== nothing && size(reuse.precision) == size(Q) && reuse.precision.colptr == Q.colptr && reuse.precision.rowval == Q.rowval
        # same pattern: values only, captured HIP graphs re-used.  The handle MOVES to the new solver: `reuse` is
        # consumed (its mean / variances / draws would silently come from the new factor otherwise)
        _live(reuse).consumed = true
        return HIP.refactor!(reuse.precision_chol, Q)
    end
    F = HIP.TridiagonalCholeskyFactor{Float64}(bp.device)
    bp.keep_l || HIP.set_keep_l!(F, false)
    return HIP.tridiagonal_cholesky(Q, bp.n_blocks; F = F)
end

"Solver of an unconditional GMRF N(mean, precision^-1)."
function construct_solver(bp::BlockTridiagonalSolverBlueprint, mean::AbstractVector, precision; reuse = nothing)
    Q = _sparse(precision)
    return BlockTridiagonalSolverState(bp, Vector{Float64}(mean), Q, _factor(bp, Q, reuse), nothing, nothing, 0, false)
end

"""
Solver of x | y for y = A x + b + eps, eps ~ N(0, Q_eps^-1) -- what `condition_on_observations(x, A, Q_eps, y)`
builds (scripts/darcy/solve_darcy_gmrf-fem.jl:188-189): posterior precision `Q + A' Q_eps A` (given), mean
`mu + Q_post^-1 A' Q_eps (y - A mu - b)`: ONE forward + backward sweep on the fresh factor.
"""
function construct_conditional_solver(bp::BlockTridiagonalSolverBlueprint, prior_mean::AbstractVector, posterior_precision,
                                      A, Q_eps, y::AbstractVector, b = nothing; reuse = nothing)
    Q = _sparse(posterior_precision)
    F = _factor(bp, Q, reuse)
    r = y .- A * prior_mean
    b === nothing || (r .-= b)
    rhs = Vector{Float64}(A' * (Q_eps isa Number ? Q_eps .* r : Q_eps * r))
    mean = prior_mean .+ HIP.ldiv(F, rhs)
    return BlockTridiagonalSolverState(bp, Vector{Float64}(mean), Q, F, nothing, nothing, 0, false)
end

gmrf_precision(s::BlockTridiagonalSolverState) = s.precision
compute_mean(s::BlockTridiagonalSolverState) = _live(s).mean

"`rand(rng, x)`: mean + L^-T z.  z comes from the device Philox stream (seed of the blueprint, sample id = draw count) so that the draw does not depend on the number of GPUs; `rng` is accepted for interface compatibility."
function compute_rand!(s::BlockTridiagonalSolverState, rng::Random.AbstractRNG, x::AbstractVector)
    _live(s)
    x .= vec(HIP.sample(s.precision_chol, 1; mean = s.mean, seed = s.bp.seed, first_id = s.n_draws))
    s.n_draws += 1
    return x
end

"`var(x)` / `std(x)` (scripts/darcy/solve_darcy_gmrf-fem.jl:192)."
function compute_variance(s::BlockTridiagonalSolverState)
    _live(s)
    s.computed_var === nothing || return s.computed_var
    if s.bp.var_strategy === :exact
        s.computed_var = HIP.marginal_var(s.precision_chol)
    else
        s.bp.var_strategy === :rbmc && s.csr === nothing && (s.csr = HIP.GmrfCsr(s.precision; device = s.bp.device))
        s.computed_var = HIP.marginal_var(s.precision_chol; method = s.bp.var_strategy, k = s.bp.n_var_samples,
                                          seed = s.bp.seed + 1, Q = s.csr)
    end
    return s.computed_var
end

"log det of the precision: 2 sum log diag(L_i) (scripts/burgers/solve_burgers_gmrf-collocation.jl:208-211)."
LinearAlgebra.logdet(s::BlockTridiagonalSolverState) = logdet(_live(s).precision_chol)

"`sqmahal(x, z)` = (z - mean)' Q (z - mean) (scripts/burgers/solve_burgers_gmrf-collocation.jl:262): one CSR SpMV on the device (K6) and a dot product."
function sqmahal(s::BlockTridiagonalSolverState, z::AbstractVector)
    _live(s)
    s.csr === nothing && (s.csr = HIP.GmrfCsr(s.precision; device = s.bp.device))
    d = Vector{Float64}(z) .- s.mean
    return dot(d, s.csr * d)
end

"Negative log-likelihood of z, `nll_soln` of the same script (:213-215): 0.5 (n log 2pi + sqmahal + logdet Sigma), logdet Sigma = -logdet Q."
nll(s::BlockTridiagonalSolverState, z::AbstractVector) = 0.5 * (length(z) * log(2pi) + sqmahal(s, z) - logdet(s))

"""
    gn_step!(s, Q, Qx_prior, J, x, obs_diff, noise)  ->  new iterate

One Gauss-Newton step of scripts/solve_burger.jl:143-149 (`A = Symmetric(Q + noise * J' * J)`,
`rhs = Q * x_prior + noise * J' * (J * x + obs_diff)`, `cholesky(A; perm) \\ rhs`) with the assembly on the device:
the assembler is created on the first call (symbolic phase), later calls move values only.
"""
mutable struct GaussNewtonWorkspace
    as::Union{Nothing,HIP.PosteriorAssembler}
    F::Union{Nothing,HIP.TridiagonalCholeskyFactor}
    vals::Vector{Float64}
    rhs::Vector{Float64}
end
GaussNewtonWorkspace() = GaussNewtonWorkspace(nothing, nothing, Float64[], Float64[])

function gn_step!(ws::GaussNewtonWorkspace, bp::BlockTridiagonalSolverBlueprint, Q::SparseMatrixCSC{Float64,Int},
                  Qx_prior::Vector{Float64}, J::SparseMatrixCSC{Float64,Int}, x::Vector{Float64}, obs_diff::Vector{Float64}, noise::Real)
    Jt = SparseMatrixCSC(J')                                     # CSR of J: the value order the assembler expects
    if ws.as === nothing
        ws.as = HIP.PosteriorAssembler(Q, J; device = bp.device)
        ws.vals = Vector{Float64}(undef, nnz(ws.as.pattern)); ws.rhs = Vector{Float64}(undef, size(Q, 1))
    end
    HIP.precision!(ws.vals, ws.as, Q.nzval, Jt.nzval, noise)
    if ws.F === nothing
        A = copy(ws.as.pattern); A.nzval .= ws.vals
        F = HIP.TridiagonalCholeskyFactor{Float64}(bp.device)
        bp.keep_l || HIP.set_keep_l!(F, false)
        ws.F = HIP.tridiagonal_cholesky(A, bp.n_blocks; F = F)
    else
        HIP.refactor!(ws.F, ws.vals)
    end
    HIP.rhs!(ws.rhs, ws.as, Qx_prior, Jt.nzval, x, obs_diff, noise)
    return HIP.ldiv(ws.F, ws.rhs)
end

end # module
