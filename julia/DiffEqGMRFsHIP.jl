# DiffEqGMRFsHIP.jl -- `ccall` shim that puts libgmrf_hip.so (MI355X / gfx950) behind the
# block-tridiagonal Cholesky API of DiffEqGMRFs.jl.
#
# It REPLACES the definitions of src/tridiagonal_cholesky.jl (reference file:line in
# brackets) one for one, same names and argument order:
#
#     TridiagonalCholeskyFactor           [:5-9]    -> handle wrapper, fields N / chos / Cs lazily
#     tridiagonal_cholesky(A, N_blocks)   [:65-82]  -> gmrf_bt_factor_csc
#     forward_solve(L, b)                 [:43-52]  -> gmrf_bt_solve(mode = FORWARD)
#     backward_solve(L, b)                [:24-33]  -> gmrf_bt_solve(mode = BACKWARD)
#     ldiv!(y, L, b), ldiv(L, b)          [:54-63]  -> gmrf_bt_solve(mode = FULL)
#     extract_blocks output               [scripts/solve_burger.jl:182-254] -> gmrf_bt_factor_blocks
#
# Usage inside the package: replace `include("tridiagonal_cholesky.jl")` in src/DiffEqGMRFs.jl:11
# by `include("DiffEqGMRFsHIP.jl")` (or load this file after the package and `using .DiffEqGMRFsHIP`).
#
# NOTE: there is no Julia toolchain in the build image of this repository, so this file has
# been written against the C header (include/gmrf_hip.h) but never executed; the Python ctypes
# binding (diffeqgmrfs.jl_amd/_cabi.py) exercises exactly the same entry points and is what the
# test-suite runs.
module DiffEqGMRFsHIP

using SparseArrays, LinearAlgebra

export TridiagonalCholeskyFactor, tridiagonal_cholesky, PosteriorAssembler

const libgmrf = get(ENV, "LIBGMRF_HIP", joinpath(@__DIR__, "..", "diffeqgmrfs.jl_amd", "csrc", "libgmrf_hip.so"))

const GMRF_OK = Int32(0)
const GMRF_ERR_NOT_SPD = Int32(-1)
const SOLVE_FULL, SOLVE_FORWARD, SOLVE_BACKWARD = Int32(0), Int32(1), Int32(2)
const BLOCK_L, BLOCK_C, BLOCK_LINV = Int32(0), Int32(1), Int32(2)
const VAR_EXACT, VAR_RBMC, VAR_MC = Int32(0), Int32(1), Int32(2)

last_error() = unsafe_string(ccall((:gmrf_last_error, libgmrf), Cstring, ()))

function check(status::Int32, info::Integer = 0)
    status == GMRF_OK && return nothing
    status == GMRF_ERR_NOT_SPD && throw(PosDefException(info))      # what `cholesky` throws at :77
    error("libgmrf_hip status $status: $(last_error())")
end

"""
Device-resident factor.  `N` is the total size (as in the reference struct, :6); `chos[i]` and
`Cs[i]` are copied from the GPU on access.
"""
mutable struct TridiagonalCholeskyFactor{T}
    handle::Ptr{Cvoid}
    N::Int
    n_blocks::Int
    function TridiagonalCholeskyFactor{T}(device::Integer = 0) where {T}
        T === Float64 || error("libgmrf_hip is fp64 only")
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:gmrf_bt_create, libgmrf), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, h))
        obj = new{T}(h[], 0, 0)
        finalizer(o -> ccall((:gmrf_bt_destroy, libgmrf), Int32, (Ptr{Cvoid},), o.handle), obj)
        return obj
    end
end

block_size(F::TridiagonalCholeskyFactor) = F.N ÷ F.n_blocks

function get_block(F::TridiagonalCholeskyFactor, kind::Int32, i::Integer)
    bs = block_size(F)
    out = Matrix{Float64}(undef, bs, bs)
    check(ccall((:gmrf_bt_get_block, libgmrf), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Float64}, Int64),
                F.handle, kind, i - 1, out, bs))
    return out
end

# F.chos[i] -> Cholesky object of block i (uplo 'L'), F.Cs[i] -> dense L_{i+1,i}
struct LazyBlocks{K} <: AbstractVector{Any}
    F::TridiagonalCholeskyFactor
    n::Int
end
Base.size(v::LazyBlocks) = (v.n,)
Base.getindex(v::LazyBlocks{:chos}, i::Int) = Cholesky(get_block(v.F, BLOCK_L, i), 'L', 0)
Base.getindex(v::LazyBlocks{:Cs}, i::Int) = get_block(v.F, BLOCK_C, i)
function Base.getproperty(F::TridiagonalCholeskyFactor, s::Symbol)
    s === :chos && return LazyBlocks{:chos}(F, getfield(F, :n_blocks))
    s === :Cs && return LazyBlocks{:Cs}(F, getfield(F, :n_blocks) - 1)
    return getfield(F, s)
end

"""
    tridiagonal_cholesky(A::SparseMatrixCSC, N_blocks)

[src/tridiagonal_cholesky.jl:65-82]  Only the lower blocks (i,i) and (i,i-1) of `A` are read.
Throws `PosDefException(block)` like `cholesky` does at :77.
"""
function tridiagonal_cholesky(A::SparseMatrixCSC{Float64,Int}, N_blocks::Integer; device::Integer = 0)
    n = size(A, 1)
    n % N_blocks == 0 || throw(DimensionMismatch("size(A,1) must be a multiple of N_blocks"))
    F = TridiagonalCholeskyFactor{Float64}(device)
    info = Ref{Int32}(0)
    GC.@preserve A begin
        st = ccall((:gmrf_bt_factor_csc, libgmrf), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int32, Ref{Int32}),
                   F.handle, n, N_blocks, A.colptr, A.rowval, A.nzval, 1, info)   # index_base = 1
    end
    check(st, info[])
    F.N = n
    F.n_blocks = N_blocks
    return F
end

"Re-factor with new values on the same sparsity pattern (Gauss-Newton loop, scripts/solve_burger.jl:143-149)."
function refactor!(F::TridiagonalCholeskyFactor, A::SparseMatrixCSC{Float64,Int})
    info = Ref{Int32}(0)
    st = GC.@preserve A ccall((:gmrf_bt_refactor_values, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int32}),
                              F.handle, A.nzval, info)
    check(st, info[])
    return F
end

function _solve(F::TridiagonalCholeskyFactor, b::StridedVecOrMat{Float64}, mode::Int32, y = similar(b))
    size(b, 1) == F.N || throw(DimensionMismatch())
    k = size(b, 2)
    GC.@preserve b y check(ccall((:gmrf_bt_solve, libgmrf), Int32,
                                 (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int32),
                                 F.handle, b, y, k, stride(b, 2) == 0 ? F.N : max(stride(b, 2), F.N), mode))
    return y
end

forward_solve(L::TridiagonalCholeskyFactor, b) = _solve(L, b, SOLVE_FORWARD)      # [:43-52]
backward_solve(L::TridiagonalCholeskyFactor, b) = _solve(L, b, SOLVE_BACKWARD)    # [:24-33]
ldiv!(y, L::TridiagonalCholeskyFactor, b) = _solve(L, b, SOLVE_FULL, y)           # [:54-58]
ldiv(L::TridiagonalCholeskyFactor, b) = ldiv!(similar(b), L, b)                   # [:60-63]

"k samples mean + L^-T z (rand(rng, x_cond), scripts/darcy/solve_darcy_gmrf-fem.jl:191)."
function sample(F::TridiagonalCholeskyFactor, k::Integer; mean = nothing, seed::Integer = 0x5EED, first_id::Integer = 0)
    out = Matrix{Float64}(undef, F.N, k)
    mp = mean === nothing ? Ptr{Float64}(C_NULL) : pointer(mean)
    GC.@preserve mean out check(ccall((:gmrf_bt_sample, libgmrf), Int32,
        (Ptr{Cvoid}, UInt64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64),
        F.handle, seed, first_id, k, mp, C_NULL, out, F.N))
    return out
end

"Marginal variances diag(A^-1): exact selected inversion (std(x_cond), solve_darcy_gmrf-fem.jl:192)."
function marginal_var(F::TridiagonalCholeskyFactor)
    out = Vector{Float64}(undef, F.N)
    check(ccall((:gmrf_bt_marginal_var, libgmrf), Int32, (Ptr{Cvoid}, Int32, Int64, UInt64, Ptr{Cvoid}, Ptr{Float64}),
                F.handle, VAR_EXACT, 0, 0, C_NULL, out))
    return out
end

function LinearAlgebra.logdet(F::TridiagonalCholeskyFactor)
    v = Ref{Float64}(0.0)
    check(ccall((:gmrf_bt_logdet, libgmrf), Int32, (Ptr{Cvoid}, Ref{Float64}), F.handle, v))
    return v[]
end

# --- Gauss-Newton assembly on the device (gn_step, scripts/solve_burger.jl:143-149) -----------------
"Symbolic phase for A = Q + noise * J' * J with fixed patterns; J is passed through its transpose's CSC arrays (= CSR of J)."
mutable struct PosteriorAssembler
    handle::Ptr{Cvoid}
    pattern::SparseMatrixCSC{Float64,Int}     # values 1.0; nzval order = output order of precision!
end

function PosteriorAssembler(Q::SparseMatrixCSC{Float64,Int}, J::SparseMatrixCSC{Float64,Int}; device::Integer = 0)
    Jt = SparseMatrixCSC(J')                   # CSC of J' = CSR of J; its nzval order is what the numeric calls expect
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Q Jt check(ccall((:gmrf_assemble_create, libgmrf), Int32,
        (Int32, Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Int32, Ref{Ptr{Cvoid}}),
        device, C_NULL, size(Q, 1), Q.colptr, Q.rowval, size(J, 1), Jt.colptr, Jt.rowval, 1, h))
    nnz_out = Ref{Int64}(0); nprod = Ref{Int64}(0)
    check(ccall((:gmrf_assemble_pattern, libgmrf), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ptr{Int64}, Ptr{Int64}, Int32),
                h[], nnz_out, nprod, C_NULL, C_NULL, 1))
    colptr = Vector{Int64}(undef, size(Q, 1) + 1); rowval = Vector{Int64}(undef, nnz_out[])
    check(ccall((:gmrf_assemble_pattern, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int32),
                h[], C_NULL, C_NULL, colptr, rowval, 1))
    as = PosteriorAssembler(h[], SparseMatrixCSC(size(Q, 1), size(Q, 1), colptr, rowval, ones(nnz_out[])))
    finalizer(a -> ccall((:gmrf_assemble_destroy, libgmrf), Int32, (Ptr{Cvoid},), a.handle), as)
    return as
end

"nzval of Q + noise * J' * J on `as.pattern` (host vectors here; device pointers work the same way)."
function precision!(out::Vector{Float64}, as::PosteriorAssembler, q_nzval::Vector{Float64}, jt_nzval::Vector{Float64}, noise::Real)
    GC.@preserve out q_nzval jt_nzval check(ccall((:gmrf_assemble_precision, libgmrf), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}), as.handle, q_nzval, jt_nzval, Float64(noise), out))
    return out
end

"rhs = base + noise * J' * (J * x + obs_diff)"
function rhs!(out::Vector{Float64}, as::PosteriorAssembler, base, jt_nzval, x, obs_diff, noise::Real)
    GC.@preserve out base jt_nzval x obs_diff check(ccall((:gmrf_assemble_rhs, libgmrf), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}),
        as.handle, base, jt_nzval, x, obs_diff, Float64(noise), out))
    return out
end

end # module
