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
#     Q * x                               [scripts/solve_burger.jl:157-158,166,177] -> gmrf_spmm (GmrfCsr)
#
# and binds the rest of the C ABI (include/gmrf_hip.h): batches of problems on one pattern,
# samples / variances / log-determinant, factor export / import, the posterior assembler and the
# RCCL communicator that shares a factor between the GPUs of a node.
#
# Usage inside the package: replace `include("tridiagonal_cholesky.jl")` in src/DiffEqGMRFs.jl:11
# by `include("DiffEqGMRFsHIP.jl")` (or load this file after the package and `using .DiffEqGMRFsHIP`).
#
# NOTE: there is no Julia toolchain in the build image of this repository, so this file has
# been written against the C header (include/gmrf_hip.h) but never executed; every `ccall` is
# checked statically against the header (symbol, argument count, C types;
# tests/test_host_logic.py::test_julia_shim_ccalls_match_the_header, which also requires every
# non-test export to be bound here), and the Python ctypes binding (diffeqgmrfs.jl_amd/_cabi.py)
# exercises exactly the same entry points in the test-suite.
module DiffEqGMRFsHIP

using SparseArrays, LinearAlgebra

export TridiagonalCholeskyFactor, tridiagonal_cholesky, forward_solve, backward_solve, ldiv, PosteriorAssembler, GmrfCsr,
       GmrfComm, DarcyP1Assembler, BurgersP1Tangent

const libgmrf = get(ENV, "LIBGMRF_HIP", joinpath(@__DIR__, "..", "diffeqgmrfs.jl_amd", "csrc", "libgmrf_hip.so"))

const GMRF_OK = Int32(0)
const GMRF_ERR_NOT_SPD = Int32(-1)
const SOLVE_FULL, SOLVE_FORWARD, SOLVE_BACKWARD = Int32(0), Int32(1), Int32(2)
const BLOCK_L, BLOCK_C, BLOCK_LINV = Int32(0), Int32(1), Int32(2)
const VAR_EXACT, VAR_RBMC, VAR_MC = Int32(0), Int32(1), Int32(2)

last_error() = unsafe_string(ccall((:gmrf_last_error, libgmrf), Cstring, ()))
version() = ccall((:gmrf_version, libgmrf), Int32, ())

function check(status::Int32, info::Integer = 0)
    status == GMRF_OK && return nothing
    status == GMRF_ERR_NOT_SPD && throw(PosDefException(info))      # what `cholesky` throws at :77
    error("libgmrf_hip status $status: $(last_error())")
end

# ------------------------------------------------------------------------------------------ factor

"""
Device-resident factor.  `N` is the total size (as in the reference struct, :6); `chos[i]` and
`Cs[i]` are copied from the GPU on access.  `batch` > 1: that many independent problems on one
sparsity pattern, factored and solved in lock step (the reference's loop over data-set problems).
"""
mutable struct TridiagonalCholeskyFactor{T}
    handle::Ptr{Cvoid}
    N::Int
    n_blocks::Int
    batch::Int
    function TridiagonalCholeskyFactor{T}(device::Integer = 0; batch::Integer = 1, stream::Ptr{Cvoid} = C_NULL) where {T}
        T === Float64 || error("libgmrf_hip is fp64 only")
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:gmrf_bt_create, libgmrf), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, stream, h))
        obj = new{T}(h[], 0, 0, 1)
        finalizer(o -> ccall((:gmrf_bt_destroy, libgmrf), Int32, (Ptr{Cvoid},), o.handle), obj)
        batch == 1 || set_batch!(obj, batch)
        return obj
    end
end

block_size(F::TridiagonalCholeskyFactor) = F.N ÷ F.n_blocks

"B independent problems on one pattern (scripts/darcy/solve_darcy_gmrf-fem.jl:176-198) per call."
function set_batch!(F::TridiagonalCholeskyFactor, batch::Integer)
    check(ccall((:gmrf_bt_set_batch, libgmrf), Int32, (Ptr{Cvoid}, Int64), getfield(F, :handle), batch))
    setfield!(F, :batch, Int(batch))
    return F
end

"The problem of a batch that `F.chos`, `F.Cs`, `logdet`, `export_factor` address (0-based)."
select_problem!(F::TridiagonalCholeskyFactor, p::Integer) =
    (check(ccall((:gmrf_bt_select_problem, libgmrf), Int32, (Ptr{Cvoid}, Int64), getfield(F, :handle), p)); F)

"`keep = false`: the triangular blocks are not retained (sweeps, samples, variances, logdet do not need them)."
set_keep_l!(F::TridiagonalCholeskyFactor, keep::Bool) =
    (check(ccall((:gmrf_bt_set_keep_l, libgmrf), Int32, (Ptr{Cvoid}, Int32), getfield(F, :handle), keep ? 1 : 0)); F)

function get_block(F::TridiagonalCholeskyFactor, kind::Int32, i::Integer)
    bs = block_size(F)
    out = Matrix{Float64}(undef, bs, bs)
    check(ccall((:gmrf_bt_get_block, libgmrf), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Float64}, Int64),
                getfield(F, :handle), kind, i - 1, out, bs))
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
Throws `PosDefException(block)` like `cholesky` does at :77.  `nzvals`: a `nnz x batch` matrix of
values for a batch of problems on the pattern of `A` (with `batch` set on the factor).
"""
function tridiagonal_cholesky(A::SparseMatrixCSC{Float64,Int}, N_blocks::Integer; device::Integer = 0,
                              F::TridiagonalCholeskyFactor = TridiagonalCholeskyFactor{Float64}(device),
                              nzvals::StridedVecOrMat{Float64} = A.nzval)
    n = size(A, 1)
    n % N_blocks == 0 || throw(DimensionMismatch("size(A,1) must be a multiple of N_blocks"))
    length(nzvals) == nnz(A) * F.batch || throw(DimensionMismatch("nzvals must hold nnz x batch values"))
    info = Ref{Int32}(0)
    GC.@preserve A nzvals begin
        st = ccall((:gmrf_bt_factor_csc, libgmrf), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int32, Ref{Int32}),
                   F.handle, n, N_blocks, A.colptr, A.rowval, nzvals, 1, info)   # index_base = 1
    end
    check(st, info[])
    setfield!(F, :N, n)
    setfield!(F, :n_blocks, Int(N_blocks))
    return F
end

"Re-factor with new values on the same sparsity pattern (Gauss-Newton loop, scripts/solve_burger.jl:143-149)."
function refactor!(F::TridiagonalCholeskyFactor, nzval::StridedVecOrMat{Float64})
    info = Ref{Int32}(0)
    st = GC.@preserve nzval ccall((:gmrf_bt_refactor_values, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int32}),
                                  F.handle, nzval, info)
    check(st, info[])
    return F
end
refactor!(F::TridiagonalCholeskyFactor, A::SparseMatrixCSC{Float64,Int}) = refactor!(F, A.nzval)

struct SparseBlock                          # gmrf_sparse_block
    nnz::Int64
    ptr::Ptr{Int64}
    idx::Ptr{Int64}
    val::Ptr{Float64}
end

"""
    tridiagonal_cholesky(diag_blocks, off_diag_blocks)

Factor from the output of `extract_blocks` (scripts/solve_burger.jl:182-254): `N` sparse diagonal
blocks and `N-1` sparse LOWER off-diagonal blocks, each `bs x bs`.
"""
function tridiagonal_cholesky(diag::Vector{SparseMatrixCSC{Float64,Int}}, off::Vector{SparseMatrixCSC{Float64,Int}};
                              device::Integer = 0)
    nb = length(diag); bs = size(diag[1], 1)
    length(off) == nb - 1 || throw(DimensionMismatch("need N-1 off-diagonal blocks"))
    F = TridiagonalCholeskyFactor{Float64}(device)
    info = Ref{Int32}(0)
    GC.@preserve diag off begin
        d = [SparseBlock(nnz(b), pointer(b.colptr), pointer(b.rowval), pointer(b.nzval)) for b in diag]
        o = [SparseBlock(nnz(b), pointer(b.colptr), pointer(b.rowval), pointer(b.nzval)) for b in off]
        isempty(o) && push!(o, SparseBlock(0, C_NULL, C_NULL, C_NULL))
        st = ccall((:gmrf_bt_factor_blocks, libgmrf), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Ptr{SparseBlock}, Ptr{SparseBlock}, Int32, Int32, Ref{Int32}),
                   F.handle, nb * bs, nb, d, o, 1, 1, info)              # 1-based, compressed by column
    end
    check(st, info[])
    setfield!(F, :N, nb * bs)
    setfield!(F, :n_blocks, nb)
    return F
end

# --- solves.  b: Vector, Matrix (n x k) or a strided view of one; for a batch of B problems an n x (k B)
# matrix whose column groups [p k + 1 : (p + 1) k] belong to problem p.
function _solve(F::TridiagonalCholeskyFactor, b::StridedVecOrMat{Float64}, mode::Int32, y::StridedVecOrMat{Float64} = similar(b))
    size(b, 1) == F.N || throw(DimensionMismatch())
    size(y) == size(b) || throw(DimensionMismatch())
    (stride(b, 1) == 1 && stride(y, 1) == 1) || throw(ArgumentError("columns must be contiguous"))
    k = size(b, 2) ÷ F.batch
    ldb = ndims(b) == 1 ? F.N : max(stride(b, 2), F.N)
    ldy = ndims(y) == 1 ? F.N : max(stride(y, 2), F.N)
    GC.@preserve b y check(ccall((:gmrf_bt_solve, libgmrf), Int32,
                                 (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Int32),
                                 F.handle, b, y, k, ldb, ldy, mode))
    return y
end

# The reference returns a Vector of chunks from the two half solves (:51, :32) and feeds it to the other
# half as if it were flat (SURVEY 0.3 defect iii); here both return the flat vector / matrix.
forward_solve(L::TridiagonalCholeskyFactor, b) = _solve(L, b, SOLVE_FORWARD)      # [:43-52]
backward_solve(L::TridiagonalCholeskyFactor, b) = _solve(L, b, SOLVE_BACKWARD)    # [:24-33]
ldiv!(y, L::TridiagonalCholeskyFactor, b) = _solve(L, b, SOLVE_FULL, y)           # [:54-58]
ldiv(L::TridiagonalCholeskyFactor, b) = ldiv!(similar(b), L, b)                   # [:60-63]

"k samples mean + L^-T z per problem (rand(rng, x_cond), scripts/darcy/solve_darcy_gmrf-fem.jl:191); z: given normals (n x k)."
function sample(F::TridiagonalCholeskyFactor, k::Integer; mean = nothing, z = nothing, seed::Integer = 0x5EED, first_id::Integer = 0)
    out = Matrix{Float64}(undef, F.N, k * F.batch)
    mp = mean === nothing ? Ptr{Float64}(C_NULL) : pointer(mean)
    zp = z === nothing ? Ptr{Float64}(C_NULL) : pointer(z)
    GC.@preserve mean z out check(ccall((:gmrf_bt_sample, libgmrf), Int32,
        (Ptr{Cvoid}, UInt64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64),
        F.handle, seed, first_id, k, mp, zp, out, F.N))
    return out
end

"""
`mean(x_cond)` and `rand(rng, x_cond, k)` of one factor in ONE call (scripts/darcy/solve_darcy_gmrf-fem.jl:190-191): returns
`(mean, samples)`, bitwise `ldiv(F, b)` and `sample(F, k; mean = ...)`.  On device arrays (pass `CuPtr`-like pointers through
`posterior!`) the samples' sweep runs beside the mean's two where the handle's sweeps are persistent launches.
"""
function posterior(F::TridiagonalCholeskyFactor, b::AbstractVector{Float64}, k::Integer; seed::Integer = 0x5EED, first_id::Integer = 0)
    mean = Vector{Float64}(undef, F.N)
    out = Matrix{Float64}(undef, F.N, k)
    GC.@preserve b mean out check(ccall((:gmrf_bt_posterior, libgmrf), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, UInt64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Int64),
        F.handle, b, seed, first_id, k, mean, out, F.N))
    return mean, out
end
"The same on raw (device) pointers: `b`, `mean` n doubles, `samples` n x k column-major with leading dimension `ld`."
posterior!(F::TridiagonalCholeskyFactor, b::Ptr{Float64}, mean::Ptr{Float64}, samples::Ptr{Float64}, k::Integer, ld::Integer;
           seed::Integer = 0x5EED, first_id::Integer = 0) =
    check(ccall((:gmrf_bt_posterior, libgmrf), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, UInt64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Int64),
        F.handle, b, seed, first_id, k, mean, samples, ld))

"The N(0,1) draws `sample` uses: Philox4x32-10 keyed by (seed, sample id, dof) -- independent of the GPU count."
function normals(F::TridiagonalCholeskyFactor, k::Integer; seed::Integer = 0x5EED, first_id::Integer = 0)
    out = Matrix{Float64}(undef, F.N, k * F.batch)
    check(ccall((:gmrf_bt_normals, libgmrf), Int32, (Ptr{Cvoid}, UInt64, Int64, Int64, Ptr{Float64}, Int64),
                F.handle, seed, first_id, k, out, F.N))
    return out
end

"""
Marginal variances diag(A^-1) (std(x_cond), solve_darcy_gmrf-fem.jl:192): `:exact` = block-tridiagonal
selected inversion; `:rbmc` = the reference's RBMCStrategy(k) (needs `Q::GmrfCsr`, the factored matrix);
`:mc`.  A batch returns an `n x batch` matrix (`:exact`).
"""
function marginal_var(F::TridiagonalCholeskyFactor; method::Symbol = :exact, k::Integer = 50, seed::Integer = 0x5EED, Q = nothing)
    m = method === :exact ? VAR_EXACT : method === :rbmc ? VAR_RBMC : VAR_MC
    out = Matrix{Float64}(undef, F.N, method === :exact ? F.batch : 1)
    qh = Q === nothing ? C_NULL : Q.handle
    check(ccall((:gmrf_bt_marginal_var, libgmrf), Int32, (Ptr{Cvoid}, Int32, Int64, UInt64, Ptr{Cvoid}, Ptr{Float64}),
                F.handle, m, k, seed, qh, out))
    return size(out, 2) == 1 ? vec(out) : out
end

"RBMC / MC variances of every problem of a batch; `q_vals`: nnz x batch values in Q's pattern order."
function marginal_var_batch(F::TridiagonalCholeskyFactor, Q, q_vals::StridedVecOrMat{Float64}; method::Symbol = :rbmc, k::Integer = 50, seed::Integer = 0x5EED)
    out = Matrix{Float64}(undef, F.N, F.batch)
    GC.@preserve q_vals check(ccall((:gmrf_bt_marginal_var_batch, libgmrf), Int32,
        (Ptr{Cvoid}, Int32, Int64, UInt64, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        F.handle, method === :rbmc ? VAR_RBMC : VAR_MC, k, seed, Q.handle, q_vals, out))
    return out
end

"Sharded variance estimation: adds this rank's samples [first_id, first_id + k) to `acc` (all-reduce it afterwards)."
function var_accumulate!(acc::Vector{Float64}, F::TridiagonalCholeskyFactor, first_id::Integer, k::Integer; method::Symbol = :rbmc, seed::Integer = 0x5EED, Q = nothing)
    qh = Q === nothing ? C_NULL : Q.handle
    check(ccall((:gmrf_bt_var_accumulate, libgmrf), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, UInt64, Ptr{Cvoid}, Ptr{Float64}),
                F.handle, method === :rbmc ? VAR_RBMC : VAR_MC, first_id, k, seed, qh, acc))
    return acc
end

function LinearAlgebra.logdet(F::TridiagonalCholeskyFactor)
    v = Ref{Float64}(0.0)
    check(ccall((:gmrf_bt_logdet, libgmrf), Int32, (Ptr{Cvoid}, Ref{Float64}), F.handle, v))
    return v[]
end

# --- factor image: checkpoint / resume, and all of F.chos / F.Cs in one call
function export_factor(F::TridiagonalCholeskyFactor)
    nbytes = Ref{Int64}(0)
    check(ccall((:gmrf_bt_export_size, libgmrf), Int32, (Ptr{Cvoid}, Ref{Int64}), F.handle, nbytes))
    buf = Vector{UInt8}(undef, nbytes[])
    check(ccall((:gmrf_bt_export_factor, libgmrf), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int64), F.handle, buf, nbytes[]))
    return buf       # reshape(reinterpret(Float64, buf[65:end]), bs, bs, 3N-1): L_1..L_N, C_1..C_{N-1}, Linv_1..Linv_N
end

function import_factor!(F::TridiagonalCholeskyFactor, buf::Vector{UInt8})
    check(ccall((:gmrf_bt_import_factor, libgmrf), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int64), F.handle, buf, length(buf)))
    hdr = reinterpret(Int64, buf[1:64])
    setfield!(F, :N, Int(hdr[3])); setfield!(F, :n_blocks, Int(hdr[4]))
    return F
end

struct GmrfStats                             # gmrf_stats
    factor_ms::Float64; solve_ms::Float64; sample_ms::Float64
    factor_flops::Float64; sweep_bytes::Float64; sweep_ms::Float64
    n::Int64; n_blocks::Int64; block_size::Int64; block_size_padded::Int64; factor_bytes::Int64
    kernel_ms::NTuple{24,Float64}; kernel_work::NTuple{24,Float64}; kernel_launches::NTuple{24,Int64}
    sweep_bytes_streamed::Float64
    persist_route::Int32; persist_aborts::Int32; persist_cus::Int32; persist_refused::Int32
    sweep_persist::Int32; sweep_persist_launches::Int32
end

function stats(F::TridiagonalCholeskyFactor)
    s = Ref{GmrfStats}()
    check(ccall((:gmrf_bt_stats, libgmrf), Int32, (Ptr{Cvoid}, Ref{GmrfStats}), F.handle, s))
    return s[]
end
set_profiling!(F::TridiagonalCholeskyFactor, level::Integer) =
    check(ccall((:gmrf_bt_set_profiling, libgmrf), Int32, (Ptr{Cvoid}, Int32), F.handle, level))
set_eager!(F::TridiagonalCholeskyFactor, flags::Integer) =
    check(ccall((:gmrf_bt_set_eager, libgmrf), Int32, (Ptr{Cvoid}, Int32), F.handle, flags))
synchronize(F::TridiagonalCholeskyFactor) = check(ccall((:gmrf_bt_synchronize, libgmrf), Int32, (Ptr{Cvoid},), F.handle))

# ------------------------------------------------------------------------------------------ K6: Q * x

"Device-resident sparse matrix for `Q * x` (scripts/solve_burger.jl:157-158,166,177; RBMC inside `std`)."
mutable struct GmrfCsr
    handle::Ptr{Cvoid}
    m::Int
    n::Int
end

"""
    GmrfCsr(Q; values_f32 = false)

`Q::SparseMatrixCSC` must be symmetric in pattern and values for the CSC arrays to be passed as CSR (true for
every precision matrix); for a general matrix pass `SparseMatrixCSC(Q')`.  `values_f32`: fp32 values, fp64 accumulation.
"""
function GmrfCsr(Q::SparseMatrixCSC{Float64,Int}; device::Integer = 0, values_f32::Bool = false)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Q check(ccall((:gmrf_csr_create, libgmrf), Int32,
        (Int32, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int32, Int32, Ref{Ptr{Cvoid}}),
        device, C_NULL, size(Q, 2), size(Q, 1), Q.colptr, Q.rowval, Q.nzval, 1, values_f32 ? 1 : 0, h))
    S = GmrfCsr(h[], size(Q, 2), size(Q, 1))
    finalizer(s -> ccall((:gmrf_csr_destroy, libgmrf), Int32, (Ptr{Cvoid},), s.handle), S)
    return S
end

function Base.:*(S::GmrfCsr, X::StridedVecOrMat{Float64})
    size(X, 1) == S.n || throw(DimensionMismatch())
    Y = similar(X, S.m, size(X)[2:end]...)
    GC.@preserve X Y check(ccall((:gmrf_spmm, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64),
                                 S.handle, X, Y, size(X, 2), ndims(X) == 1 ? S.n : stride(X, 2), S.m))
    return Y
end

"`Yt = (S * Xt')'` for node-major operands: `Xt` is k x n (the k values of a node contiguous) -- the layout of the LDS-tiled kernel."
function mul_node_major(S::GmrfCsr, Xt::StridedMatrix{Float64})
    size(Xt, 2) == S.n || throw(DimensionMismatch())
    k = size(Xt, 1)
    Yt = Matrix{Float64}(undef, k, S.m)
    GC.@preserve Xt Yt check(ccall((:gmrf_spmm_rows, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64),
                                   S.handle, Xt, Yt, k, stride(Xt, 2), k))
    return Yt
end

"""
Stream-ordered products for DEVICE arrays (raw device pointers, e.g. `pointer(::ROCArray)` of AMDGPU.jl): enqueued on
the matrix's stream, no synchronisation.  `mul_async!(S, dY, dX, k)`: each right-hand side contiguous (n x k column-major);
`mul_node_major_async!`: k x n (the k values of a node contiguous).
"""
mul_async!(S::GmrfCsr, dY::Ptr{Float64}, dX::Ptr{Float64}, k::Integer = 1) =
    check(ccall((:gmrf_spmm_async, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64), S.handle, dX, dY, k, S.n, S.m))
mul_node_major_async!(S::GmrfCsr, dYt::Ptr{Float64}, dXt::Ptr{Float64}, k::Integer) =
    check(ccall((:gmrf_spmm_rows_async, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64), S.handle, dXt, dYt, k, k, k))

# ------------------------------------------------------------------------------------------ gn_step assembly
# Gauss-Newton assembly on the device (gn_step, scripts/solve_burger.jl:143-149)
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

# ------------------------------------------------------------------------------------------ Darcy stiffness on the device
# assemble_darcy_diff_matrix (src/problems/darcy.jl:5-63) on the structured P1 mesh: per problem only the
# coefficient table crosses the bus; the values come out in the order `PosteriorAssembler` takes as J.
mutable struct DarcyP1Assembler
    handle::Ptr{Cvoid}
    pattern::SparseMatrixCSC{Float64,Int}     # TRANSPOSE of the stiffness pattern (CSC of G' = CSR of G), values 1.0
end

"order = 2: Lagrange{RefTriangle,2} with QuadratureRule{RefTriangle}(3) (src/utils.jl:32-33); dofs = the (2nx-1) x (2ny-1) lattice of vertices and edge midpoints."
function DarcyP1Assembler(nx::Integer, ny::Integer; device::Integer = 0, order::Integer = 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if order == 2
        check(ccall((:gmrf_darcy_p2_create, libgmrf), Int32, (Int32, Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), device, C_NULL, nx, ny, h))
    else
        check(ccall((:gmrf_darcy_p1_create, libgmrf), Int32, (Int32, Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), device, C_NULL, nx, ny, h))
    end
    n = order == 2 ? (2nx - 1) * (2ny - 1) : nx * ny
    nnz_out = Ref{Int64}(0)
    check(ccall((:gmrf_darcy_p1_pattern, libgmrf), Int32, (Ptr{Cvoid}, Ref{Int64}, Ptr{Int64}, Ptr{Int64}, Int32), h[], nnz_out, C_NULL, C_NULL, 1))
    rowptr = Vector{Int64}(undef, n + 1); colidx = Vector{Int64}(undef, nnz_out[])
    check(ccall((:gmrf_darcy_p1_pattern, libgmrf), Int32, (Ptr{Cvoid}, Ref{Int64}, Ptr{Int64}, Ptr{Int64}, Int32), h[], nnz_out, rowptr, colidx, 1))
    d = DarcyP1Assembler(h[], SparseMatrixCSC(n, n, rowptr, colidx, ones(nnz_out[])))
    finalizer(x -> ccall((:gmrf_darcy_p1_destroy, libgmrf), Int32, (Ptr{Cvoid},), x.handle), d)
    return d
end

"`coeff[ix, iy]` on the grid range(0, 1, ng)^2 (as `ds.darcy_vars[\"coeff\"][idx, :, :]`); returns (values of G in CSR order, f)."
function assemble!(vals::Vector{Float64}, f::Vector{Float64}, d::DarcyP1Assembler, coeff::Matrix{Float64}; beta::Real = 1.0)
    ng = size(coeff, 1)
    tab = Matrix{Float64}(coeff')              # the library wants table[x index][y index] row-major = coeff' column-major
    GC.@preserve tab vals f check(ccall((:gmrf_darcy_p1_assemble, libgmrf), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ptr{Float64}), d.handle, tab, ng, Float64(beta), vals, f))
    return vals, f
end

# Burgers residual and tangent (f_and_J, scripts/burgers/solve_burgers_gmrf-fem.jl:118-149) on the periodic P1 line
mutable struct BurgersP1Tangent
    handle::Ptr{Cvoid}
    pattern::SparseMatrixCSC{Float64,Int}     # TRANSPOSE of J's pattern (CSC of J' = CSR of J), values 1.0
    rows::Int
end

function BurgersP1Tangent(ns::Integer, nt::Integer, dt::Real, nu::Real; device::Integer = 0, order::Integer = 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if order == 2       # the quadratic periodic line of periodic_unit_interval_discretization (src/utils.jl:42-49): ns = 2 N_x dofs by position
        check(ccall((:gmrf_burgers_p2_create, libgmrf), Int32, (Int32, Ptr{Cvoid}, Int64, Int64, Float64, Float64, Ref{Ptr{Cvoid}}),
                    device, C_NULL, ns, nt, Float64(dt), Float64(nu), h))
    else
        check(ccall((:gmrf_burgers_p1_create, libgmrf), Int32, (Int32, Ptr{Cvoid}, Int64, Int64, Float64, Float64, Ref{Ptr{Cvoid}}),
                    device, C_NULL, ns, nt, Float64(dt), Float64(nu), h))
    end
    nnz_out = Ref{Int64}(0)
    check(ccall((:gmrf_burgers_p1_pattern, libgmrf), Int32, (Ptr{Cvoid}, Ref{Int64}, Ptr{Int64}, Ptr{Int64}, Int32), h[], nnz_out, C_NULL, C_NULL, 1))
    rows = (nt - 1) * ns
    rowptr = Vector{Int64}(undef, rows + 1); colidx = Vector{Int64}(undef, nnz_out[])
    check(ccall((:gmrf_burgers_p1_pattern, libgmrf), Int32, (Ptr{Cvoid}, Ref{Int64}, Ptr{Int64}, Ptr{Int64}, Int32), h[], nnz_out, rowptr, colidx, 1))
    b = BurgersP1Tangent(h[], SparseMatrixCSC(nt * ns, rows, rowptr, colidx, ones(nnz_out[])), rows)
    finalizer(x -> ccall((:gmrf_burgers_p1_destroy, libgmrf), Int32, (Ptr{Cvoid},), x.handle), b)
    return b
end

"`(J values in CSR order, f) = f_and_J(w)`: the values are what `precision!` / `rhs!` of a `PosteriorAssembler` built on `b.pattern` take as J."
function tangent!(vals::Vector{Float64}, f::Vector{Float64}, b::BurgersP1Tangent, w::Vector{Float64})
    GC.@preserve w vals f check(ccall((:gmrf_burgers_p1_tangent, libgmrf), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), b.handle, w, vals, f))
    return vals, f
end

# Linear shallow-water SPDE (src/spdes/shallow_water.jl): element kernels of assemble_system! (:17-122) and the per-step
# operators of discretize (:170-217) on the structured P1 triangle mesh; dof = 3 * node + field, fields (h, u, v)
mutable struct ShallowWaterP1
    handle::Ptr{Cvoid}
    n::Int                                       # 3 nx ny dofs
    pattern_K::SparseMatrixCSC{Float64,Int}      # TRANSPOSE patterns (CSC of A' = CSR of A), values 1.0: `nzval` order of the value arrays
    pattern_S::SparseMatrixCSC{Float64,Int}
    qpoints::Array{Float64,3}                    # (2, 3, cells): x / y of quadrature point q of a cell -- evaluate H there
end

function ShallowWaterP1(nx::Integer, ny::Integer; device::Integer = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:gmrf_shallow_water_p1_create, libgmrf), Int32, (Int32, Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), device, C_NULL, nx, ny, h))
    n = 3 * nx * ny
    pats = SparseMatrixCSC{Float64,Int}[]
    for which in (0, 1)
        nnz_out = Ref{Int64}(0)
        check(ccall((:gmrf_shallow_water_p1_pattern, libgmrf), Int32, (Ptr{Cvoid}, Int32, Ref{Int64}, Ptr{Int64}, Ptr{Int64}, Int32), h[], which, nnz_out, C_NULL, C_NULL, 1))
        rowptr = Vector{Int64}(undef, n + 1); colidx = Vector{Int64}(undef, nnz_out[])
        check(ccall((:gmrf_shallow_water_p1_pattern, libgmrf), Int32, (Ptr{Cvoid}, Int32, Ref{Int64}, Ptr{Int64}, Ptr{Int64}, Int32), h[], which, nnz_out, rowptr, colidx, 1))
        push!(pats, SparseMatrixCSC(n, n, rowptr, colidx, ones(nnz_out[])))
    end
    cells = 2 * (nx - 1) * (ny - 1)
    qp = Array{Float64,3}(undef, 2, 3, cells)
    check(ccall((:gmrf_shallow_water_p1_qpoints, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Float64}), h[], qp))
    w = ShallowWaterP1(h[], n, pats[1], pats[2], qp)
    finalizer(x -> ccall((:gmrf_shallow_water_p1_destroy, libgmrf), Int32, (Ptr{Cvoid},), x.handle), w)
    return w
end

"`assemble_system!`: H_q[q, cell] = H(qpoints[:, q, cell]); prescribed: the constraint handler's dofs as a byte mask (or nothing).  Returns (K values, lumped M, S values)."
function assemble_system(w::ShallowWaterP1, H_q::Matrix{Float64}; k::Real = 0.0, f::Real = 0.0, g::Real = 9.81, prescribed::Union{Nothing,Vector{UInt8}} = nothing)
    kv = Vector{Float64}(undef, nnz(w.pattern_K)); ml = Vector{Float64}(undef, w.n); sv = Vector{Float64}(undef, nnz(w.pattern_S))
    pm = prescribed === nothing ? Ptr{UInt8}(C_NULL) : pointer(prescribed)
    GC.@preserve H_q prescribed kv ml sv check(ccall((:gmrf_shallow_water_p1_assemble, libgmrf), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Float64, Float64, Float64, Ptr{UInt8}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        w.handle, H_q, Float64(k), Float64(f), Float64(g), pm, kv, ml, sv))
    return kv, ml, sv
end

"The operators `discretize` forms for one time step dt: (G(dt) values in K's pattern, J values in S's pattern with Q_matern = J'J, M~, beta(dt))."
function step_operators(w::ShallowWaterP1, kv::Vector{Float64}, ml::Vector{Float64}, sv::Vector{Float64}; prescribed::Union{Nothing,Vector{UInt8}} = nothing,
                        kappa_matern::Real = 1.0, tau::Real = 1.0, dt::Real = 1.0)
    gv = similar(kv); jv = similar(sv); mt = similar(ml); be = similar(ml)
    pm = prescribed === nothing ? Ptr{UInt8}(C_NULL) : pointer(prescribed)
    GC.@preserve kv ml sv prescribed gv jv mt be check(ccall((:gmrf_shallow_water_p1_operators, libgmrf), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Float64, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        w.handle, kv, ml, sv, pm, Float64(kappa_matern), Float64(tau), Float64(dt), gv, jv, mt, be))
    return gv, jv, mt, be
end

# ------------------------------------------------------------------------------------------ multi-GPU
# One Julia process per GPU (Distributed.jl / MPI.jl launches them; only the 128-byte id has to travel).
# The factor is shared, the samples are sharded (SURVEY 8e): see `shared_factor!` below.

mutable struct GmrfComm
    handle::Ptr{Cvoid}
    rank::Int
    world::Int
end

"Rank 0 calls this and ships the 128 bytes to the other ranks by any means (Distributed.jl, MPI.jl, a file)."
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:gmrf_comm_unique_id, libgmrf), Int32, (Ptr{UInt8},), id))
    return id
end

function GmrfComm(device::Integer, rank::Integer, world::Integer, id::Vector{UInt8})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:gmrf_comm_create, libgmrf), Int32, (Int32, Int32, Int32, Ptr{UInt8}, Ref{Ptr{Cvoid}}), device, rank, world, id, h))
    c = GmrfComm(h[], rank, world)
    finalizer(x -> ccall((:gmrf_comm_destroy, libgmrf), Int32, (Ptr{Cvoid},), x.handle), c)
    return c
end

"In-place broadcast of a small host array (layout records, scalars) from `root`."
bcast_host!(c::GmrfComm, a::Array, root::Integer = 0) =
    (check(ccall((:gmrf_comm_bcast_host, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32), c.handle, a, sizeof(a), root)); a)

"Sum of the ranks' DEVICE buffers (variance accumulators), in place; `dev_ptr` from the caller's GPU array package."
allreduce_sum!(c::GmrfComm, F::TridiagonalCholeskyFactor, dev_ptr::Ptr{Float64}, count::Integer) =
    check(ccall((:gmrf_comm_allreduce_sum, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64), c.handle, F.handle, dev_ptr, count))

"Layout record of the stored factor: [cmin, rmax, n_row_tiles, kst..., split p of the block inverses (0: full)]."
function get_layout(F::TridiagonalCholeskyFactor)
    cnt = Ref{Int64}(0)
    check(ccall((:gmrf_bt_get_layout, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64, Ref{Int64}), F.handle, C_NULL, 0, cnt))
    out = Vector{Int64}(undef, cnt[])
    check(ccall((:gmrf_bt_get_layout, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64, Ref{Int64}), F.handle, out, length(out), cnt))
    return out
end

function adopt_layout!(F::TridiagonalCholeskyFactor, n::Integer, n_blocks::Integer, layout::Vector{Int64})
    check(ccall((:gmrf_bt_adopt_layout, libgmrf), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Int64), F.handle, n, n_blocks, layout, length(layout)))
    setfield!(F, :N, Int(n)); setfield!(F, :n_blocks, Int(n_blocks))
    return F
end
adopt_shape!(F::TridiagonalCholeskyFactor, n::Integer, n_blocks::Integer) =
    (check(ccall((:gmrf_bt_adopt_shape, libgmrf), Int32, (Ptr{Cvoid}, Int64, Int64), F.handle, n, n_blocks)); setfield!(F, :N, Int(n)); setfield!(F, :n_blocks, Int(n_blocks)); F)
adopt_commit!(F::TridiagonalCholeskyFactor; l_blocks_valid::Bool = false) =
    check(ccall((:gmrf_bt_adopt_commit, libgmrf), Int32, (Ptr{Cvoid}, Int32), F.handle, l_blocks_valid ? 1 : 0))

# pipelined factorisation (block ranges) of the root
function factor_begin!(F::TridiagonalCholeskyFactor, A::SparseMatrixCSC{Float64,Int}, N_blocks::Integer)
    GC.@preserve A check(ccall((:gmrf_bt_factor_begin_csc, libgmrf), Int32,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int32), F.handle, size(A, 1), N_blocks, A.colptr, A.rowval, A.nzval, 1))
    setfield!(F, :N, size(A, 1)); setfield!(F, :n_blocks, Int(N_blocks))
    return F
end
factor_step_async!(F::TridiagonalCholeskyFactor, i0::Integer, i1::Integer) =      # blocks i0 .. i1-1, 0-based
    check(ccall((:gmrf_bt_factor_step_async, libgmrf), Int32, (Ptr{Cvoid}, Int64, Int64), F.handle, i0, i1))
function factor_end!(F::TridiagonalCholeskyFactor)
    info = Ref{Int32}(0)
    check(ccall((:gmrf_bt_factor_end, libgmrf), Int32, (Ptr{Cvoid}, Ref{Int32}), F.handle, info), info[])
    return F
end
# the all-gather form (every rank factors its share `Fown`; `Fall` has batch world * batch(Fown) and adopted Fown's layout)
allgather_blocks_async!(Fown::TridiagonalCholeskyFactor, Fall::TridiagonalCholeskyFactor, c::GmrfComm, i0::Integer, i1::Integer) =
    check(ccall((:gmrf_bt_allgather_blocks_async, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64), Fown.handle, Fall.handle, c.handle, i0, i1))
bcast_blocks_async!(F::TridiagonalCholeskyFactor, c::GmrfComm, i0::Integer, i1::Integer; root::Integer = 0, with_l::Bool = false) =
    check(ccall((:gmrf_bt_bcast_blocks_async, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Int64, Int32), F.handle, c.handle, root, i0, i1, with_l ? 1 : 0))
comm_wait!(F::TridiagonalCholeskyFactor, c::GmrfComm) =
    check(ccall((:gmrf_comm_wait, libgmrf), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), F.handle, c.handle))
"Factor bytes this communicator has broadcast so far."
function comm_bytes(c::GmrfComm; reset::Bool = false)
    b = Ref{Float64}(0.0)
    check(ccall((:gmrf_comm_bytes, libgmrf), Int32, (Ptr{Cvoid}, Int32, Ref{Float64}), c.handle, reset ? 1 : 0, b))
    return b[]
end

# Packed transport image of a block range (what gmrf_bt_bcast_blocks_async moves; for a transport of the caller's own,
# e.g. MPI.jl on device pointers): doubles per problem, pack from / unpack into the handle's factor storage.
function packed_size(F::TridiagonalCholeskyFactor, i0::Integer, i1::Integer)
    v = Ref{Int64}(0)
    check(ccall((:gmrf_bt_packed_size, libgmrf), Int32, (Ptr{Cvoid}, Int64, Int64, Ref{Int64}), F.handle, i0, i1, v))
    return v[]
end
pack_blocks_async!(F::TridiagonalCholeskyFactor, i0::Integer, i1::Integer, dev_buf::Ptr{Float64}) =
    check(ccall((:gmrf_bt_pack_blocks_async, libgmrf), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}), F.handle, i0, i1, dev_buf))
unpack_blocks_async!(F::TridiagonalCholeskyFactor, i0::Integer, i1::Integer, dev_buf::Ptr{Float64}) =
    check(ccall((:gmrf_bt_unpack_blocks_async, libgmrf), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}), F.handle, i0, i1, dev_buf))

"""
    create_streams(n; device = 0) -> (streams::Vector{Ptr{Cvoid}}, n_distinct)

`n` HIP streams on hardware queues of their own (probed by the library: streams that share a queue serialise), one per
handle that a task drives side by side with others: `TridiagonalCholeskyFactor{Float64}(device; stream = streams[t])`.
Release with `destroy_streams!`.
"""
function create_streams(n::Integer; device::Integer = 0)
    streams = Vector{Ptr{Cvoid}}(undef, n)
    nd = Ref{Int32}(0)
    check(ccall((:gmrf_streams_create, libgmrf), Int32, (Int32, Int32, Ptr{Ptr{Cvoid}}, Ptr{Int32}), device, n, streams, nd))
    return streams, Int(nd[])
end
destroy_streams!(streams::Vector{Ptr{Cvoid}}; device::Integer = 0) =
    check(ccall((:gmrf_streams_destroy, libgmrf), Int32, (Int32, Int32, Ptr{Ptr{Cvoid}}), device, length(streams), streams))

"""
    shared_factor!(F, c, A, N_blocks; group = 8)

The north-star split on every rank of a node: rank 0 factors `A` block range by block range, each finished
range of Linv / C blocks is broadcast over RCCL while the next one is being factored; the other ranks
receive.  Afterwards every rank solves / samples its own right-hand sides and sample ids
(`sample(F, k; first_id = c.rank * k)`).  `A` is only read on rank 0 (pass the same pattern everywhere).
"""
function shared_factor!(F::TridiagonalCholeskyFactor, c::GmrfComm, A::SparseMatrixCSC{Float64,Int}, N_blocks::Integer; group::Integer = 8)
    n = size(A, 1)
    if c.rank == 0
        factor_begin!(F, A, N_blocks)
        lay = get_layout(F)
        cnt = Int64[length(lay)]
    else
        lay = Int64[]; cnt = Int64[0]
    end
    bcast_host!(c, cnt)
    c.rank == 0 || (lay = Vector{Int64}(undef, cnt[1]))
    bcast_host!(c, lay)
    c.rank == 0 || adopt_layout!(F, n, N_blocks, lay)
    for i0 in 0:group:N_blocks-1
        i1 = min(i0 + group, N_blocks)
        c.rank == 0 && factor_step_async!(F, i0, i1)
        bcast_blocks_async!(F, c, i0, i1)
    end
    comm_wait!(F, c)
    c.rank == 0 ? factor_end!(F) : adopt_commit!(F)
    return F
end

# --- caller-owned factor storage (e.g. arrays of a GPU package that another library broadcasts)
function storage_bytes(n::Integer, n_blocks::Integer, batch::Integer = 1)
    bl = Ref{Int64}(0); bc = Ref{Int64}(0); bi = Ref{Int64}(0)
    check(ccall((:gmrf_bt_storage_bytes, libgmrf), Int32, (Int64, Int64, Int64, Ref{Int64}, Ref{Int64}, Ref{Int64}), n, n_blocks, batch, bl, bc, bi))
    return (L = bl[], C = bc[], Linv = bi[])
end
set_storage!(F::TridiagonalCholeskyFactor, n::Integer, n_blocks::Integer, dev_L::Ptr{Cvoid}, dev_C::Ptr{Cvoid}, dev_Linv::Ptr{Cvoid}) =
    (check(ccall((:gmrf_bt_set_storage, libgmrf), Int32, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                 F.handle, n, n_blocks, F.batch, dev_L, dev_C, dev_Linv)); setfield!(F, :N, Int(n)); setfield!(F, :n_blocks, Int(n_blocks)); F)
function factor_buffer(F::TridiagonalCholeskyFactor, kind::Int32)
    p = Ref{Ptr{Cvoid}}(C_NULL); nb = Ref{Int64}(0)
    check(ccall((:gmrf_bt_factor_buffer, libgmrf), Int32, (Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}, Ref{Int64}), F.handle, kind, p, nb))
    return p[], nb[]
end
function block_range(F::TridiagonalCholeskyFactor, kind::Int32, i0::Integer, i1::Integer)
    a = Ref{Int64}(0); b = Ref{Int64}(0); s = Ref{Int64}(0)
    check(ccall((:gmrf_bt_block_range, libgmrf), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ref{Int64}, Ref{Int64}, Ref{Int64}), F.handle, kind, i0, i1, a, b, s))
    return (first = a[], count = b[], problem_stride = s[])
end

end # module
