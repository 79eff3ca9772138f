# GeostatInversionHIP.jl -- Julia-side drop-in for the RandMatFact hot path of GeostatInversion.jl,
# calling libgsi_hip.so (hand-written gfx950 HIP kernels) through `ccall`.  It keeps the reference
# signatures (file:line of the reference in comments) so existing PCGA / RGA call sites work
# unchanged:
#
#     import GeostatInversionHIP as GH
#     xis = GH.getxis(Q, numxis, p, q, seed)                  # GeostatInversion.jl:63
#     xis = GH.getxis(samplefield, numfields, numxis, p, q, seed)   # GeostatInversion.jl:58
#     popt = GeostatInversion.pcgadirect(forward, s0, X, xis, R, y) # xis is Vector{Vector{Float64}}
#
# The Gaussian test matrix is drawn HERE with Julia's own `randn`, exactly where the reference
# draws it (RandMatFact.jl:54 after Random.seed! at GeostatInversion.jl:25), so for a given seed the
# GPU path sees the same Omega as the reference.
#
# NOTE: there is no Julia in the build image; this file is the binding a maintainer would add and is
# exercised only where Julia is available.  The same ABI is exercised by the Python ctypes mirror
# (geostatinversion.jl_amd/) in the test-suite.
module GeostatInversionHIP

import Random
import Libdl
import LinearAlgebra
import Distributed

const libgsi = get(ENV, "GSI_HIP_LIB", joinpath(@__DIR__, "..", "geostatinversion.jl_amd", "libgsi_hip.so"))

struct GsiError <: Exception
	code::Int
	msg::String
end
Base.showerror(io::IO, e::GsiError) = print(io, "GsiError($(e.code)): $(e.msg)")

function check(status::Cint)
	if status != 0
		msg = unsafe_string(ccall((:gsi_last_error, libgsi), Cstring, ()))
		# the reference raises ErrorException via error(...) (RandMatFact.jl:63, lowrank.jl:58)
		status == 2 && error(msg)
		# lu(Y) on an exactly zero pivot (RandMatFact.jl:60) and cholesky(Hermitian(B2)) of eig_nystrom (:95) throw these in the reference
		status == 3 && throw(LinearAlgebra.SingularException(0))
		status == 7 && throw(LinearAlgebra.PosDefException(0))
		throw(GsiError(Int(status), msg))
	end
	return nothing
end

# ---- context: one per GPU -------------------------------------------------------------------------
mutable struct Context
	h::Ptr{Cvoid}
	function Context(device::Integer=0)
		r = Ref{Ptr{Cvoid}}(C_NULL)
		check(ccall((:gsi_ctx_create, libgsi), Cint, (Ref{Ptr{Cvoid}}, Cint), r, device))
		c = new(r[])
		finalizer(c) do x
			x.h != C_NULL && ccall((:gsi_ctx_destroy, libgsi), Cint, (Ptr{Cvoid},), x.h)
			x.h = C_NULL
		end
		return c
	end
end
"Return the context's cached device memory (released panels, idle workspaces) to the driver."
release_cache!(c::Context) = check(ccall((:gsi_ctx_release_cache, libgsi), Cint, (Ptr{Cvoid},), c.h))
const default_ctx = Ref{Union{Nothing, Context}}(nothing)
"The process-wide default context (GPU 0), created at its first use.  (`something(a, b)` would evaluate `b` -- a new
context -- at every call: the arguments of a function are evaluated before it runs.)"
function ctx()
	default_ctx[] === nothing && (default_ctx[] = Context(0))
	return default_ctx[]::Context
end

"Join `nranks` contexts (one Julia process per GPU) into an RCCL communicator; rank 0 creates the id."
function unique_id()
	id = zeros(UInt8, 128)
	check(ccall((:gsi_comm_unique_id, libgsi), Cint, (Ptr{UInt8},), id))
	return id
end
comm_init!(c::Context, nranks::Integer, rank::Integer, id::Vector{UInt8}) =
	check(ccall((:gsi_ctx_comm_init, libgsi), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), c.h, nranks, rank, id))

"Every rank's `values` on every rank (`gsi_ctx_host_allgather`): a `nranks x length(values)` matrix, row r = rank r - 1.  Also
the job's barrier (it waits for the context's stream first): what a Distributed.jl host needs around the hot path."
function host_allgather(c::Context, values::Vector{Float64}, nranks::Integer)
	out = Matrix{Float64}(undef, length(values), nranks)
	check(ccall((:gsi_ctx_host_allgather, libgsi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}), c.h, values, length(values), out))
	return Matrix(out')
end

const LU_FORMS = ["none", "replicated", "per-step", "persistent-1hop", "persistent-2hop", "persistent-ov"]
"Which path ran under the communicator (`gsi_ctx_path_info`): LU form, self-test mask of the in-kernel pivot exchange,
collectives since the last phase reset, ranks joined, LU time-outs seen / hidden by a transparent re-run."
function path_info(c::Context)
	out = zeros(Int64, 13)
	check(ccall((:gsi_ctx_path_info, libgsi), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64), c.h, out, length(out)))
	return (lu_form = LU_FORMS[out[1] + 1], lu_selftest_mask = out[2], collectives = out[3], n_ranks_seen = out[4],
		lu_timeouts = out[5], lu_timeouts_recovered = out[6], svd_sweep_cap_hits = out[13], lu_forms_run = Dict(LU_FORMS[f + 1] => out[7 + f] for f = 1:5 if out[7 + f] > 0))
end

# ---- operators ------------------------------------------------------------------------------------
mutable struct DeviceOperator
	h::Ptr{Cvoid}
	c::Context
	m::Int
	n::Int
end
function finalize_op!(op::DeviceOperator)
	op.h != C_NULL && ccall((:gsi_op_destroy, libgsi), Cint, (Ptr{Cvoid},), op.h)
	op.h = C_NULL
end

"Upload a dense `Matrix{Float64}` (this rank's block of rows when a communicator is attached)."
function DeviceOperator(A::Matrix{Float64}; c::Context=ctx(), row0::Int=0, mlocal::Int=size(A, 1))
	r = Ref{Ptr{Cvoid}}(C_NULL)
	m, n = size(A)
	GC.@preserve A check(ccall((:gsi_op_dense, libgsi), Cint,
		(Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Float64}, Int64, Int64, Int64, Int64, Int64),
		c.h, r, pointer(A, row0 + 1), m, n, stride(A, 2), row0, mlocal))
	op = DeviceOperator(r[], c, m, n)
	finalizer(finalize_op!, op)
	return op
end

"`LowRankCovMatrix(samples)` (lowrank.jl:14-30): samples uploaded as an n x N matrix, centred on the device."
function LowRankCovMatrix(samples::Array{Array{Float64, 1}, 1}; c::Context=ctx())
	n, N = length(samples[1]), length(samples)
	S = Matrix{Float64}(undef, n, N)
	for i = 1:N
		S[:, i] = samples[i]
	end
	r = Ref{Ptr{Cvoid}}(C_NULL)
	check(ccall((:gsi_op_lowrank, libgsi), Cint,
		(Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Float64}, Int64, Int64, Int64, Cint, Int64, Int64),
		c.h, r, S, n, N, n, 1, 0, n))
	op = DeviceOperator(r[], c, n, n)
	finalizer(finalize_op!, op)
	return op
end

"`A.samples` of a device LowRankCovMatrix (lowrank.jl:14-16; mean-removed, lowrank.jl:25-27) back on the host as the
reference keeps them: a Vector of N sample vectors (`gsi_op_lowrank_samples`)."
function samples(A::DeviceOperator, N::Int)
	S = Matrix{Float64}(undef, A.m, N)
	check(ccall((:gsi_op_lowrank_samples, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64),
		A.c.h, A.h, S, A.m))
	return [S[:, i] for i = 1:N]
end

"Gaussian covariance exp(-d^2/(2 ell^2)) of an nx x ny unit grid as a matrix-free operator: entries are regenerated
inside the product kernel, nothing of size n^2 is stored (`gsi_op_gridcov_implicit`).  Usable wherever a Matrix is."
function GridCovImplicit(nx::Int, ny::Int, ell::Float64; c::Context=ctx())
	r = Ref{Ptr{Cvoid}}(C_NULL)
	check(ccall((:gsi_op_gridcov_implicit, libgsi), Cint,
		(Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Int64, Int64, Cdouble, Int64, Int64),
		c.h, r, nx, ny, ell, 0, nx * ny))
	op = DeviceOperator(r[], c, nx * ny, nx * ny)
	finalizer(finalize_op!, op)
	return op
end

"The same operator for another stationary kernel on the nx x ny unit grid: `kind = :gaussian` (exp(-d^2/(2 ell^2))),
`kind = :exponential` (exp(-d/ell)), or any covariance of the grid offsets handed over as a table
`table[dy + 1, dx + 1] = k(dx, dy)` (ny x nx: Matern, anisotropic, nested ...).  `gsi_op_gridcov_implicit_kind`,
`gsi_op_gridcov_implicit_table`; entries are looked up in that 8 n-byte table inside the product kernel."
function GridCovImplicit(nx::Int, ny::Int, ell::Float64, kind::Symbol; c::Context=ctx())
	k = kind == :gaussian ? 0 : kind == :exponential ? 1 : error("GridCovImplicit: kind must be :gaussian or :exponential")
	r = Ref{Ptr{Cvoid}}(C_NULL)
	check(ccall((:gsi_op_gridcov_implicit_kind, libgsi), Cint,
		(Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Int64, Int64, Cdouble, Cint, Int64, Int64),
		c.h, r, nx, ny, ell, k, 0, nx * ny))
	op = DeviceOperator(r[], c, nx * ny, nx * ny)
	finalizer(finalize_op!, op)
	return op
end
function GridCovImplicit(table::Matrix{Float64}; c::Context=ctx())
	ny, nx = size(table)                       # column-major ny x nx  ==  t[dx * ny + dy] of the C ABI
	r = Ref{Ptr{Cvoid}}(C_NULL)
	check(ccall((:gsi_op_gridcov_implicit_table, libgsi), Cint,
		(Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Int64, Int64, Ptr{Float64}, Int64, Int64),
		c.h, r, nx, ny, table, 0, nx * ny))
	op = DeviceOperator(r[], c, nx * ny, nx * ny)
	finalizer(finalize_op!, op)
	return op
end

"The covariance of scattered points as an implicit operator: `A[i, j] = sigma2 * k(|x_i - x_j| / ell)` (+ `nugget` on the
diagonal), `points` d x n as `FFTRF.powerlaw_unstructuredgrid` takes them (FFTRF.jl:102), `kind` in `:gaussian`,
`:exponential`, `:matern32`, `:matern52` (`gsi_op_pointcov_implicit`).  Nothing n x n is stored: row panels of A are
generated on a second stream while the MFMA contraction consumes the previous one.  Usable wherever a Matrix is."
function PointCovImplicit(points::Matrix{Float64}, kind::Symbol=:exponential; ell::Float64=1.0, sigma2::Float64=1.0,
		nugget::Float64=0.0, c::Context=ctx())
	k = Dict(:gaussian=>0, :exponential=>1, :matern32=>2, :matern52=>3)[kind]
	d, n = size(points)
	r = Ref{Ptr{Cvoid}}(C_NULL)
	check(ccall((:gsi_op_pointcov_implicit, libgsi), Cint,
		(Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Float64}, Int64, Cint, Cint, Cdouble, Cdouble, Cdouble, Int64, Int64),
		c.h, r, points, n, d, k, ell, sigma2, nugget, 0, n))
	op = DeviceOperator(r[], c, n, n)
	finalizer(finalize_op!, op)
	return op
end

"The covariance of `FFTRF.powerlaw_structuredgrid(Ns, k0, dk, beta)` fields themselves (up to dk^2 and the per-sample
mean / std normalisation, FFTRF.jl:94-98): FFTRF's own embedding of exactly 2N points per axis and its integer
wavenumbers (FFTRF.jl:83-90, computesqrtS_f :40-72), `gsi_op_fft_powerlaw_fftrf`.  Acts on `vec(field)`; every grid
dimension must be a power of two (the exact `getxis(samplefield, ...)` sample estimate becomes this operator as the number
of fields grows: tests/test_fftrf_covariance.py)."
function FFTRFCovariance(Ns::Vector{Int}, beta::Float64; c::Context=ctx())
	r = Ref{Ptr{Cvoid}}(C_NULL)
	N64 = Int64.(Ns)
	check(ccall((:gsi_op_fft_powerlaw_fftrf, libgsi), Cint,
		(Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Cint, Ptr{Int64}, Cdouble),
		c.h, r, length(N64), N64, beta))
	n = prod(Ns)
	op = DeviceOperator(r[], c, n, n)
	finalizer(finalize_op!, op)
	return op
end

"Matrix-free stationary power-law covariance on a structured grid (circulant embedding on the next power of two
>= 2N per axis, spectrum |k|^beta with k in cycles per grid spacing, unit diagonal): `gsi_op_fft_powerlaw`.  Acts on
`vec(field)`.  It is the covariance family FFTRF.powerlaw_structuredgrid samples from; it coincides with the
covariance of those fields (up to the per-sample normalisation) only on power-of-two grids with equal axes --
FFTRF embeds on exactly 2N points and uses integer wavenumbers (FFTRF.jl:83-90), see DESIGN.md section 4.6."
function FFTPowerlawCovariance(Ns::Vector{Int}, beta::Float64; c::Context=ctx())
	r = Ref{Ptr{Cvoid}}(C_NULL)
	N64 = Int64.(Ns)
	check(ccall((:gsi_op_fft_powerlaw, libgsi), Cint,
		(Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Cint, Ptr{Int64}, Cdouble),
		c.h, r, length(N64), N64, beta))
	n = prod(Ns)
	op = DeviceOperator(r[], c, n, n)
	finalizer(finalize_op!, op)
	return op
end

Base.size(A::DeviceOperator) = (A.m, A.n)
function Base.size(A::DeviceOperator, i::Int)
	(i == 1 || i == 2) || error("there is no $i-th dimension in a DeviceOperator")   # lowrank.jl:58
	return i == 1 ? A.m : A.n
end
Base.eltype(::DeviceOperator) = Float64
function Base.:*(A::DeviceOperator, X::Matrix{Float64})                                   # lowrank.jl:115-121
	Y = Matrix{Float64}(undef, A.m, size(X, 2))
	check(ccall((:gsi_op_mul, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
		A.c.h, A.h, 0, X, stride(X, 2), size(X, 2), Y, A.m))
	return Y
end

"`adjoint(A) * X`  (RandMatFact.jl:67,85)"
struct AdjointOperator
	parent::DeviceOperator
end
Base.adjoint(A::DeviceOperator) = AdjointOperator(A)
Base.transpose(A::DeviceOperator) = AdjointOperator(A)
Base.size(A::AdjointOperator) = (A.parent.n, A.parent.m)
function Base.size(A::AdjointOperator, i::Int)
	(i == 1 || i == 2) || error("there is no $i-th dimension in a DeviceOperator")   # lowrank.jl:58
	return i == 1 ? A.parent.n : A.parent.m
end
Base.eltype(::AdjointOperator) = Float64
Base.adjoint(At::AdjointOperator) = At.parent
Base.transpose(At::AdjointOperator) = At.parent
Base.:*(At::AdjointOperator, x::Vector{Float64}) = vec(At * reshape(x, :, 1))
function Base.:*(At::AdjointOperator, X::Matrix{Float64})
	A = At.parent
	Y = Matrix{Float64}(undef, A.n, size(X, 2))
	check(ccall((:gsi_op_mul, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
		A.c.h, A.h, 1, X, stride(X, 2), size(X, 2), Y, A.n))
	return Y
end
Base.:*(A::DeviceOperator, x::Vector{Float64}) = vec(A * reshape(x, :, 1))            # lowrank.jl:135-139

"`mul!(v, A, x)`  (lowrank.jl:75-81): v = A x into the caller's vector -- what IterativeSolvers.lsqr calls on the operator."
function LinearAlgebra.mul!(v::Vector{Float64}, A::DeviceOperator, x::Vector{Float64})
	length(v) == A.m && length(x) == A.n || throw(DimensionMismatch("mul!: A is $(A.m)x$(A.n), x has $(length(x)), v has $(length(v))"))
	check(ccall((:gsi_op_mul, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
		A.c.h, A.h, 0, x, A.n, 1, v, A.m))
	return v
end
function LinearAlgebra.mul!(v::Vector{Float64}, At::AdjointOperator, x::Vector{Float64})
	A = At.parent
	length(v) == A.n && length(x) == A.m || throw(DimensionMismatch("mul!: A' is $(A.n)x$(A.m), x has $(length(x)), v has $(length(v))"))
	check(ccall((:gsi_op_mul, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
		A.c.h, A.h, 1, x, A.m, 1, v, A.n))
	return v
end

"`B * A` for a plain matrix B  (lowrank.jl:123-129; `I * lrcm` at test/testrpcga.jl:49,90): B A = (A' B')', one device
product on the transposed panel."
function Base.:*(B::Matrix{Float64}, A::DeviceOperator)
	size(B, 2) == A.m || throw(DimensionMismatch("B has $(size(B, 2)) columns, A has $(A.m) rows"))
	return Matrix((adjoint(A) * Matrix(B'))')
end
"`B' * A` for an adjoint matrix  (lowrank.jl:131-133, RandMatFact.jl:85 `Q' * A`): (A' B)' -- for the symmetric
LowRankCovMatrix this is the reference's `(A * B.parent)'`."
function Base.:*(B::LinearAlgebra.Adjoint{Float64, Matrix{Float64}}, A::DeviceOperator)
	size(B, 2) == A.m || throw(DimensionMismatch("B' has $(size(B, 2)) columns, A has $(A.m) rows"))
	return (adjoint(A) * B.parent)'
end

"`\\(A::LowRankCovMatrix, b::Vector)`  (lowrank.jl:141-144): lsqr(A, b; maxiter=length(A.samples)) on the device."
function Base.:\(A::DeviceOperator, b::Vector{Float64})
	x = Vector{Float64}(undef, A.n)
	check(ccall((:gsi_op_lowrank_solve, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}),
		A.c.h, A.h, b, x, C_NULL))
	return x
end

# ---- device-resident matrices and the xi-basis (SURVEY.md 8b last row / 8f f1) -----------------------------
mutable struct DeviceMatrix
	h::Ptr{Cvoid}
	c::Context
	rows::Int
	cols::Int
	function DeviceMatrix(rows::Int, cols::Int; c::Context=ctx())
		r = Ref{Ptr{Cvoid}}(C_NULL)
		check(ccall((:gsi_mat_create, libgsi), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Int64, Int64), c.h, r, rows, cols))
		m = new(r[], c, rows, cols)
		finalizer(m) do x
			x.h != C_NULL && ccall((:gsi_mat_destroy, libgsi), Cint, (Ptr{Cvoid},), x.h)
			x.h = C_NULL
		end
		return m
	end
end
function DeviceMatrix(A::Matrix{Float64}; c::Context=ctx())
	m = DeviceMatrix(size(A, 1), size(A, 2); c=c)
	check(ccall((:gsi_mat_upload, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64), c.h, m.h, A, stride(A, 2)))
	return m
end

"The xi-basis resident in HBM: what `getxis_device` returns and what `pcgadirect` / `pcgalsqr` / `rga` accept in place
of `xis::Array{Array{Float64,1},1}`.  `precision = 32` stores the K columns in fp32 (all sums in fp64)."
mutable struct DeviceBasis
	h::Ptr{Cvoid}
	Z::DeviceMatrix            # kept alive: a 64-bit basis points into it
	n::Int
	K::Int
	function DeviceBasis(Z::DeviceMatrix, K::Int; precision::Int=64)
		r = Ref{Ptr{Cvoid}}(C_NULL)
		check(ccall((:gsi_basis_create, libgsi), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Cint),
			Z.c.h, r, Z.h, K, precision))
		b = new(r[], Z, Z.rows, K)
		finalizer(b) do x
			x.h != C_NULL && ccall((:gsi_basis_destroy, libgsi), Cint, (Ptr{Cvoid},), x.h)
			x.h = C_NULL
		end
		return b
	end
end
Base.length(b::DeviceBasis) = b.K
function Base.getindex(b::DeviceBasis, i::Int)                                           # xis[i] on the host
	x = Vector{Float64}(undef, b.n)
	check(ccall((:gsi_basis_download_col, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}), b.Z.c.h, b.h, i - 1, x))
	return x
end
"paramstorun as an n x (K+3) matrix: columns s + delta*xis[i], s + delta*X, s + delta*s, s   (direct.jl:39-45)"
function paramstorun(b::DeviceBasis, s::Vector{Float64}, X::Vector{Float64}, delta::Float64)
	P = Matrix{Float64}(undef, b.n, b.K + 3)
	check(ccall((:gsi_pcga_params_basis, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}),
		b.Z.c.h, b.h, s, X, delta, P))
	return P
end
"s = X*beta_bar + sum_i xis[i]*dot(etas[i], xi_bar)   (direct.jl:59-65, lsqr.jl:55-61)"
function update(b::DeviceBasis, X::Vector{Float64}, beta_bar::Float64, etas::Matrix{Float64}, xi_bar::Vector{Float64})
	s = Vector{Float64}(undef, b.n)
	check(ccall((:gsi_pcga_update_basis, libgsi), Cint,
		(Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cdouble, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}),
		b.Z.c.h, b.h, X, beta_bar, etas, size(etas, 1), xi_bar, s))
	return s
end

"`PCGALowRankMatrix(etas, HX, R)` (lowrank.jl:32-36) resident on the device; `A * x` and `lsqr(A, b)` run there."
mutable struct PCGALowRankMatrix
	h::Ptr{Cvoid}
	c::Context
	nobs::Int
	function PCGALowRankMatrix(etas::Matrix{Float64}, HX::Vector{Float64}, R::AbstractMatrix; c::Context=ctx())
		nobs, K = size(etas)
		isdiag = R == LinearAlgebra.Diagonal(LinearAlgebra.diag(R))
		Rv = isdiag ? collect(Float64, LinearAlgebra.diag(R)) : vec(Matrix{Float64}(R))
		r = Ref{Ptr{Cvoid}}(C_NULL)
		check(ccall((:gsi_pcgamat_create, libgsi), Cint,
			(Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Cint),
			c.h, r, etas, nobs, K, HX, Rv, isdiag ? 1 : 0))
		A = new(r[], c, nobs)
		finalizer(A) do x
			x.h != C_NULL && ccall((:gsi_pcgamat_destroy, libgsi), Cint, (Ptr{Cvoid},), x.h)
			x.h = C_NULL
		end
		return A
	end
end
Base.size(A::PCGALowRankMatrix) = (A.nobs + 1, A.nobs + 1)
function Base.size(A::PCGALowRankMatrix, i::Int)
	(i == 1 || i == 2) || error("there is no $i-th dimension in a PCGALowRankMatrix")     # lowrank.jl:71
	return A.nobs + 1
end
function Base.:*(A::PCGALowRankMatrix, x::Vector{Float64})                                # lowrank.jl:83-97, 109-113
	y = Vector{Float64}(undef, A.nobs + 1)
	check(ccall((:gsi_pcgamat_mul, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), A.c.h, A.h, x, y))
	return y
end
"`IterativeSolvers.lsqr(A, b)` with that package's defaults, on the device   (lsqr.jl:54)"
function lsqr(A::PCGALowRankMatrix, b::Vector{Float64})
	x = Vector{Float64}(undef, A.nobs + 1)
	check(ccall((:gsi_pcgamat_lsqr, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}),
		A.c.h, A.h, b, x, C_NULL))
	return x
end

# ---- RandMatFact ------------------------------------------------------------------------------------
module RandMatFact
import Random
import ..DeviceOperator, ..libgsi, ..check, ..ctx

"`rangefinder(A, l::Int64, numiterations::Int64)`  (RandMatFact.jl:50-80)"
function rangefinder(A::DeviceOperator, l::Int64, numiterations::Int64)
	Omega = randn(A.n, l)                                                            # RandMatFact.jl:54
	Q = Matrix{Float64}(undef, A.m, l)
	check(ccall((:gsi_rangefinder, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}),
		A.c.h, A.h, Omega, l, numiterations, Q))
	return Q
end
"The same for a host `Matrix{Float64}`: uploaded in row blocks, the sketch `A * Omega` (RandMatFact.jl:55) runs under the upload."
function rangefinder(A::Matrix{Float64}, l::Int64, numiterations::Int64; c=ctx())
	m, n = size(A)
	Omega = randn(n, l)                                                              # RandMatFact.jl:54
	Q = Matrix{Float64}(undef, m, l)
	check(ccall((:gsi_rangefinder_dense_host, libgsi), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Cvoid}),
		c.h, A, m, n, stride(A, 2), Omega, l, numiterations, Q, C_NULL))
	return Q
end

"`rangefinder(A; epsilon=1e-8, r=10)`  (RandMatFact.jl:15-48): the library pulls Julia's randn stream through a callback."
function rangefinder(A::Matrix{Float64}; epsilon=1e-8, r=10)
	op = DeviceOperator(A)
	fill!(_user::Ptr{Cvoid}, buf::Ptr{Float64}, count::Int64) = (Random.randn!(unsafe_wrap(Array, buf, count)); nothing)
	cb = @cfunction($fill!, Cvoid, (Ptr{Cvoid}, Ptr{Float64}, Int64))
	Q = Matrix{Float64}(undef, op.m, min(op.m, op.n))
	ncols = Ref{Int64}(0)
	GC.@preserve cb check(ccall((:gsi_rangefinder_adaptive, libgsi), Cint,
		(Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Ptr{Float64}, Ref{Int64}),
		op.c.h, op.h, cb, C_NULL, epsilon, r, Q, ncols))
	return Q[:, 1:ncols[]]
end

"`randsvd(A, K::Int, p::Int, q::Int)`  (RandMatFact.jl:83-90)"
function randsvd(A::DeviceOperator, K::Int, p::Int, q::Int)
	Omega = randn(A.n, K + p)                                                        # drawn inside rangefinder in the reference
	Z = Matrix{Float64}(undef, A.n, K + p)
	check(ccall((:gsi_randsvd, libgsi), Cint,
		(Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}),
		A.c.h, A.h, Omega, K, p, q, Z, C_NULL))
	return Z
end
"`randsvd(Q::Matrix, numxis, p, q)` as `getxis(Q::Matrix, ...)` calls it (GeostatInversion.jl:63-70 -> :20-27): the matrix is
still on the host; it is uploaded in row blocks through the library's pinned staging ring and the first pass runs under the upload."
function randsvd(A::Matrix{Float64}, K::Int, p::Int, q::Int; c=ctx())
	m, n = size(A)
	Omega = randn(n, K + p)
	Z = Matrix{Float64}(undef, n, K + p)
	check(ccall((:gsi_randsvd_dense_host, libgsi), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
		c.h, A, m, n, stride(A, 2), Omega, K, p, q, Z, C_NULL, C_NULL))
	return Z
end

"`eig_nystrom(A, Q)`  (RandMatFact.jl:92-102)"
function eig_nystrom(A::Matrix{Float64}, Q::Matrix{Float64})
	op = DeviceOperator(A)
	j = size(Q, 2)
	U = Matrix{Float64}(undef, op.m, j)
	Sigmavec = Vector{Float64}(undef, j)
	check(ccall((:gsi_eig_nystrom, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}),
		op.c.h, op.h, Q, j, U, Sigmavec))
	return U, Sigmavec
end
end # module RandMatFact

# ---- panel primitives: what the reference takes from LinearAlgebra (RandMatFact.jl:57-61, 86) ----------------------
"`LinearAlgebra.lu(Y).L` in pivoted row order (`gsi_lu_L`): returns (L, p) with p the 1-based LAPACK pivot rows."
function lu_L(Y::Matrix{Float64}; c::Context=ctx())
	m, l = size(Y)
	Lout = Matrix{Float64}(undef, m, l)
	piv = Vector{Int32}(undef, l)
	check(ccall((:gsi_lu_L, libgsi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Int32}),
		c.h, Y, m, l, Lout, piv))
	return Lout, piv .+ Int32(1)
end
"`lu(Y).L` of a device-resident panel, in place (`gsi_lu_L_dev`): returns the 1-based pivot rows."
function lu_L!(Y::DeviceMatrix)
	piv = Vector{Int32}(undef, Y.cols)
	check(ccall((:gsi_lu_L_dev, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}), Y.c.h, Y.h, piv))
	return piv .+ Int32(1)
end
"`Matrix(qr(Y, Val(true)).Q)` up to an orthogonal change of basis (`gsi_qr_thinQ`): returns (Q, R) with Y = Q R."
function qr_thinQ(Y::Matrix{Float64}; c::Context=ctx())
	m, l = size(Y)
	Q = Matrix{Float64}(undef, m, l)
	R = Matrix{Float64}(undef, l, l)
	check(ccall((:gsi_qr_thinQ, libgsi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}),
		c.h, Y, m, l, Q, R))
	return Q, R
end
"(S, V) of `svd(W')` for a tall W (`gsi_svd_tall`): what RandMatFact.jl:86 takes from `svd(B)`."
function svd_tall(W::Matrix{Float64}; c::Context=ctx())
	n, l = size(W)
	V = Matrix{Float64}(undef, n, l)
	S = Vector{Float64}(undef, l)
	check(ccall((:gsi_svd_tall, libgsi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}),
		c.h, W, n, l, V, S))
	return S, V
end

# ---- getxis (GeostatInversion.jl:20-70) ---------------------------------------------------------------
randsvdwithseed(Q, numxis, p, q, seed::Nothing) = RandMatFact.randsvd(Q, numxis, p, q)
function randsvdwithseed(Q, numxis, p, q, seed::Int)
	Random.seed!(seed)                                                               # GeostatInversion.jl:25
	return RandMatFact.randsvd(Q, numxis, p, q)
end

# columns of Z as the reference's Vector{Vector{Float64}}
xis_from(Z::Matrix{Float64}, numxis::Int) = [Z[:, i] for i in 1:numxis]

function getxis(::Type{Val{:iwantfields}}, samplefield::Function, numfields::Int, numxis::Int, p::Int, q::Int=3, seed=nothing)
	# sampling fans out over the workers like the reference's RobustPmap.rpmap (GeostatInversion.jl:30)
	fields = convert(Vector{Vector{Float64}}, Distributed.pmap(i->samplefield(), 1:numfields))
	Z = randsvdwithseed(LowRankCovMatrix(fields), numxis, p, q, seed)                 # :31-32
	return xis_from(Z, numxis), fields
end

getxis(samplefield::Function, numfields::Int, numxis::Int, p::Int, q::Int=3, seed=nothing) =
	first(getxis(Val{:iwantfields}, samplefield, numfields, numxis, p, q, seed))      # :58-61

getxis(Q::Matrix, numxis::Int, p::Int, q::Int=3, seed=nothing) =
	xis_from(randsvdwithseed(convert(Matrix{Float64}, Q), numxis, p, q, seed), numxis)   # :63-70

"`getxis` whose result stays in HBM: a `DeviceBasis` for the `pcgadirect` / `pcgalsqr` methods below."
function getxis_device(A::DeviceOperator, numxis::Int, p::Int, q::Int=3, seed=nothing; precision::Int=64)
	seed === nothing || Random.seed!(seed)
	Omega = DeviceMatrix(randn(A.n, numxis + p); c=A.c)
	Z = DeviceMatrix(A.n, numxis + p; c=A.c)
	check(ccall((:gsi_randsvd_dev, libgsi), Cint,
		(Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
		A.c.h, A.h, Omega.h, numxis, p, q, Z.h, C_NULL))
	return DeviceBasis(Z, numxis; precision=precision)
end

"`randsvd` with ROW-SHARDED panels over the ranks of `A.c`'s communicator (`gsi_randsvd_rows`): `Omega_rows` = this rank's rows
of Omega (`nloc x (K+p)`, the block layout `row0 = rank*ceil(n/nranks)`); returns this rank's rows of Z as a `DeviceMatrix`
-- nothing n x (K+p) is gathered for LowRankCovMatrix / FFT operators -- and S.  `DeviceBasis(Zrows, K)` over the result is a
row-sharded xi-basis: `paramstorun` / `update` then act on this rank's rows of s and X."
function randsvd_rows(A::DeviceOperator, Omega_rows::Matrix{Float64}, K::Int, p::Int, q::Int)
	Om = DeviceMatrix(Omega_rows; c=A.c)
	Z = DeviceMatrix(size(Omega_rows, 1), K + p; c=A.c)
	S = DeviceMatrix(K + p, 1; c=A.c)
	check(ccall((:gsi_randsvd_rows, libgsi), Cint,
		(Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
		A.c.h, A.h, Om.h, K, p, q, Z.h, S.h))
	Sh = Matrix{Float64}(undef, K + p, 1)
	check(ccall((:gsi_mat_download, libgsi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64), A.c.h, S.h, Sh, K + p))
	return Z, vec(Sh)
end

# ---- pcgadirect / pcgalsqr with a device-resident basis (same positional / keyword shape as direct.jl:21, lsqr.jl:20) ----
function iterationhead(forwardmodel::Function, s::Vector, X::Vector, xis::DeviceBasis, delta)
	K = length(xis)
	P = paramstorun(xis, s, X, delta)                                                   # direct.jl:39-45
	results = Distributed.pmap(i->forwardmodel(P[:, i]), 1:K + 3)                       # :46
	hs = results[K + 3]
	etas = hcat([(results[i] - hs) / delta for i in 1:K]...)                            # :49-50
	HX = (results[K + 1] - hs) / delta
	Hs = (results[K + 2] - hs) / delta
	return etas, HX, Hs, hs
end

function pcgadirect(forwardmodel::Function, s0::Vector, X::Vector, xis::DeviceBasis, R, y::Vector;
		maxiters::Int=5, delta::Float64=sqrt(eps(Float64)), xtol::Float64=1e-6, callback=(s, obs_cal)->nothing)
	s, converged, iters = s0, false, 0
	while !converged && iters < maxiters
		etas, HX, Hs, hs = iterationhead(forwardmodel, s, X, xis, delta)
		callback(s, hs)
		bigA = [(etas * etas' + R) HX; transpose(HX) 0.0]                               # direct.jl:57
		x = LinearAlgebra.pinv(Matrix(bigA)) * [y - hs + Hs; 0.0]                        # :56,58
		snew = update(xis, X, x[end], etas, x[1:end - 1])                                # :59-65
		converged = LinearAlgebra.norm(snew - s) < xtol
		s = snew
		iters += 1
	end
	return s
end

function pcgalsqr(forwardmodel::Function, s0::Vector, X::Vector, xis::DeviceBasis, R, y::Vector;
		maxiters::Int=5, delta::Float64=sqrt(eps(Float64)), xtol::Float64=1e-6)
	s, converged, iters = s0, false, 0
	while !converged && iters < maxiters
		etas, HX, Hs, hs = iterationhead(forwardmodel, s, X, xis, delta)
		x = lsqr(PCGALowRankMatrix(etas, HX, R; c=xis.Z.c), [y - hs + Hs; 0.0])         # lsqr.jl:52-54
		snew = update(xis, X, x[end], etas, x[1:end - 1])                                # :55-61
		converged = LinearAlgebra.norm(snew - s) < xtol
		s = snew
		iters += 1
	end
	return s
end

const pcga = pcgadirect                                                                  # GeostatInversion.jl:105

"`rga(forwardmodel, s0, X, xis, R, y, S; ...)`  (GeostatInversion.jl:101-103)"
rga(forwardmodel::Function, s0::Vector, X::Vector, xis::DeviceBasis, R, y::Vector, S; pcgafunc=pcgadirect, kwargs...) =
	pcgafunc(x->S * forwardmodel(x), s0, X, xis, S * R * S', S * y; kwargs...)

end # module
