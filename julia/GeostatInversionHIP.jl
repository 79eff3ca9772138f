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
const default_ctx = Ref{Union{Nothing, Context}}(nothing)
ctx() = something(default_ctx[], (default_ctx[] = Context(0)))

"Join `nranks` contexts (one Julia process per GPU) into an RCCL communicator; rank 0 creates the id."
function unique_id()
	id = zeros(UInt8, 128)
	check(ccall((:gsi_comm_unique_id, libgsi), Cint, (Ptr{UInt8},), id))
	return id
end
comm_init!(c::Context, nranks::Integer, rank::Integer, id::Vector{UInt8}) =
	check(ccall((:gsi_ctx_comm_init, libgsi), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), c.h, nranks, rank, id))

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

"Exact, matrix-free covariance of `FFTRF.powerlaw_structuredgrid(Ns, k0, dk, beta)` fields up to the factor dk^2
(circulant embedding, spectrum |k|^beta, unit diagonal): `gsi_op_fft_powerlaw`.  Acts on `vec(field)`."
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
rangefinder(A::Matrix{Float64}, l::Int64, numiterations::Int64) = rangefinder(DeviceOperator(A), l, numiterations)

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
randsvd(A::Matrix{Float64}, K::Int, p::Int, q::Int) = randsvd(DeviceOperator(A), K, p, q)

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

# ---- getxis (GeostatInversion.jl:20-70) ---------------------------------------------------------------
randsvdwithseed(Q, numxis, p, q, seed::Nothing) = RandMatFact.randsvd(Q, numxis, p, q)
function randsvdwithseed(Q, numxis, p, q, seed::Int)
	Random.seed!(seed)                                                               # GeostatInversion.jl:25
	return RandMatFact.randsvd(Q, numxis, p, q)
end

function getxis(::Type{Val{:iwantfields}}, samplefield::Function, numfields::Int, numxis::Int, p::Int, q::Int=3, seed=nothing)
	fields = Array{Float64, 1}[samplefield() for i = 1:numfields]                    # the reference uses RobustPmap.rpmap (:30)
	lrcm = LowRankCovMatrix(fields)
	Z = randsvdwithseed(lrcm, numxis, p, q, seed)
	xis = Array{Array{Float64, 1}}(undef, numxis)
	for i = 1:numxis
		xis[i] = Z[:, i]
	end
	return xis, fields
end

function getxis(samplefield::Function, numfields::Int, numxis::Int, p::Int, q::Int=3, seed=nothing)
	xis, _ = getxis(Val{:iwantfields}, samplefield, numfields, numxis, p, q, seed)
	return xis
end

function getxis(Q::Matrix, numxis::Int, p::Int, q::Int=3, seed=nothing)
	xis = Array{Array{Float64, 1}}(undef, numxis)
	Z = randsvdwithseed(convert(Matrix{Float64}, Q), numxis, p, q, seed)
	for i = 1:numxis
		xis[i] = Z[:, i]
	end
	return xis
end

end # module
