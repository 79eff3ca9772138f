# make_reference_fixtures.jl -- run the REFERENCE (GeostatInversion.jl's own RandMatFact, on the CPU) on fixed seeds and
# write its inputs and outputs as .npy files the Python tests pick up (tests/test_reference_fixtures.py):
#
#     julia --project=<an environment with GeostatInversion.jl> julia/make_reference_fixtures.jl [outdir=tests/golden]
#
# This is the ONLY way the oracle (oracle/oracle.py) and the HIP path can be pinned element-wise to the reference: the
# build image has no Julia, so nothing here has been executed there; the fixtures are absent until someone runs this.
#
# Cases (SURVEY.md 8d):
#   ref_c1/        dense Gaussian covariance of a 50 x 40 unit grid (n = 2000, ell = 5), K = 32, p = 16, q = 1, seed 0
#                  (BASELINE.json configs[0])
#   ref_lrcm625/   LowRankCovMatrix over 100 FFTRF power-law fields on a 25 x 25 grid, K = 30, p = 20, q = 3, seed 0
#                  (test/testrpcga.jl:83-102) -- the fields are saved, so no FFTRF is needed to replay it
# Each directory holds Omega.npy (what `Random.seed!(seed); randn(n, K + p)` gives: the stream RandMatFact.jl:54 consumes
# after GeostatInversion.jl:25), Z.npy (the reference's randsvd output), S.npy (S_i = |Z[:, i]|^2 for i <= K, since
# Z = V sqrt(S), RandMatFact.jl:87-88), and the operator's data (A.npy or fields.npy) plus params.txt.
import GeostatInversion
import Random
import LinearAlgebra

# NumPy .npy v1.0, Float64, Fortran order (Julia's memory layout as it is)
function writenpy(path::String, A::AbstractArray{Float64})
	shape = join(size(A), ", ") * (ndims(A) == 1 ? "," : "")
	dict = "{'descr': '<f8', 'fortran_order': True, 'shape': ($shape), }"
	pad = 64 - mod(10 + length(dict) + 1, 64)
	header = dict * " "^pad * "\n"
	open(path, "w") do io
		write(io, UInt8[0x93], "NUMPY", UInt8[1, 0], UInt16(length(header)))
		write(io, header)
		write(io, Array(A))
	end
end

function reference_randsvd(A, K, p, q, seed)
	n = size(A, 2)
	Random.seed!(seed)
	Omega = randn(n, K + p)                                   # the draw the reference is about to make
	Z = GeostatInversion.randsvdwithseed(A, K, p, q, seed)    # GeostatInversion.jl:24-27 -> RandMatFact.randsvd
	S = [sum(abs2, Z[:, i]) for i = 1:K]
	return Omega, Z, S
end

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")

let nx = 50, ny = 40, ell = 5.0, K = 32, p = 16, q = 1, seed = 0
	pts = [(Float64(i), Float64(j)) for i = 0:nx - 1 for j = 0:ny - 1]        # point index = i * ny + j
	A = [exp(-((a[1] - b[1])^2 + (a[2] - b[2])^2) / (2 * ell^2)) for a in pts, b in pts]
	Omega, Z, S = reference_randsvd(A, K, p, q, seed)
	d = mkpath(joinpath(outdir, "ref_c1"))
	writenpy(joinpath(d, "A.npy"), A); writenpy(joinpath(d, "Omega.npy"), Omega)
	writenpy(joinpath(d, "Z.npy"), Z); writenpy(joinpath(d, "S.npy"), S)
	write(joinpath(d, "params.txt"), "K=$K p=$p q=$q seed=$seed nx=$nx ny=$ny ell=$ell julia=$(VERSION)\n")
end

let numfields = 100, K = 30, p = 20, q = 3, seed = 0
	Random.seed!(2017)
	fields = [GeostatInversion.FFTRF.powerlaw_structuredgrid([25, 25], 2., 3.14, -3.5)[1:end] for i = 1:numfields]
	lrcm = GeostatInversion.LowRankCovMatrix(fields)
	Omega, Z, S = reference_randsvd(lrcm, K, p, q, seed)
	d = mkpath(joinpath(outdir, "ref_lrcm625"))
	writenpy(joinpath(d, "fields.npy"), hcat(fields...))                      # 625 x 100, field i = column i
	writenpy(joinpath(d, "Omega.npy"), Omega)
	writenpy(joinpath(d, "Z.npy"), Z); writenpy(joinpath(d, "S.npy"), S)
	write(joinpath(d, "params.txt"), "K=$K p=$p q=$q seed=$seed numfields=$numfields grid=25x25 julia=$(VERSION)\n")
end
println("reference fixtures written under ", outdir)
