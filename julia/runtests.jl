# runtests.jl -- the reference's own properties for this path, run through GeostatInversionHIP (libgsi_hip.so).
#
#     julia --project=. julia/runtests.jl            (needs an MI355X and the built library; GSI_HIP_LIB overrides the path)
#
# What is asserted is what the reference's suite asserts for RandMatFact / LowRankCovMatrix / getxis
# (test/testrmf.jl:5-29, test/testrpcga.jl:10-58, 83-102, 104-131), with the same sizes and tolerances, but every
# factorization runs on the GPU.  Where the reference package itself is importable the getxis test additionally
# compares with `GeostatInversion.getxis` on the same seed (same `randn` stream by construction).
#
# NOT executed in the build image (no Julia there): written against Julia >= 1.6 stdlib only.
import Test
import Random
import LinearAlgebra
import SparseArrays

include(joinpath(@__DIR__, "GeostatInversionHIP.jl"))
import .GeostatInversionHIP
const GH = GeostatInversionHIP
const RMF = GH.RandMatFact

const have_reference = try
	@eval import GeostatInversion
	true
catch
	false
end

exactrank(n, m) = randn(n, m) * randn(m, n)                     # a matrix of rank m exactly (test/testrmf.jl:5-9)

function rangefinders_recover_rank(n, m)
	A = exactrank(n, m)
	for Q in (RMF.rangefinder(A), RMF.rangefinder(A, m, 2))     # adaptive Alg 4.2 and fixed-rank with 2 power iterations
		Test.@test abs(size(Q, 2) - m) <= 1
		Test.@test LinearAlgebra.norm(A - Q * (Q' * A)) < 1e-8
	end
end

function nystrom_tridiagonal()
	A = Float64[2 -1 0; -1 2 -1; 0 -1 2]                          # eigenvalues 2 + sqrt 2, 2, 2 - sqrt 2
	U, Sigmavec = RMF.eig_nystrom(A, RMF.rangefinder(A))
	Test.@test LinearAlgebra.norm(Sigmavec .^ 2 - [2 + sqrt(2), 2, 2 - sqrt(2)]) < 1e-8
	Test.@test LinearAlgebra.norm(sort(LinearAlgebra.eigvals(A), rev=true) - Sigmavec .^ 2) < 1e-8
end

function lowrankcov_three_samples()
	samples = Vector{Float64}[[-.5, 0., .5], [1., -1., 0.], [-.5, 1., -.5]]
	lrcm = GH.LowRankCovMatrix(samples)
	Id = Matrix{Float64}(LinearAlgebra.I, 3, 3)
	full = Id * lrcm                                              # test/testrpcga.jl:49: Matrix{Float64}(I, 3, 3) * lrcm
	Test.@test full ≈ lrcm * Id && full ≈ Id * lrcm && full ≈ Id' * lrcm      # lowrank.jl:115-133: A*B, B*A, B'*A
	v = zeros(3)
	Test.@test LinearAlgebra.mul!(v, lrcm, [1., 2., 3.]) ≈ full * [1., 2., 3.]   # lowrank.jl:75-81
	Test.@test full ≈ sum(s * s' for s in samples) / (length(samples) - 1)
	Test.@test full ≈ [.75 -.75 0; -.75 1 -.25; 0 -.25 .25]
	Test.@test size(lrcm) == (3, 3) && size(lrcm, 1) == 3
	for i = 1:100
		x = randn(3, 3)
		Test.@test full * x ≈ lrcm * x
		Test.@test full' * x ≈ lrcm' * x
	end
	back = GH.samples(lrcm, 3)                                    # A.samples: mean-removed (these already are)
	Test.@test all(back[i] ≈ samples[i] for i = 1:3)
end

function lowrankcov_consistency(; N=10000, M=100)
	sq = randn(M, M)
	cov = sq * sq'
	samples = [sq * randn(M) for i = 1:N]
	lrcm = GH.LowRankCovMatrix(samples)
	full = lrcm * Matrix{Float64}(LinearAlgebra.I, M, M)
	Test.@test LinearAlgebra.opnorm(full - cov) < M^2 / sqrt(N) + 10
	for i = 1:100
		x = randn(M)
		Test.@test lrcm * x ≈ full * x
	end
end

# power-law fields on an nx x ny grid by spectral synthesis on the doubled periodic grid: the reference's FFTRF when it is
# there, a plain-Julia stand-in of the same kind otherwise (only "some smooth random fields" is needed here)
function samplefield_factory(Ns)
	if have_reference
		return () -> GeostatInversion.FFTRF.powerlaw_structuredgrid(Ns, 2., 3.14, -3.5)[1:end]
	end
	return function ()
		n1, n2 = Ns
		f = zeros(n1, n2)
		for k1 = 0:5, k2 = 0:5
			(k1 == 0 && k2 == 0) && continue
			amp = (k1^2 + k2^2)^(-3.5 / 4)
			ph1, ph2 = 2pi * rand(), 2pi * rand()
			for j = 1:n2, i = 1:n1
				f[i, j] += amp * cos(2pi * k1 * i / (2n1) + ph1) * cos(2pi * k2 * j / (2n2) + ph2) * randn()
			end
		end
		return vec(f)
	end
end

function getxis_lowrank_vs_dense(; numfields=100, numxis=30, p=20, Ns=[25, 25])
	samplefield = samplefield_factory(Ns)
	lrcmxis, fields = GH.getxis(Val{:iwantfields}, samplefield, numfields, numxis, p, 3, 0)
	lrcm = GH.LowRankCovMatrix(fields)
	full = Matrix{Float64}(LinearAlgebra.I, size(lrcm, 1), size(lrcm, 1)) * lrcm      # test/testrpcga.jl:90
	fullxis = GH.getxis(full, numxis, p, 3, 0)                    # same seed -> same Omega in both calls
	for i = eachindex(fullxis)
		Test.@test min(LinearAlgebra.norm(fullxis[i] - lrcmxis[i]), LinearAlgebra.norm(fullxis[i] + lrcmxis[i])) < 1e-6
	end
	if have_reference                                              # GPU path against the reference itself, same seed
		refxis = GeostatInversion.getxis(full, numxis, p, 3, 0)
		for i = eachindex(refxis)
			Test.@test min(LinearAlgebra.norm(refxis[i] - fullxis[i]), LinearAlgebra.norm(refxis[i] + fullxis[i])) < 1e-6
		end
	end
end

function pcga_lowrank_matrix(; numetas=10, numobs=20)
	for noiselevel in (1e16, 0.), etagen in (zeros, randn), HXgen in (zeros, randn)
		etas = hcat([etagen(numobs) for i = 1:numetas]...)
		HX = HXgen(numobs)
		R = noiselevel * SparseArrays.sparse(LinearAlgebra.I, numobs, numobs)
		dense = [(etas * etas' + R) HX; HX' 0.0]
		A = GH.PCGALowRankMatrix(etas, HX, R)
		Test.@test size(A) == (size(A, 1), size(A, 2)) == (numobs + 1, numobs + 1)
		for i = 1:numobs + 1
			e = zeros(numobs + 1)
			e[i] = 1.
			Test.@test dense * e ≈ A * e
		end
	end
end

function pcga_end_to_end(M::Int, N::Int, mu::Float64=0.)
	x = randn(N)
	Q0 = randn(M, N)
	Q = Q0' * Q0
	truep = real(sqrt(LinearAlgebra.Symmetric(Q)) * randn(N)) .+ mu
	forward(p::Vector) = p .* x
	noise = 1e-4
	yobs = forward(truep) + noise * randn(N)
	R = noise^2 * SparseArrays.sparse(LinearAlgebra.I, N, N)
	X, p0 = fill(mu, N), fill(mu, N)
	basis = GH.getxis_device(GH.DeviceOperator(Q), M, round(Int, 0.1 * M))     # q = 3, unseeded, like the reference
	popt = GH.pcgadirect(forward, p0, X, basis, R, yobs)
	Test.@test LinearAlgebra.norm(popt - truep) / LinearAlgebra.norm(truep) < 2e-2
	if M < N / 6
		popt = GH.pcgalsqr(forward, p0, X, basis, R, yobs)
		Test.@test LinearAlgebra.norm(popt - truep) / LinearAlgebra.norm(truep) < 2e-2
	end
end

Test.@testset "GeostatInversionHIP" begin
	Random.seed!(2017)
	Test.@testset "RMF" begin
		for (n, m) in ((10, 2), (10, 5), (100, 5), (100, 10), (100, 25))
			rangefinders_recover_rank(n, m)
		end
		nystrom_tridiagonal()
		Test.@test_throws ErrorException RMF.rangefinder(GH.DeviceOperator(exactrank(10, 2)), 2, -1)   # RandMatFact.jl:62-64
	end
	Test.@testset "LowRankCovMatrix / getxis" begin
		lowrankcov_three_samples()
		lowrankcov_consistency()
		getxis_lowrank_vs_dense()
	end
	Test.@testset "PCGA" begin
		pcga_lowrank_matrix()
		for log2N = 2:8, log2M = 0:log2N - 1
			pcga_end_to_end(2^log2M, 2^log2N)
			pcga_end_to_end(2^log2M, 2^log2N, 10.)
		end
	end
end
