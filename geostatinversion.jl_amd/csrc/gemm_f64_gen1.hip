// gemm_f64_gen1.hip -- the contraction kernel's instantiations for the TABLE-GENERATED operand (GEN 1: the implicit stationary
// covariance on a regular grid, SURVEY.md 8d "implicit"), compiled from the kernel template with the 64-bit
// tile counters that form was tuned with.  Why a second translation unit: gemm_f64_kernel.inc.hpp's header.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <algorithm>
#include "hip_common.hpp"
#include "pointcov.hpp"
#include "pointcov_gen.hpp"

namespace gsi { namespace hipk {

#define GSI_GEMM_TILE_COUNTERS_32 0
#include "gemm_f64_kernel.inc.hpp"

void gemm_dispatch_gen1(int nt, dim3 grid, hipStream_t st, int64_t M, int64_t L, int64_t K, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double* C, int64_t ldc, double alpha, double beta, double* slabs,
                        int64_t kchunk, int nchunks_x, int wide, int xmode, int tri, const GenA& gen, int64_t nitems) {
  launch_dispatch<false, 1>(nt, grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk, nchunks_x, wide, xmode, tri,
                            gen, nitems);
}

}}  // namespace gsi::hipk
