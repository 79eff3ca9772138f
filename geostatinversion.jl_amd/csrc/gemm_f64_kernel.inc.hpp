// gemm_f64_kernel.inc.hpp -- the contraction kernel template and its launchers, included by TWO translation units:
//   gemm_f64.hip       (GSI_GEMM_TILE_COUNTERS_32 = 1): the stored operand (GEN 0) and the scattered-point generator (GEN 2)
//   gemm_f64_gen1.hip  (GSI_GEMM_TILE_COUNTERS_32 = 0): the table-generated operand (GEN 1)
// At 256 VGPRs hipcc's register allocation of this tile loop moves by whole percents with edits that change nothing for the
// instantiation at hand: 32-bit tile counters gained 1-5 % on the stored forms and cost the table-generated
// form 3 % -- in the same template, whatever the spelling.  So the generated form is compiled from the text it was tuned
// with: the #else branches below are that text, byte for byte.  Not a stand-alone header: the including file provides the
// includes and the namespace.
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int MT = 2;                 // 16-row tiles per wave (wave tile = 16*MT rows x NT/2*16 columns)
constexpr int BMT = 4 * 16 * MT;      // C rows per workgroup (4 row groups)
constexpr int NTHREADS = 512;
constexpr int BK = 32;                // reduction depth per LDS tile
constexpr int BKP = BK + 2;           // padded k stride (doubles) of the [col][k] images (BKP/2 odd)
constexpr int BMP = BMT + 16;         // padded row stride (doubles) of the NN A image [k][r]
constexpr int NTMAX = 10;             // 16-column tiles per workgroup pass (160 columns)
constexpr int NSETS = 2;              // staging register sets: tiles are fetched NSETS tiles ahead
constexpr int A_PAIRS = BMT * BK / (2 * NTHREADS);   // 16-byte pairs per thread per A tile
constexpr int KSTEP_NN = 2 * NTHREADS / BMT;          // NN A tile: k advance per pair slot
constexpr int RSTEP = 2 * NTHREADS / BK;              // TN A tile / B tile: row (column) advance per pair slot
static_assert(A_PAIRS * 2 * NTHREADS == BMT * BK, "tile does not divide over the threads");

// GEN: the big operand is never stored.  A(i, j) = t2[|x_i - x_j| * ny + |y_i - y_j|] for grid points
// i = (i / ny, i % ny): any stationary covariance on a regular grid (SURVEY.md 8d "implicit" configuration), given as
// the nx * ny table of the kernel over grid offsets.  The staging registers are filled from the table instead of from
// HBM, everything after that (LDS images, fragments, MFMAs) is the stored-operand kernel unchanged.  Consecutive rows
// and consecutive reduction indices are consecutive y, so a tile touches one or two contiguous runs of a table row:
// the lookups are L1 hits.  (A first version kept separable kernels as two 1-D tables and multiplied; one 2-D lookup
// covers the non-separable kernels too -- exponential, Matern -- and costs one load instead of two and a multiply.)
struct GenA {
  const double* t2;   // GEN 1: nx * ny entries of the kernel over grid offsets.  GEN 2: the points, 4 doubles each (x, y, z, 0)
  int32_t ny;         // GEN 2: number of points
  int32_t kind;       // GEN 2: pointcov::GAUSSIAN ...
  int64_t roff;       // global index of row 0 of the product
  int64_t koff;       // global index of reduction index 0
  double inv_ell2, sigma2, nugget;   // GEN 2 (inv_ell2: unused since the points arrive pre-scaled)
  int32_t dim;        // GEN 2: coordinates per point (<= 3)
};

// GEN 2: the covariance of SCATTERED points, A(i, j) = sigma2 k(|p_i - p_j| / ell) (+ nugget on the diagonal), evaluated where
// the stored-operand kernel would write a staged tile into LDS (round 4; SURVEY.md 8b "entries generated in the tile loader").
// Round 3 generated row panels of A into HBM on a second stream and contracted them with the stored-operand kernel; the two
// never overlapped (the contraction fills the register file of every CU), so a panel cost generation + contraction:
// 46.6 TFLOP/s against 62.6 for the table-based operator.  Here a thread owns two rows (their coordinates stay in
// registers for the whole kernel), the column point of a pair slot is wave-uniform (scalar loads of 32-byte records), and
// the ~35 VALU instructions per entry issue in the shadow of the partner wave's MFMAs (an entry feeds 2 l flops of matrix
// work).  Nothing is prefetched for A -- there is no latency to hide -- so the staging registers of the stored operand are free
// for the coordinates and the polynomial.  One definition of the kernels: pointcov.hpp.
// The entry itself (pre-scaled points, table-driven exponential, 27 vector instructions): pointcov_gen.hpp.
// RAGGED: the "irregular X" instantiation.  Either the sketch width is not a multiple of 16 (K + p is the
// caller's choice), so the last columns of the X tile do not exist, or X is only 8-byte aligned (n odd as its
// leading dimension).  It keeps the 16-byte stream of the operator and loads the X pairs per column, predicated,
// with 8-byte alignment (measured at l = 150: 28.0 -> 21.3 ms); folding that into the regular kernel as a
// second fast path cost the regular case 2.7 %, hence the template parameter.
// XMODE 2 = irregular X with 64-bit per-thread offsets: leading dimensions beyond the reach of the 32-bit tile
// offsets (160 columns * ld * 8 B >= 4 GiB, i.e. panels of more than ~3.3 million rows).  Slower addressing, same code.
template <int NT, bool TRANS_A, int GEN, int XMODE>
__global__ __launch_bounds__(NTHREADS) void gemm_f64_kernel(
    int64_t M, int64_t L, int64_t K, const double* __restrict__ A, int64_t lda,
    const double* __restrict__ B, int64_t ldb, double* __restrict__ C, int64_t ldc, double alpha,
    double beta, double* __restrict__ slabs, int64_t kchunk, int nchunks_x, int wide, int tri, GenA gen, int64_t nitems) {
  static_assert(!(GEN != 0 && TRANS_A), "the generated operand is symmetric: only the NN form exists");
  // PERSISTENT mode (round 4; nitems > 0: stored operand, regular X, one K split, no triangle): the grid is one workgroup
  // per CU and a workgroup walks the output tiles blockIdx.x, + gridDim.x, ... < nitems.  Short reductions are what it is for
  // (the S T product of a LowRankCovMatrix, K = N_s = 1024: 32 tiles per workgroup; Z = T M, K = 320: 10): per output tile a
  // one-shot workgroup exposed its two-tile prologue and the 164 KB of result stores -- 0.54 ms of a 10.25 ms launch (time
  // against K: slope 9.715 ms per 1024, `tools/bench_lrcm_products.py --samples 512 ... 4096`).  Here the next tile's first two
  // operand tiles are requested BEFORE the result stores are issued (vmcnt retires in order on gfx9: loads queued behind 80
  // stores would wait for all of them; the staging registers are dead at that point, so this costs no register), and the
  // stores drain under the next tile's matrix work.
  constexpr bool CANP = (GEN == 0 && XMODE == 0);
  const bool persist = CANP && nitems > 0;
  constexpr bool RAGGED = XMODE != 0;
  constexpr bool BIG = XMODE == 2;
  using off_t = typename std::conditional<BIG, uint64_t, uint32_t>::type;
  constexpr int A_ELEMS = TRANS_A ? BMT * BKP : BK * BMP;
  constexpr int B_ELEMS = NT * 16 * BKP;
  constexpr int BUF_ELEMS = A_ELEMS + B_ELEMS;
  extern __shared__ double smem[];   // [GEN 2: 64-entry table] [2][A tile | B tile]
  // LDS0: a constant in every index, NOT a second pointer `smem_raw + 64`: with the derived pointer the stored-operand
  // instantiations (offset 0!) compiled differently and ran 4 % (NN) and 24 % (TN: 9.5 -> 11.8 ms) slower -- this kernel sits on a
  // register-allocation cliff; any edit is A/B-ed against the previous build in one process (tools/bench_lrcm_products.py with
  // GSI_HIP_LIB), a lesson of round 4.
  constexpr int LDS0 = (GEN == 2) ? 64 : 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int rg = wave & 3;    // row group: C rows 16*MT*rg .. of the workgroup tile
  const int ch = wave >> 2;   // column half: 16-column tiles ch*NTW .. of the workgroup's NT
  constexpr int NTW = (NT + 1) / 2;
  const int jl = lane & 15;   // MFMA "column" index -> C row within a 16-row tile
  const int kk = lane >> 4;   // MFMA k index within a k4 step
  // ---- workgroup -> (tile, K split), XCD-aware.  Workgroups are dealt round-robin over the 8 XCDs in dispatch order
  // (linear id % 8), each XCD has its own L2.  Two cases re-read a streamed operand from the fabric once per XCD
  // unless the workgroups that share it sit on the SAME XCD (counters at n = 1e6, N_s = 1024, l = 320: S'X moved 29.7
  // GB for 10.75 GB of operands, S T 19.6 GB):
  //   split-K (S'X: 16 tiles x 16 splits): the tiles of one K range share both operand slabs -> split s on XCD s % 8;
  //   column chunks of one row block (S T: 2 chunks): both read the same rows of the operator -> same XCD, 8 apart.
  int64_t tile_lin = blockIdx.x;
  int split = (int)blockIdx.y;
  int64_t r0 = 0, c0 = 0;
  // bx: the linear x id of the work item (blockIdx.x, or the item index of the persistent mode: gridDim.x is a multiple of 8
  // there, so item id % 8 is still the XCD the workgroup runs on)
  auto locate = [&](int64_t bx) __attribute__((always_inline)) {
    tile_lin = bx;
    const int T = (int)gridDim.x, S = (int)gridDim.y;
    if (S > 1 && (S & 7) == 0) {
      const int lin = (int)bx + T * (int)blockIdx.y;
      const int xcd = lin & 7, j = lin >> 3;
      split = xcd + 8 * (j / T);
      tile_lin = j % T;
    } else if (S == 1 && tri != 1 && nchunks_x > 1) {
      const int C8 = 8 * nchunks_x;
      const int nrb = (int)((M + BMT - 1) / BMT);
      const int grp = (int)(bx / C8), r = (int)(bx % C8);
      const int rows_here = (grp * 8 + 8 <= nrb) ? 8 : (nrb - grp * 8);      // the last group may be short
      const int rb = grp * 8 + r % rows_here, chunk = r / rows_here;
      tile_lin = (int64_t)rb * nchunks_x + chunk;
    }
    r0 = (int64_t)(tile_lin / nchunks_x) * BMT;
    c0 = (int64_t)(tile_lin % nchunks_x) * (NT * 16);
  };
  locate(blockIdx.x);
  if (tri == 1) {
    // symmetric product: the tile index counts only the tiles that touch the upper triangle (tiles entirely below the
    // diagonal are not part of the grid -- as idle workgroups they pushed the grid past one round of 256)
    int left = (int)tile_lin;
    const int nrb = (int)((M + BMT - 1) / BMT);
    for (int rb = 0; rb < nrb; ++rb) {
      const int first = (rb * BMT) / (NT * 16);           // first chunk with a column at or right of the block's first row
      const int cnt = nchunks_x - first;
      if (left < cnt) { r0 = (int64_t)rb * BMT; c0 = (int64_t)(first + left) * (NT * 16); break; }
      left -= cnt;
    }
  }
  // tri == 1 (TN, C = A'A symmetric): tiles entirely below the diagonal are not computed (the caller mirrors the
  //   upper triangle); tri == 2 (NN, B upper triangular): rows of B below the chunk's last column are zero, so the
  //   reduction stops there.  CholeskyQR spends 4 n l^2 flop instead of 8 n l^2 this way.
  int64_t kbeg = (int64_t)split * kchunk;
  int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  if (tri == 2 && kend > c0 + NT * 16) kend = c0 + NT * 16;
#if GSI_GEMM_TILE_COUNTERS_32
  // tile counts as 32-bit scalars: the 64-bit form of `t + 1 < ntiles` / `k0 + BK <= kend` compiles to vector compares
  // (v_mov_b64 + v_cmp_*_u64: there is no 64-bit s_cmp_lt), a dozen vector instructions per tile.  Measured in one process
  // against the previous build: S'X 9.53 -> 9.45 ms, S T 10.24 -> 10.13, dense 65536^2 at l = 144: 19.2 -> 18.7 / 21.2 -> 20.1;
  // the table-generated operand (GEN 1) 416 -> 428 ms per product, and no spelling of this code that keeps its old form
  // brought that back -- the allocator decides, not the source.
  const int ntiles = (kend > kbeg) ? (int)((kend - kbeg + BK - 1) / BK) : 0;
  const int nfull = (kend > kbeg) ? (int)((kend - kbeg) / BK) : 0;          // tiles with all BK reduction indices in range
#else
  int64_t ntiles = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
#endif

  double4_t acc[MT][NTW];
#pragma unroll
  for (int h = 0; h < MT; ++h)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[h][t] = (double4_t){0.0, 0.0, 0.0, 0.0};

  // Staging registers: NSETS sets (tiles are loaded NSETS tiles ahead).  Every thread owns PAIRS of
  // elements adjacent along the contiguous dimension, so that with 16-byte-aligned operands
  // (wide != 0: base pointers 16-B aligned, even leading dimensions) each pair is one
  // global_load_dwordx4 and one ds_write_b128 -- half the VMEM / LDS-write instructions.
  constexpr int B_PAIRS = (NT * 16 + RSTEP - 1) / RSTEP;
  constexpr bool B_RAGGED = (NT * 16) % RSTEP != 0;
  double2 a_reg[NSETS][A_PAIRS];
  double2 b_reg[NSETS][B_PAIRS];

  // NN A tile: pair (r = 2*(tid % (BMT/2)), k = tid/(BMT/2) + KSTEP_NN*it)
  // TN A tile: pair (k = 2*(tid % (BK/2)),  r = tid/(BK/2) + RSTEP*it)
  // B tile   : pair (k = 2*(tid % (BK/2)),  c = tid/(BK/2) + RSTEP*it)
  const int a_r = TRANS_A ? (tid / (BK / 2)) : 2 * (tid % (BMT / 2));
  const int a_k = TRANS_A ? 2 * (tid % (BK / 2)) : (tid / (BMT / 2));
  const int b_c = tid / (BK / 2);
  const int b_k = 2 * (tid % (BK / 2));

  const off_t a_off0 = (off_t)8 * (TRANS_A ? (off_t)(a_k + (int64_t)a_r * lda) : (off_t)(a_r + (int64_t)a_k * lda));
  const off_t a_step_c = (off_t)8 * (off_t)((TRANS_A ? RSTEP : KSTEP_NN) * lda);
  const off_t b_off0 = (off_t)8 * (off_t)(b_k + (int64_t)b_c * ldb);
  const off_t b_step_c = (off_t)8 * (off_t)(RSTEP * ldb);
  // byte distance between the two elements of a pair when they cannot be fetched as one 16-B load
  const char* Abase = reinterpret_cast<const char*>(TRANS_A ? A + r0 * lda : A + r0);
  const char* Bbase = reinterpret_cast<const char*>(B + c0 * ldb);
  // interior workgroups (all 128 rows and all NT*16 columns in range) take an unpredicated
  // load path on full-depth tiles; the branch is workgroup-uniform
  bool wg_full = (r0 + BMT <= M) && (RAGGED || c0 + NT * 16 <= L);
  auto setup_item = [&]() __attribute__((always_inline)) {      // persistent mode: everything that depends on the output tile
    Abase = reinterpret_cast<const char*>(TRANS_A ? A + r0 * lda : A + r0);
    Bbase = reinterpret_cast<const char*>(B + c0 * ldb);
    wg_full = (r0 + BMT <= M) && (RAGGED || c0 + NT * 16 <= L);
  };

  // GEN: grid coordinates of this thread's two rows (fixed for the whole kernel), and of the reduction
  // index the NEXT prefetched pair slot covers.  The latter is wave-uniform (a_k = wave index), lives in
  // SGPRs and is advanced incrementally -- prefetch() is called on consecutive tiles, in order.
  int g_x0 = 0, g_y0 = 0, g_x1 = 0, g_y1 = 0, g_kx = 0, g_ky = 0;
  double p0x = 0.0, p0y = 0.0, p0z = 0.0, p1x = 0.0, p1y = 0.0, p1z = 0.0;      // GEN 2: this thread's two row points
  int64_t g_row0 = 0;
  GenPointK gq{};
  double* const gtab = smem;                        // GEN 2: sigma^2 2^(j/64) / 120, j = 0..63, at LDS offset 0 (the lookup's address is the index)
  if constexpr (GEN == 2) {                        // the points ride in the (otherwise unused) A argument: const __restrict__
    g_row0 = gen.roff + r0 + a_r;
    const int64_t i0 = (g_row0 < gen.ny) ? g_row0 : (int64_t)gen.ny - 1, i1 = (g_row0 + 1 < gen.ny) ? g_row0 + 1 : (int64_t)gen.ny - 1;
    p0x = A[4 * i0]; p0y = A[4 * i0 + 1]; p0z = A[4 * i0 + 2];
    p1x = A[4 * i1]; p1y = A[4 * i1 + 1]; p1z = A[4 * i1 + 2];
    gq = gen_point_setup(gen.dim, gen.kind);
    gen_table_init(gtab, tid, gen.sigma2);
    __syncthreads();
  }
  // GEN 1 (round 4): everything in BYTE offsets into the table, x coordinates pre-multiplied by the row length -- the offset of
  // an entry is |px - qx| + |py - qy| = two v_sad_u32 (sum of absolute differences) instead of eleven integer instructions
  // with a quarter-rate multiply: on gfx950 every VALU instruction of this loop is matrix time lost (fp64 MFMAs hold the
  // vector ALU), and the table operator ran at 0.80 of the peak where the stored one reaches 0.865.
  uint32_t t_px0 = 0, t_py0 = 0, t_px1 = 0, t_py1 = 0, t_qx = 0, t_qy = 0, t_ny8 = 0;
  if constexpr (GEN == 1) {
    const int64_t gr = gen.roff + r0 + a_r;
    g_x0 = (int)(gr / gen.ny); g_y0 = (int)(gr % gen.ny);
    g_x1 = g_x0; g_y1 = g_y0 + 1;
    if (g_y1 == gen.ny) { g_y1 = 0; g_x1 = g_x0 + 1; }
    const int64_t gk = gen.koff + kbeg + __builtin_amdgcn_readfirstlane(a_k);
    g_kx = __builtin_amdgcn_readfirstlane((int)(gk / gen.ny));
    g_ky = __builtin_amdgcn_readfirstlane((int)(gk % gen.ny));
    t_ny8 = 8u * (uint32_t)gen.ny;
    t_px0 = (uint32_t)g_x0 * t_ny8; t_py0 = 8u * (uint32_t)g_y0;
    t_px1 = (uint32_t)g_x1 * t_ny8; t_py1 = 8u * (uint32_t)g_y1;
    t_qx = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)g_kx * t_ny8));
    t_qy = (uint32_t)__builtin_amdgcn_readfirstlane((int)(8u * (uint32_t)g_ky));
  }

#if GSI_GEMM_TILE_COUNTERS_32
  auto prefetch = [&](int tt, auto SET) __attribute__((always_inline)) {      // tt: tile index within this K range
#else
  auto prefetch = [&](int64_t k0, auto SET) __attribute__((always_inline)) {
#endif
    constexpr int set = decltype(SET)::value;
#if GSI_GEMM_TILE_COUNTERS_32
    const int64_t k0 = kbeg + (int64_t)tt * BK;
    const bool tile_full = tt < nfull;
#else
#endif
    if constexpr (GEN == 1) {
      const int64_t kfirst = k0 + __builtin_amdgcn_readfirstlane(a_k);
      auto advance = [&]() {                       // scalar: the reduction index of the next pair slot, in table bytes
        t_qy += 8u * KSTEP_NN;
        while (t_qy >= t_ny8) { t_qy -= t_ny8; t_qx += t_ny8; }
      };
      const char* const tb = reinterpret_cast<const char*>(gen.t2);
      auto sad = [](uint32_t a, uint32_t b, uint32_t c) -> uint32_t { return ((a > b) ? a - b : b - a) + c; };   // v_sad_u32
      auto entry = [&](uint32_t px, uint32_t py) -> double {
        return *reinterpret_cast<const double*>(tb + sad(py, t_qy, sad(px, t_qx, 0u)));
      };
#if GSI_GEMM_TILE_COUNTERS_32
      if (r0 + BMT <= M && tile_full) {      // interior: no predicates (workgroup-uniform branch)
#else
      if (r0 + BMT <= M && k0 + BK <= kend) {      // interior: no predicates (workgroup-uniform branch)
#endif
#pragma unroll
        for (int it = 0; it < A_PAIRS; ++it) {
          a_reg[set][it].x = entry(t_px0, t_py0);
          a_reg[set][it].y = entry(t_px1, t_py1);
          advance();
        }
      } else {
        const bool ok_r0 = r0 + a_r < M, ok_r1 = r0 + a_r + 1 < M;
#pragma unroll
        for (int it = 0; it < A_PAIRS; ++it) {
          const bool okk = kfirst + KSTEP_NN * it < kend;   // uniform
          a_reg[set][it].x = (okk && ok_r0) ? entry(t_px0, t_py0) : 0.0;
          a_reg[set][it].y = (okk && ok_r1) ? entry(t_px1, t_py1) : 0.0;
          advance();
        }
      }
    }
    const char* Ab = Abase + 8 * (TRANS_A ? k0 : k0 * lda);   // uniform
    const char* Bb = Bbase + 8 * k0;                          // uniform
    // launder the strides so the per-load offsets are recomputed per tile (one VALU add each)
    // instead of staying live in VGPRs across the MFMA loop
    off_t a_step = a_step_c, b_step = b_step_c;
    if constexpr (!BIG) asm volatile("" : "+s"(a_step), "+s"(b_step));
#if GSI_GEMM_TILE_COUNTERS_32
    if (wide && wg_full && tile_full) {
#else
    if (wide && wg_full && k0 + BK <= kend) {
#endif
      if constexpr (GEN == 0) {
#pragma unroll
      for (int it = 0; it < A_PAIRS; ++it)
        {   // A is streamed once: non-temporal, to keep it out of the way of the X tiles in L2
          typedef double nt_double2 __attribute__((ext_vector_type(2)));
          const nt_double2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_double2*>(Ab + (a_off0 + (off_t)it * a_step)));
          a_reg[set][it].x = v.x; a_reg[set][it].y = v.y;
        }
      }
      if constexpr (RAGGED) {
#pragma unroll
        for (int it = 0; it < B_PAIRS; ++it) {
          const int cl = b_c + RSTEP * it;
          if (cl < NT * 16 && c0 + cl < L) {   // X may be only 8-byte aligned here (odd n as leading dimension)
            typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));
            const double2_u v = *reinterpret_cast<const double2_u*>(Bb + (b_off0 + (off_t)it * b_step));
            b_reg[set][it] = make_double2(v.x, v.y);
          } else {
            b_reg[set][it] = make_double2(0.0, 0.0);
          }
        }
        return;
      }
#pragma unroll
      for (int it = 0; it < B_PAIRS; ++it)
        if (!B_RAGGED || b_c + RSTEP * it < NT * 16)
          b_reg[set][it] = *reinterpret_cast<const double2*>(Bb + (b_off0 + (off_t)it * b_step));
      return;
    }
    // general path: element-wise, predicated (edges, odd leading dimensions, unaligned views)
    if constexpr (GEN == 0) {
#pragma unroll
    for (int it = 0; it < A_PAIRS; ++it) {
      const char* p = Ab + (a_off0 + (off_t)it * a_step);
      const int64_t r = TRANS_A ? r0 + a_r + RSTEP * it : r0 + a_r;
      const int64_t k = TRANS_A ? k0 + a_k : k0 + a_k + KSTEP_NN * it;
      const bool ok0 = (r < M && k < kend);
      const bool ok1 = TRANS_A ? (r < M && k + 1 < kend) : (r + 1 < M && k < kend);
      a_reg[set][it].x = ok0 ? *reinterpret_cast<const double*>(p) : 0.0;
      a_reg[set][it].y = ok1 ? *reinterpret_cast<const double*>(p + 8) : 0.0;
    }
    }
#pragma unroll
    for (int it = 0; it < B_PAIRS; ++it) {
      const char* p = Bb + (b_off0 + (off_t)it * b_step);
      const int cl = b_c + RSTEP * it;
      const int64_t c = c0 + cl;
      const int64_t k = k0 + b_k;
      const bool okc = (cl < NT * 16) && (c < L);
      b_reg[set][it].x = (okc && k < kend) ? *reinterpret_cast<const double*>(p) : 0.0;
      b_reg[set][it].y = (okc && k + 1 < kend) ? *reinterpret_cast<const double*>(p + 8) : 0.0;
    }
  };

  auto stage = [&](int buf, auto SET, int64_t k0) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value;
    double* a_s = smem + LDS0 + buf * BUF_ELEMS;
    double* b_s = a_s + A_ELEMS;
    (void)k0;
    if constexpr (GEN == 2) {
      const int kw = __builtin_amdgcn_readfirstlane(a_k);                   // wave index: the pair slot's k is wave-uniform
      // ONE code path for interior and edge tiles: a second, select-free copy for interior tiles was measured SLOWER (633 vs
      // 584 ms per product at n = 2e5: the kernel outgrew the 64 KB instruction cache, 9040 instructions)
      const int64_t row_first = gen.roff + r0;                                // uniform
      const int krem = (kend - k0 < BK) ? (int)(kend - k0) : BK;              // reduction indices of this tile that exist
#pragma unroll
      for (int it = 0; it < A_PAIRS; ++it) {
        const int64_t kc = k0 + kw + KSTEP_NN * it;
        const bool okk = kw + KSTEP_NN * it < krem;                           // (32-bit: a scalar compare)
        const int64_t gj = gen.koff + (okk ? kc : kbeg);                      // beyond the range: any valid point (the X rows there are zero)
        // provably uniform index -> scalar loads of the 32-byte record of the column point
        const int64_t gju = ((int64_t)__builtin_amdgcn_readfirstlane((int)(gj >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(gj & 0xffffffff));
        const double qx = A[4 * gju], qy = A[4 * gju + 1];
        double dx = p0x - qx, dy = p0y - qy;
        double s0 = fma(dx, dx, dy * dy);
        dx = p1x - qx; dy = p1y - qy;
        double s1 = fma(dx, dx, dy * dy);
        const int fl = gen_flags(gq.flags);
        if (fl & 1) {
          const double qz = A[4 * gju + 2];
          const double dz0 = p0z - qz, dz1 = p1z - qz;
          s0 = fma(dz0, dz0, s0); s1 = fma(dz1, dz1, s1);
        }
        double2 pr;
        gen_point_pair(gq, fl, s0, s1, gtab, pr.x, pr.y);
        const int rel = (int)(gju - row_first);                               // (point indices are 31-bit)
        if ((unsigned)rel < (unsigned)BMT) {                                  // uniform: the diagonal crosses this slot
          pr.x += (a_r == rel) ? gen.nugget : 0.0;
          pr.y += (a_r + 1 == rel) ? gen.nugget : 0.0;
        }
        *reinterpret_cast<double2*>(a_s + (a_k + KSTEP_NN * it) * BMP + a_r) = pr;
      }
    } else
    if (TRANS_A) {
#pragma unroll
      for (int it = 0; it < A_PAIRS; ++it)
        *reinterpret_cast<double2*>(a_s + (a_r + RSTEP * it) * BKP + a_k) = a_reg[set][it];
    } else {
#pragma unroll
      for (int it = 0; it < A_PAIRS; ++it)
        *reinterpret_cast<double2*>(a_s + (a_k + KSTEP_NN * it) * BMP + a_r) = a_reg[set][it];
    }
#pragma unroll
    for (int it = 0; it < B_PAIRS; ++it)
      if (!B_RAGGED || b_c + RSTEP * it < NT * 16)
        *reinterpret_cast<double2*>(b_s + (b_c + RSTEP * it) * BKP + b_k) = b_reg[set][it];
  };

  // Fragments of one k4 step: 2 A fragments (rows 32w + jl, 32w + 16 + jl) and NT B fragments.
  // The B fragments are reloaded in place for the NEXT k4 step right after the two MFMAs that
  // consume them have issued (rolling single buffer), the A fragments one step ahead into a second
  // pair: every LDS read has a full k4 step (1280 MFMA cycles at NT = 10) to land.
  double fa[MT], fan[MT];
  double fb[NTW];
  const int t0 = ch * NTW;                       // first 16-column tile of this wave
  const int ntw = (NT - t0 < NTW) ? ((NT - t0 > 0) ? NT - t0 : 0) : NTW;   // tiles this wave owns (wave-uniform)
  auto a_frag = [&](int buf, int s, int h) -> double {
    const double* a_s = smem + LDS0 + buf * BUF_ELEMS;
    return TRANS_A ? a_s[(16 * MT * rg + 16 * h + jl) * BKP + 4 * s + kk]
                   : a_s[(4 * s + kk) * BMP + 16 * MT * rg + 16 * h + jl];
  };
  auto b_frag = [&](int buf, int s, int t) -> double {
    const double* b_s = smem + LDS0 + buf * BUF_ELEMS + A_ELEMS;
    return b_s[(16 * (t0 + t) + jl) * BKP + 4 * s + kk];
  };

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  // one tile of the pipeline; PAR = parity of t (compile time: selects LDS buffer and register set)
#if GSI_GEMM_TILE_COUNTERS_32
  auto do_tile = [&](int t, auto PAR) __attribute__((always_inline)) {
#else
  auto do_tile = [&](int64_t t, auto PAR) __attribute__((always_inline)) {
#endif
    constexpr int cur = decltype(PAR)::value;
    using NextSet = std::integral_constant<int, (NSETS == 2) ? (cur ^ 1) : 0>;
    // The two waves of a SIMD (column halves ch = 0 / 1 of the same rows) run the same program;
    // their LDS-write / VMEM chores are staggered by half a tile so that one partner is always in
    // a pure MFMA stretch (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  The chores run at priority 0 and
    // everything else at priority 1, so the SIMD's arbiter always prefers the partner that is feeding the matrix
    // pipe (measured +1.5..2 %; a static priority for waves 4-7 alone measured -1 %).
    auto chores = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_s_setprio(0);
#if GSI_GEMM_TILE_COUNTERS_32
      if (t + 1 < ntiles) stage(cur ^ 1, NextSet{}, kbeg + (int64_t)(t + 1) * BK);   // tile t+1: registers -> other LDS buffer
      if (t + 1 + NSETS < ntiles) prefetch(t + 1 + NSETS, NextSet{});                // HBM -> the set just drained
#else
      if (t + 1 < ntiles) stage(cur ^ 1, NextSet{}, kbeg + (t + 1) * BK);            // tile t+1: registers -> other LDS buffer
      if (t + 1 + NSETS < ntiles) prefetch(kbeg + (t + 1 + NSETS) * BK, NextSet{});  // HBM -> the set just drained
#endif
      __builtin_amdgcn_s_setprio(1);
    };
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      if (s == 1 && ch == 0) chores();
      if (s == 1 + BK / 8 && ch != 0) chores();
      const bool last = (s + 1 == BK / 4);
      // the one barrier per tile: tile t+1 becomes visible, and after this step's MFMAs nobody
      // reads buffer `cur` any more (its last fragments are already in registers)
      if (last) __syncthreads();
      const int nbuf = last ? (cur ^ 1) : cur;
      const int ns = last ? 0 : s + 1;
      const bool more = !last || (t + 1 < ntiles);
      if (more) {
#pragma unroll
        for (int h = 0; h < MT; ++h) fan[h] = a_frag(nbuf, ns, h);
      }
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        if ((NT % 2 == 0) || tt < ntw) {
#pragma unroll
          for (int h = 0; h < MT; ++h)
            acc[h][tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[tt], fa[h], acc[h][tt], 0, 0, 0);
          if (more) fb[tt] = b_frag(nbuf, ns, tt);
        }
      }
#pragma unroll
      for (int h = 0; h < MT; ++h) fa[h] = fan[h];
    }
  };

  int64_t item = blockIdx.x;
  bool primed = false;                 // the first two operand tiles of this item were requested at the end of the previous one
  for (;;) {
    if (ntiles > 0) {
      if (!primed) {
#if GSI_GEMM_TILE_COUNTERS_32
        prefetch(0, Set0{});
        if (NSETS == 2 && ntiles > 1) prefetch(1, Set1{});
#else
        prefetch(kbeg, Set0{});
        if (NSETS == 2 && ntiles > 1) prefetch(kbeg + BK, Set1{});
#endif
      }
      stage(0, Set0{}, kbeg);
#if GSI_GEMM_TILE_COUNTERS_32
      if (ntiles > NSETS) prefetch(NSETS, Set0{});
#else
      if (ntiles > NSETS) prefetch(kbeg + NSETS * BK, Set0{});
#endif
      __syncthreads();
#pragma unroll
      for (int h = 0; h < MT; ++h) fa[h] = a_frag(0, 0, h);
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if ((NT % 2 == 0) || t < ntw) fb[t] = b_frag(0, 0, t);
#if GSI_GEMM_TILE_COUNTERS_32
      int t = 0;
#else
      int64_t t = 0;
#endif
      for (; t + 1 < ntiles; t += 2) {
        do_tile(t, Set0{});
        do_tile(t + 1, Set1{});
      }
      if (t < ntiles) do_tile(t, Set0{});
    }
    const int64_t er0 = r0, ec0 = c0;                 // where this item's results go
    bool more = false;
    if constexpr (CANP) {
      if (persist && item + (int64_t)gridDim.x < nitems) {
        more = true;
        item += gridDim.x;
        __syncthreads();                              // every wave is done with this item's LDS images
        locate(item);
        setup_item();
        if (ntiles > 0) {                             // (K, hence ntiles, is the same for every item of this mode)
#if GSI_GEMM_TILE_COUNTERS_32
          prefetch(0, Set0{});
          if (NSETS == 2 && ntiles > 1) prefetch(1, Set1{});
#else
          prefetch(kbeg, Set0{});
          if (NSETS == 2 && ntiles > 1) prefetch(kbeg + BK, Set1{});
#endif
        }
        primed = true;
      }
    }

    // epilogue: lane holds D[i = kk + 4*reg][j = jl]  ->  C[row][col = c0 + 16*(t0+t) + kk + 4*reg]
#pragma unroll
    for (int h = 0; h < MT; ++h) {
      const int64_t row = er0 + 16 * MT * rg + 16 * h + jl;
      if (row < M) {
        double* W = (slabs != nullptr) ? slabs + (int64_t)split * M * L : nullptr;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if ((NT % 2 == 0) || t < ntw) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int64_t col = ec0 + 16 * (t0 + t) + kk + 4 * reg;
              if (col < L) {
                if (W != nullptr) {
                  W[row + col * M] = acc[h][t][reg];
                } else {
                  double v = alpha * acc[h][t][reg];
                  if (beta != 0.0) v += beta * C[row + col * ldc];
                  C[row + col * ldc] = v;
                }
              }
            }
          }
        }
      }
    }
    if (!more) break;
#pragma unroll
    for (int h = 0; h < MT; ++h)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[h][t] = (double4_t){0.0, 0.0, 0.0, 0.0};
  }
}

template <int NT, bool TRANS_A, int GEN, int XMODE>
static void launch_nt(dim3 grid, hipStream_t st, int64_t M, int64_t L, int64_t K, const double* A,
                      int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, double alpha,
                      double beta, double* slabs, int64_t kchunk, int nchunks_x, int wide, int tri, const GenA& gen, int64_t nitems) {
  constexpr size_t shmem = 2 * ((TRANS_A ? BMT * BKP : BK * BMP) + NT * 16 * BKP) * sizeof(double) + (GEN == 2 ? 64 * sizeof(double) : 0);
  static std::atomic<uint64_t> attr_mask{0};
  if (first_use_on_this_device(attr_mask))
    (void)hipFuncSetAttribute((const void*)gemm_f64_kernel<NT, TRANS_A, GEN, XMODE>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL((gemm_f64_kernel<NT, TRANS_A, GEN, XMODE>), grid, dim3(NTHREADS), shmem, st, M, L, K, A, lda, B, ldb, C,
                     ldc, alpha, beta, slabs, kchunk, nchunks_x, wide, tri, gen, nitems);
}

template <bool TRANS_A, int GEN>
static void launch_dispatch(int nt, dim3 grid, hipStream_t st, int64_t M, int64_t L, int64_t K,
                            const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                            int64_t ldc, double alpha, double beta, double* slabs, int64_t kchunk, int nchunks_x, int wide,
                            int xmode, int tri, const GenA& gen, int64_t nitems) {
#define GSI_CASE(N)                                                                             \
  case N:                                                                                       \
    if (xmode == 2)                                                                               \
      launch_nt<N, TRANS_A, GEN, 2>(grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk, nchunks_x, wide, tri, gen, nitems); \
    else if (xmode == 1)                                                                          \
      launch_nt<N, TRANS_A, GEN, 1>(grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk, nchunks_x, wide, tri, gen, nitems); \
    else                                                                                          \
      launch_nt<N, TRANS_A, GEN, 0>(grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk, nchunks_x, wide, tri, gen, nitems); \
    break;
  switch (nt) {
#ifdef GSI_GEMM_DEV_ONLY_NT10      // developer builds (resource-usage remarks of one instantiation in seconds); never set by build.py
    GSI_CASE(10)
#else
    GSI_CASE(1) GSI_CASE(2) GSI_CASE(3) GSI_CASE(4) GSI_CASE(5)
    GSI_CASE(6) GSI_CASE(7) GSI_CASE(8) GSI_CASE(9) GSI_CASE(10)
#endif
    default: break;
  }
#undef GSI_CASE
}

