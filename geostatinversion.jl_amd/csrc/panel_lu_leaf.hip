// panel_lu_leaf.hip -- the register-resident form of `F = lu(Y); Q = F.L` (RandMatFact.jl:60-61, 68-69, 72-73)
// for panels of up to 4096 rows per CU (n = 10^6 at l = 320 is the case it is built for).  Same pivots as
// LAPACK dgetrf (first maximal |entry| wins), L in pivoted row order, exactly as panel_lu.hip; what changes
// is how often the panel crosses HBM.
//
// Per-column sweeps (panel_lu.hip) read and write the <= 8 live columns of a leaf once PER COLUMN: 88 column
// passes per 8-column leaf, and the blocks in between are brought up to date by K = 8/16/32 products that the big
// contraction kernel runs at 1.3-1.7 TB/s.  Here:
//   * lu_leaf_kernel: ONE persistent launch per 8-column leaf.  Every thread keeps its rows' 8 leaf values in
//     registers for all 8 pivot steps (10^6 x 8 doubles = 64 MB = half of the chip's register file), so a leaf costs
//     one read and one write of its columns.  The launch is also LEFT-LOOKING inside its 64-column block: on load it
//     applies the pending update of the block's earlier columns (reads kp = j0 - jb columns of L, U12 = L11^-1 A12
//     is a kp x 8 solve done redundantly by every workgroup, one wave per leaf column), so no in-block trailing
//     update is ever written back.  Per 64-column block: 8 (0 + 8 + ... + 56) + 128 = 352 column passes.
//     One exchange per pivot step: every workgroup publishes {max |value|, row, that row's 8 values} (workgroup 0
//     also row j's values) as one 256-byte record with write-through (sc1) stores, a drained flag store after them;
//     every workgroup polls all flags, reads all records and reduces them in the same fixed order, so the pivot row's
//     values are known everywhere without touching rows another workgroup owns (MI355X_MICROARCH.md, "Valid forms":
//     sc1 payload -> vmcnt(0) -> sc1 flag, sc1 polls and loads; no fences, no atomics).  Two record sets by parity of
//     the step: a workgroup can publish step s+2 only after it has seen every record of step s+1, which its owner
//     wrote after it had read step s.
//   * lu_u12_kernel + lu_rankk_kernel: the block's trailing update A22 -= L21 (L11^-1 A12) as a streaming kernel:
//     a wave holds 32 rows' K = 64 multipliers as MFMA fragments in registers and walks the trailing columns,
//     C tile in, 16 x K/4 MFMAs, C tile out (8 flop per byte of C: MFMA time = HBM time at K = 64).
// Column passes at l = 320, NB = 64: 5 * 352 + 1536 = 3296 (26 GB at n = 10^6) against ~52 GB before.
#include "hip_common.hpp"
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace gsi { namespace hipk {

namespace {

constexpr int LW = LU2_LEAF;           // leaf width (columns kept in registers)
constexpr int KPMAX = LU2_NB - LW;     // deepest pending update inside a block
constexpr int LSP = KPMAX + 1;         // padded row stride of the L11 image
constexpr int REC = LU2_REC_GRANULES;  // 8-byte granules per published record (512 B): unit u = granules 2u (low half), 2u + 1
constexpr int POLL_LIMIT = 4000000;    // default poll budget (~ seconds): a record that never arrives ends the launch with info = -1

__device__ inline double readlane_d(double x, int srclane) {   // srclane wave-uniform
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}
// idamax over the wave: the largest value (values are >= 0 or the "no candidate" marker -1, never NaN), then the
// SMALLEST row among the lanes that hold it (first maximal entry wins, like LAPACK); "no candidate" rows are -1 =
// 0xFFFFFFFF and lose every tie.  All lanes end with the result.  DPP row shifts / broadcasts (register-file speed:
// the whole reduction is ~40 VALU instructions), not ds_bpermute shuffles -- 12 dependent LDS round trips measured
// 0.66 us per reduction, 2-3 of them on the critical path of every pivot step.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_fmax(double v) {
  // the two halves move as 32-bit integers (the builtin is an integer builtin: a double argument would be VALUE-converted);
  // lanes without a source lane keep the identity -1.0 = 0xbff00000'00000000
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)0xbff00000, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return fmax(v, __hiloint2double(hi, lo));
}
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_umin(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROW_MASK, 0xf, false);
  return o < v ? o : v;
}
__device__ inline void wave_argmax(double& v, int32_t& i) {
  double mx = v;
  mx = dpp_fmax<0x111, 0xf>(mx);   // row_shr:1
  mx = dpp_fmax<0x112, 0xf>(mx);   // row_shr:2
  mx = dpp_fmax<0x114, 0xf>(mx);   // row_shr:4
  mx = dpp_fmax<0x118, 0xf>(mx);   // row_shr:8   -> lane 15 of every row holds the row's maximum
  mx = dpp_fmax<0x142, 0xa>(mx);   // row_bcast:15 into rows 1 and 3
  mx = dpp_fmax<0x143, 0xc>(mx);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
  mx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mx), 63), __builtin_amdgcn_readlane(__double2loint(mx), 63));
  uint32_t key = (v == mx) ? (uint32_t)i : 0xFFFFFFFFu;
  key = dpp_umin<0x111, 0xf>(key);
  key = dpp_umin<0x112, 0xf>(key);
  key = dpp_umin<0x114, 0xf>(key);
  key = dpp_umin<0x118, 0xf>(key);
  key = dpp_umin<0x142, 0xa>(key);
  key = dpp_umin<0x143, 0xc>(key);
  v = mx;
  i = (int32_t)__builtin_amdgcn_readlane((int)key, 63);
}

// the same over entries that sit in lanes 0 .. 7 only (per-wave candidates of a workgroup, <= 8 waves): three
// shifts inside row 0, result read from lane 7
__device__ inline void wave_argmax8(double& v, int32_t& i) {
  double mx = v;
  mx = dpp_fmax<0x111, 0xf>(mx);
  mx = dpp_fmax<0x112, 0xf>(mx);
  mx = dpp_fmax<0x114, 0xf>(mx);
  mx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mx), 7), __builtin_amdgcn_readlane(__double2loint(mx), 7));
  uint32_t key = (v == mx) ? (uint32_t)i : 0xFFFFFFFFu;
  key = dpp_umin<0x111, 0xf>(key);
  key = dpp_umin<0x112, 0xf>(key);
  key = dpp_umin<0x114, 0xf>(key);
  v = mx;
  i = (int32_t)__builtin_amdgcn_readlane((int)key, 7);
}

// a record granule as a poller reads it: agent scope within one GPU, system scope when peers on other GPUs wrote it
template <bool MR>
__device__ inline unsigned long long poll_granule(const unsigned long long* p) {
  if constexpr (MR) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// What a poll that ran out was waiting for: written once per factorization into info[2 .. 5] = {phase, slot, epoch, who}
// (phase 1: U mailbox of rank 0, 2: record heads, 3: the winner's row values, 4: the ranks' result heads (two-hop exchange),
// 5: their row values, 6: row boxes of the interchange kernel; who = rank * 1024 + workgroup).  take_error puts it into
// the message: "never launched" (epoch of a leaf's first step) and "stopped mid-leaf" are different bugs.
__device__ inline void lu_timeout_note(int32_t* info, int phase, int slot, uint32_t epoch, int who) {
  if (atomicCAS(info + 2, 0, phase) == 0) { info[3] = slot; info[4] = (int32_t)epoch; info[5] = who; }
}

#ifdef GSI_LU_TRACE
// debug build only (hipcc -DGSI_LU_TRACE): 100 MHz wall-clock stamps of the phases of every pivot step of the kp = 0
// leaves, for 4 workgroups; dumped by lu2_L to $GSI_LU_TRACE
__device__ unsigned long long g_lu_trace[4 * 8 * 8];
#define LU_STAMP(ph)                                                                                    \
  do {                                                                                                  \
    if (tid == 0 && kp == 0) {                                                                          \
      const int tw = (g == 0) ? 0 : (g == 1) ? 1 : (g == G / 2) ? 2 : (g == G - 1) ? 3 : -1;              \
      if (tw >= 0) g_lu_trace[(tw * 8 + s) * 8 + (ph)] = wall_clock64();                                   \
    }                                                                                                   \
  } while (0)
#else
#define LU_STAMP(ph) do { } while (0)
#endif

}  // namespace

// One leaf [j0, j0 + w) of the block that starts at column jb (kp = j0 - jb columns of the block already factored).
// Grid: G workgroups of BS threads, all resident (G <= number of CUs).  Rows j0 .. j0+7 (the leaf's diagonal block:
// the rows that become pivot rows) are held by threads 0..7 of workgroup 0 in `d`; every other row i >= j0 + 8 by
// thread (g, tid) as row j0 + 8 + (g R + rr) BS + tid, rr < R, in `a` -- those rows are active in every step and
// never final, so the code that touches the 8 x R register-resident values is straight-line: no per-row bookkeeping,
// no control-flow joins (a first version with early exits and per-step conditions made the compiler copy all of them
// at every join: 2x the registers, 900 spills at R = 8).
// Few fat waves on purpose: a pivot step is a chain of short dependent phases, and what it costs is the instruction
// stream per SIMD (a version with 16 waves per CU spent 10 us per step issuing ~1800 instructions per wave).
//
// Exchange format.  A record is 18 "units" (doubles): [0] max |value|, [1] row, [2..10) that row's leaf values,
// [10..18) row j's leaf values (workgroup 0 only).  Every unit travels as two 8-byte granules {32 value bits, 32-bit
// step tag}, each written by ONE sc1 store: a granule is the unit of atomicity, so a reader that sees the tag of this
// step in a granule has the data of this step -- no flag, no drain wait, no second round trip behind a flag.
// Two hops: the LEADER (last workgroup) reads every record, reduces them in a fixed order and publishes ONE result
// record {max, pivot row, its 8 values, row j's 8 values}; every other workgroup polls only that result (36 granules,
// one wave).  (Every workgroup sweeping all 256 records itself pulled 10 MB of write-through lines across the fabric
// per step and measured slower.)
// A leaf narrower than 8 columns (the panel's last) still runs 8 steps; steps s >= w see zero columns and are
// gated: no pivot is recorded, nothing is interchanged, the update multiplies zeros.
// MR (several ranks, one launch per rank): this rank holds rows [gbase, gbase + m) of the mtot-row panel in Y
// (local indices), its workgroups are records [rank * G, (rank + 1) * G) of the exchange, and every record is written
// into EVERY rank's record buffer (peer-mapped memory, system-scope stores; pollers read their own memory only).  Rank 0
// owns the diagonal block.  U12 of the pending update comes ready-made (rank 0 solved it, the host sequenced an
// all-reduce); the interchange of the columns outside the leaf is done after the launch (lus_swaps_*).
// MR, hier (shards of more than 256 / nranks workgroups: weak scaling, 10^6 rows per rank): TWO hops.  A workgroup publishes
// into its OWN rank's buffer only (slots 0 .. grid - 1) and every workgroup reduces its rank's records exactly as the
// single-GPU kernel does; workgroup 0 of each rank then writes the rank's result {max, row, its 8 values, row j's 8 values}
// into every rank's buffer (slots grid + rank), and every workgroup polls those nranks records: one more store latency
// across the fabric per pivot step, and the number of records a workgroup polls stays <= 256.
struct LuMrArgs {
  int rank, nranks;
  int hier;                                  // two-hop exchange for shards too tall for nranks x grid <= 256 records (see below)
  int slots;                                 // record slots of the exchange: nranks * grid, or (hier) grid + nranks
  int32_t gbase, mtot;
  const double* us;                          // kp x LW, [c * LW + k]
  unsigned long long* peer[LU2_MAX_RANKS];   // every rank's record buffer (peer[rank] == recs)
};
// OV: the panel (or, with MR, this rank's shard) is TALLER than the grid's registers hold.  Rows beyond the resident window -- [ovb, m),
// ovb = j0 + 8 + grid * R * BS -- stay in HBM with their STORED leaf values and are evaluated lazily, as the streamed
// leaves (lu3_*) do it: the pending update is applied to them once on the way in (written back), every pivot step
// re-derives their candidates from the stored values and the pivot rows so far (s_u), the leaf's last act turns them into
// multipliers.  An overflow row that wins a pivot step hands its values over through the record like any other row and
// receives the old row j's CURRENT values in exchange -- those are already eliminated through the steps before, which a
// small list (s_lr, s_ll: row, first step still to apply) remembers.  Same operations on the same values in the same
// order as the resident rows see: bit-identical factors.  What it is for: the panels just above 4096 rows per CU, whose
// few overflow rows sit in L2 / Infinity Cache between the steps (1.2e6 rows: 9 instead of 14.7 ms).
template <int BS, int R, bool MR, bool OV = false>
__global__ __launch_bounds__(BS) void lu_leaf_kernel(double* __restrict__ Y, int64_t ld, int32_t m, int32_t l,
                                                     int32_t jb, int32_t j0, int w, unsigned long long* __restrict__ recs,
                                                     uint32_t epoch_base, int32_t* __restrict__ ipiv,
                                                     int32_t* __restrict__ info, int onehop, int poll_limit,
                                                     uint32_t mute_epoch, LuMrArgs mr) {
  constexpr int NW = BS / 64;
  constexpr int LPR = BS / 256;               // leader: consumer lanes per record (G <= 256 records)
  constexpr int GPL = (2 * (2 + LW)) / LPR;   // leader: granules per consumer lane
  static_assert(GPL * LPR == 2 * (2 + LW), "record does not divide over its consumer lanes");
  static_assert(LW == 8, "the step list below is written out for 8-column leaves");
  __shared__ double Ls[KPMAX * LSP];
  __shared__ double Us[KPMAX * LW];
  __shared__ double s_val[NW];
  __shared__ int32_t s_idx[NW];
  __shared__ double s_cand[NW][LW];
  __shared__ double s_oldpub[LW];
  __shared__ double c_val[NW];
  __shared__ int32_t c_idx[NW];
  __shared__ uint32_t c_rowbits[NW][2 * LW];  // the wave-local winner's 8 row values as 16 halves
  __shared__ uint32_t c_oldbits[2 * LW];
  __shared__ int32_t c_slot[NW];
  __shared__ int s_abort;
  __shared__ double s_u[OV ? LW * LW : 1];      // OV: the leaf's pivot rows so far (u_t) and 1 / u_tt
  __shared__ double s_rp[OV ? LW : 1];
  __shared__ int32_t s_lr[OV ? LW : 1];         // OV: overflow rows that hold values eliminated through step s_ll - 1
  __shared__ int32_t s_ll[OV ? LW : 1];
  __shared__ int32_t s_nl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x;
  const int Gl = gridDim.x;                                 // this rank's workgroups
  const bool hier = MR && mr.hier != 0;
  const int G = MR ? mr.slots : Gl;                         // record slots of the exchange (layout of the buffer)
  const int GP = hier ? Gl : G;                             // records a workgroup polls in the (first) hop
  const int gslot = MR ? (hier ? g : mr.rank * Gl + g) : g; // this workgroup's record
  const int32_t gbase = MR ? mr.gbase : 0;                  // global index of local row 0
  const int32_t mtot = MR ? mr.mtot : m;
  const int kp = j0 - jb;
  // a launch that follows a timed-out one (info < 0, same stream) drains without polling: one time-out per factorization
  if (tid == 0) s_abort = (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) ? 1 : 0;

  // addressing: one uniform 64-bit column base (SGPRs) + a 32-bit per-thread byte offset, so no 64-bit per-element
  // address stays live in VGPRs between the load at the top and the store at the bottom (m < 2^28 rows)
  auto colbase = [&](int32_t c) -> char* { return reinterpret_cast<char*>(Y + (int64_t)c * ld); };
  auto elem = [&](char* base, int32_t i) -> double* { return reinterpret_cast<double*>(base + (uint32_t)i * 8u); };
  // regular rows: local row0 + rr BS (global gbase + that); on a rank > 0 every local row is below the diagonal block
  const int32_t row0 = ((gbase == 0) ? j0 + LW : 0) + g * (R * BS) + tid;
  const int32_t grow0 = gbase + row0;
  const bool isdiag = (g == 0 && tid < LW && gbase == 0);  // holds row j0 + tid of the diagonal block in d
  const int32_t drow = j0 + tid;
  double a[R][LW];
  double d[LW];
#pragma unroll
  for (int k = 0; k < LW; ++k) {
    char* cb = colbase(j0 + (k < w ? k : 0));
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int32_t i = row0 + rr * BS;
      a[rr][k] = (i < m && k < w) ? *elem(cb, i) : 0.0;   // rows beyond m, columns beyond w: zeros
    }
    d[k] = (isdiag && drow < m && k < w) ? *elem(cb, drow) : 0.0;
  }

  // ---- pending update of the block's earlier columns: a -= L[:, jb:j0] * (L11^-1 A12) ----------------------
  if (kp > 0) {
    // U12 = L11^-1 A12 lives in rows jb .. j0 of the block: on rank 0 when the panel is sharded.  Rank 0 solves it like
    // the single-rank kernel (every workgroup for itself) and its workgroup 0 pushes the kp x 8 values, tagged with this
    // leaf's first epoch, into the other ranks' U mailboxes; they poll their own memory.
    const bool solve_here = !MR || gbase == 0;
    if (solve_here) {
    {
      // all of a thread's elements of the kp x kp block are requested before the first one is used: as a plain loop this
      // was one HBM / L2 round trip per iteration (up to KPMAX^2 / BS of them) at the head of every leaf
      constexpr int NLS = (KPMAX * KPMAX + BS - 1) / BS;
      double lsv[NLS];
#pragma unroll
      for (int i = 0; i < NLS; ++i) {
        const int e = tid + i * BS;
        const int ec = e < kp * kp ? e : 0;
        const int r = ec % kp, c = ec / kp;
        lsv[i] = Y[(jb + r) + (int64_t)(jb + c) * ld];
      }
#pragma unroll
      for (int i = 0; i < NLS; ++i) {
        const int e = tid + i * BS;
        if (e < kp * kp) Ls[(e % kp) * LSP + e / kp] = lsv[i];
      }
    }
    __syncthreads();
    for (int v = wave; v < LW; v += NW) {          // one wave per leaf column: forward substitution along the lanes
      double x = (lane < kp && v < w) ? Y[(jb + lane) + (int64_t)(j0 + v) * ld] : 0.0;
      for (int cp = 0; cp < kp; ++cp) {
        const double xc = readlane_d(x, __builtin_amdgcn_readfirstlane(cp));
        if (lane > cp && lane < kp) x -= Ls[lane * LSP + cp] * xc;
      }
      if (lane < kp) Us[lane * LW + v] = x;
    }
    __syncthreads();
    }
    if constexpr (MR) {
      const uint32_t utag = epoch_base + 1u;
      const size_t ubox = (size_t)2 * (size_t)G * REC + (size_t)((epoch_base >> 3) & 1u) * (size_t)(2 * KPMAX * LW);
      if (gbase == 0) {
        if (g == 0) {
          for (int e = tid; e < kp * LW; e += BS) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(Us[e]);
            for (int q = 0; q < mr.nranks; ++q) {
              if (q == mr.rank) continue;
              __hip_atomic_store(mr.peer[q] + ubox + 2 * e, ((unsigned long long)utag << 32) | (uint32_t)bits, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_SYSTEM);
              __hip_atomic_store(mr.peer[q] + ubox + 2 * e + 1, ((unsigned long long)utag << 32) | (uint32_t)(bits >> 32),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
        }
      } else {
        __syncthreads();                          // s_abort (set by thread 0 at the top) is read below
        for (int e = tid; e < kp * LW; e += BS) {
          unsigned long long lo = 0, hi = 0;
          int tries = s_abort ? poll_limit : 0;
          for (;;) {
            lo = poll_granule<true>(recs + ubox + 2 * e);
            hi = poll_granule<true>(recs + ubox + 2 * e + 1);
            if ((uint32_t)(lo >> 32) == utag && (uint32_t)(hi >> 32) == utag) break;
            if (++tries > poll_limit) { s_abort = 1; lu_timeout_note(info, 1, e, utag, mr.rank * 1024 + g); break; }
            __builtin_amdgcn_s_sleep(1);
          }
          Us[e] = __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
        }
        __syncthreads();
      }
    }
    for (int c = 0; c < kp; c += 2) {              // kp is a multiple of the leaf width
      double lv[2][R];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        char* cb = colbase(jb + c + cc);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          const int32_t i = row0 + rr * BS;
          lv[cc][rr] = (i < m) ? *elem(cb, i) : 0.0;
        }
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        double u[LW];
#pragma unroll
        for (int k = 0; k < LW; ++k) u[k] = Us[(c + cc) * LW + k];
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
          for (int k = 0; k < LW; ++k) a[rr][k] -= lv[cc][rr] * u[k];
      }
    }
    if (isdiag && drow < m) {                      // (8 threads of the grid) the diagonal block's rows
      for (int c = 0; c < kp; ++c) {
        const double lvd = *elem(colbase(jb + c), drow);
#pragma unroll
        for (int k = 0; k < LW; ++k) d[k] -= lvd * Us[c * LW + k];
      }
    }
  }

  // ---- OV: the rows beyond the resident window ---------------------------------------------------------------
  const int32_t ovb = ((gbase == 0) ? j0 + LW : 0) + Gl * (R * BS);   // first overflow row (LOCAL index, as every i below)
  const int32_t ovstride = Gl * BS;
  const int32_t ov0 = ovb + g * BS + tid;                   // this thread's overflow rows: ov0 + k ovstride < m
  // first step still to apply to overflow row i (0 unless it received an old row j during this leaf)
  auto ov_level = [&](int32_t i) -> int {
    int lev = 0;
    const int nl = s_nl;
    for (int q = 0; q < nl; ++q) if (s_lr[q] == i) lev = s_ll[q];
    return lev;
  };
  if constexpr (OV) {
    if (tid == 0) s_nl = 0;
    if (kp > 0) {                                           // pending update, written back (as lu3_open_kernel)
      for (int32_t i = ov0; i < m; i += ovstride) {
        double x[LW];
#pragma unroll
        for (int k = 0; k < LW; ++k) x[k] = (k < w) ? *elem(colbase(j0 + k), i) : 0.0;
        for (int c = 0; c < kp; c += 4) {
          double lv4[4];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) lv4[cc] = *elem(colbase(jb + c + cc), i);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int k = 0; k < LW; ++k) x[k] -= lv4[cc] * Us[(c + cc) * LW + k];
        }
#pragma unroll
        for (int k = 0; k < LW; ++k) if (k < w) *elem(colbase(j0 + k), i) = x[k];
      }
    }
    __syncthreads();                                        // s_nl
  }

  // Row interchanges of the columns OUTSIDE the leaf (LAPACK swaps whole rows): column c belongs to workgroup
  // c % G, thread c / G.  Pipelined one step behind: the two loads of step s are issued when its pivot is known and
  // stored swapped at step s + 1, so their latency hides behind the next exchange (same thread, program order:
  // a later load of the same element sees the earlier store).
  const int32_t swc = g + Gl * tid;
  const bool has_col = !MR && swc < l && !(swc >= j0 && swc < j0 + w);
  double* const swcol = Y + (int64_t)(has_col ? swc : 0) * ld;
  bool pend = false;
  double pa0 = 0.0, pa1 = 0.0;
  int32_t pj = 0, pr = 0;

  // one pivot step; S = leaf column (compile time: register index)
  auto step = [&](auto S_) {
    constexpr int s = decltype(S_)::value;
    const bool live = s < w;
    const int32_t j = j0 + s;
    const uint32_t epoch = epoch_base + (uint32_t)s + 1u;
    const size_t set_off = (size_t)(epoch & 1u) * (size_t)G * REC;
    unsigned long long* rec_set = recs + set_off;
    LU_STAMP(0);
    // (a) this thread's, this wave's, this workgroup's candidate for column s; lowest rows first, strict >
    double best = -1.0;
    int32_t besti = -1;
    if (isdiag && tid >= s) { best = fabs(d[s]); besti = drow; }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const double av = fabs(a[rr][s]);
      if (av > best && (!MR || row0 + rr * BS < m)) { best = av; besti = grow0 + rr * BS; }   // MR: rows beyond the shard are no candidates
    }
    if constexpr (OV) {                         // overflow rows: current value in column s from the stored values, lazily
      if (live) {
        for (int32_t i = ov0; i < m; i += ovstride) {
          const int lev = ov_level(i);
          double x[s + 1];
#pragma unroll
          for (int k = 0; k <= s; ++k) x[k] = *elem(colbase(j0 + k), i);
#pragma unroll
          for (int t = 0; t < s; ++t) {
            if (t >= lev) {
              const double rp = s_rp[t];
              const double lt = (rp != 0.0) ? x[t] * rp : x[t];
#pragma unroll
              for (int k = t + 1; k <= s; ++k) x[k] -= lt * s_u[t * LW + k];
            }
          }
          const double av = fabs(x[s]);
          if (av > best) { best = av; besti = gbase + i; }
        }
      }
    }
    double wv = best;
    int32_t wi = besti;
    wave_argmax(wv, wi);
    if (lane == 0) { s_val[wave] = wv; s_idx[wave] = wi; }
    if (besti >= 0 && besti == wi) {            // the ONE lane of the wave that owns its candidate row: one copy per
      if (isdiag && besti == drow) {            // wave (~40 instructions) is cheaper than two more barriers
#pragma unroll
        for (int k = 0; k < LW; ++k) s_cand[wave][k] = d[k];
      }
      if constexpr (OV) {
        if (besti - gbase >= ovb) {             // an overflow row: all 8 of its current values
          const int32_t bl = besti - gbase;
          const int lev = ov_level(bl);
          double x[LW];
#pragma unroll
          for (int k = 0; k < LW; ++k) x[k] = (k < w) ? *elem(colbase(j0 + k), bl) : 0.0;
#pragma unroll
          for (int t = 0; t < s; ++t) {
            if (t >= lev) {
              const double rp = s_rp[t];
              const double lt = (rp != 0.0) ? x[t] * rp : x[t];
              x[t] = lt;
#pragma unroll
              for (int k = t + 1; k < LW; ++k) x[k] -= lt * s_u[t * LW + k];
            }
          }
#pragma unroll
          for (int k = 0; k < LW; ++k) s_cand[wave][k] = x[k];
        }
      }
#pragma unroll
      for (int rr = 0; rr < R; ++rr)
        if (grow0 + rr * BS == besti) {
#pragma unroll
          for (int k = 0; k < LW; ++k) s_cand[wave][k] = a[rr][k];
        }
    }
    if (isdiag && tid == s) {                   // row j itself (the row the pivot row will be exchanged with)
#pragma unroll
      for (int k = 0; k < LW; ++k) s_oldpub[k] = d[k];
    }
    __syncthreads();
    LU_STAMP(1);
    // (b) wave 0 reduces the waves' candidates and publishes the workgroup's record: one granule per lane
    if (wave == 0) {
      double pv = (lane < NW) ? s_val[lane] : -1.0;
      int32_t pi = (lane < NW) ? s_idx[lane] : -1;
      const int32_t mywi = pi;
      static_assert(NW <= 8, "wave_argmax8 reduces lanes 0..7");
      wave_argmax8(pv, pi);
      const unsigned long long own = __ballot(lane < NW && pi >= 0 && mywi == pi);
      const int ww = own ? (__ffsll((long long)own) - 1) : 0;
      const int unit = lane >> 1;
      // mute_epoch != 0 (tests only): the last workgroup stays silent at that step, as a workgroup that never got a CU
      // would -- everyone else runs out of polls and the launch ends with info = -1
      const bool muted = (mute_epoch != 0u && epoch == mute_epoch && gslot == G - 1);
      if (!muted && (unit < 2 + LW || (gslot == 0 && unit < 2 + 2 * LW))) {
        unsigned long long bits;
        if (unit == 0) bits = (unsigned long long)__double_as_longlong(pv);
        else if (unit == 1) bits = (unsigned long long)(long long)pi;
        else if (unit < 2 + LW) bits = (unsigned long long)__double_as_longlong(s_cand[ww][unit - 2]);
        else bits = (unsigned long long)__double_as_longlong(s_oldpub[unit - 2 - LW]);
        const uint32_t half = (lane & 1) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
        if constexpr (MR) {
          if (hier) {                              // first hop stays on this rank
            __hip_atomic_store(rec_set + (size_t)gslot * REC + lane, ((unsigned long long)epoch << 32) | half, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
          } else {
            for (int q = 0; q < mr.nranks; ++q)    // one copy into every rank's buffer: remote stores, local polls
              __hip_atomic_store(mr.peer[q] + set_off + (size_t)gslot * REC + lane, ((unsigned long long)epoch << 32) | half,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        } else {
          __hip_atomic_store(rec_set + (size_t)g * REC + lane, ((unsigned long long)epoch << 32) | half,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // (c) the exchange.  Two protocols (GSI_LU_ONEHOP selects):
    //  one hop (default): every workgroup reads the (value, row) head of every record itself (3 granules each: 6 KB per
    //            workgroup and round), reduces, then fetches the winner's row values (landed long before) and row j;
    //            measured 5.6-6.5 us per step, LU 7.35 ms at n = 1e6, l = 320;
    //  two hops: a leader sweeps all WHOLE records, reduces, publishes the result; everyone else polls the result:
    //            6.5-7.5 us per step (7.7 ms).  (Every workgroup sweeping all whole records -- 40 KB each -- 14.7 us.)
    const bool leader = (g == G - 1) && !onehop;           // (MR launches always use the one-hop protocol)
    // the result is published in LU2_RES_COPIES copies on lines of their own; workgroup g polls copy g % LU2_RES_COPIES
    // (255 workgroups polling the same three lines serialise on one memory channel)
    unsigned long long* res = recs + (size_t)2 * (size_t)G * REC + (size_t)(epoch & 1u) * (size_t)LU2_RES_COPIES * REC;
    if (onehop) {
      const int ncw1 = (GP + 63) / 64;                                     // one lane per record
      if (wave < ncw1) {
        const bool mine = tid < GP;
        const unsigned long long* src = rec_set + (size_t)(mine ? tid : 0) * REC;
        const bool extra = tid < 2 * LW;                                    // workgroup 0's copy of row j
        const unsigned long long* xsrc = rec_set + 2 * (2 + LW) + (extra ? tid : 0);
        unsigned long long g0 = 0, g1 = 0, g2 = 0, gx = 0;
        int tries = s_abort ? poll_limit : 0;
        bool ok;
        for (;;) {
          if (mine) {
            g0 = poll_granule<MR>(src + 0);
            g1 = poll_granule<MR>(src + 1);
            g2 = poll_granule<MR>(src + 2);
          }
          if (extra) gx = poll_granule<MR>(xsrc);
          ok = !mine || ((uint32_t)(g0 >> 32) == epoch && (uint32_t)(g1 >> 32) == epoch && (uint32_t)(g2 >> 32) == epoch);
          if (extra) ok = ok && ((uint32_t)(gx >> 32) == epoch);
          if (__all(ok)) break;
          if (++tries > poll_limit) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!__all(ok)) { s_abort = 1; if (!ok) lu_timeout_note(info, 2, tid, epoch, (MR ? mr.rank : 0) * 1024 + g); }
        LU_STAMP(2);
        if (extra) c_oldbits[tid] = (uint32_t)gx;
        double cv = -1.0;
        int32_t ci = -1;
        if (mine) {
          cv = __longlong_as_double((long long)(((unsigned long long)(uint32_t)g1 << 32) | (uint32_t)g0));
          ci = (int32_t)(uint32_t)g2;
        }
        double rv = cv;
        int32_t ri = ci;
        wave_argmax(rv, ri);
        if (mine && ri >= 0 && ci == ri) c_slot[wave] = tid;              // the record that holds the wave's winner
        if (lane == 0) { c_val[wave] = rv; c_idx[wave] = ri; }
      }
      __syncthreads();
      if (wave == 0) {
        double fv = (lane < ncw1) ? c_val[lane] : -1.0;
        int32_t fi = (lane < ncw1) ? c_idx[lane] : -1;
        const int32_t myfi = fi;
        wave_argmax8(fv, fi);
        const unsigned long long own = __ballot(lane < ncw1 && fi >= 0 && myfi == fi);
        const int fw = own ? (__ffsll((long long)own) - 1) : 0;
        const int gw = (fi >= 0) ? c_slot[fw] : 0;
        // the winner's 8 row values: granules 4 .. 20 of its record, one per lane (they were stored with the head)
        const bool mine = lane < 2 * LW;
        const unsigned long long* rsrc = rec_set + (size_t)gw * REC + 4 + (mine ? lane : 0);
        unsigned long long gv = 0;
        int tries = s_abort ? poll_limit : 0;
        bool ok;
        for (;;) {
          if (mine) gv = poll_granule<MR>(rsrc);
          ok = !mine || ((uint32_t)(gv >> 32) == epoch);
          if (__all(ok)) break;
          if (++tries > poll_limit) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!__all(ok)) { s_abort = 1; if (!ok) lu_timeout_note(info, 3, gw, epoch, (MR ? mr.rank : 0) * 1024 + g); }
        LU_STAMP(3);
        if (mine) c_rowbits[0][lane] = (uint32_t)gv;
        if (lane == 0) { c_val[0] = fv; c_idx[0] = fi; }
      }
      __syncthreads();
      if constexpr (MR) {
        if (hier) {                               // second hop: the ranks' results (slots Gl .. Gl + nranks - 1 of every buffer)
          if (wave == 0) {
            if (g == 0) {                         // this rank's result, one granule per lane, into every rank's buffer
              const int unit = lane >> 1;
              if (unit < 2 + 2 * LW) {
                unsigned long long bits;
                if (unit == 0) bits = (unsigned long long)__double_as_longlong(c_val[0]);
                else if (unit == 1) bits = (unsigned long long)(long long)c_idx[0];
                else if (unit < 2 + LW) bits = reinterpret_cast<const unsigned long long*>(c_rowbits[0])[unit - 2];
                else bits = reinterpret_cast<const unsigned long long*>(c_oldbits)[unit - 2 - LW];
                const uint32_t half = (lane & 1) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
                for (int q = 0; q < mr.nranks; ++q)
                  __hip_atomic_store(mr.peer[q] + set_off + (size_t)(Gl + mr.rank) * REC + lane, ((unsigned long long)epoch << 32) | half,
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              }
            }
            const bool rmine = lane < mr.nranks;
            const unsigned long long* src = rec_set + (size_t)(Gl + (rmine ? lane : 0)) * REC;
            unsigned long long g0 = 0, g1 = 0, g2 = 0;
            int tries = s_abort ? poll_limit : 0;
            bool ok;
            for (;;) {
              if (rmine) {
                g0 = poll_granule<true>(src + 0);
                g1 = poll_granule<true>(src + 1);
                g2 = poll_granule<true>(src + 2);
              }
              ok = !rmine || ((uint32_t)(g0 >> 32) == epoch && (uint32_t)(g1 >> 32) == epoch && (uint32_t)(g2 >> 32) == epoch);
              if (__all(ok)) break;
              if (++tries > poll_limit) break;
              __builtin_amdgcn_s_sleep(1);
            }
            if (!__all(ok)) { s_abort = 1; if (!ok) lu_timeout_note(info, 4, lane, epoch, mr.rank * 1024 + g); }
            double cv = -1.0;
            int32_t ci = -1;
            if (rmine) {
              cv = __longlong_as_double((long long)(((unsigned long long)(uint32_t)g1 << 32) | (uint32_t)g0));
              ci = (int32_t)(uint32_t)g2;
            }
            double rv = cv;
            int32_t ri = ci;
            wave_argmax(rv, ri);
            const unsigned long long own = __ballot(rmine && ri >= 0 && ci == ri);
            const int qw = own ? (__ffsll((long long)own) - 1) : 0;
            // the winner's 8 row values (its rank's record) and row j's (rank 0's record): 16 granules each, one per lane
            const bool vmine = lane < 2 * LW, omine = lane >= 32 && lane < 32 + 2 * LW;
            const unsigned long long* vsrc = rec_set + (size_t)(Gl + qw) * REC + 4 + (vmine ? lane : 0);
            const unsigned long long* osrc = rec_set + (size_t)Gl * REC + 2 * (2 + LW) + (omine ? lane - 32 : 0);
            unsigned long long gv = 0;
            tries = s_abort ? poll_limit : 0;
            for (;;) {
              if (vmine) gv = poll_granule<true>(vsrc);
              if (omine) gv = poll_granule<true>(osrc);
              ok = !(vmine || omine) || ((uint32_t)(gv >> 32) == epoch);
              if (__all(ok)) break;
              if (++tries > poll_limit) break;
              __builtin_amdgcn_s_sleep(1);
            }
            if (!__all(ok)) { s_abort = 1; if (!ok) lu_timeout_note(info, 5, qw, epoch, mr.rank * 1024 + g); }
            if (vmine) c_rowbits[0][lane] = (uint32_t)gv;
            if (omine) c_oldbits[lane - 32] = (uint32_t)gv;
            if (lane == 0) { c_val[0] = rv; c_idx[0] = ri; }
          }
          __syncthreads();
        }
      }
    } else
    {
    const int ncw = leader ? (G * LPR + 63) / 64 : 1;                     // waves that hold entries of the reduction
    if (leader) {
      const int slot = tid / LPR, part = tid % LPR;
      const bool mine = slot < G;
      const unsigned long long* src = rec_set + (size_t)(mine ? slot : 0) * REC + part * GPL;
      const bool extra = tid < 2 * LW;                                    // workgroup 0's copy of row j
      const unsigned long long* xsrc = rec_set + 2 * (2 + LW) + (extra ? tid : 0);
      if (wave < ncw) {
        unsigned long long gl[GPL], gx = 0;
        int tries = s_abort ? poll_limit : 0;   // a timed-out launch drains without polling again
        bool ok;
        for (;;) {
#pragma unroll
          for (int i = 0; i < GPL; ++i) gl[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (extra) gx = __hip_atomic_load(xsrc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = true;
          if (mine) {
#pragma unroll
            for (int i = 0; i < GPL; ++i) ok = ok && ((uint32_t)(gl[i] >> 32) == epoch);
          }
          if (extra) ok = ok && ((uint32_t)(gx >> 32) == epoch);
          if (__all(ok)) break;
          if (++tries > poll_limit) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!__all(ok)) s_abort = 1;
        LU_STAMP(2);
        if (extra) c_oldbits[tid] = (uint32_t)gx;
        // candidate (value, row) sits in the record's first four granules = part 0's gl[0..4)
        double cv = -1.0;
        int32_t ci = -1;
        if (mine && part == 0) {
          cv = __longlong_as_double((long long)(((unsigned long long)(uint32_t)gl[1] << 32) | (uint32_t)gl[0]));
          ci = (int32_t)(uint32_t)gl[2];
        }
        double rv = cv;
        int32_t ri = ci;
        wave_argmax(rv, ri);
        // the lanes of the wave-local winner's record drop its row values (granules 4..20) into LDS
        const int32_t myi = __shfl(ci, lane - part);       // the record's row, known to all of its lanes
        if (mine && ri >= 0 && myi == ri) {
#pragma unroll
          for (int i = 0; i < GPL; ++i) {
            const int gi = part * GPL + i;
            if (gi >= 4) c_rowbits[wave][gi - 4] = (uint32_t)gl[i];
          }
        }
        if (lane == 0) { c_val[wave] = rv; c_idx[wave] = ri; }
      }
      __syncthreads();
      if (wave == 0) {                          // finish the reduction and publish the result: one granule per lane
        double fv = (lane < ncw) ? c_val[lane] : -1.0;
        int32_t fi = (lane < ncw) ? c_idx[lane] : -1;
        const int32_t myfi = fi;
        wave_argmax8(fv, fi);
        const unsigned long long own = __ballot(lane < ncw && fi >= 0 && myfi == fi);
        const int fw = own ? (__ffsll((long long)own) - 1) : 0;
        const int unit = lane >> 1;
        unsigned long long bits = 0;
        if (unit < 2 + 2 * LW) {
          if (unit == 0) bits = (unsigned long long)__double_as_longlong(fv);
          else if (unit == 1) bits = (unsigned long long)(long long)fi;
          else if (unit < 2 + LW) bits = reinterpret_cast<const unsigned long long*>(c_rowbits[fw])[unit - 2];
          else bits = reinterpret_cast<const unsigned long long*>(c_oldbits)[unit - 2 - LW];
          const uint32_t half = (lane & 1) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
#pragma unroll
          for (int cpy = 0; cpy < LU2_RES_COPIES; ++cpy)
            __hip_atomic_store(res + (size_t)cpy * REC + lane, ((unsigned long long)epoch << 32) | half, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        LU_STAMP(3);
        // the leader's own threads read the result from slot 0, like everybody else
        if (lane == 0) { c_val[0] = fv; c_idx[0] = fi; }
        if (unit >= 2 && unit < 2 + LW && (lane & 1) == 0) reinterpret_cast<unsigned long long*>(c_rowbits[0])[unit - 2] = bits;
      }
      __syncthreads();
    } else {
      if (wave == 0) {                          // poll the leader's result: granule `lane`
        const bool mine = lane < 2 * (2 + 2 * LW);
        unsigned long long gv = 0;
        int tries = s_abort ? poll_limit : 0;
        bool ok;
        for (;;) {
          if (mine) gv = __hip_atomic_load(res + (size_t)(g % LU2_RES_COPIES) * REC + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = !mine || ((uint32_t)(gv >> 32) == epoch);
          if (__all(ok)) break;
          if (++tries > poll_limit) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!__all(ok)) s_abort = 1;
        LU_STAMP(3);
        // lay the result out exactly as the leader's own LDS image: c_val / c_idx [0], c_rowbits[0], c_oldbits
        if (lane < 2) reinterpret_cast<uint32_t*>(&c_val[0])[lane] = (uint32_t)gv;
        else if (lane == 2) c_idx[0] = (int32_t)(uint32_t)gv;
        else if (lane >= 4 && lane < 4 + 2 * LW) c_rowbits[0][lane - 4] = (uint32_t)gv;
        else if (lane >= 4 + 2 * LW && lane < 4 + 4 * LW) c_oldbits[lane - 4 - 2 * LW] = (uint32_t)gv;
      }
      __syncthreads();
    }
    }
    LU_STAMP(4);
    // (d) the result: slot 0 of the LDS image
    const double bestv = c_val[0];
    int32_t r = c_idx[0];
    const bool valid = live && (r >= j && r < mtot);
    if (!valid) r = j;                           // all-NaN column (or a gated step): no interchange
    const double* c_old = reinterpret_cast<const double*>(c_oldbits);
    const double* c_row = reinterpret_cast<const double*>(c_rowbits[0]);
    double u[LW];
#pragma unroll
    for (int k = 0; k < LW; ++k) u[k] = valid ? c_row[k] : c_old[k];
    const double piv = u[s];
    const double rpiv = (piv != 0.0) ? 1.0 / piv : 0.0;
    if constexpr (OV) {
      if (live) {
        if (tid == 0) {                         // (unrolled: u[] indexed by the thread id would live in scratch)
#pragma unroll
          for (int k = 0; k < LW; ++k) s_u[s * LW + k] = u[k];
          s_rp[s] = rpiv;
        }
        const int32_t rl = r - gbase;           // (local index; on another rank's rows this is out of [ovb, m))
        if (rl >= ovb && rl < m && r != j) {    // the pivot row was one of MY overflow rows: the old row j moves there, already
          if (tid == 2 * LW) { const int q = s_nl; s_lr[q] = rl; s_ll[q] = s; s_nl = q + 1; }    // eliminated through step s - 1
          if ((rl - ovb) % ovstride == g * BS + tid) {
#pragma unroll
            for (int k = 0; k < LW; ++k) if (k < w) *elem(colbase(j0 + k), rl) = c_old[k];
          }
        }
      }
      __syncthreads();                          // s_u / the list before the next step's lazy evaluations
    }
    // (e) bookkeeping by workgroup 0; pipelined interchange of this thread's column outside the leaf
    if (g == 0 && tid == 0 && live) {
      ipiv[j] = r;
      if (!(bestv > 0.0)) atomicCAS(info, 0, j + 1);
    }
    if (has_col) {
      if (pend) { swcol[pj] = pa1; swcol[pr] = pa0; }
      pend = (r != j);
      if (pend) { pa0 = swcol[j]; pa1 = swcol[r]; pj = j; pr = r; }
    }
    // (f) rank-1 update in registers: regular rows are all active, none is row j
    bool hit = false;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) hit = hit || (grow0 + rr * BS == r);
    if (hit) {                                   // (one thread of the whole grid) the old row j moves here
#pragma unroll
      for (int rr = 0; rr < R; ++rr)
        if (grow0 + rr * BS == r) {
#pragma unroll
          for (int k = 0; k < LW; ++k) a[rr][k] = c_old[k];
        }
    }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const double x0 = a[rr][s];
      const double lij = (rpiv != 0.0) ? x0 * rpiv : x0;
      a[rr][s] = lij;
#pragma unroll
      for (int k = s + 1; k < LW; ++k) a[rr][k] -= lij * u[k];
    }
    if (isdiag && tid >= s) {                    // the diagonal block's rows (8 threads of the grid)
      if (tid == s) {                            // this position receives the pivot row and is final
#pragma unroll
        for (int k = 0; k < LW; ++k) d[k] = u[k];
      } else {
        if (drow == r) {                         // the old row j moves here
#pragma unroll
          for (int k = 0; k < LW; ++k) d[k] = c_old[k];
        }
        const double x0 = d[s];
        const double lij = (rpiv != 0.0) ? x0 * rpiv : x0;
        d[s] = lij;
#pragma unroll
        for (int k = s + 1; k < LW; ++k) d[k] -= lij * u[k];
      }
    }
    LU_STAMP(5);
  };

  // straight-line step list
  {
    using std::integral_constant;
    step(integral_constant<int, 0>{});
    step(integral_constant<int, 1>{});
    step(integral_constant<int, 2>{});
    step(integral_constant<int, 3>{});
    step(integral_constant<int, 4>{});
    step(integral_constant<int, 5>{});
    step(integral_constant<int, 6>{});
    step(integral_constant<int, 7>{});
  }
  if constexpr (OV) {                           // the overflow rows' multipliers
    for (int32_t i = ov0; i < m; i += ovstride) {
      const int lev = ov_level(i);
      double x[LW];
#pragma unroll
      for (int k = 0; k < LW; ++k) x[k] = (k < w) ? *elem(colbase(j0 + k), i) : 0.0;
#pragma unroll
      for (int t = 0; t < LW; ++t) {
        if (t >= lev && t < w) {
          const double rp = s_rp[t];
          const double lt = (rp != 0.0) ? x[t] * rp : x[t];
          x[t] = lt;
#pragma unroll
          for (int k = t + 1; k < LW; ++k) x[k] -= lt * s_u[t * LW + k];
        }
      }
#pragma unroll
      for (int k = 0; k < LW; ++k) if (k < w) *elem(colbase(j0 + k), i) = x[k];
    }
  }
  if (s_abort && tid == 0) atomicExch(info, -1);
  if (has_col && pend) { swcol[pj] = pa1; swcol[pr] = pa0; }
#pragma unroll
  for (int k = 0; k < LW; ++k) {
    if (k < w) {
      char* cb = colbase(j0 + k);
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const int32_t i = row0 + rr * BS;
        if (i < m) *elem(cb, i) = a[rr][k];
      }
      if (isdiag && drow < m) *elem(cb, drow) = d[k];
    }
  }
}

// U12 = L11^-1 A12 for the K x K unit-lower block at (jb, jb) and the columns [c0, c1): out[k + (c - c0) K].
// Thread = one column; L11 in LDS (broadcast reads), the column in registers.
// (256 threads bring L11 in -- K^2 / 256 loads each instead of K^2 / 64: the kernel sits between two blocks of the
// factorization and is all latency -- then the first wave solves its 64 columns.)
template <int K>
__global__ __launch_bounds__(256) void lu_u12_kernel(const double* __restrict__ Y, int64_t ld, int64_t jb, int64_t jbrow,
                                                     int64_t c0, int64_t c1, double* __restrict__ out) {
  // jb: the block's first COLUMN (global); jbrow: the row of Y that holds global row jb (jb - row0 for a row shard)
  __shared__ double L11[K * K];
#pragma unroll 8
  for (int e = threadIdx.x; e < K * K; e += 256) {
    const int r = e % K, c = e / K;
    L11[r * K + c] = Y[(jbrow + r) + (jb + c) * ld];    // [row][col]: a row's multipliers are contiguous
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int64_t c = c0 + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= c1) return;
  double x[K];
  const double* col = Y + jbrow + c * ld;
#pragma unroll
  for (int r = 0; r < K; ++r) x[r] = col[r];
#pragma unroll
  for (int r = 1; r < K; ++r) {
    double v = x[r];
#pragma unroll
    for (int p = 0; p < r; ++p) v -= L11[r * K + p] * x[p];
    x[r] = v;
  }
  double* o = out + (c - c0) * K;
#pragma unroll
  for (int r = 0; r < K; ++r) o[r] = x[r];
}

// A22 -= L21 * U12: rows [r_begin, m), columns [c0, c0 + t), L21 = Y[:, jb:jb+K], U12 (K x t, ld K) from lu_u12_kernel.
// Workgroup = 4 waves x 32 rows; column chunk of <= RK_CHUNK columns per blockIdx.y (its U12 slice sits in LDS).
// MFMA operands swapped like the big contraction kernel: lane (jl = lane & 15, kk = lane >> 4) holds, for C row
// jl (+16 h), the columns kk + 4 reg of a 16-column tile.
template <int K, int DEPTH, int RK_CHUNK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RK_CHUNK == 64 ? 3 : 2, RK_CHUNK == 64 ? 3 : 4))) void lu_rankk_kernel(double* __restrict__ Y, int64_t ld, int64_t m, int64_t r_begin,
                                                       int64_t jb, int64_t c0, int64_t t,
                                                       const double* __restrict__ U12) {
  typedef double double4_t __attribute__((ext_vector_type(4)));
  constexpr int KP = K + 2;                     // padded k stride of the U image [col][k] (KP / 2 odd: conflict-free b64 reads)
  extern __shared__ double us[];                // RK_CHUNK * KP doubles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & 15, kk = lane >> 4;
  const int64_t rb = r_begin + ((int64_t)blockIdx.x * 4 + wave) * 32;
  // this wave's 32 rows of multipliers as MFMA fragments: fa[h][s] = L[rb + 16 h + jl, jb + 4 s + kk]
  double fa[2][K / 4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t rrow = rb + 16 * h + jl;
#pragma unroll
    for (int s = 0; s < K / 4; ++s) fa[h][s] = (rrow < m) ? Y[rrow + (jb + 4 * s + kk) * ld] : 0.0;
  }
  for (int64_t cc0 = 0; cc0 < t; cc0 += RK_CHUNK) {   // the workgroup walks ALL trailing columns: L21 is read once
    const int tc = (int)((t - cc0 < RK_CHUNK) ? (t - cc0) : RK_CHUNK);
    __syncthreads();                                  // the previous chunk's U image is no longer read
    for (int e = tid; e < RK_CHUNK * K; e += 256) {
      const int k = e % K, c = e / K;
      us[c * KP + k] = (c < tc) ? U12[k + (cc0 + c) * K] : 0.0;
    }
    __syncthreads();
    if (rb >= m) continue;
    const int ntile = (tc + 15) / 16;
    double cin[2][4];
    auto load_tile = [&](int tt, double (&dst)[2][4]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t rrow = rb + 16 * h + jl;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int cl = 16 * tt + kk + 4 * reg;
          dst[h][reg] = (rrow < m && cl < tc) ? Y[rrow + (c0 + cc0 + cl) * ld] : 0.0;
        }
      }
    };
    load_tile(0, cin);
    double cin2[2][4];                                // DEPTH == 2: two tiles of C in flight
    if (DEPTH == 2 && ntile > 1) load_tile(1, cin2);
    for (int tt = 0; tt < ntile; ++tt) {
      double4_t acc[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[h] = (double4_t){cin[h][0], cin[h][1], cin[h][2], cin[h][3]};
      if (DEPTH == 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) cin[h][reg] = cin2[h][reg];
        if (tt + 2 < ntile) load_tile(tt + 2, cin2);
      } else if (tt + 1 < ntile) load_tile(tt + 1, cin);     // next tile's C in flight behind this tile's MFMAs
#pragma unroll
      for (int s = 0; s < K / 4; ++s) {
        const double fb = -us[(16 * tt + jl) * KP + 4 * s + kk];
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb, fa[h][s], acc[h], 0, 0, 0);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t rrow = rb + 16 * h + jl;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int cl = 16 * tt + kk + 4 * reg;
          if (rrow < m && cl < tc) Y[rrow + (c0 + cc0 + cl) * ld] = acc[h][reg];
        }
      }
    }
  }
}

template <int K, int DEPTH, int CHUNK>
static void launch_rankk_v(hipStream_t st, unsigned grid, double* Y, int64_t ld, int64_t m, int64_t r_begin, int64_t jb,
                           int64_t c0, int64_t t, const double* U12) {
  constexpr size_t shmem = (size_t)CHUNK * (K + 2) * sizeof(double);
  static std::atomic<uint64_t> attr_mask{0};
  if (first_use_on_this_device(attr_mask))
    (void)hipFuncSetAttribute((const void*)lu_rankk_kernel<K, DEPTH, CHUNK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL((lu_rankk_kernel<K, DEPTH, CHUNK>), dim3(grid), dim3(256), shmem, st, Y, ld, m, r_begin, jb, c0, t, U12);
}
// U12 columns staged per pass (its LDS image bounds the workgroups per CU: 128 columns = 67 KB = 2 workgroups, 64 = 4) and C
// tiles in flight per wave: A/B knobs GSI_LU_RK_CHUNK (64 | 128), GSI_LU_RK_DEPTH (1 | 2).
// Round 5: THREE waves per SIMD.  The kernel is a latency chain per wave (C tile in, 2 K / 4 MFMAs, C tile out) at 44 % of the
// matrix pipe and 0.54 of the HBM peak; with 164 VGPRs + 16 AGPRs it ran two waves per SIMD whatever the chunk.  Told to fit
// three (amdgpu_waves_per_eu on the 64-column instantiations: 160 VGPRs, no AGPR copies, no spills) and with 64-column chunks
// (34 KB of LDS: three workgroups per CU) the four updates of a factorization take 2.4 instead of 2.8 ms: LU 29.4 -> 27.7 - 28.4 ms
// per step in alternating runs (profiles/r05_lu_rankk_occupancy.log).  Four waves (128 VGPRs) spill 46 registers: 32.6.  The
// same 64-column chunks at two waves per SIMD were "noise" in round 3 (tools/ab_rankk.sh): it was the occupancy, not the chunk.
template <int K>
static void launch_rankk(hipStream_t st, unsigned grid, double* Y, int64_t ld, int64_t m, int64_t r_begin, int64_t jb,
                         int64_t c0, int64_t t, const double* U12) {
  static const int depth = getenv("GSI_LU_RK_DEPTH") ? atoi(getenv("GSI_LU_RK_DEPTH")) : 1;
  static const int chunk = getenv("GSI_LU_RK_CHUNK") ? atoi(getenv("GSI_LU_RK_CHUNK")) : 64;
  if (chunk == 64 && depth == 2) launch_rankk_v<K, 2, 64>(st, grid, Y, ld, m, r_begin, jb, c0, t, U12);
  else if (chunk == 64) launch_rankk_v<K, 1, 64>(st, grid, Y, ld, m, r_begin, jb, c0, t, U12);
  else if (depth == 2) launch_rankk_v<K, 2, 128>(st, grid, Y, ld, m, r_begin, jb, c0, t, U12);
  else launch_rankk_v<K, 1, 128>(st, grid, Y, ld, m, r_begin, jb, c0, t, U12);
}

// top l x l: unit diagonal, zero strict upper triangle (what Julia's F.L returns)
__global__ void lu2_extract_L_kernel(double* __restrict__ Y, int64_t ld, int64_t l) {
  const int64_t total = l * l;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % l, c = e / l;
    if (r == c) Y[r + c * ld] = 1.0;
    else if (r < c) Y[r + c * ld] = 0.0;
  }
}

bool lu2_config(int64_t m, int ncus, int* bs, int* rpt, int* grid) {
  if (ncus < 1) return false;
  if (ncus > 256) ncus = 256;               // the leader reads one record per workgroup with <= 2 lanes each
  static const int cfg[4][2] = {{256, 1}, {256, 4}, {512, 4}, {512, 8}};
  for (int c = 0; c < 4; ++c) {
    const int64_t per = (int64_t)cfg[c][0] * cfg[c][1];
    if (m <= (int64_t)ncus * per) { *bs = cfg[c][0]; *rpt = cfg[c][1]; *grid = (int)((m + per - 1) / per); return true; }
  }
  return false;
}

template <int BS, int R>
static void launch_leaf(hipStream_t st, int grid, double* Y, int64_t ld, int64_t m, int64_t l, int64_t jb, int64_t j0,
                        int w, const Lu2Work& wk, uint32_t epoch_base) {
  static const int onehop = getenv("GSI_LU_ONEHOP") ? atoi(getenv("GSI_LU_ONEHOP")) : 1;   // A/B knob; 0 = two hops via a leader
  const int poll_limit = wk.poll_limit > 0 ? wk.poll_limit : POLL_LIMIT;
  LuMrArgs none{};
  if constexpr (BS == 512 && R == 8) {
    if (wk.ov) {                          // taller than the grid's registers: overflow rows evaluated lazily
      hipLaunchKernelGGL((lu_leaf_kernel<512, 8, false, true>), dim3(grid), dim3(512), 0, st, Y, ld, (int32_t)m, (int32_t)l,
                         (int32_t)jb, (int32_t)j0, w, wk.recs, epoch_base, wk.ipiv, wk.info, 1, poll_limit, wk.mute_epoch, none);
      return;
    }
  }
  if (!wk.cooperative) {
    hipLaunchKernelGGL((lu_leaf_kernel<BS, R, false>), dim3(grid), dim3(BS), 0, st, Y, ld, (int32_t)m, (int32_t)l, (int32_t)jb,
                       (int32_t)j0, w, wk.recs, epoch_base, wk.ipiv, wk.info, onehop, poll_limit, wk.mute_epoch, none);
    return;
  }
  // cooperative launch: the runtime guarantees that all `grid` workgroups are resident together (and refuses the launch
  // otherwise) -- what the spin-waits between workgroups rely on when the device is shared with other queues
  int32_t m32 = (int32_t)m, l32 = (int32_t)l, jb32 = (int32_t)jb, j032 = (int32_t)j0;
  int oh = onehop, pl = poll_limit;
  uint32_t eb = epoch_base, mute = wk.mute_epoch;
  unsigned long long* recs = wk.recs;
  int32_t* ipiv = wk.ipiv;
  int32_t* info = wk.info;
  void* args[] = {&Y, &ld, &m32, &l32, &jb32, &j032, &w, &recs, &eb, &ipiv, &info, &oh, &pl, &mute, &none};
  (void)hipLaunchCooperativeKernel((const void*)lu_leaf_kernel<BS, R, false>, dim3(grid), dim3(BS), args, 0, st);
}

template <int BS, int R>
static int leaf_resident_per_cu() {
  int nblk = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)lu_leaf_kernel<BS, R, false>, BS, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nblk;
}
// How many workgroups of the (bs, rpt) leaf kernel one CU holds (registers, LDS, waves): the persistent launch needs
// grid <= that x CUs, or its spin-waits would wait for workgroups that cannot start.
int lu2_resident_per_cu_ov() {
  int nblk = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)lu_leaf_kernel<512, 8, false, true>, 512, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nblk;
}
int lu2_resident_per_cu(int bs, int rpt) {
  if (bs == 256 && rpt == 1) return leaf_resident_per_cu<256, 1>();
  if (bs == 256) return leaf_resident_per_cu<256, 4>();
  if (rpt == 4) return leaf_resident_per_cu<512, 4>();
  return leaf_resident_per_cu<512, 8>();
}

// ---- the leaf launch of the MULTI-RANK factorization: this rank's rows, G = w.grid workgroups per rank, records exchanged
//      through every rank's peer-mapped buffer.  Launch geometry for shards of at most `pad` rows on `nranks` ranks:
//      nranks * grid <= 256 records, every rank the same (bs, rpt, grid).
bool lu2_mr_config(int64_t pad, int nranks, int ncus, int* bs, int* rpt, int* grid, int* hier, int* ov, int force) {
  *ov = 0;
  if (nranks < 1 || nranks > LU2_MAX_RANKS) return false;
  if (force == 2) {                        // (self-test) two workgroups per rank, the rest of the shard as overflow rows
    if (nranks < 2 || pad <= 2 * 4096 || ncus < 2) return false;
    *bs = 512; *rpt = 8; *grid = 2; *hier = 1; *ov = 1;
    return true;
  }
  // GSI_LU_MR_OV_GRID=k (tests): at most k workgroups per rank, the rest of the shard as overflow rows
  static const int ov_cap = getenv("GSI_LU_MR_OV_GRID") ? atoi(getenv("GSI_LU_MR_OV_GRID")) : 0;
  static const int64_t ov_max = getenv("GSI_LU_OV_MAX") ? atoll(getenv("GSI_LU_OV_MAX")) : ((int64_t)5 << 20);
  if (ov_cap > 0 && nranks > 1 && pad > (int64_t)ov_cap * 4096 && ov_cap <= std::min(256 - nranks, ncus)) {
    *bs = 512; *rpt = 8; *grid = ov_cap; *hier = 1; *ov = 1;
    return true;
  }
  static const int cfg[4][2] = {{256, 1}, {256, 4}, {512, 4}, {512, 8}};
  static const char* he = getenv("GSI_LU_MR_HIER");              // 1: always two hops (tests), 0: never
  const bool force_hier = (he != nullptr && he[0] == '1') || force == 1, no_hier = he != nullptr && he[0] == '0' && force != 1;
  // one hop: every workgroup of every rank is a record of the exchange (nranks * grid <= 256)
  const int gmax = std::min(256 / nranks, ncus);
  for (int c = 0; c < 4 && !force_hier; ++c) {
    const int64_t per = (int64_t)cfg[c][0] * cfg[c][1];
    const int64_t g = (pad + per - 1) / per;
    if (g <= gmax) { *bs = cfg[c][0]; *rpt = cfg[c][1]; *grid = (int)std::max<int64_t>(g, 1); *hier = 0; return true; }
  }
  // two hops: a rank's workgroups reduce among themselves first (grid + nranks <= 256 record slots)
  const int gmax2 = std::min(256 - nranks, ncus);
  for (int c = 0; c < 4 && !no_hier && nranks > 1; ++c) {
    const int64_t per = (int64_t)cfg[c][0] * cfg[c][1];
    const int64_t g = (pad + per - 1) / per;
    if (g <= gmax2) { *bs = cfg[c][0]; *rpt = cfg[c][1]; *grid = (int)std::max<int64_t>(g, 1); *hier = 1; return true; }
  }
  // taller still (up to GSI_LU_OV_MAX rows per rank): every CU a <512, 8> workgroup, two hops, the rows beyond the resident
  // window evaluated lazily (OV)
  if (!no_hier && nranks > 1 && gmax2 >= 1 && pad <= ov_max && !(getenv("GSI_LU_OV") != nullptr && getenv("GSI_LU_OV")[0] == '0')) {
    *bs = 512; *rpt = 8; *grid = gmax2; *hier = 1; *ov = 1;
    return true;
  }
  return false;
}
template <int BS, int R>
static void launch_leaf_mr_t(hipStream_t st, const Lu2MrWork& w, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t m,
                             int64_t l, int64_t jb, int64_t j0, int wd, const double* us, uint32_t epoch_base) {
  LuMrArgs a{};
  a.rank = w.rank; a.nranks = w.nranks; a.gbase = (int32_t)row0; a.mtot = (int32_t)m; a.us = us;
  a.hier = w.hier; a.slots = w.hier ? w.grid + w.nranks : w.nranks * w.grid;
  for (int q = 0; q < w.nranks; ++q) a.peer[q] = w.peer[q];
  const int poll_limit = w.poll_limit > 0 ? w.poll_limit : POLL_LIMIT;
  if constexpr (BS == 512 && R == 8) {
    if (w.ov) {                           // shards taller than the grid's registers: overflow rows evaluated lazily
      hipLaunchKernelGGL((lu_leaf_kernel<512, 8, true, true>), dim3(w.grid), dim3(512), 0, st, Y, ld, (int32_t)mloc, (int32_t)l,
                         (int32_t)jb, (int32_t)j0, wd, w.peer[w.rank], epoch_base, w.ipiv, w.info, 1, poll_limit, 0u, a);
      return;
    }
  }
  hipLaunchKernelGGL((lu_leaf_kernel<BS, R, true>), dim3(w.grid), dim3(BS), 0, st, Y, ld, (int32_t)mloc, (int32_t)l, (int32_t)jb,
                     (int32_t)j0, wd, w.peer[w.rank], epoch_base, w.ipiv, w.info, 1, poll_limit, 0u, a);
}
void lu2_leaf_mr(hipStream_t st, const Lu2MrWork& w, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t m, int64_t l,
                 int64_t jb, int64_t j0, int wd, const double* us, uint32_t epoch_base) {
  if (w.bs == 256 && w.rpt == 1) launch_leaf_mr_t<256, 1>(st, w, Y, ld, mloc, row0, m, l, jb, j0, wd, us, epoch_base);
  else if (w.bs == 256) launch_leaf_mr_t<256, 4>(st, w, Y, ld, mloc, row0, m, l, jb, j0, wd, us, epoch_base);
  else if (w.rpt == 4) launch_leaf_mr_t<512, 4>(st, w, Y, ld, mloc, row0, m, l, jb, j0, wd, us, epoch_base);
  else launch_leaf_mr_t<512, 8>(st, w, Y, ld, mloc, row0, m, l, jb, j0, wd, us, epoch_base);
}
int lu2_mr_resident_per_cu_ov() {
  int nblk = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)lu_leaf_kernel<512, 8, true, true>, 512, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nblk;
}
int lu2_mr_resident_per_cu(int bs, int rpt) {
  int nblk = 0;
  hipError_t e;
  if (bs == 256 && rpt == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)lu_leaf_kernel<256, 1, true>, 256, 0);
  else if (bs == 256) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)lu_leaf_kernel<256, 4, true>, 256, 0);
  else if (rpt == 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)lu_leaf_kernel<512, 4, true>, 512, 0);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)lu_leaf_kernel<512, 8, true>, 512, 0);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return nblk;
}
// a rank's exchange buffer: two parity sets of records, two U mailboxes (kp x 8 doubles as granule pairs), two table boxes
// (16 rows x l <= LU2_MR_MAXL columns as granule pairs) for the rows a leaf's pivots exchange between ranks
size_t lu2_mr_record_granules(int nranks, int grid) {
  return (size_t)2 * (size_t)nranks * (size_t)grid * REC + (size_t)2 * (2 * KPMAX * LW) + (size_t)2 * ((size_t)2 * LW * 2 * LU2_MR_MAXL);
}

// ---- the row interchanges of one leaf's pivots on the columns OUTSIDE the leaf, across ranks (LAPACK swaps whole rows; the
//      leaf kernel moved the leaf's own 8 columns in registers).  The <= 16 rows involved -- j0 .. j0 + w - 1 and the pivot
//      rows r_s -- are collected into a table (every rank contributes the rows it owns, zeros elsewhere; the host
//      all-reduces it), the w swaps are replayed on the table, every rank writes back the rows it owns.
//      Slot t < w: row j0 + t; slot w + s: pivot row r_s unless that row already has a slot (then the slot stays zero).
__device__ inline int lus_swap_slot(const int32_t* piv, int w, int32_t j0, int32_t row) {     // canonical slot of a row
  if (row >= j0 && row < j0 + w) return row - j0;
  for (int s2 = 0; s2 < w; ++s2)
    if (piv[s2] == row) return w + s2;
  return -1;
}
// slot of a row in the peer kernel's layout: t < LW: row j0 + t; LW + s: pivot r_s (first occurrence)
__device__ inline int lus_swap_slot_lw(const int32_t* piv, int w, int32_t j0, int32_t row) {
  if (row >= j0 && row < j0 + w) return row - j0;
  for (int s2 = 0; s2 < w; ++s2)
    if (piv[s2] == row) return LW + s2;
  return -1;
}
__global__ __launch_bounds__(256) void lus_swap_pack_kernel(const double* __restrict__ Y, int64_t ld, int64_t mloc, int64_t row0,
                                                            int64_t l, int32_t j0, int w, const int32_t* __restrict__ ipiv,
                                                            double* __restrict__ table) {
  __shared__ int32_t piv[LW];
  if (threadIdx.x < LW) piv[threadIdx.x] = (threadIdx.x < (unsigned)w) ? ipiv[j0 + threadIdx.x] : -1;
  __syncthreads();
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)2 * LW * l; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e / l);
    const int64_t c = e % l;
    double v = 0.0;
    if (t < 2 * w) {
      const int32_t row = (t < w) ? j0 + t : piv[t - w];
      const bool canonical = (t < w) || (lus_swap_slot(piv, w, j0, row) == t);
      if (canonical && row >= row0 && row < row0 + mloc && !(c >= j0 && c < j0 + w)) v = Y[(row - row0) + c * ld];
    }
    table[e] = v;
  }
}
__global__ __launch_bounds__(256) void lus_swap_apply_kernel(double* __restrict__ Y, int64_t ld, int64_t mloc, int64_t row0,
                                                             int64_t l, int32_t j0, int w, const int32_t* __restrict__ ipiv,
                                                             const double* __restrict__ table) {
  __shared__ int32_t piv[LW];
  if (threadIdx.x < LW) piv[threadIdx.x] = (threadIdx.x < (unsigned)w) ? ipiv[j0 + threadIdx.x] : -1;
  __syncthreads();
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < l; c += (int64_t)gridDim.x * 256) {
    if (c >= j0 && c < j0 + w) continue;
    double v[2 * LW];
#pragma unroll
    for (int t = 0; t < 2 * LW; ++t) v[t] = table[(int64_t)t * l + c];
    for (int s2 = 0; s2 < w; ++s2) {                       // LAPACK's order: swap rows j0 + s and r_s
      const int b = lus_swap_slot(piv, w, j0, piv[s2]);
      if (b >= 0 && b != s2) {
        double va = 0.0, vb = 0.0;
#pragma unroll
        for (int t = 0; t < 2 * LW; ++t) { if (t == s2) va = v[t]; if (t == b) vb = v[t]; }
#pragma unroll
        for (int t = 0; t < 2 * LW; ++t) { if (t == s2) v[t] = vb; if (t == b) v[t] = va; }
      }
    }
#pragma unroll
    for (int t = 0; t < 2 * LW; ++t) {
      if (t < 2 * w) {
        const int32_t row = (t < w) ? j0 + t : piv[t - w];
        const bool canonical = (t < w) || (lus_swap_slot(piv, w, j0, row) == t);
        if (canonical && row >= row0 && row < row0 + mloc) Y[(row - row0) + c * ld] = v[t];
      }
    }
  }
}
// The same interchange WITHOUT a host-sequenced collective: every rank pushes the rows it owns into every other rank's table
// box (granule pairs tagged with the leaf's first epoch, system-scope stores into peer-mapped memory), polls its own box for
// the rows the others own, replays the swaps and writes back its rows.  One launch per leaf and rank, thread = one column.
// A rank that owns none of the <= 16 rows has nothing to write and leaves at once.
__global__ __launch_bounds__(256) void lus_swap_peer_kernel(double* __restrict__ Y, int64_t ld, int64_t mloc, int64_t row0,
                                                            int64_t l, int32_t j0, int w, const int32_t* __restrict__ ipiv,
                                                            LuMrArgs mr, unsigned long long* __restrict__ own, size_t box_off,
                                                            uint32_t tag, int64_t pad, int poll_limit,
                                                            int32_t* __restrict__ info) {
  __shared__ int32_t piv[LW];
  __shared__ int s_any;
  if (threadIdx.x == 0) s_any = 0;
  if (threadIdx.x < LW) piv[threadIdx.x] = (threadIdx.x < (unsigned)w) ? ipiv[j0 + threadIdx.x] : -1;
  __syncthreads();
  if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) return;   // a timed-out factorization drains
  // canonical slots and their owners (the same on every rank)
  int32_t srow[2 * LW];
  bool scan[2 * LW], smine[2 * LW];
  bool any_mine = false;
#pragma unroll
  for (int t = 0; t < 2 * LW; ++t) {
    const int32_t row = (t < w) ? j0 + t : ((t >= LW && t - LW < w) ? piv[t - LW] : -1);
    srow[t] = row;
    bool canonical = row >= 0;
    if (canonical && t >= LW) {
      if (row >= j0 && row < j0 + w) canonical = false;
      for (int s2 = 0; s2 < t - LW; ++s2) if (piv[s2] == row) canonical = false;
    }
    scan[t] = canonical;
    smine[t] = canonical && row >= row0 && row < row0 + mloc;
    any_mine = any_mine || smine[t];
  }
  if (!any_mine) return;
  const size_t lq = (size_t)l;
  bool timed_out = false;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < l; c += (int64_t)gridDim.x * 256) {
    if (c >= j0 && c < j0 + w) continue;
    double v[2 * LW];
#pragma unroll
    for (int t = 0; t < 2 * LW; ++t) {           // my rows: read, push to everyone else
      v[t] = 0.0;
      if (smine[t]) {
        v[t] = Y[(srow[t] - row0) + c * ld];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v[t]);
        const size_t off = box_off + ((size_t)t * lq + (size_t)c) * 2;
        for (int q = 0; q < mr.nranks; ++q) {
          if (q == mr.rank) continue;
          __hip_atomic_store(mr.peer[q] + off, ((unsigned long long)tag << 32) | (uint32_t)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(mr.peer[q] + off + 1, ((unsigned long long)tag << 32) | (uint32_t)(bits >> 32), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    {                                            // the others' rows: poll my own box, all slots of a round in flight together
      unsigned long long lo[2 * LW], hi[2 * LW];
      int tries = timed_out ? poll_limit : 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int t = 0; t < 2 * LW; ++t) {
          if (scan[t] && !smine[t]) {
            const size_t off = box_off + ((size_t)t * lq + (size_t)c) * 2;
            lo[t] = poll_granule<true>(own + off);
            hi[t] = poll_granule<true>(own + off + 1);
          }
        }
#pragma unroll
        for (int t = 0; t < 2 * LW; ++t)
          if (scan[t] && !smine[t]) ok = ok && ((uint32_t)(lo[t] >> 32) == tag && (uint32_t)(hi[t] >> 32) == tag);
        if (ok) break;
        if (++tries > poll_limit) { timed_out = true; lu_timeout_note(info, 6, (int)c, tag, mr.rank * 1024 + (int)blockIdx.x); break; }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int t = 0; t < 2 * LW; ++t)
        if (scan[t] && !smine[t])
          v[t] = __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi[t] << 32) | (uint32_t)lo[t]));
    }
    for (int s2 = 0; s2 < w; ++s2) {             // LAPACK's order: swap rows j0 + s and r_s
      const int b = lus_swap_slot_lw(piv, w, j0, piv[s2]);
      if (b >= 0 && b != s2) {
        double va = 0.0, vb = 0.0;
#pragma unroll
        for (int t = 0; t < 2 * LW; ++t) { if (t == s2) va = v[t]; if (t == b) vb = v[t]; }
#pragma unroll
        for (int t = 0; t < 2 * LW; ++t) { if (t == s2) v[t] = vb; if (t == b) v[t] = va; }
      }
    }
#pragma unroll
    for (int t = 0; t < 2 * LW; ++t)
      if (smine[t]) Y[(srow[t] - row0) + c * ld] = v[t];
  }
  if (timed_out) atomicExch(info, -1);
}
void lus_swap_peer(hipStream_t st, const Lu2MrWork& w, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t m, int64_t l,
                   int64_t j0, int wd, uint32_t epoch_base) {
  LuMrArgs a{};
  a.rank = w.rank; a.nranks = w.nranks; a.gbase = (int32_t)row0; a.mtot = (int32_t)m; a.us = nullptr;
  for (int q = 0; q < w.nranks; ++q) a.peer[q] = w.peer[q];
  const int64_t pad = (m + w.nranks - 1) / w.nranks;
  const size_t G = w.hier ? (size_t)w.grid + (size_t)w.nranks : (size_t)w.nranks * (size_t)w.grid;
  const size_t box = (size_t)2 * G * REC + (size_t)2 * (2 * KPMAX * LW) + (size_t)((epoch_base >> 3) & 1u) * ((size_t)2 * LW * 2 * LU2_MR_MAXL);
  const int poll_limit = w.poll_limit > 0 ? w.poll_limit : POLL_LIMIT;
  const int g = (int)std::min<int64_t>((l + 255) / 256, 64);
  hipLaunchKernelGGL(lus_swap_peer_kernel, dim3(g), dim3(256), 0, st, Y, ld, mloc, row0, l, (int32_t)j0, wd, w.ipiv, a, w.peer[w.rank], box,
                     epoch_base + 1u, pad, poll_limit, w.info);
}

void lus_swap_pack(hipStream_t st, const double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t l, int64_t j0, int w,
                   const int32_t* ipiv, double* table) {
  const int g = (int)std::min<int64_t>((2 * LW * l + 255) / 256, 256);
  hipLaunchKernelGGL(lus_swap_pack_kernel, dim3(g), dim3(256), 0, st, Y, ld, mloc, row0, l, (int32_t)j0, w, ipiv, table);
}
void lus_swap_apply(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t l, int64_t j0, int w,
                    const int32_t* ipiv, const double* table) {
  const int g = (int)std::min<int64_t>((l + 255) / 256, 64);
  hipLaunchKernelGGL(lus_swap_apply_kernel, dim3(g), dim3(256), 0, st, Y, ld, mloc, row0, l, (int32_t)j0, w, ipiv, table);
}

void lu2_L(hipStream_t st, double* Y, int64_t m, int64_t l, int64_t ld, const Lu2Work& w) {
  // the record tags count pivot steps from 1: clear both record sets
  (void)hipMemsetAsync(w.recs, 0, sizeof(unsigned long long) * 2 * ((size_t)w.grid + LU2_RES_COPIES) * REC, st);
  uint32_t epoch = 0;
  const int nb = w.nb;
  for (int64_t jb = 0; jb < l; jb += nb) {
    const int b = (int)((l - jb < nb) ? (l - jb) : nb);
    for (int64_t j0 = jb; j0 < jb + b; j0 += LW) {
      const int wd = (int)((jb + b - j0 < LW) ? (jb + b - j0) : LW);
      // every leaf keeps the same grid: workgroups whose rows lie beyond m still take part in the exchange
      if (w.bs == 256 && w.rpt == 1) launch_leaf<256, 1>(st, w.grid, Y, ld, m, l, jb, j0, wd, w, epoch);
      else if (w.bs == 256) launch_leaf<256, 4>(st, w.grid, Y, ld, m, l, jb, j0, wd, w, epoch);
      else if (w.rpt == 4) launch_leaf<512, 4>(st, w.grid, Y, ld, m, l, jb, j0, wd, w, epoch);
      else launch_leaf<512, 8>(st, w.grid, Y, ld, m, l, jb, j0, wd, w, epoch);
      epoch += (uint32_t)LW;            // a narrow last leaf still runs (gated) 8 steps
    }
    const int64_t c0 = jb + b, t = l - c0;
    if (t > 0) {                               // only full blocks have columns to their right
      const int64_t mr = m - c0;
      const unsigned gu = (unsigned)((t + 63) / 64);
      const unsigned gr = (unsigned)((mr + 127) / 128);
      if (b == 64) {
        hipLaunchKernelGGL(lu_u12_kernel<64>, dim3(gu), dim3(256), 0, st, Y, ld, jb, jb, c0, l, w.u12);
        if (mr > 0) launch_rankk<64>(st, gr, Y, ld, m, c0, jb, c0, t, w.u12);
      } else {
        hipLaunchKernelGGL(lu_u12_kernel<32>, dim3(gu), dim3(256), 0, st, Y, ld, jb, jb, c0, l, w.u12);
        if (mr > 0) launch_rankk<32>(st, gr, Y, ld, m, c0, jb, c0, t, w.u12);
      }
    }
  }
  int eb = (int)((l * l + 255) / 256);
  if (eb > 1024) eb = 1024;
  hipLaunchKernelGGL(lu2_extract_L_kernel, dim3(eb), dim3(256), 0, st, Y, ld, l);
#ifdef GSI_LU_TRACE
  if (const char* path = getenv("GSI_LU_TRACE")) {
    unsigned long long h[4 * 8 * 8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lu_trace), sizeof(h));
    if (FILE* f = fopen(path, "w")) {
      for (int tw = 0; tw < 4; ++tw)
        for (int s = 0; s < 8; ++s) {
          fprintf(f, "wg%d step%d", tw, s);
          for (int ph = 0; ph < 6; ++ph) fprintf(f, " %llu", h[(tw * 8 + s) * 8 + ph]);
          fprintf(f, "\n");
        }
      fclose(f);
    }
  }
#endif
}

// =====================================================================================================================
// Panels TALLER than the register file (more than 4096 rows per CU: 1100^2 grids, the 256^3 / 512^3 grids of the FFT
// operator): the same blocks, leaves, pivots and arithmetic -- operation for operation, so the factors are bit-identical
// to the register-resident kernel's -- with the leaf's 8 columns STREAMED instead of resident, and evaluated lazily:
//   open   (one pass):  a <- a - L[:, jb:j0] U12 for the leaf's columns, written back once (kp + 8 reads, 8 writes; a
//                       block's first leaf has nothing pending: one read of column j0); arg-max of column j0;
//   pivot s (1 workgroup): finishes the arg-max, interchanges rows j0+s and r in all l columns, eliminates the new pivot
//                       row with the s pivot rows before it and keeps it (u_s) in a 8 x 8 state block in HBM;
//   cand  s+1 (one pass over s+2 columns, READ ONLY): every row below the pivots re-derives its current value in column
//                       s+1 from its 8 STORED values and u_0..u_s (s(s+1)/2 fmas in registers) -- the right-looking
//                       sweeps of panel_lu.hip read AND write the live columns at every step instead;
//   close  (one pass):  multipliers of all rows below the leaf's pivots (8 reads, 8 writes).
// Column passes per 8-column leaf: (kp + 16) + 35 + 16 against 88 + in-block products for the sweeps; no spin-waits, no
// co-residency assumption.  Blocks end with the register-resident path's own U12 + rank-64 kernels.
// =====================================================================================================================
namespace {
constexpr int T_BS = 256;       // threads per workgroup
constexpr int T_R = 4;          // rows per thread and pass iteration (T_R * (kp-chunk + 8) loads in flight)
constexpr int T_MAXWG = 4096;   // workgroups per streaming launch (grid-stride over the rows) = partial arg-maxes per step
struct Lu3State {               // the leaf's pivot rows so far: u[t] = pivot row t after its elimination (entries k > t are U), 1 / u[t][t]
  double u[LW][LW];
  double rp[LW];
};

// What ends every candidate pass: the workgroup's arg-max goes to pval / pidx [blockIdx.x].  A one-workgroup launch then
// finishes pivot step s of the leaf at j0: reduces the partial arg-maxes to the pivot row r, interchanges rows j0 + s and r
// in all l columns, eliminates the new pivot row with the s pivot rows before it and keeps it (u_s) in the state block.
// (Letting the workgroup that arrives LAST at a ticket counter do that inside the candidate pass -- one launch per pivot step
// instead of two -- measured 1.7 x SLOWER at 1.2e6 rows: thousands of device-scope atomics on one address per step.)
struct Lu3Step {
  double* Y;
  int64_t ld, m;
  int32_t l, j0;
  int s, w;                    // pivot step 0 <= s < w of the w-column leaf
  double* pval;
  int64_t* pidx;
  Lu3State* stt;
  int32_t* ipiv;
  int32_t* info;
};
__device__ inline void lu3_step_tail(double best, int32_t besti, const Lu3Step& p) {
  __shared__ double s_v[T_BS / 64];
  __shared__ int32_t s_i[T_BS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  wave_argmax(best, besti);
  if (lane == 0) { s_v[wave] = best; s_i[wave] = besti; }
  __syncthreads();
  if (tid == 0) {
    double bv = -1.0;
    int32_t bi = -1;
    for (int q = 0; q < T_BS / 64; ++q)
      if (s_i[q] >= 0 && (s_v[q] > bv || (s_v[q] == bv && s_i[q] < bi))) { bv = s_v[q]; bi = s_i[q]; }
    p.pval[blockIdx.x] = bv;
    p.pidx[blockIdx.x] = bi;
  }
}
__global__ __launch_bounds__(T_BS) void lu3_pivot_kernel(Lu3Step p, int nwg) {
  __shared__ double s_v[T_BS / 64];
  __shared__ int32_t s_i[T_BS / 64];
  __shared__ double s_x[LW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double best = -1.0;
  int32_t besti = -1;
  for (int e = tid; e < nwg; e += T_BS) {
    const double v = p.pval[e];
    const int32_t i = (int32_t)p.pidx[e];
    if (i >= 0 && (v > best || (v == best && i < besti))) { best = v; besti = i; }
  }
  wave_argmax(best, besti);
  if (lane == 0) { s_v[wave] = best; s_i[wave] = besti; }
  __syncthreads();
  double bestv = -1.0;
  int32_t r = -1;
  for (int q = 0; q < T_BS / 64; ++q)
    if (s_i[q] >= 0 && (s_v[q] > bestv || (s_v[q] == bestv && s_i[q] < r))) { bestv = s_v[q]; r = s_i[q]; }
  double* const Y = p.Y;
  const int64_t ld = p.ld;
  const int32_t j0 = p.j0, j = p.j0 + p.s;
  const int s = p.s, w = p.w;
  const bool valid = (r >= j && (int64_t)r < p.m);
  if (!valid) r = j;                              // all-NaN column: no interchange (as lu_leaf_kernel)
  if (tid == 0) {
    p.ipiv[j] = r;
    if (!(bestv > 0.0)) atomicCAS(p.info, 0, j + 1);
  }
  if (r != j) {
    for (int32_t c = tid; c < p.l; c += T_BS) {
      if (c >= j0 && c < j0 + w) continue;
      double* col = Y + (int64_t)c * ld;
      const double vj = col[j], vr = col[r];
      col[j] = vr;
      col[r] = vj;
    }
  }
  if (tid < LW) {                                 // the leaf's own columns: STORED values travel, row r's become the pivot row
    const int k = tid;
    double xr = 0.0;
    if (k < w) {
      double* col = Y + (int64_t)(j0 + k) * ld;
      xr = col[r];
      if (r != j) col[r] = col[j];
    }
    s_x[k] = xr;
  }
  __syncthreads();
  if (tid == 0) {
    Lu3State* stt = p.stt;
    double x[LW];
#pragma unroll
    for (int k = 0; k < LW; ++k) x[k] = s_x[k];
    for (int t = 0; t < s; ++t) {
      const double rp = stt->rp[t];
      const double lt = (rp != 0.0) ? x[t] * rp : x[t];
      x[t] = lt;
      for (int k = t + 1; k < LW; ++k) x[k] -= lt * stt->u[t][k];
    }
    for (int k = 0; k < LW; ++k) stt->u[s][k] = x[k];
    const double piv = x[s];
    stt->rp[s] = (piv != 0.0) ? 1.0 / piv : 0.0;
    for (int k = 0; k < w; ++k) Y[j + (int64_t)(j0 + k) * ld] = x[k];      // row j is final: multipliers, then U
  }
}

// FUSED: the PREVIOUS leaf (columns j0 - 8 .. j0 - 1, pivot rows in `prev`) was not closed: its columns still hold stored
// values below its pivots; this pass turns them into multipliers on the way (written back once) -- 8 column reads less per leaf
template <bool PENDING, bool FUSED>
__global__ __launch_bounds__(T_BS) void lu3_open_kernel(int32_t jb, Lu3Step p) {
  static_assert(PENDING || !FUSED, "a block's first leaf has no predecessor to close");
  double* const Y = p.Y;
  const int64_t ld = p.ld, m = p.m;
  const int32_t j0 = p.j0;
  const int w = p.w;
  const Lu3State* const prev = p.stt;
  constexpr int NW = T_BS / 64;
  __shared__ double Ls[PENDING ? KPMAX * LSP : 1];
  __shared__ double Us[PENDING ? KPMAX * LW : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kp = j0 - jb;
  if (PENDING) {     // U12 = L11^-1 A12 of the leaf's columns, by every workgroup for itself (as lu_leaf_kernel does)
    for (int e = tid; e < kp * kp; e += T_BS) {
      const int r = e % kp, c = e / kp;
      Ls[r * LSP + c] = Y[(jb + r) + (int64_t)(jb + c) * ld];
    }
    __syncthreads();
    for (int v = wave; v < LW; v += NW) {
      double x = (lane < kp && v < w) ? Y[(jb + lane) + (int64_t)(j0 + v) * ld] : 0.0;
      for (int cp = 0; cp < kp; ++cp) {
        const double xc = readlane_d(x, __builtin_amdgcn_readfirstlane(cp));
        if (lane > cp && lane < kp) x -= Ls[lane * LSP + cp] * xc;
      }
      if (lane < kp) Us[lane * LW + v] = x;
    }
    __syncthreads();
  }
  double best = -1.0;
  int32_t besti = -1;
  const int64_t stride = (int64_t)gridDim.x * (T_BS * T_R);
  for (int64_t base = (int64_t)j0 + (int64_t)blockIdx.x * (T_BS * T_R) + tid; base < m; base += stride) {
    if (!PENDING) {
      double v[T_R];
#pragma unroll
      for (int rr = 0; rr < T_R; ++rr) {
        const int64_t i = base + rr * T_BS;
        v[rr] = (i < m) ? fabs(Y[i + (int64_t)j0 * ld]) : -1.0;
      }
#pragma unroll
      for (int rr = 0; rr < T_R; ++rr)
        if (v[rr] > best) { best = v[rr]; besti = (int32_t)(base + rr * T_BS); }
    } else {
      if (FUSED) {     // phase A: the previous leaf's columns become multipliers (the pending loop below re-reads them: L2 hits)
        double x[T_R][LW];
#pragma unroll
        for (int t = 0; t < LW; ++t) {
          const double* cb = Y + (int64_t)(j0 - LW + t) * ld;
#pragma unroll
          for (int rr = 0; rr < T_R; ++rr) {
            const int64_t i = base + rr * T_BS;
            x[rr][t] = (i < m) ? cb[i] : 0.0;
          }
        }
#pragma unroll
        for (int rr = 0; rr < T_R; ++rr) {
#pragma unroll
          for (int t = 0; t < LW; ++t) {
            const double rp = prev->rp[t];
            const double lt = (rp != 0.0) ? x[rr][t] * rp : x[rr][t];
            x[rr][t] = lt;
#pragma unroll
            for (int k = t + 1; k < LW; ++k) x[rr][k] -= lt * prev->u[t][k];
          }
        }
#pragma unroll
        for (int t = 0; t < LW; ++t) {
          double* cb = Y + (int64_t)(j0 - LW + t) * ld;
#pragma unroll
          for (int rr = 0; rr < T_R; ++rr) {
            const int64_t i = base + rr * T_BS;
            if (i < m) cb[i] = x[rr][t];
          }
        }
      }
      double a[T_R][LW];
#pragma unroll
      for (int k = 0; k < LW; ++k) {
        const double* cb = Y + (int64_t)(j0 + (k < w ? k : 0)) * ld;
#pragma unroll
        for (int rr = 0; rr < T_R; ++rr) {
          const int64_t i = base + rr * T_BS;
          a[rr][k] = (i < m && k < w) ? cb[i] : 0.0;
        }
      }
      for (int c = 0; c < kp; c += 4) {            // kp is a multiple of the leaf width
        double lv[4][T_R];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const double* cb = Y + (int64_t)(jb + c + cc) * ld;
#pragma unroll
          for (int rr = 0; rr < T_R; ++rr) {
            const int64_t i = base + rr * T_BS;
            lv[cc][rr] = (i < m) ? cb[i] : 0.0;
          }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          double u[LW];
#pragma unroll
          for (int k = 0; k < LW; ++k) u[k] = Us[(c + cc) * LW + k];
#pragma unroll
          for (int rr = 0; rr < T_R; ++rr)
#pragma unroll
            for (int k = 0; k < LW; ++k) a[rr][k] -= lv[cc][rr] * u[k];
        }
      }
#pragma unroll
      for (int k = 0; k < LW; ++k) {
        if (k < w) {
          double* cb = Y + (int64_t)(j0 + k) * ld;
#pragma unroll
          for (int rr = 0; rr < T_R; ++rr) {
            const int64_t i = base + rr * T_BS;
            if (i < m) cb[i] = a[rr][k];
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < T_R; ++rr) {
        const int64_t i = base + rr * T_BS;
        const double av = fabs(a[rr][0]);
        if (i < m && av > best) { best = av; besti = (int32_t)i; }
      }
    }
  }
  lu3_step_tail(best, besti, p);                  // step 0 of the leaf
}

// candidates of column j0 + S1 among the rows below the S1 pivots found so far: read only
template <int S1>
__global__ __launch_bounds__(T_BS) void lu3_cand_kernel(Lu3Step p) {      // p.s == S1
  const double* const Y = p.Y;
  const int64_t ld = p.ld, m = p.m;
  const int32_t j0 = p.j0;
  const Lu3State* const stt = p.stt;
  double u[S1][S1 + 1], rp[S1];
#pragma unroll
  for (int t = 0; t < S1; ++t) {
    rp[t] = stt->rp[t];
#pragma unroll
    for (int k = t + 1; k <= S1; ++k) u[t][k] = stt->u[t][k];
  }
  double best = -1.0;
  int32_t besti = -1;
  const int64_t stride = (int64_t)gridDim.x * (T_BS * T_R);
  for (int64_t base = (int64_t)j0 + S1 + (int64_t)blockIdx.x * (T_BS * T_R) + threadIdx.x; base < m; base += stride) {
    double x[T_R][S1 + 1];
#pragma unroll
    for (int k = 0; k <= S1; ++k) {
      const double* cb = Y + (int64_t)(j0 + k) * ld;
#pragma unroll
      for (int rr = 0; rr < T_R; ++rr) {
        const int64_t i = base + rr * T_BS;
        x[rr][k] = (i < m) ? cb[i] : 0.0;
      }
    }
#pragma unroll
    for (int rr = 0; rr < T_R; ++rr) {
#pragma unroll
      for (int t = 0; t < S1; ++t) {
        const double lt = (rp[t] != 0.0) ? x[rr][t] * rp[t] : x[rr][t];
#pragma unroll
        for (int k = t + 1; k <= S1; ++k) x[rr][k] -= lt * u[t][k];
      }
      const int64_t i = base + rr * T_BS;
      const double av = fabs(x[rr][S1]);
      if (i < m && av > best) { best = av; besti = (int32_t)i; }
    }
  }
  lu3_step_tail(best, besti, p);
}

// the leaf is done: multipliers of every row below its w pivots
__global__ __launch_bounds__(T_BS) void lu3_close_kernel(double* __restrict__ Y, int64_t ld, int64_t m, int32_t j0, int w,
                                                         const Lu3State* stt) {
  double u[LW][LW], rp[LW];
#pragma unroll
  for (int t = 0; t < LW; ++t) {
    rp[t] = stt->rp[t];
#pragma unroll
    for (int k = t + 1; k < LW; ++k) u[t][k] = stt->u[t][k];
  }
  const int64_t stride = (int64_t)gridDim.x * (T_BS * T_R);
  for (int64_t base = (int64_t)j0 + w + (int64_t)blockIdx.x * (T_BS * T_R) + threadIdx.x; base < m; base += stride) {
    double x[T_R][LW];
#pragma unroll
    for (int k = 0; k < LW; ++k) {
      const double* cb = Y + (int64_t)(j0 + (k < w ? k : 0)) * ld;
#pragma unroll
      for (int rr = 0; rr < T_R; ++rr) {
        const int64_t i = base + rr * T_BS;
        x[rr][k] = (i < m && k < w) ? cb[i] : 0.0;
      }
    }
#pragma unroll
    for (int rr = 0; rr < T_R; ++rr) {
#pragma unroll
      for (int t = 0; t < LW; ++t) {
        if (t < w) {
          const double lt = (rp[t] != 0.0) ? x[rr][t] * rp[t] : x[rr][t];
          x[rr][t] = lt;
#pragma unroll
          for (int k = t + 1; k < LW; ++k) x[rr][k] -= lt * u[t][k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < LW; ++k) {
      if (k < w) {
        double* cb = Y + (int64_t)(j0 + k) * ld;
#pragma unroll
        for (int rr = 0; rr < T_R; ++rr) {
          const int64_t i = base + rr * T_BS;
          if (i < m) cb[i] = x[rr][k];
        }
      }
    }
  }
}

static int lu3_grid(int64_t rows) {
  const int64_t per = (int64_t)T_BS * T_R;
  const int64_t g = (rows + per - 1) / per;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, T_MAXWG));
}
}  // namespace

size_t lu3_work_bytes(int64_t l) {
  return sizeof(double) * T_MAXWG + sizeof(int64_t) * T_MAXWG + sizeof(Lu3State) + sizeof(double) * (size_t)LU2_NB * (size_t)l +
         sizeof(int32_t) * (size_t)(l + 4) + 512;
}

void lu3_L(hipStream_t st, double* Y, int64_t m, int64_t l, int64_t ld, void* work, int32_t* info, int32_t** ipiv_out) {
  char* base = (char*)work;
  double* pval = (double*)base; base += sizeof(double) * T_MAXWG;
  int64_t* pidx = (int64_t*)base; base += sizeof(int64_t) * T_MAXWG;
  Lu3State* stt = (Lu3State*)base; base += sizeof(Lu3State);
  double* u12 = (double*)base; base += sizeof(double) * (size_t)LU2_NB * (size_t)l;
  int32_t* ipiv = (int32_t*)base;
  *ipiv_out = ipiv;
  const int nb = LU2_NB;
  static const bool fuse = !(getenv("GSI_LU_TALL_NOFUSE") != nullptr);     // A/B: every leaf closed by a pass of its own
  for (int64_t jb = 0; jb < l; jb += nb) {
    const int b = (int)((l - jb < nb) ? (l - jb) : nb);
    for (int64_t j0 = jb; j0 < jb + b; j0 += LW) {
      const int wd = (int)((jb + b - j0 < LW) ? (jb + b - j0) : LW);
      Lu3Step p{Y, ld, m, (int32_t)l, (int32_t)j0, 0, wd, pval, pidx, stt, ipiv, info};
      const int g = lu3_grid(m - j0);
      if (j0 == jb) hipLaunchKernelGGL((lu3_open_kernel<false, false>), dim3(g), dim3(T_BS), 0, st, (int32_t)jb, p);
      else if (fuse) hipLaunchKernelGGL((lu3_open_kernel<true, true>), dim3(g), dim3(T_BS), 0, st, (int32_t)jb, p);   // closes the leaf before
      else hipLaunchKernelGGL((lu3_open_kernel<true, false>), dim3(g), dim3(T_BS), 0, st, (int32_t)jb, p);
      hipLaunchKernelGGL(lu3_pivot_kernel, dim3(1), dim3(T_BS), 0, st, p, g);
      for (int s1 = 1; s1 < wd; ++s1) {           // candidates of column s1, then its pivot
        p.s = s1;
        const int gc = lu3_grid(m - j0 - s1);
        switch (s1) {
          case 1: hipLaunchKernelGGL(lu3_cand_kernel<1>, dim3(gc), dim3(T_BS), 0, st, p); break;
          case 2: hipLaunchKernelGGL(lu3_cand_kernel<2>, dim3(gc), dim3(T_BS), 0, st, p); break;
          case 3: hipLaunchKernelGGL(lu3_cand_kernel<3>, dim3(gc), dim3(T_BS), 0, st, p); break;
          case 4: hipLaunchKernelGGL(lu3_cand_kernel<4>, dim3(gc), dim3(T_BS), 0, st, p); break;
          case 5: hipLaunchKernelGGL(lu3_cand_kernel<5>, dim3(gc), dim3(T_BS), 0, st, p); break;
          case 6: hipLaunchKernelGGL(lu3_cand_kernel<6>, dim3(gc), dim3(T_BS), 0, st, p); break;
          default: hipLaunchKernelGGL(lu3_cand_kernel<7>, dim3(gc), dim3(T_BS), 0, st, p); break;
        }
        hipLaunchKernelGGL(lu3_pivot_kernel, dim3(1), dim3(T_BS), 0, st, p, gc);
      }
      const bool next_opens = fuse && (j0 + wd < jb + b);        // the next leaf of this block closes this one on its way
      if (m > j0 + wd && !next_opens)
        hipLaunchKernelGGL(lu3_close_kernel, dim3(lu3_grid(m - j0 - wd)), dim3(T_BS), 0, st, Y, ld, m, (int32_t)j0, wd, stt);
    }
    const int64_t c0 = jb + b, t = l - c0;
    if (t > 0) {                               // the register-resident path's own block update
      const int64_t mr = m - c0;
      const unsigned gu = (unsigned)((t + 63) / 64);
      const unsigned gr = (unsigned)((mr + 127) / 128);
      hipLaunchKernelGGL(lu_u12_kernel<64>, dim3(gu), dim3(256), 0, st, Y, ld, jb, jb, c0, l, u12);
      if (mr > 0) launch_rankk<64>(st, gr, Y, ld, m, c0, jb, c0, t, u12);
    }
  }
  int eb = (int)((l * l + 255) / 256);
  if (eb > 1024) eb = 1024;
  hipLaunchKernelGGL(lu2_extract_L_kernel, dim3(eb), dim3(256), 0, st, Y, ld, l);
}

// =====================================================================================================================
// Row-sharded form of the same factorization ("sharded alternative"): every rank keeps only its rows
// [row0, row0 + mloc) of the panel; per pivot step the ranks exchange one record each {local max |value|, its global
// row, that row, row j} (pipeline.cpp:lu_panel_sharded runs the collectives), everything else is row-local.  The
// arithmetic per element is the register-resident kernel's, operation for operation (same blocks of 64, leaves of 8,
// the same forward substitution for U12, fma(-l, u, a) in the same order, the same rank-64 MFMA update), so the
// result is bit-identical to the single-rank factorization -- tests/test_gpu_parity.py compares them on the GPU.
// Leaf columns live in HBM between the steps here (a step is host-sequenced around a collective, not a persistent launch).
// Record (doubles): [0] max |value| (-1: none), [1] global row (as a double), [2] 1.0 if this rank holds row j,
//                   [4, 4 + l) the candidate row, [4 + l, 4 + 2 l) row j.
// =====================================================================================================================
namespace {
constexpr int LUS_HDR = 4;

__global__ __launch_bounds__(256) void lus_cand_partial_kernel(const double* __restrict__ Y, int64_t ld, int64_t mloc,
                                                               int64_t row0, int64_t j, double* __restrict__ pval,
                                                               int64_t* __restrict__ pidx) {
  __shared__ double s_v[4];
  __shared__ int32_t s_i[4];
  double best = -1.0;
  int32_t besti = -1;
  for (int64_t li = (int64_t)blockIdx.x * 256 + threadIdx.x; li < mloc; li += (int64_t)gridDim.x * 256) {
    const int64_t gi = row0 + li;
    if (gi >= j) {
      const double av = fabs(Y[li + j * ld]);
      if (av > best) { best = av; besti = (int32_t)gi; }     // ascending rows per thread: the first maximum stays
    }
  }
  wave_argmax(best, besti);
  if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = best; s_i[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x < 64) {
    double v = (threadIdx.x < 4) ? s_v[threadIdx.x] : -1.0;
    int32_t i = (threadIdx.x < 4) ? s_i[threadIdx.x] : -1;
    wave_argmax8(v, i);
    if (threadIdx.x == 0) { pval[blockIdx.x] = v; pidx[blockIdx.x] = i; }
  }
}
__global__ __launch_bounds__(256) void lus_cand_final_kernel(const double* __restrict__ Y, int64_t ld, int64_t mloc,
                                                             int64_t row0, int64_t l, int64_t j, int nparts,
                                                             const double* __restrict__ pval, const int64_t* __restrict__ pidx,
                                                             double* __restrict__ rec) {
  __shared__ double s_v[4];
  __shared__ int32_t s_i[4];
  __shared__ int32_t s_win;
  double best = -1.0;
  int32_t besti = -1;
  for (int p = threadIdx.x; p < nparts; p += 256) {
    const double v = pval[p];
    const int32_t i = (int32_t)pidx[p];
    if (v > best || (v == best && (uint32_t)i < (uint32_t)besti)) { best = v; besti = i; }
  }
  wave_argmax(best, besti);
  if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = best; s_i[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x < 64) {
    double v = (threadIdx.x < 4) ? s_v[threadIdx.x] : -1.0;
    int32_t i = (threadIdx.x < 4) ? s_i[threadIdx.x] : -1;
    wave_argmax8(v, i);
    if (threadIdx.x == 0) {
      s_win = i;
      rec[0] = v;
      rec[1] = (double)i;
      rec[2] = (j >= row0 && j < row0 + mloc) ? 1.0 : 0.0;
      rec[3] = 0.0;
    }
  }
  __syncthreads();
  const int64_t wi = s_win;
  const bool has_j = (j >= row0 && j < row0 + mloc);
  for (int64_t c = threadIdx.x; c < l; c += 256) {
    rec[LUS_HDR + c] = (wi >= 0) ? Y[(wi - row0) + c * ld] : 0.0;
    rec[LUS_HDR + l + c] = has_j ? Y[(j - row0) + c * ld] : 0.0;
  }
}

// every workgroup reduces the ranks' records in rank order (same result everywhere), then: the rank that holds row j
// receives the pivot row there, the rank that holds row r the old row j, rows below j take the rank-1 update of the leaf
__global__ __launch_bounds__(256) void lus_apply_kernel(double* __restrict__ Y, int64_t ld, int64_t mloc, int64_t row0,
                                                        int64_t m, int64_t l, int64_t j0, int s, int w,
                                                        const double* __restrict__ recs, int nranks,
                                                        int32_t* __restrict__ ipiv, int32_t* __restrict__ info,
                                                        double* __restrict__ pval, int64_t* __restrict__ pidx) {
  // pval / pidx != null: this launch also leaves the per-workgroup arg-max partials of the NEXT leaf column (s + 1, over the
  // values it has just updated) where lus_cand_final_kernel expects them -- one launch less per pivot step
  __shared__ double s_v4[4];
  __shared__ int32_t s_i4[4];
  __shared__ double s_u[LW], s_old[LW];
  __shared__ int32_t s_r;
  __shared__ int s_gw, s_go;
  __shared__ double s_bestv;
  const int64_t j = j0 + s;
  const int64_t reclen = LUS_HDR + 2 * l;
  if (threadIdx.x == 0) {
    double best = -1.0;
    int32_t besti = -1;
    int gw = -1, go = -1;
    for (int g = 0; g < nranks; ++g) {
      const double v = recs[g * reclen + 0];
      const int32_t i = (int32_t)recs[g * reclen + 1];
      if (i >= 0 && (v > best || (v == best && (uint32_t)i < (uint32_t)besti))) { best = v; besti = i; gw = g; }
      if (recs[g * reclen + 2] != 0.0) go = g;
    }
    const bool valid = (besti >= j && besti < m && gw >= 0);
    s_r = valid ? besti : (int32_t)j;
    s_gw = valid ? gw : go;
    s_go = go;
    s_bestv = best;
  }
  __syncthreads();
  const int32_t r = s_r;
  const double* prow = recs + (int64_t)s_gw * reclen + ((s_gw == s_go && r == j) ? LUS_HDR + l : LUS_HDR);   // the pivot row
  const double* orow = recs + (int64_t)s_go * reclen + LUS_HDR + l;                                          // the old row j
  if (threadIdx.x < LW) {
    s_u[threadIdx.x] = (threadIdx.x < w) ? prow[j0 + threadIdx.x] : 0.0;
    s_old[threadIdx.x] = (threadIdx.x < w) ? orow[j0 + threadIdx.x] : 0.0;
  }
  __syncthreads();
  const double piv = s_u[s];
  const double rpiv = (piv != 0.0) ? 1.0 / piv : 0.0;
  const bool has_j = (j >= row0 && j < row0 + mloc), has_r = (r >= row0 && r < row0 + mloc);
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {                      // every rank keeps the whole pivot sequence
      if (ipiv != nullptr) ipiv[j] = r;
      if (!(s_bestv > 0.0)) atomicCAS(info, 0, (int32_t)(j + 1));
    }
    if (r != j) {
      for (int64_t c = threadIdx.x; c < l; c += 256) {
        const bool leafcol = (c >= j0 && c < j0 + w);
        if (has_j) Y[(j - row0) + c * ld] = prow[c];                 // the pivot row moves up (all columns)
        if (has_r && !leafcol) Y[(r - row0) + c * ld] = orow[c];     // the old row j moves down (its leaf part below)
      }
    }
  }
  double nbest = -1.0;
  int32_t nbesti = -1;
  for (int64_t li = (int64_t)blockIdx.x * 256 + threadIdx.x; li < mloc; li += (int64_t)gridDim.x * 256) {
    const int64_t gi = row0 + li;
    if (gi <= j) continue;
    double* row = Y + li + j0 * ld;
    const bool moved = (gi == r);
    double x[LW];
#pragma unroll
    for (int k = 0; k < LW; ++k) x[k] = moved ? s_old[k] : ((k >= s && k < w) ? row[k * ld] : 0.0);
    const double x0 = x[s];
    const double lij = (rpiv != 0.0) ? x0 * rpiv : x0;
    x[s] = lij;
#pragma unroll
    for (int k = 0; k < LW; ++k)
      if (k > s) x[k] -= lij * s_u[k];
#pragma unroll
    for (int k = 0; k < LW; ++k)
      if (k < w && (k >= s || moved)) row[k * ld] = x[k];
    if (pval != nullptr) {                        // candidate of column s + 1: ascending rows per thread, the first maximum stays
      double nv = 0.0;
#pragma unroll
      for (int k = 0; k < LW; ++k)
        if (k == s + 1) nv = fabs(x[k]);
      if (nv > nbest) { nbest = nv; nbesti = (int32_t)gi; }
    }
  }
  if (pval != nullptr) {                          // the reduction of lus_cand_partial_kernel, same order
    wave_argmax(nbest, nbesti);
    if ((threadIdx.x & 63) == 0) { s_v4[threadIdx.x >> 6] = nbest; s_i4[threadIdx.x >> 6] = nbesti; }
    __syncthreads();
    if (threadIdx.x < 64) {
      double v = (threadIdx.x < 4) ? s_v4[threadIdx.x] : -1.0;
      int32_t i = (threadIdx.x < 4) ? s_i4[threadIdx.x] : -1;
      wave_argmax8(v, i);
      if (threadIdx.x == 0) { pval[blockIdx.x] = v; pidx[blockIdx.x] = i; }
    }
  }
}

// rows [jb, j0) of the leaf columns -> U12 = L11^-1 A12 (kp x 8, [c * 8 + v]); the rank that holds those rows
__global__ __launch_bounds__(512) void lus_u12_leaf_kernel(const double* __restrict__ Y, int64_t ld, int64_t jb_local,
                                                           int kp, int64_t j0, int w, double* __restrict__ U12) {
  __shared__ double Ls[KPMAX * LSP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < kp * kp; e += 512) {
    const int r = e % kp, c = e / kp;
    Ls[r * LSP + c] = Y[(jb_local + r) + (int64_t)(j0 - kp + c) * ld];     // rows local, columns global jb .. j0
  }
  __syncthreads();
  for (int v = wave; v < LW; v += 8) {
    double x = (lane < kp && v < w) ? Y[(jb_local + lane) + (int64_t)(j0 + v) * ld] : 0.0;
    for (int cp = 0; cp < kp; ++cp) {
      const double xc = readlane_d(x, __builtin_amdgcn_readfirstlane(cp));
      if (lane > cp && lane < kp) x -= Ls[lane * LSP + cp] * xc;
    }
    if (lane < kp) U12[lane * LW + v] = x;
  }
}
__global__ __launch_bounds__(256) void lus_pending_kernel(double* __restrict__ Y, int64_t ld, int64_t mloc, int64_t row0,
                                                          int64_t jb, int64_t j0, int w, const double* __restrict__ U12) {
  __shared__ double Us[KPMAX * LW];
  const int kp = (int)(j0 - jb);
  for (int e = threadIdx.x; e < kp * LW; e += 256) Us[e] = U12[e];
  __syncthreads();
  for (int64_t li = (int64_t)blockIdx.x * 256 + threadIdx.x; li < mloc; li += (int64_t)gridDim.x * 256) {
    if (row0 + li < j0) continue;
    double a[LW];
#pragma unroll
    for (int k = 0; k < LW; ++k) a[k] = (k < w) ? Y[li + (j0 + k) * ld] : 0.0;
    for (int c = 0; c < kp; ++c) {
      const double lv = Y[li + (jb + c) * ld];
#pragma unroll
      for (int k = 0; k < LW; ++k) a[k] -= lv * Us[c * LW + k];
    }
#pragma unroll
    for (int k = 0; k < LW; ++k)
      if (k < w) Y[li + (j0 + k) * ld] = a[k];
  }
}
__global__ void lu_flag_export_kernel(const int32_t* __restrict__ info, double* __restrict__ flag) {
  flag[0] = (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) ? 1.0 : 0.0;
}
__global__ void lu_flag_import_kernel(int32_t* __restrict__ info, const double* __restrict__ flag) {
  if (flag[0] > 0.0 && __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) atomicExch(info, -1);
}
__global__ void lus_finish_kernel(double* __restrict__ Y, int64_t ld, int64_t mloc, int64_t row0, int64_t l) {
  const int64_t total = l * l;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % l, c = e / l;
    if (r < row0 || r >= row0 + mloc) continue;
    if (r == c) Y[(r - row0) + c * ld] = 1.0;
    else if (r < c) Y[(r - row0) + c * ld] = 0.0;
  }
}
}  // namespace

int lus_grid(int64_t mloc) {
  int64_t g = (mloc + 255) / 256;
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  return (int)g;
}
void lus_candidate(hipStream_t st, const double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t l, int64_t j, double* rec,
                   double* pval, int64_t* pidx, bool partials_ready) {
  const int g = lus_grid(mloc);
  if (!partials_ready) hipLaunchKernelGGL(lus_cand_partial_kernel, dim3(g), dim3(256), 0, st, Y, ld, mloc, row0, j, pval, pidx);
  hipLaunchKernelGGL(lus_cand_final_kernel, dim3(1), dim3(256), 0, st, Y, ld, mloc, row0, l, j, g, pval, pidx, rec);
}
// next_pval / next_pidx (may be null): leave the partials of leaf column s + 1 for the next lus_candidate
void lus_apply(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t m, int64_t l, int64_t j0, int s, int w,
               const double* recs, int nranks, int32_t* ipiv, int32_t* info, double* next_pval, int64_t* next_pidx) {
  hipLaunchKernelGGL(lus_apply_kernel, dim3(lus_grid(mloc)), dim3(256), 0, st, Y, ld, mloc, row0, m, l, j0, s, w, recs, nranks,
                     ipiv, info, next_pval, next_pidx);
}
void lus_u12_leaf(hipStream_t st, const double* Y, int64_t ld, int64_t row0, int64_t jb, int64_t j0, int w, double* U12) {
  hipLaunchKernelGGL(lus_u12_leaf_kernel, dim3(1), dim3(512), 0, st, Y, ld, jb - row0, (int)(j0 - jb), j0, w, U12);
}
void lus_pending(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t jb, int64_t j0, int w,
                 const double* U12) {
  hipLaunchKernelGGL(lus_pending_kernel, dim3(lus_grid(mloc)), dim3(256), 0, st, Y, ld, mloc, row0, jb, j0, w, U12);
}
void lus_u12_block(hipStream_t st, const double* Y, int64_t ld, int64_t row0, int64_t jb, int b, int64_t c0, int64_t c1,
                   double* U12) {
  const unsigned gu = (unsigned)((c1 - c0 + 63) / 64);
  if (b == 64) hipLaunchKernelGGL(lu_u12_kernel<64>, dim3(gu), dim3(256), 0, st, Y, ld, jb, jb - row0, c0, c1, U12);
  else hipLaunchKernelGGL(lu_u12_kernel<32>, dim3(gu), dim3(256), 0, st, Y, ld, jb, jb - row0, c0, c1, U12);
}
void lus_rankk(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t jb, int b, int64_t c0, int64_t t,
               const double* U12) {
  int64_t rbeg = c0 - row0;             // first local row below the block
  if (rbeg < 0) rbeg = 0;
  const int64_t mr = mloc - rbeg;
  if (mr <= 0) return;
  const unsigned gr = (unsigned)((mr + 127) / 128);
  if (b == 64) launch_rankk<64>(st, gr, Y, ld, mloc, rbeg, jb, c0, t, U12);
  else launch_rankk<32>(st, gr, Y, ld, mloc, rbeg, jb, c0, t, U12);
}
void lu_flag_export(hipStream_t st, const int32_t* info, double* flag) {
  hipLaunchKernelGGL(lu_flag_export_kernel, dim3(1), dim3(1), 0, st, info, flag);
}
void lu_flag_import(hipStream_t st, int32_t* info, const double* flag) {
  hipLaunchKernelGGL(lu_flag_import_kernel, dim3(1), dim3(1), 0, st, info, flag);
}
void lus_finish(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t l) {
  int eb = (int)((l * l + 255) / 256);
  if (eb > 1024) eb = 1024;
  hipLaunchKernelGGL(lus_finish_kernel, dim3(eb), dim3(256), 0, st, Y, ld, mloc, row0, l);
}

}}  // namespace gsi::hipk
