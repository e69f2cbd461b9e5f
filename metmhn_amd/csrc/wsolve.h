// Joint triangular solves in the WINDOW layout (round 4): a class-sorted matrix formulation with 14 - 15 of a patient's
// index bits on the chip instead of the 12 of a tile (kernels.h: k_psolve2).
//
// On the seed = 1 half of a paired patient's space the operator is a Kronecker sum, D - Q = A_R (+) A_C over the two
// tumour classes (reference: metmhn/jx/likelihood.py:231-262 solves it with k+1 Jacobi sweeps of
// kronvec.py:499-539).  R ("rows") is the class with more bits, C ("columns") the other one; a seeded state is
// (S, T) = (row subset, column subset) and
//     y[S,T] = (rhs + sum_{i in S} rR_i(S \ i) y[S \ i, T] + sum_{b in T} rC_b(T \ b) y[S, T \ b]) / (dR[S] + dC[T])
// (transposed: the sums run over the bits NOT in S / T, the neighbours are S | i / T | b, the rates are taken at
// the state itself).  One 1024-thread workgroup solves one patient:
//   * thread = row (w, l): the 6 lowest row bits are the LANE, the next 4 the WAVE of the thread;
//   * a thread works on one BLOCK of NC = 2^RB columns per step and keeps the last H = 2^HB blocks - a WINDOW of
//     H * NC columns - in registers: moves along the RB + HB lowest column bits are register arithmetic;
//   * everything above (the other column bits, then the row bits beyond the tenth: the EXTERNAL index Sigma) is
//     the thread's own earlier output, re-read from global memory: (K - 15) / 2 reads per state instead of 3.5;
//   * moves along the lane bits read the NEIGHBOUR LANE'S WINDOW REGISTERS directly (DPP / swizzle / bpermute, no
//     LDS storage): a lane of lane-level m (popcount of l) runs m whole windows behind, so the block it needs is
//     exactly what slot beta of the lower lane's window still holds from that lane's previous window pass;
//   * moves along the wave bits go through a two-slot ring in LDS: a wave of wave-level lam runs lam blocks behind,
//     its lower neighbour waves published the block one step earlier; one LDS-only barrier per step.
// Every wave therefore works on ONE static window slot per step (the step loop is unrolled H times), no lane
// ever waits for another lane of its wave, and no level ever idles except while the pipeline fills and drains.
// oracle/wschedule.py is a scalar model of exactly this schedule (tests/test_oracle_golden.py runs it).
//
// Storage of the seeded half ("window layout", wlayout.h):
//     position(S, T) = (((Sigma * H + beta) * 1024 + rho(w, l)) * NC + c,   T = c | beta << RB | Tx << (RB + HB),
//     Sigma = Tx | Sx << nXc,  S = l | w << 6 | Sx << 10,
// rho sorts the rows of a block by lane-level so that the lanes of a wave which work on the same external index
// (= the same lane-level) touch one contiguous run of memory.  The seed = 0 half keeps its natural positions.
// (Measured and dropped: a block as two planes of 16 bytes per row - no line requested by two instructions, but runs
// half as long: adjoint 18.6 -> 22.9 ms; a one-dword LDS-DMA touch of the next step's external rows: +2.5 / +6 ms.)
#pragma once
#include "kernels.h"
#include "wlayout.h"
#include "lanes.h"

namespace mmhn {

#ifndef MMHN_W_NTFAR
#define MMHN_W_NTFAR -1   // >= 0: external blocks of bit >= this value are loaded with the nt policy
#endif
// Round-5 switches, all measured and left off: none of them moves a launch by more than the noise - the
// launch moves its 3.84 units of a patient's seeded half at the rate HBM delivers, whatever the order inside a step
#ifndef MMHN_W_WPERM
#define MMHN_W_WPERM 0  // 1: the wave index of the rows comes from a permutation of the hardware wave id that gives each SIMD (hardware
                        // waves s, s + 4, s + 8, s + 12) the same number of ring moves (8 of the 32 of a step)
#endif
#ifndef MMHN_W_PRIO
#define MMHN_W_PRIO 0   // 1: issue priority of a wave = its wave-level (the wave with the most ring moves is the one a barrier waits for)
#endif
#ifndef MMHN_W_STAG
#define MMHN_W_STAG 0   // 1: the second external request of a step goes out half-way down the lane moves
#endif
// Timing-only ablations (WRONG results): MMHN_WABL_NOLOAD / _NOSTORE (no external block read / nothing written),
// _NOTAB (no table read in a step), _NOPERM (the two lane exchanges of the LDS crossbar as DPP moves), _NORING (no ring reads),
// _NOBAR (no barrier in the step loop)
#ifndef MMHN_W_CW
#define MMHN_W_CW 0     // 1: the rates of the four wave-bit moves kept in registers with the lane-bit rates (round 4: +4 ms with the
                        // spills of the time; re-measured in round 5 with 9 - 16 registers free)
#endif

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): LDS only, vector memory stays in flight
#ifndef MMHN_WABL_NOBAR
  __builtin_amdgcn_s_barrier();
#endif
}

template <typename T> __device__ __forceinline__ T fma_m(T a, T b, T c);
template <> __device__ __forceinline__ double fma_m<double>(double a, double b, double c) { return fma(a, b, c); }
template <> __device__ __forceinline__ float fma_m<float>(float a, float b, float c) { return fmaf(a, b, c); }

template <int I> struct IC { static constexpr int value = I; };

template <typename T> inline bool wtab_fits(int kR, int kC);
template <typename T>
inline bool window_ok(const Desc& d) {
  if (!WCfg<T>::ON || d.mode != JOINT || d.seedbit != d.k - 1) return false;
  const int kP = popc(d.maskP), kM = popc(d.maskM);
  const int kR = kP >= kM ? kP : kM, kC = kP >= kM ? kM : kP;
  return kR >= WTB && kR <= WCfg<T>::KR && kC >= WCfg<T>::RB + WCfg<T>::HB && kC <= WCfg<T>::KC && popc(d.pairP) <= WCfg<T>::KE &&
         wtab_fits<T>(kR, kC);
}
template <typename T>
inline WDesc make_wdesc(const Desc& d, int prob) {
  WDesc w{};
  w.prob = prob;
  const int kP = popc(d.maskP), kM = popc(d.maskM);
  w.majP = kP >= kM ? 1 : 0;
  w.rowmask = w.majP ? d.maskP : d.maskM;
  w.colmask = w.majP ? d.maskM : d.maskP;
  int nr = 0, nc = 0;
  for (int b = 0; b < d.k; ++b) {
    if ((w.rowmask >> b) & 1u) w.rb[nr++] = (int8_t)b;
    else if ((w.colmask >> b) & 1u) w.cb[nc++] = (int8_t)b;
  }
  w.kR = nr; w.kC = nc;
  w.nXc = nc - WCfg<T>::RB - WCfg<T>::HB;
  w.nXr = nr - WTB;
  for (int i = 0; i < nr; ++i) {
    const int b = w.rb[i];
    w.prt[i] = -1;
    if ((d.lone >> b) & 1u) { w.loneRowC |= 1u << i; continue; }
    const int pb = w.majP ? b + 1 : b - 1;                     // partner slot: the M bit sits right above its P bit
    w.pairRowC |= 1u << i;
    for (int j = 0; j < nc; ++j) if (w.cb[j] == pb) w.prt[i] = (int8_t)j;
  }
  return w;
}

// diagnostic (mmhn_debug_lane_moves): out[I * 64 + lane] = lane id received through lane_nbr<I>
template <bool TR>
__global__ void k_lane_moves(int* out) {
  const int lane = threadIdx.x & 63;
  out[0 * 64 + lane] = lane_nbr32<0, TR>(lane, lane);
  out[1 * 64 + lane] = lane_nbr32<1, TR>(lane, lane);
  out[2 * 64 + lane] = lane_nbr32<2, TR>(lane, lane);
  out[3 * 64 + lane] = lane_nbr32<3, TR>(lane, lane);
  out[4 * 64 + lane] = lane_nbr32<4, TR>(lane, lane);
  out[5 * 64 + lane] = lane_nbr32<5, TR>(lane, lane);
}

// A CHAIN is a run of consecutive entries of the (shape-sorted) WDesc list that have the same shape (kR, kC) and whose state
// vectors lie back to back in memory: one workgroup streams through it without draining its pipeline between patients -
// the lanes of low lane-level start on patient j + 1 while the lanes of high lane-level still finish patient j (a lane of
// lane-level m works on window pass sig - m of the chain, pass = patient * 2^nX + external index).
struct WChain { int start, count; };

template <typename T>
struct WPInfo {                      // what a lane needs to know about the patient it is working on (two live per workgroup)
  long long droff;                   // element offset (in the table buffer) of the row-class diagonal table
  long long soff[2];                 // transposed: start of the upper half of q_S of the row part / column part, -1: none
  T cst[2];                          //             its constant (JLink)
  uint32_t pairRowC, loneRowC;       // forward: compact row bits with / without a partner slot
  int prt[WMAXB];                    //          column bit of the partner of row bit i
};

// offsets (elements) of one patient's tables in LDS.  Full tables (fp64): compile-time constants, sized for the largest
// shape.  Factored tables (fp32): packed per shape (a chain has one shape) - twelve column bits need a big diagonal table,
// eighteen row bits a big external-row table, never both (window_ok checks the total).
struct WOff { int oFx, XS, oDC, oLr, oUr, oEr, ES, oSe, size; };

template <typename T>
struct WLds {
  using C = WCfg<T>;
  static constexpr int NC = 1 << C::RB, WB = C::RB + C::HB, WIN = 1 << WB, NXCM = C::KC - WB, WKR = C::KR, MKE = C::KE;
  // rates of the column-class events, only for the column sets that do not contain the event (half of them):
  //   bit b < WB (inside a window): [external column setting Tx][window setting with bit b squeezed out] - SZLO entries
  //   bit b >= WB (external):       [Tx with bit b - WB squeezed out][window setting]                     - SZHI entries
  static constexpr int LOS = WIN / 2 + C::PAD, HIS = WIN + C::PAD;     // row strides (the lanes of a wave differ in Tx: banks)
  static constexpr int SZLO = (1 << NXCM) * LOS, SZHI = (NXCM > 0 ? (1 << (NXCM - 1)) : 1) * HIS;
  // FACT: rate = Fw[event][window setting] * Fx[event][external column setting] (the event's own bit is skipped in both)
  static constexpr int oRh = 0;
  static constexpr int oFw = oRh, oFx = oFw + C::KC * WIN;
  static constexpr int oDC = oRh + WB * SZLO + NXCM * SZHI;            // [Tx][window setting] column part of the diagonal
  static constexpr int oLr = oDC + (1 << NXCM) * HIS;                  // [WKR][64] row-bit rate: product over the lane bits
  static constexpr int oUr = oLr + WKR * 64;                           // [WKR][16] ... base rate and the wave bits
  static constexpr int oEr = oUr + WKR * 16;                           // [WKR][32] ... the external row bits
  static constexpr int oSe = oEr + WKR * 32;                           // [2^ke] forward: seeding inflow of eq state e
  static constexpr int ring = 0;                                        // [2][NC * sizeof(T) / 16][WROWS] 16-byte pieces
  static constexpr int tab0 = ring + 2 * NC * WROWS;                    // two patients' tables
  static constexpr int LDSMAX = 160 * 1024;
  static constexpr int TABSZ = C::FACT ? ((LDSMAX - 64 - 2 * (int)sizeof(WPInfo<T>)) / (int)sizeof(T) - tab0) / 2 / 4 * 4
                                       : (oSe + (1 << MKE) + 3) / 4 * 4;  // one patient's tables
  static constexpr int end = tab0 + 2 * TABSZ;
  // while a patient enters or leaves the pipeline the ring is dead: the effect table thc [k][k] and the eq-block
  // solution e0 [2^ke] live there
  static constexpr int thc = ring, e0 = ring + (MAXK * MAXK + 3) / 4 * 4;
  static constexpr size_t bytes = (size_t)end * sizeof(T) + 2 * sizeof(WPInfo<T>) + 64;
  static_assert(e0 + (1 << MKE) <= tab0, "event scratch fits the ring");
  static_assert(bytes <= (size_t)LDSMAX, "two patients' tables and the ring fit the LDS of a CU");
  __host__ __device__ static WOff offsets(int kR, int kC) {
    WOff o;
    if (C::FACT) {
      const int nXc = kC - WB, nXr = kR - WTB;
      o.oFx = oFw + kC * WIN; o.XS = 1 << nXc;
      o.oDC = o.oFx + (kC << nXc);
      o.oLr = o.oDC + (1 << nXc) * HIS;
      o.oUr = o.oLr + kR * 64;
      o.oEr = o.oUr + kR * 16; o.ES = 1 << nXr;
      o.oSe = o.oEr + (kR << nXr);
      o.size = o.oSe + (1 << MKE);
    } else {
      o.oFx = oFx; o.XS = 1 << NXCM; o.oDC = oDC; o.oLr = oLr; o.oUr = oUr; o.oEr = oEr; o.ES = 32; o.oSe = oSe; o.size = TABSZ;
    }
    return o;
  }
};
template <typename T>
inline bool wtab_fits(int kR, int kC) { return WLds<T>::offsets(kR, kC).size <= WLds<T>::TABSZ; }
template <typename T>
constexpr size_t wsolve_lds() { return WLds<T>::bytes; }

// ------------------------------------------------------------------------------------
// k_wsolve: y = (D - Q)^-1 rhs (TR: transposed) of the joint problems in `wds`, window layout.
//   forward: rhs = E0 e_0 (the seed = 0 lattice over the paired events is solved first, seeding carries it into
//            the seeded half);  transposed: rhs = D_obs * scatter(q_S) from `links` (likelihood.py:573-575,
//            617-618), the seed = 0 lattice follows the seeded half.
// Persistent: workgroup b takes chains b, b + gridDim.x, ...
// ------------------------------------------------------------------------------------
// NXT >= 0: the number of external bits of every chain of the launch, as a compile-time constant (the step loop then has no
// scalar branches on it); NXT < 0: read per chain
// SPLIT (NXT >= 4): TWO workgroups per patient (review item 2 of round 4 in its cheapest form).  Blocks b and
// b + 8 - the same XCD under the round-robin placement of workgroups - are a pair that walks the same chains: role 0 ("A") takes the
// half of every patient's passes it does not need the partner for (forward: top bit of the external index clear; transposed: set),
// role 1 ("B") the other half, whose move along the top bit reads what A wrote.  A lane of B at pass g needs the SAME lane of A at
// pass g, i.e. A's iteration of the same number: A drains its stores at the end of every iteration of the pass loop and its last wave
// raises the pair's progress word (write-through stores, relaxed agent-scope flag: the protocol of k_csolve, tsolve.h); every wave of B
// polls that word in front of an iteration (bounded; `abort_w` / `h_abort` as in k_csolve), one agent-scope acquire, plain loads.
// Every workgroup sets the patient's tables up; the transposed solve's seed = 0 lattice is B's.
constexpr unsigned WSPLIT_SPIN_LIMIT = 1u << 22;
template <typename T, bool TR, int NXT = -1, bool SPLIT = false>
__global__ __launch_bounds__(WROWS) void k_wsolve(const Desc* __restrict__ descs, const WDesc* __restrict__ wds,
                                                  const WChain* __restrict__ chains, int nchains,
                                                  T* y, const T* __restrict__ tab,
                                                  const JLink<T>* __restrict__ links, const T* __restrict__ qS,
                                                  unsigned* wprog, unsigned wbase, unsigned* abort_w, unsigned* h_abort) {
  static_assert(!SPLIT || NXT >= 4, "two workgroups per patient: a compile-time number of external bits, 2^(NXT - 1) > 6 passes per half");
  using C = WCfg<T>;
  using L = WLds<T>;
  constexpr int RB = C::RB, HB = C::HB, NC = 1 << RB, H = 1 << HB, WB = RB + HB, WIN = 1 << WB;
  constexpr int LOS = L::LOS, HIS = L::HIS, SZLO = L::SZLO, SZHI = L::SZHI;
  constexpr int QE = 16 / (int)sizeof(T);                      // elements of a 16-byte piece
  constexpr int NQ = NC / QE;                                  // pieces of a block row (2)
  typedef T VecT __attribute__((ext_vector_type(NC)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  struct Raw { u32x4 q[NQ]; };
  static_assert(sizeof(Raw) == sizeof(VecT) && NQ == 2, "a row of a block is two 16-byte accesses");
  constexpr uint32_t BLKB = WROWS * sizeof(VecT);              // bytes of one block
  constexpr int BSH = HB + WTB + 5;                            // log2 of the bytes of one external index (H blocks)
  static_assert(sizeof(VecT) == 32, "block rows are 32 bytes");
  constexpr uint32_t OOB = 0x80000000u;
  extern __shared__ __align__(16) unsigned char smem[];
  T* const lds = reinterpret_cast<T*>(smem);
  T* const ring = lds + L::ring;
  T* const thc = lds + L::thc;
  T* const e0 = lds + L::e0;
  WPInfo<T>* const pinfo = reinterpret_cast<WPInfo<T>*>(lds + L::end);
#if MMHN_W_WPERM
  // hardware wave -> wave index of its rows (nibble h of the constant, from the top)
  const int wv = __builtin_amdgcn_readfirstlane((int)((0x0123FEDC5647A9B8ull >> (60 - 4 * ((int)threadIdx.x >> 6))) & 15ull));
  const int tid = (wv << 6) | ((int)threadIdx.x & 63);
#else
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int lam = TR ? WWB - __popc(wv) : __popc(wv);          // wave-level: blocks this wave runs behind
#if MMHN_W_PRIO
  if (lam >= 3) __builtin_amdgcn_s_setprio(3);
  else if (lam == 2) __builtin_amdgcn_s_setprio(2);
  else if (lam == 1) __builtin_amdgcn_s_setprio(1);
#endif
  // (thread-derived values are re-derived from an opaque copy of the thread id inside every step / pass: hipcc would
  // otherwise hoist a dozen loop-invariant LDS addresses out of the step loop and spill them)
#if MMHN_W_WPERM
  auto opaque_tid = [&]() -> uint32_t { uint32_t t = (uint32_t)threadIdx.x & 63u; asm volatile("" : "+v"(t)); return t | ((uint32_t)wv << 6); };
#else
  auto opaque_tid = [&]() -> uint32_t { uint32_t t = (uint32_t)threadIdx.x; asm volatile("" : "+v"(t)); return t; };
#endif
  // a column set that does not contain bit b, with bit b squeezed out
  auto squeeze = [](uint32_t v, int b) -> uint32_t { return ((v >> (b + 1)) << b) | (v & ((1u << b) - 1u)); };
  auto deskew = [&]() { for (int s = lam; s < WWB; ++s) lds_barrier(); };
  auto reskew = [&]() { for (int s = 0; s < lam; ++s) lds_barrier(); };

  // (formed once: inside the chain loop the addresses of wrho's tables were kept over the step loop and spilled)
  const uint32_t voff = wrho((uint32_t)tid >> 6, (uint32_t)tid & 63u) * (uint32_t)sizeof(VecT);
  STAMP_DECL;
  // SPLIT: pair and role of this workgroup; cum = iterations of the pass loop the pair has behind it in this launch
  const int role = SPLIT ? (int)((blockIdx.x >> 3) & 1u) : 0;
  const int cstart = SPLIT ? (int)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7u)) : (int)blockIdx.x;
  const int cstep = SPLIT ? (int)(gridDim.x >> 1) : (int)gridDim.x;
  unsigned* const prog = SPLIT ? wprog + cstart : nullptr;
  unsigned cum = 0;
  // B: wait until A has `need` iterations behind it (every wave by itself: no barrier; one acquire)
  auto wait_partner = [&](unsigned need) {
    const unsigned want = wbase + need;
    unsigned spins = 0;
    for (;;) {
      const unsigned v = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(v - want) >= 0) break;
      if ((++spins & 15u) == 0u) {
        if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (spins > WSPLIT_SPIN_LIMIT) {
          __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(h_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
      __builtin_amdgcn_s_sleep(8);
    }
#ifndef MMHN_WSPLIT_NOACQ   // (experiment, timing only)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  };
  for (int ci = cstart; ci < nchains; ci += cstep) {
    STAMP_START;
    const int first = sgpr(chains[ci].start), npat = sgpr(chains[ci].count);
    if (npat == 0) continue;                                   // (a filler entry of the host's chain deal)
    const WDesc& w0 = wds[first];
    const Desc& d0 = descs[w0.prob];
    const int k = sgpr(d0.k), kR = sgpr(w0.kR), kC = sgpr(w0.kC), nXc = sgpr(w0.nXc), nXr = sgpr(w0.nXr);
    const int nX = NXT >= 0 ? NXT : nXc + nXr;
    constexpr bool EV2 = TR && NXT == 5;
    constexpr bool DEEP = NXT >= 7;                            // (at five external bits - k = 20 - measured slower: 13.4 against 13.1 ms)
    // table offsets: constants with the full tables, per shape with the factored ones
    const WOff wo = L::offsets(kR, kC);
    const int oFx = C::FACT ? sgpr(wo.oFx) : L::oFx, XS = C::FACT ? sgpr(wo.XS) : (1 << L::NXCM);
    const int oDC = C::FACT ? sgpr(wo.oDC) : L::oDC, oLr = C::FACT ? sgpr(wo.oLr) : L::oLr, oUr = C::FACT ? sgpr(wo.oUr) : L::oUr;
    const int oEr = C::FACT ? sgpr(wo.oEr) : L::oEr, ES = C::FACT ? sgpr(wo.ES) : 32, oSe = C::FACT ? sgpr(wo.oSe) : L::oSe;
    (void)oFx; (void)XS;
    const int seedb = k - 1;
    const long long half = 1ll << (k - 1);
    const long long ybase = sgpr64(d0.off);                    // patient j of the chain: y + ybase + (j << k)
    const uint32_t NXS = 1u << nX, mXc = (1u << nXc) - 1u;
    const int nXw = SPLIT ? nX - 1 : nX;                       // log2 of the passes of a patient THIS workgroup makes
    const uint32_t NXW = 1u << nXw;
    const uint32_t Sxfull = (1u << nXr) - 1u, Txfull = mXc;
    const uint32_t PATB = (uint32_t)(sizeof(T) << k), HALFB = PATB >> 1;   // bytes of a patient's vector / of its seeded half
    const int NPASS = npat * (int)NXW + WLB;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(sgpr64((long long)reinterpret_cast<char*>(y + ybase))), 0,
                                                        (int)sgpr((uint32_t)npat * PATB), 0x00020000);
    // ---- a patient enters: its tables (buffer j & 1), forward: its seed = 0 lattice (only PT == MT states carry values).
    // Every wave is at the same point here (deskew); the ring is dead.
    auto solve_eq = [&](const WDesc& wd, const Desc& d, T* se) {
      const int tid = (int)opaque_tid();                       // (per-thread addresses formed here, not hoisted over the step loop and spilled)
      const long long base = d.off;
      const T* dP = tab + d.toff + rate_table_size(k);
      const T* dE = dP + (1ll << __popc(d.maskP)) + (1ll << __popc(d.maskM));
      const T* ym = y + base + half;
      const uint32_t pairP = sgpr(d.pairP);
      const int ke = __popc(pairP);
      const uint32_t VE = 1u << ke;
      const T seed_base = thc[seedb * k + seedb];
      for (int s = 0; s <= ke; ++s) {
        const int level = TR ? ke - s : s;
        for (uint32_t e = tid; e < VE; e += WROWS) {
          if (__popc(e) != level) continue;
          const uint32_t xp = pdep32(e, pairP);
          const uint32_t x0 = xp | (xp << 1);
          T z = (!TR && e == 0) ? e0_scale<T>() : T(0);
          T rs = seed_base;
          for (uint32_t m2 = xp; m2; m2 &= m2 - 1) rs *= thc[seedb * k + (__ffs(m2) - 1)];
          if (!TR) {
            for (uint32_t mm = xp; mm; mm &= mm - 1) {
              const int bP = __ffs(mm) - 1;
              T r = thc[bP * k + bP];
              for (uint32_t m2 = xp & ~(1u << bP); m2; m2 &= m2 - 1) r *= thc[bP * k + (__ffs(m2) - 1)];
              z += r * e0[pext32(xp & ~(1u << bP), pairP)];
            }
          } else {
            for (uint32_t mm = pairP & ~xp; mm; mm &= mm - 1) {
              const int bP = __ffs(mm) - 1;
              T r = thc[bP * k + bP];
              for (uint32_t m2 = xp; m2; m2 &= m2 - 1) r *= thc[bP * k + (__ffs(m2) - 1)];
              z += r * e0[pext32(xp | (1u << bP), pairP)];
            }
            z += rs * ym[wpos_nat<T>(wd, x0)];
          }
          const T v = z / dE[e];
          e0[e] = v;
          if (!TR) se[e] = rs * v;
          y[base + x0] = v;
        }
        __syncthreads();
      }
    };
    auto load_thc = [&](const Desc& d) {
      const int tid = (int)opaque_tid();
      const T* src = tab + d.toff;
      for (int e = tid; e < k * k; e += WROWS) thc[e] = src[e];
    };
    auto enter = [&](int j) {
      const int tid = (int)opaque_tid();
      const WDesc& wd = wds[first + j];
      const Desc& d = descs[wd.prob];
      T* const tb = lds + L::tab0 + (j & 1) * L::TABSZ;
      const bool majP = wd.majP != 0;
      const T* dP = tab + d.toff + rate_table_size(k);
      const T* dM = dP + (1ll << __popc(d.maskP));
      const T* dCg = majP ? dM : dP;
      __syncthreads();                                         // every wave has left the steps before: ring and buffer j & 1 are free
      load_thc(d);
      __syncthreads();
      if constexpr (C::FACT) {
        for (uint32_t e = tid; e < ((uint32_t)kC << WB); e += WROWS) {
          const int b = (int)(e >> WB), nb = wd.cb[b];
          const uint32_t u = e & (uint32_t)(WIN - 1);
          T r = thc[nb * k + nb];
          for (int i = 0; i < WB; ++i) if (i != b && ((u >> i) & 1u)) r *= thc[nb * k + wd.cb[i]];
          tb[L::oFw + e] = r;
        }
        for (uint32_t e = tid; e < ((uint32_t)kC << nXc); e += WROWS) {
          const int b = (int)(e >> nXc), nb = wd.cb[b];
          const uint32_t u = e & mXc;
          T r = T(1);
          for (int i = 0; i < nXc; ++i) if (WB + i != b && ((u >> i) & 1u)) r *= thc[nb * k + wd.cb[WB + i]];
          tb[oFx + b * XS + u] = r;
        }
      }
      // rates of the column-class events from the column sets without the event
      for (uint32_t e = tid; !C::FACT && e < ((uint32_t)kC << (kC - 1)); e += WROWS) {
        const int b = (int)(e >> (kC - 1)), nb = wd.cb[b];
        const uint32_t u = e & ((1u << (kC - 1)) - 1u);
        const uint32_t Tc = ((u >> b) << (b + 1)) | (u & ((1u << b) - 1u));      // the column set (bit b clear)
        T r = thc[nb * k + nb];
        for (int i = 0; i < kC; ++i) if ((Tc >> i) & 1u) r *= thc[nb * k + wd.cb[i]];
        uint32_t idx;
        if (b < WB) idx = b * SZLO + (Tc >> WB) * LOS + squeeze(Tc & (uint32_t)(WIN - 1), b);
        else idx = WB * SZLO + (b - WB) * SZHI + squeeze(Tc >> WB, b - WB) * HIS + (Tc & (uint32_t)(WIN - 1));
        tb[L::oRh + idx] = r;
      }
      for (uint32_t e = tid; e < (1u << kC); e += WROWS) tb[oDC + (e >> WB) * HIS + (e & (uint32_t)(WIN - 1))] = dCg[e];
      for (int e = tid; e < kR * 64; e += WROWS) {
        const int i = e >> 6, l = e & 63, nb = wd.rb[i];
        T r = T(1);
        for (int q = 0; q < WLB; ++q) if (q != i && ((l >> q) & 1)) r *= thc[nb * k + wd.rb[q]];
        tb[oLr + e] = r;
      }
      for (int e = tid; e < kR * 16; e += WROWS) {
        const int i = e >> 4, u = e & 15, nb = wd.rb[i];
        T r = thc[nb * k + nb];
        for (int q = 0; q < WWB; ++q) if (WLB + q != i && ((u >> q) & 1)) r *= thc[nb * k + wd.rb[WLB + q]];
        tb[oUr + e] = r;
      }
      for (int e = tid; e < kR * ES; e += WROWS) {
        const int i = C::FACT ? e >> nXr : e >> 5, u = e & (ES - 1), nb = wd.rb[i];
        T r = T(1);
        for (int q = 0; q < nXr; ++q) if (WTB + q != i && ((u >> q) & 1)) r *= thc[nb * k + wd.rb[WTB + q]];
        tb[oEr + e] = r;
      }
      if (tid == 0) {
        WPInfo<T>& pi_ = pinfo[j & 1];
        pi_.droff = d.toff + rate_table_size(k) + (majP ? 0 : (1ll << __popc(d.maskP)));
        const JLink<T>& Lk = links[wd.prob];
        const int rowpart = majP ? 0 : 1;
        long long none = -1;
        asm volatile("" : "+v"(none));                         // (not a constant to be parked in registers over the pass loop)
        for (int q = 0; q < 2; ++q) {
          const int part = q == 0 ? rowpart : 1 - rowpart;
          pi_.soff[q] = (TR && Lk.soff[part] >= 0) ? Lk.soff[part] + (1ll << (Lk.sk[part] - 1)) : none;
          pi_.cst[q] = TR ? Lk.cst[part] : T(0);
        }
        pi_.pairRowC = wd.pairRowC; pi_.loneRowC = wd.loneRowC;
        for (int i = 0; i < WMAXB; ++i) pi_.prt[i] = wd.prt[i] < 0 ? 0 : wd.prt[i];
      }
      if (!TR) solve_eq(wd, d, tb + oSe);
      __syncthreads();
    };
    auto leave = [&](int j) {                                  // transposed: the seed = 0 lattice of a patient whose seeded half is complete
      const WDesc& wd = wds[first + j];
      const Desc& d = descs[wd.prob];
      __syncthreads();                                         // (waits for the stores of every wave as well)
      load_thc(d);
      __syncthreads();
      solve_eq(wd, d, nullptr);
      __syncthreads();
    };
    auto ld_row = [&](uint32_t off, uint32_t soff) -> VecT {
      Raw r;
      r.q[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, (int)soff, 0);
      r.q[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off + 16, (int)soff, 0);
      return __builtin_bit_cast(VecT, r);
    };
    auto st_row = [&](uint32_t off, uint32_t soff, const VecT& v) {
      const Raw r = __builtin_bit_cast(Raw, v);
#ifdef MMHN_WSPLIT_PLAIN   // (experiment: plain stores - only valid while both workgroups of a pair sit on one XCD)
      constexpr int WT = 0;
#else
      constexpr int WT = SPLIT ? 16 : 0;                        // (16: sc1, write-through)
#endif
      __builtin_amdgcn_raw_buffer_store_b128(r.q[0], rsrc, (int)off, (int)soff, WT);
      __builtin_amdgcn_raw_buffer_store_b128(r.q[1], rsrc, (int)off + 16, (int)soff, WT);
    };
    auto lds_vec = [&](const T* p) -> VecT {                   // NC consecutive elements, 16-byte aligned
#ifdef MMHN_WABL_NOTAB
      VecT cv;
      for (int c = 0; c < NC; ++c) cv[c] = T(0.25);
      return cv;
#endif
      Raw r;
      r.q[0] = *reinterpret_cast<const u32x4*>(p);
      r.q[1] = *reinterpret_cast<const u32x4*>(p + QE);
      return __builtin_bit_cast(VecT, r);
    };
    // the first patient enters (before the per-lane state of the passes exists: it would be kept in registers over the
    // set-up and spilled)
    enter(0);
    reskew();
    VecT Wd[H];                                                // the window
#pragma unroll
    for (int b = 0; b < H; ++b)
#pragma unroll
      for (int c = 0; c < NC; ++c) Wd[b][c] = T(0);
    // ---- state of the current window pass (per lane: the lanes of a wave differ in their patient and external index)
    // Sigma: external index (0 on a lane that is outside the pipeline), soff: byte offset of its blocks inside the chain
    // (beyond the buffer on such a lane: its stores are dropped, its loads return zeros), tbo: its patient's tables
    // (initial values through an opaque move: hipcc otherwise keeps the constants in registers across the chain loop and spills
    // them)
    auto opq = [](uint32_t v) -> uint32_t { asm volatile("" : "+v"(v)); return v; };
    uint32_t Sigma = 0, soff = opq(OOB), tbo = L::tab0;
    T cL[WLB], dRv = T(1);
#if MMHN_W_CW
    T cW[WWB];                                                 // rates of the four wave-bit moves (same row constants as cL)
#endif
    uint32_t hitT = opq(0xffffffffu), hitE = 0;
    uint32_t rowkey = opq(0xffffffffu);                        // (patient, external row setting) the row constants were formed for
    auto begin_pass = [&](int sig) {
      const uint32_t tt = opaque_tid(), ln = tt & 63u;
      const int V = sig - (TR ? WLB - __popc(ln) : __popc(ln));
      const bool act = (unsigned)V < (unsigned)npat * NXW;
      const uint32_t Vc = act ? (uint32_t)V : 0u;
      const uint32_t j = Vc >> nXw, g = (Vc & (NXW - 1u)) | (SPLIT ? (uint32_t)role << nXw : 0u);
      Sigma = TR ? NXS - 1u - g : g;
      const uint32_t Sx = Sigma >> nXc;
      tbo = L::tab0 + (j & 1u) * L::TABSZ;
      soff = act ? j * PATB + HALFB + (Sigma << BSH) + voff : OOB;
      // the rates along the lane bits, the row part of the diagonal and the seeding column depend on the row alone:
      // formed again only in a pass in which some lane of the wave moves to another patient or external row setting
      const uint32_t key = (j << 8) | Sx;
      if (__builtin_amdgcn_ballot_w64(key != rowkey) == 0ull) return;
      rowkey = key;
      const T* tb = lds + tbo;
      const WPInfo<T>& pi_ = pinfo[j & 1u];
#pragma unroll
      for (int i = 0; i < WLB; ++i) {
        const bool has = TR ? !((ln >> i) & 1u) : ((ln >> i) & 1u);
        const T r = tb[oLr + i * 64 + ln] * tb[oUr + i * 16 + wv] * tb[oEr + i * ES + Sx];
        cL[i] = has ? r : T(0);
      }
#if MMHN_W_CW
#pragma unroll
      for (int j = 0; j < WWB; ++j)
        cW[j] = tb[oLr + (WLB + j) * 64 + ln] * tb[oUr + (WLB + j) * 16 + wv] * tb[oEr + (WLB + j) * ES + Sx];
#endif
      dRv = tab[pi_.droff + (tt | (Sx << WTB))];
      if (!TR) {
        // forward right-hand side: seeding enters row S at the one column whose paired events are those of S
        const uint32_t S = tt | (Sx << WTB);
        const uint32_t pairRowC = pi_.pairRowC;
        uint32_t hT = 0, e = 0, ne = 0;
        for (int i = 0; i < kR; ++i) {
          const uint32_t isp = (pairRowC >> i) & 1u, on = isp & (S >> i);
          hT |= on << pi_.prt[i];
          e |= on << ne;
          ne += isp;
        }
        hitT = (S & pi_.loneRowC) ? 0xffffffffu : hT;
        hitE = e;
      }
    };
    // ---- one step: the block (Sigma, beta) of every row of the wave
    auto step = [&](auto BIc, int gpar) {
      constexpr int BI = decltype(BIc)::value;
      constexpr int beta = TR ? H - 1 - BI : BI;
      constexpr uint32_t boff = (uint32_t)beta * BLKB;
      const uint32_t tt = opaque_tid(), ln = tt & 63u;
      // (derived afresh in every step from opaque copies: common subexpressions of the eight steps of a pass would be
      // kept in registers over the whole pass and spilled)
      uint32_t Sgo = Sigma, tbx = tbo;
      asm volatile("" : "+v"(Sgo), "+v"(tbx));
      const T* tb = lds + tbx;
      const uint32_t Tx = Sgo & mXc, Sx = Sgo >> nXc;
      T acc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = T(0);
      // external moves: the thread's own earlier blocks.  A move that does not exist is requested beyond the end of
      // the buffer (zeros come back, nothing is touched): no per-move control flow.
      auto ext_off = [&](int j) -> uint32_t {
#ifdef MMHN_WABL_NOLOAD   // timing-only ablation (wrong results): no external block is read
        return OOB;
#endif
        const bool has = TR ? !((Sgo >> j) & 1u) : ((Sgo >> j) & 1u);
        return (has && soff != OOB) ? (soff ^ (1u << (j + BSH))) : OOB;
      };
      auto ext_take = [&](int j, const VecT& nv) {
        if (j < nXc) {
          // column move: rate of column bit WB + j from the source column set (bit j of Tx clear there)
          if constexpr (C::FACT) {
            const VecT rr = lds_vec(tb + L::oFw + (WB + j) * WIN + beta * NC);
            const T fx = tb[oFx + (WB + j) * XS + (Tx & ~(1u << j))];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fma_m(rr[c] * fx, nv[c], acc[c]);
          } else {
            const VecT rr = lds_vec(tb + L::oRh + WB * SZLO + j * SZHI + squeeze(Tx, j) * HIS + beta * NC);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fma_m(rr[c], nv[c], acc[c]);
          }
        } else {
          const int i = WTB + (j - nXc);
#ifdef MMHN_WABL_NOTAB
          const T r = T(0.125);
#else
          const T r = tb[oLr + i * 64 + ln] * tb[oUr + i * 16 + wv] * tb[oEr + i * ES + Sx];
#endif
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c] = fma_m(r, nv[c], acc[c]);
        }
      };
      // request slot -> external bit.  REV (fp64): the far bits are requested first, the nearest - the most recently written
      // window, the likeliest cache hit - takes the slot at the end of the step that is requested and waited for in one go
      // (forward 13.5 -> 13.1 ms per 5 000 patients; no gain in fp32 at k = 25, whose slots are all requested a phase ahead)
      auto sb = [&](int sl) -> int { return C::REV ? nX - 1 - sl : sl; };
      // (MMHN_W_NTFAR = j0: the blocks of the external bits >= j0 - written 2^j0 window passes ago, far beyond what the XCD's L2
      // holds for this workgroup - are loaded non-temporally, so that they do not displace the near ones; measured, section 6)
      auto ld_ext = [&](int j) -> VecT {
        const uint32_t off = ext_off(j);
#if MMHN_W_NTFAR >= 0
        if (j >= MMHN_W_NTFAR) {
          Raw r;
          r.q[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, (int)boff, 2);
          r.q[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off + 16, (int)boff, 2);
          return __builtin_bit_cast(VecT, r);
        }
#endif
        return ld_row(off, boff);
      };
      VecT ev0, ev1;
      if (nX > 0) ev0 = ld_ext(sb(0));
#if !MMHN_W_STAG
      if (nX > 1) ev1 = ld_ext(sb(1));
#endif
      __builtin_amdgcn_sched_barrier(0);
      // lane moves: the neighbour lane's window slot (its previous window pass = this lane's pass)
      {
        const VecT old = Wd[beta];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#if MMHN_W_STAG
          if (c == NC / 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (nX > 1) ev1 = ld_ext(sb(1));
            __builtin_amdgcn_sched_barrier(0);
          }
#endif
          acc[c] = fma_m(cL[0], lane_nbr<0, TR>(old[c], (int)ln), acc[c]);
          acc[c] = fma_m(cL[1], lane_nbr<1, TR>(old[c], (int)ln), acc[c]);
          acc[c] = fma_m(cL[2], lane_nbr<2, TR>(old[c], (int)ln), acc[c]);
          acc[c] = fma_m(cL[3], lane_nbr<3, TR>(old[c], (int)ln), acc[c]);
#ifdef MMHN_WABL_NOPERM
          acc[c] = fma_m(cL[4], lane_nbr<3, TR>(old[c], (int)ln), acc[c]);
          acc[c] = fma_m(cL[5], lane_nbr<2, TR>(old[c], (int)ln), acc[c]);
#else
          acc[c] = fma_m(cL[4], lane_nbr<4, TR>(old[c], (int)ln), acc[c]);
          acc[c] = fma_m(cL[5], lane_nbr<5, TR>(old[c], (int)ln), acc[c]);
#endif
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      STAMP(0);
      if (nX > 0) ext_take(sb(0), ev0);
      if (nX > 1) ext_take(sb(1), ev1);
      if (nX > 2) {
        ev0 = ld_ext(sb(2));
        if (nX > 3) ev1 = ld_ext(sb(3));
      }
      // EV2 (transposed, five external bits - every k = 20 cohort): the fifth request goes out with the second pair into a
      // register set of its own instead of request-and-wait at the end (adjoint 14.1 -> 13.2 ms; forward 13.1 -> 13.6: off there)
      VecT ev2;
      if constexpr (EV2) ev2 = ld_ext(sb(4));
      __builtin_amdgcn_sched_barrier(0);
      STAMP(1);
      // wave moves: the block the neighbour wave published one step ago
      {
        const T* rs = ring + (uint32_t)(gpar ^ 1) * (NC * WROWS);
#pragma unroll
        for (int j = 0; j < WWB; ++j) {
#ifdef MMHN_WABL_NORING
          const bool has = false;
#else
          const bool has = TR ? !((wv >> j) & 1) : ((wv >> j) & 1);
#endif
          if (has) {                                           // wave-uniform
            const uint32_t row = tt ^ (64u << j);
            Raw r;
            r.q[0] = *reinterpret_cast<const u32x4*>(rs + row * QE);
            r.q[1] = *reinterpret_cast<const u32x4*>(rs + WROWS * QE + row * QE);
            const VecT nv = __builtin_bit_cast(VecT, r);
#if MMHN_W_CW
            const T cw = cW[j];
#elif defined(MMHN_WABL_NOTAB)
            const T cw = T(0.125);
#else
            const T cw = tb[oLr + (WLB + j) * 64 + ln] * tb[oUr + (WLB + j) * 16 + wv] * tb[oEr + (WLB + j) * ES + Sx];
#endif
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fma_m(cw, nv[c], acc[c]);
          }
        }
      }
      // DEEP (seven or more external bits, known at compile time - k = 25: 9): the pairs of external blocks are requested one
      // phase ahead of their use all the way down the step, instead of request-and-wait pairs at its end
      if constexpr (DEEP) {
        __builtin_amdgcn_sched_barrier(0);
        ext_take(sb(2), ev0);
        ext_take(sb(3), ev1);
        ev0 = ld_ext(sb(4));
        if (nX > 5) ev1 = ld_ext(sb(5));
      }
      __builtin_amdgcn_sched_barrier(0);
      STAMP(2);
      // window moves: own blocks of this window pass
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        const bool has = TR ? !((beta >> j) & 1) : ((beta >> j) & 1);
        if (has) {                                             // compile-time
          constexpr int dummy = 0; (void)dummy;
          const int sb = beta ^ (1 << j);
          const int wset = ((TR ? beta : sb) << RB);           // the window setting the rate is taken at (bit RB + j clear)
          const int wsq = ((wset >> (RB + j + 1)) << (RB + j)) | (wset & ((1 << (RB + j)) - 1));
          if constexpr (C::FACT) {
            const VecT rr = lds_vec(tb + L::oFw + (RB + j) * WIN + wset);
            const T fx = tb[oFx + (RB + j) * XS + Tx];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fma_m(rr[c] * fx, Wd[sb][c], acc[c]);
          } else {
            const VecT rr = lds_vec(tb + L::oRh + (RB + j) * SZLO + Tx * LOS + wsq);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fma_m(rr[c], Wd[sb][c], acc[c]);
          }
        }
      }
      if constexpr (DEEP) {
        __builtin_amdgcn_sched_barrier(0);
        ext_take(sb(4), ev0);
        if (nX > 5) ext_take(sb(5), ev1);
        if (nX > 6) ev0 = ld_ext(sb(6));
        if (nX > 7) ev1 = ld_ext(sb(7));
      } else if (nX > 2) {
        ext_take(sb(2), ev0);
        if (nX > 3) ext_take(sb(3), ev1);
        if constexpr (EV2) ext_take(sb(4), ev2);
        else
        for (int j0 = 4; j0 < nX; j0 += 2) {                   // (spaces of more than 20 bits)
          ev0 = ld_ext(sb(j0));
          if (j0 + 1 < nX) ev1 = ld_ext(sb(j0 + 1));
          ext_take(sb(j0), ev0);
          if (j0 + 1 < nX) ext_take(sb(j0 + 1), ev1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      STAMP(3);
      // right-hand side
      const uint32_t Tblk = (Tx << HB) | (uint32_t)beta;       // column set of the block >> RB
      if (!TR) {
        if (__builtin_amdgcn_ballot_w64((hitT >> RB) == Tblk) != 0ull) {  // (rare: skipped by a scalar branch)
          const T hv = (hitT >> RB) == Tblk ? tb[oSe + hitE] : T(0);
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c] += ((hitT & (uint32_t)(NC - 1)) == (uint32_t)c) ? hv : T(0);
        }
      } else {
        const WPInfo<T>& pi_ = pinfo[tbx != (uint32_t)L::tab0];
        if (wv == WROWS / 64 - 1) {                            // the last row of the last external row setting
          if (ln == 63u && Sx == Sxfull && soff != OOB && pi_.soff[0] >= 0) {
            const T* qr = qS + pi_.soff[0] + ((long long)Tblk << RB);
            const T cr = pi_.cst[0];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] += cr * qr[c];
          }
        }
        if (beta == H - 1) {                                   // the last column
          if (Tx == Txfull && soff != OOB && pi_.soff[1] >= 0)
            acc[NC - 1] += pi_.cst[1] * qS[pi_.soff[1] + (tt | (Sx << WTB))];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DEEP) {
        if (nX > 6) ext_take(sb(6), ev0);
        if (nX > 7) ext_take(sb(7), ev1);
        for (int j0 = 8; j0 < nX; j0 += 2) {
          ev0 = ld_ext(sb(j0));
          if (j0 + 1 < nX) ev1 = ld_ext(sb(j0 + 1));
          ext_take(sb(j0), ev0);
          if (j0 + 1 < nX) ext_take(sb(j0 + 1), ev1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the block itself: moves along the RB lowest column bits (only the rates in use are fetched), diagonal
      const VecT dcv = lds_vec(tb + oDC + Tx * HIS + beta * NC);
      T fxr[RB];                                               // FACT: external factors of the RB lowest column bits
      if constexpr (C::FACT) {
#pragma unroll
        for (int r = 0; r < RB; ++r) fxr[r] = tb[oFx + r * XS + Tx];
      }
      auto blk_rate = [&](int r, int wset, int wsq) -> T {     // rate of column bit r at window setting wset (bit r clear)
#ifdef MMHN_WABL_NOTAB
        return T(0.125);
#endif
        if constexpr (C::FACT) return tb[L::oFw + r * WIN + wset] * fxr[r];
        else return tb[L::oRh + r * SZLO + Tx * LOS + wsq];
      };
      VecT Y;
      if (!TR) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          T z = acc[c];
#pragma unroll
          for (int r = 0; r < RB; ++r)
            if ((c >> r) & 1) {
              constexpr int dummy = 0; (void)dummy;
              const int wset = (beta << RB) | (c ^ (1 << r));
              const int wsq = ((wset >> (r + 1)) << r) | (wset & ((1 << r) - 1));
              z = fma_m(blk_rate(r, wset, wsq), Y[c ^ (1 << r)], z);
            }
          Y[c] = z * fast_rcp(dRv + dcv[c]);
        }
      } else {
#pragma unroll
        for (int c = NC - 1; c >= 0; --c) {
          T z = acc[c];
#pragma unroll
          for (int r = 0; r < RB; ++r)
            if (!((c >> r) & 1)) {
              const int wset = (beta << RB) | c;
              const int wsq = ((wset >> (r + 1)) << r) | (wset & ((1 << r) - 1));
              z = fma_m(blk_rate(r, wset, wsq), Y[c | (1 << r)], z);
            }
          Y[c] = z * fast_rcp(dRv + dcv[c]);
        }
      }
      Wd[beta] = Y;
      if (TR ? wv != 0 : wv != WROWS / 64 - 1) {               // some wave above / below reads it
        T* ws = ring + (uint32_t)gpar * (NC * WROWS);
        const Raw r = __builtin_bit_cast(Raw, Y);
        *reinterpret_cast<u32x4*>(ws + tt * QE) = r.q[0];
        *reinterpret_cast<u32x4*>(ws + WROWS * QE + tt * QE) = r.q[1];
      }
#ifdef MMHN_WABL_NOSTORE  // timing-only ablation (wrong results): nothing is written
      st_row(OOB, boff, Y);
#else
      st_row(soff, boff, Y);
#endif
      STAMP(4);
    };
    // ---- the pipeline: wave-level lam delays the wave by lam steps, lane-level m delays a lane by m window passes;
    // a patient enters every 2^nX passes (the waves meet for its tables: four steps of slack) and, transposed, is
    // completed by its seed = 0 lattice six passes after the next one entered
    for (int sig = 0; sig < NPASS; ++sig) {
      const uint32_t ph = (uint32_t)sig & (NXW - 1u);
      const int jj = sig >> nXw;
      if (ph == 0 && jj >= 1 && jj < npat) { deskew(); enter(jj); reskew(); }
      // SPLIT: B needs A's iteration of the same number (A's word counts the iterations it has drained)
      if constexpr (SPLIT) { if (role == 1) wait_partner(cum + (unsigned)sig + 1u); }
      if (TR && ph == (uint32_t)WLB && jj >= 1 && !(SPLIT && role == 0)) { deskew(); leave(jj - 1); reskew(); }
 STAMP(7);
      begin_pass(sig);
      STAMP(6);
      const int g0 = lam & 1;                                  // parity of the global step (H is even)
      step(IC<0>{}, g0); lds_barrier(); STAMP(5);
      step(IC<1>{}, g0 ^ 1); lds_barrier(); STAMP(5);
      if constexpr (H > 2) {
        step(IC<2 % H>{}, g0); lds_barrier(); STAMP(5);
        step(IC<3 % H>{}, g0 ^ 1); lds_barrier(); STAMP(5);
      }
      if constexpr (H > 4) {
        step(IC<4 % H>{}, g0); lds_barrier(); STAMP(5);
        step(IC<5 % H>{}, g0 ^ 1); lds_barrier(); STAMP(5);
        step(IC<6 % H>{}, g0); lds_barrier(); STAMP(5);
        step(IC<7 % H>{}, g0 ^ 1); lds_barrier(); STAMP(5);
      }
      static_assert(H == 2 || H == 4 || H == 8, "window of 2, 4 or 8 blocks");
      if constexpr (SPLIT) {
        // A: this wave's stores up to here are complete; the wave that runs furthest behind (every other one passed this point
        // before it) raises the pair's word
        if (role == 0) {
#ifndef MMHN_WSPLIT_NODRAIN  // (experiment, timing only)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
          if (lam == WWB && (threadIdx.x & 63u) == 0u)
            __hip_atomic_store(prog, wbase + cum + (unsigned)sig + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if constexpr (SPLIT) {
      if (role == 1) wait_partner(cum + (unsigned)NPASS);      // (the transposed solve's last lattice reads A's half too)
      cum += (unsigned)NPASS;
    }
    deskew();
    if (TR && !(SPLIT && role == 0)) leave(npat - 1);
    __syncthreads();
    STAMP(7);
  }
  STAMP_FLUSH(TR ? 8 : 0);
}

// window layout -> natural index order (seeded half; the PT == MT states of the seed = 0 half are copied)
template <typename T>
__global__ __launch_bounds__(WROWS) void k_wconvert(const Desc* __restrict__ descs, const WDesc* __restrict__ wds,
                                                    const T* __restrict__ yw, T* __restrict__ yn) {
  constexpr int RB = WCfg<T>::RB, HB = WCfg<T>::HB, NC = 1 << RB;
  const WDesc& wd = wds[blockIdx.x];
  const Desc& d = descs[wd.prob];
  const int k = d.k, tid = threadIdx.x;
  const long long half = 1ll << (k - 1);
  const uint32_t rho = wrho((uint32_t)tid >> 6, (uint32_t)tid & 63u);
  const uint32_t nblk = 1u << (wd.nXc + wd.nXr + HB);
  const T* src = yw + d.off + half;
  T* dst = yn + d.off + half;
  uint32_t rowlo = 0;
  for (int i = 0; i < WTB; ++i) if ((tid >> i) & 1) rowlo |= 1u << wd.rb[i];
  for (uint32_t B = blockIdx.y; B < nblk; B += gridDim.y) {
    const uint32_t Sigma = B >> HB, beta = B & ((1u << HB) - 1u);
    const uint32_t Tx = Sigma & ((1u << wd.nXc) - 1u), Sx = Sigma >> wd.nXc;
    uint32_t xr = rowlo;
    for (int i = 0; i < wd.nXr; ++i) if ((Sx >> i) & 1u) xr |= 1u << wd.rb[WTB + i];
    const uint32_t Tc0 = (beta << RB) | (Tx << (RB + HB));
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const uint32_t Tc = Tc0 | (uint32_t)c;
      uint32_t xc = 0;
      for (int i = 0; i < wd.kC; ++i) if ((Tc >> i) & 1u) xc |= 1u << wd.cb[i];
      dst[xr | xc] = src[wpos<T>(Sigma, beta, rho, (uint32_t)c)];
    }
  }
  if (blockIdx.y == 0) {
    const uint32_t VE = 1u << __popc(d.pairP);
    for (uint32_t e = tid; e < VE; e += WROWS) {
      const uint32_t xp = pdep32(e, d.pairP);
      const uint32_t x0 = xp | (xp << 1);
      yn[d.off + x0] = yw[d.off + x0];
    }
  }
}

}  // namespace mmhn
