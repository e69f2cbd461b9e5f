// Small-space path: every single-tumour problem of a patient in ONE workgroup, whole state spaces in LDS.
//
// Real cohorts (BASELINE configs[0]: LUAD-reduced, 4 852 rows, k <= 14) are thousands of tiny problems; with one
// launch per pipeline stage the evaluation is a chain of ~45 short dependent kernels.  k_spatient replaces, for a batch
// whose single-tumour spaces all fit one tile (k <= TB), the stages
//     k_diag(KD_LIDG) -> k_tsolve (k+1 level launches) -> k_seeds -> k_tsolve^T -> k_grad_rows(GK_S) -> k_bit_marg
//     -> k_scatter_marg (dots)                                      (+ the zero fills of their outputs)
// by one launch: per patient its one or two single-tumour problems (the patient's own space for unpaired rows,
// likelihood.py:387-512; the MT / PT marginals of a paired row, :516-620) are solved forward, the adjoint seeds
// 1 / score are formed (:438, :655-660), the adjoints are solved, and the gradient rows (vanilla.py:328-393 in flow
// form), the observation-rate bit marginals of MT-only rows (vanilla.py:190-203) and the <q, rhs> dots of the paired
// parts (likelihood.py:576, :619) are reduced - all from LDS.  Same arithmetic as the staged kernels.
#pragma once
#include "kernels.h"

namespace mmhn {

// Patients are dealt to four size classes by the largest of their single-tumour spaces (k <= 4, <= 6, <= 9, <= TB): a class is
// one launch with the LDS its largest space needs, so that the thousands of tiny patients of a real cohort do not pay
// for the few large ones (a wave per patient for k <= 6, 4 or 16 waves otherwise).
constexpr int SP_NCLASS = 3;
constexpr int SP_PPB0 = 4;                    // patients (waves) per workgroup of the one-wave class
__host__ __device__ inline int spatient_class_maxk(int c) { return c == 0 ? 6 : c == 1 ? 9 : TB; }
__host__ __device__ inline int spatient_lstride(int maxk) { return maxk < 6 ? 1 << maxk : 64; }
__host__ __device__ inline int spatient_rows(int maxk) { return maxk > 6 ? 1 << (maxk - 6) : 1; }

// LDS: rate tables Lc [N][64] / Uc [N][rows] (all events: the diagonal needs the inactive ones too), observation
// products LA / UA / LB / UB [64], vectors P0, Q, LID [2^maxk] (the second forward solution of a two-part patient
// waits in the global pS buffer while the first part's gradient is formed)
template <typename T>
__host__ __device__ inline size_t spatient_lds(int N, int maxk) {
  return ((size_t)N * (spatient_lstride(maxk) + spatient_rows(maxk) + 16 + 1) + 4 * 64 + 2 * 16 + ((size_t)3 << maxk) + 64) * sizeof(T) +
         (64 + TB + 2) * sizeof(int) + 8 + DESC_PAD + (sizeof(uint16_t) << maxk);
}

// SPB threads work on one patient; PPB patients share a workgroup (PPB > 1 only with SPB = 64: a wave per patient, no
// workgroup barrier anywhere - thousands of one-wave workgroups are bound by the dispatcher, ~25 workgroups / us)
// PAIR (paired rows with BOTH marginal problems, order unknown - likelihood.py:516-620): the two problems are
// independent but for the sum of their scores, so each gets its own SPB threads and its own LDS slot (PPB slots, two per
// patient) and they run side by side: one table setup per problem instead of two, no parking of the second forward
// solution, and the chain of the patient is one problem long.  The scores meet in LDS behind one workgroup barrier.
// With SPB > 64 the two halves share every barrier, so both walk max(k0, k1) + 1 levels.
template <typename T, int SPB, int PPB, bool PAIR = false>
__device__ __forceinline__ void spatient_body(const int* __restrict__ plist, int npl, const PatRec* __restrict__ pats, const Desc* __restrict__ dS,
                                              const Params<T>* __restrict__ par,
                                              const uint16_t* __restrict__ perm, const int* __restrict__ lvl,
                                              const Desc* __restrict__ dJ, const WDesc* __restrict__ wds, const T* __restrict__ pi, JLink<T>* __restrict__ links,
                                              T* __restrict__ pS, T* __restrict__ qS,
                                              T* __restrict__ GS, T* __restrict__ bmS, T* __restrict__ dots,
                                              double* __restrict__ lp, int maxk, int N, int with_grad, int block) {
  constexpr int SPW = SPB / 64;
  static_assert(PAIR ? (PPB == 2 || (SPB == 64 && PPB % 2 == 0)) : (PPB == 1 || SPB == 64),
                "several patients per workgroup: one wave each; a pair of problems: two slots");
  extern __shared__ __align__(16) unsigned char smem_all[];
  const int pslot = PPB == 1 ? 0 : (int)(threadIdx.x / SPB);
  const size_t slot_bytes = (spatient_lds<T>(N, maxk) + 15) / 16 * 16;
  unsigned char* smem = smem_all + (size_t)pslot * slot_bytes;
  T* xch = reinterpret_cast<T*>(smem_all + (size_t)PPB * slot_bytes);      // PAIR: the scores of the slots
  const int mypart = pslot & 1;                                            // PAIR: the problem of this slot
  // synchronisation among the threads of ONE patient
  auto sync = [&]() {
    if (SPB == 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
  };
  const int RS = spatient_rows(maxk);                      // row stride of Uc
  const int LS = spatient_lstride(maxk);                   // row stride of Lc
  T* Lc = reinterpret_cast<T*>(smem);
  T* Uc = Lc + N * LS;
  T* LA = Uc + N * RS;
  T* UA = LA + 64;
  T* LB = UA + 64;
  T* UB = LB + 64;
  T* P0 = UB + 64;
  T* Qv = P0 + ((size_t)1 << maxk);
  T* LID = Qv + ((size_t)1 << maxk);
  T* red = LID + ((size_t)1 << maxk);                     // 64 scratch entries
  T* thl = red + 64;                                       // [N][16]: theta[i][event of local bit b]
  T* basel = thl + N * 16;                                 // [N] base rates
  T* dpl = basel + N;                                      // [16] d_p / d_m of the local bits
  T* dml = dpl + 16;
  int* lev = reinterpret_cast<int*>(dml + 16);            // event of local bit l
  int* loff = lev + 64;                                    // offsets of the popcount levels in pml
  // the problem's descriptor (global reads off the critical loops); 8-byte aligned whatever the element counts before it
  // (its offsets are 64-bit: for T = float an odd number of elements would put it on a 4-byte boundary)
  Desc& dsh = *reinterpret_cast<Desc*>((reinterpret_cast<uintptr_t>(loff + TB + 2) + 7u) & ~(uintptr_t)7u);
  uint16_t* pml = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(&dsh) + DESC_PAD);   // states by popcount
  const int pidx = PAIR ? block * (PPB / 2) + (pslot >> 1) : block * PPB + pslot;
  // (PAIR, one wave per problem: every wave of the workgroup meets the one workgroup barrier of the score exchange)
  if (pidx >= npl) { if (PAIR && SPB == 64) __syncthreads(); return; }
  const int pat = plist[pidx];
  const PatRec pr = pats[pat];
  const int ps0 = pr.s[0], ps1 = pr.s[1];                  // (no runtime index into the record: it would live in scratch)
  if (pr.kind == 4 || (PAIR ? (ps0 < 0 || ps1 < 0) : (ps0 < 0 && ps1 < 0))) { if (PAIR && SPB == 64) __syncthreads(); return; }
  // levels both halves of a PAIR workgroup walk (their barriers are shared)
  const int kwalk = (PAIR && SPB > 64) ? max(dS[ps0].k, dS[ps1].k) : -1;
  const int tid = (int)(threadIdx.x % SPB), lane = tid & 63, w = tid >> 6;
  const int n = N - 1;

  // tables of one problem: rate products over the lane bits (Lc) and the row bits (Uc, incl. the base rate),
  // observation-rate products (k_diag), 1 / (D_obs - diag Q) for every state
  auto setup = [&](const Desc& dg) {
    sync();
    {
      const int* sw = reinterpret_cast<const int*>(&dg);
      int* dw = reinterpret_cast<int*>(&dsh);
      for (int i = tid; i < DESC_WORDS; i += SPB) dw[i] = sw[i];
    }
    sync();
    const Desc& d = dsh;
    const int k = d.k;
    const Params<T>& P = par[d.pset];
    const int nl = k < 6 ? k : 6;
    const int R = k > 6 ? 1 << (k - 6) : 1;
    if (tid < k) { lev[tid] = d.ev[tid]; dpl[tid] = P.dp[d.ev[tid]]; dml[tid] = P.dm[d.ev[tid]]; }
    for (int e = tid; e < N * k; e += SPB) { const int i = e / k, b = e % k; thl[i * 16 + b] = P.th[i][d.ev[b]]; }
    for (int i = tid; i < N; i += SPB) basel[i] = P.baseP[i];
    if (SPB > 64) {                                        // (a single wave takes one state per lane: no order table)
      const uint16_t* pm = perm + ((size_t)k << TB);
      for (uint32_t e = tid; e < (1u << k); e += SPB) pml[e] = pm[e];
      if (tid <= k + 1) loff[tid] = lvl[(size_t)k * (TB + 2) + tid];
    }
    sync();
    const int nL = 1 << nl;
    for (int e = tid; e < N * nL; e += SPB) {
      const int i = e >> nl, l = e & (nL - 1);
      T t[6];                                              // all factors first (independent LDS reads), then the product
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) t[bb] = thl[i * 16 + bb];
      T v = 1;
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) v *= (bb < nl && ((l >> bb) & 1)) ? t[bb] : T(1);
      Lc[i * LS + l] = v;
    }
    for (int e = tid; e < N * R; e += SPB) {
      const int i = e / R, r = e % R;
      T t[6];
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) t[bb] = thl[i * 16 + 6 + bb];
      T u = basel[i];
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) u *= (6 + bb < k && ((r >> bb) & 1)) ? t[bb] : T(1);
      Uc[i * RS + r] = u;
    }
    if (tid < 64) {
      T a = 1, b = 1, ua = 1, ub = 1;
      for (int bb = 0; bb < k; ++bb) {
        if (bb == d.seedbit) continue;
        if (bb < 6) { if ((tid >> bb) & 1) { a *= dpl[bb]; b *= dml[bb]; } }
        else if (tid < R && ((tid >> (bb - 6)) & 1)) { ua *= dpl[bb]; ub *= dml[bb]; }
      }
      LA[tid] = a; LB[tid] = b; UA[tid] = ua; UB[tid] = ub;
    }
    sync();
    const uint32_t V = 1u << k;
    const T dmn = P.dm[n];
    for (uint32_t x = tid; x < V; x += SPB) {
      const uint32_t lo = x & 63u, ro = x >> 6;
      uint32_t gone = 0;                                   // events that have happened in x (bitP is the inverse of ev)
      for (int b = 0; b < k; ++b) if ((x >> b) & 1u) gone |= 1u << lev[b];
      T dq = 0;
#pragma unroll 8
      for (int i = 0; i < N; ++i) {                        // (branch-free: the table reads of all events go out together)
        const T r = Lc[i * LS + lo] * Uc[i * RS + ro];
        dq -= ((gone >> i) & 1u) ? T(0) : r;
      }
      const bool sbit = d.seedbit >= 0 && ((x >> d.seedbit) & 1u);
      T dob = 1;
      if (d.obs == OBS_MET) dob = sbit ? LB[lo] * UB[ro] * dmn : LA[lo] * UA[ro];
      LID[x] = T(1) / (dob - dq);
    }
    sync();
  };
  // (D - Q) y = rhs (TR: transposed) by substitution in popcount order; rhs(x) is a functor (evaluated for all states
  // up front: its global reads stay off the level chain)
  auto solve = [&](const Desc& d, T* y, bool tr, auto rhs) {
    const int k = d.k;
    const uint32_t V = 1u << k, full = V - 1u;
    for (uint32_t x = tid; x < V; x += SPB) y[x] = rhs(x);
    sync();
    auto state = [&](uint32_t x) {
      T z = y[x];
      if (!tr) {
        for (uint32_t m = x; m; m &= m - 1) {
          const int b = __ffs(m) - 1;
          const uint32_t xs = x ^ (1u << b);
          z += Lc[lev[b] * LS + (xs & 63u)] * Uc[lev[b] * RS + (xs >> 6)] * y[xs];
        }
      } else {
        for (uint32_t m = ~x & full; m; m &= m - 1) {
          const int b = __ffs(m) - 1;
          z += Lc[lev[b] * LS + (x & 63u)] * Uc[lev[b] * RS + (x >> 6)] * y[x | (1u << b)];
        }
      }
      y[x] = LID[x] * z;
    };
    const int ns = kwalk >= 0 ? kwalk : k;
    for (int s = 0; s <= ns; ++s) {
      const int level = tr ? k - s : s;
      if (SPB == 64) {
        if ((uint32_t)tid < V && __popc((uint32_t)tid) == level) state((uint32_t)tid);
      } else if (s <= k) {
        for (int idx = loff[level] + tid; idx < loff[level + 1]; idx += SPB) state(pml[idx]);
      }
      sync();
    }
  };

  // paired rows: the right-hand side of a marginal problem is [0 ; D * pi[compatible states]] (k_gather_marg, likelihood.py
  // :573-575, :617-618) - taken straight from the joint solution, no launch and no buffer in between; D is constant on
  // those states (obs_const: same factors in the same order) and is also the constant of the joint adjoint's
  // right-hand side, so the links are written here
  // (wl >= 0: the joint solution lives in the window layout, wlayout.h)
  struct Marg { T c; uint32_t fixed, free_, half; long long joff; int wl, kj; bool free_is_row; };
  auto marg_of = [&](int part, int k) -> Marg {
    const Desc* dj = dJ + pr.j;
    const Params<T>& PT = par[PS_THETA];
    const int kj = dj->k;
    sync();
    if (tid < 32) {
      T f = 1;
      if (tid < kj && dj->cls[tid] == (part == 0 ? CP : CM)) f = part == 0 ? PT.dp[dj->ev[tid]] : PT.dm[dj->ev[tid]];
      red[tid] = f;
    }
    sync();
    T c = part == 0 ? PT.dp[N - 1] : PT.dm[N - 1];
    for (int b = 0; b < kj; ++b) c *= red[b];
    sync();
    Marg m;
    m.c = c;
    m.fixed = (part == 0 ? dj->maskP : dj->maskM) | (1u << dj->seedbit);
    m.free_ = part == 0 ? dj->maskM : dj->maskP;
    m.half = 1u << (k - 1);
    m.joff = dj->off;
    m.wl = dj->wl; m.kj = kj;
    m.free_is_row = m.wl >= 0 && (wds[m.wl].majP != 0) == (part == 1);      // part 0 frees the M bits, part 1 the P bits
    return m;
  };
  auto marg_rhs = [&](const Marg& m, uint32_t x) -> T {
    if (x < m.half) return T(0);
    if (m.wl >= 0) return m.c * pi[m.joff + wpos_marg<T>(wds[m.wl], m.kj, m.free_is_row, x - m.half)];
    return m.c * pi[m.joff + (pdep32(x - m.half, m.free_) | m.fixed)];
  };
  Marg mg0{}, mg1{};

  // ---- forward solves of the patient's problems
  STAMP_DECL;
  STAMP_START;
  T full_score = 0;
  for (int part = 0; part < 2; ++part) {
    const int sp = part == 0 ? ps0 : ps1;
    if (sp < 0 || (PAIR && part != mypart)) continue;
    setup(dS[sp]);
    STAMP(0);
    const Desc& d = dsh;
    const bool own = pr.kind <= 2;                         // the patient's own space: right-hand side e_0
    Marg mg{};
    if (!own) {
      mg = marg_of(part, d.k);
      if (part == 0) mg0 = mg; else mg1 = mg;
      if (tid == 0) { links[pr.j].soff[part] = d.off; links[pr.j].sk[part] = d.k; links[pr.j].cst[part] = mg.c; }
    }
    const bool stash = !PAIR && part == 1 && ps0 >= 0;     // second part of a two-part patient: solved in Q, parked in pS
    T* yv = stash ? Qv : P0;
    solve(d, yv, false, [&](uint32_t x) { return own ? (x == 0 ? e0_scale<T>() : T(0)) : marg_rhs(mg, x); });
    full_score += yv[(1u << d.k) - 1u];
    if (stash) for (uint32_t x = tid; x < (1u << d.k); x += SPB) pS[d.off + x] = yv[x];
    STAMP(1);
  }
  if (pr.kind == 3 && !PAIR && tid == 0) {                 // a part the row does not have: nothing feeds the joint adjoint
    if (ps0 < 0) { links[pr.j].soff[0] = -1; links[pr.j].sk[0] = 0; links[pr.j].cst[0] = 0; }
    if (ps1 < 0) { links[pr.j].soff[1] = -1; links[pr.j].sk[1] = 0; links[pr.j].cst[1] = 0; }
  }
  if (PAIR) {                                              // score of part 0 + score of part 1, in that order
    if (tid == 0) xch[pslot] = full_score;
    __syncthreads();
    full_score = xch[pslot & ~1] + xch[pslot | 1];
  }
  const T seed = T(1) / full_score;
  if (tid == 0 && !(PAIR && mypart == 1)) {
    double l = log((double)full_score) - log((double)e0_scale<T>());
    if (pr.kind == 2) {                                    // likelihood.py:438
      const Desc& d = dsh;                                 // an MT-only row has one problem: still staged
      const Params<T>& P = par[PS_THETA];
      double dr = (double)P.dm[N - 1];
      for (int b = 0; b < d.k; ++b) if (b != d.seedbit) dr *= (double)dml[b];
      l += log(dr);
    }
    lp[pat] = l;
  }
  STAMP(2);
  if (!with_grad) { STAMP_FLUSH(0); return; }

  // ---- adjoints, gradient rows, observation-rate marginals, dots
  const int nparts = (ps0 >= 0) + (ps1 >= 0);
  for (int part = 0; part < 2; ++part) {
    const int sp = part == 0 ? ps0 : ps1;
    if (sp < 0 || (PAIR && part != mypart)) continue;
    if (nparts == 2 && !PAIR) setup(dS[sp]);               // a lone part's tables (and descriptor) are still in place
    const Desc& d = dsh;
    const int k = d.k;
    const uint32_t V = 1u << k, last = V - 1u;
    solve(d, Qv, true, [&](uint32_t x) { return x == last ? seed : T(0); });
    STAMP(3);
    if (!PAIR && part == 1 && ps0 >= 0) {                  // fetch the parked forward solution (P0 is free now)
      sync();
      for (uint32_t x = tid; x < V; x += SPB) P0[x] = pS[d.off + x];
      sync();
    }
    const T* p = P0;
    if (pr.kind == 3) {
      // the joint adjoint reads q of the marginal problem (k_psolve rhs_mode 3, k_tsolve); <q, rhs> on the upper half
      T* qg = qS + d.off;
      const Marg& mg = part == 0 ? mg0 : mg1;
      const uint32_t half = V >> 1;
      T dot = 0;
      for (uint32_t x = tid; x < V; x += SPB) {
        const T qv = Qv[x];
        qg[x] = qv;
        if (x >= half) dot += qv * marg_rhs(mg, x);
      }
      dot = wave_sum_dpp(dot);
      if (lane == 0) red[w] = dot;
      sync();
      if (tid == 0) { T t = 0; for (int v = 0; v < SPW; ++v) t += red[v]; dots[2 * pat + part] = t; }
    }
    // observation-rate bit marginals of an MT-only row (k_bit_marg, SINGLE / OBS_MET)
    if (pr.kind == 2) {
      T* bm = bmS + (long long)sp * 64;
      const Params<T>& P0 = par[d.pset];
      T accA[TB], accB[TB];
#pragma unroll
      for (int b = 0; b < TB; ++b) { accA[b] = 0; accB[b] = 0; }
      for (uint32_t x = tid; x < V; x += SPB) {
        const uint32_t lo = x & 63u, ro = x >> 6;
        const bool sbit = d.seedbit >= 0 && ((x >> d.seedbit) & 1u);
        const T pq = p[x] * Qv[x];
        const T vA = sbit ? T(0) : pq * LA[lo] * UA[ro], vB = sbit ? pq * LB[lo] * UB[ro] * P0.dm[n] : T(0);
#pragma unroll
        for (int b = 0; b < TB; ++b) if (b < k && ((x >> b) & 1u)) { accA[b] += vA; accB[b] += vB; }
      }
      sync();
      T* part_ = LID;                                      // LID is free now: [SPW][2][TB] partials
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        if (b < k) {                                         // wave-uniform
          const T a = wave_sum_dpp(accA[b]), bb = wave_sum_dpp(accB[b]);
          if (lane == 0) { part_[(w * 2 + 0) * TB + b] = a; part_[(w * 2 + 1) * TB + b] = bb; }
        }
      }
      sync();
      if (tid < 64) {
        const int ww = tid >> 5, b = tid & 31;
        T m = 0;
        if (b < k) for (int v = 0; v < SPW; ++v) m += part_[(v * 2 + ww) * TB + b];
        bm[ww * 32 + b] = m;
      }
    }
    STAMP(4);
    // gradient rows (k_grad_rows GK_S)
    T* G = GS + (long long)sp * N * N;
    if (SPW == 1) {
      // one wave: lane i owns event i and walks the states.  The state is wave-uniform, so "S contains bit l" is a
      // scalar branch and every G[i][ev(l)] is a private register sum - no cross-lane reduction at all
      // (two lanes per event, N <= 32: lane i + 32 h takes the states whose top bit is h - half the walk; the low bits
      // of the state are the same in both halves, the top bit's column is the upper half's total)
      const int i = lane & 31, h = lane >> 5;
      const bool rowok = i < N;
      const int slot = rowok ? d.bitP[i] : -1;             // local bit of event i or -1
      const int kt = k > 0 ? k - 1 : 0;                    // the top bit
      const uint32_t Vh = k > 0 ? V >> 1 : 1u, hs = h && k > 0 ? Vh : 0u;
      T tot = 0, acc[6];
#pragma unroll
      for (int l = 0; l < 6; ++l) acc[l] = 0;
      if (rowok && !(h && k == 0)) {
        const T ui = Uc[i * RS];
        const uint32_t sb = slot >= 0 ? 1u << slot : 0u;
#pragma unroll 4
        for (uint32_t Sl = 0; Sl < Vh; ++Sl) {
          const uint32_t S = Sl | hs;
          const T pv = p[S], q0 = Qv[S], q1 = Qv[S | sb], lc = Lc[i * LS + S];
          T a = -pv * q0;
          if (slot >= 0) a += pv * q1;
          const T f = (S & sb) ? T(0) : lc * ui * a;
          tot += f;
#pragma unroll
          for (int l = 0; l < 5; ++l) if ((Sl >> l) & 1u) acc[l] += f;      // wave-uniform condition
        }
      }
      // the upper half's sums come down
      const T tup = __shfl_down(tot, 32);
      T aup[5];
#pragma unroll
      for (int l = 0; l < 5; ++l) aup[l] = __shfl_down(acc[l], 32);
      if (rowok && h == 0) {
        if (k > 0) {
#pragma unroll
          for (int l = 0; l < 5; ++l) acc[l] += aup[l];
#pragma unroll
          for (int l = 0; l < 6; ++l) if (l == kt) acc[l] = tup;
          tot += tup;
        }
        T* row = G + (long long)i * N;
        for (int j = 0; j < N; ++j) row[j] = 0;
        row[i] = tot;
#pragma unroll
        for (int l = 0; l < 6; ++l) if (l < k && lev[l] != i) row[lev[l]] = acc[l];
      }
    } else {
      // wave w takes the events i = w, w + SPW, ..., lanes stride the states, DPP wave sums
      const int klo = k < 6 ? k : 6, nhi = k - klo;
      for (int i = w; i < N; i += SPW) {
        const int slot = d.bitP[i];
        const uint32_t sb = slot >= 0 ? 1u << slot : 0u;
        const T lcv = Lc[i * LS + lane];                   // (the lane part of the rate is the same in every chunk)
        T tot = 0, hi[TB - 6];
#pragma unroll
        for (int l = 0; l < TB - 6; ++l) hi[l] = 0;
        constexpr int GU = 4;                              // chunks of 64 states in flight (reads first, branch-free)
        for (uint32_t S00 = 0; S00 < V; S00 += 64 * GU) {
          T pv[GU], q0[GU], q1[GU], uc[GU];
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            const uint32_t S = S00 + 64 * u + (uint32_t)lane, Sc = S < V ? S : 0u;
            pv[u] = p[Sc]; q0[u] = Qv[Sc]; q1[u] = Qv[Sc | sb]; uc[u] = Uc[i * RS + (Sc >> 6)];
          }
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            const uint32_t S0 = S00 + 64 * u, S = S0 + (uint32_t)lane;
            T a = -pv[u] * q0[u];
            if (slot >= 0) a += pv[u] * q1[u];
            const T f = (S < V && !(S & sb)) ? lcv * uc[u] * a : T(0);
            tot += f;
#pragma unroll
            for (int l = 0; l < TB - 6; ++l) if (l < nhi && ((S0 >> (6 + l)) & 1u)) hi[l] += f;
          }
        }
        T total, ML[6];
        wave_bit_sums(tot, lane, klo, total, ML);
        T mine = 0;                                        // lane j keeps G[i][j]
        if (lane == i) mine = total;
#pragma unroll
        for (int l = 0; l < 6; ++l)
          if (l < klo && lane == lev[l] && lev[l] != i) mine = ML[l];
#pragma unroll
        for (int h = 0; h < TB - 6; ++h) {                 // static register index (no scratch)
          if (h < nhi) {
            const T m = wave_sum_dpp(hi[h]);
            if (lane == lev[klo + h] && lev[klo + h] != i) mine = m;
          }
        }
        if (lane < N) G[i * N + lane] = mine;
      }
    }
    sync();
    STAMP(5);
  }
  STAMP_FLUSH(0);
}

#define SPATIENT_PARAMS const PatRec* __restrict__ pats, const Desc* __restrict__ dS, const Params<T>* __restrict__ par, \
    const uint16_t* __restrict__ perm, const int* __restrict__ lvl, const Desc* __restrict__ dJ, const WDesc* __restrict__ wds, \
    const T* __restrict__ pi, \
    JLink<T>* __restrict__ links, T* __restrict__ pS, \
    T* __restrict__ qS, T* __restrict__ GS, T* __restrict__ bmS, T* __restrict__ dots, double* __restrict__ lp
#define SPATIENT_ARGS pats, dS, par, perm, lvl, dJ, wds, pi, links, pS, qS, GS, bmS, dots, lp

// one size class per launch (the 1024-thread class)
template <typename T, int SPB, int PPB, bool PAIR = false>
__global__ __launch_bounds__(SPB * PPB) void k_spatient(const int* __restrict__ plist, int npl, SPATIENT_PARAMS, int maxk, int N,
                                                        int with_grad) {
  spatient_body<T, SPB, PPB, PAIR>(plist, npl, SPATIENT_ARGS, maxk, N, with_grad, (int)blockIdx.x);
}

// the two 256-thread classes in ONE launch - side by side without a second stream: workgroups [0, nb0) take SP_PPB0
// one-wave patients each (list0, spaces of at most maxk0 bits), the others one patient each (list1, maxk1)
// pair0 / pair1: the paired rows of the two classes that have both marginal problems (PAIR above: two one-wave patients,
// or one patient with 128 threads per problem, to a workgroup).  Longest first: the 256-thread classes, then the waves.
template <typename T>
__global__ __launch_bounds__(256) void k_spatient2(const int* __restrict__ list0, int n0, int maxk0, const int* __restrict__ list1,
                                                   int n1, int maxk1, const int* __restrict__ pair0, int np0,
                                                   const int* __restrict__ pair1, int np1, SPATIENT_PARAMS, int N, int with_grad) {
  int b = (int)blockIdx.x;
  if (b < np1) { spatient_body<T, 128, 2, true>(pair1, np1, SPATIENT_ARGS, maxk1, N, with_grad, b); return; }
  b -= np1;
  if (b < n1) { spatient_body<T, 256, 1>(list1, n1, SPATIENT_ARGS, maxk1, N, with_grad, b); return; }
  b -= n1;
  const int nbp = (np0 + SP_PPB0 / 2 - 1) / (SP_PPB0 / 2);
  if (b < nbp) { spatient_body<T, 64, SP_PPB0, true>(pair0, np0, SPATIENT_ARGS, maxk0, N, with_grad, b); return; }
  spatient_body<T, 64, SP_PPB0>(list0, n0, SPATIENT_ARGS, maxk0, N, with_grad, b - nbp);
}
#undef SPATIENT_PARAMS
#undef SPATIENT_ARGS

}  // namespace mmhn
