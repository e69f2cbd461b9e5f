// k_psolve2: the joint triangular solve with one workgroup per PATIENT walking its seeded tiles (route RT_P of the engine).
// Reference: likelihood.py:231-262 (R_i_inv_vec).
#pragma once
#include "common.h"

namespace mmhn {

// ------------------------------------------------------------------------------------
// One workgroup per PATIENT (k_psolve2 below; round 1's k_psolve, which also took partial and seed-inside tiles, is gone
// since round 5: those problems take the cooperative tile launch of tsolve.h).
//
// With hundreds of patients in flight there is no need for parallelism inside a patient: index order is itself a
// valid substitution order (every neighbour H ^ bit of a tile has a smaller tile index; larger for the
// transpose), so one workgroup walks its patient's live tiles in that order.  Descriptor, rate tables and pext
// tables are set up once per patient instead of once per tile, the solve is a
// single launch, and the neighbour tiles a tile reads were written moments earlier by the same CU.
// Joint spaces with seeding only (class-table diagonal; right-hand side e_0 or the on-the-fly adjoint rhs).
// ------------------------------------------------------------------------------------
#ifndef MMHN_Q_TPT
#define MMHN_Q_TPT 1          // base-bit moves in flight per lane and trip of k_psolve2's in-tile solve
#endif
#ifndef MMHN_Q_TPA
#define MMHN_Q_TPA 2          // neighbour tiles in flight per thread in k_psolve2's step A
#endif
#ifndef MMHN_Q_EARLY
#define MMHN_Q_EARLY 0        // 1: first trip of neighbour-tile loads requested before the per-tile set-up barriers (measured: the
                              // values spill across the set-up at 64 VGPRs - 68 B scratch, 27.8 instead of 22.3 ms; one tile ahead: 23.2 ms)
#endif
#ifndef MMHN_Q_CLATE
#define MMHN_Q_CLATE -1
#endif
#ifndef MMHN_Q_LANES
#define MMHN_Q_LANES 2        // lanes that share one group of k_psolve2's in-tile solve (1, 2 or 4)
#endif
constexpr int PS_DL2 = 1040;                    // most LDS entries k_psolve2 spends on the per-tile dP / dM slices: up to 2^10 + 2^4

// sum_{i < l} C(n, i): offset of popcount level l in the popcount-sorted list of the n-bit states
template <int N>
__host__ __device__ constexpr uint32_t binom_prefix(int l) {
  uint32_t sum = 0, c = 1;
  for (int i = 0; i < l; ++i) { sum += c; c = c * (uint32_t)(N - i) / (uint32_t)(i + 1); }
  return sum;
}
template <int N>
struct BinomPrefix {
  uint32_t v[N + 2];
  constexpr BinomPrefix() : v{} { for (int l = 0; l <= N + 1; ++l) v[l] = binom_prefix<N>(l); }
};
// value of another lane of the same DPP quad (quad_perm control CTRL: 0xB1 = lane ^ 1, 0x4E = lane ^ 2,
// 0x00 / 0x55 / 0xAA / 0xFF = broadcast of lane 0 / 1 / 2 / 3): VALU moves, no LDS traffic
template <int CTRL>
__device__ __forceinline__ double quad_xor(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float quad_xor(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// the value lane `src` of every HP-lane cluster of a quad holds (HP = 1: the lane's own)
template <int HP, typename T>
__device__ __forceinline__ T cluster_bcast(T v, int src) {
  if (HP == 1) return v;
  if (HP == 2) return src == 0 ? quad_xor<0xA0>(v) : quad_xor<0xF5>(v);
  return src == 0 ? quad_xor<0x00>(v) : src == 1 ? quad_xor<0x55>(v) : src == 2 ? quad_xor<0xAA>(v) : quad_xor<0xFF>(v);
}

// ------------------------------------------------------------------------------------
// k_psolve2: k_psolve for launches whose tiles are all full seeded ones (MULTI), with the in-tile solve re-cut so that
// it no longer lives on three random LDS gathers per term.
//
// A thread owns a sub-cube of NJ = 2^G states of the tile: the GL = G - 1 lowest index bits and the highest tile bit
// (bit HB = TB - 1) vary inside the thread, the BB = TB - G bits between them are the thread's BASE state u:
//     x = jl | (u << GL) | (jh << HB),      slot in the thread's group  jq = jl | (jh << GL).
// The tile lives in LDS group-contiguous (yt[(u << G) | jq]): a group is one 32-byte vector access.
//   * global traffic (steps A and C) uses the natural base u = tid: 2^GL adjacent states per lane = 16-byte accesses;
//   * the in-tile solve (step B) runs level by level over the popcount of the base state (BB + 1 barriers instead of
//     TB + 1): the groups of a level are dealt to clusters of HP lanes of one DPP quad (a level has at most
//     C(BB, BB/2) groups = a quarter of the workgroup), the transitions along the in-thread bits are register
//     arithmetic, and one base-bit move serves NJ terms with ONE neighbour-group read, one Ltab vector, one Utab
//     entry and the factor of bit HB (5 LDS instructions per NJ terms instead of 3 per term).
// Measured (5 000 patients, n = 20, fp64, same box, interleaved): 22.3 / 21.9 ms forward / adjoint against 23.3 / 23.6 ms
// for k_psolve; HP = 1, 2, 4 and 1 - 2 moves or 1 - 4 neighbour tiles in flight per lane all land within 22 - 25 ms, and
// serving every neighbour read and every store from L2 (timing-only ablation) takes only 2 ms off: the kernel is bound
// by the chain of dependent, bank-conflicted LDS gathers between barriers (SQ_WAIT_ANY 82 % of the wave cycles, VALU
// 27 %, LDS 39 % busy with 52 % of its cycles conflicts), not by HBM.
// Same arithmetic as k_psolve term by term (rate_b(s) = Ltab[b][s & 63] * Utab[b][s >> 6], s the source state).
// ------------------------------------------------------------------------------------
#ifdef MMHN_ABL_PACK   // timing-only ablation (wrong results): the work of MMHN_ABL_PACK tiles per phase, one workgroup per CU
#define PS2_WPE 4
#else
#define PS2_WPE TSB_WPE
#endif
template <typename T, bool TR, bool DLOK>
__global__ __launch_bounds__(TSB, PS2_WPE) void k_psolve2(const Desc* __restrict__ descs,
                                                    const int* __restrict__ pt_off,
                                                    const uint32_t* __restrict__ ptiles,
                                                    const Params<T>* __restrict__ par, T* y, int rhs_mode,
                                                    const uint16_t* __restrict__ perm, int maxk,
                                                    const T* __restrict__ tab,
                                                    const JLink<T>* __restrict__ links,
                                                    const T* __restrict__ qS, int dl_cap,
                                                    const int* __restrict__ plist) {
  constexpr int NJ = (1 << TB) / TSB;                            // states per thread
  constexpr int G = NJ == 4 ? 2 : NJ == 8 ? 3 : NJ == 2 ? 1 : -1;
  static_assert(G >= 1 && TB == 12, "k_psolve2: 2, 4 or 8 states per thread, 2^12-state tiles");
  constexpr int GL = G - 1, NL = 1 << GL;                        // in-thread low bits
  constexpr int BB = TB - G;                                     // base bits
  constexpr int HB = TB - 1;                                     // in-thread high bit
  constexpr uint32_t LOM = (1u << (6 - GL)) - 1u;                // base bits inside the 6-bit "lane" part of an index
  constexpr uint32_t BMASK = (1u << BB) - 1u;
  struct alignas(sizeof(T) * NJ) group_t { T v[NJ]; };
  struct alignas(sizeof(T) * NL) lvec_t { T v[NL]; };
  extern __shared__ __align__(16) unsigned char smem[];
  Desc& d = *reinterpret_cast<Desc*>(smem);
  T* yt = reinterpret_cast<T*>(smem + DESC_PAD);
  T* Ltab = yt + (1 << TB) + NJ;                                 // yt[1 << TB ...]: a zero group, target of padded reads
  T* Urow = Ltab + maxk * 64;
  T* Utab = Urow + maxk * 64;
  T* thc = Utab + maxk * 64;
  T* hx = thc + maxk * maxk;
  uint32_t* pxt = reinterpret_cast<uint32_t*>(hx + maxk);       // 384 pext entries + 3 tile-uniform ones
  uint16_t* pml = reinterpret_cast<uint16_t*>(pxt + 400);       // base states sorted by popcount (2^BB entries)
  T* dl = reinterpret_cast<T*>(pml + (1 << BB));                // this tile's slices of the dP / dM tables (dl_cap entries)
  static_assert(NJ == 4, "the four-lanes-per-group in-tile solve is written for 4 states per thread");
  const int tid0 = threadIdx.x;
  int tid = tid0;
  const int prob = plist ? plist[blockIdx.x] : (int)blockIdx.x;   // (plist: the batch's problems that stay on the tile kernels)
  load_desc(&d, descs + prob);
  __syncthreads();
  const int k = sgpr(d.k);
  constexpr int t = TB;
  constexpr uint32_t tmask = (1u << TB) - 1u;
  const long long base = sgpr64(d.off);
  const long long toff = sgpr64(d.toff);
  const uint32_t maskP = sgpr(d.maskP), maskM = sgpr(d.maskM);
  {
    const T* src = tab + toff;
    for (int e = tid; e < k * k; e += TSB) thc[e] = src[e];
    for (int e = tid; e < k * 64; e += TSB) { Ltab[e] = src[k * k + e]; Urow[e] = src[k * k + k * 64 + e]; }
  }
  const uint32_t pairP = sgpr(d.pairP), lone = sgpr(d.lone);
  const int seedb = sgpr(d.seedbit);
  const uint32_t cP = maskP & tmask, cM = maskM & tmask;
  if (tid < NJ) yt[(1 << TB) + tid] = T(0);
  pml[tid] = perm[(size_t)BB * (1 << TB) + tid];
  if (tid < 256) {
    const int which = tid >> 7, half = (tid >> 6) & 1, v = tid & 63;
    const uint32_t m = which == 0 ? cP : cM;
    pxt[tid] = half == 0 ? pext32((uint32_t)v, m & 63u) : (pext32((uint32_t)v << 6, m & ~63u) << __popc(m & 63u));
  }
  const T* dP = tab + toff + rate_table_size(k);
  const T* dM = dP + (1ll << __popc(maskP));
  const T* dE = dM + (1ll << __popc(maskM));
  const int nPin = __popc(cP), nMin = __popc(cM);
  // DLOK (chosen by the host for the launch): every patient's dP / dM slices of a tile fit the dl area of LDS
  const bool dl_ok = DLOK || (1 << nPin) + (1 << nMin) <= dl_cap;
  const int t0 = pt_off[prob], ntile = pt_off[prob + 1] - t0;
  JLink<T> Lk;
  if (rhs_mode == 3) Lk = links[prob];
  __syncthreads();
  // ---- seed = 0 part: lattice over the paired events (see k_psolve)
  const int ke = __popc(pairP);
  const bool eq_block = ke <= TB;
  auto solve_eq_block = [&]() {
    const uint32_t VE = 1u << ke;
    const T seed_base = thc[seedb * k + seedb];
    for (int s = 0; s <= ke; ++s) {
      const int level = TR ? ke - s : s;
      for (uint32_t e = tid0; e < VE; e += TSB) {
        if (__popc(e) != level) continue;
        const uint32_t xp = pdep32(e, pairP);
        const uint32_t x0 = xp | (xp << 1);
        T z = (!TR && e == 0) ? e0_scale<T>() : T(0);
        if (!TR) {
          for (uint32_t m = xp; m; m &= m - 1) {
            const int bP = __ffs(m) - 1;
            T r = thc[bP * k + bP];
            for (uint32_t m2 = xp & ~(1u << bP); m2; m2 &= m2 - 1) r *= thc[bP * k + (__ffs(m2) - 1)];
            z += r * yt[pext32(xp & ~(1u << bP), pairP)];
          }
        } else {
          for (uint32_t m = pairP & ~xp; m; m &= m - 1) {
            const int bP = __ffs(m) - 1;
            T r = thc[bP * k + bP];
            for (uint32_t m2 = xp; m2; m2 &= m2 - 1) r *= thc[bP * k + (__ffs(m2) - 1)];
            z += r * yt[pext32(xp | (1u << bP), pairP)];
          }
          T rs = seed_base;
          for (uint32_t m2 = xp; m2; m2 &= m2 - 1) rs *= thc[seedb * k + (__ffs(m2) - 1)];
          z += rs * y[base + (x0 | (1u << seedb))];
        }
        const T v = z / dE[e];
        yt[e] = v;
        y[base + x0] = v;
      }
      __syncthreads();
    }
  };
  STAMP_DECL;
  STAMP_START;
  if (eq_block && !TR) solve_eq_block();
  STAMP(7);
  uint32_t Hprev = 0xffffffffu;                                 // tile whose solution yt still holds
  for (int it = 0; it < ntile; ++it) {
    STAMP_START;
    const uint32_t H = ptiles[t0 + (TR ? ntile - 1 - it : it)];
    const uint32_t xhi = H << t;
    tid = tid0;
    asm volatile("" : "+v"(tid));                     // nothing thread-dependent stays live across tiles
    // natural base of this thread (global traffic): states  jl | (tid << GL) | (jh << HB)
    const uint32_t nlo = ((uint32_t)tid & LOM) << GL, nhi = (uint32_t)tid >> (6 - GL);
    // ---- per tile: slices of the diagonal tables (land in LDS behind the next barrier), tile-bit factors
    // (up to 2^10 + 2^4 entries: thread tid takes entry tid and, for the few beyond the workgroup size, tid + TSB)
    // ---- single-bit moves above the tile (step A): scalar bit list of the tile index.  With MMHN_Q_EARLY the first
    // MMHN_Q_TPA neighbour tiles that come from HBM are requested here - they were solved at least two tiles ago, their
    // stores were waited for at the previous tile's barrier - and fly through the set-up barriers below
    uint32_t mb = (TR ? ~H : H) & ((1u << (k - t)) - 1u) & ~(1u << (seedb - t));
    const uint32_t dprev = H ^ Hprev;
    const bool prev_in_lds = Hprev != 0xffffffffu && (dprev & (dprev - 1)) == 0 && (dprev & mb);
    if (prev_in_lds) mb &= ~dprev;
    constexpr int TPA = MMHN_Q_TPA;
    auto nbr_fetch = [&](int (&bq)[TPA], lvec_t (&nv)[TPA][2]) {
#pragma unroll
      for (int q = 0; q < TPA; ++q) {
        const bool on = mb != 0;                               // wave-uniform
        bq[q] = on ? t + __ffs(mb) - 1 : -1;
        mb &= mb - 1;
        if (on) {
#ifdef MMHN_ABL_FAKE_NBR      // timing-only ablation (wrong results): neighbour reads served by L2 instead of HBM
          const T* yn = y + base + ((uint32_t)tid << GL);
#else
          const T* yn = y + base + (xhi ^ (1u << bq[q])) + ((uint32_t)tid << GL);
#endif
          nv[q][0] = *reinterpret_cast<const lvec_t*>(yn);
          nv[q][1] = *reinterpret_cast<const lvec_t*>(yn + (1u << HB));
        }
      }
    };
#if MMHN_Q_EARLY
    int bq0[TPA];
    lvec_t nv0[TPA][2];
    nbr_fetch(bq0, nv0);
#endif
    T dval = 0, dval2 = 0;
    const int ndl = dl_ok ? (1 << nPin) + (1 << nMin) : 0;
    auto dl_fetch = [&](int e) {
      const bool isM = e >= (1 << nPin);
      const uint32_t m = isM ? maskM : maskP;
      const uint32_t hi = pext32(xhi, m & ~tmask) << __popc(m & tmask);
      return isM ? dM[hi | (uint32_t)(e - (1 << nPin))] : dP[hi | (uint32_t)e];
    };
    if (tid < ndl) dval = dl_fetch(tid);
    if (tid + TSB < ndl) dval2 = dl_fetch(tid + TSB);
    if (tid < k) {
      T h = thc[tid * k + tid];
      for (int bb = t; bb < k; ++bb) if (bb != tid && ((H >> (bb - t)) & 1u)) h *= thc[tid * k + bb];
      hx[tid] = h;
    } else if (tid >= 64 && tid < 66) {
      const uint32_t m = tid == 64 ? maskP : maskM;
      pxt[384 + tid - 64] = pext32(xhi, m & ~tmask) << __popc(m & tmask);
    }
    __syncthreads();
    STAMP(0);
    for (int e = tid; e < k * 64; e += TSB) Utab[e] = Urow[e] * hx[e >> 6];
    if (tid < ndl) dl[tid] = dval;
    if (tid + TSB < ndl) dl[tid + TSB] = dval2;
    const uint32_t hP = pxt[384], hM = pxt[385];
    // ---- right-hand side (natural states).  Forward: e_0 lies in the seed = 0 part, so zero here.
#ifdef MMHN_ABL_PACK
    for (int rep = 0; rep < MMHN_ABL_PACK; ++rep) {
#endif
    T acc[NJ];
#pragma unroll
    for (int jq = 0; jq < NJ; ++jq) acc[jq] = T(0);
    if (rhs_mode == 3) {
      const bool can0 = Lk.soff[0] >= 0 && ((xhi & maskP & ~tmask) == (maskP & ~tmask));
      const bool can1 = Lk.soff[1] >= 0 && ((xhi & maskM & ~tmask) == (maskM & ~tmask));
      if (can0 || can1) {                              // tile-uniform; most tiles have neither
#pragma unroll
        for (int jq = 0; jq < NJ; ++jq) {
          const uint32_t lo = nlo | (uint32_t)(jq & (NL - 1)), hi6 = nhi | ((uint32_t)(jq >> GL) << 5);
          const uint32_t xl = lo | (hi6 << 6);
          T rv = 0;
          if (can0 && (xl & cP) == cP) rv += Lk.cst[0] * qS[Lk.soff[0] + (1ll << (Lk.sk[0] - 1)) + (hM | pxt[128 + lo] | pxt[192 + hi6])];
          if (can1 && (xl & cM) == cM) rv += Lk.cst[1] * qS[Lk.soff[1] + (1ll << (Lk.sk[1] - 1)) + (hP | pxt[lo] | pxt[64 + hi6])];
          acc[jq] = rv;
        }
      }
    }
    __syncthreads();                                   // Utab complete; the previous tile's stores have landed
    STAMP(1);
    // ---- step A: single-bit moves above the tile apply to every state
    {
      auto add_move = [&](int b, const T (&nf)[NJ]) {
        const lvec_t Lv = *reinterpret_cast<const lvec_t*>(Ltab + b * 64 + nlo);
        const T U0 = Utab[b * 64 + nhi], U1 = Utab[b * 64 + nhi + 32];
#pragma unroll
        for (int jq = 0; jq < NJ; ++jq) acc[jq] += Lv.v[jq & (NL - 1)] * ((jq >> GL) ? U1 : U0) * nf[jq];
      };
      auto nbr_take = [&](const int (&bq)[TPA], const lvec_t (&nv)[TPA][2]) {
#pragma unroll
        for (int q = 0; q < TPA; ++q) {
          if (bq[q] >= 0) {
            T nf[NJ];
#pragma unroll
            for (int jq = 0; jq < NJ; ++jq) nf[jq] = nv[q][jq >> GL].v[jq & (NL - 1)];
            add_move(bq[q], nf);
          }
        }
      };
      // the tile this workgroup solved last is still in LDS (this thread's own group)
      if (prev_in_lds) {
        const group_t gq = *reinterpret_cast<const group_t*>(yt + ((uint32_t)tid << G));
        T nf[NJ];
#pragma unroll
        for (int jq = 0; jq < NJ; ++jq) nf[jq] = gq.v[jq];
        add_move(t + __ffs(dprev) - 1, nf);
      }
      // the others stream from HBM, TPA neighbour tiles in flight per thread
#if MMHN_Q_EARLY
      nbr_take(bq0, nv0);
#endif
      while (mb) {
        int bq[TPA];
        lvec_t nv[TPA][2];
        nbr_fetch(bq, nv);
        nbr_take(bq, nv);
      }
      if (!TR && seed_move_possible(lone, pairP, xhi, tmask)) {
        // seeding into this tile: only the PT == MT states of the seed = 0 part carry values (and only they were
        // written), everything else is discarded by the select
        const lvec_t Lv = *reinterpret_cast<const lvec_t*>(Ltab + seedb * 64 + nlo);
        const T U0 = Utab[seedb * 64 + nhi], U1 = Utab[seedb * 64 + nhi + 32];
        const T* yn = y + base + (xhi ^ (1u << seedb)) + ((uint32_t)tid << GL);
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
          const lvec_t nv = *reinterpret_cast<const lvec_t*>(yn + ((uint32_t)jh << HB));
#pragma unroll
          for (int jl = 0; jl < NL; ++jl) {
            const uint32_t x = xhi | (uint32_t)jl | ((uint32_t)tid << GL) | ((uint32_t)jh << HB);
            const bool e0x = ((x & lone) == 0) && (((x & pairP) << 1) == (x & (pairP << 1)));
            const T term = Lv.v[jl] * (jh ? U1 : U0) * nv.v[jl];
            acc[jl | (jh << GL)] += e0x ? term : T(0);
          }
        }
      }
    }
    {
      group_t gq;
#pragma unroll
      for (int jq = 0; jq < NJ; ++jq) gq.v[jq] = acc[jq];
      *reinterpret_cast<group_t*>(yt + ((uint32_t)tid << G)) = gq;
    }
    STAMP(2);
    __syncthreads();
#ifdef MMHN_ABL_PACK
    }
#endif
    STAMP(3);
    // ---- step B: levels over the popcount of the base state.  HP lanes of one DPP quad share a group (HP = 1, 2, 4;
    // a level has at most C(BB, BB/2) groups, a quarter of the workgroup): each lane takes every HP-th base-bit move,
    // the partial sums meet in a quad butterfly, each lane supplies NJ / HP inverse diagonals and in-group
    // coefficients, all finish the group redundantly (a few FMAs) and store their own states.
    {
      constexpr int HP = MMHN_Q_LANES, LHP = HP == 4 ? 2 : HP == 2 ? 1 : 0;
      static_assert(HP == 1 || HP == 2 || HP == 4, "lanes per group");
      constexpr int OWN = NJ / HP;                                  // states (and coefficients) a lane supplies
      const uint32_t gi = (uint32_t)tid >> LHP, slot = (uint32_t)tid & (uint32_t)(HP - 1);
      constexpr BinomPrefix<BB> BP{};
      for (int s = 0; s <= BB; ++s) {
        const int level = TR ? BB - s : s;
        const uint32_t goff = BP.v[level], gcnt = BP.v[level + 1] - goff;
#ifdef MMHN_ABL_PACK
        for (uint32_t item = gi; item < gcnt * MMHN_ABL_PACK; item += (uint32_t)(TSB >> LHP)) {
          uint32_t gsel = item;
          while (gsel >= gcnt) gsel -= gcnt;
          const uint32_t ub = pml[goff + gsel];
#else
        if (gi < gcnt) {
          const uint32_t ub = pml[goff + gi];
#endif
          const uint32_t ulo = (ub & LOM) << GL, uhi = ub >> (6 - GL);
          // this lane's share of the group's right-hand side
          T z[NJ];
          {
            const group_t zg = *reinterpret_cast<const group_t*>(yt + (slot == 0 ? (ub << G) : (1u << TB)));
#pragma unroll
            for (int jq = 0; jq < NJ; ++jq) z[jq] = zg.v[jq];
          }
          // the lane's own states: position in the tile slices of the diagonal tables; its in-group coefficients
          // coefficient ids: 0 = bit 0 out of (0, jh 0), 1 = bit 0 out of (0, jh 1), 2 = bit HB out of (jl 0, 0), 3 = out of (jl 1, 0)
          // (all index look-ups of the group first, then everything that hangs on them: one LDS round trip each instead
          // of one per state)
          T dsum[OWN], cmine[OWN];
          uint32_t iP[OWN], iM[OWN];
#pragma unroll
          for (int o = 0; o < OWN; ++o) {
            const uint32_t jq = slot + (uint32_t)(o * HP);
            const uint32_t slo = ulo | (jq & (NL - 1)), shi = uhi | ((jq >> GL) << 5);
            iP[o] = pxt[slo] | pxt[64 + shi];
            iM[o] = pxt[128 + slo] | pxt[192 + shi];
          }
          auto coefs = [&]() {
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
              const uint32_t jq = slot + (uint32_t)(o * HP);
              const uint32_t cbit = jq < 2 ? 0u : (uint32_t)HB;
              cmine[o] = Ltab[cbit * 64 + ulo + (jq == 3 ? 1u : 0u)] * Utab[cbit * 64 + uhi + (jq == 1 ? 32u : 0u)];
            }
          };
          constexpr bool CLATE = MMHN_Q_CLATE < 0 ? HP == 1 : MMHN_Q_CLATE != 0;   // coefficients after the move loop (registers)
          if (!CLATE) coefs();
#pragma unroll
          for (int o = 0; o < OWN; ++o)
            dsum[o] = (DLOK || dl_ok) ? dl[iP[o]] + dl[(1 << nPin) + iM[o]] : dP[hP | iP[o]] + dM[hM | iM[o]];
          uint32_t todo = TR ? (~ub & BMASK) : ub;
#pragma unroll
          for (int i = 1; i < HP; ++i) if (slot >= (uint32_t)i) todo &= todo - 1;
          constexpr int TPT = MMHN_Q_TPT;                          // base-bit moves per trip
          while (todo) {
            lvec_t Lv[TPT];
            T Uv[TPT], tv[TPT];
            group_t yq[TPT];
#pragma unroll
            for (int q = 0; q < TPT; ++q) {
              const bool on = todo != 0;
              const int bb = on ? __ffs(todo) - 1 : 0;
#pragma unroll
              for (int i = 0; i < HP; ++i) todo &= todo - 1;       // the lane's moves are HP apart (0 stays 0)
              const int bx = bb + GL;
              const uint32_t un = ub ^ (1u << bb);                 // neighbour group
              const uint32_t us = TR ? ub : un;                    // source state of the transition: indexes the rate
              Lv[q] = *reinterpret_cast<const lvec_t*>(Ltab + bx * 64 + ((us & LOM) << GL));
              Uv[q] = Utab[bx * 64 + (us >> (6 - GL))];
              tv[q] = thc[bx * k + HB];
              yq[q] = *reinterpret_cast<const group_t*>(yt + (on ? (un << G) : (1u << TB)));
            }
            asm volatile("" ::: "memory");                         // all LDS reads of the trip in flight before the first use
#pragma unroll
            for (int q = 0; q < TPT; ++q) {
              const T u1 = Uv[q] * tv[q];
#pragma unroll
              for (int jq = 0; jq < NJ; ++jq) z[jq] += Lv[q].v[jq & (NL - 1)] * ((jq >> GL) ? u1 : Uv[q]) * yq[q].v[jq];
            }
          }
          if (CLATE) coefs();
          // quad butterfly: every lane of the cluster gets the full sums
          if (HP >= 2) {
#pragma unroll
            for (int jq = 0; jq < NJ; ++jq) z[jq] += quad_xor<0xB1>(z[jq]);
          }
          if (HP >= 4) {
#pragma unroll
            for (int jq = 0; jq < NJ; ++jq) z[jq] += quad_xor<0x4E>(z[jq]);
          }
          T lidf[NJ], cf[NJ];
#pragma unroll
          for (int o = 0; o < OWN; ++o) {
            const T lm = fast_rcp(dsum[o]);
#pragma unroll
            for (int sl = 0; sl < HP; ++sl) {
              lidf[o * HP + sl] = cluster_bcast<HP>(lm, sl);
              cf[o * HP + sl] = cluster_bcast<HP>(cmine[o], sl);
            }
          }
          const T c00 = cf[0], c01 = cf[1], ch0 = cf[2], ch1 = cf[3];
          // group slots: 0 = (jl 0, jh 0), 1 = (1, 0), 2 = (0, 1), 3 = (1, 1)
          T yv[NJ];
          if (!TR) {
            yv[0] = lidf[0] * z[0];
            yv[1] = lidf[1] * (z[1] + c00 * yv[0]);
            yv[2] = lidf[2] * (z[2] + ch0 * yv[0]);
            yv[3] = lidf[3] * (z[3] + c01 * yv[2] + ch1 * yv[1]);
          } else {
            yv[3] = lidf[3] * z[3];
            yv[2] = lidf[2] * (z[2] + c01 * yv[3]);
            yv[1] = lidf[1] * (z[1] + ch1 * yv[3]);
            yv[0] = lidf[0] * (z[0] + c00 * yv[1] + ch0 * yv[2]);
          }
          if (HP == 1) {
            group_t og;
#pragma unroll
            for (int jq = 0; jq < NJ; ++jq) og.v[jq] = yv[jq];
            *reinterpret_cast<group_t*>(yt + (ub << G)) = og;
          } else if (HP == 2) {
            yt[(ub << G) + slot] = slot == 0 ? yv[0] : yv[1];
            yt[(ub << G) + slot + 2] = slot == 0 ? yv[2] : yv[3];
          } else {
            yt[(ub << G) + slot] = slot == 0 ? yv[0] : slot == 1 ? yv[1] : slot == 2 ? yv[2] : yv[3];
          }
        }
        __syncthreads();
      }
    }
    STAMP(4);
    // ---- step C: the thread's natural group leaves as 2^GL adjacent states per store
#ifdef MMHN_ABL_NO_STORE       // timing-only ablation (wrong results): every tile is stored over the patient's first tile
    const uint32_t xst = 0;
#else
    const uint32_t xst = xhi;
#endif
#ifdef MMHN_ABL_PACK
    for (int rep = 0; rep < MMHN_ABL_PACK; ++rep)
#endif
    {
      const group_t gq = *reinterpret_cast<const group_t*>(yt + ((uint32_t)tid << G));
      T* yo = y + base + xst + ((uint32_t)tid << GL);
#ifdef MMHN_ABL_PACK
      asm volatile("" ::: "memory");
#endif
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        lvec_t ov;
#pragma unroll
        for (int jl = 0; jl < NL; ++jl) ov.v[jl] = gq.v[jl | (jh << GL)];
        *reinterpret_cast<lvec_t*>(yo + ((uint32_t)jh << HB)) = ov;
      }
    }
    Hprev = H;
    __builtin_amdgcn_s_waitcnt(0xc07f);                // LDS-only barrier; the stores are waited for before the next tile's neighbour loads
    __builtin_amdgcn_s_barrier();
    STAMP(5);
  }
  STAMP_START;
  if (eq_block && TR) {
    __syncthreads();
    solve_eq_block();
  }
  STAMP(7);
  STAMP_FLUSH(TR ? 8 : 0);
}

}  // namespace mmhn
