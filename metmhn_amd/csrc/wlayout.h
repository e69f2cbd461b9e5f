// The WINDOW layout of the seeded half of a joint state vector (round 4; produced by csrc/wsolve.h, read by the
// small-space kernels, the eq-block flows and csrc/wclass.h).  A seeded state is (S, T) = (row-class subset, column-
// class subset), the row class being the tumour class with more bits:
//     S = l | w << 6 | Sx << 10        l: lane bits, w: wave bits, Sx: external row bits
//     T = c | beta << RB | Tx << (RB + HB)     c: column inside a block, beta: block inside a window, Tx: external
//     Sigma = Tx | Sx << nXc            external index
//     position(S, T) = (((Sigma * H + beta) * 1024 + rho(w, l)) * NC + c
// rho sorts the rows of a block by lane-level (popcount of l).  The seed = 0 half keeps its natural positions.
// Reference: the index convention being re-ordered is kronvec.py:223-250 (active slots in slot order = index bits).
#pragma once
#include <hip/hip_runtime.h>
#include "desc.h"

namespace mmhn {

__host__ __device__ inline int popc32(uint32_t v) {
#ifdef __HIP_DEVICE_COMPILE__
  return __popc(v);
#else
  return __builtin_popcount(v);
#endif
}
__host__ __device__ inline uint32_t pext32(uint32_t x, uint32_t mask) {
  uint32_t out = 0, pos = 0;
  while (mask) {
    const uint32_t low = mask & (0u - mask);
    if (x & low) out |= 1u << pos;
    ++pos;
    mask ^= low;
  }
  return out;
}
__host__ __device__ inline uint32_t pdep32(uint32_t v, uint32_t mask) {
  uint32_t out = 0;
  while (mask) {
    const uint32_t low = mask & (0u - mask);
    if (v & 1u) out |= low;
    v >>= 1;
    mask ^= low;
  }
  return out;
}

constexpr int WLB = 6, WWB = 4, WTB = WLB + WWB;   // lane bits, wave bits, thread bits
constexpr int WROWS = 1 << WTB;                    // rows of a block = threads of a workgroup
constexpr int WMAXB = 20;                          // array bound of the per-bit lists of a WDesc

template <typename T> struct WCfg;
// RB: column bits of a block, HB: blocks of a window, KC: most column-class bits (their tables live in LDS),
// KR: most row-class bits (the ten thread bits + external row bits), KE: most paired events (the seed = 0 lattice of
// 2^KE states is solved in LDS),
// NXT: the number of external bits the solve kernel is also built for as a compile-time constant (k = 20 fp64: 5, k = 25 fp32: 9),
// PAD: elements between the table rows of two external column settings (16 bytes: the lanes of a wave differ in it)
// FACT: the rates of the column-class events are kept as two factors, [event][window setting] x [event][external column
//       setting] (one more multiply per fetched rate vector, 8 KB instead of kC * 2^(kC-1) entries): what lets twelve
//       column bits fit.  fp64 keeps the full tables (nine column bits cover every k = 20 shape).
// REV: order in which a step requests its external blocks (wsolve.h).
// ON: the engine uses the window path for this dtype.
#ifndef MMHN_WHB
#define MMHN_WHB 2    // build switch (experiments): log2 of the blocks of a window, fp64
#endif
#ifndef MMHN_WNXT_D
#define MMHN_WNXT_D (7 - MMHN_WHB)   // build switch (experiments): -2 = every fp64 chain on the generic instantiation
#endif
template <> struct WCfg<double> {
  static constexpr int RB = 2, HB = MMHN_WHB, KC = 9, KR = 15, KE = 9, PAD = 2, NXT = MMHN_WNXT_D;
  static constexpr bool FACT = false, REV = true, ON = true;
};
#ifndef MMHN_WF32
#define MMHN_WF32 1   // build switch: 0 = fp32 cohorts on the tile kernels (k_psolve2 / k_pclass) only
#endif
template <> struct WCfg<float> {
  static constexpr int RB = 3, HB = 2, KC = 12, KR = 18, KE = 12, PAD = 4, NXT = 9;
  static constexpr bool FACT = true, REV = false, ON = MMHN_WF32 != 0;
};

// static description of one joint problem on the window path (host-built, set_cohort)
struct WDesc {
  int prob;                  // index into the batch's joint descriptors
  int kR, kC, majP;          // row-class bits, column-class bits, 1: the row class is P
  int nXc, nXr;              // external column bits (kC - RB - HB), external row bits (kR - 10)
  uint32_t rowmask, colmask; // natural index bits of the two classes
  uint32_t pairRowC;         // compact row bits whose event is also active in the other tumour
  uint32_t loneRowC;         // compact row bits whose partner slot is inactive
  int8_t rb[WMAXB];          // natural bit of row bit i
  int8_t cb[WMAXB];          // natural bit of column bit i
  int8_t prt[WMAXB];         // column bit of the partner of row bit i, -1: none
};

// rows of a block sorted by lane-level: rho(w, l) = 16 * (rows of lower levels) + w * C(6, m) + rank of l in its level
struct W6 {
  uint8_t rank[64], order[64], off[8], cnt[8];
};
constexpr W6 make_w6() {
  W6 t{};
  int pos = 0;
  for (int m = 0; m <= 6; ++m) {
    t.off[m] = (uint8_t)pos;
    int r = 0;
    for (int l = 0; l < 64; ++l) {
      int pc = 0;
      for (int b = 0; b < 6; ++b) pc += (l >> b) & 1;
      if (pc == m) { t.rank[l] = (uint8_t)r++; t.order[pos++] = (uint8_t)l; }
    }
    t.cnt[m] = (uint8_t)r;
  }
  t.off[7] = 64; t.cnt[7] = 0;
  return t;
}
__host__ __device__ inline uint32_t wrho(uint32_t w, uint32_t l) {
  constexpr W6 t = make_w6();
  const int m = popc32(l);
  return 16u * t.off[m] + w * t.cnt[m] + t.rank[l];
}
// inverse: row index w << 6 | l of storage row r
__host__ __device__ inline uint32_t wrho_inv(uint32_t r) {
  constexpr W6 t = make_w6();
  int m = 0;
  while (r >= 16u * t.off[m + 1]) ++m;
  const uint32_t q = r - 16u * t.off[m];
  return ((q / t.cnt[m]) << 6) | t.order[t.off[m] + q % t.cnt[m]];
}
template <typename T>
__host__ __device__ inline long long wpos(uint32_t Sigma, uint32_t beta, uint32_t rho, uint32_t c) {
  return ((((long long)((Sigma << WCfg<T>::HB) | beta) << WTB) + rho) << WCfg<T>::RB) + c;
}
// position of the seeded natural state x (seeding bit stripped) inside the seeded half
template <typename T>
__host__ __device__ inline long long wpos_nat(const WDesc& w, uint32_t x) {
  constexpr int RB = WCfg<T>::RB, HB = WCfg<T>::HB;
  const uint32_t S = pext32(x, w.rowmask), Tc = pext32(x, w.colmask);
  const uint32_t Sigma = (Tc >> (RB + HB)) | ((S >> WTB) << w.nXc);
  return wpos<T>(Sigma, (Tc >> RB) & ((1u << HB) - 1u), wrho((S >> WLB) & ((1u << WWB) - 1u), S & 63u), Tc & ((1u << RB) - 1u));
}

// element offset (inside the whole 2^k vector) of the seeded state whose FREE class has compact index f while every bit
// of the other class is set - the states a marginal problem of a paired row reads and feeds (likelihood.py:573-575,
// 617-618); free_is_row: the free class is the row class
template <typename T>
__host__ __device__ inline long long wpos_marg(const WDesc& w, int k, bool free_is_row, uint32_t f) {
  constexpr int RB = WCfg<T>::RB, HB = WCfg<T>::HB;
  const long long half = 1ll << (k - 1);
  if (free_is_row) {
    const uint32_t Sigma = ((1u << w.nXc) - 1u) | ((f >> WTB) << w.nXc);
    return half + wpos<T>(Sigma, (1u << HB) - 1u, wrho((f >> WLB) & ((1u << WWB) - 1u), f & 63u), (1u << RB) - 1u);
  }
  const uint32_t Sigma = (f >> (RB + HB)) | (((1u << w.nXr) - 1u) << w.nXc);
  return half + wpos<T>(Sigma, (f >> RB) & ((1u << HB) - 1u), (uint32_t)WROWS - 1u, f & ((1u << RB) - 1u));
}

}  // namespace mmhn
