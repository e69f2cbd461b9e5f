// Joint triangular solves in the class-sorted MATRIX layout (round 3).
//
// On the seed = 1 half of a paired patient's space the operator is a Kronecker sum, D - Q = A_P (+) A_M
// (reference: metmhn/jx/likelihood.py:231-262 solves it with k+1 Jacobi sweeps of
// kronvec.py:499-539).  The engine owns pi / q_J, so the seeded half is stored as a matrix instead of in
// index order: the class with more bits is the MAJOR class, its MT = 10 lowest bits are the THREAD bits
// (a row t of 1024), every other bit is a SERIAL bit - the minor class first, then the major bits that are
// left over - and
//     position(x) = 2^(k-1) + (s << MT) + rank(t),     s = serial index, rank = rows sorted by popcount.
// One 1024-thread workgroup solves one patient, thread = row:
//   * moves along serial bits stay inside the thread: the predecessor is a value the thread itself produced
//     (registers inside a block of 8 columns, its own earlier stores otherwise), and because every wave works on
//     ONE column block at a time the minor-class rates are wave-uniform and come through the scalar unit;
//   * moves along thread bits cross rows: rows of popcount level l only depend on level l - 1, so level l runs
//     one column block behind level l - 1 (a software pipeline over the levels, one LDS-only barrier per step,
//     a two-slot ring of finished blocks in LDS).  Rows of one level are independent: no per-state index
//     arithmetic, no lane exchange, no rate look-ups per term - a term is one LDS read and one FMA with a
//     per-row constant.
// The seed = 0 half keeps its natural positions (only the PT == MT states carry values there).
#pragma once
#include "kernels.h"
#include <vector>
#ifndef MMHN_MT
#define MMHN_MT 10
#endif
#define MMHN_MT_VALUE MMHN_MT

namespace mmhn {

constexpr int MT = MMHN_MT_VALUE;      // thread bits
constexpr int MROWS = 1 << MT;         // rows = threads of a workgroup
constexpr int MRB = 3;                 // most register bits (columns of a block = 2^RB, MCfg)
constexpr int MKE = 9;                 // most paired events (eq block of 2^MKE states in LDS)

// static description of one joint problem on the matrix path (host-built, set_cohort)
struct MDesc {
  int prob;                 // index into the batch's joint descriptors
  int kmin, nml, kS;        // minor-class bits, major bits left over, serial bits (kmin + nml)
  int majP;                 // 1: the major class is P
  uint32_t tmask, mlmask, minmask;   // natural index bits of the three groups
  int tb[MT];               // natural bit of thread bit i
  int mlb[8];               // natural bit of left-over bit i
  int mnb[24];              // natural bit of minor bit i
};

// can this joint problem run on the matrix path?
inline bool matrix_ok(const Desc& d, int nml_max, int kmin_max) {
  if (d.mode != JOINT || d.seedbit != d.k - 1) return false;
  const int kP = popc(d.maskP), kM = popc(d.maskM);
  const int kmaj = kP >= kM ? kP : kM, kmn = kP >= kM ? kM : kP;
  return kmaj >= MT && kmaj - MT <= nml_max && kmn >= MRB && kmn <= kmin_max && popc(d.pairP) <= MKE;
}
inline MDesc make_mdesc(const Desc& d, int prob) {
  MDesc m{};
  m.prob = prob;
  const int kP = popc(d.maskP), kM = popc(d.maskM);
  m.majP = kP >= kM ? 1 : 0;
  const uint32_t maj = m.majP ? d.maskP : d.maskM, mn = m.majP ? d.maskM : d.maskP;
  int nt = 0, nl = 0, nn = 0;
  for (int b = 0; b < d.k; ++b) {
    if ((maj >> b) & 1u) {
      if (nt < MT) { m.tb[nt++] = b; m.tmask |= 1u << b; }
      else { m.mlb[nl++] = b; m.mlmask |= 1u << b; }
    } else if ((mn >> b) & 1u) { m.mnb[nn++] = b; m.minmask |= 1u << b; }
  }
  m.kmin = nn; m.nml = nl; m.kS = nn + nl;
  return m;
}

// rows sorted by popcount (ascending inside a level): rowT[rank] = t, rankT[t] = rank
inline void matrix_rows(uint16_t* rowT, uint16_t* rankT) {
  int pos = 0;
  for (int l = 0; l <= MT; ++l)
    for (int t = 0; t < MROWS; ++t)
      if (popc((uint32_t)t) == l) { rowT[pos] = (uint16_t)t; rankT[t] = (uint16_t)pos; ++pos; }
}

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): LDS only, vector memory stays in flight
  __builtin_amdgcn_s_barrier();
}

template <typename T> __device__ __forceinline__ T fma_m(T a, T b, T c);
template <> __device__ __forceinline__ double fma_m<double>(double a, double b, double c) { return fma(a, b, c); }
template <> __device__ __forceinline__ float fma_m<float>(float a, float b, float c) { return fmaf(a, b, c); }

// ---- work decomposition of a workgroup: 8 waves, each owns up to 3 UNITS = at most 64 rows of ONE popcount level
// (rows of a level are independent of each other, so the lanes of a unit never exchange anything).  The 22 units of
// the 11 levels are dealt so that every wave carries about the same number of thread-bit terms per step in both
// directions (level l next to level 10 - l).
#ifndef MMHN_MT
#define MMHN_MT 10
#endif
#if MMHN_MT == 10
constexpr int MW = 8;                               // 22 units: three per wave, 256 registers per thread
constexpr int MUPW = 3;
#else
constexpr int MW = 14;                              // 14 units: one per wave, 128 registers per thread
constexpr int MUPW = 1;
#endif
constexpr int MTHREADS = MW * 64;
struct MUnit { int level, m0, n; };                 // level < 0: empty slot; rows = ranks m0 .. m0 + n - 1
inline void matrix_units(MUnit* out) {
  std::vector<MUnit> u;
  int m0 = 0;
  for (int l = 0; l <= MT; ++l) {
    int n = 1;
    for (int i = 0; i < l; ++i) n = n * (MT - i) / (i + 1);
    const int parts = (n + 63) / 64;
    for (int i = 0; i < parts; ++i) {
      const int sz = n / parts + (i < n % parts ? 1 : 0);
      u.push_back(MUnit{l, m0, sz});
      m0 += sz;
    }
  }
  for (int i = 0; i < MW * MUPW; ++i) out[i] = MUnit{-1, 0, 0};
#if MMHN_MT == 10
  // unit indices by level: 0:L0 1:L1 2:L2 3,4:L3 5-8:L4 9-12:L5 13-16:L6 17,18:L7 19:L8 20:L9 21:L10
  static const int deal[MW][MUPW] = {{0, 21, 9}, {1, 20, 10}, {2, 19, 11}, {3, 17, 12},
                                     {4, 18, 5}, {6, 13, 14}, {7, 15, 16}, {8, -1, -1}};
  for (int w = 0; w < MW; ++w)
    for (int i = 0; i < MUPW; ++i) if (deal[w][i] >= 0) out[w * MUPW + i] = u[deal[w][i]];
#else
  // one unit per wave; neighbouring wave ids (one SIMD takes waves w, w + 4, ...) get levels from both ends
  for (size_t i = 0; i < u.size() && i < (size_t)MW; ++i) out[i] = u[i];
#endif
}

template <typename T> struct MCfg;
// columns per block 2^RB, left-over bits, prefetched moves, most minor-class bits (their rate table lives in LDS)
template <> struct MCfg<double> { static constexpr int RB = 2, NML = MMHN_MT == 10 ? 4 : 5, PF = MMHN_MT == 10 ? 7 : 5, KD = 9; };
template <> struct MCfg<float> { static constexpr int RB = 3, NML = MMHN_MT == 10 ? 6 : 7, PF = MMHN_MT == 10 ? 6 : 4, KD = 10; };

constexpr int MTHC = (MAXK * MAXK + 7) / 8 * 8;                  // thc area, padded: what follows stays 32-byte aligned
template <typename T>
constexpr size_t msolve_lds() {
  return (size_t)(2 * (1 << MCfg<T>::RB) * MROWS + MTHC + 2 * (1 << MKE) + (MCfg<T>::KD + 1) * (1 << MCfg<T>::KD) +
                  (MCfg<T>::NML + 1) * MROWS) * sizeof(T) + 64 * sizeof(int);
}

template <int I> struct IC { static constexpr int value = I; };

// Scheduling fence for a group of loaded values: the empty asm "redefines" them, so every load of the group is issued
// before it and every use comes after it - the group's LDS reads are in flight together instead of one round trip each
// (hipcc otherwise reuses one destination register and waits after every read).
template <typename T> __device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }
template <typename T> __device__ __forceinline__ void pin4(T* v) {
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}
template <typename T> __device__ __forceinline__ void pin8(T* v) {
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
// (one statement per group of up to 16 values: one wait for the whole group instead of one per value)
template <typename T, int N> __device__ __forceinline__ void pin_all(T (&v)[N]) {
  if constexpr (N == 16) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
  } else if constexpr (N == 8) {
    pin8(v);
  } else if constexpr (N == 4) {
    pin4(v);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) pin(v[i]);
  }
}

// position of the seeded state (serial index s, row rank) inside the seeded half: blocks of NC = 2^RB columns, the
// NC values of one row of a block adjacent (one 32-byte access per row and block)
template <int RB>
__host__ __device__ inline long long mpos(uint32_t s, uint32_t rank) {
  return ((((long long)(s >> RB) << MT) + rank) << RB) + (s & ((1u << RB) - 1u));
}

// ------------------------------------------------------------------------------------
// k_msolve: y = (D - Q)^-1 rhs (TR: transposed) of the joint problems in `mds`, matrix layout.
//   forward: rhs = E0 e_0 (the seed = 0 lattice over the paired events is solved first, seeding carries it
//            into the seeded half);  transposed: rhs = D_obs * scatter(q_S) from `links` (likelihood.py:573-575,
//            617-618), the seed = 0 lattice follows the seeded half.
// Persistent: workgroup b takes problems b, b + gridDim.x, ...
// One step = every unit finishes one block of NC = 2^RB columns: level l works on block step - l (transposed:
// level MT - l, blocks descending).  Inside a step a wave walks its units; while unit u runs its thread-bit terms
// out of LDS, the thread's own earlier blocks that unit u + 1 needs (moves along the serial bits above the
// register bits) are already in flight from L2 / HBM.
// ------------------------------------------------------------------------------------
template <typename T, bool TR>
__global__ __launch_bounds__(MTHREADS) void k_msolve(const Desc* __restrict__ descs, const MDesc* __restrict__ mds, int nm,
                                                     const uint16_t* __restrict__ rowT, const uint16_t* __restrict__ rankT,
                                                     const MUnit* __restrict__ units,
                                                     T* y, const T* __restrict__ tab,
                                                     const JLink<T>* __restrict__ links, const T* __restrict__ qS) {
  constexpr int RB = MCfg<T>::RB, NC = 1 << RB, NML = MCfg<T>::NML, PF = MCfg<T>::PF, KD = MCfg<T>::KD;
  typedef T VecT __attribute__((ext_vector_type(NC)));        // the NC columns of one row of a block
  constexpr uint32_t BLKB = MROWS * sizeof(VecT);             // bytes of one column block
  extern __shared__ __align__(16) unsigned char smem[];
  T* ring = reinterpret_cast<T*>(smem);                       // [2][NC][MROWS] finished blocks
  T* thc = ring + 2 * NC * MROWS;                             // [k][k] effects between index bits (k_prep)
  T* e0 = thc + MTHC;                                         // [2^ke] eq-block solution
  T* se = e0 + (1 << MKE);                                    // [2^ke] forward: seeding inflow of eq state e
  // Rmin[j][smin] = rate of the minor-class event of serial bit j from a state whose minor part is smin (base rate
  // times the effects of the minor-class events that already happened; kronvec.py:299-323 / 370-395)
  T* Rmin = se + (1 << MKE);                                  // [kmin][2^kmin]
  T* dminL = Rmin + KD * (1 << KD);                           // [2^kmin] minor-class part of the diagonal
  T* RLl = dminL + (1 << KD);                                 // [NML + 1][MROWS] rate of the i-th left-over-bit move of a row; row NML: zeros
  int* bits = reinterpret_cast<int*>(RLl + (NML + 1) * MROWS);      // [0..9] tb, [16..23] mlb, [32..55] mnb
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ul[MUPW];                                               // level of unit ui (scalar), -1: none
  // (lanes beyond a unit's rows repeat its first row: same loads, same values to the same addresses - no divergent
  // control flow anywhere in the step loop)
  uint32_t ut[MUPW], urk[MUPW];                               // the row's thread-bit state and its rank
  uint32_t pA[MUPW][(MT + 1) / 2];                                  // ring indices of its neighbour rows, two per register
#pragma unroll
  for (int ui = 0; ui < MUPW; ++ui) {
    const MUnit u = units[wave * MUPW + ui];
    ul[ui] = u.level;
    urk[ui] = (uint32_t)(u.m0 + (lane < u.n ? lane : 0));
    ut[ui] = rowT[urk[ui]];
    uint32_t m = TR ? (~ut[ui] & (uint32_t)(MROWS - 1)) : ut[ui];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      uint32_t a = urk[ui];
      if (m) { a = rankT[ut[ui] ^ (m & (0u - m))]; m &= m - 1; }
      if (j & 1) pA[ui][j >> 1] |= a << 16; else pA[ui][j >> 1] = a;
    }
  }
  STAMP_DECL;
  for (int it = blockIdx.x; it < nm; it += gridDim.x) {
    STAMP_START;
    const MDesc& md = mds[it];
    const int prob = sgpr(md.prob);
    const Desc& d = descs[prob];
    const int k = sgpr(d.k), kmin = sgpr(md.kmin), nml = sgpr(md.nml), kS = sgpr(md.kS);
    const bool majP = sgpr(md.majP) != 0;
    const long long base = sgpr64(d.off), toff = sgpr64(d.toff);
    const int seedb = k - 1;
    const uint32_t Vmin = 1u << kmin;
    const T* dP = tab + toff + rate_table_size(k);
    const T* dM = dP + (1ll << __popc(sgpr(d.maskP)));
    const T* dmajt = majP ? dP : dM;
    const T* dmint = majP ? dM : dP;
    const long long half = 1ll << (k - 1);
    T* ym = y + base + half;                                   // seeded half, matrix layout
    {
      const T* src = tab + toff;
      for (int e = tid; e < k * k; e += MTHREADS) thc[e] = src[e];
      if (tid < MT) bits[tid] = md.tb[tid];
      else if (tid >= 16 && tid < 24) bits[tid] = md.mlb[tid - 16];
      else if (tid >= 32 && tid < 56) bits[tid] = md.mnb[tid - 32];
    }
    __syncthreads();
    for (uint32_t e = tid; e < ((uint32_t)kmin << kmin); e += MTHREADS) {
      const int j = (int)(e >> kmin), nb = bits[32 + j];
      const uint32_t sv = e & (Vmin - 1u);
      T r = thc[nb * k + nb];
      for (int i = 0; i < kmin; ++i) if (i != j && ((sv >> i) & 1u)) r *= thc[nb * k + bits[32 + i]];
      Rmin[e] = r;
    }
    for (uint32_t e = tid; e < Vmin; e += MTHREADS) dminL[e] = dmint[e];
    // rows of zeros: block-bit slots that are not minor-class moves of this patient read their "rate" there
    for (uint32_t e = ((uint32_t)kmin << kmin) + tid; e < ((uint32_t)(PF + RB) << kmin) && e < (uint32_t)(KD << KD); e += MTHREADS) Rmin[e] = T(0);
    for (uint32_t e = tid; e < (uint32_t)MROWS; e += MTHREADS) RLl[NML * MROWS + e] = T(0);
    __syncthreads();
    // ---- seed = 0 part: lattice over the paired events (only PT == MT states carry values)
    auto solve_eq = [&]() {
      const uint32_t pairP = sgpr(d.pairP);
      const uint32_t tmask = sgpr(md.tmask), mlmask = sgpr(md.mlmask), minmask = sgpr(md.minmask);
      const int ke = __popc(pairP);
      const T* dE = dM + (1ll << __popc(sgpr(d.maskM)));
      const uint32_t VE = 1u << ke;
      const T seed_base = thc[seedb * k + seedb];
      for (int s = 0; s <= ke; ++s) {
        const int level = TR ? ke - s : s;
        for (uint32_t e = tid; e < VE; e += MTHREADS) {
          if (__popc(e) != level) continue;
          const uint32_t xp = pdep32(e, pairP);
          const uint32_t x0 = xp | (xp << 1);
          T z = (!TR && e == 0) ? e0_scale<T>() : T(0);
          T rs = seed_base;
          for (uint32_t m2 = xp; m2; m2 &= m2 - 1) rs *= thc[seedb * k + (__ffs(m2) - 1)];
          if (!TR) {
            for (uint32_t m = xp; m; m &= m - 1) {
              const int bP = __ffs(m) - 1;
              T r = thc[bP * k + bP];
              for (uint32_t m2 = xp & ~(1u << bP); m2; m2 &= m2 - 1) r *= thc[bP * k + (__ffs(m2) - 1)];
              z += r * e0[pext32(xp & ~(1u << bP), pairP)];
            }
          } else {
            for (uint32_t m = pairP & ~xp; m; m &= m - 1) {
              const int bP = __ffs(m) - 1;
              T r = thc[bP * k + bP];
              for (uint32_t m2 = xp; m2; m2 &= m2 - 1) r *= thc[bP * k + (__ffs(m2) - 1)];
              z += r * e0[pext32(xp | (1u << bP), pairP)];
            }
            const uint32_t sx = pext32(x0, minmask) | (pext32(x0, mlmask) << kmin);
            z += rs * ym[mpos<RB>(sx, rankT[pext32(x0, tmask)])];
          }
          const T v = z / dE[e];
          e0[e] = v;
          if (!TR) se[e] = rs * v;
          y[base + x0] = v;
        }
        __syncthreads();
      }
    };
    if (!TR) solve_eq();
    const int rowpart = majP ? 0 : 1, colpart = 1 - rowpart;
    // ---- per-row constants of the current setting of the left-over major bits
    T RT[MUPW][MT];                                            // rate of the j-th thread-bit move of the row
    T dmaj[MUPW];
    uint32_t hit_s[MUPW];                                      // forward: the one column of the row seeding enters
    T hitv[MUPW];
    auto update = [&](auto UI, uint32_t sml) {
      constexpr int ui = decltype(UI)::value;
      const uint32_t t = ut[ui];
      uint32_t m = TR ? (~t & (uint32_t)(MROWS - 1)) : t;
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        T r = T(0);
        if (m) {
          const int b = __ffs(m) - 1;
          m &= m - 1;
          const int nb = bits[b];
          const uint32_t tsrc = TR ? t : (t ^ (1u << b));
          r = thc[nb * k + nb];
          for (int i = 0; i < MT; ++i) if ((tsrc >> i) & 1u) r *= thc[nb * k + bits[i]];
          for (int i = 0; i < nml; ++i) if ((sml >> i) & 1u) r *= thc[nb * k + bits[16 + i]];
        }
        RT[ui][j] = r;
      }
      for (int i = 0; i < nml; ++i) {
        const int nb = bits[16 + i];
        T r = thc[nb * k + nb];
        for (int q = 0; q < MT; ++q) if ((t >> q) & 1u) r *= thc[nb * k + bits[q]];
        const uint32_t ssrc = TR ? sml : (sml & ~(1u << i));
        for (int q = 0; q < nml; ++q) if (q != i && ((ssrc >> q) & 1u)) r *= thc[nb * k + bits[16 + q]];
        RLl[i * MROWS + urk[ui]] = r;                          // (read back by the same row only)
      }
      dmaj[ui] = dmajt[t | (sml << MT)];
      if (!TR) {
        const uint32_t pairP = sgpr(d.pairP), lone = sgpr(d.lone);
        const uint32_t pairMaj = majP ? pairP : (pairP << 1), pairMin = majP ? (pairP << 1) : pairP;
        const uint32_t xmaj = pdep32(t, sgpr(md.tmask)) | pdep32(sml, sgpr(md.mlmask));
        hit_s[ui] = 0xffffffffu;
        hitv[ui] = T(0);
        if ((xmaj & lone) == 0) {
          const uint32_t e = pext32(xmaj, pairMaj);
          hit_s[ui] = pext32(pdep32(e, pairMin), sgpr(md.minmask));
          hitv[ui] = se[e];
        }
      }
    };
    const int NB = 1 << (kS - RB);
    const int nbm = kmin - RB;                                 // minor bits above the register bits
    const uint32_t bmask = (1u << nbm) - 1u;
    const uint32_t smlfull = (1u << nml) - 1u;
    const int nsteps = NB + MT;
    auto blk = [&](int level, int step) -> int {              // block a unit of this level finishes in this step, -1: none
      if (level < 0) return -1;
      const int bi = step - (TR ? MT - level : level);
      if (bi < 0 || bi >= NB) return -1;
      return TR ? NB - 1 - bi : bi;
    };
    char* const ymb = reinterpret_cast<char*>(ym);
    uint32_t voff[MUPW];                                       // byte offset of the row inside a column block
#pragma unroll
    for (int ui = 0; ui < MUPW; ++ui) voff[ui] = urk[ui] * (uint32_t)sizeof(VecT);
    // ---- moves along the serial bits above the register bits: the thread's own earlier blocks.  PF of them are
    // requested one unit ahead (pj / pv), any more (spaces with more than PF such bits) are fetched on the spot.
    VecT pv[PF];
    uint32_t prest = 0;
    // rows of a column block are read / written as base (buffer descriptor + scalar block offset) + 32-bit lane offset:
    // no per-lane 64-bit address arithmetic in the step loop
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    struct Raw { u32x4 lo, hi; };
    static_assert(sizeof(Raw) == sizeof(VecT), "a row of a block is two 16-byte accesses");
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(sgpr64((long long)ymb)), 0,
                                                        (int)sgpr((uint32_t)((unsigned long long)half * sizeof(T))), 0x00020000);
    auto ld_row = [&](uint32_t blkidx, uint32_t vo) -> VecT {
      Raw r;
      r.lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo, (int)(blkidx * BLKB), 0);
      r.hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo + 16, (int)(blkidx * BLKB), 0);
      return __builtin_bit_cast(VecT, r);
    };
    auto st_row = [&](uint32_t blkidx, uint32_t vo, const VecT& v) {
      const Raw r = __builtin_bit_cast(Raw, v);
      __builtin_amdgcn_raw_buffer_store_b128(r.lo, rsrc, (int)vo, (int)(blkidx * BLKB), 0);
      __builtin_amdgcn_raw_buffer_store_b128(r.hi, rsrc, (int)vo + 16, (int)(blkidx * BLKB), 0);
    };
    auto nbr_block = [&](int B, int jb) -> uint32_t {         // the block that differs from B in block bit jb (uniform)
#if defined(MMHN_ABL_NEAR)        // timing-only ablations (wrong results): every neighbour block served by L2 / by nobody
      return (uint32_t)(jb & 3);
#elif defined(MMHN_ABL_NOLOAD)
      return 0x7fffffu;
#else
      return (uint32_t)(TR ? B + (1 << jb) : B - (1 << jb));
#endif
    };
    // Prefetch slot q holds the neighbour along block bit q, whether the block has that move or not: a move that does
    // not exist is requested beyond the end of the buffer (the hardware returns zeros without touching memory), so
    // the step loop has no per-move control flow - on this kernel the scalar unit, not the vector units, is the
    // scarce resource (profiles/r3: 87 % busy with per-move branches and bit scans).
    auto issue = [&](auto UI, int B) {
      constexpr int ui = decltype(UI)::value;
      const uint32_t nb_ = (TR ? ~(uint32_t)B : (uint32_t)B) & (uint32_t)(NB - 1);
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const uint32_t nbk = (uint32_t)(TR ? B + (1 << q) : B - (1 << q));
        const uint32_t so = ((nb_ >> q) & 1u) ? nbk * BLKB : 0x80000000u;
        Raw r;
        r.lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff[ui], (int)so, 0);
        r.hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff[ui] + 16, (int)so, 0);
        pv[q] = __builtin_bit_cast(VecT, r);
      }
      prest = nb_ >> PF << PF;
    };
    auto slot = [&](auto UI, int step) {
      constexpr int ui = decltype(UI)::value;
      constexpr int un = (ui + 1) % MUPW;
      const int B = blk(ul[ui], step);
      const int Bn = blk(ul[un], ui + 1 < MUPW ? step : step + 1);
      T acc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = T(0);
      const uint32_t Bmin = (uint32_t)B & bmask, sml = B >= 0 ? (uint32_t)B >> nbm : 0u;
      const uint32_t smin0 = Bmin << RB;
      if (B >= 0) {
        if (Bmin == (TR ? bmask : 0u)) update(UI, sml);
        {
          auto rate_of = [&](int jb, T (&rr)[NC]) {           // rates of block-bit move jb for the NC columns (on-the-spot moves)
            const int sg = jb + RB;
            if (sg < kmin) {
              const T* rp = Rmin + sg * Vmin + (TR ? smin0 : smin0 - (1u << sg));
#pragma unroll
              for (int c = 0; c < NC; ++c) rr[c] = rp[c];
            } else {
              const T rl = RLl[(sg - kmin) * MROWS + urk[ui]];
#pragma unroll
              for (int c = 0; c < NC; ++c) rr[c] = rl;
            }
          };
          auto take = [&](int jb, const VecT& nv) {
            T rr[NC];
            rate_of(jb, rr);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fma_m(rr[c], nv[c], acc[c]);
          };
          // the PF prefetched slots: rate = minor-class rate of bit q (a row of zeros when bit q is no minor-class bit)
          // + left-over-bit rate of the row (a row of zeros when it is none); the value is zero when the move does not exist
          constexpr int G = 4;
#pragma unroll
          for (int q0 = 0; q0 < PF; q0 += G) {
            T rr[G * NC], rl[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const int q = q0 + g;
              if (q < PF) {
                const uint32_t src = TR ? smin0 : (smin0 & ~(1u << (q + RB)));
                const T* rp = Rmin + (((uint32_t)(q + RB)) << kmin) + src;
                const int i = q - nbm;
                const int row = (i >= 0 && i < nml) ? i : NML;
#pragma unroll
                for (int c = 0; c < NC; ++c) rr[g * NC + c] = rp[c];
                rl[g] = RLl[row * MROWS + urk[ui]];
              } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) rr[g * NC + c] = T(0);
                rl[g] = T(0);
              }
            }
            pin_all(rr);
            pin_all(rl);
#pragma unroll
            for (int g = 0; g < G; ++g)
              if (q0 + g < PF) {
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = fma_m(rr[g * NC + c] + rl[g], pv[q0 + g][c], acc[c]);
              }
          }
          uint32_t rest = prest;
          while (rest) {
            const int jb = __ffs(rest) - 1;
            rest &= rest - 1;
            const VecT nv = ld_row(nbr_block(B, jb), voff[ui]);
            take(jb, nv);
          }
        }
      }
      STAMP(0);
#ifndef MMHN_ABL_NOTAKE
      if (Bn >= 0) issue(IC<un>{}, Bn);
#endif
      STAMP(1);
      if (B >= 0) {
        const int l = ul[ui];
        // ---- moves along the thread bits: finished rows of the neighbouring level, in LDS.  One straight-line case per
        // number of moves (the level is fixed per unit): every LDS read of the block can be in flight at once
        {
          const T* rs = ring + ((uint32_t)B & 1u) * (NC * MROWS);
          const int nT = TR ? MT - l : l;
          auto tterms = [&](auto NT) {
            constexpr int nt = decltype(NT)::value;
            constexpr int G = 4;                                // terms whose reads are in flight together
#pragma unroll
            for (int j0 = 0; j0 < nt; j0 += G) {
              T v[G * NC];
#pragma unroll
              for (int g = 0; g < G; ++g) {
                if (j0 + g < nt) {
                  const int j = j0 + g;
                  const uint32_t a = (j & 1) ? (pA[ui][j >> 1] >> 16) : (pA[ui][j >> 1] & 0xffffu);
#pragma unroll
                  for (int c = 0; c < NC; ++c) v[g * NC + c] = rs[c * MROWS + a];
                } else {
#pragma unroll
                  for (int c = 0; c < NC; ++c) v[g * NC + c] = T(0);
                }
              }
              pin_all(v);
#pragma unroll
              for (int g = 0; g < G; ++g)
                if (j0 + g < nt) {
#pragma unroll
                  for (int c = 0; c < NC; ++c) acc[c] = fma_m(RT[ui][j0 + g], v[g * NC + c], acc[c]);
                }
            }
          };
#ifndef MMHN_ABL_NOT
          switch (nT) {
            case 1: tterms(IC<1>{}); break;
            case 2: tterms(IC<2>{}); break;
            case 3: tterms(IC<3>{}); break;
            case 4: tterms(IC<4>{}); break;
            case 5: tterms(IC<5>{}); break;
            case 6: tterms(IC<6>{}); break;
            case 7: tterms(IC<7>{}); break;
            case 8: tterms(IC<8>{}); break;
            case 9: tterms(IC<9>{}); break;
            case 10: tterms(IC<(MT >= 10 ? 10 : MT)>{}); break;
            default: break;
          }
#endif
        }
        STAMP(2);
        // ---- right-hand side
        if (!TR) {
          if (__builtin_amdgcn_ballot_w64((hit_s[ui] >> RB) == Bmin) != 0ull) {        // (rare: skipped by a scalar branch)
            const T hv = (hit_s[ui] >> RB) == Bmin ? hitv[ui] : T(0);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] += ((hit_s[ui] & (uint32_t)(NC - 1)) == (uint32_t)c) ? hv : T(0);
          }
        } else {
          const JLink<T>& Lk = links[prob];
          if (l == MT && sml == smlfull && Lk.soff[rowpart] >= 0) {
            const T* qr = qS + Lk.soff[rowpart] + (1ll << (Lk.sk[rowpart] - 1)) + smin0;
            const T cr = Lk.cst[rowpart];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] += cr * qr[c];
          }
          if (Bmin == bmask && Lk.soff[colpart] >= 0)
            acc[NC - 1] += Lk.cst[colpart] * qS[Lk.soff[colpart] + (1ll << (Lk.sk[colpart] - 1)) + (ut[ui] | (sml << MT))];
        }
        // ---- the block itself: register-bit moves, diagonal (every LDS operand requested up front)
        T dmv[NC], rin[RB][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) dmv[c] = dminL[smin0 + c];
        pin_all(dmv);
        T inv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) inv[c] = fast_rcp(dmaj[ui] + dmv[c]);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const bool used = TR ? !((c >> r) & 1) : ((c >> r) & 1);
            rin[r][c] = used ? Rmin[r * Vmin + smin0 + (uint32_t)(TR ? c : (c ^ (1 << r)))] : T(0);
          }
        }
        VecT Y;
        if (!TR) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            T z = acc[c];
#pragma unroll
            for (int r = 0; r < RB; ++r)
              if ((c >> r) & 1) z = fma_m(rin[r][c], Y[c ^ (1 << r)], z);
            Y[c] = z * inv[c];
          }
        } else {
#pragma unroll
          for (int c = NC - 1; c >= 0; --c) {
            T z = acc[c];
#pragma unroll
            for (int r = 0; r < RB; ++r)
              if (!((c >> r) & 1)) z = fma_m(rin[r][c], Y[c | (1 << r)], z);
            Y[c] = z * inv[c];
          }
        }
        STAMP(3);
        T* ws = ring + ((uint32_t)B & 1u) * (NC * MROWS) + urk[ui];
#pragma unroll
        for (int c = 0; c < NC; ++c) ws[c * MROWS] = Y[c];
#ifndef MMHN_ABL_NOSTORE
        st_row((uint32_t)B, voff[ui], Y);
#endif
        STAMP(4);
      }
    };
    {
      const int B0 = blk(ul[0], 0);
      if (B0 >= 0) issue(IC<0>{}, B0);
    }
    STAMP(6);
    for (int step = 0; step < nsteps; ++step) {
      slot(IC<0>{}, step);
      if constexpr (MUPW > 1) slot(IC<1 % MUPW>{}, step);
      if constexpr (MUPW > 2) slot(IC<2 % MUPW>{}, step);
#ifndef MMHN_ABL_NOBAR
      lds_barrier();
#endif
      STAMP(5);
    }
    if (TR) {
      __syncthreads();                                         // every store of the seeded half has landed
      solve_eq();
    }
    __syncthreads();
    STAMP(7);
  }
  STAMP_FLUSH(TR ? 8 : 0);
}

// matrix layout -> natural index order (seeded half; the PT == MT states of the seed = 0 half are copied)
template <typename T>
__global__ __launch_bounds__(MROWS) void k_mconvert(const Desc* __restrict__ descs, const MDesc* __restrict__ mds,
                                                    const uint16_t* __restrict__ rowT, const T* __restrict__ ym,
                                                    T* __restrict__ yn) {
  const MDesc& md = mds[blockIdx.x];
  const Desc& d = descs[md.prob];
  const int k = d.k, kmin = md.kmin, kS = md.kS, tid = threadIdx.x;
  const long long half = 1ll << (k - 1);
  const uint32_t xT = pdep32(rowT[tid], md.tmask);
  const T* src = ym + d.off + half;
  T* dst = yn + d.off + half;
  for (uint32_t s = blockIdx.y; s < (1u << kS); s += gridDim.y) {
    const uint32_t xs = pdep32(s & ((1u << kmin) - 1u), md.minmask) | pdep32(s >> kmin, md.mlmask);
    dst[xs | xT] = src[mpos<MCfg<T>::RB>(s, (uint32_t)tid)];
  }
  if (blockIdx.y == 0) {
    const uint32_t VE = 1u << __popc(d.pairP);
    for (uint32_t e = tid; e < VE; e += MROWS) {
      const uint32_t xp = pdep32(e, d.pairP);
      const uint32_t x0 = xp | (xp << 1);
      yn[d.off + x0] = ym[d.off + x0];
    }
  }
}

}  // namespace mmhn
