// Kronecker products on vectors in index order: k_sweep (any tile, fused Jacobi step), k_hx + k_kv (full tiles of
// multi-tile spaces: the launches of mmhn_kronvec_batched / mmhn_jacobi_step_batched).  Reference: kronvec.py:499-539.
#pragma once
#include "common.h"

namespace mmhn {

// ------------------------------------------------------------------------------------
// k_sweep: y = Q_off p  (TR: Q_off^T p), optionally fused Jacobi step
//          y = lidg * (Q_off p + rhs)      (likelihood.py:253-255, vanilla.py:289-290)
// rhs_mode: 0 dense vector, 1 scal[prob] * e_last, 2 e_0.  p and y may alias (in-place
// Jacobi is exact after k+1 sweeps because Q_off is nilpotent and triangular).
// ------------------------------------------------------------------------------------
#ifndef MMHN_KSB
#define MMHN_KSB 512
#endif
constexpr int KSB = MMHN_KSB;                     // threads per workgroup of k_sweep

// tile-uniform classification of a tile of a joint space
//   0: every state has seeding set (only PT / MT events, plus seeding into eq states)
//   1: seed = 0 tile without any PT == MT state: Q_off has no entries here
//   2: anything else (seeding bit inside the tile, or a seed = 0 tile with eq states): generic path
__device__ __forceinline__ int tile_kind(const Desc& d, uint32_t xhi, int t) {
  if (d.mode != JOINT) return 0;
  if (d.seedbit < t) return 2;                      // includes "no seeding slot"
  if (xhi & (1u << d.seedbit)) return 0;
  const uint32_t hmask = ~((1u << t) - 1u);
  if (xhi & d.lone & hmask) return 1;
  const uint32_t pp = d.pairP & hmask & 0x7fffffffu;
  if (((xhi & pp) << 1) != (xhi & (pp << 1))) return 1;
  return 2;
}

template <typename T, bool TR>
__global__ __launch_bounds__(KSB) void k_sweep(const Desc* __restrict__ descs,
                                               const int2* __restrict__ map,
                                               const Params<T>* __restrict__ par, const T* p, T* y,
                                               const T* __restrict__ lidg,
                                               const T* __restrict__ rhs, int rhs_mode,
                                               const T* __restrict__ scal, int maxk,
                                               const T* __restrict__ tab) {
  extern __shared__ __align__(16) unsigned char smem[];
  Desc& d = *reinterpret_cast<Desc*>(smem);
  T* tile = reinterpret_cast<T*>(smem + DESC_PAD);
  T* Ltab = tile + (1 << TB);
  T* Utab = Ltab + maxk * 64;
  const int tid = threadIdx.x;
  const uint32_t blk = xcd_chunked(blockIdx.x, gridDim.x);
  const int prob = map[blk].x;
  const uint32_t H = (uint32_t)map[blk].y;
  load_desc(&d, descs + prob);
  __syncthreads();
  const int k = d.k;
  const int t = k < TB ? k : TB;
  const uint32_t nelem = 1u << t, tmask = nelem - 1;
  const long long base = d.off;
  const int R = t > 6 ? 1 << (t - 6) : 1;
  const uint32_t xhi = H << t;
  constexpr int NW = KSB / 64;
  constexpr int NJ = 64 / NW;                  // rows per wave
  const int wave = tid >> 6, lane = tid & 63;
  const bool joint = d.mode == JOINT;
  const uint32_t last = (k >= 32) ? 0xffffffffu : ((1u << k) - 1u);
  const int kind = tile_kind(d, xhi, t);

  // own states straight into registers (and into LDS for the row-bit neighbours); rows of Q_off that are
  // identically zero (kind 1) need neither p nor the rate tables
  T v[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const uint32_t xl = ((uint32_t)(wave + NW * j) << 6) | (uint32_t)lane;
    v[j] = (kind != 1 && xl < nelem) ? p[base + xhi + xl] : T(0);
  }
  if (kind != 1) {                             // tile-uniform branch
    tile_tables(d, tab, H, Ltab, Utab, tile);  // uses the tile area as scratch, ends with a barrier
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint32_t xl = ((uint32_t)(wave + NW * j) << 6) | (uint32_t)lane;
      if (xl < nelem) tile[xl] = v[j];
    }
    __syncthreads();
  }

  T acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = 0;

  if (kind == 0) {
    // ---- fast path: every bit is a plain single-bit move
    const int nlane = t < 6 ? t : 6;
    // lane bits: neighbour = other lane of the same row, read from the staged tile (one conflict-free
    // ds_read_b64 per state instead of two ds_bpermute)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      if (b < nlane) {
        const T Lb = Ltab[b * 64 + lane];
        const bool has = (lane >> b) & 1;
        const bool on = TR ? !has : has;
        const uint32_t nl = (uint32_t)lane ^ (1u << b);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int r = wave + NW * j;
          if (r < R) {
            const T nb = tile[((uint32_t)r << 6) | nl];
            acc[j] += on ? Lb * Utab[b * 64 + r] * nb : T(0);
          }
        }
      }
    }
    // row bits: neighbour = same lane of another row of the tile (LDS, conflict-free)
    for (int b = 6; b < t; ++b) {
      const T Lb = Ltab[b * 64 + lane];
      const int rb = 1 << (b - 6);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = wave + NW * j;
        const bool has = (r & rb) != 0;
        if (r < R && (TR ? !has : has)) acc[j] += Lb * Utab[b * 64 + r] * tile[((r ^ rb) << 6) | lane];
      }
    }
    // tile bits: neighbour = same position of another tile (coalesced global rows); the moves that apply to
    // this tile are collected in a scalar bit set and taken two at a time (16 rows in flight per thread)
    uint32_t mvs = 0;
    for (int b = t; b < k; ++b) {
      const uint32_t bit = 1u << b;
      const bool has = (xhi & bit) != 0;
      const bool is_seed = joint && b == d.seedbit;
      if (is_seed ? TR : (TR ? has : !has)) continue;     // seeding enters these tiles only in Q (not Q^T)
      if (is_seed && !seed_move_possible(d.lone, d.pairP, xhi, tmask)) continue;
      mvs |= bit;
    }
    auto take = [&](int b, const T (&nv)[NJ]) {
      const bool is_seed = joint && b == d.seedbit;
      const T Lb = Ltab[b * 64 + lane];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = wave + NW * j;
        const uint32_t xl = ((uint32_t)r << 6) | (uint32_t)lane;
        const bool on = r < R && xl < nelem && (!is_seed || eq_noseed(d, xhi | xl));
        acc[j] += on ? Lb * Utab[b * 64 + (r & 63)] * nv[j] : T(0);
      }
    };
    while (mvs) {
      const int b0 = __ffs(mvs) - 1;
      mvs &= mvs - 1;
      const int b1 = mvs ? __ffs(mvs) - 1 : -1;
      if (b1 >= 0) mvs &= mvs - 1;
      T n0[NJ], n1[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const uint32_t xl = ((uint32_t)(wave + NW * j) << 6) | (uint32_t)lane;
        n0[j] = xl < nelem ? p[base + ((xhi | xl) ^ (1u << b0))] : T(0);
      }
      if (b1 >= 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const uint32_t xl = ((uint32_t)(wave + NW * j) << 6) | (uint32_t)lane;
          n1[j] = xl < nelem ? p[base + ((xhi | xl) ^ (1u << b1))] : T(0);
        }
      }
      take(b0, n0);
      if (b1 >= 0) take(b1, n1);
    }
  } else if (kind == 2) {
    // ---- generic path (seeding bit inside the tile, or seed = 0 tile with PT == MT states)
#pragma unroll 1
    for (int j = 0; j < NJ; ++j) {
      const int r = wave + NW * j;
      const uint32_t xl = ((uint32_t)r << 6) | (uint32_t)lane;
      if (r >= R || xl >= nelem) continue;
      const uint32_t x = xhi | xl;
      const bool ss = seed_set(d, x);
      const bool e0x = eq_noseed(d, x);
      T a = 0;
      for (int b = 0; b < k; ++b) {
        const uint32_t bit = 1u << b;
        const bool has = (x >> b) & 1u;
        const int c = d.cls[b];
        uint32_t nb = x ^ bit;
        bool cond;
        if (joint && c == CS) {
          cond = (TR ? !has : has) && e0x;                     // seeding event (kronvec.py:434-496)
        } else if (ss) {
          cond = TR ? !has : has;                              // PT / MT event after seeding (:290-431)
        } else if ((d.pairP >> b) & 1u) {
          const uint32_t both = 3u << b;                       // synchronised event before seeding (:214-287)
          nb = x ^ both;
          cond = e0x && (TR ? (x & both) == 0 : (x & both) == both);
        } else {
          cond = false;
        }
        if (cond) {
          const T nv = ((nb >> t) == H) ? tile[nb & tmask] : p[base + nb];
          a += Ltab[b * 64 + lane] * Utab[b * 64 + r] * nv;
        }
      }
      acc[j] = a;
    }
  }
  // kind == 1: Q_off has no entries in this tile, acc stays 0

  if (!lidg && t == TB) {
    // plain product on a full tile: y is not read again by this launch, so it leaves through LDS as 16-byte
    // write-through stores that do not stay in the XCD's L2 (`sc0 sc1`; 8-byte ones would cost 2.7x per byte) -
    // the L2 then keeps the p tiles that later tiles read as neighbours
    if (kind != 1) {
      __syncthreads();                           // every neighbour read of the staged p tile is done
#pragma unroll
      for (int j = 0; j < NJ; ++j) tile[((uint32_t)(wave + NW * j) << 6) | (uint32_t)lane] = acc[j];
      __syncthreads();
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int PER = 16 / sizeof(T);          // elements per 16-byte store
    for (uint32_t e = (uint32_t)tid * PER; e < nelem; e += KSB * PER) {
      f32x4 val = kind != 1 ? *reinterpret_cast<const f32x4*>(&tile[e]) : f32x4{0.f, 0.f, 0.f, 0.f};
      T* dst = y + base + xhi + e;
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(val) : "memory");
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int r = wave + NW * j;
    const uint32_t xl = ((uint32_t)r << 6) | (uint32_t)lane;
    if (r >= R || xl >= nelem) continue;
    const uint32_t x = xhi | xl;
    T out = acc[j];
    if (lidg) {
      T rv;
      if (rhs_mode == 0) rv = rhs[base + x];
      else if (rhs_mode == 1) rv = (x == last) ? scal[prob] : T(0);
      else rv = (x == 0) ? e0_scale<T>() : T(0);
      out = lidg[base + x] * (acc[j] + rv);
    }
    y[base + x] = out;
  }
}

// ------------------------------------------------------------------------------------
// k_kv: y = Q_off p (TR: Q_off^T p) on full tiles of multi-tile spaces - the kronvec metric (kronvec.py:499-539 with
// diag = False).  Same arithmetic as k_sweep; what changed is the workgroup's schedule: the descriptor is read
// through uniform (scalar) loads instead of an LDS copy behind a barrier, the tile-bit factors hx[b] of every tile come
// from a table (k_hx, once per parameter set) and are folded into the per-lane factor instead of a rebuilt Utab behind
// two more barriers, the tile and the tables share ONE barrier, and the first neighbour tiles are in flight while the
// lane- and row-bit terms run from LDS.  Three barriers per tile instead of seven.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int sgpr(int v);
__device__ __forceinline__ uint32_t sgpr(uint32_t v);
__device__ __forceinline__ long long sgpr64(long long v);
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T>
__global__ __launch_bounds__(64) void k_hx(const Desc* __restrict__ descs, const int2* __restrict__ map,
                                           const T* __restrict__ tab, T* __restrict__ hxt, int maxk) {
  const Desc& d = descs[map[blockIdx.x].x];
  const uint32_t H = (uint32_t)map[blockIdx.x].y;
  const int k = d.k, t = k < TB ? k : TB, b = threadIdx.x;
  if (b >= k) return;
  const T* thc = tab + d.toff;
  T h = thc[b * k + b];
  for (int bb = t; bb < k; ++bb) if (bb != b && ((H >> (bb - t)) & 1u)) h *= thc[b * k + bb];
  hxt[(long long)blockIdx.x * maxk + b] = h;
}

#ifndef MMHN_KV_PRE
#define MMHN_KV_PRE 2          // neighbour tiles k_kv keeps in flight (3 / 4: 102 / 118 VGPRs, one wave per SIMD less, slower)
#endif
#ifndef MMHN_KV_LU
#define MMHN_KV_LU 2            // lane-bit moves unrolled (3: 98 VGPRs, one wave per SIMD less)
#endif
#ifndef MMHN_KV_DIRECT
#define MMHN_KV_DIRECT 1         // y leaves as 8-byte write-through stores straight from the accumulators (0: through LDS as 16-byte stores)
#endif
#ifndef MMHN_KV_WPS
#define MMHN_KV_WPS 4          // waves per SIMD k_kv's registers are sized for (4: two 512-thread workgroups per CU, 128 VGPRs)
#endif
// zmap (optional): zmap[i] = a tile of the same vector in which Q_off has no entries (a seed = 0 tile without PT == MT
//   states, -1: none) that the workgroup of list entry i clears on its way - the product then fills ALL of y with a
//   launch over the live tiles only (no memset, no workgroups that do nothing but store zeros).
// JAC: fused Jacobi step y = lidg * (Q_off p + rhs)  (likelihood.py:253-255); a tile without entries gets lidg * rhs
//   (from its live counterpart's workgroup when zmap is given, else from its own).
template <typename T, bool TR, int TPW, bool JAC>
__global__ __launch_bounds__(KSB, MMHN_KV_WPS) void k_kv(const Desc* __restrict__ descs, const int2* __restrict__ map, int ntiles,
                                                         const T* __restrict__ p, T* __restrict__ y,
                                                         const T* __restrict__ tab, const T* __restrict__ hxt, int maxk,
                                                         const int* __restrict__ zmap, const T* __restrict__ lidg,
                                                         const T* __restrict__ rhs) {
  extern __shared__ __align__(16) unsigned char smem[];
  Desc& dsh = *reinterpret_cast<Desc*>(smem);              // only staged for the generic path
  T* tile = reinterpret_cast<T*>(smem + DESC_PAD);
  T* Ltab = tile + (1 << TB);
  T* Urow = Ltab + maxk * 64;
  T* hx = Urow + maxk * 64;
  const int tid = threadIdx.x;
  constexpr int t = TB;
  constexpr uint32_t nelem = 1u << TB, tmask = nelem - 1;
  constexpr int NW = KSB / 64, NJ = 64 / NW;
  const int wave = tid >> 6, lane = tid & 63;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int PER = 16 / sizeof(T);
  // a workgroup walks TPW consecutive tiles of the list (XCD-chunked): the tables of a problem are staged once per
  // run, and the next tile's own states are fetched while the current tile computes.  (Measured: the walk costs more
  // scalar and vector registers than it hides latency - TPW = 2 spills at 128 VGPRs - so TPW = 1 is what ships.)
  constexpr int tpw = TPW;
  const uint32_t first = xcd_chunked(blockIdx.x, gridDim.x) * (uint32_t)tpw;
  int cur_prob = -1;
  T vnext[NJ];
  {
    const int prob0 = sgpr(map[first].x);
    const uint32_t H0 = sgpr((uint32_t)map[first].y);
    const long long base0 = sgpr64(descs[prob0].off);
#pragma unroll
    for (int j = 0; j < NJ; ++j) vnext[j] = (p + base0 + (H0 << t))[(((uint32_t)(wave * NJ + j) << 6) | (uint32_t)lane)];
  }
#pragma unroll
  for (int it = 0; it < tpw; ++it) {
    const uint32_t blk = first + (uint32_t)it;
    if (blk >= (uint32_t)ntiles) break;
    const int prob = sgpr(map[blk].x);
    const uint32_t H = sgpr((uint32_t)map[blk].y);
    const Desc& dg = descs[prob];
    const int k = sgpr(dg.k);
    const long long base = sgpr64(dg.off), toff = sgpr64(dg.toff);
    const uint32_t xhi = H << t;
    const int seedb = sgpr(dg.seedbit);
    const uint32_t lone = sgpr(dg.lone), pairP = sgpr(dg.pairP);
    const bool joint = sgpr(dg.mode) == JOINT;
    int kind = 0;                                             // tile_kind on scalars
    if (joint) {
      if (seedb < t) kind = 2;
      else if (xhi & (1u << seedb)) kind = 0;
      else {
        const uint32_t hmask = ~tmask;
        const uint32_t pp = pairP & hmask & 0x7fffffffu;
        kind = ((xhi & lone & hmask) || (((xhi & pp) << 1) != (xhi & (pp << 1)))) ? 1 : 2;
      }
    }
    T v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = vnext[j];
    if (it + 1 < tpw && blk + 1 < (uint32_t)ntiles) {          // next tile's own states: in flight during this tile
      const int probn = sgpr(map[blk + 1].x);
      const uint32_t Hn = sgpr((uint32_t)map[blk + 1].y);
      const long long basen = sgpr64(descs[probn].off);
#pragma unroll
      for (int j = 0; j < NJ; ++j) vnext[j] = (p + basen + (Hn << t))[(((uint32_t)(wave * NJ + j) << 6) | (uint32_t)lane)];
    }
    if (zmap) {                                                // the tile without entries of Q_off this workgroup fills
      const int zt = sgpr(zmap[blk]);
      if (zt >= 0) {
        for (uint32_t e = (uint32_t)tid * PER; e < nelem; e += KSB * PER) {
          const long long xi = base + ((uint32_t)zt << t) + e;
          if (JAC) {                                           // lidg * rhs: the row of Q_off is empty there
#pragma unroll
            for (int u = 0; u < PER; ++u) y[xi + u] = lidg[xi + u] * rhs[xi + u];
          } else {
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(y + xi), "v"(zero4) : "memory");
          }
        }
      }
    }
    if (kind == 1) {                                           // Q_off has no entries in this tile
      for (uint32_t e = (uint32_t)tid * PER; e < nelem; e += KSB * PER) {
        T* dst = y + base + xhi + e;
        if (JAC) {
#pragma unroll
          for (int u = 0; u < PER; ++u) dst[u] = lidg[base + xhi + e + u] * rhs[base + xhi + e + u];
        } else {
          const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(zero4) : "memory");
        }
      }
      continue;
    }
    // tables (once per problem), tile-bit factors and the tile: one barrier
    if (prob != cur_prob) {
      const T* src = tab + toff + k * k;
      for (int e = tid; e < k * 64; e += KSB) Ltab[e] = src[e];
      if (kind == 2) for (int e = tid; e < k * 64; e += KSB) Urow[e] = src[k * 64 + e];    // (kind 0 reads U through the scalar unit)
      cur_prob = prob;
    }
    if (tid < k) hx[tid] = hxt[(long long)blk * maxk + tid];
    if (kind == 2) {
      const int* sw = reinterpret_cast<const int*>(&dg);
      int* dw = reinterpret_cast<int*>(&dsh);
      for (int i = tid; i < DESC_WORDS; i += KSB) dw[i] = sw[i];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) tile[((uint32_t)(wave * NJ + j) << 6) | (uint32_t)lane] = v[j];
    __syncthreads();

    T acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0;
    T jl[JAC ? NJ : 1], jr[JAC ? NJ : 1];                    // fused Jacobi step: this thread's 1/diag and rhs, in flight during the terms
    if (JAC) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const long long xi = base + xhi + ((((uint32_t)(wave * NJ + j)) << 6) | (uint32_t)lane);
        jl[j] = lidg[xi];
        jr[j] = rhs[xi];
      }
    }
    if (kind == 0) {
      // Every term is  acc[j] += L_b[lane] * hx[b] * U_b[row] * neighbour.  The wave's NJ rows are consecutive and
      // wave-uniform, so U_b[row] comes through the scalar unit from the table in global memory (one 64-byte scalar
      // load per move instead of one LDS broadcast read per term), "is this move open" is a per-lane factor (lane
      // bits), a compile-time pattern (row bits inside the wave's rows) or one scalar branch (higher row bits, tile
      // bits): all loads of a move are issued before its first use and every term is one multiply and one FMA.
      static_assert(NJ == 8, "k_kv: 8 consecutive rows per wave");
      const int ws = sgpr(wave);
      const T* __restrict__ Ug = tab + toff + k * k + k * 64 + ws * NJ;
      auto urow = [&](int b, T (&u)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) u[j] = Ug[b * 64 + j];
      };
      // tile bits first: their neighbour rows come from other tiles (L2 / HBM) and fly while the LDS terms run.
      // (Requesting them before the tile is staged, through the barrier, was measured slower: 102 VGPRs, 0.205 vs 0.197 ms.)
      uint32_t mvs = 0;
      for (int b = t; b < k; ++b) {
        const uint32_t bit = 1u << b;
        const bool has = (xhi & bit) != 0;
        const bool is_seed = joint && b == seedb;
        if (is_seed ? TR : (TR ? has : !has)) continue;       // seeding enters these tiles only in Q (not Q^T)
        if (is_seed && !seed_move_possible(lone, pairP, xhi, tmask)) continue;
        mvs |= bit;
      }
      auto fetch = [&](int b, T (&nv)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) nv[j] = (p + base + (xhi ^ (1u << b)))[(((uint32_t)(ws * NJ + j) << 6) | (uint32_t)lane)];
      };
      auto take = [&](int b, const T (&nv)[NJ]) {
        const bool is_seed = joint && b == seedb;
        const T Lb = Ltab[b * 64 + lane] * hx[b];
        T u[NJ];
        urow(b, u);
        if (!is_seed) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[j] = fma_t(Lb * u[j], nv[j], acc[j]);
        } else {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const uint32_t x = xhi | ((uint32_t)(ws * NJ + j) << 6) | (uint32_t)lane;
            const bool on = (x & lone) == 0 && (((x & pairP) << 1) == (x & (pairP << 1)));
            acc[j] += on ? Lb * u[j] * nv[j] : T(0);        // (a select: values of unwritten states may be anything)
          }
        }
      };
      // MMHN_KV_PRE neighbour tiles in flight: requested, the LDS terms run, taken, the next ones requested
      constexpr int PRE = MMHN_KV_PRE;
      T nq[PRE][NJ];
      int bq[PRE];
#pragma unroll
      for (int q = 0; q < PRE; ++q) {
        bq[q] = mvs ? __ffs(mvs) - 1 : -1;                      // (scalar)
        mvs &= mvs - 1;                                         // 0 stays 0
        if (bq[q] >= 0) fetch(bq[q], nq[q]);
      }
      // lane bits: neighbour = other lane of the same row (conflict-free ds_read_b64 from the staged tile)
#pragma unroll MMHN_KV_LU
      for (int b = 0; b < 6; ++b) {
        const bool has = (lane >> b) & 1;
        const T Lb = (TR ? !has : has) ? Ltab[b * 64 + lane] * hx[b] : T(0);
        const T* nrow = tile + ((uint32_t)(ws * NJ) << 6) + ((uint32_t)lane ^ (1u << b));
        T u[NJ], nb[NJ];
        urow(b, u);
#pragma unroll
        for (int j = 0; j < NJ; ++j) nb[j] = nrow[j << 6];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = fma_t(Lb * u[j], nb[j], acc[j]);
      }
      // row bits 6-8: the partner row is one of the wave's own rows, the pattern is known at compile time
#pragma unroll
      for (int b = 6; b < 9; ++b) {
        const T Lb = Ltab[b * 64 + lane] * hx[b];
        const int rb = 1 << (b - 6);
        const T* rows = tile + ((uint32_t)(ws * NJ) << 6) + (uint32_t)lane;
        T u[NJ];
        urow(b, u);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const bool has = (j & rb) != 0;
          if (TR ? !has : has) acc[j] = fma_t(Lb * u[j], rows[(j ^ rb) << 6], acc[j]);
        }
      }
      // row bits 9-11: open or closed for the whole wave
#pragma unroll 1
      for (int b = 9; b < t; ++b) {
        const int wb = 1 << (b - 9);
        const bool has = (ws & wb) != 0;
        if (TR ? has : !has) continue;
        const T Lb = Ltab[b * 64 + lane] * hx[b];
        const T* rows = tile + ((uint32_t)((ws ^ wb) * NJ) << 6) + (uint32_t)lane;
        T u[NJ], nb[NJ];
        urow(b, u);
#pragma unroll
        for (int j = 0; j < NJ; ++j) nb[j] = rows[j << 6];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = fma_t(Lb * u[j], nb[j], acc[j]);
      }
      for (;;) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) if (bq[q] >= 0) take(bq[q], nq[q]);
        if (!mvs) break;
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
          bq[q] = mvs ? __ffs(mvs) - 1 : -1;
          mvs &= mvs - 1;
          if (bq[q] >= 0) fetch(bq[q], nq[q]);
        }
      }
    } else {
      // generic path (seeding bit inside the tile, or a seed = 0 tile with PT == MT states): per-state conditions
      const Desc& d = dsh;
#pragma unroll 1
      for (int j = 0; j < NJ; ++j) {
        const int r = wave * NJ + j;
        const uint32_t xl = ((uint32_t)r << 6) | (uint32_t)lane;
        const uint32_t x = xhi | xl;
        const bool ss = seed_set(d, x);
        const bool e0x = eq_noseed(d, x);
        T a = 0;
        // without the seeding and without PT == MT a state has no entries in its row / column at all: true for all but
        // a handful of states of a seed = 0 tile (2^pairs of 4 096), whose bit loop would otherwise be the launch's tail
        for (int b = 0; b < ((ss || e0x) ? k : 0); ++b) {
          const uint32_t bit = 1u << b;
          const bool has = (x >> b) & 1u;
          const int c = d.cls[b];
          uint32_t nb = x ^ bit;
          bool cond;
          if (joint && c == CS) cond = (TR ? !has : has) && e0x;
          else if (ss) cond = TR ? !has : has;
          else if ((d.pairP >> b) & 1u) {
            const uint32_t both = 3u << b;
            nb = x ^ both;
            cond = e0x && (TR ? (x & both) == 0 : (x & both) == both);
          } else cond = false;
          if (cond) {
            const T nv = ((nb >> t) == H) ? tile[nb & tmask] : p[base + nb];
            a += Ltab[b * 64 + lane] * hx[b] * Urow[b * 64 + r] * nv;
          }
        }
        acc[j] = a;
      }
    }
    // y is not read again by this launch: it leaves through LDS as 16-byte write-through stores that do not stay in
    // the XCD's L2, which keeps the p tiles that later tiles read as neighbours
    if (JAC) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = jl[j] * (acc[j] + jr[j]);
    }
#if MMHN_KV_DIRECT
    // (variant: 8-byte write-through stores straight from the accumulators, no trip through LDS)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      T* dst = y + base + xhi + ((((uint32_t)(wave * NJ + j)) << 6) | (uint32_t)lane);
      if (sizeof(T) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(acc[j]) : "memory");
      else *dst = acc[j];
    }
#else
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) tile[((uint32_t)(wave * NJ + j) << 6) | (uint32_t)lane] = acc[j];
    __syncthreads();
    for (uint32_t e = (uint32_t)tid * PER; e < nelem; e += KSB * PER) {
      const f32x4 val = *reinterpret_cast<const f32x4*>(&tile[e]);
      T* dst = y + base + xhi + e;
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(val) : "memory");
    }
#endif
    if (it + 1 < tpw) __syncthreads();                       // the tile (and hx) are rewritten by the next trip
  }
}

}  // namespace mmhn
