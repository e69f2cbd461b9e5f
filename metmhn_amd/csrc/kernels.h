// gfx950 kernels of the metMHN hot path (closed-form "gather" formulation).
//
// Work decomposition shared by the streaming kernels: a problem's 2^k state vector is cut
// into tiles of 2^t contiguous states (t = min(k, TB)); one 256-thread workgroup owns one
// tile, `map[blockIdx.x] = {problem, tile}`.  Inside a tile lane l of a wave owns the states
// whose low 6 index bits are l, waves walk the 64-state rows.  Every transition rate is
//     rate_b(x) = Ltab[b][lane] * Utab[b][row]
// (a per-lane constant times a wave-uniform factor): the product over the bits of x that act
// on event ev(b) splits into lane bits (0..5), row bits (6..t-1) and tile bits (t..k-1); both
// tables live in LDS and are rebuilt per tile from the active theta row.  Neighbour states
// x ^ bit are read from the LDS copy of the tile when the bit is below t and as coalesced
// global loads otherwise.  HBM-bound elementwise / permute work: no MFMA.
//
// Reference semantics: metmhn/jx/kronvec.py (kronvec :499-539, kron_diag :964-999,
// diag_scal_* :574-671, obs_states :1056-1095), likelihood.py (R_i_inv_vec :231-262,
// x_partial_Q_y :163-201, x_partial_D_y :204-228), vanilla.py (single-tumour versions).
#pragma once
#include <hip/hip_runtime.h>
#include "desc.h"
#include "wlayout.h"

namespace mmhn {

#ifndef MMHN_TB
#define MMHN_TB 12
#endif
constexpr int TB = MMHN_TB;   // tile bits
constexpr int BLOCK = 256;    // threads per workgroup
constexpr int WAVES = BLOCK / 64;
constexpr int DESC_WORDS = (sizeof(Desc) + 3) / 4;
constexpr int DESC_PAD = ((sizeof(Desc) + 15) / 16) * 16;

// XCD-aware block -> work-item mapping: workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8 shares an XCD, MI355X_MICROARCH.md), so give every XCD one contiguous chunk of the
// list; consecutive tiles of one patient then share an L2.  Speed only, never correctness.
__device__ __forceinline__ uint32_t xcd_chunked(uint32_t b, uint32_t n) {
#ifdef MMHN_NO_XCD_REMAP
  return b;
#else
  const uint32_t q = n >> 3, rem = n & 7u, x = b & 7u, i = b >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + i;
#endif
}

// Forward solves start from E0 * e_0 instead of e_0: a power of two that keeps the deep states of large
// spaces away from the fp32 underflow range (SURVEY.md 7 "fp32 at k=25"); exact, undone in the log-prob
// (k_seeds) and invisible to the gradient, whose adjoint is seeded with 1 / score.  fp64 needs none.
template <typename T> __host__ __device__ inline T e0_scale() { return T(1); }
template <> __host__ __device__ inline float e0_scale<float>() { return 1.152921504606846976e18f; }   // 2^60

__device__ __forceinline__ void load_desc(Desc* dst, const Desc* src) {
  const int* s = reinterpret_cast<const int*>(src);
  int* d = reinterpret_cast<int*>(dst);
  for (int i = threadIdx.x; i < DESC_WORDS; i += blockDim.x) d[i] = s[i];
}

// eq(x) without the seeding bit: PT(x) == MT(x) on paired events, lone bits clear
__device__ __forceinline__ bool eq_noseed(const Desc& d, uint32_t x) {
  return ((x & d.lone) == 0) && (((x & d.pairP) << 1) == (x & (d.pairP << 1)));
}
__device__ __forceinline__ bool seed_set(const Desc& d, uint32_t x) {
  return d.mode == SINGLE || (d.seedbit >= 0 && ((x >> d.seedbit) & 1u));
}

// Can any state of the tile with high part xhi (seeding set) be reached by the seeding event, i.e. is x ^ seed a
// PT == MT state for some in-tile part?  Decided on the tile bits alone (a pair straddling the tile boundary
// stays undecided): most seeded tiles of a large space fail it and skip the seed = 0 neighbour tile.
__device__ __forceinline__ bool seed_move_possible(uint32_t lone, uint32_t pairP, uint32_t xhi, uint32_t tmask) {
  if (xhi & lone & ~tmask) return false;
  const uint32_t pp = pairP & ~tmask;
  return ((xhi & pp) << 1) == (xhi & (pp << 1));
}

// ------------------------------------------------------------------------------------
// k_prep: per-problem tables, rebuilt once per evaluation (theta changes, the bit roles do not).
// Layout at tab + d.toff (T elements):
//   THc  [k][k]   THc[b][b'] = theta[ev b][ev b'] if bit b' acts on the event of bit b (same class;
//                 the seeding bit listens to class P), else 1;  THc[b][b] = base rate of bit b
//   Ltab [k][64]  product of THc[b][.] over the lane bits (0..5) set in l, b itself excluded
//   Urow [k][64]  the same over the row bits (6..t-1)
//   JOINT with seeding only - the diagonal of (D_p + D_m - Q) in Kronecker-sum form:
//   dP [2^kP]  D_p(S) + total rate of the PT events that can still fire from PT-set S   (seed = 1 half)
//   dM [2^kM]  the same for the metastasis;        diag(x) = dP[x_P] + dM[x_M]
//   dE [2^kE]  diagonal on the seed = 0 states with PT == MT (index: subset of paired events)
// so a tile gets 1/(D - diag Q) from two small table reads instead of a 2^k vector.
// ------------------------------------------------------------------------------------
__host__ __device__ inline long long rate_table_size(int k) { return (long long)k * k + 2ll * k * 64; }
__host__ __device__ inline long long table_size(const Desc& d) {
  long long s = rate_table_size(d.k);
  if (d.mode == JOINT && d.seedbit >= 0) s += (1ll << popc32(d.maskP)) + (1ll << popc32(d.maskM)) + (1ll << popc32(d.pairP));
  return s;
}

template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(BLOCK) void k_prep(const Desc* __restrict__ descs,
                                                const Params<T>* __restrict__ par, T* tab, const int* __restrict__ plist = nullptr) {
  __shared__ T thc[(MAXN + 1) * MAXN];  // later reused as th[i][class bit l]
  __shared__ T rsplit[(MAXN + 1) * 192];  // [i][three 6-bit parts of S] partial rate products (row N: observation)
  __shared__ Desc d;
  load_desc(&d, descs + (plist ? plist[blockIdx.x] : (int)blockIdx.x));   // (plist: only these problems of the list)
  __syncthreads();
  const int k = d.k, tid = threadIdx.x;
  const int t = k < TB ? k : TB;
  const Params<T>& P = par[d.pset];
  T* out = tab + d.toff;
  // SPLIT (gridDim.y == 4): the three class tables and the rate tables of a problem are independent - a workgroup
  // each, so the kernel at the head of every evaluation is one table long (small cohorts); else all in this one
  const int job = SPLIT ? (int)(blockIdx.y & 3u) : -1;
  const int part = SPLIT ? (int)(blockIdx.y >> 2) : 0, nparts = SPLIT ? (int)(gridDim.y >> 2) : 1;   // a long table: S dealt over parts
  if (job == 3 && part > 0) return;
  if (job < 0 || job == 3) {
  for (int e = tid; e < k * k; e += BLOCK) {
    const int b = e / k, bb = e % k;
    const int row = d.ev[b], c = d.cls[b];
    const int pc = c == CS ? CP : c;
    T v;
    if (bb == b) v = (c == CM) ? P.baseM[row] : P.baseP[row];
    else v = d.cls[bb] == pc ? P.th[row][d.ev[bb]] : T(1);
    thc[e] = v;
    out[e] = v;
  }
  __syncthreads();
  const int nl = k < 6 ? k : 6;
  for (int e = tid; e < k * 64; e += BLOCK) {
    const int b = e >> 6, l = e & 63;
    T v = 1, u = 1;
    for (int bb = 0; bb < nl; ++bb) if (bb != b && ((l >> bb) & 1)) v *= thc[b * k + bb];
    for (int bb = 6; bb < t; ++bb) if (bb != b && ((l >> (bb - 6)) & 1)) u *= thc[b * k + bb];
    out[k * k + e] = v;
    out[k * k + k * 64 + e] = u;
  }
  }
  if (d.mode != JOINT || d.seedbit < 0 || job == 3) return;
  const int N = d.N, n = N - 1;
  T* o = out + rate_table_size(k);
  for (int c = 0; c < 3; ++c) {                 // 0: dP, 1: dM, 2: dE
    const uint32_t cm = c == 0 ? d.maskP : c == 1 ? d.maskM : d.pairP;
    const int kc = __popc(cm);
    if (job >= 0 && c != job) { o += 1ll << kc; continue; }
    if ((long long)part * BLOCK >= (1ll << kc)) return;        // (uniform: nothing of this table falls to this part)
    __syncthreads();
    // th[i][l] = theta[i][event of the l-th class bit]
    for (int e = tid; e < N * kc; e += BLOCK) {
      const int i = e / kc, l = e % kc;
      uint32_t m = cm;
      for (int q = 0; q < l; ++q) m &= m - 1;
      thc[e] = P.th[i][d.ev[__ffs(m) - 1]];
    }
    __syncthreads();
    const T* dv = c == 1 ? P.dm : P.dp;
    // row N of the table: the observation factors dvec[event of bit l]
    for (int l = tid; l < kc; l += BLOCK) {
      uint32_t m = cm;
      for (int q = 0; q < l; ++q) m &= m - 1;
      thc[N * kc + l] = dv[d.ev[__ffs(m) - 1]];
    }
    __syncthreads();
    // prod_{l in S} th[i][l] split over three 6-bit parts of S: three table reads per (i, S) instead of kc
    // conditional multiplies (kc <= 18; longer lattices keep the loop)
    const bool split = kc <= 18;
    const int np6 = !SPLIT ? 3 : kc <= 6 ? 1 : kc <= 12 ? 2 : 3;   // 6-bit parts in use (SPLIT: the others are left out)
    if (split) {
      for (int e = tid; e < (N + 1) * 192; e += BLOCK) {
        const int i = e / 192, part = (e % 192) >> 6, v = e & 63;
        if (part >= np6) continue;
        T r = 1;
        for (int l = 0; l < 6; ++l) {
          const int ll = part * 6 + l;
          if (ll < kc && ((v >> l) & 1)) r *= thc[i * kc + ll];
        }
        rsplit[e] = r;
      }
      __syncthreads();
    }
    for (long long S = tid + (long long)part * BLOCK; S < (1ll << kc); S += (long long)BLOCK * nparts) {
      const int s0 = (int)(S & 63), s1 = (int)((S >> 6) & 63), s2 = (int)(S >> 12);
      T obs = c == 0 ? P.dp[n] : c == 1 ? P.dm[n] : T(1);
      if (split) { T m = rsplit[N * 192 + s0]; if (np6 > 1) m *= rsplit[N * 192 + 64 + s1]; if (np6 > 2) m *= rsplit[N * 192 + 128 + s2]; obs *= m; }
      else for (int l = 0; l < kc; ++l) if ((S >> l) & 1) obs *= thc[N * kc + l];
      T tot = obs;
      const int rows = c == 2 ? N : n;          // the eq block also carries the seeding rate (row n)
      for (int i = 0; i < rows; ++i) {
        // event i already happened in S ?
        const int bi = c == 0 ? d.bitP[i] : c == 1 ? d.bitM[i] : ((i < n && d.bitP[i] >= 0 && ((d.pairP >> d.bitP[i]) & 1u)) ? d.bitP[i] : -1);
        if (bi >= 0) {
          const int l = __popc(cm & ((1u << bi) - 1u));
          if ((S >> l) & 1) continue;
        }
        T r = c == 1 ? P.baseM[i] : P.baseP[i];
        if (split) { T m = rsplit[i * 192 + s0]; if (np6 > 1) m *= rsplit[i * 192 + 64 + s1]; if (np6 > 2) m *= rsplit[i * 192 + 128 + s2]; r *= m; }
        else for (int l = 0; l < kc; ++l) if ((S >> l) & 1) r *= thc[i * kc + l];
        tot += r;
      }
      o[S] = tot;
    }
    o += 1ll << kc;
  }
}

// LDS tables of a tile: Ltab[rows*64], Utab[rows*64] with rows = max(maxk, 1).
// `scratch` (k*k + k elements) may alias memory that is filled later.
template <typename T>
__device__ __forceinline__ void tile_tables(const Desc& d, const T* __restrict__ tab, uint32_t H, T* Ltab, T* Utab,
                                            T* scratch) {
  const int k = d.k, tid = threadIdx.x, nt = blockDim.x;
  T* thc = scratch;
  T* hx = thc + k * k;
  const int t = k < TB ? k : TB;
  const T* src = tab + d.toff;
  for (int e = tid; e < k * k; e += nt) thc[e] = src[e];
  for (int e = tid; e < k * 64; e += nt) { Ltab[e] = src[k * k + e]; Utab[e] = src[k * k + k * 64 + e]; }
  __syncthreads();
  if (tid < k) {
    T h = thc[tid * k + tid];
    for (int bb = t; bb < k; ++bb) if (bb != tid && ((H >> (bb - t)) & 1u)) h *= thc[tid * k + bb];
    hx[tid] = h;
  }
  __syncthreads();
  for (int e = tid; e < k * 64; e += nt) Utab[e] *= hx[e >> 6];
  __syncthreads();
}

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

#ifndef MMHN_TSB
#define MMHN_TSB 1024
#endif
constexpr int TSB = MMHN_TSB;                // threads per workgroup of the tile solvers (tsolve.h) and of k_psolve2
constexpr int TSB_WPE = TSB == 1024 ? 8 : 4; // waves per SIMD the register budget is sized for (two workgroups per CU)


// 1 / v for a positive, normal-range v (sums of rates): hardware reciprocal + two Newton steps (full precision
// for fp64, no scaling / fix-up sequence of the IEEE division)
__device__ __forceinline__ double fast_rcp(double v) {
  double r = __builtin_amdgcn_rcp(v);
  r = fma(fma(-v, r, 1.0), r, r);
  r = fma(fma(-v, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float fast_rcp(float v) {
  float r = __builtin_amdgcn_rcpf(v);
  r = fmaf(fmaf(-v, r, 1.0f), r, r);
  return r;
}

// wave-uniform values read from LDS land in VGPRs; move them to SGPRs where registers are tight
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ long long sgpr64(long long v) {
  const uint32_t lo = sgpr((uint32_t)v), hi = sgpr((uint32_t)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

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
#ifdef MMHN_STAMPS
// diagnostic build only (scripts/build_variants.sh): wave 0 of every workgroup sums the shader cycles it spends in
// each phase of a tile; the sums leave through a buffer nothing else reads (mmhn_debug_stamps)
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_prev = 0, st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const bool st_on = threadIdx.x < 64
#define STAMP_START do { if (st_on) { __builtin_amdgcn_sched_barrier(0); st_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define STAMP(i) do { if (st_on) { __builtin_amdgcn_sched_barrier(0); const unsigned long long st_now = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); st_sum[i] += st_now - st_prev; st_prev = st_now; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define STAMP_FLUSH(base) do { if (threadIdx.x == 0) for (int si = 0; si < 8; ++si) atomicAdd(&g_stamps[(base) + si], st_sum[si]); } while (0)
#else
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#define STAMP_FLUSH(base)
#endif

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

// ------------------------------------------------------------------------------------
// k_diag: diagonal quantities of one tile.
//   KD_DQ    out = diag(Q)                           (kron_diag, kronvec.py:964-999)
//   KD_LIDG  out = 1 / (Dobs - diag(Q))              (likelihood.py:249-250, vanilla.py:294)
//   KD_ADDQP out += diag(Q) * p                      (completes kronvec(diag=True))
//   KD_DP    out = D_p * p,  KD_DM  out = D_m * p    (diag_scal_p / diag_scal_m; on a single-tumour space KD_DM
//            is the d_m part of vanilla.scal_d_pt, vanilla.py:125-142)
//   KD_QP    out = diag(Q) * p                       (vanilla.kron_diag with a caller-supplied vector, :247-260)
//   KD_SDP   out = [seeding clear] prod d_p * p      (the d_p part of vanilla.scal_d_pt)
// pbit >= 0 keeps only the states that contain index bit pbit (partial_diag_scal_p/m, kronvec.py:605-710:
// the derivative of a Kronecker diagonal w.r.t. one log-rate is the diagonal restricted to "event happened").
// ------------------------------------------------------------------------------------
enum { KD_DQ = 0, KD_LIDG = 1, KD_ADDQP = 2, KD_DP = 3, KD_DM = 4, KD_QP = 5, KD_SDP = 6 };

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_diag(const Desc* __restrict__ descs,
                                                const int2* __restrict__ map,
                                                const Params<T>* __restrict__ par,
                                                const T* __restrict__ p, T* out,
                                                const T* __restrict__ dvec, int what, int maxN, int pbit) {
  extern __shared__ __align__(16) unsigned char smem[];
  Desc& d = *reinterpret_cast<Desc*>(smem);
  T* LcP = reinterpret_cast<T*>(smem + DESC_PAD);
  T* UcP = LcP + maxN * 64;
  T* LcM = UcP + maxN * 64;
  T* UcM = LcM + maxN * 64;
  T* LA = UcM + maxN * 64;     // obs products: A = dp over P bits (SINGLE: non-seeding bits)
  T* UA = LA + 64;
  T* LB = UA + 64;             //               B = dm over M bits (SINGLE: non-seeding bits)
  T* UB = LB + 64;
  const int tid = threadIdx.x;
  const int prob = map[blockIdx.x].x;
  const uint32_t H = (uint32_t)map[blockIdx.x].y;
  load_desc(&d, descs + prob);
  __syncthreads();
  const int k = d.k, N = d.N, n = N - 1;
  const int t = k < TB ? k : TB;
  const uint32_t nelem = 1u << t;
  const long long base = d.off;
  const int R = t > 6 ? 1 << (t - 6) : 1;
  const Params<T>& P = par[d.pset];
  const bool joint = d.mode == JOINT;
  const int nl = k < 6 ? k : 6;

  for (int e = tid; e < N * 64; e += BLOCK) {
    const int i = e >> 6, l = e & 63;
    T vP = 1, vM = 1;
    for (int bb = 0; bb < nl; ++bb)
      if ((l >> bb) & 1) {
        if (d.cls[bb] == CP) vP *= P.th[i][d.ev[bb]];
        else if (d.cls[bb] == CM) vM *= P.th[i][d.ev[bb]];
      }
    LcP[e] = vP; LcM[e] = vM;
    if (l < R) {
      T uP = P.baseP[i], uM = P.baseM[i];
      for (int bb = 6; bb < k; ++bb) {
        const bool set = bb < t ? ((l >> (bb - 6)) & 1) : ((H >> (bb - t)) & 1u);
        if (set) {
          if (d.cls[bb] == CP) uP *= P.th[i][d.ev[bb]];
          else if (d.cls[bb] == CM) uM *= P.th[i][d.ev[bb]];
        }
      }
      UcP[e] = uP; UcM[e] = uM;
    }
  }
  if (tid < 64) {
    const int l = tid;
    T a = 1, b = 1, ua = 1, ub = 1;
    for (int bb = 0; bb < k; ++bb) {
      const bool isA = joint ? d.cls[bb] == CP : bb != d.seedbit;
      const bool isB = joint ? d.cls[bb] == CM : bb != d.seedbit;
      if (bb < 6) {
        if ((l >> bb) & 1) { if (isA) a *= P.dp[d.ev[bb]]; if (isB) b *= P.dm[d.ev[bb]]; }
      } else {
        const bool set = bb < t ? ((l >> (bb - 6)) & 1) : ((H >> (bb - t)) & 1u);
        if (set && l < R) { if (isA) ua *= P.dp[d.ev[bb]]; if (isB) ub *= P.dm[d.ev[bb]]; }
      }
    }
    LA[l] = a; LB[l] = b; UA[l] = ua; UB[l] = ub;
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  for (int r = wave; r < R; r += WAVES) {
    const uint32_t xl = ((uint32_t)r << 6) | (uint32_t)lane;
    if (xl >= nelem) continue;
    const uint32_t x = (H << t) | xl;
    const bool ss = seed_set(d, x);
    const bool sbit = d.seedbit >= 0 && ((x >> d.seedbit) & 1u);
    T dq = 0;
    if (what <= KD_ADDQP || what == KD_QP) {
      if (!joint) {
        for (int i = 0; i < N; ++i)
          if (d.bitP[i] < 0 || !((x >> d.bitP[i]) & 1u)) dq -= LcP[i * 64 + lane] * UcP[i * 64 + r];
      } else if (ss) {
        for (int i = 0; i < n; ++i) {
          if (d.bitP[i] < 0 || !((x >> d.bitP[i]) & 1u)) dq -= LcP[i * 64 + lane] * UcP[i * 64 + r];
          if (d.bitM[i] < 0 || !((x >> d.bitM[i]) & 1u)) dq -= LcM[i * 64 + lane] * UcM[i * 64 + r];
        }
      } else if (eq_noseed(d, x)) {
        for (int i = 0; i < n; ++i)
          if (d.bitP[i] < 0 || !((x >> d.bitP[i]) & 1u)) dq -= LcP[i * 64 + lane] * UcP[i * 64 + r];
        dq -= LcP[n * 64 + lane] * UcP[n * 64 + r];
      }
    }
    const T A = LA[lane] * UA[r], B = LB[lane] * UB[r];
    T res;
    if (what == KD_DQ) {
      res = dq;
    } else if (what == KD_LIDG) {
      T dob;
      if (d.obs == OBS_JOINT) dob = sbit ? A * P.dp[n] + B * P.dm[n] : A;
      else if (d.obs == OBS_ONE) dob = 1;
      else if (d.obs == OBS_MET) dob = sbit ? B * P.dm[n] : A;
      else dob = dvec[base + x];
      res = T(1) / (dob - dq);
    } else if (what == KD_ADDQP) {
      res = out[base + x] + dq * p[base + x];
    } else if (what == KD_DP) {
      res = (sbit ? A * P.dp[n] : A) * p[base + x];
    } else if (what == KD_DM) {
      res = (sbit ? B * P.dm[n] : T(0)) * p[base + x];
    } else if (what == KD_QP) {
      res = dq * p[base + x];
    } else {
      res = sbit ? T(0) : A * p[base + x];
    }
    if (pbit >= 0 && !((x >> pbit) & 1u)) res = 0;
    out[base + x] = res;
  }
}

// ------------------------------------------------------------------------------------
// marginal <-> joint transfers (likelihood.py:557-562, :573-575, :598-602, :617-618)
// compatible joint states of part `part` (0: PT observed first, 1: MT first): all bits of
// the observed tumour and the seeding bit set, the other tumour's bits free, ascending.
// ------------------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ T obs_const(const Desc& dj, const Params<T>& P, int part) {
  // D_p (part 0) or D_m (part 1) on the compatible states: constant, every bit of the class is set
  T c = part == 0 ? P.dp[dj.N - 1] : P.dm[dj.N - 1];
  for (int b = 0; b < dj.k; ++b)
    if (dj.cls[b] == (part == 0 ? CP : CM)) c *= (part == 0 ? P.dp[dj.ev[b]] : P.dm[dj.ev[b]]);
  return c;
}

// rhsS[part problem] = [0 ; D * pi[compatible]];  links[joint problem]: where the right-hand side of the joint adjoint
// comes from (rhs_mode 3 of k_psolve / k_tsolve) - the same constants, so they are written here
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gather_marg(const PatRec* __restrict__ pats,
                                                       const Desc* __restrict__ dJ,
                                                       const Desc* __restrict__ dS,
                                                       const Params<T>* __restrict__ par,
                                                       const T* __restrict__ pi, T* rhsS, JLink<T>* links,
                                                       const int* __restrict__ paired, const WDesc* __restrict__ wds) {
  __shared__ Desc djs;                                     // (the descriptor loops below must not be chains of global loads)
  const PatRec pr = pats[paired[blockIdx.x]];              // grid.x = the paired patients of the batch only
  const int part = blockIdx.y;
  const int sp = part == 0 ? pr.s[0] : pr.s[1];
  if (pr.j < 0) return;
  const bool writes_link = threadIdx.x == 0 && blockIdx.z == 0;
  if (sp < 0) {
    if (writes_link) { links[pr.j].soff[part] = -1; links[pr.j].sk[part] = 0; links[pr.j].cst[part] = 0; }
    return;
  }
  load_desc(&djs, dJ + pr.j);
  const int ksS = dS[sp].k;
  const long long offS = dS[sp].off;
  __syncthreads();
  const Desc& dj = djs;
  const uint32_t fixed = (part == 0 ? dj.maskP : dj.maskM) | (1u << dj.seedbit);
  const uint32_t free_ = part == 0 ? dj.maskM : dj.maskP;
  const uint32_t half = 1u << (ksS - 1);
  const T c = obs_const(dj, par[PS_THETA], part);
  if (writes_link) { links[pr.j].soff[part] = offS; links[pr.j].sk[part] = ksS; links[pr.j].cst[part] = c; }
  // (window-layout problem, wlayout.h: part 0 frees the M bits, part 1 the P bits)
  const bool free_is_row = dj.wl >= 0 && (wds[dj.wl].majP != 0) == (part == 1);
  for (uint32_t e = blockIdx.z * BLOCK + threadIdx.x; e < half; e += gridDim.z * BLOCK) {
    const long long x = dj.wl >= 0 ? wpos_marg<T>(wds[dj.wl], dj.k, free_is_row, e) : (long long)(pdep32(e, free_) | fixed);
    rhsS[offS + e] = 0;
    rhsS[offS + half + e] = c * pi[dj.off + x];
  }
}

// rhsJ[compatible] += D * qS[upper half];  dots[pat][part] = <qS upper half, rhsS upper half>
// one workgroup per patient; launched once per part (the two parts share the all-ones state)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_scatter_marg(const PatRec* __restrict__ pats,
                                                        const Desc* __restrict__ dJ,
                                                        const Desc* __restrict__ dS,
                                                        const Params<T>* __restrict__ par,
                                                        const T* __restrict__ qS,
                                                        const T* __restrict__ rhsS, T* rhsJ,
                                                        T* dots, int part, const int* __restrict__ plist) {
  __shared__ T red[BLOCK];
  const int pat = plist ? plist[blockIdx.x] : (int)blockIdx.x;     // (plist: the patients on the staged kernels)
  const PatRec pr = pats[pat];
  if (pr.j < 0 || pr.s[part] < 0) return;
  const Desc& dj = dJ[pr.j];
  const Desc& ds = dS[pr.s[part]];
  const uint32_t fixed = (part == 0 ? dj.maskP : dj.maskM) | (1u << dj.seedbit);
  const uint32_t free_ = part == 0 ? dj.maskM : dj.maskP;
  const uint32_t half = 1u << (ds.k - 1);
  const T c = obs_const(dj, par[PS_THETA], part);
  T dot = 0;
  for (uint32_t e = threadIdx.x; e < half; e += BLOCK) {
    const uint32_t x = pdep32(e, free_) | fixed;
    const T qv = qS[ds.off + half + e];
    if (rhsJ) rhsJ[dj.off + x] += c * qv;
    dot += qv * rhsS[ds.off + half + e];
  }
  red[threadIdx.x] = dot;
  __syncthreads();
  for (int s = BLOCK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dots[2 * pat + part] = red[0];
}

// head of the staged single-tumour kernels, one workgroup per staged patient: the e_0 right-hand side of an unpaired
// patient's own problem (the whole vector is written: nothing else clears it; a paired row's right-hand sides are written by
// k_gather_marg), and the accumulators the staged kernels add into - the gradient rows (k_grad_rows) and the observation-rate
// marginals (k_bit_marg) of its problems - cleared.  (The other patients' rows are STORED by the small-space kernels.)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_staged_init(const PatRec* __restrict__ pats, const Desc* __restrict__ dS, T* rhsS,
                                                       T* GS, T* bmS, int N, int with_grad, const int* __restrict__ plist) {
  const PatRec pr = pats[plist[blockIdx.x]];
  for (int part = 0; part < 2; ++part) {
    const int sp = pr.s[part];
    if (sp < 0) continue;
    if (with_grad) {
      for (int e = threadIdx.x; e < N * N; e += BLOCK) GS[(long long)sp * N * N + e] = T(0);
      if (threadIdx.x < 64) bmS[(long long)sp * 64 + threadIdx.x] = T(0);
    }
    if (pr.kind <= 2 && part == 0) {
      const long long off = dS[sp].off, V = 1ll << dS[sp].k;
      for (long long e = threadIdx.x; e < V; e += BLOCK) rhsS[off + e] = e == 0 ? e0_scale<T>() : T(0);
    }
  }
}

// per patient: total marginal score, adjoint seeds 1/score for its single problems, log-prob
template <typename T>
__global__ void k_seeds(const PatRec* __restrict__ pats, int npat, const Desc* __restrict__ dS,
                        const Params<T>* __restrict__ par, const T* __restrict__ pS, T* seedS,
                        double* lp, const int* __restrict__ plist) {
  const int ii = blockIdx.x * blockDim.x + threadIdx.x;
  if (ii >= npat) return;
  const int i = plist ? plist[ii] : ii;                    // (plist: the patients on the staged kernels)
  const PatRec pr = pats[i];
  if (pr.kind == 4) return;
  T full = 0;
  for (int part = 0; part < 2; ++part)
    if (pr.s[part] >= 0) {
      const Desc& ds = dS[pr.s[part]];
      full += pS[ds.off + (1ll << ds.k) - 1];
    }
  for (int part = 0; part < 2; ++part)
    if (pr.s[part] >= 0) seedS[pr.s[part]] = T(1) / full;
  double l = log((double)full) - log((double)e0_scale<T>());
  if (pr.kind == 2) {   // likelihood.py:438: log(pTh[-1] * d_rates[-1]), last state has seeding set
    const Desc& ds = dS[pr.s[0]];
    const Params<T>& P = par[PS_THETA];
    double dr = (double)P.dm[ds.N - 1];
    for (int b = 0; b < ds.k; ++b)
      if (b != ds.seedbit) dr *= (double)P.dm[ds.ev[b]];
    l += log(dr);
  }
  lp[i] = l;
}

// ------------------------------------------------------------------------------------
// gradient, stage 1 (joint spaces): class marginals of p (x) q on the seed = 1 half
//   slot 0      W[S]   = - sum_T p[S|T] q[S|T]
//   slot 1 + l  V_l[S] =   sum_T p[S|T] q[S|T|bit_l]      (bit_l not in S, else 0)
// for class c in {P, M}: S over subsets of the class' bits, T over the other class' bits,
// seeding bit set.  Layout at A + d.aoff: class P block [(kP+1)][2^kP], class M block
// [(kM+1)][2^kM], then the eq block of k_eq_flows.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ long long class_block_size(int kc) { return (long long)(kc + 1) << kc; }

// Tile formulation: one workgroup stages a tile of p and q in LDS and runs one phase per class.
// Lane l of a wave always owns the states whose low 6 index bits are l (conflict-free LDS rows,
// coalesced global rows).  A wave task = (slot, setting of the tile's upper class bits): the wave
// walks the settings of the upper other-class bits (independent loads, unrolled), then folds the
// other-class LANE bits with wave shuffles, and the lanes that remain add their partial sum to
// A with one atomic each; partial sums of tiles that differ only in the other class' high bits
// meet there (A is zeroed per call).
#ifndef MMHN_CMB
#define MMHN_CMB 512
#endif
constexpr int CMB = MMHN_CMB;                      // threads per workgroup of k_class_marg
template <typename T>
__global__ __launch_bounds__(CMB) void k_class_marg(const Desc* __restrict__ dJ,
                                                    const int2* __restrict__ map,
                                                    const T* __restrict__ p,
                                                    const T* __restrict__ q, T* A) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* pt = reinterpret_cast<T*>(smem);
  T* qt = pt + (1 << TB);
  const uint32_t blk = xcd_chunked(blockIdx.x, gridDim.x);
  const Desc& d = dJ[map[blk].x];
  const uint32_t H = (uint32_t)map[blk].y;
  if (d.seedbit < 0 || d.wl >= 0) return;
  const int k = d.k;
  const int t = k < TB ? k : TB;
  const uint32_t nelem = 1u << t, tmask = nelem - 1;
  const uint32_t xhi = H << t;
  const uint32_t sbm = 1u << d.seedbit;
  if (d.seedbit >= t && !(xhi & sbm)) return;             // tile lies in the seed = 0 half
  const uint32_t sfix = d.seedbit < t ? sbm : 0u;         // seeding bit inside the tile: fixed to 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: task bookkeeping stays scalar
  constexpr int NWV = CMB / 64;
  __shared__ uint32_t slotbit[34];                                 // index-bit mask of slot s (0 for the diagonal slot)
  __shared__ uint32_t ubtab[64];                                   // upper tile bits of the ou-th class setting
  __shared__ uint32_t futab[64];                                   // upper tile bits of the i-th other-class setting
  __shared__ uint32_t shi_sh;
  for (uint32_t e = tid; e < nelem; e += CMB) { pt[e] = p[d.off + xhi + e]; qt[e] = q[d.off + xhi + e]; }
  const int kP = __popc(d.maskP);
  const bool lane_ok = (uint32_t)lane < nelem && ((sfix & 63u) == 0 || ((uint32_t)lane & sfix));
  for (int c = 0; c < 2; ++c) {
    const uint32_t cmask = c == 0 ? d.maskP : d.maskM;
    const uint32_t cm = cmask & tmask;                    // class bits inside the tile
    const uint32_t fm = tmask & ~cm & ~sfix;              // bits summed over
    const uint32_t cml = cm & 63u, cmu = cm >> 6, fml = fm & 63u, fmu = fm >> 6;
    const uint32_t sfu = sfix >> 6;                       // seeding bit among the upper tile bits (or 0)
    const int nc = __popc(cm), ncl = __popc(cml), kc = __popc(cmask);
    const uint32_t nou = 1u << __popc(cmu), nfu = 1u << __popc(fmu);
    __syncthreads();                                      // tile staged / previous class done with the tables
    if (tid <= kc) slotbit[tid] = tid == 0 ? 0u : pdep32(1u << (tid - 1), cmask);
    if (tid >= 64 && tid < 64 + (int)nou) ubtab[tid - 64] = pdep32((uint32_t)(tid - 64), cmu) | sfu;
    if (tid >= 128 && tid < 128 + (int)nfu) futab[tid - 128] = pdep32((uint32_t)(tid - 128), fmu);
    if (tid == 192) shi_sh = pext32(xhi, cmask & ~tmask);
    __syncthreads();
    T* out = A + d.aoff + (c == 0 ? 0 : class_block_size(kP));
    const uint32_t shi = shi_sh;                          // compact index of the tile's high class bits
    uint32_t own_l = 0;                                   // pext(lane, cml): 6 fixed steps
    {
      int pos = 0;
#pragma unroll
      for (int b6 = 0; b6 < 6; ++b6)
        if ((cml >> b6) & 1u) { own_l |= (((uint32_t)lane >> b6) & 1u) << pos; ++pos; }
    }
    const bool writer = lane_ok && ((uint32_t)lane & fml) == 0;
    // slot lists of this class for this tile: in-tile slots (diagonal + class bits inside the tile; neighbour in
    // LDS) and high slots (class bits above the tile that are still clear in this tile; neighbour tile in HBM)
    __shared__ int lslot[MAXK + 1], hslot[MAXK + 1];
    __shared__ int nls, nhs;
    __syncthreads();
    if (tid == 0) {
      int a = 0, h = 0;
      for (int s = 0; s <= kc; ++s) {
        const uint32_t bl = slotbit[s];
        if ((bl & ~tmask) == 0) lslot[a++] = s;
        else if (!(bl & xhi)) hslot[h++] = s;
      }
      nls = a; nhs = h;
    }
    __syncthreads();
    constexpr int SC = 8;
    const int nl_ = nls, nh_ = nhs;
    const uint32_t nchunk = (uint32_t)(nl_ + SC - 1) / SC;
    const uint32_t ntask_l = nchunk * nou, ntask = ntask_l + (uint32_t)nh_ * nou;
    for (uint32_t task = wave; task < ntask; task += NWV) {
      if (task < ntask_l) {
        // ---- LDS task: SC in-tile slots of one upper class setting; one p load feeds SC q loads
        const int c0 = (int)(task / nou) * SC;
        const uint32_t ou = task % nou;
        const uint32_t ub = ubtab[ou];
        uint32_t bits[SC];
        bool live[SC];
#pragma unroll
        for (int s = 0; s < SC; ++s) {
          const bool in = c0 + s < nl_;
          const uint32_t bl = in ? slotbit[lslot[in ? c0 + s : 0]] : 0u;
          bits[s] = bl;
          live[s] = in && !((bl >> 6) & ub);                // upper-tile class bit already set in this task: no flow
        }
        T acc[SC];
#pragma unroll
        for (int s = 0; s < SC; ++s) acc[s] = 0;
        for (uint32_t i = 0; i < nfu; ++i) {
          const uint32_t xs = (((ub | futab[i]) << 6) | (uint32_t)lane) & tmask;
          const T pv = pt[xs];
#pragma unroll
          for (int s = 0; s < SC; ++s)
            if (live[s]) acc[s] += pv * qt[(xs | bits[s]) & tmask];
        }
#pragma unroll
        for (int s = 0; s < SC; ++s) {
          if (!live[s]) continue;
          T v = (lane_ok && !((uint32_t)lane & bits[s])) ? acc[s] : T(0);
          for (uint32_t m = fml; m; m &= m - 1) v += __shfl_xor(v, (int)(m & (0u - m)));
          if (writer && !((uint32_t)lane & bits[s]) && v != T(0)) {
            const int slot = lslot[c0 + s];
            const long long S = ((long long)shi << nc) | ((long long)ou << ncl) | own_l;
            atomicAdd(&out[((long long)slot << kc) + S], slot == 0 ? -v : v);
          }
        }
      } else {
        // ---- high-slot task: neighbour rows come from another tile (coalesced global rows, 8 in flight)
        const uint32_t tt = task - ntask_l;
        const int slot = hslot[tt / nou];
        const uint32_t ou = tt % nou;
        const uint32_t ub = ubtab[ou];
        const uint32_t bl = slotbit[slot];
        T acc = 0;
        for (uint32_t i0 = 0; i0 < nfu; i0 += 8) {
          T qv[8];
          uint32_t xr[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + u < nfu ? i0 + u : nfu - 1;
            xr[u] = (((ub | futab[i]) << 6) | (uint32_t)lane) & tmask;
            qv[u] = q[d.off + (xhi | bl | xr[u])];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (i0 + u < nfu) acc += pt[xr[u]] * qv[u];
        }
        T v = lane_ok ? acc : T(0);
        for (uint32_t m = fml; m; m &= m - 1) v += __shfl_xor(v, (int)(m & (0u - m)));
        if (writer && v != T(0)) {
          const long long S = ((long long)shi << nc) | ((long long)ou << ncl) | own_l;
          atomicAdd(&out[((long long)slot << kc) + S], v);
        }
      }
    }
  }
}

// eq block (seed = 0 states with PT == MT): subsets e of the paired events, x0 = both bits
//   slot 0      -p[x0] q[x0]
//   slot 1 + l   p[x0] q[x0 | pair_l]          (pair_l not in e)
//   slot ke + 1  p[x0] q[x0 | seedbit]         (0 if seeding inactive)
template <typename T>
__device__ __forceinline__ void eq_flows_body(const Desc& d, const WDesc* __restrict__ wds, const T* __restrict__ p,
                                              const T* __restrict__ q, T* A, int tid, int nthreads) {
  const int ke = __popc(d.pairP);
  T* out = A + d.aoff + class_block_size(__popc(d.maskP)) + class_block_size(__popc(d.maskM));
  const long long items = (long long)(ke + 2) << ke;
  for (long long it = tid; it < items; it += nthreads) {
    const int slot = (int)(it >> ke);
    const uint32_t e = (uint32_t)(it & ((1ll << ke) - 1));
    const uint32_t xp = pdep32(e, d.pairP);
    const uint32_t x0 = xp | (xp << 1);
    T v;
    if (slot == 0) {
      v = -p[d.off + x0] * q[d.off + x0];
    } else if (slot <= ke) {
      const uint32_t bp = pdep32(1u << (slot - 1), d.pairP);
      v = (x0 & bp) ? T(0) : p[d.off + x0] * q[d.off + (x0 | bp | (bp << 1))];
    } else {
      // (the seeded half of a window-layout problem is not in index order)
      const long long xs = d.wl >= 0 ? (1ll << (d.k - 1)) + wpos_nat<T>(wds[d.wl], x0) : (long long)(x0 | (1u << d.seedbit));
      v = d.seedbit >= 0 ? p[d.off + x0] * q[d.off + xs] : T(0);
    }
    out[it] = v;
  }
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_eq_flows(const Desc* __restrict__ dJ, const WDesc* __restrict__ wds,
                                                    const T* __restrict__ p,
                                                    const T* __restrict__ q, T* A) {
  eq_flows_body(dJ[blockIdx.x], wds, p, q, A, (int)threadIdx.x, BLOCK);
}

// ------------------------------------------------------------------------------------
// k_pclass: the class marginals of k_class_marg, one workgroup per PATIENT, accumulated in registers.
//
// For class c the joint vectors are viewed as matrices p[S][F], q[S][F] (S = setting of the class-c bits, F =
// setting of the other class's bits, seeding = 1).  The outputs are row dot products,
//   W[S] = -sum_F p[S][F] q[S][F],      V_b[S] = sum_F p[S][F] q[S | b][F]   (b a class bit clear in S),
// so a tile here is NOT the low TB index bits: it holds the lowest a = min(kc, PCA) class bits (all of them when
// kc <= PCA) and is filled up to TB bits with the lowest other-class bits.  Every in-tile slot's neighbour is
// then in LDS at a constant offset, the sum over the remaining F bits runs over the patient's tiles with the
// accumulators in registers, and each output is written once per patient (no per-tile atomics, no cross-lane
// reductions).  Tiles are staged in the permuted order e' = S + RS * F (RS = 2^a + pad), gathered from HBM in
// memory order (contiguous runs of >= 128 B whenever the four lowest index bits are tile bits).
// Class bits above the a-th (kc > PCA) make an outer loop over blocks Shi; their slots take a second pass per
// tile with the neighbour block's q staged over qt.  Per class pass p and q are read once (+ the neighbour
// blocks), i.e. about 4 vector-halves per patient against 2 + the high-slot rows of k_class_marg.
// ------------------------------------------------------------------------------------
constexpr int PCA = 10;                            // class bits inside a tile (two accumulator sets per wave)
constexpr int PCH = 5;                             // class bits above the tile (kc <= PCA + PCH)
constexpr int PC_PAD = 4;                          // row pad (elements) of the staged layout: conflict-free ds_write_b64
constexpr int PC_LDS_ELEMS = 2 * ((1 << TB) + PC_PAD * 64) + (1 << (PCA - 1));

__device__ __forceinline__ uint32_t low_bits(uint32_t m, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n && m; ++i) { r |= m & (0u - m); m &= m - 1; }
  return r;
}

// the outer loop of a class pass (host: Engine::pclass_items mirrors it): o runs over the settings of the other class's bits
// above the tile
__host__ __device__ inline int pclass_outer_bits(int kc, int kf) {
  const int a = kc < PCA ? kc : PCA;
  const int nfl = kf < TB - a ? kf : TB - a;
  return kf - nfl;
}

// SPLIT: the launch runs over work items {problem, 0 / 1: class pass, 2: the eq block's flows (k_eq_flows), o0, o1: range of
// the pass's outer loop} - short launches are one workgroup's chain long, and a large problem is several workgroups (their
// partial sums meet in the atomics of the flush).  Otherwise one workgroup per problem does both passes.
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(CMB, 4) void k_pclass(const Desc* __restrict__ dJ, const WDesc* __restrict__ wds, const T* __restrict__ p,
                                                   const T* __restrict__ q, T* A, const int4* __restrict__ items = nullptr) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* pt = reinterpret_cast<T*>(smem);
  T* qt = pt + (1 << TB) + PC_PAD * 64;                  // + 2^(PCA-1) slack behind it for the neighbour reads
  int4 item = int4{(int)blockIdx.x, 0, 0, 0};
  if (SPLIT) item = items[blockIdx.x];
  const Desc& d = dJ[item.x];
  if (SPLIT && item.y == 2) { eq_flows_body(d, wds, p, q, A, (int)threadIdx.x, CMB); return; }
  const int seedbit = d.seedbit;
  if (seedbit < 0 || d.wl >= 0) return;                   // (window-layout problems: k_wclass)
  const int k = d.k;
  const uint32_t sbm = 1u << seedbit;
  const uint32_t allbits = (k >= 32 ? 0xffffffffu : ((1u << k) - 1u)) & ~sbm;
  const uint32_t maskP = d.maskP, maskM = d.maskM;
  if (__popc(maskP) > PCA + PCH || __popc(maskM) > PCA + PCH) return;     // left to k_class_marg
  const long long off = d.off;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NWV = CMB / 64;                           // 8 waves: rows w, w + 8, ... of a tile
  constexpr int NST = (1 << TB) / CMB;
  constexpr int NROW = (1 << TB) / 64 / NWV;              // rows per wave of a full tile
  const int kP = __popc(maskP);
  for (int c = SPLIT ? item.y : 0; c < (SPLIT ? item.y + 1 : 2); ++c) {
    const uint32_t cmask = c == 0 ? maskP : maskM;
    const uint32_t other = allbits & ~cmask;
    const int kc = __popc(cmask), kf = __popc(other);
    const int a = kc < PCA ? kc : PCA;
    const int nfl = kf < TB - a ? kf : TB - a;
    const int t2 = a + nfl, nh = kc - a, no = kf - nfl;
    const uint32_t nelem2 = 1u << t2;
    const uint32_t clow = low_bits(cmask, a), chigh = cmask & ~clow;
    const uint32_t fill = low_bits(other, nfl), omask = other & ~fill;
    const uint32_t tilemask = clow | fill;
    const uint32_t mA = (1u << a) - 1u;
    const uint32_t RS = (1u << a) + (a >= 6 ? PC_PAD : 0);
    T* out = A + d.aoff + (c == 0 ? 0 : class_block_size(kP));
    // staging map of this thread: memory-order element m of the tile -> offset in the vector, slot in LDS
    uint32_t goff[NST], eo[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      // a thread takes pairs of memory-adjacent tile elements (2 j, 2 j + 1): one 16-byte load when index bit 0
      // is a tile bit
      const uint32_t m = 2u * ((uint32_t)tid + CMB * (u >> 1)) + (u & 1);
      const uint32_t g = pdep32(m, tilemask);
      goff[u] = g;
      eo[u] = m < nelem2 ? pext32(g, clow) + RS * pext32(g, fill) : 0xffffffffu;
    }
    const bool wide = (tilemask & 1u) && nelem2 >= 2;     // then goff[2 j + 1] = goff[2 j] + 1, both valid or both not
    __syncthreads();                                      // previous class done with the staged tile
    for (int e = tid; e < PC_LDS_ELEMS; e += CMB) pt[e] = T(0);
    const uint32_t nrows1 = nelem2 > 64 ? nelem2 >> 6 : 1;   // rows of 64 states (a small tile is one partial row)
    // one block of class settings (fixed bits above the tile); two instantiations so that the common case
    // kc <= PCA carries no accumulators for slots above the tile
    auto block = [&](auto hic, uint32_t Shi) {
      constexpr bool HI = decltype(hic)::value;             // class bits above the tile exist (kc > PCA)
      T acc[2][PCA + 1], acch[2][HI ? PCH : 1];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j <= PCA; ++j) acc[s][j] = T(0);
#pragma unroll
        for (int j = 0; j < (HI ? PCH : 1); ++j) acch[s][j] = T(0);
      }
      const uint32_t cbase = pdep32(Shi, chigh);
      for (uint32_t o = SPLIT ? (uint32_t)item.z : 0u; o < (SPLIT ? (uint32_t)item.w : (1u << no)); ++o) {
        const uint32_t obase = sbm | pdep32(o, omask);
        {
          const long long base = off + (long long)(obase | cbase);
          T rp[NST], rq[NST];
          if (wide) {
            struct alignas(2 * sizeof(T)) pair_t { T a, b; };
#pragma unroll
            for (int u = 0; u < NST; u += 2) {
              pair_t vp{T(0), T(0)}, vq{T(0), T(0)};
              if (eo[u] != 0xffffffffu) {
                vp = *reinterpret_cast<const pair_t*>(p + base + goff[u]);
                vq = *reinterpret_cast<const pair_t*>(q + base + goff[u]);
              }
              rp[u] = vp.a; rp[u + 1] = vp.b; rq[u] = vq.a; rq[u + 1] = vq.b;
            }
          } else {
#pragma unroll
            for (int u = 0; u < NST; ++u) {
              const bool ok = eo[u] != 0xffffffffu;
              rp[u] = ok ? p[base + goff[u]] : T(0);
              rq[u] = ok ? q[base + goff[u]] : T(0);
            }
          }
          __syncthreads();                                // previous tile reduced
#pragma unroll
          for (int u = 0; u < NST; ++u)
            if (eo[u] != 0xffffffffu) { pt[eo[u]] = rp[u]; qt[eo[u]] = rq[u]; }
          __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
          const uint32_t r = (uint32_t)w + NWV * i;
          if (r >= nrows1) break;
          const uint32_t e = (r << 6) | (uint32_t)lane;
          const uint32_t ea = (e & mA) + RS * (e >> a);
          const T pv = pt[ea];
          acc[i & 1][0] += pv * qt[ea];
#pragma unroll
          for (int j = 0; j < PCA; ++j)
            if (j < a) acc[i & 1][1 + j] += pv * qt[ea + (1u << j)];
        }
        // slots of the class bits above the tile: the neighbour block's q replaces qt for one pass each
#pragma unroll
        for (int hb = 0; hb < (HI ? PCH : 0); ++hb) {
          if (hb >= nh || ((Shi >> hb) & 1u)) continue;
          const long long nb = off + (long long)(obase | pdep32(Shi | (1u << hb), chigh));
          T rq[NST];
#pragma unroll
          for (int u = 0; u < NST; ++u) rq[u] = eo[u] != 0xffffffffu ? q[nb + goff[u]] : T(0);   // (8-byte loads: rare path)
          __syncthreads();
#pragma unroll
          for (int u = 0; u < NST; ++u)
            if (eo[u] != 0xffffffffu) qt[eo[u]] = rq[u];
          __syncthreads();
#pragma unroll
          for (int i = 0; i < NROW; ++i) {
            const uint32_t r = (uint32_t)w + NWV * i;
            if (r >= nrows1) break;
            const uint32_t e = (r << 6) | (uint32_t)lane;
            const uint32_t ea = (e & mA) + RS * (e >> a);
            acch[i & 1][hb] += pt[ea] * qt[ea];
          }
        }
      }
      // ---- flush this block's rows of the class-marginal tables
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint32_t Sl = ((((uint32_t)w + NWV * s) << 6) | (uint32_t)lane) & mA;
        const long long S = ((long long)Shi << a) | Sl;
        if (acc[s][0] != T(0)) atomicAdd(&out[S], -acc[s][0]);
#pragma unroll
        for (int j = 0; j < PCA; ++j)
          if (j < a && !((Sl >> j) & 1u) && acc[s][1 + j] != T(0))
            atomicAdd(&out[((long long)(1 + j) << kc) + S], acc[s][1 + j]);
#pragma unroll
        for (int hb = 0; hb < (HI ? PCH : 0); ++hb)
          if (hb < nh && !((Shi >> hb) & 1u) && acch[s][hb] != T(0))
            atomicAdd(&out[((long long)(1 + a + hb) << kc) + S], acch[s][hb]);
      }
    };
    if (nh == 0) block(std::false_type{}, 0u);
    else for (uint32_t Shi = 0; Shi < (1u << nh); ++Shi) block(std::true_type{}, Shi);
  }
}

// ------------------------------------------------------------------------------------
// gradient, stage 2: flows of event i over one subset lattice -> row i of a G matrix.
//   f(S)   = rate_i(S) * (A_slot(i)[S] + A_0[S])    if event i can still fire from S
//   G[i,i] = sum_S f(S);  G[i, ev(l)] = sum_{S contains l} f(S);  kind M: G[i,n] = G[i,i]
// kinds: GK_P / GK_M class marginals of a joint space, GK_E its eq block (rows 0..n),
//        GK_S a single-tumour space with A formed on the fly from (p, q)
//        (vanilla.py:328-393 in flow form).
// grid = (problems, N); one workgroup per (problem, event).
// ------------------------------------------------------------------------------------
enum { GK_P = 0, GK_M = 1, GK_E = 2, GK_S = 3 };

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// wave-wide sum with DPP moves only (VALU; the shuffle form of wave_sum is twelve dependent LDS-pipe permutes per
// fp64 value): quad, half-row and row mirrors, then the gfx9 row broadcasts; the total lands in lane 63
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, false);
  return v + __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
__device__ __forceinline__ double lane63(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ float lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v) {
  v = dpp_add<0xB1, 0xF>(v);          // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);          // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);         // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);         // row_mirror: every lane of a 16-lane row holds the row's sum
  v = dpp_add<0x142, 0xA>(v);         // row_bcast15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);         // row_bcast31 into rows 2 and 3: lane 63 holds the wave's sum
  return lane63(v);
}
// stage S of that reduction (it pairs the lanes that differ in bit S of the lane index)
template <int S, typename T>
__device__ __forceinline__ T dpp_stage(T v) {
  if constexpr (S == 0) return dpp_add<0xB1, 0xF>(v);
  else if constexpr (S == 1) return dpp_add<0x4E, 0xF>(v);
  else if constexpr (S == 2) return dpp_add<0x141, 0xF>(v);
  else if constexpr (S == 3) return dpp_add<0x140, 0xF>(v);
  else if constexpr (S == 4) return dpp_add<0x142, 0xA>(v);
  else return dpp_add<0x143, 0xC>(v);
}
template <int S, typename T>
__device__ __forceinline__ T dpp_stages_from(T v) {
  if constexpr (S < 6) return dpp_stages_from<S + 1>(dpp_stage<S>(v));
  else return v;
}
// total = sum over the lanes of v, M[l] = sum over the lanes whose index has bit l (l < nb; the others are left alone).
// The masked sums share the stages below their bit with the total: after stages 0 .. l-1 a lane holds the sum of its
// group of 2^l lanes, the groups with bit l clear are dropped there, and stages l .. 5 finish - 27 stages for the seven
// sums instead of 42.
template <typename T>
__device__ __forceinline__ void wave_bit_sums(T v, int lane, int nb, T& total, T (&M)[6]) {
  const T p0 = v;
  const T p1 = dpp_stage<0>(p0), p2 = dpp_stage<1>(p1), p3 = dpp_stage<2>(p2), p4 = dpp_stage<3>(p3), p5 = dpp_stage<4>(p4);
  total = lane63(dpp_stage<5>(p5));
  if (nb > 0) M[0] = lane63(dpp_stages_from<0>((lane & 1) ? p0 : T(0)));
  if (nb > 1) M[1] = lane63(dpp_stages_from<1>((lane & 2) ? p1 : T(0)));
  if (nb > 2) M[2] = lane63(dpp_stages_from<2>((lane & 4) ? p2 : T(0)));
  if (nb > 3) M[3] = lane63(dpp_stages_from<3>((lane & 8) ? p3 : T(0)));
  if (nb > 4) M[4] = lane63(dpp_stages_from<4>((lane & 16) ? p4 : T(0)));
  if (nb > 5) M[5] = lane63(dpp_stages_from<5>((lane & 32) ? p5 : T(0)));
}

// grid = (work list of (problem, subset chunk), ceil(N / WAVES)); wave w owns event i = blockIdx.y * WAVES + w
// and strides the subsets S across its lanes; all reductions are wave-level.
constexpr int GR_CHUNK = 11;                      // subsets per workgroup of k_grad_rows: 2^11

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_grad_rows(const Desc* __restrict__ descs,
                                                     const Params<T>* __restrict__ par,
                                                     const T* __restrict__ A,
                                                     const T* __restrict__ p,
                                                     const T* __restrict__ q, T* G, int kind_arg,
                                                     T* DJ, const int2* __restrict__ chunks,
                                                     int nprob, long long gstride) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* Tlo = reinterpret_cast<T*>(smem);          // [WAVES][3][64]: rate products over subset bits 0-5, 6-11, 12-17
  T* rowbuf = Tlo + WAVES * 192;                // [WAVES][32]
  __shared__ int lev[32];                       // event of local bit l
  // (problem, subset chunk) work list; kind_arg < 0: the kind rides in bits 24+ of the chunk field and selects the G matrix
  const int prob = chunks[blockIdx.x].x;
  const int kind = kind_arg < 0 ? chunks[blockIdx.x].y >> 24 : kind_arg;
  const int chunk = chunks[blockIdx.x].y & 0xffffff;
  if (kind_arg < 0) G += (long long)kind * gstride;
  const Desc& d = descs[prob];
  const int N = d.N, n = N - 1;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int i = blockIdx.y * WAVES + w;
  const Params<T>& P = par[d.pset];

  uint32_t cm;
  if (kind == GK_P) cm = d.maskP; else if (kind == GK_M) cm = d.maskM;
  else if (kind == GK_E) cm = d.pairP; else cm = (1u << d.k) - 1u;
  const int kc = __popc(cm);
  if (tid == 0) {
    uint32_t m = cm; int l = 0;
    while (m) { const int b = __ffs(m) - 1; lev[l] = d.ev[b]; ++l; m &= m - 1; }
  }
  __syncthreads();
  // row N (joint kinds only): observation-rate gradient from the same marginals,
  //   sum_S D(S) * (sum p q)[S] [l in S]  with D(S) = d0 * prod_{l in S} dvec[ev(l)]
  //   (x_partial_D_y, likelihood.py:204-228: GK_P -> d_dp on the seed = 1 half, GK_M -> d_dm,
  //    GK_E -> d_dp on the seed = 0 states, where D_m = 0)
  const bool drow = i == N;
  if (i > N || (drow && (kind == GK_S || DJ == nullptr))) return;
  T* row = drow ? DJ + ((long long)kind * nprob + prob) * N : G + ((long long)prob * N + i) * N;
  T* rb = rowbuf + w * 32;
  if (lane < 32) rb[lane] = 0;
  const T* fvec = drow ? (kind == GK_M ? P.dm : P.dp) : P.th[i < N ? i : 0];

  bool rowvalid = true;
  T base = drow ? (kind == GK_P ? -P.dp[n] : kind == GK_M ? -P.dm[n] : T(-1)) : (kind == GK_M ? P.baseM[i] : P.baseP[i]);
  if ((kind == GK_P || kind == GK_M) && ((!drow && i >= n) || d.seedbit < 0)) rowvalid = false;
  if (kind == GK_E && d.mode != JOINT) rowvalid = false;
  int slot = -1;                                // local slot of event i; kc = extra always-free slot
  if (drow) slot = -1;
  else if (kind == GK_E && i == n) slot = kc;
  else for (int l = 0; l < kc; ++l) if (lev[l] == i) slot = l;

  if (rowvalid) {
    const int klo = kc < 6 ? kc : 6;
    const int kin = kc < GR_CHUNK ? kc : GR_CHUNK;   // subset bits that vary inside this workgroup's chunk
    const int nhi = kin - klo;
#pragma unroll
    for (int part = 0; part < 3; ++part) {          // one table per 6-bit part of the subset index
      T v = 1;
      for (int l = 0; l < 6; ++l) {
        const int ll = part * 6 + l;
        if (ll < kc && ((lane >> l) & 1)) v *= fvec[lev[ll]];
      }
      Tlo[w * 192 + part * 64 + lane] = v;
    }
    // per-lane sums over the subsets that have subset bit 6 + l (wave-uniform tests: a scalar branch around one add)
    constexpr int NHI = GR_CHUNK - 6;
    T ha[NHI];
#pragma unroll
    for (int l = 0; l < NHI; ++l) ha[l] = 0;
    const T* Ab = nullptr;
    if (kind != GK_S) {
      long long o = d.aoff;
      if (kind != GK_P) o += class_block_size(__popc(d.maskP));
      if (kind == GK_E) o += class_block_size(__popc(d.maskM));
      Ab = A + o;
    }
    const long long nS = 1ll << kc;
    // this workgroup takes the subsets [chunk, chunk + 1) << GR_CHUNK (long lattices are split, partial rows are
    // added up)
    const long long Sbeg = (long long)chunk << GR_CHUNK;
    const long long Send = nS < Sbeg + (1ll << GR_CHUNK) ? nS : Sbeg + (1ll << GR_CHUNK);
    T tot = 0;
    constexpr int GU = 4;                          // chunks of 64 subsets in flight per wave
    for (long long S00 = Sbeg; S00 < Send; S00 += 64 * GU) {
      T a0[GU], a1[GU];
      bool live[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const long long S = S00 + 64 * u + lane;
        const uint32_t s = (uint32_t)S;
        const bool blocked = slot >= 0 && slot < kc && ((s >> slot) & 1u);
        live[u] = S < nS && !blocked;
        a0[u] = 0; a1[u] = 0;
        if (live[u]) {
          if (kind == GK_S) {
            const T pv = p[d.off + s];
            a0[u] = -pv * q[d.off + s];
            if (slot >= 0) a1[u] = pv * q[d.off + (s | (1u << slot))];
          } else {
            a0[u] = Ab[S];
            if (slot >= 0) a1[u] = Ab[((long long)(slot + 1) << kc) + S];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const long long S0 = S00 + 64 * u;
        if (S0 >= nS) break;
        const uint32_t s = (uint32_t)(S0 + lane);
        T urate = base;                            // wave-uniform part of the rate: two table reads (bits 6-17)
        urate *= Tlo[w * 192 + 64 + ((S0 >> 6) & 63)] * Tlo[w * 192 + 128 + ((S0 >> 12) & 63)];
        for (int l = 18; l < kc; ++l) if ((S0 >> l) & 1) urate *= fvec[lev[l]];
        const T f = live[u] ? urate * Tlo[w * 192 + (s & 63u)] * (a0[u] + a1[u]) : T(0);
        tot += f;
#pragma unroll
        for (int l = 0; l < NHI; ++l) if (l < nhi && ((S0 >> (6 + l)) & 1)) ha[l] += f;
      }
    }
    const T total = wave_sum(tot);
    if (lane == 0) {
      if (drow) { if (kind != GK_E) rb[n] = total; }
      else { rb[i] = total; if (kind == GK_M) rb[n] = total; }
    }
    for (int l = 0; l < klo; ++l) {
      const T m = wave_sum(((lane >> l) & 1) ? tot : T(0));
      if (lane == 0 && lev[l] != i) rb[lev[l]] = m;
    }
#pragma unroll
    for (int l = 0; l < NHI; ++l) {
      if (l < nhi) {                                             // (nhi > 0 only with klo = 6)
        const T m = wave_sum(ha[l]);
        if (lane == 0 && lev[6 + l] != i) rb[lev[6 + l]] = m;
      }
    }
    // bits at or above the chunk size are the same for every subset of the chunk
    if (lane == 0)
      for (int l = kin; l < kc; ++l)
        if (((Sbeg >> l) & 1) && lev[l] != i) rb[lev[l]] = total;
  }
  if (lane < N && rb[lane] != T(0)) atomicAdd(&row[lane], rb[lane]);
}

// ------------------------------------------------------------------------------------
// weighted bit marginals for the observation-rate gradients
//   out[prob][0][b] = sum_{x contains b} q p W_A(x),  out[prob][1][b] likewise with W_B
// JOINT: W_A = D_p, W_B = D_m (x_partial_D_y, likelihood.py:204-228);
// SINGLE/OBS_MET: W_A = d_p part, W_B = d_m part of scal_d_pt (vanilla.py:125-203).
// One workgroup per tile, atomics per (tile, bit).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_bit_marg(const Desc* __restrict__ descs,
                                                    const int2* __restrict__ map,
                                                    const Params<T>* __restrict__ par,
                                                    const T* __restrict__ p,
                                                    const T* __restrict__ q, T* out) {
  // per-wave partials: [WAVES][2 weights][13] = total + marginals of the 12 in-tile bits
  __shared__ T part[WAVES][2][16];
  const Desc& d = descs[map[blockIdx.x].x];
  const uint32_t H = (uint32_t)map[blockIdx.x].y;
  const int k = d.k, n = d.N - 1;
  const int t = k < TB ? k : TB;
  const uint32_t nelem = 1u << t;
  const Params<T>& P = par[d.pset];
  const bool joint = d.mode == JOINT;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  constexpr int NJ = (1 << TB) / BLOCK;          // 16 strided states per thread: bits 8..11 = j
  T tot[2] = {0, 0};
  T mj[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const uint32_t xl = (uint32_t)j * BLOCK + tid;
    if (xl < nelem) {
      const uint32_t x = (H << t) | xl;
      T a = 1, b = 1;
      for (int bb = 0; bb < k; ++bb)
        if ((x >> bb) & 1u) {
          if (joint ? d.cls[bb] == CP : bb != d.seedbit) a *= P.dp[d.ev[bb]];
          if (joint ? d.cls[bb] == CM : bb != d.seedbit) b *= P.dm[d.ev[bb]];
        }
      const bool sbit = d.seedbit >= 0 && ((x >> d.seedbit) & 1u);
      T wA, wB;
      if (joint) { wA = sbit ? a * P.dp[n] : a; wB = sbit ? b * P.dm[n] : T(0); }
      else { wA = sbit ? T(0) : a; wB = sbit ? b * P.dm[n] : T(0); }
      const T pq = p[d.off + x] * q[d.off + x];
      const T vA = pq * wA, vB = pq * wB;
      tot[0] += vA; tot[1] += vB;
#pragma unroll
      for (int l = 0; l < 4; ++l) if ((j >> l) & 1) { mj[0][l] += vA; mj[1][l] += vB; }
    }
  }
#pragma unroll
  for (int ww = 0; ww < 2; ++ww) {
    const T s = wave_sum(tot[ww]);
    if (lane == 0) part[w][ww][12] = s;
#pragma unroll
    for (int l = 0; l < 6; ++l) {
      const T m = wave_sum(((lane >> l) & 1) ? tot[ww] : T(0));
      if (lane == 0) part[w][ww][l] = m;
    }
    if (lane == 0) { part[w][ww][6] = (w & 1) ? s : T(0); part[w][ww][7] = (w & 2) ? s : T(0); }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const T m = wave_sum(mj[ww][l]);
      if (lane == 0) part[w][ww][8 + l] = m;
    }
  }
  __syncthreads();
  T* o = out + (long long)map[blockIdx.x].x * 64;
  if (tid < 2 * 32) {
    const int ww = tid >> 5, b = tid & 31;
    if (b < k) {
      T m = 0;
      if (b < t) { for (int v = 0; v < WAVES; ++v) m += part[v][ww][b]; }
      else if ((H >> (b - t)) & 1u) { for (int v = 0; v < WAVES; ++v) m += part[v][ww][12]; }
      if (m != T(0)) atomicAdd(&o[ww * 32 + b], m);
    }
  }
}

// ------------------------------------------------------------------------------------
// per-patient assembly (likelihood.py:441-512, :623-731) and cohort reduction
// out[pat] = [ lp, G[N][N], d_dp[N], d_dm[N] ]
// ------------------------------------------------------------------------------------
template <typename T>
struct AsmArgs {
  const PatRec* pats; const Desc* dJ; const Desc* dS; const Params<T>* par;
  const T* GS; const T* GJ; long long gj_stride; const T* dots; const T* DJ; long long dj_stride; const T* bmS;
  const double* lp; int N; int with_grad;
};

// element e of patient `pat`'s row
template <typename T>
__device__ __forceinline__ double assemble_elem(const AsmArgs<T>& a, const PatRec& pr, int pat, int e) {
  const int N = a.N, n = N - 1;
  if (pr.kind == 4) {                              // _grad_prim_obs_az, likelihood.py:464-478
    const Params<T>& P0 = a.par[PS_THETA];
    const int q = e - 1;
    const bool diag = a.with_grad && q >= 0 && q < N * N && q / N == q % N;
    if (e != 0 && !diag) return 0.0;
    double s = 0;
    for (int i = 0; i < N; ++i) s += (double)P0.th[i][i];
    return e == 0 ? -log1p(s) : -(double)P0.th[q / N][q / N] / (1.0 + s);
  }
  if (e == 0) return a.lp[pat];
  if (!a.with_grad) return 0.0;
  if (e < 1 + N * N) {                             // theta gradient
    const int q = e - 1, i = q / N, j = q % N;
    double g = 0;
    for (int part = 0; part < 2; ++part)
      if (pr.s[part] >= 0) {
        const bool prim_space = a.dS[pr.s[part]].pset == PS_PRIM;
        if (!(prim_space && j == n && i < n)) g += (double)a.GS[(long long)pr.s[part] * N * N + q];
      }
    if (pr.j >= 0)
      for (int kd = 0; kd < 3; ++kd) g += (double)a.GJ[kd * a.gj_stride + (long long)pr.j * N * N + q];
    return g;
  }
  // observation-rate gradients
  const int r0 = e - 1 - N * N, i = r0 % N;
  double gp = 0, gm = 0;
  for (int part = 0; part < 2; ++part) {
    if (pr.s[part] < 0) continue;
    const Desc& ds = a.dS[pr.s[part]];
    const T* g = a.GS + (long long)pr.s[part] * N * N;
    double dd = (double)g[i * N + i];              // d_diag[i] = -sum_{r != i} val[r, i], vanilla.py:392
#pragma unroll 7
    for (int r = 0; r < N; ++r) dd -= (double)g[r * N + i];
    if (pr.kind == 3) {
      const Desc& dj = a.dJ[pr.j];
      const double dot = (double)a.dots[2 * pat + part];
      if (part == 0) { gm += dd; if (i == n || dj.bitP[i] >= 0) gp += dot; }
      else           { gp += dd; if (i == n || dj.bitM[i] >= 0) gm += dot; }
    } else if (pr.kind == 2) {                     // _grad_met_obs, likelihood.py:481-512
      const T* bm = a.bmS + (long long)pr.s[0] * 64;
      const int b = ds.bitP[i];
      if (b >= 0) {
        if (b != ds.seedbit) gp -= (double)bm[b];
        gm += 1.0 - (double)bm[32 + b];
      }
    } else {
      gp += dd;                                    // _grad_prim_obs, likelihood.py:441-461
    }
  }
  if (pr.kind == 3) {                              // minus x_partial_D_y(q_J, pi), likelihood.py:536,694-695
    const long long o = (long long)pr.j * N + i;
    gp -= (double)a.DJ[GK_P * a.dj_stride + o] + (double)a.DJ[GK_E * a.dj_stride + o];
    gm -= (double)a.DJ[GK_M * a.dj_stride + o];
  }
  return r0 < N ? gp : gm;
}

// rows of all patients (mmhn_patient_grads)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_finalize(AsmArgs<T> a, double* out) {
  const PatRec pr = a.pats[blockIdx.x];
  const int stride = 1 + a.N * a.N + 2 * a.N;
  double* o = out + (long long)blockIdx.x * stride;
  for (int e = threadIdx.x; e < (a.with_grad ? stride : 1); e += BLOCK) o[e] = assemble_elem(a, pr, (int)blockIdx.x, e);
}

// cohort sums, stage 1: thread = element e (coalesced over the rows), workgroup (blockIdx.y) = a chunk of `per`
// consecutive patients added in index order; part[chunk][cls][e], cls 0: type != 0 (EM), 1: type 0 (NM)
constexpr int RED_MAX_CHUNKS = 128;
__host__ __device__ inline int red_per(int npat) { return max(32, (npat + RED_MAX_CHUNKS - 1) / RED_MAX_CHUNKS); }
__global__ __launch_bounds__(BLOCK) void k_reduce_rows(const PatRec* __restrict__ pats, int npat, int per,
                                                       const double* __restrict__ out, int stride, int nelem,
                                                       double* __restrict__ part) {
  const int e = blockIdx.x * BLOCK + threadIdx.x;
  if (e >= nelem) return;
  const int i0 = blockIdx.y * per, i1 = min(npat, i0 + per);
  double acc0 = 0, acc1 = 0;
#pragma unroll 8
  for (int i = i0; i < i1; ++i) {
    const int kd = pats[i].kind;
    const double v = out[(long long)i * stride + e];
    if (kd == 0 || kd == 4) acc1 += v; else acc0 += v;
  }
  part[((long long)blockIdx.y * 2 + 0) * stride + e] = acc0;
  part[((long long)blockIdx.y * 2 + 1) * stride + e] = acc1;
}

// sums[cls][e] += the chunk sums in chunk order
__global__ __launch_bounds__(BLOCK) void k_reduce_parts(const double* __restrict__ part, int stride, int nelem, int nchunk,
                                                        double* sums) {
  const int e = blockIdx.x * BLOCK + threadIdx.x, cls = blockIdx.y;
  if (e >= nelem) return;
  double acc = 0;
#pragma unroll 8
  for (int c = 0; c < nchunk; ++c) acc += part[((long long)c * 2 + cls) * stride + e];
  sums[cls * stride + e] += acc;
}

}  // namespace mmhn
