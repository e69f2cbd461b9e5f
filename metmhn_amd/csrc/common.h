// Shared pieces of the gfx950 kernels of the metMHN hot path (closed-form "gather" formulation): tile constants, the
// per-problem tables of an evaluation (k_prep, tile_tables), small device helpers.  kernels.h includes every kernel family.
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

template <typename T>
inline size_t prep_lds(int N) { return (size_t)(N + 1) * 192 * sizeof(T); }
// PB: threads of a workgroup (1 024 on the long launches: the kernel is a chain of short phases between barriers - waves to hide
// them behind, at the same LDS per workgroup; the SPLIT launches of short cohorts keep 256)
template <typename T, bool SPLIT = false, int PB = BLOCK>
__global__ __launch_bounds__(PB) void k_prep(const Desc* __restrict__ descs,
                                                const Params<T>* __restrict__ par, T* tab, const int* __restrict__ plist = nullptr) {
  __shared__ T thc[(MAXN + 1) * MAXN];  // later reused as th[i][class bit l]
  // [i][three 6-bit parts of S] partial rate products (row N: observation): (N + 1) * 192 elements of dynamic LDS (prep_lds) - sized
  // by the engine's N, not by MAXN: 42 instead of 59 KB at N = 21, three workgroups per CU instead of two
  extern __shared__ __align__(16) unsigned char prep_smem[];
  T* const rsplit = reinterpret_cast<T*>(prep_smem);
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
  for (int e = tid; e < k * k; e += PB) {
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
  for (int e = tid; e < k * 64; e += PB) {
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
    if ((long long)part * PB >= (1ll << kc)) return;        // (uniform: nothing of this table falls to this part)
    __syncthreads();
    // th[i][l] = theta[i][event of the l-th class bit]
    for (int e = tid; e < N * kc; e += PB) {
      const int i = e / kc, l = e % kc;
      uint32_t m = cm;
      for (int q = 0; q < l; ++q) m &= m - 1;
      thc[e] = P.th[i][d.ev[__ffs(m) - 1]];
    }
    __syncthreads();
    const T* dv = c == 1 ? P.dm : P.dp;
    // row N of the table: the observation factors dvec[event of bit l]
    for (int l = tid; l < kc; l += PB) {
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
      for (int e = tid; e < (N + 1) * 192; e += PB) {
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
    for (long long S = tid + (long long)part * PB; S < (1ll << kc); S += (long long)PB * nparts) {
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

}  // namespace mmhn
