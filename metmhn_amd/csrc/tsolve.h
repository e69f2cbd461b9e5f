// Tile-level triangular solves  (D - Q) y = rhs  (TR: transposed system) on vectors in index order.
//
// (D - Q) is lower triangular in index order: every transition sets bits, so y[x] only needs y on subsets of x
// (supersets for the transpose).  Instead of the reference's k+1 Jacobi sweeps (likelihood.py:253-261,
// vanilla.py:289-303) each state is computed exactly once.  A vector is cut into tiles of 2^TB contiguous states; a tile
// H depends on the tiles H ^ (tile bits of a move) - a partial order whose levels are popcount(H).  One workgroup solves one
// tile (tsolve_tile):
//   * step A streams the out-of-tile neighbours (coalesced global reads) into an accumulator,
//   * step B solves the 2^t states of the tile in LDS in popcount order (perm = states sorted by popcount, one barrier per
//     level): z = acc + sum rate * y[x ^ move], y = lidg * z,
//   * step C writes the tile back.
// Same result as the Jacobi iteration up to rounding (Q_off is nilpotent), 1/(k+1) of the arithmetic, ~1/10 of the traffic.
//
// Two schedules over the same tile body:
//   k_tsolve  one launch per level, the host walks the levels (API calls; MMHN_COOP=0).
//   k_csolve  ONE launch for all tiles of all problems of a list (round 5: the path of every heterogeneous cohort - several
//             workgroups per patient).  The tiles sit in a work list in a topological order; workgroups pull the next item
//             from a device-side queue head, set the tile up (tables, diagonal, right-hand side: nothing of that depends on
//             other tiles), wait for the flags of exactly the tiles step A is going to read, and publish their own flag
//             when the tile is stored.  Hand-off as MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup
//             visibility" / cdna_hip_programming.md Guideline 16, form R1: payload stored write-through (sc1), every storing
//             wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane stores the flag (relaxed, agent scope); the
//             consumer polls with relaxed agent-scope loads from one wave, then ONE agent-scope acquire, that wave's
//             vmcnt(0), a workgroup barrier, then plain loads.  Nothing depends on dispatch order or placement: an item's
//             dependencies precede it in the list, so whoever pulled them is resident and finishes without waiting for
//             anything later.  Every spin is bounded; a timeout sets an abort word (device + pinned host) that ends all other
//             spins and makes the host call fail.
#pragma once
#include "kernels.h"

namespace mmhn {

// work list of k_csolve (one per direction: the forward list is a topological order of the forward system, the transposed
// list one of the transposed system)
struct CItem {
  int prob;          // problem (index into the descriptor list)
  uint32_t H;        // tile
  int dep0, ndep;    // its dependencies: items deps[dep0 .. dep0 + ndep) of the same list (all of smaller index)
};
// device words shared by every cooperative launch of one engine
struct CoopCtl {
  unsigned head[8];  // queue heads (one per launch slot); the last workgroup to leave a launch puts its head back to 0
  unsigned abort;    // != 0: a spin timed out
  unsigned pad[7];
};
#ifndef MMHN_TS_WPE
#define MMHN_TS_WPE TSB_WPE      // waves per SIMD the tile solvers' register budget is sized for (8: two workgroups per CU, 4: one)
#endif
#ifndef MMHN_CS_WPE
#define MMHN_CS_WPE MMHN_TS_WPE
#endif
#ifndef MMHN_TS_APIPE
#define MMHN_TS_APIPE 0          // 1: step A as a software pipeline over the moves (next neighbour tile requested before the current terms)
#endif
#ifndef MMHN_TS_TRIP
#define MMHN_TS_TRIP 3           // in-tile moves of a state whose LDS loads are issued together (step B)
#endif
constexpr int TS_WPE = MMHN_TS_WPE;
constexpr int CS_WPE = MMHN_CS_WPE;
constexpr int CS_WG_PER_CU = CS_WPE * 256 / TSB;
#ifndef MMHN_CS_SLEEP
#define MMHN_CS_SLEEP 16         // s_sleep argument between two polls of a wait (units of 64 cycles)
#endif
constexpr unsigned COOP_SPIN_LIMIT = 1u << 21;   // polls of one wait before it gives up (seconds; a healthy wait is microseconds)

template <typename T>
__device__ __forceinline__ void store_wt(T* p, T v) {          // write-through store (sc1): leaves the XCD's L2 at once
  if constexpr (sizeof(T) == 8)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct NoWait { __device__ __forceinline__ void operator()() const {} };

// LIDGV: 1/(D - diag Q) comes from the vector `lidg` (API path, single-tumour spaces);
// otherwise from the class tables of k_prep (joint spaces with seeding: engine path).
// WT: the tile is stored write-through (another workgroup of the same launch reads it).
// `wait` runs between the set-up and the first read of another tile.
template <typename T, bool TR, bool LIDGV, bool WT, typename Wait>
__device__ __forceinline__ void tsolve_tile(unsigned char* smem, const Desc* __restrict__ descs, int prob, uint32_t H,
                                            T* y, const T* __restrict__ lidg, const T* __restrict__ rhs, int rhs_mode,
                                            const T* __restrict__ scal, const uint16_t* __restrict__ perm, int maxk,
                                            const T* __restrict__ tab, const JLink<T>* __restrict__ links,
                                            const T* __restrict__ qS, Wait&& wait) {
  Desc& d = *reinterpret_cast<Desc*>(smem);
  T* yt = reinterpret_cast<T*>(smem + DESC_PAD);
  T* Ltab = yt + (1 << TB);
  T* Utab = Ltab + maxk * 64;
  // (an opaque copy of the thread id: inside k_csolve's persistent loop hipcc otherwise hoists every thread-derived index of the
  // tile body out of the loop and spills it - 104 B of scratch per lane)
  int tid_ = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_;
  STAMP_DECL;                    // (-DMMHN_STAMPS: scripts/tile_stamps.py)
  STAMP_START;
  load_desc(&d, descs + prob);
  __syncthreads();
  const int k = d.k;
  const int t = k < TB ? k : TB;
  const uint32_t nelem = 1u << t, tmask = nelem - 1;
  const long long base = d.off;
  const uint32_t xhi = H << t;
  const bool joint = d.mode == JOINT;
  constexpr int NJ = (1 << TB) / TSB;        // states per thread
  constexpr int NW = TSB / 64;
  // ---- step-B operands of this thread's states (perm order), fetched first so that their
  // latency hides behind the table load and step A: state index and 1/(D - diag Q)
  const uint16_t* pm = perm + (size_t)t * (1 << TB);
  uint32_t px[NJ];
  T lid[NJ];
  T rhs3[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) rhs3[j] = 0;
  if (LIDGV) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint32_t idx = (uint32_t)tid + TSB * j;
      px[j] = idx < nelem ? pm[idx] : 0u;
      lid[j] = idx < nelem ? lidg[base + xhi + px[j]] : T(0);
    }
  } else {
    const T* dP = tab + d.toff + rate_table_size(k);
    const T* dM = dP + (1ll << __popc(d.maskP));
    const T* dE = dM + (1ll << __popc(d.maskM));
    const uint32_t cP = d.maskP & tmask, cM = d.maskM & tmask, cE = d.pairP & tmask;
    // pext of the 12 tile bits through two 64-entry tables per mask (low 6 / high 6 bits of xl);
    // entries 384..386: compact index of the tile's high class bits (tile-uniform, computed once)
    uint32_t* pxt = reinterpret_cast<uint32_t*>(Utab);         // Utab is filled later by tile_tables
    if (tid < 384) {
      const int which = tid >> 7, half = (tid >> 6) & 1, v = tid & 63;
      const uint32_t m = which == 0 ? cP : which == 1 ? cM : cE;
      const uint32_t part = half == 0 ? pext32((uint32_t)v, m & 63u)
                                      : (pext32((uint32_t)v << 6, m & ~63u) << __popc(m & 63u));
      pxt[tid] = part;
    } else if (tid < 387) {
      const uint32_t m = tid == 384 ? d.maskP : tid == 385 ? d.maskM : d.pairP;
      pxt[tid] = pext32(xhi, m & ~tmask) << __popc(m & tmask);
    }
    __syncthreads();
    const uint32_t hP = pxt[384], hM = pxt[385], hE = pxt[386];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint32_t idx = (uint32_t)tid + TSB * j;
      px[j] = idx < nelem ? pm[idx] : 0u;
      const uint32_t xl = px[j], x = xhi | xl;
      const uint32_t lo = xl & 63u, hi6 = xl >> 6;
      T v = 1;
      if (idx < nelem) {
        if ((x >> d.seedbit) & 1u) v = T(1) / (dP[hP | pxt[lo] | pxt[64 + hi6]] + dM[hM | pxt[128 + lo] | pxt[192 + hi6]]);
        else if (eq_noseed(d, x)) v = T(1) / dE[hE | pxt[256 + lo] | pxt[320 + hi6]];
      }
      lid[j] = v;      // seed = 0 states with PT != MT: no rates and zero right-hand side, y stays 0
    }
    if (rhs_mode == 3) {
      // right-hand side of the joint adjoint, formed on the fly: only the compatible states (all bits of the
      // observed tumour + seeding set) are non-zero and take D * q_marginal[pext(x, other tumour's bits)]
      const JLink<T> L = links[prob];
      const bool seed_hi = (xhi >> d.seedbit) & 1u;
      const bool can0 = L.soff[0] >= 0 && (d.seedbit < t || seed_hi) && ((xhi & d.maskP & ~tmask) == (d.maskP & ~tmask));
      const bool can1 = L.soff[1] >= 0 && (d.seedbit < t || seed_hi) && ((xhi & d.maskM & ~tmask) == (d.maskM & ~tmask));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const uint32_t xl = ((uint32_t)((tid >> 6) + NW * j) << 6) | (uint32_t)(tid & 63);
        const uint32_t x = xhi | xl;
        T rv = 0;
        if (xl < nelem && ((x >> d.seedbit) & 1u)) {
          const uint32_t lo = xl & 63u, hi6 = xl >> 6;
          if (can0 && (xl & cP) == cP) rv += L.cst[0] * qS[L.soff[0] + (1ll << (L.sk[0] - 1)) + ((hM | pxt[128 + lo] | pxt[192 + hi6]))];
          if (can1 && (xl & cM) == cM) rv += L.cst[1] * qS[L.soff[1] + (1ll << (L.sk[1] - 1)) + ((hP | pxt[lo] | pxt[64 + hi6]))];
        }
        rhs3[j] = rv;
      }
    }
    __syncthreads();   // pxt lives in the Utab area: done before tile_tables overwrites it
  }
  STAMP(0);
  tile_tables(d, tab, H, Ltab, Utab, yt);
  STAMP(1);

  const int wave = tid >> 6, lane = tid & 63;
  const uint32_t last = (1u << k) - 1u;

  // ---- step A: right-hand side + transitions that cross the tile boundary
  T acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const uint32_t xl = ((uint32_t)(wave + NW * j) << 6) | (uint32_t)lane;
    T rv = 0;
    if (xl < nelem) {
      const uint32_t x = xhi | xl;
      if (rhs_mode == 0) rv = rhs[base + x];
      else if (rhs_mode == 1) rv = (x == last) ? scal[prob] : T(0);
      else if (rhs_mode == 2) rv = (x == 0) ? e0_scale<T>() : T(0);
      else rv = rhs3[j];
    }
    acc[j] = rv;
  }
  STAMP(2);
  wait();                                                      // (k_csolve: the tiles read below are complete and visible)
  STAMP(3);
#if MMHN_TS_APIPE
  // software pipeline over the moves: the neighbour tile of move i + 1 is requested before the terms of move i are formed
  {
    uint32_t p_mv = 0; int p_b = 0, p_kind = 0; bool p_seed = false;
    T p_nv[NJ];
    auto consume = [&]() {
      const uint32_t ml = p_mv & tmask;
      const T Lb = Ltab[p_b * 64 + lane];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = wave + NW * j;
        const uint32_t xl = ((uint32_t)r << 6) | (uint32_t)lane;
        const uint32_t x = xhi | xl;
        const bool ss = seed_set(d, x);
        bool cond = xl < nelem && (TR ? (xl & ml) == 0 : (xl & ml) == ml);
        if (p_kind == 1) cond = cond && !ss && eq_noseed(d, x);
        else if (p_seed) cond = cond && eq_noseed(d, x);
        else cond = cond && ss;
        const T term = Lb * Utab[p_b * 64 + (r & 63)] * p_nv[j];
        acc[j] += cond ? term : T(0);
      }
    };
    for (int b = (t > 0 ? t - 1 : 0); b < k; ++b) {
      const int c = d.cls[b];
      const bool is_seed = joint && c == CS;
      const bool is_pair = joint && ((d.pairP >> b) & 1u);
      for (int kind = 0; kind < 2; ++kind) {
        if (kind == 1 && !is_pair) continue;
        const uint32_t mv = kind == 0 ? (1u << b) : (3u << b);
        const uint32_t mh = mv >> t;
        if (mh == 0) continue;
        if (TR ? (H & mh) != 0 : (H & mh) != mh) continue;
        T c_nv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const uint32_t xl = ((uint32_t)(wave + NW * j) << 6) | (uint32_t)lane;
          c_nv[j] = xl < nelem ? y[base + ((xhi | xl) ^ mv)] : T(0);
        }
        if (p_mv) consume();
        p_mv = mv; p_b = b; p_kind = kind; p_seed = is_seed;
#pragma unroll
        for (int j = 0; j < NJ; ++j) p_nv[j] = c_nv[j];
      }
    }
    if (p_mv) consume();
  }
#else
  for (int b = (t > 0 ? t - 1 : 0); b < k; ++b) {
    const int c = d.cls[b];
    const bool is_seed = joint && c == CS;
    const bool is_pair = joint && ((d.pairP >> b) & 1u);
    // candidate moves of bit b: single bit (async / seeding) and, for a paired P bit, both bits
    // (the host's dependency lists follow exactly these conditions: Engine::tile_deps)
    for (int kind = 0; kind < 2; ++kind) {
      if (kind == 1 && !is_pair) continue;
      const uint32_t mv = kind == 0 ? (1u << b) : (3u << b);
      const uint32_t mh = mv >> t, ml = mv & tmask;
      if (mh == 0) continue;                                   // stays inside the tile: step B
      if (TR ? (H & mh) != 0 : (H & mh) != mh) continue;       // tile-uniform part of the condition
      const T Lb = Ltab[b * 64 + lane];
      T nv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {                           // all neighbour loads in flight together
        const uint32_t xl = ((uint32_t)(wave + NW * j) << 6) | (uint32_t)lane;
        nv[j] = xl < nelem ? y[base + ((xhi | xl) ^ mv)] : T(0);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = wave + NW * j;
        const uint32_t xl = ((uint32_t)r << 6) | (uint32_t)lane;
        const uint32_t x = xhi | xl;
        const bool ss = seed_set(d, x);
        bool cond = xl < nelem && (TR ? (xl & ml) == 0 : (xl & ml) == ml);
        if (kind == 1) cond = cond && !ss && eq_noseed(d, x);
        else if (is_seed) cond = cond && eq_noseed(d, x);
        else cond = cond && ss;
        const T term = Lb * Utab[b * 64 + (r & 63)] * nv[j];
        acc[j] += cond ? term : T(0);
      }
    }
  }
#endif
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const uint32_t xl = ((uint32_t)(wave + NW * j) << 6) | (uint32_t)lane;
    if (xl < nelem) yt[xl] = acc[j];
  }
  __syncthreads();
  STAMP(4);

  // ---- step B: popcount-ordered substitution inside the tile (a state's level is its popcount)
  const uint32_t pairP = joint ? d.pairP : 0u;
  const uint32_t lone = d.lone;
  const int seedb = joint ? d.seedbit : -1;
  // fast tiles: every in-tile bit is a plain single-bit move for every state (single-tumour spaces, and
  // joint tiles whose seeding bit lies above the tile and is set) - no per-bit condition logic at all
  const bool fast = !joint || (seedb >= t && ((xhi >> seedb) & 1u));
  int plev[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) plev[j] = ((uint32_t)tid + TSB * j) < nelem ? __popc(px[j]) : -1;
  for (int s = 0; s <= t; ++s) {
    const int level = TR ? t - s : s;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (plev[j] != level) continue;
      const uint32_t xl = px[j];
      const uint32_t lo = xl & 63u, ro = xl >> 6;
      T z = yt[xl];
      uint32_t todo = TR ? (~xl & tmask) : xl;
      if (fast) {
        while (todo) {                         // MMHN_TS_TRIP bits per trip: their LDS loads are issued together
          T r[MMHN_TS_TRIP];
#pragma unroll
          for (int u = 0; u < MMHN_TS_TRIP; ++u) {
            const bool on = todo != 0;
            const int b = on ? __ffs(todo) - 1 : 0;
            todo &= todo - 1;                  // 0 stays 0
            const T v = Ltab[b * 64 + lo] * Utab[b * 64 + ro] * yt[xl ^ (1u << b)];
            r[u] = on ? v : T(0);
          }
#pragma unroll
          for (int u = 0; u < MMHN_TS_TRIP; ++u) z += r[u];
        }
      } else {
        const uint32_t x = xhi | xl;
        const bool ss = seedb >= 0 && ((x >> seedb) & 1u);
        const bool e0x = ((x & lone) == 0) && (((x & pairP) << 1) == (x & (pairP << 1)));
        while (todo) {
          const int b = __ffs(todo) - 1;
          todo &= todo - 1;
          uint32_t mv = 1u << b;
          bool cond;
          if (b == seedb) cond = e0x;
          else if (ss) cond = true;
          else if ((pairP >> b) & 1u) {
            mv = 3u << b;
            cond = (b + 1 < t) && e0x && (TR ? (xl & mv) == 0 : (xl & mv) == mv);
          } else cond = false;
          if (cond) z += Ltab[b * 64 + lo] * Utab[b * 64 + ro] * yt[(xl ^ mv) & tmask];
        }
      }
      yt[xl] = lid[j] * z;
    }
    __syncthreads();
  }

  STAMP(5);
  // ---- step C
  if constexpr (WT) {
    for (uint32_t e = tid; e < nelem; e += TSB) store_wt(y + base + xhi + e, yt[e]);
  } else {
    for (uint32_t e = tid; e < nelem; e += TSB) y[base + xhi + e] = yt[e];
  }
#ifdef MMHN_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  STAMP(6);
  if (st_on) st_sum[7] += 1;     // tiles
  if (t == TB) STAMP_FLUSH(TR ? 8 : 0);       // (the full tiles of the large problems only)
#endif
}

// one launch per level of tile-index popcount (host: Engine::solve)
template <typename T, bool TR, bool LIDGV>
__global__ __launch_bounds__(TSB, TS_WPE) void k_tsolve(const Desc* __restrict__ descs,
                                                const int2* __restrict__ lmap,
                                                const Params<T>* __restrict__ par, T* y,
                                                const T* __restrict__ lidg,
                                                const T* __restrict__ rhs, int rhs_mode,
                                                const T* __restrict__ scal,
                                                const uint16_t* __restrict__ perm,
                                                const int* __restrict__ lvl, int maxk,
                                                const T* __restrict__ tab,
                                                const JLink<T>* __restrict__ links,
                                                const T* __restrict__ qS) {
  extern __shared__ __align__(16) unsigned char smem[];
  (void)par; (void)lvl;
  const uint32_t blk = xcd_chunked(blockIdx.x, gridDim.x);
  tsolve_tile<T, TR, LIDGV, false>(smem, descs, lmap[blk].x, (uint32_t)lmap[blk].y, y, lidg, rhs, rhs_mode, scal, perm, maxk, tab,
                                   links, qS, NoWait{});
}

// ------------------------------------------------------------------------------------
// k_csolve: every tile of every problem of a list in ONE launch (see the head of this file).
//   items / deps: the work list of this direction;  flags[i] == epoch: item i of this launch is stored and visible;
//   ctl->head[slot]: queue head;  h_abort: the pinned host copy of ctl->abort;  fault: test hook (see the publish step).
// ------------------------------------------------------------------------------------
// WPE: waves per SIMD the registers are sized for.  8 (64 registers; 80 - 104 B per lane of scratch: the persistent loop keeps the
// kernel arguments live) lets a CU hold this workgroup AND a 1 024-thread workgroup of another stream - the small-space launches and
// the staged own-problem chain that run next to the joint solves of a heterogeneous cohort (28-event LUAD cohort: 1.41 against 1.56 ms);
// 4 (93 registers, no scratch) is faster when nothing runs beside it (LUAD-reduced cohort: 0.339 against 0.350 ms).  The engine picks
// per batch (Engine::solve).
template <typename T, bool TR, bool LIDGV, int WPE = CS_WPE>
__global__ __launch_bounds__(TSB, WPE) void k_csolve(const Desc* __restrict__ descs,
                                                const CItem* __restrict__ items, const int* __restrict__ deps, int nitems,
                                                unsigned* flags, unsigned epoch, CoopCtl* ctl, int slot, unsigned* h_abort, int fault,
                                                T* y, const T* __restrict__ lidg,
                                                const T* __restrict__ rhs, int rhs_mode,
                                                const T* __restrict__ scal,
                                                const uint16_t* __restrict__ perm, int maxk,
                                                const T* __restrict__ tab,
                                                const JLink<T>* __restrict__ links,
                                                const T* __restrict__ qS) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ unsigned s_item;
  const int tid = threadIdx.x;
  unsigned* head = &ctl->head[slot];
  for (;;) {
    __syncthreads();                                           // the previous tile's LDS (and s_item) are done with
    if (tid == 0) s_item = atomicAdd(head, 1u);
    __syncthreads();
    const unsigned it = s_item;
    if (it >= (unsigned)nitems) {
      // every workgroup fetches exactly once beyond the end: the one that draws the last ticket puts the head back
      if (tid == 0 && it == (unsigned)nitems + gridDim.x - 1u) __hip_atomic_store(head, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    const CItem ci = items[it];
    auto wait = [&]() {
      if (ci.ndep == 0) return;                                // (uniform)
      if (tid < 64) {
        // one wave polls: lane i its i-th dependency, only until it has seen it done; the abort word every eighth round;
        // between two rounds the wave sleeps (a few hundred workgroups poll at the same time: the polls are memory traffic
        // of their own - MI355X_MICROARCH.md "polling-cost")
        const int lane = tid;
        const unsigned* f = flags + deps[ci.dep0 + (lane < ci.ndep ? lane : 0)];
        bool done = lane >= ci.ndep;
        unsigned spins = 0;
        for (;;) {
          if (!done) done = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
          if (__all(done)) break;
          if ((++spins & 7u) == 0u) {
            const unsigned ab = __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ab != 0u) break;
            if (spins > COOP_SPIN_LIMIT) {
              if (lane == 0) {
                __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(h_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              }
              break;
            }
          }
          __builtin_amdgcn_s_sleep(MMHN_CS_SLEEP);
        }
#ifndef MMHN_CS_NOACQ   // (timing-only ablation: wrong results)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // ONE acquire after the match: drops this CU's stale lines
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ... and is complete before the barrier lets the loads go
#endif
      }
      __syncthreads();
    };
#ifdef MMHN_CS_PLAIN    // variant: plain stores + ONE agent-scope release (L2 write-back) instead of write-through stores
    tsolve_tile<T, TR, LIDGV, false>(smem, descs, ci.prob, ci.H, y, lidg, rhs, rhs_mode, scal, perm, maxk, tab, links, qS, wait);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(flags + it, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#else
    tsolve_tile<T, TR, LIDGV, true>(smem, descs, ci.prob, ci.H, y, lidg, rhs, rhs_mode, scal, perm, maxk, tab, links, qS, wait);
    // publish: every storing wave drains its write-through stores, the workgroup meets, one lane raises the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (fault != 0 - MMHN_COOP_FAULT=1, tests only: the first item never raises its flag, so that its dependants run into the
    // bound of their spin and the abort path is exercised)
    if (tid == 0 && !(fault && it == 0u)) __hip_atomic_store(flags + it, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  }
}

}  // namespace mmhn
