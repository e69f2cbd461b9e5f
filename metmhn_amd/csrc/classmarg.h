// Class marginals of p (x) q for vectors in index order (k_class_marg per tile, k_pclass per problem / work item) and the eq
// block's flows.  Reference: likelihood.py:25-201 (x_partial_Q_y) in the class-marginal form of DESIGN.md 3.2.
#pragma once
#include "common.h"

namespace mmhn {

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

}  // namespace mmhn
