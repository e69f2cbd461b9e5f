// Flows over a subset lattice -> rows of a G matrix (k_grad_rows), weighted bit marginals (k_bit_marg).
// Reference: likelihood.py:163-228, vanilla.py:190-203, 328-393.
#pragma once
#include "common.h"

namespace mmhn {

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

// value of lane `src` (a compile-time or wave-uniform index) in every lane
__device__ __forceinline__ double lane_bcast(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// grid = (work list of (problem, subset chunk), ceil(N / WAVES)); wave w owns event i = blockIdx.y * WAVES + w
// and strides the subsets S across its lanes; all reductions are wave-level.
constexpr int GR_CHUNK = 11;                      // subsets per workgroup of k_grad_rows: 2^11

// GW: waves (= rows) of a workgroup: 4, or 1 on long launches (engine.hip: launch_grad_rows)
template <typename T, int GW = WAVES>
__global__ __launch_bounds__(GW * 64) void k_grad_rows(const Desc* __restrict__ descs,
                                                     const Params<T>* __restrict__ par,
                                                     const T* __restrict__ A,
                                                     const T* __restrict__ p,
                                                     const T* __restrict__ q, T* G, int kind_arg,
                                                     T* DJ, const int2* __restrict__ chunks,
                                                     int nprob, long long gstride) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* Tlo = reinterpret_cast<T*>(smem);          // [WAVES][3][64]: rate products over subset bits 0-5, 6-11, 12-17
  T* rowbuf = Tlo + GW * 192;                   // [GW][32]
  __shared__ int lev[32];                       // event of local bit l
  // (problem, subset chunk) work list; kind_arg < 0: the kind rides in bits 24+ of the chunk field and selects the G matrix
  const int prob = chunks[blockIdx.x].x;
  const int kind = kind_arg < 0 ? chunks[blockIdx.x].y >> 24 : kind_arg;
  const int chunk = chunks[blockIdx.x].y & 0xffffff;
  if (kind_arg < 0) G += (long long)kind * gstride;
  const Desc& d = descs[prob];
  const int N = d.N, n = N - 1;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int i = blockIdx.y * GW + w;
  const Params<T>& P = par[d.pset];

  uint32_t cm;
  if (kind == GK_P) cm = d.maskP; else if (kind == GK_M) cm = d.maskM;
  else if (kind == GK_E) cm = d.pairP; else cm = (1u << d.k) - 1u;
  const int kc = __popc(cm);
  if (tid < kc) {                               // (lane l: the l-th bit of the class - one global read each, side by side)
    uint32_t m = cm;
    for (int q = 0; q < tid; ++q) m &= m - 1;
    lev[tid] = d.ev[__ffs(m) - 1];
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
    // (the factor of class bit l sits in lane l: one global read per lane, the products below take them out of registers)
    const T fl = lane < kc ? fvec[lev[lane < kc ? lane : 0]] : T(1);
#pragma unroll
    for (int part = 0; part < 3; ++part) {          // one table per 6-bit part of the subset index
      T v = 1;
#pragma unroll
      for (int l = 0; l < 6; ++l) {
        const int ll = part * 6 + l;
        const T f = lane_bcast(fl, ll);
        if (ll < kc && ((lane >> l) & 1)) v *= f;
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

}  // namespace mmhn
