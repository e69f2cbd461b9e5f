// Joint <-> marginal transfers, adjoint seeds, the head of the staged single-tumour kernels.  Reference: likelihood.py:540-620.
#pragma once
#include "common.h"

namespace mmhn {

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
// KB: threads of a workgroup (1 024 on long launches: a patient's marginal space of up to 2^16 states is one workgroup's loop)
template <typename T, int KB = BLOCK>
__global__ __launch_bounds__(KB) void k_scatter_marg(const PatRec* __restrict__ pats,
                                                        const Desc* __restrict__ dJ,
                                                        const Desc* __restrict__ dS,
                                                        const Params<T>* __restrict__ par,
                                                        const T* __restrict__ qS,
                                                        const T* __restrict__ rhsS, T* rhsJ,
                                                        T* dots, int part, const int* __restrict__ plist) {
  __shared__ T red[KB];
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
  for (uint32_t e = threadIdx.x; e < half; e += KB) {
    const uint32_t x = pdep32(e, free_) | fixed;
    const T qv = qS[ds.off + half + e];
    if (rhsJ) rhsJ[dj.off + x] += c * qv;
    dot += qv * rhsS[ds.off + half + e];
  }
  red[threadIdx.x] = dot;
  __syncthreads();
  for (int s = KB / 2; s > 0; s >>= 1) {
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

}  // namespace mmhn
