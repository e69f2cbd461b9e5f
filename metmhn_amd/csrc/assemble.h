// Per-patient assembly and the deterministic cohort reduction.  Reference: likelihood.py:441-512, 623-731;
// regularized_optimization.py:256-266.
#pragma once
#include "common.h"
#include "gradrows.h"

namespace mmhn {

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
