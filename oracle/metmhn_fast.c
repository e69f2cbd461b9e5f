/*
 * Optimised CPU variant of the metMHN hot path for PAIRED datapoints (type 3) - CPU BASELINE and checker only
 * (same rules as metmhn_ref.c: only tests/ and the cpu_baseline leg of bench.py load it).
 *
 * Where metmhn_ref.c keeps the reference's pass structure (one pass over the 2^k vector per Kronecker factor,
 * k+1 Jacobi sweeps per solve), this file is the formulation the GPU engine uses, written for one CPU core per
 * patient (SURVEY 8d asks for it "so the speed-up is not flattered by a deliberately slow baseline"):
 *   - closed form of the restricted generator, rates from per-bit product tables split
 *     over the low / high 10 index bits;
 *   - (D - Q)^-1 by one substitution pass in index order (Q_off is strictly lower triangular), transposed
 *     solve in descending order;
 *   - diagonal of D - Q from the Kronecker-sum tables dP[x_P] + dM[x_M] (seed = 1) and dE (PT == MT states);
 *   - theta / observation-rate gradients from the class marginals of p (x) q.
 * The per-patient pipeline is metmhn/jx/likelihood.py:516-731 (adjoint method) as restated in
 * oracle/closed_form.py:346-440, which this file follows function by function.
 * Validated against metmhn_ref.c (tests/test_oracle_golden.py).  OpenMP over patients.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { CP = 0, CM = 1, CS = 2 };
#define LOB 10 /* index bits covered by the "low" product tables */

static inline int popc(uint32_t v) { return __builtin_popcount(v); }
static inline uint32_t pext32(uint32_t x, uint32_t m) {
  uint32_t o = 0, p = 0;
  while (m) { const uint32_t l = m & (0u - m); if (x & l) o |= 1u << p; ++p; m ^= l; }
  return o;
}
static inline uint32_t pdep32(uint32_t v, uint32_t m) {
  uint32_t o = 0;
  while (m) { const uint32_t l = m & (0u - m); if (v & 1u) o |= l; v >>= 1; m ^= l; }
  return o;
}

/* ---- single-tumour space: events ev[0..k-1] (all one class), D = identity (vanilla.py:206-305) ---------- */
typedef struct {
  int k, N;
  int ev[32];
  const double* th; /* N x N, multiplicative */
  double* prod;     /* [N][2^k]: prod_{bits of S} th[i][ev(bit)] */
  double* diag;     /* [2^k]: 1 + total outflow */
} sspace;

static void ss_init(sspace* s, int N, const double* th, const int* ev, int k) {
  s->k = k; s->N = N; s->th = th;
  memcpy(s->ev, ev, sizeof(int) * k);
  const size_t V = (size_t)1 << k;
  s->prod = (double*)malloc(sizeof(double) * N * V);
  s->diag = (double*)malloc(sizeof(double) * V);
  for (int i = 0; i < N; ++i) {
    double* p = s->prod + (size_t)i * V;
    p[0] = 1.0;
    for (size_t S = 1; S < V; ++S) {
      const int b = __builtin_ctzl(S);
      p[S] = p[S & (S - 1)] * th[i * N + ev[b]];
    }
  }
  int bit_of[64];
  for (int i = 0; i < N; ++i) bit_of[i] = -1;
  for (int b = 0; b < k; ++b) bit_of[ev[b]] = b;
  for (size_t S = 0; S < V; ++S) {
    double d = 1.0;
    for (int i = 0; i < N; ++i)
      if (bit_of[i] < 0 || !((S >> bit_of[i]) & 1)) d += th[i * N + i] * s->prod[(size_t)i * V + S];
    s->diag[S] = d;
  }
}
static void ss_free(sspace* s) { free(s->prod); free(s->diag); }
/* y = (I - Q)^-1 rhs   (tr: transposed) */
static void ss_solve(const sspace* s, const double* rhs, double* y, int tr) {
  const size_t V = (size_t)1 << s->k;
  const int N = s->N;
  if (!tr) {
    for (size_t S = 0; S < V; ++S) {
      double z = rhs[S];
      for (size_t m = S; m; m &= m - 1) {
        const int b = __builtin_ctzl(m);
        const int i = s->ev[b];
        const size_t src = S ^ ((size_t)1 << b);
        z += s->th[i * N + i] * s->prod[(size_t)i * V + src] * y[src];
      }
      y[S] = z / s->diag[S];
    }
  } else {
    for (size_t S = V; S-- > 0;) {
      double z = rhs[S];
      for (size_t m = ~S & (V - 1); m; m &= m - 1) {
        const int b = __builtin_ctzl(m);
        const int i = s->ev[b];
        z += s->th[i * N + i] * s->prod[(size_t)i * V + S] * y[S | ((size_t)1 << b)];
      }
      y[S] = z / s->diag[S];
    }
  }
}
/* val[i][j] = q^T dQ/dlog th_ij p, ddiag[j] = -sum_{i != j} val[i][j]   (vanilla.py:328-393 in flow form) */
static void ss_xQy(const sspace* s, const double* q, const double* p, double* val, double* ddiag) {
  const size_t V = (size_t)1 << s->k;
  const int N = s->N, k = s->k;
  int bit_of[64];
  for (int i = 0; i < N; ++i) bit_of[i] = -1;
  for (int b = 0; b < k; ++b) bit_of[s->ev[b]] = b;
  memset(val, 0, sizeof(double) * N * N);
  for (int i = 0; i < N; ++i) {
    const int l = bit_of[i];
    const double* pr = s->prod + (size_t)i * V;
    const double base = s->th[i * N + i];
    double tot = 0, mar[32];
    for (int b = 0; b < k; ++b) mar[b] = 0;
    for (size_t S = 0; S < V; ++S) {
      double f;
      if (l >= 0) {
        if ((S >> l) & 1) continue;
        f = base * pr[S] * (p[S] * q[S | ((size_t)1 << l)] - p[S] * q[S]);
      } else {
        f = -base * pr[S] * p[S] * q[S];
      }
      tot += f;
      for (size_t m = S; m; m &= m - 1) mar[__builtin_ctzl(m)] += f;
    }
    val[i * N + i] = tot;
    for (int b = 0; b < k; ++b) if (s->ev[b] != i) val[i * N + s->ev[b]] = mar[b];
  }
  for (int j = 0; j < N; ++j) {
    double c = 0;
    for (int i = 0; i < N; ++i) if (i != j) c += val[i * N + j];
    ddiag[j] = -c;
  }
}

/* ---- joint space of a paired patient ------------------------------------------------------------------- */
typedef struct {
  int k, N, n, seedbit, kP, kM, ke;
  int ev[32], cls[32];
  uint32_t maskP, maskM, pairP, lone;
  double base[32];
  double *Rlo, *Rhi;      /* [k][2^LOB], [k][2^(k-LOB)] : rate_b(x) = base[b] * Rlo[b][x & lo] * Rhi[b][x >> LOB] */
  double *dP, *dM, *dE;   /* diagonal tables */
  double *DpT, *DmT;      /* D_p / D_m on the seed = 1 half by class setting */
  uint32_t *xP_lo, *xP_hi, *xM_lo, *xM_hi;   /* pext(x, maskP / maskM) = lo[x & lo mask] | hi[x >> LOB] */
  int khi;
} jspace;

static inline int is_eq(const jspace* J, uint32_t x) {    /* x without the seeding bit */
  return ((x & J->lone) == 0) && (((x & J->pairP) << 1) == (x & (J->pairP << 1)));
}
static inline double jrate(const jspace* J, int b, uint32_t x) {
  return J->base[b] * J->Rlo[((size_t)b << LOB) + (x & ((1u << LOB) - 1u))] * J->Rhi[((size_t)b << J->khi) + (x >> LOB)];
}

static void js_init(jspace* J, const int8_t* st, int n, const double* th, const double* dp, const double* dm) {
  const int N = n + 1;
  memset(J, 0, sizeof *J);
  J->n = n; J->N = N; J->seedbit = -1;
  int k = 0;
  for (int j = 0; j < n; ++j) {
    const int p = st[2 * j] != 0, m = st[2 * j + 1] != 0;
    if (p) { J->ev[k] = j; J->cls[k] = CP; J->maskP |= 1u << k; if (m) J->pairP |= 1u << k; else J->lone |= 1u << k; ++k; }
    if (m) { J->ev[k] = j; J->cls[k] = CM; J->maskM |= 1u << k; if (!p) J->lone |= 1u << k; ++k; }
  }
  if (st[2 * n]) { J->ev[k] = n; J->cls[k] = CS; J->seedbit = k; ++k; }
  J->k = k; J->kP = popc(J->maskP); J->kM = popc(J->maskM); J->ke = popc(J->pairP);
  J->khi = k > LOB ? k - LOB : 0;
  const size_t nlo = (size_t)1 << LOB, nhi = (size_t)1 << J->khi;
  J->Rlo = (double*)malloc(sizeof(double) * (k ? k : 1) * nlo);
  J->Rhi = (double*)malloc(sizeof(double) * (k ? k : 1) * nhi);
  for (int b = 0; b < k; ++b) {
    const int i = J->ev[b], c = J->cls[b], pc = c == CS ? CP : c;
    J->base[b] = c == CM ? th[i * N + i] * th[i * N + n] : th[i * N + i];
    double f[32];
    for (int bb = 0; bb < k; ++bb) f[bb] = (bb != b && J->cls[bb] == pc) ? th[i * N + J->ev[bb]] : 1.0;
    double* lo = J->Rlo + ((size_t)b << LOB);
    double* hi = J->Rhi + ((size_t)b << J->khi);
    lo[0] = 1.0;
    for (size_t v = 1; v < nlo; ++v) { const int bb = __builtin_ctzl(v); lo[v] = lo[v & (v - 1)] * (bb < k ? f[bb] : 1.0); }
    hi[0] = 1.0;
    for (size_t v = 1; v < nhi; ++v) { const int bb = LOB + __builtin_ctzl(v); hi[v] = hi[v & (v - 1)] * f[bb]; }
  }
  {
    const uint32_t lom = (uint32_t)nlo - 1u;
    J->xP_lo = (uint32_t*)malloc(sizeof(uint32_t) * nlo); J->xM_lo = (uint32_t*)malloc(sizeof(uint32_t) * nlo);
    J->xP_hi = (uint32_t*)malloc(sizeof(uint32_t) * nhi); J->xM_hi = (uint32_t*)malloc(sizeof(uint32_t) * nhi);
    for (size_t v = 0; v < nlo; ++v) { J->xP_lo[v] = pext32((uint32_t)v, J->maskP & lom); J->xM_lo[v] = pext32((uint32_t)v, J->maskM & lom); }
    for (size_t v = 0; v < nhi; ++v) {
      J->xP_hi[v] = pext32((uint32_t)v << LOB, J->maskP & ~lom) << popc(J->maskP & lom);
      J->xM_hi[v] = pext32((uint32_t)v << LOB, J->maskM & ~lom) << popc(J->maskM & lom);
    }
  }
  /* diagonal tables (kron_diag in closed form, kronvec.py:713-999; D_p, D_m: :574-602, :646-671) */
  for (int c = 0; c < 3; ++c) {
    const uint32_t cm = c == 0 ? J->maskP : c == 1 ? J->maskM : J->pairP;
    const int kc = popc(cm);
    const size_t V = (size_t)1 << kc;
    double* out = (double*)malloc(sizeof(double) * V);
    double* obs = c < 2 ? (double*)malloc(sizeof(double) * V) : NULL;
    int evl[32], l = 0;
    for (uint32_t m = cm; m; m &= m - 1) evl[l++] = J->ev[__builtin_ctz(m)];
    const double* dv = c == 1 ? dm : dp;
    for (size_t S = 0; S < V; ++S) {
      double o = c == 0 ? dp[n] : c == 1 ? dm[n] : 1.0;
      for (size_t m = S; m; m &= m - 1) o *= dv[evl[__builtin_ctzl(m)]];
      if (obs) obs[S] = o;
      double tot = o;
      const int rows = c == 2 ? N : n;         /* the eq block also carries the seeding rate */
      for (int i = 0; i < rows; ++i) {
        int li = -1;
        for (int q = 0; q < kc; ++q) if (evl[q] == i) li = q;
        if (li >= 0 && ((S >> li) & 1)) continue;
        double r = c == 1 ? th[i * N + i] * th[i * N + n] : th[i * N + i];
        for (size_t m = S; m; m &= m - 1) r *= th[i * N + evl[__builtin_ctzl(m)]];
        tot += r;
      }
      out[S] = tot;
    }
    if (c == 0) { J->dP = out; J->DpT = obs; } else if (c == 1) { J->dM = out; J->DmT = obs; } else J->dE = out;
  }
}
static void js_free(jspace* J) {
  free(J->Rlo); free(J->Rhi); free(J->dP); free(J->dM); free(J->dE); free(J->DpT); free(J->DmT);
  free(J->xP_lo); free(J->xP_hi); free(J->xM_lo); free(J->xM_hi);
}
static inline uint32_t cidxP(const jspace* J, uint32_t x) { return J->xP_lo[x & ((1u << LOB) - 1u)] | J->xP_hi[x >> LOB]; }
static inline uint32_t cidxM(const jspace* J, uint32_t x) { return J->xM_lo[x & ((1u << LOB) - 1u)] | J->xM_hi[x >> LOB]; }

static inline double jdiag(const jspace* J, uint32_t x) {          /* (D_p + D_m - diag Q)(x), x seeded */
  return J->dP[cidxP(J, x)] + J->dM[cidxM(J, x)];
}

/* pi = (D - Q)^-1 e_0 */
static void js_forward(const jspace* J, double* pi) {
  const size_t V = (size_t)1 << J->k;
  const uint32_t sb = J->seedbit >= 0 ? 1u << J->seedbit : 0u;
  for (size_t xi = 0; xi < V; ++xi) {
    const uint32_t x = (uint32_t)xi;
    if (sb && (x & sb)) {
      double z = 0;
      for (uint32_t m = x & ~sb; m; m &= m - 1) {
        const int b = __builtin_ctz(m);
        z += jrate(J, b, x) * pi[x ^ (1u << b)];
      }
      const uint32_t x0 = x ^ sb;
      if (is_eq(J, x0)) z += jrate(J, J->seedbit, x0) * pi[x0];
      pi[x] = z / jdiag(J, x);
    } else if (is_eq(J, x)) {
      double z = x == 0 ? 1.0 : 0.0;
      for (uint32_t m = x & J->pairP; m; m &= m - 1) {
        const int b = __builtin_ctz(m);
        z += jrate(J, b, x) * pi[x ^ (3u << b)];
      }
      pi[x] = z / J->dE[pext32(x, J->pairP)];
    } else {
      pi[x] = 0.0;
    }
  }
}
/* q = (D - Q)^-T rhs, rhs given on the seeded states by a callback-free pair of small vectors:
 *   rhs[x] = c0 * h0[x_M] if all PT bits of x are set  +  c1 * h1[x_P] if all MT bits are set  (x seeded) */
static void js_adjoint(const jspace* J, double c0, const double* h0, double c1, const double* h1, double* q) {
  const size_t V = (size_t)1 << J->k;
  const uint32_t sb = J->seedbit >= 0 ? 1u << J->seedbit : 0u;
  const uint32_t evm = J->maskP | J->maskM;
  for (size_t xi = V; xi-- > 0;) {
    const uint32_t x = (uint32_t)xi;
    if (sb && (x & sb)) {
      double z = 0;
      if (h0 && (x & J->maskP) == J->maskP) z += c0 * h0[cidxM(J, x)];
      if (h1 && (x & J->maskM) == J->maskM) z += c1 * h1[cidxP(J, x)];
      for (uint32_t m = ~x & evm; m; m &= m - 1) {
        const int b = __builtin_ctz(m);
        z += jrate(J, b, x) * q[x | (1u << b)];
      }
      q[x] = z / jdiag(J, x);
    } else if (is_eq(J, x)) {
      double z = 0;
      for (uint32_t m = ~x & J->pairP; m; m &= m - 1) {
        const int b = __builtin_ctz(m);
        z += jrate(J, b, x) * q[x | (3u << b)];
      }
      if (sb) z += jrate(J, J->seedbit, x) * q[x | sb];
      q[x] = z / J->dE[pext32(x, J->pairP)];
    } else {
      q[x] = 0.0;
    }
  }
}

/* flows of every event over one class' subset lattice (closed_form.py:195-223) */
static void grad_accum(const double* th, int N, const double* base, const int* evl, int kc, const double* Adiag,
                       double* const* Abit, double* tot, double* mar /* [N][kc] */) {
  const size_t V = (size_t)1 << kc;
  for (int i = 0; i < N; ++i) {
    tot[i] = 0;
    for (int l = 0; l < kc; ++l) mar[i * 32 + l] = 0;
    if (base[i] == 0.0) continue;
    int li = -1;
    for (int l = 0; l < kc; ++l) if (evl[l] == i) li = l;
    double* rate = (double*)malloc(sizeof(double) * V);
    rate[0] = base[i];
    for (size_t S = 1; S < V; ++S) rate[S] = rate[S & (S - 1)] * th[i * N + evl[__builtin_ctzl(S)]];
    for (size_t S = 0; S < V; ++S) {
      double f;
      if (li >= 0) {
        if ((S >> li) & 1) continue;
        f = rate[S] * (Abit[li][S] + Adiag[S]);
      } else {
        f = rate[S] * Adiag[S];
      }
      tot[i] += f;
      for (size_t m = S; m; m &= m - 1) mar[i * 32 + __builtin_ctzl(m)] += f;
    }
    free(rate);
  }
}

/* G += q^T dQ/dlog theta p ; d_dp, d_dm -= q^T dD/dlog d p   (likelihood.py:163-228 via class marginals) */
static void js_gradient(const jspace* J, const double* th, const double* dp, const double* dm, const double* q,
                        const double* p, double* G, double* ddp, double* ddm) {
  const int N = J->N, n = J->n, k = J->k;
  const uint32_t sb = J->seedbit >= 0 ? 1u << J->seedbit : 0u;
  const size_t V = (size_t)1 << k;
  double* tot = (double*)malloc(sizeof(double) * N);
  double* mar = (double*)malloc(sizeof(double) * N * 32);
  if (sb) {
    for (int c = 0; c < 2; ++c) {
      const uint32_t cm = c == 0 ? J->maskP : J->maskM;
      const int kc = popc(cm);
      const size_t VS = (size_t)1 << kc;
      int bits[32], evl[32], l = 0;
      for (uint32_t m = cm; m; m &= m - 1) { bits[l] = __builtin_ctz(m); evl[l] = J->ev[bits[l]]; ++l; }
      double* W = (double*)calloc(VS, sizeof(double));
      double* Vb[32];
      for (l = 0; l < kc; ++l) Vb[l] = (double*)calloc(VS, sizeof(double));
      for (size_t xi = sb; xi < V; ++xi) {            /* seeded states are the upper half: seeding is the MSB */
        const uint32_t x = (uint32_t)xi;
        if (!(x & sb)) continue;
        const double pv = p[x];
        if (pv == 0.0) continue;
        const uint32_t S = c == 0 ? cidxP(J, x) : cidxM(J, x);
        W[S] -= pv * q[x];
        for (l = 0; l < kc; ++l) if (!((x >> bits[l]) & 1u)) Vb[l][S] += pv * q[x | (1u << bits[l])];
      }
      double base[64];
      for (int i = 0; i < N; ++i) base[i] = i < n ? th[i * N + i] * (c == 1 ? th[i * N + n] : 1.0) : 0.0;
      grad_accum(th, N, base, evl, kc, W, Vb, tot, mar);
      for (int i = 0; i < n; ++i) {
        G[i * N + i] += tot[i];
        if (c == 1) G[i * N + n] += tot[i];
        for (l = 0; l < kc; ++l) if (evl[l] != i) G[i * N + evl[l]] += mar[i * 32 + l];
      }
      /* observation-rate rows from the same marginals: sum_S D(S) (sum_T p q)(S) [l in S] */
      const double* DT = c == 0 ? J->DpT : J->DmT;
      double* dd = c == 0 ? ddp : ddm;
      double all = 0;
      for (size_t S = 0; S < VS; ++S) {
        const double w = -W[S] * DT[S];
        all += w;
        for (size_t m = S; m; m &= m - 1) dd[evl[__builtin_ctzl(m)]] -= w;
      }
      dd[n] -= all;
      free(W);
      for (l = 0; l < kc; ++l) free(Vb[l]);
    }
  }
  /* seed = 0 region: synchronised events + seeding over the subsets e of the paired events */
  {
    const int ke = J->ke;
    const size_t VE = (size_t)1 << ke;
    int bits[32], evl[32], l = 0;
    for (uint32_t m = J->pairP; m; m &= m - 1) { bits[l] = __builtin_ctz(m); evl[l] = J->ev[bits[l]]; ++l; }
    double* Ad = (double*)calloc(VE, sizeof(double));
    double* Ab[32];
    double* As = (double*)calloc(VE, sizeof(double));
    for (l = 0; l < ke; ++l) Ab[l] = (double*)calloc(VE, sizeof(double));
    for (size_t e = 0; e < VE; ++e) {
      const uint32_t xp = pdep32((uint32_t)e, J->pairP);
      const uint32_t x0 = xp | (xp << 1);
      Ad[e] = -p[x0] * q[x0];
      for (l = 0; l < ke; ++l) if (!((e >> l) & 1)) Ab[l][e] = p[x0] * q[x0 | (3u << bits[l])];
      if (sb) As[e] = p[x0] * q[x0 | sb];
    }
    double base[64];
    for (int i = 0; i < N; ++i) base[i] = i < n ? th[i * N + i] : 0.0;
    grad_accum(th, N, base, evl, ke, Ad, Ab, tot, mar);
    for (int i = 0; i < n; ++i) {
      G[i * N + i] += tot[i];
      for (l = 0; l < ke; ++l) if (evl[l] != i) G[i * N + evl[l]] += mar[i * 32 + l];
    }
    for (size_t e = 0; e < VE; ++e) {
      double rate = th[n * N + n];
      for (size_t m = e; m; m &= m - 1) rate *= th[n * N + evl[__builtin_ctzl(m)]];
      const double f = rate * (As[e] + Ad[e]);
      G[n * N + n] += f;
      for (size_t m = e; m; m &= m - 1) G[n * N + evl[__builtin_ctzl(m)]] += f;
      /* D_p on the unseeded PT == MT states: prod dp over the PT bits (no dp[n]); D_m = 0 there */
      double w = -Ad[e];
      for (size_t m = e; m; m &= m - 1) w *= dp[evl[__builtin_ctzl(m)]];
      for (size_t m = e; m; m &= m - 1) ddp[evl[__builtin_ctzl(m)]] -= w;
    }
    free(Ad); free(As);
    for (l = 0; l < ke; ++l) free(Ab[l]);
  }
  free(tot); free(mar);
}

/* ---- one paired patient (closed_form.py:384-440) --------------------------------------------------------- */
static int paired_patient(int n, const double* lt, const double* ldp, const double* ldm, const int8_t* row, double* lp,
                          double* G, double* ddp, double* ddm) {
  const int N = n + 1;
  int order = row[2 * n + 1];
  if (order != 0 && order != 1) order = 2;                       /* regularized_optimization.py:114, 245 */
  double *th = (double*)malloc(sizeof(double) * N * N), *thM = (double*)malloc(sizeof(double) * N * N),
         *thP = (double*)malloc(sizeof(double) * N * N), dp[64], dm[64];
  for (int i = 0; i < N; ++i) { dp[i] = exp(ldp[i]); dm[i] = exp(ldm[i]); }
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      const double t = exp(lt[i * N + j]);
      th[i * N + j] = t;
      thM[i * N + j] = i == j ? t : t / dm[j];                                  /* diagnosis_theta, kronvec.py:7-21 */
      thP[i * N + j] = i == j ? t : ((j == n && i < n) ? 1.0 : t) / dp[j];      /* likelihood.py:313-314 */
    }
  memset(G, 0, sizeof(double) * N * N);
  memset(ddp, 0, sizeof(double) * N);
  memset(ddm, 0, sizeof(double) * N);
  jspace J;
  js_init(&J, row, n, th, dp, dm);
  if (J.seedbit < 0) { js_free(&J); free(th); free(thM); free(thP); return 1; }
  const size_t V = (size_t)1 << J.k;
  double* pi = (double*)malloc(sizeof(double) * V);
  double* q = (double*)malloc(sizeof(double) * V);
  js_forward(&J, pi);
  const uint32_t sb = 1u << J.seedbit;
  /* marginal problems: [0] PT observed first (metastasis keeps evolving), [1] MT observed first */
  sspace S[2];
  double *v[2] = {NULL, NULL}, *fw[2] = {NULL, NULL}, *qm[2] = {NULL, NULL};
  int use[2] = {order == 0 || order == 1, order != 1};
  double cst[2] = {0, 0};
  double full = 0;
  for (int part = 0; part < 2; ++part) {
    if (!use[part]) continue;
    const uint32_t fixed = (part == 0 ? J.maskP : J.maskM) | sb, freem = part == 0 ? J.maskM : J.maskP;
    int evs[32], ks = 0;
    for (uint32_t m = freem; m; m &= m - 1) evs[ks++] = J.ev[__builtin_ctz(m)];
    evs[ks++] = n;                                                /* seeding = MSB of the marginal space */
    ss_init(&S[part], N, part == 0 ? thM : thP, evs, ks);
    const size_t VS = (size_t)1 << ks, half = VS >> 1;
    v[part] = (double*)calloc(VS, sizeof(double));
    fw[part] = (double*)malloc(sizeof(double) * VS);
    qm[part] = (double*)malloc(sizeof(double) * VS);
    /* D_obs on the compatible states is constant: every bit of the observed tumour and seeding are set */
    const double* DT = part == 0 ? J.DpT : J.DmT;
    cst[part] = DT[((size_t)1 << (part == 0 ? J.kP : J.kM)) - 1];
    for (size_t m = 0; m < half; ++m) v[part][half + m] = cst[part] * pi[fixed | pdep32((uint32_t)m, freem)];
    ss_solve(&S[part], v[part], fw[part], 0);
    full += fw[part][VS - 1];
  }
  *lp = log(full);
  double* val = (double*)malloc(sizeof(double) * N * N);
  double dd[64];
  for (int part = 0; part < 2; ++part) {
    if (!use[part]) continue;
    const size_t VS = (size_t)1 << S[part].k, half = VS >> 1;
    double* el = (double*)calloc(VS, sizeof(double));
    el[VS - 1] = 1.0 / full;                                      /* adjoint seeded with 1/full: linear mixing */
    ss_solve(&S[part], el, qm[part], 1);
    free(el);
    ss_xQy(&S[part], qm[part], fw[part], val, dd);
    double dot = 0;
    for (size_t m = 0; m < half; ++m) dot += qm[part][half + m] * v[part][half + m];
    if (part == 0) {
      for (int e = 0; e < N * N; ++e) G[e] += val[e];
      for (int j = 0; j < N; ++j) ddm[j] += dd[j];
      ddp[n] += dot;
      for (uint32_t m = J.maskP; m; m &= m - 1) ddp[J.ev[__builtin_ctz(m)]] += dot;
    } else {
      for (int i = 0; i < n; ++i) val[i * N + n] = 0.0;
      for (int e = 0; e < N * N; ++e) G[e] += val[e];
      for (int j = 0; j < N; ++j) ddp[j] += dd[j];
      ddm[n] += dot;
      for (uint32_t m = J.maskM; m; m &= m - 1) ddm[J.ev[__builtin_ctz(m)]] += dot;
    }
  }
  {
    const size_t h0 = use[0] ? ((size_t)1 << S[0].k) >> 1 : 0, h1 = use[1] ? ((size_t)1 << S[1].k) >> 1 : 0;
    js_adjoint(&J, cst[0], use[0] ? qm[0] + h0 : NULL, cst[1], use[1] ? qm[1] + h1 : NULL, q);
  }
  js_gradient(&J, th, dp, dm, q, pi, G, ddp, ddm);
  for (int part = 0; part < 2; ++part)
    if (use[part]) { ss_free(&S[part]); free(v[part]); free(fw[part]); free(qm[part]); }
  free(val); free(pi); free(q);
  js_free(&J);
  free(th); free(thM); free(thP);
  return 0;
}

/* Paired rows only (type 3 with seeding): returns 1 + the index of the first other row, 0 on success. */
int fast_patients(int n, const double* lt, const double* ldp, const double* ldm, const int8_t* dat, int64_t n_pat,
                  int n_threads, double* lp, double* g, double* dp, double* dm) {
  const int N = n + 1, cols = 2 * n + 3;
  for (int64_t r = 0; r < n_pat; ++r)
    if (dat[r * cols + 2 * n + 2] != 3 || !dat[r * cols + 2 * n]) return (int)(r + 1);
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t r = 0; r < n_pat; ++r)
    paired_patient(n, lt, ldp, ldm, dat + r * cols, lp + r, g + r * N * N, dp + r * N, dm + r * N);
  return 0;
}
