/*
 * CPU restatement (plain C, fp64) of the metMHN likelihood / gradient hot path with the
 * reference's pass structure - TEST INFRASTRUCTURE and CPU BASELINE only.
 *
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load this
 * library (oracle/_build/libmetmhn_ref.so); the product path never does.
 *
 * Like oracle/metmhn_oracle.py (which it is validated against, and through it against the
 * golden vectors of the reference's own source) every Kronecker summand is applied factor
 * by factor to the lowest one or two index bits of the restricted vector followed by the
 * rotation of those bits to the top, i.e. `reshape(-1, w, 'C') @ T` then `flatten('F')`
 * - one read and one write pass over the 2^k vector per factor, k factors per summand,
 * 3n+1 summands per product, exactly the work the JAX reference schedules
 * (metmhn/jx/kronvec.py:214-539).  Paths below are relative to /root/reference.
 *
 * Parallelism: OpenMP over patients (the reference is single-threaded; the baseline uses
 * every host core so the GPU speed-up is not flattered).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* inner-loop parallelism: active when the caller runs patients one after the other (few large
 * patients); inside the patient-parallel region nested parallelism is off and these run serially */
#define PARFOR _Pragma("omp parallel for schedule(static) if (len >= 262144)")
#define PARSUM(v) _Pragma("omp parallel for schedule(static) reduction(+ : s) if (len >= 262144)")  /* always reduces `s` */

typedef struct {
  int n;            /* mutations; N = n + 1 */
  const double* lt; /* log_theta [N][N] */
  const double* ldp;
  const double* ldm;
} par_t;

/* ---- one factor: kinds of T ------------------------------------------------------- */
enum { F_NONE, F_SCAL, F_D2, F_D4, F_M2, F_M4 };
typedef struct {
  int kind;
  double s;        /* F_SCAL */
  double d[4];     /* diagonal factors */
  double m[16];    /* m[c_from * w + c_to], written for p_rows @ T (kronvec.py:82-147) */
} fac_t;

/* out = flatten_F(reshape_C(p, (-1, w)) @ T); returns the buffer holding the result */
static double* apply(const fac_t* f, double* p, double* tmp, size_t len) {
  switch (f->kind) {
    case F_NONE: return p;
    case F_SCAL: PARFOR for (size_t i = 0; i < len; ++i) p[i] *= f->s; return p;
    case F_D2: { size_t h = len >> 1;
      PARFOR for (size_t r = 0; r < h; ++r) { tmp[r] = p[2 * r] * f->d[0]; tmp[h + r] = p[2 * r + 1] * f->d[1]; }
      return tmp; }
    case F_D4: { size_t h = len >> 2;
      PARFOR for (size_t r = 0; r < h; ++r) for (int c = 0; c < 4; ++c) tmp[c * h + r] = p[4 * r + c] * f->d[c];
      return tmp; }
    case F_M2: { size_t h = len >> 1;
      PARFOR for (size_t r = 0; r < h; ++r) { double a = p[2 * r], b = p[2 * r + 1];
        tmp[r] = a * f->m[0] + b * f->m[2]; tmp[h + r] = a * f->m[1] + b * f->m[3]; }
      return tmp; }
    default: { size_t h = len >> 2;
      PARFOR for (size_t r = 0; r < h; ++r) for (int c = 0; c < 4; ++c) { double s = 0;
        for (int cf = 0; cf < 4; ++cf) s += p[4 * r + cf] * f->m[cf * 4 + c];
        tmp[c * h + r] = s; }
      return tmp; }
  }
}
/* apply and keep the result in `*p` / scratch in `*tmp` (ping-pong) */
static void step(const fac_t* f, double** p, double** tmp, size_t len) {
  double* r = apply(f, *p, *tmp, len);
  if (r != *p) { *tmp = *p; *p = r; }
}
static fac_t f_none(void) { fac_t f; memset(&f, 0, sizeof f); f.kind = F_NONE; return f; }
static fac_t f_scal(double s) { fac_t f = f_none(); f.kind = F_SCAL; f.s = s; return f; }
static fac_t f_d2(double a, double b) { fac_t f = f_none(); f.kind = F_D2; f.d[0] = a; f.d[1] = b; return f; }
static fac_t f_d4(double a, double b, double c, double d) {
  fac_t f = f_none(); f.kind = F_D4; f.d[0] = a; f.d[1] = b; f.d[2] = c; f.d[3] = d; return f; }
static fac_t f_k2ntt(double th, int diag, int tr) {            /* kronvec.py:82-93 */
  fac_t f = f_none(); f.kind = F_M2;
  if (tr) f.m[1 * 2 + 0] = th; else f.m[0 * 2 + 1] = th;
  if (diag) f.m[0] = -th;
  return f; }
static fac_t f_k4(double th, int diag, int tr, int s0, int d0, int s1, int d1) { /* kronvec.py:96-147 */
  fac_t f = f_none(); f.kind = F_M4;
  int src[2] = {s0, s1}, dst[2] = {d0, d1};
  for (int q = 0; q < 2; ++q) { if (src[q] < 0) continue;
    if (tr) f.m[dst[q] * 4 + src[q]] = th; else f.m[src[q] * 4 + dst[q]] = th;
    if (diag) f.m[src[q] * 4 + src[q]] = -th; }
  return f; }

static int sel(const int8_t* st, int j) { return st[2 * j] + 2 * st[2 * j + 1]; }

/* passive factor tables (kronvec.py:226, :302, :373) */
static fac_t pas(int part, int c, double th) {
  if (c == 0) return f_none();
  if (part == 0) return c == 3 ? f_d4(1, 0, 0, th) : f_d2(1, 0);
  if (part == 1) return c == 1 ? f_d2(1, th) : c == 2 ? f_d2(1, 1) : f_d4(1, th, 1, th);
  return c == 1 ? f_d2(1, 1) : c == 2 ? f_d2(1, th) : f_d4(1, 1, th, th);
}
/* acting factor tables (kronvec.py:238, :315, :386) */
static fac_t act(int part, int c, double th, int diag, int tr) {
  if (c == 0) return f_scal(-th);
  if (part == 0) return c == 3 ? f_k4(th, diag, tr, 0, 3, -1, -1) : f_d2(-th, 0);
  if (part == 1) return c == 1 ? f_k2ntt(th, diag, tr) : c == 2 ? f_d2(-th, -th) : f_k4(th, diag, tr, 0, 1, 2, 3);
  return c == 1 ? f_d2(-th, -th) : c == 2 ? f_k2ntt(th, diag, tr) : f_k4(th, diag, tr, 0, 2, 1, 3);
}

/* one summand: part 0 sync, 1 prim, 2 met (i < n) or 3 seed.  in -> out (out may be scratch);
 * returns 0 if the summand is identically zero (early-outs kronvec.py:283-287,353-359,425-431,491-496) */
static int summand(const par_t* P, int part, int i, const int8_t* st, int diag, int tr, const double* in,
                   double* out, double* tmp, size_t len) {
  const int n = P->n, N = n + 1;
  const int seed = st[2 * n];
  if (part == 0 && !diag && st[2 * i] + st[2 * i + 1] != 2) return 0;
  if (part == 1 && ((!diag && st[2 * i] == 0) || !seed)) return 0;
  if (part == 2 && ((!diag && st[2 * i + 1] == 0) || !seed)) return 0;
  if (part == 3 && !diag && !seed) return 0;
  const int row = part == 3 ? n : i;
  const double* lrow = P->lt + (size_t)row * N;
  memcpy(out, in, len * sizeof(double));
  double *p = out, *t = tmp;
  for (int j = 0; j < n; ++j) {
    const int c = sel(st, j);
    fac_t f = (part != 3 && j == i) ? act(part, c, exp(lrow[i]), diag, tr) : pas(part == 3 ? 0 : part, c, exp(lrow[j]));
    step(&f, &p, &t, len);
  }
  fac_t f;
  if (part == 0) f = seed ? f_d2(1, 0) : f_none();                 /* kronvec.py:246-250 */
  else if (part == 1) f = f_d2(0, 1);                              /* :323 */
  else if (part == 2) f = f_d2(0, exp(lrow[n]));                   /* :395 */
  else f = seed ? f_k2ntt(exp(lrow[n]), diag, tr) : f_scal(-exp(lrow[n]));   /* :457-462 */
  step(&f, &p, &t, len);
  if (p != out) memcpy(out, p, len * sizeof(double));
  return 1;
}

/* y = Q p (kronvec.py:499-539); w1, w2 scratch of length len */
static void kronvec(const par_t* P, const int8_t* st, int diag, int tr, const double* p, double* y, double* w1,
                    double* w2, size_t len) {
  memset(y, 0, len * sizeof(double));
  for (int i = 0; i < P->n; ++i)
    for (int part = 0; part < 3; ++part)
      if (summand(P, part, i, st, diag, tr, p, w1, w2, len)) { PARFOR for (size_t e = 0; e < len; ++e) y[e] += w1[e]; }
  if (summand(P, 3, 0, st, diag, tr, p, w1, w2, len)) { PARFOR for (size_t e = 0; e < len; ++e) y[e] += w1[e]; }
}

/* diag(Q) (kronvec.py:713-999) */
static void kron_diag(const par_t* P, const int8_t* st, double* y, double* w1, double* w2, size_t len) {
  const int n = P->n, N = n + 1, seed = st[2 * n];
  memset(y, 0, len * sizeof(double));
  for (int part = 0; part < 4; ++part)
    for (int i = 0; i < (part == 3 ? 1 : n); ++i) {
      if ((part == 1 || part == 2) && !seed) continue;
      const int row = part == 3 ? n : i;
      const double* lrow = P->lt + (size_t)row * N;
      for (size_t e = 0; e < len; ++e) w1[e] = 1.0;
      double *p = w1, *t = w2;
      for (int j = 0; j < n; ++j) {
        const int c = sel(st, j);
        fac_t f;
        const double th = exp(lrow[i]);
        if (part != 3 && j == i) {
          if (c == 0) f = f_scal(-th);
          else if (part == 0) f = c == 3 ? f_d4(-th, 0, 0, 0) : f_d2(-th, 0);                       /* :747-751 */
          else if (part == 1) f = c == 1 ? f_d2(-th, 0) : c == 2 ? f_d2(-th, -th) : f_d4(-th, 0, -th, 0);   /* :795-798 */
          else f = c == 1 ? f_d2(-th, -th) : c == 2 ? f_d2(-th, 0) : f_d4(-th, -th, 0, 0);         /* :871-875 */
        } else f = pas(part == 3 ? 0 : part, c, exp(lrow[j]));
        step(&f, &p, &t, len);
      }
      fac_t f;
      if (part == 0) f = seed ? f_d2(1, 0) : f_none();
      else if (part == 1) f = f_d2(0, 1);
      else if (part == 2) f = f_d2(0, exp(lrow[n]));
      else f = seed ? f_d2(-exp(lrow[n]), 0) : f_scal(-exp(lrow[n]));
      step(&f, &p, &t, len);
      { PARFOR for (size_t e = 0; e < len; ++e) y[e] += p[e]; }
    }
}

/* observation diagonals (kronvec.py:574-710): which 0 = p, 1 = m; part_i < 0: full product,
 * else the partial derivative w.r.t. log d[part_i]; out <- result */
static void diag_scal(const par_t* P, int which, int part_i, const int8_t* st, const double* in, double* out,
                      double* tmp, size_t len) {
  const int n = P->n;
  const double* ld = which == 0 ? P->ldp : P->ldm;
  if (part_i >= 0) {
    int sw = which == 0 ? st[2 * part_i] + (part_i == n) : st[(2 * part_i + 1 < 2 * n) ? 2 * part_i + 1 : 2 * n] + (part_i == n);
    if (sw == 0) { memset(out, 0, len * sizeof(double)); return; }
    if (sw == 2) {
      diag_scal(P, which, -1, st, in, out, tmp, len);
      if (which == 0) memset(out, 0, (len / 2) * sizeof(double));   /* partial_le: zero the seed = 0 half */
      return;
    }
  }
  memcpy(out, in, len * sizeof(double));
  double *p = out, *t = tmp;
  for (int j = 0; j < n; ++j) {
    const int c = sel(st, j);
    const double d = exp(ld[j]);
    fac_t f;
    if (j == part_i) {
      if (which == 0) f = (c == 1 || c == 2) ? f_d2(0, d) : f_d4(0, d, 0, d);        /* :618-621 */
      else f = (c == 1 || c == 2) ? f_d2(0, d) : f_d4(0, 0, d, d);                    /* :696-699 */
    } else f = pas(which == 0 ? 1 : 2, c, d);
    step(&f, &p, &t, len);
  }
  fac_t f = which == 0 ? f_d2(1, exp(ld[n])) : f_d2(0, exp(ld[n]));
  step(&f, &p, &t, len);
  if (p != out) memcpy(out, p, len * sizeof(double));
}

/* (D_p + D_m - Q)^-1 x by k+1 Jacobi sweeps (likelihood.py:231-262); ws: 5 scratch vectors */
static void R_i_inv_vec(const par_t* P, const int8_t* st, int k, int tr, const double* x, double* y, double* ws,
                        size_t len) {
  double *lidg = ws, *w1 = ws + len, *w2 = ws + 2 * len, *q = ws + 3 * len, *ones = ws + 4 * len;
  kron_diag(P, st, lidg, w1, w2, len);
  for (size_t e = 0; e < len; ++e) ones[e] = 1.0;
  diag_scal(P, 0, -1, st, ones, q, w1, len);
  diag_scal(P, 1, -1, st, ones, y, w1, len);
  for (size_t e = 0; e < len; ++e) lidg[e] = -1.0 / (lidg[e] - (q[e] + y[e]));
  for (size_t e = 0; e < len; ++e) y[e] = lidg[e] * x[e];
  for (int s = 0; s <= k; ++s) {
    kronvec(P, st, 0, tr, y, q, w1, w2, len);
    { PARFOR for (size_t e = 0; e < len; ++e) y[e] = lidg[e] * (q[e] + x[e]); }
  }
}

/* peel the lowest w columns: returns sums needed by the reducers and rotates z in place (via tmp) */
static void rot(double** z, double** tmp, int w, size_t len) { fac_t f = w == 2 ? f_d2(1, 1) : f_d4(1, 1, 1, 1); step(&f, z, tmp, len); }
static double colsum(const double* z, int w, int col, size_t len) { double s = 0; PARSUM(s) for (size_t r = 0; r < len / w; ++r) s += z[r * w + col]; return s; }
static double allsum(const double* z, size_t len) { double s = 0; PARSUM(s) for (size_t e = 0; e < len; ++e) s += z[e]; return s; }

/* x^T dQ y (likelihood.py:125-201); ws: 8 scratch vectors */
static void x_partial_Q_y(const par_t* P, const int8_t* st, const double* x, const double* y, double* G, double* ws,
                          size_t len) {
  const int n = P->n, N = n + 1;
  double *zs = ws, *zp = ws + len, *zm = ws + 2 * len, *t1 = ws + 3 * len, *t2 = ws + 4 * len, *t3 = ws + 5 * len,
         *w1 = ws + 6 * len;
  memset(G, 0, sizeof(double) * N * N);
  for (int i = 0; i < n; ++i) {
    double* z[3] = {zs, zp, zm};
    double* tt[3] = {t1, t2, t3};
    for (int part = 0; part < 3; ++part) {
      if (summand(P, part, i, st, 1, 0, y, z[part], w1, len)) { PARFOR for (size_t e = 0; e < len; ++e) z[part][e] *= x[e]; }
      else memset(z[part], 0, len * sizeof(double));
    }
    G[i * N + n] = allsum(z[2], len);
    for (int j = 0; j < n; ++j) {
      const int c = sel(st, j);
      double val = 0;
      if (j == i) {                                           /* t1 / t12 / t3, likelihood.py:75-107 */
        if (c == 0) val = allsum(z[0], len) + allsum(z[1], len) + allsum(z[2], len);
        else if (c == 3) val = allsum(z[0], len) + allsum(z[1], len) + allsum(z[2], len);
        else val = colsum(z[0], 2, 0, len) + allsum(z[1], len) + allsum(z[2], len);
      } else if (c == 1) val = colsum(z[1], 2, 1, len);       /* f1 */
      else if (c == 2) val = colsum(z[2], 2, 1, len);         /* f2 */
      else if (c == 3) val = colsum(z[0], 4, 3, len) + colsum(z[1], 4, 1, len) + colsum(z[1], 4, 3, len) +
                             colsum(z[2], 4, 2, len) + colsum(z[2], 4, 3, len);   /* f3 */
      if (c != 0) for (int part = 0; part < 3; ++part) rot(&z[part], &tt[part], c == 3 ? 4 : 2, len);
      G[i * N + j] = val;
    }
  }
  double *z = zs, *t = t1;
  if (summand(P, 3, 0, st, 1, 0, y, z, w1, len)) for (size_t e = 0; e < len; ++e) z[e] *= x[e];
  else memset(z, 0, len * sizeof(double));
  G[n * N + n] = allsum(z, len);
  for (int j = 0; j < n; ++j) {                               /* z0 / z1 / z3, likelihood.py:110-122 */
    const int c = sel(st, j);
    double val = 0;
    if (c == 3) val = colsum(z, 4, 3, len);
    if (c != 0) rot(&z, &t, c == 3 ? 4 : 2, len);
    G[n * N + j] = val;
  }
}

/* x^T dD y (likelihood.py:204-228); ws: 3 scratch vectors */
static void x_partial_D_y(const par_t* P, const int8_t* st, const double* x, const double* y, double* ddp,
                          double* ddm, double* ws, size_t len) {
  const int N = P->n + 1;
  double *o = ws, *t = ws + len;
  for (int i = 0; i < N; ++i) {
    diag_scal(P, 0, i, st, y, o, t, len);
    double s = 0; PARSUM(s) for (size_t e = 0; e < len; ++e) s += x[e] * o[e];
    ddp[i] = s;
    diag_scal(P, 1, i, st, y, o, t, len);
    s = 0; PARSUM(s) for (size_t e = 0; e < len; ++e) s += x[e] * o[e];
    ddm[i] = s;
  }
}

/* ---- single-tumour MHN (vanilla.py), theta given as log matrix ----------------------- */
static int v_summand(const double* lth, int N, int i, const int8_t* st, int diag, int tr, const double* in,
                     double* out, double* tmp, size_t len) {
  if (!diag && st[i] != 1) return 0;                              /* vanilla.py:70-75 */
  memcpy(out, in, len * sizeof(double));
  double *p = out, *t = tmp;
  for (int j = 0; j < N; ++j) {
    fac_t f;
    if (j == i) f = st[i] == 0 ? f_scal(-exp(lth[i * N + i])) : f_k2ntt(exp(lth[i * N + i]), diag, tr);
    else f = st[j] == 0 ? f_none() : f_d2(1, exp(lth[i * N + j]));
    step(&f, &p, &t, len);
  }
  if (p != out) memcpy(out, p, len * sizeof(double));
  return 1;
}
static void v_kronvec(const double* lth, int N, const int8_t* st, int diag, int tr, const double* p, double* y,
                      double* w1, double* w2, size_t len) {
  memset(y, 0, len * sizeof(double));
  for (int i = 0; i < N; ++i)
    if (v_summand(lth, N, i, st, diag, tr, p, w1, w2, len)) { PARFOR for (size_t e = 0; e < len; ++e) y[e] += w1[e]; }
}
static void v_kron_diag(const double* lth, int N, const int8_t* st, double* y, double* w1, double* w2, size_t len) {
  memset(y, 0, len * sizeof(double));
  for (int i = 0; i < N; ++i) {                                    /* vanilla.py:206-245 */
    for (size_t e = 0; e < len; ++e) w1[e] = 1.0;
    double *p = w1, *t = w2;
    for (int j = 0; j < N; ++j) {
      fac_t f;
      if (j == i) f = st[i] == 0 ? f_scal(-exp(lth[i * N + i])) : f_d2(-exp(lth[i * N + i]), 0);
      else f = st[j] == 0 ? f_none() : f_d2(1, exp(lth[i * N + j]));
      step(&f, &p, &t, len);
    }
    { PARFOR for (size_t e = 0; e < len; ++e) y[e] += p[e]; }
  }
}
/* (diag(d_rates) - Q)^-1 x, d_rates == NULL -> 1 (vanilla.py:269-305); ws: 4 vectors */
static void v_R_inv_vec(const double* lth, int N, const int8_t* st, int k, const double* dr, int tr, const double* x,
                        double* y, double* ws, size_t len) {
  double *lidg = ws, *w1 = ws + len, *w2 = ws + 2 * len, *q = ws + 3 * len;
  v_kron_diag(lth, N, st, lidg, w1, w2, len);
  for (size_t e = 0; e < len; ++e) lidg[e] = -1.0 / (lidg[e] - (dr ? dr[e] : 1.0));
  for (size_t e = 0; e < len; ++e) y[e] = lidg[e] * x[e];
  for (int s = 0; s <= k; ++s) {
    v_kronvec(lth, N, st, 0, tr, y, q, w1, w2, len);
    { PARFOR for (size_t e = 0; e < len; ++e) y[e] = lidg[e] * (q[e] + x[e]); }
  }
}
/* vanilla.py:328-393; ws: 3 vectors */
static void v_x_partial_Q_y(const double* lth, int N, const int8_t* st, const double* x, const double* y, double* val,
                            double* ddiag, double* ws, size_t len) {
  double *zb = ws, *tb = ws + len, *w1 = ws + 2 * len;
  memset(val, 0, sizeof(double) * N * N);
  for (int i = 0; i < N; ++i) {
    double *z = zb, *t = tb;
    v_summand(lth, N, i, st, 1, 0, y, z, w1, len);
    for (size_t e = 0; e < len; ++e) z[e] *= x[e];
    for (int j = 0; j < N; ++j) {
      if (j == i) { val[i * N + j] = allsum(z, len); if (st[i]) rot(&z, &t, 2, len); }
      else if (st[j]) { val[i * N + j] = colsum(z, 2, 1, len); rot(&z, &t, 2, len); }
    }
  }
  if (ddiag) for (int j = 0; j < N; ++j) { double s = 0; for (int i = 0; i < N; ++i) if (i != j) s -= val[i * N + j]; ddiag[j] = s; }
}
/* vanilla.py:396-418: p_theta, x = adjoint, d_th, d_diag; ws: 6 vectors */
static void v_gradient(const double* lth, int N, const int8_t* st, int k, const double* p0, double* dth,
                       double* ddiag, double* pth, double* xadj, double* ws, size_t len) {
  v_R_inv_vec(lth, N, st, k, NULL, 0, p0, pth, ws, len);
  double* el = ws + 4 * len;
  memset(el, 0, len * sizeof(double));
  el[len - 1] = 1.0 / pth[len - 1];
  v_R_inv_vec(lth, N, st, k, NULL, 1, el, xadj, ws, len);
  v_x_partial_Q_y(lth, N, st, xadj, pth, dth, ddiag, ws, len);
}

static void diagnosis_theta(const double* lt, const double* ld, int N, int zero_seed_col, double* out) {
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {       /* kronvec.py:7-21, likelihood.py:313 */
    double v = lt[i * N + j];
    if (zero_seed_col && j == N - 1 && i < N - 1) v = 0.0;
    out[i * N + j] = i == j ? lt[i * N + i] : v - ld[j];
  }
}

/* obs_states (kronvec.py:1031-1095) by its own passes -> ascending indices */
static size_t obs_indices(const par_t* P, const int8_t* st, int pt_first, size_t len, double* w1, double* w2,
                          int64_t* idx) {
  const int n = P->n;
  for (size_t e = 0; e < len; ++e) w1[e] = 1.0;
  double *p = w1, *t = w2;
  for (int j = 0; j < n; ++j) {
    const int c = sel(st, j);
    fac_t f;
    if (c == 0) f = f_none();
    else if (c == 1) f = pt_first ? f_d2(0, 1) : f_d2(1, 1);
    else if (c == 2) f = pt_first ? f_d2(1, 1) : f_d2(0, 1);
    else f = pt_first ? f_d4(0, 1, 0, 1) : f_d4(0, 0, 1, 1);
    step(&f, &p, &t, len);
  }
  if (st[2 * n]) { fac_t f = f_d2(0, 1); step(&f, &p, &t, len); }
  size_t c = 0;
  for (size_t e = 0; e < len; ++e) if (p[e] == 1.0) idx[c++] = (int64_t)e;
  return c;
}

/* ---- one patient: lp and gradient (likelihood.py:286-731, regularized_optimization.py:187-254) ---- */
typedef struct { double lp; double* g; double* dp; double* dm; } res_t;

static void single_patient(const par_t* P, const int8_t* row, int type, int with_grad, res_t* R) {
  const int n = P->n, N = n + 1;
  int8_t st[64];
  double lth[32 * 32];
  if (type == 2) { for (int j = 0; j < n; ++j) st[j] = row[2 * j + 1]; st[n] = 1; }
  else for (int j = 0; j <= n; ++j) st[j] = row[2 * j];
  int k = 0; for (int j = 0; j <= n; ++j) k += st[j];
  if (type == 0 && k == 0) {                                       /* likelihood.py:408-416, 464-478 */
    double s = 0; for (int i = 0; i < N; ++i) s += exp(P->lt[i * N + i]);
    R->lp = log(1.0 / (1.0 + s));
    if (with_grad) for (int i = 0; i < N; ++i) R->g[i * N + i] = -exp(P->lt[i * N + i]) / (1.0 + s);
    return;
  }
  const size_t len = (size_t)1 << k;
  double* ws = (double*)malloc(sizeof(double) * len * 12);
  double *p0 = ws, *pth = ws + len, *xadj = ws + 2 * len, *dr = ws + 3 * len, *w = ws + 4 * len;
  memset(p0, 0, len * sizeof(double)); p0[0] = 1.0;
  if (type != 2) {                                                 /* _lp_prim_obs / _grad_prim_obs */
    diagnosis_theta(P->lt, P->ldp, N, 1, lth);
    if (!with_grad) { v_R_inv_vec(lth, N, st, k, NULL, 0, p0, pth, w, len); R->lp = log(pth[len - 1]); }
    else {
      v_gradient(lth, N, st, k, p0, R->g, R->dp, pth, xadj, w, len);
      for (int i = 0; i < n; ++i) R->g[i * N + n] = 0.0;
      R->lp = log(pth[len - 1]);
    }
  } else {                                                         /* _lp_met_obs / _grad_met_obs */
    /* scal_d_pt(ones) (vanilla.py:125-142): d_rates = d_p part (seed = 0) + d_m part (seed = 1) */
    for (size_t x = 0; x < len; ++x) {
      double a = 1, b = 1; int bit = 0;
      for (int j = 0; j < n; ++j) if (st[j]) { if (x >> bit & 1) { a *= exp(P->ldp[j]); b *= exp(P->ldm[j]); } ++bit; }
      dr[x] = (x >> (k - 1) & 1) ? b * exp(P->ldm[n]) : a;
    }
    v_R_inv_vec(P->lt, N, st, k, dr, 0, p0, pth, w, len);
    R->lp = log(pth[len - 1] * dr[len - 1]);
    if (with_grad) {
      double* el = w + 4 * len;
      memset(el, 0, len * sizeof(double)); el[len - 1] = 1.0 / pth[len - 1];
      v_R_inv_vec(P->lt, N, st, k, dr, 1, el, xadj, w, len);
      v_x_partial_Q_y(P->lt, N, st, xadj, pth, R->g, NULL, w, len);
      /* x_partial_D_y of vanilla.py:190-203 element-wise (the d_scal_d_pt passes are diagonal) */
      int bit = 0;
      for (int j = 0; j <= n; ++j) {
        if (!st[j]) continue;
        double sp = 0, sm = 0;
        for (size_t x = 0; x < len; ++x) if (x >> bit & 1) {
          const int seeded = (int)(x >> (k - 1) & 1);
          if (!seeded) sp += xadj[x] * pth[x] * dr[x]; else sm += xadj[x] * pth[x] * dr[x];
        }
        R->dp[j] = j == n ? 0.0 : -sp;
        R->dm[j] = 1.0 - sm;                                       /* d_dm_1 - d_dm_2, likelihood.py:508-512 */
        ++bit;
      }
    }
  }
  free(ws);
}

/* k = 1 paired rows: one_event.py closed forms */
static void one_event_patient(const par_t* P, const int8_t* row, int order, int with_grad, res_t* R) {
  const int n = P->n, N = n + 1;
  double lthM[32 * 32], lthP[32 * 32];
  int8_t met[64], prim[64];
  for (int j = 0; j < n; ++j) { met[j] = row[2 * j + 1]; prim[j] = row[2 * j]; }
  met[n] = 1; prim[n] = row[2 * n];
  diagnosis_theta(P->lt, P->ldm, N, 0, lthM);
  diagnosis_theta(P->lt, P->ldp, N, 1, lthP);
  const double dpl = exp(P->ldp[n]), dml = exp(P->ldm[n]);
  double q00 = 0, q11 = 0;                                         /* small_Q, one_event.py:10-25 */
  for (int i = 0; i < N; ++i) q00 -= exp(P->lt[i * N + i]);
  for (int i = 0; i < n; ++i) q11 -= exp(P->lt[i * N + i]) * (exp(P->lt[i * N + n]) + 1.0);
  const double q10 = exp(P->lt[n * N + n]);
  const double r00 = 1.0 - q00, r10 = -q10, r11 = dpl + dml - q11;
  double pth1[2]; pth1[0] = 1.0 / r00; pth1[1] = (0.0 - pth1[0] * r10) / r11;
  double ws[2 * 8];
  double sc[2] = {0, 0};
  double g1[2][32 * 32], dd[2][32], pp[2][2], qq[2][2];
  int use[2] = {order == 0 || order == 1, order != 1};
  for (int part = 0; part < 2; ++part) {
    if (!use[part]) continue;
    double v[2] = {0.0, pth1[1] * (part == 0 ? dpl : dml)};
    const double* lth = part == 0 ? lthM : lthP;
    const int8_t* s = part == 0 ? met : prim;
    if (!with_grad) { double y[2]; v_R_inv_vec(lth, N, s, 1, NULL, 0, v, y, ws, 2); sc[part] = y[1]; }
    else {
      double xa[2];
      v_gradient(lth, N, s, 1, v, g1[part], dd[part], pp[part], xa, ws, 2);
      if (part == 1) for (int i = 0; i < n; ++i) g1[1][i * N + n] = 0.0;
      sc[part] = pp[part][1];
      double el[2] = {0.0, 1.0 / sc[part]};
      v_R_inv_vec(lth, N, s, 1, NULL, 1, el, qq[part], ws, 2);
    }
  }
  const double full = sc[0] + sc[1];
  R->lp = log(full);
  if (!with_grad) return;
  /* adjoint through the 2 x 2 joint system (one_event.py:116-137, 307-343) */
  double qin[2] = {0, 0};
  for (int part = 0; part < 2; ++part) if (use[part]) qin[1] += qq[part][1] * (part == 0 ? dpl : dml) * sc[part] / full;
  double qj[2]; qj[1] = qin[1] / r11; qj[0] = (qin[0] - qj[1] * r10) / r00;
  for (int part = 0; part < 2; ++part) {
    if (!use[part]) continue;
    const double wgt = sc[part] / full;
    for (int e = 0; e < N * N; ++e) R->g[e] += wgt * g1[part][e];
    if (part == 0) { for (int i = 0; i < N; ++i) R->dm[i] += wgt * dd[0][i]; R->dp[n] += wgt * qq[0][1] * dpl * pth1[1]; }
    else { for (int i = 0; i < N; ++i) R->dp[i] += wgt * dd[1][i]; R->dm[n] += wgt * qq[1][1] * dml * pth1[1]; }
  }
  for (int i = 0; i < n; ++i) {                                    /* one_event.x_partial_Q_y, :88-113 */
    const double tii = exp(P->lt[i * N + i]), tiM = exp(P->lt[i * N + n]);
    R->g[i * N + i] += -tii * (qj[0] * pth1[0] + qj[1] * (1.0 + tiM) * pth1[1]);
    R->g[i * N + n] += qj[1] * (-tii * tiM) * pth1[1];
  }
  R->g[n * N + n] += exp(P->lt[n * N + n]) * pth1[0] * (qj[1] - qj[0]);   /* row n: only the final assignment survives */
  R->dm[n] -= qj[1] * dml * pth1[1];
  R->dp[n] -= qj[1] * dpl * pth1[1];
}

static void coupled_patient(const par_t* P, const int8_t* row, int order, int with_grad, res_t* R) {
  const int n = P->n, N = n + 1;
  const int8_t* st = row;
  int n_prim = 0, n_met = 1;
  for (int j = 0; j < n; ++j) { n_prim += st[2 * j]; n_met += st[2 * j + 1]; }
  n_prim += st[2 * n];
  const int k = n_prim + n_met - 1;
  if (k == 1) { one_event_patient(P, row, order, with_grad, R); return; }
  const size_t len = (size_t)1 << k;
  double* ws = (double*)malloc(sizeof(double) * len * 16);
  double *pi = ws, *scal = ws + len, *rhsJ = ws + 2 * len, *qJ = ws + 3 * len, *w = ws + 4 * len;
  memset(scal, 0, len * sizeof(double)); scal[0] = 1.0;
  R_i_inv_vec(P, st, k, 0, scal, pi, w, len);
  int8_t sst[2][64];
  for (int j = 0; j < n; ++j) { sst[0][j] = st[2 * j + 1]; sst[1][j] = st[2 * j]; }
  sst[0][n] = 1; sst[1][n] = st[2 * n];
  const int ks[2] = {n_met, n_prim};
  const int use[2] = {order == 0 || order == 1, order != 1};
  double lth[2][32 * 32];
  diagnosis_theta(P->lt, P->ldm, N, 0, lth[0]);
  diagnosis_theta(P->lt, P->ldp, N, 1, lth[1]);
  double sc[2] = {0, 0};
  double* g1[2] = {NULL, NULL};
  double dd[2][32];
  int64_t* idx[2] = {NULL, NULL};
  double* qm[2] = {NULL, NULL};
  double* vm[2] = {NULL, NULL};
  for (int part = 0; part < 2; ++part) {
    if (!use[part]) continue;
    const size_t ls = (size_t)1 << ks[part], half = ls >> 1;
    diag_scal(P, part, -1, st, pi, scal, w, len);                  /* D_p pi or D_m pi */
    idx[part] = (int64_t*)malloc(sizeof(int64_t) * len);
    obs_indices(P, st, part == 0, len, w, w + len, idx[part]);
    double* sw = (double*)malloc(sizeof(double) * ls * 10);
    double *v = sw, *pth2 = sw + ls, *xa = sw + 2 * ls, *wv = sw + 3 * ls;
    memset(v, 0, ls * sizeof(double));
    for (size_t e = 0; e < half; ++e) v[half + e] = scal[idx[part][e]];
    if (!with_grad) { v_R_inv_vec(lth[part], N, sst[part], ks[part], NULL, 0, v, pth2, wv, ls); sc[part] = pth2[ls - 1]; free(sw); }
    else {
      g1[part] = (double*)calloc((size_t)N * N, sizeof(double));
      v_gradient(lth[part], N, sst[part], ks[part], v, g1[part], dd[part], pth2, xa, wv, ls);
      if (part == 1) for (int i = 0; i < n; ++i) g1[1][i * N + n] = 0.0;
      sc[part] = pth2[ls - 1];
      qm[part] = (double*)malloc(sizeof(double) * ls);
      double* el = wv + 4 * ls;
      memset(el, 0, ls * sizeof(double)); el[ls - 1] = 1.0 / sc[part];
      v_R_inv_vec(lth[part], N, sst[part], ks[part], NULL, 1, el, qm[part], wv, ls);   /* likelihood.py:570-572 */
      vm[part] = sw;                                               /* keep v alive (first ls entries) */
    }
  }
  const double full = sc[0] + sc[1];
  R->lp = log(full);
  if (with_grad) {
    memset(rhsJ, 0, len * sizeof(double));
    for (int part = 0; part < 2; ++part) {
      if (!use[part]) continue;
      const size_t ls = (size_t)1 << ks[part], half = ls >> 1;
      const double wgt = sc[part] / full;
      /* p = scatter(q[half:]) ; direct term x_partial_D_y(p, pi) ; rhs += D (x) p * weight */
      memset(scal, 0, len * sizeof(double));
      for (size_t e = 0; e < half; ++e) scal[idx[part][e]] = qm[part][half + e];
      double a[32], b[32];
      x_partial_D_y(P, st, scal, pi, a, b, w, len);
      for (int e = 0; e < N * N; ++e) R->g[e] += wgt * g1[part][e];
      if (part == 0) for (int i = 0; i < N; ++i) { R->dp[i] += wgt * a[i]; R->dm[i] += wgt * dd[0][i]; }
      else for (int i = 0; i < N; ++i) { R->dm[i] += wgt * b[i]; R->dp[i] += wgt * dd[1][i]; }
      diag_scal(P, part, -1, st, scal, qJ, w, len);
      for (size_t e = 0; e < len; ++e) rhsJ[e] += wgt * qJ[e];
    }
    R_i_inv_vec(P, st, k, 1, rhsJ, qJ, w, len);                    /* q_inv_deriv_pth, likelihood.py:516-537 */
    double* G2 = (double*)malloc(sizeof(double) * N * N);
    x_partial_Q_y(P, st, qJ, pi, G2, w, len);
    for (int e = 0; e < N * N; ++e) R->g[e] += G2[e];
    double a[32], b[32];
    x_partial_D_y(P, st, qJ, pi, a, b, w, len);
    for (int i = 0; i < N; ++i) { R->dp[i] -= a[i]; R->dm[i] -= b[i]; }
    free(G2);
  }
  for (int part = 0; part < 2; ++part) { free(idx[part]); free(g1[part]); free(qm[part]); free(vm[part]); }
  free(ws);
}

/* ======================================================================================
 * exported entry points (ctypes)
 * ==================================================================================== */

/* per-patient results: lp[n_pat]; if with_grad also g[n_pat][N*N], dp[n_pat][N], dm[n_pat][N] (pre-zeroed here) */
int ref_patients(int n, const double* lt, const double* ldp, const double* ldm, const int8_t* dat, int64_t n_pat,
                 int with_grad, int n_threads, int patient_parallel, double* lp, double* g, double* dp, double* dm) {
  const int N = n + 1, nc = 2 * n + 3;
  if (n < 1 || n > 31) return 1;
  par_t P = {n, lt, ldp, ldm};
  if (with_grad) {
    memset(g, 0, sizeof(double) * (size_t)n_pat * N * N);
    memset(dp, 0, sizeof(double) * (size_t)n_pat * N);
    memset(dm, 0, sizeof(double) * (size_t)n_pat * N);
  }
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1) if (patient_parallel)
  for (int64_t r = 0; r < n_pat; ++r) {
    const int8_t* row = dat + r * nc;
    const int type = row[nc - 1];
    int order = row[nc - 2];
    if (order != 0 && order != 1) order = 2;                       /* regularized_optimization.py:114,245 */
    res_t R = {0.0, with_grad ? g + (size_t)r * N * N : NULL, with_grad ? dp + (size_t)r * N : NULL,
               with_grad ? dm + (size_t)r * N : NULL};
    if (type == 3) coupled_patient(&P, row, order, with_grad, &R);
    else single_patient(&P, row, type, with_grad, &R);
    lp[r] = R.lp;
  }
  return 0;
}

int ref_kronvec(int n, const double* lt, const int8_t* state, const double* p, double* y, int diag, int transpose) {
  par_t P = {n, lt, NULL, NULL};
  int k = 0; for (int j = 0; j < 2 * n + 1; ++j) k += state[j];
  const size_t len = (size_t)1 << k;
  double* ws = (double*)malloc(sizeof(double) * len * 2);
  kronvec(&P, state, diag, transpose, p, y, ws, ws + len, len);
  free(ws);
  return 0;
}

int ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
