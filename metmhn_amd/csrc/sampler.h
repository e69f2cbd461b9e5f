// Gillespie sampler of the joint PT/MT process (metmhn/simulations.py:8-147).
//
// One thread simulates one patient: both tumours evolve together until seeding, independently afterwards, until
// both are diagnosed or the primary is diagnosed before seeding (`stop_fun`, simulations.py:58-62).  The state is
// two bit sets (events 0..N-2 mutations, N-1 seeding, N diagnosis) and the event rates are recomputed from the
// log-parameters in LDS every step exactly as the reference does (exp of a sum of logs, `tumor_dynamics`
// :28-44); the next event is drawn by inverting the cumulative rates with one uniform per step.
// Random numbers: Philox4x32-10 keyed by the caller's seed, counter = (trajectory, step): reproducible for a
// given (seed, n_sim), independent of the launch geometry.  The stream differs from jax.random's threefry, the
// distribution does not (tests/test_montecarlo.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmhn {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

constexpr int SIM_BLOCK = 256;
constexpr int SIM_MAXN = 32;                     // events incl. seeding

// log_theta [N][N] row-major (row i: effects ON event i), pt_d / mt_d [N].
// dat_out  [n_sim][2(N-1)+2] = [PT_0, MT_0, ..., PT_{N-2}, MT_{N-2}, seeding, order]   (simulate_dat, :117-147)
// ord_out  [n_sim][2N+2]     event sequence padded with -99 (simulate_orders, :87-114); may be null
__global__ __launch_bounds__(SIM_BLOCK) void k_gillespie(const double* __restrict__ log_theta,
                                                         const double* __restrict__ pt_d,
                                                         const double* __restrict__ mt_d, int N, long long n_sim,
                                                         uint64_t seed, int8_t* __restrict__ dat_out,
                                                         int8_t* __restrict__ ord_out) {
  __shared__ double lt[SIM_MAXN * SIM_MAXN], ltp[SIM_MAXN * SIM_MAXN], dp[SIM_MAXN], dm[SIM_MAXN];
  for (int e = threadIdx.x; e < N * N; e += SIM_BLOCK) {
    const int i = e / N, j = e % N;
    const double v = log_theta[e];
    lt[e] = v;
    ltp[e] = (j == N - 1 && i < N - 1) ? 0.0 : v;        // seeding does not act on the PT's mutations (:67-68)
  }
  for (int e = threadIdx.x; e < N; e += SIM_BLOCK) { dp[e] = pt_d[e]; dm[e] = mt_d[e]; }
  __syncthreads();
  const long long id = (long long)blockIdx.x * SIM_BLOCK + threadIdx.x;
  if (id >= n_sim) return;
  const uint32_t sbit = 1u << (N - 1), dbit = 1u << N, evmask = dbit - 1u;
  uint32_t pt = 0, mt = 0;                     // bits 0..N-1 events (N-1 = seeding), bit N = diagnosed
  int t_pt = -1, t_mt = -1;
  const int L = 2 * N + 2;
  int8_t* ord = ord_out ? ord_out + id * L : nullptr;
  if (ord) for (int e = 0; e < L; ++e) ord[e] = -99;
  for (int step = 0; step < L; ++step) {
    if ((pt & dbit) && ((mt & dbit) || !(pt & sbit))) break;
    const bool pt_on = !(pt & dbit);
    const bool mt_on = (pt & sbit) && !(mt & dbit);
    // event e of tumour T: 0..N-1 mutations / seeding, N diagnosis; index in the reference's vector: e (+ N+1 for MT)
    auto rate = [&](int tum, int e) -> double {
      const uint32_t st = tum == 0 ? pt : mt;
      if (tum == 0 ? !pt_on : !mt_on) return 0.0;
      if ((st >> e) & 1u) return 0.0;
      double s = 0.0;
      if (e < N) {
        const double* row = (tum == 0 ? ltp : lt) + e * N;
        s = row[e];                                           // b_rates = diag(log_theta)
        for (uint32_t m = st & evmask; m; m &= m - 1) s += row[__ffs(m) - 1];
      } else {
        const double* dv = tum == 0 ? dp : dm;
        for (uint32_t m = st & evmask; m; m &= m - 1) s += dv[__ffs(m) - 1];
      }
      return exp(s);
    };
    double total = 0.0;
    for (int tum = 0; tum < 2; ++tum)
      for (int e = 0; e <= N; ++e) total += rate(tum, e);
    uint32_t r[4];
    philox4x32_10((uint32_t)id, (uint32_t)((uint64_t)id >> 32), (uint32_t)step, 0u, (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    const double u = ((double)(((uint64_t)(r[0] >> 5) << 26) | (uint64_t)(r[1] >> 6)) * (1.0 / 9007199254740992.0)) * total;
    int ev_t = 0, ev_e = 0;
    {
      double cum = 0.0;
      bool found = false;
      int last_t = 0, last_e = 0;
      for (int tum = 0; tum < 2 && !found; ++tum)
        for (int e = 0; e <= N; ++e) {
          const double rr = rate(tum, e);
          if (rr > 0.0) { last_t = tum; last_e = e; }
          cum += rr;
          if (cum > u) { ev_t = tum; ev_e = e; found = true; break; }   // first index with cumulative rate > u
        }
      if (!found) { ev_t = last_t; ev_e = last_e; }             // rounding at the upper end
    }
    const bool seeded = pt & sbit;
    if (ev_t == 0) {
      pt |= 1u << ev_e;
      if (!seeded) mt |= 1u << ev_e;                            // before seeding both tumours move together (:49-52)
      if (ev_e == N) { t_pt = step; if (!seeded) t_mt = step; }
    } else {
      mt |= 1u << ev_e;
      if (ev_e == N) t_mt = step;
    }
    if (ord) ord[step] = (int8_t)(ev_t == 0 ? ev_e : ev_e + N + 1);
  }
  const int n_mut = N - 1, W = 2 * n_mut + 2;
  int8_t* o = dat_out + id * W;
  for (int j = 0; j < n_mut; ++j) { o[2 * j] = (pt >> j) & 1u; o[2 * j + 1] = (mt >> j) & 1u; }
  const bool paired = pt & sbit;
  o[2 * n_mut] = paired ? 1 : 0;
  o[2 * n_mut + 1] = paired ? (t_pt < t_mt ? 1 : 2) : 0;
}

}  // namespace mmhn
