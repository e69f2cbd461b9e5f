// k_diag: diagonal quantities of a tile (diag Q, 1/(D - diag Q), D_p p, D_m p, ...).  Reference: kronvec.py:574-710, 964-999.
#pragma once
#include "common.h"

namespace mmhn {

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

// KB: threads of a workgroup (1 024 on long launches: three workgroups per CU hold the LDS either way - 48 instead of 12 waves)
template <typename T, int KB = BLOCK>
__global__ __launch_bounds__(KB) void k_diag(const Desc* __restrict__ descs,
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

  for (int e = tid; e < N * 64; e += KB) {
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
  for (int r = wave; r < R; r += KB / 64) {
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

}  // namespace mmhn
