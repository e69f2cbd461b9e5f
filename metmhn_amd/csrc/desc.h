// Bit-role descriptors of restricted state spaces and per-evaluation parameter sets.
//
// A "problem" is one restricted space of 2^k states: either the joint PT/MT space of a
// paired patient (metmhn/jx/kronvec.py:223-250 layout: active slots of
// [PT_0,MT_0,...,PT_{n-1},MT_{n-1},seed] in that order are index bits 0..k-1) or a
// single-tumour space (metmhn/jx/vanilla.py: active events of a length n+1 state).
// Everything the kernels need to know about the space is in `Desc` (plain data, same
// struct on host and device).
#pragma once
#include <stdint.h>

namespace mmhn {

constexpr int MAXN = 32;   // events incl. seeding (n_mut <= 31)
constexpr int MAXK = 30;   // index bits of one restricted space

enum Mode { JOINT = 0, SINGLE = 1 };
enum Cls { CP = 0, CM = 1, CS = 2 };
// observation diagonal added to -diag(Q) when a resolvent is formed
enum Obs { OBS_JOINT = 0,   // D_p + D_m            (likelihood.py:249-250)
           OBS_ONE = 1,     // identity             (vanilla.py:294-295, d_rates = 1)
           OBS_MET = 2,     // scal_d_pt d_p + d_m  (vanilla.py:125-142, likelihood.py:435-436)
           OBS_VEC = 3 };   // caller-supplied d_rates vector (API only)
// parameter sets prepared per evaluation (see build_params)
enum PSet { PS_THETA = 0,   // exp(log_theta)
            PS_MET = 1,     // diagnosis_theta(log_theta, log_d_m)            (kronvec.py:7-21, likelihood.py:309)
            PS_PRIM = 2,    // diagnosis_theta(theta with [:n, n] <- 0, log_d_p) (likelihood.py:313-314)
            NPSET = 3 };

struct Desc {
  int k;               // index bits
  int mode;            // Mode
  int seedbit;         // bit of the seeding event, -1 if inactive
  int N;               // events incl. seeding
  uint32_t maskP;      // class-P bits (SINGLE: every bit)
  uint32_t maskM;      // class-M bits
  uint32_t pairP;      // P bits of events active in both tumours (partner M bit = bit + 1)
  uint32_t lone;       // P/M bits whose partner slot is inactive
  int pset;            // PSet used for transition rates
  int obs;             // Obs
  long long off;       // element offset of this problem's state vectors in the batch buffers
  long long aoff;      // element offset of its gradient work arrays
  long long toff;      // element offset of its per-evaluation tables (k_prep)
  int wl;              // >= 0: the seeded half of its vectors is stored in the window layout (wlayout.h), index of its WDesc
  int pad_;
  int8_t ev[32];       // event of bit b
  int8_t cls[32];      // Cls of bit b
  int8_t bitP[32];     // event -> class-P bit or -1
  int8_t bitM[32];     // event -> class-M bit or -1
};

// One evaluation's parameters, one instance per PSet.  T = engine dtype.
template <typename T>
struct Params {
  T th[MAXN][MAXN];    // multiplicative effects theta_ij (row i: effects ON event i)
  T baseP[MAXN];       // base rate of event i as a class-P transition (theta_ii; seeding: theta_nn)
  T baseM[MAXN];       // class-M base rate theta_ii * theta_in (JOINT only)
  T dp[MAXN];          // exp(log_d_p)
  T dm[MAXN];          // exp(log_d_m)
};

// joint problem -> its marginal problems (right-hand side of the joint adjoint, likelihood.py:573-575, 617-618):
// rhs_J[x] = sum over parts of cst[part] * q_S[part][upper half][pext(x, free mask)] on the compatible states
template <typename T>
struct JLink {
  long long soff[2];   // offset of the marginal problem's vectors, -1 if the part is absent
  int sk[2];           // its number of bits
  T cst[2];            // D_p (part 0) / D_m (part 1) on the compatible states (constant there)
};

// one patient of a batch: which problems belong to it and how their results combine
struct PatRec {
  int kind;            // dat type 0..3; 4 = all-zero type-0 row (closed form)
  int order;           // 0 / 1 / 2 (anything else -> 2, regularized_optimization.py:114,245)
  int j;               // joint problem index in the batch or -1
  int s[2];            // single problems: [0] = PT-first part (MT marginal) or the patient itself, [1] = MT-first part
  int row;             // row of `dat`
};

inline int popc(uint32_t v) { return __builtin_popcount(v); }

// joint space of `state` (length 2n+1)
inline Desc make_joint(const int8_t* state, int n) {
  Desc d{};
  d.mode = JOINT; d.N = n + 1; d.seedbit = -1; d.pset = PS_THETA; d.obs = OBS_JOINT; d.wl = -1;
  for (int i = 0; i < 32; ++i) { d.bitP[i] = -1; d.bitM[i] = -1; d.ev[i] = 0; d.cls[i] = 0; }
  int k = 0;
  for (int j = 0; j < n; ++j) {
    const bool p = state[2 * j] != 0, m = state[2 * j + 1] != 0;
    if (p) { d.ev[k] = (int8_t)j; d.cls[k] = CP; d.bitP[j] = (int8_t)k; d.maskP |= 1u << k;
             if (m) d.pairP |= 1u << k; else d.lone |= 1u << k; ++k; }
    if (m) { d.ev[k] = (int8_t)j; d.cls[k] = CM; d.bitM[j] = (int8_t)k; d.maskM |= 1u << k;
             if (!p) d.lone |= 1u << k; ++k; }
  }
  if (state[2 * n]) { d.ev[k] = (int8_t)n; d.cls[k] = CS; d.seedbit = k; ++k; }
  d.k = k;
  return d;
}

// single-tumour space of `state` (length n+1, seeding last)
inline Desc make_single(const int8_t* state, int n, int pset, int obs) {
  Desc d{};
  d.mode = SINGLE; d.N = n + 1; d.seedbit = -1; d.pset = pset; d.obs = obs; d.wl = -1;
  for (int i = 0; i < 32; ++i) { d.bitP[i] = -1; d.bitM[i] = -1; d.ev[i] = 0; d.cls[i] = 0; }
  int k = 0;
  for (int j = 0; j <= n; ++j)
    if (state[j]) { d.ev[k] = (int8_t)j; d.cls[k] = CP; d.bitP[j] = (int8_t)k; d.maskP |= 1u << k;
                    if (j == n) d.seedbit = k; ++k; }
  d.k = k;
  return d;
}

}  // namespace mmhn
