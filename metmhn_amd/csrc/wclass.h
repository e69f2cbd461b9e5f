// Class marginals of p (x) q for joint problems stored in the window layout (wlayout.h) - what k_pclass / k_class_marg
// compute for index-order vectors (kernels.h; reference: the reductions of x_partial_Q_y, likelihood.py:25-201, in the
// class-marginal form of DESIGN.md 3.2):
//   slot 0      W[S]   = - sum_T p[S,T] q[S,T]
//   slot 1 + l  V_l[S] =   sum_T p[S,T] q[S | bit_l, T]      (bit_l not in S)
// for both classes.  The seeded half is a matrix rows x columns (row class = the class with more bits); the layout
// keeps 1024 rows x NC columns ("a block") and, across blocks, RCH storage rows x every column contiguous in runs of
// RCH * 32 bytes.  One 1024-thread workgroup per patient, two phases, each reads p and q once:
//   phase R (row class):    thread = row.  Per block the rows' q values go through LDS (every thread-bit neighbour of a
//                           row is another row of the same block), the sums over the columns stay in registers for the
//                           whole patient and every table entry is written once - no atomics, no cross-lane reduction.
//                           Neighbours along the row bits beyond the tenth are other blocks: one extra read of q each.
//   phase C (column class): thread = 16-byte piece of columns x some rows of a chunk of RCH rows x all columns staged in
//                           LDS (every column neighbour is in the chunk); sums over the rows in registers, one small
//                           reduction over the threads that share a piece at the end.
#pragma once
#include "kernels.h"

namespace mmhn {

template <typename T>
struct WClass {
  static constexpr int CHE = 32768 / (int)sizeof(T);          // elements of one staged array (a block, or a chunk of rows)
  static constexpr size_t lds = 4 * (size_t)CHE * sizeof(T) + 64;
};
template <typename T>
constexpr size_t wclass_lds() { return WClass<T>::lds; }

template <typename T>
__global__ __launch_bounds__(WROWS) void k_wclass(const Desc* __restrict__ descs, const WDesc* __restrict__ wds, int nw,
                                                  const T* __restrict__ p, const T* __restrict__ q, T* A) {
  using C = WCfg<T>;
  constexpr int RB = C::RB, HB = C::HB, NC = 1 << RB, H = 1 << HB, KC = C::KC, WNXR = C::KR - WTB;
  constexpr int QE = 16 / (int)sizeof(T), LGQ = QE == 2 ? 1 : 2, NQ = NC / QE;
  constexpr int CHE = WClass<T>::CHE;
  typedef T VecT __attribute__((ext_vector_type(NC)));
  typedef T QT __attribute__((ext_vector_type(QE)));
  static_assert(NQ == 2 && CHE == WROWS * NC, "a block row is two 16-byte pieces; a block fills one staged array");
  extern __shared__ __align__(16) unsigned char smem[];
  T* const lds = reinterpret_cast<T*>(smem);                 // four staged arrays of CHE elements
  const int tid = threadIdx.x;
  const uint32_t rho = wrho((uint32_t)tid >> 6, (uint32_t)tid & 63u);
  for (int it = blockIdx.x; it < nw; it += gridDim.x) {
    const WDesc& wd = wds[it];
    const Desc& d = descs[wd.prob];
    const int k = d.k, kR = wd.kR, kC = wd.kC, nXc = wd.nXc, nXr = wd.nXr;
    const bool majP = wd.majP != 0;
    const long long half = 1ll << (k - 1);
    const T* ps = p + d.off + half;
    const T* qs = q + d.off + half;
    const int kP = __popc(d.maskP);
    T* outR = A + d.aoff + (majP ? 0 : class_block_size(kP));
    T* outC = A + d.aoff + (majP ? class_block_size(kP) : 0);
    const int NBc = (1 << nXc) << HB;                         // blocks of one external row setting
    auto ldv = [&](const T* src, uint32_t Sigma, uint32_t beta, uint32_t row) -> VecT {
      const QT* a = reinterpret_cast<const QT*>(src + wpos<T>(Sigma, beta, row, 0));
      const QT lo = a[0], hi = a[1];
      VecT v;
#pragma unroll
      for (int e = 0; e < QE; ++e) { v[e] = lo[e]; v[QE + e] = hi[e]; }
      return v;
    };
    auto dot = [&](const VecT& a, const VecT& b) -> T {
      T s = a[0] * b[0];
#pragma unroll
      for (int c = 1; c < NC; ++c) s += a[c] * b[c];
      return s;
    };
    // =============================== phase R: thread = row
    __syncthreads();                                           // the previous problem is done with the staged arrays
    for (uint32_t Sx = 0; Sx < (1u << nXr); ++Sx) {
      T acc[1 + WTB + WNXR];
#pragma unroll
      for (int s = 0; s < 1 + WTB + WNXR; ++s) acc[s] = T(0);
      VecT pn = ldv(ps, Sx << nXc, 0, rho), qn = ldv(qs, Sx << nXc, 0, rho);
      for (int B = 0; B < NBc; ++B) {
        const uint32_t Sigma = ((uint32_t)B >> HB) | (Sx << nXc), beta = (uint32_t)B & (uint32_t)(H - 1);
        const VecT pv = pn, qv = qn;
        if (B + 1 < NBc) {
          const uint32_t Sg2 = ((uint32_t)(B + 1) >> HB) | (Sx << nXc), b2 = (uint32_t)(B + 1) & (uint32_t)(H - 1);
          pn = ldv(ps, Sg2, b2, rho); qn = ldv(qs, Sg2, b2, rho);
        }
        T* qb = lds + (B & 1) * CHE;                          // [piece][row]: conflict-free 16-byte accesses
        {
          QT lo, hi;
#pragma unroll
          for (int e = 0; e < QE; ++e) { lo[e] = qv[e]; hi[e] = qv[QE + e]; }
          *reinterpret_cast<QT*>(qb + (size_t)tid * QE) = lo;
          *reinterpret_cast<QT*>(qb + (size_t)(WROWS + tid) * QE) = hi;
        }
        __syncthreads();
        acc[0] -= dot(pv, qv);
#pragma unroll
        for (int i = 0; i < WTB; ++i) {
          const uint32_t nb = (uint32_t)tid | (1u << i);
          const QT lo = *reinterpret_cast<const QT*>(qb + (size_t)nb * QE);
          const QT hi = *reinterpret_cast<const QT*>(qb + (size_t)(WROWS + nb) * QE);
          T s = T(0);
#pragma unroll
          for (int e = 0; e < QE; ++e) s += pv[e] * lo[e] + pv[QE + e] * hi[e];
          acc[1 + i] += s;                                     // (a row that has bit i never writes this slot: no select)
        }
#pragma unroll
        for (int i = 0; i < WNXR; ++i) {
          if (i < nXr && !((Sx >> i) & 1u)) {                  // (uniform)
            const VecT qx = ldv(qs, Sigma | (1u << (nXc + i)), beta, rho);
            acc[1 + WTB + i] += dot(pv, qx);
          }
        }
      }
      const long long S = (long long)tid | ((long long)Sx << WTB);
      outR[S] = acc[0];
#pragma unroll
      for (int i = 0; i < WTB; ++i)
        if (!((tid >> i) & 1)) outR[((long long)(1 + i) << kR) + S] = acc[1 + i];
#pragma unroll
      for (int i = 0; i < WNXR; ++i)
        if (i < nXr && !((Sx >> i) & 1u)) outR[((long long)(1 + WTB + i) << kR) + S] = acc[1 + WTB + i];
    }
    // =============================== phase C: thread = piece of QE columns, rows of a chunk dealt over the threads
    {
      const int NPc = (1 << kC) / QE;                          // pieces of a row
      const int RCH = CHE >> kC;                               // rows of a chunk
      const int nch = WROWS / RCH;
      const int cp = tid & (NPc - 1);                          // (NPc divides 1024)
      const int r0 = tid / NPc, rstep = WROWS / NPc;           // rows r0, r0 + rstep, ... of a chunk: CHE / QE / 1024 = 2 items
      T ac[QE][1 + KC];
#pragma unroll
      for (int e = 0; e < QE; ++e)
#pragma unroll
        for (int s = 0; s <= KC; ++s) ac[e][s] = T(0);
      // piece g of a chunk (both arrays): run = block (Tx, beta), inside the run row-major
      const int rl = RCH * NQ;                                 // pieces of a run
      auto piece_src = [&](int g, uint32_t Sx, int ch) -> long long {
        const int run = g / rl, rr = g % rl;
        const uint32_t Sigma = ((uint32_t)run >> HB) | (Sx << nXc), beta = (uint32_t)run & (uint32_t)(H - 1);
        return wpos<T>(Sigma, beta, (uint32_t)(ch * RCH + (rr >> 1)), (uint32_t)((rr & 1) * QE));
      };
      auto piece_dst = [&](int g) -> int {
        const int run = g / rl, rr = g % rl;
        return ((rr >> 1) << kC) + (run << RB) + (rr & 1) * QE;
      };
      const int ntot = nch << nXr;                             // chunks of the patient
      QT rp[2], rq[2];
      auto fetch = [&](int cidx) {
        const uint32_t Sx = (uint32_t)(cidx / nch);
        const int ch = cidx % nch;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const long long src = piece_src(tid + WROWS * j, Sx, ch);
          rp[j] = *reinterpret_cast<const QT*>(ps + src);
          rq[j] = *reinterpret_cast<const QT*>(qs + src);
        }
      };
      __syncthreads();                                         // phase R is done with the staged arrays
      fetch(0);
      for (int cidx = 0; cidx < ntot; ++cidx) {
        T* pc = lds + (size_t)(2 * (cidx & 1)) * CHE;
        T* qc = pc + CHE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int dst = piece_dst(tid + WROWS * j);
          *reinterpret_cast<QT*>(pc + dst) = rp[j];
          *reinterpret_cast<QT*>(qc + dst) = rq[j];
        }
        if (cidx + 1 < ntot) fetch(cidx + 1);
        __syncthreads();
        for (int r = r0; r < RCH; r += rstep) {
          const T* prow = pc + ((size_t)r << kC);
          const T* qrow = qc + ((size_t)r << kC);
          const QT pv = *reinterpret_cast<const QT*>(prow + cp * QE);
          const QT qv = *reinterpret_cast<const QT*>(qrow + cp * QE);
#pragma unroll
          for (int e = 0; e < QE; ++e) ac[e][0] -= pv[e] * qv[e];
#pragma unroll
          for (int b = 0; b < LGQ; ++b)
#pragma unroll
            for (int e = 0; e < QE; ++e)
              if (!((e >> b) & 1)) ac[e][1 + b] += pv[e] * qv[e | (1 << b)];
#pragma unroll
          for (int b = LGQ; b < KC; ++b) {
            if (b < kC) {
              // (a column that has bit b never writes this slot: the sum it collects here is dropped, no select)
              const QT nq = *reinterpret_cast<const QT*>(qrow + (cp | (1 << (b - LGQ))) * QE);
#pragma unroll
              for (int e = 0; e < QE; ++e) ac[e][1 + b] += pv[e] * nq[e];
            }
          }
        }
      }
      // ---- sum over the threads that share a piece, write the tables
      __syncthreads();
      T* red = lds;
#pragma unroll
      for (int s = 0; s <= KC; ++s) {
#pragma unroll
        for (int e = 0; e < QE; ++e) {
          if (s <= kC) {                                       // (uniform; no early exit: the accumulators keep static indices)
            red[tid] = ac[e][s];
            __syncthreads();
            if (tid < NPc) {
              T v = T(0);
              for (int g = 0; g < rstep; ++g) v += red[tid + g * NPc];
              const uint32_t Tc = (uint32_t)(tid * QE + e);
              if (s == 0 || !((Tc >> (s - 1)) & 1u)) outC[((long long)s << kC) + Tc] = v;
            }
            __syncthreads();
          }
        }
      }
    }
  }
}

}  // namespace mmhn
