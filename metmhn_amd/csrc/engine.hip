// Host side of the metMHN engine: cohort layout, batch scheduling, kernel launches and the
// C ABI of include/metmhn_amd.h.  One engine = one GPU, one HIP stream.
//
// Pipeline of one evaluation (reference call graph: regularized_optimization.py:163-267 ->
// likelihood.py:_g_coupled_*, _grad_prim_obs, _grad_met_obs), run batch by batch with every
// patient of the batch in flight at once:
//   1  lidg_J = 1/(D_p + D_m - diag Q)                       k_diag(KD_LIDG)
//   2  pi    = (D - Q)^-1 e_0          k+1 fused Jacobi sweeps  k_sweep<false>
//   3  v     = D_obs * pi[compatible]  -> marginal right-hand sides  k_gather_marg
//   4  single-tumour spaces (marginals of paired patients and the unpaired patients):
//      lidg_S, forward solve, scores -> adjoint seeds 1/score, adjoint solve,
//      flow gradient                  k_diag, k_sweep, k_seeds, k_sweep<true>, k_grad_rows
//   5  rhs_J = D_obs * scatter(q_S), q_J = (D - Q)^-T rhs_J  k_scatter_marg, k_sweep<true>
//   6  joint gradient: class marginals, eq block, flow rows, observation-rate marginals
//                                      k_class_marg, k_eq_flows, k_grad_rows, k_bit_marg
//   7  per-patient assembly and deterministic cohort reduction  k_finalize, k_reduce_rows, k_reduce_parts
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>                 // types only: the library is opened at run time by mmhn_comm_init
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/metmhn_amd.h"
#include "kernels.h"
#include "tsolve.h"
#include "small.h"
#include "wsolve.h"
#include "wclass.h"
#include "sampler.h"

namespace mmhn {

static thread_local std::string g_err;

struct Fail {
  std::string msg;
};
#define HIPCHECK(expr)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      throw Fail{std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + \
                 std::to_string(__LINE__) + ")"};                                            \
  } while (0)
#define REQUIRE(cond, text) \
  do {                      \
    if (!(cond)) throw Fail{std::string(text)}; \
  } while (0)

template <typename U>
struct DevArr {
  U* p = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    if (count <= n && p) return;
    release();
    if (count == 0) return;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(U)));
    n = count;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevArr() { release(); }
  DevArr() = default;
  DevArr(const DevArr&) = delete;
  DevArr& operator=(const DevArr&) = delete;
  DevArr(DevArr&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevArr& operator=(DevArr&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
};

// work list of one cooperative solve launch (tsolve.h: k_csolve), host + device copy
struct CList {
  std::vector<CItem> items;
  std::vector<int> deps;
  DevArr<CItem> d_items;
  DevArr<int> d_deps;
  int maxk = 0;
  int nlevels = 0;               // levels of the deepest problem (1: no tile waits for another one)
  void clear() { items.clear(); deps.clear(); maxk = 0; nlevels = 0; }
};
enum Route : uint8_t { RT_W = 0, RT_P = 1, RT_T = 2 };

// the lists of the staged single-tumour kernels for one group of patients of a batch
struct Staged {
  std::vector<int> pats, paired, probs;      // patients; those of them with a joint problem; their single-tumour problems
  DevArr<int> d_pats, d_paired, d_probs;
  std::vector<int2> map, lmap;               // tiles of the problems (map order / by level)
  std::vector<int> lof;
  DevArr<int2> d_map, d_lmap;
  DevArr<int2> d_grc;                        // k_grad_rows work list (GK_S)
  CList cl[2];                               // cooperative work lists, forward / transposed
  int maxk = 0;
  bool kind2 = false;                        // a patient of the group is an MT-only row
  bool empty() const { return pats.empty(); }
};

// static (per cohort) description of one batch of patients
struct Batch {
  std::vector<PatRec> pats;
  std::vector<Desc> dJ, dS;
  std::vector<int2> mapJ, mapS;
  long long vecJ = 0, vecS = 0, asize = 0, tabJ = 0, tabS = 0;
  int id = 0;
  int maxkJ = 0, maxkS = 0;
  int maxkcJ = 0;                // most bits of one class (PT / MT) over the joint problems
  bool has_kind2 = false;
  // tiles sorted by level = popcount(tile index) for the substitution solver; lof* = level offsets
  std::vector<int2> lmapJ, lmapS;
  std::vector<int> lofJ, lofS;
  // live tiles of every joint problem in index order (k_psolve: one workgroup per patient)
  // tiles of the joint problems with more than PCA + PCH bits in one class: k_pclass skips them, k_class_marg
  // takes them
  std::vector<int2> mapX;
  DevArr<int2> d_mapX;
  DevArr<int2> d_grc[4];         // k_grad_rows work lists per kind (GK_P, GK_M, GK_E: joint problems, GK_S: single)
  DevArr<int2> d_grcJ;           // the three joint kinds in one list (kind in bits 24+ of .y): one launch
  // small-space path (small.h): patients whose single-tumour spaces all fit a tile, by size class of the largest one
  // [0]: patients that are their own problem (dat types 0-2: nothing of the joint path feeds them), [1]: paired rows (their
  // single-tumour problems are marginals of the joint forward solution), [2]: the paired rows with both marginal problems
  // whose class has a side-by-side kernel (small.h PAIR) - taken out of [1]
  std::vector<int> sp_list[3][SP_NCLASS];
  DevArr<int> d_sp_list[3][SP_NCLASS];
  bool has_small = false;        // some patient takes the small-space path
  int mk1p = 9;                  // bits the 256-thread class of the paired rows is sized for (spatient_class_maxk(1) or up to 10)
  // ... the other patients (a single-tumour space of more than a tile; MMHN_SMALL=0 / the Jacobi solver: every patient)
  // take the staged kernels over these lists
  // [0]: patients that are their own problem (nothing of the joint path feeds them: a side stream), [1]: paired rows
  Staged stg[2];
  std::vector<int> paired;       // patients with a joint problem
  DevArr<int> d_paired;
  // ---- per-problem dispatch of the joint solves (round 5).  Every joint problem takes ONE of three routes:
  //   RT_W  window layout, a chain of patients per workgroup (wsolve.h) - when the batch has enough window-shaped problems
  //   RT_P  one workgroup per patient walking its tiles (k_psolve2)   - multi-tile problems, when there are enough of them
  //   RT_T  the tiles of all remaining problems in ONE cooperative launch (tsolve.h: k_csolve) - several workgroups per patient
  std::vector<uint8_t> route;
  std::vector<int> olist;        // the RT_P problems
  DevArr<int> d_olist;
  bool wpath = false;            // the batch has RT_W problems
  bool wdirect = false;          // the consumers read the window layout in place (no conversion to index order)
  std::vector<WDesc> wd;         // sorted by shape; the state vectors of consecutive entries lie back to back
  DevArr<WDesc> d_wd;
  int wnx = -1;                  // external bits of every window problem of the batch, -1: they differ
  bool wsplit = false;           // this batch runs the window route with two workgroups per patient (MMHN_WSPLIT)
  std::vector<WChain> wchains;   // runs of same-shape entries, one workgroup each (wsolve.h)
  DevArr<WChain> d_wchains;
  long long offT = 0;            // the RT_T problems' vectors start here (the tile solver skips their dead tiles: kept zero)
  CList clJ[2];                  // cooperative work lists of the RT_T tiles, forward / transposed
  std::vector<int2> lmapT;       // the RT_T tiles by level (MMHN_COOP=0: one launch per level)
  std::vector<int> lofT;
  DevArr<int2> d_lmapT;
  int maxkT = 0, maxkP = 0;      // largest RT_T / RT_P problem
  int max_dl = 0;                // largest (2^#P bits in a tile + 2^#M bits in a tile) over the RT_P problems
  std::vector<int> ptoff;        // live (seeded) tiles of the RT_P problems in index order (k_psolve2)
  std::vector<uint32_t> ptiles;
  DevArr<int> d_ptoff;
  DevArr<uint32_t> d_ptiles;
  double seeded_bytes_P = 0;     // bytes of the RT_P solutions (written once per solve)
  // class marginals of the problems in index order: work items of k_pclass (problem, class pass or eq block, range of its
  // outer loop) - a large problem is several workgroups
  std::vector<int4> pcl;
  DevArr<int4> d_pcl;
  DevArr<PatRec> d_pats;
  DevArr<Desc> d_dJ, d_dS;
  DevArr<int2> d_mapJ, d_mapS, d_lmapJ, d_lmapS;
};

// A seed = 0 tile of a joint space is dead when none of its states has PT == MT: with a right-hand
// side supported on e_0 / the seed = 1 half the solution is identically zero there, so the engine
// never launches it (the buffers are zeroed once; API calls with arbitrary vectors keep every tile).
static bool dead_tile(const Desc& d, uint32_t H) {
  if (d.mode != JOINT || d.seedbit < TB) return false;
  const uint32_t xhi = H << TB, hmask = ~((1u << TB) - 1u);
  if (xhi & (1u << d.seedbit)) return false;
  if (xhi & d.lone & hmask) return true;
  const uint32_t pp = d.pairP & hmask & ~(1u << 31);           // pairs whose P bit is a tile bit
  const uint32_t pm = (pp << 1);
  if (((xhi & pp) << 1) != (xhi & pm)) return true;
  // pair straddling the tile boundary (P bit TB-1, M bit TB): M set needs P set - possible inside the tile
  return false;
}

// sort a tile list by level (stable) and record the level offsets; prune drops dead tiles
static void build_levels(const std::vector<int2>& map, const std::vector<Desc>* descs, bool prune,
                         std::vector<int2>& lmap, std::vector<int>& lof) {
  int maxl = 0;
  std::vector<int2> keep;
  keep.reserve(map.size());
  for (const int2& m : map) {
    if (prune && descs && dead_tile((*descs)[m.x], (uint32_t)m.y)) continue;
    keep.push_back(m);
    maxl = std::max(maxl, popc((uint32_t)m.y));
  }
  lof.assign(maxl + 2, 0);
  for (const int2& m : keep) lof[popc((uint32_t)m.y) + 1]++;
  for (int l = 0; l <= maxl; ++l) lof[l + 1] += lof[l];
  lmap.resize(keep.size());
  std::vector<int> cur(lof.begin(), lof.end() - 1);
  for (const int2& m : keep) lmap[cur[popc((uint32_t)m.y)]++] = m;
}

static inline long long a_size(const Desc& d) {
  const int kP = popc(d.maskP), kM = popc(d.maskM), kE = popc(d.pairP);
  return ((long long)(kP + 1) << kP) + ((long long)(kM + 1) << kM) + ((long long)(kE + 2) << kE);
}

static void add_tiles(std::vector<int2>& map, int prob, int k) {
  const int tiles = k > TB ? 1 << (k - TB) : 1;
  for (int t = 0; t < tiles; ++t) map.push_back(make_int2(prob, t));
}

// the tiles a tile's step A reads (tsolve.h: tsolve_tile, same conditions): moves whose bits reach beyond the tile
static void tile_deps(const Desc& d, uint32_t H, bool tr, std::vector<uint32_t>& out) {
  out.clear();
  const int k = d.k, t = k < TB ? k : TB;
  const bool joint = d.mode == JOINT;
  for (int b = (t > 0 ? t - 1 : 0); b < k; ++b) {
    const bool is_pair = joint && ((d.pairP >> b) & 1u);
    for (int kind = 0; kind < 2; ++kind) {
      if (kind == 1 && !is_pair) continue;
      const uint32_t mv = kind == 0 ? (1u << b) : (3u << b);
      const uint32_t mh = mv >> t;
      if (mh == 0) continue;
      if (tr ? (H & mh) != 0 : (H & mh) != mh) continue;
      const uint32_t Hn = H ^ mh;
      if (std::find(out.begin(), out.end(), Hn) == out.end()) out.push_back(Hn);
    }
  }
}

// Work list of a cooperative solve over the tiles `map` (dead tiles of joint spaces dropped when `prune`): a topological
// order of the (transposed) system in which the deepest problems start first - key = levels a tile still has in front of
// it, counted from the END of its problem, so that every problem finishes in the last rounds (longest remaining path first).
static void build_clist(const std::vector<int2>& map, const std::vector<Desc>& descs, bool prune, bool tr, CList& cl) {
  cl.clear();
  struct Ent { int prob; uint32_t H; int key; int k; };
  std::vector<Ent> ents;
  ents.reserve(map.size());
  std::vector<int> maxlev(descs.size(), 0);
  for (const int2& m : map) {
    if (prune && dead_tile(descs[m.x], (uint32_t)m.y)) continue;
    maxlev[m.x] = std::max(maxlev[m.x], popc((uint32_t)m.y));
    ents.push_back(Ent{m.x, (uint32_t)m.y, 0, descs[m.x].k});
  }
  if (ents.empty()) return;
  for (Ent& e : ents) {
    const int lev = popc(e.H);
    e.key = tr ? -lev : lev - maxlev[e.prob];
    cl.maxk = std::max(cl.maxk, e.k);
    cl.nlevels = std::max(cl.nlevels, maxlev[e.prob] + 1);
  }
  std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.key != b.key ? a.key < b.key : a.k > b.k; });
  // position of every live tile in the list
  std::vector<long long> toff(descs.size() + 1, 0);
  std::vector<char> used(descs.size(), 0);
  for (const Ent& e : ents) used[e.prob] = 1;
  for (size_t p = 0; p < descs.size(); ++p) toff[p + 1] = toff[p] + (used[p] ? (descs[p].k > TB ? 1ll << (descs[p].k - TB) : 1) : 0);
  std::vector<int> pos((size_t)toff.back(), -1);
  for (size_t i = 0; i < ents.size(); ++i) pos[(size_t)(toff[ents[i].prob] + ents[i].H)] = (int)i;
  cl.items.resize(ents.size());
  std::vector<uint32_t> nb;
  for (size_t i = 0; i < ents.size(); ++i) {
    const Ent& e = ents[i];
    tile_deps(descs[e.prob], e.H, tr, nb);
    CItem it{e.prob, e.H, (int)cl.deps.size(), 0};
    for (uint32_t Hn : nb) {
      const int j = pos[(size_t)(toff[e.prob] + Hn)];
      if (j < 0) continue;                                   // a dead tile: zeros, written by nobody
      if (j >= (int)i) throw Fail{"cooperative solve: the work list is not a topological order"};
      cl.deps.push_back(j);
      ++it.ndep;
    }
    if (it.ndep > 63) throw Fail{"cooperative solve: more than 63 dependencies of one tile"};
    cl.items[i] = it;
  }
  if (cl.deps.empty()) cl.deps.push_back(0);
  cl.d_items.alloc(cl.items.size());
  cl.d_deps.alloc(cl.deps.size());
  HIPCHECK(hipMemcpy(cl.d_items.p, cl.items.data(), cl.items.size() * sizeof(CItem), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(cl.d_deps.p, cl.deps.data(), cl.deps.size() * sizeof(int), hipMemcpyHostToDevice));
}

struct EngineBase {
  virtual ~EngineBase() = default;
  int dtype = 0;
  int device = 0;
  hipStream_t stream = nullptr;
};

// Every ABI entry runs with the engine's GPU current and puts the caller's device back on exit, so engines on
// different GPUs can live in one process (and a handle may be used from a thread whose current device differs).
struct DevGuard {
  int prev = -1;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) HIPCHECK(hipSetDevice(dev));
    else prev = -1;
  }
  ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DevGuard(const DevGuard&) = delete;
  DevGuard& operator=(const DevGuard&) = delete;
};

// RCCL entry points, resolved on first use (a single-GPU process never loads the library).  "librccl.so.1" is the
// soname both ROCm and the PyTorch wheel ship: inside a torch.distributed process this is the copy already loaded.
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl& rccl() {
  static Rccl r;
  if (r.lib) return r;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* nm : names) if ((lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!lib) throw Fail{std::string("cannot load RCCL: ") + dlerror()};
  auto sym = [&](const char* nm) {
    void* f = dlsym(lib, nm);
    if (!f) throw Fail{std::string("RCCL symbol missing: ") + nm};
    return f;
  };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
  r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  r.lib = lib;
  return r;
}
#define RCCLCHECK(expr)                                                                          \
  do {                                                                                           \
    ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess) throw Fail{std::string(#expr) + ": " + rccl().GetErrorString(r_)};    \
  } while (0)

// plain stream for mmhn_bench_stream: the denominator the HBM-bound kernels are compared with.  Four 16-byte
// accesses per lane in flight per trip, one contiguous 4 KiB run per wave and trip.
__global__ __launch_bounds__(256) void k_stream(double2* __restrict__ a, const double2* __restrict__ b,
                                                const double2* __restrict__ c, size_t n16, int kind) {
  constexpr int U = 4;
  const size_t lane = threadIdx.x & 63, wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t base = wave * (64 * U); base < n16; base += nwave * (64 * U)) {
    double2 u[U], v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const size_t i = base + q * 64 + lane;
      if (i < n16) { u[q] = b[i]; if (kind == 1) v[q] = c[i]; }
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const size_t i = base + q * 64 + lane;
      if (i < n16) a[i] = kind == 0 ? u[q] : make_double2(u[q].x + 3.0 * v[q].x, u[q].y + 3.0 * v[q].y);
    }
  }
}

// sums[2][stride] (EM, NM rows of k_reduce_parts) -> the buffer layout of mmhn_cohort_sums (include/metmhn_amd.h)
__global__ void k_pack_sums(const double* __restrict__ sums, int N, double n_em, double n_pat, double* __restrict__ o) {
  const int st = 1 + N * N + 2 * N, NN = N * N;
  const double* em = sums;
  const double* nm = sums + st;
  const int total = 4 + 2 * NN + 3 * N;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    double v;
    if (e == 0) v = em[0];
    else if (e == 1) v = nm[0];
    else if (e == 2) v = n_em;
    else if (e == 3) v = n_pat;
    else {
      int q = e - 4;
      if (q < NN) v = em[1 + q];
      else if ((q -= NN) < NN) v = nm[1 + q];
      else if ((q -= NN) < N) v = em[1 + NN + q];
      else if ((q -= N) < N) v = nm[1 + NN + q];
      else v = em[1 + NN + N + (q - N)];
    }
    o[e] = v;
  }
}

// sums[2][stride] -> the pre-combined buffer of mmhn_cohort_wsums: [w s_EM + s_NM, w G_EM + G_NM, w p_EM + p_NM, w m_EM]
// (regularized_optimization.py:256-266 without the division by n_full): 1 + N^2 + 2 N doubles, the all-reduce
// payload of SURVEY 8e - the weight w only needs the GLOBAL counts, which every rank knows when the cohort is set
__global__ void k_pack_wsums(const double* __restrict__ sums, int N, double w, double* __restrict__ o, double flag) {
  const int st = 1 + N * N + 2 * N, NN = N * N;
  if (blockIdx.x == 0 && threadIdx.x == 0) o[st] = flag;              // (mmhn_set_reduce_flag: rides in the same all-reduce)
  const double* em = sums;
  const double* nm = sums + st;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < st; e += gridDim.x * blockDim.x)
    o[e] = e < 1 + NN + N ? w * em[e] + nm[e] : w * em[e];          // (d_d_m has no NM part, :266)
}

// host <-> device traffic of an evaluation without the copy engine: the parameters are read from, and the result is
// written to, pinned host memory by kernels of the evaluation's own queue.  (A hipMemcpyAsync in front of the first
// launch costs the hand-over between the copy engine and the compute queue - on a 0.4 ms evaluation of a small cohort
// ~60 us passed between the 27 KB upload and the first kernel - and the download the same at the other end.)
__global__ void k_copy_words(const uint4* __restrict__ src, uint4* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
// ... and the head of an evaluation in ONE launch: the parameter upload, the cleared cohort sums and the cleared gradient
// work arrays of the first batch (three launches otherwise, each a few us of queue latency on a short evaluation)
__global__ void k_begin_eval(const uint4* __restrict__ src, uint4* __restrict__ dst, int n, double* __restrict__ sums, int nsums,
                             uint4* __restrict__ z, long long nz) {
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
  if (i0 < n) dst[i0] = src[i0];
  if (i0 < nsums) sums[i0] = 0.0;
  for (long long i = i0; i < nz; i += step) z[i] = uint4{0u, 0u, 0u, 0u};
}

// k_reduce_parts of the LAST batch and the packing in one launch: thread e adds the chunk sums of both classes in chunk
// order (as k_reduce_parts does), keeps sums[] current and writes its entries of the packed buffer
// (mode 1: k_pack_sums layout, a = n_em, b = n_pat; mode 2: k_pack_wsums, a = w)
__global__ __launch_bounds__(BLOCK) void k_reduce_parts_pack(const double* __restrict__ part, int stride, int nelem, int nchunk,
                                                             double* sums, int mode, int N, double a, double b,
                                                             double* __restrict__ o) {
  const int e = blockIdx.x * BLOCK + threadIdx.x;
  if (e == 0 && mode == 2) o[stride] = b;                             // (mode 2: b = the reduce flag, behind the buffer)
  if (e >= stride) return;
  double em = sums[e], nm = sums[stride + e];
  if (e < nelem) {
    double acc0 = 0, acc1 = 0;
#pragma unroll 8
    for (int c = 0; c < nchunk; ++c) {
      acc0 += part[((long long)c * 2 + 0) * stride + e];
      acc1 += part[((long long)c * 2 + 1) * stride + e];
    }
    em += acc0; nm += acc1;
    sums[e] = em; sums[stride + e] = nm;
  }
  const int NN = N * N;
  if (mode == 2) { o[e] = e < 1 + NN + N ? a * em + nm : a * em; return; }
  if (e == 0) { o[0] = em; o[1] = nm; o[2] = a; o[3] = b; }
  else if (e < 1 + NN) { o[4 + (e - 1)] = em; o[4 + NN + (e - 1)] = nm; }
  else if (e < 1 + NN + N) { o[4 + 2 * NN + (e - 1 - NN)] = em; o[4 + 2 * NN + N + (e - 1 - NN)] = nm; }
  else o[4 + 2 * NN + 2 * N + (e - 1 - NN - N)] = em;
}

template <typename T>
struct Engine : EngineBase {
  int n = 0, N = 0;
  size_t ws_limit = 0;
  DevArr<Params<T>> d_par;
  Params<T>* h_par = nullptr;         // pinned: the per-evaluation upload is a true async copy
  double* h_abi = nullptr;            // pinned landing buffer of the result download
  void* h_par_dev = nullptr;          // device views of the two pinned buffers
  double* h_abi_dev = nullptr;
  bool zero_copy = true;              // MMHN_ZEROCOPY=0: hipMemcpyAsync up and down instead
  // HIP events around the dominant kernels (mmhn_get_counters): an event record costs ~5 us of host time, which a short
  // evaluation cannot afford (it is bound by the host's issue rate) - only batches of at least 2^24 states are timed
  bool time_kernels = true;
  // cohort
  std::vector<int8_t> dat;
  long long n_pat = 0;
  int n_cols = 0;
  std::vector<Batch> batches;
  double n_em = 0;
  // workspace (sized for the largest batch)
  DevArr<T> pi, lidgJ, qJ, rhsJ, rhsS, pS, lidgS, qS, seedS, GS, dots, bmJ, bmS, tabJ, tabS;
  // accumulators of the joint gradient that must be zero on entry - rows of the three G matrices, observation-rate rows,
  // class marginals - share one allocation, laid out per batch and cleared by ONE memset at the start of the batch
  DevArr<T> zarena;
  struct View { T* p = nullptr; } GJ, DJ, Abuf;
  static long long zarena_elems(long long nJ, long long asize, int N) { return up4(3 * nJ * N * N) + up4(3 * nJ * N) + up4(asize); }
  static long long up4(long long v) { return (v + 3) / 4 * 4; }
  int pi_owner = -1, qJ_owner = -1;   // batch whose (pruned) layout the zero-initialised buffers hold
  DevArr<double> lp, out, sums, abi_sums, redbuf;
  DevArr<JLink<T>> links;
  // patient shards on several GPUs: one communicator per engine, the all-reduce runs on the engine's stream
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  // popcount-ordered state permutations of every tile size (k_tsolve step B)
  DevArr<uint16_t> d_perm;
  DevArr<int> d_lvl;
  bool use_jacobi = false;      // MMHN_SOLVER=jacobi: the reference's k+1 sweeps instead of substitution
  bool poison = false;          // MMHN_POISON=1: NaN-fill the solution buffers of per-patient batches before each evaluation
  hipStream_t side[3] = {};               // side streams: [0], [1] of the small-space path, [2] of the staged own-problem patients
  hipEvent_t ev_fork[3] = {}, ev_join[3] = {};
  bool small_path = true;       // MMHN_SMALL=0: keep the staged kernels for single-tumour spaces that fit one tile
  int prep_split_max = 2048;    // MMHN_PREP_SPLIT: problems up to which k_prep / k_pclass run a workgroup per table / class pass
  bool pair_small = true;       // MMHN_PAIR_SMALL=0: the two marginal problems of a paired row one after the other
  int psolve_min = 384;         // multi-tile joint problems (outside the window route) in a batch from which they take one workgroup
                                // per patient (k_psolve2) instead of the cooperative tile launch (MMHN_PSOLVE_MIN)
  int wsolve_min = 128;         // window-shaped joint problems in a batch from which they take the window route (MMHN_WSOLVE_MIN;
                                // follows MMHN_PSOLVE_MIN when only that one is set)
  bool force_timing = false;    // MMHN_TIME_KERNELS=1: HIP events around the solve / class-marginal launches of every batch (bench
                                // breakdowns of small cohorts; an event pair costs the host ~10 us)
  bool coop_fault = false;      // MMHN_COOP_FAULT=1 (tests): the first tile of every cooperative launch never raises its flag
  int pcl_per = 16;             // MMHN_PCL_PER: tiles of a class pass per work item of k_pclass
  int coop_wgs = 0;             // MMHN_COOP_WGS: workgroups of a cooperative launch (default: one per CU - with two the launch holds every
                                // wave slot of the chip and the side streams' kernels wait for its end: 1.65 against 1.47 ms on the 28-event LUAD cohort)
  bool coop = true;             // MMHN_COOP=0: tile solves as one launch per level (k_tsolve) instead of one cooperative launch
  // cooperative launches (tsolve.h): queue heads + abort word, the flags of the tiles (value = epoch of the launch that
  // finished the tile), the pinned host copy of the abort word
  DevArr<CoopCtl> coop_ctl;
  DevArr<unsigned> coop_flags[2];   // one set per lane: launches of two streams may be in flight together
  unsigned coop_epoch = 0;
  int coop_slot = 0;
  int cur_lane = 0;                 // lane of the launches being issued (1: a side stream)
  unsigned* h_abort = nullptr;
  unsigned* h_abort_dev = nullptr;
  int wsolve_chain = 1;         // MMHN_WSOLVE_CHAIN=0: every window problem its own chain (the pipeline drains between patients)
  int wsolve_wgs = 0;           // MMHN_WSOLVE_WGS: workgroups of the window solve (default: one per CU)
  bool wsplit = false;          // MMHN_WSPLIT=1: two workgroups per patient on the window route (wsolve.h: SPLIT; measured, off)
  DevArr<unsigned> wprog;       // ... the progress words of its pairs
  unsigned wsplit_epoch = 0;
  int wsolve_mode = 1;          // joint solves of per-patient batches in the window layout (wsolve.h); MMHN_WSOLVE=0: the tile
                                // kernels (k_psolve2) for every problem, 2: window solves converted back to index order
  DevArr<T> piM, qM;            // matrix / window path: solutions in their own layout
  int n_cu = 256;
  // counters
  mmhn_counters cnt{};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  std::vector<double> ev_bytes;
  std::vector<int> ev_slot;           // MMHN_K_* class of the timed launch

  Engine(int dev, int n_mut) : n(n_mut), N(n_mut + 1) {
    device = dev;
    cnt.comm_rank = -1;
    REQUIRE(n_mut >= 1 && n_mut < MAXN, "n_mut must be in [1, 31]");
    DevGuard guard(device);
    HIPCHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    d_par.alloc(NPSET);
    HIPCHECK(hipHostMalloc(reinterpret_cast<void**>(&h_par), NPSET * sizeof(Params<T>), hipHostMallocDefault));
    HIPCHECK(hipHostGetDevicePointer(&h_par_dev, h_par, 0));
    static_assert(sizeof(Params<T>) % sizeof(uint4) == 0, "parameter block: whole 16-byte words");
    sums.alloc(2 * stride());
    size_t free_b = 0, total_b = 0;
    HIPCHECK(hipMemGetInfo(&free_b, &total_b));
    ws_limit = (size_t)(0.7 * (double)free_b);
    {
      std::vector<uint16_t> perm((size_t)(TB + 1) << TB, 0);
      std::vector<int> lvl((size_t)(TB + 1) * (TB + 2), 0);
      for (int t = 0; t <= TB; ++t) {
        int pos = 0;
        for (int l = 0; l <= t; ++l) {
          lvl[(size_t)t * (TB + 2) + l] = pos;
          for (uint32_t x = 0; x < (1u << t); ++x)
            if (popc(x) == l) perm[((size_t)t << TB) + pos++] = (uint16_t)x;
        }
        lvl[(size_t)t * (TB + 2) + t + 1] = pos;
      }
      d_perm.alloc(perm.size());
      d_lvl.alloc(lvl.size());
      HIPCHECK(hipMemcpy(d_perm.p, perm.data(), perm.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      HIPCHECK(hipMemcpy(d_lvl.p, lvl.data(), lvl.size() * sizeof(int), hipMemcpyHostToDevice));
      const char* sv = std::getenv("MMHN_SOLVER");
      use_jacobi = sv && std::string(sv) == "jacobi";
      if (const char* pm = std::getenv("MMHN_PSOLVE_MIN")) { psolve_min = std::atoi(pm); wsolve_min = psolve_min; }
      if (const char* pm = std::getenv("MMHN_WSOLVE_MIN")) wsolve_min = std::atoi(pm);
      if (const char* pm = std::getenv("MMHN_COOP")) coop = std::atoi(pm) != 0;
      if (const char* pm = std::getenv("MMHN_TIME_KERNELS")) force_timing = std::atoi(pm) != 0;
      if (const char* pm = std::getenv("MMHN_COOP_WGS")) coop_wgs = std::atoi(pm);
      if (const char* pm = std::getenv("MMHN_PCL_PER")) pcl_per = std::max(1, std::atoi(pm));
      if (const char* pm = std::getenv("MMHN_COOP_FAULT")) coop_fault = std::atoi(pm) != 0;
      if (const char* po = std::getenv("MMHN_POISON")) poison = std::atoi(po) != 0;
      if (const char* sp = std::getenv("MMHN_SMALL")) small_path = std::atoi(sp) != 0;
      if (const char* sp = std::getenv("MMHN_PAIR_SMALL")) pair_small = std::atoi(sp) != 0;
      if (const char* sp = std::getenv("MMHN_ZEROCOPY")) zero_copy = std::atoi(sp) != 0;
      if (const char* sp = std::getenv("MMHN_PREP_SPLIT")) prep_split_max = std::atoi(sp);
      if (const char* ms = std::getenv("MMHN_WSOLVE")) wsolve_mode = std::atoi(ms);
      if (const char* ms = std::getenv("MMHN_WSOLVE_WGS")) wsolve_wgs = std::atoi(ms);
      if (const char* ms = std::getenv("MMHN_WSPLIT")) wsplit = std::atoi(ms) != 0;
      if (const char* ms = std::getenv("MMHN_WSOLVE_CHAIN")) wsolve_chain = std::atoi(ms);
      hipDeviceProp_t prop;
      HIPCHECK(hipGetDeviceProperties(&prop, device));
      n_cu = std::max(1, prop.multiProcessorCount);
    }
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsolve_lds<T>()));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsolve_lds<T>()));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve<T, false, WCfg<T>::NXT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsolve_lds<T>()));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve<T, true, WCfg<T>::NXT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsolve_lds<T>()));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve<T, false, WCfg<T>::NXT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsolve_lds<T>()));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve<T, true, WCfg<T>::NXT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsolve_lds<T>()));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wclass<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wclass_lds<T>()));
    // kernels may need more than the default dynamic LDS window
    const int lds = 150 * 1024;
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csolve<T, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csolve<T, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csolve<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csolve<T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csolve<T, false, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csolve<T, true, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    coop_ctl.alloc(1);
    HIPCHECK(hipMemset(coop_ctl.p, 0, sizeof(CoopCtl)));
    HIPCHECK(hipHostMalloc(reinterpret_cast<void**>(&h_abort), 64, hipHostMallocDefault));
    HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h_abort_dev), h_abort, 0));
    *h_abort = 0u;
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spatient2<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spatient<T, 1024, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int i = 0; i < 3; ++i) {
      HIPCHECK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
      HIPCHECK(hipEventCreateWithFlags(&ev_fork[i], hipEventDisableTiming));
      HIPCHECK(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
    }
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_psolve2<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_psolve2<T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_psolve2<T, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_psolve2<T, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tsolve<T, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tsolve<T, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tsolve<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tsolve<T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kv<T, false, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kv<T, true, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kv<T, false, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kv<T, true, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep<T, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep<T, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_diag<T>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_diag<T, 1024>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prep<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prep<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prep<T, false, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grad_rows<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grad_rows<T>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pclass<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pclass<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_class_marg<T>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  ~Engine() override {                       // runs under the DevGuard of mmhn_destroy
    comm_destroy();
    for (auto& e : ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (int i = 0; i < 3; ++i) {
      if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
      if (ev_fork[i]) (void)hipEventDestroy(ev_fork[i]);
      if (side[i]) (void)hipStreamDestroy(side[i]);
    }
    if (h_par) (void)hipHostFree(h_par);
    if (h_abi) (void)hipHostFree(h_abi);
    if (h_abort) (void)hipHostFree(h_abort);
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
  }
  int stride() const { return 1 + N * N + 2 * N; }

  // ---------------------------------------------------------------- parameters
  void build_params(const double* lt, const double* ldp, const double* ldm, bool upload = true) {
    REQUIRE(!sums_pending, "an evaluation begun with mmhn_cohort_sums_begin has not been collected (its parameter upload may still be in flight)");
    // exp(theta_ij - d_j) = exp(theta_ij) * exp(-d_j): N^2 + 2N exponentials per evaluation instead of 3 N^2 (this runs
    // on the host before the first launch of every evaluation - 26 us of a 400 us LUAD evaluation as three full passes)
    std::vector<double> eth((size_t)N * N), enp(N, 1.0), enm(N, 1.0);
    for (int e = 0; e < N * N; ++e) eth[e] = std::exp(lt[e]);
    if (ldp) for (int j = 0; j < N; ++j) enp[j] = std::exp(-ldp[j]);
    if (ldm) for (int j = 0; j < N; ++j) enm[j] = std::exp(-ldm[j]);
    for (int s = 0; s < NPSET; ++s) {
      Params<T>& P = h_par[s];
      std::memset(&P, 0, sizeof(P));
      for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) {
          double v = eth[i * N + j];
          if (s == PS_PRIM && j == n && i < n) v = 1.0;             // likelihood.py:313
          if (s == PS_MET && i != j) v *= enm[j];                   // kronvec.py:18
          if (s == PS_PRIM && i != j) v *= enp[j];
          P.th[i][j] = (T)v;
        }
        P.baseP[i] = (T)eth[i * N + i];
        P.baseM[i] = i < n ? (T)(eth[i * N + i] * eth[i * N + n]) : (T)0;
        P.dp[i] = (T)(ldp ? std::exp(ldp[i]) : 1.0);
        P.dm[i] = (T)(ldm ? std::exp(ldm[i]) : 1.0);
      }
    }
    if (!upload) return;                                   // (the caller's first launch carries it: k_begin_eval)
    if (zero_copy) {
      const int nw = (int)(NPSET * sizeof(Params<T>) / sizeof(uint4));
      hipLaunchKernelGGL(k_copy_words, dim3((nw + 255) / 256), dim3(256), 0, stream, static_cast<const uint4*>(h_par_dev),
                         reinterpret_cast<uint4*>(d_par.p), nw);
      HIPCHECK(hipGetLastError());
    } else {
      HIPCHECK(hipMemcpyAsync(d_par.p, h_par, NPSET * sizeof(Params<T>), hipMemcpyHostToDevice, stream));
    }
  }

  // ---------------------------------------------------------------- launches
  size_t sweep_lds(int maxk) const { return DESC_PAD + ((size_t)(1 << TB) + 2 * (size_t)std::max(maxk, 1) * 64) * sizeof(T); }
  // joint: a workgroup per table of a problem (three class tables + the rate tables, k_prep) when the launch is short
  // enough for its length to be one workgroup's chain; large cohorts keep one workgroup per problem
  // (tables of more than 2^13 entries - k = 25 - are also cut into parts of 2^13: grid.y = 4 * parts)
  void prep(const Desc* descs, int nprob, T* tab, bool joint = false, int maxkc = 0) {
    if (nprob == 0) return;
    // SPLIT: a table of more than 2^9 entries is dealt over several workgroups (the launch is the head of every evaluation
    // of a short cohort; at most 32 parts: the workgroups of the shorter tables of the launch exit at once, but they are launched)
    const int parts = maxkc > 9 ? 1 << std::min(maxkc - 9, 5) : 1;
    if (joint && nprob <= prep_split_max) hipLaunchKernelGGL((k_prep<T, true>), dim3(nprob, 4 * parts), dim3(BLOCK), prep_lds<T>(N), stream, descs, d_par.p, tab);
    else if (nprob >= 256) hipLaunchKernelGGL((k_prep<T, false, 1024>), dim3(nprob), dim3(1024), prep_lds<T>(N), stream, descs, d_par.p, tab);
    else hipLaunchKernelGGL((k_prep<T, false>), dim3(nprob), dim3(BLOCK), prep_lds<T>(N), stream, descs, d_par.p, tab);
    HIPCHECK(hipGetLastError());
  }

  void launch_sweep(bool tr, const Desc* descs, const int2* map, int ntiles, int maxk, const T* p, T* y,
                    const T* lidg, const T* rhs, int rhs_mode, const T* scal, double alg_bytes, const T* tab) {
    if (ntiles == 0) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = alg_bytes > 0 && time_kernels;
    if (timed) {
      if (ev_used == ev_pool.size()) {
        hipEvent_t a, b;
        HIPCHECK(hipEventCreate(&a));
        HIPCHECK(hipEventCreate(&b));
        ev_pool.push_back({a, b});
      }
      e0 = ev_pool[ev_used].first;
      e1 = ev_pool[ev_used].second;
      ++ev_used;
      ev_bytes.push_back(alg_bytes);
      ev_slot.push_back(MMHN_K_OTHER_SOLVE);
      HIPCHECK(hipEventRecord(e0, stream));
    }
    const size_t lds = sweep_lds(maxk);
    if (tr)
      hipLaunchKernelGGL((k_sweep<T, true>), dim3(ntiles), dim3(KSB), lds, stream, descs, map, d_par.p, p, y,
                         lidg, rhs, rhs_mode, scal, std::max(maxk, 1), tab);
    else
      hipLaunchKernelGGL((k_sweep<T, false>), dim3(ntiles), dim3(KSB), lds, stream, descs, map, d_par.p, p, y,
                         lidg, rhs, rhs_mode, scal, std::max(maxk, 1), tab);
    HIPCHECK(hipGetLastError());
    if (timed) HIPCHECK(hipEventRecord(e1, stream));
  }
  // plain product on full tiles of multi-tile spaces (k_kv); hxt from k_hx for the same map
  // zmap: see k_kv; lidg / rhs: fused Jacobi step y = lidg * (Q_off p + rhs) over the same tile list
  void launch_kv(bool tr, const Desc* descs, const int2* map, int ntiles, int maxk, const T* p, T* y, const T* tab, const T* hxt,
                 const int* zmap = nullptr, const T* lidg = nullptr, const T* rhs = nullptr) {
    if (ntiles == 0) return;
    const size_t lds = DESC_PAD + ((size_t)(1 << TB) + 2 * (size_t)maxk * 64 + 32) * sizeof(T);
#define KV_ARGS dim3(ntiles), dim3(KSB), lds, stream, descs, map, ntiles, p, y, tab, hxt, maxk, zmap, lidg, rhs
    if (lidg) {
      if (tr) hipLaunchKernelGGL((k_kv<T, true, 1, true>), KV_ARGS);
      else hipLaunchKernelGGL((k_kv<T, false, 1, true>), KV_ARGS);
    } else {
      if (tr) hipLaunchKernelGGL((k_kv<T, true, 1, false>), KV_ARGS);
      else hipLaunchKernelGGL((k_kv<T, false, 1, false>), KV_ARGS);
    }
#undef KV_ARGS
    HIPCHECK(hipGetLastError());
  }
  void collect_events() {
    for (size_t i = 0; i < ev_used; ++i) {
      float ms = 0;
      HIPCHECK(hipEventElapsedTime(&ms, ev_pool[i].first, ev_pool[i].second));
      mmhn_kernel_counter& c = cnt.kernel[ev_slot[i]];
      c.ms += ms;
      c.launches += 1;
      c.alg_bytes += ev_bytes[i];
    }
    ev_used = 0;
    ev_bytes.clear();
    ev_slot.clear();
  }

  void launch_diag(const Desc* descs, const int2* map, int ntiles, const T* p, T* outp, const T* dvec, int what,
                   int pbit = -1) {
    if (ntiles == 0) return;
    const size_t lds = DESC_PAD + ((size_t)4 * N * 64 + 256) * sizeof(T);
    if (ntiles >= 256)
      hipLaunchKernelGGL((k_diag<T, 1024>), dim3(ntiles), dim3(1024), lds, stream, descs, map, d_par.p, p, outp, dvec, what, N, pbit);
    else
      hipLaunchKernelGGL((k_diag<T>), dim3(ntiles), dim3(BLOCK), lds, stream, descs, map, d_par.p, p, outp, dvec,
                         what, N, pbit);
    HIPCHECK(hipGetLastError());
  }
  // dj != nullptr (joint kinds): one extra row per problem with the observation-rate gradient
  // work list of k_grad_rows: one entry per (problem, chunk of 2^GR_CHUNK subsets of the kind's lattice)
  static std::vector<int2> grad_chunks(const std::vector<Desc>& ds, int kind) {
    std::vector<int2> out;
    for (size_t i = 0; i < ds.size(); ++i) {
      const Desc& d = ds[i];
      const int kc = kind == GK_P ? popc(d.maskP) : kind == GK_M ? popc(d.maskM) : kind == GK_E ? popc(d.pairP) : d.k;
      const int nch = 1 << std::max(0, kc - GR_CHUNK);
      for (int c = 0; c < nch; ++c) out.push_back(int2{(int)i, c});
    }
    return out;
  }
  // partial rows of the subset chunks are added up: G (and dj) must be zero on entry
  // kind < 0: the work list holds the three joint kinds (kind in bits 24+ of the chunk field), G matrices gstride apart
  void launch_grad_rows(const Desc* descs, int nprob, int maxk, const T* A, const T* p, const T* q, T* G, int kind,
                        const DevArr<int2>& chunks, T* dj = nullptr, long long gstride = 0) {
    if (nprob == 0 || chunks.n == 0) return;
    (void)maxk;
    const int rows = N + (dj ? 1 : 0);
    if (chunks.n >= 1024) {
      // long launches: one row (wave) per workgroup - no barrier, nothing shared but the class-bit list (measured on the bench cohort,
      // rows per workgroup 1 / 2 / 4 / 8 / 12: the step minus its big kernels 4.08 / 4.12 / 4.16 / 4.45 / 4.62 ms)
      const size_t lds1 = ((size_t)192 + 32) * sizeof(T);
      hipLaunchKernelGGL((k_grad_rows<T, 1>), dim3((unsigned)chunks.n, rows), dim3(64), lds1, stream,
                         descs, d_par.p, A, p, q, G, kind, dj, chunks.p, nprob, gstride);
      HIPCHECK(hipGetLastError());
      return;
    }
    const size_t lds = ((size_t)WAVES * 192 + WAVES * 32) * sizeof(T);
    hipLaunchKernelGGL((k_grad_rows<T>), dim3((unsigned)chunks.n, (rows + WAVES - 1) / WAVES), dim3(BLOCK), lds, stream,
                       descs, d_par.p, A, p, q, G, kind, dj, chunks.p, nprob, gstride);
    HIPCHECK(hipGetLastError());
  }
  void zero(T* p, long long count) {
    if (count > 0) HIPCHECK(hipMemsetAsync(p, 0, (size_t)count * sizeof(T), stream));
  }
  // timed launch helper shared by the two solvers
  template <typename F>
  void timed(int slot, double alg_bytes, F&& launch) {
    if (!time_kernels) {
      launch();
      HIPCHECK(hipGetLastError());
      return;
    }
    if (ev_used == ev_pool.size()) {
      hipEvent_t a, b;
      HIPCHECK(hipEventCreate(&a));
      HIPCHECK(hipEventCreate(&b));
      ev_pool.push_back({a, b});
    }
    hipEvent_t e0 = ev_pool[ev_used].first, e1 = ev_pool[ev_used].second;
    ++ev_used;
    ev_bytes.push_back(alg_bytes);
    ev_slot.push_back(slot);
    HIPCHECK(hipEventRecord(e0, stream));
    launch();
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventRecord(e1, stream));
  }

  // one list of problems with its tile maps (cl: its cooperative work lists, forward / transposed, or nullptr)
  struct PList {
    const Desc* d; const int2* map; int ntiles; int maxk; long long vec;
    const int2* lmap; const std::vector<int>* lof; const T* tab;
    const CList* cl = nullptr;
    int kslot = MMHN_K_OTHER_SOLVE;    // counter class of its launches
  };

  // k_psolve2 (one workgroup per patient, all-seeded-tile launches)
  size_t psolve2_lds(int maxk) const {
    return DESC_PAD + ((size_t)(1 << TB) + (size_t)((1 << TB) / TSB) + 3 * (size_t)maxk * 64 + (size_t)maxk * maxk + maxk) * sizeof(T) + 400 * sizeof(uint32_t) + (size_t)TSB * sizeof(uint16_t);
  }
  // groups of chains under MMHN_WSPLIT (a multiple of eight: the pair of a block is the block eight further on)
  int wsplit_groups() const { return std::max(8, (wsolve_wgs > 0 ? wsolve_wgs : n_cu) / 16 * 8); }
  // the joint solves of a batch, every problem on its route (Batch::route)
  void psolve(bool tr, const Batch& b, T* y, int rhs_mode) {
    const int nJ = (int)b.dJ.size();
    if (nJ == 0) return;
    if (b.wpath) {
      // window route: the solution is written once (seeded half)
      const int nW = (int)b.wd.size();
      double bytes = 0;
      for (const WDesc& w : b.wd) bytes += 0.5 * (double)(1ll << b.dJ[w.prob].k) * sizeof(T);
      T* yw = b.wdirect ? y : (tr ? qM.p : piM.p);
      timed(tr ? MMHN_K_PSOLVE_ADJ : MMHN_K_PSOLVE_FWD, bytes, [&]() {
        const int nch = (int)b.wchains.size();
        const dim3 g((unsigned)std::min(nch, wsolve_wgs > 0 ? wsolve_wgs : n_cu)), bk(WROWS);
        const size_t lds = wsolve_lds<T>();
#define WS_ARGS g, bk, lds, stream, b.d_dJ.p, b.d_wd.p, b.d_wchains.p, nch, yw, tabJ.p, links.p, qS.p, (unsigned*)nullptr, 0u, (unsigned*)nullptr, (unsigned*)nullptr
        // (a batch whose chains all have the same number of external bits - every k = 20 cohort: 5, k = 25: 9 - runs the instantiation
        // that knows it at compile time)
        if (b.wsplit) {
          // two workgroups per patient: pairs (b, b + 8) walk the chains dealt to wsplit_groups() groups
          const int npairs = std::min((nch + 7) / 8 * 8, wsplit_groups());
          if (wprog.n < (size_t)npairs) { wprog.alloc((size_t)wsplit_groups() + 64); wsplit_epoch = 0; }
          if (wsplit_epoch == 0 || wsplit_epoch >= 2047u) {
            HIPCHECK(hipMemsetAsync(wprog.p, 0, wprog.n * sizeof(unsigned), stream));
            wsplit_epoch = 0;
          }
          ++wsplit_epoch;
          coop_used = true;
          const dim3 g2((unsigned)(2 * npairs));
#define WS2_ARGS g2, bk, lds, stream, b.d_dJ.p, b.d_wd.p, b.d_wchains.p, nch, yw, tabJ.p, links.p, qS.p, wprog.p, wsplit_epoch << 20, &coop_ctl.p->abort, h_abort_dev
          if (tr) hipLaunchKernelGGL((k_wsolve<T, true, WCfg<T>::NXT, true>), WS2_ARGS);
          else hipLaunchKernelGGL((k_wsolve<T, false, WCfg<T>::NXT, true>), WS2_ARGS);
#undef WS2_ARGS
        } else if (b.wnx == WCfg<T>::NXT) { if (tr) hipLaunchKernelGGL((k_wsolve<T, true, WCfg<T>::NXT>), WS_ARGS); else hipLaunchKernelGGL((k_wsolve<T, false, WCfg<T>::NXT>), WS_ARGS); }
        else if (tr) hipLaunchKernelGGL((k_wsolve<T, true>), WS_ARGS);
        else hipLaunchKernelGGL((k_wsolve<T, false>), WS_ARGS);
#undef WS_ARGS
      });
      if (!b.wdirect) {
        hipLaunchKernelGGL((k_wconvert<T>), dim3(nW, 32), dim3(WROWS), 0, stream, b.d_dJ.p, b.d_wd.p, yw, y);
        HIPCHECK(hipGetLastError());
      }
    }
    if (!b.olist.empty()) {
      // one workgroup per patient (multi-tile problems, many of them)
      const int mk = std::max(b.maxkP, 1);
      const long long spare = (80 * 1024 - 64) - (long long)psolve2_lds(mk);
      const int dl_cap = (int)std::max<long long>(0, std::min<long long>(PS_DL2, spare / (long long)sizeof(T)));
#ifdef MMHN_ABL_PACK
      const size_t lds = std::max<size_t>(psolve2_lds(mk) + (size_t)dl_cap * sizeof(T), 100 * 1024);   // one workgroup per CU
#else
      const size_t lds = psolve2_lds(mk) + (size_t)dl_cap * sizeof(T);
#endif
      const double bytes = (double)b.ptiles.size() * (double)(1 << TB) * sizeof(T);
      const bool dlok = b.max_dl <= dl_cap;       // every patient's dP / dM tile slices fit the dl area: branch-free instantiation
      const int nold = (int)b.olist.size();
      timed(tr ? MMHN_K_PSOLVE_ADJ : MMHN_K_PSOLVE_FWD, bytes, [&]() {
#define PS2_ARGS dim3(nold), dim3(TSB), lds, stream, b.d_dJ.p, b.d_ptoff.p, b.d_ptiles.p, d_par.p, y, rhs_mode, d_perm.p, mk, tabJ.p, links.p, qS.p, dl_cap, b.d_olist.p
        if (tr) { if (dlok) hipLaunchKernelGGL((k_psolve2<T, true, true>), PS2_ARGS); else hipLaunchKernelGGL((k_psolve2<T, true, false>), PS2_ARGS); }
        else { if (dlok) hipLaunchKernelGGL((k_psolve2<T, false, true>), PS2_ARGS); else hipLaunchKernelGGL((k_psolve2<T, false, false>), PS2_ARGS); }
#undef PS2_ARGS
      });
    }
    if (!b.lmapT.empty()) {
      // everything else: all tiles in one cooperative launch (several workgroups per patient)
      const PList LT{b.d_dJ.p, nullptr, (int)b.lmapT.size(), b.maxkT, 0, b.d_lmapT.p, &b.lofT, tabJ.p, b.clJ,
                     tr ? MMHN_K_CSOLVE_ADJ : MMHN_K_CSOLVE_FWD};
      solve(tr, LT, y, nullptr, nullptr, rhs_mode, nullptr);
    }
  }

  // (D - Q)^-1 rhs (tr: transposed) on every problem of a list.
  //   default: tile-level substitution, ONE cooperative launch over the list's work list (k_csolve); a list of one level,
  //   lists without a work list (API calls) and MMHN_COOP=0: one launch per level of tile-index popcount (k_tsolve);
  //   MMHN_SOLVER=jacobi: k+1 in-place fused Jacobi sweeps from zero (the reference's iteration).
  void solve(bool tr, const PList& L, T* y, const T* lidg, const T* rhs, int rhs_mode, const T* scal) {
    if (L.ntiles == 0) return;
    if (use_jacobi) {
      zero(y, L.vec);
      const double bytes = 4.0 * (double)L.vec * sizeof(T);   // read y, lidg, rhs; write y (B_js)
      for (int s = 0; s <= L.maxk; ++s)
        launch_sweep(tr, L.d, L.map, L.ntiles, L.maxk, y, y, lidg, rhs, rhs_mode, scal, bytes, L.tab);
      return;
    }
    const int nlev = (int)L.lof->size() - 1;
    const size_t lds = sweep_lds(L.maxk);
    const int mk = std::max(L.maxk, 1);
    // compulsory traffic of a tile: write y (+ read dense rhs, + read the lidg vector when there is one)
    const double per_tile = (double)((rhs_mode == 0 ? 2 : 1) + (lidg ? 1 : 0)) * (double)(1 << std::min(L.maxk, TB)) * sizeof(T);   // modes 1-3: rhs is not a 2^k vector
    if (coop && L.cl && nlev > 1) {
      const CList& cl = L.cl[tr ? 1 : 0];
      const int nitems = (int)cl.items.size();
      REQUIRE(nitems == L.ntiles, "cooperative solve: work list and tile list differ");
      DevArr<unsigned>& flags = coop_flags[cur_lane];
      if (flags.n < (size_t)nitems) {
        HIPCHECK(hipStreamSynchronize(stream));                // (a launch of this lane may still poll the old array)
        flags.alloc((size_t)nitems + 1024);
        HIPCHECK(hipMemsetAsync(flags.p, 0, flags.n * sizeof(unsigned), stream));
      }
      if (++coop_epoch == 0u) {                                // (wrapped: no stale flag may equal a future epoch)
        HIPCHECK(hipDeviceSynchronize());
        for (auto& f : coop_flags) if (f.p) HIPCHECK(hipMemset(f.p, 0, f.n * sizeof(unsigned)));
        coop_epoch = 1u;
      }
      coop_slot = (coop_slot + 1) & 3;
      const int slot = cur_lane * 4 + coop_slot;
      coop_used = true;
      timed(L.kslot, per_tile * nitems, [&]() {
        const dim3 g((unsigned)std::min(nitems, coop_wgs > 0 ? coop_wgs : n_cu)), bk(TSB);
#define CS_ARGS g, bk, lds, stream, L.d, cl.d_items.p, cl.d_deps.p, nitems, flags.p, coop_epoch, coop_ctl.p, slot, h_abort_dev, coop_fault ? 1 : 0, \
                y, lidg, rhs, rhs_mode, scal, d_perm.p, mk, L.tab, links.p, qS.p
        if (lidg) {
          if (tr) hipLaunchKernelGGL((k_csolve<T, true, true>), CS_ARGS);
          else hipLaunchKernelGGL((k_csolve<T, false, true>), CS_ARGS);
        } else if (coop_alone) {                               // (nothing runs beside the joint solves of this batch: 93 registers)
          if (tr) hipLaunchKernelGGL((k_csolve<T, true, false, 4>), CS_ARGS);
          else hipLaunchKernelGGL((k_csolve<T, false, false, 4>), CS_ARGS);
        } else {
          if (tr) hipLaunchKernelGGL((k_csolve<T, true, false>), CS_ARGS);
          else hipLaunchKernelGGL((k_csolve<T, false, false>), CS_ARGS);
        }
#undef CS_ARGS
      });
      return;
    }
    for (int s = 0; s < nlev; ++s) {
      const int lev = tr ? nlev - 1 - s : s;
      const int beg = (*L.lof)[lev], cntl = (*L.lof)[lev + 1] - beg;
      if (cntl == 0) continue;
      timed(L.kslot, per_tile * cntl, [&]() {
        const dim3 g(cntl), bk(TSB);
#define TS_ARGS L.d, L.lmap + beg, d_par.p, y, lidg, rhs, rhs_mode, scal, d_perm.p, d_lvl.p, mk, L.tab, links.p, qS.p
        if (lidg) {
          if (tr) hipLaunchKernelGGL((k_tsolve<T, true, true>), g, bk, lds, stream, TS_ARGS);
          else hipLaunchKernelGGL((k_tsolve<T, false, true>), g, bk, lds, stream, TS_ARGS);
        } else {
          if (tr) hipLaunchKernelGGL((k_tsolve<T, true, false>), g, bk, lds, stream, TS_ARGS);
          else hipLaunchKernelGGL((k_tsolve<T, false, false>), g, bk, lds, stream, TS_ARGS);
        }
#undef TS_ARGS
      });
    }
  }
  // a spin of a cooperative launch timed out (tsolve.h): the results of the evaluation are garbage - fail the call, and put
  // the words back so that the engine stays usable.  Call after the stream has been synchronised.
  bool coop_used = false;
  bool coop_alone = false;      // set per batch: no side stream carries whole-CU work next to the joint solves (tsolve.h: WPE)
  void check_abort() {
    if (!coop_used) return;
    coop_used = false;
    if (*h_abort == 0u) return;
    *h_abort = 0u;
    (void)hipMemset(coop_ctl.p, 0, sizeof(CoopCtl));
    throw Fail{"cooperative tile solve: a wait for another workgroup's tile timed out (results discarded)"};
  }

  // ---------------------------------------------------------------- cohort
  void set_cohort(const int8_t* d_, long long np, int nc) {
    REQUIRE(nc == 2 * n + 3, "dat must have 2*n_mut+3 columns");
    REQUIRE(np >= 0, "negative patient count");
    REQUIRE(!sums_pending, "mmhn_set_cohort: an evaluation begun with mmhn_cohort_sums_begin has not been collected");
    dat.assign(d_, d_ + np * nc);
    n_pat = np;
    n_cols = nc;
    batches.clear();
    n_em = 0;
    size_t max_need = 0;
    auto bytes_of = [&](const Batch& b) {
      return (size_t)(4 * b.vecJ + 4 * b.vecS + b.asize) * sizeof(T) +
             (size_t)(b.dS.size() * N * N + 3 * b.dJ.size() * N * N) * sizeof(T) +
             b.pats.size() * (size_t)stride() * sizeof(double);
    };
    Batch cur;
    auto flush = [&]() {
      if (cur.pats.empty()) return;
      max_need = std::max(max_need, bytes_of(cur));
      batches.push_back(std::move(cur));
      cur = Batch();
    };
    // a cohort that does not fit the workspace is cut into batches of about equal size (not: full ones and a small
    // remainder - every batch should be large enough for the per-patient kernels): soft target per batch
    double soft_target = 0;
    long long pat_target = 0;
    {
      double total = 0;
      long long npJ = 0;
      for (long long r = 0; r < np; ++r) {
        const int8_t* row = dat.data() + r * nc;
        int kp = 0, km = 0, ke = 0;
        for (int j = 0; j < n; ++j) { kp += row[2 * j] != 0; km += row[2 * j + 1] != 0; ke += row[2 * j] && row[2 * j + 1]; }
        const int type = row[nc - 1];
        double el = 0;
        if (type == 3) {
          ++npJ;
          el = (use_jacobi || wsolve_mode == 2 ? 4.0 : 2.0) * std::ldexp(1.0, kp + km + 1) + 4.0 * (std::ldexp(1.0, km + 1) + std::ldexp(1.0, kp + 1)) +
               (kp + 1) * std::ldexp(1.0, kp) + (km + 1) * std::ldexp(1.0, km) + (ke + 2) * std::ldexp(1.0, ke);
        } else {
          el = 4.0 * std::ldexp(1.0, (type == 2 ? km : kp) + 1);
        }
        total += el * sizeof(T);
      }
      const double nb = std::ceil(total / (double)std::max<size_t>(ws_limit, 1));
      soft_target = nb > 1 ? total / nb * 1.02 : 0;
      // the per-patient kernels take one patient per CU at a time: a batch of 400 costs two rounds of 256.  Batches of a cut
      // cohort hold a multiple of the CU count (rounded down: more, fuller rounds)
      if (nb > 1 && npJ > 0) {
        const double per_batch = (double)ws_limit / (total / (double)npJ);     // paired rows (they are what occupies a CU)
        if (per_batch >= (double)n_cu) {
          pat_target = (long long)(per_batch / n_cu) * n_cu;
          // a remainder too small for the window / per-patient routes would be a batch on the tile route alone: spread the
          // rows evenly over the same number of batches instead
          const long long nbt = (npJ + pat_target - 1) / pat_target, rem = npJ - (nbt - 1) * pat_target;
          if (nbt > 1 && rem < std::max(psolve_min, wsolve_min)) pat_target = (npJ + nbt - 1) / nbt;
        }
      }
    }
    std::vector<int8_t> st(2 * n + 2);
    for (long long r = 0; r < np; ++r) {
      const int8_t* row = dat.data() + r * nc;
      const int type = row[nc - 1];
      const int order = row[nc - 2];
      REQUIRE(type >= 0 && type <= 3, "dat: type column must be 0..3");
      for (int c = 0; c < 2 * n + 1; ++c) REQUIRE(row[c] == 0 || row[c] == 1, "dat: event columns must be 0/1");
      n_em += row[2 * n];
      // problems of this patient
      PatRec pr{};
      pr.kind = type; pr.order = (order == 0 || order == 1) ? order : 2; pr.j = -1; pr.s[0] = pr.s[1] = -1;
      pr.row = (int)r;
      Desc dj{}, ds0{}, ds1{};
      bool hasJ = false, has0 = false, has1 = false;
      if (type == 0 || type == 1) {
        for (int j = 0; j <= n; ++j) st[j] = row[2 * j];          // PT slots + seeding (regularized_optimization.py:189)
        int np_ = 0;
        for (int j = 0; j <= n; ++j) np_ += st[j];
        if (type == 0 && np_ == 0) pr.kind = 4;
        else { ds0 = make_single(st.data(), n, PS_PRIM, OBS_ONE); has0 = true; }
      } else if (type == 2) {
        for (int j = 0; j < n; ++j) st[j] = row[2 * j + 1];       // MT slots, seeding = 1 (:216-219)
        st[n] = 1;
        ds0 = make_single(st.data(), n, PS_THETA, OBS_MET);
        has0 = true;
      } else {
        REQUIRE(row[2 * n] == 1, "dat: paired rows (type 3) must have seeding = 1");
        dj = make_joint(row, n);
        hasJ = true;
        if (pr.order == 0 || pr.order == 1) {                     // PT observed first -> MT marginal
          for (int j = 0; j < n; ++j) st[j] = row[2 * j + 1];
          st[n] = 1;
          ds0 = make_single(st.data(), n, PS_MET, OBS_ONE);
          has0 = true;
        }
        if (pr.order != 1) {                                      // MT observed first -> PT marginal
          for (int j = 0; j <= n; ++j) st[j] = row[2 * j];
          ds1 = make_single(st.data(), n, PS_PRIM, OBS_ONE);
          has1 = true;
        }
      }
      REQUIRE(!hasJ || dj.k <= MAXK, "too many active events for one patient");
      // would the batch overflow the workspace?  State vectors, class-marginal arrays, the per-evaluation tables
      // (incl. the incoming problems') and the per-problem / per-patient result buffers
      const long long nvJ = hasJ ? (1ll << dj.k) : 0, nvS = (has0 ? (1ll << ds0.k) : 0) + (has1 ? (1ll << ds1.k) : 0);
      const long long ntab = (hasJ ? table_size(dj) : 0) + (has0 ? table_size(ds0) : 0) + (has1 ? table_size(ds1) : 0);
      auto footprint = [&](long long vJ, long long vS, long long as, long long tabs, size_t nJp, size_t nSp, size_t npat) {
        const size_t small = (nSp * (size_t)(N * N + 64 + 1) + nJp * (size_t)(3 * N * N + 3 * N + 64)) * sizeof(T) +
                             nJp * sizeof(JLink<T>) + npat * ((size_t)stride() + 1) * sizeof(double) + npat * 2 * sizeof(T);
        return (size_t)((use_jacobi || wsolve_mode == 2 ? 4 : 2) * vJ + 4 * vS + as + tabs) * sizeof(T) + small;
      };
      const size_t need = footprint(cur.vecJ + nvJ, cur.vecS + nvS, cur.asize + (hasJ ? a_size(dj) : 0), cur.tabJ + cur.tabS + ntab,
                                    cur.dJ.size() + (hasJ ? 1 : 0), cur.dS.size() + (has0 ? 1 : 0) + (has1 ? 1 : 0), cur.pats.size() + 1);
      const size_t have = footprint(cur.vecJ, cur.vecS, cur.asize, cur.tabJ + cur.tabS, cur.dJ.size(), cur.dS.size(), cur.pats.size());
      if (!cur.pats.empty() && (need > ws_limit || (pat_target > 0 ? (hasJ && (long long)cur.dJ.size() >= pat_target)
                                                                         : (soft_target > 0 && (double)have >= soft_target)))) flush();
      if (hasJ) {
        dj.off = cur.vecJ; dj.aoff = cur.asize; dj.toff = cur.tabJ; cur.tabJ += table_size(dj);
        cur.vecJ += 1ll << dj.k; cur.asize += a_size(dj);
        pr.j = (int)cur.dJ.size();
        add_tiles(cur.mapJ, pr.j, dj.k);
        cur.maxkJ = std::max(cur.maxkJ, dj.k);
        cur.maxkcJ = std::max(cur.maxkcJ, std::max(popc(dj.maskP), popc(dj.maskM)));
        cur.dJ.push_back(dj);
      }
      if (has0) {
        ds0.off = cur.vecS; cur.vecS += 1ll << ds0.k; ds0.toff = cur.tabS; cur.tabS += table_size(ds0);
        pr.s[0] = (int)cur.dS.size();
        add_tiles(cur.mapS, pr.s[0], ds0.k);
        cur.maxkS = std::max(cur.maxkS, ds0.k);
        cur.dS.push_back(ds0);
      }
      if (has1) {
        ds1.off = cur.vecS; cur.vecS += 1ll << ds1.k; ds1.toff = cur.tabS; cur.tabS += table_size(ds1);
        pr.s[1] = (int)cur.dS.size();
        add_tiles(cur.mapS, pr.s[1], ds1.k);
        cur.maxkS = std::max(cur.maxkS, ds1.k);
        cur.dS.push_back(ds1);
      }
      if (pr.kind == 2) cur.has_kind2 = true;
      cur.pats.push_back(pr);
    }
    flush();
    // upload the static descriptions and size the workspace
    long long mvJ = 0, mvS = 0, mtJ = 0, mtS = 0, mvM = 0;
    size_t mZ = 0;
    int bid = 0;
    size_t mnJ = 0, mnS = 0, mp = 0;
    for (auto& b : batches) {
      b.id = bid++;
      b.d_pats.alloc(b.pats.size());
      HIPCHECK(hipMemcpy(b.d_pats.p, b.pats.data(), b.pats.size() * sizeof(PatRec), hipMemcpyHostToDevice));
      auto up = [&](auto& dev, auto& host) {
        if (host.empty()) return;
        dev.alloc(host.size());
        HIPCHECK(hipMemcpy(dev.p, host.data(), host.size() * sizeof(host[0]), hipMemcpyHostToDevice));
      };
      build_levels(b.mapJ, &b.dJ, !use_jacobi, b.lmapJ, b.lofJ);       // (every joint tile by level: the Jacobi solver's list)
      const int nJ = (int)b.dJ.size();
      // ---- routes of the joint problems (Batch::route)
      auto is_multi = [](const Desc& dj) { return dj.seedbit >= TB && popc(dj.pairP) <= TB; };   // what k_psolve2 takes
      b.route.assign((size_t)nJ, RT_T);
      b.olist.clear(); b.wd.clear(); b.wchains.clear();
      b.wpath = false;
      if (!use_jacobi && nJ > 0) {
        int nW = 0, nP = 0;
        if (wsolve_mode != 0) for (const Desc& dj : b.dJ) nW += window_ok<T>(dj) ? 1 : 0;
        b.wpath = nW > 0 && nW >= wsolve_min;
        for (int pj = 0; pj < nJ; ++pj) {
          if (b.wpath && window_ok<T>(b.dJ[pj])) b.route[pj] = RT_W;
          else if (is_multi(b.dJ[pj])) ++nP;
        }
        if (nP > 0 && nP >= psolve_min)
          for (int pj = 0; pj < nJ; ++pj) if (b.route[pj] != RT_W && is_multi(b.dJ[pj])) b.route[pj] = RT_P;
        for (int pj = 0; pj < nJ; ++pj) {
          if (b.route[pj] == RT_W) b.wd.push_back(make_wdesc<T>(b.dJ[pj], pj));
          else if (b.route[pj] == RT_P) b.olist.push_back(pj);
        }
      }
      {
        // layout of the joint vectors (the offsets are the engine's own business: every kernel goes through Desc::off):
        // the window problems sorted by shape - a chain of same-shape patients is one contiguous buffer -, then the
        // per-patient problems, then the tile route's, whose dead tiles must stay zero
        std::stable_sort(b.wd.begin(), b.wd.end(), [](const WDesc& x, const WDesc& y) { return x.kR != y.kR ? x.kR < y.kR : x.kC < y.kC; });
        long long off = 0;
        for (const WDesc& w : b.wd) { b.dJ[w.prob].off = off; off += 1ll << b.dJ[w.prob].k; }
        for (int pj : b.olist) { b.dJ[pj].off = off; off += 1ll << b.dJ[pj].k; }
        b.offT = off;
        for (int pj = 0; pj < nJ; ++pj) if (b.route[pj] == RT_T) { b.dJ[pj].off = off; off += 1ll << b.dJ[pj].k; }
        REQUIRE(off == b.vecJ, "offsets of the joint problems do not add up");
      }
      if (b.wpath) {
        // chains (wsolve.h): about one run of same-shape problems per workgroup
        const int nW = (int)b.wd.size();
        b.wnx = b.wd[0].nXc + b.wd[0].nXr;
        for (const WDesc& w : b.wd) if (w.nXc + w.nXr != b.wnx) b.wnx = -1;
        b.wsplit = wsplit && b.wnx == WCfg<T>::NXT && WCfg<T>::NXT >= 4;
        const int groups = b.wsplit ? wsplit_groups() : std::max(1, wsolve_wgs > 0 ? wsolve_wgs : n_cu);
        const int per = (nW + groups - 1) / groups;
        for (int i0 = 0; i0 < nW;) {
          const WDesc& w = b.wd[i0];
          const int k = b.dJ[w.prob].k;
          // a chain keeps two patients' tables alive at once: it needs 2^nX > 6 passes per patient; its span must stay
          // below 2 GB (32-bit buffer offsets)
          const int maxlen = (w.nXc + w.nXr >= 3 && wsolve_chain) ? (int)std::min<long long>(per, (1ll << 31) / ((long long)sizeof(T) << k) - 1) : 1;
          int len = 1;
          while (len < maxlen && i0 + len < nW && b.wd[i0 + len].kR == w.kR && b.wd[i0 + len].kC == w.kC && (i0 + len) % per != 0) ++len;
          b.wchains.push_back(WChain{i0, len});
          i0 += len;
        }
        if ((int)b.wchains.size() > groups) {
          // more chains than workgroups (shape changes cut chains short): workgroup g takes the entries g, g + groups, ... -
          // deal the chains longest first to the least loaded workgroup, empty entries fill the rounds
          std::vector<WChain> srt = b.wchains;
          std::stable_sort(srt.begin(), srt.end(), [](const WChain& x, const WChain& y) { return x.count > y.count; });
          std::vector<std::vector<WChain>> mine((size_t)groups);
          std::vector<int> load((size_t)groups, 0);
          for (const WChain& c : srt) {
            const int g = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            mine[g].push_back(c);
            load[g] += c.count;
          }
          size_t rounds = 0;
          for (const auto& m : mine) rounds = std::max(rounds, m.size());
          b.wchains.assign(rounds * (size_t)groups, WChain{0, 0});
          for (int g = 0; g < groups; ++g)
            for (size_t r = 0; r < mine[g].size(); ++r) b.wchains[r * (size_t)groups + g] = mine[g][r];
        }
        up(b.d_wd, b.wd);
        up(b.d_wchains, b.wchains);
      }
      up(b.d_olist, b.olist);
      // the tile route: every tile of its problems in one cooperative launch (or level by level)
      b.lmapT.clear(); b.lofT.clear(); b.maxkT = 0;
      b.clJ[0].clear(); b.clJ[1].clear();
      if (!use_jacobi) {
        std::vector<int2> mt;
        for (const int2& m : b.mapJ) if (b.route[m.x] == RT_T) { mt.push_back(m); b.maxkT = std::max(b.maxkT, b.dJ[m.x].k); }
        if (!mt.empty()) {
          build_levels(mt, &b.dJ, true, b.lmapT, b.lofT);
          up(b.d_lmapT, b.lmapT);
          if (b.lofT.size() > 2) { build_clist(mt, b.dJ, true, false, b.clJ[0]); build_clist(mt, b.dJ, true, true, b.clJ[1]); }
        }
      }
      b.mapX.clear();
      for (const int2& m : b.mapJ) {
        const Desc& dj = b.dJ[m.x];
        if (popc(dj.maskP) > PCA + PCH || popc(dj.maskM) > PCA + PCH) b.mapX.push_back(m);
      }
      {
        // k_psolve2 solves the seed = 0 part of a multi-tile space (only its PT == MT states carry values) as a small
        // lattice over the paired events, so only the seeded tiles of its problems are listed
        b.ptoff.assign(1, 0);
        b.ptiles.clear();
        b.maxkP = 0; b.max_dl = 0;
        for (int pj = 0; pj < nJ; ++pj) {
          const Desc& dj = b.dJ[pj];
          if (b.route[pj] == RT_P) {
            const uint32_t nt = 1u << (dj.k - TB);
            for (uint32_t Ht = 0; Ht < nt; ++Ht) if (((Ht << TB) >> dj.seedbit) & 1u) b.ptiles.push_back(Ht);
            b.maxkP = std::max(b.maxkP, dj.k);
            b.max_dl = std::max(b.max_dl, (1 << popc(dj.maskP & ((1u << TB) - 1u))) + (1 << popc(dj.maskM & ((1u << TB) - 1u))));
          }
          b.ptoff.push_back((int)b.ptiles.size());
        }
      }
      // class marginals of the problems in index order as work items (short launches; a large problem is several
      // workgroups): per problem its eq block, per class pass the outer loop cut into ranges
      b.pcl.clear();
      if (!use_jacobi && nJ <= prep_split_max) {
        for (int pj = 0; pj < nJ; ++pj) {
          const Desc& dj = b.dJ[pj];
          b.pcl.push_back(int4{pj, 2, 0, 0});
          const int kP = popc(dj.maskP), kM = popc(dj.maskM);
          if (b.route[pj] == RT_W && wsolve_mode != 2) continue;             // (k_wclass)
          if (kP > PCA + PCH || kM > PCA + PCH) continue;                     // (k_class_marg)
          for (int c = 0; c < 2; ++c) {
            const int kc = c == 0 ? kP : kM, kf = c == 0 ? kM : kP;
            const int no = pclass_outer_bits(kc, kf), nh = kc > PCA ? kc - PCA : 0;
            const int per = std::max(1, pcl_per >> nh);                       // tiles per item: (2^nh class blocks) x (per settings)
            for (int o0 = 0; o0 < (1 << no); o0 += per) b.pcl.push_back(int4{pj, c, o0, std::min(o0 + per, 1 << no)});
          }
        }
        // longest items first (tiles of the item = range x class blocks above the tile): the launch ends with the short ones
        auto item_tiles = [&](const int4& it) -> long long {
          if (it.y == 2) return 0;
          const Desc& dj = b.dJ[it.x];
          const int kc = popc(it.y == 0 ? dj.maskP : dj.maskM);
          return (long long)(it.w - it.z) << (kc > PCA ? kc - PCA : 0);
        };
        std::stable_sort(b.pcl.begin(), b.pcl.end(), [&](const int4& x, const int4& y) { return item_tiles(x) > item_tiles(y); });
        up(b.d_pcl, b.pcl);
      }
      build_levels(b.mapS, nullptr, false, b.lmapS, b.lofS);
      b.paired.clear();
      for (size_t pi_ = 0; pi_ < b.pats.size(); ++pi_) if (b.pats[pi_].j >= 0) b.paired.push_back((int)pi_);
      up(b.d_paired, b.paired);
      // ---- single-tumour problems: small-space path for the patients whose spaces all fit a tile, staged kernels for the rest
      for (int w = 0; w < 3; ++w) for (int c = 0; c < SP_NCLASS; ++c) b.sp_list[w][c].clear();
      for (int w = 0; w < 2; ++w) { Staged& g = b.stg[w]; g.pats.clear(); g.paired.clear(); g.probs.clear(); g.kind2 = false; }
      bool class_fits[SP_NCLASS];
      for (int c = 0; c < SP_NCLASS; ++c)
        class_fits[c] = (spatient_lds<T>(N, spatient_class_maxk(c)) + 15) / 16 * 16 * (size_t)(c == 0 ? SP_PPB0 : c == 1 ? 2 : 1) + 64 <= (size_t)160 * 1024;
      const bool all_staged = !small_path || use_jacobi;
      for (size_t pi_ = 0; pi_ < b.pats.size(); ++pi_) {
        const PatRec& pr = b.pats[pi_];
        int ks = -1;
        for (int part = 0; part < 2; ++part) if (pr.s[part] >= 0) ks = std::max(ks, b.dS[pr.s[part]].k);
        if (ks < 0) continue;
        int c = 0;
        while (c < SP_NCLASS - 1 && ks > spatient_class_maxk(c)) ++c;
        if (all_staged || ks > TB || !class_fits[c]) {
          // (MMHN_SMALL=0 / Jacobi: one group on the main stream - its solver clears and sweeps whole buffers)
          Staged& g = b.stg[all_staged || pr.j >= 0 ? 1 : 0];
          g.pats.push_back((int)pi_);
          if (pr.j >= 0) g.paired.push_back((int)pi_);
          for (int part = 0; part < 2; ++part) if (pr.s[part] >= 0) g.probs.push_back(pr.s[part]);
          if (pr.kind == 2) g.kind2 = true;
          continue;
        }
        const bool side_by_side = pair_small && pr.j >= 0 && pr.s[0] >= 0 && pr.s[1] >= 0 && c < 2;
        b.sp_list[side_by_side ? 2 : pr.j >= 0 ? 1 : 0][c].push_back((int)pi_);
      }
      // paired rows of the 1024-thread class: when they are few and of at most 10 bits they ride in the merged 256-thread
      // launch (its LDS sized for them) - on a small cohort their own launch is a side stream, a fork and a join
      // (~20 us of queue latency) for a handful of patients
      b.mk1p = spatient_class_maxk(1);
      if (pair_small && !b.sp_list[1][2].empty() && b.sp_list[1][2].size() <= 16) {
        int mk = 0;
        for (int pi_ : b.sp_list[1][2])
          for (int part = 0; part < 2; ++part) if (b.pats[pi_].s[part] >= 0) mk = std::max(mk, b.dS[b.pats[pi_].s[part]].k);
        if (mk <= 10 && (spatient_lds<T>(N, mk) + 15) / 16 * 16 * 2 + 64 <= (size_t)160 * 1024) {
          for (int pi_ : b.sp_list[1][2]) b.sp_list[b.pats[pi_].s[0] >= 0 && b.pats[pi_].s[1] >= 0 ? 2 : 1][1].push_back(pi_);
          b.sp_list[1][2].clear();
          b.mk1p = mk;
        }
      }
      {
        // largest spaces first: the waves of one workgroup (class 0: one patient each) then finish together and the
        // long patients do not form the tail of the launch
        auto ksize = [&](int pi_) {
          const PatRec& pr = b.pats[pi_];
          int ks = 0;
          for (int part = 0; part < 2; ++part) if (pr.s[part] >= 0) ks = std::max(ks, 64 * b.dS[pr.s[part]].k + (pr.s[0] >= 0 && pr.s[1] >= 0 ? 32 : 0) + pr.kind);
          return ks;
        };
        for (int w = 0; w < 3; ++w)
          for (int c = 0; c < SP_NCLASS; ++c)
            std::stable_sort(b.sp_list[w][c].begin(), b.sp_list[w][c].end(), [&](int x, int y) { return ksize(x) > ksize(y); });
      }
      b.has_small = false;
      for (int w = 0; w < 3; ++w)
        for (int c = 0; c < SP_NCLASS; ++c) {
          if (!b.sp_list[w][c].empty()) b.has_small = true;
          up(b.d_sp_list[w][c], b.sp_list[w][c]);
        }
      for (int w = 0; w < 2; ++w) {
        // lists of the staged kernels
        Staged& g = b.stg[w];
        std::vector<char> isg(b.dS.size(), 0);
        for (int sp : g.probs) isg[sp] = 1;
        g.map.clear(); g.maxk = 0;
        for (const int2& m : b.mapS) if (isg[m.x]) { g.map.push_back(m); g.maxk = std::max(g.maxk, b.dS[m.x].k); }
        build_levels(g.map, nullptr, false, g.lmap, g.lof);
        g.cl[0].clear(); g.cl[1].clear();
        if (!use_jacobi && g.lof.size() > 2) { build_clist(g.map, b.dS, false, false, g.cl[0]); build_clist(g.map, b.dS, false, true, g.cl[1]); }
        up(g.d_pats, g.pats); up(g.d_paired, g.paired); up(g.d_probs, g.probs);
        up(g.d_map, g.map); up(g.d_lmap, g.lmap);
        std::vector<int2> gc;
        for (const int2& e : grad_chunks(b.dS, GK_S)) if (isg[e.x]) gc.push_back(e);
        up(g.d_grc, gc);
      }
      // the window layout stays in place: every consumer of the joint vectors reads it there (k_gather_marg / the small-space
      // kernels, k_eq_flows, k_wclass); MMHN_WSOLVE=2 converts to index order after each solve instead (k_wconvert)
      b.wdirect = b.wpath && wsolve_mode != 2;
      for (Desc& dj : b.dJ) dj.wl = -1;
      if (b.wdirect) for (size_t i = 0; i < b.wd.size(); ++i) b.dJ[b.wd[i].prob].wl = (int)i;
      up(b.d_dJ, b.dJ); up(b.d_dS, b.dS); up(b.d_mapJ, b.mapJ); up(b.d_mapS, b.mapS);
      up(b.d_lmapJ, b.lmapJ); up(b.d_lmapS, b.lmapS);
      up(b.d_ptoff, b.ptoff); up(b.d_ptiles, b.ptiles); up(b.d_mapX, b.mapX);
      {
        std::vector<int2> all;
        for (int kd = 0; kd < 4; ++kd) {
          std::vector<int2> gc = grad_chunks(kd == GK_S ? b.dS : b.dJ, kd);
          up(b.d_grc[kd], gc);
          if (kd != GK_S) for (int2 e : gc) all.push_back(int2{e.x, e.y | (kd << 24)});
        }
        up(b.d_grcJ, all);
      }
      mvJ = std::max(mvJ, b.vecJ); mvS = std::max(mvS, b.vecS);
      mZ = std::max<size_t>(mZ, (size_t)zarena_elems((long long)b.dJ.size(), b.asize, N));
      mtJ = std::max(mtJ, b.tabJ); mtS = std::max(mtS, b.tabS);
      if (b.wpath && !b.wdirect) mvM = std::max(mvM, b.vecJ);
      mnJ = std::max(mnJ, b.dJ.size()); mnS = std::max(mnS, b.dS.size()); mp = std::max(mp, b.pats.size());
    }
    pi.alloc(mvJ); qJ.alloc(mvJ);
    piM.alloc(mvM); qM.alloc(mvM);
    if (use_jacobi) { lidgJ.alloc(mvJ); rhsJ.alloc(mvJ); }
    links.alloc(std::max<size_t>(mnJ, 1));
    tabJ.alloc(mtJ); tabS.alloc(mtS);
    pi_owner = qJ_owner = -1;
    rhsS.alloc(mvS); pS.alloc(mvS); lidgS.alloc(mvS); qS.alloc(mvS);
    seedS.alloc(mnS);
    GS.alloc(mnS * N * N); zarena.alloc(mZ);
    dots.alloc(2 * mp); bmJ.alloc(mnJ * 64); bmS.alloc(mnS * 64);
    lp.alloc(mp); out.alloc(mp * stride()); redbuf.alloc(((mp + red_per((int)mp) - 1) / red_per((int)mp)) * 2 * (size_t)stride());
  }

  // ---------------------------------------------------------------- one evaluation
  // host_out (optional): per-patient rows [n_pat][stride]; sums: [2][stride] (EM, NM)
  // small-space path (small.h): every single-tumour space of the batch fits one tile -> one launch per size class does
  // stage 4, the adjoint seeds, the single-tumour gradients and the <q, rhs> dots, one workgroup (or wave) per patient.
  // which = 0: the patients that are their own problem (launched first: nothing of the joint path feeds them), 1: the
  // paired ones (after k_gather_marg).  The size classes are independent of each other.
  // Streams: the own-problem launches go to side[0], next to the joint path on the main stream, and are joined before the
  // assembly; of the paired launches the 1024-thread class goes to side[1], the merged one stays on the main stream.
  bool small_forked[2] = {false, false}, fork_recorded[2] = {false, false};
  // the point of the main stream the side launches of `which` wait for (recorded ahead of time when launches of the
  // critical chain are to be issued first: the host needs ~25 us for a fork, the launches and the join record)
  void small_fork(int which) {
    HIPCHECK(hipEventRecord(ev_fork[which], stream));
    fork_recorded[which] = true;
  }
  void small_classes(const Batch& b, int which, bool grad) {
    const int n0 = (int)b.sp_list[which][0].size(), n1 = (int)b.sp_list[which][1].size(), n2 = (int)b.sp_list[which][2].size();
    const int np0 = which ? (int)b.sp_list[2][0].size() : 0, np1 = which ? (int)b.sp_list[2][1].size() : 0;
    const int mk0 = spatient_class_maxk(0), mk1 = which ? b.mk1p : spatient_class_maxk(1), mk2 = spatient_class_maxk(2);
    const bool on_side = which == 0 ? (n0 + n1 + n2 > 0) : n2 > 0;
    hipStream_t sd = side[which];
    if (on_side) {
      if (!fork_recorded[which]) HIPCHECK(hipEventRecord(ev_fork[which], stream));
      HIPCHECK(hipStreamWaitEvent(sd, ev_fork[which], 0));
    }
    fork_recorded[which] = false;
#define SP_TAIL b.d_pats.p, b.d_dS.p, d_par.p, d_perm.p, d_lvl.p, b.d_dJ.p, b.d_wd.p, pi.p, links.p, pS.p, qS.p, GS.p, bmS.p, dots.p, lp.p
    auto big_class = [&]() {
      if (!n2) return;
      const size_t lds = (spatient_lds<T>(N, mk2) + 15) / 16 * 16;
      hipLaunchKernelGGL((k_spatient<T, 1024, 1>), dim3(n2), dim3(1024), lds, sd, b.d_sp_list[which][2].p, n2, SP_TAIL, mk2, N, grad ? 1 : 0);
      HIPCHECK(hipGetLastError());
    };
    big_class();
    if (n0 + n1 + np0 + np1) {
      const size_t s0 = (spatient_lds<T>(N, mk0) + 15) / 16 * 16, s1 = (spatient_lds<T>(N, mk1) + 15) / 16 * 16;
      const size_t lds = std::max(s0 * SP_PPB0, n1 + np1 ? s1 * (np1 ? 2 : 1) : 0) + 64;
      const int nblk = (n0 + SP_PPB0 - 1) / SP_PPB0 + n1 + (np0 + SP_PPB0 / 2 - 1) / (SP_PPB0 / 2) + np1;
      hipLaunchKernelGGL((k_spatient2<T>), dim3(nblk), dim3(256), lds, which == 0 ? sd : stream,
                         b.d_sp_list[which][0].p, n0, mk0, b.d_sp_list[which][1].p, n1, mk1,
                         b.d_sp_list[2][0].p, np0, b.d_sp_list[2][1].p, np1, SP_TAIL, N, grad ? 1 : 0);
      HIPCHECK(hipGetLastError());
    }
#undef SP_TAIL
    if (on_side) HIPCHECK(hipEventRecord(ev_join[which], sd));
    small_forked[which] = on_side;                           // (the caller joins: small_join)
  }
  void small_join(int which) {
    if (small_forked[which]) HIPCHECK(hipStreamWaitEvent(stream, ev_join[which], 0));
    small_forked[which] = false;
  }

  void evaluate(const double* lt, const double* ldp, const double* ldm, bool grad, double* host_sums,
                double* host_out) {
    REQUIRE(!sums_pending, "an evaluation begun with mmhn_cohort_sums_begin has not been collected");
    auto t0 = std::chrono::steady_clock::now();
    const int st = stride();
    // (an evaluation that threw between a fork and its join must not leave its flags to the next one: the side stream
    // would wait for the previous evaluation's event and start before this one's parameters are up)
    small_forked[0] = small_forked[1] = fork_recorded[0] = fork_recorded[1] = false;
    // the head of the evaluation: parameters up, cohort sums and the first batch's gradient work arrays cleared
    bool head_done = false;
    if (zero_copy && !batches.empty()) {
      build_params(lt, ldp, ldm, false);
      const Batch& b0 = batches.front();
      // (a large batch clears its ~GB of work arrays with the runtime's fill, which is faster at that size: +1.0 ms
      // per evaluation on the 5 000-patient bench cohort when this kernel did it)
      long long nz = grad && !b0.dJ.empty() ? zarena_elems((long long)b0.dJ.size(), b0.asize, N) * (long long)sizeof(T) / 16 : 0;
      const bool fill_here = nz <= (32ll << 20) / 16;
      if (!fill_here) nz = 0;
      const int nw = (int)(NPSET * sizeof(Params<T>) / sizeof(uint4));
      const long long need = std::max<long long>(std::max<long long>(nw, 2 * st), nz);
      const int nblk = (int)std::min<long long>((need + 255) / 256, 4096);
      hipLaunchKernelGGL(k_begin_eval, dim3(nblk), dim3(256), 0, stream, static_cast<const uint4*>(h_par_dev),
                         reinterpret_cast<uint4*>(d_par.p), nw, sums.p, 2 * st, reinterpret_cast<uint4*>(zarena.p), nz);
      HIPCHECK(hipGetLastError());
      head_done = fill_here;
    } else {
      build_params(lt, ldp, ldm);
      HIPCHECK(hipMemsetAsync(sums.p, 0, 2 * st * sizeof(double), stream));
    }
    for (Batch& b : batches) {
      const int npat = (int)b.pats.size(), nJ = (int)b.dJ.size(), nS = (int)b.dS.size();
      const int tJ = (int)b.mapJ.size();
      const PList LJ{b.d_dJ.p, b.d_mapJ.p, tJ, b.maxkJ, b.vecJ, b.d_lmapJ.p, &b.lofJ, tabJ.p};       // (Jacobi solver only)
      const bool fused_small = b.has_small, staged = !b.stg[0].empty() || !b.stg[1].empty();
      coop_alone = b.stg[0].empty();                          // (no second cooperative launch on a side stream next to the joint solves)
      // the staged kernels of one group of patients (Batch::stg): forward part (tables, right-hand sides, 1/diag, forward solve,
      // scores and adjoint seeds) and gradient part (adjoint solve, gradient rows, observation-rate marginals)
      auto staged_fwd = [&](const Staged& g) {
        if (g.empty()) return;
        const int nG = (int)g.pats.size(), tG = (int)g.map.size();
        // (Jacobi / MMHN_SMALL=0: the group is every single-tumour problem, so that vec = every single-tumour vector)
        const PList LG{b.d_dS.p, g.d_map.p, tG, g.maxk, b.vecS, g.d_lmap.p, &g.lof, tabS.p, g.cl};
        // (the group's accumulators cleared, the e_0 right-hand sides written: before anything of the group runs)
        hipLaunchKernelGGL((k_staged_init<T>), dim3(nG), dim3(BLOCK), 0, stream, b.d_pats.p, b.d_dS.p, rhsS.p, GS.p, bmS.p, N, grad ? 1 : 0, g.d_pats.p);
        HIPCHECK(hipGetLastError());
        if (g.probs.size() >= 256)
          hipLaunchKernelGGL((k_prep<T, false, 1024>), dim3((unsigned)g.probs.size()), dim3(1024), prep_lds<T>(N), stream, b.d_dS.p, d_par.p, tabS.p, g.d_probs.p);
        else
          hipLaunchKernelGGL((k_prep<T, false>), dim3((unsigned)g.probs.size()), dim3(BLOCK), prep_lds<T>(N), stream, b.d_dS.p, d_par.p, tabS.p, g.d_probs.p);
        HIPCHECK(hipGetLastError());
        // marginal right-hand sides (the small-space kernels read pi themselves and write the links)
        if (!g.paired.empty()) {
          hipLaunchKernelGGL((k_gather_marg<T>), dim3((unsigned)g.paired.size(), 2, g.maxk > 10 ? 8 : 1), dim3(BLOCK), 0, stream,
                             b.d_pats.p, b.d_dJ.p, b.d_dS.p, d_par.p, pi.p, rhsS.p, links.p, g.d_paired.p, b.d_wd.p);
          HIPCHECK(hipGetLastError());
        }
        launch_diag(b.d_dS.p, g.d_map.p, tG, nullptr, lidgS.p, nullptr, KD_LIDG);
        solve(false, LG, pS.p, lidgS.p, rhsS.p, 0, nullptr);
        hipLaunchKernelGGL((k_seeds<T>), dim3((nG + 255) / 256), dim3(256), 0, stream, b.d_pats.p, nG, b.d_dS.p,
                           d_par.p, pS.p, seedS.p, lp.p, g.d_pats.p);
        HIPCHECK(hipGetLastError());
      };
      auto staged_adj = [&](const Staged& g) {
        if (g.empty()) return;
        const int tG = (int)g.map.size();
        const PList LG{b.d_dS.p, g.d_map.p, tG, g.maxk, b.vecS, g.d_lmap.p, &g.lof, tabS.p, g.cl};
        solve(true, LG, qS.p, lidgS.p, nullptr, 1, seedS.p);
        launch_grad_rows(b.d_dS.p, (int)g.probs.size(), g.maxk, nullptr, pS.p, qS.p, GS.p, GK_S, g.d_grc);
        if (g.kind2) {
          hipLaunchKernelGGL((k_bit_marg<T>), dim3(tG), dim3(BLOCK), 0, stream, b.d_dS.p, g.d_map.p, d_par.p, pS.p, qS.p, bmS.p);
          HIPCHECK(hipGetLastError());
        }
      };
      time_kernels = force_timing || b.vecJ + b.vecS >= (1ll << 26);
      const long long gjs = (long long)nJ * N * N;
      GJ.p = zarena.p;
      DJ.p = GJ.p + up4(3 * gjs);
      Abuf.p = DJ.p + up4(3ll * nJ * N);
      if (grad && nJ && !(head_done && &b == &batches.front())) zero(zarena.p, zarena_elems(nJ, b.asize, N));
      prep(b.d_dJ.p, nJ, tabJ.p, true, b.maxkcJ);                    // (first: the head of the critical chain)
      // staged patients that are their own problem: a side stream of their own from here to the assembly (a timed
      // evaluation keeps them on the main stream - events are recorded there)
      bool own_forked = false;
      if (!b.stg[0].empty() && !time_kernels) {
        HIPCHECK(hipEventRecord(ev_fork[2], stream));
        HIPCHECK(hipStreamWaitEvent(side[2], ev_fork[2], 0));
        hipStream_t keep = stream;
        stream = side[2]; cur_lane = 1;
        try {
          staged_fwd(b.stg[0]);
          if (grad) staged_adj(b.stg[0]);
        } catch (...) { stream = keep; cur_lane = 0; throw; }
        stream = keep; cur_lane = 0;
        HIPCHECK(hipEventRecord(ev_join[2], side[2]));
        own_forked = true;
      }
      // patients that are their own single-tumour problem need nothing of the joint path: their small-space kernels
      // run on a side stream from here on, next to the joint forward solve (whose launch is issued first - it is the
      // critical chain); the assembly waits for them
      if (fused_small) small_fork(0);
      // The tile solver skips dead tiles and its consumers read them: those parts of pi / q_J must hold zeros - the vectors
      // of the tile route lie behind offT and are cleared once per batch.  The window and per-patient kernels never let a
      // value of a dead tile into arithmetic (they are only ever loaded behind a per-state select): no clearing, which
      // matters when a cohort takes several batches per evaluation; MMHN_POISON=1 (tests) NaN-fills their part.
      if (!use_jacobi && nJ) {
        if (pi_owner != b.id) { zero(pi.p + b.offT, b.vecJ - b.offT); pi_owner = b.id; }
        if (grad && qJ_owner != b.id) { zero(qJ.p + b.offT, b.vecJ - b.offT); qJ_owner = b.id; }
        if (poison && b.offT > 0) {
          HIPCHECK(hipMemsetAsync(pi.p, 0xFF, (size_t)b.offT * sizeof(T), stream));
          if (grad) HIPCHECK(hipMemsetAsync(qJ.p, 0xFF, (size_t)b.offT * sizeof(T), stream));
        }
      }
      // 1-2 joint forward
      if (use_jacobi) {
        launch_diag(b.d_dJ.p, b.d_mapJ.p, tJ, nullptr, lidgJ.p, nullptr, KD_LIDG);
        solve(false, LJ, pi.p, lidgJ.p, nullptr, 2, nullptr);
      } else {
        psolve(false, b, pi.p, 2);
      }
      if (fused_small) small_classes(b, 0, grad);
      if (fused_small) small_classes(b, 1, grad);
      // 3-4 the staged single-tumour problems (the paired rows' after the joint forward solve)
      if (!own_forked) staged_fwd(b.stg[0]);
      staged_fwd(b.stg[1]);
      if (grad) {
        if (!own_forked) staged_adj(b.stg[0]);
        staged_adj(b.stg[1]);
        // (the 1024-thread class of the paired small-space launches ran on its side stream next to the staged kernels above:
        // the joint adjoint is the first consumer of what it wrote)
        small_join(1);
        if (nJ) {
          // 5 joint adjoint: right-hand side D_obs * scatter(q_S) formed on the fly inside the solve
          if (use_jacobi) zero(rhsJ.p, b.vecJ);
          for (int part = 0; part < 2 && !b.stg[1].paired.empty(); ++part) {
            if (b.stg[1].paired.size() >= 256)
              hipLaunchKernelGGL((k_scatter_marg<T, 1024>), dim3((unsigned)b.stg[1].paired.size()), dim3(1024), 0, stream, b.d_pats.p, b.d_dJ.p,
                                 b.d_dS.p, d_par.p, qS.p, rhsS.p, use_jacobi ? rhsJ.p : nullptr, dots.p, part, b.stg[1].d_paired.p);
            else
            hipLaunchKernelGGL((k_scatter_marg<T>), dim3((unsigned)b.stg[1].paired.size()), dim3(BLOCK), 0, stream, b.d_pats.p, b.d_dJ.p,
                               b.d_dS.p, d_par.p, qS.p, rhsS.p, use_jacobi ? rhsJ.p : nullptr, dots.p, part, b.stg[1].d_paired.p);
            HIPCHECK(hipGetLastError());
          }
          if (use_jacobi) solve(true, LJ, qJ.p, lidgJ.p, rhsJ.p, 0, nullptr);
          else psolve(true, b, qJ.p, 3);
          // 6 joint gradient (accumulators cleared at the start of the batch)
          if (!use_jacobi) {
            // algorithmic bytes: the seeded halves of pi and q_J read once
            const double mbytes = (double)b.vecJ * sizeof(T);
            timed(MMHN_K_PCLASS, mbytes, [&]() {
              if (b.wdirect) {                                                  // window-layout problems: both classes, two reads
                const int nW = (int)b.wd.size();
                hipLaunchKernelGGL((k_wclass<T>), dim3((unsigned)std::min(nW, n_cu)), dim3(WROWS), wclass_lds<T>(), stream, b.d_dJ.p, b.d_wd.p, nW, pi.p, qJ.p, Abuf.p);
              }
              // the problems in index order: work items (class passes of a problem cut into ranges + its eq block's flows)
              // on short launches, one workgroup per problem on long ones
              if (!b.pcl.empty())
                hipLaunchKernelGGL((k_pclass<T, true>), dim3((unsigned)b.pcl.size()), dim3(CMB), PC_LDS_ELEMS * sizeof(T), stream, b.d_dJ.p, b.d_wd.p, pi.p, qJ.p, Abuf.p, b.d_pcl.p);
              else if ((int)b.wd.size() < nJ || !b.wdirect)
                hipLaunchKernelGGL((k_pclass<T, false>), dim3(nJ), dim3(CMB), PC_LDS_ELEMS * sizeof(T), stream, b.d_dJ.p, b.d_wd.p, pi.p, qJ.p, Abuf.p);
            });
            if (!b.mapX.empty())
              hipLaunchKernelGGL((k_class_marg<T>), dim3((unsigned)b.mapX.size()), dim3(CMB), 2 * sizeof(T) << TB, stream,
                                 b.d_dJ.p, b.d_mapX.p, pi.p, qJ.p, Abuf.p);
          } else {
            hipLaunchKernelGGL((k_class_marg<T>), dim3(tJ), dim3(CMB), 2 * sizeof(T) << TB, stream, b.d_dJ.p,
                               b.d_mapJ.p, pi.p, qJ.p, Abuf.p);
          }
          HIPCHECK(hipGetLastError());
          // (its own launch: folded into the workgroups of k_pclass it cost more than the launch - k_pclass 20.5 -> 21.9 ms
          // on the bench cohort, the LUAD evaluation +25 us; short launches run it as work items of k_pclass)
          if (b.pcl.empty()) {
            hipLaunchKernelGGL((k_eq_flows<T>), dim3(nJ), dim3(BLOCK), 0, stream, b.d_dJ.p, b.d_wd.p, pi.p, qJ.p, Abuf.p);
            HIPCHECK(hipGetLastError());
          }
          launch_grad_rows(b.d_dJ.p, nJ, b.maxkcJ, Abuf.p, nullptr, nullptr, GJ.p, -1, b.d_grcJ, DJ.p, gjs);
        }
        // 7 assembly
      }
      if (fused_small) { small_join(0); small_join(1); }
      if (own_forked) HIPCHECK(hipStreamWaitEvent(stream, ev_join[2], 0));
      const AsmArgs<T> aa{b.d_pats.p, b.d_dJ.p, b.d_dS.p, d_par.p, GS.p, GJ.p, gjs, dots.p, DJ.p, (long long)nJ * N, bmS.p, lp.p, N,
                          grad ? 1 : 0};
      const int nelem = grad ? st : 1;
      hipLaunchKernelGGL((k_finalize<T>), dim3(npat), dim3(BLOCK), 0, stream, aa, out.p);
      HIPCHECK(hipGetLastError());
      {
        const int per = red_per(npat), nchunk = (npat + per - 1) / per;
        const dim3 cols((nelem + BLOCK - 1) / BLOCK);
        hipLaunchKernelGGL(k_reduce_rows, dim3(cols.x, nchunk), dim3(BLOCK), 0, stream, b.d_pats.p, npat, per, out.p, st, nelem, redbuf.p);
        HIPCHECK(hipGetLastError());
        if (pack_mode && &b == &batches.back()) {
          hipLaunchKernelGGL(k_reduce_parts_pack, dim3((st + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, redbuf.p, st, nelem, nchunk,
                             sums.p, pack_mode, N, pack_a, pack_b, pack_dst);
          packed_in_eval = true;
        } else {
          hipLaunchKernelGGL(k_reduce_parts, dim3(cols.x, 2), dim3(BLOCK), 0, stream, redbuf.p, st, nelem, nchunk, sums.p);
        }
        HIPCHECK(hipGetLastError());
      }
      if (host_out) {
        std::vector<double> tmp((size_t)npat * st);
        HIPCHECK(hipMemcpyAsync(tmp.data(), out.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        check_abort();
        for (int i = 0; i < npat; ++i)
          std::memcpy(host_out + (size_t)b.pats[i].row * st, tmp.data() + (size_t)i * st, st * sizeof(double));
      }
    }
    time_kernels = true;
    if (host_sums) {
      std::vector<double> hs(2 * st);
      HIPCHECK(hipMemcpyAsync(hs.data(), sums.p, hs.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIPCHECK(hipStreamSynchronize(stream));
      check_abort();
      std::memcpy(host_sums, hs.data(), hs.size() * sizeof(double));
      finish_eval(t0);
    }
  }
  void reset_counters() {                              // (what RCCL said about the communicator stays)
    const int r = cnt.comm_ranks, k = cnt.comm_rank;
    cnt = mmhn_counters{};
    cnt.comm_ranks = r; cnt.comm_rank = k;
  }
  void finish_eval(std::chrono::steady_clock::time_point t0) {
    collect_events();
    cnt.eval_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    cnt.evals += 1;
  }

  // sums layout of the C ABI (include/metmhn_amd.h): packed on the device, summed over the ranks of the
  // communicator (one RCCL all-reduce on this stream, regularized_optimization.py:256-266 needs nothing else),
  // then one download and one synchronisation per evaluation
  // begin: everything is issued (evaluation, packing, the all-reduce, the download into pinned memory), nothing is
  // waited for - the caller's host work (the reference computes its penalty terms on the host after the score,
  // regularized_optimization.py:296) runs next to the GPU; end: wait and copy out.
  bool sums_pending = false;
  std::chrono::steady_clock::time_point sums_t0, sums_issued;
  // set by cohort_sums_begin around evaluate(): the last batch's reduction also packs (k_reduce_parts_pack)
  int pack_mode = 0;
  double pack_a = 0, pack_b = 0;
  double* pack_dst = nullptr;
  bool packed_in_eval = false;
  // w_combined (optional): pack w * EM + NM on the device (k_pack_wsums) - 1 + N^2 + 2N doubles travel instead of
  // 4 + 2 N^2 + 3 N
  int sums_len = 0;                                           // doubles of the pending result
  double reduce_flag = 0.0, reduce_flag_sum = 0.0;            // mmhn_set_reduce_flag / mmhn_get_reduce_flag
  bool flag_pending = false;
  void cohort_sums_begin(const double* lt, const double* ldp, const double* ldm, bool grad, const double* w_combined = nullptr) {
    REQUIRE(!sums_pending, "mmhn_cohort_sums_begin: the previous evaluation has not been collected");
    sums_t0 = std::chrono::steady_clock::now();
    const int full = 4 + 2 * N * N + 3 * N;
    const int total = w_combined ? stride() : full;
    abi_sums.alloc(full);
    if (!h_abi) {
      HIPCHECK(hipHostMalloc(reinterpret_cast<void**>(&h_abi), (size_t)full * sizeof(double), hipHostMallocDefault));
      HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h_abi_dev), h_abi, 0));
    }
    // without a communicator the packing kernel writes the result straight into the pinned buffer; with one the
    // all-reduce works on device memory and the copy engine brings its result down
    double* packed = (zero_copy && !comm) ? h_abi_dev : abi_sums.p;
    pack_mode = w_combined ? 2 : 1;
    pack_a = w_combined ? *w_combined : (double)n_em; pack_b = w_combined ? reduce_flag : (double)n_pat; pack_dst = packed;
    packed_in_eval = false;
    struct Unset { int& m; ~Unset() { m = 0; } } unset{pack_mode};
    evaluate(lt, ldp, ldm, grad, nullptr, nullptr);
    if (!packed_in_eval) {                                 // (no batch: an empty cohort)
      if (w_combined) hipLaunchKernelGGL(k_pack_wsums, dim3(2), dim3(256), 0, stream, sums.p, N, *w_combined, packed, reduce_flag);
      else hipLaunchKernelGGL(k_pack_sums, dim3(2), dim3(256), 0, stream, sums.p, N, n_em, (double)n_pat, packed);
      HIPCHECK(hipGetLastError());
    }
    const int moved = total + (w_combined ? 1 : 0);            // (the reduce flag rides behind the pre-combined buffer)
    if (comm) RCCLCHECK(rccl().AllReduce(abi_sums.p, abi_sums.p, (size_t)moved, ncclFloat64, ncclSum, comm, stream));
    if (packed == abi_sums.p) HIPCHECK(hipMemcpyAsync(h_abi, abi_sums.p, moved * sizeof(double), hipMemcpyDeviceToHost, stream));
    flag_pending = w_combined != nullptr;
    reduce_flag = 0.0;
    sums_issued = std::chrono::steady_clock::now();
    sums_pending = true;
    sums_len = total;
  }
  void cohort_sums_end(double* o, int expect_len) {
    REQUIRE(sums_pending, "mmhn_cohort_sums_end without mmhn_cohort_sums_begin");
    REQUIRE(expect_len == sums_len, "mmhn_cohort_sums_end / mmhn_cohort_wsums_end does not match the _begin call");
    sums_pending = false;
    HIPCHECK(hipStreamSynchronize(stream));
    check_abort();
    std::memcpy(o, h_abi, (size_t)sums_len * sizeof(double));
    reduce_flag_sum = flag_pending ? h_abi[sums_len] : 0.0;
    static const bool trace_host = std::getenv("MMHN_TRACE_HOST") != nullptr;   // diagnostic: host time to issue vs total
    if (trace_host)
      std::fprintf(stderr, "[mmhn] evaluation issued after %.1f us, complete after %.1f us\n",
                   std::chrono::duration<double, std::micro>(sums_issued - sums_t0).count(),
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - sums_t0).count());
    finish_eval(sums_t0);
  }
  // (A hipGraph replay of the evaluation was measured on the LUAD-reduced cohort, ROCm 7.0.2: the host is free after
  // 75 us instead of 535 us, but the graph takes 660 us to execute against 550 us for the eager launches - dropped.)
  void cohort_sums(const double* lt, const double* ldp, const double* ldm, bool grad, double* o) {
    cohort_sums_begin(lt, ldp, ldm, grad);
    cohort_sums_end(o, 4 + 2 * N * N + 3 * N);
  }

  void comm_init(const ncclUniqueId& id, int rank, int nranks) {
    REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: rank / n_ranks out of range");
    comm_destroy();
    RCCLCHECK(rccl().CommInitRank(&comm, nranks, id, rank));
    comm_rank = rank; comm_size = nranks;
    // what RCCL itself says about the communicator (mmhn_get_counters: the bench line's proof that it spans the ranks)
    int cnt_ = 0, rk_ = -1;
    RCCLCHECK(rccl().CommCount(comm, &cnt_));
    RCCLCHECK(rccl().CommUserRank(comm, &rk_));
    if (cnt_ != nranks || rk_ != rank) {
      comm_destroy();
      throw Fail{"comm_init: RCCL reports " + std::to_string(cnt_) + " ranks / rank " + std::to_string(rk_) + ", asked for " +
                 std::to_string(nranks) + " / " + std::to_string(rank)};
    }
    cnt.comm_ranks = cnt_; cnt.comm_rank = rk_;
  }
  void comm_destroy() {
    if (comm) { (void)rccl().CommDestroy(comm); comm = nullptr; }
    comm_rank = 0; comm_size = 1;
    cnt.comm_ranks = 0; cnt.comm_rank = -1;
  }

  // ---------------------------------------------------------------- single-problem primitives (API / tests)
  struct Mini {
    Desc d;
    DevArr<Desc> dd;
    DevArr<int2> map, lmap;
    std::vector<int> lof;
    int ntiles = 0;
    DevArr<T> a, b, c, e, tab;
    PList plist(long long vec) const { return PList{dd.p, map.p, ntiles, d.k, vec, lmap.p, &lof, tab.p}; }
  };
  void mini_setup(Mini& m, const Desc& d) {
    m.d = d;
    m.d.off = 0; m.d.aoff = 0; m.d.toff = 0;
    REQUIRE(d.k <= MAXK, "too many active events");
    std::vector<int2> mp;
    add_tiles(mp, 0, d.k);
    m.ntiles = (int)mp.size();
    std::vector<int2> lm;
    build_levels(mp, nullptr, false, lm, m.lof);
    m.lmap.alloc(lm.size());
    HIPCHECK(hipMemcpyAsync(m.lmap.p, lm.data(), lm.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
    m.dd.alloc(1); m.map.alloc(mp.size());
    HIPCHECK(hipMemcpyAsync(m.dd.p, &m.d, sizeof(Desc), hipMemcpyHostToDevice, stream));
    HIPCHECK(hipMemcpyAsync(m.map.p, mp.data(), mp.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
    m.tab.alloc((size_t)std::max<long long>(table_size(m.d), 1));
    prep(m.dd.p, 1, m.tab.p);
    HIPCHECK(hipStreamSynchronize(stream));
  }
  void up(DevArr<T>& dst, const double* src, size_t count) {
    dst.alloc(count);
    std::vector<T> tmp(src, src + count);
    HIPCHECK(hipMemcpyAsync(dst.p, tmp.data(), count * sizeof(T), hipMemcpyHostToDevice, stream));
    HIPCHECK(hipStreamSynchronize(stream));
  }
  void down(double* dst, const T* src, size_t count) {
    std::vector<T> tmp(count);
    HIPCHECK(hipMemcpyAsync(tmp.data(), src, count * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    for (size_t i = 0; i < count; ++i) dst[i] = (double)tmp[i];
  }

  void api_kronvec(const Desc& d, const double* p, double* y, bool diag, bool tr) {
    Mini m; mini_setup(m, d);
    const size_t V = (size_t)1 << d.k;
    up(m.a, p, V);
    m.b.alloc(V);
    if (d.k > TB) {
      std::vector<int2> live;
      for (int tl = 0; tl < m.ntiles; ++tl) if (!dead_tile(m.d, (uint32_t)tl)) live.push_back(make_int2(0, tl));
      DevArr<int2> dlive;
      DevArr<T> hxt;
      dlive.alloc(live.size());
      hxt.alloc(live.size() * (size_t)d.k);
      HIPCHECK(hipMemcpyAsync(dlive.p, live.data(), live.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
      HIPCHECK(hipMemsetAsync(m.b.p, 0, V * sizeof(T), stream));      // structurally zero tiles are not launched
      hipLaunchKernelGGL((k_hx<T>), dim3((unsigned)live.size()), dim3(64), 0, stream, m.dd.p, dlive.p, m.tab.p, hxt.p, d.k);
      HIPCHECK(hipGetLastError());
      launch_kv(tr, m.dd.p, dlive.p, (int)live.size(), d.k, m.a.p, m.b.p, m.tab.p, hxt.p);
      HIPCHECK(hipStreamSynchronize(stream));
    } else {
      launch_sweep(tr, m.dd.p, m.map.p, m.ntiles, d.k, m.a.p, m.b.p, nullptr, nullptr, 0, nullptr, 0, m.tab.p);
    }
    if (diag) launch_diag(m.dd.p, m.map.p, m.ntiles, m.a.p, m.b.p, nullptr, KD_ADDQP);
    down(y, m.b.p, V);
  }
  // ---- batched product (kronvec.py:499-539 applied to `batch` vectors of one restricted space): ONE launch over every
  // tile of every vector.  Nothing is cleared beforehand: tiles where Q_off has no entries are zeroed by the kernel
  // itself (k_kv's kind 1), so y may be any buffer - this is the launch sequence mmhn_bench_kronvec times.
  struct KvBatch {
    Desc d;
    long long batch = 0, V = 0;
    int ntiles = 0, nlive = 0;                    // tiles of the batch; those of them where Q_off has entries
    bool use_kv = false;
    DevArr<Desc> dd;
    DevArr<int2> map, live;                       // every tile; the tiles with entries (what the plain product launches)
    DevArr<int> zmap;                             // per live tile: the structurally zero tile its workgroup clears, -1: none
    DevArr<T> tab, hxl;                           // hxl: tile-bit factors of the live tiles
  };
  void kv_setup(KvBatch& kb, const Desc& d0, long long batch) {
    REQUIRE(batch >= 1, "batch must be positive");
    REQUIRE(d0.k <= MAXK, "too many active events");
    kb.d = d0; kb.batch = batch; kb.V = 1ll << d0.k;
    std::vector<Desc> ds((size_t)batch, d0);
    std::vector<int2> mp;
    for (long long i = 0; i < batch; ++i) { ds[i].off = i * kb.V; ds[i].aoff = 0; ds[i].toff = 0; add_tiles(mp, (int)i, d0.k); }
    kb.ntiles = (int)mp.size();
    std::vector<int2> lv;
    std::vector<int> zm;
    for (const int2& m : mp) {
      if (dead_tile(ds[m.x], (uint32_t)m.y)) continue;
      lv.push_back(m);
      // a seeded tile clears its seed = 0 counterpart when Q_off has no entries there (dead tiles only exist with the
      // seeding bit above the tile bits, and the counterpart of a dead tile is always live)
      int z = -1;
      if (d0.mode == JOINT && d0.seedbit >= TB) {
        const uint32_t sb = 1u << (d0.seedbit - TB);
        if (((uint32_t)m.y & sb) && dead_tile(ds[m.x], (uint32_t)m.y & ~sb)) z = (int)((uint32_t)m.y & ~sb);
      }
      zm.push_back(z);
    }
    kb.nlive = (int)lv.size();
    {
      size_t cleared = 0;
      for (int z : zm) cleared += z >= 0;
      REQUIRE(cleared + lv.size() == mp.size(), "kronvec: a structurally zero tile has no live counterpart");
    }
    kb.live.alloc(lv.size()); kb.zmap.alloc(zm.size());
    HIPCHECK(hipMemcpyAsync(kb.live.p, lv.data(), lv.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
    HIPCHECK(hipMemcpyAsync(kb.zmap.p, zm.data(), zm.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    kb.dd.alloc(ds.size()); kb.map.alloc(mp.size());
    HIPCHECK(hipMemcpyAsync(kb.dd.p, ds.data(), ds.size() * sizeof(Desc), hipMemcpyHostToDevice, stream));
    HIPCHECK(hipMemcpyAsync(kb.map.p, mp.data(), mp.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
    kb.tab.alloc((size_t)std::max<long long>(table_size(d0), 1));
    prep(kb.dd.p, 1, kb.tab.p);                    // one table: every vector lives in the same space
    kb.use_kv = d0.k > TB;
    if (kb.use_kv) {
      kb.hxl.alloc(lv.size() * (size_t)d0.k);
      hipLaunchKernelGGL((k_hx<T>), dim3((unsigned)kb.nlive), dim3(64), 0, stream, kb.dd.p, kb.live.p, kb.tab.p, kb.hxl.p, d0.k);
      HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipStreamSynchronize(stream));
  }
  // the live tiles, each seeded one also filling its counterpart without entries of Q_off (zeros; lidg * rhs in the
  // fused Jacobi step): all of y is written by one launch
  void kv_launch(const KvBatch& kb, bool tr, const T* p, T* y, const T* lidg = nullptr, const T* rhs = nullptr) {
    if (kb.use_kv) launch_kv(tr, kb.dd.p, kb.live.p, kb.nlive, kb.d.k, p, y, kb.tab.p, kb.hxl.p, kb.zmap.p, lidg, rhs);
    else launch_sweep(tr, kb.dd.p, kb.map.p, kb.ntiles, kb.d.k, p, y, lidg, rhs, 0, nullptr, 0, kb.tab.p);
  }
  void api_kronvec_batched(const Desc& d, long long batch, const double* p, double* y, bool diag, bool tr) {
    KvBatch kb; kv_setup(kb, d, batch);
    const size_t tot = (size_t)(batch * kb.V);
    DevArr<T> a, b;
    up(a, p, tot);
    b.alloc(tot);
    HIPCHECK(hipMemsetAsync(b.p, 0xFF, tot * sizeof(T), stream));   // NaN pattern: every element must be written by the launch
    kv_launch(kb, tr, a.p, b.p);
    if (diag) launch_diag(kb.dd.p, kb.map.p, kb.ntiles, a.p, b.p, nullptr, KD_ADDQP);
    down(y, b.p, tot);
  }
  // one fused Jacobi step of R_i_inv_vec (likelihood.py:253-255) for `batch` vectors of one space:
  // y = lidg * (Q_off p + rhs) (transposed: Q_off^T), lidg = 1 / (D_p + D_m - diag Q) - the launch mmhn_bench_kronvec
  // times with jacobi != 0
  void api_jacobi_step_batched(const Desc& d, long long batch, const double* p, const double* rhs, double* y, bool tr) {
    KvBatch kb; kv_setup(kb, d, batch);
    const size_t tot = (size_t)(batch * kb.V);
    DevArr<T> a, b, c, r;
    up(a, p, tot);
    up(r, rhs, tot);
    b.alloc(tot); c.alloc(tot);
    launch_diag(kb.dd.p, kb.map.p, kb.ntiles, nullptr, c.p, nullptr, KD_LIDG);
    HIPCHECK(hipMemsetAsync(b.p, 0xFF, tot * sizeof(T), stream));
    kv_launch(kb, tr, a.p, b.p, c.p, r.p);
    down(y, b.p, tot);
  }
  void api_diag(const Desc& d, const double* p, double* outp, int what, int pbit = -1) {
    Mini m; mini_setup(m, d);
    const size_t V = (size_t)1 << d.k;
    if (p) up(m.a, p, V);
    m.b.alloc(V);
    launch_diag(m.dd.p, m.map.p, m.ntiles, m.a.p, m.b.p, nullptr, what, pbit);
    down(outp, m.b.p, V);
  }
  // vanilla.x_partial_D_y (vanilla.py:190-203): weighted bit marginals of x * y under the two parts of scal_d_pt
  void api_xDy_single(const Desc& d0, const double* x, const double* y, double* ddp, double* ddm) {
    Mini m; mini_setup(m, d0);
    const size_t V = (size_t)1 << d0.k;
    up(m.a, y, V);
    up(m.b, x, V);
    m.e.alloc(64);
    zero(m.e.p, 64);
    hipLaunchKernelGGL((k_bit_marg<T>), dim3(m.ntiles), dim3(BLOCK), 0, stream, m.dd.p, m.map.p, d_par.p, m.a.p,
                       m.b.p, m.e.p);
    HIPCHECK(hipGetLastError());
    double bm[64];
    down(bm, m.e.p, 64);
    for (int i = 0; i < N; ++i) {
      const int b = d0.bitP[i];
      ddp[i] = b >= 0 ? bm[b] : 0.0;
      ddm[i] = b >= 0 ? bm[32 + b] : 0.0;
    }
  }
  // achieved device-memory bandwidth of this GPU for a plain stream: kind 0 copy (b = a), 1 triad (a = b + s c);
  // 16 bytes per lane, `bytes` per array (>> Infinity Cache), HIP events around `iters` launches; GB/s of the
  // bytes the kernel is asked to move (copy 2 x, triad 3 x bytes)
  double bench_stream(size_t bytes, int iters, int kind) {
    REQUIRE(bytes >= (1u << 20) && iters >= 1 && (kind == 0 || kind == 1), "bench_stream: bad arguments");
    const size_t n16 = bytes / 16;
    DevArr<double2> a, b, c;
    a.alloc(n16); b.alloc(n16);
    if (kind == 1) c.alloc(n16);
    HIPCHECK(hipMemsetAsync(a.p, 0, n16 * 16, stream));
    HIPCHECK(hipMemsetAsync(b.p, 0, n16 * 16, stream));
    if (kind == 1) HIPCHECK(hipMemsetAsync(c.p, 0, n16 * 16, stream));
    static const int sblocks = std::getenv("MMHN_STREAM_BLOCKS") ? std::atoi(std::getenv("MMHN_STREAM_BLOCKS")) : 256 * 8;
    auto run = [&]() {
      hipLaunchKernelGGL(k_stream, dim3(sblocks), dim3(256), 0, stream, a.p, b.p, c.p, n16, kind);
    };
    run();
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0, stream));
    for (int i = 0; i < iters; ++i) run();
    HIPCHECK(hipEventRecord(e1, stream));
    HIPCHECK(hipEventSynchronize(e1));
    HIPCHECK(hipGetLastError());
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)(kind == 0 ? 2 : 3) * (double)(n16 * 16) * iters / ((double)ms * 1e6);
  }
  void api_resolvent(Desc d, const double* dvec, const double* x, double* y, bool tr) {
    if (dvec) d.obs = OBS_VEC;
    Mini m; mini_setup(m, d);
    const size_t V = (size_t)1 << d.k;
    up(m.a, x, V);
    if (dvec) up(m.e, dvec, V);
    m.b.alloc(V); m.c.alloc(V);
    launch_diag(m.dd.p, m.map.p, m.ntiles, nullptr, m.c.p, m.e.p, KD_LIDG);
    solve(tr, m.plist((long long)V), m.b.p, m.c.p, m.a.p, 0, nullptr);
    down(y, m.b.p, V);
  }
  void api_xQy_joint(const Desc& d0, const double* x, const double* y, double* G) {
    Mini m; mini_setup(m, d0);
    const size_t V = (size_t)1 << d0.k;
    up(m.a, y, V);   // p (right vector)
    up(m.b, x, V);   // q (left vector)
    m.c.alloc((size_t)a_size(m.d));
    m.e.alloc((size_t)3 * N * N);
    zero(m.e.p, 3ll * N * N);
    zero(m.c.p, a_size(m.d));
    hipLaunchKernelGGL((k_class_marg<T>), dim3(m.ntiles), dim3(CMB), 2 * sizeof(T) << TB, stream, m.dd.p, m.map.p,
                       m.a.p, m.b.p, m.c.p);
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL((k_eq_flows<T>), dim3(1), dim3(BLOCK), 0, stream, m.dd.p, static_cast<const WDesc*>(nullptr), m.a.p, m.b.p, m.c.p);
    HIPCHECK(hipGetLastError());
    for (int kd = 0; kd < 3; ++kd) {
      std::vector<int2> gc = grad_chunks(std::vector<Desc>{m.d}, kd);
      DevArr<int2> dgc; dgc.alloc(gc.size());
      HIPCHECK(hipMemcpy(dgc.p, gc.data(), gc.size() * sizeof(int2), hipMemcpyHostToDevice));
      launch_grad_rows(m.dd.p, 1, d0.k, m.c.p, nullptr, nullptr, m.e.p + kd * N * N, kd, dgc);
      HIPCHECK(hipStreamSynchronize(stream));
    }
    std::vector<double> g(3 * N * N);
    down(g.data(), m.e.p, g.size());
    for (int e = 0; e < N * N; ++e) G[e] = g[e] + g[N * N + e] + g[2 * N * N + e];
  }
  void api_xQy_single(const Desc& d0, const double* x, const double* y, double* G, double* ddiag) {
    Mini m; mini_setup(m, d0);
    const size_t V = (size_t)1 << d0.k;
    up(m.a, y, V);
    up(m.b, x, V);
    m.e.alloc((size_t)N * N);
    zero(m.e.p, (long long)N * N);
    {
      std::vector<int2> gc = grad_chunks(std::vector<Desc>{m.d}, GK_S);
      DevArr<int2> dgc; dgc.alloc(gc.size());
      HIPCHECK(hipMemcpy(dgc.p, gc.data(), gc.size() * sizeof(int2), hipMemcpyHostToDevice));
      launch_grad_rows(m.dd.p, 1, d0.k, nullptr, m.a.p, m.b.p, m.e.p, GK_S, dgc);
      HIPCHECK(hipStreamSynchronize(stream));
    }
    down(G, m.e.p, (size_t)N * N);
    if (ddiag)
      for (int j = 0; j < N; ++j) {
        double s = 0;
        for (int i = 0; i < N; ++i) if (i != j) s -= G[i * N + j];
        ddiag[j] = s;
      }
  }
  void api_xDy_joint(const Desc& d0, const double* x, const double* y, double* ddp, double* ddm) {
    Mini m; mini_setup(m, d0);
    const size_t V = (size_t)1 << d0.k;
    up(m.a, y, V);
    up(m.b, x, V);
    m.e.alloc(64);
    zero(m.e.p, 64);
    hipLaunchKernelGGL((k_bit_marg<T>), dim3(m.ntiles), dim3(BLOCK), 0, stream, m.dd.p, m.map.p, d_par.p, m.a.p,
                       m.b.p, m.e.p);
    HIPCHECK(hipGetLastError());
    double bm[64];
    down(bm, m.e.p, 64);
    for (int i = 0; i < N; ++i) {
      const int bp = i == n ? d0.seedbit : d0.bitP[i];
      const int bq = i == n ? d0.seedbit : d0.bitM[i];
      ddp[i] = bp >= 0 ? bm[bp] : 0.0;
      ddm[i] = bq >= 0 ? bm[32 + bq] : 0.0;
    }
  }
  // tiles[0] = tiles where Q_off has entries, tiles[1] = tiles per launch (both over the whole batch)
  double bench_kronvec(const Desc& d0, long long batch, int iters, bool tr, bool jacobi, long long* tiles) {
    REQUIRE(batch >= 1 && iters >= 1, "batch and iters must be positive");
    KvBatch kb; kv_setup(kb, d0, batch);
    const long long V = kb.V;
    if (tiles) { tiles[0] = kb.nlive; tiles[1] = kb.ntiles; }
    DevArr<T> a, b, c, r;
    a.alloc((size_t)(batch * V)); b.alloc((size_t)(batch * V));
    std::vector<T> host((size_t)V);
    for (long long i = 0; i < V; ++i) host[(size_t)i] = (T)(1.0 / (double)(1 + (i % 97)));
    for (long long i = 0; i < batch; ++i)
      HIPCHECK(hipMemcpy(a.p + i * V, host.data(), (size_t)V * sizeof(T), hipMemcpyHostToDevice));
    HIPCHECK(hipMemsetAsync(b.p, 0xFF, (size_t)(batch * V) * sizeof(T), stream));   // y starts as NaNs: the launch writes all of it
    if (jacobi) {
      c.alloc((size_t)(batch * V)); r.alloc((size_t)(batch * V));
      launch_diag(kb.dd.p, kb.map.p, kb.ntiles, nullptr, c.p, nullptr, KD_LIDG);
      HIPCHECK(hipMemcpyAsync(r.p, a.p, (size_t)(batch * V) * sizeof(T), hipMemcpyDeviceToDevice, stream));
    }
    // the timed launch is exactly the one mmhn_kronvec_batched issues (plain product), or the fused Jacobi step
    auto run = [&]() {
      if (jacobi) kv_launch(kb, tr, a.p, b.p, c.p, r.p);
      else kv_launch(kb, tr, a.p, b.p);
    };
    run(); run();
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0, stream));
    for (int i = 0; i < iters; ++i) run();
    HIPCHECK(hipEventRecord(e1, stream));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)ms / iters;
  }
};

// compatible indices (obs_states + jnp.where(size=)): integer host arithmetic, bit-exact
static void obs_indices(const Desc& d, bool pt_first, int64_t* idx, int64_t* count) {
  REQUIRE(d.seedbit >= 0, "obs_states needs an active seeding slot");
  const uint32_t fixed = (pt_first ? d.maskP : d.maskM) | (1u << d.seedbit);
  const uint32_t free_ = pt_first ? d.maskM : d.maskP;
  const int64_t cntv = (int64_t)1 << popc(free_);
  for (int64_t e = 0; e < cntv; ++e) {
    uint32_t v = (uint32_t)e, m = free_, o = 0;
    while (m) { const uint32_t low = m & (0u - m); if (v & 1u) o |= low; v >>= 1; m ^= low; }
    idx[e] = (int64_t)(o | fixed);
  }
  *count = cntv;
}

}  // namespace mmhn

// ======================================================================================
// C ABI
// ======================================================================================
using namespace mmhn;

struct mmhn_engine {
  int dtype;
  int n;
  EngineBase* impl;
};

#define API_BEGIN try {
#define API_END                                   \
  return 0;                                       \
  }                                               \
  catch (const Fail& f) { g_err = f.msg; return 1; } \
  catch (const std::exception& e) { g_err = e.what(); return 2; }

// engine's device current for the rest of the entry point
#define GUARD(h)                            \
  REQUIRE(h && h->impl, "null handle");     \
  DevGuard dev_guard_(h->impl->device)

#define DISPATCH(h, call)                                             \
  do {                                                                \
    REQUIRE(h && h->impl, "null handle");                             \
    if (h->dtype == MMHN_F64) static_cast<Engine<double>*>(h->impl)->call; \
    else static_cast<Engine<float>*>(h->impl)->call;                  \
  } while (0)

extern "C" {

const char* mmhn_last_error(void) { return g_err.c_str(); }

int mmhn_create(int device_id, int n_mut, int dtype, mmhn_handle* out) {
  API_BEGIN
  REQUIRE(out, "null out pointer");
  REQUIRE(dtype == MMHN_F64 || dtype == MMHN_F32, "dtype must be MMHN_F64 or MMHN_F32");
  int ndev = 0;
  HIPCHECK(hipGetDeviceCount(&ndev));
  REQUIRE(ndev > 0, "no HIP device visible: metmhn_amd needs a GPU (no CPU fallback)");
  REQUIRE(device_id >= 0 && device_id < ndev, "device_id out of range");
  auto* h = new mmhn_engine{dtype, n_mut, nullptr};
  try {
    if (dtype == MMHN_F64) h->impl = new Engine<double>(device_id, n_mut);
    else h->impl = new Engine<float>(device_id, n_mut);
  } catch (...) { delete h; throw; }
  h->impl->dtype = dtype;
  *out = h;
  API_END
}

void mmhn_destroy(mmhn_handle h) {
  if (!h) return;
  try {
    if (h->impl) {
      DevGuard guard(h->impl->device);     // device memory, stream and communicator are released on their GPU
      delete h->impl;
    }
  } catch (...) {
  }
  delete h;
}

int mmhn_set_workspace_limit(mmhn_handle h, size_t bytes) {
  API_BEGIN
  GUARD(h);
  REQUIRE(bytes >= (size_t)1 << 20, "workspace limit below 1 MiB");
  DISPATCH(h, ws_limit = bytes);
  API_END
}

int mmhn_set_cohort(mmhn_handle h, const int8_t* dat, int64_t n_pat, int n_cols) {
  API_BEGIN
  GUARD(h);
  REQUIRE(dat || n_pat == 0, "null dat");
  DISPATCH(h, set_cohort(dat, n_pat, n_cols));
  API_END
}

static void weights(double n_em, double n_pat, double perc_met, double* w, double* n_full) {
  const double n_nm = n_pat - n_em;                       // regularized_optimization.py:121-128
  *w = (n_em * n_nm != 0) ? perc_met * n_nm / ((1 - perc_met) * n_em) : 1.0;
  *n_full = *w * n_em + n_nm;
}

int mmhn_cohort_sums(mmhn_handle h, const double* lt, const double* ldp, const double* ldm, int with_grad,
                     double* sums) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && ldp && ldm && sums, "null pointer");
  DISPATCH(h, cohort_sums(lt, ldp, ldm, with_grad != 0, sums));
  API_END
}

int mmhn_cohort_sums_begin(mmhn_handle h, const double* lt, const double* ldp, const double* ldm, int with_grad) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && ldp && ldm, "null pointer");
  DISPATCH(h, cohort_sums_begin(lt, ldp, ldm, with_grad != 0));
  API_END
}

int mmhn_cohort_sums_end(mmhn_handle h, double* sums) {
  API_BEGIN
  GUARD(h);
  REQUIRE(sums, "null pointer");
  DISPATCH(h, cohort_sums_end(sums, 4 + 2 * (h->n + 1) * (h->n + 1) + 3 * (h->n + 1)));
  API_END
}

int mmhn_cohort_wsums_begin(mmhn_handle h, const double* lt, const double* ldp, const double* ldm, int with_grad, double w) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && ldp && ldm, "null pointer");
  REQUIRE(std::isfinite(w), "w must be finite");
  DISPATCH(h, cohort_sums_begin(lt, ldp, ldm, with_grad != 0, &w));
  API_END
}

int mmhn_set_reduce_flag(mmhn_handle h, double value) {
  API_BEGIN
  GUARD(h);
  REQUIRE(std::isfinite(value), "the flag must be finite");
  DISPATCH(h, reduce_flag = value);
  API_END
}
int mmhn_get_reduce_flag(mmhn_handle h, double* summed) {
  API_BEGIN
  GUARD(h);
  REQUIRE(summed, "null pointer");
  if (h->dtype == MMHN_F64) *summed = static_cast<Engine<double>*>(h->impl)->reduce_flag_sum;
  else *summed = static_cast<Engine<float>*>(h->impl)->reduce_flag_sum;
  API_END
}

int mmhn_cohort_wsums_end(mmhn_handle h, double* wsums) {
  API_BEGIN
  GUARD(h);
  REQUIRE(wsums, "null pointer");
  DISPATCH(h, cohort_sums_end(wsums, 1 + (h->n + 1) * (h->n + 1) + 2 * (h->n + 1)));
  API_END
}

int mmhn_score_and_grad(mmhn_handle h, const double* lt, const double* ldp, const double* ldm, double perc_met,
                        double* score, double* d_theta, double* d_dp, double* d_dm) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && ldp && ldm && score, "null pointer");
  const int N = h->n + 1;
  const bool grad = d_theta && d_dp && d_dm;
  std::vector<double> s(4 + 2 * N * N + 3 * N);
  DISPATCH(h, cohort_sums(lt, ldp, ldm, grad, s.data()));
  double w, nf;
  weights(s[2], s[3], perc_met, &w, &nf);
  *score = (w * s[0] + s[1]) / nf;
  if (grad) {
    const double* gem = s.data() + 4;
    const double* gnm = gem + N * N;
    const double* pem = gnm + N * N;
    const double* pnm = pem + N;
    const double* mem_ = pnm + N;
    for (int e = 0; e < N * N; ++e) d_theta[e] = (w * gem[e] + gnm[e]) / nf;
    for (int i = 0; i < N; ++i) { d_dp[i] = (w * pem[i] + pnm[i]) / nf; d_dm[i] = w * mem_[i] / nf; }
  }
  API_END
}

int mmhn_score(mmhn_handle h, const double* lt, const double* ldp, const double* ldm, double perc_met,
               double* score) {
  return mmhn_score_and_grad(h, lt, ldp, ldm, perc_met, score, nullptr, nullptr, nullptr);
}

int mmhn_patient_grads(mmhn_handle h, const double* lt, const double* ldp, const double* ldm, double* lp,
                       double* d_theta, double* d_dp, double* d_dm) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && ldp && ldm && lp, "null pointer");
  const int N = h->n + 1, st = 1 + N * N + 2 * N;
  long long np = 0;
  if (h->dtype == MMHN_F64) np = static_cast<Engine<double>*>(h->impl)->n_pat;
  else np = static_cast<Engine<float>*>(h->impl)->n_pat;
  std::vector<double> rows((size_t)np * st), s(2 * st);
  const bool grad = d_theta != nullptr;
  DISPATCH(h, evaluate(lt, ldp, ldm, grad, s.data(), rows.data()));
  for (long long i = 0; i < np; ++i) {
    const double* r = rows.data() + (size_t)i * st;
    lp[i] = r[0];
    if (grad) {
      std::memcpy(d_theta + (size_t)i * N * N, r + 1, N * N * sizeof(double));
      if (d_dp) std::memcpy(d_dp + (size_t)i * N, r + 1 + N * N, N * sizeof(double));
      if (d_dm) std::memcpy(d_dm + (size_t)i * N, r + 1 + N * N + N, N * sizeof(double));
    }
  }
  API_END
}

// ---- joint primitives
#define JOINT_DESC(state) make_joint(state, h->n)

int mmhn_kronvec(mmhn_handle h, const double* lt, const int8_t* state, const double* p, double* y, int diag,
                 int transpose) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && state && p && y, "null pointer");
  const Desc d = JOINT_DESC(state);
  DISPATCH(h, build_params(lt, nullptr, nullptr));
  DISPATCH(h, api_kronvec(d, p, y, diag != 0, transpose != 0));
  API_END
}
int mmhn_kronvec_batched(mmhn_handle h, const double* lt, const int8_t* state, int64_t batch, const double* p,
                         double* y, int diag, int transpose) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && state && p && y, "null pointer");
  REQUIRE(batch >= 1, "batch must be positive");
  const Desc d = JOINT_DESC(state);
  DISPATCH(h, build_params(lt, nullptr, nullptr));
  DISPATCH(h, api_kronvec_batched(d, batch, p, y, diag != 0, transpose != 0));
  API_END
}
int mmhn_jacobi_step_batched(mmhn_handle h, const double* lt, const double* ldp, const double* ldm, const int8_t* state,
                             int64_t batch, const double* p, const double* rhs, double* y, int transpose) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && ldp && ldm && state && p && rhs && y, "null pointer");
  REQUIRE(batch >= 1, "batch must be positive");
  const Desc d = JOINT_DESC(state);
  DISPATCH(h, build_params(lt, ldp, ldm));
  DISPATCH(h, api_jacobi_step_batched(d, batch, p, rhs, y, transpose != 0));
  API_END
}
int mmhn_kron_diag(mmhn_handle h, const double* lt, const int8_t* state, double* out) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && state && out, "null pointer");
  const Desc d = JOINT_DESC(state);
  DISPATCH(h, build_params(lt, nullptr, nullptr));
  DISPATCH(h, api_diag(d, nullptr, out, KD_DQ));
  API_END
}
int mmhn_diag_scal(mmhn_handle h, const double* log_d, const int8_t* state, const double* p, double* y, int which) {
  API_BEGIN
  GUARD(h);
  REQUIRE(log_d && state && p && y, "null pointer");
  REQUIRE(which == 0 || which == 1, "which must be 0 (d_p) or 1 (d_m)");
  const Desc d = JOINT_DESC(state);
  REQUIRE(d.seedbit >= 0, "diag_scal needs an active seeding slot");
  const int N = h->n + 1;
  std::vector<double> lt((size_t)N * N, 0.0);
  DISPATCH(h, build_params(lt.data(), which == 0 ? log_d : nullptr, which == 1 ? log_d : nullptr));
  DISPATCH(h, api_diag(d, p, y, which == 0 ? KD_DP : KD_DM));
  API_END
}
int mmhn_obs_states(mmhn_handle h, const int8_t* state, int pt_first, int64_t* idx, int64_t* count) {
  API_BEGIN
  GUARD(h);
  REQUIRE(h && state && idx && count, "null pointer");
  obs_indices(JOINT_DESC(state), pt_first != 0, idx, count);
  API_END
}
int mmhn_resolvent(mmhn_handle h, const double* lt, const double* ldp, const double* ldm, const int8_t* state,
                   const double* x, double* y, int transpose) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && ldp && ldm && state && x && y, "null pointer");
  const Desc d = JOINT_DESC(state);
  DISPATCH(h, build_params(lt, ldp, ldm));
  DISPATCH(h, api_resolvent(d, nullptr, x, y, transpose != 0));
  API_END
}
int mmhn_x_partial_Q_y(mmhn_handle h, const double* lt, const int8_t* state, const double* x, const double* y,
                       double* G) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && state && x && y && G, "null pointer");
  const Desc d = JOINT_DESC(state);
  DISPATCH(h, build_params(lt, nullptr, nullptr));
  DISPATCH(h, api_xQy_joint(d, x, y, G));
  API_END
}
int mmhn_x_partial_D_y(mmhn_handle h, const double* ldp, const double* ldm, const int8_t* state, const double* x,
                       const double* y, double* d_dp, double* d_dm) {
  API_BEGIN
  GUARD(h);
  REQUIRE(ldp && ldm && state && x && y && d_dp && d_dm, "null pointer");
  const Desc d = JOINT_DESC(state);
  const int N = h->n + 1;
  std::vector<double> lt((size_t)N * N, 0.0);
  DISPATCH(h, build_params(lt.data(), ldp, ldm));
  DISPATCH(h, api_xDy_joint(d, x, y, d_dp, d_dm));
  API_END
}

int mmhn_partial_diag_scal(mmhn_handle h, const double* log_d, const int8_t* state, const double* p, int i, int which,
                           double* y) {
  API_BEGIN
  GUARD(h);
  REQUIRE(log_d && state && p && y, "null pointer");
  REQUIRE(which == 0 || which == 1, "which must be 0 (d_p) or 1 (d_m)");
  const int n = h->n, N = n + 1;
  REQUIRE(i >= 0 && i <= n, "event index out of range");
  const Desc d = JOINT_DESC(state);
  REQUIRE(d.seedbit >= 0, "partial_diag_scal needs an active seeding slot");
  // kronvec.py:632-644, :704-710: zero when the tumour's slot of event i is inactive; i == n: the seeding bit
  const int bit = i == n ? d.seedbit : (which == 0 ? d.bitP[i] : d.bitM[i]);
  std::vector<double> lt((size_t)N * N, 0.0);
  DISPATCH(h, build_params(lt.data(), which == 0 ? log_d : nullptr, which == 1 ? log_d : nullptr));
  if (bit < 0) {
    std::memset(y, 0, sizeof(double) << d.k);
  } else {
    DISPATCH(h, api_diag(d, p, y, which == 0 ? KD_DP : KD_DM, bit));
  }
  API_END
}

// ---- single-tumour primitives
int mmhn_v_kronvec(mmhn_handle h, const double* lt, const int8_t* state, const double* p, double* y, int diag,
                   int transpose) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && state && p && y, "null pointer");
  const Desc d = make_single(state, h->n, PS_THETA, OBS_ONE);
  DISPATCH(h, build_params(lt, nullptr, nullptr));
  DISPATCH(h, api_kronvec(d, p, y, diag != 0, transpose != 0));
  API_END
}
int mmhn_v_resolvent(mmhn_handle h, const double* lt, const int8_t* state, const double* d_rates, const double* x,
                     double* y, int transpose) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && state && x && y, "null pointer");
  const Desc d = make_single(state, h->n, PS_THETA, OBS_ONE);
  DISPATCH(h, build_params(lt, nullptr, nullptr));
  DISPATCH(h, api_resolvent(d, d_rates, x, y, transpose != 0));
  API_END
}
int mmhn_v_x_partial_Q_y(mmhn_handle h, const double* lt, const int8_t* state, const double* x, const double* y,
                         double* G, double* d_diag) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && state && x && y && G, "null pointer");
  const Desc d = make_single(state, h->n, PS_THETA, OBS_ONE);
  DISPATCH(h, build_params(lt, nullptr, nullptr));
  DISPATCH(h, api_xQy_single(d, x, y, G, d_diag));
  API_END
}

int mmhn_v_kron_diag(mmhn_handle h, const double* lt, const int8_t* state, const double* diag, double* out) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && state && out, "null pointer");
  const Desc d = make_single(state, h->n, PS_THETA, OBS_ONE);
  DISPATCH(h, build_params(lt, nullptr, nullptr));
  DISPATCH(h, api_diag(d, diag, out, diag ? KD_QP : KD_DQ));
  API_END
}
int mmhn_v_scal_d_pt(mmhn_handle h, const double* ldp, const double* ldm, const int8_t* state, const double* vec,
                     double* out_p, double* out_m) {
  API_BEGIN
  GUARD(h);
  REQUIRE(ldp && ldm && state && vec && out_p && out_m, "null pointer");
  REQUIRE(state[h->n] == 1, "scal_d_pt needs the seeding event in the state (vanilla.py:142)");
  const Desc d = make_single(state, h->n, PS_THETA, OBS_MET);
  const int N = h->n + 1;
  std::vector<double> lt((size_t)N * N, 0.0);
  DISPATCH(h, build_params(lt.data(), ldp, ldm));
  DISPATCH(h, api_diag(d, vec, out_p, KD_SDP));
  DISPATCH(h, api_diag(d, vec, out_m, KD_DM));
  API_END
}
int mmhn_v_d_scal_d_pt(mmhn_handle h, const double* ldp, const double* ldm, const int8_t* state, const double* vec,
                       int i, double* out_p, double* out_m) {
  API_BEGIN
  GUARD(h);
  REQUIRE(ldp && ldm && state && vec && out_p && out_m, "null pointer");
  const int n = h->n, N = n + 1;
  REQUIRE(i >= 0 && i <= n, "event index out of range");
  REQUIRE(state[n] == 1, "d_scal_d_pt needs the seeding event in the state");
  const Desc d = make_single(state, n, PS_THETA, OBS_MET);
  std::vector<double> lt((size_t)N * N, 0.0);
  DISPATCH(h, build_params(lt.data(), ldp, ldm));
  // vanilla.py:182-187: inactive event -> zeros; i == n -> (0, d_m part); otherwise both parts restricted to "i happened"
  if (d.bitP[i] < 0 || i == n) std::memset(out_p, 0, sizeof(double) << d.k);
  else DISPATCH(h, api_diag(d, vec, out_p, KD_SDP, d.bitP[i]));
  if (d.bitP[i] < 0) std::memset(out_m, 0, sizeof(double) << d.k);
  else DISPATCH(h, api_diag(d, vec, out_m, KD_DM, i == n ? -1 : d.bitP[i]));
  API_END
}
int mmhn_v_x_partial_D_y(mmhn_handle h, const double* ldp, const double* ldm, const int8_t* state, const double* x,
                         const double* y, double* d_dp, double* d_dm) {
  API_BEGIN
  GUARD(h);
  REQUIRE(ldp && ldm && state && x && y && d_dp && d_dm, "null pointer");
  REQUIRE(state[h->n] == 1, "x_partial_D_y needs the seeding event in the state");
  const Desc d = make_single(state, h->n, PS_THETA, OBS_MET);
  const int N = h->n + 1;
  std::vector<double> lt((size_t)N * N, 0.0);
  DISPATCH(h, build_params(lt.data(), ldp, ldm));
  DISPATCH(h, api_xDy_single(d, x, y, d_dp, d_dm));
  API_END
}

// ---- patient shards on several GPUs: one RCCL communicator per engine
int mmhn_comm_unique_id(void* id128) {
  API_BEGIN
  REQUIRE(id128, "null pointer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  RCCLCHECK(rccl().GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  API_END
}
int mmhn_comm_init(mmhn_handle h, const void* id128, int rank, int n_ranks) {
  API_BEGIN
  GUARD(h);
  REQUIRE(id128, "null pointer");
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  DISPATCH(h, comm_init(id, rank, n_ranks));
  API_END
}
int mmhn_comm_destroy(mmhn_handle h) {
  API_BEGIN
  GUARD(h);
  DISPATCH(h, comm_destroy());
  API_END
}

// ---- Gillespie sampler
int mmhn_simulate(mmhn_handle h, const double* lt, const double* pt_d_ef, const double* mt_d_ef, int64_t n_sim,
                  uint64_t seed, int8_t* dat_out, int8_t* orders_out) {
  API_BEGIN
  GUARD(h);
  REQUIRE(h && lt && pt_d_ef && mt_d_ef && dat_out, "null pointer");
  REQUIRE(n_sim >= 0, "n_sim must be non-negative");
  const int N = h->n + 1;
  REQUIRE(N < SIM_MAXN, "too many events for the sampler (n_mut <= 30: event and diagnosis flags share one 32-bit set)");
  if (n_sim > 0) {
    const size_t W = (size_t)2 * h->n + 2, L = (size_t)2 * N + 2;
    DevArr<double> d_lt, d_dp, d_dm;
    DevArr<int8_t> d_dat, d_ord;
    d_lt.alloc((size_t)N * N); d_dp.alloc(N); d_dm.alloc(N); d_dat.alloc((size_t)n_sim * W);
    if (orders_out) d_ord.alloc((size_t)n_sim * L);
    HIPCHECK(hipMemcpy(d_lt.p, lt, sizeof(double) * N * N, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_dp.p, pt_d_ef, sizeof(double) * N, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_dm.p, mt_d_ef, sizeof(double) * N, hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)((n_sim + SIM_BLOCK - 1) / SIM_BLOCK);
    hipStream_t st = h->impl->stream;
    hipLaunchKernelGGL(k_gillespie, dim3(grid), dim3(SIM_BLOCK), 0, st, d_lt.p, d_dp.p, d_dm.p, N, (long long)n_sim, seed,
                       d_dat.p, orders_out ? d_ord.p : nullptr);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipMemcpy(dat_out, d_dat.p, (size_t)n_sim * W, hipMemcpyDeviceToHost));
    if (orders_out) HIPCHECK(hipMemcpy(orders_out, d_ord.p, (size_t)n_sim * L, hipMemcpyDeviceToHost));
  }
  API_END
}

// ---- measurement
int mmhn_bench_kronvec(mmhn_handle h, const double* lt, const int8_t* state, int64_t batch, int iters,
                       int transpose, int jacobi, double* ms_per_launch, int64_t* tiles) {
  API_BEGIN
  GUARD(h);
  REQUIRE(lt && state && ms_per_launch, "null pointer");
  const Desc d = JOINT_DESC(state);
  DISPATCH(h, build_params(lt, nullptr, nullptr));
  long long tl[2] = {0, 0};
  if (h->dtype == MMHN_F64)
    *ms_per_launch = static_cast<Engine<double>*>(h->impl)->bench_kronvec(d, batch, iters, transpose != 0, jacobi != 0, tl);
  else
    *ms_per_launch = static_cast<Engine<float>*>(h->impl)->bench_kronvec(d, batch, iters, transpose != 0, jacobi != 0, tl);
  if (tiles) { tiles[0] = tl[0]; tiles[1] = tl[1]; }
  API_END
}
int mmhn_bench_stream(mmhn_handle h, size_t bytes, int iters, int kind, double* gbps) {
  API_BEGIN
  GUARD(h);
  REQUIRE(gbps, "null pointer");
  if (h->dtype == MMHN_F64) *gbps = static_cast<Engine<double>*>(h->impl)->bench_stream(bytes, iters, kind);
  else *gbps = static_cast<Engine<float>*>(h->impl)->bench_stream(bytes, iters, kind);
  API_END
}
#ifdef MMHN_STAMPS
// diagnostic builds only: shader cycles wave 0 of every k_psolve workgroup spent per phase ([0..7] forward, [8..15] adjoint)
int mmhn_debug_stamps(mmhn_handle h, double* out16, int reset) {
  API_BEGIN
  GUARD(h);
  unsigned long long v[16];
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_stamps), sizeof(v)));
  for (int i = 0; i < 16; ++i) out16[i] = (double)v[i];
  if (reset) { std::memset(v, 0, sizeof(v)); HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), v, sizeof(v))); }
  API_END
}
#endif
int mmhn_get_counters(mmhn_handle h, mmhn_counters* out) {
  API_BEGIN
  GUARD(h);
  REQUIRE(h && h->impl && out, "null pointer");
  if (h->dtype == MMHN_F64) *out = static_cast<Engine<double>*>(h->impl)->cnt;
  else *out = static_cast<Engine<float>*>(h->impl)->cnt;
  API_END
}
int mmhn_debug_lane_moves(mmhn_handle h, int transposed, int* out) {
  API_BEGIN
  GUARD(h);
  REQUIRE(out, "null pointer");
  DevArr<int> d;
  d.alloc(6 * 64);
  if (transposed) hipLaunchKernelGGL((k_lane_moves<true>), dim3(1), dim3(64), 0, h->impl->stream, d.p);
  else hipLaunchKernelGGL((k_lane_moves<false>), dim3(1), dim3(64), 0, h->impl->stream, d.p);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(out, d.p, 6 * 64 * sizeof(int), hipMemcpyDeviceToHost, h->impl->stream));
  HIPCHECK(hipStreamSynchronize(h->impl->stream));
  API_END
}

int mmhn_abi_version(void) { return MMHN_ABI_VERSION; }

int mmhn_reset_counters(mmhn_handle h) {
  API_BEGIN
  GUARD(h);
  DISPATCH(h, reset_counters());
  API_END
}

}  // extern "C"
