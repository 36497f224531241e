// rtd_api.hip -- host side of the C ABI declared in include/rtd.h (plan life cycle, uploads, launches).
#include <algorithm>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <type_traits>
#include <vector>

#include "../../include/rtd.h"
#include "rtd_device.h"
#include "rtd_dd.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// (a failed runtime call also leaves its code in HIP's per-thread "last error": it is taken out here, or the launch check of
//  the next, unrelated, call on this thread -- hipGetLastError after its kernels -- would report it)
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (void)hipGetLastError();                                                                 \
      return fail(RTD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    }                                                                                          \
  } while (0)

int pad_pow2(int n) {
  int p = 4;
  while (p < n) p <<= 1;
  return p;
}

// Device-buffer and stream pool (SURVEY section 8(b) "threading": the library owns only device scratch, pooled and
// guarded by a mutex).  Two reasons, two tiers:
//  * one-column pydisort() calls create and destroy a plan per call; recycling small arenas (blocks <= 64 MB, 512 MB in all, a
//    block serves requests down to a quarter of its size) and streams removes the hipMalloc / hipFree / hipStreamCreate cost
//    (milliseconds) from that path;
//  * batch calls create and destroy plans of gigabytes.  hipFree of such a block returns in ~2 ms, but the runtime reclaims the
//    memory lazily, and every so often a later hipMalloc pays for it: 0.4 ... 4.8 s, growing with the bytes freed since
//    (profiles/r05_alloc_outliers.txt: 9 of 40 creations of a 14 GB plan, the same with bare hipMalloc / hipFree).  Blocks above
//    64 MB CAN therefore be kept too, oldest out first, up to a limit PER DEVICE -- but only when the caller asks for it
//    (rtd_pool_set_limit, or RTD_POOL_BYTES in the environment): the default is 0, a library that is imported under someone
//    else's process does not sit on gigabytes of a GPU it shares with that process's allocator (round-5 verdict).  Kept blocks
//    serve requests down to 7/8 of their size -- a serving loop repeats its shapes.  rtd_pool_trim gives everything back; a
//    hipMalloc that fails trims the pool and tries again; evicted blocks are freed OUTSIDE the pool's mutex (hipFree
//    synchronises the device).
struct DevPool {
  struct Block { void* p; size_t bytes; int dev; };
  std::mutex m;
  std::vector<Block> free_blocks, big_blocks;  // (big_blocks in order of arrival)
  std::vector<std::pair<hipStream_t, int>> free_streams;
  std::vector<Block> host_blocks;  // pinned staging slabs of rtd_plan_run_fetch (hipHostMalloc + hipHostFree: ~3 ms per plan)
  size_t cached = 0, big_cached = 0, host_cached = 0;
  int64_t big_cap = -1;  // bytes PER DEVICE; -1: not read from the environment yet
  static constexpr size_t MAX_BLOCK = 64u << 20, MAX_CACHED = 512u << 20;

  int64_t limit_locked() {  // (m held)
    if (big_cap < 0) {
      const char* env = getenv("RTD_POOL_BYTES");
      big_cap = env ? std::max<long long>(0, atoll(env)) : 0;
    }
    return big_cap;
  }
  size_t big_cached_on(int dev) const {
    size_t n = 0;
    for (const Block& b : big_blocks)
      if (b.dev == dev) n += b.bytes;
    return n;
  }
  // blocks of `dev` out, oldest first, until `room` more bytes fit under the limit; the caller frees them outside the lock
  void evict_locked(int dev, int64_t room, int64_t cap, std::vector<Block>* out) {
    int64_t held = (int64_t)big_cached_on(dev);
    for (size_t i = 0; i < big_blocks.size() && held + room > cap;) {
      if (big_blocks[i].dev != dev) {
        ++i;
        continue;
      }
      held -= (int64_t)big_blocks[i].bytes;
      big_cached -= big_blocks[i].bytes;
      out->push_back(big_blocks[i]);
      big_blocks.erase(big_blocks.begin() + i);
    }
  }
  static void release(const std::vector<Block>& blocks) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (const Block& b : blocks) {
      if (b.dev != cur) (void)hipSetDevice(b.dev);
      (void)hipFree(b.p);
    }
    if (!blocks.empty() && cur >= 0 && blocks.back().dev != cur) (void)hipSetDevice(cur);
  }
  // bytes >= 0: the new per-device limit; < 0: an eighth of the memory of device `dev`.  Returns the previous limit.
  int64_t set_limit(int64_t bytes, int dev) {
    std::vector<Block> out;
    int64_t prev;
    {
      std::lock_guard<std::mutex> g(m);
      prev = limit_locked();
      if (bytes < 0) {
        size_t fr = 0, total = 0;
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (dev >= 0 && dev != cur) (void)hipSetDevice(dev);
        bytes = hipMemGetInfo(&fr, &total) == hipSuccess ? (int64_t)(total / 8) : 0;
        if (dev >= 0 && dev != cur && cur >= 0) (void)hipSetDevice(cur);
      }
      big_cap = bytes;
      std::vector<int> devs;
      for (const Block& b : big_blocks)
        if (std::find(devs.begin(), devs.end(), b.dev) == devs.end()) devs.push_back(b.dev);
      for (int d : devs) evict_locked(d, 0, big_cap, &out);
    }
    release(out);
    return prev;
  }

  void* get(size_t bytes, int dev, size_t* got) {
    std::lock_guard<std::mutex> g(m);
    const bool big = bytes > MAX_BLOCK;
    std::vector<Block>& list = big ? big_blocks : free_blocks;
    const size_t most = big ? bytes + bytes / 7 : 4 * bytes + 4096;
    int best = -1;
    for (int i = 0; i < (int)list.size(); ++i) {
      const Block& b = list[i];
      if (b.dev == dev && b.bytes >= bytes && b.bytes <= most && (best < 0 || b.bytes < list[best].bytes)) best = i;
    }
    if (best < 0) return nullptr;
    Block b = list[best];
    list.erase(list.begin() + best);
    (big ? big_cached : cached) -= b.bytes;
    *got = b.bytes;
    return b.p;
  }
  bool put(void* p, size_t bytes, int dev) {
    std::vector<Block> out;
    {
      std::lock_guard<std::mutex> g(m);
      if (bytes <= MAX_BLOCK) {
        if (cached + bytes > MAX_CACHED) return false;
        free_blocks.push_back({p, bytes, dev});
        cached += bytes;
        return true;
      }
      const int64_t cap = limit_locked();
      if ((int64_t)bytes > cap) return false;
      evict_locked(dev, (int64_t)bytes, cap, &out);  // this device's oldest blocks make room; other devices keep theirs
      big_blocks.push_back({p, bytes, dev});
      big_cached += bytes;
    }
    release(out);
    return true;
  }
  void* get_host(size_t bytes, size_t* got) {
    std::lock_guard<std::mutex> g(m);
    int best = -1;
    for (int i = 0; i < (int)host_blocks.size(); ++i) {
      const Block& b = host_blocks[i];
      if (b.bytes >= bytes && b.bytes <= 2 * bytes + 4096 && (best < 0 || b.bytes < host_blocks[best].bytes)) best = i;
    }
    if (best < 0) return nullptr;
    Block b = host_blocks[best];
    host_blocks.erase(host_blocks.begin() + best);
    host_cached -= b.bytes;
    *got = b.bytes;
    return b.p;
  }
  bool put_host(void* p, size_t bytes) {
    std::lock_guard<std::mutex> g(m);
    if (host_cached + bytes > (256u << 20)) return false;
    host_blocks.push_back({p, bytes, -1});
    host_cached += bytes;
    return true;
  }
  // gives the cached blocks of `dev` (-1: every device) back to the runtime; returns the device bytes released
  size_t trim(int dev) {
    std::vector<Block> out, host_out;
    size_t released = 0;
    {
      std::lock_guard<std::mutex> g(m);
      if (dev < 0) {
        host_out.swap(host_blocks);
        host_cached = 0;
      }
      for (std::vector<Block>* list : {&free_blocks, &big_blocks}) {
        std::vector<Block> keep;
        for (const Block& b : *list) {
          if (dev >= 0 && b.dev != dev) {
            keep.push_back(b);
            continue;
          }
          out.push_back(b);
          released += b.bytes;
          (list == &free_blocks ? cached : big_cached) -= b.bytes;
        }
        list->swap(keep);
      }
    }
    for (const Block& b : host_out) (void)hipHostFree(b.p);
    release(out);
    return released;
  }
  hipStream_t get_stream(int dev) {
    std::lock_guard<std::mutex> g(m);
    for (int i = 0; i < (int)free_streams.size(); ++i)
      if (free_streams[i].second == dev) {
        hipStream_t s = free_streams[i].first;
        free_streams.erase(free_streams.begin() + i);
        return s;
      }
    return nullptr;
  }
  void put_stream(hipStream_t s, int dev) {
    std::lock_guard<std::mutex> g(m);
    if (free_streams.size() < 16) free_streams.emplace_back(s, dev);
    else (void)hipStreamDestroy(s);
  }
};
DevPool& pool() {
  static DevPool* p = new DevPool();  // never destroyed: outlives the HIP runtime teardown order
  return *p;
}

// pooled device allocation: returns the block and its true size
hipError_t pooled_malloc(void** q, size_t bytes, int dev, size_t* got) {
  *q = pool().get(bytes, dev, got);
  if (*q) return hipSuccess;
  *got = bytes;
  hipError_t e = hipMalloc(q, bytes);
  if (e == hipErrorOutOfMemory && pool().trim(dev) > 0) {  // what the pool holds is this library's to give up first
    (void)hipGetLastError();
    e = hipMalloc(q, bytes);
  }
  return e;
}
void pooled_free(void* q, size_t bytes, int dev) {
  if (!pool().put(q, bytes, dev)) (void)hipFree(q);
}

}  // namespace

struct rtd_plan {
  rtd_dims dims{};
  int device = 0;
  int NP = 0;
  hipStream_t stream = nullptr;
  // d.C = all columns.  The per-column INPUT pointers of d cover them all; the intermediates of the solve (Y0, att, Ym, Am,
  // kk, Ek, Bv, dq, zneg, coef, Fws) cover one window of Cw columns: launch_windows runs window after window.
  RtdDev d{};
  int Cw = 0, nwin = 1;
  // Window pipeline (plans of more than one window): the hand-off buffers of the eigen stage exist twice, and the tables +
  // eigen kernel of window w + 1 run on eig_stream while the boundary-condition + evaluation kernels of window w run on
  // `stream`.  The eigen kernel is bound by vector-instruction issue, the boundary-condition kernel by the latency of its
  // dependent chains: wavefronts of both kinds resident on a SIMD fill each other's bubbles (profiles/archive/r03_window_pipeline.json).
  struct HandOff { double *Y0, *att, *Ym, *Am, *kk, *Bv, *dq, *zneg, *Ek, *vb; } slot1{};
  // Legendre tables at -mu0 and beam attenuations of ALL columns, kept from run to run (they depend on the inputs and the
  // mode shard only): one launch after the inputs change instead of one per window and run (19 us of 1.17 ms per 256-column
  // cfg4 window).  Plans of one window keep them in d.Y0 / d.att; larger plans in these arrays when C M P doubles fit 2 GiB.
  double *Y0_all = nullptr, *att_all = nullptr;
  bool tables_cached = false, tables_valid = false;
  bool quad_tables_valid = false;  // the column-independent table Y[m][l][i] matches the quadrature and the mode shard (a plan that is
  //                                 reused for another column of the same shape keeps it: one launch less per one-column solve)
  bool pipelined = false;
  // Retained plan (rtd_plan_create_retained): the eigen stage's hand-off arrays and the coefficients cover ALL columns instead
  // of one window (two slots), so that the evaluators can be called again after a solve without solving again -- what the
  // reference's closures do with GC_collect, K_collect, B_collect (_assemble_intensity_and_fluxes.py:170-262).  Only the
  // boundary-condition workspace Fws (5 of a cfg4 column's 8.4 MB) stays windowed.
  bool retained = false;
  // Lean retained plan (round 6): what cannot be recomputed cheaply stays for ALL columns -- the boundary-condition coefficients,
  // k, E, B and the thermal vectors (0.5 of a cfg4 column's 3.1 MB) -- while Y, A stay windowed; an evaluation re-runs the EIGEN
  // stage for the wavefront chunks its points touch (same wavefront composition as in the solve: same bits), window by window,
  // and never the boundary-condition solve.  10^5 cfg4 columns: 50 GB instead of 310 GB.
  bool lean = false;
  double* sel_buf = nullptr;  // chunk lists of one window [Cw][nsel] (ints), grown on demand
  int64_t cap_sel = 0;
  bool fork_needed = true;               // inputs were (re)uploaded on `stream` since the last solve: the eigen stream must wait for them
  bool bc_recorded[2] = {false, false};  // ev_bc[slot] has been recorded by some earlier window (possibly of an earlier run)
  // Pipelined plans keep the device status words (status, col_status, sweeps) twice and alternate between them from solve to
  // solve: back-to-back runs overlap (the eigen stream starts run r + 1 while the boundary-condition and evaluation kernels of
  // run r still raise bits on the plan's stream), so run r + 1 clears and fills the other set; the set it clears was last
  // touched by run r - 1, which the eigen stream's first wait (ev_bc of run r) has behind it.
  int *status2[2] = {nullptr, nullptr}, *col_status2[2] = {nullptr, nullptr}, *sweeps2[2] = {nullptr, nullptr};
  int stat_idx = 0;
  hipStream_t eig_stream = nullptr;
  hipEvent_t ev_eig[2] = {nullptr, nullptr}, ev_bc[2] = {nullptr, nullptr}, ev_fork = nullptr;
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;
  int64_t bytes = 0;
  bool have_quad = false, have_cols = false, solved = false;
  int numeric_status = 0;  // RTD_ST_* bits of the last solve other than the tau range: reported until the next solve
  std::vector<double> h_tau;  // host copy of tau_arr [C][L]: recognises evaluation points that are the layer interfaces
  // Host images of the small uploads (quadrature, per-column inputs, evaluation points) live in the plan until the copy that reads
  // them is known to have completed: no hipStreamSynchronize per upload (each costs a one-column pydisort() call 10-20 us; the
  // call's only wait is now the one that brings its results back).  `*_pending`: an asynchronous copy from the image may still be
  // in flight -- the next writer of the image waits for the stream first; every call that drains the stream clears them.
  std::vector<char> quad_img, cols_img;
  std::vector<double> ev_img;
  bool quad_pending = false, cols_pending = false, ev_pending = false;
  // evaluation buffers (grown on demand)
  bool ev_iface = false;  // the stored evaluation points are [0, tau_arr] of every column (fused evaluation possible)
  double* um_buf = nullptr;  // [Cw][M][L+1][2 NP]: Fourier modes at the interfaces, written by the boundary-condition kernel
  int64_t cap_um = 0;
  int ev_ntau = 0, ev_nphi = 0;
  double *ev_tau = nullptr, *ev_phi = nullptr, *ev_u = nullptr, *ev_u0 = nullptr, *ev_fl = nullptr;
  int64_t cap_tau = 0, cap_phi = 0, cap_u = 0, cap_u0 = 0, cap_fl = 0;
  // export buffers
  double* ex_buf = nullptr;
  // Nakajima-Tanaka corrections (optional)
  RtdNt nt{};
  double *nt_w = nullptr, *nt_f = nullptr, *nt_ic = nullptr, *nt_ip = nullptr, *nt_R = nullptr;
  int64_t cap_nt_w = 0, cap_nt_f = 0, cap_nt_ic = 0, cap_nt_ip = 0, cap_nt_R = 0;
  bool have_nt = false, nt_tables_ready = false;
  // RCCL communicator (one rank per GPU)
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_size = 0;
  double* gathered = nullptr;  // fluxes only [nranks][3][C][ntau] (rtd_comm_allgather_fluxes)
  int64_t cap_gathered = 0;
  // gathered u [nranks][C][Q][ntau][nphi] and fluxes [nranks][3][C][ntau] (rtd_comm_allgather_results): the gather runs
  // on comm_stream behind ev_results, and the next run's first evaluation kernel waits for ev_gathered
  double *gathered_u = nullptr, *gathered_fl = nullptr;
  int64_t cap_gathered_u = 0, cap_gathered_fl = 0;
  hipStream_t comm_stream = nullptr, copy_stream = nullptr;
  hipEvent_t ev_results = nullptr, ev_gathered = nullptr;
  bool gather_inflight = false;
  bool gathered_here = false;  // the last results collective left the gathered arrays of ALL ranks on this rank
  // host-to-host pipeline (rtd_plan_run_fetch): two pinned staging slabs, one per window in flight
  char* stage[2] = {nullptr, nullptr};
  size_t stage_bytes = 0, stage_cap[2] = {0, 0};  // bytes in use of each slab; true size of each (a pooled slab may be larger)
  hipEvent_t ev_win[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
  // timing
  bool timing = false;
  static constexpr int NT = 7;  // timed kernels: tables, asm, jacobi, post, iface, sweep, eval
  hipEvent_t evt[NT + 1] = {};
  double ms[NT] = {};
  int64_t nlaunch[NT] = {};
  bool pending[NT] = {};

  template <typename T>
  int alloc(T** p, int64_t n) {
    void* q = nullptr;
    if (n <= 0) n = 1;
    size_t got = 0;
    hipError_t e = pooled_malloc(&q, (size_t)n * sizeof(T), device, &got);
    if (e != hipSuccess)
      return fail(RTD_ERR_HIP, std::string("hipMalloc of ") + std::to_string((size_t)n * sizeof(T)) + " bytes: " + hipGetErrorString(e));
    allocs.push_back(q);
    alloc_bytes.push_back(got);
    bytes += (int64_t)got;
    *p = (T*)q;
    return 0;
  }
};

namespace {

// (Re)allocates *buf for `need` doubles.  The old block goes back to the process-wide pool, where another plan on
// another stream may pick it up at once: this plan's streams are drained first (a plain hipFree would have
// synchronised the device).
int grow(rtd_plan* p, double** buf, int64_t* cap, int64_t need) {
  if (need <= *cap) return 0;
  if (*buf) {
    (void)hipStreamSynchronize(p->stream);
    if (p->comm_stream) (void)hipStreamSynchronize(p->comm_stream);
    for (size_t i = 0; i < p->allocs.size(); ++i)
      if (p->allocs[i] == *buf) {
        pooled_free(*buf, p->alloc_bytes[i], p->device);
        p->bytes -= (int64_t)p->alloc_bytes[i];
        p->allocs[i] = nullptr;
      }
  }
  *buf = nullptr;
  *cap = 0;
  int rc = p->alloc(buf, need);
  if (rc) return rc;
  *cap = need;
  return 0;
}

// collect timing of stage `k` if an event pair is pending
void harvest(rtd_plan* p) {
  for (int k = 0; k < rtd_plan::NT; ++k) {
    if (!p->pending[k]) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, p->evt[k], p->evt[k + 1]) == hipSuccess) {
      p->ms[k] += t;
      p->nlaunch[k] += 1;
    }
    p->pending[k] = false;
  }
}

// the plan's device view restricted to the columns [c0, c0 + cnt): input pointers advanced, intermediates shared
RtdDev window_dev(const rtd_plan* p, int64_t c0, int cnt, int slot = 0) {
  RtdDev w = p->d;
  if (slot == 1) {
    const rtd_plan::HandOff& h = p->slot1;
    w.Ym = h.Ym; w.Am = h.Am;
    if (!p->lean) { w.Y0 = h.Y0; w.att = h.att; w.kk = h.kk; w.Bv = h.Bv; w.dq = h.dq; w.zneg = h.zneg; w.Ek = h.Ek; w.vb = h.vb; }
  }
  if (p->tables_cached) {  // the window's part of the all-columns tables
    w.Y0 = p->Y0_all + c0 * (int64_t)p->d.M * p->d.P;
    w.att = p->att_all + c0 * ((int64_t)p->d.L + 1);
  }
  const int64_t L = w.L, M = w.M, P = w.P, NP = w.NP, Ns = w.Ns, NB = w.NBDRF;
  if (p->retained || p->lean) {  // every column has its own place in the hand-off arrays and the coefficients
    if (p->retained) { w.Ym += c0 * M * L * NP * NP; w.Am += c0 * M * L * NP * NP; }  // (lean: Y, A stay windowed)
    w.kk += c0 * M * L * NP; w.Ek += c0 * M * L * NP;
    w.Bv += c0 * M * L * 2 * NP; w.coef += c0 * M * L * 2 * NP; w.dq += c0 * L * Ns * 2 * NP; w.zneg += c0 * L * NP;
    if (Ns > 0) w.vb += c0 * L * 4 * NP;
  }
  w.C = cnt;
  w.omega += c0 * L; w.tau += c0 * L; w.taus0 += c0 * (L + 1); w.scale += c0 * L; w.wleg += c0 * L * P;
  w.mu0 += c0; w.I0 += c0; w.phi0 += c0; w.rescale += c0;
  w.bpos += c0 * M * NP; w.bneg += c0 * M * NP; w.spoly += c0 * L * Ns;
  w.bdrfq += c0 * NB * NP * NP; w.bdrfq0 += c0 * NB * NP; w.lperm += c0 * L;
  w.col_status += c0;
  return w;
}
RtdEval window_eval(const rtd_plan* p, const RtdEval& e, int64_t c0) {
  RtdEval w = e;
  const int64_t Qr = 2 * p->d.N, nt = e.ntau, np = e.nphi;
  w.tau += c0 * nt;
  if (w.u) w.u += c0 * Qr * nt * np;
  if (w.u0) w.u0 += c0 * Qr * nt;
  if (w.ulast) w.ulast += c0 * Qr * nt;
  if (w.fup) w.fup += c0 * nt;
  if (w.fdn) w.fdn += c0 * nt;
  if (w.fdir) w.fdir += c0 * nt;
  return w;
}
RtdNt window_nt(const rtd_plan* p, int64_t c0) {
  RtdNt w = p->nt;
  const int64_t L = p->d.L;
  w.wfull += c0 * L * w.nleg_all; w.f += c0 * L; w.ims_coef += c0 * w.nleg_all; w.ims_par += c0 * 2;
  w.R += c0 * 4 * p->d.NP * L;
  return w;
}

// Lean retained plans: which wavefront chunks of the eigen stage the evaluation points of a column touch.  One thread per
// column: the layer of every point exactly as the evaluation kernel finds it (rtd_eval.hip: argmax(tau <= tau_arr),
// _assemble_intensity_and_fluxes.py:185), its slot in the column's layer order lperm, slot / gpw = the chunk; the list
// sel[c][0 .. nsel) holds each touched chunk once, -1 beyond.
__global__ void rtd_chunk_select_kernel(RtdDev d, const double* tau, int ntau, int gpw, int nsel, int* sel) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.C) return;
  int* out = sel + (long)c * nsel;
  const double* tau_arr = d.tau + (long)c * d.L;
  const int* perm = d.lperm + (long)c * d.L;
  int n = 0;
  for (int t = 0; t < ntau; ++t) {
    const double x = tau[(long)c * ntau + t];
    int l = 0;
    while (l < d.L - 1 && !(x <= tau_arr[l])) ++l;
    int slot = 0;
    while (slot < d.L - 1 && perm[slot] != l) ++slot;
    const int chunk = slot / gpw;
    bool have = false;
    for (int k = 0; k < n; ++k) have = have || out[k] == chunk;
    if (!have && n < nsel) out[n++] = chunk;
  }
  for (int k = n; k < nsel; ++k) out[k] = -1;
}

// Solve (and optionally evaluate) every window.  after_window(w, c0, cnt) is called once window w's kernels are
// queued (rtd_plan_run_fetch hangs its device-to-host copies there).
template <typename F>
int launch_windows(rtd_plan* p, bool with_solve, const RtdEval* ev, bool with_nt, F&& after_window, bool allow_fused = true) {
  (void)hipGetLastError();  // (a stale code of this thread -- another library's, an earlier failed call's -- is not this launch's)
  hipStream_t s = p->stream;
  const bool tm = p->timing;
  auto mark = [&](int k) {
    if (tm) (void)hipEventRecord(p->evt[k], s);
  };
  // two-stream window pipeline: not while the stages are being timed one by one (the HIP-event pass wants them alone)
  const bool pipe = p->pipelined && with_solve && !tm && p->nwin > 1;
  hipStream_t se = pipe ? p->eig_stream : s;
  bool nt_tables_done = false;
  if (pipe && p->fork_needed) {  // the eigen stream starts behind the uploads queued on the plan's stream; without new inputs
    //                              it only waits, slot by slot, for the consumers of the previous run (ev_bc below), so that
    //                              back-to-back runs keep the pipeline full
    HIP_TRY(hipEventRecord(p->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(se, p->ev_fork, 0));
    p->fork_needed = false;
  }
  // the device status words of this solve, cleared on the stream its first eigen kernel runs on (that kernel raises bits and
  // sweep counts behind the clear).  Bits raised by an earlier solve whose results were never fetched must not be reported
  // against this one (this solve's own evaluation, queued behind the clear, raises the tau bit again).
  auto clear_status = [&](hipStream_t st) -> hipError_t {
    hipError_t e = hipMemsetAsync(p->d.sweeps, 0, sizeof(int), st);
    if (e == hipSuccess) e = hipMemsetAsync(p->d.status, 0, sizeof(int), st);
    if (e == hipSuccess) e = hipMemsetAsync(p->d.col_status, 0, sizeof(int) * (size_t)p->d.C, st);
    return e;
  };
  if (with_solve) {
    if (p->status2[1]) {  // pipelined plan: the other set of status words (see rtd_plan::status2)
      p->stat_idx ^= 1;
      p->d.status = p->status2[p->stat_idx];
      p->d.col_status = p->col_status2[p->stat_idx];
      p->d.sweeps = p->sweeps2[p->stat_idx];
    }
    if (!pipe) {
      hipError_t e = clear_status(s);
      if (e != hipSuccess) return fail(RTD_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    }  // (pipelined: inside the first eigen stage, behind its wait for the previous run)
    p->numeric_status = 0;  // a new solve starts clean
  }
  RtdDev all = p->d;
  const RtdDev* all_tables = nullptr;
  if (with_solve && p->tables_cached && !p->tables_valid) {  // the tables of all columns, once per change of the inputs
    all.Y0 = p->Y0_all;
    all.att = p->att_all;
    if (pipe) all_tables = &all;  // launched by the first eigen stage, behind its wait: the previous run may still read them
    else {
      rtd_launch_tables(all, se, !p->quad_tables_valid);
      p->quad_tables_valid = true;
    }
    p->tables_valid = true;
  }
  const bool per_window_tables = !p->tables_cached;
  auto eigen_stage = [&](int w) {  // tables + eigen kernel of window w into hand-off slot w & 1, on the eigen stream
    const int64_t c0 = (int64_t)w * p->Cw;
    const int cnt = (int)std::min<int64_t>(p->Cw, p->d.C - c0);
    const int slot = w & 1;
    RtdDev d = window_dev(p, c0, cnt, p->retained ? 0 : slot);
    // the slot's previous tenant has been consumed (a retained plan has a place per column, not two slots: its window w is only
    // rewritten by the NEXT run, which must not overtake this run's consumers of the same columns -- the same event says so)
    if (p->bc_recorded[slot]) (void)hipStreamWaitEvent(se, p->ev_bc[slot], 0);
    if (w == 0) (void)clear_status(se);  // (a failure here shows up as the launch error checked below)
    if (w == 0 && all_tables) {
      rtd_launch_tables(*all_tables, se, !p->quad_tables_valid);
      p->quad_tables_valid = true;
      all_tables = nullptr;
    }
    if (per_window_tables) {
      rtd_launch_tables(d, se, w == 0 && !p->quad_tables_valid);
      p->quad_tables_valid = true;
    }
    rtd_launch_eig(d, se, 1);
    (void)hipEventRecord(p->ev_eig[slot], se);
  };
  if (pipe) eigen_stage(0);
  for (int w = 0; w < p->nwin; ++w) {
    const int64_t c0 = (int64_t)w * p->Cw;
    const int cnt = (int)std::min<int64_t>(p->Cw, p->d.C - c0);
    RtdDev d = window_dev(p, c0, cnt, pipe && !p->retained ? (w & 1) : 0);
    // run-path points at the layer interfaces: the boundary-condition kernel evaluates the Fourier modes there itself
    // (rtd_plan_evaluate -- the closures -- always takes the evaluation kernel, whatever the window count: a column's
    // closure values must not depend on the batch it was solved in)
    const bool fused = allow_fused && with_solve && ev && ev->antider == 0 && p->ev_iface && p->um_buf && rtd_bc_fuses_eval(d);
    d.um = fused ? p->um_buf : nullptr;
    if (tm) {
      (void)hipStreamSynchronize(s);
      harvest(p);
    }
    if (pipe) {
      if (w + 1 < p->nwin) eigen_stage(w + 1);  // queued before this window's boundary-condition kernel: they run side by side
      (void)hipStreamWaitEvent(s, p->ev_eig[w & 1], 0);
      rtd_launch_bc(d, s, 0);
      rtd_launch_bc(d, s, 1);
    } else if (with_solve) {
      mark(0);
      if (per_window_tables) {
        rtd_launch_tables(d, s, w == 0 && !p->quad_tables_valid);
        p->quad_tables_valid = true;
      }
      mark(1);
      rtd_launch_eig(d, s, 0);
      mark(2);
      rtd_launch_eig(d, s, 1);
      mark(3);
      rtd_launch_eig(d, s, 2);
      mark(4);
      rtd_launch_bc(d, s, 0);
      mark(5);
      rtd_launch_bc(d, s, 1);
    }
    if (!with_solve && ev && p->lean && p->nwin > 1) {
      // Lean retained plan, evaluation only: Y, A of this window's columns are not resident -- the eigen stage is run again for
      // the wavefront chunks the evaluation points touch (every chunk when the points touch most of them, or for the
      // one-lane-per-problem kernel of 2 ... 8 streams, whose wavefronts span columns), into hand-off slot 0; k, E, B and the
      // thermal vectors are rewritten in their retained places with the bits they had; the coefficients are kept: no
      // boundary-condition solve.  Everything queued on the plan's stream: the solve's eigen stream is behind ev_eig already.
      if (p->tables_cached && !p->tables_valid) {  // (rtd_plan_invalidate_tables since the solve)
        RtdDev all = p->d;
        all.Y0 = p->Y0_all;
        all.att = p->att_all;
        rtd_launch_tables(all, s, !p->quad_tables_valid);
        p->quad_tables_valid = p->tables_valid = true;
      }
      RtdDev dsel = d;
      const int gpw = 64 / d.NP, nchunk = (d.L + gpw - 1) / gpw;
      const int nsel = d.NP == 4 || 2 * ev->ntau >= nchunk ? 0 : ev->ntau;
      if (nsel > 0) {
        int rc = grow(p, &p->sel_buf, &p->cap_sel, ((int64_t)p->Cw * nsel + 1) / 2 + 1);
        if (rc) return rc;
        int* sel = reinterpret_cast<int*>(p->sel_buf);
        hipLaunchKernelGGL(rtd_chunk_select_kernel, dim3((cnt + 127) / 128), dim3(128), 0, s, d, ev->tau + c0 * ev->ntau, ev->ntau, gpw, nsel, sel);
        dsel.chunk_sel = sel;
        dsel.nsel = nsel;
      }
      rtd_launch_eig(dsel, s, 1);
    }
    mark(6);
    if (ev) {
      // (a gather of the previous run's results may still be in flight: it reads its own snapshot, not these buffers)
      RtdEval e = window_eval(p, *ev, c0);
      e.um_in = fused ? p->um_buf : nullptr;
      rtd_launch_eval(d, e, s);
      if (with_nt && e.u != nullptr) {
        const RtdNt nt = window_nt(p, c0);
        if (!p->nt_tables_ready) {
          rtd_launch_nt_tables(d, nt, s);
          nt_tables_done = true;
        }
        rtd_launch_nt_apply(d, nt, e, s);
      }
      mark(7);
    }
    if (pipe) {  // slot w & 1 may be filled again
      (void)hipEventRecord(p->ev_bc[w & 1], s);
      p->bc_recorded[w & 1] = true;
    }
    if (tm) {
      for (int k = 0; k < 6; ++k) p->pending[k] = with_solve;
      p->pending[6] = ev != nullptr;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RTD_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    int rc = after_window(w, c0, cnt);
    if (rc) return rc;
  }
  if (nt_tables_done) p->nt_tables_ready = true;  // every window's tables were made in this pass
  if (with_solve) p->solved = true;
  if (with_solve && !pipe) p->fork_needed = true;  // slot 0 was filled on the plan's own stream
  return 0;
}

int launch_solve(rtd_plan* p, bool with_eval, const RtdEval* ev, bool with_nt = false) {
  return launch_windows(p, true, with_eval ? ev : nullptr, with_nt, [](int, int64_t, int) { return 0; });
}

// read and clear the device status word after the stream has drained; maps it to an error code.  all_modes = false: the
// caller's outputs come from Fourier mode 0 alone (fluxes, u0): a failure confined to the modes m > 0 is not theirs -- the
// reference returns them unharmed then (and NaN for u; here u reports the failure).
int check_status(rtd_plan* p, const bool all_modes = true) {
  int st = 0;
  HIP_TRY(hipMemcpyAsync(&st, p->d.status, sizeof(int), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->quad_pending = p->cols_pending = p->ev_pending = false;
  if (st != 0) {
    HIP_TRY(hipMemsetAsync(p->d.status, 0, sizeof(int), p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  p->numeric_status |= st & ~RTD_ST_TAU;  // a failed solve stays failed for every later evaluation of it
  if (st & RTD_ST_TAU) return fail(RTD_ERR_TAU_RANGE, "tau input outside the tau range specified for the atmosphere");
  const int low = p->numeric_status & 0xFF, high = (p->numeric_status >> RTD_ST_HIGH_MODE_SHIFT) & 0xFF;
  st = low | (all_modes ? high : 0);
  if (st == 0) return 0;
  std::string what;
  if (st & RTD_ST_CHOL) what += " non-positive Cholesky pivot or non-finite eigenvalue (phase function not positive definite?);";
  if (st & RTD_ST_JACOBI) what += " Jacobi eigen-iteration did not converge;";
  if (st & RTD_ST_BEAM) what += " non-finite beam particular solution (1/mu0 coincides with an eigenvalue?);";
  if (st & RTD_ST_BC) what += " singular boundary-condition system (non-finite coefficients);";
  if (low == 0) what += " in Fourier modes m > 0 only: the fluxes and u0 of this solve are valid";
  {  // which columns (rtd_plan_get_column_status has the bits of every column)
    std::vector<int> cs((size_t)p->d.C);
    if (hipMemcpy(cs.data(), p->d.col_status, cs.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
      int64_t n = 0;
      std::string first;
      for (size_t c = 0; c < cs.size(); ++c)
        if (cs[c] & (all_modes ? 0xFFFF : 0xFF)) {
          if (n < 8) first += (n ? ", " : "") + std::to_string(c);
          ++n;
        }
      if (n > 0 && !what.empty() && what.back() == ';') what.pop_back();
      if (n > 0) what += "; " + std::to_string(n) + " of " + std::to_string(cs.size()) + " columns (" + first + (n > 8 ? ", ..." : "") + ")";
    }
  }
  return fail(RTD_ERR_NUMERIC, "numerical failure on the device:" + what);
}

}  // namespace

extern "C" {

int rtd_comm_destroy(rtd_plan* p);

int rtd_version(void) { return 212; }  // 210: rtd_plan_create_retained, rtd_plan_retained, rtd_comm_size; 211: rtd_pool_trim, rtd_pool_bytes (round 5); 212: rtd_pool_set_limit (large-block pool off by default), rtd_comm_transport (round 6)

const char* rtd_last_error(void) { return g_err.c_str(); }

int rtd_device_count(int32_t* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(RTD_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return 0;
}

static int plan_build(rtd_plan* p, const rtd_dims* dims, int32_t device, int32_t work_columns, int64_t retain_bytes, int32_t form) {
  const int N = dims->nquad / 2;
  p->dims = *dims;
  p->device = device;
  p->NP = pad_pow2(N);
  p->stream = pool().get_stream(device);
  if (!p->stream) HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  RtdDev& d = p->d;
  const int64_t C = dims->ncols, L = dims->nlayers, M = dims->nfourier, P = dims->nleg, NP = p->NP,
                Ns = dims->nscoeffs, NB = dims->nbdrf, Q2 = 2 * NP;
  d.C = (int)C; d.L = (int)L; d.N = N; d.NP = (int)NP; d.P = (int)P; d.M = (int)M; d.Ns = (int)Ns;
  d.NBDRF = (int)NB; d.beam = dims->beam ? 1 : 0;
  // RTD_BC_FORCE_PIVOT=1: every speculative elimination of the fused kernels is redone by the LDS pivoted path (bit 0);
  // =2: every chain takes the register-resident pivoted elimination of rtd_bc_mfma_kernel throughout (bit 2)
  const char* fp = getenv("RTD_BC_FORCE_PIVOT");
  d.flags = (fp ? (atoi(fp) == 2 ? 4 : 1) : 0) | (getenv("RTD_BC_FORCE_HANDOVER") ? 2 : 0);
  d.m0 = 0; d.mstep = 1; d.mtot = (int)M;
  d.l0 = 0; d.ln = (int)L;
  // window of columns whose intermediates are resident: bytes of intermediates per column.  A plan of more than one window
  // that pipelines them holds the eigen stage's hand-off buffers twice; a one-window (or RTD_NO_PIPELINE) plan once.
  const int64_t handoff_col = 8 * (M * P + (L + 1) + 2 * M * L * NP * NP + 2 * M * L * NP + M * L * Q2 + L * Ns * Q2 + L * NP + (Ns > 0 ? 4 * L * NP : 0));
  const int64_t rest_col = 8 * (M * L * Q2 + M * (L - 1) * Q2 * Q2);
  const bool may_pipeline = !getenv("RTD_NO_PIPELINE");
  const int64_t per_col_one = handoff_col + rest_col, per_col_win = (may_pipeline ? 2 : 1) * handoff_col + rest_col;
  const char* env = getenv("RTD_WORK_BYTES");
  const double budget = env ? atof(env) : 24.0 * (double)(1ull << 30);
  auto fit_window = [&](double bytes) {  // columns per window of a multi-window plan within `bytes`
    int64_t fit = (int64_t)(bytes / (double)per_col_win);
    if (fit < 1) fit = 1;
    if (fit > 256) fit -= fit % 256;
    return fit;
  };
  int64_t Cw = C;
  if (work_columns > 0) {
    // The caller's window is honoured while it fits: RTD_WORK_BYTES when set, else 80 % of the device memory that is free
    // now (the 24 GiB default is the automatic path's choice, not a limit on an explicit request).  hipMemGetInfo is only
    // asked when the request is large: one-column plans are created per pydisort() call.
    Cw = std::min<int64_t>(C, work_columns);
    const double want = (double)Cw * (double)(Cw < C ? per_col_win : per_col_one);
    if (env ? want > budget : want > 4.0 * (double)(1ull << 30)) {
      double cap = budget;
      if (!env) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        cap = 0.8 * (double)free_b;
      }
      if (want > cap) Cw = std::min<int64_t>(Cw, fit_window(cap));
    }
  } else if ((double)C * (double)per_col_one > budget) {  // (a batch that fits one window pays for one hand-off slot only)
    Cw = std::min<int64_t>(C, fit_window(budget));
  }
  p->Cw = (int)Cw;
  p->nwin = (int)((C + Cw - 1) / Cw);
  p->pipelined = p->nwin > 1 && may_pipeline;
  // Retained evaluator state (rtd_plan_create_retained): everything but the boundary-condition workspace for ALL columns, when
  // that fits the caller's budget (< 0: three tenths of the device memory that is free now)
  // that fits the caller's budget; else (or when the caller asks for it: form 2) the LEAN form: the same without Y, A -- what an
  // evaluation can recompute from the inputs without solving the boundary-condition system again (see rtd_plan::lean)
  const int64_t retain_col = handoff_col + 8 * (M * L * Q2);  // + coef
  const int64_t lean_col = retain_col - 8 * (2 * M * L * NP * NP);
  if (p->nwin > 1 && retain_bytes != 0) {
    double cap = (double)retain_bytes;
    if (retain_bytes < 0) {
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipSetDevice(device));
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      cap = 0.3 * (double)free_b;
    }
    p->retained = form != 2 && (double)C * (double)retain_col <= cap;
    // (not below 10 streams: the one-lane-per-problem eigen kernel's wavefronts span columns, and its matrices are 128 bytes)
    p->lean = !p->retained && form != 1 && NP > 4 && (double)C * (double)lean_col <= cap;
  }
  const int64_t Ch = p->retained || p->lean ? C : Cw;  // columns the coefficients, k, E, B and the thermal vectors cover
  const int64_t Cy = p->retained ? C : Cw;             // columns Y, A cover
  rtd_plan::HandOff& h1 = p->slot1;
  double *mu = nullptr, *w = nullptr, *invmu = nullptr, *S = nullptr, *T = nullptr, *omega = nullptr, *tau = nullptr,
         *taus0 = nullptr, *scale = nullptr, *wleg = nullptr, *mu0 = nullptr, *I0 = nullptr, *phi0 = nullptr,
         *rescale = nullptr, *bpos = nullptr, *bneg = nullptr, *spoly = nullptr, *bq = nullptr, *bq0 = nullptr;
  int* lperm = nullptr;
  // one arena for every fixed-size buffer (a single hipMalloc keeps plan creation cheap for one-column calls):
  // pass 0 sizes it, pass 1 carves it
  char* arena = nullptr;
  int64_t off = 0;
  for (int pass = 0; pass < 2; ++pass) {
    off = 0;
    auto carve = [&](auto** ptr, int64_t n) {
      using E = std::remove_pointer_t<std::remove_pointer_t<decltype(ptr)>>;
      if (n <= 0) n = 1;
      const int64_t bytes = (n * (int64_t)sizeof(E) + 255) / 256 * 256;
      if (pass == 1) *ptr = reinterpret_cast<E*>(arena + off);
      off += bytes;
    };
#define A(ptr, n) carve(&ptr, (n));
    A(mu, NP) A(w, NP) A(invmu, NP) A(S, NP) A(T, NP)
    A(d.Y, M * P * NP)
    // inputs: all C columns
    A(lperm, C * L) A(omega, C * L) A(tau, C * L) A(taus0, C * (L + 1)) A(scale, C * L) A(wleg, C * L * P)
    A(mu0, C) A(I0, C) A(phi0, C) A(rescale, C) A(d.col_status, C)
    A(bpos, C * M * NP) A(bneg, C * M * NP) A(spoly, C * L * Ns) A(bq, C * NB * NP * NP) A(bq0, C * NB * NP)
    // intermediates: one window of Cw columns
    A(d.Y0, Ch * M * P) A(d.att, Ch * (L + 1))
    A(d.Ym, Cy * M * L * NP * NP) A(d.Am, Cy * M * L * NP * NP) A(d.kk, Ch * M * L * NP) A(d.Bv, Ch * M * L * Q2)
    A(d.dq, Ch * L * Ns * Q2) A(d.zneg, Ch * L * NP) A(d.vb, Ns > 0 ? Ch * L * 4 * NP : 1) A(d.coef, Ch * M * L * Q2)
    A(d.Fws, Cw * M * (L - 1) * Q2 * Q2) A(d.Ek, Ch * M * L * NP) A(d.need_split, Cw * M)
    A(d.sweeps, 1) A(d.status, 1) A(d.split_any, 1)
    if (p->pipelined && !p->retained) {
      A(h1.Ym, Cw * M * L * NP * NP) A(h1.Am, Cw * M * L * NP * NP)
      if (!p->lean) {  // (a lean plan keeps these per column: window_dev)
        A(h1.Y0, Cw * M * P) A(h1.att, Cw * (L + 1))
        A(h1.kk, Cw * M * L * NP) A(h1.Bv, Cw * M * L * Q2) A(h1.dq, Cw * L * Ns * Q2) A(h1.zneg, Cw * L * NP) A(h1.vb, Ns > 0 ? Cw * L * 4 * NP : 1) A(h1.Ek, Cw * M * L * NP)
      }
    }
    if (p->pipelined) {
      A(p->status2[1], 1) A(p->col_status2[1], C) A(p->sweeps2[1], 1)
    }
#undef A
    if (pass == 0) {
      void* q = nullptr;
      size_t got = 0;
      hipError_t e = pooled_malloc(&q, (size_t)off, device, &got);
      if (e != hipSuccess)
        return fail(RTD_ERR_HIP, std::string("hipMalloc of ") + std::to_string(off) + " bytes: " + hipGetErrorString(e));
      p->allocs.push_back(q);
      p->alloc_bytes.push_back(got);
      p->bytes += (int64_t)got;
      arena = static_cast<char*>(q);
    }
  }
  d.mu = mu; d.w = w; d.invmu = invmu; d.S = S; d.T = T;
  d.omega = omega; d.tau = tau; d.taus0 = taus0; d.scale = scale; d.wleg = wleg;
  d.mu0 = mu0; d.I0 = I0; d.phi0 = phi0; d.rescale = rescale; d.bpos = bpos; d.bneg = bneg;
  d.spoly = spoly; d.bdrfq = bq; d.bdrfq0 = bq0; d.lperm = lperm;
  if (p->nwin == 1 || p->retained || p->lean) {
    p->Y0_all = d.Y0;
    p->att_all = d.att;
    p->tables_cached = true;
  } else if ((double)C * (double)(M * P + L + 1) * 8.0 <= 2.0 * (double)(1ull << 30)) {
    int rc;
    if ((rc = p->alloc(&p->Y0_all, C * M * P))) return rc;
    if ((rc = p->alloc(&p->att_all, C * (L + 1)))) return rc;
    p->tables_cached = true;
  }
  // (d.sweeps, d.status, d.split_any and d.Bv, d.dq were carved next to each other: one fill each -- a fill costs ~8 us of host
  //  time, which a one-column pydisort() call notices)
  HIP_TRY(hipMemsetAsync(d.sweeps, 0, (size_t)(reinterpret_cast<char*>(d.split_any) - reinterpret_cast<char*>(d.sweeps)) + sizeof(int), p->stream));
  HIP_TRY(hipMemsetAsync(d.col_status, 0, sizeof(int) * (size_t)C, p->stream));
  if (p->pipelined) {
    p->status2[0] = d.status; p->col_status2[0] = d.col_status; p->sweeps2[0] = d.sweeps;
    HIP_TRY(hipMemsetAsync(p->status2[1], 0, sizeof(int), p->stream));
    HIP_TRY(hipMemsetAsync(p->col_status2[1], 0, sizeof(int) * (size_t)C, p->stream));
    HIP_TRY(hipMemsetAsync(p->sweeps2[1], 0, sizeof(int), p->stream));
  }
  HIP_TRY(hipMemsetAsync(d.Bv, 0, (size_t)(reinterpret_cast<char*>(d.dq) - reinterpret_cast<char*>(d.Bv)) + (Ns > 0 ? (size_t)(Ch * L * Ns * Q2) * 8 : 0), p->stream));
  if (p->pipelined) {
    if (!p->retained && !p->lean) {
      HIP_TRY(hipMemsetAsync(h1.Bv, 0, (size_t)(Cw * M * L * Q2) * 8, p->stream));
      if (Ns > 0) HIP_TRY(hipMemsetAsync(h1.dq, 0, (size_t)(Cw * L * Ns * Q2) * 8, p->stream));
    }
    // (stream priorities either way changed nothing: profiles/archive/r03_experiments.json)
    p->eig_stream = pool().get_stream(device);
    if (!p->eig_stream) HIP_TRY(hipStreamCreateWithFlags(&p->eig_stream, hipStreamNonBlocking));
    for (hipEvent_t* e : {&p->ev_eig[0], &p->ev_eig[1], &p->ev_bc[0], &p->ev_bc[1], &p->ev_fork})
      HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  // (no synchronisation here: the fills are stream-ordered before everything the plan does later, and waiting for them costs a
  //  one-column call ~15 us; a failed fill surfaces at the next synchronising call)
  return 0;
}

int rtd_plan_create_retained(const rtd_dims* dims, int32_t device, int32_t work_columns, int64_t retain_bytes, rtd_plan** out) {
  return rtd_plan_create_retained_form(dims, device, work_columns, retain_bytes, 0, out);
}

int rtd_plan_create_retained_form(const rtd_dims* dims, int32_t device, int32_t work_columns, int64_t retain_bytes, int32_t form,
                                  rtd_plan** out) {
  if (!dims || !out) return fail(RTD_ERR_ARG, "null argument");
  *out = nullptr;
  if (form < 0 || form > 2) return fail(RTD_ERR_ARG, "retained form: 0 (full, else lean), 1 (full only) or 2 (lean)");
  const int N = dims->nquad / 2;
  if (dims->ncols < 1 || dims->nlayers < 1 || dims->nquad < 2 || (dims->nquad & 1) || dims->nleg < 1 ||
      dims->nfourier < 1 || dims->nfourier > dims->nleg || dims->nscoeffs < 0 || dims->nbdrf < 0)
    return fail(RTD_ERR_ARG, "invalid dimensions");
  // include/rtd.h: nleg <= nquad (pydisort.py:232-234 of the reference).  The one-lane-per-problem eigen kernel of 2 ... 8 streams
  // reads exactly 2 NP moments per parity: more would be dropped silently.
  if (dims->nleg > dims->nquad) return fail(RTD_ERR_ARG, "invalid dimensions: nleg > nquad");
  // 2 ... 64 streams; 66 ... 128 streams: one eigenproblem per wavefront, one boundary-condition chain per workgroup (rtd_bc_wide.hip)
  if (N > 64) return fail(RTD_ERR_ARG, "NQuad > 128 is not supported by this build (N = NQuad/2 <= 64)");
  HIP_TRY(hipSetDevice(device));
  rtd_plan* p = new rtd_plan();
  const int rc = plan_build(p, dims, device, work_columns, retain_bytes, form);
  if (rc) {
    const std::string keep = g_err;  // rtd_plan_destroy must not lose the message
    rtd_plan_destroy(p);
    g_err = keep;
    return rc;
  }
  *out = p;
  return 0;
}

int rtd_plan_create_windowed(const rtd_dims* dims, int32_t device, int32_t work_columns, rtd_plan** out) {
  return rtd_plan_create_retained(dims, device, work_columns, 0, out);
}

int rtd_plan_retained(rtd_plan* p, int32_t* retained) {
  if (!p || !retained) return fail(RTD_ERR_ARG, "null argument");
  *retained = p->nwin == 1 || p->retained ? 1 : p->lean ? 2 : 0;
  return 0;
}

int rtd_plan_create(const rtd_dims* dims, int32_t device, rtd_plan** out) {
  return rtd_plan_create_windowed(dims, device, 0, out);
}

int rtd_plan_windows(rtd_plan* p, int32_t* work_columns, int32_t* nwindows) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  if (work_columns) *work_columns = p->Cw;
  if (nwindows) *nwindows = p->nwin;
  return 0;
}

int rtd_plan_destroy(rtd_plan* p) {
  if (!p) return 0;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->comm_stream) (void)hipStreamSynchronize(p->comm_stream);
  if (p->copy_stream) (void)hipStreamSynchronize(p->copy_stream);
  if (p->eig_stream) (void)hipStreamSynchronize(p->eig_stream);
  rtd_comm_destroy(p);
  for (size_t i = 0; i < p->allocs.size(); ++i)
    if (p->allocs[i]) pooled_free(p->allocs[i], p->alloc_bytes[i], p->device);
  for (auto& e : p->evt)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : {p->ev_results, p->ev_gathered, p->ev_win[0], p->ev_win[1], p->ev_copied[0], p->ev_copied[1], p->ev_eig[0],
                       p->ev_eig[1], p->ev_bc[0], p->ev_bc[1], p->ev_fork})
    if (e) (void)hipEventDestroy(e);
  // (every stream was drained above: the streams and the pinned slabs can serve another plan at once)
  if (p->eig_stream) pool().put_stream(p->eig_stream, p->device);
  for (int k = 0; k < 2; ++k)
    if (p->stage[k] && !pool().put_host(p->stage[k], p->stage_cap[k])) (void)hipHostFree(p->stage[k]);
  if (p->comm_stream) (void)hipStreamDestroy(p->comm_stream);
  if (p->copy_stream) pool().put_stream(p->copy_stream, p->device);
  if (p->stream) pool().put_stream(p->stream, p->device);
  delete p;
  return 0;
}

int rtd_plan_synchronize(rtd_plan* p) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->quad_pending = p->cols_pending = p->ev_pending = false;
  if (p->eig_stream) HIP_TRY(hipStreamSynchronize(p->eig_stream));  // (every eigen stage is consumed on `stream`: a formality)
  if (p->comm_stream) HIP_TRY(hipStreamSynchronize(p->comm_stream));
  return 0;
}

int rtd_plan_get_column_status(rtd_plan* p, int32_t* status) {
  if (!p || !status) return fail(RTD_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (p->eig_stream) HIP_TRY(hipStreamSynchronize(p->eig_stream));
  HIP_TRY(hipMemcpy(status, p->d.col_status, sizeof(int32_t) * (size_t)p->d.C, hipMemcpyDeviceToHost));
  return 0;
}

int rtd_plan_device_bytes(rtd_plan* p, int64_t* bytes) {
  if (!p || !bytes) return fail(RTD_ERR_ARG, "null argument");
  *bytes = p->bytes;
  return 0;
}

int rtd_pool_trim(int32_t device, int64_t* released) {
  const size_t n = pool().trim(device);
  if (released) *released = (int64_t)n;
  return 0;
}

int rtd_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes) {
  size_t fr = 0, total = 0;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemGetInfo(&fr, &total));
  if (free_bytes) *free_bytes = (int64_t)fr;
  if (total_bytes) *total_bytes = (int64_t)total;
  return 0;
}

int rtd_pool_set_limit(int64_t bytes, int32_t device, int64_t* previous) {
  const int64_t prev = pool().set_limit(bytes, device);
  if (previous) *previous = prev;
  return 0;
}

int rtd_pool_bytes(int64_t* cached) {
  if (!cached) return fail(RTD_ERR_ARG, "null argument");
  DevPool& q = pool();
  std::lock_guard<std::mutex> g(q.m);
  *cached = (int64_t)(q.cached + q.big_cached);
  return 0;
}

int rtd_plan_set_quadrature(rtd_plan* p, const double* mu_pos, const double* weights) {
  if (!p || !mu_pos || !weights) return fail(RTD_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(p->device));
  const int N = p->d.N, NP = p->NP;
  std::vector<double> mu(NP), w(NP), im(NP), S(NP), T(NP);
  for (int i = 0; i < NP; ++i) {
    if (i < N) {
      if (!(mu_pos[i] > 0.0) || !(weights[i] > 0.0)) return fail(RTD_ERR_ARG, "quadrature nodes/weights must be positive");
      mu[i] = mu_pos[i]; w[i] = weights[i];
      S[i] = std::sqrt(w[i] / mu[i]); T[i] = std::sqrt(w[i] * mu[i]);
    } else {  // padding stream: decoupled, never the beam direction (1/mu = 0.5 < 1/mu0)
      mu[i] = 2.0; w[i] = 0.0; S[i] = 0.0; T[i] = 1.0;
    }
    im[i] = 1.0 / mu[i];
  }
  // mu, w, 1/mu, S, T were carved one after the other (plan_build): ONE copy of a host image of that stretch instead of five
  // (a small copy from pageable memory costs ~5 us of host time: it matters to a one-column pydisort() call)
  const size_t nb = (size_t)NP * 8;
  const char* base = reinterpret_cast<const char*>(p->d.mu);
  const size_t span = (size_t)(reinterpret_cast<const char*>(p->d.T) - base) + nb;
  if (p->quad_pending) HIP_TRY(hipStreamSynchronize(p->stream));  // (an earlier upload may still read the image)
  std::vector<char>& img = p->quad_img;
  img.assign(span, 0);
  std::memcpy(img.data(), mu.data(), nb);
  std::memcpy(img.data() + (reinterpret_cast<const char*>(p->d.w) - base), w.data(), nb);
  std::memcpy(img.data() + (reinterpret_cast<const char*>(p->d.invmu) - base), im.data(), nb);
  std::memcpy(img.data() + (reinterpret_cast<const char*>(p->d.S) - base), S.data(), nb);
  std::memcpy(img.data() + (reinterpret_cast<const char*>(p->d.T) - base), T.data(), nb);
  HIP_TRY(hipMemcpyAsync((void*)p->d.mu, img.data(), span, hipMemcpyHostToDevice, p->stream));
  p->quad_pending = true;  // (no wait: the image stays with the plan)
  p->have_quad = true;
  p->fork_needed = true;
  p->tables_valid = false;
  p->quad_tables_valid = false;
  p->solved = false;
  return 0;
}

int rtd_plan_set_columns(rtd_plan* p, const double* scaled_omega, const double* tau, const double* taus0,
                         const double* scale_tau, const double* wleg, const double* mu0, const double* I0,
                         const double* phi0, const double* rescale, const double* b_pos, const double* b_neg,
                         const double* s_poly, const double* bdrf_q, const double* bdrf_q0) {
  if (!p || !scaled_omega || !tau || !taus0 || !scale_tau || !wleg || !mu0 || !I0 || !phi0 || !rescale)
    return fail(RTD_ERR_ARG, "null argument");
  const RtdDev& d = p->d;
  if (d.Ns > 0 && !s_poly) return fail(RTD_ERR_ARG, "s_poly is required when nscoeffs > 0");
  if (d.NBDRF > 0 && (!bdrf_q || !bdrf_q0)) return fail(RTD_ERR_ARG, "BDRF tables are required when nbdrf > 0");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = d.C, L = d.L, M = d.M, P = d.P, N = d.N, NP = d.NP, Ns = d.Ns, NB = d.NBDRF;
  // a plan created without a beam skips the beam terms on the device: refuse inputs that have one
  if (!d.beam)
    for (int64_t c = 0; c < C; ++c)
      if (I0[c] > 0.0) return fail(RTD_ERR_ARG, "the plan was created with beam = 0 but a column has I0 > 0");
  hipStream_t s = p->stream;
  // The per-column inputs were carved one after the other, lperm first and the BDRF tables last (plan_build).  A small plan -- a
  // one-column pydisort() call above all -- gets ONE copy of a host image of that stretch instead of fifteen small ones (~5 us of
  // host time each from pageable memory); d.col_status lies inside the stretch and is cleared by every solve anyway.
  const char* in_base = reinterpret_cast<const char*>(d.lperm);
  const size_t in_span = (size_t)(reinterpret_cast<const char*>(d.bdrfq0) - in_base) + (size_t)std::max<int64_t>(C * NB * NP, 1) * 8;
  const bool one_copy = in_span <= (size_t)(1u << 20);
  if (one_copy && p->cols_pending) HIP_TRY(hipStreamSynchronize(s));  // (an earlier upload may still read the image)
  std::vector<char>& img = p->cols_img;
  if (one_copy) img.assign(in_span, 0);
#define UP(dst, src, n)                                                                                              \
  do {                                                                                                                \
    if (one_copy) std::memcpy(img.data() + (reinterpret_cast<const char*>(dst) - in_base), (src), (size_t)(n) * 8);   \
    else HIP_TRY(hipMemcpyAsync((void*)(dst), (src), (size_t)(n) * 8, hipMemcpyHostToDevice, s));                    \
  } while (0)
  UP(d.omega, scaled_omega, C * L);
  // Layer order of the eigen stage: a wavefront iterates until the slowest of its 64/NP eigenproblems has converged,
  // and the Jacobi sweep count grows with omega* / (1 - g*) (g* = the first scaled moment).  Sorting each column's layers
  // by that key before they are dealt to wavefronts brings the mean of the per-wavefront maximum from 3.1 to 2.8
  // sweeps on the benchmark columns (2.73 is the mean per problem).  Results do not depend on the order.
  std::vector<int> perm((size_t)(C * L));
  {
    std::vector<std::pair<double, int>> key((size_t)L);
    for (int64_t c = 0; c < C; ++c) {
      for (int64_t l = 0; l < L; ++l) {
        const double g = P > 1 ? wleg[(c * L + l) * P + 1] / 3.0 : 0.0;
        key[(size_t)l] = {scaled_omega[c * L + l] / std::max(1.0 - g, 1e-6), (int)l};
      }
      std::sort(key.begin(), key.end());
      for (int64_t l = 0; l < L; ++l) perm[(size_t)(c * L + l)] = key[(size_t)l].second;
    }
  }
  if (one_copy) std::memcpy(img.data(), perm.data(), perm.size() * sizeof(int));
  else HIP_TRY(hipMemcpyAsync((void*)d.lperm, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice, s));
  UP(d.tau, tau, C * L);
  UP(d.taus0, taus0, C * (L + 1));
  UP(d.scale, scale_tau, C * L);
  UP(d.wleg, wleg, C * L * P);
  UP(d.mu0, mu0, C);
  UP(d.I0, I0, C);
  UP(d.phi0, phi0, C);
  UP(d.rescale, rescale, C);
  // The thermal source polynomials arrive as the reference's scaled_s_poly_coeffs: coefficients in the ABSOLUTE scaled optical
  // depth.  The kernels keep them about the top of their own layer (rtd_dd.h): Taylor shift by taus0[l], in double-double.
  std::vector<double> sloc;
  if (Ns > 0) {
    sloc.assign(s_poly, s_poly + C * L * Ns);
    for (int64_t c = 0; c < C; ++c)
      for (int64_t l = 0; l < L; ++l) rtd_taylor_shift(sloc.data() + (c * L + l) * Ns, (int)Ns, taus0[c * (L + 1) + l]);
    UP(d.spoly, sloc.data(), C * L * Ns);
  }
  // pad the per-stream arrays from N to NP
  std::vector<double> bp, bn;
  if (N == NP) {
    if (b_pos) UP(d.bpos, b_pos, C * M * NP);
    else if (!one_copy) HIP_TRY(hipMemsetAsync((void*)d.bpos, 0, (size_t)(C * M * NP) * 8, s));  // (the image is zero there)
    if (b_neg) UP(d.bneg, b_neg, C * M * NP);
    else if (!one_copy) HIP_TRY(hipMemsetAsync((void*)d.bneg, 0, (size_t)(C * M * NP) * 8, s));
  } else {
    bp.assign((size_t)(C * M * NP), 0.0);
    bn.assign((size_t)(C * M * NP), 0.0);
    for (int64_t cm = 0; cm < C * M; ++cm)
      for (int64_t i = 0; i < N; ++i) {
        if (b_pos) bp[cm * NP + i] = b_pos[cm * N + i];
        if (b_neg) bn[cm * NP + i] = b_neg[cm * N + i];
      }
    UP(d.bpos, bp.data(), C * M * NP);
    UP(d.bneg, bn.data(), C * M * NP);
  }
  std::vector<double> q, q0;
  if (NB > 0) {
    q.assign((size_t)(C * NB * NP * NP), 0.0);
    q0.assign((size_t)(C * NB * NP), 0.0);
    for (int64_t cb = 0; cb < C * NB; ++cb)
      for (int64_t i = 0; i < N; ++i) {
        if (bdrf_q0) q0[cb * NP + i] = bdrf_q0[cb * N + i];
        if (bdrf_q)
          for (int64_t j = 0; j < N; ++j) q[(cb * NP + i) * NP + j] = bdrf_q[(cb * N + i) * N + j];
      }
    UP(d.bdrfq, q.data(), C * NB * NP * NP);
    UP(d.bdrfq0, q0.data(), C * NB * NP);
  }
#undef UP
  if (one_copy) {
    HIP_TRY(hipMemcpyAsync((void*)d.lperm, img.data(), in_span, hipMemcpyHostToDevice, s));
    p->cols_pending = true;  // (no wait: the image stays with the plan; the caller's arrays have been copied into it)
  } else {
    HIP_TRY(hipStreamSynchronize(s));  // the caller's arrays and the host staging vectors must outlive their copies
  }
  p->h_tau.assign(tau, tau + C * L);
  p->ev_iface = false;  // stored evaluation points, if any, are no longer known to be this batch's interfaces
  p->have_cols = true;
  p->fork_needed = true;
  p->tables_valid = false;
  p->solved = false;
  return 0;
}

int rtd_plan_set_columns_raw(rtd_plan* p, const double* tau_arr, const double* omega_arr, const double* leg_all,
                             int32_t nleg_all, const double* f_arr, const double* mu0, const double* I0,
                             const double* phi0, const double* b_pos, const double* b_neg, const double* s_poly,
                             const double* bdrf_q, const double* bdrf_q0) {
  if (!p || !tau_arr || !omega_arr || !leg_all || !f_arr || !mu0 || !I0 || !phi0) return fail(RTD_ERR_ARG, "null argument");
  const RtdDev& d = p->d;
  if (nleg_all < d.P) return fail(RTD_ERR_ARG, "nleg_all must be >= nleg");
  if (d.Ns > 0 && !s_poly) return fail(RTD_ERR_ARG, "s_poly is required when nscoeffs > 0");
  if (d.NBDRF > 0 && (!bdrf_q || !bdrf_q0)) return fail(RTD_ERR_ARG, "BDRF tables are required when nbdrf > 0");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = d.C, L = d.L, M = d.M, N = d.N, NP = d.NP, Ns = d.Ns, NB = d.NBDRF;
  if (!d.beam)
    for (int64_t c = 0; c < C; ++c)
      if (I0[c] > 0.0) return fail(RTD_ERR_ARG, "the plan was created with beam = 0 but a column has I0 > 0");
  hipStream_t s = p->stream;
  // one temporary device block for the raw arrays
  const int64_t n_cl = C * L, n_leg = C * L * nleg_all, n_b = C * M * N, n_sp = C * L * Ns;
  const int64_t total = 3 * n_cl + n_leg + 3 * C + (b_pos ? n_b : 0) + (b_neg ? n_b : 0) + n_sp;
  // staging block from the process-wide pool (a raw hipMalloc / hipFree pair synchronises the whole device per call)
  void* raw_v = nullptr;
  size_t raw_got = 0;
  HIP_TRY(pooled_malloc(&raw_v, (size_t)total * 8, p->device, &raw_got));
  double* raw = (double*)raw_v;
  RtdRaw r{};
  r.nleg_all = nleg_all;
  double* q = raw;
  hipError_t e = hipSuccess;
  auto up = [&](const double*& dst, const double* src, int64_t n) {
    dst = q;
    if (e == hipSuccess && n > 0) e = hipMemcpyAsync(q, src, (size_t)n * 8, hipMemcpyHostToDevice, s);
    q += n;
  };
  up(r.tau, tau_arr, n_cl); up(r.omega, omega_arr, n_cl); up(r.f, f_arr, n_cl); up(r.leg, leg_all, n_leg);
  up(r.mu0, mu0, C); up(r.I0, I0, C); up(r.phi0, phi0, C);
  if (b_pos) up(r.bpos, b_pos, n_b);
  if (b_neg) up(r.bneg, b_neg, n_b);
  if (Ns > 0) up(r.spoly, s_poly, n_sp);
  std::vector<double> qh, q0h;
  if (e == hipSuccess && NB > 0) {  // BDRF tables: padded from N to NP on the host (small)
    qh.assign((size_t)(C * NB * NP * NP), 0.0);
    q0h.assign((size_t)(C * NB * NP), 0.0);
    for (int64_t cb = 0; cb < C * NB; ++cb)
      for (int64_t i = 0; i < N; ++i) {
        q0h[cb * NP + i] = bdrf_q0[cb * N + i];
        for (int64_t j = 0; j < N; ++j) qh[(cb * NP + i) * NP + j] = bdrf_q[(cb * N + i) * N + j];
      }
    e = hipMemcpyAsync((void*)d.bdrfq, qh.data(), qh.size() * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync((void*)d.bdrfq0, q0h.data(), q0h.size() * 8, hipMemcpyHostToDevice, s);
  }
  if (e == hipSuccess) {
    rtd_launch_prepare(d, r, s);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  else (void)hipStreamSynchronize(s);  // nothing may still read the block when it goes back to the pool
  pooled_free(raw, raw_got, p->device);
  if (e != hipSuccess) return fail(RTD_ERR_HIP, std::string("set_columns_raw: ") + hipGetErrorString(e));
  p->h_tau.assign(tau_arr, tau_arr + C * L);
  p->ev_iface = false;
  p->have_cols = true;
  p->fork_needed = true;
  p->tables_valid = false;
  p->solved = false;
  return 0;
}

int rtd_plan_set_bdrf_samples(rtd_plan* p, int32_t nphi, const double* rho_qq, const double* rho_q0) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  p->fork_needed = true;  // works on the hand-off buffers from the plan's own stream
  if (!p->have_cols) return fail(RTD_ERR_STATE, "set_columns must precede set_bdrf_samples");
  const RtdDev& d = p->d;
  if (d.NBDRF <= 0) return fail(RTD_ERR_ARG, "the plan was created with nbdrf = 0");
  if (nphi < 2 || nphi > 16384 || !rho_qq) return fail(RTD_ERR_ARG, "need 2 <= nphi <= 16384 and rho_qq");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int64_t nqq = (int64_t)d.C * d.N * d.N * nphi, nq0 = rho_q0 ? (int64_t)d.C * d.N * nphi : 0;
  double* tmp = nullptr;
  HIP_TRY(hipMalloc(&tmp, (size_t)(nqq + nq0) * sizeof(double)));
  hipError_t e = hipMemcpyAsync(tmp, rho_qq, (size_t)nqq * sizeof(double), hipMemcpyHostToDevice, s);
  if (e == hipSuccess && nq0)
    e = hipMemcpyAsync(tmp + nqq, rho_q0, (size_t)nq0 * sizeof(double), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(const_cast<double*>(d.bdrfq), 0, (size_t)d.C * d.NBDRF * d.NP * d.NP * sizeof(double), s);
  if (e == hipSuccess) e = hipMemsetAsync(const_cast<double*>(d.bdrfq0), 0, (size_t)d.C * d.NBDRF * d.NP * sizeof(double), s);
  if (e == hipSuccess) {
    rtd_launch_bdrf_modes(d, nphi, tmp, nq0 ? tmp + nqq : nullptr, s);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(RTD_ERR_HIP, hipGetErrorString(e));
  p->solved = false;
  return 0;
}

int rtd_plan_set_mode_shard(rtd_plan* p, int32_t first, int32_t stride, int32_t total) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  RtdDev& d = p->d;
  if (first < 0 || stride < 1 || total < 1 || first + (int64_t)stride * (d.M - 1) > total - 1)
    return fail(RTD_ERR_ARG, "mode shard: need 0 <= first, stride >= 1 and first + stride (nfourier - 1) <= total - 1");
  if (total > d.P) return fail(RTD_ERR_ARG, "mode shard: total number of Fourier modes exceeds nleg");
  d.m0 = first;
  d.mstep = stride;
  d.mtot = total;
  p->solved = false;
  p->tables_valid = false;
  p->quad_tables_valid = false;
  return 0;
}

int rtd_plan_invalidate_tables(rtd_plan* p) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  p->tables_valid = false;
  p->quad_tables_valid = false;  // (everything a fresh call would compute, the column-independent table included)
  p->fork_needed = true;  // as after an upload: the next solve starts behind everything queued on the plan's stream
  return 0;
}

int rtd_plan_solve(rtd_plan* p) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  if (!p->have_quad || !p->have_cols) return fail(RTD_ERR_STATE, "set_quadrature and set_columns must precede solve");
  HIP_TRY(hipSetDevice(p->device));
  return launch_solve(p, false, nullptr);
}

int rtd_plan_set_eval_points(rtd_plan* p, int32_t ntau, const double* tau, int32_t nphi, const double* phi) {
  if (!p || !tau || ntau < 1 || nphi < 0 || (nphi > 0 && !phi)) return fail(RTD_ERR_ARG, "bad evaluation points");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = p->d.C, Qr = 2 * p->d.N;
  int rc;
  if ((rc = grow(p, &p->ev_tau, &p->cap_tau, C * ntau))) return rc;
  if ((rc = grow(p, &p->ev_phi, &p->cap_phi, nphi > 0 ? nphi : 1))) return rc;
  if ((rc = grow(p, &p->ev_u, &p->cap_u, C * Qr * ntau * (nphi > 0 ? nphi : 1)))) return rc;
  if ((rc = grow(p, &p->ev_u0, &p->cap_u0, 2 * C * Qr * ntau))) return rc;  // u0 and ulast
  if ((rc = grow(p, &p->ev_fl, &p->cap_fl, 3 * C * ntau))) return rc;
  if ((C * ntau + nphi) * 8 <= (int64_t)(1 << 20)) {  // small: through an image that stays with the plan, no wait
    if (p->ev_pending) HIP_TRY(hipStreamSynchronize(p->stream));
    p->ev_img.assign(tau, tau + C * ntau);
    if (nphi > 0) p->ev_img.insert(p->ev_img.end(), phi, phi + nphi);
    HIP_TRY(hipMemcpyAsync(p->ev_tau, p->ev_img.data(), (size_t)(C * ntau) * 8, hipMemcpyHostToDevice, p->stream));
    if (nphi > 0) HIP_TRY(hipMemcpyAsync(p->ev_phi, p->ev_img.data() + C * ntau, (size_t)nphi * 8, hipMemcpyHostToDevice, p->stream));
    p->ev_pending = true;
  } else {
    HIP_TRY(hipMemcpyAsync(p->ev_tau, tau, (size_t)(C * ntau) * 8, hipMemcpyHostToDevice, p->stream));
    if (nphi > 0) HIP_TRY(hipMemcpyAsync(p->ev_phi, phi, (size_t)nphi * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  p->ev_ntau = ntau;
  p->ev_nphi = nphi;
  // points = the layer interfaces [0, tau_arr] of every column?  Then rtd_plan_run takes the fused evaluation.
  const int64_t L = p->d.L;
  bool iface = ntau == L + 1 && (int64_t)p->h_tau.size() == C * L;
  for (int64_t c = 0; iface && c < C; ++c) {
    const double* t = tau + c * ntau;
    iface = t[0] == 0.0 && std::memcmp(t + 1, p->h_tau.data() + c * L, (size_t)L * 8) == 0;
  }
  p->ev_iface = iface;
  if (iface && rtd_bc_fuses_eval(p->d)) {
    if ((rc = grow(p, &p->um_buf, &p->cap_um, (int64_t)p->Cw * p->d.M * (L + 1) * 2 * p->d.NP))) return rc;
  }
  return 0;
}

static RtdEval make_eval(rtd_plan* p, int antider, bool want_u) {
  RtdEval e{};
  const int64_t C = p->d.C, Qr = 2 * p->d.N;
  e.ntau = p->ev_ntau;
  e.nphi = p->ev_nphi;
  e.antider = antider;
  e.tau = p->ev_tau;
  e.phi = p->ev_phi;
  e.u = (want_u && p->ev_nphi > 0) ? p->ev_u : nullptr;
  e.u0 = p->ev_u0;
  e.ulast = p->ev_u0 + C * Qr * p->ev_ntau;
  e.fup = p->ev_fl;
  e.fdn = p->ev_fl + C * p->ev_ntau;
  e.fdir = p->ev_fl + 2 * C * p->ev_ntau;
  return e;
}

int rtd_plan_run(rtd_plan* p) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  if (!p->have_quad || !p->have_cols || p->ev_ntau < 1) return fail(RTD_ERR_STATE, "inputs or evaluation points missing");
  HIP_TRY(hipSetDevice(p->device));
  RtdEval e = make_eval(p, 0, true);
  return launch_solve(p, true, &e, p->have_nt);
}

static int fetch_queued(rtd_plan* p, double* u, double* u0, double* flux_up, double* fdn, double* fdir) {
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = p->d.C, Qr = 2 * p->d.N, nt = p->ev_ntau, np = p->ev_nphi;
  hipStream_t s = p->stream;
  if (u && np > 0) HIP_TRY(hipMemcpyAsync(u, p->ev_u, (size_t)(C * Qr * nt * np) * 8, hipMemcpyDeviceToHost, s));
  if (u0) HIP_TRY(hipMemcpyAsync(u0, p->ev_u0, (size_t)(C * Qr * nt) * 8, hipMemcpyDeviceToHost, s));
  if (flux_up) HIP_TRY(hipMemcpyAsync(flux_up, p->ev_fl, (size_t)(C * nt) * 8, hipMemcpyDeviceToHost, s));
  if (fdn) HIP_TRY(hipMemcpyAsync(fdn, p->ev_fl + C * nt, (size_t)(C * nt) * 8, hipMemcpyDeviceToHost, s));
  if (fdir) HIP_TRY(hipMemcpyAsync(fdir, p->ev_fl + 2 * C * nt, (size_t)(C * nt) * 8, hipMemcpyDeviceToHost, s));
  return check_status(p, u != nullptr);  // (u0 and the fluxes come from Fourier mode 0 alone)
}

// Whatever goes wrong, no copy into the caller's (pageable) arrays is left in flight behind the return: the caller may free or
// reuse them at once.  check_status drains the stream on its own way out; the early returns before it do not.
static int drained(rtd_plan* p, int rc) {
  if (rc != 0 && p->stream) {
    const std::string keep = g_err;
    (void)hipStreamSynchronize(p->stream);
    g_err = keep;
  }
  return rc;
}

int rtd_plan_fetch(rtd_plan* p, double* u, double* u0, double* flux_up, double* fdn, double* fdir) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  if (p->ev_ntau < 1 || !p->solved) return fail(RTD_ERR_STATE, "nothing to fetch");
  return drained(p, fetch_queued(p, u, u0, flux_up, fdn, fdir));
}

// solve + evaluate + copy out, window by window: the device-to-host copy of window w (copy stream, pinned staging)
// and the host memcpy into the caller's arrays overlap the kernels of window w + 1
int rtd_plan_run_fetch(rtd_plan* p, double* u, double* u0, double* flux_up, double* fdn, double* fdir) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  if (!p->have_quad || !p->have_cols || p->ev_ntau < 1) return fail(RTD_ERR_STATE, "inputs or evaluation points missing");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = p->d.C, Qr = 2 * p->d.N, nt = p->ev_ntau, np = p->ev_nphi;
  const int64_t per_u = (u && np > 0) ? Qr * nt * np : 0, per_u0 = u0 ? Qr * nt : 0, per_fl = nt;
  const int nfl = (flux_up ? 1 : 0) + (fdn ? 1 : 0) + (fdir ? 1 : 0);
  const size_t slab = (size_t)p->Cw * (size_t)(per_u + per_u0 + nfl * per_fl) * 8;
  if (!p->copy_stream) p->copy_stream = pool().get_stream(p->device);
  if (!p->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
  for (int k = 0; k < 2; ++k) {
    if (!p->ev_win[k]) HIP_TRY(hipEventCreateWithFlags(&p->ev_win[k], hipEventDisableTiming));
    if (!p->ev_copied[k]) HIP_TRY(hipEventCreateWithFlags(&p->ev_copied[k], hipEventDisableTiming));
  }
  if (slab > p->stage_bytes) {
    for (int k = 0; k < 2; ++k) {
      char*& st = p->stage[k];
      if (st && !pool().put_host(st, p->stage_cap[k])) (void)hipHostFree(st);
      st = static_cast<char*>(pool().get_host(slab, &p->stage_cap[k]));
      if (!st) HIP_TRY(hipHostMalloc((void**)&st, p->stage_cap[k] = slab, hipHostMallocDefault));
    }
    p->stage_bytes = slab;
  }
  double* fls[3] = {flux_up, fdn, fdir};
  // host side of window w: wait for its copy, scatter the staging slab into the caller's arrays
  auto drain = [&](int w) -> int {
    const int k = w & 1;
    const int64_t c0 = (int64_t)w * p->Cw, cnt = std::min<int64_t>(p->Cw, C - c0);
    HIP_TRY(hipEventSynchronize(p->ev_copied[k]));
    const char* src = p->stage[k];
    if (per_u) { std::memcpy(u + c0 * per_u, src, (size_t)(cnt * per_u) * 8); src += cnt * per_u * 8; }
    if (per_u0) { std::memcpy(u0 + c0 * per_u0, src, (size_t)(cnt * per_u0) * 8); src += cnt * per_u0 * 8; }
    for (int f = 0; f < 3; ++f)
      if (fls[f]) { std::memcpy(fls[f] + c0 * nt, src, (size_t)(cnt * nt) * 8); src += cnt * nt * 8; }
    return 0;
  };
  RtdEval e = make_eval(p, 0, true);
  int rc = launch_windows(p, true, &e, p->have_nt, [&](int w, int64_t c0, int cnt) -> int {
    const int k = w & 1;
    if (w >= 2) {  // the slab of window w - 2 must have been drained before it is overwritten
      int r = drain(w - 2);
      if (r) return r;
    }
    HIP_TRY(hipEventRecord(p->ev_win[k], p->stream));
    HIP_TRY(hipStreamWaitEvent(p->copy_stream, p->ev_win[k], 0));
    char* dst = p->stage[k];
    hipStream_t cs = p->copy_stream;
    if (per_u) { HIP_TRY(hipMemcpyAsync(dst, p->ev_u + c0 * per_u, (size_t)(cnt * per_u) * 8, hipMemcpyDeviceToHost, cs)); dst += (int64_t)cnt * per_u * 8; }
    if (per_u0) { HIP_TRY(hipMemcpyAsync(dst, p->ev_u0 + c0 * per_u0, (size_t)(cnt * per_u0) * 8, hipMemcpyDeviceToHost, cs)); dst += (int64_t)cnt * per_u0 * 8; }
    for (int f = 0; f < 3; ++f)
      if (fls[f]) { HIP_TRY(hipMemcpyAsync(dst, p->ev_fl + f * C * nt + c0 * nt, (size_t)((int64_t)cnt * nt) * 8, hipMemcpyDeviceToHost, cs)); dst += (int64_t)cnt * nt * 8; }
    HIP_TRY(hipEventRecord(p->ev_copied[k], cs));
    return 0;
  });
  if (rc) return rc;
  for (int w = std::max(0, p->nwin - 2); w < p->nwin; ++w)
    if ((rc = drain(w))) return rc;
  return check_status(p, u != nullptr);
}

int rtd_plan_result_dev_ptrs(rtd_plan* p, void** u_dev, int64_t* u_bytes, void** flux_dev, int64_t* flux_bytes) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  const int64_t C = p->d.C, Qr = 2 * p->d.N, nt = p->ev_ntau, np = p->ev_nphi;
  if (u_dev) *u_dev = p->ev_u;
  if (u_bytes) *u_bytes = C * Qr * nt * np * 8;
  if (flux_dev) *flux_dev = p->ev_fl;
  if (flux_bytes) *flux_bytes = 3 * C * nt * 8;
  return 0;
}

int rtd_plan_evaluate(rtd_plan* p, int32_t ntau, const double* tau, int32_t nphi, const double* phi,
                      int32_t antiderivative, double* u, double* u0, double* flux_up, double* fdn, double* fdir,
                      double* ulast) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  if (!p->solved) return fail(RTD_ERR_STATE, "evaluate before solve");
  int rc = rtd_plan_set_eval_points(p, ntau, tau, nphi, phi);
  if (rc) return rc;
  const bool skip_nt = (antiderivative & 2) != 0;
  RtdEval e = make_eval(p, antiderivative & 1, u != nullptr);
  // one window, or a retained plan (rtd_plan_create_retained): what the evaluators need of the solve is resident for every
  // column, only the evaluation kernels run.  Several windows without retention: they are solved again, window by window,
  // with the evaluation behind each (the throughput form rtd_plan_run is the intended entry point for such batches).
  const bool solve_again = p->nwin > 1 && !p->retained && !p->lean;
  rc = launch_windows(p, solve_again, &e, p->have_nt && !skip_nt, [](int, int64_t, int) { return 0; }, false);
  if (rc) return rc;
  if (p->pipelined && !solve_again) p->fork_needed = true;  // the next solve's eigen stream must not overwrite what this pass still reads
  if (ulast) {  // queued before the fetch, whose status check drains the stream: a numerical failure of SOME columns must not
    //             leave the healthy columns' ulast unwritten (numeric_errors = "nan" keeps them)
    const int64_t C = p->d.C, Qr = 2 * p->d.N;
    hipError_t he = hipMemcpyAsync(ulast, p->ev_u0 + C * Qr * ntau, (size_t)(C * Qr * ntau) * 8, hipMemcpyDeviceToHost, p->stream);
    if (he != hipSuccess) return drained(p, fail(RTD_ERR_HIP, std::string("hipMemcpyAsync(ulast): ") + hipGetErrorString(he)));
  }
  return drained(p, rtd_plan_fetch(p, u, u0, flux_up, fdn, fdir));  // (also behind fetch's own early returns: the ulast copy is pending)
}

int rtd_plan_set_nt(rtd_plan* p, int32_t nleg_all, const double* weighted_leg_all, const double* f_arr,
                    const double* ims_coef, const double* ims_par) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  if (nleg_all <= 0) {  // switch the corrections off
    p->have_nt = false;
    return 0;
  }
  if (!weighted_leg_all || !f_arr || !ims_coef || !ims_par) return fail(RTD_ERR_ARG, "null argument");
  if (!p->d.beam) return fail(RTD_ERR_ARG, "NT corrections need a beam source");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = p->d.C, L = p->d.L;
  int rc;  // buffers are reused (or grown) on repeated calls, not leaked
  if ((rc = grow(p, &p->nt_w, &p->cap_nt_w, C * L * nleg_all)) || (rc = grow(p, &p->nt_f, &p->cap_nt_f, C * L)) ||
      (rc = grow(p, &p->nt_ic, &p->cap_nt_ic, C * nleg_all)) || (rc = grow(p, &p->nt_ip, &p->cap_nt_ip, C * 2)) ||
      (rc = grow(p, &p->nt_R, &p->cap_nt_R, C * 4 * p->d.NP * L)))
    return rc;
  hipStream_t s = p->stream;
  HIP_TRY(hipMemcpyAsync(p->nt_w, weighted_leg_all, (size_t)(C * L * nleg_all) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(p->nt_f, f_arr, (size_t)(C * L) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(p->nt_ic, ims_coef, (size_t)(C * nleg_all) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(p->nt_ip, ims_par, (size_t)(C * 2) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  p->nt.nleg_all = nleg_all;
  p->nt.wfull = p->nt_w; p->nt.f = p->nt_f; p->nt.ims_coef = p->nt_ic; p->nt.ims_par = p->nt_ip; p->nt.R = p->nt_R;
  p->have_nt = true;
  p->nt_tables_ready = false;
  return 0;
}

int rtd_plan_get_tensors(rtd_plan* p, int32_t column, double* GC, double* K, double* B, double* Gim, double* G) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  p->fork_needed = true;  // works on the hand-off buffers from the plan's own stream
  if (!p->solved) return fail(RTD_ERR_STATE, "get_tensors before solve");
  if (column < 0 || column >= p->d.C) return fail(RTD_ERR_ARG, "column out of range");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t M = p->d.M, L = p->d.L, Qr = 2 * p->d.N;
  const int64_t nG = M * L * Qr * Qr, nK = M * L * Qr, nZ = L * Qr;
  if (!p->ex_buf) {
    int rc = p->alloc(&p->ex_buf, 2 * nG + 2 * nK + nZ);
    if (rc) return rc;
  }
  hipStream_t s = p->stream;
  RtdDev d = p->d;
  int local = column;
  if (p->nwin > 1 && p->retained) {  // the column's own place in the retained arrays
    d = window_dev(p, column, 1);
    local = 0;
  } else if (p->nwin > 1 && p->lean) {  // lean: the column's Y, A again (a wavefront of the eigen stage never spans columns from
    //                                      8 streams up; below, one column on its own is what a one-column plan computes), coefficients kept
    d = window_dev(p, column, 1);
    local = 0;
    if (!p->tables_valid) {
      rtd_launch_tables(d, s, !p->quad_tables_valid);
      p->quad_tables_valid = true;
    }
    for (int part = 0; part < 3; ++part) rtd_launch_eig(d, s, part);
  } else if (p->nwin > 1) {  // the column is solved again on its own (its window's intermediates may have been overwritten)
    d = window_dev(p, column, 1);
    local = 0;
    if (!p->tables_cached || !p->tables_valid) rtd_launch_tables(d, s, false);
    for (int part = 0; part < 3; ++part) rtd_launch_eig(d, s, part);
    for (int part = 0; part < 2; ++part) rtd_launch_bc(d, s, part);
  }
  double *dGC = p->ex_buf, *dG = dGC + nG, *dK = dG + nG, *dB = dK + nK, *dZ = dB + nK;
  HIP_TRY(hipMemsetAsync(dZ, 0, (size_t)nZ * 8, s));
  rtd_launch_export(d, local, dGC, dK, dB, dZ, dG, s);
  HIP_TRY(hipStreamSynchronize(s));
  if (GC) HIP_TRY(hipMemcpy(GC, dGC, (size_t)nG * 8, hipMemcpyDeviceToHost));
  if (G) HIP_TRY(hipMemcpy(G, dG, (size_t)nG * 8, hipMemcpyDeviceToHost));
  if (K) HIP_TRY(hipMemcpy(K, dK, (size_t)nK * 8, hipMemcpyDeviceToHost));
  if (B) HIP_TRY(hipMemcpy(B, dB, (size_t)nK * 8, hipMemcpyDeviceToHost));
  if (Gim) HIP_TRY(hipMemcpy(Gim, dZ, (size_t)nZ * 8, hipMemcpyDeviceToHost));
  return check_status(p);
}

static int solve_once(const rtd_dims* dims, int32_t device, const rtd_inputs* in, rtd_plan** out) {
  if (!dims || !in || !out) return fail(RTD_ERR_ARG, "null argument");
  rtd_plan* p = nullptr;
  int rc = rtd_plan_create(dims, device, &p);
  if (rc) return rc;
  rc = rtd_plan_set_quadrature(p, in->mu_pos, in->weights);
  if (!rc)
    rc = rtd_plan_set_columns(p, in->scaled_omega, in->tau, in->scaled_tau_with_0, in->scale_tau, in->wleg, in->mu0,
                              in->I0, in->phi0, in->rescale, in->b_pos, in->b_neg, in->s_poly, in->bdrf_q, in->bdrf_q0);
  if (!rc) rc = rtd_plan_solve(p);
  if (rc) {
    const std::string keep = g_err;
    rtd_plan_destroy(p);
    g_err = keep;
    return rc;
  }
  *out = p;
  return 0;
}

int rtd_solve_batch(const rtd_dims* dims, int32_t device, const rtd_inputs* in, int32_t ntau, const double* tau,
                    int32_t nphi, const double* phi, double* u, double* u0, double* flux_up, double* fdn, double* fdir) {
  rtd_plan* p = nullptr;
  int rc = solve_once(dims, device, in, &p);
  if (rc) return rc;
  rc = rtd_plan_evaluate(p, ntau, tau, nphi, phi, 0, u, u0, flux_up, fdn, fdir, nullptr);
  const std::string keep = g_err;
  rtd_plan_destroy(p);
  g_err = keep;
  return rc;
}

int rtd_solve_tensors(const rtd_dims* dims, int32_t device, const rtd_inputs* in, int32_t column, double* GC, double* K,
                      double* B, double* Gim, double* G) {
  rtd_plan* p = nullptr;
  int rc = solve_once(dims, device, in, &p);
  if (rc) return rc;
  rc = rtd_plan_get_tensors(p, column, GC, K, B, Gim, G);
  const std::string keep = g_err;
  rtd_plan_destroy(p);
  g_err = keep;
  return rc;
}

int rtd_plan_enable_timing(rtd_plan* p, int32_t enable) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  if (enable && !p->evt[0]) {
    HIP_TRY(hipSetDevice(p->device));
    for (auto& e : p->evt) HIP_TRY(hipEventCreate(&e));
  }
  if (!enable && p->timing) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    harvest(p);
  }
  p->timing = enable != 0;
  return 0;
}

int rtd_plan_get_timing(rtd_plan* p, double ms[7], int64_t nlaunch[7], int32_t reset) {
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  HIP_TRY(hipStreamSynchronize(p->stream));
  harvest(p);
  for (int k = 0; k < rtd_plan::NT; ++k) {
    if (ms) ms[k] = p->ms[k];
    if (nlaunch) nlaunch[k] = p->nlaunch[k];
    if (reset) {
      p->ms[k] = 0;
      p->nlaunch[k] = 0;
    }
  }
  return 0;
}

int rtd_plan_pivoted_chains(rtd_plan* p, int32_t* chains) {
  if (!p || !chains) return fail(RTD_ERR_ARG, "null argument");
  *chains = 0;
  if (p->d.NP != 32 && !getenv("RTD_BC_TILED")) return 0;
  const size_t n = (size_t)p->Cw * p->d.M;
  std::vector<int> h(n);
  HIP_TRY(hipMemcpyAsync(h.data(), p->d.need_split, n * sizeof(int), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  int k = 0;
  for (size_t i = 0; i < n; ++i)
    if (h[i] != 0) {
      ++k;
      if (getenv("RTD_DEBUG")) fprintf(stderr, "[rtd] pivoted chain: column %zu mode %zu\n", i / p->d.M, i % p->d.M);
    }
  *chains = k;
  return 0;
}

int rtd_plan_max_sweeps(rtd_plan* p, int32_t* sweeps) {
  if (!p || !sweeps) return fail(RTD_ERR_ARG, "null argument");
  int v = 0;
  HIP_TRY(hipMemcpyAsync(&v, p->d.sweeps, sizeof(int), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *sweeps = v;
  return 0;
}

/* ---- RCCL over xGMI: one communicator rank per plan/GPU ------------------ */
namespace {
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  const char* (*StubTransport)(const ncclComm_t) = nullptr;  // only the tests' stand-in transport exports it
  bool stub = false;
};
RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;  // plans may be used from different host threads (one thread per plan)
  std::call_once(once, [] {
    // The ROCm install's RCCL first, by absolute path: a process that also imports PyTorch carries a second,
    // bundled RCCL/HIP pair under the same SONAMEs, and RCCL must bind to the HIP runtime librtd itself uses.
    // RTD_RCCL_STUB (tests only; never set by the library or its Python package): the path of a stand-in for these entry
    // points, for rank processes that share ONE GPU -- RCCL refuses a second rank on a device, so this is the only way the
    // rank > 0 code of rtd_comm_* can execute on a one-GPU box (tests/stub/rccl_stub.cpp).  When it is set nothing else is
    // tried: a stub that fails to load is an error, not a silent switch to RCCL; rtd_comm_transport() reports which one runs.
    const char* stub = getenv("RTD_RCCL_STUB");
    if (stub && *stub) {
      api.h = dlopen(stub, RTLD_NOW | RTLD_LOCAL);
      api.stub = true;
      if (!api.h) fprintf(stderr, "[rtd] RTD_RCCL_STUB=%s: %s\n", stub, dlerror());
    } else {
      for (const char* name : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"}) {
        api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.h) break;
      }
    }
    if (api.h) {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
      api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
      api.AllReduce = (decltype(api.AllReduce))dlsym(api.h, "ncclAllReduce");
      api.Send = (decltype(api.Send))dlsym(api.h, "ncclSend");
      api.Recv = (decltype(api.Recv))dlsym(api.h, "ncclRecv");
      api.GroupStart = (decltype(api.GroupStart))dlsym(api.h, "ncclGroupStart");
      api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.h, "ncclGroupEnd");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
      api.CommCount = (decltype(api.CommCount))dlsym(api.h, "ncclCommCount");
      api.CommUserRank = (decltype(api.CommUserRank))dlsym(api.h, "ncclCommUserRank");
      api.CommCuDevice = (decltype(api.CommCuDevice))dlsym(api.h, "ncclCommCuDevice");
      if (api.stub) api.StubTransport = (decltype(api.StubTransport))dlsym(api.h, "rcclStubTransport");
      if (api.stub && !api.StubTransport) api.h = nullptr;  // RTD_RCCL_STUB must name the stub, not some other RCCL
      if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy || !api.GetErrorString) api.h = nullptr;
    }
  });
  return api.h ? &api : nullptr;
}
}  // namespace

int rtd_comm_preload(void) {
  return rccl() ? 0 : fail(RTD_ERR_HIP, "librccl.so could not be loaded");
}

int rtd_comm_unique_id(char id[128]) {
  RcclApi* r = rccl();
  if (!r) return fail(RTD_ERR_HIP, "librccl.so could not be loaded");
  ncclUniqueId u;
  ncclResult_t rc = r->GetUniqueId(&u);
  if (rc != ncclSuccess) return fail(RTD_ERR_HIP, std::string("ncclGetUniqueId: ") + r->GetErrorString(rc));
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int rtd_comm_init(rtd_plan* p, const char id[128], int32_t rank, int32_t nranks) {
  if (!p || !id || rank < 0 || rank >= nranks) return fail(RTD_ERR_ARG, "bad communicator arguments");
  RcclApi* r = rccl();
  if (!r) return fail(RTD_ERR_HIP, "librccl.so could not be loaded");
  HIP_TRY(hipSetDevice(p->device));
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t rc = r->CommInitRank(&p->comm, nranks, u, rank);
  if (rc != ncclSuccess) {
    p->comm = nullptr;
    return fail(RTD_ERR_HIP, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
  }
  p->comm_rank = rank;
  p->comm_size = nranks;
  return 0;
}

// What RCCL itself says about the plan's communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) -- not the
// arguments rtd_comm_init was called with.  A caller that reports "N ranks" reports this.
int rtd_comm_size(rtd_plan* p, int32_t* nranks, int32_t* rank, int32_t* device) {
  if (!p || !p->comm) return fail(RTD_ERR_STATE, "communicator not initialised");
  RcclApi* r = rccl();
  if (!r || !r->CommCount || !r->CommUserRank) return fail(RTD_ERR_HIP, "librccl.so lacks ncclCommCount / ncclCommUserRank");
  int n = -1, me = -1, dev = -1;
  ncclResult_t rc = r->CommCount(p->comm, &n);
  if (rc == ncclSuccess) rc = r->CommUserRank(p->comm, &me);
  if (rc == ncclSuccess && r->CommCuDevice) rc = r->CommCuDevice(p->comm, &dev);
  if (rc != ncclSuccess) return fail(RTD_ERR_HIP, std::string("ncclCommCount / ncclCommUserRank: ") + r->GetErrorString(rc));
  if (n != p->comm_size || me != p->comm_rank)
    return fail(RTD_ERR_STATE, "RCCL reports rank " + std::to_string(me) + " of " + std::to_string(n) + ", the plan was initialised as rank " +
                                   std::to_string(p->comm_rank) + " of " + std::to_string(p->comm_size));
  if (nranks) *nranks = n;
  if (rank) *rank = me;
  if (device) *device = dev;
  return 0;
}

// Which transport carries the collectives: "rccl" (librccl.so of the ROCm install), or "stub:ipc" / "stub:shm" when the
// environment named the tests' stand-in (RTD_RCCL_STUB).  A caller that reports a multi-rank rate must report this beside it.
int rtd_comm_transport(rtd_plan* p, char* buf, int32_t nbuf) {
  if (!buf || nbuf < 2) return fail(RTD_ERR_ARG, "transport: no buffer");
  RcclApi* r = rccl();
  if (!r) return fail(RTD_ERR_HIP, "librccl.so could not be loaded");
  const char* name = !r->stub ? "rccl" : r->StubTransport(p ? p->comm : nullptr);
  std::snprintf(buf, (size_t)nbuf, "%s", name);
  return 0;
}

int rtd_comm_allgather_fluxes(rtd_plan* p) {
  if (!p || !p->comm) return fail(RTD_ERR_STATE, "communicator not initialised");
  if (p->ev_ntau < 1) return fail(RTD_ERR_STATE, "no evaluation results to gather");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t n = 3 * (int64_t)p->d.C * p->ev_ntau;
  int rc = grow(p, &p->gathered, &p->cap_gathered, n * p->comm_size);
  if (rc) return rc;
  ncclResult_t nr = rccl()->AllGather(p->ev_fl, p->gathered, (size_t)n, ncclDouble, p->comm, p->stream);
  if (nr != ncclSuccess) return fail(RTD_ERR_HIP, std::string("ncclAllGather: ") + rccl()->GetErrorString(nr));
  return 0;
}

int rtd_comm_allgather_results(rtd_plan* p) {
  if (!p || !p->comm) return fail(RTD_ERR_STATE, "communicator not initialised");
  if (p->ev_ntau < 1 || !p->solved) return fail(RTD_ERR_STATE, "no evaluation results to gather");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = p->d.C, Qr = 2 * p->d.N, nt = p->ev_ntau, np = p->ev_nphi;
  const int64_t nu = np > 0 ? C * Qr * nt * np : 0, nfl = 3 * C * nt;
  int rc;
  if ((rc = grow(p, &p->gathered_u, &p->cap_gathered_u, std::max<int64_t>(nu, 1) * p->comm_size))) return rc;
  if ((rc = grow(p, &p->gathered_fl, &p->cap_gathered_fl, nfl * p->comm_size))) return rc;
  if (!p->comm_stream) HIP_TRY(hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking));
  if (!p->ev_results) HIP_TRY(hipEventCreateWithFlags(&p->ev_results, hipEventDisableTiming));
  if (!p->ev_gathered) HIP_TRY(hipEventCreateWithFlags(&p->ev_gathered, hipEventDisableTiming));
  // The rank's own results are first copied (device to device, on the plan's stream, behind the run that produced them) into
  // this rank's slot of the gathered arrays, and the all-gather runs IN PLACE from there on the communication stream: the
  // result buffers are free at once, so the next run's evaluation kernels never wait for the collective (with windows of
  // 256 columns only ~1 ms of the next run precedes its first evaluation kernel: waiting there would expose most of the
  // gather).  The copy of the NEXT gather waits for this gather (one whole step later).
  if (p->gather_inflight) HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_gathered, 0));
  double* const my_u = p->gathered_u + (int64_t)p->comm_rank * nu;
  double* const my_fl = p->gathered_fl + (int64_t)p->comm_rank * nfl;
  if (nu > 0) HIP_TRY(hipMemcpyAsync(my_u, p->ev_u, (size_t)nu * 8, hipMemcpyDeviceToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(my_fl, p->ev_fl, (size_t)nfl * 8, hipMemcpyDeviceToDevice, p->stream));
  HIP_TRY(hipEventRecord(p->ev_results, p->stream));
  HIP_TRY(hipStreamWaitEvent(p->comm_stream, p->ev_results, 0));
  RcclApi* r = rccl();
  ncclResult_t nr = ncclSuccess;
  if (nu > 0) nr = r->AllGather(my_u, p->gathered_u, (size_t)nu, ncclDouble, p->comm, p->comm_stream);
  if (nr == ncclSuccess) nr = r->AllGather(my_fl, p->gathered_fl, (size_t)nfl, ncclDouble, p->comm, p->comm_stream);
  if (nr != ncclSuccess) return fail(RTD_ERR_HIP, std::string("ncclAllGather: ") + r->GetErrorString(nr));
  HIP_TRY(hipEventRecord(p->ev_gathered, p->comm_stream));
  p->gather_inflight = true;
  p->gathered_here = true;
  return 0;
}

int rtd_comm_gather_results(rtd_plan* p, int32_t root) {
  if (!p || !p->comm) return fail(RTD_ERR_STATE, "communicator not initialised");
  if (p->ev_ntau < 1 || !p->solved) return fail(RTD_ERR_STATE, "no evaluation results to gather");
  if (root < 0 || root >= p->comm_size) return fail(RTD_ERR_ARG, "root outside the communicator");
  RcclApi* r = rccl();
  if (!r->Send || !r->Recv || !r->GroupStart || !r->GroupEnd) return fail(RTD_ERR_HIP, "this RCCL has no ncclSend / ncclRecv");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = p->d.C, Qr = 2 * p->d.N, nt = p->ev_ntau, np = p->ev_nphi;
  const int64_t nu = np > 0 ? C * Qr * nt * np : 0, nfl = 3 * C * nt;
  const bool is_root = p->comm_rank == root;
  int rc;
  // the root holds the gathered arrays; the other ranks one shard of each, as the send buffer (a snapshot of their results)
  const int64_t slots = is_root ? p->comm_size : 1;
  if ((rc = grow(p, &p->gathered_u, &p->cap_gathered_u, std::max<int64_t>(nu, 1) * slots))) return rc;
  if ((rc = grow(p, &p->gathered_fl, &p->cap_gathered_fl, nfl * slots))) return rc;
  if (!p->comm_stream) HIP_TRY(hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking));
  if (!p->ev_results) HIP_TRY(hipEventCreateWithFlags(&p->ev_results, hipEventDisableTiming));
  if (!p->ev_gathered) HIP_TRY(hipEventCreateWithFlags(&p->ev_gathered, hipEventDisableTiming));
  // as rtd_comm_allgather_results: the rank's results are snapshot on the plan's stream (the root: into its own slot), the
  // transfers read the snapshot, the next run's evaluation kernels do not wait for them
  if (p->gather_inflight) HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_gathered, 0));
  double* const my_u = p->gathered_u + (is_root ? (int64_t)root * nu : 0);
  double* const my_fl = p->gathered_fl + (is_root ? (int64_t)root * nfl : 0);
  if (nu > 0) HIP_TRY(hipMemcpyAsync(my_u, p->ev_u, (size_t)nu * 8, hipMemcpyDeviceToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(my_fl, p->ev_fl, (size_t)nfl * 8, hipMemcpyDeviceToDevice, p->stream));
  HIP_TRY(hipEventRecord(p->ev_results, p->stream));
  HIP_TRY(hipStreamWaitEvent(p->comm_stream, p->ev_results, 0));
  hipStream_t cs = p->comm_stream;
  ncclResult_t nr = r->GroupStart();
  if (is_root) {
    for (int q = 0; q < p->comm_size && nr == ncclSuccess; ++q) {
      if (q == root) continue;
      if (nu > 0) nr = r->Recv(p->gathered_u + q * nu, (size_t)nu, ncclDouble, q, p->comm, cs);
      if (nr == ncclSuccess) nr = r->Recv(p->gathered_fl + q * nfl, (size_t)nfl, ncclDouble, q, p->comm, cs);
    }
  } else {
    if (nu > 0) nr = r->Send(my_u, (size_t)nu, ncclDouble, root, p->comm, cs);
    if (nr == ncclSuccess) nr = r->Send(my_fl, (size_t)nfl, ncclDouble, root, p->comm, cs);
  }
  ncclResult_t ne = r->GroupEnd();
  if (nr == ncclSuccess) nr = ne;
  if (nr != ncclSuccess) return fail(RTD_ERR_HIP, std::string("ncclSend / ncclRecv: ") + r->GetErrorString(nr));
  HIP_TRY(hipEventRecord(p->ev_gathered, cs));
  p->gather_inflight = true;
  p->gathered_here = is_root;  // (a non-root rank may still hold full-size buffers of an earlier all-gather: they are stale now)
  return 0;
}

int rtd_comm_fetch_gathered_results(rtd_plan* p, double* u, double* fluxes) {
  if (!p || !p->gathered_fl) return fail(RTD_ERR_STATE, "nothing gathered");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = p->d.C, Qr = 2 * p->d.N, nt = p->ev_ntau, np = p->ev_nphi;
  const int64_t nu = np > 0 ? C * Qr * nt * np : 0, nfl = 3 * C * nt;
  if (!p->gathered_here || p->cap_gathered_fl < nfl * p->comm_size || (u && nu > 0 && p->cap_gathered_u < nu * p->comm_size))
    return fail(RTD_ERR_STATE, "this rank holds no gathered arrays of the last collective (rtd_comm_gather_results: only the root does)");
  hipStream_t s = p->comm_stream;
  if (u && nu > 0) HIP_TRY(hipMemcpyAsync(u, p->gathered_u, (size_t)(nu * p->comm_size) * 8, hipMemcpyDeviceToHost, s));
  if (fluxes) HIP_TRY(hipMemcpyAsync(fluxes, p->gathered_fl, (size_t)(nfl * p->comm_size) * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int rtd_comm_fetch_gathered_columns(rtd_plan* p, int32_t rank, int32_t first, int32_t count, double* u, double* fluxes) {
  if (!p || !p->gathered_fl) return fail(RTD_ERR_STATE, "nothing gathered");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = p->d.C, Qr = 2 * p->d.N, nt = p->ev_ntau, np = p->ev_nphi;
  const int64_t per_u = np > 0 ? Qr * nt * np : 0, nu = C * per_u, nfl = 3 * C * nt;
  if (rank < 0 || rank >= p->comm_size || first < 0 || count < 1 || (int64_t)first + count > C)
    return fail(RTD_ERR_ARG, "gathered columns: rank or column range outside the gathered arrays");
  if (!p->gathered_here || p->cap_gathered_fl < nfl * p->comm_size || (u && nu > 0 && p->cap_gathered_u < nu * p->comm_size))
    return fail(RTD_ERR_STATE, "this rank holds no gathered arrays of the last collective (rtd_comm_gather_results: only the root does)");
  hipStream_t s = p->comm_stream;  // behind the collective
  if (u && per_u > 0)
    HIP_TRY(hipMemcpyAsync(u, p->gathered_u + rank * nu + first * per_u, (size_t)(count * per_u) * 8, hipMemcpyDeviceToHost, s));
  if (fluxes)
    for (int f = 0; f < 3; ++f)
      HIP_TRY(hipMemcpyAsync(fluxes + (int64_t)f * count * nt, p->gathered_fl + rank * nfl + f * C * nt + (int64_t)first * nt,
                             (size_t)(count * nt) * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

namespace {
// [CM][L][E] <-> layer-major staging [L'][CM][E] for the layers [l0, l0 + ln): to_stage packs, else unpacks
__global__ void rtd_layer_stage_kernel(double* arr, double* stage, long CM, int L, int E, int l0, int ln, int to_stage) {
  const long total = (long)ln * CM * E;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int e = (int)(idx % E);
    const long cm = (idx / E) % CM;
    const int ll = (int)(idx / ((long)E * CM));
    double* a = arr + (cm * L + l0 + ll) * E + e;
    if (to_stage) stage[idx] = *a;
    else *a = stage[idx];
  }
}
struct LayerSeg { double* arr; long CM; int E; };
int layer_segments(rtd_plan* p, LayerSeg seg[8]) {
  const RtdDev& d = p->d;
  const long CM = (long)d.C * d.M;
  const int NP = d.NP, Q2 = 2 * d.NP;
  int n = 0;
  seg[n++] = {d.Ym, CM, NP * NP};
  seg[n++] = {d.Am, CM, NP * NP};
  seg[n++] = {d.kk, CM, NP};
  seg[n++] = {d.Ek, CM, NP};
  seg[n++] = {d.Bv, CM, Q2};
  if (d.Ns > 0) {
    seg[n++] = {d.dq, (long)d.C, d.Ns * Q2};
    seg[n++] = {d.zneg, (long)d.C, NP};
    seg[n++] = {d.vb, (long)d.C, 4 * NP};
  }
  return n;
}
}  // namespace

int rtd_plan_solve_layers(rtd_plan* p, int32_t first, int32_t count) {
  (void)hipGetLastError();  // (as launch_windows: a stale code of this thread is not this call's)
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  p->fork_needed = true;  // works on the hand-off buffers from the plan's own stream
  if (!p->have_quad || !p->have_cols) return fail(RTD_ERR_STATE, "set_quadrature and set_columns must precede solve");
  if (p->nwin != 1) return fail(RTD_ERR_STATE, "layer shards need a plan of one window");
  if (first < 0 || count < 1 || first + count > p->d.L) return fail(RTD_ERR_ARG, "layer range outside the atmosphere");
  HIP_TRY(hipSetDevice(p->device));
  RtdDev d = p->d;
  d.l0 = first;
  d.ln = count;
  d.um = nullptr;
  HIP_TRY(hipMemsetAsync(d.sweeps, 0, sizeof(int), p->stream));
  HIP_TRY(hipMemsetAsync(d.status, 0, sizeof(int), p->stream));  // as launch_windows: a new solve starts clean
  HIP_TRY(hipMemsetAsync(d.col_status, 0, sizeof(int) * (size_t)d.C, p->stream));
  p->numeric_status = 0;
  rtd_launch_tables(d, p->stream, !p->quad_tables_valid);
  p->quad_tables_valid = true;
  p->tables_valid = true;  // (one-window plan: d.Y0 / d.att are the all-columns tables)
  rtd_launch_eig(d, p->stream, 1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RTD_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  p->solved = false;
  return 0;
}

int rtd_comm_allgather_layers(rtd_plan* p, int32_t count) {
  if (!p || !p->comm) return fail(RTD_ERR_STATE, "communicator not initialised");
  if (p->nwin != 1) return fail(RTD_ERR_STATE, "layer shards need a plan of one window");
  if (count < 1 || (int64_t)count * p->comm_size != p->d.L)
    return fail(RTD_ERR_ARG, "layer shards: nlayers must equal count_per_rank x nranks");
  HIP_TRY(hipSetDevice(p->device));
  LayerSeg seg[8];
  const int nseg = layer_segments(p, seg);
  int64_t per_rank = 0;
  for (int k = 0; k < nseg; ++k) per_rank += (int64_t)count * seg[k].CM * seg[k].E;
  int rc;
  // staging: this rank's packed layers, then everybody's (reuses the gathered-u buffer slot of the plan)
  double* mine = nullptr;
  HIP_TRY(hipMalloc(&mine, (size_t)per_rank * 8));
  p->gathered_here = false;  // the gathered-u buffer is layer staging from here on: no results fetch may read it as gathered u
  if ((rc = grow(p, &p->gathered_u, &p->cap_gathered_u, per_rank * p->comm_size))) {
    (void)hipFree(mine);
    return rc;
  }
  hipStream_t s = p->stream;
  const int L = p->d.L, l0 = p->comm_rank * count;
  int64_t off = 0;
  for (int k = 0; k < nseg; ++k) {
    hipLaunchKernelGGL(rtd_layer_stage_kernel, dim3(1024), dim3(256), 0, s, seg[k].arr, mine + off, seg[k].CM, L, seg[k].E, l0, count, 1);
    off += (int64_t)count * seg[k].CM * seg[k].E;
  }
  ncclResult_t nr = rccl()->AllGather(mine, p->gathered_u, (size_t)per_rank, ncclDouble, p->comm, s);
  if (nr == ncclSuccess) {
    for (int g = 0; g < p->comm_size; ++g) {
      if (g == p->comm_rank) continue;  // this rank's own layers are in place
      off = 0;
      for (int k = 0; k < nseg; ++k) {
        hipLaunchKernelGGL(rtd_layer_stage_kernel, dim3(1024), dim3(256), 0, s, seg[k].arr, p->gathered_u + g * per_rank + off,
                           seg[k].CM, L, seg[k].E, g * count, count, 0);
        off += (int64_t)count * seg[k].CM * seg[k].E;
      }
    }
  }
  hipError_t e = hipStreamSynchronize(s);
  (void)hipFree(mine);
  if (nr != ncclSuccess) return fail(RTD_ERR_HIP, std::string("ncclAllGather: ") + rccl()->GetErrorString(nr));
  if (e != hipSuccess) return fail(RTD_ERR_HIP, hipGetErrorString(e));
  return 0;
}

int rtd_plan_solve_bc(rtd_plan* p) {
  (void)hipGetLastError();  // (as launch_windows: a stale code of this thread is not this call's)
  if (!p) return fail(RTD_ERR_ARG, "null plan");
  if (!p->have_quad || !p->have_cols) return fail(RTD_ERR_STATE, "inputs missing");
  if (p->nwin != 1) return fail(RTD_ERR_STATE, "layer shards need a plan of one window");
  HIP_TRY(hipSetDevice(p->device));
  RtdDev d = p->d;
  d.um = nullptr;
  rtd_launch_bc(d, p->stream, 0);
  rtd_launch_bc(d, p->stream, 1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RTD_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  p->solved = true;
  return 0;
}

int rtd_comm_allreduce_results(rtd_plan* p) {
  if (!p || !p->comm) return fail(RTD_ERR_STATE, "communicator not initialised");
  if (p->ev_ntau < 1) return fail(RTD_ERR_STATE, "no evaluation results to reduce");
  RcclApi* r = rccl();
  if (!r || !r->AllReduce) return fail(RTD_ERR_HIP, "ncclAllReduce not available");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t C = p->d.C, Qr = 2 * p->d.N, nt = p->ev_ntau, np = p->ev_nphi;
  struct { double* ptr; int64_t n; } bufs[3] = {{p->ev_u, np > 0 ? C * Qr * nt * np : 0}, {p->ev_u0, C * Qr * nt}, {p->ev_fl, 3 * C * nt}};
  for (auto& b : bufs) {
    if (!b.ptr || b.n <= 0) continue;
    ncclResult_t nr = r->AllReduce(b.ptr, b.ptr, (size_t)b.n, ncclDouble, ncclSum, p->comm, p->stream);
    if (nr != ncclSuccess) return fail(RTD_ERR_HIP, std::string("ncclAllReduce: ") + r->GetErrorString(nr));
  }
  return 0;
}

int rtd_comm_fetch_gathered(rtd_plan* p, double* out) {
  if (!p || !out || !p->gathered) return fail(RTD_ERR_STATE, "nothing gathered");
  HIP_TRY(hipSetDevice(p->device));
  const int64_t n = 3 * (int64_t)p->d.C * p->ev_ntau * p->comm_size;
  HIP_TRY(hipMemcpyAsync(out, p->gathered, (size_t)n * 8, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return 0;
}

int rtd_comm_destroy(rtd_plan* p) {
  if (!p) return 0;
  if (p->comm && rccl()) {
    (void)hipStreamSynchronize(p->stream);
    if (p->comm_stream) (void)hipStreamSynchronize(p->comm_stream);
    rccl()->CommDestroy(p->comm);
  }
  p->comm = nullptr;
  p->gather_inflight = false;
  return 0;
}

}  // extern "C"
