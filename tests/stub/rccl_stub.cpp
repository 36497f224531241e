// TEST INFRASTRUCTURE, NOT PRODUCT.  A stand-in for the thirteen RCCL entry points librtd binds (csrc/rtd_api.hip: RcclApi),
// for rank PROCESSES THAT SHARE ONE GPU: RCCL itself refuses a second rank on a device ("Duplicate GPU detected"), so on a
// 1-GPU box rtd_comm_* with rank > 0 -- slot offsets, the root's receive loop, the mode all-reduce, the layer stitch -- could
// never execute.  librtd loads this library ONLY when RTD_RCCL_STUB names it (rtd_comm_transport() then says "stub"); nothing
// measured through it is a rate.
//
//   rendezvous : ncclGetUniqueId names a POSIX shared-memory control block; ncclCommInitRank maps it and waits for nranks
//   transport  : "ipc" (default) -- peers' device buffers through hipIpcGetMemHandle / hipIpcOpenMemHandle, device-to-device
//                hipMemcpyAsync on the CALLER's stream; "shm" (RCCL_STUB_TRANSPORT=shm, or when the IPC probe of
//                ncclCommInitRank fails on any rank) -- staged through per-rank shared-memory segments
//   semantics  : every collective is host-synchronous (returns when it is complete on every rank) and synchronises ONLY the
//                stream it was given -- an ordering the caller forgot (an event wait before the collective) still shows as
//                wrong data; sums of ncclAllReduce run in rank order on every rank (all ranks get identical bits)
//   liveness   : every wait has a time limit (RCCL_STUB_TIMEOUT_S, default 120): a rank that never arrives ends the others
//                with ncclSystemError instead of hanging the box
//
// Build: hipcc --offload-arch=gfx950 -O2 -fPIC -shared tests/stub/rccl_stub.cpp -o tests/stub/librccl_stub.so -lrt
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
constexpr int MAXR = 16;
constexpr int RING = 16;

struct Slot {
  hipIpcMemHandle_t h;
  uint64_t offset, bytes;
  uint64_t seg_bytes;  // shm transport: the size the rank's data segment has now
};
struct MailEntry {
  hipIpcMemHandle_t h;
  uint64_t offset, bytes;
  uint64_t seg_offset;  // shm transport: where in the sender's segment
};
struct Mail {
  std::atomic<uint64_t> posted, consumed;
  MailEntry e[RING];
};
struct Ctl {
  std::atomic<uint32_t> nranks, joined, arrived, generation, ipc_fail, aborted;
  Slot slot[MAXR];
  Mail mail[MAXR][MAXR];  // [src][dst]
};

struct Seg {
  int fd = -1;
  char* p = nullptr;
  size_t bytes = 0;
};

double now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
double time_limit() {
  const char* s = getenv("RCCL_STUB_TIMEOUT_S");
  const double v = s ? atof(s) : 0.0;
  return v > 0 ? v : 120.0;
}
void nap(int spins) {
  if (spins < 200) sched_yield();
  else usleep(50);
}
}  // namespace

struct ncclComm {
  int rank = 0, n = 0, device = 0;
  bool shm = false;
  std::string name;
  Ctl* ctl = nullptr;
  Seg seg[MAXR];  // shm transport: [me] is mine (read-write), the others are peers' (mapped on demand)
  std::map<std::string, void*> opened;  // IPC handle bytes -> mapped base
  double* staging = nullptr;  // all-reduce: a copy of this rank's input that peers read
  size_t staging_bytes = 0;
  const double** ptrs_dev = nullptr;
  void* probe = nullptr;
  uint64_t sent[MAXR] = {}, recvd[MAXR] = {};
  double limit = 120.0;
};

namespace {
std::mutex g_mutex;
std::atomic<uint32_t> g_ids{0};

bool wait_until(ncclComm* c, const std::function<bool()>& ready) {
  const double t0 = now();
  for (int spins = 0;; ++spins) {
    if (ready()) return true;
    if (c->ctl->aborted.load()) return false;
    if (now() - t0 > c->limit) {
      c->ctl->aborted.store(1);
      fprintf(stderr, "[rccl_stub] rank %d of %d: a peer did not arrive within %.0f s\n", c->rank, c->n, c->limit);
      return false;
    }
    nap(spins);
  }
}

bool barrier(ncclComm* c) {
  Ctl* k = c->ctl;
  const uint32_t gen = k->generation.load();
  if (k->arrived.fetch_add(1) + 1 == (uint32_t)c->n) {
    k->arrived.store(0);
    k->generation.fetch_add(1);
    return !k->aborted.load();
  }
  return wait_until(c, [&] { return k->generation.load() != gen; });
}

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

// ---- "ipc": a device pointer of this process as (handle of its allocation, offset) and back -----------------------------
bool describe(const void* ptr, hipIpcMemHandle_t* h, uint64_t* offset) {
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr) != hipSuccess) return false;
  if (hipIpcGetMemHandle(h, (void*)base) != hipSuccess) return false;
  *offset = (uint64_t)((const char*)ptr - (const char*)base);
  return true;
}
void* peer_pointer(ncclComm* c, const hipIpcMemHandle_t& h, uint64_t offset) {
  std::lock_guard<std::mutex> g(g_mutex);
  const std::string key((const char*)&h, sizeof(h));
  auto it = c->opened.find(key);
  if (it == c->opened.end()) {
    void* base = nullptr;
    if (hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    it = c->opened.emplace(key, base).first;
  }
  return (char*)it->second + offset;
}

// ---- "shm": per-rank data segments -------------------------------------------------------------------------------------------
std::string seg_name(ncclComm* c, int r) { return c->name + "_d" + std::to_string(r); }
bool seg_reserve(ncclComm* c, size_t bytes) {  // my own segment, at least `bytes`
  Seg& s = c->seg[c->rank];
  if (s.bytes >= bytes) return true;
  const size_t want = ((bytes + (1u << 20)) * 5 / 4 + 4095) & ~(size_t)4095;
  if (s.fd < 0) s.fd = shm_open(seg_name(c, c->rank).c_str(), O_CREAT | O_RDWR, 0600);
  if (s.fd < 0 || ftruncate(s.fd, (off_t)want) != 0) return false;
  if (s.p) munmap(s.p, s.bytes);
  s.p = (char*)mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED, s.fd, 0);
  if (s.p == MAP_FAILED) {
    s.p = nullptr;
    s.bytes = 0;
    return false;
  }
  s.bytes = want;
  c->ctl->slot[c->rank].seg_bytes = want;
  return true;
}
const char* seg_peer(ncclComm* c, int r, size_t upto) {  // a peer's segment, mapped to what it has published
  if (r == c->rank) return c->seg[r].p;
  Seg& s = c->seg[r];
  if (s.bytes < upto) {
    const size_t have = c->ctl->slot[r].seg_bytes;
    if (have < upto) return nullptr;
    if (s.fd < 0) s.fd = shm_open(seg_name(c, r).c_str(), O_RDWR, 0600);
    if (s.fd < 0) return nullptr;
    if (s.p) munmap(s.p, s.bytes);
    s.p = (char*)mmap(nullptr, have, PROT_READ, MAP_SHARED, s.fd, 0);
    if (s.p == MAP_FAILED) {
      s.p = nullptr;
      s.bytes = 0;
      return nullptr;
    }
    s.bytes = have;
  }
  return s.p;
}

__global__ void stub_sum_kernel(double* out, const double* const* in, int n, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    double s = in[0][i];
    for (int r = 1; r < n; ++r) s += in[r][i];  // rank order on every rank: identical bits everywhere
    out[i] = s;
  }
}

#define HIPOK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      fprintf(stderr, "[rccl_stub] %s: %s\n", #x, hipGetErrorString(e_));                          \
      return ncclUnhandledCudaError;                                                               \
    }                                                                                              \
  } while (0)

// point-to-point operations queued between ncclGroupStart and ncclGroupEnd (per host thread, like RCCL's groups)
struct P2p {
  bool send;
  void* buf;
  size_t bytes;
  int peer;
  ncclComm* comm;
  hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<P2p> g_queue;

ncclResult_t run_p2p(std::vector<P2p>& ops) {
  // (1) what the caller queued on the streams so far is final
  for (auto& o : ops) HIPOK(hipStreamSynchronize(o.stream));
  // (2) every send is posted before any receive is waited for (a ring of RING entries per ordered pair)
  std::map<ncclComm*, size_t> seg_used;
  for (auto& o : ops) {
    if (!o.send) continue;
    ncclComm* c = o.comm;
    Mail& m = c->ctl->mail[c->rank][o.peer];
    if (!wait_until(c, [&] { return m.posted.load() - m.consumed.load() < (uint64_t)RING; })) return ncclSystemError;
    MailEntry& e = m.e[m.posted.load() % RING];
    e.bytes = o.bytes;
    if (c->shm) {
      size_t& used = seg_used[c];
      if (!seg_reserve(c, used + o.bytes)) return ncclSystemError;
      HIPOK(hipMemcpy(c->seg[c->rank].p + used, o.buf, o.bytes, hipMemcpyDeviceToHost));
      e.seg_offset = used;
      used += (o.bytes + 63) & ~(size_t)63;
    } else if (!describe(o.buf, &e.h, &e.offset)) {
      return ncclUnhandledCudaError;
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);
    m.posted.fetch_add(1);
  }
  // (3) receives, in the order they were queued
  std::vector<std::pair<Mail*, uint64_t>> done;
  for (auto& o : ops) {
    if (o.send) continue;
    ncclComm* c = o.comm;
    Mail& m = c->ctl->mail[o.peer][c->rank];
    const uint64_t seq = c->recvd[o.peer]++;
    if (!wait_until(c, [&] { return m.posted.load() > seq; })) return ncclSystemError;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const MailEntry& e = m.e[seq % RING];
    if (e.bytes != o.bytes) {
      fprintf(stderr, "[rccl_stub] rank %d: ncclRecv of %zu bytes from rank %d meets an ncclSend of %llu bytes\n", c->rank, o.bytes,
              o.peer, (unsigned long long)e.bytes);
      c->ctl->aborted.store(1);
      return ncclInvalidArgument;
    }
    if (c->shm) {
      const char* src = seg_peer(c, o.peer, e.seg_offset + e.bytes);
      if (!src) return ncclSystemError;
      HIPOK(hipMemcpyAsync(o.buf, src + e.seg_offset, o.bytes, hipMemcpyHostToDevice, o.stream));
    } else {
      void* src = peer_pointer(c, e.h, e.offset);
      if (!src) return ncclUnhandledCudaError;
      HIPOK(hipMemcpyAsync(o.buf, src, o.bytes, hipMemcpyDeviceToDevice, o.stream));
    }
    done.emplace_back(&m, seq + 1);
  }
  for (auto& o : ops)
    if (!o.send) HIPOK(hipStreamSynchronize(o.stream));
  for (auto& d : done) d.first->consumed.store(d.second);
  // (4) a send returns when its buffer has been read
  for (auto& o : ops) {
    if (!o.send) continue;
    ncclComm* c = o.comm;
    Mail& m = c->ctl->mail[c->rank][o.peer];
    if (!wait_until(c, [&] { return m.consumed.load() == m.posted.load(); })) return ncclSystemError;
  }
  return ncclSuccess;
}

ncclResult_t p2p(bool send, void* buf, size_t count, ncclDataType_t t, int peer, ncclComm* c, hipStream_t s) {
  if (!c || !buf || peer < 0 || peer >= c->n || peer == c->rank || !type_bytes(t)) return ncclInvalidArgument;
  g_queue.push_back({send, buf, count * type_bytes(t), peer, c, s});
  if (g_depth > 0) return ncclSuccess;
  std::vector<P2p> ops;
  ops.swap(g_queue);
  return run_p2p(ops);
}
}  // namespace

extern "C" {

// what librtd asks to tell the stub from RCCL (rtd_comm_transport)
const char* rcclStubTransport(const ncclComm_t comm) { return comm ? (comm->shm ? "stub:shm" : "stub:ipc") : "stub"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id->internal, 0, sizeof(id->internal));
  timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  snprintf(id->internal, sizeof(id->internal), "/rccl_stub_%d_%u_%lx", (int)getpid(), g_ids.fetch_add(1), (unsigned long)t.tv_nsec);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  if (std::strncmp(id.internal, "/rccl_stub_", 11) != 0 || !std::memchr(id.internal, 0, sizeof(id.internal))) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->n = nranks;
  c->name = id.internal;
  c->limit = time_limit();
  HIPOK(hipGetDevice(&c->device));
  const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Ctl)) != 0) {
    perror("[rccl_stub] control block");
    delete c;
    return ncclSystemError;
  }
  c->ctl = (Ctl*)mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);  // a fresh object is all zeros
  close(fd);
  if (c->ctl == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  Ctl* k = c->ctl;
  k->nranks.store((uint32_t)nranks);
  k->joined.fetch_add(1);
  if (!wait_until(c, [&] { return k->joined.load() >= (uint32_t)nranks; })) return ncclSystemError;
  // which transport: every rank exports a probe buffer, its right-hand neighbour reads it
  const char* forced = getenv("RCCL_STUB_TRANSPORT");
  bool mine_ok = !(forced && std::strcmp(forced, "shm") == 0);
  uint64_t pattern = 0x5a5a000000000000ull + (uint64_t)rank;
  if (mine_ok) {
    mine_ok = hipMalloc(&c->probe, 4096) == hipSuccess && hipMemcpy(c->probe, &pattern, 8, hipMemcpyHostToDevice) == hipSuccess &&
              describe(c->probe, &k->slot[rank].h, &k->slot[rank].offset);
  }
  if (!mine_ok) k->ipc_fail.store(1);
  if (!barrier(c)) return ncclSystemError;
  if (!k->ipc_fail.load() && nranks > 1) {
    const int q = (rank + 1) % nranks;
    uint64_t got = 0;
    void* src = peer_pointer(c, k->slot[q].h, k->slot[q].offset);
    if (!src || hipMemcpy(&got, src, 8, hipMemcpyDeviceToHost) != hipSuccess || got != 0x5a5a000000000000ull + (uint64_t)q) {
      (void)hipGetLastError();
      k->ipc_fail.store(1);
    }
  }
  if (!barrier(c)) return ncclSystemError;
  c->shm = k->ipc_fail.load() != 0;
  if (forced && std::strcmp(forced, "ipc") == 0 && c->shm) {
    fprintf(stderr, "[rccl_stub] rank %d: RCCL_STUB_TRANSPORT=ipc but hipIpc between the rank processes does not work here\n", rank);
    return ncclSystemError;
  }
  if (!barrier(c)) return ncclSystemError;
  if (rank == 0) shm_unlink(c->name.c_str());  // everyone has it mapped: nothing is left behind in /dev/shm
  if (getenv("RCCL_STUB_DEBUG"))
    fprintf(stderr, "[rccl_stub] rank %d of %d on device %d: transport %s\n", rank, nranks, c->device, c->shm ? "shm" : "ipc");
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  {
    std::lock_guard<std::mutex> g(g_mutex);
    for (auto& kv : c->opened) (void)hipIpcCloseMemHandle(kv.second);
    c->opened.clear();
  }
  if (c->staging) (void)hipFree(c->staging);
  if (c->ptrs_dev) (void)hipFree(c->ptrs_dev);
  if (c->probe) (void)hipFree(c->probe);
  for (int r = 0; r < MAXR; ++r) {
    if (c->seg[r].p) munmap(c->seg[r].p, c->seg[r].bytes);
    if (c->seg[r].fd >= 0) close(c->seg[r].fd);
  }
  if (c->seg[c->rank].fd >= 0) shm_unlink(seg_name(c, c->rank).c_str());
  if (c->ctl) munmap(c->ctl, sizeof(Ctl));
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* n) {
  if (!c || !n) return ncclInvalidArgument;
  *n = (int)c->ctl->nranks.load();  // what the ranks agreed on, not this rank's argument
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) {
  if (!c || !r) return ncclInvalidArgument;
  *r = c->rank;
  return ncclSuccess;
}
ncclResult_t ncclCommCuDevice(const ncclComm_t c, int* d) {
  if (!c || !d) return ncclInvalidArgument;
  *d = c->device;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "stub transport: unhandled HIP error";
    case ncclSystemError: return "stub transport: system error (a rank did not arrive, or shared memory failed)";
    case ncclInternalError: return "stub transport: internal error";
    case ncclInvalidArgument: return "stub transport: invalid argument";
    case ncclInvalidUsage: return "stub transport: invalid usage";
    default: return "stub transport: error";
  }
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
  const size_t bytes = count * type_bytes(t);
  if (!c || !send || !recv || !type_bytes(t)) return ncclInvalidArgument;
  if (g_depth > 0) return ncclInvalidUsage;
  Slot& me = c->ctl->slot[c->rank];
  me.bytes = bytes;
  HIPOK(hipStreamSynchronize(s));  // this rank's contribution is final
  if (c->shm) {
    if (!seg_reserve(c, bytes)) return ncclSystemError;
    HIPOK(hipMemcpy(c->seg[c->rank].p, send, bytes, hipMemcpyDeviceToHost));
  } else if (!describe(send, &me.h, &me.offset)) {
    return ncclUnhandledCudaError;
  }
  if (!barrier(c)) return ncclSystemError;
  for (int q = 0; q < c->n; ++q) {
    char* dst = (char*)recv + (size_t)q * bytes;
    const Slot& sl = c->ctl->slot[q];
    if (sl.bytes != bytes) {
      fprintf(stderr, "[rccl_stub] rank %d: ncclAllGather of %zu bytes, rank %d contributes %llu\n", c->rank, bytes, q, (unsigned long long)sl.bytes);
      c->ctl->aborted.store(1);
      return ncclInvalidArgument;
    }
    if (q == c->rank) {
      if (dst != send) HIPOK(hipMemcpyAsync(dst, send, bytes, hipMemcpyDeviceToDevice, s));
    } else if (c->shm) {
      const char* src = seg_peer(c, q, bytes);
      if (!src) return ncclSystemError;
      HIPOK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    } else {
      void* src = peer_pointer(c, sl.h, sl.offset);
      if (!src) return ncclUnhandledCudaError;
      HIPOK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    }
  }
  HIPOK(hipStreamSynchronize(s));
  return barrier(c) ? ncclSuccess : ncclSystemError;  // nobody's contribution changes while a peer still reads it
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
  if (!c || !send || !recv) return ncclInvalidArgument;
  if (t != ncclFloat64 || op != ncclSum) return ncclInvalidArgument;  // what librtd uses
  if (g_depth > 0) return ncclInvalidUsage;
  const size_t bytes = count * 8;
  Slot& me = c->ctl->slot[c->rank];
  me.bytes = bytes;
  if (c->shm) {
    HIPOK(hipStreamSynchronize(s));
    if (!seg_reserve(c, bytes)) return ncclSystemError;
    HIPOK(hipMemcpy(c->seg[c->rank].p, send, bytes, hipMemcpyDeviceToHost));
    if (!barrier(c)) return ncclSystemError;
    std::vector<double> sum(count);
    for (int q = 0; q < c->n; ++q) {
      if (c->ctl->slot[q].bytes != bytes) return ncclInvalidArgument;
      const double* src = (const double*)seg_peer(c, q, bytes);
      if (!src) return ncclSystemError;
      if (q == 0) std::memcpy(sum.data(), src, bytes);
      else
        for (size_t i = 0; i < count; ++i) sum[i] += src[i];
    }
    HIPOK(hipMemcpyAsync(recv, sum.data(), bytes, hipMemcpyHostToDevice, s));
    HIPOK(hipStreamSynchronize(s));
    return barrier(c) ? ncclSuccess : ncclSystemError;
  }
  // ipc: the input is copied to a staging buffer of this rank (the reduction may be in place), peers read the staging buffers
  if (c->staging_bytes < bytes) {
    // (a barrier-separated point: no peer reads the old staging buffer any more)
    if (c->staging) HIPOK(hipFree(c->staging));
    c->staging = nullptr;
    HIPOK(hipMalloc((void**)&c->staging, bytes));
    c->staging_bytes = bytes;
  }
  if (!c->ptrs_dev) HIPOK(hipMalloc((void**)&c->ptrs_dev, MAXR * sizeof(double*)));
  HIPOK(hipMemcpyAsync(c->staging, send, bytes, hipMemcpyDeviceToDevice, s));
  HIPOK(hipStreamSynchronize(s));
  if (!describe(c->staging, &me.h, &me.offset)) return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  const double* ptrs[MAXR];
  for (int q = 0; q < c->n; ++q) {
    const Slot& sl = c->ctl->slot[q];
    if (sl.bytes != bytes) return ncclInvalidArgument;
    ptrs[q] = q == c->rank ? c->staging : (const double*)peer_pointer(c, sl.h, sl.offset);
    if (!ptrs[q]) return ncclUnhandledCudaError;
  }
  HIPOK(hipMemcpyAsync(c->ptrs_dev, ptrs, c->n * sizeof(double*), hipMemcpyHostToDevice, s));
  const int blocks = (int)std::min<size_t>(2048, (count + 255) / 256);
  hipLaunchKernelGGL(stub_sum_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, (double*)recv, c->ptrs_dev, c->n, count);
  HIPOK(hipGetLastError());
  HIPOK(hipStreamSynchronize(s));
  return barrier(c) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  return p2p(true, const_cast<void*>(buf), count, t, peer, c, s);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  return p2p(false, buf, count, t, peer, c, s);
}
ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  std::vector<P2p> ops;
  ops.swap(g_queue);
  return ops.empty() ? ncclSuccess : run_p2p(ops);
}

}  // extern "C"
