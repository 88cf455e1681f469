// chem_comm.hpp -- neighbour transport of the slab domain decomposition.
//
// The reference reaches MPI through boost::mpi inside ESPResSo++ (storage.DomainDecomposition,
// src/start_simulation.py:152-163); here one process drives one GPU and the ghost layers travel as
// contiguous device slices over RCCL point-to-point (xGMI).  RCCL is opened with dlopen at
// chem_comm_init time, so single-GPU use has no RCCL dependency.
//
//   RcclTransport : production, one rank per GPU.
//   SelfTransport : one rank whose lower and upper z-neighbour are itself (device-to-device copies);
//                   exercises the whole ghost/migration machinery on a single GPU.
//   LocalTransport: several ranks as threads of one process (shared hub).
//   IpcTransport  : one PROCESS per rank, buffers mapped through hipIpc handles, rendezvous in POSIX shared
//                   memory: the real multi-process flow on a box with fewer GPUs than ranks.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "chem_host.hpp"

namespace chem {

struct Transport {
  int nranks = 1, rank = 0;
  virtual ~Transport() {}
  // One exchange phase with the z-neighbours: `dn` goes to the lower neighbour, `up` to the upper
  // one; `from_up` receives what the upper neighbour sent down, `from_lo` what the lower sent up.
  // Byte counts; all four may differ.  Enqueued on `s`.
  virtual void exchange(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes,
                        void* from_lo, size_t from_lo_bytes, int lower, int upper, hipStream_t s) = 0;
  // Two exchanges with the same neighbours in one communication phase (ghost positions + ghost tags).
  struct Msg { const void* dn; size_t dn_bytes; const void* up; size_t up_bytes; void* from_up; size_t from_up_bytes; void* from_lo; size_t from_lo_bytes; };
  virtual void exchange2(const Msg& a, const Msg& b, int lower, int upper, hipStream_t s) {
    exchange(a.dn, a.dn_bytes, a.up, a.up_bytes, a.from_up, a.from_up_bytes, a.from_lo, a.from_lo_bytes, lower, upper, s);
    exchange(b.dn, b.dn_bytes, b.up, b.up_bytes, b.from_up, b.from_up_bytes, b.from_lo, b.from_lo_bytes, lower, upper, s);
  }
  // The same exchange plus, in the same communication phase, every rank's `my` value delivered to
  // slot [rank] of every other rank's `all` array (the step's displacement maximum: the list-rebuild
  // decision needs the global max, and a separate all-reduce would cost a second launch + latency).
  virtual void exchange_with_scalar(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes,
                                    void* from_lo, size_t from_lo_bytes, int lower, int upper, const double* my, double* all, hipStream_t s, int count = 1) = 0;
  // (count: 8-byte words per rank -- slot [rank * count .. ) of `all`)
  virtual void allreduce_max_f64(double* dev, size_t count, hipStream_t s) = 0;
  virtual void allreduce_sum_f64(double* dev, size_t count, hipStream_t s) = 0;
  // out must hold nranks*bytes; in may alias out + rank*bytes
  virtual void allgather(const void* in, void* out, size_t bytes, hipStream_t s) = 0;
};

struct SelfTransport : Transport {
  void exchange(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes, void* from_lo,
                size_t from_lo_bytes, int, int, hipStream_t s) override {
    // my own "down" message arrives from my upper neighbour (= me) and vice versa
    if (dn_bytes != from_up_bytes || up_bytes != from_lo_bytes) throw ChemError(CHEM_ECOMM, "self transport: size mismatch");
    if (dn_bytes && hipMemcpyAsync(from_up, dn, dn_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) throw ChemError(CHEM_ECOMM, "self copy");
    if (up_bytes && hipMemcpyAsync(from_lo, up, up_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) throw ChemError(CHEM_ECOMM, "self copy");
  }
  void exchange_with_scalar(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes,
                            void* from_lo, size_t from_lo_bytes, int lower, int upper, const double* my, double* all, hipStream_t s, int count = 1) override {
    exchange(dn, dn_bytes, up, up_bytes, from_up, from_up_bytes, from_lo, from_lo_bytes, lower, upper, s);
    (void)hipMemcpyAsync(all, my, sizeof(double) * count, hipMemcpyDeviceToDevice, s);
  }
  void allreduce_max_f64(double*, size_t, hipStream_t) override {}
  void allreduce_sum_f64(double*, size_t, hipStream_t) override {}
  void allgather(const void* in, void* out, size_t bytes, hipStream_t s) override {
    if (in != out && bytes) (void)hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, s);
  }
};

// ---- RCCL through dlopen ---------------------------------------------------------------
struct RcclApi {
  typedef struct { char internal[128]; } UniqueId;
  typedef void* Comm;
  // enums mirror rccl.h
  enum { kInt8 = 0, kFloat64 = 8 };
  enum { kSum = 0, kMax = 2 };
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
  std::string err;

  bool load() {
    if (handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (handle) break; }
    if (!handle) { err = std::string("cannot dlopen librccl: ") + dlerror(); return false; }
#define SYM(field, name) *(void**)(&field) = dlsym(handle, name); if (!field) { err = std::string("missing RCCL symbol ") + name; return false; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return true;
  }
};

inline RcclApi& rccl_api() { static RcclApi a; return a; }

struct RcclTransport : Transport {
  RcclApi::Comm comm = nullptr;
  void ck(int rc, const char* what) {
    if (rc != 0) throw ChemError(CHEM_ECOMM, std::string(what) + ": " + rccl_api().GetErrorString(rc));
  }
  RcclTransport(int nr, int rk, const char uid[128]) {
    nranks = nr; rank = rk;
    RcclApi& a = rccl_api();
    if (!a.load()) throw ChemError(CHEM_ECOMM, a.err);
    RcclApi::UniqueId id; std::memcpy(id.internal, uid, 128);
    ck(a.CommInitRank(&comm, nr, id, rk), "ncclCommInitRank");
  }
  ~RcclTransport() override { if (comm) rccl_api().CommDestroy(comm); }
  void exchange(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes, void* from_lo,
                size_t from_lo_bytes, int lower, int upper, hipStream_t s) override {
    RcclApi& a = rccl_api();
    // Order matters when lower == upper (two ranks): the k-th send to a peer pairs with its k-th
    // receive from us.  Everybody sends [down, up] and receives [from upper, from lower].
    ck(a.GroupStart(), "ncclGroupStart");
    ck(a.Send(dn, dn_bytes, RcclApi::kInt8, lower, comm, s), "ncclSend");
    ck(a.Send(up, up_bytes, RcclApi::kInt8, upper, comm, s), "ncclSend");
    ck(a.Recv(from_up, from_up_bytes, RcclApi::kInt8, upper, comm, s), "ncclRecv");
    ck(a.Recv(from_lo, from_lo_bytes, RcclApi::kInt8, lower, comm, s), "ncclRecv");
    ck(a.GroupEnd(), "ncclGroupEnd");
  }
  void exchange2(const Msg& m1, const Msg& m2, int lower, int upper, hipStream_t s) override {
    RcclApi& a = rccl_api();
    ck(a.GroupStart(), "ncclGroupStart");          // same pairing rule as exchange(): k-th send to a peer <-> its k-th receive from us
    for (const Msg* m : {&m1, &m2}) {
      ck(a.Send(m->dn, m->dn_bytes, RcclApi::kInt8, lower, comm, s), "ncclSend");
      ck(a.Send(m->up, m->up_bytes, RcclApi::kInt8, upper, comm, s), "ncclSend");
    }
    for (const Msg* m : {&m1, &m2}) {
      ck(a.Recv(m->from_up, m->from_up_bytes, RcclApi::kInt8, upper, comm, s), "ncclRecv");
      ck(a.Recv(m->from_lo, m->from_lo_bytes, RcclApi::kInt8, lower, comm, s), "ncclRecv");
    }
    ck(a.GroupEnd(), "ncclGroupEnd");
  }
  void exchange_with_scalar(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes,
                            void* from_lo, size_t from_lo_bytes, int lower, int upper, const double* my, double* all, hipStream_t s, int count = 1) override {
    RcclApi& a = rccl_api();
    ck(a.GroupStart(), "ncclGroupStart");
    ck(a.Send(dn, dn_bytes, RcclApi::kInt8, lower, comm, s), "ncclSend");
    ck(a.Send(up, up_bytes, RcclApi::kInt8, upper, comm, s), "ncclSend");
    ck(a.Recv(from_up, from_up_bytes, RcclApi::kInt8, upper, comm, s), "ncclRecv");
    ck(a.Recv(from_lo, from_lo_bytes, RcclApi::kInt8, lower, comm, s), "ncclRecv");
    for (int r = 0; r < nranks; ++r) {   // 8-byte all-to-all riding in the same group (one launch, one latency)
      ck(a.Send(my, sizeof(double) * count, RcclApi::kInt8, r, comm, s), "ncclSend");
      ck(a.Recv(all + (size_t)r * count, sizeof(double) * count, RcclApi::kInt8, r, comm, s), "ncclRecv");
    }
    ck(a.GroupEnd(), "ncclGroupEnd");
  }
  void allreduce_max_f64(double* dev, size_t count, hipStream_t s) override {
    ck(rccl_api().AllReduce(dev, dev, count, RcclApi::kFloat64, RcclApi::kMax, comm, s), "ncclAllReduce");
  }
  void allreduce_sum_f64(double* dev, size_t count, hipStream_t s) override {
    ck(rccl_api().AllReduce(dev, dev, count, RcclApi::kFloat64, RcclApi::kSum, comm, s), "ncclAllReduce");
  }
  void allgather(const void* in, void* out, size_t bytes, hipStream_t s) override {
    ck(rccl_api().AllGather(in, out, bytes, RcclApi::kInt8, comm, s), "ncclAllGather");
  }
};


// ---- in-process transport: several contexts (ranks) of ONE process, each driven by its own host
// thread, exchange through a shared hub with device-to-device copies.  Used to validate the
// multi-rank logic (distinct slabs, real neighbours, gathers) on a single GPU; also a way to run
// several domains on one device.
struct LocalHub {
  int P = 0;
  std::mutex mu; std::condition_variable cv;
  int arrived = 0; long long gen = 0;
  struct Post { const void* a = nullptr; size_t ab = 0; const void* b = nullptr; size_t bb = 0; };
  std::vector<Post> post;
  std::vector<double> red;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long long g = gen;
    if (++arrived == P) { arrived = 0; ++gen; cv.notify_all(); }
    else if (!cv.wait_for(lk, std::chrono::seconds(90), [&] { return gen != g; }))      // (a rank that failed never arrives: an error here, not a hang)
      throw ChemError(CHEM_ECOMM, "local transport: a rank did not reach the rendezvous within 90 s (did it fail?)");
  }
};

inline std::shared_ptr<LocalHub> local_hub(int id, int P) {
  static std::mutex m; static std::map<int, std::shared_ptr<LocalHub>> hubs;
  std::lock_guard<std::mutex> lk(m);
  auto& h = hubs[id];
  if (!h) { h = std::make_shared<LocalHub>(); h->P = P; h->post.resize(P); }
  if (h->P != P) throw ChemError(CHEM_EINVAL, "local hub: rank count mismatch");
  return h;
}

struct LocalTransport : Transport {
  std::shared_ptr<LocalHub> hub;
  LocalTransport(int nr, int rk, int hub_id) { nranks = nr; rank = rk; hub = local_hub(hub_id, nr); }
  static void ck(hipError_t e) { if (e != hipSuccess) throw ChemError(CHEM_ECOMM, std::string("local transport: ") + hipGetErrorString(e)); }
  void exchange(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes, void* from_lo,
                size_t from_lo_bytes, int lower, int upper, hipStream_t s) override {
    ck(hipStreamSynchronize(s));
    hub->post[rank] = LocalHub::Post{dn, dn_bytes, up, up_bytes};
    hub->barrier();
    const LocalHub::Post& pu = hub->post[upper]; const LocalHub::Post& pl = hub->post[lower];
    if (pu.ab != from_up_bytes || pl.bb != from_lo_bytes) throw ChemError(CHEM_ECOMM, "local transport: message size mismatch");
    if (from_up_bytes) ck(hipMemcpyAsync(from_up, pu.a, from_up_bytes, hipMemcpyDeviceToDevice, s));
    if (from_lo_bytes) ck(hipMemcpyAsync(from_lo, pl.b, from_lo_bytes, hipMemcpyDeviceToDevice, s));
    ck(hipStreamSynchronize(s));
    hub->barrier();
  }
  void exchange_with_scalar(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes,
                            void* from_lo, size_t from_lo_bytes, int lower, int upper, const double* my, double* all, hipStream_t s, int count = 1) override {
    exchange(dn, dn_bytes, up, up_bytes, from_up, from_up_bytes, from_lo, from_lo_bytes, lower, upper, s);
    allgather(my, all, sizeof(double) * count, s);
  }
  void allreduce(double* dev, size_t count, hipStream_t s, bool is_max) {
    std::vector<double> v(count);
    ck(hipMemcpyAsync(v.data(), dev, count * sizeof(double), hipMemcpyDeviceToHost, s)); ck(hipStreamSynchronize(s));
    {
      std::lock_guard<std::mutex> lk(hub->mu);
      if (hub->red.size() != count * (size_t)nranks) hub->red.assign(count * (size_t)nranks, 0.0);
      for (size_t k = 0; k < count; ++k) hub->red[(size_t)rank * count + k] = v[k];
    }
    hub->barrier();
    for (size_t k = 0; k < count; ++k) {
      double r = hub->red[k];
      for (int q = 1; q < nranks; ++q) { const double o = hub->red[(size_t)q * count + k]; r = is_max ? (o > r ? o : r) : r + o; }
      v[k] = r;
    }
    ck(hipMemcpyAsync(dev, v.data(), count * sizeof(double), hipMemcpyHostToDevice, s)); ck(hipStreamSynchronize(s));
    hub->barrier();
  }
  void allreduce_max_f64(double* dev, size_t count, hipStream_t s) override { allreduce(dev, count, s, true); }
  void allreduce_sum_f64(double* dev, size_t count, hipStream_t s) override { allreduce(dev, count, s, false); }
  void allgather(const void* in, void* out, size_t bytes, hipStream_t s) override {
    ck(hipStreamSynchronize(s));
    hub->post[rank] = LocalHub::Post{in, bytes, nullptr, 0};
    hub->barrier();
    for (int q = 0; q < nranks; ++q) {
      void* dst = (char*)out + (size_t)q * bytes;
      if (hub->post[q].a != dst && bytes) ck(hipMemcpyAsync(dst, hub->post[q].a, bytes, hipMemcpyDeviceToDevice, s));
    }
    ck(hipStreamSynchronize(s));
    hub->barrier();
  }
};


// ---- inter-process transport over hipIpc memory handles: one process per rank, ranks on the same or on
// different devices of one node.  Same protocol as LocalTransport (post buffer, rendezvous, copy what the
// neighbour posted, rendezvous), with the rendezvous in a POSIX shared-memory segment and the posted
// buffers mapped through hipIpcOpenMemHandle (mappings cached per peer allocation).
// Why it exists: RCCL refuses two ranks on one device, so on a one-GPU box this is the only way to run the
// REAL multi-process flow (torchrun rendezvous, per-rank set-up, collective order, teardown) end to end;
// production multi-GPU runs use RcclTransport.
struct IpcShm {
  static constexpr int kMaxRanks = 16;
  struct Post { hipIpcMemHandle_t h[2]; unsigned long long off[2], bytes[2]; };
  unsigned int magic, P;
  volatile unsigned int arrived, gen;         // sense-reversing barrier (GCC atomics)
  volatile unsigned int attached;
  Post post[kMaxRanks];
  double red[kMaxRanks][64];
};

struct IpcTransport : Transport {
  IpcShm* shm = nullptr; std::string name; bool owner = false;
  std::map<std::string, void*> opened;        // (rank, handle bytes) -> mapped base pointer
  static void ck(hipError_t e, const char* what) { if (e != hipSuccess) throw ChemError(CHEM_ECOMM, std::string("ipc transport: ") + what + ": " + hipGetErrorString(e)); }
  IpcTransport(int nr, int rk, const char* shm_name);
  ~IpcTransport() override;
  void barrier() {
    const unsigned int g = __atomic_load_n(&shm->gen, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&shm->arrived, 1u, __ATOMIC_ACQ_REL) == (unsigned)nranks) {
      __atomic_store_n(&shm->arrived, 0u, __ATOMIC_RELAXED);
      __atomic_add_fetch(&shm->gen, 1u, __ATOMIC_RELEASE);
      return;
    }
    const double t0 = now();
    while (__atomic_load_n(&shm->gen, __ATOMIC_ACQUIRE) == g) {
      sched_yield();
      if (now() - t0 > 120.0) throw ChemError(CHEM_ECOMM, "ipc transport: a rank did not reach the rendezvous within 120 s");
    }
  }
  static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
  void fill(IpcShm::Post& p, int k, const void* ptr, size_t bytes) {
    p.bytes[k] = bytes; p.off[k] = 0;
    if (!bytes) return;
    hipDeviceptr_t base = nullptr; size_t size = 0;
    ck(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr), "hipMemGetAddressRange");
    ck(hipIpcGetMemHandle(&p.h[k], base), "hipIpcGetMemHandle");
    p.off[k] = (unsigned long long)((const char*)ptr - (const char*)base);
  }
  const void* peer_ptr(int q, int k) {
    const IpcShm::Post& p = shm->post[q];
    if (q == rank) throw ChemError(CHEM_ECOMM, "ipc transport: internal (self peer)");
    std::string key((const char*)&p.h[k], sizeof(hipIpcMemHandle_t));
    key.push_back((char)q);
    auto it = opened.find(key);
    if (it == opened.end()) {
      void* base = nullptr;
      ck(hipIpcOpenMemHandle(&base, p.h[k], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
      it = opened.emplace(key, base).first;
    }
    return (const char*)it->second + p.off[k];
  }
  void copy_from(int q, int k, void* dst, size_t bytes, const void* own_src, hipStream_t s) {
    if (shm->post[q].bytes[k] != bytes) throw ChemError(CHEM_ECOMM, "ipc transport: message size mismatch");
    if (!bytes) return;
    const void* src = q == rank ? own_src : peer_ptr(q, k);
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) {
      hipDeviceptr_t db = nullptr; size_t dsz = 0;
      (void)hipMemGetAddressRange(&db, &dsz, (hipDeviceptr_t)dst);
      throw ChemError(CHEM_ECOMM, std::string("ipc transport: copy from rank ") + std::to_string(q) + " slot " + std::to_string(k) + ": " + hipGetErrorString(e) + " (" +
                      std::to_string(bytes) + " bytes at peer offset " + std::to_string((unsigned long long)shm->post[q].off[k]) + ", destination offset " +
                      std::to_string((size_t)((const char*)dst - (const char*)db)) + " of " + std::to_string(dsz) + ", mappings cached " + std::to_string(opened.size()) + ")");
    }
  }
  void exchange(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes, void* from_lo,
                size_t from_lo_bytes, int lower, int upper, hipStream_t s) override {
    ck(hipStreamSynchronize(s), "sync");
    fill(shm->post[rank], 0, dn, dn_bytes); fill(shm->post[rank], 1, up, up_bytes);
    barrier();
    copy_from(upper, 0, from_up, from_up_bytes, dn, s);    // what the upper neighbour sent down
    copy_from(lower, 1, from_lo, from_lo_bytes, up, s);    // what the lower neighbour sent up
    ck(hipStreamSynchronize(s), "sync");
    barrier();
  }
  void exchange_with_scalar(const void* dn, size_t dn_bytes, const void* up, size_t up_bytes, void* from_up, size_t from_up_bytes,
                            void* from_lo, size_t from_lo_bytes, int lower, int upper, const double* my, double* all, hipStream_t s, int count = 1) override {
    exchange(dn, dn_bytes, up, up_bytes, from_up, from_up_bytes, from_lo, from_lo_bytes, lower, upper, s);
    if (count == 1) {
      double v = 0;
      ck(hipMemcpyAsync(&v, my, sizeof(double), hipMemcpyDeviceToHost, s), "scalar to host"); ck(hipStreamSynchronize(s), "sync");
      shm->red[rank][0] = v;
      barrier();
      double a[IpcShm::kMaxRanks];
      for (int q = 0; q < nranks; ++q) a[q] = shm->red[q][0];
      ck(hipMemcpyAsync(all, a, sizeof(double) * nranks, hipMemcpyHostToDevice, s), "scalars to device"); ck(hipStreamSynchronize(s), "sync");
      barrier();
    } else allgather(my, all, sizeof(double) * count, s);
  }
  void allreduce(double* dev, size_t count, hipStream_t s, bool is_max) {
    if (count > 64) throw ChemError(CHEM_ECOMM, "ipc transport: reduction of more than 64 values");
    double v[64];
    ck(hipMemcpyAsync(v, dev, count * sizeof(double), hipMemcpyDeviceToHost, s), "reduction to host"); ck(hipStreamSynchronize(s), "sync");
    for (size_t k = 0; k < count; ++k) shm->red[rank][k] = v[k];
    barrier();
    for (size_t k = 0; k < count; ++k) {
      double r = shm->red[0][k];
      for (int q = 1; q < nranks; ++q) { const double o = shm->red[q][k]; r = is_max ? (o > r ? o : r) : r + o; }
      v[k] = r;
    }
    ck(hipMemcpyAsync(dev, v, count * sizeof(double), hipMemcpyHostToDevice, s), "reduction to device"); ck(hipStreamSynchronize(s), "sync");
    barrier();
  }
  void allreduce_max_f64(double* dev, size_t count, hipStream_t s) override { allreduce(dev, count, s, true); }
  void allreduce_sum_f64(double* dev, size_t count, hipStream_t s) override { allreduce(dev, count, s, false); }
  void allgather(const void* in, void* out, size_t bytes, hipStream_t s) override {
    ck(hipStreamSynchronize(s), "sync");
    fill(shm->post[rank], 0, in, bytes); shm->post[rank].bytes[1] = 0;
    barrier();
    for (int q = 0; q < nranks; ++q) {
      void* dst = (char*)out + (size_t)q * bytes;
      if (q == rank) { if (in != dst && bytes) ck(hipMemcpyAsync(dst, in, bytes, hipMemcpyDeviceToDevice, s), "gather: own share"); }
      else copy_from(q, 0, dst, bytes, nullptr, s);
    }
    ck(hipStreamSynchronize(s), "sync");
    barrier();
  }
};

}  // namespace chem