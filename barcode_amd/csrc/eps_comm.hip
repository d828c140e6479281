// eps_comm.hip -- the path's only exchange: per-sample all-gather of step-size / acceptance records between the
// independent chains of one node (include/bchmc.h "cross-chain step-size statistics").
//
// Two transports behind one entry point:
//   * RCCL (ncclAllGather over xGMI): 520 bytes per rank on a side stream of the chain's GPU.  librccl is dlopen'ed
//     on first use, so a single-chain run -- the reference's only mode -- never depends on it.
//   * custom: a caller-supplied host all-gather (MPI in an MPI-launched barcode, an in-process stub in the tests).
// The collective is latency-bound (tens of microseconds once per SAMPLE, i.e. per 1..itmax trajectories of several
// milliseconds each); nothing here is on the timed leapfrog path.
#include "../../include/bchmc.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

namespace {

// wire format of one rank's contribution: 8 + 32 * 16 = 520 bytes
struct Packet {
  int32_t n;
  int32_t reserved;
  bchmc_eps_record rec[BCHMC_EPS_BATCH];
};
static_assert(sizeof(Packet) == 8 + 16 * BCHMC_EPS_BATCH, "packet layout");
static_assert(sizeof(ncclUniqueId) == BCHMC_UNIQUE_ID_BYTES, "ncclUniqueId size");

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the process may already hold an RCCL (PyTorch ships one): dlopen by SONAME returns that copy
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      r.err = std::string("dlopen(librccl.so.1): ") + dlerror();
      return;
    }
    auto sym = [&](const char *n) {
      void *p = dlsym(r.lib, n);
      if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return &r;
}

}  // namespace

struct bchmc_comm {
  int rank = 0, world = 1, device = 0;
  std::string err;
  std::deque<bchmc_eps_record> queue;  // own records not yet sent
  std::vector<Packet> recv;            // world packets (host)
  // custom transport
  bchmc_allgather_fn fn = nullptr;
  void *ctx = nullptr;
  // RCCL transport
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  Packet *d_send = nullptr, *d_recv = nullptr;  // device staging
  Packet *h_pin = nullptr;                      // pinned host staging: 1 + world packets

  int fail(int code, const std::string &m) {
    err = m;
    return code;
  }
};

extern "C" {

const char *bchmc_comm_last_error(const bchmc_comm *c) { return c ? c->err.c_str() : ""; }

int bchmc_comm_unique_id(unsigned char id[BCHMC_UNIQUE_ID_BYTES]) {
  if (!id) return BCHMC_ERR_ARG;
  Rccl *r = rccl();
  if (!r->err.empty()) return BCHMC_ERR_UNSUPPORTED;
  ncclUniqueId u;
  if (r->GetUniqueId(&u) != ncclSuccess) return BCHMC_ERR_HIP;
  std::memcpy(id, &u, BCHMC_UNIQUE_ID_BYTES);
  return BCHMC_OK;
}

int bchmc_comm_create_custom(bchmc_allgather_fn fn, void *ctx, int rank, int world, bchmc_comm **out) {
  if (!out) return BCHMC_ERR_ARG;
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world || (!fn && world > 1)) return BCHMC_ERR_ARG;
  bchmc_comm *c = new bchmc_comm();
  c->rank = rank;
  c->world = world;
  c->fn = fn;
  c->ctx = ctx;
  c->recv.resize((size_t)world);
  *out = c;
  return BCHMC_OK;
}

int bchmc_comm_create(const unsigned char id[BCHMC_UNIQUE_ID_BYTES], int rank, int world, int device,
                      bchmc_comm **out) {
  if (!out) return BCHMC_ERR_ARG;
  *out = nullptr;
  if (!id || world < 1 || rank < 0 || rank >= world) return BCHMC_ERR_ARG;
  bchmc_comm *c = new bchmc_comm();
  *out = c;  // kept on failure so that the caller can read bchmc_comm_last_error; freed by bchmc_comm_destroy
  c->rank = rank;
  c->world = world;
  c->device = device;
  c->recv.resize((size_t)world);
  Rccl *r = rccl();
  if (!r->err.empty()) return c->fail(BCHMC_ERR_UNSUPPORTED, r->err);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_send, sizeof(Packet));
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_recv, sizeof(Packet) * (size_t)world);
  if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_pin, sizeof(Packet) * (size_t)(world + 1));
  if (e != hipSuccess) return c->fail(BCHMC_ERR_HIP, std::string("bchmc_comm_create: ") + hipGetErrorString(e));
  ncclUniqueId u;
  std::memcpy(&u, id, BCHMC_UNIQUE_ID_BYTES);
  const ncclResult_t rc = r->CommInitRank(&c->comm, world, u, rank);
  if (rc != ncclSuccess) {
    c->comm = nullptr;
    return c->fail(BCHMC_ERR_HIP, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
  }
  return BCHMC_OK;
}

void bchmc_comm_destroy(bchmc_comm *c) {
  if (!c) return;
  if (c->stream || c->comm) (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl()->CommDestroy(c->comm);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int bchmc_comm_pending(const bchmc_comm *c) { return c ? (int)c->queue.size() : 0; }
int bchmc_comm_world(const bchmc_comm *c) { return c ? c->world : 0; }
int bchmc_comm_rank(const bchmc_comm *c) { return c ? c->rank : -1; }
const char *bchmc_comm_transport(const bchmc_comm *c) {
  if (!c) return "";
  if (c->comm) return "rccl";
  return c->fn ? "custom" : "none";  // "none": a one-rank communicator without a transport
}

int bchmc_eps_exchange(bchmc_comm *c, const bchmc_eps_record *mine, int n_mine, bchmc_eps_record *all, int *rank_of,
                       int cap, int *n_all) {
  if (!c || !all || !n_all || n_mine < 0 || (n_mine > 0 && !mine) || cap < 0) return BCHMC_ERR_ARG;
  *n_all = 0;
  // every argument and state check comes before anything is queued or sent: a call that fails has no effect on this
  // rank (its records are NOT kept: retry with the same `mine`), and one that could fail after the collective ran
  // would leave the peers holding records this rank reports as unsent
  if ((long long)cap < (long long)c->world * BCHMC_EPS_BATCH)
    return c->fail(BCHMC_ERR_ARG, "output capacity " + std::to_string(cap) + " < world * BCHMC_EPS_BATCH = " +
                                      std::to_string((long long)c->world * BCHMC_EPS_BATCH));
  if (c->world > 1 && !c->fn && !c->comm) return c->fail(BCHMC_ERR_STATE, "communicator was not initialised");
  for (int i = 0; i < n_mine; i++) c->queue.push_back(mine[i]);
  struct Unqueue {  // failure below: take this call's records back out
    std::deque<bchmc_eps_record> &q;
    int n;
    ~Unqueue() {
      for (int i = 0; i < n; i++) q.pop_back();
    }
  } unqueue{c->queue, n_mine};
  Packet send;
  std::memset(&send, 0, sizeof send);
  send.n = (int32_t)std::min<size_t>(c->queue.size(), BCHMC_EPS_BATCH);
  for (int i = 0; i < send.n; i++) send.rec[i] = c->queue[(size_t)i];
  if (c->world == 1) {
    c->recv[0] = send;
  } else if (c->fn) {
    const int rc = c->fn(c->ctx, &send, c->recv.data(), sizeof(Packet));
    if (rc) return c->fail(BCHMC_ERR_STATE, "custom all-gather transport returned " + std::to_string(rc));
  } else {
    if (!c->comm) return c->fail(BCHMC_ERR_STATE, "communicator was not initialised");
    Rccl *r = rccl();
    hipError_t e = hipSetDevice(c->device);
    c->h_pin[0] = send;
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_send, c->h_pin, sizeof(Packet), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return c->fail(BCHMC_ERR_HIP, std::string("bchmc_eps_exchange: ") + hipGetErrorString(e));
    const ncclResult_t rc = r->AllGather(c->d_send, c->d_recv, sizeof(Packet), ncclChar, c->comm, c->stream);
    if (rc != ncclSuccess) return c->fail(BCHMC_ERR_HIP, std::string("ncclAllGather: ") + r->GetErrorString(rc));
    e = hipMemcpyAsync(c->h_pin + 1, c->d_recv, sizeof(Packet) * (size_t)c->world, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return c->fail(BCHMC_ERR_HIP, std::string("bchmc_eps_exchange: ") + hipGetErrorString(e));
    std::memcpy(c->recv.data(), c->h_pin + 1, sizeof(Packet) * (size_t)c->world);
  }
  // validate before consuming: a transport that scribbles must not turn into an out-of-bounds read here
  for (int rk = 0; rk < c->world; rk++) {
    const int n = c->recv[(size_t)rk].n;  // <= BCHMC_EPS_BATCH each: the sum fits `cap` (checked above)
    if (n < 0 || n > BCHMC_EPS_BATCH) return c->fail(BCHMC_ERR_STATE, "malformed packet from rank " + std::to_string(rk));
  }
  if (c->recv[(size_t)c->rank].n != send.n) return c->fail(BCHMC_ERR_STATE, "own packet came back altered");
  unqueue.n = 0;                                          // the exchange happened: this call's records stay queued / sent
  for (int i = 0; i < send.n; i++) c->queue.pop_front();  // sent: every rank now holds them
  int k = 0;
  for (int rk = 0; rk < c->world; rk++)
    for (int i = 0; i < c->recv[(size_t)rk].n; i++) {
      all[k] = c->recv[(size_t)rk].rec[i];
      if (rank_of) rank_of[k] = rk;
      k++;
    }
  *n_all = k;
  return BCHMC_OK;
}

}  // extern "C"
