// The one collective of the path: the per-step all-gather of updated position shards for multi-GPU all-pairs
// (north_star) on RCCL over xGMI.  The reference has no counterpart (one process, one device).
//
// Partition (fixed by the ABI so that every rank, the tests and the host agree): rank r of W owns bodies
// [sz*r/W, sz*(r+1)/W).  Every rank holds the full x array; after K3 has moved the owned rows, the exchange fills in
// everybody else's rows IN PLACE — no staging copy:
//   * sz % W == 0: one ncclAllGather with sendbuff = recvbuff + rank*count (NCCL's in-place form);
//   * otherwise:   one grouped ncclSend/ncclRecv per peer straight between the x arrays.  On the fully connected
//     xGMI mesh of an MI355X node (7 links per GPU, one per peer) that is also the natural schedule: every shard
//     travels over its own link, there is no ring.
// Message size at N = 2^20, W = 8, 3D double: 3 MiB per rank, 24 MiB gathered — microseconds of wire time against a
// >= 90 ms force pass, so the exchange is issued on the compute stream right after K3 and not overlapped.
//
// RCCL is resolved lazily (dlopen of librccl.so.1) so that the library has no load-time dependency on it: single-GPU
// users never touch it, and inside a PyTorch process the copy PyTorch already mapped is reused (one RCCL per process,
// as for libamdhip64).
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <mutex>
#include <vector>

namespace nbody {
namespace {

struct rccl_api {
  decltype(&ncclGetUniqueId) GetUniqueId   = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll   = nullptr;
  decltype(&ncclCommDestroy) CommDestroy   = nullptr;
  decltype(&ncclAllGather) AllGather       = nullptr;
  decltype(&ncclSend) Send                 = nullptr;
  decltype(&ncclRecv) Recv                 = nullptr;
  decltype(&ncclGroupStart) GroupStart     = nullptr;
  decltype(&ncclGroupEnd) GroupEnd         = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion     = nullptr;
  void* handle = nullptr;
  bool ok      = false;
};

rccl_api g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return;
  bool all = true;
#define NB_SYM(field, name)                                                        \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name)); \
  all          = all && g_rccl.field != nullptr
  NB_SYM(GetUniqueId, "ncclGetUniqueId");
  NB_SYM(CommInitRank, "ncclCommInitRank");
  NB_SYM(CommInitAll, "ncclCommInitAll");
  NB_SYM(CommDestroy, "ncclCommDestroy");
  NB_SYM(AllGather, "ncclAllGather");
  NB_SYM(Send, "ncclSend");
  NB_SYM(Recv, "ncclRecv");
  NB_SYM(GroupStart, "ncclGroupStart");
  NB_SYM(GroupEnd, "ncclGroupEnd");
  NB_SYM(GetErrorString, "ncclGetErrorString");
  NB_SYM(GetVersion, "ncclGetVersion");
#undef NB_SYM
  g_rccl.ok = all;
}

int need_rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.ok) {
    set_error("RCCL is not available: %s", g_rccl.handle ? "librccl.so.1 lacks a required symbol" : dlerror());
    return NBODY_ERR_STATE;
  }
  return NBODY_OK;
}

int rccl_fail(ncclResult_t r, const char* what) {
  set_error("RCCL error %d (%s) in %s", int(r), g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what);
  return NBODY_ERR_HIP;
}

// RCCL prints a version banner on stdout when the first communicator of a process is created.  stdout belongs to the
// caller (the CLI prints its CSV rows and --print-state text there): while a communicator is being created, fd 1 is
// pointed at stderr.
struct stdout_to_stderr {
  int saved = -1;
  stdout_to_stderr() {
    fflush(stdout);
    saved = dup(1);
    if (saved >= 0) (void)dup2(2, 1);
  }
  ~stdout_to_stderr() {
    if (saved >= 0) {
      fflush(stdout);
      (void)dup2(saved, 1);
      close(saved);
    }
  }
};

#define NB_RCCL(call)                                        \
  do {                                                       \
    ncclResult_t r_ = (call);                                \
    if (r_ != ncclSuccess) return rccl_fail(r_, #call);      \
  } while (0)

}  // namespace
}  // namespace nbody

using namespace nbody;

struct nbody_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
};

static_assert(NBODY_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "nbody_hip.h must carry RCCL's unique-id size");

extern "C" int nbody_comm_get_unique_id(void* id_out) {
  NB_ARG(id_out != nullptr, "id_out is NULL");
  if (int r = need_rccl()) return r;
  ncclUniqueId id;
  NB_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return NBODY_OK;
}

extern "C" int nbody_comm_create(nbody_comm** out, int world, int rank, const void* unique_id, int device) {
  NB_ARG(out != nullptr && unique_id != nullptr, "NULL argument");
  *out = nullptr;
  NB_ARG(world >= 1 && rank >= 0 && rank < world, "bad (world=%d, rank=%d)", world, rank);
  if (int r = need_rccl()) return r;
  int ndev = 0;
  NB_HIP(hipGetDeviceCount(&ndev));
  NB_ARG(device >= 0 && device < ndev, "device %d out of range (%d HIP devices visible)", device, ndev);
  device_guard guard(device);
  ncclUniqueId id;
  memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  {
    stdout_to_stderr quiet;
    NB_RCCL(g_rccl.CommInitRank(&comm, world, id, rank));
  }
  auto* c   = new nbody_comm;
  c->comm   = comm;
  c->world  = world;
  c->rank   = rank;
  c->device = device;
  *out      = c;
  return NBODY_OK;
}

extern "C" int nbody_comm_create_all(nbody_comm** out, int ndev, const int* devices) {
  NB_ARG(out != nullptr && ndev >= 1, "bad argument");
  for (int i = 0; i < ndev; ++i) out[i] = nullptr;
  if (int r = need_rccl()) return r;
  int visible = 0;
  NB_HIP(hipGetDeviceCount(&visible));
  NB_ARG(ndev <= visible, "%d GPUs requested, %d HIP devices visible", ndev, visible);
  std::vector<int> devs(static_cast<size_t>(ndev));
  for (int i = 0; i < ndev; ++i) {
    devs[size_t(i)] = devices ? devices[i] : i;
    NB_ARG(devs[size_t(i)] >= 0 && devs[size_t(i)] < visible, "device %d out of range (%d HIP devices visible)", devs[size_t(i)],
           visible);
  }
  std::vector<ncclComm_t> comms(static_cast<size_t>(ndev), nullptr);
  {
    stdout_to_stderr quiet;
    NB_RCCL(g_rccl.CommInitAll(comms.data(), ndev, devs.data()));
  }
  for (int i = 0; i < ndev; ++i) {
    auto* c   = new nbody_comm;
    c->comm   = comms[size_t(i)];
    c->world  = ndev;
    c->rank   = i;
    c->device = devs[size_t(i)];
    out[i]    = c;
  }
  return NBODY_OK;
}

extern "C" void nbody_comm_destroy(nbody_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.ok) {
    device_guard guard(c->device);
    (void)g_rccl.CommDestroy(c->comm);
  }
  delete c;
}

extern "C" int nbody_comm_world(const nbody_comm* c) { return c ? c->world : 0; }
extern "C" int nbody_comm_rank(const nbody_comm* c) { return c ? c->rank : -1; }

extern "C" int nbody_comm_rccl_version(void) {
  if (need_rccl()) return 0;
  int v = 0;
  if (g_rccl.GetVersion(&v) != ncclSuccess) return 0;
  return v;
}

extern "C" void nbody_shard_range(uint32_t sz, int world, int rank, uint32_t* first, uint32_t* count) {
  const uint64_t f = uint64_t(sz) * uint64_t(rank) / uint64_t(world);
  const uint64_t e = uint64_t(sz) * uint64_t(rank + 1) / uint64_t(world);
  if (first) *first = uint32_t(f);
  if (count) *count = uint32_t(e - f);
}

extern "C" int nbody_comm_group_begin(void) {
  if (int r = need_rccl()) return r;
  NB_RCCL(g_rccl.GroupStart());
  return NBODY_OK;
}

extern "C" int nbody_comm_group_end(void) {
  if (int r = need_rccl()) return r;
  NB_RCCL(g_rccl.GroupEnd());
  return NBODY_OK;
}

extern "C" int nbody_allgather_positions(nbody_comm* c, const nbody_state* s, void* stream) {
  NB_ARG(c != nullptr && c->comm != nullptr, "nbody_comm is NULL");
  if (int r = check_state(s)) return r;
  uint32_t first = 0, count = 0;
  nbody_shard_range(s->sz, c->world, c->rank, &first, &count);
  NB_ARG(s->first == first && s->count == count, "rank %d of %d owns [%u, %u+%u) of %u bodies, state says [%u, %u+%u)", c->rank,
         c->world, first, first, count, s->sz, s->first, s->first, s->count);
  if (c->world == 1) return NBODY_OK;  // nothing to fetch
  device_guard guard(c->device);
  const ncclDataType_t dt = s->dtype == NBODY_F32 ? ncclFloat : ncclDouble;
  const size_t esz        = s->dtype == NBODY_F32 ? 4 : 8;
  const size_t D          = size_t(s->dim);
  char* x                 = static_cast<char*>(s->x);
  hipStream_t st          = as_stream(stream);
  if (s->sz % uint32_t(c->world) == 0) {
    NB_RCCL(g_rccl.AllGather(x + size_t(first) * D * esz, x, size_t(count) * D, dt, c->comm, st));
    return NBODY_OK;
  }
  NB_RCCL(g_rccl.GroupStart());
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    uint32_t pf = 0, pc = 0;
    nbody_shard_range(s->sz, c->world, p, &pf, &pc);
    ncclResult_t r = ncclSuccess;
    if (count) r = g_rccl.Send(x + size_t(first) * D * esz, size_t(count) * D, dt, p, c->comm, st);
    if (r == ncclSuccess && pc) r = g_rccl.Recv(x + size_t(pf) * D * esz, size_t(pc) * D, dt, p, c->comm, st);
    if (r != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return rccl_fail(r, "ncclSend/ncclRecv");
    }
  }
  NB_RCCL(g_rccl.GroupEnd());
  return NBODY_OK;
}
