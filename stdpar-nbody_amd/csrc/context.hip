// Error plumbing, device identification and the owning context (device mirrors of a host System)
// used by the ISO-C++ CLI host.  See include/nbody_hip.h for the contract.
#include "common.hpp"

#include <cstring>
#include <string>

namespace nbody {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) in %s at %s:%d", int(e), hipGetErrorString(e), what, file, line);
  return NBODY_ERR_HIP;
}

}  // namespace nbody

using namespace nbody;

struct nbody_ctx {
  int dtype = 0, dim = 0, device = 0;
  uint32_t n = 0;
  size_t tsz = 0;
  void *m = nullptr, *x = nullptr, *v = nullptr, *a = nullptr, *ao = nullptr;
  double dt = 0, c = 0;
  hipStream_t stream = nullptr;
  uint32_t first = 0, count = 0;  // owned targets (nbody_ctx_set_shard); whole system by default
  uint32_t tuning = 0;            // K1 launch shape of this context (nbody_ctx_configure_all_pairs)
};

extern "C" int nbody_abi_version(void) { return NBODY_HIP_ABI_VERSION; }

extern "C" const char* nbody_last_error(void) { return g_last_error.c_str(); }

extern "C" int nbody_device_info(int device, char* arch_out, size_t arch_len, int* cu_count) {
  int ndev = 0;
  NB_HIP(hipGetDeviceCount(&ndev));
  NB_ARG(device >= 0 && device < ndev, "device %d out of range (%d HIP devices visible)", device, ndev);
  hipDeviceProp_t p;
  NB_HIP(hipGetDeviceProperties(&p, device));
  if (arch_out && arch_len) {
    strncpy(arch_out, p.gcnArchName, arch_len - 1);
    arch_out[arch_len - 1] = 0;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  return NBODY_OK;
}

extern "C" int nbody_create(nbody_ctx** out, int dtype, int dim, uint32_t n, int device) {
  NB_ARG(out != nullptr, "out is NULL");
  *out = nullptr;
  NB_ARG(dtype == NBODY_F32 || dtype == NBODY_F64, "bad dtype %d", dtype);
  NB_ARG(dim == 2 || dim == 3, "bad dim %d", dim);
  NB_ARG(n >= 1, "n must be >= 1");
  int ndev = 0;
  NB_HIP(hipGetDeviceCount(&ndev));
  NB_ARG(device >= 0 && device < ndev, "device %d out of range (%d HIP devices visible)", device, ndev);
  device_guard guard(device);
  auto* c   = new nbody_ctx;
  c->dtype  = dtype;
  c->dim    = dim;
  c->n      = n;
  c->count  = n;
  c->device = device;
  c->tsz    = dtype == NBODY_F32 ? 4 : 8;
  const size_t vb = c->tsz * size_t(n) * size_t(dim);
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&c->m, c->tsz * size_t(n));
  if (e == hipSuccess) e = hipMalloc(&c->x, vb);
  if (e == hipSuccess) e = hipMalloc(&c->v, vb);
  if (e == hipSuccess) e = hipMalloc(&c->a, vb);
  if (e == hipSuccess) e = hipMalloc(&c->ao, vb);
  if (e != hipSuccess) {
    int r = hip_fail(e, "nbody_create allocation", __FILE__, __LINE__);
    nbody_destroy(c);
    return r;
  }
  nbody_state view;
  (void)nbody_ctx_state(c, &view);
  if (int r = ap_scratch_reserve(c->stream, &view)) {  // so that a recorded step never allocates
    nbody_destroy(c);
    return r;
  }
  *out = c;
  return NBODY_OK;
}

extern "C" void nbody_destroy(nbody_ctx* c) {
  if (!c) return;
  device_guard guard(c->device);
  (void)hipFree(c->m);
  (void)hipFree(c->x);
  (void)hipFree(c->v);
  (void)hipFree(c->a);
  (void)hipFree(c->ao);
  if (c->stream) ap_scratch_release(c->stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int nbody_upload(nbody_ctx* c, const void* m, const void* x, const void* v, const void* a, const void* ao, double dt,
                            double cc) {
  NB_ARG(c && m && x && v && a && ao, "NULL argument");
  device_guard guard(c->device);
  const size_t vb = c->tsz * size_t(c->n) * size_t(c->dim);
  NB_HIP(hipMemcpyAsync(c->m, m, c->tsz * size_t(c->n), hipMemcpyHostToDevice, c->stream));
  NB_HIP(hipMemcpyAsync(c->x, x, vb, hipMemcpyHostToDevice, c->stream));
  NB_HIP(hipMemcpyAsync(c->v, v, vb, hipMemcpyHostToDevice, c->stream));
  NB_HIP(hipMemcpyAsync(c->a, a, vb, hipMemcpyHostToDevice, c->stream));
  NB_HIP(hipMemcpyAsync(c->ao, ao, vb, hipMemcpyHostToDevice, c->stream));
  NB_HIP(hipStreamSynchronize(c->stream));
  c->dt = dt;
  c->c  = cc;
  return NBODY_OK;
}

extern "C" int nbody_download(nbody_ctx* c, void* m, void* x, void* v, void* a, void* ao) {
  NB_ARG(c != nullptr, "ctx is NULL");
  device_guard guard(c->device);
  // a sharded context returns its own rows only, at their place in the caller's full-size arrays
  const size_t row = c->tsz * size_t(c->dim);
  const size_t off = row * size_t(c->first), vb = row * size_t(c->count);
  auto rows = [&](void* host, const void* dev) {
    return hipMemcpyAsync(static_cast<char*>(host) + off, static_cast<const char*>(dev) + off, vb, hipMemcpyDeviceToHost, c->stream);
  };
  if (m) NB_HIP(hipMemcpyAsync(m, c->m, c->tsz * size_t(c->n), hipMemcpyDeviceToHost, c->stream));
  if (x) NB_HIP(rows(x, c->x));
  if (v) NB_HIP(rows(v, c->v));
  if (a) NB_HIP(rows(a, c->a));
  if (ao) NB_HIP(rows(ao, c->ao));
  return ap_status_read(c->stream, nullptr, false);  // waits for the stream; a failed K1 chunk hand-off must not pass for a result
}

extern "C" int nbody_ctx_state(nbody_ctx* c, nbody_state* out) {
  NB_ARG(c && out, "NULL argument");
  const size_t off = c->tsz * size_t(c->dim) * size_t(c->first);
  out->m      = c->m;
  out->x      = c->x;
  out->v      = static_cast<char*>(c->v) + off;  // record k of v/a/ao = body first + k
  out->a      = static_cast<char*>(c->a) + off;
  out->ao     = static_cast<char*>(c->ao) + off;
  out->dt     = c->dt;
  out->c      = c->c;
  out->sz     = c->n;
  out->first  = c->first;
  out->count  = c->count;
  out->dtype  = c->dtype;
  out->dim    = c->dim;
  out->tuning = c->tuning;
  return NBODY_OK;
}

extern "C" int nbody_ctx_set_shard(nbody_ctx* c, uint32_t first, uint32_t count) {
  NB_ARG(c != nullptr, "ctx is NULL");
  NB_ARG(uint64_t(first) + uint64_t(count) <= uint64_t(c->n), "shard [%u, %u+%u) exceeds n=%u", first, first, count, c->n);
  device_guard guard(c->device);
  c->first = first;
  c->count = count;
  nbody_state view;
  (void)nbody_ctx_state(c, &view);
  return count ? ap_scratch_reserve(c->stream, &view) : int(NBODY_OK);  // the window's launch may need scratch the whole system's did not
}

extern "C" int nbody_ctx_configure_all_pairs(nbody_ctx* c, int split, int targets_per_thread, int source_path) {
  NB_ARG(c != nullptr, "ctx is NULL");
  if (int r = check_tuning(split, targets_per_thread, source_path)) return r;
  c->tuning = (split || targets_per_thread || source_path) ? NBODY_TUNING(split, targets_per_thread, source_path) : 0u;
  device_guard guard(c->device);
  nbody_state view;
  (void)nbody_ctx_state(c, &view);
  return view.count ? ap_scratch_reserve(c->stream, &view) : int(NBODY_OK);
}

extern "C" void* nbody_ctx_stream(nbody_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

extern "C" int nbody_stream_sync(void* stream) {
  device_guard guard(stream_device(as_stream(stream)));
  return ap_status_read(as_stream(stream), nullptr, false);  // hipStreamSynchronize + the stream's K1 hand-off status (sticky)
}

// ---- step graphs ---------------------------------------------------------------------------------------------
struct nbody_graph {
  hipGraph_t graph    = nullptr;
  hipGraphExec_t exec = nullptr;
};

extern "C" int nbody_graph_begin(void* stream) {
  NB_ARG(stream != nullptr, "graph capture needs an explicit (non-default) stream");
  device_guard guard(stream_device(as_stream(stream)));
  NB_HIP(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
  return NBODY_OK;
}

extern "C" int nbody_graph_end(void* stream, nbody_graph** out) {
  NB_ARG(out != nullptr, "out is NULL");
  *out = nullptr;
  device_guard guard(stream_device(as_stream(stream)));
  hipGraph_t graph = nullptr;
  NB_HIP(hipStreamEndCapture(as_stream(stream), &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e        = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__);
  }
  auto* g  = new nbody_graph;
  g->graph = graph;
  g->exec  = exec;
  *out     = g;
  return NBODY_OK;
}

extern "C" int nbody_graph_launch(nbody_graph* g, void* stream) {
  NB_ARG(g != nullptr && g->exec != nullptr, "graph is NULL");
  device_guard guard(stream_device(as_stream(stream)));
  NB_HIP(hipGraphLaunch(g->exec, as_stream(stream)));
  ap_status_mark(as_stream(stream));
  return NBODY_OK;
}

extern "C" void nbody_graph_destroy(nbody_graph* g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
}
