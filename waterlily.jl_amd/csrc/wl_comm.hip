// z-slab communication: RCCL (dlopen'ed, shared with PyTorch) and host-callback implementations.
#include "wl_comm.hpp"

#include <dlfcn.h>

#include <cstring>

wl_comm::~wl_comm() {
  if (gather) (void)hipFree(gather);
  if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
  if (ev_ready) (void)hipEventDestroy(ev_ready);
  if (ev_done) (void)hipEventDestroy(ev_done);
}
int wl_comm::ensure_async() {
  if (!cs) {
    WL_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    WL_HIP(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
    WL_HIP(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
  }
  return 0;
}
int wl_comm::ensure_scratch() {
  if (!gather) WL_HIP(hipMalloc(&gather, (size_t)size * 128));
  return 0;
}

namespace {
// ---- minimal RCCL surface, resolved at run time (librccl.so.1: PyTorch's copy when torch is loaded) ----------
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclChar = 0 };
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  if (r.h) return r;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
  if (!r.h) return r;
#define LOADSYM(field, sym) *(void**)(&r.field) = dlsym(r.h, sym)
  LOADSYM(GetUniqueId, "ncclGetUniqueId"); LOADSYM(CommInitRank, "ncclCommInitRank"); LOADSYM(CommDestroy, "ncclCommDestroy");
  LOADSYM(Send, "ncclSend"); LOADSYM(Recv, "ncclRecv"); LOADSYM(AllGather, "ncclAllGather");
  LOADSYM(GroupStart, "ncclGroupStart"); LOADSYM(GroupEnd, "ncclGroupEnd"); LOADSYM(GetErrorString, "ncclGetErrorString");
#undef LOADSYM
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.AllGather && r.GroupStart && r.GroupEnd;
  return r;
}
#define WL_NCCL(call)                                                                                          \
  do {                                                                                                         \
    ncclResult_t e__ = (call);                                                                                 \
    if (e__ != 0) { wl_set_error(std::string(#call) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e__) : "rccl error")); return WL_ECOMM; } \
  } while (0)

// Two communicators per rank: `comm` carries everything issued on the compute stream, `comm_async` the halo exchanges that run
// on the communicator's own stream (wl::halo_async_begin) — a communicator is only ever used from ONE stream, so the two
// streams' operations need no common issue order across ranks.  Until wl_comm_rccl_add_async has been called the async
// exchanges fall back to `comm` (legal only because every rank issues the same sequence; kept for single-communicator tests).
struct RcclComm : wl_comm {
  ncclComm_t comm = nullptr, comm_async = nullptr;
  int depth = 0;
  ~RcclComm() override { if (comm_async) (void)rccl().CommDestroy(comm_async); if (comm) (void)rccl().CommDestroy(comm); }
  int group_begin() override { gdepth++; if (virt && virt_null) return 0; if (depth++ == 0) WL_NCCL(rccl().GroupStart()); return 0; }
  int group_end() override { if (gdepth > 0) gdepth--; if (virt && virt_null) return 0; if (depth > 0 && --depth == 0) WL_NCCL(rccl().GroupEnd()); return 0; }
  int sendrecv_body(ncclComm_t cm, const void* slo, void* rlo, const void* shi, void* rhi, size_t bytes, hipStream_t s) {
    // neighbours: lo = rank-1, hi = rank+1 (loopback: both are this rank — what it sends down comes back as its upper ghost planes and
    // vice versa, the z-periodic wrap; sends and receives to one peer match in issue order, hence lo-send / hi-recv first)
    const bool self = loopback || virt;
    const int me = virt ? real_rank : rank;
    const int plo = self ? me : (zperiodic ? (rank + size - 1) % size : rank - 1), phi = self ? me : (zperiodic ? (rank + 1) % size : rank + 1);
    if (slo) WL_NCCL(rccl().Send(slo, bytes, ncclChar, plo, cm, s));
    if (rhi) WL_NCCL(rccl().Recv(rhi, bytes, ncclChar, phi, cm, s));
    if (shi) WL_NCCL(rccl().Send(shi, bytes, ncclChar, phi, cm, s));
    if (rlo) WL_NCCL(rccl().Recv(rlo, bytes, ncclChar, plo, cm, s));
    return 0;
  }
  int sendrecv(const void* slo, void* rlo, const void* shi, void* rhi, size_t bytes, hipStream_t s) override {
    if (virt && virt_null) return 0;
    WL_TRY(group_begin());
    ncclComm_t cm = (comm_async && s == cs && cs) ? comm_async : comm;
    const int rc = sendrecv_body(cm, slo, rlo, shi, rhi, bytes, s);
    const int re = group_end();            // always closes the group opened above, also on the error path
    return rc ? rc : re;
  }
  int allgather(const void* send, void* recv, size_t bytes_each, hipStream_t s) override {
    if (virt && virt_null) return 0;
    if (virt) {   // rehearsal: the one-rank all-gather lands in this rank's block; the other ranks' blocks are filled with copies of it (every pretended rank = this slab)
      char* mine = (char*)recv + (size_t)rank * bytes_each;
      WL_NCCL(rccl().AllGather(send, mine, bytes_each, ncclChar, comm, s));
      for (int r = 0; r < size; r++) if (r != rank) WL_HIP(hipMemcpyAsync((char*)recv + (size_t)r * bytes_each, mine, bytes_each, hipMemcpyDeviceToDevice, s));
      return 0;
    }
    WL_NCCL(rccl().AllGather(send, recv, bytes_each, ncclChar, comm, s));
    return 0;
  }
};
struct CallbackComm : wl_comm {
  void* ctx = nullptr; wl_sendrecv_fn f_sr = nullptr; wl_allgather_fn f_ag = nullptr;
  int sendrecv(const void* slo, void* rlo, const void* shi, void* rhi, size_t bytes, hipStream_t s) override {
    const int rc = f_sr(ctx, slo, rlo, shi, rhi, bytes, (void*)s);
    if (rc != 0) { wl_set_error("halo callback failed"); return WL_ECOMM; }
    return 0;
  }
  int allgather(const void* send, void* recv, size_t bytes_each, hipStream_t s) override {
    const int rc = f_ag(ctx, send, recv, bytes_each, (void*)s);
    if (rc != 0) { wl_set_error("allgather callback failed"); return WL_ECOMM; }
    return 0;
  }
};

// res_d[q] <- Σ_r gathered[r].d[q] ; res_f[q] <- max_r gathered[r].f[q]   (one record = 8 doubles + 8 floats + pad = 128 B)
__global__ void k_combine(const char* __restrict__ gathered, int nranks, double* __restrict__ res_d, float* __restrict__ res_f) {
  const int q = threadIdx.x;
  if (q < 8) { double s = 0.0; for (int r = 0; r < nranks; r++) s += ((const double*)(gathered + (size_t)r * 128))[q]; res_d[q] = s; }
  else if (q < 16) { float m = -INFINITY; for (int r = 0; r < nranks; r++) m = fmaxf(m, ((const float*)(gathered + (size_t)r * 128 + 64))[q - 8]); res_f[q - 8] = m; }
}
}  // namespace

namespace wl {
int halo(wl_comm* c, float* a, const GridX& g, int ncomp, int depth, hipStream_t s, bool wrap) {
  if (!c || (c->size == 1 && !c->loopback) || g.D != 3 || g.nz == g.gnz) return 0;   // single domain / replicated level: nothing to exchange
  const bool per = c->zperiodic && wrap;
  const bool has_lo = c->loopback || per || (g.gk + g.k0 > 1), has_hi = c->loopback || per || (g.gk + g.k1 < g.gnz - 1);
  const size_t bytes = (size_t)depth * (size_t)g.sz * sizeof(float);
  if (g.k1 - g.k0 < depth || g.k0 < depth) { wl_set_error("halo deeper than the slab"); return WL_EINVAL; }
  if (c->gdepth == 0) c->n_halo++;     // an exchange inside an open group belongs to the round that opened it
  c->halo_bytes += (long)bytes * ncomp * ((has_lo ? 1 : 0) + (has_hi ? 1 : 0));
  WL_TRY(c->group_begin());
  for (int q = 0; q < ncomp; q++) {
    float* b = a + (size_t)q * g.cs;
    const int rc = c->sendrecv(has_lo ? b + (size_t)g.k0 * g.sz : nullptr, has_lo ? b + (size_t)(g.k0 - depth) * g.sz : nullptr,
                               has_hi ? b + (size_t)(g.k1 - depth) * g.sz : nullptr, has_hi ? b + (size_t)g.k1 * g.sz : nullptr, bytes, s);
    if (rc != 0) { (void)c->group_end(); return rc; }
  }
  return c->group_end();
}
int halo_async_begin(wl_comm* c, float* a, const GridX& g, int ncomp, int depth, hipStream_t compute) {
  if (!c || (c->size == 1 && !c->loopback) || g.D != 3 || g.nz == g.gnz) return 0;
  WL_TRY(c->ensure_async());
  WL_HIP(hipEventRecord(c->ev_ready, compute));
  WL_HIP(hipStreamWaitEvent(c->cs, c->ev_ready, 0));
  WL_TRY(halo(c, a, g, ncomp, depth, c->cs));
  WL_HIP(hipEventRecord(c->ev_done, c->cs));
  return 0;
}
int halo_async_wait(wl_comm* c, hipStream_t compute) {
  if (!c || (c->size == 1 && !c->loopback) || !c->cs) return 0;
  WL_HIP(hipStreamWaitEvent(compute, c->ev_done, 0));
  return 0;
}
int combine_results(wl_comm* c, const RedWs& ws, hipStream_t s) {
  if (!c || (c->size == 1 && !c->loopback)) return 0;
  WL_TRY(c->ensure_scratch());
  c->n_combine++;
  // res_d (64 B) and res_f (32 B at +64) are adjacent: one 128-byte record per rank
  WL_TRY(c->allgather(ws.res_d, c->gather, 128, s));
  hipLaunchKernelGGL(k_combine, dim3(1), dim3(64), 0, s, (const char*)c->gather, c->size, ws.res_d, ws.res_f);
  WL_LAUNCH_CHECK(); return 0;
}
int allgather_planes(wl_comm* c, float* a, const GridX& view, int ncomp, hipStream_t s) {
  if (!c || (c->size == 1 && !c->loopback)) return 0;
  const size_t block = (size_t)(view.k1 - view.k0) * (size_t)view.sz;       // floats per rank
  c->n_gather += ncomp;
  for (int q = 0; q < ncomp; q++) {
    float* base = a + (size_t)q * view.cs + (size_t)view.sz;                  // first interior plane of the full array
    WL_TRY(c->allgather(base + (size_t)c->rank * block, base, block * sizeof(float), s));
  }
  return 0;
}
}  // namespace wl

extern "C" {
int wl_comm_rccl_unique_id(char out[128]) {
  Rccl& r = rccl();
  if (!r.ok) { wl_set_error("librccl.so.1 could not be loaded"); return WL_ECOMM; }
  ncclUniqueId id; WL_NCCL(r.GetUniqueId(&id));
  memcpy(out, id.internal, 128); return 0;
}
int wl_comm_rccl_create(wl_comm** out, int rank, int size, const char uid[128]) {
  WL_CHECK(out && size >= 1 && rank >= 0 && rank < size, "bad rank/size");
  Rccl& r = rccl();
  if (!r.ok) { wl_set_error("librccl.so.1 could not be loaded"); return WL_ECOMM; }
  RcclComm* c = new RcclComm(); c->rank = rank; c->size = size;
  ncclUniqueId id; memcpy(id.internal, uid, 128);
  ncclResult_t e = r.CommInitRank(&c->comm, size, id, rank);
  if (e != 0) { wl_set_error(std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(e) : "error")); delete c; return WL_ECOMM; }
  *out = c; return 0;
}
int wl_comm_rccl_available(void) { return rccl().ok ? 1 : 0; }
int wl_comm_rccl_add_async(wl_comm* cc, const char uid[128]) {
  RcclComm* c = dynamic_cast<RcclComm*>(cc);
  WL_CHECK(c && !c->comm_async, "not an RCCL communicator (or it already has its async communicator)");
  ncclUniqueId id; memcpy(id.internal, uid, 128);
  ncclResult_t e = rccl().CommInitRank(&c->comm_async, c->size, id, c->rank);
  if (e != 0) { c->comm_async = nullptr; wl_set_error(std::string("ncclCommInitRank (async): ") + (rccl().GetErrorString ? rccl().GetErrorString(e) : "error")); return WL_ECOMM; }
  return 0;
}
int wl_comm_set_periodic(wl_comm* c, int on) { WL_CHECK(c, "null communicator"); c->zperiodic = on != 0; return 0; }
int wl_comm_set_virtual(wl_comm* cc, int rank, int size) {
  RcclComm* c = dynamic_cast<RcclComm*>(cc);
  WL_CHECK(c && c->size == 1 && !c->virt, "the rehearsal mode needs a one-rank RCCL communicator");
  WL_CHECK(size >= 2 && rank >= 0 && rank < size, "bad pretended rank/size");
  if (c->gather) { (void)hipFree(c->gather); c->gather = nullptr; }      // (sized for the pretended number of ranks at the next use)
  c->virt = true; c->real_rank = c->rank; c->real_size = c->size; c->rank = rank; c->size = size;
  return 0;
}
int wl_comm_set_virtual_transport(wl_comm* c, int on) { WL_CHECK(c && c->virt, "not a communicator in rehearsal mode"); c->virt_null = on == 0; return 0; }
int wl_comm_set_loopback(wl_comm* c, int on) { WL_CHECK(c && c->size == 1, "loopback is a one-rank test mode"); c->loopback = on != 0; return 0; }
int wl_comm_halo_async(wl_comm* c, float* a, const wl_grid* g, int ncomp, int depth, void* st) {
  WL_CHECK(g && g->D == 3, "halo exchange needs a 3-D slab grid");
  WL_TRY(wl::halo_async_begin(c, a, gx(*g), ncomp, depth, wl_stream(st)));
  return wl::halo_async_wait(c, wl_stream(st));
}
int wl_comm_combine_test(wl_comm* c, double* d8, float* f8, void* st) {
  // test hook: Σ over ranks of d8[0..7], max over ranks of f8[0..7] through the production combine path (device pointers, adjacent 64+32 B)
  WL_CHECK(c && d8 && (char*)f8 == (char*)d8 + 64, "d8/f8 must be one 128-byte record");
  RedWs ws{}; ws.res_d = d8; ws.res_f = f8;
  return wl::combine_results(c, ws, wl_stream(st));
}
int wl_comm_callbacks_create(wl_comm** out, int rank, int size, void* ctx, wl_sendrecv_fn sr, wl_allgather_fn ag) {
  WL_CHECK(out && sr && ag && size >= 1 && rank >= 0 && rank < size, "bad arguments");
  CallbackComm* c = new CallbackComm(); c->rank = rank; c->size = size; c->ctx = ctx; c->f_sr = sr; c->f_ag = ag;
  *out = c; return 0;
}
int wl_comm_destroy(wl_comm* c) { delete c; return 0; }
int wl_comm_rank(const wl_comm* c) { return c ? c->rank : 0; }
int wl_comm_size(const wl_comm* c) { return c ? c->size : 1; }
int wl_comm_stats(const wl_comm* c, int64_t out[4]) {
  out[0] = c ? c->n_halo : 0; out[1] = c ? c->halo_bytes : 0; out[2] = c ? c->n_combine : 0; out[3] = c ? c->n_gather : 0;
  return 0;
}
int wl_halo_exchange(wl_comm* c, float* a, const wl_grid* g, int ncomp, int depth, void* st) {
  WL_CHECK(g && g->D == 3, "halo exchange needs a 3-D slab grid");
  return wl::halo(c, a, gx(*g), ncomp, depth, wl_stream(st));
}
int wl_allgather_planes(wl_comm* c, float* a, const wl_grid* view, int ncomp, void* st) {
  WL_CHECK(view && view->D == 3, "needs a 3-D grid");
  return wl::allgather_planes(c, a, gx(*view), ncomp, wl_stream(st));
}
int wl_grid_slab(wl_grid* out, int D, const int32_t* gd, int rank, int size, int halo) {
  WL_CHECK(out && D == 3, "z-slab decomposition is 3-D only");
  const int nzi = gd[2] - 2;
  WL_CHECK(size >= 1 && rank >= 0 && rank < size && nzi % size == 0, "interior nz must be divisible by the number of ranks");
  const int nloc = nzi / size;
  WL_CHECK(halo >= 1 && nloc >= halo, "slab thinner than its halo");
  out->D = 3; out->nx = gd[0]; out->ny = gd[1];
  out->k0 = halo; out->k1 = halo + nloc; out->nz = nloc + 2 * halo;
  out->gk = 1 + rank * nloc - halo; out->gnz = gd[2];
  if (size == 1) { out->k0 = 1; out->k1 = gd[2] - 1; out->nz = gd[2]; out->gk = 0; }
  return 0;
}
}  // extern "C"
