// z-slab communication layer (see include/wlhip.h "multi-GPU").
#pragma once
#include "wl_common.hpp"

struct wl_comm {
  int rank = 0, size = 1;
  bool zperiodic = false;   // the domain is periodic in z: rank 0's lower neighbour is rank size-1 and vice versa (halo exchanges wrap around)
  bool loopback = false;    // one-rank TEST mode: both neighbours are this rank (exercises the transport calls that size==1 skips)
  // one-rank REHEARSAL mode (wl_comm_set_virtual): rank/size above are those of a pretended P-rank run — slab geometry, wall logic and the exchange
  // pattern follow them — while the transport underneath has one rank: planes sent to a neighbour come back as this rank's own ghost planes.
  // What one GPU can measure of a P-GPU step: the rank's compute, the number and size of its exchanges, the issue cost of the RCCL calls.
  bool virt = false; int real_rank = 0, real_size = 1;
  bool virt_null = false;   // rehearsal without any transfer (wl_comm_set_virtual_transport(c,0)): exchanges and all-gathers return at once — the rank's pure compute
  void* gather = nullptr;   // device scratch for scalar all-gathers: size * 128 bytes
  virtual ~wl_comm();
  // lo neighbour = rank-1, hi neighbour = rank+1; pointers are NULL where there is no neighbour
  virtual int sendrecv(const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes, hipStream_t s) = 0;
  virtual int allgather(const void* send, void* recv, size_t bytes_each, hipStream_t s) = 0;
  // exchanges issued between group_begin and group_end form ONE network round (RCCL: one ncclGroup); n_halo counts rounds
  int gdepth = 0;
  virtual int group_begin() { gdepth++; return 0; }
  virtual int group_end() { if (gdepth > 0) gdepth--; return 0; }
  int ensure_scratch();
  // counters since creation (wl_comm_stats): halo exchanges, bytes this rank sent in them, scalar combines, plane all-gathers
  long n_halo = 0, halo_bytes = 0, n_combine = 0, n_gather = 0;
  // second HIP stream for halo exchanges that overlap interior stencil work (+ the two events that order it with the compute stream)
  hipStream_t cs = nullptr; hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  int ensure_async();
};

namespace wl {
// exchange `depth` planes of an ncomp-component field along z with both neighbours
// wrap: on a z-periodic domain also exchange across the periodic boundary (rank 0 <-> rank size-1).  The reference refreshes periodic ghost cells
// only where it calls BC!/perBC! — an exchange that stands for such a call wraps; one that merely keeps neighbouring slabs coupled between two
// colour sweeps (the single domain reads live interior cells there, but STALE ghost cells at the periodic boundary) must not.
int halo(wl_comm* c, float* a, const GridX& g, int ncomp, int depth, hipStream_t s, bool wrap = true);
// the same exchange on the communicator's own stream: it starts when everything queued on `compute` so far is done (begin) and
// `compute` waits for it only where the caller says so (wait) — kernels launched in between overlap with the transfer
int halo_async_begin(wl_comm* c, float* a, const GridX& g, int ncomp, int depth, hipStream_t compute);
int halo_async_wait(wl_comm* c, hipStream_t compute);
// ws.res_d[0..7] <- Σ over ranks, ws.res_f[0..7] <- max over ranks (on device, stream ordered); no-op without comm
int combine_results(wl_comm* c, const RedWs& ws, hipStream_t s);
// in-place all-gather of the owned planes [k0,k1) of a replicated (full) array whose rank blocks are contiguous
int allgather_planes(wl_comm* c, float* a, const GridX& view, int ncomp, hipStream_t s);
}  // namespace wl
