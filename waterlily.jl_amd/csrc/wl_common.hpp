// Shared device/host helpers for libwlhip (gfx950 only; wave = 64).
#pragma once
#include <functional>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/wlhip.h"
#include "../../include/wlhip_bench.h"

#define WL_WAVE 64
#define WL_BLOCK 256

// ---- error plumbing ---------------------------------------------------------------------------
void wl_set_error(const std::string& s);
#define WL_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess) {                                                              \
      wl_set_error(std::string(#call) + ": " + hipGetErrorString(e__));                   \
      return (int)e__;                                                                    \
    }                                                                                     \
  } while (0)
#define WL_CHECK(cond, msg)                                                               \
  do {                                                                                    \
    if (!(cond)) { wl_set_error(std::string(msg) + " [" #cond "]"); return WL_EINVAL; }   \
  } while (0)
#define WL_TRY(call)                 \
  do {                               \
    int r__ = (call);                \
    if (r__ != 0) return r__;        \
  } while (0)
#define WL_LAUNCH_CHECK() WL_HIP(hipGetLastError())

static inline hipStream_t wl_stream(void* s) { return (hipStream_t)s; }
// every kernel launch of the library is counted (wl_launch_count(): bench.py's config.launches_per_step)
extern long g_wl_launches;
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) do { ++g_wl_launches; kern<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__); } while (0)

// Experiment switches (tile heights, chunk lengths, block orders … — tools/*.sh scan them): read from the environment only in a
// -DWL_EXPERIMENTS build (`tools/build_variant.sh exp "-DWL_EXPERIMENTS" <all sources>`); the product library always takes the default.
static inline int wl_exp_int(const char* name, int dflt) {
#ifdef WL_EXPERIMENTS
  const char* e = getenv(name); return e ? atoi(e) : dflt;
#else
  (void)name; return dflt;
#endif
}
// ---- grid helpers (host + device) -------------------------------------------------------------
struct GridX {  // wl_grid + precomputed strides, passed by value to kernels
  int D, nx, ny, nz, k0, k1, gk, gnz;
  long sy, sz, cs;  // row, plane and component strides (elements)
};
static inline GridX gx(const wl_grid& g) {
  GridX x;
  x.D = g.D; x.nx = g.nx; x.ny = g.ny; x.nz = g.nz; x.k0 = g.k0; x.k1 = g.k1; x.gk = g.gk; x.gnz = g.gnz;
  x.sy = g.nx; x.sz = (long)g.nx * g.ny; x.cs = x.sz * g.nz;
  return x;
}
static inline int wl_grid_ok(const wl_grid* g) {
  if (!g) return 0;
  if (g->D == 2) return g->nx >= 3 && g->ny >= 3 && g->nz == 1 && g->k0 == 0 && g->k1 == 1;
  if (g->D == 3) return g->nx >= 3 && g->ny >= 3 && g->nz >= 3 && g->k0 >= 1 && g->k1 > g->k0 && g->k1 <= g->nz - 1 && g->gnz >= g->nz - 2 * (g->k0 - 1);
  return 0;
}
static inline long wl_ncell(const wl_grid& g) { return (long)g.nx * g.ny * g.nz; }
// number of interior cells owned by this rank
static inline long wl_ninside_local(const wl_grid& g) { return (long)(g.nx - 2) * (g.ny - 2) * (g.D == 3 ? (g.k1 - g.k0) : 1); }
static inline long wl_ninside_global(const wl_grid& g) { return (long)(g.nx - 2) * (g.ny - 2) * (g.D == 3 ? (g.gnz - 2) : 1); }

// Launch geometry: 1-D grid; threads run linearly over an x-y plane (perfectly coalesced, ghosts masked).
// XCD-aware block->tile map: the 8 XCDs of MI355X have private L2s and workgroups are dealt to them
// round-robin by linear block id (b and b+8 share an XCD).  Hardware block h therefore takes strip q = h%8 of
// the plane (a contiguous 1/8 of its rows) and walks that strip plane by plane, so the ±y and ±z stencil
// neighbours of a cell are fetched by the SAME XCD's L2 instead of being re-fetched by up to three of them.
static inline int wl_strip_blocks(const GridX& g) { const long nbx = (g.sz + WL_BLOCK - 1) / WL_BLOCK; return (int)((nbx + 7) >> 3); }
static inline dim3 wl_plane_grid(const GridX& g, int nplanes) { return dim3((unsigned)(8L * wl_strip_blocks(g) * nplanes), 1, 1); }
// plane-slot count for grid-stride reduction kernels: keeps the number of per-block partials <= WL_REDPART
#define WL_REDPART 8192
static inline int wl_red_slots(const GridX& g, int nplanes) { long by = WL_REDPART / (8L * wl_strip_blocks(g)); if (by < 1) by = 1; return (int)(by < nplanes ? by : nplanes); }

// z-marching kernels: every plane slot owns a CONTIGUOUS chunk of planes (z-neighbours stay in registers).  Chunk length:
// up to 32 planes, shorter on small grids so that a launch still has a few thousand workgroups.
static inline int wl_march_chunk(const GridX& g, int nplanes) {
  const long bp = 8L * wl_strip_blocks(g);
  static const long cap = [] { const long v = wl_exp_int("WL_MARCH_CHUNK_CAP", 32); return v >= 1 ? v : 32; }();   // experiments only
  long c = (long)nplanes * bp / 4096; if (c > cap) c = cap; if (c < 1) c = 1;
  while ((nplanes + c - 1) / c * bp > 65536 && c < nplanes) c++;      // per-workgroup partials must fit the reduction workspace
  return (int)c;
}
static inline int wl_march_slots(int nplanes, int chunk) { return (nplanes + chunk - 1) / chunk; }

#ifdef __HIPCC__
// block -> (flattened in-plane index m of this thread, plane slot p); false when the block lies beyond the plane
__device__ __forceinline__ bool wl_tile(const GridX& g, long& m, int& p) {
  const unsigned h = blockIdx.x;
  const unsigned q = h & 7u, s = h >> 3;
  const long nbx = (g.sz + WL_BLOCK - 1) / WL_BLOCK;
  const unsigned per = (unsigned)((nbx + 7) >> 3);
  p = (int)(s / per);
  const long bx = (long)q * per + (s - (unsigned)p * per);
  m = bx * WL_BLOCK + threadIdx.x;
  return bx < nbx;
}
// The same grid read in LINEAR order: block h takes chunk h mod nb8 of plane slot h / nb8 (nb8 = the plane's 256-cell chunks rounded up to
// a multiple of 8, so that a chunk position belongs to the same XCD on every plane: its z-neighbours are hits in that XCD's L2).  With one
// plane per slot the chip sweeps the arrays front to back like an element-wise kernel — the order in which HBM delivers the most
// (profiles/r03_shape_probe.md: 5.5 TB/s against 4.6 for z-marching chunks of the projection tail's 4 reads + 4 writes).
__device__ __forceinline__ bool wl_tile_lin(const GridX& g, long& m, int& p) {
  const long nbx = (g.sz + WL_BLOCK - 1) / WL_BLOCK;
  const unsigned nb8 = (unsigned)(((nbx + 7) >> 3) << 3);
  const unsigned h = blockIdx.x;
  p = (int)(h / nb8);
  const long bx = (long)(h - (unsigned)p * nb8);
  m = bx * WL_BLOCK + threadIdx.x;
  return bx < nbx;
}
__device__ __forceinline__ int wl_nslots(const GridX& g) {   // number of plane slots of this launch
  const long nbx = (g.sz + WL_BLOCK - 1) / WL_BLOCK;
  return (int)(gridDim.x / (8u * (unsigned)((nbx + 7) >> 3)));
}
// ---- wave / block reductions (wave64 shuffles, then LDS across the 4 waves of a 256-thread block) ----
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  return v;
}
// result valid in thread 0
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double sh[WL_BLOCK / WL_WAVE];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) { v = sh[0]; for (int i = 1; i < WL_BLOCK / WL_WAVE; i++) v += sh[i]; }
  return v;
}
__device__ __forceinline__ float block_max(float v) {
  __shared__ float shm[WL_BLOCK / WL_WAVE];
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) shm[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) { v = shm[0]; for (int i = 1; i < WL_BLOCK / WL_WAVE; i++) v = fmaxf(v, shm[i]); }
  return v;
}
#endif

// ---- reduction workspace layout (device) --------------------------------------------------------
// [0 .. WL_MAXPART)              double partial sums  A
// [WL_MAXPART .. 2*WL_MAXPART)   double partial sums  B
// then WL_MAXPART floats         partial max
// then 8 doubles + 8 floats      results  (res_d[0..7], res_f[0..7])
#define WL_MAXPART 65536
struct RedWs {
  double* pa; double* pb; float* pm; double* res_d; float* res_f;
};
static inline size_t wl_red_bytes() { return (size_t)WL_MAXPART * (8 + 8 + 4) + 8 * 8 + 8 * 4 + 64; }
static inline RedWs wl_red_ws(void* base) {
  RedWs w; char* b = (char*)base;
  w.pa = (double*)b; w.pb = (double*)(b + (size_t)WL_MAXPART * 8); w.pm = (float*)(b + (size_t)WL_MAXPART * 16);
  w.res_d = (double*)(b + (size_t)WL_MAXPART * 20); w.res_f = (float*)(b + (size_t)WL_MAXPART * 20 + 64);
  return w;
}

// process-wide context: pinned host scalars + a default reduction workspace
struct WlCtx {
  bool inited = false;
  int device = 0;
  void* red = nullptr;          // device reduction workspace
  double* h_d = nullptr;        // pinned host: 8 doubles
  float* h_f = nullptr;         // pinned host: 8 floats
};
WlCtx& wl_ctx();
int wl_ctx_ensure();

// ---- measurement hooks: event pairs on the launch stream (see wlhip.h WL_PROF_*) ----------------------
struct WlProf {
  bool on = false;
  bool only_roofline = false;   // wl_prof_enable(2): event pairs only around the finest level's smoother kernels A and B
  struct Slot { std::vector<hipEvent_t> a, b; size_t used = 0; };
  Slot slot[WL_PROF_NSLOTS];
};
WlProf& wl_prof();
struct ProfScope {   // RAII: records start at construction and stop at destruction when profiling is enabled
  int id; hipStream_t s; bool active; size_t idx;
  ProfScope(int id_, hipStream_t s_);
  ~ProfScope();
};

// BC!(u,U) for a tuple U folded into a producer's stores (wl_bcfold.hpp): request (on = 1, U) / report (on = 1 if it was applied)
struct BcFold { int on; float U[3];
                // in: x of a projection whose velocity update was deferred to this conv_diff! launch (wl_convf.hip, PROJ); out: 1 = the launch applied it
                const float* proj_x = nullptr; int proj_done = 0;
                // in: BC!(u_in,U) was deferred — the producer of u_in wrote the interior only (wl_sim, mom_step!): the tail reads the wall-normal boundary faces as U
                int usub = 0;
                // in: device flag — the tail kernel does nothing unless *go != 0 (the solver's convergence decision taken on the device: wl::decide_converged; the tail is
                // queued behind the V-cycle before the host has read the norms)
                const float* go = nullptr;
                // in (conv_diff!+BDIM! launches): Δt is read from this device location instead of the argument (wl_sim_mom_steps: the next step's predictor is queued
                // before the host has read the CFL maximum); honoured by the flux-once tiled kernel only
                const float* dt_dev = nullptr; };

// ---- kernel launchers shared between the leaf C ABI and the composite handles --------------------
namespace wl {
// A level whose face coefficients are verified (on device, exact comparison) to be "c[a] inside, 0 on the wall faces":
// the NoBody hierarchy.  Poisson-side kernels then evaluate L, D, iD instead of loading them (identical bits).
// Dt/iDt: D and iD of a cell as a function of how many of its two faces per direction are non-wall (index nx+3ny+9nz),
// computed on the host in set_diag!'s operation order (IEEE: same bits as the device would produce).
struct ConstL { int on; float c[3]; float Dt[27]; float iDt[27]; };
#ifdef __HIPCC__
__device__ __forceinline__ float wl_cl_coef(int Ia, int Na, float c) { return (Ia <= 2 || Ia >= Na) ? 0.f : c; }
// number of non-wall faces (lower + upper) of the cell with Julia index Ia along a direction of extent Na
__device__ __forceinline__ int wl_cl_cnt(int Ia, int Na) { return ((Ia <= 2 || Ia >= Na) ? 0 : 1) + ((Ia + 1 <= 2 || Ia + 1 >= Na) ? 0 : 1); }
#endif
int check_const_L(const float* L, const GridX& g, ConstL* out, int* dev_flag, hipStream_t s);
int const_plane_range(const float* L, const GridX& g, const float* c, int* za, int* zb, hipStream_t s);
int fill(float* a, float v, size_t n, hipStream_t s);
int scale(float* a, float s_, size_t n, hipStream_t s);
int div_scalar(float* a, float s_, size_t n, hipStream_t s);
int div_scalar_to(float* out, const float* in, float s_, size_t n, hipStream_t s);
// reductions leave results in ws.res_d / ws.res_f on device; *_host variants copy to host & sync
int sum_dev(const float* a, size_t n, const RedWs& ws, int slot, hipStream_t s);
int l1_linf_dev(const float* a, size_t n, const RedWs& ws, int slot_d, int slot_f, hipStream_t s);
int max_dev(const float* a, size_t n, const RedWs& ws, int slot_f, hipStream_t s);
int dot_dev(const float* a, const float* b, size_t n, const RedWs& ws, int slot, hipStream_t s);
int read_results(const RedWs& ws, double* hd, int nd, float* hf, int nf, hipStream_t s);
int read_results_overlapped(const RedWs& ws, double* hd, int nd, float* hf, int nf, hipStream_t s, hipEvent_t copied, const std::function<int()>& queue_behind);   // work queued behind the copy before the host waits for the copy alone
std::mutex& wl_read_mutex();   // guards the process-wide pinned staging scalars of WlCtx

int bc_vec(float* a, const GridX& g, const float* U, int saveexit, unsigned per, hipStream_t s);
int bc_per_scalar(float* a, const GridX& g, unsigned per, hipStream_t s);
int conv_diff(float* r, const float* u, float* Phi, const GridX& g, float nu, unsigned per, int scheme, hipStream_t s);
int conv_diff_bdim(float* f, const float* u_adv, float* Phi, const float* u0, const float* mu0, float* u_out, const GridX& g, float nu, unsigned per, int scheme,
                   float dt, float pre, float post, const ConstL& cl, hipStream_t s, int ka = -(1 << 30), int kb = 1 << 30, bool q1 = true, BcFold* fold = nullptr);   // [ka,kb): plane sub-range; q1: also the Φ ghost pass; fold: BC!(u_out,U) folded into the stores where possible (in/out)
int conv_q1(float* Phi, const float* u, const GridX& g, float nu, unsigned per, int scheme, hipStream_t s);
bool conv_z_ok(const GridX& g, unsigned per);
int conv_diff_z(float* f, const float* u_adv, const float* u0, const float* mu0, float* u_out, const GridX& g, float nu, int scheme, float dt, float pre, float post, hipStream_t s);
int bdim(float* u, const float* u0, float* f, const float* V, const float* mu0, const float* mu1, const GridX& g, float dt, float pre, float post, hipStream_t s);
int accelerate(float* r, const GridX& g, const float* a, hipStream_t s);
int bc_vec_fn(float* a, const float* Ub, const GridX& g, int saveexit, unsigned per, hipStream_t s);
int add_field(float* r, const float* gfield, size_t n, hipStream_t s);
int meanflow_update(float* P, float* U, float* UU, const float* p, const float* u, const GridX& g, float e, hipStream_t s);
int meanflow_uu(float* tau, const float* UU, const float* U, const GridX& g, hipStream_t s);
// z-marching LDS-tiled conv_diff!+BDIM! (wl_convt.hip): the default fused path on grids that fill the chip
int bc_zplanes(float* u, const GridX& g, float U2, hipStream_t s);
void conv_tile_enable(int on, int chunk);
void conv_tile_min(long tile_planes);
bool conv_tile_ok(const GridX& g, unsigned per, int nplanes);
int conv_tile(const float* u_adv, const GridX& g, float nu, int scheme, int ka, int kb, const void* bdim_args, hipStream_t s);
// the flux-once form of the tiled kernel (wl_convf.hip); conv_tile dispatches to it unless conv_flux_enable(0)
void conv_flux_enable(int on);
bool conv_flux_on();
bool conv_proj_ok(const GridX& g, unsigned per);   // geometry of the fused projection: whole tiles, whole single domain
int conv_flux(const float* u_adv, const GridX& g, float nu, int scheme, int ka, int kb, int zchunk, const void* bdim_args, hipStream_t s);
void conv_march_enable(int on);
void jacobi_march_enable(int on);
bool conv_march_ok(const GridX& g);
int conv_march(float* r, const float* u, const GridX& g, float nu, unsigned per, int scheme, int kfirst, int klast, const void* bdim_args, hipStream_t s);
int bdim_f(float* f, const float* u0, const float* V, const GridX& g, float dt, hipStream_t s);
int bdim_u(float* u, const float* f, const float* V, const float* mu0, const float* mu1, const GridX& g, float pre, float post, hipStream_t s, const unsigned char* far = nullptr);
size_t body_mask_bytes(const GridX& g);
int conv_diff_bdim_body(float* f, const float* u_adv, float* Phi, const float* u0, const float* mu0, float* u_out, const GridX& g, float nu, unsigned per, int scheme,
                        float dt, float pre, float post, const unsigned char* near, const unsigned char* needf, const unsigned char* m0var, int nbm, int store_all, hipStream_t s,
                        int dz0 = 0, int dz1 = -1);   // dz0..dz1: body_masks_planes (planes outside run the tiled NoBody kernel)
int body_masks_planes(const unsigned char* near, const unsigned char* needf, const unsigned char* m0var, const GridX& g, int* dz, hipStream_t s);
void conv_body_tile_enable(int on);
int body_masks_nbm(const GridX& g);
int body_masks(unsigned char* near, unsigned char* needf, unsigned char* m0var, const float* V, const float* mu1, const float* mu0, const GridX& g, hipStream_t s);
int bdim_near(float* uout, const float* uin, const float* u0, float* f, const float* V, const float* mu0, const float* mu1, const GridX& g, float dt, float pre, float post,
              const unsigned char* near, int nbm, const int* box, hipStream_t s);
int body_masks_box(const unsigned char* near, const GridX& g, int* box, hipStream_t s);
int body_mask(unsigned char* far, const float* V, const float* mu1, const GridX& g, hipStream_t s);
int scale_u(float* u, const GridX& g, float sc, hipStream_t s);
int div(float* z, const float* u, const GridX& g, hipStream_t s);
int div_scale(float* z, float* x, const float* u, const GridX& g, float dt, hipStream_t s);   // z=div(u); x*=dt  (fused, src/Flow.jl:225)
int project(float* u, const float* L, const float* x, const GridX& g, hipStream_t s);
int cfl_dev(const float* u, float* sigma, const GridX& g, const RedWs& ws, int slot_f, hipStream_t s);

int set_diag(float* D, float* iD, const float* L, const GridX& g, hipStream_t s);
int mult(float* z, const float* L, const float* D, const float* x, const GridX& g, hipStream_t s);
int residual(float* r, const float* x, const float* z, const float* L, const float* D, const float* iD, const GridX& g, const RedWs& ws, hipStream_t s);
int residual_part(float* r, const float* x, const float* z, const float* L, const float* D, const float* iD, const GridX& g, const RedWs& ws, hipStream_t s);
int mean_shift(float* r, const GridX& g, const RedWs& ws, hipStream_t s);
int div_residual(float* z, float* xout, float* r, const float* x, const float* u, const float* L, const float* D, const float* iD, const GridX& g, float dt, const RedWs& ws, const ConstL& cl, hipStream_t s);
int div_residual_split(float* z, float* xout, float* r, const float* x, const float* u, const float* L, const float* D, const float* iD, const GridX& g, float dt, const RedWs& ws,
                       const ConstL& near, const ConstL& far, int na, int nb, hipStream_t s);
// fused head of mom_project! + the V-cycle's first Jacobi! on the finest level (wl_resjac.hip)
void resjac_enable(int on, long min_cells);
bool par_streams_ok(); hipStream_t par_stream(int i); int par_fork(hipStream_t s); int par_join(hipStream_t s);   // wl_capi.hip
bool resjac_ok(const GridX& g, const ConstL& cl);
int resjac(float* xout, float* rout, const float* x, const float* u, const GridX& g, float dt, float w, const ConstL& cl, const RedWs& ws, int slot_d, int slot_f, hipStream_t s,
           bool shell = true, const float* bcU = nullptr);   // shell = false: x's (and x_out's) ghost cells are known to be +0 — the ghost-shell scaling pass is skipped
int shell_nonzero(const float* a, const GridX& g, int* dev_flag, hipStream_t s);
int project_unscale(float* u, const float* L, const float* x, float* pout, const GridX& g, float dt, const ConstL& cl, hipStream_t s, const BcFold* fold = nullptr);
int decide_converged(const RedWs& ws, double r1tol, double rinftol, double ninside, int check_head, int slot_d, int slot_f, int out_slot, hipStream_t s);   // res_f[out_slot] = 1/0: solver!'s break test (and the fused head's mean-shift test) on the device
bool project_cfl_pair_path(const GridX& g, const ConstL& cl);   // the two-cells-per-thread tail will run (the form that honours BcFold::usub)
int project_cfl(float* uout, const float* uin, const float* L, const float* x, float* pout, float* sigma, const GridX& g, float dt, const ConstL& cl, const RedWs& ws, int slot_f, hipStream_t s, int store_sigma = 1, const BcFold* fold = nullptr);
int project_unscale_split(float* u, const float* L, const float* x, float* pout, const GridX& g, float dt, const ConstL& near, const ConstL& far, int na, int nb, hipStream_t s);
int project_cfl_split(float* uout, const float* uin, const float* L, const float* x, float* pout, float* sigma, const GridX& g, float dt, const ConstL& near, const ConstL& far,
                      int na, int nb, const RedWs& ws, int slot_f, hipStream_t s, int store_sigma = 1);
// computes L₁/L∞ of r into ws.res_d[slot_d], ws.res_f[slot_f] (device) — ghosts of r are zero by construction
int norms_dev(const float* r, const GridX& g, const RedWs& ws, int slot_d, int slot_f, hipStream_t s);
int increment(float* r, float* x, const float* eps, const float* L, const float* D, const GridX& g, float w, hipStream_t s);
int jacobi(float* eps, float* r, float* x, const float* L, const float* D, const float* iD, const GridX& g, float w, bool write_eps, hipStream_t s);
int gs_init(float* eps, const float* r, const float* iD, const GridX& g, hipStream_t s);
int gs_sweep(float* eps, const float* r, const float* L, const float* iD, const GridX& g, int k0, hipStream_t s);
int gs_init_sweep1(float* eps, const float* r, const float* L, const float* iD, const GridX& g, hipStream_t s);
int jacobi_pp(float* rout, const float* r, float* x, const float* L, const float* D, const float* iD, const GridX& g, float w, const ConstL& cl, hipStream_t s, int xzero = 0);
bool jacobi_takes_shift(const GridX& g, const ConstL& cl);
int jacobi_pp_shift(float* rout, const float* r, float* x, const GridX& g, float w, const ConstL& cl, const RedWs& ws, int slot_d, int slot_f, hipStream_t s);
int shift_norms_dev(float* r, const GridX& g, const RedWs& ws, int slot_d, int slot_f, hipStream_t s);
// coarse tail of the V-cycle in one launch (wl_poisson.hip)
#define WL_TAIL_MAXLV 8
#define WL_TAIL_CELLS 8192
struct TailLevelHost { GridX g; const float* L; const float* D; const float* iD; float* x; float* eps; float* r; int cx, cy, cz; const ConstL* cl = nullptr; };
void tail_lds_enable(int on);   // 1 (default): the tail keeps r, x, ϵ of its levels in LDS; 0: the global-memory tail
int vcycle_tail(const TailLevelHost* lv, int n, float w, hipStream_t s);
int pcg_stage(int stage, float* eps, float* r, float* x, float* z, const float* L, const float* Dg, const float* iD, const GridX& g, float a, int more, const RedWs& ws, hipStream_t s);
bool gsrb_fused_ok(const GridX& g, unsigned per, bool dist);
int gsrb_fused_A(float* emid, const float* r, const float* L, const GridX& g, const ConstL& cl, hipStream_t s);
int gsrb_fused_A_pro(float* emid, float* rnew, float* x, const float* r, const float* xc, const float* L, const GridX& g, const GridX& gc, float w, const ConstL& cl, hipStream_t s,
                     int xk0 = -(1 << 30), int xk1 = 1 << 30, bool* defer_x = nullptr, bool range = false);   // defer_x (in/out): leave `x += ω·x_c↓` to kernel B (cleared if this path cannot); range: g is a short plane sub-range of a qualifying level
// The V-cycle's `x += ω·x_c↓` handed from kernel A to kernel B of the same smooth! (pair kernels only): A leaves x alone, B applies both
// increments of x in order — x is read and written once per smooth! instead of twice.
struct XDefer { const float* xc; GridX gc; float w; };
bool gsrb_pair_B_ok(const float* eps, const float* rout, const float* x, const float* emid, const float* r, const GridX& g, const ConstL& cl);
int gsrb_fused_B(float* eps, float* rout, float* x, const float* emid, const float* r, const float* L, const GridX& g, float w,
                 const RedWs* ws, int slot_d, int slot_f, const ConstL& cl, hipStream_t s, const XDefer* xd = nullptr);   // xd: only when gsrb_pair_B_ok
int finalize_sum_max(const RedWs& ws, int nparts, int slot_d, int slot_f, hipStream_t s);
// pair variant of the blocked smoother for constant-coefficient levels (wl_fused2.hip); chosen inside gsrb_fused_* when eligible
void gsrb_pair_enable(int on);
bool gsrb_pair_ok(const GridX& g, const ConstL& cl);
bool gsrb_pair_ok_range(const GridX& g, const ConstL& cl);
bool gsrb_pair_geom_ok(const GridX& g);
int gsrb_pair_A(float* emid, const float* r, const GridX& g, const ConstL& cl, hipStream_t s);
int gsrb_pair_A_pro(float* emid, float* rnew, float* x, const float* r, const float* xc, const GridX& g, const GridX& gc, float w, const ConstL& cl, hipStream_t s,
                    int xk0 = -(1 << 30), int xk1 = 1 << 30);   // [xk0,xk1): planes on which x is updated (default: every output plane)
int gsrb_pair_B(float* eps, float* rout, float* x, const float* emid, const float* r, const GridX& g, float w,
                const RedWs* ws, int slot_d, int slot_f, const ConstL& cl, hipStream_t s, const XDefer* xd = nullptr);
int restrict_(float* a, const GridX& gc, const float* b, const GridX& gf, hipStream_t s);
int prolongate(float* a, const GridX& gf, const float* b, const GridX& gc, hipStream_t s);
int prolong_increment(float* r, float* x, float* eps, const float* xc, const float* L, const float* D, const GridX& gf, const GridX& gc, float w, bool write_eps, hipStream_t s);
int restrictL(float* a, const GridX& gc, const float* b, const GridX& gf, unsigned per, hipStream_t s);
}  // namespace wl
