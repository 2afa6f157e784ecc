// Simulation/Flow composite handle: mom_step! orchestration (src/Flow.jl:156-237) on one HIP stream,
// plus the device-side input generators / read-outs used by the configs (TGV initial condition,
// closed-form sphere measure!, pressure_force) and exitBC!, L₂.
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "wl_common.hpp"
#include <algorithm>
#include "wl_mg.hpp"

namespace {
__device__ __forceinline__ bool cell_ij(const GridX& g, long m, int& i, int& j) {
  if (m >= g.sz) return false;
  j = (int)(m / g.nx);
  i = (int)(m - (long)j * g.nx);
  return true;
}
__device__ __forceinline__ bool interior_ij(const GridX& g, int i, int j) { return i >= 1 && i <= g.nx - 2 && j >= 1 && j <= g.ny - 2; }

// apply!(u0,u): u[I,i] = u0(i, loc(i,I))   src/Flow.jl:81-83, loc src/core.jl:177 (Float32)
// kind 1: wall-bounded TGV κ=π/N ; kind 2: periodic TGV κ=2π/N  
template <int D>
__global__ void k_apply_tgv(GridX g, float* __restrict__ u, float kx, float ky, float kz) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j)) return;
  const int k = pz;
  const long o = m + (long)k * g.sz;
  const int I[3] = {i + 1, j + 1, (D == 3) ? g.gk + k + 1 : 1};
  for (int a = 0; a < D; a++) {
    float x[3];
    for (int c = 0; c < 3; c++) x[c] = (float)I[c] - 1.5f - ((c == a) ? 1.f : 0.f) / 2.f;
    const double X = (double)x[0] * kx, Y = (double)x[1] * ky, Z = (D == 3) ? (double)x[2] * kz : 0.0;
    double v;
    if (a == 0) v = -sin(X) * cos(Y) * ((D == 3) ? cos(Z) : 1.0);
    else if (a == 1) v = cos(X) * sin(Y) * ((D == 3) ? cos(Z) : 1.0);
    else v = 0.0;
    u[(long)a * g.cs + o] = (float)v;
  }
}
template <int D>
__global__ void k_apply_const(GridX g, float* __restrict__ u, float U0, float U1, float U2) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j)) return;
  const long o = m + (long)pz * g.sz;
  u[o] = U0; u[g.cs + o] = U1; if (D == 3) u[2 * g.cs + o] = U2;
}

// BDIM kernel moments   src/Body.jl:54-60
__device__ __forceinline__ float kern_(float d) { return (1 + cosf(3.14159265358979323846f * d)) / 2; }
__device__ __forceinline__ float kern0_(float d) { return (1 + d + sinf(3.14159265358979323846f * d) / 3.14159265358979323846f) / 2; }
__device__ __forceinline__ float kern1_(float d) { return (1 - d * d) / 4 - (d * sinf(3.14159265358979323846f * d) + (1 + cosf(3.14159265358979323846f * d)) / 3.14159265358979323846f) / (2 * 3.14159265358979323846f); }
__device__ __forceinline__ float eps_at(float d) { d = fabsf(d); return d == 0.f ? 1.4e-45f : nextafterf(d, INFINITY) - d; }
__device__ __forceinline__ float mu0_(float d, float e) { return d / e < -1 + sqrtf(eps_at(d)) ? 0.f : kern0_(fminf(d / e, 1.f)); }
__device__ __forceinline__ float mu1_(float d, float e) { return e * kern1_(fminf(fmaxf(d / e, -1.f), 1.f)); }
// Closed-form AutoBody (src/AutoBody.jl:21,29-37): kind 1 sdf = |m∘(x−c)|−R (sphere/circle; an axis with m=0 is dropped: cylinder
// along it), kind 2 sdf = m·(x−c) (plane, m need not be unit).  The map x−V·t is folded into c by the caller, V is the body velocity.
struct BodyArg { int kind; float c[3], R, m[3], V[3]; };
template <int D>
__device__ __forceinline__ float body_sdf(const BodyArg& b, const float* x) {
  float s = 0.f;
  if (b.kind == 2) { for (int q = 0; q < D; q++) s += b.m[q] * (x[q] - b.c[q]); return s; }
  for (int q = 0; q < D; q++) { const float dx = b.m[q] * (x[q] - b.c[q]); s += dx * dx; }
  return sqrtf(s) - b.R;
}
// measure(body,x;fastd²): returns true when n was evaluated (then V = the body velocity), false on the early exits (n = V = 0)
template <int D>
__device__ __forceinline__ bool body_measure(const BodyArg& b, const float* x, float fastd2, float& d, float* n) {
  float rr = 0.f;
  for (int q = 0; q < D; q++) n[q] = 0.f;
  if (b.kind == 2) d = body_sdf<D>(b, x);
  else { float s = 0.f; for (int q = 0; q < D; q++) { const float dx = b.m[q] * (x[q] - b.c[q]); s += dx * dx; } rr = sqrtf(s); d = rr - b.R; }
  if (d * d > fastd2) return false;
  float gq[3]; bool nan = false;
  for (int q = 0; q < D; q++) { gq[q] = b.kind == 2 ? b.m[q] : (b.m[q] * (x[q] - b.c[q])) / rr; nan = nan || isnan(gq[q]); }
  if (nan) return false;
  float mm = 0.f; for (int q = 0; q < D; q++) mm += gq[q] * gq[q];
  mm = sqrtf(mm); d /= mm;
  for (int q = 0; q < D; q++) n[q] = gq[q] / mm;
  return true;
}
// measure!(flow,body;ϵ) for the sphere: fills σ(sdf), μ₀, μ₁, V(=0) on the interior   src/Body.jl:28-48
template <int D>
__global__ void k_measure_body(GridX g, float* __restrict__ sig, float* __restrict__ mu0, float* __restrict__ mu1, float* __restrict__ V, BodyArg bd, float e) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const int k = g.k0 + pz;
  const long o = m + (long)k * g.sz;
  const int I[3] = {i + 1, j + 1, (D == 3) ? g.gk + k + 1 : 1};
  float x[3]; for (int q = 0; q < 3; q++) x[q] = (float)I[q] - 1.5f;
  const float dc = body_sdf<D>(bd, x);
  sig[o] = dc;
  const float d2 = (2 + e) * (2 + e);
  if (dc * dc < d2) {
    for (int a = 0; a < D; a++) {
      float xf[3]; for (int q = 0; q < 3; q++) xf[q] = x[q] - ((q == a) ? 0.5f : 0.f);
      float di, ni[3]; const bool full = body_measure<D>(bd, xf, d2, di, ni);
      di = fabsf(di) <= 0.5f ? di : copysignf(di, dc);
      if (full && bd.V[a] != 0.f) V[(long)a * g.cs + o] = bd.V[a];       // (V was zero-filled: Body.jl:29)
      mu0[(long)a * g.cs + o] = mu0_(di, e);
      for (int b = 0; b < D; b++) mu1[(long)(a + b * D) * g.cs + o] = mu1_(di, e) * ni[b];
    }
  } else if (dc < 0.f) {
    for (int a = 0; a < D; a++) mu0[(long)a * g.cs + o] = 0.f;
  }
}
// cross(a,b) as the reference's broadcast stores it: the 3-D vector product, in 2-D the scalar a₁b₂−a₂b₁ in every component
template <int D>
__device__ __forceinline__ void cross_(const float* a, const float* b, float* o) {
  if (D == 2) { const float m = a[0] * b[1] - a[1] * b[0]; o[0] = m; o[1] = m; o[2] = 0.f; }
  else { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
}
struct MomArg { int on; float x0[3]; };      // on: moments about x0 (pressure_moment / viscous_moment, src/Metrics.jl:169-188) instead of forces
// pressure_force: Σ_inside p[I]·n·kern(clamp(d,-1,1)) in Float64   src/Metrics.jl:116-133
template <int D>
__global__ void k_pforce_body(GridX g, const float* __restrict__ p, BodyArg bd, MomArg mo, double* __restrict__ part) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  double acc[3] = {0, 0, 0};
  const int nsl = wl_nslots(g);
  if (cell_ij(g, m, i, j) && interior_ij(g, i, j)) {
    for (int k = g.k0 + pz; k < g.k1; k += nsl) {
      const int I[3] = {i + 1, j + 1, (D == 3) ? g.gk + k + 1 : 1};
      float x[3]; for (int q = 0; q < 3; q++) x[q] = (float)I[q] - 1.5f;
      float d, n[3]; body_measure<D>(bd, x, 1.f, d, n);
      const float kk = kern_(fminf(fmaxf(d, -1.f), 1.f));
      const float pv = p[m + (long)k * g.sz];
      if (mo.on) {
        float nds[3] = {0.f, 0.f, 0.f}, rr[3] = {0.f, 0.f, 0.f}, cr[3];
        for (int a = 0; a < D; a++) { nds[a] = n[a] * kk; rr[a] = x[a] - mo.x0[a]; }
        cross_<D>(rr, nds, cr);
        for (int a = 0; a < D; a++) acc[a] += (double)(pv * cr[a]);
      } else {
        for (int a = 0; a < D; a++) acc[a] += (double)(pv * (n[a] * kk));
      }
    }
  }
  const long b = blockIdx.x, nb = gridDim.x;
  for (int a = 0; a < 3; a++) { const double v = block_sum(acc[a]); if (threadIdx.x == 0) part[a * nb + b] = v; __syncthreads(); }
}
// viscous_force: Σ_inside −2ν·S(I,u)·n·kern(clamp(d,-1,1)) in Float64   src/Metrics.jl:140-154, ∂(i,j,I,u) :42-44
template <int D>
__global__ void k_vforce_body(GridX g, const float* __restrict__ u, float nu, BodyArg bd, MomArg mo, double* __restrict__ part) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  double acc[3] = {0, 0, 0};
  const long st[3] = {1, g.sy, g.sz};
  const int nsl = wl_nslots(g);
  if (cell_ij(g, m, i, j) && interior_ij(g, i, j)) {
    for (int k = g.k0 + pz; k < g.k1; k += nsl) {
      const int I[3] = {i + 1, j + 1, (D == 3) ? g.gk + k + 1 : 1};
      float x[3]; for (int q = 0; q < 3; q++) x[q] = (float)I[q] - 1.5f;
      float d, n[3]; body_measure<D>(bd, x, 1.f, d, n);
      const float kk = kern_(fminf(fmaxf(d, -1.f), 1.f));
      const long o = m + (long)k * g.sz;
      auto du = [&](int a, int b) -> float {       // ∂u_a/∂x_b at the cell centre
        const float* __restrict__ f = u + (long)a * g.cs;
        if (a == b) return f[o + st[a]] - f[o];
        return (f[o + st[b]] + f[o + st[b] + st[a]] - f[o - st[b]] - f[o - st[b] + st[a]]) / 4;
      };
      if (mo.on) {
        float sn[3] = {0.f, 0.f, 0.f}, rr[3] = {0.f, 0.f, 0.f}, cr[3];
        for (int a = 0; a < D; a++) {
          float v = 0.f;
          for (int b = 0; b < D; b++) { const float Sab = (du(a, b) + du(b, a)) / 2; v += Sab * (n[b] * kk); }
          sn[a] = v; rr[a] = x[a] - mo.x0[a];
        }
        cross_<D>(rr, sn, cr);
        for (int a = 0; a < D; a++) acc[a] += (double)((-2 * nu) * cr[a]);
      } else {
        for (int a = 0; a < D; a++) {
          float v = 0.f;
          for (int b = 0; b < D; b++) { const float Sab = (du(a, b) + du(b, a)) / 2; v += ((-2 * nu) * Sab) * (n[b] * kk); }
          acc[a] += (double)v;
        }
      }
    }
  }
  const long b = blockIdx.x, nb = gridDim.x;
  for (int a = 0; a < 3; a++) { const double v = block_sum(acc[a]); if (threadIdx.x == 0) part[a * nb + b] = v; __syncthreads(); }
}
__global__ void k_fin3(const double* __restrict__ part, int nb, double* __restrict__ out) {
  for (int a = 0; a < 3; a++) {
    double s = 0.0; for (int q = threadIdx.x; q < nb; q += WL_BLOCK) s += part[(long)a * nb + q];
    s = block_sum(s); if (threadIdx.x == 0) out[a] = s; __syncthreads();
  }
}
template <int D>
__global__ void k_l2_inside(GridX g, const float* __restrict__ a, double* __restrict__ part) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  double acc = 0.0;
  const int nsl = wl_nslots(g);
  if (cell_ij(g, m, i, j) && interior_ij(g, i, j))
    for (int k = g.k0 + pz; k < g.k1; k += nsl) { const double v = (double)a[m + (long)k * g.sz]; acc += v * v; }
  acc = block_sum(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void k_fin1(const double* __restrict__ part, int n, double* __restrict__ out) {
  double a = 0.0; for (int q = threadIdx.x; q < n; q += WL_BLOCK) a += part[q];
  a = block_sum(a); if (threadIdx.x == 0) *out = a;
}

// exitBC!(u,u⁰,Δt)   src/core.jl:226-233 — single-domain; face sums by one block (deterministic), all on device
// mode 0: out[0] = Σ u[2, 2:N-1.., 1]/len (inflow) ; mode 1: out[1] = Σ u[N, ..,1]/len − out[0]
template <int D>
__global__ void k_exit_facesum(GridX g, const float* __restrict__ u, double* __restrict__ out, int mode) {
  const int ny = g.ny - 2, nz = (D == 3) ? (g.nz - 2) : 1;
  const long cnt = (long)ny * nz;
  const int ix = (mode == 0) ? 1 : g.nx - 1;
  double acc = 0.0;
  for (long q = threadIdx.x; q < cnt; q += blockDim.x) {
    const int j = 1 + (int)(q % ny), k = (D == 3) ? 1 + (int)(q / ny) : 0;
    acc += (double)u[ix + (long)j * g.sy + (long)k * g.sz];
  }
  __shared__ double sh[16];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += sh[w];
    const float mean = (float)s / (float)cnt;
    if (mode == 0) out[0] = (double)mean; else out[1] = (double)(mean - (float)out[0]);
  }
}
// The x-exit face of the normal component (Julia index N of u[:,:,:,1], every row and plane): with the convective exit BC! leaves it alone
// (saveexit, src/core.jl:207) and only exitBC! rewrites its interior rows — so when the roles of the velocity buffers rotate instead of
// `u⁰ .= u` being a copy, the face has to travel with the role.
template <int D>
__global__ void k_copy_exit_face(GridX g, float* __restrict__ dst, const float* __restrict__ src) {
  const long cnt = (long)g.ny * (D == 3 ? g.nz : 1);
  const long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x;
  if (q >= cnt) return;
  const long o = (g.nx - 1) + q * g.sy;      // (rows are contiguous over planes: j + k·ny)
  dst[o] = src[o];
}
// z-slab variant: every rank sums its owned planes of the face (res_d[slot], summed over ranks by combine_results), then
// k_exit_mean turns the global sum into the mean exactly as above (float division by the global face size)
template <int D>
__global__ void k_exit_facesum_part(GridX g, const float* __restrict__ u, double* __restrict__ out, int mode) {
  const int ny = g.ny - 2, nz = g.k1 - g.k0;
  const long cnt = (long)ny * nz;
  const int ix = (mode == 0) ? 1 : g.nx - 1;
  double acc = 0.0;
  for (long q = threadIdx.x; q < cnt; q += blockDim.x) {
    const int j = 1 + (int)(q % ny), k = g.k0 + (int)(q / ny);
    acc += (double)u[ix + (long)j * g.sy + (long)k * g.sz];
  }
  __shared__ double sh[16];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += sh[w]; *out = s; }
}
__global__ void k_exit_mean(const double* __restrict__ gsum, double* __restrict__ sc, long cnt, int mode) {
  const float mean = (float)(*gsum) / (float)cnt;
  if (mode == 0) sc[0] = (double)mean; else sc[1] = (double)(mean - (float)sc[0]);
}
template <int D>
__global__ void k_exit_update_slab(GridX g, float* __restrict__ u, const float* __restrict__ u0, const double* __restrict__ sc, float dt, int mode) {
  const int ny = g.ny - 2;
  const long cnt = (long)ny * (g.k1 - g.k0);
  const long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x;
  if (q >= cnt) return;
  const int j = 1 + (int)(q % ny), k = g.k0 + (int)(q / ny);
  const long o = (g.nx - 1) + (long)j * g.sy + (long)k * g.sz;
  if (mode == 0) { const float U = (float)sc[0]; u[o] = u0[o] - U * dt * (u0[o] - u0[o - 1]); }
  else u[o] -= (float)sc[1];
}
// Δt = min(10, 1/(max σ + 5ν))  (src/Flow.jl:166, CFL :234-237) on the device: the statement wl_sim::cfl evaluates on the host from the same maximum
__global__ void k_dt_from_cfl(float* __restrict__ res_f, int in_slot, int out_slot, float nu) { res_f[out_slot] = fminf(10.f, 1.0f / (res_f[in_slot] + 5 * nu)); }
template <int D>
__global__ void k_exit_update(GridX g, float* __restrict__ u, const float* __restrict__ u0, const double* __restrict__ sc, float dt, int mode) {
  const int ny = g.ny - 2;
  const long cnt = (long)ny * ((D == 3) ? (g.nz - 2) : 1);
  const long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x;
  if (q >= cnt) return;
  const int j = 1 + (int)(q % ny), k = (D == 3) ? 1 + (int)(q / ny) : 0;
  const long o = (g.nx - 1) + (long)j * g.sy + (long)k * g.sz;
  if (mode == 0) { const float U = (float)sc[0]; u[o] = u0[o] - U * dt * (u0[o] - u0[o - 1]); }
  else u[o] -= (float)sc[1];
}
}  // namespace

#define DSEL(D, KERN, ...)                                                           \
  do { if ((D) == 3) hipLaunchKernelGGL(KERN<3>, __VA_ARGS__); else hipLaunchKernelGGL(KERN<2>, __VA_ARGS__); } while (0)

// ================================================================================================
struct wl_sim {
  wl_sim_desc d;
  wl_grid g; GridX G;
  float *u = nullptr, *u0 = nullptr, *f = nullptr, *p = nullptr, *sigma = nullptr, *V = nullptr, *mu0 = nullptr, *mu1 = nullptr;
  float* own = nullptr;
  wl_mg* mg = nullptr;
  wl_comm* comm = nullptr;   // not owned; NULL for a single domain
  bool swap_ok = false;      // u and u⁰ are handle-owned and every ghost of u is rewritten by BC! (no exitBC)
  float* ps = nullptr;       // spare pressure array (out-of-place x·dt and x/dt around the solve; two swaps restore p's identity)
  bool use_fuse_p = true;
  double* exit_sc = nullptr; // exitBC! on slabs: global face means (device)
  bool forcing = false;      // uniform g(i,t)+dU(i,t)/dt supplied by the host for the current step (accelerate!, src/Flow.jl:69-73)
  float acc0[3] = {0, 0, 0}, acc1[3] = {0, 0, 0};   // at t₀ (predictor) and t₁ (corrector)
  float* us = nullptr;       // spare velocity array: the fused corrector writes here, then u and us trade places
  std::vector<float> dt;
  bool own_mg = true;        // false: the multigrid handle belongs to the caller (wl_sim_create_on)
  ~wl_sim() { if (u_pending && comm && comm->cs) (void)hipStreamSynchronize(comm->cs); if (own_mg) delete mg; if (own) (void)hipFree(own); if (exit_sc) (void)hipFree(exit_sc); if (farmask) (void)hipFree(farmask); if (mnear) (void)hipFree(mnear); if (mneedf) (void)hipFree(mneedf); if (mm0var) (void)hipFree(mm0var); }

  // BC!(u) on the physical faces this rank holds, then the z-halo planes (depth 2: QUICK reads f[I-2δ], src/Flow.jl:8)
  // On slabs the exchange runs on the communicator's own stream; the compute stream waits for it (sync_u) only where the halo
  // planes are first read, so the interior planes of the next conv_diff! overlap with the transfer.
  bool use_overlap = true;
  bool u_pending = false;    // an exchange of the array that is now `u` or `u0` is in flight
  int sync_u(hipStream_t s) { if (u_pending) { u_pending = false; return wl::halo_async_wait(comm, s); } return 0; }
  // BC!(u,U) folded into the stores of the kernel that produced u (wl_bcfold.hpp): single domain, tuple U, no exit, no periodic direction
  // measured at 512³: projection tails −0.04 ms/step (kept), tiled conv_diff! +0.2…0.4 ms/step — the ghost writes are sector-granular
  // wherever they happen, and inside the tiled kernel they sit on the wall tiles' critical path (off; `bcfold` = 3 turns it on)
  int use_bcfold = 1; bool bc_folded = false;    // bit 0: projection tails, bit 1: tiled conv_diff!+BDIM!
  bool fold_ok(int bit) const { return (use_bcfold & bit) && d.D == 3 && !comm && !d.exitBC && !d.perdir_mask && G.nz == G.gnz && G.nx >= 6 && G.ny >= 6 && G.nz >= 6; }
  BcFold fold_req(int bit) const { BcFold f{fold_ok(bit) ? 1 : 0, {d.uBC[0], d.uBC[1], d.uBC[2]}}; return f; }
  int bc_u(hipStream_t s) {
    WL_TRY(sync_u(s));
    if (bc_folded) { bc_folded = false; return 0; }      // the producer already wrote every boundary location
    WL_TRY(wl::bc_vec(u, G, d.uBC, d.exitBC, d.perdir_mask, s));
    if (comm && use_overlap) { WL_TRY(wl::halo_async_begin(comm, u, G, d.D, 2, s)); u_pending = true; return 0; }
    return wl::halo(comm, u, G, d.D, 2, s);
  }
  // fused conv_diff!+BDIM! (NoBody): interior planes first when the advecting field's halo is still in flight
  bool store_f = false;      // the fused paths materialise the intermediates f = u⁰+Δt·r and z = ∇·u only on request: nothing on the time-step path reads them again
  // want_q1: also leave conv_diff!'s stale Φ in σ's ghost cells (quirk Q1: CFL's maximum(σ) sees them) — one small launch.  The predictor's are dead inside
  // mom_step!: the corrector's conv_diff! overwrites every one of them before anything reads σ's ghost cells.
  int conv_fused(const float* uadv, float* uout, float pre, float post, hipStream_t s, bool want_q1 = true, const float* dt_dev = nullptr) {
    const wl::ConstL& cl = mg->lv[0].cl;
    float* f = store_f ? this->f : nullptr;
    if (u_pending && G.D == 3 && G.k1 - G.k0 > 4) {
      if (dt_dev) { wl_set_error("conv_fused: Δt on the device is for the single domain"); return WL_EINVAL; }
      WL_TRY(wl::conv_diff_bdim(f, uadv, sigma, u0, mu0, uout, G, d.nu, d.perdir_mask, d.scheme, dt.back(), pre, post, cl, s, G.k0 + 2, G.k1 - 2, false));
      WL_TRY(sync_u(s));
      WL_TRY(wl::conv_diff_bdim(f, uadv, sigma, u0, mu0, uout, G, d.nu, d.perdir_mask, d.scheme, dt.back(), pre, post, cl, s, -(1 << 30), G.k0 + 2, false));
      return wl::conv_diff_bdim(f, uadv, sigma, u0, mu0, uout, G, d.nu, d.perdir_mask, d.scheme, dt.back(), pre, post, cl, s, G.k1 - 2, 1 << 30, want_q1);
    }
    WL_TRY(sync_u(s));
    BcFold fr = fold_req(2);
    fr.proj_x = proj_pending;
    fr.dt_dev = dt_dev;
    WL_TRY(wl::conv_diff_bdim(f, uadv, sigma, u0, mu0, uout, G, d.nu, d.perdir_mask, d.scheme, dt.back(), pre, post, cl, s, -(1 << 30), 1 << 30, want_q1, &fr));
    if (proj_pending && !fr.proj_done) { wl_set_error("mom_step!: the corrector did not take the deferred projection"); return WL_EINVAL; }
    proj_pending = nullptr;
    bc_folded = fr.on != 0;
    return 0;
  }
  // mom_project!'s first tail deferred into the corrector's conv_diff! (wl_convf.hip, PROJ): the projected predictor velocity has exactly one reader — the
  // corrector — so inside mom_step! the tail's u −= L∇x and the BC! after it are evaluated by that kernel's loader and the field is never written back
  // (−24 B/cell, one launch); p = x/Δt keeps its own small launch.  Only where the loader's closed form is the whole story (fold_ok: single domain, tuple U,
  // no periodic direction, no convective exit) and the corrector is the fused NoBody launch on whole tiles.
  // OPT-IN (option "tailfuse"): bit-identical, but at 512³ the loader's extra loads cost the corrector what the tail launch took (corrector 1.19 -> 1.83 ms
  // + 0.22 ms for p against 0.88 ms for the tail: the kernel sits at the 128-register budget of its two workgroups per CU, the x operands spill, and a
  // spill reload's in-order vmcnt wait drains the plane prefetch) — profiles/r03_experiments.md §13.
  bool use_tailfuse = false;
  const float* proj_pending = nullptr;
  bool tailfuse_ok() const {
    return use_tailfuse && fold_ok(3) && us && !d.has_body && !forcing && !store_f && !use_convz && !u_pending && mg->lv[0].cl.on && !mg->lv[0].part &&
           wl::conv_proj_ok(G, d.perdir_mask);
  }
  bool use_convz = false;    // z-marching conv_diff! (each flux once): bit-identical but measured 6 % SLOWER than the gather kernel at 512³ (opt-in)
  int conv_only(const float* uadv, hipStream_t s) {     // conv_diff!(f,uadv,σ) without BDIM!
    WL_TRY(sync_u(s));
    if (use_convz && wl::conv_z_ok(G, d.perdir_mask)) {
      WL_TRY(wl::conv_diff_z(f, uadv, nullptr, nullptr, nullptr, G, d.nu, d.scheme, 0.f, 0.f, 1.f, s));
      return wl::conv_q1(sigma, uadv, G, d.nu, d.perdir_mask, d.scheme, s);
    }
    return wl::conv_diff(f, uadv, sigma, G, d.nu, d.perdir_mask, d.scheme, s);
  }
  unsigned char* farmask = nullptr;   // per workgroup of BDIM's u pass: 1 = μ₁ ≡ 0 and V ≡ 0 there (refreshed by measure!/update!)
  bool use_farmask = true, mask_valid = false;   // handing out V or μ₁ (wl_sim_field) invalidates the mask until the next update!
  // body-aware conv_diff!+BDIM! (k_conv_diff<…,FUSE=2>): near / needf masks per (plane, in-plane workgroup)
  unsigned char *mnear = nullptr, *mneedf = nullptr, *mm0var = nullptr;
  int near_box[4] = {0, -1, 0, -1};   // {b0,b1,k0,k1}: bounding box of the near workgroups
  int dirty_z[2] = {0, -1};           // first / last plane with any near / f-keeping / μ₀-loading workgroup (the other planes are NoBody planes)
  bool use_hybrid = true;
  bool hybrid_ok() const { return d.has_body && use_hybrid && mask_valid && mnear && us && !comm && !forcing && !d.exitBC; }
  int refresh_body_mask(hipStream_t s) {
    if (!d.has_body || !mu1 || !V) return 0;
    if (!farmask) WL_HIP(hipMalloc((void**)&farmask, wl::body_mask_bytes(G)));
    if (!mnear) { const size_t nb = (size_t)wl::body_masks_nbm(G) * (size_t)G.nz; WL_HIP(hipMalloc((void**)&mnear, nb)); WL_HIP(hipMalloc((void**)&mneedf, nb)); WL_HIP(hipMalloc((void**)&mm0var, nb)); }
    mask_valid = true;
    WL_TRY(wl::body_masks(mnear, mneedf, mm0var, V, mu1, mu0, G, s));
    WL_TRY(wl::body_masks_box(mnear, G, near_box, s));
    WL_TRY(wl::body_masks_planes(mnear, mneedf, mm0var, G, dirty_z, s));
    return wl::body_mask(farmask, V, mu1, G, s);
  }
  // conv_diff!(f,uadv) + BDIM! with a body: fused NoBody form far from the body, two-pass BDIM! on the near workgroups only
  int conv_bdim_body(const float* uadv, float* uout, float pre, float post, hipStream_t s) {
    WL_TRY(sync_u(s));
    { ProfScope pc(WL_PROF_CONVDIFF, s);
      WL_TRY(wl::conv_diff_bdim_body(f, uadv, sigma, u0, mu0, uout, G, d.nu, d.perdir_mask, d.scheme, dt.back(), pre, post, mnear, mneedf, mm0var, wl::body_masks_nbm(G), store_f ? 1 : 0, s, dirty_z[0], dirty_z[1])); }
    ProfScope pb(WL_PROF_BDIM, s);
    return wl::bdim_near(uout, u, u0, f, V, mu0, mu1, G, dt.back(), pre, post, mnear, wl::body_masks_nbm(G), near_box, s);
  }
  int bdim_step(float pre, float post, hipStream_t s) {
    ProfScope pb(WL_PROF_BDIM, s);
    if (d.has_body) {
      WL_TRY(wl::bdim_f(f, u0, V, G, dt.back(), s));
      if (comm) WL_TRY(wl::halo(comm, f, G, d.D, 1, s));   // μddn reads f[I±δz] across the slab face: exchange f between the two passes
      return wl::bdim_u(u, f, V, mu0, mu1, G, pre, post, s, (use_farmask && mask_valid) ? farmask : nullptr);
    }
    return wl::bdim(u, u0, f, nullptr, mu0, nullptr, G, dt.back(), pre, post, s);
  }
  int exit_bc(hipStream_t s);
  int copy_exit_face(float* dst, const float* src, hipStream_t s);
  // BC!(u,U) after the fused conv_diff!+BDIM! DEFERRED inside mom_step! (option "bcdefer"): between that launch and the projection's tail — which rewrites
  // every boundary location of u through its folded stores — the only reader of boundary locations is the projection head's ∇·u (and the second tail's
  // flux_out), and only where a component is normal to the face, where BC! writes the constant U: the fused head and the pair tail substitute U on load
  // (wl_resjac_body.inc, k_project_cfl2) and the two k_bc_vec launches per step (2 × 0.07 ms at 512³: strided x faces) disappear.  Every other path that
  // would read u first (two-kernel head after a redo, unfused tails, wl_sim_phase, fields handed out) applies BC! before it does.
  bool use_bcdefer = true, in_step = false, bc_deferred = false;
  long n_bcdefer = 0;
  bool head_fused_ok() const {      // the projection will start with the fused head (wl_resjac.hip) — the condition project() tests
    const wl_mg::Level& l0 = mg->lv[0];
    return ps && use_fuse_p && use_resjac && !resjac_backoff && (!resjac_force_redo || redo_unannounced) && !d.exitBC && !store_f && !d.perdir_mask && !l0.part && mg->defer_shift && mg->lv.size() > 1 &&
           wl::resjac_ok(G, l0.cl) && !comm;
  }
  bool bcdefer_ok(bool second) const {
    if (!(use_bcdefer && in_step && !bc_folded && fold_ok(1) && head_fused_ok())) return false;
    return !second || (use_fuse_cfl && us && wl::project_cfl_pair_path(G, mg->lv[0].cl));
  }
  int bc_u_or_defer(bool second, hipStream_t s) {
    if (bcdefer_ok(second)) { bc_deferred = true; n_bcdefer++; return 0; }
    return bc_u(s);
  }
  int flush_bc(hipStream_t s) {       // apply a deferred BC! now (somebody is about to read u's boundary locations from memory)
    if (!bc_deferred) return 0;
    bc_deferred = false;
    return wl::bc_vec(u, G, d.uBC, d.exitBC, d.perdir_mask, s);
  }
  int predict(hipStream_t s, const float* dt_dev = nullptr) {                            // mom_predict! src/Flow.jl:190-196 (dt_dev: Δt still on the device — lazydt_ok() paths only)
    if (hybrid_ok()) {
      WL_TRY(conv_bdim_body(u0, u, 0.f, 1.f, s));
      return bc_u(s);
    }
    bool fused_conv = false;
    if (us && !d.has_body && !forcing) {   // conv_diff!(f,u⁰) + BDIM! in one launch (u⁰ is the advecting field, u the output)
      ProfScope pc(WL_PROF_CONVDIFF, s);
      if (use_convz && wl::conv_z_ok(G, d.perdir_mask)) {
        WL_TRY(sync_u(s));
        WL_TRY(wl::conv_diff_z(store_f ? f : nullptr, u0, u0, mu0, u, G, d.nu, d.scheme, dt.back(), 0.f, 1.f, s));
        WL_TRY(wl::conv_q1(sigma, u0, G, d.nu, d.perdir_mask, d.scheme, s));
      } else { WL_TRY(conv_fused(u0, u, 0.f, 1.f, s, !(in_step && !store_f), dt_dev)); fused_conv = true; }
    } else {
      { ProfScope pc(WL_PROF_CONVDIFF, s); WL_TRY(conv_only(u0, s)); }
      if (forcing) WL_TRY(wl::accelerate(f, G, acc0, s));                                  // accelerate!(f,t₀,g,uBC)
      WL_TRY(bdim_step(0.f, 1.f, s));   // scale_u!(a,0) folded (pre=0)
    }
    WL_TRY(fused_conv ? bc_u_or_defer(false, s) : bc_u(s));      // (deferral implies !exitBC: fold_ok)
    if (d.exitBC) WL_TRY(exit_bc(s));
    return 0;
  }
  int correct(hipStream_t s) {                                                           // mom_correct! :205-210
    if (hybrid_ok()) {
      WL_TRY(conv_bdim_body(u, us, 1.f, 0.5f, s));
      std::swap(u, us);
      return bc_u(s);
    }
    if (us && !d.has_body && !forcing) {   // the advecting field is u itself: write the new u to the spare array and swap
      bool fused_conv = false;
      { ProfScope pc(WL_PROF_CONVDIFF, s);
        if (use_convz && wl::conv_z_ok(G, d.perdir_mask)) {
          WL_TRY(sync_u(s));
          WL_TRY(wl::conv_diff_z(store_f ? f : nullptr, u, u0, mu0, us, G, d.nu, d.scheme, dt.back(), 1.f, 0.5f, s));
          WL_TRY(wl::conv_q1(sigma, u, G, d.nu, d.perdir_mask, d.scheme, s));
        } else { WL_TRY(conv_fused(u, us, 1.f, 0.5f, s)); fused_conv = true; } }
      std::swap(u, us);
      if (d.exitBC) WL_TRY(copy_exit_face(u, us, s));   // BC!(…,saveexit) keeps the predictor's exit face
      return fused_conv ? bc_u_or_defer(true, s) : bc_u(s);
    }
    { ProfScope pc(WL_PROF_CONVDIFF, s); WL_TRY(conv_only(u, s)); }
    if (forcing) WL_TRY(wl::accelerate(f, G, acc1, s));                                    // accelerate!(f,t₁,g,uBC)
    WL_TRY(bdim_step(1.f, 0.5f, s));  // scale_u!(a,0.5) folded (post)
    return bc_u(s);
  }
  int itmx = 32;             // solver!'s iteration cap (src/MultiLevelPoisson.jl:108); the multi-GPU rehearsal (tools/slab_rank_bench.py) lowers it to the 1 V-cycle the real run takes
  bool use_resjac = true;    // projection head + first Jacobi! in one launch (wl_resjac.hip) where eligible
  bool use_headspec = true;  // … and the first V-cycle queued behind it without waiting for Σr (single GPU)
  bool resjac_force_redo = false;   // test hook: behave as if the mean shift were always due (exercises the redo path)
  bool redo_unannounced = false;    // test hook: … and do not let the BC! deferral know in advance (as with a real shift)
  long n_tailfuse = 0;       // projections whose velocity update ran inside the corrector's conv_diff!
  long n_resjac = 0, n_resjac_redo = 0;   // how often the fused head stood / had to be redone because the mean shift was due
  int resjac_redo_run = 0;                // consecutive redos: after WL_RESJAC_BACKOFF of them the fused head is switched off for this handle
  bool resjac_backoff = false;            // (a flow whose residual needs the mean shift on every solve would pay launch + sync + two-kernel path each time); re-armed by update!
  int p_shell = -1;          // ghost shell of p / the spare pressure array: -1 unknown (check before the next fused head), 0 all +0, 1 something else, 2 caller-owned p (never assumed)
  bool use_fuse_cfl = true;  // the corrector's projection tail also produces CFL's σ and max(σ)
  bool cfl_done = false;
  static constexpr int CFL_SLOT = 5;   // res_f slot of CFL's maximum (not slot 0: a tail queued ahead of the solver's read must leave the head's L∞ there for the log)
  bool use_tailspec = true;  // the projection tail is queued behind the smoother before the host has read the norms, gated on the device by the break test (single GPU)
  long n_tailspec = 0;
  int project(float w, hipStream_t s, bool with_cfl = false, bool defer_tail = false) {    // mom_project! :223-232 (defer_tail: inside mom_step!, the corrector follows)
    const float dtl = w * dt.back();
    cfl_done = false;
    WL_TRY(sync_u(s));                                                                     // div(u) reads the halo planes
    if (bc_deferred && !head_fused_ok()) WL_TRY(flush_bc(s));                              // (cannot happen: the deferral tested the same condition — kept as the invariant's guard)
    if (ps && use_fuse_p && !(comm && d.perdir_mask)) {   // (z-slabs: p's ghost planes are current — exchanged at the end of the last solve, scaled with the rest)
      WL_TRY(wl::bc_per_scalar(p, G, d.perdir_mask, s));                                   // residual!: perBC!(x) :93 (copies commute with the scaling)
      // head: z=div(u); x.*=dt; residual! in one pass — the scaled pressure goes to the spare array, which becomes p
      wl_mg::Level& l0 = mg->lv[0];
      bool head_done = false;
      double pre_r1 = 0.0; float pre_rinf = 0.f;
      bool solved = false;      // the speculative solve behind the fused head stood
      // ---- the tail, as a function of a device flag (go != nullptr: queued inside the solver loop ahead of its read — runs iff the flag says "converged")
      const bool split = l0.part && mg->use_zsplit && !comm;        // a body: the three plane ranges of the z-split (see above)
      const int zm = 4, zna = split ? std::max(l0.g.k0, l0.za - zm) : 0, znb = split ? std::min(l0.g.k1, l0.zb + zm + 1) : 0;
      int tail_kind = 0;        // 1: projection + flux_out + max σ into the spare array, 2: projection in place, 3: left to the corrector's loader
      bool tail_stood = false;
      auto launch_tail = [&](const float* go) -> int {
        // (p is the solver's x by now, ps the array the unscaled pressure goes to: both solve() call sites swap before they call)
        if (with_cfl && use_fuse_cfl && us && !d.exitBC && !d.perdir_mask) {   // + flux_out and its maximum; projected u lands in the spare array
          tail_kind = 1;
          if (split) { WL_TRY(flush_bc(s)); WL_TRY(wl::project_cfl_split(us, u, mu0, p, ps, sigma, G, dtl, l0.cl, l0.clp, zna, znb, mg->ws, CFL_SLOT, s, store_f ? 1 : 0)); }
          else {
            BcFold fr = fold_req(1);
            if (bc_deferred && !(fr.on && wl::project_cfl_pair_path(G, l0.cl))) WL_TRY(flush_bc(s));
            fr.usub = bc_deferred ? 1 : 0;      // flux_out reads the wall-normal boundary faces of the corrector's output: U on load
            fr.go = go;
            WL_TRY(wl::project_cfl(us, u, mu0, p, ps, sigma, G, dtl, l0.cl, mg->ws, CFL_SLOT, s, store_f ? 1 : 0, &fr)); bc_folded = fr.on != 0;
          }
        } else if (split) { tail_kind = 2; WL_TRY(wl::project_unscale_split(u, mu0, p, ps, G, dtl, l0.cl, l0.clp, zna, znb, s)); }
        else if (defer_tail && tailfuse_ok()) {   // p = x/Δt now; u −= L∇x and BC! when the corrector reads u (the scaled x stays untouched in the spare pressure array until then)
          tail_kind = 3;
          WL_TRY(wl::div_scalar_to(ps, p, dtl, (size_t)G.cs, s));
        }
        else { tail_kind = 2; BcFold fr = fold_req(1); fr.go = go; WL_TRY(wl::project_unscale(u, mu0, p, ps, G, dtl, l0.cl, s, &fr)); bc_folded = fr.on != 0; }
        return 0;
      };
      // the forms that honour the flag: the in-place tail and the pair tail with CFL (not the z-split of a body, not the corrector-loader form)
      const bool tail_gateable = !split && !(defer_tail && tailfuse_ok()) &&
                                 (!(with_cfl && use_fuse_cfl && us && !d.exitBC && !d.perdir_mask) || (wl::project_cfl_pair_path(G, l0.cl) && (!bc_deferred || fold_req(1).on)));
      if (use_resjac && !resjac_backoff && !d.exitBC && !store_f && !d.perdir_mask && !l0.part && mg->defer_shift && mg->lv.size() > 1 && wl::resjac_ok(G, l0.cl) &&
          (!comm || (l0.dist && mg->x_halo_depth >= 2))) {   // (exitBC: the convective exit leaves a net flux imbalance to the solver's tolerance — the shift is usually due; z-slab: p's ghost planes are current two deep)
        // head + the V-cycle's first Jacobi!(fine) in one launch, assuming residual!'s mean shift is not due (wl_resjac.hip); Σr decides
        { ProfScope pr(WL_PROF_RESIDUAL, s);
          // p's and the spare's ghost cells are +0 unless someone wrote them from outside (checked once after a pointer to p was handed out): no shell pass then
          if (comm) p_shell = 1;   // (a slab's ghost planes hold the neighbours' pressure: always scaled with the rest)
          if (p_shell < 0) p_shell = (wl::shell_nonzero(p, G, (int*)(mg->ws.res_f + 7), s) || wl::shell_nonzero(ps, G, (int*)(mg->ws.res_f + 7), s)) ? 1 : 0;
          WL_TRY(wl::resjac(ps, l0.eps, p, u, G, dtl, 1.f, l0.cl, mg->ws, 1, 0, s, p_shell != 0, bc_deferred ? d.uBC : nullptr)); }
        if (use_headspec && !comm && itmx >= 1) {
          // solver! runs its V-cycle at least once whatever the initial norms are (src/MultiLevelPoisson.jl:113-123), so Σr is not needed before the first cycle is
          // queued: the cycle is launched behind the head at once and Σr comes back with the first iteration's norms (one host round trip per solve fewer, no idle
          // GPU while the host decides).  If the shift turns out to be due, that solve is discarded — the head's inputs are untouched — and the two-kernel path taken.
          std::swap(p, ps); l0.x = p;
          std::swap(l0.r, l0.eps);
          mg->jacobi0_done = true;
          if (use_tailspec && tail_gateable && !resjac_force_redo) { mg->spec_tail = launch_tail; mg->spec_check_head = true; }
          WL_TRY(mg->solve(2e-3, itmx, nullptr, nullptr, nullptr, s, true, nullptr, nullptr));
          tail_stood = mg->tail_stood; if (tail_stood) n_tailspec++;
          const float sm = (float)mg->first_hd0 / (float)(double)wl_ninside_global(mg->lv[0].g);
          if (std::fabs(sm) <= 2.f * 1.1920929e-7f && !resjac_force_redo) { head_done = true; solved = true; n_resjac++; resjac_redo_run = 0; }
          else {
            std::swap(l0.r, l0.eps); std::swap(p, ps); l0.x = p;
            mg->n.pop_back(); mg->jacobi0_done = false;
            n_resjac_redo++;
            if (!resjac_force_redo && ++resjac_redo_run >= 3) resjac_backoff = true;
          }
        } else {
        WL_TRY(wl::combine_results(comm, mg->ws, s));            // z-slabs: Σr, L₁ (sums) and L∞ (max) over the ranks — every rank takes the same branch below
        double hd2[2]; WL_TRY(wl::read_results(mg->ws, hd2, 2, &pre_rinf, 1, s));
        const double sr = hd2[0]; pre_r1 = hd2[1];
        const float sm = (float)sr / (float)(double)wl_ninside_global(mg->lv[0].g);
        if (std::fabs(sm) <= 2.f * 1.1920929e-7f && !resjac_force_redo) {                                       // src/Poisson.jl:96: no shift — the fused results stand
          std::swap(p, ps); l0.x = p;
          std::swap(l0.r, l0.eps);
          mg->jacobi0_done = true; head_done = true; n_resjac++; resjac_redo_run = 0;
        } else {                                                                           // shift due: the inputs are untouched, take the two-kernel path
          n_resjac_redo++;
          if (!resjac_force_redo && ++resjac_redo_run >= 3) resjac_backoff = true;
        }
        }
      }
      if (!head_done) {
        WL_TRY(flush_bc(s));      // the two-kernel head reads u's boundary faces from memory
        ProfScope pr(WL_PROF_RESIDUAL, s);
        if (l0.part && mg->use_zsplit && !comm) {   // a body: coefficients from the position on the plane ranges away from it (as in smooth!)
          const int m = 4, na = std::max(l0.g.k0, l0.za - m), nb = std::min(l0.g.k1, l0.zb + m + 1);
          WL_TRY(wl::div_residual_split(store_f ? sigma : nullptr, ps, l0.r, p, u, mu0, l0.D, l0.iD, G, dtl, mg->ws, l0.cl, l0.clp, na, nb, s));
        } else WL_TRY(wl::div_residual(store_f ? sigma : nullptr, ps, l0.r, p, u, mu0, l0.D, l0.iD, G, dtl, mg->ws, l0.cl, s));
      }
      if (!head_done) { std::swap(p, ps); l0.x = p; }
      if (!solved) WL_TRY(mg->solve(2e-3, itmx, nullptr, nullptr, nullptr, s, true, head_done ? &pre_r1 : nullptr, head_done ? &pre_rinf : nullptr));
      // tail: u -= L∇x ; x./=dt in one pass — the unscaled pressure goes back to the original array
      if (!tail_stood) WL_TRY(launch_tail(nullptr));
      if (tail_kind == 3) {   // deferred into the corrector's loader (tailfuse)
        bc_deferred = false;                  // (the corrector's loader reads this u through the projection AND BC!: nothing in memory is missing)
        proj_pending = p;
        std::swap(p, ps); l0.x = p;
        n_tailfuse++;
        return 0;
      }
      if (tail_kind == 1) {
        bc_deferred = false;                  // the folded stores wrote every boundary location of the new u
        WL_TRY(wl::combine_results(comm, mg->ws, s));   // max over ranks — issued BEFORE the u exchange starts on the other stream, so that
        std::swap(u, us); cfl_done = true;              // exchange stays in flight across the Δt read-back and the next predictor's interior
      }
      std::swap(p, ps); l0.x = p;
      bc_deferred = false;      // the tails update a cell from its own value only; whatever BC! had not been applied is applied now (folded stores or bc_u)
      return bc_u(s);
    }
    WL_TRY(flush_bc(s));
    WL_TRY(wl::div_scale(sigma, p, u, G, dtl, s));                                       // z=div(u); x.*=dt
    WL_TRY(mg->solve(2e-3, itmx, nullptr, nullptr, nullptr, s));
    WL_TRY(wl::project(u, mu0, p, G, s));
    WL_TRY(wl::div_scalar(p, dtl, (size_t)G.cs, s));                                     // x./=dt
    return bc_u(s);
  }
  // Δt of the NEXT step left on the device (wl_sim_mom_steps only: more steps follow inside the same call, nobody can look at the history in between): the finaliser
  // of CFL's maximum is followed by a one-thread kernel with mom_step!'s formula, the next predictor reads Δt through a pointer and is queued at once; the host
  // copies the maximum while that predictor runs and appends the same Δt to the history (same statements on the same number: same bits).
  bool use_lazydt = true, dt_pending = false;
  hipEvent_t ev_dt = nullptr;
  bool lazydt_ok() const {      // the next predictor will be the flux-once tiled launch on the single domain
    return use_lazydt && in_step && !comm && us && !d.has_body && !forcing && !use_convz && !d.exitBC && !d.perdir_mask && !store_f && wl::conv_flux_on() &&
           wl::conv_tile_ok(G, d.perdir_mask, G.k1 - G.k0) && mg->lv[0].cl.on;
  }
  int cfl(hipStream_t s, bool more_follow = false) {                                     // CFL :234-237
    if (!cfl_done) { WL_TRY(sync_u(s)); WL_TRY(wl::cfl_dev(u, sigma, G, mg->ws, CFL_SLOT, s)); WL_TRY(wl::combine_results(comm, mg->ws, s)); }   // max over ranks
    cfl_done = false;
    if (more_follow && lazydt_ok()) {
      hipLaunchKernelGGL(k_dt_from_cfl, dim3(1), dim3(1), 0, s, mg->ws.res_f, CFL_SLOT, CFL_SLOT + 1, d.nu);
      dt_pending = true;
      return 0;
    }
    float hf6[CFL_SLOT + 1]; WL_TRY(wl::read_results(mg->ws, nullptr, 0, hf6, CFL_SLOT + 1, s)); const float mx = hf6[CFL_SLOT];
    dt.push_back(std::fmin(10.f, 1.0f / (mx + 5 * d.nu)));
    return 0;
  }
  int mom_step(hipStream_t s, bool more_follow = false) {                                // mom_step! :156-167 (more_follow: wl_sim_mom_steps — another step comes inside the same call)
    ProfScope pstep(WL_PROF_STEP, s);
    // u⁰ .= u ; scale_u!(a,0): when the handle owns both arrays the copy is a pointer swap — the predictor overwrites
    // every interior cell of u (BDIM! with pre=0) and BC! every ghost cell, so nothing of the old u survives anyway.
    if (swap_ok) { std::swap(u, u0); if (d.exitBC) WL_TRY(copy_exit_face(u, u0, s)); }   // (an exchange still in flight belongs to the array that is now u⁰ — the predictor's advecting field)
    else { WL_TRY(sync_u(s)); WL_HIP(hipMemcpyAsync(u0, u, sizeof(float) * (size_t)G.cs * d.D, hipMemcpyDeviceToDevice, s)); }   // u⁰ .= u
    struct InStep { bool& f; InStep(bool& b) : f(b) { f = true; } ~InStep() { f = false; } } guard(in_step);
    if (dt_pending) {   // the CFL maximum of the previous step is copied back between ITS finaliser and THIS predictor, which takes Δt from the device
      if (!ev_dt) WL_HIP(hipEventCreateWithFlags(&ev_dt, hipEventDisableTiming));
      double hd1[1]; float hf7[CFL_SLOT + 2];
      WL_TRY(wl::read_results_overlapped(mg->ws, hd1, 1, hf7, CFL_SLOT + 2, s, ev_dt, [&]() -> int { return predict(s, mg->ws.res_f + CFL_SLOT + 1); }));
      dt.push_back(std::fmin(10.f, 1.0f / (hf7[CFL_SLOT] + 5 * d.nu)));
      dt_pending = false;
    } else
    WL_TRY(predict(s));
    WL_TRY(project(1.f, s, false, true));
    WL_TRY(correct(s));
    WL_TRY(project(0.5f, s, true));
    WL_TRY(flush_bc(s));      // (nothing is pending here: every projection ends with BC! applied — guard)
    return cfl(s, more_follow);
  }
};
int wl_sim::copy_exit_face(float* dst, const float* src, hipStream_t s) {
  const long cnt = (long)G.ny * (G.D == 3 ? G.nz : 1);
  DSEL(G.D, k_copy_exit_face, dim3((unsigned)((cnt + WL_BLOCK - 1) / WL_BLOCK)), dim3(WL_BLOCK), 0, s, G, dst, src);
  WL_LAUNCH_CHECK(); return 0;
}
int wl_sim::exit_bc(hipStream_t s) {
  if (comm) {   // z-slabs: the exit face is shared by all ranks — global means from per-rank sums (one 128-byte all-gather each)
    if (!exit_sc) WL_HIP(hipMalloc((void**)&exit_sc, 2 * sizeof(double)));
    WL_TRY(sync_u(s));
    const long gcnt = (long)(G.ny - 2) * (G.gnz - 2), lcnt = (long)(G.ny - 2) * (G.k1 - G.k0);
    const unsigned nbl = (unsigned)((lcnt + WL_BLOCK - 1) / WL_BLOCK);
    for (int mode = 0; mode < 2; mode++) {
      hipLaunchKernelGGL(k_exit_facesum_part<3>, dim3(1), dim3(1024), 0, s, G, (const float*)u, mg->ws.res_d + 4, mode);
      WL_TRY(wl::combine_results(comm, mg->ws, s));
      hipLaunchKernelGGL(k_exit_mean, dim3(1), dim3(1), 0, s, (const double*)(mg->ws.res_d + 4), exit_sc, gcnt, mode);
      hipLaunchKernelGGL(k_exit_update_slab<3>, dim3(nbl), dim3(WL_BLOCK), 0, s, G, u, (const float*)u0, (const double*)exit_sc, dt.back(), mode);
    }
    WL_LAUNCH_CHECK();
    return wl::halo(comm, u, G, d.D, 2, s);      // the exit face changed after BC!'s exchange
  }
  double* sc = mg->ws.res_d + 4;
  const long cnt = (long)(G.ny - 2) * (G.D == 3 ? (G.nz - 2) : 1);
  const unsigned nb = (unsigned)((cnt + WL_BLOCK - 1) / WL_BLOCK);
  DSEL(G.D, k_exit_facesum, dim3(1), dim3(1024), 0, s, G, u, sc, 0);
  DSEL(G.D, k_exit_update, dim3(nb), dim3(WL_BLOCK), 0, s, G, u, u0, sc, dt.back(), 0);
  DSEL(G.D, k_exit_facesum, dim3(1), dim3(1024), 0, s, G, u, sc, 1);
  DSEL(G.D, k_exit_update, dim3(nb), dim3(WL_BLOCK), 0, s, G, u, u0, sc, dt.back(), 1);
  WL_LAUNCH_CHECK(); return 0;
}

extern "C" {

int wl_exit_bc(float* u, const float* u0, const wl_grid* g, float dt, void* st) {
  WL_CHECK(wl_grid_ok(g), "bad wl_grid"); WL_CHECK(g->D == 2 || g->nz == g->gnz, "exitBC! needs the whole x-exit face on one rank");
  WL_TRY(wl_ctx_ensure());
  const GridX G = gx(*g); hipStream_t s = wl_stream(st);
  double* sc = wl_red_ws(wl_ctx().red).res_d + 4;
  const long cnt = (long)(G.ny - 2) * (G.D == 3 ? (G.nz - 2) : 1);
  const unsigned nb = (unsigned)((cnt + WL_BLOCK - 1) / WL_BLOCK);
  DSEL(G.D, k_exit_facesum, dim3(1), dim3(1024), 0, s, G, (const float*)u, sc, 0);
  DSEL(G.D, k_exit_update, dim3(nb), dim3(WL_BLOCK), 0, s, G, u, u0, sc, dt, 0);
  DSEL(G.D, k_exit_facesum, dim3(1), dim3(1024), 0, s, G, (const float*)u, sc, 1);
  DSEL(G.D, k_exit_update, dim3(nb), dim3(WL_BLOCK), 0, s, G, u, u0, sc, dt, 1);
  WL_LAUNCH_CHECK(); return 0;
}
int wl_L2_inside(const float* a, const wl_grid* g, double* out, void* st) {
  WL_CHECK(wl_grid_ok(g), "bad wl_grid"); WL_TRY(wl_ctx_ensure());
  const GridX G = gx(*g); hipStream_t s = wl_stream(st);
  const RedWs ws = wl_red_ws(wl_ctx().red);
  dim3 grid = wl_plane_grid(G, wl_red_slots(G, G.k1 - G.k0));
  DSEL(G.D, k_l2_inside, grid, dim3(WL_BLOCK), 0, s, G, a, ws.pa);
  hipLaunchKernelGGL(k_fin1, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, (int)grid.x, ws.res_d + 0);
  WL_LAUNCH_CHECK();
  return wl::read_results(ws, out, 1, nullptr, 0, s);
}

static int sim_create_common(wl_sim** out, const wl_sim_desc* desc, wl_comm* comm, wl_mg* adopt = nullptr) {
  WL_CHECK(out && desc, "null pointer"); WL_CHECK(desc->D == 2 || desc->D == 3, "D must be 2 or 3");
  WL_TRY(wl_ctx_ensure());
  const bool slab = comm && comm->size > 1;
  wl_sim* s = new wl_sim(); s->d = *desc; s->comm = slab ? comm : nullptr;
  if (slab && desc->D == 3 && ((desc->perdir_mask >> 2) & 1u)) comm->zperiodic = true;      // the halo exchanges wrap around (rank 0 <-> rank size-1)
  int32_t ng[3] = {desc->dims[0] + 2, desc->dims[1] + 2, desc->D == 3 ? desc->dims[2] + 2 : 1};
  if (slab) {
    if (desc->u || desc->u0 || desc->f || desc->p || desc->sigma || desc->V || desc->mu0 || desc->mu1) { delete s; wl_set_error("slab simulations own their arrays"); return WL_EINVAL; }
    static const int ghost = [] { const char* e = getenv("WL_SLAB_GHOST"); const int v = e ? atoi(e) : 5; return v >= 3 ? v : 5; }();
    const int rc = wl_grid_slab(&s->g, desc->D, ng, comm->rank, comm->size, ghost);   // ghost planes: QUICK needs 2, kernel B of the blocked smoother 3, the one-exchange smooth! 5
    if (rc != 0) { delete s; return rc; }
  } else s->g = wl_grid_single(desc->D, ng);
  s->G = gx(s->g);
  const size_t nc = (size_t)s->G.cs; const int D = desc->D;
  float** ptrs[8] = {&s->u, &s->u0, &s->f, &s->p, &s->sigma, &s->V, &s->mu0, &s->mu1};
  float* given[8] = {desc->u, desc->u0, desc->f, desc->p, desc->sigma, desc->V, desc->mu0, desc->mu1};
  const size_t sz[8] = {nc * D, nc * D, nc * D, nc, nc, nc * D, nc * D, nc * D * D};
  size_t total = 0;
  const bool none = !desc->u && !desc->u0, all3 = desc->u && desc->u0 && desc->us;
  const bool rot_ok = !desc->exitBC || !slab;               // convective exit: the buffers rotate on a single domain only (the exit face travels with the role, k_copy_exit_face)
  const bool want_us = none && !desc->us && rot_ok;         // spare velocity array of the out-of-place fused kernels (handle-owned)
  if (want_us) total += nc * D;
  total += nc;   // ps
  for (int q = 0; q < 8; q++) if (!given[q] && !(q == 7 && !desc->has_body) && !(q == 5 && !desc->has_body)) total += sz[q];
  if (total) { hipError_t e = hipMalloc((void**)&s->own, total * sizeof(float)); if (e != hipSuccess) { delete s; wl_set_error("hipMalloc failed for flow arrays"); return (int)e; } (void)hipMemset(s->own, 0, total * sizeof(float)); }
  float* pcur = s->own;
  for (int q = 0; q < 8; q++) {
    if (given[q]) *ptrs[q] = given[q];
    else if ((q == 7 || q == 5) && !desc->has_body) *ptrs[q] = nullptr;
    else { *ptrs[q] = pcur; pcur += sz[q]; }
  }
  if (want_us) { s->us = pcur; pcur += nc * D; }
  else if (desc->us && (none || all3) && rot_ok) s->us = desc->us;   // caller-owned spare: the roles of {u,u0,us} rotate (wlhip.h)
  s->ps = pcur; pcur += nc;
  s->dt.assign(1, desc->dt0);
  if (desc->p) s->p_shell = 2;     // a caller-owned p can be written behind the library's back: its ghost shell is always scaled
  s->swap_ok = (none || all3) && rot_ok;
  // μ₀ = 1 with BC!(μ₀,0)   src/Flow.jl:144-145  (only when the handle owns μ₀; a caller-owned μ₀ is taken as is)
  if (!desc->mu0) {
    int rc = wl::fill(s->mu0, 1.f, nc * D, 0); const float zero[3] = {0, 0, 0};
    if (rc == 0) rc = wl::bc_vec(s->mu0, s->G, zero, 0, desc->perdir_mask, 0);
    if (rc == 0) rc = wl::halo(s->comm, s->mu0, s->G, D, 2, 0);
    if (rc != 0) { delete s; return rc; }
  }
  if (adopt) {   // the caller's MultiLevelPoisson (wl_mg_create on the same p, μ₀, σ): used, not owned
    if (slab || adopt->lv.empty() || adopt->lv[0].x != s->p || adopt->lv[0].L != s->mu0 || adopt->lv[0].z != s->sigma) {
      delete s; *out = nullptr; wl_set_error("wl_sim_create_on: the wl_mg handle was not built on this flow's p, mu0, sigma"); return WL_EINVAL;
    }
    if (adopt->perdir != desc->perdir_mask) {   // the hierarchy's perBC! pattern is fixed at wl_mg_create: it has to be the flow's
      delete s; *out = nullptr; wl_set_error("wl_sim_create_on: the wl_mg handle was created with a different perdir mask than desc->perdir_mask"); return WL_EINVAL;
    }
    s->mg = adopt; s->own_mg = false;
  } else {
    s->mg = new wl_mg();
    int rc = s->mg->build(s->p, s->mu0, s->sigma, s->g, desc->perdir_mask, 10, s->comm);   // pois_ctor default  src/WaterLily.jl:97
    if (rc != 0) { delete s; *out = nullptr; return rc; }
  }
  s->mg->store_eps = false;   // p.ϵ is pure scratch on the time-step path
  if (slab && s->G.k0 >= 2 && s->G.nz - s->G.k1 >= 2) s->mg->x_halo_depth = 2;   // z-slab: the fused projection head recomputes the residual of the neighbour's boundary plane (x two planes deep)
  *out = s; return 0;
}
// Placement trials.  Where the driver puts a multi-GB allocation decides what the z-marching kernels get out of HBM: the same binary ran the
// finest-level smoother kernels 7–12 % and the fused projection head 9 % slower on one allocation than on the next (one process, simulations created
// one after the other: profiles/r03_placement_trial.txt; a plain z-marching copy: 3.6 vs 4.7 TB/s, profiles/r03_place_probe2.txt) — element-wise
// kernels do not care.  A handle that owns all its arrays therefore creates up to `trials` candidates (each in new memory: the others are held meanwhile),
// times mom_project! once on each (zero fields: every kernel of the projection runs, nothing changes) and keeps the fastest.  Results are placement-
// independent; only large grids take part (small ones live in the caches).
// OFF by default (WL_PLACEMENT_TRIALS=1; set it to 2…8 to try): six default bench runs with 6 candidates against six without (profiles/r03_bench_distribution.txt) —
// mean step 9.88 vs 9.89 ms, the smoother pair in its fast state in 2 of 6 vs 3 of 6 runs.  Candidates allocated after the first one are rarely in the fast state,
// so holding memory to force new placements buys almost nothing; the finding (placement decides 7–12 % of the pair) stands, the remedy does not work.
static double placement_score(wl_sim* s) {
  hipStream_t q = 0;
  const size_t n0 = s->mg->n.size();
  if (s->project(1.f, q) != 0) return 1e30;            // warm-up (first-launch costs)
  if (hipStreamSynchronize(q) != hipSuccess) return 1e30;
  hipEvent_t a, b; if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 1e30;
  float ms = 1e30f; int rc = 0;
  for (int rep = 0; rep < 2 && rc == 0; rep++) {        // the faster of two timed pairs (each holds host read-backs of the solver)
    (void)hipEventRecord(a, q);
    rc = s->project(1.f, q); if (rc == 0) rc = s->project(0.5f, q, true);
    (void)hipEventRecord(b, q); (void)hipEventSynchronize(b);
    float t = 1e30f; (void)hipEventElapsedTime(&t, a, b);
    if (t < ms) ms = t;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  // back to the state of a fresh handle: pois.n, counters, Δt untouched by mom_project!; u, p are still zero (BC! wrote the boundary values the
  // initial condition / init_flow will write again)
  s->mg->n.resize(n0); s->n_resjac = 0; s->n_resjac_redo = 0; s->resjac_redo_run = 0; s->resjac_backoff = false; s->cfl_done = false;
  s->mg->log_r1.clear(); s->mg->log_rinf.clear(); s->mg->log_w.clear();
  return rc == 0 ? (double)ms : 1e30;
}
static double g_last_placement[8]; static int g_last_placement_n = 0;
int wl_sim_create(wl_sim** out, const wl_sim_desc* desc) {
  WL_CHECK(out && desc, "null pointer");
  static const int trials_env = [] { const char* e = getenv("WL_PLACEMENT_TRIALS"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
  const bool owned = !desc->u && !desc->u0 && !desc->f && !desc->p && !desc->sigma && !desc->V && !desc->mu0 && !desc->mu1 && !desc->us;
  const long cells = (long)desc->dims[0] * desc->dims[1] * (desc->D == 3 ? desc->dims[2] : 1);
  int trials = (owned && desc->D == 3 && cells >= (48L << 20)) ? trials_env : 1;      // (≥ 48 Mi cells: the arrays are far larger than the Infinity Cache)
  if (trials > 1) {   // all candidates are alive until the choice is made: never take more than half of the free memory for them (≈160 B per cell and candidate)
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) { const long fit = (long)(fr / 2 / ((size_t)cells * 160 + 1)); if (fit < trials) trials = fit < 1 ? 1 : (int)fit; }
  }
  g_last_placement_n = 0;
  if (trials == 1) return sim_create_common(out, desc, nullptr);
  wl_sim* cand[8] = {nullptr}; double score[8]; int best = -1;
  for (int i = 0; i < trials; i++) {
    const int rc = sim_create_common(&cand[i], desc, nullptr);
    if (rc != 0) { cand[i] = nullptr; if (best < 0 && i == trials - 1) return rc; break; }      // (out of memory for another candidate: go with what exists)
    score[i] = placement_score(cand[i]);
    g_last_placement[g_last_placement_n++] = score[i];
    if (best < 0 || score[i] < score[best]) best = i;
  }
  if (best < 0) return WL_EINVAL;
  for (int i = 0; i < trials; i++) if (cand[i] && i != best) delete cand[i];
  *out = cand[best];
  return 0;
}
// scores (ms of the timed mom_project! pair) of the candidates of the last wl_sim_create: measurement interface
int wl_placement_scores(double* out, int cap) { int k = 0; for (; k < g_last_placement_n && k < cap; k++) out[k] = g_last_placement[k]; return g_last_placement_n; }
int wl_sim_create_on(wl_sim** out, const wl_sim_desc* desc, wl_mg* mg) { WL_CHECK(mg, "null wl_mg"); return sim_create_common(out, desc, nullptr, mg); }
int wl_sim_create_slab(wl_sim** out, const wl_sim_desc* desc, wl_comm* comm) { return sim_create_common(out, desc, comm); }
int wl_sim_destroy(wl_sim* s) { delete s; return 0; }
float* wl_sim_field(wl_sim* s, const char* name) {
  const std::string n(name);
  (void)s->sync_u(0);        // the caller is about to read or write the arrays: finish an exchange that is still in flight
  if (n == "V" || n == "mu1" || n == "mu0") s->mask_valid = false;
  if (n == "u") return s->u; if (n == "u0") return s->u0; if (n == "f") return s->f; if (n == "p") { if (s->p_shell != 2) s->p_shell = -1; return s->p; }
  if (n == "sigma") return s->sigma; if (n == "V") return s->V; if (n == "mu0") return s->mu0; if (n == "mu1") return s->mu1;
  if (n == "us") return s->us;
  return nullptr;
}
wl_mg* wl_sim_pois(wl_sim* s) { return s->mg; }
int wl_sim_grid(const wl_sim* s, wl_grid* out) { *out = s->g; return 0; }
int wl_sim_init_flow(wl_sim* s, void* st) {                                               // Flow ctor :141-142
  hipStream_t q = wl_stream(st);
  WL_TRY(s->bc_u(q));
  if (s->d.exitBC) {   // exitBC!(u,u,zero(T)): u⁰ aliases u, Δt = 0
    const float keep = s->dt.back(); float* keep0 = s->u0;
    s->dt.back() = 0.f; s->u0 = s->u;
    const int rc = s->exit_bc(q);
    s->dt.back() = keep; s->u0 = keep0;
    if (rc != 0) return rc;
  }
  WL_HIP(hipMemcpyAsync(s->u0, s->u, sizeof(float) * (size_t)s->G.cs * s->d.D, hipMemcpyDeviceToDevice, q));
  return 0;
}
int wl_sim_set_option(wl_sim* s, const char* name, int value) {
  const std::string n(name);
  if (n == "convz") { s->use_convz = value != 0; return 0; }
  if (n == "fused_smoother") { s->mg->use_fused = value != 0; return 0; }
  if (n == "store_eps") { s->mg->store_eps = value != 0; return 0; }
  if (n == "constl") { s->mg->use_constl = value != 0; return s->mg->update(0); }
  if (n == "tail") { s->mg->use_tail = value != 0; return 0; }
  if (n == "tail_lds") { wl::tail_lds_enable(value); return 0; }
  if (n == "xdefer") { s->mg->use_xdefer = value != 0; return 0; }
  if (n == "overlap_smooth") { s->mg->overlap_smooth = value != 0; return 0; }
  if (n == "body_tile") { wl::conv_body_tile_enable(value); return 0; }
  if (n == "skip_fill") { s->mg->skip_fill = value != 0; return 0; }
  if (n == "defer_shift") { s->mg->defer_shift = value != 0; return 0; }
  if (n == "zsplit") {   // 0 off, 1 default size gate, 2 levels of any size, v >= 4: levels of at least v·2^20 cells (takes effect at the next update!)
    s->mg->use_zsplit = value != 0; s->mg->zsplit_min = value == 2 ? 0 : (value >= 4 ? (long)value << 20 : 16L << 20); return 0;
  }
  if (n == "hybrid") { s->use_hybrid = value != 0; return 0; }
  if (n == "farmask") { s->use_farmask = value != 0; return 0; }
  if (n == "store_f") { s->store_f = value != 0; return 0; }
  if (n == "overlap") { WL_TRY(s->sync_u(0)); s->use_overlap = value != 0; return 0; }
  if (n == "fuse_cfl") { s->use_fuse_cfl = value != 0; return 0; }
  if (n == "jacobi_march") { wl::jacobi_march_enable(value); return 0; }
  if (n == "convm") { wl::conv_march_enable(value); return 0; }
  if (n == "deep_halo") { s->mg->deep_halo = value != 0; return 0; }
  if (n == "x_halo") { if (value < 1 || value > s->G.k0) { wl_set_error("x_halo: 1 .. ghost depth of the slab"); return WL_EINVAL; } s->mg->x_halo_depth = value; return 0; }   // 1: the fused head stays off on z-slabs
  if (n == "bcfold") { s->use_bcfold = value; return 0; }   // bit 0: projection tails, bit 1: tiled conv_diff!+BDIM!
  if (n == "resjac") { s->use_resjac = value != 0; s->resjac_force_redo = value == 2 || value == 3; s->redo_unannounced = value == 3; return 0; }   // 2: always take the redo path (tests); 3: the same, unknown to the BC! deferral (tests: its flush before the two-kernel head)
  if (n == "resjac_min") { wl::resjac_enable(1, value); return 0; }                            // cells threshold of the fused head (tests: 0)
  if (n == "convt_min") { wl::conv_tile_min(value); return 0; }                               // tile-planes threshold of the tiled conv_diff! (tests: 0)
  if (n == "lazydt") { s->use_lazydt = value != 0; return 0; }                                 // wl_sim_mom_steps: between its steps Δt stays on the device until the next predictor is queued (default 1)
  if (n == "tailspec") { s->use_tailspec = value != 0; return 0; }                             // the projection tail is queued ahead of the solver's convergence read, gated by the device's break test (default 1)
  if (n == "headspec") { s->use_headspec = value != 0; return 0; }                             // the first V-cycle is queued behind the fused head before Σr is known (default 1)
  if (n == "bcdefer") { s->use_bcdefer = value != 0; return 0; }                               // mom_step!: BC! after the fused conv_diff!+BDIM! left to the projection (its head reads U on the wall-normal faces, its tail rewrites the boundary); default 1
  if (n == "tailfuse") { s->use_tailfuse = value != 0; return 0; }                             // mom_step!: the first projection's u −= L∇x + BC! inside the corrector's conv_diff! (default 0: no gain measured)
  if (n == "convf") { wl::conv_flux_enable(value != 0); return 0; }                            // 1: tiled conv_diff! evaluates every flux once (default), 0: k_conv_tile
  if (n == "convt") { wl::conv_tile_enable(value != 0, value > 1 ? value : 0); return 0; }   // 0 off, 1 on, >1: on with that z-chunk
  if (n == "pair") { wl::gsrb_pair_enable(value); return 0; }
  if (n == "fuse_p") { s->use_fuse_p = value != 0; return 0; }
  if (n == "zsplit_par") { s->mg->par_ranges = value != 0; return 0; }
  if (n == "itmx") { if (value < 1) { wl_set_error("itmx must be >= 1"); return WL_EINVAL; } s->itmx = value; return 0; }
  wl_set_error("unknown option " + n); return WL_EINVAL;
}
int wl_sim_update(wl_sim* s, void* st) { s->resjac_backoff = false; s->resjac_redo_run = 0; WL_TRY(s->refresh_body_mask(wl_stream(st))); return s->mg->update(wl_stream(st)); }

long wl_launch_count(void) { return g_wl_launches; }
// the size gates and kernel-family switches that wl_sim_set_option / wl_mg_set_fused keep PROCESS-wide (they select code, not results): back to the defaults
int wl_reset_process_options(void) {
  wl::resjac_enable(1, 6L << 20); wl::conv_tile_min(2048); wl::conv_tile_enable(1, 0); wl::conv_flux_enable(1); wl::tail_lds_enable(1); wl::conv_body_tile_enable(1);
  wl::gsrb_pair_enable(1); wl::jacobi_march_enable(1); wl::conv_march_enable(0);
  return 0;
}
int wl_sim_counter(wl_sim* s, const char* name, long* out) {
  WL_CHECK(s && name && out, "bad argument");
  const std::string n(name);
  if (n == "resjac") { *out = s->n_resjac; return 0; }
  if (n == "resjac_redo") { *out = s->n_resjac_redo; return 0; }
  if (n == "resjac_backoff") { *out = s->resjac_backoff ? 1 : 0; return 0; }
  if (n == "tailfuse") { *out = s->n_tailfuse; return 0; }
  if (n == "bcdefer") { *out = s->n_bcdefer; return 0; }
  if (n == "tailspec") { *out = s->n_tailspec; return 0; }
  if (n == "xdefer") { *out = s->mg->last_xdefer; return 0; }
  wl_set_error("unknown counter " + n); return WL_EINVAL;
}
int wl_sim_set_forcing(wl_sim* s, const float* U1, const float* a0, const float* a1) {
  const int D = s->d.D;
  if (U1) for (int c = 0; c < D; c++) s->d.uBC[c] = U1[c];
  s->forcing = a0 != nullptr || a1 != nullptr;
  for (int c = 0; c < 3; c++) { s->acc0[c] = (a0 && c < D) ? a0[c] : 0.f; s->acc1[c] = (a1 && c < D) ? a1[c] : 0.f; }
  return 0;
}
int wl_accelerate(float* r, const wl_grid* g, const float* a, void* st) {
  WL_CHECK(wl_grid_ok(g) && a, "bad argument"); WL_TRY(wl_ctx_ensure());
  return wl::accelerate(r, gx(*g), a, wl_stream(st));
}
int wl_sim_mom_step(wl_sim* s, void* st) { return s->mom_step(wl_stream(st)); }
int wl_sim_mom_steps(wl_sim* s, int n, void* st) {
  WL_CHECK(s && n >= 0, "bad argument");
  for (int k = 0; k < n; k++) WL_TRY(s->mom_step(wl_stream(st), k + 1 < n));
  return 0;
}
int wl_sim_dt(const wl_sim* s, float* out, int cap) { const int n = (int)s->dt.size(); for (int k = 0; k < n && k < cap; k++) out[k] = s->dt[(size_t)k]; return n; }
float wl_sim_dt_last(const wl_sim* s) { return s->dt.back(); }
int wl_sim_set_dt_last(wl_sim* s, float dt) { WL_CHECK(s && dt > 0.f, "bad Δt"); s->dt.back() = dt; return 0; }
double wl_sim_time(const wl_sim* s) { float t = 0.f; for (size_t k = 0; k + 1 < s->dt.size(); k++) t += s->dt[k]; return (double)t; }
int wl_sim_phase(wl_sim* s, int phase, void* st) {
  hipStream_t q = wl_stream(st);
  switch (phase) {
    case 0: WL_TRY(s->sync_u(q)); WL_HIP(hipMemcpyAsync(s->u0, s->u, sizeof(float) * (size_t)s->G.cs * s->d.D, hipMemcpyDeviceToDevice, q)); return wl::scale_u(s->u, s->G, 0.f, q);
    case 1: return s->predict(q);
    case 2: return s->project(1.f, q);
    case 3: return s->correct(q);
    case 4: return s->project(0.5f, q);
    case 5: return s->cfl(q);
  }
  wl_set_error("bad phase"); return WL_EINVAL;
}
int wl_sim_apply_ic(wl_sim* s, int kind, void* st) {
  hipStream_t q = wl_stream(st); const GridX& G = s->G; const int D = s->d.D;
  if (kind == 0) { DSEL(D, k_apply_const, wl_plane_grid(G, G.nz), dim3(WL_BLOCK), 0, q, G, s->u, s->d.uBC[0], s->d.uBC[1], s->d.uBC[2]); }
  else {
    const float two = (kind == 2) ? 2.f : 1.f;
    const float kx = two * 3.14159265358979323846f / (float)s->d.dims[0], ky = two * 3.14159265358979323846f / (float)s->d.dims[1];
    const float kz = (D == 3) ? two * 3.14159265358979323846f / (float)s->d.dims[2] : 0.f;
    DSEL(D, k_apply_tgv, wl_plane_grid(G, G.nz), dim3(WL_BLOCK), 0, q, G, s->u, kx, ky, kz);
  }
  WL_LAUNCH_CHECK(); return 0;
}
static int to_body_arg(int D, const wl_body* b, BodyArg* o) {
  WL_CHECK(b && (b->kind == WL_BODY_SPHERE || b->kind == WL_BODY_PLANE), "wl_body.kind must be WL_BODY_SPHERE or WL_BODY_PLANE");
  o->kind = b->kind; o->R = b->R;
  float mm = 0.f;
  for (int q = 0; q < 3; q++) { o->c[q] = q < D ? b->c[q] : 0.f; o->m[q] = q < D ? b->m[q] : 0.f; o->V[q] = q < D ? b->V[q] : 0.f; mm += o->m[q] * o->m[q]; }
  WL_CHECK(mm > 0.f, "wl_body.m (axis mask / plane normal) is zero");
  return 0;
}
static wl_body sphere_body(const float* c, float R) {
  wl_body b{}; b.kind = WL_BODY_SPHERE; b.R = R;
  for (int q = 0; q < 3; q++) { b.c[q] = c[q]; b.m[q] = 1.f; }
  return b;
}
// measure!(flow,body;ϵ) on the caller's arrays (without the halo exchange / update!(pois) of the composite)   src/Body.jl:28-51
static int measure_fields(float* sigma, float* mu0, float* mu1, float* V, const GridX& G, const BodyArg& bd, float eps, int exitBC, unsigned perdir, hipStream_t q) {
  const int D = G.D; const size_t nc = (size_t)G.cs;
  WL_TRY(wl::fill(V, 0.f, nc * D, q)); WL_TRY(wl::fill(mu0, 1.f, nc * D, q)); WL_TRY(wl::fill(mu1, 0.f, nc * D * D, q));             // Body.jl:29
  DSEL(D, k_measure_body, wl_plane_grid(G, G.k1 - G.k0), dim3(WL_BLOCK), 0, q, G, sigma, mu0, mu1, V, bd, eps);
  WL_LAUNCH_CHECK();
  const float zero[3] = {0, 0, 0};
  WL_TRY(wl::bc_vec(mu0, G, zero, 0, perdir, q));                                                                                   // Body.jl:49
  return wl::bc_vec(V, G, zero, exitBC, perdir, q);                                                                                 // Body.jl:50
}
// which: 0 pressure_force(p) (src/Metrics.jl:116-133), 1 viscous_force(u,ν) (:140-154); Float64 partial sums, flow.f untouched.
// On z-slabs every rank sums its own planes and the per-rank sums are added on device (one 128-byte all-gather).
static int force_reduce(int which, const float* a, float nu, const GridX& G, const BodyArg& bd, const RedWs& ws, wl_comm* comm, double* out, hipStream_t q, const float* x0 = nullptr) {
  const int D = G.D;
  MomArg mo{}; if (x0) { mo.on = 1; for (int c = 0; c < D; c++) mo.x0[c] = x0[c]; }
  dim3 grid = wl_plane_grid(G, wl_red_slots(G, G.k1 - G.k0));
  // partials need 3*grid.x doubles (<= 3*WL_REDPART): pa and pb are contiguous (2*WL_MAXPART doubles)
  if (which == 0) { DSEL(D, k_pforce_body, grid, dim3(WL_BLOCK), 0, q, G, a, bd, mo, ws.pa); }
  else { DSEL(D, k_vforce_body, grid, dim3(WL_BLOCK), 0, q, G, a, nu, bd, mo, ws.pa); }
  hipLaunchKernelGGL(k_fin3, dim3(1), dim3(WL_BLOCK), 0, q, ws.pa, (int)grid.x, ws.res_d + 4);
  WL_LAUNCH_CHECK();
  WL_TRY(wl::combine_results(comm, ws, q));
  std::lock_guard<std::mutex> lock(wl::wl_read_mutex());
  WlCtx& cx = wl_ctx();
  WL_HIP(hipMemcpyAsync(cx.h_d, ws.res_d + 4, 3 * sizeof(double), hipMemcpyDeviceToHost, q));
  WL_HIP(hipStreamSynchronize(q));
  for (int c = 0; c < D; c++) out[c] = cx.h_d[c];
  return 0;
}
int wl_measure_body(float* sigma, float* mu0, float* mu1, float* V, const wl_grid* g, const wl_body* body, float eps, int exitBC, uint32_t perdir_mask, void* st) {
  WL_CHECK(wl_grid_ok(g), "bad wl_grid"); WL_CHECK(sigma && mu0 && mu1 && V, "null field");
  BodyArg bd; WL_TRY(to_body_arg(g->D, body, &bd));
  return measure_fields(sigma, mu0, mu1, V, gx(*g), bd, eps, exitBC, perdir_mask, wl_stream(st));
}
int wl_pressure_force_body(const float* p, const wl_grid* g, const wl_body* body, double* out, void* st) {
  WL_CHECK(wl_grid_ok(g), "bad wl_grid"); WL_TRY(wl_ctx_ensure());
  BodyArg bd; WL_TRY(to_body_arg(g->D, body, &bd));
  return force_reduce(0, p, 0.f, gx(*g), bd, wl_red_ws(wl_ctx().red), nullptr, out, wl_stream(st));
}
int wl_viscous_force_body(const float* u, const wl_grid* g, float nu, const wl_body* body, double* out, void* st) {
  WL_CHECK(wl_grid_ok(g), "bad wl_grid"); WL_TRY(wl_ctx_ensure());
  BodyArg bd; WL_TRY(to_body_arg(g->D, body, &bd));
  return force_reduce(1, u, nu, gx(*g), bd, wl_red_ws(wl_ctx().red), nullptr, out, wl_stream(st));
}
int wl_pressure_moment_body(const float* x0, const float* p, const wl_grid* g, const wl_body* body, double* out, void* st) {
  WL_CHECK(wl_grid_ok(g) && x0, "bad wl_grid / x0"); WL_TRY(wl_ctx_ensure());
  BodyArg bd; WL_TRY(to_body_arg(g->D, body, &bd));
  return force_reduce(0, p, 0.f, gx(*g), bd, wl_red_ws(wl_ctx().red), nullptr, out, wl_stream(st), x0);
}
int wl_viscous_moment_body(const float* x0, const float* u, const wl_grid* g, float nu, const wl_body* body, double* out, void* st) {
  WL_CHECK(wl_grid_ok(g) && x0, "bad wl_grid / x0"); WL_TRY(wl_ctx_ensure());
  BodyArg bd; WL_TRY(to_body_arg(g->D, body, &bd));
  return force_reduce(1, u, nu, gx(*g), bd, wl_red_ws(wl_ctx().red), nullptr, out, wl_stream(st), x0);
}
int wl_sim_pressure_moment_body(wl_sim* s, const float* x0, const wl_body* body, double* out, void* st) {
  WL_CHECK(x0, "null x0");
  BodyArg bd; WL_TRY(to_body_arg(s->d.D, body, &bd));
  return force_reduce(0, s->p, 0.f, s->G, bd, s->mg->ws, s->comm, out, wl_stream(st), x0);
}
int wl_sim_viscous_moment_body(wl_sim* s, const float* x0, const wl_body* body, double* out, void* st) {
  WL_CHECK(x0, "null x0");
  BodyArg bd; WL_TRY(to_body_arg(s->d.D, body, &bd));
  WL_TRY(s->sync_u(wl_stream(st)));
  return force_reduce(1, s->u, s->d.nu, s->G, bd, s->mg->ws, s->comm, out, wl_stream(st), x0);
}
int wl_sim_measure_body(wl_sim* s, const wl_body* body, float eps, void* st) {
  WL_CHECK(s->d.has_body && s->mu1 && s->V, "simulation was created with has_body=0");
  BodyArg bd; WL_TRY(to_body_arg(s->d.D, body, &bd));
  hipStream_t q = wl_stream(st); const GridX& G = s->G; const int D = s->d.D;
  WL_TRY(measure_fields(s->sigma, s->mu0, s->mu1, s->V, G, bd, eps, s->d.exitBC, s->d.perdir_mask, q));
  WL_TRY(wl::halo(s->comm, s->mu0, G, D, 2, q)); WL_TRY(wl::halo(s->comm, s->V, G, D, 2, q));
  WL_TRY(s->refresh_body_mask(q));
  return s->mg->update(q);                                                                                                          // WaterLily.jl:148
}
int wl_sim_pressure_force_body(wl_sim* s, const wl_body* body, double* out, void* st) {
  BodyArg bd; WL_TRY(to_body_arg(s->d.D, body, &bd));
  return force_reduce(0, s->p, 0.f, s->G, bd, s->mg->ws, s->comm, out, wl_stream(st));
}
int wl_sim_viscous_force_body(wl_sim* s, const wl_body* body, double* out, void* st) {
  BodyArg bd; WL_TRY(to_body_arg(s->d.D, body, &bd));
  WL_TRY(s->sync_u(wl_stream(st)));                  // ∂u/∂z at the slab faces reads the neighbours' planes
  return force_reduce(1, s->u, s->d.nu, s->G, bd, s->mg->ws, s->comm, out, wl_stream(st));
}
int wl_sim_measure_sphere(wl_sim* s, const float* c, float R, float eps, void* st) { const wl_body b = sphere_body(c, R); return wl_sim_measure_body(s, &b, eps, st); }
int wl_sim_pressure_force_sphere(wl_sim* s, const float* c, float R, double* out, void* st) { const wl_body b = sphere_body(c, R); return wl_sim_pressure_force_body(s, &b, out, st); }
int wl_sim_viscous_force_sphere(wl_sim* s, const float* c, float R, double* out, void* st) { const wl_body b = sphere_body(c, R); return wl_sim_viscous_force_body(s, &b, out, st); }
}  // extern "C"
