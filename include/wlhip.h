/* wlhip.h — C ABI of libwlhip.so: the MI355X (gfx950) backend for WaterLily's time-step hot path.
 *
 * Boundary being replaced: the reference has no FFI — it selects a backend by Julia multiple dispatch
 * on the array type (`Simulation(...; mem=ArrayType)`, /root/reference/src/WaterLily.jl:67-68,98;
 * src/Flow.jl:133-146) and generates every kernel from `@loop` through KernelAbstractions
 * (src/core.jl:125-156).  This header is what a Julia package extension binds with `ccall` for a new
 * array type: every entry point cites the reference function it stands in for.
 *
 * Conventions
 *  - Arrays are dense, column-major, x fastest, vector component slowest — byte-identical to the Julia
 *    arrays (scalar (Ng...), vector (Ng...,D), tensor (Ng...,D,D)), so `pointer(a)` is passed with no copy.
 *  - All pointers are DEVICE pointers unless named host_*.  Element type: Float32.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *    unless it returns a host scalar, in which case it synchronises that stream.
 *  - Return value: 0 = ok; >0 = hipError_t; <0 = library error (WL_E*).  wl_last_error_string() gives text.
 *    No exception crosses this boundary.
 *  - Aliasing pois.x≡flow.p, pois.L≡flow.μ₀, pois.z≡flow.σ (src/WaterLily.jl:97) is allowed everywhere.
 */
#ifndef WLHIP_H
#define WLHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WL_EINVAL (-1)   /* bad argument (the reference's @assert failures map here) */
#define WL_ENOGPU (-2)   /* no usable HIP device */
#define WL_ELEVELS (-3)  /* "MultiLevelPoisson requires size=a2ⁿ, where n>2" (src/MultiLevelPoisson.jl:73-74) */
#define WL_ECOMM (-4)    /* halo/collective failure */

/* Grid descriptor of one (slab of a) Cartesian array with one ghost layer per side in x,y and
 * `k0` ghost planes in z.  A single-GPU array is the case gk=0, gnz=nz, k0=1, k1=nz-1.
 * For D==2: nz=1, k0=0, k1=1, gnz=1. */
typedef struct wl_grid {
  int32_t D;        /* 2 or 3 */
  int32_t nx, ny, nz; /* allocated extents INCLUDING ghosts */
  int32_t k0, k1;   /* local z planes [k0,k1) are the interior planes this rank owns */
  int32_t gk;       /* global z index (0-based, ghosts included) of local plane 0 */
  int32_t gnz;      /* global z extent including ghosts */
} wl_grid;

/* convective schemes λ(u,c,d): src/Flow.jl:4-6 */
enum { WL_QUICK = 0, WL_VANLEER = 1, WL_CDS = 2 };

/* ---- lifecycle ---------------------------------------------------------------------------- */
int wl_init(int device);                       /* hipSetDevice + capability check (gfx950)           */
const char* wl_last_error_string(void);
int wl_version(void);
int wl_malloc(void** p, size_t bytes);          /* device allocation owned by the caller (Julia finalizer -> wl_free) */
int wl_free(void* p);
int wl_h2d(void* dst, const void* host_src, size_t bytes, void* stream);   /* `mem(::Array)` ctor, src/Flow.jl:139,143-144 */
int wl_d2h(void* host_dst, const void* src, size_t bytes, void* stream);   /* `Array(a)`                      */
int wl_d2d(void* dst, const void* src, size_t bytes, void* stream);        /* copy / `u⁰ .= u`, src/Flow.jl:157 */
int wl_stream_sync(void* stream);
wl_grid wl_grid_single(int D, const int32_t* dims_with_ghosts);            /* host helper: single-domain descriptor */

/* ---- generic array ops the array type must support ------------------------------ */
int wl_fill(float* a, float v, size_t n, void* stream);                    /* fill!, `r .= 0` src/Flow.jl:39, MultiLevelPoisson.jl:94 */
int wl_scale(float* a, float s, size_t n, void* stream);                   /* `x .*= dt` src/Flow.jl:225             */
int wl_div_scalar(float* a, float s, size_t n, void* stream);              /* `x ./= dt` src/Flow.jl:230             */
int wl_sum(const float* a, size_t n, double* host_out, void* stream);      /* sum(a)     src/Poisson.jl:95            */
int wl_sum_abs_max_abs(const float* a, size_t n, double* host_l1, float* host_linf, void* stream); /* L₁,L∞ src/Poisson.jl:190-191 */
int wl_max(const float* a, size_t n, float* host_out, void* stream);       /* maximum(a) src/Flow.jl:236              */
int wl_dot(const float* a, const float* b, size_t n, double* host_out, void* stream); /* a⋅b  src/Poisson.jl:156,189   */
int wl_L2_inside(const float* a, const wl_grid* g, double* host_out, void* stream);   /* L₂(a) src/Poisson.jl:188; ext/WaterLilyAMDGPUExt.jl:18 */

/* ---- boundary conditions: src/core.jl:200-243 ------------------------------------------------ */
/* BC!(a,U::tuple,saveexit,perdir): all faces, all components, one launch (edge/corner values equal the
 * reference's sequential (i,j) order).  perdir_mask bit (j-1) set = direction j periodic.            */
int wl_bc_vec(float* a, const wl_grid* g, const float* host_U, int saveexit, unsigned perdir_mask, void* stream);
/* Function-valued BCs and body forces that depend on position.  A C ABI cannot call a Julia closure, so the host
   tabulates it: Ub has the shape of `a` and holds uBC(i,loc(i,I),t) on the two outermost layers of every non-periodic direction
   (other cells are never read); G holds g(i,loc(i,I),t)+∂ₜuBC for every cell. */
int wl_bc_vec_fn(float* a, const float* Ub, const wl_grid* g, int saveexit, unsigned perdir_mask, void* stream);  /* BC!(a,uBC::Function,…) src/core.jl:201-219 */
int wl_accelerate_field(float* r, const float* G, const wl_grid* g, void* stream);                               /* accelerate!(r,t,g,U) src/Flow.jl:69-73 */
int wl_bc_per_scalar(float* a, const wl_grid* g, unsigned perdir_mask, void* stream);   /* perBC!   src/core.jl:239-243 */
int wl_exit_bc(float* u, const float* u0, const wl_grid* g, float dt, void* stream);    /* exitBC!  src/core.jl:226-233 */

/* ---- Flow: src/Flow.jl ------------------------------------------------------------------------ */
/* conv_diff!(r,u,Φ,λ;ν,perdir) :38-62 — gather form, one launch; Φ (=flow.σ) receives the same stale
 * ghost-plane fluxes the reference leaves there. Φ may be NULL.                 */
int wl_conv_diff(float* r, const float* u, float* Phi, const wl_grid* g, float nu, unsigned perdir_mask, int scheme, void* stream);
/* BDIM!(a) :176-180 with the neighbouring scale_u! folded in: u = (u*pre + μddn(μ₁,f) + V + μ₀ f)*post,
 * f = u⁰ + dt f − V on all cells.  mu1 may be NULL (== zeros, NoBody fast path), V may be NULL (== zeros). */
int wl_bdim(float* u, const float* u0, float* f, const float* V, const float* mu0, const float* mu1,
            const wl_grid* g, float dt, float pre, float post, void* stream);
int wl_scale_u(float* u, const wl_grid* g, float s, void* stream);                      /* scale_u! :211-214 */
int wl_div(float* z, const float* u, const wl_grid* g, void* stream);                   /* @inside z = div(I,u) :225 */
int wl_project(float* u, const float* L, const float* x, const wl_grid* g, void* stream); /* u[I,i] -= L[I,i]∂ᵢx :227-229 */
/* CFL(a) :234-237 — writes σ=flux_out on the interior and returns min(Δt_max, 1/(max(σ over ALL cells)+5ν)) */
int wl_cfl(const float* u, float* sigma, const wl_grid* g, float nu, float dt_max, float* host_dt, void* stream);

/* ---- Poisson leaf operations: src/Poisson.jl --------------------------------------------------- */
int wl_set_diag(float* D, float* iD, const float* L, const wl_grid* g, void* stream);   /* set_diag! :43-46 */
int wl_mult(float* z, const float* L, const float* D, const float* x, const wl_grid* g, void* stream); /* mult! :63-69 (no perBC, z ghosts zeroed) */
/* residual! :92-98 — r = iD==0 ? 0 : z−Ax; s=Σr/N; if |s|>2eps: r-=s.  No host sync.
 * scratch: device workspace of wl_reduce_workspace_bytes() bytes.                                   */
int wl_residual(float* r, const float* x, const float* z, const float* L, const float* D, const float* iD,
                const wl_grid* g, void* scratch, void* stream);
int wl_increment(float* r, float* x, const float* eps, const float* L, const float* D, const wl_grid* g, float omega, void* stream); /* increment! :100-104 */
int wl_jacobi(float* eps, float* r, float* x, const float* L, const float* D, const float* iD, const wl_grid* g, int it, float omega,
              unsigned perdir_mask, void* stream);                                      /* Jacobi! :111-114 */
int wl_gsrb(float* eps, float* r, float* x, const float* L, const float* D, const float* iD, const wl_grid* g, int it, float omega,
            unsigned perdir_mask, void* stream);                                        /* GaussSeidelRB! :141-148 */
int wl_norms(const float* r, const wl_grid* g, double* host_l1, float* host_linf, void* scratch, void* stream); /* L₁, L∞ :190-191 */
/* single-level solver.  z doubles as pcg!'s work array, exactly as p.z does in the reference. */
int wl_pcg(float* eps, float* r, float* x, float* z, const float* L, const float* D, const float* iD, const wl_grid* g, int it,
           unsigned perdir_mask, void* stream);                                         /* pcg!(p;it=6) :166-186 */
int wl_poisson_solve(float* eps, float* r, float* x, float* z, const float* L, const float* D, const float* iD, const wl_grid* g,
                     double tol, int itmx, unsigned perdir_mask, int* host_n, double* host_r1, float* host_rinf,
                     void* stream);                                                     /* solver!(p::Poisson;tol,itmx) :212-223 */
size_t wl_reduce_workspace_bytes(void);

/* ---- temporal averages: src/Metrics.jl:200-255 --------------------------------- */
/* update!(meanflow,flow) with the weight ε computed by the caller (:237-239); UU may be NULL (uu_stats=false) */
int wl_meanflow_update(float* P, float* U, float* UU, const float* p, const float* u, const wl_grid* g, float eps, void* stream);
int wl_meanflow_uu(float* tau, const float* UU, const float* U, const wl_grid* g, void* stream);   /* uu!(τ,a) :250-252 */

/* ---- multigrid transfer: src/MultiLevelPoisson.jl ---------------------------------------------- */
int wl_restrict(float* a_coarse, const wl_grid* gc, const float* b_fine, const wl_grid* gf, void* stream);    /* restrict! :49 */
int wl_prolongate(float* a_fine, const wl_grid* gf, const float* b_coarse, const wl_grid* gc, void* stream);  /* prolongate! :50 */
int wl_restrictL(float* a_coarse, const wl_grid* gc, const float* b_fine, const wl_grid* gf, unsigned perdir_mask, void* stream); /* restrictL! :42-48 */
int wl_coarsen_dims(int D, const int32_t* fine, int32_t* coarse);                      /* coarsen_mask/divisible :29,52; returns #dirs coarsened */

/* ---- MultiLevelPoisson handle: struct :61-77, update! :79-86, Vcycle! :88-101, solver! :108-128 -- */
typedef struct wl_mg wl_mg;
/* x, L, z are the caller's (aliased) level-1 arrays; coarse levels and r,ϵ,D,iD are owned by the handle. */
int wl_mg_create(wl_mg** out, float* x, float* L, float* z, const wl_grid* g, unsigned perdir_mask, int maxlevels);
int wl_mg_destroy(wl_mg* mg);
int wl_mg_update(wl_mg* mg, void* stream);
int wl_mg_nlevels(const wl_mg* mg);
int wl_mg_level_grid(const wl_mg* mg, int level, wl_grid* out);
/* name ∈ "L","D","iD","x","eps","r","z" — device pointer of that level's array (tests read pois.levels[k].D etc.) */
float* wl_mg_level_field(const wl_mg* mg, int level, const char* name);
int wl_mg_vcycle(wl_mg* mg, int level, float omega, void* stream);
int wl_mg_smooth(wl_mg* mg, int level, int it, float omega, void* stream);      /* smooth! = GaussSeidelRB! on one level (:106) */
int wl_mg_level_is_const(const wl_mg* mg, int level);   /* 1 if the level's L was verified to be 'constant inside, 0 on wall faces' */
int wl_mg_smoother_kind(const wl_mg* mg, int level);   /* how smooth! runs on the level: 0 one kernel per pass, 1 temporally blocked, 2 blocked pair kernels,
                                                          3 z-split (pair kernels away from the body, general blocked kernels near it) */
int wl_mg_set_fused(wl_mg* mg, int on);   /* bit0 (default 1): temporally blocked smoother on eligible levels, 0: one kernel per pass;
                                             bit1: do not store the final ϵ (scratch of the reference that nothing reads again);
                                             bit2: no pair kernels; bit3: no single-launch coarse tail; bit4: no z-split on body levels;
                                             bit5: coarse tail in global memory instead of LDS (process-wide switch); bit6: x increment of the prolongation not deferred to kernel B;
                                             bit7: z-slabs: the smoother's deep r exchange is not overlapped with kernel A's interior planes */
/* solver!(ml;tol,itmx): returns iterations in *host_n and the last L₁/L∞; appends to the n history. */
int wl_mg_solve(wl_mg* mg, double tol, int itmx, int* host_n, double* host_r1, float* host_rinf, void* stream);
int wl_mg_history(const wl_mg* mg, int16_t* host_out, int cap);                        /* pois.n :66 */
/* per-iteration log of the last solve: what `@log` prints (:112,117): r∞, r₁, ω */
int wl_mg_last_log(const wl_mg* mg, double* host_r1, double* host_rinf, double* host_omega, int cap);

/* ---- Simulation/Flow composite (what bench.py times): src/Flow.jl:156-167, src/WaterLily.jl:128-139 */
typedef struct wl_sim wl_sim;
typedef struct wl_sim_desc {
  int32_t D;
  int32_t dims[3];          /* interior cells N (no ghosts) */
  float uBC[3];             /* tuple boundary velocity */
  float nu, dt0;
  uint32_t perdir_mask;
  int32_t exitBC;
  int32_t scheme;           /* WL_QUICK ... */
  int32_t has_body;         /* 0: NoBody fast path (μ₁≡0, V≡0 never read) */
  /* caller-owned flow arrays (NULL => the handle allocates and owns them) */
  float *u, *u0, *f, *p, *sigma, *V, *mu0, *mu1;
  /* optional caller-owned SPARE velocity array (same size as u).  The fused kernels are out of place (conv_diff!+BDIM! and the
     projection tail write a velocity array other than the one they read) and `u⁰ .= u` (src/Flow.jl:157) is a pointer swap, so
     the handle PERMUTES the roles of the three arrays {u, u0, us} from step to step.  With caller-owned u and u0:
       us given : full-speed path; after every wl_sim_mom_step read the current roles back with wl_sim_field(s,"u"|"u0"|"us")
                  (the Julia binding re-points its HipArray objects; the three buffers stay the caller's to free);
       us NULL  : pointers never move — `u⁰ .= u` is a copy and conv_diff!/BDIM! run as separate passes (slower).
     Convective exit (exitBC=1, single domain): the buffers rotate as well; BC! leaves the x-exit face of u alone (saveexit, src/core.jl:207),
     so that face is copied from the old role holder to the new one at each rotation (one strided plane).  On z-slabs exitBC flows keep the copy. */
  float *us;
} wl_sim_desc;
int wl_sim_create(wl_sim** out, const wl_sim_desc* desc);
/* the same on an EXISTING multigrid handle (wl_mg_create on desc->p, desc->mu0, desc->sigma): the Simulation constructor of the
   reference builds the AbstractPoisson first (pois_ctor, src/WaterLily.jl:96-105) and mom_step!(flow,pois) receives both.  The wl_mg
   stays the caller's (destroy the wl_sim first).  Side effects on the adopted handle: the final ϵ of smooth! is no longer stored
   (wl_mg_set_fused bit1 — scratch nothing on the time-step path reads).  WL_EINVAL if the handle was not built on desc's p, mu0, sigma or
   with a different perdir mask. */
int wl_sim_create_on(wl_sim** out, const wl_sim_desc* desc, wl_mg* mg);
int wl_sim_destroy(wl_sim* s);
float* wl_sim_field(wl_sim* s, const char* name);       /* "u","u0","f","p","sigma","V","mu0","mu1","us" (current roles) */
wl_mg* wl_sim_pois(wl_sim* s);
int wl_sim_grid(const wl_sim* s, wl_grid* out);
int wl_sim_init_flow(wl_sim* s, void* stream);          /* BC!(u), u⁰=u, μ₀ BC, (src/Flow.jl:141-145) after the caller filled u */
/* implementation switches (the tests compare the variants bit for bit).  Defaults in brackets.
   "fused_smoother"[1] temporally blocked GaussSeidelRB!   "pair"[1] its two-cells-per-thread constant-coefficient variant
   "constl"[1] constant-coefficient (NoBody) kernels        "fuse_p"[1] fused projection head (div+scale+residual!) and tail
   "fuse_cfl"[1] CFL folded into the corrector's tail       "tail"[1] smallest V-cycle levels in one launch
   "store_f"[0] materialise the intermediates f, z          "store_eps"[0] materialise the smoother's final ϵ
   "overlap"[1] u exchange on a second stream (slabs)       "convz"[0], "convm"[0] alternative conv_diff! kernels (slower)
   "jacobi_march"[1] z-marching constant-coefficient Jacobi  "farmask"[1], "hybrid"[1] BDIM! fast paths away from a body
   "zsplit"[1] pair smoother on the planes away from a body (levels >= 16 M cells; 2: any size, v >= 4: >= v·2^20 cells)
   "defer_shift"[1] residual!'s mean shift + solver!'s first norms folded into the finest level's z-marching Jacobi!
   "skip_fill"[1] Vcycle!'s fill!(coarse.x,0) folded into the coarse level's Jacobi!
   "convt"[1] LDS-tiled z-marching conv_diff!+BDIM! (NoBody, no periodic direction, f not stored; v > 1: on with z-chunks of v planes)
   "bcfold"[1] BC!(u,U) for a tuple U folded into the stores of the kernels that produce u (single domain, no exit, no periodic direction):
   bit 0 the projection tails, bit 1 the tiled conv_diff!+BDIM! (measured slower: off by default)
   "resjac"[1] projection head (div, x·=dt, residual!) + the V-cycle's first Jacobi! in one launch on single-domain NoBody levels (the
   mean shift is checked on the host afterwards; if due, the two-kernel path is taken)   "resjac_min"[6 Mi cells] size gate (tests: 0)
   "lazydt"[1] wl_sim_mom_steps: see there
   "tailspec"[1] the projection tail is queued behind the smoother before the host has read that iteration's norms and gated on the device by solver!'s break test
       (the flag the host then takes its own decision from): it runs iff the iteration was the last one.  Single GPU, the in-place tail and the pair tail with CFL.
   "headspec"[1] the solver's first V-cycle is queued behind the fused projection head before Σr (residual!'s mean-shift test) has been read back — solver! runs at
       least one cycle whatever the norms are; Σr returns with the first iteration's norms, and if the shift was due after all that solve is discarded (inputs untouched)
       and the two-kernel path taken.  Single GPU.  One host round trip per solve fewer.
   "bcdefer"[1] wl_sim_mom_step: BC!(u,U) after the fused conv_diff!+BDIM! is left to the projection that follows — its fused head (and the second tail's flux_out) read U
       on the wall-normal boundary faces, its tail's folded stores rewrite every boundary location — two BC! launches fewer per step; results identical on every cell.
       Only where all of that holds (tuple U, single domain, no periodic direction / exit / body, fused head and folded tails in use); wl_sim_phase applies BC! as before.
   "tailfuse"[0] wl_sim_mom_step: the first projection's tail (u −= L∇x, BC!) is evaluated by the corrector's conv_diff! loader; the projected predictor
       velocity is never written (whole tiles, single domain, tuple U, no periodic direction / exit / body; results identical; measured: no gain, hence off).
       wl_sim_phase always keeps the tail launch.
   "convf"[1] the tiled conv_diff!+BDIM! evaluates every face flux once (wl_convf.hip); 0: the two-cells-per-thread kernel that re-evaluates upper faces
   "convt_min"[2048] tile-planes below which "convt" leaves the launch to the plane kernel (tests: 0)
   "xdefer"[1] pair smoother: the V-cycle's x += ω·x_c↓ is applied by kernel B together with its own increment (x makes one round trip per smooth!)
   "tail_lds"[1] the single-launch coarse tail keeps r, x, ϵ of its levels in LDS (0: in global memory)
   "body_tile"[1] with a body: conv_diff!+BDIM! on the body-free plane ranges through the tiled NoBody kernel ("convt")
   "itmx"[32] solver!'s iteration cap `itmx` (src/MultiLevelPoisson.jl:108) */
int wl_sim_set_option(wl_sim* s, const char* name, int value);
/* "resjac_min", "convt_min", "convt", "convf", "tail_lds", "body_tile", "pair", "jacobi_march", "convm" (and wl_mg_set_fused bits 2 and 5) are PROCESS-wide:
   they choose between kernels that produce identical bits, for every handle of the process.  wl_reset_process_options() restores their defaults. */
int wl_reset_process_options(void);
/* time-dependent but spatially uniform boundary velocity / body force: the host evaluates uBC(i,t₁) and
   g(i,t)+dU(i,t)/dt at t₀ (predictor) and t₁ (corrector) before each mom_step! (src/Flow.jl:156-167, accelerate! :69-73).
   NULL U1 keeps the boundary velocity; NULL a0 and a1 switches the forcing off. */
int wl_sim_set_forcing(wl_sim* s, const float* U1, const float* a0, const float* a1);
int wl_accelerate(float* r, const wl_grid* g, const float a[3], void* stream);   /* accelerate! for a uniform acceleration: r[I,i] += a_i */
int wl_sim_update(wl_sim* s, void* stream);             /* update!(pois) after μ₀ changed (measure!, src/WaterLily.jl:148) */
int wl_sim_mom_step(wl_sim* s, void* stream);           /* mom_step!(flow,pois): appends Δt */
/* n × mom_step! in one call — the loop of sim_step!(sim,t_end) without measure! (src/WaterLily.jl:136-139 with remeasure=false).  Same results as n calls of
   wl_sim_mom_step; between its steps the library may keep Δt on the device until the next predictor has been queued (option "lazydt"): one host round trip per
   step fewer.  The Δt history is complete when the call returns. */
int wl_sim_mom_steps(wl_sim* s, int n, void* stream);
int wl_sim_dt(const wl_sim* s, float* host_out, int cap);      /* flow.Δt (host vector, src/Flow.jl:127) */
double wl_sim_time(const wl_sim* s);                    /* time(flow) = sum(Δt[1:end-1]) :174 */
float wl_sim_dt_last(const wl_sim* s);                 /* Δt[end] */
int wl_sim_set_dt_last(wl_sim* s, float dt);           /* Δt[end] = dt: the host owns flow.Δt (src/Flow.jl:127) and may have changed it */
/* sub-phases for parity tests: 0 u⁰.=u;scale_u!(0) 1 mom_predict! 2 mom_project!(1) 3 mom_correct! 4 mom_project!(.5) 5 push!(Δt,CFL) */
int wl_sim_phase(wl_sim* s, int phase, void* stream);
/* analytic initial conditions evaluated on device (apply!(u0,u), src/Flow.jl:81-83): kind 0 = uBC tuple,
 * 1 = wall-bounded 3-D TGV κ=π/N, 2 = periodic TGV κ=2π/N */
int wl_sim_apply_ic(wl_sim* s, int kind, void* stream);
/* measure!(flow, sphere(c,R); ϵ): closed-form AutoBody sphere/circle (src/Body.jl:28-51, src/AutoBody.jl:29-37) */
int wl_sim_measure_sphere(wl_sim* s, const float* host_center, float R, float eps, void* stream);
/* Closed-form AutoBody shapes (src/AutoBody.jl:21,29-37 with the gradient of the sdf written out; arbitrary Julia sdf/map closures
 * cannot cross a C ABI).  WL_BODY_SPHERE: sdf = |m∘(x−c)|−R — m = (1,1,1) the sphere/circle, an axis with m = 0 is dropped: the
 * cylinder along that axis.  WL_BODY_PLANE: sdf = m·(x−c), m the (not necessarily unit) normal pointing into the fluid.
 * A translating body map(x,t) = x − V·t: the caller passes the current c and V, measure! stores V in flow.V (AutoBody.jl:36-37). */
enum { WL_BODY_SPHERE = 1, WL_BODY_PLANE = 2 };
typedef struct wl_body { int32_t kind; float c[3]; float R; float m[3]; float V[3]; } wl_body;
/* the same on the caller's arrays — what measure!(a::Flow,body), pressure_force(sim), viscous_force(sim) bind for a closed-form body */
int wl_measure_body(float* sigma, float* mu0, float* mu1, float* V, const wl_grid* g, const wl_body* host_body, float eps, int exitBC, uint32_t perdir_mask, void* stream);
int wl_pressure_force_body(const float* p, const wl_grid* g, const wl_body* host_body, double out[3], void* stream);
int wl_viscous_force_body(const float* u, const wl_grid* g, float nu, const wl_body* host_body, double out[3], void* stream);
/* pressure_moment(x₀,p,df,body) / viscous_moment(x₀,u,ν,df,body) (src/Metrics.jl:169-188): moments about host_x0[D]; in 2-D the scalar
 * moment is returned in out[0] and out[1] (the reference's broadcast) */
int wl_pressure_moment_body(const float* host_x0, const float* p, const wl_grid* g, const wl_body* host_body, double out[3], void* stream);
int wl_viscous_moment_body(const float* host_x0, const float* u, const wl_grid* g, float nu, const wl_body* host_body, double out[3], void* stream);
int wl_sim_pressure_moment_body(wl_sim* s, const float* host_x0, const wl_body* host_body, double out[3], void* stream);
int wl_sim_viscous_moment_body(wl_sim* s, const float* host_x0, const wl_body* host_body, double out[3], void* stream);
int wl_sim_measure_body(wl_sim* s, const wl_body* host_body, float eps, void* stream);          /* measure! + update!(pois) */
int wl_sim_pressure_force_body(wl_sim* s, const wl_body* host_body, double out[3], void* stream); /* src/Metrics.jl:116-133 */
int wl_sim_viscous_force_body(wl_sim* s, const wl_body* host_body, double out[3], void* stream);  /* src/Metrics.jl:140-154 */
/* pressure_force(sim) for that sphere (src/Metrics.jl:116-133): Float64 accumulation, does not touch flow.f */
int wl_sim_pressure_force_sphere(wl_sim* s, const float* host_center, float R, double* host_out, void* stream);
int wl_sim_viscous_force_sphere(wl_sim* s, const float* center, float R, double* out, void* stream);   /* viscous_force(sim) src/Metrics.jl:140-154 (single domain) */

/* ---- multi-GPU: z-slab decomposition, one process per GPU (NEW — the reference has no multi-device path,
 * /root/reference/README.md:153-155).  A wl_comm carries the two primitives the slab path needs, stream-ordered:
 * a nearest-neighbour plane exchange along z and an all-gather; scalars (Σr, L₁, L∞, max σ) are combined on
 * device from an all-gather, so every rank takes identical control-flow decisions.  Two implementations:
 *   rccl      — ncclSend/ncclRecv groups + ncclAllGather on the compute stream (RCCL over xGMI), librccl.so.1
 *               resolved at run time so the process shares PyTorch's copy;
 *   callbacks — host function pointers (tests: torch.distributed/gloo staging through host memory).          */
typedef struct wl_comm wl_comm;
typedef int (*wl_sendrecv_fn)(void* ctx, const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes, void* stream);
typedef int (*wl_allgather_fn)(void* ctx, const void* send, void* recv, size_t bytes_each, void* stream);
int wl_comm_rccl_unique_id(char out128[128]);
int wl_comm_rccl_create(wl_comm** out, int rank, int size, const char uid128[128]);
/* 1 when librccl.so.1 and every entry point used here resolved (ranks agree on this BEFORE anyone calls ncclCommInitRank) */
int wl_comm_rccl_available(void);
/* second communicator (its own unique id) for the exchanges that run on the communicator's own HIP stream, overlapped with
 * stencil work: one ncclComm_t is then only ever used from one stream.  Collective: every rank calls it, after _create. */
int wl_comm_rccl_add_async(wl_comm* c, const char uid128[128]);
int wl_comm_callbacks_create(wl_comm** out, int rank, int size, void* ctx, wl_sendrecv_fn sendrecv, wl_allgather_fn allgather);
/* TEST mode of a ONE-rank communicator: both neighbours are this rank (z-periodic wrap onto itself), so that the transport
 * calls a one-rank run would skip (ncclSend/ncclRecv groups, in-place ncclAllGather, the scalar combine) execute on a one-GPU box */
int wl_comm_set_loopback(wl_comm* c, int on);
/* REHEARSAL mode of a ONE-rank RCCL communicator: it reports itself as rank `rank` of `size` — slab geometry, wall logic, exchange pattern and
 * byte counts are those of that rank in a `size`-GPU run — while the planes it sends to a neighbour come back as its own ghost planes and
 * all-gathers fill every rank's block with copies of its own.  The flow is not the P-rank flow (the slab sees itself as its neighbours);
 * what it measures on ONE GPU is the rank's compute per step, its exchange rounds/bytes, and the issue cost of the RCCL calls
 * (tools/slab_rank_bench.py -> profiles/r03_slab8_rank_compute.json).  Call right after wl_comm_rccl_create (+ _add_async). */
int wl_comm_set_virtual(wl_comm* c, int rank, int size);
/* rehearsal transport: 1 (default) = every exchange goes through RCCL to this rank itself; 0 = no transfer at all (ghost planes keep what they hold):
 * the difference between the two runs is what issuing the exchanges costs on this box, the second is the rank's pure compute */
int wl_comm_set_virtual_transport(wl_comm* c, int on);
/* z-periodic domain on z-slabs: the halo exchanges wrap around (rank 0's lower neighbour is the last rank).  wl_sim_create sets it from the
 * descriptor's perdir mask; callback transports get both neighbours on every rank and must address them modulo the size. */
int wl_comm_set_periodic(wl_comm* c, int on);
/* test hooks: the overlapped exchange (begin on the communicator's stream + wait) and the device-side scalar combine
 * (d8/f8: one 128-byte device record, 8 doubles then 8 floats; Σ / max over ranks in place) */
int wl_comm_halo_async(wl_comm* c, float* a, const wl_grid* g, int ncomp, int depth, void* stream);
int wl_comm_combine_test(wl_comm* c, double* d8, float* f8, void* stream);
int wl_comm_destroy(wl_comm* c);
int wl_comm_rank(const wl_comm* c);
int wl_comm_size(const wl_comm* c);
/* counters since creation: {halo exchanges, bytes this rank sent in them, scalar combines (all-gather of 128 B), plane all-gathers} */
int wl_comm_stats(const wl_comm* c, int64_t out4[4]);
/* exchange `depth` z-planes of an (ncomp)-component field with both neighbours (test hook; the composites call it internally) */
int wl_halo_exchange(wl_comm* c, float* a, const wl_grid* g, int ncomp, int depth, void* stream);
/* in-place all-gather of the planes [view->k0,view->k1) each rank computed of a replicated (full) array (test hook) */
int wl_allgather_planes(wl_comm* c, float* a, const wl_grid* view, int ncomp, void* stream);
/* slab grid of `rank` for a GLOBAL ghosted size; halo = ghost planes kept in z (2 for velocity-like fields) */
int wl_grid_slab(wl_grid* out, int D, const int32_t* global_dims_with_ghosts, int rank, int size, int halo);
/* Simulation on a z-slab: desc->dims are the GLOBAL interior sizes; arrays are allocated by the handle (caller pointers must be NULL) */
int wl_sim_create_slab(wl_sim** out, const wl_sim_desc* desc, wl_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* WLHIP_H */
