// MultiLevelPoisson handle (struct src/MultiLevelPoisson.jl:61-77) — internal C++ definition behind `wl_mg`.
#pragma once
#include <functional>
#include <vector>

#include "wl_comm.hpp"

struct wl_mg {
  struct Level {            // one `Poisson` (src/Poisson.jl:22-39)
    wl_grid g; GridX x_;
    float *L = nullptr, *D = nullptr, *iD = nullptr, *x = nullptr, *eps = nullptr, *r = nullptr, *z = nullptr;
    float *em = nullptr, *rs = nullptr;   // scratch of the fused smoother: ϵ after sweep 2, new residual (ghosts stay zero)
    wl::ConstL cl{};                 // constant-coefficient level (verified at update!)
    bool xzero = false;              // x ≡ 0 is implied (the V-cycle's fill!(x,0) was skipped): the next Jacobi! writes x instead of updating it
    // body levels: the coefficients deviate from the constant pattern only on planes [za,zb]; smooth! runs the pair kernels on the other planes
    bool part = false; int za = 0, zb = -1; wl::ConstL clp{};
    bool pend = false;       // the V-cycle's prolongate!+increment! of this level is deferred into the next smooth! (fused kernel A)
    bool dist = false;       // z-slab distributed level (halo exchanges) vs replicated on every rank
    GridX view;              // replicated level fed by a distributed parent: the planes of the full array this rank computes
    bool has_view = false;
  };
  wl_comm* comm = nullptr;   // not owned
  std::vector<Level> lv;
  std::vector<int16_t> n;   // pois.n :66
  unsigned perdir = 0;
  bool use_constl = true;   // allow the constant-coefficient specialisations where the pattern is verified
  bool store_eps = true;    // the blocked smoother also stores the final ϵ (p.ϵ of the reference); the mom_step! composite turns it off
  bool use_fused = true;    // temporally blocked GaussSeidelRB! on eligible levels (wl_fused.hip)
  bool use_zsplit = true;   // body levels: constant-coefficient pair kernels on the planes away from the body, general kernels on the rest
  long zsplit_min = 16L << 20;   // ... on levels of at least this many cells (smaller ranges do not fill 256 CUs; with the 16-row pair tiles a 256³ level gains 2 %, 384³ 4 %, 512³ 7 %; 128³ levels lose)
  bool skip_fill = true;    // Vcycle!'s fill!(coarse.x,0) folded into the coarse level's Jacobi! (x = ω·ϵ instead of x += ω·ϵ)
  bool defer_shift = true;  // residual!'s mean shift and solver!'s first norms are folded into the finest level's Jacobi! (z-march kernel) when that is what runs next
  bool shift_pending = false;
  bool deep_halo = true;     // z-slabs with >= 5 ghost planes: one r exchange (5 planes) per smooth! instead of r (2) + ϵ_mid (3) + r' (2)
  // One-shot hook of the next solve(): the projection tail, queued behind every iteration's smoother BEFORE the host reads that iteration's norms and gated on the
  // device by the break test (wl::decide_converged → res_f[4], also the host's decision): no idle GPU while the host decides, nothing happens if the loop goes on.
  // spec_check_head: the flag of the first iteration also requires the fused head's mean-shift test to pass (wl_sim's early V-cycle).  tail_stood: the tail ran.
  std::function<int(const float*)> spec_tail; bool spec_check_head = false, tail_stood = false;
  hipStream_t side = nullptr; hipEvent_t ev_decided = nullptr;   // the host waits for the copy of such an iteration's norms (this event), not for the tail queued behind it
  double first_hd0 = 0.0;   // res_d[0] as the first iteration's read found it (the fused head's Σr when its check is deferred: wl_sim)
  bool jacobi0_done = false; // the fused projection head (wl_resjac.hip) already ran the V-cycle's first Jacobi! on the finest level and left solver!'s first norms
  int norm_slots = 0;       // z-split smoother: which plane ranges left an (L₁, L∞) pair in their own result slots
  bool par_ranges = false;  // levels with a body: the plane ranges of the z-split on concurrent streams (wl::par_fork / par_join) — measured SLOWER (sphere 256³ 2.84 -> 3.07 ms: the fork/join events cost more than the overlap of 35–76 µs launches returns); "zsplit_par" turns it on
  int x_halo_depth = 1;     // z-slabs: ghost planes of x refreshed at the end of solver! (the projection tail reads 1; the fused projection head of the NEXT solve reads 2)
  int last_xdefer = -1;     // what the finest level's last smooth! with a pending prolongation decided: 1 = x += ω·x_c↓ deferred to kernel B, 0 = applied by kernel A (−1: none yet)
  bool use_xdefer = true;   // pair smoother: the V-cycle's `x += ω·x_c↓` is applied by kernel B together with its own increment (wl::XDefer)
  bool overlap_smooth = true;   // z-slabs: the one deep r exchange of a smooth! overlaps kernel A's interior planes (boundary slices after the wait)
  bool use_tail = true;     // levels of <= WL_TAIL_CELLS cells: the rest of the V-cycle in one launch (k_vcycle_tail)
  bool tail_ok(int first) const;
  int tail(int first, float w, hipStream_t s);
  float* slab = nullptr;    // owns r,ϵ,D,iD of every level and L,x,z of the coarse levels
  void* red = nullptr;      // reduction workspace
  RedWs ws;
  std::vector<double> log_r1, log_rinf, log_w;

  int build(float* x, float* L, float* z, const wl_grid& g0, unsigned per, int maxlevels, wl_comm* c = nullptr);
  int halo(Level& v, float* a, int ncomp, hipStream_t s, int depth = 1, bool wrap = true) { return v.dist ? wl::halo(comm, a, v.x_, ncomp, depth, s, wrap) : 0; }
  // a distributed level whose smooth! runs as the blocked pair kernels (constant coefficients, 3 ghost planes)
  bool pair_slab(const Level& v) const { return v.dist && v.g.k0 >= 3 && use_fused && !perdir && wl::gsrb_pair_ok(v.x_, v.cl); }
  ~wl_mg();
  int update(hipStream_t s);
  int smooth(int l, int it, float w, hipStream_t s, bool want_norms = false, bool* norms_done = nullptr);
  int vcycle(int l, float w, hipStream_t s, bool defer = false);
  int flush_pending(int l, float w, hipStream_t s);
  // have_residual: r and the local Σr (ws.res_d[0]) were already produced by the caller's fused div+residual kernel
  // pre_r1 / pre_rinf: the norms of the initial residual are already on the host (the fused projection head's read-back, combined over ranks)
  int solve(double tol, int itmx, int* host_n, double* host_r1, float* host_rinf, hipStream_t s, bool have_residual = false, const double* pre_r1 = nullptr, const float* pre_rinf = nullptr);
};
