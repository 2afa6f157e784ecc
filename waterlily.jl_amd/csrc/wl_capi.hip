// C ABI of libwlhip.so (see include/wlhip.h): context, leaf wrappers and the MultiLevelPoisson handle.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "wl_common.hpp"
#include <algorithm>

#include "wl_mg.hpp"

static thread_local std::string g_err;
void wl_set_error(const std::string& s) { g_err = s; }

WlCtx& wl_ctx() { static WlCtx c; return c; }
int wl_ctx_ensure() {
  WlCtx& c = wl_ctx();
  if (c.inited) return 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { wl_set_error("libwlhip: no HIP device visible — the HIP path has no CPU fallback"); return WL_ENOGPU; }
  WL_HIP(hipGetDevice(&c.device));
  WL_HIP(hipMalloc(&c.red, wl_red_bytes()));
  // one pinned 128-byte record, laid out like RedWs' device record (8 doubles, then floats): both halves come back in ONE copy
  WL_HIP(hipHostMalloc((void**)&c.h_d, 128, hipHostMallocDefault));
  c.h_f = (float*)((char*)c.h_d + 64);
  c.inited = true;
  return 0;
}

long g_wl_launches = 0;
// two auxiliary streams for launches that are independent of each other (plane ranges of a level with a body): fork/join with events
namespace wl {
namespace { hipStream_t g_par[2] = {nullptr, nullptr}; hipEvent_t g_par_fork = nullptr, g_par_join[2] = {nullptr, nullptr}; int g_par_state = 0; }
bool par_streams_ok() {
  if (g_par_state == 0) {
    g_par_state = -1;
    if (hipStreamCreateWithFlags(&g_par[0], hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&g_par[1], hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&g_par_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&g_par_join[0], hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&g_par_join[1], hipEventDisableTiming) == hipSuccess) g_par_state = 1;
  }
  return g_par_state == 1;
}
hipStream_t par_stream(int i) { return g_par[i]; }
int par_fork(hipStream_t s) {
  WL_HIP(hipEventRecord(g_par_fork, s));
  WL_HIP(hipStreamWaitEvent(g_par[0], g_par_fork, 0)); WL_HIP(hipStreamWaitEvent(g_par[1], g_par_fork, 0));
  return 0;
}
int par_join(hipStream_t s) {
  for (int i = 0; i < 2; i++) { WL_HIP(hipEventRecord(g_par_join[i], g_par[i])); WL_HIP(hipStreamWaitEvent(s, g_par_join[i], 0)); }
  return 0;
}
}  // namespace wl
WlProf& wl_prof() { static WlProf p; return p; }
ProfScope::ProfScope(int id_, hipStream_t s_) : id(id_), s(s_), active(false), idx(0) {
  WlProf& p = wl_prof();
  if (!p.on || id < 0) return;
  if (p.only_roofline && id != WL_PROF_GS_A && id != WL_PROF_GS_B) return;
  WlProf::Slot& sl = p.slot[id];
  if (sl.used >= 16384) return;
  if (sl.used >= sl.a.size()) { hipEvent_t ea, eb; if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return; sl.a.push_back(ea); sl.b.push_back(eb); }
  idx = sl.used++; active = true;
  (void)hipEventRecord(sl.a[idx], s);
}
ProfScope::~ProfScope() { if (active) (void)hipEventRecord(wl_prof().slot[id].b[idx], s); }

// ================================================================================================
// MultiLevelPoisson handle
// ================================================================================================
static inline bool divisible(int n) { return (n % 2 == 0) && n > 4; }   // src/MultiLevelPoisson.jl:52

int wl_mg::build(float* x, float* L, float* z, const wl_grid& g0, unsigned per, int maxlevels, wl_comm* c) {
  perdir = per; comm = (c && c->size > 1) ? c : nullptr;
  WL_TRY(wl_ctx_ensure());
  WL_HIP(hipMalloc(&red, wl_red_bytes()));
  ws = wl_red_ws(red);
  const bool dist0 = comm && g0.D == 3 && g0.nz != g0.gnz;
  if (dist0 && ((per >> 2) & 1u) && !comm->zperiodic) { wl_set_error("z-periodic z-slabs need a communicator in periodic mode (wl_comm_set_periodic)"); return WL_EINVAL; }
  // level 1 aliases the caller's arrays; r,ϵ,D,iD owned                                 src/Poisson.jl:32-38
  std::vector<wl_grid> grids; std::vector<char> isdist; std::vector<wl_grid> views; std::vector<char> hasview;
  grids.push_back(g0); isdist.push_back(dist0); views.push_back(g0); hasview.push_back(0);
  while ((int)grids.size() <= maxlevels) {                                               // :70
    const wl_grid f = grids.back(); const bool fd = isdist.back();
    const bool cx = divisible(f.nx), cy = divisible(f.ny), cz = (f.D == 3) && divisible(f.gnz);
    if (!(cx || cy || cz)) break;                                                        // divisible(l) :54
    wl_grid cgr = f; wl_grid vw = f; bool cd = fd, hv = false;
    if (cx) cgr.nx = 1 + f.nx / 2;                                                       // restrictML :36
    if (cy) cgr.ny = 1 + f.ny / 2;
    const int gnz_c = cz ? 1 + f.gnz / 2 : f.gnz;
    if (!fd) { if (f.D == 3) { cgr.gnz = gnz_c; cgr.nz = gnz_c; cgr.k0 = 1; cgr.k1 = gnz_c - 1; cgr.gk = 0; } }
    else {
      const int nloc = f.k1 - f.k0;
      const int nc = cz ? nloc / 2 : nloc;
      // stay distributed while the local planes pair up and the level is still big; otherwise replicate on every rank
      // levels of <= 64 planes are replicated: their halo exchanges would be pure latency (7 per V-cycle and level) while the
      // whole level costs less than that to recompute on every rank
      // (WL_REPLICATE_PLANES overrides the 64: the tests use it to build, with 4 ranks, the small distributed slabs an 8-rank run has)
      static const int repl = [] { const char* e = getenv("WL_REPLICATE_PLANES"); const int v = e ? atoi(e) : 64; return v >= 8 ? v : 64; }();
      // … and a level whose local plane count is odd cannot be coarsened slab by slab (its plane pairs would straddle the ranks): the level that
      // would come out odd AND is coarsened again in z is replicated right away — any nz with an even number of planes per rank works, not only P·2^k
      const bool odd_next = divisible(gnz_c) && (nc % 2 != 0);
      const bool keep = (!cz || (nloc % 2 == 0)) && nc >= 1 && (gnz_c - 2) > repl && !odd_next;
      if (cz && (nloc % 2 != 0)) { wl_set_error("z-slab: the finest level needs an even number of planes per rank"); return WL_EINVAL; }
      if (keep) {
        cgr.gnz = gnz_c; cgr.k0 = f.k0; cgr.k1 = cgr.k0 + nc; cgr.nz = nc + 2 * cgr.k0;
        cgr.gk = (cz ? (f.gk + f.k0 + 1) / 2 : f.gk + f.k0) - cgr.k0;
      } else {   // replicated: full array; this rank computes the planes below its own fine planes, then all-gathers
        cd = false; hv = true;
        cgr.gnz = gnz_c; cgr.nz = gnz_c; cgr.k0 = 1; cgr.k1 = gnz_c - 1; cgr.gk = 0;
        vw = cgr; vw.k0 = cz ? (f.gk + f.k0 + 1) / 2 : f.gk + f.k0; vw.k1 = vw.k0 + nc;
      }
    }
    grids.push_back(cgr); isdist.push_back(cd); views.push_back(vw); hasview.push_back(hv);
  }
  if (grids.size() <= 2) { wl_set_error("MultiLevelPoisson requires size=a2ⁿ, where n>2"); return WL_ELEVELS; }   // :73-74
  // one slab allocation for everything the handle owns
  size_t total = 0;
  for (size_t l = 0; l < grids.size(); l++) { const size_t nc = (size_t)wl_ncell(grids[l]); total += (l == 0 ? 6 : 6 + 2 + (size_t)grids[l].D) * nc; }
  WL_HIP(hipMalloc((void**)&slab, total * sizeof(float)));
  WL_HIP(hipMemset(slab, 0, total * sizeof(float)));
  float* p = slab;
  lv.resize(grids.size());
  for (size_t l = 0; l < grids.size(); l++) {
    Level& v = lv[l]; v.g = grids[l]; v.x_ = gx(grids[l]); v.dist = isdist[l]; v.has_view = hasview[l]; v.view = gx(views[l]);
    const size_t nc = (size_t)wl_ncell(grids[l]);
    v.r = p; p += nc; v.eps = p; p += nc; v.D = p; p += nc; v.iD = p; p += nc; v.em = p; p += nc; v.rs = p; p += nc;
    if (l == 0) { v.x = x; v.L = L; v.z = z; }
    else { v.L = p; p += nc * (size_t)grids[l].D; v.x = p; p += nc; v.z = p; p += nc; }
  }
  hipStream_t s = 0;
  WL_TRY(update(s));                                                                       // restrictML :39 + Poisson ctor :36
  WL_HIP(hipStreamSynchronize(s));
  return 0;
}
wl_mg::~wl_mg() { if (slab) (void)hipFree(slab); if (red) (void)hipFree(red); if (side) (void)hipStreamDestroy(side); if (ev_decided) (void)hipEventDestroy(ev_decided); }

// coarse face coefficients of level l from level l-1 (restrictL! :42-48), slab aware
static int restrictL_level(wl_mg& m, size_t l, hipStream_t s) {
  wl_mg::Level& c = m.lv[l]; wl_mg::Level& f = m.lv[l - 1];
  if (c.has_view) {          // distributed parent -> replicated child: compute my planes, all-gather, then BC!(a,0) on the full array
    const float zero[3] = {0.f, 0.f, 0.f};
    WL_TRY(wl::restrictL(c.L, c.view, f.L, f.x_, m.perdir, s));     // (its BC pass is redone below on the complete array)
    WL_TRY(wl::allgather_planes(m.comm, c.L, c.view, c.g.D, s));
    return wl::bc_vec(c.L, c.x_, zero, 0, m.perdir, s);
  }
  WL_TRY(wl::restrictL(c.L, c.x_, f.L, f.x_, m.perdir, s));
  return m.halo(c, c.L, c.g.D, s);
}
int wl_mg::update(hipStream_t s) {                                                        // update! :79-86
  WL_TRY(halo(lv[0], lv[0].L, lv[0].g.D, s));
  WL_TRY(wl::set_diag(lv[0].D, lv[0].iD, lv[0].L, lv[0].x_, s));
  for (size_t l = 1; l < lv.size(); l++) {
    WL_TRY(restrictL_level(*this, l, s));
    WL_TRY(wl::set_diag(lv[l].D, lv[l].iD, lv[l].L, lv[l].x_, s));
  }
  // constant-coefficient detection (exact, on device): only the levels that run the specialised kernels are checked
  for (size_t l = 0; l < lv.size(); l++) {
    lv[l].cl.on = 0; lv[l].part = false;
    // (the levels of the single-launch tail too, when the finest level passed: the LDS-resident tail then evaluates L, D, iD instead of loading them)
    const bool tail_level = l > 0 && use_tail && lv[0].cl.on && lv[l].g.D == 3 && !lv[l].dist && lv[l].x_.cs <= WL_TAIL_CELLS;
    if (use_constl && !perdir && (l == 0 || tail_level || wl::gsrb_fused_ok(lv[l].x_, perdir, lv[l].dist) || (lv[l].dist && wl::gsrb_pair_geom_ok(lv[l].x_)))) {
      WL_TRY(wl::check_const_L(lv[l].L, lv[l].x_, &lv[l].cl, (int*)(ws.res_f + 7), s));
      if (lv[l].dist && comm && comm->size > 1) {   // every rank must take the same path (the slab kernels differ in their halo exchanges)
        const float bad = lv[l].cl.on ? 0.f : 1.f; float any = 1.f;
        WL_HIP(hipMemcpyAsync(ws.res_f + 7, &bad, sizeof(float), hipMemcpyHostToDevice, s));
        WL_TRY(wl::combine_results(comm, ws, s));                                          // res_f: max over ranks
        WL_HIP(hipMemcpyAsync(&any, ws.res_f + 7, sizeof(float), hipMemcpyDeviceToHost, s));
        WL_HIP(hipStreamSynchronize(s));
        if (any != 0.f) lv[l].cl.on = 0;
      }
      // a body: the pattern holds on most planes — find the planes where it does not (z-split smoother)
      Level& v = lv[l];
      if (!v.cl.on && !v.dist && v.g.D == 3 && wl::gsrb_fused_ok(v.x_, perdir, v.dist) && wl::gsrb_pair_geom_ok(v.x_) && v.cl.c[0] != 0.f) {
        WL_TRY(wl::const_plane_range(v.L, v.x_, v.cl.c, &v.za, &v.zb, s));
        const int m = 4, na = std::max(v.g.k0, v.za - m), nb = std::min(v.g.k1, v.zb + m + 1);
        const int far = (na - v.g.k0) + (v.g.k1 - nb);
        v.part = v.zb >= v.za && far >= 16 && far * 4 >= (v.g.k1 - v.g.k0) && v.x_.cs >= zsplit_min;     // worth it: at least a quarter of the planes are far
        v.clp = v.cl; v.clp.on = 1;
      }
    }
  }
  return 0;
}
// the deferred `prolongate!; increment!` of level l, executed on its own (when the next smooth! cannot absorb it)
int wl_mg::flush_pending(int l, float w, hipStream_t s) {
  Level& fine = lv[(size_t)l]; Level& coarse = lv[(size_t)l + 1];
  fine.pend = false;
  ProfScope pp(l == 0 ? WL_PROF_PROLONG : -1, s);
  if (perdir) {
    WL_TRY(wl::prolongate(fine.eps, fine.x_, coarse.x, coarse.x_, s));
    WL_TRY(wl::bc_per_scalar(fine.eps, fine.x_, perdir, s));
    WL_TRY(halo(fine, fine.eps, 1, s));
    return wl::increment(fine.r, fine.x, fine.eps, fine.L, fine.D, fine.x_, w, s);
  }
  return wl::prolong_increment(fine.r, fine.x, fine.eps, coarse.x, fine.L, fine.D, fine.x_, coarse.x_, w, true, s);
}
// GaussSeidelRB!(p;it,ω)                                                                 src/Poisson.jl:141-148
int wl_mg::smooth(int l, int it, float w, hipStream_t s, bool want_norms, bool* norms_done) {
  Level& p = lv[(size_t)l];
  if (norms_done) *norms_done = false;
  const bool fused = it == 4 && use_fused && (wl::gsrb_fused_ok(p.x_, perdir, p.dist) || pair_slab(p));
  if (p.pend && !fused) WL_TRY(flush_pending(l, w, s));
  ProfScope ps(l == 0 ? WL_PROF_SMOOTH : -1, s);   // only the finest level is a named slot
  if (fused && p.part && use_zsplit && !p.dist) {
    // Level with a body: the blocked kernels take plane sub-ranges (k0/k1 of the grid they are given only delimit the planes a launch
    // outputs; inputs are read across the cut, outputs are separate arrays).  Planes at least 4 away from the body run the
    // constant-coefficient pair kernels, the others the general kernels — the same bits either way.
    const int m = 4, na = std::max(p.g.k0, p.za - m), nb = std::min(p.g.k1, p.zb + m + 1);
    auto sub = [&](int a, int b) { GridX g = p.x_; g.k0 = a; g.k1 = b; return g; };
    struct Part { int a, b; const wl::ConstL* cl; } parts[3] = {{na, nb, &p.cl}, {p.g.k0, na, &p.clp}, {nb, p.g.k1, &p.clp}};
    const bool pro = p.pend;
    Level& coarse = lv[(size_t)(pro ? l + 1 : l)];
    p.pend = false;
    bool xdef[3] = {false, false, false};                    // per range: `x += ω·x_c↓` handed from kernel A to kernel B (wl::XDefer)
    const wl::XDefer xd{coarse.x, coarse.x_, w};
    // The three plane ranges write disjoint planes and read only what the previous phase left: their launches are independent, and each of them is a
    // latency-bound march on an under-filled chip (sphere 256³: 35–76 µs each).  They run concurrently on the main stream and two auxiliary streams
    // (fork: the aux streams wait for an event of the main stream; join: the main stream waits for theirs); kernel B needs ALL of kernel A's ranges
    // (its halo reaches across the cuts): one join between the two phases.  Partial norms of the ranges land in disjoint thirds of the workspace.
    const bool par = par_ranges && wl::par_streams_ok() && (long)p.g.nx * p.g.ny <= 520L * 520L;
    hipStream_t sr[3] = {s, par ? wl::par_stream(0) : s, par ? wl::par_stream(1) : s};
    {
      ProfScope pa(l == 0 ? WL_PROF_GS_A : -1, s);
      if (par) WL_TRY(wl::par_fork(s));
      for (int i = 0; i < 3; i++) if (parts[i].b > parts[i].a) {
        const GridX g = sub(parts[i].a, parts[i].b);
        if (pro) {
          xdef[i] = use_xdefer && wl::gsrb_pair_B_ok(store_eps ? p.eps : nullptr, p.r, p.x, p.em, p.rs, g, *parts[i].cl);
          if (l == 0) last_xdefer = xdef[i] ? 1 : 0;
          WL_TRY(wl::gsrb_fused_A_pro(p.em, p.rs, p.x, p.r, coarse.x, p.L, g, coarse.x_, w, *parts[i].cl, sr[i], -(1 << 30), 1 << 30, &xdef[i]));
        } else WL_TRY(wl::gsrb_fused_A(p.em, p.r, p.L, g, *parts[i].cl, sr[i]));
      }
      if (par) WL_TRY(wl::par_join(s));
    }
    {
      ProfScope pb(l == 0 ? WL_PROF_GS_B : -1, s);
      // L₁/L∞ of the new residual: every range leaves its own pair in a slot of its own (res_d[2|5|6], res_f[1|2|3]); solver! adds them
      static const int SD[3] = {2, 5, 6}, SF[3] = {1, 2, 3};
      norm_slots = 0;
      if (par) WL_TRY(wl::par_fork(s));
      for (int i = 0; i < 3; i++) if (parts[i].b > parts[i].a) {
        const GridX g = sub(parts[i].a, parts[i].b);
        RedWs wsi = ws;
        if (par) { wsi.pa += (size_t)i * (WL_MAXPART / 3); wsi.pm += (size_t)i * (WL_MAXPART / 3); }
        const RedWs* nws = want_norms ? &wsi : nullptr;
        if (want_norms) norm_slots |= 1 << i;
        if (pro) WL_TRY(wl::gsrb_fused_B(store_eps ? p.eps : nullptr, p.r, p.x, p.em, p.rs, p.L, g, w, nws, SD[i], SF[i], *parts[i].cl, sr[i], xdef[i] ? &xd : nullptr));
        else WL_TRY(wl::gsrb_fused_B(store_eps ? p.eps : nullptr, p.rs, p.x, p.em, p.r, p.L, g, w, nws, SD[i], SF[i], *parts[i].cl, sr[i]));
      }
      if (par) WL_TRY(wl::par_join(s));
    }
    if (!pro) std::swap(p.r, p.rs);
    if (norms_done) *norms_done = want_norms;
    return 0;
  }
  if (fused) {   // two z-marching kernels instead of six passes (+ the pending prolongation as an extra stage of kernel A)
    const RedWs* nws = want_norms ? &ws : nullptr;
    if (p.pend) {
      Level& coarse = lv[(size_t)l + 1];
      p.pend = false;
      bool xdef = use_xdefer && wl::gsrb_pair_B_ok(store_eps ? p.eps : nullptr, p.r, p.x, p.em, p.rs, p.x_, p.cl);   // `x += ω·x_c↓` handed from kernel A to kernel B
      const wl::XDefer xd{coarse.x, coarse.x_, w};
      if (l == 0) last_xdefer = xdef ? 1 : 0;
      // z-slab: the tile pipeline recomputes the neighbour's planes it needs, so the exchanges are r (2 planes) before A and
      // ϵ_mid (3 planes) + r' (2 planes) before B — instead of one exchange per colour sweep
      if (p.dist && deep_halo && p.g.k0 >= 5 && p.g.k1 - p.g.k0 >= 5) {
        // z-slab, ONE exchange round per smooth!: r travels 5 planes deep and kernel A also computes r' and ϵ_mid on the 3 (2) ghost planes
        // kernel B reads, instead of receiving them (x is updated on the owned planes only).  5 planes instead of 2+3+2, one latency
        // instead of two, ≈6 redundant planes of kernel A per rank.
        GridX ge = p.x_; ge.k0 = p.x_.k0 - 3; ge.k1 = p.x_.k1 + 3;
        if (overlap_smooth && p.x_.k1 - p.x_.k0 >= 16) {
          // the exchange runs on the communicator's own stream while kernel A computes the planes that need no ghost plane of r (its outputs
          // [k0+2,k1−2) read r on [k0,k1) only); the two boundary slices (5 planes each, ghost planes included) follow the wait — as conv_diff! does with u
          WL_TRY(wl::halo_async_begin(comm, p.r, p.x_, 1, 5, s));
          GridX gi = p.x_; gi.k0 = p.x_.k0 + 2; gi.k1 = p.x_.k1 - 2;
          GridX glo = p.x_; glo.k0 = ge.k0; glo.k1 = p.x_.k0 + 2;
          GridX ghi = p.x_; ghi.k0 = p.x_.k1 - 2; ghi.k1 = ge.k1;
          ProfScope pa(l == 0 ? WL_PROF_GS_A : -1, s);
          bool xd_i = xdef, xd_l = xdef, xd_h = xdef;
          WL_TRY(wl::gsrb_fused_A_pro(p.em, p.rs, p.x, p.r, coarse.x, p.L, gi, coarse.x_, w, p.cl, s, p.x_.k0, p.x_.k1, &xd_i, true));
          WL_TRY(wl::halo_async_wait(comm, s));
          WL_TRY(wl::gsrb_fused_A_pro(p.em, p.rs, p.x, p.r, coarse.x, p.L, glo, coarse.x_, w, p.cl, s, p.x_.k0, p.x_.k1, &xd_l, true));
          WL_TRY(wl::gsrb_fused_A_pro(p.em, p.rs, p.x, p.r, coarse.x, p.L, ghi, coarse.x_, w, p.cl, s, p.x_.k0, p.x_.k1, &xd_h, true));
          if (xd_i != xdef || xd_l != xdef || xd_h != xdef) { wl_set_error("smooth!: the slices of kernel A disagree on the deferred x increment"); return WL_EINVAL; }
        } else {
        WL_TRY(halo(p, p.r, 1, s, 5));
        { ProfScope pa(l == 0 ? WL_PROF_GS_A : -1, s); WL_TRY(wl::gsrb_fused_A_pro(p.em, p.rs, p.x, p.r, coarse.x, p.L, ge, coarse.x_, w, p.cl, s, p.x_.k0, p.x_.k1, &xdef)); }
        }
        { ProfScope pb(l == 0 ? WL_PROF_GS_B : -1, s); WL_TRY(wl::gsrb_fused_B(store_eps ? p.eps : nullptr, p.r, p.x, p.em, p.rs, p.L, p.x_, w, nws, 2, 1, p.cl, s, xdef ? &xd : nullptr)); }
        norm_slots = 0;
        if (norms_done) *norms_done = want_norms;
        return 0;
      }
      WL_TRY(halo(p, p.r, 1, s, 2));
      { ProfScope pa(l == 0 ? WL_PROF_GS_A : -1, s); WL_TRY(wl::gsrb_fused_A_pro(p.em, p.rs, p.x, p.r, coarse.x, p.L, p.x_, coarse.x_, w, p.cl, s, -(1 << 30), 1 << 30, &xdef)); }
      {   // one RCCL group for both arrays: one exchange latency instead of two
        const bool grp = comm && comm->size > 1 && p.dist;
        if (grp) { WL_TRY(comm->group_begin()); comm->n_halo++; }   // one network round for both arrays
        int rc = halo(p, p.em, 1, s, 3);
        if (rc == 0) rc = halo(p, p.rs, 1, s, 2);
        if (grp) { const int rc2 = comm->group_end(); if (rc == 0) rc = rc2; }
        WL_TRY(rc);
      }
      { ProfScope pb(l == 0 ? WL_PROF_GS_B : -1, s); WL_TRY(wl::gsrb_fused_B(store_eps ? p.eps : nullptr, p.r, p.x, p.em, p.rs, p.L, p.x_, w, nws, 2, 1, p.cl, s, xdef ? &xd : nullptr)); }
    } else {
      WL_TRY(halo(p, p.r, 1, s, 2));
      { ProfScope pa(l == 0 ? WL_PROF_GS_A : -1, s); WL_TRY(wl::gsrb_fused_A(p.em, p.r, p.L, p.x_, p.cl, s)); }
      WL_TRY(halo(p, p.em, 1, s, 3));
      { ProfScope pb(l == 0 ? WL_PROF_GS_B : -1, s); WL_TRY(wl::gsrb_fused_B(store_eps ? p.eps : nullptr, p.rs, p.x, p.em, p.r, p.L, p.x_, w, nws, 2, 1, p.cl, s)); }
      std::swap(p.r, p.rs);
    }
    norm_slots = 0;
    if (norms_done) *norms_done = want_norms;
    return 0;
  }
  const bool fuse = !perdir && !p.dist && it >= 1;   // ghost ϵ are plain memory reads only on these levels
  if (fuse) WL_TRY(wl::gs_init_sweep1(p.eps, p.r, p.L, p.iD, p.x_, s));
  else {
    WL_TRY(wl::gs_init(p.eps, p.r, p.iD, p.x_, s));
    WL_TRY(wl::bc_per_scalar(p.eps, p.x_, perdir, s));
    WL_TRY(halo(p, p.eps, 1, s));
  }
  for (int k0 = fuse ? 2 : 1; k0 <= it; k0++) {
    ProfScope pk(l == 0 ? WL_PROF_GS_SWEEP : -1, s);
    WL_TRY(wl::gs_sweep(p.eps, p.r, p.L, p.iD, p.x_, k0, s));
    WL_TRY(halo(p, p.eps, 1, s, 1, false));                                                // neighbour slabs need this colour before the next sweep (no periodic wrap: the reference's ghost cells are stale here)
  }
  WL_TRY(wl::bc_per_scalar(p.eps, p.x_, perdir, s));                                      // perBC!(ϵ) inside increment! :101
  if (comm && comm->zperiodic) WL_TRY(halo(p, p.eps, 1, s));                              // … across the periodic z boundary too
  return wl::increment(p.r, p.x, p.eps, p.L, p.D, p.x_, w, s);
}
// the levels first..end as one launch: "if (first is not the coarsest) Vcycle!(first); smooth!(first)"
bool wl_mg::tail_ok(int first) const {
  if (!use_tail || perdir || first < 1 || first >= (int)lv.size() || (int)lv.size() - first > WL_TAIL_MAXLV) return false;
  if (lv[(size_t)first].g.D != 3 || lv[(size_t)first].x_.cs > WL_TAIL_CELLS) return false;
  for (size_t l = (size_t)first; l < lv.size(); l++) if (lv[l].dist || lv[l].pend) return false;
  return true;
}
int wl_mg::tail(int first, float w, hipStream_t s) {
  wl::TailLevelHost h[WL_TAIL_MAXLV];
  const int n = (int)lv.size() - first;
  for (int q = 0; q < n; q++) {
    const Level& v = lv[(size_t)(first + q)];
    h[q] = wl::TailLevelHost{v.x_, v.L, v.D, v.iD, v.x, v.eps, v.r, 0, 0, 0, &v.cl};
    if (q + 1 < n) { const Level& c = lv[(size_t)(first + q + 1)]; h[q].cx = c.g.nx < v.g.nx; h[q].cy = c.g.ny < v.g.ny; h[q].cz = c.g.gnz < v.g.gnz; }
  }
  return wl::vcycle_tail(h, n, w, s);
}
int wl_mg::vcycle(int l, float w, hipStream_t s, bool defer) {                            // Vcycle! :88-101
  Level& fine = lv[(size_t)l]; Level& coarse = lv[(size_t)l + 1];
  // Jacobi!(fine): ϵ=r·iD; increment!(ω=1)   (perBC!(ϵ) inside increment!)
  if (l == 0 && jacobi0_done) jacobi0_done = false;   // Jacobi!(fine) was fused into the projection head (wl_sim::project → wl::resjac)
  else {
    ProfScope pj(l == 0 ? WL_PROF_JACOBI : -1, s);
    if (!perdir && (!fine.dist || fine.cl.on)) {   // one pass; new residual lands in the ϵ buffer, then the two buffers trade places
      WL_TRY(halo(fine, fine.r, 1, s));              // (slab: ϵ=r·iD of the neighbour's boundary plane is recomputed from its r; iD is evaluated from the position)
      if (l == 0 && shift_pending) { shift_pending = false; WL_TRY(wl::jacobi_pp_shift(fine.eps, fine.r, fine.x, fine.x_, 1.f, fine.cl, ws, 1, 0, s)); }
      else if (fine.part && use_zsplit && !fine.dist) {
        // level with a body: constant-coefficient (z-marching) Jacobi on the plane ranges away from it, the general kernel around it —
        // same ranges as the z-split smoother (the output is a separate array, r is read across the cuts)
        const int xz = fine.xzero ? 1 : 0; fine.xzero = false;
        const int m = 4, na = std::max(fine.g.k0, fine.za - m), nb = std::min(fine.g.k1, fine.zb + m + 1);
        auto sub = [&](int a, int b) { GridX g = fine.x_; g.k0 = a; g.k1 = b; return g; };
        if (nb > na) WL_TRY(wl::jacobi_pp(fine.eps, fine.r, fine.x, fine.L, fine.D, fine.iD, sub(na, nb), 1.f, fine.cl, s, xz));
        if (na > fine.g.k0) WL_TRY(wl::jacobi_pp(fine.eps, fine.r, fine.x, fine.L, fine.D, fine.iD, sub(fine.g.k0, na), 1.f, fine.clp, s, xz));
        if (fine.g.k1 > nb) WL_TRY(wl::jacobi_pp(fine.eps, fine.r, fine.x, fine.L, fine.D, fine.iD, sub(nb, fine.g.k1), 1.f, fine.clp, s, xz));
      }
      else { const int xz = fine.xzero ? 1 : 0; fine.xzero = false; WL_TRY(wl::jacobi_pp(fine.eps, fine.r, fine.x, fine.L, fine.D, fine.iD, fine.x_, 1.f, fine.cl, s, xz)); }
      std::swap(fine.r, fine.eps);
    } else {
      WL_TRY(wl::gs_init(fine.eps, fine.r, fine.iD, fine.x_, s));
      WL_TRY(wl::bc_per_scalar(fine.eps, fine.x_, perdir, s));
      WL_TRY(halo(fine, fine.eps, 1, s));
      WL_TRY(wl::increment(fine.r, fine.x, fine.eps, fine.L, fine.D, fine.x_, 1.f, s));
    }
  }
  {
    ProfScope pc(l == 0 ? WL_PROF_COARSE : -1, s);   // everything below the finest level
    if (coarse.has_view) {
      WL_TRY(wl::restrict_(coarse.r, coarse.view, fine.r, fine.x_, s));
      WL_TRY(wl::allgather_planes(comm, coarse.r, coarse.view, 1, s));
    } else WL_TRY(wl::restrict_(coarse.r, coarse.x_, fine.r, fine.x_, s));
    // fill!(coarse.x,0) :92 — folded into the coarse level's Jacobi! when that is what touches x next (single-domain level, one-pass
    // Jacobi kernels): its ghost cells are zero since allocation and nothing writes them
    const bool to_tail = tail_ok(l + 1);
    coarse.xzero = skip_fill && !to_tail && l + 2 < (int)lv.size() && !perdir && !coarse.dist && !coarse.has_view;
    if (!coarse.xzero) WL_TRY(wl::fill(coarse.x, 0.f, (size_t)coarse.x_.cs, s));
    if (to_tail) WL_TRY(tail(l + 1, w, s));                                         // everything below in one launch
    else {
      if (l + 2 < (int)lv.size()) WL_TRY(vcycle(l + 1, w, s, true));                       // its last step may be deferred into the smooth! below
      WL_TRY(smooth(l + 1, 4, w, s));
    }
    WL_TRY(halo(coarse, coarse.x, 1, s, pair_slab(fine) ? ((deep_halo && fine.g.k0 >= 5 && fine.g.k1 - fine.g.k0 >= 5 && coarse.g.k0 >= 3 && coarse.g.k1 - coarse.g.k0 >= 3) ? 3 : 2) : 1));   // prolongation reads the coarse cells under my halo planes (deep halo: kernel A starts 5 planes out)
  }
  // prolongate!(fine.ϵ,coarse.x); increment!(fine;ω): the caller's next operation is smooth!(fine;ω) with the same ω — when that
  // smooth! runs as the temporally blocked kernel pair it absorbs this step as an extra pipeline stage (defer).
  if (defer && use_fused && (wl::gsrb_fused_ok(fine.x_, perdir, fine.dist) || pair_slab(fine))) { fine.pend = true; return 0; }
  fine.pend = true;
  return flush_pending(l, w, s);
}
int wl_mg::solve(double tol, int itmx, int* host_n, double* host_r1, float* host_rinf, hipStream_t s, bool have_residual, const double* pre_r1, const float* pre_rinf) {   // solver! :108-128
  Level& p = lv[0];
  const double r1tol = (tol / 10.0) * (double)wl_ninside_global(p.g);                     // l1n_tol  src/Poisson.jl:194
  const double rinftol = tol;
  {
    ProfScope pr(WL_PROF_RESIDUAL, s);
    if (!have_residual) {
      WL_TRY(wl::bc_per_scalar(p.x, p.x_, perdir, s));                                    // residual!: perBC!(x) :93
      WL_TRY(halo(p, p.x, 1, s));
      WL_TRY(wl::residual_part(p.r, p.x, p.z, p.L, p.D, p.iD, p.x_, ws, s));             // r and the local Σr -> res_d[0]
    }
    if (!(jacobi0_done && pre_r1)) WL_TRY(wl::combine_results(comm, ws, s));   // (fused head: Σr, r₁, r∞ were combined and read by the caller)
    // mean shift + r₁ -> res_d[1], r∞ -> res_f[0] — unless the V-cycle's first operation is the z-marching Jacobi! on this level
    // (always run: nᵖ ≥ 1): that kernel applies the shift as it loads r and accumulates the norms, no pass over r at all
    shift_pending = !jacobi0_done && defer_shift && itmx >= 1 && !(comm && comm->size > 1) && !perdir && lv.size() > 1 && wl::jacobi_takes_shift(p.x_, p.cl);
    if (!shift_pending && !jacobi0_done) WL_TRY(wl::shift_norms_dev(p.r, p.x_, ws, 1, 0, s));
  }
  double hd[8]; float hf[8];
  float w = 1.f;
  // r₁ of the initial residual is only needed for the ω rule after the first V-cycle: fetched with the first iteration's norms
  bool have_r1 = false; float r1 = 0.f, rinf = 0.f;
  int np = 0;
  log_r1.clear(); log_rinf.clear(); log_w.clear();
  if (jacobi0_done && pre_r1 && pre_rinf) { r1 = (float)*pre_r1; log_r1.push_back(*pre_r1); log_rinf.push_back((double)*pre_rinf); log_w.push_back(1.0); have_r1 = true; }
  std::function<int(const float*)> tail; tail.swap(spec_tail);     // one-shot
  const bool check_head = spec_check_head; spec_check_head = false;
  tail_stood = false;
  while (np < itmx) {
    WL_TRY(vcycle(0, w, s, true));
    bool nd = false;
    norm_slots = 0;
    WL_TRY(smooth(0, 4, w, s, true, &nd));                                                // fused path: norms come out of kernel B
    if (!nd) { norm_slots = 0; WL_TRY(wl::norms_dev(lv[0].r, p.x_, ws, 2, 1, s)); }       // rnew -> res_d[2], r∞ -> res_f[1]
    WL_TRY(wl::combine_results(comm, ws, s));                                             // (slot 0 becomes P·Σr: not used again)
    const bool spec = (bool)tail && norm_slots == 0 && !comm;
    if (spec) {   // the break test on the device, and the projection tail behind it: runs iff this iteration is the last one
      WL_TRY(wl::decide_converged(ws, r1tol, rinftol, (double)wl_ninside_global(p.g), (check_head && np == 0) ? 1 : 0, 2, 1, 4, s));
      if (!ev_decided) WL_HIP(hipEventCreateWithFlags(&ev_decided, hipEventDisableTiming));
      const float* go = ws.res_f + 4;
      WL_TRY(wl::read_results_overlapped(ws, hd, 7, hf, 5, s, ev_decided, [&]() -> int { return tail(go); }));   // the copy of the norms sits between the decision and the tail: the host wakes for the copy and goes on queueing work behind the running tail
    } else
    WL_TRY(wl::read_results(ws, hd, 7, hf, 5, s));
    if (norm_slots) {   // z-split smoother: one (L₁, L∞) pair per plane range
      static const int SD[3] = {2, 5, 6}, SF[3] = {1, 2, 3};
      double a = 0.0; float m = 0.f;
      for (int i = 0; i < 3; i++) if (norm_slots & (1 << i)) { a += hd[SD[i]]; m = std::fmax(m, hf[SF[i]]); }
      hd[2] = a; hf[1] = m;
    }
    if (np == 0) first_hd0 = hd[0];
    if (!have_r1) { r1 = (float)hd[1]; log_r1.push_back(hd[1]); log_rinf.push_back(hf[0]); log_w.push_back(1.0); have_r1 = true; }
    const float rnew = (float)hd[2]; rinf = hf[1]; np++;
    log_r1.push_back((double)rnew); log_rinf.push_back((double)rinf); log_w.push_back((double)w);
    if (rnew >= r1) w = (float)std::fmax(0.2, 0.9 * (double)w);                           // :118-119
    else if (rnew < r1) w = (float)std::fmin(1.0, 1.02 * (double)w);                      // :120-121
    r1 = rnew;
    if (spec) {
      // the device's flag IS the decision (same statements as below on the same two numbers; with check_head also the head's mean-shift test — if that one failed
      // the caller discards this solve: stop here, the tail has not run)
      if (check_head && np == 1 && !(std::fabs((float)hd[0] / (float)(double)wl_ninside_global(p.g)) <= 2.f * 1.1920929e-7f)) break;
      if (hf[4] != 0.f) { tail_stood = true; break; }
      continue;
    }
    if ((double)r1 < r1tol && (double)rinf < rinftol) break;
  }
  WL_TRY(wl::bc_per_scalar(p.x, p.x_, perdir, s));                                        // :126
  WL_TRY(halo(p, p.x, 1, s, x_halo_depth));                                               // projection reads x[I-δz] across the slab face (the next solve's fused head two planes deep)
  n.push_back((int16_t)np);
  if (host_n) *host_n = np;
  if (host_r1) *host_r1 = (double)r1;
  if (host_rinf) *host_rinf = rinf;
  return 0;
}

// ================================================================================================
extern "C" {

int wl_version(void) { return 100; }
int wl_prof_enable(int on) {
  WlProf& p = wl_prof();
  WL_HIP(hipDeviceSynchronize());
  for (int q = 0; q < WL_PROF_NSLOTS; q++) p.slot[q].used = 0;
  p.on = on != 0; p.only_roofline = on == 2;
  return 0;
}
int wl_prof_read(int slot, int* count, double* total_ms) {
  WL_CHECK(slot >= 0 && slot < WL_PROF_NSLOTS, "bad profiling slot");
  WL_HIP(hipDeviceSynchronize());
  WlProf::Slot& sl = wl_prof().slot[slot];
  double tot = 0.0;
  for (size_t q = 0; q < sl.used; q++) { float ms = 0.f; WL_HIP(hipEventElapsedTime(&ms, sl.a[q], sl.b[q])); tot += (double)ms; }
  if (count) *count = (int)sl.used;
  if (total_ms) *total_ms = tot;
  return 0;
}
const char* wl_last_error_string(void) { return g_err.c_str(); }
int wl_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { wl_set_error("libwlhip: no HIP device visible"); return WL_ENOGPU; }
  WL_CHECK(device >= 0 && device < n, "device index out of range");
  WL_HIP(hipSetDevice(device));
  hipDeviceProp_t prop; WL_HIP(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) { wl_set_error(std::string("libwlhip is built for gfx950 only, device is ") + prop.gcnArchName); return WL_ENOGPU; }
  return wl_ctx_ensure();
}
int wl_malloc(void** p, size_t bytes) { WL_CHECK(p != nullptr, "null out pointer"); WL_HIP(hipMalloc(p, bytes)); return 0; }
int wl_free(void* p) { WL_HIP(hipFree(p)); return 0; }
int wl_h2d(void* dst, const void* src, size_t bytes, void* stream) { WL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, wl_stream(stream))); return 0; }
int wl_d2h(void* dst, const void* src, size_t bytes, void* stream) { WL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, wl_stream(stream))); WL_HIP(hipStreamSynchronize(wl_stream(stream))); return 0; }
int wl_d2d(void* dst, const void* src, size_t bytes, void* stream) { WL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, wl_stream(stream))); return 0; }
int wl_stream_sync(void* stream) { WL_HIP(hipStreamSynchronize(wl_stream(stream))); return 0; }
wl_grid wl_grid_single(int D, const int32_t* d) {
  wl_grid g; g.D = D; g.nx = d[0]; g.ny = d[1];
  if (D == 3) { g.nz = d[2]; g.k0 = 1; g.k1 = d[2] - 1; g.gk = 0; g.gnz = d[2]; }
  else { g.nz = 1; g.k0 = 0; g.k1 = 1; g.gk = 0; g.gnz = 1; }
  return g;
}
size_t wl_reduce_workspace_bytes(void) { return wl_red_bytes(); }

#define GRID_ARG(g) WL_CHECK(wl_grid_ok(g), "bad wl_grid"); const GridX G = gx(*g)
#define DEFAULT_WS() WL_TRY(wl_ctx_ensure()); const RedWs ws = wl_red_ws(wl_ctx().red)

int wl_fill(float* a, float v, size_t n, void* st) { return wl::fill(a, v, n, wl_stream(st)); }
int wl_scale(float* a, float s, size_t n, void* st) { return wl::scale(a, s, n, wl_stream(st)); }
int wl_div_scalar(float* a, float s, size_t n, void* st) { return wl::div_scalar(a, s, n, wl_stream(st)); }
int wl_sum(const float* a, size_t n, double* out, void* st) { DEFAULT_WS(); WL_TRY(wl::sum_dev(a, n, ws, 0, wl_stream(st))); return wl::read_results(ws, out, 1, nullptr, 0, wl_stream(st)); }
int wl_sum_abs_max_abs(const float* a, size_t n, double* l1, float* linf, void* st) { DEFAULT_WS(); WL_TRY(wl::l1_linf_dev(a, n, ws, 0, 0, wl_stream(st))); return wl::read_results(ws, l1, 1, linf, 1, wl_stream(st)); }
int wl_max(const float* a, size_t n, float* out, void* st) { DEFAULT_WS(); WL_TRY(wl::max_dev(a, n, ws, 0, wl_stream(st))); return wl::read_results(ws, nullptr, 0, out, 1, wl_stream(st)); }
int wl_dot(const float* a, const float* b, size_t n, double* out, void* st) { DEFAULT_WS(); WL_TRY(wl::dot_dev(a, b, n, ws, 0, wl_stream(st))); return wl::read_results(ws, out, 1, nullptr, 0, wl_stream(st)); }

int wl_bc_vec(float* a, const wl_grid* g, const float* U, int saveexit, unsigned per, void* st) { GRID_ARG(g); return wl::bc_vec(a, G, U, saveexit, per, wl_stream(st)); }
int wl_bc_vec_fn(float* a, const float* Ub, const wl_grid* g, int saveexit, unsigned per, void* st) { GRID_ARG(g); WL_CHECK(a && Ub && a != Ub, "bad argument"); return wl::bc_vec_fn(a, Ub, G, saveexit, per, wl_stream(st)); }
int wl_meanflow_update(float* P, float* U, float* UU, const float* p, const float* u, const wl_grid* g, float eps, void* st) { GRID_ARG(g); WL_CHECK(P && U && p && u, "bad argument"); return wl::meanflow_update(P, U, UU, p, u, G, eps, wl_stream(st)); }
int wl_meanflow_uu(float* tau, const float* UU, const float* U, const wl_grid* g, void* st) { GRID_ARG(g); WL_CHECK(tau && UU && U, "bad argument"); return wl::meanflow_uu(tau, UU, U, G, wl_stream(st)); }
int wl_accelerate_field(float* r, const float* gfield, const wl_grid* g, void* st) { GRID_ARG(g); WL_CHECK(r && gfield, "bad argument"); return wl::add_field(r, gfield, (size_t)G.cs * (size_t)G.D, wl_stream(st)); }
int wl_bc_per_scalar(float* a, const wl_grid* g, unsigned per, void* st) { GRID_ARG(g); return wl::bc_per_scalar(a, G, per, wl_stream(st)); }
int wl_conv_diff(float* r, const float* u, float* Phi, const wl_grid* g, float nu, unsigned per, int scheme, void* st) { GRID_ARG(g); return wl::conv_diff(r, u, Phi, G, nu, per, scheme, wl_stream(st)); }
int wl_bdim(float* u, const float* u0, float* f, const float* V, const float* mu0, const float* mu1, const wl_grid* g, float dt, float pre, float post, void* st) {
  GRID_ARG(g); return wl::bdim(u, u0, f, V, mu0, mu1, G, dt, pre, post, wl_stream(st));
}
int wl_scale_u(float* u, const wl_grid* g, float s, void* st) { GRID_ARG(g); return wl::scale_u(u, G, s, wl_stream(st)); }
int wl_div(float* z, const float* u, const wl_grid* g, void* st) { GRID_ARG(g); return wl::div(z, u, G, wl_stream(st)); }
int wl_project(float* u, const float* L, const float* x, const wl_grid* g, void* st) { GRID_ARG(g); return wl::project(u, L, x, G, wl_stream(st)); }
int wl_cfl(const float* u, float* sigma, const wl_grid* g, float nu, float dt_max, float* host_dt, void* st) {
  GRID_ARG(g); DEFAULT_WS();
  WL_TRY(wl::cfl_dev(u, sigma, G, ws, 0, wl_stream(st)));
  float mx; WL_TRY(wl::read_results(ws, nullptr, 0, &mx, 1, wl_stream(st)));
  *host_dt = std::fmin(dt_max, 1.0f / (mx + 5 * nu));                                    // src/Flow.jl:236
  return 0;
}
int wl_set_diag(float* D, float* iD, const float* L, const wl_grid* g, void* st) { GRID_ARG(g); return wl::set_diag(D, iD, L, G, wl_stream(st)); }
int wl_mult(float* z, const float* L, const float* D, const float* x, const wl_grid* g, void* st) { GRID_ARG(g); return wl::mult(z, L, D, x, G, wl_stream(st)); }
int wl_residual(float* r, const float* x, const float* z, const float* L, const float* D, const float* iD, const wl_grid* g, void* scratch, void* st) {
  GRID_ARG(g); WL_TRY(wl_ctx_ensure());
  const RedWs ws = wl_red_ws(scratch ? scratch : wl_ctx().red);
  return wl::residual(r, x, z, L, D, iD, G, ws, wl_stream(st));
}
int wl_increment(float* r, float* x, const float* eps, const float* L, const float* D, const wl_grid* g, float w, void* st) { GRID_ARG(g); return wl::increment(r, x, eps, L, D, G, w, wl_stream(st)); }
int wl_jacobi(float* eps, float* r, float* x, const float* L, const float* D, const float* iD, const wl_grid* g, int it, float w, unsigned per, void* st) {
  GRID_ARG(g); hipStream_t s = wl_stream(st);
  for (int k = 0; k < (it <= 0 ? 1 : it); k++) {
    WL_TRY(wl::gs_init(eps, r, iD, G, s));
    WL_TRY(wl::bc_per_scalar(eps, G, per, s));       // perBC!(ϵ) inside increment!  src/Poisson.jl:101
    WL_TRY(wl::increment(r, x, eps, L, D, G, w, s));
  }
  return 0;
}
int wl_gsrb(float* eps, float* r, float* x, const float* L, const float* D, const float* iD, const wl_grid* g, int it, float w, unsigned per, void* st) {
  GRID_ARG(g); hipStream_t s = wl_stream(st);
  WL_TRY(wl::gs_init(eps, r, iD, G, s));
  WL_TRY(wl::bc_per_scalar(eps, G, per, s));
  for (int k0 = 1; k0 <= it; k0++) WL_TRY(wl::gs_sweep(eps, r, L, iD, G, k0, s));
  WL_TRY(wl::bc_per_scalar(eps, G, per, s));
  return wl::increment(r, x, eps, L, D, G, w, s);
}
// pcg!(p;it)   src/Poisson.jl:166-186.  The scalars ρ, α, β steer early exits exactly as in the reference, so each is read
// back (two host reads per iteration, like the reference's blocking `⋅`).  Single domain only.
static int pcg_impl(float* eps, float* r, float* x, float* z, const float* L, const float* D, const float* iD, const GridX& G, int it, unsigned per, const RedWs& ws, hipStream_t s) {
  const float tiny = 10 * 1.1920929e-07f;                                                  // 10eps(T)
  double h;
  WL_TRY(wl::pcg_stage(0, eps, r, x, z, L, D, iD, G, 0.f, 0, ws, s));                       // z = ϵ = r·iD ; rho = r⋅z
  WL_TRY(wl::read_results(ws, &h, 1, nullptr, 0, s));
  float rho = (float)h;
  if (std::fabs(rho) < tiny) return 0;
  for (int i = 1; i <= it; i++) {
    WL_TRY(wl::bc_per_scalar(eps, G, per, s));                                             // perBC!(ϵ)
    WL_TRY(wl::pcg_stage(1, eps, r, x, z, L, D, iD, G, 0.f, 0, ws, s));                     // z = Aϵ ; perdot(z,ϵ)
    WL_TRY(wl::read_results(ws, &h, 1, nullptr, 0, s));
    const float alpha = rho / (float)h;
    if (std::fabs(alpha) < 1e-2f || std::fabs(alpha) > 1e2f) return 0;                      // alpha should be O(1)
    const int more = i < it;
    WL_TRY(wl::pcg_stage(2, eps, r, x, z, L, D, iD, G, alpha, more, ws, s));                // x += αϵ ; r -= αz [; z = r·iD ; rho2 = r⋅z]
    if (!more) return 0;
    WL_TRY(wl::read_results(ws, &h, 1, nullptr, 0, s));
    const float rho2 = (float)h;
    if (std::fabs(rho2) < tiny) return 0;
    const float beta = rho2 / rho;
    WL_TRY(wl::pcg_stage(3, eps, r, x, z, L, D, iD, G, beta, 0, ws, s));                    // ϵ = βϵ + z
    rho = rho2;
  }
  return 0;
}
int wl_pcg(float* eps, float* r, float* x, float* z, const float* L, const float* D, const float* iD, const wl_grid* g, int it, unsigned per, void* st) {
  GRID_ARG(g); WL_CHECK(g->D == 2 || g->nz == g->gnz, "pcg! is single-domain"); DEFAULT_WS();
  return pcg_impl(eps, r, x, z, L, D, iD, G, it <= 0 ? 6 : it, per, ws, wl_stream(st));
}
// solver!(p::Poisson;tol,itmx)   src/Poisson.jl:212-223
int wl_poisson_solve(float* eps, float* r, float* x, float* z, const float* L, const float* D, const float* iD, const wl_grid* g, double tol, int itmx, unsigned per,
                     int* host_n, double* host_r1, float* host_rinf, void* st) {
  GRID_ARG(g); WL_CHECK(g->D == 2 || g->nz == g->gnz, "solver!(::Poisson) is single-domain"); DEFAULT_WS();
  hipStream_t s = wl_stream(st);
  const double r1tol = (tol / 10.0) * (double)wl_ninside_global(*g);
  WL_TRY(wl::bc_per_scalar(x, G, per, s));                                                 // residual!: perBC!(x) :93
  WL_TRY(wl::residual(r, x, z, L, D, iD, G, ws, s));
  double r1 = 0.0; float rinf = 0.f;
  int np = 0;
  const int cap = itmx <= 0 ? 1000 : itmx;
  while (np < cap) {
    WL_TRY(pcg_impl(eps, r, x, z, L, D, iD, G, 6, per, ws, s));
    WL_TRY(wl::norms_dev(r, G, ws, 1, 0, s));
    double hd[2]; float hf[1];
    WL_TRY(wl::read_results(ws, hd, 2, hf, 1, s));
    r1 = (double)(float)hd[1]; rinf = hf[0]; np++;
    if (r1 < r1tol && (double)rinf < tol) break;
  }
  WL_TRY(wl::bc_per_scalar(x, G, per, s));                                                 // :221
  if (host_n) *host_n = np;
  if (host_r1) *host_r1 = r1;
  if (host_rinf) *host_rinf = rinf;
  return 0;
}
int wl_norms(const float* r, const wl_grid* g, double* l1, float* linf, void* scratch, void* st) {
  GRID_ARG(g); WL_TRY(wl_ctx_ensure());
  const RedWs ws = wl_red_ws(scratch ? scratch : wl_ctx().red);
  WL_TRY(wl::norms_dev(r, G, ws, 0, 0, wl_stream(st)));
  return wl::read_results(ws, l1, 1, linf, 1, wl_stream(st));
}
int wl_restrict(float* a, const wl_grid* gc, const float* b, const wl_grid* gf, void* st) { WL_CHECK(wl_grid_ok(gc) && wl_grid_ok(gf), "bad wl_grid"); return wl::restrict_(a, gx(*gc), b, gx(*gf), wl_stream(st)); }
int wl_prolongate(float* a, const wl_grid* gf, const float* b, const wl_grid* gc, void* st) { WL_CHECK(wl_grid_ok(gc) && wl_grid_ok(gf), "bad wl_grid"); return wl::prolongate(a, gx(*gf), b, gx(*gc), wl_stream(st)); }
int wl_restrictL(float* a, const wl_grid* gc, const float* b, const wl_grid* gf, unsigned per, void* st) { WL_CHECK(wl_grid_ok(gc) && wl_grid_ok(gf), "bad wl_grid"); return wl::restrictL(a, gx(*gc), b, gx(*gf), per, wl_stream(st)); }
int wl_coarsen_dims(int D, const int32_t* fine, int32_t* coarse) {
  int c = 0;
  for (int d = 0; d < D; d++) { if (divisible(fine[d])) { coarse[d] = 1 + fine[d] / 2; c++; } else coarse[d] = fine[d]; }
  return c;
}

int wl_mg_create(wl_mg** out, float* x, float* L, float* z, const wl_grid* g, unsigned per, int maxlevels) {
  WL_CHECK(out && x && L && z, "null pointer"); WL_CHECK(wl_grid_ok(g), "bad wl_grid");
  wl_mg* mg = new wl_mg();
  int rc = mg->build(x, L, z, *g, per, maxlevels <= 0 ? 10 : maxlevels, nullptr);
  if (rc != 0) { delete mg; *out = nullptr; return rc; }
  *out = mg; return 0;
}
int wl_mg_destroy(wl_mg* mg) { delete mg; return 0; }
int wl_mg_update(wl_mg* mg, void* st) { return mg->update(wl_stream(st)); }
int wl_mg_nlevels(const wl_mg* mg) { return (int)mg->lv.size(); }
int wl_mg_level_grid(const wl_mg* mg, int l, wl_grid* out) { WL_CHECK(l >= 0 && l < (int)mg->lv.size(), "level out of range"); *out = mg->lv[(size_t)l].g; return 0; }
float* wl_mg_level_field(const wl_mg* mg, int l, const char* name) {
  if (l < 0 || l >= (int)mg->lv.size()) return nullptr;
  const wl_mg::Level& v = mg->lv[(size_t)l]; const std::string s(name);
  if (s == "L") return v.L; if (s == "D") return v.D; if (s == "iD") return v.iD; if (s == "x") return v.x;
  if (s == "eps") return v.eps; if (s == "r") return v.r; if (s == "z") return v.z;
  return nullptr;
}
int wl_mg_smooth(wl_mg* mg, int l, int it, float w, void* st) { WL_CHECK(l >= 0 && l < (int)mg->lv.size(), "level out of range"); return mg->smooth(l, it <= 0 ? 4 : it, w, wl_stream(st)); }
int wl_mg_smoother_kind(const wl_mg* mg, int l) {   // 0 one kernel per pass, 1 temporally blocked (one cell per thread), 2 blocked pair kernels (constant coefficients)
  if (l < 0 || l >= (int)mg->lv.size()) return -1;
  const wl_mg::Level& p = mg->lv[(size_t)l];
  if (mg->pair_slab(p)) return 2;
  if (p.part && mg->use_zsplit && mg->use_fused && !p.dist) return 3;
  if (!(mg->use_fused && wl::gsrb_fused_ok(p.x_, mg->perdir, p.dist))) return 0;
  return wl::gsrb_pair_ok(p.x_, p.cl) ? 2 : 1;
}
int wl_mg_level_is_const(const wl_mg* mg, int l) { return (l >= 0 && l < (int)mg->lv.size()) ? mg->lv[(size_t)l].cl.on : 0; }
int wl_mg_set_fused(wl_mg* mg, int on) { mg->use_fused = (on & 1) != 0; mg->store_eps = (on & 2) == 0; wl::gsrb_pair_enable((on & 4) == 0); mg->use_tail = (on & 8) == 0; mg->use_zsplit = (on & 16) == 0; wl::tail_lds_enable((on & 32) == 0); mg->use_xdefer = (on & 64) == 0; mg->overlap_smooth = (on & 128) == 0; return 0; }
int wl_mg_vcycle(wl_mg* mg, int l, float w, void* st) { WL_CHECK(l >= 0 && l + 1 < (int)mg->lv.size(), "level out of range"); return mg->vcycle(l, w, wl_stream(st), false); }
int wl_mg_solve(wl_mg* mg, double tol, int itmx, int* n, double* r1, float* rinf, void* st) { return mg->solve(tol, itmx <= 0 ? 32 : itmx, n, r1, rinf, wl_stream(st)); }
int wl_mg_history(const wl_mg* mg, int16_t* out, int cap) { const int n = (int)mg->n.size(); for (int k = 0; k < n && k < cap; k++) out[k] = mg->n[(size_t)k]; return n; }
int wl_mg_last_log(const wl_mg* mg, double* r1, double* rinf, double* w, int cap) {
  const int n = (int)mg->log_r1.size();
  for (int k = 0; k < n && k < cap; k++) { r1[k] = mg->log_r1[(size_t)k]; rinf[k] = mg->log_rinf[(size_t)k]; w[k] = mg->log_w[(size_t)k]; }
  return n;
}
}  // extern "C"
