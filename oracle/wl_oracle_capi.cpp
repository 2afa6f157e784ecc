// ORACLE — TEST INFRASTRUCTURE ONLY (see wl_oracle.hpp header).  Flat C ABI over the templated
// restatement so tests/bench can drive it from Python ctypes.  dtype: 0 = Float32, 1 = Float64.
#include "wl_oracle.hpp"

#include <cmath>
#include <limits>
#include <string>
#ifdef WLO_OMP
#include <omp.h>
#endif

using namespace wlo;

static thread_local std::string g_err;

#define DISPATCH(dtype, D, ...)                                                              \
  do {                                                                                        \
    if ((dtype) == 0 && (D) == 2) { using T = float; constexpr int DD = 2; __VA_ARGS__; }       \
    else if ((dtype) == 0 && (D) == 3) { using T = float; constexpr int DD = 3; __VA_ARGS__; }  \
    else if ((dtype) == 1 && (D) == 2) { using T = double; constexpr int DD = 2; __VA_ARGS__; } \
    else if ((dtype) == 1 && (D) == 3) { using T = double; constexpr int DD = 3; __VA_ARGS__; } \
    else { g_err = "bad dtype/D"; }                                                           \
  } while (0)

template <class T, int D> static S<T, D> mkS(void* p, const int* n) { S<T, D> s; s.p = (T*)p; for (int d = 0; d < D; d++) s.n[d] = n[d]; return s; }
template <class T, int D> static V<T, D> mkV(void* p, const int* n) { V<T, D> s; s.p = (T*)p; for (int d = 0; d < D; d++) s.n[d] = n[d]; return s; }
template <class T, int D> static TT<T, D> mkT(void* p, const int* n) { TT<T, D> s; s.p = (T*)p; for (int d = 0; d < D; d++) s.n[d] = n[d]; return s; }
template <class T, int D> static UBC<T, D> mkU(const double* U, bc_fn_t fn, bc_fn_t dfn, void* user) {
  UBC<T, D> u; for (int d = 0; d < D; d++) u.U[d] = U ? (T)U[d] : (T)0;
  if (fn) { u.is_fn = true; u.fn = fn; u.dfn = dfn; u.user = user; }
  return u;
}
static PerDir mkP(unsigned m) { PerDir p; p.mask = m; return p; }

struct Handle { int dtype, D; void* obj; };

// general analytic body: kind 1 |m∘(x−c)|−R, kind 2 m·(x−c); vel = translation velocity (may be NULL)
template <class T, int DD>
static Body<T, DD> mkBody(int kind, const double* c, double R, const double* m, const double* vel) {
  Body<T, DD> b; b.kind = kind; b.R = (T)R;
  for (int d = 0; d < DD; d++) { b.c[d] = c ? (T)c[d] : (T)0; if (m) b.m[d] = (T)m[d]; if (vel) b.vel[d] = (T)vel[d]; }
  return b;
}

extern "C" {

const char* wlo_last_error() { return g_err.c_str(); }
int wlo_max_threads() {
#ifdef WLO_OMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void wlo_set_threads(int n) {
#ifdef WLO_OMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

// ---- built-in boundary functors (native, thread-safe; avoid Python callbacks on big grids) -------
// id 0: uBC(i,x,t) = i==1 ? t : 0   (test/test_flow.jl:163 "circle in accelerating flow");  id 1: its d/dt
static double bc_accel_x(int i, const double*, double t, void*) { return i == 1 ? t : 0.0; }
static double dbc_accel_x(int i, const double*, double, void*) { return i == 1 ? 1.0 : 0.0; }
void* wlo_builtin_fn(int id) { return id == 0 ? (void*)bc_accel_x : (id == 1 ? (void*)dbc_accel_x : nullptr); }

// ---- scalar helpers (double precision entry, evaluated in the requested dtype) ---------------
double wlo_quick(int dtype, double u, double c, double d) { return dtype == 0 ? (double)quick<float>((float)u, (float)c, (float)d) : quick<double>(u, c, d); }
double wlo_vanLeer(int dtype, double u, double c, double d) { return dtype == 0 ? (double)vanLeer<float>((float)u, (float)c, (float)d) : vanLeer<double>(u, c, d); }
double wlo_cds(int dtype, double u, double c, double d) { return dtype == 0 ? (double)cds<float>((float)u, (float)c, (float)d) : cds<double>(u, c, d); }
double wlo_median(double a, double b, double c) { return median3<double>(a, b, c); }
double wlo_mu0(int dtype, double d, double e) { return dtype == 0 ? (double)mu0<float>((float)d, (float)e) : mu0<double>(d, e); }
double wlo_mu1(int dtype, double d, double e) { return dtype == 0 ? (double)mu1<float>((float)d, (float)e) : mu1<double>(d, e); }
double wlo_kern(int dtype, double d) { return dtype == 0 ? (double)kern<float>((float)d) : kern<double>(d); }
// 1-D flux helpers on a double vector f of length n (1-based index I); which: 0 ϕu, 1 ϕuL, 2 ϕuR, 3 ϕuP, 4 ϕ
double wlo_flux1d(int which, const double* f, int n, int I, int Ip, double u, int scheme) {
  // emulate a 1-D array by a 2-D one with a single column
  int dims[2] = {n, 1}; S<double, 2> s = mkS<double, 2>((void*)f, dims);
  CI<2> Ii{{I, 1}}, Ipp{{Ip, 1}};
  switch (which) {
    case 0: return phiu<double, 2>(1, Ii, s, u, scheme);
    case 1: return phiuL<double, 2>(1, Ii, s, u, scheme);
    case 2: return phiuR<double, 2>(1, Ii, s, u, scheme);
    case 3: return phiuP<double, 2>(1, Ipp, Ii, s, u, scheme);
    default: return phi<double, 2>(1, Ii, s);
  }
}
void wlo_loc(int D, int i, const int* I, double* x) {
  if (D == 2) { CI<2> c{{I[0], I[1]}}; loc<double, 2>(i, c, x); } else { CI<3> c{{I[0], I[1], I[2]}}; loc<double, 3>(i, c, x); }
}
int wlo_divisible(int n) { return divisible(n) ? 1 : 0; }
void wlo_down(int D, const int* I, const int* c, int* out) {
  if (D == 2) { CI<2> a{{I[0], I[1]}}; Mask<2> m{{c[0] != 0, c[1] != 0}}; CI<2> r = down<2>(a, m); out[0] = r.I[0]; out[1] = r.I[1]; }
  else { CI<3> a{{I[0], I[1], I[2]}}; Mask<3> m{{c[0] != 0, c[1] != 0, c[2] != 0}}; CI<3> r = down<3>(a, m); for (int d = 0; d < 3; d++) out[d] = r.I[d]; }
}
void wlo_up(int D, const int* I, const int* c, int* lo, int* hi) {
  if (D == 2) { CI<2> a{{I[0], I[1]}}; Mask<2> m{{c[0] != 0, c[1] != 0}}; Box<2> b = up<2>(a, m); for (int d = 0; d < 2; d++) { lo[d] = b.lo[d]; hi[d] = b.hi[d]; } }
  else { CI<3> a{{I[0], I[1], I[2]}}; Mask<3> m{{c[0] != 0, c[1] != 0, c[2] != 0}}; Box<3> b = up<3>(a, m); for (int d = 0; d < 3; d++) { lo[d] = b.lo[d]; hi[d] = b.hi[d]; } }
}

// ---- leaf array operations -------------------------------------------------------------------
void wlo_BC(int dtype, int D, void* a, const int* dims, const double* U, bc_fn_t fn, void* user, int saveexit, unsigned perdir, double t) {
  DISPATCH(dtype, D, (BC<T, DD>(mkV<T, DD>(a, dims), mkU<T, DD>(U, fn, nullptr, user), saveexit != 0, mkP(perdir), (T)t)));
}
void wlo_perBC(int dtype, int D, void* a, const int* dims, unsigned perdir) { DISPATCH(dtype, D, (perBC<T, DD>(mkS<T, DD>(a, dims), mkP(perdir)))); }
void wlo_exitBC(int dtype, int D, void* u, void* u0, const int* dims, double dt) { DISPATCH(dtype, D, (exitBC<T, DD>(mkV<T, DD>(u, dims), mkV<T, DD>(u0, dims), (T)dt))); }
void wlo_conv_diff(int dtype, int D, void* r, void* u, void* Phi, const int* dims, int scheme, double nu, unsigned perdir) {
  DISPATCH(dtype, D, (conv_diff<T, DD>(mkV<T, DD>(r, dims), mkV<T, DD>(u, dims), mkS<T, DD>(Phi, dims), scheme, (T)nu, mkP(perdir))));
}
void wlo_BDIM(int dtype, int D, void* u, void* u0, void* f, void* Vb, void* mu0p, void* mu1p, const int* dims, double dt) {
  DISPATCH(dtype, D, {
    Flow<T, DD> fl; for (int d = 0; d < DD; d++) fl.Ng[d] = dims[d];
    fl.u = mkV<T, DD>(u, dims); fl.u0 = mkV<T, DD>(u0, dims); fl.f = mkV<T, DD>(f, dims); fl.Vb = mkV<T, DD>(Vb, dims);
    fl.mu0v = mkV<T, DD>(mu0p, dims); fl.mu1v = mkT<T, DD>(mu1p, dims); fl.dt.assign(1, (T)dt); fl.BDIM();
  });
}
void wlo_scale_u(int dtype, int D, void* u, const int* dims, double s) {
  DISPATCH(dtype, D, { Flow<T, DD> fl; for (int d = 0; d < DD; d++) fl.Ng[d] = dims[d]; fl.u = mkV<T, DD>(u, dims); fl.scale_u((T)s); });
}
void wlo_div(int dtype, int D, void* z, void* u, const int* dims) {
  DISPATCH(dtype, D, { auto zz = mkS<T, DD>(z, dims); auto uu = mkV<T, DD>(u, dims); for_box<DD>(inside<DD>(dims), [&](const CI<DD>& I) { zz(I) = divu<T, DD>(I, uu); }); });
}
// u[I,i] -= L[I,i]*∂(i,I,x)   (src/Flow.jl:227-229)
void wlo_project(int dtype, int D, void* u, void* L, void* x, const int* dims) {
  DISPATCH(dtype, D, {
    auto uu = mkV<T, DD>(u, dims); auto LL = mkV<T, DD>(L, dims); auto xx = mkS<T, DD>(x, dims);
    for (int i = 1; i <= DD; i++) for_box<DD>(inside<DD>(dims), [&](const CI<DD>& I) { uu(I, i) -= LL(I, i) * d_scalar<T, DD>(i, I, xx); });
  });
}
double wlo_CFL(int dtype, int D, void* u, void* sig, const int* dims, double nu) {
  double out = 0;
  DISPATCH(dtype, D, { Flow<T, DD> fl; for (int d = 0; d < DD; d++) fl.Ng[d] = dims[d]; fl.u = mkV<T, DD>(u, dims); fl.sig = mkS<T, DD>(sig, dims); fl.nu = (T)nu; out = (double)fl.CFL(); });
  return out;
}
void wlo_restrict(int dtype, int D, void* a, const int* adims, void* b, const int* bdims) {
  DISPATCH(dtype, D, (restrict_<T, DD>(mkS<T, DD>(a, adims), mkS<T, DD>(b, bdims), coarsen_mask<DD>(bdims, adims))));
}
void wlo_prolongate(int dtype, int D, void* a, const int* adims, void* b, const int* bdims) {
  DISPATCH(dtype, D, (prolongate_<T, DD>(mkS<T, DD>(a, adims), mkS<T, DD>(b, bdims), coarsen_mask<DD>(adims, bdims))));
}
void wlo_restrictL(int dtype, int D, void* a, const int* adims, void* b, const int* bdims, unsigned perdir) {
  DISPATCH(dtype, D, (restrictL<T, DD>(mkV<T, DD>(a, adims), mkV<T, DD>(b, bdims), coarsen_mask<DD>(bdims, adims), mkP(perdir))));
}
double wlo_L2_inside(int dtype, int D, void* a, const int* dims) { double o = 0; DISPATCH(dtype, D, (o = L2_inside<T, DD>(mkS<T, DD>(a, dims)))); return o; }
// pressure_force(p,df,body)   body = sphere(c,R)
void wlo_pressure_force(int dtype, int D, void* p, void* df, const int* dims, const double* c, double R, double* out) {
  DISPATCH(dtype, D, { Body<T, DD> b; b.kind = 1; for (int d = 0; d < DD; d++) b.c[d] = (T)c[d]; b.R = (T)R; pressure_force<T, DD>(mkS<T, DD>(p, dims), mkV<T, DD>(df, dims), b, out); });
}
// viscous_force(u,ν,df,body)   body = sphere(c,R)
void wlo_viscous_force(int dtype, int D, void* u, double nu, void* df, const int* dims, const double* c, double R, double* out) {
  DISPATCH(dtype, D, { Body<T, DD> b; b.kind = 1; for (int d = 0; d < DD; d++) b.c[d] = (T)c[d]; b.R = (T)R; viscous_force<T, DD>(mkV<T, DD>(u, dims), (T)nu, mkV<T, DD>(df, dims), b, out); });
}

// measure(body,x,t;fastd²) at one point: out = {d, n[D], V[D]} (as doubles)
void wlo_body_measure(int dtype, int D, int kind, const double* c, double R, const double* m, const double* vel, const double* x, double fastd2, double* out) {
  DISPATCH(dtype, D, {
    Body<T, DD> b = mkBody<T, DD>(kind, c, R, m, vel);
    T xx[DD], d, n[DD], V[DD]; for (int k = 0; k < DD; k++) xx[k] = (T)x[k];
    b.measure(xx, std::isinf(fastd2) ? std::numeric_limits<T>::infinity() : (T)fastd2, d, n, V);
    out[0] = (double)d; for (int k = 0; k < DD; k++) { out[1 + k] = (double)n[k]; out[1 + DD + k] = (double)V[k]; }
  });
}
void wlo_pressure_force_body(int dtype, int D, void* p, void* df, const int* dims, int kind, const double* c, double R, const double* m, double* out) {
  DISPATCH(dtype, D, { Body<T, DD> b = mkBody<T, DD>(kind, c, R, m, nullptr); pressure_force<T, DD>(mkS<T, DD>(p, dims), mkV<T, DD>(df, dims), b, out); });
}
void wlo_pressure_moment_body(int dtype, int D, const double* x0, void* p, void* df, const int* dims, int kind, const double* c, double R, const double* m, double* out) {
  DISPATCH(dtype, D, { Body<T, DD> b = mkBody<T, DD>(kind, c, R, m, nullptr); T xx[DD]; for (int k = 0; k < DD; k++) xx[k] = (T)x0[k];
                       pressure_moment<T, DD>(xx, mkS<T, DD>(p, dims), mkV<T, DD>(df, dims), b, out); });
}
void wlo_viscous_moment_body(int dtype, int D, const double* x0, void* u, double nu, void* df, const int* dims, int kind, const double* c, double R, const double* m, double* out) {
  DISPATCH(dtype, D, { Body<T, DD> b = mkBody<T, DD>(kind, c, R, m, nullptr); T xx[DD]; for (int k = 0; k < DD; k++) xx[k] = (T)x0[k];
                       viscous_moment<T, DD>(xx, mkV<T, DD>(u, dims), (T)nu, mkV<T, DD>(df, dims), b, out); });
}
void wlo_viscous_force_body(int dtype, int D, void* u, double nu, void* df, const int* dims, int kind, const double* c, double R, const double* m, double* out) {
  DISPATCH(dtype, D, { Body<T, DD> b = mkBody<T, DD>(kind, c, R, m, nullptr); viscous_force<T, DD>(mkV<T, DD>(u, dims), (T)nu, mkV<T, DD>(df, dims), b, out); });
}

// ---- Poisson / MultiLevelPoisson handle (x,L,z alias the caller's arrays) ---------------------
void* wlo_pois_create(int dtype, int D, void* x, void* L, void* z, const int* dims, unsigned perdir, int multilevel) {
  Handle* h = new Handle{dtype, D, nullptr};
  try {
    DISPATCH(dtype, D, {
      if (multilevel) { auto* ml = new MultiLevelPoisson<T, DD>(); try { ml->init((T*)x, (T*)L, (T*)z, dims, mkP(perdir)); } catch (...) { delete ml; throw; } h->obj = ml; }
      else { auto* p = new Poisson<T, DD>(); p->init((T*)x, (T*)L, (T*)z, dims, mkP(perdir)); h->obj = p; }
    });
  } catch (const std::exception& e) { g_err = e.what(); delete h; return nullptr; }
  h->dtype = dtype | (multilevel ? 0x100 : 0);
  return h;
}
#define PH_ML(h) (((Handle*)(h))->dtype & 0x100)
#define PH_DT(h) (((Handle*)(h))->dtype & 0xff)
#define PH_D(h) (((Handle*)(h))->D)
#define WITH_LEVEL(h, l, ...)                                                                                    \
  DISPATCH(PH_DT(h), PH_D(h), {                                                                                  \
    Poisson<T, DD>* P = PH_ML(h) ? ((MultiLevelPoisson<T, DD>*)((Handle*)(h))->obj)->levels[(size_t)(l)] : (Poisson<T, DD>*)((Handle*)(h))->obj; \
    __VA_ARGS__; \
  })
void wlo_pois_destroy(void* h) {
  if (!h) return;
  DISPATCH(PH_DT(h), PH_D(h), { if (PH_ML(h)) delete (MultiLevelPoisson<T, DD>*)((Handle*)h)->obj; else delete (Poisson<T, DD>*)((Handle*)h)->obj; });
  delete (Handle*)h;
}
int wlo_pois_nlevels(void* h) { int n = 1; DISPATCH(PH_DT(h), PH_D(h), { if (PH_ML(h)) n = (int)((MultiLevelPoisson<T, DD>*)((Handle*)h)->obj)->levels.size(); }); return n; }
void wlo_pois_level_dims(void* h, int l, int* dims) { WITH_LEVEL(h, l, { for (int d = 0; d < DD; d++) dims[d] = P->x.n[d]; }); }
void* wlo_pois_level_field(void* h, int l, const char* name) {
  void* out = nullptr; std::string s(name);
  WITH_LEVEL(h, l, {
    if (s == "L") out = P->L.p; else if (s == "D") out = P->Dg.p; else if (s == "iD") out = P->iD.p; else if (s == "x") out = P->x.p;
    else if (s == "eps") out = P->eps.p; else if (s == "r") out = P->r.p; else if (s == "z") out = P->z.p;
  });
  return out;
}
int wlo_pois_solve(void* h, double tol, int itmx) {
  int n = -1;
  DISPATCH(PH_DT(h), PH_D(h), {
    if (PH_ML(h)) n = ((MultiLevelPoisson<T, DD>*)((Handle*)h)->obj)->solve(tol, itmx < 0 ? 32 : itmx);
    else n = ((Poisson<T, DD>*)((Handle*)h)->obj)->solve(tol, itmx < 0 ? 1e3 : (double)itmx);
  });
  return n;
}
int wlo_pois_log(void* h, double* r1, double* rinf, double* w, int cap) {
  int n = 0;
  DISPATCH(PH_DT(h), PH_D(h), {
    if (PH_ML(h)) { auto* ml = (MultiLevelPoisson<T, DD>*)((Handle*)h)->obj; n = (int)ml->log_r1.size(); for (int k = 0; k < n && k < cap; k++) { r1[k] = ml->log_r1[(size_t)k]; rinf[k] = ml->log_rinf[(size_t)k]; w[k] = ml->log_w[(size_t)k]; } }
  });
  return n;
}
void wlo_pois_update(void* h) {
  DISPATCH(PH_DT(h), PH_D(h), { if (PH_ML(h)) ((MultiLevelPoisson<T, DD>*)((Handle*)h)->obj)->update(); else ((Poisson<T, DD>*)((Handle*)h)->obj)->update(); });
}
// op: 0 residual!, 1 Jacobi!, 2 GaussSeidelRB!(it,ω), 3 increment!(ω), 4 pcg!, 5 mult!(p, arg) [arg = array pointer], 6 set_diag!
void wlo_pois_op(void* h, int l, int op, int it, double w, void* arg) {
  WITH_LEVEL(h, l, {
    switch (op) {
      case 0: P->residual(); break;
      case 1: P->Jacobi(it <= 0 ? 1 : it, (T)w); break;
      case 2: P->GaussSeidelRB(it <= 0 ? 4 : it, (T)w); break;
      case 3: P->increment((T)w); break;
      case 4: P->pcg(it <= 0 ? 6 : it); break;
      case 5: P->mult_into_z(mkS<T, DD>(arg, P->x.n)); break;
      case 6: P->set_diag(); break;
    }
  });
}
void wlo_pois_vcycle(void* h, int l, double w) {
  DISPATCH(PH_DT(h), PH_D(h), { if (PH_ML(h)) ((MultiLevelPoisson<T, DD>*)((Handle*)h)->obj)->Vcycle(l, (T)w); });
}
// which: 0 L₁, 1 L∞, 2 L₂ of level l
double wlo_pois_norm(void* h, int l, int which) {
  double o = 0; WITH_LEVEL(h, l, { o = which == 0 ? (double)P->L1() : (which == 1 ? (double)P->Linf() : (double)P->L2()); }); return o;
}
int wlo_pois_nhist(void* h, int* out, int cap) {
  int n = 0;
  DISPATCH(PH_DT(h), PH_D(h), {
    const std::vector<int16_t>& v = PH_ML(h) ? ((MultiLevelPoisson<T, DD>*)((Handle*)h)->obj)->n : ((Poisson<T, DD>*)((Handle*)h)->obj)->n;
    n = (int)v.size(); for (int k = 0; k < n && k < cap; k++) out[k] = v[(size_t)k];
  });
  return n;
}

// ---- Simulation handle -----------------------------------------------------------------------
typedef double (*ic_fn_c)(int i, const double* x, void* user);
void* wlo_sim_create(int dtype, int D, const int* N, const double* U, bc_fn_t ufn, bc_fn_t dufn, double L, double Uscale,
                     double dt0, double nu, double eps, unsigned perdir, int exitBC, int scheme, ic_fn_c u0fn,
                     int body_kind, const double* c, double R, bc_fn_t gfn, void* user) {
  Handle* h = new Handle{dtype, D, nullptr};
  try {
    DISPATCH(dtype, D, {
      auto* s = new Simulation<T, DD>();
      Body<T, DD> b; b.kind = body_kind; if (body_kind) { for (int d = 0; d < DD; d++) b.c[d] = (T)c[d]; b.R = (T)R; }
      UBC<T, DD> g = mkU<T, DD>(nullptr, gfn, nullptr, user);
      try { s->init(N, mkU<T, DD>(U, ufn, dufn, user), L, Uscale, (T)dt0, (T)nu, (T)eps, mkP(perdir), exitBC != 0, scheme, u0fn, user, b, gfn ? &g : nullptr); }
      catch (...) { delete s; throw; }
      h->obj = s;
    });
  } catch (const std::exception& e) { g_err = e.what(); delete h; return nullptr; }
  return h;
}
#define SIM(h, ...) DISPATCH(((Handle*)(h))->dtype, ((Handle*)(h))->D, { auto* sim = (Simulation<T, DD>*)((Handle*)(h))->obj; __VA_ARGS__; })
void wlo_sim_destroy(void* h) { if (!h) return; SIM(h, delete sim); delete (Handle*)h; }
void* wlo_sim_field(void* h, const char* name) {
  void* out = nullptr; std::string s(name);
  SIM(h, {
    if (s == "u") out = sim->flow.u.p; else if (s == "u0") out = sim->flow.u0.p; else if (s == "f") out = sim->flow.f.p; else if (s == "p") out = sim->flow.p.p;
    else if (s == "sigma") out = sim->flow.sig.p; else if (s == "V") out = sim->flow.Vb.p; else if (s == "mu0") out = sim->flow.mu0v.p; else if (s == "mu1") out = sim->flow.mu1v.p;
  });
  return out;
}
void wlo_sim_step(void* h, int remeasure) { SIM(h, sim->step(remeasure != 0)); }
int wlo_sim_step_until(void* h, double t_end, int remeasure, int max_steps) { int n = 0; SIM(h, n = sim->step_until(t_end, remeasure != 0, max_steps)); return n; }
void wlo_sim_measure(void* h) { SIM(h, sim->measure()); }
// replace the simulation's body (a moving body: new centre and velocity), to be followed by wlo_sim_measure / a remeasuring step
void wlo_sim_set_body(void* h, int kind, const double* c, double R, const double* m, const double* vel) {
  SIM(h, (sim->body = mkBody<T, DD>(kind, c, R, m, vel)));
}
double wlo_sim_time(void* h) { double t = 0; SIM(h, t = sim->sim_time()); return t; }
double wlo_sim_flow_time(void* h) { double t = 0; SIM(h, t = (double)sim->flow.time()); return t; }
int wlo_sim_dt(void* h, double* out, int cap) { int n = 0; SIM(h, { n = (int)sim->flow.dt.size(); for (int k = 0; k < n && k < cap; k++) out[k] = (double)sim->flow.dt[(size_t)k]; }); return n; }
int wlo_sim_nhist(void* h, int* out, int cap) { int n = 0; SIM(h, { n = (int)sim->pois.n.size(); for (int k = 0; k < n && k < cap; k++) out[k] = sim->pois.n[(size_t)k]; }); return n; }
int wlo_sim_nlevels(void* h) { int n = 0; SIM(h, n = (int)sim->pois.levels.size()); return n; }
void wlo_sim_level_dims(void* h, int l, int* dims) { SIM(h, { for (int d = 0; d < DD; d++) dims[d] = sim->pois.levels[(size_t)l]->x.n[d]; }); }
void* wlo_sim_level_field(void* h, int l, const char* name) {
  void* out = nullptr; std::string s(name);
  SIM(h, {
    Poisson<T, DD>* P = sim->pois.levels[(size_t)l];
    if (s == "L") out = P->L.p; else if (s == "D") out = P->Dg.p; else if (s == "iD") out = P->iD.p; else if (s == "x") out = P->x.p;
    else if (s == "eps") out = P->eps.p; else if (s == "r") out = P->r.p; else if (s == "z") out = P->z.p;
  });
  return out;
}
void wlo_sim_pressure_force(void* h, double* out) { SIM(h, (pressure_force<T, DD>(sim->flow.p, sim->flow.f, sim->body, out))); }
void wlo_sim_pressure_moment(void* h, const double* x0, double* out) {
  SIM(h, { T xx[DD]; for (int k = 0; k < DD; k++) xx[k] = (T)x0[k]; pressure_moment<T, DD>(xx, sim->flow.p, sim->flow.f, sim->body, out); });
}
void wlo_sim_viscous_moment(void* h, const double* x0, double* out) {
  SIM(h, { T xx[DD]; for (int k = 0; k < DD; k++) xx[k] = (T)x0[k]; viscous_moment<T, DD>(xx, sim->flow.u, sim->flow.nu, sim->flow.f, sim->body, out); });
}
void wlo_sim_viscous_force(void* h, double* out) { SIM(h, (viscous_force<T, DD>(sim->flow.u, sim->flow.nu, sim->flow.f, sim->body, out))); }
double wlo_sim_pois_norm(void* h, int which) { double o = 0; SIM(h, { auto* P = sim->pois.levels[0]; o = which == 0 ? (double)P->L1() : (which == 1 ? (double)P->Linf() : (double)P->L2()); }); return o; }
// sub-phases of mom_step! for per-phase parity checks: 0 u⁰.=u;scale_u!(0)  1 mom_predict!  2 mom_project!(w=1)  3 mom_correct!  4 mom_project!(w=.5)  5 push!(Δt,CFL)
void wlo_sim_phase(void* h, int phase) {
  SIM(h, {
    auto& a = sim->flow; T t1 = a.sum_dt(); T t0 = t1 - a.dt.back();
    switch (phase) {
      case 0: std::copy(a.u.p, a.u.p + a.u.len(), a.u0.p); a.scale_u((T)0); break;
      case 1: a.mom_predict(t0, t1); break;
      case 2: a.mom_project(sim->pois, *sim->pois.levels[0], (T)1, t1); break;
      case 3: a.mom_correct(t1); break;
      case 4: a.mom_project(sim->pois, *sim->pois.levels[0], (T)0.5, t1); break;
      case 5: a.dt.push_back(a.CFL()); break;
    }
  });
}
}  // extern "C"
