// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
//
// Literal CPU restatement of the time-step hot path of TzuYaoHuang/WaterLily.jl
// (pure Julia; no Julia runtime exists in this pipeline, so the reference itself cannot be
// executed).  One function per reference function, same loop ranges, same statement order.
// Every function cites the reference file:line it follows (paths relative to /root/reference).
//
// PINNING: this restatement is pinned by the reference's own analytical / known-answer tests
// (test/test_poisson.jl, test/test_flow.jl, test/test_core.jl, test/test_bodies.jl,
// test/test_metrics.jl) which tests/test_oracle_*.py re-run against it.  Reductions
// (sum/maximum/dot) come from Julia Base/BLAS in the reference and have unspecified
// association order => bitwise parity of reductions is unpinned; tolerance-level only.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
//
// Conventions: indices are Julia 1-based everywhere (CI<D>::I[d] in 1..n[d]); arrays are
// column-major (x fastest, component slowest) exactly like the Julia arrays.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <vector>

namespace wlo {

// ----------------------------------------------------------------------------------------------
// CartesianIndex / ranges                                                  src/core.jl:26-61
// ----------------------------------------------------------------------------------------------
template <int D>
struct CI {
  int I[D];
  int& operator[](int d) { return I[d]; }
  int operator[](int d) const { return I[d]; }
  CI operator+(const CI& o) const { CI r; for (int d = 0; d < D; d++) r.I[d] = I[d] + o.I[d]; return r; }
  CI operator-(const CI& o) const { CI r; for (int d = 0; d < D; d++) r.I[d] = I[d] - o.I[d]; return r; }
  CI operator*(int s) const { CI r; for (int d = 0; d < D; d++) r.I[d] = I[d] * s; return r; }
};
// δ(i,I): unit offset in (1-based) direction i                              src/core.jl:39-40
template <int D> inline CI<D> delta(int i) { CI<D> r; for (int d = 0; d < D; d++) r.I[d] = (d == i - 1) ? 1 : 0; return r; }
// CIj(j,I,k): replace j-th component by k                                   src/core.jl:31
template <int D> inline CI<D> CIj(int j, CI<D> I, int k) { I.I[j - 1] = k; return I; }

template <int D>
struct Box {  // CartesianIndices(lo:hi), inclusive, iterated column-major (x fastest)
  int lo[D], hi[D];
  long length() const { long n = 1; for (int d = 0; d < D; d++) n *= std::max(0, hi[d] - lo[d] + 1); return n; }
};

// serial column-major loop == Julia `@simd for I ∈ R` (backend="SIMD", src/core.jl:146-155);
// with WLO_OMP the outermost dimension is shared between threads, which is the analogue of the
// KernelAbstractions CPU backend (src/core.jl:134-145).  All @loop bodies are order independent.
template <int D, class F>
inline void for_box(const Box<D>& R, F&& f) {
  if constexpr (D == 2) {
#ifdef WLO_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int j = R.lo[1]; j <= R.hi[1]; j++)
      for (int i = R.lo[0]; i <= R.hi[0]; i++) { CI<2> I{{i, j}}; f(I); }
  } else {
#ifdef WLO_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int k = R.lo[2]; k <= R.hi[2]; k++)
      for (int j = R.lo[1]; j <= R.hi[1]; j++)
        for (int i = R.lo[0]; i <= R.hi[0]; i++) { CI<3> I{{i, j, k}}; f(I); }
  }
}
// strictly serial variant (used where the loop carries a reduction)
template <int D, class F>
inline void for_box_serial(const Box<D>& R, F&& f) {
  if constexpr (D == 2) {
    for (int j = R.lo[1]; j <= R.hi[1]; j++)
      for (int i = R.lo[0]; i <= R.hi[0]; i++) { CI<2> I{{i, j}}; f(I); }
  } else {
    for (int k = R.lo[2]; k <= R.hi[2]; k++)
      for (int j = R.lo[1]; j <= R.hi[1]; j++)
        for (int i = R.lo[0]; i <= R.hi[0]; i++) { CI<3> I{{i, j, k}}; f(I); }
  }
}

// ----------------------------------------------------------------------------------------------
// array views (non-owning).  S: scalar (Ng...), V: vector (Ng...,D), TT: tensor (Ng...,D,D)
// ----------------------------------------------------------------------------------------------
template <class T, int D>
struct S {
  T* p = nullptr;
  int n[D] = {};
  long len() const { long l = 1; for (int d = 0; d < D; d++) l *= n[d]; return l; }
  inline long off(const CI<D>& I) const {
    long o = 0, s = 1;
    for (int d = 0; d < D; d++) { o += (long)(I.I[d] - 1) * s; s *= n[d]; }
    return o;
  }
  inline T& operator()(const CI<D>& I) const { return p[off(I)]; }
};
template <class T, int D>
struct V {
  T* p = nullptr;
  int n[D] = {};
  long cs() const { long l = 1; for (int d = 0; d < D; d++) l *= n[d]; return l; }
  long len() const { return cs() * D; }
  inline long off(const CI<D>& I) const {
    long o = 0, s = 1;
    for (int d = 0; d < D; d++) { o += (long)(I.I[d] - 1) * s; s *= n[d]; }
    return o;
  }
  inline T& operator()(const CI<D>& I, int i) const { return p[off(I) + (long)(i - 1) * cs()]; }
  S<T, D> comp(int i) const { S<T, D> s; s.p = p + (long)(i - 1) * cs(); for (int d = 0; d < D; d++) s.n[d] = n[d]; return s; }
};
template <class T, int D>
struct TT {
  T* p = nullptr;
  int n[D] = {};
  long cs() const { long l = 1; for (int d = 0; d < D; d++) l *= n[d]; return l; }
  long len() const { return cs() * D * D; }
  inline long off(const CI<D>& I) const {
    long o = 0, s = 1;
    for (int d = 0; d < D; d++) { o += (long)(I.I[d] - 1) * s; s *= n[d]; }
    return o;
  }
  inline T& operator()(const CI<D>& I, int i, int j) const { return p[off(I) + (long)((i - 1) + (j - 1) * D) * cs()]; }
};

// inside(a;buff=1)                                                          src/core.jl:47
template <int D> inline Box<D> inside(const int* n, int buff = 1) {
  Box<D> b; for (int d = 0; d < D; d++) { b.lo[d] = 1 + buff; b.hi[d] = n[d] - buff; } return b;
}
// inside_u(dims,j): 3:dims[j]-1 in j, 2:dims[k] elsewhere                   src/core.jl:55-57
template <int D> inline Box<D> inside_u(const int* n, int j) {
  Box<D> b; for (int d = 0; d < D; d++) { if (d == j - 1) { b.lo[d] = 3; b.hi[d] = n[d] - 1; } else { b.lo[d] = 2; b.hi[d] = n[d]; } } return b;
}
// slice(dims,i,j,low): index i in dim j, low:dims[k] elsewhere              src/core.jl:188-190
template <int D> inline Box<D> slice(const int* n, int i, int j, int low = 1) {
  Box<D> b; for (int d = 0; d < D; d++) { if (d == j - 1) { b.lo[d] = i; b.hi[d] = i; } else { b.lo[d] = low; b.hi[d] = n[d]; } } return b;
}
template <int D> inline Box<D> whole(const int* n) { return inside<D>(n, 0); }

// loc(i,I,T) = I - 1.5 - δ(i)/2   (i=0: cell centre)                        src/core.jl:177
template <class T, int D> inline void loc(int i, const CI<D>& I, T* x) {
  for (int d = 0; d < D; d++) x[d] = (T)I.I[d] - (T)1.5 - (T)((d == i - 1) ? 1 : 0) / (T)2;
}

// ----------------------------------------------------------------------------------------------
// boundary functions.  uBC is either a tuple (constant per component) or a Function (i,x,t).
// The Function form also carries the analytic time derivative that the reference obtains by
// ForwardDiff (src/Flow.jl:72-73; src/core.jl:280) — AD is out of scope, a callback stands in.
// ----------------------------------------------------------------------------------------------
typedef double (*bc_fn_t)(int i, const double* x, double t, void* user);
template <class T, int D>
struct UBC {
  bool is_fn = false;
  T U[D] = {};
  bc_fn_t fn = nullptr;     // uBC(i,x,t)
  bc_fn_t dfn = nullptr;    // d uBC/dt (i,x,t)
  void* user = nullptr;
  inline T eval(int i, const T* x, T t) const {
    if (!is_fn) return U[i - 1];
    double xd[D]; for (int d = 0; d < D; d++) xd[d] = (double)x[d];
    return (T)fn(i, xd, (double)t, user);
  }
};
struct PerDir {  // perdir tuple as a mask over 1-based directions
  unsigned mask = 0;
  bool has(int j) const { return (mask >> (j - 1)) & 1u; }
  bool empty() const { return mask == 0; }
};

// BC!(a,uBC,saveexit,perdir,t)                                              src/core.jl:200-219
// NB the reference evaluates uBC at loc(i,I) with the DEFAULT T=Float32 (src/core.jl:177).
template <class T, int D>
void BC(const V<T, D>& a, const UBC<T, D>& uBC, bool saveexit, PerDir perdir, T t) {
  const int* N = a.n;
  auto ub = [&](int i, const CI<D>& I) -> T {
    float xf[D]; loc<float, D>(i, I, xf);
    T x[D]; for (int d = 0; d < D; d++) x[d] = (T)xf[d];
    return uBC.eval(i, x, t);
  };
  for (int i = 1; i <= D; i++)
    for (int j = 1; j <= D; j++) {
      if (perdir.has(j)) {
        for_box<D>(slice<D>(N, 1, j), [&](const CI<D>& I) { a(I, i) = a(CIj<D>(j, I, N[j - 1] - 1), i); });      // :205
        for_box<D>(slice<D>(N, N[j - 1], j), [&](const CI<D>& I) { a(I, i) = a(CIj<D>(j, I, 2), i); });            // :206
      } else if (i == j) {  // normal direction, Dirichlet                                                           :208-212
        for (int s = 1; s <= 2; s++)
          for_box<D>(slice<D>(N, s, j), [&](const CI<D>& I) { a(I, i) = ub(i, I); });
        if (!saveexit || i > 1)
          for_box<D>(slice<D>(N, N[j - 1], j), [&](const CI<D>& I) { a(I, i) = ub(i, I); });
      } else {  // tangential, Neumann                                                                              :214-215
        const CI<D> dj = delta<D>(j);
        if (!uBC.is_fn) {  // tuple: (i,x,t)->U[i] inlines and @fastmath folds U+a-U to a; test/test_core.jl:27-30 demands ==
          for_box<D>(slice<D>(N, 1, j), [&](const CI<D>& I) { a(I, i) = a(I + dj, i); });
          for_box<D>(slice<D>(N, N[j - 1], j), [&](const CI<D>& I) { a(I, i) = a(I - dj, i); });
        } else {
          for_box<D>(slice<D>(N, 1, j), [&](const CI<D>& I) { a(I, i) = ub(i, I) + a(I + dj, i) - ub(i, I + dj); });
          for_box<D>(slice<D>(N, N[j - 1], j), [&](const CI<D>& I) { a(I, i) = ub(i, I) + a(I - dj, i) - ub(i, I - dj); });
        }
      }
    }
}

// Julia Base pairwise summation (mapreduce_impl, blksize 1024) restated for T accumulators.
template <class T, class F>
T pairwise_sum(F&& f, long ifirst, long ilast) {  // inclusive, 0-based
  if (ifirst > ilast) return (T)0;
  if (ifirst == ilast) return f(ifirst);
  if (ilast - ifirst < 1024) {
    T v = f(ifirst) + f(ifirst + 1);
    for (long i = ifirst + 2; i <= ilast; i++) v += f(i);
    return v;
  }
  long imid = ifirst + ((ilast - ifirst) >> 1);
  return pairwise_sum<T>(f, ifirst, imid) + pairwise_sum<T>(f, imid + 1, ilast);
}
// sum over a box view, in column-major order of the view (what `sum(@view a[R])` iterates)
template <class T, int D, class F>
T box_sum(const Box<D>& R, F&& f) {
  std::vector<T> tmp; tmp.reserve((size_t)R.length());
  for_box_serial<D>(R, [&](const CI<D>& I) { tmp.push_back(f(I)); });
  return pairwise_sum<T>([&](long i) { return tmp[(size_t)i]; }, 0, (long)tmp.size() - 1);
}

// exitBC!(u,u⁰,Δt)                                                          src/core.jl:226-233
template <class T, int D>
void exitBC(const V<T, D>& u, const V<T, D>& u0, T dt) {
  const int* N = u.n;
  int Nm1[D]; for (int d = 0; d < D; d++) Nm1[d] = N[d] - 1;
  Box<D> exitR = slice<D>(Nm1, N[0], 1, 2);                 // exit slice excluding ghosts       :228
  Box<D> inR = slice<D>(Nm1, 2, 1, 2);
  T U = box_sum<T, D>(inR, [&](const CI<D>& I) { return u(I, 1); }) / (T)exitR.length();          // :229
  const CI<D> d1 = delta<D>(1);
  for_box<D>(exitR, [&](const CI<D>& I) { u(I, 1) = u0(I, 1) - U * dt * (u0(I, 1) - u0(I - d1, 1)); });  // :230
  T ou = box_sum<T, D>(exitR, [&](const CI<D>& I) { return u(I, 1); }) / (T)exitR.length() - U;   // :231
  for_box<D>(exitR, [&](const CI<D>& I) { u(I, 1) -= ou; });                                        // :232
}

// perBC!(a,perdir)                                                          src/core.jl:239-243
template <class T, int D>
void perBC(const S<T, D>& a, PerDir perdir) {
  if (perdir.empty()) return;
  const int* N = a.n;
  for (int j = 1; j <= D; j++) {
    if (!perdir.has(j)) continue;
    for_box<D>(slice<D>(N, 1, j), [&](const CI<D>& I) { a(I) = a(CIj<D>(j, I, N[j - 1] - 1)); });
    for_box<D>(slice<D>(N, N[j - 1], j), [&](const CI<D>& I) { a(I) = a(CIj<D>(j, I, 2)); });
  }
}

// ----------------------------------------------------------------------------------------------
// Flow.jl stencil primitives                                                src/Flow.jl:1-36
// ----------------------------------------------------------------------------------------------
template <class T> inline T median3(T a, T b, T c) {  // src/Flow.jl:27-36
  if (a > b) { if (b >= c) return b; if (a > c) return c; }
  else       { if (b <= c) return b; if (a < c) return c; }
  return a;
}
enum Scheme { QUICK = 0, VANLEER = 1, CDS = 2 };
template <class T> inline T quick(T u, T c, T d) { return median3<T>((5 * c + 2 * d - u) / 6, c, median3<T>(10 * c - 9 * u, c, d)); }  // :4
template <class T> inline T vanLeer(T u, T c, T d) { return (c <= std::min(u, d) || c >= std::max(u, d)) ? c : c + (d - c) * (c - u) / (d - u); }  // :5
template <class T> inline T cds(T u, T c, T d) { return (c + d) / 2; }  // :6
template <class T> inline T lam(int s, T u, T c, T d) { return s == QUICK ? quick<T>(u, c, d) : (s == VANLEER ? vanLeer<T>(u, c, d) : cds<T>(u, c, d)); }

// f is "component i of u" viewed as a scalar field: index CI(I,i)
template <class T, int D> inline T d_scalar(int a, const CI<D>& I, const S<T, D>& f) { return f(I) - f(I - delta<D>(a)); }        // ∂ :1
template <class T, int D> inline T d_vec(int a, const CI<D>& I, const V<T, D>& u) { return u(I + delta<D>(a), a) - u(I, a); }      // ∂ :2
template <class T, int D> inline T phi(int a, const CI<D>& I, const S<T, D>& f) { return (f(I) + f(I - delta<D>(a))) / 2; }        // ϕ :3
template <class T, int D> inline T phiu(int a, const CI<D>& I, const S<T, D>& f, T u, int s) {                                      // ϕu :8
  const CI<D> d = delta<D>(a);
  return u > 0 ? u * lam<T>(s, f(I - d - d), f(I - d), f(I)) : u * lam<T>(s, f(I + d), f(I), f(I - d));
}
template <class T, int D> inline T phiuP(int a, const CI<D>& Ip, const CI<D>& I, const S<T, D>& f, T u, int s) {                    // ϕuP :9
  const CI<D> d = delta<D>(a);
  return u > 0 ? u * lam<T>(s, f(Ip), f(I - d), f(I)) : u * lam<T>(s, f(I + d), f(I), f(I - d));
}
template <class T, int D> inline T phiuL(int a, const CI<D>& I, const S<T, D>& f, T u, int s) {                                     // ϕuL :10
  const CI<D> d = delta<D>(a);
  return u > 0 ? u * phi<T, D>(a, I, f) : u * lam<T>(s, f(I + d), f(I), f(I - d));
}
template <class T, int D> inline T phiuR(int a, const CI<D>& I, const S<T, D>& f, T u, int s) {                                     // ϕuR :11
  const CI<D> d = delta<D>(a);
  return u < 0 ? u * phi<T, D>(a, I, f) : u * lam<T>(s, f(I - d - d), f(I - d), f(I));
}
template <class T, int D> inline T divu(const CI<D>& I, const V<T, D>& u) {  // div :13-19
  T init = 0; for (int i = 1; i <= D; i++) init += d_vec<T, D>(i, I, u); return init;
}
template <class T, int D> inline T muddn(const CI<D>& I, int i, const TT<T, D>& mu, const V<T, D>& f) {  // μddn :20-26 (I carries comp i)
  T s = 0;
  for (int j = 1; j <= D; j++) s += mu(I, i, j) * (f(I + delta<D>(j), i) - f(I - delta<D>(j), i));
  return s / 2;
}

// conv_diff!(r,u,Φ,λ;ν,perdir)                                              src/Flow.jl:38-62
template <class T, int D>
void conv_diff(const V<T, D>& r, const V<T, D>& u, const S<T, D>& Phi, int scheme, T nu, PerDir perdir) {
  std::fill(r.p, r.p + r.len(), (T)0);                                                          // :39
  const int* N = u.n;
  for (int i = 1; i <= D; i++)
    for (int j = 1; j <= D; j++) {
      const bool tagper = perdir.has(j);
      const S<T, D> f = u.comp(i);       // u[:,i] == CI(I,i) indexing
      const S<T, D> uj = u.comp(j);      // ϕ(i,CI(I,j),u) interpolates component j along direction i
      const CI<D> dj = delta<D>(j);
      // lowerBoundary!                                                                          :45,56,60-61
      if (!tagper) {
        for_box<D>(slice<D>(N, 2, j, 2), [&](const CI<D>& I) {
          r(I, i) += phiuL<T, D>(j, I, f, phi<T, D>(i, I, uj), scheme) - nu * d_scalar<T, D>(j, I, f);
        });
      } else {
        for_box<D>(slice<D>(N, 2, j, 2), [&](const CI<D>& I) {
          Phi(I) = phiuP<T, D>(j, CIj<D>(j, I, N[j - 1] - 2), I, f, phi<T, D>(i, I, uj), scheme) - nu * d_scalar<T, D>(j, I, f);
          r(I, i) += Phi(I);
        });
      }
      // inner cells                                                                             :47-49
      for_box<D>(inside_u<D>(N, j), [&](const CI<D>& I) {
        Phi(I) = phiu<T, D>(j, I, f, phi<T, D>(i, I, uj), scheme) - nu * d_scalar<T, D>(j, I, f);
        r(I, i) += Phi(I);
      });
      for_box<D>(inside_u<D>(N, j), [&](const CI<D>& I) { r(I - dj, i) -= Phi(I); });
      // upperBoundary!                                                                          :51,57,62
      if (!tagper) {
        for_box<D>(slice<D>(N, N[j - 1], j, 2), [&](const CI<D>& I) {
          r(I - dj, i) += -phiuR<T, D>(j, I, f, phi<T, D>(i, I, uj), scheme) + nu * d_scalar<T, D>(j, I, f);
        });
      } else {
        for_box<D>(slice<D>(N, N[j - 1], j, 2), [&](const CI<D>& I) { r(I - dj, i) -= Phi(CIj<D>(j, I, 2)); });
      }
    }
}

// accelerate!(r,t,g,uBC)                                                    src/Flow.jl:69-73
template <class T, int D>
void accelerate(const V<T, D>& r, T t, const UBC<T, D>* g, const UBC<T, D>& uBC) {
  const bool hasg = (g != nullptr && g->is_fn);
  const bool hasU = uBC.is_fn;
  if (!hasg && !hasU) return;                                                                   // :69
  for (int i = 1; i <= D; i++)
    for_box<D>(whole<D>(r.n), [&](const CI<D>& I) {
      T x[D]; loc<T, D>(i, I, x);                                                               // loc(Ii,eltype(r)) :70
      double xd[D]; for (int d = 0; d < D; d++) xd[d] = (double)x[d];
      T a = 0;
      if (hasg) a += (T)g->fn(i, xd, (double)t, g->user);
      if (hasU) a += (T)uBC.dfn(i, xd, (double)t, uBC.user);
      r(I, i) += a;
    });
}

// ----------------------------------------------------------------------------------------------
// Poisson                                                                   src/Poisson.jl
// ----------------------------------------------------------------------------------------------
template <class T> inline T eps_of();
template <> inline float eps_of<float>() { return 1.1920929e-7f; }
template <> inline double eps_of<double>() { return 2.220446049250313e-16; }

template <class T, int D> inline T diagL(const CI<D>& I, const V<T, D>& L) {  // diag :49-55
  T s = 0; for (int i = 1; i <= D; i++) s -= (L(I, i) + L(I + delta<D>(i), i)); return s;
}
template <class T, int D> inline T mult(const CI<D>& I, const V<T, D>& L, const S<T, D>& Dg, const S<T, D>& x) {  // mult :70-76
  T s = x(I) * Dg(I);
  for (int i = 1; i <= D; i++) s += (x(I - delta<D>(i)) * L(I, i) + x(I + delta<D>(i)) * L(I + delta<D>(i), i));
  return s;
}
template <class T, int D> inline T gauss(const CI<D>& I, const S<T, D>& r, const V<T, D>& L, const S<T, D>& iD, const S<T, D>& x) {  // gauss :116-122
  T s = r(I);
  for (int i = 1; i <= D; i++) s -= (x(I - delta<D>(i)) * L(I, i) + x(I + delta<D>(i)) * L(I + delta<D>(i), i));
  return s * iD(I);
}

template <class T, int D>
struct Poisson {  // struct Poisson :22-39 — L,x,z alias caller arrays; D,iD,ϵ,r owned
  V<T, D> L;
  S<T, D> Dg, iD, x, eps, r, z;
  std::vector<int16_t> n;
  PerDir perdir;
  std::vector<T> own;  // storage for Dg,iD,eps,r (+ L,x,z when this level owns them)
  bool owns_Lxz = false;

  long ncell() const { return x.len(); }
  long n_inside() const { long l = 1; for (int d = 0; d < D; d++) l *= (x.n[d] - 2); return l; }

  void init(T* xp, T* Lp, T* zp, const int* dims, PerDir pd) {  // Poisson(x,L,z;perdir) :32-38
    perdir = pd;
    long len = 1; for (int d = 0; d < D; d++) len *= dims[d];
    own.assign((size_t)(4 * len), (T)0);
    auto mk = [&](T* p) { S<T, D> s; s.p = p; for (int d = 0; d < D; d++) s.n[d] = dims[d]; return s; };
    x = mk(xp); z = mk(zp);
    L.p = Lp; for (int d = 0; d < D; d++) L.n[d] = dims[d];
    r = mk(own.data()); eps = mk(own.data() + len); Dg = mk(own.data() + 2 * len); iD = mk(own.data() + 3 * len);
    set_diag();
  }
  void set_diag() {  // set_diag! :43-46
    for_box<D>(inside<D>(Dg.n), [&](const CI<D>& I) { Dg(I) = diagL<T, D>(I, L); });
    for_box<D>(inside<D>(Dg.n), [&](const CI<D>& I) { iD(I) = (Dg(I) == 0) ? Dg(I) : (T)1 / Dg(I); });
  }
  void update() { set_diag(); }  // :47
  // mult!(p,x): p.z = A x                                                                       :63-69
  void mult_into_z(const S<T, D>& xin) {
    perBC<T, D>(xin, perdir);
    std::fill(z.p, z.p + z.len(), (T)0);
    for_box<D>(inside<D>(z.n), [&](const CI<D>& I) { z(I) = mult<T, D>(I, L, Dg, xin); });
  }
  T sum_all(const S<T, D>& a) const { return pairwise_sum<T>([&](long i) { return a.p[i]; }, 0, a.len() - 1); }
  // residual!                                                                                   :92-98
  void residual() {
    perBC<T, D>(x, perdir);
    for_box<D>(inside<D>(r.n), [&](const CI<D>& I) { r(I) = (iD(I) == 0) ? (T)0 : z(I) - mult<T, D>(I, L, Dg, x); });
    T s = sum_all(r) / (T)n_inside();
    if (std::abs(s) <= 2 * eps_of<T>()) return;
    for_box<D>(inside<D>(r.n), [&](const CI<D>& I) { r(I) = r(I) - s; });
  }
  // increment!(p;ω)                                                                             :100-104
  void increment(T w) {
    perBC<T, D>(eps, perdir);
    for_box<D>(inside<D>(x.n), [&](const CI<D>& I) {
      r(I) = r(I) - w * mult<T, D>(I, L, Dg, eps);
      x(I) = x(I) + w * eps(I);
    });
  }
  // Jacobi!(p;it=1,ω=1)                                                                         :111-114
  void Jacobi(int it = 1, T w = 1) {
    for (int k = 0; k < it; k++) {
      for_box<D>(inside<D>(eps.n), [&](const CI<D>& I) { eps(I) = r(I) * iD(I); });
      increment(w);
    }
  }
  // GaussSeidelRB!(p;it=4,ω=1) with gauss_rb / half_rangek                                     :124-148
  void GaussSeidelRB(int it = 4, T w = 1) {
    for_box<D>(inside<D>(eps.n), [&](const CI<D>& I) { eps(I) = r(I) * iD(I); });               // :142
    perBC<T, D>(eps, perdir);                                                                   // :143
    Box<D> half;                                                                                // half_rangek :130-132
    for (int d = 0; d < D; d++) { half.lo[d] = 2; half.hi[d] = (d == D - 1) ? eps.n[d] / 2 : eps.n[d] - 1; }
    for (int k0 = 1; k0 <= it; k0++) {
      for_box<D>(half, [&](const CI<D>& Iv) {                                                   // gauss_rb :124-128
        int sf = 0; for (int d = 0; d < D - 1; d++) sf += Iv.I[d];
        int k = 2 * Iv.I[D - 1] - 1 - (sf + k0) % 2;
        CI<D> I = Iv; I.I[D - 1] = k;
        eps(I) = gauss<T, D>(I, r, L, iD, eps);
      });
    }
    increment(w);                                                                               // :147
  }
  T dot_all(const S<T, D>& a, const S<T, D>& b) const {  // ⋅ (BLAS in the reference: order unpinned)
    double s = 0; for (long i = 0; i < a.len(); i++) s += (double)a.p[i] * (double)b.p[i]; return (T)s;
  }
  T perdot(const S<T, D>& a, const S<T, D>& b) const {  // perdot :156-157
    if (perdir.empty()) return dot_all(a, b);
    double s = 0; for_box_serial<D>(inside<D>(a.n), [&](const CI<D>& I) { s += (double)a(I) * (double)b(I); }); return (T)s;
  }
  // pcg!(p;it=6)                                                                                :166-186
  void pcg(int it = 6) {
    for_box<D>(inside<D>(z.n), [&](const CI<D>& I) { z(I) = eps(I) = r(I) * iD(I); });
    T rho = dot_all(r, z);
    if (std::abs(rho) < 10 * eps_of<T>()) return;
    for (int i = 1; i <= it; i++) {
      perBC<T, D>(eps, perdir);
      for_box<D>(inside<D>(z.n), [&](const CI<D>& I) { z(I) = mult<T, D>(I, L, Dg, eps); });
      T alpha = rho / perdot(z, eps);
      if (std::abs(alpha) < (T)1e-2 || std::abs(alpha) > (T)1e2) return;
      for_box<D>(inside<D>(x.n), [&](const CI<D>& I) { x(I) += alpha * eps(I); r(I) -= alpha * z(I); });
      if (i == it) return;
      for_box<D>(inside<D>(z.n), [&](const CI<D>& I) { z(I) = r(I) * iD(I); });
      T rho2 = dot_all(r, z);
      if (std::abs(rho2) < 10 * eps_of<T>()) return;
      T beta = rho2 / rho;
      for_box<D>(inside<D>(eps.n), [&](const CI<D>& I) { eps(I) = beta * eps(I) + z(I); });
      rho = rho2;
    }
  }
  T L2() const { return dot_all(r, r); }                                                        // :189
  T L1() const { return pairwise_sum<T>([&](long i) { return std::abs(r.p[i]); }, 0, r.len() - 1); }  // :190
  T Linf() const { T m = 0; for (long i = 0; i < r.len(); i++) m = std::max(m, std::abs(r.p[i])); return m; }  // :191
  double l1n_tol(double tol) const { return (tol / 10) * (double)n_inside(); }                  // :194
  // solver!(p::Poisson;tol=2e-3,itmx=1e3)                                                       :212-223
  int solve(double tol = 2e-3, double itmx = 1e3) {
    double r1tol = l1n_tol(tol), rinftol = tol;
    residual(); T r1 = L1(); T rinf = Linf();
    int np = 0;
    while (np < itmx) {
      pcg(); r1 = L1(); rinf = Linf(); np++;
      if ((double)r1 < r1tol && (double)rinf < rinftol) break;
    }
    perBC<T, D>(x, perdir);
    n.push_back((int16_t)np);
    return np;
  }
};

// L₂(a) = Σ_{inside} a²  (sum of squares, no sqrt)                          src/Poisson.jl:188
template <class T, int D> double L2_inside(const S<T, D>& a) {
  double s = 0; for_box_serial<D>(inside<D>(a.n), [&](const CI<D>& I) { s += (double)a(I) * (double)a(I); }); return s;
}

// ----------------------------------------------------------------------------------------------
// MultiLevelPoisson                                                         src/MultiLevelPoisson.jl
// ----------------------------------------------------------------------------------------------
inline bool divisible(int N) { return (N % 2 == 0) && N > 4; }                                  // :52
template <int D> struct Mask { bool c[D]; };
template <int D> inline Mask<D> coarsen_mask(const int* N) { Mask<D> m; for (int d = 0; d < D; d++) m.c[d] = divisible(N[d]); return m; }  // :29
template <int D> inline Mask<D> coarsen_mask(const int* fine, const int* coarse) { Mask<D> m; for (int d = 0; d < D; d++) m.c[d] = coarse[d] < fine[d]; return m; }  // :31
// up(I,c)                                                                                       :6
template <int D> inline Box<D> up(const CI<D>& I, const Mask<D>& c) {
  Box<D> b; for (int d = 0; d < D; d++) { if (c.c[d]) { b.lo[d] = 2 * I.I[d] - 2; b.hi[d] = 2 * I.I[d] - 1; } else { b.lo[d] = b.hi[d] = I.I[d]; } } return b;
}
// down(I,c)                                                                                     :7
template <int D> inline CI<D> down(const CI<D>& I, const Mask<D>& c) {
  CI<D> r; for (int d = 0; d < D; d++) r.I[d] = c.c[d] ? (I.I[d] + 2) / 2 : I.I[d]; return r;
}
// upL(I,i,c)                                                                                    :9-11
template <int D> inline Box<D> upL(const CI<D>& I, int i, const Mask<D>& c) {
  Box<D> b;
  for (int d = 0; d < D; d++) {
    if (d == i - 1) { if (c.c[d]) { b.lo[d] = b.hi[d] = 2 * I.I[d] - 2; } else { b.lo[d] = b.hi[d] = I.I[d]; } }
    else { if (c.c[d]) { b.lo[d] = 2 * I.I[d] - 2; b.hi[d] = 2 * I.I[d] - 1; } else { b.lo[d] = b.hi[d] = I.I[d]; } }
  }
  return b;
}
template <class T, int D> inline T restrict1(const CI<D>& I, const S<T, D>& b, const Mask<D>& c) {  // restrict :13-19
  T s = 0; for_box_serial<D>(up<D>(I, c), [&](const CI<D>& J) { s += b(J); }); return s;
}
template <class T, int D> inline T restrictL1(const CI<D>& I, int i, const V<T, D>& b, const Mask<D>& c) {  // restrictL :20-26
  T s = 0; for_box_serial<D>(upL<D>(I, i, c), [&](const CI<D>& J) { s += b(J, i); });
  return c.c[i - 1] ? s / 2 : s;
}
// restrictL!(a,b,c;perdir)                                                                      :42-48
template <class T, int D>
void restrictL(const V<T, D>& a, const V<T, D>& b, const Mask<D>& c, PerDir perdir) {
  for (int i = 1; i <= D; i++)
    for_box<D>(inside<D>(a.n), [&](const CI<D>& I) { a(I, i) = restrictL1<T, D>(I, i, b, c); });
  UBC<T, D> zero;  // zero(SVector)
  BC<T, D>(a, zero, false, perdir, (T)0);                                                        // :47
}
template <class T, int D> void restrict_(const S<T, D>& a, const S<T, D>& b, const Mask<D>& c) {  // restrict! :49
  for_box<D>(inside<D>(a.n), [&](const CI<D>& I) { a(I) = restrict1<T, D>(I, b, c); });
}
template <class T, int D> void prolongate_(const S<T, D>& a, const S<T, D>& b, const Mask<D>& c) {  // prolongate! :50
  for_box<D>(inside<D>(a.n), [&](const CI<D>& I) { a(I) = b(down<D>(I, c)); });
}

template <class T, int D>
struct MultiLevelPoisson {  // :61-77
  std::vector<Poisson<T, D>*> levels;
  std::vector<std::vector<T>> store;  // coarse-level L,x,z storage
  std::vector<int16_t> n;
  PerDir perdir;
  // diagnostics of the last solve (what @log prints, :112,117)
  std::vector<double> log_r1, log_rinf, log_w;

  ~MultiLevelPoisson() { for (auto* p : levels) delete p; }
  static bool level_divisible(const Poisson<T, D>& l) { for (int d = 0; d < D; d++) if (divisible(l.x.n[d])) return true; return false; }  // :54
  // restrictML                                                                                  :33-41
  Poisson<T, D>* restrictML(const Poisson<T, D>& b) {
    const int* N = b.L.n;
    Mask<D> c = coarsen_mask<D>(N);
    int Na[D]; long len = 1;
    for (int d = 0; d < D; d++) { Na[d] = c.c[d] ? 1 + N[d] / 2 : N[d]; len *= Na[d]; }
    store.emplace_back((size_t)(len * (D + 2)), (T)0);
    T* base = store.back().data();
    V<T, D> aL; aL.p = base; for (int d = 0; d < D; d++) aL.n[d] = Na[d];
    restrictL<T, D>(aL, b.L, c, b.perdir);
    auto* p = new Poisson<T, D>();
    p->init(base + len * D, base, base + len * (D + 1), Na, b.perdir);
    return p;
  }
  void init(T* x, T* L, T* z, const int* dims, PerDir pd, int maxlevels = 10) {  // :68-76
    perdir = pd;
    auto* p0 = new Poisson<T, D>(); p0->init(x, L, z, dims, pd);
    levels.push_back(p0);
    store.reserve(64);
    while (level_divisible(*levels.back()) && (int)levels.size() <= maxlevels) levels.push_back(restrictML(*levels.back()));
    if (!(levels.size() > 2)) throw std::runtime_error("MultiLevelPoisson requires size=a2ⁿ, where n>2");
  }
  void update() {  // update! :79-86
    levels[0]->update();
    for (size_t l = 1; l < levels.size(); l++) {
      Mask<D> c = coarsen_mask<D>(levels[l - 1]->x.n, levels[l]->x.n);
      restrictL<T, D>(levels[l]->L, levels[l - 1]->L, c, levels[l - 1]->perdir);
      levels[l]->update();
    }
  }
  void Vcycle(int l /*0-based*/, T w) {  // Vcycle! :88-101
    Poisson<T, D>& fine = *levels[l]; Poisson<T, D>& coarse = *levels[l + 1];
    Mask<D> c = coarsen_mask<D>(fine.x.n, coarse.x.n);
    fine.Jacobi();
    restrict_<T, D>(coarse.r, fine.r, c);
    std::fill(coarse.x.p, coarse.x.p + coarse.x.len(), (T)0);
    if (l + 2 < (int)levels.size()) Vcycle(l + 1, w);
    coarse.GaussSeidelRB(4, w);                                                                 // smooth! :106
    prolongate_<T, D>(fine.eps, coarse.x, c);
    fine.increment(w);
  }
  // solver!(ml;tol=2e-3,itmx=32)                                                                :108-128
  int solve(double tol = 2e-3, int itmx = 32) {
    Poisson<T, D>& p = *levels[0];
    double r1tol = p.l1n_tol(tol), rinftol = tol;
    p.residual(); T r1 = p.L1(); T rinf = p.Linf(); T w = 1;
    int np = 0;
    log_r1.assign(1, (double)r1); log_rinf.assign(1, (double)rinf); log_w.assign(1, (double)w);
    while (np < itmx) {
      Vcycle(0, w);
      p.GaussSeidelRB(4, w);
      T rnew = p.L1(); rinf = p.Linf(); np++;
      log_r1.push_back((double)rnew); log_rinf.push_back((double)rinf); log_w.push_back((double)w);
      if (rnew >= r1) w = (T)std::max(0.2, 0.9 * (double)w);                                    // :118-119
      else if (rnew < r1) w = (T)std::min(1.0, 1.02 * (double)w);                               // :120-121
      r1 = rnew;
      if ((double)r1 < r1tol && (double)rinf < rinftol) break;
    }
    perBC<T, D>(p.x, perdir);
    n.push_back((int16_t)np);
    return np;
  }
};

// ----------------------------------------------------------------------------------------------
// Body: BDIM kernel moments + measure! for a closed-form sphere/circle      src/Body.jl, src/AutoBody.jl
// ----------------------------------------------------------------------------------------------
template <class T> inline T kern(T d) { return (1 + std::cos((T)M_PI * d)) / 2; }                                      // Body.jl:54
template <class T> inline T kern0(T d) { return (1 + d + std::sin((T)M_PI * d) / (T)M_PI) / 2; }                       // :55
template <class T> inline T kern1(T d) { return (1 - d * d) / 4 - (d * std::sin((T)M_PI * d) + (1 + std::cos((T)M_PI * d)) / (T)M_PI) / (2 * (T)M_PI); }  // :56
// NB: Julia's `eps(d)` in μ₀ is the float spacing at d, not eps(T): restated exactly.
template <class T> inline T eps_at(T d) { d = std::abs(d); if (d == 0) return std::numeric_limits<T>::denorm_min(); return std::nextafter(d, std::numeric_limits<T>::infinity()) - d; }
template <class T> inline T mu0(T d, T e) { return d / e < -1 + std::sqrt(eps_at<T>(d)) ? (T)0 : kern0<T>(std::min(d / e, (T)1)); }   // Body.jl:59
template <class T> inline T mu1(T d, T e) { return e * kern1<T>(std::min(std::max(d / e, (T)-1), (T)1)); }                             // Body.jl:60

template <class T, int D>
struct Body {  // kind 0: NoBody (Body.jl:81-83); kind 1: AutoBody(sdf = |m∘(x−c)|−R) — sphere / circle, or with an axis masked out
               // (m=0) the cylinder along that axis; kind 2: AutoBody(sdf = m·(x−c)), a plane with (not necessarily unit) normal m.
               // map(x,t) = x − vel·t is folded into c by the caller; its time derivative gives the body velocity vel (AutoBody.jl:36-37)
  int kind = 0;
  T c[D] = {};
  T R = 0;
  T m[D];
  T vel[D] = {};
  Body() { for (int k = 0; k < D; k++) m[k] = 1; }
  // sdf(body,x,t) — AutoBody.jl:21
  T sdf(const T* x) const {
    if (kind == 0) return std::numeric_limits<T>::infinity();
    T s = 0;
    if (kind == 2) { for (int k = 0; k < D; k++) s += m[k] * (x[k] - c[k]); return s; }
    for (int k = 0; k < D; k++) { const T dx = m[k] * (x[k] - c[k]); s += dx * dx; }
    return std::sqrt(s) - R;
  }
  // measure(body,x,t;fastd²)  — AutoBody.jl:29-37 with the gradient of the sdf in closed form
  void measure(const T* x, T fastd2, T& d, T* nrm, T* Vb) const {
    for (int k = 0; k < D; k++) { nrm[k] = 0; Vb[k] = 0; }
    if (kind == 0) { d = std::numeric_limits<T>::infinity(); return; }
    T rr = 0;
    if (kind == 2) d = sdf(x);
    else { T s = 0; for (int k = 0; k < D; k++) { const T dx = m[k] * (x[k] - c[k]); s += dx * dx; } rr = std::sqrt(s); d = rr - R; }
    if (d * d > fastd2) return;                                                                 // :31
    T g[D]; bool nan = false;
    if (kind == 2) { for (int k = 0; k < D; k++) g[k] = m[k]; }
    else { for (int k = 0; k < D; k++) g[k] = (m[k] * (x[k] - c[k])) / rr; }                     // gradient of the sdf :32
    for (int k = 0; k < D; k++) if (std::isnan(g[k])) nan = true;
    if (nan) return;                                                                            // :33
    T mm = 0; for (int k = 0; k < D; k++) mm += g[k] * g[k];                                     // pseudo-sdf :34-35
    mm = std::sqrt(mm); d /= mm; for (int k = 0; k < D; k++) nrm[k] = g[k] / mm;
    for (int k = 0; k < D; k++) Vb[k] = vel[k];                                                 // −J\ṁ with J = I, ṁ = −vel :36-37
  }
};

// ----------------------------------------------------------------------------------------------
// Flow                                                                      src/Flow.jl:114-257
// ----------------------------------------------------------------------------------------------
template <class T, int D>
struct Flow {
  V<T, D> u, u0, f, Vb, mu0v;
  S<T, D> p, sig;
  TT<T, D> mu1v;
  UBC<T, D> uBC;
  UBC<T, D> g; bool has_g = false;
  std::vector<T> dt;
  T nu = 0;
  bool exitBC_ = false;
  PerDir perdir;
  int scheme = QUICK;
  std::vector<T> store;
  int Ng[D];

  // Flow(N,uBC;Δt,ν,g,u0,perdir,exitBC,λ,T)                                                     :133-147
  // u0fn==nullptr -> ic_function(uBC) (:85-86,138)
  typedef double (*ic_fn_t)(int i, const double* x, void* user);
  void init(const int* N, const UBC<T, D>& ubc, T dt0, T nu_, PerDir pd, bool exitbc, int scheme_, ic_fn_t u0fn, void* u0user) {
    long cs = 1; for (int d = 0; d < D; d++) { Ng[d] = N[d] + 2; cs *= Ng[d]; }
    store.assign((size_t)(cs * (5 * D + 2 + D * D)), (T)0);
    T* b = store.data();
    auto mkV = [&](T* p_) { V<T, D> v; v.p = p_; for (int d = 0; d < D; d++) v.n[d] = Ng[d]; return v; };
    auto mkS = [&](T* p_) { S<T, D> s; s.p = p_; for (int d = 0; d < D; d++) s.n[d] = Ng[d]; return s; };
    u = mkV(b); b += cs * D; u0 = mkV(b); b += cs * D; f = mkV(b); b += cs * D; Vb = mkV(b); b += cs * D; mu0v = mkV(b); b += cs * D;
    p = mkS(b); b += cs; sig = mkS(b); b += cs;
    mu1v.p = b; for (int d = 0; d < D; d++) mu1v.n[d] = Ng[d];
    uBC = ubc; nu = nu_; perdir = pd; exitBC_ = exitbc; scheme = scheme_; dt.assign(1, dt0);
    // apply!(u0,u): c[Ii] = f(last(Ii), loc(Ii,eltype(c)))                                      :82,140
    for (int i = 1; i <= D; i++)
      for_box<D>(whole<D>(Ng), [&](const CI<D>& I) {
        T x[D]; loc<T, D>(i, I, x);
        if (u0fn) { double xd[D]; for (int d = 0; d < D; d++) xd[d] = (double)x[d]; u(I, i) = (T)u0fn(i, xd, u0user); }
        else u(I, i) = uBC.eval(i, x, (T)0);
      });
    BC<T, D>(u, uBC, exitBC_, perdir, (T)0); exitBC<T, D>(u, u, (T)0);                            // :141
    std::copy(u.p, u.p + u.len(), u0.p);                                                         // :142
    std::fill(mu0v.p, mu0v.p + mu0v.len(), (T)1);                                                // :144
    UBC<T, D> zero; BC<T, D>(mu0v, zero, false, perdir, (T)0);                                   // :145
  }
  T sum_dt() const { T s = 0; for (T d_ : dt) s += d_; return s; }
  T time() const { T s = 0; for (size_t k = 0; k + 1 < dt.size(); k++) s += dt[k]; return s; }   // :174
  void scale_u(T s) {                                                                            // :211-214
    for (int i = 1; i <= D; i++) for_box<D>(inside<D>(Ng), [&](const CI<D>& I) { u(I, i) *= s; });
  }
  void BDIM() {                                                                                  // :176-180
    T dtl = dt.back();
    for (int i = 1; i <= D; i++) for_box<D>(whole<D>(Ng), [&](const CI<D>& I) { f(I, i) = u0(I, i) + dtl * f(I, i) - Vb(I, i); });
    for (int i = 1; i <= D; i++) for_box<D>(inside<D>(Ng), [&](const CI<D>& I) { u(I, i) += muddn<T, D>(I, i, mu1v, f) + Vb(I, i) + mu0v(I, i) * f(I, i); });
  }
  void mom_predict(T t0, T t1) {                                                                 // :190-196
    conv_diff<T, D>(f, u0, sig, scheme, nu, perdir);
    accelerate<T, D>(f, t0, has_g ? &g : nullptr, uBC);
    BDIM(); BC<T, D>(u, uBC, exitBC_, perdir, t1);
    if (exitBC_) exitBC<T, D>(u, u0, dt.back());
  }
  void mom_correct(T t) {                                                                        // :205-210
    conv_diff<T, D>(f, u, sig, scheme, nu, perdir);
    accelerate<T, D>(f, t, has_g ? &g : nullptr, uBC);
    BDIM(); scale_u((T)0.5); BC<T, D>(u, uBC, exitBC_, perdir, t);
  }
  template <class Pois>
  void mom_project(Pois& b, Poisson<T, D>& p0, T w, T t) {                                       // :223-232
    T dtl = w * dt.back();
    for_box<D>(inside<D>(Ng), [&](const CI<D>& I) { p0.z(I) = divu<T, D>(I, u); });
    for (long k = 0; k < p0.x.len(); k++) p0.x.p[k] *= dtl;
    b.solve();
    for (int i = 1; i <= D; i++)
      for_box<D>(inside<D>(Ng), [&](const CI<D>& I) { u(I, i) -= p0.L(I, i) * d_scalar<T, D>(i, I, p0.x); });
    for (long k = 0; k < p0.x.len(); k++) p0.x.p[k] /= dtl;
    BC<T, D>(u, uBC, exitBC_, perdir, t);
  }
  T CFL(T dt_max = 10) {                                                                         // :234-244
    for_box<D>(inside<D>(Ng), [&](const CI<D>& I) {
      T s = 0, z = 0;
      for (int i = 1; i <= D; i++) s += std::max(z, u(I + delta<D>(i), i)) + std::max(z, -u(I, i));
      sig(I) = s;
    });
    T m = -std::numeric_limits<T>::infinity();
    for (long k = 0; k < sig.len(); k++) m = std::max(m, sig.p[k]);                              // maximum(a.σ): WHOLE array (quirk Q1)
    return std::min(dt_max, (T)1 / (m + 5 * nu));
  }
  template <class Pois>
  void mom_step(Pois& b, Poisson<T, D>& p0) {                                                    // :156-167
    std::copy(u.p, u.p + u.len(), u0.p); scale_u((T)0);
    T t1 = sum_dt(); T t0 = t1 - dt.back();
    mom_predict(t0, t1);
    mom_project(b, p0, (T)1, t1);
    mom_correct(t1);
    mom_project(b, p0, (T)0.5, t1);
    dt.push_back(CFL());
  }
  // measure!(flow,body;t,ϵ)                                                  src/Body.jl:28-51
  void measure(const Body<T, D>& body, T t, T e) {
    if (body.kind == 0) return;                                                                  // Body.jl:83
    std::fill(Vb.p, Vb.p + Vb.len(), (T)0); std::fill(mu0v.p, mu0v.p + mu0v.len(), (T)1); std::fill(mu1v.p, mu1v.p + mu1v.len(), (T)0);
    T d2 = (T)(2 + e) * (T)(2 + e);
    for_box<D>(inside<D>(Ng), [&](const CI<D>& I) { T x[D]; loc<T, D>(0, I, x); sig(I) = body.sdf(x); });  // measure_sdf! :74
    for_box<D>(inside<D>(Ng), [&](const CI<D>& I) {
      if (sig(I) * sig(I) < d2) {
        for (int i = 1; i <= D; i++) {
          T x[D]; loc<T, D>(i, I, x);
          T di, ni[D], Vi[D]; body.measure(x, d2, di, ni, Vi);
          di = std::abs(di) <= (T)0.5 ? di : std::copysign(di, sig(I));                         // :35
          Vb(I, i) = Vi[i - 1];
          mu0v(I, i) = mu0<T>(di, e);
          for (int j = 1; j <= D; j++) mu1v(I, i, j) = mu1<T>(di, e) * ni[j - 1];
        }
      } else if (sig(I) < 0) {
        for (int i = 1; i <= D; i++) mu0v(I, i) = 0;
      }
    });
    UBC<T, D> zero;
    BC<T, D>(mu0v, zero, false, perdir, (T)0);                                                   // :49
    BC<T, D>(Vb, zero, exitBC_, perdir, (T)0);                                                   // :50
  }
};

// pressure_force(p,df,body,t)                                               src/Metrics.jl:116-133
template <class T, int D>
void pressure_force(const S<T, D>& p, const V<T, D>& df, const Body<T, D>& body, double* out) {
  std::fill(df.p, df.p + df.len(), (T)0);
  for_box<D>(inside<D>(p.n), [&](const CI<D>& I) {
    T x[D]; loc<T, D>(0, I, x);
    T d, nrm[D], Vv[D]; body.measure(x, (T)1, d, nrm, Vv);                                       // nds :116-119 (fastd²=1)
    T k = kern<T>(std::min(std::max(d, (T)-1), (T)1));
    for (int i = 1; i <= D; i++) df(I, i) = p(I) * (nrm[i - 1] * k);
  });
  for (int i = 1; i <= D; i++) { double s = 0; S<T, D> c = df.comp(i); for (long k = 0; k < c.len(); k++) s += (double)c.p[k]; out[i - 1] = s; }
}

// ∂(i,j,I,u) = ∂uᵢ/∂xⱼ at the centre of cell I                                     src/Metrics.jl:42-44
template <class T, int D> inline T dudx(int i, int j, const CI<D>& I, const V<T, D>& u) {
  if (i == j) return u(I + delta<D>(i), i) - u(I, i);                                              // ∂(i,I,u) src/Flow.jl:2
  const CI<D> Ip = I + delta<D>(j), Im = I - delta<D>(j);
  return (u(Ip, i) + u(Ip + delta<D>(i), i) - u(Im, i) - u(Im + delta<D>(i), i)) / 4;
}
// viscous_force(u,ν,df,body,t):  df[I,:] = -2ν·S(I,u)·nds(body,loc(0,I),t) over the inside cells, summed in Float64     src/Metrics.jl:140-154
template <class T, int D>
void viscous_force(const V<T, D>& u, T nu, const V<T, D>& df, const Body<T, D>& body, double* out) {
  std::fill(df.p, df.p + df.len(), (T)0);
  for_box<D>(inside<D>(df.n), [&](const CI<D>& I) {
    T x[D]; loc<T, D>(0, I, x);
    T d, nrm[D], Vv[D]; body.measure(x, (T)1, d, nrm, Vv);
    const T k = kern<T>(std::min(std::max(d, (T)-1), (T)1));
    T nds[D]; for (int j = 0; j < D; j++) nds[j] = nrm[j] * k;
    for (int i = 1; i <= D; i++) {
      T acc = 0;
      for (int j = 1; j <= D; j++) { const T Sij = (dudx<T, D>(i, j, I, u) + dudx<T, D>(j, i, I, u)) / 2; acc += ((-2 * nu) * Sij) * nds[j - 1]; }
      df(I, i) = acc;
    }
  });
  for (int i = 1; i <= D; i++) { double s = 0; S<T, D> c = df.comp(i); for (long k = 0; k < c.len(); k++) s += (double)c.p[k]; out[i - 1] = s; }
}

// cross(a,b): 3-D vector product; in 2-D the scalar a₁b₂−a₂b₁, which the broadcast df[I,:] .= … stores in every component
template <class T, int D> inline void cross_(const T* a, const T* b, T* out) {
  if (D == 2) { const T m = a[0] * b[1] - a[1] * b[0]; out[0] = m; out[1] = m; }
  else { out[0] = a[1] * b[2] - a[2] * b[1]; out[1] = a[2] * b[0] - a[0] * b[2]; out[2] = a[0] * b[1] - a[1] * b[0]; }
}
// pressure_moment(x₀,p,df,body,t): df[I,:] = p[I]·cross(loc(0,I)−x₀, nds)                     src/Metrics.jl:169-174
template <class T, int D>
void pressure_moment(const T* x0, const S<T, D>& p, const V<T, D>& df, const Body<T, D>& body, double* out) {
  std::fill(df.p, df.p + df.len(), (T)0);
  for_box<D>(inside<D>(p.n), [&](const CI<D>& I) {
    T x[D]; loc<T, D>(0, I, x);
    T d, nrm[D], Vv[D]; body.measure(x, (T)1, d, nrm, Vv);
    const T k = kern<T>(std::min(std::max(d, (T)-1), (T)1));
    T nds[D], r[D], c[D];
    for (int j = 0; j < D; j++) { nds[j] = nrm[j] * k; r[j] = x[j] - x0[j]; }
    cross_<T, D>(r, nds, c);
    for (int i = 1; i <= D; i++) df(I, i) = p(I) * c[i - 1];
  });
  for (int i = 1; i <= D; i++) { double s = 0; S<T, D> c = df.comp(i); for (long k = 0; k < c.len(); k++) s += (double)c.p[k]; out[i - 1] = s; }
}
// viscous_moment(x₀,u,ν,df,body,t): df[I,:] = −2ν·cross(loc(0,I)−x₀, S(I,u)·nds)              src/Metrics.jl:183-188
template <class T, int D>
void viscous_moment(const T* x0, const V<T, D>& u, T nu, const V<T, D>& df, const Body<T, D>& body, double* out) {
  std::fill(df.p, df.p + df.len(), (T)0);
  for_box<D>(inside<D>(df.n), [&](const CI<D>& I) {
    T x[D]; loc<T, D>(0, I, x);
    T d, nrm[D], Vv[D]; body.measure(x, (T)1, d, nrm, Vv);
    const T k = kern<T>(std::min(std::max(d, (T)-1), (T)1));
    T nds[D], r[D], sn[D], c[D];
    for (int j = 0; j < D; j++) { nds[j] = nrm[j] * k; r[j] = x[j] - x0[j]; }
    for (int i = 1; i <= D; i++) {
      T acc = 0;
      for (int j = 1; j <= D; j++) { const T Sij = (dudx<T, D>(i, j, I, u) + dudx<T, D>(j, i, I, u)) / 2; acc += Sij * nds[j - 1]; }
      sn[i - 1] = acc;
    }
    cross_<T, D>(r, sn, c);
    for (int i = 1; i <= D; i++) df(I, i) = (-2 * nu) * c[i - 1];
  });
  for (int i = 1; i <= D; i++) { double s = 0; S<T, D> c = df.comp(i); for (long k = 0; k < c.len(); k++) s += (double)c.p[k]; out[i - 1] = s; }
}

// ----------------------------------------------------------------------------------------------
// Simulation                                                                src/WaterLily.jl:86-149
// ----------------------------------------------------------------------------------------------
template <class T, int D>
struct Simulation {
  double U = 1, L = 1; T eps = 1;
  Flow<T, D> flow;
  Body<T, D> body;
  MultiLevelPoisson<T, D> pois;
  void init(const int* N, const UBC<T, D>& ubc, double L_, double U_, T dt0, T nu, T eps_, PerDir pd, bool exitbc, int scheme,
            typename Flow<T, D>::ic_fn_t u0fn, void* u0user, const Body<T, D>& body_, const UBC<T, D>* g) {
    L = L_; U = U_; eps = eps_; body = body_;
    flow.init(N, ubc, dt0, nu, pd, exitbc, scheme, u0fn, u0user);                                // :103
    if (g && g->is_fn) { flow.g = *g; flow.has_g = true; }
    flow.measure(body, (T)0, eps);                                                               // :104
    pois.init(flow.p.p, flow.mu0v.p, flow.sig.p, flow.Ng, pd);                                   // :105, pois_ctor default :97
  }
  double sim_time() const { return (double)flow.time() * U / L; }                               // :117
  void measure() { flow.measure(body, flow.sum_dt(), eps); pois.update(); }                      // :146-149
  void step(bool remeasure) { if (remeasure) measure(); flow.mom_step(pois, *pois.levels[0]); }  // :136-139
  int step_until(double t_end, bool remeasure, int max_steps) {                                  // :128-135
    int steps = 0;
    while (sim_time() < t_end && steps < max_steps) { step(remeasure); steps++; }
    return steps;
  }
};

}  // namespace wlo
