// Access-shape scan for the smoother's memory mixes (VERDICT r02 "next" #1a).  Trivial arithmetic, no halo, no LDS: what the device
// delivers for a z-marching workgroup as a function of the tile shape, workgroup size, loads in flight and cache hints, on arrays with
// the REAL layout (514-float rows, 514x514 planes, 514 planes: the 512^3 level with ghosts).
//   mixes:  B = kernel B's (read e, r, x; write r', x in place; 20 B/cell)     A = kernel A's (read r; write r', e; 12 B/cell)
//           C = copy (read 1, write 1; 8 B/cell)                                 R = read-only (3 streams, 12 B/cell)
//   shapes: tile W x H cells of T threads (float2 per lane, rows split into W/2 lanes), z-marched in chunks;
//           "span": a workgroup owns S contiguous floats of every plane (full rows: the whole x extent)
// build: hipcc -O3 --offload-arch=gfx950 -o shape_probe shape_probe.hip ; run: ./shape_probe > profiles/r03_shape_probe.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

enum { MIX_B = 0, MIX_A = 1, MIX_C = 2, MIX_R = 3, MIX_P = 4 };   // P: the projection tail's mix: read 3 + 1, write the 3 in place + 1 (32 B/cell)
struct Arr { float* e; float* r; float* x; float* ro; float* eo; };

template <int NT> __device__ __forceinline__ float2 ldv(const float* p) {
  if (NT) return make_float2(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1));
  return *reinterpret_cast<const float2*>(p);
}
template <int NT> __device__ __forceinline__ void stv(float* p, float2 v) {
  if (NT) { __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p + 1); }
  else *reinterpret_cast<float2*>(p) = v;
}
template <int NT> __device__ __forceinline__ float2 ld2(const float* p) {
  if (NT) { typedef float v2 __attribute__((ext_vector_type(2))); v2 t = __builtin_nontemporal_load(reinterpret_cast<const v2*>(p)); return make_float2(t.x, t.y); }
  return *reinterpret_cast<const float2*>(p);
}
template <int NT> __device__ __forceinline__ void st2(float* p, float2 v) {
  if (NT) { typedef float v2 __attribute__((ext_vector_type(2))); v2 t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<v2*>(p)); }
  else *reinterpret_cast<float2*>(p) = v;
}

// one plane's worth of work for NV float2 slots per thread
template <int MIX, int NV, int NT> struct Regs { float2 a[NV], b[NV], c[NV], d[NV]; };
template <int MIX, int NV, int NT> __device__ __forceinline__ void load_plane(Regs<MIX, NV, NT>& q, const Arr& A, const size_t* o, size_t zo) {
#pragma unroll
  for (int v = 0; v < NV; v++) {
    if (MIX == MIX_B || MIX == MIX_R) { q.a[v] = ld2<NT>(A.e + o[v] + zo); q.b[v] = ld2<NT>(A.r + o[v] + zo); q.c[v] = ld2<NT>(A.x + o[v] + zo); }
    if (MIX == MIX_A || MIX == MIX_C) { q.b[v] = ld2<NT>(A.r + o[v] + zo); }
    if (MIX == MIX_P) { q.a[v] = ld2<NT>(A.e + o[v] + zo); q.b[v] = ld2<NT>(A.r + o[v] + zo); q.c[v] = ld2<NT>(A.x + o[v] + zo); q.d[v] = ld2<NT>(A.ro + o[v] + zo); }
  }
}
template <int MIX, int NV, int NT> __device__ __forceinline__ void store_plane(Regs<MIX, NV, NT>& q, const Arr& A, const size_t* o, size_t zo, float w, float2& acc) {
#pragma unroll
  for (int v = 0; v < NV; v++) {
    if (MIX == MIX_B) {
      float2 b = q.b[v], c = q.c[v], a = q.a[v];
      b.x -= w * a.x; b.y -= w * a.y; c.x += w * a.x; c.y += w * a.y;
      st2<NT>(A.ro + o[v] + zo, b); st2<NT>(A.x + o[v] + zo, c);
    }
    if (MIX == MIX_A) { float2 b = q.b[v]; st2<NT>(A.ro + o[v] + zo, b); b.x *= w; b.y *= w; st2<NT>(A.eo + o[v] + zo, b); }
    if (MIX == MIX_C) { st2<NT>(A.ro + o[v] + zo, q.b[v]); }
    if (MIX == MIX_P) { float2 a = q.a[v], b = q.b[v], c = q.c[v], d = q.d[v]; a.x -= w * d.x; a.y -= w * d.y; b.x -= w * d.x; b.y -= w * d.y; c.x -= w * d.x; c.y -= w * d.y; d.x *= w; d.y *= w;
      st2<NT>(A.e + o[v] + zo, a); st2<NT>(A.r + o[v] + zo, b); st2<NT>(A.x + o[v] + zo, c); st2<NT>(A.eo + o[v] + zo, d); }
    if (MIX == MIX_R) { acc.x += q.a[v].x + q.b[v].x + q.c[v].x; acc.y += q.a[v].y + q.b[v].y + q.c[v].y; }
  }
}

// tile W x H cells, T threads, NV = W*H/(2T) float2 per thread and plane; P planes of loads in flight (1 or 2)
template <int MIX, int W, int H, int T, int P, int NT>
__global__ void __launch_bounds__(T) k_tile(Arr A, int nx, int ny, int nz, int pitch, size_t psz, int zc, float w, float* sink, int oi = 2, int oj = 1, int ok = 1) {
  constexpr int NV = W * H / (2 * T);
  static_assert(NV >= 1, "tile too small");
  const int ntx = (nx + W - 1) / W, nty = (ny + H - 1) / H, ntiles = ntx * nty;
  const unsigned h = blockIdx.x, q = h & 7u, s = h >> 3;
  const unsigned per = (unsigned)((ntiles + 7) >> 3);
  const int ch = (int)(s / per);
  const int tl = (int)(q * per + (s - (unsigned)ch * per));
  if (tl >= ntiles) return;
  const int tx = tl % ntx, ty = tl / ntx;
  size_t o[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) {
    const int slot = threadIdx.x + v * T, row = slot / (W / 2), col = (slot % (W / 2)) * 2;
    int i = tx * W + col, j = ty * H + row;
    if (i > nx - 2) i = nx - 2;
    if (j > ny - 1) j = ny - 1;                     // clamp (ragged edge tiles re-do the last row: harmless for a probe)
    o[v] = (size_t)(oj + j) * pitch + (size_t)(oi + i);   // interior cells start at column 2 (8-byte aligned pairs), row 1
  }
  const int k0 = ok + ch * zc, k1 = min(k0 + zc, ok + nz);
  float2 acc = {0.f, 0.f};
  Regs<MIX, NV, NT> r0, r1;
  load_plane<MIX, NV, NT>(r0, A, o, (size_t)k0 * psz);
  if (P == 2) load_plane<MIX, NV, NT>(r1, A, o, (size_t)min(k0 + 1, k1 - 1) * psz);
  for (int k = k0; k < k1; k += P) {
    if (P == 1) {
      Regs<MIX, NV, NT> n;
      load_plane<MIX, NV, NT>(n, A, o, (size_t)min(k + 1, k1 - 1) * psz);
      store_plane<MIX, NV, NT>(r0, A, o, (size_t)k * psz, w, acc);
      r0 = n;
    } else {
      Regs<MIX, NV, NT> n0, n1;
      load_plane<MIX, NV, NT>(n0, A, o, (size_t)min(k + 2, k1 - 1) * psz);
      store_plane<MIX, NV, NT>(r0, A, o, (size_t)k * psz, w, acc);
      load_plane<MIX, NV, NT>(n1, A, o, (size_t)min(k + 3, k1 - 1) * psz);
      if (k + 1 < k1) store_plane<MIX, NV, NT>(r1, A, o, (size_t)(k + 1) * psz, w, acc);
      r0 = n0; r1 = n1;
    }
  }
  if (MIX == MIX_R && acc.x + acc.y == 1.2345f) sink[0] = acc.x;
}

// span: a workgroup owns S contiguous floats of every plane (the plane as a 1-D array incl. ghost columns), T threads, z-marched
template <int MIX, int S, int T, int P, int NT>
__global__ void __launch_bounds__(T) k_span(Arr A, int nz, size_t plane_floats, size_t psz, int zc, float w, float* sink) {
  constexpr int NV = S / (2 * T);
  static_assert(NV >= 1, "span too small");
  const int nsp = (int)((plane_floats + S - 1) / S);
  const unsigned h = blockIdx.x, q = h & 7u, s = h >> 3;
  const unsigned per = (unsigned)((nsp + 7) >> 3);
  const int ch = (int)(s / per);
  const int sp = (int)(q * per + (s - (unsigned)ch * per));
  if (sp >= nsp) return;
  size_t o[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) {
    size_t i = (size_t)sp * S + 2 * (threadIdx.x + (size_t)v * T);
    if (i > plane_floats - 2) i = plane_floats - 2;
    o[v] = i;
  }
  const int k0 = 1 + ch * zc, k1 = min(k0 + zc, 1 + nz);
  float2 acc = {0.f, 0.f};
  Regs<MIX, NV, NT> r0, r1;
  load_plane<MIX, NV, NT>(r0, A, o, (size_t)k0 * psz);
  if (P == 2) load_plane<MIX, NV, NT>(r1, A, o, (size_t)min(k0 + 1, k1 - 1) * psz);
  for (int k = k0; k < k1; k += P) {
    if (P == 1) {
      Regs<MIX, NV, NT> n;
      load_plane<MIX, NV, NT>(n, A, o, (size_t)min(k + 1, k1 - 1) * psz);
      store_plane<MIX, NV, NT>(r0, A, o, (size_t)k * psz, w, acc);
      r0 = n;
    } else {
      Regs<MIX, NV, NT> n0, n1;
      load_plane<MIX, NV, NT>(n0, A, o, (size_t)min(k + 2, k1 - 1) * psz);
      store_plane<MIX, NV, NT>(r0, A, o, (size_t)k * psz, w, acc);
      load_plane<MIX, NV, NT>(n1, A, o, (size_t)min(k + 3, k1 - 1) * psz);
      if (k + 1 < k1) store_plane<MIX, NV, NT>(r1, A, o, (size_t)(k + 1) * psz, w, acc);
      r0 = n0; r1 = n1;
    }
  }
  if (MIX == MIX_R && acc.x + acc.y == 1.2345f) sink[0] = acc.x;
}

// grid-stride elementwise over the whole array (incl. ghosts), V floats per lane
template <int MIX, typename V>
__global__ void __launch_bounds__(256) k_elem(Arr A, size_t n, float w, float* sink) {
  float accs = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    V a, b, c;
    V d;
    if (MIX == MIX_B || MIX == MIX_R || MIX == MIX_P) { a = ((const V*)A.e)[i]; b = ((const V*)A.r)[i]; c = ((const V*)A.x)[i]; }
    else b = ((const V*)A.r)[i];
    if (MIX == MIX_P) { d = ((const V*)A.ro)[i]; float* pd = (float*)&d; float* qa = (float*)&a; float* qb = (float*)&b; float* qc = (float*)&c;
      for (unsigned q = 0; q < sizeof(V) / 4; q++) { qa[q] -= w * pd[q]; qb[q] -= w * pd[q]; qc[q] -= w * pd[q]; pd[q] *= w; }
      ((V*)A.e)[i] = a; ((V*)A.r)[i] = b; ((V*)A.x)[i] = c; ((V*)A.eo)[i] = d; }
    float* pa = (float*)&a; float* pb = (float*)&b; float* pc = (float*)&c;
    if (MIX == MIX_B) { for (unsigned q = 0; q < sizeof(V) / 4; q++) { pb[q] -= w * pa[q]; pc[q] += w * pa[q]; } ((V*)A.ro)[i] = b; ((V*)A.x)[i] = c; }
    if (MIX == MIX_A) { ((V*)A.ro)[i] = b; for (unsigned q = 0; q < sizeof(V) / 4; q++) pb[q] *= w; ((V*)A.eo)[i] = b; }
    if (MIX == MIX_C) ((V*)A.ro)[i] = b;
    if (MIX == MIX_R) for (unsigned q = 0; q < sizeof(V) / 4; q++) accs += pa[q] + pb[q] + pc[q];
  }
  if (MIX == MIX_R && accs == 1.2345f) sink[0] = accs;
}


// locality test: every block moves one 2-KB segment per stream and iteration (float2 per lane); segment index = scatter(linear index):
// groups of G consecutive segments stay together, the groups are permuted pseudo-randomly over the array (odd multiplier mod 2^m).
template <int MIX>
__global__ void __launch_bounds__(256) k_scatter(Arr A, unsigned nseg_log2, unsigned glog2, float w, float* sink) {
  const unsigned nseg = 1u << nseg_log2, ngrp_mask = (nseg >> glog2) - 1u, gmask = (1u << glog2) - 1u;
  float accs = 0.f;
  for (unsigned i = blockIdx.x; i < nseg; i += gridDim.x) {
    const unsigned grp = ((i >> glog2) * 2654435761u) & ngrp_mask;
    const size_t o = ((size_t)((grp << glog2) | (i & gmask)) * 256 + threadIdx.x) * 2;
    float2 a = {0, 0}, b, c = {0, 0};
    if (MIX == MIX_B || MIX == MIX_R || MIX == MIX_P) { a = *(const float2*)(A.e + o); b = *(const float2*)(A.r + o); c = *(const float2*)(A.x + o); }
    else b = *(const float2*)(A.r + o);
    if (MIX == MIX_P) { float2 d = *(const float2*)(A.ro + o); a.x -= w * d.x; a.y -= w * d.y; b.x -= w * d.x; b.y -= w * d.y; c.x -= w * d.x; c.y -= w * d.y; d.x *= w; d.y *= w;
      *(float2*)(A.e + o) = a; *(float2*)(A.r + o) = b; *(float2*)(A.x + o) = c; *(float2*)(A.eo + o) = d; }
    if (MIX == MIX_B) { b.x -= w * a.x; b.y -= w * a.y; c.x += w * a.x; c.y += w * a.y; *(float2*)(A.ro + o) = b; *(float2*)(A.x + o) = c; }
    if (MIX == MIX_A) { *(float2*)(A.ro + o) = b; b.x *= w; b.y *= w; *(float2*)(A.eo + o) = b; }
    if (MIX == MIX_C) *(float2*)(A.ro + o) = b;
    if (MIX == MIX_R) accs += a.x + a.y + b.x + b.y + c.x + c.y;
  }
  if (MIX == MIX_R && accs == 1.2345f) sink[0] = accs;
}

static const int N = 512;
static int NG = 514;            // row pitch = planes' side with ghosts (argv[2]: 512 = no ghost cells, power-of-two strides; 516/520/528: padded rows)
static size_t PSZ, NTOT;
static int OI = 2, OJ = 1, OK = 1;   // first interior cell
static hipEvent_t t0, t1;
static const double mixbytes[5] = {20.0, 12.0, 8.0, 12.0, 32.0};
static const char* mixname[5] = {"B(3R+2W)", "A(1R+2W)", "C(copy)", "R(3R)", "P(4R+4W)"};
struct Row { std::string name; int mix; double ms, tbs; };
static std::vector<Row> rows;

template <typename F> static void timeit(const std::string& name, int mix, F&& launch) {
  for (int q = 0; q < 2; q++) launch();
  float best = 1e30f, sum = 0.f; const int reps = 5;
  for (int rep = 0; rep < reps; rep++) {
    CK(hipEventRecord(t0)); launch(); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1)); best = std::min(best, ms); sum += ms;
  }
  CK(hipGetLastError());
  const double cells = (double)N * N * N;
  const double avg = sum / reps;
  printf("%-10s %-44s avg %7.3f ms  min %7.3f ms  %5.2f TB/s (avg)\n", mixname[mix], name.c_str(), avg, best, mixbytes[mix] * cells / (avg * 1e-3) / 1e12);
  fflush(stdout);
  rows.push_back({name, mix, avg, mixbytes[mix] * cells / (avg * 1e-3) / 1e12});
}

static Arr A; static float* sink;

template <int MIX, int W, int H, int T, int P, int NT> static void run_tile(int chunks) {
  const int ntx = (N + W - 1) / W, nty = (N + H - 1) / H, per = (ntx * nty + 7) >> 3;
  const int zc = (N + chunks - 1) / chunks, nch = (N + zc - 1) / zc;
  char nm[96]; snprintf(nm, 96, "tile %3dx%-2d T%-4d P%d %s ch%-2d (%d wg)", W, H, T, P, NT ? "nt" : "  ", nch, 8 * per * nch);
  timeit(nm, MIX, [&] { k_tile<MIX, W, H, T, P, NT><<<8 * per * nch, T>>>(A, N, N, N, NG, PSZ, zc, 0.5f, sink, OI, OJ, OK); });
}
template <int MIX, int S, int T, int P, int NT> static void run_span(int chunks) {
  const int nsp = (int)((PSZ + S - 1) / S), per = (nsp + 7) >> 3;
  const int zc = (N + chunks - 1) / chunks, nch = (N + zc - 1) / zc;
  char nm[96]; snprintf(nm, 96, "span %5d    T%-4d P%d %s ch%-2d (%d wg)", S, T, P, NT ? "nt" : "  ", nch, 8 * per * nch);
  timeit(nm, MIX, [&] { k_span<MIX, S, T, P, NT><<<8 * per * nch, T>>>(A, N, PSZ, PSZ, zc, 0.5f, sink); });
}
template <int MIX> static void run_elems() {
  for (int grid : {2048, 16384, 65536, 262144}) {
    char nm[96];
    snprintf(nm, 96, "elem  4B/lane grid %d", grid); timeit(nm, MIX, [&] { k_elem<MIX, float><<<grid, 256>>>(A, NTOT, 0.5f, sink); });
    snprintf(nm, 96, "elem  8B/lane grid %d", grid); timeit(nm, MIX, [&] { k_elem<MIX, float2><<<grid, 256>>>(A, NTOT / 2, 0.5f, sink); });
    snprintf(nm, 96, "elem 16B/lane grid %d", grid); timeit(nm, MIX, [&] { k_elem<MIX, float4><<<grid, 256>>>(A, NTOT / 4, 0.5f, sink); });
  }
}
template <int MIX> static void run_shapes() {
  // tile shapes at 1024 threads, one plane in flight
  run_tile<MIX, 64, 32, 1024, 1, 0>(10);
  run_tile<MIX, 128, 16, 1024, 1, 0>(10);
  run_tile<MIX, 256, 8, 1024, 1, 0>(10);
  run_tile<MIX, 512, 4, 1024, 1, 0>(10);
  run_tile<MIX, 128, 32, 1024, 1, 0>(20);
  run_tile<MIX, 256, 16, 1024, 1, 0>(20);
  run_tile<MIX, 512, 8, 1024, 1, 0>(20);
  // 512 and 256 threads
  run_tile<MIX, 64, 16, 512, 1, 0>(10);
  run_tile<MIX, 64, 32, 512, 1, 0>(20);
  run_tile<MIX, 128, 8, 512, 1, 0>(10);
  run_tile<MIX, 256, 4, 512, 1, 0>(10);
  run_tile<MIX, 512, 4, 512, 1, 0>(20);
  run_tile<MIX, 64, 8, 256, 1, 0>(8);
  run_tile<MIX, 64, 32, 256, 1, 0>(32);
  run_tile<MIX, 128, 4, 256, 1, 0>(8);
  run_tile<MIX, 256, 4, 256, 1, 0>(16);
  run_tile<MIX, 512, 4, 256, 1, 0>(32);
  // two planes in flight
  run_tile<MIX, 64, 32, 1024, 2, 0>(10);
  run_tile<MIX, 128, 16, 1024, 2, 0>(10);
  run_tile<MIX, 512, 4, 1024, 2, 0>(10);
  run_tile<MIX, 64, 16, 512, 2, 0>(10);
  run_tile<MIX, 256, 4, 512, 2, 0>(10);
  run_tile<MIX, 64, 8, 256, 2, 0>(8);
  run_tile<MIX, 256, 4, 256, 2, 0>(16);
  // non-temporal hints
  run_tile<MIX, 64, 32, 1024, 1, 1>(10);
  run_tile<MIX, 512, 4, 1024, 1, 1>(10);
  run_tile<MIX, 64, 16, 512, 1, 1>(10);
  // chunk counts for the reference shape and the full-row shape
  for (int ch : {4, 7, 16, 32}) run_tile<MIX, 64, 32, 1024, 1, 0>(ch);
  for (int ch : {4, 7, 16, 32}) run_tile<MIX, 512, 4, 1024, 1, 0>(ch);
  // contiguous spans of the plane (full rows)
  run_span<MIX, 2048, 1024, 1, 0>(10);
  run_span<MIX, 4096, 1024, 1, 0>(16);
  run_span<MIX, 8192, 1024, 1, 0>(32);
  run_span<MIX, 2048, 512, 1, 0>(10);
  run_span<MIX, 1024, 512, 1, 0>(5);
  run_span<MIX, 1024, 256, 1, 0>(5);
  run_span<MIX, 2048, 256, 1, 0>(10);
  run_span<MIX, 512, 256, 1, 0>(3);
  run_span<MIX, 2048, 1024, 2, 0>(10);
  run_span<MIX, 1024, 256, 2, 0>(5);
  run_span<MIX, 4096, 256, 2, 0>(20);
  run_span<MIX, 2048, 1024, 1, 1>(10);
}


template <int MIX> static void run_locality() {
  // 2^27 floats per array (512 MiB) = 2^18 segments of 2 KB
  const double scale = (double)(1u << 27) / ((double)N * N * N);   // timeit() assumes 512^3 cells: these runs move exactly that many (2^27)
  (void)scale;
  for (int grid : {2048, 1 << 18}) for (unsigned gl : {0u, 2u, 4u, 6u, 8u, 10u, 12u, 18u}) {
    char nm[96]; snprintf(nm, 96, "scatter grid %-6d group %6u KB", grid, (1u << gl) * 2);
    timeit(nm, MIX, [&] { k_scatter<MIX><<<grid, 256>>>(A, 18u, gl, 0.5f, sink); });
  }
  // z-march with a small plane stride: the same arrays viewed as 512 x 32 x 8192 cells (plane = 34 rows of 514 floats = 70 KB)
  {
    const int nx = 512, ny = 32, nz = 7600; const size_t psz = (size_t)514 * 34;
    for (int chunks : {80, 160, 320}) {
      const int ntx = nx / 64, nty = 1, per = (ntx * nty + 7) >> 3; const int zc = (nz + chunks - 1) / chunks, nch = (nz + zc - 1) / zc;
      char nm[96]; snprintf(nm, 96, "thin-plane tile 64x32 T1024 zc %d (%d wg) [x%.3f cells]", zc, 8 * per * nch, (double)nx * ny * nz / ((double)N * N * N));
      timeit(nm, MIX, [&] { k_tile<MIX, 64, 32, 1024, 1, 0><<<8 * per * nch, 1024>>>(A, nx, ny, nz, 514, psz, zc, 0.5f, sink); });
    }
  }
}

int main(int argc, char** argv) {
  if (argc > 2) NG = atoi(argv[2]);
  if (NG == N) { OI = 0; OJ = 0; OK = 0; }
  PSZ = (size_t)NG * NG; NTOT = PSZ * NG;
  float *e, *r, *x, *ro, *eo;
  const size_t bytes = NTOT * 4 + 4096;
  // argv[3] (bytes, optional): all five arrays in ONE allocation, array q at q·(round_up(bytes, 2 MiB) + delta) — relative placement under control
  const long delta = argc > 3 ? atol(argv[3]) : -1;
  if (delta >= 0) {
    const size_t sp = ((bytes + (2u << 20) - 1) >> 21 << 21) + (size_t)delta;
    char* base; CK(hipMalloc(&base, 5 * sp + (4u << 20)));
    e = (float*)base; r = (float*)(base + sp); x = (float*)(base + 2 * sp); ro = (float*)(base + 3 * sp); eo = (float*)(base + 4 * sp);
    printf("# one allocation, spacing %zu + %ld bytes\n", sp - (size_t)delta, delta);
  } else if (getenv("PLACE_CONTIG")) {   // physically contiguous allocations (hipDeviceMallocContiguous): a deterministic placement
    for (float** q : {&e, &r, &x, &ro, &eo}) CK(hipExtMallocWithFlags((void**)q, bytes, hipDeviceMallocContiguous));
    printf("# hipDeviceMallocContiguous\n");
  } else { CK(hipMalloc(&e, bytes)); CK(hipMalloc(&r, bytes)); CK(hipMalloc(&x, bytes)); CK(hipMalloc(&ro, bytes)); CK(hipMalloc(&eo, bytes)); }
  CK(hipMalloc(&sink, 64));
  // random-ish data (DVFS: zero-filled inputs clock higher than real data)
  std::vector<float> hbuf(NTOT);
  unsigned s = 12345u; for (size_t i = 0; i < NTOT; i++) { s = s * 1664525u + 1013904223u; hbuf[i] = (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
  for (float* p : {e, r, x, ro, eo}) CK(hipMemcpy(p, hbuf.data(), NTOT * 4, hipMemcpyHostToDevice));
  A = Arr{e, r, x, ro, eo};
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  printf("# arrays %dx%dx%d floats (512^3 cells; pitch = side incl. ghost/pad cells), bytes per cell: B 20, A 12, C 8, R 12; TB/s = bytes/cell x 512^3 / avg time\n", NG, NG, NG);
  printf("# pointers e %p r %p x %p ro %p eo %p\n", (void*)e, (void*)r, (void*)x, (void*)ro, (void*)eo);
  const char* only = argc > 1 ? argv[1] : "BACR";
  for (const char* p = only; *p; p++) {
    if (*p == 'T') {   // the core comparisons, three rounds interleaved (clock states and placement luck show as spread)
      for (int rep = 0; rep < 3; rep++) {
        timeit("elem 4B/lane grid 262144", MIX_B, [&] { k_elem<MIX_B, float><<<262144, 256>>>(A, NTOT, 0.5f, sink); });
        run_tile<MIX_B, 64, 32, 1024, 1, 0>(10); run_tile<MIX_B, 64, 32, 1024, 1, 0>(32); run_tile<MIX_B, 256, 4, 256, 1, 0>(16);
        timeit("elem 8B/lane grid 262144", MIX_A, [&] { k_elem<MIX_A, float2><<<262144, 256>>>(A, NTOT / 2, 0.5f, sink); });
        run_tile<MIX_A, 64, 32, 1024, 1, 0>(10); run_tile<MIX_A, 64, 32, 1024, 1, 0>(32);
      }
    }
    if (*p == 'S') {   // stride study: the reference tile shapes only (run with argv[2] = 512, 514, 516, 520, 528, 544)
      run_tile<MIX_B, 64, 32, 1024, 1, 0>(10); run_tile<MIX_B, 64, 32, 1024, 1, 0>(32); run_tile<MIX_B, 256, 4, 256, 1, 0>(16);
      run_tile<MIX_A, 64, 32, 1024, 1, 0>(10); run_tile<MIX_A, 64, 32, 1024, 1, 0>(32); run_tile<MIX_P, 64, 32, 1024, 1, 0>(10);
    }
    if (*p == 'G') {   // geometry study on ONE set of allocations (same placement state throughout; run with argv[2] = 576 so that the arrays hold every case):
                       // row pitch and the column of the first interior cell — (514,1) is the reference's layout, (514,2) what modes S/T use, the others padded rows
      const int maxng = NG;
      static const int geo[][2] = {{514, 1}, {514, 2}, {528, 16}, {576, 64}, {520, 8}, {544, 32}, {512, 0}, {516, 1}, {518, 1}, {522, 1}, {530, 1}, {514, 1}};
      for (int rep = 0; rep < 2; rep++) for (auto& gq : geo) {
        if (gq[0] > maxng) continue;
        NG = gq[0]; OI = gq[1]; PSZ = (size_t)NG * 514;
        printf("## pitch %d floats (%d B), first interior column %d (byte %d of its row), plane %zu B\n", NG, NG * 4, OI, OI * 4, PSZ * 4);
        run_tile<MIX_B, 64, 32, 1024, 1, 0>(10); run_tile<MIX_A, 64, 32, 1024, 1, 0>(10); run_tile<MIX_P, 64, 32, 1024, 1, 0>(10); run_tile<MIX_R, 64, 16, 512, 1, 0>(8);
      }
      NG = maxng;
    }
    if (*p == 'L') { run_locality<MIX_B>(); run_locality<MIX_R>(); run_locality<MIX_C>(); }
    if (*p == 'P') {
      run_elems<MIX_P>();
      run_span<MIX_P, 512, 256, 1, 0>(16); run_span<MIX_P, 512, 256, 1, 0>(57); run_span<MIX_P, 512, 256, 1, 0>(128); run_span<MIX_P, 512, 256, 1, 0>(512);
      run_span<MIX_P, 1024, 256, 1, 0>(16); run_span<MIX_P, 1024, 256, 1, 0>(128); run_span<MIX_P, 1024, 256, 1, 0>(512);
      run_span<MIX_P, 2048, 256, 1, 0>(16); run_span<MIX_P, 2048, 256, 1, 0>(512); run_span<MIX_P, 2048, 1024, 1, 0>(512);
      run_tile<MIX_P, 64, 32, 1024, 1, 0>(10); run_tile<MIX_P, 512, 4, 1024, 1, 0>(10); run_tile<MIX_P, 256, 4, 256, 1, 0>(16);
      run_locality<MIX_P>();
    }
    if (*p == 'B') { run_elems<MIX_B>(); run_shapes<MIX_B>(); }
    if (*p == 'A') { run_elems<MIX_A>(); run_shapes<MIX_A>(); }
    if (*p == 'C') { run_elems<MIX_C>(); run_shapes<MIX_C>(); }
    if (*p == 'R') { run_elems<MIX_R>(); run_shapes<MIX_R>(); }
  }
  // best per mix
  for (int m = 0; m < 5; m++) {
    const Row* b = nullptr; for (auto& rw : rows) if (rw.mix == m && (!b || rw.tbs > b->tbs)) b = &rw;
    if (b) printf("# best %-10s %-44s %5.2f TB/s\n", mixname[m], b->name.c_str(), b->tbs);
  }
  CK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
