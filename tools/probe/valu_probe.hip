// Issue cost of the vector instructions conv_diff!'s flux arithmetic is made of, relative to v_fma_f32 (4 cycles per wave64 on a SIMD16).
// One workgroup per CU-slot, W waves per SIMD; each wave runs ITER iterations of 16 independent copies of one instruction (inline asm, so
// the compiler cannot fuse or drop them).  Reported: ns per instruction per wave-slot = time / (ITER*16*waves_per_simd), and the ratio to v_fma_f32.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/valu_probe.hip -o tools/probe/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 4096
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int OP> __global__ void __launch_bounds__(1024) k(float* out, float s, unsigned long long mk) {
  unsigned long long msk = mk; unsigned long long mm[4] = {0, 0, 0, 0};
  f2 a[16]; float b = s, c = s * 0.5f; f2 bb = {s, s * 2}, cc = {s * 3, s * 0.25f};
  double d[16];
  for (int i = 0; i < 16; i++) { a[i] = f2{(float)i + threadIdx.x, (float)i * 0.5f}; d[i] = i; }
  for (int it = 0; it < ITER; it++) {
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b), "v"(c));
#define MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(bb), "v"(cc));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(bb));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(bb));
#define MED3(i) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b), "v"(c));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i].x) : "v"(b) : );
#define CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i].x), "v"(b) : "vcc");
#define CVT64(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i].x));
#define MUL64(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
#define CVT32(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i].x) : "v"(d[i]));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].y) : "v"(a[i].x));
#define DPPR(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].y) : "v"(a[i].x));
#define ADDDPP(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].y) : "v"(a[i].x));
#define PKMOV(i) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(a[i]) : "v"(bb), "v"(cc));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].x) : "v"(b));
#define ADD(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b));
#define MAXF(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b));
#define MINF(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b));
#define MAX3(i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b), "v"(c));
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(a[i].x) : "v"(b), "s"(msk));
#define CMPS(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mm[i & 3]) : "v"(a[i].x), "v"(b));
#define CMPCND(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2\n v_cndmask_b32_e64 %1, %3, %1, %0" : "=&s"(mm[i & 3]), "+v"(a[i].x) : "v"(b), "v"(c));
#define ANDB(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i].x) : "v"(b));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b), "v"(c));
#define ASHR(i) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[i].x));
#define ADDU(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i].x) : "v"(b));
#define ADD64(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
#define CNDVCCW(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %0, vcc" : "+v"(a[i].x) : "v"(a[i].y), "v"(b), "v"(c) : "vcc");
#define SUBREV(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b));
    if (OP == 0) { REP16(FMA) }
    if (OP == 1) { REP16(MUL) }
    if (OP == 2) { REP16(PKFMA) }
    if (OP == 3) { REP16(PKMUL) }
    if (OP == 4) { REP16(PKADD) }
    if (OP == 5) { REP16(MED3) }
    if (OP == 6) { REP16(CNDM) }
    if (OP == 7) { REP16(CMP) }
    if (OP == 8) { REP16(CVT64) }
    if (OP == 9) { REP16(MUL64) }
    if (OP == 10) { REP16(CVT32) }
    if (OP == 11) { REP16(DPP) }
    if (OP == 12) { REP16(DPPR) }
    if (OP == 13) { REP16(ADDDPP) }
    if (OP == 14) { REP16(PKMOV) }
    if (OP == 15) { REP16(MOV) }
    if (OP == 17) { REP16(ADD) }
    if (OP == 18) { REP16(MAXF) }
    if (OP == 19) { REP16(MINF) }
    if (OP == 20) { REP16(MAX3) }
    if (OP == 21) { REP16(CNDS) }
    if (OP == 22) { REP16(CMPS) }
    if (OP == 23) { REP16(CMPCND) }
    if (OP == 24) { REP16(ANDB) }
    if (OP == 25) { REP16(BFI) }
    if (OP == 26) { REP16(ASHR) }
    if (OP == 27) { REP16(ADDU) }
    if (OP == 28) { REP16(ADD64) }
    if (OP == 29) { REP16(CNDVCCW) }
    if (OP == 16) { REP16(FMA) REP16(PKMUL) }   // mixed stream: do packed and plain ops alternate freely?
  }
  float r = 0; for (int i = 0; i < 16; i++) r += a[i].x + a[i].y + (float)d[i];
  if (r == 12345.678f) out[0] = r + (float)(mm[0] + mm[1] + mm[2] + mm[3]);
}
static const char* names[] = {"v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_med3_f32", "v_cndmask_b32", "v_cmp_lt_f32", "v_cvt_f64_f32", "v_mul_f64",
                              "v_cvt_f32_f64", "v_mov_dpp wave_shr:1", "v_mov_dpp row_shr:1", "v_add_f32_dpp row_shr:1", "v_pk_mov_b32", "v_mov_b32", "16 fma + 16 pk_mul", "v_add_f32", "v_max_f32", "v_min_f32", "v_max3_f32", "v_cndmask_b32_e64 sgpr", "v_cmp_lt_f32_e64 sgpr", "cmp_e64 + cndmask_e64 (pair)", "v_and_b32", "v_bfi_b32", "v_ashrrev_i32", "v_add_u32", "v_add_f64", "cmp vcc + cndmask vcc (pair)"};
template <int OP> float run(int wps, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 256 * wps;       // wps waves on each of the CU's 4 SIMDs
  k<OP><<<256, threads>>>(out, 1.0f, 0x5555aaaa3333ccccull); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<256, threads>>>(out, 1.0f, 0x5555aaaa3333ccccull); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 64);
  for (int wps = 1; wps <= 4; wps *= 2) {
    float t[30];
    t[0] = run<0>(wps, out); t[1] = run<1>(wps, out); t[2] = run<2>(wps, out); t[3] = run<3>(wps, out); t[4] = run<4>(wps, out); t[5] = run<5>(wps, out);
    t[6] = run<6>(wps, out); t[7] = run<7>(wps, out); t[8] = run<8>(wps, out); t[9] = run<9>(wps, out); t[10] = run<10>(wps, out); t[11] = run<11>(wps, out);
    t[12] = run<12>(wps, out); t[13] = run<13>(wps, out); t[14] = run<14>(wps, out); t[15] = run<15>(wps, out); t[16] = run<16>(wps, out); t[17] = run<17>(wps, out); t[18] = run<18>(wps, out); t[19] = run<19>(wps, out); t[20] = run<20>(wps, out); t[21] = run<21>(wps, out); t[22] = run<22>(wps, out); t[23] = run<23>(wps, out); t[24] = run<24>(wps, out); t[25] = run<25>(wps, out); t[26] = run<26>(wps, out); t[27] = run<27>(wps, out); t[28] = run<28>(wps, out); t[29] = run<29>(wps, out);
    printf("waves per SIMD %d\n", wps);
    for (int i = 0; i < 30; i++) {
      const double n = ((i == 16 || i == 23 || i == 29) ? 32.0 : 16.0) * ITER * wps;
      printf("  %-26s %8.3f ms  %6.3f ns/instr/SIMD  = %5.2f x v_fma_f32 (%.1f cycles if fma = 4)\n", names[i], t[i], t[i] * 1e6 / n, (t[i] / n) / (t[0] / (16.0 * ITER * wps)), 4.0 * (t[i] / n) / (t[0] / (16.0 * ITER * wps)));
    }
  }
  return 0;
}
