#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float f_of(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t b_of(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
int main() {
  const float c = 0x1.555556p-3f;
  unsigned long long bad = 0, badnorm = 0; uint32_t first = 0, maxbad_exp = 0, minbad_exp = 255;
#pragma omp parallel for reduction(+:bad,badnorm) schedule(static)
  for (long long i = 0; i < (1LL << 32); i++) {
    uint32_t b = (uint32_t)i;
    uint32_t e = (b >> 23) & 255;
    if (e == 255) continue;
    float x = f_of(b);
    float ref = x / 6.0f;
    float q = x * c;
    float r = fmaf(-6.f, q, x);
    float q1 = fmaf(r, c, q);
    if (b_of(q1) != b_of(ref)) {
      bad++;
      if (e >= 30) badnorm++;
#pragma omp critical
      { if (e > maxbad_exp) maxbad_exp = e; if (e < minbad_exp) minbad_exp = e; if (!first) first = b; }
    }
  }
  printf("mismatches %llu (with biased exponent >= 30: %llu) exp range [%u,%u] first 0x%08x\n", bad, badnorm, minbad_exp, maxbad_exp, first);
  return 0;
}
