// gmx_exp_fixed(d, shift) (csrc/gmx_math.h: the integer form of the CDF's fixed-point weight) against its definition
// floor(gmx_expf(d) * 2^shift) (0 if NaN / negative / >= 2^63), over the float bit patterns of d:
//   gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math [-DSTRIDE=k] -I genjax_amd/csrc -I include tools/check_exp_fixed.c -lm
// STRIDE = 1 (default) visits all 2^32 patterns for 8 shifts (~10 core-hours... minutes on 8 cores: ~11 min); the CPU
// test suite runs it with a stride.  Exit status 0 = no mismatch.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
#define GMX_HD static inline
#include "gmx_math.h"
static uint64_t ref_def(float d, int shift) {
  float q = gmx_expf(d) * gmx_pow2i(shift);
  return (q >= 0.0f && q < 0x1p63f) ? (uint64_t)q : 0ull;
}
static uint64_t ref_nonpos(float d, int shift) {
  float q = gmx_expf_nonpos(d) * gmx_pow2i(shift);
  if (!(q >= 0.0f) || !(q < 0x1p63f)) return 0ull;
  return (uint64_t)q;
}
int main(int argc, char** argv) {
  int shifts[] = {1, 2, 23, 24, 40, 42, 61, 62};
#ifndef STRIDE
#define STRIDE 1
#endif
  long bad = 0, bad2 = 0;
  for (int si = 0; si < 8; ++si) {
    int shift = shifts[si];
#pragma omp parallel for reduction(+:bad,bad2) schedule(static)
    for (int64_t b = 0; b < (1ll << 32); b += STRIDE) {
      uint32_t u = (uint32_t)b; float d; memcpy(&d, &u, 4);
      uint64_t a = gmx_exp_fixed(d, shift), r = ref_def(d, shift);
      if (a != r) { if (bad < 5) printf("MISMATCH def shift %d d=%a (%08x): %llu vs %llu\n", shift, d, u, (unsigned long long)a, (unsigned long long)r); ++bad; }
      if (d <= 88.0f || d != d) { uint64_t r2 = ref_nonpos(d, shift); if (a != r2) ++bad2; }
    }
    printf("shift %d done, mismatches so far %ld (vs nonpos form %ld)\n", shift, bad, bad2); fflush(stdout);
  }
  return bad != 0;
}
