/* The shortcut in finalize_plan_sys / finalize_plan (rust_robotics_amd/csrc/resample_core.hpp): a shard that holds EVERYTHING
 * (base = 0, local total = global total T > 0) serves all n output slots of a systematic resample -- slots_upto(0) = 0 and
 * slots_upto(T) = n whenever the plan's offset is below T, which rr_sys_plan_make guarantees for rho in [0, 1).  The kernel
 * relies on it instead of evaluating rr_sys_slots_upto_exact twice; this program checks the identity against the exact function
 * on the host (same header, same code) over edge cases and a few million random plans.  Exit code 0 = holds everywhere. */
#include <stdint.h>
#include <stdio.h>

#define RR_HD static inline
#include "rr_pf_spec.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t next(void) { /* splitmix64 */
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static int check(double rho, uint64_t total, uint64_t n) {
  rr_sys_plan p = rr_sys_plan_make(rho, total, n);
  if (!(p.offs < total)) {
    printf("offs >= T: rho %.17g T %llu n %llu offs %llu\n", rho, (unsigned long long)total, (unsigned long long)n, (unsigned long long)p.offs);
    return 1;
  }
  uint64_t first = rr_sys_slots_upto_exact(p, total, 0), end = rr_sys_slots_upto_exact(p, total, total);
  if (first != 0 || end != n) {
    printf("rho %.17g T %llu n %llu: served [%llu, %llu), expected [0, %llu)\n", rho, (unsigned long long)total, (unsigned long long)n,
           (unsigned long long)first, (unsigned long long)end, (unsigned long long)n);
    return 1;
  }
  return 0;
}

int main(void) {
  const double rhos[] = {0.0, 0x1p-53, 0x1p-30, 0.25, 0.5, 0.75, 1.0 - 0x1p-53};
  const uint64_t totals[] = {1, 2, 3, 1000, (1ull << 32) - 1, 1ull << 32, (1ull << 52) + 12345, (1ull << 62) + 7, ~0ull >> 1, ~0ull};
  const uint64_t ns[] = {1, 2, 3, 100, 1000, 1000000, 16000000, (1ull << 31) - 1};
  int bad = 0;
  for (unsigned a = 0; a < sizeof rhos / sizeof *rhos; ++a)
    for (unsigned b = 0; b < sizeof totals / sizeof *totals; ++b)
      for (unsigned c = 0; c < sizeof ns / sizeof *ns; ++c) bad += check(rhos[a], totals[b], ns[c]);
  for (int k = 0; k < 3000000 && !bad; ++k) {
    const double rho = (double)(next() >> 11) * 0x1p-53;
    uint64_t total = next() >> (next() & 63);
    if (total == 0) total = 1;
    uint64_t n = (next() >> 33) >> (next() % 31);
    if (n == 0) n = 1;
    bad += check(rho, total, n);
  }
  if (bad) return 1;
  printf("SERVED_RANGE_OK\n");
  return 0;
}
