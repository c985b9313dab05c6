// CPU harness of tests/test_fixed_point_sum.py: sums doubles through the fixed-point digits of pg_fixed_point.h exactly as the
// kernels and the host do (digits -> int64 limbs -> one rounding) and prints the result bits.  stdin: "<mode> <q> <limbs> <n>"
// then n values (mode d: doubles as hex bit patterns; mode l: int64 decimal, two digits).  stdout: the sum's bits in hex.
#include <cinttypes>
#include <cstdio>
#include <cstring>

#include "../../pinot_amd/csrc/pg_fixed_point.h"

int main() {
  char mode;
  int q, limbs;
  long n;
  if (scanf(" %c %d %d %ld", &mode, &q, &limbs, &n) != 4) return 2;
  int64_t acc[4] = {0, 0, 0, 0};
  for (long i = 0; i < n; i++) {
    if (mode == 'd') {
      uint64_t b;
      if (scanf("%" SCNx64, &b) != 1) return 2;
      double x;
      memcpy(&x, &b, 8);
      for (int j = 0; j < limbs; j++) acc[j] += pg_fx_digit(x, q, j);
    } else {
      int64_t v;
      if (scanf("%" SCNd64, &v) != 1) return 2;
      for (int j = 0; j < limbs; j++) acc[j] += pg_long_digit(v, j);
    }
  }
  const double r = pg_limbs_to_double(acc, limbs, q);
  uint64_t rb;
  memcpy(&rb, &r, 8);
  printf("%016" PRIx64 "\n", rb);
  return 0;
}
