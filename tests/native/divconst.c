/* Host check of div_c (pytsc_amd/csrc/tsc_kernels.hip): the five-operation division by a constant whose correctly rounded
 * reciprocal is known, against the IEEE division, bit for bit.  Compiled by tests/test_divconst.py with
 * gcc -O2 -ffp-contract=off; fma() is libm's correctly rounded one. */
#include <math.h>
#include <stdint.h>
#include <string.h>

static double div_c(double a, double b, double rb) {
  double q = a * rb;
  double r = fma(-b, q, a);
  q = fma(r, rb, q);
  r = fma(-b, q, a);
  return fma(r, rb, q);
}

static uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

/* number of inputs (of n) on which div_c(a, b, RN(1 / b)) differs from a / b; *first = the first such input */
long divconst_mismatches(const double *a, long n, double b, double *first) {
  volatile double rbv = 1.0 / b;
  const double rb = rbv;
  long bad = 0;
  for (long i = 0; i < n; i++) {
    volatile double ref = a[i] / b;
    const double got = div_c(a[i], b, rb);
    if (bits(got) != bits((double)ref)) {
      if (!bad && first) *first = a[i];
      bad++;
    }
  }
  return bad;
}
