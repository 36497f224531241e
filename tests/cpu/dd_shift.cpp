// CPU harness for pythonic-disort_amd/csrc/rtd_dd.h (the header is host + device): reads "n t c0 ... c(n-1)" lines from stdin,
// prints the Taylor-shifted coefficients with 17 significant digits.  Used by tests/test_host_logic.py.
#include <cstdio>
#include <vector>

#include "../../pythonic-disort_amd/csrc/rtd_dd.h"

int main() {
  int n;
  double t;
  while (std::scanf("%d %lf", &n, &t) == 2) {
    std::vector<double> c(n);
    for (int i = 0; i < n; ++i)
      if (std::scanf("%lf", &c[i]) != 1) return 1;
    rtd_taylor_shift(c.data(), n, t);
    for (int i = 0; i < n; ++i) std::printf("%.17g%c", c[i], i + 1 < n ? ' ' : '\n');
  }
  return 0;
}
