// CPU harness for csrc/cert_setup.h: reads R (9), N, then src, dst (3N each, xyz interleaved), theta (N),
// noise_bound, cbar2 from stdin; prints mu and the M_init blocks.  (tests/test_certifier_setup.py)
#include <cstdio>
#include <vector>

#include "cert_setup.h"

int main() {
  double R[9], nb, cbar2;
  int N;
  for (double& v : R)
    if (scanf("%lf", &v) != 1) return 1;
  if (scanf("%d", &N) != 1) return 1;
  std::vector<double> src(3 * N), dst(3 * N), th(N);
  for (double& v : src)
    if (scanf("%lf", &v) != 1) return 1;
  for (double& v : dst)
    if (scanf("%lf", &v) != 1) return 1;
  for (double& v : th)
    if (scanf("%lf", &v) != 1) return 1;
  if (scanf("%lf %lf", &nb, &cbar2) != 2) return 1;
  std::vector<double> thp, diag, row0, col0;
  double mu;
  thip::cert_setup(R, src.data(), dst.data(), th.data(), N, nb, cbar2, &thp, &diag, &row0, &col0, &mu);
  printf("%.17g\n", mu);
  for (double v : diag) printf("%.17g\n", v);
  for (double v : row0) printf("%.17g\n", v);
  for (double v : col0) printf("%.17g\n", v);
  return 0;
}
