// Scratch probe: cost of the host-side field / curve arithmetic that sits on the proving thread's critical path (encodes of the
// few-term commitments, challenge inversions), per compiler:  <cxx> -O2 -std=c++17 -Ispartan_amd/csrc bench/host_arith_probe.cc -o /tmp/hap
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstring>
#include "curve.hpp"
using namespace sp;
template <class F> static double ns_per(int n, F f) {
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) f();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / n * 1e9;
}
int main() {
  Fq a = fq_from_u64(123456789), b = fq_from_u64(987654321);
  printf("fq_mul       %8.1f ns\n", ns_per(5000000, [&] { a = fq_mul(a, b); }));
  printf("fq_invert    %8.1f ns\n", ns_per(20000, [&] { a = fq_invert(a); }));
  Fp x = fp_one(); x.v[0] = 0x1234567; Fp y = x; y.v[1] = 77;
  printf("fp_mul       %8.1f ns\n", ns_per(5000000, [&] { x = fp_mul(x, y); }));
  printf("fp_sqr       %8.1f ns\n", ns_per(5000000, [&] { x = fp_sqr(x); }));
  Pt p = pt_identity();
  Niels nn; nn.yp = y; nn.ym = x; nn.t2d = fp_mul(x, y);
  printf("pt_madd      %8.1f ns\n", ns_per(1000000, [&] { p = pt_madd(p, nn, false); }));
  Pt q = p;
  printf("pt_add       %8.1f ns\n", ns_per(1000000, [&] { q = pt_add(q, p); }));
  uint8_t enc[32];
  unsigned acc = 0;
  printf("pt_compress  %8.1f ns\n", ns_per(100000, [&] { pt_compress(q, enc); acc += enc[0]; q.X.v[0] ^= enc[1]; }));
  Pt qq[4] = {q, p, pt_add(q, p), pt_add(p, p)};
  uint8_t enc4[128];
  for (size_t n = 2; n <= 4; n++)
    printf("pt_compress_many(%zu) %8.1f ns\n", n, ns_per(100000, [&] { pt_compress_many(qq, n, enc4); acc += enc4[0]; qq[0].X.v[0] ^= enc4[1]; }));
  printf("(%llx %llx %u)\n", (unsigned long long)a.l[0], (unsigned long long)x.v[0], acc);
}
