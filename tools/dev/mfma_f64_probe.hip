// Probe: layout and rounding order of v_mfma_f64_16x16x4_f64 on gfx950 (development aid).
// Prints the operand/result layout (one-hot inputs) and checks, bit for bit, which chain of
// fused multiply-adds the instruction evaluates for D = A*B + C.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdint>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const double *a, const double *b, const double *c, double *d) {
  const int l = threadIdx.x;
  double4_t cc = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
  double4_t r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], cc, 0, 0, 0);
  for (int v = 0; v < 4; v++) d[l * 4 + v] = r[v];
}

int main() {
  double ha[64], hb[64], hc[256], hd[256], *da, *db, *dc, *dd;
  (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dc, 2048); (void)hipMalloc(&dd, 2048);
  auto run = [&]() {
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, hc, 2048, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(da, db, dc, dd);
    (void)hipMemcpy(hd, dd, 2048, hipMemcpyDeviceToHost);
  };
  // ---- layout: assume A[i][k] in lane i + 16k, B[k][j] in lane j + 16k; find D[i][j] position
  // A = one-hot rows: A[i][0] = i+1, B[0][j] = 100*(j+1) -> D[i][j] = (i+1)*100*(j+1)
  for (int l = 0; l < 64; l++) { ha[l] = (l / 16 == 0) ? (l % 16) + 1 : 0; hb[l] = (l / 16 == 0) ? 100.0 * ((l % 16) + 1) : 0; }
  memset(hc, 0, sizeof hc);
  run();
  int layout_ok = 1;
  for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
    int i = (l / 16) + 4 * v, j = l % 16;  // hypothesis
    if (hd[l * 4 + v] != (i + 1) * 100.0 * (j + 1)) layout_ok = 0;
  }
  printf("layout D[i=lane/16+4*v][j=lane%%16], A lane=i+16k, B lane=j+16k: %s\n", layout_ok ? "confirmed" : "NOT as assumed");
  if (!layout_ok) { for (int l = 0; l < 64; l += 5) printf("lane %d: %g %g %g %g\n", l, hd[l*4], hd[l*4+1], hd[l*4+2], hd[l*4+3]); }
  // ---- rounding order on random data
  uint64_t s = 88172645463325252ULL;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 * 2 - 1; };
  long bad_seq = 0, bad_rev = 0, bad_c0 = 0, total = 0;
  for (int t = 0; t < 200; t++) {
    for (int l = 0; l < 64; l++) { ha[l] = rnd() * 1e-4; hb[l] = rnd() * 0.07; }
    for (int q = 0; q < 256; q++) hc[q] = (t & 1) ? 0.0 : rnd() * 1e-5;
    run();
    for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
      int i = (l / 16) + 4 * v, j = l % 16;
      double c0 = hc[l * 4 + v];
      double seq = c0, rev = c0;
      for (int k = 0; k < 4; k++) seq = fma(ha[i + 16 * k], hb[j + 16 * k], seq);
      for (int k = 3; k >= 0; k--) rev = fma(ha[i + 16 * k], hb[j + 16 * k], rev);
      double got = hd[l * 4 + v];
      total++;
      if (memcmp(&got, &seq, 8)) bad_seq++;
      if (memcmp(&got, &rev, 8)) bad_rev++;
    }
  }
  printf("vs fma chain k=0..3 from C: %ld mismatches of %ld; vs chain k=3..0: %ld\n", bad_seq, total, bad_rev);
  // ---- our use: C = 0, k0 = gy*y, k1 = gx*x, k2 = gz*z, k3 = 0*0
  long bad_use = 0; total = 0;
  for (int t = 0; t < 200; t++) {
    for (int l = 0; l < 64; l++) { ha[l] = (l / 16 < 3) ? rnd() * 1e-4 : 0.0; hb[l] = (l / 16 < 3) ? rnd() * 0.07 : 0.0; }
    memset(hc, 0, sizeof hc);
    run();
    for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
      int i = (l / 16) + 4 * v, j = l % 16;
      double want = fma(ha[i + 32], hb[j + 32], fma(ha[i + 16], hb[j + 16], ha[i] * hb[j]));
      double got = hd[l * 4 + v];
      total++;
      if (memcmp(&got, &want, 8)) bad_use++;
    }
  }
  printf("inner = fma(a2,b2, fma(a1,b1, a0*b0)) with C = 0, k3 = 0: %ld mismatches of %ld\n", bad_use, total);
  return 0;
}
