#include <hip/hip_runtime.h>
#include <cstdio>
__device__ int wave_inclusive_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}
__global__ void k(const int *in, int *out) { out[threadIdx.x] = wave_inclusive_sum(in[threadIdx.x]); }
int main() {
  int h[64], o[64], *di, *dq; int bad = 0;
  hipMalloc(&di, 256); hipMalloc(&dq, 256);
  for (int t = 0; t < 50; t++) {
    for (int i = 0; i < 64; i++) h[i] = (i * 7919 + t * 104729) % 97;
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dq);
    hipMemcpy(o, dq, 256, hipMemcpyDeviceToHost);
    int acc = 0;
    for (int i = 0; i < 64; i++) { acc += h[i]; if (o[i] != acc) bad++; }
  }
  printf("scan mismatches: %d\n", bad);
  return bad != 0;
}
