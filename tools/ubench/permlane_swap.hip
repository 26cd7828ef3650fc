// What v_permlane32_swap / v_permlane16_swap do on gfx950, printed lane by lane (tools/ubench: measurement aids, not product):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pls tools/ubench/permlane_swap.hip && /tmp/pls
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int l = threadIdx.x;
  unsigned a = 100 + l, b = 200 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
  for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%d", l, h[v * 64 + l]); printf("\n"); }
  return 0;
}
