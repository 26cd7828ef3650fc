// Micro-benchmark: the 256x256 weight-gradient kernel on row-major planes (plane[row*Np + n]) against the same kernel
// reading pass-major tiles (plane[(n/128)*R*128 + row*128 + n%128], compile with -DAON_EXP_TILEMAJOR), R = 256.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../articulated-object-nerf_amd/csrc wgrad_layout.hip -o wgrad_layout
#include "aon_wgrad.h"
#include <cstdio>
using namespace aon;
int main() {
  const int64_t Np = 790528;  // 4096 rays x 193 samples
  float *A, *B, *partial, *bias;
  hipMalloc(&A, 256 * Np * 4); hipMalloc(&B, 256 * Np * 4);
  hipMemset(A, 0, 256 * Np * 4); hipMemset(B, 0, 256 * Np * 4);
  hipMalloc(&partial, (size_t)256 * 256 * 256 * 4); hipMalloc(&bias, 256 * 256 * 4);
  constexpr int lds = 2 * (256 + 256) * 32 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  WgradArgs a{A, B, Np, (int)(Np / 32), partial, bias};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 8; ++i) wgrad_kernel<2, 8><<<256, 256, lds>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("wgrad<2,8> Np=%lld: %.3f ms per launch, %.1f TFLOP/s, %.2f TB/s operand reads\n", (long long)Np, ms / 8,
           2.0 * 256 * 256 * Np / (ms / 8) / 1e9, 512.0 * Np * 4 / (ms / 8) / 1e9);
  }
  hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16x3_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 8; ++i) wgrad_bf16x3_kernel<2, 8><<<256, 256, lds>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("wgrad_bf16x3<2,8> Np=%lld: %.3f ms per launch, %.1f TFLOP/s algorithmic, %.2f TB/s operand reads\n", (long long)Np, ms / 8,
           2.0 * 256 * 256 * Np / (ms / 8) / 1e9, 512.0 * Np * 4 / (ms / 8) / 1e9);
  }
  return 0;
}
