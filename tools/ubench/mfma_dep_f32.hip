// Micro-benchmark: issue interval of v_mfma_f32_32x32x2_f32 on ONE wave per SIMD when consecutive MFMAs accumulate into
// the same registers (runs of RUN dependent instructions, as the MLP kernels issue them: 4 per A-fragment) versus fully
// independent accumulators.   hipcc --offload-arch=gfx950 -O3 mfma_dep_f32.hip -o mfma_dep_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int RUN>
__global__ void __launch_bounds__(256) k(const float* in, float* out, long long* cycles, int iters) {
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int i = (u / RUN) % 8;  // RUN consecutive MFMAs share an accumulator, then the next accumulator
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
template <int RUN> void run(const float* in, float* out, long long* cyc) {
  const int iters = 4000;
  k<RUN><<<256, 256>>>(in, out, cyc, iters); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<RUN><<<256, 256>>>(in, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("runs of %2d dependent MFMAs: %.2f cycles per MFMA, %.1f TFLOP/s\n", RUN, (double)c / (32.0 * iters),
         256.0 * 4 * 32 * iters * 2.0 * 32 * 32 * 2 / ms / 1e9);
}
int main() {
  float* in; float* out; long long* cyc;
  hipMalloc(&in, 512 * 4); hipMemset(in, 0x3c, 512 * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<1>(in, out, cyc); run<2>(in, out, cyc); run<4>(in, out, cyc); run<32>(in, out, cyc);
  return 0;
}
