// Micro-benchmark: issue interval of v_mfma_f32_32x32x16_bf16 on ONE wave per SIMD with 1, 2 or 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k(const u32x4* in, float* out, long long* cycles) {
  u32x4 a = in[threadIdx.x], b = in[threadIdx.x + 256];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
template <int NACC> void run(const u32x4* in, float* out, long long* cyc) {
  k<NACC><<<256, 256>>>(in, out, cyc); hipDeviceSynchronize();
  k<NACC><<<256, 256>>>(in, out, cyc); hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("NACC=%d: %.2f clock-counter ticks per MFMA (%lld ticks / %d MFMAs)\n", NACC, (double)c / (256 * 8 * NACC), c, 256 * 8 * NACC);
}
int main() {
  u32x4* in; float* out; long long* cyc;
  hipMalloc(&in, 512 * 16); hipMemset(in, 0x3c, 512 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<1>(in, out, cyc); run<2>(in, out, cyc); run<4>(in, out, cyc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int n : {1, 2}) {
    hipEventRecord(e0);
    if (n == 1) k<1><<<1024, 256>>>(in, out, cyc); else k<2><<<1024, 256>>>(in, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 1024.0 * 4 * 256 * 8 * n * 2.0 * 32 * 32 * 16;
    printf("NACC=%d wall: %.3f ms -> %.0f TFLOP/s bf16\n", n, ms, flops / ms / 1e9);
  }
  return 0;
}
