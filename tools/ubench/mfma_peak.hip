// Micro-benchmark: SUSTAINED matrix-pipe rate of gfx950 for the two instructions the kernels use, on register-resident
// operands (no memory traffic): v_mfma_f32_32x32x2_f32 (64 cycles) and v_mfma_f32_32x32x16_bf16 (32 cycles).
// Reports wall TFLOP/s, shader cycles per MFMA (s_memtime) and the effective shader clock = cycles / wall time,
// for 1 and 2 waves per SIMD and for constant vs random operand data (data toggling changes the power draw).
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool BF16>
__global__ void __launch_bounds__(256) k(const u32x4* in, float* out, long long* cycles, int iters) {
  u32x4 a = in[threadIdx.x], b = in[threadIdx.x + 256];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (BF16)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        else
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a[u]), __builtin_bit_cast(float, b[i]), acc[i], 0, 0, 0);
      }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <bool BF16> void run(const char* what, const u32x4* in, float* out, long long* cyc, int wgs, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<BF16><<<wgs, 256>>>(in, out, cyc, iters / 8); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<BF16><<<wgs, 256>>>(in, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mfmas = 16.0 * iters;
  const double flops = (double)wgs * 4 * mfmas * 2.0 * 32 * 32 * (BF16 ? 16 : 2);
  printf("%-34s wgs=%4d  %8.3f ms  %8.1f TFLOP/s  %6.2f cyc/MFMA(wave)  wave-0 clock %.0f MHz\n", what, wgs, ms, flops / ms / 1e9,
         (double)c / mfmas, (double)c / (ms * 1e3));
}

int main() {
  u32x4* in; float* out; long long* cyc;
  hipMalloc(&in, 512 * 16); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  unsigned host[2048];
  for (int pass = 0; pass < 2; ++pass) {
    // bf16 pairs / floats of magnitude ~1: constant pattern, then random mantissas and signs
    for (int i = 0; i < 2048; ++i) host[i] = pass == 0 ? 0x3f803f80u : (0x3f003f00u | (rand() & 0x80ff80ffu) | ((rand() & 0x7f) << 16));
    hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice);
    const char* d = pass == 0 ? "const" : "random";
    char name[64];
    for (int wgs : {256, 512}) {
      snprintf(name, 64, "fp32 32x32x2  %s data", d);  run<false>(name, in, out, cyc, wgs, 40000);
      snprintf(name, 64, "bf16 32x32x16 %s data", d);  run<true>(name, in, out, cyc, wgs, 80000);
    }
  }
  return 0;
}
