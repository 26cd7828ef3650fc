// Micro-benchmark: how long a wave waits at `s_waitcnt vmcnt(0)` for streaming (nt) / default 16-byte-per-lane stores issued `age` cycles
// earlier, while all 256 CUs write at a realistic rate (the training chain writes ~1.3 TB/s: 4 waves x 4 KiB per CU every ~3 us).
//   hipcc --offload-arch=gfx950 -O3 store_ack.hip -o store_ack && ./store_ack
// Per iteration a wave: issues 4 stores of 1 KiB (one 16-byte unit per lane, contiguous per wave, fresh addresses: a plane stream), spins
// `age` cycles on the ALU, then times s_waitcnt vmcnt(0); then spins the rest of the iteration period.  Prints mean / max stall per age.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void spin(long long until) {
  while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(1);
}

template <bool NT>
__global__ void __launch_bounds__(256) k(float* dst, long long stride_floats, int iters, int age, int period, long long* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* base = dst + ((long long)blockIdx.x * 4 + wave) * stride_floats + lane * 4;
  long long sum = 0, mx = 0;
  f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int it = 0; it < iters; ++it) {
    const long long t_start = __builtin_readcyclecounter();
    float* p = base + (long long)it * 4 * 256;   // 4 stores x 256 floats (1 KiB) per iteration per wave
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p + s * 256));
      else *reinterpret_cast<f32x4*>(p + s * 256) = v;
    }
    spin(t_start + age);
    const long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    sum += t1 - t0; mx = (t1 - t0) > mx ? (t1 - t0) : mx;
    spin(t_start + period);
  }
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = sum; out[(blockIdx.x * 4 + wave) * 2 + 1] = mx; }
}

template <bool NT>
void run(float* dst, long long stride, long long* out, int period) {
  const int iters = 2000;
  const int ages[] = {0, 500, 1000, 2000, 4000, 6000, 8000, 12000, 16000};
  for (int age : ages) {
    if (age >= period) continue;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NT><<<256, 256>>>(dst, stride, iters, age, period, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[256 * 4 * 2];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; long long m = 0;
    for (int i = 0; i < 1024; ++i) { s += h[2 * i]; m = h[2 * i + 1] > m ? h[2 * i + 1] : m; }
    printf("%s period %6d cyc  age %6d cyc: mean stall %8.1f cyc, max %7lld cyc, write rate %.2f TB/s, counter %.0f MHz\n",
           NT ? "nt     " : "default", period, age, s / (1024.0 * iters), m, 256.0 * 4 * 4096 * iters / (ms * 1e-3) / 1e12, (double)iters * period / (ms * 1e3));
  }
}

int main() {
  const long long stride = 2000LL * 4 * 256 + 1024;       // floats per wave stream
  float* dst; long long* out;
  hipMalloc(&dst, 1024 * stride * 4); hipMalloc(&out, 1024 * 2 * 8);
  for (int period : {800, 1600, 3200}) { run<true>(dst, stride, out, period); run<false>(dst, stride, out, period); }
  return 0;
}
