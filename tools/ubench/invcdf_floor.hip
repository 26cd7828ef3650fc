// Instruction floor of the BIT-EXACT inverse CDF + merge (R6 + R7) in the wave-per-ray layout of csrc/aon_render.hip.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I articulated-object-nerf_amd/csrc tools/ubench/invcdf_floor.hip -o /tmp/invcdf_floor
//   /tmp/invcdf_floor            (gpurun -- 'bash tools/ubench/run_invcdf_floor.sh')
//
// Every kernel gives one wavefront to a ray, exactly like composite_kernel<., FUSE_PDF>, takes its operands from REGISTERS (they
// are synthesised from the ray index: no HBM traffic besides the 128 shared draws and 4 bytes of result per ray) and runs the
// product's own device functions (csrc/aon_ray_core.h) up to a stage:
//   stage 0  nothing (launch, operand synthesis, the 4-byte result)
//   stage 1  + weights.sum(-1) in ATen's association (torch_sum63)
//   stage 2  + padding, pdf = w / sum, the 64-entry CDF as torch.cumsum computes it (double running sum) -> LDS   [invcdf_build]
//   stage 3  + the two draws of a lane: search, gathers, the reference's interpolation with its IEEE division    [invcdf_draw]
//   stage 4  + the sorted union of 65 coarse t and 128 draws, rank merge with its checks                           [invcdf_merge]
// Differences of consecutive stages price each step at a frame's 307,200 rays.  The fused coarse-level kernel moves 2,104 B per
// ray; 50 % of the 8 TB/s HBM peak is 161 us per frame for EVERYTHING, of which the compositing of the 65 samples alone takes
// 114 us (composite_kernel<true,false,65>, profiles/r02_ray_kernels.jsonl).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "aon_ray_core.h"

using namespace aon;

template <int STAGE>
__global__ void __launch_bounds__(256) floor_kernel(const float* __restrict__ u, float* __restrict__ out, float* __restrict__ tf, int64_t n_rays) {
  __shared__ float lds[4][kPdfLdsFloats];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
  if (ray >= n_rays) return;
  float* L = lds[wv];
  // operands of a ray, from its index: coarse t on the reference's grid (near 2, far 6), bins = mids, weights = a bump whose
  // position and width depend on the ray (non-negative, most of them tiny: what compositing produces)
  const float tc = 2.0f + 4.0f * (float)lane / 64.0f, tn = 2.0f + 4.0f * (float)(lane + 1) / 64.0f, t64 = 6.0f;
  const float b = 0.5f * (tc + tn);
  const float c = 8.0f + (float)(ray % 47), s = 1.5f + (float)(ray % 5);
  const float d = ((float)lane - c) / s;
  float w = lane < 63 ? __expf(-d * d) * 0.3f + 1e-7f : 0.f;
  float acc = w;
  if constexpr (STAGE == 1) acc += torch_sum63(w, lane);
  if constexpr (STAGE >= 2) {
    invcdf_build(L, lane, tc, t64, b, w, STAGE >= 4);
    if constexpr (STAGE == 2) acc += L[kPdfCdf + (lane ^ 1)];
  }
  if constexpr (STAGE >= 3) {
    float smp[2];
    int guess[2];
    invcdf_draw(L, lane, u, nullptr, smp, guess);
    if constexpr (STAGE == 3) acc += smp[0] + smp[1] + (float)guess[0] + (float)guess[1];
    if constexpr (STAGE >= 4) invcdf_merge(L, lane, tc, t64, smp, guess, tf + ray * 193);   // writes 772 B per ray, as the product does
  }
  acc += __shfl_xor(acc, 1);
  if (lane == 0) out[ray] = acc;
}

template <int STAGE>
static float run(const float* u, float* out, float* tf, int64_t n, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const dim3 grid((unsigned)((n + 3) / 4));
  floor_kernel<STAGE><<<grid, 256>>>(u, out, tf, n);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(a);
    floor_kernel<STAGE><<<grid, 256>>>(u, out, tf, n);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  return best * 1e3f;
}

int main() {
  const int64_t n = 640 * 480;
  std::vector<float> u(128);
  for (int j = 0; j < 128; ++j) u[j] = (float)((double)j * (1.0 - 0x1p-32) / 127.0);   // close enough to torch.linspace for timing
  float *du, *dout, *dtf;
  (void)hipMalloc(&du, 128 * 4);
  (void)hipMalloc(&dout, n * 4);
  (void)hipMalloc(&dtf, n * 193 * 4);
  (void)hipMemcpy(du, u.data(), 128 * 4, hipMemcpyHostToDevice);
  const float t0 = run<0>(du, dout, dtf, n, 20), t1 = run<1>(du, dout, dtf, n, 20), t2 = run<2>(du, dout, dtf, n, 20),
              t3 = run<3>(du, dout, dtf, n, 20), t4 = run<4>(du, dout, dtf, n, 20);
  printf("{\"rays\": %lld, \"us_stage0_empty\": %.1f, \"us_stage1_sum63\": %.1f, \"us_stage2_cdf\": %.1f, \"us_stage3_draws\": %.1f, "
         "\"us_stage4_merge\": %.1f, \"us_sum63\": %.1f, \"us_normalise_cumsum\": %.1f, \"us_draws\": %.1f, \"us_merge_incl_772B_store\": %.1f, "
         "\"us_inverse_cdf_total\": %.1f, \"budget_us_at_half_hbm_peak\": %.1f, \"compositing_alone_us\": 114.0}\n",
         (long long)n, t0, t1, t2, t3, t4, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, 2104.0 * n / 4e12 * 1e6);
  return 0;
}
