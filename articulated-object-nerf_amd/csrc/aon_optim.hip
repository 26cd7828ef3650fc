// The end of a training step on ONE flat parameter arena (round 6): Adam over the arena in a single launch, and the code library's
// three lookups / their dense table gradients in one launch each.  The reference: torch.optim.Adam(lr, betas=(0.9, 0.999)) in
// configure_optimizers (models/vanilla_nerf/model.py:386-389, model_autodecoder.py:604-606) and CodeLibraryArticulated.forward
// (models/code_library.py:36-53, three nn.Embedding lookups whose autograd produces dense table gradients).
//
// Why: torch's fused Adam walks 83 tensors as three multi_tensor_apply launches of ~44 us each (0.34 TB/s over 45 MB), the embedding
// backward is a fill + a scatter per table; with parameters, gradients and both moments as views into four flat buffers the whole update
// is one grid-stride pass at HBM rate (profiles/r06_step_timeline.txt).
#include "aon_common.h"

#include <cmath>

namespace aon {

// Element-wise Adam, written as the operations of torch's single-tensor CPU implementation (torch/optim/adam.py:_single_tensor_adam,
// the form the parity tests' oracle runs): fp32 throughout, one rounding per operation (-ffp-contract=off, FMAs would be explicit):
//   m  = m + w1 * (g - m)                      exp_avg.lerp_(grad, 1 - beta1)          (weight < 0.5 branch of lerp)
//   v  = v * beta2 + (w2 * g) * g              exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
//   d  = sqrt(v) / bc2_sqrt + eps              (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
//   p  = p + (neg_step * m) / d                param.addcdiv_(exp_avg, denom, value=-step_size)
// bc2_sqrt = sqrt(1 - beta2^t) and neg_step = -lr / (1 - beta1^t) are evaluated by the caller in double and rounded once.
struct AdamArgs {
  float* p; const float* g; float* m; float* v;
  int64_t n;
  float w1, beta2, w2, bc2_sqrt, eps, neg_step;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  m = m + a.w1 * (g - m);
  v = v * a.beta2 + (a.w2 * g) * g;
  const float d = sqrtf(v) / a.bc2_sqrt + a.eps;   // (hipcc: sqrt and divide are correctly rounded by default)
  p = p + (a.neg_step * m) / d;
}

__global__ void __launch_bounds__(256) adam_arena_kernel(AdamArgs a, int vec) {
  const int64_t n4 = vec ? (a.n >> 2) : 0;   // vec: all four bases 16-byte aligned (an arena or an aligned range of one)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = tid; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 m = reinterpret_cast<float4*>(a.m)[i];
    float4 v = reinterpret_cast<float4*>(a.v)[i];
    adam_one(p.x, g.x, m.x, v.x, a);
    adam_one(p.y, g.y, m.y, v.y, a);
    adam_one(p.z, g.z, m.z, v.z, a);
    adam_one(p.w, g.w, m.w, v.w, a);
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
  }
  for (int64_t t = (n4 << 2) + tid; t < a.n; t += stride) adam_one(a.p[t], a.g[t], a.m[t], a.v[t], a);   // the last n % 4 (or an unaligned range)
}

int num_cus();

hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps, int64_t step,
                       hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  const AdamArgs a{p, g, m, v, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)std::sqrt(bc2), (float)eps, (float)(-(lr / bc1))};
  const int64_t work = vec ? ((n >> 2) + 3) : n;
  int64_t blocks = (work + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  adam_arena_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(a, vec);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// CodeLibraryArticulated (models/code_library.py:36-53) for the reference's batch of ONE object in ONE state: three row gathers,
// and the dense table gradients nn.Embedding's autograd produces (zero everywhere but the looked-up row).
// ---------------------------------------------------------------------------------------------
struct CodeLibArgs {
  const float* table[3]; float* out[3];        // forward: table -> out row; backward: `table` = incoming row gradient, `out` = table gradient
  const int64_t* idx[3];
  int rows[3], dim[3];
  int blk_begin[4];
};

__global__ void __launch_bounds__(256) code_library_fwd_kernel(CodeLibArgs a) {
  const int t = blockIdx.x;   // one block per table
  const int64_t r = a.idx[t][0];
  // nn.Embedding raises on an out-of-range index; a kernel cannot raise, and a clamped row would be silently plausible: the row is NaN,
  // which every loss downstream shows
  const bool ok = r >= 0 && r < a.rows[t];
  for (int c = threadIdx.x; c < a.dim[t]; c += blockDim.x) a.out[t][c] = ok ? a.table[t][r * a.dim[t] + c] : __builtin_nanf("");
}

__global__ void __launch_bounds__(256) code_library_bwd_kernel(CodeLibArgs a) {
  int t = 0;
#pragma unroll 1
  for (int q = 1; q < 3; ++q)
    if ((int)blockIdx.x >= a.blk_begin[q]) t = q;
  const int64_t r = a.idx[t][0];
  const int i = ((int)blockIdx.x - a.blk_begin[t]) * 256 + threadIdx.x;
  if (i >= a.rows[t] * a.dim[t]) return;
  const int row = i / a.dim[t], c = i % a.dim[t];
  a.out[t][i] = row == r ? a.table[t][c] : 0.f;
}

hipError_t launch_code_library(bool backward, const float* const* src, const int64_t* const* idx, const int* rows, const int* dim, float* const* dst,
                               hipStream_t stream) {
  CodeLibArgs a{};
  int blk = 0;
  for (int t = 0; t < 3; ++t) {
    a.table[t] = src[t]; a.out[t] = dst[t]; a.idx[t] = idx[t]; a.rows[t] = rows[t]; a.dim[t] = dim[t];
    a.blk_begin[t] = blk;
    blk += (rows[t] * dim[t] + 255) / 256;
  }
  a.blk_begin[3] = blk;
  if (backward)
    code_library_bwd_kernel<<<dim3(blk), dim3(256), 0, stream>>>(a);
  else
    code_library_fwd_kernel<<<dim3(3), dim3(256), 0, stream>>>(a);
  return hipGetLastError();
}

}  // namespace aon
