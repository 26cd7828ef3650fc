// Non-GEMM stages of the NeRF render path for gfx950: ray generation, stratified sampling, positional
// encoding (stage-level entry point), alpha compositing and hierarchical inverse-CDF sampling.
//
// Their algorithmic bound is HBM bytes; what actually limits compositing and the inverse CDF is VALU issue (round-2 counters:
// 437 / 538 wave instructions per ray), so these kernels are written for few instructions as much as for coalesced bytes.
// Rays are the parallel axis: compositing and the inverse CDF give one 64-lane wavefront to each ray, lanes stride over the
// ray's samples so every load/store of a per-ray sample buffer is a contiguous burst, and the transmittance product / CDF sum /
// merge are wavefront scans and shuffles (no LDS round trip except the 64-entry CDF table the binary search gathers from and
// the 193-slot row the merge scatters into).  The coarse level's compositing and the fine level's sampling are one kernel.
#include "aon_ray_core.h"

namespace aon {


// ---------------------------------------------------------------------------------------------
// R1 + R2  ray generation   (datasets/ray_utils.py:71-90, 118-159)
// ---------------------------------------------------------------------------------------------
struct RaygenArgs {
  float c2w[12];  // row-major (3,4)
  int H, W;
  float focal;
  int64_t pix_begin, pix_end;  // row-major pixel range [begin, end) to generate
  float* rays_o;   // (n,3)
  float* viewdirs; // (n,3) unit directions
  float* rays_d;   // (n,3) or null; the reference's rays_d aliases viewdirs (ray_utils.py:146-147)
  const float* directions;  // (H*W,3) precomputed camera-space directions, or null -> pinhole model below
};

__global__ void raygen_kernel(RaygenArgs a) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t pix = a.pix_begin + k;
  if (pix >= a.pix_end) return;
  const int j = (int)(pix / a.W), i = (int)(pix % a.W);
  // ((i - W/2)/focal, -(j - H/2)/focal, -1), no +0.5 pixel centre (ray_utils.py:86-88)
  float dx = __fdiv_rn(__fsub_rn((float)i, (float)a.W * 0.5f), a.focal);
  float dy = -__fdiv_rn(__fsub_rn((float)j, (float)a.H * 0.5f), a.focal);
  float dz = -1.0f;
  if (a.directions) { dx = a.directions[pix * 3]; dy = a.directions[pix * 3 + 1]; dz = a.directions[pix * 3 + 2]; }
  float d[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)  // directions @ c2w[:, :3].T
    d[r] = __builtin_fmaf(dz, a.c2w[4 * r + 2], __builtin_fmaf(dy, a.c2w[4 * r + 1], __fmul_rn(dx, a.c2w[4 * r + 0])));
  const float nrm = __fsqrt_rn(__builtin_fmaf(d[2], d[2], __builtin_fmaf(d[1], d[1], __fmul_rn(d[0], d[0]))));
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float v = __fdiv_rn(d[r], nrm);
    a.viewdirs[k * 3 + r] = v;
    if (a.rays_d) a.rays_d[k * 3 + r] = v;
    a.rays_o[k * 3 + r] = a.c2w[4 * r + 3];
  }
}

__global__ void ray_directions_kernel(int H, int W, float focal, float* __restrict__ out) {
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (int64_t)H * W) return;
  const int j = (int)(pix / W), i = (int)(pix % W);
  out[pix * 3 + 0] = __fdiv_rn(__fsub_rn((float)i, (float)W * 0.5f), focal);
  out[pix * 3 + 1] = -__fdiv_rn(__fsub_rn((float)j, (float)H * 0.5f), focal);
  out[pix * 3 + 2] = -1.0f;
}

hipError_t launch_ray_directions(int H, int W, float focal, float* out, hipStream_t stream) {
  const int64_t n = (int64_t)H * W;
  ray_directions_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(H, W, focal, out);
  return hipGetLastError();
}

hipError_t launch_raygen(const float* c2w, int H, int W, float focal, const float* directions, int64_t pix_begin,
                         int64_t pix_end, float* rays_o, float* viewdirs, float* rays_d, hipStream_t stream) {
  RaygenArgs a;
  a.directions = directions;
  for (int i = 0; i < 12; ++i) a.c2w[i] = c2w[i];
  a.H = H; a.W = W; a.focal = focal; a.pix_begin = pix_begin; a.pix_end = pix_end;
  a.rays_o = rays_o; a.viewdirs = viewdirs; a.rays_d = rays_d;
  const int64_t n = pix_end - pix_begin;
  if (n <= 0) return hipSuccess;
  raygen_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a);
  return hipGetLastError();
}

// radii of get_rays(..., output_radii=True)  (datasets/ray_utils.py:138-143; a mip-NeRF leftover the render path never
// reads, but part of the only call form the reference datasets use: sapien.py:102,145, sapien_multi.py:301,343):
//   d = directions @ c2w[:, :3].T (un-normalised);  dx[j,i] = || d[j,i] - d[j+1,i] ||  for j < H-1,
//   row H-1 <- row H-3 (`cat([dx, dx[-2:-1]])`: dx has H-1 rows, so -2 is image row H-3);  radius = dx * 2 / sqrt(12)
struct RadiiArgs {
  float c2w[12];
  int H, W;
  const float* directions;  // (H*W,3)
  float* radii;             // (H*W,)
};

__global__ void ray_radii_kernel(RadiiArgs a) {
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (int64_t)a.H * a.W) return;
  int j = (int)(pix / a.W);
  const int i = (int)(pix % a.W);
  if (j == a.H - 1) j = a.H - 3;
  auto world = [&](int64_t q, float (&d)[3]) {
    const float dx = a.directions[q * 3], dy = a.directions[q * 3 + 1], dz = a.directions[q * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      d[r] = __builtin_fmaf(dz, a.c2w[4 * r + 2], __builtin_fmaf(dy, a.c2w[4 * r + 1], __fmul_rn(dx, a.c2w[4 * r + 0])));
  };
  float d0[3], d1[3];
  world((int64_t)j * a.W + i, d0);
  world((int64_t)(j + 1) * a.W + i, d1);
  const float e0 = __fsub_rn(d0[0], d1[0]), e1 = __fsub_rn(d0[1], d1[1]), e2 = __fsub_rn(d0[2], d1[2]);
  const float dx = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(e0, e0), __fmul_rn(e1, e1)), __fmul_rn(e2, e2)));
  a.radii[pix] = __fdiv_rn(__fmul_rn(dx, 2.0f), 3.4641016151377544f);  // sqrt(tensor(12, int8)) -> fp32 sqrt(12)
}

hipError_t launch_ray_radii(const float* directions, const float* c2w, int H, int W, float* radii, hipStream_t stream) {
  RadiiArgs a;
  for (int i = 0; i < 12; ++i) a.c2w[i] = c2w[i];
  a.H = H; a.W = W; a.directions = directions; a.radii = radii;
  const int64_t n = (int64_t)H * W;
  ray_radii_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// R3  stratified sampling   (models/vanilla_nerf/helper.py:106-133, both branches of `lindisp`)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float linspace01(int idx, int steps) {
  // torch.linspace(0, 1, steps) (CPU kernel): step = 1/(steps-1) in fp32; first half counts up from start, second half counts
  // down from end -- `end - step * k` as ONE fused multiply-add (the kernel is compiled with contraction; found by probing: an
  // un-fused subtraction differs in 2-25 % of the elements unless 1/(steps-1) is a power of two, as it is for the reference's
  // 65 -- and checked for every length 2 .. 300 and on the reference's lindisp outputs at 8, 33 and 201 steps, G16).
  const float step = __fdiv_rn(1.0f, (float)(steps - 1));
  return idx < steps / 2 ? __fmul_rn(step, (float)idx) : __builtin_fmaf(-step, (float)(steps - idx - 1), 1.0f);
}

// The planes of a level: near / far as the reference's fp32 tensor arithmetic sees its Python scalars, and -- lindisp, helper.py:117 --
// fp32(1.0 / near), fp32(1.0 / far), which the reference evaluates in Python DOUBLE precision before they meet the tensor (so they are
// the caller's to compute: from the fp32 `near` alone the last bit can differ).
struct TRange {
  float near, far, inv_near, inv_far;
  int lindisp;
};

__device__ __forceinline__ float coarse_t(int idx, int steps, const TRange& r) {
  const float s = linspace01(idx, steps);
  if (r.lindisp)  // 1.0 / (1.0/near * (1 - s) + 1.0/far * s); `1.0 / tensor` is tensor.reciprocal() * 1.0: one IEEE division
    return __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(r.inv_near, __fsub_rn(1.0f, s)), __fmul_rn(r.inv_far, s)));
  return __fadd_rn(__fmul_rn(r.near, __fsub_rn(1.0f, s)), __fmul_rn(r.far, s));  // near*(1-s) + far*s
}

__global__ void sample_along_rays_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                         int64_t n_rays, int S, TRange tr,
                                         const float* __restrict__ t_rand, float* __restrict__ t_vals,
                                         float* __restrict__ coords) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_rays * S) return;
  const int64_t ray = g / S;
  const int s = (int)(g - ray * S);
  float t = coarse_t(s, S, tr);
  if (t_rand) {  // stratified jitter between interval mid-points (helper.py:122-127)
    const float lo = s == 0 ? t : __fmul_rn(0.5f, __fadd_rn(t, coarse_t(s - 1, S, tr)));
    const float hi = s == S - 1 ? t : __fmul_rn(0.5f, __fadd_rn(coarse_t(s + 1, S, tr), t));
    t = __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), t_rand[g]));
  }
  t_vals[g] = t;
  if (coords) {
#pragma unroll
    for (int a = 0; a < 3; ++a) coords[g * 3 + a] = __fadd_rn(rays_o[ray * 3 + a], __fmul_rn(t, rays_d[ray * 3 + a]));
  }
}

__global__ void cast_rays_kernel(const float* __restrict__ t_vals, const float* __restrict__ o, const float* __restrict__ d,
                                 int64_t n_rays, int S, float* __restrict__ coords) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_rays * S) return;
  const int64_t ray = g / S;
  const float t = t_vals[g];
#pragma unroll
  for (int a = 0; a < 3; ++a) coords[g * 3 + a] = __fadd_rn(o[ray * 3 + a], __fmul_rn(t, d[ray * 3 + a]));
}

hipError_t launch_cast_rays(const float* t_vals, const float* o, const float* d, int64_t n_rays, int S, float* coords,
                            hipStream_t stream) {
  const int64_t n = n_rays * S;
  if (n <= 0) return hipSuccess;
  cast_rays_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(t_vals, o, d, n_rays, S, coords);
  return hipGetLastError();
}

// The render / training paths want only t (the fused MLP kernels cast the rays themselves): four consecutive elements of the
// flat (n*S) array per thread, one 16-byte store (and one 16-byte load of t_rand) each -- a pure streaming write.
__global__ void sample_t4_kernel(int64_t total, int S, TRange tr, const float* __restrict__ t_rand, float* __restrict__ t_vals) {
  const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (g0 >= total) return;
  int s = (int)(g0 % S);
  float r[4] = {0.f, 0.f, 0.f, 0.f};
  const bool full = g0 + 4 <= total;
  if (t_rand) {
    if (full) {
      const float4 q = *reinterpret_cast<const float4*>(t_rand + g0);
      r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
    } else {
      for (int e = 0; e < 4; ++e) if (g0 + e < total) r[e] = t_rand[g0 + e];
    }
  }
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = coarse_t(s, S, tr);
    if (t_rand) {  // stratified jitter between interval mid-points (helper.py:122-127)
      const float lo = s == 0 ? t : __fmul_rn(0.5f, __fadd_rn(t, coarse_t(s - 1, S, tr)));
      const float hi = s == S - 1 ? t : __fmul_rn(0.5f, __fadd_rn(coarse_t(s + 1, S, tr), t));
      t = __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), r[e]));
    }
    v[e] = t;
    s = s + 1 == S ? 0 : s + 1;
  }
  if (full) {
    *reinterpret_cast<float4*>(t_vals + g0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    for (int e = 0; e < 4; ++e) if (g0 + e < total) t_vals[g0 + e] = v[e];
  }
}

hipError_t launch_sample_along_rays(const float* rays_o, const float* rays_d, int64_t n_rays, int S, float near,
                                    float far, const float* t_rand, float* t_vals, float* coords, hipStream_t stream,
                                    int lindisp, float inv_near, float inv_far) {
  const int64_t n = n_rays * S;
  if (n <= 0) return hipSuccess;
  const TRange tr{near, far, inv_near, inv_far, lindisp};
  const bool aligned = (reinterpret_cast<uintptr_t>(t_vals) & 15) == 0 && (reinterpret_cast<uintptr_t>(t_rand) & 15) == 0;
  if (!coords && aligned) {
    const int64_t threads = (n + 3) / 4;
    sample_t4_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream>>>(n, S, tr, t_rand, t_vals);
    return hipGetLastError();
  }
  sample_along_rays_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(rays_o, rays_d, n_rays, S, tr,
                                                                                       t_rand, t_vals, coords);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// R4  positional encoding, stage-level entry point   (helper.py:136-140)
// (the render path never materialises this tensor: the fused MLP kernel encodes in registers)
// ---------------------------------------------------------------------------------------------
// `ld` >= F floats between output rows; the pad columns F .. ld-1 are written as zeros (the layer-wise engine reads its operands in
// whole 16-byte pieces: csrc/aon_gmlp.hip).  `levels_out` > L: the output has the column layout of an encoding with that many
// levels -- [x ; sin block of 3 levels_out ; shifted block] -- with zeros in the slots of the levels this encoding lacks (the
// fused kernels' 63 / 27-wide input slots fed by a network with fewer levels, aon_pack_vanilla_mlp_deg).
__global__ void pos_enc_kernel(const float* __restrict__ x, int64_t n, int min_deg, int max_deg, int ld, int levels_out, float* __restrict__ out) {
  const int L = max_deg - min_deg;
  const int F = 3 + 6 * levels_out;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * ld) return;
  const int64_t row = g / ld;
  const int f = (int)(g - row * ld);
  float v = 0.f;
  if (f < 3) {
    v = x[row * 3 + f];
  } else if (f < F) {
    const int e = (f - 3) % (3 * levels_out);
    const bool shifted = (f - 3) >= 3 * levels_out;
    if (e / 3 < L) {
      const float xb = __fmul_rn(x[row * 3 + e % 3], __builtin_ldexpf(1.0f, min_deg + e / 3));
      v = sin_f32(shifted ? __fadd_rn(xb, AON_HALF_PI_F32) : xb);
    }
  }
  out[g] = v;
}

hipError_t launch_pos_enc(const float* x, int64_t n, int min_deg, int max_deg, float* out, hipStream_t stream, int ld, int levels_out) {
  if (levels_out < max_deg - min_deg) levels_out = max_deg - min_deg;
  const int F = 3 + 6 * levels_out;
  if (ld < F) ld = F;
  const int64_t tot = n * ld;
  if (tot <= 0) return hipSuccess;
  pos_enc_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream>>>(x, n, min_deg, max_deg, ld, levels_out, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// R6 + R7  inverse-CDF sampling and sort-merge   (helper.py:203-252), one wavefront per ray
// ---------------------------------------------------------------------------------------------
// Fixed to the reference's default geometry: 64 bins (mids of 65 coarse t's), 63 weights, 128 new samples.
struct PdfArgs {
  const float* bins;     // (n,64) or null -> mids of t_coarse
  const float* weights;  // pointer to the first of the 63 pdf weights of ray 0
  int64_t w_stride;      // floats between rays (63 for a dense (n,63) tensor, 65 for coarse weights[...,1:-1])
  const float* t_coarse; // (n,65) or null (samples-only call)
  const float* u;        // (128,) if u_stride == 0 else (n,128)
  int64_t u_stride;
  int64_t n_rays;
  float* samples;        // (n,128) or null
  float* t_fine;         // (n,193) or null
};


// ---------------------------------------------------------------------------------------------
// R8  alpha compositing   (helper.py:157-195), one wavefront per ray -- optionally fused with R6 + R7 of the coarse level
// ---------------------------------------------------------------------------------------------
// act: 0 = inputs already activated (stage-level parity with volumetric_rendering)
//      1 = vanilla NeRF: rgb = sigmoid(raw), sigma = relu(raw)                     (model.py:186-187)
//      2 = articulated:  rgb = sigmoid(raw)*(1+2*0.001)-0.001, sigma = softplus(raw-1)  (model_autodecoder.py:321-323)
struct CompositeArgs {
  const float* rgb;    int rgb_stride;    // floats between consecutive samples (3 or 4)
  const float* sigma;  int sigma_stride;  // 1 or 4
  const float* t_vals;  // (n,S)
  const float* dirs;    // (n,3)
  int64_t n_rays; int S; int white_bkgd; ActParams ap;
  float* comp_rgb;  // (n,3)
  float* acc;       // (n,)
  float* depth;     // (n,)
  float* weights;   // (n,S) or null
  // fused coarse level (S == 65 only): the level's weights never leave the registers, the kernel goes on to draw the
  // fine samples (model.py:162-173) and writes sort(cat[t_coarse, draws]) itself
  const float* u; int64_t u_stride;   // (128,) if u_stride == 0 else (n,128)
  float* t_fine;                      // (n,193)
};

// 1 / (1 + exp(-x)).  The compositing kernels are bound by VALU issue (SQ_INSTS_VALU: 437 wave instructions per 193-sample ray,
// three sigmoids per sample among them), so this is the short form: e = 2^(-x log2 e) straight on the transcendental unit --
// the rounding of the product costs |x| 4e-8 relative on e, which reaches the RESULT as at most 1e-8 absolute (e / (1 + e)^2
// is 0.25 at x = 0, where the product is exact, and 0.0066 at |x| = 5) -- then v_rcp_f32 (1 ulp) refined by one Newton step;
// 8 instructions against 18 with the library expf and its range fix-ups.  x < -87: e overflows, the true value is below
// 1.7e-38 -> 0 (a NaN input fails the comparison and stays NaN).  The density's alpha = 1 - exp(-sigma delta) keeps the
// library expf: its error goes into the transmittance product and the inverse CDF's weights.
__device__ __forceinline__ float sigmoid_f32(float x) {
  const float e = __builtin_amdgcn_exp2f(__fmul_rn(x, -1.44269502162933349609375f));
  const float d = __fadd_rn(1.0f, e);
  const float r = __builtin_amdgcn_rcpf(d);
  const float s = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  return x < -87.0f ? 0.f : s;
}

// torch Softplus(beta=1, threshold=20): x > 20 ? x : log1p(exp(x)), written as max(x, 0) + log1p(z), z = exp(-|x|) in (0, 1]
// (above 20 the second term is below half an ulp of x: the threshold needs no branch).  log1p(z) = z * P(z) with P the degree-10
// Chebyshev interpolant of log1p(z)/z on [0, 1] (1.1e-7 relative in fp32 Horner form, tests/diag/diag_sigmoid.py checks the
// function on the device); z straight from the transcendental unit.  ~20 instructions against ~70 with the library expf and
// log1pf, which had the articulated compositing at twice the vanilla kernel's instruction count.  NaN stays NaN (through z),
// +inf -> +inf, -inf -> 0.
__device__ __forceinline__ float softplus_f32(float x) {
  const float z = __builtin_amdgcn_exp2f(__fmul_rn(__builtin_fabsf(x), -1.44269502162933349609375f));
  float p = 0.001986696617677808f;
  p = __builtin_fmaf(p, z, -0.013187826611101627f);
  p = __builtin_fmaf(p, z, 0.041006576269865036f);
  p = __builtin_fmaf(p, z, -0.08188041299581528f);
  p = __builtin_fmaf(p, z, 0.12377995997667313f);
  p = __builtin_fmaf(p, z, -0.16087622940540314f);
  p = __builtin_fmaf(p, z, 0.19885820150375366f);
  p = __builtin_fmaf(p, z, -0.24986496567726135f);
  p = __builtin_fmaf(p, z, 0.33332496881484985f);
  p = __builtin_fmaf(p, z, -0.4999997913837433f);
  p = __builtin_fmaf(p, z, 1.0f);
  return __fadd_rn(__builtin_fmaxf(x, 0.f), __fmul_rn(z, p));
}

// output activations of the two networks on one (rgb, sigma) record of sample g; everything in `ap` is wave-uniform
__device__ __forceinline__ void activate_record(const ActParams& ap, int64_t g, float& c0, float& c1, float& c2, float& sg) {
  if (ap.noise) sg = __fadd_rn(sg, __fmul_rn(ap.noise[g], ap.noise_std));   // model.py:183-184
  if (ap.act == 1) {
    sg = __builtin_fmaxf(sg, 0.f);
    c0 = sigmoid_f32(c0); c1 = sigmoid_f32(c1); c2 = sigmoid_f32(c2);
  } else if (ap.act == 2) {
    sg = softplus_f32(__fadd_rn(sg, ap.sigma_bias));
    c0 = __fsub_rn(__fmul_rn(sigmoid_f32(c0), ap.rgb_scale), ap.rgb_shift);
    c1 = __fsub_rn(__fmul_rn(sigmoid_f32(c1), ap.rgb_scale), ap.rgb_shift);
    c2 = __fsub_rn(__fmul_rn(sigmoid_f32(c2), ap.rgb_scale), ap.rgb_shift);
  }
}

// gfx950 lane-swap instructions: permlane32_swap exchanges lanes 32..63 of `a` with lanes 0..31 of `b`, permlane16_swap the
// odd 16-lane rows of `a` with the even rows of `b`.  `a + b` afterwards holds pair sums of BOTH inputs: one swap and one add
// take two values one reduction level down.
// (inline asm, not __builtin_amdgcn_permlane32_swap: hipcc 7.2 extracts the builtin's second result from the FIRST register in
// this kernel -- `v_add_f32 v2, v9, v9` -- and every sum came out as 4 x its first row; the asm names both in/out registers.
// s_nop 1: the two wait states a lane-swap needs after a VALU write of its operands, which the compiler cannot see in here.)
__device__ __forceinline__ float swap32_add(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __fadd_rn(a, b);   // [a_i + a_{i+32} | b_i + b_{i+32}]
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __fadd_rn(a, b);   // rows [a0+a1, b0+b1, a2+a3, b2+b3]
}
// sums over the 64 lanes of four values at once: 3 swaps + 3 adds + 4 row-rotate adds (24 DPP adds done one value at a
// time); every lane of row 0 / 1 / 2 / 3 ends up holding the total of v0 / v2 / v1 / v3.  The association is a fixed tree
// (lane pairs 32 apart, then 16, 8, 4, 2, 1), the same for every ray.
__device__ __forceinline__ void wave_sum4(float& v0, float& v1, float& v2, float& v3) {
  float z = swap16_add(swap32_add(v0, v1), swap32_add(v2, v3));   // rows: v0, v2, v1, v3 -- 16 partial sums each
  z = __fadd_rn(z, dpp_f32<0x128, 0xf>(0.f, z));  // row_ror:8
  z = __fadd_rn(z, dpp_f32<0x124, 0xf>(0.f, z));  // row_ror:4
  z = __fadd_rn(z, dpp_f32<0x122, 0xf>(0.f, z));  // row_ror:2
  z = __fadd_rn(z, dpp_f32<0x121, 0xf>(0.f, z));  // row_ror:1
  const int zi = __builtin_bit_cast(int, z);
  v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(zi, 0));
  v2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(zi, 16));
  v1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(zi, 32));
  v3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(zi, 48));
}

// One wavefront per ray, lanes over samples (the transmittance product is a wave scan); FUSE_PDF: the coarse level's kernel,
// which goes on to the inverse CDF with the weights still in registers.
// PACKED: rgb and sigma are the (n*S,4) float4 records the MLP kernels write (rgb_stride = sigma_stride = 4,
// sigma = rgb + 3): one 16-byte load per sample instead of four 4-byte ones.
// The kernel is bound by VALU issue (round-2 counters: ~300 wave instructions per ray against 1,592 bytes), so: 64-sample
// blocks over the first S-1 samples only -- the LAST sample (the 1e10-long interval, helper.py:163) sits alone in a 65th /
// 193rd position and is evaluated once, wave-uniformly, from operands fetched ahead of the blocks, instead of as a further
// block with one live lane; four of the five ray sums are reduced together (wave_sum4).
// SC: the sample count as a compile-time constant (65 / 193: the two levels of the reference geometry; full 64-sample blocks
// lose their bounds predicates and the row addressing its multiplies) or 0 = read it from the arguments.
template <bool PACKED, bool FUSE_PDF, int SC>
__global__ void __launch_bounds__(256) composite_kernel(CompositeArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[FUSE_PDF ? 4 : 1][FUSE_PDF ? kPdfLdsFloats : 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
  if (ray >= a.n_rays) return;  // wave-uniform; no block-level barrier below
  const int S = SC ? SC : a.S, last = S - 1;
  const ActParams ap = a.ap;
  const float* tv = a.t_vals + ray * S;
  // the last sample's operands (same address in every lane: one request), in flight while the blocks run
  const int64_t gl = ray * S + last;
  const float t_last = tv[last];
  float l0, l1, l2, lsg;
  if constexpr (PACKED) {
    const float4 r = reinterpret_cast<const float4*>(a.rgb)[gl];
    l0 = r.x; l1 = r.y; l2 = r.z; lsg = r.w;
  } else {
    lsg = a.sigma[gl * a.sigma_stride];
    l0 = a.rgb[gl * a.rgb_stride + 0]; l1 = a.rgb[gl * a.rgb_stride + 1]; l2 = a.rgb[gl * a.rgb_stride + 2];
  }
  const float dn = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(a.dirs[ray * 3], a.dirs[ray * 3]),
                                                  __fmul_rn(a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 1])),
                                        __fmul_rn(a.dirs[ray * 3 + 2], a.dirs[ray * 3 + 2])));
  float carry = 1.0f;  // transmittance entering this 64-sample block
  float s_r = 0.f, s_g = 0.f, s_b = 0.f, s_w = 0.f, s_d = 0.f;
  float w0 = 0.f, t0 = 0.f, tn0 = 0.f;   // block 0's weights and t, t_next (the fused inverse CDF's operands)
  for (int base = 0; base < last; base += 64) {
    const int s = base + lane;
    const bool in = s < last;
    float alpha = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, t = 0.f, tn = 0.f;
    if (in) {
      const int64_t g = ray * S + s;
      t = tv[s];
      tn = tv[s + 1];
      const float dist = __fmul_rn(__fsub_rn(tn, t), dn);
      float sg;
      if constexpr (PACKED) {
        const float4 r = reinterpret_cast<const float4*>(a.rgb)[g];
        c0 = r.x; c1 = r.y; c2 = r.z; sg = r.w;
      } else {
        sg = a.sigma[g * a.sigma_stride];
        c0 = a.rgb[g * a.rgb_stride + 0]; c1 = a.rgb[g * a.rgb_stride + 1]; c2 = a.rgb[g * a.rgb_stride + 2];
      }
      activate_record(ap, g, c0, c1, c2, sg);
      alpha = __fsub_rn(1.0f, expf(-__fmul_rn(sg, dist)));
    }
    // T_i = prod_{j<i} (1 - alpha_j + 1e-10)   (helper.py:169-176)
    const float f = in ? __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f) : 1.0f;
    const float incl = wave_inclusive_scan<true>(f, lane);
    const float excl = dpp_f32<0x138, 0xf>(1.0f, incl);  // wave_shr:1, lane 0 keeps 1
    const float T = __fmul_rn(carry, excl);
    carry = __fmul_rn(carry, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63)));
    const float w = __fmul_rn(alpha, T);
    if (FUSE_PDF && base == 0) { w0 = w; t0 = t; tn0 = tn; }
    if (in) {
      s_r = __fadd_rn(s_r, __fmul_rn(w, c0));
      s_g = __fadd_rn(s_g, __fmul_rn(w, c1));
      s_b = __fadd_rn(s_b, __fmul_rn(w, c2));
      s_w = __fadd_rn(s_w, w);
      s_d = __fadd_rn(s_d, __fmul_rn(w, t));
      if (a.weights) a.weights[ray * S + s] = w;
    }
  }
  // the last sample: delta = 1e10 (helper.py:163), T = everything before it
  activate_record(ap, gl, l0, l1, l2, lsg);
  const float a_last = __fsub_rn(1.0f, expf(-__fmul_rn(lsg, __fmul_rn(1e10f, dn))));
  const float w_last = __fmul_rn(a_last, carry);
  if (a.weights && lane == 0) a.weights[gl] = w_last;
  wave_sum4(s_r, s_g, s_b, s_w);
  s_d = wave_sum(s_d);
  s_r = __fadd_rn(s_r, __fmul_rn(w_last, l0));
  s_g = __fadd_rn(s_g, __fmul_rn(w_last, l1));
  s_b = __fadd_rn(s_b, __fmul_rn(w_last, l2));
  s_w = __fadd_rn(s_w, w_last);
  s_d = __fadd_rn(s_d, __fmul_rn(w_last, t_last));
  if (lane == 0) {
    if (a.white_bkgd) {  // comp_rgb + (1 - acc)
      const float bg = __fsub_rn(1.0f, s_w);
      s_r = __fadd_rn(s_r, bg); s_g = __fadd_rn(s_g, bg); s_b = __fadd_rn(s_b, bg);
    }
    a.comp_rgb[ray * 3 + 0] = s_r; a.comp_rgb[ray * 3 + 1] = s_g; a.comp_rgb[ray * 3 + 2] = s_b;
    a.acc[ray] = s_w;
    // helper.py:182-183: nan_to_num(depth, nan=inf); the clamp to the batch's own [min,max] is the identity
    a.depth[ray] = (s_d != s_d) ? __builtin_inff() : s_d;
  }
  if constexpr (FUSE_PDF) {
    // model.py:162-166: bins = mids of t_coarse, pdf weights = weights[..., 1:-1] -> lane i takes w_{i+1} (i < 63)
    const float w_next = dpp_f32<0x130, 0xf>(0.f, w0);   // wave_shl:1; lane 63 reads 0
    const float b = __fmul_rn(0.5f, __fadd_rn(tn0, t0));
    inverse_cdf_merge(lds[wv], lane, t0, t_last, b, lane < 63 ? w_next : 0.f, a.u + ray * a.u_stride, nullptr, a.t_fine + ray * 193);
  }
}

hipError_t launch_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* t_vals,
                            const float* dirs, int64_t n_rays, int S, int white_bkgd, const ActParams& ap, float* comp_rgb,
                            float* acc, float* depth, float* weights, hipStream_t stream) {
  if (n_rays <= 0) return hipSuccess;
  CompositeArgs a{rgb, rgb_stride, sigma, sigma_stride, t_vals, dirs, n_rays, S, white_bkgd, ap, comp_rgb, acc, depth, weights,
                  nullptr, 0, nullptr};
  const bool packed = rgb_stride == 4 && sigma_stride == 4 && sigma == rgb + 3 && (reinterpret_cast<uintptr_t>(rgb) & 15) == 0;
  const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
  if (packed && S == 193) composite_kernel<true, false, 193><<<grid, block, 0, stream>>>(a);
  else if (packed && S == 65) composite_kernel<true, false, 65><<<grid, block, 0, stream>>>(a);
  else if (packed) composite_kernel<true, false, 0><<<grid, block, 0, stream>>>(a);
  else composite_kernel<false, false, 0><<<grid, block, 0, stream>>>(a);
  return hipGetLastError();
}

// coarse level of NeRF.forward in one launch: compositing of the 65 coarse samples (outputs as launch_composite; `weights`
// optional) + the 128 inverse-CDF draws from weights[..., 1:-1] over the mids of t_coarse + the sorted union -> t_fine (n,193).
// `raw` are the MLP kernels' packed (rgb, sigma) records.
hipError_t launch_composite_pdf(const float* raw, const float* t_coarse, const float* dirs, int64_t n_rays, int white_bkgd, const ActParams& ap,
                                const float* u, int64_t u_stride, float* comp_rgb, float* acc, float* depth, float* weights,
                                float* t_fine, hipStream_t stream) {
  if (n_rays <= 0) return hipSuccess;
  CompositeArgs a{raw, 4, raw + 3, 4, t_coarse, dirs, n_rays, 65, white_bkgd, ap, comp_rgb, acc, depth, weights, u, u_stride, t_fine};
  composite_kernel<true, true, 65><<<dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, stream>>>(a);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) sample_pdf_kernel(PdfArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[4][kPdfLdsFloats];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
  if (ray >= a.n_rays) return;  // wave-uniform; no block-level barrier below

  // bins: 0.5*(t[i+1]+t[i])  (model.py:163)
  float tc = 0.f, t64 = 0.f;
  if (a.t_coarse) {
    tc = a.t_coarse[ray * 65 + lane];
    t64 = a.t_coarse[ray * 65 + 64];
  }
  float b;
  if (a.bins) {
    b = a.bins[ray * 64 + lane];
  } else {
    const float tn = a.t_coarse[ray * 65 + lane + 1];
    b = __fmul_rn(0.5f, __fadd_rn(tn, tc));
  }
  const float w = lane < 63 ? a.weights[ray * a.w_stride + lane] : 0.f;
  inverse_cdf_merge(lds[wv], lane, tc, t64, b, w, a.u + ray * a.u_stride, a.samples ? a.samples + ray * 128 : nullptr,
                    a.t_fine ? a.t_fine + ray * 193 : nullptr);
}

hipError_t launch_sample_pdf(const float* bins, const float* weights, int64_t w_stride, const float* t_coarse,
                             const float* u, int64_t u_stride, int64_t n_rays, float* samples, float* t_fine,
                             hipStream_t stream) {
  if (n_rays <= 0) return hipSuccess;
  PdfArgs a{bins, weights, w_stride, t_coarse, u, u_stride, n_rays, samples, t_fine};
  sample_pdf_kernel<<<dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, stream>>>(a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// R6 + R7 for ANY geometry (num_coarse_samples, num_fine_samples of NeRF.__init__, model.py:129-130): nb bins, nb - 1 weights,
// nf draws, nt coarse t's -> nt + nf sorted t's.  One wavefront per ray, everything of the ray in LDS.  Same arithmetic contract
// as the 64/128 kernel above -- the draws are bit-exact against torch's CPU kernels -- with ATen's reductions followed literally
// instead of being unrolled for one length:
//   weights.sum(-1): cpu/SumKernel.cpp.  K >= 8: `vectorized_inner_sum` = `row_sum` over 8-float vectors (vector lane l of this
//     wave plays SIMD lane l), ILP factor 4, `multi_row_sum`'s four-level cascade (level step max(16, 2^(ceil(log2 n)/4)));
//     then the K % 8 tail and the eight lane sums in order.  K < 8: the same row_sum on scalars.  (Checked against torch.sum on
//     K = 1 .. 1000 in tests/test_oracle_golden.py::test_aten_sum_model, and through the reference's draws in G16.)
//   torch.cumsum: a double running sum in index order, every prefix rounded to float.
//   torch.sort: the sorted multiset is unique -- a bitonic network over the +inf-padded union.
// ---------------------------------------------------------------------------------------------
struct PdfNArgs {
  const float* bins;     // (n,nb) or null -> mids of t_coarse (then nt == nb + 1)
  const float* weights;  // first of ray 0's nb-1 weights
  int64_t w_stride;
  const float* t_coarse; // (n,nt) or null (samples-only call)
  const float* u; int64_t u_stride;   // (nf,) shared when u_stride == 0, else (n,nf)
  int64_t n_rays;
  int nb, nf, nt, P;     // P = power of two >= nt + nf
  float* samples;        // (n,nf) or null
  float* t_fine;         // (n,nt+nf) or null
};

// ATen's multi_row_sum on four interleaved rows: element (i, k) = x[(4 i + k) * stride + off], i < size
__device__ __forceinline__ void aten_multi_row_sum4(const float* x, int stride, int off, int size, float (&out)[4]) {
  float acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[j][k] = 0.f;
  int cl = 0;
  while ((1 << cl) < size) ++cl;                       // ceil(log2(size)) (0 for size <= 1)
  const int level_power = cl / 4 > 4 ? cl / 4 : 4;
  const int level_step = 1 << level_power, level_mask = level_step - 1;
  int i = 0;
  while (i + level_step <= size) {
    for (int j = 0; j < level_step; ++j, ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[0][k] = __fadd_rn(acc[0][k], x[(4 * i + k) * stride + off]);
    }
    bool go = true;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      if (go) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[j][k] = __fadd_rn(acc[j][k], acc[j - 1][k]); acc[j - 1][k] = 0.f; }
        if ((i & (level_mask << (j * level_power))) != 0) go = false;
      }
    }
  }
  for (; i < size; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[0][k] = __fadd_rn(acc[0][k], x[(4 * i + k) * stride + off]);
  }
#pragma unroll
  for (int j = 1; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[0][k] = __fadd_rn(acc[0][k], acc[j][k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = acc[0][k];
}

// torch's CPU sum of the K floats x[0..K) (LDS), returned in every lane
__device__ __forceinline__ float aten_row_sum(const float* x, int K, int lane) {
  if (K < 8) {   // scalar_inner_sum -> row_sum with the scalar load policy (wave-uniform: every lane does the same work)
    const int size_ilp = K / 4;
    float part[4];
    aten_multi_row_sum4(x, 1, 0, size_ilp, part);
    for (int i = size_ilp * 4; i < K; ++i) part[0] = __fadd_rn(part[0], x[i]);
    for (int k = 1; k < 4; ++k) part[0] = __fadd_rn(part[0], part[k]);
    return part[0];
  }
  const int vec = K / 8, size_ilp = vec / 4;
  const int l = lane & 7;   // lanes 8..63 repeat lanes 0..7 (in-bounds reads, no divergence)
  float part[4];
  aten_multi_row_sum4(x, 8, l, size_ilp, part);
  for (int i = size_ilp * 4; i < vec; ++i) part[0] = __fadd_rn(part[0], x[i * 8 + l]);
  for (int k = 1; k < 4; ++k) part[0] = __fadd_rn(part[0], part[k]);
  float fin = 0.f;
  for (int k = vec * 8; k < K; ++k) fin = __fadd_rn(fin, x[k]);
#pragma unroll
  for (int k = 0; k < 8; ++k) fin = __fadd_rn(fin, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part[0]), k)));
  return fin;
}

__global__ void __launch_bounds__(64) sample_pdf_n_kernel(PdfNArgs a) {
  extern __shared__ __attribute__((aligned(16))) float L[];
  const int lane = threadIdx.x;
  const int64_t ray = blockIdx.x;
  const int nb = a.nb, K = nb - 1, nf = a.nf, nt = a.nt;
  float* w = L;              // K weights -> pdf
  float* cdf = L + nb;       // nb
  float* bin = L + 2 * nb;   // nb
  float* key = L + 3 * nb;   // P: coarse t, draws, +inf padding
  if (a.t_coarse)
    for (int i = lane; i < nt; i += 64) key[i] = a.t_coarse[ray * nt + i];
  for (int i = lane; i < K; i += 64) w[i] = a.weights[ray * a.w_stride + i];
  if (a.bins)
    for (int i = lane; i < nb; i += 64) bin[i] = a.bins[ray * nb + i];
  wave_lds_sync();
  if (!a.bins)   // model.py:163  0.5 * (t[1:] + t[:-1])
    for (int i = lane; i < nb; i += 64) bin[i] = __fmul_rn(0.5f, __fadd_rn(key[i + 1], key[i]));
  // helper.py:205-211
  float wsum = aten_row_sum(w, K, lane);
  const float padding = __builtin_fmaxf(0.f, __fsub_rn(1e-5f, wsum));
  const float pad_each = __fdiv_rn(padding, (float)K);
  wsum = __fadd_rn(wsum, padding);
  wave_lds_sync();
  for (int i = lane; i < K; i += 64) w[i] = __fdiv_rn(__fadd_rn(w[i], pad_each), wsum);
  wave_lds_sync();
  // helper.py:212-222: cdf = [0, min(1, cumsum(pdf[:-1])), 1]
  double run = 0.0;
  for (int i = 0; i + 1 < K; ++i) {   // wave-uniform chain (broadcast LDS reads); lane i % 64 keeps prefix i
    run += (double)w[i];
    if ((i & 63) == lane) cdf[i + 1] = __builtin_fminf(1.f, (float)run);
  }
  if (lane == 0) { cdf[0] = 0.f; cdf[nb - 1] = 1.f; }
  wave_lds_sync();
  // helper.py:224-243
  for (int j = lane; j < nf; j += 64) {
    const float u = a.u[ray * a.u_stride + j];
    int lo = 0, hi = nb;               // idx = #(cdf <= u): the mask `u >= cdf` of the reference is a prefix (cdf non-decreasing)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int i0 = lo - 1 < 0 ? 0 : lo - 1, i1 = lo > nb - 1 ? nb - 1 : lo;
    const float c0 = cdf[i0], c1 = cdf[i1], b0 = bin[i0], b1 = bin[i1];
    float t = __fdiv_rn(__fsub_rn(u, c0), __fsub_rn(c1, c0));
    if (t != t) t = 0.f;
    t = __builtin_fminf(__builtin_fmaxf(t, 0.f), 1.f);
    const float smp = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
    if (a.samples) a.samples[ray * nf + j] = smp;
    key[nt + j] = smp;
  }
  if (!a.t_fine) return;
  const int P = a.P, tot = nt + nf;
  for (int i = tot + lane; i < P; i += 64) key[i] = __builtin_inff();
  wave_lds_sync();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = lane; p < P / 2; p += 64) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));   // the pair's lower index: bit j clear
        const int q = i | j;
        const float x = key[i], y = key[q];
        const bool up = (i & k) == 0;
        const float mn = __builtin_fminf(x, y), mx = __builtin_fmaxf(x, y);
        key[i] = up ? mn : mx;
        key[q] = up ? mx : mn;
      }
      wave_lds_sync();
    }
  }
  for (int i = lane; i < tot; i += 64) a.t_fine[ray * tot + i] = key[i];
}

int64_t sample_pdf_n_lds_bytes(int nb, int nf, int nt, int* P_out) {
  int P = 2;
  while (P < nt + nf) P <<= 1;
  if (P_out) *P_out = P;
  return (int64_t)(3 * nb + P) * 4;
}

hipError_t launch_sample_pdf_n(const float* bins, const float* weights, int64_t w_stride, const float* t_coarse, const float* u,
                               int64_t u_stride, int64_t n_rays, int nb, int nf, int nt, float* samples, float* t_fine, hipStream_t stream) {
  if (n_rays <= 0) return hipSuccess;
  PdfNArgs a{bins, weights, w_stride, t_coarse, u, u_stride, n_rays, nb, nf, nt, 0, samples, t_fine};
  const int64_t lds = sample_pdf_n_lds_bytes(nb, nf, nt, &a.P);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  sample_pdf_n_kernel<<<dim3((unsigned)n_rays), dim3(64), (size_t)lds, stream>>>(a);
  return hipGetLastError();
}

// ---- Training losses as TWO launches: helper.py:17-22 (img2mse, mse2psnr), model.py:271-273 (loss0 + loss1),
// model_autodecoder.py:460-466 (+ 1e-4 * sum of mean |code|).  Written in torch the forward and backward of these lines are ~47 launches
// of 5-7 us around 12,288 numbers -- 0.29 ms of a 31 ms step (profiles/r05_step_timeline.txt).
// Forward: one workgroup, fp64 sums in a fixed order, every output rounded once.  stats = {loss0, loss1, reg, loss, psnr0, psnr1};
// loss = fl(fl(loss1 + loss0) + reg) as the reference adds them.
struct LossArgs {
  const float* rgb[2];   // coarse, fine (n,3); rgb[0] may be null (num_levels = 1: loss0 = 0 and no gradient)
  const float* target;   // (n,3)
  int64_t numel;         // 3 n
  const float* lat[3];   // latent codes whose mean |.| is regularised (null: none)
  int lat_len[3];
  float reg_scale;       // 1e-4
  float* stats;          // (8,)
  float* loss;           // (1,) the differentiable output
  const float* go;       // backward: d loss (device scalar)
  float* d_rgb[2];       // backward outputs
  float* d_lat[3];
};

__global__ __launch_bounds__(1024) void train_loss_fwd_kernel(LossArgs a) {
  __shared__ double red[2][16];
  __shared__ double lat_red[3][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s[2] = {0.0, 0.0};
  // four strides per trip, every load issued before the first use: one workgroup's sum is a chain of load latencies otherwise
  // (12 trips at 4096 rays); the order of the additions per thread is the element order either way
  // (round 6: the loads are UNCONDITIONAL, from an index clamped into the arrays, and out-of-range elements are turned into d = 0
  // afterwards -- written as `in ? load : 0` every load sat in a branch of its own with an s_waitcnt vmcnt(0) behind it, and the twelve
  // loads of a trip ran one after the other: 16 us for 12,288 numbers.  Same values, same order of additions.)
  const float* rgb0 = a.rgb[0] ? a.rgb[0] : a.target;   // (no coarse level: d = target - target = 0, as before)
  for (int64_t i0 = tid; i0 < a.numel; i0 += 4 * 1024) {
    float t[4], x[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = i0 + (int64_t)e * 1024;
      const int64_t ic = i < a.numel ? i : a.numel - 1;
      t[e] = a.target[ic];
      x[0][e] = rgb0[ic];
      x[1][e] = a.rgb[1][ic];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = i0 + (int64_t)e * 1024 < a.numel;
#pragma unroll
      for (int l = 0; l < 2; ++l) { const float d = in ? __fsub_rn(x[l][e], t[e]) : 0.f; s[l] += (double)d * (double)d; }
    }
  }
  double ls[3] = {0.0, 0.0, 0.0};
  {
    // the three latents' first elements per thread together (codes are 128 / 128 / 32 long: one trip), then whatever is left
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int len = a.lat[k] ? a.lat_len[k] : 0;
      const float* src = a.lat[k] ? a.lat[k] : a.target;
      const float x = src[tid < len ? tid : 0];
      v[k] = tid < len ? x : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (a.lat[k] && tid < a.lat_len[k]) ls[k] += (double)__builtin_fabsf(v[k]);
      if (a.lat[k])
        for (int i = tid + 1024; i < a.lat_len[k]; i += 1024) ls[k] += (double)__builtin_fabsf(a.lat[k][i]);
    }
  }
  // wave sums by shuffles (fixed order), then the 16 wave partials in order
  for (int off = 32; off > 0; off >>= 1) {
    s[0] += __shfl_down(s[0], off); s[1] += __shfl_down(s[1], off);
    ls[0] += __shfl_down(ls[0], off); ls[1] += __shfl_down(ls[1], off); ls[2] += __shfl_down(ls[2], off);
  }
  if (lane == 0) { red[0][wave] = s[0]; red[1][wave] = s[1]; lat_red[0][wave] = ls[0]; lat_red[1][wave] = ls[1]; lat_red[2][wave] = ls[2]; }
  __syncthreads();
  if (tid == 0) {
    double t[2] = {0.0, 0.0}, lt[3] = {0.0, 0.0, 0.0};
    for (int w = 0; w < 16; ++w) { t[0] += red[0][w]; t[1] += red[1][w]; lt[0] += lat_red[0][w]; lt[1] += lat_red[1][w]; lt[2] += lat_red[2][w]; }
    const float loss0 = a.rgb[0] ? (float)(t[0] / (double)a.numel) : 0.f;
    const float loss1 = (float)(t[1] / (double)a.numel);
    float means = 0.f;   // torch.mean(...) + torch.mean(...) + torch.mean(...), left to right in fp32
    for (int k = 0; k < 3; ++k)
      if (a.lat[k]) means = __fadd_rn(means, (float)(lt[k] / (double)a.lat_len[k]));
    const float reg = __fmul_rn(a.reg_scale, means);
    const float loss = __fadd_rn(__fadd_rn(loss1, loss0), reg);
    const double inv_ln10 = 1.0 / 2.302585092994046;
    a.stats[0] = loss0; a.stats[1] = loss1; a.stats[2] = reg; a.stats[3] = loss;
    a.stats[4] = a.rgb[0] ? (float)(-10.0 * log((double)loss0) * inv_ln10) : 0.f;
    a.stats[5] = (float)(-10.0 * log((double)loss1) * inv_ln10);
    a.stats[6] = 0.f; a.stats[7] = 0.f;
    a.loss[0] = loss;
  }
}

// Backward, the operations autograd would run: mean -> g = fl(go / numel); pow(2) -> fl(g * fl(2 d)); torch.norm(code, dim=0) of a
// one-row code is |x|: fl(x * fl(gm / |x|)), 0 where x = 0 (norm_backward's masked_fill); gm = fl(fl(go * scale) / len).
__global__ __launch_bounds__(256) void train_loss_bwd_kernel(LossArgs a) {
  const float go = a.go[0];
  const float g = __fdiv_rn(go, (float)a.numel);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.numel; i += (int64_t)gridDim.x * 256) {
    const float t = a.target[i];
    for (int l = 0; l < 2; ++l)
      if (a.rgb[l] && a.d_rgb[l]) a.d_rgb[l][i] = __fmul_rn(g, __fmul_rn(2.f, __fsub_rn(a.rgb[l][i], t)));
  }
  if (blockIdx.x == 0) {
    const float gs = __fmul_rn(go, a.reg_scale);
    for (int k = 0; k < 3; ++k) {
      if (!a.lat[k] || !a.d_lat[k]) continue;
      const float gm = __fdiv_rn(gs, (float)a.lat_len[k]);
      for (int i = threadIdx.x; i < a.lat_len[k]; i += 256) {
        const float x = a.lat[k][i], n = __builtin_fabsf(x);
        a.d_lat[k][i] = n == 0.f ? 0.f : __fmul_rn(x, __fdiv_rn(gm, n));
      }
    }
  }
}

hipError_t launch_train_loss(bool backward, const float* rgb_c, const float* rgb_f, const float* target, int64_t n, const float* const* lat, const int* lat_len,
                             float reg_scale, float* stats, float* loss, const float* go, float* d_rgb_c, float* d_rgb_f, float* const* d_lat, hipStream_t stream) {
  LossArgs a{};
  a.rgb[0] = rgb_c; a.rgb[1] = rgb_f; a.target = target; a.numel = 3 * n; a.reg_scale = reg_scale;
  for (int k = 0; k < 3; ++k) { a.lat[k] = lat ? lat[k] : nullptr; a.lat_len[k] = lat_len ? lat_len[k] : 0; a.d_lat[k] = d_lat ? d_lat[k] : nullptr; }
  a.stats = stats; a.loss = loss; a.go = go; a.d_rgb[0] = d_rgb_c; a.d_rgb[1] = d_rgb_f;
  if (!backward) {
    train_loss_fwd_kernel<<<dim3(1), dim3(1024), 0, stream>>>(a);
  } else {
    const int64_t blocks = (a.numel + 255) / 256;
    train_loss_bwd_kernel<<<dim3((unsigned)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks))), dim3(256), 0, stream>>>(a);
  }
  return hipGetLastError();
}

}  // namespace aon
