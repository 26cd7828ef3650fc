// Non-GEMM stages of the NeRF render path for gfx950: ray generation, stratified sampling, positional
// encoding (stage-level entry point), alpha compositing and hierarchical inverse-CDF sampling.
//
// Their algorithmic bound is HBM bytes; what actually limits compositing and the inverse CDF is VALU issue (round-2 counters:
// 437 / 538 wave instructions per ray), so these kernels are written for few instructions as much as for coalesced bytes.
// Rays are the parallel axis: compositing and the inverse CDF give one 64-lane wavefront to each ray, lanes stride over the
// ray's samples so every load/store of a per-ray sample buffer is a contiguous burst, and the transmittance product / CDF sum /
// merge are wavefront scans and shuffles (no LDS round trip except the 64-entry CDF table the binary search gathers from and
// the 193-slot row the merge scatters into).  The coarse level's compositing and the fine level's sampling are one kernel.
#include "aon_common.h"

namespace aon {

// ---------------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------------
// Cross-lane data movement by DPP modifiers (no LDS-path ds_bpermute): row_shr:n inside rows of 16 lanes,
// row_bcast:15 / row_bcast:31 to carry a row's last lane into the following rows (GFX9 DPP controls).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float identity, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, false));
}

// inclusive scan over the 64 lanes; lanes without a source keep the identity
template <bool MUL>
__device__ __forceinline__ float wave_inclusive_scan(float v, int /*lane*/) {
  if constexpr (MUL) {
    // v <- dpp(v) * v with the DPP modifier on the multiply itself: a lane without a source (or outside the row mask) is
    // disabled and keeps its value, which is what multiplying by the identity did -- one instruction per step instead of
    // [move 1.0, v_mov_b32_dpp, v_mul_f32] (hipcc folds the DPP move into an add but not into a multiply).  s_nop 1: the two
    // wait states a DPP read needs after the VALU write of its source, which the compiler cannot see inside the asm.
    asm volatile(
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
  } else {
    v = v + dpp_f32<0x111, 0xf>(0.f, v);  // row_shr:1
    v = v + dpp_f32<0x112, 0xf>(0.f, v);  // row_shr:2
    v = v + dpp_f32<0x114, 0xf>(0.f, v);  // row_shr:4
    v = v + dpp_f32<0x118, 0xf>(0.f, v);  // row_shr:8
    v = v + dpp_f32<0x142, 0xa>(0.f, v);  // row_bcast:15 -> rows 1 and 3
    v = v + dpp_f32<0x143, 0xc>(0.f, v);  // row_bcast:31 -> rows 2 and 3
    return v;
  }
}

// wave_shr:1 of a double (two 32-bit DPP moves), zero into lane 0
__device__ __forceinline__ double dpp_shr1_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  // bound_ctrl: a lane without a source reads 0 -- no separate "old = 0" move per half per step
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x138, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// a DPP move of a double (two 32-bit moves); lanes without a source / outside ROW_MASK read +0.0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// inclusive prefix sum of doubles over the 64 lanes, as a 6-step tree (NOT index order: see exact_prefix_f64)
__device__ __forceinline__ double wave_inclusive_sum_f64(double v) {
  v += dpp_f64<0x111, 0xf>(v);
  v += dpp_f64<0x112, 0xf>(v);
  v += dpp_f64<0x114, 0xf>(v);
  v += dpp_f64<0x118, 0xf>(v);
  v += dpp_f64<0x142, 0xa>(v);
  v += dpp_f64<0x143, 0xc>(v);
  return v;
}

// inclusive prefix minimum of ints over the 64 lanes
__device__ __forceinline__ int wave_inclusive_min_i32(int v) {
  constexpr int id = 0x7fffffff;
  auto step = [](int a, int b) { return a < b ? a : b; };
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x114, 0xf, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x118, 0xf, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x142, 0xa, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x143, 0xc, 0xf, false));
  return v;
}

// sum over the 64 lanes, returned in every lane
__device__ __forceinline__ float wave_sum(float v) {
  v = wave_inclusive_scan<false>(v, 0);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

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
// R3  stratified sampling   (models/vanilla_nerf/helper.py:106-133, lindisp=False)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float linspace01(int idx, int steps) {
  // torch.linspace(0, 1, steps) (CPU kernel): step = 1/(steps-1); first half counts up from start,
  // second half counts down from end.
  const float step = __fdiv_rn(1.0f, (float)(steps - 1));
  return idx < steps / 2 ? __fmul_rn(step, (float)idx) : __fsub_rn(1.0f, __fmul_rn(step, (float)(steps - idx - 1)));
}

__device__ __forceinline__ float coarse_t(int idx, int steps, float near, float far) {
  const float s = linspace01(idx, steps);
  return __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, s)), __fmul_rn(far, s));  // near*(1-s) + far*s
}

__global__ void sample_along_rays_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                         int64_t n_rays, int S, float near, float far,
                                         const float* __restrict__ t_rand, float* __restrict__ t_vals,
                                         float* __restrict__ coords) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_rays * S) return;
  const int64_t ray = g / S;
  const int s = (int)(g - ray * S);
  float t = coarse_t(s, S, near, far);
  if (t_rand) {  // stratified jitter between interval mid-points (helper.py:122-127)
    const float lo = s == 0 ? t : __fmul_rn(0.5f, __fadd_rn(t, coarse_t(s - 1, S, near, far)));
    const float hi = s == S - 1 ? t : __fmul_rn(0.5f, __fadd_rn(coarse_t(s + 1, S, near, far), t));
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
__global__ void sample_t4_kernel(int64_t total, int S, float near, float far, const float* __restrict__ t_rand, float* __restrict__ t_vals) {
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
    float t = coarse_t(s, S, near, far);
    if (t_rand) {  // stratified jitter between interval mid-points (helper.py:122-127)
      const float lo = s == 0 ? t : __fmul_rn(0.5f, __fadd_rn(t, coarse_t(s - 1, S, near, far)));
      const float hi = s == S - 1 ? t : __fmul_rn(0.5f, __fadd_rn(coarse_t(s + 1, S, near, far), t));
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
                                    float far, const float* t_rand, float* t_vals, float* coords, hipStream_t stream) {
  const int64_t n = n_rays * S;
  if (n <= 0) return hipSuccess;
  const bool aligned = (reinterpret_cast<uintptr_t>(t_vals) & 15) == 0 && (reinterpret_cast<uintptr_t>(t_rand) & 15) == 0;
  if (!coords && aligned) {
    const int64_t threads = (n + 3) / 4;
    sample_t4_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream>>>(n, S, near, far, t_rand, t_vals);
    return hipGetLastError();
  }
  sample_along_rays_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(rays_o, rays_d, n_rays, S, near, far,
                                                                                       t_rand, t_vals, coords);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// R4  positional encoding, stage-level entry point   (helper.py:136-140)
// (the render path never materialises this tensor: the fused MLP kernel encodes in registers)
// ---------------------------------------------------------------------------------------------
__global__ void pos_enc_kernel(const float* __restrict__ x, int64_t n, int min_deg, int max_deg, float* __restrict__ out) {
  const int L = max_deg - min_deg;
  const int F = 3 + 6 * L;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * F) return;
  const int64_t row = g / F;
  const int f = (int)(g - row * F);
  float v;
  if (f < 3) {
    v = x[row * 3 + f];
  } else {
    const int e = (f - 3) % (3 * L);
    const bool shifted = (f - 3) >= 3 * L;
    const float xb = __fmul_rn(x[row * 3 + e % 3], __builtin_ldexpf(1.0f, min_deg + e / 3));
    v = sin_f32(shifted ? __fadd_rn(xb, AON_HALF_PI_F32) : xb);
  }
  out[g] = v;
}

hipError_t launch_pos_enc(const float* x, int64_t n, int min_deg, int max_deg, float* out, hipStream_t stream) {
  const int64_t tot = n * (3 + 6 * (max_deg - min_deg));
  if (tot <= 0) return hipSuccess;
  pos_enc_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream>>>(x, n, min_deg, max_deg, out);
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

template <int K, int J>
__device__ __forceinline__ void bitonic_step(float (&v)[4], int lane) {
  if constexpr (J >= 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float o = __shfl_xor(v[r], J >> 2);
      const int e = lane * 4 + r;
      const bool up = (e & K) == 0, lower = (e & J) == 0;
      v[r] = (up == lower) ? __builtin_fminf(v[r], o) : __builtin_fmaxf(v[r], o);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if ((r & J) == 0) {
        const int e = lane * 4 + r;
        const bool up = (e & K) == 0;
        const float lo = __builtin_fminf(v[r], v[r ^ J]), hi = __builtin_fmaxf(v[r], v[r ^ J]);
        v[r] = up ? lo : hi;
        v[r ^ J] = up ? hi : lo;
      }
    }
  }
}

template <int K>
__device__ __forceinline__ void bitonic_merge(float (&v)[4], int lane) {
  if constexpr (K >= 256) bitonic_step<K, 128>(v, lane);
  if constexpr (K >= 128) bitonic_step<K, 64>(v, lane);
  if constexpr (K >= 64) bitonic_step<K, 32>(v, lane);
  if constexpr (K >= 32) bitonic_step<K, 16>(v, lane);
  if constexpr (K >= 16) bitonic_step<K, 8>(v, lane);
  if constexpr (K >= 8) bitonic_step<K, 4>(v, lane);
  if constexpr (K >= 4) bitonic_step<K, 2>(v, lane);
  bitonic_step<K, 1>(v, lane);
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wave is serviced in order; this only pins the compiler's ordering and drains lgkmcnt.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// weights.sum(-1) over the 63 pdf weights in EXACTLY the association torch's CPU sum kernel uses for a contiguous fp32 row
// (helper.py:205; ATen cpu/SumKernel.cpp `vectorized_inner_sum` -> `row_sum`, 8-float vectors, ILP factor 4), measured on
// torch 2.10 with three-element probes (1, 2^-24, 2^-24) over all position triples and confirmed bit-for-bit on 20,000 rows:
//   P[l] = (((((x[l] + x[32+l]) + x[40+l]) + x[48+l]) + x[8+l]) + x[16+l]) + x[24+l]          l = 0..7   (vector lanes)
//   s    = (((((x56 + x57) + x58) + x59) + x60) + x61) + x62                                               (scalar remainder)
//   s    = (((((((s + P0) + P1) + P2) + P3) + P4) + P5) + P6) + P7
// A tree reduction differs from this in the last bit on ~30 % of rows, the cdf inherits that bit, and every draw that falls
// next to the affected knot moves: the whole reason round 1's inverse CDF was "99 % within 2e-6" instead of exact.
__device__ __forceinline__ float torch_sum63(float x, int lane) {
  float P = x;                                                   // lanes 0..7: vector accumulator lane l
  P = __fadd_rn(P, __shfl(x, lane + 32));
  P = __fadd_rn(P, __shfl(x, lane + 40));
  P = __fadd_rn(P, __shfl(x, lane + 48));
  P = __fadd_rn(P, __shfl(x, lane + 8));
  P = __fadd_rn(P, __shfl(x, lane + 16));
  P = __fadd_rn(P, __shfl(x, lane + 24));
  // y[0..14] = x56..x62, P0..P7 in lanes 0..14; the running sum in index order by the shifted-add chain used for the cdf
  const float y_tail = __shfl(x, lane + 56), y_part = __shfl(P, lane - 7);   // both shuffles by ALL lanes (a shuffle inside
  const float y = lane < 7 ? y_tail : y_part;                                 // a divergent branch cannot read disabled lanes)
  float run = y;
#pragma unroll
  for (int k = 0; k < 14; ++k) run = __fadd_rn(dpp_f32<0x138, 0xf>(0.f, run), y);   // wave_shr:1, zero into lane 0
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, run), 14));
}

// torch.cumsum of a float row on the CPU: the running sum is a DOUBLE (at::acc_type<float, false>), taken in index order, and
// every prefix is rounded to float on store.  Returns, in lane j+1, that float for prefix j of the first 62 lanes' values
// (lane 0 gets 0) -- bit for bit.
//
// Fast path: a 6-step tree scan in double.  It gives the same bits as the index-order sum whenever
//   (a) every prefix is exactly representable in a double -- then no addition in either order rounds at all.  With
//       non-negative terms that is a statement about exponents only: [exponent of the prefix] - [smallest ulp exponent among
//       its float terms] <= 52, checked per lane from a prefix minimum; or
//   (b) the prefix is far from every float rounding boundary: the two double sums differ by at most (61 + 6) roundings of
//       2^-53 relative, so if P(1 - 2^-46) and P(1 + 2^-46) round to the same float, so does the index-order sum.
// A wave in which some lane satisfies neither (or holds a negative / NaN term) runs the 61-step index-order chain
//   c <- wave_shr1(c) + p   (lane i holds (((p0 + p1) + p2) + ...) + p_i after i steps; "0 + p" of finished lanes is exact)
// which was the only path in the first round-2 version (57 us per 61,440-ray launch; a float running sum reproduces only 44 %
// of torch's prefixes, a double one all of them).  On NeRF weights the fallback is rare: 0 of 3,000 rays on the synthetic
// scenes, 0.4 % of rows with weights spread over 17 decades (tests/diag/diag_cumsum_guard.py).
__device__ __forceinline__ float exact_prefix_f64(float pdf, int lane) {
  const double pd = (double)pdf;
  const unsigned pb = __builtin_bit_cast(unsigned, pdf);
  int ue = (int)((pb >> 23) & 255u);
  ue = ue < 1 ? 1 : ue;                                     // denormals share the ulp of the smallest normal binade
  if (pdf == 0.f) ue = 0x7fffffff;                          // a zero term constrains nothing
  const int min_ulp = wave_inclusive_min_i32(ue);
  double P = wave_inclusive_sum_f64(pd);
  const int ed = (int)((__builtin_bit_cast(unsigned long long, P) >> 52) & 0x7ffu);
  // (ed - 1023) - (min_ulp - 127 - 23) <= 52, i.e. ed - min_ulp <= 925
  const bool exact = ed - min_ulp <= 924;   // one binade of slack: P's exponent may be one below the true sum's
  const float f_lo = (float)(P * (1.0 - 0x1p-46)), f_hi = (float)(P * (1.0 + 0x1p-46));
  const bool ok = (exact || f_lo == f_hi) && pdf >= 0.f;    // NaN terms fail both comparisons
  if (__builtin_amdgcn_ballot_w64(!ok && lane < 62) != 0ull) {   // wave-uniform
    double run = pd;
#pragma unroll
    for (int j = 0; j < 61; ++j) run = dpp_shr1_f64(run) + pd;
    P = run;
  }
  return (float)dpp_shr1_f64(P);
}

// inclusive prefix sum of ints over the 64 lanes
__device__ __forceinline__ int wave_inclusive_add_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// Per-wave LDS image of one ray's inverse-CDF problem.
constexpr int kPdfCdf = 0, kPdfBin = 64, kPdfT = 128, kPdfOut = 196, kPdfLdsFloats = 392;

// R6 + R7 for ONE ray by one wavefront.  In: lane i holds bin i (`b`), pdf weight i (`w`, lanes 0..62) and, if `t_fine` is
// wanted, t_coarse[i] (`tc`) with t_coarse[64] in `t64`; `u_row` points at the ray's 128 draws.  Shared by the stand-alone
// kernel (operands from memory) and the fused coarse compositing kernel (operands still in registers).
__device__ __forceinline__ void inverse_cdf_merge(float* L, int lane, float tc, float t64, float b, float w, const float* u_row,
                                                  float* samples, float* t_fine) {
  float* cdf = L + kPdfCdf;
  float* bin = L + kPdfBin;
  float* tt = L + kPdfT;
  float* ob = L + kPdfOut;
  bin[lane] = b;
  if (t_fine) {
    tt[lane] = tc;
    if (lane == 0) tt[64] = t64;
  }

  // pdf / cdf  (helper.py:206-222)
  float wsum = torch_sum63(w, lane);
  const float padding = __builtin_fmaxf(0.f, __fsub_rn(1e-5f, wsum));
  if (padding != 0.f) {   // wave-uniform (wsum is); w + 0/63 and wsum + 0 are the identity, so the common case skips a division
    w = __fadd_rn(w, __fdiv_rn(padding, 63.0f));
    wsum = __fadd_rn(wsum, padding);
  }
  const float pdf = __fdiv_rn(w, wsum);
  // cdf64 = [0, min(1, cumsum(pdf[:-1])) (62 entries), 1]
  const float prefix = exact_prefix_f64(pdf, lane);   // lane j+1 <- float(p0 + ... + pj) in torch's arithmetic; lane 0 <- 0
  const float mine = __builtin_fminf(1.f, prefix);
  cdf[lane] = lane == 63 ? 1.f : mine;  // lane 0 keeps 0
  wave_lds_sync();

  float smp[2];
  int guess[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = lane + 64 * k;
    const float u = u_row[j];
    // idx = #(cdf <= u)  == searchsorted(cdf, u, right=True); cdf is non-decreasing
    int idx = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
      if (cdf[idx + step - 1] <= u) idx += step;
    }
    if (idx == 63 && cdf[63] <= u) idx = 64;
    const int i0 = idx - 1 < 0 ? 0 : idx - 1;
    const int i1 = idx > 63 ? 63 : idx;
    const float c0 = cdf[i0], c1 = cdf[i1], b0 = bin[i0], b1 = bin[i1];
    float t = __fdiv_rn(__fsub_rn(u, c0), __fsub_rn(c1, c0));
    if (t != t) t = 0.f;                                   // nan_to_num(., 0)
    t = __builtin_fminf(__builtin_fmaxf(t, 0.f), 1.f);     // clip (+-inf land on the same ends as nan_to_num + clip)
    smp[k] = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
    guess[k] = i0 + 1;   // with bins = interval mids: t_0 .. t_i0 <= bin_i0 <= sample
    if (samples) samples[j] = smp[k];
  }
  if (!t_fine) return;

  // sort(cat[t_coarse(65), samples(128)]) -> 193   (helper.py:250).  A sorted multiset is unique, so any correct merge gives
  // torch.sort's values.  Fast path -- t_coarse non-decreasing (always, out of sample_along_rays) and the draws too (they are
  // when u is: the deterministic grid, or pre-sorted random u), decided by a wave-uniform order check: every sample finds its
  // rank c = #{i : t_i <= s} among the coarse t by a short walk from the bin index its draw landed in and goes to slot j + c
  // of a NaN-filled 193-slot LDS row; the coarse t fill the remaining slots in order (slot e takes t[e - #samples before e],
  // counted with one integer scan).  ~75 wave instructions against ~170 for the 8-stage bitonic merge of the first version.
  // (cross-lane reads are done by ALL lanes into temporaries and selected afterwards: inside a `lane == 63 ? a : shuffle`
  // arm lane 63 is disabled and lane 62 would read the identity instead of lane 63's value)
  const float t_next = tt[lane + 1];
  const float s1_first = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, smp[1]), 0));
  const float s0_shl = dpp_f32<0x130, 0xf>(0.f, smp[0]);          // wave_shl:1 -> lane+1's value
  const float s1_next = dpp_f32<0x130, 0xf>(0.f, smp[1]);
  const float s0_next = lane == 63 ? s1_first : s0_shl;
  const bool ok = tc <= t_next && smp[0] <= s0_next && (lane == 63 || smp[1] <= s1_next);
  float v[4];
  bool merged = false;   // wave-uniform
  if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) {
    const float qnan = __builtin_nanf("");
    if (lane < 49) *reinterpret_cast<float4*>(ob + 4 * lane) = make_float4(qnan, qnan, qnan, qnan);
    // (LDS instructions of one wave execute in order: the fill above lands before the scatter below)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      // rank c = #{i : t_i <= s}: t_0 .. t_g-1 lie below the sample's bin (g = guess), t_g+2 .. above it
      const int g = guess[k];
      const float sv = smp[k];
      const int g1 = g < 64 ? g : 64, g2 = g + 1 < 64 ? g + 1 : 64;
      const float ta = tt[g1], tb = tt[g2];
      const int c = g + ((g <= 64 && ta <= sv) ? 1 : 0) + ((g + 1 <= 64 && tb <= sv) ? 1 : 0);
      ob[lane + 64 * k + c] = sv;
    }
    wave_lds_sync();
    float4 q = make_float4(qnan, qnan, qnan, qnan);
    if (lane < 49) q = *reinterpret_cast<const float4*>(ob + 4 * lane);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) cnt += (v[r] == v[r]) ? 1 : 0;
    const int incl = wave_inclusive_add_i32(cnt);
    int before = incl - cnt;   // samples in slots < 4 * lane
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = lane * 4 + r;
      const int ti = e - before;
      const float tv = tt[ti < 0 ? 0 : (ti > 64 ? 64 : ti)];
      if (v[r] == v[r]) ++before;
      else if (e < 193) v[r] = tv;
    }
    // The rank above leans on bins being the mids of t_coarse (any `bins` can be passed to the stand-alone entry) and on the
    // draws' rounding; it is therefore CHECKED: all 128 samples landed in distinct slots and the 193 keys are in order -- a
    // permutation of the inputs in non-decreasing order is the sorted sequence.  Otherwise fall through to the full sort.
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const float nxt = dpp_f32<0x130, 0xf>(0.f, v[0]);  // wave_shl:1 -> lane+1's first element
    const bool in_order = (lane * 4 + 1 >= 193 || v[0] <= v[1]) && (lane * 4 + 2 >= 193 || v[1] <= v[2]) &&
                          (lane * 4 + 3 >= 193 || v[2] <= v[3]) && (lane * 4 + 4 >= 193 || v[3] <= nxt);
    merged = total == 128 && __builtin_amdgcn_ballot_w64(in_order) == ~0ull;
  }
  if (!merged) {
    // general order: full 36-stage bitonic sort of 256 (+inf padded) keys held 4 per lane (element e = lane*4 + r)
    ob[lane] = tc;
    if (lane == 0) ob[64] = t64;
    ob[65 + lane] = smp[0];
    ob[129 + lane] = smp[1];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = lane * 4 + r;
      v[r] = e < 193 ? ob[e] : __builtin_inff();
    }
    bitonic_merge<2>(v, lane); bitonic_merge<4>(v, lane); bitonic_merge<8>(v, lane); bitonic_merge<16>(v, lane);
    bitonic_merge<32>(v, lane); bitonic_merge<64>(v, lane); bitonic_merge<128>(v, lane); bitonic_merge<256>(v, lane);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = lane * 4 + r;
    if (e < 193) t_fine[e] = v[r];
  }
}

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
  int64_t n_rays; int S; int white_bkgd; int act;
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

// output activations of the two networks on one (rgb, sigma) record; `act` is wave-uniform
__device__ __forceinline__ void activate_record(int act, float& c0, float& c1, float& c2, float& sg) {
  if (act == 1) {
    sg = __builtin_fmaxf(sg, 0.f);
    c0 = sigmoid_f32(c0); c1 = sigmoid_f32(c1); c2 = sigmoid_f32(c2);
  } else if (act == 2) {
    sg = softplus_f32(__fadd_rn(sg, -1.0f));
    c0 = __fsub_rn(__fmul_rn(sigmoid_f32(c0), 1.002f), 0.001f);
    c1 = __fsub_rn(__fmul_rn(sigmoid_f32(c1), 1.002f), 0.001f);
    c2 = __fsub_rn(__fmul_rn(sigmoid_f32(c2), 1.002f), 0.001f);
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
  const int act = a.act;
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
      activate_record(act, c0, c1, c2, sg);
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
  activate_record(act, l0, l1, l2, lsg);
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
                            const float* dirs, int64_t n_rays, int S, int white_bkgd, int act, float* comp_rgb,
                            float* acc, float* depth, float* weights, hipStream_t stream) {
  if (n_rays <= 0) return hipSuccess;
  CompositeArgs a{rgb, rgb_stride, sigma, sigma_stride, t_vals, dirs, n_rays, S, white_bkgd, act, comp_rgb, acc, depth, weights,
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
hipError_t launch_composite_pdf(const float* raw, const float* t_coarse, const float* dirs, int64_t n_rays, int white_bkgd, int act,
                                const float* u, int64_t u_stride, float* comp_rgb, float* acc, float* depth, float* weights,
                                float* t_fine, hipStream_t stream) {
  if (n_rays <= 0) return hipSuccess;
  CompositeArgs a{raw, 4, raw + 3, 4, t_coarse, dirs, n_rays, 65, white_bkgd, act, comp_rgb, acc, depth, weights, u, u_stride, t_fine};
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

}  // namespace aon
