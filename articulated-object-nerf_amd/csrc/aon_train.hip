// Backward pass of the vanilla render path for gfx950 (what autograd does implicitly for
// models/vanilla_nerf/model.py:264-273 loss.backward()).  Gradients reach only the MLP parameters: t_samples is
// detached (helper.py:249) and rays / t are data.
//
// Kernels, all in the register layout of the fused forward (lane = sample, registers = features) and the step-major planes of
// aon_mlp_core.h:
//   composite_bwd_kernel   d(comp_rgb, acc, depth) -> d(raw rgb, raw sigma) per sample, one wavefront per ray.
//   mlp_bwd_chain_kernel   the data-gradient chain rgb head -> view layer -> bottleneck (+sigma head) -> trunk 7..1,
//                          register-resident exactly like the forward: dH_{l-1}^T = W_l^T . dZ_l^T with the
//                          TRANSPOSED weights streamed as MFMA A operands and the gradient tiles as B operands;
//                          writes every layer's pre-activation gradient dZ_l to the gradient planes.
//   wgrad_grouped_kernel   (aon_wgrad.h) dW_l = dZ_l^T[M x N] . H_{l-1}[N x K] for every layer of a level in one launch: the
//                          sample range of each layer split over workgroups in proportion to its cost, each holding its
//                          output block in accumulator registers; a deterministic second stage sums the partials (no atomics).
//   head_wgrad_kernel      the 1- and 3-row head weights and all bias sums (plane rows x one 16-byte record per sample).
#include <atomic>
#define AON_WGRAD_KERNELS
#define AON_CHAIN_STAGE_MASKED   // see BwdSideOf (aon_mlp_core.h)
#include "aon_fold.h"
#include "aon_wgrad.h"

namespace aon {

// ---------------------------------------------------------------------------------------------
// composite backward   (helper.volumetric_rendering, helper.py:157-195, + the activations of model.py:186-187 /
// model_autodecoder.py:321-323)
// ---------------------------------------------------------------------------------------------
struct CompositeBwdArgs {
  const float* raw;     // (n*S,4) raw rgb, raw sigma (as written by the forward)
  const float* t_vals;  // (n,S)
  const float* dirs;    // (n,3)
  const float* g_rgb;   // (n,3) dL/d comp_rgb
  const float* g_acc;   // (n,) or null
  const float* g_depth; // (n,) or null
  int64_t n_rays; int S; int white_bkgd; ActParams ap;
  float* d_raw;         // (n*S,4) dL/d raw
};

// Round 4: the whole backward of the compositing is evaluated in fp64 on the fp32 inputs and rounded to fp32 ONCE per output.
// dL/dalpha_i = T_i gw_i - Sfx_i / f_i is a difference of nearly equal terms, and the density-head bias gradient is the plain sum of
// dL/d raw_sigma over every sample of a level -- on a field where that sum cancels to 1e-4 of its terms (constructor-fuzz seed 112:
// x6,368) the fp32 form's correlated rounding along each ray (both scans feed every sample in front of them) left the bias 3e-3 from
// the fp64 truth where torch's fp32 autograd happened to land at 9e-5 (tests/diag/diag_density_bias.py: per-sample errors 1.2x
// torch's, their per-ray sums 3-5x).  The kernel handles 20 B per sample and runs ~15 us per 4096-ray step: fp64 costs nothing
// visible, and every d_raw value is now the correctly rounded derivative of the reference's formula at the forward's fp32 inputs.
__device__ __forceinline__ double sigmoidd_(double x) { return 1.0 / (1.0 + exp(-x)); }

__device__ __forceinline__ double shfl_up_d(double v, int off) { return __shfl_up(v, off); }
__device__ __forceinline__ double shfl_down_d(double v, int off) { return __shfl_down(v, off); }

// NB: blocks of 64 samples held in registers -- 4 for the reference geometry (S <= 256), 8 for anything up to S = 512 (the training
// entry points stop at 512 samples per ray).  Per block and lane four doubles and three floats survive between the two scans.
template <int NB>
__global__ void __launch_bounds__(256) composite_bwd_kernel(CompositeBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.n_rays) return;
  const int S = a.S;
  const ActParams ap = a.ap;
  const int nblk = (S + 63) >> 6;  // <= NB
  const float* tv = a.t_vals + ray * S;
  // ||d|| as the forward takes it (fp32, helper.py:167): the interval lengths are INPUTS of the function being differentiated
  const float dn = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(a.dirs[ray * 3], a.dirs[ray * 3]),
                                                  __fmul_rn(a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 1])),
                                        __fmul_rn(a.dirs[ray * 3 + 2], a.dirs[ray * 3 + 2])));
  const double gC0 = a.g_rgb[ray * 3], gC1 = a.g_rgb[ray * 3 + 1], gC2 = a.g_rgb[ray * 3 + 2];
  const double gA = a.g_acc ? (double)a.g_acc[ray] : 0.0, gD = a.g_depth ? (double)a.g_depth[ray] : 0.0;
  // dL/dw_i = gC.c_i - [white] sum(gC) + g_acc + t_i g_depth     (comp_rgb += 1 - acc, helper.py:187-188)
  const double gw_const = gA - (a.white_bkgd ? (gC0 + gC1 + gC2) : 0.0);

  double tgw[NB], f[NB], kf[NB], wgw[NB];   // T_i gw_i;  1 - alpha_i + 1e-10;  d alpha_i / d raw_sigma_i;  w_i gw_i
  float o0[NB], o1[NB], o2[NB];             // dL/d raw rgb, finished in the first sweep
  double carry = 1.0;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    tgw[b] = 0.0; f[b] = 1.0; kf[b] = 0.0; wgw[b] = 0.0; o0[b] = o1[b] = o2[b] = 0.f;
    if (b < nblk) {
      const int s = b * 64 + lane;
      const bool in = s < S;
      double alpha = 0.0, gw = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;
      if (in) {
        const int64_t g = ray * S + s;
        const float t = tv[s];
        const double dist = (double)__fmul_rn(s == S - 1 ? 1e10f : __fsub_rn(tv[s + 1], t), dn);
        float4 r = reinterpret_cast<const float4*>(a.raw)[g];
        if (ap.noise) r.w = __fadd_rn(r.w, __fmul_rn(ap.noise[g], ap.noise_std));   // model.py:183-184 (d/d raw_sigma = 1)
        double sg, dsg, c0, c1, c2;
        if (ap.act == 1) {
          sg = r.w > 0.f ? (double)r.w : 0.0; dsg = r.w > 0.f ? 1.0 : 0.0;
          c0 = sigmoidd_(r.x); c1 = sigmoidd_(r.y); c2 = sigmoidd_(r.z);
          d0 = c0 * (1.0 - c0); d1 = c1 * (1.0 - c1); d2 = c2 * (1.0 - c2);
        } else if (ap.act == 2) {
          const float xs = __fadd_rn(r.w, ap.sigma_bias);   // (an fp32 add in the forward: part of the function)
          sg = xs > 20.0f ? (double)xs : log1p(exp((double)xs)); dsg = xs > 20.0f ? 1.0 : sigmoidd_(xs);   // torch Softplus, threshold 20
          const double s0 = sigmoidd_(r.x), s1 = sigmoidd_(r.y), s2 = sigmoidd_(r.z);
          const double sc = ap.rgb_scale, sh = ap.rgb_shift;
          c0 = s0 * sc - sh; c1 = s1 * sc - sh; c2 = s2 * sc - sh;
          d0 = sc * s0 * (1.0 - s0); d1 = sc * s1 * (1.0 - s1); d2 = sc * s2 * (1.0 - s2);
        } else {
          sg = r.w; dsg = 1.0; c0 = r.x; c1 = r.y; c2 = r.z; d0 = d1 = d2 = 1.0;
        }
        const double ex = exp(-sg * dist);
        alpha = 1.0 - ex;
        f[b] = (1.0 - alpha) + 1e-10;
        kf[b] = dist * ex * dsg;   // d alpha / d sigma = dist exp(-sigma dist), times the density activation's derivative
        gw = gC0 * c0 + gC1 * c1 + gC2 * c2 + gw_const + (double)t * gD;
      }
      // forward transmittance T_i = prod_{j<i} f_j
      double incl = f[b];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double o = shfl_up_d(incl, off);
        if (lane >= off) incl = incl * o;
      }
      double excl = shfl_up_d(incl, 1);
      if (lane == 0) excl = 1.0;
      const double T = carry * excl;
      carry = carry * __shfl(incl, 63);
      const double w = alpha * T;
      tgw[b] = T * gw;
      wgw[b] = in ? w * gw : 0.0;
      o0[b] = (float)(w * gC0 * d0); o1[b] = (float)(w * gC1 * d1); o2[b] = (float)(w * gC2 * d2);
    }
  }
  // suffix sums  Sfx_i = sum_{k>i} w_k gw_k, scanned from the far end
  double sfx_carry = 0.0;
#pragma unroll
  for (int b = NB - 1; b >= 0; --b) {
    if (b < nblk) {
      double incl = wgw[b];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double o = shfl_down_d(incl, off);
        if (lane + off < 64) incl = incl + o;
      }
      double excl = shfl_down_d(incl, 1);
      if (lane == 63) excl = 0.0;
      const double sfx = excl + sfx_carry;
      sfx_carry = sfx_carry + __shfl(incl, 0);
      const int s = b * 64 + lane;
      if (s < S) {
        // dL/dalpha_i = T_i gw_i - Sfx_i / (1 - alpha_i + 1e-10)
        const double dalpha = tgw[b] - sfx / f[b];
        float4 o;
        o.x = o0[b]; o.y = o1[b]; o.z = o2[b];
        o.w = (float)(dalpha * kf[b]);
        reinterpret_cast<float4*>(a.d_raw)[ray * S + s] = o;
      }
    }
  }
}

hipError_t launch_composite_bwd(const float* raw, const float* t_vals, const float* dirs, const float* g_rgb, const float* g_acc,
                                const float* g_depth, int64_t n_rays, int S, int white_bkgd, const ActParams& ap, float* d_raw,
                                hipStream_t stream) {
  if (n_rays <= 0) return hipSuccess;
  if (S > 512) return hipErrorInvalidValue;
  CompositeBwdArgs a{raw, t_vals, dirs, g_rgb, g_acc, g_depth, n_rays, S, white_bkgd, ap, d_raw};
  if (S <= 256) composite_bwd_kernel<4><<<dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, stream>>>(a);
  else composite_bwd_kernel<8><<<dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, stream>>>(a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// backward data chain
// ---------------------------------------------------------------------------------------------
// Transposed weight stream: 68 chunks of 32 KiB; chunk = 32 OUTPUT features j of a layer x all 256 INPUT features:
//   chunk[q][Tp][lane][c] = W[j = 32T + 8q + 4(lane>>5) + c][f = 32Tp + (lane&31)]
constexpr int kBwView = 0;   // views_linear.0  (128 x 283): 4 chunks (cols 0..255 -> d bottleneck)
constexpr int kBwBott = 4;   // bottleneck_layer: 8
constexpr int kBwL7 = 12;    // pts_linears.7 ... pts_linears.1: 8 each, in backward order
constexpr int kBwNumChunks = 68;

struct BwdNet {
  static constexpr int kSlotBytes = kPairSlotBytes;  // a slot holds a pair of chunks
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = kBwNumChunks;
  static constexpr int chunk_bytes(int) { return kBigChunkBytes; }
};
constexpr int64_t kBwStreamBytes = (int64_t)kBwNumChunks * kBigChunkBytes;

// FOLDED form (aon_common.h: W' = W_v0[:, :256] W_b): chunks 0..3 are W'^T -- d H7 = W'^T dZ_view (+ the density head's term) -- and the
// bottleneck's eight chunks are gone: 60 chunks.  Behind the stream the buffer keeps what the un-folding of the gradients needs
// (launch_unfold_view): W_v0[:, :256] compact (128 x 256), W_b (256 x 256), b_b (256); then W' / b' for the pack kernel.
constexpr int kBwFL7 = 4;
constexpr int kBwFNumChunks = 60;
struct BwdFoldNet {
  static constexpr int kSlotBytes = kPairSlotBytes;
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = kBwFNumChunks;
  static constexpr int chunk_bytes(int) { return kBigChunkBytes; }
};
constexpr int64_t kBwFStreamBytes = (int64_t)kBwFNumChunks * kBigChunkBytes;
constexpr int64_t kBwFOffWv = kBwFStreamBytes;                       // 128 x 256 floats
constexpr int64_t kBwFOffWb = kBwFOffWv + 128 * 256 * 4;             // 256 x 256
constexpr int64_t kBwFOffBb = kBwFOffWb + 256 * 256 * 4;             // 256
constexpr int64_t kBwFOffWf = kBwFOffBb + 256 * 4;                   // 128 x 256, then b' (128)
constexpr int64_t kBwFBytes = kBwFOffWf + (128 * 256 + 128) * 4;
constexpr int64_t kBwBufferBytes = kBwFBytes > kBwStreamBytes ? kBwFBytes : kBwStreamBytes;   // aon_bwd_packed_bytes(): either form fits

struct PackArgs24 {
  const float* p[kNumVanillaParams];
};

// P / V: widths of the network's encodings (63 / 27 by default; other degrees: only the row strides of the two concatenating layers
// change -- the chain never needs the encoding columns, gradients do not reach the inputs)
template <bool FOLD>
__global__ void pack_vanilla_bwd_kernel(PackArgs24 a, float* __restrict__ packed, int P, int V) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (FOLD) {   // raw copies for launch_unfold_view, behind the stream
    const int64_t i = idx - kBwFOffWv / 4;
    if (i >= 0) {
      if (i < 128 * 256) packed[idx] = a.p[16][(i >> 8) * (256 + V) + (i & 255)];
      else if (i < 128 * 256 + 256 * 256) packed[idx] = a.p[18][i - 128 * 256];
      else if (i < 128 * 256 + 256 * 256 + 256) packed[idx] = a.p[19][i - 128 * 256 - 256 * 256];
      return;
    }
  }
  if (idx >= (FOLD ? kBwFStreamBytes : kBwStreamBytes) / 4) return;
  int c = (int)(idx / (kBigChunkBytes / 4));
  const int r = (int)(idx % (kBigChunkBytes / 4));
  const int cc = r & 3, lane = (r >> 2) & 63, rest = r >> 8;
  const int tp = rest & 7, q = rest >> 3;
  const int h = lane >> 5, f = 32 * tp + (lane & 31);
  const int jo = 8 * q + 4 * h + cc;
  const float* W; int ld, j;
  if constexpr (FOLD) { if (c >= kBwFL7) c += kBwL7 - kBwFL7; }   // the trunk's chunks take the literal branches below
  if (c < kBwBott) {
    W = a.p[16]; ld = 256 + V; j = 32 * c + jo;
    if constexpr (FOLD) { W = packed + kBwFOffWf / 4; ld = 256; }   // W' (launch_fold_view, same stream, in front of this kernel)
  }
  else if (c < kBwL7) { W = a.p[18]; ld = 256; j = 32 * (c - kBwBott) + jo; }
  else {
    const int l = 7 - (c - kBwL7) / 8;  // 7,6,5,4,3,2,1
    W = a.p[2 * l]; ld = l == 5 ? 256 + P : 256; j = 32 * ((c - kBwL7) % 8) + jo;
  }
  packed[idx] = W[(int64_t)j * ld + f];
}

// One SEGMENT of a chain launch = the passes of one level (see ArtBwdSeg, aon_train_art.hip: the two levels' chains are one launch).
struct BwdSeg {
  const char* packed_bwd;   // kBwStreamBytes
  const float* small;       // forward small block (head weights): packed_fwd + kStreamBytes
  const float* d_raw;       // (Np,4)   zero for padded samples
  const u32x4* masks;       // forward ReLU bit masks, kMaskLayers x (Np*2)
  float* dplanes;           // pre-activation gradient planes, same row map
  int64_t Np;
  int npass;
};
struct BwdArgs {
  BwdSeg seg[2];
  int npass_total;          // seg[1].npass == 0: a one-segment launch
};

template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&x)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = 0.f;
}

// FOLD: the transposed stream is the folded form (BwdFoldNet): d H7 = W'^T dZ_view + W_sigma^T d_sigma in one layer, no bottleneck gradient.
template <bool FOLD>
__global__ void __launch_bounds__(256) mlp_bwd_chain_kernel(BwdArgs args) {
  using Net = std::conditional_t<FOLD, BwdFoldNet, BwdNet>;
  constexpr int kL7 = FOLD ? kBwFL7 : kBwL7;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem + kRingBytes);
  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63, wave = tid0 >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int npass0 = args.seg[0].npass;
  int cur = (int)blockIdx.x >= npass0 ? 1 : 0;
  auto load_small = [&](const float* small) {
    const f32x4* src = reinterpret_cast<const f32x4*>(small);
    f32x4* dst = reinterpret_cast<f32x4*>(sm);
    for (int i = threadIdx.x; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
  };
  load_small(args.seg[cur].small);
  Pipe p;
  pipe_init<Net>(p, args.seg[cur].packed_bwd, smem, wave, lane);  // also publishes the small block just written to LDS

  for (int gpass = blockIdx.x; gpass < args.npass_total; gpass += gridDim.x) {
    const int si = gpass >= npass0 ? 1 : 0;
    if (si != cur) {   // (workgroup-uniform, at most once per launch) the other level's head weights replace the resident block
      __syncthreads();
      load_small(args.seg[si].small);
      __syncthreads();
      cur = si;
    }
    const BwdSeg& sg = args.seg[si];
    const int pass = gpass - (si ? npass0 : 0);
    {
      const int nxt = gpass + (int)gridDim.x;
      p.stream = sg.packed_bwd;
      p.next_stream = args.seg[(nxt >= npass0 && nxt < args.npass_total) ? 1 : si].packed_bwd;
    }
    // lane coordinates re-derived once per pass (v_mbcnt + the wave index in an SGPR) instead of kept across the pass loop, as in
    // art_bwd_chain_kernel: with the segment bookkeeping added in round 4 the loop-invariant per-lane values were otherwise spilled
    int lane_p;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
    const int tid = (wave_s << 6) | lane_p;
    const int m = lane_p & 31, h = lane_p >> 5;
    const int64_t col = (int64_t)pass * 128 + wave_s * 32 + m;
    const PlaneIO io = make_plane_io(sg.dplanes, kPlRows, (int64_t)pass * 4 + wave_s, m, h);
    // The 16-byte decision-bit word of a layer is fetched ONE layer ahead of its use (round 1 fetched all nine up front:
    // 36 registers held through the whole pass).  The offset is made opaque at the point of use so the load stays there.
    const unsigned moff = mask_lane_off(pass, tid);
    auto load_mask = [&](int layer) { return *mask_ptr(sg.masks, sg.Np, layer, moff); };
    const float4 dr = reinterpret_cast<const float4*>(sg.d_raw)[col];
    // half-wave index as the LDS reads below see it: opaque per pass, otherwise every head-weight address (the small block
    // sits beyond the 64 KiB immediate-offset range of the ring) is hoisted out of the pass loop into its own register
    int hl = h;
    asm volatile("" : "+v"(hl));
    u32x4 mk = load_mask(8), mk_next;
    // rgb head (model.py:118):  dHV[f] = sum_c W_rgb[c][f] * d_rgb[c]
    f32x16 Z[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int fo = 32 * t + 8 * gq + 4 * hl;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + kSmWRgb + 0 * kCondWidth + fo);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + kSmWRgb + 1 * kCondWidth + fo);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(sm + kSmWRgb + 2 * kCondWidth + fo);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          Z[t][4 * gq + cc] = __builtin_fmaf(w2[cc], dr.z, __builtin_fmaf(w1[cc], dr.y, w0[cc] * dr.x));
      }
    }
    f32x16 X[8], Y[8];
    mk_next = load_mask(7);
    apply_mask_tile(Z[0], mk, 0);
    auto sigma_head_into = [&](f32x16 (&T)[8]) {   // W_sigma^T * d_sigma: the density head reads the post-ReLU layer-7 output (model.py:105)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(sm + kSmWSigma + 32 * t + 8 * gq + 4 * hl);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) T[t][4 * gq + cc] = w[cc] * dr.w;
        }
      }
    };
    if constexpr (FOLD) {
      // dH7 = W'^T . dZ_view + W_sigma^T * d_sigma, dZ_view = view-layer ReLU mask . dHV   (W' = W_v0[:, :256] W_b: model.py:109-116 as one layer)
      // (the view layer first, from zero, the density head's term added behind it: initialising Y in front of the layer cost 24 B/lane of scratch)
      dense_layer<Net, kBwView, 4, 8, BwdSideOf<4, true>, true>(p, Z, Y, BwdSideOf<4, true>{Z, kPlHV, io, mk, &mk_next});
#pragma unroll
      for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(sm + kSmWSigma + 32 * t + 8 * gq + 4 * hl);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) Y[t][4 * gq + cc] = __builtin_fmaf(w[cc], dr.w, Y[t][4 * gq + cc]);
        }
      }
    } else {
      // d bottleneck = W_view[:, :256]^T . dZ_view, dZ_view = view-layer ReLU mask . dHV   (model.py:109-116)
      dense_layer<Net, kBwView, 4, 8, BwdSideOf<4, true>, true>(p, Z, X, BwdSideOf<4, true>{Z, kPlHV, io, mk, &mk_next});   // X starts from zero
      // dH7 = W_bott^T . dBot + W_sigma^T * d_sigma   (the bottleneck has no activation)
      sigma_head_into(Y);
      dense_layer<Net, kBwBott, 8, 8>(p, X, Y, BwdSideOf<8, false>{X, kPlBot, io, mk});
    }
    // trunk: dZ_l = mask_l . dH_l (stored by the chunks that consume it), dH_{l-1} = W_l^T . dZ_l
#define AON_BWD_LAYER(IN, OUT, CB, L)                                                                                  \
    mk = mk_next; if (L > 0) mk_next = load_mask(L - 1);                                                               \
    apply_mask_tile(IN[0], mk, 0);                                                                                     \
    dense_layer<Net, CB, 8, 8, BwdSideOf<8, true>, true>(p, IN, OUT, BwdSideOf<8, true>{IN, plane_h(L), io, mk, L > 0 ? &mk_next : nullptr});   /* OUT starts from zero; the next layer's bits are waited for two chunks in (BwdSideOf::touch) */
    AON_BWD_LAYER(Y, X, kL7 + 0, 7)
    AON_BWD_LAYER(X, Y, kL7 + 8, 6)
    AON_BWD_LAYER(Y, X, kL7 + 16, 5)
    AON_BWD_LAYER(X, Y, kL7 + 24, 4)
    AON_BWD_LAYER(Y, X, kL7 + 32, 3)
    AON_BWD_LAYER(X, Y, kL7 + 40, 2)
    AON_BWD_LAYER(Y, X, kL7 + 48, 1)
#undef AON_BWD_LAYER
    // dZ0: no data gradient flows into the encoding, so no chunk consumes it -- masked and stored here (128 values)
    apply_mask_bits(X, mk_next);
    store_plane(X, io, plane_h(0));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
int num_cus();  // aon_mlp.hip

// The form packed is the process default at the time of the call (aon_set_bottleneck_fold), remembered for `packed` (stream_form).
void vanilla_fold_jobs_bwd(const float* const* params, float* packed, int view_size, FoldGemm jobs[2]) {
  float* Wf = packed + kBwFOffWf / 4;
  fold_view_jobs(params[16], 256 + view_size, params[17], params[18], params[19], Wf, Wf + 128 * 256, jobs);
}

hipError_t launch_pack_vanilla_bwd(const float* const* params, float* packed, hipStream_t stream, int pos_size, int view_size, bool fold_done) {
  PackArgs24 a;
  for (int i = 0; i < kNumVanillaParams; ++i) a.p[i] = params[i];
  const int form = fold_default();
  set_stream_form(packed, form);
  if (form == kFormFolded) {
    if (!fold_done) {
      FoldGemm jobs[2];
      vanilla_fold_jobs_bwd(params, packed, view_size, jobs);
      if (hipError_t e = launch_fold_gemms(jobs, 2, stream); e != hipSuccess) return e;
    }
    const int64_t n = kBwFOffWf / 4;   // the stream and the raw copies behind it
    pack_vanilla_bwd_kernel<true><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed, pos_size, view_size);
  } else {
    const int64_t n = kBwStreamBytes / 4;
    pack_vanilla_bwd_kernel<false><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed, pos_size, view_size);
  }
  return hipGetLastError();
}

int64_t bwd_stream_bytes() { return kBwBufferBytes; }

template <bool FOLD>
static hipError_t launch_chain_f(const BwdArgs& a, int grid, hipStream_t stream) {
  static DeviceOnce lds_once;
  constexpr int lds = kRingBytes + (int)kSmallBytes;
  if (hipError_t e = set_max_lds(&mlp_bwd_chain_kernel<FOLD>, lds, lds_once); e != hipSuccess) return e;
  mlp_bwd_chain_kernel<FOLD><<<dim3(grid), dim3(256), lds, stream>>>(a);
  return hipGetLastError();
}

hipError_t launch_mlp_bwd_chain2(const ChainSeg* segs, int nsegs, hipStream_t stream) {
  if (nsegs < 1 || nsegs > 2) return hipErrorInvalidValue;
  const int form = stream_form(segs[0].packed_bwd);
  if (form == kFormUnknown) return hipErrorInvalidValue;   // never packed / declared (a copy): refuse instead of guessing
  if (nsegs == 2 && stream_form(segs[1].packed_bwd) != form) return hipErrorInvalidValue;
  BwdArgs a{};
  for (int i = 0; i < nsegs; ++i) {
    const ChainSeg& c = segs[i];
    a.seg[i] = BwdSeg{c.packed_bwd, c.small, c.d_raw, static_cast<const u32x4*>(c.masks), c.dplanes, c.Np, (int)(c.Np / 128)};
    a.npass_total += a.seg[i].npass;
  }
  if (nsegs == 1) { a.seg[1] = a.seg[0]; a.seg[1].npass = 0; }
  else if (a.seg[0].npass == 0) { a.seg[0] = a.seg[1]; a.seg[1].npass = 0; }
  const int cus = num_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  const int grid = a.npass_total < cus ? a.npass_total : cus;
  if (grid <= 0) return hipSuccess;
  return form == kFormFolded ? launch_chain_f<true>(a, grid, stream) : launch_chain_f<false>(a, grid, stream);
}

hipError_t launch_mlp_bwd_chain(const char* packed_bwd, const char* packed_fwd, const float* d_raw, const void* masks,
                                float* dplanes, int64_t Np, hipStream_t stream) {
  const ChainSeg one{packed_bwd, reinterpret_cast<const float*>(packed_fwd + kStreamBytes), d_raw, masks, nullptr, dplanes, nullptr, Np};
  return launch_mlp_bwd_chain2(&one, 1, stream);
}

int64_t wgrad_workspace_bytes() { return wgrad_workspace_bytes_impl(); }

// measurement aid: when set, every grouped weight-gradient launch writes its workgroups' entry / exit clocks (100 MHz) there
static std::atomic<long long*> g_wgrad_probe{nullptr};
void set_wgrad_probe(long long* buf) { g_wgrad_probe.store(buf, std::memory_order_relaxed); }

// Launch sequence shared by both networks: grouped weight gradients, heads, ONE second stage for both.
// phase / n_early (round 5): the first n_early head jobs read only what exists BEFORE the backward chain runs (forward planes, d_raw).
//   kWgAll    everything, as rounds 3-4;
//   kWgEarly  only those head reductions -- the fused backward launches them on a side stream beside the chain, whose last, quarter-full
//             round of workgroups (8,256 passes on 256 CUs) leaves three quarters of the chip idle for one pass;
//   kWgRest   the remaining head jobs, the grouped weight gradients and the second stage (which sums the early partials too).
// The plan (workspace offsets of every partial) is a pure function of the arguments: the two phases of a level compute the same one.
hipError_t run_wgrad_plan(const WgLayerDesc* layers, int nlayers, const HeadDesc* heads, int nheads, const HeadOut* outs, const int* out_head, int nouts,
                          const float* planes, const float* dplanes, int rows_total, int64_t Np, float* ws, hipStream_t stream, const WgAux* aux,
                          int phase, int n_early, const WgPost* post, hipStream_t* post_stream, ReduceArgs* defer_reduce, int* defer_blocks) {
  if (post_stream) *post_stream = stream;
  static DeviceOnce lds_once;
  if (hipError_t e = set_max_lds(&wgrad_grouped_kernel, kWgLdsBytes, lds_once); e != hipSuccess) return e;
  const int cus = num_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  if (Np <= 0 || (Np & 31)) return hipErrorInvalidValue;
  WgPlan plan;
  if (!wg_make_plan(layers, nlayers, planes, dplanes, rows_total, Np, cus < 304 ? cus : 304, ws, 0, plan)) return hipErrorInvalidValue;
  int64_t off = plan.ws_floats;
  HeadArgs H{};
  int part_offs[kHeadMaxJobs];
  if (nheads > kHeadMaxJobs || nouts > kHeadMaxOut) return hipErrorInvalidValue;
  const int head_blocks = head_make_plan(heads, nheads, rows_total, Np, ws, off, H, part_offs);
  if (off * 4 > wgrad_workspace_bytes_impl() - (1 << 20)) return hipErrorInvalidValue;   // (the last MiB: slot-layout blocks of the articulated network at other degrees)
  int nseg; int64_t seg_len;
  head_segments(Np, nseg, seg_len);
  ReduceArgs& R = plan.red;
  R.nhead = nouts; R.nseg = nseg; R.head_blk_begin = plan.reduce_blocks;
  for (int o = 0; o < nouts; ++o) { R.head[o] = outs[o]; R.head[o].part_off = part_offs[out_head[o]]; }
  // the heads go first, on the side stream when there is one (their workgroups are resident before the weight-gradient launch
  // fills every compute unit) and run beside it; the second stage waits for both
  if (n_early < 0 || n_early > nheads) return hipErrorInvalidValue;
  const int early_blocks = n_early == nheads ? head_blocks : H.job[n_early].blk_begin;
  if (phase == kWgEarly) {
    if (early_blocks > 0) head_wgrad_kernel<<<dim3(early_blocks, nseg), dim3(256), 0, stream>>>(H);
    return hipGetLastError();
  }
  const int first_block = phase == kWgRest ? early_blocks : 0;
  H.blk_offset = first_block;
  hipStream_t hs = aux ? aux->stream : stream;
  if (aux) {
    if (hipError_t e = hipEventRecord(aux->fork, stream); e != hipSuccess) return e;
    if (hipError_t e = hipStreamWaitEvent(hs, aux->fork, 0); e != hipSuccess) return e;
  }
  if (head_blocks > first_block) head_wgrad_kernel<<<dim3(head_blocks - first_block, nseg), dim3(256), 0, hs>>>(H);
  hipError_t err = hipGetLastError();
  if (aux) {
    const hipError_t e = hipEventRecord(aux->join, hs);
    if (err == hipSuccess) err = e;
  }
  if (err == hipSuccess) {
    plan.args.probe = g_wgrad_probe.load(std::memory_order_relaxed);
    wgrad_grouped_kernel<<<dim3(plan.total_wgs), dim3(256), kWgLdsBytes, stream>>>(plan.args);
    err = hipGetLastError();
  }
  if (aux) {   // joined whatever happened above: the side stream never runs past the caller's view of this call
    const hipError_t e = hipStreamWaitEvent(stream, aux->join, 0);
    if (err == hipSuccess) err = e;
  }
  if (err != hipSuccess) return err;
  if (defer_reduce) {   // the caller launches this level's second stage itself, together with the other level's (launch_wgrad_reduce2)
    if (post || !defer_blocks) return hipErrorInvalidValue;
    *defer_reduce = R;
    *defer_blocks = plan.reduce_blocks + nouts;
    return hipSuccess;
  }
  hipStream_t ps = stream;
  if (post) {
    if (post->wait_first)
      if (hipError_t e = hipStreamWaitEvent(stream, post->wait_first, 0); e != hipSuccess) return e;
    if (post->side) {   // the second stage and everything behind it on the side stream, ordered behind the grouped kernel
      if (hipError_t e = hipEventRecord(post->side->fork, stream); e != hipSuccess) return e;
      if (hipError_t e = hipStreamWaitEvent(post->side->stream, post->side->fork, 0); e != hipSuccess) return e;
      ps = post->side->stream;
    }
  }
  if (post_stream) *post_stream = ps;
  wgrad_reduce_kernel<<<dim3(plan.reduce_blocks + nouts), dim3(256), 0, ps>>>(R);
  return hipGetLastError();
}

// The second stages of two levels in one launch: each block does what it does in wgrad_reduce_kernel (same sums, same order).
hipError_t launch_wgrad_reduce2(const ReduceArgs& a0, int n0, const ReduceArgs& a1, int n1, hipStream_t stream) {
  if (n0 < 1 || n1 < 1) return hipErrorInvalidValue;
  wgrad_reduce2_kernel<<<dim3(n0 + n1), dim3(256), 0, stream>>>(a0, a1, n0);
  return hipGetLastError();
}

// Measurement aid (tools/kernel_bench.py --wgrad-kinds): `nlayers` identical jobs of one kind on arbitrary plane rows, the
// grouped kernel only -- the isolated rate of a kind with the whole chip running it (what wg_cost() is tuned from).
hipError_t launch_wgrad_kind_bench(int kind, int nlayers, const float* planes, const float* dplanes, int rows_total, int64_t Np, float* ws,
                                   float* out_scratch, hipStream_t stream) {
  static DeviceOnce lds_once;
  if (hipError_t e = set_max_lds(&wgrad_grouped_kernel, kWgLdsBytes, lds_once); e != hipSuccess) return e;
  const int cus = num_cus();
  if (cus <= 0 || kind < 0 || kind >= kWgNumKinds || nlayers < 1 || nlayers > kWgMaxJobs) return hipErrorInvalidValue;
  WgLayerDesc L[kWgMaxJobs];
  const int M = wg_M(kind), K = wg_K(kind);
  for (int l = 0; l < nlayers; ++l) {
    const int a_row = (l * 256) % (rows_total - M + 1) / 32 * 32, b_row = (l * 256 + 128) % (rows_total - K + 1) / 32 * 32;
    L[l] = WgLayerDesc{kind, a_row, b_row, out_scratch, K, 0, K, nullptr};
  }
  WgPlan plan;
  if (!wg_make_plan(L, nlayers, planes, dplanes, rows_total, Np, cus < 304 ? cus : 304, ws, 0, plan)) return hipErrorInvalidValue;
  if (plan.ws_floats * 4 > wgrad_workspace_bytes_impl()) return hipErrorInvalidValue;
  wgrad_grouped_kernel<<<dim3(plan.total_wgs), dim3(256), kWgLdsBytes, stream>>>(plan.args);
  return hipGetLastError();
}

// grads: 24 device pointers in the parameter order of aon_pack_vanilla_mlp (each the full (out,in) / (out,) tensor), overwritten.
// the weight-gradient jobs of one vanilla level
// fold_tmp != null: the planes are the folded form's (no bottleneck rows).  The bottleneck and the view layer's hidden columns are ONE job
// dW' = dZ_view . H7^T (128 x 256) into fold_tmp, db' = db_v0 straight into views_linear.0.bias; launch_unfold_view turns (dW', db') into
// the gradients of bottleneck_layer and views_linear.0[:, :256].
int vanilla_wgrad_layers(float* const* grads, WgLayerDesc* L, float* fold_tmp) {
  int n = 0;
  // trunk: dW_l = dZ_l . H_{l-1}^T  (+ the pos-enc columns for layers 0 and 5)
  L[n++] = WgLayerDesc{kWg256x64, plane_h(0), kPlE, grads[0], kPosEnc, 0, kPosEnc, grads[1]};
  for (int l = 1; l < 8; ++l) {
    const int ld = l == 5 ? 256 + kPosEnc : 256;
    L[n++] = WgLayerDesc{kWg256x256, plane_h(l), plane_h(l - 1), grads[2 * l], ld, 0, 256, grads[2 * l + 1]};
    if (l == 5) L[n++] = WgLayerDesc{kWg256x64, plane_h(5), kPlE, grads[10], ld, 256, kPosEnc, nullptr};
  }
  if (fold_tmp) {
    L[n++] = WgLayerDesc{kWg128x256, kPlHV, plane_h(7), fold_tmp, 256, 0, 256, grads[17]};
  } else {
    // bottleneck (input: post-ReLU layer-7 output)
    L[n++] = WgLayerDesc{kWg256x256, kPlBot, plane_h(7), grads[18], 256, 0, 256, grads[19]};
    // view layer: cat[bottleneck(256), viewenc(27)]: two column blocks of one weight
    L[n++] = WgLayerDesc{kWg128x256, kPlHV, kPlBot, grads[16], 256 + kViewEnc, 0, 256, grads[17]};
  }
  L[n++] = WgLayerDesc{kWg128x32, kPlHV, kPlVE, grads[16], 256 + kViewEnc, 256, kViewEnc, nullptr};
  return n;
}
int vanilla_wgrad_layers(float* const* grads, WgLayerDesc* L) { return vanilla_wgrad_layers(grads, L, nullptr); }

// dW' (128 x 256) of a folded level: the last MiB of the weight-gradient workspace, which no plan reaches (run_wgrad_plan refuses plans beyond it;
// the articulated network's slot-layout blocks at other degrees take its first 144 KiB: art_wgrad_layers)
float* wgrad_fold_tmp(float* ws) { return ws + (wgrad_workspace_bytes_impl() - (1 << 20)) / 4 + 48 * 1024; }   // 192 KiB into that MiB

// packed_bwd: the transposed stream the chain of these planes ran with -- its FORM says whether the planes carry bottleneck rows, and the
// folded form's buffer holds the raw W_v0[:, :256], W_b, b_b the un-folding needs.  Null: literal planes.
// Round 6: a vanilla level's second stage handed back instead of launched (the two levels' grouped kernels then run back to back and ONE
// reduce launch + ONE launch of the un-folding products serve both: launch_vanilla_wgrad_post2); see ArtWgDeferred in aon_train_art.hip.
struct VanillaWgDeferred {
  ReduceArgs reduce;
  int reduce_blocks;
  int n_unfold;           // 3 (folded form) or 0
  FoldGemm unfold[3];
};
constexpr int kVanillaWgDeferredBytes = 4096;   // (aon_capi.hip keeps two of these on its stack)
static_assert(sizeof(VanillaWgDeferred) <= kVanillaWgDeferredBytes && alignof(VanillaWgDeferred) <= 16, "VanillaWgDeferred outgrew its storage in aon_capi.hip");
int vanilla_wgrad_deferred_bytes() { return (int)sizeof(VanillaWgDeferred); }   // (aon_capi.hip checks its storage against this)

hipError_t launch_vanilla_wgrad(const float* planes, const float* dplanes, const float* d_raw, int64_t Np, float* const* grads,
                                float* ws, hipStream_t stream, const WgAux* aux, const void* packed_bwd, int phase, const WgPost* post,
                                VanillaWgDeferred* defer) {
  WgLayerDesc L[kWgMaxJobs];
  if (defer && (phase == kWgEarly || post)) return hipErrorInvalidValue;
  if (packed_bwd && stream_form(packed_bwd) == kFormUnknown) return hipErrorInvalidValue;   // (a copy nobody declared)
  const bool fold = packed_bwd && stream_form(packed_bwd) == kFormFolded;
  float* fold_tmp = fold ? wgrad_fold_tmp(ws) : nullptr;
  const int n = vanilla_wgrad_layers(grads, L, fold_tmp);
  // heads and their biases: density_layer (1,256) <- H7 x d_raw.w, rgb_layer (3,128) <- HV x d_raw.xyz, bias sums of d_raw
  const HeadDesc H[3] = {{planes, plane_h(7), 256, d_raw, 128}, {planes, kPlHV, 128, d_raw, 128}, {nullptr, 0, 1, d_raw, 128}};
  const HeadOut O[4] = {{0, 256, 3, 1, 256, 1, grads[20]}, {0, 128, 0, 3, 128, 1, grads[22]}, {0, 1, 3, 1, 1, 1, grads[21]}, {0, 1, 0, 3, 1, 1, grads[23]}};
  const int OH[4] = {0, 1, 2, 2};
  // all three head jobs (density head on H7, rgb head on HV, the sums of d_raw) are independent of the chain: n_early = 3
  if (hipError_t e = run_wgrad_plan(L, n, H, 3, O, OH, 4, planes, dplanes, kPlRows, Np, ws, stream, aux, phase, 3, phase == kWgEarly ? nullptr : post, nullptr,
                                    defer ? &defer->reduce : nullptr, defer ? &defer->reduce_blocks : nullptr); e != hipSuccess) return e;
  if (defer) defer->n_unfold = 0;
  if (!fold || phase == kWgEarly) return hipSuccess;
  const float* raw = reinterpret_cast<const float*>(static_cast<const char*>(packed_bwd) + kBwFOffWv);
  // (grads[16]'s row stride is the 27-slot layout's here: other view degrees write a slot-layout temporary first, aon_render_bwd_ex)
  if (defer) {
    defer->n_unfold = 3;
    unfold_view_jobs(fold_tmp, grads[17], raw, 256, raw + 128 * 256, raw + 128 * 256 + 256 * 256, grads[16], 256 + kViewEnc, grads[18], grads[19], defer->unfold);
    return hipSuccess;
  }
  return launch_unfold_view(fold_tmp, grads[17], raw, 256, raw + 128 * 256, raw + 128 * 256 + 256 * 256, grads[16], 256 + kViewEnc, grads[18], grads[19], stream);
}

// the deferred second stages of two vanilla levels: one reduce launch, one launch of the (up to) six un-folding products; every block of
// either does what it does in the per-level launches (same bits)
hipError_t launch_vanilla_wgrad_post2(const VanillaWgDeferred* d0, const VanillaWgDeferred* d1, hipStream_t stream) {
  if (hipError_t e = launch_wgrad_reduce2(d0->reduce, d0->reduce_blocks, d1->reduce, d1->reduce_blocks, stream); e != hipSuccess) return e;
  FoldGemm jobs[6];
  int n = 0;
  for (const VanillaWgDeferred* d : {d0, d1})
    for (int j = 0; j < d->n_unfold; ++j) jobs[n++] = d->unfold[j];
  return n > 0 ? launch_fold_gemms(jobs, n, stream) : hipSuccess;
}


// Host-only view of the plan a level would run on `cus` compute units (tests/test_abi_cpu.py checks its invariants without a
// GPU): per job (kind, first workgroup, workgroups that own a step of it, steps of the job, partial offset in floats, partial count);
// `line`: {W, G} of the work line.
int art_wgrad_layers(float* const* grads, WgLayerDesc* L, int Lp, int Lv, float* enc_tmp, float* fold_tmp);   // aon_train_art.hip

// (the host-only plan views describe the plan of the process's current default form)
static int plan_layers(bool art, float* const* grads, WgLayerDesc* L) {
  float* fold_tmp = fold_default() == kFormFolded ? reinterpret_cast<float*>((uintptr_t)0x100000) : nullptr;   // never dereferenced
  return art ? art_wgrad_layers(grads, L, 10, 4, nullptr, fold_tmp) : vanilla_wgrad_layers(grads, L, fold_tmp);
}

int wgrad_plan_describe(bool art, int64_t Np, int cus, int32_t* out6, int max_jobs, int64_t* ws_bytes) {
  if (Np <= 0 || (Np & 31) || cus < 1) return -1;
  float* grads[40];   // >= both parameter counts (24 vanilla, 40 articulated)
  for (int i = 0; i < 40; ++i) grads[i] = reinterpret_cast<float*>((uintptr_t)0x1000 + 64 * i);   // never dereferenced
  WgLayerDesc L[kWgMaxJobs];
  const int n = plan_layers(art, grads, L);
  WgPlan plan;
  if (!wg_make_plan(L, n, nullptr, nullptr, art ? kAPlRows : kPlRows, Np, cus < 304 ? cus : 304, nullptr, 0, plan)) return -2;
  if (n > max_jobs) return -3;
  for (int j = 0; j < n; ++j) {
    const WgJob& J = plan.args.job[j];
    int32_t* o = out6 + 6 * j;
    o[0] = J.kind; o[1] = J.first_wg; o[2] = J.last_wg - J.first_wg + 1; o[3] = plan.args.nsteps; o[4] = J.part_off;
    o[5] = (J.last_wg - J.first_wg + 1) * wg_nsplit(J.kind);
  }
  if (ws_bytes) *ws_bytes = plan.ws_floats * 4;
  return n;
}

// the steps [begin, end) of job j that workgroup wg owns under the plan above (host restatement of the kernel's arithmetic: tests)
int wgrad_plan_segment(bool art, int64_t Np, int cus, int j, int wg, int32_t* begin_end) {
  if (Np <= 0 || (Np & 31) || cus < 1) return -1;
  float* grads[40];
  for (int i = 0; i < 40; ++i) grads[i] = reinterpret_cast<float*>((uintptr_t)0x1000 + 64 * i);
  WgLayerDesc L[kWgMaxJobs];
  const int n = plan_layers(art, grads, L);
  WgPlan plan;
  if (!wg_make_plan(L, n, nullptr, nullptr, art ? kAPlRows : kPlRows, Np, cus < 304 ? cus : 304, nullptr, 0, plan)) return -2;
  if (j < 0 || j >= n || wg < 0 || wg >= plan.total_wgs) return -3;
  const WgArgs& A = plan.args;
  const WgJob& J = A.job[j];
  auto first_step = [&](int64_t x) {
    const int64_t d = x - J.p_begin;
    if (d <= 0) return 0;
    const int64_t q = (d + J.cost - 1) / J.cost;
    return q < A.nsteps ? (int)q : A.nsteps;
  };
  begin_end[0] = first_step(A.w_total * wg / A.nwgs); begin_end[1] = first_step(A.w_total * (wg + 1) / A.nwgs);
  return (wg >= J.first_wg && wg <= J.last_wg) ? 1 : 0;
}

}  // namespace aon
