// Shared constants and small device helpers for the gfx950 (MI355X / CDNA4) NeRF render kernels.
// Wave size is 64 on CDNA; every wave-width constant below is hard-coded to 64.
#pragma once
// Timing-experiment switches (AON_EXP_*: kernels with a piece removed, WRONG results -- tools/exp_*.sh build them as side libraries) must never
// reach a product build: they are honoured only together with AON_EXPERIMENT_BUILD.
#if (defined(AON_EXP_GEMM_NOBARRIER) || defined(AON_EXP_GEMM_NOSTORE) || defined(AON_EXP_L0E_SIDE) || defined(AON_EXP_MASK_VGPR) || defined(AON_EXP_NOBARRIER) || defined(AON_EXP_NOBIAS) || defined(AON_EXP_NOENC) || defined(AON_EXP_NOFINAL) || defined(AON_EXP_NOMASK) || defined(AON_EXP_NOPARK) || defined(AON_EXP_NORELU) || defined(AON_EXP_NOSTORE) || defined(AON_EXP_NOSTORE4) || defined(AON_EXP_NOSTORE8) || defined(AON_EXP_NOSTORE_L0E) || defined(AON_EXP_NOSTORE_L5) || defined(AON_EXP_NOVMWAIT) || defined(AON_EXP_NO_NT) || defined(AON_EXP_STORE_OF) || defined(AON_EXP_WG_NOBAR) || defined(AON_EXP_WG_NOVM)) && !defined(AON_EXPERIMENT_BUILD)
#error "an AON_EXP_* timing-experiment switch is defined without AON_EXPERIMENT_BUILD: these kernels compute wrong results"
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace aon {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// fp32(0.5*pi): the reference computes cos(x) as sin(x + 0.5*np.pi) with the add done in fp32
// (models/vanilla_nerf/helper.py:139).
#define AON_HALF_PI_F32 1.57079637050628662109375f

// sin(x) for |x| < ~1e5, max abs error 9.3e-8 against fp64 (|x| <= 3200): three-term Cody-Waite reduction by
// multiples of pi/2 with FMA, then degree-7/8 minimax polynomials on [-pi/4, pi/4].  Branch-free (the library
// sinf carries a Payne-Hanek slow path that costs registers and divergence and is never taken here: the
// largest argument on this path is 2^9 * 6 + pi/2).
__device__ __forceinline__ float sin_f32(float x) {
  const float ax = __builtin_fabsf(x);
  const float k = __builtin_rintf(ax * 0.636619747f);
  const int q = (int)k;
  float r = __builtin_fmaf(k, -0x1.921fb4p+0f, ax);   // pi/2 = 0x3fc90fda + 0x33a22168 + 0x27c234c4
  r = __builtin_fmaf(k, -0x1.4442d0p-24f, r);
  r = __builtin_fmaf(k, -0x1.846988p-48f, r);
  const float z = r * r;
  float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
  const float s = __builtin_fmaf(r * z, ps, r);
  float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
  const float c = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
  float res = (q & 1) ? c : s;
  res = (q & 2) ? -res : res;
  return x < 0.f ? -res : res;
}

// cos(x) with the same reduction / polynomials (the derivative of the encoding in the backward pass).
__device__ __forceinline__ float cos_f32(float x) {
  const float ax = __builtin_fabsf(x);
  const float k = __builtin_rintf(ax * 0.636619747f);
  const int q = (int)k;
  float r = __builtin_fmaf(k, -0x1.921fb4p+0f, ax);
  r = __builtin_fmaf(k, -0x1.4442d0p-24f, r);
  r = __builtin_fmaf(k, -0x1.846988p-48f, r);
  const float z = r * r;
  float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
  const float s = __builtin_fmaf(r * z, ps, r);
  float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
  const float c = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
  // cos(|x|) = cos(r + q pi/2): q=0: c, 1: -s, 2: -c, 3: s
  float res = (q & 1) ? s : c;
  return ((q + 1) & 2) ? -res : res;
}

// ------------------------------------------------------------------------------------------------
// Packed-weight stream of one vanilla NeRFMLP (models/vanilla_nerf/model.py:39-120).
//
// A *chunk* is the slice of one layer's weight matrix that multiplies one 32-feature input tile:
//   chunk[q][Tp][lane][c]  (q=0..3, Tp=0..NT_OUT-1, lane=0..63, c=0..3)  fp32, i.e. NT_OUT*4 KiB
//     = W[32*Tp + (lane&31)][ col(tile, 8q + 4*(lane>>5) + c) ]
// which is exactly the A operand of v_mfma_f32_32x32x2_f32 for output tile Tp when the B operand is
// accumulator register r = 4q+c of the producing layer's tile (rows (r&3)+8*(r>>2)+4*(lane>>5)).
// ------------------------------------------------------------------------------------------------
constexpr int kNetWidth = 256;
constexpr int kCondWidth = 128;
constexpr int kPosEnc = 63;    // 3 + 2*10*3
constexpr int kViewEnc = 27;   // 3 + 2*4*3

constexpr int kChL0 = 0;       // 2 chunks  (pos-enc tiles)            -> 256
constexpr int kChL1 = 2;       // 8 chunks each for L1..L4
constexpr int kChL5 = 34;      // 8 hidden + 2 pos-enc (skip concat)
constexpr int kChL6 = 44;
constexpr int kChL7 = 52;
constexpr int kChBott = 60;    // bottleneck 256 -> 256 (no activation)
constexpr int kChView = 68;    // 8 hidden + 1 view-enc               -> 128
constexpr int kNumChunks = 77;
constexpr int kNumBigChunks = 68;            // chunks with 8 output tiles (32 KiB)
constexpr int kBigChunkBytes = 8 * 4096;
constexpr int kSmallChunkBytes = 4 * 4096;   // view layer: 4 output tiles (16 KiB)

__host__ __device__ constexpr int chunk_bytes(int c) { return c < kNumBigChunks ? kBigChunkBytes : kSmallChunkBytes; }
__host__ __device__ constexpr int64_t chunk_offset(int c) {
  return c < kNumBigChunks ? (int64_t)c * kBigChunkBytes
                           : (int64_t)kNumBigChunks * kBigChunkBytes + (int64_t)(c - kNumBigChunks) * kSmallChunkBytes;
}
constexpr int64_t kStreamBytes = chunk_offset(kNumChunks);  // 2,375,680 B

// FOLDED form of the stream (round 5).  bottleneck_layer has NO activation and feeds views_linear[0] directly (model.py:109-114), so
//   W_v0[:, :256] (W_b h + b_b) + W_v0[:, 256:] ve + b_v0  =  (W_v0[:, :256] W_b) h + W_v0[:, 256:] ve + (W_v0[:, :256] b_b + b_v0):
// ONE 256 -> 128 layer W' = W_v0[:, :256] W_b where the literal form runs 256 -> 256 then 256 -> 128 -- 65,536 of the network's 593,408
// MACs per sample (11.0 %).  W' and b' are evaluated in fp64 from the fp32 parameters at pack time and rounded once
// (aon_fold.hip).  Chunks 0 .. 59 are the literal stream's; then the view-encoding chunk and 8 small chunks of W'.  The buffer keeps the
// literal size: the small block stays at kStreamBytes, and the 256 KiB between the end of the folded stream and the small block hold
// W' (128 x 256) and b' (128) for the pack kernel.  Which form a buffer holds is remembered per pointer (stream_form, aon_fold.hip).
// The view-encoding chunk comes FIRST in the view layer: b' + W_v0[:, 256:] ve, the head of every accumulation chain, is a constant of
// the RAY, and the whole-path calls hand it to the kernel as a per-ray bias (view_bias_kernel: the same fused multiply-adds in the same
// order, so both forms give the same bits) instead of 56 MFMAs, 12 sines and a 16 KiB chunk per 128-sample pass.
constexpr int kChFView = 60;        // 1 view-enc + 8 hidden (W'), 4 output tiles each
constexpr int kNumChunksF = 69;
__host__ __device__ constexpr int chunk_bytes_f(int c) { return c < kChFView ? kBigChunkBytes : kSmallChunkBytes; }
__host__ __device__ constexpr int64_t chunk_offset_f(int c) {
  return c < kChFView ? (int64_t)c * kBigChunkBytes : (int64_t)kChFView * kBigChunkBytes + (int64_t)(c - kChFView) * kSmallChunkBytes;
}
constexpr int64_t kStreamBytesF = chunk_offset_f(kNumChunksF);   // 2,113,536 B
constexpr int64_t kFoldTmpOff = kStreamBytesF;                   // W' (128 x 256 floats) then b' (128 floats), inside the literal-size buffer
static_assert(kStreamBytesF + (128 * 256 + 128) * 4 <= kStreamBytes, "fold temporaries fit between the folded stream and the small block");

// Small per-layer vectors, kept resident in LDS for the whole kernel (floats):
constexpr int kSmBias = 0;                    // 8 x 256  trunk biases
constexpr int kSmBiasBott = 8 * 256;          // 256
constexpr int kSmBiasView = kSmBiasBott + 256;  // 128
constexpr int kSmWSigma = kSmBiasView + 128;  // 256
constexpr int kSmWRgb = kSmWSigma + 256;      // 3 x 128
constexpr int kSmBSigma = kSmWRgb + 384;      // 1
constexpr int kSmBRgb = kSmBSigma + 1;        // 3
constexpr int kSmallFloats = 3076;            // padded to a multiple of 4
constexpr int64_t kSmallBytes = kSmallFloats * 4;

constexpr int64_t kPackedBytes = kStreamBytes + kSmallBytes;  // one packed MLP

// Training: ROW MAP of the activation planes of one vanilla NeRFMLP level (the memory layout of a row is step-major, see
// aon_mlp_core.h: [step of 32 samples][row / 4][sample][row % 4]; Np = samples padded to a multiple of 128).  Rows are TRUE
// feature indices (encodings: the reference's column order), so weight gradients come out in nn.Linear's (out,in) order
// without un-permuting.  The same row map is used for the pre-activation gradient planes written by the backward chain.
constexpr int kPlE = 0;                        // 64 rows: pos-enc (63 + zero pad)            input of L0 / L5 skip
__host__ __device__ constexpr int plane_h(int l) { return 64 + 256 * l; }  // 8 x 256 rows: trunk outputs (post-ReLU)
constexpr int kPlBot = 64 + 8 * 256;           // 256 rows: bottleneck output (no activation)
constexpr int kPlVE = kPlBot + 256;            // 32 rows: view-enc (27 + zero pad)
constexpr int kPlHV = kPlVE + 32;              // 128 rows: view-layer output (post-ReLU)
constexpr int kPlRows = kPlHV + 128;           // 2528 rows = 10,112 B per sample
constexpr int kMaskLayers = 9;                 // ReLU bit masks: trunk 0..7, view layer; [layer][Np*2] x 16 B

// Articulated network (model_autodecoder.py): plane rows and mask slots of one level
constexpr int kAPlPos = 0;                                   // 32 rows: 0..2 = sample position x, 3..5 = deformed position x'
__host__ __device__ constexpr int aplane_d(int l) { return 32 + 128 * l; }    // 4 x 128: deformation MLP outputs
constexpr int kAPlE = 32 + 4 * 128;                          // 64 rows: pos-enc of x'
__host__ __device__ constexpr int aplane_h(int l) { return kAPlE + 64 + 256 * l; }  // 8 x 256: trunk outputs
constexpr int kAPlBot = kAPlE + 64 + 8 * 256;                // 256
constexpr int kAPlVE = kAPlBot + 256;                        // 32
__host__ __device__ constexpr int aplane_v(int l) { return kAPlVE + 32 + 128 * l; }  // 4 x 128: view-branch outputs
constexpr int kAPlRows = kAPlVE + 32 + 4 * 128;              // 3456 rows = 13,824 B per sample
constexpr int kAMaskLayers = 16;                             // slots: D0..3 -> 0..3, H0..7 -> 4..11, V0..3 -> 12..15

// Parameter order of the `params` pointer array handed to aon_pack_vanilla_mlp (device pointers to the
// unmodified torch nn.Linear storages, (out,in) row-major fp32):
//   0..15  pts_linears.{0..7}.{weight,bias}
//   16,17  views_linear.0.{weight,bias}
//   18,19  bottleneck_layer.{weight,bias}
//   20,21  density_layer.{weight,bias}
//   22,23  rgb_layer.{weight,bias}
constexpr int kNumVanillaParams = 24;

// Output activations of a level (model.py:183-187, model_autodecoder.py:318-323) with the constructor's scalars as the reference's
// fp32 tensor arithmetic sees them, shared by the compositing kernel and its backward:
//   raw_sigma <- raw_sigma + noise * noise_std        when `noise` is given (the caller's torch.rand_like draw; model.py:183-184)
//   act 1: rgb = sigmoid(raw), sigma = relu(raw_sigma)
//   act 2: rgb = sigmoid(raw) * rgb_scale - rgb_shift  (rgb_scale = fp32(1 + 2 rgb_padding), rgb_shift = fp32(rgb_padding)),
//          sigma = softplus(raw_sigma + sigma_bias)     (sigma_bias = fp32(density_bias))
struct ActParams {
  int act;
  float rgb_scale, rgb_shift, sigma_bias;
  const float* noise;   // (n*S,) or null
  float noise_std;
};
inline ActParams default_act(int act) { return ActParams{act, 1.002f, 0.001f, -1.0f, nullptr, 0.f}; }

// One segment of a backward-chain launch (host side): a level's transposed stream, head weights and buffers.  The chains of the two
// levels of a training step are independent: launch_*_bwd_chain2 runs them as one persistent launch of two segments.
struct ChainSeg {
  const char* packed_bwd;
  const float* small;      // the level's small block (vanilla: packed_fwd + kStreamBytes)
  const float* d_raw;      // (Np,4)
  const void* masks;
  const float* planes;     // articulated: the level's forward planes (deformed position rows); vanilla: null
  float* dplanes;
  float* dxp;              // articulated: (Np,4); vanilla: null
  int64_t Np;
};

// One segment of a training-forward launch (host side): a network's streams and one ray range's buffers at one level.  The fused
// training forwards take one or two of these per launch (launch_*_fwd_train2): e.g. the fine level of ray range A together with
// the coarse level of ray range B, so the partial last round of workgroups of one is filled with passes of the other.
struct TrainSeg {
  const char* packed;      // the level's packed forward stream
  const float* small;      // articulated: the level's per-call small block; vanilla: null (it follows the stream)
  const float* rays_o; const float* rays_d; const float* viewdirs;   // of the ray range
  const float* t_vals;     // (n_rays, S)
  int64_t n_rays; int S;
  float* raw;              // (n_rays * S, 4)
  float* planes; void* masks;   // the range's offsets into the level's planes / decision bits
  int64_t np_total;        // padded samples of the WHOLE level (slot stride of the decision bits); 0: this range alone
  const float* view_bias = nullptr;   // vanilla, folded form: (n_rays, 128) per-ray bias of the view layer (launch_view_bias), or null
};

// ------------------------------------------------------------------------------------------------
// Host side.  A kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-DEVICE function
// attribute and the CU count is a per-device property: both are remembered per device ordinal, lock-free (a racing
// second thread merely repeats an idempotent call), so a process that drives cuda:0 and later cuda:1 launches correctly
// on both.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxDevices = 64;

struct DeviceOnce {
  std::atomic<uint64_t> done{0};
};

template <class Kernel>
inline hipError_t set_max_lds(Kernel* kernel, int bytes, DeviceOnce& once) {
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
  const uint64_t bit = 1ull << dev;
  if (once.done.load(std::memory_order_acquire) & bit) return hipSuccess;
  if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      e != hipSuccess)
    return e;
  once.done.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

}  // namespace aon
