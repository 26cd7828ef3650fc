// Weight-gradient machinery shared by the vanilla (aon_train.hip) and articulated (aon_train_art.hip) backward passes.
#pragma once
#include "aon_mlp_core.h"
#include "aon_bf16_split.h"

namespace aon {

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------------
// weight gradients
// ---------------------------------------------------------------------------------------------
// One workgroup: OUT[M x K] partial = A[M rows x n-range] . B[K rows x n-range]^T, M = 128*RT, K = 32*CT.
// wave w owns rows [32*RT*w, 32*RT*(w+1)) x all K columns -> RT x CT accumulator tiles.
//
// Operand staging is LDS-DMA (global_load_lds, 16 bytes per lane), no vector registers and no ds_write spent on it:
// one DMA instruction moves 8 rows x 128 bytes (each row segment a full, coalesced 128-byte line).  The LDS image
// keeps rows at their natural 128-byte pitch; bank conflicts of the fragment reads (32 lanes reading the same 16-byte
// column of 32 different rows) are removed by an XOR swizzle applied on the GLOBAL side: the lane that fills 16-byte
// slot p of row r fetches column p ^ (r & 7) of that row, so column c of row r lives in slot c ^ (r & 7) and any 8
// consecutive rows hit 8 distinct slots.  (Round-1 measurements, tools/ubench/wgrad_layout.hip, 256x256 block,
// 790,528 samples: register-staged global_load + padded ds_write_b128 126 TFLOP/s; DMA gathering 32-byte pieces
// directly into fragment order 109; this form: see DESIGN.md.)
constexpr int kWgTileFloats = 32 * 32;  // one 32-row tile of one 32-sample step

struct WgradArgs {
  const float* A;  // dZ plane rows (row 0 of this block)
  const float* B;  // activation plane rows
  int64_t Np;
  int nchunks;     // Np / 32
  float* partial;  // [gridDim.x][M][K]
  float* bias_partial;  // [gridDim.x][M] or null
};

template <int RT, int CT>
__global__ void __launch_bounds__(256) wgrad_kernel(WgradArgs a) {
  constexpr int M = 128 * RT, K = 32 * CT;
  constexpr int NTILE = (M + K) / 32;           // 32-row tiles staged per 32-sample step
  constexpr int DMA_EVERY = CT < 4 ? CT : 4;   // one DMA instruction per DMA_EVERY MFMAs
  constexpr int STAGE_FLOATS = NTILE * kWgTileFloats;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;

  // contiguous range of 32-sample steps for this workgroup
  const int per = (a.nchunks + gridDim.x - 1) / gridDim.x;
  const int c_begin = blockIdx.x * per;
  const int c_end = c_begin + per < a.nchunks ? c_begin + per : a.nchunks;

  f32x16 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) bsum[i] = 0.f;

  // DMA lane map: lane = (r, p) = (lane >> 3, lane & 7) fills slot p of row r of an 8-row group with column p ^ r
  const int64_t lane_off = (int64_t)(lane >> 3) * a.Np * 4 + (((lane & 7) ^ (lane >> 3)) << 4);
  // one of this wave's NTILE DMA instructions of a step: 8-row group G = 4*d + wave
  auto dma_one = [&](int chunk, int buf, int d) {
    int64_t step_off = (int64_t)chunk * 128;
    asm volatile("" : "+s"(step_off));  // keep the per-group addresses on the scalar unit (no hoisted VGPR pairs)
    const int G = 4 * d + wave;
    const float* rows = 8 * G < M ? a.A + (int64_t)(8 * G) * a.Np : a.B + (int64_t)(8 * G - M) * a.Np;
    const char* g = reinterpret_cast<const char*>(rows) + step_off + lane_off;
    char* l = reinterpret_cast<char*>(smem + buf * STAGE_FLOATS + G * 256);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_void*)l, 16, 0, 0);
  };
  // fragment of k-step s for this lane: row li of a tile, column 2s + kh -> slot (2s + kh) ^ (li & 7)
  int frag_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) frag_off[s] = li * 32 + (((2 * s + kh) ^ (li & 7)) << 2);

  if (c_begin < c_end) {
#pragma unroll
    for (int d = 0; d < NTILE; ++d) dma_one(c_begin, 0, d);
  }
  __syncthreads();  // vmcnt(0) + barrier: step c_begin is in LDS for every wave
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    const bool more = c + 1 < c_end;
    const float* sa = smem + buf * STAGE_FLOATS + (RT * wave) * kWgTileFloats;
    const float* sb = smem + buf * STAGE_FLOATS + (M / 32) * kWgTileFloats;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 af[RT], bf[CT];
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        af[i] = *reinterpret_cast<const f32x4*>(sa + i * kWgTileFloats + frag_off[s]);
        bsum[i] += (af[i][0] + af[i][1]) + (af[i][2] + af[i][3]);
      }
#pragma unroll
      for (int j = 0; j < CT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(sb + j * kWgTileFloats + frag_off[s]);
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          // The next step's DMA instructions are issued one per CT MFMAs over the FIRST part of this step.  A wave issues
          // in order: a burst of all of them in front of the MFMAs stalls the matrix pipe while the memory pipeline
          // accepts them (-4 %); issued too late, the vmcnt(0) in front of the barrier waits out their HBM latency (-8 %).
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            if (j % DMA_EVERY == 0) {
              const int d = ((4 * s + cc) * RT + i) * ((CT + DMA_EVERY - 1) / DMA_EVERY) + j / DMA_EVERY;
              if (d < NTILE && more) dma_one(c + 1, buf ^ 1, d);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][cc], bf[j][cc], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();  // drains this wave's DMA of step c+1 and publishes it; everyone is done reading `buf`
  }
  // partial[wg][row][col]; accumulator layout: col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5)
  float* out = a.partial + (int64_t)blockIdx.x * M * K;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (RT * wave + i) + (r & 3) + 8 * (r >> 2) + 4 * kh;
        out[(int64_t)row * K + 32 * j + li] = acc[i][j][r];
      }
  if (a.bias_partial) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32);
      if (kh == 0) a.bias_partial[(int64_t)blockIdx.x * M + 32 * (RT * wave + i) + li] = v;
    }
  }
}

// Split-precision form of wgrad_kernel (opt-in "bf16x3" training engine): identical staging, partials and epilogue;
// the contraction runs on v_mfma_f32_32x32x16_bf16 with both operands split exactly into three bf16 limbs in registers
// (fragment = 8 consecutive samples of a row = two ds_read_b128) and the six limb products of weight >= 2^-24 accumulated
// in fp32 -- fp32-class error at 6 x 32 instead of 8 x 64 matrix-pipe cycles per (tile pair, 16 samples).  The B tile of
// column block j is split while the MFMAs of block j-1 run.
template <int RT, int CT>
__global__ void __launch_bounds__(256) wgrad_bf16x3_kernel(WgradArgs a) {
  constexpr int M = 128 * RT, K = 32 * CT;
  constexpr int NTILE = (M + K) / 32;
  constexpr int STAGE_FLOATS = NTILE * kWgTileFloats;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int per = (a.nchunks + gridDim.x - 1) / gridDim.x;
  const int c_begin = blockIdx.x * per;
  const int c_end = c_begin + per < a.nchunks ? c_begin + per : a.nchunks;

  f32x16 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) bsum[i] = 0.f;

  const int64_t lane_off = (int64_t)(lane >> 3) * a.Np * 4 + (((lane & 7) ^ (lane >> 3)) << 4);
  auto dma_one = [&](int chunk, int buf, int d) {
    int64_t step_off = (int64_t)chunk * 128;
    asm volatile("" : "+s"(step_off));
    const int G = 4 * d + wave;
    const float* rows = 8 * G < M ? a.A + (int64_t)(8 * G) * a.Np : a.B + (int64_t)(8 * G - M) * a.Np;
    const char* g = reinterpret_cast<const char*>(rows) + step_off + lane_off;
    char* l = reinterpret_cast<char*>(smem + buf * STAGE_FLOATS + G * 256);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_void*)l, 16, 0, 0);
  };
  // fragment of k16-step ks: row li, 16-byte columns 4ks + 2kh and 4ks + 2kh + 1 (8 consecutive samples), swizzled slots
  int frag_off[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int c = 0; c < 2; ++c) frag_off[ks][c] = li * 32 + (((4 * ks + 2 * kh + c) ^ (li & 7)) << 2);

  if (c_begin < c_end) {
#pragma unroll
    for (int d = 0; d < NTILE; ++d) dma_one(c_begin, 0, d);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    const bool more = c + 1 < c_end;
    const float* sa = smem + buf * STAGE_FLOATS + (RT * wave) * kWgTileFloats;
    const float* sb = smem + buf * STAGE_FLOATS + (M / 32) * kWgTileFloats;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Limb8 la[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(sa + i * kWgTileFloats + frag_off[ks][0]);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(sa + i * kWgTileFloats + frag_off[ks][1]);
        bsum[i] += ((r0[0] + r0[1]) + (r0[2] + r0[3])) + ((r1[0] + r1[1]) + (r1[2] + r1[3]));
        la[i] = split8(r0, r1);
      }
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(sb + j * kWgTileFloats + frag_off[ks][0]);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(sb + j * kWgTileFloats + frag_off[ks][1]);
        const Limb8 lb = split8(r0, r1);
        // the next step's DMA instructions, spread over the 2*CT column-block groups of this step (early ones first)
        constexpr int PER = (NTILE + 2 * CT - 1) / (2 * CT);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int d = (ks * CT + j) * PER + q;
          if (d < NTILE && more) dma_one(c + 1, buf ^ 1, d);
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i][j] = mfma_bf16x3(la[i], lb, acc[i][j]);
      }
    }
    __syncthreads();
  }
  float* out = a.partial + (int64_t)blockIdx.x * M * K;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (RT * wave + i) + (r & 3) + 8 * (r >> 2) + 4 * kh;
        out[(int64_t)row * K + 32 * j + li] = acc[i][j][r];
      }
  if (a.bias_partial) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32);
      if (kh == 0) a.bias_partial[(int64_t)blockIdx.x * M + 32 * (RT * wave + i) + li] = v;
    }
  }
}

// Second stage (deterministic, no atomics):
//   out[row*ld + col_off + col] = sum_wg partial[wg][row][col]   for col < k_valid      (blocks [0, M*K/256))
//   bias_out[row]               = sum_wg bias_partial[wg][row]                           (blocks [M*K/256, +M/16))
// Weight blocks: 64 float4 columns x 4 waves; wave w sums the partials p = w, w+4, ... with 8 independent 16-byte loads in
// flight per lane, then the four wave sums are added in wave order through LDS.  64 MB of partials per 256x256 layer are
// read at HBM/MALL speed instead of through one dependent 4-byte load chain per thread.
static __global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, int nparts, int M, int K, int k_valid,
                                                                  float* __restrict__ out, int ld, int col_off,
                                                                  const float* __restrict__ bias_partial, float* __restrict__ bias_out) {
  __shared__ f32x4 red[4][64];
  const int nb_w = M * K / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < nb_w) {
    const int idx = (blockIdx.x * 64 + lane) * 4;  // first of 4 consecutive columns of one row (K is a multiple of 32)
    const f32x4* src = reinterpret_cast<const f32x4*>(partial + idx);
    const int64_t stride4 = (int64_t)M * K / 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int pp = wave;
    for (; pp + 28 < nparts; pp += 32) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(pp + 4 * u) * stride4];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; pp < nparts; pp += 4) s += src[(int64_t)pp * stride4];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
      const f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
      const int row = idx / K, col = idx % K;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (col + c < k_valid) out[(int64_t)row * ld + col_off + col + c] = t[c];
    }
  } else if (bias_out) {
    // 16 rows per block; thread (r, g) sums partials g, g+16, ...; the 16 group sums are added in group order
    float* redf = reinterpret_cast<float*>(&red[0][0]);
    const int r = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int row = ((int)blockIdx.x - nb_w) * 16 + r;
    float s = 0.f;
    if (row < M)
      for (int pp = g; pp < nparts; pp += 16) s += bias_partial[(int64_t)pp * M + row];
    redf[g * 16 + r] = s;
    __syncthreads();
    if (g == 0 && row < M) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += redf[q * 16 + r];
      bias_out[row] = t;
    }
  }
}

// Heads: dW_sigma[k] = sum_n d_raw[n].w * H7[k][n];  dW_rgb[c][k] = sum_n d_raw[n][c] * HV[k][n];  d bias = sum_n d_raw[n]
// grid = (ceil(rows / 8), nseg); a block reduces EIGHT plane rows over one segment of samples -> partial[seg][row][4].  The
// 16-byte d_raw record of a sample is read once per eight rows (round 1 read it once per row: 16 of every 20 bytes the kernel
// moved were that re-read; 95 us per launch, 0.95 ms per articulated training step).
constexpr int kHeadRows = 8;
static __global__ void __launch_bounds__(256) head_wgrad_kernel(const float* __restrict__ plane, int64_t Np, const float* __restrict__ d_raw,
                                                         int64_t seg_len, float* __restrict__ partial, int rows) {
  const int row0 = blockIdx.x * kHeadRows, seg = blockIdx.y;
  const int nr = rows - row0 < kHeadRows ? rows - row0 : kHeadRows;
  const int64_t n0 = (int64_t)seg * seg_len;
  const int64_t n1 = n0 + seg_len < Np ? n0 + seg_len : Np;
  float s[kHeadRows][4];
#pragma unroll
  for (int r = 0; r < kHeadRows; ++r) s[r][0] = s[r][1] = s[r][2] = s[r][3] = 0.f;
  for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
    const float4 d = reinterpret_cast<const float4*>(d_raw)[n];
#pragma unroll
    for (int r = 0; r < kHeadRows; ++r) {
      const float x = plane ? (r < nr ? plane[(int64_t)(row0 + r) * Np + n] : 0.f) : 1.0f;  // plane == null: bias sums
      s[r][0] = __builtin_fmaf(x, d.x, s[r][0]); s[r][1] = __builtin_fmaf(x, d.y, s[r][1]);
      s[r][2] = __builtin_fmaf(x, d.z, s[r][2]); s[r][3] = __builtin_fmaf(x, d.w, s[r][3]);
    }
  }
  __shared__ float red[4][kHeadRows][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < kHeadRows; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v = wsum64(s[r][c]);
      if (lane == 0) red[wv][r][c] = v;
    }
  __syncthreads();
  if (threadIdx.x < kHeadRows * 4) {
    const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
    if (r < nr) partial[((int64_t)seg * rows + row0 + r) * 4 + c] = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
  }
}

// out[c*ld_out + row] (channel-major rows of a (C,rows) weight) = sum_seg partial[seg][row][chan_of(c)]
static __global__ void head_reduce_kernel(const float* __restrict__ partial, int nseg, int rows, int chan0, int nchan, float* __restrict__ out,
                                   int ld_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * nchan) return;
  const int row = idx / nchan, c = idx % nchan;
  float s = 0.f;
  for (int p = 0; p < nseg; ++p) s += partial[((int64_t)p * rows + row) * 4 + chan0 + c];
  out[(int64_t)c * ld_out + row] = s;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Training engine of the weight-gradient GEMMs: 0 = exact fp32 MFMA (default), 1 = split-bf16 ("bf16x3", fp32-equivalent
// products).  Process-wide (one process drives one GPU from one thread); set through aon_set_train_engine().
inline int& train_engine() {
  static int e = 0;
  return e;
}

template <int RT, int CT>
static hipError_t run_wgrad(const float* A, const float* B, int64_t Np, int nparts, float* partial, float* bias_partial,
                            float* out, int ld, int col_off, int k_valid, float* bias_out, hipStream_t stream) {
  constexpr int M = 128 * RT, K = 32 * CT;
  constexpr int lds = 2 * (M + K) * 32 * 4;
  static DeviceOnce once_f32, once_bf16;
  if (hipError_t e = set_max_lds(&wgrad_kernel<RT, CT>, lds, once_f32); e != hipSuccess) return e;
  if (hipError_t e = set_max_lds(&wgrad_bf16x3_kernel<RT, CT>, lds, once_bf16); e != hipSuccess) return e;
  WgradArgs a{A, B, Np, (int)(Np / 32), partial, bias_out ? bias_partial : nullptr};
  if (train_engine() == 1) wgrad_bf16x3_kernel<RT, CT><<<dim3(nparts), dim3(256), lds, stream>>>(a);
  else wgrad_kernel<RT, CT><<<dim3(nparts), dim3(256), lds, stream>>>(a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int nblocks = M * K / 256 + (bias_out ? (M + 15) / 16 : 0);
  wgrad_reduce_kernel<<<dim3(nblocks), dim3(256), 0, stream>>>(partial, nparts, M, K, k_valid, out, ld, col_off, bias_partial, bias_out);
  return hipGetLastError();
}

inline int64_t wgrad_workspace_bytes_impl() {
  // per-workgroup partial of the largest block (256 x 256) + bias partials + head partials, for up to 256 workgroups
  return (int64_t)256 * (256 * 256 + 256) * 4 + (int64_t)256 * 257 * 4 * 4 + 4096;
}


// workspace carve shared by both networks
struct WgradWs {
  float* partial;       // 256 x (256 x 256)
  float* bias_partial;  // 256 x 256
  float* head_partial;  // 256 segments x 257 rows x 4
};
inline WgradWs carve_wgrad_ws(float* ws) {
  WgradWs w;
  w.partial = ws;
  w.bias_partial = ws + (int64_t)256 * 256 * 256;
  w.head_partial = w.bias_partial + (int64_t)256 * 256;
  return w;
}

// rows x 4-channel reductions of a plane against a sample-major (Np,4) gradient buffer (heads, biases)
inline void head_segments(int64_t Np, int& nseg, int64_t& seg_len) {
  nseg = (int)((Np + 16383) / 16384);
  if (nseg > 256) nseg = 256;
  seg_len = ((Np + nseg - 1) / nseg + 3) / 4 * 4;
  nseg = (int)((Np + seg_len - 1) / seg_len);
}

}  // namespace aon
