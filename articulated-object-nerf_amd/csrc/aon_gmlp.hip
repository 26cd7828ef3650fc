// General-geometry NeRFMLP engine for gfx950: every NeRFMLP(min_deg_point, max_deg_point, deg_view, netdepth, netwidth,
// netdepth_condition, netwidth_condition, skip_layer, ...) the reference's constructor accepts (models/vanilla_nerf/model.py:40-93),
// forward (model.py:95-120) and backward, as LAYER-WISE fp32 MFMA GEMMs on the unmodified nn.Linear storages.
//
// The reference's default geometry has the fused register-resident kernels (aon_mlp.hip, aon_train.hip); they are compiled for that
// geometry and nothing else.  This file is the path for everything else: activations live in HBM between layers (an fp32 layer is
// still matrix-pipe-bound on this chip: 2 x K x N flops against (K + N) x 4 bytes per sample = 128 flop/B at 256 x 256, the
// machine balance of the fp32 matrix pipe over HBM is ~20), one launch per layer:
//   gemm_tn_kernel   Y[M x N] = epi(sum_seg X_seg[M x K_seg] . W_seg[N x K_seg]^T + bias)        forward layers, backward data
//                    up to two (X, W) segments: torch.cat([x, inputs]) (model.py:103-104) and cat([bottleneck, condition_tile])
//                    (:111) are never materialised -- the second segment reads the encoding / the per-RAY condition row (row / S);
//                    epilogue: none | ReLU | mask by aux > 0 (dZ = dH . [H > 0]: the saved post-ReLU output IS the mask)
//   wgrad_nk_kernel  dW[N x K] += dZ[m0:m1, N]^T . X[m0:m1, K]    split over samples, one partial per split, no atomics
//   colsum_kernel    db[N] partials;   reduce_kernel: partials -> gradient in a fixed order;   transpose_kernel: W -> W^T
// 128 x 128 output tiles, 4 waves x (2 x 2) 32x32 MFMA tiles (v_mfma_f32_32x32x2_f32: exact fp32 like the fused kernels),
// operands staged through LDS with the next tile's global loads in flight in registers.
#include "aon_gmlp.h"

namespace aon {

constexpr int kGBM = 128, kGBN = 128, kGBK = 32, kGLD = 36;   // LDS rows of 32 k-values padded to 36 floats

// one 128-row x 32-k tile of a k-contiguous operand into registers.  Thread t takes the 16-byte piece (t & 7) of rows (t >> 3) + 32 q,
// q = 0..3: eight lanes cover the 128 contiguous bytes of one row, a wave-instruction touches 8 cache lines (the first version
// gave a thread 64 contiguous bytes of ONE row -- 32 lines per instruction, each a quarter used -- and the kernel sat at 0.51 of the
// matrix peak waiting on its own operand fetch).  The row address of q = 0 is formed once per workgroup.
struct RowSrc {
  const float* p;      // row (t >> 3) of the tile (or the base for a row outside the matrix: never dereferenced)
  int64_t row, nrows, ld;
  const float* base;
  int rowdiv;
  bool vec;
};
__device__ __forceinline__ RowSrc make_row_src(const float* base, int64_t ld, int rowdiv, int64_t row0, int64_t nrows, int tid) {
  RowSrc r;
  r.base = base; r.ld = ld; r.rowdiv = rowdiv; r.nrows = nrows;
  r.row = row0 + (tid >> 3);
  r.p = base + (r.row < nrows ? (rowdiv == 1 ? r.row : r.row / rowdiv) * ld : 0);
  r.vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
  return r;
}
__device__ __forceinline__ void load_tile_rows(const RowSrc& r, int k0, int K, int tid, f32x4 (&v)[4]) {
  const int k = k0 + (tid & 7) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = r.row + 32 * q;
    const bool ok = row < r.nrows;
    const float* p = r.rowdiv == 1 ? r.p + (int64_t)(32 * q) * r.ld : r.base + (ok ? (row / r.rowdiv) * r.ld : 0);
    if (ok && r.vec && k + 4 <= K) {
      v[q] = *reinterpret_cast<const f32x4*>(p + k);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[q][e] = (ok && k + e < K) ? p[k + e] : 0.f;
    }
  }
}

__device__ __forceinline__ void store_tile_rows(float* lds, int tid, const f32x4 (&v)[4]) {
  float* p = lds + (tid >> 3) * kGLD + (tid & 7) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(p + 32 * q * kGLD) = v[q];
}

__global__ void __launch_bounds__(256) gemm_tn_kernel(GemmArgs a) {
  // (A two-stage LDS variant with ONE barrier per k-chunk -- 74 KB, two workgroups per CU instead of three -- was measured slower:
  // 0.452 against 0.510 of the matrix peak, profiles/r03_general_engine_pmc.txt: this kernel lives on its occupancy.)
  __shared__ __attribute__((aligned(16))) float As[1][kGBM * kGLD];
  __shared__ __attribute__((aligned(16))) float Bs[1][kGBN * kGLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware tile order: workgroup L lands on XCD L % 8 (round-robin dispatch), each XCD has its own L2.  The column tiles of
  // one row tile read the same 128 rows of X: they get consecutive slots of the SAME XCD, so the second read hits that L2.
  const int tiles_n = (a.N + kGBN - 1) / kGBN;
  const int64_t total = ((a.M + kGBM - 1) / kGBM) * tiles_n;
  const int64_t per_xcd = (total + 7) / 8;
  const int64_t q = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (q >= total) return;
  const int64_t m0 = (q / tiles_n) * kGBM;
  const int n0 = (int)(q % tiles_n) * kGBN;
  const int wr = (wave & 1) * 64, wc = (wave >> 1) * 64;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // flat list of k-chunks over the segments
  const int nch0 = (a.seg[0].K + kGBK - 1) / kGBK;
  const int nch1 = a.nseg > 1 ? (a.seg[1].K + kGBK - 1) / kGBK : 0;
  const int nch = nch0 + nch1;
  f32x4 va[4], vb[4];
  RowSrc xa[2], wb[2];
#pragma unroll
  for (int sg = 0; sg < 2; ++sg) {
    if (sg < a.nseg) {
      xa[sg] = make_row_src(a.seg[sg].X, a.seg[sg].ldx, a.seg[sg].rowdiv, m0, a.M, tid);
      wb[sg] = make_row_src(a.seg[sg].W, a.seg[sg].ldw, 1, n0, a.N, tid);
    } else {
      xa[sg] = xa[0]; wb[sg] = wb[0];
    }
  }
  auto fetch = [&](int c) {
    const bool second = c >= nch0;
    const int k0 = (second ? c - nch0 : c) * kGBK;
    const int K = second ? a.seg[1].K : a.seg[0].K;
    load_tile_rows(second ? xa[1] : xa[0], k0, K, tid, va);
    load_tile_rows(second ? wb[1] : wb[0], k0, K, tid, vb);
  };
  auto compute = [&](const float* Ab, const float* Bb) {
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + (wr + 32 * i + r) * kGLD + kg * 8 + 4 * h);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + (wc + 32 * j + r) * kGLD + kg * 8 + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
  };
  fetch(0);
  for (int c = 0; c < nch; ++c) {
#ifndef AON_EXP_GEMM_NOBARRIER     // timing experiment only (WRONG results)
    __syncthreads();                 // the previous chunk's fragment reads are done
#endif
    store_tile_rows(As[0], tid, va);
    store_tile_rows(Bs[0], tid, vb);
#ifndef AON_EXP_GEMM_NOBARRIER
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    if (c + 1 < nch) fetch(c + 1);   // in flight under the MFMAs below
    compute(As[0], Bs[0]);
  }
#ifdef AON_EXP_GEMM_NOSTORE
  float exp_sum = 0.f;
#endif
  // epilogue: lane holds column n = .. + r, rows 8 (e >> 2) + (e & 3) + 4 h of each 32 x 32 tile
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc + 32 * j + r;
    if (n >= a.N) continue;
    const float b = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t mbase = m0 + wr + 32 * i + 4 * h;
      float* yrow = a.Y + mbase * a.ldy + n;
      const float* arow = a.epi == 2 ? a.aux + mbase * a.ldaux + n : nullptr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int dm = 8 * (e >> 2) + (e & 3);
        if (mbase + dm >= a.M) continue;
        float y = __fadd_rn(acc[i][j][e], b);
        if (a.epi == 1) y = __builtin_fmaxf(y, 0.f);
        else if (a.epi == 2) y = arow[dm * a.ldaux] > 0.f ? y : 0.f;
#ifdef AON_EXP_GEMM_NOSTORE        // timing experiment only (WRONG results): the 64 values summed into ONE store per lane
        exp_sum += y;
        if (e == 15 && i == 1 && j == 1) yrow[dm * a.ldy] = exp_sum;
#else
        yrow[dm * a.ldy] = y;
#endif
      }
    }
  }
}

hipError_t launch_gemm_tn(const GemmArgs& a, hipStream_t stream) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  const int64_t total = ((a.M + kGBM - 1) / kGBM) * ((a.N + kGBN - 1) / kGBN);
  gemm_tn_kernel<<<dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), 0, stream>>>(a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// weight gradients: part[split][N x K] = A[m0:m1, 0:N]^T . B[m0:m1, 0:K]   (A = dZ, B = the layer's input segment)
// ---------------------------------------------------------------------------------------------
constexpr int kWLD = 132;   // LDS rows of 128 features padded to 132 floats
constexpr int kWBM = 32;    // samples per staged step (64: 54.9 ms per training step against 54.1)

struct WgradArgs {
  const float* A; int64_t lda;               // (M, N)
  const float* B; int64_t ldb; int rowdiv;   // row m reads B[(m / rowdiv) * ldb + k]
  int64_t M; int N, K;
  int64_t rows_per_split;                    // multiple of kWBM
  float* part;                               // [splits][N * K]
};

// 32 rows x 128 columns: thread t takes row t >> 3 and the four 16-byte pieces at columns (t & 7) * 4 + 32 q -- eight lanes cover
// 128 contiguous bytes of a row per instruction, in HBM and in LDS (the first version gave a thread 16 consecutive columns: lanes
// 64 bytes apart, SQ_LDS_BANK_CONFLICT = half of the LDS cycles)
__device__ __forceinline__ void load_tile_cols(const float* base, int64_t ld, int rowdiv, int64_t m0, int64_t m_end, int c0, int C, int tid,
                                               f32x4 (&v)[kWBM / 8]) {
  const bool vec_base = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
#pragma unroll
  for (int rd = 0; rd < kWBM / 32; ++rd) {
    const int64_t m = m0 + 32 * rd + (tid >> 3);
    const bool row_ok = m < m_end;
    const float* p = base + (row_ok ? (rowdiv == 1 ? m : m / rowdiv) * ld : 0);
    const bool vec_ok = row_ok && vec_base;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + (tid & 7) * 4 + 32 * q;
      if (vec_ok && c + 4 <= C) {
        v[4 * rd + q] = *reinterpret_cast<const f32x4*>(p + c);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * rd + q][e] = (row_ok && c + e < C) ? p[c + e] : 0.f;
      }
    }
  }
}

// One workgroup = one 128 x 128 tile of dW over a range of samples; each of its four waves accumulates the WHOLE tile (16 MFMA
// tiles = 256 accumulator registers, one wave per SIMD) over a quarter of every staged 32-sample step and writes its own partial.
// Lane (r, h) reads ONE 16-byte piece per operand and sample pair -- features n0 + 4 r .. + 3 of sample 2 p + h -- and uses its four
// values as the A (or B) operand of FOUR tiles: tile e holds the features n0 + 4 i + e, so one A read and one B read feed 16 MFMAs
// (the fused weight-gradient kernel's fragment scheme, csrc/aon_wgrad.h).  The first version read one float per operand and MFMA
// (64 x 64 per wave): 64 LDS reads per 64 MFMAs, 0.44 of the matrix pipe.
__global__ void __launch_bounds__(256) wgrad_nk_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float As[kWBM * kWLD];
  __shared__ __attribute__((aligned(16))) float Bs[kWBM * kWLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
  const int64_t mb = (int64_t)blockIdx.z * a.rows_per_split;
  const int64_t me = mb + a.rows_per_split < a.M ? mb + a.rows_per_split : a.M;
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 va[kWBM / 8], vb[kWBM / 8];
  if (mb < me) {
    load_tile_cols(a.A, a.lda, 1, mb, me, n0, a.N, tid, va);
    load_tile_cols(a.B, a.ldb, a.rowdiv, mb, me, k0, a.K, tid, vb);
  }
  for (int64_t m = mb; m < me; m += kWBM) {
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < kWBM / 32; ++rd) {
      float* pa = As + (32 * rd + (tid >> 3)) * kWLD + (tid & 7) * 4;
      float* pb = Bs + (32 * rd + (tid >> 3)) * kWLD + (tid & 7) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) { *reinterpret_cast<f32x4*>(pa + 32 * q) = va[4 * rd + q]; *reinterpret_cast<f32x4*>(pb + 32 * q) = vb[4 * rd + q]; }
    }
    __syncthreads();
    if (m + kWBM < me) {
      load_tile_cols(a.A, a.lda, 1, m + kWBM, me, n0, a.N, tid, va);
      load_tile_cols(a.B, a.ldb, a.rowdiv, m + kWBM, me, k0, a.K, tid, vb);
    }
#pragma unroll
    for (int p = 0; p < kWBM / 8; ++p) {           // this wave's samples (kWBM / 4) w + 2 p + h of the step
      const int ml = (kWBM / 4) * wave + 2 * p + h;
      const f32x4 fa = *reinterpret_cast<const f32x4*>(As + ml * kWLD + 4 * r);
      const f32x4 fb = *reinterpret_cast<const f32x4*>(Bs + ml * kWLD + 4 * r);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  // this wave's partial: tile (i, j), register x, lane (r, h) holds dW[n0 + 4 (8 (x >> 2) + (x & 3) + 4 h) + i][k0 + 4 r + j]
  float* out = a.part + ((int64_t)blockIdx.z * 4 + wave) * a.N * a.K;
  const int kc = k0 + 4 * r;
  const bool vec = (a.K & 3) == 0 && kc + 4 <= a.K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int n = n0 + 4 * (8 * (x >> 2) + (x & 3) + 4 * h) + i;
      if (n >= a.N) continue;
      float* o = out + (int64_t)n * a.K + kc;
      if (vec) {
        f32x4 v; v[0] = acc[i][0][x]; v[1] = acc[i][1][x]; v[2] = acc[i][2][x]; v[3] = acc[i][3][x];
        *reinterpret_cast<f32x4*>(o) = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (kc + j < a.K) o[j] = acc[i][j][x];
      }
    }
  }
}

// column sums of A[m0:m1, 0:N] -> part[split][N] (doubles).  Block = 64 columns x 4 row lanes: a wave reads 256 contiguous bytes of one
// row, the four waves take rows m, m+1, m+2, m+3 (fixed association: lane sums in row order, then the four lanes in order).
// Bias gradients are signed sums over every sample (the head biases: pure cancellation), so the running sums, the partials and the
// second stage are fp64 and the result is rounded to float once (the kernel is HBM-bound; rounds 1-3 summed in fp32).
// (The first version gave a thread a column and the whole row range: 2.6 ms per call, half of the layer-wise training step.)
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int N, int64_t rows_per_split,
                                                     double* __restrict__ part) {
  __shared__ double red[4][64];
  const int c = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const int64_t mb = (int64_t)blockIdx.y * rows_per_split;
  const int64_t me = mb + rows_per_split < M ? mb + rows_per_split : M;
  double s0 = 0.0, s1 = 0.0;
  if (n < N) {
    int64_t m = mb + y;
    for (; m + 4 < me; m += 8) { s0 += (double)A[m * lda + n]; s1 += (double)A[(m + 4) * lda + n]; }
    if (m < me) s0 += (double)A[m * lda + n];
  }
  red[y][c] = s0 + s1;
  __syncthreads();
  if (y == 0 && n < N) part[(int64_t)blockIdx.y * N + n] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

__global__ void __launch_bounds__(256) colsum_reduce_kernel(const double* __restrict__ part, int splits, int N, float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double s = 0.0;
  for (int k = 0; k < splits; ++k) s += part[(int64_t)k * N + i];
  dst[i] = (float)s;
}

// dst[(i / cols) * ldd + i % cols] = sum_s part[s * count + i]   in split order (deterministic)
__global__ void __launch_bounds__(256) reduce_kernel(const float* __restrict__ part, int splits, int64_t count, int cols, int64_t ldd,
                                                     float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += part[(int64_t)k * count + i];
  dst[(i / cols) * ldd + i % cols] = s;
}

// WT[k * N + n] = W[n * ldw + k]
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, float* __restrict__ WT) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  for (int i = ty; i < 32; i += 8)
    tile[i][tx] = (n0 + i < N && k0 + tx < K) ? W[(int64_t)(n0 + i) * ldw + k0 + tx] : 0.f;
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (k0 + i < K && n0 + tx < N) WT[(int64_t)(k0 + i) * N + n0 + tx] = tile[tx][i];
}

hipError_t launch_transpose(const float* W, int64_t ldw, int N, int K, float* WT, hipStream_t stream) {
  transpose_kernel<<<dim3((unsigned)((K + 31) / 32), (unsigned)((N + 31) / 32)), dim3(256), 0, stream>>>(W, ldw, N, K, WT);
  return hipGetLastError();
}

// number of sample splits (workgroups per output tile) of a weight-gradient product: about one workgroup per CU in total (a
// workgroup holds 256 accumulator registers per lane: one wave per SIMD), at least 2,048 samples each; every workgroup leaves
// FOUR partials (one per wave)
int wgrad_splits(int64_t M, int N, int K) {
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  int64_t s = (256 + tiles - 1) / tiles;
  const int64_t cap = (M + 2047) / 2048;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}
int64_t wgrad_part_floats(int64_t M, int N, int K) { return (int64_t)wgrad_splits(M, N, K) * 4 * N * K; }

// dW[0:N, 0:K] (row stride ldd: a column block of an nn.Linear weight) = A^T B, bias gradient optional
hipError_t launch_wgrad_nk(const float* A, int64_t lda, const float* B, int64_t ldb, int rowdiv, int64_t M, int N, int K, float* dW, int64_t ldd,
                           float* part, hipStream_t stream) {
  if (N <= 0 || K <= 0) return hipSuccess;
  const int splits = wgrad_splits(M, N, K);
  int64_t rows = (M + splits - 1) / splits;
  rows = (rows + kWBM - 1) / kWBM * kWBM;
  WgradArgs a{A, lda, B, ldb, rowdiv, M, N, K, rows, part};
  wgrad_nk_kernel<<<dim3((unsigned)((N + 127) / 128), (unsigned)((K + 127) / 128), (unsigned)splits), dim3(256), 0, stream>>>(a);
  const int64_t count = (int64_t)N * K;
  reduce_kernel<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream>>>(part, splits * 4, count, K, ldd, dW);
  return hipGetLastError();
}

hipError_t launch_colsum(const float* A, int64_t lda, int64_t M, int N, float* db, float* part, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  int64_t splits = (M + 2047) / 2048;
  if (splits > 512) splits = 512;
  if (splits < 1) splits = 1;
  const int64_t rows = (M + splits - 1) / splits;
  double* dpart = reinterpret_cast<double*>(part);   // 512 x N doubles at most (carve_gscratch)
  colsum_kernel<<<dim3((unsigned)((N + 63) / 64), (unsigned)splits), dim3(256), 0, stream>>>(A, lda, M, N, rows, dpart);
  colsum_reduce_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream>>>(dpart, (int)splits, N, db);
  return hipGetLastError();
}

}  // namespace aon
