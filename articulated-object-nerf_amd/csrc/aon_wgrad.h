// Weight-gradient machinery shared by the vanilla (aon_train.hip) and articulated (aon_train_art.hip) backward passes.
//
// Round 3: ONE grouped launch per level for every layer's  dW_l = dZ_l^T[M x N] . H_{l-1}[N x K]  (N = samples), one launch
// for every head / bias-only reduction, one for all second-stage sums, one for the latent outer products -- 4 launches per
// level where round 2 issued ~75 (9 GEMM launches + 36 partial reductions + 22 head kernels + fills).
//
// Operands are the STEP-MAJOR planes of aon_mlp_core.h: a 16-byte unit = four consecutive feature rows of one sample, a step =
// 32 samples, an operand of a step = (rows / 4) consecutive 512-byte unit rows = one contiguous run.
//   * staging: LDS-DMA, 1 KiB (two unit rows) per wave-instruction, contiguous in HBM; NSTAGE-deep ring, the DMA of step
//     c + NSTAGE - 1 issued between the MFMA groups of step c, s_waitcnt vmcnt(N) keeps the younger stages in flight across the
//     workgroup barrier; the loop body is branch-free (a DMA past the end of the range re-reads the last step).
//   * fragments: lane (i, kh) reads the unit (unit row g0 + i, sample 2p + kh) with ONE ds_read_b128 = the A (or B) values of
//     FOUR feature rows for k-index kh: one A read and one B read feed 16 MFMAs (128 x 128 outputs x 2 samples).  Round 2
//     read one row x 4 samples per lane: 10 reads per 64 MFMAs in a batch with the full LDS latency exposed every 64 MFMAs.
//     Bank conflicts are removed on the GLOBAL side of the DMA: the lane that fills slot s of unit row g fetches sample
//     s ^ (g & 15), so the 16 lanes of a ds_read_b128 group (16 different unit rows, same sample) hit 16 different slots.
//   * the sample range of a layer is split over a number of workgroups proportional to its cost (all layers of a level run
//     concurrently and finish together); a workgroup keeps its output block in accumulator registers and writes one partial,
//     a deterministic second stage (fixed order, no atomics) sums the partials.  ~8x fewer partial bytes than round 2, which
//     split EVERY layer over all 256 workgroups.
#pragma once
#include "aon_mlp_core.h"

namespace aon {

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------------
// grouped weight gradients
// ---------------------------------------------------------------------------------------------
// Job kinds (M = rows of dZ, K = rows of the activation operand; a wave owns a 128 x (32 NB) block of 4 x NB accumulator tiles):
//   kWg256x256   4 waves = 2 x 2 blocks of 128 x 128, every wave takes all 16 sample pairs of a step
//   kWg128x128   one block, the four waves take 4 sample pairs each (four partials)
//   kWg256x64    two 128 x 64 blocks x two sample halves                      (positional-encoding inputs, 63 valid columns)
//   kWg128x256   two 128 x 128 blocks x two sample halves                     (view layer 0 on the bottleneck output)
//   kWg128x32    one 128 x 32 block (one B register per lane), four sample quarters   (view-encoding inputs, 27 valid columns)
// No wave holds more than 16 accumulator tiles = the 256 AGPRs: a 128 x 288 kind with a 4-tile extension (320 registers) made
// hipcc shuttle tiles between AGPRs and VGPRs around every MFMA group (160 v_accvgpr moves per 20 MFMAs, 0.60 of peak).
enum : int { kWg256x256 = 0, kWg128x128 = 1, kWg256x64 = 2, kWg128x256 = 3, kWg128x32 = 4, kWgNumKinds = 5 };

// Round 4: the steps of ALL jobs of a launch form one work line, priced in cost units (steps x wg_cost(kind)); workgroup i of G owns
// the steps whose START falls into [W i / G, W (i + 1) / G).  A workgroup therefore runs the tail of one layer's range and the head
// of the next (two or three segments, one partial each) and every workgroup carries the same cost -- round 3 gave each layer a
// whole number of workgroups (a 256 x 256 layer 23 or 24 of 256, a 128 x 128 one 6 or 7), and the launch lasted as long as the
// layer whose rounding came out worst (tools/kernel_bench.py --wgrad-probe: profiles/r04_wgrad_probe.txt).
struct WgJob {
  int a_unit;      // first unit row (plane row / 4) of dZ in the gradient planes
  int b_unit;      // first unit row of the layer input in the forward planes
  int kind;
  int cost;        // wg_cost(kind), integer cost units per step
  int first_wg, last_wg;    // workgroups that own a step of this job (each writes nsplit partials, indexed from first_wg)
  int part_off;    // float offset in the workspace of partial[(last_wg - first_wg + 1) * nsplit][M][K]
  int bias_off;    // float offset of bias_partial[... * nsplit][M], or -1
  int64_t p_begin; // position of step 0 on the work line
};

// second stage: out[row * ld + col_off + col] = sum_p partial[p][row][col]  (col < k_valid);  bias_out[row] = sum_p bias_partial[p][row]
struct WgReduce {
  int part_off, nparts, M, K, k_valid, ld, col_off;
  int bias_off;          // -1: none
  int blk_begin;         // first block of the reduce launch for this entry (weights: M*K/256 blocks, then bias: ceil(M/16))
  int nblk_w;
  float* out;
  float* bias_out;
};

constexpr int kWgMaxJobs = 20;
constexpr int kWgMaxReduce = 24;

struct WgArgs {
  const float* dplanes;
  const float* planes;
  int64_t step_bytes;   // rows * 128
  int nsteps;           // Np / 32
  int njobs;
  int nwgs;             // G
  int64_t w_total;      // W: sum over jobs of nsteps * cost
  float* ws;
  long long* probe;     // measurement aid (aon_set_wgrad_probe): [2 * workgroup] = wall_clock64() at entry / exit, or null
  WgJob job[kWgMaxJobs];
};

template <int KIND> struct WgTraits;
template <> struct WgTraits<kWg256x256> { static constexpr int NGA = 64, NGB = 64, NB = 4, NAB = 2, NBB = 2, NSPLIT = 1, NSTAGE = 2; };
template <> struct WgTraits<kWg128x128> { static constexpr int NGA = 32, NGB = 32, NB = 4, NAB = 1, NBB = 1, NSPLIT = 4, NSTAGE = 4; };
template <> struct WgTraits<kWg256x64>  { static constexpr int NGA = 64, NGB = 16, NB = 2, NAB = 2, NBB = 1, NSPLIT = 2, NSTAGE = 3; };
template <> struct WgTraits<kWg128x256> { static constexpr int NGA = 32, NGB = 64, NB = 4, NAB = 1, NBB = 2, NSPLIT = 2, NSTAGE = 3; };
template <> struct WgTraits<kWg128x32>  { static constexpr int NGA = 32, NGB = 8,  NB = 1, NAB = 1, NBB = 1, NSPLIT = 4, NSTAGE = 4; };

template <int KIND> constexpr int wg_stage_bytes() { return (WgTraits<KIND>::NGA + WgTraits<KIND>::NGB) * 512; }
constexpr int kWgLdsBytes = 144 * 1024;   // max over kinds of NSTAGE * stage bytes (2 x 64 KiB, 4 x 32 KiB, 3 x 40 KiB, 3 x 48 KiB, 4 x 20 KiB)

__host__ __device__ constexpr int wg_nsplit(int kind) { return kind == kWg256x256 ? 1 : (kind == kWg128x128 || kind == kWg128x32) ? 4 : 2; }
__host__ __device__ constexpr int wg_M(int kind) { return (kind == kWg256x256 || kind == kWg256x64) ? 256 : 128; }
__host__ __device__ constexpr int wg_K(int kind) {
  return kind == kWg256x256 ? 256 : kind == kWg128x128 ? 128 : kind == kWg256x64 ? 64 : kind == kWg128x256 ? 256 : 32;
}

#ifdef AON_WGRAD_KERNELS   // the kernels are compiled once, in aon_train.hip (run_wgrad_plan is the only launcher)
// workgroup barrier that leaves the N youngest vector-memory operations (LDS-DMA of later stages) in flight.  __syncthreads()
// would drain them all (its release fence is an s_waitcnt vmcnt(0)); everything this barrier has to order is spelled out:
// the stage about to be read has landed (vmcnt), this wave's LDS reads of the stage about to be overwritten have returned
// (lgkmcnt), and the "memory" clobber keeps the compiler from moving LDS accesses across it.
template <int N>
__device__ __forceinline__ void stage_barrier() {
#if defined(AON_EXP_WG_NOBAR)     // timing experiment only (WRONG results): the waits without the workgroup barrier
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
#elif defined(AON_EXP_WG_NOVM)    // timing experiment only (WRONG results): the barrier without the wait for the stage's DMA
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

template <int KIND>
__device__ __forceinline__ void wgrad_job(const WgArgs& a, const WgJob& J, const int c_begin, const int c_end, const int part_wg, char* smem) {
  using T = WgTraits<KIND>;
  constexpr int NB = T::NB, NSTAGE = T::NSTAGE, NSPLIT = T::NSPLIT;
  constexpr int STAGE = wg_stage_bytes<KIND>();
  constexpr int ND = (T::NGA + T::NGB) / 2;   // 1 KiB DMA instructions per step
  static_assert(ND % 4 == 0 && (T::NGA / 2) % 4 == 0, "DMA instructions split evenly over the four waves, A/B boundary on a multiple of 4");
  constexpr int D = ND / 4;                            // per wave
  constexpr int PW = 16 / NSPLIT;                      // sample pairs per wave per step
  static_assert(NSTAGE * STAGE <= kWgLdsBytes, "LDS ring");
  static_assert((NSTAGE - 2) * D <= 63, "vmcnt range");

  const int tid = threadIdx.x;
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kh = lane >> 5;
  const int wb = wave % T::NBB, wa = (wave / T::NBB) % T::NAB, split = wave / (T::NBB * T::NAB);

  // steps [c_begin, c_end) of this job belong to this workgroup (empty only in tiny problems: then the partial is zeros)
  const int c_last = c_end - 1;

  f32x16 acc[4][NB];
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ca][cb][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // ---- DMA side: instruction d = 4k + wave moves unit rows 2d, 2d+1 of the concatenated [A | B] operand list ----
  // lane (gl = lane >> 5, slot = lane & 31) fills slot `slot` of unit row 2d + gl with sample slot ^ ((2d + gl) & 15);
  // (2d) & 15 = 8 (k & 1) + 2 wave, so two per-lane offsets cover every instruction of this wave
  const unsigned v0 = (unsigned)((lane >> 5) * 512 + (((lane & 31) ^ (lane >> 5)) << 4));
  const unsigned voff_e = v0 ^ (unsigned)((2 * wave) << 4), voff_o = v0 ^ (unsigned)((8 + 2 * wave) << 4);
  const char* gA = reinterpret_cast<const char*>(a.dplanes) + (int64_t)J.a_unit * 512;
  const char* gB = reinterpret_cast<const char*>(a.planes) + (int64_t)J.b_unit * 512;
  auto dma_step = [&](int c, int stage, int k) {   // k-th instruction of this wave for step c (clamped) into `stage`
    const int cc = c < c_last ? c : c_last;
    int64_t soff = (int64_t)cc * a.step_bytes;
    asm volatile("" : "+s"(soff));
    const int d = 4 * k + wave;
    const char* g = (k < T::NGA / 8 ? gA + (int64_t)d * 1024 : gB + (int64_t)(d - T::NGA / 2) * 1024) + soff + ((k & 1) ? voff_o : voff_e);
    char* l = smem + stage * STAGE + d * 1024;
    // (default cache policy: `nt` on these loads measured no different, 9.18 vs 9.22 ms for a 4096 x 193 level)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_void*)l, 16, 0, 0);
  };

  // ---- fragment side ----
  // unit (unit row g, sample S) sits at g * 512 + ((S ^ (g & 15)) << 4); S = 2p + kh  ->  (lane part) ^ (p << 5)
  const unsigned offA = (unsigned)(wa * 32 * 512 + i * 512 + ((kh ^ (i & 15)) << 4));
  unsigned offB;
  if constexpr (NB == 4) offB = (unsigned)(T::NGA * 512 + wb * 32 * 512 + i * 512 + ((kh ^ (i & 15)) << 4));
  else if constexpr (NB == 2) offB = (unsigned)(T::NGA * 512 + (i & 15) * 512 + ((kh ^ (i & 15)) << 4));   // lanes i and i + 16 share a unit
  else offB = (unsigned)(T::NGA * 512 + (i & 7) * 512 + ((kh ^ (i & 7)) << 4));                             // NB == 1: four lanes share a unit
  auto pair_of = [&](int q) { return split * PW + q; };   // this wave's pairs of a step
  // Fragment reads are inline asm with hand-placed waits.  Written as plain loads, hipcc (ROCm 7.2) (a) sinks each ds_read to
  // just in front of its first use unless pinned, and (b) pinned, still guards every second MFMA group with s_waitcnt
  // lgkmcnt(0), i.e. waits for the reads it has JUST issued for the next pair: the full LDS latency exposed per 32 MFMAs
  // (measured: 256 x 256 jobs 0.864 of peak).  Here the reads of pair q + 1 are issued, then `wait_frag<N>` waits until only
  // those N reads are outstanding (LDS returns in order) and hands the registers of pair q to the MFMAs through a data
  // dependence.  The compiler never copies a fragment between its read and its wait (straight-line, fully unrolled code;
  // the one loop-carried fragment is waited for before the back edge).
  struct Frag { f32x4 a, b; };
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
  auto lds_read = [](unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };
  auto read_frag = [&](int stage, int q) {
    Frag f;
    const unsigned px = (unsigned)pair_of(q) << 5;
    const unsigned sb = lds0 + (unsigned)(stage * STAGE);
    f.a = lds_read(sb + (offA ^ px));
    f.b = lds_read(sb + (offB ^ px));
    return f;
  };

  if (c_end > c_begin) {
  // prologue: NSTAGE - 1 steps in flight, the first one landed
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
#pragma unroll
    for (int k = 0; k < D; ++k) dma_step(c_begin + s, s, k);
  stage_barrier<(NSTAGE - 2) * D>();
  int stage = 0;
  Frag cur = read_frag(0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur.a), "+v"(cur.b));
  for (int c = c_begin; c < c_end; ++c) {
    const int stage_dma = stage == 0 ? NSTAGE - 1 : stage - 1;     // (stage + NSTAGE - 1) % NSTAGE: consumed in step c - 1
    const int stage_next = stage == NSTAGE - 1 ? 0 : stage + 1;
#pragma unroll
    for (int q = 0; q < PW; ++q) {
      Frag nxt;
      if (q + 1 < PW) {
        nxt = read_frag(stage, q + 1);
      } else {
        // every DMA of step c + NSTAGE - 1 has been issued (first half of the step): step c + 1 must have landed, all waves
        // are done reading `stage` except through registers
        stage_barrier<(NSTAGE - 2) * D>();
        nxt = read_frag(stage_next, 0);
      }
      // DMA instructions of step c + NSTAGE - 1, spread over the FIRST HALF of the step's pairs: with two stages they must
      // land before this step's barrier, and the HBM latency is a good part of a step of the narrow kinds
      constexpr int HALF = PW / 2 > 0 ? PW / 2 : 1;
      constexpr int PER = (D + HALF - 1) / HALF;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int k = q * PER + u;
        if (q < HALF && k < D) dma_step(c + NSTAGE - 1, stage_dma, k);
      }
      __builtin_amdgcn_sched_barrier(0);   // the reads / DMA instructions above stay above the MFMAs below
      if (q > 0) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur.a), "+v"(cur.b));   // (q == 0: waited for at the end of the previous step)
#pragma unroll
      for (int ca = 0; ca < 4; ++ca) bsum[ca] += cur.a[ca];   // bias gradient = row sums of dZ (written once per A block below)
      f32x4 be = cur.b;
      if constexpr (NB == 2) { if (i >= 16) { be[0] = cur.b[2]; be[1] = cur.b[3]; } }
      if constexpr (NB == 1) {   // column n of the 32-wide block = row 4 (n & 7) + (n >> 3)
        const int sel = i >> 3;
        be[0] = sel == 0 ? cur.b[0] : sel == 1 ? cur.b[1] : sel == 2 ? cur.b[2] : cur.b[3];
      }
#pragma unroll
      for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[ca], be[cb], acc[ca][cb], 0, 0, 0);
      cur = nxt;
    }
    // the loop-carried fragment (pair 0 of the next step) is complete before the back edge
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur.a), "+v"(cur.b));
    stage = stage_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped look-ahead DMAs must land before the LDS is released
  }

  // ---- partials ----
  // accumulator (ca, cb), register r, lane (n, hh): row 4 (32 wa + (r&3) + 8 (r>>2) + 4 hh) + ca, column 128 wb + 4 n + cb
  constexpr int M = 128 * T::NAB, K = NB == 2 ? 64 : NB == 1 ? 32 : 128 * T::NBB;
  const int part = part_wg * NSPLIT + split;
  float* P = a.ws + J.part_off + (int64_t)part * M * K;
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 128 * wa + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + ca;
      if constexpr (NB == 4) {
        f32x4 v; v[0] = acc[ca][0][r]; v[1] = acc[ca][1][r]; v[2] = acc[ca][2][r]; v[3] = acc[ca][3][r];
        *reinterpret_cast<f32x4*>(P + (int64_t)row * K + 128 * wb + 4 * i) = v;
      } else if constexpr (NB == 2) {
        float2 v; v.x = acc[ca][0][r]; v.y = acc[ca][1][r];
        *reinterpret_cast<float2*>(P + (int64_t)row * K + 4 * (i & 15) + 2 * (i >> 4)) = v;
      } else {
        P[(int64_t)row * K + 4 * (i & 7) + (i >> 3)] = acc[ca][0][r];
      }
    }
  if (J.bias_off >= 0 && wb == 0) {
    f32x4 v;
#pragma unroll
    for (int ca = 0; ca < 4; ++ca) v[ca] = bsum[ca] + __shfl_xor(bsum[ca], 32);
    if (kh == 0) *reinterpret_cast<f32x4*>(a.ws + J.bias_off + (int64_t)part * M + 128 * wa + 4 * i) = v;
  }
}

__device__ __forceinline__ int wg_first_step(int64_t x, int64_t p_begin, int cost, int nsteps) {
  // number of steps of the job whose start lies below x on the work line: ceil((x - p_begin) / cost), clamped to [0, nsteps]
  const int64_t d = x - p_begin;
  if (d <= 0) return 0;
  const int64_t q = (d + cost - 1) / cost;
  return q < nsteps ? (int)q : nsteps;
}

__global__ void __launch_bounds__(256) wgrad_grouped_kernel(WgArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char wg_smem[];
  const int wg = (int)blockIdx.x;
  if (a.probe && threadIdx.x == 0) a.probe[2 * wg] = wall_clock64();
  const int64_t lo = a.w_total * wg / a.nwgs, hi = a.w_total * (wg + 1) / a.nwgs;
  bool ran = false;
#pragma unroll 1
  for (int j = 0; j < a.njobs; ++j) {
    const WgJob& J = a.job[j];
    if (wg < J.first_wg || wg > J.last_wg) continue;   // (workgroup-uniform)
    const int c_begin = wg_first_step(lo, J.p_begin, J.cost, a.nsteps), c_end = wg_first_step(hi, J.p_begin, J.cost, a.nsteps);
    if (ran) __syncthreads();   // the previous segment's last fragment reads are done in every wave before its stages are refilled
    ran = true;
    const int part_wg = wg - J.first_wg;
    switch (J.kind) {
      case kWg256x256: wgrad_job<kWg256x256>(a, J, c_begin, c_end, part_wg, wg_smem); break;
      case kWg128x128: wgrad_job<kWg128x128>(a, J, c_begin, c_end, part_wg, wg_smem); break;
      case kWg256x64: wgrad_job<kWg256x64>(a, J, c_begin, c_end, part_wg, wg_smem); break;
      case kWg128x256: wgrad_job<kWg128x256>(a, J, c_begin, c_end, part_wg, wg_smem); break;
      default: wgrad_job<kWg128x32>(a, J, c_begin, c_end, part_wg, wg_smem); break;
    }
  }
  if (a.probe && threadIdx.x == 0) a.probe[2 * wg + 1] = wall_clock64();
}

#endif  // AON_WGRAD_KERNELS

// ---------------------------------------------------------------------------------------------
// heads, biases of the heads, and the first deformation layer: rows of a plane against ONE 16-byte record per sample
// ---------------------------------------------------------------------------------------------
//   s[row][c] = sum_n plane[row][n] * vec[n][c]  (c = 0..3),   s[row][4] = sum_n plane[row][n]
// `vec` is either a sample-major (Np,4) buffer (d_raw, dx') or a unit row of the forward planes (the sample position: the
// first deformation layer's three input columns); plane == null: the bias sums  s[0][c] = sum_n vec[n][c].
// A block reduces EIGHT plane rows (two unit rows) over one segment of samples -> partial[seg][row][8].
struct HeadJob {
  const float* plane;      // dplanes / planes base, or null
  int unit;                // first unit row
  int rows;                // 1 when plane == null
  const float* vec;        // record of sample n at vec + (n >> 5) * vec_step + (n & 31) * 4
  int64_t vec_step;        // floats per 32 samples: 128 (sample-major) or rows_total * 32 (a plane unit row)
  int blk_begin;           // first blockIdx.x of this job (ceil(rows / 8) blocks)
  int part_off;            // float offset (even: the records are doubles) of partial[nseg][rows][8 doubles]
};
constexpr int kHeadMaxJobs = 8;
struct HeadArgs {
  HeadJob job[kHeadMaxJobs];
  int njobs;
  int blk_offset;          // added to blockIdx.x: a launch may cover a suffix of the jobs' blocks (round 5: early / late head reductions)
  int64_t step_floats;     // rows_total * 32
  int64_t Np, seg_len;
  float* ws;
};

#ifdef AON_WGRAD_KERNELS
// The sums are SIGNED sums over every sample of a level (the one- and three-element head biases are pure cancellation on a
// trained field), so every stage accumulates in fp64: the product of two floats is exact in a double, the partials are
// doubles, and the result is rounded to float ONCE, in the second stage.  The kernel is HBM-bound: the fp64 pipe is idle
// otherwise.  (Rounds 1-3 summed in fp32: one constructor-fuzz seed in 150 had the coarse density bias 3e-3 from the fp64
// truth where the reference's own fp32 sits at 9e-5.)
__device__ __forceinline__ double wsum64d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ void __launch_bounds__(256) head_wgrad_kernel(HeadArgs a) {
  const int bx = (int)blockIdx.x + a.blk_offset;
  int j = 0;
#pragma unroll 1
  for (int t = 1; t < a.njobs; ++t)
    if (bx >= a.job[t].blk_begin) j = t;
  const HeadJob& J = a.job[j];
  const int row0 = (bx - J.blk_begin) * 8, seg = blockIdx.y;
  const int nr = J.rows - row0 < 8 ? J.rows - row0 : 8;
  const int64_t n0 = (int64_t)seg * a.seg_len;
  const int64_t n1 = n0 + a.seg_len < a.Np ? n0 + a.seg_len : a.Np;
  double s[8][5];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) s[r][c] = 0.0;
  const float* u0 = J.plane ? J.plane + (int64_t)(J.unit + row0 / 4) * 128 : nullptr;
  for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
    const int64_t st = n >> 5;
    const int sl = (int)(n & 31);
    const f32x4 d = *reinterpret_cast<const f32x4*>(J.vec + st * J.vec_step + sl * 4);
    f32x4 x0 = {1.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
    if (u0) {
      x0 = *reinterpret_cast<const f32x4*>(u0 + st * a.step_floats + sl * 4);
      if (nr > 4) x1 = *reinterpret_cast<const f32x4*>(u0 + 128 + st * a.step_floats + sl * 4);
    }
    const double dd[4] = {(double)d[0], (double)d[1], (double)d[2], (double)d[3]};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const double x = (double)(r < 4 ? x0[r] : x1[r - 4]);
#pragma unroll
      for (int c = 0; c < 4; ++c) s[r][c] = __builtin_fma(x, dd[c], s[r][c]);
      s[r][4] += x;
    }
  }
  __shared__ double red[4][8][5];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const double v = wsum64d(s[r][c]);
      if (lane == 0) red[wv][r][c] = v;
    }
  __syncthreads();
  if (threadIdx.x < 40) {
    const int r = threadIdx.x / 5, c = threadIdx.x % 5;
    double* P = reinterpret_cast<double*>(a.ws + J.part_off);
    if (r < nr) P[((int64_t)seg * J.rows + row0 + r) * 8 + c] = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
  }
}

#endif  // AON_WGRAD_KERNELS

// second stage of the heads: out[c * stride_c + row * stride_r] = sum_seg partial[seg][row][chan0 + c]
struct HeadOut {
  int part_off, rows, chan0, nchan, stride_c, stride_r;
  float* out;
};
constexpr int kHeadMaxOut = 12;

// ---------------------------------------------------------------------------------------------
// second stage (deterministic, no atomics), ONE launch for every layer and every head of a level
// ---------------------------------------------------------------------------------------------
// Weight blocks: 64 float4 columns x 4 waves; wave w sums the partials p = w, w+4, ... with 8 independent 16-byte loads in
// flight per lane, then the four wave sums are added in wave order through LDS.
struct ReduceArgs {
  WgReduce red[kWgMaxReduce];
  HeadOut head[kHeadMaxOut];
  int nred, nhead, nseg;
  int head_blk_begin;      // blocks >= this one belong to the heads (one block per HeadOut)
  float* ws;
};

#ifdef AON_WGRAD_KERNELS
// (the body of one block of the second stage; `bx`: the block's index in ITS level's launch)
__device__ __forceinline__ void wgrad_reduce_block(const ReduceArgs& a, const int bx, f32x4 (*red)[64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (bx >= a.head_blk_begin) {
    const HeadOut& H = a.head[bx - a.head_blk_begin];
    for (int idx = threadIdx.x; idx < H.rows * H.nchan; idx += 256) {
      const int row = idx / H.nchan, c = idx % H.nchan;
      const double* P = reinterpret_cast<const double*>(a.ws + H.part_off);
      // (round 6: the loads of eight segments in flight at once, the additions in the SAME order as the plain loop -- same bits.  As a
      // plain loop the 49 segment partials of a 4096 x 193 level were 49 dependent-latency loads in a row: these few blocks made the
      // whole second stage 58 us long, at the very end of the backward)
      const double* Pr = P + (int64_t)row * 8 + H.chan0 + c;
      const int64_t seg_stride = (int64_t)H.rows * 8;
      double s = 0.0;
      int p = 0;
      for (; p + 8 <= a.nseg; p += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = Pr[(int64_t)(p + u) * seg_stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; p < a.nseg; ++p) s += Pr[(int64_t)p * seg_stride];
      H.out[(int64_t)c * H.stride_c + (int64_t)row * H.stride_r] = (float)s;
    }
    return;
  }
  int e = 0;
#pragma unroll 1
  for (int t = 1; t < a.nred; ++t)
    if (bx >= a.red[t].blk_begin) e = t;
  const WgReduce& R = a.red[e];
  const int blk = bx - R.blk_begin;
  if (blk < R.nblk_w) {
    const int idx = (blk * 64 + lane) * 4;  // first of 4 consecutive columns of one row (K is a multiple of 32)
    if (idx < R.M * R.K) {
      const f32x4* src = reinterpret_cast<const f32x4*>(a.ws + R.part_off + idx);
      const int64_t stride4 = (int64_t)R.M * R.K / 4;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      int pp = wave;
      for (; pp + 28 < R.nparts; pp += 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(pp + 4 * u) * stride4];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; pp < R.nparts; pp += 4) s += src[(int64_t)pp * stride4];
      red[wave][lane] = s;
    }
    __syncthreads();
    if (wave == 0 && idx < R.M * R.K) {
      const f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
      const int row = idx / R.K, col = idx % R.K;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (col + c < R.k_valid) R.out[(int64_t)row * R.ld + R.col_off + col + c] = t[c];
    }
  } else if (R.bias_off >= 0) {
    // 16 rows per block; thread (r, g) sums partials g, g+16, ...; the 16 group sums are added in group order
    // (the partials are the fp32 row sums a weight-gradient workgroup took over its steps; from here on in fp64)
    double* redd = reinterpret_cast<double*>(&red[0][0]);   // 256 doubles of the 4 KiB
    const int r = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int row = (blk - R.nblk_w) * 16 + r;
    double s = 0.0;
    if (row < R.M) {
      const float* B = a.ws + R.bias_off + row;
      int pp = g;
      for (; pp + 48 < R.nparts; pp += 64) {   // four loads in flight, added in the order of the plain loop (same bits)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = B[(int64_t)(pp + 16 * u) * R.M];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (double)v[u];
      }
      for (; pp < R.nparts; pp += 16) s += (double)B[(int64_t)pp * R.M];
    }
    redd[g * 16 + r] = s;
    __syncthreads();
    if (g == 0 && row < R.M) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += redd[q * 16 + r];
      R.bias_out[row] = (float)t;
    }
  }
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(ReduceArgs a) {
  __shared__ f32x4 red[4][64];
  wgrad_reduce_block(a, (int)blockIdx.x, red);
}

// The second stages of TWO levels as one launch (round 6: both levels' grouped kernels run back to back and their partial sums are
// reduced together): blocks [0, n0) are level 0's, the rest level 1's; each block does exactly what it does in a launch of its own.
__global__ void __launch_bounds__(256) wgrad_reduce2_kernel(ReduceArgs a0, ReduceArgs a1, int n0) {
  __shared__ f32x4 red[4][64];
  if ((int)blockIdx.x < n0) wgrad_reduce_block(a0, (int)blockIdx.x, red);
  else wgrad_reduce_block(a1, (int)blockIdx.x - n0, red);
}
static_assert(2 * sizeof(ReduceArgs) + 16 <= 4096, "two levels' second-stage arguments must fit the kernel argument segment");

#endif  // AON_WGRAD_KERNELS

// ---------------------------------------------------------------------------------------------
// host side: the plan of one level
// ---------------------------------------------------------------------------------------------
struct WgLayerDesc {   // one nn.Linear weight (or a column block of one)
  int kind;
  int a_row, b_row;    // first plane rows of dZ (gradient planes) and of the layer input (forward planes)
  float* out; int ld, col_off, k_valid;
  float* bias_out;     // or null
};

// cost of one step of a job (integer units): 16,384 matrix-pipe cycles for a 256 x 256 job, the others scaled by their time per step
// MEASURED INSIDE THE MIXED LAUNCH with the workgroup clock probe (tools/kernel_bench.py --wgrad-probe, round 4, articulated level of
// 4096 x 193 / x 65 samples: 7.70 / 8.76 us per 256x256 step, the other kinds 0.261 / 0.260, 0.282 / 0.281, 0.507 / 0.508, 0.093 /
// 0.092 of it).  Round 3 priced them from whole-chip runs of one kind (0.275, 0.299, 0.52, 0.11):
// step and workgroup with the whole chip running one kind (tools/kernel_bench.py --wgrad-kinds, 4096 x 193 samples, two boxes:
// 128x128 0.274-0.276, 256x64 0.298-0.301, 128x256 0.51-0.53, 128x32 0.107-0.108 of a 256x256 job).  The narrow kinds are bound
// by operand traffic per flop (3.9-6.3 TB/s when they run alone), not by the pipe.
inline int wg_cost(int kind) {
  switch (kind) {
    case kWg256x256: return 16384;
    case kWg128x128: return 4280;
    case kWg256x64: return 4620;
    case kWg128x256: return 8300;
    default: return 1510;   // kWg128x32: 1,024 matrix-pipe cycles per step, but 20 KB of operands -- DMA-bound
  }
}

struct WgPlan {
  WgArgs args;
  ReduceArgs red;
  int total_wgs, reduce_blocks;
  int64_t ws_floats;   // workspace floats used by the partials (heads are appended behind by the caller)
};

// workgroup that owns position x of the work line: the i with floor(W i / G) <= x < floor(W (i + 1) / G)
inline int wg_owner(int64_t x, int64_t W, int G) {
  int i = (int)((x * G) / W);
  if (i >= G) i = G - 1;
  while (i + 1 < G && W * (i + 1) / G <= x) ++i;
  while (i > 0 && W * i / G > x) --i;
  return i;
}

// One launch of G <= cus workgroups (all co-resident: ONE round), every workgroup the same share of the work line.
inline bool wg_make_plan(const WgLayerDesc* layers, int nlayers, const float* planes, const float* dplanes, int rows_total, int64_t Np,
                         int cus, float* ws, int64_t ws_off, WgPlan& plan) {
  if (nlayers > kWgMaxJobs || nlayers < 1 || cus < 1) return false;
  const int nsteps = (int)(Np / 32);
  if (nsteps < 1) return false;
  WgArgs& A = plan.args;
  A.dplanes = dplanes; A.planes = planes; A.step_bytes = (int64_t)rows_total * 128; A.nsteps = nsteps; A.njobs = nlayers; A.ws = ws;
  A.probe = nullptr;
  int64_t W = 0;
  for (int l = 0; l < nlayers; ++l) { A.job[l].p_begin = W; W += (int64_t)nsteps * wg_cost(layers[l].kind); }
  // never more workgroups than steps on the line (tiny problems), never more than the compute units
  const int64_t total_steps = (int64_t)nsteps * nlayers;
  int G = cus;
  if (G > total_steps) G = (int)total_steps;
  A.nwgs = G; A.w_total = W;
  ReduceArgs& R = plan.red;
  R.nred = 0; R.ws = ws;
  int64_t off = ws_off;
  int blk = 0;
  for (int l = 0; l < nlayers; ++l) {
    const WgLayerDesc& L = layers[l];
    WgJob& J = A.job[l];
    const int M = wg_M(L.kind), K = wg_K(L.kind), nsplit = wg_nsplit(L.kind);
    J.a_unit = L.a_row / 4; J.b_unit = L.b_row / 4; J.kind = L.kind; J.cost = wg_cost(L.kind);
    J.first_wg = wg_owner(J.p_begin, W, G);
    J.last_wg = wg_owner(J.p_begin + (int64_t)(nsteps - 1) * J.cost, W, G);
    const int nparts = (J.last_wg - J.first_wg + 1) * nsplit;
    J.part_off = (int)off; off += (int64_t)nparts * M * K;
    J.bias_off = -1;
    if (L.bias_out) { J.bias_off = (int)off; off += (int64_t)nparts * M; }
    if (R.nred >= kWgMaxReduce) return false;
    WgReduce& E = R.red[R.nred++];
    E.part_off = J.part_off; E.nparts = nparts; E.M = M; E.K = K; E.k_valid = L.k_valid; E.ld = L.ld; E.col_off = L.col_off;
    E.bias_off = J.bias_off; E.out = L.out; E.bias_out = L.bias_out;
    E.blk_begin = blk; E.nblk_w = (M * K / 4 + 63) / 64; blk += E.nblk_w + (L.bias_out ? (M + 15) / 16 : 0);
  }
  plan.total_wgs = G;
  plan.reduce_blocks = blk;
  plan.ws_floats = off;
  return off < (int64_t)1 << 31;
}

inline int64_t wgrad_workspace_bytes_impl() {
  // one 256 x 256 partial per workgroup of a 256-CU launch at most, + the four-way partials of the narrow layers, bias
  // partials and the head partials: 70 MB for the articulated network on 256 CUs; 96 MiB covers every plan wg_make_plan can
  // produce for <= 304 CUs (run_wgrad_plan checks the plan against it)
  return (int64_t)96 << 20;
}

// rows x 4-channel reductions of a plane against a 16-byte record per sample (heads, biases)
inline void head_segments(int64_t Np, int& nseg, int64_t& seg_len) {
  nseg = (int)((Np + 16383) / 16384);
  if (nseg > 256) nseg = 256;
  seg_len = ((Np + nseg - 1) / nseg + 31) / 32 * 32;
  nseg = (int)((Np + seg_len - 1) / seg_len);
}

// An optional side stream for the head reductions of a level (with its fork / join events): they are HBM-bound plane-row sums
// with a small register / LDS footprint, so their workgroups fit next to the weight-gradient workgroups (384 of the 512 registers
// per SIMD, 144 of the 160 KB of LDS) and run in their shadow instead of behind them.  Null: everything on one stream.
struct WgAux {
  hipStream_t stream;
  hipEvent_t fork, join;
};

// Round 6: what follows a level's grouped kernel -- the second stage, the un-folding products, the latent columns (three launches of 15-60 us
// in a row) -- may run on a side stream (`side`), beside the NEXT level's head reductions and grouped kernel instead of in front of them;
// `wait_first`: an event the second stage of THIS level waits for first (the other level's finishing kernels, whose latent gradients this
// level's are added to).  The caller orders its stream behind `side->join` before it reads any result.
struct WgPost {
  const WgAux* side;
  hipEvent_t wait_first;
};

enum : int { kWgAll = 0, kWgEarly = 1, kWgRest = 2 };   // phases of a level's weight-gradient call (run_wgrad_plan)

struct HeadDesc {
  const float* plane; int row; int rows;   // plane == null: record sums only (rows = 1)
  const float* vec; int64_t vec_step;
};

// builds the head launch; returns the grid's x extent; `outs` are appended to the reduce launch by the caller
inline int head_make_plan(const HeadDesc* heads, int nheads, int rows_total, int64_t Np, float* ws, int64_t& ws_off, HeadArgs& H, int* part_offs) {
  int nseg; int64_t seg_len;
  head_segments(Np, nseg, seg_len);
  H.njobs = nheads; H.step_floats = (int64_t)rows_total * 32; H.Np = Np; H.seg_len = seg_len; H.ws = ws;
  int blk = 0;
  for (int j = 0; j < nheads; ++j) {
    HeadJob& J = H.job[j];
    J.plane = heads[j].plane; J.unit = heads[j].row / 4; J.rows = heads[j].rows; J.vec = heads[j].vec; J.vec_step = heads[j].vec_step;
    J.blk_begin = blk; blk += (heads[j].rows + 7) / 8;
    ws_off += ws_off & 1;   // records of 8 doubles
    J.part_off = (int)ws_off; part_offs[j] = J.part_off;
    ws_off += (int64_t)nseg * heads[j].rows * 16;
  }
  return blk;
}

}  // namespace aon
