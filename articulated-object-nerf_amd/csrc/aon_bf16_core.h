// Core of the split-precision ("bf16x3") register-resident MLP kernels, shared by the vanilla (aon_mlp_bf16.hip) and the
// articulated (aon_mlp_art_bf16.hip) engines: limb fragments, the (pre-op +) split of an fp32 tile, the chunk MFMA loop with
// the next tile's split and the next chunk's DMA interleaved, 8-tile layers, ReLU'd head dot products.
#pragma once
#include "aon_mlp_core.h"
#include "aon_bf16_split.h"

namespace aon {

struct Bf16Net {
  static constexpr int kNumChunks = aon::kNumChunks;
  static constexpr int kSlotBytes = 8 * 6144;  // 48 KiB
  static constexpr bool kPair = false;         // own single-chunk schedule (chunk_mma_bf16)
  // chunk = [k16 step s (2)][out tile][limb (3)][lane (64)][8 bf16]  ->  6 KiB per output tile
  static constexpr int chunk_bytes(int c) { return (c < kNumBigChunks ? 8 : 4) * 6144; }
};
constexpr int64_t kBfStreamBytes = (int64_t)kNumBigChunks * 8 * 6144 + (int64_t)(kNumChunks - kNumBigChunks) * 4 * 6144;
constexpr int kBfRingBytes = 2 * Bf16Net::kSlotBytes;
constexpr int kBfEncStashBytes = 4 * 2 * 64 * 64;  // per wave: 2 encoding tiles x 64 lanes x 16 floats (parked in LDS between L0 and L5)
constexpr int kBfLdsBytes = kBfRingBytes + (int)kSmallBytes + 16 /*pad to 16 B*/ + kBfEncStashBytes;

// host+device scalar version for the pack kernel
__device__ __forceinline__ unsigned short bf16_rne_bits(float x) {
  unsigned u = __builtin_bit_cast(unsigned, x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
struct LimbFrag {  // B operand of one k16 step of one 32-feature tile
  u32x4 hi, mid, lo;
};

// (ReLU +) exact 3-limb split of registers 8s .. 8s+7 of an accumulator tile
template <bool RELU>
__device__ __forceinline__ LimbFrag split_step(const f32x16& t, int s8) {
  LimbFrag f;
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    float x0 = t[s8 + 2 * jp], x1 = t[s8 + 2 * jp + 1];
    if (RELU) { x0 = __builtin_fmaxf(x0, 0.f); x1 = __builtin_fmaxf(x1, 0.f); }
    const unsigned ph = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - bf16_lo_as_f32(ph), r1 = x1 - bf16_hi_as_f32(ph);
    const unsigned pm = cvt_pk_bf16(r0, r1);
    const float q0 = r0 - bf16_lo_as_f32(pm), q1 = r1 - bf16_hi_as_f32(pm);
    f.hi[jp] = ph; f.mid[jp] = pm; f.lo[jp] = cvt_pk_bf16(q0, q1);
  }
  return f;
}

// One (ReLU +) split of a register pair: the unit of VALU work interleaved between MFMA groups.
// STORE (training forward): the (ReLU'd) fp32 values -- the activation the backward pass needs -- also go to their plane
// rows, two stores per pair, so a tile's 16 stores ride between the MFMA groups of the chunk that pre-splits it.
// PRE: what happens to the fp32 values before the split -- 0 nothing, 1 ReLU (forward), 2 multiply by the forward's ReLU
// decision bits (backward: dZ = relu'(Z) * dH; `bits` holds the 16 decisions of this tile, bit r <-> register r).
template <int PRE, bool STORE = false>
__device__ __forceinline__ void split_pair(const f32x16& t, int pi /*0..7*/, LimbFrag (&f)[2], float* tile_plane = nullptr,
                                           const PlaneIO* io = nullptr, unsigned bits = 0u) {
  const int s = pi >> 2, jp = pi & 3;
  const int r0 = 8 * s + 2 * jp, r1 = r0 + 1;
  float x0 = t[r0], x1 = t[r1];
  if (PRE == 1) {  // plain v_max_f32 (fmaxf would add a canonicalising v_max in front of the real one)
    asm("v_max_f32 %0, 0, %1" : "=v"(x0) : "v"(x0));
    asm("v_max_f32 %0, 0, %1" : "=v"(x1) : "v"(x1));
  }
  if (PRE == 2) {
    x0 = ((bits >> r0) & 1u) ? x0 : 0.f;
    x1 = ((bits >> r1) & 1u) ? x1 : 0.f;
  }
  if constexpr (STORE) {
    *plane_addr(tile_plane, *io, (r0 & 3) + 8 * (r0 >> 2)) = x0;
    *plane_addr(tile_plane, *io, (r1 & 3) + 8 * (r1 >> 2)) = x1;
  }
  const unsigned ph = cvt_pk_bf16(x0, x1);
  const float r0f = x0 - bf16_lo_as_f32(ph), r1f = x1 - bf16_hi_as_f32(ph);
  const unsigned pm = cvt_pk_bf16(r0f, r1f);
  const float q0 = r0f - bf16_lo_as_f32(pm), q1 = r1f - bf16_hi_as_f32(pm);
  f[s].hi[jp] = ph; f[s].mid[jp] = pm; f[s].lo[jp] = cvt_pk_bf16(q0, q1);
}

// One 1-KiB-per-wave round of the LDS-DMA of chunk C (see issue_chunk): issued one per MFMA group so that the twelve
// address-setup + issue sequences ride between MFMAs instead of forming a serial block at the chunk boundary.
template <class Net>
__device__ __forceinline__ void issue_round(const Pipe& p, unsigned off, int slot, int r) {
  gbl_char* src = (gbl_char*)(p.stream + off);
  char* dst = p.ring + slot * Net::kSlotBytes + p.wave_off;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * 4096 + p.voff),
                                   (lds_void*)(dst + r * 4096), 16, 0, 0);
}

// Consumes chunk C with the B fragments `b` of the current input tile; meanwhile splits `next` (the input tile of chunk
// C+1) into `bn`, one register pair per MFMA group, so the VALU work rides in the shadow of the matrix pipe.
template <int C, int NT_OUT, bool HAS_NEXT, int PRE_NEXT, bool TRAIN = false, class Net = Bf16Net>
__device__ __forceinline__ void chunk_mma_bf16(Pipe& p, const LimbFrag (&b)[2], f32x16 (&out)[NT_OUT], const f32x16& next,
                                               LimbFrag (&bn)[2], float* next_plane = nullptr, const PlaneIO* io = nullptr,
                                               unsigned next_bits = 0u) {
  static_assert(Net::chunk_bytes(C) == NT_OUT * 6144, "chunk/out-tile mismatch");
  // acquire, with the first A fragments requested BEFORE the next chunk's DMA is issued: their LDS latency then overlaps
  // the twelve DMA issues instead of following them
  __syncthreads();
  p.slot ^= 1;
  const char* buf = p.ring + p.slot * Net::kSlotBytes + p.lane_off;
  // (round-1 experiments: a distance-2 fragment prefetch changes nothing (59.9 vs 58.6 ms per 65,536 x 193 launch);
  //  alternating two accumulators per group -- to dodge a dependent-accumulator latency -- measured
  //  slower than this single-accumulator chain; the six limb products of one output tile are issued back to back.)
  constexpr int NSTEP = 2 * NT_OUT;
  u32x4 ah = *reinterpret_cast<const u32x4*>(buf);
  u32x4 am = *reinterpret_cast<const u32x4*>(buf + 1024);
  u32x4 al = *reinterpret_cast<const u32x4*>(buf + 2048);
  constexpr int CN = (C + 1) % Net::kNumChunks;           // chunk streamed in while this one is consumed
  constexpr int ROUNDS = Net::chunk_bytes(CN) / 4096;     // 12 or 6
  unsigned dma_off = p.issue_off;
  asm volatile("" : "+s"(dma_off));
  p.issue_off = (CN == Net::kNumChunks - 1) ? 0u : dma_off + (unsigned)Net::chunk_bytes(CN);
#pragma unroll
  for (int i = 0; i < NSTEP; ++i) {
    const int s = i / NT_OUT, tp = i % NT_OUT;
    u32x4 nh = ah, nm = am, nl = al;
    if (i + 1 < NSTEP) {
      nh = *reinterpret_cast<const u32x4*>(buf + (i + 1) * 3072);
      nm = *reinterpret_cast<const u32x4*>(buf + (i + 1) * 3072 + 1024);
      nl = *reinterpret_cast<const u32x4*>(buf + (i + 1) * 3072 + 2048);
    }
    // keep the three reads of group i+1 ABOVE the six MFMAs of group i (hipcc otherwise sinks them next to their use and
    // exposes the LDS latency once per group: a 32-cycle MFMA covers far less of it than the fp32 kernel's 64-cycle one)
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc = out[tp];
    acc = mfma_bf16(al, b[s].hi, acc);   // smallest terms first
    acc = mfma_bf16(ah, b[s].lo, acc);
    acc = mfma_bf16(am, b[s].mid, acc);
    if (HAS_NEXT) {  // the 8 pair-splits of the next input tile, spread over the groups (one per group when there are >= 8)
      constexpr int PPG = (8 + NSTEP - 1) / NSTEP;
#pragma unroll
      for (int q = 0; q < PPG; ++q)
        if (i * PPG + q < 8) split_pair<PRE_NEXT, TRAIN>(next, i * PPG + q, bn, next_plane, io, next_bits);
    }
    // DMA rounds of the next chunk, spread over the groups (all issued well before this chunk ends)
    {
      constexpr int RPG = (ROUNDS + NSTEP - 1) / NSTEP;
#pragma unroll
      for (int q = 0; q < RPG; ++q)
        if (i * RPG + q < ROUNDS) issue_round<Net>(p, dma_off, p.slot ^ 1, i * RPG + q);
    }
    acc = mfma_bf16(am, b[s].hi, acc);
    acc = mfma_bf16(ah, b[s].mid, acc);
    acc = mfma_bf16(ah, b[s].hi, acc);
    out[tp] = acc;
    ah = nh; am = nm; al = nl;
  }
}

template <int PRE, bool STORE = false>
__device__ __forceinline__ void split_tile(const f32x16& t, LimbFrag (&f)[2], float* tile_plane = nullptr, const PlaneIO* io = nullptr,
                                           unsigned bits = 0u) {
#pragma unroll
  for (int pi = 0; pi < 8; ++pi) split_pair<PRE, STORE>(t, pi, f, tile_plane, io, bits);
}

// the 16 ReLU decisions of tile t inside a layer's mask word (relu_mask_bits layout)
__device__ __forceinline__ unsigned tile_bits(const u32x4 w, int t) { return (w[t >> 1] >> ((t & 1) * 16)) & 0xffffu; }

// 256 -> NT_OUT*32 layer over eight input tiles; `tail` = the tile consumed by the chunk that follows this layer's last
// chunk (next layer's first input, or an encoding tile), pre-split during the last chunk when HAS_TAIL.
// On entry `cur` holds the fragments of in[0]; on exit it holds the fragments of `tail` (if HAS_TAIL).
template <int CBASE, int NT_OUT, int PRE_IN, bool HAS_TAIL, int PRE_TAIL, bool TRAIN = false, bool STORE_TAIL = false, class Net = Bf16Net>
__device__ __forceinline__ void layer8_bf16(Pipe& p, LimbFrag (&cur)[2], const f32x16 (&in)[8], f32x16 (&out)[NT_OUT], const f32x16& tail,
                                            float* in_plane = nullptr, const PlaneIO* io = nullptr, int64_t tile_bytes = 0,
                                            float* tail_plane = nullptr, const u32x4 in_mask = u32x4{0u, 0u, 0u, 0u}) {
  LimbFrag nxt[2];
  auto tp = [&](int j) { return TRAIN ? reinterpret_cast<float*>(reinterpret_cast<char*>(in_plane) + j * tile_bytes) : nullptr; };
#define AON_BF_STEP(T)                                                                                                          \
  chunk_mma_bf16<CBASE + T, NT_OUT, true, PRE_IN, TRAIN, Net>(p, cur, out, in[T + 1], nxt, tp(T + 1), io, tile_bits(in_mask, T + 1)); \
  cur[0] = nxt[0]; cur[1] = nxt[1];
  AON_BF_STEP(0) AON_BF_STEP(1) AON_BF_STEP(2) AON_BF_STEP(3) AON_BF_STEP(4) AON_BF_STEP(5) AON_BF_STEP(6)
#undef AON_BF_STEP
  chunk_mma_bf16<CBASE + 7, NT_OUT, HAS_TAIL, PRE_TAIL, STORE_TAIL, Net>(p, cur, out, tail, nxt, tail_plane, io);
  if (HAS_TAIL) { cur[0] = nxt[0]; cur[1] = nxt[1]; }
}

// w . relu(x) over the features this lane holds
template <int NT>
__device__ __forceinline__ float head_partial_relu(const f32x16 (&x)[NT], const float* sm_w, int h) {
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(sm_w + 32 * t + 8 * g + 4 * h);
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        acc = __builtin_fmaf(w[cc], relu1(x[t][4 * g + cc]), acc);  // (no asm on MFMA results: aon_mlp_core.h, hazard rule)
      }
    }
  }
  return acc;
}

}  // namespace aon
