// Shared machinery of the fused, register-resident MLP kernels (vanilla: aon_mlp.hip, articulated: aon_mlp_art.hip).
// See the header comment of aon_mlp.hip for the mapping onto CDNA4.
#pragma once
#include "aon_common.h"

#include <type_traits>

namespace aon {

// Weight-stream pipeline of the fp32 kernels: two 64 KiB LDS slots, each holding a PAIR of consecutive chunks, and one
// workgroup barrier per pair (39 instead of 77 per pass for the vanilla network): the barrier itself, the wave skew it
// exposes and the LDS latency of the first fragment after it are paid half as often.  With an odd chunk count the last
// chunk forms a "pair" of one.  Chunk C streams in chunk C+2 (the same position of the next pair) between its own first
// MFMA groups; the last chunk of an odd-length stream brings in chunks 0 and 1 of the next pass, and its predecessor
// nothing.  (A three-slot variant with the barrier in the middle of each chunk and the first A fragment of the next chunk
// prefetched across the boundary was built and measured in round 1: correct, but 0.85-0.87 of the fp32-matrix peak --
// the mid-chunk barrier splits hipcc's MFMA/ds_read scheduling region -- so it was dropped.)
constexpr int kPairSlotBytes = 2 * kBigChunkBytes;
constexpr int kRingBytes = 2 * kPairSlotBytes;

template <class Net> constexpr int pair_offset(int C) { return (C & 1) ? Net::chunk_bytes(C - 1) : 0; }
template <class Net> constexpr int dma_count(int C) {
  return (Net::kNumChunks & 1) ? (C == Net::kNumChunks - 1 ? 2 : (C == Net::kNumChunks - 2 ? 0 : 1)) : 1;
}
template <class Net> constexpr int dma_target(int C, int k) {
  return ((Net::kNumChunks & 1) && C == Net::kNumChunks - 1) ? k : (C + 2) % Net::kNumChunks;
}

// A net may leave a gap in its stream in front of a chunk (`static constexpr int skip_before(int c)`, bytes): the per-ray-bias form of the
// folded vanilla network reads the same buffer as the chunk form and steps over the view-encoding chunk.
template <class Net, class = void> struct NetSkip { static constexpr int at(int) { return 0; } };
template <class Net> struct NetSkip<Net, std::void_t<decltype(Net::skip_before(0))>> { static constexpr int at(int c) { return Net::skip_before(c); } };

struct Pipe {
  const char* stream;  // packed stream base (wave-uniform -> SGPR pair)
  const char* next_stream;  // stream of this workgroup's NEXT pass (round 4: a launch may carry two segments with different networks,
                            // e.g. the fine level of one ray range and the coarse level of another; the chunks that wrap around the end
                            // of the stream -- the first pair of the next pass -- are fetched from here).  == stream in one-segment launches.
  char* ring;          // LDS ring base
  unsigned voff;       // this lane's byte offset inside a 4 KiB round: wave*1024 + lane*16
  int wave_off;        // wave*1024
  int lane_off;        // lane*16
  int slot;            // slot holding the chunk being consumed
  unsigned issue_off;  // byte offset (in the stream) of the next chunk to issue
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) char gbl_char;

// Round 4: the weight stream is fetched with BUFFER loads to LDS (`buffer_load_dwordx4 v_off, s[rsrc], s_soff offen lds`): the stream is a
// raw buffer whose descriptor sits in four SGPRs, the lane's slice is ONE 32-bit VGPR offset that never changes, and everything that
// does change -- chunk, round -- is a scalar offset.  The `global_load_lds` form of rounds 1-3 took a 64-bit per-lane address: hipcc
// hoisted (stream + lane offset) into a VGPR pair and paid a 64-bit VALU add per DMA, plus a v_readfirstlane + s_mov to bring the
// LDS destination (derived from threadIdx, so "divergent") into M0 -- ~1,200 of the ~5,100 vector instructions of a pass of the
// headline kernel that were not MFMAs (tools/isa_mix.py).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stream_rsrc(const char* stream) {
  // raw buffer, stride 0, no range limit below 2 GiB, gfx9 data format word (the 0x00020000 every CDNA kernel library uses)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(stream), 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ void dma_1k(const char* stream, unsigned soff, unsigned voff, char* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(stream_rsrc(stream), (lds_void*)lds_dst, 16, (int)voff, (int)soff, 0, 0);
}

// LDS-DMA one chunk: global (SGPR base + per-lane VGPR offset) -> LDS (M0 base + lane*16), 1 KiB per wave per
// instruction.  The stream offset is kept as an opaque loop-carried scalar so that the several hundred
// distinct chunk addresses are recomputed with one s_add instead of being hoisted out of the pass loop.
template <class Net, int C>
__device__ __forceinline__ void issue_chunk(Pipe& p, int slot) {
  constexpr int rounds = Net::chunk_bytes(C) / 4096;
  unsigned off = p.issue_off;
  asm volatile("" : "+s"(off));
  char* dst = p.ring + slot * Net::kSlotBytes + p.wave_off;  // wave-uniform; hardware adds lane*16
#pragma unroll
  for (int r = 0; r < rounds; ++r) dma_1k(p.stream, off + (unsigned)(r * 4096), p.voff, dst + r * 4096);
  p.issue_off = (C == Net::kNumChunks - 1) ? 0u : off + (unsigned)Net::chunk_bytes(C);
}

// Kernel prologue: the first pair (fp32 nets) / chunk 0 (bf16x3 net, which runs its own single-chunk schedule) in flight.
// The caller's LDS writes (resident small vectors) are published by the barrier inside.
template <class Net>
__device__ __forceinline__ void pipe_init(Pipe& p, const char* stream, char* ring, int wave, int lane) {
  p.stream = stream; p.next_stream = stream; p.ring = ring;
  // the wave's 1 KiB slice of a DMA round is WAVE-UNIFORM: kept on the scalar unit (round 4).  Derived from threadIdx it lived in a
  // VGPR, and every one of the ~600 DMA instructions of a pass paid a v_readfirstlane + s_mov to get its LDS destination into M0,
  // with a dozen pre-computed destinations parked in VGPRs (tools/isa_mix.py).
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  p.voff = (unsigned)(wave * 1024 + lane * 16);
  p.wave_off = wave_s * 1024; p.lane_off = lane * 16;
  p.slot = 1; p.issue_off = 0;  // the first acquire flips to slot 0
  issue_chunk<Net, 0>(p, 0);
  if constexpr (Net::kPair) {
    char* dst = p.ring + Net::chunk_bytes(0) + p.wave_off;
#pragma unroll
    for (int r = 0; r < Net::chunk_bytes(1) / 4096; ++r)
      dma_1k(p.stream, (unsigned)(Net::chunk_bytes(0) + r * 4096), p.voff, dst + r * 4096);
    p.issue_off = (unsigned)(Net::chunk_bytes(0) + Net::chunk_bytes(1));
  }
  __syncthreads();
}

// Start of chunk C: the first chunk of a pair waits for the pair (DMA issued one pair earlier) and releases the other
// slot.  Returns the stream offset of this chunk's DMA targets, which chunk_mma issues between its first MFMA groups: a
// wave issues in order, so the address-setup + DMA instructions issued as one block at the boundary would hold back the
// MFMAs behind them while the memory pipeline accepts them; spread out they cost nothing, and being early in the chunk
// they have landed long before the next barrier's vmcnt(0).
// first chunk whose DMA targets belong to the NEXT pass (chunk C fetches C + 2; the last chunk of an odd stream fetches 0 and 1)
template <class Net> constexpr int first_wrapping_chunk() { return (Net::kNumChunks & 1) ? Net::kNumChunks - 1 : Net::kNumChunks - 2; }

template <class Net, int C>
__device__ __forceinline__ unsigned acquire(Pipe& p) {
  // every DMA issued from here to the end of the pass fetches the next pass's first pair: ONE stream base is live at any time (a
  // second base selected per chunk kept two 64-bit per-lane addresses alive and tipped the vanilla chain into 3.5 KB of scratch)
  if constexpr (C == first_wrapping_chunk<Net>()) p.stream = p.next_stream;
  if constexpr ((C & 1) == 0) {
#if defined(AON_EXP_NOVMWAIT)    // timing experiment only (WRONG results): the barrier without waiting for this wave's DMA
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#elif defined(AON_EXP_NOBARRIER) // timing experiment only (WRONG results): the DMA wait without the workgroup barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
    __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)) + workgroup barrier
#endif
    p.slot ^= 1;
  }
  unsigned off = p.issue_off;
  asm volatile("" : "+s"(off));
  constexpr int n = dma_count<Net>(C);
  if constexpr (n > 0) {
    static_assert(n == 1 || NetSkip<Net>::at(dma_target<Net>(C, n - 1)) == 0, "no gap in front of the second target of a two-target chunk");
    static_assert(NetSkip<Net>::at(0) == 0 && NetSkip<Net>::at(1) == 0, "no gap in front of the first pair (pipe_init)");
    if constexpr (NetSkip<Net>::at(dma_target<Net>(C, 0)) != 0) off += (unsigned)NetSkip<Net>::at(dma_target<Net>(C, 0));
    constexpr int last = dma_target<Net>(C, n - 1);
    constexpr int bytes = Net::chunk_bytes(dma_target<Net>(C, 0)) + (n > 1 ? Net::chunk_bytes(dma_target<Net>(C, 1)) : 0);
    p.issue_off = (last == Net::kNumChunks - 1) ? 0u : off + (unsigned)bytes;
  }
  return off;
}

// round r of the DMA targets of chunk C (targets are consecutive in the stream; each lands at its pair position in the
// other slot)
template <class Net, int C>
__device__ __forceinline__ void dma_round(const Pipe& p, unsigned off, int r) {
  constexpr int T0 = dma_target<Net>(C, 0);
  constexpr int R0 = Net::chunk_bytes(T0) / 4096;
  // (from the first wrapping chunk on, p.stream IS the next pass's stream: acquire)
  char* slot = p.ring + (p.slot ^ 1) * kPairSlotBytes + p.wave_off;
  char* dst = r < R0 ? slot + pair_offset<Net>(T0) + r * 4096
                     : slot + pair_offset<Net>(dma_target<Net>(C, 1)) + (r - R0) * 4096;
  dma_1k(p.stream, off + (unsigned)(r * 4096), p.voff, dst);
}

// out[Tp] += W_chunk[Tp] * in   for one 32-feature input tile held in accumulator layout.
// The A-operand reads are software-pipelined one ds_read_b128 (= 4 MFMA steps, 256 matrix-pipe cycles) ahead.
// SIDE JOB: `side(i)` is called for i = 0 .. max(16, NSTEP)-1, spread over the chunk's steps (a step = a group of four
// MFMAs, 256 matrix-pipe cycles), between the MFMA groups of the chunk.  The training kernels hang their per-value bookkeeping on it -- storing the input tile
// to its activation / gradient plane, collecting or applying ReLU decision bits -- one value per step, so that this VALU
// and store work drains in the shadow of the matrix pipe instead of in a burst at a layer boundary, where no MFMA is in
// flight (and in front of the next chunk barrier, whose s_waitcnt vmcnt(0) also waits for every outstanding store).
// Measured round 2 (tools/kernel_bench.py, 4096 x 193 samples, experiment builds with the stores / the mask arithmetic compiled
// out): producer-side bursts cost the articulated training forward 25 % (stores) + 8 % (masks), its backward chain 18 %, the
// vanilla forward 5.5 % + 4.7 %; as side jobs the articulated forward went 10.67 -> 9.29 ms, its chain 10.18 -> 9.50 ms.
// What is left of the store cost (articulated forward 9.29 ms, 8.33 without stores) did NOT respond to: storing only in the
// first chunk of each barrier pair so the pair's vmcnt(0) never waits on a fresh store (9.25); pinning the side job outside
// the four-MFMA accumulator chain with sched_barrier (9.37-9.40); 16-byte stores into a [pass][feature/4][sample][4] layout,
// 512 contiguous bytes per half-wave (9.02; with `nt` 10.2, with write-through `sc1` 11.9); it halves when every pass writes
// the same cache-resident 1024 columns (8.83), i.e. about half of it is the L2 -> HBM write path itself.
struct NoSide {
  __device__ __forceinline__ void operator()(int) const {}
};

// ZERO_C: `out` holds nothing yet -- the first MFMA into each output tile takes the constant 0 as its C operand instead of the
// tile, so a layer that starts from zero (the backward chains: dH = W^T dZ) needs no 128 accumulator writes per wave in front
// of it, where no MFMA is in flight.  Same bits: fma(a, b, 0) is what the first step on a zeroed tile computes.
template <class Net, int C, int NT_OUT, int NREG, class Side = NoSide, bool ZERO_C = false>
__device__ __forceinline__ void chunk_mma(Pipe& p, const f32x16& in, f32x16 (&out)[NT_OUT], Side side = Side{}) {
  static_assert(Net::chunk_bytes(C) == NT_OUT * 4096, "chunk/out-tile mismatch");
  static_assert(NREG % 2 == 0 && NREG > 12, "register count");
  const unsigned dma_off = acquire<Net, C>(p);
  constexpr int NDMA = dma_count<Net>(C);
  constexpr int ROUNDS = NDMA == 0 ? 0 : (Net::chunk_bytes(dma_target<Net>(C, 0)) + (NDMA > 1 ? Net::chunk_bytes(dma_target<Net>(C, 1)) : 0)) / 4096;
  const char* buf = p.ring + p.slot * kPairSlotBytes + pair_offset<Net>(C) + p.lane_off;
  constexpr int NQ = (NREG + 3) / 4;
  constexpr int NSTEP = NQ * NT_OUT;
  // (round-1 experiments on the forward kernels, all slower than the compiler's own interleave of this distance-1 form at
  // 0.915 of peak: a distance-2 prefetch pinned by sched_barrier(0) per step 0.900; this distance-1 read pinned above its
  // four MFMAs 0.891 -- which does pay in the vanilla backward chain, see AON_PIN_PREFETCH; ReLU applied lazily here on
  // the input tile instead of as a block between layers 0.869 -- 16 more live registers.)
  // (Round 3: hipcc sinks the read of step i+1 from here to just in front of its own use -- `ds_read_b128; s_waitcnt lgkmcnt(0);
  // 4 x v_mfma` -- in 32 % of the steps of the inference kernel and 71 % of the articulated chain's, and after any LDS-DMA
  // instruction its waits are lgkmcnt(0) whatever is in flight (SIInsertWaitcnts' pending-FLAT state).  It does not matter: with
  // the reads as inline asm, distance 1 enforced and `s_waitcnt lgkmcnt(1)` placed by hand the loop looked ideal in the ISA and
  // ran SLOWER, 9.08 vs 8.65 ms for the one kernel that did not spill (the others lost 230-1,600 B/lane to the address
  // registers): the four queued MFMAs of the previous step cover the LDS latency of a read issued right behind them.)
  f32x4 a_cur = *reinterpret_cast<const f32x4*>(buf);
#pragma unroll
  for (int i = 0; i < NSTEP; ++i) {
    const int q = i / NT_OUT, tp = i % NT_OUT;
    f32x4 a_nxt = a_cur;
    if (i + 1 < NSTEP) a_nxt = *reinterpret_cast<const f32x4*>(buf + (i + 1) * 1024);
    static_assert(ROUNDS <= NSTEP, "one DMA round per step");
    if constexpr (ROUNDS > 0) {
      if (i < ROUNDS) dma_round<Net, C>(p, dma_off, i);
    }
    // a side job has 16 slots (one per register of the input tile); chunks with fewer than 16 steps (two output tiles)
    // take several per step
    constexpr int SIDE_PER_STEP = NSTEP >= 16 ? 1 : (16 + NSTEP - 1) / NSTEP;
#pragma unroll
    for (int k = 0; k < SIDE_PER_STEP; ++k) side(i * SIDE_PER_STEP + k);
    // (Round 3 experiment: a sched_barrier here that keeps this step's vector-memory instructions -- DMA round, plane store --
    // inside the step while letting MFMA / VALU / LDS cross.  hipcc does sink a chunk's stores to its end, in front of the next
    // pair's barrier, but pinning them changed nothing: forward 8.66 vs 8.61 ms, chains within 0.1 %.  Not kept.)
#ifdef AON_PIN_PREFETCH   // per translation unit (build.py): keeps the read of step i+1 above the four MFMAs of step i
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      if (4 * q + cc < NREG) {
        if (ZERO_C && q == 0 && cc == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          out[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[cc], in[4 * q + cc], zero, 0, 0, 0);
        } else {
          out[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[cc], in[4 * q + cc], out[tp], 0, 0, 0);
        }
      }
    }
    a_cur = a_nxt;
  }
}

template <int NT_OUT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NT_OUT], const float* sm_bias, int h) {
#ifdef AON_EXP_NOBIAS    // timing experiment only (WRONG results): what the per-layer bias reads cost
#pragma unroll
  for (int tp = 0; tp < NT_OUT; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
  return;
#endif
#pragma unroll
  for (int tp = 0; tp < NT_OUT; ++tp) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(sm_bias + 32 * tp + 8 * g + 4 * h);
      acc[tp][4 * g + 0] = b[0]; acc[tp][4 * g + 1] = b[1]; acc[tp][4 * g + 2] = b[2]; acc[tp][4 * g + 3] = b[3];
    }
  }
}

// HAZARD RULE for this code base: no inline-asm INSTRUCTION may read a register written by an MFMA.  The hardware needs up
// to 18 wait states between an MFMA's VGPR write and a VALU read of it; hipcc inserts them for instructions it knows, but
// an asm block is opaque to its hazard recognizer.  Round 1's `asm("v_max_f32 ...")` ReLU and round 2's first mask helpers
// broke this: harmless while the allocator kept accumulators in AGPRs (the v_accvgpr_read in front is hazard-checked), wrong
// results as soon as a tile was allocated in architectural VGPRs and the scheduler moved the asm next to the producing MFMA
// (articulated training forward, view layer 1, output tile 0: tests/diag/diag_art_planes.py).  Empty asm statements
// ("value barriers") on VALU-produced values are fine: they emit nothing.
//
// ReLU as ONE integer instruction the compiler knows: max(int(bits(x)), 0).  Non-negative floats are non-negative integers
// and keep their bits; anything with the sign bit set (negative values, -0) is a negative integer and becomes +0.  No
// canonicalising second v_max as with fmaxf(), and relu(NaN) stays NaN as in torch.
__device__ __forceinline__ float relu1(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// ReLU on accumulator tiles.  (An AGPR->AGPR variant -- v_accvgpr_read / v_max / v_accvgpr_write in one asm block with
// "a" constraints, which frees ~75 arch VGPRs -- was measured in round 1: 0.875 of peak against 0.908 for this form.)
template <int NT>
__device__ __forceinline__ void relu_tiles(f32x16 (&x)[NT]) {
#ifdef AON_EXP_NORELU    // timing experiment only (WRONG results): what the per-layer ReLU bursts cost
  return;
#endif
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float y = relu1(x[t][r]);  // the instruction that reads the accumulator is the compiler's (hazard rule)
      asm("" : "+v"(y));         // ... and its result is opaque, as the asm ReLU's was (see kernel_resources notes in DESIGN.md)
      x[t][r] = y;
    }
  }
}

// NT_IN*32 -> NT_OUT*32 layer, input/output both in accumulator layout; chunks CBASE .. CBASE+NT_IN-1.
// `side_of(j)` returns the side job of the chunk that consumes input tile j (see chunk_mma).
struct NoSideOf {
  __device__ __forceinline__ NoSide operator()(int) const { return NoSide{}; }
};

template <class Net, int CBASE, int NT_IN, int NT_OUT, class SideOf = NoSideOf, bool ZERO_OUT = false>
__device__ __forceinline__ void dense_layer(Pipe& p, const f32x16 (&in)[NT_IN], f32x16 (&out)[NT_OUT], SideOf side_of = SideOf{}) {
  chunk_mma<Net, CBASE + 0, NT_OUT, 16, decltype(side_of(0)), ZERO_OUT>(p, in[0], out, side_of(0));
  chunk_mma<Net, CBASE + 1, NT_OUT, 16>(p, in[1], out, side_of(1));
  chunk_mma<Net, CBASE + 2, NT_OUT, 16>(p, in[2], out, side_of(2));
  chunk_mma<Net, CBASE + 3, NT_OUT, 16>(p, in[3], out, side_of(3));
  if constexpr (NT_IN == 8) {
    chunk_mma<Net, CBASE + 4, NT_OUT, 16>(p, in[4], out, side_of(4));
    chunk_mma<Net, CBASE + 5, NT_OUT, 16>(p, in[5], out, side_of(5));
    chunk_mma<Net, CBASE + 6, NT_OUT, 16>(p, in[6], out, side_of(6));
    chunk_mma<Net, CBASE + 7, NT_OUT, 16>(p, in[7], out, side_of(7));
  }
}

// per-lane partial of  w . x  over the features this lane holds (NT tiles of 32 features)
template <int NT>
__device__ __forceinline__ float head_partial(const f32x16 (&x)[NT], const float* sm_w, int h) {
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(sm_w + 32 * t + 8 * g + 4 * h);
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) acc = __builtin_fmaf(w[cc], x[t][4 * g + cc], acc);
    }
  }
  return acc;
}


// Training planes, STEP-MAJOR (round 3).  A *step* is the 32 samples one wave carries through the network in a pass (pass p,
// wave w -> step 4p + w); a *unit* is 16 bytes = FOUR consecutive feature rows of ONE sample.  Feature row f of sample n lives at
//   float offset  ((n >> 5) * (rows / 4) + (f >> 2)) * 128 + (n & 31) * 4 + (f & 3)          [step][f / 4][sample][f % 4]
// * the forward / the backward chain hold rows 8g + 4h + {0..3} of a tile in four consecutive accumulator registers: ONE 16-byte
//   store per lane, the 64 lanes of a wave writing 1 KiB contiguous (round 2: two 128-byte row segments per 4-byte store,
//   3,456 rows x 512 B scattered per pass);
// * the weight-gradient kernel fetches an operand of a step -- the (rows_of_layer / 4) x 512 B of consecutive units -- as one
//   contiguous run with LDS-DMA and reads a unit per lane as 4 features x 1 sample: 16 MFMAs per pair of ds_read_b128.
// Addressing:  [wave-uniform 64-bit base of the step, re-made opaque every pass so it lives on the scalar unit]  +
//              [compile-time row offset]  +  [one 32-bit per-lane offset h * 512 + m * 16 that never changes].
struct PlaneIO {
  char* base;      // planes + step * rows * 128 bytes (wave-uniform)
  unsigned voff;   // h * 512 + m * 16: unit (row group + h, sample m)
  unsigned soff;   // m * 16: this lane's sample inside a unit row (scalar accesses)
};

__device__ __forceinline__ PlaneIO make_plane_io(const float* planes, int rows, int64_t step, int m, int h) {
  PlaneIO io;
  int64_t sb = step * (int64_t)rows * 128;
  asm volatile("" : "+s"(sb));
  io.base = const_cast<char*>(reinterpret_cast<const char*>(planes)) + sb;
  io.voff = (unsigned)(h * 512 + m * 16);
  io.soff = (unsigned)(m * 16);
  return io;
}

// registers 4g .. 4g+3 of the tile whose first plane row is `row` (a multiple of 32): rows row + 8g + 4h + {0..3}
__device__ __forceinline__ f32x4* quad_ptr(const PlaneIO& io, int row, int g) {
  return reinterpret_cast<f32x4*>(io.base + (int64_t)(row * 128 + g * 1024) + io.voff);
}
// one row of this lane's sample (any row)
__device__ __forceinline__ float* row_ptr(const PlaneIO& io, int row) {
  return reinterpret_cast<float*>(io.base + (int64_t)((row >> 2) * 512 + (row & 3) * 4) + io.soff);
}
template <bool STAGE = true>
__device__ __forceinline__ void store_quad(const PlaneIO& io, int row, int g, const f32x16& t) {
#ifndef AON_EXP_NOSTORE   // experiment builds only (tools/exp_train.sh): what the plane stores cost
  f32x4 v; v[0] = t[4 * g]; v[1] = t[4 * g + 1]; v[2] = t[4 * g + 2]; v[3] = t[4 * g + 3];
  // (STAGE = false, round 4: the forward's tiles are post-ReLU values that already live in architectural VGPRs; the staging asm, which
  // "modifies" its operand, forced a copy of every quad there -- 770 v_mov_b64 per pass of the articulated training forward)
  if constexpr (STAGE)
  // The store's data is staged in architectural VGPRs (four v_accvgpr_read where the tile sits in AGPRs, as the gradient
  // tiles of the backward chains do): a global_store whose data operand is an AGPR range reads it while MFMAs are streaming
  // their accumulators through the same register banks and holds the wave's issue port meanwhile.  Measured round 3
  // (tools/exp_train.sh): articulated backward chain 9.60 -> 9.03 ms (4096 x 193 samples), 3.42 -> 3.25 (x 65).
  asm volatile("" : "+v"(v));
  // ... and it is a streaming (`nt`) store: the 10.9 GB of planes a level writes are read back once, much later, by the
  // weight-gradient kernel; written with the default policy they push the 2.8-3.3 MB weight stream, which every workgroup
  // re-reads every pass, out of the 4 MB L2 of its XCD.  (Round 2 measured `nt` as a loss on the feature-major layout, 4-byte
  // stores to 3,456 scattered rows; on 1 KiB contiguous units: forward 9.04 -> 8.67 ms, chain 9.01 -> 8.68 ms.)
#ifdef AON_EXP_NO_NT
  *quad_ptr(io, row, g) = v;
#else
  __builtin_nontemporal_store(v, quad_ptr(io, row, g));
#endif
#endif
}

// ReLU masks of one layer as bits (bit (t&1)*16 + r of word t>>1 <-> tile t, register r): 16 bytes per lane per layer,
// written lane-linearly (slot = pass*256 + tid) by the training forward and read back by the backward chain, instead of
// re-reading 128 activation values per lane per layer.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Address of a lane's 16-byte decision-bit word: masks[slot][pass * 256 + tid].  [uniform slot base on the scalar unit] +
// [one 32-bit per-lane byte offset, made opaque once per pass]: written as one 64-bit per-lane index, the loop-invariant
// part (slot * Np * 2 + tid) of every slot is hoisted out of the pass loop into its own register pair.
__device__ __forceinline__ unsigned mask_lane_off(int pass, int tid) {
  unsigned off = (unsigned)(pass * 256 + tid) * 16u;
  asm volatile("" : "+v"(off));
  return off;
}
template <class T>
__device__ __forceinline__ T* mask_ptr(T* masks, int64_t Np, int slot, unsigned lane_off) {
  int64_t sb = (int64_t)slot * Np * 32;
  asm volatile("" : "+s"(sb));
  using C = std::conditional_t<std::is_const_v<T>, const char, char>;
  return reinterpret_cast<T*>(reinterpret_cast<C*>(masks) + sb + lane_off);
}

// Decision bits, two plain VALU instructions per value in both directions, no SGPR / VCC (a v_cmp + v_cndmask pair costs
// the compare, the select, an or, and on gfx950 two wait states between a VALU SGPR write and its VALU read).  The
// arithmetic is left to the compiler (hazard rule above); where it would fold the sequence back into compare + select an
// empty asm makes the intermediate opaque.
//   forward, on a POST-ReLU value y (>= +0, never -0: relu1):  w <- (w << 1) | (y > 0)  =  alignbit(w, 0 - bits(y), 31),
//             because 0 - b has its top bit set exactly for 0 < b <= 0x7fffffff.  Values are pushed in register order, tile
//             2k then tile 2k+1, into word k; 32 pushes later the first value sits in bit 31, so the finished word is
//             bit-reversed once (mask_word_finish) into the stored layout: bit (t&1)*16 + r of word t>>1 <-> tile t, register r.
//   backward  m = sign-extended one-bit field of w at pos (0 or 0xffffffff);  dz = dh & m
__device__ __forceinline__ unsigned mask_push_post(unsigned w, float y_post_relu) {
#ifdef AON_EXP_NOMASK     // experiment builds only: what the decision-bit arithmetic costs
  return w;
#else
  const unsigned nb = 0u - __builtin_bit_cast(unsigned, y_post_relu);
  return __builtin_amdgcn_alignbit(w, nb, 31);
#endif
}
__device__ __forceinline__ unsigned mask_word_finish(unsigned w) { return __builtin_bitreverse32(w); }

__device__ __forceinline__ float mask_apply(unsigned w, float dh, int pos) {
  int m = __builtin_amdgcn_sbfe((int)w, (unsigned)pos, 1u);  // v_bfe_i32: 0 or 0xffffffff
  asm("" : "+v"(m));
  return __builtin_bit_cast(float, __builtin_bit_cast(int, dh) & m);
}

template <int NT>
__device__ __forceinline__ u32x4 relu_mask_bits(const f32x16 (&x)[NT]) {
  u32x4 w = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) w[t >> 1] |= (x[t][r] > 0.f ? 1u : 0u) << ((t & 1) * 16 + r);  // any x (pre- or post-ReLU)
  }
  return w;
}

template <int NT>
__device__ __forceinline__ void apply_mask_bits(f32x16 (&x)[NT], const u32x4 w) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = mask_apply(w[t >> 1], x[t][r], (t & 1) * 16 + r);
  }
}

__device__ __forceinline__ void apply_mask_tile(f32x16& x, const u32x4 w, int t) {
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = mask_apply(w[t >> 1], x[r], (t & 1) * 16 + r);
}

// Side jobs of the backward data chains (see chunk_mma).  The chunk that consumes gradient tile j of a layer
//   * stores it -- it IS the pre-activation gradient dZ the weight-gradient kernels read -- to its plane rows, and
//   * [MASKED] turns tile j+1 from dH into dZ = relu'(Z) . dH with the forward's decision bits,
// one value per MFMA group; tile 0 is masked by the caller before the layer starts (16 values).
// STORE = false: the chunks that consume the tiles only turn the next one into dZ; their stores ride on another layer's chunks
// (StoreSideOf).  Used where the consuming chunks are TINY (the articulated chain's encoding chunks: 2 output tiles = 8 MFMA groups =
// 0.9 us each, a workgroup barrier every 1.8 us): vmcnt retires in order, so the barrier's wait for the next pair's weight DMA also
// waits for every plane store issued before that DMA, and a streaming store's acknowledgement takes longer than such a chunk lasts.
template <int NT, bool MASKED, bool STORE = true>
struct BwdSideOf {
  f32x16 (&tiles)[NT];
  int row;             // plane row of tile 0 in the gradient planes
  const PlaneIO& io;
  const u32x4& mk;
  // `touch`: the NEXT layer's decision bits, fetched at the start of this layer.  vmcnt retires in order, so the wait for a load is
  // also a wait for every plane store issued before it -- and left to its first use (the next layer's start) that wait sits right
  // behind this layer's last stores, whose acknowledgements take microseconds.  The second chunk's last slot "uses" the word
  // (an empty asm): the wait lands there, two chunks behind the load, when everything older has long been acknowledged and the
  // stores issued since do not matter (round 4: the articulated chain lost ~40 us per pass = 12 % to sixteen such waits).
  const u32x4* touch = nullptr;
  __device__ __forceinline__ auto operator()(int j) const {
    f32x16 (&t)[NT] = tiles;
    const int trow = row + 32 * j;
    const PlaneIO& pio = io;
    const u32x4& m = mk;
    const u32x4* tch = touch;
    return [&t, trow, &pio, &m, j, tch](int i) {
      if (j == 1 && i == 15 && tch) asm volatile("" ::"v"(*tch));
      if (i < 16) {
        // one 16-byte store per four slots.  A MASKED tile was turned into dZ by the previous chunk's side job (or by the caller, tile 0):
        // its values already sit in architectural VGPRs, staging them again is a copy (round 4: 694 v_mov_b64 per pass of the
        // articulated chain); an unmasked tile (d bottleneck) comes straight from the accumulators and is staged
        // (per translation unit: in the vanilla chain hipcc keeps masked tiles in AGPRs -- 192 of its stores would read them directly --
        // so aon_train.hip defines AON_CHAIN_STAGE_MASKED and stages every quad as before)
        if constexpr (STORE) {
#ifdef AON_CHAIN_STAGE_MASKED
          if ((i & 3) == 0) store_quad<true>(pio, trow, i >> 2, t[j]);
#else
          if ((i & 3) == 0) store_quad<!MASKED>(pio, trow, i >> 2, t[j]);
#endif
        }
        if constexpr (MASKED) {
          if (j + 1 < NT) {
            float z = mask_apply(m[(j + 1) >> 1], t[j + 1][i], ((j + 1) & 1) * 16 + i);
#ifdef AON_EXP_MASK_VGPR   // experiment: the masked gradient lives in an architectural VGPR from here on (B operand + store source)
            asm volatile("" : "+v"(z));
#endif
            t[j + 1][i] = z;
          }
        }
      }
    };
  }
};

// side job that only stores already-masked tiles (their masking rode on the chunks of BwdSideOf<.., true, false>)
template <int NT>
struct StoreSideOf {
  f32x16 (&tiles)[NT];
  int row;
  const PlaneIO& io;
  __device__ __forceinline__ auto operator()(int j) const {
    f32x16 (&t)[NT] = tiles;
    const int trow = row + 32 * j;
    const PlaneIO& pio = io;
    return [&t, trow, &pio, j](int i) {
      if (i < 16 && (i & 3) == 0) store_quad<false>(pio, trow, i >> 2, t[j]);
    };
  }
};

template <int NT>
__device__ __forceinline__ void store_plane(const f32x16 (&x)[NT], const PlaneIO& io, int row) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) store_quad(io, row + 32 * t, g, x[t]);
  }
}

template <int NT>
__device__ __forceinline__ void load_plane(f32x16 (&x)[NT], const PlaneIO& io, int row) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *quad_ptr(io, row + 32 * t, g);
      x[t][4 * g] = v[0]; x[t][4 * g + 1] = v[1]; x[t][4 * g + 2] = v[2]; x[t][4 * g + 3] = v[3];
    }
  }
}

// encodings are held in a permuted register order (posenc_col / viewenc_col); planes use the reference's columns, so these
// rows are written one value at a time (62 + 26 four-byte stores per sample against 3,400 rows in 16-byte stores).  The row of
// a register depends on the half-wave (sin rows for h = 0, sin(. + pi/2) rows for h = 1): both byte offsets are compile-time
// constants and the lane picks one.
__device__ __forceinline__ unsigned enc_row_off(int row) { return (unsigned)((row >> 2) * 512 + (row & 3) * 4); }

// (`h` is re-made opaque at every call: the selects below are loop-invariant per lane, and hoisted out of the pass loop each
// of the 44 offsets becomes a 64-bit register pair the allocator then spills -- 26 scratch instructions per pass.)
__device__ __forceinline__ void store_pos_enc_plane(const f32x16 (&E)[2], const PlaneIO& io, int row0, int h) {
  asm volatile("" : "+v"(h));
  char* base = io.base + io.soff;
#pragma unroll
  for (int rho = 0; rho < 30; ++rho) {
    const unsigned off = h ? enc_row_off(row0 + 33 + rho) : enc_row_off(row0 + 3 + rho);
    *reinterpret_cast<float*>(base + off) = E[rho >> 4][rho & 15];
  }
  const unsigned off_id = h ? enc_row_off(row0 + 2) : enc_row_off(row0);
  *reinterpret_cast<float*>(base + off_id) = E[1][14];
  if (!h) *reinterpret_cast<float*>(base + enc_row_off(row0 + 1)) = E[1][15];
}

__device__ __forceinline__ void store_view_enc_plane(const f32x16& V, const PlaneIO& io, int row0, int h) {
  asm volatile("" : "+v"(h));
  char* base = io.base + io.soff;
#pragma unroll
  for (int rho = 0; rho < 12; ++rho) {
    const unsigned off = h ? enc_row_off(row0 + 15 + rho) : enc_row_off(row0 + 3 + rho);
    *reinterpret_cast<float*>(base + off) = V[rho];
  }
  const unsigned off_id = h ? enc_row_off(row0 + 2) : enc_row_off(row0);
  *reinterpret_cast<float*>(base + off_id) = V[12];
  if (!h) *reinterpret_cast<float*>(base + enc_row_off(row0 + 1)) = V[13];
}

// Positional / view encodings directly in accumulator (= next layer's B operand) layout: lanes 0-31 hold the sin
// features, lanes 32-63 the sin(. + fp32(pi/2)) features of the same sample; the identity features ride in the
// last registers (pack kernels: posenc_col / viewenc_col).  helper.py:136-140.
// L / Lv: frequency levels of a network's encodings (max_deg_point - min_deg_point <= 10, deg_view <= 4).  The streams always have the
// 63 / 27-wide slots of the default geometry; a network with fewer levels leaves the slots of the missing levels at zero weight.
__device__ __forceinline__ int pos_col_in(int c63, int L) {   // column of the 63-slot layout -> column of the (3 + 6 L)-wide weight
  if (c63 < 3) return c63;
  const bool second = c63 >= 33;
  const int e = second ? c63 - 33 : c63 - 3;
  return e / 3 < L ? 3 + e + (second ? 3 * L : 0) : -1;
}
__device__ __forceinline__ int view_col_in(int c27, int Lv) {
  if (c27 < 3) return c27;
  const bool second = c27 >= 15;
  const int e = second ? c27 - 15 : c27 - 3;
  return e / 3 < Lv ? 3 + e + (second ? 3 * Lv : 0) : -1;
}

// pos_enc with RUN-TIME scales (round 4: the articulated network at other encoding degrees): scale[l] = 2^(min_deg + l) for the levels
// the network has, 0 for the slots it lacks (their weights are zero; a zero argument keeps the slot finite) -- ten floats of the
// per-call small block in LDS.  x * 2^(min_deg + l) is ONE exact multiplication, as helper.py:137-138 does it.
__device__ __forceinline__ void encode_pos_scaled(const float (&x)[3], int h, const float* sm_scale, f32x16 (&E)[2]) {
  const float phase = h ? AON_HALF_PI_F32 : 0.f;
  float sc[10];
#pragma unroll
  for (int l = 0; l < 10; ++l) sc[l] = sm_scale[l];
#pragma unroll
  for (int rho = 0; rho < 30; ++rho) {
    const float xb = __fmul_rn(x[rho % 3], sc[rho / 3]);
    E[rho >> 4][rho & 15] = sin_f32(__fadd_rn(xb, phase));
  }
  E[1][14] = h ? x[2] : x[0];
  E[1][15] = h ? 0.f : x[1];
}

__device__ __forceinline__ void encode_pos(const float (&x)[3], int h, f32x16 (&E)[2]) {
  const float phase = h ? AON_HALF_PI_F32 : 0.f;
#pragma unroll
  for (int rho = 0; rho < 30; ++rho) {
    const float xb = __fmul_rn(x[rho % 3], (float)(1 << (rho / 3)));
    E[rho >> 4][rho & 15] = sin_f32(__fadd_rn(xb, phase));
  }
  E[1][14] = h ? x[2] : x[0];
  E[1][15] = h ? 0.f : x[1];
}

__device__ __forceinline__ void encode_view(const float (&vd)[3], int h, f32x16& V) {
  const float phase = h ? AON_HALF_PI_F32 : 0.f;
#pragma unroll
  for (int rho = 0; rho < 12; ++rho) {
    const float xb = __fmul_rn(vd[rho % 3], (float)(1 << (rho / 3)));
    V[rho] = sin_f32(__fadd_rn(xb, phase));
  }
  V[12] = h ? vd[2] : vd[0];
  V[13] = h ? 0.f : vd[1];
  V[14] = 0.f; V[15] = 0.f;
}

__device__ __forceinline__ void load_pos_enc(const float* se, int h, f32x16 (&E)[2]) {  // caller-encoded (n*S,63)
#pragma unroll
  for (int rho = 0; rho < 30; ++rho) E[rho >> 4][rho & 15] = se[3 + rho + 30 * h];
  E[1][14] = h ? se[2] : se[0];
  E[1][15] = h ? 0.f : se[1];
}

__device__ __forceinline__ void load_view_enc(const float* ve, int h, f32x16& V) {  // caller-encoded (n,27)
#pragma unroll
  for (int rho = 0; rho < 12; ++rho) V[rho] = ve[3 + rho + 12 * h];
  V[12] = h ? ve[2] : ve[0];
  V[13] = h ? 0.f : ve[1];
  V[14] = 0.f; V[15] = 0.f;
}

__device__ __forceinline__ int posenc_col(int tile, int q, int cc, int h) {
  // position of packed input (tile, reg r=4q+cc, half h) in the reference's 63-wide encoding
  // [x(3) ; sin(2^l x) l-major (30) ; sin(2^l x + pi/2) (30)]   (helper.py:136-140)
  const int rho = 16 * tile + 4 * q + cc;
  if (rho < 30) return 3 + rho + 30 * h;
  if (rho == 30) return h ? 2 : 0;
  return h ? -1 : 1;
}

__device__ __forceinline__ int viewenc_col(int q, int cc, int h) {
  const int rho = 4 * q + cc;  // 27-wide: [v(3) ; sin (12) ; sin(+pi/2) (12)]
  if (rho < 12) return 3 + rho + 12 * h;
  if (rho == 12) return h ? 2 : 0;
  if (rho == 13) return h ? -1 : 1;
  return -1;
}

}  // namespace aon
