// Backward pass of the articulated network for gfx950 (SURVEY 8(a) R14 for R10/R11: what autograd does for
// model_autodecoder.py:395-477 training_step).  Gradients reach the MLP parameters AND the three latents
// (model_autodecoder.py:172-178), through the view branch, the trunk, the positional encoding of the deformed point
// and the deformation MLP.
//
// Same scheme as aon_train.hip: one register-resident data-gradient chain kernel (transposed weight stream as MFMA A
// operands, gradient tiles as B operands), pre-activation gradient planes, split-N weight-gradient GEMMs.  Specific:
//   * d(pos-enc) = W0[:, :63]^T dZ0 + W5[:, 256:319]^T dZ5 is accumulated in two 32x32 tiles in the encoding's permuted
//     register order, then pulled back through sin(2^l x' + phase): d/dx' = 2^l sin(2^l x' + phase + pi/2);
//   * every latent is broadcast to all samples, so  dW[:, latent cols] = db (x) latent  and  d latent = W[:, cols]^T db:
//     both come from the bias gradients, no per-sample work.
#include "aon_art_common.h"
#include "aon_fold.h"
#include "aon_wgrad.h"

namespace aon {

constexpr int kTinyChunkBytes = 2 * 4096;

struct ArtBwdNet {
  static constexpr int kSlotBytes = kPairSlotBytes;  // a slot holds a pair of chunks
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = kABwNumChunks;
  static constexpr int chunk_bytes(int c) {
    return (c < kABwV0 || c >= kABwD3) ? kSmallChunkBytes
         : ((c >= kABwL5E && c < kABwL5) || (c >= kABwL0E && c < kABwD3)) ? kTinyChunkBytes : kBigChunkBytes;
  }
};
__host__ __device__ constexpr int64_t abw_offset(int c) {
  int64_t off = 0;
  for (int i = 0; i < c; ++i) off += ArtBwdNet::chunk_bytes(i);
  return off;
}
constexpr int64_t kABwStreamBytes = abw_offset(kABwNumChunks);

struct ArtBwdFoldNet {   // folded form (aon_art_common.h): the literal stream without its eight bottleneck chunks
  static constexpr int kSlotBytes = kPairSlotBytes;
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = kABwFNumChunks;
  static constexpr int chunk_bytes(int c) { return ArtBwdNet::chunk_bytes(c < kABwBott ? c : c + (kABwL7 - kABwBott)); }
};
constexpr int64_t kABwFStreamBytes = kABwStreamBytes - (int64_t)(kABwL7 - kABwBott) * kBigChunkBytes;
constexpr int64_t kABwFOffWf = kABwFStreamBytes;   // W' (128 x 256 floats) for the pack kernel, inside the literal-size buffer
static_assert(kABwFStreamBytes + 128 * 256 * 4 <= kABwStreamBytes, "fold temporary fits behind the folded stream");

struct ArtParams {
  const float* p[kNumArtParams];
};

// the chunks of a stream as runs of equal size (compile time): first chunk, offset in floats and log2(floats per chunk) of each run
template <class N>
struct ChunkRuns {
  int n = 0;
  int first[8] = {};
  int shift[8] = {};
  int64_t off[8] = {};
  constexpr ChunkRuns() {
    int64_t o = 0;
    int prev = -1;
    for (int c = 0; c < N::kNumChunks; ++c) {
      const int b = N::chunk_bytes(c);
      if (b != prev) {
        first[n] = c; off[n] = o;
        int lg = 0;
        while ((4 << lg) < b) ++lg;
        shift[n] = lg;
        ++n;
        prev = b;
      }
      o += b / 4;
    }
  }
};

template <bool FOLD>
__device__ __forceinline__ void pack_art_bwd_element(const ArtParams& a, float* __restrict__ packed, int L, int Lv, const int64_t idx0, const int64_t idx) {
  using N = std::conditional_t<FOLD, ArtBwdFoldNet, ArtBwdNet>;
  const int P = 3 + 6 * L, V = 3 + 6 * Lv;   // (row strides of the three concatenating layers; the view-encoding columns are never read)
  if (idx >= (FOLD ? kABwFStreamBytes : kABwStreamBytes) / 4) return;
  // locate the chunk.  The stream is a handful of RUNS of equal-sized chunks (16 / 32 / 8 / 32 / 8 / 16 KiB): find the run of the BLOCK's
  // first index `idx0` (chunks are multiples of 4 KiB, a block is 1 KiB of the stream: one chunk per block, scalar instructions), then the
  // chunk inside it by a shift.  Rounds 2-5 scanned the ~100 chunks one by one on the thread's own index -- a few thousand vector
  // instructions per element, which made this the longest kernel of a training step's prologue.
  constexpr ChunkRuns<N> runs{};
  int c0 = 0, sh = runs.shift[0];
  int64_t off0 = 0;
#pragma unroll
  for (int t = 1; t < runs.n; ++t)
    if (idx0 >= runs.off[t]) { c0 = runs.first[t]; off0 = runs.off[t]; sh = runs.shift[t]; }
  int c = c0 + (int)((idx0 - off0) >> sh);
  const int64_t base = off0 + ((int64_t)(c - c0) << sh);
  const int r = (int)(idx - base);
  const int nt = N::chunk_bytes(c) / 4096;
  if constexpr (FOLD) { if (c >= kABwBott) c += kABwL7 - kABwBott; }   // the chunks behind views_linear.0 take the literal branches below
  const int cc = r & 3, lane = (r >> 2) & 63, rest = r >> 8;
  const int tp = rest % nt, q = rest / nt;
  const int h = lane >> 5, i = lane & 31;
  const int jo = 8 * q + 4 * h + cc;   // forward-output feature inside the chunk's 32-wide j tile
  const int f = 32 * tp + i;            // forward-input feature (row of W^T)
  const float* W; int ld, j, col = f;
  auto enc_col = [&]() {  // accumulator row i of tile tp -> encoding register rho of half h'
    const int rr = (i & 3) + 4 * (i >> 3), hh = (i >> 2) & 1;
    const int c63 = posenc_col(tp, rr >> 2, rr & 3, hh);
    return c63 < 0 ? -1 : pos_col_in(c63, L);   // a level the network lacks: zero weight, zero gradient into its slot
  };
  if (c < kABwV0) { const int l = 3 - c / 4; W = a.p[26 + 2 * l]; ld = 128; j = 32 * (c % 4) + jo; }
  else if (c < kABwBott) {
    W = a.p[26]; ld = 256 + V + 128; j = 32 * (c - kABwV0) + jo;
    if constexpr (FOLD) { W = packed + kABwFOffWf / 4; ld = 256; }   // W' (launch_fold_gemms on the same stream, in front of this kernel)
  }
  else if (c < kABwL7) { W = a.p[34]; ld = 256; j = 32 * (c - kABwBott) + jo; }
  else if (c < kABwL5E) { const int l = 7 - (c - kABwL7) / 8; W = a.p[10 + 2 * l]; ld = 256; j = 32 * ((c - kABwL7) % 8) + jo; }
  else if (c < kABwL5) { W = a.p[20]; ld = 256 + P + 128; j = 32 * (c - kABwL5E) + jo; col = enc_col(); if (col >= 0) col += 256; }
  else if (c < kABwL0E) {
    const int l = 5 - (c - kABwL5) / 8;  // 5,4,3,2,1
    W = a.p[10 + 2 * l]; ld = l == 5 ? 256 + P + 128 : 256; j = 32 * ((c - kABwL5) % 8) + jo;
  }
  else if (c < kABwD3) { W = a.p[10]; ld = P + 128; j = 32 * (c - kABwL0E) + jo; col = enc_col(); }
  else { const int l = 3 - (c - kABwD3) / 4; W = a.p[2 * l]; ld = 128; j = 32 * ((c - kABwD3) % 4) + jo; }
  packed[idx] = col >= 0 ? W[(int64_t)j * ld + col] : 0.f;
}
template <bool FOLD>
__global__ void __launch_bounds__(256) pack_art_bwd_kernel(ArtParams a, float* __restrict__ packed, int L, int Lv) {
  static_assert(kSmallChunkBytes % 4096 == 0 && kTinyChunkBytes % 4096 == 0 && kBigChunkBytes % 4096 == 0, "a 256-thread block never straddles two chunks");
  pack_art_bwd_element<FOLD>(a, packed, L, Lv, (int64_t)blockIdx.x * 256, (int64_t)blockIdx.x * 256 + threadIdx.x);
}
// the transposed streams of TWO networks in one launch (round 6; blockIdx.y: network), element by element the kernel above
struct ArtParams2 {
  ArtParams net[2];
  float* packed[2];
};
template <bool FOLD>
__global__ void __launch_bounds__(256) pack_art_bwd2_kernel(ArtParams2 a, int L, int Lv) {
  pack_art_bwd_element<FOLD>(a.net[blockIdx.y], a.packed[blockIdx.y], L, Lv, (int64_t)blockIdx.x * 256, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// One SEGMENT of a chain launch: the passes of one level (its transposed stream, small block, decision bits, planes).  Round 4: the
// backward chains of the two levels of a training step are independent, so they run as ONE persistent launch of two segments --
// 8,256 passes = 33 rounds of 256 workgroups at 4096 x (65 + 193) samples where two launches cost 9 + 25.
struct ArtBwdSeg {
  const char* packed_bwd;
  const float* small;     // forward per-call small block (head weights)
  const float* d_raw;     // (Np,4)
  const u32x4* masks;     // kAMaskLayers x (Np*2)
  const float* planes;    // forward planes (only the deformed position rows 3..5 are read)
  float* dplanes;         // gradient planes, art row map
  float* dxp;             // (Np,4): d x' per sample (for deformation_layer's weight gradient)
  int64_t Np;
  int npass;
};
struct ArtBwdArgs {
  ArtBwdSeg seg[2];
  int npass_total;        // seg[1].npass == 0: a one-segment launch
};

// timing experiments only (WRONG results): the chain without the plane stores of its 128-wide layers (view branch, deformation MLP:
// tiny chunks, a barrier every 128 MFMAs) or of its 256-wide layers -- which stores cost what (profiles/r05_chain_store_attribution.txt)
#if defined(AON_EXP_NOSTORE4)
#define AON_EXP_STORE_OF(NT) ((NT) != 4)
#elif defined(AON_EXP_NOSTORE8)
#define AON_EXP_STORE_OF(NT) ((NT) != 8)
#else
#define AON_EXP_STORE_OF(NT) true
#endif

template <int NT>
__device__ __forceinline__ void zero_tiles_a(f32x16 (&x)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = 0.f;
}

// FOLD: transposed stream of the folded form: d H7 = W'^T dZ_V0 + W_sigma^T d_sigma in one layer, no bottleneck gradient.
template <bool FOLD>
__global__ void __launch_bounds__(256) art_bwd_chain_kernel(ArtBwdArgs args) {
  constexpr int kL7 = FOLD ? kABwFL7 : kABwL7, kL5E = FOLD ? kABwFL5E : kABwL5E, kL5 = FOLD ? kABwFL5 : kABwL5, kL0E = FOLD ? kABwFL0E : kABwL0E,
                kD3 = FOLD ? kABwFD3 : kABwD3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem + kRingBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int npass0 = args.seg[0].npass;
  int cur = (int)blockIdx.x >= npass0 ? 1 : 0;
  auto load_small = [&](const float* small) {
    const f32x4* src = reinterpret_cast<const f32x4*>(small);
    f32x4* dst = reinterpret_cast<f32x4*>(sm);
    for (int i = tid; i < kASmallFloats / 4; i += 256) dst[i] = src[i];
  };
  load_small(args.seg[cur].small);
  Pipe p;
  pipe_init<std::conditional_t<FOLD, ArtBwdFoldNet, ArtBwdNet>>(p, args.seg[cur].packed_bwd, smem, wave, lane);  // also publishes the small block just written to LDS
  using N = std::conditional_t<FOLD, ArtBwdFoldNet, ArtBwdNet>;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);

  for (int gpass = blockIdx.x; gpass < args.npass_total; gpass += gridDim.x) {
    const int si = gpass >= npass0 ? 1 : 0;
    if (si != cur) {   // (workgroup-uniform, at most once per launch) the other level's head weights replace the resident block
      __syncthreads();
      load_small(args.seg[si].small);
      __syncthreads();
      cur = si;
    }
    const ArtBwdSeg& sg = args.seg[si];
    const int pass = gpass - (si ? npass0 : 0);
    {   // transposed stream of this pass, and of this workgroup's next one (its first chunk pair is fetched by this pass's last two chunks)
      const int nxt = gpass + (int)gridDim.x;
      p.stream = sg.packed_bwd;
      p.next_stream = args.seg[(nxt >= npass0 && nxt < args.npass_total) ? 1 : si].packed_bwd;
    }
    // Lane coordinates are RE-DERIVED once per pass (v_mbcnt + the wave index in an SGPR) instead of kept: with all 256 + 256
    // registers taken by the two activation sets, every loop-invariant per-lane value -- the thread id itself, lane ^ 32, the
    // 64-bit row offset built from it -- was hoisted out of the pass loop and spilled (24 B/lane of scratch).
    int lane_p;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
    const int tid_p = (wave_s << 6) | lane_p;
    const int m = lane_p & 31, h = lane_p >> 5;
    const int64_t col = (int64_t)pass * 128 + (tid_p >> 6) * 32 + m;
    const int64_t step = (int64_t)pass * 4 + wave_s;
    const PlaneIO io = make_plane_io(sg.dplanes, kAPlRows, step, m, h);
    // decision bits of a layer: fetched one layer ahead of their use, offset opaque so the load stays where it is written
    // (round 1 fetched all sixteen words up front: 64 registers held through the pass)
    const unsigned moff = mask_lane_off(pass, tid_p);
    auto load_mask = [&](int slot) { return *mask_ptr(sg.masks, sg.Np, slot, moff); };
    int hl = h;  // half-wave index for the LDS reads of head weights: opaque per pass (see mlp_bwd_chain_kernel)
    asm volatile("" : "+v"(hl));
    const float4 dr = reinterpret_cast<const float4*>(sg.d_raw)[col];
    u32x4 mk = load_mask(15), mk_next;

    // ---- view branch, backwards (model_autodecoder.py:231-236) ----
    f32x16 Z0[4], Z1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int fo = 32 * t + 8 * gq + 4 * hl;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + kA_WRGB + 0 * kCondWidth + fo);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + kA_WRGB + 1 * kCondWidth + fo);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(sm + kA_WRGB + 2 * kCondWidth + fo);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          Z1[t][4 * gq + cc] = __builtin_fmaf(w2[cc], dr.z, __builtin_fmaf(w1[cc], dr.y, w0[cc] * dr.x));
      }
    }
    // one layer of the chain: IN holds dH of the layer in mask slot SLOT -> dZ (masked, stored by the consuming chunks),
    // OUT = W^T . dZ
#define AON_ABWD_LAYER(NT_IN, NT_OUT, IN, OUT, CB, ROW, NEXT_SLOT)                                                   \
    if (NEXT_SLOT >= 0) mk_next = load_mask(NEXT_SLOT);                                                               \
    apply_mask_tile(IN[0], mk, 0);                                                                                   \
    dense_layer<N, CB, NT_IN, NT_OUT, BwdSideOf<NT_IN, true, AON_EXP_STORE_OF(NT_IN)>, true>(p, IN, OUT, BwdSideOf<NT_IN, true, AON_EXP_STORE_OF(NT_IN)>{IN, ROW, io, mk, NEXT_SLOT >= 0 ? &mk_next : nullptr});   /* OUT starts from zero */ \
    mk = mk_next;
    AON_ABWD_LAYER(4, 4, Z1, Z0, kABwV3 + 0, aplane_v(3), 14)
    AON_ABWD_LAYER(4, 4, Z0, Z1, kABwV3 + 4, aplane_v(2), 13)
    AON_ABWD_LAYER(4, 4, Z1, Z0, kABwV3 + 8, aplane_v(1), 12)
    f32x16 X[8], Y[8];
    if constexpr (!FOLD) {
      AON_ABWD_LAYER(4, 8, Z0, X, kABwV0, aplane_v(0), 11)   // X = d bottleneck (no activation)
    }
    // ---- trunk ----
    // the sample index is re-derived where it is needed (one v_mbcnt pair) instead of held in a register pair across the view
    // branch / the trunk: that pair was the kernel's last scratch spill
    auto sample_now = [&]() {
      int l;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
      return (int64_t)pass * 128 + wave_s * 32 + (l & 31);
    };
    const float dsig = dr.w;   // (round 3 re-read it here to save a register across the view branch: a load whose wait also waits for the view branch's stores)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sm + kA_WSIG + 32 * t + 8 * gq + 4 * hl);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) Y[t][4 * gq + cc] = w[cc] * dsig;
      }
    }
    if constexpr (FOLD) {
      // Y = dH7 = W'^T dZ_V0 on top of the density head's term (W' = W_v0[:, :256] W_b: model_autodecoder.py:223-230 as one layer)
      mk_next = load_mask(11);
      apply_mask_tile(Z0[0], mk, 0);
      dense_layer<N, kABwV0, 4, 8, BwdSideOf<4, true>, false>(p, Z0, Y, BwdSideOf<4, true>{Z0, aplane_v(0), io, mk, &mk_next});
      mk = mk_next;
    } else {
      dense_layer<N, kABwBott, 8, 8>(p, X, Y, BwdSideOf<8, false>{X, kAPlBot, io, mk});   // Y = dH7
    }
    AON_ABWD_LAYER(8, 8, Y, X, kL7 + 0, aplane_h(7), 10)
    AON_ABWD_LAYER(8, 8, X, Y, kL7 + 8, aplane_h(6), 9)
    // Y = dH5 -> dZ5, consumed twice: by the skip-connection chunks (d enc += W5[:, 256:319]^T dZ5), which mask and store it,
    // then by layer 5's own transposed chunks
    f32x16 dE[2];
    mk_next = load_mask(8);
    apply_mask_tile(Y[0], mk, 0);
    // (the eight encoding chunks are tiny -- 8 MFMA groups each: they only mask; dZ5's stores ride on layer 5's own chunks below)
    dense_layer<N, kL5E, 8, 2, BwdSideOf<8, true, false>, true>(p, Y, dE, BwdSideOf<8, true, false>{Y, aplane_h(5), io, mk, &mk_next});   // dE starts from zero
    mk = mk_next;
    // The partial d enc (32 accumulator registers) would have to stay live across layers 5..1 on top of the two 128-register
    // activation sets; it is parked in the (otherwise unused) pos-enc rows of the gradient planes instead -- 128 B per lane out
    // and back per pass, against 13.8 KB of plane traffic -- rather than left to the register allocator's scratch spills.
#ifndef AON_EXP_NOPARK      // timing experiment only (WRONG results): d enc not parked in the planes
    store_plane(dE, io, kAPlE);
#endif
#ifdef AON_EXP_NOSTORE_L5   // timing experiment only (WRONG results): dZ5 never stored
    dense_layer<N, kL5 + 0, 8, 8, NoSideOf, true>(p, Y, X);
#else
    dense_layer<N, kL5 + 0, 8, 8, StoreSideOf<8>, true>(p, Y, X, StoreSideOf<8>{Y, aplane_h(5), io});   // X = dH4 (from zero); stores dZ5
#endif
    AON_ABWD_LAYER(8, 8, X, Y, kL5 + 8, aplane_h(4), 7)
    AON_ABWD_LAYER(8, 8, Y, X, kL5 + 16, aplane_h(3), 6)
    AON_ABWD_LAYER(8, 8, X, Y, kL5 + 24, aplane_h(2), 5)
    AON_ABWD_LAYER(8, 8, Y, X, kL5 + 32, aplane_h(1), 4)
    // X = dH0 -> dZ0, consumed by the encoding chunks: d enc += W0[:, :63]^T dZ0
    mk_next = load_mask(3);
    apply_mask_tile(X[0], mk, 0);
    load_plane(dE, io, kAPlE);
    // the deformed position x' for the encoding's backward is fetched HERE, in front of the encoding chunks, not behind them: its wait
    // then covers only what is older than this point, not the burst of dZ0 stores below
    float xd[3];  // deformed position x' (forward stored it in rows 3..5 of the position block)
    const PlaneIO fio = make_plane_io(sg.planes, kAPlRows, step, m, h);
#pragma unroll
    for (int a = 0; a < 3; ++a)
      xd[a] = *row_ptr(fio, kAPlPos + 3 + a);
#if defined(AON_EXP_NOSTORE_L0E)   // timing experiment only (dZ0 never stored: WRONG layer-0 weight gradients)
    dense_layer<N, kL0E, 8, 2>(p, X, dE, BwdSideOf<8, true, false>{X, aplane_h(0), io, mk});
#elif defined(AON_EXP_L0E_SIDE)    // round-3 form: dZ0 stored by the tiny encoding chunks themselves
    dense_layer<N, kL0E, 8, 2>(p, X, dE, BwdSideOf<8, true>{X, aplane_h(0), io, mk});
#else
    // the encoding chunks (tiny, see BwdSideOf) only mask; dZ0 goes out in one burst behind them, in front of the ~600 VALU
    // instructions of the encoding's backward, which give its acknowledgements time before the next weight DMA is waited for
    dense_layer<N, kL0E, 8, 2>(p, X, dE, BwdSideOf<8, true, false>{X, aplane_h(0), io, mk, &mk_next});
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) store_quad<false>(io, aplane_h(0) + 32 * t8, gq, X[t8]);
#endif
    mk = mk_next;

    // ---- positional encoding, backwards (helper.py:136-140 on the deformed point) ----
    const float phase = h ? AON_HALF_PI_F32 : 0.f;
    float dx[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int rho = 0; rho < 30; ++rho) {
      const float scale = sm[kA_ESC + rho / 3];   // 2^(min_deg_point + l), 0 for a level the network lacks (aon_art_prepare)
      const float arg = __fadd_rn(__fmul_rn(xd[rho % 3], scale), phase);
      const float c = cos_f32(arg);  // d/d(arg) sin(arg), at the forward's own (rounded) argument
      dx[rho % 3] = __builtin_fmaf(scale * c, dE[rho >> 4][rho & 15], dx[rho % 3]);
    }
    if (h) dx[2] += dE[1][14]; else { dx[0] += dE[1][14]; dx[1] += dE[1][15]; }
#pragma unroll
    for (int a = 0; a < 3; ++a)   // + the other half-wave's partial (lane ^ 32, index from this pass's lane id)
      dx[a] = dx[a] + __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane_p ^ 32) << 2, __builtin_bit_cast(int, dx[a])));
    if (h == 0) {
      float4 o; o.x = dx[0]; o.y = dx[1]; o.z = dx[2]; o.w = 0.f;
      reinterpret_cast<float4*>(sg.dxp)[sample_now()] = o;
    }

    // ---- deformation MLP, backwards (x' = deformation_layer(h3) + pos, :200-205) ----
    f32x16 (&H1)[4] = Z1;
    f32x16 (&H0)[4] = Z0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int fo = 32 * t + 8 * gq + 4 * hl;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + kA_WDL + 0 * 128 + fo);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + kA_WDL + 1 * 128 + fo);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(sm + kA_WDL + 2 * 128 + fo);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          H1[t][4 * gq + cc] = __builtin_fmaf(w2[cc], dx[2], __builtin_fmaf(w1[cc], dx[1], w0[cc] * dx[0]));
      }
    }
    AON_ABWD_LAYER(4, 4, H1, H0, kD3 + 0, aplane_d(3), 2)
    AON_ABWD_LAYER(4, 4, H0, H1, kD3 + 4, aplane_d(2), 1)
    AON_ABWD_LAYER(4, 4, H1, H0, kD3 + 8, aplane_d(1), 0)
#undef AON_ABWD_LAYER
    // dZ of deformation layer 0: its input is (pos, latents) -- no data gradient continues, no consuming chunk: 64 values here
    apply_mask_bits(H0, mk);
#ifndef AON_EXP_NOFINAL     // timing experiment only (WRONG results): the pass's last burst (dZ of deformation layer 0) not stored
    store_plane(H0, io, aplane_d(0));
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Everything that follows from the bias gradients, ONE launch (1,024-thread blocks):
//   blocks 0..2   d latent[k] = sum over (W, db) pairs, sum_f W[f][col_off + k] * db[f]  (block = latent).  8 row groups x 128
//                 columns: group g takes rows f = g, g+8, ... (independent loads, coalesced over k), the eight group sums are
//                 added in group order through LDS (deterministic);
//   blocks 3..    the latent columns of the weights, dW[f][col_off + k] = db[f] * latent[k]  (every latent is broadcast to all
//                 samples by the reference, model_autodecoder.py:186-194).
struct LatentJob {
  const float* W[3]; const float* db[3]; int ld[3]; int col_off[3]; int M[3];
  int npairs; int L;
  float* out;
  const float* add;   // null, or the other level's result: out = add + this level's (both MLPs see the same latents; may alias out)
};
struct OuterJob {
  const float* db; const float* latent; float* out;
  int M, L, ld, col_off, blk_begin;
};
struct ArtFinishArgs {
  LatentJob lat[3];
  OuterJob outer[5];
};
// (blocks 3.. of a level's finishing launch; bx: the block's index in that launch)
__device__ __forceinline__ void art_finish_outer(const ArtFinishArgs& a, const int bx) {
  int j = 0;
#pragma unroll 1
  for (int t = 1; t < 5; ++t)
    if (bx >= a.outer[t].blk_begin) j = t;
  const OuterJob& O = a.outer[j];
  const int idx = (bx - O.blk_begin) * 1024 + threadIdx.x;
  if (idx < O.M * O.L) {
    const int f = idx / O.L, k = idx % O.L;
    O.out[(int64_t)f * O.ld + O.col_off + k] = O.db[f] * O.latent[k];
  }
}
// (blocks 0..2: this level's d latent[k], valid in the threads of row group 0 with k < L)
__device__ __forceinline__ float art_finish_latent(const LatentJob& j, float (*red)[128]) {
  const int k = threadIdx.x & 127, g = threadIdx.x >> 7;
  float s = 0.f;
  if (k < j.L) {
    for (int pi = 0; pi < j.npairs; ++pi) {
      const float* W = j.W[pi] + j.col_off[pi] + k;
      const float* db = j.db[pi];
      const int ld = j.ld[pi];
#pragma unroll 4
      for (int f = g; f < j.M[pi]; f += 8) s = __builtin_fmaf(W[(int64_t)f * ld], db[f], s);
    }
  }
  red[g][k] = s;
  __syncthreads();
  float t = 0.f;
  if (g == 0 && k < j.L) {
    t = red[0][k];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += red[q][k];
  }
  return t;
}
__global__ void __launch_bounds__(1024) art_finish_kernel(ArtFinishArgs a) {
  __shared__ float red[8][128];
  if (blockIdx.x >= 3) { art_finish_outer(a, (int)blockIdx.x); return; }
  const LatentJob& j = a.lat[blockIdx.x];
  const float t = art_finish_latent(j, red);
  const int k = threadIdx.x & 127, g = threadIdx.x >> 7;
  if (g == 0 && k < j.L) j.out[k] = j.add ? j.add[k] + t : t;
}
// The finishing kernels of TWO levels as one launch (round 6): blocks 0..2 take each latent through level 0 and then level 1 --
// out = (level 0's sum) + (level 1's sum), the value the two launches in a row leave there (level 0 stores its fp32 sum, level 1 adds
// its own to it) -- blocks [3, n0) are level 0's latent-column blocks, the rest level 1's.
// (the two levels' sums run in lockstep -- two independent chains of multiply-adds, each in its own order, their loads in flight together:
// the kernel is load latency from end to end)
__global__ void __launch_bounds__(1024) art_finish2_kernel(ArtFinishArgs a0, ArtFinishArgs a1, int n0) {
  __shared__ float red[2][8][128];
  const int bx = (int)blockIdx.x;
  if (bx >= n0) { art_finish_outer(a1, bx - n0 + 3); return; }
  if (bx >= 3) { art_finish_outer(a0, bx); return; }
  const LatentJob& j0 = a0.lat[bx];
  const LatentJob& j1 = a1.lat[bx];
  const int k = threadIdx.x & 127, g = threadIdx.x >> 7;
  float s0 = 0.f, s1 = 0.f;
  if (k < j0.L) {   // (launch_art_wgrad_post2 checked that the two levels' jobs have the same shape: L, pairs, M)
    for (int pi = 0; pi < j0.npairs; ++pi) {
      const float* W0 = j0.W[pi] + j0.col_off[pi] + k;
      const float* W1 = j1.W[pi] + j1.col_off[pi] + k;
      const float* db0 = j0.db[pi];
      const float* db1 = j1.db[pi];
      const int ld0 = j0.ld[pi], ld1 = j1.ld[pi];
#pragma unroll 4
      for (int f = g; f < j0.M[pi]; f += 8) {
        s0 = __builtin_fmaf(W0[(int64_t)f * ld0], db0[f], s0);
        s1 = __builtin_fmaf(W1[(int64_t)f * ld1], db1[f], s1);
      }
    }
  }
  red[0][g][k] = s0;
  red[1][g][k] = s1;
  __syncthreads();
  if (g == 0 && k < j0.L) {
    float t0 = red[0][0][k], t1 = red[1][0][k];
#pragma unroll
    for (int q = 1; q < 8; ++q) { t0 += red[0][q][k]; t1 += red[1][q][k]; }
    const float v0 = j0.add ? j0.add[k] + t0 : t0;
    j1.out[k] = v0 + t1;
  }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
int num_cus();

// The form packed is the process default at the time of the call (aon_set_bottleneck_fold), remembered for `packed` (stream_form).
FoldGemm art_fold_job_bwd(const float* const* params, float* packed, int view_levels) {
  return FoldGemm{params[26], 256 + 3 + 6 * view_levels + 128, 1, params[34], 256, 1, packed + kABwFOffWf / 4, 256, 128, 256, 256, nullptr, nullptr};
}

// both networks of a two-level model, one launch (the folded form's W' must be in place: aon_art_pack_step)
hipError_t launch_pack_art_bwd2(const float* const* const params[2], float* const packed[2], hipStream_t stream, int pos_levels, int view_levels, int form) {
  ArtParams2 a;   // (form: decided ONCE by the caller)
  for (int l = 0; l < 2; ++l) {
    for (int i = 0; i < kNumArtParams; ++i) a.net[l].p[i] = params[l][i];
    a.packed[l] = packed[l];
    set_stream_form(packed[l], form);
  }
  const int64_t n = (form == kFormFolded ? kABwFStreamBytes : kABwStreamBytes) / 4;
  const dim3 grid((unsigned)((n + 255) / 256), 2);
  if (form == kFormFolded) pack_art_bwd2_kernel<true><<<grid, dim3(256), 0, stream>>>(a, pos_levels, view_levels);
  else pack_art_bwd2_kernel<false><<<grid, dim3(256), 0, stream>>>(a, pos_levels, view_levels);
  return hipGetLastError();
}

hipError_t launch_pack_art_bwd(const float* const* params, float* packed, hipStream_t stream, int pos_levels, int view_levels, bool fold_done) {
  ArtParams a;
  for (int i = 0; i < kNumArtParams; ++i) a.p[i] = params[i];
  const int form = fold_default();
  set_stream_form(packed, form);
  if (form == kFormFolded) {
    if (!fold_done) {
      const FoldGemm job = art_fold_job_bwd(params, packed, view_levels);
      if (hipError_t e = launch_fold_gemms(&job, 1, stream); e != hipSuccess) return e;
    }
    const int64_t n = kABwFStreamBytes / 4;
    pack_art_bwd_kernel<true><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed, pos_levels, view_levels);
  } else {
    const int64_t n = kABwStreamBytes / 4;
    pack_art_bwd_kernel<false><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed, pos_levels, view_levels);
  }
  return hipGetLastError();
}

int64_t art_bwd_stream_bytes() { return kABwStreamBytes; }

template <bool FOLD>
static hipError_t launch_art_chain_f(const ArtBwdArgs& a, int grid, hipStream_t stream) {
  static DeviceOnce lds_once;
  if (hipError_t e = set_max_lds(&art_bwd_chain_kernel<FOLD>, kALdsBytes, lds_once); e != hipSuccess) return e;
  art_bwd_chain_kernel<FOLD><<<dim3(grid), dim3(256), kALdsBytes, stream>>>(a);
  return hipGetLastError();
}

hipError_t launch_art_bwd_chain2(const ChainSeg* segs, int nsegs, hipStream_t stream) {
  if (nsegs < 1 || nsegs > 2) return hipErrorInvalidValue;
  const int form = stream_form(segs[0].packed_bwd);
  if (form == kFormUnknown) return hipErrorInvalidValue;   // never packed / declared (a copy): refuse instead of guessing
  for (int i = 0; i < nsegs; ++i)   // streams and per-call blocks of one launch: one form
    if (stream_form(segs[i].packed_bwd) != form || stream_form(segs[i].small) != form) return hipErrorInvalidValue;
  ArtBwdArgs a{};
  for (int i = 0; i < nsegs; ++i) {
    const ChainSeg& c = segs[i];
    a.seg[i] = ArtBwdSeg{c.packed_bwd, c.small, c.d_raw, static_cast<const u32x4*>(c.masks), c.planes, c.dplanes, c.dxp, c.Np, (int)(c.Np / 128)};
    a.npass_total += a.seg[i].npass;
  }
  if (nsegs == 1) { a.seg[1] = a.seg[0]; a.seg[1].npass = 0; }
  else if (a.seg[0].npass == 0) { a.seg[0] = a.seg[1]; a.seg[1].npass = 0; }
  const int cus = num_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  const int grid = a.npass_total < cus ? a.npass_total : cus;
  if (grid <= 0) return hipSuccess;
  return form == kFormFolded ? launch_art_chain_f<true>(a, grid, stream) : launch_art_chain_f<false>(a, grid, stream);
}

hipError_t launch_art_bwd_chain(const char* packed_bwd, const float* small, const float* d_raw, const void* masks, const float* planes,
                                float* dplanes, float* dxp, int64_t Np, hipStream_t stream) {
  const ChainSeg one{packed_bwd, small, d_raw, masks, planes, dplanes, dxp, Np};
  return launch_art_bwd_chain2(&one, 1, stream);
}

hipError_t run_wgrad_plan(const WgLayerDesc* layers, int nlayers, const HeadDesc* heads, int nheads, const HeadOut* outs, const int* out_head, int nouts,
                          const float* planes, const float* dplanes, int rows_total, int64_t Np, float* ws, hipStream_t stream, const WgAux* aux,
                          int phase, int n_early, const WgPost* post, hipStream_t* post_stream, ReduceArgs* defer_reduce, int* defer_blocks);   // aon_train.hip
hipError_t launch_wgrad_reduce2(const ReduceArgs& a0, int n0, const ReduceArgs& a1, int n1, hipStream_t stream);   // aon_train.hip

// the weight-gradient jobs of one articulated level.  Lp / Lv: frequency levels of the network (10 / 4 by default).  With other degrees
// the three encoding-fed column blocks come out in the kernels' 63 / 27-slot layout into `enc_tmp` (256 x 64 | 256 x 64 | 128 x 32
// floats) and lose the empty slots afterwards (art_remap_enc_kernel); the row strides of the concatenating layers follow P and V.
// fold_tmp != null: the planes are the folded form's (no bottleneck rows): the bottleneck and views_linear.0's bottleneck columns are ONE job
// dW' = dZ_V0 . H7^T (128 x 256) into fold_tmp, db' = db_v0 straight into views_linear.0.bias; launch_unfold_view makes the reference's gradients.
int art_wgrad_layers(float* const* grads, WgLayerDesc* L, int Lp, int Lv, float* enc_tmp, float* fold_tmp) {
  const bool dflt = Lp == 10 && Lv == 4;
  const int P = 3 + 6 * Lp, V = 3 + 6 * Lv;
  int n = 0;
  // deformation MLP (model_autodecoder.py:196-203): layers 1..3 here; layer 0's three position columns go with the heads below
  // (its input is cat[pos(3), shape(128), articulation(32)]: a 16-byte record per sample, not a plane operand)
  for (int l = 1; l < 4; ++l) L[n++] = WgLayerDesc{kWg128x128, aplane_d(l), aplane_d(l - 1), grads[2 * l], 128, 0, 128, grads[2 * l + 1]};
  // trunk (:210-217): layer 0 input cat[enc(P), shape(128)], layer 5 input cat[h(256), enc(P), shape(128)]
  if (dflt) L[n++] = WgLayerDesc{kWg256x64, aplane_h(0), kAPlE, grads[10], P + 128, 0, kPosEnc, grads[11]};
  else L[n++] = WgLayerDesc{kWg256x64, aplane_h(0), kAPlE, enc_tmp, 64, 0, kPosEnc, grads[11]};
  for (int l = 1; l < 8; ++l) {
    const int ld = l == 5 ? 256 + P + 128 : 256;
    L[n++] = WgLayerDesc{kWg256x256, aplane_h(l), aplane_h(l - 1), grads[10 + 2 * l], ld, 0, 256, grads[11 + 2 * l]};
    if (l == 5) {
      if (dflt) L[n++] = WgLayerDesc{kWg256x64, aplane_h(5), kAPlE, grads[20], ld, 256, kPosEnc, nullptr};
      else L[n++] = WgLayerDesc{kWg256x64, aplane_h(5), kAPlE, enc_tmp + 256 * 64, 64, 0, kPosEnc, nullptr};
    }
  }
  if (fold_tmp) {
    L[n++] = WgLayerDesc{kWg128x256, aplane_v(0), aplane_h(7), fold_tmp, 256, 0, 256, grads[27]};
  } else {
    L[n++] = WgLayerDesc{kWg256x256, kAPlBot, aplane_h(7), grads[34], 256, 0, 256, grads[35]};
    // view branch (:227-234): layer 0 input cat[bottleneck(256), viewenc(V), appearance(128)]: two column blocks from the planes
    L[n++] = WgLayerDesc{kWg128x256, aplane_v(0), kAPlBot, grads[26], 256 + V + 128, 0, 256, grads[27]};
  }
  if (dflt) L[n++] = WgLayerDesc{kWg128x32, aplane_v(0), kAPlVE, grads[26], 256 + V + 128, 256, kViewEnc, nullptr};
  else L[n++] = WgLayerDesc{kWg128x32, aplane_v(0), kAPlVE, enc_tmp + 2 * 256 * 64, 32, 0, kViewEnc, nullptr};
  for (int l = 1; l < 4; ++l) L[n++] = WgLayerDesc{kWg128x128, aplane_v(l), aplane_v(l - 1), grads[26 + 2 * l], 128, 0, 128, grads[27 + 2 * l]};
  return n;
}
int art_wgrad_layers(float* const* grads, WgLayerDesc* L) { return art_wgrad_layers(grads, L, 10, 4, nullptr, nullptr); }
float* wgrad_fold_tmp(float* ws);   // aon_train.hip

// dst[r * ldd + col_off + c] = src[r * lds + slot(c)] for the c < 3 + 6 L columns of an encoding with L levels, taken out of the kernels'
// Lfull-level slot layout [x ; first block of 3 Lfull ; shifted block of 3 Lfull]
__global__ void art_remap_enc_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int col_off, int rows, int L, int Lfull) {
  const int cols = 3 + 6 * L;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, c = i % cols;
  const int sc = c < 3 + 3 * L ? c : c + 3 * (Lfull - L);
  dst[(int64_t)r * ldd + col_off + c] = src[(int64_t)r * lds + sc];
}

// grads: 40 parameter gradients (order of aon_pack_art_mlp, full shapes) + 3 latent gradients (shape 128, appearance 128,
// articulation 32); params / latents: the forward's inputs (needed for the latent-column products).
// Round 6: a level's second stage NOT launched by its own call but handed back, so that the two levels' grouped kernels run back to back
// and ONE second stage serves both (launch_art_wgrad_post2): reduce -> un-folding products -> finishing kernel were three launches in a
// row per level with the chip all but idle, level 0's in front of level 1's grouped kernel.  Opaque to the C ABI layer (kArtWgDeferredBytes).
struct ArtWgDeferred {
  ReduceArgs reduce;
  int reduce_blocks;
  int n_unfold;           // 3 (folded form) or 0
  FoldGemm unfold[3];
  ArtFinishArgs finish;
  int finish_blocks;
};
constexpr int kArtWgDeferredBytes = 4096;   // (aon_capi.hip keeps two of these on its stack)
static_assert(sizeof(ArtWgDeferred) <= kArtWgDeferredBytes && alignof(ArtWgDeferred) <= 16, "ArtWgDeferred outgrew its storage in aon_capi.hip");
int art_wgrad_deferred_bytes() { return (int)sizeof(ArtWgDeferred); }   // (aon_capi.hip checks its storage against this: the constant is repeated there)

hipError_t launch_art_wgrad(const float* planes, const float* dplanes, const float* d_raw, const float* dxp, int64_t Np,
                            const float* const* params, const float* shape, const float* app, const float* art,
                            float* const* grads, float* g_shape, float* g_app, float* g_art, float* ws, hipStream_t stream, const WgAux* aux,
                            int Lp, int Lv, const void* packed_bwd, int phase, bool accumulate_latents, const WgPost* post, ArtWgDeferred* defer) {
  // packed_bwd: the transposed stream the chain of these planes ran with -- its FORM says whether the planes carry bottleneck rows (null: literal)
  if (packed_bwd && stream_form(packed_bwd) == kFormUnknown) return hipErrorInvalidValue;   // (a copy nobody declared)
  const bool fold = packed_bwd && stream_form(packed_bwd) == kFormFolded;
  float* fold_tmp = fold ? wgrad_fold_tmp(ws) : nullptr;
  WgLayerDesc L[kWgMaxJobs];
  const int P = 3 + 6 * Lp, V = 3 + 6 * Lv;
  const bool dflt = Lp == 10 && Lv == 4;
  // (other degrees: the slot-layout blocks live in the last MiB of the weight-gradient workspace, which no plan reaches: 70 of 96 MiB
  // are used at 4096 x 193 samples, and run_wgrad_plan refuses plans beyond the workspace)
  float* enc_tmp = ws + (wgrad_workspace_bytes_impl() - (1 << 20)) / 4;
  const int n = art_wgrad_layers(grads, L, Lp, Lv, enc_tmp, fold_tmp);
  // heads: density (H7 x d_raw.w), rgb (V3 x d_raw.xyz), deformation_layer (D3 x dx'), their bias sums, and deformation layer 0:
  // dW[:, 0:3] = dZ_D0 x pos (the position is rows 0..2 of unit row kAPlPos / 4 of the forward planes), db = row sums of dZ_D0
  const int64_t unit_step = (int64_t)kAPlRows * 32;
  const HeadDesc H[6] = {{planes, aplane_h(7), 256, d_raw, 128}, {planes, aplane_v(3), 128, d_raw, 128}, {nullptr, 0, 1, d_raw, 128},
                         {planes, aplane_d(3), 128, dxp, 128},   {nullptr, 0, 1, dxp, 128},
                         {dplanes, aplane_d(0), 128, planes + (int64_t)(kAPlPos / 4) * 128, unit_step}};
  const HeadOut O[8] = {{0, 256, 3, 1, 256, 1, grads[36]}, {0, 128, 0, 3, 128, 1, grads[38]}, {0, 1, 3, 1, 1, 1, grads[37]}, {0, 1, 0, 3, 1, 1, grads[39]},
                        {0, 128, 0, 3, 128, 1, grads[8]},  {0, 1, 0, 3, 1, 1, grads[9]},
                        {0, 128, 0, 3, 1, 163, grads[0]},  {0, 128, 4, 1, 1, 1, grads[1]}};
  const int OH[8] = {0, 1, 2, 2, 3, 4, 5, 5};
  // head jobs 0..2 (density head on H7, rgb head on V3, the sums of d_raw) read forward planes and d_raw only: independent of the chain
  hipStream_t caller_stream = stream;
  if (defer && (phase == kWgEarly || post || !dflt)) return hipErrorInvalidValue;   // (other degrees: remap launches between the stages; not deferred)
  if (hipError_t e = run_wgrad_plan(L, n, H, 6, O, OH, 8, planes, dplanes, kAPlRows, Np, ws, stream, aux, phase, 3, phase == kWgEarly ? nullptr : post, &stream,
                                    defer ? &defer->reduce : nullptr, defer ? &defer->reduce_blocks : nullptr); e != hipSuccess)
    return e;
  if (phase == kWgEarly) return hipSuccess;
  // (from here on `stream` is the stream of the second stage: the caller's, or the side stream of `post`)
  (void)caller_stream;
  if (!dflt) {
    auto remap = [&](const float* src, int lds, float* dst, int ldd, int col_off, int rows, int Lx, int Lfull) {
      const int tot = rows * (3 + 6 * Lx);
      art_remap_enc_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream>>>(src, lds, dst, ldd, col_off, rows, Lx, Lfull);
    };
    remap(enc_tmp, 64, grads[10], P + 128, 0, 256, Lp, 10);
    remap(enc_tmp + 256 * 64, 64, grads[20], 256 + P + 128, 256, 256, Lp, 10);
    remap(enc_tmp + 2 * 256 * 64, 32, grads[26], 256 + V + 128, 256, 128, Lv, 4);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  }
  if (defer) {
    defer->n_unfold = fold ? 3 : 0;
    if (fold) unfold_view_jobs(fold_tmp, grads[27], params[26], 256 + V + 128, params[34], params[35], grads[26], 256 + V + 128, grads[34], grads[35], defer->unfold);
  } else if (fold)   // bottleneck_layer's and views_linear.0[:, :256]'s gradients from (dW', db' = views_linear.0.bias's gradient)
    if (hipError_t e = launch_unfold_view(fold_tmp, grads[27], params[26], 256 + V + 128, params[34], params[35], grads[26], 256 + V + 128, grads[34], grads[35], stream);
        e != hipSuccess) return e;
  // latent columns of the weights and the latent gradients, both from the bias gradients
  ArtFinishArgs F{};
  LatentJob& ls = F.lat[0];   // shape: deformation layer 0, trunk layers 0 and 5
  ls.W[0] = params[0]; ls.db[0] = grads[1]; ls.ld[0] = 163; ls.col_off[0] = 3; ls.M[0] = 128;
  ls.W[1] = params[10]; ls.db[1] = grads[11]; ls.ld[1] = P + 128; ls.col_off[1] = P; ls.M[1] = 256;
  ls.W[2] = params[20]; ls.db[2] = grads[21]; ls.ld[2] = 256 + P + 128; ls.col_off[2] = 256 + P; ls.M[2] = 256;
  ls.npairs = 3; ls.L = 128; ls.out = g_shape; ls.add = accumulate_latents ? g_shape : nullptr;
  LatentJob& la = F.lat[1];   // appearance: view layer 0
  la.W[0] = params[26]; la.db[0] = grads[27]; la.ld[0] = 256 + V + 128; la.col_off[0] = 256 + V; la.M[0] = 128;
  la.npairs = 1; la.L = 128; la.out = g_app; la.add = accumulate_latents ? g_app : nullptr;
  LatentJob& lt = F.lat[2];   // articulation: deformation layer 0
  lt.W[0] = params[0]; lt.db[0] = grads[1]; lt.ld[0] = 163; lt.col_off[0] = 131; lt.M[0] = 128;
  lt.npairs = 1; lt.L = 32; lt.out = g_art; lt.add = accumulate_latents ? g_art : nullptr;
  int blk = 3;
  auto outer = [&](int i, const float* db, const float* latent, float* out, int M, int Ll, int ld, int col_off) {
    F.outer[i] = OuterJob{db, latent, out, M, Ll, ld, col_off, blk};
    blk += (M * Ll + 1023) / 1024;
  };
  outer(0, grads[1], shape, grads[0], 128, 128, 163, 3);
  outer(1, grads[1], art, grads[0], 128, 32, 163, 131);
  outer(2, grads[11], shape, grads[10], 256, 128, P + 128, P);
  outer(3, grads[21], shape, grads[20], 256, 128, 256 + P + 128, 256 + P);
  outer(4, grads[27], app, grads[26], 128, 128, 256 + V + 128, 256 + V);
  if (defer) {
    defer->finish = F;
    defer->finish_blocks = blk;
    return hipSuccess;
  }
  art_finish_kernel<<<dim3(blk), dim3(1024), 0, stream>>>(F);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (post && post->side) return hipEventRecord(post->side->join, stream);   // the caller (or the next level's second stage) waits for this
  return hipSuccess;
}

// The deferred second stages of two levels: ONE reduce launch, ONE launch of the (up to) six un-folding products, then the finishing
// kernels of both levels as one launch (art_finish2_kernel; level 1's latent gradients are added onto level 0's).  Every block of every launch does what it does in the per-level
// launches: same bits.
hipError_t launch_art_wgrad_post2(const ArtWgDeferred* d0, const ArtWgDeferred* d1, hipStream_t stream) {
  if (hipError_t e = launch_wgrad_reduce2(d0->reduce, d0->reduce_blocks, d1->reduce, d1->reduce_blocks, stream); e != hipSuccess) return e;
  FoldGemm jobs[6];
  int n = 0;
  for (const ArtWgDeferred* d : {d0, d1})
    for (int j = 0; j < d->n_unfold; ++j) jobs[n++] = d->unfold[j];
  if (n > 0)
    if (hipError_t e = launch_fold_gemms(jobs, n, stream); e != hipSuccess) return e;
  // one finishing launch when level 1 adds onto level 0's latent gradients in place (the training step's arrangement), else two
  bool chained = true;
  for (int k = 0; k < 3; ++k)
  {
    const LatentJob &a = d0->finish.lat[k], &b = d1->finish.lat[k];
    chained = chained && b.add == a.out && b.out == a.out && b.L == a.L && b.npairs == a.npairs;
    for (int pi = 0; pi < a.npairs && chained; ++pi) chained = b.M[pi] == a.M[pi];
  }
  if (chained && d0->finish_blocks >= 3 && d1->finish_blocks >= 3) {
    art_finish2_kernel<<<dim3(d0->finish_blocks + d1->finish_blocks - 3), dim3(1024), 0, stream>>>(d0->finish, d1->finish, d0->finish_blocks);
    return hipGetLastError();
  }
  for (const ArtWgDeferred* d : {d0, d1}) art_finish_kernel<<<dim3(d->finish_blocks), dim3(1024), 0, stream>>>(d->finish);
  return hipGetLastError();
}

}  // namespace aon
