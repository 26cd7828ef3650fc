// Fused vanilla NeRFMLP forward for gfx950 (MI355X / CDNA4): encode -> 8x256 trunk (+skip) -> sigma head,
// bottleneck -> view branch -> rgb head, one launch, activations never leave the register file.
//
// Replaces the eager-op sequence of the reference's  helper.cast_rays + helper.pos_enc + NeRFMLP.forward
// (models/vanilla_nerf/helper.py:25-26,136-140; models/vanilla_nerf/model.py:95-120).
//
// Mapping onto CDNA4 (not a GEMM-library call chain, not a CUDA tiling):
//   * one workgroup = 4 waves = one wave per SIMD; each wave owns 32 samples for the whole network.
//   * every layer is computed TRANSPOSED:  out^T[feature][sample] = W[feature][k] * act^T[k][sample]
//     with v_mfma_f32_32x32x2_f32 (exact fp32, bit-equal to an fmaf chain).  The weights are the A operand,
//     the activations the B operand.  In that orientation the accumulator layout of a 32x32 tile
//     (lane l, reg r  ->  feature (r&3)+8*(r>>2)+4*(l>>5), sample l&31) is *already* a valid B-operand
//     layout for the next layer: lanes 0-31 supply "k=0", lanes 32-63 "k=1" of a 2-deep MFMA step, and the
//     k order of a dot product is free as long as the weight operand is permuted the same way.  The
//     host-side pack kernel bakes that permutation into the weight stream once, so activations stay in
//     VGPR/AGPRs from the positional encoding to the rgb/sigma heads: no LDS or HBM round trip per layer.
//   * weights stream HBM/L2 -> LDS with LDS-DMA (global_load_lds_dwordx4, lane-linear image) in 32 KiB chunks
//     (one 32-feature input tile x all output tiles), double-buffered; the 4 waves share each chunk and
//     read their A operands with conflict-free ds_read_b128 (4 MFMA steps per read).
//   * per 128-sample pass a wave issues ~9.3k MFMAs (64 cycles each) against ~2.4k ds_read_b128 and a few
//     hundred VALU ops (bias init, ReLU, heads), so the kernel is bound by the fp32 matrix pipe.
#include "aon_fold.h"
#include "aon_mlp_core.h"

namespace aon {

struct VanillaNet {
  static constexpr int kSlotBytes = kPairSlotBytes;  // a slot holds a pair of chunks
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = aon::kNumChunks;
  static constexpr int chunk_bytes(int c) { return aon::chunk_bytes(c); }
};
// bottleneck_layer folded into views_linear[0] (aon_common.h): chunks 0 .. 59 of the literal stream, the view-encoding chunk, 8 small chunks of W'
struct VanillaFoldNet {
  static constexpr int kSlotBytes = kPairSlotBytes;
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = aon::kNumChunksF;
  static constexpr int chunk_bytes(int c) { return aon::chunk_bytes_f(c); }
};
// ... with the view-encoding term as a per-ray bias: the same buffer without its chunk 60 (chunk c >= 60 here is chunk c + 1 there)
struct VanillaFoldVbNet {
  static constexpr int kSlotBytes = kPairSlotBytes;
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = aon::kNumChunksF - 1;
  static constexpr int chunk_bytes(int c) { return aon::chunk_bytes_f(c < kChFView ? c : c + 1); }
  static constexpr int skip_before(int c) { return c == kChFView ? kSmallChunkBytes : 0; }
};

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
struct PackArgs {
  const float* p[kNumVanillaParams];
};

// (pos_col_in / view_col_in: aon_mlp_core.h)
// FOLD: the folded form (aon_common.h).  W' / b' were written to packed + kFoldTmpOff by launch_fold_view on the same stream; the pack
// kernel's own writes stay below kStreamBytesF or at / above kStreamBytes, so it never overwrites what it reads.
template <bool FOLD>
__global__ void pack_vanilla_kernel(PackArgs a, float* __restrict__ packed, int L, int Lv) {
  const int P = 3 + 6 * L, V = 3 + 6 * Lv;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int64_t stream_floats = kStreamBytes / 4;
  constexpr int64_t used_floats = FOLD ? kStreamBytesF / 4 : stream_floats;
  const float* Wf = packed + kFoldTmpOff / 4;      // [FOLD] (128, 256), then b' (128)
  if (idx >= stream_floats + kSmallFloats) return;
  if (idx >= used_floats && idx < stream_floats) return;   // [FOLD] the fold temporaries / unused tail of the literal-size buffer
  if (idx >= stream_floats) {  // resident small vectors
    const int s = (int)(idx - stream_floats);
    float v = 0.f;
    if (s < kSmBiasBott) v = a.p[2 * (s >> 8) + 1][s & 255];
    else if (s < kSmBiasView) v = a.p[19][s - kSmBiasBott];
    else if (s < kSmWSigma) v = FOLD ? Wf[128 * 256 + s - kSmBiasView] : a.p[17][s - kSmBiasView];
    else if (s < kSmWRgb) v = a.p[20][s - kSmWSigma];
    else if (s < kSmBSigma) v = a.p[22][s - kSmWRgb];
    else if (s < kSmBRgb) v = a.p[21][0];
    else if (s < kSmBRgb + 3) v = a.p[23][s - kSmBRgb];
    packed[idx] = v;
    return;
  }
  int c, r, nt;
  constexpr int nbig = FOLD ? kChFView : kNumBigChunks;
  if (idx < (int64_t)nbig * (kBigChunkBytes / 4)) {
    c = (int)(idx / (kBigChunkBytes / 4)); r = (int)(idx % (kBigChunkBytes / 4)); nt = 8;
  } else {
    const int64_t i2 = idx - (int64_t)nbig * (kBigChunkBytes / 4);
    c = nbig + (int)(i2 / (kSmallChunkBytes / 4)); r = (int)(i2 % (kSmallChunkBytes / 4)); nt = 4;
    // the folded view chunks take the literal view layer's branches below, with W' for the hidden columns; the view-encoding chunk is the
    // FIRST of the folded view layer (aon_common.h), the last of the literal one
    if constexpr (FOLD) c = c == kChFView ? kChView + 8 : c - 1 + (kChView - kChFView);
  }
  const int cc = r & 3, lane = (r >> 2) & 63, rest = r >> 8;
  const int tp = rest % nt, q = rest / nt;
  const int h = lane >> 5, row = 32 * tp + (lane & 31);
  const float* W; int ld, n_out = 256, col;
  const int hid_col = 8 * q + 4 * h + cc;  // + 32*T
  auto pcol = [&](int c63) { return c63 < 0 ? -1 : pos_col_in(c63, L); };
  auto vcol = [&](int c27) { return c27 < 0 ? -1 : view_col_in(c27, Lv); };
  if (c < kChL1) { W = a.p[0]; ld = P; col = pcol(posenc_col(c, q, cc, h)); }
  else if (c < kChL5) { const int l = 1 + (c - kChL1) / 8; W = a.p[2 * l]; ld = 256; col = 32 * ((c - kChL1) % 8) + hid_col; }
  else if (c < kChL5 + 8) { W = a.p[10]; ld = 256 + P; col = 32 * (c - kChL5) + hid_col; }
  else if (c < kChL6) { W = a.p[10]; ld = 256 + P; col = pcol(posenc_col(c - kChL5 - 8, q, cc, h)); if (col >= 0) col += 256; }
  else if (c < kChL7) { W = a.p[12]; ld = 256; col = 32 * (c - kChL6) + hid_col; }
  else if (c < kChBott) { W = a.p[14]; ld = 256; col = 32 * (c - kChL7) + hid_col; }
  else if (c < kChView) { W = a.p[18]; ld = 256; col = 32 * (c - kChBott) + hid_col; }
  else if (c < kChView + 8) {
    W = a.p[16]; ld = 256 + V; n_out = kCondWidth; col = 32 * (c - kChView) + hid_col;
    if constexpr (FOLD) { W = Wf; ld = 256; }
  }
  else { W = a.p[16]; ld = 256 + V; n_out = kCondWidth; col = vcol(viewenc_col(q, cc, h)); if (col >= 0) col += 256; }
  packed[idx] = (col >= 0 && row < n_out) ? W[(int64_t)row * ld + col] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// fused forward
// ---------------------------------------------------------------------------------------------
// One SEGMENT of a launch: a run of 128-sample passes of one network over one ray range (see ArtSeg, aon_mlp_art.hip: the training
// forward merges the fine level of one ray range with the coarse level of another into ONE persistent launch).
struct MlpSeg {
  const char* packed;        // kPackedBytes
  const float* rays_o;       // (n_rays,3)      [ENC_IN_KERNEL]
  const float* rays_d;       // (n_rays,3)
  const float* viewdirs;     // (n_rays,3)
  const float* t_vals;       // (n_rays,S)
  const float* samples_enc;  // (n_rays*S,63)   [!ENC_IN_KERNEL]
  const float* viewdirs_enc; // (n_rays,27)
  float* raw;                // (n_rays*S,4) = raw rgb, raw sigma
  int64_t total;             // n_rays*S
  int S;
  int npass;                 // ceil(total/128)
  float* planes;             // [TRAIN] kPlRows x Np activation planes, step-major (aon_mlp_core.h)
  u32x4* masks;              // [TRAIN] kMaskLayers x (Np*2) ReLU bit masks
  int64_t Np;                // npass * 128
  const float* view_bias;    // [VB] (n_rays,128): b' + W_v0[:, 256:] ve of the ray (view_bias_kernel)
};
struct MlpArgs {
  MlpSeg seg[2];
  int npass_total;           // seg[0].npass + seg[1].npass (seg[1].npass == 0: a one-segment launch)
};

constexpr int kLdsBytes = kRingBytes + (int)kSmallBytes;

// TRAIN additionally stores every layer's input/output activations as step-major planes (aon_mlp_core.h) for the backward pass.
// FOLD: the stream is the folded form -- the view layer reads the post-ReLU layer-7 output through W' (aon_common.h), there is no
// bottleneck layer (and, [TRAIN], no bottleneck rows in the planes: rows kPlBot .. kPlBot + 255 stay unwritten).
// VB (folded form only): the view layer's accumulators start from the ray's b' + W_v0[:, 256:] ve, fetched from `view_bias` while the last
// chunk of layer 7 runs, instead of from b' followed by the view-encoding chunk -- same bits (aon_common.h), 56 MFMAs, 12 sines and a
// 16 KiB chunk fewer per pass; the inference kernel no longer carries the 16 registers of the view encoding through the trunk.
template <bool ENC_IN_KERNEL, bool TRAIN, bool FOLD, bool VB = false>
__global__ void __launch_bounds__(256) mlp_fwd_kernel(MlpArgs args) {
  static_assert(FOLD || !VB, "the per-ray view bias belongs to the folded form");
  using Net = std::conditional_t<VB, VanillaFoldVbNet, std::conditional_t<FOLD, VanillaFoldNet, VanillaNet>>;
  // <false, true>: training on caller-encoded inputs (other encoding degrees in the padded 63 / 27-slot layout): where the
  // in-kernel form re-encodes from x[] / vd[], this one re-reads the encodings.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem + kRingBytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // the step base of the training planes is wave-uniform: keep it scalar

  const int npass0 = args.seg[0].npass;
  int cur = (int)blockIdx.x >= npass0 ? 1 : 0;               // segment of this workgroup's first pass
  auto load_small = [&](const char* packed) {  // resident small vectors -> LDS (visible after the next workgroup barrier)
    const f32x4* src = reinterpret_cast<const f32x4*>(packed + kStreamBytes);
    f32x4* dst = reinterpret_cast<f32x4*>(sm);
    for (int i = tid; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
  };
  load_small(args.seg[cur].packed);

  Pipe p;
  pipe_init<Net>(p, args.seg[cur].packed, smem, wave, lane);  // also publishes the small block just written to LDS

  for (int gpass = blockIdx.x; gpass < args.npass_total; gpass += gridDim.x) {
    const int si = gpass >= npass0 ? 1 : 0;
    if (si != cur) {   // (workgroup-uniform, at most once per launch) the other network's biases / head weights replace the resident block
      __syncthreads();
      load_small(args.seg[si].packed);
      __syncthreads();
      cur = si;
    }
    const MlpSeg& sg = args.seg[si];
    const int pass = gpass - (si ? npass0 : 0);
    {   // weight stream of this pass, and of this workgroup's next one (its first chunk pair is fetched during this pass's last chunk)
      const int nxt = gpass + (int)gridDim.x;
      p.stream = sg.packed;
      p.next_stream = args.seg[(nxt >= npass0 && nxt < args.npass_total) ? 1 : si].packed;
    }
    const int64_t g = (int64_t)pass * 128 + wave * 32 + m;
    const bool valid = g < sg.total;
    const int64_t gc = valid ? g : sg.total - 1;
    const int64_t ray = gc / sg.S;

    // ---- encode: E[0..1] = 63-wide positional encoding, V = 27-wide view encoding, accumulator layout ----
    f32x16 E[2], V;
    float x[3] = {0.f, 0.f, 0.f}, vd[3] = {0.f, 0.f, 0.f};
#ifdef AON_EXP_NOENC     // timing experiment only (WRONG results): no sample fetch, no encoding (tools/exp_tu.sh, profiles/r05_infer_attribution.txt)
    if constexpr (ENC_IN_KERNEL && !TRAIN) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { E[0][r] = 0.25f; E[1][r] = 0.5f; V[r] = 0.125f; }
    } else
#endif
    if constexpr (ENC_IN_KERNEL) {
      const float t = sg.t_vals[gc];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        // helper.cast_rays (helper.py:25-26): origins + t * directions, multiply then add
        x[a] = __fadd_rn(sg.rays_o[ray * 3 + a], __fmul_rn(t, sg.rays_d[ray * 3 + a]));
        vd[a] = sg.viewdirs[ray * 3 + a];
      }
      encode_pos(x, h, E);
      if constexpr (!VB || TRAIN) encode_view(vd, h, V);   // [VB] only the training planes still want the view encoding
    } else {
      load_pos_enc(sg.samples_enc + gc * kPosEnc, h, E);
      if constexpr (!VB || TRAIN) load_view_enc(sg.viewdirs_enc + ray * kViewEnc, h, V);
    }
    int ray32 = (int)ray;   // [VB] the one value of the prologue that is still needed at the view layer
    (void)ray32;

    PlaneIO io{};
    unsigned moff = 0;
    if constexpr (TRAIN) { io = make_plane_io(sg.planes, kPlRows, (int64_t)pass * 4 + wave_s, m, h); moff = mask_lane_off(pass, tid); }
    if constexpr (TRAIN) {
      store_pos_enc_plane(E, io, kPlE, h);
      store_view_enc_plane(V, io, kPlVE, h);
    }
    // [TRAIN] Every hidden activation tile is stored, and its ReLU decision bits are collected, by the chunk that CONSUMES
    // it (side job of chunk_mma: one value per MFMA group, one 16-byte store per four), so neither a store burst nor a block of
    // mask arithmetic sits at a layer boundary.  `in_row` = plane row of input tile 0; `with_mask` false: the bottleneck output
    // has no activation.
    auto consume = [&](const f32x16 (&in)[8], int in_row, u32x4& mw, bool with_mask) {
      return [&, in_row, with_mask](int j) {
        return [&, in_row, with_mask, j](int i) {
          if constexpr (TRAIN) {
            if (i < 16) {
              if ((i & 3) == 0) store_quad<false>(io, in_row + 32 * j, i >> 2, in[j]);
              if (with_mask) mw[j >> 1] = mask_push_post(mw[j >> 1], in[j][i]);
            }
          }
        };
      };
    };
    auto put_mask = [&](const u32x4& mw, int mask_layer) {   // all 8 tiles (32 pushes per word) of the layer have been consumed
      if constexpr (TRAIN)
        *mask_ptr(sg.masks, sg.Np, mask_layer, moff) = u32x4{mask_word_finish(mw[0]), mask_word_finish(mw[1]), mask_word_finish(mw[2]), mask_word_finish(mw[3])};
    };
    f32x16 X[8], Y[8];
    u32x4 mw;
    // L0: enc(63) -> 256
    init_bias(X, sm + kSmBias + 0 * 256, h);
    chunk_mma<Net, kChL0 + 0, 8, 16>(p, E[0], X);
    chunk_mma<Net, kChL0 + 1, 8, 16>(p, E[1], X);
    relu_tiles(X);
    // L1..L4
    mw = u32x4{0u, 0u, 0u, 0u}; init_bias(Y, sm + kSmBias + 1 * 256, h); dense_layer<Net, kChL1 + 0, 8, 8>(p, X, Y, consume(X, plane_h(0), mw, true)); put_mask(mw, 0); relu_tiles(Y);
    mw = u32x4{0u, 0u, 0u, 0u}; init_bias(X, sm + kSmBias + 2 * 256, h); dense_layer<Net, kChL1 + 8, 8, 8>(p, Y, X, consume(Y, plane_h(1), mw, true)); put_mask(mw, 1); relu_tiles(X);
    mw = u32x4{0u, 0u, 0u, 0u}; init_bias(Y, sm + kSmBias + 3 * 256, h); dense_layer<Net, kChL1 + 16, 8, 8>(p, X, Y, consume(X, plane_h(2), mw, true)); put_mask(mw, 2); relu_tiles(Y);
    mw = u32x4{0u, 0u, 0u, 0u}; init_bias(X, sm + kSmBias + 4 * 256, h); dense_layer<Net, kChL1 + 24, 8, 8>(p, Y, X, consume(Y, plane_h(3), mw, true)); put_mask(mw, 3); relu_tiles(X);
    // L5: cat[h(256), enc(63)] -> 256     (model.py:102-103: concat after layer 4's ReLU)
    mw = u32x4{0u, 0u, 0u, 0u}; init_bias(Y, sm + kSmBias + 5 * 256, h);
    dense_layer<Net, kChL5, 8, 8>(p, X, Y, consume(X, plane_h(4), mw, true)); put_mask(mw, 4);
    // (the in-kernel encoding stays live across layers 1-4, 32 registers, as in the inference kernel; rounds 2-4 re-encoded it here:
    // see art_mlp_fwd_kernel)
#ifdef AON_TRAIN_REENCODE
    if constexpr (TRAIN && ENC_IN_KERNEL) {  // (x made opaque: otherwise the two identical encodings are merged and the first stays live)
      asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
      encode_pos(x, h, E);
    }
#endif
    if constexpr (TRAIN && !ENC_IN_KERNEL) {
      int64_t gq = gc;
      asm volatile("" : "+v"(gq));   // opaque: a second read, not the first one kept live
      load_pos_enc(sg.samples_enc + gq * kPosEnc, h, E);
    }
    chunk_mma<Net, kChL5 + 8, 8, 16>(p, E[0], Y);
    chunk_mma<Net, kChL5 + 9, 8, 16>(p, E[1], Y);
    relu_tiles(Y);
    // L6, L7
    mw = u32x4{0u, 0u, 0u, 0u}; init_bias(X, sm + kSmBias + 6 * 256, h); dense_layer<Net, kChL6, 8, 8>(p, Y, X, consume(Y, plane_h(5), mw, true)); put_mask(mw, 5); relu_tiles(X);
    f32x16 Z[4];
    mw = u32x4{0u, 0u, 0u, 0u}; init_bias(Y, sm + kSmBias + 7 * 256, h);
    if constexpr (VB) {
      // the ray's view bias straight into the view layer's accumulators, one 16-byte load per side slot of layer 7's LAST chunk (tiles 0 .. 6
      // of X are dead by then): a chunk of MFMAs (3.4 us) between the loads and their first use
      asm volatile("" : "+v"(ray32));
      const float* vb = sg.view_bias + (int64_t)ray32 * kCondWidth + 4 * h;
      auto l7_side = [&](int j) {
        auto base = consume(X, plane_h(6), mw, true)(j);
        return [&, base, j, vb](int i) {
          base(i);
          if (j == 7 && i < 16) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(vb + 32 * (i >> 2) + 8 * (i & 3));
            Z[i >> 2][4 * (i & 3)] = v[0]; Z[i >> 2][4 * (i & 3) + 1] = v[1]; Z[i >> 2][4 * (i & 3) + 2] = v[2]; Z[i >> 2][4 * (i & 3) + 3] = v[3];
          }
        };
      };
      dense_layer<Net, kChL7, 8, 8>(p, X, Y, l7_side);
    } else {
      dense_layer<Net, kChL7, 8, 8>(p, X, Y, consume(X, plane_h(6), mw, true));
    }
    put_mask(mw, 6); relu_tiles(Y);
    // density head (model.py:105) on the post-ReLU layer-7 output
    float sigma = head_partial<8>(Y, sm + kSmWSigma, h);
    sigma = sigma + __shfl_xor(sigma, 32) + sm[kSmBSigma];
    constexpr int kChV = FOLD ? kChFView : kChView;
    auto reencode_view = [&]() {   // [TRAIN] the view encoding again (same function, same bits): 16 registers not held across the trunk
      if constexpr (TRAIN && ENC_IN_KERNEL) {
        asm volatile("" : "+v"(vd[0]), "+v"(vd[1]), "+v"(vd[2]));
        encode_view(vd, h, V);
      }
      if constexpr (TRAIN && !ENC_IN_KERNEL) {
        int64_t rq = ray;
        asm volatile("" : "+v"(rq));
        load_view_enc(sg.viewdirs_enc + rq * kViewEnc, h, V);
      }
    };
    if constexpr (FOLD) {
      // bottleneck (no activation, model.py:109) and the view layer's hidden columns (model.py:110-116) as ONE layer W' = W_v0[:, :256] W_b,
      // b' = W_v0[:, :256] b_b + b_v0 on the post-ReLU layer-7 output; the view-encoding columns FIRST (chunk form) or already in Z (VB)
      mw = u32x4{0u, 0u, 0u, 0u};
      if constexpr (!VB) {
        init_bias(Z, sm + kSmBiasView, h);
        reencode_view();
        chunk_mma<Net, kChV, 4, 14>(p, V, Z);
      }
      dense_layer<Net, kChV + (VB ? 0 : 1), 8, 4>(p, Y, Z, consume(Y, plane_h(7), mw, true)); put_mask(mw, 7);
    } else {
      // bottleneck, no activation (model.py:109)
      mw = u32x4{0u, 0u, 0u, 0u}; init_bias(X, sm + kSmBiasBott, h); dense_layer<Net, kChBott, 8, 8>(p, Y, X, consume(Y, plane_h(7), mw, true)); put_mask(mw, 7);
      // view branch: cat[bottleneck(256), viewenc(27)] -> 128, ReLU (model.py:110-116)
      init_bias(Z, sm + kSmBiasView, h);
      dense_layer<Net, kChV, 8, 4>(p, X, Z, consume(X, kPlBot, mw, false));
      reencode_view();
      chunk_mma<Net, kChV + 8, 4, 14>(p, V, Z);
    }
    relu_tiles(Z);
    if constexpr (TRAIN) {  // the view layer's output feeds the rgb head on the VALU: no consuming chunk, 64 values stored here
      *mask_ptr(sg.masks, sg.Np, 8, moff) = relu_mask_bits(Z);   // burst form: already in the stored bit layout
      store_plane(Z, io, kPlHV);
    }
    // rgb head (model.py:118)
    float rgb[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float v = head_partial<4>(Z, sm + kSmWRgb + ch * kCondWidth, h);
      rgb[ch] = v + __shfl_xor(v, 32) + sm[kSmBRgb + ch];
    }
    if (valid && h == 0) {
      f32x4 o; o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; o[3] = sigma;
      reinterpret_cast<f32x4*>(sg.raw)[g] = o;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last prefetched chunk must land before the LDS is released
}

// ---------------------------------------------------------------------------------------------
// host launchers (called from the C ABI)
// ---------------------------------------------------------------------------------------------
// The form packed is the process default at the time of the call (aon_set_bottleneck_fold), remembered for `packed` (stream_form).
// W' and b' of one network into the fold area of its forward stream (the jobs launch_pack_vanilla runs unless told they have been run)
void vanilla_fold_jobs_fwd(const float* const* params, float* packed, int view_levels, FoldGemm jobs[2]) {
  float* Wf = packed + kFoldTmpOff / 4;
  fold_view_jobs(params[16], 256 + 3 + 6 * view_levels, params[17], params[18], params[19], Wf, Wf + 128 * 256, jobs);
}

hipError_t launch_pack_vanilla(const float* const* params, float* packed, hipStream_t stream, int pos_levels, int view_levels, bool fold_done) {
  PackArgs a;
  for (int i = 0; i < kNumVanillaParams; ++i) a.p[i] = params[i];
  const int64_t n = kStreamBytes / 4 + kSmallFloats;
  const int form = fold_default();
  set_stream_form(packed, form);
  if (form == kFormFolded) {
    if (!fold_done) {   // (aon_vanilla_pack_step runs the products of both networks and both directions as ONE launch in front)
      FoldGemm jobs[2];
      vanilla_fold_jobs_fwd(params, packed, view_levels, jobs);
      if (hipError_t e = launch_fold_gemms(jobs, 2, stream); e != hipSuccess) return e;
    }
    pack_vanilla_kernel<true><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed, pos_levels, view_levels);
  } else {
    pack_vanilla_kernel<false><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed, pos_levels, view_levels);
  }
  return hipGetLastError();
}

int num_cus() {  // CUs of the CURRENT device, cached per device ordinal (ops.py may drive any tensor.device)
  static std::atomic<int> cache[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev < 0 || dev >= kMaxDevices) return -1;
  int cus = cache[dev].load(std::memory_order_relaxed);
  if (cus == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    cus = prop.multiProcessorCount;
    cache[dev].store(cus, std::memory_order_relaxed);
  }
  return cus;
}

template <bool ENC, bool TRAIN, bool FOLD, bool VB = false>
static hipError_t launch_mlp_tf(const MlpArgs& args, hipStream_t stream) {
  static DeviceOnce lds_once;  // one per template instance
  if (hipError_t e = set_max_lds(&mlp_fwd_kernel<ENC, TRAIN, FOLD, VB>, kLdsBytes, lds_once); e != hipSuccess) return e;
  const int g_num_cus = num_cus();
  if (g_num_cus <= 0) return hipErrorInvalidDevice;
  const int grid = args.npass_total < g_num_cus ? args.npass_total : g_num_cus;
  if (grid <= 0) return hipSuccess;
  mlp_fwd_kernel<ENC, TRAIN, FOLD, VB><<<dim3(grid), dim3(256), kLdsBytes, stream>>>(args);
  return hipGetLastError();
}

// the kernel of the form the launch's streams were packed in; the segments of one launch must agree
template <bool ENC, bool TRAIN>
static hipError_t launch_mlp_t(const MlpArgs& args, hipStream_t stream) {
  const int form = stream_form(args.seg[0].packed);
  if (form == kFormUnknown) return hipErrorInvalidValue;   // never packed / declared (a copy): refuse instead of guessing
  if (args.seg[1].npass > 0 && stream_form(args.seg[1].packed) != form) return hipErrorInvalidValue;
  // the per-ray view bias: every segment of the launch or none (whole-path calls on the in-kernel encodings; folded form only)
  const bool vb = args.seg[0].view_bias != nullptr;
  if (args.seg[1].npass > 0 && (args.seg[1].view_bias != nullptr) != vb) return hipErrorInvalidValue;
  if (vb && form != kFormFolded) return hipErrorInvalidValue;
  if constexpr (ENC) {
    if (vb) return launch_mlp_tf<ENC, TRAIN, true, true>(args, stream);
  } else {
    if (vb) return hipErrorInvalidValue;
  }
  return form == kFormFolded ? launch_mlp_tf<ENC, TRAIN, true>(args, stream) : launch_mlp_tf<ENC, TRAIN, false>(args, stream);
}

// b' + W_v0[:, 256:] ve per ray, the head of the view layer's accumulation chains (aon_common.h): the fused multiply-adds the view-encoding
// chunk performs, in its order -- register r = 0 .. 13 of the encoding tile, half-wave 0 then 1 (one v_mfma_f32_32x32x2_f32 step adds the
// k = 0 product, then the k = 1 product) -- with the weights read from the packed chunk itself.  One workgroup serves eight rays.
__global__ __launch_bounds__(256) void view_bias_kernel(const float* chunk, const float* bias_vec, const float* viewdirs, int64_t n_rays, float* out) {
  __shared__ float enc[8][2][16];
  const int tid = threadIdx.x;
  const int64_t ray0 = (int64_t)blockIdx.x * 8;
  if (tid < 16) {
    const int64_t ray = ray0 + (tid >> 1);
    if (ray < n_rays) {
      float vd[3] = {viewdirs[ray * 3], viewdirs[ray * 3 + 1], viewdirs[ray * 3 + 2]};
      f32x16 V;
      encode_view(vd, tid & 1, V);
#pragma unroll
      for (int r = 0; r < 16; ++r) enc[tid >> 1][tid & 1][r] = V[r];
    }
  }
  const int row = tid & 127, tp = row >> 5;
  const float bias = bias_vec[row];
  float w[14][2];
#pragma unroll
  for (int r = 0; r < 14; ++r)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) w[r][hh] = chunk[((((r >> 2) * 4 + tp) * 64 + hh * 32 + (row & 31)) << 2) + (r & 3)];
  __syncthreads();
  for (int k = tid >> 7; k < 8; k += 2) {
    const int64_t ray = ray0 + k;
    if (ray >= n_rays) break;
    float acc = bias;
#pragma unroll
    for (int r = 0; r < 14; ++r) {
      acc = __builtin_fmaf(w[r][0], enc[k][0][r], acc);
      acc = __builtin_fmaf(w[r][1], enc[k][1][r], acc);
    }
    out[ray * kCondWidth + row] = acc;
  }
}

// chunk: the view-encoding chunk of a folded stream (4 output tiles); bias_vec: the 128 biases the chunk form starts the view layer from
hipError_t launch_view_bias_raw(const float* chunk, const float* bias_vec, const float* viewdirs, int64_t n_rays, float* out, hipStream_t stream) {
  if (n_rays <= 0) return hipSuccess;
  view_bias_kernel<<<dim3((unsigned)((n_rays + 7) / 8)), dim3(256), 0, stream>>>(chunk, bias_vec, viewdirs, n_rays, out);
  return hipGetLastError();
}

hipError_t launch_view_bias(const char* packed, const float* viewdirs, int64_t n_rays, float* out, hipStream_t stream) {
  if (stream_form(packed) != kFormFolded) return hipErrorInvalidValue;
  return launch_view_bias_raw(reinterpret_cast<const float*>(packed + chunk_offset_f(kChFView)), reinterpret_cast<const float*>(packed + kStreamBytes) + kSmBiasView,
                              viewdirs, n_rays, out, stream);
}

hipError_t launch_mlp_fwd(const char* packed, const float* rays_o, const float* rays_d, const float* viewdirs,
                          const float* t_vals, int64_t n_rays, int S, float* raw, hipStream_t stream, const float* view_bias) {
  MlpArgs args{};
  MlpSeg& a = args.seg[0];
  a.packed = packed; a.rays_o = rays_o; a.rays_d = rays_d; a.viewdirs = viewdirs; a.t_vals = t_vals; a.view_bias = view_bias;
  a.raw = raw; a.total = n_rays * S; a.S = S; a.npass = (int)((a.total + 127) / 128);
  args.seg[1] = a; args.seg[1].npass = 0; args.npass_total = a.npass;
  return launch_mlp_t<true, false>(args, stream);
}

hipError_t launch_mlp_fwd_train2(const TrainSeg* segs, int nsegs, hipStream_t stream);

// Training forward: as launch_mlp_fwd, plus the activation planes (kPlRows x Np floats, Np = 128*ceil(n*S/128)).
// np_total: the padded sample count of the WHOLE batch when this launch covers a ray range of it starting on a pass boundary
// (planes / masks / raw / t_vals already point at the range): the decision bits' slot stride is the whole batch's.
hipError_t launch_mlp_fwd_train(const char* packed, const float* rays_o, const float* rays_d, const float* viewdirs,
                                const float* t_vals, int64_t n_rays, int S, float* raw, float* planes, void* masks,
                                hipStream_t stream, int64_t np_total, const float* view_bias) {
  const TrainSeg one{packed, nullptr, rays_o, rays_d, viewdirs, t_vals, n_rays, S, raw, planes, masks, np_total, view_bias};
  return launch_mlp_fwd_train2(&one, 1, stream);
}

// one or two segments in ONE persistent launch (see MlpSeg)
hipError_t launch_mlp_fwd_train2(const TrainSeg* segs, int nsegs, hipStream_t stream) {
  if (nsegs < 1 || nsegs > 2) return hipErrorInvalidValue;
  MlpArgs args{};
  for (int i = 0; i < nsegs; ++i) {
    const TrainSeg& t = segs[i];
    MlpSeg& a = args.seg[i];
    a.packed = t.packed; a.rays_o = t.rays_o; a.rays_d = t.rays_d; a.viewdirs = t.viewdirs; a.t_vals = t.t_vals;
    a.raw = t.raw; a.total = t.n_rays * t.S; a.S = t.S; a.npass = (int)((a.total + 127) / 128);
    a.planes = t.planes; a.masks = static_cast<u32x4*>(t.masks); a.Np = t.np_total > 0 ? t.np_total : (int64_t)a.npass * 128;
    a.view_bias = t.view_bias;
    args.npass_total += a.npass;
  }
  if (nsegs == 1) { args.seg[1] = args.seg[0]; args.seg[1].npass = 0; }
  else if (args.seg[0].npass == 0) { args.seg[0] = args.seg[1]; args.seg[1].npass = 0; }   // (an empty first segment: the second one alone)
  return launch_mlp_t<true, true>(args, stream);
}

// training forward on caller-encoded inputs (padded 63 / 27-column layout): planes and decision bits as launch_mlp_fwd_train
hipError_t launch_mlp_fwd_train_enc(const char* packed, const float* samples_enc, const float* viewdirs_enc, int64_t n_rays, int S, float* raw,
                                    float* planes, void* masks, hipStream_t stream, int64_t np_total) {
  MlpArgs args{};
  MlpSeg& a = args.seg[0];
  a.packed = packed; a.samples_enc = samples_enc; a.viewdirs_enc = viewdirs_enc;
  a.raw = raw; a.total = n_rays * S; a.S = S; a.npass = (int)((a.total + 127) / 128);
  a.planes = planes; a.masks = static_cast<u32x4*>(masks); a.Np = np_total > 0 ? np_total : (int64_t)a.npass * 128;
  args.seg[1] = a; args.seg[1].npass = 0; args.npass_total = a.npass;
  return launch_mlp_t<false, true>(args, stream);
}

hipError_t launch_mlp_fwd_enc(const char* packed, const float* samples_enc, const float* viewdirs_enc,
                              int64_t n_rays, int S, float* raw, hipStream_t stream) {
  MlpArgs args{};
  MlpSeg& a = args.seg[0];
  a.packed = packed; a.samples_enc = samples_enc; a.viewdirs_enc = viewdirs_enc;
  a.raw = raw; a.total = n_rays * S; a.S = S; a.npass = (int)((a.total + 127) / 128);
  args.seg[1] = a; args.seg[1].npass = 0; args.npass_total = a.npass;
  return launch_mlp_t<false, false>(args, stream);
}

}  // namespace aon
