// Fused articulated NeRFMLP forward for gfx950: deformation MLP -> positional encoding of the deformed point ->
// latent-conditioned 8x256 trunk (+skip) -> sigma head, bottleneck -> 4x128 view branch -> rgb head.
//
// Replaces the reference's  models/vanilla_nerf/model_autodecoder.py:172-239 (NeRFMLP.forward with
// deformation_mlp=True, enc_after=True) preceded by helper.cast_rays (helper.py:25-26).
//
// Same register-resident transposed-MFMA scheme as aon_mlp.hip (see its header).  What is specific here:
//   * the three latents (shape 128, appearance 128, articulation 32) are broadcast to every sample by the
//     reference (einops.repeat, :186-194), i.e. every weight column that multiplies a latent contributes a
//     per-call constant.  aon_art_prepare folds those columns into effective bias vectors once per call
//     (163->3, 191->63, 447->319, 411->283 effective K), so the kernel never touches them.
//   * the 3->128 first deformation layer and the 128->3 deformation head are VALU work on register tiles;
//     the deformed point is encoded in registers exactly like the vanilla path.
#include "aon_art_common.h"
#include "aon_fold.h"

namespace aon {

struct ArtNet {
  static constexpr int kSlotBytes = kPairSlotBytes;  // a slot holds a pair of chunks
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = kANumChunks;
  static constexpr int chunk_bytes(int c) { return (c < kAChT0 || c >= kAChV0) ? kSmallChunkBytes : kBigChunkBytes; }
  static constexpr int64_t chunk_offset(int c) {
    return c < kAChT0 ? (int64_t)c * kSmallChunkBytes
         : c < kAChV0 ? (int64_t)kAChT0 * kSmallChunkBytes + (int64_t)(c - kAChT0) * kBigChunkBytes
                      : (int64_t)kAChT0 * kSmallChunkBytes + (int64_t)(kAChV0 - kAChT0) * kBigChunkBytes +
                        (int64_t)(c - kAChV0) * kSmallChunkBytes;
  }
};
constexpr int64_t kAStreamBytes = ArtNet::chunk_offset(kANumChunks);  // 2,768,896 B

struct ArtFoldNet {   // folded form (aon_art_common.h)
  static constexpr int kSlotBytes = kPairSlotBytes;
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = kANumChunksF;
  static constexpr int chunk_bytes(int c) { return (c < kAChT0 || c >= kAChFV0) ? kSmallChunkBytes : kBigChunkBytes; }
};
struct ArtFoldVbNet {   // folded form, view-encoding term as a per-ray bias: chunk c >= kAChFV0 here is chunk c + 1 of ArtFoldNet
  static constexpr int kSlotBytes = kPairSlotBytes;
  static constexpr bool kPair = true;
  static constexpr int kNumChunks = kANumChunksF - 1;
  static constexpr int chunk_bytes(int c) { return (c < kAChT0 || c >= kAChFV0) ? kSmallChunkBytes : kBigChunkBytes; }
  static constexpr int skip_before(int c) { return c == kAChFV0 ? kSmallChunkBytes : 0; }
};
constexpr int64_t kAStreamBytesF = (int64_t)kAChT0 * kSmallChunkBytes + (int64_t)(kAChFV0 - kAChT0) * kBigChunkBytes + (int64_t)(kANumChunksF - kAChFV0) * kSmallChunkBytes;
constexpr int64_t kAFoldTmpOff = kAStreamBytesF;   // W' (128 x 256 floats) for the pack kernel, inside the literal-size buffer
static_assert(kAStreamBytesF + 128 * 256 * 4 <= kAStreamBytes, "fold temporary fits behind the folded stream");

struct ArtPackArgs {
  const float* p[kNumArtParams];
};

// L / Lv: frequency levels of the network (defaults 10 / 4); the weights' row strides follow: pts_linears.0 (256, P + 128),
// pts_linears.5 (256, 256 + P + 128), views_linear.0 (128, 256 + V + 128) with P = 3 + 6 L, V = 3 + 6 Lv
// FOLD: the folded form; W' was written to packed + kAFoldTmpOff by launch_fold_view on the same stream (b' belongs to the per-call block)
template <bool FOLD, class Args>   // Args: anything with the 40 parameter pointers as `p` (kernel arguments: read in place, never copied)
__device__ __forceinline__ void pack_art_element(const Args& a, float* __restrict__ packed, int L, int Lv, const int64_t idx) {
  const int P = 3 + 6 * L, V = 3 + 6 * Lv;
  auto pcol = [&](int c63) { return c63 < 0 ? -1 : pos_col_in(c63, L); };
  auto vcol = [&](int c27) { return c27 < 0 ? -1 : view_col_in(c27, Lv); };
  if (idx >= (FOLD ? kAStreamBytesF : kAStreamBytes) / 4) return;
  // locate the chunk
  int c, r, nt;
  constexpr int first_small_tail = FOLD ? kAChFV0 : kAChV0;
  const int64_t s0 = (int64_t)kAChT0 * (kSmallChunkBytes / 4);
  const int64_t s1 = s0 + (int64_t)(first_small_tail - kAChT0) * (kBigChunkBytes / 4);
  if (idx < s0) { c = (int)(idx / (kSmallChunkBytes / 4)); r = (int)(idx % (kSmallChunkBytes / 4)); nt = 4; }
  else if (idx < s1) { c = kAChT0 + (int)((idx - s0) / (kBigChunkBytes / 4)); r = (int)((idx - s0) % (kBigChunkBytes / 4)); nt = 8; }
  else {
    c = first_small_tail + (int)((idx - s1) / (kSmallChunkBytes / 4)); r = (int)((idx - s1) % (kSmallChunkBytes / 4)); nt = 4;
    // the view branch's chunks take the literal branches below, W' for views_linear.0's hidden columns; views_linear.0's view-encoding chunk
    // is the FIRST of the folded layer (aon_art_common.h), the last of the literal one
    if constexpr (FOLD) c = c == kAChFV0 ? kAChV0 + 8 : (c <= kAChFV0 + 8 ? c - 1 : c) + (kAChV0 - kAChFV0);
  }
  const int cc = r & 3, lane = (r >> 2) & 63, rest = r >> 8;
  const int tp = rest % nt, q = rest / nt;
  const int h = lane >> 5, row = 32 * tp + (lane & 31);
  const int hid = 8 * q + 4 * h + cc;
  const float* W; int ld, col;
  if (c < kAChT0) { const int l = 1 + c / 4; W = a.p[2 * l]; ld = 128; col = 32 * (c % 4) + hid; }
  else if (c < kAChT1) { W = a.p[10]; ld = P + 128; col = pcol(posenc_col(c - kAChT0, q, cc, h)); }          // cols 0..P-1
  else if (c < kAChT5) { const int l = 1 + (c - kAChT1) / 8; W = a.p[10 + 2 * l]; ld = 256; col = 32 * ((c - kAChT1) % 8) + hid; }
  else if (c < kAChT5 + 8) { W = a.p[20]; ld = 256 + P + 128; col = 32 * (c - kAChT5) + hid; }
  else if (c < kAChT6) { W = a.p[20]; ld = 256 + P + 128; col = pcol(posenc_col(c - kAChT5 - 8, q, cc, h)); if (col >= 0) col += 256; }
  else if (c < kAChT7) { W = a.p[22]; ld = 256; col = 32 * (c - kAChT6) + hid; }
  else if (c < kAChBott) { W = a.p[24]; ld = 256; col = 32 * (c - kAChT7) + hid; }
  else if (c < kAChV0) { W = a.p[34]; ld = 256; col = 32 * (c - kAChBott) + hid; }
  else if (c < kAChV0 + 8) {
    W = a.p[26]; ld = 256 + V + 128; col = 32 * (c - kAChV0) + hid;
    if constexpr (FOLD) { W = packed + kAFoldTmpOff / 4; ld = 256; }
  }
  else if (c < kAChV1) { W = a.p[26]; ld = 256 + V + 128; col = vcol(viewenc_col(q, cc, h)); if (col >= 0) col += 256; }
  else { const int l = 1 + (c - kAChV1) / 4; W = a.p[26 + 2 * l]; ld = 128; col = 32 * ((c - kAChV1) % 4) + hid; }
  packed[idx] = col >= 0 ? W[(int64_t)row * ld + col] : 0.f;  // every layer here has a multiple of 32 outputs
}
template <bool FOLD>
__global__ void pack_art_kernel(ArtPackArgs a, float* __restrict__ packed, int L, int Lv) {
  pack_art_element<FOLD>(a, packed, L, Lv, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

struct ArtPrepArgs {
  const float* p[kNumArtParams];
  const float* shape;  // (128) latents["density"]
  const float* app;    // (128) latents["color"]
  const float* art;    // (32)  latents["articulation"]
};

// small[] = plain copies of the small vectors + the latent-folded effective biases
// fold: the block of a FOLDED stream -- views_linear.0's effective bias additionally carries W_v0[:, :256] b_b: that sum is accumulated in fp64 and
// added to the fp32 value of b_v0 + (appearance term) -- which the literal form rounds as well -- so the effective bias is rounded TWICE (the
// fp32 chain of the latent term, then once more with the fp64 sum on top), not once like the vanilla b' (the earlier wording
// claimed one rounding; the un-folded gradients treat the bias as exact either way, and the second rounding is half an ulp of the bias)
__device__ __forceinline__ void prepare_art_element(const ArtPrepArgs& a, float* __restrict__ small, int min_deg, int L, int Lv, int fold, const int s) {
  const int P = 3 + 6 * L, V = 3 + 6 * Lv;
  if (s >= kASmallFloats) return;
  float v = 0.f;
  if (s >= kA_ESC) {   // encoding scales (exact powers of two), 0 for the levels this network lacks and for the pad
    const int l = s - kA_ESC;
    small[s] = l < L ? ldexpf(1.0f, min_deg + l) : 0.f;
    return;
  }
  if (s < kA_WD0) {  // b + W[:,3:131].shape + W[:,131:163].art   (input = cat[pos, shape, articulation], :196-198)
    const float* w = a.p[0] + (int64_t)s * 163;
    float acc = a.p[1][s];
    for (int k = 0; k < 128; ++k) acc = __builtin_fmaf(w[3 + k], a.shape[k], acc);
    for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(w[131 + k], a.art[k], acc);
    v = acc;
  } else if (s < kA_BD) { const int i = s - kA_WD0; v = a.p[0][(int64_t)(i & 127) * 163 + (i >> 7)]; }
  else if (s < kA_WDL) { const int i = s - kA_BD; v = a.p[2 * (1 + (i >> 7)) + 1][i & 127]; }
  else if (s < kA_BDL) { v = a.p[8][s - kA_WDL]; }
  else if (s < kA_BT) { const int i = s - kA_BDL; v = i < 3 ? a.p[9][i] : 0.f; }
  else if (s < kA_BBOT) {
    const int i = s - kA_BT, l = i >> 8, f = i & 255;
    float acc = a.p[10 + 2 * l + 1][f];
    if (l == 0) {        // cat[enc(63), shape(128)]  (:210)
      const float* w = a.p[10] + (int64_t)f * (P + 128) + P;
      for (int k = 0; k < 128; ++k) acc = __builtin_fmaf(w[k], a.shape[k], acc);
    } else if (l == 5) { // cat[h(256), enc(63), shape(128)]  (:216-217)
      const float* w = a.p[20] + (int64_t)f * (256 + P + 128) + 256 + P;
      for (int k = 0; k < 128; ++k) acc = __builtin_fmaf(w[k], a.shape[k], acc);
    }
    v = acc;
  } else if (s < kA_BV) { v = a.p[35][s - kA_BBOT]; }
  else if (s < kA_WSIG) {
    const int i = s - kA_BV, l = i >> 7, f = i & 127;
    float acc = a.p[26 + 2 * l + 1][f];
    if (l == 0) {        // cat[bottleneck(256), viewenc(27), appearance(128)]  (:228-230)
      const float* w = a.p[26] + (int64_t)f * (256 + V + 128) + 256 + V;
      for (int k = 0; k < 128; ++k) acc = __builtin_fmaf(w[k], a.app[k], acc);
      if (fold) {
        const float* wb = a.p[26] + (int64_t)f * (256 + V + 128);
        double t = 0.0;
        for (int k = 0; k < 256; ++k) t = __builtin_fma((double)wb[k], (double)a.p[35][k], t);
        acc = (float)((double)acc + t);
      }
    }
    v = acc;
  } else if (s < kA_WRGB) { v = a.p[36][s - kA_WSIG]; }
  else if (s < kA_BSIG) { v = a.p[38][s - kA_WRGB]; }
  else if (s < kA_BRGB) { v = a.p[37][0]; }
  else if (s < kA_BRGB + 3) { v = a.p[39][s - kA_BRGB]; }
  small[s] = v;
}
__global__ void prepare_art_kernel(ArtPrepArgs a, float* __restrict__ small, int min_deg, int L, int Lv, int fold) {
  prepare_art_element(a, small, min_deg, L, Lv, fold, blockIdx.x * blockDim.x + threadIdx.x);
}

// Round 6: the per-call blocks AND the forward streams of TWO networks (a training step's coarse and fine level) in one launch --
// blockIdx.y: network; blocks [0, kPrepBlocks): the per-call block, the rest: the stream.  Every element is computed by the code of the
// single-network kernels above (same bits); the folded form expects W' in each stream's fold area (aon_art_pack_step runs the products first).
constexpr int kPrepBlocks = (kASmallFloats + 255) / 256;
struct ArtPackPrep2Args {
  ArtPrepArgs net[2];
  float* small[2];
  float* packed[2];
};
template <bool FOLD>
__global__ void __launch_bounds__(256) pack_prepare_art2_kernel(ArtPackPrep2Args a, int min_deg, int L, int Lv) {
  const ArtPrepArgs& n = a.net[blockIdx.y];
  if ((int)blockIdx.x < kPrepBlocks) {
    prepare_art_element(n, a.small[blockIdx.y], min_deg, L, Lv, FOLD ? 1 : 0, (int)blockIdx.x * 256 + (int)threadIdx.x);
  } else {
    pack_art_element<FOLD>(n, a.packed[blockIdx.y], L, Lv, (int64_t)((int)blockIdx.x - kPrepBlocks) * 256 + threadIdx.x);
  }
}

// One SEGMENT of a launch: a run of 128-sample passes of one network over one ray range.  A launch carries one or two of them
// (round 4): the training forward merges the fine level of ray range A with the coarse level of ray range B -- different weight
// streams, small blocks, rays, planes -- into ONE persistent launch, so that the partial last round of the 65-sample level
// (8.125 rounds of 256 workgroups cost 9) is filled with passes of the other segment instead of idling 7/8 of the chip.
struct ArtSeg {
  const char* packed;     // kAStreamBytes
  const float* small;     // kASmallFloats (from prepare_art_kernel)
  const float* rays_o;    // [POS_IN_KERNEL]
  const float* rays_d;
  const float* viewdirs;
  const float* t_vals;
  const float* pos;          // (n*S,3)   [!POS_IN_KERNEL]
  const float* viewdirs_enc; // (n,27)
  float* raw;             // (n*S,4)
  int64_t total;
  int S;
  int npass;
  float* planes;          // [TRAIN] kAPlRows x Np
  u32x4* masks;           // [TRAIN] kAMaskLayers x (Np*2)
  int64_t Np;
  const float* view_bias; // [VB] (n_rays,128): views_linear.0's effective bias + W_v0[:, 256:283] ve of the ray (launch_art_view_bias)
};
struct ArtMlpArgs {
  ArtSeg seg[2];
  int npass_total;        // seg[0].npass + seg[1].npass (seg[1].npass == 0: a one-segment launch)
};

// FOLD: stream and per-call block are the folded form's (aon_art_common.h): views_linear.0 reads the post-ReLU layer-7 output through W';
// no bottleneck layer and, [TRAIN], no bottleneck rows in the planes (rows kAPlBot .. kAPlBot + 255 stay unwritten).
// VB (folded form, in-kernel ray cast): views_linear.0's accumulators start from the ray's bias + view-encoding term (see mlp_fwd_kernel).
template <bool POS_IN_KERNEL, bool TRAIN, bool FOLD, bool VB = false>
__global__ void __launch_bounds__(256) art_mlp_fwd_kernel(ArtMlpArgs args) {
  static_assert((FOLD && POS_IN_KERNEL) || !VB, "the per-ray view bias belongs to the folded form of the whole-path kernels");
  using Net = std::conditional_t<VB, ArtFoldVbNet, std::conditional_t<FOLD, ArtFoldNet, ArtNet>>;
  constexpr int kV0 = FOLD ? kAChFV0 : kAChV0, kV1 = (FOLD ? kAChFV1 : kAChV1) - (VB ? 1 : 0);
  static_assert(!TRAIN || POS_IN_KERNEL, "the training path encodes the view direction from vd[], which only the in-kernel ray cast fills");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem + kRingBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // the step base of the training planes is wave-uniform: keep it scalar
  const int npass0 = args.seg[0].npass;
  int cur = (int)blockIdx.x >= npass0 ? 1 : 0;               // segment of this workgroup's first pass
  auto load_small = [&](const float* small) {
    const f32x4* src = reinterpret_cast<const f32x4*>(small);
    f32x4* dst = reinterpret_cast<f32x4*>(sm);
    for (int i = tid; i < kASmallFloats / 4; i += 256) dst[i] = src[i];
  };
  load_small(args.seg[cur].small);
  Pipe p;
  pipe_init<Net>(p, args.seg[cur].packed, smem, wave, lane);  // also publishes the small block just written to LDS

  for (int gpass = blockIdx.x; gpass < args.npass_total; gpass += gridDim.x) {
    const int si = gpass >= npass0 ? 1 : 0;
    if (si != cur) {   // (workgroup-uniform, at most once per launch) the other network's biases / head weights replace the resident block
      __syncthreads();
      load_small(args.seg[si].small);
      __syncthreads();
      cur = si;
    }
    const ArtSeg& sg = args.seg[si];
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
    float x[3], vd[3] = {0.f, 0.f, 0.f};
    f32x16 V;
    if constexpr (POS_IN_KERNEL) {
      const float t = sg.t_vals[gc];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        x[a] = __fadd_rn(sg.rays_o[ray * 3 + a], __fmul_rn(t, sg.rays_d[ray * 3 + a]));
        vd[a] = sg.viewdirs[ray * 3 + a];
      }
      if constexpr (!TRAIN && !VB) encode_view(vd, h, V);  // [TRAIN] encoded where the view branch needs it: 16 registers not held across the trunk
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) x[a] = sg.pos[gc * 3 + a];
      load_view_enc(sg.viewdirs_enc + ray * kViewEnc, h, V);
    }

    PlaneIO io{};
    unsigned moff = 0;
    if constexpr (TRAIN) { io = make_plane_io(sg.planes, kAPlRows, (int64_t)pass * 4 + wave_s, m, h); moff = mask_lane_off(pass, tid); }
    // [TRAIN] activation tiles are stored, and their ReLU decision bits collected, by the chunk that CONSUMES them (side job
    // of chunk_mma, one value per MFMA group, one 16-byte store per four); only the tiles consumed on the VALU (deformation head,
    // rgb head) and the VALU-computed first deformation layer's output go out in a burst of 64 values.
    auto side = [&](const f32x16& tile, int row, unsigned& word, bool with_mask) {
      return [&, row, with_mask](int i) {
        if constexpr (TRAIN) {
          if (i < 16) {
            if ((i & 3) == 0) store_quad<false>(io, row, i >> 2, tile);
            if (with_mask) word = mask_push_post(word, tile[i]);
          }
        }
      };
    };
    unsigned mw[4] = {0u, 0u, 0u, 0u};
    auto consume4 = [&](const f32x16 (&in)[4], int in_row, bool with_mask) {
      return [&, in_row, with_mask](int j) { return side(in[j], in_row + 32 * j, mw[j >> 1], with_mask); };
    };
    auto consume8 = [&](const f32x16 (&in)[8], int in_row, bool with_mask) {
      return [&, in_row, with_mask](int j) { return side(in[j], in_row + 32 * j, mw[j >> 1], with_mask); };
    };
    auto put_mask = [&](int slot) {
      if constexpr (TRAIN)   // every word took 0 or 32 pushes (4- and 8-tile layers)
        *mask_ptr(sg.masks, sg.Np, slot, moff) = u32x4{mask_word_finish(mw[0]), mask_word_finish(mw[1]), mask_word_finish(mw[2]), mask_word_finish(mw[3])};
      mw[0] = mw[1] = mw[2] = mw[3] = 0u;
    };
    auto burst = [&](auto& tiles, int row, int mask_slot) {   // VALU-consumed tiles
      if constexpr (TRAIN) {
        store_plane(tiles, io, row);
        *mask_ptr(sg.masks, sg.Np, mask_slot, moff) = relu_mask_bits(tiles);
      }
    };
    auto save_row = [&](int row, float v) {  // one scalar per sample (lanes 0..31)
      if constexpr (TRAIN) { if (h == 0) *row_ptr(io, row) = v; }
    };
    if constexpr (TRAIN) {
#pragma unroll
      for (int a = 0; a < 3; ++a) save_row(kAPlPos + a, x[a]);
    }

    // ---- deformation MLP (:196-205) ----
    f32x16 H0[4], H1[4];
    init_bias(H0, sm + kA_BD0, h);  // effective bias, then the three xyz columns on the VALU
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(sm + kA_WD0 + a * 128 + 32 * t + 8 * gq + 4 * h);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) H0[t][4 * gq + cc] = __builtin_fmaf(w[cc], x[a], H0[t][4 * gq + cc]);
        }
      }
    }
    relu_tiles(H0);
    init_bias(H1, sm + kA_BD + 0 * 128, h); dense_layer<Net, kAChD1 + 0, 4, 4>(p, H0, H1, consume4(H0, aplane_d(0), true)); put_mask(0); relu_tiles(H1);
    init_bias(H0, sm + kA_BD + 1 * 128, h); dense_layer<Net, kAChD1 + 4, 4, 4>(p, H1, H0, consume4(H1, aplane_d(1), true)); put_mask(1); relu_tiles(H0);
    init_bias(H1, sm + kA_BD + 2 * 128, h); dense_layer<Net, kAChD1 + 8, 4, 4>(p, H0, H1, consume4(H0, aplane_d(2), true)); put_mask(2); relu_tiles(H1);
    burst(H1, aplane_d(3), 3);
    float xd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // x' = deformation_layer(h) + pos   (:205)
      float v = head_partial<4>(H1, sm + kA_WDL + a * 128, h);
      v = v + __shfl_xor(v, 32) + sm[kA_BDL + a];
      xd[a] = __fadd_rn(v, x[a]);
    }
    f32x16 E[2];
    encode_pos_scaled(xd, h, sm + kA_ESC, E);  // pos_enc after the deformation (enc_after=True, :207-208); scales: the network's degrees
    if constexpr (TRAIN) {
#pragma unroll
      for (int a = 0; a < 3; ++a) save_row(kAPlPos + 3 + a, xd[a]);
      store_pos_enc_plane(E, io, kAPlE, h);
    }

    // ---- trunk (:212-217), shape latent folded into the biases of layers 0 and 5 ----
    f32x16 X[8], Y[8];
    init_bias(X, sm + kA_BT + 0 * 256, h);
    chunk_mma<Net, kAChT0 + 0, 8, 16>(p, E[0], X);
    chunk_mma<Net, kAChT0 + 1, 8, 16>(p, E[1], X);
    relu_tiles(X);
    init_bias(Y, sm + kA_BT + 1 * 256, h); dense_layer<Net, kAChT1 + 0, 8, 8>(p, X, Y, consume8(X, aplane_h(0), true)); put_mask(4); relu_tiles(Y);
    init_bias(X, sm + kA_BT + 2 * 256, h); dense_layer<Net, kAChT1 + 8, 8, 8>(p, Y, X, consume8(Y, aplane_h(1), true)); put_mask(5); relu_tiles(X);
    init_bias(Y, sm + kA_BT + 3 * 256, h); dense_layer<Net, kAChT1 + 16, 8, 8>(p, X, Y, consume8(X, aplane_h(2), true)); put_mask(6); relu_tiles(Y);
    init_bias(X, sm + kA_BT + 4 * 256, h); dense_layer<Net, kAChT1 + 24, 8, 8>(p, Y, X, consume8(Y, aplane_h(3), true)); put_mask(7); relu_tiles(X);
    init_bias(Y, sm + kA_BT + 5 * 256, h);
    dense_layer<Net, kAChT5, 8, 8>(p, X, Y, consume8(X, aplane_h(4), true)); put_mask(8);
    // (the encoding stays live across layers 1-4, 32 registers, as in the inference kernel.  Rounds 2-4 re-encoded it here -- 30 sines --
    // to stay clear of spills; with the view encoding gone from the trunk (per-ray view bias) both forms build with 0 scratch and this one
    // is 0.04 ms per step faster: profiles/r05_view_bias_ab.txt)
#ifdef AON_TRAIN_REENCODE
    if constexpr (TRAIN) {
      asm volatile("" : "+v"(xd[0]), "+v"(xd[1]), "+v"(xd[2]));
      encode_pos_scaled(xd, h, sm + kA_ESC, E);
    }
#endif
    chunk_mma<Net, kAChT5 + 8, 8, 16>(p, E[0], Y);
    chunk_mma<Net, kAChT5 + 9, 8, 16>(p, E[1], Y);
    relu_tiles(Y);
    init_bias(X, sm + kA_BT + 6 * 256, h); dense_layer<Net, kAChT6, 8, 8>(p, Y, X, consume8(Y, aplane_h(5), true)); put_mask(9); relu_tiles(X);
    f32x16 Z0[4], Z1[4];
    init_bias(Y, sm + kA_BT + 7 * 256, h);
    if constexpr (VB) {
      // the ray's view bias straight into views_linear.0's accumulators, one 16-byte load per side slot of layer 7's LAST chunk
      int ray32 = (int)ray;
      asm volatile("" : "+v"(ray32));
      const float* vb = sg.view_bias + (int64_t)ray32 * kCondWidth + 4 * h;
      auto l7_side = [&](int j) {
        auto base = consume8(X, aplane_h(6), true)(j);
        return [&, base, j, vb](int i) {
          base(i);
          if (j == 7 && i < 16) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(vb + 32 * (i >> 2) + 8 * (i & 3));
            Z0[i >> 2][4 * (i & 3)] = v[0]; Z0[i >> 2][4 * (i & 3) + 1] = v[1]; Z0[i >> 2][4 * (i & 3) + 2] = v[2]; Z0[i >> 2][4 * (i & 3) + 3] = v[3];
          }
        };
      };
      dense_layer<Net, kAChT7, 8, 8>(p, X, Y, l7_side);
    } else {
      dense_layer<Net, kAChT7, 8, 8>(p, X, Y, consume8(X, aplane_h(6), true));
    }
    put_mask(10); relu_tiles(Y);
    float sigma = head_partial<8>(Y, sm + kA_WSIG, h);  // density_layer (:219)
    sigma = sigma + __shfl_xor(sigma, 32) + sm[kA_BSIG];
    // ---- view branch (:227-234): cat[bottleneck, viewenc, appearance] -> 4 x (128, ReLU) ----
    auto view_enc_here = [&]() {   // [TRAIN] the view encoding where the branch needs it (and its plane rows)
      if constexpr (TRAIN) {
        encode_view(vd, h, V);
        store_view_enc_plane(V, io, kAPlVE, h);
      }
    };
    if constexpr (FOLD) {
      // bottleneck (:223, no activation) and views_linear.0's bottleneck columns as ONE layer W' on the layer-7 output (effective bias incl.
      // W_v0[:, :256] b_b); the view-encoding columns FIRST (chunk form) or already in Z0 with the bias (VB)
      view_enc_here();
      if constexpr (!VB) {
        init_bias(Z0, sm + kA_BV + 0 * 128, h);
        chunk_mma<Net, kV0, 4, 14>(p, V, Z0);
      }
      dense_layer<Net, kV0 + (VB ? 0 : 1), 8, 4>(p, Y, Z0, consume8(Y, aplane_h(7), true)); put_mask(11);
    } else {
      init_bias(X, sm + kA_BBOT, h); dense_layer<Net, kAChBott, 8, 8>(p, Y, X, consume8(Y, aplane_h(7), true)); put_mask(11);  // bottleneck (:223)
      init_bias(Z0, sm + kA_BV + 0 * 128, h);
      dense_layer<Net, kV0, 8, 4>(p, X, Z0, consume8(X, kAPlBot, false));
      view_enc_here();
      chunk_mma<Net, kV0 + 8, 4, 14>(p, V, Z0);
    }
    relu_tiles(Z0);
    init_bias(Z1, sm + kA_BV + 1 * 128, h); dense_layer<Net, kV1 + 0, 4, 4>(p, Z0, Z1, consume4(Z0, aplane_v(0), true)); put_mask(12); relu_tiles(Z1);
    init_bias(Z0, sm + kA_BV + 2 * 128, h); dense_layer<Net, kV1 + 4, 4, 4>(p, Z1, Z0, consume4(Z1, aplane_v(1), true)); put_mask(13); relu_tiles(Z0);
    init_bias(Z1, sm + kA_BV + 3 * 128, h); dense_layer<Net, kV1 + 8, 4, 4>(p, Z0, Z1, consume4(Z0, aplane_v(2), true)); put_mask(14); relu_tiles(Z1);
    burst(Z1, aplane_v(3), 15);
    float rgb[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {  // rgb_layer (:236)
      float v = head_partial<4>(Z1, sm + kA_WRGB + ch * kCondWidth, h);
      rgb[ch] = v + __shfl_xor(v, 32) + sm[kA_BRGB + ch];
    }
    if (valid && h == 0) {
      f32x4 o; o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; o[3] = sigma;
      reinterpret_cast<f32x4*>(sg.raw)[g] = o;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
// The form packed is the process default at the time of the call (aon_set_bottleneck_fold), remembered for `packed` (stream_form).
// W' = W_v0[:, :256] W_b of one network into the fold area of its forward stream (the job launch_pack_art runs unless told it has been run)
FoldGemm art_fold_job_fwd(const float* const* params, float* packed, int view_levels) {
  // (b' is the per-call block's business: prepare_art_kernel; the 128 floats it would take here are not written)
  return FoldGemm{params[26], 256 + 3 + 6 * view_levels + 128, 1, params[34], 256, 1, packed + kAFoldTmpOff / 4, 256, 128, 256, 256, nullptr, nullptr};
}

hipError_t launch_pack_art(const float* const* params, float* packed, hipStream_t stream, int pos_levels, int view_levels, bool fold_done) {
  ArtPackArgs a;
  for (int i = 0; i < kNumArtParams; ++i) a.p[i] = params[i];
  const int form = fold_default();
  set_stream_form(packed, form);
  if (form == kFormFolded) {
    if (!fold_done) {   // (aon_art_pack_step runs the products of both networks and both directions as ONE launch in front)
      const FoldGemm job = art_fold_job_fwd(params, packed, view_levels);
      if (hipError_t e = launch_fold_gemms(&job, 1, stream); e != hipSuccess) return e;
    }
    const int64_t n = kAStreamBytesF / 4;
    pack_art_kernel<true><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed, pos_levels, view_levels);
  } else {
    const int64_t n = kAStreamBytes / 4;
    pack_art_kernel<false><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed, pos_levels, view_levels);
  }
  return hipGetLastError();
}

hipError_t launch_prepare_art(const float* const* params, const float* shape, const float* app, const float* art,
                              float* small, hipStream_t stream, int min_deg, int pos_levels, int view_levels) {
  ArtPrepArgs a;
  for (int i = 0; i < kNumArtParams; ++i) a.p[i] = params[i];
  a.shape = shape; a.app = app; a.art = art;
  // the block's form is the process default at the time of the call, remembered for `small`: a launch refuses a block and a stream of two forms
  const int form = fold_default();
  set_stream_form(small, form);
  prepare_art_kernel<<<dim3((kASmallFloats + 255) / 256), dim3(256), 0, stream>>>(a, small, min_deg, pos_levels, view_levels, form == kFormFolded ? 1 : 0);
  return hipGetLastError();
}

// both networks of a two-level model: per-call blocks + forward streams, one launch (the folded form's W' must be in place: fold_done)
hipError_t launch_pack_prepare_art2(const float* const* const params[2], const float* shape, const float* app, const float* art, float* const packed[2],
                                    float* const small[2], hipStream_t stream, int min_deg, int pos_levels, int view_levels, int form) {
  ArtPackPrep2Args a;   // (form: decided ONCE by the caller -- it has, or has not, run the fold products for it)
  for (int l = 0; l < 2; ++l) {
    for (int i = 0; i < kNumArtParams; ++i) a.net[l].p[i] = params[l][i];
    a.net[l].shape = shape; a.net[l].app = app; a.net[l].art = art;
    a.small[l] = small[l]; a.packed[l] = packed[l];
    set_stream_form(packed[l], form);
    set_stream_form(small[l], form);
  }
  const int64_t n = (form == kFormFolded ? kAStreamBytesF : kAStreamBytes) / 4;
  const dim3 grid((unsigned)(kPrepBlocks + (n + 255) / 256), 2);
  if (form == kFormFolded) pack_prepare_art2_kernel<true><<<grid, dim3(256), 0, stream>>>(a, min_deg, pos_levels, view_levels);
  else pack_prepare_art2_kernel<false><<<grid, dim3(256), 0, stream>>>(a, min_deg, pos_levels, view_levels);
  return hipGetLastError();
}

int num_cus();  // aon_mlp.hip

template <bool POS, bool TRAIN, bool FOLD, bool VB = false>
static hipError_t launch_art_tf(const ArtMlpArgs& args, hipStream_t stream) {
  static DeviceOnce lds_once;
  if (hipError_t e = set_max_lds(&art_mlp_fwd_kernel<POS, TRAIN, FOLD, VB>, kALdsBytes, lds_once); e != hipSuccess) return e;
  const int cus = num_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  const int grid = args.npass_total < cus ? args.npass_total : cus;
  if (grid <= 0) return hipSuccess;
  art_mlp_fwd_kernel<POS, TRAIN, FOLD, VB><<<dim3(grid), dim3(256), kALdsBytes, stream>>>(args);
  return hipGetLastError();
}

// the kernel of the form the launch's streams AND per-call blocks were made in; everything in one launch must agree
template <bool POS, bool TRAIN>
static hipError_t launch_art_t(const ArtMlpArgs& args, hipStream_t stream) {
  const int form = stream_form(args.seg[0].packed);
  if (form == kFormUnknown) return hipErrorInvalidValue;   // never packed / declared (a copy): refuse instead of guessing
  if (stream_form(args.seg[0].small) != form) return hipErrorInvalidValue;
  if (args.seg[1].npass > 0 && (stream_form(args.seg[1].packed) != form || stream_form(args.seg[1].small) != form)) return hipErrorInvalidValue;
  const bool vb = args.seg[0].view_bias != nullptr;   // every segment of the launch or none; folded form, in-kernel ray cast only
  if (args.seg[1].npass > 0 && (args.seg[1].view_bias != nullptr) != vb) return hipErrorInvalidValue;
  if (vb && form != kFormFolded) return hipErrorInvalidValue;
  if constexpr (POS) {
    if (vb) return launch_art_tf<POS, TRAIN, true, true>(args, stream);
  } else {
    if (vb) return hipErrorInvalidValue;
  }
  return form == kFormFolded ? launch_art_tf<POS, TRAIN, true>(args, stream) : launch_art_tf<POS, TRAIN, false>(args, stream);
}

hipError_t launch_view_bias_raw(const float* chunk, const float* bias_vec, const float* viewdirs, int64_t n_rays, float* out, hipStream_t stream);   // aon_mlp.hip

// views_linear.0's effective bias (per-call block: appearance latent and W_v0[:, :256] b_b folded in) + its view-encoding term, per ray
hipError_t launch_art_view_bias(const char* packed, const float* small, const float* viewdirs, int64_t n_rays, float* out, hipStream_t stream) {
  if (stream_form(packed) != kFormFolded || stream_form(small) != kFormFolded) return hipErrorInvalidValue;
  constexpr int64_t off = (int64_t)kAChT0 * kSmallChunkBytes + (int64_t)(kAChFV0 - kAChT0) * kBigChunkBytes;
  return launch_view_bias_raw(reinterpret_cast<const float*>(packed + off), small + kA_BV, viewdirs, n_rays, out, stream);
}

hipError_t launch_art_mlp_fwd(const char* packed, const float* small, const float* rays_o, const float* rays_d,
                              const float* viewdirs, const float* t_vals, int64_t n_rays, int S, float* raw,
                              hipStream_t stream, const float* view_bias) {
  ArtMlpArgs args{};
  ArtSeg& a = args.seg[0];
  a.packed = packed; a.small = small; a.rays_o = rays_o; a.rays_d = rays_d; a.viewdirs = viewdirs; a.t_vals = t_vals; a.view_bias = view_bias;
  a.raw = raw; a.total = n_rays * S; a.S = S; a.npass = (int)((a.total + 127) / 128);
  args.seg[1] = a; args.seg[1].npass = 0; args.npass_total = a.npass;
  return launch_art_t<true, false>(args, stream);
}

hipError_t launch_art_mlp_fwd_train2(const TrainSeg* segs, int nsegs, hipStream_t stream);

hipError_t launch_art_mlp_fwd_train(const char* packed, const float* small, const float* rays_o, const float* rays_d,
                                    const float* viewdirs, const float* t_vals, int64_t n_rays, int S, float* raw, float* planes,
                                    void* masks, hipStream_t stream, int64_t np_total, const float* view_bias) {
  const TrainSeg one{packed, small, rays_o, rays_d, viewdirs, t_vals, n_rays, S, raw, planes, masks, np_total, view_bias};
  return launch_art_mlp_fwd_train2(&one, 1, stream);
}

// one or two segments in ONE persistent launch (see ArtSeg)
hipError_t launch_art_mlp_fwd_train2(const TrainSeg* segs, int nsegs, hipStream_t stream) {
  if (nsegs < 1 || nsegs > 2) return hipErrorInvalidValue;
  ArtMlpArgs args{};
  for (int i = 0; i < nsegs; ++i) {
    const TrainSeg& t = segs[i];
    ArtSeg& a = args.seg[i];
    a.packed = t.packed; a.small = t.small; a.rays_o = t.rays_o; a.rays_d = t.rays_d; a.viewdirs = t.viewdirs; a.t_vals = t.t_vals;
    a.raw = t.raw; a.total = t.n_rays * t.S; a.S = t.S; a.npass = (int)((a.total + 127) / 128);
    a.planes = t.planes; a.masks = static_cast<u32x4*>(t.masks); a.Np = t.np_total > 0 ? t.np_total : (int64_t)a.npass * 128;   // (launch_mlp_fwd_train)
    a.view_bias = t.view_bias;
    args.npass_total += a.npass;
  }
  if (nsegs == 1) { args.seg[1] = args.seg[0]; args.seg[1].npass = 0; }
  else if (args.seg[0].npass == 0) { args.seg[0] = args.seg[1]; args.seg[1].npass = 0; }   // (an empty first segment: the second one alone)
  return launch_art_t<true, true>(args, stream);
}

hipError_t launch_art_mlp_fwd_pos(const char* packed, const float* small, const float* pos, const float* viewdirs_enc,
                                  int64_t n_rays, int S, float* raw, hipStream_t stream) {
  ArtMlpArgs args{};
  ArtSeg& a = args.seg[0];
  a.packed = packed; a.small = small; a.pos = pos; a.viewdirs_enc = viewdirs_enc;
  a.raw = raw; a.total = n_rays * S; a.S = S; a.npass = (int)((a.total + 127) / 128);
  args.seg[1] = a; args.seg[1].npass = 0; args.npass_total = a.npass;
  return launch_art_t<false, false>(args, stream);
}

int64_t art_stream_bytes() { return kAStreamBytes; }
int64_t art_small_bytes() { return (int64_t)kASmallFloats * 4; }

}  // namespace aon
