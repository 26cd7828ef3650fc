// Split-precision ("bf16x3") form of the fused articulated NeRFMLP forward (fp32 twin: aon_mlp_art.hip;
// reference: models/vanilla_nerf/model_autodecoder.py:172-239 with deformation_mlp=True, enc_after=True).
// Same chunk order (101 chunks) and the same per-call small block (aon_art_prepare: latents folded into effective
// biases); every MFMA layer runs on v_mfma_f32_32x32x16_bf16 with weights and activations as three exact bf16 limbs and
// the six limb products of weight >= 2^-24 accumulated in fp32 (aon_bf16_core.h).  Opt-in, like the vanilla bf16x3 engine.
#include "aon_bf16_core.h"
#include "aon_art_common.h"

namespace aon {

struct Bf16ArtNet {
  static constexpr int kNumChunks = kANumChunks;
  static constexpr int kSlotBytes = 8 * 6144;  // 48 KiB
  static constexpr bool kPair = false;
  static constexpr int chunk_tiles(int c) { return (c < kAChT0 || c >= kAChV0) ? 4 : 8; }  // 32-feature output tiles
  static constexpr int chunk_bytes(int c) { return chunk_tiles(c) * 6144; }
};
constexpr int64_t kBaStreamBytes =
    ((int64_t)kAChT0 * 4 + (int64_t)(kAChV0 - kAChT0) * 8 + (int64_t)(kANumChunks - kAChV0) * 4) * 6144;
constexpr int kBaSmallBytes = kASmallFloats * 4;
constexpr int kBaLdsBytes = kBfRingBytes + kBaSmallBytes + kBfEncStashBytes;
static_assert((kBfRingBytes + kBaSmallBytes) % 16 == 0, "stash alignment");

struct ArtPackArgsB {
  const float* p[kNumArtParams];
};

__global__ void pack_art_bf16x3_kernel(ArtPackArgsB a, char* __restrict__ packed) {
  // one thread per (chunk, k16 step s, out tile tp, lane): 8 weights -> 3 x 16 bytes
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int64_t t0 = (int64_t)kAChT0 * 2 * 4 * 64;                      // 16 KiB-class chunks of the deformation MLP
  constexpr int64_t t1 = t0 + (int64_t)(kAChV0 - kAChT0) * 2 * 8 * 64;      // trunk + bottleneck
  constexpr int64_t t2 = t1 + (int64_t)(kANumChunks - kAChV0) * 2 * 4 * 64; // view branch
  if (idx >= t2) return;
  int c, r, nt;
  int64_t chunk_base;
  if (idx < t0) { c = (int)(idx / (2 * 4 * 64)); r = (int)(idx % (2 * 4 * 64)); nt = 4; chunk_base = (int64_t)c * 4 * 6144; }
  else if (idx < t1) {
    const int64_t i2 = idx - t0;
    c = kAChT0 + (int)(i2 / (2 * 8 * 64)); r = (int)(i2 % (2 * 8 * 64)); nt = 8;
    chunk_base = (int64_t)kAChT0 * 4 * 6144 + (int64_t)(c - kAChT0) * 8 * 6144;
  } else {
    const int64_t i2 = idx - t1;
    c = kAChV0 + (int)(i2 / (2 * 4 * 64)); r = (int)(i2 % (2 * 4 * 64)); nt = 4;
    chunk_base = (int64_t)kAChT0 * 4 * 6144 + (int64_t)(kAChV0 - kAChT0) * 8 * 6144 + (int64_t)(c - kAChV0) * 4 * 6144;
  }
  const int lane = r & 63, tp = (r >> 6) % nt, s = (r >> 6) / nt;
  const int h = lane >> 5, row = 32 * tp + (lane & 31);
  const float* W; int ld;
  int kind, tile, off = 0;  // kind 0: hidden columns 32*tile + feature, 1: pos-enc (+off), 2: view-enc (+off)
  if (c < kAChT0) { const int l = 1 + c / 4; W = a.p[2 * l]; ld = 128; kind = 0; tile = c % 4; }
  else if (c < kAChT1) { W = a.p[10]; ld = 191; kind = 1; tile = c - kAChT0; }
  else if (c < kAChT5) { const int l = 1 + (c - kAChT1) / 8; W = a.p[10 + 2 * l]; ld = 256; kind = 0; tile = (c - kAChT1) % 8; }
  else if (c < kAChT5 + 8) { W = a.p[20]; ld = 447; kind = 0; tile = c - kAChT5; }
  else if (c < kAChT6) { W = a.p[20]; ld = 447; kind = 1; tile = c - kAChT5 - 8; off = 256; }
  else if (c < kAChT7) { W = a.p[22]; ld = 256; kind = 0; tile = c - kAChT6; }
  else if (c < kAChBott) { W = a.p[24]; ld = 256; kind = 0; tile = c - kAChT7; }
  else if (c < kAChV0) { W = a.p[34]; ld = 256; kind = 0; tile = c - kAChBott; }
  else if (c < kAChV0 + 8) { W = a.p[26]; ld = 411; kind = 0; tile = c - kAChV0; }
  else if (c < kAChV1) { W = a.p[26]; ld = 411; kind = 2; tile = 0; off = 256; }
  else { const int l = 1 + (c - kAChV1) / 4; W = a.p[26 + 2 * l]; ld = 128; kind = 0; tile = (c - kAChV1) % 4; }
  unsigned short hi[8], mid[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int reg = 8 * s + j;  // accumulator register of the producing tile that supplies k-slot j of step s
    int col;
    if (kind == 0) col = 32 * tile + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    else if (kind == 1) { col = posenc_col(tile, reg >> 2, reg & 3, h); if (col >= 0) col += off; }
    else { col = viewenc_col(reg >> 2, reg & 3, h); if (col >= 0) col += off; }
    const float w = col >= 0 ? W[(int64_t)row * ld + col] : 0.f;  // every layer here has a multiple of 32 outputs
    hi[j] = bf16_rne_bits(w);
    const float r1 = w - bf16_bits_to_f32(hi[j]);
    mid[j] = bf16_rne_bits(r1);
    lo[j] = bf16_rne_bits(r1 - bf16_bits_to_f32(mid[j]));
  }
  char* dst = packed + chunk_base + ((int64_t)(s * nt + tp) * 3) * 1024 + lane * 16;
  auto put = [&](int limb, const unsigned short (&v)[8]) {
    u32x4 o;
    o[0] = v[0] | ((unsigned)v[1] << 16); o[1] = v[2] | ((unsigned)v[3] << 16);
    o[2] = v[4] | ((unsigned)v[5] << 16); o[3] = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<u32x4*>(dst + limb * 1024) = o;
  };
  put(0, hi); put(1, mid); put(2, lo);
}

// 128 -> 128 layer over four input tiles (deformation layers 1-3, view layers 1-3): the input is the previous layer's
// pre-activation (ReLU applied in the split).  On entry `cur` holds the fragments of in[0].
template <int CBASE, bool TRAIN = false>
__device__ __forceinline__ void layer4_bf16(Pipe& p, LimbFrag (&cur)[2], const f32x16 (&in)[4], f32x16 (&out)[4], float* in_plane = nullptr,
                                            const PlaneIO* io = nullptr, int64_t tile_bytes = 0) {
  LimbFrag nxt[2];
  auto tp = [&](int j) { return TRAIN ? reinterpret_cast<float*>(reinterpret_cast<char*>(in_plane) + j * tile_bytes) : nullptr; };
  chunk_mma_bf16<CBASE + 0, 4, true, 1, TRAIN, Bf16ArtNet>(p, cur, out, in[1], nxt, tp(1), io); cur[0] = nxt[0]; cur[1] = nxt[1];
  chunk_mma_bf16<CBASE + 1, 4, true, 1, TRAIN, Bf16ArtNet>(p, cur, out, in[2], nxt, tp(2), io); cur[0] = nxt[0]; cur[1] = nxt[1];
  chunk_mma_bf16<CBASE + 2, 4, true, 1, TRAIN, Bf16ArtNet>(p, cur, out, in[3], nxt, tp(3), io); cur[0] = nxt[0]; cur[1] = nxt[1];
  chunk_mma_bf16<CBASE + 3, 4, false, 0, false, Bf16ArtNet>(p, cur, out, in[3], nxt);
}

struct ArtBfArgs {
  const char* packed;   // kBaStreamBytes
  const float* small;   // kASmallFloats (aon_art_prepare)
  const float* rays_o; const float* rays_d; const float* viewdirs; const float* t_vals;
  float* raw;
  int64_t total; int S; int npass;
  float* planes;   // [TRAIN] kAPlRows x Np activation planes (row map of aon_mlp_art.hip's training forward)
  u32x4* masks;    // [TRAIN] kAMaskLayers x (Np*2) ReLU bit masks
  int64_t Np;
};

template <bool TRAIN>
__global__ void __launch_bounds__(256) art_mlp_fwd_bf16x3_kernel(ArtBfArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem + kBfRingBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(args.small);
    f32x4* dst = reinterpret_cast<f32x4*>(sm);
    for (int i = tid; i < kASmallFloats / 4; i += 256) dst[i] = src[i];
  }
  Pipe p;
  pipe_init<Bf16ArtNet>(p, args.packed, smem, wave, lane);

  for (int pass = blockIdx.x; pass < args.npass; pass += gridDim.x) {
    const int64_t g = (int64_t)pass * 128 + wave * 32 + m;
    const bool valid = g < args.total;
    const int64_t gc = valid ? g : args.total - 1;
    const int64_t ray = gc / args.S;
    const float t = args.t_vals[gc];
    float x[3], vd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      x[a] = __fadd_rn(args.rays_o[ray * 3 + a], __fmul_rn(t, args.rays_d[ray * 3 + a]));  // helper.cast_rays
      vd[a] = args.viewdirs[ray * 3 + a];
    }
    f32x4* stash = reinterpret_cast<f32x4*>(smem + kBfRingBytes + kBaSmallBytes) + (wave * 2 * 64 + lane) * 4;
    auto load_enc = [&](int tile) {
      f32x16 e;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 v4 = stash[tile * 64 * 4 + k];
        e[4 * k] = v4[0]; e[4 * k + 1] = v4[1]; e[4 * k + 2] = v4[2]; e[4 * k + 3] = v4[3];
      }
      return e;
    };
    LimbFrag cur[2], nxt[2];
    PlaneIO io{};
    if constexpr (TRAIN) io = make_plane_io(args.Np, g, h);
    auto rows = [&](int row) { return reinterpret_cast<float*>(reinterpret_cast<char*>(args.planes) + (int64_t)row * io.row_bytes); };
    const int64_t tile_bytes = 32 * io.row_bytes;
    auto mask = [&](auto& tiles, int slot) {
      if constexpr (TRAIN) args.masks[(int64_t)slot * args.Np * 2 + (int64_t)pass * 256 + tid] = relu_mask_bits(tiles);
    };
    auto save_row = [&](int row, float v) {  // one scalar per sample (lanes 0..31)
      if constexpr (TRAIN) { if (h == 0) *reinterpret_cast<float*>(reinterpret_cast<char*>(args.planes) + (int64_t)row * io.row_bytes + g * 4) = v; }
    };
    if constexpr (TRAIN) {
#pragma unroll
      for (int a = 0; a < 3; ++a) save_row(kAPlPos + a, x[a]);
    }

    // ---- deformation MLP (:196-205): layer 0 (3 -> 128, effective bias) on the VALU, layers 1-3 on the matrix pipe ----
    // [TRAIN] every tile's ReLU'd fp32 values are stored by the split that consumes it; tiles consumed by a VALU head
    // (deformation layer 3, view layer 3) are stored directly; the ReLU decisions go out as bit masks after each layer.
    f32x16 H0[4], H1[4];   // PRE-activations; the ReLU is applied by the split / head that consumes them
    init_bias(H0, sm + kA_BD0, h);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(sm + kA_WD0 + a * 128 + 32 * tt + 8 * gq + 4 * h);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) H0[tt][4 * gq + cc] = __builtin_fmaf(w[cc], x[a], H0[tt][4 * gq + cc]);
        }
      }
    }
    mask(H0, 0);
    split_tile<1, TRAIN>(H0[0], cur, rows(aplane_d(0)), &io); init_bias(H1, sm + kA_BD + 0 * 128, h);
    layer4_bf16<kAChD1 + 0, TRAIN>(p, cur, H0, H1, rows(aplane_d(0)), &io, tile_bytes); mask(H1, 1);
    split_tile<1, TRAIN>(H1[0], cur, rows(aplane_d(1)), &io); init_bias(H0, sm + kA_BD + 1 * 128, h);
    layer4_bf16<kAChD1 + 4, TRAIN>(p, cur, H1, H0, rows(aplane_d(1)), &io, tile_bytes); mask(H0, 2);
    split_tile<1, TRAIN>(H0[0], cur, rows(aplane_d(2)), &io); init_bias(H1, sm + kA_BD + 2 * 128, h);
    layer4_bf16<kAChD1 + 8, TRAIN>(p, cur, H0, H1, rows(aplane_d(2)), &io, tile_bytes); mask(H1, 3);
    if constexpr (TRAIN) { relu_tiles(H1); store_plane(H1, rows(aplane_d(3)), io); }
    float xd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // x' = deformation_layer(relu(h)) + pos   (:205)
      float v = head_partial_relu<4>(H1, sm + kA_WDL + a * 128, h);
      v = v + __shfl_xor(v, 32) + sm[kA_BDL + a];
      xd[a] = __fadd_rn(v, x[a]);
    }
    {  // pos_enc of the deformed point (enc_after=True, :207-208), parked in LDS between trunk layers 0 and 5
      f32x16 E[2];
      encode_pos(xd, h, E);
      if constexpr (TRAIN) {
#pragma unroll
        for (int a = 0; a < 3; ++a) save_row(kAPlPos + 3 + a, xd[a]);
        store_pos_enc_plane(E, rows(kAPlE), io, g, h);
      }
#pragma unroll
      for (int tile = 0; tile < 2; ++tile)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f32x4 v4; v4[0] = E[tile][4 * k]; v4[1] = E[tile][4 * k + 1]; v4[2] = E[tile][4 * k + 2]; v4[3] = E[tile][4 * k + 3];
          stash[tile * 64 * 4 + k] = v4;
        }
    }

    // ---- trunk (:212-217), shape latent folded into the biases of layers 0 and 5 ----
    f32x16 X[8], Y[8];
    {
      const f32x16 e0 = load_enc(0), e1 = load_enc(1);
      split_tile<0>(e0, cur);
      init_bias(X, sm + kA_BT + 0 * 256, h);
      chunk_mma_bf16<kAChT0 + 0, 8, true, 0, false, Bf16ArtNet>(p, cur, X, e1, nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
      chunk_mma_bf16<kAChT0 + 1, 8, false, 0, false, Bf16ArtNet>(p, cur, X, e1, nxt);
    }
    mask(X, 4);
#define AON_BA_LAYER(IN, OUT, L_IN, BIAS, CB)                                                                          \
    split_tile<1, TRAIN>(IN[0], cur, rows(aplane_h(L_IN)), &io); init_bias(OUT, sm + (BIAS), h);                         \
    layer8_bf16<CB, 8, 1, false, 0, TRAIN, false, Bf16ArtNet>(p, cur, IN, OUT, IN[0], rows(aplane_h(L_IN)), &io, tile_bytes);
    AON_BA_LAYER(X, Y, 0, kA_BT + 1 * 256, kAChT1 + 0)  mask(Y, 5);
    AON_BA_LAYER(Y, X, 1, kA_BT + 2 * 256, kAChT1 + 8)  mask(X, 6);
    AON_BA_LAYER(X, Y, 2, kA_BT + 3 * 256, kAChT1 + 16) mask(Y, 7);
    AON_BA_LAYER(Y, X, 3, kA_BT + 4 * 256, kAChT1 + 24) mask(X, 8);
    // layer 5: cat[relu(h4) (8 tiles), enc (2 tiles)]
    split_tile<1, TRAIN>(X[0], cur, rows(aplane_h(4)), &io); init_bias(Y, sm + kA_BT + 5 * 256, h);
    {
      const f32x16 e0 = load_enc(0);
      layer8_bf16<kAChT5, 8, 1, true, 0, TRAIN, false, Bf16ArtNet>(p, cur, X, Y, e0, rows(aplane_h(4)), &io, tile_bytes);
      const f32x16 e1 = load_enc(1);
      chunk_mma_bf16<kAChT5 + 8, 8, true, 0, false, Bf16ArtNet>(p, cur, Y, e1, nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
      chunk_mma_bf16<kAChT5 + 9, 8, false, 0, false, Bf16ArtNet>(p, cur, Y, e1, nxt);
    }
    mask(Y, 9);
    AON_BA_LAYER(Y, X, 5, kA_BT + 6 * 256, kAChT6) mask(X, 10);
    AON_BA_LAYER(X, Y, 6, kA_BT + 7 * 256, kAChT7) mask(Y, 11);
    float sigma = head_partial_relu<8>(Y, sm + kA_WSIG, h);  // density_layer on relu(layer 7) (:219)
    sigma = sigma + __shfl_xor(sigma, 32) + sm[kA_BSIG];
    AON_BA_LAYER(Y, X, 7, kA_BBOT, kAChBott)                    // bottleneck, linear output (:223)
#undef AON_BA_LAYER

    // ---- view branch (:227-234): cat[bottleneck, viewenc, appearance (folded)] -> 4 x (128, ReLU) ----
    f32x16 Z0[4], Z1[4], V;
    encode_view(vd, h, V);
    if constexpr (TRAIN) store_view_enc_plane(V, rows(kAPlVE), io, g, h);
    split_tile<0, TRAIN>(X[0], cur, rows(kAPlBot), &io); init_bias(Z0, sm + kA_BV + 0 * 128, h);
    layer8_bf16<kAChV0, 4, 0, true, 0, TRAIN, false, Bf16ArtNet>(p, cur, X, Z0, V, rows(kAPlBot), &io, tile_bytes);
    chunk_mma_bf16<kAChV0 + 8, 4, false, 0, false, Bf16ArtNet>(p, cur, Z0, V, nxt);
    mask(Z0, 12);
    split_tile<1, TRAIN>(Z0[0], cur, rows(aplane_v(0)), &io); init_bias(Z1, sm + kA_BV + 1 * 128, h);
    layer4_bf16<kAChV1 + 0, TRAIN>(p, cur, Z0, Z1, rows(aplane_v(0)), &io, tile_bytes); mask(Z1, 13);
    split_tile<1, TRAIN>(Z1[0], cur, rows(aplane_v(1)), &io); init_bias(Z0, sm + kA_BV + 2 * 128, h);
    layer4_bf16<kAChV1 + 4, TRAIN>(p, cur, Z1, Z0, rows(aplane_v(1)), &io, tile_bytes); mask(Z0, 14);
    split_tile<1, TRAIN>(Z0[0], cur, rows(aplane_v(2)), &io); init_bias(Z1, sm + kA_BV + 3 * 128, h);
    layer4_bf16<kAChV1 + 8, TRAIN>(p, cur, Z0, Z1, rows(aplane_v(2)), &io, tile_bytes); mask(Z1, 15);
    if constexpr (TRAIN) { relu_tiles(Z1); store_plane(Z1, rows(aplane_v(3)), io); }
    float rgb[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {  // rgb_layer on relu(view layer 3) (:236)
      float v = head_partial_relu<4>(Z1, sm + kA_WRGB + ch * kCondWidth, h);
      rgb[ch] = v + __shfl_xor(v, 32) + sm[kA_BRGB + ch];
    }
    if (valid && h == 0) {
      f32x4 o; o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; o[3] = sigma;
      reinterpret_cast<f32x4*>(args.raw)[g] = o;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// backward data chain of the articulated network on the bf16x3 engine (fp32 twin: art_bwd_chain_kernel, aon_train_art.hip)
// ---------------------------------------------------------------------------------------------
// Same 108-chunk transposed order; chunk c = the 32 gradient features j of one tile x the F features f of the layer's input
// (F = 64 for the two pos-enc pull-backs, 128 or 256 otherwise), A operand = W^T in limb form:
//   [k16 step s][out tile tp][limb][lane][8 bf16], lane (i, h), k-slot jj of step s <-> W[j = 32T + (r&3)+8(r>>2)+4h][f], r = 8s+jj
// with f = 32tp + i, or, for the pos-enc tiles, the encoding column whose register/half is accumulator row i.
struct Bf16ArtBwdNet {
  static constexpr int kNumChunks = kABwNumChunks;
  static constexpr int kSlotBytes = 8 * 6144;
  static constexpr bool kPair = false;
  static constexpr int chunk_tiles(int c) {
    return (c < kABwV0 || c >= kABwD3) ? 4 : ((c >= kABwL5E && c < kABwL5) || (c >= kABwL0E && c < kABwD3)) ? 2 : 8;
  }
  static constexpr int chunk_bytes(int c) { return chunk_tiles(c) * 6144; }
};
__host__ __device__ constexpr int64_t babw_offset(int c) {
  int64_t off = 0;
  for (int i = 0; i < c; ++i) off += Bf16ArtBwdNet::chunk_bytes(i);
  return off;
}
constexpr int64_t kBaBwStreamBytes = babw_offset(kABwNumChunks);

__global__ void pack_art_bwd_bf16x3_kernel(ArtPackArgsB a, char* __restrict__ packed) {
  // one thread per 16-byte limb vector triple: (chunk, s, tp, lane)
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int c = 0;
  int64_t base = 0;  // in threads; a chunk has 2 * tiles * 64 threads
  while (c < kABwNumChunks && idx >= base + 2 * Bf16ArtBwdNet::chunk_tiles(c) * 64) { base += 2 * Bf16ArtBwdNet::chunk_tiles(c) * 64; ++c; }
  if (c >= kABwNumChunks) return;
  const int r = (int)(idx - base), nt = Bf16ArtBwdNet::chunk_tiles(c);
  const int lane = r & 63, tp = (r >> 6) % nt, s = (r >> 6) / nt;
  const int h = lane >> 5, i = lane & 31;
  int col = 32 * tp + i;  // forward-input feature (row of W^T)
  auto enc_col = [&]() {  // accumulator row i of tile tp -> encoding register rho of half h'
    const int rr = (i & 3) + 4 * (i >> 3), hh = (i >> 2) & 1;
    return posenc_col(tp, rr >> 2, rr & 3, hh);
  };
  const float* W; int ld, T;
  if (c < kABwV0) { const int l = 3 - c / 4; W = a.p[26 + 2 * l]; ld = 128; T = c % 4; }
  else if (c < kABwBott) { W = a.p[26]; ld = 411; T = c - kABwV0; }
  else if (c < kABwL7) { W = a.p[34]; ld = 256; T = c - kABwBott; }
  else if (c < kABwL5E) { const int l = 7 - (c - kABwL7) / 8; W = a.p[10 + 2 * l]; ld = 256; T = (c - kABwL7) % 8; }
  else if (c < kABwL5) { W = a.p[20]; ld = 447; T = c - kABwL5E; col = enc_col(); if (col >= 0) col += 256; }
  else if (c < kABwL0E) { const int l = 5 - (c - kABwL5) / 8; W = a.p[10 + 2 * l]; ld = l == 5 ? 447 : 256; T = (c - kABwL5) % 8; }
  else if (c < kABwD3) { W = a.p[10]; ld = 191; T = c - kABwL0E; col = enc_col(); }
  else { const int l = 3 - (c - kABwD3) / 4; W = a.p[2 * l]; ld = 128; T = (c - kABwD3) % 4; }
  unsigned short hi[8], mid[8], lo[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int reg = 8 * s + jj;
    const int j = 32 * T + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    const float w = col >= 0 ? W[(int64_t)j * ld + col] : 0.f;
    hi[jj] = bf16_rne_bits(w);
    const float r1 = w - bf16_bits_to_f32(hi[jj]);
    mid[jj] = bf16_rne_bits(r1);
    lo[jj] = bf16_rne_bits(r1 - bf16_bits_to_f32(mid[jj]));
  }
  char* dst = packed + babw_offset(c) + ((int64_t)(s * nt + tp) * 3) * 1024 + lane * 16;
  auto put = [&](int limb, const unsigned short (&v)[8]) {
    u32x4 o;
    o[0] = v[0] | ((unsigned)v[1] << 16); o[1] = v[2] | ((unsigned)v[3] << 16);
    o[2] = v[4] | ((unsigned)v[5] << 16); o[3] = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<u32x4*>(dst + limb * 1024) = o;
  };
  put(0, hi); put(1, mid); put(2, lo);
}

// out += W^T-chunks . in  over NT_IN gradient tiles (already masked fp32), out = NT_OUT tiles
template <int CBASE, int NT_IN, int NT_OUT>
__device__ __forceinline__ void bwd_layer_bf16(Pipe& p, const f32x16 (&in)[NT_IN], f32x16 (&out)[NT_OUT]) {
  LimbFrag cur[2], nxt[2];
  split_tile<0>(in[0], cur);
  chunk_mma_bf16<CBASE + 0, NT_OUT, true, 0, false, Bf16ArtBwdNet>(p, cur, out, in[1], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
  chunk_mma_bf16<CBASE + 1, NT_OUT, true, 0, false, Bf16ArtBwdNet>(p, cur, out, in[2], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
  chunk_mma_bf16<CBASE + 2, NT_OUT, true, 0, false, Bf16ArtBwdNet>(p, cur, out, in[3], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
  if constexpr (NT_IN == 4) {
    chunk_mma_bf16<CBASE + 3, NT_OUT, false, 0, false, Bf16ArtBwdNet>(p, cur, out, in[3], nxt);
  } else {
    chunk_mma_bf16<CBASE + 3, NT_OUT, true, 0, false, Bf16ArtBwdNet>(p, cur, out, in[4], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
    chunk_mma_bf16<CBASE + 4, NT_OUT, true, 0, false, Bf16ArtBwdNet>(p, cur, out, in[5], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
    chunk_mma_bf16<CBASE + 5, NT_OUT, true, 0, false, Bf16ArtBwdNet>(p, cur, out, in[6], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
    chunk_mma_bf16<CBASE + 6, NT_OUT, true, 0, false, Bf16ArtBwdNet>(p, cur, out, in[7], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
    chunk_mma_bf16<CBASE + 7, NT_OUT, false, 0, false, Bf16ArtBwdNet>(p, cur, out, in[7], nxt);
  }
}

struct ArtBfBwdArgs {
  const char* packed_bwd;   // kBaBwStreamBytes
  const float* small;       // per-call small block (head weights)
  const float* d_raw;       // (Np,4)
  const u32x4* masks;       // kAMaskLayers x (Np*2)
  const float* planes;      // forward planes (deformed position rows 3..5 are read)
  float* dplanes;
  float* dxp;               // (Np,4)
  int64_t Np; int npass;
};

template <int NT>
__device__ __forceinline__ void zero_tiles_ba(f32x16 (&x)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = 0.f;
}
template <int NT>
__device__ __forceinline__ void mask_store_ba(f32x16 (&x)[NT], const u32x4 bits, float* dplane, const PlaneIO& io) {
  apply_mask_bits(x, bits);
  store_plane(x, dplane, io);
}

__global__ void __launch_bounds__(256) art_bwd_chain_bf16x3_kernel(ArtBfBwdArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem + kBfRingBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(args.small);
    f32x4* dst = reinterpret_cast<f32x4*>(sm);
    for (int i = tid; i < kASmallFloats / 4; i += 256) dst[i] = src[i];
  }
  Pipe p;
  pipe_init<Bf16ArtBwdNet>(p, args.packed_bwd, smem, wave, lane);

  for (int pass = blockIdx.x; pass < args.npass; pass += gridDim.x) {
    const int64_t col = (int64_t)pass * 128 + wave * 32 + m;
    const PlaneIO io = make_plane_io(args.Np, col, h);
    auto dp = [&](int row) { return reinterpret_cast<float*>(reinterpret_cast<char*>(args.dplanes) + (int64_t)row * io.row_bytes); };
    u32x4 mk[kAMaskLayers];
#pragma unroll
    for (int l = 0; l < kAMaskLayers; ++l) mk[l] = args.masks[(int64_t)l * args.Np * 2 + (int64_t)pass * 256 + tid];
    const float4 dr = reinterpret_cast<const float4*>(args.d_raw)[col];
    float xd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
      xd[a] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(args.planes) + (int64_t)(kAPlPos + 3 + a) * io.row_bytes + col * 4);

    // ---- view branch, backwards (model_autodecoder.py:231-236) ----
    f32x16 Z0[4], Z1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int fo = 32 * t + 8 * gq + 4 * h;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + kA_WRGB + 0 * kCondWidth + fo);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + kA_WRGB + 1 * kCondWidth + fo);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(sm + kA_WRGB + 2 * kCondWidth + fo);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          Z1[t][4 * gq + cc] = __builtin_fmaf(w2[cc], dr.z, __builtin_fmaf(w1[cc], dr.y, w0[cc] * dr.x));
      }
    }
    mask_store_ba(Z1, mk[15], dp(aplane_v(3)), io);
    zero_tiles_ba(Z0); bwd_layer_bf16<kABwV3 + 0, 4, 4>(p, Z1, Z0); mask_store_ba(Z0, mk[14], dp(aplane_v(2)), io);
    zero_tiles_ba(Z1); bwd_layer_bf16<kABwV3 + 4, 4, 4>(p, Z0, Z1); mask_store_ba(Z1, mk[13], dp(aplane_v(1)), io);
    zero_tiles_ba(Z0); bwd_layer_bf16<kABwV3 + 8, 4, 4>(p, Z1, Z0); mask_store_ba(Z0, mk[12], dp(aplane_v(0)), io);
    f32x16 X[8], Y[8];
    zero_tiles_ba(X);
    bwd_layer_bf16<kABwV0, 4, 8>(p, Z0, X);
    store_plane(X, dp(kAPlBot), io);  // bottleneck: no activation
    // ---- trunk ----
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sm + kA_WSIG + 32 * t + 8 * gq + 4 * h);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) Y[t][4 * gq + cc] = w[cc] * dr.w;
      }
    }
    bwd_layer_bf16<kABwBott, 8, 8>(p, X, Y); mask_store_ba(Y, mk[11], dp(aplane_h(7)), io);
    zero_tiles_ba(X); bwd_layer_bf16<kABwL7 + 0, 8, 8>(p, Y, X); mask_store_ba(X, mk[10], dp(aplane_h(6)), io);
    zero_tiles_ba(Y); bwd_layer_bf16<kABwL7 + 8, 8, 8>(p, X, Y); mask_store_ba(Y, mk[9], dp(aplane_h(5)), io);
    f32x16 dE[2];
    zero_tiles_ba(dE);
    bwd_layer_bf16<kABwL5E, 8, 2>(p, Y, dE);  // skip connection: d enc += W5[:, 256:319]^T dZ5
    zero_tiles_ba(X); bwd_layer_bf16<kABwL5 + 0, 8, 8>(p, Y, X); mask_store_ba(X, mk[8], dp(aplane_h(4)), io);
    zero_tiles_ba(Y); bwd_layer_bf16<kABwL5 + 8, 8, 8>(p, X, Y); mask_store_ba(Y, mk[7], dp(aplane_h(3)), io);
    zero_tiles_ba(X); bwd_layer_bf16<kABwL5 + 16, 8, 8>(p, Y, X); mask_store_ba(X, mk[6], dp(aplane_h(2)), io);
    zero_tiles_ba(Y); bwd_layer_bf16<kABwL5 + 24, 8, 8>(p, X, Y); mask_store_ba(Y, mk[5], dp(aplane_h(1)), io);
    zero_tiles_ba(X); bwd_layer_bf16<kABwL5 + 32, 8, 8>(p, Y, X); mask_store_ba(X, mk[4], dp(aplane_h(0)), io);
    bwd_layer_bf16<kABwL0E, 8, 2>(p, X, dE);  // d enc += W0[:, :63]^T dZ0

    // ---- positional encoding, backwards (helper.py:136-140 on the deformed point) ----
    const float phase = h ? AON_HALF_PI_F32 : 0.f;
    float dx[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int rho = 0; rho < 30; ++rho) {
      const float scale = (float)(1 << (rho / 3));
      const float arg = __fadd_rn(__fmul_rn(xd[rho % 3], scale), phase);
      const float c = cos_f32(arg);
      dx[rho % 3] = __builtin_fmaf(scale * c, dE[rho >> 4][rho & 15], dx[rho % 3]);
    }
    if (h) dx[2] += dE[1][14]; else { dx[0] += dE[1][14]; dx[1] += dE[1][15]; }
#pragma unroll
    for (int a = 0; a < 3; ++a) dx[a] = dx[a] + __shfl_xor(dx[a], 32);
    if (h == 0) {
      float4 o; o.x = dx[0]; o.y = dx[1]; o.z = dx[2]; o.w = 0.f;
      reinterpret_cast<float4*>(args.dxp)[col] = o;
    }

    // ---- deformation MLP, backwards (x' = deformation_layer(h3) + pos, :200-205) ----
    f32x16 (&H1)[4] = Z1;
    f32x16 (&H0)[4] = Z0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int fo = 32 * t + 8 * gq + 4 * h;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + kA_WDL + 0 * 128 + fo);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + kA_WDL + 1 * 128 + fo);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(sm + kA_WDL + 2 * 128 + fo);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          H1[t][4 * gq + cc] = __builtin_fmaf(w2[cc], dx[2], __builtin_fmaf(w1[cc], dx[1], w0[cc] * dx[0]));
      }
    }
    mask_store_ba(H1, mk[3], dp(aplane_d(3)), io);
    zero_tiles_ba(H0); bwd_layer_bf16<kABwD3 + 0, 4, 4>(p, H1, H0); mask_store_ba(H0, mk[2], dp(aplane_d(2)), io);
    zero_tiles_ba(H1); bwd_layer_bf16<kABwD3 + 4, 4, 4>(p, H0, H1); mask_store_ba(H1, mk[1], dp(aplane_d(1)), io);
    zero_tiles_ba(H0); bwd_layer_bf16<kABwD3 + 8, 4, 4>(p, H1, H0); mask_store_ba(H0, mk[0], dp(aplane_d(0)), io);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int num_cus();

int64_t art_bf16x3_packed_bytes() { return kBaStreamBytes; }

hipError_t launch_pack_art_bf16x3(const float* const* params, char* packed, hipStream_t stream) {
  ArtPackArgsB a;
  for (int i = 0; i < kNumArtParams; ++i) a.p[i] = params[i];
  const int64_t n = ((int64_t)kAChT0 * 4 + (int64_t)(kAChV0 - kAChT0) * 8 + (int64_t)(kANumChunks - kAChV0) * 4) * 2 * 64;
  pack_art_bf16x3_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed);
  return hipGetLastError();
}

template <bool TRAIN>
static hipError_t launch_art_bf16x3_t(ArtBfArgs a, hipStream_t stream) {
  static DeviceOnce lds_once;
  if (hipError_t e = set_max_lds(&art_mlp_fwd_bf16x3_kernel<TRAIN>, kBaLdsBytes, lds_once); e != hipSuccess) return e;
  const int cus = num_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  const int grid = a.npass < cus ? a.npass : cus;
  if (grid <= 0) return hipSuccess;
  art_mlp_fwd_bf16x3_kernel<TRAIN><<<dim3(grid), dim3(256), kBaLdsBytes, stream>>>(a);
  return hipGetLastError();
}

hipError_t launch_art_mlp_fwd_bf16x3(const char* packed, const float* small, const float* rays_o, const float* rays_d,
                                     const float* viewdirs, const float* t_vals, int64_t n_rays, int S, float* raw, hipStream_t stream) {
  ArtBfArgs a{packed, small, rays_o, rays_d, viewdirs, t_vals, raw, n_rays * S, S, (int)((n_rays * S + 127) / 128), nullptr, nullptr, 0};
  return launch_art_bf16x3_t<false>(a, stream);
}

// training forward of the bf16x3 engine: planes / masks contract of launch_art_mlp_fwd_train (aon_mlp_art.hip)
hipError_t launch_art_mlp_fwd_train_bf16x3(const char* packed, const float* small, const float* rays_o, const float* rays_d,
                                           const float* viewdirs, const float* t_vals, int64_t n_rays, int S, float* raw, float* planes,
                                           void* masks, hipStream_t stream) {
  ArtBfArgs a{packed, small, rays_o, rays_d, viewdirs, t_vals, raw, n_rays * S, S, (int)((n_rays * S + 127) / 128), planes,
              static_cast<u32x4*>(masks), 0};
  a.Np = (int64_t)a.npass * 128;
  return launch_art_bf16x3_t<true>(a, stream);
}

int64_t art_bwd_bf16x3_packed_bytes() { return kBaBwStreamBytes; }

hipError_t launch_pack_art_bwd_bf16x3(const float* const* params, char* packed, hipStream_t stream) {
  ArtPackArgsB a;
  for (int i = 0; i < kNumArtParams; ++i) a.p[i] = params[i];
  const int64_t n = kBaBwStreamBytes / 6144 * 2 * 64;  // (tiles in the stream) x 2 k16 steps x 64 lanes
  pack_art_bwd_bf16x3_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed);
  return hipGetLastError();
}

hipError_t launch_art_bwd_chain_bf16x3(const char* packed_bwd, const float* small, const float* d_raw, const void* masks,
                                       const float* planes, float* dplanes, float* dxp, int64_t Np, hipStream_t stream) {
  static DeviceOnce lds_once;
  constexpr int lds = kBfRingBytes + kBaSmallBytes;
  if (hipError_t e = set_max_lds(&art_bwd_chain_bf16x3_kernel, lds, lds_once); e != hipSuccess) return e;
  ArtBfBwdArgs a{packed_bwd, small, d_raw, static_cast<const u32x4*>(masks), planes, dplanes, dxp, Np, (int)(Np / 128)};
  const int cus = num_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  const int grid = a.npass < cus ? a.npass : cus;
  if (grid <= 0) return hipSuccess;
  art_bwd_chain_bf16x3_kernel<<<dim3(grid), dim3(256), lds, stream>>>(a);
  return hipGetLastError();
}

}  // namespace aon
