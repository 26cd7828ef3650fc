// Opt-in "bf16x3" engine of the fused vanilla NeRFMLP forward: fp32-equivalent arithmetic on the bf16 matrix pipe.
//
// CDNA4's fp32 MFMA runs at the fp32 VECTOR rate (157 TFLOP/s), 1/16 of the bf16 MFMA rate.  Here every fp32 operand
// (weight or activation) is split EXACTLY into three bf16 limbs  x = hi + mid + lo  (hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid); the residual is below 2^-26 |x| because each limb carries its own exponent), and a product
// a*b is evaluated as the six limb products whose magnitude is >= 2^-18 |ab|:
//      hi*hi + (hi*mid + mid*hi) + (mid*mid + hi*lo + lo*hi)            (dropped: mid*lo, lo*mid, lo*lo <= 2^-25 |ab|)
// Each limb product is exact in fp32 (8 x 8 significant bits) and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the
// result carries fp32-class error (the parity suite runs unchanged at the fp32 kernel's tolerances with this engine
// selected).  Six bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 16-deep k step: 2.67x less matrix
// time.  The exact-fp32 kernel (aon_mlp.hip) stays the default; this engine is selected explicitly.
//
// Structure = aon_mlp.hip (transposed layers, accumulator layout == next layer's B layout, register-resident activations,
// LDS-DMA weight chunks), with two differences: the previous layer's fp32 accumulator tile is (ReLU'd and) split into limb
// fragments just before the chunk that consumes it, and a chunk carries three limb images of the weights (48 KiB).
#include "aon_bf16_core.h"

namespace aon {

// ---------------------------------------------------------------------------------------------
// packing: three limb images per chunk
// ---------------------------------------------------------------------------------------------
struct PackArgsB {
  const float* p[kNumVanillaParams];
};

__global__ void pack_vanilla_bf16x3_kernel(PackArgsB a, char* __restrict__ packed) {
  // one thread per (chunk, s, tp, lane): 8 weights -> 3 x 16 bytes
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int64_t big_threads = (int64_t)kNumBigChunks * 2 * 8 * 64;
  constexpr int64_t all_threads = big_threads + (int64_t)(kNumChunks - kNumBigChunks) * 2 * 4 * 64;
  if (idx >= all_threads) {
    // resident small vectors (fp32, identical to the fp32 engine's block)
    const int64_t s = idx - all_threads;
    if (s >= kSmallFloats) return;
    float v = 0.f;
    if (s < kSmBiasBott) v = a.p[2 * (s >> 8) + 1][s & 255];
    else if (s < kSmBiasView) v = a.p[19][s - kSmBiasBott];
    else if (s < kSmWSigma) v = a.p[17][s - kSmBiasView];
    else if (s < kSmWRgb) v = a.p[20][s - kSmWSigma];
    else if (s < kSmBSigma) v = a.p[22][s - kSmWRgb];
    else if (s < kSmBRgb) v = a.p[21][0];
    else if (s < kSmBRgb + 3) v = a.p[23][s - kSmBRgb];
    reinterpret_cast<float*>(packed + kBfStreamBytes)[s] = v;
    return;
  }
  int c, r, nt;
  int64_t chunk_base;
  if (idx < big_threads) { c = (int)(idx / (2 * 8 * 64)); r = (int)(idx % (2 * 8 * 64)); nt = 8; chunk_base = (int64_t)c * 8 * 6144; }
  else {
    const int64_t i2 = idx - big_threads;
    c = kNumBigChunks + (int)(i2 / (2 * 4 * 64)); r = (int)(i2 % (2 * 4 * 64)); nt = 4;
    chunk_base = (int64_t)kNumBigChunks * 8 * 6144 + (int64_t)(c - kNumBigChunks) * 4 * 6144;
  }
  const int lane = r & 63, tp = (r >> 6) % nt, s = (r >> 6) / nt;
  const int h = lane >> 5, row = 32 * tp + (lane & 31);
  const float* W; int ld, n_out = 256;
  int kind, tile;  // kind 0: hidden columns 32*tile + feature, 1: pos-enc (+off), 2: view-enc (+off)
  int off = 0;
  if (c < kChL1) { W = a.p[0]; ld = kPosEnc; kind = 1; tile = c; }
  else if (c < kChL5) { const int l = 1 + (c - kChL1) / 8; W = a.p[2 * l]; ld = 256; kind = 0; tile = (c - kChL1) % 8; }
  else if (c < kChL5 + 8) { W = a.p[10]; ld = 256 + kPosEnc; kind = 0; tile = c - kChL5; }
  else if (c < kChL6) { W = a.p[10]; ld = 256 + kPosEnc; kind = 1; tile = c - kChL5 - 8; off = 256; }
  else if (c < kChL7) { W = a.p[12]; ld = 256; kind = 0; tile = c - kChL6; }
  else if (c < kChBott) { W = a.p[14]; ld = 256; kind = 0; tile = c - kChL7; }
  else if (c < kChView) { W = a.p[18]; ld = 256; kind = 0; tile = c - kChBott; }
  else if (c < kChView + 8) { W = a.p[16]; ld = 256 + kViewEnc; n_out = kCondWidth; kind = 0; tile = c - kChView; }
  else { W = a.p[16]; ld = 256 + kViewEnc; n_out = kCondWidth; kind = 2; tile = 0; off = 256; }
  unsigned short hi[8], mid[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int reg = 8 * s + j;  // accumulator register of the producing tile that supplies k-slot j of step s
    int col;
    if (kind == 0) col = 32 * tile + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    else if (kind == 1) { col = posenc_col(tile, reg >> 2, reg & 3, h); if (col >= 0) col += off; }
    else { col = viewenc_col(reg >> 2, reg & 3, h); if (col >= 0) col += off; }
    const float w = (col >= 0 && row < n_out) ? W[(int64_t)row * ld + col] : 0.f;
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

struct BfArgs {
  const char* packed;
  const float* rays_o; const float* rays_d; const float* viewdirs; const float* t_vals;
  float* raw;
  int64_t total; int S; int npass;
  float* planes;   // [TRAIN] kPlRows x Np feature-major activation planes (layout of aon_mlp.hip's training forward)
  u32x4* masks;    // [TRAIN] kMaskLayers x (Np*2) ReLU bit masks
  int64_t Np;
};

template <bool TRAIN>
__global__ void __launch_bounds__(256) mlp_fwd_bf16x3_kernel(BfArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem + kBfRingBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA bases stay scalar
  const int m = lane & 31, h = lane >> 5;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(args.packed + kBfStreamBytes);
    f32x4* dst = reinterpret_cast<f32x4*>(sm);
    for (int i = tid; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
  }
  Pipe p;
  pipe_init<Bf16Net>(p, args.packed, smem, wave, lane);

  for (int pass = blockIdx.x; pass < args.npass; pass += gridDim.x) {
    const int64_t g = (int64_t)pass * 128 + wave * 32 + m;
    const bool valid = g < args.total;
    const int64_t gc = valid ? g : args.total - 1;
    const int64_t ray = gc / args.S;
    // The 63-wide encoding is needed at layer 0 and again at the skip layer 5: it is parked in LDS (32 floats per lane)
    // instead of occupying 32 VGPRs through layers 1-4; the view encoding is computed only when the view layer needs it.
    float vd[3];
    PlaneIO io{};
    if constexpr (TRAIN) io = make_plane_io(args.Np, g, h);
    auto rows = [&](int row) { return reinterpret_cast<float*>(reinterpret_cast<char*>(args.planes) + (int64_t)row * io.row_bytes); };
    const int64_t tile_bytes = 32 * io.row_bytes;
    auto mask = [&](auto& tiles, int layer) {
      if constexpr (TRAIN) args.masks[(int64_t)layer * args.Np * 2 + (int64_t)pass * 256 + tid] = relu_mask_bits(tiles);
    };
    f32x4* stash = reinterpret_cast<f32x4*>(smem + kBfRingBytes + ((kSmallBytes + 15) / 16) * 16) + (wave * 2 * 64 + lane) * 4;
    auto load_enc = [&](int tile) {
      f32x16 e;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 v4 = stash[tile * 64 * 4 + k];
        e[4 * k] = v4[0]; e[4 * k + 1] = v4[1]; e[4 * k + 2] = v4[2]; e[4 * k + 3] = v4[3];
      }
      return e;
    };
    {
      const float t = args.t_vals[gc];
      float x[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        x[a] = __fadd_rn(args.rays_o[ray * 3 + a], __fmul_rn(t, args.rays_d[ray * 3 + a]));
        vd[a] = args.viewdirs[ray * 3 + a];
      }
      f32x16 E[2];
      encode_pos(x, h, E);
      if constexpr (TRAIN) store_pos_enc_plane(E, rows(kPlE), io, g, h);
#pragma unroll
      for (int tile = 0; tile < 2; ++tile)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f32x4 v4; v4[0] = E[tile][4 * k]; v4[1] = E[tile][4 * k + 1]; v4[2] = E[tile][4 * k + 2]; v4[3] = E[tile][4 * k + 3];
          stash[tile * 64 * 4 + k] = v4;
        }
    }
    // X / Y hold PRE-activation outputs; the ReLU is applied when a tile is split into limbs for the next layer.
    // `cur` always holds the limb fragments of the tile the next chunk consumes; each chunk pre-splits its successor's
    // tile between its own MFMA groups.  A layer's first input tile depends on the previous layer's LAST chunk, so it
    // is split at the layer boundary (the only exposed VALU work besides the bias initialisation).
    // [TRAIN] the (ReLU'd) fp32 value of every tile is stored to its plane rows by the split that consumes it; the ReLU
    // decisions go out as bit masks (sign of the pre-activation) right after each layer.
    f32x16 X[8], Y[8];
    LimbFrag cur[2], nxt[2];
    {
      const f32x16 e0 = load_enc(0), e1 = load_enc(1);
      split_tile<false>(e0, cur);
      init_bias(X, sm + kSmBias + 0 * 256, h);
      chunk_mma_bf16<kChL0 + 0, 8, true, false>(p, cur, X, e1, nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
      chunk_mma_bf16<kChL0 + 1, 8, false, false>(p, cur, X, e1, nxt);
    }
    mask(X, 0);
    split_tile<true, TRAIN>(X[0], cur, rows(plane_h(0)), &io); init_bias(Y, sm + kSmBias + 1 * 256, h);
    layer8_bf16<kChL1 + 0, 8, true, false, false, TRAIN>(p, cur, X, Y, X[0], rows(plane_h(0)), &io, tile_bytes); mask(Y, 1);
    split_tile<true, TRAIN>(Y[0], cur, rows(plane_h(1)), &io); init_bias(X, sm + kSmBias + 2 * 256, h);
    layer8_bf16<kChL1 + 8, 8, true, false, false, TRAIN>(p, cur, Y, X, Y[0], rows(plane_h(1)), &io, tile_bytes); mask(X, 2);
    split_tile<true, TRAIN>(X[0], cur, rows(plane_h(2)), &io); init_bias(Y, sm + kSmBias + 3 * 256, h);
    layer8_bf16<kChL1 + 16, 8, true, false, false, TRAIN>(p, cur, X, Y, X[0], rows(plane_h(2)), &io, tile_bytes); mask(Y, 3);
    split_tile<true, TRAIN>(Y[0], cur, rows(plane_h(3)), &io); init_bias(X, sm + kSmBias + 4 * 256, h);
    layer8_bf16<kChL1 + 24, 8, true, false, false, TRAIN>(p, cur, Y, X, Y[0], rows(plane_h(3)), &io, tile_bytes); mask(X, 4);
    // L5: cat[relu(h4) (8 tiles), enc (2 tiles)]
    split_tile<true, TRAIN>(X[0], cur, rows(plane_h(4)), &io); init_bias(Y, sm + kSmBias + 5 * 256, h);
    {
      const f32x16 e0 = load_enc(0);
      layer8_bf16<kChL5, 8, true, true, false, TRAIN, false>(p, cur, X, Y, e0, rows(plane_h(4)), &io, tile_bytes);
      const f32x16 e1 = load_enc(1);
      chunk_mma_bf16<kChL5 + 8, 8, true, false>(p, cur, Y, e1, nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
      chunk_mma_bf16<kChL5 + 9, 8, false, false>(p, cur, Y, e1, nxt);
    }
    mask(Y, 5);
    split_tile<true, TRAIN>(Y[0], cur, rows(plane_h(5)), &io); init_bias(X, sm + kSmBias + 6 * 256, h);
    layer8_bf16<kChL6, 8, true, false, false, TRAIN>(p, cur, Y, X, Y[0], rows(plane_h(5)), &io, tile_bytes); mask(X, 6);
    split_tile<true, TRAIN>(X[0], cur, rows(plane_h(6)), &io); init_bias(Y, sm + kSmBias + 7 * 256, h);
    layer8_bf16<kChL7, 8, true, false, false, TRAIN>(p, cur, X, Y, X[0], rows(plane_h(6)), &io, tile_bytes); mask(Y, 7);
    float sigma = head_partial_relu<8>(Y, sm + kSmWSigma, h);  // density head on relu(layer 7)
    sigma = sigma + __shfl_xor(sigma, 32) + sm[kSmBSigma];
    // bottleneck: input relu(h7), output linear
    split_tile<true, TRAIN>(Y[0], cur, rows(plane_h(7)), &io); init_bias(X, sm + kSmBiasBott, h);
    layer8_bf16<kChBott, 8, true, false, false, TRAIN>(p, cur, Y, X, Y[0], rows(plane_h(7)), &io, tile_bytes);
    // view layer: cat[bottleneck (8 tiles, no activation), viewenc (1 tile)]
    f32x16 Z[4], V;
    encode_view(vd, h, V);
    if constexpr (TRAIN) store_view_enc_plane(V, rows(kPlVE), io, g, h);
    split_tile<false, TRAIN>(X[0], cur, rows(kPlBot), &io); init_bias(Z, sm + kSmBiasView, h);
    layer8_bf16<kChView, 4, false, true, false, TRAIN, false>(p, cur, X, Z, V, rows(kPlBot), &io, tile_bytes);
    chunk_mma_bf16<kChView + 8, 4, false, false>(p, cur, Z, V, nxt);
    if constexpr (TRAIN) {
      mask(Z, 8);
      relu_tiles(Z);
      store_plane(Z, rows(kPlHV), io);
    }
    float rgb[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float v = head_partial_relu<4>(Z, sm + kSmWRgb + ch * kCondWidth, h);
      rgb[ch] = v + __shfl_xor(v, 32) + sm[kSmBRgb + ch];
    }
    if (valid && h == 0) {
      f32x4 o; o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; o[3] = sigma;
      reinterpret_cast<f32x4*>(args.raw)[g] = o;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// backward data chain on the bf16x3 engine (fp32 twin: mlp_bwd_chain_kernel, aon_train.hip)
// ---------------------------------------------------------------------------------------------
// Transposed weight stream in limb form: 68 chunks of 48 KiB; chunk c = the 32 gradient features j of one tile x all 256
// features f of the layer's input, A operand = W^T: [k16 step s][out tile tp][limb][lane][8 bf16] with
//   lane (i = lane&31, h = lane>>5), k-slot jj of step s  <->  W[j = 32T + (r&3) + 8(r>>2) + 4h][f = 32tp + i],  r = 8s + jj
// (the accumulator register order of the producing tile, exactly as in the forward stream).
constexpr int kBbView = 0;    // views_linear.0 (cols 0..255): 4 chunks
constexpr int kBbBott = 4;    // bottleneck_layer: 8
constexpr int kBbL7 = 12;     // pts_linears.7 ... pts_linears.1: 8 each, backward order
constexpr int kBbNumChunks = 68;

struct Bf16BwdNet {
  static constexpr int kNumChunks = kBbNumChunks;
  static constexpr int kSlotBytes = 8 * 6144;
  static constexpr bool kPair = false;
  static constexpr int chunk_bytes(int) { return 8 * 6144; }
};
constexpr int64_t kBbStreamBytes = (int64_t)kBbNumChunks * 8 * 6144;

__global__ void pack_vanilla_bwd_bf16x3_kernel(PackArgsB a, char* __restrict__ packed) {
  // one thread per (chunk, s, tp, lane): 8 transposed weights -> 3 x 16 bytes
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)kBbNumChunks * 2 * 8 * 64) return;
  const int c = (int)(idx / (2 * 8 * 64)), r = (int)(idx % (2 * 8 * 64));
  const int lane = r & 63, tp = (r >> 6) & 7, s = r >> 9;
  const int h = lane >> 5, f = 32 * tp + (lane & 31);
  const float* W; int ld, T;
  if (c < kBbBott) { W = a.p[16]; ld = 256 + kViewEnc; T = c; }
  else if (c < kBbL7) { W = a.p[18]; ld = 256; T = c - kBbBott; }
  else {
    const int l = 7 - (c - kBbL7) / 8;  // 7,6,5,4,3,2,1
    W = a.p[2 * l]; ld = l == 5 ? 256 + kPosEnc : 256; T = (c - kBbL7) % 8;
  }
  unsigned short hi[8], mid[8], lo[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int reg = 8 * s + jj;
    const int j = 32 * T + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    const float w = W[(int64_t)j * ld + f];
    hi[jj] = bf16_rne_bits(w);
    const float r1 = w - bf16_bits_to_f32(hi[jj]);
    mid[jj] = bf16_rne_bits(r1);
    lo[jj] = bf16_rne_bits(r1 - bf16_bits_to_f32(mid[jj]));
  }
  char* dst = packed + (int64_t)c * 8 * 6144 + ((int64_t)(s * 8 + tp) * 3) * 1024 + lane * 16;
  auto put = [&](int limb, const unsigned short (&v)[8]) {
    u32x4 o;
    o[0] = v[0] | ((unsigned)v[1] << 16); o[1] = v[2] | ((unsigned)v[3] << 16);
    o[2] = v[4] | ((unsigned)v[5] << 16); o[3] = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<u32x4*>(dst + limb * 1024) = o;
  };
  put(0, hi); put(1, mid); put(2, lo);
}

struct BfBwdArgs {
  const char* packed_bwd;   // kBbStreamBytes
  const float* small;       // fp32 small block of the forward stream (head weights)
  const float* d_raw;       // (Np,4)
  const u32x4* masks;       // forward ReLU bit masks, kMaskLayers x (Np*2)
  float* dplanes;           // pre-activation gradient planes
  int64_t Np; int npass;
};

template <int NT>
__device__ __forceinline__ void zero_tiles_bf(f32x16 (&x)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = 0.f;
}

__global__ void __launch_bounds__(256) mlp_bwd_chain_bf16x3_kernel(BfBwdArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem + kBfRingBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(args.small);
    f32x4* dst = reinterpret_cast<f32x4*>(sm);
    for (int i = tid; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
  }
  Pipe p;
  pipe_init<Bf16BwdNet>(p, args.packed_bwd, smem, wave, lane);

  for (int pass = blockIdx.x; pass < args.npass; pass += gridDim.x) {
    const int64_t col = (int64_t)pass * 128 + wave * 32 + m;
    const PlaneIO io = make_plane_io(args.Np, col, h);
    u32x4 mk[kMaskLayers];
#pragma unroll
    for (int l = 0; l < kMaskLayers; ++l) mk[l] = args.masks[(int64_t)l * args.Np * 2 + (int64_t)pass * 256 + tid];
    auto dp = [&](int row) { return reinterpret_cast<float*>(reinterpret_cast<char*>(args.dplanes) + (int64_t)row * io.row_bytes); };
    const int64_t tile_bytes = 32 * io.row_bytes;
    const float4 dr = reinterpret_cast<const float4*>(args.d_raw)[col];
    // rgb head (model.py:118): dHV[f] = sum_c W_rgb[c][f] * d_rgb[c];  view-layer ReLU; plane kPlHV
    f32x16 Z[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int fo = 32 * t + 8 * gq + 4 * h;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + kSmWRgb + 0 * kCondWidth + fo);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + kSmWRgb + 1 * kCondWidth + fo);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(sm + kSmWRgb + 2 * kCondWidth + fo);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          Z[t][4 * gq + cc] = __builtin_fmaf(w2[cc], dr.z, __builtin_fmaf(w1[cc], dr.y, w0[cc] * dr.x));
      }
    }
    apply_mask_bits(Z, mk[8]);
    store_plane(Z, dp(kPlHV), io);
    f32x16 X[8], Y[8];
    LimbFrag cur[2], nxt[2];
    // d bottleneck = W_view[:, :256]^T . dZ_view   (no activation on the bottleneck, model.py:109)
    zero_tiles_bf(X);
    split_tile<0>(Z[0], cur);
    chunk_mma_bf16<kBbView + 0, 8, true, 0, false, Bf16BwdNet>(p, cur, X, Z[1], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
    chunk_mma_bf16<kBbView + 1, 8, true, 0, false, Bf16BwdNet>(p, cur, X, Z[2], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
    chunk_mma_bf16<kBbView + 2, 8, true, 0, false, Bf16BwdNet>(p, cur, X, Z[3], nxt); cur[0] = nxt[0]; cur[1] = nxt[1];
    chunk_mma_bf16<kBbView + 3, 8, false, 0, false, Bf16BwdNet>(p, cur, X, Z[3], nxt);
    // dH7 = W_bott^T . dBot + W_sigma^T * d_sigma; the consumer of X stores it (plane kPlBot, no mask)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sm + kSmWSigma + 32 * t + 8 * gq + 4 * h);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) Y[t][4 * gq + cc] = w[cc] * dr.w;
      }
    }
    split_tile<0, true>(X[0], cur, dp(kPlBot), &io);
    layer8_bf16<kBbBott, 8, 0, false, 0, true, false, Bf16BwdNet>(p, cur, X, Y, X[0], dp(kPlBot), &io, tile_bytes);
    // trunk, layers 7 .. 1: dZ_l = mask_l * dH_l (stored by the consumer), dH_{l-1} = W_l^T . dZ_l
#define AON_BB_LAYER(IN, OUT, L, CB)                                                                                         \
    zero_tiles_bf(OUT);                                                                                                        \
    split_tile<2, true>(IN[0], cur, dp(plane_h(L)), &io, tile_bits(mk[L], 0));                                                 \
    layer8_bf16<CB, 8, 2, false, 0, true, false, Bf16BwdNet>(p, cur, IN, OUT, IN[0], dp(plane_h(L)), &io, tile_bytes, nullptr, mk[L]);
    AON_BB_LAYER(Y, X, 7, kBbL7 + 0)
    AON_BB_LAYER(X, Y, 6, kBbL7 + 8)
    AON_BB_LAYER(Y, X, 5, kBbL7 + 16)
    AON_BB_LAYER(X, Y, 4, kBbL7 + 24)
    AON_BB_LAYER(Y, X, 3, kBbL7 + 32)
    AON_BB_LAYER(X, Y, 2, kBbL7 + 40)
    AON_BB_LAYER(Y, X, 1, kBbL7 + 48)
#undef AON_BB_LAYER
    apply_mask_bits(X, mk[0]);
    store_plane(X, dp(plane_h(0)), io);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
int num_cus();

int64_t bf16x3_packed_bytes() { return kBfStreamBytes + kSmallBytes; }

hipError_t launch_pack_vanilla_bf16x3(const float* const* params, char* packed, hipStream_t stream) {
  PackArgsB a;
  for (int i = 0; i < kNumVanillaParams; ++i) a.p[i] = params[i];
  const int64_t n = (int64_t)kNumBigChunks * 2 * 8 * 64 + (int64_t)(kNumChunks - kNumBigChunks) * 2 * 4 * 64 + kSmallFloats;
  pack_vanilla_bf16x3_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed);
  return hipGetLastError();
}

template <bool TRAIN>
static hipError_t launch_bf16x3_t(BfArgs a, hipStream_t stream) {
  static DeviceOnce lds_once;
  if (hipError_t e = set_max_lds(&mlp_fwd_bf16x3_kernel<TRAIN>, kBfLdsBytes, lds_once); e != hipSuccess) return e;
  const int cus = num_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  const int grid = a.npass < cus ? a.npass : cus;
  if (grid <= 0) return hipSuccess;
  mlp_fwd_bf16x3_kernel<TRAIN><<<dim3(grid), dim3(256), kBfLdsBytes, stream>>>(a);
  return hipGetLastError();
}

hipError_t launch_mlp_fwd_bf16x3(const char* packed, const float* rays_o, const float* rays_d, const float* viewdirs,
                                 const float* t_vals, int64_t n_rays, int S, float* raw, hipStream_t stream) {
  BfArgs a{packed, rays_o, rays_d, viewdirs, t_vals, raw, n_rays * S, S, (int)((n_rays * S + 127) / 128), nullptr, nullptr, 0};
  return launch_bf16x3_t<false>(a, stream);
}

// training forward of the bf16x3 engine: same planes / masks contract as launch_mlp_fwd_train (aon_mlp.hip)
hipError_t launch_mlp_fwd_train_bf16x3(const char* packed, const float* rays_o, const float* rays_d, const float* viewdirs,
                                       const float* t_vals, int64_t n_rays, int S, float* raw, float* planes, void* masks,
                                       hipStream_t stream) {
  BfArgs a{packed, rays_o, rays_d, viewdirs, t_vals, raw, n_rays * S, S, (int)((n_rays * S + 127) / 128), planes,
           static_cast<u32x4*>(masks), 0};
  a.Np = (int64_t)a.npass * 128;
  return launch_bf16x3_t<true>(a, stream);
}

int64_t bwd_bf16x3_stream_bytes() { return kBbStreamBytes; }

hipError_t launch_pack_vanilla_bwd_bf16x3(const float* const* params, char* packed, hipStream_t stream) {
  PackArgsB a;
  for (int i = 0; i < kNumVanillaParams; ++i) a.p[i] = params[i];
  const int64_t n = (int64_t)kBbNumChunks * 2 * 8 * 64;
  pack_vanilla_bwd_bf16x3_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(a, packed);
  return hipGetLastError();
}

// packed_fwd_small: the fp32 small block (biases / head weights) = fp32 forward stream + kStreamBytes
hipError_t launch_mlp_bwd_chain_bf16x3(const char* packed_bwd, const float* packed_fwd_small, const float* d_raw, const void* masks,
                                       float* dplanes, int64_t Np, hipStream_t stream) {
  static DeviceOnce lds_once;
  constexpr int lds = kBfRingBytes + (int)kSmallBytes;
  if (hipError_t e = set_max_lds(&mlp_bwd_chain_bf16x3_kernel, lds, lds_once); e != hipSuccess) return e;
  BfBwdArgs a{packed_bwd, packed_fwd_small, d_raw, static_cast<const u32x4*>(masks), dplanes, Np, (int)(Np / 128)};
  const int cus = num_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  const int grid = a.npass < cus ? a.npass : cus;
  if (grid <= 0) return hipSuccess;
  mlp_bwd_chain_bf16x3_kernel<<<dim3(grid), dim3(256), lds, stream>>>(a);
  return hipGetLastError();
}

}  // namespace aon
