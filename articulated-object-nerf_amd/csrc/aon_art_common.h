// Constants shared by the articulated forward (aon_mlp_art.hip) and backward (aon_train_art.hip).
#pragma once
#include "aon_mlp_core.h"

namespace aon {

// ---- per-call small block (floats), rebuilt by aon_art_prepare because it depends on the latents ----
constexpr int kA_BD0 = 0;       // 128   effective bias of deformations_linear.0 (shape + articulation folded in)
constexpr int kA_WD0 = 128;     // 3x128 deformations_linear.0 weight, [xyz][feature]
constexpr int kA_BD = 512;      // 3x128 biases of deformations_linear.1..3
constexpr int kA_WDL = 896;     // 3x128 deformation_layer weight rows
constexpr int kA_BDL = 1280;    // 3 (+1 pad)
constexpr int kA_BT = 1284;     // 8x256 trunk biases (layers 0 and 5 effective: shape latent folded in)
constexpr int kA_BBOT = 3332;   // 256
constexpr int kA_BV = 3588;     // 4x128 view-branch biases (layer 0 effective: appearance latent folded in)
constexpr int kA_WSIG = 4100;   // 256
constexpr int kA_WRGB = 4356;   // 3x128
constexpr int kA_BSIG = 4740;   // 1
constexpr int kA_BRGB = 4741;   // 3
constexpr int kA_ESC = 4744;    // 10 (+2 pad): encoding scales 2^(min_deg_point + l), 0 for levels the network lacks (round 4: other degrees)
constexpr int kASmallFloats = 4756;
constexpr int kALdsBytes = kRingBytes + kASmallFloats * 4;

// parameter order of the articulated NeRFMLP (model_autodecoder.py:60-170):
//   0..7   deformations_linear.{0..3}.{weight,bias}      8,9  deformation_layer.{weight,bias}
//   10..25 pts_linears.{0..7}.{weight,bias}              26..33 views_linear.{0..3}.{weight,bias}
//   34,35  bottleneck_layer   36,37 density_layer   38,39 rgb_layer
constexpr int kNumArtParams = 40;

// ---- chunk stream of one articulated MLP (execution order) ----
constexpr int kAChD1 = 0;     // deformations_linear.1..3 : 4 chunks each, 4 output tiles (16 KiB)
constexpr int kAChT0 = 12;    // pts_linears.0 : 2 pos-enc chunks (32 KiB)
constexpr int kAChT1 = 14;    // pts_linears.1..4 : 8 chunks each
constexpr int kAChT5 = 46;    // pts_linears.5 : 8 hidden + 2 pos-enc
constexpr int kAChT6 = 56;
constexpr int kAChT7 = 64;
constexpr int kAChBott = 72;
constexpr int kAChV0 = 80;    // views_linear.0 : 8 hidden + 1 view-enc, 4 output tiles (16 KiB)
constexpr int kAChV1 = 89;    // views_linear.1..3 : 4 chunks each
constexpr int kANumChunks = 101;

// ---- transposed chunk stream (execution order of the backward chain) ----
constexpr int kABwV3 = 0;      // views_linear.3^T, .2^T, .1^T : 4 chunks each, 4 output tiles (16 KiB)
constexpr int kABwV0 = 12;     // views_linear.0[:, :256]^T : 4 chunks, 8 output tiles (32 KiB)
constexpr int kABwBott = 16;   // bottleneck^T : 8
constexpr int kABwL7 = 24;     // pts_linears.7^T, .6^T : 8 each
constexpr int kABwL5E = 40;    // pts_linears.5[:, 256:319]^T : 8 chunks, 2 output tiles (8 KiB)  -> d pos-enc
constexpr int kABwL5 = 48;     // pts_linears.5[:, :256]^T, then .4 .3 .2 .1 : 8 each
constexpr int kABwL0E = 88;    // pts_linears.0[:, :63]^T : 8 chunks, 2 output tiles              -> d pos-enc
constexpr int kABwD3 = 96;     // deformations_linear.3^T, .2^T, .1^T : 4 chunks each, 4 output tiles
constexpr int kABwNumChunks = 108;

// ---- FOLDED forms (round 5; aon_common.h kChFView for the algebra): bottleneck_layer (no activation, model_autodecoder.py:223) folded into
// views_linear.0's 256 bottleneck columns, W' = W_v0[:, :256] W_b -- 65,536 of the 692,480 MACs per sample the un-folded kernels execute.
// Forward stream: the eight bottleneck chunks are gone, every later chunk moves up by 8; views_linear.0's hidden chunks hold W' and
// read the post-ReLU layer-7 output.  The per-call block's effective bias of views_linear.0 additionally carries W_v0[:, :256] b_b.
// views_linear.0's view-encoding chunk comes FIRST (as in the vanilla network, aon_common.h): its term and the effective bias are a constant
// of the ray, handed to the whole-path kernels as a per-ray bias (ArtFoldVbNet: the same buffer without that chunk).
constexpr int kAChFV0 = 72;    // W' : 1 view-enc + 8 hidden
constexpr int kAChFV1 = 81;
constexpr int kANumChunksF = 93;
// Transposed stream: views_linear.0's four chunks hold W'^T (d H7 directly), the eight bottleneck chunks are gone.
constexpr int kABwFL7 = 16;
constexpr int kABwFL5E = 32;
constexpr int kABwFL5 = 40;
constexpr int kABwFL0E = 80;
constexpr int kABwFD3 = 88;
constexpr int kABwFNumChunks = 100;


}  // namespace aon
