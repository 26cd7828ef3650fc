// Wave64 primitives and the per-ray inverse-CDF / merge stages of the render path (R6, R7), shared by csrc/aon_render.hip and
// the instruction-floor micro-benchmark tools/ubench/invcdf_floor.hip.  One wavefront per ray; see aon_render.hip.
#pragma once
#include "aon_common.h"

namespace aon {

// ---------------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------------
// Cross-lane data movement by DPP modifiers (no LDS-path ds_bpermute): row_shr:n inside rows of 16 lanes,
// row_bcast:15 / row_bcast:31 to carry a row's last lane into the following rows (GFX9 DPP controls).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float identity, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, false));
}

// inclusive scan over the 64 lanes; lanes without a source keep the identity
template <bool MUL>
__device__ __forceinline__ float wave_inclusive_scan(float v, int /*lane*/) {
  if constexpr (MUL) {
    // v <- dpp(v) * v with the DPP modifier on the multiply itself: a lane without a source (or outside the row mask) is
    // disabled and keeps its value, which is what multiplying by the identity did -- one instruction per step instead of
    // [move 1.0, v_mov_b32_dpp, v_mul_f32] (hipcc folds the DPP move into an add but not into a multiply).  s_nop 1: the two
    // wait states a DPP read needs after the VALU write of its source, which the compiler cannot see inside the asm.
    asm volatile(
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
  } else {
    v = v + dpp_f32<0x111, 0xf>(0.f, v);  // row_shr:1
    v = v + dpp_f32<0x112, 0xf>(0.f, v);  // row_shr:2
    v = v + dpp_f32<0x114, 0xf>(0.f, v);  // row_shr:4
    v = v + dpp_f32<0x118, 0xf>(0.f, v);  // row_shr:8
    v = v + dpp_f32<0x142, 0xa>(0.f, v);  // row_bcast:15 -> rows 1 and 3
    v = v + dpp_f32<0x143, 0xc>(0.f, v);  // row_bcast:31 -> rows 2 and 3
    return v;
  }
}

// wave_shr:1 of a double (two 32-bit DPP moves), zero into lane 0
__device__ __forceinline__ double dpp_shr1_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  // bound_ctrl: a lane without a source reads 0 -- no separate "old = 0" move per half per step
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x138, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// a DPP move of a double (two 32-bit moves); lanes without a source / outside ROW_MASK read +0.0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// inclusive prefix sum of doubles over the 64 lanes, as a 6-step tree (NOT index order: see exact_prefix_f64)
__device__ __forceinline__ double wave_inclusive_sum_f64(double v) {
  v += dpp_f64<0x111, 0xf>(v);
  v += dpp_f64<0x112, 0xf>(v);
  v += dpp_f64<0x114, 0xf>(v);
  v += dpp_f64<0x118, 0xf>(v);
  v += dpp_f64<0x142, 0xa>(v);
  v += dpp_f64<0x143, 0xc>(v);
  return v;
}

// inclusive prefix minimum of ints over the 64 lanes
__device__ __forceinline__ int wave_inclusive_min_i32(int v) {
  constexpr int id = 0x7fffffff;
  auto step = [](int a, int b) { return a < b ? a : b; };
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x114, 0xf, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x118, 0xf, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x142, 0xa, 0xf, false));
  v = step(v, __builtin_amdgcn_update_dpp(id, v, 0x143, 0xc, 0xf, false));
  return v;
}

// sum over the 64 lanes, returned in every lane
__device__ __forceinline__ float wave_sum(float v) {
  v = wave_inclusive_scan<false>(v, 0);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int K, int J>
__device__ __forceinline__ void bitonic_step(float (&v)[4], int lane) {
  if constexpr (J >= 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float o = __shfl_xor(v[r], J >> 2);
      const int e = lane * 4 + r;
      const bool up = (e & K) == 0, lower = (e & J) == 0;
      v[r] = (up == lower) ? __builtin_fminf(v[r], o) : __builtin_fmaxf(v[r], o);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if ((r & J) == 0) {
        const int e = lane * 4 + r;
        const bool up = (e & K) == 0;
        const float lo = __builtin_fminf(v[r], v[r ^ J]), hi = __builtin_fmaxf(v[r], v[r ^ J]);
        v[r] = up ? lo : hi;
        v[r ^ J] = up ? hi : lo;
      }
    }
  }
}

template <int K>
__device__ __forceinline__ void bitonic_merge(float (&v)[4], int lane) {
  if constexpr (K >= 256) bitonic_step<K, 128>(v, lane);
  if constexpr (K >= 128) bitonic_step<K, 64>(v, lane);
  if constexpr (K >= 64) bitonic_step<K, 32>(v, lane);
  if constexpr (K >= 32) bitonic_step<K, 16>(v, lane);
  if constexpr (K >= 16) bitonic_step<K, 8>(v, lane);
  if constexpr (K >= 8) bitonic_step<K, 4>(v, lane);
  if constexpr (K >= 4) bitonic_step<K, 2>(v, lane);
  bitonic_step<K, 1>(v, lane);
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wave is serviced in order; this only pins the compiler's ordering and drains lgkmcnt.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// weights.sum(-1) over the 63 pdf weights in EXACTLY the association torch's CPU sum kernel uses for a contiguous fp32 row
// (helper.py:205; ATen cpu/SumKernel.cpp `vectorized_inner_sum` -> `row_sum`, 8-float vectors, ILP factor 4), measured on
// torch 2.10 with three-element probes (1, 2^-24, 2^-24) over all position triples and confirmed bit-for-bit on 20,000 rows:
//   P[l] = (((((x[l] + x[32+l]) + x[40+l]) + x[48+l]) + x[8+l]) + x[16+l]) + x[24+l]          l = 0..7   (vector lanes)
//   s    = (((((x56 + x57) + x58) + x59) + x60) + x61) + x62                                               (scalar remainder)
//   s    = (((((((s + P0) + P1) + P2) + P3) + P4) + P5) + P6) + P7
// A tree reduction differs from this in the last bit on ~30 % of rows, the cdf inherits that bit, and every draw that falls
// next to the affected knot moves: the whole reason round 1's inverse CDF was "99 % within 2e-6" instead of exact.
__device__ __forceinline__ float torch_sum63(float x, int lane) {
  float P = x;                                                   // lanes 0..7: vector accumulator lane l
  P = __fadd_rn(P, __shfl(x, lane + 32));
  P = __fadd_rn(P, __shfl(x, lane + 40));
  P = __fadd_rn(P, __shfl(x, lane + 48));
  P = __fadd_rn(P, __shfl(x, lane + 8));
  P = __fadd_rn(P, __shfl(x, lane + 16));
  P = __fadd_rn(P, __shfl(x, lane + 24));
  // y[0..14] = x56..x62, P0..P7 in lanes 0..14; the running sum in index order by the shifted-add chain used for the cdf
  const float y_tail = __shfl(x, lane + 56), y_part = __shfl(P, lane - 7);   // both shuffles by ALL lanes (a shuffle inside
  const float y = lane < 7 ? y_tail : y_part;                                 // a divergent branch cannot read disabled lanes)
  float run = y;
#pragma unroll
  for (int k = 0; k < 14; ++k) run = __fadd_rn(dpp_f32<0x138, 0xf>(0.f, run), y);   // wave_shr:1, zero into lane 0
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, run), 14));
}

// torch.cumsum of a float row on the CPU: the running sum is a DOUBLE (at::acc_type<float, false>), taken in index order, and
// every prefix is rounded to float on store.  Returns, in lane j+1, that float for prefix j of the first 62 lanes' values
// (lane 0 gets 0) -- bit for bit.
//
// Fast path: a 6-step tree scan in double.  It gives the same bits as the index-order sum whenever
//   (a) every prefix is exactly representable in a double -- then no addition in either order rounds at all.  With
//       non-negative terms that is a statement about exponents only: [exponent of the prefix] - [smallest ulp exponent among
//       its float terms] <= 52, checked per lane from a prefix minimum; or
//   (b) the prefix is far from every float rounding boundary: the two double sums differ by at most (61 + 6) roundings of
//       2^-53 relative, so if P(1 - 2^-46) and P(1 + 2^-46) round to the same float, so does the index-order sum.
// A wave in which some lane satisfies neither (or holds a negative / NaN term) runs the 61-step index-order chain
//   c <- wave_shr1(c) + p   (lane i holds (((p0 + p1) + p2) + ...) + p_i after i steps; "0 + p" of finished lanes is exact)
// which was the only path in the first round-2 version (57 us per 61,440-ray launch; a float running sum reproduces only 44 %
// of torch's prefixes, a double one all of them).  On NeRF weights the fallback is rare: 0 of 3,000 rays on the synthetic
// scenes, 0.4 % of rows with weights spread over 17 decades (tests/diag/diag_cumsum_guard.py).
__device__ __forceinline__ float exact_prefix_f64(float pdf, int lane) {
  const double pd = (double)pdf;
  const unsigned pb = __builtin_bit_cast(unsigned, pdf);
  int ue = (int)((pb >> 23) & 255u);
  ue = ue < 1 ? 1 : ue;                                     // denormals share the ulp of the smallest normal binade
  if (pdf == 0.f) ue = 0x7fffffff;                          // a zero term constrains nothing
  const int min_ulp = wave_inclusive_min_i32(ue);
  double P = wave_inclusive_sum_f64(pd);
  const int ed = (int)((__builtin_bit_cast(unsigned long long, P) >> 52) & 0x7ffu);
  // (ed - 1023) - (min_ulp - 127 - 23) <= 52, i.e. ed - min_ulp <= 925
  const bool exact = ed - min_ulp <= 924;   // one binade of slack: P's exponent may be one below the true sum's
  const float f_lo = (float)(P * (1.0 - 0x1p-46)), f_hi = (float)(P * (1.0 + 0x1p-46));
  const bool ok = (exact || f_lo == f_hi) && pdf >= 0.f;    // NaN terms fail both comparisons
  if (__builtin_amdgcn_ballot_w64(!ok && lane < 62) != 0ull) {   // wave-uniform
    double run = pd;
#pragma unroll
    for (int j = 0; j < 61; ++j) run = dpp_shr1_f64(run) + pd;
    P = run;
  }
  return (float)dpp_shr1_f64(P);
}

// inclusive prefix sum of ints over the 64 lanes
__device__ __forceinline__ int wave_inclusive_add_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// Per-wave LDS image of one ray's inverse-CDF problem.
constexpr int kPdfCdf = 0, kPdfBin = 64, kPdfT = 128, kPdfOut = 196, kPdfLdsFloats = 392;

// R6 + R7 for ONE ray by one wavefront, in three stages (also timed one by one by tools/ubench/invcdf_floor.hip):
//   invcdf_build   the 64-entry CDF of the 63 weights into LDS, in torch's arithmetic (sum association, double running sum)
//   invcdf_draw    two draws per lane: binary search in the LDS CDF, gathers, the reference's interpolation
//   invcdf_merge   sorted union of the 65 coarse t and the 128 draws
// In: lane i holds bin i (`b`), pdf weight i (`w`, lanes 0..62) and, if `t_fine` is wanted, t_coarse[i] (`tc`) with t_coarse[64]
// in `t64`; `u_row` points at the ray's 128 draws.  Shared by the stand-alone kernel (operands from memory) and the fused coarse
// compositing kernel (operands still in registers).

// pdf / cdf  (helper.py:206-222): cdf64 = [0, min(1, cumsum(pdf[:-1])) (62 entries), 1] -> L[kPdfCdf ..]; also parks the bins
// (and the coarse t) in LDS for the later stages' gathers
__device__ __forceinline__ void invcdf_build(float* L, int lane, float tc, float t64, float b, float w, bool want_t) {
  float* cdf = L + kPdfCdf;
  float* bin = L + kPdfBin;
  float* tt = L + kPdfT;
  bin[lane] = b;
  if (want_t) {
    tt[lane] = tc;
    if (lane == 0) tt[64] = t64;
  }
  float wsum = torch_sum63(w, lane);
  const float padding = __builtin_fmaxf(0.f, __fsub_rn(1e-5f, wsum));
  if (padding != 0.f) {   // wave-uniform (wsum is); w + 0/63 and wsum + 0 are the identity, so the common case skips a division
    w = __fadd_rn(w, __fdiv_rn(padding, 63.0f));
    wsum = __fadd_rn(wsum, padding);
  }
  const float pdf = __fdiv_rn(w, wsum);
  const float prefix = exact_prefix_f64(pdf, lane);   // lane j+1 <- float(p0 + ... + pj) in torch's arithmetic; lane 0 <- 0
  const float mine = __builtin_fminf(1.f, prefix);
  cdf[lane] = lane == 63 ? 1.f : mine;  // lane 0 keeps 0
  wave_lds_sync();
}

// draws j = lane, lane + 64 (helper.py:224-243): smp[k] = the sample, guess[k] = index of the first coarse t that can exceed it
__device__ __forceinline__ void invcdf_draw(const float* L, int lane, const float* u_row, float* samples, float (&smp)[2], int (&guess)[2]) {
  const float* cdf = L + kPdfCdf;
  const float* bin = L + kPdfBin;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = lane + 64 * k;
    const float u = u_row[j];
    // idx = #(cdf <= u)  == searchsorted(cdf, u, right=True); cdf is non-decreasing
    int idx = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
      if (cdf[idx + step - 1] <= u) idx += step;
    }
    if (idx == 63 && cdf[63] <= u) idx = 64;
    const int i0 = idx - 1 < 0 ? 0 : idx - 1;
    const int i1 = idx > 63 ? 63 : idx;
    const float c0 = cdf[i0], c1 = cdf[i1], b0 = bin[i0], b1 = bin[i1];
    float t = __fdiv_rn(__fsub_rn(u, c0), __fsub_rn(c1, c0));
    if (t != t) t = 0.f;                                   // nan_to_num(., 0)
    t = __builtin_fminf(__builtin_fmaxf(t, 0.f), 1.f);     // clip (+-inf land on the same ends as nan_to_num + clip)
    smp[k] = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
    guess[k] = i0 + 1;   // with bins = interval mids: t_0 .. t_i0 <= bin_i0 <= sample
    if (samples) samples[j] = smp[k];
  }
}

// sort(cat[t_coarse(65), samples(128)]) -> 193   (helper.py:250).  A sorted multiset is unique, so any correct merge gives
// torch.sort's values.  Fast path -- t_coarse non-decreasing (always, out of sample_along_rays) and the draws too (they are
// when u is: the deterministic grid, or pre-sorted random u), decided by a wave-uniform order check: every sample finds its
// rank c = #{i : t_i <= s} among the coarse t by a short walk from the bin index its draw landed in and goes to slot j + c
// of a NaN-filled 193-slot LDS row; the coarse t fill the remaining slots in order (slot e takes t[e - #samples before e],
// counted with one integer scan).  ~75 wave instructions against ~170 for the 8-stage bitonic merge of the first version.
// (cross-lane reads are done by ALL lanes into temporaries and selected afterwards: inside a `lane == 63 ? a : shuffle`
// arm lane 63 is disabled and lane 62 would read the identity instead of lane 63's value)
__device__ __forceinline__ void invcdf_merge(float* L, int lane, float tc, float t64, const float (&smp)[2], const int (&guess)[2], float* t_fine) {
  float* tt = L + kPdfT;
  float* ob = L + kPdfOut;
  const float t_next = tt[lane + 1];
  const float s1_first = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, smp[1]), 0));
  const float s0_shl = dpp_f32<0x130, 0xf>(0.f, smp[0]);          // wave_shl:1 -> lane+1's value
  const float s1_next = dpp_f32<0x130, 0xf>(0.f, smp[1]);
  const float s0_next = lane == 63 ? s1_first : s0_shl;
  const bool ok = tc <= t_next && smp[0] <= s0_next && (lane == 63 || smp[1] <= s1_next);
  float v[4];
  bool merged = false;   // wave-uniform
  if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) {
    const float qnan = __builtin_nanf("");
    if (lane < 49) *reinterpret_cast<float4*>(ob + 4 * lane) = make_float4(qnan, qnan, qnan, qnan);
    // (LDS instructions of one wave execute in order: the fill above lands before the scatter below)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      // rank c = #{i : t_i <= s}: t_0 .. t_g-1 lie below the sample's bin (g = guess), t_g+2 .. above it
      const int g = guess[k];
      const float sv = smp[k];
      const int g1 = g < 64 ? g : 64, g2 = g + 1 < 64 ? g + 1 : 64;
      const float ta = tt[g1], tb = tt[g2];
      const int c = g + ((g <= 64 && ta <= sv) ? 1 : 0) + ((g + 1 <= 64 && tb <= sv) ? 1 : 0);
      ob[lane + 64 * k + c] = sv;
    }
    wave_lds_sync();
    float4 q = make_float4(qnan, qnan, qnan, qnan);
    if (lane < 49) q = *reinterpret_cast<const float4*>(ob + 4 * lane);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) cnt += (v[r] == v[r]) ? 1 : 0;
    const int incl = wave_inclusive_add_i32(cnt);
    int before = incl - cnt;   // samples in slots < 4 * lane
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = lane * 4 + r;
      const int ti = e - before;
      const float tv = tt[ti < 0 ? 0 : (ti > 64 ? 64 : ti)];
      if (v[r] == v[r]) ++before;
      else if (e < 193) v[r] = tv;
    }
    // The rank above leans on bins being the mids of t_coarse (any `bins` can be passed to the stand-alone entry) and on the
    // draws' rounding; it is therefore CHECKED: all 128 samples landed in distinct slots and the 193 keys are in order -- a
    // permutation of the inputs in non-decreasing order is the sorted sequence.  Otherwise fall through to the full sort.
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const float nxt = dpp_f32<0x130, 0xf>(0.f, v[0]);  // wave_shl:1 -> lane+1's first element
    const bool in_order = (lane * 4 + 1 >= 193 || v[0] <= v[1]) && (lane * 4 + 2 >= 193 || v[1] <= v[2]) &&
                          (lane * 4 + 3 >= 193 || v[2] <= v[3]) && (lane * 4 + 4 >= 193 || v[3] <= nxt);
    merged = total == 128 && __builtin_amdgcn_ballot_w64(in_order) == ~0ull;
  }
  if (!merged) {
    // general order: full 36-stage bitonic sort of 256 (+inf padded) keys held 4 per lane (element e = lane*4 + r)
    ob[lane] = tc;
    if (lane == 0) ob[64] = t64;
    ob[65 + lane] = smp[0];
    ob[129 + lane] = smp[1];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = lane * 4 + r;
      v[r] = e < 193 ? ob[e] : __builtin_inff();
    }
    bitonic_merge<2>(v, lane); bitonic_merge<4>(v, lane); bitonic_merge<8>(v, lane); bitonic_merge<16>(v, lane);
    bitonic_merge<32>(v, lane); bitonic_merge<64>(v, lane); bitonic_merge<128>(v, lane); bitonic_merge<256>(v, lane);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = lane * 4 + r;
    if (e < 193) t_fine[e] = v[r];
  }
}

__device__ __forceinline__ void inverse_cdf_merge(float* L, int lane, float tc, float t64, float b, float w, const float* u_row,
                                                  float* samples, float* t_fine) {
  invcdf_build(L, lane, tc, t64, b, w, t_fine != nullptr);
  float smp[2];
  int guess[2];
  invcdf_draw(L, lane, u_row, samples, smp, guess);
  if (t_fine) invcdf_merge(L, lane, tc, t64, smp, guess, t_fine);
}

}  // namespace aon
