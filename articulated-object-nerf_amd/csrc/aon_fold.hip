// bottleneck_layer folded into views_linear[0] (models/vanilla_nerf/model.py:105-118, model_autodecoder.py:222-237): the products on the
// PARAMETERS (pack time: W' = W_v0[:, :256] W_b, b' = W_v0[:, :256] b_b + b_v0) and on their GRADIENTS (backward: the reference's
// dW_b, db_b, dW_v0[:, :256] from the folded layer's dW', db').  128 x 256 x 256 multiply-adds each -- 0.03 % of a training step's
// arithmetic -- accumulated in fp64 and rounded once, so the folded layer is the correctly rounded product of the fp32 parameters
// and the un-folded gradients carry no summation error of their own.
#include "aon_fold.h"

#include <mutex>
#include <unordered_map>

namespace aon {

// ---------------------------------------------------------------------------------------------
// which form a packed buffer holds
// ---------------------------------------------------------------------------------------------
static std::atomic<int> g_fold_default{1};
int fold_default() { return g_fold_default.load(std::memory_order_relaxed); }
void set_fold_default(int on) { g_fold_default.store(on ? kFormFolded : kFormLiteral, std::memory_order_relaxed); }

namespace {
std::mutex g_form_mu;
std::unordered_map<uintptr_t, int>& form_table() {
  static std::unordered_map<uintptr_t, int> t;
  return t;
}
}  // namespace

void set_stream_form(const void* p, int form) {
  std::lock_guard<std::mutex> lk(g_form_mu);
  // (never cleared: an entry is 50 bytes and a caching allocator hands the same addresses out again; clearing while buffers are live
  // would make them "never packed" -- ADVICE r5)
  form_table()[reinterpret_cast<uintptr_t>(p)] = form;
}

int stream_form(const void* p) {
  std::lock_guard<std::mutex> lk(g_form_mu);
  auto& t = form_table();
  auto it = t.find(reinterpret_cast<uintptr_t>(p));
  return it == t.end() ? kFormUnknown : it->second;
}

// ---------------------------------------------------------------------------------------------
// small products in fp64
// ---------------------------------------------------------------------------------------------
struct FoldArgs {
  FoldGemm job[kFoldMaxJobs];
  int blk_begin[kFoldMaxJobs + 1];
  int njobs;
};

// 16 x 16 outputs per block, K in tiles of 64 through LDS (as doubles: the conversion is paid once per element, not once per use).
// Tile loads pick the thread order that makes the faster-varying thread index walk the operand's unit stride.  The kernel is pure load
// latency (128 x 256 x 256 multiply-adds on a whole chip), so the loads are what is shaped (round 6): every load is UNCONDITIONAL, from
// an index clamped into the operand, and the out-of-range ones are replaced by 0 afterwards -- as `in_range ? load : 0` each load sat
// in a branch of its own with an `s_waitcnt vmcnt(0)` behind it, so the A and B loads of a tile ran one after the other, 2 x 16
// memory round trips for K = 256 -- and a tile is 64 deep: eight loads in flight per thread, four rounds.  The multiply-adds of an
// output run in the same order of k with the same values (zeros where they were): same bits.  The four products of a step's prologue:
// 18 -> 12 us; the six of its un-folding: 20 -> 17 us (tiles of 128: no further change -- what is left is the chain of 256 dependent
// fp64 multiply-adds fed from LDS and the launch itself).
constexpr int kFoldKT = 64;
__global__ void __launch_bounds__(256) fold_gemm_kernel(FoldArgs a) {
  __shared__ double As[16][kFoldKT + 1], Bs[kFoldKT][17];
  int j = 0;
#pragma unroll 1
  for (int t = 1; t < a.njobs; ++t)
    if ((int)blockIdx.x >= a.blk_begin[t]) j = t;
  const FoldGemm& G = a.job[j];
  const int blk = (int)blockIdx.x - a.blk_begin[j];
  const int nbn = (G.N + 15) / 16;
  const int m0 = (blk / nbn) * 16, n0 = (blk % nbn) * 16;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const bool a_k_fast = G.sak <= G.sam, b_n_fast = G.sbn <= G.sbk;
  const int amm = a_k_fast ? ty : tx, akk = a_k_fast ? tx : ty;
  const int bkk = b_n_fast ? ty : tx, bnn = b_n_fast ? tx : ty;
  const int am = m0 + amm, bn = n0 + bnn;
  const int am_c = am < G.M ? am : G.M - 1, bn_c = bn < G.N ? bn : G.N - 1;   // (M, N, K >= 1)
  double acc = 0.0;
  for (int k0 = 0; k0 < G.K; k0 += kFoldKT) {
    float av[kFoldKT / 16], bv[kFoldKT / 16];
#pragma unroll
    for (int q = 0; q < kFoldKT / 16; ++q) {
      const int ka = k0 + 16 * q + akk, kb = k0 + 16 * q + bkk;
      const int ka_c = ka < G.K ? ka : G.K - 1, kb_c = kb < G.K ? kb : G.K - 1;
      av[q] = G.A[(int64_t)am_c * G.sam + (int64_t)ka_c * G.sak];
      bv[q] = G.B[(int64_t)kb_c * G.sbk + (int64_t)bn_c * G.sbn];
    }
#pragma unroll
    for (int q = 0; q < kFoldKT / 16; ++q) {
      const int ka = k0 + 16 * q + akk, kb = k0 + 16 * q + bkk;
      As[amm][16 * q + akk] = (am < G.M && ka < G.K) ? (double)av[q] : 0.0;
      Bs[16 * q + bkk][bnn] = (kb < G.K && bn < G.N) ? (double)bv[q] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kFoldKT; ++kk) acc = __builtin_fma(As[ty][kk], Bs[kk][tx], acc);
    __syncthreads();
  }
  const int m = m0 + ty, n = n0 + tx;
  if (m < G.M && n < G.N) {
    if (G.u) acc += (double)G.u[m] * (G.v ? (double)G.v[n] : 1.0);
    G.C[(int64_t)m * G.ldc + n] = (float)acc;
  }
}

hipError_t launch_fold_gemms(const FoldGemm* jobs, int njobs, hipStream_t stream) {
  if (njobs < 1 || njobs > kFoldMaxJobs) return hipErrorInvalidValue;
  FoldArgs a{};
  a.njobs = njobs;
  int blk = 0;
  for (int j = 0; j < njobs; ++j) {
    a.job[j] = jobs[j];
    a.blk_begin[j] = blk;
    blk += ((jobs[j].M + 15) / 16) * ((jobs[j].N + 15) / 16);
  }
  a.blk_begin[njobs] = blk;
  fold_gemm_kernel<<<dim3(blk), dim3(256), 0, stream>>>(a);
  return hipGetLastError();
}

void fold_view_jobs(const float* Wv, int ldv, const float* bv, const float* Wb, const float* bb, float* Wf, float* bf, FoldGemm jobs[2]) {
  jobs[0] = FoldGemm{Wv, ldv, 1, Wb, 256, 1, Wf, 256, 128, 256, 256, nullptr, nullptr};      // W'[o][i] = sum_k Wv[o][k] Wb[k][i]
  jobs[1] = FoldGemm{Wv, ldv, 1, bb, 1, 0, bf, 1, 128, 1, 256, bv, nullptr};                  // b'[o] = sum_k Wv[o][k] bb[k] + bv[o]
}

hipError_t launch_fold_view(const float* Wv, int ldv, const float* bv, const float* Wb, const float* bb, float* Wf, float* bf, hipStream_t stream) {
  FoldGemm jobs[2];
  fold_view_jobs(Wv, ldv, bv, Wb, bb, Wf, bf, jobs);
  return launch_fold_gemms(jobs, 2, stream);
}

void unfold_view_jobs(const float* dWf, const float* dbf, const float* Wv, int ldv, const float* Wb, const float* bb, float* dWv, int ld_dwv,
                      float* dWb, float* dbb, FoldGemm jobs[3]) {
  jobs[0] = FoldGemm{Wv, 1, ldv, dWf, 256, 1, dWb, 256, 256, 256, 128, nullptr, nullptr};     // dWb[k][i] = sum_o Wv[o][k] dW'[o][i]
  jobs[1] = FoldGemm{dWf, 256, 1, Wb, 1, 256, dWv, ld_dwv, 128, 256, 256, dbf, bb};           // dWv[o][k] = sum_i dW'[o][i] Wb[k][i] + db'[o] bb[k]
  jobs[2] = FoldGemm{Wv, 1, ldv, dbf, 1, 0, dbb, 1, 256, 1, 128, nullptr, nullptr};           // dbb[k] = sum_o Wv[o][k] db'[o]
}

hipError_t launch_unfold_view(const float* dWf, const float* dbf, const float* Wv, int ldv, const float* Wb, const float* bb, float* dWv, int ld_dwv,
                              float* dWb, float* dbb, hipStream_t stream) {
  FoldGemm jobs[3];
  unfold_view_jobs(dWf, dbf, Wv, ldv, Wb, bb, dWv, ld_dwv, dWb, dbb, jobs);
  return launch_fold_gemms(jobs, 3, stream);
}

}  // namespace aon
