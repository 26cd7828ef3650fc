// General-geometry NeRFMLP engine (csrc/aon_gmlp.hip): launch interface shared with the C ABI layer.
#pragma once
#include "aon_common.h"

namespace aon {

struct GemmSeg {
  const float* X; int64_t ldx; int rowdiv;   // row m of the product reads X[(m / rowdiv) * ldx + k]
  const float* W; int64_t ldw;               // W[n * ldw + k]
  int K;
};
struct GemmArgs {
  GemmSeg seg[2]; int nseg;
  const float* bias;         // (N,) or null
  float* Y; int64_t ldy;
  int64_t M; int N;
  int epi;                   // 0 none, 1 ReLU, 2 mask by aux[m * ldaux + n] > 0
  const float* aux; int64_t ldaux;
};

hipError_t launch_gemm_tn(const GemmArgs& a, hipStream_t stream);
hipError_t launch_transpose(const float* W, int64_t ldw, int N, int K, float* WT, hipStream_t stream);
int64_t wgrad_part_floats(int64_t M, int N, int K);
// dW[0:N, 0:K] (row stride ldd: a column block of an nn.Linear weight) = A[M x N]^T . B[M x K] (row m reads B row m / rowdiv)
hipError_t launch_wgrad_nk(const float* A, int64_t lda, const float* B, int64_t ldb, int rowdiv, int64_t M, int N, int K, float* dW, int64_t ldd,
                           float* part, hipStream_t stream);
// db[0:N] = column sums of A[M x N]; `part` holds at least 512 * N floats
hipError_t launch_colsum(const float* A, int64_t lda, int64_t M, int N, float* db, float* part, hipStream_t stream);

}  // namespace aon
