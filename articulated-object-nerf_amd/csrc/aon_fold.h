// Folding bottleneck_layer into views_linear[0] (round 5): small fp64-accumulating products on the parameters and their gradients, and
// the per-pointer record of which form a packed buffer holds.  See aon_common.h (kChFView) for the algebra.
#pragma once
#include "aon_common.h"

namespace aon {

enum : int { kFormUnknown = -1, kFormLiteral = 0, kFormFolded = 1 };

// process-wide default of the pack entry points (aon_set_bottleneck_fold); every pack call reads it ONCE
int fold_default();
void set_fold_default(int on);
// The form of a packed buffer is a property of the BUFFER, decided when it was packed: the launchers look it up by pointer, so flipping
// the switch between packing a stream and using it cannot pair a stream with the other form's kernel.  A pointer this process never
// packed or declared (a copy of a packed buffer) is kFormUnknown and every launcher refuses it (round 6, ADVICE r5: rounds 5 assumed the
// current default, which pairs a cloned literal stream with the folded kernels after the switch moved).  The owner of a copy states its
// form with aon_declare_stream_form; the Python binding carries the form next to the tensor and re-declares it on every call.
void set_stream_form(const void* p, int form);
int stream_form(const void* p);

// C[m * ldc + n] = sum_k A[m * sam + k * sak] * B[k * sbk + n * sbn]  (+ u[m] * (v ? v[n] : 1)),  accumulated in fp64, rounded once
struct FoldGemm {
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbk, sbn;
  float* C; int64_t ldc;
  int M, N, K;
  const float* u; const float* v;
};
constexpr int kFoldMaxJobs = 8;
hipError_t launch_fold_gemms(const FoldGemm* jobs, int njobs, hipStream_t stream);

// Wf (128, 256) = Wv[:, :256] Wb,  bf (128) = Wv[:, :256] bb + bv      (Wv: (128, ldv), Wb: (256, 256))
hipError_t launch_fold_view(const float* Wv, int ldv, const float* bv, const float* Wb, const float* bb, float* Wf, float* bf, hipStream_t stream);
// ... its two products as jobs, for a caller that folds several networks / buffers in one launch (launch_fold_gemms)
void fold_view_jobs(const float* Wv, int ldv, const float* bv, const float* Wb, const float* bb, float* Wf, float* bf, FoldGemm jobs[2]);
// the parameter gradients the reference's autograd would produce, from the folded layer's:
//   dWb (256, 256) = Wv[:, :256]^T dWf        dbb (256) = Wv[:, :256]^T dbf        dWv[:, :256] (ld ld_dwv) = dWf Wb^T + dbf (x) bb
// (dbv = dbf is written by the weight-gradient kernels directly)
hipError_t launch_unfold_view(const float* dWf, const float* dbf, const float* Wv, int ldv, const float* Wb, const float* bb, float* dWv, int ld_dwv,
                              float* dWb, float* dbb, hipStream_t stream);
// ... its three products as jobs, for a caller that launches several levels' at once (launch_fold_gemms)
void unfold_view_jobs(const float* dWf, const float* dbf, const float* Wv, int ldv, const float* Wb, const float* bb, float* dWv, int ld_dwv,
                      float* dWb, float* dbb, FoldGemm jobs[3]);

}  // namespace aon
