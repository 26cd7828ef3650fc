// extern "C" surface of libaon_hip.so (declared in include/aon_hip.h) and the whole-path orchestration.
#include "../../include/aon_hip.h"
#include "aon_common.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "aon_fold.h"
#include "aon_gmlp.h"

namespace aon {
hipError_t launch_pack_vanilla(const float* const* params, float* packed, hipStream_t stream, int pos_levels = 10, int view_levels = 4, bool fold_done = false);
void vanilla_fold_jobs_fwd(const float* const* params, float* packed, int view_levels, FoldGemm jobs[2]);
void vanilla_fold_jobs_bwd(const float* const* params, float* packed, int view_size, FoldGemm jobs[2]);
hipError_t launch_mlp_fwd(const char* packed, const float* rays_o, const float* rays_d, const float* viewdirs,
                          const float* t_vals, int64_t n_rays, int S, float* raw, hipStream_t stream, const float* view_bias = nullptr);
hipError_t launch_view_bias(const char* packed, const float* viewdirs, int64_t n_rays, float* out, hipStream_t stream);
hipError_t launch_mlp_fwd_enc(const char* packed, const float* samples_enc, const float* viewdirs_enc, int64_t n_rays,
                              int S, float* raw, hipStream_t stream);
hipError_t launch_pack_art(const float* const* params, float* packed, hipStream_t stream, int pos_levels = 10, int view_levels = 4, bool fold_done = false);
FoldGemm art_fold_job_fwd(const float* const* params, float* packed, int view_levels);
hipError_t launch_pack_prepare_art2(const float* const* const params[2], const float* shape, const float* app, const float* art, float* const packed[2],
                                    float* const small[2], hipStream_t stream, int min_deg, int pos_levels, int view_levels, int form);
hipError_t launch_pack_art_bwd2(const float* const* const params[2], float* const packed[2], hipStream_t stream, int pos_levels, int view_levels, int form);
FoldGemm art_fold_job_bwd(const float* const* params, float* packed, int view_levels);
hipError_t launch_prepare_art(const float* const* params, const float* shape, const float* app, const float* art,
                              float* small, hipStream_t stream, int min_deg = 0, int pos_levels = 10, int view_levels = 4);
hipError_t launch_art_mlp_fwd(const char* packed, const float* small, const float* rays_o, const float* rays_d,
                              const float* viewdirs, const float* t_vals, int64_t n_rays, int S, float* raw,
                              hipStream_t stream, const float* view_bias = nullptr);
hipError_t launch_art_view_bias(const char* packed, const float* small, const float* viewdirs, int64_t n_rays, float* out, hipStream_t stream);
hipError_t launch_art_mlp_fwd_pos(const char* packed, const float* small, const float* pos, const float* viewdirs_enc,
                                  int64_t n_rays, int S, float* raw, hipStream_t stream);
int64_t art_stream_bytes();
int64_t art_small_bytes();
hipError_t launch_mlp_fwd_train(const char* packed, const float* rays_o, const float* rays_d, const float* viewdirs,
                                const float* t_vals, int64_t n_rays, int S, float* raw, float* planes, void* masks,
                                hipStream_t stream, int64_t np_total = 0, const float* view_bias = nullptr);
hipError_t launch_composite_bwd(const float* raw, const float* t_vals, const float* dirs, const float* g_rgb, const float* g_acc,
                                const float* g_depth, int64_t n_rays, int S, int white_bkgd, const ActParams& ap, float* d_raw,
                                hipStream_t stream);
hipError_t launch_pack_vanilla_bwd(const float* const* params, float* packed, hipStream_t stream, int pos_size = 63, int view_size = 27, bool fold_done = false);
hipError_t launch_mlp_fwd_train_enc(const char* packed, const float* samples_enc, const float* viewdirs_enc, int64_t n_rays, int S, float* raw,
                                    float* planes, void* masks, hipStream_t stream, int64_t np_total = 0);
int64_t bwd_stream_bytes();
hipError_t launch_mlp_bwd_chain(const char* packed_bwd, const char* packed_fwd, const float* d_raw, const void* masks,
                                float* dplanes, int64_t Np, hipStream_t stream);
hipError_t launch_mlp_bwd_chain2(const ChainSeg* segs, int nsegs, hipStream_t stream);
hipError_t launch_art_bwd_chain2(const ChainSeg* segs, int nsegs, hipStream_t stream);
int64_t wgrad_workspace_bytes();
int wgrad_plan_describe(bool art, int64_t Np, int cus, int32_t* out6, int max_jobs, int64_t* ws_bytes);
void set_wgrad_probe(long long* buf);
int wgrad_plan_segment(bool art, int64_t Np, int cus, int j, int wg, int32_t* begin_end);
hipError_t launch_wgrad_kind_bench(int kind, int nlayers, const float* planes, const float* dplanes, int rows_total, int64_t Np, float* ws,
                                   float* out_scratch, hipStream_t stream);
struct WgAux { hipStream_t stream; hipEvent_t fork, join; };   // aon_wgrad.h: optional side stream of a level's head reductions
struct WgPost { const WgAux* side; hipEvent_t wait_first; };   // aon_wgrad.h: a level's second stage + finishing kernels on a side stream
hipError_t launch_vanilla_wgrad(const float* planes, const float* dplanes, const float* d_raw, int64_t Np, float* const* grads,
                                float* ws, hipStream_t stream, const WgAux* aux, const void* packed_bwd, int phase = 0, const struct WgPost* post = nullptr,
                                struct VanillaWgDeferred* defer = nullptr);
constexpr int kVanillaWgDeferredBytes = 4096;   // aon_train.hip (static_assert there)
hipError_t launch_vanilla_wgrad_post2(const struct VanillaWgDeferred* d0, const struct VanillaWgDeferred* d1, hipStream_t stream);
int vanilla_wgrad_deferred_bytes();
hipError_t launch_art_mlp_fwd_train(const char* packed, const float* small, const float* rays_o, const float* rays_d,
                                    const float* viewdirs, const float* t_vals, int64_t n_rays, int S, float* raw, float* planes,
                                    void* masks, hipStream_t stream, int64_t np_total = 0, const float* view_bias = nullptr);
hipError_t launch_art_mlp_fwd_train2(const TrainSeg* segs, int nsegs, hipStream_t stream);
hipError_t launch_mlp_fwd_train2(const TrainSeg* segs, int nsegs, hipStream_t stream);
int num_cus();
hipError_t launch_pack_art_bwd(const float* const* params, float* packed, hipStream_t stream, int pos_levels = 10, int view_levels = 4, bool fold_done = false);
int64_t art_bwd_stream_bytes();
hipError_t launch_art_bwd_chain(const char* packed_bwd, const float* small, const float* d_raw, const void* masks, const float* planes,
                                float* dplanes, float* dxp, int64_t Np, hipStream_t stream);
hipError_t launch_art_wgrad(const float* planes, const float* dplanes, const float* d_raw, const float* dxp, int64_t Np,
                            const float* const* params, const float* shape, const float* app, const float* art,
                            float* const* grads, float* g_shape, float* g_app, float* g_art, float* ws, hipStream_t stream, const WgAux* aux, int pos_levels, int view_levels,
                            const void* packed_bwd, int phase = 0, bool accumulate_latents = false, const struct WgPost* post = nullptr,
                            struct ArtWgDeferred* defer = nullptr);
constexpr int kArtWgDeferredBytes = 4096;   // aon_train_art.hip (static_assert there): a level's second stage handed back instead of launched
hipError_t launch_art_wgrad_post2(const struct ArtWgDeferred* d0, const struct ArtWgDeferred* d1, hipStream_t stream);
int art_wgrad_deferred_bytes();
hipError_t launch_train_loss(bool backward, const float* rgb_c, const float* rgb_f, const float* target, int64_t n, const float* const* lat, const int* lat_len,
                             float reg_scale, float* stats, float* loss, const float* go, float* d_rgb_c, float* d_rgb_f, float* const* d_lat, hipStream_t stream);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps, int64_t step, hipStream_t stream);
hipError_t launch_code_library(bool backward, const float* const* src, const int64_t* const* idx, const int* rows, const int* dim, float* const* dst, hipStream_t stream);
hipError_t launch_raygen(const float* c2w, int H, int W, float focal, const float* directions, int64_t pix_begin,
                         int64_t pix_end, float* rays_o, float* viewdirs, float* rays_d, hipStream_t stream);
hipError_t launch_ray_directions(int H, int W, float focal, float* out, hipStream_t stream);
hipError_t launch_ray_radii(const float* directions, const float* c2w, int H, int W, float* radii, hipStream_t stream);
hipError_t launch_cast_rays(const float* t_vals, const float* o, const float* d, int64_t n_rays, int S, float* coords,
                            hipStream_t stream);
hipError_t launch_sample_along_rays(const float* rays_o, const float* rays_d, int64_t n_rays, int S, float near, float far,
                                    const float* t_rand, float* t_vals, float* coords, hipStream_t stream, int lindisp = 0,
                                    float inv_near = 0.f, float inv_far = 0.f);
hipError_t launch_pos_enc(const float* x, int64_t n, int min_deg, int max_deg, float* out, hipStream_t stream, int ld = 0, int levels_out = 0);
hipError_t launch_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* t_vals,
                            const float* dirs, int64_t n_rays, int S, int white_bkgd, const ActParams& ap, float* comp_rgb,
                            float* acc, float* depth, float* weights, hipStream_t stream);
hipError_t launch_sample_pdf(const float* bins, const float* weights, int64_t w_stride, const float* t_coarse,
                             const float* u, int64_t u_stride, int64_t n_rays, float* samples, float* t_fine,
                             hipStream_t stream);
hipError_t launch_composite_pdf(const float* raw, const float* t_coarse, const float* dirs, int64_t n_rays, int white_bkgd, const ActParams& ap,
                                const float* u, int64_t u_stride, float* comp_rgb, float* acc, float* depth, float* weights,
                                float* t_fine, hipStream_t stream);
hipError_t launch_sample_pdf_n(const float* bins, const float* weights, int64_t w_stride, const float* t_coarse, const float* u,
                               int64_t u_stride, int64_t n_rays, int nb, int nf, int nt, float* samples, float* t_fine, hipStream_t stream);
int64_t sample_pdf_n_lds_bytes(int nb, int nf, int nt, int* P_out);
}  // namespace aon

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char* msg) {
  std::snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check(hipError_t e, const char* where) {
  if (e == hipSuccess) return AON_OK;
  std::snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return AON_E_HIP_BASE - (int)e;
}

constexpr int kSc = 65, kSf = 193;

// round 5: the articulated calls take a packed stream AND a per-call block; both carry a form (bottleneck folded / literal, aon_fold.h)
const char* kFormsMsg = "two of the packed buffers of this call were made in different forms (aon_set_bottleneck_fold changed between the pack / prepare calls), "
                        "or one of them was never packed or declared by this process (a copy: aon_declare_stream_form)";
const char* kNullBwdMsg = "packed_bwd is NULL (= literal planes) while the process default is the folded form: pass the transposed stream the chain ran with";
// (an unknown pointer -- a copy nobody declared -- differs from everything, itself included)
bool forms_differ(const void* a, const void* b) {
  return a && b && (aon::stream_form(a) != aon::stream_form(b) || aon::stream_form(a) == aon::kFormUnknown);
}

// Optional live timing of the path's kernels with HIP events on the LAUNCH stream (torch.cuda.Event would only see torch's
// current stream), by kernel class; bench.py turns the totals into roofline figures.  Off unless aon_profile_begin() was
// called: one mutex-guarded branch per launch otherwise.
enum KClass { kMlpFwd = AON_PROF_MLP_FWD, kBwdChain = AON_PROF_BWD_CHAIN, kWgrad = AON_PROF_WGRAD, kComposite = AON_PROF_COMPOSITE,
              kSamplePdf = AON_PROF_SAMPLE_PDF, kCompositeBwd = AON_PROF_COMPOSITE_BWD, kCompositePdf = AON_PROF_COMPOSITE_PDF, kSampleT = AON_PROF_SAMPLE_T,
              kNumClasses = AON_PROF_NUM_CLASSES };

struct Profiler {
  std::mutex mu;
  bool on = false;
  struct Rec { int cls; hipEvent_t start, stop; };
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  std::vector<Rec> recs;
  int64_t units[kNumClasses] = {};
  // totals of the last completed aon_profile_end()
  double last_ms[kNumClasses] = {};
  int64_t last_launches[kNumClasses] = {};
  int64_t last_units[kNumClasses] = {};
} g_prof;

struct KTimer {
  hipEvent_t stop = nullptr;
  hipStream_t stream;
  KTimer(int cls, hipStream_t s, int64_t units) : stream(s) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.on) return;
    while (g_prof.used + 2 > g_prof.pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      g_prof.pool.push_back(e);
    }
    hipEvent_t start = g_prof.pool[g_prof.used];
    stop = g_prof.pool[g_prof.used + 1];
    g_prof.used += 2;
    g_prof.recs.push_back({cls, start, stop});
    g_prof.units[cls] += units;
    (void)hipEventRecord(start, stream);
  }
  ~KTimer() { if (stop) (void)hipEventRecord(stop, stream); }
};
struct MlpTimer : KTimer {
  MlpTimer(hipStream_t s, int64_t samples) : KTimer(kMlpFwd, s, samples) {}
};

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// The sampler / activation geometry of one call: aon_render_opts resolved against the reference's defaults.
struct Geo {
  int Sc, Sf, nf;          // t values of level 0 / level 1, draws of the inverse CDF
  bool default_sizes;      // 65 / 193: the specialised per-ray kernels apply
  int lindisp; float inv_near, inv_far;
  const float* noise[2]; float noise_std;
  float rgb_scale, rgb_shift, sigma_bias;
  int min_deg, max_deg, deg_view;   // encoding degrees of a vanilla network on the fused kernels
  bool other_degrees;               // != (0, 10, 4): encodings computed outside the MLP kernel, in its padded 63 / 27 layout
  int S(int l) const { return l == 0 ? Sc : Sf; }
  aon::ActParams act(bool art, int level, int64_t ray0) const {
    aon::ActParams ap{art ? AON_ACT_ARTICULATED : AON_ACT_VANILLA, rgb_scale, rgb_shift, sigma_bias, nullptr, noise_std};
    if (noise[level] && noise_std > 0.f) ap.noise = noise[level] + ray0 * S(level);
    return ap;
  }
};

// returns nullptr when fine, else what is wrong
// `general_engine`: the aon_grender_* entry points take the network's degrees from aon_mlp_geometry and ignore the degree fields of
// aon_render_opts, so those fields are neither validated against the fused kernels' limits nor turned into `other_degrees` there (a C
// caller who fills them to match a (1, 12, 5) geometry used to get AON_E_INVALID from the general engine: ADVICE r3).
const char* make_geo(const aon_render_opts* o, Geo& g, bool general_engine = false) {
  aon_render_opts d;
  aon_render_opts_init(&d);
  if (o) d = *o;
  if (d.num_coarse_samples < 2 || d.num_coarse_samples > 1023) return "num_coarse_samples must be in [2, 1023]";
  if (d.num_fine_samples < 1) return "num_fine_samples must be >= 1";
  g.Sc = d.num_coarse_samples + 1; g.nf = d.num_fine_samples; g.Sf = g.Sc + g.nf;
  g.default_sizes = g.Sc == kSc && g.Sf == kSf;
  if (!g.default_sizes && aon::sample_pdf_n_lds_bytes(g.Sc - 1, g.nf, g.Sc, nullptr) > 64 * 1024) return "num_coarse_samples / num_fine_samples too large for the per-ray LDS image";
  g.lindisp = d.lindisp != 0; g.inv_near = d.inv_near; g.inv_far = d.inv_far;
  g.noise[0] = d.noise_c; g.noise[1] = d.noise_f; g.noise_std = d.noise_std;
  g.rgb_scale = d.rgb_scale; g.rgb_shift = d.rgb_shift; g.sigma_bias = d.sigma_bias;
  g.min_deg = d.min_deg_point; g.max_deg = d.max_deg_point; g.deg_view = d.deg_view;
  if (general_engine) { g.min_deg = 0; g.max_deg = 10; g.deg_view = 4; g.other_degrees = false; return nullptr; }
  if (g.max_deg < g.min_deg || g.max_deg - g.min_deg > 10 || g.deg_view < 0 || g.deg_view > 4)
    return "the fused kernels hold up to 10 position and 4 view frequency levels (other degrees: aon_grender_*)";
  g.other_degrees = !(g.min_deg == 0 && g.max_deg == 10 && g.deg_view == 4);
  return nullptr;
}

// workspace layout for a chunk of n rays
struct Ws {
  float* t_c;   // n*Sc
  float* w_c;   // n*Sc
  float* t_f;   // n*Sf
  float* raw;   // n*Sf*4 (coarse raw uses the first n*Sc*4)
  float* coords; float* enc; float* venc;   // other_degrees only: n*Sf*3, n*Sf*63, n*27
  float* vbias;   // n*128: the level's per-ray view bias (vanilla, folded form; launch_view_bias) -- both levels in turn
  int64_t bytes;
};

Ws carve(char* base, int64_t n, const Geo& g) {
  Ws w{};
  int64_t off = 0;
  w.t_c = reinterpret_cast<float*>(base + off); off += align_up(n * g.Sc * 4, 256);
  w.w_c = reinterpret_cast<float*>(base + off); off += align_up(n * g.Sc * 4, 256);
  w.t_f = reinterpret_cast<float*>(base + off); off += align_up(n * g.Sf * 4, 256);
  w.raw = reinterpret_cast<float*>(base + off); off += align_up(n * g.Sf * 16, 256);
  if (g.other_degrees) {
    w.coords = reinterpret_cast<float*>(base + off); off += align_up(n * g.Sf * 12, 256);
    w.enc = reinterpret_cast<float*>(base + off); off += align_up(n * g.Sf * (int64_t)aon::kPosEnc * 4, 256);
    w.venc = reinterpret_cast<float*>(base + off); off += align_up(n * (int64_t)aon::kViewEnc * 4, 256);
  } else {
    w.vbias = reinterpret_cast<float*>(base + off); off += align_up(n * (int64_t)aon::kCondWidth * 4, 256);
  }
  w.bytes = off;
  return w;
}

}  // namespace

extern "C" {

int aon_abi_version(void) { return AON_ABI_VERSION; }

const char* aon_last_error(void) { return g_err; }

int aon_raygen(const float* c2w_host, int H, int W, float focal, int64_t pix_begin, int64_t pix_end, float* rays_o,
               float* viewdirs, float* rays_d, void* stream) {
  if (!c2w_host || !rays_o || !viewdirs) return fail(AON_E_INVALID, "aon_raygen: null pointer");
  if (H <= 0 || W <= 0 || !(focal > 0.f) || pix_begin < 0 || pix_end < pix_begin || pix_end > (int64_t)H * W)
    return fail(AON_E_INVALID, "aon_raygen: bad geometry");
  return check(aon::launch_raygen(c2w_host, H, W, focal, nullptr, pix_begin, pix_end, rays_o, viewdirs, rays_d,
                                  (hipStream_t)stream), "aon_raygen");
}

int aon_train_loss_fwd(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n, const float* const* latents_host, const int* latent_len_host,
                       float reg_scale, float* stats, float* loss, void* stream) {
  if (!rgb_fine || !target || !stats || !loss) return fail(AON_E_INVALID, "aon_train_loss_fwd: null pointer");
  if (n <= 0) return fail(AON_E_INVALID, "aon_train_loss_fwd: bad size");
  for (int k = 0; k < 3 && latents_host; ++k)
    if (latents_host[k] && (!latent_len_host || latent_len_host[k] <= 0)) return fail(AON_E_INVALID, "aon_train_loss_fwd: bad latent length");
  return check(aon::launch_train_loss(false, rgb_coarse, rgb_fine, target, n, latents_host, latent_len_host, reg_scale, stats, loss, nullptr, nullptr, nullptr, nullptr,
                                      (hipStream_t)stream), "aon_train_loss_fwd");
}

int aon_train_loss_bwd(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n, const float* const* latents_host, const int* latent_len_host,
                       float reg_scale, const float* grad_loss, float* d_rgb_coarse, float* d_rgb_fine, float* const* d_latents_host, void* stream) {
  if (!rgb_fine || !target || !grad_loss || !d_rgb_fine) return fail(AON_E_INVALID, "aon_train_loss_bwd: null pointer");
  if (n <= 0) return fail(AON_E_INVALID, "aon_train_loss_bwd: bad size");
  if ((rgb_coarse == nullptr) != (d_rgb_coarse == nullptr)) return fail(AON_E_INVALID, "aon_train_loss_bwd: coarse level input / gradient mismatch");
  for (int k = 0; k < 3 && latents_host; ++k)
    if (latents_host[k] && (!latent_len_host || latent_len_host[k] <= 0)) return fail(AON_E_INVALID, "aon_train_loss_bwd: bad latent length");
  return check(aon::launch_train_loss(true, rgb_coarse, rgb_fine, target, n, latents_host, latent_len_host, reg_scale, nullptr, nullptr, grad_loss, d_rgb_coarse, d_rgb_fine,
                                      d_latents_host, (hipStream_t)stream), "aon_train_loss_bwd");
}

int aon_view_bias(const void* packed, const float* viewdirs, int64_t n_rays, float* view_bias, void* stream) {
  if (n_rays < 0) return fail(AON_E_INVALID, "aon_view_bias: bad size");
  if (n_rays == 0) return AON_OK;
  if (!packed || !viewdirs || !view_bias) return fail(AON_E_INVALID, "aon_view_bias: null pointer");
  if (aon::stream_form(packed) != aon::kFormFolded) return fail(AON_E_INVALID, "aon_view_bias: the stream was packed in the literal form (aon_set_bottleneck_fold)");
  return check(aon::launch_view_bias(static_cast<const char*>(packed), viewdirs, n_rays, view_bias, (hipStream_t)stream), "aon_view_bias");
}

int aon_ray_directions(int H, int W, float focal, float* directions, void* stream) {
  if (H <= 0 || W <= 0 || !(focal > 0.f) || !directions) return fail(AON_E_INVALID, "aon_ray_directions: bad argument");
  return check(aon::launch_ray_directions(H, W, focal, directions, (hipStream_t)stream), "aon_ray_directions");
}

int aon_get_rays(const float* directions, const float* c2w_host, int64_t n, float* rays_o, float* viewdirs, float* rays_d,
                 void* stream) {
  if (n < 0) return fail(AON_E_INVALID, "aon_get_rays: bad size");
  if (n == 0) return AON_OK;
  if (!directions || !c2w_host || !rays_o || !viewdirs) return fail(AON_E_INVALID, "aon_get_rays: null pointer");
  // geometry arguments are unused when directions are supplied; W = n keeps the pixel index math in range
  return check(aon::launch_raygen(c2w_host, 1, (int)(n > INT32_MAX ? INT32_MAX : n), 1.0f, directions, 0, n, rays_o, viewdirs,
                                  rays_d, (hipStream_t)stream), "aon_get_rays");
}

int aon_ray_radii(const float* directions, const float* c2w_host, int H, int W, float* radii, void* stream) {
  if (!directions || !c2w_host || !radii) return fail(AON_E_INVALID, "aon_ray_radii: null pointer");
  if (H < 3 || W < 1) return fail(AON_E_INVALID, "aon_ray_radii: needs H >= 3 image rows (ray_utils.py:141 indexes dx[-2])");
  return check(aon::launch_ray_radii(directions, c2w_host, H, W, radii, (hipStream_t)stream), "aon_ray_radii");
}

int aon_cast_rays(const float* t_vals, const float* origins, const float* directions, int64_t n_rays, int S, float* coords,
                  void* stream) {
  if (n_rays < 0 || S < 1) return fail(AON_E_INVALID, "aon_cast_rays: bad size");
  if (n_rays == 0) return AON_OK;
  if (!t_vals || !origins || !directions || !coords) return fail(AON_E_INVALID, "aon_cast_rays: null pointer");
  return check(aon::launch_cast_rays(t_vals, origins, directions, n_rays, S, coords, (hipStream_t)stream), "aon_cast_rays");
}

int aon_sample_along_rays(const float* rays_o, const float* rays_d, int64_t n_rays, int S, float near_, float far_,
                          const float* t_rand, float* t_vals, float* coords, void* stream) {
  if (n_rays < 0 || S < 2) return fail(AON_E_INVALID, "aon_sample_along_rays: bad size");
  if (n_rays == 0) return AON_OK;
  if (!t_vals || (coords && (!rays_o || !rays_d))) return fail(AON_E_INVALID, "aon_sample_along_rays: null pointer");
  return check(aon::launch_sample_along_rays(rays_o, rays_d, n_rays, S, near_, far_, t_rand, t_vals, coords, (hipStream_t)stream),
               "aon_sample_along_rays");
}

int aon_pos_enc(const float* x, int64_t n, int min_deg, int max_deg, float* out, void* stream) {
  if (n < 0 || max_deg < min_deg) return fail(AON_E_INVALID, "aon_pos_enc: bad size");
  if (n == 0) return AON_OK;
  if (!x || !out) return fail(AON_E_INVALID, "aon_pos_enc: null pointer");
  return check(aon::launch_pos_enc(x, n, min_deg, max_deg, out, (hipStream_t)stream), "aon_pos_enc");
}

int64_t aon_mlp_packed_bytes(void) { return aon::kPackedBytes; }

int aon_pack_vanilla_mlp(const float* const* params_host, void* packed, void* stream) {
  if (!params_host || !packed) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp: null pointer");
  for (int i = 0; i < aon::kNumVanillaParams; ++i)
    if (!params_host[i]) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp: null parameter pointer");
  if (reinterpret_cast<uintptr_t>(packed) & 15) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp: packed must be 16-byte aligned");
  return check(aon::launch_pack_vanilla(params_host, static_cast<float*>(packed), (hipStream_t)stream), "aon_pack_vanilla_mlp");
}

int aon_pack_vanilla_mlp_deg(const float* const* params_host, int min_deg_point, int max_deg_point, int deg_view, void* packed, void* stream) {
  if (!params_host || !packed) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_deg: null pointer");
  for (int i = 0; i < aon::kNumVanillaParams; ++i)
    if (!params_host[i]) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_deg: null parameter pointer");
  if (reinterpret_cast<uintptr_t>(packed) & 15) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_deg: packed must be 16-byte aligned");
  const int L = max_deg_point - min_deg_point;
  if (L < 0 || L > 10 || deg_view < 0 || deg_view > 4)
    return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_deg: the stream holds up to 10 position and 4 view frequency levels");
  return check(aon::launch_pack_vanilla(params_host, static_cast<float*>(packed), (hipStream_t)stream, L, deg_view), "aon_pack_vanilla_mlp_deg");
}

int aon_mlp_fwd(const void* packed, const float* rays_o, const float* rays_d, const float* viewdirs, const float* t_vals,
                int64_t n_rays, int S, float* raw, void* stream) {
  if (n_rays < 0 || S < 1) return fail(AON_E_INVALID, "aon_mlp_fwd: bad size");
  if (n_rays == 0) return AON_OK;
  if (!packed || !rays_o || !rays_d || !viewdirs || !t_vals || !raw) return fail(AON_E_INVALID, "aon_mlp_fwd: null pointer");
  if (n_rays * (int64_t)S > (int64_t)INT32_MAX * 64) return fail(AON_E_INVALID, "aon_mlp_fwd: too many samples for one call");
  MlpTimer timer((hipStream_t)stream, n_rays * S);
  return check(aon::launch_mlp_fwd(static_cast<const char*>(packed), rays_o, rays_d, viewdirs, t_vals, n_rays, S, raw,
                                   (hipStream_t)stream), "aon_mlp_fwd");
}

int aon_mlp_fwd_enc(const void* packed, const float* samples_enc, const float* viewdirs_enc, int64_t n_rays, int S,
                    float* raw, void* stream) {
  if (n_rays < 0 || S < 1) return fail(AON_E_INVALID, "aon_mlp_fwd_enc: bad size");
  if (n_rays == 0) return AON_OK;
  if (!packed || !samples_enc || !viewdirs_enc || !raw) return fail(AON_E_INVALID, "aon_mlp_fwd_enc: null pointer");
  return check(aon::launch_mlp_fwd_enc(static_cast<const char*>(packed), samples_enc, viewdirs_enc, n_rays, S, raw,
                                       (hipStream_t)stream), "aon_mlp_fwd_enc");
}

int aon_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* t_vals,
                  const float* dirs, int64_t n_rays, int S, int white_bkgd, int act, float* comp_rgb, float* acc,
                  float* depth, float* weights, void* stream) {
  if (n_rays < 0 || S < 1 || rgb_stride < 3 || sigma_stride < 1 || act < 0 || act > 2)
    return fail(AON_E_INVALID, "aon_composite: bad size / stride / act");
  if (n_rays == 0) return AON_OK;
  if (!rgb || !sigma || !t_vals || !dirs || !comp_rgb || !acc || !depth) return fail(AON_E_INVALID, "aon_composite: null pointer");
  KTimer timer(kComposite, (hipStream_t)stream, n_rays);
  return check(aon::launch_composite(rgb, rgb_stride, sigma, sigma_stride, t_vals, dirs, n_rays, S, white_bkgd, aon::default_act(act), comp_rgb,
                                     acc, depth, weights, (hipStream_t)stream), "aon_composite");
}

void aon_render_opts_init(aon_render_opts* o) {
  if (!o) return;
  o->num_coarse_samples = 64; o->num_fine_samples = 128; o->lindisp = 0; o->inv_near = 0.f; o->inv_far = 0.f;
  o->noise_std = 0.f; o->noise_c = nullptr; o->noise_f = nullptr;
  o->rgb_scale = 1.002f; o->rgb_shift = 0.001f; o->sigma_bias = -1.0f;
  o->min_deg_point = 0; o->max_deg_point = 10; o->deg_view = 4;
}

int aon_sample_along_rays_ex(const float* rays_o, const float* rays_d, int64_t n_rays, int S, float near_, float far_, int lindisp,
                             float inv_near, float inv_far, const float* t_rand, float* t_vals, float* coords, void* stream) {
  if (n_rays < 0 || S < 2) return fail(AON_E_INVALID, "aon_sample_along_rays_ex: bad size");
  if (n_rays == 0) return AON_OK;
  if (!t_vals || (coords && (!rays_o || !rays_d))) return fail(AON_E_INVALID, "aon_sample_along_rays_ex: null pointer");
  return check(aon::launch_sample_along_rays(rays_o, rays_d, n_rays, S, near_, far_, t_rand, t_vals, coords, (hipStream_t)stream,
                                             lindisp != 0, inv_near, inv_far), "aon_sample_along_rays_ex");
}

int aon_composite_ex(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* t_vals, const float* dirs,
                     int64_t n_rays, int S, int white_bkgd, int act, const aon_render_opts* opts, float* comp_rgb, float* acc,
                     float* depth, float* weights, void* stream) {
  if (n_rays < 0 || S < 1 || rgb_stride < 3 || sigma_stride < 1 || act < 0 || act > 2)
    return fail(AON_E_INVALID, "aon_composite_ex: bad size / stride / act");
  if (n_rays == 0) return AON_OK;
  if (!rgb || !sigma || !t_vals || !dirs || !comp_rgb || !acc || !depth) return fail(AON_E_INVALID, "aon_composite_ex: null pointer");
  aon::ActParams ap = aon::default_act(act);
  if (opts) {
    ap.rgb_scale = opts->rgb_scale; ap.rgb_shift = opts->rgb_shift; ap.sigma_bias = opts->sigma_bias;
    if (opts->noise_c && opts->noise_std > 0.f) { ap.noise = opts->noise_c; ap.noise_std = opts->noise_std; }
  }
  KTimer timer(kComposite, (hipStream_t)stream, n_rays);
  return check(aon::launch_composite(rgb, rgb_stride, sigma, sigma_stride, t_vals, dirs, n_rays, S, white_bkgd, ap, comp_rgb, acc, depth,
                                     weights, (hipStream_t)stream), "aon_composite_ex");
}

int aon_sample_pdf_n(const float* bins, const float* weights, int64_t w_stride, const float* t_coarse, const float* u,
                     int64_t u_stride, int64_t n_rays, int num_bins, int num_samples, int num_t, float* samples, float* t_fine,
                     void* stream) {
  if (n_rays < 0 || num_bins < 2 || num_samples < 1 || num_t < 0 || w_stride < num_bins - 1 || (u_stride != 0 && u_stride < num_samples))
    return fail(AON_E_INVALID, "aon_sample_pdf_n: bad size / stride");
  if (!bins && num_t != num_bins + 1) return fail(AON_E_INVALID, "aon_sample_pdf_n: bins == NULL needs num_t == num_bins + 1 (mid-points of t_coarse)");
  if (aon::sample_pdf_n_lds_bytes(num_bins, num_samples, num_t, nullptr) > 64 * 1024) return fail(AON_E_INVALID, "aon_sample_pdf_n: sizes exceed the per-ray LDS image");
  if (n_rays == 0) return AON_OK;
  if (!weights || !u || (!bins && !t_coarse) || (t_fine && !t_coarse) || (!samples && !t_fine))
    return fail(AON_E_INVALID, "aon_sample_pdf_n: null pointer");
  KTimer timer(kSamplePdf, (hipStream_t)stream, n_rays);
  return check(aon::launch_sample_pdf_n(bins, weights, w_stride, t_coarse, u, u_stride, n_rays, num_bins, num_samples, t_coarse ? num_t : 0,
                                        samples, t_fine, (hipStream_t)stream), "aon_sample_pdf_n");
}

int aon_sample_pdf(const float* bins, const float* weights, int64_t w_stride, const float* t_coarse, const float* u,
                   int64_t u_stride, int64_t n_rays, float* samples, float* t_fine, void* stream) {
  if (n_rays < 0 || w_stride < 63 || (u_stride != 0 && u_stride < 128)) return fail(AON_E_INVALID, "aon_sample_pdf: bad size / stride");
  if (n_rays == 0) return AON_OK;
  if (!weights || !u || (!bins && !t_coarse) || (t_fine && !t_coarse) || (!samples && !t_fine))
    return fail(AON_E_INVALID, "aon_sample_pdf: null pointer");
  KTimer timer(kSamplePdf, (hipStream_t)stream, n_rays);
  return check(aon::launch_sample_pdf(bins, weights, w_stride, t_coarse, u, u_stride, n_rays, samples, t_fine,
                                      (hipStream_t)stream), "aon_sample_pdf");
}

// The coarse level's compositing and the fine level's sampling as ONE kernel (model.py:160-173): the whole-path entry points
// use it; aon_set_coarse_fusion(0) puts them back on the two stage kernels (A/B measurements, equality tests).
static std::atomic<int> g_fuse_coarse{1};   // process-wide switch; every call reads it ONCE
int aon_set_coarse_fusion(int on) {
  g_fuse_coarse.store(on ? 1 : 0, std::memory_order_relaxed);
  return AON_OK;
}

int aon_composite_pdf(const float* raw, const float* t_coarse, const float* dirs, int64_t n_rays, int white_bkgd, int act,
                      const float* u, int64_t u_stride, float* comp_rgb, float* acc, float* depth, float* weights, float* t_fine,
                      void* stream) {
  if (n_rays < 0 || act < 0 || act > 2 || (u_stride != 0 && u_stride < 128)) return fail(AON_E_INVALID, "aon_composite_pdf: bad size / stride / act");
  if (n_rays == 0) return AON_OK;
  if (!raw || !t_coarse || !dirs || !u || !comp_rgb || !acc || !depth || !t_fine) return fail(AON_E_INVALID, "aon_composite_pdf: null pointer");
  if (reinterpret_cast<uintptr_t>(raw) & 15) return fail(AON_E_INVALID, "aon_composite_pdf: raw must be 16-byte aligned");
  KTimer timer(kCompositePdf, (hipStream_t)stream, n_rays);
  return check(aon::launch_composite_pdf(raw, t_coarse, dirs, n_rays, white_bkgd, aon::default_act(act), u, u_stride, comp_rgb, acc, depth, weights,
                                         t_fine, (hipStream_t)stream), "aon_composite_pdf");
}

// ---- training (R14) ----
int64_t aon_train_plane_rows(void) { return aon::kPlRows; }
int64_t aon_bwd_packed_bytes(void) { return aon::bwd_stream_bytes(); }
int64_t aon_wgrad_workspace_bytes(void) { return aon::wgrad_workspace_bytes(); }

int aon_wgrad_plan(int articulated, int64_t Np, int cus, int32_t* jobs6, int max_jobs, int64_t* ws_bytes) {
  if (!jobs6 || max_jobs < 1) return fail(AON_E_INVALID, "aon_wgrad_plan: null / empty output");
  const int n = aon::wgrad_plan_describe(articulated != 0, Np, cus, jobs6, max_jobs, ws_bytes);
  if (n < 0) return fail(AON_E_INVALID, n == -1 ? "aon_wgrad_plan: Np must be a positive multiple of 32, cus >= 1"
                                       : n == -2 ? "aon_wgrad_plan: no plan" : "aon_wgrad_plan: max_jobs too small");
  return n;
}

int aon_wgrad_plan_segment(int articulated, int64_t Np, int cus, int job, int workgroup, int32_t* begin_end) {
  if (!begin_end) return fail(AON_E_INVALID, "aon_wgrad_plan_segment: null output");
  const int r = aon::wgrad_plan_segment(articulated != 0, Np, cus, job, workgroup, begin_end);
  if (r < 0) return fail(AON_E_INVALID, "aon_wgrad_plan_segment: bad argument");
  return r;
}

int aon_set_wgrad_probe(void* device_buffer) {
  aon::set_wgrad_probe(static_cast<long long*>(device_buffer));
  return AON_OK;
}

int aon_wgrad_kind_bench(int kind, int nlayers, const float* planes, const float* dplanes, int rows, int64_t Np, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  if (!planes || !dplanes || !workspace || rows < 512 || (rows & 31) || Np <= 0 || (Np & 127))
    return fail(AON_E_INVALID, "aon_wgrad_kind_bench: bad argument");
  if (workspace_bytes < aon::wgrad_workspace_bytes()) return fail(AON_E_WORKSPACE, "aon_wgrad_kind_bench: workspace too small");
  KTimer timer(kWgrad, (hipStream_t)stream, Np);
  return check(aon::launch_wgrad_kind_bench(kind, nlayers, planes, dplanes, rows, Np, static_cast<float*>(workspace), nullptr,
                                            (hipStream_t)stream), "aon_wgrad_kind_bench");
}

int aon_pack_vanilla_mlp_bwd(const float* const* params_host, void* packed_bwd, void* stream) {
  if (!params_host || !packed_bwd) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_bwd: null pointer");
  for (int i = 0; i < aon::kNumVanillaParams; ++i)
    if (!params_host[i]) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_bwd: null parameter pointer");
  if (reinterpret_cast<uintptr_t>(packed_bwd) & 15) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_bwd: buffer must be 16-byte aligned");
  return check(aon::launch_pack_vanilla_bwd(params_host, static_cast<float*>(packed_bwd), (hipStream_t)stream), "aon_pack_vanilla_mlp_bwd");
}

int aon_pack_vanilla_mlp_bwd_deg(const float* const* params_host, int min_deg_point, int max_deg_point, int deg_view, void* packed_bwd, void* stream) {
  if (!params_host || !packed_bwd) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_bwd_deg: null pointer");
  for (int i = 0; i < aon::kNumVanillaParams; ++i)
    if (!params_host[i]) return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_bwd_deg: null parameter pointer");
  const int L = max_deg_point - min_deg_point;
  if (L < 0 || L > 10 || deg_view < 0 || deg_view > 4)
    return fail(AON_E_INVALID, "aon_pack_vanilla_mlp_bwd_deg: up to 10 position and 4 view frequency levels");
  return check(aon::launch_pack_vanilla_bwd(params_host, static_cast<float*>(packed_bwd), (hipStream_t)stream, 3 + 6 * L, 3 + 6 * deg_view),
               "aon_pack_vanilla_mlp_bwd_deg");
}

// Round 6: everything a training step of a TWO-level vanilla model packs, in one call -- both networks' forward and transposed streams --
// with the eight fp64 fold products (W' and b' of each network, once for its forward and once for its transposed stream) as ONE launch in
// front instead of four launches of 13-16 us in a row with their pack kernels.  The same kernels on the same operands as the four separate
// calls: same bytes in every buffer.  packed_bwd_* may be NULL (no backward wanted).
int aon_vanilla_pack_step(const float* const* params_coarse_host, const float* const* params_fine_host, int min_deg_point, int max_deg_point, int deg_view,
                          void* packed_coarse, void* packed_bwd_coarse, void* packed_fine, void* packed_bwd_fine, void* stream_) {
  if (!params_coarse_host || !params_fine_host || !packed_coarse || !packed_fine) return fail(AON_E_INVALID, "aon_vanilla_pack_step: null pointer");
  for (int i = 0; i < aon::kNumVanillaParams; ++i)
    if (!params_coarse_host[i] || !params_fine_host[i]) return fail(AON_E_INVALID, "aon_vanilla_pack_step: null parameter pointer");
  for (const void* p : {(const void*)packed_coarse, (const void*)packed_fine, (const void*)packed_bwd_coarse, (const void*)packed_bwd_fine})
    if (reinterpret_cast<uintptr_t>(p) & 15) return fail(AON_E_INVALID, "aon_vanilla_pack_step: buffers must be 16-byte aligned");
  const int L = max_deg_point - min_deg_point;
  if (L < 0 || L > 10 || deg_view < 0 || deg_view > 4)
    return fail(AON_E_INVALID, "aon_vanilla_pack_step: the streams hold up to 10 position and 4 view frequency levels");
  hipStream_t stream = (hipStream_t)stream_;
  const float* const* P[2] = {params_coarse_host, params_fine_host};
  float* fwd[2] = {static_cast<float*>(packed_coarse), static_cast<float*>(packed_fine)};
  float* bwd[2] = {static_cast<float*>(packed_bwd_coarse), static_cast<float*>(packed_bwd_fine)};
  const bool folded = aon::fold_default() == aon::kFormFolded;
  if (folded) {
    aon::FoldGemm jobs[8];
    int n = 0;
    for (int l = 0; l < 2; ++l) { aon::vanilla_fold_jobs_fwd(P[l], fwd[l], deg_view, jobs + n); n += 2; }
    for (int l = 0; l < 2; ++l)
      if (bwd[l]) { aon::vanilla_fold_jobs_bwd(P[l], bwd[l], 3 + 6 * deg_view, jobs + n); n += 2; }
    if (int rc = check(aon::launch_fold_gemms(jobs, n, stream), "aon_vanilla_pack_step")) return rc;
  }
  for (int l = 0; l < 2; ++l)
    if (int rc = check(aon::launch_pack_vanilla(P[l], fwd[l], stream, L, deg_view, folded), "aon_vanilla_pack_step")) return rc;
  for (int l = 0; l < 2; ++l)
    if (bwd[l])
      if (int rc = check(aon::launch_pack_vanilla_bwd(P[l], bwd[l], stream, 3 + 6 * L, 3 + 6 * deg_view, folded), "aon_vanilla_pack_step")) return rc;
  return AON_OK;
}

int64_t aon_train_mask_bytes(int64_t Np) { return (int64_t)aon::kMaskLayers * Np * 2 * 16; }

int aon_mlp_fwd_train(const void* packed, const float* rays_o, const float* rays_d, const float* viewdirs, const float* t_vals,
                      int64_t n_rays, int S, float* raw, float* planes, void* masks, void* stream) {
  if (n_rays < 0 || S < 1) return fail(AON_E_INVALID, "aon_mlp_fwd_train: bad size");
  if (n_rays == 0) return AON_OK;
  if (!packed || !rays_o || !rays_d || !viewdirs || !t_vals || !raw || !planes || !masks)
    return fail(AON_E_INVALID, "aon_mlp_fwd_train: null pointer");
  if (reinterpret_cast<uintptr_t>(masks) & 15) return fail(AON_E_INVALID, "aon_mlp_fwd_train: masks must be 16-byte aligned");
  MlpTimer timer((hipStream_t)stream, n_rays * S);
  return check(aon::launch_mlp_fwd_train(static_cast<const char*>(packed), rays_o, rays_d, viewdirs, t_vals, n_rays, S, raw, planes,
                                         masks, (hipStream_t)stream), "aon_mlp_fwd_train");
}

int aon_composite_bwd(const float* raw, const float* t_vals, const float* dirs, const float* g_rgb, const float* g_acc,
                      const float* g_depth, int64_t n_rays, int S, int white_bkgd, int act, float* d_raw, void* stream) {
  if (n_rays < 0 || S < 1 || S > 512 || act < 0 || act > 2) return fail(AON_E_INVALID, "aon_composite_bwd: bad size / act (S <= 512)");
  if (n_rays == 0) return AON_OK;
  if (!raw || !t_vals || !dirs || !g_rgb || !d_raw) return fail(AON_E_INVALID, "aon_composite_bwd: null pointer");
  KTimer timer(kCompositeBwd, (hipStream_t)stream, n_rays);
  return check(aon::launch_composite_bwd(raw, t_vals, dirs, g_rgb, g_acc, g_depth, n_rays, S, white_bkgd, aon::default_act(act), d_raw,
                                         (hipStream_t)stream), "aon_composite_bwd");
}

int aon_mlp_bwd_chain(const void* packed_bwd, const void* packed_fwd, const float* d_raw, const void* masks, float* dplanes,
                      int64_t Np, void* stream) {
  if (Np < 0 || (Np & 127)) return fail(AON_E_INVALID, "aon_mlp_bwd_chain: Np must be a multiple of 128");
  if (Np == 0) return AON_OK;
  if (!packed_bwd || !packed_fwd || !d_raw || !masks || !dplanes) return fail(AON_E_INVALID, "aon_mlp_bwd_chain: null pointer");
  if (forms_differ(packed_bwd, packed_fwd)) return fail(AON_E_INVALID, kFormsMsg);   // (the masks / planes came from packed_fwd's kernel: ADVICE r5)
  KTimer timer(kBwdChain, (hipStream_t)stream, Np);
  return check(aon::launch_mlp_bwd_chain(static_cast<const char*>(packed_bwd), static_cast<const char*>(packed_fwd), d_raw, masks,
                                         dplanes, Np, (hipStream_t)stream), "aon_mlp_bwd_chain");
}

int aon_vanilla_wgrad(const float* planes, const float* dplanes, const float* d_raw, int64_t Np, float* const* grads_host,
                      void* workspace, int64_t workspace_bytes, void* stream, const void* packed_bwd) {
  if (Np <= 0 || (Np & 127)) return fail(AON_E_INVALID, "aon_vanilla_wgrad: Np must be a positive multiple of 128");
  if (!planes || !dplanes || !d_raw || !grads_host || !workspace) return fail(AON_E_INVALID, "aon_vanilla_wgrad: null pointer");
  for (int i = 0; i < aon::kNumVanillaParams; ++i)
    if (!grads_host[i]) return fail(AON_E_INVALID, "aon_vanilla_wgrad: null gradient pointer");
  if (workspace_bytes < aon::wgrad_workspace_bytes()) return fail(AON_E_WORKSPACE, "aon_vanilla_wgrad: workspace too small");
  if (!packed_bwd && aon::fold_default() == aon::kFormFolded) return fail(AON_E_INVALID, kNullBwdMsg);
  KTimer timer(kWgrad, (hipStream_t)stream, Np);
  return check(aon::launch_vanilla_wgrad(planes, dplanes, d_raw, Np, grads_host, static_cast<float*>(workspace), (hipStream_t)stream, nullptr, packed_bwd),
               "aon_vanilla_wgrad");
}

// ---- training, articulated network ----
int64_t aon_art_train_plane_rows(void) { return aon::kAPlRows; }
int64_t aon_art_train_mask_bytes(int64_t Np) { return (int64_t)aon::kAMaskLayers * Np * 2 * 16; }
int64_t aon_art_bwd_packed_bytes(void) { return aon::art_bwd_stream_bytes(); }

static const char* art_degrees_ok(int min_deg_point, int max_deg_point, int deg_view) {
  const int L = max_deg_point - min_deg_point;
  if (L < 0 || L > 10 || deg_view < 0 || deg_view > 4) return "up to 10 position and 4 view frequency levels";
  if (min_deg_point < -32 || max_deg_point > 32) return "min_deg_point / max_deg_point must lie in [-32, 32]";
  return nullptr;
}

int aon_pack_art_mlp_bwd(const float* const* params_host, void* packed_bwd, void* stream) {
  return aon_pack_art_mlp_bwd_deg(params_host, 0, 10, 4, packed_bwd, stream);
}
int aon_pack_art_mlp_bwd_deg(const float* const* params_host, int min_deg_point, int max_deg_point, int deg_view, void* packed_bwd, void* stream) {
  if (!params_host || !packed_bwd) return fail(AON_E_INVALID, "aon_pack_art_mlp_bwd: null pointer");
  for (int i = 0; i < 40; ++i)
    if (!params_host[i]) return fail(AON_E_INVALID, "aon_pack_art_mlp_bwd: null parameter pointer");
  if (reinterpret_cast<uintptr_t>(packed_bwd) & 15) return fail(AON_E_INVALID, "aon_pack_art_mlp_bwd: buffer must be 16-byte aligned");
  if (const char* bad = art_degrees_ok(min_deg_point, max_deg_point, deg_view)) return fail(AON_E_INVALID, bad);
  return check(aon::launch_pack_art_bwd(params_host, static_cast<float*>(packed_bwd), (hipStream_t)stream, max_deg_point - min_deg_point, deg_view),
               "aon_pack_art_mlp_bwd");
}

int aon_art_mlp_fwd_train(const void* packed, const void* small, const float* rays_o, const float* rays_d, const float* viewdirs,
                          const float* t_vals, int64_t n_rays, int S, float* raw, float* planes, void* masks, void* stream) {
  if (n_rays < 0 || S < 1) return fail(AON_E_INVALID, "aon_art_mlp_fwd_train: bad size");
  if (n_rays == 0) return AON_OK;
  if (!packed || !small || !rays_o || !rays_d || !viewdirs || !t_vals || !raw || !planes || !masks)
    return fail(AON_E_INVALID, "aon_art_mlp_fwd_train: null pointer");
  if (forms_differ(packed, small)) return fail(AON_E_INVALID, kFormsMsg);
  if (reinterpret_cast<uintptr_t>(masks) & 15) return fail(AON_E_INVALID, "aon_art_mlp_fwd_train: masks must be 16-byte aligned");
  MlpTimer timer((hipStream_t)stream, n_rays * S);
  return check(aon::launch_art_mlp_fwd_train(static_cast<const char*>(packed), static_cast<const float*>(small), rays_o, rays_d, viewdirs,
                                             t_vals, n_rays, S, raw, planes, masks, (hipStream_t)stream), "aon_art_mlp_fwd_train");
}

int aon_art_bwd_chain(const void* packed_bwd, const void* small, const float* d_raw, const void* masks, const float* planes,
                      float* dplanes, float* dxp, int64_t Np, void* stream) {
  if (Np < 0 || (Np & 127)) return fail(AON_E_INVALID, "aon_art_bwd_chain: Np must be a multiple of 128");
  if (Np == 0) return AON_OK;
  if (!packed_bwd || !small || !d_raw || !masks || !planes || !dplanes || !dxp) return fail(AON_E_INVALID, "aon_art_bwd_chain: null pointer");
  if (forms_differ(packed_bwd, small)) return fail(AON_E_INVALID, kFormsMsg);
  KTimer timer(kBwdChain, (hipStream_t)stream, Np);
  return check(aon::launch_art_bwd_chain(static_cast<const char*>(packed_bwd), static_cast<const float*>(small), d_raw, masks, planes,
                                         dplanes, dxp, Np, (hipStream_t)stream), "aon_art_bwd_chain");
}

int aon_art_wgrad(const float* planes, const float* dplanes, const float* d_raw, const float* dxp, int64_t Np,
                  const float* const* params_host, const float* shape, const float* appearance, const float* articulation,
                  float* const* grads_host, float* g_shape, float* g_appearance, float* g_articulation, void* workspace,
                  int64_t workspace_bytes, void* stream, const void* packed_bwd) {
  return aon_art_wgrad_deg(planes, dplanes, d_raw, dxp, Np, params_host, shape, appearance, articulation, grads_host, g_shape, g_appearance,
                           g_articulation, workspace, workspace_bytes, stream, 0, 10, 4, packed_bwd);
}
int aon_art_wgrad_deg(const float* planes, const float* dplanes, const float* d_raw, const float* dxp, int64_t Np,
                      const float* const* params_host, const float* shape, const float* appearance, const float* articulation,
                      float* const* grads_host, float* g_shape, float* g_appearance, float* g_articulation, void* workspace,
                      int64_t workspace_bytes, void* stream, int min_deg_point, int max_deg_point, int deg_view, const void* packed_bwd) {
  if (const char* bad = art_degrees_ok(min_deg_point, max_deg_point, deg_view)) return fail(AON_E_INVALID, bad);
  if (Np <= 0 || (Np & 127)) return fail(AON_E_INVALID, "aon_art_wgrad: Np must be a positive multiple of 128");
  if (!planes || !dplanes || !d_raw || !dxp || !params_host || !shape || !appearance || !articulation || !grads_host || !g_shape ||
      !g_appearance || !g_articulation || !workspace)
    return fail(AON_E_INVALID, "aon_art_wgrad: null pointer");
  for (int i = 0; i < 40; ++i)
    if (!params_host[i] || !grads_host[i]) return fail(AON_E_INVALID, "aon_art_wgrad: null parameter / gradient pointer");
  if (workspace_bytes < aon::wgrad_workspace_bytes()) return fail(AON_E_WORKSPACE, "aon_art_wgrad: workspace too small");
  if (!packed_bwd && aon::fold_default() == aon::kFormFolded) return fail(AON_E_INVALID, kNullBwdMsg);
  KTimer timer(kWgrad, (hipStream_t)stream, Np);
  return check(aon::launch_art_wgrad(planes, dplanes, d_raw, dxp, Np, params_host, shape, appearance, articulation, grads_host, g_shape,
                                     g_appearance, g_articulation, static_cast<float*>(workspace), (hipStream_t)stream, nullptr,
                                     max_deg_point - min_deg_point, deg_view, packed_bwd), "aon_art_wgrad");
}

// ---- round 6: the end of a training step on one parameter arena (csrc/aon_optim.hip) ----
int aon_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                  int64_t step, void* stream) {
  if (n < 0 || step < 1) return fail(AON_E_INVALID, "aon_adam_step: n must be >= 0 and step >= 1 (the count AFTER this update, as torch.optim.Adam's state['step'])");
  if (n == 0) return AON_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq) return fail(AON_E_INVALID, "aon_adam_step: null pointer");
  if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return fail(AON_E_INVALID, "aon_adam_step: bad hyper-parameter");
  return check(aon::launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, (hipStream_t)stream), "aon_adam_step");
}

static int code_library_call(bool backward, const float* const* src, const int64_t* const* ids, const int* rows, const int* dims, float* const* dst, void* stream,
                             const char* who) {
  if (!src || !ids || !rows || !dims || !dst) return fail(AON_E_INVALID, "aon_code_library: null array");
  for (int t = 0; t < 3; ++t)
    if (!src[t] || !ids[t] || !dst[t] || rows[t] < 1 || dims[t] < 1 || (int64_t)rows[t] * dims[t] > (1 << 28)) return fail(AON_E_INVALID, "aon_code_library: null pointer or bad table size");
  return check(aon::launch_code_library(backward, src, ids, rows, dims, dst, (hipStream_t)stream), who);
}
int aon_code_library_fwd(const float* const* tables_host, const int64_t* const* ids_host, const int* rows_host, const int* dims_host, float* const* out_host,
                         void* stream) {
  return code_library_call(false, tables_host, ids_host, rows_host, dims_host, out_host, stream, "aon_code_library_fwd");
}
int aon_code_library_bwd(const float* const* g_rows_host, const int64_t* const* ids_host, const int* rows_host, const int* dims_host, float* const* g_tables_host,
                         void* stream) {
  return code_library_call(true, g_rows_host, ids_host, rows_host, dims_host, g_tables_host, stream, "aon_code_library_bwd");
}

int aon_profile_begin(void) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.on = true; g_prof.used = 0; g_prof.recs.clear();
  for (int c = 0; c < kNumClasses; ++c) g_prof.units[c] = 0;
  return AON_OK;
}

int aon_profile_end(double* mlp_ms, int64_t* mlp_launches, int64_t* mlp_samples) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.on = false;
  for (int c = 0; c < kNumClasses; ++c) { g_prof.last_ms[c] = 0.0; g_prof.last_launches[c] = 0; g_prof.last_units[c] = g_prof.units[c]; }
  for (const auto& r : g_prof.recs) {
    hipError_t e = hipEventSynchronize(r.stop);
    if (e != hipSuccess) return check(e, "aon_profile_end");
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, r.start, r.stop);
    if (e != hipSuccess) return check(e, "aon_profile_end");
    g_prof.last_ms[r.cls] += ms;
    g_prof.last_launches[r.cls] += 1;
  }
  if (mlp_ms) *mlp_ms = g_prof.last_ms[kMlpFwd];
  if (mlp_launches) *mlp_launches = g_prof.last_launches[kMlpFwd];
  if (mlp_samples) *mlp_samples = g_prof.last_units[kMlpFwd];
  g_prof.used = 0; g_prof.recs.clear();
  return AON_OK;
}

int aon_profile_class(int cls, double* ms, int64_t* launches, int64_t* units) {
  if (cls < 0 || cls >= kNumClasses) return fail(AON_E_INVALID, "aon_profile_class: unknown kernel class");
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (ms) *ms = g_prof.last_ms[cls];
  if (launches) *launches = g_prof.last_launches[cls];
  if (units) *units = g_prof.last_units[cls];
  return AON_OK;
}

int64_t aon_render_workspace_bytes_ex(int64_t n_rays, const aon_render_opts* opts) {
  if (n_rays < 1) n_rays = 1;
  Geo g;
  if (const char* bad = make_geo(opts, g)) return fail(AON_E_INVALID, bad);
  return carve(nullptr, n_rays, g).bytes;
}
int64_t aon_render_workspace_bytes(int64_t n_rays) { return aon_render_workspace_bytes_ex(n_rays, nullptr); }

// Whole-path orchestration shared by the vanilla and the articulated network (NeRF.forward, model.py:147-199;
// NeRF_AE_Art.forward, model_autodecoder.py:278-337): only the MLP launch and the output activation differ.
// Round 5: whole-path calls of the vanilla network in its folded form hand the view layer b' + W_v0[:, 256:] ve as a per-ray bias
// (launch_view_bias) instead of running the view-encoding chunk per sample -- same bits (aon_common.h).  0: the chunk form.
std::atomic<int> g_view_bias{1};

struct NetRef {
  bool articulated;
  const void* packed;
  const float* small;  // articulated only
};

static hipError_t launch_net(const NetRef& net, const float* o, const float* d, const float* v, const float* t, int64_t n, int S,
                             float* raw, hipStream_t stream, const Geo* g = nullptr, const Ws* w = nullptr) {
  if (g && g->other_degrees) {
    // NeRF(min_deg_point, max_deg_point, deg_view) with at most 10 / 4 levels: the encodings are computed by the stage kernels in
    // the fused kernel's 63 / 27-slot layout (zeros in the missing levels' slots, matched by zero weights in the packed stream,
    // aon_pack_vanilla_mlp_deg) and the MLP runs as NeRFMLP.forward(x, condition) on them: 252 B/sample of extra HBM traffic
    // against 1.19 MFLOP/sample
    if (hipError_t e = aon::launch_cast_rays(t, o, d, n, S, w->coords, stream); e != hipSuccess) return e;
    if (hipError_t e = aon::launch_pos_enc(w->coords, n * S, g->min_deg, g->max_deg, w->enc, stream, aon::kPosEnc, 10); e != hipSuccess) return e;
    if (hipError_t e = aon::launch_pos_enc(v, n, 0, g->deg_view, w->venc, stream, aon::kViewEnc, 4); e != hipSuccess) return e;
    MlpTimer timer(stream, n * S);
    return aon::launch_mlp_fwd_enc(static_cast<const char*>(net.packed), w->enc, w->venc, n, S, raw, stream);
  }
  const float* vbias = nullptr;
  if (w && w->vbias && g_view_bias.load(std::memory_order_relaxed) != 0 && aon::stream_form(net.packed) == aon::kFormFolded) {
    const hipError_t e = net.articulated ? aon::launch_art_view_bias(static_cast<const char*>(net.packed), net.small, v, n, w->vbias, stream)
                                         : aon::launch_view_bias(static_cast<const char*>(net.packed), v, n, w->vbias, stream);
    if (e != hipSuccess) return e;
    vbias = w->vbias;
  }
  MlpTimer timer(stream, n * S);
  if (net.articulated)
    return aon::launch_art_mlp_fwd(static_cast<const char*>(net.packed), net.small, o, d, v, t, n, S, raw, stream, vbias);
  return aon::launch_mlp_fwd(static_cast<const char*>(net.packed), o, d, v, t, n, S, raw, stream, vbias);
}

static int render_impl(const char* who, const NetRef& coarse, const NetRef& fine, const float* rays_o, const float* rays_d,
                       const float* viewdirs, int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels,
                       const float* t_rand, const float* u, int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c,
                       float* rgb_f, float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, hipStream_t stream,
                       const aon_render_opts* opts) {
  Geo g;
  if (const char* bad = make_geo(opts, g)) return fail(AON_E_INVALID, bad);
  if (n_rays < 0 || (num_levels != 1 && num_levels != 2)) return fail(AON_E_INVALID, "render: bad size / num_levels");
  if (n_rays == 0) return AON_OK;
  if (!coarse.packed || !rays_o || !rays_d || !viewdirs || !rgb_c || !acc_c || !depth_c || !workspace)
    return fail(AON_E_INVALID, "render: null pointer");
  if (num_levels == 2 && (!fine.packed || !rgb_f || !acc_f || !depth_f || !u))
    return fail(AON_E_INVALID, "render: null fine-level pointer");
  if (coarse.articulated && (!coarse.small || (num_levels == 2 && !fine.small))) return fail(AON_E_INVALID, "render: null latent block");
  if (coarse.articulated && (forms_differ(coarse.packed, coarse.small) || (num_levels == 2 && forms_differ(fine.packed, fine.small)))) return fail(AON_E_INVALID, kFormsMsg);
  if (num_levels == 2 && u_stride != 0 && u_stride < g.nf) return fail(AON_E_INVALID, "render: bad u_stride");
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(AON_E_INVALID, "render: workspace must be 256-byte aligned");
  const bool art = coarse.articulated;
  if (art) g.other_degrees = false;   // the articulated kernels carry their degrees in the packed stream and the small block (aon_*_deg)
  const bool fuse_coarse = num_levels == 2 && g.default_sizes && g_fuse_coarse.load(std::memory_order_relaxed) != 0;

  // largest chunk the workspace admits
  int64_t chunk = n_rays;
  if (carve(nullptr, chunk, g).bytes > workspace_bytes) {
    const int64_t per_ray = (int64_t)(g.Sc + g.Sc + g.Sf + 4 * g.Sf + (g.other_degrees ? (3 + aon::kPosEnc) * g.Sf + aon::kViewEnc : aon::kCondWidth)) * 4;
    const int64_t slack = g.other_degrees ? 7 * 256 : 5 * 256;
    chunk = (workspace_bytes - slack) / per_ray;
    while (chunk > 0 && carve(nullptr, chunk, g).bytes > workspace_bytes) --chunk;
    if (chunk < 1) return fail(AON_E_WORKSPACE, "render: workspace smaller than aon_render_workspace_bytes(1)");
  }
  const Ws w = carve(static_cast<char*>(workspace), chunk, g);

  for (int64_t r0 = 0; r0 < n_rays; r0 += chunk) {
    const int64_t n = n_rays - r0 < chunk ? n_rays - r0 : chunk;
    const float* o = rays_o + r0 * 3;
    const float* d = rays_d + r0 * 3;
    const float* v = viewdirs + r0 * 3;
    const float* uu = u_stride ? u + r0 * u_stride : u;
    int rc;
    // level 0 (model.py:150-160, :175-197)
    {
      KTimer timer(kSampleT, stream, n);
      rc = check(aon::launch_sample_along_rays(o, d, n, g.Sc, near_, far_, t_rand ? t_rand + r0 * g.Sc : nullptr, w.t_c, nullptr, stream,
                                               g.lindisp, g.inv_near, g.inv_far), who);
    }
    if (rc) return rc;
    rc = check(launch_net(coarse, o, d, v, w.t_c, n, g.Sc, w.raw, stream, &g, &w), who);
    if (rc) return rc;
    if (fuse_coarse) {
      // compositing + the fine level's sampling (model.py:162-173) in one kernel: the coarse weights stay in registers
      KTimer timer(kCompositePdf, stream, n);
      rc = check(aon::launch_composite_pdf(w.raw, w.t_c, d, n, white_bkgd, g.act(art, 0, r0), uu, u_stride, rgb_c + r0 * 3,
                                           acc_c + r0, depth_c + r0, nullptr, w.t_f, stream), who);
    } else {
      KTimer timer(kComposite, stream, n);
      rc = check(aon::launch_composite(w.raw, 4, w.raw + 3, 4, w.t_c, d, n, g.Sc, white_bkgd, g.act(art, 0, r0), rgb_c + r0 * 3, acc_c + r0,
                                       depth_c + r0, num_levels == 2 ? w.w_c : nullptr, stream), who);
    }
    if (rc) return rc;
    if (num_levels == 1) continue;
    // level 1 (model.py:162-173, :175-197)
    if (!fuse_coarse) {
      KTimer timer(kSamplePdf, stream, n);
      rc = check(g.default_sizes ? aon::launch_sample_pdf(nullptr, w.w_c + 1, kSc, w.t_c, uu, u_stride, n, nullptr, w.t_f, stream)
                                 : aon::launch_sample_pdf_n(nullptr, w.w_c + 1, g.Sc, w.t_c, uu, u_stride, n, g.Sc - 1, g.nf, g.Sc, nullptr,
                                                            w.t_f, stream), who);
      if (rc) return rc;
    }
    rc = check(launch_net(fine, o, d, v, w.t_f, n, g.Sf, w.raw, stream, &g, &w), who);
    if (rc) return rc;
    {
      KTimer timer(kComposite, stream, n);
      rc = check(aon::launch_composite(w.raw, 4, w.raw + 3, 4, w.t_f, d, n, g.Sf, white_bkgd, g.act(art, 1, r0), rgb_f + r0 * 3, acc_f + r0,
                                       depth_f + r0, nullptr, stream), who);
    }
    if (rc) return rc;
  }
  return AON_OK;
}

int aon_render_fwd_ex(const void* packed_coarse, const void* packed_fine, const float* rays_o, const float* rays_d,
                      const float* viewdirs, int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels,
                      const float* t_rand, const float* u, int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c,
                      float* rgb_f, float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream,
                      const aon_render_opts* opts) {
  const NetRef c{false, packed_coarse, nullptr}, f{false, packed_fine, nullptr};
  return render_impl("aon_render_fwd", c, f, rays_o, rays_d, viewdirs, n_rays, near_, far_, white_bkgd, num_levels, t_rand, u,
                     u_stride, rgb_c, acc_c, depth_c, rgb_f, acc_f, depth_f, workspace, workspace_bytes, (hipStream_t)stream, opts);
}
int aon_render_fwd(const void* packed_coarse, const void* packed_fine, const float* rays_o, const float* rays_d,
                   const float* viewdirs, int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels,
                   const float* t_rand, const float* u, int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c,
                   float* rgb_f, float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream) {
  return aon_render_fwd_ex(packed_coarse, packed_fine, rays_o, rays_d, viewdirs, n_rays, near_, far_, white_bkgd, num_levels, t_rand, u,
                           u_stride, rgb_c, acc_c, depth_c, rgb_f, acc_f, depth_f, workspace, workspace_bytes, stream, nullptr);
}

// ---- training step in two calls (aon_render_fwd_train + aon_render_bwd) ----
// The whole forward of NeRF.forward / NeRF_AE_Art.forward under grad mode, then the whole backward of the reference's
// training loss, each as ONE C call on the exact-fp32 engine: the per-level staging that autograd.py used to drive from
// Python (sample -> fused MLP + planes -> composite -> inverse CDF | composite backward -> chain -> weight gradients)
// happens here.  Everything the backward needs stays in the caller's workspace between the two calls.
namespace {

struct TrainLevel {
  float* t;        // n*S
  float* raw;      // Np*4 (first n*S records valid)
  float* planes;   // rows*Np
  char* masks;     // mask_layers*Np*32
  float* coords; float* enc; float* venc;   // other encoding degrees only: n*S*3, n*S*63, n*27 (forward-only temporaries)
  float* vbias;    // n*128, the level's per-ray view bias (forward-only temporary; not for the vanilla network at other degrees)
  int S; int64_t Np;
};
// What the forward leaves for the backward (caller-owned, pinned by the autograd graph): per level t, raw, planes, ReLU bits.
struct TrainWs {
  TrainLevel lvl[2];
  float* w_c;      // n*Sc coarse weights
  int64_t bytes;
};
// Backward-only temporaries (round 3: a separate `scratch` of aon_render_bwd, allocated when the backward runs -- round 2 carved
// them into the forward's workspace, so every live graph pinned 26 GB instead of 15 GB at 4096 articulated rays), one set per
// level: the two levels' backward passes are independent and run on two streams.
struct TrainScratch {
  float* d_raw[2];    // Np*4
  float* dplanes[2];  // rows*Np
  float* dxp[2];      // Np*4 (articulated)
  float* wgrad_ws[2];
  float* lat_tmp;     // 288 floats: second level's latent gradients before they are added (articulated)
  float* grad_tmp[2]; // other encoding degrees: the three encoding-fed weight gradients in the kernels' 63 / 27-column layout
  int64_t bytes;
};
constexpr int64_t kGradTmpFloats = 256 * 63 + 256 * (256 + 63) + 128 * (256 + 27);

int64_t level_np(int64_t n, int l, const Geo& g) { return align_up(n * g.S(l), 128); }

// both carves cover only the levels in use: num_levels = 1 (BASELINE config 1) takes a quarter of the two-level size
TrainWs carve_train(char* base, int64_t n, bool art, int num_levels, const Geo& g) {
  TrainWs w{};
  const int64_t rows = art ? aon::kAPlRows : aon::kPlRows;
  const int64_t mlayers = art ? aon::kAMaskLayers : aon::kMaskLayers;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = base + off; off += align_up(bytes, 256); return p; };
  for (int l = 0; l < num_levels; ++l) {
    const int S = g.S(l);
    const int64_t Np = level_np(n, l, g);
    w.lvl[l].S = S; w.lvl[l].Np = Np;
    w.lvl[l].t = reinterpret_cast<float*>(take(n * S * 4));
    w.lvl[l].raw = reinterpret_cast<float*>(take(Np * 16));
    w.lvl[l].planes = reinterpret_cast<float*>(take(rows * Np * 4));
    w.lvl[l].masks = take(mlayers * Np * 32);
    if (g.other_degrees && !art) {
      w.lvl[l].coords = reinterpret_cast<float*>(take(n * S * 12));
      w.lvl[l].enc = reinterpret_cast<float*>(take(n * S * (int64_t)aon::kPosEnc * 4));
      w.lvl[l].venc = reinterpret_cast<float*>(take(n * (int64_t)aon::kViewEnc * 4));
    }
    if (!g.other_degrees || art) w.lvl[l].vbias = reinterpret_cast<float*>(take(n * (int64_t)aon::kCondWidth * 4));
  }
  w.w_c = reinterpret_cast<float*>(take(n * g.Sc * 4));
  w.bytes = off;
  return w;
}

TrainScratch carve_scratch(char* base, int64_t n, bool art, int num_levels, const Geo& g) {
  TrainScratch sc{};
  const int64_t rows = art ? aon::kAPlRows : aon::kPlRows;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = base + off; off += align_up(bytes, 256); return p; };
  for (int l = 0; l < num_levels; ++l) {
    const int64_t Np = level_np(n, l, g);
    sc.d_raw[l] = reinterpret_cast<float*>(take(Np * 16));
    sc.dplanes[l] = reinterpret_cast<float*>(take(rows * Np * 4));
    sc.dxp[l] = reinterpret_cast<float*>(take(Np * 16));
    sc.wgrad_ws[l] = reinterpret_cast<float*>(take(aon::wgrad_workspace_bytes()));
    if (g.other_degrees && !art) sc.grad_tmp[l] = reinterpret_cast<float*>(take(kGradTmpFloats * 4));
  }
  sc.lat_tmp = reinterpret_cast<float*>(take(288 * 4));
  sc.bytes = off;
  return sc;
}

// dst (rows, hidden + 3 + 6 L) <- src (rows, hidden + 3 + 6 Lfull): the columns of an encoding with L levels picked out of the
// kernels' Lfull-level slot layout [x ; first block of 3 Lfull ; shifted block of 3 Lfull]
__global__ void remap_enc_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int hidden, int L, int Lfull) {
  const int cols = hidden + 3 + 6 * L, lds = hidden + 3 + 6 * Lfull;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  int sc = c;
  if (c >= hidden + 3 + 3 * L) sc = c + 3 * (Lfull - L);
  dst[i] = src[(int64_t)r * lds + sc];
}

__global__ void add_into_kernel(float* __restrict__ dst, const float* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// Two library-owned streams per device for the backward of the two levels (independent until the latent gradients are
// added): kernels of one level fill the CUs the other level's tail rounds leave idle.  Ordered against the caller's stream by
// events on both sides, so from the caller's view the whole backward is enqueued on its stream.
struct LevelStreams {
  hipStream_t s[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
  aon::WgAux aux[2] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};   // per level: side stream of the head reductions
  std::mutex enqueue;   // held from the fork record to the join waits: the one event set is re-recorded by every call
};
LevelStreams* level_streams() {
  static LevelStreams per_dev[aon::kMaxDevices];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= aon::kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  LevelStreams& ls = per_dev[dev];
  if (!ls.fork) {
    // AON_SIDE_PRIORITY=1 (experiment, round 6): the side streams at the LOWEST priority the device offers, so that when a side-stream kernel and
    // a persistent launch on the caller's stream become eligible at the same moment the command processor dispatches the persistent launch
    // first.  Built while hunting whole processes that ran the config-5 step at 33-40 ms instead of 30.4 (tools/slowmode_probe.sh); the cause
    // turned out to be a hipMalloc of the 11-15 GB workspaces inside the step (ops._TRAIN_POOL), not the dispatch order -- and at low
    // priority the early head reductions start too late to fill the chain's last round (+0.25 ms per step).  Default: normal priority.
    int least = 0, greatest = 0;
    const char* e = std::getenv("AON_SIDE_PRIORITY");
    const bool low = e && e[0] == '1';
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    auto make = [&](hipStream_t* st) { return low ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, least) : hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
    for (int i = 0; i < 2; ++i) {
      if (make(&ls.s[i]) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&ls.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
      if (make(&ls.aux[i].stream) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&ls.aux[i].fork, hipEventDisableTiming) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&ls.aux[i].join, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    if (hipEventCreateWithFlags(&ls.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &ls;
}

// One backward's use of the two streams.  The constructor takes the device's enqueue lock and forks (fork event on the
// caller's stream, both side streams wait for it); join() -- also run by the destructor, so on EVERY return path, error
// returns included -- records a join event behind whatever was enqueued on each side stream and makes the caller's stream
// wait for both.  Two host threads driving the same device from different caller streams therefore cannot interleave their
// event records, and no side-stream work is ever left un-joined while the caller frees the workspace on its stream.
class LevelFork {
 public:
  // aux_only: no level streams -- the caller's stream carries both levels -- but the side streams of the head reductions (aux()) are
  // handed out (each use forks from / joins to the stream it is given inside run_wgrad_plan); the device's enqueue lock is held alike
  LevelFork(bool overlap, hipStream_t caller, const char* who, bool aux_only = false) : caller_(caller), who_(who), aux_only_(aux_only) {
    if (!overlap) return;
    ls_ = level_streams();
    if (!ls_) return;   // no streams: run serially on the caller's stream
    lk_ = std::unique_lock<std::mutex>(ls_->enqueue);
    if (!aux_only_) {
      rc_ = check(hipEventRecord(ls_->fork, caller_), who_);
      for (int l = 0; l < 2 && !rc_; ++l) rc_ = check(hipStreamWaitEvent(ls_->s[l], ls_->fork, 0), who_);
    }
    forked_ = true;     // even on a partial failure: join() is harmless and keeps the caller ordered behind the side streams
  }
  ~LevelFork() { (void)join(); }
  int rc() const { return rc_; }
  hipStream_t stream(int level) const { return forked_ && !aux_only_ ? ls_->s[level] : caller_; }
  // side stream of a level's head reductions (only while this object holds the device's enqueue lock: the events are shared)
  const aon::WgAux* aux(int level) const { return forked_ ? &ls_->aux[level] : nullptr; }
  int join() {
    if (!forked_) return AON_OK;
    forked_ = false;
    int rc = AON_OK;
    for (int l = 0; l < 2 && !aux_only_; ++l) {
      hipError_t e = hipEventRecord(ls_->join[l], ls_->s[l]);
      if (e == hipSuccess) e = hipStreamWaitEvent(caller_, ls_->join[l], 0);
      if (e != hipSuccess && !rc) rc = check(e, who_);
    }
    lk_.unlock();
    return rc;
  }
 private:
  LevelStreams* ls_ = nullptr;
  hipStream_t caller_;
  const char* who_;
  std::unique_lock<std::mutex> lk_;
  int rc_ = AON_OK;
  bool forked_ = false;
  bool aux_only_ = false;
};
std::atomic<int> g_bwd_overlap{1};
std::atomic<int> g_fwd_overlap{2};
std::atomic<int> g_fwd_merge{1};
std::atomic<int> g_bwd_merge{1};   // the backward chains of the two levels as ONE persistent launch of two segments (round 4)
// Round 5: the head / bias reductions that need nothing from the chain (density head on H7, rgb head, sums of d_raw: 60 % of the head
// kernel's bytes) run on a library side stream BESIDE the merged chain launch, whose last round of workgroups is a quarter full
// (8,256 passes on 256 CUs): they fill compute units that would idle for one pass.  Phases kWgEarly / kWgRest of the levels' weight-gradient calls.
std::atomic<int> g_bwd_early_heads{1};
constexpr int kWgAll = 0, kWgEarly = 1, kWgRest = 2;   // aon_wgrad.h

// The merged training forward (round 4) runs the two levels of two ray ranges A = [0, kA), B = [kA, n) as THREE persistent launches
//   coarse(A)  |  fine(A) + coarse(B)  |  fine(B)
// (the levels of a range depend on each other through its own inverse CDF only).  kA, a multiple of 128 rays, is chosen so that the
// three launches together take the fewest rounds of `cus` workgroups: 4096 rays x (65 + 193) samples on 256 CUs are 32.25 rounds of
// work; one launch per level costs 9 + 25 = 34 rounds, the split at 1,920 rays 4 + 16 + 13 = 33.  Ties: the more balanced split.
int64_t merged_split(int64_t n, int Sc, int Sf, int cus, bool always) {
  auto rounds = [&](int64_t passes) { return (passes + cus - 1) / cus; };
  auto passes = [&](int64_t rays, int S) { return (rays * S + 127) / 128; };
  int64_t best = 0, best_cost = -1, best_bal = 0;
  for (int64_t k = 128; k < n; k += 128) {
    const int64_t cost = rounds(passes(k, Sc)) + rounds(passes(k, Sf) + passes(n - k, Sc)) + rounds(passes(n - k, Sf));
    const int64_t bal = k < n - k ? n - 2 * k : 2 * k - n;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && bal < best_bal)) { best = k; best_cost = cost; best_bal = bal; }
  }
  // not worth it when one launch per level is as good (small batches: everything fits one round)
  if (best_cost < 0 || (!always && best_cost >= rounds(passes(n, Sc)) + rounds(passes(n, Sf)))) return 0;
  return best;
}

struct TrainNet {   // one level's network handles
  const void* packed_fwd; const float* small; const void* packed_bwd;
};

int train_fwd_impl(const char* who, bool art, const TrainNet* nets, const float* rays_o, const float* rays_d, const float* viewdirs,
                   int64_t n, float near_, float far_, int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride,
                   float* const* rgb, float* const* acc, float* const* depth, void* workspace, int64_t workspace_bytes, hipStream_t stream,
                   const aon_render_opts* opts) {
  Geo g;
  if (const char* bad = make_geo(opts, g)) return fail(AON_E_INVALID, bad);
  if (g.Sf > 512) return fail(AON_E_INVALID, "train forward: more than 512 samples per ray at the fine level");
  if (art) g.other_degrees = false;   // (as in render_impl: no stage-kernel encodings for the articulated network)
  if (n <= 0 || (num_levels != 1 && num_levels != 2)) return fail(AON_E_INVALID, "train forward: bad size / num_levels");
  if (!rays_o || !rays_d || !viewdirs || !workspace) return fail(AON_E_INVALID, "train forward: null pointer");
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(AON_E_INVALID, "train forward: workspace must be 256-byte aligned");
  const TrainWs w = carve_train(static_cast<char*>(workspace), n, art, num_levels, g);
  if (w.bytes > workspace_bytes) return fail(AON_E_WORKSPACE, "train forward: workspace smaller than aon_train_workspace_bytes()");
  if (num_levels == 2 && (!u || (u_stride != 0 && u_stride < g.nf))) return fail(AON_E_INVALID, "train forward: bad u / u_stride");
  const bool fuse = num_levels == 2 && g.default_sizes && g_fuse_coarse.load(std::memory_order_relaxed) != 0;
  for (int l = 0; l < num_levels; ++l)
    if (!nets[l].packed_fwd || (art && !nets[l].small) || !rgb[l] || !acc[l] || !depth[l]) return fail(AON_E_INVALID, "train forward: null level pointer");
  for (int l = 0; l < num_levels; ++l)
    if ((art && forms_differ(nets[l].packed_fwd, nets[l].small)) || forms_differ(nets[l].packed_fwd, nets[0].packed_fwd))
      return fail(AON_E_INVALID, "train forward: the levels' streams / per-call blocks were made in different forms (aon_set_bottleneck_fold changed in between)");
  const int64_t rows = art ? aon::kAPlRows : aon::kPlRows;
  // the view-encoding term of the first view layer as a per-ray bias (aon_set_view_bias): both levels' biases of the whole batch up front
  const bool use_vb = !g.other_degrees && g_view_bias.load(std::memory_order_relaxed) != 0 && aon::stream_form(nets[0].packed_fwd) == aon::kFormFolded;
  if (use_vb)
    for (int l = 0; l < num_levels; ++l)
      if (int rc = check(art ? aon::launch_art_view_bias(static_cast<const char*>(nets[l].packed_fwd), nets[l].small, viewdirs, n, w.lvl[l].vbias, stream)
                             : aon::launch_view_bias(static_cast<const char*>(nets[l].packed_fwd), viewdirs, n, w.lvl[l].vbias, stream), who)) return rc;

  // Both levels of the ray range [r0, r0 + nk) on stream `st`.  r0 is a multiple of 128, so the range's samples start on a pass
  // boundary at both levels: its planes / decision bits / raw records are the whole batch's buffers at an offset, the slot stride of
  // the decision bits stays the whole batch's Np.
  auto run_range = [&](int64_t r0, int64_t nk, hipStream_t st) -> int {
    const float* o = rays_o + r0 * 3; const float* d = rays_d + r0 * 3; const float* v = viewdirs + r0 * 3;
    const float* uu = (u && u_stride) ? u + r0 * u_stride : u;
    for (int l = 0; l < num_levels; ++l) {
      const TrainLevel& L = w.lvl[l];
      const int64_t s0 = r0 * L.S;
      float* t = L.t + s0; float* raw = L.raw + s0 * 4;
      float* planes = L.planes + s0 * rows; char* masks = L.masks + s0 * 32;
      float* t_next = num_levels == 2 ? w.lvl[1].t + r0 * w.lvl[1].S : nullptr;
      int rc = AON_OK;
      if (l == 0) {
        KTimer timer(kSampleT, st, nk);
        rc = check(aon::launch_sample_along_rays(o, d, nk, g.Sc, near_, far_, t_rand ? t_rand + r0 * g.Sc : nullptr, t, nullptr, st, g.lindisp, g.inv_near,
                                                 g.inv_far), who);
      } else if (!fuse) {
        KTimer timer(kSamplePdf, st, nk);
        const float* wc = w.w_c + r0 * g.Sc; const float* tc = w.lvl[0].t + r0 * g.Sc;
        rc = check(g.default_sizes ? aon::launch_sample_pdf(nullptr, wc + 1, kSc, tc, uu, u_stride, nk, nullptr, t, st)
                                   : aon::launch_sample_pdf_n(nullptr, wc + 1, g.Sc, tc, uu, u_stride, nk, g.Sc - 1, g.nf, g.Sc, nullptr, t, st), who);
      }
      if (rc) return rc;
      if (g.other_degrees) {
        // other encoding degrees: encodings by the stage kernels in the padded 63 / 27-slot layout, then the training forward on
        // caller-encoded inputs (same planes, same decision bits)
        float* coords = L.coords + s0 * 3; float* enc = L.enc + s0 * aon::kPosEnc; float* venc = L.venc + r0 * aon::kViewEnc;
        if ((rc = check(aon::launch_cast_rays(t, o, d, nk, L.S, coords, st), who))) return rc;
        if ((rc = check(aon::launch_pos_enc(coords, nk * L.S, g.min_deg, g.max_deg, enc, st, aon::kPosEnc, 10), who))) return rc;
        if ((rc = check(aon::launch_pos_enc(v, nk, 0, g.deg_view, venc, st, aon::kViewEnc, 4), who))) return rc;
        MlpTimer timer(st, nk * L.S);
        rc = check(aon::launch_mlp_fwd_train_enc(static_cast<const char*>(nets[l].packed_fwd), enc, venc, nk, L.S, raw, planes, masks, st, L.Np), who);
      } else {
        MlpTimer timer(st, nk * L.S);
        const float* vb = use_vb ? L.vbias + r0 * aon::kCondWidth : nullptr;
        rc = check(art ? aon::launch_art_mlp_fwd_train(static_cast<const char*>(nets[l].packed_fwd), nets[l].small, o, d, v, t, nk, L.S, raw, planes, masks,
                                                       st, L.Np, vb)
                       : aon::launch_mlp_fwd_train(static_cast<const char*>(nets[l].packed_fwd), o, d, v, t, nk, L.S, raw, planes, masks, st, L.Np, vb), who);
      }
      if (rc) return rc;
      if (l == 0 && fuse) {
        KTimer timer(kCompositePdf, st, nk);
        rc = check(aon::launch_composite_pdf(raw, t, d, nk, white_bkgd, g.act(art, 0, r0), uu, u_stride, rgb[0] + r0 * 3, acc[0] + r0, depth[0] + r0, nullptr,
                                             t_next, st), who);
      } else {
        KTimer timer(kComposite, st, nk);
        rc = check(aon::launch_composite(raw, 4, raw + 3, 4, t, d, nk, L.S, white_bkgd, g.act(art, l, r0), rgb[l] + r0 * 3, acc[l] + r0, depth[l] + r0,
                                         (l == 0 && num_levels == 2) ? w.w_c + r0 * g.Sc : nullptr, st), who);
      }
      if (rc) return rc;
    }
    return AON_OK;
  };

  // Merged form (round 4, default): coarse(A) | fine(A) + coarse(B) | fine(B) as three persistent launches on the caller's stream.
  if (const int merge = g_fwd_merge.load(std::memory_order_relaxed); num_levels == 2 && !g.other_degrees && merge) {
    const int cus = aon::num_cus();
    const int64_t kA = cus > 0 ? merged_split(n, g.Sc, g.Sf, cus, merge == 2) : 0;
    if (kA > 0) {
      struct Rng { int64_t r0, nk; };
      const Rng R[2] = {{0, kA}, {kA, n - kA}};
      auto seg_of = [&](const Rng& r, int l) {
        const TrainLevel& L = w.lvl[l];
        const int64_t s0 = r.r0 * L.S;
        return aon::TrainSeg{static_cast<const char*>(nets[l].packed_fwd), nets[l].small, rays_o + r.r0 * 3, rays_d + r.r0 * 3, viewdirs + r.r0 * 3,
                             L.t + s0, r.nk, L.S, L.raw + s0 * 4, L.planes + s0 * rows, L.masks + s0 * 32, L.Np,
                             use_vb ? L.vbias + r.r0 * aon::kCondWidth : nullptr};
      };
      auto mlp = [&](const aon::TrainSeg* segs, int ns) {
        int64_t samples = 0;
        for (int i = 0; i < ns; ++i) samples += segs[i].n_rays * segs[i].S;
        MlpTimer timer(stream, samples);
        return check(art ? aon::launch_art_mlp_fwd_train2(segs, ns, stream) : aon::launch_mlp_fwd_train2(segs, ns, stream), who);
      };
      auto coarse_tail = [&](const Rng& r) {   // compositing + inverse CDF + merge of the range's coarse level -> its fine t
        const float* d = rays_d + r.r0 * 3;
        const float* uu = (u && u_stride) ? u + r.r0 * u_stride : u;
        const TrainLevel& L = w.lvl[0];
        float* t = L.t + r.r0 * L.S; float* raw = L.raw + r.r0 * L.S * 4;
        float* t_next = w.lvl[1].t + r.r0 * w.lvl[1].S;
        if (fuse) {
          KTimer timer(kCompositePdf, stream, r.nk);
          return check(aon::launch_composite_pdf(raw, t, d, r.nk, white_bkgd, g.act(art, 0, r.r0), uu, u_stride, rgb[0] + r.r0 * 3, acc[0] + r.r0, depth[0] + r.r0,
                                                 nullptr, t_next, stream), who);
        }
        {
          KTimer timer(kComposite, stream, r.nk);
          if (int rc = check(aon::launch_composite(raw, 4, raw + 3, 4, t, d, r.nk, L.S, white_bkgd, g.act(art, 0, r.r0), rgb[0] + r.r0 * 3, acc[0] + r.r0,
                                                   depth[0] + r.r0, w.w_c + r.r0 * g.Sc, stream), who)) return rc;
        }
        KTimer timer(kSamplePdf, stream, r.nk);
        const float* wc = w.w_c + r.r0 * g.Sc;
        return check(g.default_sizes ? aon::launch_sample_pdf(nullptr, wc + 1, kSc, t, uu, u_stride, r.nk, nullptr, t_next, stream)
                                     : aon::launch_sample_pdf_n(nullptr, wc + 1, g.Sc, t, uu, u_stride, r.nk, g.Sc - 1, g.nf, g.Sc, nullptr, t_next, stream), who);
      };
      auto fine_tail = [&](const Rng& r) {
        const TrainLevel& L = w.lvl[1];
        KTimer timer(kComposite, stream, r.nk);
        return check(aon::launch_composite(L.raw + r.r0 * L.S * 4, 4, L.raw + r.r0 * L.S * 4 + 3, 4, L.t + r.r0 * L.S, rays_d + r.r0 * 3, r.nk, L.S, white_bkgd,
                                           g.act(art, 1, r.r0), rgb[1] + r.r0 * 3, acc[1] + r.r0, depth[1] + r.r0, nullptr, stream), who);
      };
      {   // stratified t of the whole batch
        KTimer timer(kSampleT, stream, n);
        if (int rc = check(aon::launch_sample_along_rays(rays_o, rays_d, n, g.Sc, near_, far_, t_rand, w.lvl[0].t, nullptr, stream, g.lindisp, g.inv_near,
                                                         g.inv_far), who)) return rc;
      }
      const aon::TrainSeg cA = seg_of(R[0], 0), fA = seg_of(R[0], 1), cB = seg_of(R[1], 0), fB = seg_of(R[1], 1);
      if (int rc = mlp(&cA, 1)) return rc;
      if (int rc = coarse_tail(R[0])) return rc;
      const aon::TrainSeg mid[2] = {fA, cB};
      if (int rc = mlp(mid, 2)) return rc;
      if (int rc = coarse_tail(R[1])) return rc;
      if (int rc = fine_tail(R[0])) return rc;
      if (int rc = mlp(&fB, 1)) return rc;
      return fine_tail(R[1]);
    }
  }

  // Two ray halves on the two library streams (round 3): the levels of a half depend on each other through its own inverse CDF
  // only, so one half's fine level fills the CUs the other half's coarse level leaves idle in its last, partial round of
  // workgroups (8.125 rounds of 256 cost 9 at 4096 x 65 samples), as the backward does with its two levels.
  const int parts = g_fwd_overlap.load(std::memory_order_relaxed);   // 0 / 1: one stream; k >= 2: k ray ranges alternating on the two streams
  const int64_t piece = parts >= 2 ? (n / parts) / 128 * 128 : 0;
  if (num_levels == 2 && piece > 0) {
    LevelFork fork(true, stream, who);
    if (fork.rc()) return fork.rc();
    for (int k = 0; k < parts; ++k) {
      const int64_t r0 = k * piece, nk = k == parts - 1 ? n - r0 : piece;
      if (int rc = run_range(r0, nk, fork.stream(k & 1))) return rc;
    }
    return fork.join();
  }
  return run_range(0, n, stream);
}

}  // namespace

int aon_set_bottleneck_fold(int on) {
  aon::set_fold_default(on);
  return AON_OK;
}
int aon_get_bottleneck_fold(void) { return aon::fold_default(); }
int aon_stream_is_folded(const void* packed) { return aon::stream_form(packed) == aon::kFormFolded ? 1 : 0; }
int aon_stream_form(const void* packed) { return aon::stream_form(packed); }
int aon_declare_stream_form(const void* packed, int form) {
  if (!packed || (form != aon::kFormLiteral && form != aon::kFormFolded)) return fail(AON_E_INVALID, "aon_declare_stream_form: null pointer or form not 0 / 1");
  aon::set_stream_form(packed, form);
  return AON_OK;
}

int aon_set_bwd_overlap(int on) {
  g_bwd_overlap.store(on == 2 ? 2 : (on ? 1 : 0), std::memory_order_relaxed);
  return AON_OK;
}

int aon_set_view_bias(int on) {
  g_view_bias.store(on ? 1 : 0, std::memory_order_relaxed);
  return AON_OK;
}

int aon_get_view_bias(void) { return g_view_bias.load(std::memory_order_relaxed); }

int aon_set_bwd_early_heads(int on) {
  g_bwd_early_heads.store(on ? 1 : 0, std::memory_order_relaxed);
  return AON_OK;
}

int aon_set_bwd_merge(int on) {
  g_bwd_merge.store(on ? 1 : 0, std::memory_order_relaxed);
  return AON_OK;
}

int aon_set_fwd_merge(int on) {
  g_fwd_merge.store(on == 2 ? 2 : (on ? 1 : 0), std::memory_order_relaxed);   // 2 (tests): merge whenever there are two ranges, gain or not
  return AON_OK;
}

int aon_set_fwd_overlap(int on) {
  g_fwd_overlap.store(on < 0 ? 0 : (on == 1 ? 2 : (on > 16 ? 16 : on)), std::memory_order_relaxed);   // 1 = the default two halves; k >= 2: k ranges (measurements)
  return AON_OK;
}

int64_t aon_train_workspace_bytes_ex(int64_t n_rays, int articulated, int num_levels, const aon_render_opts* opts) {
  if (n_rays < 1) n_rays = 1;
  Geo g;
  if (const char* bad = make_geo(opts, g)) return fail(AON_E_INVALID, bad);
  return carve_train(nullptr, n_rays, articulated != 0, num_levels == 1 ? 1 : 2, g).bytes;
}
int64_t aon_train_workspace_bytes(int64_t n_rays, int articulated, int num_levels) {
  return aon_train_workspace_bytes_ex(n_rays, articulated, num_levels, nullptr);
}

int64_t aon_train_scratch_bytes_ex(int64_t n_rays, int articulated, int num_levels, const aon_render_opts* opts) {
  if (n_rays < 1) n_rays = 1;
  Geo g;
  if (const char* bad = make_geo(opts, g)) return fail(AON_E_INVALID, bad);
  return carve_scratch(nullptr, n_rays, articulated != 0, num_levels == 1 ? 1 : 2, g).bytes;
}
int64_t aon_train_scratch_bytes(int64_t n_rays, int articulated, int num_levels) {
  return aon_train_scratch_bytes_ex(n_rays, articulated, num_levels, nullptr);
}

int aon_render_fwd_train_ex(const void* packed_coarse, const void* packed_fine, const float* rays_o, const float* rays_d, const float* viewdirs,
                            int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels, const float* t_rand, const float* u,
                            int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c, float* rgb_f, float* acc_f, float* depth_f,
                            void* workspace, int64_t workspace_bytes, void* stream, const aon_render_opts* opts) {
  const TrainNet nets[2] = {{packed_coarse, nullptr, nullptr}, {packed_fine, nullptr, nullptr}};
  float* const rgb[2] = {rgb_c, rgb_f}; float* const acc[2] = {acc_c, acc_f}; float* const dep[2] = {depth_c, depth_f};
  return train_fwd_impl("aon_render_fwd_train", false, nets, rays_o, rays_d, viewdirs, n_rays, near_, far_, white_bkgd, num_levels, t_rand, u,
                        u_stride, rgb, acc, dep, workspace, workspace_bytes, (hipStream_t)stream, opts);
}
int aon_render_fwd_train(const void* packed_coarse, const void* packed_fine, const float* rays_o, const float* rays_d, const float* viewdirs,
                         int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels, const float* t_rand, const float* u,
                         int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c, float* rgb_f, float* acc_f, float* depth_f,
                         void* workspace, int64_t workspace_bytes, void* stream) {
  return aon_render_fwd_train_ex(packed_coarse, packed_fine, rays_o, rays_d, viewdirs, n_rays, near_, far_, white_bkgd, num_levels, t_rand, u,
                                 u_stride, rgb_c, acc_c, depth_c, rgb_f, acc_f, depth_f, workspace, workspace_bytes, stream, nullptr);
}

int aon_art_render_fwd_train_ex(const void* packed_coarse, const void* small_coarse, const void* packed_fine, const void* small_fine,
                                const float* rays_o, const float* rays_d, const float* viewdirs, int64_t n_rays, float near_, float far_,
                                int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride, float* rgb_c,
                                float* acc_c, float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace,
                                int64_t workspace_bytes, void* stream, const aon_render_opts* opts) {
  const TrainNet nets[2] = {{packed_coarse, static_cast<const float*>(small_coarse), nullptr}, {packed_fine, static_cast<const float*>(small_fine), nullptr}};
  float* const rgb[2] = {rgb_c, rgb_f}; float* const acc[2] = {acc_c, acc_f}; float* const dep[2] = {depth_c, depth_f};
  return train_fwd_impl("aon_art_render_fwd_train", true, nets, rays_o, rays_d, viewdirs, n_rays, near_, far_, white_bkgd, num_levels, t_rand, u,
                        u_stride, rgb, acc, dep, workspace, workspace_bytes, (hipStream_t)stream, opts);
}
int aon_art_render_fwd_train(const void* packed_coarse, const void* small_coarse, const void* packed_fine, const void* small_fine,
                             const float* rays_o, const float* rays_d, const float* viewdirs, int64_t n_rays, float near_, float far_,
                             int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride, float* rgb_c,
                             float* acc_c, float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  return aon_art_render_fwd_train_ex(packed_coarse, small_coarse, packed_fine, small_fine, rays_o, rays_d, viewdirs, n_rays, near_, far_, white_bkgd,
                                     num_levels, t_rand, u, u_stride, rgb_c, acc_c, depth_c, rgb_f, acc_f, depth_f, workspace, workspace_bytes,
                                     stream, nullptr);
}

int aon_render_bwd(const void* packed_bwd_coarse, const void* packed_fwd_coarse, const void* packed_bwd_fine, const void* packed_fwd_fine,
                   const float* rays_d, int64_t n_rays, int white_bkgd, int num_levels, const float* const* g_rgb_host,
                   const float* const* g_acc_host, const float* const* g_depth_host, float* const* grads_coarse_host,
                   float* const* grads_fine_host, void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes,
                   void* stream_) {
  return aon_render_bwd_ex(packed_bwd_coarse, packed_fwd_coarse, packed_bwd_fine, packed_fwd_fine, rays_d, n_rays, white_bkgd, num_levels, g_rgb_host,
                           g_acc_host, g_depth_host, grads_coarse_host, grads_fine_host, workspace, workspace_bytes, scratch, scratch_bytes, stream_,
                           nullptr);
}
int aon_render_bwd_ex(const void* packed_bwd_coarse, const void* packed_fwd_coarse, const void* packed_bwd_fine, const void* packed_fwd_fine,
                      const float* rays_d, int64_t n_rays, int white_bkgd, int num_levels, const float* const* g_rgb_host,
                      const float* const* g_acc_host, const float* const* g_depth_host, float* const* grads_coarse_host,
                      float* const* grads_fine_host, void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes,
                      void* stream_, const aon_render_opts* opts) {
  hipStream_t stream = (hipStream_t)stream_;
  Geo g;
  if (const char* bad = make_geo(opts, g)) return fail(AON_E_INVALID, bad);
  if (n_rays <= 0 || (num_levels != 1 && num_levels != 2)) return fail(AON_E_INVALID, "aon_render_bwd: bad size / num_levels");
  if (!rays_d || !g_rgb_host || !workspace || !scratch || !grads_coarse_host) return fail(AON_E_INVALID, "aon_render_bwd: null pointer");
  if (reinterpret_cast<uintptr_t>(scratch) & 255) return fail(AON_E_INVALID, "aon_render_bwd: scratch must be 256-byte aligned");
  const TrainWs w = carve_train(static_cast<char*>(workspace), n_rays, false, num_levels, g);
  if (w.bytes > workspace_bytes) return fail(AON_E_WORKSPACE, "aon_render_bwd: workspace smaller than aon_train_workspace_bytes()");
  const TrainScratch sc = carve_scratch(static_cast<char*>(scratch), n_rays, false, num_levels, g);
  if (sc.bytes > scratch_bytes) return fail(AON_E_WORKSPACE, "aon_render_bwd: scratch smaller than aon_train_scratch_bytes()");
  const void* pb[2] = {packed_bwd_coarse, packed_bwd_fine};
  const void* pf[2] = {packed_fwd_coarse, packed_fwd_fine};
  float* const* grads[2] = {grads_coarse_host, grads_fine_host};
  for (int l = 0; l < num_levels; ++l) {
    if (!pb[l] || !pf[l] || !grads[l] || !g_rgb_host[l]) return fail(AON_E_INVALID, "aon_render_bwd: null level pointer");
    if (aon::stream_form(pb[l]) != aon::stream_form(pf[l]) || aon::stream_form(pb[l]) != aon::stream_form(pb[0]))
      return fail(AON_E_INVALID, "aon_render_bwd: forward and transposed streams were packed in different forms (aon_set_bottleneck_fold changed in between)");
    for (int i = 0; i < aon::kNumVanillaParams; ++i)
      if (!grads[l][i]) return fail(AON_E_INVALID, "aon_render_bwd: null gradient pointer");
  }
  hipStream_t caller = stream;
  auto composite_bwd = [&](int l, hipStream_t st) -> int {
    const TrainLevel& L = w.lvl[l];
    const int64_t valid = n_rays * L.S;
    if (int rc = check(hipMemsetAsync(sc.d_raw[l] + valid * 4, 0, (size_t)(L.Np - valid) * 16, st), "aon_render_bwd")) return rc;
    KTimer timer(kCompositeBwd, st, n_rays);
    return check(aon::launch_composite_bwd(L.raw, L.t, rays_d, g_rgb_host[l], g_acc_host ? g_acc_host[l] : nullptr, g_depth_host ? g_depth_host[l] : nullptr,
                                           n_rays, L.S, white_bkgd, g.act(false, l, 0), sc.d_raw[l], st), "aon_render_bwd");
  };
  auto chain_seg = [&](int l) {
    const TrainLevel& L = w.lvl[l];
    return aon::ChainSeg{static_cast<const char*>(pb[l]), reinterpret_cast<const float*>(static_cast<const char*>(pf[l]) + aon::kStreamBytes), sc.d_raw[l],
                         L.masks, nullptr, sc.dplanes[l], nullptr, L.Np};
  };
  // Round 4: the two levels' chains are independent -> ONE persistent launch of two segments on the caller's stream (33 rounds of
  // workgroups instead of 9 + 25 at 4096 x (65 + 193) samples), then the weight gradients of the two levels on the two streams.
  const bool merged = num_levels == 2 && g_bwd_merge.load(std::memory_order_relaxed) != 0;
  const int overlap_mode = g_bwd_overlap.load(std::memory_order_relaxed);   // 0: none; 1: round-3 level streams when not merged; 2: + head reductions on side streams when merged
  const bool early_heads = merged && g_bwd_early_heads.load(std::memory_order_relaxed) != 0;
  // fork: with two levels each runs on its own library stream, ordered after everything already enqueued on the caller's
  // (merged: no LEVEL streams -- chain and weight gradients follow each other on the caller's stream; with equal-cost workgroups filling the
  // chip in every launch the dispatcher's sharing of compute units between streams costs more than the tails it used to fill:
  // profiles/r04_backward_schedules.txt.  The one tail that is left, the chain's last quarter-full round, takes the early head reductions.)
  LevelFork fork(num_levels == 2 && (merged ? (overlap_mode == 2 || early_heads) : overlap_mode != 0), caller, "aon_render_bwd", merged);
  if (fork.rc()) return fork.rc();
  const aon::WgAux* side = early_heads ? fork.aux(0) : nullptr;
  bool early_unjoined = false;   // (the early reductions are on the side stream and the caller's stream has not been told to wait for them)
  if (merged) {
    for (int l = 0; l < 2; ++l)
      if (int rc = composite_bwd(l, caller)) return rc;
    if (side)
      if (int rc = check(hipEventRecord(side->fork, caller), "aon_render_bwd")) return rc;
    {
      const aon::ChainSeg segs[2] = {chain_seg(1), chain_seg(0)};
      KTimer timer(kBwdChain, caller, w.lvl[0].Np + w.lvl[1].Np);
      if (int rc = check(aon::launch_mlp_bwd_chain2(segs, 2, caller), "aon_render_bwd")) return rc;
    }
    if (side) {   // enqueued BEHIND the chain in host order, eligible from the fork event on: the chain's workgroups take the chip first
      int rc = check(hipStreamWaitEvent(side->stream, side->fork, 0), "aon_render_bwd");
      for (int l = 0; l < 2 && !rc; ++l) {
        float* gl[aon::kNumVanillaParams];
        for (int i = 0; i < aon::kNumVanillaParams; ++i) gl[i] = grads[l][i];
        rc = check(aon::launch_vanilla_wgrad(w.lvl[l].planes, sc.dplanes[l], sc.d_raw[l], w.lvl[l].Np, gl, sc.wgrad_ws[l], side->stream, nullptr, pb[l], kWgEarly),
                   "aon_render_bwd");
      }
      const int rj = check(hipEventRecord(side->join, side->stream), "aon_render_bwd");
      // Who waits for the early reductions (round 6, as in aon_art_render_bwd_ex): nobody needs their partial sums before level 0's SECOND
      // stage, so the caller's stream does not wait here -- level 0's grouped kernel follows the chain directly instead of sitting out the
      // reductions' tail (~160 us past the chain's end at 4096 rays) -- and level 0's second stage waits for the event (WgPost::wait_first).
      // AON_EARLY_JOIN=1 in the environment: wait here as before (A/B).
      static const bool join_here = [] { const char* e = std::getenv("AON_EARLY_JOIN"); return e && e[0] == '1'; }();
      early_unjoined = !join_here && !rc && !rj;
      const int rw = early_unjoined ? AON_OK : check(hipStreamWaitEvent(caller, side->join, 0), "aon_render_bwd");
      if (rc || rj || rw) return rc ? rc : (rj ? rj : rw);
    }
  }
  const aon::WgPost wait_early{nullptr, side ? side->join : nullptr};
  // Round 6: ONE second stage for both levels, as in aon_art_render_bwd_ex (the grouped kernels back to back, then one reduce launch and one
  // launch of the six un-folding products; same bits).  Not with other encoding degrees (remap launches between a level's stages).
  // AON_POST_MERGE=0 in the environment: per level as before (A/B).
  static const bool post_merge_env = [] { const char* e = std::getenv("AON_POST_MERGE"); return !(e && e[0] == '0'); }();
  const bool post_merged = merged && post_merge_env && !g.other_degrees && aon::vanilla_wgrad_deferred_bytes() <= aon::kVanillaWgDeferredBytes;
  alignas(16) unsigned char defer_store[2][aon::kVanillaWgDeferredBytes];
  auto deferred = [&](int l) { return reinterpret_cast<aon::VanillaWgDeferred*>(defer_store[l]); };
  for (int l = 0; l < num_levels; ++l) {
    const TrainLevel& L = w.lvl[l];
    stream = merged ? caller : fork.stream(l);
    int rc = AON_OK;
    if (!merged) {
      if ((rc = composite_bwd(l, stream))) return rc;
      const aon::ChainSeg seg = chain_seg(l);
      KTimer timer(kBwdChain, stream, L.Np);
      rc = check(aon::launch_mlp_bwd_chain2(&seg, 1, stream), "aon_render_bwd");
    }
    if (rc) return rc;
    {
      KTimer timer(kWgrad, stream, L.Np);
      float* gl[aon::kNumVanillaParams];
      for (int i = 0; i < aon::kNumVanillaParams; ++i) gl[i] = grads[l][i];
      if (g.other_degrees) {   // the three encoding-fed weights come out in the kernels' 63 / 27-column layout, then lose the empty slots
        gl[0] = sc.grad_tmp[l]; gl[10] = gl[0] + 256 * 63; gl[16] = gl[10] + 256 * (256 + 63);
      }
      rc = check(aon::launch_vanilla_wgrad(L.planes, sc.dplanes[l], sc.d_raw[l], L.Np, gl, sc.wgrad_ws[l], stream, (merged && overlap_mode != 2) ? nullptr : fork.aux(l), pb[l],
                                           side ? kWgRest : kWgAll, (early_unjoined && l == 0 && !post_merged) ? &wait_early : nullptr,
                                           post_merged ? deferred(l) : nullptr), "aon_render_bwd");
      if (early_unjoined && post_merged) {
        if (rc) { (void)hipStreamWaitEvent(caller, side->join, 0); early_unjoined = false; }   // (else: joined in front of the merged second stage below)
      } else if (early_unjoined && l == 0) {
        // (level 0's second stage has been told to wait for the early reductions; if its call failed before that, the caller's stream waits here:
        // no side-stream work is left behind the caller's view of this call)
        if (rc) (void)hipStreamWaitEvent(caller, side->join, 0);
        early_unjoined = false;
      }
      if (!rc && g.other_degrees) {
        const int Lp = g.max_deg - g.min_deg, P = 3 + 6 * Lp, V = 3 + 6 * g.deg_view;
        auto remap = [&](const float* src, float* dst, int rows, int hidden, int Lx, int Lfull, int cols) {
          const int64_t tot = (int64_t)rows * cols;
          remap_enc_cols_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream>>>(src, dst, rows, hidden, Lx, Lfull);
        };
        remap(gl[0], grads[l][0], 256, 0, Lp, 10, P);
        remap(gl[10], grads[l][10], 256, 256, Lp, 10, 256 + P);
        remap(gl[16], grads[l][16], 128, 256, g.deg_view, 4, 256 + V);
        rc = check(hipGetLastError(), "aon_render_bwd");
      }
    }
    if (rc) return rc;
  }
  if (post_merged) {
    if (early_unjoined)
      if (int rc = check(hipStreamWaitEvent(caller, side->join, 0), "aon_render_bwd")) return rc;
    KTimer timer(kWgrad, caller, 0);
    if (int rc = check(aon::launch_vanilla_wgrad_post2(deferred(0), deferred(1), caller), "aon_render_bwd")) return rc;
  }
  if (int rc = fork.join()) return rc;   // the caller's stream continues after both levels
  return AON_OK;
}

int aon_art_render_bwd(const void* packed_bwd_coarse, const void* small_coarse, const void* packed_bwd_fine, const void* small_fine,
                       const float* rays_d, int64_t n_rays, int white_bkgd, int num_levels, const float* const* g_rgb_host,
                       const float* const* g_acc_host, const float* const* g_depth_host, const float* const* params_coarse_host,
                       const float* const* params_fine_host, const float* shape, const float* appearance, const float* articulation,
                       float* const* grads_coarse_host, float* const* grads_fine_host, float* g_shape, float* g_appearance,
                       float* g_articulation, void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream_) {
  return aon_art_render_bwd_ex(packed_bwd_coarse, small_coarse, packed_bwd_fine, small_fine, rays_d, n_rays, white_bkgd, num_levels, g_rgb_host, g_acc_host,
                               g_depth_host, params_coarse_host, params_fine_host, shape, appearance, articulation, grads_coarse_host, grads_fine_host,
                               g_shape, g_appearance, g_articulation, workspace, workspace_bytes, scratch, scratch_bytes, stream_, nullptr);
}
int aon_art_render_bwd_ex(const void* packed_bwd_coarse, const void* small_coarse, const void* packed_bwd_fine, const void* small_fine,
                          const float* rays_d, int64_t n_rays, int white_bkgd, int num_levels, const float* const* g_rgb_host,
                          const float* const* g_acc_host, const float* const* g_depth_host, const float* const* params_coarse_host,
                          const float* const* params_fine_host, const float* shape, const float* appearance, const float* articulation,
                          float* const* grads_coarse_host, float* const* grads_fine_host, float* g_shape, float* g_appearance,
                          float* g_articulation, void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream_,
                          const aon_render_opts* opts) {
  hipStream_t stream = (hipStream_t)stream_;
  Geo g;
  if (const char* bad = make_geo(opts, g)) return fail(AON_E_INVALID, bad);
  if (n_rays <= 0 || (num_levels != 1 && num_levels != 2)) return fail(AON_E_INVALID, "aon_art_render_bwd: bad size / num_levels");
  if (!rays_d || !g_rgb_host || !workspace || !scratch || !grads_coarse_host || !params_coarse_host || !shape || !appearance || !articulation ||
      !g_shape || !g_appearance || !g_articulation)
    return fail(AON_E_INVALID, "aon_art_render_bwd: null pointer");
  if (reinterpret_cast<uintptr_t>(scratch) & 255) return fail(AON_E_INVALID, "aon_art_render_bwd: scratch must be 256-byte aligned");
  const TrainWs w = carve_train(static_cast<char*>(workspace), n_rays, true, num_levels, g);
  if (w.bytes > workspace_bytes) return fail(AON_E_WORKSPACE, "aon_art_render_bwd: workspace smaller than aon_train_workspace_bytes()");
  const TrainScratch sc = carve_scratch(static_cast<char*>(scratch), n_rays, true, num_levels, g);
  if (sc.bytes > scratch_bytes) return fail(AON_E_WORKSPACE, "aon_art_render_bwd: scratch smaller than aon_train_scratch_bytes()");
  const void* pb[2] = {packed_bwd_coarse, packed_bwd_fine};
  const float* sm[2] = {static_cast<const float*>(small_coarse), static_cast<const float*>(small_fine)};
  const float* const* params[2] = {params_coarse_host, params_fine_host};
  float* const* grads[2] = {grads_coarse_host, grads_fine_host};
  for (int l = 0; l < num_levels; ++l) {
    if (!pb[l] || !sm[l] || !grads[l] || !params[l] || !g_rgb_host[l]) return fail(AON_E_INVALID, "aon_art_render_bwd: null level pointer");
    if (aon::stream_form(pb[l]) != aon::stream_form(sm[l]) || aon::stream_form(pb[l]) != aon::stream_form(pb[0]))
      return fail(AON_E_INVALID, "aon_art_render_bwd: transposed stream and per-call block were made in different forms (aon_set_bottleneck_fold changed in between)");
    for (int i = 0; i < 40; ++i)
      if (!grads[l][i] || !params[l][i]) return fail(AON_E_INVALID, "aon_art_render_bwd: null parameter / gradient pointer");
  }
  hipStream_t caller = stream;
  auto composite_bwd = [&](int l, hipStream_t st) -> int {
    const TrainLevel& L = w.lvl[l];
    const int64_t valid = n_rays * L.S;
    if (int rc = check(hipMemsetAsync(sc.d_raw[l] + valid * 4, 0, (size_t)(L.Np - valid) * 16, st), "aon_art_render_bwd")) return rc;
    KTimer timer(kCompositeBwd, st, n_rays);
    return check(aon::launch_composite_bwd(L.raw, L.t, rays_d, g_rgb_host[l], g_acc_host ? g_acc_host[l] : nullptr, g_depth_host ? g_depth_host[l] : nullptr,
                                           n_rays, L.S, white_bkgd, g.act(true, l, 0), sc.d_raw[l], st), "aon_art_render_bwd");
  };
  auto chain_seg = [&](int l) {
    const TrainLevel& L = w.lvl[l];
    return aon::ChainSeg{static_cast<const char*>(pb[l]), sm[l], sc.d_raw[l], L.masks, L.planes, sc.dplanes[l], sc.dxp[l], L.Np};
  };
  // Round 4: the two levels' chains as ONE persistent launch of two segments on the caller's stream (see aon_render_bwd_ex)
  const bool merged = num_levels == 2 && g_bwd_merge.load(std::memory_order_relaxed) != 0;
  // 0: none; 1: round-3 level streams when not merged; 2: + head reductions on side streams when merged.  ARTICULATED network, round 6: mode 1
  // means mode 2 here -- each level's remaining head reductions (six jobs, 70 + 185 us, HBM-bound ordinary blocks) go onto the level's aux
  // stream just BEFORE its grouped kernel and run beside it instead of in front of it: 30.418 -> 30.309 ms per 4096-ray step over eight
  // alternating runs each (all eight below all eight), same bits (tools/grad_hash.py).  Round 4 had measured this form slower (34.5 vs 34.1 ms,
  // profiles/r04_backward_schedules.txt; the kernels have changed since); the vanilla network's three head jobs gain nothing (24.446 vs
  // 24.423 ms) and keep mode 1.  AON_ART_AUX_HEADS=0 in the environment: mode 1 as before (A/B).
  static const bool art_aux_heads = [] { const char* e = std::getenv("AON_ART_AUX_HEADS"); return !(e && e[0] == '0'); }();
  const int overlap_raw = g_bwd_overlap.load(std::memory_order_relaxed);
  const int overlap_mode = (overlap_raw == 1 && art_aux_heads) ? 2 : overlap_raw;
  const bool early_heads = merged && g_bwd_early_heads.load(std::memory_order_relaxed) != 0;
  // fork: see aon_render_bwd_ex (merged: no level streams; the chain's last, quarter-full round takes the early head reductions on a side stream)
  LevelFork fork(num_levels == 2 && (merged ? (overlap_mode == 2 || early_heads) : overlap_mode != 0), caller, "aon_art_render_bwd", merged);
  if (fork.rc()) return fork.rc();
  const aon::WgAux* side = early_heads ? fork.aux(0) : nullptr;
  // Round 6 experiment, OFF by default (AON_POST_ASIDE=1 enables it): level 0's second stage, un-folding products and latent columns
  // (reduce 30 us -> fold 19 us -> finish 15 us in a row, the chip all but idle) on a side stream beside level 1's head reductions and
  // grouped kernel; level 1's own second stage waits for them (its finishing kernel adds onto level 0's latent gradients).  Same bits
  // (tools/grad_hash.py).  Three-run A/Bs read -0.05 ms; six alternating runs each, with the allocator stall of ops._TRAIN_POOL out of
  // the way, read 30.404 vs 30.403 ms -- nothing.  (The 34-42 ms steps first blamed on it were that stall.)  Left off: no gain, and side
  // work beside the start of a persistent launch is where this code has been burnt before (profiles/r04_backward_schedules.txt).
  static const bool post_aside = [] { const char* e = std::getenv("AON_POST_ASIDE"); return e && e[0] == '1'; }();   // (read once)
  // (overlap mode 2 runs level l's head reductions on aux(l) beside its grouped kernel: level 0's aux stream is free again by then)
  const aon::WgAux* post_side = (merged && post_aside && fork.aux(0)) ? (overlap_mode == 2 ? fork.aux(0) : fork.aux(1)) : nullptr;
  const aon::WgPost post0{post_side, nullptr}, post1{nullptr, post_side ? post_side->join : nullptr};
  // Round 6: ONE second stage for both levels.  Level 0's (reduce 20 us -> un-folding products 18 us -> finishing kernel 15 us, plus an
  // event gap either side) used to sit between the two grouped kernels with the chip all but idle; deferred, the grouped kernels follow
  // each other and the levels' reduce blocks, un-folding products and finishing kernels go out as 1 + 1 + 2 launches behind level 1's
  // (launch_art_wgrad_post2; every block does what it did: same bits, tools/grad_hash.py).  Default degrees only (other degrees put
  // remap launches between the stages).  AON_POST_MERGE=0 in the environment: per level as before (A/B).
  static const bool post_merge_env = [] { const char* e = std::getenv("AON_POST_MERGE"); return !(e && e[0] == '0'); }();
  const bool post_merged = merged && post_merge_env && !post_side && g.max_deg - g.min_deg == 10 && g.deg_view == 4 &&
                           aon::art_wgrad_deferred_bytes() <= aon::kArtWgDeferredBytes;   // (the storage below is sized by a constant repeated here)
  alignas(16) unsigned char defer_store[2][aon::kArtWgDeferredBytes];
  auto deferred = [&](int l) { return reinterpret_cast<aon::ArtWgDeferred*>(defer_store[l]); };
  auto level_wgrad = [&](int l, hipStream_t st, const aon::WgAux* aux, int phase) {
    // level 0 writes the latent gradients, level 1 adds its own (both MLPs see the same latents).  Merged schedule (round 6): both levels'
    // finishing kernels run on the caller's stream in level order, so level 1's adds onto level 0's result in place (g = coarse + fine, the
    // bits of the three add launches this replaces); level streams: into a temporary, added after the join
    const bool acc = merged && l == 1;
    float* gs = (l == 0 || acc) ? g_shape : sc.lat_tmp, *ga = (l == 0 || acc) ? g_appearance : sc.lat_tmp + 128, *gt = (l == 0 || acc) ? g_articulation : sc.lat_tmp + 256;
    return check(aon::launch_art_wgrad(w.lvl[l].planes, sc.dplanes[l], sc.d_raw[l], sc.dxp[l], w.lvl[l].Np, params[l], shape, appearance, articulation, grads[l], gs, ga, gt,
                                       sc.wgrad_ws[l], st, aux, g.max_deg - g.min_deg, g.deg_view, pb[l], phase, acc,
                                       (post_side && phase != kWgEarly) ? (l == 0 ? &post0 : &post1) : nullptr,
                                       (post_merged && phase != kWgEarly) ? deferred(l) : nullptr), "aon_art_render_bwd");
  };
  bool early_unjoined = false;   // (the early reductions are on the side stream and the caller's stream has not been told to wait for them)
  if (merged) {
    for (int l = 0; l < 2; ++l)
      if (int rc = composite_bwd(l, caller)) return rc;
    if (side)
      if (int rc = check(hipEventRecord(side->fork, caller), "aon_art_render_bwd")) return rc;
    {
      const aon::ChainSeg segs[2] = {chain_seg(1), chain_seg(0)};
      KTimer timer(kBwdChain, caller, w.lvl[0].Np + w.lvl[1].Np);
      if (int rc = check(aon::launch_art_bwd_chain2(segs, 2, caller), "aon_art_render_bwd")) return rc;
    }
    if (side) {   // enqueued BEHIND the chain in host order, eligible from the fork event on: the chain's workgroups take the chip first
      int rc = check(hipStreamWaitEvent(side->stream, side->fork, 0), "aon_art_render_bwd");
      for (int l = 0; l < 2 && !rc; ++l) rc = level_wgrad(l, side->stream, nullptr, kWgEarly);
      const int rj = check(hipEventRecord(side->join, side->stream), "aon_art_render_bwd");
      // Who waits for the early reductions: nobody needs their partial sums before level 0's SECOND stage (wgrad_reduce), and in overlap
      // mode 2 that stage already waits for this very stream -- level 0's remaining reductions are queued on it (aux(0) == side), behind
      // the early ones, and run_wgrad_plan joins it in front of the second stage.  So the caller's stream does not wait here: level 0's
      // grouped kernel follows the chain directly instead of sitting out the early reductions' tail (~90 us past the chain's end, plus the
      // event round trip).  Mode 1 (remaining reductions on the caller's stream, no later join): waits here as before.
      // AON_EARLY_JOIN=1 in the environment: wait here in mode 2 as well (A/B).
      static const bool join_here = [] { const char* e = std::getenv("AON_EARLY_JOIN"); return e && e[0] == '1'; }();
      early_unjoined = overlap_mode == 2 && !join_here && !rc && !rj;
      const int rw = early_unjoined ? AON_OK : check(hipStreamWaitEvent(caller, side->join, 0), "aon_art_render_bwd");
      if (rc || rj || rw) return rc ? rc : (rj ? rj : rw);
    }
  }
  for (int l = 0; l < num_levels; ++l) {
    const TrainLevel& L = w.lvl[l];
    stream = merged ? caller : fork.stream(l);
    int rc = AON_OK;
    if (!merged) {
      if ((rc = composite_bwd(l, stream))) return rc;
      const aon::ChainSeg seg = chain_seg(l);
      KTimer timer(kBwdChain, stream, L.Np);
      rc = check(aon::launch_art_bwd_chain2(&seg, 1, stream), "aon_art_render_bwd");
    }
    if (rc) return rc;
    {
      KTimer timer(kWgrad, stream, L.Np);
      rc = level_wgrad(l, stream, (merged && overlap_mode != 2) ? nullptr : fork.aux(l), side ? kWgRest : kWgAll);
    }
    if (rc) {
      // level 0's call failed before it joined the side stream (a refused plan): the early reductions are still un-joined -- never leave
      // side-stream work behind the caller's view of this call (the workspace is the caller's to free)
      if (early_unjoined) (void)hipStreamWaitEvent(caller, side->join, 0);
      return rc;
    }
    early_unjoined = false;   // (run_wgrad_plan joined aux(0) == the early reductions' stream in front of level 0's second stage)
  }
  if (post_merged) {
    KTimer timer(kWgrad, caller, 0);
    if (int rc = check(aon::launch_art_wgrad_post2(deferred(0), deferred(1), caller), "aon_art_render_bwd")) return rc;
  }
  if (int rc = fork.join()) return rc;   // the caller's stream continues after both levels
  if (num_levels == 2 && !merged) {
    add_into_kernel<<<dim3(1), dim3(128), 0, caller>>>(g_shape, sc.lat_tmp, 128);
    add_into_kernel<<<dim3(1), dim3(128), 0, caller>>>(g_appearance, sc.lat_tmp + 128, 128);
    add_into_kernel<<<dim3(1), dim3(32), 0, caller>>>(g_articulation, sc.lat_tmp + 256, 32);
    return check(hipGetLastError(), "aon_art_render_bwd");
  }
  return AON_OK;
}

// ---- articulated network (model_autodecoder.py) ----
int64_t aon_art_packed_bytes(void) { return aon::art_stream_bytes(); }
int64_t aon_art_small_bytes(void) { return aon::art_small_bytes(); }

int aon_pack_art_mlp(const float* const* params_host, void* packed, void* stream) {
  return aon_pack_art_mlp_deg(params_host, 0, 10, 4, packed, stream);
}
int aon_pack_art_mlp_deg(const float* const* params_host, int min_deg_point, int max_deg_point, int deg_view, void* packed, void* stream) {
  if (const char* bad = art_degrees_ok(min_deg_point, max_deg_point, deg_view)) return fail(AON_E_INVALID, bad);
  if (!params_host || !packed) return fail(AON_E_INVALID, "aon_pack_art_mlp: null pointer");
  for (int i = 0; i < 40; ++i)
    if (!params_host[i]) return fail(AON_E_INVALID, "aon_pack_art_mlp: null parameter pointer");
  if (reinterpret_cast<uintptr_t>(packed) & 15) return fail(AON_E_INVALID, "aon_pack_art_mlp: packed must be 16-byte aligned");
  return check(aon::launch_pack_art(params_host, static_cast<float*>(packed), (hipStream_t)stream, max_deg_point - min_deg_point, deg_view),
               "aon_pack_art_mlp");
}

int aon_art_prepare(const float* const* params_host, const float* shape, const float* appearance, const float* articulation,
                    void* small, void* stream) {
  return aon_art_prepare_deg(params_host, shape, appearance, articulation, 0, 10, 4, small, stream);
}
int aon_art_prepare_deg(const float* const* params_host, const float* shape, const float* appearance, const float* articulation,
                        int min_deg_point, int max_deg_point, int deg_view, void* small, void* stream) {
  if (const char* bad = art_degrees_ok(min_deg_point, max_deg_point, deg_view)) return fail(AON_E_INVALID, bad);
  if (!params_host || !shape || !appearance || !articulation || !small) return fail(AON_E_INVALID, "aon_art_prepare: null pointer");
  for (int i = 0; i < 40; ++i)
    if (!params_host[i]) return fail(AON_E_INVALID, "aon_art_prepare: null parameter pointer");
  if (reinterpret_cast<uintptr_t>(small) & 15) return fail(AON_E_INVALID, "aon_art_prepare: small must be 16-byte aligned");
  return check(aon::launch_prepare_art(params_host, shape, appearance, articulation, static_cast<float*>(small), (hipStream_t)stream, min_deg_point,
                                       max_deg_point - min_deg_point, deg_view), "aon_art_prepare");
}

// Round 6: everything a training step of a TWO-level articulated model packs, in one call -- both networks' forward streams, per-call blocks
// and transposed streams -- with the four 128 x 256 x 256 fp64 products (W' of each network, once for its forward and once for its transposed
// stream) as ONE launch in front instead of four launches of 17 us each in a row with their pack kernels, then both networks' per-call blocks +
// forward streams as one launch and both transposed streams as another (the step's prologue: 174 -> 68 us, profiles/r06_step_timeline.txt).
// The same kernels on the same operands as the six separate calls: same bytes in every buffer.  packed_bwd_* may be NULL (no backward wanted).
int aon_art_pack_step(const float* const* params_coarse_host, const float* const* params_fine_host, const float* shape, const float* appearance,
                      const float* articulation, int min_deg_point, int max_deg_point, int deg_view, void* packed_coarse, void* small_coarse,
                      void* packed_bwd_coarse, void* packed_fine, void* small_fine, void* packed_bwd_fine, void* stream_) {
  if (const char* bad = art_degrees_ok(min_deg_point, max_deg_point, deg_view)) return fail(AON_E_INVALID, bad);
  if (!params_coarse_host || !params_fine_host || !shape || !appearance || !articulation || !packed_coarse || !small_coarse || !packed_fine || !small_fine)
    return fail(AON_E_INVALID, "aon_art_pack_step: null pointer");
  for (int i = 0; i < 40; ++i)
    if (!params_coarse_host[i] || !params_fine_host[i]) return fail(AON_E_INVALID, "aon_art_pack_step: null parameter pointer");
  for (const void* p : {(const void*)packed_coarse, (const void*)small_coarse, (const void*)packed_fine, (const void*)small_fine, (const void*)packed_bwd_coarse, (const void*)packed_bwd_fine})
    if (reinterpret_cast<uintptr_t>(p) & 15) return fail(AON_E_INVALID, "aon_art_pack_step: buffers must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int Lp = max_deg_point - min_deg_point, Lv = deg_view;
  const float* const* P[2] = {params_coarse_host, params_fine_host};
  float* fwd[2] = {static_cast<float*>(packed_coarse), static_cast<float*>(packed_fine)};
  float* sm[2] = {static_cast<float*>(small_coarse), static_cast<float*>(small_fine)};
  float* bwd[2] = {static_cast<float*>(packed_bwd_coarse), static_cast<float*>(packed_bwd_fine)};
  const int form = aon::fold_default();   // read ONCE: the merged launches below are told, the single-buffer ones cope with either answer
  const bool folded = form == aon::kFormFolded;
  if (folded) {
    aon::FoldGemm jobs[4];
    int n = 0;
    for (int l = 0; l < 2; ++l) jobs[n++] = aon::art_fold_job_fwd(P[l], fwd[l], Lv);
    for (int l = 0; l < 2; ++l)
      if (bwd[l]) jobs[n++] = aon::art_fold_job_bwd(P[l], bwd[l], Lv);
    if (int rc = check(aon::launch_fold_gemms(jobs, n, stream), "aon_art_pack_step")) return rc;
  }
  // (AON_PACK_MERGE=0 in the environment: one launch per network and buffer as before, for A/B)
  static const bool merge = [] { const char* e = std::getenv("AON_PACK_MERGE"); return !(e && e[0] == '0'); }();
  if (merge) {
    if (int rc = check(aon::launch_pack_prepare_art2(P, shape, appearance, articulation, fwd, sm, stream, min_deg_point, Lp, Lv, form), "aon_art_pack_step")) return rc;
    if (bwd[0] && bwd[1]) return check(aon::launch_pack_art_bwd2(P, bwd, stream, Lp, Lv, form), "aon_art_pack_step");
  } else {
    for (int l = 0; l < 2; ++l) {
      if (int rc = check(aon::launch_prepare_art(P[l], shape, appearance, articulation, sm[l], stream, min_deg_point, Lp, Lv), "aon_art_pack_step")) return rc;
      if (int rc = check(aon::launch_pack_art(P[l], fwd[l], stream, Lp, Lv, folded), "aon_art_pack_step")) return rc;
    }
  }
  for (int l = 0; l < 2; ++l)
    if (bwd[l] && !(merge && bwd[0] && bwd[1]))
      if (int rc = check(aon::launch_pack_art_bwd(P[l], bwd[l], stream, Lp, Lv, folded), "aon_art_pack_step")) return rc;
  return AON_OK;
}

int aon_art_mlp_fwd(const void* packed, const void* small, const float* rays_o, const float* rays_d, const float* viewdirs,
                    const float* t_vals, int64_t n_rays, int S, float* raw, void* stream) {
  if (n_rays < 0 || S < 1) return fail(AON_E_INVALID, "aon_art_mlp_fwd: bad size");
  if (n_rays == 0) return AON_OK;
  if (!packed || !small || !rays_o || !rays_d || !viewdirs || !t_vals || !raw) return fail(AON_E_INVALID, "aon_art_mlp_fwd: null pointer");
  if (forms_differ(packed, small)) return fail(AON_E_INVALID, kFormsMsg);
  MlpTimer timer((hipStream_t)stream, n_rays * S);
  return check(aon::launch_art_mlp_fwd(static_cast<const char*>(packed), static_cast<const float*>(small), rays_o, rays_d, viewdirs,
                                       t_vals, n_rays, S, raw, (hipStream_t)stream), "aon_art_mlp_fwd");
}

int aon_art_mlp_fwd_pos(const void* packed, const void* small, const float* pos, const float* viewdirs_enc, int64_t n_rays, int S,
                        float* raw, void* stream) {
  if (n_rays < 0 || S < 1) return fail(AON_E_INVALID, "aon_art_mlp_fwd_pos: bad size");
  if (n_rays == 0) return AON_OK;
  if (!packed || !small || !pos || !viewdirs_enc || !raw) return fail(AON_E_INVALID, "aon_art_mlp_fwd_pos: null pointer");
  if (forms_differ(packed, small)) return fail(AON_E_INVALID, kFormsMsg);
  return check(aon::launch_art_mlp_fwd_pos(static_cast<const char*>(packed), static_cast<const float*>(small), pos, viewdirs_enc,
                                           n_rays, S, raw, (hipStream_t)stream), "aon_art_mlp_fwd_pos");
}

int aon_art_render_fwd_ex(const void* packed_coarse, const void* small_coarse, const void* packed_fine, const void* small_fine,
                          const float* rays_o, const float* rays_d, const float* viewdirs, int64_t n_rays, float near_, float far_,
                          int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride, float* rgb_c,
                          float* acc_c, float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace,
                          int64_t workspace_bytes, void* stream, const aon_render_opts* opts) {
  const NetRef c{true, packed_coarse, static_cast<const float*>(small_coarse)}, f{true, packed_fine, static_cast<const float*>(small_fine)};
  return render_impl("aon_art_render_fwd", c, f, rays_o, rays_d, viewdirs, n_rays, near_, far_, white_bkgd, num_levels, t_rand, u,
                     u_stride, rgb_c, acc_c, depth_c, rgb_f, acc_f, depth_f, workspace, workspace_bytes, (hipStream_t)stream, opts);
}
int aon_art_render_fwd(const void* packed_coarse, const void* small_coarse, const void* packed_fine, const void* small_fine,
                       const float* rays_o, const float* rays_d, const float* viewdirs, int64_t n_rays, float near_, float far_,
                       int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride, float* rgb_c,
                       float* acc_c, float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace,
                       int64_t workspace_bytes, void* stream) {
  return aon_art_render_fwd_ex(packed_coarse, small_coarse, packed_fine, small_fine, rays_o, rays_d, viewdirs, n_rays, near_, far_, white_bkgd,
                               num_levels, t_rand, u, u_stride, rgb_c, acc_c, depth_c, rgb_f, acc_f, depth_f, workspace, workspace_bytes, stream,
                               nullptr);
}


// ---------------------------------------------------------------------------------------------------------------------
// NeRFMLP of any constructor geometry: layer-wise GEMM engine (csrc/aon_gmlp.hip)
// ---------------------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

struct GG {   // aon_mlp_geometry, validated
  int P, V, D, W, Dc, Wc, skip, Crgb, Cd, min_deg, max_deg, deg_view, in_ch, in_ch_view;
  // row strides of the engine's OWN encoding buffers: padded to whole 8-float groups (zeros) so the GEMM reads 16-byte pieces only
  int ldP() const { return (P + 7) & ~7; }
  int ldV() const { return (V + 7) & ~7; }
  int nparams() const { return 2 * (D + Dc + 3); }
  int pts(int l) const { return 2 * l; }
  int view(int i) const { return 2 * (D + i); }
  int bott() const { return 2 * (D + Dc); }
  int dens() const { return 2 * (D + Dc) + 2; }
  int rgb() const { return 2 * (D + Dc) + 4; }
  bool cat_before(int l) const { return l >= 2 && (l - 1) % skip == 0; }   // layer l reads cat([H_{l-1}, inputs]) (model.py:75-76, :103-104)
  int in_width(int l) const { return l == 0 ? P : (cat_before(l) ? W + P : W); }
};

const char* make_gg(const aon_mlp_geometry* g, GG& o) {
  if (!g) return "null geometry";
  if (g->netdepth < 1 || g->netwidth < 1 || g->netdepth_condition < 1 || g->netwidth_condition < 1 || g->skip_layer < 1 || g->input_ch < 1 ||
      g->input_ch_view < 1 || g->num_rgb_channels < 1 || g->num_density_channels < 1 || g->max_deg_point < g->min_deg_point || g->deg_view < 0)
    return "bad NeRFMLP geometry";
  if (g->netdepth > 64 || g->netdepth_condition > 64 || g->netwidth > 4096 || g->netwidth_condition > 4096) return "NeRFMLP geometry too large";
  o.min_deg = g->min_deg_point; o.max_deg = g->max_deg_point; o.deg_view = g->deg_view; o.in_ch = g->input_ch; o.in_ch_view = g->input_ch_view;
  o.P = ((g->max_deg_point - g->min_deg_point) * 2 + 1) * g->input_ch;
  o.V = (g->deg_view * 2 + 1) * g->input_ch_view;
  o.D = g->netdepth; o.W = g->netwidth; o.Dc = g->netdepth_condition; o.Wc = g->netwidth_condition; o.skip = g->skip_layer;
  o.Crgb = g->num_rgb_channels; o.Cd = g->num_density_channels;
  if (o.D - 1 > 0 && (o.D - 1) % o.skip == 0)
    return "the last trunk layer would concatenate the encoding: the reference's forward fails on this geometry (model.py:90 vs :103-104)";
  return nullptr;
}

// per-sample activation buffers of one MLP evaluation over M samples of n rays
struct GActs {
  float* E;        // M x P, row stride ldE (caller-owned when the encoding is given: then ldE = P)
  float* cond;     // n x V, row stride ldC
  int ldE, ldC;
  float* H[64];    // trunk outputs, M x W each (inference: two buffers alternate)
  float* bott;     // M x W
  float* Vh[64];   // view-branch outputs, M x Wc
};

int gmlp_forward(const GG& g, const float* const* p, const GActs& a, int64_t n_rays, int S, float* raw_rgb, int64_t ld_rgb, float* raw_density,
                 int64_t ld_density, hipStream_t stream, const char* who) {
  const int64_t M = n_rays * S;
  MlpTimer timer(stream, M);
  for (int l = 0; l < g.D; ++l) {
    aon::GemmArgs ga{};
    const int ldw = g.in_width(l);
    if (l == 0) ga.seg[0] = {a.E, a.ldE, 1, p[g.pts(0)], ldw, g.P};
    else ga.seg[0] = {a.H[l - 1], g.W, 1, p[g.pts(l)], ldw, g.W};
    ga.nseg = 1;
    if (g.cat_before(l)) { ga.seg[1] = {a.E, a.ldE, 1, p[g.pts(l)] + g.W, ldw, g.P}; ga.nseg = 2; }
    ga.bias = p[g.pts(l) + 1]; ga.Y = a.H[l]; ga.ldy = g.W; ga.M = M; ga.N = g.W; ga.epi = 1;
    if (int rc = check(aon::launch_gemm_tn(ga, stream), who)) return rc;
  }
  const float* x = a.H[g.D - 1];
  {
    aon::GemmArgs ga{};
    ga.seg[0] = {x, g.W, 1, p[g.dens()], g.W, g.W}; ga.nseg = 1;
    ga.bias = p[g.dens() + 1]; ga.Y = raw_density; ga.ldy = ld_density; ga.M = M; ga.N = g.Cd; ga.epi = 0;
    if (int rc = check(aon::launch_gemm_tn(ga, stream), who)) return rc;
    ga.seg[0].W = p[g.bott()]; ga.bias = p[g.bott() + 1]; ga.Y = a.bott; ga.ldy = g.W; ga.N = g.W;
    if (int rc = check(aon::launch_gemm_tn(ga, stream), who)) return rc;
  }
  for (int i = 0; i < g.Dc; ++i) {
    aon::GemmArgs ga{};
    if (i == 0) {
      ga.seg[0] = {a.bott, g.W, 1, p[g.view(0)], g.W + g.V, g.W};
      ga.seg[1] = {a.cond, a.ldC, S, p[g.view(0)] + g.W, g.W + g.V, g.V};   // condition_tile (model.py:107-110): the ray's row
      ga.nseg = 2;
    } else {
      ga.seg[0] = {a.Vh[i - 1], g.Wc, 1, p[g.view(i)], g.Wc, g.Wc}; ga.nseg = 1;
    }
    ga.bias = p[g.view(i) + 1]; ga.Y = a.Vh[i]; ga.ldy = g.Wc; ga.M = M; ga.N = g.Wc; ga.epi = 1;
    if (int rc = check(aon::launch_gemm_tn(ga, stream), who)) return rc;
  }
  {
    aon::GemmArgs ga{};
    ga.seg[0] = {a.Vh[g.Dc - 1], g.Wc, 1, p[g.rgb()], g.Wc, g.Wc}; ga.nseg = 1;
    ga.bias = p[g.rgb() + 1]; ga.Y = raw_rgb; ga.ldy = ld_rgb; ga.M = M; ga.N = g.Crgb; ga.epi = 0;
    if (int rc = check(aon::launch_gemm_tn(ga, stream), who)) return rc;
  }
  return AON_OK;
}

struct Carver {
  char* base; int64_t off = 0;
  explicit Carver(void* b) : base(static_cast<char*>(b)) {}
  float* f(int64_t floats) { char* p = base + off; off += align_up(floats * 4, 256); return reinterpret_cast<float*>(p); }
};

// activation buffers: train = every layer keeps its own output, else two alternate
GActs carve_acts(Carver& c, const GG& g, int64_t M, int64_t n_rays, bool train, bool own_enc) {
  GActs a{};
  a.ldE = g.P; a.ldC = g.V;
  if (own_enc) { a.ldE = g.ldP(); a.ldC = g.ldV(); a.E = c.f(M * a.ldE); a.cond = c.f(n_rays * a.ldC); }
  if (train) {
    for (int l = 0; l < g.D; ++l) a.H[l] = c.f(M * g.W);
    for (int i = 0; i < g.Dc; ++i) a.Vh[i] = c.f(M * g.Wc);
  } else {
    float* h0 = c.f(M * g.W); float* h1 = c.f(M * g.W);
    for (int l = 0; l < g.D; ++l) a.H[l] = (l & 1) ? h1 : h0;
    float* v0 = c.f(M * g.Wc); float* v1 = c.f(M * g.Wc);
    for (int i = 0; i < g.Dc; ++i) a.Vh[i] = (i & 1) ? v1 : v0;
  }
  a.bott = c.f(M * g.W);
  return a;
}

// whole-path workspace of a chunk of n rays (inference)
struct GWs { float* t_c; float* w_c; float* t_f; float* raw; float* coords; GActs acts; int64_t bytes; };
GWs carve_grender(void* base, const GG& g, const Geo& geo, int64_t n) {
  Carver c(base);
  GWs w{};
  w.t_c = c.f(n * geo.Sc); w.w_c = c.f(n * geo.Sc); w.t_f = c.f(n * geo.Sf); w.raw = c.f(n * geo.Sf * 4); w.coords = c.f(n * geo.Sf * 3);
  w.acts = carve_acts(c, g, n * geo.Sf, n, false, true);
  w.bytes = c.off;
  return w;
}

// encodings of one level: cast_rays + pos_enc of the samples, pos_enc of the view directions (model.py:175-180)
int g_encode(const GG& g, const float* o, const float* d, const float* v, const float* t, int64_t n, int S, float* coords, const GActs& a,
             hipStream_t stream, const char* who) {
  if (int rc = check(aon::launch_cast_rays(t, o, d, n, S, coords, stream), who)) return rc;
  if (int rc = check(aon::launch_pos_enc(coords, n * S, g.min_deg, g.max_deg, a.E, stream, a.ldE), who)) return rc;
  return check(aon::launch_pos_enc(v, n, 0, g.deg_view, a.cond, stream, a.ldC), who);
}

const char* whole_path_ok(const GG& g) {
  if (g.in_ch != 3 || g.in_ch_view != 3 || g.Crgb != 3 || g.Cd != 1)
    return "NeRF.forward needs input_ch = input_ch_view = 3, num_rgb_channels = 3, num_density_channels = 1";
  return nullptr;
}

// training: what one level's forward leaves for the backward
struct GTrainLevel { float* t; float* raw; float* coords; GActs acts; int S; int64_t M; };
struct GTrainWs { GTrainLevel lvl[2]; float* w_c; int64_t bytes; };
GTrainWs carve_gtrain(void* base, const GG& g, const Geo& geo, int64_t n, int num_levels) {
  Carver c(base);
  GTrainWs w{};
  for (int l = 0; l < num_levels; ++l) {
    GTrainLevel& L = w.lvl[l];
    L.S = geo.S(l); L.M = n * L.S;
    L.t = c.f(L.M); L.raw = c.f(L.M * 4); L.coords = c.f(L.M * 3);
    L.acts = carve_acts(c, g, L.M, n, true, true);
  }
  w.w_c = c.f(n * geo.Sc);
  w.bytes = c.off;
  return w;
}
struct GScratch { float* d_raw; float* dz[2]; float* dbott; float* dv[2]; float* wt; float* part; int64_t bytes; };
GScratch carve_gscratch(void* base, const GG& g, const Geo& geo, int64_t n, int num_levels) {
  Carver c(base);
  GScratch s{};
  const int64_t M = n * geo.S(num_levels - 1);   // the larger level; the levels run one after the other
  s.d_raw = c.f(M * 4);
  s.dz[0] = c.f(M * g.W); s.dz[1] = c.f(M * g.W); s.dbott = c.f(M * g.W);
  s.dv[0] = c.f(M * g.Wc); s.dv[1] = c.f(M * g.Wc);
  const int64_t wmax = (int64_t)(g.W > g.Wc ? g.W : g.Wc);
  int64_t wt = (int64_t)g.W * g.W + (int64_t)g.Cd * g.W;   // bottleneck + density head together; everything else one at a time
  for (int64_t cand : {(int64_t)g.Crgb * g.Wc, (int64_t)g.Wc * g.Wc, (int64_t)g.Wc * g.W}) wt = cand > wt ? cand : wt;
  s.wt = c.f(wt);
  int64_t part = 0;
  auto need = [&](int N, int K) { const int64_t f = aon::wgrad_part_floats(M, N, K); if (f > part) part = f; };
  need(g.W, g.P); need(g.W, g.W); need(g.Wc, g.W); need(g.Wc, g.V); need(g.Wc, g.Wc); need(g.Crgb, g.Wc); need(g.Cd, g.W);
  int64_t nmax = wmax;                                  // widest bias vector: the column-sum partials are 512 x N DOUBLES
  for (int64_t cand : {(int64_t)g.Crgb, (int64_t)g.Cd}) nmax = cand > nmax ? cand : nmax;
  const int64_t cs = 2 * 512 * nmax;
  s.part = c.f(part > cs ? part : cs);
  s.bytes = c.off;
  return s;
}

// backward of one level: parameter gradients (order / shapes of the params array) from d_raw (M x 4)
int gmlp_backward(const GG& g, const float* const* p, float* const* grads, const GActs& a, const GScratch& sc, int64_t n_rays, int S,
                  hipStream_t stream, const char* who) {
  const int64_t M = n_rays * S;
  const float* d_rgb = sc.d_raw;        // (M, 3) with row stride 4
  const float* d_sig = sc.d_raw + 3;    // (M, 1) with row stride 4
  int rc;
  auto wgrad = [&](const float* dZ, int64_t ldz, int N, const float* X, int64_t ldx, int rowdiv, int K, float* dW, int64_t ldd) {
    KTimer timer(kWgrad, stream, M);
    return check(aon::launch_wgrad_nk(dZ, ldz, X, ldx, rowdiv, M, N, K, dW, ldd, sc.part, stream), who);
  };
  auto bgrad = [&](const float* dZ, int64_t ldz, int N, float* db) { return check(aon::launch_colsum(dZ, ldz, M, N, db, sc.part, stream), who); };
  // dX[M x K] = dZ[M x N] . W[N x K0:K0+K]  (+ a second product), masked by `mask` > 0 when given
  auto bdata = [&](const float* dZ, int64_t ldz, int N, const float* Wt, int K, const float* dZ2, int64_t ldz2, int N2, const float* Wt2,
                   const float* mask, float* dX) {
    KTimer timer(kBwdChain, stream, M);
    aon::GemmArgs ga{};
    ga.seg[0] = {dZ, ldz, 1, Wt, N, N}; ga.nseg = 1;
    if (dZ2) { ga.seg[1] = {dZ2, ldz2, 1, Wt2, N2, N2}; ga.nseg = 2; }
    ga.bias = nullptr; ga.Y = dX; ga.ldy = K; ga.M = M; ga.N = K; ga.epi = mask ? 2 : 0; ga.aux = mask; ga.ldaux = K;
    return check(aon::launch_gemm_tn(ga, stream), who);
  };
  // rgb head
  if ((rc = wgrad(d_rgb, 4, g.Crgb, a.Vh[g.Dc - 1], g.Wc, 1, g.Wc, grads[g.rgb()], g.Wc))) return rc;
  if ((rc = bgrad(d_rgb, 4, g.Crgb, grads[g.rgb() + 1]))) return rc;
  if ((rc = check(aon::launch_transpose(p[g.rgb()], g.Wc, g.Crgb, g.Wc, sc.wt, stream), who))) return rc;
  float* dv = sc.dv[(g.Dc - 1) & 1];
  if ((rc = bdata(d_rgb, 4, g.Crgb, sc.wt, g.Wc, nullptr, 0, 0, nullptr, a.Vh[g.Dc - 1], dv))) return rc;   // dZ of the last view layer
  // view branch
  for (int i = g.Dc - 1; i >= 1; --i) {
    if ((rc = wgrad(dv, g.Wc, g.Wc, a.Vh[i - 1], g.Wc, 1, g.Wc, grads[g.view(i)], g.Wc))) return rc;
    if ((rc = bgrad(dv, g.Wc, g.Wc, grads[g.view(i) + 1]))) return rc;
    if ((rc = check(aon::launch_transpose(p[g.view(i)], g.Wc, g.Wc, g.Wc, sc.wt, stream), who))) return rc;
    float* nx = sc.dv[(i - 1) & 1];
    if ((rc = bdata(dv, g.Wc, g.Wc, sc.wt, g.Wc, nullptr, 0, 0, nullptr, a.Vh[i - 1], nx))) return rc;
    dv = nx;
  }
  {
    const int ldw = g.W + g.V;
    if ((rc = wgrad(dv, g.Wc, g.Wc, a.bott, g.W, 1, g.W, grads[g.view(0)], ldw))) return rc;
    if ((rc = wgrad(dv, g.Wc, g.Wc, a.cond, a.ldC, S, g.V, grads[g.view(0)] + g.W, ldw))) return rc;
    if ((rc = bgrad(dv, g.Wc, g.Wc, grads[g.view(0) + 1]))) return rc;
    if ((rc = check(aon::launch_transpose(p[g.view(0)], ldw, g.Wc, g.W, sc.wt, stream), who))) return rc;   // the bottleneck columns only
    if ((rc = bdata(dv, g.Wc, g.Wc, sc.wt, g.W, nullptr, 0, 0, nullptr, nullptr, sc.dbott))) return rc;      // no activation on the bottleneck
  }
  // bottleneck + density heads -> the last trunk output
  const float* x = a.H[g.D - 1];
  if ((rc = wgrad(sc.dbott, g.W, g.W, x, g.W, 1, g.W, grads[g.bott()], g.W))) return rc;
  if ((rc = bgrad(sc.dbott, g.W, g.W, grads[g.bott() + 1]))) return rc;
  if ((rc = wgrad(d_sig, 4, g.Cd, x, g.W, 1, g.W, grads[g.dens()], g.W))) return rc;
  if ((rc = bgrad(d_sig, 4, g.Cd, grads[g.dens() + 1]))) return rc;
  {
    float* wt2 = sc.wt + (int64_t)g.W * g.W;
    if ((rc = check(aon::launch_transpose(p[g.bott()], g.W, g.W, g.W, sc.wt, stream), who))) return rc;
    if ((rc = check(aon::launch_transpose(p[g.dens()], g.W, g.Cd, g.W, wt2, stream), who))) return rc;
    if ((rc = bdata(sc.dbott, g.W, g.W, sc.wt, g.W, d_sig, 4, g.Cd, wt2, x, sc.dz[(g.D - 1) & 1]))) return rc;
  }
  // trunk
  for (int l = g.D - 1; l >= 0; --l) {
    const float* dz = sc.dz[l & 1];
    const int ldw = g.in_width(l);
    if (l == 0) {
      if ((rc = wgrad(dz, g.W, g.W, a.E, a.ldE, 1, g.P, grads[g.pts(0)], ldw))) return rc;
    } else {
      if ((rc = wgrad(dz, g.W, g.W, a.H[l - 1], g.W, 1, g.W, grads[g.pts(l)], ldw))) return rc;
      if (g.cat_before(l) && (rc = wgrad(dz, g.W, g.W, a.E, a.ldE, 1, g.P, grads[g.pts(l)] + g.W, ldw))) return rc;
    }
    if ((rc = bgrad(dz, g.W, g.W, grads[g.pts(l) + 1]))) return rc;
    if (l > 0) {
      if ((rc = check(aon::launch_transpose(p[g.pts(l)], ldw, g.W, g.W, sc.wt, stream), who))) return rc;   // the hidden columns only
      if ((rc = bdata(dz, g.W, g.W, sc.wt, g.W, nullptr, 0, 0, nullptr, a.H[l - 1], sc.dz[(l - 1) & 1]))) return rc;
    }
  }
  return AON_OK;
}

int check_params(const GG& g, const float* const* p, const char* what) {
  if (!p) return fail(AON_E_INVALID, what);
  for (int i = 0; i < g.nparams(); ++i)
    if (!p[i]) return fail(AON_E_INVALID, what);
  return AON_OK;
}

}  // namespace

extern "C" {

void aon_mlp_geometry_init(aon_mlp_geometry* g) {
  if (!g) return;
  *g = aon_mlp_geometry{0, 10, 4, 8, 256, 1, 128, 4, 3, 3, 3, 1};
}

int aon_gmlp_param_count(const aon_mlp_geometry* geom) {
  GG g;
  if (const char* bad = make_gg(geom, g)) return fail(AON_E_INVALID, bad);
  return g.nparams();
}

int64_t aon_gmlp_workspace_bytes(const aon_mlp_geometry* geom, int64_t n_samples) {
  GG g;
  if (const char* bad = make_gg(geom, g)) return fail(AON_E_INVALID, bad);
  Carver c(nullptr);
  carve_acts(c, g, n_samples < 1 ? 1 : n_samples, 1, false, false);
  return c.off;
}

int aon_gmlp_fwd(const aon_mlp_geometry* geom, const float* const* params_host, const float* samples_enc, const float* viewdirs_enc,
                 int64_t n_rays, int S, float* raw_rgb, float* raw_density, void* workspace, int64_t workspace_bytes, void* stream) {
  GG g;
  if (const char* bad = make_gg(geom, g)) return fail(AON_E_INVALID, bad);
  if (n_rays < 0 || S < 1) return fail(AON_E_INVALID, "aon_gmlp_fwd: bad size");
  if (n_rays == 0) return AON_OK;
  if (int rc = check_params(g, params_host, "aon_gmlp_fwd: null parameter pointer")) return rc;
  if (!samples_enc || !viewdirs_enc || !raw_rgb || !raw_density || !workspace) return fail(AON_E_INVALID, "aon_gmlp_fwd: null pointer");
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(AON_E_INVALID, "aon_gmlp_fwd: workspace must be 256-byte aligned");
  Carver c(workspace);
  GActs a = carve_acts(c, g, n_rays * S, n_rays, false, false);
  if (c.off > workspace_bytes) return fail(AON_E_WORKSPACE, "aon_gmlp_fwd: workspace smaller than aon_gmlp_workspace_bytes()");
  a.E = const_cast<float*>(samples_enc); a.cond = const_cast<float*>(viewdirs_enc);
  return gmlp_forward(g, params_host, a, n_rays, S, raw_rgb, g.Crgb, raw_density, g.Cd, (hipStream_t)stream, "aon_gmlp_fwd");
}

int64_t aon_grender_workspace_bytes(const aon_mlp_geometry* geom, int64_t n_rays, const aon_render_opts* opts) {
  GG g; Geo geo;
  if (const char* bad = make_gg(geom, g)) return fail(AON_E_INVALID, bad);
  if (const char* bad = make_geo(opts, geo, true)) return fail(AON_E_INVALID, bad);
  return carve_grender(nullptr, g, geo, n_rays < 1 ? 1 : n_rays).bytes;
}

int aon_grender_fwd(const aon_mlp_geometry* geom, const float* const* params_coarse_host, const float* const* params_fine_host,
                    const float* rays_o, const float* rays_d, const float* viewdirs, int64_t n_rays, float near_, float far_,
                    int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride, float* rgb_c, float* acc_c,
                    float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream_,
                    const aon_render_opts* opts) {
  const char* who = "aon_grender_fwd";
  hipStream_t stream = (hipStream_t)stream_;
  GG g; Geo geo;
  if (const char* bad = make_gg(geom, g)) return fail(AON_E_INVALID, bad);
  if (const char* bad = whole_path_ok(g)) return fail(AON_E_INVALID, bad);
  if (const char* bad = make_geo(opts, geo, true)) return fail(AON_E_INVALID, bad);
  if (n_rays < 0 || (num_levels != 1 && num_levels != 2)) return fail(AON_E_INVALID, "aon_grender_fwd: bad size / num_levels");
  if (n_rays == 0) return AON_OK;
  if (int rc = check_params(g, params_coarse_host, "aon_grender_fwd: null parameter pointer")) return rc;
  if (num_levels == 2)
    if (int rc = check_params(g, params_fine_host, "aon_grender_fwd: null parameter pointer")) return rc;
  if (!rays_o || !rays_d || !viewdirs || !rgb_c || !acc_c || !depth_c || !workspace) return fail(AON_E_INVALID, "aon_grender_fwd: null pointer");
  if (num_levels == 2 && (!rgb_f || !acc_f || !depth_f || !u || (u_stride != 0 && u_stride < geo.nf)))
    return fail(AON_E_INVALID, "aon_grender_fwd: null fine-level pointer / bad u_stride");
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(AON_E_INVALID, "aon_grender_fwd: workspace must be 256-byte aligned");
  int64_t chunk = n_rays;
  if (carve_grender(nullptr, g, geo, chunk).bytes > workspace_bytes) {
    const int64_t one = carve_grender(nullptr, g, geo, 1).bytes, two = carve_grender(nullptr, g, geo, 1025).bytes;
    const int64_t per_ray = (two - one) / 1024 + 1;
    chunk = (workspace_bytes - one) / per_ray;
    while (chunk > 0 && carve_grender(nullptr, g, geo, chunk).bytes > workspace_bytes) --chunk;
    if (chunk < 1) return fail(AON_E_WORKSPACE, "aon_grender_fwd: workspace smaller than aon_grender_workspace_bytes(geom, 1, opts)");
  }
  const GWs w = carve_grender(workspace, g, geo, chunk);
  const float* const* params[2] = {params_coarse_host, params_fine_host};
  for (int64_t r0 = 0; r0 < n_rays; r0 += chunk) {
    const int64_t n = n_rays - r0 < chunk ? n_rays - r0 : chunk;
    const float* o = rays_o + r0 * 3; const float* d = rays_d + r0 * 3; const float* v = viewdirs + r0 * 3;
    const float* uu = u_stride ? u + r0 * u_stride : u;
    float* rgb[2] = {rgb_c + r0 * 3, rgb_f ? rgb_f + r0 * 3 : nullptr};
    float* acc[2] = {acc_c + r0, acc_f ? acc_f + r0 : nullptr};
    float* dep[2] = {depth_c + r0, depth_f ? depth_f + r0 : nullptr};
    for (int l = 0; l < num_levels; ++l) {
      const int S = geo.S(l);
      float* t = l == 0 ? w.t_c : w.t_f;
      int rc;
      if (l == 0) {
        KTimer timer(kSampleT, stream, n);
        rc = check(aon::launch_sample_along_rays(o, d, n, geo.Sc, near_, far_, t_rand ? t_rand + r0 * geo.Sc : nullptr, t, nullptr, stream, geo.lindisp,
                                                 geo.inv_near, geo.inv_far), who);
      } else {
        KTimer timer(kSamplePdf, stream, n);
        rc = check(geo.default_sizes ? aon::launch_sample_pdf(nullptr, w.w_c + 1, kSc, w.t_c, uu, u_stride, n, nullptr, t, stream)
                                     : aon::launch_sample_pdf_n(nullptr, w.w_c + 1, geo.Sc, w.t_c, uu, u_stride, n, geo.Sc - 1, geo.nf, geo.Sc, nullptr, t,
                                                                stream), who);
      }
      if (rc) return rc;
      if ((rc = g_encode(g, o, d, v, t, n, S, w.coords, w.acts, stream, who))) return rc;
      if ((rc = gmlp_forward(g, params[l], w.acts, n, S, w.raw, 4, w.raw + 3, 4, stream, who))) return rc;
      {
        KTimer timer(kComposite, stream, n);
        rc = check(aon::launch_composite(w.raw, 4, w.raw + 3, 4, t, d, n, S, white_bkgd, geo.act(false, l, r0), rgb[l], acc[l], dep[l],
                                         (l == 0 && num_levels == 2) ? w.w_c : nullptr, stream), who);
      }
      if (rc) return rc;
    }
  }
  return AON_OK;
}

int64_t aon_grender_train_workspace_bytes(const aon_mlp_geometry* geom, int64_t n_rays, int num_levels, const aon_render_opts* opts) {
  GG g; Geo geo;
  if (const char* bad = make_gg(geom, g)) return fail(AON_E_INVALID, bad);
  if (const char* bad = make_geo(opts, geo, true)) return fail(AON_E_INVALID, bad);
  return carve_gtrain(nullptr, g, geo, n_rays < 1 ? 1 : n_rays, num_levels == 1 ? 1 : 2).bytes;
}
int64_t aon_grender_train_scratch_bytes(const aon_mlp_geometry* geom, int64_t n_rays, int num_levels, const aon_render_opts* opts) {
  GG g; Geo geo;
  if (const char* bad = make_gg(geom, g)) return fail(AON_E_INVALID, bad);
  if (const char* bad = make_geo(opts, geo, true)) return fail(AON_E_INVALID, bad);
  return carve_gscratch(nullptr, g, geo, n_rays < 1 ? 1 : n_rays, num_levels == 1 ? 1 : 2).bytes;
}

int aon_grender_fwd_train(const aon_mlp_geometry* geom, const float* const* params_coarse_host, const float* const* params_fine_host,
                          const float* rays_o, const float* rays_d, const float* viewdirs, int64_t n, float near_, float far_,
                          int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride, float* rgb_c,
                          float* acc_c, float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace,
                          int64_t workspace_bytes, void* stream_, const aon_render_opts* opts) {
  const char* who = "aon_grender_fwd_train";
  hipStream_t stream = (hipStream_t)stream_;
  GG g; Geo geo;
  if (const char* bad = make_gg(geom, g)) return fail(AON_E_INVALID, bad);
  if (const char* bad = whole_path_ok(g)) return fail(AON_E_INVALID, bad);
  if (const char* bad = make_geo(opts, geo, true)) return fail(AON_E_INVALID, bad);
  if (geo.Sf > 512) return fail(AON_E_INVALID, "aon_grender_fwd_train: more than 512 samples per ray at the fine level");
  if (n <= 0 || (num_levels != 1 && num_levels != 2)) return fail(AON_E_INVALID, "aon_grender_fwd_train: bad size / num_levels");
  if (int rc = check_params(g, params_coarse_host, "aon_grender_fwd_train: null parameter pointer")) return rc;
  if (num_levels == 2)
    if (int rc = check_params(g, params_fine_host, "aon_grender_fwd_train: null parameter pointer")) return rc;
  if (!rays_o || !rays_d || !viewdirs || !rgb_c || !acc_c || !depth_c || !workspace) return fail(AON_E_INVALID, "aon_grender_fwd_train: null pointer");
  if (num_levels == 2 && (!rgb_f || !acc_f || !depth_f || !u || (u_stride != 0 && u_stride < geo.nf)))
    return fail(AON_E_INVALID, "aon_grender_fwd_train: null fine-level pointer / bad u_stride");
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(AON_E_INVALID, "aon_grender_fwd_train: workspace must be 256-byte aligned");
  const GTrainWs w = carve_gtrain(workspace, g, geo, n, num_levels);
  if (w.bytes > workspace_bytes) return fail(AON_E_WORKSPACE, "aon_grender_fwd_train: workspace smaller than aon_grender_train_workspace_bytes()");
  const float* const* params[2] = {params_coarse_host, params_fine_host};
  float* rgb[2] = {rgb_c, rgb_f}; float* acc[2] = {acc_c, acc_f}; float* dep[2] = {depth_c, depth_f};
  for (int l = 0; l < num_levels; ++l) {
    const GTrainLevel& L = w.lvl[l];
    int rc;
    if (l == 0) {
      KTimer timer(kSampleT, stream, n);
      rc = check(aon::launch_sample_along_rays(rays_o, rays_d, n, geo.Sc, near_, far_, t_rand, L.t, nullptr, stream, geo.lindisp, geo.inv_near,
                                               geo.inv_far), who);
    } else {
      KTimer timer(kSamplePdf, stream, n);
      rc = check(geo.default_sizes ? aon::launch_sample_pdf(nullptr, w.w_c + 1, kSc, w.lvl[0].t, u, u_stride, n, nullptr, L.t, stream)
                                   : aon::launch_sample_pdf_n(nullptr, w.w_c + 1, geo.Sc, w.lvl[0].t, u, u_stride, n, geo.Sc - 1, geo.nf, geo.Sc, nullptr,
                                                              L.t, stream), who);
    }
    if (rc) return rc;
    if ((rc = g_encode(g, rays_o, rays_d, viewdirs, L.t, n, L.S, L.coords, L.acts, stream, who))) return rc;
    if ((rc = gmlp_forward(g, params[l], L.acts, n, L.S, L.raw, 4, L.raw + 3, 4, stream, who))) return rc;
    {
      KTimer timer(kComposite, stream, n);
      rc = check(aon::launch_composite(L.raw, 4, L.raw + 3, 4, L.t, rays_d, n, L.S, white_bkgd, geo.act(false, l, 0), rgb[l], acc[l], dep[l],
                                       (l == 0 && num_levels == 2) ? w.w_c : nullptr, stream), who);
    }
    if (rc) return rc;
  }
  return AON_OK;
}

int aon_grender_bwd(const aon_mlp_geometry* geom, const float* const* params_coarse_host, const float* const* params_fine_host,
                    const float* rays_d, int64_t n, int white_bkgd, int num_levels, const float* const* g_rgb_host,
                    const float* const* g_acc_host, const float* const* g_depth_host, float* const* grads_coarse_host,
                    float* const* grads_fine_host, void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes,
                    void* stream_, const aon_render_opts* opts) {
  const char* who = "aon_grender_bwd";
  hipStream_t stream = (hipStream_t)stream_;
  GG g; Geo geo;
  if (const char* bad = make_gg(geom, g)) return fail(AON_E_INVALID, bad);
  if (const char* bad = whole_path_ok(g)) return fail(AON_E_INVALID, bad);
  if (const char* bad = make_geo(opts, geo, true)) return fail(AON_E_INVALID, bad);
  if (n <= 0 || (num_levels != 1 && num_levels != 2)) return fail(AON_E_INVALID, "aon_grender_bwd: bad size / num_levels");
  if (!rays_d || !g_rgb_host || !workspace || !scratch) return fail(AON_E_INVALID, "aon_grender_bwd: null pointer");
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) || (reinterpret_cast<uintptr_t>(scratch) & 255))
    return fail(AON_E_INVALID, "aon_grender_bwd: workspace / scratch must be 256-byte aligned");
  const GTrainWs w = carve_gtrain(workspace, g, geo, n, num_levels);
  if (w.bytes > workspace_bytes) return fail(AON_E_WORKSPACE, "aon_grender_bwd: workspace smaller than aon_grender_train_workspace_bytes()");
  const GScratch sc = carve_gscratch(scratch, g, geo, n, num_levels);
  if (sc.bytes > scratch_bytes) return fail(AON_E_WORKSPACE, "aon_grender_bwd: scratch smaller than aon_grender_train_scratch_bytes()");
  const float* const* params[2] = {params_coarse_host, params_fine_host};
  float* const* grads[2] = {grads_coarse_host, grads_fine_host};
  for (int l = 0; l < num_levels; ++l) {
    const GTrainLevel& L = w.lvl[l];
    if (int rc = check_params(g, params[l], "aon_grender_bwd: null parameter pointer")) return rc;
    if (!grads[l] || !g_rgb_host[l]) return fail(AON_E_INVALID, "aon_grender_bwd: null level pointer");
    for (int i = 0; i < g.nparams(); ++i)
      if (!grads[l][i]) return fail(AON_E_INVALID, "aon_grender_bwd: null gradient pointer");
    int rc;
    {
      KTimer timer(kCompositeBwd, stream, n);
      rc = check(aon::launch_composite_bwd(L.raw, L.t, rays_d, g_rgb_host[l], g_acc_host ? g_acc_host[l] : nullptr, g_depth_host ? g_depth_host[l] : nullptr,
                                           n, L.S, white_bkgd, geo.act(false, l, 0), sc.d_raw, stream), who);
    }
    if (rc) return rc;
    if ((rc = gmlp_backward(g, params[l], grads[l], L.acts, sc, n, L.S, stream, who))) return rc;
  }
  return AON_OK;
}

}  // extern "C"
