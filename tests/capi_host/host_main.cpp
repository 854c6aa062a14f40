// tests/capi_host/host_main.cpp -- a non-Python host of the C ABI (include/ftr.h): plain C++ + the HIP runtime, no torch.
// Stand-in for the reference-side TF op shim that cannot be built in this image (tf_fast_rnnt/python/csrc/
// tf_fast_rnnt_op.cc:48-165 does exactly this: allocate temps/outputs, call forward, call backward with ones, return):
// it runs ftr_mutual_information_{fwd,bwd}_ws_f32, ftr_cummin_i32 and ftr_prune_ranges_i32 on raw files written by
// tests/test_gpu_capi_host.py from a golden fixture and writes the outputs back for the test to compare.
//
//   capi_host.bin <dir> B S T modified s_range cummin_rows cummin_cols
//   in : <dir>/px.bin py.bin boundary.bin gx.bin gy.bin cummin_in.bin      out: ans.bin px_grad.bin py_grad.bin
//                                                                               ans_grad.bin cummin_out.bin ranges.bin
//
// Second mode, the entry points that replace the reference's Python-level functions (what a TF binding would register as
// additional ops), also from this compiled host:
//   capi_host.bin pipeline <dir> B S T C r blank delay_penalty
//   in : <dir>/am.bin lm.bin symbols.bin boundary.bin ranges.bin logits.bin
//   out: builder_px.bin builder_py.bin         ftr_rowmax_exp_f32 x2 -> ftr_simple_logprobs_fused_fwd_f32  (when C % 4 == 0)
//        simple_d_am.bin simple_d_lm.bin       ... -> recursion fwd + bwd -> ftr_simple_logprobs_bwd_w_scaled_f32 ->
//                                              ftr_normalizer_gemm_f32 (1, 2) -> ..._bwd_am_scaled_f32, ..._bwd_lm_f32
//        am_pruned.bin lm_pruned.bin           ftr_do_pruning_f32 with both output pointers (what tf.broadcast_to + tf.gather give)
//        lm_pruned_only.bin                    ftr_do_pruning_f32 with am_pruned = NULL (the gather alone)
//        pruned_ans.bin logits_grad.bin        ftr_pruned_band_fwd_f32 -> ftr_mutual_information_band_ws_f32 ->
//                                              ftr_pruned_band_bwd_scaled_f32 (reduction "mean": scale_mul = -1/B)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "../../include/ftr.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(3); } } while (0)
#define FTR_CALL(x) do { int rc_ = (x); if (rc_ != FTR_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, ftr_last_error()); exit(4); } } while (0)

template <typename T>
static std::vector<T> read_file(const std::string& path, size_t n) {
  std::vector<T> v(n);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f || fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "cannot read %zu items from %s\n", n, path.c_str()); exit(2); }
  fclose(f);
  return v;
}
template <typename T>
static void write_file(const std::string& path, const T* dev, size_t n) {
  std::vector<T> v(n);
  HIP_OK(hipMemcpy(v.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
  fclose(f);
}
template <typename T>
static T* to_device(const std::vector<T>& v) {
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, (v.size() ? v.size() : 1) * sizeof(T)));
  if (!v.empty()) HIP_OK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
template <typename T>
static T* device_alloc(size_t n) { T* d = nullptr; HIP_OK(hipMalloc(&d, (n ? n : 1) * sizeof(T))); return d; }

static int pipeline_main(int argc, char** argv) {
  if (argc != 10) { fprintf(stderr, "usage: %s pipeline dir B S T C r blank delay_penalty\n", argv[0]); return 1; }
  const std::string dir = argv[2];
  const int B = atoi(argv[3]), S = atoi(argv[4]), T = atoi(argv[5]), C = atoi(argv[6]), r = atoi(argv[7]), blank = atoi(argv[8]);
  const double penalty = atof(argv[9]);
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  const size_t nam = (size_t)B * T * C, nlm = (size_t)B * (S + 1) * C, nband = (size_t)B * T * r, nlog = nband * C;
  float* am = to_device(read_file<float>(dir + "/am.bin", nam));
  float* lm = to_device(read_file<float>(dir + "/lm.bin", nlm));
  int32_t* sym = to_device(read_file<int32_t>(dir + "/symbols.bin", (size_t)B * S));
  int32_t* bd = to_device(read_file<int32_t>(dir + "/boundary.bin", (size_t)B * 4));
  int32_t* ranges = to_device(read_file<int32_t>(dir + "/ranges.bin", nband));
  float* logits = to_device(read_file<float>(dir + "/logits.bin", nlog));

  // get_rnnt_logprobs (rnnt_loss.py:63-223): row maxima + exponentials, then the fused normaliser contraction + px / py writer
  int builder = 0;
  if (ftr_simple_logprobs_fused_supported(C)) {
    float* am_probs = device_alloc<float>(nam); float* lm_probs = device_alloc<float>(nlm);
    float* am_max = device_alloc<float>((size_t)B * T); float* lm_max = device_alloc<float>((size_t)B * (S + 1));
    const size_t npx = (size_t)B * S * (T + 1), npy = (size_t)B * (S + 1) * T;
    float* px = device_alloc<float>(npx); float* py = device_alloc<float>(npy); float* prod = device_alloc<float>(npy);
    FTR_CALL(ftr_rowmax_exp_f32(am, am_probs, am_max, (long long)B * T, C, st));
    FTR_CALL(ftr_rowmax_exp_f32(lm, lm_probs, lm_max, (long long)B * (S + 1), C, st));
    FTR_CALL(ftr_simple_logprobs_fused_fwd_f32(am, lm, sym, am_probs, lm_probs, am_max, lm_max, bd, blank, 0.0, px, py, prod, B, T, S, C, 0, st));
    HIP_OK(hipStreamSynchronize(st));
    write_file(dir + "/builder_px.bin", px, npx);
    write_file(dir + "/builder_py.bin", py, npy);
    builder = 1;
    // rnnt_loss_simple(reduction="sum") backward from here, as autodiff replays rnnt_loss.py:175-221: occupancies from the
    // recursion, W, the two matmuls (ftr_normalizer_gemm_f32 kinds 1 and 2: rocBLAS behind the C ABI; called twice so that
    // the second call measures the library's candidates), the two epilogues.  d loss / d ans = -1: scale = NULL, scale_mul = -1.
    const size_t nws1 = ftr_mutual_information_workspace_floats(B, S, T);
    float* ws1 = device_alloc<float>(nws1);
    float* ans1 = device_alloc<float>(B); float* gx = device_alloc<float>(npx); float* gy = device_alloc<float>(npy);
    FTR_CALL(ftr_mutual_information_fwd_ws_f32(px, py, bd, ws1, nws1, 0, ans1, B, S, T, 0, st));
    FTR_CALL(ftr_mutual_information_bwd_ws_f32(px, py, bd, ws1, nws1, 0, nullptr, gx, gy, nullptr, 0, B, S, T, 0, st));
    float* W = device_alloc<float>(npy); float* rsx = device_alloc<float>((size_t)B * (S + 1)); float* rsy = device_alloc<float>((size_t)B * (S + 1));
    float* dlmp = device_alloc<float>(nlm); float* damp = device_alloc<float>(nam);
    float* d_am = device_alloc<float>(nam); float* d_lm = device_alloc<float>(nlm);
    FTR_CALL(ftr_simple_logprobs_bwd_w_scaled_f32(gx, gy, nullptr, 0, -1.0f, prod, bd, W, rsx, rsy, B, T, S, 0, st));
    for (int rep = 0; rep < 2; ++rep) {
      FTR_CALL(ftr_normalizer_gemm_f32(1, W, am_probs, dlmp, B, T, S + 1, C, st));
      FTR_CALL(ftr_normalizer_gemm_f32(2, W, lm_probs, damp, B, T, S + 1, C, st));
    }
    int sol = -1, cand = -2;
    if (!ftr_normalizer_gemm_choice(1, B, T, S + 1, C, &sol, nullptr, nullptr, &cand) || cand < 0) { fprintf(stderr, "GEMM kind 1 was not measured at its second call (candidates %d)\n", cand); return 10; }
    FTR_CALL(ftr_simple_logprobs_bwd_am_scaled_f32(gx, gy, nullptr, 0, -1.0f, damp, am_probs, sym, bd, blank, d_am, B, T, S, C, 0, st));
    FTR_CALL(ftr_simple_logprobs_bwd_lm_f32(dlmp, lm_probs, sym, rsx, rsy, blank, d_lm, B, S, C, st));
    HIP_OK(hipStreamSynchronize(st));
    write_file(dir + "/simple_d_am.bin", d_am, nam);
    write_file(dir + "/simple_d_lm.bin", d_lm, nlm);
  }

  // do_rnnt_pruning (rnnt_loss.py:763-812): both outputs, then the gather alone
  float* am_p = device_alloc<float>(nlog); float* lm_p = device_alloc<float>(nlog); float* lm_p2 = device_alloc<float>(nlog);
  FTR_CALL(ftr_do_pruning_f32(am, lm, ranges, am_p, lm_p, B, T, S + 1, C, r, st));
  FTR_CALL(ftr_do_pruning_f32(am, lm, ranges, nullptr, lm_p2, B, T, S + 1, C, r, st));

  // rnnt_loss_pruned on the band (rnnt_loss.py:1022-1130): builder, recursion forward + backward, gradient w.r.t. logits
  if (ftr_mutual_information_band_supported(T, S, r) == 0) { fprintf(stderr, "band kernels do not cover T=%d S=%d r=%d\n", T, S, r); return 9; }
  float* lse = device_alloc<float>(nband); float* pxb = device_alloc<float>(nband); float* pyb = device_alloc<float>(nband);
  float* gxb = device_alloc<float>(nband); float* gyb = device_alloc<float>(nband); float* ans = device_alloc<float>(B);
  float* glog = device_alloc<float>(nlog);
  const size_t nws = ftr_mutual_information_band_workspace_floats(B, T, S, r);
  float* ws = device_alloc<float>(nws);
  FTR_CALL(ftr_pruned_band_fwd_f32(logits, sym, ranges, bd, blank, penalty, lse, pxb, pyb, B, T, S, C, r, 0, st));
  FTR_CALL(ftr_mutual_information_band_ws_f32(pxb, pyb, ranges, bd, nws ? ws : nullptr, nws, ans, gxb, gyb, B, T, S, r, 0, st));
  FTR_CALL(ftr_pruned_band_bwd_scaled_f32(logits, sym, ranges, bd, blank, lse, gxb, gyb, nullptr, 0, -1.0f / (float)B, glog, B, T, S, C, r, 0, st));
  HIP_OK(hipStreamSynchronize(st));
  write_file(dir + "/am_pruned.bin", am_p, nlog);
  write_file(dir + "/lm_pruned.bin", lm_p, nlog);
  write_file(dir + "/lm_pruned_only.bin", lm_p2, nlog);
  write_file(dir + "/pruned_ans.bin", ans, (size_t)B);
  write_file(dir + "/logits_grad.bin", glog, nlog);
  printf("capi_host pipeline OK: B=%d S=%d T=%d C=%d r=%d builder=%d band workspace %zu floats\n", B, S, T, C, r, builder, nws);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "pipeline") return pipeline_main(argc, argv);
  if (argc != 9) { fprintf(stderr, "usage: %s dir B S T modified s_range cummin_rows cummin_cols\n", argv[0]); return 1; }
  const std::string dir = argv[1];
  const int B = atoi(argv[2]), S = atoi(argv[3]), T = atoi(argv[4]), modified = atoi(argv[5]), s_range = atoi(argv[6]);
  const int crows = atoi(argv[7]), ccols = atoi(argv[8]);
  const int T1 = modified ? T : T + 1;
  const size_t npx = (size_t)B * S * T1, npy = (size_t)B * (S + 1) * T;
  if (ftr_abi_version() < 110) { fprintf(stderr, "library ABI %d too old\n", ftr_abi_version()); return 5; }

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  float* px = to_device(read_file<float>(dir + "/px.bin", npx));
  float* py = to_device(read_file<float>(dir + "/py.bin", npy));
  int32_t* bd = to_device(read_file<int32_t>(dir + "/boundary.bin", (size_t)B * 4));

  // FastRNNTOpBase::Compute (tf_fast_rnnt_op.cc:59-112): temp p, outputs, forward, ones, backward
  const size_t p_floats = ftr_mutual_information_workspace_floats(B, S, T);
  float* p = device_alloc<float>(p_floats);
  float* ans = device_alloc<float>(B);
  float* px_grad = device_alloc<float>(npx);
  float* py_grad = device_alloc<float>(npy);
  float* ans_grad = to_device(std::vector<float>(B, 1.0f));
  // an undersized workspace must be refused, not written past (what a {B,S+1,T+1} temp would be)
  if (ftr_mutual_information_fwd_ws_f32(px, py, bd, p, (size_t)B * (S + 1) * (T + 1), 0, ans, B, S, T, modified, st) != FTR_ERR_INVALID_ARG) {
    fprintf(stderr, "an undersized workspace was accepted\n"); return 6;
  }
  FTR_CALL(ftr_mutual_information_fwd_ws_f32(px, py, bd, p, p_floats, 0, ans, B, S, T, modified, st));
  FTR_CALL(ftr_mutual_information_bwd_ws_f32(px, py, bd, p, p_floats, 0, nullptr, px_grad, py_grad, ans_grad, 1, B, S, T, modified, st));

  // Cummin op (tf_fast_rnnt_op.cc:135-165)
  int32_t* cin = to_device(read_file<int32_t>(dir + "/cummin_in.bin", (size_t)crows * ccols));
  int32_t* cout_ = device_alloc<int32_t>((size_t)crows * ccols);
  FTR_CALL(ftr_cummin_i32(cin, cout_, crows, ccols, st));

  // get_rnnt_prune_ranges (rnnt_loss.py:647-761) on the fixture's occupancies
  float* gx = to_device(read_file<float>(dir + "/gx.bin", npx));
  float* gy = to_device(read_file<float>(dir + "/gy.bin", npy));
  int r_eff = 0;
  const int r_max = s_range > S ? S + 1 : s_range;
  int32_t* ranges = device_alloc<int32_t>((size_t)B * T * r_max);
  int32_t* scratch = device_alloc<int32_t>((size_t)B * T);
  FTR_CALL(ftr_prune_ranges_i32(gx, gy, bd, ranges, scratch, B, S, T, T1, s_range, &r_eff, st));
  if (r_eff != r_max) { fprintf(stderr, "r_eff %d != %d\n", r_eff, r_max); return 7; }

  HIP_OK(hipStreamSynchronize(st));
  int status = -1;
  FTR_CALL(ftr_mutual_information_status(p, p_floats, B, S, T, &status, nullptr, st));
  if (status != 0) { fprintf(stderr, "workspace status word %d\n", status); return 8; }
  write_file(dir + "/ans.bin", ans, (size_t)B);
  write_file(dir + "/px_grad.bin", px_grad, npx);
  write_file(dir + "/py_grad.bin", py_grad, npy);
  write_file(dir + "/ans_grad.bin", ans_grad, (size_t)B);
  write_file(dir + "/cummin_out.bin", cout_, (size_t)crows * ccols);
  write_file(dir + "/ranges.bin", ranges, (size_t)B * T * r_eff);
  printf("capi_host OK: B=%d S=%d T=%d modified=%d r=%d workspace %zu floats\n", B, S, T, modified, r_eff, p_floats);
  return 0;
}
