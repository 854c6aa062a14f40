// integration/tf_fast_rnnt_op_rocm.cc -- the TensorFlow-ROCm op library over the C ABI of include/ftr.h.
//
// Drop-in for tf_fast_rnnt/python/csrc/tf_fast_rnnt_op.cc of Samsung/tf-fast-rnnt: the SAME two ops with the same names,
// inputs, outputs and attributes (REGISTER_OP blocks as at tf_fast_rnnt_op.cc:27-38), GPU kernels registered the same way
// (:131-133, :164-165), so tf_fast_rnnt/__init__.py (tf.load_op_library + @ops.RegisterGradient("FastRNNTLoss"),
// __init__.py:38-40,154-162) and every existing training loop load the resulting _tf_fast_rnnt.so unchanged.  What
// changes is the body of the kernels: instead of MutualInformationCuda / MutualInformationBackwardCuda / CumminCuda
// (mutual_information.h:134-168) they call libftr_hip.so.
//
// Two further ops expose the entry points that replace Python-level TF code of rnnt_loss.py, for callers that want them
// native as well (one-line changes in rnnt_loss.py: get_rnnt_prune_ranges -> _tf_fast_rnnt.ftr_prune_ranges, do_rnnt_pruning
// -> _tf_fast_rnnt.ftr_do_pruning); the remaining symbols of INTEGRATION.md section 4 follow the same pattern.
//
// NOT BUILT OR TESTED IN THIS REPOSITORY: the build image has no TensorFlow.  The same call sequences are compiled and
// tested from a plain C++ host, tests/capi_host/host_main.cpp (tests/test_gpu_capi_host.py).  Build on a TF-ROCm box with
// integration/Makefile.
#include <cstdint>

#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"
#define EIGEN_USE_GPU
#include "tensorflow/core/util/gpu_kernel_helper.h"

#include "ftr.h"

namespace tf = tensorflow;

// ----------------------------------------------------------------------------------------------- op definitions
// As in the reference (tf_fast_rnnt_op.cc:27-38), plus shape functions (the reference registers none).
REGISTER_OP("FastRNNTLoss")
    .Input("px: float32")
    .Input("py: float32")
    .Input("boundary: int32")
    .Input("calc_gradients: bool")
    .Output("ans: float32")
    .Output("px_grad: float32")
    .Output("py_grad: float32")
    .SetShapeFn([](tf::shape_inference::InferenceContext* c) {
      c->set_output(0, c->Vector(c->Dim(c->input(0), 0)));
      c->set_output(1, c->input(0));      // the shape of px (the reference always allocates [B,S,T+1], :84)
      c->set_output(2, c->input(1));
      return tf::OkStatus();
    });

REGISTER_OP("Cummin").Input("in: int32").Output("out: int32").SetShapeFn(tf::shape_inference::UnchangedShape);

REGISTER_OP("FtrPruneRanges")           // get_rnnt_prune_ranges, rnnt_loss.py:647-761
    .Input("px_grad: float32")
    .Input("py_grad: float32")
    .Input("boundary: int32")
    .Attr("s_range: int")
    .Output("ranges: int32");

REGISTER_OP("FtrDoPruning")             // do_rnnt_pruning, rnnt_loss.py:763-812
    .Input("am: float32")
    .Input("lm: float32")
    .Input("ranges: int32")
    .Output("am_pruned: float32")
    .Output("lm_pruned: float32");

namespace {

void* StreamOf(tf::OpKernelContext* ctx) {          // hipStream_t on TF-ROCm (cudaStream_t in the reference, :58,145)
  return ctx->eigen_device<Eigen::GpuDevice>().stream();
}
#define FTR_TF_CALL(ctx, what, expr)                                                                          \
  do {                                                                                                        \
    const int rc_ = (expr);                                                                                   \
    OP_REQUIRES(ctx, rc_ == FTR_OK, tf::errors::Internal("rnnt_loss error in ", what, ": ", ftr_last_error())); \
  } while (0)

// ----------------------------------------------------------------------------------------------- FastRNNTLoss
// Replaces FastRNNTOpBase::Compute (tf_fast_rnnt_op.cc:48-117).
class FastRNNTOpROCm : public tf::OpKernel {
 public:
  using tf::OpKernel::OpKernel;
  void Compute(tf::OpKernelContext* ctx) override {
    const tf::Tensor& px = ctx->input(0);
    const tf::Tensor& py = ctx->input(1);
    const tf::Tensor& boundary = ctx->input(2);
    const bool calc_gradients = ctx->input(3).scalar<bool>()();           // HostMemory("calc_gradients"), :131-132
    OP_REQUIRES(ctx, px.dims() == 3 && py.dims() == 3, tf::errors::InvalidArgument("px and py must be 3-dimensional"));
    const int B = px.dim_size(0), S = px.dim_size(1), T = py.dim_size(2);
    const int modified = (px.dim_size(2) == T);
    void* stream = StreamOf(ctx);
    // forward -> backward workspace.  NOT the reference's {B, S+1, T+1} temp (:65-67): two lattices of split ratios, the
    // cut vectors and the hand-off region -- ask the library.  The _ws entry points refuse a buffer that is too small.
    const tf::int64 p_floats = static_cast<tf::int64>(ftr_mutual_information_workspace_floats(B, S, T));
    tf::Tensor p;                                                          // TF temps are 64-byte aligned (16 needed)
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(tf::DT_FLOAT, tf::TensorShape({p_floats}), &p));
    tf::Tensor *ans = nullptr, *px_grad = nullptr, *py_grad = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("ans", tf::TensorShape({B}), &ans));
    OP_REQUIRES_OK(ctx, ctx->allocate_output("px_grad", px.shape(), &px_grad));
    OP_REQUIRES_OK(ctx, ctx->allocate_output("py_grad", py.shape(), &py_grad));
    FTR_TF_CALL(ctx, "compute_rnnt_loss",
                ftr_mutual_information_fwd_ws_f32(px.flat<float>().data(), py.flat<float>().data(),
                                                  boundary.flat<int32_t>().data(), p.flat<float>().data(),
                                                  static_cast<size_t>(p_floats), /*flags=*/0,   // 0: the library zeroes its hand-off region
                                                  ans->flat<float>().data(), B, S, T, modified, stream));
    if (calc_gradients) {
      // ans_grad = NULL: the seed of ones of tf_fast_rnnt_op.cc:100-107 without the temp, the upload and the write-back;
      // px_grad / py_grad are written completely, zeros outside the boundary rectangles (no memsets, :93-96)
      FTR_TF_CALL(ctx, "compute_rnnt_loss",
                  ftr_mutual_information_bwd_ws_f32(px.flat<float>().data(), py.flat<float>().data(),
                                                    boundary.flat<int32_t>().data(), p.flat<float>().data(),
                                                    static_cast<size_t>(p_floats), /*flags=*/0, /*p_grad=*/nullptr,
                                                    px_grad->flat<float>().data(), py_grad->flat<float>().data(),
                                                    /*ans_grad=*/nullptr, /*overwrite_ans_grad=*/0, B, S, T, modified, stream));
    } else {
      // the reference leaves px_grad / py_grad uninitialised here and its registered gradient reads them (:83-98,
      // __init__.py:158-159): zeros are the defined answer
      tf::functor::SetZeroFunctor<Eigen::GpuDevice, float> zero;
      zero(ctx->eigen_device<Eigen::GpuDevice>(), px_grad->flat<float>());
      zero(ctx->eigen_device<Eigen::GpuDevice>(), py_grad->flat<float>());
    }
    // no cudaStreamSynchronize (:113): the temp is stream-ordered by TF's allocator, the kernels by `stream`
  }
};
REGISTER_KERNEL_BUILDER(Name("FastRNNTLoss").Device(tf::DEVICE_GPU).HostMemory("calc_gradients"), FastRNNTOpROCm);

// ----------------------------------------------------------------------------------------------- Cummin
// Replaces CumminOpGPU::Compute (tf_fast_rnnt_op.cc:135-165).
class CumminOpROCm : public tf::OpKernel {
 public:
  using tf::OpKernel::OpKernel;
  void Compute(tf::OpKernelContext* ctx) override {
    const tf::Tensor& in = ctx->input(0);
    OP_REQUIRES(ctx, in.dims() == 2, tf::errors::InvalidArgument("cummin expects a 2-D tensor"));
    tf::Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("out", in.shape(), &out));
    FTR_TF_CALL(ctx, "cummin", ftr_cummin_i32(in.flat<int32_t>().data(), out->flat<int32_t>().data(),
                                              static_cast<int>(in.dim_size(0)), static_cast<int>(in.dim_size(1)), StreamOf(ctx)));
  }
};
REGISTER_KERNEL_BUILDER(Name("Cummin").Device(tf::DEVICE_GPU), CumminOpROCm);

// ----------------------------------------------------------------------------------------------- FtrPruneRanges
// get_rnnt_prune_ranges + _adjust_pruning_lower_bound + _monotonic_lower_bound (rnnt_loss.py:553-761) in one call.
class PruneRangesOpROCm : public tf::OpKernel {
 public:
  explicit PruneRangesOpROCm(tf::OpKernelConstruction* c) : tf::OpKernel(c) { OP_REQUIRES_OK(c, c->GetAttr("s_range", &s_range_)); }
  void Compute(tf::OpKernelContext* ctx) override {
    const tf::Tensor& gx = ctx->input(0);
    const tf::Tensor& gy = ctx->input(1);
    const tf::Tensor& boundary = ctx->input(2);
    const int B = gx.dim_size(0), S = gx.dim_size(1), T1 = gx.dim_size(2), T = gy.dim_size(2);
    const int r = s_range_ > S ? S + 1 : s_range_;                       // rnnt_loss.py:710-711
    tf::Tensor* ranges = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("ranges", tf::TensorShape({B, T, r}), &ranges));
    tf::Tensor scratch;
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(tf::DT_INT32, tf::TensorShape({B, T}), &scratch));
    int r_eff = 0;
    FTR_TF_CALL(ctx, "prune_ranges",
                ftr_prune_ranges_i32(gx.flat<float>().data(), gy.flat<float>().data(), boundary.flat<int32_t>().data(),
                                     ranges->flat<int32_t>().data(), scratch.flat<int32_t>().data(), B, S, T, T1, s_range_,
                                     &r_eff, StreamOf(ctx)));
    OP_REQUIRES(ctx, r_eff == r, tf::errors::Internal("prune_ranges: effective s_range mismatch"));
  }
 private:
  int s_range_ = 0;
};
REGISTER_KERNEL_BUILDER(Name("FtrPruneRanges").Device(tf::DEVICE_GPU), PruneRangesOpROCm);

// ----------------------------------------------------------------------------------------------- FtrDoPruning
// do_rnnt_pruning (rnnt_loss.py:763-812): both dense outputs, as tf.broadcast_to + tf.gather produce them.
class DoPruningOpROCm : public tf::OpKernel {
 public:
  using tf::OpKernel::OpKernel;
  void Compute(tf::OpKernelContext* ctx) override {
    const tf::Tensor& am = ctx->input(0);
    const tf::Tensor& lm = ctx->input(1);
    const tf::Tensor& ranges = ctx->input(2);
    const int B = am.dim_size(0), T = am.dim_size(1), C = am.dim_size(2), S1 = lm.dim_size(1), r = ranges.dim_size(2);
    tf::Tensor *am_p = nullptr, *lm_p = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("am_pruned", tf::TensorShape({B, T, r, C}), &am_p));
    OP_REQUIRES_OK(ctx, ctx->allocate_output("lm_pruned", tf::TensorShape({B, T, r, C}), &lm_p));
    FTR_TF_CALL(ctx, "do_pruning",
                ftr_do_pruning_f32(am.flat<float>().data(), lm.flat<float>().data(), ranges.flat<int32_t>().data(),
                                   am_p->flat<float>().data(), lm_p->flat<float>().data(), B, T, S1, C, r, StreamOf(ctx)));
  }
};
REGISTER_KERNEL_BUILDER(Name("FtrDoPruning").Device(tf::DEVICE_GPU), DoPruningOpROCm);

}  // namespace
