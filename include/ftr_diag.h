/* include/ftr_diag.h -- symbols of the TEST-ONLY library tf-fast-rnnt_amd/csrc/_build/libftr_hip_diag.so (make -C
 * tf-fast-rnnt_amd/csrc tests), on top of everything in ftr.h.  None of this is in the product library libftr_hip.so:
 * a process-global switch in a shared library is something a multi-threaded host (the TensorFlow executor) could flip
 * under another thread's launch, and the "plain" kernels are a bisecting aid, not a product path. */
#ifndef FTR_DIAG_H_
#define FTR_DIAG_H_
#include "ftr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Selects the mutual-information kernel family of ftr_mutual_information_{fwd,bwd}[_ws]_f32: 0 = "wavefront" (the
 * product kernels, mi_wave_bidir.hip), 1 = "plain" (one thread per lattice row, the reference's own arithmetic --
 * mutual_information_cuda.cu:149-239, 441-481 -- on the device, up to 1024 rows; needs a [B,S+1,T+1] workspace and the
 * p_grad scratch lattice, and a non-NULL ans_grad).  Also settable with FTR_MI_IMPL=wavefront|plain.  Returns the
 * previous value. */
int ftr_set_mi_impl(int impl);
int ftr_get_mi_impl(void);

/* Copies 16 counters out of the library (host pointer).  All zero unless the library was built with -DFTR_STAMPS, in
 * which case they are per-wave busy / barrier-wait ticks of the forward kernel's slots.  Synchronises the device. */
int ftr_debug_stamps(unsigned long long* out16);
/* Copies n <= 1024 words of the kernel timeline out of the library (host pointer) and re-arms it.  All zero unless the
 * library was built with -DFTR_TRACE=1 (forward) or =2 (flow): [0] earliest workgroup start, [1] latest workgroup end,
 * [2]/[3] start/end of one traced workgroup, [4] slots recorded, [16+k] its slot times, in 100 MHz ticks.  Synchronises
 * the device. */
int ftr_debug_trace(unsigned long long* out, int n);

#ifdef __cplusplus
}
#endif
#endif /* FTR_DIAG_H_ */
