"""tf_fast_rnnt on AMD MI355X (gfx950): the pruned RNN-T loss of Samsung/tf-fast-rnnt with its hot path
rebuilt as hand-written HIP behind a C ABI (include/ftr.h).  The public names, signatures and
``__version__`` are those of the reference package (tf_fast_rnnt/python/tf_fast_rnnt/__init__.py:24-36,
42,151); tensors are torch tensors on a HIP device (TensorFlow is not needed; a TF-ROCm op shim over the
same C ABI is described in INTEGRATION.md)."""
from ._lib import FtrError, lib as _load_native          # noqa: F401
from .mutual_information import cummin, mutual_information_recursion
from .rnnt_loss import do_rnnt_pruning
from .rnnt_loss import get_rnnt_logprobs
from .rnnt_loss import get_rnnt_logprobs_joint
from .rnnt_loss import get_rnnt_logprobs_pruned
from .rnnt_loss import get_rnnt_logprobs_smoothed
from .rnnt_loss import get_rnnt_prune_ranges
from .rnnt_loss import rnnt_loss
from .rnnt_loss import rnnt_loss_pruned
from .rnnt_loss import rnnt_loss_simple
from .rnnt_loss import rnnt_loss_smoothed
from .rnnt_loss import tune_normalizer_gemms, normalizer_gemm_choice, set_normalizer_gemm_choice                # MI355X addition: library-GEMM kernel selection, see its docstring

__version__ = '1.2'

# Fail loudly at import if the HIP extension has not been built (no CPU fallback exists).
_load_native()
