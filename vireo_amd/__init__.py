"""vireo_amd -- MI355X-native implementation of vireoSNP's variational-EM hot path.

Same names as the reference package surface (vireoSNP/__init__.py:7-15) for the path
that is in scope: ``Vireo``, ``BinomMixtureVB``, ``vireo_wrap`` and the helpers they use.
Host code is plain Python/NumPy; all per-iteration arithmetic runs in hand-written HIP
kernels for gfx950 behind a ctypes C ABI (include/vireo_hip.h).  There is no CPU
fallback and no PyTorch in the compute path.
"""
__version__ = "0.1.0"

from .counts import DeviceCounts, device_counts
from .vireo_base import (normalize, tensor_normalize, loglik_amplify, binom_coeff_sum,
                         match, optimal_match, donor_select)
from .vireo_model import Vireo
from .bmm_model import BinomMixtureVB
from .vireo_doublet import predict_doublet, add_doublet_GT, add_doublet_theta
from .vireo_wrap import vireo_wrap

__all__ = ["Vireo", "BinomMixtureVB", "vireo_wrap", "predict_doublet", "DeviceCounts",
           "device_counts", "normalize", "tensor_normalize", "loglik_amplify",
           "binom_coeff_sum", "match", "optimal_match", "donor_select"]
