"""vireo_amd -- MI355X-native implementation of vireoSNP's variational-EM hot path.

Same names as the reference package surface (vireoSNP/__init__.py:7-15) for the path
that is in scope: ``Vireo``, ``BinomMixtureVB``, ``vireo_wrap`` and the helpers they use.
Host code is plain Python/NumPy; all per-iteration arithmetic runs in hand-written HIP
kernels for gfx950 behind a ctypes C ABI (include/vireo_hip.h).  There is no CPU
fallback and no PyTorch in the compute path.
"""
__version__ = "0.2.0"

from . import vcf_utils as vcf
from . import vireo_base as base
from . import vireo_model as model
from .counts import DeviceCounts, device_counts
from .vcf_utils import load_VCF, match_SNPs
from .io_utils import read_cellSNP, read_vartrix
from .vireo_base import (normalize, tensor_normalize, loglik_amplify, get_binom_coeff,
                         binom_coeff_sum, beta_entropy, match, optimal_match, donor_select)
from .vireo_model import Vireo
from .bmm_model import BinomMixtureVB
from .vireo_doublet import predict_doublet, add_doublet_GT, add_doublet_theta
from .vireo_wrap import vireo_wrap

__all__ = ["__version__", "Vireo", "BinomMixtureVB", "vireo_wrap", "predict_doublet",
           "DeviceCounts", "device_counts", "load_VCF", "match_SNPs", "read_cellSNP",
           "read_vartrix", "normalize", "tensor_normalize", "loglik_amplify", "get_binom_coeff",
           "binom_coeff_sum", "beta_entropy", "match", "optimal_match", "donor_select",
           "vcf", "base", "model"]
