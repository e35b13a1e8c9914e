"""voiceconversion.jl_amd -- MI355X-native hot path of r9y9/VoiceConversion.jl.

Host-side mirror of the reference's exported API (src/VoiceConversion.jl:12-38) over the C-ABI of
libvcmi.so (include/vcmi.h).  The directory name contains a dot, so the package is imported through the
root-level shim `voiceconversion_jl_amd` (see voiceconversion_jl_amd.py)."""
from ._lib import (DimensionMismatch, PosDefException, VCMIError, device_count, get_devices, is_pinned, pin,  # noqa: F401
                   set_device, set_devices, unpin)
from .common import (AbstractConverter, FrameByFrameConverter, TrajectoryConverter, dim, fvconvert,  # noqa: F401
                     ncomponents, size, vc)
from .gmm import GMM, predict, predict_proba  # noqa: F401
from .gmmmap import GMMMap  # noqa: F401
from .dtw import DTW, backward, fit_, fit_batch, set_template_, update_  # noqa: F401,E402
from .align import align, align_batch  # noqa: F401,E402
from .estep import (ESTEP_AUTO, ESTEP_HARD, ESTEP_SOFT, estep_diag, estep_diag_allreduce, estep_diag_dev, estep_get_path,  # noqa: F401,E402
                    estep_set_path, mstep_diag, stats_len, unpack_stats,
                    estep_full, estep_full_allreduce, estep_full_dev, fit_full, full_stats_len, mstep_full,
                    unpack_full_stats)
from .trajectory_gmmmap import TrajectoryGVGMMMap, TrajectoryGMMMap, constructW, push_delta  # noqa: F401,E402
from . import dist  # noqa: F401,E402
from .train import EMState, train_gmm  # noqa: F401,E402
from .gv import VarianceScaling, diffgmm, fvpostf, fvpostf_  # noqa: F401,E402
from .datasets import GVDataset, ParallelDataset, align_mcep, mc2e  # noqa: F401,E402
