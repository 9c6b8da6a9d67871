"""fissure_segmentation_amd -- MI355X-native hot path of kaftanski/fissure-segmentation.

Host side mirrors the reference's module layout (models/, losses/, utils/, shapes/) so that
`train.py` / `train_pc_ae.py` / `model_trainer.py` find the same names, constructor arguments,
`forward` signatures and `state_dict` keys; the arithmetic runs in hand-written HIP kernels reached
through the C ABI of `libfsg_hip.so` (include/fsg_hip.h).  There is no CPU fallback: modules raise
if their input is not on a GPU or the library is missing.
"""
from . import _lib  # noqa: F401  (fails loudly when libfsg_hip.so is absent)
from . import functional  # noqa: F401
from . import augmentations  # noqa: F401

__version__ = "0.1.0"


def install_reference_aliases():
    """Register this package's modules under the reference's import names (`models.dgcnn`,
    `losses.chamfer_loss`, ...) so unmodified reference scripts pick them up.  See INTEGRATION.md."""
    import importlib
    import sys
    pairs = {
        "models.dgcnn": ".models.dgcnn", "models.dgcnn_opensrc": ".models.dgcnn_opensrc",
        "models.folding_net": ".models.folding_net", "models.point_net": ".models.point_net",
        "models.point_seg_net": ".models.point_seg_net", "models.modelio": ".models.modelio",
        "models.access_models": ".models.access_models",
        "models.pointtransformer.pointops": ".models.pointtransformer.pointops",
        "models.pointtransformer.seg_model": ".models.pointtransformer.seg_model",
        "losses.chamfer_loss": ".losses.chamfer_loss", "losses.nnu_loss": ".losses.nnu_loss",
        "losses.access_losses": ".losses.access_losses", "losses.mesh_loss": ".losses.mesh_loss",
    }
    for ref_name, ours in pairs.items():
        sys.modules[ref_name] = importlib.import_module(ours, __name__)
