"""Loss registry (reference: losses/access_losses.py:16-93): `train.py:38` and `train_pc_ae.py` obtain their criterion
through `get_loss_fn(name, class_weights, term_weights)`.

On the hot path: 'nnunet' (CE + generalised Dice, the default of train.py), 'chamfer' and the Chamfer term of 'mesh'
(losses/mesh_loss.py: surface samples in, regularisers only with pytorch3d) run on the HIP kernels; 'ce' is
torch's own criterion exactly as in the reference.  The remaining names ('recall', 'ssm', 'dpsr') belong to
pipelines outside SURVEY section 8 (mesh / shape-model / DPSR losses on pytorch3d): they are handed to the reference's own
classes when those are importable next to this package (a reference checkout with its dependencies), and raise
NotImplementedError otherwise -- never a silent substitute."""
import importlib
from enum import Enum
from typing import List

import torch
from torch import nn

from .chamfer_loss import ChamferLoss
from .mesh_loss import RegularizedMeshLoss
from .nnu_loss import NNULoss


class Losses(Enum):
    NNUNET = "nnunet"
    CE = "ce"
    RECALL = "recall"
    SSM = "ssm"
    CHAMFER = "chamfer"
    MESH = "mesh"
    DPSR = "dpsr"

    @classmethod
    def list(cls):
        return [c.value for c in cls]


_OUT_OF_SCOPE = {  # name -> (reference module, class, keyword names of term_weights, takes class_weights first)
    Losses.RECALL.value: ("losses.recall_loss", "BatchRecallLoss", None, False),
    Losses.SSM.value: ("losses.dgssm_loss", "DGSSMLoss", ("w_point", "w_coefficients", "w_affine"), False),
    Losses.DPSR.value: ("losses.dpsr_loss", "DPSRLoss", ("w_seg", "w_mesh", "epoch_start_mesh_loss"), True),
}


def _reference_class(name):
    module, cls, _, _ = _OUT_OF_SCOPE[name]
    try:
        mod = importlib.import_module(module)
        if getattr(mod, "__package__", "").startswith(__package__.split(".")[0]):
            raise ImportError("aliased to this package")
        return getattr(mod, cls)
    except Exception as e:  # missing checkout or missing pytorch3d
        raise NotImplementedError(
            f'loss "{name}" ({module}.{cls}) is outside the MI355X hot path (SURVEY section 8) and the reference '
            f'implementation is not importable here: {type(e).__name__}: {e}') from e


def get_loss_fn(loss: Losses, class_weights: torch.Tensor = None, term_weights: List[float] = None):
    if isinstance(loss, Losses):
        loss = loss.value
    if loss == Losses.NNUNET.value:
        return NNULoss(class_weights)
    if loss == Losses.CE.value:
        return nn.CrossEntropyLoss(class_weights)
    if loss == Losses.CHAMFER.value:
        return ChamferLoss()
    if loss == Losses.MESH.value:   # access_losses.py:67-77 of the reference
        if term_weights is not None:
            assert len(term_weights) == 4
            return RegularizedMeshLoss(w_chamfer=term_weights[0], w_edge_length=term_weights[1],
                                       w_normal_consistency=term_weights[2], w_laplacian=term_weights[3])
        return RegularizedMeshLoss()
    if loss in _OUT_OF_SCOPE:
        _, _, names, takes_cw = _OUT_OF_SCOPE[loss]
        cls = _reference_class(loss)
        args = (class_weights,) if takes_cw else ()
        if term_weights is not None and names is not None:
            assert len(term_weights) == len(names)
            return cls(*args, **dict(zip(names, term_weights)))
        return cls(*args)
    raise ValueError(f'No loss function named "{loss}". Please choose one from {Losses.list()} instead.')
