"""Name -> class registry used by the entry scripts (reference: models/access_models.py:7-22)."""
from .dgcnn import DGCNNSeg
from .point_net import PointNetSeg
from .pointtransformer.seg_model import PointTransformerCompatibility

_REGISTRY = {'DGCNN': DGCNNSeg, 'PointNet': PointNetSeg, 'PointTransformer': PointTransformerCompatibility}


def get_point_seg_model_class(model_string):
    try:
        return _REGISTRY[model_string]
    except KeyError:
        raise NotImplementedError(model_string)


def get_point_seg_model_class_from_args(args):
    return get_point_seg_model_class(args.model) if 'model' in args else DGCNNSeg
