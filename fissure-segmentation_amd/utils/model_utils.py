"""Weight initialisation used by every model of the path (reference: utils/model_utils.py:11-15)."""
from torch import nn


def init_weights(m):
    if isinstance(m, (nn.modules.conv._ConvNd, nn.Linear)):
        nn.init.xavier_normal_(m.weight)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
