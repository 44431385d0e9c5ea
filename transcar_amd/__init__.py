"""MI355X-native (gfx950 / CDNA4) implementation of the TransCAR camera-radar
fusion decoder hot path, behind the reference's own plugin API.

    from transcar_amd import build_head, configs
    head = build_head(configs.head_cfg()).cuda().eval()
    outs = head(mlvl_feats, img_metas)          # HEAD:248-261 contract

Importing the package registers the classes under the reference's registry
names (Detr3DHead, Detr3DTransformer, Detr3DTransformerDecoder,
Detr3DCrossAtten, NMSFreeCoder, DetrTransformerDecoderLayer,
MultiheadAttention).  The arithmetic lives in transcar_amd/lib/
libtranscar_hip.so (C ABI: include/transcar_hip.h); there is no CPU fallback.
"""
from . import configs, synth                                    # noqa: F401
from . import bricks, detr3d_transformer, nms_free_coder        # noqa: F401
from . import detr3d_head, radar, bbox_util                     # noqa: F401
from ._lib import TransCARHipError, lib                         # noqa: F401
from .detr3d_head import Detr3DHead                             # noqa: F401
from .detr3d_transformer import (Detr3DCrossAtten, Detr3DTransformer,  # noqa
                                 Detr3DTransformerDecoder)
from .nms_free_coder import NMSFreeCoder                        # noqa: F401
from .registry import (build_bbox_coder, build_head,            # noqa: F401
                       build_transformer)

__version__ = '0.1.0'
