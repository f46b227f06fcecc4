from .blocks import BaseConv, Bottleneck, CSPLayer, Focus, SPPBottleneck
from .postprocess import batched_nms, postprocess
from .yolox import YOLOX
from .yolox_net import CSPDarknet, YOLOPAFPN, YOLOXHead, build_cspdarknetx_backbone
from .detr_matcher import HungarianMatcher
from .detr_criterion import SetCriterion
from .attention import mha_core
from .iou_loss import IOUlossV6
from .transformer import (MultiheadAttention, TransformerEncoderLayer, TransformerDecoderLayer, TransformerEncoder,
                          TransformerDecoder, Transformer)
from .position_encoding import PositionEmbeddingSine
from .detr import DETR, MLP
