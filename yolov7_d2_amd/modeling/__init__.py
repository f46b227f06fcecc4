from .blocks import BaseConv, Bottleneck, CSPLayer, Focus, SPPBottleneck
from .postprocess import (batched_nms, postprocess, generalized_batched_nms, batched_softnms, batched_clusternms,
                          matrix_nms, mask_nms)
from .yolox import YOLOX
from .yolox_net import CSPDarknet, YOLOPAFPN, YOLOXHead, build_cspdarknetx_backbone
from .detr_matcher import HungarianMatcher
from .detr_criterion import SetCriterion
from .attention import mha_core
from .iou_loss import IOUlossV6, IOUloss, pairwise_bbox_iou, bboxes_iou
from .transformer import (MultiheadAttention, TransformerEncoderLayer, TransformerDecoderLayer, TransformerEncoder,
                          TransformerDecoder, Transformer)
from .position_encoding import PositionEmbeddingSine
from .detr import DETR, MLP
from .resnet import ResNet, build_resnet_backbone, FrozenBatchNorm2d
from .box_ops import box_cxcywh_to_xyxy, box_xyxy_to_cxcywh, box_iou, generalized_box_iou
from .detr_meta import (Detr, Joiner, MaskedBackbone, MaskedBackboneTraceFriendly, NestedTensor, PostProcess,
                        nested_tensor_from_tensor_list)
from .sparseinst import (SparseInst, InstanceContextEncoder, PyramidPoolingModule, MyAdaptiveAvgPool2d, InstanceBranch,
                         GroupInstanceBranch, MaskBranch, BaseIAMDecoder, GroupIAMDecoder, SparseInstCriterion,
                         SparseInstMatcher, build_sparse_inst_encoder, build_sparse_inst_decoder,
                         build_sparse_inst_criterion, rescoring_mask)
from .yolov6_loss import ComputeLoss
from .bifpn import (BiFPN, BiFpnLayer, FpnCombine, ResampleFeatureMap, SeparableConv2d, ConvBnAct2d, Swish, GroupNorm,
                    get_fpn_config, build_resnet_bifpn_backbone)
