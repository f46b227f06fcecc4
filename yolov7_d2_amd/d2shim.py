"""The detectron2 surface the YOLOX hot path binds to.

If a real detectron2 is importable its registries / structures are used (the classes below are then
registered into detectron2's own META_ARCH_REGISTRY / BACKBONE_REGISTRY, which is what makes the
reference's YAMLs drop-in).  detectron2 is not vendored in the reference and not installed in this
image, so a minimal restatement of exactly the pieces yolov7/modeling/meta_arch/yolox.py uses
(yolox.py:5-11,101,238-249) is provided: Registry, Backbone, ShapeSpec, ImageList.from_tensors,
Boxes, Instances, detector_postprocess, build_backbone, build_model.  (d2 upstream semantics.)
"""
from dataclasses import dataclass
from typing import Any, List, Optional, Tuple

import torch
from torch import nn

try:  # pragma: no cover - not available in this image
    from detectron2.layers import ShapeSpec
    from detectron2.modeling import BACKBONE_REGISTRY, META_ARCH_REGISTRY, Backbone
    from detectron2.modeling.postprocessing import detector_postprocess
    from detectron2.structures import Boxes, ImageList, Instances

    HAVE_D2 = True
except Exception:
    HAVE_D2 = False

    class Registry:
        def __init__(self, name):
            self._name, self._map = name, {}

        def register(self, obj=None):
            if obj is None:
                def deco(o):
                    self._map[o.__name__] = o
                    return o
                return deco
            self._map[obj.__name__] = obj
            return obj

        def get(self, name):
            if name not in self._map:
                raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
            return self._map[name]

        def __contains__(self, name):
            return name in self._map

    META_ARCH_REGISTRY = Registry("META_ARCH")
    BACKBONE_REGISTRY = Registry("BACKBONE")

    @dataclass
    class ShapeSpec:
        channels: Optional[int] = None
        height: Optional[int] = None
        width: Optional[int] = None
        stride: Optional[int] = None

    class Backbone(nn.Module):
        def forward(self, x):
            raise NotImplementedError

        @property
        def size_divisibility(self) -> int:
            return 0

        def output_shape(self):
            return {}

    class Boxes:
        def __init__(self, tensor: torch.Tensor):
            if tensor.numel() == 0:
                tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
            self.tensor = tensor.to(torch.float32) if tensor.dtype != torch.float32 else tensor

        def to(self, device):
            return Boxes(self.tensor.to(device))

        def scale(self, scale_x, scale_y):
            self.tensor[:, 0::2] *= scale_x
            self.tensor[:, 1::2] *= scale_y

        def clip(self, box_size):
            h, w = box_size
            x1 = self.tensor[:, 0].clamp(min=0, max=w)
            y1 = self.tensor[:, 1].clamp(min=0, max=h)
            x2 = self.tensor[:, 2].clamp(min=0, max=w)
            y2 = self.tensor[:, 3].clamp(min=0, max=h)
            self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

        def nonempty(self, threshold: float = 0.0):
            b = self.tensor
            return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

        def __len__(self):
            return self.tensor.shape[0]

        def __getitem__(self, item):
            if isinstance(item, int):
                return Boxes(self.tensor[item].view(1, -1))
            return Boxes(self.tensor[item])

    class Instances:
        def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
            object.__setattr__(self, "_image_size", image_size)
            object.__setattr__(self, "_fields", {})
            for k, v in kwargs.items():
                self.set(k, v)

        @property
        def image_size(self):
            return self._image_size

        def __setattr__(self, name, val):
            if name.startswith("_"):
                object.__setattr__(self, name, val)
            else:
                self.set(name, val)

        def __getattr__(self, name):
            if name == "_fields" or name not in self._fields:
                raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
            return self._fields[name]

        def set(self, name, value):
            self._fields[name] = value

        def has(self, name):
            return name in self._fields

        def get_fields(self):
            return self._fields

        def to(self, device):
            r = Instances(self._image_size)
            for k, v in self._fields.items():
                r.set(k, v.to(device) if hasattr(v, "to") else v)
            return r

        def __getitem__(self, item):
            r = Instances(self._image_size)
            for k, v in self._fields.items():
                r.set(k, v[item])
            return r

        def __len__(self):
            for v in self._fields.values():
                return len(v)
            return 0

    class ImageList:
        def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
            self.tensor, self.image_sizes = tensor, image_sizes

        def __len__(self):
            return len(self.image_sizes)

        @property
        def device(self):
            return self.tensor.device

        @staticmethod
        def from_tensors(tensors, size_divisibility: int = 0, pad_value: float = 0.0):
            """batch shape = per-dim max, H/W rounded up to size_divisibility; images copied top-left."""
            image_sizes = [(int(t.shape[-2]), int(t.shape[-1])) for t in tensors]
            mh = max(s[0] for s in image_sizes)
            mw = max(s[1] for s in image_sizes)
            if size_divisibility > 1:
                mh = (mh + size_divisibility - 1) // size_divisibility * size_divisibility
                mw = (mw + size_divisibility - 1) // size_divisibility * size_divisibility
            out = tensors[0].new_full((len(tensors), tensors[0].shape[0], mh, mw), pad_value)
            for i, t in enumerate(tensors):
                out[i, :, : t.shape[-2], : t.shape[-1]].copy_(t)
            return ImageList(out.contiguous(), image_sizes)

    def detector_postprocess(results, output_height: int, output_width: int):
        sx = output_width / results.image_size[1]
        sy = output_height / results.image_size[0]
        r = Instances((output_height, output_width), **results.get_fields())
        boxes = r.pred_boxes
        boxes.scale(sx, sy)
        boxes.clip(r.image_size)
        return r[boxes.nonempty()]


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)
    assert isinstance(backbone, Backbone)
    return backbone


def build_model(cfg):
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
