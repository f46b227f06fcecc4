"""The constants a frozen backbone derives from its buffers (FrozenBatchNorm2d's scale / shift pair, the packed image of a
frozen convolution) are launch arguments of every captured step.  They must keep their ADDRESS when they are recomputed:
GraphedTrainStep restores the module buffers with copy_ after its warm-up passes (a version bump, same values), and the
next forward - the next shape's warm-up, an evaluation between training steps - recomputes them.  Replacing the tensors
frees memory an earlier capture still reads (found on the device: a second captured shape read another tensor's bytes as the
BatchNorm shift after an eager forward in between)."""
import torch

from yolov7_d2_amd.modeling.resnet import FrozenBatchNorm2d, _same_address


def test_frozen_batchnorm_affine_is_address_stable():
    bn = FrozenBatchNorm2d(8)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, 8)); bn.bias.copy_(torch.linspace(-1, 1, 8))
        bn.running_mean.copy_(torch.linspace(-0.3, 0.3, 8)); bn.running_var.copy_(torch.linspace(0.5, 2.0, 8))
    s0, h0 = bn.affine()
    ref_s = bn.weight * (bn.running_var + bn.eps).rsqrt()
    assert torch.allclose(s0, ref_s) and torch.allclose(h0, bn.bias - bn.running_mean * ref_s)
    assert bn.affine()[0] is s0                                   # (unchanged buffers: the cached pair)
    ps, ph = s0.data_ptr(), h0.data_ptr()
    with torch.no_grad():
        for b in bn.buffers():                                    # what GraphedTrainStep._restore does: same values, new versions
            b.copy_(b.clone())
    s1, h1 = bn.affine()
    assert (s1.data_ptr(), h1.data_ptr()) == (ps, ph) and torch.allclose(s1, ref_s)
    with torch.no_grad():
        bn.running_var.mul_(4.0)                                  # a real change is recomputed INTO the same tensors
    s2, h2 = bn.affine()
    assert (s2.data_ptr(), h2.data_ptr()) == (ps, ph)
    assert torch.allclose(s2, bn.weight * (bn.running_var + bn.eps).rsqrt()) and s0 is s2
    bn.running_var = bn.running_var.double().float()              # a REPLACED buffer (other storage) as well
    assert bn.affine()[0].data_ptr() == ps


def test_repacked_constant_image_keeps_its_tensor():
    old = torch.arange(12, dtype=torch.bfloat16)
    new = torch.ones(12, dtype=torch.bfloat16)
    out = _same_address(("key", old), new)
    assert out is old and bool((old == 1).all())
    other = torch.ones(16, dtype=torch.bfloat16)                  # a different geometry cannot reuse it
    assert _same_address(("key", old), other) is other
    assert _same_address(None, new) is new
