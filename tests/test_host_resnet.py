"""host logic of the ResNet mirror that needs no GPU: the FrozenBatchNorm2d (scale, shift) pair is computed once and kept
until one of the four buffers is written or replaced (detectron2 layers/batch_norm.py FrozenBatchNorm2d; un-vendored)"""
import torch

from yolov7_d2_amd.modeling.resnet import FrozenBatchNorm2d, ResNet


def test_frozen_norm_affine_is_cached_until_a_buffer_changes():
    torch.manual_seed(0)
    m = FrozenBatchNorm2d(8)
    m.weight.uniform_(0.5, 1.5); m.bias.normal_(); m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0)
    s1, b1 = m.affine()
    s2, b2 = m.affine()
    assert s1 is s2 and b1 is b2
    ref_s = m.weight * (m.running_var + m.eps).rsqrt()
    torch.testing.assert_close(s1, ref_s)
    torch.testing.assert_close(b1, m.bias - m.running_mean * ref_s)
    old = s1.clone()
    m.running_var.mul_(2.0)                                   # an in-place write bumps the buffer's version: recomputed -
    s3, _ = m.affine()                                        # INTO the same tensors (captured graphs hold their addresses:
    assert s3 is s1 and not torch.equal(s3, old)              #  tests/test_frozen_constants.py)
    torch.testing.assert_close(s3, m.weight * (m.running_var + m.eps).rsqrt())
    keep = s3.clone()
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})     # copy_ into the buffers: recomputed, same values
    s4, _ = m.affine()
    assert s4 is s1 and torch.equal(s4, keep)
    m.double().float()                                        # buffers REPLACED by new tensors (module._apply)
    s5, _ = m.affine()
    assert s5 is s1 and torch.equal(s5, keep)


def test_resnet_freeze_at_marks_the_prefix_only():
    r = ResNet(50, ("res5",), freeze_at=2)
    frozen = {n for n, p in r.named_parameters() if not p.requires_grad}
    assert all(n.startswith(("stem.", "res2.")) for n in frozen) and any(n.startswith("res2.") for n in frozen)
    assert sum(p.requires_grad for p in r.parameters()) == 42      # 13 blocks x 3 convs + 3 shortcuts
