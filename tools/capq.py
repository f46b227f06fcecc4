import torch, ctypes as C
from yolov7_d2_amd import _lib as L
torch.zeros(1, device="cuda")
print("fused capacity", L.lib().mi_bn_fused_set_capacity(0))
p = torch.cuda.get_device_properties(0)
print(p.multi_processor_count, getattr(p, "shared_memory_per_multiprocessor", None), getattr(p, "shared_memory_per_block", None), getattr(p,"shared_memory_per_block_optin",None))
