#!/bin/bash
# what kind of box is this?  (the pool has "fast" and "slow" MI355X boxes: whole-step times differ by ~15 %)
rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showcomputepartition --showmemorypartition 2>/dev/null | grep -vE "^=|^$|amdgpu.ids" | head -30
python - <<'P' 2>/dev/null | grep -v amdgpu.ids
import torch, time
p = torch.cuda.get_device_properties(0)
print("CUs", p.multi_processor_count, "clock", getattr(p, "clock_rate", None), "memclk", getattr(p, "memory_clock_rate", None), "L2", p.L2_cache_size)
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
for n in (1 << 30, 1 << 26):
    for _ in range(3): y[:n].copy_(x[:n])
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): y[:n].copy_(x[:n])
    b.record(); torch.cuda.synchronize()
    print(f"copy {n >> 20} MiB: {2 * n * 20 / (a.elapsed_time(b) * 1e-3) / 1e12:.2f} TB/s (read + write)")
m = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
for _ in range(3): m @ m
torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): m @ m
b.record(); torch.cuda.synchronize()
print(f"bf16 gemm 8192^3: {2 * 8192 ** 3 * 10 / (a.elapsed_time(b) * 1e-3) / 1e12:.0f} TFLOP/s")
P
[ -x tools/micro/launch_floor.bin ] && tools/micro/launch_floor.bin 2>/dev/null | head -8
