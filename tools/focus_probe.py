"""time mi_focus_pack_u8 alone (HIP events, 50 launches) for the kernel variants: MI_FOCUS_ROWS / MI_FOCUS_TH via subprocesses"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from yolov7_d2_amd import _lib as L
    N, H, W = 16, 640, 640
    img = torch.randint(0, 256, (N, 3, H, W), dtype=torch.uint8, device="cuda")
    out = torch.empty(N, H // 2, W // 2, 16, dtype=torch.bfloat16, device="cuda")
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    f = lambda: L.check(L.lib().mi_focus_pack_u8(img.data_ptr(), N, H, W, out.data_ptr(), 16, L.stream_ptr()), "focus")
    for _ in range(3):
        f()
    ts = []
    for _ in range(20):
        big.fill_(1)                         # cold caches, as in the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"cold median {sorted(ts)[10]:.1f} us, back-to-back {e0.elapsed_time(e1) * 1e3 / 50:.1f} us")
    sys.exit(0)
for env in ({"MI_FOCUS_ROWS": "0"}, {"MI_FOCUS_ROWS": "1"}, {"MI_FOCUS_ROWS": "2", "MI_FOCUS_TH": "1"}, {"MI_FOCUS_ROWS": "2", "MI_FOCUS_TH": "2"},
            {"MI_FOCUS_ROWS": "2", "MI_FOCUS_TH": "4"}, {"MI_FOCUS_ROWS": "2", "MI_FOCUS_TH": "8"}):
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
    print(env, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
