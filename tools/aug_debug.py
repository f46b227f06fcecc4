import random, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import augment_oracle as A
from yolov7_d2_amd.data_pipeline import GpuMosaicMapper
from test_gpu_augment import _pool
sizes = [(480, 640), (375, 500), (640, 427), (333, 500), (720, 1280), (1080, 1920), (200, 150), (427, 640), (512, 512),
         (900, 700), (96, 2000), (1500, 110)]
pool, imgs, labs = _pool(10, sizes)
mapper = GpuMosaicMapper(device="cuda")
rng_np, rng_py = np.random.RandomState(20), random.Random(30)
B = 6
groups = [tuple(int(i) for i in rng_np.randint(0, len(sizes), 4)) for _ in range(B)]
params = [mapper.draw(rng_np, rng_py) for _ in range(B)]
out, rows, dims = mapper.make_batch(pool, groups, params)
torch.cuda.synchronize()
canvas = mapper._keep[0].cpu().numpy()
off = 0
for b, (g, p) in enumerate(zip(groups, params)):
    dim = p["input_dim"]
    c4, l4 = A.mosaic4([imgs[i] for i in g], [labs[i] for i in g], dim, p["yc"], p["xc"])
    n = 4 * dim[0] * dim[1] * 3
    got = canvas[off:off + n].reshape(2 * dim[0], 2 * dim[1], 3); off += n
    bad = np.argwhere((got != c4).any(-1))
    print("sample", b, "groups", g, "dim", dim, "yc,xc", p["yc"], p["xc"], "canvas mismatches", len(bad))
    if len(bad):
        print("   y range", bad[:, 0].min(), bad[:, 0].max(), "x range", bad[:, 1].min(), bad[:, 1].max(), "first", bad[:5].tolist(),
              "got", got[tuple(bad[0])], "ref", c4[tuple(bad[0])])
    ref, t = A.random_perspective(c4, l4, p["draws"], [-dim[0] // 2, -dim[1] // 2])
    o = out[b, :, :ref.shape[0], :ref.shape[1]].cpu().numpy().transpose(1, 2, 0)
    bad = np.argwhere((o != ref).any(-1))
    print("   warp mismatches", len(bad), (bad[:5].tolist() if len(bad) else ""))
    if len(bad):
        print("   got", o[tuple(bad[0])], "ref", ref[tuple(bad[0])], "y range", bad[:, 0].min(), bad[:, 0].max(), "x range", bad[:, 1].min(), bad[:, 1].max())
