import sys, ctypes as C, numpy as np
from pathlib import Path; R0 = Path(__file__).resolve().parent.parent; sys.path.insert(0, str(R0)); sys.path.insert(0, str(R0 / "tests"))
from test_gpu_cabi import Ctx, sphere
rng = np.random.default_rng(7)
for n in (1800, 2000, 2200, 2600):
    c = rng.uniform(-10, 10, (n, 3)).astype(np.float32); r = np.exp(rng.normal(-1, 1, n)).astype(np.float32) * 0.3
    ctx = Ctx()
    sc = ctx.upload(spheres=[sphere(tuple(ci), float(ri), order=i, ti=0, oi=i) for i, (ci, ri) in enumerate(zip(c, r))])
    ctx.render(sc, 64, 64, 1)
    import time
    ctx.render(sc, 1920, 1080, 2)
    t0 = time.perf_counter(); ctx.render(sc, 1920, 1080, 32); dt = time.perf_counter() - t0
    print(f'n={n}: {1920*1080*32/dt/1e6:.0f} Msamples/s (render call incl. readback)', flush=True)
    ctx.close()
