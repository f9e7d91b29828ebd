"""A/B of library builds on the host-fed path inside ONE gpurun call (boxes differ a lot in host memory speed):
    python tools/ab_host.py lib1.so lib2.so ...   -- each build in its own subprocess, three alternating rounds"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from waterorderlib_amd import _lib, engine, synth
n, F, lw = 100000, 256, 4
base, box = synth.lattice_sites(n, seed=2026)
rng = np.random.default_rng(0)
O = base[None] + 0.3 * rng.standard_normal((F, n, 3)); O -= box * np.floor(O / box)
host = np.zeros((F, n * lw, 3)); host[:, 0::lw] = O
pin = torch.from_numpy(host).pin_memory().numpy()
boxes = np.tile(box[None], (F, 1))
ctx = _lib.Context(0)
out = []
for name, arr in (("pageable", host), ("pinned", pin)):
    for th in (8, 16):
        ctx.set_option("host_pack", 1); ctx.set_option("host_threads", th)
        engine.tet_order_traj(arr[:64], boxes[:64], len_wat=lw, ctx=ctx)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); engine.tet_order_traj(arr, boxes, len_wat=lw, ctx=ctx); best = min(best, time.perf_counter() - t0)
        out.append("%%s %%2d thr %%6.0f" %% (name, th, F / best))
print("  |  ".join(out))
''' % ROOT
for rnd in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ, WOL_LIB_PATH=os.path.join(ROOT, lib))
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print("%-36s %s" % (lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
