"""Seeded input generators shared by the soak scripts (tools/soak_random.py, soak_traj.py, soak_lattice.py) and the
bounded slices of them that run under `pytest -m gpu` (tests/test_gpu_round3.py).  These generators found all three
correctness bugs of round 2; a seed means the same input in both places."""
import numpy as np


def random_case(seed, nmax=20000):
    """random_case(seed) -> (pos[n,3], box[3]): box shapes (incl. axes barely above the 13.6 A minimum), densities from
    vapour to twice liquid, every other seed with a third of the waters in five dense blobs, waters left unwrapped"""
    rng = np.random.default_rng(seed)
    box = rng.uniform(13.6, 120.0, size=3)
    if seed % 4 == 0:
        box[rng.integers(3)] = rng.uniform(13.6, 15.0)
    rho = rng.choice([0.0334, 0.0334, 0.02, 0.05, 0.01, 0.07])
    n = int(min(max(rho * np.prod(box), 300), nmax))
    pos = rng.random((n, 3)) * box
    if seed % 2 == 1:
        k = n // 3
        centres = rng.random((5, 3)) * box
        pos[:k] = centres[rng.integers(5, size=k)] + rng.normal(0.0, 2.5, size=(k, 3))
    pos += rng.integers(-1, 2, size=pos.shape) * box
    return pos, box


def traj_case(seed, nmax=None):
    """traj_case(seed) -> dict(frames[T, n*len_wat, 3], boxes[T,3], len_wat, lo, hi, want_q, f32): frame counts, water
    counts, per-frame boxes (NPT), interleaved waters, shell cuts, float32 storage, frames thousands of boxes away"""
    rng = np.random.default_rng(1000 + seed)
    big = seed >= 100                                     # seeds from 100: larger boxes, fewer frames
    T = int(rng.choice([1, 2, 3, 7, 16] if big else [1, 2, 3, 7, 16, 33, 70]))
    box0 = rng.uniform(14.0, 110.0 if big else 70.0, size=3)
    if seed % 5 == 0:
        box0[rng.integers(3)] = rng.uniform(13.7, 15.0)
    rho = rng.choice([0.0334, 0.0334, 0.025, 0.045, 0.012])
    cap = (40000 if big else 12000) if nmax is None else nmax
    n = int(min(max(rho * np.prod(box0), 200), cap))
    if seed >= 10000:                                     # seeds from 10000: many frames of 25-60 k waters (the single-pass
        T = int(rng.integers(64, 81))                     # first sort level, full batches)
        n = int(rng.integers(25000, 60000))
        box0 = box0 * (n / (rho * np.prod(box0))) ** (1.0 / 3.0)
    npt = seed % 3 != 0
    boxes = box0 * (1.0 + (rng.uniform(-0.04, 0.04, size=(T, 3)) if npt else np.zeros((T, 3))))
    len_wat = int(rng.choice([1, 1, 3, 4]))
    frames = np.zeros((T, n * len_wat, 3))
    base = rng.random((n, 3))
    for f in range(T):
        p = (base + rng.normal(0.0, 0.01, size=(n, 3))) * boxes[f]
        if seed % 4 == 1:   # a dense droplet in every frame
            k = n // 4
            p[:k] = boxes[f] * 0.5 + rng.normal(0.0, 3.0, size=(k, 3))
        p += rng.integers(-1, 2, size=p.shape) * boxes[f]
        frames[f, ::len_wat] = p
        if len_wat > 1:
            frames[f, 1::len_wat] = 1.0e6     # other atoms of the molecule: must never be read as waters
    hi = float(rng.choice([10.0, 10.0, 3.5, 5.0]))
    lo = float(rng.choice([0.0, 0.0, 2.7]))
    want_q = bool(seed % 2)
    f32 = seed % 7 == 3                                   # float32 storage: promoted exactly, same fp64 arithmetic
    if f32:
        frames = frames.astype(np.float32)
    if seed % 11 == 5:
        frames = frames + (1000.0 * boxes[:, None, :]).astype(frames.dtype)    # far outside the box
    return dict(frames=frames, boxes=boxes, len_wat=len_wat, lo=lo, hi=hi, want_q=want_q, f32=f32, npt=npt, T=T, n=n)


_BASES = {
    "sc": np.array([[0, 0, 0]], float),
    "fcc": np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0]], float),
    "diamond": np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                         [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]], float),
}


def lattice_case(seed):
    """lattice_case(seed) -> (pos, box, kind, jitter): perfect lattices (six / twelve / four + twelve equidistant
    neighbours), with and without a jitter at the level of the last bits, whole-box shifts"""
    rng = np.random.default_rng(5000 + seed)
    kind = ["sc", "fcc", "diamond"][seed % 3]
    a = {"sc": 3.1, "fcc": 4.4, "diamond": 6.35}[kind] * rng.choice([1.0, 1.0, 0.97, 1.1])
    reps = rng.integers(4, 9, size=3) if kind != "sc" else rng.integers(6, 16, size=3)
    if seed >= 100:
        reps = reps * 2                           # seeds from 100: eight times the waters
    g = np.stack(np.meshgrid(*[np.arange(r) for r in reps], indexing="ij"), axis=-1).reshape(-1, 1, 3)
    pos = ((g + _BASES[kind][None]) * a).reshape(-1, 3)
    box = reps * a
    jitter = [0.0, 0.0, 1e-13, 1e-9, 1e-4][seed % 5]
    if jitter:
        pos = pos + rng.normal(0.0, jitter, size=pos.shape)
    pos = pos[rng.permutation(len(pos))]
    if seed % 4 == 1:
        pos = pos + rng.integers(-2, 3, size=pos.shape) * box
    if seed % 4 == 2:
        pos = pos + rng.random(3) * box           # off the cell walls
    return pos, box.astype(np.float64), kind, jitter
