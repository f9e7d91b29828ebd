"""Deterministic synthetic periodic water boxes (SURVEY 8d "Synthetic inputs").

Oxygen sites sit on a diamond-cubic (ice-Ic-like) lattice at the density the reference assumes
(water_properties.py:55, watdens = 0.033456 molecules/A^3), thinned by random vacancies to the
requested count, with independent Gaussian jitter per frame (sigma 0.3 A: liquid-like q
distribution, no exact distance ties).  Optional rigid H sites at 0.9572 A / 104.52 deg (the
geometry of water_properties.py:924) and a TIP4P-style virtual site give the interleaved
[O,H,H(,EP)] atom layout that TrajObject.getWatInds (TrajObject.py:35-52) indexes with
``lenWat``.  numpy's PCG64 is bit-reproducible across machines, so the dev container and the GPU
box see identical frames.
"""
import numpy as np

WATDENS = 0.033456  # water_properties.py:55

_DIAMOND = np.array([[0, 0, 0], [0, 2, 2], [2, 0, 2], [2, 2, 0],
                     [1, 1, 1], [1, 3, 3], [3, 1, 3], [3, 3, 1]], dtype=np.float64) / 4.0


def box_length(n_waters, density=WATDENS):
    return (n_waters / density) ** (1.0 / 3.0)


def lattice_sites(n_waters, seed=0, box=None):
    """n_waters O sites on a vacancy-thinned diamond lattice filling ``box`` (default cubic at WATDENS)."""
    rng = np.random.default_rng([seed, 0x5157])
    if box is None:
        L = box_length(n_waters)
        box = np.array([L, L, L])
    box = np.asarray(box, dtype=np.float64)
    m = int(np.ceil((n_waters / 8.0) ** (1.0 / 3.0)))
    g = np.arange(m, dtype=np.float64)
    cells = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 1, 3)
    sites = ((cells + _DIAMOND[None, :, :]) / m).reshape(-1, 3)
    keep = rng.permutation(len(sites))[:n_waters]
    keep.sort()
    return sites[keep] * box[None, :], box


def water_frames(n_waters, n_frames=1, seed=0, sigma=0.3, len_wat=1, box=None, wrap=True):
    """Returns (xyz[T, n_waters*len_wat, 3] float64, boxes[T,3]).  len_wat: 1 (O only), 3 (O,H,H)
    or 4 (O,H,H,EP).  Oxygen of water w is atom w*len_wat, as with the reference's watInds."""
    base, box = lattice_sites(n_waters, seed, box)
    out = np.empty((n_frames, n_waters * len_wat, 3))
    for t in range(n_frames):
        rng = np.random.default_rng([seed, 0xF4A3, t])
        o = base + rng.normal(0.0, sigma, size=base.shape)
        if wrap:
            o = o - box * np.floor(o / box)
            o = np.where(o >= box, 0.0, o)  # guard the x == L rounding case
        out[t, 0::len_wat] = o
        if len_wat >= 3:
            h1, h2, ep = _rigid_sites(rng, n_waters)
            out[t, 1::len_wat] = o + h1
            out[t, 2::len_wat] = o + h2
            if len_wat >= 4:
                out[t, 3::len_wat] = o + ep
    boxes = np.tile(box[None, :], (n_frames, 1))
    return out, boxes


def _rigid_sites(rng, n):
    # random orthonormal frames from two Gaussian vectors
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = rng.normal(size=(n, 3)); v -= u * np.sum(u * v, axis=1, keepdims=True)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    r, half = 0.9572, np.deg2rad(104.52) / 2.0
    h1 = r * (np.cos(half) * u + np.sin(half) * v)
    h2 = r * (np.cos(half) * u - np.sin(half) * v)
    ep = 0.1546 * u  # TIP4P-Ew M-site on the bisector
    return h1, h2, ep


def ideal_gas(n, seed=0, box=None):
    rng = np.random.default_rng([seed, 0x6A5])
    if box is None:
        L = box_length(n)
        box = np.array([L, L, L])
    box = np.asarray(box, dtype=np.float64)
    return rng.random((n, 3)) * box, box


def slab_frames(n_waters, n_frames=1, seed=0, sigma=0.3, gap_factor=0.5, n_vapour=100):
    """BASELINE config 5 geometry: a cubic water block of edge (n/rho)^(1/3) in a box elongated
    along z by ``gap_factor`` (vacuum gap >> 10 A cutoff) plus ``n_vapour`` isolated waters in the
    gap.  Returns (xyz[T, n_waters + n_vapour, 3], boxes[T,3])."""
    L = box_length(n_waters)
    base, cube = lattice_sites(n_waters, seed, np.array([L, L, L]))
    box = np.array([L, L, L * (1.0 + gap_factor)])
    rng0 = np.random.default_rng([seed, 0x51AB])
    vap = rng0.random((n_vapour, 3)) * np.array([L, L, L * gap_factor * 0.6]) + np.array([0, 0, L * (1 + 0.2 * gap_factor)])
    out = np.empty((n_frames, n_waters + n_vapour, 3))
    for t in range(n_frames):
        rng = np.random.default_rng([seed, 0xF4A3, t])
        o = base + rng.normal(0.0, sigma, size=base.shape)
        o[:, :2] -= box[:2] * np.floor(o[:, :2] / box[:2])
        out[t, :n_waters] = o
        out[t, n_waters:] = vap
    return out, np.tile(box[None, :], (n_frames, 1))
