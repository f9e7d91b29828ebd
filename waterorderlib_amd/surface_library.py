"""Host-side mirror of the density-field part of the reference's structureLibs/surface_library.py.

Only the piece that feeds the GPU path is here: `densityGrid` (surface_library.py:170-210)
builds an 80^3 grid around the solute and calls `wl.willarddensityfield`; the marching-cubes step after it is
scikit-image's and stays on the host.  `densityField` returns what the reference hands to marching cubes.
"""
import numpy as np

from . import waterlib as wl


def densityField(heavyPos, watPos, thisbox, nBins=81, smoothlen=2.4):
    """The grid construction and density call of surface_library.densityGrid (:172-199).

    thisbox is the reference's (1, 3) box array.  Returns (dens (80,80,80), dens_norm (80,80,80,3), (xSpan, ySpan,
    zSpan), (xSpace, ySpace, zSpace)) -- the arguments of the reference's measure.marching_cubes call."""
    heavyPos = np.asarray(heavyPos, dtype=np.float64)
    thisbox = np.asarray(thisbox, dtype=np.float64).reshape(1, -1)
    allMin = np.min(heavyPos); allMax = np.max(heavyPos)
    spans, spaces = [], []
    for _ in range(3):  # the reference uses the x box length for all three axes (:184-189)
        span = np.linspace(allMin - thisbox[0, 0] / 2.0, allMax + thisbox[0, 0] / 2.0, nBins).reshape(1, nBins)
        space = span[:, 1] - span[:, 0]
        spans.append(span[:, :-1] + space)
        spaces.append(space)
    dens, dens_norm = wl.willarddensityfield(watPos, spans[0], spans[1], spans[2], thisbox, smoothlen=smoothlen)
    return dens, dens_norm, tuple(spans), tuple(spaces)


def densityGrid(heavyPos, watPos, thisbox, level=0.016, minFrac=0.7):
    """surface_library.densityGrid (:170-210): instantaneous-interface mesh (verts, faces).  Needs scikit-image for
    the marching-cubes step, exactly as the reference does."""
    try:
        from skimage import measure
    except ImportError as e:  # pragma: no cover - scikit-image is not part of this image
        raise ImportError("densityGrid needs scikit-image (measure.marching_cubes), as the reference does; "
                          "densityField returns the density field without it") from e
    dens, _, _, sp = densityField(heavyPos, watPos, thisbox)
    verts, faces, norms, vals = measure.marching_cubes(dens, level, spacing=(sp[0], sp[1], sp[2]))
    verts = verts - np.min(heavyPos)
    verts = verts - 0.5 * np.max(verts)
    return verts, faces
