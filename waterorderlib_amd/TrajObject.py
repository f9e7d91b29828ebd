"""Array-backed stand-in for the reference's structureLibs/TrajObject.py (:24-52).

The reference's TrajObject wraps parmed/pytraj (not installed here, and I/O is out of scope); the
hot path only touches three things on it -- ``.traj`` (iterable of frames with ``.xyz`` and
``.box.values``), ``.top`` and ``getWatInds()`` -- so this class provides exactly those over
in-memory arrays.  A real pytraj-backed TrajObject can be passed to
``orderParam_lib.tetOrderCalc`` unchanged: it is duck-typed on the same attributes.
"""
import numpy as np


class _Box:
    def __init__(self, lengths):
        v = np.zeros(6)
        v[:3] = lengths
        v[3:] = 90.0
        self.values = v


class Frame:
    def __init__(self, xyz, box):
        self.xyz = xyz
        self.box = _Box(box)


class ArrayTrajectory:
    def __init__(self, xyz, boxes, stride=1):
        self.xyz_all = np.asarray(xyz)[::stride]
        self.boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 3)[::stride]

    def __len__(self):
        return len(self.xyz_all)

    def __iter__(self):
        for x, b in zip(self.xyz_all, self.boxes):
            yield Frame(x, b)


class TrajObject:
    """TrajObject(xyz[T,natoms,3], boxes[T,3], lenWat, stride=1): waters are the first
    nWaters*lenWat atoms, oxygen first in each water (AMBER WAT residue order O,H1,H2[,EPW])."""

    def __init__(self, xyz, boxes, lenWat=1, nWaters=None, stride=1, solInds=None):
        self.solInds = None if solInds is None else np.asarray(solInds, dtype=int)
        self.traj = ArrayTrajectory(xyz, boxes, stride)
        self.top = None
        self.stride = stride
        natoms = self.traj.xyz_all.shape[1]
        self.lenWat = int(lenWat)
        self.nWaters = natoms // self.lenWat if nWaters is None else int(nWaters)

    def getWatInds(self):
        """(watInds, watHInds, lenWat) as TrajObject.py:35-52: oxygen and hydrogen atom indices."""
        base = np.arange(self.nWaters) * self.lenWat
        watInds = base.copy()
        watHInds = np.sort(np.concatenate([base + 1, base + 2])) if self.lenWat >= 3 else np.array([], dtype=int)
        return watInds, watHInds, self.lenWat

    def getSolInds(self):
        """(solInds, solHInds, solCInds, solNInds, solOInds, solSInds) as TrajObject.py:54-78; only the heavy-atom
        list (given at construction) is known to an array-backed trajectory"""
        e = np.array([], dtype=int)
        return (e if self.solInds is None else self.solInds), e, e, e, e, e
