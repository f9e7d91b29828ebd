"""TEST INFRASTRUCTURE ONLY.  Imports the *reference's own* Python module
``structureLibs/water_properties.py`` from where it lies under /root/reference, against the
f2py modules that ``oracle/build_ref.sh`` compiled from the reference's Fortran into
``oracle/_ref/fortran``.  Only usable in the development container (the reference does not
travel to the GPU box); used by ``tests/golden/make_golden.py`` to generate the committed
fixtures and by ``tests/test_oracle_vs_reference.py`` (skipped when the reference is absent).

water_properties.py:15-43 does ``os.chdir('../fortran')``, counts ``*.so`` files carrying the
running Python's version tag, (re)builds them if fewer than three match, chdirs back and
imports ``waterlib`` from ``../fortran``.  We satisfy that by making the process cwd
``oracle/_ref/structureLibs`` (an empty directory whose sibling ``fortran/`` holds the three
freshly built modules) for the duration of the import.  No reference file is copied.
"""
import importlib
import sys as _sys
_sys.dont_write_bytecode = True
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("WOL_REFERENCE", "/root/reference")
STAGE = os.path.join(HERE, "_ref")


def reference_available():
    return (os.path.isfile(os.path.join(REF_ROOT, "structureLibs", "water_properties.py"))
            and os.path.isdir(os.path.join(STAGE, "fortran"))
            and len([f for f in os.listdir(os.path.join(STAGE, "fortran")) if f.endswith(".so")]) >= 3)


_cached = None


def load_reference():
    """Returns (water_properties module, waterlib f2py module) of the reference."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference not available: run oracle/build_ref.sh in the dev container")
    cwd = os.getcwd()
    ref_py = os.path.join(REF_ROOT, "structureLibs")
    fdir = os.path.join(STAGE, "fortran")
    os.chdir(os.path.join(STAGE, "structureLibs"))
    # absolute entry so that `import waterlib` keeps resolving after we chdir back
    sys.path.insert(0, fdir)
    sys.path.insert(0, ref_py)
    try:
        wp = importlib.import_module("water_properties")
        wl = importlib.import_module("waterlib")
    finally:
        os.chdir(cwd)
        sys.path.remove(ref_py)
    _cached = (wp, wl)
    return _cached
