#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds the *real* reference (WaterOrderLib's Fortran,
# compiled where it lies under /root/reference) into oracle/_ref/ so that the C
# restatement in oracle/wol_oracle.c and the golden fixtures in tests/golden/ can be
# pinned against it.  Nothing here is linked, imported or executed by the product
# (waterorderlib_amd/); oracle/_ref/ is git-ignored and no reference source is copied.
#
# What it does (replaces fortran/buildWrappers.sh:3-5, which needs gfortran and
# `numpy.f2py -c`, neither usable in this image):
#   for m in waterlib imagelib sortlib:
#     numpy.f2py <ref>/fortran/m.f90 -m m --lower   -> C/API wrapper sources (generated)
#     gcc   -c  mmodule.c fortranobject.c
#     flang -c  <ref>/fortran/m.f90  (+ the f77 wrapper file if non-empty)
#     flang -shared -> oracle/_ref/fortran/m.cpython-310-x86_64-linux-gnu.so
#   and an empty oracle/_ref/structureLibs/ to serve as the cwd that
#   structureLibs/water_properties.py:15-43 expects (it does os.chdir('../fortran')).
# Floating point: -O2, baseline x86-64, -ffp-contract=off (no FMA), matching a stock
# gfortran -O2 x86-64 build of the reference.
set -euo pipefail
REF=${WOL_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/fortran" ]; then
  echo "build_ref.sh: $REF/fortran not present (GPU box?) -- nothing to do" >&2
  exit 0
fi
FLANG=${FLANG:-/opt/rocm/lib/llvm/bin/flang}
PY=${PYTHON:-python3}
mkdir -p "$OUT/build" "$OUT/fortran" "$OUT/structureLibs"
NPINC=$($PY -c "import numpy; print(numpy.get_include())")
F2PYSRC=$($PY -c "import numpy.f2py as f, os; print(os.path.join(os.path.dirname(f.__file__), 'src'))")
PYINC=$($PY -c "import sysconfig; print(sysconfig.get_paths()['include'])")
EXT=$($PY -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
FFLAGS="-O2 -fPIC -ffp-contract=off"
cd "$OUT/build"
gcc -O2 -fPIC -I"$PYINC" -I"$NPINC" -I"$F2PYSRC" -c "$F2PYSRC/fortranobject.c" -o fortranobject.o
for m in waterlib imagelib sortlib; do
  if [ "$OUT/fortran/$m$EXT" -nt "$REF/fortran/$m.f90" ]; then continue; fi
  $PY -m numpy.f2py "$REF/fortran/$m.f90" -m $m --lower > f2py_$m.log 2>&1
  gcc -O2 -fPIC -I"$PYINC" -I"$NPINC" -I"$F2PYSRC" -c ${m}module.c -o ${m}module.o
  $FLANG $FFLAGS -c "$REF/fortran/$m.f90" -o ${m}_f.o
  OBJS="${m}module.o fortranobject.o ${m}_f.o"
  if [ -s "$m-f2pywrappers.f" ]; then
    $FLANG $FFLAGS -c "$m-f2pywrappers.f" -o ${m}_w.o
    OBJS="$OBJS ${m}_w.o"
  fi
  $FLANG -shared -o "$OUT/fortran/$m$EXT" $OBJS
done
# plain shared library with the Fortran-ABI symbols (allnearneighbors_, reimage_, ...)
# for ctypes-level checks of the C restatement
if [ ! "$OUT/libwaterlib_ref.so" -nt "$REF/fortran/waterlib.f90" ]; then
  $FLANG $FFLAGS -shared "$REF/fortran/waterlib.f90" -o "$OUT/libwaterlib_ref.so"
fi
echo "build_ref.sh: reference modules in $OUT/fortran"
