"""waterorderlib_amd -- MI355X (gfx950) implementation of WaterOrderLib's per-frame water
order-parameter hot path (tetrahedral q, 4-NN distances/angles) behind the reference's own call
surface.  Modules mirror the reference's:

    waterorderlib_amd.waterlib          <- f2py module `waterlib`   (fortran/waterlib.f90)
    waterorderlib_amd.water_properties  <- structureLibs/water_properties.py (hot-path functions)
    waterorderlib_amd.orderParam_lib    <- structureLibs/orderParam_lib.py   (tetOrderCalc ...)
    waterorderlib_amd.TrajObject        <- structureLibs/TrajObject.py       (array-backed)

All compute goes through the C ABI of libwol.so (include/wol.h) into hand-written HIP kernels;
there is no CPU fallback -- a missing library or GPU raises.
"""
__version__ = "0.1.0"
