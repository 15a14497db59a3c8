"""h5write / h5read of HDF5.jl (the two calls the reference's result block uses, src/monte_carlo.jl:334-343) over the system's
libhdf5 through ctypes — no h5py in the image, but the HDF5 C library is there (/opt/conda/lib). Datasets are written with the
array's shape in C order; a Julia array (column-major) of size (a, b) therefore corresponds to a NumPy array of shape (b, a)
with the same bytes, which is how HDF5.jl itself stores it. float64 / int32 / int64 only.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_LIB = None
hid_t = C.c_int64      # HDF5 >= 1.10; a 1.8 library (32-bit ids) is refused by _lib(), not mis-called


def _candidates():
    """resolved on first use, not at import: find_library spawns ldconfig / gcc subprocesses, which every process that merely
    imports the package (GPU workers included) should not pay for"""
    return (os.environ.get("TSAT_HDF5_LIB"), ctypes.util.find_library("hdf5"), "/opt/conda/lib/libhdf5.so", "libhdf5.so",
            "libhdf5_serial.so")


H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5T_INTEGER, H5T_FLOAT = 0, 1


def available():
    try:
        _lib()
        return True
    except OSError:
        return False


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    last = None
    for name in _candidates():
        if not name:
            continue
        try:
            lib = C.CDLL(name)
        except OSError as e:
            last = e
            continue
        for f, res, args in (("H5open", C.c_int, []), ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
                             ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]), ("H5Fclose", C.c_int, [hid_t]),
                             ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
                             ("H5Sclose", C.c_int, [hid_t]),
                             ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
                             ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Dclose", C.c_int, [hid_t]),
                             ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
                             ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
                             ("H5Dget_space", hid_t, [hid_t]), ("H5Dget_type", hid_t, [hid_t]),
                             ("H5Sget_simple_extent_ndims", C.c_int, [hid_t]),
                             ("H5Sget_simple_extent_dims", C.c_int, [hid_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
                             ("H5Tget_class", C.c_int, [hid_t]), ("H5Tget_size", C.c_size_t, [hid_t]), ("H5Tclose", C.c_int, [hid_t])):
            fn = getattr(lib, f)
            fn.restype, fn.argtypes = res, args
        if lib.H5open() < 0:
            last = OSError("H5open failed")
            continue
        maj, mnr, rel = C.c_uint(0), C.c_uint(0), C.c_uint(0)
        lib.H5get_libversion.restype, lib.H5get_libversion.argtypes = C.c_int, [C.POINTER(C.c_uint)] * 3
        if lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel)) < 0 or (maj.value, mnr.value) < (1, 10):
            last = OSError(f"{name}: HDF5 {maj.value}.{mnr.value}.{rel.value} has 32-bit identifiers; 1.10 or newer is needed")
            continue
        _LIB = lib
        return lib
    raise OSError(f"no usable libhdf5 ({last}); set TSAT_HDF5_LIB")


def _native(lib, dtype):
    sym = {np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g", np.dtype(np.int32): "H5T_NATIVE_INT32_g",
           np.dtype(np.int64): "H5T_NATIVE_INT64_g"}.get(np.dtype(dtype))
    if sym is None:
        raise TypeError(f"h5write: unsupported dtype {dtype}")
    return hid_t.in_dll(lib, sym).value


def h5write(path, name, data):
    """HDF5.jl's h5write(path, name, data): creates the file if it does not exist, adds dataset `name`."""
    lib = _lib()
    a = np.ascontiguousarray(data)
    if a.dtype not in (np.float64, np.int32, np.int64):
        a = a.astype(np.float64)
    f = lib.H5Fopen(path.encode(), H5F_ACC_RDWR, 0) if os.path.exists(path) else lib.H5Fcreate(path.encode(), H5F_ACC_TRUNC, 0, 0)
    if f < 0:
        raise OSError(f"h5write: cannot open {path}")
    sp = ds = -1
    try:
        ty = _native(lib, a.dtype)
        dims = (C.c_uint64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
        sp = lib.H5Screate_simple(max(a.ndim, 1), dims, None)
        if sp < 0:
            raise OSError(f"h5write: cannot create the dataspace of {name}")
        ds = lib.H5Dcreate2(f, name.encode(), ty, sp, 0, 0, 0)
        if ds < 0:
            raise OSError(f"h5write: cannot create dataset {name} in {path} (does it exist already?)")
        if lib.H5Dwrite(ds, ty, 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0:
            raise OSError(f"h5write: write of {name} failed")
    finally:
        if ds >= 0:
            lib.H5Dclose(ds)
        if sp >= 0:
            lib.H5Sclose(sp)
        lib.H5Fclose(f)


def h5read(path, name):
    lib = _lib()
    f = lib.H5Fopen(path.encode(), H5F_ACC_RDONLY, 0)
    if f < 0:
        raise OSError(f"h5read: cannot open {path}")
    ds = sp = ty = -1
    try:
        ds = lib.H5Dopen2(f, name.encode(), 0)
        if ds < 0:
            raise KeyError(name)
        sp, ty = lib.H5Dget_space(ds), lib.H5Dget_type(ds)
        if sp < 0 or ty < 0:
            raise OSError(f"h5read: cannot query {name}")
        nd = lib.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * max(nd, 1))()
        lib.H5Sget_simple_extent_dims(sp, dims, None)
        cls, size = lib.H5Tget_class(ty), lib.H5Tget_size(ty)
        dt = {(H5T_FLOAT, 8): np.float64, (H5T_INTEGER, 4): np.int32, (H5T_INTEGER, 8): np.int64}.get((cls, size))
        if dt is None:
            raise TypeError(f"h5read: unsupported stored type (class {cls}, {size} bytes)")
        out = np.empty(tuple(dims[i] for i in range(nd)), dtype=dt)
        if lib.H5Dread(ds, _native(lib, dt), 0, 0, 0, out.ctypes.data_as(C.c_void_p)) < 0:
            raise OSError(f"h5read: read of {name} failed")
        return out
    finally:
        if ty >= 0:
            lib.H5Tclose(ty)
        if sp >= 0:
            lib.H5Sclose(sp)
        if ds >= 0:
            lib.H5Dclose(ds)
        lib.H5Fclose(f)
