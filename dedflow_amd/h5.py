"""ctypes view of libdedflow_h5.so: the reference's HDF5 mesh / solution formats."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import api

_LIB = None
vp = C.c_void_p


def lib():
    global _LIB
    if _LIB is None:
        api.lib()  # libdedflow.so first (RTLD_GLOBAL): the H5 library links against it
        p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdedflow_h5.so")
        if not os.path.exists(p):
            raise api.MissingExtension(f"{p} not found: run `make`")
        L = C.CDLL(p, mode=C.RTLD_GLOBAL)
        L.H5OpenFile.restype = vp
        L.H5OpenFile.argtypes = [C.c_char_p, C.c_char_p]
        L.H5CloseFile.argtypes = [vp]
        L.H5GetDatasetSize.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32)]
        L.H5ReadDatasetf64.argtypes = [vp, C.c_char_p, vp]
        L.H5ReadDatasetInd.argtypes = [vp, C.c_char_p, vp]
        L.H5WriteDatasetf64.argtypes = [vp, C.c_char_p, C.c_int32, vp]
        L.H5WriteDatasetInd.argtypes = [vp, C.c_char_p, C.c_int32, vp]
        L.DflMeshWriteH5.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp]
        L.Mesh3DCreateH5.restype = C.POINTER(api.Mesh3D)
        L.Mesh3DCreateH5.argtypes = [vp, C.c_char_p]
        L.DflSolutionWriteH5.argtypes = [C.c_char_p, C.c_int32, vp, vp]
        L.DflSolutionReadH5.argtypes = [C.c_char_p, C.c_int32, vp, vp]
        _LIB = L
    return _LIB


def write_mesh(path, mesh, group="mesh"):
    L = lib()
    f = L.H5OpenFile(path.encode(), b"w")
    L.DflMeshWriteH5(f, group.encode(), mesh.num_node, mesh.num_tet, mesh.xg.ctypes.data, mesh.ien.ctypes.data, mesh.num_bound,
                     mesh.bound_node_offset.ctypes.data, mesh.bound_node.ctypes.data, mesh.bound_elem_offset.ctypes.data,
                     mesh.bound_ien.ctypes.data, mesh.bound_f2e.ctypes.data, mesh.bound_forn.ctypes.data)
    L.H5CloseFile(f)


def read_dataset(path, name, dtype):
    L = lib()
    f = L.H5OpenFile(path.encode(), b"r")
    n = C.c_int32(0)
    L.H5GetDatasetSize(f, name.encode(), C.byref(n))
    out = np.empty(n.value, dtype)
    if n.value:
        (L.H5ReadDatasetf64 if np.dtype(dtype) == np.float64 else L.H5ReadDatasetInd)(f, name.encode(), out.ctypes.data)
    L.H5CloseFile(f)
    return out
