"""MPAS NetCDF files either side of the hot path: the mesh / vertical-mesh /
initial-state readers and the output writer of the reference, restated for this package.

    read_mesh(path)                 readPrimaryMesh / readDualMesh / readEdgeInfo + signIndexField!
                                    (src/infra/MPASMesh/HorzMesh.jl:166-290, 292-332)
    read_vertical_mesh(path, mesh)  VerticalMesh(mesh_fp, mesh)            (src/infra/MPASMesh/VertMesh.jl:46-82)
    read_initial_state(path, ...)   PrognosticVars(config, mesh)           (src/ocn/PrognosticVars.jl:59-106)
    write_output(path, ...)         write_netcdf(Setup, Diag, Prog)        (src/infra/OutPut.jl:117-215)
    write_mesh(path, mesh, ...)     (no reference counterpart: MPAS tools write such files; used for round trips)

Formats.  NetCDF classic / 64-bit offset (CDF-1/2) through scipy.io.netcdf_file, and NetCDF-4 (= HDF5) through
ctypes on the libhdf5 of this image -- there is no netCDF4 / h5py / xarray here.  A NetCDF-4 variable is an HDF5
dataset of the same name in the root group and a dimension is the extent of those datasets, which is all the readers
need.  The writers produce 64-bit-offset NetCDF-3 (the reference writes NetCDF-4 through NCDatasets; every NetCDF
reader opens both).  Host-only: nothing here touches the GPU or the library.

On disk (C order) MPAS stores e.g. edgesOnCell(nCells, maxEdges): exactly Julia's column-major (maxEdges, nCells) and
exactly the (n, slots) arrays of meshgen.MeshData, so no transposition happens anywhere.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .meshgen import MeshData, sign_index_fields

I32 = np.int32
_HDF5_CANDIDATES = ("libhdf5.so", "/opt/conda/lib/libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200")


class MpasIOError(RuntimeError):
    """error("...") / KeyError of the reference readers."""


# ------------------------------------------------------------------------------------------------
# NetCDF-4 = HDF5 through ctypes
# ------------------------------------------------------------------------------------------------
_h5 = None


def hdf5():
    """The HDF5 C library, or None when this machine has none (callers raise a clear error)."""
    global _h5
    if _h5 is None:
        for name in _HDF5_CANDIDATES:
            try:
                lib = C.CDLL(name)
            except OSError:
                continue
            hid, i64p = C.c_int64, C.POINTER(C.c_uint64)
            lib.H5open.restype = C.c_int
            lib.H5Fopen.argtypes, lib.H5Fopen.restype = [C.c_char_p, C.c_uint, hid], hid
            lib.H5Fcreate.argtypes, lib.H5Fcreate.restype = [C.c_char_p, C.c_uint, hid, hid], hid
            lib.H5Fclose.argtypes = [hid]
            lib.H5Lexists.argtypes, lib.H5Lexists.restype = [hid, C.c_char_p, hid], C.c_int
            lib.H5Dopen2.argtypes, lib.H5Dopen2.restype = [hid, C.c_char_p, hid], hid
            lib.H5Dcreate2.argtypes, lib.H5Dcreate2.restype = [hid, C.c_char_p, hid, hid, hid, hid, hid], hid
            lib.H5Dget_space.argtypes, lib.H5Dget_space.restype = [hid], hid
            lib.H5Dget_type.argtypes, lib.H5Dget_type.restype = [hid], hid
            lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
            lib.H5Dwrite.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
            lib.H5Dclose.argtypes = [hid]
            lib.H5Screate_simple.argtypes, lib.H5Screate_simple.restype = [C.c_int, i64p, i64p], hid
            lib.H5Screate.argtypes, lib.H5Screate.restype = [C.c_int], hid
            lib.H5Sget_simple_extent_ndims.argtypes = [hid]
            lib.H5Sget_simple_extent_dims.argtypes = [hid, i64p, i64p]
            lib.H5Sclose.argtypes = [hid]
            lib.H5Tget_class.argtypes = [hid]
            lib.H5Tget_size.argtypes, lib.H5Tget_size.restype = [hid], C.c_size_t
            lib.H5Tis_variable_str.argtypes = [hid]
            lib.H5Tcopy.argtypes, lib.H5Tcopy.restype = [hid], hid
            lib.H5Tset_size.argtypes = [hid, C.c_size_t]
            lib.H5Tclose.argtypes = [hid]
            lib.H5Aexists.argtypes = [hid, C.c_char_p]
            lib.H5Aopen.argtypes, lib.H5Aopen.restype = [hid, C.c_char_p, hid], hid
            lib.H5Acreate2.argtypes, lib.H5Acreate2.restype = [hid, C.c_char_p, hid, hid, hid, hid], hid
            lib.H5Aget_type.argtypes, lib.H5Aget_type.restype = [hid], hid
            lib.H5Aread.argtypes = [hid, hid, C.c_void_p]
            lib.H5Awrite.argtypes = [hid, hid, C.c_void_p]
            lib.H5Aclose.argtypes = [hid]
            lib.H5Eset_auto2.argtypes = [hid, C.c_void_p, C.c_void_p]
            lib.H5open()
            lib.H5Eset_auto2(0, None, None)            # errors come back as return codes, not as stderr dumps
            _h5 = lib
            break
        else:
            _h5 = False
    return _h5 or None


def _h5_native(lib, dtype):
    sym = {np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g", np.dtype(np.float32): "H5T_NATIVE_FLOAT_g",
           np.dtype(np.int32): "H5T_NATIVE_INT_g", np.dtype(np.int64): "H5T_NATIVE_LLONG_g"}[np.dtype(dtype)]
    return C.c_int64.in_dll(lib, sym).value


class _H5File:
    """Read-only view of the root group of an HDF5 / NetCDF-4 file."""

    H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3

    def __init__(self, path):
        self.lib = hdf5()
        if self.lib is None:
            raise MpasIOError(f"{path} is an HDF5/NetCDF-4 file and no libhdf5 could be loaded on this machine")
        self.f = self.lib.H5Fopen(os.fsencode(path), 0, 0)
        if self.f < 0:
            raise MpasIOError(f"cannot open {path} as HDF5")

    def close(self):
        if self.f >= 0:
            self.lib.H5Fclose(self.f)
            self.f = -1

    def has(self, name):
        return self.lib.H5Lexists(self.f, name.encode(), 0) > 0

    def var(self, name):
        L = self.lib
        if not self.has(name):
            raise MpasIOError(f"variable {name} not found")
        d = L.H5Dopen2(self.f, name.encode(), 0)
        if d < 0:
            raise MpasIOError(f"{name} is not a dataset")
        try:
            sp, tp = L.H5Dget_space(d), L.H5Dget_type(d)
            nd = L.H5Sget_simple_extent_ndims(sp)
            dims = (C.c_uint64 * max(nd, 1))()
            L.H5Sget_simple_extent_dims(sp, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            cls, size = L.H5Tget_class(tp), L.H5Tget_size(tp)
            if cls == self.H5T_FLOAT:
                dt = np.float64 if size == 8 else np.float32
            elif cls == self.H5T_INTEGER:
                dt = np.int64 if size == 8 else np.int32
            else:
                raise MpasIOError(f"{name}: unsupported HDF5 type class {cls}")
            out = np.empty(shape, dtype=dt)
            if out.size and L.H5Dread(d, _h5_native(L, dt), 0, 0, 0, out.ctypes.data_as(C.c_void_p)) < 0:
                raise MpasIOError(f"reading {name} failed")
            L.H5Sclose(sp); L.H5Tclose(tp)
            return out
        finally:
            L.H5Dclose(d)

    def attr(self, name):
        L = self.lib
        if L.H5Aexists(self.f, name.encode()) <= 0:
            return None
        a = L.H5Aopen(self.f, name.encode(), 0)
        tp = L.H5Aget_type(a)
        try:
            cls, size = L.H5Tget_class(tp), L.H5Tget_size(tp)
            if cls == self.H5T_STRING:
                if L.H5Tis_variable_str(tp) > 0:
                    ptr = C.c_char_p()
                    L.H5Aread(a, tp, C.byref(ptr))
                    return (ptr.value or b"").decode()
                buf = C.create_string_buffer(size + 1)
                L.H5Aread(a, tp, buf)
                return buf.value.decode().rstrip("\x00 ")
            if cls == self.H5T_FLOAT:
                v = C.c_double()
                L.H5Aread(a, _h5_native(L, np.float64), C.byref(v))
                return v.value
            v = C.c_int64()
            L.H5Aread(a, _h5_native(L, np.int64), C.byref(v))
            return v.value
        finally:
            L.H5Tclose(tp); L.H5Aclose(a)


class _NC3File:
    def __init__(self, path):
        from scipy.io import netcdf_file
        self.f = netcdf_file(path, "r", mmap=False)

    def close(self):
        self.f.close()

    def has(self, name):
        return name in self.f.variables

    def var(self, name):
        if name not in self.f.variables:
            raise MpasIOError(f"variable {name} not found")
        return np.array(self.f.variables[name][...])

    def attr(self, name):
        v = getattr(self.f, name, None)
        if isinstance(v, bytes):
            v = v.decode()
        return v


def open_dataset(path):
    """NCDataset(path, "r"): classic NetCDF or NetCDF-4/HDF5, by the file's magic bytes."""
    with open(path, "rb") as fh:
        magic = fh.read(8)
    if magic[:3] == b"CDF":
        if magic[3] not in (1, 2):
            raise MpasIOError(f"{path}: NetCDF CDF-{magic[3]} is not supported (classic, 64-bit offset and NetCDF-4 are)")
        return _NC3File(path)
    if magic == b"\x89HDF\r\n\x1a\n":
        return _H5File(path)
    raise MpasIOError(f"{path} is neither a NetCDF classic nor an HDF5/NetCDF-4 file")


# ------------------------------------------------------------------------------------------------
# readers
# ------------------------------------------------------------------------------------------------
def _f64(ds, name, n, default_zero=False):
    if not ds.has(name):
        if default_zero:                          # fCell / fEdge / fVertex default to zeros (HorzMesh.jl:177-182)
            return np.zeros(n)
        raise MpasIOError(f"variable {name} not found")
    return np.ascontiguousarray(ds.var(name), dtype=np.float64).reshape(-1)


def _i32(ds, name):
    return np.ascontiguousarray(ds.var(name), dtype=I32)


def read_mesh(path) -> MeshData:
    """HorzMesh: PrimaryCells, DualCells, Edges as the reference reads them, then both signIndexField!s."""
    ds = open_dataset(path)
    try:
        xC = _f64(ds, "xCell", 0)
        nC = xC.size
        eoc = _i32(ds, "edgesOnCell")
        if eoc.ndim != 2 or eoc.shape[0] != nC:
            raise MpasIOError("edgesOnCell must be (nCells, maxEdges)")
        maxEdges = eoc.shape[1]
        xE = _f64(ds, "xEdge", 0)
        nE = xE.size
        xV = _f64(ds, "xVertex", 0)
        nV = xV.size
        eov = _i32(ds, "edgesOnVertex")
        vertexDegree = eov.shape[1]
        eoe = _i32(ds, "edgesOnEdge")
        coe, voe = _i32(ds, "cellsOnEdge"), _i32(ds, "verticesOnEdge")
        neoc = _i32(ds, "nEdgesOnCell").reshape(-1)
        esc, esv = sign_index_fields(coe, voe, neoc, eoc, eov, maxEdges, vertexDegree)
        on_sphere = str(ds.attr("on_a_sphere") or "NO").strip().upper().startswith("Y")
        radius = ds.attr("sphere_radius")
        periodic = str(ds.attr("is_periodic") or "NO").strip().upper() == "YES"
        return MeshData(
            nCells=nC, nEdges=nE, nVertices=nV, maxEdges=maxEdges, maxEdges2=eoe.shape[1], vertexDegree=vertexDegree,
            xCell=xC, yCell=_f64(ds, "yCell", nC), zCell=_f64(ds, "zCell", nC), fCell=_f64(ds, "fCell", nC, True),
            areaCell=_f64(ds, "areaCell", nC), nEdgesOnCell=neoc, edgesOnCell=eoc,
            verticesOnCell=_i32(ds, "verticesOnCell"), cellsOnCell=_i32(ds, "cellsOnCell"), edgeSignOnCell=esc,
            xEdge=xE, yEdge=_f64(ds, "yEdge", nE), zEdge=_f64(ds, "zEdge", nE), fEdge=_f64(ds, "fEdge", nE, True),
            dvEdge=_f64(ds, "dvEdge", nE), dcEdge=_f64(ds, "dcEdge", nE), angleEdge=_f64(ds, "angleEdge", nE),
            nEdgesOnEdge=_i32(ds, "nEdgesOnEdge").reshape(-1), cellsOnEdge=coe, verticesOnEdge=voe, edgesOnEdge=eoe,
            weightsOnEdge=np.ascontiguousarray(ds.var("weightsOnEdge"), dtype=np.float64),
            xVertex=xV, yVertex=_f64(ds, "yVertex", nV), zVertex=_f64(ds, "zVertex", nV),
            fVertex=_f64(ds, "fVertex", nV, True), areaTriangle=_f64(ds, "areaTriangle", nV), edgesOnVertex=eov,
            cellsOnVertex=_i32(ds, "cellsOnVertex"), edgeSignOnVertex=esv, on_sphere=on_sphere,
            sphere_radius=float(radius) if radius is not None else 0.0, is_periodic=periodic,
            meta={"source": os.fspath(path)},
            kiteAreasOnVertex=(np.ascontiguousarray(ds.var("kiteAreasOnVertex"), dtype=np.float64)
                               if ds.has("kiteAreasOnVertex") else None))
    finally:
        ds.close()


def read_vertical_mesh(path, mesh: MeshData):
    """VerticalMesh(mesh_fp, mesh) (VertMesh.jl:46-82): returns dict(nVertLevels, minLevelCell, maxLevelCell,
    restingThickness (nCells, K), restingThicknessSum (nCells), stacked).  Raises for non-periodic meshes (:50-52);
    a mesh that is not stacked is only reported (`stacked` False), as the reference only logs it (:60-65)."""
    ds = open_dataset(path)
    try:
        if str(ds.attr("is_periodic") or "").strip().upper() != "YES":
            raise MpasIOError("Support for non-periodic meshes is not yet implemented")
        rest = np.ascontiguousarray(ds.var("restingThickness"), dtype=np.float64)
        if rest.ndim == 3:
            rest = rest[0]                         # ds["restingThickness"][:,:,1]: first time record
        if rest.ndim != 2 or rest.shape[0] != mesh.nCells:
            raise MpasIOError("restingThickness must be (Time, nCells, nVertLevels)")
        K = rest.shape[1]
        minL, maxL = _i32(ds, "minLevelCell").reshape(-1), _i32(ds, "maxLevelCell").reshape(-1)
        return {"nVertLevels": K, "minLevelCell": minL, "maxLevelCell": maxL, "restingThickness": rest,
                "restingThicknessSum": rest.sum(axis=1), "stacked": bool(np.all(maxL == K))}
    finally:
        ds.close()


def read_initial_state(path, mesh: MeshData, nVertLevels: int):
    """ssh (nCells), normalVelocity (nEdges, K), layerThickness (nCells, K): first time record of the input stream
    (PrognosticVars.jl:95-99)."""
    ds = open_dataset(path)
    try:
        def first_record(name, shape):
            a = np.ascontiguousarray(ds.var(name), dtype=np.float64)
            if a.ndim == len(shape) + 1:
                a = a[0]
            if a.shape != shape:
                raise MpasIOError(f"{name} has shape {a.shape}, expected (Time,) + {shape}")
            return a
        return (first_record("ssh", (mesh.nCells,)), first_record("normalVelocity", (mesh.nEdges, nVertLevels)),
                first_record("layerThickness", (mesh.nCells, nVertLevels)))
    finally:
        ds.close()


# ------------------------------------------------------------------------------------------------
# writers (NetCDF-3, 64-bit offset)
# ------------------------------------------------------------------------------------------------
def _nc_create(path):
    from scipy.io import netcdf_file
    return netcdf_file(path, "w", version=2)


def _put(f, name, typ, dims, data):
    v = f.createVariable(name, typ, dims)
    v[...] = data
    return v


def write_mesh(path, mesh: MeshData, restingThickness=None, state=None):
    """An MPAS mesh (+ optional vertical mesh and initial state) file with the variables the readers above expect."""
    f = _nc_create(path)
    f.createDimension("Time", 1)
    for n, v in (("nCells", mesh.nCells), ("nEdges", mesh.nEdges), ("nVertices", mesh.nVertices),
                 ("maxEdges", mesh.maxEdges), ("maxEdges2", mesh.maxEdges2), ("TWO", 2),
                 ("vertexDegree", mesh.vertexDegree)):
        f.createDimension(n, int(v))
    f.on_a_sphere = "YES" if mesh.on_sphere else "NO"
    f.sphere_radius = float(mesh.sphere_radius)
    f.is_periodic = "YES" if mesh.is_periodic else "NO"
    for n in ("xCell", "yCell", "zCell", "fCell", "areaCell"):
        _put(f, n, "d", ("nCells",), getattr(mesh, n))
    for n in ("xEdge", "yEdge", "zEdge", "fEdge", "dvEdge", "dcEdge", "angleEdge"):
        _put(f, n, "d", ("nEdges",), getattr(mesh, n))
    for n in ("xVertex", "yVertex", "zVertex", "fVertex", "areaTriangle"):
        _put(f, n, "d", ("nVertices",), getattr(mesh, n))
    _put(f, "nEdgesOnCell", "i", ("nCells",), mesh.nEdgesOnCell)
    _put(f, "nEdgesOnEdge", "i", ("nEdges",), mesh.nEdgesOnEdge)
    for n in ("edgesOnCell", "verticesOnCell", "cellsOnCell"):
        _put(f, n, "i", ("nCells", "maxEdges"), getattr(mesh, n))
    for n in ("cellsOnEdge", "verticesOnEdge"):
        _put(f, n, "i", ("nEdges", "TWO"), getattr(mesh, n))
    _put(f, "edgesOnEdge", "i", ("nEdges", "maxEdges2"), mesh.edgesOnEdge)
    _put(f, "weightsOnEdge", "d", ("nEdges", "maxEdges2"), mesh.weightsOnEdge)
    for n in ("edgesOnVertex", "cellsOnVertex"):
        _put(f, n, "i", ("nVertices", "vertexDegree"), getattr(mesh, n))
    if mesh.kiteAreasOnVertex is not None:
        _put(f, "kiteAreasOnVertex", "d", ("nVertices", "vertexDegree"), mesh.kiteAreasOnVertex)
    if restingThickness is not None:
        rest = np.asarray(restingThickness, dtype=np.float64).reshape(mesh.nCells, -1)
        K = rest.shape[1]
        f.createDimension("nVertLevels", K)
        _put(f, "restingThickness", "d", ("Time", "nCells", "nVertLevels"), rest[None])
        _put(f, "minLevelCell", "i", ("nCells",), np.ones(mesh.nCells, I32))
        _put(f, "maxLevelCell", "i", ("nCells",), np.full(mesh.nCells, K, I32))
        if state is not None:
            ssh, u, h = state
            _put(f, "ssh", "d", ("Time", "nCells"), np.asarray(ssh, dtype=np.float64).reshape(1, mesh.nCells))
            _put(f, "normalVelocity", "d", ("Time", "nEdges", "nVertLevels"),
                 np.asarray(u, dtype=np.float64).reshape(1, mesh.nEdges, K))
            _put(f, "layerThickness", "d", ("Time", "nCells", "nVertLevels"),
                 np.asarray(h, dtype=np.float64).reshape(1, mesh.nCells, K))
    f.close()


def write_output(path, mesh: MeshData, nVertLevels: int, dt_seconds: float, elapsed_seconds: float, ssh, layerThickness,
                 normalVelocity):
    """write_netcdf(Setup, Diag, Prog) (OutPut.jl:117-215): same dimensions, global attribute `dt`, coordinate, metric
    and connectivity variables and the three prognostic fields of the current time level.  The reference declares
    layerThickness("nCells","nVertLevels") in Julia's column-major order, i.e. (nVertLevels, nCells) on disk, and
    defines but never fills angleEdge / edgeSignOnCell / cellsOnEdge / verticesOnCell / verticesOnEdge (:200-209);
    here those are filled."""
    K = int(nVertLevels)
    f = _nc_create(path)
    for n, v in (("time", 1), ("nCells", mesh.nCells), ("nEdges", mesh.nEdges), ("nVertices", mesh.nVertices),
                 ("nVertLevels", K), ("maxEdges", mesh.maxEdges), ("TWO", 2)):
        f.createDimension(n, int(v))
    f.dt = float(dt_seconds)
    _put(f, "time", "d", ("time",), [float(elapsed_seconds)])
    for n, dim in (("xCell", "nCells"), ("yCell", "nCells"), ("xEdge", "nEdges"), ("yEdge", "nEdges"),
                   ("xVertex", "nVertices"), ("yVertex", "nVertices"), ("dcEdge", "nEdges"), ("areaCell", "nCells"),
                   ("angleEdge", "nEdges"), ("areaTriangle", "nVertices")):
        _put(f, n, "d", (dim,), getattr(mesh, n))
    _put(f, "edgeSignOnCell", "i", ("nCells", "maxEdges"), mesh.edgeSignOnCell)
    _put(f, "nEdgesOnCell", "i", ("nCells",), mesh.nEdgesOnCell)
    _put(f, "nEdgesOnEdge", "i", ("nEdges",), mesh.nEdgesOnEdge)
    _put(f, "cellsOnEdge", "i", ("nEdges", "TWO"), mesh.cellsOnEdge)
    _put(f, "verticesOnCell", "i", ("nCells", "maxEdges"), mesh.verticesOnCell)
    _put(f, "verticesOnEdge", "i", ("nEdges", "TWO"), mesh.verticesOnEdge)
    _put(f, "ssh", "d", ("nCells",), np.asarray(ssh, dtype=np.float64).reshape(mesh.nCells))
    _put(f, "layerThickness", "d", ("nVertLevels", "nCells"),
         np.asarray(layerThickness, dtype=np.float64).reshape(mesh.nCells, K).T)
    _put(f, "normalVelocity", "d", ("nVertLevels", "nEdges"),
         np.asarray(normalVelocity, dtype=np.float64).reshape(mesh.nEdges, K).T)
    f.close()


# ------------------------------------------------------------------------------------------------
# minimal HDF5 writer (root-group datasets + string attributes): lets tests exercise the NetCDF-4 reader path
# ------------------------------------------------------------------------------------------------
def write_hdf5(path, arrays: dict, attrs: dict | None = None):
    L = hdf5()
    if L is None:
        raise MpasIOError("no libhdf5 on this machine")
    f = L.H5Fcreate(os.fsencode(path), 2, 0, 0)        # H5F_ACC_TRUNC
    if f < 0:
        raise MpasIOError(f"cannot create {path}")
    try:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype not in (np.float64, np.float32, np.int32, np.int64):
                a = a.astype(np.float64 if a.dtype.kind == "f" else np.int32)
            dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
            sp = L.H5Screate_simple(a.ndim, dims, None)
            tp = _h5_native(L, a.dtype)
            d = L.H5Dcreate2(f, name.encode(), tp, sp, 0, 0, 0)
            if d < 0 or (a.size and L.H5Dwrite(d, tp, 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0):
                raise MpasIOError(f"writing {name} failed")
            L.H5Dclose(d); L.H5Sclose(sp)
        for name, v in (attrs or {}).items():
            sp = L.H5Screate(0)                        # H5S_SCALAR
            if isinstance(v, str):
                b = v.encode()
                tp = L.H5Tcopy(C.c_int64.in_dll(L, "H5T_C_S1_g").value)
                L.H5Tset_size(tp, max(len(b), 1))
                a = L.H5Acreate2(f, name.encode(), tp, sp, 0, 0)
                L.H5Awrite(a, tp, C.create_string_buffer(b, max(len(b), 1)))
                L.H5Tclose(tp)
            else:
                tp = _h5_native(L, np.float64)
                a = L.H5Acreate2(f, name.encode(), tp, sp, 0, 0)
                L.H5Awrite(a, tp, C.byref(C.c_double(float(v))))
            L.H5Aclose(a); L.H5Sclose(sp)
    finally:
        L.H5Fclose(f)
