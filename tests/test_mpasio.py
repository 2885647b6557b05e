"""MPAS NetCDF readers / writer: round trips through both on-disk formats the
reference's NCDataset opens -- NetCDF classic (scipy) and NetCDF-4 = HDF5 (ctypes on libhdf5) -- and the reader
semantics of HorzMesh.jl:166-290, VertMesh.jl:46-82, PrognosticVars.jl:85-99, OutPut.jl:117-215."""
import dataclasses

import numpy as np
import pytest

import oracle as orc
from moka_hip import lib as L
from moka_hip import meshgen as mg
from moka_hip import mpasio as io

MESH_VARS = ["xCell", "yCell", "zCell", "fCell", "areaCell", "nEdgesOnCell", "edgesOnCell", "verticesOnCell",
             "cellsOnCell", "xEdge", "yEdge", "zEdge", "fEdge", "dvEdge", "dcEdge", "angleEdge", "nEdgesOnEdge",
             "cellsOnEdge", "verticesOnEdge", "edgesOnEdge", "weightsOnEdge", "xVertex", "yVertex", "zVertex",
             "fVertex", "areaTriangle", "edgesOnVertex", "cellsOnVertex", "kiteAreasOnVertex"]


def same_mesh(a, b):
    for f in dataclasses.fields(a):
        if f.name == "meta":
            continue
        x, y = getattr(a, f.name), getattr(b, f.name)
        if isinstance(x, np.ndarray):
            assert x.shape == y.shape and np.array_equal(x, y), f.name
        else:
            assert x == y, f.name


@pytest.fixture(scope="module")
def case():
    mesh = mg.igw_mesh(200.0)
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    return mesh, ssh, u.reshape(mesh.nEdges, 1), h.reshape(mesh.nCells, 1), np.asarray(rest).reshape(mesh.nCells, 1)


def test_netcdf3_round_trip_mesh_vertical_state(tmp_path, case):
    mesh, ssh, u, h, rest = case
    p = tmp_path / "igw.nc"
    io.write_mesh(p, mesh, restingThickness=rest, state=(ssh, u, h))
    got = io.read_mesh(p)
    same_mesh(mesh, got)                              # includes both signIndexField! results
    vm = io.read_vertical_mesh(p, got)
    assert vm["nVertLevels"] == 1 and vm["stacked"] and np.array_equal(vm["restingThickness"], rest)
    assert np.array_equal(vm["restingThicknessSum"], rest.sum(1))            # sum(dims=1), VertMesh.jl:73
    s2, u2, h2 = io.read_initial_state(p, got, 1)
    assert np.array_equal(s2, ssh) and np.array_equal(u2, u) and np.array_equal(h2, h)
    # the file drives the oracle to the same bits as the generated mesh
    a = orc.OracleState(orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.sum(1)), ssh, u, h)
    b = orc.OracleState(orc.OracleMesh(got, 1, resting_thickness_sum=vm["restingThicknessSum"]), s2, u2, h2)
    for _ in range(3):
        a.step_fe(400.0)
        b.step_fe(400.0)
    assert np.array_equal(a.ssh[1], b.ssh[1]) and np.array_equal(a.u[1], b.u[1])
    # and the host plan built from it is the same plan
    assert np.array_equal(L.Plan(mesh, 1).permutation(L.EDGE), L.Plan(got, 1).permutation(L.EDGE))


def test_missing_coriolis_defaults_to_zero_and_missing_variable_raises(tmp_path, case):
    mesh = case[0]
    if io.hdf5() is None:
        pytest.skip("no libhdf5 on this machine")
    arrays = {n: getattr(mesh, n) for n in MESH_VARS if n not in ("fCell", "fEdge", "fVertex", "kiteAreasOnVertex")}
    p = tmp_path / "nof.h5"
    io.write_hdf5(p, arrays, {"is_periodic": "YES", "on_a_sphere": "NO"})
    got = io.read_mesh(p)
    assert not got.fEdge.any() and not got.fCell.any() and not got.fVertex.any()    # HorzMesh.jl:177-182,220-225,257-262
    assert got.kiteAreasOnVertex is None                                            # optional: the reference never reads it
    del arrays["weightsOnEdge"]
    io.write_hdf5(p, arrays, {"is_periodic": "YES"})
    with pytest.raises(io.MpasIOError):
        io.read_mesh(p)


def test_netcdf4_hdf5_reader_path(tmp_path):
    if io.hdf5() is None:
        pytest.skip("no libhdf5 on this machine")
    mesh = mg.icosahedral_mesh(4)
    K = 3
    ssh, u, h, rest, _ = mg.sphere_synthetic_state(mesh, K)
    arrays = {n: getattr(mesh, n) for n in MESH_VARS}
    arrays.update(restingThickness=rest[None], minLevelCell=np.ones(mesh.nCells, np.int32),
                  maxLevelCell=np.full(mesh.nCells, K, np.int32), ssh=ssh[None], normalVelocity=u[None],
                  layerThickness=h[None])
    p = tmp_path / "sphere.nc4"
    io.write_hdf5(p, arrays, {"is_periodic": "YES", "on_a_sphere": "YES", "sphere_radius": mesh.sphere_radius})
    got = io.read_mesh(p)
    same_mesh(mesh, got)
    vm = io.read_vertical_mesh(p, got)
    assert vm["nVertLevels"] == K and np.array_equal(vm["restingThickness"], rest)
    s2, u2, h2 = io.read_initial_state(p, got, K)
    assert np.array_equal(s2, ssh) and np.array_equal(u2, u) and np.array_equal(h2, h)
    # non-periodic meshes are refused by the vertical-mesh constructor (VertMesh.jl:50-52)
    io.write_hdf5(p, arrays, {"is_periodic": "NO"})
    with pytest.raises(io.MpasIOError, match="non-periodic"):
        io.read_vertical_mesh(p, got)
    # a mesh that is not stacked is only reported (VertMesh.jl:60-65 logs, does not throw)
    arrays["maxLevelCell"] = np.full(mesh.nCells, K - 1, np.int32)
    io.write_hdf5(p, arrays, {"is_periodic": "YES"})
    assert io.read_vertical_mesh(p, got)["stacked"] is False


def test_write_output_layout(tmp_path, case):
    mesh, ssh, u, h, rest = case
    K = 2
    h2, u2 = np.repeat(h, K, axis=1) / K, np.repeat(u, K, axis=1)
    p = tmp_path / "out.nc"
    io.write_output(p, mesh, K, 400.0, 36000.0, ssh, h2, u2)
    ds = io.open_dataset(p)
    try:
        assert ds.attr("dt") == 400.0                                     # OutPut.jl:147-149
        assert ds.var("time")[0] == 36000.0                               # seconds since startTime, :186
        assert np.array_equal(ds.var("ssh"), ssh)
        # declared ("nCells","nVertLevels") in column-major Julia = (nVertLevels, nCells) on disk
        assert ds.var("layerThickness").shape == (K, mesh.nCells) and np.array_equal(ds.var("layerThickness").T, h2)
        assert ds.var("normalVelocity").shape == (K, mesh.nEdges) and np.array_equal(ds.var("normalVelocity").T, u2)
        for n in ("xCell", "yCell", "xEdge", "yEdge", "xVertex", "yVertex", "dcEdge", "areaCell", "areaTriangle",
                  "nEdgesOnCell", "nEdgesOnEdge", "edgeSignOnCell", "cellsOnEdge", "verticesOnCell", "verticesOnEdge"):
            assert np.array_equal(ds.var(n), getattr(mesh, n)), n
    finally:
        ds.close()


def test_unknown_file_format_is_refused(tmp_path):
    p = tmp_path / "junk.nc"
    p.write_bytes(b"not a netcdf file at all")
    with pytest.raises(io.MpasIOError):
        io.open_dataset(p)
    p.write_bytes(b"CDF\x05" + b"\0" * 32)
    with pytest.raises(io.MpasIOError, match="CDF-5"):
        io.open_dataset(p)
