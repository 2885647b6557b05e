"""Python transliteration of julia/MokaHIP.jl -- the logic a MOKA.jl maintainer adds so that the reference's OWN
constructors and driver (src/forward/init.jl:3-30, src/ocn/PrognosticVars.jl:59-106, DiagnosticVars.jl:75-99,
TendencyVars.jl:51-67, src/driver/mpas_ocean.jl:20-53) run on libmoka_hip unchanged.

Julia cannot run in this pipeline; this module is how that logic gets executed: same lazy arrays, same binding at the
first device call, same version-stamped host copies, same dispatch points, same order of library calls.  The names
follow the shim (MArray, adapt, zeros, ones, state_of, ...) and the reference (`RefPrognosticVars` etc. restate the
reference's struct constructors: what they check and copy, not what they compute).

    backend   = shim.Backend()                                  # mpas_ocean.jl:28  (was CUDABackend())
    Setup, Diag, Tend, Prog = shim.ocn_init(config_fp, backend) # init.jl:3-30, constructor by constructor
    clock, simAlarm, outAlarm = shim.ocn_init_alarms(Setup)     # init.jl:111-127
    timestep = shim.zeros(backend, np.float64, (1,)); timestep[0] = dt      # mpas_ocean.jl:36-37
    shim.ocn_run_loop(timestep, Prog, Diag, Tend, Setup, ForwardEuler, clock, simAlarm, outAlarm, backend=backend)
    shim.write_netcdf(Setup, Diag, Prog)                        # OutPut.jl:117-215 through adapt_structure(CPU(), x)
"""
from __future__ import annotations

import copy
import ctypes as C
import math
import weakref

import numpy as np

from . import api
from . import lib as L
from . import mpasio
from .config import ConfigGet, ConfigRead
from .timemanager import Second, advance, changeTimeStep, isRinging, period_seconds, reset

ForwardEuler, RungeKutta4 = api.ForwardEuler, api.RungeKutta4
REFERENCE_COMPAT = L.FE_REFERENCE_COMPAT


class CPU:                      # KA.CPU()
    pass


class Backend(api.MokaHIP):
    """`mutable struct Backend <: KA.GPU`: the context, reference-counted by what lives on it (Julia finalizers run in no
    particular order; in Python the children simply hold a reference)."""
    is_gpu = True               # typeof(backend) <: KA.GPU   (mpas_ocean.jl:49)


# ---- arrays "on the backend" ----------------------------------------------------------------------
class State:
    """One moka_state behind Prog / Diag / Tend of a model (created at the first device call)."""

    def __init__(self, handle, mesh, backend):
        self.handle, self.mesh, self.backend = handle, mesh, backend
        self.version = 0                     # bumped by every call that changes device fields
        self.bound = []                      # weak references to the MArrays bound to this state
        api._own(self, L.lib().moka_state_destroy, handle, mesh, backend)

    def close(self):
        self._fin()


class MArray:
    """Host array + optional binding to a field of a device state (`mutable struct MArray{T,N} <: AbstractArray{T,N}`).
    Stored in the reference's layout: a Julia (K, n) array is the C-ordered numpy (n, K) array."""

    def __init__(self, host: np.ndarray, backend: Backend):
        self.host, self.backend = host, backend
        self.state, self.field, self.level = None, -1, 0
        self.host_version, self.host_dirty = -1, False

    @property
    def shape(self):
        return self.host.shape

    @property
    def dtype(self):
        return self.host.dtype

    def __deepcopy__(self, memo):            # Base.deepcopy_internal: a still unbound array copies its host data
        return MArray(np.array(_array(self), copy=True), self.backend)

    def __getitem__(self, i):                # getindex: one download per device change, not one per element
        return sync_host(self).host[i]

    def __setitem__(self, i, v):             # setindex!  (`@allowscalar timestep[1] = dt`, mpas_ocean.jl:37)
        sync_host(self).host[i] = v
        if self.state is not None:
            self.host_dirty = True

    def __array__(self, dtype=None, copy=None):
        return _array(self)

    def __len__(self):
        return len(self.host)


def sync_host(a: MArray) -> MArray:
    s = a.state
    if s is None or a.host_dirty or a.host_version == s.version:
        return a
    L.check(L.lib().moka_state_download(s.handle, a.field, a.level, L.f64(a.host)), s.backend._h)
    a.host_version = s.version
    return a


def _array(a: MArray) -> np.ndarray:         # Base.Array(a)
    return np.array(sync_host(a).host, copy=True)


def get_backend(a):                          # KA.get_backend
    return a.backend if isinstance(a, MArray) else CPU()


def adapt(to, a):
    """Adapt.adapt(to, a): Adapt.adapt_storage(::Backend, ::Array) -> MArray; (::KA.CPU, ::MArray) -> Array; structs go
    through their adapt_structure (below)."""
    if isinstance(a, (RefPrognosticVars, RefDiagnosticVars, RefTendencyVars, RefMesh)):
        return a.adapt_structure(to)
    if isinstance(to, Backend):
        return a if isinstance(a, MArray) else MArray(np.array(a, copy=True), to)
    if isinstance(to, CPU):
        return _array(a) if isinstance(a, MArray) else a
    raise TypeError("adapt: unknown target")


def zeros(backend: Backend, dtype, dims) -> MArray:          # KA.zeros(backend, T, dims...)
    return MArray(np.zeros(dims, dtype=dtype), backend)


def ones(backend: Backend, dtype, dims) -> MArray:           # KA.ones (VertMesh.jl:32-33)
    return MArray(np.ones(dims, dtype=dtype), backend)


# ---- the reference's structs: what their constructors check and copy --------------------------------
def _check_args(args):
    """Architectures.jl:19-46: same type name, same backend, same eltype."""
    if len({type(a).__name__ for a in args}) != 1:
        raise TypeError("Input arguments must be of all the same type")
    if len({id(get_backend(a)) if isinstance(a, MArray) else "cpu" for a in args}) != 1:
        raise TypeError("All input arguments must have the same backend")
    if len({np.asarray(a.host if isinstance(a, MArray) else a).dtype for a in args}) != 1:
        raise TypeError("All input arguments must have the same eltype")


class RefPrognosticVars:
    """PrognosticVars(ssh, normalVelocity, layerThickness, nTimeLevels) (PrognosticVars.jl:30-56): vectors of
    nTimeLevels deep copies."""

    def __init__(self, ssh, normalVelocity, layerThickness, nTimeLevels):
        _check_args((ssh, normalVelocity, layerThickness))
        self.ssh = [copy.deepcopy(ssh) for _ in range(nTimeLevels)]
        self.normalVelocity = [copy.deepcopy(normalVelocity) for _ in range(nTimeLevels)]
        self.layerThickness = [copy.deepcopy(layerThickness) for _ in range(nTimeLevels)]

    def adapt_structure(self, to):           # PrognosticVars.jl:108-113
        return RefPrognosticVars(adapt(to, self.ssh[0]), adapt(to, self.normalVelocity[0]), adapt(to, self.layerThickness[0]),
                                 len(self.ssh))


class RefDiagnosticVars:
    def __init__(self, layerThicknessEdge, thicknessFlux, velocityDivCell, relativeVorticity):   # DiagnosticVars.jl:52-72
        _check_args((layerThicknessEdge, thicknessFlux, velocityDivCell, relativeVorticity))
        self.layerThicknessEdge, self.thicknessFlux = layerThicknessEdge, thicknessFlux
        self.velocityDivCell, self.relativeVorticity = velocityDivCell, relativeVorticity

    def adapt_structure(self, to):           # DiagnosticVars.jl:101-106
        return RefDiagnosticVars(adapt(to, self.layerThicknessEdge), adapt(to, self.thicknessFlux),
                                 adapt(to, self.velocityDivCell), adapt(to, self.relativeVorticity))


class RefTendencyVars:
    def __init__(self, tendNormalVelocity, tendLayerThickness):          # TendencyVars.jl:33-48
        _check_args((tendNormalVelocity, tendLayerThickness))
        self.tendNormalVelocity, self.tendLayerThickness = tendNormalVelocity, tendLayerThickness


class RefMesh:
    """Mesh(HorzMesh, VertMesh) with every array adapted to the backend (ReadHorzMesh(...; backend) ends in
    Adapt.adapt_structure(backend, mesh), HorzMesh.jl:334-398; VerticalMesh(mesh_fp, h_mesh; backend), VertMesh.jl:46-82)."""

    ARRAYS = ("xCell", "yCell", "zCell", "areaCell", "nEdgesOnCell", "edgesOnCell", "edgeSignOnCell", "cellsOnEdge",
              "verticesOnEdge", "nEdgesOnEdge", "edgesOnEdge", "weightsOnEdge", "dvEdge", "dcEdge", "fEdge", "angleEdge",
              "edgesOnVertex", "cellsOnVertex", "edgeSignOnVertex", "areaTriangle", "kiteAreasOnVertex", "fVertex")

    def __init__(self, data, vert, arrays):
        self.data, self.vert, self.arrays = data, vert, arrays          # host MeshData (counts), VerticalMesh, {name: MArray}
        self.HorzMesh = self
        self.Edges = self.PrimaryCells = self.DualCells = self
        self.VertMesh = vert

    def __getattr__(self, name):             # mesh.HorzMesh.Edges.dcEdge -> the adapted array
        arrays = self.__dict__.get("arrays", {})
        if name in arrays:
            return arrays[name]
        return getattr(self.__dict__["data"], name)

    def adapt_structure(self, to):           # MPASMesh.jl:26 -> HorzMesh.jl:53-56, :357-398
        return RefMesh(self.data, self.vert, {k: adapt(to, v) for k, v in self.arrays.items()})


def ocn_setup_mesh(Config, backend: Backend) -> RefMesh:
    """ocn_setup_mesh(Config; backend) (init.jl:41-55)."""
    mesh_fp = ConfigGet(ConfigGet(Config.streams, "mesh"), "filename_template")
    data = mpasio.read_mesh(mesh_fp)                                    # ReadHorzMesh: host read + signIndexField! ...
    arrays = {n: adapt(backend, getattr(data, n)) for n in RefMesh.ARRAYS if getattr(data, n, None) is not None}   # ... then adapt
    vm = mpasio.read_vertical_mesh(mesh_fp, data)                       # VerticalMesh(mesh_fp, h_mesh; backend)
    vert = api.VerticalMesh(api.HorzMesh(data), nVertLevels=vm["nVertLevels"], restingThickness=vm["restingThickness"])
    vert.maxLevelEdge.Top = ones(backend, np.int32, (data.nEdges,))     # ActiveLevels: KA.ones(backend, Int32, nEdges), VertMesh.jl:31-44
    vert.maxLevelEdge.Bot = ones(backend, np.int32, (data.nEdges,))
    vert.restingThicknessSum = adapt(backend, vert.restingThicknessSum)
    return RefMesh(data, vert, arrays)


# ---- binding -----------------------------------------------------------------------------------------
# const MESHES (MokaHIP.jl): by identity of the Mesh's arrays + backend, weakly.  Python objects hash by identity, so a
# WeakKeyDictionary keyed by the RefMesh object is that (one device mesh per RefMesh object; a RefMesh names its backend)
_MESHES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _host(a):
    return a.host if isinstance(a, MArray) else a


def device_mesh(m: RefMesh, b: Backend) -> api.Mesh:
    """moka_mesh_create from the arrays the reference's Mesh holds."""
    if m in _MESHES:
        return _MESHES[m]
    import types
    d = types.SimpleNamespace(**{k: getattr(m.data, k) for k in ("nCells", "nEdges", "nVertices", "maxEdges", "maxEdges2", "vertexDegree")})
    for k, v in m.arrays.items():
        setattr(d, k, _host(v))
    vert = m.vert
    dm = api.Mesh.__new__(api.Mesh)
    dm.HorzMesh, dm.VertMesh, dm.backend, dm.state_bytes = api.HorzMesh(d), vert, b, 8
    dm._h = C.c_void_p()
    desc, keep = L.make_desc(d, vert.nVertLevels, np.ascontiguousarray(_host(vert.restingThicknessSum)).reshape(-1),
                             np.ascontiguousarray(_host(vert.maxLevelEdge.Top)), L.ORDER_DEFAULT, 0)
    L.check(L.lib().moka_mesh_create(b._h, C.byref(desc), C.byref(dm._h)), b._h)
    api._own(dm, L.lib().moka_mesh_destroy, dm._h, b)
    _MESHES[m] = dm
    return dm


def _bind(a: MArray, s: State, field: int, level: int, upload: bool):
    if a.state is s:
        return
    if a.state is not None:
        raise L.MokaError(L.ERR_ARG, "MokaHIP: array is already bound to another model state")
    if upload:
        L.check(L.lib().moka_state_upload(s.handle, field, level, L.f64(np.ascontiguousarray(a.host, dtype=np.float64))), s.backend._h)
    a.state, a.field, a.level = s, field, level
    a.host_version, a.host_dirty = (s.version if upload else -1), False
    s.bound.append(weakref.ref(a))


def flush_host_writes(s: State):
    for w in s.bound:
        a = w()
        if a is None or not a.host_dirty:
            continue
        L.check(L.lib().moka_state_upload(s.handle, a.field, a.level, L.f64(a.host)), s.backend._h)
        a.host_dirty, a.host_version = False, s.version


PLACEMENT_TRIES = 24        # const PLACEMENT_TRIES = Ref{Cint}(24)


def state_of(Prog: RefPrognosticVars, Diag, Tend, mesh: RefMesh, b: Backend) -> State:
    s = Prog.ssh[-1].state
    if s is None:
        dm = device_mesh(mesh, b)
        h = C.c_void_p()
        L.check(L.lib().moka_state_create(b._h, dm._h, C.byref(h)), b._h)
        s = State(h, dm, b)
        if len(Prog.ssh) != 2:
            raise L.MokaError(L.ERR_ARG, "nTimeLevels must be <= 2")      # time_integration.jl:23
        for t in range(2):                                               # Julia index 1 = previous = level 0, end = current = 1
            _bind(Prog.ssh[t], s, L.F_SSH, t, True)
            _bind(Prog.normalVelocity[t], s, L.F_NORMAL_VELOCITY, t, True)
            _bind(Prog.layerThickness[t], s, L.F_LAYER_THICKNESS, t, True)
        # the library's own per-array placement search, at binding (MokaHIP.jl state_of: PLACEMENT_TRIES)
        if PLACEMENT_TRIES > 1:
            L.lib().moka_state_optimize_placement(h, PLACEMENT_TRIES, None, None)     # an optimisation: its failure is not the model's
    if Diag is not None and Diag.layerThicknessEdge.state is None:
        for a, f in ((Diag.layerThicknessEdge, L.F_LAYER_THICKNESS_EDGE), (Diag.thicknessFlux, L.F_THICKNESS_FLUX),
                     (Diag.velocityDivCell, L.F_VELOCITY_DIV_CELL), (Diag.relativeVorticity, L.F_RELATIVE_VORTICITY)):
            _bind(a, s, f, 1, True)
    if Tend is not None and Tend.tendNormalVelocity.state is None:
        _bind(Tend.tendNormalVelocity, s, L.F_TEND_NORMAL_VELOCITY, 1, True)
        _bind(Tend.tendLayerThickness, s, L.F_TEND_LAYER_THICKNESS, 1, True)
    flush_host_writes(s)
    return s


# ---- the reference's init path, constructor by constructor (init.jl:3-30) ---------------------------
def ocn_init(Config_filepath, backend: Backend):
    Config = ConfigRead(Config_filepath)                                 # init.jl:6
    Mesh = ocn_setup_mesh(Config, backend)                               # :11
    Clock = api.ocn_setup_clock(Config)                                  # :13
    Setup = api.ModelSetup(Config, Mesh, Clock)                          # :16
    # PrognosticVars(Config, Mesh; backend)   PrognosticVars.jl:59-106
    if ConfigGet(ConfigGet(Config.namelist, "time_management"), "config_do_restart"):
        raise L.MokaError(L.ERR_UNSUPPORTED, "restart not yet supported")
    input_fp = ConfigGet(ConfigGet(Config.streams, "input"), "filename_template")
    nT = ConfigGet(ConfigGet(Config.namelist, "time_integration"), "config_number_of_time_levels")
    K = Mesh.vert.nVertLevels
    ssh, u, h = mpasio.read_initial_state(input_fp, Mesh.data, K)        # host zeros(...) filled from the file (:91-99)
    Prog = RefPrognosticVars(adapt(backend, ssh), adapt(backend, u), adapt(backend, h), nT)   # :101-104
    # DiagnosticVars(Config, Mesh; backend)   DiagnosticVars.jl:75-99: KA.zeros(backend, Float64, nVertLevels, n)
    nE, nC, nV = Mesh.data.nEdges, Mesh.data.nCells, Mesh.data.nVertices
    thicknessFlux = zeros(backend, np.float64, (nE, K))
    velocityDivCell = zeros(backend, np.float64, (nC, K))
    relativeVorticity = zeros(backend, np.float64, (nV, K))
    layerThicknessEdge = zeros(backend, np.float64, (nE, K))
    Diag = RefDiagnosticVars(layerThicknessEdge, thicknessFlux, velocityDivCell, relativeVorticity)
    # TendencyVars(Config, Mesh; backend)     TendencyVars.jl:51-67: one host zeros adapted, one KA.zeros
    tendNormalVelocity = np.zeros((nE, K))
    tendLayerThickness = zeros(backend, np.float64, (nC, K))
    Tend = RefTendencyVars(adapt(backend, tendNormalVelocity), tendLayerThickness)
    return Setup, Diag, Tend, Prog


def ocn_init_alarms(Setup):
    """ocn_init_alarms (init.jl:111-127): mean(dcEdge) iterates the adapted array (scalar reads of an unbound MArray)."""
    dcEdge = Setup.mesh.HorzMesh.Edges.dcEdge
    mean = float(np.mean(np.asarray(dcEdge)))
    dt = math.floor(2 * (mean / 1e3) * mean / 200e3)
    changeTimeStep(Setup.timeManager, Second(dt))
    clock = Setup.timeManager
    return clock, clock.alarms["simulation_end"], clock.alarms["outputAlarm"]


# ---- forward model: the dispatch points of the shim ---------------------------------------------------
def ocn_timestep(*args, backend: Backend):
    """MOKA.ocn_timestep(timestep, Prog::MProg, Diag::MDiag, Tend::MTend, S, ForwardEuler; backend) / (Prog, ..., RungeKutta4)."""
    if len(args) == 6:
        timestep, Prog, Diag, Tend, S, method = args
    else:
        Prog, Diag, Tend, S, method = args
        timestep = None
    s = state_of(Prog, Diag, Tend, S.mesh, backend)
    if method is ForwardEuler:
        dt = float(timestep[0])              # the reference's 1-element array on the backend: never bound
        L.check(L.lib().moka_step_fe(s.handle, dt, REFERENCE_COMPAT), backend._h)
    else:
        dt = float(period_seconds(S.timeManager.timeStep))
        L.check(L.lib().moka_step_rk4(s.handle, dt), backend._h)
    s.version += 1                           # device_changed!


def diagnostic_compute(Mesh: RefMesh, Diag, Prog, backend: Backend):
    s = state_of(Prog, Diag, None, Mesh, backend)
    L.check(L.lib().moka_diagnostic_compute(s.handle, REFERENCE_COMPAT), backend._h)
    s.version += 1


def ocn_run_loop(*args, backend: Backend):
    """run_loop.jl:8-22 verbatim (it needs no method of its own in the shim) and the (sumCPU, sumGPU, ...) form :26-45."""
    want_sum = len(args) == 11
    if want_sum:
        sumCPU, sumGPU, *args = args
    timestep, Prog, Diag, Tend, Setup, method, clock, simulationAlarm, outputAlarm = args
    while not isRinging(simulationAlarm):
        advance(clock)
        ocn_timestep(timestep, Prog, Diag, Tend, Setup, method, backend=backend)
        if isRinging(outputAlarm):
            reset(outputAlarm)
    if want_sum:
        s = state_of(Prog, Diag, Tend, Setup.mesh, backend)
        out = C.c_double()
        L.check(L.lib().moka_sum_sq(s.handle, L.F_SSH, 1, C.byref(out)), backend._h)
        sumGPU[0] = sumGPU[0] + out.value
        sumCPU[0] = sumGPU[0]                # mycopyto!(sumCPU, sumGPU)
        return sumCPU[0]
    return None


def write_netcdf(Setup, Diag, Prog):
    """write_netcdf(Setup, Diag, Prog) (OutPut.jl:117-215): everything comes back through Adapt.adapt_structure(KA.CPU(), x)."""
    Mesh = adapt(CPU(), Setup.mesh)                                      # OutPut.jl:122
    Diag = adapt(CPU(), Diag)                                            # :123
    Prog = adapt(CPU(), Prog)                                            # :124
    assert all(isinstance(a, np.ndarray) for a in (Prog.ssh[-1], Diag.thicknessFlux, Mesh.arrays["dcEdge"]))
    out_fp = ConfigGet(ConfigGet(Setup.config.streams, "output"), "filename_template")
    clock = Setup.timeManager
    mpasio.write_output(out_fp, Setup.mesh.data, Setup.mesh.vert.nVertLevels, period_seconds(clock.timeStep),
                        (clock.currTime - clock.startTime).total_seconds(), Prog.ssh[-1], Prog.layerThickness[-1],
                        Prog.normalVelocity[-1])
    return out_fp


def ocn_run(config_fp, device: int = 0):
    """src/driver/mpas_ocean.jl:20-53, line by line; only :28 differs."""
    backend = Backend(device)                                            # :28   backend = CUDABackend()
    Setup, Diag, Tend, Prog = ocn_init(config_fp, backend)               # :31
    clock, simulationAlarm, outputAlarm = ocn_init_alarms(Setup)         # :33
    timestep = zeros(backend, np.float64, (1,))                          # :36   KA.zeros(backend, Float64, (1,))
    timestep[0] = float(period_seconds(Setup.timeManager.timeStep))      # :37   @allowscalar timestep[1] = ...
    ocn_run_loop(timestep, Prog, Diag, Tend, Setup, ForwardEuler, clock, simulationAlarm, outputAlarm, backend=backend)   # :39
    out = write_netcdf(Setup, Diag, Prog)                                # :46
    backend2 = get_backend(Tend.tendNormalVelocity)                      # :48
    arch = "GPU" if getattr(backend2, "is_gpu", False) else "CPU"        # :49
    return out, arch, clock, (Setup, Diag, Tend, Prog)
