"""ctypes binding of libmoka_hip.so (include/moka_hip.h).  Fails loudly: no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))     # mpas-ocean.jl_amd/
# MOKA_HIP_LIB: another build of the same library (A/B experiments: `make exp EXP=...` -> libmoka_hip_exp.so)
LIB_PATH = os.environ.get("MOKA_HIP_LIB") or os.path.join(PKG_DIR, "libmoka_hip.so")

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)

OK = 0
ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_ALLOC, ERR_UNSUPPORTED, ERR_COMM = -1, -2, -3, -4, -5, -6
ORDER_DEFAULT, ORDER_NONE, ORDER_RCM, ORDER_RCB = 0, 1, 2, 3
CELL, EDGE, VERTEX = 0, 1, 2
(F_SSH, F_NORMAL_VELOCITY, F_LAYER_THICKNESS, F_LAYER_THICKNESS_EDGE, F_THICKNESS_FLUX,
 F_VELOCITY_DIV_CELL, F_RELATIVE_VORTICITY, F_TEND_NORMAL_VELOCITY, F_TEND_LAYER_THICKNESS) = range(9)
FE_STALE_HEDGE, FE_ACCUM_VORT, FE_LEVEL1_ONLY, FE_REFERENCE_COMPAT = 1, 2, 4, 7
FORWARD_EULER, RUNGE_KUTTA_4 = 0, 1


class MokaError(RuntimeError):
    """Julia side: `error(msg)` raised by the shim when a call returns non-zero."""

    def __init__(self, code, msg):
        super().__init__(f"libmoka_hip error {code}: {msg}")
        self.code = code


class MeshDesc(C.Structure):
    _fields_ = [
        ("nCells", C.c_int32), ("nEdges", C.c_int32), ("nVertices", C.c_int32),
        ("maxEdges", C.c_int32), ("maxEdges2", C.c_int32), ("vertexDegree", C.c_int32),
        ("nVertLevels", C.c_int32), ("edgeSignOnVertexLD", C.c_int32),
        ("xCell", _f64p), ("yCell", _f64p), ("zCell", _f64p),
        ("nEdgesOnCell", _i32p), ("edgesOnCell", _i32p), ("edgeSignOnCell", _i32p), ("areaCell", _f64p),
        ("cellsOnEdge", _i32p), ("verticesOnEdge", _i32p), ("nEdgesOnEdge", _i32p), ("edgesOnEdge", _i32p),
        ("weightsOnEdge", _f64p), ("dvEdge", _f64p), ("dcEdge", _f64p), ("fEdge", _f64p),
        ("edgesOnVertex", _i32p), ("cellsOnVertex", _i32p), ("edgeSignOnVertex", _i32p), ("areaTriangle", _f64p),
        ("maxLevelEdgeTop", _i32p), ("restingThicknessSum", _f64p),
        ("ordering", C.c_int32), ("patch_cells", C.c_int32),
        ("cellClass", _i32p),
        ("stateBytes", C.c_int32),
        ("kiteAreasOnVertex", _f64p), ("fVertex", _f64p),
    ]


class HaloStats(C.Structure):          # moka_halo_stats
    _fields_ = [("steps", C.c_int64), ("exchanges", C.c_int64), ("boundary_launches", C.c_int64), ("interior_launches", C.c_int64),
                ("host_step_ms", C.c_double), ("host_signal_wait_ms", C.c_double), ("host_flag_store_ms", C.c_double),
                ("host_wait_ms", C.c_double), ("boundary_launch_ms", C.c_double), ("interior_launch_ms", C.c_double)]


class PlacementTrial(C.Structure):     # moka_placement_trial
    _fields_ = [("field", C.c_int32), ("ms_old", C.c_double), ("ms_new", C.c_double), ("kept", C.c_int32)]


class MeshInfo(C.Structure):
    _fields_ = [
        ("nCells", C.c_int32), ("nEdges", C.c_int32), ("nVertices", C.c_int32), ("nVertLevels", C.c_int32),
        ("ordering", C.c_int32), ("patch_cells", C.c_int32), ("nPatches", C.c_int32),
        ("maxEdgesUsed", C.c_int32), ("maxEdges2Used", C.c_int32), ("lanesPerColumn", C.c_int32),
        ("meshBytesDevice", C.c_int64), ("cellBandwidth", C.c_int64),
        ("maxPatchRows", C.c_int32), ("ldsBytesPerBlock", C.c_int32),
        ("maxPatchCells", C.c_int32), ("maxPatchEdges", C.c_int32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/moka_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "moka_version", "moka_ctx_create", "moka_ctx_destroy", "moka_last_error", "moka_sync",
    "moka_timer_start", "moka_timer_stop",
    "moka_plan_create", "moka_plan_destroy", "moka_plan_info", "moka_plan_permutation", "moka_plan_patch_ranges", "moka_plan_array",
    "moka_mesh_create", "moka_mesh_destroy", "moka_mesh_info_get",
    "moka_gradient_on_edge", "moka_divergence_on_cell", "moka_curl_on_vertex", "moka_interpolate_cell2edge",
    "moka_state_create", "moka_state_destroy", "moka_state_upload", "moka_state_download",
    "moka_advance_time_levels", "moka_diagnostic_compute", "moka_compute_normal_velocity_tendency",
    "moka_compute_layer_thickness_tendency", "moka_tendencies", "moka_step_fe", "moka_step_rk4", "moka_run",
    "moka_sum_sq", "moka_set_kernel_variant", "moka_kernel_variant_available", "moka_stage_timing", "moka_stage_timing_read",
    "moka_ctx_streams", "moka_halo_create", "moka_halo_destroy", "moka_halo_buffer_elems", "moka_halo_pack",
    "moka_halo_unpack", "moka_rk4_dist_begin", "moka_rk4_dist_stage", "moka_rk4_dist_end",
    "moka_plan_class_ranges", "moka_mesh_class_ranges", "moka_mesh_permutation", "moka_halo_direct_available", "moka_halo_set_overlap", "moka_halo_pack_fields", "moka_halo_unpack_fields",
    "moka_tape_record_rk4", "moka_tape_commit_rk4", "moka_adjoint_rk4_stage_fields", "moka_adjoint_rk4_stage",
    "moka_tape_record_fe", "moka_tape_commit_fe", "moka_adjoint_fe_step_fields", "moka_adjoint_fe_step",
    "moka_halo_export", "moka_halo_connect", "moka_halo_push_begin", "moka_halo_push_signal", "moka_halo_push_wait",
    "moka_rk4_dist_stage_launch", "moka_rk4_dist_step", "moka_fe_dist_launch", "moka_fe_dist_end", "moka_fe_dist_step",
    "moka_set_nonlinear", "moka_last_fe_path", "moka_set_viscosity_del2", "moka_tape_create", "moka_tape_destroy", "moka_step_fe_taped", "moka_step_rk4_taped", "moka_adjoint_seed_sum_sq_ssh", "moka_adjoint_sweep",
    "moka_adjoint_download",
    "moka_mark", "moka_marks_reset", "moka_marks_read", "moka_bw_probe", "moka_bw_probe_streams", "moka_bw_probe_reread", "moka_bw_probe_gather_big", "moka_ctx_pci_bus_id", "moka_halo_set_acquire", "moka_set_tuning", "moka_get_tuning", "moka_rk4_dist_parts_available", "moka_adjoint_rk4_stage_part", "moka_adjoint_rk4_parts_available", "moka_adjoint_rk4_stage_out_fields",
    "moka_gradient_on_edge_vjp", "moka_gradient_on_edge_jvp", "moka_divergence_on_cell_vjp", "moka_divergence_on_cell_jvp",
    "moka_curl_on_vertex_vjp", "moka_curl_on_vertex_jvp", "moka_fe_lazy_pending",
    "moka_state_optimize_placement", "moka_state_placement_log", "moka_state_download_rows",
    "moka_halo_stats_enable", "moka_halo_stats_read", "moka_halo_set_stream_flags", "moka_state_array_address", "moka_state_placement_launches",
    "moka_state_rk4_streams",
]


def build(force: bool = False) -> str:
    """Compile libmoka_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", PKG_DIR, "--no-print-directory", "clean"])
    subprocess.check_call(["make", "-C", PKG_DIR, "--no-print-directory", "-j4"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MokaError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: build it with `make -C {PKG_DIR}` "
                                       "(there is no CPU fallback for the HIP path)")
    try:
        # PyTorch bundles its own libamdhip64; if it is going to be used in this process (halo buffers, torch.distributed)
        # it has to be loaded first, or two HIP runtimes end up side by side and torch finds no device.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.moka_version.restype = C.c_char_p
    L.moka_last_error.restype = C.c_char_p
    L.moka_last_error.argtypes = [vp]
    L.moka_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.moka_ctx_destroy.argtypes = [vp]
    L.moka_ctx_destroy.restype = None
    L.moka_sync.argtypes = [vp]
    L.moka_timer_start.argtypes = [vp]
    L.moka_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.moka_plan_create.argtypes = [C.POINTER(MeshDesc), C.POINTER(vp)]
    L.moka_plan_destroy.argtypes = [vp]
    L.moka_plan_destroy.restype = None
    L.moka_plan_info.argtypes = [vp, C.POINTER(MeshInfo)]
    L.moka_plan_permutation.argtypes = [vp, C.c_int, _i32p]
    L.moka_plan_patch_ranges.argtypes = [vp, _i32p, _i32p, _i32p]
    L.moka_plan_array.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.moka_mesh_create.argtypes = [vp, C.POINTER(MeshDesc), C.POINTER(vp)]
    L.moka_mesh_destroy.argtypes = [vp]
    L.moka_mesh_destroy.restype = None
    L.moka_mesh_info_get.argtypes = [vp, C.POINTER(MeshInfo)]
    L.moka_gradient_on_edge.argtypes = [vp, _f64p, _f64p]
    L.moka_divergence_on_cell.argtypes = [vp, _f64p, _f64p, _f64p]
    L.moka_curl_on_vertex.argtypes = [vp, _f64p, _f64p]
    L.moka_interpolate_cell2edge.argtypes = [vp, _f64p, _f64p, C.c_int]
    L.moka_gradient_on_edge_vjp.argtypes = [vp, _f64p, _f64p]
    L.moka_gradient_on_edge_jvp.argtypes = [vp, _f64p, _f64p]
    L.moka_divergence_on_cell_vjp.argtypes = [vp, _f64p, _f64p, _f64p]
    L.moka_divergence_on_cell_jvp.argtypes = [vp, _f64p, _f64p, _f64p]
    L.moka_curl_on_vertex_vjp.argtypes = [vp, _f64p, _f64p]
    L.moka_curl_on_vertex_jvp.argtypes = [vp, _f64p, _f64p]
    L.moka_state_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.moka_state_destroy.argtypes = [vp]
    L.moka_state_destroy.restype = None
    L.moka_state_optimize_placement.argtypes = [vp, C.c_int, _f64p, _f64p]
    L.moka_state_placement_launches.argtypes = [vp]
    L.moka_state_rk4_streams.argtypes = [vp]
    L.moka_state_placement_launches.restype = C.c_int64
    L.moka_state_placement_log.argtypes = [vp, C.c_int32, C.POINTER(PlacementTrial), _i32p]
    L.moka_state_array_address.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.moka_state_download_rows.argtypes = [vp, C.c_int, C.c_int, C.c_int64, _i32p, _f64p]
    L.moka_state_upload.argtypes = [vp, C.c_int, C.c_int, _f64p]
    L.moka_state_download.argtypes = [vp, C.c_int, C.c_int, _f64p]
    L.moka_advance_time_levels.argtypes = [vp, C.c_int]
    L.moka_diagnostic_compute.argtypes = [vp, C.c_int]
    L.moka_compute_normal_velocity_tendency.argtypes = [vp, C.c_int]
    L.moka_compute_layer_thickness_tendency.argtypes = [vp, C.c_int]
    L.moka_tendencies.argtypes = [vp]
    L.moka_step_fe.argtypes = [vp, C.c_double, C.c_int]
    L.moka_step_rk4.argtypes = [vp, C.c_double]
    L.moka_run.argtypes = [vp, C.c_int, C.c_double, C.c_int64, C.c_int]
    L.moka_sum_sq.argtypes = [vp, C.c_int, C.c_int, _f64p]
    L.moka_set_kernel_variant.argtypes = [vp, C.c_int]
    L.moka_kernel_variant_available.argtypes = [C.c_int]
    L.moka_stage_timing.argtypes = [vp, C.c_int]
    L.moka_stage_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.moka_ctx_streams.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    i64p = C.POINTER(C.c_int64)
    L.moka_halo_create.argtypes = [vp, C.c_int32, _i32p, i64p, _i32p, i64p, _i32p, i64p, _i32p, i64p,
                                   C.c_int32, C.c_int32, C.POINTER(vp)]
    L.moka_halo_destroy.argtypes = [vp]
    L.moka_halo_set_stream_flags.argtypes = [vp, C.c_int]
    L.moka_halo_stats_enable.argtypes = [vp, C.c_int]
    L.moka_halo_stats_read.argtypes = [vp, C.POINTER(HaloStats)]
    L.moka_halo_destroy.restype = None
    L.moka_halo_buffer_elems.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.moka_halo_pack.argtypes = [vp, C.c_int, vp]
    L.moka_halo_unpack.argtypes = [vp, C.c_int, vp]
    L.moka_rk4_dist_begin.argtypes = [vp, C.c_double]
    L.moka_rk4_dist_stage.argtypes = [vp, C.c_int, C.c_int]
    L.moka_rk4_dist_parts_available.argtypes = [vp]
    L.moka_rk4_dist_end.argtypes = [vp]
    L.moka_plan_class_ranges.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), _i32p, _i32p, _i32p]
    L.moka_mesh_class_ranges.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), _i32p, _i32p, _i32p]
    L.moka_mesh_permutation.argtypes = [vp, C.c_int, _i32p]
    L.moka_halo_direct_available.argtypes = [vp]
    L.moka_halo_set_overlap.argtypes = [vp, C.c_int]
    L.moka_halo_set_acquire.argtypes = [vp, C.c_int]
    L.moka_halo_pack_fields.argtypes = [vp, vp, vp, vp, vp]
    L.moka_halo_unpack_fields.argtypes = [vp, vp, vp, vp, vp]
    L.moka_tape_record_rk4.argtypes = [vp, C.c_int, C.c_int]
    L.moka_tape_commit_rk4.argtypes = [vp, C.c_double]
    L.moka_adjoint_rk4_stage_fields.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.moka_adjoint_rk4_stage.argtypes = [vp, C.c_int]
    L.moka_adjoint_rk4_stage_part.argtypes = [vp, C.c_int, C.c_int]
    L.moka_adjoint_rk4_parts_available.argtypes = [vp]
    L.moka_adjoint_rk4_stage_out_fields.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.moka_tape_record_fe.argtypes = [vp, C.c_int, C.c_int]
    L.moka_tape_commit_fe.argtypes = [vp, C.c_double, C.c_int]
    L.moka_adjoint_fe_step_fields.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.moka_adjoint_fe_step.argtypes = [vp]
    L.moka_halo_export.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(HaloPeerInfo)]
    L.moka_halo_connect.argtypes = [vp, C.c_int32, C.POINTER(HaloPeerInfo), C.c_int32]
    L.moka_halo_push_begin.argtypes = [vp, C.c_int]
    L.moka_halo_push_signal.argtypes = [vp]
    L.moka_halo_push_wait.argtypes = [vp, C.c_double]
    L.moka_rk4_dist_stage_launch.argtypes = [vp, C.c_int]
    L.moka_rk4_dist_step.argtypes = [vp, C.c_double, TRANSPORT_FN, vp, vp, vp, C.c_double]
    L.moka_fe_dist_launch.argtypes = [vp, C.c_double, C.c_int, C.c_int]
    L.moka_fe_dist_end.argtypes = [vp]
    L.moka_fe_dist_step.argtypes = [vp, C.c_double, C.c_int, TRANSPORT_FN, vp, vp, vp, C.c_double]
    L.moka_set_nonlinear.argtypes = [vp, C.c_int]
    L.moka_last_fe_path.argtypes = [vp]
    L.moka_fe_lazy_pending.argtypes = [vp]
    L.moka_set_viscosity_del2.argtypes = [vp, C.c_double]
    L.moka_tape_create.argtypes = [vp, C.c_int64, C.POINTER(vp)]
    L.moka_tape_destroy.argtypes = [vp]
    L.moka_tape_destroy.restype = None
    L.moka_step_fe_taped.argtypes = [vp, C.c_double, C.c_int]
    L.moka_step_rk4_taped.argtypes = [vp, C.c_double]
    L.moka_adjoint_seed_sum_sq_ssh.argtypes = [vp]
    L.moka_adjoint_sweep.argtypes = [vp]
    L.moka_adjoint_download.argtypes = [vp, C.c_int, _f64p]
    L.moka_mark.argtypes = [vp]
    L.moka_marks_reset.argtypes = [vp]
    L.moka_marks_read.argtypes = [vp, C.c_int64, _f64p, C.POINTER(C.c_int64)]
    L.moka_bw_probe.argtypes = [vp, C.c_int64, C.c_int, _f64p]
    L.moka_bw_probe_streams.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.moka_bw_probe_reread.argtypes = [vp, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.moka_bw_probe_gather_big.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.c_double)]
    L.moka_ctx_pci_bus_id.argtypes = [vp, C.c_char_p, C.c_int32]
    L.moka_set_tuning.argtypes = [C.c_int, C.c_int]
    L.moka_get_tuning.argtypes = [C.c_int, C.POINTER(C.c_int)]
    _lib = L
    return L


def check(rc, ctx=None):
    if rc != OK:
        msg = lib().moka_last_error(ctx)
        raise MokaError(rc, msg.decode() if msg else "unknown")


def f64(a):
    return a.ctypes.data_as(_f64p)


def i32(a):
    return a.ctypes.data_as(_i32p)


def i64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def make_desc(mesh, K, resting_thickness_sum=None, max_level_edge_top=None, ordering=ORDER_DEFAULT, patch_cells=0,
              cell_class=None, state_bytes=8):
    """Build a moka_mesh_desc from reference-convention arrays.  Returns (desc, keepalive)."""
    keep = {}

    def A(name, dt, val=None):
        arr = np.ascontiguousarray(getattr(mesh, name) if val is None else val, dtype=dt)
        keep[name] = arr
        return arr

    d = MeshDesc()
    d.nCells, d.nEdges, d.nVertices = mesh.nCells, mesh.nEdges, mesh.nVertices
    d.maxEdges, d.maxEdges2, d.vertexDegree = mesh.maxEdges, mesh.maxEdges2, mesh.vertexDegree
    d.nVertLevels = int(K)
    esv = A("edgeSignOnVertex", np.int32)
    d.edgeSignOnVertexLD = esv.shape[1]
    d.edgeSignOnVertex = i32(esv)
    for n in ("xCell", "yCell", "zCell", "areaCell", "weightsOnEdge", "dvEdge", "dcEdge", "fEdge", "areaTriangle"):
        setattr(d, n, f64(A(n, np.float64)))
    for n in ("nEdgesOnCell", "edgesOnCell", "edgeSignOnCell", "cellsOnEdge", "verticesOnEdge", "nEdgesOnEdge",
              "edgesOnEdge", "edgesOnVertex", "cellsOnVertex"):
        setattr(d, n, i32(A(n, np.int32)))
    if max_level_edge_top is None:
        d.maxLevelEdgeTop = None                       # all ones, VertMesh.jl:32
    else:
        mlt = np.full(mesh.nEdges, int(max_level_edge_top), np.int32) if np.isscalar(max_level_edge_top) \
            else np.asarray(max_level_edge_top)
        d.maxLevelEdgeTop = i32(A("maxLevelEdgeTop", np.int32, mlt))
    if resting_thickness_sum is None:
        resting_thickness_sum = np.ones(mesh.nCells)   # VertMesh.jl:99-100 (test constructor)
    d.restingThicknessSum = f64(A("restingThicknessSum", np.float64, np.asarray(resting_thickness_sum).reshape(-1)))
    d.ordering, d.patch_cells = int(ordering), int(patch_cells)
    d.cellClass = i32(A("cellClass", np.int32, cell_class)) if cell_class is not None else None
    d.stateBytes = int(state_bytes)
    kite = getattr(mesh, "kiteAreasOnVertex", None)          # only the optional nonlinear terms read these two
    if kite is not None:
        d.kiteAreasOnVertex = f64(A("kiteAreasOnVertex", np.float64, kite))
        d.fVertex = f64(A("fVertex", np.float64))
    return d, keep


PLAN_ARRAYS = {  # name -> (id, dtype)
    "eoc": (0, np.int32), "coc": (1, np.int32), "mltc": (2, np.int32), "sdv": (3, np.float64),
    "invArea": (4, np.float64), "areaCell": (5, np.float64), "rsum": (6, np.float64),
    "ehdr": (7, np.int32), "eoe": (8, np.int32), "woe": (9, np.float64), "gInvDc": (10, np.float64),
    "dcEdge": (11, np.float64), "dvEdge": (12, np.float64), "fEdge": (13, np.float64),
    "eov": (14, np.int32), "cv": (15, np.float64),
    "haloStart": (16, np.int32), "haloEdge": (17, np.int32), "leoc": (18, np.uint8), "leoe": (19, np.uint8),
    "cRec": (20, np.uint32), "eRec": (21, np.uint32), "feoe": (22, np.float64),
    "pvStart": (23, np.int32), "pvList": (24, np.int32), "lvoe": (25, np.uint16),
}


class HaloPeerInfo(C.Structure):
    """moka_halo_peer_info: what a rank tells a neighbour so that the neighbour can push halo rows to it (plain data)."""
    _fields_ = [("ipc", (C.c_ubyte * 64) * 15), ("ptr", C.c_uint64 * 15), ("flagPtr", C.c_uint64),
                ("shmName", C.c_char * 64), ("dstCell", C.c_int32), ("dstEdge", C.c_int32), ("nCells", C.c_int32),
                ("nEdges", C.c_int32), ("slot", C.c_int32), ("nNeighbors", C.c_int32), ("pid", C.c_int32),
                ("device", C.c_int32), ("stateBytes", C.c_int32), ("nVertLevels", C.c_int32)]


# int transport(void *user, int what, void *sendbuf_device, void *recvbuf_device)
TRANSPORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)


def class_ranges(handle, of_mesh: bool):
    """(patchStart, cellStart, edgeStart) of the cell classes of a plan / device mesh, nClasses + 1 entries each."""
    fn = lib().moka_mesh_class_ranges if of_mesh else lib().moka_plan_class_ranges
    n = C.c_int32()
    check(fn(handle, 0, C.byref(n), None, None, None))
    a, b, c = (np.empty(n.value + 1, dtype=np.int32) for _ in range(3))
    check(fn(handle, n.value + 1, C.byref(n), i32(a), i32(b), i32(c)))
    return a, b, c


class Plan:
    """Host-only reordered mesh (moka_plan_*): usable without a GPU."""

    def __init__(self, mesh, K, resting_thickness_sum=None, max_level_edge_top=None, ordering=ORDER_DEFAULT,
                 patch_cells=0, cell_class=None, state_bytes=8):
        self._h = C.c_void_p()
        desc, self._keep = make_desc(mesh, K, resting_thickness_sum, max_level_edge_top, ordering, patch_cells,
                                     cell_class, state_bytes)
        check(lib().moka_plan_create(C.byref(desc), C.byref(self._h)))
        inf = MeshInfo()
        check(lib().moka_plan_info(self._h, C.byref(inf)))
        self.info = inf.as_dict()

    def permutation(self, kind):
        n = {CELL: self.info["nCells"], EDGE: self.info["nEdges"], VERTEX: self.info["nVertices"]}[kind]
        out = np.empty(n, dtype=np.int32)
        check(lib().moka_plan_permutation(self._h, kind, i32(out)))
        return out

    def class_ranges(self):
        return class_ranges(self._h, False)

    def patch_ranges(self):
        n = self.info["nPatches"] + 1
        a, b, c = (np.empty(n, dtype=np.int32) for _ in range(3))
        check(lib().moka_plan_patch_ranges(self._h, i32(a), i32(b), i32(c)))
        return a, b, c

    def array(self, name):
        which, dt = PLAN_ARRAYS[name]
        ptr, cnt = C.c_void_p(), C.c_int64()
        check(lib().moka_plan_array(self._h, which, C.byref(ptr), C.byref(cnt)))
        buf = (C.c_char * (cnt.value * np.dtype(dt).itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def close(self):
        if self._h:
            lib().moka_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
