# MokaHIP.jl -- Julia shim that puts libmoka_hip.so behind MOKA.jl's own interface.
#
# NOT EXECUTED IN THIS PIPELINE (no Julia toolchain on either box).  It documents, method by method,
# the binding a MOKA.jl maintainer adds so that src/driver/mpas_ocean.jl runs unchanged except for
# its backend line (`backend = MokaHIP.Backend()` instead of `CUDABackend()`, mpas_ocean.jl:28).
# The Python mirror mpas-ocean.jl_amd/moka_hip/api.py makes the same calls in the same order and is
# what the tests exercise.
#
# Design: arrays stay plain host `Array`s on the Julia side (so OutPut.jl, the tests' norms, etc. keep
# working); the device copy lives in a moka_state owned by the library.  `MArray` is a thin
# AbstractArray whose getindex/copyto! download lazily and whose setindex!/copyto! upload.
module MokaHIP

import Adapt
import KernelAbstractions as KA
using MOKA
using MOKA: Mesh, HorzMesh, VerticalMesh, PrognosticVars, DiagnosticVars, TendencyVars, ModelSetup,
            ForwardEuler, RungeKutta4

const lib = joinpath(@__DIR__, "..", "libmoka_hip.so")

# ---- backend tag -------------------------------------------------------------------------------
mutable struct Backend <: KA.Backend
    ctx::Ptr{Cvoid}
    function Backend(device::Integer = 0)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:moka_ctx_create, lib), Cint, (Cint, Ref{Ptr{Cvoid}}), device, ref), C_NULL)
        b = new(ref[])
        finalizer(x -> ccall((:moka_ctx_destroy, lib), Cvoid, (Ptr{Cvoid},), x.ctx), b)
        b
    end
end

function check(rc::Cint, ctx)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:moka_last_error, lib), Cstring, (Ptr{Cvoid},), ctx))
    error("libmoka_hip: $msg")            # reference style: error("...") (time_integration.jl:23, VertMesh.jl:51)
end

KA.synchronize(b::Backend) = check(ccall((:moka_sync, lib), Cint, (Ptr{Cvoid},), b.ctx), b.ctx)

# ---- mesh: Adapt.adapt_structure(backend, ::Mesh)  (MPASMesh.jl:26) ------------------------------
# mirrors `struct moka_mesh_desc` of include/moka_hip.h field by field (isbits, C layout)
struct MeshDesc
    nCells::Int32; nEdges::Int32; nVertices::Int32
    maxEdges::Int32; maxEdges2::Int32; vertexDegree::Int32
    nVertLevels::Int32; edgeSignOnVertexLD::Int32
    xCell::Ptr{Float64}; yCell::Ptr{Float64}; zCell::Ptr{Float64}
    nEdgesOnCell::Ptr{Int32}; edgesOnCell::Ptr{Int32}; edgeSignOnCell::Ptr{Int32}; areaCell::Ptr{Float64}
    cellsOnEdge::Ptr{Int32}; verticesOnEdge::Ptr{Int32}; nEdgesOnEdge::Ptr{Int32}; edgesOnEdge::Ptr{Int32}
    weightsOnEdge::Ptr{Float64}; dvEdge::Ptr{Float64}; dcEdge::Ptr{Float64}; fEdge::Ptr{Float64}
    edgesOnVertex::Ptr{Int32}; cellsOnVertex::Ptr{Int32}; edgeSignOnVertex::Ptr{Int32}; areaTriangle::Ptr{Float64}
    maxLevelEdgeTop::Ptr{Int32}; restingThicknessSum::Ptr{Float64}
    ordering::Int32; patch_cells::Int32
    cellClass::Ptr{Int32}          # C_NULL on one GPU; 0/1/2 per local cell in the multi-GPU layer
    stateBytes::Int32              # 0/8 = Float64 state (the reference); 4 = fp32 storage, fp64 arithmetic (RK4 only)
    kiteAreasOnVertex::Ptr{Float64}; fVertex::Ptr{Float64}   # C_NULL unless the optional nonlinear terms are wanted
end

struct DeviceMesh{HM,VM}           # what Adapt returns: the host Mesh plus the library handle
    host::Mesh{HM,VM}
    handle::Ptr{Cvoid}
    backend::Backend
end

function Adapt.adapt_structure(b::Backend, m::Mesh)
    C, D, E, V = m.HorzMesh.PrimaryCells, m.HorzMesh.DualCells, m.HorzMesh.Edges, m.VertMesh
    rsum = vec(Array(V.restingThicknessSum))                      # (1,nC) or (nC): indexed linearly (SURVEY N5)
    GC.@preserve C D E V rsum begin
        d = MeshDesc(C.nCells, E.nEdges, D.nVertices, C.maxEdges, size(E.edgesOnEdge, 1), D.vertexDegree,
                     V.nVertLevels, size(D.edgeSignOnVertex, 1),
                     pointer(C.xᶜ), pointer(C.yᶜ), pointer(C.zᶜ),
                     pointer(C.nEdgesOnCell), pointer(C.edgesOnCell), pointer(C.edgeSignOnCell), pointer(C.areaCell),
                     pointer(E.cellsOnEdge), pointer(E.verticesOnEdge), pointer(E.nEdgesOnEdge), pointer(E.edgesOnEdge),
                     pointer(E.weightsOnEdge), pointer(E.dvEdge), pointer(E.dcEdge), pointer(E.fᵉ),
                     pointer(D.edgesOnVertex), pointer(D.cellsOnVertex), pointer(D.edgeSignOnVertex), pointer(D.areaTriangle),
                     pointer(V.maxLevelEdge.Top), pointer(rsum), 0, 0, C_NULL, 0, C_NULL, C_NULL)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:moka_mesh_create, lib), Cint, (Ptr{Cvoid}, Ref{MeshDesc}, Ref{Ptr{Cvoid}}), b.ctx, d, ref), b.ctx)
        return DeviceMesh(m, ref[], b)
    end
end

# ---- operators (Operators.jl:46,102,151,179): host arrays in/out, synchronous -------------------
function MOKA.GradientOnEdge!(grad::Matrix{Float64}, h::Matrix{Float64}, m::DeviceMesh; backend = m.backend, workgroupsize = 64)
    check(ccall((:moka_gradient_on_edge, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), m.handle, h, grad), m.backend.ctx)
end
function MOKA.DivergenceOnCell!(div::Matrix{Float64}, V::Matrix{Float64}, temp::Matrix{Float64}, m::DeviceMesh; backend = m.backend, nthreads = 50)
    check(ccall((:moka_divergence_on_cell, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), m.handle, V, temp, div), m.backend.ctx)
end
function MOKA.CurlOnVertex!(curl::Matrix{Float64}, V::Matrix{Float64}, m::DeviceMesh; backend = m.backend)
    check(ccall((:moka_curl_on_vertex, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), m.handle, V, curl), m.backend.ctx)
end
function MOKA.interpolateCell2Edge!(e::Matrix{Float64}, c::Matrix{Float64}, m::DeviceMesh; backend = m.backend)
    check(ccall((:moka_interpolate_cell2edge, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint), m.handle, c, e, 1), m.backend.ctx)
end

# ---- state: one moka_state behind Prog / Diag / Tend ---------------------------------------------
mutable struct State
    handle::Ptr{Cvoid}
    mesh::DeviceMesh
end
function State(m::DeviceMesh)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:moka_state_create, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), m.backend.ctx, m.handle, ref), m.backend.ctx)
    s = State(ref[], m)
    finalizer(x -> ccall((:moka_state_destroy, lib), Cvoid, (Ptr{Cvoid},), x.handle), s)
    s
end

# field ids of include/moka_hip.h (moka_field)
const F_SSH, F_U, F_H, F_HEDGE, F_FLUX, F_DIV, F_VORT, F_TENDU, F_TENDH = Int32.(0:8)

"Device-backed array: Adapt.adapt(KA.CPU(), a) downloads (OutPut.jl:122-124), copyto!(a, host) uploads."
struct MArray{N} <: AbstractArray{Float64,N}
    state::State; field::Int32; level::Int32; dims::NTuple{N,Int}
end
Base.size(a::MArray) = a.dims
function Base.Array(a::MArray{N}) where {N}
    out = Array{Float64,N}(undef, a.dims)
    check(ccall((:moka_state_download, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}), a.state.handle, a.field, a.level, out), a.state.mesh.backend.ctx)
    out
end
Base.getindex(a::MArray, i...) = Array(a)[i...]                 # scalar indexing = @allowscalar: correct, slow
function Base.copyto!(a::MArray, src::Array{Float64})
    check(ccall((:moka_state_upload, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}), a.state.handle, a.field, a.level, src), a.state.mesh.backend.ctx)
    a
end
Adapt.adapt_storage(::KA.CPU, a::MArray) = Array(a)
KA.get_backend(a::MArray) = a.state.mesh.backend                 # mpas_ocean.jl:48

# PrognosticVars(Config, Mesh; backend) (PrognosticVars.jl:59): read on the host as the reference does,
# then Adapt.adapt(backend, ...) becomes "create state + upload into both time levels".
function device_state(ssh::Vector{Float64}, u::Matrix{Float64}, h::Matrix{Float64}, m::DeviceMesh)
    s = State(m); K, nE = size(u); nC = length(ssh); nV = m.host.HorzMesh.DualCells.nVertices
    mk(f, lev, dims) = MArray{length(dims)}(s, f, Int32(lev), dims)
    sshv = [mk(F_SSH, t, (nC,)) for t in 0:1]; uv = [mk(F_U, t, (K, nE)) for t in 0:1]; hv = [mk(F_H, t, (K, nC)) for t in 0:1]
    for t in 1:2; copyto!(sshv[t], ssh); copyto!(uv[t], u); copyto!(hv[t], h); end
    Prog = (ssh = sshv, normalVelocity = uv, layerThickness = hv, state = s)
    Diag = (layerThicknessEdge = mk(F_HEDGE, 1, (K, nE)), thicknessFlux = mk(F_FLUX, 1, (K, nE)),
            velocityDivCell = mk(F_DIV, 1, (K, nC)), relativeVorticity = mk(F_VORT, 1, (K, nV)), state = s)
    Tend = (tendNormalVelocity = mk(F_TENDU, 1, (K, nE)), tendLayerThickness = mk(F_TENDH, 1, (K, nC)), state = s)
    return Prog, Diag, Tend
end

# ---- forward model --------------------------------------------------------------------------------
const REFERENCE_COMPAT = Int32(7)        # MOKA_FE_STALE_HEDGE | ACCUM_VORT | LEVEL1_ONLY

# ocn_timestep(timestep, Prog, Diag, Tend, S, ForwardEuler; backend)   time_integration.jl:150
function MOKA.ocn_timestep(timestep, Prog, Diag, Tend, S::ModelSetup, ::Type{ForwardEuler}; backend::Backend)
    dt = Array(timestep)[1]              # the reference's 1-element device array (mpas_ocean.jl:36-37)
    check(ccall((:moka_step_fe, lib), Cint, (Ptr{Cvoid}, Cdouble, Cint), Prog.state.handle, dt, REFERENCE_COMPAT), backend.ctx)
end
# ocn_timestep(Prog, Diag, Tend, S, RungeKutta4; backend)              time_integration.jl:61
function MOKA.ocn_timestep(Prog, Diag, Tend, S::ModelSetup, ::Type{RungeKutta4}; backend::Backend)
    dt = convert(Float64, Dates.value(Dates.Second(S.timeManager.timeStep)))
    check(ccall((:moka_step_rk4, lib), Cint, (Ptr{Cvoid}, Cdouble), Prog.state.handle, dt), backend.ctx)
end
# diagnostic_compute!(Mesh, Diag, Prog; backend)                       DiagnosticVars.jl:108
MOKA.diagnostic_compute!(m::DeviceMesh, Diag, Prog; backend::Backend) =
    check(ccall((:moka_diagnostic_compute, lib), Cint, (Ptr{Cvoid}, Cint), Prog.state.handle, REFERENCE_COMPAT), backend.ctx)
# computeNormalVelocityTendency! / computeLayerThicknessTendency!      normalVelocity.jl:21, layerThickness.jl:14
MOKA.computeNormalVelocityTendency!(Tend, Prog, Diag, m::DeviceMesh, Config; backend::Backend) =
    check(ccall((:moka_compute_normal_velocity_tendency, lib), Cint, (Ptr{Cvoid}, Cint), Prog.state.handle, REFERENCE_COMPAT), backend.ctx)
MOKA.computeLayerThicknessTendency!(Tend, Prog, Diag, m::DeviceMesh, Config; backend::Backend) =
    check(ccall((:moka_compute_layer_thickness_tendency, lib), Cint, (Ptr{Cvoid}, Cint), Prog.state.handle, REFERENCE_COMPAT), backend.ctx)
# sumArray + mycopyto! of ocn_run_loop(sumCPU, sumGPU, ...)            run_loop.jl:39-43
function sum_sq_ssh(Prog)
    out = Ref{Float64}(0.0)
    check(ccall((:moka_sum_sq, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Float64}), Prog.state.handle, F_SSH, 1, out), Prog.state.mesh.backend.ctx)
    out[]
end
# ocn_run_loop itself (run_loop.jl:8-22) needs no change: it only calls advance!, ocn_timestep, isRinging, reset!.

# ---- beyond the reference's GPU path (all optional) ------------------------------------------------
# nonlinear terms (potential-vorticity Coriolis + kinetic-energy gradient); the mesh descriptor must carry
# kiteAreasOnVertex / fVertex.  Off by default: the reference has only the linear terms.
set_nonlinear!(Prog, on::Bool = true) =
    check(ccall((:moka_set_nonlinear, lib), Cint, (Ptr{Cvoid}, Cint), Prog.state.handle, on ? 1 : 0), Prog.state.mesh.backend.ctx)
# Del2 momentum mixing on top of them (what horizontal_momentum_mixing.jl:53-80 sketches); 0 = off
set_viscosity_del2!(Prog, viscDel2::Float64) =
    check(ccall((:moka_set_viscosity_del2, lib), Cint, (Ptr{Cvoid}, Cdouble), Prog.state.handle, viscDel2), Prog.state.mesh.backend.ctx)

# reverse mode: what an EnzymeRules rule for ocn_run_loop on this backend calls (ext/MPASEnzymeExt.jl registers such
# rules for mycopyto! already, :13-38).  d sum(ssh^2) / d initial state, test/enzyme/test_Enzyme_end2end.jl.
mutable struct Tape; handle::Ptr{Cvoid}; state::State; end
function Tape(Prog, capacity::Integer)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:moka_tape_create, lib), Cint, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), Prog.state.handle, capacity, ref), Prog.state.mesh.backend.ctx)
    t = Tape(ref[], Prog.state)
    finalizer(x -> ccall((:moka_tape_destroy, lib), Cvoid, (Ptr{Cvoid},), x.handle), t)
    t
end
step_fe!(t::Tape, dt; flags = REFERENCE_COMPAT) =
    check(ccall((:moka_step_fe_taped, lib), Cint, (Ptr{Cvoid}, Cdouble, Cint), t.handle, dt, flags), t.state.mesh.backend.ctx)
step_rk4!(t::Tape, dt) =
    check(ccall((:moka_step_rk4_taped, lib), Cint, (Ptr{Cvoid}, Cdouble), t.handle, dt), t.state.mesh.backend.ctx)
function gradient!(t::Tape, d_ssh::Vector{Float64}, d_u::Matrix{Float64}, d_h::Matrix{Float64})   # d_Prog of the reference test
    ctx = t.state.mesh.backend.ctx
    check(ccall((:moka_adjoint_seed_sum_sq_ssh, lib), Cint, (Ptr{Cvoid},), t.handle), ctx)
    check(ccall((:moka_adjoint_sweep, lib), Cint, (Ptr{Cvoid},), t.handle), ctx)
    for (f, a) in ((F_SSH, d_ssh), (F_U, d_u), (F_H, d_h))
        check(ccall((:moka_adjoint_download, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), t.handle, f, a), ctx)
    end
end

end # module
