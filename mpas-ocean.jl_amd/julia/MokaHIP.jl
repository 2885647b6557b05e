# MokaHIP.jl -- Julia shim that puts libmoka_hip.so behind MOKA.jl's own interface.
#
# STATUS: written against the reference sources, NEVER EXECUTED (no Julia toolchain on either box of this pipeline).
# mpas-ocean.jl_amd/moka_hip/shim.py is a line-by-line Python transliteration of the logic below (lazy arrays, binding at
# the first device call, version-stamped host copies) and tests/test_gpu_parity.py::test_driver_in_the_reference_constructor_order
# replays src/driver/mpas_ocean.jl:20-53 through it on the GPU; INTEGRATION.md lists, per driver line, the method that
# catches it.  The only line of the driver that changes is `backend = CUDABackend()` (mpas_ocean.jl:28) ->
# `backend = MokaHIP.Backend()`.
#
# How the reference's own constructors end up on the library without being touched:
#   * every array the reference puts "on the backend" -- Adapt.adapt(backend, a) (HorzMesh.jl:354-398,
#     PrognosticVars.jl:101-104, TendencyVars.jl:66), KA.zeros(backend, T, dims...) (DiagnosticVars.jl:90-93,
#     TendencyVars.jl:62, mpas_ocean.jl:36), KA.ones (VertMesh.jl:32-33) -- becomes an `MArray`: a host Array plus an
#     optional binding to a field of a device state.  Unbound, it behaves like the Array it wraps (the mesh arrays and
#     the 1-element `timestep` stay that way for ever).
#   * the inner constructors' checks (Architectures.jl:19-46: same type name, same backend, same eltype) hold for MArrays;
#     `deepcopy` per time level (PrognosticVars.jl:50-54) copies the host data of a still unbound array.
#   * the first call that needs the device (ocn_timestep, diagnostic_compute!, compute...Tendency!) BINDS: the device
#     mesh is created from the MArrays of `Setup.mesh` (moka_mesh_create), one moka_state is created, the host data of
#     Prog's arrays is uploaded, and every array of Prog / Diag / Tend gets (state, field id, time level).  From then on
#     the device holds the truth; reading a bound array downloads it ONCE per device change (version stamp), scalar
#     writes are collected on the host and uploaded before the next device call.
#   * Adapt.adapt_structure(KA.CPU(), x) of write_netcdf (OutPut.jl:122-124) reaches Adapt.adapt_storage(::KA.CPU, ::MArray)
#     = a plain Array with the current device contents.
module MokaHIP

import Adapt
import Dates
import KernelAbstractions as KA
using MOKA
using MOKA: Mesh, HorzMesh, VerticalMesh, PrognosticVars, DiagnosticVars, TendencyVars, ModelSetup,
            ForwardEuler, RungeKutta4

const lib = joinpath(@__DIR__, "..", "libmoka_hip.so")

# ---- backend tag -------------------------------------------------------------------------------
# A GPU backend for KernelAbstractions' purposes (`typeof(backend) <: KA.GPU`, mpas_ocean.jl:49).  The context is
# reference-counted on the Julia side: finalizers run in no particular order, and moka_state_destroy / moka_mesh_destroy /
# moka_tape_destroy dereference the context, so it is destroyed only when the tag AND everything created on it are gone.
mutable struct Backend <: KA.GPU
    ctx::Ptr{Cvoid}
    refs::Int                      # live meshes / states / tapes
    dead::Bool                     # the tag itself has been finalized
    function Backend(device::Integer = 0)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:moka_ctx_create, lib), Cint, (Cint, Ref{Ptr{Cvoid}}), device, ref), C_NULL)
        b = new(ref[], 0, false)
        finalizer(b) do x
            x.dead = true
            x.refs == 0 && destroy_ctx!(x)
        end
        b
    end
end
function destroy_ctx!(b::Backend)
    b.ctx == C_NULL && return
    ccall((:moka_ctx_destroy, lib), Cvoid, (Ptr{Cvoid},), b.ctx)
    b.ctx = C_NULL
end
retain!(b::Backend) = (b.refs += 1; b)
function release!(b::Backend)
    b.refs -= 1
    b.dead && b.refs == 0 && destroy_ctx!(b)
end

function check(rc::Cint, ctx)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:moka_last_error, lib), Cstring, (Ptr{Cvoid},), ctx))
    error("libmoka_hip: $msg")            # reference style: error("...") (time_integration.jl:23, VertMesh.jl:51)
end

KA.synchronize(b::Backend) = check(ccall((:moka_sync, lib), Cint, (Ptr{Cvoid},), b.ctx), b.ctx)

# ---- arrays "on the backend" ----------------------------------------------------------------------
# field ids of include/moka_hip.h (moka_field)
const F_SSH, F_U, F_H, F_HEDGE, F_FLUX, F_DIV, F_VORT, F_TENDU, F_TENDH = Int32.(0:8)

mutable struct State                   # one moka_state behind Prog / Diag / Tend of a model
    handle::Ptr{Cvoid}
    mesh                               # DeviceMesh (keeps it alive)
    backend::Backend
    version::Int                       # bumped by every call that changes device fields
    bound::Vector{WeakRef}             # the MArrays bound to this state (to flush pending host writes)
    tapes::Int                         # live tapes on this state: moka_tape_destroy dereferences the state (it may have to
    dead::Bool                         # materialise lazily pending tendencies), and finalizers run in no particular order --
end                                    # so the state is destroyed only when its own finalizer AND every tape's have run
function destroy_state!(s::State)
    s.handle == C_NULL && return
    ccall((:moka_state_destroy, lib), Cvoid, (Ptr{Cvoid},), s.handle)
    s.handle = C_NULL
    release!(s.backend)
end

mutable struct MArray{T,N} <: AbstractArray{T,N}
    host::Array{T,N}
    backend::Backend
    state::Union{Nothing,State}        # nothing = unbound: `host` is the data
    field::Int32
    level::Int32
    host_version::Int                  # state.version the host copy corresponds to (-1 = never downloaded)
    host_dirty::Bool                   # host copy carries writes the device has not seen
end
MArray(a::Array{T,N}, b::Backend) where {T,N} = MArray{T,N}(a, b, nothing, Int32(-1), Int32(0), -1, false)

Base.size(a::MArray) = size(a.host)
Base.IndexStyle(::Type{<:MArray}) = IndexLinear()
KA.get_backend(a::MArray) = a.backend                                   # Architectures.jl:33; mpas_ocean.jl:48
Base.similar(a::MArray, ::Type{T}, dims::Dims) where {T} = MArray(Array{T}(undef, dims), a.backend)
# deepcopy of a still unbound array copies the host data (PrognosticVars.jl:50-54); a bound one is read back first
function Base.deepcopy_internal(a::MArray{T,N}, d::IdDict) where {T,N}
    haskey(d, a) && return d[a]
    c = MArray(copy(Array(a)), a.backend)
    d[a] = c
    c
end

"bring the host copy of a bound array up to date (one download per device change, not one per element)"
function sync_host!(a::MArray{Float64})
    s = a.state
    (s === nothing || a.host_dirty || a.host_version == s.version) && return a
    check(ccall((:moka_state_download, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}), s.handle, a.field, a.level, a.host), s.backend.ctx)
    a.host_version = s.version
    a
end
sync_host!(a::MArray) = a                                               # Int32 mesh arrays are never bound
Base.Array(a::MArray) = copy(sync_host!(a).host)
Base.getindex(a::MArray, i::Int) = sync_host!(a).host[i]                # @allowscalar-style access: correct and cheap
function Base.setindex!(a::MArray, v, i::Int)                           # e.g. `@allowscalar timestep[1] = dt` (mpas_ocean.jl:37)
    sync_host!(a).host[i] = v
    a.state === nothing || (a.host_dirty = true)
    a
end
function Base.copyto!(a::MArray{T,N}, src::Array{T,N}) where {T,N}
    copyto!(a.host, src)
    a.state === nothing || (a.host_dirty = true)
    a
end
Base.copyto!(dst::Array{T,N}, a::MArray{T,N}) where {T,N} = copyto!(dst, sync_host!(a).host)   # mycopyto!(sumCPU, sumGPU)

# Adapt / KernelAbstractions entry points the reference's constructors use
Adapt.adapt_storage(b::Backend, a::Array) = MArray(copy(a), b)          # on_architecture (Architectures.jl:12)
Adapt.adapt_storage(::Backend, a::MArray) = a
Adapt.adapt_storage(::KA.CPU, a::MArray) = Array(a)                     # write_netcdf (OutPut.jl:122-124)
KA.allocate(b::Backend, ::Type{T}, dims::Tuple) where {T} = MArray(Array{T}(undef, dims), b)
KA.zeros(b::Backend, ::Type{T}, dims::Tuple) where {T} = MArray(zeros(T, dims), b)   # DiagnosticVars.jl:90-93, mpas_ocean.jl:36
KA.ones(b::Backend, ::Type{T}, dims::Tuple) where {T} = MArray(ones(T, dims), b)     # VertMesh.jl:32-33
KA.zeros(b::Backend, ::Type{T}, dims::Integer...) where {T} = KA.zeros(b, T, Tuple(dims))
KA.ones(b::Backend, ::Type{T}, dims::Integer...) where {T} = KA.ones(b, T, Tuple(dims))

# ---- mesh: built from the MArrays of the reference's Mesh at binding time ---------------------------
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
    cellClass::Ptr{Int32}          # C_NULL on one GPU; 0 / 1 / 2 + neighbour index per local cell in the multi-GPU layer
    stateBytes::Int32              # 0/8 = Float64 state (the reference); 4 = fp32 storage, fp64 arithmetic (RK4 only)
    kiteAreasOnVertex::Ptr{Float64}; fVertex::Ptr{Float64}   # C_NULL unless the optional nonlinear terms are wanted
end

mutable struct DeviceMesh
    handle::Ptr{Cvoid}
    backend::Backend
end
function destroy_mesh!(dm::DeviceMesh)
    dm.handle == C_NULL && return
    ccall((:moka_mesh_destroy, lib), Cvoid, (Ptr{Cvoid},), dm.handle)
    dm.handle = C_NULL
    release!(dm.backend)
end
# reference Mesh object -> its device mesh, by IDENTITY.  `Mesh` is an immutable struct in the reference (MPASMesh.jl:19), so
# the identity used is that of two of the (mutable) MArrays it holds -- areaCell for the horizontal mesh,
# restingThicknessSum for the vertical one -- together with the backend: two Mesh objects read from the same file, or one
# HorzMesh under two VerticalMeshes (another nVertLevels / restingThickness), or the same Mesh on two backends, get a
# DeviceMesh each.  (Round 3 keyed a WeakKeyDict by the areaCell array itself; AbstractArray keys hash and compare by
# CONTENT, so equal-valued meshes shared one DeviceMesh with the wrong nVertLevels or context, and every lookup hashed
# nCells elements: ADVICE r03.)  The arrays are referenced weakly: once the caller has dropped its Mesh the entry is pruned
# at the next lookup, the DeviceMesh becomes collectable (states that use it hold it themselves), its finalizer runs
# moka_mesh_destroy and the context's count can reach zero.
struct MeshEntry
    horz::WeakRef
    vert::WeakRef
    dm::DeviceMesh
end
const MESHES = Dict{NTuple{3,UInt},MeshEntry}()
mesh_arrays(m::Mesh) = (m.HorzMesh.PrimaryCells.areaCell, m.VertMesh.restingThicknessSum)
mesh_key(m::Mesh, b::Backend) = (objectid(mesh_arrays(m)[1]), objectid(mesh_arrays(m)[2]), objectid(b))
function prune_meshes!()
    for (k, e) in collect(MESHES)
        (e.horz.value === nothing || e.vert.value === nothing) && delete!(MESHES, k)
    end
end
function lookup_mesh(m::Mesh, b::Backend)
    e = get(MESHES, mesh_key(m, b), nothing)
    e === nothing && return nothing
    h, v = mesh_arrays(m)
    (e.horz.value === h && e.vert.value === v) ? e.dm : nothing      # (an objectid can be reused after its object is gone)
end
"release the device copies of `m` now (otherwise: when `m` is collected)"
function close!(m::Mesh)
    h, v = mesh_arrays(m)
    for (k, e) in collect(MESHES)
        if e.horz.value === h && e.vert.value === v
            delete!(MESHES, k)
            destroy_mesh!(e.dm)
        end
    end
    nothing
end

hostof(a::MArray) = a.host
hostof(a::Array) = a

"moka_mesh_create from the arrays the reference's Mesh holds (Adapt.adapt_structure(backend, ::Mesh), MPASMesh.jl:26)"
function device_mesh(m::Mesh, b::Backend)
    dm0 = lookup_mesh(m, b)
    dm0 === nothing || return dm0
    prune_meshes!()
    C, D, E, V = m.HorzMesh.PrimaryCells, m.HorzMesh.DualCells, m.HorzMesh.Edges, m.VertMesh
    rsum = vec(copy(hostof(V.restingThicknessSum)))               # (1,nC) or (nC): indexed linearly (SURVEY N5)
    arrs = map(hostof, (C.xᶜ, C.yᶜ, C.zᶜ, C.nEdgesOnCell, C.edgesOnCell, C.edgeSignOnCell, C.areaCell,
                        E.cellsOnEdge, E.verticesOnEdge, E.nEdgesOnEdge, E.edgesOnEdge, E.weightsOnEdge, E.dvEdge, E.dcEdge, E.fᵉ,
                        D.edgesOnVertex, D.cellsOnVertex, D.edgeSignOnVertex, D.areaTriangle, V.maxLevelEdge.Top))
    GC.@preserve arrs rsum begin
        d = MeshDesc(C.nCells, E.nEdges, D.nVertices, C.maxEdges, size(arrs[11], 1), D.vertexDegree,
                     V.nVertLevels, size(arrs[18], 1),
                     pointer(arrs[1]), pointer(arrs[2]), pointer(arrs[3]),
                     pointer(arrs[4]), pointer(arrs[5]), pointer(arrs[6]), pointer(arrs[7]),
                     pointer(arrs[8]), pointer(arrs[9]), pointer(arrs[10]), pointer(arrs[11]),
                     pointer(arrs[12]), pointer(arrs[13]), pointer(arrs[14]), pointer(arrs[15]),
                     pointer(arrs[16]), pointer(arrs[17]), pointer(arrs[18]), pointer(arrs[19]),
                     pointer(arrs[20]), pointer(rsum), 0, 0, C_NULL, 0, C_NULL, C_NULL)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:moka_mesh_create, lib), Cint, (Ptr{Cvoid}, Ref{MeshDesc}, Ref{Ptr{Cvoid}}), b.ctx, d, ref), b.ctx)
        dm = DeviceMesh(ref[], retain!(b))
        finalizer(destroy_mesh!, dm)
        MESHES[mesh_key(m, b)] = MeshEntry(WeakRef(mesh_arrays(m)[1]), WeakRef(mesh_arrays(m)[2]), dm)
        return dm
    end
end

# ---- operators (Operators.jl:46,102,151,179): arrays on the backend in/out, synchronous -----------------
# called by test/ocn/test_Operators.jl:47,67,85 with arrays adapted to the backend: unbound MArrays, i.e. host data.
# A BOUND array (a field of Prog / Diag / Tend) works too: inputs are read back first (sync_host!), outputs are written on
# the host copy and marked dirty, so the next device call of the model uploads them (flush_host!).
"host memory of an operator INPUT, up to date"
input_host(a::MArray) = sync_host!(a).host
"host memory of an operator OUTPUT: current contents first (read-modify-write operators), pending upload afterwards"
function output_host(a::MArray)
    sync_host!(a)
    a.state === nothing || (a.host_dirty = true)
    a.host
end
function MOKA.GradientOnEdge!(grad::MArray{Float64,2}, h::MArray{Float64,2}, m::Mesh; backend = grad.backend, workgroupsize = 64)
    dm = device_mesh(m, backend)
    check(ccall((:moka_gradient_on_edge, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), dm.handle, input_host(h), output_host(grad)), backend.ctx)
end
function MOKA.DivergenceOnCell!(div::MArray{Float64,2}, V::MArray{Float64,2}, temp::MArray{Float64,2}, m::Mesh; backend = div.backend, nthreads = 50)
    dm = device_mesh(m, backend)
    check(ccall((:moka_divergence_on_cell, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), dm.handle, input_host(V), output_host(temp), output_host(div)), backend.ctx)
end
function MOKA.CurlOnVertex!(curl::MArray{Float64,2}, V::MArray{Float64,2}, m::Mesh; backend = curl.backend)
    dm = device_mesh(m, backend)
    check(ccall((:moka_curl_on_vertex, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), dm.handle, input_host(V), output_host(curl)), backend.ctx)
end
function MOKA.interpolateCell2Edge!(e::MArray{Float64,2}, c::MArray{Float64,2}, m::Mesh; backend = e.backend)
    dm = device_mesh(m, backend)
    check(ccall((:moka_interpolate_cell2edge, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint), dm.handle, input_host(c), output_host(e), 1), backend.ctx)
end

# reverse / forward mode of the three operators (include/moka_hip.h: moka_*_vjp / _jvp): what the EnzymeRules of
# MokaHIPEnzymeExt.jl call.  Arguments are the SHADOW arrays; conventions are Enzyme's (input shadows accumulate, shadows of
# overwritten outputs are zeroed, the curl shadow stays).
function gradient_vjp!(d_grad::MArray{Float64,2}, d_h::MArray{Float64,2}, m::Mesh, backend)
    dm = device_mesh(m, backend)
    check(ccall((:moka_gradient_on_edge_vjp, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), dm.handle, output_host(d_grad), output_host(d_h)), backend.ctx)
end
function gradient_jvp!(d_grad::MArray{Float64,2}, d_h::MArray{Float64,2}, m::Mesh, backend)
    dm = device_mesh(m, backend)
    check(ccall((:moka_gradient_on_edge_jvp, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), dm.handle, input_host(d_h), output_host(d_grad)), backend.ctx)
end
function divergence_vjp!(d_div::MArray{Float64,2}, d_V::MArray{Float64,2}, d_temp::MArray{Float64,2}, m::Mesh, backend)
    dm = device_mesh(m, backend)
    check(ccall((:moka_divergence_on_cell_vjp, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), dm.handle, output_host(d_div), output_host(d_V), output_host(d_temp)), backend.ctx)
end
function divergence_jvp!(d_div::MArray{Float64,2}, d_V::MArray{Float64,2}, d_temp::MArray{Float64,2}, m::Mesh, backend)
    dm = device_mesh(m, backend)
    check(ccall((:moka_divergence_on_cell_jvp, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), dm.handle, input_host(d_V), output_host(d_temp), output_host(d_div)), backend.ctx)
end
function curl_vjp!(d_curl::MArray{Float64,2}, d_V::MArray{Float64,2}, m::Mesh, backend)
    dm = device_mesh(m, backend)
    check(ccall((:moka_curl_on_vertex_vjp, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), dm.handle, input_host(d_curl), output_host(d_V)), backend.ctx)
end
function curl_jvp!(d_curl::MArray{Float64,2}, d_V::MArray{Float64,2}, m::Mesh, backend)
    dm = device_mesh(m, backend)
    check(ccall((:moka_curl_on_vertex_jvp, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), dm.handle, input_host(d_V), output_host(d_curl)), backend.ctx)
end

# ---- binding Prog / Diag / Tend to one device state ---------------------------------------------------
"upper limit of the per-array placement trials at binding (moka_state_optimize_placement); <= 1 switches the search off"
const PLACEMENT_TRIES = Ref{Cint}(24)
const MProg = PrognosticVars{<:Any,<:MArray}          # the reference's struct, parametrised by our array type
const MDiag = DiagnosticVars{<:Any,<:MArray}
const MTend = TendencyVars{<:Any,<:MArray}

function bind!(a::MArray{Float64}, s::State, field, level; upload::Bool)
    a.state === s && return
    a.state === nothing || error("MokaHIP: array is already bound to another model state")
    if upload
        check(ccall((:moka_state_upload, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}), s.handle, field, level, a.host), s.backend.ctx)
    end
    a.state, a.field, a.level = s, Int32(field), Int32(level)
    a.host_version, a.host_dirty = upload ? s.version : -1, false
    push!(s.bound, WeakRef(a))
end

"the state behind Prog (created and filled at the first device call); Diag / Tend join it"
function state_of(Prog::MProg, Diag, Tend, S::ModelSetup, b::Backend)
    s = Prog.ssh[end].state
    if s === nothing
        dm = device_mesh(S.mesh, b)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:moka_state_create, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), b.ctx, dm.handle, ref), b.ctx)
        s = State(ref[], dm, retain!(b), 0, WeakRef[], 0, false)
        finalizer(s) do x
            x.dead = true
            x.tapes == 0 && destroy_state!(x)
        end
        length(Prog.ssh) == 2 || error("nTimeLevels must be <= 2")           # time_integration.jl:23
        for t in 1:2                                                           # Julia index 1 = previous = level 0, end = current = 1
            bind!(Prog.ssh[t], s, F_SSH, t - 1; upload = true)
            bind!(Prog.normalVelocity[t], s, F_U, t - 1; upload = true)
            bind!(Prog.layerThickness[t], s, F_H, t - 1; upload = true)
        end
        # The caller (src/driver/mpas_ocean.jl:28-39) only ever calls ocn_init and the step, so where the allocator put this
        # state's arrays -- 5-14 % of every stage launch, DESIGN.md section 5 -- is settled here, by the library itself:
        # up to PLACEMENT_TRIES per-array re-allocations, each kept only if the launches it takes part in got faster.  The
        # state's contents are untouched.  Must come before any tape of the state exists (the library refuses afterwards).
        # (an optimisation: its failure -- e.g. no memory for a candidate -- leaves the state as it was and is not an error of the model)
        PLACEMENT_TRIES[] > 1 && ccall((:moka_state_optimize_placement, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}),
                                       s.handle, PLACEMENT_TRIES[], C_NULL, C_NULL)
    end
    if Diag !== nothing && Diag.layerThicknessEdge.state === nothing         # KA.zeros on the host == zero-initialised device fields
        for (a, f) in ((Diag.layerThicknessEdge, F_HEDGE), (Diag.thicknessFlux, F_FLUX), (Diag.velocityDivCell, F_DIV), (Diag.relativeVorticity, F_VORT))
            bind!(a, s, f, 1; upload = true)
        end
    end
    if Tend !== nothing && Tend.tendNormalVelocity.state === nothing
        bind!(Tend.tendNormalVelocity, s, F_TENDU, 1; upload = true)
        bind!(Tend.tendLayerThickness, s, F_TENDH, 1; upload = true)
    end
    flush_host_writes!(s)
    s
end

"upload what the host changed through setindex! / copyto! since the last device call"
function flush_host_writes!(s::State)
    for w in s.bound
        a = w.value
        (a === nothing || !a.host_dirty) && continue
        check(ccall((:moka_state_upload, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}), s.handle, a.field, a.level, a.host), s.backend.ctx)
        a.host_dirty = false
        a.host_version = s.version
    end
end
device_changed!(s::State) = (s.version += 1; nothing)

# ---- forward model --------------------------------------------------------------------------------
const REFERENCE_COMPAT = Int32(7)        # MOKA_FE_STALE_HEDGE | ACCUM_VORT | LEVEL1_ONLY: the live reference step, quirks included

# ocn_timestep(timestep, Prog, Diag, Tend, S, ForwardEuler; backend)   time_integration.jl:150
# More specific than the reference's method in the Prog / Diag / Tend array type, so dispatch picks it for MArrays.
function MOKA.ocn_timestep(timestep, Prog::MProg, Diag::MDiag, Tend::MTend, S::ModelSetup, ::Type{ForwardEuler}; backend = Prog.ssh[end].backend)
    s = state_of(Prog, Diag, Tend, S, backend)
    dt = Float64(timestep[1])             # the reference's 1-element array on the backend (mpas_ocean.jl:36-37): never bound
    check(ccall((:moka_step_fe, lib), Cint, (Ptr{Cvoid}, Cdouble, Cint), s.handle, dt, REFERENCE_COMPAT), backend.ctx)
    device_changed!(s)
end
# ocn_timestep(Prog, Diag, Tend, S, RungeKutta4; backend)              time_integration.jl:61 (dead code there: the spec)
function MOKA.ocn_timestep(Prog::MProg, Diag::MDiag, Tend::MTend, S::ModelSetup, ::Type{RungeKutta4}; backend = Prog.ssh[end].backend)
    s = state_of(Prog, Diag, Tend, S, backend)
    dt = convert(Float64, Dates.value(Dates.Second(S.timeManager.timeStep)))   # :75
    check(ccall((:moka_step_rk4, lib), Cint, (Ptr{Cvoid}, Cdouble), s.handle, dt), backend.ctx)
    device_changed!(s)
end
# diagnostic_compute!(Mesh, Diag, Prog; backend)                       DiagnosticVars.jl:108
function MOKA.diagnostic_compute!(m::Mesh, Diag::MDiag, Prog::MProg; backend = Prog.ssh[end].backend)
    s = state_of(Prog, Diag, nothing, ModelSetup(nothing, m, nothing), backend)
    check(ccall((:moka_diagnostic_compute, lib), Cint, (Ptr{Cvoid}, Cint), s.handle, REFERENCE_COMPAT), backend.ctx)
    device_changed!(s)
end
# computeNormalVelocityTendency! / computeLayerThicknessTendency!      normalVelocity.jl:21, layerThickness.jl:14
function MOKA.computeNormalVelocityTendency!(Tend::MTend, Prog::MProg, Diag::MDiag, m::Mesh, Config; backend = Prog.ssh[end].backend)
    s = state_of(Prog, Diag, Tend, ModelSetup(Config, m, nothing), backend)
    check(ccall((:moka_compute_normal_velocity_tendency, lib), Cint, (Ptr{Cvoid}, Cint), s.handle, REFERENCE_COMPAT), backend.ctx)
    device_changed!(s)
end
function MOKA.computeLayerThicknessTendency!(Tend::MTend, Prog::MProg, Diag::MDiag, m::Mesh, Config; backend = Prog.ssh[end].backend)
    s = state_of(Prog, Diag, Tend, ModelSetup(Config, m, nothing), backend)
    check(ccall((:moka_compute_layer_thickness_tendency, lib), Cint, (Ptr{Cvoid}, Cint), s.handle, REFERENCE_COMPAT), backend.ctx)
    device_changed!(s)
end
# advanceTimeLevels!(Prog; backend)                                    time_integration.jl:10
function MOKA.advanceTimeLevels!(Prog::MProg; backend = Prog.ssh[end].backend)
    s = Prog.ssh[end].state
    s === nothing && return (copyto!(Prog.ssh[1].host, Prog.ssh[2].host); copyto!(Prog.normalVelocity[1].host, Prog.normalVelocity[2].host);
                             copyto!(Prog.layerThickness[1].host, Prog.layerThickness[2].host); nothing)
    flush_host_writes!(s)
    check(ccall((:moka_advance_time_levels, lib), Cint, (Ptr{Cvoid}, Cint), s.handle, 0), backend.ctx)
    device_changed!(s)
end
# ocn_run_loop (run_loop.jl:8-22) needs no method: it only calls advance!, ocn_timestep, isRinging, reset!.
# Its (sumCPU, sumGPU, ...) form (run_loop.jl:26-45) launches the one-thread KA kernel sumArray on the backend: here the
# same strictly serial sum is a library call, placed where the reference puts it.
function MOKA.ocn_run_loop(sumCPU, sumGPU::MArray, timestep, Prog::MProg, Diag::MDiag, Tend::MTend, Setup, ::Type{ForwardEuler},
                           clock, simulationAlarm, outputAlarm; backend = Prog.ssh[end].backend)
    MOKA.ocn_run_loop(timestep, Prog, Diag, Tend, Setup, ForwardEuler, clock, simulationAlarm, outputAlarm; backend = backend)
    s = state_of(Prog, Diag, Tend, Setup, backend)
    out = Ref{Float64}(0.0)
    check(ccall((:moka_sum_sq, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Float64}), s.handle, F_SSH, 1, out), backend.ctx)
    sumGPU[1] = sumGPU[1] + out[]              # sumGPU[1] = sumGPU[1] + array[j]^2 ... (run_loop.jl:47-51)
    MOKA.mycopyto!(sumCPU, sumGPU)
    return sumCPU[1]
end

# ---- beyond the reference's GPU path (all optional) ------------------------------------------------
# nonlinear terms (potential-vorticity Coriolis + kinetic-energy gradient); the mesh descriptor must carry
# kiteAreasOnVertex / fVertex.  Off by default: the reference has only the linear terms.
function set_nonlinear!(Prog::MProg, on::Bool = true)
    s = Prog.ssh[end].state
    s === nothing && error("MokaHIP: take one step (or call diagnostic_compute!) first: the model is not on the device yet")
    check(ccall((:moka_set_nonlinear, lib), Cint, (Ptr{Cvoid}, Cint), s.handle, on ? 1 : 0), s.backend.ctx)
end
# Del2 momentum mixing on top of them (what horizontal_momentum_mixing.jl:53-80 sketches); 0 = off
function set_viscosity_del2!(Prog::MProg, viscDel2::Float64)
    s = Prog.ssh[end].state
    s === nothing && error("MokaHIP: the model is not on the device yet")
    check(ccall((:moka_set_viscosity_del2, lib), Cint, (Ptr{Cvoid}, Cdouble), s.handle, viscDel2), s.backend.ctx)
end

# The opt-in 13-stream form of the RK4 step (moka_set_tuning key 7; include/moka_hip.h): the same Runge-Kutta step with another
# round-off than time_integration.jl:134-135's running sum (<= 1e-12 relative per step), 12 % fewer bytes.  Process-wide;
# rk4_streams tells which form the next step of a bound model takes (13, or 16 = the reference's).
function set_rk4_13_streams!(on::Bool = true)
    rc = ccall((:moka_set_tuning, lib), Cint, (Cint, Cint), 7, on ? 1 : 0)
    rc == 0 || error("MokaHIP: moka_set_tuning(7) failed")
    nothing
end
function rk4_streams(Prog::MProg)
    s = Prog.ssh[end].state
    s === nothing && error("MokaHIP: the model is not on the device yet")
    Int(ccall((:moka_state_rk4_streams, lib), Cint, (Ptr{Cvoid},), s.handle))
end

# reverse mode: the hand-written adjoint of the step loop (what Enzyme differentiates in the reference:
# test/enzyme/test_Enzyme_end2end.jl).  MokaHIPEnzymeExt.jl registers it as the EnzymeRules rule of ocn_run_loop.
mutable struct Tape
    handle::Ptr{Cvoid}
    state::State
end
function Tape(s::State, capacity::Integer)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:moka_tape_create, lib), Cint, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), s.handle, capacity, ref), s.backend.ctx)
    t = Tape(ref[], s)                 # holds the state (hence mesh and context count) alive
    retain!(s.backend)
    s.tapes += 1
    finalizer(t) do x
        ccall((:moka_tape_destroy, lib), Cvoid, (Ptr{Cvoid},), x.handle)      # the state is still there: it counts its tapes
        x.state.tapes -= 1
        x.state.dead && x.state.tapes == 0 && destroy_state!(x.state)
        release!(x.state.backend)
    end
    t
end
function step_fe!(t::Tape, dt; flags = REFERENCE_COMPAT)
    check(ccall((:moka_step_fe_taped, lib), Cint, (Ptr{Cvoid}, Cdouble, Cint), t.handle, dt, flags), t.state.backend.ctx)
    device_changed!(t.state)
end
function step_rk4!(t::Tape, dt)
    check(ccall((:moka_step_rk4_taped, lib), Cint, (Ptr{Cvoid}, Cdouble), t.handle, dt), t.state.backend.ctx)
    device_changed!(t.state)
end
"d sum(ssh^2) / d (initial ssh, normalVelocity, layerThickness) into host arrays: `d_Prog` of the reference test"
function gradient!(t::Tape, d_ssh::Vector{Float64}, d_u::Matrix{Float64}, d_h::Matrix{Float64})
    ctx = t.state.backend.ctx
    check(ccall((:moka_adjoint_seed_sum_sq_ssh, lib), Cint, (Ptr{Cvoid},), t.handle), ctx)
    check(ccall((:moka_adjoint_sweep, lib), Cint, (Ptr{Cvoid},), t.handle), ctx)
    for (f, a) in ((F_SSH, d_ssh), (F_U, d_u), (F_H, d_h))
        check(ccall((:moka_adjoint_download, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), t.handle, f, a), ctx)
    end
end

end # module
