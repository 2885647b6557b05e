# MokaHIPEnzymeExt.jl -- reverse-mode rule for the step loop on the MokaHIP backend.
#
# STATUS: written against ext/MPASEnzymeExt.jl and test/enzyme/test_Enzyme_end2end.jl, NEVER EXECUTED (no Julia here).
# The arithmetic it delegates to (moka_step_fe_taped / moka_adjoint_*) IS tested: tests/test_gpu_parity.py::
# test_fe_adjoint_bitwise compares it bit for bit with the oracle's adjoint, test_fe_adjoint_central_difference_through_the_c_abi
# repeats the reference's own check (AD vs central differences of sum(ssh^2), test_Enzyme_end2end.jl:98-181).
#
# The reference lets Enzyme differentiate `ocn_run_loop(sumCPU, sumGPU, timestep, Prog, Diag, Tend, Setup, ForwardEuler,
# clock, simulationAlarm, outputAlarm; backend)` (run_loop.jl:26-45) and registers one hand-written rule, for mycopyto!
# (ext/MPASEnzymeExt.jl:13-38).  On this backend the kernels are not Julia code Enzyme could see, so the whole loop gets
# the rule: the augmented primal runs the loop with every step recorded on a library tape, the reverse pass sweeps the
# tape and adds d sum(ssh^2) / d(initial state) -- scaled by the incoming adjoint of the returned sum -- to the shadows
# `d_Prog` the caller passes as Duplicated (test_Enzyme_end2end.jl:55-60,78-96).
module MokaHIPEnzymeExt

using MOKA
using MOKA: ForwardEuler, isRinging, advance!, reset!
using MokaHIP
using MokaHIP: MProg, MDiag, MTend, MArray, Tape, state_of, step_fe!, gradient!
using Enzyme
using Enzyme: EnzymeCore
using Enzyme: EnzymeCore.EnzymeRules
import Dates

"steps the loop will take: the clock is advanced by the primal itself, so count on a copy"
function count_steps(clock, simulationAlarm)
    c = deepcopy(clock)
    n = 0
    a = c.alarms[simulationAlarm.name]
    while !isRinging(a)
        advance!(c)
        n += 1
    end
    n
end

function EnzymeRules.augmented_primal(config, func::Const{typeof(MOKA.ocn_run_loop)}, ::Type{RT},
                                      sumCPU::Annotation, sumGPU::Annotation{<:MArray}, timestep::Annotation,
                                      Prog::Annotation{<:MProg}, Diag::Annotation{<:MDiag}, Tend::Annotation{<:MTend},
                                      Setup::Annotation, fe::Annotation{Type{ForwardEuler}}, clock::Annotation,
                                      simulationAlarm::Annotation, outputAlarm::Annotation; backend) where {RT}
    s = state_of(Prog.val, Diag.val, Tend.val, Setup.val, backend)
    tape = Tape(s, count_steps(clock.val, simulationAlarm.val))
    dt = Float64(timestep.val[1])
    while !isRinging(simulationAlarm.val)                         # run_loop.jl:30-38, with taped steps
        advance!(clock.val)
        step_fe!(tape, dt)
        isRinging(outputAlarm.val) && reset!(outputAlarm.val)
    end
    out = Ref{Float64}(0.0)
    MokaHIP.check(ccall((:moka_sum_sq, MokaHIP.lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Float64}), s.handle, MokaHIP.F_SSH, 1, out), backend.ctx)
    sumGPU.val[1] = sumGPU.val[1] + out[]
    MOKA.mycopyto!(sumCPU.val, sumGPU.val)
    primal = EnzymeRules.needs_primal(config) ? sumCPU.val[1] : nothing
    return EnzymeRules.AugmentedReturn(primal, nothing, tape)
end

function EnzymeRules.reverse(config, func::Const{typeof(MOKA.ocn_run_loop)}, dret, tape,
                             sumCPU::Annotation, sumGPU::Annotation{<:MArray}, timestep::Annotation,
                             Prog::Annotation{<:MProg}, Diag::Annotation, Tend::Annotation, Setup::Annotation, fe::Annotation,
                             clock::Annotation, simulationAlarm::Annotation, outputAlarm::Annotation; backend)
    seed = dret isa Active ? dret.val : 1.0                       # adjoint of the returned sum (Active return: autodiff(Reverse, ...))
    dP = Prog.dval
    nC = length(dP.ssh[1]); K, nE = size(dP.normalVelocity[1])
    g_ssh, g_u, g_h = zeros(nC), zeros(K, nE), zeros(K, nC)
    gradient!(tape, g_ssh, g_u, g_h)                              # d sum(ssh^2) / d state the loop started from
    # the loop started from the CURRENT level (index `end`); the reference's AD accumulates there too (test:93-94)
    copyto!(dP.ssh[end], Array(dP.ssh[end]) .+ seed .* g_ssh)
    copyto!(dP.normalVelocity[end], Array(dP.normalVelocity[end]) .+ seed .* g_u)
    copyto!(dP.layerThickness[end], Array(dP.layerThickness[end]) .+ seed .* g_h)
    return ntuple(_ -> nothing, 11)
end

# ---------------------------------------------------------------------------------------------------------------------
# The stand-alone operators (test/enzyme/test_Enzyme_Operators.jl:42-131, 137-225).  The reference lets Enzyme differentiate
#     gradient_test(grad, h, mesh, backend)       = GradientOnEdge!(grad, h, mesh; backend)
#     divergence_test(div, F, temp, mesh, backend) = DivergenceOnCell!(div, F, temp, mesh; backend, nthreads = 64)
# with Duplicated arrays and a Duplicated mesh (whose shadow stays zero: the mesh is not differentiated through here).
# On this backend the operators are library calls, so they get rules.  They are LINEAR in their array argument:
#   forward : the tangent of the output is the operator applied to the tangent of the input (moka_*_jvp);
#   reverse : the augmented primal runs the operator and needs no tape; the reverse pass adds J^T (shadow of the output) to
#             the shadow of the input and zeroes the shadow of the overwritten output (moka_*_vjp) -- curl accumulates into its
#             output, so that shadow stays.
# STATUS: NEVER EXECUTED (no Julia in this pipeline).  What they delegate to is tested through the C ABI with the reference's
# own check (AD vs central differences, eps = 1e-8, atol = 1e-6): tests/test_gpu_parity.py::
# test_operator_reverse_and_forward_mode_against_central_differences and ..._transposes_bitwise_against_the_oracle.
# ---------------------------------------------------------------------------------------------------------------------
using MokaHIP: gradient_vjp!, gradient_jvp!, divergence_vjp!, divergence_jvp!, curl_vjp!, curl_jvp!

const DupArr = Union{Duplicated{<:MArray{Float64,2}},DuplicatedNoNeed{<:MArray{Float64,2}}}

# ---- GradientOnEdge!(grad, h, mesh; backend, workgroupsize) ----
function EnzymeRules.forward(config, func::Const{typeof(MOKA.GradientOnEdge!)}, ::Type{RT}, grad::DupArr, h::DupArr, mesh::Annotation;
                             backend = grad.val.backend, workgroupsize = 64) where {RT}
    MOKA.GradientOnEdge!(grad.val, h.val, mesh.val; backend, workgroupsize)
    gradient_jvp!(grad.dval, h.dval, mesh.val, backend)
    return nothing
end
function EnzymeRules.augmented_primal(config, func::Const{typeof(MOKA.GradientOnEdge!)}, ::Type{RT}, grad::DupArr, h::DupArr,
                                      mesh::Annotation; backend = grad.val.backend, workgroupsize = 64) where {RT}
    MOKA.GradientOnEdge!(grad.val, h.val, mesh.val; backend, workgroupsize)
    return EnzymeRules.AugmentedReturn(nothing, nothing, nothing)             # linear: nothing to remember
end
function EnzymeRules.reverse(config, func::Const{typeof(MOKA.GradientOnEdge!)}, dret, tape, grad::DupArr, h::DupArr, mesh::Annotation;
                             backend = grad.val.backend, workgroupsize = 64)
    gradient_vjp!(grad.dval, h.dval, mesh.val, backend)                       # d_h += J^T d_grad; d_grad = 0
    return (nothing, nothing, nothing)
end

# ---- DivergenceOnCell!(div, V, temp, mesh; backend, nthreads) ----
function EnzymeRules.forward(config, func::Const{typeof(MOKA.DivergenceOnCell!)}, ::Type{RT}, div::DupArr, V::DupArr, temp::DupArr,
                             mesh::Annotation; backend = div.val.backend, nthreads = 50) where {RT}
    MOKA.DivergenceOnCell!(div.val, V.val, temp.val, mesh.val; backend, nthreads)
    divergence_jvp!(div.dval, V.dval, temp.dval, mesh.val, backend)
    return nothing
end
function EnzymeRules.augmented_primal(config, func::Const{typeof(MOKA.DivergenceOnCell!)}, ::Type{RT}, div::DupArr, V::DupArr,
                                      temp::DupArr, mesh::Annotation; backend = div.val.backend, nthreads = 50) where {RT}
    MOKA.DivergenceOnCell!(div.val, V.val, temp.val, mesh.val; backend, nthreads)
    return EnzymeRules.AugmentedReturn(nothing, nothing, nothing)
end
function EnzymeRules.reverse(config, func::Const{typeof(MOKA.DivergenceOnCell!)}, dret, tape, div::DupArr, V::DupArr, temp::DupArr,
                             mesh::Annotation; backend = div.val.backend, nthreads = 50)
    divergence_vjp!(div.dval, V.dval, temp.dval, mesh.val, backend)           # d_V += P1^T (d_temp + P2^T d_div); d_div = d_temp = 0
    return (nothing, nothing, nothing, nothing)
end

# ---- CurlOnVertex!(curl, V, mesh; backend): accumulates into curl (Operators.jl:142) ----
function EnzymeRules.forward(config, func::Const{typeof(MOKA.CurlOnVertex!)}, ::Type{RT}, curl::DupArr, V::DupArr, mesh::Annotation;
                             backend = curl.val.backend) where {RT}
    MOKA.CurlOnVertex!(curl.val, V.val, mesh.val; backend)
    curl_jvp!(curl.dval, V.dval, mesh.val, backend)                           # d_curl += J d_V
    return nothing
end
function EnzymeRules.augmented_primal(config, func::Const{typeof(MOKA.CurlOnVertex!)}, ::Type{RT}, curl::DupArr, V::DupArr,
                                      mesh::Annotation; backend = curl.val.backend) where {RT}
    MOKA.CurlOnVertex!(curl.val, V.val, mesh.val; backend)
    return EnzymeRules.AugmentedReturn(nothing, nothing, nothing)
end
function EnzymeRules.reverse(config, func::Const{typeof(MOKA.CurlOnVertex!)}, dret, tape, curl::DupArr, V::DupArr, mesh::Annotation;
                             backend = curl.val.backend)
    curl_vjp!(curl.dval, V.dval, mesh.val, backend)                           # d_V += J^T d_curl; d_curl stays
    return (nothing, nothing, nothing)
end

end # module
