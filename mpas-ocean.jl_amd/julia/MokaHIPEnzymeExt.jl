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

end # module
