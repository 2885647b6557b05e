// stage_fallback.hip -- the two fallback shapes of the fused tendency / RK-stage kernel (gfx950) that the product library
// carries next to the default k_stage_rec2c / k_stage_rec2c_f32 (kernels.hip): the generic index kernel k_stage (any
// nVertLevels: K <= 32, odd K, K > 128) and the plain column kernel k_stage_col (one wavefront per entity, lane = level:
// even or odd K from 33 up, several sweeps beyond 64).  Same arithmetic and bit-identical results as the default kernel.
// The other measured design points of rounds 1-3 (profiles/r0*_variants.txt) lost and were removed in round 4; they were only built
// with `make VARIANTS=1`.
#include "kernels_common.hpp"

namespace moka {

// ------------------------------------------------------------------------------------------------
// Fused tendency / RK-stage kernel.
//   cells : hEdge (K5, Operators.jl:217) -> thicknessFlux (K7, DiagnosticVars.jl:165)
//           -> flux divergence (K8, horizontal_advection.jl:60-66) [-> state update, ssh (K14)]
//   edges : -g grad ssh (K9, pressure_gradient.jl:58-64) + Coriolis (K10,
//           horizontal_advection_and_coriolis.jl:61-73)          [-> state update]
// The stage update is the RK4 specification of time_integration.jl:112-137:
//   Provis' = Curr + a*tend ; New = New + b*tend ; ssh from layerThickness.
// ------------------------------------------------------------------------------------------------
template <int LPC, int ME, int ME2>
__global__ __launch_bounds__(BLOCK) void k_stage(const MeshDev m, const StageArgs a)
{
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC);
    const int l = threadIdx.x % LPC;
    const int K = m.K;
    const int Kc = ((K + LPC - 1) / LPC) * LPC;

    // ---------------- cells ----------------
    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
    for (int c = c0 + grp; c < c1; c += NG) {
        CP<int32_t> re = cptr(m.eoc) + (size_t)c * ME;
        CP<int32_t> rc = cptr(m.coc) + (size_t)c * ME;
        CP<int32_t> rm = cptr(m.mltc) + (size_t)c * ME;
        CP<double> rs = cptr(m.sdv) + (size_t)c * ME;
        const double invA = cptr(m.invArea)[c];
        int ei[ME], ci[ME], mi[ME];
        double si[ME];
#pragma unroll
        for (int i = 0; i < ME; ++i) {
            ei[i] = re[i];
            ci[i] = rc[i];
            mi[i] = rm[i];
            si[i] = rs[i];
        }
        double sshAcc = 0.0;
        bool first = true;
        for (int k = l; k < Kc; k += LPC) {
            const bool act = k < K;
            const size_t off = (size_t)c * K + k;
            double hc = 0.0, uv[ME], hv[ME];
            if (act) {
                hc = a.ph[off];
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    const int es = ei[i] >= 0 ? ei[i] : ei[0];
                    const int cs = ci[i] >= 0 ? ci[i] : c;
                    uv[i] = a.pu[(size_t)es * K + k];
                    hv[i] = a.ph[(size_t)cs * K + k];
                }
            }
            double t = 0.0;
            if (act) {
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    if (ei[i] >= 0 && k < mi[i]) {
                        const double hE = 0.5 * (hc + hv[i]);      // Operators.jl:217
                        const double F = uv[i] * hE;               // DiagnosticVars.jl:165
                        t += F * si[i] * invA;                     // horizontal_advection.jl:63-64
                    }
                }
            }
            double hs = 0.0;   // the thickness whose column sum gives ssh_out
            if (act) {
                if (a.tendH) a.tendH[off] = t;
                const double hcur = a.ch ? a.ch[off] : hc;
                if (a.ph_out) {
                    const double hp = hcur + a.a * t;              // time_integration.jl:125
                    a.ph_out[off] = hp;
                    hs = hp;
                }
                if (a.nh_out) {
                    const double hn = (a.nh_in ? a.nh_in[off] : hcur) + a.b * t;   // :135
                    a.nh_out[off] = hn;
                    if (!a.ph_out) hs = hn;
                }
            }
            sshAcc = first ? hs : sshAcc + hs;                     // oracle_ksum strided partials
            first = false;
        }
        if (a.ssh_out) {
            const double s = group_sum<LPC>(sshAcc);
            if (l == 0) a.ssh_out[c] = s - cptr(m.rsum)[c];        // time_integration.jl:209 (+N3)
        }
    }

    // ---------------- edges ----------------
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
    for (int e = e0 + grp; e < e1; e += NG) {
        CP<int32_t> rh = cptr(m.ehdr) + (size_t)e * 4;
        const int4 hdr = make_int4(rh[0], rh[1], rh[2], rh[3]);
        CP<int32_t> re = cptr(m.eoe) + (size_t)e * ME2;
        CP<double> rw = cptr(m.woe) + (size_t)e * ME2;
        int xi[ME2];
        double wi[ME2], fi[ME2];
#pragma unroll
        for (int i = 0; i < ME2; ++i) {
            xi[i] = re[i];
            wi[i] = rw[i];
        }
#pragma unroll
        for (int i = 0; i < ME2; ++i) fi[i] = cptr(m.fEdge)[xi[i] >= 0 ? xi[i] : e];
        const double g = cptr(m.gInvDc)[e];
        // ssh was written by the previous launch and is never written by this one (ssh_out is another buffer)
        const double ds = cptr(a.ssh)[hdr.y] - cptr(a.ssh)[hdr.x];   // ssh[c2] - ssh[c1]
        const int mlt = hdr.w;
        for (int k = l; k < K; k += LPC) {
            const size_t off = (size_t)e * K + k;
            double uv[ME2];
#pragma unroll
            for (int i = 0; i < ME2; ++i) uv[i] = a.pu[(size_t)(xi[i] >= 0 ? xi[i] : e) * K + k];
            double t = 0.0;
            if (k < mlt) {
                t -= g * ds;                                       // pressure_gradient.jl:63
#pragma unroll
                for (int i = 0; i < ME2; ++i)
                    if (xi[i] >= 0) t += wi[i] * uv[i] * fi[i];    // ...coriolis.jl:70-72
            }
            if (a.tendU) a.tendU[off] = t;
            const double ucur = a.cu ? a.cu[off] : a.pu[off];
            if (a.pu_out) a.pu_out[off] = ucur + a.a * t;          // time_integration.jl:124
            if (a.nu_out) a.nu_out[off] = (a.nu_in ? a.nu_in[off] : ucur) + a.b * t;   // :134
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Column kernel, instruction-lean form for LPC = 64 (one wavefront per entity, lane = level).
//
// rocprof on the generic k_stage showed the SIMDs busy *issuing* ~150 VALU + ~150 SALU per entity
// (64-bit index*K*8 address arithmetic, selects, SGPR spills) for ~40 essential fp64 operations,
// and the per-CU scalar unit saturated.  Here the plan stores 32-bit BYTE offsets of every
// neighbour row (cRec / eRec), all of an entity's connectivity arrives in SGPRs with two or three
// s_load_dwordx8/x16, and every gather is `buffer_load_dwordx2 v, v_lane8, s[rsrc], s_off offen`:
// no address arithmetic at all.  Weights, fEdge and metric factors are SGPR operands of the fp64
// instructions.  Slot validity and "all levels active" are wave-uniform (scalar branches).
// Lanes >= K read past the row (the buffer range check returns 0 past the array) and never store.
// ------------------------------------------------------------------------------------------------
// own-row access: base pointer (SGPR pair) + 32-bit byte offset (VGPR) -> global_load/store saddr form
template <int ME, int ME2>
__global__ __launch_bounds__(BLOCK) void k_stage_col(const ColMesh m, const StageArgs a)
{
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    const int K = m.K;
    const uint32_t rowB = (uint32_t)K * 8u;
    const rsrc_t ph = make_rsrc(a.ph, (uint32_t)m.nC * rowB), pu = make_rsrc(a.pu, (uint32_t)m.nE * rowB);

    // ---------------- cells ----------------
    {
        const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
        for (int c = c0 + wave; c < c1; c += BLOCK / 64) {
            CP<uint32_t> r = cptr(m.cRec) + (size_t)c * m.CI;
            CP<double> rs = cptr(m.sdv) + (size_t)c * ME;
            uint32_t eo[ME], co[ME];
            double sd[ME];
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                eo[i] = r[i];
                co[i] = r[ME + i];
                sd[i] = rs[i];
            }
            const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
            const double invA = cptr(m.invArea)[c];
            const uint32_t own = (uint32_t)c * rowB;
            double sshAcc = 0.0;
            for (int kb = 0; kb < K; kb += 64) {
                const int k = kb + l, voff = k * 8;
                const uint32_t ooff = own + (uint32_t)voff;
                const double hc = bload(ph, voff, own);
                double uv[ME], hv[ME];
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    uv[i] = bload(pu, voff, eo[i]);
                    hv[i] = bload(ph, voff, co[i]);
                }
                double t = 0.0;
                if (all) {
#pragma unroll
                    for (int i = 0; i < ME; ++i)
                        if ((mask >> i) & 1u) t += uv[i] * (0.5 * (hc + hv[i])) * sd[i] * invA;   // Operators.jl:217,
                } else {                                                                          // DiagnosticVars.jl:165,
#pragma unroll
                    for (int i = 0; i < ME; ++i)                                                   // horizontal_advection.jl:63
                        if (((mask >> i) & 1u) && k < cptr(m.mltc)[(size_t)c * ME + i])
                            t += uv[i] * (0.5 * (hc + hv[i])) * sd[i] * invA;
                }
                double hs = 0.0;
                if (k < K) {
                    if (a.tendH) gstore(a.tendH, ooff, t);
                    double hcur = hc;
                    if (a.ch) hcur = gload(a.ch, ooff);
                    if (a.ph_out) {
                        hs = hcur + a.a * t;                           // time_integration.jl:125
                        gstore(a.ph_out, ooff, hs);
                    }
                    if (a.nh_out) {
                        double nb = hcur;
                        if (a.nh_in) nb = gload(a.nh_in, ooff);
                        const double hn = nb + a.b * t;                // :135
                        gstore(a.nh_out, ooff, hn);
                        if (!a.ph_out) hs = hn;
                    }
                }
                if (kb == 0) sshAcc = hs;
                else sshAcc = sshAcc + hs;
            }
            if (a.ssh_out) {
                const double sum = group_sum<64>(sshAcc);
                if (l == 0) a.ssh_out[c] = sum - cptr(m.rsum)[c];      // time_integration.jl:209 (+N3)
            }
        }
    }

    // ---------------- edges ----------------
    {
        const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
        for (int e = e0 + wave; e < e1; e += BLOCK / 64) {
            const size_t er = (size_t)e;
            CP<uint32_t> r = cptr(m.eRec) + er * m.EI;
            CP<double> rw = cptr(m.woe) + er * ME2;
            CP<double> rf = cptr(m.feoe) + er * ME2;
            uint32_t xo[ME2];
            double wi[ME2], fi[ME2];
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                xo[i] = r[i];
                wi[i] = rw[i];
                fi[i] = rf[i];
            }
            const uint32_t cA = r[ME2], cB = r[ME2 + 1], mask = r[ME2 + 2];
            const int mlt = (int)r[ME2 + 3];
            const double g = cptr(m.gInvDc)[e];
            const double ds = cptr(a.ssh)[cB] - cptr(a.ssh)[cA];       // ssh[c2] - ssh[c1]
            const uint32_t own = (uint32_t)e * rowB;
            for (int kb = 0; kb < K; kb += 64) {
                const int k = kb + l, voff = k * 8;
                const uint32_t ooff = own + (uint32_t)voff;
                double uv[ME2];
#pragma unroll
                for (int i = 0; i < ME2; ++i) uv[i] = bload(pu, voff, xo[i]);
                double t = 0.0;
                if (k < mlt) {
                    t -= g * ds;                                       // pressure_gradient.jl:63
#pragma unroll
                    for (int i = 0; i < ME2; ++i)
                        if ((mask >> i) & 1u) t += wi[i] * uv[i] * fi[i];   // ...coriolis.jl:70-72
                }
                if (k < K) {
                    if (a.tendU) gstore(a.tendU, ooff, t);
                    const double ucur = gload(a.cu ? a.cu : a.pu, ooff);
                    if (a.pu_out) gstore(a.pu_out, ooff, ucur + a.a * t);   // time_integration.jl:124
                    if (a.nu_out) {
                        double nb = ucur;
                        if (a.nu_in) nb = gload(a.nu_in, ooff);
                        gstore(a.nu_out, ooff, nb + a.b * t);          // :134
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <int LPC>
static hipError_t launch_stage_lpc(const MeshDev &m, const StageArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(m)), b(BLOCK);
    if (m.ME == 6 && m.ME2 == 10) hipLaunchKernelGGL((k_stage<LPC, 6, 10>), g, b, 0, s, m, a);
    else if (m.ME == 8 && m.ME2 == 14) hipLaunchKernelGGL((k_stage<LPC, 8, 14>), g, b, 0, s, m, a);
    else if (m.ME <= 6 && m.ME2 <= 14) hipLaunchKernelGGL((k_stage<LPC, 6, 14>), g, b, 0, s, m, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_stage(const MeshDev &m, const StageArgs &a, int lpc, hipStream_t s)
{
#define CALL(L) launch_stage_lpc<L>(m, a, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

hipError_t launch_stage_col(const MeshDev &md, const StageArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(md)), b(BLOCK);
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc};
    if (md.ME == 6 && md.ME2 == 10) hipLaunchKernelGGL((k_stage_col<6, 10>), g, b, 0, s, m, a);
    else if (md.ME == 8 && md.ME2 == 14) hipLaunchKernelGGL((k_stage_col<8, 14>), g, b, 0, s, m, a);
    else if (md.ME <= 6 && md.ME2 <= 14) hipLaunchKernelGGL((k_stage_col<6, 14>), g, b, 0, s, m, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace moka
