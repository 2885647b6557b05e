// nonlinear.hip -- kernels of the optional nonlinear terms of libmoka_hip (gfx950).
#include "kernels_common.hpp"

namespace moka {

// ------------------------------------------------------------------------------------------------
// Optional nonlinear terms (moka_set_nonlinear; NOT in the reference, see oracle_tendencies_nonlinear for the
// algebra and the operand order these kernels reproduce bit for bit).  Generic column kernels: LPC lanes span a
// column, one entity per lane group.  Three preparation passes over the whole mesh, then the stage kernel.
// ------------------------------------------------------------------------------------------------
template <int LPC>
__global__ __launch_bounds__(BLOCK) void k_nl_vertex(const MeshDev m, const double *u, const double *h, double *qv, double *zv)
{
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC), l = threadIdx.x % LPC;   // LPC = 64: records come through scalar loads
    const int K = m.K, VD = m.VD;
    for (int v = blockIdx.x * NG + grp; v < m.nV; v += gridDim.x * NG) {
        const double invA = cptr(m.invAreaTri)[v], fv = cptr(m.fVertex)[v];
        for (int k = l; k < K; k += LPC) {
            double zeta = 0.0, hv = 0.0;
            for (int j = 0; j < VD; ++j) {
                zeta += cptr(m.cv)[(size_t)v * VD + j] * u[(size_t)cptr(m.eov)[(size_t)v * VD + j] * K + k];   // (dc*invA*sign)*u, sign = +-1
                hv += cptr(m.kite)[(size_t)v * VD + j] * h[(size_t)cptr(m.cov)[(size_t)v * VD + j] * K + k];
            }
            hv = hv * invA;
            qv[(size_t)v * K + k] = (fv + zeta) / hv;
            if (zv) zv[(size_t)v * K + k] = zeta;                                                      // relativeVorticity, Operators.jl:137-146
        }
    }
}

// F and q_e are stored interleaved, (K, nE) pairs {F, q_e}: the stage kernel fetches both of a neighbour edge in one 16-byte load
template <int LPC>
__global__ __launch_bounds__(BLOCK) void k_nl_edge(const MeshDev m, const double *u, const double *h, const NlArgs nl)
{
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC), l = threadIdx.x % LPC;
    const int K = m.K;
    double2 *fq = reinterpret_cast<double2 *>(nl.fq);
    for (int e = blockIdx.x * NG + grp; e < m.nE; e += gridDim.x * NG) {
        const int c1 = cptr(m.ehdr)[(size_t)e * 4], c2 = cptr(m.ehdr)[(size_t)e * 4 + 1];
        const int v1 = cptr(m.voe)[(size_t)e * 2], v2 = cptr(m.voe)[(size_t)e * 2 + 1];
        for (int k = l; k < K; k += LPC) {
            const size_t off = (size_t)e * K + k;
            const double hE = 0.5 * (h[(size_t)c1 * K + k] + h[(size_t)c2 * K + k]);      // Operators.jl:217
            fq[off] = make_double2(u[off] * hE,                                           // DiagnosticVars.jl:165
                                   0.5 * (nl.qv[(size_t)v1 * K + k] + nl.qv[(size_t)v2 * K + k]));
        }
    }
}

template <int LPC>
__global__ __launch_bounds__(BLOCK) void k_nl_cell(const MeshDev m, const double *u, double *ke, double *divc)
{
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC), l = threadIdx.x % LPC;
    const int K = m.K, ME = m.ME;
    for (int c = blockIdx.x * NG + grp; c < m.nC; c += gridDim.x * NG) {
        const double invA = cptr(m.invArea)[c], area = cptr(m.areaCell)[c];
        for (int k = l; k < K; k += LPC) {
            double acc = 0.0, d = 0.0;
            for (int i = 0; i < ME; ++i) {
                const int e = cptr(m.eoc)[(size_t)c * ME + i];
                if (e < 0) continue;
                const double ue = u[(size_t)e * K + k];
                acc += cptr(m.keCoef)[e] * ue * ue;
                d -= ue * cptr(m.sdv)[(size_t)c * ME + i];                                              // (u*dvEdge)*sign, Operators.jl:18,36
            }
            ke[(size_t)c * K + k] = acc * invA;
            if (divc) divc[(size_t)c * K + k] = d / area;                                               // velocityDivCell, Operators.jl:41
        }
    }
}

// the stage kernel with the nonlinear velocity tendency; the thickness part is that of k_stage
template <int LPC>
__global__ __launch_bounds__(BLOCK) void k_stage_nl(const MeshDev m, const StageArgs a, const NlArgs nl)
{
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC), l = threadIdx.x % LPC;
    const int K = m.K, ME = m.ME, ME2 = m.ME2;
    const int Kc = ((K + LPC - 1) / LPC) * LPC;
    const double2 *fq = reinterpret_cast<const double2 *>(nl.fq);
    for (int c = blockIdx.x * NG + grp; c < m.nC; c += gridDim.x * NG) {
        const double invA = cptr(m.invArea)[c];
        double sshAcc = 0.0;
        bool first = true;
        for (int k = l; k < Kc; k += LPC) {
            const bool act = k < K;
            const size_t off = (size_t)c * K + k;
            double t = 0.0, hc = 0.0, hs = 0.0;
            if (act) {
                hc = a.ph[off];
                for (int i = 0; i < ME; ++i) {
                    const int e = cptr(m.eoc)[(size_t)c * ME + i];
                    if (e < 0 || k >= cptr(m.mltc)[(size_t)c * ME + i]) continue;
                    const double hE = 0.5 * (hc + a.ph[(size_t)cptr(m.coc)[(size_t)c * ME + i] * K + k]);
                    const double F = a.pu[(size_t)e * K + k] * hE;
                    t += F * cptr(m.sdv)[(size_t)c * ME + i] * invA;                    // horizontal_advection.jl:63-64
                }
                if (a.tendH) a.tendH[off] = t;
                const double hcur = a.ch ? a.ch[off] : hc;
                if (a.ph_out) {
                    const double hp = hcur + a.a * t;
                    a.ph_out[off] = hp;
                    hs = hp;
                }
                if (a.nh_out) {
                    const double hn = (a.nh_in ? a.nh_in[off] : hcur) + a.b * t;
                    a.nh_out[off] = hn;
                    if (!a.ph_out) hs = hn;
                }
            }
            sshAcc = first ? hs : sshAcc + hs;
            first = false;
        }
        if (a.ssh_out) {
            const double s = group_sum<LPC>(sshAcc);
            if (l == 0) a.ssh_out[c] = s - cptr(m.rsum)[c];
        }
    }
    for (int e = blockIdx.x * NG + grp; e < m.nE; e += gridDim.x * NG) {
        const int c1 = cptr(m.ehdr)[(size_t)e * 4], c2 = cptr(m.ehdr)[(size_t)e * 4 + 1], mlt = cptr(m.ehdr)[(size_t)e * 4 + 3];
        const double g = cptr(m.gInvDc)[e], invDc = cptr(m.invDc)[e];
        const double ds = a.ssh[c2] - a.ssh[c1];
        const bool del2 = nl.zv != nullptr;
        const double invDv = del2 ? 1.0 / cptr(m.dvEdge)[e] : 0.0;
        const int v1 = del2 ? cptr(m.voe)[(size_t)e * 2] : 0, v2 = del2 ? cptr(m.voe)[(size_t)e * 2 + 1] : 0;
        for (int k = l; k < K; k += LPC) {
            const size_t off = (size_t)e * K + k;
            double t = 0.0;
            if (k < mlt) {
                t -= g * ds;
                t -= invDc * (nl.ke[(size_t)c2 * K + k] - nl.ke[(size_t)c1 * K + k]);
                const double qe = fq[off].y;
                for (int i = 0; i < ME2; ++i) {
                    const int x = cptr(m.eoe)[(size_t)e * ME2 + i];
                    if (x < 0) continue;
                    const double2 n = fq[(size_t)x * K + k];                            // {F, q_e} of the neighbour edge
                    t += cptr(m.woe)[(size_t)e * ME2 + i] * n.x * (0.5 * (qe + n.y));
                }
                if (del2)                                                               // horizontal_momentum_mixing.jl:75-78
                    t += ((nl.divc[(size_t)c2 * K + k] - nl.divc[(size_t)c1 * K + k]) * invDc -
                          (nl.zv[(size_t)v2 * K + k] - nl.zv[(size_t)v1 * K + k]) * invDv) * nl.visc;
            }
            if (a.tendU) a.tendU[off] = t;
            const double ucur = a.cu ? a.cu[off] : a.pu[off];
            if (a.pu_out) a.pu_out[off] = ucur + a.a * t;
            if (a.nu_out) a.nu_out[off] = (a.nu_in ? a.nu_in[off] : ucur) + a.b * t;
        }
    }
}

template <int LPC>
static hipError_t launch_nl_prepare_lpc(const MeshDev &m, const double *u, const double *h, const NlArgs &nl, hipStream_t s)
{
    const int ng = BLOCK / LPC;
    auto grid = [&](int n) { return dim3((unsigned)std::min(std::max((n + ng - 1) / ng, 1), 65536)); };
    hipLaunchKernelGGL((k_nl_vertex<LPC>), grid(m.nV), dim3(BLOCK), 0, s, m, u, h, nl.qv, nl.zv);
    hipLaunchKernelGGL((k_nl_cell<LPC>), grid(m.nC), dim3(BLOCK), 0, s, m, u, nl.ke, nl.divc);
    hipLaunchKernelGGL((k_nl_edge<LPC>), grid(m.nE), dim3(BLOCK), 0, s, m, u, h, nl);   // after k_nl_vertex (same stream)
    return hipGetLastError();
}

template <int LPC>
static hipError_t launch_stage_nl_lpc(const MeshDev &m, const StageArgs &a, const NlArgs &nl, hipStream_t s)
{
    const int ng = BLOCK / LPC;
    const int grid = std::min(std::max((std::max(m.nE, m.nC) + ng - 1) / ng, 1), 65536);
    hipLaunchKernelGGL((k_stage_nl<LPC>), dim3(grid), dim3(BLOCK), 0, s, m, a, nl);
    return hipGetLastError();
}

hipError_t launch_nl_prepare(const MeshDev &m, const double *u, const double *h, const NlArgs &nl, int lpc, hipStream_t s)
{
#define CALL(L) launch_nl_prepare_lpc<L>(m, u, h, nl, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

hipError_t launch_stage_nl(const MeshDev &m, const StageArgs &a, const NlArgs &nl, int lpc, hipStream_t s)
{
#define CALL(L) launch_stage_nl_lpc<L>(m, a, nl, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

}  // namespace moka
