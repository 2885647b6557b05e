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

// ------------------------------------------------------------------------------------------------
// The same four kernels for even K <= 64 with 16-byte lanes: half a wave per entity, a lane owns levels 2l and 2l+1
// (the layout of the forward stage kernel); same operand order, bit-identical results.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 ld2(const double *p, size_t row, int K, int l) { return reinterpret_cast<const double2 *>(p + row * K)[l]; }
__device__ __forceinline__ void st2(double *p, size_t row, int K, int l, double2 v) { reinterpret_cast<double2 *>(p + row * K)[l] = v; }

__global__ __launch_bounds__(BLOCK) void k_nl_vertex2(const MeshDev m, const double *u, const double *h, double *qv, double *zv)
{
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int K = m.K, VD = m.VD;
    if (2 * l >= K) return;
    for (int v = blockIdx.x * (BLOCK / 32) + grp; v < m.nV; v += gridDim.x * (BLOCK / 32)) {
        const double invA = m.invAreaTri[v], fv = m.fVertex[v];
        double2 zeta = make_double2(0.0, 0.0), hv = zeta;
        for (int j = 0; j < VD; ++j) {
            const double c = m.cv[(size_t)v * VD + j], kt = m.kite[(size_t)v * VD + j];
            const double2 uu = ld2(u, m.eov[(size_t)v * VD + j], K, l), hh = ld2(h, m.cov[(size_t)v * VD + j], K, l);
            zeta.x += c * uu.x; zeta.y += c * uu.y;
            hv.x += kt * hh.x; hv.y += kt * hh.y;
        }
        hv.x = hv.x * invA; hv.y = hv.y * invA;
        st2(qv, v, K, l, make_double2((fv + zeta.x) / hv.x, (fv + zeta.y) / hv.y));
        if (zv) st2(zv, v, K, l, zeta);
    }
}

__global__ __launch_bounds__(BLOCK) void k_nl_edge2(const MeshDev m, const double *u, const double *h, const NlArgs nl)
{
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int K = m.K;
    if (2 * l >= K) return;
    double2 *fq = reinterpret_cast<double2 *>(nl.fq);
    for (int e = blockIdx.x * (BLOCK / 32) + grp; e < m.nE; e += gridDim.x * (BLOCK / 32)) {
        const int c1 = m.ehdr[(size_t)e * 4], c2 = m.ehdr[(size_t)e * 4 + 1];
        const int v1 = m.voe[(size_t)e * 2], v2 = m.voe[(size_t)e * 2 + 1];
        const double2 h1 = ld2(h, c1, K, l), h2 = ld2(h, c2, K, l), q1 = ld2(nl.qv, v1, K, l), q2 = ld2(nl.qv, v2, K, l);
        const double2 uu = ld2(u, e, K, l);
        double2 *dst = fq + (size_t)e * K + 2 * l;                     // {F, q_e} pairs of levels 2l, 2l+1
        dst[0] = make_double2(uu.x * (0.5 * (h1.x + h2.x)), 0.5 * (q1.x + q2.x));
        dst[1] = make_double2(uu.y * (0.5 * (h1.y + h2.y)), 0.5 * (q1.y + q2.y));
    }
}

__global__ __launch_bounds__(BLOCK) void k_nl_cell2(const MeshDev m, const double *u, double *ke, double *divc)
{
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int K = m.K, ME = m.ME;
    if (2 * l >= K) return;
    for (int c = blockIdx.x * (BLOCK / 32) + grp; c < m.nC; c += gridDim.x * (BLOCK / 32)) {
        const double invA = m.invArea[c], area = m.areaCell[c];
        double2 acc = make_double2(0.0, 0.0), d = acc;
        for (int i = 0; i < ME; ++i) {
            const int e = m.eoc[(size_t)c * ME + i];
            if (e < 0) continue;
            const double2 ue = ld2(u, e, K, l);
            const double kc = m.keCoef[e], sd = m.sdv[(size_t)c * ME + i];
            acc.x += kc * ue.x * ue.x; acc.y += kc * ue.y * ue.y;
            d.x -= ue.x * sd; d.y -= ue.y * sd;
        }
        st2(ke, c, K, l, make_double2(acc.x * invA, acc.y * invA));
        if (divc) st2(divc, c, K, l, make_double2(d.x / area, d.y / area));
    }
}

__global__ __launch_bounds__(BLOCK) void k_stage_nl2(const MeshDev m, const StageArgs a, const NlArgs nl)
{
    constexpr int NG = BLOCK / 32;
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int K = m.K, ME = m.ME, ME2 = m.ME2, k0 = 2 * l;
    const bool act = k0 < K;
    const double2 *fq = reinterpret_cast<const double2 *>(nl.fq);
    for (int c = blockIdx.x * NG + grp; c < m.nC; c += gridDim.x * NG) {
        const double invA = m.invArea[c];
        double2 hs = make_double2(0.0, 0.0);
        if (act) {
            const double2 hc = ld2(a.ph, c, K, l);
            double2 t = make_double2(0.0, 0.0);
            for (int i = 0; i < ME; ++i) {
                const int e = m.eoc[(size_t)c * ME + i];
                if (e < 0) continue;
                const int ml = m.mltc[(size_t)c * ME + i];
                const double2 hn = ld2(a.ph, m.coc[(size_t)c * ME + i], K, l), ue = ld2(a.pu, e, K, l);
                const double sd = m.sdv[(size_t)c * ME + i];
                if (k0 < ml) t.x += ue.x * (0.5 * (hc.x + hn.x)) * sd * invA;         // horizontal_advection.jl:63-64
                if (k0 + 1 < ml) t.y += ue.y * (0.5 * (hc.y + hn.y)) * sd * invA;
            }
            if (a.tendH) st2(a.tendH, c, K, l, t);
            const double2 hcur = a.ch ? ld2(a.ch, c, K, l) : hc;
            if (a.ph_out) {
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);
                st2(a.ph_out, c, K, l, hs);
            }
            if (a.nh_out) {
                const double2 nb = a.nh_in ? ld2(a.nh_in, c, K, l) : hcur;
                const double2 hn = make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y);
                st2(a.nh_out, c, K, l, hn);
                if (!a.ph_out) hs = hn;
            }
        }
        if (a.ssh_out) {
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) {                   // oracle_ksum order
                const double ox = __shfl_xor(hs.x, sft, 32), oy = __shfl_xor(hs.y, sft, 32);
                hs = make_double2(hs.x + ox, hs.y + oy);
            }
            if (l == 0) a.ssh_out[c] = (hs.x + hs.y) - m.rsum[c];
        }
    }
    if (!act) return;
    const bool del2 = nl.zv != nullptr;
    for (int e = blockIdx.x * NG + grp; e < m.nE; e += gridDim.x * NG) {
        const int c1 = m.ehdr[(size_t)e * 4], c2 = m.ehdr[(size_t)e * 4 + 1], mlt = m.ehdr[(size_t)e * 4 + 3];
        const double g = m.gInvDc[e], invDc = m.invDc[e];
        const double ds = a.ssh[c2] - a.ssh[c1];
        const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
        double2 t = make_double2(0.0, 0.0);
        const double2 *own = fq + (size_t)e * K + k0;
        if (ax) t.x -= g * ds;
        if (ay) t.y -= g * ds;
        const double2 k1 = ld2(nl.ke, c1, K, l), k2 = ld2(nl.ke, c2, K, l);
        if (ax) t.x -= invDc * (k2.x - k1.x);
        if (ay) t.y -= invDc * (k2.y - k1.y);
        const double qx = own[0].y, qy = own[1].y;
        for (int i = 0; i < ME2; ++i) {
            const int x = m.eoe[(size_t)e * ME2 + i];
            if (x < 0) continue;
            const double w = m.woe[(size_t)e * ME2 + i];
            const double2 *nb = fq + (size_t)x * K + k0;                // {F, q_e} of the neighbour edge, two levels
            const double2 n0 = nb[0], n1 = nb[1];
            if (ax) t.x += w * n0.x * (0.5 * (qx + n0.y));
            if (ay) t.y += w * n1.x * (0.5 * (qy + n1.y));
        }
        if (del2) {                                                     // horizontal_momentum_mixing.jl:75-78
            const double invDv = 1.0 / m.dvEdge[e];
            const int v1 = m.voe[(size_t)e * 2], v2 = m.voe[(size_t)e * 2 + 1];
            const double2 d1 = ld2(nl.divc, c1, K, l), d2 = ld2(nl.divc, c2, K, l), z1 = ld2(nl.zv, v1, K, l), z2 = ld2(nl.zv, v2, K, l);
            if (ax) t.x += ((d2.x - d1.x) * invDc - (z2.x - z1.x) * invDv) * nl.visc;
            if (ay) t.y += ((d2.y - d1.y) * invDc - (z2.y - z1.y) * invDv) * nl.visc;
        }
        if (a.tendU) st2(a.tendU, e, K, l, t);
        const double2 ucur = a.cu ? ld2(a.cu, e, K, l) : ld2(a.pu, e, K, l);
        if (a.pu_out) st2(a.pu_out, e, K, l, make_double2(ucur.x + a.a * t.x, ucur.y + a.a * t.y));
        if (a.nu_out) {
            const double2 nb = a.nu_in ? ld2(a.nu_in, e, K, l) : ucur;
            st2(a.nu_out, e, K, l, make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y));
        }
    }
}

static inline dim3 grid2(int n) { return dim3((unsigned)std::min(std::max((n + 7) / 8, 1), 65536)); }

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
    if (lpc == 64 && m.K <= 64 && !(m.K & 1)) {     // even 34 <= K <= 64: 16-byte lanes
        hipLaunchKernelGGL(k_nl_vertex2, grid2(m.nV), dim3(BLOCK), 0, s, m, u, h, nl.qv, nl.zv);
        hipLaunchKernelGGL(k_nl_cell2, grid2(m.nC), dim3(BLOCK), 0, s, m, u, nl.ke, nl.divc);
        hipLaunchKernelGGL(k_nl_edge2, grid2(m.nE), dim3(BLOCK), 0, s, m, u, h, nl);   // after k_nl_vertex2 (same stream)
        return hipGetLastError();
    }
#define CALL(L) launch_nl_prepare_lpc<L>(m, u, h, nl, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

hipError_t launch_stage_nl(const MeshDev &m, const StageArgs &a, const NlArgs &nl, int lpc, hipStream_t s)
{
    if (lpc == 64 && m.K <= 64 && !(m.K & 1)) {
        hipLaunchKernelGGL(k_stage_nl2, grid2(std::max(m.nE, m.nC)), dim3(BLOCK), 0, s, m, a, nl);
        return hipGetLastError();
    }
#define CALL(L) launch_stage_nl_lpc<L>(m, a, nl, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

}  // namespace moka
