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

// ------------------------------------------------------------------------------------------------
// Round 2: the same algebra for even K <= 64, one workgroup per PATCH (the plan's 16-cell patches own their cells, edges
// and vertices as contiguous ranges), record widths as template constants so that every index, weight and row load of an
// entity is issued as one batch before the first use.  Two launches per stage instead of four:
//   k_nl_prep4  : qv (+zv) of the patch's vertices, ke (+divc) of its cells, F = u * layerThicknessEdge of its edges --
//                 the u and h rows of a patch are fetched once from HBM and twice more from L2.  NlArgs.fq holds F alone.
//   k_stage_nl3 / k_stage_nl4 : the thickness tendency from F (6 rows instead of 6 u + 6 h rows); the velocity tendency averages the
//                 vertex potential vorticity to the edges on the fly (q_e is never stored: 10 F + 20 qv rows per edge,
//                 the qv rows shared by the edges around a cell) -- the operand order of k_stage_nl2, bit for bit.
// ------------------------------------------------------------------------------------------------
// k_nl_prep4: the records of the patch are staged in LDS first (one round trip for every index, weight and metric factor of
// the patch instead of one per entity in front of its gathers), then two vertices / two cells / three edges per half-wave
// round: the kernel is bound by the number of dependent memory round trips a workgroup makes, not by bytes.
// block -> patch by patch_of_block (kernels_common.hpp): an XCD walks one contiguous eighth of the patch list, so the patches in
// flight on an XCD are neighbours and their halo rows meet in that XCD's L2 (config 4, PMC: k_stage_nl4 fetches 8.7 GB per
// launch instead of 13.3, k_nl_prep4 3.1 instead of 5.4 -- at unchanged times: both kernels are bound by latency, not bytes)
static inline unsigned nl_grid(int nPatches) { return 8u * (unsigned)((nPatches + 7) / 8); }

constexpr int NL4_VCH = 64, NL4_CCH = 16, NL4_ECH = 96;   // vertices / cells / edges staged per chunk

template <int ME_, int VD_>
__global__ __launch_bounds__(BLOCK) void k_nl_prep4(const MeshDev m, const double *__restrict__ u, const double *__restrict__ h, const NlArgs nl)
{
    constexpr int NG = BLOCK / 32;
    __shared__ int sVi[NL4_VCH * 2 * VD_];        // edgesOnVertex | cellsOnVertex
    __shared__ double sVd[NL4_VCH * (2 * VD_ + 2)];   // cv | kite | invAreaTri, fVertex
    __shared__ int sCi[NL4_CCH * ME_];
    __shared__ double sCd[NL4_CCH * (2 * ME_ + 2)];   // sdv | keCoef | invArea, areaCell
    __shared__ int2 sEc[NL4_ECH];
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31, K = m.K;
    const bool act = 2 * l < K;
    const int p = patch_of_block(m.nPatches);
    if (p >= m.nPatches) return;
    const int v0 = m.patchVertStart[p], v1 = m.patchVertStart[p + 1], c0 = m.patchCellStart[p], c1 = m.patchCellStart[p + 1];
    const int e0 = m.patchEdgeStart[p], e1 = m.patchEdgeStart[p + 1];
    const int nChunks = max(max((v1 - v0 + NL4_VCH - 1) / NL4_VCH, (c1 - c0 + NL4_CCH - 1) / NL4_CCH), (e1 - e0 + NL4_ECH - 1) / NL4_ECH);
    for (int ch = 0; ch < nChunks; ++ch) {
        const int vb = v0 + ch * NL4_VCH, nv = min(max(v1 - vb, 0), NL4_VCH);
        const int cb = c0 + ch * NL4_CCH, nc = min(max(c1 - cb, 0), NL4_CCH);
        const int eb = e0 + ch * NL4_ECH, ne = min(max(e1 - eb, 0), NL4_ECH);
        if (ch) __syncthreads();
        for (int i = threadIdx.x; i < nv * VD_; i += BLOCK) {
            const int v = i / VD_, j = i % VD_;
            sVi[v * 2 * VD_ + j] = m.eov[(size_t)vb * VD_ + i]; sVi[v * 2 * VD_ + VD_ + j] = m.cov[(size_t)vb * VD_ + i];
            sVd[v * (2 * VD_ + 2) + j] = m.cv[(size_t)vb * VD_ + i]; sVd[v * (2 * VD_ + 2) + VD_ + j] = m.kite[(size_t)vb * VD_ + i];
        }
        for (int i = threadIdx.x; i < nv; i += BLOCK) {
            sVd[i * (2 * VD_ + 2) + 2 * VD_] = m.invAreaTri[vb + i]; sVd[i * (2 * VD_ + 2) + 2 * VD_ + 1] = m.fVertex[vb + i];
        }
        for (int i = threadIdx.x; i < nc * ME_; i += BLOCK) {
            const int c = i / ME_, j = i % ME_;
            const int e = m.eoc[(size_t)cb * ME_ + i];
            sCi[i] = e;
            sCd[c * (2 * ME_ + 2) + j] = m.sdv[(size_t)cb * ME_ + i];
            sCd[c * (2 * ME_ + 2) + ME_ + j] = m.keoc[(size_t)cb * ME_ + i];
        }
        for (int i = threadIdx.x; i < nc; i += BLOCK) {
            sCd[i * (2 * ME_ + 2) + 2 * ME_] = m.invArea[cb + i]; sCd[i * (2 * ME_ + 2) + 2 * ME_ + 1] = m.areaCell[cb + i];
        }
        for (int i = threadIdx.x; i < ne; i += BLOCK) sEc[i] = reinterpret_cast<const int2 *>(m.ehdr)[2 * (size_t)(eb + i)];
        __syncthreads();
        if (!act) continue;
        for (int vi = grp; vi < nv; vi += 2 * NG) {                           // two vertices per round
            double2 uu[2][VD_], hh[2][VD_];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int v = vi + q * NG < nv ? vi + q * NG : vi;
#pragma unroll
                for (int j = 0; j < VD_; ++j) { uu[q][j] = ld2(u, sVi[v * 2 * VD_ + j], K, l); hh[q][j] = ld2(h, sVi[v * 2 * VD_ + VD_ + j], K, l); }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int v = vi + q * NG;
                if (v >= nv) break;
                const double *rd = sVd + v * (2 * VD_ + 2);
                double2 zeta = make_double2(0.0, 0.0), hv = zeta;
#pragma unroll
                for (int j = 0; j < VD_; ++j) {
                    zeta.x += rd[j] * uu[q][j].x; zeta.y += rd[j] * uu[q][j].y;
                    hv.x += rd[VD_ + j] * hh[q][j].x; hv.y += rd[VD_ + j] * hh[q][j].y;
                }
                const double invA = rd[2 * VD_], fv = rd[2 * VD_ + 1];
                hv.x = hv.x * invA; hv.y = hv.y * invA;
                st2(nl.qv, vb + v, K, l, make_double2((fv + zeta.x) / hv.x, (fv + zeta.y) / hv.y));
                if (nl.zv) st2(nl.zv, vb + v, K, l, zeta);
            }
        }
        for (int ci = grp; ci < nc; ci += 2 * NG) {                           // two cells per round
            double2 ue[2][ME_];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = ci + q * NG < nc ? ci + q * NG : ci;
#pragma unroll
                for (int i = 0; i < ME_; ++i) { const int e = sCi[c * ME_ + i]; ue[q][i] = ld2(u, e < 0 ? 0 : e, K, l); }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = ci + q * NG;
                if (c >= nc) break;
                const double *rd = sCd + c * (2 * ME_ + 2);
                double2 acc = make_double2(0.0, 0.0), d = acc;
#pragma unroll
                for (int i = 0; i < ME_; ++i) {
                    const bool ok = sCi[c * ME_ + i] >= 0;
                    const double kc = rd[ME_ + i], sd = rd[i];
                    const double ax = acc.x + kc * ue[q][i].x * ue[q][i].x, ay = acc.y + kc * ue[q][i].y * ue[q][i].y;
                    const double dx = d.x - ue[q][i].x * sd, dy = d.y - ue[q][i].y * sd;
                    acc.x = ok ? ax : acc.x; acc.y = ok ? ay : acc.y;
                    d.x = ok ? dx : d.x; d.y = ok ? dy : d.y;
                }
                const double invA = rd[2 * ME_], area = rd[2 * ME_ + 1];
                st2(nl.ke, cb + c, K, l, make_double2(acc.x * invA, acc.y * invA));
                if (nl.divc) st2(nl.divc, cb + c, K, l, make_double2(d.x / area, d.y / area));
            }
        }
        for (int ei = grp; ei < ne; ei += 3 * NG) {                           // three edges per round
            double2 h1[3], h2[3], uu[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = ei + q * NG < ne ? ei + q * NG : ei;
                const int2 cc = sEc[e];
                h1[q] = ld2(h, cc.x, K, l); h2[q] = ld2(h, cc.y, K, l); uu[q] = ld2(u, eb + e, K, l);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = ei + q * NG;
                if (e >= ne) break;
                st2(nl.fq, eb + e, K, l, make_double2(uu[q].x * (0.5 * (h1[q].x + h2[q].x)), uu[q].y * (0.5 * (h1[q].y + h2[q].y))));   // Operators.jl:217, DiagnosticVars.jl:165
            }
        }
    }
}

constexpr int NL3_MAXE = 96;   // own edges of a patch the LDS records of k_stage_nl3 hold (P = 16 cells x 6)

template <int ME_, int ME2_>
__global__ __launch_bounds__(BLOCK, 3) void k_stage_nl3(const MeshDev m, const StageArgs a, const NlArgs nl)
{
    constexpr int NG = BLOCK / 32;
    // records of the patch's own edges: neighbour edge, its two vertices (looked up here, once per patch), weight
    __shared__ int sX[NL3_MAXE * ME2_];
    __shared__ int2 sV[NL3_MAXE * ME2_];
    __shared__ double sW[NL3_MAXE * ME2_];
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31, K = m.K, k0 = 2 * l;
    const bool act = k0 < K;
    const int p = patch_of_block(m.nPatches);
    if (p >= m.nPatches) return;
    const double *__restrict__ F = nl.fq;
    const int e0 = m.patchEdgeStart[p], e1 = m.patchEdgeStart[p + 1];
    for (int i = threadIdx.x; i < (e1 - e0) * ME2_; i += BLOCK) {
        const int x = m.eoe[(size_t)e0 * ME2_ + i];
        sX[i] = x;
        sW[i] = m.woe[(size_t)e0 * ME2_ + i];
        sV[i] = reinterpret_cast<const int2 *>(m.voe)[x < 0 ? e0 : x];
    }
    for (int c = m.patchCellStart[p] + grp; c < m.patchCellStart[p + 1]; c += NG) {
        double2 hs = make_double2(0.0, 0.0);
        if (act) {
            int e[ME_], ml[ME_];
            double sd[ME_];
#pragma unroll
            for (int i = 0; i < ME_; ++i) {
                e[i] = m.eoc[(size_t)c * ME_ + i]; ml[i] = m.mltc[(size_t)c * ME_ + i]; sd[i] = m.sdv[(size_t)c * ME_ + i];
            }
            const double invA = m.invArea[c];
            double2 Fe[ME_];
#pragma unroll
            for (int i = 0; i < ME_; ++i) Fe[i] = ld2(F, e[i] < 0 ? 0 : e[i], K, l);
            const double2 hcur = a.ch ? ld2(a.ch, c, K, l) : ld2(a.ph, c, K, l);
            double2 nb = hcur;
            if (a.nh_out && a.nh_in) nb = ld2(a.nh_in, c, K, l);
            double2 t = make_double2(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < ME_; ++i) {
                const double tx = t.x + Fe[i].x * sd[i] * invA, ty = t.y + Fe[i].y * sd[i] * invA;   // horizontal_advection.jl:63-64
                t.x = (e[i] >= 0 && k0 < ml[i]) ? tx : t.x;
                t.y = (e[i] >= 0 && k0 + 1 < ml[i]) ? ty : t.y;
            }
            if (a.tendH) st2(a.tendH, c, K, l, t);
            if (a.ph_out) {
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);
                st2(a.ph_out, c, K, l, hs);
            }
            if (a.nh_out) {
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
    __syncthreads();
    if (!act) return;
    const bool del2 = nl.zv != nullptr;
    for (int e = e0 + grp; e < e1; e += NG) {
        const int4 hd = reinterpret_cast<const int4 *>(m.ehdr)[e];
        const int c1 = hd.x, c2 = hd.y, mlt = hd.w;
        const int2 vo = reinterpret_cast<const int2 *>(m.voe)[e];
        const double g = m.gInvDc[e], invDc = m.invDc[e];
        const double ds = a.ssh[c2] - a.ssh[c1];
        const double2 k1 = ld2(nl.ke, c1, K, l), k2 = ld2(nl.ke, c2, K, l);
        const double2 q1 = ld2(nl.qv, vo.x, K, l), q2 = ld2(nl.qv, vo.y, K, l);
        const double2 ucur = a.cu ? ld2(a.cu, e, K, l) : ld2(a.pu, e, K, l);
        double2 nbu = ucur;
        if (a.nu_out && a.nu_in) nbu = ld2(a.nu_in, e, K, l);
        const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
        double2 t = make_double2(0.0, 0.0);
        if (ax) t.x -= g * ds;
        if (ay) t.y -= g * ds;
        if (ax) t.x -= invDc * (k2.x - k1.x);
        if (ay) t.y -= invDc * (k2.y - k1.y);
        const double qx = 0.5 * (q1.x + q2.x), qy = 0.5 * (q1.y + q2.y);
        const int r0 = (e - e0) * ME2_;
        constexpr int HB = (ME2_ + 1) / 2;                                  // two batches: bounds the registers held by gathers
#pragma unroll
        for (int b0 = 0; b0 < ME2_; b0 += HB) {
            double2 Fx[HB], qa[HB], qb[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int i = b0 + j;
                if (i < ME2_) {
                    const int x = sX[r0 + i];
                    const int2 vx = sV[r0 + i];
                    Fx[j] = ld2(F, x < 0 ? e : x, K, l);
                    qa[j] = ld2(nl.qv, vx.x, K, l); qb[j] = ld2(nl.qv, vx.y, K, l);
                }
            }
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int i = b0 + j;
                if (i < ME2_) {
                    const double w = sW[r0 + i];
                    const bool ok = sX[r0 + i] >= 0;
                    const double nx = 0.5 * (qa[j].x + qb[j].x), ny = 0.5 * (qa[j].y + qb[j].y);   // q_e of the neighbour edge
                    const double tx = t.x + w * Fx[j].x * (0.5 * (qx + nx)), ty = t.y + w * Fx[j].y * (0.5 * (qy + ny));
                    t.x = (ok && ax) ? tx : t.x;
                    t.y = (ok && ay) ? ty : t.y;
                }
            }
        }
        if (del2) {                                                     // horizontal_momentum_mixing.jl:75-78
            const double invDv = 1.0 / m.dvEdge[e];
            const double2 d1 = ld2(nl.divc, c1, K, l), d2 = ld2(nl.divc, c2, K, l), z1 = ld2(nl.zv, vo.x, K, l), z2 = ld2(nl.zv, vo.y, K, l);
            if (ax) t.x += ((d2.x - d1.x) * invDc - (z2.x - z1.x) * invDv) * nl.visc;
            if (ay) t.y += ((d2.y - d1.y) * invDc - (z2.y - z1.y) * invDv) * nl.visc;
        }
        if (a.tendU) st2(a.tendU, e, K, l, t);
        if (a.pu_out) st2(a.pu_out, e, K, l, make_double2(ucur.x + a.a * t.x, ucur.y + a.a * t.y));
        if (a.nu_out) st2(a.nu_out, e, K, l, make_double2(nbu.x + a.b * t.x, nbu.y + a.b * t.y));
    }
}

// k_stage_nl3 is bound by the L2 -> L1 request rate (34 row gathers per edge, 13 TB/s of requests at config 4).  This form
// averages the potential vorticity to the edges ONCE per patch: q_e of every edge row the patch touches (its own edges, then
// the halo edges of the plan's row list, ~110 rows of a 16-cell patch) is built in LDS from 2 qv rows each, and the edge
// loop reads q_e of its neighbour edges from LDS by the plan's patch-local row ids (leoe).  F rows stay global gathers.
template <int ME_, int ME2_, int NT>
__global__ __launch_bounds__(NT, 4) void k_stage_nl4(const MeshDev m, const StageArgs a, const NlArgs nl)
{
    constexpr int NG = NT / 32, RB = 8;
    extern __shared__ __align__(16) unsigned char nl4_smem[];
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31, K = m.K, k0 = 2 * l;
    const bool act = k0 < K;
    double *sQ = reinterpret_cast<double *>(nl4_smem);                     // [maxRows][K]    q_e rows
    double *sW = sQ + (size_t)m.maxRows * K;                               // [maxOwnE][ME2]  weightsOnEdge
    int4 *sH = reinterpret_cast<int4 *>(sW + (size_t)m.maxOwnE * ME2_);    // [maxOwnE]       {c1, c2, nEdgesOnEdge, maxLevelEdgeTop}
    double2 *sG = reinterpret_cast<double2 *>(sH + m.maxOwnE);             // [maxOwnE]       {g / dcEdge, 1 / dcEdge}
    double *sCs = reinterpret_cast<double *>(sG + m.maxOwnE);              // [maxOwnC][ME+1] sdv | invArea
    int2 *sV = reinterpret_cast<int2 *>(sCs + (size_t)m.maxOwnC * (ME_ + 1));   // [maxRows]  verticesOnEdge of the row's edge
    int *sX = reinterpret_cast<int *>(sV + m.maxRows);                     // [maxOwnE][ME2]  edgesOnEdge (global ids, -1 = none)
    int *sCe = sX + (size_t)m.maxOwnE * ME2_;                              // [maxOwnC][2 ME] edgesOnCell | maxLevelEdgeTop of the edge
    unsigned char *sL = reinterpret_cast<unsigned char *>(sCe + (size_t)m.maxOwnC * 2 * ME_);   // [maxOwnE][16]  patch-local row of each slot
    const int p = patch_of_block(m.nPatches);
    if (p >= m.nPatches) return;
    const double *__restrict__ F = nl.fq;
    const int e0 = m.patchEdgeStart[p], e1 = m.patchEdgeStart[p + 1], nOwn = e1 - e0;
    const int r0 = m.rowStart[p], nRows = m.rowStart[p + 1] - r0;
    for (int i = threadIdx.x; i < nRows; i += NT) sV[i] = reinterpret_cast<const int2 *>(m.rowVoe)[r0 + i];
    for (int i = threadIdx.x; i < nOwn * ME2_; i += NT) { sX[i] = m.eoe[(size_t)e0 * ME2_ + i]; sW[i] = m.woe[(size_t)e0 * ME2_ + i]; }
    for (int i = threadIdx.x; i < nOwn * 4; i += NT) reinterpret_cast<int *>(sL)[i] = reinterpret_cast<const int *>(m.leoe)[(size_t)e0 * 4 + i];
    const int c0 = m.patchCellStart[p], nC = m.patchCellStart[p + 1] - c0;
    for (int i = threadIdx.x; i < nOwn; i += NT) {
        sH[i] = reinterpret_cast<const int4 *>(m.ehdr)[e0 + i];
        sG[i] = make_double2(m.gInvDc[e0 + i], m.invDc[e0 + i]);
    }
    for (int i = threadIdx.x; i < nC * ME_; i += NT) {
        const int c = i / ME_, j = i % ME_;
        sCe[c * 2 * ME_ + j] = m.eoc[(size_t)c0 * ME_ + i]; sCe[c * 2 * ME_ + ME_ + j] = m.mltc[(size_t)c0 * ME_ + i];
        sCs[c * (ME_ + 1) + j] = m.sdv[(size_t)c0 * ME_ + i];
    }
    for (int i = threadIdx.x; i < nC; i += NT) sCs[i * (ME_ + 1) + ME_] = m.invArea[c0 + i];
    __syncthreads();
    for (int c = c0 + grp; c < c0 + nC; c += NG) {
        double2 hs = make_double2(0.0, 0.0);
        if (act) {
            int e[ME_], ml[ME_];
            double sd[ME_];
            double2 Fe[ME_];
#pragma unroll
            for (int i = 0; i < ME_; ++i) { e[i] = sCe[(c - c0) * 2 * ME_ + i]; Fe[i] = ld2(F, e[i] < 0 ? 0 : e[i], K, l); }
#pragma unroll
            for (int i = 0; i < ME_; ++i) { ml[i] = sCe[(c - c0) * 2 * ME_ + ME_ + i]; sd[i] = sCs[(c - c0) * (ME_ + 1) + i]; }
            const double invA = sCs[(c - c0) * (ME_ + 1) + ME_];
            const double2 hcur = a.ch ? ld2(a.ch, c, K, l) : ld2(a.ph, c, K, l);
            double2 nb = hcur;
            if (a.nh_out && a.nh_in) nb = ld2(a.nh_in, c, K, l);
            double2 t = make_double2(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < ME_; ++i) {
                const double tx = t.x + Fe[i].x * sd[i] * invA, ty = t.y + Fe[i].y * sd[i] * invA;   // horizontal_advection.jl:63-64
                t.x = (e[i] >= 0 && k0 < ml[i]) ? tx : t.x;
                t.y = (e[i] >= 0 && k0 + 1 < ml[i]) ? ty : t.y;
            }
            if (a.tendH) st2(a.tendH, c, K, l, t);
            if (a.ph_out) {
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);
                st2(a.ph_out, c, K, l, hs);
            }
            if (a.nh_out) {
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
    for (int r = grp; r < nRows; r += NG * RB) {                         // q_e of the patch's rows, RB rows in flight per half-wave
        double2 qa[RB], qb[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int rr = r + j * NG;
            const int2 v = sV[rr < nRows ? rr : r];
            if (act) { qa[j] = ld2(nl.qv, v.x, K, l); qb[j] = ld2(nl.qv, v.y, K, l); }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int rr = r + j * NG;
            if (act && rr < nRows)
                reinterpret_cast<double2 *>(sQ + (size_t)rr * K)[l] = make_double2(0.5 * (qa[j].x + qb[j].x), 0.5 * (qa[j].y + qb[j].y));
        }
    }
    __syncthreads();
    if (!act) return;
    const bool del2 = nl.zv != nullptr;
    for (int e = e0 + grp; e < e1; e += NG) {
        const int le = e - e0;
        const int4 hd = sH[le];
        const int c1 = hd.x, c2 = hd.y, mlt = hd.w;
        double2 Fx[ME2_];
#pragma unroll
        for (int i = 0; i < ME2_; ++i) { const int x = sX[le * ME2_ + i]; Fx[i] = ld2(F, x < 0 ? e : x, K, l); }
        const double g = sG[le].x, invDc = sG[le].y;
        const double ds = a.ssh[c2] - a.ssh[c1];
        const double2 k1 = ld2(nl.ke, c1, K, l), k2 = ld2(nl.ke, c2, K, l);
        const double2 ucur = a.cu ? ld2(a.cu, e, K, l) : ld2(a.pu, e, K, l);
        double2 nbu = ucur;
        if (a.nu_out && a.nu_in) nbu = ld2(a.nu_in, e, K, l);
        const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
        double2 t = make_double2(0.0, 0.0);
        if (ax) t.x -= g * ds;
        if (ay) t.y -= g * ds;
        if (ax) t.x -= invDc * (k2.x - k1.x);
        if (ay) t.y -= invDc * (k2.y - k1.y);
        const double2 qo = reinterpret_cast<const double2 *>(sQ + (size_t)le * K)[l];
#pragma unroll
        for (int i = 0; i < ME2_; ++i) {
            const int lr = sL[le * 16 + i];
            const bool ok = sX[le * ME2_ + i] >= 0;
            const double w = sW[le * ME2_ + i];
            const double2 qn = reinterpret_cast<const double2 *>(sQ + (size_t)(ok ? lr : le) * K)[l];   // q_e of the neighbour edge
            const double tx = t.x + w * Fx[i].x * (0.5 * (qo.x + qn.x)), ty = t.y + w * Fx[i].y * (0.5 * (qo.y + qn.y));
            t.x = (ok && ax) ? tx : t.x;
            t.y = (ok && ay) ? ty : t.y;
        }
        if (del2) {                                                     // horizontal_momentum_mixing.jl:75-78
            const int2 vo = reinterpret_cast<const int2 *>(m.voe)[e];
            const double invDv = 1.0 / m.dvEdge[e];
            const double2 d1 = ld2(nl.divc, c1, K, l), d2 = ld2(nl.divc, c2, K, l), z1 = ld2(nl.zv, vo.x, K, l), z2 = ld2(nl.zv, vo.y, K, l);
            if (ax) t.x += ((d2.x - d1.x) * invDc - (z2.x - z1.x) * invDv) * nl.visc;
            if (ay) t.y += ((d2.y - d1.y) * invDc - (z2.y - z1.y) * invDv) * nl.visc;
        }
        if (a.tendU) st2(a.tendU, e, K, l, t);
        if (a.pu_out) st2(a.pu_out, e, K, l, make_double2(ucur.x + a.a * t.x, ucur.y + a.a * t.y));
        if (a.nu_out) st2(a.nu_out, e, K, l, make_double2(nbu.x + a.b * t.x, nbu.y + a.b * t.y));
    }
}

static inline size_t nl4_lds_bytes(const MeshDev &m)
{
    return (size_t)m.maxRows * m.K * 8 + (size_t)m.maxOwnE * m.ME2 * 8 + (size_t)m.maxOwnE * 32 + (size_t)m.maxOwnC * (m.ME + 1) * 8 +
           (size_t)m.maxRows * 8 + (size_t)m.maxOwnE * m.ME2 * 4 + (size_t)m.maxOwnC * 2 * m.ME * 4 + (size_t)m.maxOwnE * 16;
}

// 1 = the patch form serves this mesh (even K <= 64, hexagon-dominated widths); NlArgs.fq then holds F alone
static inline bool nl3_ok(const MeshDev &m) { return m.K <= 64 && !(m.K & 1) && m.ME == 6 && m.ME2 == 10 && m.VD == 3 && m.patchVertStart && m.maxOwnE <= NL3_MAXE; }

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

hipError_t launch_nl_prepare(const MeshDev &m, const double *u, const double *h, const NlArgs &nl, int lpc, int form, hipStream_t s)
{
    if (lpc == 64 && nl3_ok(m) && form <= 1) {
        hipLaunchKernelGGL((k_nl_prep4<6, 3>), dim3(nl_grid(m.nPatches)), dim3(BLOCK), 0, s, m, u, h, nl);
        return hipGetLastError();
    }
    if (lpc == 64 && m.K <= 64 && !(m.K & 1) && form <= 2) {     // even 34 <= K <= 64: 16-byte lanes
        hipLaunchKernelGGL(k_nl_vertex2, grid2(m.nV), dim3(BLOCK), 0, s, m, u, h, nl.qv, nl.zv);
        hipLaunchKernelGGL(k_nl_cell2, grid2(m.nC), dim3(BLOCK), 0, s, m, u, nl.ke, nl.divc);
        hipLaunchKernelGGL(k_nl_edge2, grid2(m.nE), dim3(BLOCK), 0, s, m, u, h, nl);   // after k_nl_vertex2 (same stream)
        return hipGetLastError();
    }
#define CALL(L) launch_nl_prepare_lpc<L>(m, u, h, nl, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

hipError_t launch_stage_nl(const MeshDev &m, const StageArgs &a, const NlArgs &nl, int lpc, bool rowsOk, int form, hipStream_t s)
{
    if (lpc == 64 && nl3_ok(m) && rowsOk && form == 0 && nl4_lds_bytes(m) <= 80 * 1024) {     // two 512-thread workgroups per CU
        const size_t lds = nl4_lds_bytes(m);
        if (lds_attr_needed(3)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stage_nl4<6, 10, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((k_stage_nl4<6, 10, 512>), dim3(nl_grid(m.nPatches)), dim3(512), lds, s, m, a, nl);
        return hipGetLastError();
    }
    if (lpc == 64 && nl3_ok(m) && form <= 1) {
        hipLaunchKernelGGL((k_stage_nl3<6, 10>), dim3(nl_grid(m.nPatches)), dim3(BLOCK), 0, s, m, a, nl);
        return hipGetLastError();
    }
    if (lpc == 64 && m.K <= 64 && !(m.K & 1) && form <= 2) {
        hipLaunchKernelGGL(k_stage_nl2, grid2(std::max(m.nE, m.nC)), dim3(BLOCK), 0, s, m, a, nl);
        return hipGetLastError();
    }
#define CALL(L) launch_stage_nl_lpc<L>(m, a, nl, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

}  // namespace moka
