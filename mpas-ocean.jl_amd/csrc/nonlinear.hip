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
// launch shape of the patch form of the nonlinear stage kernel (moka_set_tuning key 5; identical results):
//   0 (default) k_stage_nl5 (vertex rows + own F rows in LDS), 512 threads; 1 k_stage_nl4 (q_e rows in LDS);
//   2 k_stage_nl5 with 256 threads; 3 k_stage_nl5 without the own F rows
static std::atomic<int> g_nlShape{0}, g_nlCapLimit{0};
void set_nl_shape(int v) { g_nlShape.store(v); }
int nl_shape() { return g_nlShape.load(); }
// key 6: upper limit of the vertex rows k_stage_nl5 keeps resident (0 = what the LDS budget holds): lets a test drive small meshes
// through the path of patches that list more vertices than fit
void set_nl_cap_limit(int v) { g_nlCapLimit.store(v); }
int nl_cap_limit() { return g_nlCapLimit.load(); }

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
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;       // (a launch may cover a patch range of a partitioned mesh)
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

// rows addressed by 32-bit byte offsets from a uniform base (the launcher checks that every field stays below 4 GiB)
__device__ __forceinline__ double2 ldo(const double *base, unsigned off) { return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + off); }
__device__ __forceinline__ void sto(double *base, unsigned off, double2 v) { *reinterpret_cast<double2 *>(reinterpret_cast<char *>(base) + off) = v; }

// k_nl_prep5 (round 3): k_nl_prep4 with the default stage kernel's row cache.  The normalVelocity rows of the patch's own edges and
// the layerThickness rows of its own cells are staged in LDS with the records (one phase); an entity then reads its cached rows in
// one burst of ds_read_b128 and overwrites the uncached lanes with exec-masked global loads (kernels_common.hpp, lds_burst).  Rows
// gathered from global memory per 16-cell patch: 432 -> ~190 (own: 70 % of a vertex's and 80 % of a cell's edges, 65 % of the cells).
template <int ME_, int VD_>
__global__ __launch_bounds__(BLOCK, 4) void k_nl_prep5(const MeshDev m, const double *__restrict__ u, const double *__restrict__ h, const NlArgs nl)
{
    constexpr int NG = BLOCK / 32;
    static_assert(ME_ == 6 && VD_ == 3, "burst widths");
    extern __shared__ __align__(16) unsigned char np5_smem[];
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31, K = m.K;
    const bool act = 2 * l < K;
    const unsigned rowB = (unsigned)K * 8u, lo = (unsigned)l * 16u;
    double *sU = reinterpret_cast<double *>(np5_smem);                      // [maxOwnE][K]  normalVelocity rows of the own edges
    double *sHh = sU + (size_t)m.maxOwnE * K;                               // [maxOwnC][K]  layerThickness rows of the own cells
    double *sVd = sHh + (size_t)m.maxOwnC * K;                              // [maxOwnV][2 VD + 2]  cv | kite | invAreaTri, fVertex
    double *sCd = sVd + (size_t)m.maxOwnV * (2 * VD_ + 2);                  // [maxOwnC][2 ME + 2]  sdv | keCoef | invArea, areaCell
    int2 *sEc = reinterpret_cast<int2 *>(sCd + (size_t)m.maxOwnC * (2 * ME_ + 2));   // [maxOwnE]  cellsOnEdge
    int *sVi = reinterpret_cast<int *>(sEc + m.maxOwnE);                    // [maxOwnV][2 VD]  edgesOnVertex | cellsOnVertex
    int *sCi = sVi + (size_t)m.maxOwnV * 2 * VD_;                           // [maxOwnC][ME]    edgesOnCell
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    const int v0 = m.patchVertStart[p], nv = m.patchVertStart[p + 1] - v0, c0 = m.patchCellStart[p], nc = m.patchCellStart[p + 1] - c0;
    const int e0 = m.patchEdgeStart[p], ne = m.patchEdgeStart[p + 1] - e0;
    for (int i = threadIdx.x; i < nv * VD_; i += BLOCK) {
        const int v = i / VD_, j = i % VD_;
        sVi[v * 2 * VD_ + j] = m.eov[(size_t)v0 * VD_ + i]; sVi[v * 2 * VD_ + VD_ + j] = m.cov[(size_t)v0 * VD_ + i];
        sVd[v * (2 * VD_ + 2) + j] = m.cv[(size_t)v0 * VD_ + i]; sVd[v * (2 * VD_ + 2) + VD_ + j] = m.kite[(size_t)v0 * VD_ + i];
    }
    for (int i = threadIdx.x; i < nv; i += BLOCK) {
        sVd[i * (2 * VD_ + 2) + 2 * VD_] = m.invAreaTri[v0 + i]; sVd[i * (2 * VD_ + 2) + 2 * VD_ + 1] = m.fVertex[v0 + i];
    }
    for (int i = threadIdx.x; i < nc * ME_; i += BLOCK) {
        const int c = i / ME_, j = i % ME_;
        sCi[i] = m.eoc[(size_t)c0 * ME_ + i];
        sCd[c * (2 * ME_ + 2) + j] = m.sdv[(size_t)c0 * ME_ + i];
        sCd[c * (2 * ME_ + 2) + ME_ + j] = m.keoc[(size_t)c0 * ME_ + i];
    }
    for (int i = threadIdx.x; i < nc; i += BLOCK) {
        sCd[i * (2 * ME_ + 2) + 2 * ME_] = m.invArea[c0 + i]; sCd[i * (2 * ME_ + 2) + 2 * ME_ + 1] = m.areaCell[c0 + i];
    }
    for (int i = threadIdx.x; i < ne; i += BLOCK) sEc[i] = reinterpret_cast<const int2 *>(m.ehdr)[2 * (size_t)(e0 + i)];
    if (act) {
        for (int r = grp; r < ne; r += 6 * NG) {                            // own rows: six in flight per half-wave
            double2 t[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) if (r + j * NG < ne) t[j] = ldo(u, (unsigned)(e0 + r + j * NG) * rowB + lo);
#pragma unroll
            for (int j = 0; j < 6; ++j) if (r + j * NG < ne) reinterpret_cast<double2 *>(sU + (size_t)(r + j * NG) * K)[l] = t[j];
        }
        for (int r = grp; r < nc; r += 2 * NG) {
            double2 t[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) if (r + j * NG < nc) t[j] = ldo(h, (unsigned)(c0 + r + j * NG) * rowB + lo);
#pragma unroll
            for (int j = 0; j < 2; ++j) if (r + j * NG < nc) reinterpret_cast<double2 *>(sHh + (size_t)(r + j * NG) * K)[l] = t[j];
        }
    }
    __syncthreads();
    if (!act) return;
    const glb_bytes_t uG = (glb_bytes_t)u, hG = (glb_bytes_t)h;
    const uint32_t ldsU = (uint32_t)(size_t)sU + lo, ldsH = (uint32_t)(size_t)sHh + lo;
    // cached row or row 0 of the cache (then overwritten by the masked global load); goff stays in a VGPR (see k_stage_rec2c)
    auto urow = [&](int e, bool &cached, uint32_t &goff) -> uint32_t {
        const unsigned loc = (unsigned)(e - e0);
        cached = loc < (unsigned)ne;
        goff = (unsigned)e * rowB + lo;
        asm("" : "+v"(goff));
        return ldsU + (cached ? loc * rowB : 0u);
    };
    auto hrow = [&](int c, bool &cached, uint32_t &goff) -> uint32_t {
        const unsigned loc = (unsigned)(c - c0);
        cached = loc < (unsigned)nc;
        goff = (unsigned)c * rowB + lo;
        asm("" : "+v"(goff));
        return ldsH + (cached ? loc * rowB : 0u);
    };
    for (int vi = grp; vi < nv; vi += 2 * NG) {                           // two vertices per round
        double2 uu[2][VD_], hh[2][VD_];
        {
            bool cu[2 * VD_], chh[2 * VD_];
            uint32_t au[2 * VD_], ah[2 * VD_], gu[2 * VD_], gh[2 * VD_];
            v4u_t ru[2 * VD_], rh[2 * VD_];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int v = vi + q * NG < nv ? vi + q * NG : vi;
#pragma unroll
                for (int j = 0; j < VD_; ++j) {
                    au[q * VD_ + j] = urow(sVi[v * 2 * VD_ + j], cu[q * VD_ + j], gu[q * VD_ + j]);
                    ah[q * VD_ + j] = hrow(sVi[v * 2 * VD_ + VD_ + j], chh[q * VD_ + j], gh[q * VD_ + j]);
                }
            }
            lds_burst<2 * VD_>(ru, au);
            lds_burst<2 * VD_>(rh, ah);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < VD_; ++j) {
                    uu[q][j] = __builtin_bit_cast(double2, ru[q * VD_ + j]);
                    if (!cu[q * VD_ + j]) uu[q][j] = glb_row2(uG + gu[q * VD_ + j]);
                    hh[q][j] = __builtin_bit_cast(double2, rh[q * VD_ + j]);
                    if (!chh[q * VD_ + j]) hh[q][j] = glb_row2(hG + gh[q * VD_ + j]);
                }
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
            sto(nl.qv, (unsigned)(v0 + v) * rowB + lo, make_double2((fv + zeta.x) / hv.x, (fv + zeta.y) / hv.y));
            if (nl.zv) sto(nl.zv, (unsigned)(v0 + v) * rowB + lo, zeta);
        }
    }
    for (int ci = grp; ci < nc; ci += 2 * NG) {                           // two cells per round
        double2 ue[2][ME_];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = ci + q * NG < nc ? ci + q * NG : ci;
            bool cu[ME_];
            uint32_t au[ME_], gu[ME_];
            v4u_t ru[ME_];
#pragma unroll
            for (int i = 0; i < ME_; ++i) { const int e = sCi[c * ME_ + i]; au[i] = urow(e < 0 ? e0 : e, cu[i], gu[i]); }
            lds_burst<ME_>(ru, au);
#pragma unroll
            for (int i = 0; i < ME_; ++i) {
                ue[q][i] = __builtin_bit_cast(double2, ru[i]);
                if (!cu[i]) ue[q][i] = glb_row2(uG + gu[i]);
            }
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
            sto(nl.ke, (unsigned)(c0 + c) * rowB + lo, make_double2(acc.x * invA, acc.y * invA));
            if (nl.divc) sto(nl.divc, (unsigned)(c0 + c) * rowB + lo, make_double2(d.x / area, d.y / area));
        }
    }
    for (int ei = grp; ei < ne; ei += 3 * NG) {                           // three edges per round
        double2 h1[3], h2[3], uu[3];
        {
            bool chh[6];
            uint32_t ah[6], gh[6];
            v4u_t rh[6];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = ei + q * NG < ne ? ei + q * NG : ei;
                const int2 cc = sEc[e];
                ah[2 * q] = hrow(cc.x, chh[2 * q], gh[2 * q]); ah[2 * q + 1] = hrow(cc.y, chh[2 * q + 1], gh[2 * q + 1]);
                uu[q] = reinterpret_cast<const double2 *>(sU + (size_t)e * K)[l];
            }
            lds_burst<6>(rh, ah);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                h1[q] = __builtin_bit_cast(double2, rh[2 * q]);
                if (!chh[2 * q]) h1[q] = glb_row2(hG + gh[2 * q]);
                h2[q] = __builtin_bit_cast(double2, rh[2 * q + 1]);
                if (!chh[2 * q + 1]) h2[q] = glb_row2(hG + gh[2 * q + 1]);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = ei + q * NG;
            if (e >= ne) break;
            sto(nl.fq, (unsigned)(e0 + e) * rowB + lo, make_double2(uu[q].x * (0.5 * (h1[q].x + h2[q].x)), uu[q].y * (0.5 * (h1[q].y + h2[q].y))));   // Operators.jl:217, DiagnosticVars.jl:165
        }
    }
}

static inline size_t np5_lds_bytes(const MeshDev &m)
{
    return ((size_t)m.maxOwnE + m.maxOwnC) * m.K * 8 + (size_t)m.maxOwnV * (2 * m.VD + 2) * 8 + (size_t)m.maxOwnC * (2 * m.ME + 2) * 8 +
           (size_t)m.maxOwnE * 8 + (size_t)m.maxOwnV * 2 * m.VD * 4 + (size_t)m.maxOwnC * m.ME * 4;
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
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;       // (a launch may cover a patch range of a partitioned mesh)
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
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;       // (a launch may cover a patch range of a partitioned mesh)
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

// k_stage_nl4 holds q_e of every edge row a patch touches (~110-135 rows, 53-65 KB): a chain of four barrier-separated bursts at two
// workgroups per CU (profiles/r02_variants.txt).  Round 3, k_stage_nl5:
//   * the potential vorticity of the patch's VERTICES in LDS instead (~80 rows: the vertices of the own edges and of their
//     edgesOnEdge; plan: pvList / lvoe), the two vertex rows of an edge averaged on use -- the same 0.5 * (qa + qb), the same bits;
//     a third of the qv gathers (every vertex once instead of twice per edge row);
//   * CF: the thickness flux of the patch's OWN edges in LDS too (the default stage kernel's row cache): 45 % of the ten F rows an
//     edge gathers are rows of its own patch;
//   * the qv / F row loads are issued in front of the cell loop and land in LDS behind it: the two phases share one round trip;
//   * 32-bit byte offsets from uniform bases (one address register per gather).
// Measured (profiles/r03_variants.txt, config 4): 3.2-3.3 ms per launch for k_stage_nl4 -> 2.9-3.1 here; the vertex rows bring
// 7 %, the F row cache another 1.4 % (a gather that hits the L2 is cheap: rows gathered from memory per patch, 880 -> 660, do not
// predict the time).  Three workgroups per CU (512 threads bounded to 80 registers, the F gathers in two batches) were built and
// measured, too: no faster than two with all ten gathers in one batch.
template <int ME_, int ME2_, int NT, int MINW, bool CF>
__global__ __launch_bounds__(NT, MINW) void k_stage_nl5(const MeshDev m, const StageArgs a, const NlArgs nl)
{
    constexpr int NG = NT / 32;
    constexpr int RB = (80 + NG - 1) / NG;           // vertex rows in flight per half-wave: one round serves 80 vertices
    constexpr int FB = (48 + NG - 1) / NG;           // own F rows in flight per half-wave: one round serves 48 edges
    static_assert(ME2_ == 10, "lvoe's layout: ten edgesOnEdge slots, then the edge itself");
    extern __shared__ __align__(16) unsigned char nl5_smem[];
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31, K = m.K, k0 = 2 * l;
    const bool act = k0 < K;
    const unsigned rowB = (unsigned)K * 8u, lo = (unsigned)l * 16u;   // 32-bit byte offsets: one address register per gather
    double *sQ = reinterpret_cast<double *>(nl5_smem);                     // [pvCap][K]      q_v rows of the patch's first pvCap vertices
    double *sF = sQ + (size_t)m.pvCap * K;                                 // [maxOwnE][K]    F rows of the own edges (CF)
    double *sW = sF + (CF ? (size_t)m.maxOwnE * K : 0);                    // [maxOwnE][ME2]  weightsOnEdge
    int4 *sH = reinterpret_cast<int4 *>(sW + (size_t)m.maxOwnE * ME2_);    // [maxOwnE]       {c1, c2, nEdgesOnEdge, maxLevelEdgeTop}
    double2 *sG = reinterpret_cast<double2 *>(sH + m.maxOwnE);             // [maxOwnE]       {g / dcEdge, 1 / dcEdge}
    double *sCs = reinterpret_cast<double *>(sG + m.maxOwnE);              // [maxOwnC][ME+1] sdv | invArea
    int *sX = reinterpret_cast<int *>(sCs + (size_t)m.maxOwnC * (ME_ + 1));   // [maxOwnE][ME2]  edgesOnEdge (global ids, -1 = none)
    int *sCe = sX + (size_t)m.maxOwnE * ME2_;                              // [maxOwnC][2 ME] edgesOnCell | maxLevelEdgeTop of the edge
    int *sPv = sCe + (size_t)m.maxOwnC * 2 * ME_;                          // [maxPV]         the patch's vertices
    unsigned *sLv = reinterpret_cast<unsigned *>(sPv + m.maxPV);           // [maxOwnE][12]   patch-local vertex ids, 16 bits each (plan.cpp: lvoe)
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    const double *__restrict__ F = nl.fq;
    const glb_bytes_t FG = (glb_bytes_t)nl.fq;
    const int e0 = m.patchEdgeStart[p], e1 = m.patchEdgeStart[p + 1], nOwn = e1 - e0;
    const int pv0 = m.pvStart[p], nPV = m.pvStart[p + 1] - pv0;
    const int nLds = min(nPV, m.pvCap);       // the few patches with more vertices than the LDS budget holds gather the rest on use
    for (int i = threadIdx.x; i < nPV; i += NT) sPv[i] = m.pvList[pv0 + i];
    for (int i = threadIdx.x; i < nOwn * ME2_; i += NT) { sX[i] = m.eoe[(size_t)e0 * ME2_ + i]; sW[i] = m.woe[(size_t)e0 * ME2_ + i]; }
    for (int i = threadIdx.x; i < nOwn * 12; i += NT) sLv[i] = reinterpret_cast<const unsigned *>(m.lvoe)[(size_t)e0 * 12 + i];
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
    double2 qr[RB], fr[CF ? FB : 1];                                      // first round of rows: in flight across the cell loop
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int rr = grp + j * NG;
        qr[j] = make_double2(0.0, 0.0);
        if (act && rr < nLds) qr[j] = ldo(nl.qv, (unsigned)(sPv[rr]) * rowB + lo);
    }
    if (CF) {
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            const int rr = grp + j * NG;
            fr[j] = make_double2(0.0, 0.0);
            if (act && rr < nOwn) fr[j] = ldo(F, (unsigned)(e0 + rr) * rowB + lo);
        }
    }
    for (int c = c0 + grp; c < c0 + nC; c += NG) {
        double2 hs = make_double2(0.0, 0.0);
        if (act) {
            double2 Fe[ME_];
#pragma unroll
            for (int i = 0; i < ME_; ++i) { const int ei = sCe[(c - c0) * 2 * ME_ + i]; Fe[i] = ldo(F, (unsigned)(ei < 0 ? 0 : ei) * rowB + lo); }
            const double invA = sCs[(c - c0) * (ME_ + 1) + ME_];
            const double2 hcur = a.ch ? ldo(a.ch, (unsigned)(c) * rowB + lo) : ldo(a.ph, (unsigned)(c) * rowB + lo);
            double2 nb = hcur;
            if (a.nh_out && a.nh_in) nb = ldo(a.nh_in, (unsigned)(c) * rowB + lo);
            double2 t = make_double2(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < ME_; ++i) {
                const int ei = sCe[(c - c0) * 2 * ME_ + i], ml = sCe[(c - c0) * 2 * ME_ + ME_ + i];
                const double sd = sCs[(c - c0) * (ME_ + 1) + i];
                const double tx = t.x + Fe[i].x * sd * invA, ty = t.y + Fe[i].y * sd * invA;   // horizontal_advection.jl:63-64
                t.x = (ei >= 0 && k0 < ml) ? tx : t.x;
                t.y = (ei >= 0 && k0 + 1 < ml) ? ty : t.y;
            }
            if (a.tendH) sto(a.tendH, (unsigned)(c) * rowB + lo, t);
            if (a.ph_out) {
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);
                sto(a.ph_out, (unsigned)(c) * rowB + lo, hs);
            }
            if (a.nh_out) {
                double2 hn = make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y);
                if (a.rkMode == 9) {                                      // 13-stream form, stage 4: New from Curr, P2 (nb), P3, P4 (own rows)
                    const double2 q3 = ldo(a.q3h, (unsigned)(c) * rowB + lo), p4 = ldo(a.ph, (unsigned)(c) * rowB + lo);
                    hn = make_double2(rk13_combine(hcur.x, nb.x, q3.x, p4.x, a.b, t.x), rk13_combine(hcur.y, nb.y, q3.y, p4.y, a.b, t.y));
                }
                sto(a.nh_out, (unsigned)(c) * rowB + lo, hn);
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
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int rr = grp + j * NG;
        if (act && rr < nLds) reinterpret_cast<double2 *>(sQ + (size_t)rr * K)[l] = qr[j];
    }
    if (CF) {
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            const int rr = grp + j * NG;
            if (act && rr < nOwn) reinterpret_cast<double2 *>(sF + (size_t)rr * K)[l] = fr[j];
        }
    }
    for (int r = grp + NG * RB; r < nLds; r += NG * RB) {                 // patches with more than 80 vertices (irregular regions)
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int rr = r + j * NG;
            if (act && rr < nLds) qr[j] = ldo(nl.qv, (unsigned)(sPv[rr]) * rowB + lo);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int rr = r + j * NG;
            if (act && rr < nLds) reinterpret_cast<double2 *>(sQ + (size_t)rr * K)[l] = qr[j];
        }
    }
    if (CF) {
        for (int rr = grp + NG * FB; rr < nOwn; rr += NG)                 // (patches that own more than 48 edges)
            if (act) reinterpret_cast<double2 *>(sF + (size_t)rr * K)[l] = ldo(F, (unsigned)(e0 + rr) * rowB + lo);
    }
    __syncthreads();
    if (!act) return;
    const bool del2 = nl.zv != nullptr;
    const bool over = nPV > nLds;     // uniform: this patch lists more vertices than the LDS budget holds
    const uint32_t ldsF = (uint32_t)(size_t)sF + lo;
    // Stores are deferred by one iteration (as in k_stage_rec2c): an edge's results are written during the NEXT iteration, right
    // after that iteration's loads have arrived, so that their acknowledgement overlaps the arithmetic instead of delaying the loads.
    bool pend = false;
    unsigned pOff = 0;
    double2 pT = make_double2(0.0, 0.0), pA = pT, pB = pT;
    auto flush = [&]() {
        if (a.tendU) sto(a.tendU, pOff, pT);
        if (a.pu_out) sto(a.pu_out, pOff, pA);
        if (a.nu_out) sto(a.nu_out, pOff, pB);
    };
    for (int e = e0 + grp; e < e1; e += NG) {
        const int le = e - e0;
        auto qrow = [&](unsigned id) -> double2 { return reinterpret_cast<const double2 *>(sQ + (size_t)id * K)[l]; };   // resident q_v row
        auto qany = [&](unsigned id) -> double2 {                         // q_v row of any of the patch's vertices
            if ((int)id >= nLds) return ldo(nl.qv, (unsigned)sPv[id] * rowB + lo);
            return reinterpret_cast<const double2 *>(sQ + (size_t)id * K)[l];
        };
        const int4 hd = sH[le];
        const int c1 = hd.x, c2 = hd.y, mlt = hd.w;
        const double g = sG[le].x, invDc = sG[le].y;
        const double ds = a.ssh[c2] - a.ssh[c1];
        const double2 k1 = ldo(nl.ke, (unsigned)(c1) * rowB + lo), k2 = ldo(nl.ke, (unsigned)(c2) * rowB + lo);
        const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
        double2 t = make_double2(0.0, 0.0);
        double2 qo;
        {
            const unsigned w = sLv[le * 12 + 10];
            const double2 qa = over ? qany(w & 0xFFFFu) : qrow(w & 0xFFFFu), qb = over ? qany(w >> 16) : qrow(w >> 16);
            qo = make_double2(0.5 * (qa.x + qb.x), 0.5 * (qa.y + qb.y));   // q_e of this edge
        }
        double2 ucur, nbu;
        if (over) {                                                       // plain slot-by-slot form, few registers: rare patches only
            if (pend) flush();
            if (ax) t.x -= g * ds;
            if (ay) t.y -= g * ds;
            if (ax) t.x -= invDc * (k2.x - k1.x);
            if (ay) t.y -= invDc * (k2.y - k1.y);
            const unsigned short *lb = reinterpret_cast<const unsigned short *>(sLv + le * 12);
#pragma unroll 1
            for (int i = 0; i < ME2_; ++i) {
                const int x = sX[le * ME2_ + i];
                const double2 Fi = ldo(F, (unsigned)(x < 0 ? e : x) * rowB + lo);
                const double2 qa = qany(lb[2 * i]), qb = qany(lb[2 * i + 1]);
                const double w = sW[le * ME2_ + i];
                const double qnx = 0.5 * (qa.x + qb.x), qny = 0.5 * (qa.y + qb.y);
                const double tx = t.x + w * Fi.x * (0.5 * (qo.x + qnx)), ty = t.y + w * Fi.y * (0.5 * (qo.y + qny));
                t.x = (x >= 0 && ax) ? tx : t.x;
                t.y = (x >= 0 && ay) ? ty : t.y;
            }
            ucur = a.cu ? ldo(a.cu, (unsigned)(e) * rowB + lo) : ldo(a.pu, (unsigned)(e) * rowB + lo);
            nbu = ucur;
            if (a.nu_out && a.nu_in) nbu = ldo(a.nu_in, (unsigned)(e) * rowB + lo);
        } else {
            double2 Fx[ME2_];
            if (CF) {           // rows of the own patch from LDS in one burst, the others overwritten by exec-masked global loads
                bool cached[ME2_];
                uint32_t ad[ME2_], goff[ME2_];
                v4u_t raw[ME2_];
#pragma unroll
                for (int i = 0; i < ME2_; ++i) {
                    const int x = sX[le * ME2_ + i];
                    const unsigned loc = (unsigned)((x < 0 ? e : x) - e0);
                    cached[i] = loc < (unsigned)nOwn;
                    ad[i] = ldsF + (cached[i] ? loc * rowB : 0u);
                    goff[i] = (unsigned)(x < 0 ? e : x) * rowB + lo;
                    asm("" : "+v"(goff[i]));
                }
                lds_burst<ME2_>(raw, ad);
#pragma unroll
                for (int i = 0; i < ME2_; ++i) {
                    Fx[i] = __builtin_bit_cast(double2, raw[i]);
                    if (!cached[i]) Fx[i] = glb_row2(FG + goff[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < ME2_; ++i) { const int x = sX[le * ME2_ + i]; Fx[i] = ldo(F, (unsigned)(x < 0 ? e : x) * rowB + lo); }
            }
            ucur = a.cu ? ldo(a.cu, (unsigned)(e) * rowB + lo) : ldo(a.pu, (unsigned)(e) * rowB + lo);
            nbu = ucur;
            if (a.nu_out && a.nu_in) nbu = ldo(a.nu_in, (unsigned)(e) * rowB + lo);
            __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0): this iteration's loads (needed now anyway) ...
            if (pend) flush();                                            // ... so that the stores queue up behind them, not ahead
            if (ax) t.x -= g * ds;
            if (ay) t.y -= g * ds;
            if (ax) t.x -= invDc * (k2.x - k1.x);
            if (ay) t.y -= invDc * (k2.y - k1.y);

#pragma unroll
            for (int i = 0; i < ME2_; ++i) {
                const unsigned lvi = sLv[le * 12 + i];
                const unsigned ia = lvi & 0xFFFFu, ib = lvi >> 16;
                const bool ok = sX[le * ME2_ + i] >= 0;
                const double w = sW[le * ME2_ + i];
                const double2 qa = qrow(ia), qb = qrow(ib);
                const double qnx = 0.5 * (qa.x + qb.x), qny = 0.5 * (qa.y + qb.y);                  // q_e of the neighbour edge
                const double tx = t.x + w * Fx[i].x * (0.5 * (qo.x + qnx)), ty = t.y + w * Fx[i].y * (0.5 * (qo.y + qny));
                t.x = (ok && ax) ? tx : t.x;
                t.y = (ok && ay) ? ty : t.y;
            }
        }
        if (del2) {                                                     // horizontal_momentum_mixing.jl:75-78
            const int2 vo = reinterpret_cast<const int2 *>(m.voe)[e];
            const double invDv = 1.0 / m.dvEdge[e];
            const double2 d1 = ldo(nl.divc, (unsigned)(c1) * rowB + lo), d2 = ldo(nl.divc, (unsigned)(c2) * rowB + lo), z1 = ldo(nl.zv, (unsigned)(vo.x) * rowB + lo), z2 = ldo(nl.zv, (unsigned)(vo.y) * rowB + lo);
            if (ax) t.x += ((d2.x - d1.x) * invDc - (z2.x - z1.x) * invDv) * nl.visc;
            if (ay) t.y += ((d2.y - d1.y) * invDc - (z2.y - z1.y) * invDv) * nl.visc;
        }
        pOff = (unsigned)(e) * rowB + lo;
        pT = t;
        pA = make_double2(ucur.x + a.a * t.x, ucur.y + a.a * t.y);
        pB = make_double2(nbu.x + a.b * t.x, nbu.y + a.b * t.y);
        if (a.rkMode == 9) {                                              // 13-stream form, stage 4 (own rows; no gathers)
            const double2 q3 = ldo(a.q3u, pOff), p4 = ldo(a.pu, pOff);
            pB = make_double2(rk13_combine(ucur.x, nbu.x, q3.x, p4.x, a.b, t.x), rk13_combine(ucur.y, nbu.y, q3.y, p4.y, a.b, t.y));
        }
        pend = true;
    }
    if (pend) flush();
}

// dynamic LDS of k_stage_nl5 with `cap` vertex rows resident (cf: and the F rows of the own edges)
static inline size_t nl5_lds_bytes(const MeshDev &m, int cap, bool cf)
{
    return (size_t)cap * m.K * 8 + (cf ? (size_t)m.maxOwnE * m.K * 8 : 0) + (size_t)m.maxOwnE * m.ME2 * 8 + (size_t)m.maxOwnE * 32 +
           (size_t)m.maxOwnC * (m.ME + 1) * 8 + (size_t)m.maxOwnE * m.ME2 * 4 + (size_t)m.maxOwnC * 2 * m.ME * 4 + (size_t)m.maxPV * 4 +
           (size_t)m.maxOwnE * 48;
}

// vertex rows kept resident: all of the largest patch's when two workgroups per CU still fit, otherwise what that budget holds -- the
// few larger patches (config 4: 80 vertices on average, 101 at most) gather the rest on use
static inline int nl5_cap(const MeshDev &m, bool cf)
{
    const size_t budget = 80 * 1024, rec = nl5_lds_bytes(m, 0, cf);
    if (rec >= budget) return 0;
    int cap = (int)std::min<size_t>((size_t)m.maxPV, (budget - rec) / ((size_t)m.K * 8));
    if (const int lim = g_nlCapLimit.load(); lim > 0) cap = std::min(cap, lim);
    return cap;
}

static inline size_t nl4_lds_bytes(const MeshDev &m)
{
    return (size_t)m.maxRows * m.K * 8 + (size_t)m.maxOwnE * m.ME2 * 8 + (size_t)m.maxOwnE * 32 + (size_t)m.maxOwnC * (m.ME + 1) * 8 +
           (size_t)m.maxRows * 8 + (size_t)m.maxOwnE * m.ME2 * 4 + (size_t)m.maxOwnC * 2 * m.ME * 4 + (size_t)m.maxOwnE * 16;
}

// 1 = the patch form serves this mesh (even K <= 64, hexagon-dominated widths); NlArgs.fq then holds F alone
static inline bool nl3_ok(const MeshDev &m) { return m.K <= 64 && !(m.K & 1) && m.ME == 6 && m.ME2 == 10 && m.VD == 3 && m.patchVertStart && m.maxOwnE <= NL3_MAXE; }

bool nl_patch_forms(const MeshDev &m, int lpc, int form) { return lpc == 64 && nl3_ok(m) && form <= 1; }

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
    if (lpc == 64 && nl3_ok(m) && form == 0 && g_nlShape.load() != 1 && m.maxOwnV > 0 && np5_lds_bytes(m) <= 64 * 1024 &&
        (uint64_t)std::max(m.nE, std::max(m.nV, m.nC)) * m.K * 8 < (1ull << 32)) {
        hipLaunchKernelGGL((k_nl_prep5<6, 3>), dim3(nl_grid(m.nPatches)), dim3(BLOCK), np5_lds_bytes(m), s, m, u, h, nl);
        return hipGetLastError();
    }
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

// the stage launch goes through k_stage_nl5 (the only nonlinear stage kernel that knows the 13-stream form: StageArgs.rkMode 9)
bool nl_stage_is_nl5(const MeshDev &m, int lpc, int form)
{
    const int shape = g_nlShape.load();
    if (!(lpc == 64 && nl3_ok(m) && form == 0 && shape != 1 && m.pvStart && m.maxPV > 0 &&
          (uint64_t)std::max(m.nE, std::max(m.nV, m.nC)) * m.K * 8 < (1ull << 32)))
        return false;
    const bool cf = shape != 3 && nl5_cap(m, true) >= std::min(m.maxPV, 64);
    return nl5_cap(m, cf) >= 16;
}

hipError_t launch_stage_nl(const MeshDev &m, const StageArgs &a, const NlArgs &nl, int lpc, bool rowsOk, int form, hipStream_t s)
{
    const int shape = g_nlShape.load();
    if (lpc == 64 && nl3_ok(m) && form == 0 && shape != 1 && m.pvStart && m.maxPV > 0 &&
        (uint64_t)std::max(m.nE, std::max(m.nV, m.nC)) * m.K * 8 < (1ull << 32)) {
        // own F rows resident too where that leaves room for most of a patch's vertex rows (launches that carry the halo-straddling
        // patches of a partitioned mesh, with up to 6 own edges per cell, go without)
        const bool cf = shape != 3 && nl5_cap(m, true) >= std::min(m.maxPV, 64);
        MeshDev mc = m;
        mc.pvCap = nl5_cap(m, cf);
        if (mc.pvCap >= 16) {
            const size_t lds = nl5_lds_bytes(mc, mc.pvCap, cf);
            const int slot = 17 + (shape == 2 ? 2 : 0) + (cf ? 1 : 0);
            const void *fn = shape == 2 ? (cf ? reinterpret_cast<const void *>(k_stage_nl5<6, 10, 256, 3, true>) : reinterpret_cast<const void *>(k_stage_nl5<6, 10, 256, 3, false>))
                                        : (cf ? reinterpret_cast<const void *>(k_stage_nl5<6, 10, 512, 4, true>) : reinterpret_cast<const void *>(k_stage_nl5<6, 10, 512, 4, false>));
            if (lds > 64 * 1024 && lds_attr_needed(slot)) {
                hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                if (e != hipSuccess) return e;
            }
            const dim3 g(nl_grid(m.nPatches));
            if (shape == 2 && cf) hipLaunchKernelGGL((k_stage_nl5<6, 10, 256, 3, true>), g, dim3(256), lds, s, mc, a, nl);
            else if (shape == 2) hipLaunchKernelGGL((k_stage_nl5<6, 10, 256, 3, false>), g, dim3(256), lds, s, mc, a, nl);
            else if (cf) hipLaunchKernelGGL((k_stage_nl5<6, 10, 512, 4, true>), g, dim3(512), lds, s, mc, a, nl);
            else hipLaunchKernelGGL((k_stage_nl5<6, 10, 512, 4, false>), g, dim3(512), lds, s, mc, a, nl);
            return hipGetLastError();
        }
    }
    if (a.rkMode == 9) return hipErrorNotSupported;      // the 13-stream form's last stage exists in k_stage_nl5 only (mk::rk13_usable asks nl_stage_is_nl5 first)
    if (lpc == 64 && nl3_ok(m) && rowsOk && form == 0 && nl4_lds_bytes(m) <= 80 * 1024) {     // two 512-thread workgroups per CU
        const size_t lds = nl4_lds_bytes(m);
        if (lds_attr_needed(16)) {
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
