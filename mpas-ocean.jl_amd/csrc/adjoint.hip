// adjoint.hip -- reverse-mode kernels of libmoka_hip (gfx950): transposes of the Forward-Euler step and of the
// tendency evaluation, and the element-wise helpers of the RK4 reverse sweep.
#include "kernels_common.hpp"

namespace moka {

// ------------------------------------------------------------------------------------------------
// Reverse mode of one Forward-Euler step (reference: Enzyme over ocn_run_loop, ext/MPASEnzymeExt.jl and
// test/enzyme/test_Enzyme_end2end.jl; here the hand transposition, see oracle_step_fe_adjoint for the algebra).
// Gather form -- no atomics: an edge gathers the cell adjoints of its two cells and the velocity adjoints of the
// edges whose Coriolis stencil contains it (transposed lists built at tape creation); a cell gathers from its edges.
// LPC lanes span a column; every sum runs in the oracle's order, so the results are bit-identical to it.
// ------------------------------------------------------------------------------------------------
// TT = false: transpose of one Forward-Euler step (identity parts included, dt folded in, ssh a state variable);
// TT = true : transpose of the tendency evaluation alone, (outU, outH) = T'(u,h)^T (kU, kH) -- the RK4 building block
//             (kU = lamU1, kH = lamH1; layerThicknessEdge is recomputed from the stage's h; ssh's adjoint goes into h).
template <int LPC, bool TT>
__global__ __launch_bounds__(BLOCK) void k_adj_edge(const AdjMesh m, const AdjArgs a)
{
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC), l = threadIdx.x % LPC;
    const int K = m.K;
    const int Kc = ((K + LPC - 1) / LPC) * LPC;
    for (int e = blockIdx.x * NG + grp; e < m.nE; e += gridDim.x * NG) {
        const int c1 = cptr(m.ehdr)[(size_t)e * 4], c2 = cptr(m.ehdr)[(size_t)e * 4 + 1], mlt = cptr(m.ehdr)[(size_t)e * 4 + 3];
        const double sd1 = cptr(m.sd)[(size_t)e * 2], sd2 = cptr(m.sd)[(size_t)e * 2 + 1];
        const double fe = cptr(m.fEdge)[e];
        double s1 = 0.0, s2 = 0.0;
        if constexpr (!TT) { s1 = a.lamS1[c1]; s2 = a.lamS1[c2]; }
        double acc = 0.0;
        bool first = true;
        for (int k = l; k < Kc; k += LPC) {
            double tu = 0.0;
            if (k < K) {
                const size_t off = (size_t)e * K + k;
                double Fbar = 0.0;
                if (k < mlt) {
                    if constexpr (TT) {
                        Fbar = sd1 * a.lamH1[(size_t)c1 * K + k] + sd2 * a.lamH1[(size_t)c2 * K + k];
                    } else {
                        const double tH1 = a.dt * (a.lamH1[(size_t)c1 * K + k] + s1);
                        const double tH2 = a.dt * (a.lamH1[(size_t)c2 * K + k] + s2);
                        Fbar = sd1 * tH1 + sd2 * tH2;
                    }
                }
                double cor = 0.0;
                for (int j = 0; j < m.W; ++j) {
                    const int s = cptr(m.teoe)[(size_t)e * m.W + j];
                    if (s < 0 || k >= cptr(m.ehdr)[(size_t)s * 4 + 3]) continue;
                    if constexpr (TT) cor += (cptr(m.tw)[(size_t)e * m.W + j] * fe) * a.lamU1[(size_t)s * K + k];
                    else cor += (cptr(m.tw)[(size_t)e * m.W + j] * fe) * (a.dt * a.lamU1[(size_t)s * K + k]);
                }
                const double lu = a.lamU1[off];
                if constexpr (TT) {
                    const double hI = 0.5 * (a.h[(size_t)c1 * K + k] + a.h[(size_t)c2 * K + k]);    // Operators.jl:217
                    const double pb = hI * Fbar + cor;
                    if (a.accOutU) {                      // fused element-wise steps of the RK4 reverse sweep
                        const double x = a.xU[off];
                        a.accOutU[off] = (a.accInU ? a.accInU[off] : x) + pb;
                        if (a.kNextU) a.kNextU[off] = a.cbNext * x + a.caNext * pb;
                    } else {
                        a.lamU0[off] = pb;
                    }
                } else {
                    a.lamU0[off] = (lu + a.hEuse[off] * Fbar) + cor;
                }
                a.Enew[off] = a.u[off] * Fbar;
                if (k < mlt) tu = TT ? lu : a.dt * lu;
            }
            acc = first ? tu : acc + tu;                  // oracle_ksum: lane partial sums, then the butterfly
            first = false;
        }
        const double cs = group_sum<LPC>(acc);
        if (l == 0) a.csum[e] = cs;
    }
}

template <int LPC, bool TT>
__global__ __launch_bounds__(BLOCK) void k_adj_cell(const AdjMesh m, const AdjArgs a)
{
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC), l = threadIdx.x % LPC;
    const int K = m.K, ME = m.ME;
    const double *Eread = (TT || !a.stale) ? a.Enew : a.lamE1;
    for (int c = blockIdx.x * NG + grp; c < m.nC; c += gridDim.x * NG) {
        CP<int32_t> re = cptr(m.eoc) + (size_t)c * ME;
        double ls = 0.0;                                  // every lane: TT adds it to each level
        for (int i = 0; i < ME; ++i) {
            const int e = re[i];
            if (e < 0) continue;
            ls += (-(double)cptr(m.csgn)[(size_t)c * ME + i]) * cptr(m.gInvDc)[e] * a.csum[e];
        }
        double s1 = 0.0;
        if constexpr (!TT) {
            s1 = a.lamS1[c];
            if (l == 0) a.lamS0[c] = ls;
        }
        for (int k = l; k < K; k += LPC) {
            double acc = 0.0;
            for (int i = 0; i < ME; ++i) {
                const int e = re[i];
                if (e >= 0) acc += Eread[(size_t)e * K + k];
            }
            if constexpr (TT) {
                const double pb = 0.5 * acc + ls;
                const size_t off = (size_t)c * K + k;
                if (a.accOutH) {
                    const double x = a.xH[off];
                    a.accOutH[off] = (a.accInH ? a.accInH[off] : x) + pb;
                    if (a.kNextH) a.kNextH[off] = a.cbNext * x + a.caNext * pb;
                } else {
                    a.lamH0[off] = pb;
                }
            } else {
                a.lamH0[(size_t)c * K + k] = (a.lamH1[(size_t)c * K + k] + s1) + 0.5 * acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same two kernels for even K <= 64 with 16-byte lanes: half a wave per entity, a lane owns levels 2l and 2l+1 (the
// layout of the forward stage kernel), every sum in the same order -- bit-identical to k_adj_edge / k_adj_cell.  The
// column sum csum follows oracle_ksum: butterfly over the lanes for each of the two levels, then their sum.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 ld2(const double *p, size_t row, int K, int l)
{
    return reinterpret_cast<const double2 *>(p + row * K)[l];
}
__device__ __forceinline__ void st2(double *p, size_t row, int K, int l, double2 v)
{
    reinterpret_cast<double2 *>(p + row * K)[l] = v;
}

// WF > 0: the transposed lists have exactly WF slots (10 on hexagonal meshes): the regular-edge path is fully unrolled,
// its index / weight loads and its WF row gathers go out as batches (no per-source maxLevelEdgeTop lookups, no masks)
template <bool TT, int WF>
__global__ __launch_bounds__(BLOCK) void k_adj_edge2(const AdjMesh m, const AdjArgs a)
{
    constexpr int NG = BLOCK / 32;
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int K = m.K, k0 = 2 * l;
    const bool act = k0 < K;
    for (int e = blockIdx.x * NG + grp; e < m.nE; e += gridDim.x * NG) {
        const int c1 = m.ehdr[(size_t)e * 4], c2 = m.ehdr[(size_t)e * 4 + 1], mlt = m.ehdr[(size_t)e * 4 + 3];
        const double sd1 = m.sd[(size_t)e * 2], sd2 = m.sd[(size_t)e * 2 + 1];
        const double fe = m.fEdge[e];
        double s1 = 0.0, s2 = 0.0;
        if constexpr (!TT) { s1 = a.lamS1[c1]; s2 = a.lamS1[c2]; }
        double2 tu = make_double2(0.0, 0.0);
        // both half-waves of the wave on regular edges: wave-uniform branch
        const bool plain = WF > 0 && __builtin_amdgcn_ballot_w64(!m.efull[e]) == 0;
        if (act) {
            const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
            const double2 l1 = ld2(a.lamH1, c1, K, l), l2 = ld2(a.lamH1, c2, K, l);
            double2 Fbar;
            if constexpr (TT) {
                Fbar = make_double2(sd1 * l1.x + sd2 * l2.x, sd1 * l1.y + sd2 * l2.y);
            } else {
                const double2 tH1 = make_double2(a.dt * (l1.x + s1), a.dt * (l1.y + s1));
                const double2 tH2 = make_double2(a.dt * (l2.x + s2), a.dt * (l2.y + s2));
                Fbar = make_double2(sd1 * tH1.x + sd2 * tH2.x, sd1 * tH1.y + sd2 * tH2.y);
            }
            if (!ax) Fbar.x = 0.0;
            if (!ay) Fbar.y = 0.0;
            double2 cor = make_double2(0.0, 0.0);
            if (plain) {
                constexpr int WU = WF > 0 ? WF : 1;
                int32_t src[WU];
                double w[WU];
                double2 ls[WU];
#pragma unroll
                for (int j = 0; j < WU; ++j) { src[j] = m.teoe[(size_t)e * WU + j]; w[j] = m.tw[(size_t)e * WU + j]; }
#pragma unroll
                for (int j = 0; j < WU; ++j) ls[j] = ld2(a.lamU1, src[j], K, l);
#pragma unroll
                for (int j = 0; j < WU; ++j) {
                    const double wf = w[j] * fe;
                    cor.x += TT ? wf * ls[j].x : wf * (a.dt * ls[j].x);
                    cor.y += TT ? wf * ls[j].y : wf * (a.dt * ls[j].y);
                }
            } else {
                for (int j = 0; j < m.W; ++j) {
                    const int s = m.teoe[(size_t)e * m.W + j];
                    if (s < 0) continue;
                    const int ms = m.ehdr[(size_t)s * 4 + 3];
                    const double w = m.tw[(size_t)e * m.W + j] * fe;
                    const double2 ls = ld2(a.lamU1, s, K, l);
                    if (k0 < ms) cor.x += TT ? w * ls.x : w * (a.dt * ls.x);
                    if (k0 + 1 < ms) cor.y += TT ? w * ls.y : w * (a.dt * ls.y);
                }
            }
            const double2 lu = ld2(a.lamU1, e, K, l);
            if constexpr (TT) {
                const double2 h1 = ld2(a.h, c1, K, l), h2 = ld2(a.h, c2, K, l);
                const double2 hI = make_double2(0.5 * (h1.x + h2.x), 0.5 * (h1.y + h2.y));           // Operators.jl:217
                const double2 pb = make_double2(hI.x * Fbar.x + cor.x, hI.y * Fbar.y + cor.y);
                if (a.accOutU) {                          // fused element-wise steps of the RK4 reverse sweep
                    const double2 x = ld2(a.xU, e, K, l), ai = a.accInU ? ld2(a.accInU, e, K, l) : x;
                    st2(a.accOutU, e, K, l, make_double2(ai.x + pb.x, ai.y + pb.y));
                    if (a.kNextU) st2(a.kNextU, e, K, l, make_double2(a.cbNext * x.x + a.caNext * pb.x, a.cbNext * x.y + a.caNext * pb.y));
                } else {
                    st2(a.lamU0, e, K, l, pb);
                }
            } else {
                const double2 hE = ld2(a.hEuse, e, K, l);
                st2(a.lamU0, e, K, l, make_double2((lu.x + hE.x * Fbar.x) + cor.x, (lu.y + hE.y * Fbar.y) + cor.y));
            }
            const double2 uu = ld2(a.u, e, K, l);
            st2(a.Enew, e, K, l, make_double2(uu.x * Fbar.x, uu.y * Fbar.y));
            if (ax) tu.x = TT ? lu.x : a.dt * lu.x;
            if (ay) tu.y = TT ? lu.y : a.dt * lu.y;
        }
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) {                       // oracle_ksum order
            const double ox = __shfl_xor(tu.x, sft, 32), oy = __shfl_xor(tu.y, sft, 32);
            tu = make_double2(tu.x + ox, tu.y + oy);
        }
        if (l == 0) a.csum[e] = tu.x + tu.y;
    }
}

// k_adj_edge2<TT, 10> with a workgroup per chunk of 64 consecutive edges (plan order: a compact piece of the mesh; chunks are
// dealt to the XCDs like the forward kernels' patches): the chunk's records -- transposed lists, weights, headers, metric
// factors -- are staged in LDS in one round trip, so that an edge's 18 row loads all go out together instead of behind
// three dependent record loads.  Same sums in the same order.
constexpr int ADJ_CH = 64;

template <bool TT>
__global__ __launch_bounds__(BLOCK, TT ? 3 : 4) void k_adj_edge3(const AdjMesh m, const AdjArgs a)
{
    constexpr int NG = BLOCK / 32, WU = 10;
    __shared__ int sSrc[ADJ_CH * WU];
    __shared__ double sW[ADJ_CH * WU];
    __shared__ int4 sHd[ADJ_CH];
    __shared__ double2 sSd[ADJ_CH];
    __shared__ double sFe[ADJ_CH];
    __shared__ int sFull[ADJ_CH];
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int K = m.K, k0 = 2 * l;
    const bool act = k0 < K;
    const int nCh = (m.eCount + ADJ_CH - 1) / ADJ_CH, ch = patch_of_block(nCh);
    if (ch >= nCh) return;
    const int e0 = m.eBegin + ch * ADJ_CH, ne = min(ADJ_CH, m.eBegin + m.eCount - e0);
    for (int i = threadIdx.x; i < ne * WU; i += BLOCK) { sSrc[i] = m.teoe[(size_t)e0 * WU + i]; sW[i] = m.tw[(size_t)e0 * WU + i]; }
    for (int i = threadIdx.x; i < ne; i += BLOCK) {
        sHd[i] = reinterpret_cast<const int4 *>(m.ehdr)[e0 + i];
        sSd[i] = reinterpret_cast<const double2 *>(m.sd)[e0 + i];
        sFe[i] = m.fEdge[e0 + i];
        sFull[i] = m.efull[e0 + i];
    }
    __syncthreads();
    for (int le = grp; le < ne; le += NG) {
        const int e = e0 + le;
        const int4 hd = sHd[le];
        const int c1 = hd.x, c2 = hd.y, mlt = hd.w;
        const double sd1 = sSd[le].x, sd2 = sSd[le].y;
        const double fe = sFe[le];
        double s1 = 0.0, s2 = 0.0;
        if constexpr (!TT) { s1 = a.lamS1[c1]; s2 = a.lamS1[c2]; }
        double2 tu = make_double2(0.0, 0.0);
        const bool plain = __builtin_amdgcn_ballot_w64(!sFull[le]) == 0;      // both half-waves of the wave on regular edges
        if (act) {
            const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
            double2 l1 = ld2(a.lamH1, c1, K, l), l2 = ld2(a.lamH1, c2, K, l);
            double2 cor = make_double2(0.0, 0.0);
            double2 ls[WU];
            if (plain) {
#pragma unroll
                for (int j = 0; j < WU; ++j) ls[j] = ld2(a.lamU1, sSrc[le * WU + j], K, l);
            }
            double2 lu = ld2(a.lamU1, e, K, l);
            const bool storeE = !(TT && a.fuseE);
            double2 uu = make_double2(0.0, 0.0);
            if (storeE) uu = ld2(a.u, e, K, l);
            double2 h1, h2, x, ai, hE;
            if constexpr (TT) {
                h1 = ld2(a.h, c1, K, l); h2 = ld2(a.h, c2, K, l);
                if (a.accOutU) { x = ld2(a.xU, e, K, l); ai = a.accInU ? ld2(a.accInU, e, K, l) : x; }
            } else {
                hE = ld2(a.hEuse, e, K, l);
            }
            double2 Fbar;
            if constexpr (TT) {
                const double sc = a.lamScale;                        // k-bar = lamScale * stored rows (1.0 except in stage 4)
                l1 = make_double2(sc * l1.x, sc * l1.y); l2 = make_double2(sc * l2.x, sc * l2.y); lu = make_double2(sc * lu.x, sc * lu.y);
                if (plain) {
#pragma unroll
                    for (int j = 0; j < WU; ++j) ls[j] = make_double2(sc * ls[j].x, sc * ls[j].y);
                }
                Fbar = make_double2(sd1 * l1.x + sd2 * l2.x, sd1 * l1.y + sd2 * l2.y);
            } else {
                const double2 tH1 = make_double2(a.dt * (l1.x + s1), a.dt * (l1.y + s1));
                const double2 tH2 = make_double2(a.dt * (l2.x + s2), a.dt * (l2.y + s2));
                Fbar = make_double2(sd1 * tH1.x + sd2 * tH2.x, sd1 * tH1.y + sd2 * tH2.y);
            }
            if (!ax) Fbar.x = 0.0;
            if (!ay) Fbar.y = 0.0;
            if (plain) {
#pragma unroll
                for (int j = 0; j < WU; ++j) {
                    const double wf = sW[le * WU + j] * fe;
                    cor.x += TT ? wf * ls[j].x : wf * (a.dt * ls[j].x);
                    cor.y += TT ? wf * ls[j].y : wf * (a.dt * ls[j].y);
                }
            } else {
                for (int j = 0; j < WU; ++j) {
                    const int sx = sSrc[le * WU + j];
                    if (sx < 0) continue;
                    const int ms = m.ehdr[(size_t)sx * 4 + 3];
                    const double w = sW[le * WU + j] * fe;
                    double2 lv = ld2(a.lamU1, sx, K, l);
                    if constexpr (TT) lv = make_double2(a.lamScale * lv.x, a.lamScale * lv.y);
                    if (k0 < ms) cor.x += TT ? w * lv.x : w * (a.dt * lv.x);
                    if (k0 + 1 < ms) cor.y += TT ? w * lv.y : w * (a.dt * lv.y);
                }
            }
            if constexpr (TT) {
                const double2 hI = make_double2(0.5 * (h1.x + h2.x), 0.5 * (h1.y + h2.y));           // Operators.jl:217
                const double2 pb = make_double2(hI.x * Fbar.x + cor.x, hI.y * Fbar.y + cor.y);
                if (a.accOutU) {                          // fused element-wise steps of the RK4 reverse sweep
                    st2(a.accOutU, e, K, l, make_double2(ai.x + pb.x, ai.y + pb.y));
                    if (a.kNextU) st2(a.kNextU, e, K, l, make_double2(a.cbNext * x.x + a.caNext * pb.x, a.cbNext * x.y + a.caNext * pb.y));
                } else {
                    st2(a.lamU0, e, K, l, pb);
                }
            } else {
                st2(a.lamU0, e, K, l, make_double2((lu.x + hE.x * Fbar.x) + cor.x, (lu.y + hE.y * Fbar.y) + cor.y));
            }
            if (storeE) st2(a.Enew, e, K, l, make_double2(uu.x * Fbar.x, uu.y * Fbar.y));
            if (ax) tu.x = TT ? lu.x : a.dt * lu.x;
            if (ay) tu.y = TT ? lu.y : a.dt * lu.y;
        }
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) {                       // oracle_ksum order
            const double ox = __shfl_xor(tu.x, sft, 32), oy = __shfl_xor(tu.y, sft, 32);
            tu = make_double2(tu.x + ox, tu.y + oy);
        }
        if (l == 0) a.csum[e] = tu.x + tu.y;
    }
}

// The cell half of a transposed tendency evaluation (tt) with u*Fbar recomputed instead of read: a workgroup per chunk of 64
// cells, per (cell, slot) the edge's header and metric factors staged in LDS (two dependent round trips per chunk), then per
// cell one batch of 15 row loads: the k-bar rows of the cell and of the cells across its edges, the stage's u rows, x and the
// running sum.  Fbar = sd1 * kbar[c1] + sd2 * kbar[c2] with the edge's own (c1, c2) order and mask: the value the edge
// kernel used, so the sums equal k_adj_cell2's bit for bit.  The edge kernel then neither loads u nor stores u*Fbar.
constexpr int ADJ_CCH = 64;

template <int ME_>
__global__ __launch_bounds__(BLOCK, 4) void k_adj_cell3(const AdjMesh m, const AdjArgs a)
{
    constexpr int NG = BLOCK / 32;
    __shared__ int4 sI[ADJ_CCH * ME_];        // edge (-1 none), the cell across, maxLevelEdgeTop of the edge, 1 if this cell is the edge's c1
    __shared__ double4 sD[ADJ_CCH * ME_];     // sd1, sd2 of the edge, -sign * g / dcEdge, csum of the edge
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int K = m.K, k0 = 2 * l;
    const bool act = k0 < K;
    const int nCh = (m.cCount + ADJ_CCH - 1) / ADJ_CCH, ch = patch_of_block(nCh);
    if (ch >= nCh) return;
    const int c0 = m.cBegin + ch * ADJ_CCH, nc = min(ADJ_CCH, m.cBegin + m.cCount - c0);
    for (int i = threadIdx.x; i < nc * ME_; i += BLOCK) {
        const int c = c0 + i / ME_;
        const int e = m.eoc[(size_t)c0 * ME_ + i];
        const int es = e < 0 ? 0 : e;
        const int4 hd = reinterpret_cast<const int4 *>(m.ehdr)[es];
        const double2 sd = reinterpret_cast<const double2 *>(m.sd)[es];
        sI[i] = make_int4(e, hd.x == c ? hd.y : hd.x, hd.w, hd.x == c ? 1 : 0);
        sD[i] = make_double4(sd.x, sd.y, (-(double)m.csgn[(size_t)c0 * ME_ + i]) * m.gInvDc[es], a.csum[es]);
    }
    __syncthreads();
    if (!act) return;
    const double sc = a.lamScale;
    for (int lc = grp; lc < nc; lc += NG) {
        const int c = c0 + lc;
        double2 own = ld2(a.lamH1, c, K, l);
        double2 oth[ME_], uu[ME_];
#pragma unroll
        for (int i = 0; i < ME_; ++i) {
            const int4 r = sI[lc * ME_ + i];
            oth[i] = ld2(a.lamH1, r.x < 0 ? c : r.y, K, l);
            uu[i] = ld2(a.u, r.x < 0 ? 0 : r.x, K, l);
        }
        const double2 x = ld2(a.xH, c, K, l), ai = a.accInH ? ld2(a.accInH, c, K, l) : x;
        own = make_double2(sc * own.x, sc * own.y);
        double ls = 0.0;
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < ME_; ++i) {
            const int4 r = sI[lc * ME_ + i];
            const double4 d = sD[lc * ME_ + i];
            if (r.x < 0) continue;
            ls += d.z * d.w;
            const double2 o = make_double2(sc * oth[i].x, sc * oth[i].y);
            const double2 l1 = r.w ? own : o, l2 = r.w ? o : own;
            double2 Fbar = make_double2(d.x * l1.x + d.y * l2.x, d.x * l1.y + d.y * l2.y);
            if (!(k0 < r.z)) Fbar.x = 0.0;
            if (!(k0 + 1 < r.z)) Fbar.y = 0.0;
            acc.x += uu[i].x * Fbar.x;
            acc.y += uu[i].y * Fbar.y;
        }
        const double2 pb = make_double2(0.5 * acc.x + ls, 0.5 * acc.y + ls);
        if (a.accOutH) {
            st2(a.accOutH, c, K, l, make_double2(ai.x + pb.x, ai.y + pb.y));
            if (a.kNextH) st2(a.kNextH, c, K, l, make_double2(a.cbNext * x.x + a.caNext * pb.x, a.cbNext * x.y + a.caNext * pb.y));
        } else {
            st2(a.lamH0, c, K, l, pb);
        }
    }
}

template <bool TT>
__global__ __launch_bounds__(BLOCK) void k_adj_cell2(const AdjMesh m, const AdjArgs a)
{
    constexpr int NG = BLOCK / 32;
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int K = m.K, ME = m.ME;
    const bool act = 2 * l < K;
    const double *Eread = (TT || !a.stale) ? a.Enew : a.lamE1;
    for (int c = blockIdx.x * NG + grp; c < m.nC; c += gridDim.x * NG) {
        const int32_t *re = m.eoc + (size_t)c * ME;
        double ls = 0.0;
        double2 acc = make_double2(0.0, 0.0);
        for (int i = 0; i < ME; ++i) {
            const int e = re[i];
            if (e < 0) continue;
            ls += (-(double)m.csgn[(size_t)c * ME + i]) * m.gInvDc[e] * a.csum[e];
            if (act) {
                const double2 v = ld2(Eread, e, K, l);
                acc.x += v.x;
                acc.y += v.y;
            }
        }
        if constexpr (TT) {
            if (act) {
                const double2 pb = make_double2(0.5 * acc.x + ls, 0.5 * acc.y + ls);
                if (a.accOutH) {
                    const double2 x = ld2(a.xH, c, K, l), ai = a.accInH ? ld2(a.accInH, c, K, l) : x;
                    st2(a.accOutH, c, K, l, make_double2(ai.x + pb.x, ai.y + pb.y));
                    if (a.kNextH) st2(a.kNextH, c, K, l, make_double2(a.cbNext * x.x + a.caNext * pb.x, a.cbNext * x.y + a.caNext * pb.y));
                } else {
                    st2(a.lamH0, c, K, l, pb);
                }
            }
        } else {
            const double s1 = a.lamS1[c];
            if (l == 0) a.lamS0[c] = ls;
            if (act) {
                const double2 lh = ld2(a.lamH1, c, K, l);
                st2(a.lamH0, c, K, l, make_double2((lh.x + s1) + 0.5 * acc.x, (lh.y + s1) + 0.5 * acc.y));
            }
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_scale_copy(double *dst, const double *src, double f, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[i] = f * src[i];
}

// dst = a*x + b*y (y == nullptr: dst = a*x; a == 1 and x == dst: dst += ... is written as dst = x + y by k_add)
__global__ __launch_bounds__(BLOCK) void k_axpby(double *dst, double a, const double *x, double b, const double *y, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        dst[i] = a * x[i] + b * y[i];
}

__global__ __launch_bounds__(BLOCK) void k_add(double *dst, const double *x, const double *y, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[i] = x[i] + y[i];
}

// dst[c][k] = f * src[c] for every level k
__global__ __launch_bounds__(BLOCK) void k_bcast_rows(double *dst, const double *src, double f, int64_t n, int K)
{
    const int64_t total = n * K;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) dst[i] = f * src[i / K];
}

template <int LPC>
static hipError_t launch_adj_edge_lpc(const AdjMesh &m, const AdjArgs &a, hipStream_t s)
{
    const int ng = BLOCK / LPC;
    int grid = (m.nE + ng - 1) / ng;
    if (grid > 65536) grid = 65536;
    if (a.tt) hipLaunchKernelGGL((k_adj_edge<LPC, true>), dim3(grid), dim3(BLOCK), 0, s, m, a);
    else hipLaunchKernelGGL((k_adj_edge<LPC, false>), dim3(grid), dim3(BLOCK), 0, s, m, a);
    return hipGetLastError();
}

template <int LPC>
static hipError_t launch_adj_cell_lpc(const AdjMesh &m, const AdjArgs &a, hipStream_t s)
{
    const int ng = BLOCK / LPC;
    int grid = (m.nC + ng - 1) / ng;
    if (grid > 65536) grid = 65536;
    if (a.tt) hipLaunchKernelGGL((k_adj_cell<LPC, true>), dim3(grid), dim3(BLOCK), 0, s, m, a);
    else hipLaunchKernelGGL((k_adj_cell<LPC, false>), dim3(grid), dim3(BLOCK), 0, s, m, a);
    return hipGetLastError();
}

static int grid2(int n) { return std::min(std::max((n + 7) / 8, 1), 65536); }

bool adj_fused_available(const AdjMesh &m, int lpc) { return lpc == 64 && m.K <= 64 && !(m.K & 1) && m.W == 10 && m.ME == 6; }
static inline bool adj_whole(const AdjMesh &m) { return m.eBegin == 0 && m.eCount == m.nE && m.cBegin == 0 && m.cCount == m.nC; }

hipError_t launch_adj_edge(const AdjMesh &m, const AdjArgs &a, int lpc, hipStream_t s)
{
    if (a.tt && (a.fuseE || a.lamScale != 1.0) && !adj_fused_available(m, lpc)) return hipErrorNotSupported;
    if (!adj_whole(m) && !(adj_fused_available(m, lpc) && a.tt && a.fuseE)) return hipErrorNotSupported;   // entity ranges: chunk kernels only
    if (lpc == 64 && m.K <= 64 && !(m.K & 1)) {   // even 34 <= K <= 64: 16-byte lanes
        if (m.W == 10) {
            if (m.eCount <= 0) return hipSuccess;
            const unsigned g = 8u * (unsigned)(((m.eCount + ADJ_CH - 1) / ADJ_CH + 7) / 8);
            if (a.tt) hipLaunchKernelGGL((k_adj_edge3<true>), dim3(g), dim3(BLOCK), 0, s, m, a);
            else hipLaunchKernelGGL((k_adj_edge3<false>), dim3(g), dim3(BLOCK), 0, s, m, a);
        } else {
            if (a.tt) hipLaunchKernelGGL((k_adj_edge2<true, 0>), dim3(grid2(m.nE)), dim3(BLOCK), 0, s, m, a);
            else hipLaunchKernelGGL((k_adj_edge2<false, 0>), dim3(grid2(m.nE)), dim3(BLOCK), 0, s, m, a);
        }
        return hipGetLastError();
    }
#define CALL(L) launch_adj_edge_lpc<L>(m, a, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

hipError_t launch_adj_cell(const AdjMesh &m, const AdjArgs &a, int lpc, hipStream_t s)
{
    if (!adj_whole(m) && !(adj_fused_available(m, lpc) && a.tt && a.fuseE)) return hipErrorNotSupported;
    if (a.tt && a.fuseE) {
        if (!adj_fused_available(m, lpc)) return hipErrorNotSupported;
        if (m.cCount <= 0) return hipSuccess;
        const unsigned g = 8u * (unsigned)(((m.cCount + ADJ_CCH - 1) / ADJ_CCH + 7) / 8);
        hipLaunchKernelGGL((k_adj_cell3<6>), dim3(g), dim3(BLOCK), 0, s, m, a);
        return hipGetLastError();
    }
    if (lpc == 64 && m.K <= 64 && !(m.K & 1)) {
        if (a.tt) hipLaunchKernelGGL((k_adj_cell2<true>), dim3(grid2(m.nC)), dim3(BLOCK), 0, s, m, a);
        else hipLaunchKernelGGL((k_adj_cell2<false>), dim3(grid2(m.nC)), dim3(BLOCK), 0, s, m, a);
        return hipGetLastError();
    }
#define CALL(L) launch_adj_cell_lpc<L>(m, a, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

static unsigned ew_blocks(int64_t n)
{
    int64_t blocks = (n + BLOCK - 1) / BLOCK;
    return (unsigned)std::min<int64_t>(std::max<int64_t>(blocks, 1), 65536);
}

hipError_t launch_axpby(double *dst, double a, const double *x, double b, const double *y, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_axpby, dim3(ew_blocks(n)), dim3(BLOCK), 0, s, dst, a, x, b, y, n);
    return hipGetLastError();
}

hipError_t launch_add(double *dst, const double *x, const double *y, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_add, dim3(ew_blocks(n)), dim3(BLOCK), 0, s, dst, x, y, n);
    return hipGetLastError();
}

hipError_t launch_bcast_rows(double *dst, const double *src, double f, int64_t n, int K, hipStream_t s)
{
    hipLaunchKernelGGL(k_bcast_rows, dim3(ew_blocks(n * K)), dim3(BLOCK), 0, s, dst, src, f, n, K);
    return hipGetLastError();
}

hipError_t launch_scale_copy(double *dst, const double *src, double f, int64_t n, hipStream_t s)
{
    int64_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_scale_copy, dim3((unsigned)blocks), dim3(BLOCK), 0, s, dst, src, f, n);
    return hipGetLastError();
}

}  // namespace moka
