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
                    a.lamU0[off] = hI * Fbar + cor;
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
            if constexpr (TT) a.lamH0[(size_t)c * K + k] = 0.5 * acc + ls;
            else a.lamH0[(size_t)c * K + k] = (a.lamH1[(size_t)c * K + k] + s1) + 0.5 * acc;
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

hipError_t launch_adj_edge(const AdjMesh &m, const AdjArgs &a, int lpc, hipStream_t s)
{
#define CALL(L) launch_adj_edge_lpc<L>(m, a, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

hipError_t launch_adj_cell(const AdjMesh &m, const AdjArgs &a, int lpc, hipStream_t s)
{
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
