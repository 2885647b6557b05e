/* moka_oracle.c -- TEST INFRASTRUCTURE ONLY (see moka_oracle.h).
 *
 * One C function per reference kernel, same loop nest, same operand order.  Citations are
 * relative to the reference tree (jlk9/MPAS-Ocean.jl @ 2025-02-16).  Nothing here is derived
 * from the HIP implementation; the HIP implementation is checked against this file.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared   (oracle/Makefile)
 */
#include "moka_oracle.h"
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1; /* KA CPU() under JULIA_NUM_THREADS=1 is single threaded */

void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int  oracle_get_threads(void) { return g_threads; }

#define PFOR _Pragma("omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)")

/* index helpers: Julia A[i,j] (1-based, column major, leading dim ld) -> C offset */
#define IX(i, j, ld) ((int64_t)((j) - 1) * (ld) + ((i) - 1))

/* ------------------------------------------------------------------------------------------
 * K1 + K2  DivergenceOnCell_P1 / _P2                         src/ocn/Operators.jl:12-74
 *   P1: temp[k,e] = VecEdge[k,e] * dvEdge[e]                 (:18)
 *   P2: Div[k,c]  = 0; Div -= temp[k,eoc[i,c]] * sign[i,c]; Div /= areaCell[c]   (:34-42)
 * ndrange (nEdges,K) / (nCells,K): all levels.
 * ------------------------------------------------------------------------------------------ */
void oracle_divergence_on_cell(const oracle_mesh *m, double *div, const double *vecEdge, double *temp)
{
    const int K = m->nVertLevels;
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e)
        for (int k = 1; k <= K; ++k)
            temp[IX(k, e, K)] = vecEdge[IX(k, e, K)] * m->dvEdge[e - 1];
    PFOR
    for (int64_t c = 1; c <= m->nCells; ++c)
        for (int k = 1; k <= K; ++k) {
            double d = 0.0;
            for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i) {
                int32_t e = m->edgesOnCell[IX(i, c, m->maxEdges)];
                d -= temp[IX(k, e, K)] * (double)m->edgeSignOnCell[IX(i, c, m->maxEdges)];
            }
            div[IX(k, c, K)] = d / m->areaCell[c - 1];
        }
}

/* K3  GradientOnEdge                                         src/ocn/Operators.jl:84-120
 *   Grad[k,e] = (S[k,c2] - S[k,c1]) / dcEdge[e]              (:97) */
void oracle_gradient_on_edge(const oracle_mesh *m, double *grad, const double *s)
{
    const int K = m->nVertLevels;
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e) {
        int32_t c1 = m->cellsOnEdge[IX(1, e, 2)], c2 = m->cellsOnEdge[IX(2, e, 2)];
        for (int k = 1; k <= K; ++k)
            grad[IX(k, e, K)] = (s[IX(k, c2, K)] - s[IX(k, c1, K)]) / m->dcEdge[e - 1];
    }
}

/* K4  CurlOnVertex                                           src/ocn/Operators.jl:122-177
 *   invA = 1/areaTriangle[v]; Curl[k,v] += dcEdge[e]*invA*Vec[k,e]*sign[j,v]     (:137-146)
 * NOTE accumulates into the caller's array: the zeroing at :135 is commented out. */
void oracle_curl_on_vertex(const oracle_mesh *m, double *curl, const double *vecEdge)
{
    const int K = m->nVertLevels;
    PFOR
    for (int64_t v = 1; v <= m->nVertices; ++v) {
        double invA = 1.0 / m->areaTriangle[v - 1];
        for (int k = 1; k <= K; ++k) {
            double c = curl[IX(k, v, K)];
            for (int j = 1; j <= m->vertexDegree; ++j) {
                int32_t e = m->edgesOnVertex[IX(j, v, m->vertexDegree)];
                c += m->dcEdge[e - 1] * invA * vecEdge[IX(k, e, K)] *
                     (double)m->edgeSignOnVertex[IX(j, v, m->edgeSignOnVertexLD)];
            }
            curl[IX(k, v, K)] = c;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Reverse mode of the three stand-alone operators: what the reference obtains from Enzyme in
 * test/enzyme/test_Enzyme_Operators.jl:42-63 (gradient) and :137-162 (divergence), written out by hand.
 * Conventions are Enzyme's for in-place kernels with Duplicated arguments: the shadow of an input is
 * ACCUMULATED into (d_in += J^T d_out); the shadow of an output the kernel overwrites is ZERO afterwards;
 * the shadow of the curl output, which the kernel accumulates into (:142), stays as it is.
 * Every sum has a fixed order (gather form where the transposed stencil is a per-entity list), so that the
 * HIP twins (moka_*_vjp) can be compared bit for bit; the functions themselves are pinned the way the
 * reference pins Enzyme: against central differences (eps = 1e-8 relative, atol = 1e-6; :102,127,196,221)
 * and by the adjoint identity <J x, y> = <x, J^T y> (tests/test_oracle_operators.py).
 * Forward mode needs no twin: the operators are linear, the tangent of the output is the operator applied
 * to the tangent of the input (oracle_gradient_on_edge / _divergence_on_cell / _curl_on_vertex).
 * ------------------------------------------------------------------------------------------ */
/* Grad[k,e] = (S[k,c2] - S[k,c1]) / dcEdge[e]  =>  dS[k,c] += sum_i sign[i,c] * (dGrad[k,e_i] / dcEdge[e_i]),
 * i in edgesOnCell order, sign = edgeSignOnCell (-1 where c is cellsOnEdge[1,e], HorzMesh.jl:302-306); then dGrad = 0 */
void oracle_gradient_on_edge_vjp(const oracle_mesh *m, double *dScalar, double *dGrad)
{
    const int K = m->nVertLevels;
    PFOR
    for (int64_t c = 1; c <= m->nCells; ++c)
        for (int k = 1; k <= K; ++k) {
            double a = dScalar[IX(k, c, K)];
            for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i) {
                int32_t e = m->edgesOnCell[IX(i, c, m->maxEdges)];
                a += (double)m->edgeSignOnCell[IX(i, c, m->maxEdges)] * (dGrad[IX(k, e, K)] / m->dcEdge[e - 1]);
            }
            dScalar[IX(k, c, K)] = a;
        }
    memset(dGrad, 0, sizeof(double) * (size_t)K * (size_t)m->nEdges);
}

/* edgeSignOnCell of edge e in cell c (0 if c does not list e) */
static double sign_of_edge_in_cell(const oracle_mesh *m, int32_t e, int32_t c)
{
    for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i)
        if (m->edgesOnCell[IX(i, c, m->maxEdges)] == e) return (double)m->edgeSignOnCell[IX(i, c, m->maxEdges)];
    return 0.0;
}

/* P2^T: dTemp[k,e] -= sign(e in c1) * (dDiv[k,c1] / areaCell[c1]); the same for c2 (cellsOnEdge order); dDiv = 0
 * P1^T: dVec[k,e] += dTemp[k,e] * dvEdge[e]; dTemp = 0                                  (Operators.jl:18,34-42) */
void oracle_divergence_on_cell_vjp(const oracle_mesh *m, double *dVec, double *dTemp, double *dDiv)
{
    const int K = m->nVertLevels;
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e) {
        const int32_t c1 = m->cellsOnEdge[IX(1, e, 2)], c2 = m->cellsOnEdge[IX(2, e, 2)];
        const double s1 = sign_of_edge_in_cell(m, (int32_t)e, c1), s2 = sign_of_edge_in_cell(m, (int32_t)e, c2);
        for (int k = 1; k <= K; ++k) {
            double t = dTemp[IX(k, e, K)];
            t -= s1 * (dDiv[IX(k, c1, K)] / m->areaCell[c1 - 1]);
            t -= s2 * (dDiv[IX(k, c2, K)] / m->areaCell[c2 - 1]);
            dVec[IX(k, e, K)] += t * m->dvEdge[e - 1];
            dTemp[IX(k, e, K)] = 0.0;
        }
    }
    memset(dDiv, 0, sizeof(double) * (size_t)K * (size_t)m->nCells);
}

/* Curl[k,v] += dcEdge[e]*invA*Vec[k,e]*sign[j,v]  =>  dVec[k,e] += (dcEdge[e]*invA*sign[j,v]) * dCurl[k,v], in ascending
 * (v, j) order (a serial scatter: the order the gather lists of the HIP twin are sorted in); dCurl stays (Operators.jl:142) */
void oracle_curl_on_vertex_vjp(const oracle_mesh *m, double *dVec, const double *dCurl)
{
    const int K = m->nVertLevels;
    for (int64_t v = 1; v <= m->nVertices; ++v) {
        const double invA = 1.0 / m->areaTriangle[v - 1];
        for (int j = 1; j <= m->vertexDegree; ++j) {
            const int32_t e = m->edgesOnVertex[IX(j, v, m->vertexDegree)];
            const double w = m->dcEdge[e - 1] * invA * (double)m->edgeSignOnVertex[IX(j, v, m->edgeSignOnVertexLD)];
            for (int k = 1; k <= K; ++k) dVec[IX(k, e, K)] += w * dCurl[IX(k, v, K)];
        }
    }
}

/* K5  interpolateCell2Edge                                   src/ocn/Operators.jl:179-222
 *   edge[k,e] = 0.5 * (cell[k,c1] + cell[k,c2]), reference: k = 1 only (:207-208).
 * nlev = 1 reproduces the reference; nlev = K is the N3 extension. */
void oracle_interpolate_cell2edge(const oracle_mesh *m, double *edgeValue, const double *cellValue, int nlev)
{
    const int K = m->nVertLevels;
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e) {
        int32_t c1 = m->cellsOnEdge[IX(1, e, 2)], c2 = m->cellsOnEdge[IX(2, e, 2)];
        for (int k = 1; k <= nlev; ++k)
            edgeValue[IX(k, e, K)] = 0.5 * (cellValue[IX(k, c1, K)] + cellValue[IX(k, c2, K)]);
    }
}

/* K6  ZeroOutVector!                                         src/ocn/Operators.jl:225-231
 *   A[1,j] = 0  (level 1 only in the reference) */
void oracle_zero_out(double *a, int64_t n, int K, int nlev)
{
    PFOR
    for (int64_t j = 1; j <= n; ++j)
        for (int k = 1; k <= nlev; ++k) a[IX(k, j, K)] = 0.0;
}

/* K7  compute_thicknessFlux!                                 src/ocn/DiagnosticVars.jl:158-173
 *   F[1,e] = u[1,e] * hEdge[1,e] */
void oracle_thickness_flux(const oracle_mesh *m, double *F, const double *u, const double *hEdge, int nlev)
{
    const int K = m->nVertLevels;
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e)
        for (int k = 1; k <= nlev; ++k) F[IX(k, e, K)] = u[IX(k, e, K)] * hEdge[IX(k, e, K)];
}

/* diagnostic_compute!                                        src/ocn/DiagnosticVars.jl:108-117
 * fixed order: thicknessFlux (with the OLD hEdge) -> velocityDivCell (hEdge used as scratch,
 * :185-190) -> relativeVorticity (accumulating) -> layerThicknessEdge. */
void oracle_diagnostic_compute(const oracle_mesh *m, double *hEdge, double *F, double *div, double *vort,
                               const double *u, const double *h, int nlev)
{
    oracle_thickness_flux(m, F, u, hEdge, nlev);
    oracle_divergence_on_cell(m, div, u, hEdge);
    oracle_curl_on_vertex(m, vort, u);
    oracle_interpolate_cell2edge(m, hEdge, h, nlev);
}

/* K9  SSHGradOnEdge!            src/ocn/Tendencies/normalVelocity/pressure_gradient.jl:45-65
 *   InvDc = 1/dcEdge[e]; for k in 1:maxLevelEdgeTop[e]: T[k,e] -= 9.80616*InvDc*(ssh[c2]-ssh[c1])
 * K10 coriolis_force_tendency_kernel!   .../horizontal_advection_and_coriolis.jl:50-75
 *   for i in 1:nEdgesOnEdge[e]: eoe = edgesOnEdge[i,e]; eoe == 0 && continue;
 *       for k in 1:maxLevelEdgeTop[e]: T[k,e] += weightsOnEdge[i,e]*u[k,eoe]*fEdge[eoe]
 * computeNormalVelocityTendency! (normalVelocity.jl:21-53): zero (K6), K9, K10. */
void oracle_normal_velocity_tendency(const oracle_mesh *m, double *tendU, const double *ssh,
                                     const double *u, int nlev)
{
    const int K = m->nVertLevels;
    oracle_zero_out(tendU, m->nEdges, K, nlev);
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e) {
        int32_t c1 = m->cellsOnEdge[IX(1, e, 2)], c2 = m->cellsOnEdge[IX(2, e, 2)];
        double invDc = 1. / m->dcEdge[e - 1];
        for (int k = 1; k <= m->maxLevelEdgeTop[e - 1]; ++k)
            tendU[IX(k, e, K)] -= 9.80616 * invDc * (ssh[c2 - 1] - ssh[c1 - 1]);
    }
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e) {
        for (int i = 1; i <= m->nEdgesOnEdge[e - 1]; ++i) {
            int32_t eoe = m->edgesOnEdge[IX(i, e, m->maxEdges2)];
            if (eoe == 0) continue;
            for (int k = 1; k <= m->maxLevelEdgeTop[e - 1]; ++k)
                tendU[IX(k, e, K)] += m->weightsOnEdge[IX(i, e, m->maxEdges2)] * u[IX(k, eoe, K)] *
                                      m->fEdge[eoe - 1];
        }
    }
}

/* K8  thicknessFluxDivOnCell!   src/ocn/Tendencies/layerThickness/horizontal_advection.jl:42-68
 *   invArea = 1/areaCell[c]; for i: e = eoc[i,c]; for k in 1:maxLevelEdgeTop[e]:
 *       T[k,c] += F[k,e]*dvEdge[e]*edgeSignOnCell[i,c]*invArea
 * computeLayerThicknessTendency! (layerThickness.jl:14-28): zero (K6), K8. */
void oracle_layer_thickness_tendency(const oracle_mesh *m, double *tendH, const double *F, int nlev)
{
    const int K = m->nVertLevels;
    oracle_zero_out(tendH, m->nCells, K, nlev);
    PFOR
    for (int64_t c = 1; c <= m->nCells; ++c) {
        double invArea = 1. / m->areaCell[c - 1];
        for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i) {
            int32_t e = m->edgesOnCell[IX(i, c, m->maxEdges)];
            for (int k = 1; k <= m->maxLevelEdgeTop[e - 1]; ++k)
                tendH[IX(k, c, K)] += F[IX(k, e, K)] * m->dvEdge[e - 1] *
                                      (double)m->edgeSignOnCell[IX(i, c, m->maxEdges)] * invArea;
        }
    }
}

/* Column sum used for ssh when K > 1 (a build decision, not a reference fact).
 * K = 1 returns col[0] exactly, i.e. Update_ssh! (time_integration.jl:209).  For K > 1 the
 * order is fixed as: 64 strided partial sums acc[l] = col[l] + col[l+64] + ..., then an XOR
 * butterfly acc[l] += acc[l^s], s = 32,16,...,1 -- the order a 64-lane wavefront reduction
 * produces, so the GPU can match it bit for bit.  (fp add is commutative, so every lane of the
 * butterfly holds the same value; lane 0 is returned.) */
double oracle_ksum(const double *col, int K)
{
    double acc[64];
    for (int l = 0; l < 64; ++l) {
        double a = 0.0;
        int first = 1;
        for (int k = l; k < K; k += 64) {
            a = first ? col[k] : a + col[k];
            first = 0;
        }
        acc[l] = a;
    }
    for (int s = 32; s >= 1; s >>= 1) {
        double t[64];
        for (int l = 0; l < 64; ++l) t[l] = acc[l] + acc[l ^ s];
        memcpy(acc, t, sizeof acc);
    }
    return acc[0];
}

/* K14 Update_ssh!                                            src/forward/time_integration.jl:205-211
 *   ssh[j] = layerThickness[1,j] - restingThicknessSum[j]        (nlev = 1)
 *   N3:      ssh[j] = ksum_k layerThickness[k,j] - restingThicknessSum[j]   (nlev = K) */
void oracle_update_ssh(const oracle_mesh *m, double *ssh, const double *h, int nlev)
{
    const int K = m->nVertLevels;
    PFOR
    for (int64_t c = 1; c <= m->nCells; ++c)
        ssh[c - 1] = oracle_ksum(&h[IX(1, c, K)], nlev) - m->restingThicknessSum[c - 1];
}

/* Tendency evaluation with diagnostics consistent with (u,h): SURVEY.md Appendix C. */
void oracle_tendencies_clean(const oracle_mesh *m, double *tendU, double *tendH,
                             const double *u, const double *h, double *ssh_out,
                             double *hEdge, double *F)
{
    const int K = m->nVertLevels;
    oracle_update_ssh(m, ssh_out, h, K);
    oracle_interpolate_cell2edge(m, hEdge, h, K);
    oracle_thickness_flux(m, F, u, hEdge, K);
    oracle_normal_velocity_tendency(m, tendU, ssh_out, u, K);
    oracle_layer_thickness_tendency(m, tendH, F, K);
}

/* K11/K12 advance_2d_array / advance_3d_array               src/forward/time_integration.jl:10-59
 *   prev[j] = next[j];  prev[1,j] = next[1,j] (level 1 only) */
static void advance_levels(double *prev, const double *next, int64_t n, int K, int nlev)
{
    PFOR
    for (int64_t j = 1; j <= n; ++j)
        for (int k = 1; k <= nlev; ++k) prev[IX(k, j, K)] = next[IX(k, j, K)];
}

/* K13 UpdateStateVariable!                                   src/forward/time_integration.jl:196-202
 *   var[1,j] = var[1,j] + dt[1]*tend[1,j] */
static void update_state(double *var, const double *tend, double dt, int64_t n, int K, int nlev)
{
    PFOR
    for (int64_t j = 1; j <= n; ++j)
        for (int k = 1; k <= nlev; ++k) var[IX(k, j, K)] = var[IX(k, j, K)] + dt * tend[IX(k, j, K)];
}

/* ocn_timestep(..., ForwardEuler)                            src/forward/time_integration.jl:150-193
 * flags = ORACLE_FE_REFERENCE_COMPAT reproduces the live reference step.
 * Clearing STALE_HEDGE refreshes layerThicknessEdge before the flux (and uses tendU as the
 * divergence scratch); clearing ACCUM_VORT zeroes relativeVorticity first; clearing
 * LEVEL1_ONLY applies the "[1,j]" kernels to all K levels (N3). */
void oracle_step_fe(const oracle_mesh *m, oracle_state *s, double dt, int flags)
{
    const int K = m->nVertLevels;
    const int nlev = (flags & ORACLE_FE_LEVEL1_ONLY) ? 1 : K;
    /* advanceTimeLevels! (:163) */
    advance_levels(s->ssh[0], s->ssh[1], m->nCells, 1, 1);
    advance_levels(s->u[0], s->u[1], m->nEdges, K, nlev);
    advance_levels(s->h[0], s->h[1], m->nCells, K, nlev);
    /* diagnostic_compute! (:169) */
    if (flags & ORACLE_FE_STALE_HEDGE) {
        if (!(flags & ORACLE_FE_ACCUM_VORT)) memset(s->vort, 0, sizeof(double) * (size_t)K * m->nVertices);
        oracle_diagnostic_compute(m, s->hEdge, s->F, s->div, s->vort, s->u[1], s->h[1], nlev);
    } else {
        oracle_interpolate_cell2edge(m, s->hEdge, s->h[1], nlev);
        oracle_thickness_flux(m, s->F, s->u[1], s->hEdge, nlev);
        oracle_divergence_on_cell(m, s->div, s->u[1], s->tendU);
        if (!(flags & ORACLE_FE_ACCUM_VORT)) memset(s->vort, 0, sizeof(double) * (size_t)K * m->nVertices);
        oracle_curl_on_vertex(m, s->vort, s->u[1]);
    }
    /* tendencies (:172-177) */
    oracle_normal_velocity_tendency(m, s->tendU, s->ssh[1], s->u[1], nlev);
    oracle_layer_thickness_tendency(m, s->tendH, s->F, nlev);
    /* updates (:183-189) */
    update_state(s->u[1], s->tendU, dt, m->nEdges, K, nlev);
    update_state(s->h[1], s->tendH, dt, m->nCells, K, nlev);
    oracle_update_ssh(m, s->ssh[1], s->h[1], nlev);
}

/* ocn_timestep(..., RungeKutta4)  -- dead code in the reference; algorithm specification only
 *                                                            src/forward/time_integration.jl:61-148
 *   a = [dt/2, dt/2, dt] (:77);  b = [dt/6, dt/3, dt/3, dt/6] (:78)
 *   Curr = level end-1, Provis = level end, New = copy of Provis (:93-110)
 *   for s in 1:4: tend = T(Provis); s<4: Provis = Curr + a[s]*tend, ssh from layerThickness (:124-127)
 *                 New = New + b[s]*tend (:134-135)
 *   state[end] = New (:140-141); diagnostics of the new state (:147)
 * All K levels (N3).  Diagnostics at the end are the clean ones (vorticity zeroed first). */
void oracle_step_rk4(const oracle_mesh *m, oracle_state *s, double dt, double *work)
{
    const int K = m->nVertLevels;
    const int64_t nu = (int64_t)K * m->nEdges, nh = (int64_t)K * m->nCells;
    double *newU = work, *newH = work + nu, *hEdge = s->hEdge, *F = s->F;
    const double a[3] = {dt / 2., dt / 2., dt};
    const double b[4] = {dt / 6., dt / 3., dt / 3., dt / 6.};
    advance_levels(s->ssh[0], s->ssh[1], m->nCells, 1, 1);
    advance_levels(s->u[0], s->u[1], m->nEdges, K, K);
    advance_levels(s->h[0], s->h[1], m->nCells, K, K);
    memcpy(newU, s->u[1], sizeof(double) * (size_t)nu);
    memcpy(newH, s->h[1], sizeof(double) * (size_t)nh);
    for (int st = 0; st < 4; ++st) {
        oracle_tendencies_clean(m, s->tendU, s->tendH, s->u[1], s->h[1], s->ssh[1], hEdge, F);
        if (st < 3) {
            const double as = a[st];
            double *pu = s->u[1], *ph = s->h[1];
            const double *cu = s->u[0], *ch = s->h[0], *tu = s->tendU, *th = s->tendH;
            PFOR
            for (int64_t i = 0; i < nu; ++i) pu[i] = cu[i] + as * tu[i];
            PFOR
            for (int64_t i = 0; i < nh; ++i) ph[i] = ch[i] + as * th[i];
        }
        {
            const double bs = b[st];
            const double *tu = s->tendU, *th = s->tendH;
            PFOR
            for (int64_t i = 0; i < nu; ++i) newU[i] = newU[i] + bs * tu[i];
            PFOR
            for (int64_t i = 0; i < nh; ++i) newH[i] = newH[i] + bs * th[i];
        }
    }
    memcpy(s->u[1], newU, sizeof(double) * (size_t)nu);
    memcpy(s->h[1], newH, sizeof(double) * (size_t)nh);
    oracle_update_ssh(m, s->ssh[1], s->h[1], K);
    oracle_interpolate_cell2edge(m, s->hEdge, s->h[1], K);
    oracle_thickness_flux(m, s->F, s->u[1], s->hEdge, K);
    oracle_divergence_on_cell(m, s->div, s->u[1], newU); /* newU is free again: scratch */
    memset(s->vort, 0, sizeof(double) * (size_t)K * m->nVertices);
    oracle_curl_on_vertex(m, s->vort, s->u[1]);
}

/* The RK4 step with 13 instead of 16 state streams (the twin of libmoka_hip's opt-in form, moka_set_tuning key 7; NOT the
 * reference's arithmetic): the stages store only the provisional states P2 = C + dt/2 k1, P3 = C + dt/2 k2, P4 = C + dt k3
 * (time_integration.jl:124-125), and New = C + dt/6 k1 + dt/3 k2 + dt/3 k3 + dt/6 k4 (:78,134-135) is formed at the end as
 *   (C + ((P2 - C) + ((P3 - C) + (P3 - C)) + (P4 - C)) * (1/3)) + dt/6 * k4
 * in exactly this order -- the same Runge-Kutta step up to round-off.  work: 2 * K * (nEdges + nCells) doubles. */
static double rk13_combine(double c, double p2, double p3, double p4, double b4, double t)
{
    const double d2 = p2 - c, d3 = p3 - c, d4 = p4 - c;
    const double acc = (d2 + (d3 + d3)) + d4;
    return (c + acc * (1.0 / 3.0)) + b4 * t;
}

void oracle_step_rk4_s13(const oracle_mesh *m, oracle_state *s, double dt, double *work)
{
    const int K = m->nVertLevels;
    const int64_t nu = (int64_t)K * m->nEdges, nh = (int64_t)K * m->nCells;
    double *p2u = work, *p2h = work + nu, *p3u = work + nu + nh, *p3h = work + 2 * nu + nh;
    double *hEdge = s->hEdge, *F = s->F;
    const double a[3] = {dt / 2., dt / 2., dt};
    advance_levels(s->ssh[0], s->ssh[1], m->nCells, 1, 1);
    advance_levels(s->u[0], s->u[1], m->nEdges, K, K);
    advance_levels(s->h[0], s->h[1], m->nCells, K, K);
    for (int st = 0; st < 4; ++st) {
        oracle_tendencies_clean(m, s->tendU, s->tendH, s->u[1], s->h[1], s->ssh[1], hEdge, F);
        double *pu = s->u[1], *ph = s->h[1];
        const double *cu = s->u[0], *ch = s->h[0], *tu = s->tendU, *th = s->tendH;
        if (st < 3) {
            const double as = a[st];
            PFOR
            for (int64_t i = 0; i < nu; ++i) pu[i] = cu[i] + as * tu[i];
            PFOR
            for (int64_t i = 0; i < nh; ++i) ph[i] = ch[i] + as * th[i];
            if (st == 0) { memcpy(p2u, pu, sizeof(double) * (size_t)nu); memcpy(p2h, ph, sizeof(double) * (size_t)nh); }
            if (st == 1) { memcpy(p3u, pu, sizeof(double) * (size_t)nu); memcpy(p3h, ph, sizeof(double) * (size_t)nh); }
        } else {
            const double b4 = dt / 6.;
            PFOR
            for (int64_t i = 0; i < nu; ++i) pu[i] = rk13_combine(cu[i], p2u[i], p3u[i], pu[i], b4, tu[i]);
            PFOR
            for (int64_t i = 0; i < nh; ++i) ph[i] = rk13_combine(ch[i], p2h[i], p3h[i], ph[i], b4, th[i]);
        }
    }
    oracle_update_ssh(m, s->ssh[1], s->h[1], K);
    oracle_interpolate_cell2edge(m, s->hEdge, s->h[1], K);
    oracle_thickness_flux(m, s->F, s->u[1], s->hEdge, K);
    oracle_divergence_on_cell(m, s->div, s->u[1], p2u);   /* p2u is free again: scratch */
    memset(s->vort, 0, sizeof(double) * (size_t)K * m->nVertices);
    oracle_curl_on_vertex(m, s->vort, s->u[1]);
}

/* ---------------------------------------------------------------------------------------------
 * fp32 state / fp64 arithmetic ("mixed", BASELINE.json config 5).  Not a reference feature (the reference
 * hard-codes Float64: PrognosticVars.jl:91-93): a storage option of this build, so PARITY UNPINNED beyond
 * agreeing with the fp64 path to fp32 round-off.  Semantics: ssh, normalVelocity, layerThickness of every
 * time level and RK provisional state are STORED as fp32; every load widens to fp64; all arithmetic is the
 * fp64 arithmetic above, in the same order; a store rounds to nearest fp32.  The tendency ARRAYS (what moka_tendencies
 * leaves in Tend) are stored fp32 too -- oracle.py rounds them; inside an RK step the fp64 tendency is used unrounded.
 * The oracle keeps double arrays whose values are fp32-representable (rnd32 at each store).
 * --------------------------------------------------------------------------------------------- */
static inline double rnd32(double x) { return (double)(float)x; }

void oracle_round_f32(double *a, int64_t n)
{
    PFOR
    for (int64_t i = 0; i < n; ++i) a[i] = rnd32(a[i]);
}

void oracle_tendencies_mixed(const oracle_mesh *m, double *tendU, double *tendH,
                             const double *u, const double *h, double *ssh_out,
                             double *hEdge, double *F)
{
    const int K = m->nVertLevels;
    oracle_update_ssh(m, ssh_out, h, K);
    oracle_round_f32(ssh_out, m->nCells);          /* ssh is stored fp32 before the pressure gradient reads it */
    oracle_interpolate_cell2edge(m, hEdge, h, K);
    oracle_thickness_flux(m, F, u, hEdge, K);
    oracle_normal_velocity_tendency(m, tendU, ssh_out, u, K);
    oracle_layer_thickness_tendency(m, tendH, F, K);
}

/* oracle_step_rk4 with fp32 storage of Provis / New (same sequence, time_integration.jl:61-148).
 * Inputs must already be fp32-representable (oracle_round_f32).  No end-of-step diagnostics. */
void oracle_step_rk4_mixed(const oracle_mesh *m, oracle_state *s, double dt, double *work)
{
    const int K = m->nVertLevels;
    const int64_t nu = (int64_t)K * m->nEdges, nh = (int64_t)K * m->nCells;
    double *newU = work, *newH = work + nu, *hEdge = s->hEdge, *F = s->F;
    const double a[3] = {dt / 2., dt / 2., dt};
    const double b[4] = {dt / 6., dt / 3., dt / 3., dt / 6.};
    advance_levels(s->ssh[0], s->ssh[1], m->nCells, 1, 1);
    advance_levels(s->u[0], s->u[1], m->nEdges, K, K);
    advance_levels(s->h[0], s->h[1], m->nCells, K, K);
    memcpy(newU, s->u[1], sizeof(double) * (size_t)nu);
    memcpy(newH, s->h[1], sizeof(double) * (size_t)nh);
    for (int st = 0; st < 4; ++st) {
        oracle_tendencies_mixed(m, s->tendU, s->tendH, s->u[1], s->h[1], s->ssh[1], hEdge, F);
        if (st < 3) {
            const double as = a[st];
            double *pu = s->u[1], *ph = s->h[1];
            const double *cu = s->u[0], *ch = s->h[0], *tu = s->tendU, *th = s->tendH;
            PFOR
            for (int64_t i = 0; i < nu; ++i) pu[i] = rnd32(cu[i] + as * tu[i]);
            PFOR
            for (int64_t i = 0; i < nh; ++i) ph[i] = rnd32(ch[i] + as * th[i]);
        }
        {
            const double bs = b[st];
            const double *tu = s->tendU, *th = s->tendH;
            PFOR
            for (int64_t i = 0; i < nu; ++i) newU[i] = rnd32(newU[i] + bs * tu[i]);
            PFOR
            for (int64_t i = 0; i < nh; ++i) newH[i] = rnd32(newH[i] + bs * th[i]);
        }
    }
    memcpy(s->u[1], newU, sizeof(double) * (size_t)nu);
    memcpy(s->h[1], newH, sizeof(double) * (size_t)nh);
    oracle_update_ssh(m, s->ssh[1], s->h[1], K);
    oracle_round_f32(s->ssh[1], m->nCells);
}

/* oracle_step_fe with fp32 storage of every array of Prog, Diag and Tend (all levels; not a reference feature -- the
 * reference is Float64 throughout).  Inputs must already be fp32-representable.  Every array element is computed in fp64
 * from the stored (fp32) inputs exactly as oracle_step_fe computes it and rounded ONCE when it is stored: within the step
 * thicknessFlux feeds the thickness tendency unrounded, the tendencies feed the updates unrounded; ssh is the column sum of
 * the STORED (rounded) new layerThickness, like the provisional ssh of oracle_step_rk4_mixed. */
void oracle_step_fe_mixed(const oracle_mesh *m, oracle_state *s, double dt, int flags)
{
    const int K = m->nVertLevels;
    const int64_t nu = (int64_t)K * m->nEdges, nh = (int64_t)K * m->nCells, nv = (int64_t)K * m->nVertices;
    oracle_step_fe(m, s, dt, flags & ~ORACLE_FE_LEVEL1_ONLY);
    oracle_round_f32(s->u[1], nu);
    oracle_round_f32(s->h[1], nh);
    oracle_update_ssh(m, s->ssh[1], s->h[1], K);
    oracle_round_f32(s->ssh[1], m->nCells);
    oracle_round_f32(s->hEdge, nu);
    oracle_round_f32(s->F, nu);
    oracle_round_f32(s->div, nh);
    oracle_round_f32(s->vort, nv);
    oracle_round_f32(s->tendU, nu);
    oracle_round_f32(s->tendH, nh);
}

/* ---------------------------------------------------------------------------------------------
 * Reverse mode of one Forward-Euler step.  The reference obtains it from Enzyme
 * (ext/MPASEnzymeExt.jl; test/enzyme/test_Enzyme_end2end.jl differentiates sum(ssh^2) after ocn_run_loop with
 * respect to the initial layerThickness / normalVelocity and checks one entry against central differences); this is
 * the hand transposition of oracle_step_fe, written in GATHER form with a fixed summation order so that the HIP
 * kernels can match it bit for bit.  Pinned like the reference's own AD test: against finite differences of the
 * forward oracle (tests/test_oracle_adjoint.py) -- no independent values exist.
 *
 * State of the FE map: (u, h, ssh, hE) with hE = DiagnosticVars.layerThicknessEdge carried between steps when
 * ORACLE_FE_STALE_HEDGE is set.  Forward, all levels (LEVEL1_ONLY is supported for K = 1 only):
 *   hEuse = stale ? hE : interp(h);  F = u*hEuse;  tendH = sum_i F*dv*sign/area;  tendU = -g*(ssh2-ssh1)/dc + Coriolis(u)
 *   u' = u + dt*tendU;  h' = h + dt*tendH;  ssh' = ksum_k h' - rsum;  hE' = interp(h)
 * Inputs: adjoints after the step (lamU1, lamH1, lamS1, lamE1), the forward u and hEuse of the step.
 * teoe/tw (tWidth, nEdges): for edge e the j-th (source edge s, weightsOnEdge[i,s]) with edgesOnEdge[i,s] == e,
 * sorted by (s, i) -- the transpose of the Coriolis stencil; 0 = no entry.
 * --------------------------------------------------------------------------------------------- */
static double cell_sign_of_edge(const oracle_mesh *m, int32_t c, int64_t e)
{
    for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i)
        if (m->edgesOnCell[IX(i, c, m->maxEdges)] == e) return (double)m->edgeSignOnCell[IX(i, c, m->maxEdges)];
    return 0.0;
}

void oracle_step_fe_adjoint(const oracle_mesh *m, const int32_t *teoe, const double *tw, int tWidth, double dt, int flags,
                            const double *u, const double *hEuse,
                            const double *lamU1, const double *lamH1, const double *lamS1, const double *lamE1,
                            double *lamU0, double *lamH0, double *lamS0, double *lamE0, double *Enew, double *csum)
{
    const int K = m->nVertLevels;
    const int stale = (flags & ORACLE_FE_STALE_HEDGE) != 0;
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e) {
        const int32_t c1 = m->cellsOnEdge[IX(1, e, 2)], c2 = m->cellsOnEdge[IX(2, e, 2)];
        const int mlt = m->maxLevelEdgeTop[e - 1];
        const double sd1 = m->dvEdge[e - 1] * cell_sign_of_edge(m, c1, e) * (1. / m->areaCell[c1 - 1]);
        const double sd2 = m->dvEdge[e - 1] * cell_sign_of_edge(m, c2, e) * (1. / m->areaCell[c2 - 1]);
        double col[ORACLE_MAX_LEVELS];
        for (int k = 1; k <= K; ++k) {
            double Fbar = 0.0;
            if (k <= mlt) {
                const double tH1 = dt * (lamH1[IX(k, c1, K)] + lamS1[c1 - 1]);
                const double tH2 = dt * (lamH1[IX(k, c2, K)] + lamS1[c2 - 1]);
                Fbar = sd1 * tH1 + sd2 * tH2;
            }
            double cor = 0.0;
            for (int j = 1; j <= tWidth; ++j) {
                const int32_t s = teoe[IX(j, e, tWidth)];
                if (s == 0 || k > m->maxLevelEdgeTop[s - 1]) continue;
                cor += (tw[IX(j, e, tWidth)] * m->fEdge[e - 1]) * (dt * lamU1[IX(k, s, K)]);
            }
            lamU0[IX(k, e, K)] = (lamU1[IX(k, e, K)] + hEuse[IX(k, e, K)] * Fbar) + cor;
            Enew[IX(k, e, K)] = u[IX(k, e, K)] * Fbar;
            col[k - 1] = k <= mlt ? dt * lamU1[IX(k, e, K)] : 0.0;
        }
        csum[e - 1] = oracle_ksum(col, K);
    }
    const double *Eread = stale ? lamE1 : Enew;
    PFOR
    for (int64_t c = 1; c <= m->nCells; ++c) {
        double ls = 0.0;
        for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i) {
            const int32_t e = m->edgesOnCell[IX(i, c, m->maxEdges)];
            ls += (-(double)m->edgeSignOnCell[IX(i, c, m->maxEdges)]) * (9.80616 * (1. / m->dcEdge[e - 1])) * csum[e - 1];
        }
        lamS0[c - 1] = ls;
        for (int k = 1; k <= K; ++k) {
            double acc = 0.0;
            for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i)
                acc += Eread[IX(k, m->edgesOnCell[IX(i, c, m->maxEdges)], K)];
            lamH0[IX(k, c, K)] = (lamH1[IX(k, c, K)] + lamS1[c - 1]) + 0.5 * acc;
        }
    }
    PFOR
    for (int64_t i = 0; i < (int64_t)K * m->nEdges; ++i) lamE0[i] = stale ? Enew[i] : 0.0;
}

/* Transpose of the tendency evaluation T(u,h) = (tendU, tendH) of oracle_tendencies_clean at the point (u,h):
 *   (outU, outH) = T'(u,h)^T (kU, kH)
 * -- the building block of the RK4 reverse sweep (each stage applies T to a provisional state).  ssh is computed
 * inside T from h, so its adjoint flows into every level of outH.  Gather form, fixed order, like the FE transpose. */
void oracle_tendency_transpose(const oracle_mesh *m, const int32_t *teoe, const double *tw, int tWidth,
                               const double *u, const double *h, const double *kU, const double *kH,
                               double *outU, double *outH, double *Enew, double *csum)
{
    const int K = m->nVertLevels;
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e) {
        const int32_t c1 = m->cellsOnEdge[IX(1, e, 2)], c2 = m->cellsOnEdge[IX(2, e, 2)];
        const int mlt = m->maxLevelEdgeTop[e - 1];
        const double sd1 = m->dvEdge[e - 1] * cell_sign_of_edge(m, c1, e) * (1. / m->areaCell[c1 - 1]);
        const double sd2 = m->dvEdge[e - 1] * cell_sign_of_edge(m, c2, e) * (1. / m->areaCell[c2 - 1]);
        double col[ORACLE_MAX_LEVELS];
        for (int k = 1; k <= K; ++k) {
            const double hI = 0.5 * (h[IX(k, c1, K)] + h[IX(k, c2, K)]);
            double Fbar = 0.0;
            if (k <= mlt) Fbar = sd1 * kH[IX(k, c1, K)] + sd2 * kH[IX(k, c2, K)];
            double cor = 0.0;
            for (int j = 1; j <= tWidth; ++j) {
                const int32_t s = teoe[IX(j, e, tWidth)];
                if (s == 0 || k > m->maxLevelEdgeTop[s - 1]) continue;
                cor += (tw[IX(j, e, tWidth)] * m->fEdge[e - 1]) * kU[IX(k, s, K)];
            }
            outU[IX(k, e, K)] = hI * Fbar + cor;
            Enew[IX(k, e, K)] = u[IX(k, e, K)] * Fbar;
            col[k - 1] = k <= mlt ? kU[IX(k, e, K)] : 0.0;
        }
        csum[e - 1] = oracle_ksum(col, K);
    }
    PFOR
    for (int64_t c = 1; c <= m->nCells; ++c) {
        double ls = 0.0;
        for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i) {
            const int32_t e = m->edgesOnCell[IX(i, c, m->maxEdges)];
            ls += (-(double)m->edgeSignOnCell[IX(i, c, m->maxEdges)]) * (9.80616 * (1. / m->dcEdge[e - 1])) * csum[e - 1];
        }
        for (int k = 1; k <= K; ++k) {
            double acc = 0.0;
            for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i)
                acc += Enew[IX(k, m->edgesOnCell[IX(i, c, m->maxEdges)], K)];
            outH[IX(k, c, K)] = 0.5 * acc + ls;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Nonlinear shallow-water tendencies (SURVEY.md section 8(f) rank 4 / note N4): the vector-invariant TRiSK form of
 * MPAS-Ocean (Ringler et al. 2010) that north_star names -- potential-vorticity flux over edgesOnEdge, gradient of
 * kinetic energy + ssh over cellsOnEdge, vertex relativeVorticity.  The REFERENCE HAS NO SUCH TERMS (its Coriolis
 * term is the linear f*u_perp, horizontal_advection_and_coriolis.jl:61-73; relativeVorticity is computed and never
 * used), so PARITY IS UNPINNED: this is an optional extension, off by default, pinned only by its own properties
 * (steady solid-body rotation = Williamson test case 2, linear limit) in tests/test_oracle_nonlinear.py.
 *   hEdge = 1/2 (h[c1]+h[c2]);  F = u hEdge
 *   zeta[v] = sum_j dc*invA_v*u*sign;  hv[v] = (sum_j kite[j,v] h[c_j]) invA_v;  q_v = (fVertex + zeta)/hv
 *   q_e = 1/2 (q_v[v1] + q_v[v2]);  KE[c] = (sum_i (1/4 dc dv) u u) invArea_c
 *   tendH = as the linear form;  tendU = -g (ssh2-ssh1)/dc - (KE2-KE1)/dc + sum_i w[i,e] F[eoe_i] 1/2 (q_e[e] + q_e[eoe_i])
 * Extra mesh arrays (1-based, reference conventions): verticesOnEdge (2,nE), cellsOnVertex (VD,nV),
 * kiteAreasOnVertex (VD,nV), fVertex (nV).
 * --------------------------------------------------------------------------------------------- */
void oracle_tendencies_nonlinear(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                                 const double *kiteAreasOnVertex, const double *fVertex,
                                 double *tendU, double *tendH, const double *u, const double *h, double *ssh_out,
                                 double *hEdge, double *F, double *qv, double *qe, double *ke)
{
    oracle_tendencies_nonlinear_del2(m, verticesOnEdge, cellsOnVertex, kiteAreasOnVertex, fVertex, tendU, tendH, u, h,
                                     ssh_out, hEdge, F, qv, qe, ke, 0.0, NULL, NULL);
}

/* ... plus the Del2 momentum mixing of the reference's (never called, not runnable) sketch
 *   src/ocn/Tendencies/normalVelocity/horizontal_momentum_mixing.jl:53-80
 *   tendency[k,e] += ((div[k,c2] - div[k,c1]) * (1/dcEdge[e]) - (relVort[k,v2] - relVort[k,v1]) * (1/dvEdge[e])) * viscDel2
 * with div = velocityDivCell as DivergenceOnCell forms it (Operators.jl:18,34-42: -= (u*dvEdge)*sign, / areaCell) and
 * relVort = relativeVorticity as CurlOnVertex forms it from zero (Operators.jl:137-146).  viscDel2 = 0 (or zv NULL)
 * leaves the term out altogether; zv (K,nV) and divc (K,nC) receive the two diagnostics. */
void oracle_tendencies_nonlinear_del2(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                                      const double *kiteAreasOnVertex, const double *fVertex,
                                      double *tendU, double *tendH, const double *u, const double *h, double *ssh_out,
                                      double *hEdge, double *F, double *qv, double *qe, double *ke,
                                      double viscDel2, double *zv, double *divc)
{
    const int del2 = viscDel2 != 0.0 && zv && divc;
    const int K = m->nVertLevels, VD = m->vertexDegree;
    oracle_update_ssh(m, ssh_out, h, K);
    oracle_interpolate_cell2edge(m, hEdge, h, K);
    oracle_thickness_flux(m, F, u, hEdge, K);
    oracle_layer_thickness_tendency(m, tendH, F, K);
    PFOR
    for (int64_t v = 1; v <= m->nVertices; ++v) {
        const double invA = 1.0 / m->areaTriangle[v - 1];
        for (int k = 1; k <= K; ++k) {
            double zeta = 0.0, hv = 0.0;
            for (int j = 1; j <= VD; ++j) {
                const int32_t e = m->edgesOnVertex[IX(j, v, VD)];
                zeta += m->dcEdge[e - 1] * invA * u[IX(k, e, K)] * (double)m->edgeSignOnVertex[IX(j, v, m->edgeSignOnVertexLD)];
                hv += kiteAreasOnVertex[IX(j, v, VD)] * h[IX(k, cellsOnVertex[IX(j, v, VD)], K)];
            }
            hv = hv * invA;
            qv[IX(k, v, K)] = (fVertex[v - 1] + zeta) / hv;
            if (del2) zv[IX(k, v, K)] = zeta;
        }
    }
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e) {
        const int32_t v1 = verticesOnEdge[IX(1, e, 2)], v2 = verticesOnEdge[IX(2, e, 2)];
        for (int k = 1; k <= K; ++k) qe[IX(k, e, K)] = 0.5 * (qv[IX(k, v1, K)] + qv[IX(k, v2, K)]);
    }
    PFOR
    for (int64_t c = 1; c <= m->nCells; ++c) {
        const double invA = 1.0 / m->areaCell[c - 1];
        for (int k = 1; k <= K; ++k) {
            double acc = 0.0, d = 0.0;
            for (int i = 1; i <= m->nEdgesOnCell[c - 1]; ++i) {
                const int32_t e = m->edgesOnCell[IX(i, c, m->maxEdges)];
                acc += (0.25 * m->dcEdge[e - 1] * m->dvEdge[e - 1]) * u[IX(k, e, K)] * u[IX(k, e, K)];
                d -= (u[IX(k, e, K)] * m->dvEdge[e - 1]) * (double)m->edgeSignOnCell[IX(i, c, m->maxEdges)];
            }
            ke[IX(k, c, K)] = acc * invA;
            if (del2) divc[IX(k, c, K)] = d / m->areaCell[c - 1];
        }
    }
    PFOR
    for (int64_t e = 1; e <= m->nEdges; ++e) {
        const int32_t c1 = m->cellsOnEdge[IX(1, e, 2)], c2 = m->cellsOnEdge[IX(2, e, 2)];
        const double invDc = 1. / m->dcEdge[e - 1], invDv = 1. / m->dvEdge[e - 1];
        const int32_t v1 = verticesOnEdge[IX(1, e, 2)], v2 = verticesOnEdge[IX(2, e, 2)];
        for (int k = 1; k <= K; ++k) {
            double t = 0.0;
            if (k <= m->maxLevelEdgeTop[e - 1]) {
                t -= 9.80616 * invDc * (ssh_out[c2 - 1] - ssh_out[c1 - 1]);
                t -= invDc * (ke[IX(k, c2, K)] - ke[IX(k, c1, K)]);
                for (int i = 1; i <= m->nEdgesOnEdge[e - 1]; ++i) {
                    const int32_t eoe = m->edgesOnEdge[IX(i, e, m->maxEdges2)];
                    if (eoe == 0) continue;
                    t += m->weightsOnEdge[IX(i, e, m->maxEdges2)] * F[IX(k, eoe, K)] *
                         (0.5 * (qe[IX(k, e, K)] + qe[IX(k, eoe, K)]));
                }
                if (del2)
                    t += ((divc[IX(k, c2, K)] - divc[IX(k, c1, K)]) * invDc -
                          (zv[IX(k, v2, K)] - zv[IX(k, v1, K)]) * invDv) * viscDel2;
            }
            tendU[IX(k, e, K)] = t;
        }
    }
}

/* oracle_step_rk4 with the nonlinear tendencies (no end-of-step diagnostics); work = 2*K*(nE+nC) doubles,
 * scratch = 4*K*nE + K*nV + K*nC doubles */
void oracle_step_rk4_nonlinear(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                               const double *kiteAreasOnVertex, const double *fVertex, oracle_state *s, double dt,
                               double *work, double *scratch)
{
    oracle_step_rk4_nonlinear_del2(m, verticesOnEdge, cellsOnVertex, kiteAreasOnVertex, fVertex, s, dt, work, scratch, 0.0);
}

/* the same with Del2 mixing; scratch then holds K*nV + K*nC doubles more (4*K*nE + 2*K*nV + 2*K*nC) */
void oracle_step_rk4_nonlinear_del2(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                                    const double *kiteAreasOnVertex, const double *fVertex, oracle_state *s, double dt,
                                    double *work, double *scratch, double viscDel2)
{
    const int K = m->nVertLevels;
    const int64_t nu = (int64_t)K * m->nEdges, nh = (int64_t)K * m->nCells;
    double *newU = work, *newH = work + nu;
    double *qe = scratch, *qv = scratch + nu, *ke = qv + (int64_t)K * m->nVertices;
    double *zv = viscDel2 != 0.0 ? scratch + 4 * nu + (int64_t)K * m->nVertices + nh : NULL;
    double *divc = zv ? zv + (int64_t)K * m->nVertices : NULL;
    const double a[3] = {dt / 2., dt / 2., dt};
    const double b[4] = {dt / 6., dt / 3., dt / 3., dt / 6.};
    advance_levels(s->ssh[0], s->ssh[1], m->nCells, 1, 1);
    advance_levels(s->u[0], s->u[1], m->nEdges, K, K);
    advance_levels(s->h[0], s->h[1], m->nCells, K, K);
    memcpy(newU, s->u[1], sizeof(double) * (size_t)nu);
    memcpy(newH, s->h[1], sizeof(double) * (size_t)nh);
    for (int st = 0; st < 4; ++st) {
        oracle_tendencies_nonlinear_del2(m, verticesOnEdge, cellsOnVertex, kiteAreasOnVertex, fVertex, s->tendU, s->tendH,
                                         s->u[1], s->h[1], s->ssh[1], s->hEdge, s->F, qv, qe, ke, viscDel2, zv, divc);
        if (st < 3) {
            const double as = a[st];
            double *pu = s->u[1], *ph = s->h[1];
            const double *cu = s->u[0], *ch = s->h[0], *tu = s->tendU, *th = s->tendH;
            PFOR
            for (int64_t i = 0; i < nu; ++i) pu[i] = cu[i] + as * tu[i];
            PFOR
            for (int64_t i = 0; i < nh; ++i) ph[i] = ch[i] + as * th[i];
        }
        {
            const double bs = b[st];
            const double *tu = s->tendU, *th = s->tendH;
            PFOR
            for (int64_t i = 0; i < nu; ++i) newU[i] = newU[i] + bs * tu[i];
            PFOR
            for (int64_t i = 0; i < nh; ++i) newH[i] = newH[i] + bs * th[i];
        }
    }
    memcpy(s->u[1], newU, sizeof(double) * (size_t)nu);
    memcpy(s->h[1], newH, sizeof(double) * (size_t)nh);
    oracle_update_ssh(m, s->ssh[1], s->h[1], K);
}

/* The nonlinear RK4 step in the 13-stream form (twin of the library's opt-in form for nonlinear states: moka_set_tuning key 7):
 * the stages of oracle_step_rk4_nonlinear_del2 with the update of oracle_step_rk4_s13 (rk13_combine above, same order).
 * work: 2 * K * (nEdges + nCells) doubles. */
void oracle_step_rk4_nonlinear_s13(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                                   const double *kiteAreasOnVertex, const double *fVertex, oracle_state *s, double dt,
                                   double *work, double *scratch, double viscDel2)
{
    const int K = m->nVertLevels;
    const int64_t nu = (int64_t)K * m->nEdges, nh = (int64_t)K * m->nCells;
    double *p2u = work, *p2h = work + nu, *p3u = work + nu + nh, *p3h = work + 2 * nu + nh;
    double *qe = scratch, *qv = scratch + nu, *ke = qv + (int64_t)K * m->nVertices;
    double *zv = viscDel2 != 0.0 ? scratch + 4 * nu + (int64_t)K * m->nVertices + nh : NULL;
    double *divc = zv ? zv + (int64_t)K * m->nVertices : NULL;
    const double a[3] = {dt / 2., dt / 2., dt};
    advance_levels(s->ssh[0], s->ssh[1], m->nCells, 1, 1);
    advance_levels(s->u[0], s->u[1], m->nEdges, K, K);
    advance_levels(s->h[0], s->h[1], m->nCells, K, K);
    for (int st = 0; st < 4; ++st) {
        oracle_tendencies_nonlinear_del2(m, verticesOnEdge, cellsOnVertex, kiteAreasOnVertex, fVertex, s->tendU, s->tendH,
                                         s->u[1], s->h[1], s->ssh[1], s->hEdge, s->F, qv, qe, ke, viscDel2, zv, divc);
        double *pu = s->u[1], *ph = s->h[1];
        const double *cu = s->u[0], *ch = s->h[0], *tu = s->tendU, *th = s->tendH;
        if (st < 3) {
            const double as = a[st];
            PFOR
            for (int64_t i = 0; i < nu; ++i) pu[i] = cu[i] + as * tu[i];
            PFOR
            for (int64_t i = 0; i < nh; ++i) ph[i] = ch[i] + as * th[i];
            if (st == 0) { memcpy(p2u, pu, sizeof(double) * (size_t)nu); memcpy(p2h, ph, sizeof(double) * (size_t)nh); }
            if (st == 1) { memcpy(p3u, pu, sizeof(double) * (size_t)nu); memcpy(p3h, ph, sizeof(double) * (size_t)nh); }
        } else {
            const double b4 = dt / 6.;
            PFOR
            for (int64_t i = 0; i < nu; ++i) pu[i] = rk13_combine(cu[i], p2u[i], p3u[i], pu[i], b4, tu[i]);
            PFOR
            for (int64_t i = 0; i < nh; ++i) ph[i] = rk13_combine(ch[i], p2h[i], p3h[i], ph[i], b4, th[i]);
        }
    }
    oracle_update_ssh(m, s->ssh[1], s->h[1], K);
}

/* K15 sumArray (serial, one work-item)                       src/forward/run_loop.jl:47-51
 *   sum = sum + a[j]*a[j] */
double oracle_sum_sq(const double *a, int64_t n)
{
    double sum = 0.0;
    for (int64_t j = 0; j < n; ++j) sum = sum + a[j] * a[j];
    return sum;
}
