/* moka_oracle.h -- CPU restatement of the MOKA.jl (jlk9/MPAS-Ocean.jl) shallow-water hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C restatement, loop nest by loop
 * nest, of the reference's KernelAbstractions kernels K1..K15 and of the
 * Forward-Euler / RK4 step that drives them.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (libmoka_hip.so) never links or calls it.
 *
 * Pinning: the reference is Julia and cannot run in this pipeline (no julia binary, no network).
 * The operators are pinned by the six known-answer norms of reference test/ocn/test_Operators.jl
 * :52-53,72-73,90-91 (tests/test_oracle_operators.py); the step loop is pinned by the analytic
 * inertia-gravity-wave solution of reference src/inertialGravityWave.jl.  The Forward-Euler
 * "reference_compat" sequencing itself has no stored golden output in the reference: for that
 * sequencing parity is UNPINNED beyond the source text.
 *
 * Conventions (identical to what the Julia arrays hold in memory):
 *   - connectivity: 1-based int32, slot index fastest ((maxEdges,nCells) column-major ==
 *     C array [nCells][maxEdges]); 0 marks "no neighbour" in edgesOnEdge;
 *   - fields: double, level index fastest ((nVertLevels,n) column-major == C [n][nVertLevels]);
 *   - arithmetic: every expression is evaluated in the reference's left-to-right order and the
 *     file is compiled with -ffp-contract=off (Julia does not contract a*b+c on the CPU backend).
 */
#ifndef MOKA_ORACLE_H
#define MOKA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t nCells, nEdges, nVertices;
    int32_t maxEdges, maxEdges2, vertexDegree, nVertLevels;
    int32_t edgeSignOnVertexLD; /* leading dimension of edgeSignOnVertex (= maxEdges, HorzMesh.jl:234) */
    /* PrimaryCells (HorzMesh.jl:102-132) */
    const int32_t *nEdgesOnCell, *edgesOnCell, *edgeSignOnCell;
    const double  *areaCell;
    /* Edges (HorzMesh.jl:64-95) */
    const int32_t *cellsOnEdge, *nEdgesOnEdge, *edgesOnEdge;
    const double  *weightsOnEdge, *dvEdge, *dcEdge, *fEdge;
    /* DualCells (HorzMesh.jl:135-162) */
    const int32_t *edgesOnVertex, *edgeSignOnVertex;
    const double  *areaTriangle;
    /* VerticalMesh (VertMesh.jl:3-26) */
    const int32_t *maxLevelEdgeTop;      /* all ones in the reference (VertMesh.jl:32) */
    const double  *restingThicknessSum;  /* (nCells) */
} oracle_mesh;

/* Flags for oracle_step_fe -- the order-of-evaluation quirks of SURVEY.md section 0.6 */
#define ORACLE_FE_STALE_HEDGE   1  /* thicknessFlux uses the previous step's layerThicknessEdge */
#define ORACLE_FE_ACCUM_VORT    2  /* CurlOnVertex accumulates into a never-zeroed array       */
#define ORACLE_FE_LEVEL1_ONLY   4  /* the "[1,j]" kernels touch level 1 only (matters for K>1)  */
#define ORACLE_FE_REFERENCE_COMPAT 7

void   oracle_set_threads(int n);
int    oracle_get_threads(void);

/* ---- operators (src/ocn/Operators.jl) ---- */
void oracle_divergence_on_cell(const oracle_mesh *m, double *div, const double *vecEdge, double *temp);
void oracle_gradient_on_edge(const oracle_mesh *m, double *grad, const double *scalarCell);
void oracle_curl_on_vertex(const oracle_mesh *m, double *curl, const double *vecEdge);
/* reverse mode of the three (test/enzyme/test_Enzyme_Operators.jl): input shadows accumulate, overwritten-output shadows are
 * zeroed, the accumulated-into curl shadow stays; forward mode = the operators themselves applied to tangents (linear) */
void oracle_gradient_on_edge_vjp(const oracle_mesh *m, double *dScalarCell, double *dGradEdge);
void oracle_divergence_on_cell_vjp(const oracle_mesh *m, double *dVecEdge, double *dTempEdge, double *dDivCell);
void oracle_curl_on_vertex_vjp(const oracle_mesh *m, double *dVecEdge, const double *dCurlVertex);
void oracle_interpolate_cell2edge(const oracle_mesh *m, double *edgeValue, const double *cellValue, int nlev);
void oracle_zero_out(double *a, int64_t n, int K, int nlev);

/* ---- diagnostics / tendencies (src/ocn/DiagnosticVars.jl, src/ocn/Tendencies) ---- */
void oracle_thickness_flux(const oracle_mesh *m, double *F, const double *u, const double *hEdge, int nlev);
void oracle_diagnostic_compute(const oracle_mesh *m, double *hEdge, double *F, double *div, double *vort,
                               const double *u, const double *h, int nlev);
void oracle_normal_velocity_tendency(const oracle_mesh *m, double *tendU, const double *ssh,
                                     const double *u, int nlev);
void oracle_layer_thickness_tendency(const oracle_mesh *m, double *tendH, const double *F, int nlev);

/* ssh[c] = ksum_k h[k,c] - restingThicknessSum[c]   (K=1: Update_ssh!, time_integration.jl:205-211) */
double oracle_ksum(const double *col, int K);
void   oracle_update_ssh(const oracle_mesh *m, double *ssh, const double *h, int nlev);

/* clean tendency: diagnostics consistent with the state they are used with */
void oracle_tendencies_clean(const oracle_mesh *m, double *tendU, double *tendH,
                             const double *u, const double *h, double *ssh_out,
                             double *hEdge_scratch, double *F_scratch);

/* ---- time stepping (src/forward/time_integration.jl) ----
 * State arrays are the two reference time levels: index 0 = previous, 1 = current/new. */
typedef struct {
    double *ssh[2], *u[2], *h[2];           /* PrognosticVars.jl:6-57  */
    double *hEdge, *F, *div, *vort;         /* DiagnosticVars.jl:6-73  */
    double *tendU, *tendH;                  /* TendencyVars.jl:7-49    */
} oracle_state;

void oracle_step_fe(const oracle_mesh *m, oracle_state *s, double dt, int flags);
/* RK4 per the (dead) specification time_integration.jl:61-148; work = 2*K*(nE+nC)+nC doubles */
void oracle_step_rk4(const oracle_mesh *m, oracle_state *s, double dt, double *work);
/* the 13-stream form of the same step (twin of libmoka_hip's opt-in form; not the reference's round-off); work: 2*K*(nE+nC) */
void oracle_step_rk4_s13(const oracle_mesh *m, oracle_state *s, double dt, double *work);
/* fp32-state / fp64-arithmetic storage emulation (config 5; parity unpinned: not a reference feature) */
void oracle_round_f32(double *a, int64_t n);
void oracle_tendencies_mixed(const oracle_mesh *m, double *tendU, double *tendH, const double *u, const double *h,
                             double *ssh_out, double *hEdge, double *F);
void oracle_step_rk4_mixed(const oracle_mesh *m, oracle_state *s, double dt, double *work);
void oracle_step_fe_mixed(const oracle_mesh *m, oracle_state *s, double dt, int flags);
double oracle_sum_sq(const double *a, int64_t n);   /* sumArray, run_loop.jl:47-51 */

/* nonlinear (vector-invariant TRiSK) tendencies and RK4 step: an extension, NOT in the reference -- parity unpinned */
void oracle_tendencies_nonlinear(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                                 const double *kiteAreasOnVertex, const double *fVertex,
                                 double *tendU, double *tendH, const double *u, const double *h, double *ssh_out,
                                 double *hEdge, double *F, double *qv, double *qe, double *ke);
/* ... with the Del2 momentum mixing of horizontal_momentum_mixing.jl:53-80 (a sketch the reference never calls) */
void oracle_tendencies_nonlinear_del2(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                                      const double *kiteAreasOnVertex, const double *fVertex,
                                      double *tendU, double *tendH, const double *u, const double *h, double *ssh_out,
                                      double *hEdge, double *F, double *qv, double *qe, double *ke,
                                      double viscDel2, double *zv, double *divc);
void oracle_step_rk4_nonlinear_del2(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                                    const double *kiteAreasOnVertex, const double *fVertex, oracle_state *s, double dt,
                                    double *work, double *scratch, double viscDel2);
/* 13-stream form of the nonlinear step (twin of moka_set_tuning key 7 on a nonlinear state); work: 2*K*(nEdges+nCells) */
void oracle_step_rk4_nonlinear_s13(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                                   const double *kiteAreasOnVertex, const double *fVertex, oracle_state *s, double dt,
                                   double *work, double *scratch, double viscDel2);
void oracle_step_rk4_nonlinear(const oracle_mesh *m, const int32_t *verticesOnEdge, const int32_t *cellsOnVertex,
                               const double *kiteAreasOnVertex, const double *fVertex, oracle_state *s, double dt,
                               double *work, double *scratch);

/* reverse mode of one Forward-Euler step (gather form, fixed order); see moka_oracle.c */
#define ORACLE_MAX_LEVELS 512
void oracle_step_fe_adjoint(const oracle_mesh *m, const int32_t *teoe, const double *tw, int tWidth, double dt, int flags,
                            const double *u, const double *hEuse,
                            const double *lamU1, const double *lamH1, const double *lamS1, const double *lamE1,
                            double *lamU0, double *lamH0, double *lamS0, double *lamE0, double *Enew, double *csum);
/* (outU, outH) = T'(u,h)^T (kU, kH): transpose of oracle_tendencies_clean, the building block of the RK4 reverse sweep */
void oracle_tendency_transpose(const oracle_mesh *m, const int32_t *teoe, const double *tw, int tWidth,
                               const double *u, const double *h, const double *kU, const double *kH,
                               double *outU, double *outH, double *Enew, double *csum);

#ifdef __cplusplus
}
#endif
#endif
