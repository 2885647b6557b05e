/* moka_hip.h -- C ABI of libmoka_hip.so: the MI355X-native (gfx950, HIP) implementation of the
 * MOKA.jl (jlk9/MPAS-Ocean.jl) shallow-water hot path: TRiSK tendency operators of src/ocn and
 * the Forward-Euler / RK4 step loop of src/forward.
 *
 * This is the drop-in boundary.  The reference has no FFI; its seam is Julia dispatch on the
 * `backend` keyword + `Adapt.adapt_structure` (src/Architectures.jl:12, MPASMesh.jl:26,
 * HorzMesh.jl:53,357,373,388, VertMesh.jl:119,124, PrognosticVars.jl:108, DiagnosticVars.jl:101).
 * A Julia shim (mpas-ocean.jl_amd/julia/MokaHIP.jl) overloads those methods for a `MokaHIP`
 * backend tag and `ccall`s the functions below; INTEGRATION.md shows the binding.  Each entry
 * point cites the reference method it replaces.
 *
 * Conventions
 *   - Plain C types only.  All host arrays are exactly what Julia holds in memory: Float64 /
 *     Int32, column-major, 1-based connectivity (0 = "no neighbour" in edgesOnEdge), the slot
 *     index fastest for connectivity ((maxEdges,nCells)...), the level index fastest for fields
 *     ((nVertLevels,n)).
 *   - Host pointers are borrowed for the duration of a call only.  The library owns all device
 *     memory, its reordered copies of the mesh and the permutations; uploads/downloads present
 *     the caller's original numbering.
 *   - Every function returns 0 (MOKA_OK) or a negative moka_status; the message is available from
 *     moka_last_error().  Nothing aborts, nothing prints.
 *   - One context = one device = one owning host thread (not thread-safe).
 *   - Operator entry points are synchronous on return (the reference ends every operator with
 *     KA.synchronize: Operators.jl:72,119,176,198).  moka_step_ / moka_run are asynchronous on the
 *     context's compute stream; download, sum_sq, timer_stop and sync are completion points.
 *   - There is no CPU fallback: without a usable HIP device moka_ctx_create fails.
 */
#ifndef MOKA_HIP_H
#define MOKA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct moka_ctx   moka_ctx;
typedef struct moka_plan  moka_plan;   /* host-side reordered mesh (no GPU needed) */
typedef struct moka_mesh  moka_mesh;   /* plan resident on a device                */
typedef struct moka_state moka_state;  /* Prog + Diag + Tend on a device           */

typedef enum {
    MOKA_OK = 0,
    MOKA_ERR_ARG = -1,       /* bad argument / inconsistent mesh (Julia: error("..."))  */
    MOKA_ERR_HIP = -2,       /* a HIP runtime call failed (message has hipGetErrorString) */
    MOKA_ERR_NO_DEVICE = -3, /* no usable gfx950 device: there is no CPU fallback         */
    MOKA_ERR_ALLOC = -4,
    MOKA_ERR_UNSUPPORTED = -5,
    MOKA_ERR_COMM = -6
} moka_status;

/* cell ordering applied at mesh upload (north_star: "re-laid out SoA and RCM-reordered") */
typedef enum {
    MOKA_ORDER_DEFAULT = 0,  /* RCB patches when coordinates are given, else RCM */
    MOKA_ORDER_NONE = 1,     /* keep the caller's numbering                      */
    MOKA_ORDER_RCM = 2,      /* reverse Cuthill-McKee on the cell graph          */
    MOKA_ORDER_RCB = 3       /* recursive coordinate bisection into compact patches */
} moka_ordering;

/* Mesh as the reference holds it: Edges (HorzMesh.jl:64-95), PrimaryCells (:102-132),
 * DualCells (:135-162), VerticalMesh (VertMesh.jl:3-26).  Pointers marked (opt) may be NULL. */
typedef struct moka_mesh_desc {
    int32_t nCells, nEdges, nVertices;
    int32_t maxEdges, maxEdges2, vertexDegree;
    int32_t nVertLevels;
    int32_t edgeSignOnVertexLD;      /* leading dim of edgeSignOnVertex (= maxEdges, HorzMesh.jl:234); 0 -> vertexDegree */
    /* PrimaryCells */
    const double  *xCell, *yCell, *zCell;            /* (opt) only used to order cells          */
    const int32_t *nEdgesOnCell;                     /* (nCells)                                */
    const int32_t *edgesOnCell;                      /* (maxEdges, nCells)                      */
    const int32_t *edgeSignOnCell;                   /* (maxEdges, nCells)  signIndexField! :292 */
    const double  *areaCell;                         /* (nCells)                                */
    /* Edges */
    const int32_t *cellsOnEdge;                      /* (2, nEdges)                             */
    const int32_t *verticesOnEdge;                   /* (opt) (2, nEdges)                       */
    const int32_t *nEdgesOnEdge;                     /* (nEdges)                                */
    const int32_t *edgesOnEdge;                      /* (maxEdges2, nEdges), 0 = none           */
    const double  *weightsOnEdge;                    /* (maxEdges2, nEdges)                     */
    const double  *dvEdge, *dcEdge, *fEdge;          /* (nEdges)                                */
    /* DualCells */
    const int32_t *edgesOnVertex;                    /* (vertexDegree, nVertices)               */
    const int32_t *cellsOnVertex;                    /* (opt) (vertexDegree, nVertices)         */
    const int32_t *edgeSignOnVertex;                 /* (edgeSignOnVertexLD, nVertices) :313    */
    const double  *areaTriangle;                     /* (nVertices)                             */
    /* VerticalMesh */
    const int32_t *maxLevelEdgeTop;                  /* (opt) (nEdges); NULL -> all ones (VertMesh.jl:32) */
    const double  *restingThicknessSum;              /* (nCells) (VertMesh.jl:73,100)           */
    /* layout controls */
    int32_t ordering;                                /* moka_ordering                            */
    int32_t patch_cells;                             /* cells per patch (0 = library default)    */
    /* (opt) (nCells) small non-negative class per cell; cells are ordered by class first, then by `ordering`
     * inside a class.  The multi-GPU layer passes 0 = owned & needed by another rank, 1 = owned interior,
     * 2 = halo, so that patch ranges [0,pb) / [pb,po) / [po,nPatches) are boundary / interior / halo. */
    const int32_t *cellClass;
    /* bytes per real of the prognostic state this mesh will carry: 0 or 8 = Float64 (the reference,
     * PrognosticVars.jl:91-93); 4 = fp32 STORAGE of every array of PrognosticVars (every time level and RK provisional
     * state), DiagnosticVars and TendencyVars, with fp64 arithmetic: an element is accumulated in fp64 and rounded once
     * when it is stored (BASELINE config 5; not a reference feature).
     * It fixes the byte offsets baked into the gather records, so a mesh serves states of one storage type. */
    int32_t stateBytes;
    /* (opt) only for the optional nonlinear terms (moka_set_nonlinear): MPAS mesh files carry both, the reference
     * reads neither.  With them verticesOnEdge and cellsOnVertex are required too. */
    const double *kiteAreasOnVertex;                 /* (vertexDegree, nVertices) */
    const double *fVertex;                           /* (nVertices)               */
} moka_mesh_desc;

typedef struct moka_mesh_info {
    int32_t nCells, nEdges, nVertices, nVertLevels;
    int32_t ordering, patch_cells, nPatches;
    int32_t maxEdgesUsed, maxEdges2Used;             /* record widths after compaction           */
    int32_t lanesPerColumn;                          /* wavefront lanes that span one k-column   */
    int64_t meshBytesDevice;                         /* reordered mesh resident in HBM           */
    int64_t cellBandwidth;                           /* max |new(c1)-new(c2)| over edges         */
    int32_t maxPatchRows;                            /* most u-rows (own + halo edges) any patch stages in LDS */
    int32_t ldsBytesPerBlock;                        /* dynamic LDS of the LDS-tiled stage kernel (0: not applicable) */
    int32_t maxPatchCells, maxPatchEdges;            /* largest own cell / edge range of any patch (sizes the record staging) */
} moka_mesh_info;

/* entity kinds for permutations */
enum { MOKA_CELL = 0, MOKA_EDGE = 1, MOKA_VERTEX = 2 };

/* state fields (PrognosticVars.jl:6-57, DiagnosticVars.jl:6-73, TendencyVars.jl:7-49) */
typedef enum {
    MOKA_F_SSH = 0,                  /* (nCells)            time levels 0,1 */
    MOKA_F_NORMAL_VELOCITY = 1,      /* (K, nEdges)         time levels 0,1 */
    MOKA_F_LAYER_THICKNESS = 2,      /* (K, nCells)         time levels 0,1 */
    MOKA_F_LAYER_THICKNESS_EDGE = 3, /* (K, nEdges)                         */
    MOKA_F_THICKNESS_FLUX = 4,       /* (K, nEdges)                         */
    MOKA_F_VELOCITY_DIV_CELL = 5,    /* (K, nCells)                         */
    MOKA_F_RELATIVE_VORTICITY = 6,   /* (K, nVertices)                      */
    MOKA_F_TEND_NORMAL_VELOCITY = 7, /* (K, nEdges)                         */
    MOKA_F_TEND_LAYER_THICKNESS = 8  /* (K, nCells)                         */
} moka_field;

/* Forward-Euler order-of-evaluation switches; all set = the live reference step */
#define MOKA_FE_STALE_HEDGE      1  /* flux uses previous step's layerThicknessEdge (DiagnosticVars.jl:113-116) */
#define MOKA_FE_ACCUM_VORT       2  /* relativeVorticity accumulates (Operators.jl:135,142)                      */
#define MOKA_FE_LEVEL1_ONLY      4  /* "[1,j]" kernels touch level 1 only (Operators.jl:207,228 ...)             */
#define MOKA_FE_REFERENCE_COMPAT 7

typedef enum { MOKA_FORWARD_EULER = 0, MOKA_RUNGE_KUTTA_4 = 1 } moka_integrator;

/* ---- library / context ---------------------------------------------------------------- */
const char *moka_version(void);
/* replaces the `backend = CUDABackend()` choice, src/driver/mpas_ocean.jl:28 */
int  moka_ctx_create(int device, moka_ctx **out);
void moka_ctx_destroy(moka_ctx *ctx);
const char *moka_last_error(const moka_ctx *ctx);   /* ctx may be NULL: last error of the calling thread */
int  moka_sync(moka_ctx *ctx);                      /* KA.synchronize(backend) */
/* HIP-event timer on the compute stream (used by bench.py for the roofline figure) */
int  moka_timer_start(moka_ctx *ctx);
int  moka_timer_stop(moka_ctx *ctx, float *elapsed_ms);

/* ---- mesh ------------------------------------------------------------------------------ */
/* Host-only: validate, convert to 0-based, order cells (RCB patches / RCM), renumber edges and
 * vertices by first-touching cell, build per-entity records.  Needs no GPU. */
int  moka_plan_create(const moka_mesh_desc *desc, moka_plan **out);
void moka_plan_destroy(moka_plan *plan);
int  moka_plan_info(const moka_plan *plan, moka_mesh_info *info);
/* new_to_old[new] = caller's 0-based index; kind = MOKA_CELL / MOKA_EDGE / MOKA_VERTEX */
int  moka_plan_permutation(const moka_plan *plan, int kind, int32_t *new_to_old);
/* first cell / edge / vertex of every patch, nPatches+1 entries each */
int  moka_plan_patch_ranges(const moka_plan *plan, int32_t *cellStart, int32_t *edgeStart, int32_t *vertexStart);
/* patch / cell / edge range of every cell class (see moka_mesh_class_ranges) */
int  moka_plan_class_ranges(const moka_plan *plan, int32_t capacity, int32_t *nClasses, int32_t *patchStart,
                            int32_t *cellStart, int32_t *edgeStart);

/* Read-only view of one of the plan's per-entity record arrays (device layout, new numbering):
 * lets host tests check the reordered mesh without a GPU.  `data` stays valid until plan destroy. */
enum {
    MOKA_PA_EOC = 0, MOKA_PA_COC, MOKA_PA_MLTC, MOKA_PA_SDV, MOKA_PA_INVAREA, MOKA_PA_AREACELL, MOKA_PA_RSUM,
    MOKA_PA_EHDR, MOKA_PA_EOE, MOKA_PA_WOE, MOKA_PA_GINVDC, MOKA_PA_DCEDGE, MOKA_PA_DVEDGE, MOKA_PA_FEDGE,
    MOKA_PA_EOV, MOKA_PA_CV, MOKA_PA_HALO_START, MOKA_PA_HALO_EDGE, MOKA_PA_LEOC, MOKA_PA_LEOE,
    MOKA_PA_CREC, MOKA_PA_EREC, MOKA_PA_FEOE, MOKA_PA_PVSTART, MOKA_PA_PVLIST, MOKA_PA_LVOE
};
int  moka_plan_array(const moka_plan *plan, int which, const void **data, int64_t *count);

/* replaces Adapt.adapt_structure(backend, ::Mesh) (MPASMesh.jl:26; HorzMesh.jl:354) */
int  moka_mesh_create(moka_ctx *ctx, const moka_mesh_desc *desc, moka_mesh **out);
void moka_mesh_destroy(moka_mesh *mesh);
int  moka_mesh_info_get(const moka_mesh *mesh, moka_mesh_info *info);

/* ---- operators on host arrays (synchronous) -------------------------------------------- */
/* GradientOnEdge!(grad, h, Mesh)            src/ocn/Operators.jl:102-120 */
int moka_gradient_on_edge(moka_mesh *mesh, const double *scalarCell, double *gradEdge);
/* DivergenceOnCell!(div, V, temp, Mesh)     src/ocn/Operators.jl:46-74; temp (opt) receives V*dvEdge */
int moka_divergence_on_cell(moka_mesh *mesh, const double *vecEdge, double *tempEdge, double *divCell);
/* CurlOnVertex!(curl, V, Mesh)              src/ocn/Operators.jl:151-177; ACCUMULATES into curlVertex */
int moka_curl_on_vertex(moka_mesh *mesh, const double *vecEdge, double *curlVertex);
/* interpolateCell2Edge!(e, c, Mesh)         src/ocn/Operators.jl:179-222; nlev = 1 is the reference (level 1 only) */
int moka_interpolate_cell2edge(moka_mesh *mesh, const double *cellValue, double *edgeValue, int nlev);

/* Reverse (vjp) and forward (jvp) mode of the three operators -- what the reference obtains from Enzyme in
 * test/enzyme/test_Enzyme_Operators.jl:42-131 (gradient) and :137-225 (divergence); the rules a maintainer registers are in
 * julia/MokaHIPEnzymeExt.jl.  Host arrays of the primal's shapes.  Shadow conventions are Enzyme's for in-place kernels with
 * Duplicated arguments: vjp ACCUMULATES into the shadow of the input and ZEROES the shadow of an output the primal overwrites
 * (grad; div and temp); the shadow of the curl output, which the primal accumulates into, is left as it is.  The operators
 * are linear, so jvp is the operator applied to the tangent of the input (curl: accumulated onto the tangent of the output).
 * Transposes are gathers with a fixed summation order (oracle twins: oracle_*_vjp, bit-identical). */
int moka_gradient_on_edge_vjp(moka_mesh *mesh, double *dGradEdge /* in, zeroed */, double *dScalarCell /* += */);
int moka_gradient_on_edge_jvp(moka_mesh *mesh, const double *dScalarCell, double *dGradEdge);
int moka_divergence_on_cell_vjp(moka_mesh *mesh, double *dDivCell /* in, zeroed */, double *dVecEdge /* += */,
                                double *dTempEdge /* (opt) in, zeroed */);
int moka_divergence_on_cell_jvp(moka_mesh *mesh, const double *dVecEdge, double *dTempEdge /* (opt) */, double *dDivCell);
int moka_curl_on_vertex_vjp(moka_mesh *mesh, const double *dCurlVertex, double *dVecEdge /* += */);
int moka_curl_on_vertex_jvp(moka_mesh *mesh, const double *dVecEdge, double *dCurlVertex /* += */);

/* ---- state ----------------------------------------------------------------------------- */
/* PrognosticVars/DiagnosticVars/TendencyVars constructors with KA.zeros on the backend
 * (PrognosticVars.jl:59-106, DiagnosticVars.jl:75-99, TendencyVars.jl:51-67); nTimeLevels = 2 */
/* A mesh created with stateBytes = 4 yields an fp32-STORAGE state: host arrays stay double (upload rounds to fp32,
 * download widens); moka_tendencies / moka_step_rk4 / moka_step_fe / moka_run / moka_sum_sq / the halo API work on it.
 * moka_step_fe steps all levels of a whole mesh (no MOKA_FE_LEVEL1_ONLY; on a partitioned mesh: moka_fe_dist_step).  DiagnosticVars come out of
 * Forward-Euler steps only: after an RK4 step they are unavailable (MOKA_ERR_UNSUPPORTED on download), and the next
 * Forward-Euler step must carry none over (flags 0).  The piecewise calls (moka_diagnostic_compute, moka_compute_*_tendency,
 * moka_advance_time_levels with level-1-only flags) stay Float64-only.  Needs nVertLevels % 4 == 0 and <= 128. */
int  moka_state_create(moka_ctx *ctx, moka_mesh *mesh, moka_state **out);
void moka_state_destroy(moka_state *st);
/* Adapt.adapt(backend, array) / Adapt.adapt_structure(KA.CPU(), x) (OutPut.jl:122-124).
 * time_level: 0 = previous, 1 = current (reference indices 1 and end); ignored for Diag/Tend. */
int  moka_state_upload(moka_state *st, int field, int time_level, const double *host);
int  moka_state_download(moka_state *st, int field, int time_level, double *host);

/* Selected rows of a field (caller's 0-based ids; any order, repeats allowed) into host (nVertLevels, nRows) -- (1, nRows) for
 * ssh -- widened to double: what a row-sampled check of a full-size state needs without moving the whole field across PCIe.
 * time_level 0 / 1 as moka_state_download; 2 / 3 = the two RK4 provisional states as the last stage launches left them
 * (inspection only: prognostic fields; they exist after the first RK4 step). */
int  moka_state_download_rows(moka_state *st, int field, int time_level, int64_t nRows, const int32_t *rows, double *host);

/* Device address of a prognostic array (inspection; time_level as above, 0 when the array does not exist yet). */
int  moka_state_array_address(moka_state *st, int field, int time_level, uint64_t *address);

/* Placement of the state's arrays in device memory.  Where the allocator puts the four buffer sets an RK4 step streams through
 * (current level, previous level = New accumulator, two provisional states) decides 5-14 % of every stage launch (DESIGN.md
 * section 5): a stable property of the memory behind an array, not visible in its address.  The reference's driver only ever
 * calls ocn_init and the step (src/driver/mpas_ocean.jl:28-39), so the choice is made here, behind the boundary: the four
 * launches of an RK4 step are timed with the library's own events (dt = 0), then up to max_tries times ONE array is given a
 * second allocation while the first is still alive, the launches that array takes part in are timed again, and the faster
 * allocation is kept (the loser is held back, within a quarter of the free memory, so that it is not handed out again; all
 * are released on return).  Peak extra memory: the previous level (saved / restored) + one candidate + the held-back losers.
 * The state's observable contents are unchanged (every array of Prog / Diag / Tend, both time levels, what is lazily
 * pending); only the provisional RK states, which every RK4 step overwrites before reading, are clobbered.
 * *ms_before / *ms_after (opt): sum of the four stage launches' median times before / with the kept layout.
 * max_tries <= 0 measures only.  Stops early after 12 consecutive trials without a gain (every array tried once: normalVelocity,
 * layerThickness and ssh of the four buffer sets).
 * MOKA_ERR_UNSUPPORTED once a halo or a tape of the state exists (they export / hold addresses): call it right after
 * moka_state_create (+ uploads), which is what julia/MokaHIP.jl, moka_hip.shim and moka_hip.parallel do. */
int  moka_state_optimize_placement(moka_state *st, int max_tries, double *ms_before, double *ms_after);
/* What the last moka_state_optimize_placement tried: field = 3 * set + (0 normalVelocity, 1 layerThickness, 2 ssh), set 0 = current
 * level, 1 = previous level, 2 / 3 = RK provisional states; ms_old / ms_new = summed median times of the stage launches the
 * array takes part in with the old / the candidate allocation; kept = 1 when the candidate replaced the old one.
 * *n = number of trials; out receives min(*n, capacity) of them (may be NULL). */
typedef struct { int32_t field; double ms_old, ms_new; int32_t kept; } moka_placement_trial;
int  moka_state_placement_log(const moka_state *st, int32_t capacity, moka_placement_trial *out, int32_t *n);
/* stage launches the last search issued (bookkeeping for whoever lines a kernel trace up with a timed region) */
int64_t moka_state_placement_launches(const moka_state *st);

/* advanceTimeLevels!(Prog)                  src/forward/time_integration.jl:10-40 */
int moka_advance_time_levels(moka_state *st, int flags);
/* diagnostic_compute!(Mesh, Diag, Prog)     src/ocn/DiagnosticVars.jl:108-117 (flags: MOKA_FE_*) */
int moka_diagnostic_compute(moka_state *st, int flags);
/* computeNormalVelocityTendency!            src/ocn/Tendencies/normalVelocity/normalVelocity.jl:21-53 */
int moka_compute_normal_velocity_tendency(moka_state *st, int flags);
/* computeLayerThicknessTendency!            src/ocn/Tendencies/layerThickness/layerThickness.jl:14-28 */
int moka_compute_layer_thickness_tendency(moka_state *st, int flags);
/* Fused tendency evaluation (u,h) -> (tendU,tendH) with consistent diagnostics:
 * the kernel the roofline figure is quoted on. */
int moka_tendencies(moka_state *st);
/* ocn_timestep(timestep, Prog, Diag, Tend, S, ForwardEuler)   time_integration.jl:150-193 */
int moka_step_fe(moka_state *st, double dt, int flags);
/* ocn_timestep(Prog, Diag, Tend, S, RungeKutta4)              time_integration.jl:61-148 (spec) */
int moka_step_rk4(moka_state *st, double dt);
/* the body of ocn_run_loop                   src/forward/run_loop.jl:8-22 : nsteps x ocn_timestep */
int moka_run(moka_state *st, int integrator, double dt, int64_t nsteps, int flags);
/* sumArray                                   src/forward/run_loop.jl:39-51 : sum_j a[j]^2 */
int moka_sum_sq(moka_state *st, int field, int time_level, double *out);

/* ---- multi-GPU (one process per GPU; the reference has no distributed code) -----------------
 * The mesh handed to moka_mesh_create is the rank's LOCAL mesh: owned cells + a one-cell-deep halo, with
 * cellClass = 0 (owned, needed by another rank) / 1 (owned interior) / 2 + i (halo cells owned by the rank's i-th
 * neighbour; plain 2 for all of them also works, but then only the buffered transport is available).  The plan orders
 * cells class-major and never lets a patch straddle a class: moka_mesh_class_ranges reports the patch / cell / edge range
 * of every class, moka_mesh_permutation the library's numbering (new -> caller's), which is the order in which a rank
 * has to list the halo cells / edges it receives for the direct transport.
 *
 * Two transports, per neighbour and stage the rows [layerThickness | ssh | normalVelocity] of the listed cells / edges:
 *   buffered: moka_halo_pack -> one contiguous device buffer -> the host layer moves it (torch.distributed on RCCL over
 *     xGMI in bench.py; gloo in the tests; MPI.jl from Julia) -> moka_halo_unpack.  Buffer layout, per neighbour i:
 *     [K reals of layerThickness for cells[off[i]..off[i+1])] [ssh of the same cells] [K reals of normalVelocity for
 *      edges[off[i]..off[i+1])]   (reals = double, or float for an fp32-storage state; sizes in elements)
 *   direct: after moka_halo_export / moka_halo_connect with every neighbour, moka_halo_push_begin stores the rows straight
 *     into the neighbours' fields (peer-mapped memory over xGMI: hipIpcOpenMemHandle between processes, plain pointers
 *     inside one process), moka_halo_push_signal / _wait complete the exchange through flag words in host (shared)
 *     memory.  No send buffer, no unpack, no collective library, no kernel that spins on the device.
 * Entity ids are in the caller's (local mesh) numbering; the *Off arrays have nNeighbors+1 entries. */
typedef struct moka_halo moka_halo;
int  moka_ctx_streams(moka_ctx *ctx, void **compute_stream, void **comm_stream);   /* hipStream_t handles */
int  moka_mesh_permutation(const moka_mesh *mesh, int kind, int32_t *new_to_old);
/* class k covers patches [patchStart[k], patchStart[k+1]), cells [cellStart[k], ...), edges [edgeStart[k], ...) of the
 * library's numbering; the arrays take min(nClasses + 1, capacity) entries (any of them may be NULL) */
int  moka_mesh_class_ranges(const moka_mesh *mesh, int32_t capacity, int32_t *nClasses, int32_t *patchStart,
                            int32_t *cellStart, int32_t *edgeStart);
/* nPatchesBoundary / nPatchesOwned: patchStart[1] / patchStart[2] of moka_mesh_class_ranges */
int  moka_halo_create(moka_state *st, int32_t nNeighbors, const int32_t *sendCells, const int64_t *sendCellOff,
                      const int32_t *sendEdges, const int64_t *sendEdgeOff, const int32_t *recvCells,
                      const int64_t *recvCellOff, const int32_t *recvEdges, const int64_t *recvEdgeOff,
                      int32_t nPatchesBoundary, int32_t nPatchesOwned, moka_halo **out);
void moka_halo_destroy(moka_halo *h);
int  moka_halo_buffer_elems(const moka_halo *h, int64_t *sendElems, int64_t *recvElems);
/* what: 0 = current time level, 1..4 = output of RK4 stage `what`, 5 = the new level of a distributed Forward-Euler step
 * (before moka_fe_dist_end rotates the levels).  pack runs on the comm stream after the work already queued on the compute
 * stream; unpack makes later compute-stream work wait for it. */
int  moka_halo_pack(moka_halo *h, int what, void *sendbuf_device);
int  moka_halo_unpack(moka_halo *h, int what, const void *recvbuf_device);

/* direct transport.  What a rank tells neighbour `nbr` so that the neighbour can push to it (moka_halo_export), to be
 * carried to that neighbour by the host layer (any byte transport: it is plain data) and given to moka_halo_connect
 * there.  shared = 1: the two ranks are different processes (IPC handles + a POSIX shared-memory flag block);
 * shared = 0: same process (raw pointers; several devices are mapped with hipDeviceEnablePeerAccess). */
typedef struct {
    unsigned char ipc[15][64];   /* hipIpcMemHandle_t of the five buffer sets (two time levels, two RK provisional states, the
                                    Forward-Euler spare level) x (normalVelocity, layerThickness, ssh) */
    uint64_t ptr[15];            /* the same allocations as raw device pointers */
    uint64_t flagPtr;            /* the rank's flag block as a raw host pointer */
    char     shmName[64];        /* ... and as a POSIX shared-memory object (shared = 1) */
    int32_t  dstCell, dstEdge;   /* first cell / edge (library numbering) of the ranges the neighbour's rows go to */
    int32_t  nCells, nEdges;     /* their lengths: must equal what the neighbour lists as its send cells / edges */
    int32_t  slot;               /* the neighbour's slot in the flag block */
    int32_t  nNeighbors, pid, device, stateBytes, nVertLevels;
} moka_halo_peer_info;
/* How a distributed RK4 stage queues its two launches.  0: boundary patches, then interior patches, on the compute stream
 * (the exchange starts behind the first).  1: boundary patches on the high-priority communication stream, interior patches
 * on the compute stream, in flight together (a small boundary launch no longer leaves the chip idle).  -1 (default):
 * chosen from the size of the interior launch.  Takes effect at the next moka_rk4_dist_begin.  Results are identical. */
int  moka_halo_set_overlap(moka_halo *h, int mode);
/* on != 0: a system-scope acquire (cache invalidate on every XCD) is launched in front of every launch that reads received
 * rows (the boundary patches of a stage, the Forward-Euler vertex pass).  Default off: the acquire of a kernel dispatch makes
 * peer-written rows visible (csrc/halo.hip header); the transport selection falls back to this form ("ipc-acq") when whole steps
 * over the plain direct transport do not reproduce the host-staged exchange bit for bit. */
int  moka_halo_set_acquire(moka_halo *h, int on);
/* the same exchange for arbitrary device fields of the state's shapes and storage type (u-like, h-like, ssh-like), library
 * numbering: pack waits for the compute stream, unpack makes the compute stream wait for the received rows */
int  moka_halo_pack_fields(moka_halo *h, const void *uField, const void *hField, const void *sField, void *sendbuf);
int  moka_halo_unpack_fields(moka_halo *h, void *uField, void *hField, void *sField, const void *recvbuf);
int  moka_halo_direct_available(const moka_halo *h);   /* 1: the receive lists are contiguous ranges (see above) */
int  moka_halo_export(moka_halo *h, int32_t nbr, int32_t shared, moka_halo_peer_info *out);
int  moka_halo_connect(moka_halo *h, int32_t nbr, const moka_halo_peer_info *peer, int32_t shared);
int  moka_halo_push_begin(moka_halo *h, int what);        /* queue the push behind the work on the compute stream */
int  moka_halo_push_signal(moka_halo *h);                 /* host: wait for the own push, then signal the neighbours */
int  moka_halo_push_wait(moka_halo *h, double timeout_s); /* host: wait for every neighbour's signal (MOKA_ERR_COMM on timeout) */

/* Experiment: the direct transport's handshake as stream memory operations (hipStreamWriteValue64 behind the push kernel,
 * hipStreamWaitValue64 in front of the next boundary launch) instead of a host thread that waits for the push event, stores the
 * flags and polls the neighbours': a distributed step becomes enqueue-only.  Call after every neighbour is connected.  A stream
 * wait has no timeout (a dead neighbour blocks the queue until the process ends), so this is a transport CANDIDATE ("ipc-smo" in
 * moka_hip.parallel), qualified like the others; MOKA_ERR_UNSUPPORTED where the device or the flag memory does not allow it. */
int  moka_halo_set_stream_flags(moka_halo *h, int on);

/* Measurement: where a distributed step spends its time on this rank.  moka_halo_stats_enable(h, 1) forgets earlier samples and
 * starts recording, (h, 0) stops; moka_halo_stats_read synchronises the rank's two streams and returns the sums:
 *   steps / host_step_ms              moka_rk4_dist_step calls and the host time spent inside them
 *   exchanges                         direct exchanges signalled
 *   host_signal_wait_ms               host time in moka_halo_push_signal waiting for the rank's own push kernel (its event)
 *   host_flag_store_ms                ... and from that event to the last flag store (event -> flag)
 *   host_wait_ms                      host time in moka_halo_push_wait polling the neighbours' flags
 *   boundary_ / interior_launch_ms    device time of the boundary / interior stage launches (HIP events around each launch, on the
 *   boundary_ / interior_launches     stream it was queued to; at most 2048 launches are recorded) and how many were recorded */
typedef struct {
    int64_t steps, exchanges, boundary_launches, interior_launches;
    double host_step_ms, host_signal_wait_ms, host_flag_store_ms, host_wait_ms, boundary_launch_ms, interior_launch_ms;
} moka_halo_stats;
int  moka_halo_stats_enable(moka_halo *h, int on);
int  moka_halo_stats_read(moka_halo *h, moka_halo_stats *out);

/* distributed form of moka_step_rk4, piecewise: begin; for stage 1..4 { stage(s,0) boundary patches; pack(s) or
 * push_begin(s); stage(s,1) interior patches (overlaps the exchange); transport + unpack(s), or push_signal + push_wait };
 * end.  moka_rk4_dist_stage_launch(s) = stage(s,0) + push_begin(s) + stage(s,1).
 * part = 2: the whole local mesh in one launch, halo entities included (redundantly), exchange behind it.
 * States with the optional nonlinear terms (their stencil needs a halo two cells deep: build_local(rings = 2)) split a stage
 * into its preparation pass and its stage kernel: part 3 = preparation over the boundary and halo patches (after the previous
 * stage's exchange has arrived), part 4 = preparation over the interior patches (owned rows only: may be queued as soon as the
 * previous stage's launches are, and then overlaps that stage's exchange), parts 0 / 1 = the stage kernel over the boundary /
 * interior patches.  moka_rk4_dist_step orders them so; part 2 remains for the kernel variants without per-patch kernels
 * (moka_rk4_dist_parts_available == 0). */
int  moka_rk4_dist_begin(moka_halo *h, double dt);
int  moka_rk4_dist_stage(moka_halo *h, int stage, int part);
int  moka_rk4_dist_parts_available(const moka_halo *h);
int  moka_rk4_dist_stage_launch(moka_halo *h, int stage);
int  moka_rk4_dist_end(moka_halo *h);
/* ... and as ONE call per step.  transport == NULL: the direct exchange (every neighbour must be connected, MOKA_ERR_ARG
 * otherwise); a callback, when given, is always used, connected or not: it moves the packed send buffer of a stage to the
 * neighbours and fills the receive buffer (stream-ordered on the comm stream, or synchronously; returns 0 on success).
 * timeout_s bounds every wait of the direct form. */
typedef int (*moka_transport_fn)(void *user, int what, void *sendbuf_device, void *recvbuf_device);
int  moka_rk4_dist_step(moka_halo *h, double dt, moka_transport_fn transport, void *user, void *sendbuf_device,
                        void *recvbuf_device, double timeout_s);
/* distributed form of moka_step_fe (ocn_timestep(..., ForwardEuler), time_integration.jl:150-193) with the same flags:
 * part 0 boundary patches, 1 interior patches, 2 relativeVorticity; the new level is exchanged as `what` = 5 between
 * part 0 and the end; moka_fe_dist_end rotates the time levels (previous <- current <- new: a third level set takes the new
 * level, so the step can form the reference's stale layerThicknessEdge from the previous level's layerThickness).  Order with the direct transport: part 2 FIRST (it reads
 * old-level rows of halo edges, which a neighbour's next step overwrites once this rank's push has been signalled), then
 * part 0, push, part 1.  moka_fe_dist_step does all of it in one call, in that order. */
int  moka_fe_dist_launch(moka_halo *h, double dt, int flags, int part);
int  moka_fe_dist_end(moka_halo *h);
int  moka_fe_dist_step(moka_halo *h, double dt, int flags, moka_transport_fn transport, void *user, void *sendbuf_device,
                       void *recvbuf_device, double timeout_s);

/* kernel variant selection for measurement (all variants give identical results): 0 = auto [default: 11 when the mesh
 * allows it (even 34 <= nVertLevels <= 64), else 4 (nVertLevels >= 33), else 3]; 11 = record-staged, 16-byte lanes, two
 * entities per wave, own u rows cached in LDS; 4 = plain column kernel; 3 = generic index kernel.  The other design points
 * measured in rounds 1-3 (1, 2, 5-10, 12-14) lost and were removed in round 4 (their numbers: profiles/r0*_variants.txt);
 * moka_set_kernel_variant returns MOKA_ERR_UNSUPPORTED for them. */
int moka_kernel_variant_available(int variant);
int moka_set_kernel_variant(moka_ctx *ctx, int variant);
/* Process-wide launch-shape switches for A/B measurements (every setting gives identical results).  key 1: bit mask of the
 * modes (0 tendency, 1..3 RK4 stages, 4..6 Forward Euler, 10 / 11 lean Forward Euler) of the fp32-storage stage kernel that run as 512-thread workgroups
 * bounded to 128 registers = 4 waves per SIMD instead of 3 (default: mode 0, the tendency launch).  key 2: 0 = Forward-Euler
 * steps always gather the stored layerThicknessEdge (mode 4); 1 (default) = formed from the previous level when valid (mode 6).
 * key 3: relativeVorticity of a Forward-Euler step inside the stage launches (1, default) or as a launch of its own (0).
 * key 4: lean Forward-Euler steps (1, default) or every array stored every step (0).  key 5: launch shape of the nonlinear
 * stage kernel's patch form (0 default: potential vorticity of the patch's vertices in LDS, three workgroups per CU; 1: q_e of
 * its edge rows in LDS, two workgroups; 2 / 3: other shapes of the default).  key 6 (test hook): upper limit of the vertex rows
 * that form keeps resident (0 = what the LDS budget holds).
 * key 8: bit mask of the modes of the Float64 stage kernel whose large whole-range launches take two consecutive patches per
 * 512-thread workgroup, one row cache over both (default: the tendency launch and RK stage 1, modes 0 / 1, and the
 * 13-stream stage 1, mode 7; 0 = one patch per workgroup everywhere).
 * key 9: lean Forward-Euler launches through their own kernel instances (1, default: stage-kernel modes 10 / 11, the optional
 * DiagnosticVars / TendencyVars outputs compiled out) or through the general Forward-Euler instances (0; measurement).
 * key 7 is NOT result-neutral and therefore off by default: 1 = moka_step_rk4 / moka_run step Float64 states on whole meshes
 * with 13 instead of 16 state streams per step.  The reference accumulates New += b_s k_s through the stages
 * (time_integration.jl:134-135); here stages 1-3 store only the provisional states and stage 4 forms
 * New = C + ((P2 - C) + 2 (P3 - C) + (P4 - C)) / 3 + dt/6 k4 from own rows -- the same Runge-Kutta step up to round-off (a
 * few units in the last place of the state per step; oracle twin oracle_step_rk4_s13, tests/test_oracle_igw.py holds the
 * tolerance against the reference form).  With the nonlinear terms on: the same where the stage launch is the default patch
 * kernel (even 34 <= nVertLevels <= 64; twin oracle_step_rk4_nonlinear_s13).  Distributed, taped and fp32-storage steps keep
 * the reference's form.  moka_state_rk4_streams tells which form the next moka_step_rk4 of a state takes. */
int moka_set_tuning(int key, int value);
int moka_get_tuning(int key, int *value);
/* 13 = the next moka_step_rk4 / moka_run of this state runs in the 13-stream form (key 7 set and the state qualifies), 16 = the
 * reference's running sum (time_integration.jl:134-135); 0 for NULL. */
int moka_state_rk4_streams(const moka_state *st);
/* Per-stage durations of moka_step_rk4 from HIP events on the compute stream (measurement: bench.py's per-mode roofline
 * lines).  moka_stage_timing(ctx, 1) forgets earlier samples and records 5 events per step from now on; (ctx, 0) stops.
 * moka_stage_timing_read: ms[s-1] = mean duration of the stage-s launch over the *steps recorded steps. */
int moka_stage_timing(moka_ctx *ctx, int enable);
int moka_stage_timing_read(moka_ctx *ctx, double ms[4], int64_t *steps);
/* PCI bus id of the context's device ("0000:c5:00.0"; buf of >= 16 bytes): the key to its clock / power files in sysfs and to
 * counting the distinct devices the ranks of a launch really use (measurement). */
int moka_ctx_pci_bus_id(moka_ctx *ctx, char *buf, int32_t len);
/* Per-step statistics of a timed region (measurement: bench.py's median / min / max): moka_mark records one HIP event on the
 * compute stream, n marks give n - 1 intervals; moka_marks_read: ms[i] = time between marks i and i + 1 (at most `capacity`
 * of them), *n = how many exist; moka_marks_reset forgets them. */
int moka_mark(moka_ctx *ctx);
int moka_marks_reset(moka_ctx *ctx);
int moka_marks_read(moka_ctx *ctx, int64_t capacity, double *ms, int64_t *n);
/* Same-run bandwidth calibration (measurement): a plain 16-byte-per-lane copy and a read-only sweep over a buffer of `bytes`
 * (two halves), each launch timed with HIP events on the compute stream; gbs[0] = best copy rate (bytes read + written, GB/s),
 * gbs[1] = best read-only rate, gbs[2] = mean copy rate over the `iters` launches, gbs[3] = best rate of a scattered gather of
 * 480-byte rows, a half-wave per row (the stage kernels' access pattern).  The buffer is kept for the next probe;
 * bytes = 0 frees it.  (MI355X_MICROARCH.md quotes 6.29 TB/s for such a copy; SURVEY.md 8d asks for the copy and read figures.) */
int moka_bw_probe(moka_ctx *ctx, int64_t bytes, int iters, double gbs[4]);
/* ... and a fifth figure: three streams read and two written at once (the mix of the RK stage launches), over the buffer
 * moka_bw_probe allocated: GB/s of all five streams, best of `iters` launches */
int moka_bw_probe_streams(moka_ctx *ctx, int iters, double *gbs);
/* ... and a sixth: `region` bytes of that buffer (between the L2s' 32 MB and the Infinity Cache's 256 MB) read `reps` times back to
 * back -- the rate at which re-read data returns; GB/s over all passes, best of `iters` */
int moka_bw_probe_reread(moka_ctx *ctx, int64_t region, int reps, int iters, double *gbs);
/* ... and a seventh: the scattered row gather over a footprint of its own of `bytes` (allocated and freed here; e.g. 32 GiB, the span
 * of a state with all its arrays), a quarter of the rows once each: nearly every fetch needs an address translation the TLBs do not hold */
int moka_bw_probe_gather_big(moka_ctx *ctx, int64_t bytes, int iters, double *gbs);
/* which kernel the last moka_step_fe of this state used: 1 = the tuned stage kernel (+ vertex pass), 2 = the same with the
 * stale layerThicknessEdge formed from the previous level's layerThickness instead of gathered (every MOKA_FE_STALE_HEDGE
 * step after the first of a run), 0 = the generic one-launch kernel, -1 = no Forward-Euler step yet.  For tests and
 * measurement; results are identical either way. */
int moka_last_fe_path(const moka_state *st);
/* 1 while the TendencyVars / DiagnosticVars of the last Forward-Euler step are pending.  With the default kernel choice a
 * Forward-Euler step of all levels is LEAN: it stores the new time level and relativeVorticity; tendNormalVelocity,
 * tendLayerThickness, thicknessFlux, velocityDivCell and layerThicknessEdge of the step are produced on the first read
 * (download, sum_sq, upload, the piecewise reference calls, a taped step ...), from the level the step started from -- same
 * arithmetic, same bits -- and superseded by the next step otherwise, like the DiagnosticVars an RK4 step leaves pending.
 * The first step of a run with MOKA_FE_STALE_HEDGE, steps on a tape, MOKA_FE_LEVEL1_ONLY steps and moka_set_tuning(4, 0)
 * store everything at once. */
int moka_fe_lazy_pending(const moka_state *st);

/* ---- optional nonlinear terms (extension; NOT in the reference, parity unpinned) ---------------------------------
 * north_star names potential-vorticity Coriolis over edgesOnEdge, the KE + ssh gradient over cellsOnEdge and vertex
 * relativeVorticity; the reference has only the linear f*u_perp term (horizontal_advection_and_coriolis.jl:61-73,
 * SURVEY.md N4).  on = 1 switches moka_tendencies / moka_step_rk4 / moka_run(RK4) of this state to the
 * vector-invariant TRiSK form (Ringler et al. 2010):  q = (fVertex + zeta)/h_vertex at vertices, averaged to edges;
 *   tendU = sum_i w[i,e] F[eoe_i] (q_e + q_eoe_i)/2 - g grad(ssh) - grad(KE),  KE = sum_e dc dv u^2 / (4 areaCell).
 * Default off.  Needs the optional mesh arrays above; Float64 states on whole meshes; no Forward Euler, no tape. */
int  moka_set_nonlinear(moka_state *st, int on);
/* Del2 momentum mixing on top of the nonlinear terms -- the reference's sketch (never called there, and not runnable:
 * src/ocn/Tendencies/normalVelocity/horizontal_momentum_mixing.jl:53-80, with viscDel2 hard-wired to 1.0 at :30):
 *   tendU[k,e] += ((div[k,c2] - div[k,c1]) / dcEdge[e] - (relVort[k,v2] - relVort[k,v1]) / dvEdge[e]) * viscDel2
 * with div = velocityDivCell and relVort = relativeVorticity of the stage's provisional velocity.  0 (default) = off.
 * MOKA_ERR_UNSUPPORTED unless moka_set_nonlinear(st, 1) came first. */
int  moka_set_viscosity_del2(moka_state *st, double viscDel2);

/* ---- reverse mode of the Forward-Euler loop ----------------------------------------------------------------
 * The reference gets d sum(ssh^2) / d (initial normalVelocity, layerThickness) from Enzyme over ocn_run_loop
 * (ext/MPASEnzymeExt.jl, test/enzyme/test_Enzyme_end2end.jl:62-96; on CUDA it yields NaNs there, :183-186).
 * Here: a tape of the forward values each step's transpose needs, and hand-written transposed kernels in gather
 * form (no atomics).  Float64 states on whole (unpartitioned) meshes; flags without MOKA_FE_LEVEL1_ONLY unless
 * nVertLevels = 1.  Usage: tape_create -> n x step_fe_taped -> seed -> sweep -> download. */
typedef struct moka_tape moka_tape;
int  moka_tape_create(moka_state *st, int64_t capacity_steps, moka_tape **out);   /* 2*K*nEdges doubles per step */
void moka_tape_destroy(moka_tape *t);
int  moka_step_fe_taped(moka_tape *t, double dt, int flags);                        /* records, then moka_step_fe */
/* RK4 (time_integration.jl:61-148): records the four provisional states, 4*K*(nEdges+nCells) doubles per step.
 * One integrator per tape.  The RK4 map's state is (normalVelocity, layerThickness): the ssh gradient is zero. */
int  moka_step_rk4_taped(moka_tape *t, double dt);
/* lambda := d sum(ssh^2) / d state at the current state (the objective of run_loop.jl:26-45) */
int  moka_adjoint_seed_sum_sq_ssh(moka_tape *t);
int  moka_adjoint_sweep(moka_tape *t);                                              /* reverse over (and pop) every recorded step */
/* Reverse mode of a PARTITIONED RK4 run (one rank's part; the host layer moves the halo rows between the calls):
 *   taping   -- the caller runs the distributed step itself and records, halo rows included, the four provisional states:
 *               slot 0 := the current level before the step (what = 0), slot s := the output of stage s after its exchange
 *               (what = s = 1..3); moka_tape_commit_rk4 closes the step.
 *   reversal -- per recorded step, for sg = 4, 3, 2, 1: exchange the halo rows of the fields moka_adjoint_rk4_stage_fields
 *               names (moka_halo_pack_fields / unpack_fields; sg = 4: the adjoint state itself), then moka_adjoint_rk4_stage.
 * The transposed lists are sorted by the caller's edge ids: a local mesh whose edges keep the order of their global ids
 * (moka_hip.parallel.build_local) reproduces the single-domain sums bit for bit. */
int  moka_tape_record_rk4(moka_tape *t, int slot, int what);
int  moka_tape_commit_rk4(moka_tape *t, double dt);
int  moka_adjoint_rk4_stage_fields(moka_tape *t, int stage, void **fieldU, void **fieldH, void **scratchS);
int  moka_adjoint_rk4_stage(moka_tape *t, int stage);
/* ... and with the exchange overlapped: moka_adjoint_rk4_stage_part(t, sg, 0) transposes the entities of the boundary cell class
 * (what other ranks gather from), the caller starts the exchange of the rows that produced (moka_adjoint_rk4_stage_out_fields names
 * the arrays stage sg writes for the next transposed stage; after sg = 1: the adjoint state itself), part 1 = the interior class
 * runs meanwhile (owned rows only), then the exchange completes.  Halo entities are not computed.  Chunk kernels only
 * (even 34 <= nVertLevels <= 64), MOKA_ERR_UNSUPPORTED otherwise. */
int  moka_adjoint_rk4_stage_part(moka_tape *t, int stage, int part);
int  moka_adjoint_rk4_parts_available(const moka_tape *t);   /* 1: moka_adjoint_rk4_stage_part serves this tape's mesh */
int  moka_adjoint_rk4_stage_out_fields(moka_tape *t, int stage, void **fieldU, void **fieldH, void **scratchS);
/* Forward-Euler runs on a partitioned mesh: record before (after = 0) and behind (after = 1) the distributed step, commit;
 * reversal per recorded step: exchange the halo rows of the three fields moka_adjoint_fe_step_fields names (the adjoints of
 * normalVelocity, layerThickness and ssh; that of the carried layerThicknessEdge needs none), then moka_adjoint_fe_step. */
int  moka_tape_record_fe(moka_tape *t, int flags, int after);
int  moka_tape_commit_fe(moka_tape *t, double dt, int flags);
int  moka_adjoint_fe_step_fields(moka_tape *t, void **fieldU, void **fieldH, void **fieldS);
int  moka_adjoint_fe_step(moka_tape *t);
/* field: MOKA_F_SSH, MOKA_F_NORMAL_VELOCITY, MOKA_F_LAYER_THICKNESS (d_Prog of the reference test) or
 * MOKA_F_LAYER_THICKNESS_EDGE (the carried diagnostic of the reference_compat sequence); caller's numbering */
int  moka_adjoint_download(moka_tape *t, int field, double *host);

#ifdef __cplusplus
}
#endif
#endif /* MOKA_HIP_H */
