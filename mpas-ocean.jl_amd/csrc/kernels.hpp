// kernels.hpp -- argument blocks and launchers of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "moka_internal.hpp"

namespace moka {

// Fused tendency / RK-stage kernel.  NULL pointers switch the corresponding read/write off.
struct StageArgs {
    const double *pu, *ph;        // provisional state: the gather source of the tendency
    const double *ssh;            // ssh of the provisional state (nC)
    const double *cu, *ch;        // Curr (NULL: Curr == Provis, e.g. RK stage 1 / Forward Euler)
    const double *nu_in, *nh_in;  // New accumulator in (NULL: start from Curr)
    double *nu_out, *nh_out;      // New accumulator out
    double *pu_out, *ph_out;      // next provisional state  Curr + a*tend
    double *ssh_out;              // ssh of ph_out (or of nh_out when ph_out == NULL)
    double *tendU, *tendH;        // tendencies
    double a, b;
    // Forward-Euler step in the default stage kernel (k_stage_rec2c modes 4 / 5; every other kernel ignores these):
    // pu/ph/ssh = current level, pu_out/ph_out/ssh_out = new level, a = dt, tendU/tendH, and the diagnostics below
    // feMode != 0 marks a Forward-Euler launch and names its kernel mode (4 / 5 / 6).  Every OUTPUT group of such a launch is
    // optional: {pu_out, ph_out, ssh_out} = the new level; {tendU, tendH, F, div, hEdgeNew} = the step's TendencyVars /
    // DiagnosticVars; vort.  A lean step passes the first (and vort), the launch that materialises a lean step's arrays on
    // demand passes the second.
    int feMode;
    const double *hEdgeOld;       // previous step's layerThicknessEdge (mode 4: MOKA_FE_STALE_HEDGE) or NULL (modes 5, 6)
    const double *hPrev;          // mode 6: the previous time level's layerThickness, which hEdgeOld is the interpolation of
    double *hEdgeNew, *F, *div;   // layerThicknessEdge, thicknessFlux, velocityDivCell
    const double *areaCell;
    // relativeVorticity of the OLD state (CurlOnVertex, Operators.jl:137-146) by the same launch: the vertices of the launched
    // patches, normalVelocity rows from the patch's LDS row cache where it has them.  NULL: the caller launches the vertex
    // pass itself (k_curl3 / k_fe).  vertexDegree 3 meshes with byte-offset records only (stage_curl_fused()).
    double *vort;
    int accumVort;                // MOKA_FE_ACCUM_VORT: on top of what the array holds (Operators.jl:135,142)
    // 13-stream RK4 form (k_stage_rec2c modes 7 / 8 / 9; Float64 states): rkMode names the mode.  7: pu/ph -> pu_out/ph_out/ssh_out;
    // 8: + cu/ch; 9: pu/ph = P4 (gathered), own rows cu/ch = Curr, nu_in/nh_in = P2, q3u/q3h = P3 -> nu_out/nh_out (may alias
    // nu_in/nh_in: every entity reads its own row of P2 before it writes there), ssh_out = ssh of New, b = dt/6
    int rkMode;
    const double *q3u, *q3h;
};

// the slice of MeshDev the column kernel reads (kept small: kernel arguments live in SGPRs)
struct ColMesh {
    int32_t nC, nE, K, nPatches, patchBegin, CI, EI;
    const int32_t *patchCellStart, *patchEdgeStart;
    const uint32_t *cRec, *eRec;
    const int32_t *mltc;
    const double *sdv, *invArea, *rsum, *woe, *feoe, *gInvDc;
    int32_t tailPlus1;   // != 0: the launch's last workgroup takes patch tailPlus1 - 1 instead of patchBegin + nPatches - 1
    // vertex pass of the Forward-Euler modes (StageArgs.vort): the patches' vertex ranges, u-row offsets and coefficients
    const int32_t *patchVertStart;
    const uint32_t *vRec;
    const double *cv;
    int32_t maxOwnV;
    // > 0: every workgroup of this launch takes TWO consecutive patches (2 * block, 2 * block + 1 of the launched range, the last one
    // alone when their number is odd) as one unit -- one staging phase, one row cache over both patches' own edges -- and pairEnd
    // is the end of the launched patch range (absolute).  k_stage_rec2c, 512 threads; see launch_stage_rec2c.
    int32_t pairEnd;
};

enum : int {
    FE_FLUX = 1, FE_DIV = 2, FE_CURL = 4, FE_HEDGE = 8, FE_TENDU = 16, FE_TENDH = 32, FE_UPDATE = 64,
    FE_TENDH_FROM_F = 128
};

struct FeArgs {
    int ops, flags, nlev;
    double dt;
    const double *u, *h, *ssh;    // current time level
    const double *hEdgeOld;       // layerThicknessEdge as the previous step left it
    const double *Fin;            // stored thicknessFlux (FE_TENDH_FROM_F)
    double *hEdgeNew, *F, *div, *vort, *tendU, *tendH;
    double *u_new, *h_new, *ssh_new;
};

// OP_*_T: the transposes (reverse mode of the stand-alone operators, moka_*_vjp), gather form, fixed summation order
enum : int { OP_GRADIENT = 0, OP_INTERP = 1, OP_DIV_P1 = 2, OP_DIV_P2 = 3, OP_CURL = 4, OP_GRAD_T = 5, OP_DIV_T = 6, OP_CURL_T = 7 };

struct OpArgs {
    int op, nlev;
    const double *in;
    double *out;
    // transposes only
    const double *in2 = nullptr;       // OP_DIV_T: the shadow of temp (added in before the multiplication by dvEdge)
    const int32_t *auxI = nullptr;     // OP_CURL_T: (W, nE) vertices whose edgesOnVertex lists name the edge, by (caller's vertex id, slot); -1 = none
    const double *auxD = nullptr;      // OP_CURL_T: (W, nE) their coefficients; OP_DIV_T: (2, nE) edgeSignOnCell of the edge in c1, c2
    int auxW = 0;
};

// the product's stage kernels: default (kernels.hip) and the two fallbacks (stage_fallback.hip)
hipError_t launch_stage(const MeshDev &m, const StageArgs &a, int lpc, hipStream_t s);          // generic index kernel
hipError_t launch_stage_col(const MeshDev &m, const StageArgs &a, hipStream_t s);               // plain column kernel
hipError_t launch_stage_rec2c(const MeshDev &m, const StageArgs &a, hipStream_t s);
bool rec2c_supported(const MeshDev &m);
// can the Forward-Euler modes of the stage kernels carry the relativeVorticity pass of this mesh (StageArgs.vort)
bool stage_curl_fused(const MeshDev &m);
bool stage_curl_fits(const MeshDev &m, bool f32);      // ... with patches of m.maxOwnE / m.maxOwnC own edges / cells
// fp32-state form (state pointers of StageArgs are float arrays); stage_f32_supported: can this mesh carry one
bool stage_f32_supported(const MeshDev &m);
hipError_t launch_stage_rec2c_f32(const MeshDev &m, const StageArgs &a, hipStream_t s);
void set_f32_wide_modes(int mask);      // measurement: which modes of the fp32-storage kernel run as (512 threads, 4 waves per SIMD)
int f32_wide_modes();
void set_curl_fused(int on);            // measurement: 0 = the Forward-Euler vertex pass always gets a launch of its own
int curl_fused();
void set_fe_lean(int on);               // measurement: 0 = Forward-Euler steps always store every array (no lean steps)
int fe_lean_enabled();
void set_fe_prev_mode(int on);          // measurement: 0 = never form the stale layerThicknessEdge from the previous level (mode 6)
int fe_prev_mode();
void set_fe_lean_instances(int on);    // measurement: 0 = lean Forward-Euler launches through the general instances (modes 5 / 6) instead of 10 / 11
int fe_lean_instances();
void set_pair_modes(int mask);          // measurement: which modes of the Float64 stage kernel take two patches per 512-thread workgroup
int pair_modes();
hipError_t launch_update_ssh_f32(const MeshDev &m, const float *h, float *ssh, int nlev, int lpc, hipStream_t s);
hipError_t launch_permute_rows_f32(void *dst, const void *src, const int32_t *n2o, int64_t n, int K, int to_device,
                                   hipStream_t s);
hipError_t launch_halo_map_f32(float *buf, float *h, float *ssh, float *u, const uint32_t *map, int64_t n, int unpack,
                               hipStream_t s);
hipError_t launch_fe(const MeshDev &m, const FeArgs &a, int lpc, hipStream_t s);
hipError_t launch_curl2(const MeshDev &m, const double *u, double *vort, bool accum, hipStream_t s);
hipError_t launch_curl_f32(const MeshDev &m, const float *u, float *vort, bool accum, hipStream_t s);   // fp32-storage states
hipError_t launch_operator(const MeshDev &m, const OpArgs &a, int lpc, hipStream_t s);
hipError_t launch_update_ssh(const MeshDev &m, const double *h, double *ssh, int nlev, int lpc, hipStream_t s);
hipError_t launch_permute_rows(double *dst, const double *src, const int32_t *n2o, int64_t n, int K, int to_device,
                               hipStream_t s);
hipError_t launch_sum_sq_serial(const double *a, int64_t n, double *out, hipStream_t s);
hipError_t launch_copy(double *dst, const double *src, int64_t n, hipStream_t s);
// bandwidth probes of the same-run calibration (moka_bw_probe): 16-byte-per-lane copy / read-only sweep of `bytes`
hipError_t launch_bw_copy(void *dst, const void *src, int64_t bytes, int nCUs, hipStream_t s);
hipError_t launch_bw_streams(void *buf, int64_t bytes, hipStream_t s);
hipError_t launch_bw_gather_n(const void *src, int64_t nRows, uint32_t rowB, int64_t nFetch, uint32_t *sink, int nCUs, hipStream_t s);
hipError_t launch_bw_read(const void *src, int64_t bytes, uint32_t *sink, int nCUs, hipStream_t s);
hipError_t launch_bw_gather(const void *src, int64_t bytes, uint32_t rowB, uint32_t *sink, int nCUs, hipStream_t s);
// halo pack / unpack: rows of (K doubles) gathered into / scattered from a contiguous buffer
hipError_t launch_halo_map(double *buf, double *h, double *ssh, double *u, const uint32_t *map, int64_t n, int unpack,
                           hipStream_t s);
hipError_t launch_pack_rows(double *buf, const double *field, const int32_t *rows, int64_t n, int K, int unpack, hipStream_t s);

// ---- optional nonlinear terms (not in the reference): scratch arrays of the preparation passes ----
struct NlArgs {
    double *qv;     // (K, nV) potential vorticity at vertices
    double *fq;     // (K, nE) pairs {thickness flux u * layerThicknessEdge, potential vorticity averaged to the edge}
    double *ke;     // (K, nC) kinetic energy at cells
    // Del2 momentum mixing (moka_set_viscosity_del2; horizontal_momentum_mixing.jl:53-80): nullptr / 0 = term absent
    double *zv;     // (K, nV) relativeVorticity
    double *divc;   // (K, nC) velocityDivCell
    double visc;
};
// form: 0 = best available, 1 = patch kernels without the LDS q_e rows, 2 = 16-byte-lane entity kernels, 3 = generic lane-group kernels
// (prepare and stage must be called with the same form: forms 0 / 1 keep F alone in NlArgs.fq, forms 2 / 3 {F, q_e} pairs)
void set_nl_shape(int v);
int nl_shape();
bool nl_stage_is_nl5(const MeshDev &m, int lpc, int form);  // the nonlinear stage launch of this mesh is k_stage_nl5 (knows StageArgs.rkMode 9)
bool nl_patch_forms(const MeshDev &m, int lpc, int form);   // do the nonlinear launches of this mesh go through the per-patch kernels (which serve patch ranges)
void set_nl_cap_limit(int v);
int nl_cap_limit();
hipError_t launch_nl_prepare(const MeshDev &m, const double *u, const double *h, const NlArgs &nl, int lpc, int form, hipStream_t s);
// rowsOk: the plan built the patch row lists (rowStart / rowEdge / leoe; Plan.ldsOk)
hipError_t launch_stage_nl(const MeshDev &m, const StageArgs &a, const NlArgs &nl, int lpc, bool rowsOk, int form, hipStream_t s);

// ---- reverse mode of one Forward-Euler step: gather form, the oracle's summation order ----
struct AdjMesh {
    int32_t nC, nE, K, ME, W;          // W = width of the transposed Coriolis lists
    const int32_t *eoc;                // (ME, nC) edges of a cell, -1 = none
    const int32_t *csgn;               // (ME, nC) edgeSignOnCell
    const int32_t *ehdr;               // (4, nE) c1, c2, -, maxLevelEdgeTop
    const int32_t *teoe;               // (W, nE) source edges s with edgesOnEdge[i,s] == e, sorted by (original s, i); -1 = none
    const double *tw;                  // (W, nE) weightsOnEdge[i,s]
    const double *sd;                  // (2, nE) dvEdge*edgeSign*invArea for c1, c2
    const double *fEdge, *gInvDc;      // (nE)
    const int32_t *efull;              // (nE) 1: all W sources exist and the edge and every source are active on all levels
    // the entities a launch of the chunk kernels (k_adj_edge3 / k_adj_cell3) covers: edges [eBegin, eBegin + eCount), cells
    // [cBegin, cBegin + cCount) -- everything by default, one cell class of a partitioned mesh in moka_adjoint_rk4_stage_part
    int32_t eBegin, eCount, cBegin, cCount;
};
struct AdjArgs {
    double dt;
    int stale;                         // MOKA_FE_STALE_HEDGE
    int tt;                            // 1: transpose of the tendency evaluation only (RK4 stages); 0: of a Forward-Euler step
    const double *u, *hEuse;           // forward values of the step (tape)
    const double *h;                   // tt: layerThickness of the stage (layerThicknessEdge is recomputed from it)
    const double *lamU1, *lamH1, *lamS1, *lamE1;
    double *lamU0, *lamH0, *lamS0;
    double *Enew, *csum;               // u*Fbar (K, nE); ksum_k dt*lamU1 (nE)
    // RK4 reverse sweep (tt = 1): the element-wise steps around T'(P)^T fused into the two kernels.  With accOut set the
    // result Pb = T'(P)^T k is not stored (lamU0 / lamH0 unused); instead, row by row,
    //   accOut = (accIn ? accIn : x) + Pb          X + Pb4, then (...) + Pb3 ...   (time_integration.jl:61-148 transposed)
    //   kNext  = cbNext * x + caNext * Pb          the next stage's k-bar (nullptr after the last stage)
    const double *xU, *xH, *accInU, *accInH;
    double *accOutU, *accOutH, *kNextU, *kNextH;
    double cbNext, caNext;
    // tt only.  lamScale: the k-bar actually used is lamScale * lamU1 / lamH1 (stage 4 reads X itself: kb4 = b4 * X is never
    // stored).  fuseE: u*Fbar is not stored by the edge kernel; the cell kernel recomputes it from the stage's u rows and the
    // k-bar rows of the edge's two cells (same products, same order) -- set only when adj_fused_available() says so.
    double lamScale;
    int fuseE;
};
// both fused forms (lamScale != 1, fuseE) need the 16-byte-lane chunk kernels: even K <= 64, hexagon-width lists
bool adj_fused_available(const AdjMesh &m, int lpc);
hipError_t launch_adj_edge(const AdjMesh &m, const AdjArgs &a, int lpc, hipStream_t s);
hipError_t launch_adj_cell(const AdjMesh &m, const AdjArgs &a, int lpc, hipStream_t s);
hipError_t launch_scale_copy(double *dst, const double *src, double f, int64_t n, hipStream_t s);   // dst = f*src
hipError_t launch_axpby(double *dst, double a, const double *x, double b, const double *y, int64_t n, hipStream_t s);
hipError_t launch_add(double *dst, const double *x, const double *y, int64_t n, hipStream_t s);       // dst = x + y
hipError_t launch_bcast_rows(double *dst, const double *src, double f, int64_t n, int K, hipStream_t s);   // dst[c][k] = f*src[c]

}  // namespace moka
