// state.hpp -- the objects behind the opaque handles of include/moka_hip.h and the helpers api.hip shares with halo.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "kernels.hpp"
#include "moka_internal.hpp"

struct moka_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t comm = nullptr;                       // halo pack / transport / unpack
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evBoundary = nullptr, evInterior = nullptr, evHalo = nullptr;
    int variant = 0;
    int nCUs = 256;
    std::string err;
    // per-stage HIP-event timing of moka_step_rk4 (moka_stage_timing): 5 events per recorded step, read back on request
    bool stageTiming = false;
    std::vector<hipEvent_t> evPool;          // events owned by the context (reused between measurements)
    size_t evUsed = 0;
    // per-step statistics of a timed region (moka_mark / moka_marks_read): one event per mark on the compute stream
    std::vector<hipEvent_t> marks;
    size_t marksUsed = 0;
    // same-run bandwidth calibration (moka_bw_probe): two halves of one allocation, kept between the probes of a run
    void *bwBuf = nullptr;
    size_t bwBytes = 0;
};

struct moka_mesh {
    moka_ctx *ctx = nullptr;
    moka::Plan plan;          // host copy (permutations, sizes)
    moka::MeshDev dev{};
    std::vector<void *> allocs;
    int lpc = 1;
    bool colOk = false;       // byte-offset records exist (every field < 4 GiB)
    double *opBuf[4] = {nullptr, nullptr, nullptr, nullptr};   // operator scratch [0], [1], [3] / transfer staging [2], lazily sized
    size_t opBufElems = 0;
    // transposed lists of the operator reverse mode (moka_*_vjp), built at first use
    const int32_t *opTVert = nullptr;     // (opTW, nE) vertices naming the edge, sorted by (caller's vertex id, slot); -1 = none
    const double *opTCoef = nullptr;      // (opTW, nE) their CurlOnVertex coefficients
    const double *opESign = nullptr;      // (2, nE) edgeSignOnCell of the edge in cellsOnEdge[1], [2] (0: not listed there)
    int opTW = 0;
    // (maxOwnE, maxOwnC) of a launched patch sub-range: a partition's halo-only patches own up to 6 edges per cell
    // and are never launched, so the LDS carve of a boundary / interior launch is sized by the patches it covers
    std::map<std::pair<int, int>, std::pair<int, int>> rangeMax;
};

struct LevelBufs {
    double *ssh = nullptr, *u = nullptr, *h = nullptr;
};

struct moka_state {
    moka_ctx *ctx = nullptr;
    moka_mesh *mesh = nullptr;
    LevelBufs lev[2];                 // [0] previous, [1] current   (reference Vector index 1 / end)
    double *hEdge[2] = {nullptr, nullptr};   // [0] is Diag.layerThicknessEdge, [1] the write target of the next step
    double *F = nullptr, *div = nullptr, *vort = nullptr, *tendU = nullptr, *tendH = nullptr;
    LevelBufs rk[2];                  // RK4 provisional states (lazily allocated)
    LevelBufs spare;                  // third time-level set of the tuned Forward-Euler step (lazily allocated): the step writes
                                      // the new level here, so the PREVIOUS level's layerThickness stays readable while it runs
                                      // (k_stage_rec2c mode 6), and the three sets rotate: prev <- cur <- new <- old prev
    static constexpr int NPHYS = 5;
    LevelBufs phys[NPHYS];            // the same buffer sets by allocation: [0],[1] the time levels as created (lev[] swaps /
                                      // rotates), [2],[3] = rk[], [4] = the Forward-Euler spare: what a neighbour rank addresses
                                      // when it pushes halo rows (halo.hip)
    // Diag.layerThicknessEdge (hEdge[0]) is, bit for bit, the interpolation of lev[0].layerThickness on every edge a launch
    // of this state computes: true after a Forward-Euler step of all levels through the stage kernel, false after anything
    // else that writes either array (uploads, RK4 steps, the piecewise reference calls, lazily produced diagnostics)
    bool hEdgePrev = false;
    // The last Forward-Euler step was LEAN: it stored the new time level (and relativeVorticity) only.  Its TendencyVars and
    // DiagnosticVars (tendNormalVelocity, tendLayerThickness, thicknessFlux, velocityDivCell, layerThicknessEdge) are pending:
    // produced on the first read (flush_lazy) from the level the step started from -- the previous level now -- and, for the
    // reference's stale flux thickness, the level before it, which the rotation of three level sets has kept in `spare`.
    // Same arithmetic, same bits as a step that stores everything; the next lean step reads none of those arrays and
    // supersedes them, exactly as an RK4 step supersedes its lazily produced diagnostics.
    bool feLazy = false, feLazyStale = false;
    bool feForceEager = false;        // a tape is recording: every step stores all of its arrays
    bool feLeanInteriorOnly = false;  // a direct (peer-store) halo is connected: a lean distributed step stores every array of its
                                      // BOUNDARY patches at once and leaves only the interior patches' pending (halo.hip header)
    int feLazyBegin = 0, feLazyCount = -1;   // the patch range whose arrays are pending (-1: every launched patch)
    double *scalar = nullptr;         // 1 double (sum_sq result)
    bool sshConsistent = false;       // lev[1].ssh == ksum(lev[1].h) - restingThicknessSum
    // moka_step_rk4 ends with diagnostic_compute! of the new state and leaves the stage-4 tendencies in
    // Tend (time_integration.jl:114-147).  Neither is needed by the next RK4 step, so they are produced
    // lazily -- on the first read (download, Forward-Euler step, reference-sequenced calls) -- with
    // results identical to computing them at the end of the step.
    bool diagDirty = false;
    bool tendDirty = false;           // stage-4 provisional state still sits in rk[0] -- or, after a taped step, in the tape:
    const double *lazyPu = nullptr, *lazyPh = nullptr;   // where (nullptr: rk[0]); ssh of it is rk[0].ssh either way
    const void *lazyOwner = nullptr;                     // the tape those rows belong to (it materialises them before it goes away)
    // fp32 storage of the prognostic fields (mesh stateBytes == 4): lev[] / rk[] then point at float arrays
    // (the pointer type stays double* so that one StageArgs block serves both), Diag arrays do not exist.
    bool f32 = false;
    // optional nonlinear terms (moka_set_nonlinear): scratch of the three preparation passes
    bool nonlinear = false;
    int nlPhase = 0;                  // what run_stage launches for a nonlinear state: 0 preparation + stage, 1 preparation only, 2 stage only
    double *nlQv = nullptr, *nlQe = nullptr, *nlKe = nullptr;
    int feFast = -1;                            // moka_last_fe_path
    double *nlZv = nullptr, *nlDiv = nullptr;   // Del2 mixing (moka_set_viscosity_del2)
    double viscDel2 = 0.0;
    std::vector<void *> allocs;
    // objects that hold or have exported the addresses of this state's arrays (halos, tapes): while any exists the arrays stay
    // where they are (moka_state_optimize_placement refuses)
    int attached = 0;
    std::vector<moka_placement_trial> placementLog;   // what the last moka_state_optimize_placement tried
    int64_t placementLaunches = 0;                    // ... and how many stage launches it issued (measurement bookkeeping)
};

namespace mk {

using namespace moka;

int fail(moka_ctx *ctx, int code, const std::string &msg);
// Halos and tapes count themselves in moka_state.attached while they hold the state's addresses.  Handles may be destroyed in
// any order (garbage-collected callers: a state can go before its tape), so the count is only touched while the state is alive.
void state_attach(moka_state *st);
void state_detach(moka_state *st);

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return mk::fail(ctx, MOKA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

// every host->device copy of the library: on the context's stream, then synchronised (see api.hip)
int h2d(moka_ctx *ctx, void *dst, const void *src, size_t bytes);
int alloc_field(moka_state *st, double **out, size_t elems, size_t elemBytes = sizeof(double));
int ensure_rk_bufs(moka_state *st);
int ensure_spare(moka_state *st);
// the set a Forward-Euler step writes its new level to / the rotation that makes it the current level
LevelBufs &fe_new_level(moka_state *st);
void fe_rotate_levels(moka_state *st);
int flush_lazy(moka_state *st, bool diag, bool tend);
// one fused tendency / RK-stage launch over patches [pBegin, pBegin + pCount) (default: all) on the compute stream (or `on`)
hipError_t run_stage(moka_state *st, const StageArgs &g, int pBegin = 0, int pCount = -1, hipStream_t on = nullptr, int tail = -1);
StageArgs rk4_stage_args(moka_state *st, int s, double dt, const double *ssh0);
LevelBufs &rk4_stage_output(moka_state *st, int s);
int rk4_begin(moka_state *st, const double **ssh0);
void rk4_end(moka_state *st);
bool rk13_usable(const moka_state *st);       // the 13-stream RK4 form (moka_set_tuning key 7), see api.hip
StageArgs rk13_stage_args(moka_state *st, int s, double dt, const double *ssh0);
void rk13_end(moka_state *st);
FeArgs fe_args(moka_state *st, int ops, int flags, double dt);
StageArgs fe_stage_args(moka_state *st, const FeArgs &a, int flags);
// Forward-Euler step in the stage kernels: is that path open to this state / these flags; will the step be lean (see
// moka_state.feLazy); what a step has to do about lazily pending arrays before its launches (idempotent within a step)
bool fe_stage_path(const moka_state *st, int flags);
bool fe_lean(const moka_state *st, int flags);
int fe_begin(moka_state *st, int flags);
void fe_end(moka_state *st, int flags, bool stageKernel, bool lean, bool prevMode);

}  // namespace mk
