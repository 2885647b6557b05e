// api.hip -- C ABI of libmoka_hip.so (include/moka_hip.h): context, device mesh, state, steps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include <mutex>
#include <unordered_set>

#include "state.hpp"

namespace moka { void fill_mesh_info(const Plan &p, moka_mesh_info *info); }

namespace mk {

static std::atomic<bool> g_rk13{false};      // moka_set_tuning key 7: RK4 steps in the 13-stream form where mk::rk13_usable
static std::mutex g_liveMutex;
static std::unordered_set<const moka_state *> g_liveStates;
void state_attach(moka_state *st)
{
    std::lock_guard<std::mutex> lk(g_liveMutex);
    if (g_liveStates.count(st)) ++st->attached;
}
void state_detach(moka_state *st)
{
    std::lock_guard<std::mutex> lk(g_liveMutex);
    if (g_liveStates.count(st)) --st->attached;
}

int fail(moka_ctx *ctx, int code, const std::string &msg)
{
    set_error(msg);
    if (ctx) ctx->err = msg;
    return code;
}

int lanes_per_column(int K)
{
    int l = 1;
    while (l < K && l < 64) l <<= 1;
    return l;
}

// Rule of this file: no null-stream hipMemcpy / hipMemset after context creation.  The context's streams are non-blocking,
// i.e. NOT ordered against the null stream, so a null-stream copy can overtake (or be overtaken by) a hipMemsetAsync queued
// on the context stream (the tape-list race of round 1).  Every host->device copy goes through h2d(): on the context's
// stream, then synchronised (the source is usually a temporary).  tools/check_streams.sh greps for offenders.
int h2d(moka_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return MOKA_OK;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MOKA_OK;
}

template <class T>
int upload_vec(moka_mesh *m, const std::vector<T> &v, const T **out)
{
    void *d = nullptr;
    const size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIPCHK(m->ctx, hipMalloc(&d, bytes));
    m->allocs.push_back(d);
    if (int rc = h2d(m->ctx, d, v.data(), v.size() * sizeof(T))) return rc;
    *out = static_cast<const T *>(d);
    return MOKA_OK;
}

int ensure_op_bufs(moka_mesh *m)
{
    const Plan &p = m->plan;
    const size_t need = (size_t)p.K * std::max(p.nE, std::max(p.nC, p.nV));
    if (m->opBufElems >= need) return MOKA_OK;
    for (auto &b : m->opBuf) {
        if (b) HIPCHK(m->ctx, hipFree(b));
        b = nullptr;
    }
    for (auto &b : m->opBuf) HIPCHK(m->ctx, hipMalloc((void **)&b, need * sizeof(double)));
    m->opBufElems = need;
    return MOKA_OK;
}

int alloc_field(moka_state *st, double **out, size_t elems, size_t elemBytes)
{
    void *d = nullptr;
    HIPCHK(st->ctx, hipMalloc(&d, std::max<size_t>(elems * elemBytes, 16)));
    st->allocs.push_back(d);
    HIPCHK(st->ctx, hipMemsetAsync(d, 0, elems * elemBytes, st->ctx->stream));   // KA.zeros
    *out = static_cast<double *>(d);
    return MOKA_OK;
}

int ensure_rk_bufs(moka_state *st)
{
    if (st->rk[1].ssh) return MOKA_OK;       // the last of the six: set only when all of them exist
    const Plan &p = st->mesh->plan;
    LevelBufs tmp[2];
    const size_t mark = st->allocs.size();
    const size_t sb = st->f32 ? 4 : 8;
    int rc = MOKA_OK;
    for (auto &r : tmp) {
        if (rc == MOKA_OK) rc = alloc_field(st, &r.u, (size_t)p.K * p.nE, sb);
        if (rc == MOKA_OK) rc = alloc_field(st, &r.h, (size_t)p.K * p.nC, sb);
        if (rc == MOKA_OK) rc = alloc_field(st, &r.ssh, (size_t)p.nC, sb);
    }
    if (rc != MOKA_OK) {                     // free what a partial failure left behind: a later call starts over
        (void)hipStreamSynchronize(st->ctx->stream);
        while (st->allocs.size() > mark) { (void)hipFree(st->allocs.back()); st->allocs.pop_back(); }
        return rc;
    }
    st->rk[0] = tmp[0]; st->rk[1] = tmp[1];
    st->phys[2] = tmp[0]; st->phys[3] = tmp[1];
    return MOKA_OK;
}

int ensure_spare(moka_state *st)
{
    if (st->spare.ssh) return MOKA_OK;
    const Plan &p = st->mesh->plan;
    LevelBufs tmp;
    const size_t mark = st->allocs.size();
    const size_t sb = st->f32 ? 4 : 8;
    int rc = alloc_field(st, &tmp.u, (size_t)p.K * p.nE, sb);
    if (rc == MOKA_OK) rc = alloc_field(st, &tmp.h, (size_t)p.K * p.nC, sb);
    if (rc == MOKA_OK) rc = alloc_field(st, &tmp.ssh, (size_t)p.nC, sb);
    if (rc != MOKA_OK) {
        (void)hipStreamSynchronize(st->ctx->stream);
        while (st->allocs.size() > mark) { (void)hipFree(st->allocs.back()); st->allocs.pop_back(); }
        return rc;
    }
    st->spare = tmp;
    st->phys[4] = tmp;
    return MOKA_OK;
}

// Where a Forward-Euler step writes its new level: the spare set when it exists (the step may then read the previous level
// while it runs), else the previous level's buffers (which the new level replaces anyway).
LevelBufs &fe_new_level(moka_state *st) { return st->spare.ssh ? st->spare : st->lev[0]; }

// prev <- cur <- new (<- old prev becomes the spare).  Without a spare set: the plain swap.
void fe_rotate_levels(moka_state *st)
{
    if (st->spare.ssh) {
        const LevelBufs oldPrev = st->lev[0];
        st->lev[0] = st->lev[1];
        st->lev[1] = st->spare;
        st->spare = oldPrev;
    } else {
        std::swap(st->lev[0], st->lev[1]);
    }
}

struct FieldRef {
    double *ptr;
    int kind;     // MOKA_CELL / EDGE / VERTEX
    int64_t n;
    int K;
    bool f32 = false;   // ptr is a float array (prognostic field of an fp32-storage state)
};

int field_ref(moka_state *st, int field, int level, FieldRef *r)
{
    const Plan &p = st->mesh->plan;
    if (level != 0 && level != 1) return fail(st->ctx, MOKA_ERR_ARG, "time_level must be 0 (previous) or 1 (current)");
    switch (field) {
        case MOKA_F_SSH: *r = {st->lev[level].ssh, MOKA_CELL, p.nC, 1}; break;
        case MOKA_F_NORMAL_VELOCITY: *r = {st->lev[level].u, MOKA_EDGE, p.nE, p.K}; break;
        case MOKA_F_LAYER_THICKNESS: *r = {st->lev[level].h, MOKA_CELL, p.nC, p.K}; break;
        case MOKA_F_LAYER_THICKNESS_EDGE: *r = {st->hEdge[0], MOKA_EDGE, p.nE, p.K}; break;
        case MOKA_F_THICKNESS_FLUX: *r = {st->F, MOKA_EDGE, p.nE, p.K}; break;
        case MOKA_F_VELOCITY_DIV_CELL: *r = {st->div, MOKA_CELL, p.nC, p.K}; break;
        case MOKA_F_RELATIVE_VORTICITY: *r = {st->vort, MOKA_VERTEX, p.nV, p.K}; break;
        case MOKA_F_TEND_NORMAL_VELOCITY: *r = {st->tendU, MOKA_EDGE, p.nE, p.K}; break;
        case MOKA_F_TEND_LAYER_THICKNESS: *r = {st->tendH, MOKA_CELL, p.nC, p.K}; break;
        default: return fail(st->ctx, MOKA_ERR_ARG, "unknown field id");
    }
    r->f32 = st->f32;     // every field of an fp32-storage state is a float array
    if (!r->ptr) return fail(st->ctx, MOKA_ERR_ARG, "field not allocated");
    return MOKA_OK;
}

const int32_t *perm_of(const moka_mesh *m, int kind)
{
    return kind == MOKA_CELL ? m->dev.cellN2O : kind == MOKA_EDGE ? m->dev.edgeN2O : m->dev.vertN2O;
}

// host (caller numbering) -> device field (device numbering)
int put_rows(moka_mesh *m, double *dst, const double *host, int kind, int64_t n, int K, bool f32 = false)
{
    int rc = ensure_op_bufs(m);
    if (rc) return rc;
    hipStream_t s = m->ctx->stream;
    HIPCHK(m->ctx, hipMemcpyAsync(m->opBuf[2], host, (size_t)n * K * sizeof(double), hipMemcpyHostToDevice, s));
    if (f32) HIPCHK(m->ctx, launch_permute_rows_f32(dst, m->opBuf[2], perm_of(m, kind), n, K, 1, s));   // rounds to fp32
    else HIPCHK(m->ctx, launch_permute_rows(dst, m->opBuf[2], perm_of(m, kind), n, K, 1, s));
    HIPCHK(m->ctx, hipStreamSynchronize(s));
    return MOKA_OK;
}

int get_rows(moka_mesh *m, double *host, const double *src, int kind, int64_t n, int K, bool f32 = false)
{
    int rc = ensure_op_bufs(m);
    if (rc) return rc;
    hipStream_t s = m->ctx->stream;
    if (f32) HIPCHK(m->ctx, launch_permute_rows_f32(m->opBuf[2], src, perm_of(m, kind), n, K, 0, s));
    else HIPCHK(m->ctx, launch_permute_rows(m->opBuf[2], src, perm_of(m, kind), n, K, 0, s));
    HIPCHK(m->ctx, hipMemcpyAsync(host, m->opBuf[2], (size_t)n * K * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(m->ctx, hipStreamSynchronize(s));
    return MOKA_OK;
}

int nlev_of(const moka_state *st, int flags) { return (flags & MOKA_FE_LEVEL1_ONLY) ? 1 : st->mesh->plan.K; }

FeArgs fe_args(moka_state *st, int ops, int flags, double dt)
{
    FeArgs a{};
    a.ops = ops;
    a.flags = flags;
    a.nlev = nlev_of(st, flags);
    a.dt = dt;
    a.u = st->lev[1].u; a.h = st->lev[1].h; a.ssh = st->lev[1].ssh;
    a.hEdgeOld = st->hEdge[0]; a.hEdgeNew = st->hEdge[1];
    a.Fin = st->F; a.F = st->F; a.div = st->div; a.vort = st->vort;
    a.tendU = st->tendU; a.tendH = st->tendH;
    const LevelBufs &nw = fe_new_level(st);
    a.u_new = nw.u; a.h_new = nw.h; a.ssh_new = nw.ssh;
    return a;
}

// One fused tendency / RK-stage launch over patches [pBegin, pBegin + pCount) (default: all) on the compute stream (or `on`).
// variant 0 (auto): k_stage_rec2c, then rec2 / rec / col / generic as the mesh allows; fp32-storage and nonlinear states have
// their own kernels.
hipError_t run_stage(moka_state *st, const StageArgs &g_in, int pBegin, int pCount, hipStream_t on, int tail)
{
    const StageArgs &g = g_in;
    const moka_mesh *m = st->mesh;
    MeshDev dev = m->dev;                 // the launch covers patches [pBegin, pBegin + pCount) (+ the patch `tail`)
    dev.tailPatch = -1;
    hipStream_t s = on ? on : st->ctx->stream;
    if (tail >= 0) {
        // One extra, non-adjacent patch in the same launch: only the default kernels can carry it.  Anything else
        // (explicit variants, fallbacks for other K, the nonlinear path) gets a launch of its own for it.
        hipError_t e = hipErrorNotSupported;
        if (pCount > 0 && !st->nonlinear && (st->f32 || ((st->ctx->variant == 0 || st->ctx->variant == 11) && m->lpc == 64 && m->colOk))) {
            const moka::Plan &p = m->plan;
            dev.patchBegin = pBegin; dev.nPatches = pCount; dev.tailPatch = tail;
            int mE = p.patchEdgeStart[tail + 1] - p.patchEdgeStart[tail], mC = p.patchCellStart[tail + 1] - p.patchCellStart[tail];
            for (int q = pBegin; q < pBegin + pCount; ++q) {
                mE = std::max(mE, p.patchEdgeStart[q + 1] - p.patchEdgeStart[q]);
                mC = std::max(mC, p.patchCellStart[q + 1] - p.patchCellStart[q]);
            }
            dev.maxOwnE = mE; dev.maxOwnC = mC;
            e = st->f32 ? launch_stage_rec2c_f32(dev, g, s) : launch_stage_rec2c(dev, g, s);
        }
        if (e != hipErrorNotSupported) return e;
        e = run_stage(st, g_in, pBegin, pCount, on);
        return e != hipSuccess ? e : run_stage(st, g_in, tail, 1, on);
    }
    if (pCount >= 0) {
        dev.patchBegin = pBegin; dev.nPatches = pCount;
        if (pCount > 0) {
            auto &cache = st->mesh->rangeMax;
            auto it = cache.find({pBegin, pCount});
            if (it == cache.end()) {
                const moka::Plan &p = m->plan;
                int mE = 1, mC = 1;
                for (int q = pBegin; q < pBegin + pCount; ++q) {
                    mE = std::max(mE, p.patchEdgeStart[q + 1] - p.patchEdgeStart[q]);
                    mC = std::max(mC, p.patchCellStart[q + 1] - p.patchCellStart[q]);
                }
                it = cache.emplace(std::make_pair(pBegin, pCount), std::make_pair(mE, mC)).first;
            }
            dev.maxOwnE = it->second.first; dev.maxOwnC = it->second.second;
        }
    }
    if (dev.nPatches <= 0) return hipSuccess;
    if (st->nonlinear) {
        // vector-invariant form: potential vorticity at vertices -> edges, kinetic energy at cells, thickness flux at edges
        // (whole mesh: the stencil of the edge pass reaches two cells deep), then the generic stage kernel's nonlinear twin
        const bool del2 = st->viscDel2 != 0.0;
        const NlArgs nl{st->nlQv, st->nlQe, st->nlKe, del2 ? st->nlZv : nullptr, del2 ? st->nlDiv : nullptr, st->viscDel2};
        // kernel variants 4 / 3 select the plainer forms of the nonlinear kernels too (tests run every form against the oracle)
        const int form = st->ctx->variant == 4 ? 1 : st->ctx->variant == 3 ? 3 : 0;
        // a patch range (partitioned meshes: moka_rk4_dist_stage) is served by the patch forms only
        if (pCount >= 0 && !nl_patch_forms(dev, m->lpc, form)) return hipErrorNotSupported;
        if (st->nlPhase != 2) {
            hipError_t e = launch_nl_prepare(dev, g.pu, g.ph, nl, m->lpc, form, s);
            if (e != hipSuccess) return e;
        }
        if (st->nlPhase == 1) return hipSuccess;
        return launch_stage_nl(dev, g, nl, m->lpc, m->plan.ldsOk, form, s);
    }
    if (st->f32) {   // the one fp32-storage kernel (checked at state creation against the patches that are ever launched)
        if (pCount < 0 && m->plan.nPatchesLaunch < m->plan.nPatches) {
            // whole-mesh launch on a partitioned mesh: the halo-only patches are skipped (their rows arrive by exchange)
            dev.nPatches = m->plan.nPatchesLaunch;
            dev.maxOwnE = std::max(m->plan.maxOwnELaunch, 1); dev.maxOwnC = std::max(m->plan.maxOwnCLaunch, 1);
        }
        return launch_stage_rec2c_f32(dev, g, s);
    }
    const int v = st->ctx->variant;
    // 0 = auto: rec2c (even 34 <= K <= 64, patches whose records + own rows fit the LDS), else the plain column kernel (K >= 33),
    // else the generic index kernel.  11 rec2c, 4 column, 3 generic.  (The other execution shapes measured in rounds 1-3 -- pipelined
    // and 16-byte-lane column kernels, LDS-tiled and LDS-DMA forms, persistent double-buffered tiles -- lost and are gone; their
    // numbers are in profiles/r01_variants.txt ... r03_variants.txt.)
    if ((v == 0 || v == 11) && m->lpc == 64 && m->colOk) {   // default: 16-byte lanes + own-edge u rows cached in LDS
        hipError_t e = launch_stage_rec2c(dev, g, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (v != 3 && m->lpc == 64 && m->colOk) return launch_stage_col(dev, g, s);
    return launch_stage(dev, g, m->lpc, s);
}

// Forward-Euler step in the stage kernels (k_stage_rec2c* modes 4 / 5 / 6): all levels, the default kernel choice, a mesh the
// kernels can carry.  (MOKA_FE_LEVEL1_ONLY, odd or large K, explicit kernel variants take the generic one-launch kernel k_fe.)
bool fe_stage_path(const moka_state *st, int flags)
{
    const moka_mesh *mm = st->mesh;
    if (flags & MOKA_FE_LEVEL1_ONLY) return false;
    if (st->f32) return true;                                  // checked when the state was created
    MeshDev dev = mm->dev;
    dev.maxOwnE = std::max(mm->plan.maxOwnELaunch, 1); dev.maxOwnC = std::max(mm->plan.maxOwnCLaunch, 1);
    return (st->ctx->variant == 0 || st->ctx->variant == 11) && mm->lpc == 64 && mm->colOk && rec2c_supported(dev);
}

// A LEAN step stores the new level and relativeVorticity only (moka_state.feLazy).  It needs the stage-kernel path, the spare
// level set, and -- with the reference's stale flux thickness -- that thickness to be derivable from the previous level.
bool fe_lean(const moka_state *st, int flags)
{
    return moka::fe_lean_enabled() && !st->feForceEager && fe_stage_path(st, flags) && st->spare.ssh &&
           (!(flags & MOKA_FE_STALE_HEDGE) || (st->hEdgePrev && moka::fe_prev_mode()));
}

// What a Forward-Euler step does about lazily pending arrays before its first launch.  A lean step reads none of them and
// supersedes them -- except relativeVorticity when it accumulates onto a value an RK4 step left pending; anything else needs
// them in memory.  Idempotent: the parts of a distributed step all call it.
int fe_begin(moka_state *st, int flags)
{
    // an fp32-storage state after an RK4 step has no current DiagnosticVars: a step that carries none over (flags 0) may follow
    if (st->f32 && st->diagDirty && !(flags & (MOKA_FE_STALE_HEDGE | MOKA_FE_ACCUM_VORT))) st->diagDirty = false;
    if (fe_lean(st, flags) && !(st->diagDirty && (flags & MOKA_FE_ACCUM_VORT))) {
        st->diagDirty = st->tendDirty = false;
        st->lazyPu = st->lazyPh = nullptr; st->lazyOwner = nullptr;
        return MOKA_OK;
    }
    return flush_lazy(st, true, true);
}

void fe_end(moka_state *st, int flags, bool stageKernel, bool lean, bool prevMode)
{
    fe_rotate_levels(st);
    if (!lean) std::swap(st->hEdge[0], st->hEdge[1]);          // a lean step has not written layerThicknessEdge
    st->feFast = stageKernel ? (prevMode ? 2 : 1) : 0;
    st->feLazy = lean;
    st->feLazyCount = -1;              // (a distributed step with a direct halo narrows it: moka_fe_dist_end)
    st->feLazyStale = lean && (flags & MOKA_FE_STALE_HEDGE);
    // the stage kernel interpolated (or, lean, will interpolate) layerThicknessEdge of every computed edge from the level that is
    // the previous one now: the next step may form the reference's stale flux thickness from that level (mode 6)
    st->hEdgePrev = stageKernel;
}

// Argument block of a Forward-Euler launch of the stage kernels from the generic kernel's.  MOKA_FE_STALE_HEDGE: the stored
// layerThicknessEdge is gathered (mode 4) unless it is known to be the interpolation of the previous level's layerThickness and
// that level survives the step (the spare set takes the new level): then it is formed from those rows (mode 6).  A lean step
// passes no TendencyVars / DiagnosticVars outputs.
StageArgs fe_stage_args(moka_state *st, const FeArgs &a, int flags)
{
    StageArgs s{};
    s.pu = a.u; s.ph = a.h; s.ssh = a.ssh;
    s.pu_out = a.u_new; s.ph_out = a.h_new; s.ssh_out = a.ssh_new;
    s.a = a.dt;
    const bool stale = flags & MOKA_FE_STALE_HEDGE;
    const bool prev = stale && st->hEdgePrev && st->spare.ssh && a.h_new != st->lev[0].h && moka::fe_prev_mode();
    s.hEdgeOld = stale && !prev ? a.hEdgeOld : nullptr;
    s.hPrev = prev ? st->lev[0].h : nullptr;
    s.feMode = prev ? 6 : stale ? 4 : 5;
    s.areaCell = st->mesh->dev.areaCell;
    if (!fe_lean(st, flags)) {
        s.tendU = a.tendU; s.tendH = a.tendH;
        s.hEdgeNew = a.hEdgeNew; s.F = a.F; s.div = a.div;
    }
    // relativeVorticity by the same launch (the vertices of the launched patches) where the stage kernels can carry it
    {
        MeshDev dev = st->mesh->dev;
        dev.maxOwnE = std::max(st->mesh->plan.maxOwnELaunch, 1); dev.maxOwnC = std::max(st->mesh->plan.maxOwnCLaunch, 1);
        if (moka::stage_curl_fits(dev, st->f32)) { s.vort = a.vort; s.accumVort = (flags & MOKA_FE_ACCUM_VORT) ? 1 : 0; }
    }
    return s;
}

// The arrays a lean Forward-Euler step left pending (moka_state.feLazy), produced now: the same launch over the same patches
// with the new-level outputs switched off and the diagnostic outputs on, from the level the step started from (the previous
// one now) and, for the stale flux thickness, the level before it (`spare`).
static int materialize_fe(moka_state *st)
{
    const moka_mesh *mm = st->mesh;
    StageArgs g{};
    g.pu = st->lev[0].u; g.ph = st->lev[0].h; g.ssh = st->lev[0].ssh;
    g.feMode = st->feLazyStale ? 6 : 5;
    g.hPrev = st->feLazyStale ? st->spare.h : nullptr;
    g.tendU = st->tendU; g.tendH = st->tendH; g.F = st->F; g.div = st->div;
    g.hEdgeNew = st->hEdge[0];                                  // mode 6 reads no stored layerThicknessEdge: written in place
    g.areaCell = mm->dev.areaCell;
    MeshDev dev = mm->dev;
    dev.tailPatch = -1;
    dev.patchBegin = 0; dev.nPatches = mm->plan.nPatchesLaunch;
    if (st->feLazyCount >= 0) {                                 // a distributed step with a direct halo: the interior patches only
        dev.patchBegin = st->feLazyBegin; dev.nPatches = st->feLazyCount;
    }
    dev.maxOwnE = std::max(mm->plan.maxOwnELaunch, 1); dev.maxOwnC = std::max(mm->plan.maxOwnCLaunch, 1);
    if (dev.nPatches > 0)
        HIPCHK(st->ctx, st->f32 ? launch_stage_rec2c_f32(dev, g, st->ctx->stream) : launch_stage_rec2c(dev, g, st->ctx->stream));
    st->feLazy = false;
    st->feLazyCount = -1;
    return MOKA_OK;
}

int flush_lazy(moka_state *st, bool diag, bool tend)
{
    if ((diag || tend) && st->feLazy)
        if (int rc = materialize_fe(st)) return rc;
    if (diag && st->diagDirty && st->f32)   // no diagnostics-only kernel for float arrays: they come out of Forward-Euler steps
        return fail(st->ctx, MOKA_ERR_UNSUPPORTED,
                    "fp32-storage state: DiagnosticVars exist after Forward-Euler steps only (not after RK4 steps or uploads)");
    if (diag && st->diagDirty) {
        // clean diagnostics of the current state: hEdge = interp(h); F = u*hEdge; div; vort zeroed + curl
        FeArgs a = fe_args(st, FE_FLUX | FE_DIV | FE_CURL | FE_HEDGE, 0, 0.0);
        a.nlev = st->mesh->plan.K;
        HIPCHK(st->ctx, launch_fe(st->mesh->dev, a, st->mesh->lpc, st->ctx->stream));
        std::swap(st->hEdge[0], st->hEdge[1]);
        st->diagDirty = false;
        st->hEdgePrev = false;            // layerThicknessEdge now belongs to the CURRENT level
    }
    if (tend && st->tendDirty) {
        StageArgs g{};
        g.pu = st->lazyPu ? st->lazyPu : st->rk[0].u; g.ph = st->lazyPh ? st->lazyPh : st->rk[0].h; g.ssh = st->rk[0].ssh;
        g.tendU = st->tendU; g.tendH = st->tendH;
        HIPCHK(st->ctx, run_stage(st, g));
        st->tendDirty = false;
        st->lazyPu = st->lazyPh = nullptr; st->lazyOwner = nullptr;
    }
    return MOKA_OK;
}

bool is_diag_field(int f) { return f >= MOKA_F_LAYER_THICKNESS_EDGE && f <= MOKA_F_RELATIVE_VORTICITY; }
bool is_tend_field(int f) { return f == MOKA_F_TEND_NORMAL_VELOCITY || f == MOKA_F_TEND_LAYER_THICKNESS; }

}  // namespace mk

using namespace mk;

extern "C" {

const char *moka_version(void) { return "moka-hip 0.1.0 (gfx950)"; }

const char *moka_last_error(const moka_ctx *ctx) { return ctx ? ctx->err.c_str() : moka::get_error(); }

int moka_ctx_create(int device, moka_ctx **out)
{
    if (!out) return fail(nullptr, MOKA_ERR_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, MOKA_ERR_NO_DEVICE,
                    std::string("no HIP device available (libmoka_hip has no CPU fallback): ") +
                        (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= ndev) return fail(nullptr, MOKA_ERR_ARG, "device index out of range");
    HIPCHK(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MOKA_ERR_NO_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code objects only");
    moka_ctx *c = new (std::nothrow) moka_ctx();
    if (!c) return fail(nullptr, MOKA_ERR_ALLOC, "out of host memory");
    c->device = device;
    c->nCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e1 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    hipError_t e2 = hipEventCreate(&c->ev0);
    hipError_t e3 = hipEventCreate(&c->ev1);
    if (e1 == hipSuccess) {
        // the comm stream carries the boundary patches, pack / unpack and the halo transport: highest priority, so that its
        // few workgroups are dispatched ahead of the thousands the interior launch has queued
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e1 = hipStreamCreateWithPriority(&c->comm, hipStreamNonBlocking, hi);
    }
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->evBoundary, hipEventDisableTiming);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->evInterior, hipEventDisableTiming);
    if (e3 == hipSuccess) e3 = hipEventCreateWithFlags(&c->evHalo, hipEventDisableTiming);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        moka_ctx_destroy(c);                  // destroys whatever was created
        return fail(nullptr, MOKA_ERR_HIP, "failed to create stream/events");
    }
    *out = c;
    return MOKA_OK;
}

void moka_ctx_destroy(moka_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) { (void)hipStreamSynchronize(ctx->comm); (void)hipStreamDestroy(ctx->comm); }
    for (hipEvent_t e : {ctx->evBoundary, ctx->evInterior, ctx->evHalo}) if (e) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->evPool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->marks) (void)hipEventDestroy(e);
    if (ctx->bwBuf) (void)hipFree(ctx->bwBuf);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int moka_sync(moka_ctx *ctx)
{
    if (!ctx) return fail(nullptr, MOKA_ERR_ARG, "ctx is NULL");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->comm));
    return MOKA_OK;
}

int moka_ctx_streams(moka_ctx *ctx, void **compute, void **comm)
{
    if (!ctx) return fail(nullptr, MOKA_ERR_ARG, "ctx is NULL");
    if (compute) *compute = (void *)ctx->stream;
    if (comm) *comm = (void *)ctx->comm;
    return MOKA_OK;
}

int moka_timer_start(moka_ctx *ctx)
{
    if (!ctx) return fail(nullptr, MOKA_ERR_ARG, "ctx is NULL");
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return MOKA_OK;
}

int moka_timer_stop(moka_ctx *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return fail(ctx, MOKA_ERR_ARG, "NULL argument");
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return MOKA_OK;
}

// PCI bus id ("0000:c5:00.0") of the context's device: lets a caller find the device's clock / power files in sysfs
// (/sys/bus/pci/devices/<id>/pp_dpm_sclk ...) and count distinct devices among the ranks of a launch.
int moka_ctx_pci_bus_id(moka_ctx *ctx, char *buf, int32_t len)
{
    if (!ctx || !buf || len < 16) return fail(ctx, MOKA_ERR_ARG, "buf is NULL or shorter than 16 bytes");
    HIPCHK(ctx, hipDeviceGetPCIBusId(buf, len, ctx->device));
    return MOKA_OK;
}

// Per-step statistics of a timed region (bench.py: median / min / max of the step times).  moka_mark records one event on the
// compute stream; n marks give n - 1 intervals.  moka_marks_reset forgets them (the events are kept for reuse).
int moka_mark(moka_ctx *ctx)
{
    if (!ctx) return fail(nullptr, MOKA_ERR_ARG, "ctx is NULL");
    if (ctx->marksUsed == ctx->marks.size()) {
        hipEvent_t e = nullptr;
        HIPCHK(ctx, hipEventCreate(&e));
        ctx->marks.push_back(e);
    }
    HIPCHK(ctx, hipEventRecord(ctx->marks[ctx->marksUsed++], ctx->stream));
    return MOKA_OK;
}

int moka_marks_reset(moka_ctx *ctx)
{
    if (!ctx) return fail(nullptr, MOKA_ERR_ARG, "ctx is NULL");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->marksUsed = 0;
    return MOKA_OK;
}

// ms[i] = time between mark i and mark i + 1 (at most `capacity` intervals); *n = number of intervals available
int moka_marks_read(moka_ctx *ctx, int64_t capacity, double *ms, int64_t *n)
{
    if (!ctx || !n || (capacity > 0 && !ms)) return fail(ctx, MOKA_ERR_ARG, "NULL argument");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t have = ctx->marksUsed > 0 ? (int64_t)ctx->marksUsed - 1 : 0;
    *n = have;
    for (int64_t i = 0; i < have && i < capacity; ++i) {
        float t = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&t, ctx->marks[i], ctx->marks[i + 1]));
        ms[i] = t;
    }
    return MOKA_OK;
}

// Same-run bandwidth calibration: what this device, in the state it is in NOW, moves with a plain 16-byte-per-lane copy and a
// read-only sweep (MI355X_MICROARCH.md quotes 6.29 TB/s for the copy).  `bytes` = total footprint (two halves: source and
// destination); every launch is timed on its own with HIP events on the compute stream, the best of `iters` is reported
// (the first launch also faults the pages in).  gbs[0] = copy, bytes read + written per second; gbs[1] = read-only sweep of both
// halves; gbs[2] = mean copy rate over the launches; gbs[3] = gather of 480-byte rows in a scattered order, a half-wave per
// row (the access pattern of the stage kernels at 60 layers).  bytes = 0 releases the buffers.
int moka_bw_probe(moka_ctx *ctx, int64_t bytes, int iters, double gbs[4])
{
    if (!ctx) return fail(nullptr, MOKA_ERR_ARG, "ctx is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (bytes <= 0) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->bwBuf) HIPCHK(ctx, hipFree(ctx->bwBuf));
        ctx->bwBuf = nullptr; ctx->bwBytes = 0;
        return MOKA_OK;
    }
    if (!gbs || iters < 1) return fail(ctx, MOKA_ERR_ARG, "gbs is NULL or iters < 1");
    const size_t half = ((size_t)bytes / 2) & ~(size_t)4095;
    if (half < ((size_t)1 << 20)) return fail(ctx, MOKA_ERR_ARG, "bandwidth probe: at least 2 MiB");
    hipStream_t s = ctx->stream;
    if (ctx->bwBytes != 2 * half) {
        if (ctx->bwBuf) { HIPCHK(ctx, hipStreamSynchronize(s)); HIPCHK(ctx, hipFree(ctx->bwBuf)); ctx->bwBuf = nullptr; ctx->bwBytes = 0; }
        hipError_t e = hipMalloc(&ctx->bwBuf, 2 * half + 64);
        if (e != hipSuccess) return fail(ctx, MOKA_ERR_ALLOC, std::string("bandwidth probe: hipMalloc: ") + hipGetErrorString(e));
        ctx->bwBytes = 2 * half;
        HIPCHK(ctx, hipMemsetAsync(ctx->bwBuf, 0x5A, 2 * half + 64, s));
    }
    unsigned char *a = static_cast<unsigned char *>(ctx->bwBuf), *b = a + half;
    uint32_t *sink = reinterpret_cast<uint32_t *>(a + 2 * half);
    double bestCopy = 0.0, bestRead = 0.0, sumCopy = 0.0, bestGather = 0.0;
    const uint32_t rowB = 480;
    for (int pass = 0; pass < 3; ++pass)
        for (int i = 0; i < iters; ++i) {
            HIPCHK(ctx, hipEventRecord(ctx->ev0, s));
            if (pass == 0) HIPCHK(ctx, launch_bw_copy(b, a, (int64_t)half, ctx->nCUs, s));
            else if (pass == 1) HIPCHK(ctx, launch_bw_read(a, (int64_t)(2 * half), sink, ctx->nCUs, s));
            else HIPCHK(ctx, launch_bw_gather(a, (int64_t)(2 * half), rowB, sink, ctx->nCUs, s));
            HIPCHK(ctx, hipEventRecord(ctx->ev1, s));
            HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
            float ms = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            const double r = (double)(2 * half) / ((double)ms * 1e-3) / 1e9;
            if (pass == 0) { bestCopy = std::max(bestCopy, r); sumCopy += r; }
            else if (pass == 1) bestRead = std::max(bestRead, r);
            else bestGather = std::max(bestGather, (double)((2 * half) / rowB * rowB) / ((double)ms * 1e-3) / 1e9);
        }
    gbs[0] = bestCopy; gbs[1] = bestRead; gbs[2] = sumCopy / iters; gbs[3] = bestGather;
    return MOKA_OK;
}

// A fifth figure for the same calibration: three streams read and two written at once (k_bw_streams) over the buffer moka_bw_probe
// allocated (call that first); GB/s of all five, best of `iters` launches.
int moka_bw_probe_streams(moka_ctx *ctx, int iters, double *gbs)
{
    if (!ctx || !gbs || iters < 1) return fail(ctx, MOKA_ERR_ARG, "NULL argument or iters < 1");
    if (!ctx->bwBuf || ctx->bwBytes < ((size_t)5 << 20)) return fail(ctx, MOKA_ERR_ARG, "moka_bw_probe first: it owns the buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int64_t n = (int64_t)(ctx->bwBytes / 5 / 16);
    double best = 0.0;
    for (int i = 0; i < iters; ++i) {
        HIPCHK(ctx, hipEventRecord(ctx->ev0, s));
        HIPCHK(ctx, launch_bw_streams(ctx->bwBuf, (int64_t)ctx->bwBytes, s));
        HIPCHK(ctx, hipEventRecord(ctx->ev1, s));
        HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        best = std::max(best, (double)(5 * 16 * n) / ((double)ms * 1e-3) / 1e9);
    }
    // the probes XOR the pattern away: restore it, the read probe's sentinel test relies on it
    HIPCHK(ctx, hipMemsetAsync(ctx->bwBuf, 0x5A, ctx->bwBytes + 64, s));
    *gbs = best;
    return MOKA_OK;
}

// A sixth figure: `region` bytes (more than the 32 MB of L2s, less than the 256 MB Infinity Cache: 128 MiB) read `reps` times back
// to back, GB/s over all passes, best of `iters` -- the rate at which re-read data comes back.  The stage kernels fetch a fifth
// to a quarter of their bytes a second time (rows of neighbouring patches, profiles/r03_traffic_attribution.txt); the streaming
// probes above never read anything twice.
int moka_bw_probe_reread(moka_ctx *ctx, int64_t region, int reps, int iters, double *gbs)
{
    if (!ctx || !gbs || iters < 1 || reps < 1) return fail(ctx, MOKA_ERR_ARG, "NULL argument or iters / reps < 1");
    if (!ctx->bwBuf || region < ((int64_t)1 << 20) || (size_t)region > ctx->bwBytes) return fail(ctx, MOKA_ERR_ARG, "moka_bw_probe first: it owns the buffer (region must fit it)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    unsigned char *a = static_cast<unsigned char *>(ctx->bwBuf);
    uint32_t *sink = reinterpret_cast<uint32_t *>(a + ctx->bwBytes);
    region &= ~(int64_t)4095;
    double best = 0.0;
    HIPCHK(ctx, launch_bw_read(a, region, sink, ctx->nCUs, s));          // first touch: from HBM
    for (int i = 0; i < iters; ++i) {
        HIPCHK(ctx, hipEventRecord(ctx->ev0, s));
        for (int r = 0; r < reps; ++r) HIPCHK(ctx, launch_bw_read(a, region, sink, ctx->nCUs, s));
        HIPCHK(ctx, hipEventRecord(ctx->ev1, s));
        HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        best = std::max(best, (double)region * reps / ((double)ms * 1e-3) / 1e9);
    }
    *gbs = best;
    return MOKA_OK;
}

// A seventh: the scattered row gather over a LARGE footprint of its own (`bytes`, e.g. 32 GiB: what a state with all its arrays
// spans), a quarter of the rows fetched once each -- consecutive half-waves land ~19 MB apart, so nearly every fetch needs an
// address translation the TLBs do not hold.  The 4 GiB gather of moka_bw_probe fits them.
int moka_bw_probe_gather_big(moka_ctx *ctx, int64_t bytes, int iters, double *gbs)
{
    if (!ctx || !gbs || iters < 1 || bytes < ((int64_t)1 << 24)) return fail(ctx, MOKA_ERR_ARG, "NULL argument, iters < 1 or less than 16 MiB");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    void *buf = nullptr;
    hipError_t e = hipMalloc(&buf, (size_t)bytes + 64);
    if (e != hipSuccess) return fail(ctx, MOKA_ERR_ALLOC, std::string("big gather probe: hipMalloc: ") + hipGetErrorString(e));
    const uint32_t rowB = 480;
    const int64_t nRows = bytes / rowB, nFetch = nRows / 4;
    uint32_t *sink = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(buf) + bytes);
    double best = 0.0;
    hipError_t rc = hipMemsetAsync(buf, 0x5A, (size_t)bytes + 64, s);
    for (int i = 0; i < iters && rc == hipSuccess; ++i) {
        if ((rc = hipEventRecord(ctx->ev0, s)) != hipSuccess) break;
        if ((rc = launch_bw_gather_n(buf, nRows, rowB, nFetch, sink, ctx->nCUs, s)) != hipSuccess) break;
        if ((rc = hipEventRecord(ctx->ev1, s)) != hipSuccess) break;
        if ((rc = hipEventSynchronize(ctx->ev1)) != hipSuccess) break;
        float ms = 0.f;
        if ((rc = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1)) != hipSuccess) break;
        best = std::max(best, (double)nFetch * rowB / ((double)ms * 1e-3) / 1e9);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(buf);
    HIPCHK(ctx, rc);
    *gbs = best;
    return MOKA_OK;
}

// Process-wide launch-shape switches for A/B measurements (results are identical for every setting):
//   key 1: bit mask of the modes of the fp32-storage stage kernel that run as 512-thread workgroups bounded to 128 registers
//          (default: mode 0; see kernels.hip, g_f32WideModes)
//   key 2: 0 = Forward-Euler steps always gather the stored layerThicknessEdge (stage-kernel mode 4), 1 (default) = they form
//          it from the previous level's layerThickness whenever that is the same thing (mode 6)
//   key 4: 0 = every Forward-Euler step stores all of its DiagnosticVars / TendencyVars; 1 (default) = lean steps where possible
//          (new level and relativeVorticity stored, the rest produced on first read: moka_state.feLazy)
//   key 5: launch shape of the nonlinear stage kernel's patch form (nonlinear.hip, g_nlShape): 0 (default) vertex rows in LDS, three
//          512-thread workgroups per CU; 1 = round 2's edge rows in LDS, two workgroups; 2 / 3 = other shapes of the vertex-row form
//   key 6: test hook, upper limit of the vertex rows that form keeps in LDS (0 = none)
//   key 3: 0 = the relativeVorticity pass of a Forward-Euler step always gets a launch of its own, 1 (default) = it rides in the
//          stage-kernel launches where they can carry it
//   key 8: bit mask of the modes of the Float64 stage kernel whose large launches take TWO consecutive patches per 512-thread workgroup
//          (default: 0, 1 and the 13-stream 7; kernels.hip, launch_stage_rec2c)
//   key 9: 1 (default) = lean Forward-Euler launches run the kernels' lean instances (modes 10 / 11: optional outputs compiled out),
//          0 = the general Forward-Euler instances (outputs tested at run time)
//   key 7: NOT result-neutral, opt-in (default 0): moka_step_rk4 / moka_run of Float64 states on whole meshes in the 13-stream form
//          (mk::rk13_usable; New formed in stage 4 from the provisional states instead of accumulated through the stages)
int moka_set_tuning(int key, int value)
{
    if (key == 1) { moka::set_f32_wide_modes(value); return MOKA_OK; }
    if (key == 2) { moka::set_fe_prev_mode(value); return MOKA_OK; }
    if (key == 9) { moka::set_fe_lean_instances(value); return MOKA_OK; }
    if (key == 3) { moka::set_curl_fused(value); return MOKA_OK; }
    if (key == 4) { moka::set_fe_lean(value); return MOKA_OK; }
    if (key == 5) { moka::set_nl_shape(value); return MOKA_OK; }
    if (key == 6) { moka::set_nl_cap_limit(value); return MOKA_OK; }
    if (key == 7) { g_rk13.store(value != 0); return MOKA_OK; }
    if (key == 8) { moka::set_pair_modes(value); return MOKA_OK; }
    return fail(nullptr, MOKA_ERR_ARG, "unknown tuning key");
}

int moka_get_tuning(int key, int *value)
{
    if (!value) return fail(nullptr, MOKA_ERR_ARG, "value is NULL");
    if (key == 1) { *value = moka::f32_wide_modes(); return MOKA_OK; }
    if (key == 2) { *value = moka::fe_prev_mode(); return MOKA_OK; }
    if (key == 9) { *value = moka::fe_lean_instances(); return MOKA_OK; }
    if (key == 3) { *value = moka::curl_fused(); return MOKA_OK; }
    if (key == 4) { *value = moka::fe_lean_enabled(); return MOKA_OK; }
    if (key == 5) { *value = moka::nl_shape(); return MOKA_OK; }
    if (key == 6) { *value = moka::nl_cap_limit(); return MOKA_OK; }
    if (key == 7) { *value = g_rk13.load() ? 1 : 0; return MOKA_OK; }
    if (key == 8) { *value = moka::pair_modes(); return MOKA_OK; }
    return fail(nullptr, MOKA_ERR_ARG, "unknown tuning key");
}

int moka_kernel_variant_available(int variant)
{
    return variant == 0 || variant == 3 || variant == 4 || variant == 11;
}

int moka_set_kernel_variant(moka_ctx *ctx, int variant)
{
    if (!ctx) return fail(nullptr, MOKA_ERR_ARG, "ctx is NULL");
    if (variant < 0 || variant > 14) return fail(ctx, MOKA_ERR_ARG, "variant must be 0..14");
    if (!moka_kernel_variant_available(variant))
        return fail(ctx, MOKA_ERR_UNSUPPORTED, "this kernel variant was an experiment of rounds 1-3 and no longer exists (0, 3, 4, 11 do)");
    ctx->variant = variant;
    return MOKA_OK;
}

// ---------------------------------------------------------------------------------------------
// mesh
// ---------------------------------------------------------------------------------------------
int moka_mesh_create(moka_ctx *ctx, const moka_mesh_desc *desc, moka_mesh **out)
{
    if (!ctx || !out) return fail(ctx, MOKA_ERR_ARG, "NULL argument");
    *out = nullptr;
    moka_mesh *m = new (std::nothrow) moka_mesh();
    if (!m) return fail(ctx, MOKA_ERR_ALLOC, "out of host memory");
    m->ctx = ctx;
    int rc;
    try {
        rc = build_plan(desc, m->plan);
    } catch (const std::bad_alloc &) {
        rc = fail(ctx, MOKA_ERR_ALLOC, "out of host memory while building the mesh plan");
    }
    if (rc != MOKA_OK) {
        ctx->err = moka::get_error();
        delete m;
        return rc;
    }
    const Plan &p = m->plan;
    if (p.ME > 8 || p.ME2 > 14) {
        delete m;
        return fail(ctx, MOKA_ERR_UNSUPPORTED, "cells with more than 8 edges are not supported by the gfx950 kernels");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    MeshDev &d = m->dev;
    d.nC = p.nC; d.nE = p.nE; d.nV = p.nV; d.K = p.K; d.ME = p.ME; d.ME2 = p.ME2; d.VD = p.VD; d.nPatches = p.nPatches;
    d.tailPatch = -1;      // no extra patch rides in a launch of this view (0, the zero-initialised value, would name patch 0: a
                           // whole-mesh Forward-Euler launch then ran patch 0 twice -- harmless while every result was a pure
                           // function of the inputs, wrong once relativeVorticity accumulates in the same launch)
    m->lpc = lanes_per_column(p.K);
#define UP(field)                                                \
    if ((rc = upload_vec(m, p.field, &d.field)) != MOKA_OK) {    \
        moka_mesh_destroy(m);                                    \
        return rc;                                               \
    }
    UP(patchCellStart) UP(patchEdgeStart) UP(patchVertStart)
    UP(eoc) UP(coc) UP(mltc) UP(sdv) UP(invArea) UP(areaCell) UP(rsum)
    UP(ehdr) UP(eoe) UP(woe) UP(gInvDc) UP(dcEdge) UP(dvEdge) UP(fEdge)
    UP(eov) UP(cv) UP(cellN2O) UP(edgeN2O) UP(vertN2O)
    UP(leoe) UP(cRec) UP(eRec) UP(feoe) UP(vRec) UP(rowStart) UP(rowEdge)
    if (p.nlOk) { UP(voe) UP(cov) UP(kite) UP(invAreaTri) UP(fVertex) UP(keCoef) UP(invDc) UP(keoc) UP(rowVoe) }
    if (p.nl5Ok) { UP(pvStart) UP(pvList) UP(lvoe) }
#undef UP
    d.maxPV = p.nl5Ok ? p.maxPV : 0;
    if (!p.nl5Ok) { d.pvStart = nullptr; d.pvList = nullptr; d.lvoe = nullptr; }
    d.CI = p.CI; d.EI = p.EI;
    m->colOk = p.colOk;

    d.maxRows = p.maxRows; d.maxOwnE = p.maxOwnE; d.maxOwnC = p.maxOwnC;
    d.maxOwnV = p.maxOwnV;
    if (p.vRec.empty()) d.vRec = nullptr;       // (upload_vec hands out a dummy allocation for an empty vector)
    *out = m;
    return MOKA_OK;
}

void moka_mesh_destroy(moka_mesh *mesh)
{
    if (!mesh) return;
    (void)hipSetDevice(mesh->ctx->device);
    (void)hipStreamSynchronize(mesh->ctx->stream);
    for (void *q : mesh->allocs) (void)hipFree(q);
    for (auto b : mesh->opBuf)
        if (b) (void)hipFree(b);
    delete mesh;
}

int moka_mesh_permutation(const moka_mesh *mesh, int kind, int32_t *new_to_old)
{
    if (!mesh || !new_to_old) return fail(nullptr, MOKA_ERR_ARG, "NULL argument");
    const Plan &p = mesh->plan;
    const std::vector<int32_t> *v = kind == MOKA_CELL ? &p.cellN2O : kind == MOKA_EDGE ? &p.edgeN2O : kind == MOKA_VERTEX ? &p.vertN2O : nullptr;
    if (!v) return fail(mesh->ctx, MOKA_ERR_ARG, "kind must be MOKA_CELL, MOKA_EDGE or MOKA_VERTEX");
    std::copy(v->begin(), v->end(), new_to_old);
    return MOKA_OK;
}

int moka_mesh_class_ranges(const moka_mesh *mesh, int32_t capacity, int32_t *nClasses, int32_t *patchStart, int32_t *cellStart,
                           int32_t *edgeStart)
{
    if (!mesh || !nClasses) return fail(nullptr, MOKA_ERR_ARG, "NULL argument");
    const Plan &p = mesh->plan;
    const int n = (int)p.classPatchStart.size() - 1;
    *nClasses = n;
    for (int k = 0; k <= n && k < capacity; ++k) {
        if (patchStart) patchStart[k] = p.classPatchStart[k];
        if (cellStart) cellStart[k] = p.classCellStart[k];
        if (edgeStart) edgeStart[k] = p.classEdgeStart[k];
    }
    return MOKA_OK;
}

int moka_mesh_info_get(const moka_mesh *mesh, moka_mesh_info *info)
{
    if (!mesh || !info) return fail(nullptr, MOKA_ERR_ARG, "NULL argument");
    fill_mesh_info(mesh->plan, info);
    return MOKA_OK;
}

// ---------------------------------------------------------------------------------------------
// operators on host arrays
// ---------------------------------------------------------------------------------------------
static int run_operator(moka_mesh *m, int op, int nlev, const double *in, int inKind, double *out, int outKind,
                        bool out_is_inout)
{
    if (!m || !in || !out) return fail(m ? m->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    const Plan &p = m->plan;
    auto count = [&](int kind) -> int64_t { return kind == MOKA_CELL ? p.nC : kind == MOKA_EDGE ? p.nE : p.nV; };
    int rc = ensure_op_bufs(m);
    if (rc) return rc;
    hipStream_t s = m->ctx->stream;
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    if ((rc = put_rows(m, m->opBuf[0], in, inKind, count(inKind), p.K))) return rc;
    if (out_is_inout) {
        if ((rc = put_rows(m, m->opBuf[1], out, outKind, count(outKind), p.K))) return rc;
    } else {
        // untouched levels (interpolate, nlev < K) must come back as the caller had them
        if (op == OP_INTERP && nlev < p.K) {
            if ((rc = put_rows(m, m->opBuf[1], out, outKind, count(outKind), p.K))) return rc;
        }
    }
    OpArgs a{op, nlev, m->opBuf[0], m->opBuf[1]};
    HIPCHK(m->ctx, launch_operator(m->dev, a, m->lpc, s));
    return get_rows(m, out, m->opBuf[1], outKind, count(outKind), p.K);
}

int moka_gradient_on_edge(moka_mesh *mesh, const double *scalarCell, double *gradEdge)
{
    return run_operator(mesh, OP_GRADIENT, 0, scalarCell, MOKA_CELL, gradEdge, MOKA_EDGE, false);
}

int moka_interpolate_cell2edge(moka_mesh *mesh, const double *cellValue, double *edgeValue, int nlev)
{
    if (mesh && (nlev < 1 || nlev > mesh->plan.K)) return fail(mesh->ctx, MOKA_ERR_ARG, "nlev out of range");
    return run_operator(mesh, OP_INTERP, nlev, cellValue, MOKA_CELL, edgeValue, MOKA_EDGE, false);
}

int moka_curl_on_vertex(moka_mesh *mesh, const double *vecEdge, double *curlVertex)
{
    return run_operator(mesh, OP_CURL, 0, vecEdge, MOKA_EDGE, curlVertex, MOKA_VERTEX, true);
}

// ---------------------------------------------------------------------------------------------
// reverse and forward mode of the stand-alone operators (test/enzyme/test_Enzyme_Operators.jl:42-131, 137-225).
// The operators are linear: forward mode is the operator applied to the tangent, reverse mode its transpose applied to the
// cotangent.  Shadow conventions are Enzyme's for in-place kernels with Duplicated arguments: the shadow of an input is
// accumulated into, the shadow of an output the kernel overwrites is zero afterwards, the shadow of the (accumulated-into)
// curl output stays.  Transposes are gathers with a fixed order (oracle twins: oracle_*_vjp).
// ---------------------------------------------------------------------------------------------
static int ensure_op_transposes(moka_mesh *m)
{
    if (m->opESign) return MOKA_OK;
    const Plan &p = m->plan;
    std::vector<std::vector<std::pair<std::pair<int32_t, int32_t>, int32_t>>> lists(p.nE);     // ((caller's vertex, slot), new vertex)
    for (int v = 0; v < p.nV; ++v)
        for (int j = 0; j < p.VD; ++j) {
            const int e = p.eov[(size_t)v * p.VD + j];
            if (e >= 0) lists[e].push_back({{p.vertN2O[v], j}, v});
        }
    int W = 1;
    for (auto &l : lists) { std::sort(l.begin(), l.end()); W = std::max(W, (int)l.size()); }
    std::vector<int32_t> tv((size_t)p.nE * W, -1);
    std::vector<double> tc((size_t)p.nE * W, 0.0), es((size_t)p.nE * 2, 0.0);
    for (int e = 0; e < p.nE; ++e) {
        for (size_t q = 0; q < lists[e].size(); ++q) {
            const int v = lists[e][q].second, j = lists[e][q].first.second;
            tv[(size_t)e * W + q] = v;
            tc[(size_t)e * W + q] = p.cv[(size_t)v * p.VD + j];
        }
        for (int q = 0; q < 2; ++q) {
            const int c = p.ehdr[(size_t)e * 4 + q];
            for (int i = 0; i < p.ME; ++i)
                if (p.eoc[(size_t)c * p.ME + i] == e) { es[(size_t)e * 2 + q] = p.sdv[(size_t)c * p.ME + i] < 0.0 ? -1.0 : 1.0; break; }
        }
    }
    int rc;
    if ((rc = upload_vec(m, tv, &m->opTVert)) || (rc = upload_vec(m, tc, &m->opTCoef))) return rc;
    if ((rc = upload_vec(m, es, &m->opESign))) return rc;
    m->opTW = W;
    return MOKA_OK;
}

int moka_gradient_on_edge_vjp(moka_mesh *m, double *dGradEdge, double *dScalarCell)
{
    if (!m || !dGradEdge || !dScalarCell) return fail(m ? m->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    const Plan &p = m->plan;
    int rc = ensure_op_bufs(m);
    if (rc) return rc;
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    if ((rc = put_rows(m, m->opBuf[0], dGradEdge, MOKA_EDGE, p.nE, p.K))) return rc;
    if ((rc = put_rows(m, m->opBuf[1], dScalarCell, MOKA_CELL, p.nC, p.K))) return rc;
    OpArgs a{OP_GRAD_T, 0, m->opBuf[0], m->opBuf[1]};
    HIPCHK(m->ctx, launch_operator(m->dev, a, m->lpc, m->ctx->stream));
    if ((rc = get_rows(m, dScalarCell, m->opBuf[1], MOKA_CELL, p.nC, p.K))) return rc;
    std::memset(dGradEdge, 0, sizeof(double) * (size_t)p.K * p.nE);       // the overwritten output's shadow
    return MOKA_OK;
}

int moka_gradient_on_edge_jvp(moka_mesh *m, const double *dScalarCell, double *dGradEdge)
{
    return moka_gradient_on_edge(m, dScalarCell, dGradEdge);              // linear: the tangent map is the operator
}

int moka_divergence_on_cell_vjp(moka_mesh *m, double *dDivCell, double *dVecEdge, double *dTempEdge)
{
    if (!m || !dDivCell || !dVecEdge) return fail(m ? m->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    const Plan &p = m->plan;
    int rc = ensure_op_bufs(m);
    if (rc) return rc;
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    if ((rc = ensure_op_transposes(m))) return rc;
    if ((rc = put_rows(m, m->opBuf[0], dDivCell, MOKA_CELL, p.nC, p.K))) return rc;
    if ((rc = put_rows(m, m->opBuf[1], dVecEdge, MOKA_EDGE, p.nE, p.K))) return rc;
    if (dTempEdge && (rc = put_rows(m, m->opBuf[3], dTempEdge, MOKA_EDGE, p.nE, p.K))) return rc;
    OpArgs a{OP_DIV_T, 0, m->opBuf[0], m->opBuf[1]};
    a.in2 = dTempEdge ? m->opBuf[3] : nullptr;
    a.auxD = m->opESign;
    HIPCHK(m->ctx, launch_operator(m->dev, a, m->lpc, m->ctx->stream));
    if ((rc = get_rows(m, dVecEdge, m->opBuf[1], MOKA_EDGE, p.nE, p.K))) return rc;
    std::memset(dDivCell, 0, sizeof(double) * (size_t)p.K * p.nC);
    if (dTempEdge) std::memset(dTempEdge, 0, sizeof(double) * (size_t)p.K * p.nE);
    return MOKA_OK;
}

int moka_divergence_on_cell_jvp(moka_mesh *m, const double *dVecEdge, double *dTempEdge, double *dDivCell)
{
    return moka_divergence_on_cell(m, dVecEdge, dTempEdge, dDivCell);
}

int moka_curl_on_vertex_vjp(moka_mesh *m, const double *dCurlVertex, double *dVecEdge)
{
    if (!m || !dCurlVertex || !dVecEdge) return fail(m ? m->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    const Plan &p = m->plan;
    int rc = ensure_op_bufs(m);
    if (rc) return rc;
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    if ((rc = ensure_op_transposes(m))) return rc;
    if ((rc = put_rows(m, m->opBuf[0], dCurlVertex, MOKA_VERTEX, p.nV, p.K))) return rc;
    if ((rc = put_rows(m, m->opBuf[1], dVecEdge, MOKA_EDGE, p.nE, p.K))) return rc;
    OpArgs a{OP_CURL_T, 0, m->opBuf[0], m->opBuf[1]};
    a.auxI = m->opTVert; a.auxD = m->opTCoef; a.auxW = m->opTW;
    HIPCHK(m->ctx, launch_operator(m->dev, a, m->lpc, m->ctx->stream));
    return get_rows(m, dVecEdge, m->opBuf[1], MOKA_EDGE, p.nE, p.K);       // the curl shadow stays: the primal accumulates
}

int moka_curl_on_vertex_jvp(moka_mesh *m, const double *dVecEdge, double *dCurlVertex)
{
    return moka_curl_on_vertex(m, dVecEdge, dCurlVertex);                 // accumulates, like the primal
}

int moka_divergence_on_cell(moka_mesh *m, const double *vecEdge, double *tempEdge, double *divCell)
{
    if (!m || !vecEdge || !divCell) return fail(m ? m->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    const Plan &p = m->plan;
    int rc = ensure_op_bufs(m);
    if (rc) return rc;
    hipStream_t s = m->ctx->stream;
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    if ((rc = put_rows(m, m->opBuf[0], vecEdge, MOKA_EDGE, p.nE, p.K))) return rc;
    OpArgs a1{OP_DIV_P1, 0, m->opBuf[0], m->opBuf[1]};      // temp = V*dvEdge (Operators.jl:60)
    HIPCHK(m->ctx, launch_operator(m->dev, a1, m->lpc, s));
    if (tempEdge && (rc = get_rows(m, tempEdge, m->opBuf[1], MOKA_EDGE, p.nE, p.K))) return rc;
    OpArgs a2{OP_DIV_P2, 0, m->opBuf[0], m->opBuf[1]};      // second pass reads V, folds dvEdge*sign exactly
    HIPCHK(m->ctx, launch_operator(m->dev, a2, m->lpc, s));
    return get_rows(m, divCell, m->opBuf[1], MOKA_CELL, p.nC, p.K);
}

// ---------------------------------------------------------------------------------------------
// state
// ---------------------------------------------------------------------------------------------
int moka_state_create(moka_ctx *ctx, moka_mesh *mesh, moka_state **out)
{
    if (!ctx || !mesh || !out) return fail(ctx, MOKA_ERR_ARG, "NULL argument");
    *out = nullptr;
    moka_state *st = new (std::nothrow) moka_state();
    if (!st) return fail(ctx, MOKA_ERR_ALLOC, "out of host memory");
    st->ctx = ctx;
    st->mesh = mesh;
    const Plan &p = mesh->plan;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nEK = (size_t)p.K * p.nE, nCK = (size_t)p.K * p.nC, nVK = (size_t)p.K * p.nV;
    st->f32 = p.stateBytes == 4;
    MeshDev launchDev = mesh->dev;        // what the stage launches will see: maxima over the patches that are ever computed
    launchDev.maxOwnE = std::max(p.maxOwnELaunch, 1); launchDev.maxOwnC = std::max(p.maxOwnCLaunch, 1);
    if (st->f32 && !stage_f32_supported(launchDev)) {
        delete st;
        return fail(ctx, MOKA_ERR_UNSUPPORTED,
                    "fp32-storage state: needs nVertLevels % 4 == 0, nVertLevels <= 128, fields below 4 GiB and patches whose "
                    "records + own u rows fit 64 KB of LDS (smaller patch_cells)");
    }
    const size_t sb = st->f32 ? 4 : 8;
    int rc = MOKA_OK;
    auto A = [&](double **q, size_t n, size_t eb = 8) { if (rc == MOKA_OK) rc = alloc_field(st, q, n, eb); };
    for (auto &l : st->lev) { A(&l.ssh, p.nC, sb); A(&l.u, nEK, sb); A(&l.h, nCK, sb); }
    A(&st->hEdge[0], nEK, sb); A(&st->hEdge[1], nEK, sb);     // DiagnosticVars arrays have the state's storage type
    A(&st->F, nEK, sb); A(&st->div, nCK, sb); A(&st->vort, nVK, sb);
    A(&st->tendU, nEK, sb); A(&st->tendH, nCK, sb);       // fp32-storage states store their tendencies fp32 as well
    A(&st->scalar, 2);
    if (rc != MOKA_OK) { moka_state_destroy(st); return rc; }
    st->phys[0] = st->lev[0]; st->phys[1] = st->lev[1];
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    {
        std::lock_guard<std::mutex> lk(g_liveMutex);
        g_liveStates.insert(st);
    }
    *out = st;
    return MOKA_OK;
}

void moka_state_destroy(moka_state *st)
{
    if (!st) return;
    {
        std::lock_guard<std::mutex> lk(g_liveMutex);
        g_liveStates.erase(st);
    }
    (void)hipSetDevice(st->ctx->device);
    (void)hipStreamSynchronize(st->ctx->stream);
    for (void *q : st->allocs) (void)hipFree(q);
    delete st;
}

int moka_state_upload(moka_state *st, int field, int time_level, const double *host)
{
    if (!st || !host) return fail(st ? st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    FieldRef r;
    int rc = field_ref(st, field, time_level, &r);
    if (rc) return rc;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    // the arrays of a lean Forward-Euler step are derived from the previous level (and the one before): whatever the caller
    // overwrites, they are produced from the values the step saw
    if (st->feLazy && !(field <= MOKA_F_LAYER_THICKNESS && time_level == 1))
        if ((rc = flush_lazy(st, true, true))) return rc;
    const bool prog1 = field <= MOKA_F_LAYER_THICKNESS && time_level == 1;   // pending lazy results refer to the old state
    // (an fp32-storage state cannot materialise pending diagnostics: they stay pending, i.e. unavailable)
    if ((rc = flush_lazy(st, (prog1 && !st->f32) || is_diag_field(field), prog1 || is_tend_field(field)))) return rc;
    if ((rc = field_ref(st, field, time_level, &r))) return rc;   // flush may have swapped buffers
    if (time_level == 1 && (field == MOKA_F_SSH || field == MOKA_F_LAYER_THICKNESS)) st->sshConsistent = false;
    if ((field == MOKA_F_LAYER_THICKNESS && time_level == 0) || field == MOKA_F_LAYER_THICKNESS_EDGE) st->hEdgePrev = false;
    return put_rows(st->mesh, r.ptr, host, r.kind, r.n, r.K, r.f32);
}

int moka_state_download(moka_state *st, int field, int time_level, double *host)
{
    if (!st || !host) return fail(st ? st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    FieldRef r;
    int rc = field_ref(st, field, time_level, &r);
    if (rc) return rc;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if ((rc = flush_lazy(st, is_diag_field(field), is_tend_field(field)))) return rc;
    if ((rc = field_ref(st, field, time_level, &r))) return rc;   // flush may have swapped buffers
    return get_rows(st->mesh, host, r.ptr, r.kind, r.n, r.K, r.f32);
}

int moka_advance_time_levels(moka_state *st, int flags)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    const Plan &p = st->mesh->plan;
    hipStream_t s = st->ctx->stream;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (int rcl = flush_lazy(st, st->feLazy, st->feLazy)) return rcl;      // a lean step's arrays derive from the level overwritten now
    st->hEdgePrev = false;                 // the previous level changes under the stored layerThicknessEdge
    // advance_2d_array / advance_3d_array (time_integration.jl:42-59): prev <- next.
    // Level-1-only copies (K > 1) go through the FE kernel's carry-over path instead.
    if ((flags & MOKA_FE_LEVEL1_ONLY) && p.K > 1)
        return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "level-1-only advanceTimeLevels! is only available inside moka_step_fe");
    if (st->f32) {   // float arrays: plain device copies
        HIPCHK(st->ctx, hipMemcpyAsync(st->lev[0].ssh, st->lev[1].ssh, (size_t)p.nC * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(st->ctx, hipMemcpyAsync(st->lev[0].u, st->lev[1].u, (size_t)p.K * p.nE * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(st->ctx, hipMemcpyAsync(st->lev[0].h, st->lev[1].h, (size_t)p.K * p.nC * 4, hipMemcpyDeviceToDevice, s));
        return MOKA_OK;
    }
    HIPCHK(st->ctx, launch_copy(st->lev[0].ssh, st->lev[1].ssh, p.nC, s));
    HIPCHK(st->ctx, launch_copy(st->lev[0].u, st->lev[1].u, (int64_t)p.K * p.nE, s));
    HIPCHK(st->ctx, launch_copy(st->lev[0].h, st->lev[1].h, (int64_t)p.K * p.nC, s));
    return MOKA_OK;
}

int moka_diagnostic_compute(moka_state *st, int flags)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    if (st->nonlinear) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "nonlinear terms: moka_tendencies / RK4 only");
    if (st->f32) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "fp32-storage state: moka_step_fe, moka_step_rk4 and moka_tendencies only (the piecewise reference calls are Float64)");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (int rcl = flush_lazy(st, true, false)) return rcl;
    FeArgs a = fe_args(st, FE_FLUX | FE_DIV | FE_CURL | FE_HEDGE, flags, 0.0);
    HIPCHK(st->ctx, launch_fe(st->mesh->dev, a, st->mesh->lpc, st->ctx->stream));
    std::swap(st->hEdge[0], st->hEdge[1]);
    st->hEdgePrev = false;
    return MOKA_OK;
}

int moka_compute_normal_velocity_tendency(moka_state *st, int flags)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    if (st->nonlinear) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "nonlinear terms: moka_tendencies / RK4 only");
    if (st->f32) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "fp32-storage state: moka_step_fe, moka_step_rk4 and moka_tendencies only (the piecewise reference calls are Float64)");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (int rcl = flush_lazy(st, false, true)) return rcl;
    FeArgs a = fe_args(st, FE_TENDU, flags, 0.0);
    HIPCHK(st->ctx, launch_fe(st->mesh->dev, a, st->mesh->lpc, st->ctx->stream));
    return MOKA_OK;
}

int moka_compute_layer_thickness_tendency(moka_state *st, int flags)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    if (st->nonlinear) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "nonlinear terms: moka_tendencies / RK4 only");
    if (st->f32) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "fp32-storage state: moka_step_fe, moka_step_rk4 and moka_tendencies only (the piecewise reference calls are Float64)");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (int rcl = flush_lazy(st, true, true)) return rcl;
    FeArgs a = fe_args(st, FE_TENDH | FE_TENDH_FROM_F, flags, 0.0);
    HIPCHK(st->ctx, launch_fe(st->mesh->dev, a, st->mesh->lpc, st->ctx->stream));
    return MOKA_OK;
}

static int make_ssh_consistent(moka_state *st, double *dst)
{
    const Plan &p = st->mesh->plan;
    if (st->f32)
        HIPCHK(st->ctx, launch_update_ssh_f32(st->mesh->dev, reinterpret_cast<const float *>(st->lev[1].h),
                                              reinterpret_cast<float *>(dst), p.K, st->mesh->lpc, st->ctx->stream));
    else
        HIPCHK(st->ctx, launch_update_ssh(st->mesh->dev, st->lev[1].h, dst, p.K, st->mesh->lpc, st->ctx->stream));
    return MOKA_OK;
}

int moka_tendencies(moka_state *st)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    int rc;
    if (!st->sshConsistent) {
        if ((rc = make_ssh_consistent(st, st->lev[1].ssh))) return rc;
        st->sshConsistent = true;
    }
    if (st->feLazy && (rc = flush_lazy(st, true, true))) return rc;        // (it would overwrite the tendencies written here later)
    StageArgs a{};
    a.pu = st->lev[1].u; a.ph = st->lev[1].h; a.ssh = st->lev[1].ssh;
    a.tendU = st->tendU; a.tendH = st->tendH;
    HIPCHK(st->ctx, run_stage(st, a));
    st->tendDirty = false;
    return MOKA_OK;
}

int moka_step_fe(moka_state *st, double dt, int flags)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    if (st->nonlinear) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "nonlinear terms: moka_tendencies / RK4 only");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    const moka_mesh *mm = st->mesh;
    const bool whole = mm->plan.nPatchesLaunch == mm->plan.nPatches;
    if (st->f32 && ((flags & MOKA_FE_LEVEL1_ONLY) || !whole))
        return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "fp32-storage state: Forward Euler steps all levels of a whole mesh (no MOKA_FE_LEVEL1_ONLY, no partitions)");
    // All levels on a whole mesh with the default kernel choice: the step runs in the stage kernels (modes 4 / 5 / 6 of
    // k_stage_rec2c*), relativeVorticity in the same launch where the vertex records fit.  Anything else -- MOKA_FE_LEVEL1_ONLY, odd
    // or large K, explicit kernel variants, local meshes of a partition outside the halo API -- takes the generic one-launch kernel.
    const bool stagePath = whole && fe_stage_path(st, flags);
    if (stagePath)
        if (int rcs = ensure_spare(st)) return rcs;            // the new level goes to the spare set; the previous one stays readable
    const bool lean = stagePath && fe_lean(st, flags);
    if (int rcl = fe_begin(st, flags)) return rcl;
    // advanceTimeLevels! + diagnostic_compute! + both tendencies + updates (time_integration.jl:163-189) in one launch: the new
    // level is written into the spare set (or, without one, over the previous level), then the levels rotate.
    FeArgs a = fe_args(st, FE_FLUX | FE_DIV | FE_CURL | FE_HEDGE | FE_TENDU | FE_TENDH | FE_UPDATE, flags, dt);
    bool fast = false, prevMode = false;
    if (stagePath) {
        StageArgs g = fe_stage_args(st, a, flags);
        MeshDev dev = mm->dev;
        dev.tailPatch = -1;
        dev.maxOwnE = std::max(mm->plan.maxOwnELaunch, 1); dev.maxOwnC = std::max(mm->plan.maxOwnCLaunch, 1);
        auto launch = [&]() { return st->f32 ? launch_stage_rec2c_f32(dev, g, st->ctx->stream) : launch_stage_rec2c(dev, g, st->ctx->stream); };
        hipError_t e = launch();
        if (e == hipErrorNotSupported && g.vort) {             // vertex records do not fit beside the rows after all: own launch
            g.vort = nullptr;
            e = launch();
        }
        if (e == hipSuccess) {
            fast = true;
            prevMode = g.hPrev != nullptr;
            if (!g.vort) {                                     // ssh as stored, relativeVorticity of the OLD state
                if (st->f32) {
                    HIPCHK(st->ctx, launch_curl_f32(dev, reinterpret_cast<const float *>(a.u), reinterpret_cast<float *>(a.vort),
                                                    flags & MOKA_FE_ACCUM_VORT, st->ctx->stream));
                } else {
                    const hipError_t ec = launch_curl2(mm->dev, a.u, a.vort, flags & MOKA_FE_ACCUM_VORT, st->ctx->stream);
                    if (ec == hipErrorNotSupported) {
                        a.ops = FE_CURL;
                        HIPCHK(st->ctx, launch_fe(mm->dev, a, mm->lpc, st->ctx->stream));
                    } else {
                        HIPCHK(st->ctx, ec);
                    }
                }
            }
        } else if (e != hipErrorNotSupported || st->f32 || lean) {
            HIPCHK(st->ctx, e);                                // (fp32 storage and lean steps have no generic form behind them)
            return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "internal: the stage kernel refused a Forward-Euler launch it was selected for");
        }
    }
    if (!fast) HIPCHK(st->ctx, launch_fe(mm->dev, a, mm->lpc, st->ctx->stream));
    fe_end(st, flags, fast, fast && lean, prevMode);
    st->sshConsistent = !(flags & MOKA_FE_LEVEL1_ONLY) || mm->plan.K == 1;
    return MOKA_OK;
}

int moka_fe_lazy_pending(const moka_state *st) { return st && st->feLazy ? 1 : 0; }

// Argument block of RK4 stage s (1..4).  A = Curr (current level), B = New accumulator (the previous
// level's buffers, becomes the current level at the end), R1/R2 = provisional states.
}  // extern "C"
namespace mk {
StageArgs rk4_stage_args(moka_state *st, int s, double dt, const double *ssh0)
{
    const double a[3] = {dt / 2., dt / 2., dt};                         // time_integration.jl:77
    const double b[4] = {dt / 6., dt / 3., dt / 3., dt / 6.};           // :78
    LevelBufs &A = st->lev[1], &B = st->lev[0], &R1 = st->rk[0], &R2 = st->rk[1];
    StageArgs g{};
    g.nu_out = B.u; g.nh_out = B.h; g.b = b[s - 1];
    if (s == 1) {          // Provis == Curr == A;  New = Curr + b1*k1 -> B;  Provis' = Curr + a1*k1 -> R1
        g.pu = A.u; g.ph = A.h; g.ssh = ssh0;
        g.pu_out = R1.u; g.ph_out = R1.h; g.ssh_out = R1.ssh; g.a = a[0];
    } else if (s == 2) {   // Provis = R1 -> R2
        g.pu = R1.u; g.ph = R1.h; g.ssh = R1.ssh; g.cu = A.u; g.ch = A.h; g.nu_in = B.u; g.nh_in = B.h;
        g.pu_out = R2.u; g.ph_out = R2.h; g.ssh_out = R2.ssh; g.a = a[1];
    } else if (s == 3) {   // Provis = R2 -> R1
        g.pu = R2.u; g.ph = R2.h; g.ssh = R2.ssh; g.cu = A.u; g.ch = A.h; g.nu_in = B.u; g.nh_in = B.h;
        g.pu_out = R1.u; g.ph_out = R1.h; g.ssh_out = R1.ssh; g.a = a[2];
    } else {               // Provis = R1; New += b4*k4; ssh of New
        g.pu = R1.u; g.ph = R1.h; g.ssh = R1.ssh; g.cu = A.u; g.ch = A.h; g.nu_in = B.u; g.nh_in = B.h;
        g.ssh_out = B.ssh; g.a = 0.0;
    }
    return g;
}

// buffers a stage writes that other ranks gather from next: stage 1,3 -> R1; 2 -> R2; 4 -> B; 0 -> current level
// 5 = the new level of a distributed Forward-Euler step before its levels rotate
LevelBufs &rk4_stage_output(moka_state *st, int s)
{
    return s == 0 ? st->lev[1] : s == 5 ? fe_new_level(st) : s == 4 ? st->lev[0] : s == 2 ? st->rk[1] : st->rk[0];
}

int rk4_begin(moka_state *st, const double **ssh0)
{
    int rc = ensure_rk_bufs(st);
    if (rc) return rc;
    st->feLazy = false;              // the arrays a lean Forward-Euler step left pending are superseded by this step's (rk4_end)
    *ssh0 = st->lev[1].ssh;
    if (!st->sshConsistent) {    // the tendency of stage 1 uses ssh computed from layerThickness
        if ((rc = make_ssh_consistent(st, st->rk[1].ssh))) return rc;
        *ssh0 = st->rk[1].ssh;
    }
    return MOKA_OK;
}

void rk4_end(moka_state *st)
{
    std::swap(st->lev[0], st->lev[1]);
    st->sshConsistent = true;
    st->diagDirty = true;
    st->tendDirty = true;
    st->hEdgePrev = false;
    st->lazyPu = st->lazyPh = nullptr; st->lazyOwner = nullptr;
}

// ---- the RK4 step with 13 instead of 16 state streams (opt-in: moka_set_tuning key 7) -----------------------------------------
// The reference accumulates New += b_s k_s through the four stages (time_integration.jl:134-135): New is written by stage 1 and
// read + written by stages 2-4 = 7 of the step's 16 state streams.  The provisional states carry the same information:
// P2 - C = dt/2 k1, P3 - C = dt/2 k2, P4 - C = dt k3, so New = C + ((P2 - C) + 2 (P3 - C) + (P4 - C)) / 3 + dt/6 k4 can be formed by
// stage 4 alone from OWN rows (no gathers): streams per stage 2 / 3 / 3 / 5 = 13.  Same four buffer sets: P2 -> R1, P3 -> R2,
// P4 -> the previous level's set, New over P2 in place; the sets then rotate (current <- R1, previous <- old current, R1 <- old
// previous holding P4, which the lazily produced stage-4 tendencies read).  Round-off differs from the running sum (a few
// units in the last place of the state per step), hence opt-in; Float64 states on whole meshes through the default stage kernel
// (with the nonlinear terms: through k_stage_nl5, twin oracle_step_rk4_nonlinear_s13).
bool rk13_usable(const moka_state *st)
{
    if (!g_rk13.load() || st->f32) return false;
    const moka_mesh *mm = st->mesh;
    if (mm->plan.nPatchesLaunch != mm->plan.nPatches) return false;
    if (st->nonlinear)             // nonlinear terms: k_stage_nl5 carries the form (StageArgs.rkMode 9), the plainer kernels do not
        return nl_stage_is_nl5(mm->dev, mm->lpc, st->ctx->variant == 4 ? 1 : st->ctx->variant == 3 ? 3 : 0);
    MeshDev dev = mm->dev;
    dev.maxOwnE = std::max(mm->plan.maxOwnELaunch, 1); dev.maxOwnC = std::max(mm->plan.maxOwnCLaunch, 1);
    return (st->ctx->variant == 0 || st->ctx->variant == 11) && mm->lpc == 64 && mm->colOk && rec2c_supported(dev);
}

StageArgs rk13_stage_args(moka_state *st, int s, double dt, const double *ssh0)
{
    LevelBufs &A = st->lev[1], &B = st->lev[0], &R1 = st->rk[0], &R2 = st->rk[1];
    StageArgs g{};
    if (s == 1) {          // P2 = C + dt/2 k1 -> R1
        g.pu = A.u; g.ph = A.h; g.ssh = ssh0;
        g.pu_out = R1.u; g.ph_out = R1.h; g.ssh_out = R1.ssh; g.a = dt / 2.; g.rkMode = 7;
    } else if (s == 2) {   // P3 = C + dt/2 k2 -> R2
        g.pu = R1.u; g.ph = R1.h; g.ssh = R1.ssh; g.cu = A.u; g.ch = A.h;
        g.pu_out = R2.u; g.ph_out = R2.h; g.ssh_out = R2.ssh; g.a = dt / 2.; g.rkMode = 8;
    } else if (s == 3) {   // P4 = C + dt k3 -> B
        g.pu = R2.u; g.ph = R2.h; g.ssh = R2.ssh; g.cu = A.u; g.ch = A.h;
        g.pu_out = B.u; g.ph_out = B.h; g.ssh_out = B.ssh; g.a = dt; g.rkMode = 8;
    } else {               // New over P2 (R1), ssh of New
        g.pu = B.u; g.ph = B.h; g.ssh = B.ssh; g.cu = A.u; g.ch = A.h;
        g.nu_in = R1.u; g.nh_in = R1.h; g.q3u = R2.u; g.q3h = R2.h;
        g.nu_out = R1.u; g.nh_out = R1.h; g.ssh_out = R1.ssh; g.b = dt / 6.; g.rkMode = 9;
    }
    return g;
}

void rk13_end(moka_state *st)
{
    const LevelBufs A = st->lev[1], B = st->lev[0], R1 = st->rk[0];
    st->lev[1] = R1;             // New
    st->lev[0] = A;              // the level the step started from
    st->rk[0] = B;               // P4: the provisional state of stage 4 (lazily produced tendencies read it, like rk4_end's rk[0])
    st->sshConsistent = true;
    st->diagDirty = true;
    st->tendDirty = true;
    st->hEdgePrev = false;
    st->lazyPu = st->lazyPh = nullptr; st->lazyOwner = nullptr;
}
}  // namespace mk
extern "C" {

int moka_state_rk4_streams(const moka_state *st) { return !st ? 0 : rk13_usable(st) ? 13 : 16; }

int moka_step_rk4(moka_state *st, double dt)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    const double *ssh0 = nullptr;
    int rc = rk4_begin(st, &ssh0);
    if (rc) return rc;
    moka_ctx *c = st->ctx;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(c->stream, &cap);
    const bool timed = c->stageTiming && cap == hipStreamCaptureStatusNone;
    auto stamp = [&]() -> hipError_t {
        if (c->evUsed == c->evPool.size()) {
            hipEvent_t e = nullptr;
            if (hipError_t er = hipEventCreate(&e); er != hipSuccess) return er;
            c->evPool.push_back(e);
        }
        return hipEventRecord(c->evPool[c->evUsed++], c->stream);
    };
    const bool s13 = rk13_usable(st);
    for (int s = 1; s <= 4; ++s) {
        if (timed) HIPCHK(c, stamp());
        HIPCHK(c, run_stage(st, s13 ? rk13_stage_args(st, s, dt, ssh0) : rk4_stage_args(st, s, dt, ssh0)));
    }
    if (timed) HIPCHK(c, stamp());
    if (s13) rk13_end(st);
    else rk4_end(st);
    return MOKA_OK;
}

// Per-stage kernel durations of moka_step_rk4 from HIP events on the compute stream (bench.py's per-mode roofline lines).
// enable != 0: forget earlier samples and start recording 5 events per step; enable == 0: stop.
int moka_stage_timing(moka_ctx *ctx, int enable)
{
    if (!ctx) return fail(nullptr, MOKA_ERR_ARG, "ctx is NULL");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stageTiming = enable != 0;
    if (enable) ctx->evUsed = 0;
    return MOKA_OK;
}

// ms[s-1] = mean duration of the stage-s launch over the recorded steps; *steps = how many steps were recorded
int moka_stage_timing_read(moka_ctx *ctx, double ms[4], int64_t *steps)
{
    if (!ctx || !ms || !steps) return fail(ctx, MOKA_ERR_ARG, "NULL argument");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n = ctx->evUsed / 5;
    for (int s = 0; s < 4; ++s) ms[s] = 0.0;
    for (size_t i = 0; i < n; ++i)
        for (int s = 0; s < 4; ++s) {
            float t = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&t, ctx->evPool[5 * i + s], ctx->evPool[5 * i + s + 1]));
            ms[s] += t;
        }
    for (int s = 0; s < 4; ++s) ms[s] = n ? ms[s] / (double)n : 0.0;
    *steps = (int64_t)n;
    return MOKA_OK;
}

int moka_run(moka_state *st, int integrator, double dt, int64_t nsteps, int flags)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    if (nsteps < 0) return fail(st->ctx, MOKA_ERR_ARG, "nsteps must be >= 0");
    if (integrator != MOKA_FORWARD_EULER && integrator != MOKA_RUNGE_KUTTA_4) return fail(st->ctx, MOKA_ERR_ARG, "unknown integrator");
    auto one = [&]() { return integrator == MOKA_FORWARD_EULER ? moka_step_fe(st, dt, flags) : moka_step_rk4(st, dt); };
    int64_t done = 0;
    int rc;
    // Launch-bound regime (small meshes): capture one PERIOD of consecutive steps into a hipGraph and replay it.  The
    // pointer pattern of a step repeats after 2 steps when the two time-level sets swap (RK4; Forward Euler without a spare
    // set) and after 6 when a Forward-Euler step rotates three level sets (period 3) while layerThicknessEdge's two buffers
    // swap (period 2).  The first step runs eagerly: it may have to make ssh consistent and allocate the RK buffers or the
    // spare set, none of which may happen during capture -- and it brings the state into the steady regime every later step
    // is launched in (e.g. layerThicknessEdge formed from the previous level from the second step on).
    if (nsteps >= 6) {
        if ((rc = one())) return rc;
        ++done;
        if (integrator == MOKA_FORWARD_EULER && st->spare.ssh && nsteps - done >= 2) {   // one more: the regime settles with step 2
            if ((rc = one())) return rc;
            ++done;
        }
    }
    // (the 13-stream RK4 form rotates three buffer sets: period 3; 6 serves it too)
    const int period = ((integrator == MOKA_FORWARD_EULER && st->spare.ssh) || (integrator == MOKA_RUNGE_KUTTA_4 && rk13_usable(st))) ? 6 : 2;
    if (done > 0 && nsteps - done >= period) {
        HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
        // nothing lazy may fire inside the capture (pending diagnostics of an fp32-storage state cannot be produced: they
        // stay pending -- an RK4 run leaves them pending anyway, a Forward-Euler step has none)
        if ((rc = flush_lazy(st, !st->f32, true))) return rc;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipStream_t s = st->ctx->stream;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const bool d0 = st->diagDirty, t0 = st->tendDirty;
            int rc2 = MOKA_OK;
            for (int i = 0; i < period && rc2 == MOKA_OK; ++i) {
                st->diagDirty = st->tendDirty = false;         // keep flush_lazy inside the steps a no-op while capturing
                rc2 = one();
            }
            hipError_t ec = hipStreamEndCapture(s, &graph);
            if (rc2 || ec != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                st->diagDirty = d0; st->tendDirty = t0;
                return rc2 ? rc2 : fail(st->ctx, MOKA_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ec));
            }
            // the capture recorded the launches but executed nothing: the host-side swaps / rotations of one period cancel
            if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                hipError_t el = hipSuccess;
                while (nsteps - done >= period && (el = hipGraphLaunch(exec, s)) == hipSuccess) done += period;
                (void)hipGraphExecDestroy(exec);
                if (el != hipSuccess) {
                    (void)hipGraphDestroy(graph);
                    return fail(st->ctx, MOKA_ERR_HIP, std::string("hipGraphLaunch: ") + hipGetErrorString(el));
                }
            }
            (void)hipGraphDestroy(graph);
            st->diagDirty = st->tendDirty = (integrator == MOKA_RUNGE_KUTTA_4);
        }
    }
    for (; done < nsteps; ++done)
        if ((rc = one())) return rc;
    return MOKA_OK;
}

int moka_sum_sq(moka_state *st, int field, int time_level, double *out)
{
    if (!st || !out) return fail(st ? st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    FieldRef r;
    int rc = field_ref(st, field, time_level, &r);
    if (rc) return rc;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if ((rc = flush_lazy(st, is_diag_field(field), is_tend_field(field)))) return rc;     // lazily pending arrays are produced on a read
    if ((rc = field_ref(st, field, time_level, &r))) return rc;                            // (the flush may have swapped buffers)
    if ((rc = ensure_op_bufs(st->mesh))) return rc;
    hipStream_t s = st->ctx->stream;
    // caller's numbering, then the strictly serial order of sumArray (run_loop.jl:47-51)
    if (r.f32) HIPCHK(st->ctx, launch_permute_rows_f32(st->mesh->opBuf[2], r.ptr, perm_of(st->mesh, r.kind), r.n, r.K, 0, s));
    else HIPCHK(st->ctx, launch_permute_rows(st->mesh->opBuf[2], r.ptr, perm_of(st->mesh, r.kind), r.n, r.K, 0, s));
    HIPCHK(st->ctx, launch_sum_sq_serial(st->mesh->opBuf[2], r.n * r.K, st->scalar, s));
    HIPCHK(st->ctx, hipMemcpyAsync(out, st->scalar, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(st->ctx, hipStreamSynchronize(s));
    return MOKA_OK;
}

int moka_last_fe_path(const moka_state *st) { return st ? st->feFast : -1; }

int moka_set_nonlinear(moka_state *st, int on)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    if (!on) { st->nonlinear = false; return MOKA_OK; }
    const Plan &p = st->mesh->plan;
    if (st->f32) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "nonlinear terms: Float64 states only");
    if (!p.nlOk)
        return fail(st->ctx, MOKA_ERR_UNSUPPORTED,
                    "nonlinear terms need kiteAreasOnVertex, fVertex, verticesOnEdge and cellsOnVertex in the mesh descriptor");
    // (a rank-local mesh is fine when its halo is two cells deep -- moka_hip.parallel.build_local(rings = 2) -- and every stage
    //  is launched over the whole local mesh: moka_rk4_dist_stage(h, s, 2); what its rim computes is never read by an owned entity)
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    int rc = flush_lazy(st, true, true);
    if (rc) return rc;
    if (!st->nlQv) {
        if ((rc = alloc_field(st, &st->nlQv, (size_t)p.K * p.nV))) return rc;
        if ((rc = alloc_field(st, &st->nlQe, 2 * (size_t)p.K * p.nE))) return rc;   // {F, q_e} pairs
        if ((rc = alloc_field(st, &st->nlKe, (size_t)p.K * p.nC))) return rc;
    }
    st->nonlinear = true;
    return MOKA_OK;
}

int moka_set_viscosity_del2(moka_state *st, double viscDel2)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    if (!(viscDel2 >= 0.0)) return fail(st->ctx, MOKA_ERR_ARG, "viscDel2 must be >= 0");
    if (viscDel2 != 0.0 && !st->nonlinear)
        return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "Del2 mixing rides on the nonlinear tendencies: call moka_set_nonlinear first");
    if (viscDel2 != 0.0 && !st->nlZv) {
        const Plan &p = st->mesh->plan;
        HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
        int rc;
        if ((rc = alloc_field(st, &st->nlZv, (size_t)p.K * p.nV))) return rc;
        if ((rc = alloc_field(st, &st->nlDiv, (size_t)p.K * p.nC))) return rc;
    }
    st->viscDel2 = viscDel2;
    return MOKA_OK;
}

// ---------------------------------------------------------------------------------------------
// reverse mode of the Forward-Euler loop.  The reference differentiates
// ocn_run_loop with Enzyme (ext/MPASEnzymeExt.jl; test/enzyme/test_Enzyme_end2end.jl: d sum(ssh^2) / d initial
// layerThickness, normalVelocity); here the tape and the transposed kernels are written by hand.
// ---------------------------------------------------------------------------------------------
struct moka_tape {
    moka_state *st = nullptr;
    int64_t capacity = 0, n = 0;
    double *uTape = nullptr, *hTape = nullptr;      // capacity x (K, nE): u_n and the hEdge the flux of step n used
    std::vector<double> dts;
    std::vector<int> flags;
    double *lamU[2] = {nullptr, nullptr}, *lamH[2] = {nullptr, nullptr}, *lamS[2] = {nullptr, nullptr}, *lamE[2] = {nullptr, nullptr};
    double *Enew = nullptr, *csum = nullptr;
    // RK4: per step the four provisional states the tendencies were evaluated at; work arrays of the reverse step
    double *rkU = nullptr, *rkH = nullptr;          // capacity x 4 x (K, nE) / (K, nC), allocated at the first RK4 step
    double *kbU = nullptr, *kbH = nullptr, *pbU = nullptr, *pbH = nullptr;
    int kind = -1;                                   // -1 empty, 0 Forward Euler, 1 RK4 (one integrator per tape)
    int cur = 0;                                     // index of the adjoint state that is current
    int revNext = 4;                                 // stage-wise reverse RK4 step (moka_adjoint_rk4_stage): the stage that comes next
    int revPart = 0;                                 // ... and, part by part (moka_adjoint_rk4_stage_part), which part of it
    int recMask = 0;                                 // piecewise taping (moka_tape_record_rk4): slots of the open step already filled
    bool seeded = false;
    moka::AdjMesh am{};
    std::vector<void *> allocs;
    bool counted = false;                            // st->attached includes this tape
};

static int tape_alloc(moka_tape *t, void **out, size_t bytes)
{
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, std::max<size_t>(bytes, 16));
    if (e != hipSuccess)
        return fail(t->st->ctx, MOKA_ERR_ALLOC, std::string("tape: hipMalloc of ") + std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
    t->allocs.push_back(d);
    HIPCHK(t->st->ctx, hipMemsetAsync(d, 0, bytes, t->st->ctx->stream));
    *out = d;
    return MOKA_OK;
}

extern "C++" {
template <class T>
static int tape_upload(moka_tape *t, const std::vector<T> &v, const T **out)
{
    void *d = nullptr;
    int rc = tape_alloc(t, &d, v.size() * sizeof(T));
    if (rc) return rc;
    // on the context's stream, behind tape_alloc's memset: a plain hipMemcpy runs on the null stream, which the (non-blocking)
    // context stream does not order against -- once in ~1500 tapes the zero fill landed after the copy and wiped the lists
    if ((rc = h2d(t->st->ctx, d, v.data(), v.size() * sizeof(T)))) return rc;     // v is a temporary of the caller
    *out = static_cast<const T *>(d);
    return MOKA_OK;
}
}  // extern "C++"

int moka_tape_create(moka_state *st, int64_t capacity_steps, moka_tape **out)
{
    if (!st || !out || capacity_steps < 0) return fail(st ? st->ctx : nullptr, MOKA_ERR_ARG, "bad argument");
    *out = nullptr;
    if (st->f32) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "reverse mode: Float64 states only");
    if (st->nonlinear) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "reverse mode covers the reference's linear terms only");
    const Plan &p = st->mesh->plan;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    // transposed Coriolis stencil, sources sorted by (caller's edge id, slot): the oracle's summation order
    const int nE = p.nE, nC = p.nC, ME = p.ME, ME2 = p.ME2;
    std::vector<std::vector<std::pair<std::pair<int32_t, int32_t>, int32_t>>> lists(nE);   // ((orig src, slot), new src)
    for (int s = 0; s < nE; ++s)
        for (int i = 0; i < ME2; ++i) {
            const int tgt = p.eoe[(size_t)s * ME2 + i];
            if (tgt >= 0) lists[tgt].push_back({{p.edgeN2O[s], i}, s});
        }
    int W = 1;
    for (auto &l : lists) { std::sort(l.begin(), l.end()); W = std::max(W, (int)l.size()); }
    std::vector<int32_t> teoe((size_t)nE * W, -1), csgn((size_t)nC * ME, 0);
    std::vector<double> tw((size_t)nE * W, 0.0), sd((size_t)nE * 2, 0.0);
    for (int e = 0; e < nE; ++e)
        for (size_t j = 0; j < lists[e].size(); ++j) {
            const int s = lists[e][j].second, i = lists[e][j].first.second;
            teoe[(size_t)e * W + j] = s;
            tw[(size_t)e * W + j] = p.woe[(size_t)s * ME2 + i];
        }
    for (int c = 0; c < nC; ++c)
        for (int i = 0; i < ME; ++i)
            if (p.eoc[(size_t)c * ME + i] >= 0) csgn[(size_t)c * ME + i] = p.sdv[(size_t)c * ME + i] < 0.0 ? -1 : 1;
    for (int e = 0; e < nE; ++e) {
        const int cc[2] = {p.ehdr[(size_t)e * 4], p.ehdr[(size_t)e * 4 + 1]};
        // (cc[0] == cc[1]: an outermost edge of a rank-local mesh, both sides set to the inside halo cell.  Its adjoint is never
        //  computed here -- it arrives by exchange, like its forward value -- so its factors may be anything finite.)
        for (int q = 0; q < 2; ++q) {
            double sgn = 0.0;
            for (int i = 0; i < ME; ++i)
                if (p.eoc[(size_t)cc[q] * ME + i] == e) { sgn = (double)csgn[(size_t)cc[q] * ME + i]; break; }
            sd[(size_t)e * 2 + q] = p.dvEdge[e] * sgn * p.invArea[cc[q]];
        }
    }
    moka_tape *t = new (std::nothrow) moka_tape();
    if (!t) return fail(st->ctx, MOKA_ERR_ALLOC, "out of host memory");
    t->st = st;
    t->capacity = capacity_steps;
    const size_t nEK = (size_t)p.K * nE, nCK = (size_t)p.K * nC;
    int rc = MOKA_OK;
    auto A = [&](double **q, size_t n) { if (rc == MOKA_OK) { void *d = nullptr; rc = tape_alloc(t, &d, n * sizeof(double)); *q = static_cast<double *>(d); } };
    A(&t->uTape, nEK * (size_t)capacity_steps); A(&t->hTape, nEK * (size_t)capacity_steps);
    for (int b = 0; b < 2; ++b) { A(&t->lamU[b], nEK); A(&t->lamH[b], nCK); A(&t->lamS[b], nC); A(&t->lamE[b], nEK); }
    A(&t->Enew, nEK); A(&t->csum, nE);
    moka::AdjMesh &am = t->am;
    am.nC = nC; am.nE = nE; am.K = p.K; am.ME = ME; am.W = W;
    am.eBegin = 0; am.eCount = nE; am.cBegin = 0; am.cCount = nC;
    am.eoc = st->mesh->dev.eoc; am.ehdr = st->mesh->dev.ehdr; am.fEdge = st->mesh->dev.fEdge; am.gInvDc = st->mesh->dev.gInvDc;
    if (rc == MOKA_OK) rc = tape_upload(t, teoe, &am.teoe);
    if (rc == MOKA_OK) rc = tape_upload(t, tw, &am.tw);
    if (rc == MOKA_OK) rc = tape_upload(t, sd, &am.sd);
    if (rc == MOKA_OK) rc = tape_upload(t, csgn, &am.csgn);
    {   // regular edges (everything but the neighbourhood of the 12 pentagons on a sphere without land): mask-free path
        std::vector<int32_t> efull(nE, 0);
        for (int e = 0; e < nE; ++e) {
            bool full = p.ehdr[(size_t)e * 4 + 3] >= p.K;
            for (int j = 0; j < W && full; ++j) {
                const int sidx = teoe[(size_t)e * W + j];
                full = sidx >= 0 && p.ehdr[(size_t)sidx * 4 + 3] >= p.K;
            }
            efull[e] = full ? 1 : 0;
        }
        if (rc == MOKA_OK) rc = tape_upload(t, efull, &am.efull);
    }
    if (rc != MOKA_OK) { moka_tape_destroy(t); return rc; }
    HIPCHK(st->ctx, hipStreamSynchronize(st->ctx->stream));
    state_attach(st);
    t->counted = true;
    *out = t;
    return MOKA_OK;
}

void moka_tape_destroy(moka_tape *t)
{
    if (!t) return;
    if (t->counted) state_detach(t->st);
    (void)hipSetDevice(t->st->ctx->device);
    if (t->st->lazyOwner == t) {        // the stage-4 tendencies of the last taped step would be produced from a slot of this tape
        if (t->st->tendDirty) (void)flush_lazy(t->st, false, true);
        t->st->lazyPu = t->st->lazyPh = nullptr; t->st->lazyOwner = nullptr;
    }
    (void)hipStreamSynchronize(t->st->ctx->stream);
    for (void *q : t->allocs) (void)hipFree(q);
    delete t;
}

int moka_step_fe_taped(moka_tape *t, double dt, int flags)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    moka_state *st = t->st;
    const Plan &p = st->mesh->plan;
    if (t->n >= t->capacity) return fail(st->ctx, MOKA_ERR_ARG, "tape is full");
    if ((flags & MOKA_FE_LEVEL1_ONLY) && p.K != 1)
        return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "reverse mode: level-1-only stepping is supported for nVertLevels = 1 only");
    if (t->n > 0 && t->kind != 0) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "reverse mode: one integrator per tape");
    t->kind = 0;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    int rc = flush_lazy(st, true, true);
    if (rc) return rc;
    hipStream_t s = st->ctx->stream;
    const size_t nEK = (size_t)p.K * p.nE, bytes = nEK * sizeof(double);
    HIPCHK(st->ctx, hipMemcpyAsync(t->uTape + nEK * t->n, st->lev[1].u, bytes, hipMemcpyDeviceToDevice, s));
    const bool stale = flags & MOKA_FE_STALE_HEDGE;
    if (stale) HIPCHK(st->ctx, hipMemcpyAsync(t->hTape + nEK * t->n, st->hEdge[0], bytes, hipMemcpyDeviceToDevice, s));
    st->feForceEager = true;            // the tape copies DiagnosticVars around the step: it stores all of its arrays
    rc = moka_step_fe(st, dt, flags);
    st->feForceEager = false;
    if (rc) return rc;
    // a refreshed layerThicknessEdge (= interp of the pre-step thickness) is what the flux used: it is Diag's after the step
    if (!stale) HIPCHK(st->ctx, hipMemcpyAsync(t->hTape + nEK * t->n, st->hEdge[0], bytes, hipMemcpyDeviceToDevice, s));
    t->dts.push_back(dt);
    t->flags.push_back(flags);
    ++t->n;
    t->seeded = false;
    return MOKA_OK;
}

int moka_step_rk4_taped(moka_tape *t, double dt)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    moka_state *st = t->st;
    const Plan &p = st->mesh->plan;
    if (t->n >= t->capacity) return fail(st->ctx, MOKA_ERR_ARG, "tape is full");
    if (t->n > 0 && t->kind != 1) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "reverse mode: one integrator per tape");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    const size_t nEK = (size_t)p.K * p.nE, nCK = (size_t)p.K * p.nC;
    int rc = MOKA_OK;
    if (!t->rkU) {
        auto A = [&](double **q, size_t n) { if (rc == MOKA_OK) { void *d = nullptr; rc = tape_alloc(t, &d, n * sizeof(double)); *q = static_cast<double *>(d); } };
        A(&t->rkU, nEK * 4 * (size_t)t->capacity); A(&t->rkH, nCK * 4 * (size_t)t->capacity);
        A(&t->kbU, nEK); A(&t->kbH, nCK); A(&t->pbU, nEK); A(&t->pbH, nCK);
        if (rc) return rc;
    }
    const double *ssh0 = nullptr;       // like moka_step_rk4: pending lazy diagnostics / tendencies of the previous step are simply superseded
    if ((rc = rk4_begin(st, &ssh0))) return rc;
    hipStream_t s = st->ctx->stream;
    double *tu = t->rkU + nEK * 4 * t->n, *th = t->rkH + nCK * 4 * t->n;
    // The tape needs the provisional states P1..P4 the four tendencies are evaluated at.  P1 (the current level) is copied;
    // P2, P3 and P4 are written by stages 1, 2 and 3 straight into their tape slots and read from there by the next stage
    // (and, P4, by the lazily produced stage-4 tendencies: moka_state.lazyPu / lazyPh).
    auto slotU = [&](int i) { return tu + nEK * i; };
    auto slotH = [&](int i) { return th + nCK * i; };
    HIPCHK(st->ctx, hipMemcpyAsync(slotU(0), st->lev[1].u, nEK * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHK(st->ctx, hipMemcpyAsync(slotH(0), st->lev[1].h, nCK * sizeof(double), hipMemcpyDeviceToDevice, s));
    for (int sg = 1; sg <= 4; ++sg) {
        StageArgs g = rk4_stage_args(st, sg, dt, ssh0);
        if (sg >= 2) { g.pu = slotU(sg - 1); g.ph = slotH(sg - 1); }
        if (sg <= 3) { g.pu_out = slotU(sg); g.ph_out = slotH(sg); }
        HIPCHK(st->ctx, run_stage(st, g));
    }
    rk4_end(st);
    st->lazyPu = slotU(3); st->lazyPh = slotH(3); st->lazyOwner = t;
    t->kind = 1;
    t->dts.push_back(dt);
    t->flags.push_back(0);
    ++t->n;
    t->seeded = false;
    return MOKA_OK;
}

// Piecewise taping for a step the caller runs itself (the distributed RK4 step, whose stages are separated by halo
// exchanges): slot 0..3 of the step being recorded := (normalVelocity, layerThickness) of buffer set `what` (0 = the current
// level, 1..3 = the output of RK stage `what`), halo rows included; moka_tape_commit_rk4 closes the step.
int moka_tape_record_rk4(moka_tape *t, int slot, int what)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    moka_state *st = t->st;
    const Plan &p = st->mesh->plan;
    if (slot < 0 || slot > 3 || what < 0 || what > 3) return fail(st->ctx, MOKA_ERR_ARG, "slot and what must be 0..3");
    if (t->n >= t->capacity) return fail(st->ctx, MOKA_ERR_ARG, "tape is full");
    if (t->n > 0 && t->kind != 1) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "reverse mode: one integrator per tape");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    const size_t nEK = (size_t)p.K * p.nE, nCK = (size_t)p.K * p.nC;
    int rc = MOKA_OK;
    if (!t->rkU) {
        auto A = [&](double **q, size_t n) { if (rc == MOKA_OK) { void *d = nullptr; rc = tape_alloc(t, &d, n * sizeof(double)); *q = static_cast<double *>(d); } };
        A(&t->rkU, nEK * 4 * (size_t)t->capacity); A(&t->rkH, nCK * 4 * (size_t)t->capacity);
        A(&t->kbU, nEK); A(&t->kbH, nCK); A(&t->pbU, nEK); A(&t->pbH, nCK);
        if (rc) return rc;
    }
    if ((rc = ensure_rk_bufs(st))) return rc;
    const LevelBufs &o = rk4_stage_output(st, what);
    hipStream_t s = st->ctx->stream;
    // the rows may have been produced on either stream (boundary patches, halo unpack): the copy waits for both
    HIPCHK(st->ctx, hipEventRecord(st->ctx->evHalo, st->ctx->comm));
    HIPCHK(st->ctx, hipStreamWaitEvent(s, st->ctx->evHalo, 0));
    HIPCHK(st->ctx, hipMemcpyAsync(t->rkU + nEK * (4 * t->n + slot), o.u, nEK * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHK(st->ctx, hipMemcpyAsync(t->rkH + nCK * (4 * t->n + slot), o.h, nCK * sizeof(double), hipMemcpyDeviceToDevice, s));
    t->recMask |= 1 << slot;
    return MOKA_OK;
}

// The same for a Forward-Euler step the caller runs itself (moka_fe_dist_step): after = 0 before the step (normalVelocity and,
// with MOKA_FE_STALE_HEDGE, the carried layerThicknessEdge the flux is about to use), after = 1 behind it (without the flag:
// the refreshed layerThicknessEdge the flux used, Diag's after the step); moka_tape_commit_fe closes the step.
int moka_tape_record_fe(moka_tape *t, int flags, int after)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    moka_state *st = t->st;
    const Plan &p = st->mesh->plan;
    if (t->n >= t->capacity) return fail(st->ctx, MOKA_ERR_ARG, "tape is full");
    if ((flags & MOKA_FE_LEVEL1_ONLY) && p.K != 1)
        return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "reverse mode: level-1-only stepping is supported for nVertLevels = 1 only");
    if (t->n > 0 && t->kind != 0) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "reverse mode: one integrator per tape");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    hipStream_t s = st->ctx->stream;
    const size_t nEK = (size_t)p.K * p.nE, bytes = nEK * sizeof(double);
    const bool stale = flags & MOKA_FE_STALE_HEDGE;
    HIPCHK(st->ctx, hipEventRecord(st->ctx->evHalo, st->ctx->comm));       // rows produced on either stream
    HIPCHK(st->ctx, hipStreamWaitEvent(s, st->ctx->evHalo, 0));
    if (!after) {
        if (int rc = flush_lazy(st, true, true)) return rc;
        st->feForceEager = true;        // until moka_tape_commit_fe: the recorded step stores all of its arrays
        HIPCHK(st->ctx, hipMemcpyAsync(t->uTape + nEK * t->n, st->lev[1].u, bytes, hipMemcpyDeviceToDevice, s));
        if (stale) HIPCHK(st->ctx, hipMemcpyAsync(t->hTape + nEK * t->n, st->hEdge[0], bytes, hipMemcpyDeviceToDevice, s));
        t->recMask = 1;
    } else {
        if (t->recMask != 1) return fail(st->ctx, MOKA_ERR_ARG, "moka_tape_record_fe: record before the step first");
        if (!stale) HIPCHK(st->ctx, hipMemcpyAsync(t->hTape + nEK * t->n, st->hEdge[0], bytes, hipMemcpyDeviceToDevice, s));
        t->recMask = 3;
    }
    return MOKA_OK;
}

int moka_tape_commit_fe(moka_tape *t, double dt, int flags)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    if (t->recMask != 3) return fail(t->st->ctx, MOKA_ERR_ARG, "moka_tape_commit_fe: record before and after the step first");
    t->st->feForceEager = false;
    t->recMask = 0;
    t->kind = 0;
    t->dts.push_back(dt);
    t->flags.push_back(flags);
    ++t->n;
    t->seeded = false;
    return MOKA_OK;
}

int moka_tape_commit_rk4(moka_tape *t, double dt)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    if (t->recMask != 15) return fail(t->st->ctx, MOKA_ERR_ARG, "moka_tape_commit_rk4: record the four provisional states first");
    t->recMask = 0;
    t->kind = 1;
    t->dts.push_back(dt);
    t->flags.push_back(0);
    ++t->n;
    t->seeded = false;
    return MOKA_OK;
}

int moka_adjoint_seed_sum_sq_ssh(moka_tape *t)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    moka_state *st = t->st;
    const Plan &p = st->mesh->plan;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    hipStream_t s = st->ctx->stream;
    const size_t nEK = (size_t)p.K * p.nE, nCK = (size_t)p.K * p.nC;
    t->cur = 0;
    HIPCHK(st->ctx, hipMemsetAsync(t->lamU[0], 0, nEK * sizeof(double), s));
    HIPCHK(st->ctx, hipMemsetAsync(t->lamH[0], 0, nCK * sizeof(double), s));
    HIPCHK(st->ctx, hipMemsetAsync(t->lamE[0], 0, nEK * sizeof(double), s));
    if (t->kind == 1) {
        // RK4: ssh is recomputed from layerThickness inside every tendency, so the objective lives on h: every level gets 2 ssh
        for (int b = 0; b < 2; ++b) {   // the RK4 sweep never writes the ssh / layerThicknessEdge adjoints: both stay zero
            HIPCHK(st->ctx, hipMemsetAsync(t->lamS[b], 0, (size_t)p.nC * sizeof(double), s));
            HIPCHK(st->ctx, hipMemsetAsync(t->lamE[b], 0, nEK * sizeof(double), s));
        }
        HIPCHK(st->ctx, launch_bcast_rows(t->lamH[0], st->lev[1].ssh, 2.0, p.nC, p.K, s));
    } else {
        HIPCHK(st->ctx, launch_scale_copy(t->lamS[0], st->lev[1].ssh, 2.0, p.nC, s));      // d sum(ssh^2) = 2 ssh
    }
    t->seeded = true;
    return MOKA_OK;
}

// one stage (sg = 4, 3, 2, 1) of one RK4 step backwards (time_integration.jl:61-148 transposed):
//   kb4 = b4*X; Pb = T'(P4)^T kb4; acc = X + Pb;  for s = 3,2,1: kb = b_s*X + a_s*Pb; Pb = T'(P_s)^T kb; acc += Pb;  X = acc
// sg == 1 completes the step (the adjoint states swap, the step is popped).  Between two stages the host layer of a partitioned
// run exchanges the halo rows of the k-bar the next stage gathers from (rk4_reverse_fields).
static void rk4_reverse_fields(moka_tape *t, int sg, double **fU, double **fH)
{
    double *kU[2] = {t->kbU, t->pbU}, *kH[2] = {t->kbH, t->pbH};
    const int kc = (4 - sg) & 1;                      // stage 4 reads k-bar 0 (or X itself), 3 reads 1, 2 reads 0, 1 reads 1
    *fU = sg == 4 ? t->lamU[t->cur] : kU[kc];
    *fH = sg == 4 ? t->lamH[t->cur] : kH[kc];
}

// part < 0: every local entity; 0 / 1: the edges and cells of cell class 0 (boundary: what other ranks gather from) / 1 (interior)
static int rk4_reverse_stage(moka_tape *t, int sg, int part = -1)
{
    moka_state *st = t->st;
    const Plan &p = st->mesh->plan;
    hipStream_t s = st->ctx->stream;
    const size_t nEK = (size_t)p.K * p.nE, nCK = (size_t)p.K * p.nC;
    const int64_t i = t->n - 1;
    const double dt = t->dts[i];
    const double ca[3] = {dt / 2., dt / 2., dt}, cb[4] = {dt / 6., dt / 3., dt / 3., dt / 6.};
    const int in = t->cur, o = 1 - t->cur;
    const double *XU = t->lamU[in], *XH = t->lamH[in];
    double *accU = t->lamU[o], *accH = t->lamH[o];
    // every k-bar after the first, and the running sum X + Pb4 + Pb3 + ..., come out of the two transposed kernels themselves
    // (fused epilogues: AdjArgs.accOut / kNext), k-bar double-buffered because the current one is being gathered while the
    // next one is written
    double *kU[2] = {t->kbU, t->pbU}, *kH[2] = {t->kbH, t->pbH};
    // chunk kernels (even K <= 64, hexagon-width lists): stage 4 reads X scaled on the fly (kb4 is never stored) and
    // u*Fbar is recomputed by the cell kernel instead of travelling through memory
    const bool fused = moka::adj_fused_available(t->am, st->mesh->lpc);
    moka::AdjMesh am = t->am;
    if (part >= 0) {
        if (!fused) return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "reverse RK4 stages part by part need the chunk kernels (even 34 <= K <= 64, hexagon-width lists)");
        if ((int)p.classCellStart.size() < 3) return fail(st->ctx, MOKA_ERR_ARG, "the mesh has no cell classes (moka_mesh_desc.cellClass)");
        am.cBegin = p.classCellStart[part]; am.cCount = p.classCellStart[part + 1] - am.cBegin;
        am.eBegin = p.classEdgeStart[part]; am.eCount = p.classEdgeStart[part + 1] - am.eBegin;
    }
    if (sg == 4 && !fused) {
        HIPCHK(st->ctx, launch_scale_copy(kU[0], XU, cb[3], (int64_t)nEK, s));
        HIPCHK(st->ctx, launch_scale_copy(kH[0], XH, cb[3], (int64_t)nCK, s));
    }
    const int kc = (4 - sg) & 1;
    moka::AdjArgs a{};
    a.tt = 1; a.dt = 1.0;
    a.u = t->rkU + nEK * (4 * i + (sg - 1)); a.h = t->rkH + nCK * (4 * i + (sg - 1));
    a.lamU1 = kU[kc]; a.lamH1 = kH[kc];
    a.lamScale = 1.0; a.fuseE = fused ? 1 : 0;
    if (fused && sg == 4) { a.lamU1 = XU; a.lamH1 = XH; a.lamScale = cb[3]; }
    a.Enew = t->Enew; a.csum = t->csum;
    a.xU = XU; a.xH = XH;
    a.accInU = sg == 4 ? nullptr : accU; a.accInH = sg == 4 ? nullptr : accH;
    a.accOutU = accU; a.accOutH = accH;
    if (sg > 1) {
        a.kNextU = kU[1 - kc]; a.kNextH = kH[1 - kc];
        a.cbNext = cb[sg - 2]; a.caNext = ca[sg - 2];
    }
    HIPCHK(st->ctx, launch_adj_edge(am, a, st->mesh->lpc, s));     // the cell kernel reads csum of the cells' edges: written by the edge
    HIPCHK(st->ctx, launch_adj_cell(am, a, st->mesh->lpc, s));     // kernel of the same part or (interior cells) of the boundary part
    if (part == 0) return MOKA_OK;                                 // the step's bookkeeping moves with the last part
    if (sg == 1) {
        t->cur = o;
        t->dts.pop_back(); t->flags.pop_back();
        --t->n;
    }
    return MOKA_OK;
}

// one Forward-Euler step backwards (the transposition of oracle_step_fe: see oracle_step_fe_adjoint)
static int fe_reverse_step(moka_tape *t)
{
    moka_state *st = t->st;
    const Plan &p = st->mesh->plan;
    hipStream_t s = st->ctx->stream;
    const size_t nEK = (size_t)p.K * p.nE;
    const int64_t i = t->n - 1;
    const int in = t->cur, o = 1 - t->cur;
    moka::AdjArgs a{};
    a.dt = t->dts[i];
    a.stale = (t->flags[i] & MOKA_FE_STALE_HEDGE) ? 1 : 0;
    a.u = t->uTape + nEK * i; a.hEuse = t->hTape + nEK * i;
    a.lamU1 = t->lamU[in]; a.lamH1 = t->lamH[in]; a.lamS1 = t->lamS[in]; a.lamE1 = t->lamE[in];
    a.lamU0 = t->lamU[o]; a.lamH0 = t->lamH[o]; a.lamS0 = t->lamS[o];
    a.Enew = a.stale ? t->lamE[o] : t->Enew;        // stale: u*Fbar IS the adjoint of the carried hEdge
    a.csum = t->csum;
    HIPCHK(st->ctx, launch_adj_edge(t->am, a, st->mesh->lpc, s));
    HIPCHK(st->ctx, launch_adj_cell(t->am, a, st->mesh->lpc, s));
    if (!a.stale) HIPCHK(st->ctx, hipMemsetAsync(t->lamE[o], 0, nEK * sizeof(double), s));
    t->cur = o;
    t->dts.pop_back(); t->flags.pop_back();
    --t->n;
    return MOKA_OK;
}

int moka_adjoint_fe_step_fields(moka_tape *t, void **fieldU, void **fieldH, void **fieldS)
{
    if (!t || !fieldU || !fieldH || !fieldS) return fail(t ? t->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    if (t->kind != 0 || t->n <= 0) return fail(t->st->ctx, MOKA_ERR_ARG, "no recorded Forward-Euler step");
    *fieldU = t->lamU[t->cur]; *fieldH = t->lamH[t->cur]; *fieldS = t->lamS[t->cur];
    return MOKA_OK;
}

int moka_adjoint_fe_step(moka_tape *t)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    moka_state *st = t->st;
    if (!t->seeded) return fail(st->ctx, MOKA_ERR_ARG, "seed the adjoint first (moka_adjoint_seed_sum_sq_ssh)");
    if (t->kind != 0 || t->n <= 0) return fail(st->ctx, MOKA_ERR_ARG, "no recorded Forward-Euler step");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    return fe_reverse_step(t);
}

int moka_adjoint_rk4_stage_fields(moka_tape *t, int sg, void **fieldU, void **fieldH, void **scratchS)
{
    if (!t || !fieldU || !fieldH) return fail(t ? t->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    if (sg < 1 || sg > 4 || t->kind != 1 || t->n <= 0) return fail(t->st->ctx, MOKA_ERR_ARG, "no recorded RK4 step / stage must be 4..1");
    double *fU, *fH;
    rk4_reverse_fields(t, sg, &fU, &fH);
    *fieldU = fU; *fieldH = fH;
    if (scratchS) *scratchS = t->lamS[1 - t->cur];     // an (nCells) array nothing reads: stands in for ssh in the exchange maps
    return MOKA_OK;
}

int moka_adjoint_rk4_stage(moka_tape *t, int sg)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    moka_state *st = t->st;
    if (!t->seeded) return fail(st->ctx, MOKA_ERR_ARG, "seed the adjoint first (moka_adjoint_seed_sum_sq_ssh)");
    if (sg < 1 || sg > 4 || t->kind != 1 || t->n <= 0) return fail(st->ctx, MOKA_ERR_ARG, "no recorded RK4 step / stage must be 4..1");
    if (sg != t->revNext || t->revPart != 0) return fail(st->ctx, MOKA_ERR_ARG, "reverse RK4 stages run 4, 3, 2, 1");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (int rc = rk4_reverse_stage(t, sg)) return rc;
    t->revNext = sg == 1 ? 4 : sg - 1;
    return MOKA_OK;
}

// The same stage in two launches on a partitioned mesh: part 0 = the entities of the boundary class (whose rows other ranks
// gather from in the next transposed stage), part 1 = the interior class.  Between the two the caller starts the exchange of
// the rows part 0 produced (moka_adjoint_rk4_stage_out_fields + moka_halo_pack_fields); it then overlaps part 1, which reads and
// writes owned rows only.  Halo entities are not computed at all (moka_adjoint_rk4_stage computes them redundantly).
int moka_adjoint_rk4_stage_part(moka_tape *t, int sg, int part)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    moka_state *st = t->st;
    if (!t->seeded) return fail(st->ctx, MOKA_ERR_ARG, "seed the adjoint first (moka_adjoint_seed_sum_sq_ssh)");
    if (sg < 1 || sg > 4 || t->kind != 1 || t->n <= 0) return fail(st->ctx, MOKA_ERR_ARG, "no recorded RK4 step / stage must be 4..1");
    if (part < 0 || part > 1) return fail(st->ctx, MOKA_ERR_ARG, "part must be 0 (boundary class) or 1 (interior class)");
    if (sg != t->revNext || part != t->revPart) return fail(st->ctx, MOKA_ERR_ARG, "reverse RK4 stages run 4, 3, 2, 1, each as part 0 then part 1");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (int rc = rk4_reverse_stage(t, sg, part)) return rc;
    t->revPart = 1 - part;
    if (part == 1) t->revNext = sg == 1 ? 4 : sg - 1;
    return MOKA_OK;
}

// 1: this tape's transposed RK4 stages can run class by class (the chunk kernels serve the mesh and the mesh has cell classes)
int moka_adjoint_rk4_parts_available(const moka_tape *t)
{
    if (!t) return 0;
    const moka_state *st = t->st;
    return moka::adj_fused_available(t->am, st->mesh->lpc) && (int)st->mesh->plan.classCellStart.size() >= 3 ? 1 : 0;
}

// The arrays stage `sg` of the reverse step in progress WRITES and the next transposed stage gathers from (stage sg - 1's k-bar;
// after stage 1: the adjoint state handed to the previous recorded step): what has to be exchanged behind part 0 of the stage.
int moka_adjoint_rk4_stage_out_fields(moka_tape *t, int sg, void **fieldU, void **fieldH, void **scratchS)
{
    if (!t || !fieldU || !fieldH) return fail(t ? t->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    if (sg < 1 || sg > 4 || t->kind != 1 || t->n <= 0) return fail(t->st->ctx, MOKA_ERR_ARG, "no recorded RK4 step / stage must be 4..1");
    double *kU[2] = {t->kbU, t->pbU}, *kH[2] = {t->kbH, t->pbH};
    const int kc = (4 - sg) & 1, o = 1 - t->cur;
    *fieldU = sg == 1 ? t->lamU[o] : kU[1 - kc];
    *fieldH = sg == 1 ? t->lamH[o] : kH[1 - kc];
    if (scratchS) *scratchS = t->lamS[o];              // an (nCells) array nothing reads here: stands in for ssh in the exchange maps
    return MOKA_OK;
}

int moka_adjoint_sweep(moka_tape *t)
{
    if (!t) return fail(nullptr, MOKA_ERR_ARG, "tape is NULL");
    moka_state *st = t->st;
    if (!t->seeded) return fail(st->ctx, MOKA_ERR_ARG, "seed the adjoint first (moka_adjoint_seed_sum_sq_ssh)");
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (t->kind == 1 && t->revNext != 4) return fail(st->ctx, MOKA_ERR_ARG, "a stage-wise reverse step is in progress");
    while (t->n > 0 && t->kind == 1)
        for (int sg = 4; sg >= 1; --sg)
            if (int rc = rk4_reverse_stage(t, sg)) return rc;
    while (t->n > 0)
        if (int rc = fe_reverse_step(t)) return rc;
    return MOKA_OK;
}

int moka_adjoint_download(moka_tape *t, int field, double *host)
{
    if (!t || !host) return fail(t ? t->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    moka_state *st = t->st;
    const Plan &p = st->mesh->plan;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    switch (field) {
        case MOKA_F_SSH: return get_rows(st->mesh, host, t->lamS[t->cur], MOKA_CELL, p.nC, 1);
        case MOKA_F_NORMAL_VELOCITY: return get_rows(st->mesh, host, t->lamU[t->cur], MOKA_EDGE, p.nE, p.K);
        case MOKA_F_LAYER_THICKNESS: return get_rows(st->mesh, host, t->lamH[t->cur], MOKA_CELL, p.nC, p.K);
        case MOKA_F_LAYER_THICKNESS_EDGE: return get_rows(st->mesh, host, t->lamE[t->cur], MOKA_EDGE, p.nE, p.K);
        default: return fail(st->ctx, MOKA_ERR_ARG, "adjoint fields: ssh, normalVelocity, layerThickness, layerThicknessEdge");
    }
}

}  // extern "C"
