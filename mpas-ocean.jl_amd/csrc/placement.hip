// placement.hip -- moka_state_optimize_placement: where the allocator put a state's arrays decides 5-14 % of every stage launch
// ("Run-to-run spread": a stable property of the memory behind the arrays, invisible to copy / read / gather
// probes and not steerable through the addresses this side of the allocator sees).  The caller of the C ABI -- the Julia shim
// binds PrognosticVars to a library state once (src/driver/mpas_ocean.jl:28-39 only ever calls ocn_init and the step) -- cannot
// try several states itself, so the library does it, and per FIELD: the four launches of an RK4 step are timed with the library's
// own events (dt = 0: the same loads and stores, nothing of the caller's state changes that is not restored), then one array at a
// time is given a new allocation while the old one is still alive (so the new one lands elsewhere), the launches the array takes
// part in are timed again and the faster of the two allocations is kept.  Peak extra memory: the previous level's arrays (saved
// and restored around the measurement) + the candidate + the rejected allocations held back so that the allocator cannot hand
// them out again (bounded by a share of the free memory) -- not a multiple of the whole state.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "state.hpp"

using namespace mk;

namespace {

struct FieldSlot {
    int id;              // moka_placement_trial.field: buffer set * 3 + (0 = normalVelocity, 1 = layerThickness, 2 = ssh)
    int set;             // 0 current level (A), 1 previous level / New accumulator (B), 2 / 3 the RK provisional states R1 / R2
    int which;           // 0 normalVelocity, 1 layerThickness, 2 ssh
    unsigned stages;     // bit s-1: the stage-s launch reads or writes the array (rk4_stage_args)
};
// A: gathered by stage 1, Curr of 2..4.  B: written by 1, read + written by 2..4.  R1: out of 1, in of 2, out of 3, in of 4.  R2: out of 2, in of 3.
// The ssh arrays are small (nCells reals) but every edge gathers two of their elements: they are candidates like the rows.
// (A.ssh is read by stage 1 only, B.ssh written by stage 4 only.)
const FieldSlot kSlots[12] = {{6, 2, 0, 0xF}, {9, 3, 0, 0x6}, {3, 1, 0, 0xF}, {0, 0, 0, 0xF},
                              {7, 2, 1, 0xF}, {10, 3, 1, 0x6}, {4, 1, 1, 0xF}, {1, 0, 1, 0xF},
                              {8, 2, 2, 0xF}, {11, 3, 2, 0x6}, {5, 1, 2, 0x8}, {2, 0, 2, 0x1}};
constexpr int NSLOTS = 12;

LevelBufs &set_of(moka_state *st, int set) { return set == 0 ? st->lev[1] : set == 1 ? st->lev[0] : st->rk[set - 2]; }

// every name the state has for `oldp` now names `newp` (lev[] / rk[] / spare and the by-allocation table a halo exports from)
void repoint(moka_state *st, double *oldp, double *newp)
{
    auto fix = [&](LevelBufs &b) {
        if (b.u == oldp) b.u = newp;
        if (b.h == oldp) b.h = newp;
        if (b.ssh == oldp) b.ssh = newp;
    };
    for (auto &b : st->lev) fix(b);
    for (auto &b : st->rk) fix(b);
    fix(st->spare);
    for (auto &b : st->phys) fix(b);
    for (void *&q : st->allocs) if (q == (void *)oldp) q = (void *)newp;
}

struct Timer {
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    ~Timer() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
};

// median over `reps` dt = 0 steps of each stage launch's duration (ms); one untimed step first
int time_stages(moka_state *st, Timer &T, int reps, double ms[4])
{
    moka_ctx *c = st->ctx;
    std::vector<float> v[4];
    for (int r = -1; r < reps; ++r) {
        for (int s = 1; s <= 4; ++s) {
            if (r >= 0) HIPCHK(c, hipEventRecord(T.ev[s - 1], c->stream));
            HIPCHK(c, run_stage(st, rk4_stage_args(st, s, 0.0, st->lev[1].ssh)));
            ++st->placementLaunches;
        }
        if (r < 0) continue;
        HIPCHK(c, hipEventRecord(T.ev[4], c->stream));
        HIPCHK(c, hipEventSynchronize(T.ev[4]));
        for (int s = 0; s < 4; ++s) {
            float t = 0.f;
            HIPCHK(c, hipEventElapsedTime(&t, T.ev[s], T.ev[s + 1]));
            v[s].push_back(t);
        }
    }
    for (int s = 0; s < 4; ++s) {
        std::sort(v[s].begin(), v[s].end());
        ms[s] = v[s][v[s].size() / 2];
    }
    return MOKA_OK;
}

double over(const double ms[4], unsigned stages)
{
    double t = 0.0;
    for (int s = 0; s < 4; ++s) if (stages >> s & 1u) t += ms[s];
    return t;
}

}  // namespace

extern "C" {

int moka_state_optimize_placement(moka_state *st, int max_tries, double *ms_before, double *ms_after)
{
    if (!st) return fail(nullptr, MOKA_ERR_ARG, "state is NULL");
    moka_ctx *c = st->ctx;
    if (ms_before) *ms_before = 0.0;
    if (ms_after) *ms_after = 0.0;
    if (st->attached > 0)
        return fail(c, MOKA_ERR_UNSUPPORTED, "moka_state_optimize_placement: a halo or a tape of this state exists (they hold or have exported "
                                            "the arrays' addresses); call it before moka_halo_create / moka_tape_create");
    HIPCHK(c, hipSetDevice(c->device));
    st->placementLog.clear();
    st->placementLaunches = 0;
    int rc = ensure_rk_bufs(st);
    if (rc) return rc;
    // what a finished RK4 or lean Forward-Euler step left pending is produced from arrays the measurement overwrites (the provisional
    // states) or moves: produce it now.  (DiagnosticVars of an fp32-storage state that are pending after an RK4 step stay pending;
    // they derive from the current level, which keeps its contents.)
    if ((rc = flush_lazy(st, !(st->f32 && st->diagDirty), true))) return rc;
    const Plan &p = st->mesh->plan;
    const size_t sb = st->f32 ? 4 : 8;
    const size_t bytesU = (size_t)p.K * p.nE * sb, bytesH = (size_t)p.K * p.nC * sb, bytesS = (size_t)p.nC * sb;
    hipStream_t s = c->stream;
    Timer T;
    for (hipEvent_t &e : T.ev) HIPCHK(c, hipEventCreate(&e));

    // the dt = 0 stages write Curr's values over the New accumulator = the previous time level: saved here, restored at the end
    void *save[3] = {nullptr, nullptr, nullptr};
    std::vector<std::pair<void *, size_t>> held;          // rejected / replaced allocations, kept so that they are not handed out again
    size_t heldBytes = 0;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
        for (void *q : save) if (q) (void)hipFree(q);
        for (auto &h : held) (void)hipFree(h.first);
    };
    const size_t saveBytes[3] = {bytesU, bytesH, bytesS};
    for (int i = 0; i < 3; ++i)
        if (hipMalloc(&save[i], std::max<size_t>(saveBytes[i], 16)) != hipSuccess) {
            (void)hipGetLastError();
            cleanup();
            return fail(c, MOKA_ERR_ALLOC, "moka_state_optimize_placement: no memory to save the previous time level");
        }
    {
        const LevelBufs &B = st->lev[0];
        hipError_t e = hipMemcpyAsync(save[0], B.u, bytesU, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(save[1], B.h, bytesH, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(save[2], B.ssh, bytesS, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { cleanup(); return fail(c, MOKA_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e)); }
    }
    auto restore = [&]() -> hipError_t {
        const LevelBufs &B = st->lev[0];
        hipError_t e = hipMemcpyAsync(B.u, save[0], bytesU, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(B.h, save[1], bytesH, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(B.ssh, save[2], bytesS, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        return e;
    };
    constexpr int REPS = 5;
    double best[4];
    // Warm-up first: the launches right after set-up run 1-2 % slower than the steady state (the package takes a few hundred
    // milliseconds to settle at its power limit), and a baseline taken then makes the FIRST trial look like a gain whatever it
    // moves (round 4, first version: "rk1.normalVelocity" won every search and the timed steps never showed it).  Measure until
    // two consecutive passes agree within 0.3 %, at most ten passes.
    {
        double prev = 0.0;
        for (int pass = 0; pass < 10; ++pass) {
            if ((rc = time_stages(st, T, REPS, best))) { (void)restore(); cleanup(); return rc; }
            const double now = over(best, 0xF);
            if (pass > 0 && std::fabs(now - prev) <= 0.003 * prev) break;
            prev = now;
        }
    }
    const double t0 = over(best, 0xF);
    if (ms_before) *ms_before = t0;

    int sinceGain = 0;
    for (int tr = 0; tr < max_tries && sinceGain < NSLOTS; ++tr) {
        const FieldSlot &f = kSlots[tr % NSLOTS];
        LevelBufs &set = set_of(st, f.set);
        double *oldp = f.which == 0 ? set.u : f.which == 1 ? set.h : set.ssh;
        const size_t bytes = f.which == 0 ? bytesU : f.which == 1 ? bytesH : bytesS;
        // hold back at most a quarter of what is free now (and never less than room for this candidate)
        size_t freeB = 0, totalB = 0;
        (void)hipMemGetInfo(&freeB, &totalB);
        while (!held.empty() && (heldBytes > freeB / 4 || freeB < 2 * bytes)) {
            (void)hipFree(held.front().first);
            heldBytes -= held.front().second;
            held.erase(held.begin());
            (void)hipMemGetInfo(&freeB, &totalB);
        }
        void *cand = nullptr;
        if (hipMalloc(&cand, std::max<size_t>(bytes, 16)) != hipSuccess) {
            (void)hipGetLastError();
            break;                                       // no room for a candidate: keep what we have
        }
        hipError_t e = hipMemcpyAsync(cand, oldp, bytes, hipMemcpyDeviceToDevice, s);   // (the current level's contents must survive)
        if (e != hipSuccess) { (void)hipFree(cand); (void)restore(); cleanup(); return fail(c, MOKA_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e)); }
        repoint(st, oldp, static_cast<double *>(cand));
        double ms[4];
        if ((rc = time_stages(st, T, REPS, ms))) {
            repoint(st, static_cast<double *>(cand), oldp);
            (void)hipStreamSynchronize(s);
            (void)hipFree(cand);
            (void)restore();
            cleanup();
            return rc;
        }
        double was = over(best, f.stages);
        const double now = over(ms, f.stages);
        // keep the candidate when the launches it takes part in got faster by more than the repeatability of the medians -- and
        // only after the INCUMBENT has been timed once more right behind it (A / B / A): a drift of the box between the two
        // measurements must not pass for a better placement
        bool keep = now < was * (1.0 - 0.004);
        if (keep) {
            double again[4];
            repoint(st, static_cast<double *>(cand), oldp);
            rc = time_stages(st, T, REPS, again);
            repoint(st, oldp, static_cast<double *>(cand));
            if (rc) {
                repoint(st, static_cast<double *>(cand), oldp);
                (void)hipStreamSynchronize(s);
                (void)hipFree(cand);
                (void)restore();
                cleanup();
                return rc;
            }
            for (int k = 0; k < 4; ++k) if (f.stages >> k & 1u) best[k] = std::min(best[k], again[k]);
            was = over(best, f.stages);
            keep = now < was * (1.0 - 0.004);
        }
        moka_placement_trial log{f.id, was, now, keep ? 1 : 0};
        st->placementLog.push_back(log);
        void *loser = keep ? (void *)oldp : cand;
        if (keep) {
            for (int k = 0; k < 4; ++k) if (f.stages >> k & 1u) best[k] = ms[k];
            sinceGain = 0;
        } else {
            // back to the old array: what the trial stored went to the candidate, but those were arrays every step writes before it
            // reads them (B is restored below, R1 / R2 are scratch); the current level (A) is never written
            repoint(st, static_cast<double *>(cand), oldp);
            ++sinceGain;
        }
        held.push_back({loser, bytes});
        heldBytes += bytes;
    }
    hipError_t er = restore();
    cleanup();
    if (er != hipSuccess) return fail(c, MOKA_ERR_HIP, std::string("restoring the previous time level: ") + hipGetErrorString(er));
    if (ms_after) *ms_after = over(best, 0xF);
    return MOKA_OK;
}

int64_t moka_state_placement_launches(const moka_state *st) { return st ? st->placementLaunches : 0; }

int moka_state_placement_log(const moka_state *st, int32_t capacity, moka_placement_trial *out, int32_t *n)
{
    if (!st || !n) return fail(st ? st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    *n = (int32_t)st->placementLog.size();
    for (int32_t i = 0; out && i < capacity && i < *n; ++i) out[i] = st->placementLog[i];
    return MOKA_OK;
}

}  // extern "C"
