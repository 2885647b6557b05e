// halo.hip -- multi-GPU layer of libmoka_hip (the reference has no distributed code at all).
//
// One process per GPU; the mesh of a state is the rank's LOCAL mesh (owned cells + a one-cell-deep halo).  Cell classes
// (moka_mesh_desc.cellClass): 0 = owned and needed by another rank, 1 = owned interior, 2 + i = halo cells owned by the
// rank's i-th neighbour.  The plan numbers cells class-major and never lets a patch straddle a class, so
//   * patches [0, pBoundary) produce everything other ranks need, [pBoundary, pOwned) is the interior, halo patches are
//     never computed,
//   * what neighbour i sends lands in ONE contiguous range of cells and ONE of edges (edges are numbered by owner cell).
//
// Two transports for the rows [h | ssh | u] of the cells / edges a neighbour needs, once per RK stage:
//   buffered : moka_halo_pack -> one contiguous send buffer -> the host layer moves it (torch.distributed on RCCL over
//              xGMI, gloo in tests, MPI.jl from Julia) -> moka_halo_unpack.
//   direct   : the sender's push kernel stores the rows straight into the receiver's fields over xGMI (peer-mapped
//              memory: hipIpcOpenMemHandle between processes, plain pointers inside one process) -- no send buffer, no
//              unpack, no collective library.  Completion is signalled through flag words in host shared memory: the
//              sender's host thread waits for its push kernel (one event) and then stores the exchange's sequence number
//              into each neighbour's flag slot; the receiver's host thread polls its slots before it launches the next
//              stage's boundary patches.  All of that happens while the interior patches of the stage run on the compute
//              stream, so the host never sits on the critical path unless the exchange is later than the interior launch.
//              No kernel ever spins on the device (a spinning kernel can starve whatever shares its hardware queue).
// Why the direct form needs no back-pressure: a rank signals exchange n only when its push kernel of exchange n is done, and
// that kernel is queued behind EVERY launch of this rank that reads halo rows of the buffer set a neighbour overwrites next:
//   RK4: halo rows are read only by boundary launches; the rows a neighbour overwrites at stage s+1 (its output set) were
//        last read by my boundary launch of an earlier stage, which precedes my push of that stage;
//   Forward Euler: besides the boundary launch, the vertex pass (relativeVorticity over every local vertex) reads
//        old-level normalVelocity rows of halo edges -- the very buffer the neighbours' NEXT step pushes into.  It is
//        therefore launched FIRST, ahead of the boundary launch and the push (moka_fe_dist_step; round 3: launched behind the
//        interior patches it could still be reading when a neighbour that was a step ahead overwrote those rows).
// A neighbour has waited for that flag before it computed what it now pushes.
//   Lean Forward-Euler steps (moka_state.feLazy) with a connected direct halo (moka_state.feLeanInteriorOnly): the arrays a lean
//        step leaves pending are produced on first read from the level before the previous one -- the spare set -- and the halo
//        rows of that set are what a neighbour's NEXT step pushes its new level into (it needs only this rank's flag of the
//        current step for that, long sent when a caller reads diagnostics between steps).  Only BOUNDARY patches read halo rows.
//        So the boundary launch of such a step stores every array of its patches at once (in place: a lean step's launches read
//        no stored layerThicknessEdge), ahead of this rank's own push, and only the INTERIOR patches' arrays stay pending
//        (moka_state.feLazyBegin / feLazyCount): their stencils reach owned rows only -- a cell next to a halo cell is a boundary
//        cell, and an edge between a boundary cell and a halo cell is owned by the boundary cell (round 4a had
//        switched lean steps off altogether for such states: 2.85 instead of 1.65 ms per step at config 4).
//
// Visibility of peer-written rows (the fields are ordinary, coarse-grained hipMalloc memory): a push kernel ends with
// __threadfence_system() and its completion event precedes the flag store (release) the reader's host thread acquires
// before it LAUNCHES the kernel that reads the rows.  Peer stores arrive through the fabric at the owner's memory, not
// through the owner's L2s; a line of a halo row that an earlier launch left in one of the owner's eight L2s is dropped by
// the acquire every kernel dispatch performs (the same mechanism that lets launch N + 1 on XCD j read what launch N stored
// through XCD i's L2 -- the per-XCD L2s are not coherent with each other either, MI355X_MICROARCH.md "Correctness
// boundaries").  For a platform where that should not hold, moka_halo_set_acquire(h, 1) puts an explicit system-scope
// acquire (a grid of small workgroups across all XCDs, each executing a system-scope fence = cache invalidate) in front of
// every launch that reads received rows; moka_hip.parallel.choose_transport qualifies "ipc" by comparing whole steps with the
// host-staged exchange bit for bit and tries "ipc-acq" (the same transport with that fence) when the plain form fails.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cerrno>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "state.hpp"

using namespace mk;

namespace {

struct PushDst {              // where the rows for one neighbour go, for one physical buffer set of the peer
    unsigned char *u, *h, *ssh;
};

struct PeerLink {
    bool connected = false, ipc = false;
    void *mapped[15] = {};                // peer field bases as this process sees them: [set][u, h, ssh]
    volatile uint64_t *flags = nullptr;   // the peer's flag block (host memory)
    uint64_t *flagsDev = nullptr;         // ... as the device sees it (hipHostRegister), for stream memory operations
    size_t flagsBytes = 0;
    int32_t dstCell = 0, dstEdge = 0, slot = 0;
};

}  // namespace

struct moka_halo {
    moka_state *st = nullptr;
    int32_t nNbr = 0;
    // buffered transport
    uint32_t *sendMap = nullptr, *recvMap = nullptr;     // element maps, device
    int64_t nSend = 0, nRecv = 0;                        // elements
    int32_t pBoundary = 0, pOwned = 0;
    int32_t pFirst = 0;                                  // the first launch of a stage covers patches [0, pFirst), see below
    double dt = 0.0;
    const double *ssh0 = nullptr;
    int feFlags = 0;
    bool feStageKernel = false, fePrev = false;   // the running distributed Forward-Euler step: stage kernel? mode 6?
    // direct transport
    bool directOk = false;
    std::string directWhy;
    std::vector<int32_t> recvCellStart, recvCellCount, recvEdgeStart, recvEdgeCount;   // per neighbour, library numbering
    std::vector<int32_t> sendCellCount, sendEdgeCount;
    uint32_t *pushRows = nullptr;         // device: {source row (library numbering), row within the message part, nbr | kind << 16}
    int64_t nPushRows = 0;
    std::vector<PeerLink> peers;
    PushDst *peerTab = nullptr;           // device: [5 sets][nNbr]
    bool tabDirty = true;
    volatile uint64_t *flags = nullptr;   // my flag block: slot i = last exchange neighbour i has completed towards me
    uint64_t *flagsDev = nullptr;         // ... as the device sees it (stream memory operations)
    bool streamFlags = false;             // moka_halo_set_stream_flags: the handshake is enqueued, the host neither waits nor polls
    size_t flagsBytes = 0;
    std::string shmName;                  // non-empty: the block is a POSIX shared-memory object (multi-process)
    uint64_t seq = 0;                     // exchanges started so far
    hipEvent_t evPush = nullptr;
    hipEvent_t evB[2] = {nullptr, nullptr};   // boundary launch of an odd / even stage done (comm stream)
    bool acquireFence = false;                // moka_halo_set_acquire: system-scope acquire in front of launches that read received rows
    bool overlapB = false;                    // boundary patches on the comm stream, in flight together with the interior launch
    bool overlapNow = false;                  // ... as the running step was begun
    std::vector<void *> allocs;           // device allocations of this object
    bool counted = false;                 // st->attached includes this object
    // measurement (moka_halo_stats_enable / _read): where a distributed step spends its time on this rank
    bool statsOn = false;
    double stSignalWaitMs = 0.0, stFlagStoreMs = 0.0, stWaitMs = 0.0, stStepHostMs = 0.0;   // host clocks, summed
    int64_t stExchanges = 0, stSteps = 0;
    std::vector<hipEvent_t> stEv;         // timing events around the boundary / interior launches: 4 per recorded stage
    size_t stEvUsed = 0;
    std::vector<int> stEvPart;            // part (0 boundary, 1 interior) of every recorded pair
};

namespace {

int hfail(moka_halo *h, int code, const std::string &msg) { return fail(h ? h->st->ctx : nullptr, code, msg); }

// Element map of one direction (buffered transport).  Per neighbour i the buffer segment is
//   [h rows of cells[co[i]..co[i+1]) | ssh of the same cells | u rows of edges[eo[i]..eo[i+1])]   (one message)
int build_halo_map(moka_halo *hh, int nNbr, const int32_t *cells, const int64_t *co, const int32_t *edges,
                   const int64_t *eo, uint32_t **outDev, int64_t *outN)
{
    moka_state *st = hh->st;
    const Plan &p = st->mesh->plan;
    const int K = p.K;
    if ((int64_t)p.K * std::max(p.nE, p.nC) >= (1ll << 30))
        return fail(st->ctx, MOKA_ERR_UNSUPPORTED, "halo element map: a local field has more than 2^30 elements");
    std::vector<uint32_t> map;
    map.reserve((size_t)(co[nNbr] * (K + 1) + eo[nNbr] * K));
    for (int i = 0; i < nNbr; ++i) {
        for (int64_t j = co[i]; j < co[i + 1]; ++j) {
            if (cells[j] < 0 || cells[j] >= p.nC) return fail(st->ctx, MOKA_ERR_ARG, "halo cell id out of range");
            const uint32_t base = (uint32_t)p.cellO2N[cells[j]] * (uint32_t)K;
            for (int k = 0; k < K; ++k) map.push_back((0u << 30) | (base + k));
        }
        for (int64_t j = co[i]; j < co[i + 1]; ++j) map.push_back((1u << 30) | (uint32_t)p.cellO2N[cells[j]]);
        for (int64_t j = eo[i]; j < eo[i + 1]; ++j) {
            if (edges[j] < 0 || edges[j] >= p.nE) return fail(st->ctx, MOKA_ERR_ARG, "halo edge id out of range");
            const uint32_t base = (uint32_t)p.edgeO2N[edges[j]] * (uint32_t)K;
            for (int k = 0; k < K; ++k) map.push_back((2u << 30) | (base + k));
        }
    }
    *outN = (int64_t)map.size();
    *outDev = nullptr;
    if (map.empty()) return MOKA_OK;
    void *d = nullptr;
    HIPCHK(st->ctx, hipMalloc(&d, map.size() * sizeof(uint32_t)));
    hh->allocs.push_back(d);
    if (int rc = h2d(st->ctx, d, map.data(), map.size() * sizeof(uint32_t))) return rc;
    *outDev = static_cast<uint32_t *>(d);
    return MOKA_OK;
}

// index (0..4) of the physical buffer set a LevelBufs currently names: 0/1 the two time levels as allocated, 2/3 the RK
// provisional states, 4 the Forward-Euler spare.  Ranks that have applied the same sequence of steps agree on it.
int phys_index(const moka_state *st, const LevelBufs &b)
{
    for (int i = 0; i < moka_state::NPHYS; ++i)
        if (st->phys[i].u && st->phys[i].u == b.u) return i;
    return -1;
}

// One half-wave (32 lanes) per row: 16 bytes per lane when the row size allows it, else one element per lane.
__global__ __launch_bounds__(256) void k_halo_push(const uint32_t *rows, int64_t nRows, const PushDst *tab, const unsigned char *u,
                                                   const unsigned char *h, const unsigned char *ssh, uint32_t rowB, uint32_t elemB)
{
    const int l = threadIdx.x & 31;
    const int64_t g0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5, ng = ((int64_t)gridDim.x * 256) >> 5;
    for (int64_t r = g0; r < nRows; r += ng) {
        const uint32_t src = rows[3 * r], dst = rows[3 * r + 1], meta = rows[3 * r + 2];
        const uint32_t nbr = meta & 0xFFFFu, kind = meta >> 16;          // 0 = h row, 1 = ssh element, 2 = u row
        const PushDst d = tab[nbr];
        if (kind == 1) {
            if (l == 0) {
                if (elemB == 8) reinterpret_cast<double *>(d.ssh)[dst] = reinterpret_cast<const double *>(ssh)[src];
                else reinterpret_cast<float *>(d.ssh)[dst] = reinterpret_cast<const float *>(ssh)[src];
            }
            continue;
        }
        const unsigned char *s = (kind == 0 ? h : u) + (size_t)src * rowB;
        unsigned char *t = (kind == 0 ? d.h : d.u) + (size_t)dst * rowB;
        if ((rowB & 15u) == 0) {
            for (uint32_t off = (uint32_t)l * 16u; off < rowB; off += 512u)
                *reinterpret_cast<uint4 *>(t + off) = *reinterpret_cast<const uint4 *>(s + off);
        } else if (elemB == 8) {
            for (uint32_t off = (uint32_t)l * 8u; off < rowB; off += 256u)
                *reinterpret_cast<double *>(t + off) = *reinterpret_cast<const double *>(s + off);
        } else {
            for (uint32_t off = (uint32_t)l * 4u; off < rowB; off += 128u)
                *reinterpret_cast<float *>(t + off) = *reinterpret_cast<const float *>(s + off);
        }
    }
    __threadfence_system();          // the rows are in the peer's memory before the kernel (hence its event) completes
}

// System-scope acquire on every XCD: 64 single-wave workgroups (the dispatcher deals consecutive workgroups round-robin over
// the eight XCDs), each invalidating the caches it sits behind.  Nothing is loaded: the launch that follows does that.
__global__ __launch_bounds__(64) void k_halo_acquire(uint32_t *sink)
{
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                      // system scope: buffer_inv sc0 sc1
    if (sink && threadIdx.x == 0xFFFF) *sink = blockIdx.x;        // (never true: keeps the kernel from being empty)
}

int launch_acquire(moka_halo *h, hipStream_t s)
{
    if (!h->acquireFence) return MOKA_OK;
    hipLaunchKernelGGL(k_halo_acquire, dim3(64), dim3(64), 0, s, (uint32_t *)nullptr);
    HIPCHK(h->st->ctx, hipGetLastError());
    return MOKA_OK;
}

int upload_peer_tab(moka_halo *h)
{
    if (!h->tabDirty) return MOKA_OK;
    const moka_state *st = h->st;
    const Plan &p = st->mesh->plan;
    const size_t sb = st->f32 ? 4 : 8, rowB = (size_t)p.K * sb;
    std::vector<PushDst> tab((size_t)moka_state::NPHYS * std::max(h->nNbr, 1));
    for (int t = 0; t < moka_state::NPHYS; ++t)
        for (int i = 0; i < h->nNbr; ++i) {
            const PeerLink &pl = h->peers[i];
            PushDst d{};
            d.u = static_cast<unsigned char *>(pl.mapped[3 * t + 0]) + (size_t)pl.dstEdge * rowB;
            d.h = static_cast<unsigned char *>(pl.mapped[3 * t + 1]) + (size_t)pl.dstCell * rowB;
            d.ssh = static_cast<unsigned char *>(pl.mapped[3 * t + 2]) + (size_t)pl.dstCell * sb;
            tab[(size_t)t * h->nNbr + i] = d;
        }
    if (!h->peerTab) {
        void *d = nullptr;
        HIPCHK(st->ctx, hipMalloc(&d, tab.size() * sizeof(PushDst)));
        h->allocs.push_back(d);
        h->peerTab = static_cast<PushDst *>(d);
    }
    if (int rc = h2d(st->ctx, h->peerTab, tab.data(), tab.size() * sizeof(PushDst))) return rc;
    h->tabDirty = false;
    return MOKA_OK;
}

bool all_connected(const moka_halo *h)
{
    for (const PeerLink &pl : h->peers)
        if (!pl.connected) return false;
    return h->directOk;
}

int ensure_flags(moka_halo *h, bool shared)
{
    if (h->flags) return MOKA_OK;
    const size_t bytes = sizeof(uint64_t) * (size_t)std::max(h->nNbr, 1);
    if (shared) {
        char name[64];
        snprintf(name, sizeof name, "/moka_halo_%d_%llx", (int)getpid(), (unsigned long long)(uintptr_t)h);
        int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 && errno == EEXIST) {      // left behind by a killed process that had this pid and address: it is ours now
            shm_unlink(name);
            fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        }
        if (fd < 0) return hfail(h, MOKA_ERR_COMM, std::string("shm_open(") + name + ") failed");
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return hfail(h, MOKA_ERR_COMM, "ftruncate on the flag block failed"); }
        void *q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (q == MAP_FAILED) { shm_unlink(name); return hfail(h, MOKA_ERR_COMM, "mmap of the flag block failed"); }
        h->flags = static_cast<volatile uint64_t *>(q);
        h->shmName = name;
    } else {
        h->flags = static_cast<volatile uint64_t *>(calloc(1, bytes));
        if (!h->flags) return hfail(h, MOKA_ERR_ALLOC, "out of host memory");
    }
    h->flagsBytes = bytes;
    for (int i = 0; i < h->nNbr; ++i) h->flags[i] = 0;
    return MOKA_OK;
}

// the part of one stage that goes on the device queues: boundary patches, then (direct) the push on the comm stream or
// (buffered) nothing yet -- the caller packs --, then the interior patches
int dist_stage_part(moka_halo *h, int stage, int part)
{
    moka_state *st = h->st;
    const StageArgs g = rk4_stage_args(st, stage, h->dt, h->ssh0);
    const int p0 = part == 0 ? 0 : h->pFirst, cnt = part == 0 ? h->pFirst : h->pOwned - h->pFirst;
    moka_ctx *c = st->ctx;
    // measurement: one event in front of and one behind the launch, on the stream it goes to (at most 2048 launches recorded)
    hipStream_t on = (h->overlapNow && part == 0) ? c->comm : c->stream;
    const bool rec = h->statsOn && cnt > 0 && h->stEvPart.size() < 2048;
    auto stamp = [&]() -> hipError_t {
        if (h->stEvUsed == h->stEv.size()) {
            hipEvent_t e = nullptr;
            if (hipError_t er = hipEventCreate(&e); er != hipSuccess) return er;
            h->stEv.push_back(e);
        }
        return hipEventRecord(h->stEv[h->stEvUsed++], on);
    };
    struct Closer {                     // the closing event, whichever way the function returns
        bool on; decltype(stamp) &f; moka_halo *h; int part;
        ~Closer() { if (on && f() == hipSuccess) h->stEvPart.push_back(part); }
    } closer{false, stamp, h, part};
    if (rec) {
        if (h->overlapNow && part == 0) HIPCHK(c, hipStreamWaitEvent(c->comm, c->evInterior, 0));   // (not counted as launch time)
        HIPCHK(c, stamp());
        closer.on = true;
    }
    if (!h->overlapNow) {
        // boundary group first, interior right behind it on the same (compute) stream
        if (part == 0) if (int rc = launch_acquire(h, c->stream)) return rc;
        HIPCHK(c, run_stage(st, g, p0, cnt));
    } else if (part == 0) {
        // boundary patches on the (high-priority) comm stream, interior patches on the compute stream, in flight together:
        // B_s waits for I_(s-1) (it gathers rows of interior patches), I_s waits for B_(s-1), never for B_s
        HIPCHK(c, hipStreamWaitEvent(c->comm, c->evInterior, 0));
        if (int rc = launch_acquire(h, c->comm)) return rc;
        HIPCHK(c, run_stage(st, g, p0, cnt, c->comm));
        HIPCHK(c, hipEventRecord(h->evB[stage & 1], c->comm));
    } else {
        if (stage > 1) HIPCHK(c, hipStreamWaitEvent(c->stream, h->evB[(stage - 1) & 1], 0));
        HIPCHK(c, run_stage(st, g, p0, cnt));
        HIPCHK(c, hipEventRecord(c->evInterior, c->stream));
    }
    return MOKA_OK;
}

}  // namespace

extern "C" {

int moka_halo_create(moka_state *st, int32_t nNeighbors, const int32_t *sendCells, const int64_t *sendCellOff,
                     const int32_t *sendEdges, const int64_t *sendEdgeOff, const int32_t *recvCells,
                     const int64_t *recvCellOff, const int32_t *recvEdges, const int64_t *recvEdgeOff,
                     int32_t nPatchesBoundary, int32_t nPatchesOwned, moka_halo **out)
{
    if (!st || !out || nNeighbors < 0 || !sendCellOff || !sendEdgeOff || !recvCellOff || !recvEdgeOff)
        return fail(st ? st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    *out = nullptr;
    const Plan &p = st->mesh->plan;
    if (nNeighbors > 60) return fail(st->ctx, MOKA_ERR_ARG, "at most 60 neighbours");
    if (nPatchesBoundary < 0 || nPatchesOwned < nPatchesBoundary || nPatchesOwned > p.nPatches)
        return fail(st->ctx, MOKA_ERR_ARG, "patch ranges must satisfy 0 <= boundary <= owned <= nPatches");
    // the two ranges must be whole cell classes of the plan (patches never straddle a class)
    {
        const auto &cp = p.classPatchStart;
        const bool okB = std::find(cp.begin(), cp.end(), nPatchesBoundary) != cp.end();
        const bool okO = std::find(cp.begin(), cp.end(), nPatchesOwned) != cp.end();
        if (!okB || !okO)
            return fail(st->ctx, MOKA_ERR_ARG, "boundary / owned patch counts must be class boundaries of the mesh (moka_mesh_class_ranges)");
    }
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (int rc = ensure_rk_bufs(st)) return rc;     // the five physical buffer sets exist from here on: a neighbour
    if (int rc = ensure_spare(st)) return rc;       // addresses them by index when it pushes
    moka_halo *h = new (std::nothrow) moka_halo();
    if (!h) return fail(st->ctx, MOKA_ERR_ALLOC, "out of host memory");
    h->st = st;
    h->nNbr = nNeighbors;
    h->pBoundary = nPatchesBoundary; h->pOwned = nPatchesOwned;
    // The first launch of a stage holds exactly the boundary patches.  (Padding it with interior patches up to one full
    // generation of workgroups -- 4 per CU -- so that its ~30 us do more work was measured and is slower: 1024 workgroups
    // that start together also stage and gather together; 8-rank share of config 4 0.97 -> 1.04 ms per step.)
    h->pFirst = nPatchesBoundary;
    // Small interior launches (an 8-way share of config 4: 7 800 patches) run together with the boundary launch, which alone
    // fills an eighth of the chip for ~30 us; measured -6 % per step there, +3 % on a 2-way share (32 000 patches): see
    // moka_halo_set_overlap
    h->overlapB = nPatchesBoundary > 0 && nPatchesOwned - nPatchesBoundary < 12000;
    int rc = MOKA_OK;
    try {
        if ((rc = build_halo_map(h, nNeighbors, sendCells, sendCellOff, sendEdges, sendEdgeOff, &h->sendMap, &h->nSend)) ||
            (rc = build_halo_map(h, nNeighbors, recvCells, recvCellOff, recvEdges, recvEdgeOff, &h->recvMap, &h->nRecv))) {
            moka_halo_destroy(h);
            return rc;
        }
        // direct transport: every neighbour's receive lists must be contiguous ranges of the library's numbering
        h->peers.resize(nNeighbors);
        h->recvCellStart.assign(nNeighbors, 0); h->recvCellCount.assign(nNeighbors, 0);
        h->recvEdgeStart.assign(nNeighbors, 0); h->recvEdgeCount.assign(nNeighbors, 0);
        h->sendCellCount.resize(nNeighbors); h->sendEdgeCount.resize(nNeighbors);
        for (int i = 0; i < nNeighbors; ++i) {
            h->sendCellCount[i] = (int32_t)(sendCellOff[i + 1] - sendCellOff[i]);
            h->sendEdgeCount[i] = (int32_t)(sendEdgeOff[i + 1] - sendEdgeOff[i]);
        }
        h->directOk = true;
        for (int i = 0; i < nNeighbors && h->directOk; ++i) {
            const int64_t nc = recvCellOff[i + 1] - recvCellOff[i], ne = recvEdgeOff[i + 1] - recvEdgeOff[i];
            h->recvCellCount[i] = (int32_t)nc; h->recvEdgeCount[i] = (int32_t)ne;
            if (nc) h->recvCellStart[i] = p.cellO2N[recvCells[recvCellOff[i]]];
            if (ne) h->recvEdgeStart[i] = p.edgeO2N[recvEdges[recvEdgeOff[i]]];
            for (int64_t j = 0; j < nc && h->directOk; ++j)
                if (p.cellO2N[recvCells[recvCellOff[i] + j]] != h->recvCellStart[i] + j) h->directOk = false;
            for (int64_t j = 0; j < ne && h->directOk; ++j)
                if (p.edgeO2N[recvEdges[recvEdgeOff[i] + j]] != h->recvEdgeStart[i] + j) h->directOk = false;
            if (!h->directOk)
                h->directWhy = "the receive lists of neighbour " + std::to_string(i) + " are not a contiguous range in the library's order "
                               "(give halo cells the class 2 + neighbour index and list them in moka_mesh_permutation order)";
        }
        if (h->directOk) {
            std::vector<uint32_t> rows;
            rows.reserve((size_t)(2 * sendCellOff[nNeighbors] + sendEdgeOff[nNeighbors]) * 3);
            for (int i = 0; i < nNeighbors; ++i) {
                for (int64_t j = sendCellOff[i]; j < sendCellOff[i + 1]; ++j) {
                    const uint32_t src = (uint32_t)p.cellO2N[sendCells[j]], dst = (uint32_t)(j - sendCellOff[i]);
                    rows.insert(rows.end(), {src, dst, (uint32_t)i | (0u << 16)});
                    rows.insert(rows.end(), {src, dst, (uint32_t)i | (1u << 16)});
                }
                for (int64_t j = sendEdgeOff[i]; j < sendEdgeOff[i + 1]; ++j)
                    rows.insert(rows.end(), {(uint32_t)p.edgeO2N[sendEdges[j]], (uint32_t)(j - sendEdgeOff[i]), (uint32_t)i | (2u << 16)});
            }
            h->nPushRows = (int64_t)rows.size() / 3;
            if (!rows.empty()) {
                void *d = nullptr;
                hipError_t e = hipMalloc(&d, rows.size() * sizeof(uint32_t));
                if (e != hipSuccess) { moka_halo_destroy(h); return fail(st->ctx, MOKA_ERR_ALLOC, "hipMalloc of the push row list failed"); }
                h->allocs.push_back(d);
                if ((rc = h2d(st->ctx, d, rows.data(), rows.size() * sizeof(uint32_t)))) { moka_halo_destroy(h); return rc; }
                h->pushRows = static_cast<uint32_t *>(d);
            }
        }
    } catch (const std::bad_alloc &) {
        moka_halo_destroy(h);
        return fail(st->ctx, MOKA_ERR_ALLOC, "out of host memory building the halo maps");
    }
    for (hipEvent_t &e : h->evB) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (hipEventCreateWithFlags(&h->evPush, hipEventDisableTiming) != hipSuccess) {
        moka_halo_destroy(h);
        return fail(st->ctx, MOKA_ERR_HIP, "hipEventCreate failed");
    }
    state_attach(st);
    h->counted = true;
    *out = h;
    return MOKA_OK;
}

void moka_halo_destroy(moka_halo *h)
{
    if (!h) return;
    if (h->counted) state_detach(h->st);
    (void)hipSetDevice(h->st->ctx->device);
    (void)hipStreamSynchronize(h->st->ctx->stream);
    (void)hipStreamSynchronize(h->st->ctx->comm);
    for (hipEvent_t e : h->stEv) (void)hipEventDestroy(e);
    for (PeerLink &pl : h->peers)
        if (pl.flagsDev && pl.ipc) (void)hipHostUnregister((void *)pl.flags);
    // my own block: registered by moka_halo_set_stream_flags of this object -- or of a neighbour in the same process, which
    // registers the very same memory; it is freed below, so the registration goes whoever made it (an error = none existed)
    if (h->flags) (void)hipHostUnregister((void *)h->flags);
    (void)hipGetLastError();
    for (PeerLink &pl : h->peers) {
        if (pl.connected && pl.ipc) {
            for (void *q : pl.mapped) if (q) (void)hipIpcCloseMemHandle(q);
            if (pl.flags) munmap((void *)pl.flags, pl.flagsBytes);
        }
    }
    if (h->flags) {
        if (!h->shmName.empty()) { munmap((void *)h->flags, h->flagsBytes); shm_unlink(h->shmName.c_str()); }
        else free((void *)h->flags);
    }
    if (h->evPush) (void)hipEventDestroy(h->evPush);
    for (hipEvent_t e : h->evB) if (e) (void)hipEventDestroy(e);
    for (void *q : h->allocs) (void)hipFree(q);
    delete h;
}

int moka_halo_buffer_elems(const moka_halo *h, int64_t *sendElems, int64_t *recvElems)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    if (sendElems) *sendElems = h->nSend;
    if (recvElems) *recvElems = h->nRecv;
    return MOKA_OK;
}

// what: 0 = the current time level, 1..4 = the output of RK4 stage `what` (valid between dist_begin and dist_end);
// 5 = the new time level of a distributed Forward-Euler step before its levels rotate
int moka_halo_pack(moka_halo *h, int what, void *sendbuf)
{
    if (!h || (!sendbuf && h->nSend)) return fail(h ? h->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    if (what < 0 || what > 5) return fail(h->st->ctx, MOKA_ERR_ARG, "what must be 0..5");
    moka_state *st = h->st;
    moka_ctx *c = st->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    const LevelBufs &o = rk4_stage_output(st, what);
    // the rows to send are produced by the boundary patches (or by whatever last ran on the compute stream)
    HIPCHK(c, hipEventRecord(c->evBoundary, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm, c->evBoundary, 0));
    if (st->f32)
        HIPCHK(c, launch_halo_map_f32(static_cast<float *>(sendbuf), (float *)o.h, (float *)o.ssh, (float *)o.u, h->sendMap,
                                      h->nSend, 0, c->comm));
    else
        HIPCHK(c, launch_halo_map(static_cast<double *>(sendbuf), o.h, o.ssh, o.u, h->sendMap, h->nSend, 0, c->comm));
    return MOKA_OK;
}

int moka_halo_unpack(moka_halo *h, int what, const void *recvbuf)
{
    if (!h || (!recvbuf && h->nRecv)) return fail(h ? h->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    if (what < 0 || what > 5) return fail(h->st->ctx, MOKA_ERR_ARG, "what must be 0..5");
    moka_state *st = h->st;
    moka_ctx *c = st->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    const LevelBufs &o = rk4_stage_output(st, what);
    // Inside a step (what >= 1) no launch of the stage writes a received row (patches never straddle into the halo
    // classes), so the exchange and the unpack overlap the interior launch completely.  For the current time level
    // (what == 0) anything may have run on the compute stream before: wait for it.
    if (what == 0) {
        HIPCHK(c, hipEventRecord(c->evInterior, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->comm, c->evInterior, 0));
    }
    if (st->f32)
        HIPCHK(c, launch_halo_map_f32(static_cast<float *>(const_cast<void *>(recvbuf)), (float *)o.h, (float *)o.ssh,
                                      (float *)o.u, h->recvMap, h->nRecv, 1, c->comm));
    else
        HIPCHK(c, launch_halo_map(static_cast<double *>(const_cast<void *>(recvbuf)), o.h, o.ssh, o.u, h->recvMap, h->nRecv, 1,
                                  c->comm));
    HIPCHK(c, hipEventRecord(c->evHalo, c->comm));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->evHalo, 0));     // whatever comes next on the compute stream sees the halo
    return MOKA_OK;
}

// The same exchange for arbitrary fields with the state's shapes (u-like (K, nEdges), h-like (K, nCells), ssh-like (nCells), in
// the library's numbering and the state's storage type): the adjoint state of a partitioned run (moka_adjoint_rk4_stage_fields).
// Whatever ran on the compute stream before is waited for; whatever comes next on it sees the received rows.
int moka_halo_pack_fields(moka_halo *h, const void *uField, const void *hField, const void *sField, void *sendbuf)
{
    if (!h || !uField || !hField || !sField || (!sendbuf && h->nSend)) return fail(h ? h->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    moka_state *st = h->st;
    moka_ctx *c = st->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->evBoundary, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm, c->evBoundary, 0));
    if (st->f32)
        HIPCHK(c, launch_halo_map_f32(static_cast<float *>(sendbuf), (float *)const_cast<void *>(hField), (float *)const_cast<void *>(sField),
                                      (float *)const_cast<void *>(uField), h->sendMap, h->nSend, 0, c->comm));
    else
        HIPCHK(c, launch_halo_map(static_cast<double *>(sendbuf), (double *)const_cast<void *>(hField), (double *)const_cast<void *>(sField),
                                  (double *)const_cast<void *>(uField), h->sendMap, h->nSend, 0, c->comm));
    return MOKA_OK;
}

int moka_halo_unpack_fields(moka_halo *h, void *uField, void *hField, void *sField, const void *recvbuf)
{
    if (!h || !uField || !hField || !sField || (!recvbuf && h->nRecv)) return fail(h ? h->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    moka_state *st = h->st;
    moka_ctx *c = st->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->evInterior, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm, c->evInterior, 0));
    if (st->f32)
        HIPCHK(c, launch_halo_map_f32(static_cast<float *>(const_cast<void *>(recvbuf)), (float *)hField, (float *)sField, (float *)uField,
                                      h->recvMap, h->nRecv, 1, c->comm));
    else
        HIPCHK(c, launch_halo_map(static_cast<double *>(const_cast<void *>(recvbuf)), (double *)hField, (double *)sField, (double *)uField,
                                  h->recvMap, h->nRecv, 1, c->comm));
    HIPCHK(c, hipEventRecord(c->evHalo, c->comm));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->evHalo, 0));
    return MOKA_OK;
}

// ---------------------------------------------------------------------------------------------
// direct transport
// ---------------------------------------------------------------------------------------------
int moka_halo_set_overlap(moka_halo *h, int mode)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    if (mode < -1 || mode > 1) return hfail(h, MOKA_ERR_ARG, "mode must be -1 (automatic), 0 or 1");
    h->overlapB = mode < 0 ? (h->pBoundary > 0 && h->pOwned - h->pBoundary < 12000) : mode == 1;
    return MOKA_OK;
}

int moka_halo_set_acquire(moka_halo *h, int on)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    h->acquireFence = on != 0;
    return MOKA_OK;
}

int moka_halo_direct_available(const moka_halo *h)
{
    return h && h->directOk ? 1 : 0;
}

int moka_halo_export(moka_halo *h, int32_t nbr, int32_t shared, moka_halo_peer_info *out)
{
    if (!h || !out) return fail(h ? h->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    if (nbr < 0 || nbr >= h->nNbr) return hfail(h, MOKA_ERR_ARG, "neighbour index out of range");
    if (!h->directOk) return hfail(h, MOKA_ERR_UNSUPPORTED, "direct halo transport unavailable: " + h->directWhy);
    moka_state *st = h->st;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (int rc = ensure_flags(h, shared != 0)) return rc;
    if (shared && h->shmName.empty()) return hfail(h, MOKA_ERR_ARG, "the flag block of this halo was created process-local");
    std::memset(out, 0, sizeof *out);
    for (int t = 0; t < moka_state::NPHYS; ++t) {
        void *ptrs[3] = {st->phys[t].u, st->phys[t].h, st->phys[t].ssh};
        for (int f = 0; f < 3; ++f) {
            out->ptr[3 * t + f] = (uint64_t)(uintptr_t)ptrs[f];
            if (shared) {
                hipIpcMemHandle_t hd;
                HIPCHK(st->ctx, hipIpcGetMemHandle(&hd, ptrs[f]));
                static_assert(sizeof hd <= sizeof out->ipc[0], "IPC handle larger than its slot");
                std::memcpy(out->ipc[3 * t + f], &hd, sizeof hd);
            }
        }
    }
    out->flagPtr = (uint64_t)(uintptr_t)h->flags;
    std::strncpy(out->shmName, h->shmName.c_str(), sizeof out->shmName - 1);
    out->dstCell = h->recvCellStart[nbr]; out->dstEdge = h->recvEdgeStart[nbr];
    out->nCells = h->recvCellCount[nbr]; out->nEdges = h->recvEdgeCount[nbr];
    out->slot = nbr;
    out->nNeighbors = h->nNbr;
    out->pid = (int32_t)getpid();
    out->device = st->ctx->device;
    out->stateBytes = st->f32 ? 4 : 8;
    out->nVertLevels = st->mesh->plan.K;
    return MOKA_OK;
}

int moka_halo_connect(moka_halo *h, int32_t nbr, const moka_halo_peer_info *peer, int32_t shared)
{
    if (!h || !peer) return fail(h ? h->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    if (nbr < 0 || nbr >= h->nNbr) return hfail(h, MOKA_ERR_ARG, "neighbour index out of range");
    if (!h->directOk) return hfail(h, MOKA_ERR_UNSUPPORTED, "direct halo transport unavailable: " + h->directWhy);
    moka_state *st = h->st;
    const Plan &p = st->mesh->plan;
    if (peer->stateBytes != (st->f32 ? 4 : 8) || peer->nVertLevels != p.K)
        return hfail(h, MOKA_ERR_ARG, "peer state has another storage type or nVertLevels");
    // what I send to this neighbour must be exactly what it expects to receive from me
    if (peer->nCells != h->sendCellCount[nbr] || peer->nEdges != h->sendEdgeCount[nbr])
        return hfail(h, MOKA_ERR_ARG, "neighbour " + std::to_string(nbr) + " expects " + std::to_string(peer->nCells) + " cells / " +
                                          std::to_string(peer->nEdges) + " edges, this rank lists " + std::to_string(h->sendCellCount[nbr]) +
                                          " / " + std::to_string(h->sendEdgeCount[nbr]));
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    PeerLink pl;
    pl.ipc = shared != 0;
    pl.dstCell = peer->dstCell; pl.dstEdge = peer->dstEdge; pl.slot = peer->slot;
    if (shared) {
        if (peer->pid == (int32_t)getpid()) return hfail(h, MOKA_ERR_ARG, "shared = 1 is for peers in ANOTHER process (use shared = 0 inside one process)");
        for (int i = 0; i < 3 * moka_state::NPHYS; ++i) {
            hipIpcMemHandle_t hd;
            std::memcpy(&hd, peer->ipc[i], sizeof hd);
            void *q = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&q, hd, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                for (void *m : pl.mapped) if (m) (void)hipIpcCloseMemHandle(m);
                return hfail(h, MOKA_ERR_COMM, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
            }
            pl.mapped[i] = q;
        }
        auto unmap_all = [&]() { for (void *m : pl.mapped) if (m) (void)hipIpcCloseMemHandle(m); };
        const int fd = shm_open(peer->shmName, O_RDWR, 0600);
        if (fd < 0) {
            unmap_all();
            return hfail(h, MOKA_ERR_COMM, std::string("shm_open(") + peer->shmName + ") of the peer's flag block failed");
        }
        pl.flagsBytes = sizeof(uint64_t) * (size_t)std::max(peer->nNeighbors, 1);
        void *q = mmap(nullptr, pl.flagsBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (q == MAP_FAILED) {
            unmap_all();
            return hfail(h, MOKA_ERR_COMM, "mmap of the peer's flag block failed");
        }
        pl.flags = static_cast<volatile uint64_t *>(q);
    } else {
        for (int i = 0; i < 3 * moka_state::NPHYS; ++i) pl.mapped[i] = (void *)(uintptr_t)peer->ptr[i];
        pl.flags = (volatile uint64_t *)(uintptr_t)peer->flagPtr;
        pl.flagsBytes = sizeof(uint64_t) * (size_t)std::max(peer->nNeighbors, 1);
        if (peer->device != st->ctx->device) {        // one process driving several devices: map the peer's memory
            const hipError_t e = hipDeviceEnablePeerAccess(peer->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                return hfail(h, MOKA_ERR_COMM, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            (void)hipGetLastError();
        }
    }
    if (!pl.flags) return hfail(h, MOKA_ERR_ARG, "peer info carries no flag block");
    pl.connected = true;
    PeerLink &old = h->peers[nbr];
    if (old.connected && old.ipc) {                   // connected before: release the earlier mappings
        for (void *m : old.mapped) if (m) (void)hipIpcCloseMemHandle(m);
        if (old.flagsDev) { (void)hipHostUnregister((void *)old.flags); (void)hipGetLastError(); }
        if (old.flags) munmap((void *)old.flags, old.flagsBytes);
    }
    if (old.connected && h->streamFlags) h->streamFlags = false;   // the new link's flag block is not registered: host handshake until moka_halo_set_stream_flags is called again
    h->peers[nbr] = pl;
    h->tabDirty = true;
    st->feLeanInteriorOnly = true;   // peers may store into this state's level sets from now on: see the header (lean steps)
    return MOKA_OK;
}

// queue the push of `what` on the comm stream behind whatever the compute stream holds now (the boundary patches)
int moka_halo_push_begin(moka_halo *h, int what)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    if (what < 0 || what > 5) return hfail(h, MOKA_ERR_ARG, "what must be 0..5");
    if (!all_connected(h)) return hfail(h, MOKA_ERR_ARG, "direct halo transport: not every neighbour is connected");
    moka_state *st = h->st;
    moka_ctx *c = st->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_peer_tab(h)) return rc;
    const LevelBufs &o = rk4_stage_output(st, what);
    const int t = phys_index(st, o);
    if (t < 0) return hfail(h, MOKA_ERR_ARG, "internal: unknown buffer set");
    HIPCHK(c, hipEventRecord(c->evBoundary, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm, c->evBoundary, 0));
    if (h->nPushRows > 0) {
        const Plan &p = st->mesh->plan;
        const uint32_t sb = st->f32 ? 4u : 8u, rowB = (uint32_t)p.K * sb;
        int64_t blocks = (h->nPushRows + 7) / 8;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_halo_push, dim3((unsigned)blocks), dim3(256), 0, c->comm, h->pushRows, h->nPushRows,
                           h->peerTab + (size_t)t * h->nNbr, (const unsigned char *)o.u, (const unsigned char *)o.h,
                           (const unsigned char *)o.ssh, rowB, sb);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(h->evPush, c->comm));
    ++h->seq;
    return MOKA_OK;
}

// wait (host) for this rank's push kernel, then tell every neighbour that exchange `seq` has arrived
int moka_halo_push_signal(moka_halo *h)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    moka_ctx *c = h->st->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (h->streamFlags) {     // the flag words are written by the comm stream itself, behind the push kernel: nothing to wait for here
        for (const PeerLink &pl : h->peers)
            HIPCHK(c, hipStreamWriteValue64(c->comm, pl.flagsDev + pl.slot, h->seq, 0));
        if (h->statsOn) ++h->stExchanges;
        return MOKA_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(c, hipEventSynchronize(h->evPush));
    const auto t1 = std::chrono::steady_clock::now();
    for (const PeerLink &pl : h->peers)
        __atomic_store_n(const_cast<uint64_t *>(pl.flags) + pl.slot, h->seq, __ATOMIC_RELEASE);
    if (h->statsOn) {
        h->stSignalWaitMs += std::chrono::duration<double, std::milli>(t1 - t0).count();
        h->stFlagStoreMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        ++h->stExchanges;
    }
    return MOKA_OK;
}

// wait (host) until every neighbour has signalled exchange `seq`; MOKA_ERR_COMM after timeout_s seconds
int moka_halo_push_wait(moka_halo *h, double timeout_s)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    if (h->streamFlags) {     // the stream that takes the next boundary launch waits for the neighbours' flag words itself
        moka_ctx *c = h->st->ctx;
        HIPCHK(c, hipSetDevice(c->device));
        // on BOTH streams: which of them takes the next launch that reads received rows depends on what follows (an RK4 stage in
        // either launch order, a Forward-Euler step, a pack, a download) and is not known here.  In the overlapped launch order
        // this holds the next interior launch back until the exchange has arrived -- correctness before the last microseconds
        // of an experimental transport.
        for (hipStream_t q : {c->stream, c->comm})
            for (int i = 0; i < h->nNbr; ++i)
                HIPCHK(c, hipStreamWaitValue64(q, h->flagsDev + i, h->seq, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
        return MOKA_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < h->nNbr; ++i) {
        unsigned spins = 0;
        while (__atomic_load_n(const_cast<uint64_t *>(h->flags) + i, __ATOMIC_ACQUIRE) < h->seq) {
            if ((++spins & 1023u) == 0) {
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (el > timeout_s)
                    return hfail(h, MOKA_ERR_COMM, "halo exchange " + std::to_string(h->seq) + ": neighbour " + std::to_string(i) +
                                                       " has not signalled after " + std::to_string(timeout_s) + " s");
                if (el > 0.002) std::this_thread::yield();
            }
        }
    }
    if (h->statsOn) h->stWaitMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MOKA_OK;
}

// Experiment: the exchange handshake as stream memory operations.  on = 1: the flag blocks (this rank's and
// every connected neighbour's) are registered with the device; from then on moka_halo_push_signal enqueues
// hipStreamWriteValue64(comm stream, neighbour's slot, seq) behind the push kernel and moka_halo_push_wait enqueues
// hipStreamWaitValue64(>= seq) for every neighbour on the stream that takes the next boundary launch: a distributed step is
// enqueue-only, no host thread waits or polls.  A stream wait has no timeout: a neighbour that dies leaves the queue blocked
// until the process ends -- which is why this is a candidate the transport selection has to qualify ("ipc-smo"), not the default.
// MOKA_ERR_UNSUPPORTED when the device cannot wait on memory or the blocks cannot be registered.
int moka_halo_set_stream_flags(moka_halo *h, int on)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    moka_ctx *c = h->st->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!on) { h->streamFlags = false; return MOKA_OK; }
    if (!all_connected(h)) return hfail(h, MOKA_ERR_ARG, "stream flags: connect every neighbour first");
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device) != hipSuccess || !can) {
        (void)hipGetLastError();
        return hfail(h, MOKA_ERR_UNSUPPORTED, "this device cannot wait on memory from a stream (hipDeviceAttributeCanUseStreamWaitValue)");
    }
    auto reg = [&](volatile uint64_t *host, size_t bytes, uint64_t **dev) -> int {
        if (*dev) return MOKA_OK;
        void *q = const_cast<uint64_t *>(host), *d = nullptr;
        hipError_t e = hipHostRegister(q, bytes, hipHostRegisterMapped);
        if (e != hipSuccess && e != hipErrorHostMemoryAlreadyRegistered) {
            (void)hipGetLastError();
            return hfail(h, MOKA_ERR_UNSUPPORTED, std::string("hipHostRegister of a flag block: ") + hipGetErrorString(e));
        }
        (void)hipGetLastError();
        if ((e = hipHostGetDevicePointer(&d, q, 0)) != hipSuccess) {
            (void)hipGetLastError();
            return hfail(h, MOKA_ERR_UNSUPPORTED, std::string("hipHostGetDevicePointer of a flag block: ") + hipGetErrorString(e));
        }
        *dev = static_cast<uint64_t *>(d);
        return MOKA_OK;
    };
    if (h->nNbr > 0)
        if (int rc = reg(h->flags, h->flagsBytes, &h->flagsDev)) return rc;
    for (PeerLink &pl : h->peers)
        if (int rc = reg(pl.flags, pl.flagsBytes, &pl.flagsDev)) return rc;
    h->streamFlags = true;
    return MOKA_OK;
}

// Measurement: where does a distributed step spend its time on this rank?  enable(1) forgets earlier samples and starts
// recording: host time inside moka_halo_push_signal (waiting for the own push kernel, then storing the flags) and inside
// moka_halo_push_wait (polling the neighbours' flags), host time of whole moka_rk4_dist_step calls, and HIP events around every
// boundary / interior launch (at most 2048 launches).  read synchronises both streams.
int moka_halo_stats_enable(moka_halo *h, int on)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    moka_ctx *c = h->st->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->comm));
    h->statsOn = on != 0;
    if (on) {
        h->stSignalWaitMs = h->stFlagStoreMs = h->stWaitMs = h->stStepHostMs = 0.0;
        h->stExchanges = h->stSteps = 0;
        h->stEvUsed = 0;
        h->stEvPart.clear();
    }
    return MOKA_OK;
}

int moka_halo_stats_read(moka_halo *h, moka_halo_stats *out)
{
    if (!h || !out) return fail(h ? h->st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    moka_ctx *c = h->st->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->comm));
    *out = moka_halo_stats{};
    out->steps = h->stSteps; out->exchanges = h->stExchanges;
    out->host_signal_wait_ms = h->stSignalWaitMs; out->host_flag_store_ms = h->stFlagStoreMs;
    out->host_wait_ms = h->stWaitMs; out->host_step_ms = h->stStepHostMs;
    for (size_t i = 0; i < h->stEvPart.size() && 2 * i + 1 < h->stEvUsed; ++i) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, h->stEv[2 * i], h->stEv[2 * i + 1]));
        if (h->stEvPart[i] == 0) { out->boundary_launch_ms += ms; ++out->boundary_launches; }
        else { out->interior_launch_ms += ms; ++out->interior_launches; }
    }
    return MOKA_OK;
}

// ---------------------------------------------------------------------------------------------
// distributed RK4 step
// ---------------------------------------------------------------------------------------------
int moka_rk4_dist_begin(moka_halo *h, double dt)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    moka_state *st = h->st;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    // like moka_step_rk4: lazily pending diagnostics / stage-4 tendencies of the previous step are superseded, not computed
    h->dt = dt;
    if (int rc = rk4_begin(st, &h->ssh0)) return rc;
    h->overlapNow = h->overlapB && !st->nonlinear;   // fixed for the step: the two forms order their launches by different events
    if (h->overlapNow) HIPCHK(st->ctx, hipEventRecord(st->ctx->evInterior, st->ctx->stream));
    return MOKA_OK;
}

// part 0: patches [0, first) = the boundary patches (their rows are what other ranks need); part 1: [first, owned).
// Halo patches [owned, nPatches) are never computed: their rows arrive through the exchange.
// part 2: the whole local mesh in one launch, halo entities included (redundantly: the exchange behind the stage overwrites them).
// States with the optional nonlinear terms (two-ring halo: every owned stencil is local) split a stage into its two kernels:
//   part 3: the preparation pass (potential vorticity, kinetic energy, thickness flux of the stage's provisional state) over the
//           boundary and the halo patches -- it reads halo rows, so the exchange of the previous stage must have arrived;
//   part 4: the same over the interior patches, which read owned rows only: it can be queued behind the PREVIOUS stage's
//           interior launch and then runs while that stage's exchange is still under way;
//   parts 0 / 1: the stage kernel over the boundary / interior patches (after parts 3 and 4 of the same stage).
// moka_rk4_dist_step orders them; part 2 stays the plain form (and the only one for the kernel variants without patch kernels).
int moka_rk4_dist_stage(moka_halo *h, int stage, int part)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    if (stage < 1 || stage > 4 || part < 0 || part > 4) return hfail(h, MOKA_ERR_ARG, "stage must be 1..4, part 0..4");
    moka_state *st = h->st;
    HIPCHK(st->ctx, hipSetDevice(st->ctx->device));
    if (part == 2) {
        const StageArgs g = rk4_stage_args(st, stage, h->dt, h->ssh0);
        HIPCHK(st->ctx, run_stage(st, g));
        return MOKA_OK;
    }
    if (part >= 3 && !st->nonlinear) return hfail(h, MOKA_ERR_ARG, "parts 3 and 4 are the preparation passes of the nonlinear terms");
    if (st->nonlinear) {
        if (h->overlapNow) return hfail(h, MOKA_ERR_ARG, "nonlinear terms: moka_rk4_dist_begin first");
        const StageArgs g = rk4_stage_args(st, stage, h->dt, h->ssh0);
        const int nP = st->mesh->plan.nPatches;
        hipError_t e = hipSuccess;
        st->nlPhase = part >= 3 ? 1 : 2;
        if (part == 0) {
            if (int rc = launch_acquire(h, st->ctx->stream)) { st->nlPhase = 0; return rc; }
            e = run_stage(st, g, 0, h->pFirst);
        } else if (part == 1) {
            e = run_stage(st, g, h->pFirst, h->pOwned - h->pFirst);
        } else if (part == 4) {
            e = run_stage(st, g, h->pFirst, h->pOwned - h->pFirst);
        } else {
            if (int rc = launch_acquire(h, st->ctx->stream)) { st->nlPhase = 0; return rc; }
            e = run_stage(st, g, 0, h->pFirst);
            if (e == hipSuccess) e = run_stage(st, g, h->pOwned, nP - h->pOwned);
        }
        st->nlPhase = 0;
        if (e == hipErrorNotSupported)
            return hfail(h, MOKA_ERR_UNSUPPORTED, "nonlinear terms on patch ranges need the patch kernels (kernel variant 0 or 4, even K <= 64): use part 2");
        HIPCHK(st->ctx, e);
        return MOKA_OK;
    }
    // Boundary group first, interior right behind it on the same (compute) stream: in-order, no cross-queue wait in the
    // compute chain.  Launched concurrently the two kernels share the CUs and the ~130 boundary workgroups finish no
    // earlier than the thousands of interior ones (measured 290 us instead of 35 us), which would push the exchange
    // behind the interior compute it is meant to hide under.  The comm stream waits for the boundary group only (its
    // event is recorded before the interior launch is queued).
    return dist_stage_part(h, stage, part);
}

// can this state's stages run part by part (linear terms: always; nonlinear terms: with the patch kernels)
int moka_rk4_dist_parts_available(const moka_halo *h)
{
    if (!h) return 0;
    const moka_state *st = h->st;
    if (!st->nonlinear) return 1;
    const int form = st->ctx->variant == 4 ? 1 : st->ctx->variant == 3 ? 3 : 0;
    return nl_patch_forms(st->mesh->dev, st->mesh->lpc, form) ? 1 : 0;
}

// direct transport, the device-queue half of a stage: boundary patches, push (comm stream), interior patches
int moka_rk4_dist_stage_launch(moka_halo *h, int stage)
{
    int rc;
    if ((rc = moka_rk4_dist_stage(h, stage, 0))) return rc;
    if ((rc = moka_halo_push_begin(h, stage))) return rc;
    return moka_rk4_dist_stage(h, stage, 1);
}

int moka_rk4_dist_end(moka_halo *h)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    if (h->overlapNow) HIPCHK(h->st->ctx, hipStreamWaitEvent(h->st->ctx->stream, h->evB[0], 0));    // stage 4's boundary launch
    rk4_end(h->st);
    return MOKA_OK;
}

// One whole distributed RK4 step in one call.  With every neighbour connected (moka_halo_connect) the exchange is the
// direct one; otherwise `transport(user, stage, sendbuf, recvbuf)` has to move the packed send buffer of the stage to
// the neighbours and fill the receive buffer (stream-ordered on the comm stream, or synchronously), and sendbuf / recvbuf
// are the device buffers it works on.
int moka_rk4_dist_step(moka_halo *h, double dt, moka_transport_fn transport, void *user, void *sendbuf, void *recvbuf,
                       double timeout_s)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    int rc;
    // the caller chooses: transport == NULL is the direct exchange (every neighbour must be connected), a callback is ALWAYS
    // used when given -- a connected halo does not override it (round 2 did: the buffered candidates of the transport
    // selection were then never exercised)
    const bool direct = h->nNbr > 0 && !transport;
    if (direct && !all_connected(h)) return hfail(h, MOKA_ERR_ARG, "no transport callback and not every neighbour is connected");
    struct StepClock {                  // measurement: host time of the whole call
        moka_halo *h; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~StepClock() { if (h->statsOn) { h->stStepHostMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ++h->stSteps; } }
    } stepClock{h};
    const bool nl = h->st->nonlinear;
    if (nl && !moka_rk4_dist_parts_available(h))     // before the step is opened (rk4_begin supersedes what is lazily pending)
        return hfail(h, MOKA_ERR_UNSUPPORTED, "nonlinear terms: this kernel variant has whole-mesh stages only (moka_rk4_dist_stage part 2)");
    if ((rc = moka_rk4_dist_begin(h, dt))) return rc;
    // nonlinear terms: the preparation pass of the interior patches of stage s + 1 is queued right behind the interior launch of
    // stage s (it reads owned rows only) and so overlaps exchange s as well; that of the boundary and halo patches follows the wait
    if (nl && (rc = moka_rk4_dist_stage(h, 1, 4))) return rc;
    for (int s = 1; s <= 4; ++s) {
        if (nl && (rc = moka_rk4_dist_stage(h, s, 3))) return rc;
        if ((rc = moka_rk4_dist_stage(h, s, 0))) return rc;
        if (direct) {
            if ((rc = moka_halo_push_begin(h, s))) return rc;
            if ((rc = moka_rk4_dist_stage(h, s, 1))) return rc;
            if (nl && s < 4 && (rc = moka_rk4_dist_stage(h, s + 1, 4))) return rc;
            if ((rc = moka_halo_push_signal(h))) return rc;
            if ((rc = moka_halo_push_wait(h, timeout_s))) return rc;
        } else if (h->nNbr > 0) {
            if ((rc = moka_halo_pack(h, s, sendbuf))) return rc;
            if ((rc = moka_rk4_dist_stage(h, s, 1))) return rc;
            if (nl && s < 4 && (rc = moka_rk4_dist_stage(h, s + 1, 4))) return rc;
            if (int trc = transport(user, s, sendbuf, recvbuf))
                return hfail(h, MOKA_ERR_COMM, "the halo transport callback failed at stage " + std::to_string(s) + " (code " + std::to_string(trc) + ")");
            if ((rc = moka_halo_unpack(h, s, recvbuf))) return rc;
        } else {
            if ((rc = moka_rk4_dist_stage(h, s, 1))) return rc;
            if (nl && s < 4 && (rc = moka_rk4_dist_stage(h, s + 1, 4))) return rc;
        }
    }
    return moka_rk4_dist_end(h);
}

// ---------------------------------------------------------------------------------------------
// distributed Forward-Euler step: the reference's live integrator (time_integration.jl:150-193) on a partitioned mesh.
// Everything a computed entity reads is local: an edge with an owned cell is computed here (its layerThicknessEdge
// too, from the exchanged layerThickness of the halo cell -- so the stale-thickness flux of reference_compat needs no
// exchange of its own), an edge without one arrives by exchange together with the new level's layerThickness / ssh.
//   launch: relativeVorticity -> boundary patches -> (push or pack of the NEW level, `what` = 5) -> interior patches
//   then the exchange completes (push_signal / push_wait, or transport + unpack), then moka_fe_dist_end swaps the levels.
// ---------------------------------------------------------------------------------------------
// Do the patch launches of a distributed Forward-Euler step carry the vertex pass themselves (StageArgs.vort)?  A pure function
// of the mesh, the storage type and the kernel choice, so every part of a step -- in whatever order they are called -- agrees.
static bool fe_vort_fused(const moka_halo *h)
{
    const moka_state *st = h->st;
    const moka_mesh *mm = st->mesh;
    MeshDev dev = mm->dev;
    dev.maxOwnE = std::max(mm->plan.maxOwnELaunch, 1); dev.maxOwnC = std::max(mm->plan.maxOwnCLaunch, 1);   // bounds both ranges
    if (st->f32) return stage_curl_fits(dev, true);
    return (st->ctx->variant == 0 || st->ctx->variant == 11) && mm->lpc == 64 && mm->colOk && stage_curl_fits(dev, false);
}

static int fe_dist_args(moka_state *st, double dt, int flags, StageArgs *g, FeArgs *a)
{
    *a = fe_args(st, FE_FLUX | FE_DIV | FE_CURL | FE_HEDGE | FE_TENDU | FE_TENDH | FE_UPDATE, flags, dt);
    *g = fe_stage_args(st, *a, flags);      // the new level goes to the spare set: the previous level stays readable (mode 6)
    return MOKA_OK;
}

int moka_fe_dist_launch(moka_halo *h, double dt, int flags, int part)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    if (part < 0 || part > 2) return hfail(h, MOKA_ERR_ARG, "part must be 0 (boundary), 1 (interior) or 2 (vertices)");
    moka_state *st = h->st;
    moka_ctx *c = st->ctx;
    if (st->nonlinear) return hfail(h, MOKA_ERR_UNSUPPORTED, "distributed Forward Euler: the reference's linear terms only");
    if (flags & MOKA_FE_LEVEL1_ONLY && st->mesh->plan.K > 1)
        return hfail(h, MOKA_ERR_UNSUPPORTED, "distributed Forward Euler steps all levels (MOKA_FE_LEVEL1_ONLY is for nVertLevels = 1)");
    HIPCHK(c, hipSetDevice(c->device));
    // whichever part of a step is launched first deals with lazily pending arrays (idempotent: the other parts find nothing)
    if (int rc = fe_begin(st, flags)) return rc;
    StageArgs g;
    FeArgs a;
    fe_dist_args(st, dt, flags, &g, &a);
    h->feFlags = flags;
    const moka_mesh *mm = st->mesh;
    if (part == 2) {
        // relativeVorticity of the OLD state (DiagnosticVars.jl:108-117 runs before the update): every local vertex -- unless
        // the patch launches carry the vertices of their patches themselves (StageArgs.vort).  Then this part is empty: the
        // vertices of the boundary patches are computed by the boundary launch, i.e. ahead of this rank's push, and no vertex of
        // an interior patch has an edge whose row is received (its three cells are all interior: a cell next to a halo cell is
        // a boundary cell, and a vertex with a boundary cell belongs to a boundary patch).
        if (fe_vort_fused(h)) return MOKA_OK;
        if (int rc = launch_acquire(h, c->stream)) return rc;          // it reads received rows (old normalVelocity of halo edges)
        if (st->f32) {
            HIPCHK(c, launch_curl_f32(mm->dev, reinterpret_cast<const float *>(a.u), reinterpret_cast<float *>(a.vort),
                                      flags & MOKA_FE_ACCUM_VORT, c->stream));
            return MOKA_OK;
        }
        const hipError_t ec = launch_curl2(mm->dev, a.u, a.vort, flags & MOKA_FE_ACCUM_VORT, c->stream);
        if (ec == hipErrorNotSupported) {
            a.ops = FE_CURL;
            HIPCHK(c, launch_fe(mm->dev, a, mm->lpc, c->stream));
        } else {
            HIPCHK(c, ec);
        }
        return MOKA_OK;
    }
    const int p0 = part == 0 ? 0 : h->pFirst, cnt = part == 0 ? h->pFirst : h->pOwned - h->pFirst;
    if (cnt <= 0) return MOKA_OK;
    if (part == 0) if (int rc = launch_acquire(h, c->stream)) return rc;
    MeshDev dev = mm->dev;
    dev.patchBegin = p0; dev.nPatches = cnt; dev.tailPatch = -1;
    {
        const moka::Plan &p = mm->plan;
        int mE = 1, mC = 1;
        for (int q = p0; q < p0 + cnt; ++q) {
            mE = std::max(mE, p.patchEdgeStart[q + 1] - p.patchEdgeStart[q]);
            mC = std::max(mC, p.patchCellStart[q + 1] - p.patchCellStart[q]);
        }
        dev.maxOwnE = mE; dev.maxOwnC = mC;
    }
    hipError_t e = hipErrorNotSupported;
    h->fePrev = g.hPrev != nullptr;
    const bool vortFused = fe_vort_fused(h);
    if (!vortFused) g.vort = nullptr;       // part 2 runs the vertex pass over every local vertex
    if (part == 0 && st->feLeanInteriorOnly && fe_lean(st, flags)) {
        // lean step, direct halo: the boundary patches store their DiagnosticVars / TendencyVars now (see the header), in place
        g.tendU = a.tendU; g.tendH = a.tendH; g.F = a.F; g.div = a.div;
        g.hEdgeNew = st->hEdge[0];
    }
    if (st->f32) {                         // the Forward-Euler modes of the fp32-storage kernel: no generic form behind them
        HIPCHK(c, launch_stage_rec2c_f32(dev, g, c->stream));
        h->feStageKernel = true;
        return MOKA_OK;
    }
    if ((c->variant == 0 || c->variant == 11) && mm->lpc == 64 && mm->colOk) e = launch_stage_rec2c(dev, g, c->stream);
    h->feStageKernel = e == hipSuccess;
    if (e == hipErrorNotSupported) {
        if (vortFused) return hfail(h, MOKA_ERR_UNSUPPORTED, "internal: the stage kernel refused a launch whose vertex pass it was to carry");
        a.ops &= ~FE_CURL;                 // the generic one-launch kernel over the same patch range, vertices in part 2
        e = launch_fe(dev, a, mm->lpc, c->stream);
    }
    HIPCHK(c, e);
    return MOKA_OK;
}

int moka_fe_dist_end(moka_halo *h)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    moka_state *st = h->st;
    // (lean: decided by the same pure function the launches used; the stage kernel interpolates layerThicknessEdge of every edge
    //  with an owned cell from the level that becomes the previous one now)
    fe_end(st, h->feFlags, h->feStageKernel, h->feStageKernel && fe_lean(st, h->feFlags), h->fePrev);
    if (st->feLazy && st->feLeanInteriorOnly) {      // the boundary patches have stored theirs: pending = the interior patches
        st->feLazyBegin = h->pFirst;
        st->feLazyCount = h->pOwned - h->pFirst;
    }
    st->sshConsistent = true;
    return MOKA_OK;
}

int moka_fe_dist_step(moka_halo *h, double dt, int flags, moka_transport_fn transport, void *user, void *sendbuf, void *recvbuf,
                      double timeout_s)
{
    if (!h) return fail(nullptr, MOKA_ERR_ARG, "halo is NULL");
    int rc;
    const bool direct = h->nNbr > 0 && !transport;          // the caller chooses, as in moka_rk4_dist_step
    if (direct && !all_connected(h)) return hfail(h, MOKA_ERR_ARG, "no transport callback and not every neighbour is connected");
    // The vertex pass goes FIRST: it reads old-level normalVelocity rows of halo edges, i.e. rows of the buffer set the
    // neighbours' next step pushes into.  Ahead of the boundary launch it is also ahead of this rank's push, whose completion
    // is what lets a neighbour run on (see the header of this file).
    if ((rc = moka_fe_dist_launch(h, dt, flags, 2))) return rc;
    if ((rc = moka_fe_dist_launch(h, dt, flags, 0))) return rc;
    if (direct) {
        if ((rc = moka_halo_push_begin(h, 5))) return rc;
    } else if (h->nNbr > 0) {
        if ((rc = moka_halo_pack(h, 5, sendbuf))) return rc;
    }
    if ((rc = moka_fe_dist_launch(h, dt, flags, 1))) return rc;
    if (direct) {
        if ((rc = moka_halo_push_signal(h))) return rc;
        if ((rc = moka_halo_push_wait(h, timeout_s))) return rc;
    } else if (h->nNbr > 0) {
        if (int trc = transport(user, 5, sendbuf, recvbuf))
            return hfail(h, MOKA_ERR_COMM, "the halo transport callback failed (code " + std::to_string(trc) + ")");
        if ((rc = moka_halo_unpack(h, 5, recvbuf))) return rc;
    }
    return moka_fe_dist_end(h);
}

}  // extern "C"
