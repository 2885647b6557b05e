// plan.cpp -- host-side mesh plan: validation, 0-based conversion, cell ordering (RCB patches or
// RCM), edge/vertex renumbering by first-touching cell, per-entity records.  No GPU needed.
//
// Input is the reference's mesh exactly as Julia holds it (HorzMesh.jl:64-162, VertMesh.jl:3-26).
#include <algorithm>
#include <array>
#include <cmath>
#include <new>
#include <numeric>
#include <queue>

#include "moka_internal.hpp"

namespace moka {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const char *get_error() { return g_err.c_str(); }

namespace {

inline int64_t IX(int i, int64_t j, int ld) { return j * ld + i; }   // 0-based slot i of entity j

#define REQUIRE(cond, msg)                 \
    do {                                   \
        if (!(cond)) {                     \
            set_error(msg);                \
            return MOKA_ERR_ARG;           \
        }                                  \
    } while (0)

// ---- recursive coordinate bisection into patches of exactly P cells (last one may be short) ----
// stable partition of an ordering by cell class; returns the class boundaries
std::vector<int> partition_by_class(const moka_mesh_desc *d, std::vector<int32_t> &n2o)
{
    std::vector<int> bounds{0};
    if (!d->cellClass) { bounds.push_back((int)n2o.size()); return bounds; }
    int maxc = 0;
    for (int c = 0; c < d->nCells; ++c) maxc = std::max(maxc, (int)d->cellClass[c]);
    std::vector<int32_t> out;
    out.reserve(n2o.size());
    for (int k = 0; k <= maxc; ++k) {
        for (int32_t c : n2o) if (d->cellClass[c] == k) out.push_back(c);
        bounds.push_back((int)out.size());
    }
    n2o.swap(out);
    return bounds;
}

void rcb_order(const moka_mesh_desc *d, int P, std::vector<int32_t> &n2o)
{
    const int n = d->nCells;
    n2o.resize(n);
    std::iota(n2o.begin(), n2o.end(), 0);
    const double *X[3] = {d->xCell, d->yCell, d->zCell};
    struct Range { int lo, hi; };
    std::vector<Range> stack;
    {
        const std::vector<int> b = partition_by_class(d, n2o);   // RCB inside each class range
        for (size_t i = b.size() - 1; i >= 1; --i) if (b[i] > b[i - 1]) stack.push_back({b[i - 1], b[i]});
    }
    while (!stack.empty()) {
        Range r = stack.back();
        stack.pop_back();
        const int cnt = r.hi - r.lo;
        if (cnt <= P) {
            // inside a patch: order along the patch's longest axis (cheap locality)
            continue;
        }
        int axis = 0;
        double best = -1.0;
        for (int a = 0; a < 3; ++a) {
            if (!X[a]) continue;
            double mn = 1e300, mx = -1e300;
            for (int i = r.lo; i < r.hi; ++i) {
                double v = X[a][n2o[i]];
                mn = std::min(mn, v);
                mx = std::max(mx, v);
            }
            if (mx - mn > best) { best = mx - mn; axis = a; }
        }
        const int leaves = (cnt + P - 1) / P;
        const int left = (leaves / 2) * P;   // multiple of P => every leaf but the last is full
        const double *xa = X[axis];
        std::nth_element(n2o.begin() + r.lo, n2o.begin() + r.lo + left, n2o.begin() + r.hi,
                         [xa](int32_t a, int32_t b) { return xa[a] < xa[b] || (xa[a] == xa[b] && a < b); });
        // process left first => push right first
        stack.push_back({r.lo + left, r.hi});
        stack.push_back({r.lo, r.lo + left});
    }
}

// ---- reverse Cuthill-McKee on the cell graph (cells adjacent across an edge) ----
void rcm_order(const moka_mesh_desc *d, std::vector<int32_t> &n2o)
{
    const int n = d->nCells;
    std::vector<int32_t> deg(n, 0), start(n + 1, 0), adj;
    for (int64_t e = 0; e < d->nEdges; ++e) {
        int a = d->cellsOnEdge[2 * e] - 1, b = d->cellsOnEdge[2 * e + 1] - 1;
        if (a >= 0 && b >= 0 && a != b) { ++deg[a]; ++deg[b]; }
    }
    for (int i = 0; i < n; ++i) start[i + 1] = start[i] + deg[i];
    adj.resize(start[n]);
    std::vector<int32_t> fill(start.begin(), start.end() - 1);
    for (int64_t e = 0; e < d->nEdges; ++e) {
        int a = d->cellsOnEdge[2 * e] - 1, b = d->cellsOnEdge[2 * e + 1] - 1;
        if (a >= 0 && b >= 0 && a != b) { adj[fill[a]++] = b; adj[fill[b]++] = a; }
    }
    std::vector<char> seen(n, 0);
    std::vector<int32_t> order;
    order.reserve(n);
    auto bfs_far = [&](int s, std::vector<int32_t> &lvl) {   // returns last node of a BFS from s
        std::vector<int32_t> q{s};
        lvl.assign(n, -1);
        lvl[s] = 0;
        size_t h = 0;
        while (h < q.size()) {
            int u = q[h++];
            for (int k = start[u]; k < start[u + 1]; ++k)
                if (lvl[adj[k]] < 0 && !seen[adj[k]]) { lvl[adj[k]] = lvl[u] + 1; q.push_back(adj[k]); }
        }
        return q.back();
    };
    std::vector<int32_t> lvl;
    for (int s0 = 0; s0 < n; ++s0) {
        if (seen[s0]) continue;
        int s = bfs_far(bfs_far(s0, lvl), lvl);   // two sweeps: pseudo-peripheral start
        std::vector<int32_t> q{s};
        seen[s] = 1;
        size_t h = 0;
        std::vector<int32_t> nb;
        while (h < q.size()) {
            int u = q[h++];
            nb.clear();
            for (int k = start[u]; k < start[u + 1]; ++k)
                if (!seen[adj[k]]) { seen[adj[k]] = 1; nb.push_back(adj[k]); }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] < deg[b] || (deg[a] == deg[b] && a < b); });
            q.insert(q.end(), nb.begin(), nb.end());
        }
        order.insert(order.end(), q.begin(), q.end());
    }
    n2o.assign(order.rbegin(), order.rend());
}

}  // namespace

int build_plan(const moka_mesh_desc *d, Plan &p)
{
    REQUIRE(d != nullptr, "mesh descriptor is NULL");
    REQUIRE(d->nCells > 0 && d->nEdges > 0 && d->nVertices > 0, "mesh counts must be positive");
    REQUIRE(d->nVertLevels >= 1, "nVertLevels must be >= 1");
    REQUIRE(d->maxEdges >= 3 && d->maxEdges2 >= 1 && d->vertexDegree >= 3, "bad maxEdges/maxEdges2/vertexDegree");
    REQUIRE(d->nEdgesOnCell && d->edgesOnCell && d->edgeSignOnCell && d->areaCell, "PrimaryCells arrays missing");
    REQUIRE(d->cellsOnEdge && d->nEdgesOnEdge && d->edgesOnEdge && d->weightsOnEdge && d->dvEdge && d->dcEdge && d->fEdge,
            "Edges arrays missing");
    REQUIRE(d->edgesOnVertex && d->edgeSignOnVertex && d->areaTriangle, "DualCells arrays missing");
    REQUIRE(d->restingThicknessSum, "restingThicknessSum missing");
    const int nC = d->nCells, nE = d->nEdges, nV = d->nVertices, VD = d->vertexDegree;
    const int ldSV = d->edgeSignOnVertexLD > 0 ? d->edgeSignOnVertexLD : VD;
    REQUIRE(ldSV >= VD, "edgeSignOnVertexLD < vertexDegree");

    // ---- validation of connectivity ranges ----
    int maxEoC = 0, maxEoE = 0;
    for (int c = 0; c < nC; ++c) {
        int n = d->nEdgesOnCell[c];
        REQUIRE(n >= 1 && n <= d->maxEdges, "nEdgesOnCell out of range");
        maxEoC = std::max(maxEoC, n);
        for (int i = 0; i < n; ++i) {
            int e = d->edgesOnCell[IX(i, c, d->maxEdges)];
            REQUIRE(e >= 1 && e <= nE, "edgesOnCell out of range");
            int s = d->edgeSignOnCell[IX(i, c, d->maxEdges)];
            REQUIRE(s == 1 || s == -1, "edgeSignOnCell must be +-1 on active slots");
        }
        REQUIRE(d->areaCell[c] > 0.0, "areaCell must be positive");
    }
    for (int e = 0; e < nE; ++e) {
        int a = d->cellsOnEdge[2 * (int64_t)e], b = d->cellsOnEdge[2 * (int64_t)e + 1];
        REQUIRE(a >= 1 && a <= nC && b >= 1 && b <= nC, "cellsOnEdge out of range (non-periodic meshes are not supported, VertMesh.jl:50)");
        int n = d->nEdgesOnEdge[e];
        REQUIRE(n >= 0 && n <= d->maxEdges2, "nEdgesOnEdge out of range");
        maxEoE = std::max(maxEoE, n);
        for (int i = 0; i < n; ++i) {
            int x = d->edgesOnEdge[IX(i, e, d->maxEdges2)];
            REQUIRE(x >= 0 && x <= nE, "edgesOnEdge out of range");
        }
        REQUIRE(d->dcEdge[e] > 0.0, "dcEdge must be positive");
        if (d->maxLevelEdgeTop) REQUIRE(d->maxLevelEdgeTop[e] >= 0 && d->maxLevelEdgeTop[e] <= d->nVertLevels, "maxLevelEdgeTop out of range");
    }
    for (int v = 0; v < nV; ++v) {
        for (int j = 0; j < VD; ++j) {
            int e = d->edgesOnVertex[IX(j, v, VD)];
            REQUIRE(e >= 1 && e <= nE, "edgesOnVertex out of range");
            int s = d->edgeSignOnVertex[IX(j, v, ldSV)];
            REQUIRE(s == 1 || s == -1, "edgeSignOnVertex must be +-1");
        }
        REQUIRE(d->areaTriangle[v] > 0.0, "areaTriangle must be positive");
    }

    p.nC = nC; p.nE = nE; p.nV = nV; p.K = d->nVertLevels; p.VD = VD;
    p.ME = maxEoC <= 6 ? 6 : (maxEoC <= 8 ? 8 : maxEoC);
    p.ME2 = maxEoE <= 10 ? 10 : (maxEoE <= 14 ? 14 : maxEoE);
    if (p.ME == 8 && p.ME2 < 14) p.ME2 = 14;   // kernels are instantiated for (6,10), (6,14), (8,14)
    // default patch size (bench sweeps in profiles/r01_variants.txt): where the default stage kernel keeps the patch's own
    // u rows in LDS (k_stage_rec2c: even 34 <= K <= 64; k_stage_rec2c_f32: K % 4 == 0, 34 <= K <= 128) the largest patch
    // that still lets four (fp64: 16 cells, ~51 edges, 38 KB) or three (fp32: 24 cells) workgroups share a CU;
    // 32 cells otherwise (generic / plain column kernels).
    {
        const int K = d->nVertLevels;
        REQUIRE(d->stateBytes == 0 || d->stateBytes == 4 || d->stateBytes == 8, "stateBytes must be 0, 4 or 8");
        p.stateBytes = d->stateBytes == 4 ? 4 : 8;
        int def = 32;
        if (p.stateBytes == 8 && K >= 34 && K <= 64 && !(K & 1)) def = 16;
        if (p.stateBytes == 4 && K >= 34 && K <= 128 && !(K & 3)) def = 24;
        p.P = d->patch_cells > 0 ? d->patch_cells : def;
    }
    REQUIRE(p.P <= 4096, "patch_cells too large");

    // ---- cell ordering ----
    int ordering = d->ordering;
    const bool haveXYZ = d->xCell && d->yCell;
    if (ordering == MOKA_ORDER_DEFAULT) ordering = haveXYZ ? MOKA_ORDER_RCB : MOKA_ORDER_RCM;
    REQUIRE(ordering >= MOKA_ORDER_NONE && ordering <= MOKA_ORDER_RCB, "unknown ordering");
    REQUIRE(ordering != MOKA_ORDER_RCB || haveXYZ, "RCB ordering needs xCell/yCell");
    p.ordering = ordering;
    if (d->cellClass)
        for (int c = 0; c < nC; ++c) REQUIRE(d->cellClass[c] >= 0 && d->cellClass[c] < 64, "cellClass out of range");
    if (ordering == MOKA_ORDER_RCB) rcb_order(d, p.P, p.cellN2O);
    else if (ordering == MOKA_ORDER_RCM) { rcm_order(d, p.cellN2O); partition_by_class(d, p.cellN2O); }
    else { p.cellN2O.resize(nC); std::iota(p.cellN2O.begin(), p.cellN2O.end(), 0); partition_by_class(d, p.cellN2O); }
    p.cellO2N.assign(nC, -1);
    for (int i = 0; i < nC; ++i) p.cellO2N[p.cellN2O[i]] = i;

    // ---- patches: runs of P consecutive cells that never straddle a cell class (the last patch of a class may be short).
    // On a partitioned mesh the classes are: 0 owned cells another rank needs, 1 owned interior, 2 + i halo cells owned by
    // the rank's i-th neighbour -- so a launch over the patches of a class range touches rows of that range only, halo
    // patches are never computed, and what a neighbour sends lands in one contiguous range of cells and edges.
    std::vector<int32_t> patchOf(nC);
    p.classCellStart.assign(1, 0);
    p.classPatchStart.assign(1, 0);
    p.patchCellStart.assign(1, 0);
    {
        auto cls0 = [&](int cn) { return d->cellClass ? d->cellClass[p.cellN2O[cn]] : 0; };
        int begin = 0;
        while (begin < nC) {
            const int k = cls0(begin);
            int end = begin;
            while (end < nC && cls0(end) == k) ++end;
            REQUIRE((int)p.classCellStart.size() - 1 <= k, "internal: cell ordering is not class-major");
            while ((int)p.classCellStart.size() - 1 < k) {          // empty classes below k
                p.classCellStart.push_back(begin);
                p.classPatchStart.push_back((int)p.patchCellStart.size() - 1);
            }
            for (int c = begin; c < end; c += p.P) {
                const int q = (int)p.patchCellStart.size() - 1;
                for (int x = c; x < std::min(c + p.P, end); ++x) patchOf[x] = q;
                p.patchCellStart.push_back(std::min(c + p.P, end));
            }
            p.classCellStart.push_back(end);
            p.classPatchStart.push_back((int)p.patchCellStart.size() - 1);
            begin = end;
        }
    }
    p.nPatches = (int)p.patchCellStart.size() - 1;

    // ---- edges / vertices numbered by their owner cell (new numbering): counting sort by owner ----
    auto renumber = [&](int n, auto owner_of, std::vector<int32_t> &n2o, std::vector<int32_t> &o2n) {
        std::vector<int32_t> owner(n), cnt(nC + 1, 0);
        for (int i = 0; i < n; ++i) { owner[i] = owner_of(i); ++cnt[owner[i] + 1]; }
        for (int c = 0; c < nC; ++c) cnt[c + 1] += cnt[c];
        n2o.resize(n); o2n.resize(n);
        for (int i = 0; i < n; ++i) { int pos = cnt[owner[i]]++; n2o[pos] = i; o2n[i] = pos; }
        return owner;
    };
    // Edge ownership.  An edge whose two cells sit in one patch belongs to it.  An edge between two patches goes to the
    // cell of lower class when the classes differ (multi-GPU: the boundary launch must own every edge another rank needs,
    // and no computed edge may belong to a halo patch), otherwise to whichever patch owns fewer edges so far, followed by
    // a few balancing sweeps: the LDS carve of the record-staging kernels is sized by the LARGEST patch, and
    // "lowest-numbered cell owns the edge" gives early patches up to 1.7x the average (61 vs 36 edges at P = 12).
    std::vector<int32_t> ownerCell(nE);
    {
        const int P = p.P, nP = p.nPatches;
        std::vector<int32_t> cnt(nP, 0), flex;
        auto PO = [&](int cn) { return patchOf[cn]; };
        auto cls = [&](int cn) { return d->cellClass ? d->cellClass[p.cellN2O[cn]] : 0; };
        std::vector<int32_t> lo(nE), hi(nE);
        for (int e = 0; e < nE; ++e) {
            const int a = p.cellO2N[d->cellsOnEdge[2 * (int64_t)e] - 1], b = p.cellO2N[d->cellsOnEdge[2 * (int64_t)e + 1] - 1];
            lo[e] = std::min(a, b); hi[e] = std::max(a, b);
            if (PO(lo[e]) == PO(hi[e]) || cls(lo[e]) != cls(hi[e])) {   // class order == numbering order: lower class = lo
                ownerCell[e] = lo[e];
                ++cnt[PO(lo[e])];
            } else {
                ownerCell[e] = -1;
                flex.push_back(e);
            }
        }
        std::stable_sort(flex.begin(), flex.end(), [&](int x, int y) { return lo[x] < lo[y]; });
        for (int e : flex) {
            // start from a coin flip per edge (a hash: deterministic): unlike "the patch with fewer edges so far", which
            // drifts -- whole regions end one edge up, others one down, and levelling that takes mesh-wide paths -- it
            // leaves only local fluctuations, which the sweeps and the short breadth-first searches below remove
            const int pa = PO(lo[e]), pb = PO(hi[e]);
            if (((uint32_t)e * 2654435761u >> 15) & 1u) { ownerCell[e] = lo[e]; ++cnt[pa]; }
            else { ownerCell[e] = hi[e]; ++cnt[pb]; }
        }
        for (int sweep = 0; sweep < 16; ++sweep) {
            int moved = 0;
            for (int e : flex) {
                const int cur = ownerCell[e], oth = cur == lo[e] ? hi[e] : lo[e];
                if (cnt[PO(cur)] > cnt[PO(oth)] + 1) { ownerCell[e] = oth; --cnt[PO(cur)]; ++cnt[PO(oth)]; ++moved; }
            }
            if (!moved) break;
        }
        // Then level the remaining +-1 differences: a half-wave group handles one edge per iteration, so a patch costs
        // ceil(edges / groups) iterations and a patch one edge above the mean pays a whole iteration for it.  Every patch
        // above the mean looks (breadth first, over edges it could hand over) for the nearest patch below the mean and
        // passes one edge along that path.
        {
            // the mean is taken per cell class (partitioned meshes: boundary patches own every edge they share with interior
            // ones, halo patches are never launched -- one mean over all of them would miss the interior's)
            std::vector<int32_t> pcls(nP), Tc(64, 0);
            {
                std::vector<int64_t> tot(64, 0), num(64, 0);
                for (int q = 0; q < nP; ++q) {
                    pcls[q] = cls(p.patchCellStart[q]);
                    if (p.patchCellStart[q + 1] - p.patchCellStart[q] == P) { tot[pcls[q]] += cnt[q]; ++num[pcls[q]]; }   // full patches only
                }
                for (int c = 0; c < 64; ++c) Tc[c] = num[c] ? (int)((tot[c] + num[c] - 1) / num[c]) : 0;
            }
            auto T_of = [&](int q) { return Tc[pcls[q]]; };
            std::vector<int32_t> pstart(nP + 1, 0), plist;
            for (int e : flex) { ++pstart[PO(lo[e]) + 1]; ++pstart[PO(hi[e]) + 1]; }
            for (int q = 0; q < nP; ++q) pstart[q + 1] += pstart[q];
            plist.resize(pstart[nP]);
            {
                std::vector<int32_t> fill(pstart.begin(), pstart.end() - 1);
                for (int e : flex) { plist[fill[PO(lo[e])]++] = e; plist[fill[PO(hi[e])]++] = e; }
            }
            std::vector<int32_t> stamp(nP, -1), via(nP, -1), queue;
            int32_t tick = 0;
            for (int s = 0; s < nP; ++s) {
                for (int guard = 0; cnt[s] > T_of(s) && guard < 8; ++guard) {
                    queue.assign(1, s);
                    ++tick;
                    stamp[s] = tick; via[s] = -1;
                    int found = -1;
                    for (size_t h = 0; h < queue.size() && found < 0 && queue.size() < 2048; ++h) {
                        const int x = queue[h];
                        for (int k = pstart[x]; k < pstart[x + 1] && found < 0; ++k) {
                            const int e = plist[k];
                            if (PO(ownerCell[e]) != x) continue;                 // x can only hand over what it owns
                            const int y = PO(ownerCell[e] == lo[e] ? hi[e] : lo[e]);
                            if (stamp[y] == tick) continue;
                            stamp[y] = tick; via[y] = e;
                            if (cnt[y] < T_of(y)) found = y;
                            else queue.push_back(y);
                        }
                    }
                    if (found < 0) break;
                    for (int y = found; y != s;) {                               // pass one edge along every hop of the path
                        const int e = via[y];
                        const int x = PO(ownerCell[e]);
                        ownerCell[e] = ownerCell[e] == lo[e] ? hi[e] : lo[e];
                        --cnt[x]; ++cnt[y];
                        y = x;
                    }
                }
            }
        }
    }
    auto edgeOwner = renumber(nE, [&](int e) { return ownerCell[e]; }, p.edgeN2O, p.edgeO2N);
    auto vertOwner = renumber(nV, [&](int v) {
        int best = nC;
        for (int j = 0; j < VD; ++j) {
            if (d->cellsOnVertex) {
                int c = d->cellsOnVertex[IX(j, v, VD)];
                if (c >= 1 && c <= nC) best = std::min(best, (int)p.cellO2N[c - 1]);
            } else {
                int e = d->edgesOnVertex[IX(j, v, VD)] - 1;
                best = std::min(best, (int)p.cellO2N[d->cellsOnEdge[2 * (int64_t)e] - 1]);
                best = std::min(best, (int)p.cellO2N[d->cellsOnEdge[2 * (int64_t)e + 1] - 1]);
            }
        }
        return best == nC ? 0 : best;
    }, p.vertN2O, p.vertO2N);

    // ---- the edges / vertices the patches own ----
    {
        std::vector<int32_t> ce(p.nPatches + 1, 0), cv(p.nPatches + 1, 0);
        for (int e = 0; e < nE; ++e) ++ce[patchOf[edgeOwner[e]] + 1];
        for (int v = 0; v < nV; ++v) ++cv[patchOf[vertOwner[v]] + 1];
        for (int q = 0; q < p.nPatches; ++q) { ce[q + 1] += ce[q]; cv[q + 1] += cv[q]; }
        p.patchEdgeStart = ce;
        p.patchVertStart = cv;
        p.classEdgeStart.resize(p.classPatchStart.size());
        for (size_t k = 0; k < p.classPatchStart.size(); ++k) p.classEdgeStart[k] = ce[p.classPatchStart[k]];
    }

    // ---- records ----
    const int ME = p.ME, ME2 = p.ME2;
    p.eoc.assign((size_t)nC * ME, -1);
    p.coc.assign((size_t)nC * ME, -1);
    p.mltc.assign((size_t)nC * ME, 0);
    p.sdv.assign((size_t)nC * ME, 0.0);
    p.invArea.resize(nC); p.areaCell.resize(nC); p.rsum.resize(nC);
    p.cellBandwidth = 0;
    for (int cn = 0; cn < nC; ++cn) {
        const int co = p.cellN2O[cn];
        const int n = d->nEdgesOnCell[co];
        for (int i = 0; i < n; ++i) {
            const int eo = d->edgesOnCell[IX(i, co, d->maxEdges)] - 1;
            const int a = d->cellsOnEdge[2 * (int64_t)eo] - 1, b = d->cellsOnEdge[2 * (int64_t)eo + 1] - 1;
            REQUIRE(a == co || b == co, "edgesOnCell / cellsOnEdge are inconsistent");
            const int other = (a == co) ? b : a;
            p.eoc[IX(i, cn, ME)] = p.edgeO2N[eo];
            p.coc[IX(i, cn, ME)] = p.cellO2N[other];
            p.mltc[IX(i, cn, ME)] = d->maxLevelEdgeTop ? d->maxLevelEdgeTop[eo] : 1;
            p.sdv[IX(i, cn, ME)] = d->dvEdge[eo] * (double)d->edgeSignOnCell[IX(i, co, d->maxEdges)];
            p.cellBandwidth = std::max<int64_t>(p.cellBandwidth, std::abs((int64_t)p.cellO2N[other] - cn));
        }
        p.invArea[cn] = 1. / d->areaCell[co];
        p.areaCell[cn] = d->areaCell[co];
        p.rsum[cn] = d->restingThicknessSum[co];
    }
    p.ehdr.resize((size_t)nE * 4);
    p.eoe.assign((size_t)nE * ME2, -1);
    p.woe.assign((size_t)nE * ME2, 0.0);
    p.gInvDc.resize(nE); p.dcEdge.resize(nE); p.dvEdge.resize(nE); p.fEdge.resize(nE);
    for (int en = 0; en < nE; ++en) {
        const int eo = p.edgeN2O[en];
        const int n = d->nEdgesOnEdge[eo];
        p.ehdr[4 * (size_t)en + 0] = p.cellO2N[d->cellsOnEdge[2 * (int64_t)eo] - 1];
        p.ehdr[4 * (size_t)en + 1] = p.cellO2N[d->cellsOnEdge[2 * (int64_t)eo + 1] - 1];
        p.ehdr[4 * (size_t)en + 2] = n;
        p.ehdr[4 * (size_t)en + 3] = d->maxLevelEdgeTop ? d->maxLevelEdgeTop[eo] : 1;
        for (int i = 0; i < n; ++i) {
            const int x = d->edgesOnEdge[IX(i, eo, d->maxEdges2)];
            p.eoe[IX(i, en, ME2)] = x == 0 ? -1 : p.edgeO2N[x - 1];   // eoe == 0 => skipped (:67)
            p.woe[IX(i, en, ME2)] = d->weightsOnEdge[IX(i, eo, d->maxEdges2)];
        }
        const double invDc = 1. / d->dcEdge[eo];
        p.gInvDc[en] = 9.80616 * invDc;
        p.dcEdge[en] = d->dcEdge[eo];
        p.dvEdge[en] = d->dvEdge[eo];
        p.fEdge[en] = d->fEdge[eo];
    }
    p.eov.resize((size_t)nV * VD);
    p.cv.resize((size_t)nV * VD);
    for (int vn = 0; vn < nV; ++vn) {
        const int vo = p.vertN2O[vn];
        const double invA = 1.0 / d->areaTriangle[vo];
        for (int j = 0; j < VD; ++j) {
            const int eo = d->edgesOnVertex[IX(j, vo, VD)] - 1;
            p.eov[IX(j, vn, VD)] = p.edgeO2N[eo];
            p.cv[IX(j, vn, VD)] = (d->dcEdge[eo] * invA) * (double)d->edgeSignOnVertex[IX(j, vo, ldSV)];
        }
    }
    // ---- optional nonlinear terms: vertex-side connectivity and metric factors ----
    p.nlOk = d->kiteAreasOnVertex && d->fVertex && d->verticesOnEdge && d->cellsOnVertex;
    if (p.nlOk) {
        p.voe.resize((size_t)nE * 2); p.cov.resize((size_t)nV * VD); p.kite.resize((size_t)nV * VD);
        p.invAreaTri.resize(nV); p.fVertex.resize(nV); p.keCoef.resize(nE); p.invDc.resize(nE);
        for (int en = 0; en < nE; ++en) {
            const int eo = p.edgeN2O[en];
            for (int q = 0; q < 2; ++q) {
                const int v = d->verticesOnEdge[2 * (int64_t)eo + q];
                REQUIRE(v >= 1 && v <= nV, "verticesOnEdge out of range");
                p.voe[(size_t)en * 2 + q] = p.vertO2N[v - 1];
            }
            p.keCoef[en] = 0.25 * d->dcEdge[eo] * d->dvEdge[eo];
            p.invDc[en] = 1. / d->dcEdge[eo];
        }
        for (int vn = 0; vn < nV; ++vn) {
            const int vo = p.vertN2O[vn];
            p.invAreaTri[vn] = 1.0 / d->areaTriangle[vo];
            p.fVertex[vn] = d->fVertex[vo];
            for (int j = 0; j < VD; ++j) {
                const int c = d->cellsOnVertex[IX(j, vo, VD)];
                REQUIRE(c >= 1 && c <= nC, "cellsOnVertex out of range");
                p.cov[IX(j, vn, VD)] = p.cellO2N[c - 1];
                p.kite[IX(j, vn, VD)] = d->kiteAreasOnVertex[IX(j, vo, VD)];
            }
        }
    }

    // ---- packed byte-offset records of the column kernel ----
    {
        const uint64_t rowB = (uint64_t)p.K * p.stateBytes;
        p.colOk = rowB * (uint64_t)std::max(nE, std::max(nC, nV)) < (1ull << 32) - 1024;
        p.CI = ((2 * ME + 2) + 3) & ~3;
        p.EI = ((ME2 + 4) + 3) & ~3;
        p.cRec.clear(); p.eRec.clear(); p.feoe.clear();
        if (p.colOk) {
            p.cRec.assign((size_t)nC * p.CI, 0u);
            p.eRec.assign((size_t)nE * p.EI, 0u);
            p.feoe.assign((size_t)nE * ME2, 0.0);
            for (int c = 0; c < nC; ++c) {
                uint32_t *r = &p.cRec[(size_t)c * p.CI];
                uint32_t mask = 0, all = 1;
                for (int i = 0; i < ME; ++i) {
                    const int e = p.eoc[IX(i, c, ME)];
                    if (e >= 0) {
                        mask |= 1u << i;
                        r[i] = (uint32_t)((uint64_t)e * rowB);
                        r[ME + i] = (uint32_t)((uint64_t)p.coc[IX(i, c, ME)] * rowB);
                        if (p.mltc[IX(i, c, ME)] < p.K) all = 0;
                    } else {
                        r[i] = (uint32_t)((uint64_t)p.eoc[IX(0, c, ME)] * rowB);
                        r[ME + i] = (uint32_t)((uint64_t)c * rowB);
                    }
                }
                r[2 * ME] = mask;
                r[2 * ME + 1] = all;
            }
            for (int e = 0; e < nE; ++e) {
                uint32_t *r = &p.eRec[(size_t)e * p.EI];
                uint32_t mask = 0;
                for (int i = 0; i < ME2; ++i) {
                    const int x = p.eoe[IX(i, e, ME2)];
                    if (x >= 0) {
                        mask |= 1u << i;
                        r[i] = (uint32_t)((uint64_t)x * rowB);
                        p.feoe[IX(i, e, ME2)] = p.fEdge[x];
                    } else {
                        r[i] = (uint32_t)((uint64_t)e * rowB);
                    }
                }
                r[ME2] = (uint32_t)p.ehdr[4 * (size_t)e];
                r[ME2 + 1] = (uint32_t)p.ehdr[4 * (size_t)e + 1];
                r[ME2 + 2] = mask;
                r[ME2 + 3] = (uint32_t)p.ehdr[4 * (size_t)e + 3];
            }
        }
        p.vRec.clear();
        if (p.colOk && p.VD == 3) {
            p.vRec.assign((size_t)nV * 4, 0u);
            for (int v = 0; v < nV; ++v)
                for (int j = 0; j < 3; ++j) p.vRec[(size_t)v * 4 + j] = (uint32_t)((uint64_t)p.eov[(size_t)v * 3 + j] * rowB);
        }
        p.maxOwnV = 0;
        for (int q = 0; q < p.nPatches; ++q) p.maxOwnV = std::max(p.maxOwnV, p.patchVertStart[q + 1] - p.patchVertStart[q]);
    }

    // ---- patch-local row lists for the LDS-tiled kernel ----
    p.haloStart.assign(p.nPatches + 1, 0);
    p.haloEdge.clear();
    p.leoc.assign((size_t)nC * 8, 0xFF);
    p.leoe.assign((size_t)nE * 16, 0xFF);
    p.maxRows = p.maxOwnE = p.maxOwnC = 0;
    p.maxOwnELaunch = p.maxOwnCLaunch = 0;
    for (int q = 0; q < p.nPatches; ++q) {   // needed by every LDS-staging kernel: must cover ALL patches
        const int nE_q = p.patchEdgeStart[q + 1] - p.patchEdgeStart[q], nC_q = p.patchCellStart[q + 1] - p.patchCellStart[q];
        p.maxOwnE = std::max(p.maxOwnE, nE_q);
        p.maxOwnC = std::max(p.maxOwnC, nC_q);
        const bool launched = !d->cellClass || d->cellClass[p.cellN2O[p.patchCellStart[q]]] < 2;
        if (launched) {
            p.nPatchesLaunch = q + 1;
            p.maxOwnELaunch = std::max(p.maxOwnELaunch, nE_q);
            p.maxOwnCLaunch = std::max(p.maxOwnCLaunch, nC_q);
        }
    }
    p.ldsOk = (ME <= 8 && ME2 <= 16);
    {
        std::vector<int32_t> local(nE, -1), touched;
        for (int q = 0; q < p.nPatches && p.ldsOk; ++q) {
            const int c0 = p.patchCellStart[q], c1 = p.patchCellStart[q + 1];
            const int e0 = p.patchEdgeStart[q], e1 = p.patchEdgeStart[q + 1];
            const int nOwn = e1 - e0;
            int rows = nOwn;
            touched.clear();
            auto local_of = [&](int e) {
                if (e >= e0 && e < e1) return e - e0;
                if (local[e] < 0) { local[e] = rows++; touched.push_back(e); p.haloEdge.push_back(e); }
                return local[e];
            };
            for (int c = c0; c < c1; ++c)
                for (int i = 0; i < ME; ++i) {
                    const int e = p.eoc[IX(i, c, ME)];
                    if (e >= 0) { int r = local_of(e); p.leoc[(size_t)c * 8 + i] = (uint8_t)std::min(r, 255); }
                }
            for (int e = e0; e < e1; ++e)
                for (int i = 0; i < ME2; ++i) {
                    const int x = p.eoe[IX(i, e, ME2)];
                    if (x >= 0) { int r = local_of(x); p.leoe[(size_t)e * 16 + i] = (uint8_t)std::min(r, 255); }
                }
            for (int e : touched) local[e] = -1;
            p.haloStart[q + 1] = (int32_t)p.haloEdge.size();
            if (rows > 254) p.ldsOk = false;
            p.maxRows = std::max(p.maxRows, rows);
        }
        if (!p.ldsOk) { p.haloEdge.clear(); std::fill(p.haloStart.begin(), p.haloStart.end(), 0); }
        p.haloEdge.push_back(0);   // one element of slack: kernels read haloEdge[h0 + clamped index] unconditionally
    }
    p.rowStart.assign(p.nPatches + 1, 0);
    p.rowEdge.clear();
    if (p.ldsOk) {
        for (int q = 0; q < p.nPatches; ++q) {
            for (int e = p.patchEdgeStart[q]; e < p.patchEdgeStart[q + 1]; ++e) p.rowEdge.push_back(e);
            for (int j = p.haloStart[q]; j < p.haloStart[q + 1]; ++j) p.rowEdge.push_back(p.haloEdge[j]);
            p.rowStart[q + 1] = (int32_t)p.rowEdge.size();
        }
    }
    p.rowEdge.push_back(0);
    if (p.nlOk) {
        p.rowVoe.resize(2 * p.rowEdge.size());
        for (size_t r = 0; r < p.rowEdge.size(); ++r) {
            p.rowVoe[2 * r] = p.voe[2 * (size_t)p.rowEdge[r]]; p.rowVoe[2 * r + 1] = p.voe[2 * (size_t)p.rowEdge[r] + 1];
        }
        p.keoc.assign((size_t)nC * ME, 0.0);
        for (int c = 0; c < nC; ++c)
            for (int i = 0; i < ME; ++i)
                if (p.eoc[IX(i, c, ME)] >= 0) p.keoc[IX(i, c, ME)] = p.keCoef[p.eoc[IX(i, c, ME)]];
    }
    // ---- patch vertex lists of k_stage_nl5: the distinct vertices of a patch's own edges and of their edgesOnEdge, and per
    // own edge a record of 24 patch-local vertex ids (16 bits each: the boundary patches of a partitioned mesh can be scattered
    // cells with ~300 vertices): entries 2 i, 2 i + 1 = the two vertices of edgesOnEdge slot i (i < 10), entries 20, 21 = the
    // edge's own two vertices, 22, 23 unused ----
    p.pvStart.assign(p.nPatches + 1, 0);
    p.pvList.clear(); p.lvoe.clear();
    p.maxPV = 0; p.nl5Ok = false;
    if (p.nlOk && ME2 <= 10) {
        p.nl5Ok = true;
        p.lvoe.assign((size_t)nE * 24, 0);
        std::vector<int32_t> lv(p.nV, -1);
        for (int q = 0; q < p.nPatches; ++q) {
            const size_t base = p.pvList.size();
            auto lid = [&](int v) {
                if (lv[v] < 0) { lv[v] = (int32_t)(p.pvList.size() - base); p.pvList.push_back(v); }
                return (uint16_t)std::min(lv[v], 65535);
            };
            for (int e = p.patchEdgeStart[q]; e < p.patchEdgeStart[q + 1]; ++e) {
                uint16_t *r = &p.lvoe[(size_t)e * 24];
                r[20] = lid(p.voe[2 * (size_t)e]); r[21] = lid(p.voe[2 * (size_t)e + 1]);
                for (int i = 0; i < 10; ++i) {
                    const int x = i < ME2 ? p.eoe[IX(i, e, ME2)] : -1;
                    r[2 * i] = x >= 0 ? lid(p.voe[2 * (size_t)x]) : r[20];
                    r[2 * i + 1] = x >= 0 ? lid(p.voe[2 * (size_t)x + 1]) : r[21];
                }
            }
            const int n = (int)(p.pvList.size() - base);
            if (n > 65535) p.nl5Ok = false;
            p.maxPV = std::max(p.maxPV, n);
            for (size_t j = base; j < p.pvList.size(); ++j) lv[p.pvList[j]] = -1;
            p.pvStart[q + 1] = (int32_t)p.pvList.size();
        }
    }
    p.pvList.push_back(0);

    return MOKA_OK;
}

}  // namespace moka

// ------------------------------------------------------------------------------------------------
// C ABI: plan
// ------------------------------------------------------------------------------------------------
extern "C" {

int moka_plan_create(const moka_mesh_desc *desc, moka_plan **out)
{
    if (!out) { moka::set_error("out is NULL"); return MOKA_ERR_ARG; }
    *out = nullptr;
    moka_plan *pl = new (std::nothrow) moka_plan();
    if (!pl) { moka::set_error("out of host memory"); return MOKA_ERR_ALLOC; }
    int rc;
    try {
        rc = moka::build_plan(desc, pl->p);
    } catch (const std::bad_alloc &) {
        moka::set_error("out of host memory while building the mesh plan");
        rc = MOKA_ERR_ALLOC;
    }
    if (rc != MOKA_OK) { delete pl; return rc; }
    *out = pl;
    return MOKA_OK;
}

void moka_plan_destroy(moka_plan *plan) { delete plan; }

static int lanes_per_column(int K)
{
    int l = 1;
    while (l < K && l < 64) l <<= 1;
    return l;
}

static void fill_info(const moka::Plan &p, moka_mesh_info *info)
{
    info->nCells = p.nC; info->nEdges = p.nE; info->nVertices = p.nV; info->nVertLevels = p.K;
    info->ordering = p.ordering; info->patch_cells = p.P; info->nPatches = p.nPatches;
    info->maxEdgesUsed = p.ME; info->maxEdges2Used = p.ME2;
    info->lanesPerColumn = lanes_per_column(p.K);
    info->cellBandwidth = p.cellBandwidth;
    info->maxPatchRows = p.maxRows;
    info->maxPatchCells = p.maxOwnC;
    info->maxPatchEdges = p.maxOwnE;
    info->ldsBytesPerBlock = (p.ldsOk && p.K % 2 == 0)
                                 ? (int32_t)moka::lds_stage_bytes(p.K, p.ME, p.ME2, p.maxRows, p.maxOwnE, p.maxOwnC) : 0;
    info->meshBytesDevice =
        (int64_t)p.nC * (p.ME * (3 * 4 + 8) + 3 * 8) + (int64_t)p.nE * (16 + p.ME2 * 12 + 4 * 8) +
        (int64_t)p.nV * p.VD * 12 + (int64_t)(p.nPatches + 1) * 12 + 4ll * (p.nC + p.nE + p.nV);
}

int moka_plan_info(const moka_plan *plan, moka_mesh_info *info)
{
    if (!plan || !info) { moka::set_error("NULL argument"); return MOKA_ERR_ARG; }
    fill_info(plan->p, info);
    return MOKA_OK;
}

int moka_plan_permutation(const moka_plan *plan, int kind, int32_t *new_to_old)
{
    if (!plan || !new_to_old) { moka::set_error("NULL argument"); return MOKA_ERR_ARG; }
    const std::vector<int32_t> *v = kind == MOKA_CELL ? &plan->p.cellN2O : kind == MOKA_EDGE ? &plan->p.edgeN2O
                                  : kind == MOKA_VERTEX ? &plan->p.vertN2O : nullptr;
    if (!v) { moka::set_error("kind must be MOKA_CELL, MOKA_EDGE or MOKA_VERTEX"); return MOKA_ERR_ARG; }
    std::copy(v->begin(), v->end(), new_to_old);
    return MOKA_OK;
}

int moka_plan_patch_ranges(const moka_plan *plan, int32_t *cellStart, int32_t *edgeStart, int32_t *vertexStart)
{
    if (!plan) { moka::set_error("NULL argument"); return MOKA_ERR_ARG; }
    const moka::Plan &p = plan->p;
    if (cellStart) std::copy(p.patchCellStart.begin(), p.patchCellStart.end(), cellStart);
    if (edgeStart) std::copy(p.patchEdgeStart.begin(), p.patchEdgeStart.end(), edgeStart);
    if (vertexStart) std::copy(p.patchVertStart.begin(), p.patchVertStart.end(), vertexStart);
    return MOKA_OK;
}

int moka_plan_class_ranges(const moka_plan *plan, int32_t capacity, int32_t *nClasses, int32_t *patchStart, int32_t *cellStart,
                           int32_t *edgeStart)
{
    if (!plan || !nClasses) { moka::set_error("NULL argument"); return MOKA_ERR_ARG; }
    const moka::Plan &p = plan->p;
    const int n = (int)p.classPatchStart.size() - 1;
    *nClasses = n;
    for (int k = 0; k <= n && k < capacity; ++k) {
        if (patchStart) patchStart[k] = p.classPatchStart[k];
        if (cellStart) cellStart[k] = p.classCellStart[k];
        if (edgeStart) edgeStart[k] = p.classEdgeStart[k];
    }
    return MOKA_OK;
}

int moka_plan_array(const moka_plan *plan, int which, const void **data, int64_t *count)
{
    if (!plan || !data || !count) { moka::set_error("NULL argument"); return MOKA_ERR_ARG; }
    const moka::Plan &p = plan->p;
#define VEC(id, v) case id: *data = p.v.data(); *count = (int64_t)p.v.size(); return MOKA_OK;
    switch (which) {
        VEC(MOKA_PA_EOC, eoc) VEC(MOKA_PA_COC, coc) VEC(MOKA_PA_MLTC, mltc) VEC(MOKA_PA_SDV, sdv)
        VEC(MOKA_PA_INVAREA, invArea) VEC(MOKA_PA_AREACELL, areaCell) VEC(MOKA_PA_RSUM, rsum)
        VEC(MOKA_PA_EHDR, ehdr) VEC(MOKA_PA_EOE, eoe) VEC(MOKA_PA_WOE, woe) VEC(MOKA_PA_GINVDC, gInvDc)
        VEC(MOKA_PA_DCEDGE, dcEdge) VEC(MOKA_PA_DVEDGE, dvEdge) VEC(MOKA_PA_FEDGE, fEdge)
        VEC(MOKA_PA_EOV, eov) VEC(MOKA_PA_CV, cv)
        VEC(MOKA_PA_HALO_START, haloStart) VEC(MOKA_PA_HALO_EDGE, haloEdge) VEC(MOKA_PA_LEOC, leoc) VEC(MOKA_PA_LEOE, leoe)
        VEC(MOKA_PA_CREC, cRec) VEC(MOKA_PA_EREC, eRec) VEC(MOKA_PA_FEOE, feoe)
        VEC(MOKA_PA_PVSTART, pvStart) VEC(MOKA_PA_PVLIST, pvList) VEC(MOKA_PA_LVOE, lvoe)
        default: moka::set_error("unknown plan array id"); return MOKA_ERR_ARG;
    }
#undef VEC
}

}  // extern "C"

// shared with api.hip
namespace moka { void fill_mesh_info(const Plan &p, moka_mesh_info *info) { fill_info(p, info); } }
