// moka_internal.hpp -- shared declarations of libmoka_hip (host plan + device views).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/moka_hip.h"

namespace moka {

void set_error(const std::string &msg);   // thread-local last error
const char *get_error();

// ----------------------------------------------------------------------------------------------
// Host-side reordered mesh.  All indices 0-based in the NEW numbering, -1 = none / padding.
// Records are entity-major with the slot index fastest (one contiguous record per entity): the
// column kernels give one wavefront (or one LPC-lane group) to an entity, so its record is read
// with wave-uniform scalar loads.
// ----------------------------------------------------------------------------------------------
struct Plan {
    int32_t nC = 0, nE = 0, nV = 0, K = 1;
    int32_t ME = 0;    // record width for cells  (>= max nEdgesOnCell, 6 or 8 or generic)
    int32_t ME2 = 0;   // record width for edges  (>= max nEdgesOnEdge)
    int32_t VD = 3;
    int32_t ordering = 0, P = 0, nPatches = 0;
    int32_t stateBytes = 8;           // bytes per real of the prognostic state (row offsets in cRec/eRec use it)
    int64_t cellBandwidth = 0;

    std::vector<int32_t> cellN2O, cellO2N, edgeN2O, edgeO2N, vertN2O, vertO2N;
    std::vector<int32_t> patchCellStart, patchEdgeStart, patchVertStart;   // nPatches + 1
    // cell classes (moka_mesh_desc.cellClass; one class without it): class k covers patches [classPatchStart[k], [k+1]),
    // cells [classCellStart[k], [k+1]) and -- edges being numbered by their owner cell -- edges [classEdgeStart[k], [k+1])
    std::vector<int32_t> classPatchStart, classCellStart, classEdgeStart;   // nClasses + 1

    // cells
    std::vector<int32_t> eoc;      // nC*ME  edge of slot i                (edgesOnCell)
    std::vector<int32_t> coc;      // nC*ME  the cell across that edge     (from cellsOnEdge)
    std::vector<int32_t> mltc;     // nC*ME  maxLevelEdgeTop of that edge (0 for padding)
    std::vector<double>  sdv;      // nC*ME  dvEdge[e]*edgeSignOnCell[i,c] (exact: sign = +-1)
    std::vector<double>  invArea;  // nC     1/areaCell      (horizontal_advection.jl:53)
    std::vector<double>  areaCell; // nC                    (Operators.jl:41 divides by it)
    std::vector<double>  rsum;     // nC     restingThicknessSum
    // edges
    std::vector<int32_t> ehdr;     // nE*4   {c1, c2, nEdgesOnEdge, maxLevelEdgeTop}
    std::vector<int32_t> eoe;      // nE*ME2 edgesOnEdge (-1: zero entry or beyond nEdgesOnEdge)
    std::vector<double>  woe;      // nE*ME2 weightsOnEdge
    std::vector<double>  gInvDc;   // nE     9.80616 * (1/dcEdge)   (pressure_gradient.jl:58,63)
    std::vector<double>  dcEdge, dvEdge, fEdge;   // nE
    // packed records of the column kernel (LPC = 64): 32-bit BYTE offsets of the neighbour rows, so that
    // a gather is one buffer_load with the offset in an SGPR and no address arithmetic at all.
    //   cRec[c][CI]: [0,ME) u-row offsets of the cell's edges | [ME,2ME) h-row offsets of the cells across
    //                | [2ME] valid-slot mask | [2ME+1] 1 if every valid slot has maxLevelEdgeTop >= K
    //   eRec[e][EI]: [0,ME2) u-row offsets of edgesOnEdge | c1 | c2 | valid-slot mask | maxLevelEdgeTop
    // invalid slots carry the entity's own (valid) offset.  feoe = fEdge[edgesOnEdge] per slot.
    std::vector<uint32_t> cRec, eRec;
    std::vector<double>   feoe;       // nE*ME2
    // vRec[v][4] (vertexDegree 3 only): u-row byte offsets of the vertex's three edges | 0 -- the relativeVorticity pass of the
    // Forward-Euler modes of the stage kernels (weights: cv)
    std::vector<uint32_t> vRec;
    int32_t maxOwnV = 0;              // most vertices any patch owns
    int32_t CI = 0, EI = 0;
    bool colOk = false;               // K*stateBytes*nE < 4 GiB: offsets fit 32 bits
    // patch-local view (the nonlinear stage kernel's LDS q_e rows; host tests): the u-rows a patch needs are its own edges
    // [patchEdgeStart[p], patchEdgeStart[p+1]) followed by haloEdge[haloStart[p] .. haloStart[p+1]);
    // leoc / leoe hold, per cell slot / edgesOnEdge slot, the row index inside that list (0xFF = none).
    std::vector<int32_t> rowStart;    // nPatches + 1 : the full staged-row list (own edges then halo edges) ...
    std::vector<int32_t> rowEdge;     // ... as explicit edge ids, so kernels read it with one unconditional load
    std::vector<int32_t> haloStart;   // nPatches + 1
    std::vector<int32_t> haloEdge;    // new edge ids
    std::vector<uint8_t> leoc;        // nC*8
    std::vector<uint8_t> leoe;        // nE*16
    int32_t maxRows = 0, maxOwnE = 0, maxOwnC = 0;
    // the same maxima over the patches that are ever launched (those starting with a cell of class < 2: halo-only
    // patches of a partitioned mesh own up to 6 edges per cell and are never computed)
    int32_t maxOwnELaunch = 0, maxOwnCLaunch = 0, nPatchesLaunch = 0;   // launchable patches are a prefix (class-major order)
    bool ldsOk = false;               // every patch has <= 254 rows
    // vertices
    std::vector<int32_t> eov;      // nV*VD  edgesOnVertex
    std::vector<double>  cv;       // nV*VD  (dcEdge[e]*(1/areaTriangle[v]))*sign  (Operators.jl:137-146)
    // optional nonlinear terms (moka_set_nonlinear): present when the mesh brought kiteAreasOnVertex and fVertex
    bool nlOk = false;
    std::vector<int32_t> voe;      // nE*2   verticesOnEdge
    std::vector<int32_t> cov;      // nV*VD  cellsOnVertex
    std::vector<double>  kite;     // nV*VD  kiteAreasOnVertex
    std::vector<double>  invAreaTri, fVertex;   // nV
    std::vector<double>  keCoef, invDc;         // nE  0.25*dcEdge*dvEdge ; 1/dcEdge
    std::vector<double>  keoc;     // nC*ME  keCoef of the cell's edge in slot i (0 for padding): no dependent load in the kernels
    std::vector<int32_t> rowVoe;   // 2 per entry of rowEdge: verticesOnEdge of that row's edge (patch row lists only)
    // k_stage_nl5: pvList[pvStart[q] .. pvStart[q+1]) = the distinct vertices of patch q's own edges and of their edgesOnEdge;
    // lvoe[e][24] = patch-local ids (16 bits each) of the two vertices of every edgesOnEdge slot and of e itself (layout: plan.cpp)
    std::vector<int32_t> pvStart, pvList;
    std::vector<uint16_t> lvoe;
    int32_t maxPV = 0;
    bool nl5Ok = false;
};

int build_plan(const moka_mesh_desc *d, Plan &out);   // returns moka_status

// ----------------------------------------------------------------------------------------------
// Device view handed to kernels (raw device pointers, by value).
// ----------------------------------------------------------------------------------------------
struct MeshDev {
    int32_t nC, nE, nV, K, ME, ME2, VD, nPatches;
    int32_t patchBegin;   // first patch of this launch (nPatches = patches in the launch)
    const int32_t *patchCellStart, *patchEdgeStart, *patchVertStart;
    const int32_t *eoc, *coc, *mltc;
    const double  *sdv, *invArea, *areaCell, *rsum;
    const int32_t *ehdr, *eoe;
    const double  *woe, *gInvDc, *dcEdge, *dvEdge, *fEdge;
    const int32_t *eov;
    const double  *cv;
    const int32_t *cellN2O, *edgeN2O, *vertN2O;
    // column kernel records
    const uint32_t *cRec, *eRec;
    const double *feoe;
    const uint32_t *vRec;
    int32_t maxOwnV;
    int32_t CI, EI;
    // patch row lists (the nonlinear stage kernel with q_e rows in LDS)
    const int32_t *rowStart, *rowEdge;
    const uint8_t *leoe;
    int32_t maxRows, maxOwnE, maxOwnC;
    // optional nonlinear terms (nullptr when the mesh did not bring them)
    const int32_t *voe, *cov;
    const double *kite, *invAreaTri, *fVertex, *keCoef, *invDc, *keoc;
    const int32_t *rowVoe;
    const int32_t *pvStart, *pvList;   // k_stage_nl5 (nullptr when the plan could not build them)
    const uint16_t *lvoe;
    int32_t maxPV, pvCap;              // most vertices any patch lists; rows k_stage_nl5 keeps in LDS (set per launch)
    int32_t tailPatch;    // >= 0: one extra, non-adjacent patch rides in this launch (default stage kernels only)
};

// Dynamic LDS the LDS-tiled stage kernel carves up (same formula on host and device):
//   u rows [maxRows][K] | fEdge [maxRows] | weightsOnEdge [maxOwnE][ME2] | gInvDc [maxOwnE] |
//   sdv [maxOwnC][ME] | invArea [maxOwnC] | rsum [maxOwnC] | ehdr [maxOwnE][4] i32 | coc [maxOwnC][ME] i32 |
//   mltc [maxOwnC][ME] i32 | leoe [maxOwnE][16] u8 | leoc [maxOwnC][8] u8
inline int64_t lds_stage_bytes(int K, int ME, int ME2, int maxRows, int maxOwnE, int maxOwnC)
{
    int64_t dbl = (int64_t)maxRows * K + maxRows + (int64_t)maxOwnE * ME2 + maxOwnE + (int64_t)maxOwnC * ME + 2 * maxOwnC;
    int64_t i32 = (int64_t)maxOwnE * 4 + 2 * (int64_t)maxOwnC * ME + (int64_t)maxOwnE * 4 + (int64_t)maxOwnC * 2;
    return dbl * 8 + i32 * 4 + 16;
}

}  // namespace moka

struct moka_plan {
    moka::Plan p;
};
