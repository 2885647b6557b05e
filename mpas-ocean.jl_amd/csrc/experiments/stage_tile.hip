// stage_tile.hip -- "tile3": the fused tendency / RK-stage kernel with EVERY normalVelocity row a patch touches staged in
// LDS by LDS-DMA (global_load_lds_dwordx4: global -> LDS without passing through registers).
//
// Why (profiles/r02_variants.txt): the default kernel k_stage_rec2c caches a patch's own edge rows in LDS and gathers the
// other ~60 rows of a 16-cell patch from global memory inside its entity loops.  Two costs follow.  (1) Those gathers
// reach the L2 late -- up to a workgroup lifetime after the owner patch streamed the same rows -- so ~40 % of them miss
// and the rows cross the fabric a second time: 1.3-2.4 GB of the 6.2-12.7 GB a launch moves, and every stage launch runs
// at the copy ceiling for the bytes it moves.  (2) Every entity iteration is a dependent memory round trip (9 per
// workgroup), which bounds the tendency launch at ~1.0 ms even with its stores removed.
// Here a workgroup fetches all its rows at once, at its start, when the neighbouring patches of the same XCD are doing
// the same (the gathers meet the owners' reads in L2), and the entity loops read LDS only: a workgroup pays the staging
// round trip, one round trip for the layerThickness rows of its cells, and nothing else.
// The round-1 kernels of the same idea (csrc/experiments: tile, ptile, lds) staged through registers -- 68+ VGPRs held
// across the staging latency, two waves per SIMD -- and only tied; LDS-DMA keeps the register file out of it.
//
// Shape: NT threads = NT/32 half-wave groups, lane = two consecutive levels (16 B), patch = P cells + the edges they own.
// LDS image of the rows: pieces of 1 KiB = what one wave-instruction of LDS-DMA writes (lane i -> piece + 16 i); a piece
// holds rpp = 1024 / rowBytes whole rows (2 at K = 60: lanes 0-29 row A, 30-59 row B, 60-63 masked off), so no piece
// crosses a row and a lane's source address is its row's base + its chunk.  Local row r sits at (r / rpp) KiB + (r % rpp)
// rowBytes; the plan's patch-local row ids (leoe / leoc: own edges first, then the halo rows) are turned into these byte
// offsets once per patch while the rows are in flight.
// Arithmetic: k_stage_rec2c's, expression for expression (bit-identical to the oracle).
//
// RESULT (round 2, profiles/r02_variants.txt): correct, and slower -- an EXPERIMENT, built with `make VARIANTS=1` only
// (variants 12 = 256 threads, 13 = 512 threads per patch).  All rows of a 16-cell patch are 107 on average and 128 at most:
// 64 KB of row image + 12 KB of records per workgroup, i.e. ONE or two workgroups per CU instead of rec2c's four, and a
// workgroup idles through its staging round trip.  Config 4, ms per RK4 step: rec2c 6.99 (P = 16) / 7.22 (P = 12);
// tile3 with 512 threads 12.06 (P = 16) / 8.52 (P = 12), with 256 threads 17.7 / 11.8.  The LDS (160 KB per CU = 341 rows
// of 480 B) cannot hold the rows of enough patches to cover the staging latency; the register-free staging was not the
// missing piece of the round-1 tile kernels.
#include "../kernels_common.hpp"

namespace moka {

struct TileMesh {
    const int32_t *rowStart, *rowEdge;   // rows of patch p: rowEdge[rowStart[p] .. rowStart[p+1]) (global edge ids, own edges first)
    const uint8_t *leoe, *leoc;          // (16, nE) / (8, nC): local row id of every edgesOnEdge / edgesOnCell slot, 0xFF = none
    int32_t maxRows;
};

__device__ __forceinline__ uint32_t tile_row_off(uint32_t r, uint32_t rpp, uint32_t rowB) { return (r / rpp) * 1024u + (r % rpp) * rowB; }

template <int ME, int ME2, int MODE, int NT>
__global__ __launch_bounds__(NT) void k_stage_tile3(const ColMesh m, const TileMesh tm, const StageArgs a, int maxOwnE, int maxOwnC)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NG = NT / 32, NW = NT / 64;
    const int tid = threadIdx.x;
    const int grp = tid >> 5, l = tid & 31;
    const int K = m.K, K2 = K >> 1;
    const uint32_t voff = (uint32_t)l * 16u, rowB = (uint32_t)K * 8u, rpp = 1024u / rowB;
    const uint32_t nPiecesMax = ((uint32_t)tm.maxRows + rpp - 1) / rpp;
    unsigned char *ubase = smem;                                   // row image first (16-byte aligned), records behind it
    const RecLds L = rec_carve(smem + (size_t)nPiecesMax * 1024, m, ME, ME2, maxOwnE, maxOwnC);
    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
    const int nOwnC = c1 - c0, nOwnE = e1 - e0;
    const int rs0 = cptr(tm.rowStart)[p], R = cptr(tm.rowStart)[p + 1] - rs0;

    {   // ---- staging ----
        // (1) rows by LDS-DMA: wave w takes pieces w, w + NW, ...; nothing lands in a register
        const int lane = tid & 63, wave = tid >> 6;
        const uint32_t sub = (uint32_t)lane / (uint32_t)K2, chunk = (uint32_t)lane - sub * (uint32_t)K2;
        const uint32_t nPieces = ((uint32_t)R + rpp - 1) / rpp;
        for (uint32_t q = (uint32_t)wave; q < nPieces; q += NW) {
            const uint32_t row = q * rpp + sub;
            if (sub < rpp && row < (uint32_t)R) {
                const int32_t e = tm.rowEdge[rs0 + (int)row];
                const unsigned char *src = reinterpret_cast<const unsigned char *>(a.pu) + (size_t)e * rowB + chunk * 16u;
                // lane i's 16 bytes land at (piece base) + 16 i: the builtin sets M0 to the wave-uniform piece base
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(ubase + (size_t)q * 1024), 16, 0, 0);
            }
        }
        // (2) records through registers, every load issued before the first LDS write (see k_stage_rec2c); the u-row slots of
        //     eRec / cRec get the LOCAL byte offsets of the rows instead of the global ones
        const int nER = nOwnE * m.EI, nW = nOwnE * ME2, nCR = nOwnC * m.CI, nS = nOwnC * ME;
        constexpr int UE = (3 * 256 + NT - 1) / NT, UW = (2 * 256 + NT - 1) / NT;
        uint32_t vE[UE], vC;
        uint8_t vLe[UE], vLc;
        double vW[UW], vF[UW], vG, vS, vA, vR;
#pragma unroll
        for (int j = 0; j < UE; ++j) {
            const int i = tid + j * NT;
            vE[j] = i < nER ? m.eRec[(size_t)e0 * m.EI + i] : 0u;
            const int ee = i / m.EI, slot = i - ee * m.EI;
            vLe[j] = (i < nER && slot < ME2) ? tm.leoe[(size_t)(e0 + ee) * 16 + slot] : (uint8_t)0xFF;
        }
#pragma unroll
        for (int j = 0; j < UW; ++j) {
            vW[j] = (tid + j * NT < nW) ? m.woe[(size_t)e0 * ME2 + tid + j * NT] : 0.0;
            vF[j] = (tid + j * NT < nW) ? m.feoe[(size_t)e0 * ME2 + tid + j * NT] : 0.0;
        }
        vG = tid < nOwnE ? m.gInvDc[e0 + tid] : 0.0;
        vC = tid < nCR ? m.cRec[(size_t)c0 * m.CI + tid] : 0u;
        {
            const int cc = tid / m.CI, slot = tid - cc * m.CI;
            vLc = (tid < nCR && slot < ME) ? tm.leoc[(size_t)(c0 + cc) * 8 + slot] : (uint8_t)0xFF;
        }
        vS = tid < nS ? m.sdv[(size_t)c0 * ME + tid] : 0.0;
        vA = tid < nOwnC ? m.invArea[c0 + tid] : 0.0;
        vR = tid < nOwnC ? m.rsum[c0 + tid] : 0.0;
#pragma unroll
        for (int j = 0; j < UE; ++j) {
            const int i = tid + j * NT;
            if (i < nER) {
                const int ee = i / m.EI, slot = i - ee * m.EI;
                L.eRec[i] = slot < ME2 ? tile_row_off(vLe[j] == 0xFF ? 0u : vLe[j], rpp, rowB) : vE[j];
            }
        }
#pragma unroll
        for (int j = 0; j < UW; ++j)
            if (tid + j * NT < nW) {
                L.woe[tid + j * NT] = vW[j];
                L.feoe[tid + j * NT] = vF[j];
            }
        if (tid < nOwnE) L.g[tid] = vG;
        if (tid < nCR) {
            const int cc = tid / m.CI, slot = tid - cc * m.CI;
            L.cRec[tid] = slot < ME ? tile_row_off(vLc == 0xFF ? 0u : vLc, rpp, rowB) : vC;
        }
        if (tid < nS) L.sdv[tid] = vS;
        if (tid < nOwnC) {
            L.invA[tid] = vA;
            L.rsum[tid] = vR;
        }
        // larger patches than the unrolled part covers
        for (int i = tid + UE * NT; i < nER; i += NT) {
            const int ee = i / m.EI, slot = i - ee * m.EI;
            if (slot < ME2) {
                const uint8_t id = tm.leoe[(size_t)(e0 + ee) * 16 + slot];
                L.eRec[i] = tile_row_off(id == 0xFF ? 0u : id, rpp, rowB);
            } else {
                L.eRec[i] = m.eRec[(size_t)e0 * m.EI + i];
            }
        }
        for (int i = tid + UW * NT; i < nW; i += NT) {
            L.woe[i] = m.woe[(size_t)e0 * ME2 + i];
            L.feoe[i] = m.feoe[(size_t)e0 * ME2 + i];
        }
        for (int i = tid + NT; i < nOwnE; i += NT) L.g[i] = m.gInvDc[e0 + i];
        for (int i = tid + NT; i < nCR; i += NT) {
            const int cc = i / m.CI, slot = i - cc * m.CI;
            if (slot < ME) {
                const uint8_t id = tm.leoc[(size_t)(c0 + cc) * 8 + slot];
                L.cRec[i] = tile_row_off(id == 0xFF ? 0u : id, rpp, rowB);
            } else {
                L.cRec[i] = m.cRec[(size_t)c0 * m.CI + i];
            }
        }
        for (int i = tid + NT; i < nS; i += NT) L.sdv[i] = m.sdv[(size_t)c0 * ME + i];
        for (int i = tid + NT; i < nOwnC; i += NT) {
            L.invA[i] = m.invArea[c0 + i];
            L.rsum[i] = m.rsum[c0 + i];
        }
        // the LDS-DMA transfers are counted by vmcnt, and a barrier does not wait for them
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __syncthreads();

    const int k0 = 2 * l;
    const bool act = k0 < K;
    const uint32_t ldsU = (uint32_t)(size_t)(lds_bytes_t)ubase + voff;

    constexpr bool FE = MODE >= 4, STALE = MODE == 4;
    double2 pA = make_double2(0.0, 0.0), pB = pA, pD = pA, pE = pA;
    double pS = 0.0;
    uint32_t pOff = 0;
    int pC = 0;
    bool pend = false;
    auto flush_cell = [&]() {
        if (act) {
            if constexpr (MODE == 0) gstore2o(a.tendH, pOff, pA);
            if constexpr (MODE == 1 || MODE == 2) {
                gstore2o(a.ph_out, pOff, pA);
                gstore2o(a.nh_out, pOff, pB);
            }
            if constexpr (MODE == 3) gstore2o(a.nh_out, pOff, pB);
            if constexpr (FE) {
                gstore2o(a.ph_out, pOff, pA);
                gstore2o(a.tendH, pOff, pB);
                gstore2o(a.div, pOff, pD);
            }
        }
        if constexpr (MODE != 0)
            if (l == 0) a.ssh_out[pC] = pS;
    };
    // ---------------- cells: u rows from LDS, layerThickness rows from global memory (one round trip per iteration) ----
#pragma nounroll
    for (int ci = grp; ci < nOwnC; ci += NG) {
        const int c = c0 + ci;
        const uint32_t *r = L.cRec + (size_t)ci * m.CI;
        const double *rs = L.sdv + (size_t)ci * ME;
        const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
        const double invA = L.invA[ci];
        const uint32_t own = (uint32_t)c * rowB + voff;
        double2 hc = make_double2(0.0, 0.0), uv[ME], hv[ME], cur = hc, nin = hc;
        if (act) {
            uint32_t ad[ME];
            v4u_t raw[ME];
            hc = gload2(a.ph, own);
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                hv[i] = STALE ? gload2(a.hEdgeOld, cptr(m.cRec)[(size_t)c * m.CI + i] + voff) : gload2(a.ph, r[ME + i] + voff);
                ad[i] = ldsU + r[i];
            }
            lds_burst<ME>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME; ++i) uv[i] = __builtin_bit_cast(double2, raw[i]);
            if constexpr (MODE == 2) cur = gload2(a.ch, own);
            if constexpr (MODE == 2 || MODE == 3) nin = gload2(a.nh_in, own);
        }
        double area = 0.0;
        if constexpr (FE) area = a.areaCell[c];
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (pend) flush_cell();
        double2 t = make_double2(0.0, 0.0);
        const bool plain = __builtin_amdgcn_ballot_w64(!(mask == (1u << ME) - 1u && all)) == 0;
        double2 dv = make_double2(0.0, 0.0);
        auto hE = [&](int i) { return STALE ? hv[i] : make_double2(0.5 * (hc.x + hv[i].x), 0.5 * (hc.y + hv[i].y)); };
        if (plain) {
            if (act) {
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    const double2 he = hE(i);
                    t.x += uv[i].x * he.x * rs[i] * invA;
                    t.y += uv[i].y * he.y * rs[i] * invA;
                    if constexpr (FE) {
                        dv.x -= uv[i].x * rs[i];
                        dv.y -= uv[i].y * rs[i];
                    }
                }
            }
        } else if (act) {
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                const int ml = all ? K : cptr(m.mltc)[(size_t)c * ME + i];
                const bool on = (mask >> i) & 1u;
                const double2 he = hE(i);
                const double dx = uv[i].x * he.x * rs[i] * invA;   // Operators.jl:217, DiagnosticVars.jl:165,
                const double dy = uv[i].y * he.y * rs[i] * invA;   // horizontal_advection.jl:63
                if (on && k0 < ml) t.x += dx;
                if (on && k0 + 1 < ml) t.y += dy;
                if constexpr (FE) {
                    if (on) {
                        dv.x -= uv[i].x * rs[i];
                        dv.y -= uv[i].y * rs[i];
                    }
                }
            }
        }
        double2 hs = make_double2(0.0, 0.0);
        if (act) {
            if constexpr (MODE == 0) pA = t;
            if constexpr (MODE == 1 || MODE == 2) {
                const double2 hcur = MODE == 2 ? cur : hc;
                const double2 nb = MODE == 2 ? nin : hcur;
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);                    // time_integration.jl:125
                pA = hs;
                pB = make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y);                        // :135
            }
            if constexpr (MODE == 3) {
                hs = make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y);
                pB = hs;
            }
            if constexpr (FE) {
                hs = make_double2(hc.x + a.a * t.x, hc.y + a.a * t.y);                        // time_integration.jl:199
                pA = hs;
                pB = t;
                pD = make_double2(dv.x / area, dv.y / area);                                  // Operators.jl:41
            }
        }
        if constexpr (MODE != 0) {
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) {                   // oracle_ksum order
                const double ox = __shfl_xor(hs.x, sft, 32), oy = __shfl_xor(hs.y, sft, 32);
                hs = make_double2(hs.x + ox, hs.y + oy);
            }
            pS = (hs.x + hs.y) - L.rsum[ci];                                                  // :209 (+N3)
        }
        pOff = own;
        pC = c;
        pend = true;
    }
    if (pend) flush_cell();
    pend = false;
    auto flush_edge = [&]() {
        if (act) {
            if constexpr (MODE == 0) gstore2o(a.tendU, pOff, pA);
            if constexpr (MODE == 1 || MODE == 2) {
                gstore2o(a.pu_out, pOff, pA);
                gstore2o(a.nu_out, pOff, pB);
            }
            if constexpr (MODE == 3) gstore2o(a.nu_out, pOff, pB);
            if constexpr (FE) {
                gstore2o(a.pu_out, pOff, pA);
                gstore2o(a.tendU, pOff, pB);
                gstore2o(a.F, pOff, pD);
                gstore2o(a.hEdgeNew, pOff, pE);
            }
        }
    };

    // ---------------- edges: every normalVelocity row from LDS; only the edge's own Curr / New rows and ssh are loaded ----
#pragma nounroll
    for (int ei = grp; ei < nOwnE; ei += NG) {
        const int e = e0 + ei;
        const uint32_t *r = L.eRec + (size_t)ei * m.EI;
        const double *rw = L.woe + (size_t)ei * ME2;
        const double *rf = L.feoe + (size_t)ei * ME2;
        const uint32_t mask = r[ME2 + 2];
        const int mlt = (int)r[ME2 + 3];
        const double g = L.g[ei];
        const uint32_t own = (uint32_t)e * rowB + voff;
        double sv = 0.0;
        double2 uv[ME2], cur = make_double2(0.0, 0.0), nin = cur, hx = cur, hy = cur, hEo = cur, up = cur;
        if (act) {
            uint32_t ad[ME2];
            v4u_t raw[ME2];
            if constexpr (MODE == 2) cur = gload2(a.cu, own);
            if constexpr (MODE == 2 || MODE == 3) nin = gload2(a.nu_in, own);
            if constexpr (FE) {
                hx = gload2(a.ph, r[ME2] * rowB + voff);               // layerThickness of cellsOnEdge[1], [2]
                hy = gload2(a.ph, r[ME2 + 1] * rowB + voff);
                if constexpr (STALE) hEo = gload2(a.hEdgeOld, own);
            }
#pragma unroll
            for (int i = 0; i < ME2; ++i) ad[i] = ldsU + r[i];
            lds_burst<ME2>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME2; ++i) uv[i] = __builtin_bit_cast(double2, raw[i]);
            if constexpr (MODE == 1 || FE)                              // the edge's own row is local row ei
                up = __builtin_bit_cast(double2, *(const __attribute__((address_space(3))) v2d_t *)((lds_bytes_t)ubase +
                                                                                                    tile_row_off((uint32_t)ei, rpp, rowB) + voff));
        }
        if (l < 2) sv = a.ssh[r[ME2 + l]];                             // ssh of cellsOnEdge[l]
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (pend) flush_edge();
        const double ds = __shfl(sv, 1, 32) - __shfl(sv, 0, 32);       // ssh[c2] - ssh[c1]
        const bool plain = __builtin_amdgcn_ballot_w64(!(mask == (1u << ME2) - 1u && mlt >= K)) == 0;   // wave-uniform
        if (act) {
            const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
            double2 t = make_double2(0.0, 0.0);
            if (plain) {
                t.x -= g * ds;
                t.y -= g * ds;
#pragma unroll
                for (int i = 0; i < ME2; ++i) {
                    t.x += rw[i] * uv[i].x * rf[i];
                    t.y += rw[i] * uv[i].y * rf[i];
                }
            } else {
                if (ax) t.x -= g * ds;                                 // pressure_gradient.jl:63
                if (ay) t.y -= g * ds;
#pragma unroll
                for (int i = 0; i < ME2; ++i) {
                    const bool on = (mask >> i) & 1u;
                    const double px = rw[i] * uv[i].x * rf[i], py = rw[i] * uv[i].y * rf[i];   // ...coriolis.jl:70-72
                    if (on && ax) t.x += px;
                    if (on && ay) t.y += py;
                }
            }
            if constexpr (MODE == 0) pA = t;
            if constexpr (MODE == 1) {
                pA = make_double2(up.x + a.a * t.x, up.y + a.a * t.y);  // time_integration.jl:124
                pB = make_double2(up.x + a.b * t.x, up.y + a.b * t.y);  // :134
            }
            if constexpr (MODE == 2) {
                pA = make_double2(cur.x + a.a * t.x, cur.y + a.a * t.y);
                pB = make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y);
            }
            if constexpr (MODE == 3) pB = make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y);
            if constexpr (FE) {
                pE = make_double2(0.5 * (hx.x + hy.x), 0.5 * (hx.y + hy.y));              // layerThicknessEdge, Operators.jl:217
                const double2 hF = STALE ? hEo : pE;
                pD = make_double2(up.x * hF.x, up.y * hF.y);                              // thicknessFlux, DiagnosticVars.jl:165
                pA = make_double2(up.x + a.a * t.x, up.y + a.a * t.y);                    // time_integration.jl:199
                pB = t;
            }
        }
        pOff = own;
        pend = true;
    }
    if (pend) flush_edge();
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
size_t tile3_lds_bytes(const MeshDev &md)
{
    const size_t rowB = (size_t)md.K * 8, rpp = 1024 / rowB;
    return ((size_t)md.maxRows + rpp - 1) / rpp * 1024 + rec_lds_bytes(md) + 16;
}

bool stage_tile3_usable(const MeshDev &md)
{
    return md.cRec && md.eRec && md.rowEdge && md.leoe && md.K >= 2 && md.K <= 64 && !(md.K & 1) && md.maxRows >= 1 && md.maxRows <= 254 &&
           tile3_lds_bytes(md) <= 160 * 1024 && md.maxOwnC >= 1 && md.maxOwnE >= 1 &&
           ((md.ME == 6 && md.ME2 == 10) || (md.ME == 8 && md.ME2 == 14) || (md.ME <= 6 && md.ME2 <= 14));
}

template <int ME, int ME2, int NT>
static hipError_t launch_tile3(const ColMesh &m, const TileMesh &tm, const StageArgs &a, int mode, dim3 g, size_t lds, int mE, int mC, hipStream_t s)
{
    static size_t raised = 0;
    if (lds > 64 * 1024 && lds > raised) {       // more than 64 KB of dynamic LDS needs the opt-in attribute
        hipError_t e = hipSuccess;
#define RAISE(MODE) if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_stage_tile3<ME, ME2, MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        RAISE(0) RAISE(1) RAISE(2) RAISE(3) RAISE(4) RAISE(5)
#undef RAISE
        if (e != hipSuccess) return e;
        raised = 160 * 1024;
    }
    const dim3 b(NT);
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_tile3<ME, ME2, 0, NT>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 1: hipLaunchKernelGGL((k_stage_tile3<ME, ME2, 1, NT>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 2: hipLaunchKernelGGL((k_stage_tile3<ME, ME2, 2, NT>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 3: hipLaunchKernelGGL((k_stage_tile3<ME, ME2, 3, NT>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 4: hipLaunchKernelGGL((k_stage_tile3<ME, ME2, 4, NT>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 5: hipLaunchKernelGGL((k_stage_tile3<ME, ME2, 5, NT>), g, b, lds, s, m, tm, a, mE, mC); break;
        default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

// threads: 256 or 512 per patch
hipError_t launch_stage_tile3(const MeshDev &md, const StageArgs &a, int threads, hipStream_t s)
{
    const int mode = colp_mode(a);
    if (mode < 0 || !stage_tile3_usable(md) || md.tailPatch >= 0) return hipErrorNotSupported;
    const dim3 g(8 * ((md.nPatches + 7) / 8));
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc, 0};
    const TileMesh tm{md.rowStart, md.rowEdge, md.leoe, md.leoc, md.maxRows};
    const size_t lds = tile3_lds_bytes(md);
#define GO(ME, ME2)                                                                                           \
    return threads == 512 ? launch_tile3<ME, ME2, 512>(m, tm, a, mode, g, lds, md.maxOwnE, md.maxOwnC, s)     \
                          : launch_tile3<ME, ME2, 256>(m, tm, a, mode, g, lds, md.maxOwnE, md.maxOwnC, s);
    if (md.ME == 6 && md.ME2 == 10) { GO(6, 10) }
    if (md.ME == 8 && md.ME2 == 14) { GO(8, 14) }
    if (md.ME <= 6 && md.ME2 <= 14) { GO(6, 14) }
#undef GO
    return hipErrorNotSupported;
}

}  // namespace moka
