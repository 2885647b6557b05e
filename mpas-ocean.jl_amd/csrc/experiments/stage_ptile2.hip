// stage_ptile2.hip -- "ptile2": persistent, double-buffered form of tile3 (stage_tile.hip).
//
// One workgroup per CU walks a sequence of patches.  Eight compute waves (16 half-wave groups) evaluate patch i from LDS
// buffer i & 1 while ONE loader wave fills the other buffer with patch i + 1 by LDS-DMA: every normalVelocity row the patch
// touches and its records (the tile-layout copies eRecT / cRecT of the plan, so nothing has to be computed on the way).
// vmcnt is per wave, so the loader's transfers never sit in front of a compute wave's own loads.  One barrier per patch:
// behind it the next patch has landed and the buffer of the finished one is free.
// The 32 workgroups of an XCD take consecutive patches at the same time, so the rows neighbouring patches share are
// fetched within microseconds of each other and meet in that XCD's L2.
// Arithmetic: k_stage_rec2c's, expression for expression.
#include "../kernels_common.hpp"

namespace moka {

struct PTileMesh {
    const int32_t *rowStart, *rowEdge;
    const uint32_t *eRecT, *cRecT;
    int32_t maxRows;
};

constexpr int PT_CW = 15;                // compute waves
constexpr int PT_NT = (PT_CW + 1) * 64;  // + one loader wave
constexpr int PT_NG = PT_CW * 2;         // half-wave groups
constexpr int PT_MAXHP = 48;             // halo pieces a patch may have (rows beyond the own ones, rpp per piece)
constexpr int PT_NCI = 1, PT_NEI = 2;    // cells / edges a half-wave group handles per patch: patches of <= 16 cells, <= 64 own edges

struct PRec {                            // byte offsets of the record arrays inside a buffer's record area (all 16-byte aligned)
    uint32_t woe, feoe, g, sdv, invA, rsum, eRec, cRec, bytes;
};
__host__ __device__ inline uint32_t al16(uint32_t x) { return (x + 15u) & ~15u; }
__host__ __device__ inline PRec prec_layout(int ME, int ME2, int EI, int CI, int maxOwnE, int maxOwnC)
{
    PRec r;
    uint32_t o = 0;
    r.woe = o; o = al16(o + (uint32_t)maxOwnE * ME2 * 8);
    r.feoe = o; o = al16(o + (uint32_t)maxOwnE * ME2 * 8);
    r.g = o; o = al16(o + (uint32_t)maxOwnE * 8);
    r.sdv = o; o = al16(o + (uint32_t)maxOwnC * ME * 8);
    r.invA = o; o = al16(o + (uint32_t)maxOwnC * 8);
    r.rsum = o; o = al16(o + (uint32_t)maxOwnC * 8);
    r.eRec = o; o = al16(o + (uint32_t)maxOwnE * EI * 4);
    r.cRec = o; o = al16(o + (uint32_t)maxOwnC * CI * 4);
    r.bytes = o;
    return r;
}

typedef __attribute__((address_space(3))) unsigned char *lds_wr_t;
typedef const __attribute__((address_space(1))) unsigned char *glb_rd_t;

// contiguous copy global -> LDS by one wave: 16 bytes per lane and instruction (nbytes % 16 == 0, both sides 16-byte aligned)
__device__ __forceinline__ void dma16(lds_wr_t dst, const void *src, uint32_t nbytes, int lane)
{
    for (uint32_t off = 0; off < nbytes; off += 1024u)
        if (off + (uint32_t)lane * 16u < nbytes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((glb_rd_t)src + off + lane * 16),
                                             (__attribute__((address_space(3))) void *)(dst + off), 16, 0, 0);
}
// the same with 4 bytes per lane (8-byte aligned sources: gInvDc, invArea, restingThicknessSum)
__device__ __forceinline__ void dma4(lds_wr_t dst, const void *src, uint32_t nbytes, int lane)
{
    for (uint32_t off = 0; off < nbytes; off += 256u)
        if (off + (uint32_t)lane * 4u < nbytes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((glb_rd_t)src + off + lane * 4),
                                             (__attribute__((address_space(3))) void *)(dst + off), 4, 0, 0);
}

template <int ME, int ME2, int MODE>
__global__ __launch_bounds__(PT_NT) void k_stage_ptile2(const ColMesh m, const PTileMesh tm, const StageArgs a, int maxOwnE, int maxOwnC)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = m.K, K2 = K >> 1;
    const uint32_t rowB = (uint32_t)K * 8u, rpp = 1024u / rowB;
    const uint32_t nPiecesMax = ((uint32_t)tm.maxRows + rpp - 1) / rpp;
    const PRec R = prec_layout(ME, ME2, m.EI, m.CI, maxOwnE, maxOwnC);
    const uint32_t bufB = nPiecesMax * 1024u + R.bytes;
    // patch sequence of this workgroup: the WGs of an XCD (blockIdx % 8, observed round-robin dispatch; speed only) take
    // consecutive patches of that XCD's chunk
    const int xcd = (int)(blockIdx.x & 7), jw = (int)(blockIdx.x >> 3), W = (int)(gridDim.x >> 3);
    const int chunk = (m.nPatches + 7) >> 3;
    const int base = xcd * chunk, lim = min(chunk, m.nPatches - base);
    const int nSeq = lim > jw ? (lim - jw + W - 1) / W : 0;
    if (nSeq <= 0) return;
    auto patch_at = [&](int i) { return base + jw + W * i + m.patchBegin; };

    if (wave == PT_CW) {
        // ================= loader wave =================
        const uint32_t sub = (uint32_t)lane / (uint32_t)K2, chunkl = (uint32_t)lane - sub * (uint32_t)K2;
        const bool lane_on = sub < rpp;
        int32_t hidx[PT_MAXHP];                       // source edge of this lane's row in every halo piece of the NEXT patch to stage
        auto load_idx = [&](int p) {
            const int nOwn = cptr(m.patchEdgeStart)[p + 1] - cptr(m.patchEdgeStart)[p];
            const int rs0 = cptr(tm.rowStart)[p], Rr = cptr(tm.rowStart)[p + 1] - rs0;
#pragma unroll
            for (int j = 0; j < PT_MAXHP; ++j) {
                // halo pieces are the pieces from floor(nOwn / rpp) on (the piece that mixes own and halo rows included)
                const uint32_t q = (uint32_t)nOwn / rpp + (uint32_t)j, row = q * rpp + sub;
                hidx[j] = (lane_on && row < (uint32_t)Rr) ? tm.rowEdge[rs0 + (int)row] : -1;
            }
        };
        auto stage = [&](int p, int b) {
            lds_wr_t buf = (lds_wr_t)smem + (size_t)b * bufB;
            const int c0 = cptr(m.patchCellStart)[p], nOwnC = cptr(m.patchCellStart)[p + 1] - c0;
            const int e0 = cptr(m.patchEdgeStart)[p], nOwnE = cptr(m.patchEdgeStart)[p + 1] - e0;
            const int Rr = cptr(tm.rowStart)[p + 1] - cptr(tm.rowStart)[p];
            const uint32_t nPieces = ((uint32_t)Rr + rpp - 1) / rpp, qOwn = (uint32_t)nOwnE / rpp;
            // own rows: consecutive edges e0 .. -- no index needed
            for (uint32_t q = 0; q < qOwn; ++q) {
                const uint32_t row = q * rpp + sub;
                if (lane_on)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((glb_rd_t)a.pu + (size_t)(e0 + row) * rowB + chunkl * 16u),
                                                     (__attribute__((address_space(3))) void *)(buf + (size_t)q * 1024), 16, 0, 0);
            }
            // the remaining pieces through the prefetched indices
#pragma unroll
            for (int j = 0; j < PT_MAXHP; ++j) {
                const uint32_t q = qOwn + (uint32_t)j;
                if (q < nPieces && hidx[j] >= 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((glb_rd_t)a.pu + (size_t)hidx[j] * rowB + chunkl * 16u),
                                                     (__attribute__((address_space(3))) void *)(buf + (size_t)q * 1024), 16, 0, 0);
            }
            // records: contiguous ranges
            lds_wr_t rec = buf + (size_t)nPiecesMax * 1024;
            dma16(rec + R.eRec, tm.eRecT + (size_t)e0 * m.EI, (uint32_t)nOwnE * m.EI * 4u, lane);
            dma16(rec + R.woe, m.woe + (size_t)e0 * ME2, (uint32_t)nOwnE * ME2 * 8u, lane);
            dma16(rec + R.feoe, m.feoe + (size_t)e0 * ME2, (uint32_t)nOwnE * ME2 * 8u, lane);
            dma16(rec + R.cRec, tm.cRecT + (size_t)c0 * m.CI, (uint32_t)nOwnC * m.CI * 4u, lane);
            dma16(rec + R.sdv, m.sdv + (size_t)c0 * ME, (uint32_t)nOwnC * ME * 8u, lane);
            dma4(rec + R.g, m.gInvDc + e0, (uint32_t)nOwnE * 8u, lane);
            dma4(rec + R.invA, m.invArea + c0, (uint32_t)nOwnC * 8u, lane);
            dma4(rec + R.rsum, m.rsum + c0, (uint32_t)nOwnC * 8u, lane);
        };
        load_idx(patch_at(0));
        __builtin_amdgcn_s_waitcnt(0x0F70);
        stage(patch_at(0), 0);
        if (nSeq > 1) load_idx(patch_at(1));
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();                                                   // patch 0 has landed
        for (int i = 0; i < nSeq; ++i) {
            if (i + 1 < nSeq) stage(patch_at(i + 1), (i + 1) & 1);         // while the compute waves work on patch i
            if (i + 2 < nSeq) load_idx(patch_at(i + 2));                   // the transfers just issued have read their addresses
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();                                               // patch i + 1 landed; buffer i & 1 is free again
        }
        return;
    }

    // ================= compute waves =================
    // Per patch: EVERY global load of the group's entities is issued first (the layerThickness rows of its cells, ssh and
    // the Curr / New rows of its edges: registers are plentiful at one workgroup per CU), one wait, then the entities are
    // evaluated from registers + LDS and stored as they come: one exposed memory round trip per patch, which the loader's
    // transfers of the next patch overlap.
    const int grp = tid >> 5, l = tid & 31;
    const uint32_t voff = (uint32_t)l * 16u;
    const int k0 = 2 * l;
    const bool act = k0 < K;
    constexpr bool FE = MODE >= 4, STALE = MODE == 4;
    constexpr int NCI = PT_NCI, NEI = PT_NEI;                              // entity iterations per group and patch (host checks the fit)
    const double2 z2 = make_double2(0.0, 0.0);
    __syncthreads();                                                       // patch 0 has landed
    for (int it = 0; it < nSeq; ++it) {
        const int p = patch_at(it);
        unsigned char *buf = smem + (size_t)(it & 1) * bufB;
        unsigned char *rec = buf + (size_t)nPiecesMax * 1024;
        const double *Lwoe = reinterpret_cast<const double *>(rec + R.woe), *Lfeoe = reinterpret_cast<const double *>(rec + R.feoe);
        const double *Lg = reinterpret_cast<const double *>(rec + R.g), *Lsdv = reinterpret_cast<const double *>(rec + R.sdv);
        const double *LinvA = reinterpret_cast<const double *>(rec + R.invA), *Lrsum = reinterpret_cast<const double *>(rec + R.rsum);
        const uint32_t *LeRec = reinterpret_cast<const uint32_t *>(rec + R.eRec), *LcRec = reinterpret_cast<const uint32_t *>(rec + R.cRec);
        const int c0 = cptr(m.patchCellStart)[p], nOwnC = cptr(m.patchCellStart)[p + 1] - c0;
        const int e0 = cptr(m.patchEdgeStart)[p], nOwnE = cptr(m.patchEdgeStart)[p + 1] - e0;
        const uint32_t ldsU = (uint32_t)(size_t)(lds_bytes_t)buf + voff;

        // ---- issue: cells ----
        double2 chc[NCI], chv[NCI][ME], ccur[NCI], cnin[NCI];
        double carea[NCI];
#pragma unroll
        for (int j = 0; j < NCI; ++j) {
            const int ci = grp + PT_NG * j;
            chc[j] = z2; ccur[j] = z2; cnin[j] = z2; carea[j] = 0.0;
#pragma unroll
            for (int i = 0; i < ME; ++i) chv[j][i] = z2;
            if (ci < nOwnC && act) {
                const int c = c0 + ci;
                const uint32_t *r = LcRec + (size_t)ci * m.CI;
                const uint32_t own = (uint32_t)c * rowB + voff;
                chc[j] = gload2(a.ph, own);
#pragma unroll
                for (int i = 0; i < ME; ++i)
                    chv[j][i] = STALE ? gload2(a.hEdgeOld, cptr(m.cRec)[(size_t)c * m.CI + i] + voff) : gload2(a.ph, r[ME + i] + voff);
                if constexpr (MODE == 2) ccur[j] = gload2(a.ch, own);
                if constexpr (MODE == 2 || MODE == 3) cnin[j] = gload2(a.nh_in, own);
                if constexpr (FE) carea[j] = a.areaCell[c];
            }
        }
        // ---- issue: edges ----
        double esv[NEI];
        double2 ecur[NEI], enin[NEI], ehx[NEI], ehy[NEI], ehEo[NEI];
#pragma unroll
        for (int j = 0; j < NEI; ++j) {
            const int ei = grp + PT_NG * j;
            esv[j] = 0.0; ecur[j] = z2; enin[j] = z2; ehx[j] = z2; ehy[j] = z2; ehEo[j] = z2;
            if (ei < nOwnE) {
                const uint32_t *r = LeRec + (size_t)ei * m.EI;
                const uint32_t own = (uint32_t)(e0 + ei) * rowB + voff;
                if (l < 2) esv[j] = a.ssh[r[ME2 + l]];                     // ssh of cellsOnEdge[l]
                if (act) {
                    if constexpr (MODE == 2) ecur[j] = gload2(a.cu, own);
                    if constexpr (MODE == 2 || MODE == 3) enin[j] = gload2(a.nu_in, own);
                    if constexpr (FE) {
                        ehx[j] = gload2(a.ph, r[ME2] * rowB + voff);
                        ehy[j] = gload2(a.ph, r[ME2 + 1] * rowB + voff);
                        if constexpr (STALE) ehEo[j] = gload2(a.hEdgeOld, own);
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // the one exposed round trip of the patch

        // ---------------- cells ----------------
#pragma unroll
        for (int j = 0; j < NCI; ++j) {
            const int ci = grp + PT_NG * j;
            const bool valid = ci < nOwnC;
            const int cc = valid ? ci : 0, c = c0 + cc;
            const uint32_t *r = LcRec + (size_t)cc * m.CI;
            const double *rs = Lsdv + (size_t)cc * ME;
            const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
            const double invA = LinvA[cc];
            const uint32_t own = (uint32_t)c * rowB + voff;
            const double2 hc = chc[j];
            double2 uv[ME];
            {
                uint32_t ad[ME];
                v4u_t raw[ME];
#pragma unroll
                for (int i = 0; i < ME; ++i) ad[i] = ldsU + r[i];
                lds_burst<ME>(raw, ad);
#pragma unroll
                for (int i = 0; i < ME; ++i) uv[i] = __builtin_bit_cast(double2, raw[i]);
            }
            double2 t = z2, dv = z2;
            const bool plain = __builtin_amdgcn_ballot_w64(valid && !(mask == (1u << ME) - 1u && all)) == 0;
            auto hE = [&](int i) { return STALE ? chv[j][i] : make_double2(0.5 * (hc.x + chv[j][i].x), 0.5 * (hc.y + chv[j][i].y)); };
            if (plain) {
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
            } else {
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
            double2 hs = z2;
            const bool wr = valid && act;
            if constexpr (MODE == 0) { if (wr) gstore2o(a.tendH, own, t); }
            if constexpr (MODE == 1 || MODE == 2) {
                const double2 hcur = MODE == 2 ? ccur[j] : hc;
                const double2 nb = MODE == 2 ? cnin[j] : hcur;
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);                    // time_integration.jl:125
                if (wr) {
                    gstore2o(a.ph_out, own, hs);
                    gstore2o(a.nh_out, own, make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y));   // :135
                }
            }
            if constexpr (MODE == 3) {
                hs = make_double2(cnin[j].x + a.b * t.x, cnin[j].y + a.b * t.y);
                if (wr) gstore2o(a.nh_out, own, hs);
            }
            if constexpr (FE) {
                hs = make_double2(hc.x + a.a * t.x, hc.y + a.a * t.y);                        // time_integration.jl:199
                if (wr) {
                    gstore2o(a.ph_out, own, hs);
                    gstore2o(a.tendH, own, t);
                    gstore2o(a.div, own, make_double2(dv.x / carea[j], dv.y / carea[j]));     // Operators.jl:41
                }
            }
            if (!wr) hs = z2;
            if constexpr (MODE != 0) {
#pragma unroll
                for (int sft = 16; sft >= 1; sft >>= 1) {                   // oracle_ksum order
                    const double ox = __shfl_xor(hs.x, sft, 32), oy = __shfl_xor(hs.y, sft, 32);
                    hs = make_double2(hs.x + ox, hs.y + oy);
                }
                if (valid && l == 0) a.ssh_out[c] = (hs.x + hs.y) - Lrsum[cc];                 // :209 (+N3)
            }
        }
        // ---------------- edges ----------------
#pragma unroll
        for (int j = 0; j < NEI; ++j) {
            const int ei = grp + PT_NG * j;
            const bool valid = ei < nOwnE;
            const int ee = valid ? ei : 0;
            const uint32_t *r = LeRec + (size_t)ee * m.EI;
            const double *rw = Lwoe + (size_t)ee * ME2;
            const double *rf = Lfeoe + (size_t)ee * ME2;
            const uint32_t mask = r[ME2 + 2];
            const int mlt = (int)r[ME2 + 3];
            const double g = Lg[ee];
            const uint32_t own = (uint32_t)(e0 + ee) * rowB + voff;
            double2 uv[ME2], up = z2;
            {
                uint32_t ad[ME2];
                v4u_t raw[ME2];
#pragma unroll
                for (int i = 0; i < ME2; ++i) ad[i] = ldsU + r[i];
                lds_burst<ME2>(raw, ad);
#pragma unroll
                for (int i = 0; i < ME2; ++i) uv[i] = __builtin_bit_cast(double2, raw[i]);
                if constexpr (MODE == 1 || FE)
                    up = __builtin_bit_cast(double2, *(const __attribute__((address_space(3))) v2d_t *)((lds_bytes_t)buf +
                                                         ((uint32_t)ee / rpp) * 1024u + ((uint32_t)ee % rpp) * rowB + voff));
            }
            const double ds = __shfl(esv[j], 1, 32) - __shfl(esv[j], 0, 32);                  // ssh[c2] - ssh[c1]
            const bool plain = __builtin_amdgcn_ballot_w64(valid && !(mask == (1u << ME2) - 1u && mlt >= K)) == 0;
            const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
            double2 t = z2;
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
            if (valid && act) {
                if constexpr (MODE == 0) gstore2o(a.tendU, own, t);
                if constexpr (MODE == 1) {
                    gstore2o(a.pu_out, own, make_double2(up.x + a.a * t.x, up.y + a.a * t.y));     // time_integration.jl:124
                    gstore2o(a.nu_out, own, make_double2(up.x + a.b * t.x, up.y + a.b * t.y));     // :134
                }
                if constexpr (MODE == 2) {
                    gstore2o(a.pu_out, own, make_double2(ecur[j].x + a.a * t.x, ecur[j].y + a.a * t.y));
                    gstore2o(a.nu_out, own, make_double2(enin[j].x + a.b * t.x, enin[j].y + a.b * t.y));
                }
                if constexpr (MODE == 3) gstore2o(a.nu_out, own, make_double2(enin[j].x + a.b * t.x, enin[j].y + a.b * t.y));
                if constexpr (FE) {
                    const double2 hEn = make_double2(0.5 * (ehx[j].x + ehy[j].x), 0.5 * (ehx[j].y + ehy[j].y));   // Operators.jl:217
                    const double2 hF = STALE ? ehEo[j] : hEn;
                    gstore2o(a.pu_out, own, make_double2(up.x + a.a * t.x, up.y + a.a * t.y));     // time_integration.jl:199
                    gstore2o(a.tendU, own, t);
                    gstore2o(a.F, own, make_double2(up.x * hF.x, up.y * hF.y));                    // DiagnosticVars.jl:165
                    gstore2o(a.hEdgeNew, own, hEn);
                }
            }
        }
        __syncthreads();                                                   // next patch landed; this buffer may be refilled
    }
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
static size_t ptile2_lds_bytes(const MeshDev &md)
{
    const size_t rowB = (size_t)md.K * 8, rpp = 1024 / rowB;
    const PRec R = prec_layout(md.ME, md.ME2, md.EI, md.CI, md.maxOwnE, md.maxOwnC);
    return 2 * ((((size_t)md.maxRows + rpp - 1) / rpp) * 1024 + R.bytes);
}

// maxHaloPieces: max over the patches of (pieces of the patch) - (pieces made of own rows only), computed by the caller from
// the host plan: the loader wave keeps that many row indices per lane
int stage_ptile2_halo_piece_budget() { return PT_MAXHP; }

bool stage_ptile2_usable(const MeshDev &md)
{
    if (!(md.tileRecOk && md.K >= 2 && md.K <= 64 && !(md.K & 1) && md.maxRows >= 1 && md.maxOwnC >= 1 && md.maxOwnE >= 1)) return false;
    if (md.maxOwnC > PT_NG * PT_NCI || md.maxOwnE > PT_NG * PT_NEI) return false;
    return ptile2_lds_bytes(md) <= 160 * 1024 &&
           ((md.ME == 6 && md.ME2 == 10) || (md.ME == 8 && md.ME2 == 14) || (md.ME <= 6 && md.ME2 <= 14));
}

template <int ME, int ME2>
static hipError_t launch_ptile2(const ColMesh &m, const PTileMesh &tm, const StageArgs &a, int mode, dim3 g, size_t lds, int mE, int mC, hipStream_t s)
{
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipSuccess;
#define RAISE(MODE) if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_stage_ptile2<ME, ME2, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        RAISE(0) RAISE(1) RAISE(2) RAISE(3) RAISE(4) RAISE(5)
#undef RAISE
        if (e != hipSuccess) return e;
        raised = true;
    }
    const dim3 b(PT_NT);
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_ptile2<ME, ME2, 0>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 1: hipLaunchKernelGGL((k_stage_ptile2<ME, ME2, 1>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 2: hipLaunchKernelGGL((k_stage_ptile2<ME, ME2, 2>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 3: hipLaunchKernelGGL((k_stage_ptile2<ME, ME2, 3>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 4: hipLaunchKernelGGL((k_stage_ptile2<ME, ME2, 4>), g, b, lds, s, m, tm, a, mE, mC); break;
        case 5: hipLaunchKernelGGL((k_stage_ptile2<ME, ME2, 5>), g, b, lds, s, m, tm, a, mE, mC); break;
        default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

hipError_t launch_stage_ptile2(const MeshDev &md, const StageArgs &a, int nCUs, hipStream_t s)
{
    const int mode = colp_mode(a);
    if (mode < 0 || !stage_ptile2_usable(md) || md.tailPatch >= 0) return hipErrorNotSupported;
    const int wgs = std::min(8 * ((nCUs + 7) / 8), 8 * ((md.nPatches + 7) / 8));
    const dim3 g(wgs);
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc, 0};
    const PTileMesh tm{md.rowStart, md.rowEdge, md.eRecT, md.cRecT, md.maxRows};
    const size_t lds = ptile2_lds_bytes(md);
    if (md.ME == 6 && md.ME2 == 10) return launch_ptile2<6, 10>(m, tm, a, mode, g, lds, md.maxOwnE, md.maxOwnC, s);
    if (md.ME == 8 && md.ME2 == 14) return launch_ptile2<8, 14>(m, tm, a, mode, g, lds, md.maxOwnE, md.maxOwnC, s);
    if (md.ME <= 6 && md.ME2 <= 14) return launch_ptile2<6, 14>(m, tm, a, mode, g, lds, md.maxOwnE, md.maxOwnC, s);
    return hipErrorNotSupported;
}

}  // namespace moka
