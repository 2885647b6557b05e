// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of libmoka_hip: the DEFAULT fused tendency / RK-stage
// kernels (k_stage_rec2c for Float64 states, k_stage_rec2c_f32 for fp32-storage states), the one-launch Forward-Euler
// step (k_fe), the stand-alone operators and the utility kernels (ssh, permutations, halo maps).
// The other execution shapes of the stage kernel live in stage_variants.hip, the optional nonlinear terms in
// nonlinear.hip, reverse mode in adjoint.hip; shared device helpers in kernels_common.hpp.
//
// Layout.  Fields are (nVertLevels, n) with the level index fastest, exactly the reference layout
// (PrognosticVars.jl:11-17), so every neighbour gather is one contiguous row read.  One workgroup (256 threads)
// walks one *patch*: P consecutive cells of the RCB ordering plus the edges those cells own; the blockIdx -> patch map
// keeps consecutive patches on one XCD (blocks b, b+8, ... share an XCD's L2).  The generic kernels (k_fe, operators)
// give a group of LPC lanes (smallest power of two >= nVertLevels, capped at 64) one entity at a time, lane l owning
// levels l, l+LPC, ...; the default stage kernels give a lane 2 (fp64) or 4 (fp32) consecutive levels = one 16-byte
// load, and a wave two or more entities.
//
// Arithmetic.  Compiled with -ffp-contract=off and written in the reference's operand order
// (each expression cites the reference line), so results are bit-identical to the CPU oracle.
// Memory-bound indirect stencil: no MFMA by design.
#include "kernels_common.hpp"

namespace moka {

// ------------------------------------------------------------------------------------------------
// rec2 + own-edge cache ("rec2c"): k_stage_rec2 with the u-rows of the patch's OWN edges copied once into LDS
// (a contiguous range: one coalesced copy, no halo list).  ~65 % of all u gathers of a compact patch refer to
// its own edges; those become ds_read_b128 and their texture-address transactions disappear.  A gather whose
// row is not cached is an exec-masked global load (the two half-waves of a wave decide independently).
// Not pipelined: loads sit behind per-lane branches, so the compiler waits with vmcnt(0) at first use; 12+
// waves per CU cover the latency instead.
// ------------------------------------------------------------------------------------------------
// (MODE 7 / 8 / 9, the 13-stream RK4 form: rk13_combine in kernels_common.hpp -- the nonlinear stage kernel uses the same expression)

template <int ME, int ME2, int MODE, int NT = BLOCK>
__global__ __launch_bounds__(NT) void k_stage_rec2c(const ColMesh m, const StageArgs a, int maxOwnE, int maxOwnC)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    // the unit a workgroup walks: one patch, or (m.pairEnd > 0) two consecutive patches as one -- their cell and edge ranges are
    // consecutive too, so everything below sees one larger patch whose row cache covers both
    const int p = m.pairEnd ? m.patchBegin + 2 * pl_ : (m.tailPlus1 && pl_ == m.nPatches - 1) ? m.tailPlus1 - 1 : pl_ + m.patchBegin;
    const int pEnd = m.pairEnd ? (p + 2 < m.pairEnd ? p + 2 : m.pairEnd) : p + 1;
    constexpr int NG = NT / 32;
    const int tid = threadIdx.x;
    const int grp = tid >> 5, l = tid & 31;
    const int K = m.K, K2 = K >> 1;
    const uint32_t voff = (uint32_t)l * 16u, rowB = (uint32_t)K * 8u;
    const RecLds L = rec_carve(smem, m, ME, ME2, maxOwnE, maxOwnC);
    // the row cache sits behind the records
    const size_t recBytes = ((size_t)maxOwnE * (2 * ME2 + 1) * 8 + (size_t)maxOwnC * (ME + 2) * 8 +
                             ((size_t)maxOwnE * m.EI + (size_t)maxOwnC * m.CI) * 4 + 15) & ~(size_t)15;
    double2 *ubuf2 = reinterpret_cast<double2 *>(smem + recBytes);
    const unsigned char *ubytes = reinterpret_cast<const unsigned char *>(ubuf2) + voff;
    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[pEnd];
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[pEnd];
    const int nOwnC = c1 - c0, nOwnE = e1 - e0;
    const uint32_t e0B = (uint32_t)e0 * rowB, nOwnB = (uint32_t)nOwnE * rowB;
    // Forward-Euler modes with the vertex pass in the same launch (a.vort): the patch's vertex records sit behind the row cache
    double *Lvw = reinterpret_cast<double *>(smem + recBytes + (size_t)maxOwnE * rowB);    // [maxOwnV][3] coefficients
    uint32_t *Lvo = reinterpret_cast<uint32_t *>(Lvw + (size_t)m.maxOwnV * 3);              // [maxOwnV][4] u-row byte offsets
    // MODE 10 / 11: the LEAN forms of modes 5 / 6 (a lean Forward-Euler step stores the new level and relativeVorticity only): the
    // same loads and sums with every optional DiagnosticVars / TendencyVars output compiled out instead of tested at run time --
    // no velocityDivCell sum, no layerThickness rows in the edge loop, fewer live registers
    constexpr bool LEAN = MODE >= 10;
    constexpr int BASE = LEAN ? MODE - 5 : MODE;
    constexpr bool FE = BASE >= 4 && BASE <= 6, STALE = BASE == 4, PREV = BASE == 6;
    int v0 = 0, nOwnV = 0;
    if constexpr (FE) {
        if (a.vort) {
            v0 = cptr(m.patchVertStart)[p];
            nOwnV = cptr(m.patchVertStart)[p + 1] - v0;
        }
    }

    {   // Staging: every global load of the patch's records and own u rows is issued before the first LDS write, so the workgroup
        // pays one memory latency here instead of one per array.  The unrolled part covers the default patch (P = 16: 48-51
        // own edges at K <= 64); larger patches finish in the plain loops at the end.
        const double2 *src = reinterpret_cast<const double2 *>(a.pu) + (size_t)e0 * K2;
        const int nU = nOwnE * K2, nER = nOwnE * m.EI, nW = nOwnE * ME2, nCR = nOwnC * m.CI, nS = nOwnC * ME;
        constexpr int UU = 6, UE = 3, UW = 2;
        double2 vU[UU];
        uint32_t vE[UE], vC;
        double vW[UW], vF[UW], vG, vS, vA, vR;
#pragma unroll
        for (int j = 0; j < UU; ++j) vU[j] = (tid + j * NT < nU) ? src[tid + j * NT] : make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < UE; ++j) vE[j] = (tid + j * NT < nER) ? m.eRec[(size_t)e0 * m.EI + tid + j * NT] : 0u;
#pragma unroll
        for (int j = 0; j < UW; ++j) {
            vW[j] = (tid + j * NT < nW) ? m.woe[(size_t)e0 * ME2 + tid + j * NT] : 0.0;
            vF[j] = (tid + j * NT < nW) ? m.feoe[(size_t)e0 * ME2 + tid + j * NT] : 0.0;
        }
        vG = tid < nOwnE ? m.gInvDc[e0 + tid] : 0.0;
        vC = tid < nCR ? m.cRec[(size_t)c0 * m.CI + tid] : 0u;
        vS = tid < nS ? m.sdv[(size_t)c0 * ME + tid] : 0.0;
        vA = tid < nOwnC ? m.invArea[c0 + tid] : 0.0;
        vR = tid < nOwnC ? m.rsum[c0 + tid] : 0.0;
        uint32_t vVo = 0u;
        double vVw = 0.0;
        if constexpr (FE) {
            if (tid < nOwnV * 4) vVo = m.vRec[(size_t)v0 * 4 + tid];
            if (tid < nOwnV * 3) vVw = m.cv[(size_t)v0 * 3 + tid];
        }
#pragma unroll
        for (int j = 0; j < UU; ++j) if (tid + j * NT < nU) ubuf2[tid + j * NT] = vU[j];
#pragma unroll
        for (int j = 0; j < UE; ++j) if (tid + j * NT < nER) L.eRec[tid + j * NT] = vE[j];
#pragma unroll
        for (int j = 0; j < UW; ++j)
            if (tid + j * NT < nW) {
                L.woe[tid + j * NT] = vW[j];
                L.feoe[tid + j * NT] = vF[j];
            }
        if (tid < nOwnE) L.g[tid] = vG;
        if (tid < nCR) L.cRec[tid] = vC;
        if (tid < nS) L.sdv[tid] = vS;
        if (tid < nOwnC) {
            L.invA[tid] = vA;
            L.rsum[tid] = vR;
        }
        if constexpr (FE) {
            if (tid < nOwnV * 4) Lvo[tid] = vVo;
            if (tid < nOwnV * 3) Lvw[tid] = vVw;
            for (int i = tid + NT; i < nOwnV * 4; i += NT) Lvo[i] = m.vRec[(size_t)v0 * 4 + i];
            for (int i = tid + NT; i < nOwnV * 3; i += NT) Lvw[i] = m.cv[(size_t)v0 * 3 + i];
        }
        for (int i = tid + UU * NT; i < nU; i += NT) ubuf2[i] = src[i];
        for (int i = tid + UE * NT; i < nER; i += NT) L.eRec[i] = m.eRec[(size_t)e0 * m.EI + i];
        for (int i = tid + UW * NT; i < nW; i += NT) {
            L.woe[i] = m.woe[(size_t)e0 * ME2 + i];
            L.feoe[i] = m.feoe[(size_t)e0 * ME2 + i];
        }
        for (int i = tid + NT; i < nOwnE; i += NT) L.g[i] = m.gInvDc[e0 + i];
        for (int i = tid + NT; i < nCR; i += NT) L.cRec[i] = m.cRec[(size_t)c0 * m.CI + i];
        for (int i = tid + NT; i < nS; i += NT) L.sdv[i] = m.sdv[(size_t)c0 * ME + i];
        for (int i = tid + NT; i < nOwnC; i += NT) {
            L.invA[i] = m.invArea[c0 + i];
            L.rsum[i] = m.rsum[c0 + i];
        }
    }
    __syncthreads();

    const int k0 = 2 * l;
    const bool act = k0 < K;
    // Explicit address spaces: with generic pointers the compiler folds the two branches into ONE flat_load from a
    // selected pointer, and a flat load of an LDS address still goes through the texture-address unit.
    const lds_bytes_t ubytesL = (lds_bytes_t)ubytes;
    const glb_bytes_t puG = (glb_bytes_t)a.pu;
    // u rows in two phases so that nothing serialises: every lane reads the cache (lds_burst; a lane whose row is not
    // cached reads row 0), then only those lanes overwrite the value with an exec-masked global_load_dwordx4.
    const uint32_t ldsU = (uint32_t)(size_t)ubytesL;
    auto urow_addr = [&](uint32_t off, bool &cached) -> uint32_t {
        const uint32_t loc = off - e0B;
        cached = loc < nOwnB;
        return ldsU + (cached ? loc : 0u);
    };

    // Stores are deferred by one iteration: an entity's results are written during the NEXT iteration, right after that
    // iteration's loads have arrived.  Issued at the end of their own iteration they force a vmcnt(0) at the loop top
    // (their data registers are reused at once), i.e. every iteration waited for its stores to be acknowledged before
    // the next gathers could start; now the acknowledgement overlaps the arithmetic and the next record reads.
    // MODE 4 / 5 / 6: one Forward-Euler step (time_integration.jl:150-193) with the diagnostics of diagnostic_compute!
    // (DiagnosticVars.jl:108-117) except relativeVorticity; 4 = thickness flux from the previous step's
    // layerThicknessEdge (MOKA_FE_STALE_HEDGE, the reference's behaviour), 5 = from this step's; 6 = the reference's
    // behaviour again, with that stale layerThicknessEdge FORMED in the cell loop from the previous time level's
    // layerThickness rows (a.hPrev) -- what the previous step interpolated it from, so the same bits (0.5 * (x + y) commutes)
    // -- instead of six gathered rows of the stored array per cell: those were 9.35 GB of fetches for 4.2 GB of inputs
    // (profiles/r02_variants.txt).  The host selects 6 only when the stored array IS that interpolation (moka_state.hEdgePrev).
    // MODE 7 / 8 / 9: the RK4 step with 13 instead of 16 state streams (moka_set_tuning key 7; rk13_combine above): stage 1
    // stores Provis' only (7), stages 2 and 3 read Provis and Curr and store Provis' only (8), stage 4 forms New from the own
    // rows of Curr and of the three provisional states and the last tendency (9) -- no New accumulator travels through stages 1-3.
    double2 pA = make_double2(0.0, 0.0), pB = pA, pD = pA, pE = pA;
    double pS = 0.0;
    // `pend` is false only in a group's first iteration; the loops carry `#pragma nounroll` so that the compiler does not peel
    // a copy of the whole loop body for it (a third more code, more registers)
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
            if constexpr (MODE == 3 || MODE == 9) gstore2o(a.nh_out, pOff, pB);
            if constexpr (MODE == 7 || MODE == 8) gstore2o(a.ph_out, pOff, pA);
            if constexpr (LEAN) gstore2o(a.ph_out, pOff, pA);
            if constexpr (FE && !LEAN) {                 // every output group of a Forward-Euler launch is optional (wave-uniform):
                if (a.ph_out) gstore2o(a.ph_out, pOff, pA);   // a lean step stores the new level only (modes 10 / 11), the launch that
                if (a.tendH) gstore2o(a.tendH, pOff, pB);     // materialises the step's DiagnosticVars / TendencyVars on demand stores
                if (a.div) gstore2o(a.div, pOff, pD);         // only those
            }
        }
        if constexpr (MODE != 0)
            if (l == 0 && (!FE || LEAN || a.ssh_out)) a.ssh_out[pC] = pS;
    };
    // ---------------- cells ----------------
#pragma nounroll
    for (int ci = grp; ci < nOwnC; ci += NG) {
        const int c = c0 + ci;
        const uint32_t *r = L.cRec + (size_t)ci * m.CI;
        const double *rs = L.sdv + (size_t)ci * ME;
        const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
        const double invA = L.invA[ci];
        const uint32_t own = (uint32_t)c * rowB + voff;
        double2 hc = make_double2(0.0, 0.0), uv[ME], hv[ME], cur = hc, nin = hc, hpc = hc, q3 = hc;
        if (act) {
            bool cached[ME];
            uint32_t ad[ME], goff[ME];
            v4u_t raw[ME];
            hc = gload2(a.ph, own);
            if constexpr (PREV) hpc = gload2(a.hPrev, own);
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                const uint32_t off = r[i];
                hv[i] = STALE ? gload2(a.hEdgeOld, off + voff) : gload2(PREV ? a.hPrev : a.ph, r[ME + i] + voff);
                ad[i] = urow_addr(off, cached[i]);
                goff[i] = off + voff;
                asm("" : "+v"(goff[i]));               // stays in a VGPR (see the edge loop)
            }
            lds_burst<ME>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                uv[i] = __builtin_bit_cast(double2, raw[i]);
                if (!cached[i]) uv[i] = glb_row2(puG + goff[i]);
            }
            if constexpr (MODE == 2 || MODE == 8 || MODE == 9) cur = gload2(a.ch, own);
            if constexpr (MODE == 2 || MODE == 3 || MODE == 9) nin = gload2(a.nh_in, own);     // (9: the own row of P2)
            if constexpr (MODE == 9) q3 = gload2(a.q3h, own);                                   // ... and of P3
        }
        double area = 1.0;
        if constexpr (FE && !LEAN) if (a.div) area = a.areaCell[c];
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): this iteration's loads (needed next anyway) ...
        if (pend) flush_cell();                                        // ... so that the stores queue up behind them, not ahead
        double2 t = make_double2(0.0, 0.0);
        // regular entity (every slot valid, every level active) in BOTH half-waves: no per-slot masks (wave-uniform branch)
        const bool plain = __builtin_amdgcn_ballot_w64(!(mask == (1u << ME) - 1u && all)) == 0;
        double2 dv = make_double2(0.0, 0.0);                            // velocityDivCell (FE): Operators.jl:18,39
        // thickness at the edge: interpolated (Operators.jl:217) or, MODE 4, what the previous step stored
        const double2 hself = PREV ? hpc : hc;
        auto hE = [&](int i) { return STALE ? hv[i] : make_double2(0.5 * (hself.x + hv[i].x), 0.5 * (hself.y + hv[i].y)); };
        if (plain) {
            if (act) {
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    const double2 he = hE(i);
                    t.x += uv[i].x * he.x * rs[i] * invA;
                    t.y += uv[i].y * he.y * rs[i] * invA;
                    if constexpr (FE && !LEAN) {
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
                if constexpr (FE && !LEAN) {
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
            if constexpr (MODE == 7 || MODE == 8) {
                const double2 hcur = MODE == 8 ? cur : hc;
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);                    // time_integration.jl:125
                pA = hs;
            }
            if constexpr (MODE == 9) {
                hs = make_double2(rk13_combine(cur.x, nin.x, q3.x, hc.x, a.b, t.x), rk13_combine(cur.y, nin.y, q3.y, hc.y, a.b, t.y));
                pB = hs;
            }
            if constexpr (FE) {
                hs = make_double2(hc.x + a.a * t.x, hc.y + a.a * t.y);                        // time_integration.jl:199
                pA = hs;
                if constexpr (!LEAN) {
                    pB = t;
                    if (a.div) pD = make_double2(dv.x / area, dv.y / area);                   // Operators.jl:41
                }
            }
        }
        if constexpr (MODE != 0) {
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) {                   // oracle_ksum order (see k_stage_rec2)
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
            if constexpr (MODE == 3 || MODE == 9) gstore2o(a.nu_out, pOff, pB);
            if constexpr (MODE == 7 || MODE == 8) gstore2o(a.pu_out, pOff, pA);
            if constexpr (LEAN) gstore2o(a.pu_out, pOff, pA);
            if constexpr (FE && !LEAN) {
                if (a.pu_out) gstore2o(a.pu_out, pOff, pA);
                if (a.tendU) gstore2o(a.tendU, pOff, pB);
                if (a.F) gstore2o(a.F, pOff, pD);
                if (a.hEdgeNew) gstore2o(a.hEdgeNew, pOff, pE);
            }
        }
    };

    // ---------------- edges ----------------
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
        double2 uv[ME2], cur = make_double2(0.0, 0.0), nin = cur, hx = cur, hy = cur, hEo = cur, q3 = cur;
        if (act) {
            bool cached[ME2];
            uint32_t ad[ME2], goff[ME2];
            v4u_t raw[ME2];
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                const uint32_t off = r[i];
                ad[i] = urow_addr(off, cached[i]);
                goff[i] = off + voff;
                asm("" : "+v"(goff[i]));               // keep it in a VGPR: otherwise each masked load re-reads r[i] from LDS first
            }
            lds_burst<ME2>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                uv[i] = __builtin_bit_cast(double2, raw[i]);
                if (!cached[i]) uv[i] = glb_row2(puG + goff[i]);
            }
            if constexpr (MODE == 2 || MODE == 8 || MODE == 9) cur = gload2(a.cu, own);
            if constexpr (MODE == 2 || MODE == 3 || MODE == 9) nin = gload2(a.nu_in, own);
            if constexpr (MODE == 9) q3 = gload2(a.q3u, own);
            if constexpr (FE && !LEAN) {
                // the edge's own diagnostics: loaded for only when they are stored (a lean step stores neither)
                if (a.hEdgeNew || (MODE == 5 && a.F)) {
                    hx = gload2(a.ph, r[ME2] * rowB + voff);           // layerThickness of cellsOnEdge[1], [2]
                    hy = gload2(a.ph, r[ME2 + 1] * rowB + voff);
                }
                if (a.F) {
                    if constexpr (STALE) hEo = gload2(a.hEdgeOld, own);
                    if constexpr (PREV) {                              // the same two cells one level back: rows the cell loop has
                        const double2 px = gload2(a.hPrev, r[ME2] * rowB + voff), py = gload2(a.hPrev, r[ME2 + 1] * rowB + voff);   // just fetched
                        hEo = make_double2(0.5 * (px.x + py.x), 0.5 * (px.y + py.y));      // what the previous step stored (Operators.jl:217)
                    }
                }
            }
        }
        if (l < 2) sv = a.ssh[r[ME2 + l]];                             // ssh of cellsOnEdge[l]: after the gathers in the queue
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): this iteration's loads (needed next anyway) ...
        if (pend) flush_edge();                                        // ... so that the stores queue up behind them, not ahead
        const double ds = __shfl(sv, 1, 32) - __shfl(sv, 0, 32);       // ssh[c2] - ssh[c1]
        const bool plain = __builtin_amdgcn_ballot_w64(!(mask == (1u << ME2) - 1u && mlt >= K)) == 0;   // wave-uniform
        if (act) {
            const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
            double2 t = make_double2(0.0, 0.0);
            if (plain) {                                               // all 2*ME2/2 slots valid, all levels active
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
                const double2 up = ubuf2[(size_t)ei * K2 + l];          // own row is in the cache
                pA = make_double2(up.x + a.a * t.x, up.y + a.a * t.y);  // time_integration.jl:124
                pB = make_double2(up.x + a.b * t.x, up.y + a.b * t.y);  // :134
            }
            if constexpr (MODE == 2) {
                pA = make_double2(cur.x + a.a * t.x, cur.y + a.a * t.y);
                pB = make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y);
            }
            if constexpr (MODE == 3) pB = make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y);
            if constexpr (MODE == 7) {
                const double2 up = ubuf2[(size_t)ei * K2 + l];          // own row is in the cache
                pA = make_double2(up.x + a.a * t.x, up.y + a.a * t.y);  // time_integration.jl:124
            }
            if constexpr (MODE == 8) pA = make_double2(cur.x + a.a * t.x, cur.y + a.a * t.y);
            if constexpr (MODE == 9) {
                const double2 up = ubuf2[(size_t)ei * K2 + l];          // the own row of P4
                pB = make_double2(rk13_combine(cur.x, nin.x, q3.x, up.x, a.b, t.x), rk13_combine(cur.y, nin.y, q3.y, up.y, a.b, t.y));
            }
            if constexpr (LEAN) {
                const double2 up = ubuf2[(size_t)ei * K2 + l];          // own row is in the cache
                pA = make_double2(up.x + a.a * t.x, up.y + a.a * t.y);                    // time_integration.jl:199
            }
            if constexpr (FE && !LEAN) {
                const double2 up = ubuf2[(size_t)ei * K2 + l];          // own row is in the cache
                pE = make_double2(0.5 * (hx.x + hy.x), 0.5 * (hx.y + hy.y));              // layerThicknessEdge, Operators.jl:217
                const double2 hF = (STALE || PREV) ? hEo : pE;
                pD = make_double2(up.x * hF.x, up.y * hF.y);                              // thicknessFlux, DiagnosticVars.jl:165
                pA = make_double2(up.x + a.a * t.x, up.y + a.a * t.y);                    // time_integration.jl:199
                pB = t;
            }
        }
        pOff = own;
        pend = true;
    }
    if (pend) flush_edge();

    // ---------------- vertices: relativeVorticity of the OLD state (CurlOnVertex, Operators.jl:137-146) ----------------
    // the summation of k_curl3 (edgesOnVertex order, on top of the stored value when it accumulates); normalVelocity rows of the
    // patch's own edges come from the LDS row cache
    if constexpr (FE) {
#pragma nounroll
        for (int vi = grp; vi < nOwnV; vi += NG) {
            if (!act) continue;
            const uint32_t *rv = Lvo + (size_t)vi * 4;
            const double *wv = Lvw + (size_t)vi * 3;
            const uint32_t own = (uint32_t)(v0 + vi) * rowB + voff;
            bool cached[3];
            uint32_t ad[3], goff[3];
            v4u_t raw[3];
            double2 uv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint32_t off = rv[j];
                ad[j] = urow_addr(off, cached[j]);
                goff[j] = off + voff;
                asm("" : "+v"(goff[j]));
            }
            double2 c = a.accumVort ? gload2(a.vort, own) : make_double2(0.0, 0.0);
            lds_burst<3>(raw, ad);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                uv[j] = __builtin_bit_cast(double2, raw[j]);
                if (!cached[j]) uv[j] = glb_row2(puG + goff[j]);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                c.x += wv[j] * uv[j].x;
                c.y += wv[j] * uv[j].y;
            }
            gstore2(a.vort, own, c);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32-state form of rec2c (BASELINE config 5: "fp32 state with fp64 tendency accumulation").
// ssh / normalVelocity / layerThickness of every time level and RK provisional state are stored as fp32
// (rows of K*4 bytes); a lane owns FOUR consecutive levels (one 16-byte load), K/4 lanes one entity and a wave
// 64/(K/4) entities (K <= 128, K % 4 == 0).  Every load widens to fp64, the arithmetic is that of k_stage_rec2c in the
// same order, stores round to nearest fp32 -- the tendencies of MODE 0 too (accumulated in fp64, stored fp32 like the
// state: inside an RK step they never leave the registers).  The StageArgs pointers of state and tendency arrays are float
// arrays in disguise (the host keeps one argument block for both storage types).
// The byte-offset records of such a mesh are built for K*4-byte rows (moka_mesh_desc.stateBytes = 4).
// ssh column sum: oracle_ksum order -- lanes l and l^16 hold levels k and k^64, then k^32 ... k^4, and the four
// levels of a lane combine as (x+z)+(y+w), i.e. k^2 then k^1.
// ------------------------------------------------------------------------------------------------
struct d4 {
    double x, y, z, w;
};
__device__ __forceinline__ d4 widen4(float4 v) { return d4{(double)v.x, (double)v.y, (double)v.z, (double)v.w}; }
__device__ __forceinline__ float4 narrow4(d4 v) { return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
__device__ __forceinline__ float4 gload4f(const double *base, uint32_t off)
{
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + off);
}
__device__ __forceinline__ d4 gload4(const double *base, uint32_t off)
{
    return widen4(*reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + off));
}
__device__ __forceinline__ void gstore4(double *base, uint32_t off, d4 v)
{
    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(base) + off) = narrow4(v);
}
__device__ __forceinline__ d4 axpy4(d4 x, double a, d4 t)      // x + a*t, the reference's operand order
{
    return d4{x.x + a * t.x, x.y + a * t.y, x.z + a * t.z, x.w + a * t.w};
}
__device__ __forceinline__ d4 round4(d4 v) { return widen4(narrow4(v)); }

// NT threads per patch, WPE waves per SIMD the register allocation is bounded for: (256, 3) is the whole-mesh default (three
// workgroups per CU at P = 24); (512, 4) gives the modes that fit 128 registers two 8-wave workgroups per CU = 4 waves per SIMD.
template <int ME, int ME2, int MODE, int NT = BLOCK, int WPE = 3>
__global__ __launch_bounds__(NT, WPE) void k_stage_rec2c_f32(const ColMesh m, const StageArgs a, int maxOwnE, int maxOwnC)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = (m.tailPlus1 && pl_ == m.nPatches - 1) ? m.tailPlus1 - 1 : pl_ + m.patchBegin;
    // K/4 lanes carry one entity, so a wave carries 64 / (K/4) of them (3 at K = 80, 4 at K = 60 or 64, 2 at K = 128):
    // lanes beyond the last whole group idle.  Shuffles address lanes of the own group only.
    const int tid = threadIdx.x;
    const int K = m.K, K4 = K >> 2;
    const int EPW = 64 / K4, NG = (NT / 64) * EPW;
    const int lane = tid & 63, sub = lane / K4;
    const int l = lane - sub * K4, gbase = sub * K4;
    const bool lane_on = sub < EPW;
    const int grp = lane_on ? (tid >> 6) * EPW + sub : (1 << 28);          // idle lanes never enter the entity loops
    auto gxor = [&](double v, int sft) -> double {                          // v of lane l^sft of the group, 0 beyond it
        const int pl = l ^ sft;
        const double o = __shfl(v, gbase + (pl < K4 ? pl : l), 64);
        return pl < K4 ? o : 0.0;
    };
    const uint32_t voff = (uint32_t)l * 16u, rowB = (uint32_t)K * 4u;      // fp32 rows
    const RecLds L = rec_carve(smem, m, ME, ME2, maxOwnE, maxOwnC);
    const size_t recBytes = ((size_t)maxOwnE * (2 * ME2 + 1) * 8 + (size_t)maxOwnC * (ME + 2) * 8 +
                             ((size_t)maxOwnE * m.EI + (size_t)maxOwnC * m.CI) * 4 + 15) & ~(size_t)15;
    float4 *ubuf4 = reinterpret_cast<float4 *>(smem + recBytes);
    const unsigned char *ubytes = reinterpret_cast<const unsigned char *>(ubuf4) + voff;
    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
    const int nOwnC = c1 - c0, nOwnE = e1 - e0;
    const uint32_t e0B = (uint32_t)e0 * rowB, nOwnB = (uint32_t)nOwnE * rowB;
    const float *sshf = reinterpret_cast<const float *>(a.ssh);
    // Forward-Euler modes with the vertex pass in the same launch (a.vort): vertex records behind the row cache (see k_stage_rec2c)
    double *Lvw = reinterpret_cast<double *>(smem + recBytes + (((size_t)maxOwnE * rowB + 15) & ~(size_t)15));
    uint32_t *Lvo = reinterpret_cast<uint32_t *>(Lvw + (size_t)m.maxOwnV * 3);
    // MODE 10 / 11: the LEAN forms of modes 5 / 6 (a lean Forward-Euler step: new level and relativeVorticity only) -- the same
    // loads and sums with every optional DiagnosticVars / TendencyVars output compiled out instead of tested at run time: fewer
    // live registers (no spills at three waves per SIMD), no velocityDivCell sum, no layerThickness rows in the edge loop
    constexpr bool LEAN = MODE >= 10;
    constexpr int BASE = LEAN ? MODE - 5 : MODE;
    int v0 = 0, nOwnV = 0;
    if constexpr (BASE >= 4) {
        if (a.vort) {
            v0 = cptr(m.patchVertStart)[p];
            nOwnV = cptr(m.patchVertStart)[p + 1] - v0;
        }
    }

    {   // staging in one phase (see k_stage_rec2c): every global load before the first LDS write; the unrolled part covers the
        // default patch (P = 24: 72-75 own edges at K = 80), the plain loops at the end whatever is larger
        const float4 *src = reinterpret_cast<const float4 *>(a.pu) + (size_t)e0 * K4;
        const int nU = nOwnE * K4, nER = nOwnE * m.EI, nW = nOwnE * ME2, nCR = nOwnC * m.CI, nS = nOwnC * ME;
        constexpr int UU = 6, UE = 4, UW = 3, UC = 2;
        float4 vU[UU];
        uint32_t vE[UE], vC[UC];
        double vW[UW], vF[UW], vG, vS, vA, vR;
#pragma unroll
        for (int j = 0; j < UU; ++j) vU[j] = (tid + j * NT < nU) ? src[tid + j * NT] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < UE; ++j) vE[j] = (tid + j * NT < nER) ? m.eRec[(size_t)e0 * m.EI + tid + j * NT] : 0u;
#pragma unroll
        for (int j = 0; j < UW; ++j) {
            vW[j] = (tid + j * NT < nW) ? m.woe[(size_t)e0 * ME2 + tid + j * NT] : 0.0;
            vF[j] = (tid + j * NT < nW) ? m.feoe[(size_t)e0 * ME2 + tid + j * NT] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < UC; ++j) vC[j] = (tid + j * NT < nCR) ? m.cRec[(size_t)c0 * m.CI + tid + j * NT] : 0u;
        vG = tid < nOwnE ? m.gInvDc[e0 + tid] : 0.0;
        vS = tid < nS ? m.sdv[(size_t)c0 * ME + tid] : 0.0;
        vA = tid < nOwnC ? m.invArea[c0 + tid] : 0.0;
        vR = tid < nOwnC ? m.rsum[c0 + tid] : 0.0;
#pragma unroll
        for (int j = 0; j < UU; ++j) if (tid + j * NT < nU) ubuf4[tid + j * NT] = vU[j];
#pragma unroll
        for (int j = 0; j < UE; ++j) if (tid + j * NT < nER) L.eRec[tid + j * NT] = vE[j];
#pragma unroll
        for (int j = 0; j < UW; ++j)
            if (tid + j * NT < nW) {
                L.woe[tid + j * NT] = vW[j];
                L.feoe[tid + j * NT] = vF[j];
            }
#pragma unroll
        for (int j = 0; j < UC; ++j) if (tid + j * NT < nCR) L.cRec[tid + j * NT] = vC[j];
        if (tid < nOwnE) L.g[tid] = vG;
        if (tid < nS) L.sdv[tid] = vS;
        if (tid < nOwnC) {
            L.invA[tid] = vA;
            L.rsum[tid] = vR;
        }
        if constexpr (BASE >= 4) {
            for (int i = tid; i < nOwnV * 4; i += NT) Lvo[i] = m.vRec[(size_t)v0 * 4 + i];
            for (int i = tid; i < nOwnV * 3; i += NT) Lvw[i] = m.cv[(size_t)v0 * 3 + i];
        }
        for (int i = tid + UU * NT; i < nU; i += NT) ubuf4[i] = src[i];
        for (int i = tid + UE * NT; i < nER; i += NT) L.eRec[i] = m.eRec[(size_t)e0 * m.EI + i];
        for (int i = tid + UW * NT; i < nW; i += NT) {
            L.woe[i] = m.woe[(size_t)e0 * ME2 + i];
            L.feoe[i] = m.feoe[(size_t)e0 * ME2 + i];
        }
        for (int i = tid + UC * NT; i < nCR; i += NT) L.cRec[i] = m.cRec[(size_t)c0 * m.CI + i];
        for (int i = tid + NT; i < nOwnE; i += NT) L.g[i] = m.gInvDc[e0 + i];
        for (int i = tid + NT; i < nS; i += NT) L.sdv[i] = m.sdv[(size_t)c0 * ME + i];
        for (int i = tid + NT; i < nOwnC; i += NT) {
            L.invA[i] = m.invArea[c0 + i];
            L.rsum[i] = m.rsum[c0 + i];
        }
    }
    __syncthreads();

    const int k0 = 4 * l;
    const bool act = lane_on;
    const lds_bytes_t ubytesL = (lds_bytes_t)ubytes;                    // explicit address spaces: see k_stage_rec2c
    const glb_bytes_t puG = (glb_bytes_t)a.pu;
    const uint32_t ldsU = (uint32_t)(size_t)ubytesL;                    // two-phase gather: see k_stage_rec2c
    auto urow_addr = [&](uint32_t off, bool &cached) -> uint32_t {
        const uint32_t loc = off - e0B;
        cached = loc < nOwnB;
        return ldsU + (cached ? loc : 0u);
    };
    const d4 zero{0.0, 0.0, 0.0, 0.0};
    // MODE 4 / 5 / 6: one Forward-Euler step with the diagnostics of diagnostic_compute! except relativeVorticity, as in
    // k_stage_rec2c; layerThicknessEdge, thicknessFlux, velocityDivCell and the tendencies are float arrays like the state.
    // (6: the stored layerThicknessEdge is the fp32-rounded interpolation; formed again from the previous level it has to be
    // rounded the same way before it is used -- round4.)
    constexpr bool FE = BASE >= 4, STALE = BASE == 4, PREV = BASE == 6;

    // ---------------- cells ----------------
    for (int ci = grp; ci < nOwnC; ci += NG) {
        const int c = c0 + ci;
        const uint32_t *r = L.cRec + (size_t)ci * m.CI;
        const double *rs = L.sdv + (size_t)ci * ME;
        const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
        const double invA = L.invA[ci];
        const uint32_t own = (uint32_t)c * rowB + voff;
        // the cell loop keeps its 14 gathered rows packed as fp32 until the arithmetic needs them: widened at load they held
        // twice the registers through the whole latency window (mode 2 spilled)
        const float4 zf = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 hcf = zf, uf[ME], hf[ME], curf = zf, ninf = zf, hpcf = zf;
        if (act) {
            bool cached[ME];
            uint32_t ad[ME];
            v4u_t raw[ME];
            uint32_t goff[ME];
            hcf = gload4f(a.ph, own);
            if constexpr (PREV) hpcf = gload4f(a.hPrev, own);
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                const uint32_t off = r[i];
                hf[i] = STALE ? gload4f(a.hEdgeOld, off + voff) : gload4f(PREV ? a.hPrev : a.ph, r[ME + i] + voff);
                ad[i] = urow_addr(off, cached[i]);
                goff[i] = off + voff;
                asm("" : "+v"(goff[i]));               // stays in a VGPR (see k_stage_rec2c).  The deferred stores of that kernel
                                                       // are NOT taken over: they cost registers this one does not have
                                                       // (spills at three waves per SIMD: 26.8 -> 30.6 ms on config 5)
            }
            lds_burst<ME>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                uf[i] = __builtin_bit_cast(float4, raw[i]);
                if (!cached[i]) uf[i] = glb_row4f(puG + goff[i]);
            }
            if constexpr (MODE == 2) curf = gload4f(a.ch, own);
            if constexpr (MODE == 2 || MODE == 3) ninf = gload4f(a.nh_in, own);
        }
        double area = 1.0;
        if constexpr (FE && !LEAN) if (a.div) area = a.areaCell[c];
        const d4 hc = widen4(hcf);
        d4 t = zero, dv = zero;                                         // dv: velocityDivCell (FE), Operators.jl:18,39
        // thickness at the edge: interpolated (Operators.jl:217) or, MODE 4, what the previous step stored
        const d4 hself = PREV ? widen4(hpcf) : hc;
        auto hE = [&](const d4 &hvi) {
            if constexpr (STALE) return hvi;
            const d4 m{0.5 * (hself.x + hvi.x), 0.5 * (hself.y + hvi.y), 0.5 * (hself.z + hvi.z), 0.5 * (hself.w + hvi.w)};
            if constexpr (PREV) return round4(m);       // as the previous step stored it
            else return m;
        };
        const bool plain = __builtin_amdgcn_ballot_w64(!(mask == (1u << ME) - 1u && all)) == 0;   // see k_stage_rec2c
        if (plain) {
            if (act) {
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    const d4 uvi = widen4(uf[i]), he = hE(widen4(hf[i]));
                    t.x += uvi.x * he.x * rs[i] * invA;
                    t.y += uvi.y * he.y * rs[i] * invA;
                    t.z += uvi.z * he.z * rs[i] * invA;
                    t.w += uvi.w * he.w * rs[i] * invA;
                    if constexpr (FE && !LEAN) {
                        dv.x -= uvi.x * rs[i]; dv.y -= uvi.y * rs[i]; dv.z -= uvi.z * rs[i]; dv.w -= uvi.w * rs[i];
                    }
                }
            }
        } else if (act) {
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                const int ml = all ? K : cptr(m.mltc)[(size_t)c * ME + i];
                const bool on = (mask >> i) & 1u;
                const d4 uvi = widen4(uf[i]), he = hE(widen4(hf[i]));
                const double dx = uvi.x * he.x * rs[i] * invA;   // Operators.jl:217, DiagnosticVars.jl:165,
                const double dy = uvi.y * he.y * rs[i] * invA;   // horizontal_advection.jl:63
                const double dz = uvi.z * he.z * rs[i] * invA;
                const double dw = uvi.w * he.w * rs[i] * invA;
                if (on && k0 < ml) t.x += dx;
                if (on && k0 + 1 < ml) t.y += dy;
                if (on && k0 + 2 < ml) t.z += dz;
                if (on && k0 + 3 < ml) t.w += dw;
                if constexpr (FE && !LEAN) {
                    if (on) { dv.x -= uvi.x * rs[i]; dv.y -= uvi.y * rs[i]; dv.z -= uvi.z * rs[i]; dv.w -= uvi.w * rs[i]; }
                }
            }
        }
        d4 hs = zero;
        if (act) {
            if constexpr (MODE == 0) gstore4(a.tendH, own, t);          // the sum was formed in fp64; stored like the state, fp32
            if constexpr (MODE == 1 || MODE == 2) {
                const d4 hcur = MODE == 2 ? widen4(curf) : hc;
                const d4 nb = MODE == 2 ? widen4(ninf) : hcur;
                hs = round4(axpy4(hcur, a.a, t));                                             // time_integration.jl:125
                gstore4(a.ph_out, own, hs);
                gstore4(a.nh_out, own, axpy4(nb, a.b, t));                                    // :135
            }
            if constexpr (MODE == 3) {
                hs = round4(axpy4(widen4(ninf), a.b, t));
                gstore4(a.nh_out, own, hs);
            }
            if constexpr (FE) {                          // every output group is optional (see k_stage_rec2c)
                hs = round4(axpy4(hc, a.a, t));                                               // time_integration.jl:199
                if (LEAN || a.ph_out) gstore4(a.ph_out, own, hs);
                if constexpr (!LEAN) {
                    if (a.tendH) gstore4(a.tendH, own, t);
                    if (a.div) gstore4(a.div, own, d4{dv.x / area, dv.y / area, dv.z / area, dv.w / area});  // Operators.jl:41
                }
            }
        }
        if constexpr (MODE != 0) {
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) {
                hs = d4{hs.x + gxor(hs.x, sft), hs.y + gxor(hs.y, sft), hs.z + gxor(hs.z, sft), hs.w + gxor(hs.w, sft)};
            }
            if (l == 0 && (!FE || LEAN || a.ssh_out))                                                 // :209 (+N3), stored fp32
                reinterpret_cast<float *>(a.ssh_out)[c] = (float)(((hs.x + hs.z) + (hs.y + hs.w)) - L.rsum[ci]);
        }
    }

    // ---------------- edges ----------------
    for (int ei = grp; ei < nOwnE; ei += NG) {
        const int e = e0 + ei;
        const uint32_t *r = L.eRec + (size_t)ei * m.EI;
        const double *rw = L.woe + (size_t)ei * ME2;
        const double *rf = L.feoe + (size_t)ei * ME2;
        const uint32_t mask = r[ME2 + 2];
        const int mlt = (int)r[ME2 + 3];
        const double g = L.g[ei];
        const uint32_t own = (uint32_t)e * rowB + voff;
        float sA = 0.f, sB = 0.f;
        d4 uv[ME2], cur = zero, nin = zero;
        float4 hxf = make_float4(0.f, 0.f, 0.f, 0.f), hyf = hxf, hEof = hxf;
        if (act) {
            bool cached[ME2];
            uint32_t ad[ME2];
            v4u_t raw[ME2];
            float4 uf[ME2];
            uint32_t goff[ME2];
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                const uint32_t off = r[i];
                ad[i] = urow_addr(off, cached[i]);
                goff[i] = off + voff;
                asm("" : "+v"(goff[i]));         
            }
            lds_burst<ME2>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                uf[i] = __builtin_bit_cast(float4, raw[i]);
                if (!cached[i]) uf[i] = glb_row4f(puG + goff[i]);
            }
            if constexpr (MODE == 2) cur = gload4(a.cu, own);
            if constexpr (MODE == 2 || MODE == 3) nin = gload4(a.nu_in, own);
            if constexpr (FE && !LEAN) {
                if (a.hEdgeNew || (MODE == 5 && a.F)) {
                    hxf = gload4f(a.ph, r[ME2] * rowB + voff);         // layerThickness of cellsOnEdge[1], [2]
                    hyf = gload4f(a.ph, r[ME2 + 1] * rowB + voff);
                }
                if (a.F) {
                    if constexpr (STALE) hEof = gload4f(a.hEdgeOld, own);
                    if constexpr (PREV) {                              // the same two cells one level back, rounded as it was stored
                        const d4 px = gload4(a.hPrev, r[ME2] * rowB + voff), py = gload4(a.hPrev, r[ME2 + 1] * rowB + voff);
                        hEof = narrow4(d4{0.5 * (px.x + py.x), 0.5 * (px.y + py.y), 0.5 * (px.z + py.z), 0.5 * (px.w + py.w)});
                    }
                }
            }
            if (l == 0) {                                              // behind the gathers in the queue: nothing waits for these two alone
                sA = sshf[r[ME2]];
                sB = sshf[r[ME2 + 1]];
            }
#pragma unroll
            for (int i = 0; i < ME2; ++i) uv[i] = widen4(uf[i]);
        }
        const double ds = __shfl((double)sB - (double)sA, gbase, 64);  // ssh[c2] - ssh[c1], from the group's first lane
        const bool plain = __builtin_amdgcn_ballot_w64(!(mask == (1u << ME2) - 1u && mlt >= K)) == 0;   // wave-uniform
        if (act) {
            const bool ax = k0 < mlt, ay = k0 + 1 < mlt, az = k0 + 2 < mlt, aw = k0 + 3 < mlt;
            d4 t = zero;
            if (plain) {
                t.x -= g * ds; t.y -= g * ds; t.z -= g * ds; t.w -= g * ds;
#pragma unroll
                for (int i = 0; i < ME2; ++i) {
                    t.x += rw[i] * uv[i].x * rf[i];
                    t.y += rw[i] * uv[i].y * rf[i];
                    t.z += rw[i] * uv[i].z * rf[i];
                    t.w += rw[i] * uv[i].w * rf[i];
                }
            } else {
                if (ax) t.x -= g * ds;                                 // pressure_gradient.jl:63
                if (ay) t.y -= g * ds;
                if (az) t.z -= g * ds;
                if (aw) t.w -= g * ds;
#pragma unroll
                for (int i = 0; i < ME2; ++i) {
                    const bool on = (mask >> i) & 1u;
                    const double px = rw[i] * uv[i].x * rf[i], py = rw[i] * uv[i].y * rf[i];   // ...coriolis.jl:70-72
                    const double pz = rw[i] * uv[i].z * rf[i], pw = rw[i] * uv[i].w * rf[i];
                    if (on && ax) t.x += px;
                    if (on && ay) t.y += py;
                    if (on && az) t.z += pz;
                    if (on && aw) t.w += pw;
                }
            }
            if constexpr (MODE == 0) gstore4(a.tendU, own, t);
            if constexpr (MODE == 1) {
                const d4 up = widen4(ubuf4[(size_t)ei * K4 + l]);       // own row is in the cache
                gstore4(a.pu_out, own, axpy4(up, a.a, t));              // time_integration.jl:124
                gstore4(a.nu_out, own, axpy4(up, a.b, t));              // :134
            }
            if constexpr (MODE == 2) {
                gstore4(a.pu_out, own, axpy4(cur, a.a, t));
                gstore4(a.nu_out, own, axpy4(nin, a.b, t));
            }
            if constexpr (MODE == 3) gstore4(a.nu_out, own, axpy4(nin, a.b, t));
            if constexpr (LEAN) {
                const d4 up = widen4(ubuf4[(size_t)ei * K4 + l]);       // own row is in the cache
                gstore4(a.pu_out, own, axpy4(up, a.a, t));              // time_integration.jl:199
            }
            if constexpr (FE && !LEAN) {
                const d4 up = widen4(ubuf4[(size_t)ei * K4 + l]);       // own row is in the cache
                const d4 hx = widen4(hxf), hy = widen4(hyf);
                const d4 pE{0.5 * (hx.x + hy.x), 0.5 * (hx.y + hy.y), 0.5 * (hx.z + hy.z), 0.5 * (hx.w + hy.w)};   // layerThicknessEdge, Operators.jl:217
                const d4 hF = (STALE || PREV) ? widen4(hEof) : pE;
                if (a.F) gstore4(a.F, own, d4{up.x * hF.x, up.y * hF.y, up.z * hF.z, up.w * hF.w});                  // thicknessFlux, DiagnosticVars.jl:165
                if (a.hEdgeNew) gstore4(a.hEdgeNew, own, pE);
                if (a.pu_out) gstore4(a.pu_out, own, axpy4(up, a.a, t));   // time_integration.jl:199
                if (a.tendU) gstore4(a.tendU, own, t);
            }
        }
    }
    // ---------------- vertices: relativeVorticity of the OLD state (see k_stage_rec2c; the sums of k_curl3_f32) ----------------
    if constexpr (FE) {
        for (int vi = grp; vi < nOwnV; vi += NG) {
            if (!act) continue;
            const uint32_t *rv = Lvo + (size_t)vi * 4;
            const double *wv = Lvw + (size_t)vi * 3;
            const uint32_t own = (uint32_t)(v0 + vi) * rowB + voff;
            bool cached[3];
            uint32_t ad[3], goff[3];
            v4u_t raw[3];
            float4 uf[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint32_t off = rv[j];
                ad[j] = urow_addr(off, cached[j]);
                goff[j] = off + voff;
                asm("" : "+v"(goff[j]));
            }
            d4 c = a.accumVort ? gload4(a.vort, own) : zero;
            lds_burst<3>(raw, ad);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                uf[j] = __builtin_bit_cast(float4, raw[j]);
                if (!cached[j]) uf[j] = glb_row4f(puG + goff[j]);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const d4 uv = widen4(uf[j]);
                c.x += wv[j] * uv.x; c.y += wv[j] * uv.y; c.z += wv[j] * uv.z; c.w += wv[j] * uv.w;
            }
            gstore4(a.vort, own, c);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Forward-Euler step / reference-sequenced pieces (time_integration.jl:150-193) in one launch.
// `ops` selects which reference calls are performed, `flags` the quirks of SURVEY.md 0.6.
// All reads come from the current time level and the *old* layerThicknessEdge buffer, all writes
// go to other buffers, so the single launch is race-free.
// ------------------------------------------------------------------------------------------------
template <int LPC, int ME, int ME2>
__global__ __launch_bounds__(BLOCK) void k_fe(const MeshDev m, const FeArgs a)
{
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC);
    const int l = threadIdx.x % LPC;
    const int K = m.K;
    const int Kc = ((K + LPC - 1) / LPC) * LPC;
    const int nlev = a.nlev;
    const bool stale = a.flags & MOKA_FE_STALE_HEDGE;

    // ---------------- cells: velocityDivCell, tendLayerThickness, h update, ssh ----------------
    if (a.ops & (FE_DIV | FE_TENDH | FE_UPDATE)) {
        const int c0 = m.patchCellStart[p], c1 = m.patchCellStart[p + 1];
        for (int c = c0 + grp; c < c1; c += NG) {
            const int32_t *re = m.eoc + (size_t)c * ME;
            const int32_t *rc = m.coc + (size_t)c * ME;
            const int32_t *rm = m.mltc + (size_t)c * ME;
            const double *rs = m.sdv + (size_t)c * ME;
            const double invA = m.invArea[c];
            const double area = m.areaCell[c];
            double sshAcc = 0.0;
            bool first = true;
            for (int k = l; k < Kc; k += LPC) {
                const bool act = k < K;
                const size_t off = (size_t)c * K + k;
                double hs = 0.0;
                if (act) {
                    const double hc = a.h[off];
                    double d = 0.0, t = 0.0;
#pragma unroll
                    for (int i = 0; i < ME; ++i) {
                        const int e = re[i];
                        if (e < 0) continue;
                        const double uv = a.u[(size_t)e * K + k];
                        // DivergenceOnCell_P1/_P2 (Operators.jl:18,39): Div -= (V*dv)*sign
                        d -= uv * rs[i];
                        if ((a.ops & FE_TENDH) && k < nlev && k < rm[i]) {
                            double F;
                            if (a.ops & FE_TENDH_FROM_F) F = a.Fin[(size_t)e * K + k];
                            else if (stale) F = uv * a.hEdgeOld[(size_t)e * K + k];
                            else F = uv * (0.5 * (hc + a.h[(size_t)rc[i] * K + k]));
                            t += F * rs[i] * invA;                 // horizontal_advection.jl:63-64
                        }
                    }
                    if (a.ops & FE_DIV) a.div[off] = d / area;     // Operators.jl:41
                    if ((a.ops & FE_TENDH) && k < nlev) a.tendH[off] = t;
                    if (a.ops & FE_UPDATE) {
                        // UpdateStateVariable! (time_integration.jl:199); untouched levels carried over
                        const double hn = k < nlev ? hc + a.dt * t : hc;
                        a.h_new[off] = hn;
                        if (k < nlev) hs = hn;
                    }
                }
                sshAcc = first ? hs : sshAcc + hs;
                first = false;
            }
            if (a.ops & FE_UPDATE) {
                const double s = group_sum<LPC>(sshAcc);
                if (l == 0) a.ssh_new[c] = s - m.rsum[c];          // Update_ssh! (:209)
            }
        }
    }

    // ---------------- edges: thicknessFlux, layerThicknessEdge, tendNormalVelocity, u update ----
    if (a.ops & (FE_FLUX | FE_HEDGE | FE_TENDU | FE_UPDATE)) {
        const int e0 = m.patchEdgeStart[p], e1 = m.patchEdgeStart[p + 1];
        for (int e = e0 + grp; e < e1; e += NG) {
            const int4 hdr = *reinterpret_cast<const int4 *>(m.ehdr + (size_t)e * 4);
            const int32_t *re = m.eoe + (size_t)e * ME2;
            const double *rw = m.woe + (size_t)e * ME2;
            const double g = m.gInvDc[e];
            const double dv = m.dvEdge[e];
            const int mlt = hdr.w;
            double ds = 0.0;
            if (a.ops & FE_TENDU) ds = a.ssh[hdr.y] - a.ssh[hdr.x];
            for (int k = l; k < K; k += LPC) {
                const size_t off = (size_t)e * K + k;
                const double uk = a.u[off];
                double hfresh = 0.0;
                if (a.ops & (FE_FLUX | FE_HEDGE))
                    hfresh = 0.5 * (a.h[(size_t)hdr.x * K + k] + a.h[(size_t)hdr.y * K + k]);   // Operators.jl:217
                if ((a.ops & FE_FLUX) && k < nlev)
                    a.F[off] = uk * (stale ? a.hEdgeOld[off] : hfresh);        // DiagnosticVars.jl:165
                if (a.ops & FE_HEDGE) {
                    // levels >= nlev keep what DivergenceOnCell_P1 left in the scratch (compat) or the old value
                    a.hEdgeNew[off] = k < nlev ? hfresh : (stale ? uk * dv : a.hEdgeOld[off]);
                }
                double t = 0.0;
                if ((a.ops & FE_TENDU) && k < nlev) {
                    if (k < mlt) {
                        t -= g * ds;                                           // pressure_gradient.jl:63
#pragma unroll
                        for (int i = 0; i < ME2; ++i) {
                            const int x = re[i];
                            if (x >= 0) t += rw[i] * a.u[(size_t)x * K + k] * m.fEdge[x];   // coriolis.jl:70-72
                        }
                    }
                    a.tendU[off] = t;
                }
                if (a.ops & FE_UPDATE) a.u_new[off] = k < nlev ? uk + a.dt * t : uk;   // time_integration.jl:199
            }
        }
    }

    // ---------------- vertices: relativeVorticity (CurlOnVertex, Operators.jl:137-146) ----------
    if (a.ops & FE_CURL) {
        const int v0 = m.patchVertStart[p], v1 = m.patchVertStart[p + 1];
        const bool accum = a.flags & MOKA_FE_ACCUM_VORT;
        for (int v = v0 + grp; v < v1; v += NG) {
            for (int k = l; k < K; k += LPC) {
                const size_t off = (size_t)v * K + k;
                double cacc = accum ? a.vort[off] : 0.0;
                for (int j = 0; j < m.VD; ++j) {
                    const int e = m.eov[(size_t)v * m.VD + j];
                    cacc += m.cv[(size_t)v * m.VD + j] * a.u[(size_t)e * K + k];
                }
                a.vort[off] = cacc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// relativeVorticity alone (CurlOnVertex, Operators.jl:137-146) for even K <= 64: half a wave per vertex, a lane owns two
// levels (16-byte loads), summation in edgesOnVertex order as k_fe's vertex pass.  The companion of the tuned
// Forward-Euler step (k_stage_rec2c modes 4 / 5).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_curl2(const MeshDev m, const double *u, double *vort, int accum)
{
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    if (2 * l >= m.K) return;
    const uint32_t rowB = (uint32_t)m.K * 8u, voff = (uint32_t)l * 16u;
    const int v0 = m.patchVertStart[p], v1 = m.patchVertStart[p + 1], VD = m.VD;
    for (int v = v0 + grp; v < v1; v += BLOCK / 32) {
        const uint32_t own = (uint32_t)v * rowB + voff;
        double2 c = accum ? gload2(vort, own) : make_double2(0.0, 0.0);
        for (int j = 0; j < VD; ++j) {
            const double w = m.cv[(size_t)v * VD + j];
            const double2 uv = gload2(u, (uint32_t)m.eov[(size_t)v * VD + j] * rowB + voff);
            c.x += w * uv.x;
            c.y += w * uv.y;
        }
        gstore2(vort, own, c);
    }
}

// The same for vertexDegree 3 with the patch's vertex records (edges, weights) staged in LDS in one round trip and two
// vertices per half-wave round: 1 + 2 dependent round trips per patch instead of one per (vertex, edge).
template <int VD_>
__global__ __launch_bounds__(BLOCK) void k_curl3(const MeshDev m, const double *u, double *vort, int accum)
{
    constexpr int NG = BLOCK / 32, VCH = 64;
    __shared__ int sE[VCH * VD_];
    __shared__ double sWt[VCH * VD_];
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const bool act = 2 * l < m.K;
    const uint32_t rowB = (uint32_t)m.K * 8u, voff = (uint32_t)l * 16u;
    const int v0 = m.patchVertStart[p], v1 = m.patchVertStart[p + 1];
    for (int vb = v0; vb < v1; vb += VCH) {
        const int nv = min(VCH, v1 - vb);
        if (vb != v0) __syncthreads();
        for (int i = threadIdx.x; i < nv * VD_; i += BLOCK) { sE[i] = m.eov[(size_t)vb * VD_ + i]; sWt[i] = m.cv[(size_t)vb * VD_ + i]; }
        __syncthreads();
        if (!act) continue;
        for (int vi = grp; vi < nv; vi += 2 * NG) {
            double2 uv[2][VD_], c[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int v = vi + q * NG < nv ? vi + q * NG : vi;
#pragma unroll
                for (int j = 0; j < VD_; ++j) uv[q][j] = gload2(u, (uint32_t)sE[v * VD_ + j] * rowB + voff);
                c[q] = accum ? gload2(vort, (uint32_t)(vb + v) * rowB + voff) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int v = vi + q * NG;
                if (v >= nv) break;
#pragma unroll
                for (int j = 0; j < VD_; ++j) {
                    const double w = sWt[v * VD_ + j];
                    c[q].x += w * uv[q][j].x;
                    c[q].y += w * uv[q][j].y;
                }
                gstore2(vort, (uint32_t)(vb + v) * rowB + voff, c[q]);
            }
        }
    }
}

// relativeVorticity of an fp32-storage state (float rows, K % 4 == 0): a lane owns four levels of one vertex, the sum is
// formed in fp64 in edgesOnVertex order and stored fp32 (accum: on top of the stored value, widened)
template <int VD_>
__global__ __launch_bounds__(BLOCK) void k_curl3_f32(const MeshDev m, const float *u, float *vort, int accum)
{
    constexpr int VCH = 64;
    __shared__ int sE[VCH * VD_];
    __shared__ double sWt[VCH * VD_];
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    const int K = m.K, K4 = K >> 2;
    const int v0 = m.patchVertStart[p], v1 = m.patchVertStart[p + 1];
    for (int vb = v0; vb < v1; vb += VCH) {
        const int nv = min(VCH, v1 - vb), items = nv * K4;
        if (vb != v0) __syncthreads();
        for (int i = threadIdx.x; i < nv * VD_; i += BLOCK) { sE[i] = m.eov[(size_t)vb * VD_ + i]; sWt[i] = m.cv[(size_t)vb * VD_ + i]; }
        __syncthreads();
        for (int it = threadIdx.x; it < items; it += 2 * BLOCK) {
            float4 uf[2][VD_], cf[2];
            int vq[2], lq[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int item = it + q * BLOCK < items ? it + q * BLOCK : it;
                vq[q] = item / K4; lq[q] = item - vq[q] * K4;
#pragma unroll
                for (int j = 0; j < VD_; ++j) uf[q][j] = reinterpret_cast<const float4 *>(u + (size_t)sE[vq[q] * VD_ + j] * K)[lq[q]];
                cf[q] = accum ? reinterpret_cast<const float4 *>(vort + (size_t)(vb + vq[q]) * K)[lq[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (it + q * BLOCK >= items) break;
                d4 c = widen4(cf[q]);
#pragma unroll
                for (int j = 0; j < VD_; ++j) {
                    const double w = sWt[vq[q] * VD_ + j];
                    const d4 uv = widen4(uf[q][j]);
                    c.x += w * uv.x; c.y += w * uv.y; c.z += w * uv.z; c.w += w * uv.w;
                }
                reinterpret_cast<float4 *>(vort + (size_t)(vb + vq[q]) * K)[lq[q]] = narrow4(c);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone operators on arbitrary (K,n) arrays in the new numbering (Operators.jl).
// ------------------------------------------------------------------------------------------------
template <int LPC>
__global__ __launch_bounds__(BLOCK) void k_operator(const MeshDev m, const OpArgs a)
{
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC);
    const int l = threadIdx.x % LPC;
    const int K = m.K;
    if (a.op == OP_GRADIENT || a.op == OP_INTERP || a.op == OP_DIV_P1) {
        const int e0 = m.patchEdgeStart[p], e1 = m.patchEdgeStart[p + 1];
        for (int e = e0 + grp; e < e1; e += NG) {
            const int c1 = m.ehdr[(size_t)e * 4], c2 = m.ehdr[(size_t)e * 4 + 1];
            for (int k = l; k < K; k += LPC) {
                const size_t off = (size_t)e * K + k;
                if (a.op == OP_GRADIENT)       // (S[k,c2]-S[k,c1]) / dcEdge     Operators.jl:97
                    a.out[off] = (a.in[(size_t)c2 * K + k] - a.in[(size_t)c1 * K + k]) / m.dcEdge[e];
                else if (a.op == OP_INTERP) {  // 0.5*(C[k,c1]+C[k,c2])          Operators.jl:217
                    if (k < a.nlev) a.out[off] = 0.5 * (a.in[(size_t)c1 * K + k] + a.in[(size_t)c2 * K + k]);
                } else                         // temp = V*dvEdge                Operators.jl:18
                    a.out[off] = a.in[off] * m.dvEdge[e];
            }
        }
    } else if (a.op == OP_DIV_P2) {
        const int c0 = m.patchCellStart[p], c1 = m.patchCellStart[p + 1];
        for (int c = c0 + grp; c < c1; c += NG) {
            for (int k = l; k < K; k += LPC) {
                double d = 0.0;
                for (int i = 0; i < m.ME; ++i) {
                    const int e = m.eoc[(size_t)c * m.ME + i];
                    if (e >= 0) d -= a.in[(size_t)e * K + k] * m.sdv[(size_t)c * m.ME + i];   // Operators.jl:18,39
                }
                a.out[(size_t)c * K + k] = d / m.areaCell[c];                                // :41
            }
        }
    } else if (a.op == OP_CURL) {
        const int v0 = m.patchVertStart[p], v1 = m.patchVertStart[p + 1];
        for (int v = v0 + grp; v < v1; v += NG) {
            for (int k = l; k < K; k += LPC) {
                const size_t off = (size_t)v * K + k;
                double cacc = a.out[off];                                                    // accumulates
                for (int j = 0; j < m.VD; ++j)
                    cacc += m.cv[(size_t)v * m.VD + j] * a.in[(size_t)m.eov[(size_t)v * m.VD + j] * K + k];
                a.out[off] = cacc;
            }
        }
    } else if (a.op == OP_GRAD_T) {
        // transpose of GradientOnEdge (Operators.jl:97): dS[k,c] += sum_i sign[i,c] * (dG[k,e_i] / dcEdge[e_i]), edgesOnCell order
        const int c0 = m.patchCellStart[p], c1 = m.patchCellStart[p + 1];
        for (int c = c0 + grp; c < c1; c += NG) {
            for (int k = l; k < K; k += LPC) {
                double acc = a.out[(size_t)c * K + k];
                for (int i = 0; i < m.ME; ++i) {
                    const int e = m.eoc[(size_t)c * m.ME + i];
                    if (e >= 0) acc += (m.sdv[(size_t)c * m.ME + i] < 0.0 ? -1.0 : 1.0) * (a.in[(size_t)e * K + k] / m.dcEdge[e]);
                }
                a.out[(size_t)c * K + k] = acc;
            }
        }
    } else if (a.op == OP_DIV_T) {
        // transpose of DivergenceOnCell_P2, then _P1 (Operators.jl:34-42, :18): t = dTemp - s1 * (dD[c1] / area[c1]) - s2 * (...);
        // dV += t * dvEdge
        const int e0 = m.patchEdgeStart[p], e1 = m.patchEdgeStart[p + 1];
        for (int e = e0 + grp; e < e1; e += NG) {
            const int c1 = m.ehdr[(size_t)e * 4], c2 = m.ehdr[(size_t)e * 4 + 1];
            const double s1 = a.auxD[(size_t)e * 2], s2 = a.auxD[(size_t)e * 2 + 1];
            for (int k = l; k < K; k += LPC) {
                const size_t off = (size_t)e * K + k;
                double t = a.in2 ? a.in2[off] : 0.0;
                t -= s1 * (a.in[(size_t)c1 * K + k] / m.areaCell[c1]);
                t -= s2 * (a.in[(size_t)c2 * K + k] / m.areaCell[c2]);
                a.out[off] += t * m.dvEdge[e];
            }
        }
    } else if (a.op == OP_CURL_T) {
        // transpose of CurlOnVertex (Operators.jl:137-146): dV[k,e] += sum over the (vertex, slot) pairs naming e, in ascending
        // (caller's vertex id, slot) order, of coefficient * dC[k,v]
        const int e0 = m.patchEdgeStart[p], e1 = m.patchEdgeStart[p + 1];
        for (int e = e0 + grp; e < e1; e += NG) {
            for (int k = l; k < K; k += LPC) {
                double acc = a.out[(size_t)e * K + k];
                for (int q = 0; q < a.auxW; ++q) {
                    const int v = a.auxI[(size_t)e * a.auxW + q];
                    if (v >= 0) acc += a.auxD[(size_t)e * a.auxW + q] * a.in[(size_t)v * K + k];
                }
                a.out[(size_t)e * K + k] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ssh from layerThickness (Update_ssh!, time_integration.jl:205-211, N3 column sum)
// ------------------------------------------------------------------------------------------------
template <int LPC, class T>
__global__ __launch_bounds__(BLOCK) void k_update_ssh(const MeshDev m, const T *h, T *ssh, int nlev)
{
    constexpr int NG = BLOCK / LPC;
    const int grp = threadIdx.x / LPC, l = threadIdx.x % LPC;
    const int K = m.K;
    const int Kc = ((nlev + LPC - 1) / LPC) * LPC;
    for (int c = blockIdx.x * NG + grp; c < m.nC; c += gridDim.x * NG) {
        double acc = 0.0;
        bool first = true;
        for (int k = l; k < Kc; k += LPC) {
            const double v = k < nlev ? (double)h[(size_t)c * K + k] : 0.0;
            acc = first ? v : acc + v;
            first = false;
        }
        const double s = group_sum<LPC>(acc);
        if (l == 0) ssh[c] = (T)(s - m.rsum[c]);        // fp32 state: stored rounded
    }
}

// ------------------------------------------------------------------------------------------------
// row permutation between the caller's numbering and the device numbering
//   to_device: dev[n][k] = host_order[n2o[n]][k]     else: host_order[n2o[n]][k] = dev[n][k]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_permute_rows(double *dst, const double *src, const int32_t *n2o,
                                                       int64_t n, int K, int to_device)
{
    const int64_t total = n * K;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        const int64_t r = i / K;
        const int k = (int)(i - r * K);
        const int64_t o = (int64_t)n2o[r] * K + k;
        if (to_device) dst[i] = src[o];
        else dst[o] = src[i];
    }
}

// sumArray (run_loop.jl:47-51): strictly serial sum_j a[j]^2 in the caller's numbering, one wave.
__global__ __launch_bounds__(64) void k_sum_sq_serial(const double *a, int64_t n, double *out)
{
    const int lane = threadIdx.x;
    double sum = 0.0;
    for (int64_t base = 0; base < n; base += 64) {
        const int64_t j = base + lane;
        const double v = j < n ? a[j] : 0.0;
        const int cnt = (int)((n - base) < 64 ? (n - base) : 64);
        for (int t = 0; t < cnt; ++t) {
            const double x = __shfl(v, t, 64);
            sum = sum + x * x;
        }
    }
    if (lane == 0) *out = sum;
}

// halo pack (unpack = 0): buf[i][k] = field[rows[i]][k];  unpack: field[rows[i]][k] = buf[i][k]
__global__ __launch_bounds__(BLOCK) void k_pack_rows(double *buf, double *field, const int32_t *rows, int64_t n, int K, int unpack)
{
    const int64_t total = n * K;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        const int64_t r = i / K;
        const int k = (int)(i - r * K);
        const int64_t o = (int64_t)rows[r] * K + k;
        if (unpack) field[o] = buf[i];
        else buf[i] = field[o];
    }
}

// halo pack / unpack through an element map: map[j] = (field tag << 30) | element index, tag 0 = h, 1 = ssh, 2 = u
__global__ __launch_bounds__(BLOCK) void k_halo_map(double *buf, double *h, double *ssh, double *u, const uint32_t *map,
                                                   int64_t n, int unpack)
{
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * BLOCK) {
        const uint32_t mj = map[j];
        const uint32_t tag = mj >> 30, idx = mj & 0x3FFFFFFFu;
        double *f = tag == 0 ? h : tag == 1 ? ssh : u;
        if (unpack) f[idx] = buf[j];
        else buf[j] = f[idx];
    }
}

// fp32-state forms: the caller's side (host order) is always double, the device field is float
__global__ __launch_bounds__(BLOCK) void k_permute_rows_f32(void *dst, const void *src, const int32_t *n2o, int64_t n, int K,
                                                           int to_device)
{
    const int64_t total = n * K;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        const int64_t r = i / K;
        const int k = (int)(i - r * K);
        const int64_t o = (int64_t)n2o[r] * K + k;
        if (to_device) static_cast<float *>(dst)[i] = (float)static_cast<const double *>(src)[o];
        else static_cast<double *>(dst)[o] = (double)static_cast<const float *>(src)[i];
    }
}

__global__ __launch_bounds__(BLOCK) void k_halo_map_f32(float *buf, float *h, float *ssh, float *u, const uint32_t *map,
                                                       int64_t n, int unpack)
{
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * BLOCK) {
        const uint32_t mj = map[j];
        const uint32_t tag = mj >> 30, idx = mj & 0x3FFFFFFFu;
        float *f = tag == 0 ? h : tag == 1 ? ssh : u;
        if (unpack) f[idx] = buf[j];
        else buf[j] = f[idx];
    }
}

__global__ __launch_bounds__(BLOCK) void k_copy(double *dst, const double *src, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[i] = src[i];
}

// Bandwidth probes for bench.py's same-run calibration (moka_bw_probe).  Shapes chosen by measurement
// (tools/micro/bw_shapes.hip, profiles/r03_variants.txt): one 16-byte word per thread with the grid covering the buffer copies
// at 6.2 TB/s (the guide's 6.29 TB/s "float4 copy"), grid-stride forms of the same copy stay at 4.7-5.8; a read-only sweep
// reaches 6.4 TB/s with plain loads and 7.0-7.1 with nontemporal ones.
//   k_bw_copy   : dst[i] = src[i], n 16-byte words                                  (2 * 16 * n bytes of traffic)
//   k_bw_read   : every word read once, four nontemporal loads in flight per lane; the XOR of everything read decides a
//                 store that never happens, so no load can be dropped
//   k_bw_gather : the access pattern of the stage kernels -- rows of `rowB` bytes (a half-wave per row, 16 bytes per lane,
//                 four rows in flight) fetched in a scattered order: row (i * 40503 + 977 * (i >> 7)) mod nRows, i.e.
//                 consecutive half-waves touch rows far apart, every row about once
typedef unsigned int bw_v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(BLOCK) void k_bw_copy(bw_v4u *__restrict__ dst, const bw_v4u *__restrict__ src, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ __launch_bounds__(BLOCK) void k_bw_read(const bw_v4u *__restrict__ src, int64_t n, uint32_t *sink)
{
    const int64_t b = (int64_t)blockIdx.x * (4 * BLOCK) + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = b + j * BLOCK;
        if (i < n) {
            const bw_v4u a = __builtin_nontemporal_load(&src[i]);
            acc ^= a.x ^ a.y ^ a.z ^ a.w;
        }
    }
    if (acc == 0x9E3779B9u) *sink = acc;          // the buffers hold a byte pattern whose XOR never gives this
}

__global__ __launch_bounds__(BLOCK) void k_bw_gather(const unsigned char *__restrict__ src, int64_t nRows, uint32_t rowB, int64_t nFetch,
                                                    uint32_t *sink)
{
    const int l = threadIdx.x & 31;
    const int64_t hw = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * BLOCK) >> 5;
    const bool act = (uint32_t)l * 16u < rowB;
    uint32_t acc = 0;
    for (int64_t i = hw * 4; i < nFetch; i += nhw * 4) {
        bw_v4u v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t f = i + j < nFetch ? i + j : i;
            const int64_t r = (f * 40503 + 977 * (f >> 7)) % nRows;
            v[j] = act ? *reinterpret_cast<const bw_v4u *>(src + r * rowB + (size_t)l * 16u) : bw_v4u{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x9E3779B9u) *sink = acc;
}

// k_bw_streams: five streams at once, three read and two written, one 16-byte word per thread and stream -- the traffic mix of the RK
// stage launches of modes 2 / 3 (Provis and Curr and New in, Provis' and New out), which the boxes that run this library 10 % slower
// slow down most, while their copy, read and gather rates are those of the fast boxes (profiles/r03_variants.txt)
__global__ __launch_bounds__(BLOCK) void k_bw_streams(bw_v4u *__restrict__ o1, bw_v4u *__restrict__ o2, const bw_v4u *__restrict__ a,
                                                     const bw_v4u *__restrict__ b, const bw_v4u *__restrict__ c, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) {
        const bw_v4u x = a[i], y = b[i], z = c[i];
        o1[i] = x ^ y;
        o2[i] = x ^ z;
    }
}

// `bytes` = the whole buffer: split into five equal parts (a, b, c read; two written)
hipError_t launch_bw_streams(void *buf, int64_t bytes, hipStream_t s)
{
    const int64_t n = bytes / 5 / 16;
    bw_v4u *p = static_cast<bw_v4u *>(buf);
    hipLaunchKernelGGL(k_bw_streams, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, p + 3 * n, p + 4 * n, p, p + n, p + 2 * n, n);
    return hipGetLastError();
}

hipError_t launch_bw_copy(void *dst, const void *src, int64_t bytes, int, hipStream_t s)
{
    const int64_t n = bytes / 16;
    hipLaunchKernelGGL(k_bw_copy, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, static_cast<bw_v4u *>(dst),
                       static_cast<const bw_v4u *>(src), n);
    return hipGetLastError();
}

hipError_t launch_bw_read(const void *src, int64_t bytes, uint32_t *sink, int, hipStream_t s)
{
    const int64_t n = bytes / 16;
    hipLaunchKernelGGL(k_bw_read, dim3((unsigned)((n + 4 * BLOCK - 1) / (4 * BLOCK))), dim3(BLOCK), 0, s, static_cast<const bw_v4u *>(src), n, sink);
    return hipGetLastError();
}

// every row of the buffer (bytes / rowB of them) fetched about once, in the scattered order above
hipError_t launch_bw_gather(const void *src, int64_t bytes, uint32_t rowB, uint32_t *sink, int nCUs, hipStream_t s)
{
    const int64_t nRows = bytes / rowB;
    hipLaunchKernelGGL(k_bw_gather, dim3((unsigned)(nCUs * 16)), dim3(BLOCK), 0, s, static_cast<const unsigned char *>(src), nRows, rowB, nRows, sink);
    return hipGetLastError();
}

// the same gather with the footprint (nRows) and the number of fetches given separately
hipError_t launch_bw_gather_n(const void *src, int64_t nRows, uint32_t rowB, int64_t nFetch, uint32_t *sink, int nCUs, hipStream_t s)
{
    hipLaunchKernelGGL(k_bw_gather, dim3((unsigned)(nCUs * 16)), dim3(BLOCK), 0, s, static_cast<const unsigned char *>(src), nRows, rowB, nFetch, sink);
    return hipGetLastError();
}

template <int LPC>
static hipError_t launch_fe_lpc(const MeshDev &m, const FeArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(m)), b(BLOCK);
    if (m.ME == 6 && m.ME2 == 10) hipLaunchKernelGGL((k_fe<LPC, 6, 10>), g, b, 0, s, m, a);
    else if (m.ME == 8 && m.ME2 == 14) hipLaunchKernelGGL((k_fe<LPC, 8, 14>), g, b, 0, s, m, a);
    else if (m.ME <= 6 && m.ME2 <= 14) hipLaunchKernelGGL((k_fe<LPC, 6, 14>), g, b, 0, s, m, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}


template <int ME, int ME2, int NT>
static bool launch_rec2c_nt(const ColMesh &m, const StageArgs &a, int mode, dim3 g, size_t lds, int mE, int mC, hipStream_t s)
{
    const dim3 b(NT);
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 0, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 1: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 1, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 2: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 2, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 3: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 3, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 4: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 4, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 5: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 5, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 6: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 6, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 7: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 7, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 8: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 8, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 9: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 9, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 10: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 10, NT>), g, b, lds, s, m, a, mE, mC); return true;
        case 11: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 11, NT>), g, b, lds, s, m, a, mE, mC); return true;
    }
    return false;
}

// A launch of at most two workgroups per CU (the boundary patches of a partitioned mesh: ~180 of them, alone on the chip) is
// as long as ONE workgroup lives: 512 threads per patch halve the number of entity rounds it makes (1 + 3 + 1 instead of
// 1 + 6 + 2).  Whole-mesh launches keep 256 (four workgroups per CU; 512 threads tie there, profiles/r01_variants.txt).
template <int ME, int ME2>
static bool launch_rec2c(const ColMesh &m, const StageArgs &a, int mode, dim3 g, dim3, size_t lds, int mE, int mC, hipStream_t s)
{
    if (m.nPatches <= 512) return launch_rec2c_nt<ME, ME2, 512>(m, a, mode, g, lds, mE, mC, s);
    return launch_rec2c_nt<ME, ME2, BLOCK>(m, a, mode, g, lds, mE, mC, s);
}

// measurement: 0 = lean launches run the general Forward-Euler instances (outputs tested at run time) instead of modes 10 / 11
static std::atomic<int> g_feLeanInst{1};
void set_fe_lean_instances(int on) { g_feLeanInst.store(on); }
int fe_lean_instances() { return g_feLeanInst.load(); }

// the pair form (launch_stage_rec2c): 512-thread workgroups over two patches, whose records and rows need more than the default
// 64 KB of dynamic LDS.  Default: the tendency launch (mode 0) and RK stage 1 (mode 1; 7 in the 13-stream form) -- measured on the
// product mesh (P = 16, three interleaved rounds): tendency 1.188 -> 1.170 ms, stage 1 1.513 -> 1.479, stage 4 (mode 3) 1.443 ->
// 1.461: mode 3 does not gain, stages 2 / 3 lose (profiles/r04_variants.txt section 8)
static std::atomic<int> g_pairModes{(1 << 0) | (1 << 1) | (1 << 7)};
void set_pair_modes(int mask) { g_pairModes.store(mask); }
int pair_modes() { return g_pairModes.load(); }

template <int ME, int ME2>
static bool launch_rec2c_pair(const ColMesh &m, const StageArgs &a, int mode, dim3 g, size_t lds, int mE, int mC, hipStream_t s)
{
    const dim3 b(512);
    // (once per device and mode, and never inside a stream capture: moka_run steps eagerly before it captures)
    const bool raise = lds > 64 * 1024 && lds_attr_needed(22 + mode);
#define PAIR_CASE(M)                                                                                                                      \
    case M:                                                                                                                               \
        if (raise)                                                                                                                        \
            (void)hipFuncSetAttribute((const void *)k_stage_rec2c<ME, ME2, M, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, M, 512>), g, b, lds, s, m, a, mE, mC);                                                  \
        return true;
    switch (mode) {
        PAIR_CASE(0) PAIR_CASE(1) PAIR_CASE(2) PAIR_CASE(3) PAIR_CASE(7) PAIR_CASE(8) PAIR_CASE(9)
    }
#undef PAIR_CASE
    return false;
}

size_t rec2c_lds_bytes(const MeshDev &md)
{
    return ((rec_lds_bytes(md) + 15) & ~(size_t)15) + (size_t)md.maxOwnE * md.K * 8 + 16;
}

// the vertex records a Forward-Euler launch with the vertex pass stages behind the row cache: 3 coefficients + 4 offsets per vertex
static inline size_t vert_lds_bytes(const MeshDev &md) { return (size_t)md.maxOwnV * (3 * 8 + 4 * 4) + 16; }

// measurement: 0 = the vertex pass always gets a launch of its own (k_curl3)
static std::atomic<int> g_curlFused{1};
void set_curl_fused(int on) { g_curlFused.store(on); }
int curl_fused() { return g_curlFused.load(); }

bool stage_curl_fused(const MeshDev &md)
{
    return g_curlFused.load() && md.vRec != nullptr && md.VD == 3 && md.patchVertStart != nullptr && md.maxOwnV > 0;
}

// ... and do its vertex records fit beside the records and own rows of patches as large as md.maxOwnE / md.maxOwnC
bool stage_curl_fits(const MeshDev &md, bool f32)
{
    if (!stage_curl_fused(md)) return false;
    const size_t rows = (size_t)md.maxOwnE * md.K * (f32 ? 4 : 8);
    const size_t lds = ((rec_lds_bytes(md) + 15) & ~(size_t)15) + rows + 16 + vert_lds_bytes(md) + 16;
    return f32 ? (stage_f32_supported(md) && lds <= 160 * 1024) : (rec2c_supported(md) && lds <= 64 * 1024);
}

static inline ColMesh col_mesh(const MeshDev &md, int nLaunch)
{
    return ColMesh{md.nC, md.nE, md.K, nLaunch, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                   md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc, md.tailPatch >= 0 ? md.tailPatch + 1 : 0,
                   md.patchVertStart, md.vRec, md.cv, md.maxOwnV, 0};
}

// can launch_stage_rec2c serve whole-mesh launches of this mesh at all (even K <= 64, records + own rows within 64 KB of LDS)
bool rec2c_supported(const MeshDev &md)
{
    const bool shape = (md.ME == 6 && md.ME2 == 10) || (md.ME == 8 && md.ME2 == 14) || (md.ME <= 6 && md.ME2 <= 14);
    return md.cRec && md.eRec && md.K <= 64 && !(md.K & 1) && rec2c_lds_bytes(md) <= 64 * 1024 && md.maxOwnC >= 1 && md.maxOwnE >= 1 && shape;
}

hipError_t launch_stage_rec2c(const MeshDev &md, const StageArgs &a, hipStream_t s)
{
    const int nLaunch = md.nPatches + (md.tailPatch >= 0 ? 1 : 0);
    const dim3 g(8 * ((nLaunch + 7) / 8)), b(BLOCK);
    const ColMesh m = col_mesh(md, nLaunch);
    int mode = colp_mode(a);
    if (a.vort && (mode < 4 || !stage_curl_fused(md))) return hipErrorNotSupported;
    if (fe_lean_instances() && colp_lean(a, mode)) mode += 5;        // a lean Forward-Euler launch: modes 10 / 11
    const size_t lds = rec2c_lds_bytes(md) + (a.vort ? vert_lds_bytes(md) : 0);
    if (mode < 0 || md.K > 64 || (md.K & 1) || lds > 64 * 1024 || md.maxOwnC < 1 || md.maxOwnE < 1) return hipErrorNotSupported;
    // Two consecutive patches per 512-thread workgroup for the LIGHT modes (round 4; moka_set_tuning key 8).  Consecutive patches of
    // the bisection order are neighbours, so one row cache over both serves 7-11 % of what they would otherwise fetch from each
    // other through the L2 (FETCH_SIZE, profiles/r04_variants.txt section 8): the tendency launch and RK stage 1 gain 1.5-2 % on the
    // product mesh (4-10 % on a mesh bisected down to 32-cell patches); stage 4 does not, stages 2 and 3 (five streams in flight per
    // entity) lose: those keep one patch per 256-thread workgroup.  Same entities, same arithmetic, same bits.  Only launches large enough to fill the chip either way,
    // without a tail patch, and when two such workgroups fit a CU's LDS (two per CU = the 16 waves of four small workgroups).
    // (bit 16 of the mask is a test hook: small launches too)
    if (((pair_modes() >> mode) & 1) && md.tailPatch < 0 && (md.nPatches >= 4096 || ((pair_modes() >> 16) & 1)) && md.nPatches >= 2 &&
        md.ME == 6 && md.ME2 == 10) {
        MeshDev two = md;
        two.maxOwnE = 2 * md.maxOwnE; two.maxOwnC = 2 * md.maxOwnC;
        const size_t lds2 = rec2c_lds_bytes(two);
        if (2 * lds2 + 2048 <= 160 * 1024) {
            const int nPair = (md.nPatches + 1) / 2;
            ColMesh mp = col_mesh(md, nPair);
            mp.pairEnd = md.patchBegin + md.nPatches;
            if (launch_rec2c_pair<6, 10>(mp, a, mode, dim3(8 * ((nPair + 7) / 8)), lds2, two.maxOwnE, two.maxOwnC, s)) return hipGetLastError();
        }
    }
    bool ok = false;
    if (md.ME == 6 && md.ME2 == 10) ok = launch_rec2c<6, 10>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME == 8 && md.ME2 == 14) ok = launch_rec2c<8, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_rec2c<6, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    return ok ? hipGetLastError() : hipErrorNotSupported;
}

// Which modes of the fp32-storage kernel run as 512-thread workgroups bounded to 128 registers (two per CU = 4 waves per
// SIMD instead of three 4-wave workgroups = 3): bit m = mode m.  Modes 0 and 1 (no Curr / New rows in flight) fit 128
// registers without spills, the others do not (tools/kernel_regs.py).  Measured on config 5 (profiles/r03_variants.txt):
// mode 0, the tendency launch, 3.81 -> 3.71 ms; mode 1, RK stage 1, 4.81 -> 5.07 ms (its two result streams per entity
// queue up behind twice as many waves) -- so mode 0 only.  moka_set_tuning(1, mask) changes it for measurements.
static std::atomic<int> g_f32WideModes{1 << 0};
void set_f32_wide_modes(int mask) { g_f32WideModes.store(mask); }
int f32_wide_modes() { return g_f32WideModes.load(); }
// measurement: 0 = every Forward-Euler step stores all of its DiagnosticVars / TendencyVars (no lean steps)
static std::atomic<int> g_feLean{1};
void set_fe_lean(int on) { g_feLean.store(on); }
int fe_lean_enabled() { return g_feLean.load(); }
// measurement: 0 keeps Forward-Euler steps on the gathered layerThicknessEdge (mode 4) even when mode 6 applies
static std::atomic<int> g_fePrevMode{1};
void set_fe_prev_mode(int on) { g_fePrevMode.store(on); }
int fe_prev_mode() { return g_fePrevMode.load(); }

template <int ME, int ME2, int NT, int WPE>
static bool launch_rec2c_f32_nt(const ColMesh &m, const StageArgs &a, int mode, dim3 g, size_t lds, int mE, int mC, hipStream_t s)
{
    // a launch that carries a halo-straddling patch of a partitioned mesh (up to 6 own edges per cell) may need more than
    // the default 64 KB of dynamic LDS; such launches are small (the boundary group), occupancy does not matter there
    if (lds > 64 * 1024 && lds_attr_needed((ME == 6 ? (ME2 == 10 ? 0 : 1) : 2) * 2 + (NT == 512 ? 1 : 0))) {
        (void)hipFuncSetAttribute((const void *)k_stage_rec2c_f32<ME, ME2, 0, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stage_rec2c_f32<ME, ME2, 1, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stage_rec2c_f32<ME, ME2, 2, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stage_rec2c_f32<ME, ME2, 3, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stage_rec2c_f32<ME, ME2, 4, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stage_rec2c_f32<ME, ME2, 5, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stage_rec2c_f32<ME, ME2, 6, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stage_rec2c_f32<ME, ME2, 10, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stage_rec2c_f32<ME, ME2, 11, NT, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const dim3 b(NT);
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 0, NT, WPE>), g, b, lds, s, m, a, mE, mC); return true;
        case 1: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 1, NT, WPE>), g, b, lds, s, m, a, mE, mC); return true;
        case 2: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 2, NT, WPE>), g, b, lds, s, m, a, mE, mC); return true;
        case 3: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 3, NT, WPE>), g, b, lds, s, m, a, mE, mC); return true;
        case 4: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 4, NT, WPE>), g, b, lds, s, m, a, mE, mC); return true;
        case 5: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 5, NT, WPE>), g, b, lds, s, m, a, mE, mC); return true;
        case 6: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 6, NT, WPE>), g, b, lds, s, m, a, mE, mC); return true;
        case 10: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 10, NT, WPE>), g, b, lds, s, m, a, mE, mC); return true;
        case 11: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 11, NT, WPE>), g, b, lds, s, m, a, mE, mC); return true;
    }
    return false;
}

template <int ME, int ME2>
static bool launch_rec2c_f32(const ColMesh &m, const StageArgs &a, int mode, dim3 g, dim3, size_t lds, int mE, int mC, hipStream_t s)
{
    // two 512-thread workgroups must fit a CU's 160 KB of LDS for the wide form to mean 4 waves per SIMD
    if (((f32_wide_modes() >> mode) & 1) && 2 * lds <= 160 * 1024)
        return launch_rec2c_f32_nt<ME, ME2, 512, 4>(m, a, mode, g, lds, mE, mC, s);
    return launch_rec2c_f32_nt<ME, ME2, BLOCK, 3>(m, a, mode, g, lds, mE, mC, s);
}

// fp32-state meshes: K % 4 == 0, K <= 128, byte-offset records, records + own u rows of the largest launched patch within
// the 160 KB of LDS a workgroup can have (64 KB without raising the kernel attribute: every whole-mesh launch stays below)
bool stage_f32_supported(const MeshDev &md)
{
    const size_t lds = ((rec_lds_bytes(md) + 15) & ~(size_t)15) + (size_t)md.maxOwnE * md.K * 4 + 16;
    const bool shape = (md.ME == 6 && md.ME2 == 10) || (md.ME == 8 && md.ME2 == 14) || (md.ME <= 6 && md.ME2 <= 14);
    return md.cRec && md.eRec && md.K >= 4 && md.K <= 128 && (md.K & 3) == 0 && lds <= 160 * 1024 && md.maxOwnC >= 1 &&
           md.maxOwnE >= 1 && shape;
}

hipError_t launch_stage_rec2c_f32(const MeshDev &md, const StageArgs &a, hipStream_t s)
{
    const int nLaunch = md.nPatches + (md.tailPatch >= 0 ? 1 : 0);
    const dim3 g(8 * ((nLaunch + 7) / 8)), b(BLOCK);
    const ColMesh m = col_mesh(md, nLaunch);
    int mode = colp_mode(a);
    if (mode < 0 || !stage_f32_supported(md)) return hipErrorNotSupported;
    if (a.vort && (mode < 4 || !stage_curl_fused(md))) return hipErrorNotSupported;
    if (fe_lean_instances() && colp_lean(a, mode)) mode += 5;        // a lean Forward-Euler launch: modes 10 / 11
    const size_t lds = ((rec_lds_bytes(md) + 15) & ~(size_t)15) + (size_t)md.maxOwnE * md.K * 4 + 16 + (a.vort ? vert_lds_bytes(md) + 16 : 0);
    if (lds > 160 * 1024) return hipErrorNotSupported;
    bool ok = false;
    if (md.ME == 6 && md.ME2 == 10) ok = launch_rec2c_f32<6, 10>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME == 8 && md.ME2 == 14) ok = launch_rec2c_f32<8, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_rec2c_f32<6, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    return ok ? hipGetLastError() : hipErrorNotSupported;
}

// valid for even K <= 64 on meshes whose rows stay below 4 GiB (the caller checks: the tuned Forward-Euler path)
hipError_t launch_curl2(const MeshDev &m, const double *u, double *vort, bool accum, hipStream_t s)
{
    if (m.K > 64 || (m.K & 1) || (size_t)std::max(m.nE, m.nV) * m.K * 8 >= ((size_t)1 << 32)) return hipErrorNotSupported;
    if (m.VD == 3) hipLaunchKernelGGL((k_curl3<3>), dim3(patch_grid(m)), dim3(BLOCK), 0, s, m, u, vort, accum ? 1 : 0);
    else hipLaunchKernelGGL(k_curl2, dim3(patch_grid(m)), dim3(BLOCK), 0, s, m, u, vort, accum ? 1 : 0);
    return hipGetLastError();
}

// fp32-storage states (float rows, K % 4 == 0, vertexDegree 3)
hipError_t launch_curl_f32(const MeshDev &m, const float *u, float *vort, bool accum, hipStream_t s)
{
    if ((m.K & 3) || m.VD != 3) return hipErrorNotSupported;
    hipLaunchKernelGGL((k_curl3_f32<3>), dim3(patch_grid(m)), dim3(BLOCK), 0, s, m, u, vort, accum ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_fe(const MeshDev &m, const FeArgs &a, int lpc, hipStream_t s)
{
#define CALL(L) launch_fe_lpc<L>(m, a, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

template <int LPC>
static hipError_t launch_operator_lpc(const MeshDev &m, const OpArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL((k_operator<LPC>), dim3(patch_grid(m)), dim3(BLOCK), 0, s, m, a);
    return hipGetLastError();
}

hipError_t launch_operator(const MeshDev &m, const OpArgs &a, int lpc, hipStream_t s)
{
#define CALL(L) launch_operator_lpc<L>(m, a, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

template <int LPC, class T>
static hipError_t launch_update_ssh_lpc(const MeshDev &m, const T *h, T *ssh, int nlev, hipStream_t s)
{
    const int ng = BLOCK / LPC;
    int grid = (m.nC + ng - 1) / ng;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL((k_update_ssh<LPC, T>), dim3(grid), dim3(BLOCK), 0, s, m, h, ssh, nlev);
    return hipGetLastError();
}

hipError_t launch_update_ssh(const MeshDev &m, const double *h, double *ssh, int nlev, int lpc, hipStream_t s)
{
#define CALL(L) launch_update_ssh_lpc<L, double>(m, h, ssh, nlev, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

hipError_t launch_update_ssh_f32(const MeshDev &m, const float *h, float *ssh, int nlev, int lpc, hipStream_t s)
{
#define CALL(L) launch_update_ssh_lpc<L, float>(m, h, ssh, nlev, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

hipError_t launch_permute_rows(double *dst, const double *src, const int32_t *n2o, int64_t n, int K, int to_device,
                               hipStream_t s)
{
    int64_t blocks = (n * K + BLOCK - 1) / BLOCK;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)blocks), dim3(BLOCK), 0, s, dst, src, n2o, n, K, to_device);
    return hipGetLastError();
}

hipError_t launch_sum_sq_serial(const double *a, int64_t n, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_sum_sq_serial, dim3(1), dim3(64), 0, s, a, n, out);
    return hipGetLastError();
}

hipError_t launch_pack_rows(double *buf, const double *field, const int32_t *rows, int64_t n, int K, int unpack, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n * K + BLOCK - 1) / BLOCK;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)blocks), dim3(BLOCK), 0, s, buf, const_cast<double *>(field), rows, n, K, unpack);
    return hipGetLastError();
}

hipError_t launch_halo_map(double *buf, double *h, double *ssh, double *u, const uint32_t *map, int64_t n, int unpack,
                           hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_halo_map, dim3((unsigned)blocks), dim3(BLOCK), 0, s, buf, h, ssh, u, map, n, unpack);
    return hipGetLastError();
}

hipError_t launch_permute_rows_f32(void *dst, const void *src, const int32_t *n2o, int64_t n, int K, int to_device,
                                   hipStream_t s)
{
    int64_t blocks = (n * K + BLOCK - 1) / BLOCK;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_permute_rows_f32, dim3((unsigned)blocks), dim3(BLOCK), 0, s, dst, src, n2o, n, K, to_device);
    return hipGetLastError();
}

hipError_t launch_halo_map_f32(float *buf, float *h, float *ssh, float *u, const uint32_t *map, int64_t n, int unpack,
                               hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_halo_map_f32, dim3((unsigned)blocks), dim3(BLOCK), 0, s, buf, h, ssh, u, map, n, unpack);
    return hipGetLastError();
}

hipError_t launch_copy(double *dst, const double *src, int64_t n, hipStream_t s)
{
    int64_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_copy, dim3((unsigned)blocks), dim3(BLOCK), 0, s, dst, src, n);
    return hipGetLastError();
}

}  // namespace moka
