// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of libmoka_hip.
//
// Execution shape ("column kernels").  Fields are (nVertLevels, n) with the level index fastest,
// exactly the reference layout (PrognosticVars.jl:11-17).  A group of LPC lanes (LPC = smallest
// power of two >= nVertLevels, capped at 64: one whole wavefront for the 60/80-layer configs)
// owns one mesh entity at a time; lane l of the group owns levels l, l+LPC, ...  Every neighbour
// gather is therefore one contiguous K*8-byte row read, and with LPC = 64 all connectivity of the
// entity is wave-uniform (scalar loads).  One workgroup (256 threads) walks one *patch*: P
// consecutive cells of the RCB/RCM ordering plus the edges and vertices those cells own, so the
// rows a workgroup gathers are the rows its neighbours in the grid also touch; the blockIdx ->
// patch map keeps consecutive patches on one XCD (blocks b, b+8, ... share an XCD's L2).
//
// Arithmetic.  Compiled with -ffp-contract=off and written in the reference's operand order
// (each expression cites the reference line), so results are bit-identical to the CPU oracle.
// Memory-bound indirect stencil: no MFMA by design.
#include "kernels_common.hpp"

namespace moka {

// experiment (MOKA_DBG 16 / 32): 16 = identity map (consecutive patches on different XCDs); 32 = tiles of 64
// consecutive patches per XCD, tiles dealt round-robin, so the 8 XCDs sweep memory together
__device__ __forceinline__ int patch_of_block_dbg(int nPatches, int dbg)
{
    if (dbg & 16) return (int)blockIdx.x;
    if (dbg & 32) {
        const int x = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
        return ((j >> 6) * 8 + x) * 64 + (j & 63);
    }
    return patch_of_block(nPatches);
}

// ------------------------------------------------------------------------------------------------
// Fused tendency / RK-stage kernel.
//   cells : hEdge (K5, Operators.jl:217) -> thicknessFlux (K7, DiagnosticVars.jl:165)
//           -> flux divergence (K8, horizontal_advection.jl:60-66) [-> state update, ssh (K14)]
//   edges : -g grad ssh (K9, pressure_gradient.jl:58-64) + Coriolis (K10,
//           horizontal_advection_and_coriolis.jl:61-73)          [-> state update]
// The stage update is the RK4 specification of time_integration.jl:112-137:
//   Provis' = Curr + a*tend ; New = New + b*tend ; ssh from layerThickness.
// ------------------------------------------------------------------------------------------------
template <int LPC, int ME, int ME2>
__global__ __launch_bounds__(BLOCK) void k_stage(const MeshDev m, const StageArgs a)
{
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NG = BLOCK / LPC;
    const int grp = uniform_if_wave<LPC>(threadIdx.x / LPC);
    const int l = threadIdx.x % LPC;
    const int K = m.K;
    const int Kc = ((K + LPC - 1) / LPC) * LPC;

    // ---------------- cells ----------------
    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
    for (int c = c0 + grp; c < c1; c += NG) {
        CP<int32_t> re = cptr(m.eoc) + (size_t)c * ME;
        CP<int32_t> rc = cptr(m.coc) + (size_t)c * ME;
        CP<int32_t> rm = cptr(m.mltc) + (size_t)c * ME;
        CP<double> rs = cptr(m.sdv) + (size_t)c * ME;
        const double invA = cptr(m.invArea)[c];
        int ei[ME], ci[ME], mi[ME];
        double si[ME];
#pragma unroll
        for (int i = 0; i < ME; ++i) {
            ei[i] = re[i];
            ci[i] = rc[i];
            mi[i] = rm[i];
            si[i] = rs[i];
        }
        double sshAcc = 0.0;
        bool first = true;
        for (int k = l; k < Kc; k += LPC) {
            const bool act = k < K;
            const size_t off = (size_t)c * K + k;
            double hc = 0.0, uv[ME], hv[ME];
            if (act) {
                hc = a.ph[off];
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    const int es = ei[i] >= 0 ? ei[i] : ei[0];
                    const int cs = ci[i] >= 0 ? ci[i] : c;
                    uv[i] = a.pu[(size_t)es * K + k];
                    hv[i] = a.ph[(size_t)cs * K + k];
                }
            }
            double t = 0.0;
            if (act) {
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    if (ei[i] >= 0 && k < mi[i]) {
                        const double hE = 0.5 * (hc + hv[i]);      // Operators.jl:217
                        const double F = uv[i] * hE;               // DiagnosticVars.jl:165
                        t += F * si[i] * invA;                     // horizontal_advection.jl:63-64
                    }
                }
            }
            double hs = 0.0;   // the thickness whose column sum gives ssh_out
            if (act) {
                if (a.tendH) a.tendH[off] = t;
                const double hcur = a.ch ? a.ch[off] : hc;
                if (a.ph_out) {
                    const double hp = hcur + a.a * t;              // time_integration.jl:125
                    a.ph_out[off] = hp;
                    hs = hp;
                }
                if (a.nh_out) {
                    const double hn = (a.nh_in ? a.nh_in[off] : hcur) + a.b * t;   // :135
                    a.nh_out[off] = hn;
                    if (!a.ph_out) hs = hn;
                }
            }
            sshAcc = first ? hs : sshAcc + hs;                     // oracle_ksum strided partials
            first = false;
        }
        if (a.ssh_out) {
            const double s = group_sum<LPC>(sshAcc);
            if (l == 0) a.ssh_out[c] = s - cptr(m.rsum)[c];        // time_integration.jl:209 (+N3)
        }
    }

    // ---------------- edges ----------------
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
    for (int e = e0 + grp; e < e1; e += NG) {
        CP<int32_t> rh = cptr(m.ehdr) + (size_t)e * 4;
        const int4 hdr = make_int4(rh[0], rh[1], rh[2], rh[3]);
        CP<int32_t> re = cptr(m.eoe) + (size_t)e * ME2;
        CP<double> rw = cptr(m.woe) + (size_t)e * ME2;
        int xi[ME2];
        double wi[ME2], fi[ME2];
#pragma unroll
        for (int i = 0; i < ME2; ++i) {
            xi[i] = re[i];
            wi[i] = rw[i];
        }
#pragma unroll
        for (int i = 0; i < ME2; ++i) fi[i] = cptr(m.fEdge)[xi[i] >= 0 ? xi[i] : e];
        const double g = cptr(m.gInvDc)[e];
        // ssh was written by the previous launch and is never written by this one (ssh_out is another buffer)
        const double ds = cptr(a.ssh)[hdr.y] - cptr(a.ssh)[hdr.x];   // ssh[c2] - ssh[c1]
        const int mlt = hdr.w;
        for (int k = l; k < K; k += LPC) {
            const size_t off = (size_t)e * K + k;
            double uv[ME2];
#pragma unroll
            for (int i = 0; i < ME2; ++i) uv[i] = a.pu[(size_t)(xi[i] >= 0 ? xi[i] : e) * K + k];
            double t = 0.0;
            if (k < mlt) {
                t -= g * ds;                                       // pressure_gradient.jl:63
#pragma unroll
                for (int i = 0; i < ME2; ++i)
                    if (xi[i] >= 0) t += wi[i] * uv[i] * fi[i];    // ...coriolis.jl:70-72
            }
            if (a.tendU) a.tendU[off] = t;
            const double ucur = a.cu ? a.cu[off] : a.pu[off];
            if (a.pu_out) a.pu_out[off] = ucur + a.a * t;          // time_integration.jl:124
            if (a.nu_out) a.nu_out[off] = (a.nu_in ? a.nu_in[off] : ucur) + a.b * t;   // :134
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Column kernel, instruction-lean form for LPC = 64 (one wavefront per entity, lane = level).
//
// rocprof on the generic k_stage showed the SIMDs busy *issuing* ~150 VALU + ~150 SALU per entity
// (64-bit index*K*8 address arithmetic, selects, SGPR spills) for ~40 essential fp64 operations,
// and the per-CU scalar unit saturated.  Here the plan stores 32-bit BYTE offsets of every
// neighbour row (cRec / eRec), all of an entity's connectivity arrives in SGPRs with two or three
// s_load_dwordx8/x16, and every gather is `buffer_load_dwordx2 v, v_lane8, s[rsrc], s_off offen`:
// no address arithmetic at all.  Weights, fEdge and metric factors are SGPR operands of the fp64
// instructions.  Slot validity and "all levels active" are wave-uniform (scalar branches).
// Lanes >= K read past the row (the buffer range check returns 0 past the array) and never store.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double bload(rsrc_t r, int voff, uint32_t soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, (int)soff, 0));
}
__device__ __forceinline__ void bstore(rsrc_t r, int voff, uint32_t soff, double x)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r, voff, (int)soff, 0);
}

// own-row access: base pointer (SGPR pair) + 32-bit byte offset (VGPR) -> global_load/store saddr form
__device__ __forceinline__ double gload(const double *base, uint32_t off)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + off);
}
__device__ __forceinline__ void gstore(double *base, uint32_t off, double x)
{
    *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + off) = x;
}

template <int ME, int ME2>
__global__ __launch_bounds__(BLOCK) void k_stage_col(const ColMesh m, const StageArgs a)
{
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    const int K = m.K;
    const uint32_t rowB = (uint32_t)K * 8u;
    const rsrc_t ph = make_rsrc(a.ph, (uint32_t)m.nC * rowB), pu = make_rsrc(a.pu, (uint32_t)m.nE * rowB);

    // ---------------- cells ----------------
    {
        const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
        for (int c = c0 + wave; c < c1; c += BLOCK / 64) {
            CP<uint32_t> r = cptr(m.cRec) + (size_t)c * m.CI;
            CP<double> rs = cptr(m.sdv) + (size_t)c * ME;
            uint32_t eo[ME], co[ME];
            double sd[ME];
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                eo[i] = r[i];
                co[i] = r[ME + i];
                sd[i] = rs[i];
            }
            if (a.dbg & 2) {
#pragma unroll
                for (int i = 0; i < ME; ++i) { eo[i] = (uint32_t)c * rowB; co[i] = (uint32_t)c * rowB; }
            }
            const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
            const double invA = cptr(m.invArea)[c];
            const uint32_t own = (uint32_t)c * rowB;
            double sshAcc = 0.0;
            for (int kb = 0; kb < K; kb += 64) {
                const int k = kb + l, voff = k * 8;
                const uint32_t ooff = own + (uint32_t)voff;
                const double hc = bload(ph, voff, own);
                double uv[ME], hv[ME];
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    uv[i] = bload(pu, voff, eo[i]);
                    hv[i] = bload(ph, voff, co[i]);
                }
                double t = 0.0;
                if (all) {
#pragma unroll
                    for (int i = 0; i < ME; ++i)
                        if ((mask >> i) & 1u) t += uv[i] * (0.5 * (hc + hv[i])) * sd[i] * invA;   // Operators.jl:217,
                } else {                                                                          // DiagnosticVars.jl:165,
#pragma unroll
                    for (int i = 0; i < ME; ++i)                                                   // horizontal_advection.jl:63
                        if (((mask >> i) & 1u) && k < cptr(m.mltc)[(size_t)c * ME + i])
                            t += uv[i] * (0.5 * (hc + hv[i])) * sd[i] * invA;
                }
                double hs = 0.0;
                if (k < K && !((a.dbg & 1) && t != 12345.678)) {
                    if (a.tendH) gstore(a.tendH, ooff, t);
                    double hcur = hc;
                    if (a.ch) hcur = gload(a.ch, ooff);
                    if (a.ph_out) {
                        hs = hcur + a.a * t;                           // time_integration.jl:125
                        gstore(a.ph_out, ooff, hs);
                    }
                    if (a.nh_out) {
                        double nb = hcur;
                        if (a.nh_in) nb = gload(a.nh_in, ooff);
                        const double hn = nb + a.b * t;                // :135
                        gstore(a.nh_out, ooff, hn);
                        if (!a.ph_out) hs = hn;
                    }
                }
                if (kb == 0) sshAcc = hs;
                else sshAcc = sshAcc + hs;
            }
            if (a.ssh_out) {
                const double sum = group_sum<64>(sshAcc);
                if (l == 0) a.ssh_out[c] = sum - cptr(m.rsum)[c];      // time_integration.jl:209 (+N3)
            }
        }
    }

    // ---------------- edges ----------------
    {
        const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
        for (int e = e0 + wave; e < e1; e += BLOCK / 64) {
            const size_t er = (a.dbg & 4) ? (size_t)(e & 3) : (size_t)e;
            CP<uint32_t> r = cptr(m.eRec) + er * m.EI;
            CP<double> rw = cptr(m.woe) + er * ME2;
            CP<double> rf = cptr(m.feoe) + er * ME2;
            uint32_t xo[ME2];
            double wi[ME2], fi[ME2];
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                xo[i] = r[i];
                wi[i] = rw[i];
                fi[i] = rf[i];
            }
            if (a.dbg & 2) {
#pragma unroll
                for (int i = 0; i < ME2; ++i) xo[i] = (uint32_t)e * rowB;
            }
            const uint32_t cA = r[ME2], cB = r[ME2 + 1], mask = r[ME2 + 2];
            const int mlt = (int)r[ME2 + 3];
            const double g = cptr(m.gInvDc)[e];
            const double ds = cptr(a.ssh)[cB] - cptr(a.ssh)[cA];       // ssh[c2] - ssh[c1]
            const uint32_t own = (uint32_t)e * rowB;
            for (int kb = 0; kb < K; kb += 64) {
                const int k = kb + l, voff = k * 8;
                const uint32_t ooff = own + (uint32_t)voff;
                double uv[ME2];
#pragma unroll
                for (int i = 0; i < ME2; ++i) uv[i] = bload(pu, voff, xo[i]);
                double t = 0.0;
                if (k < mlt) {
                    t -= g * ds;                                       // pressure_gradient.jl:63
#pragma unroll
                    for (int i = 0; i < ME2; ++i)
                        if ((mask >> i) & 1u) t += wi[i] * uv[i] * fi[i];   // ...coriolis.jl:70-72
                }
                if (k < K && !((a.dbg & 1) && t != 12345.678)) {
                    if (a.tendU) gstore(a.tendU, ooff, t);
                    const double ucur = gload(a.cu ? a.cu : a.pu, ooff);
                    if (a.pu_out) gstore(a.pu_out, ooff, ucur + a.a * t);   // time_integration.jl:124
                    if (a.nu_out) {
                        double nb = ucur;
                        if (a.nu_in) nb = gload(a.nu_in, ooff);
                        gstore(a.nu_out, ooff, nb + a.b * t);          // :134
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined column kernel (K <= 64, one sweep): the memory latency of a task was paid
// twice per entity (scalar record, then the row gathers) with nothing else of that wave in flight,
// so ~70 % of wave time was parked on s_waitcnt.  Here each wave keeps TWO entities in flight:
// the gathers of entity t+1 are issued (into the other register set) before entity t is computed,
// so the compiler's counted `s_waitcnt vmcnt(N)` leaves the younger batch outstanding.  The issue
// is unconditional (the entity index is clamped) so that every compute is preceded by exactly one
// batch: a conditional issue would force vmcnt(0) at the join.  MODE fixes which own rows are read:
//   0 tendency only | 1 RK stage 1 (Curr == Provis) | 2 RK stage 2,3 | 3 RK stage 4 (New only)
// ------------------------------------------------------------------------------------------------
template <int ME, int MODE>
struct CellBatch {
    double hc, uv[ME], hv[ME], cur, nin;
};
template <int ME2, int MODE>
struct EdgeBatch {
    double uv[ME2], own, cur, nin;
};

template <int ME, int MODE>
__device__ __forceinline__ void cell_issue(CellBatch<ME, MODE> &b, const ColMesh &m, const StageArgs &a, rsrc_t ph,
                                           rsrc_t pu, int c, uint32_t rowB, int voff)
{
    CP<uint32_t> r = cptr(m.cRec) + (size_t)c * m.CI;
    const uint32_t own = (uint32_t)c * rowB;
    b.hc = bload(ph, voff, own);
#pragma unroll
    for (int i = 0; i < ME; ++i) {
        b.uv[i] = bload(pu, voff, r[i]);
        b.hv[i] = bload(ph, voff, r[ME + i]);
    }
    if constexpr (MODE == 2) b.cur = gload(a.ch, own + (uint32_t)voff);
    if constexpr (MODE >= 2) b.nin = gload(a.nh_in, own + (uint32_t)voff);
}

template <int ME, int MODE>
__device__ __forceinline__ void cell_finish(const CellBatch<ME, MODE> &b, const ColMesh &m, const StageArgs &a, int c,
                                            uint32_t rowB, int voff, int l, int K)
{
    CP<uint32_t> r = cptr(m.cRec) + (size_t)c * m.CI;
    CP<double> rs = cptr(m.sdv) + (size_t)c * ME;
    const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
    const double invA = cptr(m.invArea)[c];
    const uint32_t ooff = (uint32_t)c * rowB + (uint32_t)voff;
    double t = 0.0;
    if (all) {
#pragma unroll
        for (int i = 0; i < ME; ++i)
            if ((mask >> i) & 1u) t += b.uv[i] * (0.5 * (b.hc + b.hv[i])) * rs[i] * invA;   // Operators.jl:217,
    } else {                                                                                // DiagnosticVars.jl:165,
#pragma unroll
        for (int i = 0; i < ME; ++i)                                                         // horizontal_advection.jl:63
            if (((mask >> i) & 1u) && l < cptr(m.mltc)[(size_t)c * ME + i]) t += b.uv[i] * (0.5 * (b.hc + b.hv[i])) * rs[i] * invA;
    }
    double hs = 0.0;
    if (l < K) {
        if constexpr (MODE == 0) gstore(a.tendH, ooff, t);
        if constexpr (MODE == 1 || MODE == 2) {
            const double hcur = MODE == 2 ? b.cur : b.hc;
            hs = hcur + a.a * t;                                       // time_integration.jl:125
            gstore(a.ph_out, ooff, hs);
            gstore(a.nh_out, ooff, (MODE == 2 ? b.nin : hcur) + a.b * t);   // :135
        }
        if constexpr (MODE == 3) {
            hs = b.nin + a.b * t;
            gstore(a.nh_out, ooff, hs);
        }
    }
    if constexpr (MODE != 0) {
        const double sum = group_sum<64>(hs);
        if (l == 0) a.ssh_out[c] = sum - cptr(m.rsum)[c];              // time_integration.jl:209 (+N3)
    }
}

template <int ME2, int MODE>
__device__ __forceinline__ void edge_issue(EdgeBatch<ME2, MODE> &b, const ColMesh &m, const StageArgs &a, rsrc_t pu,
                                           int e, uint32_t rowB, int voff)
{
    CP<uint32_t> r = cptr(m.eRec) + (size_t)e * m.EI;
    const uint32_t own = (uint32_t)e * rowB;
#pragma unroll
    for (int i = 0; i < ME2; ++i) b.uv[i] = bload(pu, voff, r[i]);
    if constexpr (MODE == 1) b.own = bload(pu, voff, own);
    if constexpr (MODE == 2) b.cur = gload(a.cu, own + (uint32_t)voff);
    if constexpr (MODE >= 2) b.nin = gload(a.nu_in, own + (uint32_t)voff);
}

template <int ME2, int MODE>
__device__ __forceinline__ void edge_finish(const EdgeBatch<ME2, MODE> &b, const ColMesh &m, const StageArgs &a, int e,
                                            uint32_t rowB, int voff, int l, int K)
{
    CP<uint32_t> r = cptr(m.eRec) + (size_t)e * m.EI;
    CP<double> rw = cptr(m.woe) + (size_t)e * ME2;
    CP<double> rf = cptr(m.feoe) + (size_t)e * ME2;
    const uint32_t cA = r[ME2], cB = r[ME2 + 1], mask = r[ME2 + 2];
    const int mlt = (int)r[ME2 + 3];
    const double g = cptr(m.gInvDc)[e];
    const double ds = cptr(a.ssh)[cB] - cptr(a.ssh)[cA];               // ssh[c2] - ssh[c1]
    const uint32_t ooff = (uint32_t)e * rowB + (uint32_t)voff;
    double t = 0.0;
    if (l < mlt) {
        t -= g * ds;                                                   // pressure_gradient.jl:63
#pragma unroll
        for (int i = 0; i < ME2; ++i)
            if ((mask >> i) & 1u) t += rw[i] * b.uv[i] * rf[i];        // ...coriolis.jl:70-72
    }
    if (l < K) {
        if constexpr (MODE == 0) gstore(a.tendU, ooff, t);
        if constexpr (MODE == 1) {
            gstore(a.pu_out, ooff, b.own + a.a * t);                   // time_integration.jl:124
            gstore(a.nu_out, ooff, b.own + a.b * t);                   // :134
        }
        if constexpr (MODE == 2) {
            gstore(a.pu_out, ooff, b.cur + a.a * t);
            gstore(a.nu_out, ooff, b.nin + a.b * t);
        }
        if constexpr (MODE == 3) gstore(a.nu_out, ooff, b.nin + a.b * t);
    }
}

template <int ME, int ME2, int MODE>
__global__ __launch_bounds__(BLOCK) void k_stage_colp(const ColMesh m, const StageArgs a)
{
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NW = BLOCK / 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    const int K = m.K, voff = l * 8;
    const uint32_t rowB = (uint32_t)K * 8u;
    const rsrc_t ph = make_rsrc(a.ph, (uint32_t)m.nC * rowB), pu = make_rsrc(a.pu, (uint32_t)m.nE * rowB);
    {
        const int c0 = cptr(m.patchCellStart)[p] + wave, c1 = cptr(m.patchCellStart)[p + 1];
        const int n = c1 > c0 ? (c1 - c0 + NW - 1) / NW : 0;           // tasks of this wave: c0, c0+NW, ...
        if (n > 0) {
            CellBatch<ME, MODE> A, B;
            cell_issue<ME, MODE>(A, m, a, ph, pu, c0, rowB, voff);
            for (int t = 0;;) {
                cell_issue<ME, MODE>(B, m, a, ph, pu, c0 + NW * (t + 1 < n ? t + 1 : n - 1), rowB, voff);
                cell_finish<ME, MODE>(A, m, a, c0 + NW * t, rowB, voff, l, K);
                if (++t >= n) break;
                cell_issue<ME, MODE>(A, m, a, ph, pu, c0 + NW * (t + 1 < n ? t + 1 : n - 1), rowB, voff);
                cell_finish<ME, MODE>(B, m, a, c0 + NW * t, rowB, voff, l, K);
                if (++t >= n) break;
            }
        }
    }
    {
        const int e0 = cptr(m.patchEdgeStart)[p] + wave, e1 = cptr(m.patchEdgeStart)[p + 1];
        const int n = e1 > e0 ? (e1 - e0 + NW - 1) / NW : 0;
        if (n > 0) {
            EdgeBatch<ME2, MODE> A, B;
            edge_issue<ME2, MODE>(A, m, a, pu, e0, rowB, voff);
            for (int t = 0;;) {
                edge_issue<ME2, MODE>(B, m, a, pu, e0 + NW * (t + 1 < n ? t + 1 : n - 1), rowB, voff);
                edge_finish<ME2, MODE>(A, m, a, e0 + NW * t, rowB, voff, l, K);
                if (++t >= n) break;
                edge_issue<ME2, MODE>(A, m, a, pu, e0 + NW * (t + 1 < n ? t + 1 : n - 1), rowB, voff);
                edge_finish<ME2, MODE>(B, m, a, e0 + NW * t, rowB, voff, l, K);
                if (++t >= n) break;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Column kernel with 16-byte lanes (K even, K <= 64): rocprof's TCP_TOTAL_CACHE_ACCESSES showed the
// vector L1 serving the 8-byte-per-lane row gathers at ~32 B per access (17-18 accesses per 480-byte
// row), i.e. the L1, not HBM, paced the gathers.  Here lanes 0..K/2-1 of the wavefront each own two
// consecutive levels and read 16 bytes (buffer_load_dwordx4): half the L1 accesses per row.  The
// upper lanes are masked off; fp64 instruction count doubles (two components) but it is small.
// Pipelined like k_stage_colp (two entities in flight per wave).
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 bload2(rsrc_t r, int voff, uint32_t soff)
{
    return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, (int)soff, 0));
}
__device__ __forceinline__ double2 gload2(const double *base, uint32_t off)
{
    return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + off);
}
__device__ __forceinline__ void gstore2(double *base, uint32_t off, double2 x)
{
    *reinterpret_cast<double2 *>(reinterpret_cast<char *>(base) + off) = x;
}

template <int ME, int MODE>
struct CellBatch2 {
    double2 hc, uv[ME], hv[ME], cur, nin;
};
template <int ME2, int MODE>
struct EdgeBatch2 {
    double2 uv[ME2], own, cur, nin;
};

template <int ME, int MODE>
__device__ __forceinline__ void cell_issue2(CellBatch2<ME, MODE> &b, const ColMesh &m, const StageArgs &a, rsrc_t ph,
                                            rsrc_t pu, int c, uint32_t rowB, int voff)
{
    CP<uint32_t> r = cptr(m.cRec) + (size_t)c * m.CI;
    const uint32_t own = (uint32_t)c * rowB;
    b.hc = bload2(ph, voff, own);
#pragma unroll
    for (int i = 0; i < ME; ++i) {
        b.uv[i] = bload2(pu, voff, r[i]);
        b.hv[i] = bload2(ph, voff, r[ME + i]);
    }
    if constexpr (MODE == 2) b.cur = gload2(a.ch, own + (uint32_t)voff);
    if constexpr (MODE >= 2) b.nin = gload2(a.nh_in, own + (uint32_t)voff);
}

template <int ME, int MODE>
__device__ __forceinline__ void cell_finish2(const CellBatch2<ME, MODE> &b, const ColMesh &m, const StageArgs &a, int c,
                                             uint32_t rowB, int voff, int l, int K)
{
    CP<uint32_t> r = cptr(m.cRec) + (size_t)c * m.CI;
    CP<double> rs = cptr(m.sdv) + (size_t)c * ME;
    const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
    const double invA = cptr(m.invArea)[c];
    const uint32_t ooff = (uint32_t)c * rowB + (uint32_t)voff;
    const int k0 = 2 * l;
    double2 t = make_double2(0.0, 0.0);
    if (all) {
#pragma unroll
        for (int i = 0; i < ME; ++i)
            if ((mask >> i) & 1u) {
                t.x += b.uv[i].x * (0.5 * (b.hc.x + b.hv[i].x)) * rs[i] * invA;   // Operators.jl:217, DiagnosticVars.jl:165,
                t.y += b.uv[i].y * (0.5 * (b.hc.y + b.hv[i].y)) * rs[i] * invA;   // horizontal_advection.jl:63
            }
    } else {
#pragma unroll
        for (int i = 0; i < ME; ++i)
            if ((mask >> i) & 1u) {
                const int ml = cptr(m.mltc)[(size_t)c * ME + i];
                if (k0 < ml) t.x += b.uv[i].x * (0.5 * (b.hc.x + b.hv[i].x)) * rs[i] * invA;
                if (k0 + 1 < ml) t.y += b.uv[i].y * (0.5 * (b.hc.y + b.hv[i].y)) * rs[i] * invA;
            }
    }
    double2 hs = make_double2(0.0, 0.0);
    if (k0 < K) {
        if constexpr (MODE == 0) gstore2(a.tendH, ooff, t);
        if constexpr (MODE == 1 || MODE == 2) {
            const double2 hcur = MODE == 2 ? b.cur : b.hc;
            const double2 nb = MODE == 2 ? b.nin : hcur;
            hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);                    // time_integration.jl:125
            gstore2(a.ph_out, ooff, hs);
            gstore2(a.nh_out, ooff, make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y));    // :135
        }
        if constexpr (MODE == 3) {
            hs = make_double2(b.nin.x + a.b * t.x, b.nin.y + a.b * t.y);
            gstore2(a.nh_out, ooff, hs);
        }
    }
    if constexpr (MODE != 0) {
        // oracle_ksum order: lane-xor 16..1 on (even, odd) levels == level-xor 32..2, then level-xor 1
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) {
            const double ox = __shfl_xor(hs.x, sft, 64), oy = __shfl_xor(hs.y, sft, 64);
            hs = make_double2(hs.x + ox, hs.y + oy);
        }
        if (l == 0) a.ssh_out[c] = (hs.x + hs.y) - cptr(m.rsum)[c];                       // :209 (+N3)
    }
}

template <int ME2, int MODE>
__device__ __forceinline__ void edge_issue2(EdgeBatch2<ME2, MODE> &b, const ColMesh &m, const StageArgs &a, rsrc_t pu,
                                            int e, uint32_t rowB, int voff)
{
    CP<uint32_t> r = cptr(m.eRec) + (size_t)e * m.EI;
    const uint32_t own = (uint32_t)e * rowB;
#pragma unroll
    for (int i = 0; i < ME2; ++i) b.uv[i] = bload2(pu, voff, r[i]);
    if constexpr (MODE == 1) b.own = bload2(pu, voff, own);
    if constexpr (MODE == 2) b.cur = gload2(a.cu, own + (uint32_t)voff);
    if constexpr (MODE >= 2) b.nin = gload2(a.nu_in, own + (uint32_t)voff);
}

template <int ME2, int MODE>
__device__ __forceinline__ void edge_finish2(const EdgeBatch2<ME2, MODE> &b, const ColMesh &m, const StageArgs &a, int e,
                                             uint32_t rowB, int voff, int l, int K)
{
    CP<uint32_t> r = cptr(m.eRec) + (size_t)e * m.EI;
    CP<double> rw = cptr(m.woe) + (size_t)e * ME2;
    CP<double> rf = cptr(m.feoe) + (size_t)e * ME2;
    const uint32_t cA = r[ME2], cB = r[ME2 + 1], mask = r[ME2 + 2];
    const int mlt = (int)r[ME2 + 3];
    const double g = cptr(m.gInvDc)[e];
    const double ds = cptr(a.ssh)[cB] - cptr(a.ssh)[cA];               // ssh[c2] - ssh[c1]
    const uint32_t ooff = (uint32_t)e * rowB + (uint32_t)voff;
    const int k0 = 2 * l;
    double2 t = make_double2(0.0, 0.0);
    if (k0 < mlt) {
        t.x -= g * ds;                                                 // pressure_gradient.jl:63
#pragma unroll
        for (int i = 0; i < ME2; ++i)
            if ((mask >> i) & 1u) t.x += rw[i] * b.uv[i].x * rf[i];    // ...coriolis.jl:70-72
    }
    if (k0 + 1 < mlt) {
        t.y -= g * ds;
#pragma unroll
        for (int i = 0; i < ME2; ++i)
            if ((mask >> i) & 1u) t.y += rw[i] * b.uv[i].y * rf[i];
    }
    if (k0 < K) {
        if constexpr (MODE == 0) gstore2(a.tendU, ooff, t);
        if constexpr (MODE == 1) {
            gstore2(a.pu_out, ooff, make_double2(b.own.x + a.a * t.x, b.own.y + a.a * t.y));   // time_integration.jl:124
            gstore2(a.nu_out, ooff, make_double2(b.own.x + a.b * t.x, b.own.y + a.b * t.y));   // :134
        }
        if constexpr (MODE == 2) {
            gstore2(a.pu_out, ooff, make_double2(b.cur.x + a.a * t.x, b.cur.y + a.a * t.y));
            gstore2(a.nu_out, ooff, make_double2(b.nin.x + a.b * t.x, b.nin.y + a.b * t.y));
        }
        if constexpr (MODE == 3) gstore2(a.nu_out, ooff, make_double2(b.nin.x + a.b * t.x, b.nin.y + a.b * t.y));
    }
}

template <int ME, int ME2, int MODE, bool PIPE>
__global__ __launch_bounds__(BLOCK) void k_stage_colx(const ColMesh m, const StageArgs a)
{
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NW = BLOCK / 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (l >= 32) return;                     // 16-byte lanes: the lower half-wave covers K <= 64 levels
    const int K = m.K, voff = l * 16;
    const uint32_t rowB = (uint32_t)K * 8u;
    const rsrc_t ph = make_rsrc(a.ph, (uint32_t)m.nC * rowB), pu = make_rsrc(a.pu, (uint32_t)m.nE * rowB);
    {
        const int c0 = cptr(m.patchCellStart)[p] + wave, c1 = cptr(m.patchCellStart)[p + 1];
        const int n = c1 > c0 ? (c1 - c0 + NW - 1) / NW : 0;
        if (n > 0) {
            if constexpr (PIPE) {
                CellBatch2<ME, MODE> A, B;
                cell_issue2<ME, MODE>(A, m, a, ph, pu, c0, rowB, voff);
                for (int t = 0;;) {
                    cell_issue2<ME, MODE>(B, m, a, ph, pu, c0 + NW * (t + 1 < n ? t + 1 : n - 1), rowB, voff);
                    cell_finish2<ME, MODE>(A, m, a, c0 + NW * t, rowB, voff, l, K);
                    if (++t >= n) break;
                    cell_issue2<ME, MODE>(A, m, a, ph, pu, c0 + NW * (t + 1 < n ? t + 1 : n - 1), rowB, voff);
                    cell_finish2<ME, MODE>(B, m, a, c0 + NW * t, rowB, voff, l, K);
                    if (++t >= n) break;
                }
            } else {
                for (int t = 0; t < n; ++t) {
                    CellBatch2<ME, MODE> A;
                    cell_issue2<ME, MODE>(A, m, a, ph, pu, c0 + NW * t, rowB, voff);
                    cell_finish2<ME, MODE>(A, m, a, c0 + NW * t, rowB, voff, l, K);
                }
            }
        }
    }
    {
        const int e0 = cptr(m.patchEdgeStart)[p] + wave, e1 = cptr(m.patchEdgeStart)[p + 1];
        const int n = e1 > e0 ? (e1 - e0 + NW - 1) / NW : 0;
        if (n > 0) {
            if constexpr (PIPE) {
                EdgeBatch2<ME2, MODE> A, B;
                edge_issue2<ME2, MODE>(A, m, a, pu, e0, rowB, voff);
                for (int t = 0;;) {
                    edge_issue2<ME2, MODE>(B, m, a, pu, e0 + NW * (t + 1 < n ? t + 1 : n - 1), rowB, voff);
                    edge_finish2<ME2, MODE>(A, m, a, e0 + NW * t, rowB, voff, l, K);
                    if (++t >= n) break;
                    edge_issue2<ME2, MODE>(A, m, a, pu, e0 + NW * (t + 1 < n ? t + 1 : n - 1), rowB, voff);
                    edge_finish2<ME2, MODE>(B, m, a, e0 + NW * t, rowB, voff, l, K);
                    if (++t >= n) break;
                }
            } else {
                for (int t = 0; t < n; ++t) {
                    EdgeBatch2<ME2, MODE> A;
                    edge_issue2<ME2, MODE>(A, m, a, pu, e0 + NW * t, rowB, voff);
                    edge_finish2<ME2, MODE>(A, m, a, e0 + NW * t, rowB, voff, l, K);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Record-staged column kernel ("rec"): the default for 33 <= K <= 64.
//
// Ablation (MOKA_DBG on the plain column kernel, profiles/r01_ablation.txt) showed that neither the
// neighbour gathers nor the stores set the time: with every gather redirected to the entity's own
// (L1-hot) row the kernel was just as slow.  What each wave waited for, once per entity and with
// nothing else of its own in flight, was the *scalar load of the entity's connectivity record*: a
// cold, never-reused stream (~840 MB per evaluation) that misses the scalar cache and pays a full
// HBM round trip (~2.8 us per entity per wave).
// Here a workgroup first copies the records of its whole patch (contiguous ranges of eRec / woe /
// feoe / gInvDc / cRec / sdv / invArea / rsum: ~27 KB for 32 cells) into LDS with coalesced vector
// loads -- one round trip per patch instead of one per entity -- and each wave then reads its
// entity's offsets and weights from LDS (broadcast reads).  Row gathers are software-pipelined two
// entities deep per wave (register sets A/B, counted vmcnt), ssh[c1], ssh[c2] ride in the same batch.
// ------------------------------------------------------------------------------------------------
struct RecLds {
    uint32_t *eRec, *cRec;
    double *woe, *feoe, *g, *sdv, *invA, *rsum;
};

__device__ __forceinline__ RecLds rec_carve(unsigned char *smem, const ColMesh &m, int ME, int ME2, int maxOwnE, int maxOwnC)
{
    RecLds L;
    L.woe = reinterpret_cast<double *>(smem);
    L.feoe = L.woe + (size_t)maxOwnE * ME2;
    L.g = L.feoe + (size_t)maxOwnE * ME2;
    L.sdv = L.g + maxOwnE;
    L.invA = L.sdv + (size_t)maxOwnC * ME;
    L.rsum = L.invA + maxOwnC;
    L.eRec = reinterpret_cast<uint32_t *>(L.rsum + maxOwnC);
    L.cRec = L.eRec + (size_t)maxOwnE * m.EI;
    return L;
}

template <int ME, int MODE>
struct RCell {
    double hc, uv[ME], hv[ME], cur, nin;
};
template <int ME2, int MODE>
struct REdge {
    double uv[ME2], sA, sB, own, cur, nin;
};

template <int ME, int MODE>
__device__ __forceinline__ void rcell_issue(RCell<ME, MODE> &b, const RecLds &L, const ColMesh &m, const StageArgs &a,
                                            int ci, int c, uint32_t rowB, uint32_t voff)
{
    const uint32_t *r = L.cRec + (size_t)ci * m.CI;
    const uint32_t own = (uint32_t)c * rowB + voff;
    b.hc = gload(a.ph, own);
    if (a.dbg & 8) {
#pragma unroll
        for (int i = 0; i < ME; ++i) {
            b.uv[i] = (i & 1) ? 1.0 : gload(a.pu, r[i] + voff);
            b.hv[i] = (i & 1) ? 1.0 : gload(a.ph, r[ME + i] + voff);
        }
    } else {
#pragma unroll
        for (int i = 0; i < ME; ++i) {
            b.uv[i] = gload(a.pu, r[i] + voff);
            b.hv[i] = gload(a.ph, r[ME + i] + voff);
        }
    }
    if constexpr (MODE == 2) b.cur = gload(a.ch, own);
    if constexpr (MODE >= 2) b.nin = gload(a.nh_in, own);
}

template <int ME, int MODE>
__device__ __forceinline__ void rcell_finish(const RCell<ME, MODE> &b, const RecLds &L, const ColMesh &m, const StageArgs &a,
                                             int ci, int c, uint32_t rowB, uint32_t voff, int l, int K)
{
    const uint32_t *r = L.cRec + (size_t)ci * m.CI;
    const double *rs = L.sdv + (size_t)ci * ME;
    const uint32_t mask = __builtin_amdgcn_readfirstlane(r[2 * ME]), all = __builtin_amdgcn_readfirstlane(r[2 * ME + 1]);
    const double invA = L.invA[ci];
    const uint32_t ooff = (uint32_t)c * rowB + voff;
    double t = 0.0;
    if (all) {
#pragma unroll
        for (int i = 0; i < ME; ++i)
            if ((mask >> i) & 1u) t += b.uv[i] * (0.5 * (b.hc + b.hv[i])) * rs[i] * invA;   // Operators.jl:217,
    } else {                                                                                // DiagnosticVars.jl:165,
#pragma unroll
        for (int i = 0; i < ME; ++i)                                                         // horizontal_advection.jl:63
            if (((mask >> i) & 1u) && l < cptr(m.mltc)[(size_t)c * ME + i]) t += b.uv[i] * (0.5 * (b.hc + b.hv[i])) * rs[i] * invA;
    }
    double hs = 0.0;
    if (l < K) {
        if constexpr (MODE == 0) gstore(a.tendH, ooff, t);
        if constexpr (MODE == 1 || MODE == 2) {
            const double hcur = MODE == 2 ? b.cur : b.hc;
            hs = hcur + a.a * t;                                       // time_integration.jl:125
            gstore(a.ph_out, ooff, hs);
            gstore(a.nh_out, ooff, (MODE == 2 ? b.nin : hcur) + a.b * t);   // :135
        }
        if constexpr (MODE == 3) {
            hs = b.nin + a.b * t;
            gstore(a.nh_out, ooff, hs);
        }
    }
    if constexpr (MODE != 0) {
        const double sum = group_sum<64>(hs);
        if (l == 0) a.ssh_out[c] = sum - L.rsum[ci];                   // time_integration.jl:209 (+N3)
    }
}

template <int ME2, int MODE>
__device__ __forceinline__ void redge_issue(REdge<ME2, MODE> &b, const RecLds &L, const ColMesh &m, const StageArgs &a,
                                            int ei, int e, uint32_t rowB, uint32_t voff)
{
    const uint32_t *r = L.eRec + (size_t)ei * m.EI;
    const uint32_t own = (uint32_t)e * rowB + voff;
    if (a.dbg & 8) {   // diagnostics: issue only half of the gathers (results are wrong, timing only)
#pragma unroll
        for (int i = 0; i < ME2; ++i) b.uv[i] = (i & 1) ? 1.0 : gload(a.pu, r[i] + voff);
    } else {
#pragma unroll
        for (int i = 0; i < ME2; ++i) b.uv[i] = gload(a.pu, r[i] + voff);
    }
    b.sA = a.ssh[r[ME2]];
    b.sB = a.ssh[r[ME2 + 1]];
    if constexpr (MODE == 1) b.own = gload(a.pu, own);
    if constexpr (MODE == 2) b.cur = gload(a.cu, own);
    if constexpr (MODE >= 2) b.nin = gload(a.nu_in, own);
}

template <int ME2, int MODE>
__device__ __forceinline__ void redge_finish(const REdge<ME2, MODE> &b, const RecLds &L, const ColMesh &m, const StageArgs &a,
                                             int ei, int e, uint32_t rowB, uint32_t voff, int l, int K)
{
    const uint32_t *r = L.eRec + (size_t)ei * m.EI;
    const double *rw = L.woe + (size_t)ei * ME2;
    const double *rf = L.feoe + (size_t)ei * ME2;
    const uint32_t mask = __builtin_amdgcn_readfirstlane(r[ME2 + 2]);
    const int mlt = (int)r[ME2 + 3];
    const double g = L.g[ei];
    const double ds = b.sB - b.sA;                                     // ssh[c2] - ssh[c1]
    const uint32_t ooff = (uint32_t)e * rowB + voff;
    double t = 0.0;
    if (l < mlt) {
        t -= g * ds;                                                   // pressure_gradient.jl:63
#pragma unroll
        for (int i = 0; i < ME2; ++i)
            if ((mask >> i) & 1u) t += rw[i] * b.uv[i] * rf[i];        // ...coriolis.jl:70-72
    }
    if (l < K) {
        if constexpr (MODE == 0) gstore(a.tendU, ooff, t);
        if constexpr (MODE == 1) {
            gstore(a.pu_out, ooff, b.own + a.a * t);                   // time_integration.jl:124
            gstore(a.nu_out, ooff, b.own + a.b * t);                   // :134
        }
        if constexpr (MODE == 2) {
            gstore(a.pu_out, ooff, b.cur + a.a * t);
            gstore(a.nu_out, ooff, b.nin + a.b * t);
        }
        if constexpr (MODE == 3) gstore(a.nu_out, ooff, b.nin + a.b * t);
    }
}

template <int ME, int ME2, int MODE>
__global__ __launch_bounds__(BLOCK) void k_stage_rec(const ColMesh m, const StageArgs a, int maxOwnE, int maxOwnC)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NW = BLOCK / 64;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int K = m.K;
    const uint32_t voff = (uint32_t)l * 8u, rowB = (uint32_t)K * 8u;
    const RecLds L = rec_carve(smem, m, ME, ME2, maxOwnE, maxOwnC);
    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
    const int nOwnC = c1 - c0, nOwnE = e1 - e0;

    // ---- 1. the patch's records: contiguous ranges -> coalesced copies, one round trip per patch ----
    for (int i = tid; i < nOwnE * m.EI; i += BLOCK) L.eRec[i] = m.eRec[(size_t)e0 * m.EI + i];
    for (int i = tid; i < nOwnE * ME2; i += BLOCK) {
        L.woe[i] = m.woe[(size_t)e0 * ME2 + i];
        L.feoe[i] = m.feoe[(size_t)e0 * ME2 + i];
    }
    for (int i = tid; i < nOwnE; i += BLOCK) L.g[i] = m.gInvDc[e0 + i];
    for (int i = tid; i < nOwnC * m.CI; i += BLOCK) L.cRec[i] = m.cRec[(size_t)c0 * m.CI + i];
    for (int i = tid; i < nOwnC * ME; i += BLOCK) L.sdv[i] = m.sdv[(size_t)c0 * ME + i];
    for (int i = tid; i < nOwnC; i += BLOCK) {
        L.invA[i] = m.invArea[c0 + i];
        L.rsum[i] = m.rsum[c0 + i];
    }
    __syncthreads();

    // ---- 2. cells, two in flight per wave ----
    {
        const int n = nOwnC > wave ? (nOwnC - wave + NW - 1) / NW : 0;   // local ids wave, wave+NW, ...
        if (n > 0) {
            RCell<ME, MODE> A, B;
            rcell_issue<ME, MODE>(A, L, m, a, wave, c0 + wave, rowB, voff);
            for (int t = 0;;) {
                int nx = wave + NW * (t + 1 < n ? t + 1 : n - 1);
                rcell_issue<ME, MODE>(B, L, m, a, nx, c0 + nx, rowB, voff);
                rcell_finish<ME, MODE>(A, L, m, a, wave + NW * t, c0 + wave + NW * t, rowB, voff, l, K);
                if (++t >= n) break;
                nx = wave + NW * (t + 1 < n ? t + 1 : n - 1);
                rcell_issue<ME, MODE>(A, L, m, a, nx, c0 + nx, rowB, voff);
                rcell_finish<ME, MODE>(B, L, m, a, wave + NW * t, c0 + wave + NW * t, rowB, voff, l, K);
                if (++t >= n) break;
            }
        }
    }
    // ---- 3. edges, two in flight per wave ----
    {
        const int n = nOwnE > wave ? (nOwnE - wave + NW - 1) / NW : 0;
        if (n > 0) {
            REdge<ME2, MODE> A, B;
            redge_issue<ME2, MODE>(A, L, m, a, wave, e0 + wave, rowB, voff);
            for (int t = 0;;) {
                int nx = wave + NW * (t + 1 < n ? t + 1 : n - 1);
                redge_issue<ME2, MODE>(B, L, m, a, nx, e0 + nx, rowB, voff);
                redge_finish<ME2, MODE>(A, L, m, a, wave + NW * t, e0 + wave + NW * t, rowB, voff, l, K);
                if (++t >= n) break;
                nx = wave + NW * (t + 1 < n ? t + 1 : n - 1);
                redge_issue<ME2, MODE>(A, L, m, a, nx, e0 + nx, rowB, voff);
                redge_finish<ME2, MODE>(B, L, m, a, wave + NW * t, e0 + wave + NW * t, rowB, voff, l, K);
                if (++t >= n) break;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Record-staged column kernel with 16-byte lanes and two entities per wavefront ("rec2").
//
// Measured on k_stage_rec (profiles/r01_ablation.txt): TA_BUSY 86 % of the kernel, and removing
// half of the row gathers removed 0.65 ms = 16 cycles per wave-level load per CU: the texture
// address path moves 4 lanes per cycle whatever their width, so an 8-byte-per-lane row read costs
// the same 16 cycles as a 16-byte-per-lane one.  Here each 32-lane half-wave owns one entity and
// each lane two consecutive levels (K even, K <= 64): every vector memory instruction moves two
// 480-byte rows (1 KiB) in those 16 cycles -- twice the bytes per TA cycle, for loads and stores.
// Records still come from LDS (per-half broadcast reads), gathers are pipelined two deep.
// ------------------------------------------------------------------------------------------------
template <int ME, int MODE>
struct R2Cell {
    double2 hc, uv[ME], hv[ME], cur, nin;
};
template <int ME2, int MODE>
struct R2Edge {
    double2 uv[ME2], own, cur, nin;
    double sA, sB;
};

template <int ME, int MODE>
__device__ __forceinline__ void r2cell_issue(R2Cell<ME, MODE> &b, const RecLds &L, const ColMesh &m, const StageArgs &a,
                                             int ci, int c, uint32_t rowB, uint32_t voff)
{
    const uint32_t *r = L.cRec + (size_t)ci * m.CI;
    const uint32_t own = (uint32_t)c * rowB + voff;
    b.hc = gload2(a.ph, own);
#pragma unroll
    for (int i = 0; i < ME; ++i) {
        b.uv[i] = gload2(a.pu, r[i] + voff);
        b.hv[i] = gload2(a.ph, r[ME + i] + voff);
    }
    if constexpr (MODE == 2) b.cur = gload2(a.ch, own);
    if constexpr (MODE >= 2) b.nin = gload2(a.nh_in, own);
}

template <int ME, int MODE>
__device__ __forceinline__ void r2cell_finish(const R2Cell<ME, MODE> &b, const RecLds &L, const ColMesh &m, const StageArgs &a,
                                              int ci, int c, uint32_t rowB, uint32_t voff, int l, int K, bool valid)
{
    const uint32_t *r = L.cRec + (size_t)ci * m.CI;
    const double *rs = L.sdv + (size_t)ci * ME;
    const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
    const double invA = L.invA[ci];
    const uint32_t ooff = (uint32_t)c * rowB + voff;
    const int k0 = 2 * l;
    double2 t = make_double2(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < ME; ++i) {
        const int ml = all ? K : cptr(m.mltc)[(size_t)c * ME + i];
        const bool on = (mask >> i) & 1u;
        const double dx = b.uv[i].x * (0.5 * (b.hc.x + b.hv[i].x)) * rs[i] * invA;   // Operators.jl:217, DiagnosticVars.jl:165,
        const double dy = b.uv[i].y * (0.5 * (b.hc.y + b.hv[i].y)) * rs[i] * invA;   // horizontal_advection.jl:63
        if (on && k0 < ml) t.x += dx;
        if (on && k0 + 1 < ml) t.y += dy;
    }
    double2 hs = make_double2(0.0, 0.0);
    if (valid && k0 < K) {
        if constexpr (MODE == 0) gstore2(a.tendH, ooff, t);
        if constexpr (MODE == 1 || MODE == 2) {
            const double2 hcur = MODE == 2 ? b.cur : b.hc;
            const double2 nb = MODE == 2 ? b.nin : hcur;
            hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);                    // time_integration.jl:125
            gstore2(a.ph_out, ooff, hs);
            gstore2(a.nh_out, ooff, make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y));    // :135
        }
        if constexpr (MODE == 3) {
            hs = make_double2(b.nin.x + a.b * t.x, b.nin.y + a.b * t.y);
            gstore2(a.nh_out, ooff, hs);
        }
    }
    if constexpr (MODE != 0) {
        // oracle_ksum order: lane-xor 16..1 on (even, odd) levels == level-xor 32..2, then level-xor 1
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) {
            const double ox = __shfl_xor(hs.x, sft, 64), oy = __shfl_xor(hs.y, sft, 64);
            hs = make_double2(hs.x + ox, hs.y + oy);
        }
        if (valid && l == 0) a.ssh_out[c] = (hs.x + hs.y) - L.rsum[ci];                   // :209 (+N3)
    }
}

template <int ME2, int MODE>
__device__ __forceinline__ void r2edge_issue(R2Edge<ME2, MODE> &b, const RecLds &L, const ColMesh &m, const StageArgs &a,
                                             int ei, int e, uint32_t rowB, uint32_t voff)
{
    const uint32_t *r = L.eRec + (size_t)ei * m.EI;
    const uint32_t own = (uint32_t)e * rowB + voff;
#pragma unroll
    for (int i = 0; i < ME2; ++i) b.uv[i] = gload2(a.pu, r[i] + voff);
    // ssh[c1], ssh[c2]: one lane pair per half-wave fetches them (a 32-lane broadcast load would cost the texture
    // address path as much as a full row); r2edge_finish broadcasts with a shuffle
    b.sA = 0.0;
    b.sB = 0.0;
    if (voff == 0u) b.sA = a.ssh[r[ME2]];
    if (voff == 16u) b.sB = a.ssh[r[ME2 + 1]];
    if constexpr (MODE == 1) b.own = gload2(a.pu, own);
    if constexpr (MODE == 2) b.cur = gload2(a.cu, own);
    if constexpr (MODE >= 2) b.nin = gload2(a.nu_in, own);
}

template <int ME2, int MODE>
__device__ __forceinline__ void r2edge_finish(const R2Edge<ME2, MODE> &b, const RecLds &L, const ColMesh &m, const StageArgs &a,
                                              int ei, int e, uint32_t rowB, uint32_t voff, int l, int K, bool valid)
{
    const uint32_t *r = L.eRec + (size_t)ei * m.EI;
    const double *rw = L.woe + (size_t)ei * ME2;
    const double *rf = L.feoe + (size_t)ei * ME2;
    const uint32_t mask = r[ME2 + 2];
    const int mlt = (int)r[ME2 + 3];
    const double g = L.g[ei];
    const double sA = __shfl(b.sA, 0, 32), sB = __shfl(b.sB, 1, 32);    // from lanes 0 / 1 of this half-wave
    const double ds = sB - sA;                                         // ssh[c2] - ssh[c1]
    const uint32_t ooff = (uint32_t)e * rowB + voff;
    const int k0 = 2 * l;
    const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
    double2 t = make_double2(0.0, 0.0);
    if (ax) t.x -= g * ds;                                             // pressure_gradient.jl:63
    if (ay) t.y -= g * ds;
#pragma unroll
    for (int i = 0; i < ME2; ++i) {
        const bool on = (mask >> i) & 1u;
        const double px = rw[i] * b.uv[i].x * rf[i], py = rw[i] * b.uv[i].y * rf[i];   // ...coriolis.jl:70-72
        if (on && ax) t.x += px;
        if (on && ay) t.y += py;
    }
    if (valid && k0 < K) {
        if constexpr (MODE == 0) gstore2(a.tendU, ooff, t);
        if constexpr (MODE == 1) {
            gstore2(a.pu_out, ooff, make_double2(b.own.x + a.a * t.x, b.own.y + a.a * t.y));   // time_integration.jl:124
            gstore2(a.nu_out, ooff, make_double2(b.own.x + a.b * t.x, b.own.y + a.b * t.y));   // :134
        }
        if constexpr (MODE == 2) {
            gstore2(a.pu_out, ooff, make_double2(b.cur.x + a.a * t.x, b.cur.y + a.a * t.y));
            gstore2(a.nu_out, ooff, make_double2(b.nin.x + a.b * t.x, b.nin.y + a.b * t.y));
        }
        if constexpr (MODE == 3) gstore2(a.nu_out, ooff, make_double2(b.nin.x + a.b * t.x, b.nin.y + a.b * t.y));
    }
}

template <int ME, int ME2, int MODE>
__global__ __launch_bounds__(BLOCK) void k_stage_rec2(const ColMesh m, const StageArgs a, int maxOwnE, int maxOwnC)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int pl_ = patch_of_block_dbg(m.nPatches, a.dbg);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NG = BLOCK / 32;               // 8 half-wave groups
    const int tid = threadIdx.x;
    const int grp = tid >> 5, l = tid & 31;
    const int K = m.K;
    const uint32_t voff = (uint32_t)l * 16u, rowB = (uint32_t)K * 8u;
    const RecLds L = rec_carve(smem, m, ME, ME2, maxOwnE, maxOwnC);
    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
    const int nOwnC = c1 - c0, nOwnE = e1 - e0;

    for (int i = tid; i < nOwnE * m.EI; i += BLOCK) L.eRec[i] = m.eRec[(size_t)e0 * m.EI + i];
    for (int i = tid; i < nOwnE * ME2; i += BLOCK) {
        L.woe[i] = m.woe[(size_t)e0 * ME2 + i];
        L.feoe[i] = m.feoe[(size_t)e0 * ME2 + i];
    }
    for (int i = tid; i < nOwnE; i += BLOCK) L.g[i] = m.gInvDc[e0 + i];
    for (int i = tid; i < nOwnC * m.CI; i += BLOCK) L.cRec[i] = m.cRec[(size_t)c0 * m.CI + i];
    for (int i = tid; i < nOwnC * ME; i += BLOCK) L.sdv[i] = m.sdv[(size_t)c0 * ME + i];
    for (int i = tid; i < nOwnC; i += BLOCK) {
        L.invA[i] = m.invArea[c0 + i];
        L.rsum[i] = m.rsum[c0 + i];
    }
    __syncthreads();

    // the two half-waves of a wave run in lockstep: both iterate max(n_lo, n_hi) times, indices clamped
    {
        const int n = nOwnC > grp ? (nOwnC - grp + NG - 1) / NG : 0;
        const int no = __shfl_xor(n, 32, 64);
        const int nmax = n > no ? n : no;
        if (nmax > 0) {
            auto idx = [&](int t) { int tc = t < n ? t : n - 1; return tc < 0 ? 0 : grp + NG * tc; };
            R2Cell<ME, MODE> A, B;
            r2cell_issue<ME, MODE>(A, L, m, a, idx(0), c0 + idx(0), rowB, voff);
            for (int t = 0;;) {
                r2cell_issue<ME, MODE>(B, L, m, a, idx(t + 1), c0 + idx(t + 1), rowB, voff);
                r2cell_finish<ME, MODE>(A, L, m, a, idx(t), c0 + idx(t), rowB, voff, l, K, t < n);
                if (++t >= nmax) break;
                r2cell_issue<ME, MODE>(A, L, m, a, idx(t + 1), c0 + idx(t + 1), rowB, voff);
                r2cell_finish<ME, MODE>(B, L, m, a, idx(t), c0 + idx(t), rowB, voff, l, K, t < n);
                if (++t >= nmax) break;
            }
        }
    }
    {
        const int n = nOwnE > grp ? (nOwnE - grp + NG - 1) / NG : 0;
        const int no = __shfl_xor(n, 32, 64);
        const int nmax = n > no ? n : no;
        if (nmax > 0) {
            auto idx = [&](int t) { int tc = t < n ? t : n - 1; return tc < 0 ? 0 : grp + NG * tc; };
            R2Edge<ME2, MODE> A, B;
            r2edge_issue<ME2, MODE>(A, L, m, a, idx(0), e0 + idx(0), rowB, voff);
            for (int t = 0;;) {
                r2edge_issue<ME2, MODE>(B, L, m, a, idx(t + 1), e0 + idx(t + 1), rowB, voff);
                r2edge_finish<ME2, MODE>(A, L, m, a, idx(t), e0 + idx(t), rowB, voff, l, K, t < n);
                if (++t >= nmax) break;
                r2edge_issue<ME2, MODE>(A, L, m, a, idx(t + 1), e0 + idx(t + 1), rowB, voff);
                r2edge_finish<ME2, MODE>(B, L, m, a, idx(t), e0 + idx(t), rowB, voff, l, K, t < n);
                if (++t >= nmax) break;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Tiled stage kernel ("tile"): u-rows AND records of a 16-cell patch in LDS, everything else a workgroup
// needs fetched in two dependent bursts, then a compute phase that touches global memory only for stores.
//
// Why: k_stage_rec2 is paced by the per-CU texture-address path (TA_BUSY 82 %): every u-row is fetched
// ~12 times per evaluation (10 Coriolis neighbours + 2 cells), each fetch a 13-cycle TA transaction even
// when it hits L1.  Here each u-row a patch touches (own + halo edges, <= 136 rows) crosses the TA once,
// into LDS; the 36 u-reads per cell become ds_read_b128.  Per 16-cell patch that is ~270 vector memory
// instructions instead of ~520.
// Shape: 256 threads = 8 half-wave groups; lane = two consecutive levels (16 B).  Group g stages rows
// g, g+8, ... and later owns cells g, g+8 and edges g, g+8, ...  Two workgroups per CU (<= 80 KB LDS each),
// 2 waves per SIMD, so up to 256 VGPRs: the h-rows of the group's two cells are prefetched into registers
// in the same burst as the row staging.  One thread per own edge fetches ssh[c1], ssh[c2] and leaves
// ssh[c2]-ssh[c1] in LDS.
// ------------------------------------------------------------------------------------------------
// staged rows per group (RB) and cells per group (MAXC) are template parameters: (17, 2) covers 16-cell patches at
// two workgroups per CU; (11, 1) covers 8-cell patches (<= 88 rows) at three workgroups per CU

struct TileLds {
    double *ubuf, *woe, *feoe, *g, *ds, *sdv, *invA, *rsum;
    int32_t *ehdr, *coc, *mltc;
    uint32_t *leOff, *lcOff;
};

__device__ __forceinline__ TileLds tile_carve(unsigned char *smem, int K, int ME, int ME2, int maxRows, int maxOwnE, int maxOwnC)
{
    TileLds L;
    L.ubuf = reinterpret_cast<double *>(smem);
    L.woe = L.ubuf + (size_t)maxRows * K;
    L.feoe = L.woe + (size_t)maxOwnE * ME2;
    L.g = L.feoe + (size_t)maxOwnE * ME2;
    L.ds = L.g + maxOwnE;
    L.sdv = L.ds + maxOwnE;
    L.invA = L.sdv + (size_t)maxOwnC * ME;
    L.rsum = L.invA + maxOwnC;
    L.ehdr = reinterpret_cast<int32_t *>(L.rsum + maxOwnC);
    L.coc = L.ehdr + (size_t)maxOwnE * 4;
    L.mltc = L.coc + (size_t)maxOwnC * ME;
    L.leOff = reinterpret_cast<uint32_t *>(L.mltc + (size_t)maxOwnC * ME);
    L.lcOff = L.leOff + (size_t)maxOwnE * ME2;
    return L;
}

template <int ME, int MODE>
struct TCell {
    double2 hc, hv[ME], cur, nin;
};

template <int ME, int ME2, int MODE, int TILE_RB, int TILE_MAXC>
__global__ __launch_bounds__(BLOCK, (TILE_RB <= 11 ? 3 : 2)) void k_stage_tile(const MeshDev m, const StageArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    constexpr int NG = BLOCK / 32;
    const int tid = threadIdx.x, grp = tid >> 5, l = tid & 31;
    const int K = m.K, K2 = K >> 1;
    const uint32_t rowB = (uint32_t)K * 8u, voff = (uint32_t)l * 16u;
    const bool act = l < K2;
    const uint32_t voffc = act ? voff : 0u;                            // clamped: every lane issues a valid load
    const TileLds L = tile_carve(smem, K, ME, ME2, m.maxRows, m.maxOwnE, m.maxOwnC);
    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
    const int h0 = cptr(m.haloStart)[p], h1 = cptr(m.haloStart)[p + 1];
    const int nOwnC = c1 - c0, nOwnE = e1 - e0, R = nOwnE + (h1 - h0);

    // ---------------- burst 1: indices ----------------
    int src[TILE_RB];                                                  // global edge of each row this group stages
    const int rs0 = cptr(m.rowStart)[p];
#pragma unroll
    for (int i = 0; i < TILE_RB; ++i) {
        const int r = grp + NG * i;
        const int rc = r < R ? r : (R > 0 ? R - 1 : 0);
        src[i] = m.rowEdge[rs0 + rc];                                  // one unconditional load (a load inside a select gets a vmcnt(0))
    }
    int cn[TILE_MAXC][ME];
    int cidx[TILE_MAXC];
#pragma unroll
    for (int j = 0; j < TILE_MAXC; ++j) {
        const int ci = grp + NG * j;
        cidx[j] = c0 + (ci < nOwnC ? ci : 0);
#pragma unroll
        for (int i = 0; i < ME; ++i) {
            const int x = m.coc[(size_t)cidx[j] * ME + i];
            cn[j][i] = x >= 0 ? x : cidx[j];
        }
    }
    int4 hdr = make_int4(0, 0, 0, 0);
    if (tid < nOwnE) hdr = *reinterpret_cast<const int4 *>(m.ehdr + (size_t)(e0 + tid) * 4);

    // ---------------- burst 2: rows ----------------
    double2 st[TILE_RB];
#pragma unroll
    for (int i = 0; i < TILE_RB; ++i) st[i] = gload2(a.pu, (uint32_t)src[i] * rowB + voffc);
    TCell<ME, MODE> tc[TILE_MAXC];
#pragma unroll
    for (int j = 0; j < TILE_MAXC; ++j) {
        const uint32_t own = (uint32_t)cidx[j] * rowB + voffc;
        tc[j].hc = gload2(a.ph, own);
#pragma unroll
        for (int i = 0; i < ME; ++i) tc[j].hv[i] = gload2(a.ph, (uint32_t)cn[j][i] * rowB + voffc);
        if constexpr (MODE == 2) tc[j].cur = gload2(a.ch, own);
        if constexpr (MODE >= 2) tc[j].nin = gload2(a.nh_in, own);
    }
    double sA = 0.0, sB = 0.0;
    if (tid < nOwnE) {
        sA = a.ssh[hdr.x];
        sB = a.ssh[hdr.y];
    }
    // records of the patch -> LDS (contiguous ranges, coalesced)
    for (int i = tid; i < nOwnE * ME2; i += BLOCK) {
        L.woe[i] = m.woe[(size_t)e0 * ME2 + i];
        L.feoe[i] = m.feoe[(size_t)e0 * ME2 + i];
    }
    for (int i = tid; i < nOwnE; i += BLOCK) L.g[i] = m.gInvDc[e0 + i];
    for (int i = tid; i < nOwnE * ME2; i += BLOCK) L.leOff[i] = m.leOff[(size_t)e0 * ME2 + i];
    if (tid < nOwnE) {
        L.ehdr[tid * 4 + 0] = hdr.x; L.ehdr[tid * 4 + 1] = hdr.y; L.ehdr[tid * 4 + 2] = hdr.z; L.ehdr[tid * 4 + 3] = hdr.w;
    }
    for (int i = tid; i < nOwnC * ME; i += BLOCK) {
        L.sdv[i] = m.sdv[(size_t)c0 * ME + i];
        L.mltc[i] = m.mltc[(size_t)c0 * ME + i];
    }
    for (int i = tid; i < nOwnC; i += BLOCK) {
        L.invA[i] = m.invArea[c0 + i];
        L.rsum[i] = m.rsum[c0 + i];
    }
    for (int i = tid; i < nOwnC * ME; i += BLOCK) L.lcOff[i] = m.lcOff[(size_t)c0 * ME + i];
    // staged rows -> LDS
    double2 *ubuf2 = reinterpret_cast<double2 *>(L.ubuf);
    const unsigned char *ubytes = reinterpret_cast<const unsigned char *>(L.ubuf) + (act ? voff : 0u);
    const bool regular = cptr(m.patchRegular)[p] != 0;                 // block-uniform: predicate-free fast path
#pragma unroll
    for (int i = 0; i < TILE_RB; ++i) {
        const int r = grp + NG * i;
        if (r < R && act) ubuf2[(size_t)r * K2 + l] = st[i];
    }
    if (tid < nOwnE) L.ds[tid] = sB - sA;                              // ssh[c2] - ssh[c1]
    __syncthreads();

    // ---------------- cells (registers + LDS only) ----------------
    const int k0 = 2 * l;
#pragma unroll
    for (int j = 0; j < TILE_MAXC; ++j) {
        const int ci = grp + NG * j;
        const bool valid = ci < nOwnC;
        const int cc = valid ? ci : 0;
        const int c = c0 + cc;
        const double invA = L.invA[cc];
        double2 t = make_double2(0.0, 0.0);
        if (regular) {
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                const double2 uv = *reinterpret_cast<const double2 *>(ubytes + L.lcOff[cc * ME + i]);
                const double sd = L.sdv[cc * ME + i];
                t.x += uv.x * (0.5 * (tc[j].hc.x + tc[j].hv[i].x)) * sd * invA;   // Operators.jl:217, DiagnosticVars.jl:165,
                t.y += uv.y * (0.5 * (tc[j].hc.y + tc[j].hv[i].y)) * sd * invA;   // horizontal_advection.jl:63
            }
        } else {
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                const uint32_t lo = L.lcOff[cc * ME + i];
                const bool on = lo != 0xFFFFFFFFu;
                const double2 uv = *reinterpret_cast<const double2 *>(ubytes + (on ? lo : 0u));
                const int ml = L.mltc[cc * ME + i];
                const double sd = L.sdv[cc * ME + i];
                const double dx = uv.x * (0.5 * (tc[j].hc.x + tc[j].hv[i].x)) * sd * invA;
                const double dy = uv.y * (0.5 * (tc[j].hc.y + tc[j].hv[i].y)) * sd * invA;
                if (on && k0 < ml) t.x += dx;
                if (on && k0 + 1 < ml) t.y += dy;
            }
        }
        const uint32_t ooff = (uint32_t)c * rowB + voff;
        double2 hs = make_double2(0.0, 0.0);
        if (valid && act) {
            if constexpr (MODE == 0) gstore2(a.tendH, ooff, t);
            if constexpr (MODE == 1 || MODE == 2) {
                const double2 hcur = MODE == 2 ? tc[j].cur : tc[j].hc;
                const double2 nb = MODE == 2 ? tc[j].nin : hcur;
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);                    // time_integration.jl:125
                gstore2(a.ph_out, ooff, hs);
                gstore2(a.nh_out, ooff, make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y));    // :135
            }
            if constexpr (MODE == 3) {
                hs = make_double2(tc[j].nin.x + a.b * t.x, tc[j].nin.y + a.b * t.y);
                gstore2(a.nh_out, ooff, hs);
            }
        }
        if constexpr (MODE != 0) {
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) {                   // oracle_ksum order (see k_stage_rec2)
                const double ox = __shfl_xor(hs.x, sft, 64), oy = __shfl_xor(hs.y, sft, 64);
                hs = make_double2(hs.x + ox, hs.y + oy);
            }
            if (valid && l == 0) a.ssh_out[c] = (hs.x + hs.y) - L.rsum[cc];                   // :209 (+N3)
        }
    }

    // ---------------- edges: u from LDS; own Curr/New rows pipelined two deep ----------------
    {
        const int n = nOwnE > grp ? (nOwnE - grp + NG - 1) / NG : 0;
        const int no = __shfl_xor(n, 32, 64);
        const int nmax = n > no ? n : no;
        auto idx = [&](int t) { int tcl = t < n ? t : n - 1; return tcl < 0 ? 0 : grp + NG * tcl; };
        auto issue = [&](double2 &cur, double2 &nin, int ei) {
            const uint32_t own = (uint32_t)(e0 + ei) * rowB + voffc;
            if constexpr (MODE == 2) cur = gload2(a.cu, own);
            if constexpr (MODE >= 2) nin = gload2(a.nu_in, own);
        };
        auto finish = [&](const double2 &cur, const double2 &nin, int ei, bool valid) {
            const double g = L.g[ei], ds = L.ds[ei];
            double2 t = make_double2(0.0, 0.0);
            if (regular) {
                t.x -= g * ds;                                          // pressure_gradient.jl:63
                t.y -= g * ds;
#pragma unroll
                for (int i = 0; i < ME2; ++i) {
                    const double2 uv = *reinterpret_cast<const double2 *>(ubytes + L.leOff[ei * ME2 + i]);
                    const double w = L.woe[ei * ME2 + i], f = L.feoe[ei * ME2 + i];
                    t.x += w * uv.x * f;                                // ...coriolis.jl:70-72
                    t.y += w * uv.y * f;
                }
            } else {
                const int mlt = L.ehdr[ei * 4 + 3];
                const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
                if (ax) t.x -= g * ds;
                if (ay) t.y -= g * ds;
#pragma unroll
                for (int i = 0; i < ME2; ++i) {
                    const uint32_t lo = L.leOff[ei * ME2 + i];
                    const bool on = lo != 0xFFFFFFFFu;
                    const double2 uv = *reinterpret_cast<const double2 *>(ubytes + (on ? lo : 0u));
                    const double w = L.woe[ei * ME2 + i], f = L.feoe[ei * ME2 + i];
                    const double px = w * uv.x * f, py = w * uv.y * f;
                    if (on && ax) t.x += px;
                    if (on && ay) t.y += py;
                }
            }
            const uint32_t ooff = (uint32_t)(e0 + ei) * rowB + voff;
            if (valid && act) {
                if constexpr (MODE == 0) gstore2(a.tendU, ooff, t);
                if constexpr (MODE == 1) {
                    const double2 up = ubuf2[(size_t)ei * K2 + l];      // own row = local row ei
                    gstore2(a.pu_out, ooff, make_double2(up.x + a.a * t.x, up.y + a.a * t.y));   // time_integration.jl:124
                    gstore2(a.nu_out, ooff, make_double2(up.x + a.b * t.x, up.y + a.b * t.y));   // :134
                }
                if constexpr (MODE == 2) {
                    gstore2(a.pu_out, ooff, make_double2(cur.x + a.a * t.x, cur.y + a.a * t.y));
                    gstore2(a.nu_out, ooff, make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y));
                }
                if constexpr (MODE == 3) gstore2(a.nu_out, ooff, make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y));
            }
        };
        if (nmax > 0) {
            double2 cA = make_double2(0, 0), nA = cA, cB = cA, nB = cA;
            issue(cA, nA, idx(0));
            for (int t = 0;;) {
                issue(cB, nB, idx(t + 1));
                finish(cA, nA, idx(t), t < n);
                if (++t >= nmax) break;
                issue(cA, nA, idx(t + 1));
                finish(cB, nB, idx(t), t < n);
                if (++t >= nmax) break;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent, double-buffered tiled stage kernel ("ptile").
//
// k_stage_tile showed that staging u-rows + records in LDS takes the texture-address path out of the
// picture (TA_BUSY 24-40 %) but serialises every workgroup into load -> barrier -> compute with only 2-3
// workgroups per CU to overlap.  Here ONE 512-thread workgroup per CU walks a contiguous chunk of patches
// and software-pipelines across patches: at the top of iteration q it issues *every* global load patch q+1
// needs (u rows to stage, the h rows of its cells, the Curr/New rows of its own cells and edges, ssh pairs,
// records), then computes patch q purely from registers + LDS buffer q&1, and only then parks the arrived
// rows of patch q+1 in LDS buffer (q+1)&1.  One barrier per patch.  vmcnt is in-order, so the compute
// phase must not consume any load younger than the burst: that is why the own rows are prefetched too.
// Index data (row ids, neighbour cells, edge headers) is prefetched one patch further ahead.
// Shape: 16 half-wave groups; per group <= RB staged rows, 1 cell, <= EPG edges per patch.
// ------------------------------------------------------------------------------------------------
constexpr int PBLOCK = 512;

template <int ME, int MODE>
struct PCell {
    double2 hc, hv[ME], cur, nin;
};
struct PEdgeOwn {
    double2 cur, nin;
};

template <int ME, int ME2, int MODE, int RB, int EPG>
__global__ __launch_bounds__(PBLOCK, 2) void k_stage_ptile(const MeshDev m, const StageArgs a, int patchesPerBlock, size_t bufBytes)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NG = PBLOCK / 32;
    const int tid = threadIdx.x, grp = tid >> 5, l = tid & 31;
    const int K = m.K, K2 = K >> 1;
    const uint32_t rowB = (uint32_t)K * 8u, voff = (uint32_t)l * 16u;
    const bool act = l < K2;
    const uint32_t voffc = act ? voff : 0u;
    const int k0 = 2 * l;
    // blocks of one XCD (blockIdx % 8) take adjacent chunks of patches
    const int nb = (int)gridDim.x, chunkB = (nb + 7) >> 3;
    const int bl = (int)(blockIdx.x & 7) * chunkB + (int)(blockIdx.x >> 3);
    const int first = m.patchBegin + bl * patchesPerBlock;
    int n = m.patchBegin + m.nPatches - first;
    if (n > patchesPerBlock) n = patchesPerBlock;
    if (bl >= nb || n <= 0) return;                                    // whole workgroup leaves together
    // two LDS buffers; never indexed with a runtime value (that would push the pointer table to scratch)
    const TileLds L0 = tile_carve(smem, K, ME, ME2, m.maxRows, m.maxOwnE, m.maxOwnC);
    const TileLds L1 = tile_carve(smem + bufBytes, K, ME, ME2, m.maxRows, m.maxOwnE, m.maxOwnC);

    struct IdxS {                                                      // wave-uniform (SGPR) part
        int c0, e0, nOwnC, nOwnE, R, rs;
    };
    struct IdxV {                                                      // per-group part, only needed to issue the loads
        int src[RB];
        int cn[ME];
        int cidx;
        int4 hdr;
    };
    // Patch ranges are read with VECTOR loads on purpose: scalar loads share lgkmcnt with the LDS reads of the
    // compute phase and return out of order, so any LDS wait would become lgkmcnt(0) and stall on them.
    auto load_idx_s = [&](int q) {
        IdxS I;
        const int p = first + (q < n ? q : n - 1);
        const int cA = m.patchCellStart[p], cB = m.patchCellStart[p + 1];
        const int eA = m.patchEdgeStart[p], eB = m.patchEdgeStart[p + 1];
        const int rA = m.rowStart[p], rBv = m.rowStart[p + 1];
        I.c0 = cA; I.e0 = eA; I.nOwnC = cB - cA; I.nOwnE = eB - eA; I.rs = rA; I.R = rBv - rA;
        return I;
    };
    // every load below is unconditional on a clamped index: a load inside a conditional makes the compiler wait
    // for it (vmcnt(0)) at the end of the branch, which serialised the RB halo-list reads of the first version
    auto load_idx_v = [&](const IdxS &I) {
        IdxV V;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int r = grp + NG * i;
            const int rc = r < I.R ? r : (I.R > 0 ? I.R - 1 : 0);
            V.src[i] = m.rowEdge[I.rs + rc];                            // explicit row list (+ one slack element)
        }
        V.cidx = I.c0 + (grp < I.nOwnC ? grp : 0);
#pragma unroll
        for (int i = 0; i < ME; ++i) {
            const int x = m.coc[(size_t)V.cidx * ME + i];
            V.cn[i] = x >= 0 ? x : V.cidx;
        }
        int et = I.e0 + (tid < I.nOwnE ? tid : 0);
        et = et < m.nE ? et : m.nE - 1;
        V.hdr = *reinterpret_cast<const int4 *>(m.ehdr + (size_t)et * 4);
        return V;
    };

    double2 st[RB];                                                    // rows in flight for the NEXT patch
    PCell<ME, MODE> cellC;                                             // h rows (+ own rows) of this group's cell
    PEdgeOwn eoC[EPG], eoN[EPG];                                       // own rows of this group's edges: current / next patch
    double sA = 0.0, sB = 0.0;
    int hdrw = 0;

    auto issue = [&](const IdxS &I, const IdxV &V) {
#pragma unroll
        for (int i = 0; i < RB; ++i) st[i] = gload2(a.pu, (uint32_t)V.src[i] * rowB + voffc);
        const uint32_t own = (uint32_t)V.cidx * rowB + voffc;
        cellC.hc = gload2(a.ph, own);
#pragma unroll
        for (int i = 0; i < ME; ++i) cellC.hv[i] = gload2(a.ph, (uint32_t)V.cn[i] * rowB + voffc);
        if constexpr (MODE == 2) cellC.cur = gload2(a.ch, own);
        if constexpr (MODE >= 2) cellC.nin = gload2(a.nh_in, own);
#pragma unroll
        for (int j = 0; j < EPG; ++j) {
            const int ei = grp + NG * j;
            const uint32_t eown = (uint32_t)(I.e0 + (ei < I.nOwnE ? ei : 0)) * rowB + voffc;
            if constexpr (MODE == 2) eoN[j].cur = gload2(a.cu, eown);
            if constexpr (MODE >= 2) eoN[j].nin = gload2(a.nu_in, eown);
        }
        hdrw = V.hdr.w;
        sA = a.ssh[V.hdr.x];                                           // hdr is always a valid edge's header
        sB = a.ssh[V.hdr.y];
    };
    // records of the patch ride in the same burst, one element per thread, and are parked with the rows: a
    // load -> LDS-store pair placed before the compute phase would make the compute wait for the whole burst
    // (vmcnt is in order).  16 * EPG * ME2 <= 512 and 16 * ME <= 512, so one element per thread is enough.
    double rW = 0.0, rF = 0.0, rG = 0.0, rSd = 0.0, rIa = 0.0, rRs = 0.0;
    uint32_t rLe = 0u, rLc = 0u;
    int rMl = 0;
    auto issue_records = [&](const IdxS &I) {
        const int ne = I.nOwnE * ME2, nc = I.nOwnC * ME;
        const size_t ie = (size_t)I.e0 * ME2 + (tid < ne ? tid : 0), ic = (size_t)I.c0 * ME + (tid < nc ? tid : 0);
        const size_t je = (size_t)I.e0 + (tid < I.nOwnE ? tid : 0), jc = (size_t)I.c0 + (tid < I.nOwnC ? tid : 0);
        const size_t ieC = ie < (size_t)m.nE * ME2 ? ie : 0, jeC = je < (size_t)m.nE ? je : 0;
        rW = m.woe[ieC];
        rF = m.feoe[ieC];
        rLe = m.leOff[ieC];
        rG = m.gInvDc[jeC];
        rSd = m.sdv[ic];
        rMl = m.mltc[ic];
        rLc = m.lcOff[ic];
        rIa = m.invArea[jc];
        rRs = m.rsum[jc];
    };
    auto park_records = [&](const IdxS &I, const TileLds &L) {
        if (tid < I.nOwnE * ME2) {
            L.woe[tid] = rW;
            L.feoe[tid] = rF;
            L.leOff[tid] = rLe;
        }
        if (tid < I.nOwnE) L.g[tid] = rG;
        if (tid < I.nOwnC * ME) {
            L.sdv[tid] = rSd;
            L.mltc[tid] = rMl;
            L.lcOff[tid] = rLc;
        }
        if (tid < I.nOwnC) {
            L.invA[tid] = rIa;
            L.rsum[tid] = rRs;
        }
    };
    auto park = [&](const IdxS &I, const TileLds &L) {                  // arrived rows -> LDS
        double2 *ubuf2 = reinterpret_cast<double2 *>(L.ubuf);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int r = grp + NG * i;
            if (r < I.R && act) ubuf2[(size_t)r * K2 + l] = st[i];
        }
        if (tid < I.nOwnE) {
            L.ds[tid] = sB - sA;                                       // ssh[c2] - ssh[c1]
            L.ehdr[tid * 4 + 3] = hdrw;
        }
    };

    auto compute_cell = [&](const IdxS &I, const TileLds &L) {
        const unsigned char *ubytes = reinterpret_cast<const unsigned char *>(L.ubuf) + (act ? voff : 0u);
        const bool valid = grp < I.nOwnC;
        const int cc = valid ? grp : 0;
        const int c = I.c0 + cc;
        const double invA = L.invA[cc];
        double2 t = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < ME; ++i) {
            const uint32_t lo = L.lcOff[cc * ME + i];
            const bool on = lo != 0xFFFFFFFFu;
            const double2 uv = *reinterpret_cast<const double2 *>(ubytes + (on ? lo : 0u));
            const int ml = L.mltc[cc * ME + i];
            const double sd = L.sdv[cc * ME + i];
            const double dx = uv.x * (0.5 * (cellC.hc.x + cellC.hv[i].x)) * sd * invA;   // Operators.jl:217, DiagnosticVars.jl:165,
            const double dy = uv.y * (0.5 * (cellC.hc.y + cellC.hv[i].y)) * sd * invA;   // horizontal_advection.jl:63
            if (on && k0 < ml) t.x += dx;
            if (on && k0 + 1 < ml) t.y += dy;
        }
        const uint32_t ooff = (uint32_t)c * rowB + voff;
        double2 hs = make_double2(0.0, 0.0);
        if (valid && act) {
            if constexpr (MODE == 0) gstore2(a.tendH, ooff, t);
            if constexpr (MODE == 1 || MODE == 2) {
                const double2 hcur = MODE == 2 ? cellC.cur : cellC.hc;
                const double2 nbv = MODE == 2 ? cellC.nin : hcur;
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);                    // time_integration.jl:125
                gstore2(a.ph_out, ooff, hs);
                gstore2(a.nh_out, ooff, make_double2(nbv.x + a.b * t.x, nbv.y + a.b * t.y));  // :135
            }
            if constexpr (MODE == 3) {
                hs = make_double2(cellC.nin.x + a.b * t.x, cellC.nin.y + a.b * t.y);
                gstore2(a.nh_out, ooff, hs);
            }
        }
        if constexpr (MODE != 0) {
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) {                   // oracle_ksum order (see k_stage_rec2)
                const double ox = __shfl_xor(hs.x, sft, 64), oy = __shfl_xor(hs.y, sft, 64);
                hs = make_double2(hs.x + ox, hs.y + oy);
            }
            if (valid && l == 0) a.ssh_out[c] = (hs.x + hs.y) - L.rsum[cc];                   // :209 (+N3)
        }
    };
    auto compute_edges = [&](const IdxS &I, const TileLds &L) {
        const unsigned char *ubytes = reinterpret_cast<const unsigned char *>(L.ubuf) + (act ? voff : 0u);
        const double2 *ubuf2 = reinterpret_cast<const double2 *>(L.ubuf);
#pragma unroll
        for (int j = 0; j < EPG; ++j) {
            const int eiq = grp + NG * j;
            const bool valid = eiq < I.nOwnE;
            const int ei = valid ? eiq : 0;
            const int mlt = L.ehdr[ei * 4 + 3];
            const double g = L.g[ei], ds = L.ds[ei];
            const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
            double2 t = make_double2(0.0, 0.0);
            if (ax) t.x -= g * ds;                                      // pressure_gradient.jl:63
            if (ay) t.y -= g * ds;
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                const uint32_t lo = L.leOff[ei * ME2 + i];
                const bool on = lo != 0xFFFFFFFFu;
                const double2 uv = *reinterpret_cast<const double2 *>(ubytes + (on ? lo : 0u));
                const double w = L.woe[ei * ME2 + i], f = L.feoe[ei * ME2 + i];
                const double px = w * uv.x * f, py = w * uv.y * f;      // ...coriolis.jl:70-72
                if (on && ax) t.x += px;
                if (on && ay) t.y += py;
            }
            const uint32_t ooff = (uint32_t)(I.e0 + ei) * rowB + voff;
            if (valid && act) {
                if constexpr (MODE == 0) gstore2(a.tendU, ooff, t);
                if constexpr (MODE == 1) {
                    const double2 up = ubuf2[(size_t)ei * K2 + l];      // own row = local row ei
                    gstore2(a.pu_out, ooff, make_double2(up.x + a.a * t.x, up.y + a.a * t.y));   // time_integration.jl:124
                    gstore2(a.nu_out, ooff, make_double2(up.x + a.b * t.x, up.y + a.b * t.y));   // :134
                }
                if constexpr (MODE == 2) {
                    gstore2(a.pu_out, ooff, make_double2(eoC[j].cur.x + a.a * t.x, eoC[j].cur.y + a.a * t.y));
                    gstore2(a.nu_out, ooff, make_double2(eoC[j].nin.x + a.b * t.x, eoC[j].nin.y + a.b * t.y));
                }
                if constexpr (MODE == 3) gstore2(a.nu_out, ooff, make_double2(eoC[j].nin.x + a.b * t.x, eoC[j].nin.y + a.b * t.y));
            }
            __builtin_amdgcn_sched_barrier(0);                          // keep one edge's LDS reads from piling onto the next's
        }
    };

    // ---------------- prologue: patch 0 into buffer 0 ----------------
    IdxS Icur = load_idx_s(0);
    {
        const IdxV V0 = load_idx_v(Icur);
        issue(Icur, V0);
    }
    issue_records(Icur);
    IdxS Inext = load_idx_s(1);
    IdxV Vnext = load_idx_v(Inext);
    IdxS Iaft = load_idx_s(2);
    park(Icur, L0);
    park_records(Icur, L0);
#pragma unroll
    for (int j = 0; j < EPG; ++j) eoC[j] = eoN[j];
    __syncthreads();

    // ---------------- steady state ----------------
    // iteration q: ranges of patch q+3 and row/neighbour ids of patch q+2 are requested, the burst of patch q+1 is
    // issued, patch q is computed.  Each of those is consumed one iteration after it was requested, behind a
    // counted vmcnt, so nothing in an iteration waits for that iteration's own loads except the final park.
    auto iteration = [&](int q, const TileLds &Lcur, const TileLds &Lnxt) {
        compute_cell(Icur, Lcur);                                      // frees cellC for the next patch
        __builtin_amdgcn_sched_barrier(0);
        issue(Inext, Vnext);                                           // unconditional (index clamped): static load count
        issue_records(Inext);
        const IdxS Iaft2 = load_idx_s(q + 3);
        const IdxV Vaft = load_idx_v(Iaft);
        __builtin_amdgcn_sched_barrier(0);
        compute_edges(Icur, Lcur);                                     // registers + LDS only: overlaps the burst above
        __builtin_amdgcn_sched_barrier(0);
        park(Inext, Lnxt);
        park_records(Inext, Lnxt);
#pragma unroll
        for (int j = 0; j < EPG; ++j) eoC[j] = eoN[j];
        Icur = Inext;
        Inext = Iaft;
        Vnext = Vaft;
        Iaft = Iaft2;
        __syncthreads();
    };
    for (int q = 0; q < n; q += 2) {
        iteration(q, L0, L1);
        if (q + 1 < n) iteration(q + 1, L1, L0);
    }
}

typedef const __attribute__((address_space(3))) unsigned char *lds_bytes_t;
typedef const __attribute__((address_space(1))) unsigned char *glb_bytes_t;
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
// N 16-byte LDS reads issued back to back, one wait.  Inline assembly because a plain LDS load next to a global
// load of the other branch is merged by the compiler into ONE flat_load from a selected pointer (a flat load of an LDS
// address still occupies the texture-address unit), and a volatile LDS load is waited for individually.
typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
template <int N>
__device__ __forceinline__ void lds_burst(v4u_t (&v)[N], const uint32_t (&ad)[N]);
template <>
__device__ __forceinline__ void lds_burst<6>(v4u_t (&v)[6], const uint32_t (&ad)[6])
{
    asm volatile("ds_read_b128 %[o0], %[a0]\n\t"
                 "ds_read_b128 %[o1], %[a1]\n\t"
                 "ds_read_b128 %[o2], %[a2]\n\t"
                 "ds_read_b128 %[o3], %[a3]\n\t"
                 "ds_read_b128 %[o4], %[a4]\n\t"
                 "ds_read_b128 %[o5], %[a5]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [o0] "=&v"(v[0]), [o1] "=&v"(v[1]), [o2] "=&v"(v[2]), [o3] "=&v"(v[3]), [o4] "=&v"(v[4]), [o5] "=&v"(v[5])
                 : [a0] "v"(ad[0]), [a1] "v"(ad[1]), [a2] "v"(ad[2]), [a3] "v"(ad[3]), [a4] "v"(ad[4]), [a5] "v"(ad[5])
                 : "memory");
}
template <>
__device__ __forceinline__ void lds_burst<8>(v4u_t (&v)[8], const uint32_t (&ad)[8])
{
    asm volatile("ds_read_b128 %[o0], %[a0]\n\t"
                 "ds_read_b128 %[o1], %[a1]\n\t"
                 "ds_read_b128 %[o2], %[a2]\n\t"
                 "ds_read_b128 %[o3], %[a3]\n\t"
                 "ds_read_b128 %[o4], %[a4]\n\t"
                 "ds_read_b128 %[o5], %[a5]\n\t"
                 "ds_read_b128 %[o6], %[a6]\n\t"
                 "ds_read_b128 %[o7], %[a7]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [o0] "=&v"(v[0]), [o1] "=&v"(v[1]), [o2] "=&v"(v[2]), [o3] "=&v"(v[3]), [o4] "=&v"(v[4]), [o5] "=&v"(v[5]), [o6] "=&v"(v[6]), [o7] "=&v"(v[7])
                 : [a0] "v"(ad[0]), [a1] "v"(ad[1]), [a2] "v"(ad[2]), [a3] "v"(ad[3]), [a4] "v"(ad[4]), [a5] "v"(ad[5]), [a6] "v"(ad[6]), [a7] "v"(ad[7])
                 : "memory");
}
template <>
__device__ __forceinline__ void lds_burst<10>(v4u_t (&v)[10], const uint32_t (&ad)[10])
{
    asm volatile("ds_read_b128 %[o0], %[a0]\n\t"
                 "ds_read_b128 %[o1], %[a1]\n\t"
                 "ds_read_b128 %[o2], %[a2]\n\t"
                 "ds_read_b128 %[o3], %[a3]\n\t"
                 "ds_read_b128 %[o4], %[a4]\n\t"
                 "ds_read_b128 %[o5], %[a5]\n\t"
                 "ds_read_b128 %[o6], %[a6]\n\t"
                 "ds_read_b128 %[o7], %[a7]\n\t"
                 "ds_read_b128 %[o8], %[a8]\n\t"
                 "ds_read_b128 %[o9], %[a9]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [o0] "=&v"(v[0]), [o1] "=&v"(v[1]), [o2] "=&v"(v[2]), [o3] "=&v"(v[3]), [o4] "=&v"(v[4]), [o5] "=&v"(v[5]), [o6] "=&v"(v[6]), [o7] "=&v"(v[7]), [o8] "=&v"(v[8]), [o9] "=&v"(v[9])
                 : [a0] "v"(ad[0]), [a1] "v"(ad[1]), [a2] "v"(ad[2]), [a3] "v"(ad[3]), [a4] "v"(ad[4]), [a5] "v"(ad[5]), [a6] "v"(ad[6]), [a7] "v"(ad[7]), [a8] "v"(ad[8]), [a9] "v"(ad[9])
                 : "memory");
}
template <>
__device__ __forceinline__ void lds_burst<14>(v4u_t (&v)[14], const uint32_t (&ad)[14])
{
    asm volatile("ds_read_b128 %[o0], %[a0]\n\t"
                 "ds_read_b128 %[o1], %[a1]\n\t"
                 "ds_read_b128 %[o2], %[a2]\n\t"
                 "ds_read_b128 %[o3], %[a3]\n\t"
                 "ds_read_b128 %[o4], %[a4]\n\t"
                 "ds_read_b128 %[o5], %[a5]\n\t"
                 "ds_read_b128 %[o6], %[a6]\n\t"
                 "ds_read_b128 %[o7], %[a7]\n\t"
                 "ds_read_b128 %[o8], %[a8]\n\t"
                 "ds_read_b128 %[o9], %[a9]\n\t"
                 "ds_read_b128 %[o10], %[a10]\n\t"
                 "ds_read_b128 %[o11], %[a11]\n\t"
                 "ds_read_b128 %[o12], %[a12]\n\t"
                 "ds_read_b128 %[o13], %[a13]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [o0] "=&v"(v[0]), [o1] "=&v"(v[1]), [o2] "=&v"(v[2]), [o3] "=&v"(v[3]), [o4] "=&v"(v[4]), [o5] "=&v"(v[5]), [o6] "=&v"(v[6]), [o7] "=&v"(v[7]), [o8] "=&v"(v[8]), [o9] "=&v"(v[9]), [o10] "=&v"(v[10]), [o11] "=&v"(v[11]), [o12] "=&v"(v[12]), [o13] "=&v"(v[13])
                 : [a0] "v"(ad[0]), [a1] "v"(ad[1]), [a2] "v"(ad[2]), [a3] "v"(ad[3]), [a4] "v"(ad[4]), [a5] "v"(ad[5]), [a6] "v"(ad[6]), [a7] "v"(ad[7]), [a8] "v"(ad[8]), [a9] "v"(ad[9]), [a10] "v"(ad[10]), [a11] "v"(ad[11]), [a12] "v"(ad[12]), [a13] "v"(ad[13])
                 : "memory");
}
__device__ __forceinline__ double2 glb_row2(glb_bytes_t p)
{
    const v2d_t v = *(const __attribute__((address_space(1))) v2d_t *)p;
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ float4 glb_row4f(glb_bytes_t p)
{
    const v4f_t v = *(const __attribute__((address_space(1))) v4f_t *)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// ------------------------------------------------------------------------------------------------
// rec2 + own-edge cache ("rec2c"): k_stage_rec2 with the u-rows of the patch's OWN edges copied once into LDS
// (a contiguous range: one coalesced copy, no halo list).  ~65 % of all u gathers of a compact patch refer to
// its own edges; those become ds_read_b128 and their texture-address transactions disappear.  A gather whose
// row is not cached is an exec-masked global load (the two half-waves of a wave decide independently).
// Not pipelined: loads sit behind per-lane branches, so the compiler waits with vmcnt(0) at first use; 12+
// waves per CU cover the latency instead.
// ------------------------------------------------------------------------------------------------
template <int ME, int ME2, int MODE, int NT = BLOCK>
__global__ __launch_bounds__(NT) void k_stage_rec2c(const ColMesh m, const StageArgs a, int maxOwnE, int maxOwnC)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
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
    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
    const int nOwnC = c1 - c0, nOwnE = e1 - e0;
    const uint32_t e0B = (uint32_t)e0 * rowB, nOwnB = (uint32_t)nOwnE * rowB;

    for (int i = tid; i < nOwnE * m.EI; i += NT) L.eRec[i] = m.eRec[(size_t)e0 * m.EI + i];
    for (int i = tid; i < nOwnE * ME2; i += NT) {
        L.woe[i] = m.woe[(size_t)e0 * ME2 + i];
        L.feoe[i] = m.feoe[(size_t)e0 * ME2 + i];
    }
    for (int i = tid; i < nOwnE; i += NT) L.g[i] = m.gInvDc[e0 + i];
    for (int i = tid; i < nOwnC * m.CI; i += NT) L.cRec[i] = m.cRec[(size_t)c0 * m.CI + i];
    for (int i = tid; i < nOwnC * ME; i += NT) L.sdv[i] = m.sdv[(size_t)c0 * ME + i];
    for (int i = tid; i < nOwnC; i += NT) {
        L.invA[i] = m.invArea[c0 + i];
        L.rsum[i] = m.rsum[c0 + i];
    }
    {   // own u rows: one contiguous, fully coalesced copy
        const double2 *src = reinterpret_cast<const double2 *>(a.pu) + (size_t)e0 * K2;
        for (int i = tid; i < nOwnE * K2; i += NT) ubuf2[i] = src[i];
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
    auto urow_glb = [&](uint32_t off) -> double2 { return glb_row2(puG + (off + voff)); };

    // ---------------- cells ----------------
    for (int ci = grp; ci < nOwnC; ci += NG) {
        const int c = c0 + ci;
        const uint32_t *r = L.cRec + (size_t)ci * m.CI;
        const double *rs = L.sdv + (size_t)ci * ME;
        const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
        const double invA = L.invA[ci];
        const uint32_t own = (uint32_t)c * rowB + voff;
        double2 hc = make_double2(0.0, 0.0), uv[ME], hv[ME], cur = hc, nin = hc;
        if (act) {
            bool cached[ME];
            uint32_t ad[ME];
            v4u_t raw[ME];
            hc = gload2(a.ph, own);
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                hv[i] = gload2(a.ph, r[ME + i] + voff);
                ad[i] = urow_addr(r[i], cached[i]);
            }
            lds_burst<ME>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                uv[i] = __builtin_bit_cast(double2, raw[i]);
                if (!cached[i]) uv[i] = urow_glb(r[i]);
            }
            if constexpr (MODE == 2) cur = gload2(a.ch, own);
            if constexpr (MODE >= 2) nin = gload2(a.nh_in, own);
        }
        double2 t = make_double2(0.0, 0.0);
        // regular entity (every slot valid, every level active) in BOTH half-waves: no per-slot masks (wave-uniform branch)
        const bool plain = __builtin_amdgcn_ballot_w64(!(mask == (1u << ME) - 1u && all)) == 0;
        if (plain) {
            if (act) {
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    t.x += uv[i].x * (0.5 * (hc.x + hv[i].x)) * rs[i] * invA;
                    t.y += uv[i].y * (0.5 * (hc.y + hv[i].y)) * rs[i] * invA;
                }
            }
        } else if (act) {
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                const int ml = all ? K : cptr(m.mltc)[(size_t)c * ME + i];
                const bool on = (mask >> i) & 1u;
                const double dx = uv[i].x * (0.5 * (hc.x + hv[i].x)) * rs[i] * invA;   // Operators.jl:217, DiagnosticVars.jl:165,
                const double dy = uv[i].y * (0.5 * (hc.y + hv[i].y)) * rs[i] * invA;   // horizontal_advection.jl:63
                if (on && k0 < ml) t.x += dx;
                if (on && k0 + 1 < ml) t.y += dy;
            }
        }
        double2 hs = make_double2(0.0, 0.0);
        if (act) {
            if constexpr (MODE == 0) gstore2(a.tendH, own, t);
            if constexpr (MODE == 1 || MODE == 2) {
                const double2 hcur = MODE == 2 ? cur : hc;
                const double2 nb = MODE == 2 ? nin : hcur;
                hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);                    // time_integration.jl:125
                gstore2(a.ph_out, own, hs);
                gstore2(a.nh_out, own, make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y));     // :135
            }
            if constexpr (MODE == 3) {
                hs = make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y);
                gstore2(a.nh_out, own, hs);
            }
        }
        if constexpr (MODE != 0) {
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) {                   // oracle_ksum order (see k_stage_rec2)
                const double ox = __shfl_xor(hs.x, sft, 32), oy = __shfl_xor(hs.y, sft, 32);
                hs = make_double2(hs.x + ox, hs.y + oy);
            }
            if (l == 0) a.ssh_out[c] = (hs.x + hs.y) - L.rsum[ci];                            // :209 (+N3)
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
        double sA = 0.0, sB = 0.0;
        if (l == 0) sA = a.ssh[r[ME2]];
        if (l == 1) sB = a.ssh[r[ME2 + 1]];
        double2 uv[ME2], cur = make_double2(0.0, 0.0), nin = cur;
        if (act) {
            bool cached[ME2];
            uint32_t ad[ME2];
            v4u_t raw[ME2];
#pragma unroll
            for (int i = 0; i < ME2; ++i) ad[i] = urow_addr(r[i], cached[i]);
            lds_burst<ME2>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                uv[i] = __builtin_bit_cast(double2, raw[i]);
                if (!cached[i]) uv[i] = urow_glb(r[i]);
            }
            if constexpr (MODE == 2) cur = gload2(a.cu, own);
            if constexpr (MODE >= 2) nin = gload2(a.nu_in, own);
        }
        const double ds = __shfl(sB, 1, 32) - __shfl(sA, 0, 32);       // ssh[c2] - ssh[c1]
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
            if constexpr (MODE == 0) gstore2(a.tendU, own, t);
            if constexpr (MODE == 1) {
                const double2 up = ubuf2[(size_t)ei * K2 + l];          // own row is in the cache
                gstore2(a.pu_out, own, make_double2(up.x + a.a * t.x, up.y + a.a * t.y));   // time_integration.jl:124
                gstore2(a.nu_out, own, make_double2(up.x + a.b * t.x, up.y + a.b * t.y));   // :134
            }
            if constexpr (MODE == 2) {
                gstore2(a.pu_out, own, make_double2(cur.x + a.a * t.x, cur.y + a.a * t.y));
                gstore2(a.nu_out, own, make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y));
            }
            if constexpr (MODE == 3) gstore2(a.nu_out, own, make_double2(nin.x + a.b * t.x, nin.y + a.b * t.y));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32-state form of rec2c (BASELINE config 5: "fp32 state with fp64 tendency accumulation").
// ssh / normalVelocity / layerThickness of every time level and RK provisional state are stored as fp32
// (rows of K*4 bytes); a lane owns FOUR consecutive levels (one 16-byte load), K/4 lanes one entity and a wave
// 64/(K/4) entities (K <= 128, K % 4 == 0).  Every load widens to fp64, the arithmetic is that of k_stage_rec2c in the
// same order, stores round to nearest fp32; tendencies (MODE 0) are written as fp64.  The StageArgs pointers
// of state arrays are float arrays in disguise (the host keeps one argument block for both storage types).
// The byte-offset records of such a mesh are built for K*4-byte rows (moka_mesh_desc.stateBytes = 4).
// ssh column sum: oracle_ksum order -- lanes l and l^16 hold levels k and k^64, then k^32 ... k^4, and the four
// levels of a lane combine as (x+z)+(y+w), i.e. k^2 then k^1.
// ------------------------------------------------------------------------------------------------
struct d4 {
    double x, y, z, w;
};
__device__ __forceinline__ d4 widen4(float4 v) { return d4{(double)v.x, (double)v.y, (double)v.z, (double)v.w}; }
__device__ __forceinline__ float4 narrow4(d4 v) { return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
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

template <int ME, int ME2, int MODE>
__global__ __launch_bounds__(BLOCK, 3) void k_stage_rec2c_f32(const ColMesh m, const StageArgs a, int maxOwnE, int maxOwnC)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int pl_ = patch_of_block(m.nPatches);
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    // K/4 lanes carry one entity, so a wave carries 64 / (K/4) of them (3 at K = 80, 4 at K = 60 or 64, 2 at K = 128):
    // lanes beyond the last whole group idle.  Shuffles address lanes of the own group only.
    const int tid = threadIdx.x;
    const int K = m.K, K4 = K >> 2;
    const int EPW = 64 / K4, NG = (BLOCK / 64) * EPW;
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
    const uint32_t voffD = (uint32_t)l * 32u, rowBD = (uint32_t)K * 8u;    // fp64 rows (tendency outputs)
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

    for (int i = tid; i < nOwnE * m.EI; i += BLOCK) L.eRec[i] = m.eRec[(size_t)e0 * m.EI + i];
    for (int i = tid; i < nOwnE * ME2; i += BLOCK) {
        L.woe[i] = m.woe[(size_t)e0 * ME2 + i];
        L.feoe[i] = m.feoe[(size_t)e0 * ME2 + i];
    }
    for (int i = tid; i < nOwnE; i += BLOCK) L.g[i] = m.gInvDc[e0 + i];
    for (int i = tid; i < nOwnC * m.CI; i += BLOCK) L.cRec[i] = m.cRec[(size_t)c0 * m.CI + i];
    for (int i = tid; i < nOwnC * ME; i += BLOCK) L.sdv[i] = m.sdv[(size_t)c0 * ME + i];
    for (int i = tid; i < nOwnC; i += BLOCK) {
        L.invA[i] = m.invArea[c0 + i];
        L.rsum[i] = m.rsum[c0 + i];
    }
    {   // own u rows: one contiguous, fully coalesced copy
        const float4 *src = reinterpret_cast<const float4 *>(a.pu) + (size_t)e0 * K4;
        for (int i = tid; i < nOwnE * K4; i += BLOCK) ubuf4[i] = src[i];
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
    auto urow_glb = [&](uint32_t off) -> float4 { return glb_row4f(puG + (off + voff)); };
    const d4 zero{0.0, 0.0, 0.0, 0.0};

    // ---------------- cells ----------------
    for (int ci = grp; ci < nOwnC; ci += NG) {
        const int c = c0 + ci;
        const uint32_t *r = L.cRec + (size_t)ci * m.CI;
        const double *rs = L.sdv + (size_t)ci * ME;
        const uint32_t mask = r[2 * ME], all = r[2 * ME + 1];
        const double invA = L.invA[ci];
        const uint32_t own = (uint32_t)c * rowB + voff;
        d4 hc = zero, uv[ME], hv[ME], cur = zero, nin = zero;
        if (act) {
            bool cached[ME];
            uint32_t ad[ME];
            v4u_t raw[ME];
            float4 uf[ME];
            hc = gload4(a.ph, own);
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                hv[i] = gload4(a.ph, r[ME + i] + voff);
                ad[i] = urow_addr(r[i], cached[i]);
            }
            lds_burst<ME>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                uf[i] = __builtin_bit_cast(float4, raw[i]);
                if (!cached[i]) uf[i] = urow_glb(r[i]);
            }
            if constexpr (MODE == 2) cur = gload4(a.ch, own);
            if constexpr (MODE >= 2) nin = gload4(a.nh_in, own);
#pragma unroll
            for (int i = 0; i < ME; ++i) uv[i] = widen4(uf[i]);
        }
        d4 t = zero;
        const bool plain = __builtin_amdgcn_ballot_w64(!(mask == (1u << ME) - 1u && all)) == 0;   // see k_stage_rec2c
        if (plain) {
            if (act) {
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    t.x += uv[i].x * (0.5 * (hc.x + hv[i].x)) * rs[i] * invA;
                    t.y += uv[i].y * (0.5 * (hc.y + hv[i].y)) * rs[i] * invA;
                    t.z += uv[i].z * (0.5 * (hc.z + hv[i].z)) * rs[i] * invA;
                    t.w += uv[i].w * (0.5 * (hc.w + hv[i].w)) * rs[i] * invA;
                }
            }
        } else if (act) {
#pragma unroll
            for (int i = 0; i < ME; ++i) {
                const int ml = all ? K : cptr(m.mltc)[(size_t)c * ME + i];
                const bool on = (mask >> i) & 1u;
                const double dx = uv[i].x * (0.5 * (hc.x + hv[i].x)) * rs[i] * invA;   // Operators.jl:217, DiagnosticVars.jl:165,
                const double dy = uv[i].y * (0.5 * (hc.y + hv[i].y)) * rs[i] * invA;   // horizontal_advection.jl:63
                const double dz = uv[i].z * (0.5 * (hc.z + hv[i].z)) * rs[i] * invA;
                const double dw = uv[i].w * (0.5 * (hc.w + hv[i].w)) * rs[i] * invA;
                if (on && k0 < ml) t.x += dx;
                if (on && k0 + 1 < ml) t.y += dy;
                if (on && k0 + 2 < ml) t.z += dz;
                if (on && k0 + 3 < ml) t.w += dw;
            }
        }
        d4 hs = zero;
        if (act) {
            if constexpr (MODE == 0) {
                const uint32_t ownD = (uint32_t)c * rowBD + voffD;
                gstore2(a.tendH, ownD, make_double2(t.x, t.y));
                gstore2(a.tendH, ownD + 16u, make_double2(t.z, t.w));
            }
            if constexpr (MODE == 1 || MODE == 2) {
                const d4 hcur = MODE == 2 ? cur : hc;
                const d4 nb = MODE == 2 ? nin : hcur;
                hs = round4(axpy4(hcur, a.a, t));                                             // time_integration.jl:125
                gstore4(a.ph_out, own, hs);
                gstore4(a.nh_out, own, axpy4(nb, a.b, t));                                    // :135
            }
            if constexpr (MODE == 3) {
                hs = round4(axpy4(nin, a.b, t));
                gstore4(a.nh_out, own, hs);
            }
        }
        if constexpr (MODE != 0) {
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) {
                hs = d4{hs.x + gxor(hs.x, sft), hs.y + gxor(hs.y, sft), hs.z + gxor(hs.z, sft), hs.w + gxor(hs.w, sft)};
            }
            if (l == 0)                                                                       // :209 (+N3), stored fp32
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
        double ds0 = 0.0;
        if (l == 0) ds0 = (double)sshf[r[ME2 + 1]] - (double)sshf[r[ME2]];     // ssh[c2] - ssh[c1], in the group's first lane
        d4 uv[ME2], cur = zero, nin = zero;
        if (act) {
            bool cached[ME2];
            uint32_t ad[ME2];
            v4u_t raw[ME2];
            float4 uf[ME2];
#pragma unroll
            for (int i = 0; i < ME2; ++i) ad[i] = urow_addr(r[i], cached[i]);
            lds_burst<ME2>(raw, ad);
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                uf[i] = __builtin_bit_cast(float4, raw[i]);
                if (!cached[i]) uf[i] = urow_glb(r[i]);
            }
            if constexpr (MODE == 2) cur = gload4(a.cu, own);
            if constexpr (MODE >= 2) nin = gload4(a.nu_in, own);
#pragma unroll
            for (int i = 0; i < ME2; ++i) uv[i] = widen4(uf[i]);
        }
        const double ds = __shfl(ds0, gbase, 64);
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
            if constexpr (MODE == 0) {
                const uint32_t ownD = (uint32_t)e * rowBD + voffD;
                gstore2(a.tendU, ownD, make_double2(t.x, t.y));
                gstore2(a.tendU, ownD + 16u, make_double2(t.z, t.w));
            }
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
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS patch-tiled variant of the fused tendency / RK-stage kernel (same arithmetic, same results).
//
// The direct kernel above re-reads every u-row ~12 times through the vector L1 (10 Coriolis
// neighbours + 2 cells), which rocprof shows as ~60 % TA utilisation and ~70 % of wave time parked
// on memory.  Here one 512-thread workgroup owns one patch and
//   1. stages the u-rows of every edge the patch touches (own + halo, <= 254 rows of K*8 bytes) and
//      all of the patch's connectivity / weight records into LDS in one burst of coalesced 16-byte
//      loads (many rows in flight per wave -> deep memory-level parallelism),
//   2. after one barrier, evaluates its cells and edges out of LDS: a 32-lane half-wave owns one
//      entity, each lane two consecutive levels (ds_read_b128, conflict-free on 480-byte rows);
//      neighbour indices are patch-local bytes read from LDS.
// h-rows (7 per cell) and the RK Curr/New rows are read straight from global memory.
// Two workgroups fit a CU (<= 80 KB LDS each) so one loads while the other computes.
// ------------------------------------------------------------------------------------------------
constexpr int LBLOCK = 512;

__device__ __forceinline__ double2 shfl_xor2(double2 v, int s)
{
    return make_double2(__shfl_xor(v.x, s, 64), __shfl_xor(v.y, s, 64));
}

template <int ME, int ME2>
__global__ __launch_bounds__(LBLOCK, 4) void k_stage_lds(const MeshDev m, const StageArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
    if (pl_ >= m.nPatches) return;
    const int p = pl_ + m.patchBegin;
    const int K = m.K, K2 = K >> 1;                 // K is even (checked on the host)
    constexpr int NG = LBLOCK / 32;                 // 16 half-wave groups
    const int tid = threadIdx.x, grp = tid >> 5, l = tid & 31;

    double *ubuf = reinterpret_cast<double *>(smem);
    double *fbuf = ubuf + (size_t)m.maxRows * K;
    double *wbuf = fbuf + m.maxRows;
    double *gbuf = wbuf + (size_t)m.maxOwnE * ME2;
    double *sbuf = gbuf + m.maxOwnE;
    double *iabuf = sbuf + (size_t)m.maxOwnC * ME;
    double *rsbuf = iabuf + m.maxOwnC;
    int32_t *hbuf = reinterpret_cast<int32_t *>(rsbuf + m.maxOwnC);
    int32_t *cbuf = hbuf + (size_t)m.maxOwnE * 4;
    int32_t *mbuf = cbuf + (size_t)m.maxOwnC * ME;
    uint32_t *lebuf = reinterpret_cast<uint32_t *>(mbuf + (size_t)m.maxOwnC * ME);
    uint32_t *lcbuf = lebuf + (size_t)m.maxOwnE * 4;
    double2 *ubuf2 = reinterpret_cast<double2 *>(ubuf);

    const int c0 = cptr(m.patchCellStart)[p], c1 = cptr(m.patchCellStart)[p + 1];
    const int e0 = cptr(m.patchEdgeStart)[p], e1 = cptr(m.patchEdgeStart)[p + 1];
    const int h0 = cptr(m.haloStart)[p], h1 = cptr(m.haloStart)[p + 1];
    const int nOwnC = c1 - c0, nOwnE = e1 - e0, R = nOwnE + (h1 - h0);

    // ---- 1. stage: u rows (own edges are contiguous in memory, halo rows are gathered) ----
    const double2 *pu2 = reinterpret_cast<const double2 *>(a.pu);
    constexpr int RB = 5;                           // rows per half-wave per batch: 16*5*480 B = 38 KB in flight
    static_assert(RB == 5, "the staging batch below is written out for 5 rows");
    for (int j0 = 0; j0 < K2; j0 += 32) {
        const int j = j0 + l;
        const int jc = j < K2 ? j : K2 - 1;         // clamped: every lane issues a valid load
        for (int rb = 0; rb < R; rb += NG * RB) {
            double2 t0, t1, t2, t3, t4;
            auto row = [&](int i) {
                const int r = rb + grp + NG * i;
                const int rc = r < R ? r : R - 1;
                const int src = rc < nOwnE ? e0 + rc : m.haloEdge[h0 + rc - nOwnE];
                return pu2[(size_t)src * K2 + jc];
            };
            t0 = row(0); t1 = row(1); t2 = row(2); t3 = row(3); t4 = row(4);
            auto put = [&](int i, const double2 &v) {
                const int r = rb + grp + NG * i;
                if (r < R && j < K2) ubuf2[(size_t)r * K2 + j] = v;
            };
            put(0, t0); put(1, t1); put(2, t2); put(3, t3); put(4, t4);
        }
    }
    for (int r = tid; r < R; r += LBLOCK) fbuf[r] = m.fEdge[r < nOwnE ? e0 + r : m.haloEdge[h0 + r - nOwnE]];
    // patch records: contiguous ranges of the global record arrays -> straight coalesced copies
    for (int i = tid; i < nOwnE * ME2; i += LBLOCK) wbuf[i] = m.woe[(size_t)e0 * ME2 + i];
    for (int i = tid; i < nOwnE; i += LBLOCK) gbuf[i] = m.gInvDc[e0 + i];
    for (int i = tid; i < nOwnE * 4; i += LBLOCK) {
        hbuf[i] = m.ehdr[(size_t)e0 * 4 + i];
        lebuf[i] = reinterpret_cast<const uint32_t *>(m.leoe)[(size_t)e0 * 4 + i];
    }
    for (int i = tid; i < nOwnC * ME; i += LBLOCK) {
        sbuf[i] = m.sdv[(size_t)c0 * ME + i];
        cbuf[i] = m.coc[(size_t)c0 * ME + i];
        mbuf[i] = m.mltc[(size_t)c0 * ME + i];
    }
    for (int i = tid; i < nOwnC; i += LBLOCK) {
        iabuf[i] = m.invArea[c0 + i];
        rsbuf[i] = m.rsum[c0 + i];
    }
    for (int i = tid; i < nOwnC * 2; i += LBLOCK) lcbuf[i] = reinterpret_cast<const uint32_t *>(m.leoc)[(size_t)c0 * 2 + i];
    __syncthreads();

    const int K2c = (K2 + 31) & ~31;                // keep all 32 lanes in the loop for the shuffles
    // ---- 2a. cells ----
    const double2 *ph2 = reinterpret_cast<const double2 *>(a.ph);
    for (int ci = grp; ci < nOwnC; ci += NG) {
        const int c = c0 + ci;
        const double invA = iabuf[ci];
        int le[ME], cn[ME], ml[ME];
        double sd[ME];
#pragma unroll
        for (int i = 0; i < ME; ++i) {
            le[i] = (lcbuf[ci * 2 + (i >> 2)] >> (8 * (i & 3))) & 0xFF;
            cn[i] = cbuf[ci * ME + i];
            ml[i] = mbuf[ci * ME + i];
            sd[i] = sbuf[ci * ME + i];
        }
        double2 sshAcc = make_double2(0.0, 0.0);
        bool first = true;
        for (int j = l; j < K2c; j += 32) {
            const bool act = j < K2;
            const size_t off = (size_t)c * K2 + j;
            double2 hs = make_double2(0.0, 0.0);
            if (act) {
                const double2 hc = ph2[off];
                double2 hv[ME], uv[ME];
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    hv[i] = ph2[(size_t)(cn[i] >= 0 ? cn[i] : c) * K2 + j];
                    uv[i] = ubuf2[(size_t)(le[i] != 0xFF ? le[i] : 0) * K2 + j];
                }
                double2 t = make_double2(0.0, 0.0);
                const int k0 = 2 * j;
#pragma unroll
                for (int i = 0; i < ME; ++i) {
                    if (le[i] != 0xFF) {
                        if (k0 < ml[i]) t.x += uv[i].x * (0.5 * (hc.x + hv[i].x)) * sd[i] * invA;       // Operators.jl:217,
                        if (k0 + 1 < ml[i]) t.y += uv[i].y * (0.5 * (hc.y + hv[i].y)) * sd[i] * invA;   // DiagnosticVars.jl:165, horizontal_advection.jl:63
                    }
                }
                if (a.tendH) reinterpret_cast<double2 *>(a.tendH)[off] = t;
                const double2 hcur = a.ch ? reinterpret_cast<const double2 *>(a.ch)[off] : hc;
                if (a.ph_out) {
                    hs = make_double2(hcur.x + a.a * t.x, hcur.y + a.a * t.y);                          // time_integration.jl:125
                    reinterpret_cast<double2 *>(a.ph_out)[off] = hs;
                }
                if (a.nh_out) {
                    const double2 nb = a.nh_in ? reinterpret_cast<const double2 *>(a.nh_in)[off] : hcur;
                    const double2 hn = make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y);                // :135
                    reinterpret_cast<double2 *>(a.nh_out)[off] = hn;
                    if (!a.ph_out) hs = hn;
                }
            }
            sshAcc = first ? hs : make_double2(sshAcc.x + hs.x, sshAcc.y + hs.y);
            first = false;
        }
        if (a.ssh_out) {
            // oracle_ksum order: lane-xor 16,8,4,2,1 on (even, odd) levels == level-xor 32,...,2; then level-xor 1
#pragma unroll
            for (int s = 16; s >= 1; s >>= 1) {
                const double2 o = shfl_xor2(sshAcc, s);
                sshAcc = make_double2(sshAcc.x + o.x, sshAcc.y + o.y);
            }
            if (l == 0) a.ssh_out[c] = (sshAcc.x + sshAcc.y) - rsbuf[ci];                               // :209 (+N3)
        }
    }

    // ---- 2b. edges ----
    for (int ei = grp; ei < nOwnE; ei += NG) {
        const int e = e0 + ei;
        const int cA = hbuf[ei * 4], cB = hbuf[ei * 4 + 1], mlt = hbuf[ei * 4 + 3];
        const double g = gbuf[ei];
        const double ds = a.ssh[cB] - a.ssh[cA];                                                        // ssh[c2] - ssh[c1]
        uint32_t lw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lw[i] = lebuf[ei * 4 + i];
        for (int j = l; j < K2; j += 32) {
            const size_t off = (size_t)e * K2 + j;
            double2 t = make_double2(0.0, 0.0);
            const int k0 = 2 * j;
            const bool ax = k0 < mlt, ay = k0 + 1 < mlt;
            if (ax) t.x -= g * ds;                                                                      // pressure_gradient.jl:63
            if (ay) t.y -= g * ds;
#pragma unroll
            for (int i = 0; i < ME2; ++i) {
                const int le = (lw[i >> 2] >> (8 * (i & 3))) & 0xFF;
                if (le != 0xFF) {
                    const double2 uv = ubuf2[(size_t)le * K2 + j];
                    const double w = wbuf[ei * ME2 + i], f = fbuf[le];
                    if (ax) t.x += w * uv.x * f;                                                        // coriolis.jl:70-72
                    if (ay) t.y += w * uv.y * f;
                }
            }
            if (a.tendU) reinterpret_cast<double2 *>(a.tendU)[off] = t;
            const double2 up = ubuf2[(size_t)ei * K2 + j];                                              // own row = local row ei
            const double2 ucur = a.cu ? reinterpret_cast<const double2 *>(a.cu)[off] : up;
            if (a.pu_out) reinterpret_cast<double2 *>(a.pu_out)[off] = make_double2(ucur.x + a.a * t.x, ucur.y + a.a * t.y);
            if (a.nu_out) {
                const double2 nb = a.nu_in ? reinterpret_cast<const double2 *>(a.nu_in)[off] : ucur;
                reinterpret_cast<double2 *>(a.nu_out)[off] = make_double2(nb.x + a.b * t.x, nb.y + a.b * t.y);
            }
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

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int patch_grid(const MeshDev &m) { return 8 * ((m.nPatches + 7) / 8); }

template <int LPC>
static hipError_t launch_stage_lpc(const MeshDev &m, const StageArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(m)), b(BLOCK);
    if (m.ME == 6 && m.ME2 == 10) hipLaunchKernelGGL((k_stage<LPC, 6, 10>), g, b, 0, s, m, a);
    else if (m.ME == 8 && m.ME2 == 14) hipLaunchKernelGGL((k_stage<LPC, 8, 14>), g, b, 0, s, m, a);
    else if (m.ME <= 6 && m.ME2 <= 14) hipLaunchKernelGGL((k_stage<LPC, 6, 14>), g, b, 0, s, m, a);
    else return hipErrorInvalidValue;
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

hipError_t launch_stage(const MeshDev &m, const StageArgs &a, int lpc, hipStream_t s)
{
#define CALL(L) launch_stage_lpc<L>(m, a, s)
    DISPATCH_LPC(lpc, CALL)
#undef CALL
}

template <int ME, int ME2>
static bool launch_colp(const ColMesh &m, const StageArgs &a, int mode, dim3 g, dim3 b, hipStream_t s)
{
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_colp<ME, ME2, 0>), g, b, 0, s, m, a); return true;
        case 1: hipLaunchKernelGGL((k_stage_colp<ME, ME2, 1>), g, b, 0, s, m, a); return true;
        case 2: hipLaunchKernelGGL((k_stage_colp<ME, ME2, 2>), g, b, 0, s, m, a); return true;
        case 3: hipLaunchKernelGGL((k_stage_colp<ME, ME2, 3>), g, b, 0, s, m, a); return true;
    }
    return false;
}

template <int ME, int ME2, bool PIPE>
static bool launch_colx(const ColMesh &m, const StageArgs &a, int mode, dim3 g, dim3 b, hipStream_t s)
{
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_colx<ME, ME2, 0, PIPE>), g, b, 0, s, m, a); return true;
        case 1: hipLaunchKernelGGL((k_stage_colx<ME, ME2, 1, PIPE>), g, b, 0, s, m, a); return true;
        case 2: hipLaunchKernelGGL((k_stage_colx<ME, ME2, 2, PIPE>), g, b, 0, s, m, a); return true;
        case 3: hipLaunchKernelGGL((k_stage_colx<ME, ME2, 3, PIPE>), g, b, 0, s, m, a); return true;
    }
    return false;
}

// which pipelined specialisation serves this argument block (-1: none, use the plain column kernel)
static int colp_mode(const StageArgs &a)
{
    const bool outs = a.pu_out || a.ph_out || a.nu_out || a.nh_out;
    if (a.tendU && a.tendH && !outs && !a.ssh_out) return 0;
    if (a.tendU || a.tendH) return -1;
    if (!a.cu && !a.ch && !a.nu_in && !a.nh_in && a.pu_out && a.ph_out && a.nu_out && a.nh_out && a.ssh_out) return 1;
    if (a.cu && a.ch && a.nu_in && a.nh_in && a.pu_out && a.ph_out && a.nu_out && a.nh_out && a.ssh_out) return 2;
    if (a.nu_in && a.nh_in && !a.pu_out && !a.ph_out && a.nu_out && a.nh_out && a.ssh_out) return 3;
    return -1;
}

template <int ME, int ME2>
static bool launch_rec(const ColMesh &m, const StageArgs &a, int mode, dim3 g, dim3 b, size_t lds, int mE, int mC, hipStream_t s)
{
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_rec<ME, ME2, 0>), g, b, lds, s, m, a, mE, mC); return true;
        case 1: hipLaunchKernelGGL((k_stage_rec<ME, ME2, 1>), g, b, lds, s, m, a, mE, mC); return true;
        case 2: hipLaunchKernelGGL((k_stage_rec<ME, ME2, 2>), g, b, lds, s, m, a, mE, mC); return true;
        case 3: hipLaunchKernelGGL((k_stage_rec<ME, ME2, 3>), g, b, lds, s, m, a, mE, mC); return true;
    }
    return false;
}

static int colp_mode(const StageArgs &a);

template <int ME, int ME2>
static bool launch_rec2(const ColMesh &m, const StageArgs &a, int mode, dim3 g, dim3 b, size_t lds, int mE, int mC, hipStream_t s)
{
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_rec2<ME, ME2, 0>), g, b, lds, s, m, a, mE, mC); return true;
        case 1: hipLaunchKernelGGL((k_stage_rec2<ME, ME2, 1>), g, b, lds, s, m, a, mE, mC); return true;
        case 2: hipLaunchKernelGGL((k_stage_rec2<ME, ME2, 2>), g, b, lds, s, m, a, mE, mC); return true;
        case 3: hipLaunchKernelGGL((k_stage_rec2<ME, ME2, 3>), g, b, lds, s, m, a, mE, mC); return true;
    }
    return false;
}

size_t tile_lds_bytes(const MeshDev &md)
{
    return ((size_t)md.maxRows * md.K + (size_t)md.maxOwnE * (2 * md.ME2 + 2) + (size_t)md.maxOwnC * (md.ME + 2)) * 8 +
           ((size_t)md.maxOwnE * (4 + md.ME2) + (size_t)md.maxOwnC * (3 * md.ME)) * 4 + 16;
}

template <int ME, int ME2, int RB, int MC>
static bool launch_tile(const MeshDev &m, const StageArgs &a, int mode, dim3 g, dim3 b, size_t lds, hipStream_t s)
{
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_tile<ME, ME2, 0, RB, MC>), g, b, lds, s, m, a); return true;
        case 1: hipLaunchKernelGGL((k_stage_tile<ME, ME2, 1, RB, MC>), g, b, lds, s, m, a); return true;
        case 2: hipLaunchKernelGGL((k_stage_tile<ME, ME2, 2, RB, MC>), g, b, lds, s, m, a); return true;
        case 3: hipLaunchKernelGGL((k_stage_tile<ME, ME2, 3, RB, MC>), g, b, lds, s, m, a); return true;
    }
    return false;
}

template <int ME, int ME2, int RB, int MC>
static hipError_t prepare_tile(size_t lds)
{
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void *)k_stage_tile<ME, ME2, 0, RB, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return e;
    if ((e = hipFuncSetAttribute((const void *)k_stage_tile<ME, ME2, 1, RB, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return e;
    if ((e = hipFuncSetAttribute((const void *)k_stage_tile<ME, ME2, 2, RB, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return e;
    return hipFuncSetAttribute((const void *)k_stage_tile<ME, ME2, 3, RB, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

static bool tile_small(const MeshDev &md) { return md.maxRows <= 8 * 11 && md.maxOwnC <= 8; }

// usable when the patch-local row lists exist, K is even and <= 64, and a patch fits one of the static shapes
bool stage_tile_usable(const MeshDev &md, bool ldsOk)
{
    return ldsOk && md.K >= 2 && md.K <= 64 && !(md.K & 1) && md.maxRows <= 8 * 17 && md.maxOwnC <= 8 * 2 &&
           md.maxOwnE <= BLOCK && tile_lds_bytes(md) <= 160 * 1024 && md.ME == 6 && md.ME2 == 10;
}

hipError_t prepare_stage_tile(const MeshDev &md)
{
    const size_t lds = tile_lds_bytes(md);
    return tile_small(md) ? prepare_tile<6, 10, 11, 1>(lds) : prepare_tile<6, 10, 17, 2>(lds);
}

hipError_t launch_stage_tile(const MeshDev &md, const StageArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(md)), b(BLOCK);
    const int mode = colp_mode(a);
    if (mode < 0) return hipErrorNotSupported;
    const size_t lds = tile_lds_bytes(md);
    const bool ok = tile_small(md) ? launch_tile<6, 10, 11, 1>(md, a, mode, g, b, lds, s)
                                   : launch_tile<6, 10, 17, 2>(md, a, mode, g, b, lds, s);
    return ok ? hipGetLastError() : hipErrorNotSupported;
}

// ---- persistent tiled kernel ----
constexpr int PT_RB = 7, PT_EPG = 3;      // 16 groups x 7 rows = 112 rows, <= 16 cells, <= 48 own edges per patch (P <= 10)

bool stage_ptile_usable(const MeshDev &md, bool ldsOk)
{
    return ldsOk && md.K >= 2 && md.K <= 64 && !(md.K & 1) && md.ME == 6 && md.ME2 == 10 && md.maxRows <= 16 * PT_RB &&
           md.maxOwnC <= 16 && md.maxOwnE <= 16 * PT_EPG && 2 * ((tile_lds_bytes(md) + 255) & ~(size_t)255) <= 160 * 1024;
}

template <int MODE>
static hipError_t prepare_ptile_mode(size_t lds)
{
    return hipFuncSetAttribute((const void *)k_stage_ptile<6, 10, MODE, PT_RB, PT_EPG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t prepare_stage_ptile(const MeshDev &md)
{
    const size_t lds = 2 * ((tile_lds_bytes(md) + 255) & ~(size_t)255);
    hipError_t e;
    if ((e = prepare_ptile_mode<0>(lds))) return e;
    if ((e = prepare_ptile_mode<1>(lds))) return e;
    if ((e = prepare_ptile_mode<2>(lds))) return e;
    return prepare_ptile_mode<3>(lds);
}

hipError_t launch_stage_ptile(const MeshDev &md, const StageArgs &a, int nCUs, hipStream_t s)
{
    const int mode = colp_mode(a);
    if (mode < 0) return hipErrorNotSupported;
    const size_t buf = (tile_lds_bytes(md) + 255) & ~(size_t)255, lds = 2 * buf;
    int nb = nCUs > 0 ? nCUs : 256;
    if (nb > md.nPatches) nb = md.nPatches;
    nb = 8 * ((nb + 7) / 8);                                 // the XCD-chunk map wants a multiple of 8
    const int ppb = (md.nPatches + nb - 1) / nb;
    const dim3 g(nb), b(PBLOCK);
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_ptile<6, 10, 0, PT_RB, PT_EPG>), g, b, lds, s, md, a, ppb, buf); break;
        case 1: hipLaunchKernelGGL((k_stage_ptile<6, 10, 1, PT_RB, PT_EPG>), g, b, lds, s, md, a, ppb, buf); break;
        case 2: hipLaunchKernelGGL((k_stage_ptile<6, 10, 2, PT_RB, PT_EPG>), g, b, lds, s, md, a, ppb, buf); break;
        default: hipLaunchKernelGGL((k_stage_ptile<6, 10, 3, PT_RB, PT_EPG>), g, b, lds, s, md, a, ppb, buf); break;
    }
    return hipGetLastError();
}

size_t rec2c_lds_bytes(const MeshDev &md);

template <int ME, int ME2>
static bool launch_rec2c(const ColMesh &m, const StageArgs &a, int mode, dim3 g, dim3 b, size_t lds, int mE, int mC, hipStream_t s)
{
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 0>), g, b, lds, s, m, a, mE, mC); return true;
        case 1: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 1>), g, b, lds, s, m, a, mE, mC); return true;
        case 2: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 2>), g, b, lds, s, m, a, mE, mC); return true;
        case 3: hipLaunchKernelGGL((k_stage_rec2c<ME, ME2, 3>), g, b, lds, s, m, a, mE, mC); return true;
    }
    return false;
}

size_t rec_lds_bytes(const MeshDev &md)
{
    return (size_t)md.maxOwnE * (2 * md.ME2 + 1) * 8 + (size_t)md.maxOwnC * (md.ME + 2) * 8 +
           ((size_t)md.maxOwnE * md.EI + (size_t)md.maxOwnC * md.CI) * 4 + 16;
}

size_t rec2c_lds_bytes(const MeshDev &md)
{
    return ((rec_lds_bytes(md) + 15) & ~(size_t)15) + (size_t)md.maxOwnE * md.K * 8 + 16;
}

hipError_t launch_stage_rec2c(const MeshDev &md, const StageArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(md)), b(BLOCK);
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc};
    const int mode = colp_mode(a);
    const size_t lds = rec2c_lds_bytes(md);
    if (mode < 0 || md.K > 64 || (md.K & 1) || lds > 64 * 1024 || md.maxOwnC < 1 || md.maxOwnE < 1) return hipErrorNotSupported;
    bool ok = false;
    if (md.ME == 6 && md.ME2 == 10) ok = launch_rec2c<6, 10>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME == 8 && md.ME2 == 14) ok = launch_rec2c<8, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_rec2c<6, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    return ok ? hipGetLastError() : hipErrorNotSupported;
}

template <int ME, int ME2>
static bool launch_rec2c_f32(const ColMesh &m, const StageArgs &a, int mode, dim3 g, dim3 b, size_t lds, int mE, int mC, hipStream_t s)
{
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 0>), g, b, lds, s, m, a, mE, mC); return true;
        case 1: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 1>), g, b, lds, s, m, a, mE, mC); return true;
        case 2: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 2>), g, b, lds, s, m, a, mE, mC); return true;
        case 3: hipLaunchKernelGGL((k_stage_rec2c_f32<ME, ME2, 3>), g, b, lds, s, m, a, mE, mC); return true;
    }
    return false;
}

// fp32-state meshes: K % 4 == 0, K <= 128, byte-offset records, records + own u rows within 64 KB of LDS
bool stage_f32_supported(const MeshDev &md)
{
    const size_t lds = ((rec_lds_bytes(md) + 15) & ~(size_t)15) + (size_t)md.maxOwnE * md.K * 4 + 16;
    const bool shape = (md.ME == 6 && md.ME2 == 10) || (md.ME == 8 && md.ME2 == 14) || (md.ME <= 6 && md.ME2 <= 14);
    return md.cRec && md.eRec && md.K >= 4 && md.K <= 128 && (md.K & 3) == 0 && lds <= 64 * 1024 && md.maxOwnC >= 1 &&
           md.maxOwnE >= 1 && shape;
}

hipError_t launch_stage_rec2c_f32(const MeshDev &md, const StageArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(md)), b(BLOCK);
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc};
    const int mode = colp_mode(a);
    if (mode < 0 || !stage_f32_supported(md)) return hipErrorNotSupported;
    const size_t lds = ((rec_lds_bytes(md) + 15) & ~(size_t)15) + (size_t)md.maxOwnE * md.K * 4 + 16;
    bool ok = false;
    if (md.ME == 6 && md.ME2 == 10) ok = launch_rec2c_f32<6, 10>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME == 8 && md.ME2 == 14) ok = launch_rec2c_f32<8, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_rec2c_f32<6, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    return ok ? hipGetLastError() : hipErrorNotSupported;
}

hipError_t launch_stage_rec2(const MeshDev &md, const StageArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(md)), b(BLOCK);
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc};
    const int mode = colp_mode(a);
    const size_t lds = rec_lds_bytes(md);
    if (mode < 0 || md.K > 64 || (md.K & 1) || lds > 64 * 1024 || md.maxOwnC < 1 || md.maxOwnE < 1) return hipErrorNotSupported;
    bool ok = false;
    if (md.ME == 6 && md.ME2 == 10) ok = launch_rec2<6, 10>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME == 8 && md.ME2 == 14) ok = launch_rec2<8, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_rec2<6, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    return ok ? hipGetLastError() : hipErrorNotSupported;
}

hipError_t launch_stage_rec(const MeshDev &md, const StageArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(md)), b(BLOCK);
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc};
    const int mode = colp_mode(a);
    const size_t lds = rec_lds_bytes(md);
    if (mode < 0 || md.K > 64 || lds > 64 * 1024) return hipErrorNotSupported;
    bool ok = false;
    if (md.ME == 6 && md.ME2 == 10) ok = launch_rec<6, 10>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME == 8 && md.ME2 == 14) ok = launch_rec<8, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_rec<6, 14>(m, a, mode, g, b, lds, md.maxOwnE, md.maxOwnC, s);
    return ok ? hipGetLastError() : hipErrorNotSupported;
}

hipError_t launch_stage_colx(const MeshDev &md, const StageArgs &a, bool pipelined, hipStream_t s)
{
    const dim3 g(patch_grid(md)), b(BLOCK);
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc};
    const int mode = colp_mode(a);
    if (mode < 0 || md.K > 64 || (md.K & 1)) return hipErrorNotSupported;
    bool ok = false;
    if (pipelined) {
        if (md.ME == 6 && md.ME2 == 10) ok = launch_colx<6, 10, true>(m, a, mode, g, b, s);
        else if (md.ME == 8 && md.ME2 == 14) ok = launch_colx<8, 14, true>(m, a, mode, g, b, s);
        else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_colx<6, 14, true>(m, a, mode, g, b, s);
    } else {
        if (md.ME == 6 && md.ME2 == 10) ok = launch_colx<6, 10, false>(m, a, mode, g, b, s);
        else if (md.ME == 8 && md.ME2 == 14) ok = launch_colx<8, 14, false>(m, a, mode, g, b, s);
        else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_colx<6, 14, false>(m, a, mode, g, b, s);
    }
    return ok ? hipGetLastError() : hipErrorNotSupported;
}

hipError_t launch_stage_col(const MeshDev &md, const StageArgs &a, bool pipelined, hipStream_t s)
{
    const dim3 g(patch_grid(md)), b(BLOCK);
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc};
    const int mode = (pipelined && md.K <= 64) ? colp_mode(a) : -1;
    if (mode >= 0) {
        bool ok = false;
        if (md.ME == 6 && md.ME2 == 10) ok = launch_colp<6, 10>(m, a, mode, g, b, s);
        else if (md.ME == 8 && md.ME2 == 14) ok = launch_colp<8, 14>(m, a, mode, g, b, s);
        else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_colp<6, 14>(m, a, mode, g, b, s);
        if (ok) return hipGetLastError();
    }
    if (md.ME == 6 && md.ME2 == 10) hipLaunchKernelGGL((k_stage_col<6, 10>), g, b, 0, s, m, a);
    else if (md.ME == 8 && md.ME2 == 14) hipLaunchKernelGGL((k_stage_col<8, 14>), g, b, 0, s, m, a);
    else if (md.ME <= 6 && md.ME2 <= 14) hipLaunchKernelGGL((k_stage_col<6, 14>), g, b, 0, s, m, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_stage_lds(const MeshDev &m, const StageArgs &a, size_t ldsBytes, hipStream_t s)
{
    const dim3 g(patch_grid(m)), b(LBLOCK);
    if (m.ME == 6 && m.ME2 == 10) hipLaunchKernelGGL((k_stage_lds<6, 10>), g, b, ldsBytes, s, m, a);
    else if (m.ME == 8 && m.ME2 == 14) hipLaunchKernelGGL((k_stage_lds<8, 14>), g, b, ldsBytes, s, m, a);
    else if (m.ME <= 6 && m.ME2 <= 14) hipLaunchKernelGGL((k_stage_lds<6, 14>), g, b, ldsBytes, s, m, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t prepare_stage_lds(size_t ldsBytes)
{
    // > 64 KB of dynamic LDS needs the opt-in attribute
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void *)k_stage_lds<6, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes))) return e;
    if ((e = hipFuncSetAttribute((const void *)k_stage_lds<8, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes))) return e;
    if ((e = hipFuncSetAttribute((const void *)k_stage_lds<6, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes))) return e;
    return hipSuccess;
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
