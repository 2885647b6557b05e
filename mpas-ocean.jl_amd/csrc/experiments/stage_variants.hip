// stage_variants.hip -- EXPERIMENTS, not part of the product build (make VARIANTS=1 adds them): the other execution
// shapes of the fused tendency / RK-stage kernel measured in round 1 (profiles/r01_variants.txt) -- the pipelined column
// kernels, the record-staged kernels "rec" / "rec2", and the three LDS-staging designs ("lds", "tile", "ptile").  Same
// arithmetic and bit-identical results as the default kernel k_stage_rec2c (kernels.hip); selectable with
// moka_set_kernel_variant when built, and then exercised by the parity tests.
#include "../kernels_common.hpp"

namespace moka {

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
#pragma unroll
    for (int i = 0; i < ME; ++i) {
        b.uv[i] = gload(a.pu, r[i] + voff);
        b.hv[i] = gload(a.ph, r[ME + i] + voff);
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
#pragma unroll
    for (int i = 0; i < ME2; ++i) b.uv[i] = gload(a.pu, r[i] + voff);
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
    const int pl_ = patch_of_block(m.nPatches);      // m.nPatches = patches in this launch
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

hipError_t launch_stage_colp(const MeshDev &md, const StageArgs &a, hipStream_t s)
{
    const dim3 g(patch_grid(md)), b(BLOCK);
    const ColMesh m{md.nC, md.nE, md.K, md.nPatches, md.patchBegin, md.CI, md.EI, md.patchCellStart, md.patchEdgeStart,
                    md.cRec, md.eRec, md.mltc, md.sdv, md.invArea, md.rsum, md.woe, md.feoe, md.gInvDc};
    const int mode = md.K <= 64 ? colp_mode(a) : -1;
    if (mode < 0) return hipErrorNotSupported;
    bool ok = false;
    if (md.ME == 6 && md.ME2 == 10) ok = launch_colp<6, 10>(m, a, mode, g, b, s);
    else if (md.ME == 8 && md.ME2 == 14) ok = launch_colp<8, 14>(m, a, mode, g, b, s);
    else if (md.ME <= 6 && md.ME2 <= 14) ok = launch_colp<6, 14>(m, a, mode, g, b, s);
    return ok ? hipGetLastError() : hipErrorNotSupported;
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

}  // namespace moka

