// kernels_common.hpp -- device helpers and launch macros shared by the kernel translation units of libmoka_hip
// (kernels.hip: stage / Forward-Euler / utility kernels; nonlinear.hip; adjoint.hip).
#pragma once
#include <atomic>
#include <algorithm>
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace moka {

constexpr int BLOCK = 256;

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
template <int LPC>
__device__ __forceinline__ double group_sum(double v)
{
    // XOR butterfly over the LPC lanes of a group: the summation order fixed by the oracle
    // (oracle_ksum).  fp add is commutative, so every lane ends with the same bits.
#pragma unroll
    for (int s = LPC / 2; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
    return v;
}

// Mesh records and other kernel-invariant data are read through the constant address space: with a
// wave-uniform address (LPC = 64) the compiler then emits scalar loads (s_load_dwordx4/x8/x16 into SGPRs)
// instead of 64 identical vector loads, which frees the vector memory pipe and ~50 VGPRs per lane.
template <class T> using CP = const T __attribute__((address_space(4))) *;
template <class T> __device__ __forceinline__ CP<T> cptr(const T *p) { return (CP<T>)(uintptr_t)p; }

template <int LPC>
__device__ __forceinline__ int uniform_if_wave(int x)
{
    if constexpr (LPC == 64) return __builtin_amdgcn_readfirstlane(x);
    else return x;
}

// blockIdx -> patch: XCD x (= blockIdx % 8 by the observed round-robin dispatch; speed only)
// walks the contiguous patch range [x*chunk, (x+1)*chunk).
__device__ __forceinline__ int patch_of_block(int nPatches)
{
    const int chunk = (nPatches + 7) >> 3;
    return (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
}

// ---- small device helpers shared by the stage kernels ----
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

__device__ __forceinline__ double gload(const double *base, uint32_t off)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + off);
}

__device__ __forceinline__ void gstore(double *base, uint32_t off, double x)
{
    *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + off) = x;
}

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

// ---- explicit address spaces and the LDS read burst of the record-staged kernels (k_stage_rec2c*, k_stage_tile3) ----
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
__device__ __forceinline__ void lds_burst<3>(v4u_t (&v)[3], const uint32_t (&ad)[3])
{
    asm volatile("ds_read_b128 %[o0], %[a0]\n\t"
                 "ds_read_b128 %[o1], %[a1]\n\t"
                 "ds_read_b128 %[o2], %[a2]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [o0] "=&v"(v[0]), [o1] "=&v"(v[1]), [o2] "=&v"(v[2])
                 : [a0] "v"(ad[0]), [a1] "v"(ad[1]), [a2] "v"(ad[2])
                 : "memory");
}
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

// Result stores of the default stage kernels.  Experiment MOKA_EXP_WT_STORES: write-through (sc1) stores, which do not
// keep the line in the XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour"): results are never re-read inside a
// launch, so the L2 would hold gathered rows instead.
__device__ __forceinline__ void gstore2o(double *base, uint32_t off, double2 x)
{
#if defined(MOKA_EXP_WT_STORES)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), make_rsrc(base, 0xFFFFFFFFu), (int)off, 0, 16);
#elif defined(MOKA_EXP_NT_STORES)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), make_rsrc(base, 0xFFFFFFFFu), (int)off, 0, 2);
#elif defined(MOKA_EXP_NO_STORES)
    if (x.x == 1.2345e300) gstore2(base, off, x);      // ablation (wrong results): what do the stores cost / displace?
#else
    gstore2(base, off, x);
#endif
}

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

// ---- host-side helpers of the launchers ----
// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int patch_grid(const MeshDev &m) { return 8 * ((m.nPatches + 7) / 8); }

// which pipelined specialisation serves this argument block (-1: none, use the plain column kernel)
inline int colp_mode(const StageArgs &a)
{
    if (a.feMode) {       // Forward-Euler launch (k_stage_rec2c and k_stage_rec2c_f32): output groups are optional, as groups
        const bool lvl = a.pu_out && a.ph_out && a.ssh_out, noLvl = !a.pu_out && !a.ph_out && !a.ssh_out;
        const bool dia = a.tendU && a.tendH && a.F && a.div && a.hEdgeNew && a.areaCell, noDia = !a.tendU && !a.tendH && !a.F && !a.div && !a.hEdgeNew;
        const bool ok = (lvl || noLvl) && (dia || noDia) && (lvl || dia) && !a.cu && !a.ch && !a.nu_in && !a.nh_in && !a.nu_out && !a.nh_out &&
                        (a.feMode == 4 ? a.hEdgeOld != nullptr : a.feMode == 6 ? a.hPrev != nullptr : a.feMode == 5);
        return ok ? a.feMode : -1;
    }
    if (a.rkMode) {       // 13-stream RK4 form (k_stage_rec2c only)
        const bool p = a.pu_out && a.ph_out && a.ssh_out, c = a.cu && a.ch, n = a.nu_out && a.nh_out && a.ssh_out;
        if (a.rkMode == 7) return p && !c ? 7 : -1;
        if (a.rkMode == 8) return p && c ? 8 : -1;
        if (a.rkMode == 9) return n && c && a.nu_in && a.nh_in && a.q3u && a.q3h ? 9 : -1;
        return -1;
    }
    const bool outs = a.pu_out || a.ph_out || a.nu_out || a.nh_out;
    if (a.tendU && a.tendH && !outs && !a.ssh_out) return 0;
    if (a.tendU || a.tendH) return -1;
    if (!a.cu && !a.ch && !a.nu_in && !a.nh_in && a.pu_out && a.ph_out && a.nu_out && a.nh_out && a.ssh_out) return 1;
    if (a.cu && a.ch && a.nu_in && a.nh_in && a.pu_out && a.ph_out && a.nu_out && a.nh_out && a.ssh_out) return 2;
    if (a.nu_in && a.nh_in && !a.pu_out && !a.ph_out && a.nu_out && a.nh_out && a.ssh_out) return 3;
    return -1;
}

// a Forward-Euler launch of mode 5 / 6 that stores the new level (and, separately selected, relativeVorticity) and none of the
// optional DiagnosticVars / TendencyVars outputs: a LEAN step's launch, served by the kernels' modes 10 / 11
inline bool colp_lean(const StageArgs &a, int mode)
{
    return (mode == 5 || mode == 6) && a.pu_out && a.ph_out && a.ssh_out && !a.tendU && !a.tendH && !a.F && !a.div && !a.hEdgeNew;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel: remember per device (a process may
// drive several, e.g. LocalCluster over a device list) whether `slot` (one bit per kernel family) has been raised there
inline bool lds_attr_needed(int slot)
{
    static std::atomic<uint32_t> done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return true;
    const uint32_t bit = 1u << slot;
    return !(done[dev].fetch_or(bit) & bit);
}

inline size_t rec_lds_bytes(const MeshDev &md)
{
    return (size_t)md.maxOwnE * (2 * md.ME2 + 1) * 8 + (size_t)md.maxOwnC * (md.ME + 2) * 8 +
           ((size_t)md.maxOwnE * md.EI + (size_t)md.maxOwnC * md.CI) * 4 + 16;
}


// New of the 13-stream RK4 form from the own values of Curr and the three provisional states (P2 = C + dt/2 k1, P3 = C + dt/2 k2,
// P4 = C + dt k3) and the last tendency:  C + ((P2 - C) + 2 (P3 - C) + (P4 - C)) / 3 + dt/6 k4  -- the reference's
// C + dt/6 k1 + dt/3 k2 + dt/3 k3 + dt/6 k4 (time_integration.jl:78,134-135) up to round-off, NOT bit for bit: opt-in, with its own
// oracle twins (oracle_step_rk4_s13, oracle_step_rk4_nonlinear_s13: the same expression) and a tolerance test against the
// reference form.
__device__ __forceinline__ double rk13_combine(double c, double p2, double p3, double p4, double b4, double t)
{
    const double d2 = p2 - c, d3 = p3 - c, d4 = p4 - c;
    const double acc = (d2 + (d3 + d3)) + d4;
    return (c + acc * (1.0 / 3.0)) + b4 * t;
}

// lanes-per-column dispatch of the generic column kernels (LPC = smallest power of two >= nVertLevels, capped at 64)
#define DISPATCH_LPC(lpc, CALL)                 \
    switch (lpc) {                              \
        case 1: return CALL(1);                 \
        case 2: return CALL(2);                 \
        case 4: return CALL(4);                 \
        case 8: return CALL(8);                 \
        case 16: return CALL(16);               \
        case 32: return CALL(32);               \
        default: return CALL(64);               \
    }

}  // namespace moka
