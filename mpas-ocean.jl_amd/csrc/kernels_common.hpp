// kernels_common.hpp -- device helpers and launch macros shared by the kernel translation units of libmoka_hip
// (kernels.hip: stage / Forward-Euler / utility kernels; nonlinear.hip; adjoint.hip).
#pragma once
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
