// inspect.hip -- moka_state_download_rows: selected rows of a state field, in the caller's numbering, widened to double.
// What a row-sampled check of a full-size run needs (tests/test_gpu_configs.py: the oracle evaluates ~10 000 sampled cells and
// edges of the 3.7 M-cell x 80-level fp32-storage state from their gathered neighbour rows); also a cheap way for a caller to
// look at a few columns without moving 7 GB across PCIe.  time_level 2 / 3 name the two RK4 provisional states (inspection:
// what the last stage launches left there).
#include <algorithm>

#include "state.hpp"

using namespace mk;

namespace {

template <class T>
__global__ __launch_bounds__(256) void k_gather_rows(double *dst, const T *src, const int32_t *rows, int64_t n, int K)
{
    const int64_t total = n * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / K;
        const int k = (int)(i - r * K);
        dst[i] = (double)src[(int64_t)rows[r] * K + k];
    }
}

}  // namespace

extern "C" int moka_state_download_rows(moka_state *st, int field, int time_level, int64_t nRows, const int32_t *rows, double *host)
{
    if (!st || (nRows > 0 && (!rows || !host)) || nRows < 0) return fail(st ? st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument or nRows < 0");
    moka_ctx *c = st->ctx;
    const Plan &p = st->mesh->plan;
    HIPCHK(c, hipSetDevice(c->device));
    const bool diag = field >= MOKA_F_LAYER_THICKNESS_EDGE && field <= MOKA_F_RELATIVE_VORTICITY;
    const bool tend = field == MOKA_F_TEND_NORMAL_VELOCITY || field == MOKA_F_TEND_LAYER_THICKNESS;
    if (int rc = flush_lazy(st, diag, tend)) return rc;          // lazily pending arrays are produced on a read
    const double *src = nullptr;
    int kind = MOKA_CELL, K = p.K;
    if (time_level == 2 || time_level == 3) {
        const LevelBufs &b = st->rk[time_level - 2];
        if (!b.ssh) return fail(c, MOKA_ERR_ARG, "the RK4 provisional states exist after the first RK4 step only");
        switch (field) {
            case MOKA_F_SSH: src = b.ssh; K = 1; break;
            case MOKA_F_NORMAL_VELOCITY: src = b.u; kind = MOKA_EDGE; break;
            case MOKA_F_LAYER_THICKNESS: src = b.h; break;
            default: return fail(c, MOKA_ERR_ARG, "time_level 2 / 3 (RK4 provisional states): prognostic fields only");
        }
    } else if (time_level == 0 || time_level == 1) {
        switch (field) {
            case MOKA_F_SSH: src = st->lev[time_level].ssh; K = 1; break;
            case MOKA_F_NORMAL_VELOCITY: src = st->lev[time_level].u; kind = MOKA_EDGE; break;
            case MOKA_F_LAYER_THICKNESS: src = st->lev[time_level].h; break;
            case MOKA_F_LAYER_THICKNESS_EDGE: src = st->hEdge[0]; kind = MOKA_EDGE; break;
            case MOKA_F_THICKNESS_FLUX: src = st->F; kind = MOKA_EDGE; break;
            case MOKA_F_VELOCITY_DIV_CELL: src = st->div; break;
            case MOKA_F_RELATIVE_VORTICITY: src = st->vort; kind = MOKA_VERTEX; break;
            case MOKA_F_TEND_NORMAL_VELOCITY: src = st->tendU; kind = MOKA_EDGE; break;
            case MOKA_F_TEND_LAYER_THICKNESS: src = st->tendH; break;
            default: return fail(c, MOKA_ERR_ARG, "unknown field id");
        }
    } else {
        return fail(c, MOKA_ERR_ARG, "time_level must be 0 (previous), 1 (current), 2 or 3 (RK4 provisional states)");
    }
    if (!src) return fail(c, MOKA_ERR_ARG, "field not allocated");
    if (nRows == 0) return MOKA_OK;
    const int64_t nEnt = kind == MOKA_CELL ? p.nC : kind == MOKA_EDGE ? p.nE : p.nV;
    const std::vector<int32_t> &o2n = kind == MOKA_CELL ? p.cellO2N : kind == MOKA_EDGE ? p.edgeO2N : p.vertO2N;
    std::vector<int32_t> lib(nRows);
    for (int64_t i = 0; i < nRows; ++i) {
        if (rows[i] < 0 || rows[i] >= nEnt) return fail(c, MOKA_ERR_ARG, "row id out of range");
        lib[i] = o2n[rows[i]];
    }
    int32_t *dRows = nullptr;
    double *dOut = nullptr;
    auto done = [&](int rc) {
        if (dRows) (void)hipFree(dRows);
        if (dOut) (void)hipFree(dOut);
        return rc;
    };
    if (hipMalloc((void **)&dRows, (size_t)nRows * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void **)&dOut, (size_t)nRows * K * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        return done(fail(c, MOKA_ERR_ALLOC, "moka_state_download_rows: hipMalloc failed"));
    }
    hipStream_t s = c->stream;
    hipError_t e = hipMemcpyAsync(dRows, lib.data(), (size_t)nRows * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        const unsigned blocks = (unsigned)std::min<int64_t>((nRows * K + 255) / 256, 16384);
        if (st->f32) hipLaunchKernelGGL(k_gather_rows<float>, dim3(blocks), dim3(256), 0, s, dOut, reinterpret_cast<const float *>(src), dRows, nRows, K);
        else hipLaunchKernelGGL(k_gather_rows<double>, dim3(blocks), dim3(256), 0, s, dOut, src, dRows, nRows, K);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, dOut, (size_t)nRows * K * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return done(fail(c, MOKA_ERR_HIP, std::string("moka_state_download_rows: ") + hipGetErrorString(e)));
    return done(MOKA_OK);
}

// Device address of a prognostic array (inspection: tools/placement_addresses.py looks for what distinguishes the allocations the
// placement search keeps from the ones it drops).  time_level as moka_state_download_rows.
extern "C" int moka_state_array_address(moka_state *st, int field, int time_level, uint64_t *address)
{
    if (!st || !address) return fail(st ? st->ctx : nullptr, MOKA_ERR_ARG, "NULL argument");
    if (time_level < 0 || time_level > 3 || field < MOKA_F_SSH || field > MOKA_F_LAYER_THICKNESS)
        return fail(st->ctx, MOKA_ERR_ARG, "prognostic fields, time_level 0..3");
    const LevelBufs &b = time_level < 2 ? st->lev[time_level] : st->rk[time_level - 2];
    const double *q = field == MOKA_F_SSH ? b.ssh : field == MOKA_F_NORMAL_VELOCITY ? b.u : b.h;
    *address = (uint64_t)(uintptr_t)q;
    return MOKA_OK;
}
