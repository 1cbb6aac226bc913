// lec_diag.hip -- 850-hPa track diagnostics of the moving framework on the device (gfx950, wave64).
//
// Per time step: relative vorticity and wind speed on the 850-hPa slice, and inside that step's box the vorticity minimum and
// maximum, the height minimum and the wind maximum with their grid positions (lec_moving_framework.py:269-417,650-663;
// tools.py:95-128).  O(box points) per step: one workgroup per time step; not a bandwidth kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"

namespace {

constexpr int kThreads = 256;

struct DiagParams {
    const double* u; const double* v; const double* h;
    int nt, ny, nx;
    const int* box;
    const double* xcoef;    // [ny][nx][3]
    const double* ycoef;    // [ny][3]
    const double* curv;     // [ny]
    double* val;
    int* pos;
};

// running extremum with numpy's tie rule: the first in row-major order (lowest n) among equal values; NaN never enters
struct Best {
    double v; int n;
    __device__ __forceinline__ void take_min(double x, int m) { if (x < v || (x == v && m < n)) { v = x; n = m; } }
    __device__ __forceinline__ void take_max(double x, int m) { if (x > v || (x == v && m < n)) { v = x; n = m; } }
};

// zeta = dv/dx - du/dy + curv u with the three-point stencils of metpy.calc.first_derivative: the parabola through the point and its
// two neighbours (the three points nearest the edge at either end of the slice), coefficients in 1/m from the host's tables -- which
// metric the distances follow (and whether the sphere's curvature term is there) is the caller's choice, not the kernel's
__device__ __forceinline__ double zeta_at(const DiagParams& p, const double* u, const double* v, int j, int i) {
    const int i0 = min(max(i - 1, 0), p.nx - 3), j0 = min(max(j - 1, 0), p.ny - 3);
    const double* cx = p.xcoef + 3 * ((size_t)j * p.nx + i);
    const double* cy = p.ycoef + 3 * (size_t)j;
    const double* vr = v + (size_t)j * p.nx + i0;
    const double dv = cx[0] * vr[0] + cx[1] * vr[1] + cx[2] * vr[2];
    const double* uc = u + (size_t)j0 * p.nx + i;
    const double du = cy[0] * uc[0] + cy[1] * uc[p.nx] + cy[2] * uc[2 * (size_t)p.nx];
    return dv - du + p.curv[j] * u[(size_t)j * p.nx + i];
}

// grid nt, block kThreads
__global__ void __launch_bounds__(kThreads) lec_diag_kernel(const DiagParams p) {
    __shared__ double sv[4][kThreads];
    __shared__ int sn[4][kThreads];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int* bx = p.box + 6 * (size_t)t;
    // the box table lives in device memory (the library cannot read it at launch): clamp, so that no table can index past the slice
    const int iw = min(max(bx[0], 0), p.nx - 1), ie = min(max(bx[1], iw), p.nx - 1);
    const int js = min(max(bx[2], 0), p.ny - 1), jn = min(max(bx[3], js), p.ny - 1);
    const int jc = min(max(bx[4], 0), p.ny - 1), ic = min(max(bx[5], 0), p.nx - 1);
    const int nxb = ie - iw + 1, npt = nxb * (jn - js + 1);
    const size_t plane = (size_t)p.ny * p.nx;
    const double* u = p.u + t * plane;
    const double* v = p.v + t * plane;
    const double* h = p.h + t * plane;
    const double inf = __builtin_huge_val();
    const int none = 0x7fffffff;
    Best zmin{inf, none}, zmax{-inf, none}, hmin{inf, none}, wmax{-inf, none};
    for (int n = tid; n < npt; n += kThreads) {
        const int j = js + n / nxb, i = iw + n % nxb;
        const size_t e = (size_t)j * p.nx + i;
        const double z = zeta_at(p, u, v, j, i);
        if (z == z) { zmin.take_min(z, n); zmax.take_max(z, n); }
        const double hh = h[e];
        if (hh == hh) hmin.take_min(hh, n);
        const double w = sqrt(u[e] * u[e] + v[e] * v[e]);
        if (w == w) wmax.take_max(w, n);
    }
    sv[0][tid] = zmin.v; sn[0][tid] = zmin.n; sv[1][tid] = zmax.v; sn[1][tid] = zmax.n;
    sv[2][tid] = hmin.v; sn[2][tid] = hmin.n; sv[3][tid] = wmax.v; sn[3][tid] = wmax.n;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Best a{sv[q][tid], sn[q][tid]};
                if (q == 0 || q == 2) a.take_min(sv[q][tid + s], sn[q][tid + s]); else a.take_max(sv[q][tid + s], sn[q][tid + s]);
                sv[q][tid] = a.v; sn[q][tid] = a.n;
            }
        }
        __syncthreads();
    }
    if (tid < 4) {
        const int n = sn[tid][0];
        const bool found = n != none;
        p.val[5 * (size_t)t + tid] = found ? sv[tid][0] : nan("");
        p.pos[8 * (size_t)t + 2 * tid + 0] = found ? js + n / nxb : -1;
        p.pos[8 * (size_t)t + 2 * tid + 1] = found ? iw + n % nxb : -1;
    }
    if (tid == 4) p.val[5 * (size_t)t + 4] = zeta_at(p, u, v, jc, ic);
}

}  // namespace

extern "C" int lec_track_diag(const lec_diag_args* a) {
    if (!a) return lec_set_error(LEC_ERR_ARG, "lec_track_diag: null args");
    if (!a->u_d || !a->v_d || !a->hgt_d || !a->box_d || !a->xcoef_d || !a->ycoef_d || !a->curv_d || !a->val_d || !a->pos_d)
        return lec_set_error(LEC_ERR_ARG, "lec_track_diag: null pointer argument");
    if (a->nt < 1 || a->ny < 3 || a->nx < 3) return lec_set_error(LEC_ERR_ARG, "lec_track_diag: needs nt >= 1 and at least 3 x 3 grid points");
    if (a->reserved0 != 0) return lec_set_error(LEC_ERR_ARG, "lec_track_diag: reserved0 must be 0");
    if ((unsigned long long)a->ny * (unsigned long long)a->nx > 0x7fffffffULL) return lec_set_error(LEC_ERR_UNSUPPORTED, "lec_track_diag: slice too large");
    DiagParams p;
    p.u = a->u_d; p.v = a->v_d; p.h = a->hgt_d; p.nt = a->nt; p.ny = a->ny; p.nx = a->nx;
    p.box = a->box_d; p.xcoef = a->xcoef_d; p.ycoef = a->ycoef_d; p.curv = a->curv_d; p.val = a->val_d; p.pos = a->pos_d;
    hipStream_t st = (hipStream_t)a->stream;
    hipLaunchKernelGGL(lec_diag_kernel, dim3(a->nt), dim3(kThreads), 0, st, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    return LEC_OK;
}
