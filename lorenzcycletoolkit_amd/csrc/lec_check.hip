// lec_check.hip -- validation of the index tables that live in device memory (include/lec_hip.h: lec_check_boxes, lec_check_maps).
//
// The reference validates its box on the host (lec_fixed_framework.py:98-154) and so does this package's Python host
// (tables.box_indices, build_box_tables); a plain-C caller that fills box_d / the ingest maps itself gets the same guarantee from
// these two entry points: one small kernel scans the table, the call waits for it and reports the first offending entry.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"

namespace {

// status[0] = number of bad entries, status[1] = lowest bad entry index (or INT_MAX), status[2] = which table (maps: 0 k, 1 j, 2 i)
__global__ void __launch_bounds__(256) lec_check_boxes_kernel(const int* __restrict__ box, int n_box, int nx, int ny, int nxb_max, int nyb_max,
                                                              int* __restrict__ status) {
    int bad = 0, first = 0x7fffffff;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < n_box; b += gridDim.x * blockDim.x) {
        const int iw = box[4 * b + 0], ie = box[4 * b + 1], js = box[4 * b + 2], jn = box[4 * b + 3];
        const bool ok = iw >= 0 && ie < nx && js >= 0 && jn < ny && ie >= iw + 1 && jn >= js + 1 &&      // at least 2 x 2 points
                        ie - iw + 1 <= nxb_max && jn - js + 1 <= nyb_max;
        if (!ok) { ++bad; first = min(first, b); }
    }
    if (bad) { atomicAdd(&status[0], bad); atomicMin(&status[1], first); }
}

__global__ void __launch_bounds__(256) lec_check_maps_kernel(const int* __restrict__ kmap, int nl, int nl_in, const int* __restrict__ jmap, int ny,
                                                             int ny_in, const int* __restrict__ imap, int nx, int nx_in, int* __restrict__ status) {
    const int n = nl + ny + nx;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        int v, lim, which, at;
        if (e < nl) { v = kmap[e]; lim = nl_in; which = 0; at = e; }
        else if (e < nl + ny) { v = jmap[e - nl]; lim = ny_in; which = 1; at = e - nl; }
        else { v = imap[e - nl - ny]; lim = nx_in; which = 2; at = e - nl - ny; }
        if (v < 0 || v >= lim) {
            atomicAdd(&status[0], 1);
            const int old = atomicMin(&status[1], which * 0x1000000 + min(at, 0xffffff));      // lowest (table, entry)
            (void)old;
        }
    }
}

// the per-step gather table of a box-packed series (lec_ingest_args.step_d): {source step, latitude offset, longitude offset} per
// output step.  status[0] += bad entries, status[1] = lowest (table 3, step)
__global__ void __launch_bounds__(256) lec_check_steps_kernel(const int* __restrict__ step, int nt, int step_base, int nt_src, int ny, int jmap_len, int nx,
                                                              int imap_len, int* __restrict__ status) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += gridDim.x * blockDim.x) {
        const int ts = step[3 * t] - step_base, oj = step[3 * t + 1], oi = step[3 * t + 2];
        if ((unsigned)ts >= (unsigned)nt_src || oj < 0 || oj > jmap_len - ny || oi < 0 || oi > imap_len - nx) {
            atomicAdd(&status[0], 1);
            atomicMin(&status[1], 3 * 0x1000000 + min(t, 0xffffff));
        }
    }
}

int fetch_status(int32_t* status_d, int (&h)[4], hipStream_t st) {
    if (hipMemcpyAsync(h, status_d, sizeof(int) * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    return 0;
}

int reset_status(int32_t* status_d, hipStream_t st) {
    const int init[4] = {0, 0x7fffffff, 0, 0};
    return hipMemcpyAsync(status_d, init, sizeof(init), hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess ? 0 : 1;
}

}  // namespace

extern "C" int lec_check_boxes(const lec_rowstats_args* a, int32_t* status_d) {
    if (!a || !status_d) return lec_set_error(LEC_ERR_ARG, "lec_check_boxes: null args / status_d");
    if (!a->box_d) return lec_set_error(LEC_ERR_ARG, "lec_check_boxes: null box_d");
    if (a->n_box < 1 || a->nx < 2 || a->ny < 2 || a->nxb_max < 2 || a->nyb_max < 2) return lec_set_error(LEC_ERR_ARG, "lec_check_boxes: n_box >= 1 and extents >= 2 needed");
    hipStream_t st = (hipStream_t)a->stream;
    if (reset_status(status_d, st)) return lec_set_error(LEC_ERR_LAUNCH, "lec_check_boxes: could not initialise status_d");
    const int blocks = a->n_box < 256 * 64 ? (a->n_box + 255) / 256 : 64;
    hipLaunchKernelGGL(lec_check_boxes_kernel, dim3(blocks), dim3(256), 0, st, a->box_d, a->n_box, a->nx, a->ny, a->nxb_max, a->nyb_max, status_d);
    int h[4];
    if (hipGetLastError() != hipSuccess || fetch_status(status_d, h, st)) return lec_set_error(LEC_ERR_LAUNCH, "lec_check_boxes: kernel or copy failed");
    if (h[0] == 0) return LEC_OK;
    char msg[256];
    snprintf(msg, sizeof msg, "lec_check_boxes: %d of %d boxes are outside the %d x %d grid, smaller than 2 x 2 points or larger than nxb_max x nyb_max = %d x %d; "
             "first: box %d", h[0], a->n_box, a->nx, a->ny, a->nxb_max, a->nyb_max, h[1]);
    return lec_set_error(LEC_ERR_ARG, msg);
}

extern "C" int lec_check_maps(const lec_ingest_args* a, int32_t* status_d) {
    if (!a || !status_d) return lec_set_error(LEC_ERR_ARG, "lec_check_maps: null args / status_d");
    if (!a->kmap_d || !a->jmap_d || !a->imap_d) return lec_set_error(LEC_ERR_ARG, "lec_check_maps: null map pointer");
    if (a->nl < 1 || a->ny < 1 || a->nx < 1 || a->nl_in < 1 || a->ny_in < 1 || a->nx_in < 1) return lec_set_error(LEC_ERR_ARG, "lec_check_maps: extents must be >= 1");
    // the maps are scanned over their full lengths (a per-step gather enters them at an offset: they are longer than ny / nx then)
    const int nj = a->jmap_len > 0 ? a->jmap_len : a->ny, ni = a->imap_len > 0 ? a->imap_len : a->nx;
    if (nj < a->ny || ni < a->nx) return lec_set_error(LEC_ERR_ARG, "lec_check_maps: jmap_len / imap_len must be 0 or >= ny / nx");
    if (a->step_d && (a->nt < 1 || a->nt_src < 1)) return lec_set_error(LEC_ERR_ARG, "lec_check_maps: a step_d table needs nt >= 1 and nt_src >= 1");
    hipStream_t st = (hipStream_t)a->stream;
    if (reset_status(status_d, st)) return lec_set_error(LEC_ERR_LAUNCH, "lec_check_maps: could not initialise status_d");
    const long long n = (long long)a->nl + nj + ni;
    const int blocks = n < 256 * 64 ? (int)((n + 255) / 256) : 64;
    hipLaunchKernelGGL(lec_check_maps_kernel, dim3(blocks), dim3(256), 0, st, a->kmap_d, a->nl, a->nl_in, a->jmap_d, nj, a->ny_in, a->imap_d, ni,
                       a->nx_in, status_d);
    if (a->step_d) {
        const int sb = a->nt < 256 * 64 ? (a->nt + 255) / 256 : 64;
        hipLaunchKernelGGL(lec_check_steps_kernel, dim3(sb), dim3(256), 0, st, a->step_d, a->nt, a->step_base, a->nt_src, a->ny, nj, a->nx, ni, status_d);
    }
    int h[4];
    if (hipGetLastError() != hipSuccess || fetch_status(status_d, h, st)) return lec_set_error(LEC_ERR_LAUNCH, "lec_check_maps: kernel or copy failed");
    if (h[0] == 0) return LEC_OK;
    static const char* names[4] = {"kmap_d", "jmap_d", "imap_d", "step_d"};
    const int which = (h[1] >> 24) & 3;
    char msg[384];
    if (which < 3)
        snprintf(msg, sizeof msg, "lec_check_maps: %d table entries are out of range; first: %s[%d] points outside the source cube (%d x %d x %d)", h[0],
                 names[which], h[1] & 0xffffff, a->nl_in, a->ny_in, a->nx_in);
    else
        snprintf(msg, sizeof msg, "lec_check_maps: %d table entries are out of range (the maps themselves are fine); first: step_d[%d]: its source step must lie in "
                 "[%d, %d), its latitude offset in [0, %d], its longitude offset in [0, %d]", h[0], h[1] & 0xffffff, a->step_base, a->step_base + a->nt_src,
                 nj - a->ny, ni - a->nx);
    return lec_set_error(LEC_ERR_ARG, msg);
}
