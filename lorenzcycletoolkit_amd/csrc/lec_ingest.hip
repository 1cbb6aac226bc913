// lec_ingest.hip -- raw file bytes in device memory -> the field cubes stage 1 reads.
//
// One gather pass does what the reference does in four host passes over the whole data set: CF decode
// (scale_factor / add_offset / _FillValue), longitude wrap and the sorts of process_data, the domain
// crop of slice_domain and the unit conversion of BoxData._extract_data.  HBM-bound: reads
// sizeof(src) bytes per selected element, writes sizeof(out).  One workgroup per output row
// (t, k, j); lanes walk the row, so the writes are coalesced and the reads are two contiguous
// segments of the source row (a rolled longitude axis) or one.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"

namespace {

struct IngestParams {
    const void* src; void* out;
    int nt, nl_in, ny_in, nx_in, nl, ny, nx;
    const int* kmap; const int* jmap; const int* imap;
    int swap, has_packing, has_fill, decode_f32;
    double scale, offset, fill, unit;
    const int* step; int step_base;      // per-step gathers: {source step, latitude offset, longitude offset} per output step, or null
    int nt_src, jmap_len, imap_len;      // ... and what bounds that table's entries
};

__device__ __forceinline__ int16_t load_elem(const int16_t* p, bool swap) {
    uint16_t b = *(const uint16_t*)p;
    if (swap) b = (uint16_t)((b >> 8) | (b << 8));
    return (int16_t)b;
}
__device__ __forceinline__ int8_t load_elem(const int8_t* p, bool) { return *p; }
__device__ __forceinline__ int32_t load_elem(const int32_t* p, bool swap) {
    uint32_t b = *(const uint32_t*)p;
    if (swap) b = __builtin_bswap32(b);
    return (int32_t)b;
}
__device__ __forceinline__ float load_elem(const float* p, bool swap) {
    uint32_t b = *(const uint32_t*)p;
    if (swap) b = __builtin_bswap32(b);
    return __uint_as_float(b);
}
__device__ __forceinline__ double load_elem(const double* p, bool swap) {
    uint64_t b = *(const uint64_t*)p;
    if (swap) b = __builtin_bswap64(b);
    return __longlong_as_double((long long)b);
}

template <typename TSRC, typename TOUT>
__global__ void __launch_bounds__(256) lec_ingest_kernel(const IngestParams p) {
#pragma clang fp contract(off)
    const int row = blockIdx.x;                     // (t, k, j) of the output
    const int j = row % p.ny;
    const int k = (row / p.ny) % p.nl;
    const int t = row / (p.ny * p.nl);
    // (a box-packed series: this output step's box may start anywhere in the maps and come from another source step)
    const int ts = p.step ? p.step[3 * t] - p.step_base : t, oj = p.step ? p.step[3 * t + 1] : 0, oi = p.step ? p.step[3 * t + 2] : 0;
    if (p.step && ((unsigned)ts >= (unsigned)p.nt_src || oj < 0 || oj > p.jmap_len - p.ny || oi < 0 || oi > p.imap_len - p.nx)) {
        // an entry of the device-resident table that points outside the source or the maps: nothing is read, the step's rows say so
        TOUT* __restrict__ bad = (TOUT*)p.out + (size_t)row * p.nx;
        for (int i = threadIdx.x; i < p.nx; i += blockDim.x) bad[i] = (TOUT)__builtin_nan("");
        return;
    }
    const TSRC* __restrict__ src = (const TSRC*)p.src + (((size_t)ts * p.nl_in + p.kmap[k]) * p.ny_in + p.jmap[j + oj]) * (size_t)p.nx_in;
    TOUT* __restrict__ out = (TOUT*)p.out + (size_t)row * p.nx;
    const int* __restrict__ imap = p.imap + oi;
    const bool swap = p.swap != 0;
    for (int i = threadIdx.x; i < p.nx; i += blockDim.x) {
        const TSRC raw = load_elem(src + imap[i], swap);
        const bool is_fill = p.has_fill && ((double)raw == p.fill);
        TOUT o;
        if (!p.decode_f32) {
            double v = (double)raw;
            if (p.has_packing) { v = v * p.scale; v = v + p.offset; }
            v = v * p.unit;
            o = (TOUT)v;
        } else {
            // the reference's decode in float32 (xarray 2024.2.0 on NumPy 2: float32 data, float64 attributes -- every
            // operation is computed in float64 and rounded back to float32 by the in-place store)
            float v = (float)raw;
            if (p.has_packing) { v = (float)((double)v * p.scale); v = (float)((double)v + p.offset); }
            v = v * (float)p.unit;
            o = (TOUT)v;
        }
        out[i] = is_fill ? (TOUT)__builtin_nan("") : o;
    }
}

template <typename TSRC>
void launch(const IngestParams& p, int out_dtype, long long rows, hipStream_t st) {
    // one workgroup per output row; its threads walk the row: short rows (a box-packed series: 61 columns) get one wave, not four
    const dim3 block(p.nx <= 64 ? 64 : (p.nx <= 128 ? 128 : 256));
    if (out_dtype == LEC_F64) hipLaunchKernelGGL((lec_ingest_kernel<TSRC, double>), dim3((unsigned)rows), block, 0, st, p);
    else hipLaunchKernelGGL((lec_ingest_kernel<TSRC, float>), dim3((unsigned)rows), block, 0, st, p);
}

}  // namespace

extern "C" int lec_ingest(const lec_ingest_args* a) {
    if (!a) return lec_set_error(LEC_ERR_ARG, "lec_ingest: null args");
    if (!a->src_d || !a->out_d || !a->kmap_d || !a->jmap_d || !a->imap_d) return lec_set_error(LEC_ERR_ARG, "lec_ingest: null pointer argument");
    if (a->src_dtype < LEC_F64 || a->src_dtype > LEC_I8) return lec_set_error(LEC_ERR_ARG, "lec_ingest: src_dtype must be LEC_I8, LEC_I16, LEC_I32, LEC_F32 or LEC_F64");
    if (a->out_dtype != LEC_F64 && a->out_dtype != LEC_F32) return lec_set_error(LEC_ERR_ARG, "lec_ingest: out_dtype must be LEC_F64 or LEC_F32");
    if (a->nt < 1 || a->nl_in < 1 || a->ny_in < 1 || a->nx_in < 1 || a->nl < 1 || a->ny < 1 || a->nx < 1 ||
        a->nl > a->nl_in || a->ny > a->ny_in || a->nx > a->nx_in)
        return lec_set_error(LEC_ERR_ARG, "lec_ingest: output extents must be 1..source extents");
    const long long rows = (long long)a->nt * a->nl * a->ny;
    if (rows > 0x7fffffffLL) return lec_set_error(LEC_ERR_UNSUPPORTED, "lec_ingest: more than 2^31-1 rows in one call");
    IngestParams p;
    p.src = a->src_d; p.out = a->out_d;
    p.nt = a->nt; p.nl_in = a->nl_in; p.ny_in = a->ny_in; p.nx_in = a->nx_in; p.nl = a->nl; p.ny = a->ny; p.nx = a->nx;
    p.kmap = a->kmap_d; p.jmap = a->jmap_d; p.imap = a->imap_d;
    p.step = a->step_d; p.step_base = a->step_base;
    p.nt_src = a->nt_src; p.jmap_len = a->jmap_len; p.imap_len = a->imap_len;
    if (a->step_d) {
        if (a->nt_src < 1 || a->jmap_len < a->ny || a->imap_len < a->nx)
            return lec_set_error(LEC_ERR_ARG, "lec_ingest: a step_d table needs nt_src >= 1, jmap_len >= ny and imap_len >= nx (what bounds its entries)");
    } else if ((a->jmap_len != 0 && a->jmap_len != a->ny) || (a->imap_len != 0 && a->imap_len != a->nx)) {
        return lec_set_error(LEC_ERR_ARG, "lec_ingest: without step_d the maps have ny / nx entries (jmap_len / imap_len: 0 or exactly that)");
    }
    if (a->decode_dtype != LEC_F64 && a->decode_dtype != LEC_F32) return lec_set_error(LEC_ERR_ARG, "lec_ingest: decode_dtype must be LEC_F64 or LEC_F32");
    if (a->decode_dtype == LEC_F32 && (a->src_dtype == LEC_F64 || a->src_dtype == LEC_I32))
        return lec_set_error(LEC_ERR_ARG, "lec_ingest: float64 and int32 data do not decode to float32");
    p.swap = a->swap_bytes; p.has_packing = a->has_packing; p.has_fill = a->has_fill; p.decode_f32 = a->decode_dtype == LEC_F32;
    p.scale = a->scale_factor; p.offset = a->add_offset; p.fill = a->fill_value; p.unit = a->unit_scale;
    hipStream_t st = (hipStream_t)a->stream;
    if (a->src_dtype == LEC_I16) launch<int16_t>(p, a->out_dtype, rows, st);
    else if (a->src_dtype == LEC_I32) launch<int32_t>(p, a->out_dtype, rows, st);
    else if (a->src_dtype == LEC_I8) launch<int8_t>(p, a->out_dtype, rows, st);
    else if (a->src_dtype == LEC_F32) launch<float>(p, a->out_dtype, rows, st);
    else launch<double>(p, a->out_dtype, rows, st);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    return LEC_OK;
}
