// lec_hostio.hip -- host-memory plumbing of the device ingest (include/lec_hip.h: lec_host_register, lec_host_unregister,
// lec_copy_rows_async).
//
// The reference reads its file through xarray into NumPy memory (src/utils/preprocessing.py:35-146).  The device ingest moves the
// file's bytes to the GPU as they are; with these three entry points it does so WITHOUT a staging copy: a span of the memory-mapped
// file is registered with the HIP runtime (its page-cache pages are pinned and mapped for the GPU's copy engines), the rows a
// chunk needs are copied from it asynchronously (a strided 2-D copy picks a latitude band out of every level), and the span is
// unregistered when the copies have completed.  The caller keeps the bookkeeping (which spans are registered); the library only
// forwards to the one HIP runtime of the process.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"

extern "C" int lec_host_register(const void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return lec_set_error(LEC_ERR_ARG, "lec_host_register: null pointer or empty span");
    const hipError_t e = hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();                   // the failure is reported through the return code: do not leave it pending
        char msg[200];
        snprintf(msg, sizeof msg, "lec_host_register: hipHostRegister(%p, %zu) failed: %s", ptr, bytes, hipGetErrorString(e));
        return lec_set_error(LEC_ERR_LAUNCH, msg);
    }
    return LEC_OK;
}

extern "C" int lec_host_unregister(const void* ptr) {
    if (!ptr) return lec_set_error(LEC_ERR_ARG, "lec_host_unregister: null pointer");
    const hipError_t e = hipHostUnregister(const_cast<void*>(ptr));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    }
    return LEC_OK;
}

extern "C" int lec_copy_rows_async(void* dst_d, size_t dst_pitch, const void* src_h, size_t src_pitch, size_t width_bytes, size_t rows,
                                   void* stream) {
    if (!dst_d || !src_h) return lec_set_error(LEC_ERR_ARG, "lec_copy_rows_async: null pointer");
    if (width_bytes == 0 || rows == 0) return LEC_OK;
    if (rows > 1 && (dst_pitch < width_bytes || src_pitch < width_bytes)) return lec_set_error(LEC_ERR_ARG, "lec_copy_rows_async: pitch shorter than a row");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (rows == 1 || (dst_pitch == width_bytes && src_pitch == width_bytes))
        e = hipMemcpyAsync(dst_d, src_h, width_bytes * rows, hipMemcpyHostToDevice, st);
    else
        e = hipMemcpy2DAsync(dst_d, dst_pitch, src_h, src_pitch, width_bytes, rows, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    }
    return LEC_OK;
}
