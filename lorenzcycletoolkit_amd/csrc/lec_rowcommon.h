// lec_rowcommon.h -- device helpers shared by the stage-1 kernels (vector loads, block sums, parameters).
#ifndef LEC_ROWCOMMON_H
#define LEC_ROWCOMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"

namespace lec {

constexpr double kCp = LEC_CP_D;

// ---------------------------------------------------------------------------------------------
// vector loads: VEC elements of TIN -> TO[VEC] (TO = double, or TIN itself to keep the raw values and convert at use)
// ---------------------------------------------------------------------------------------------
template <typename TIN, int VEC>
struct VecLoad;

typedef double dbl2_t __attribute__((ext_vector_type(2)));
typedef float flt4_t __attribute__((ext_vector_type(4)));

// NT = nontemporal (streaming) load: the line is not kept in L2 ahead of re-used rows
template <>
struct VecLoad<double, 2> {
    template <bool NT, typename TO>
    static __device__ __forceinline__ void load(const double* p, TO (&o)[2]) {
        const dbl2_t* q = reinterpret_cast<const dbl2_t*>(p);
        const dbl2_t v = NT ? __builtin_nontemporal_load(q) : *q;
        o[0] = v.x; o[1] = v.y;
    }
};
template <>
struct VecLoad<double, 1> {
    template <bool NT, typename TO>
    static __device__ __forceinline__ void load(const double* p, TO (&o)[1]) { o[0] = (TO)(NT ? __builtin_nontemporal_load(p) : *p); }
};
template <>
struct VecLoad<float, 4> {
    template <bool NT, typename TO>
    static __device__ __forceinline__ void load(const float* p, TO (&o)[4]) {
        const flt4_t* q = reinterpret_cast<const flt4_t*>(p);
        const flt4_t v = NT ? __builtin_nontemporal_load(q) : *q;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
typedef float flt2_t __attribute__((ext_vector_type(2)));
template <>
struct VecLoad<float, 2> {
    template <bool NT, typename TO>
    static __device__ __forceinline__ void load(const float* p, TO (&o)[2]) {
        const flt2_t* q = reinterpret_cast<const flt2_t*>(p);
        const flt2_t v = NT ? __builtin_nontemporal_load(q) : *q;
        o[0] = v.x; o[1] = v.y;
    }
};
template <>
struct VecLoad<float, 1> {
    template <bool NT, typename TO>
    static __device__ __forceinline__ void load(const float* p, TO (&o)[1]) { o[0] = (TO)(NT ? __builtin_nontemporal_load(p) : *p); }
};

// Branch-free row loads.  `e0c` is the lane's first element index clamped so that the 16-byte vector
// always lies inside the row's own memory (every vector that holds at least one box element is
// inside the cube because rows start/end on vector boundaries in the aligned instantiation; lanes
// wholly outside the box are clamped onto the last such vector).  Out-of-box elements are zeroed by
// the caller with a select -- no divergent branch, so the compiler can keep every load of a row in
// flight at once (a branchy version serialised them behind s_waitcnt vmcnt(0)).
// `rowa` = row pointer moved back to its 16-byte boundary (wave-uniform, lives in SGPRs), `off` = the
// lane's non-negative element offset from it: the loads become `global_load ... v_off, s[base]` with a
// 32-bit VGPR offset instead of a 64-bit per-lane address (saves two VGPRs and a 64-bit add per load).
template <typename TIN, int VEC, bool NT, typename TO>
__device__ __forceinline__ void load_vec(const TIN* __restrict__ rowa, unsigned off, TO (&o)[VEC]) {
    const unsigned boff = off * (unsigned)sizeof(TIN);     // 32-bit byte offset: rows are far shorter than 4 GiB
    VecLoad<TIN, VEC>::template load<NT, TO>(reinterpret_cast<const TIN*>(reinterpret_cast<const char*>(rowa) + boff), o);
}

// ---------------------------------------------------------------------------------------------
// block-wide sums of N per-thread values through an LDS transpose.
// NTHR threads; R lanes cooperate on one statistic (8 for NTHR >= 256, else NTHR / 32), each adds
// NTHR / R partials in a fixed order, then an R-lane butterfly.  On return the lanes with
// (tid % R) == 0 and tid / R < N hold the total of statistic tid / R.
// `red` needs N * red_stride(NTHR) doubles.
// ---------------------------------------------------------------------------------------------
// synchronisation among the NTHR threads that reduce one row: a single wave needs no s_barrier -- its LDS operations are
// processed in order -- only the compiler must not move them across this point; so one-wave rows of a multi-wave
// workgroup stay independent of each other (no workgroup barrier in their epilogue)
template <int NTHR>
__device__ __forceinline__ void row_sync() {
    if (NTHR <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

constexpr int red_stride(int nthr) { return nthr + 8; }   // (stride mod 32) == 8: a wave's 8 statistics land on disjoint LDS banks
constexpr int red_rshift(int nthr) { return nthr >= 256 ? 3 : (nthr == 128 ? 2 : 1); }
constexpr int kRedStride = red_stride(256);

template <int N, int NTHR, int RSHIFT = red_rshift(NTHR)>
__device__ __forceinline__ double block_sums(const double (&v)[N], double* red, int tid) {
    constexpr int stride = red_stride(NTHR), rshift = RSHIFT, R = 1 << rshift, M = NTHR / R;
    static_assert(N * R <= NTHR, "too many statistics for this block size");
#pragma unroll
    for (int s = 0; s < N; ++s) red[s * stride + tid] = v[s];
    row_sync<NTHR>();
    const int s = tid >> rshift, part = tid & (R - 1);
    double acc = 0.0;
    if (s < N) {
        const double* src = red + s * stride + part;
#pragma unroll 8
        for (int m = 0; m < M; ++m) acc += src[m << rshift];
    }
#pragma unroll
    for (int o = R >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
}

// runtime-block-size form used by the sweep kernel (nthr in {64, 128, 256})
template <int N>
__device__ __forceinline__ double block_sums(const double (&v)[N], double* red, int tid, int nthr) {
#pragma unroll
    for (int s = 0; s < N; ++s) red[s * kRedStride + tid] = v[s];
    __syncthreads();
    const int rshift = (nthr == 256) ? 3 : (nthr == 128 ? 2 : 1);
    const int R = 1 << rshift;
    const int s = tid >> rshift, part = tid & (R - 1);
    double acc = 0.0;
    if (s < N) {
        const double* src = red + s * kRedStride + part;
#pragma unroll 8
        for (int m = 0; m < 32; ++m) acc += src[m << rshift];
    }
    for (int o = R >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
}

struct RowParams {
    const void* T; const void* U; const void* V; const void* W; const void* P; const void* DT;
    const void* TM; const void* TP;      // box-packed series (box-tile kernel only): T(t-1), T(t+1) on the box of step t; else null
    int nt, nl, ny, nx;
    int t_begin, t_count;
    int n_box, nxb_max, nyb_max;
    const int* box;
    const double* boxtab;
    const double* wlon;
    const double* glon;
    const double* lattab;
    const double* levtab;
    const double* tcoef;
    double* rows;
    int order;   // block -> row mapping: 0 memory order, 1 XCD-chunked latitudes with level fastest
    int jchunk;  // order >= 1: latitudes per XCD chunk
    int ntrips;  // single-sweep kernel: trips of 64 vectors that cover the longest row
    int tgroup;  // order 7: time steps per tile
    int jgroup;  // order 7: latitudes per tile
    int jrows;   // sweep kernel: latitudes per workgroup
    int cpx;     // sweep kernel: latitude chunks per XCD
};


// the same launch without its first time step (fixed box only: per-step box tables are indexed by the local step)
inline RowParams later_steps(const RowParams& p) {
    RowParams q = p;
    q.t_begin += 1; q.t_count -= 1;
    q.rows += (size_t)p.nl * p.nyb_max * LEC_NSTAT;
    return q;
}

}  // namespace lec
#endif
