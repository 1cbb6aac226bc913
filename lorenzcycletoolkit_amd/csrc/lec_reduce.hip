// lec_reduce.hip -- stage 2 of the MI355X Lorenz-Energy-Cycle engine (gfx950, wave64).
//
// Consumes the per-(time, level, lat) row records of lec_rowstats and produces, per time step, the 16
// integrated terms and the 21 per-level tables.  Three small kernels (O(level x lat) work, <1 % of the
// stage-1 bytes):
//   lec_area_means_kernel   {[X]} cos-weighted meridional means of the six zonal means        (calc_averages.py:46-78)
//   lec_level_terms_kernel  sigma, every per-level integrand and boundary piece of one level
//                           (energy_contents.py:99-165, conversion_terms.py:103-245,
//                            boundary_terms.py:125-418, generation_and_dissipation_terms.py:122-152)
//   lec_vertical_kernel     _handle_nans + integrate(level) + boundary assembly               (energy_contents.py:190-208)
// Formulas: SURVEY.md appendix A / F (factored through row statistics; exact in exact arithmetic).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"
#include "lec_rowcommon.h"

// Two kernels form the per-level sums (lec_level_terms_kernel for any box height, lec_level_small_kernel for up to 64 rows) from the
// same source lines (lec_level_row.inc).  Contraction is off in this file so that they also give the same BITS: which products the
// compiler would fuse into multiply-adds depends on the surrounding code.
#pragma clang fp contract(off)

namespace {
using lec::dbl2_t;

constexpr double kG = LEC_G, kRe = LEC_RE, kRd = LEC_RD, kCp = LEC_CP_D;
constexpr int kMaxNl = 160;
constexpr int kStageRows = 66;     // lec_level_terms_kernel: latitude rows staged through LDS at a time (64 + one either side)
constexpr int kPartStride = 65;    // partial sums of one statistic, one per lane (+1: conflict-free both ways)
constexpr int kRecStride = LEC_NSTAT + 1, kLatStride = 9;      // LDS row strides (doubles): odd, so a lane per row is conflict-free

enum {
    V_AZ = 0, V_AE, V_KZ, V_KE, V_CZ2, V_CE2, V_CA1, V_CA2, V_CK1, V_CK2, V_CK3, V_CK4, V_CK5, V_GZ, V_GE,
    V_B1 = 15,  // BAz BAe BKz BKe BPhiZ BPhiE : east-west term, integrated over phi
    V_B2 = 21,  // north-south term
    V_B3 = 27,  // bottom-top term (area mean)
    V_SIG = 33,
    V_COUNT = 34
};
static_assert(V_COUNT <= LEC_NLEVRAW, "levraw record too small");

struct RedParams {
    const double* rows;
    int t_count, nl, n_box, nyb_max;
    const int* box;
    const double* boxtab2;
    const double* lattab2;
    const double* levtab2;
    double phi_scale;
    int drop_any_time;      // fixed framework: a level still NaN at ANY time step is dropped for every time step
    int* dropmask;          // [F_COUNT][nl], filled by lec_dropmask_kernel
    double* am;
    double* levraw;
    double* scalars;
    double* levels;
    int* nanflag;
    long long sstride, lstride;     // doubles between consecutive time steps of `scalars` / `levels`
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// grid (nl * t_count) -- level fastest: neighbouring workgroups are neighbouring levels of one time step --, block 64.  Like
// lec_level_terms_kernel: the record loads are addressed from the kernel arguments alone (nyb_max rows) and issued before the box
// height arrives; rows below a lower box are masked out afterwards.
__global__ void __launch_bounds__(64) lec_area_means_kernel(const RedParams p) {
    const int tl = blockIdx.x / p.nl, k = blockIdx.x - tl * p.nl, lane = threadIdx.x;
    const int bi = (p.n_box == 1) ? 0 : tl;
    const double* rec = p.rows + (size_t)(tl * p.nl + k) * p.nyb_max * LEC_NSTAT;
    const double* lt = p.lattab2 + (size_t)bi * p.nyb_max * 8;
    double a[6] = {0, 0, 0, 0, 0, 0};
    int nyb = 0;
    for (int j0 = 0; j0 < p.nyb_max; j0 += 64) {
        const int jb = j0 + lane, jc = min(jb, p.nyb_max - 1);
        const dbl2_t* r = reinterpret_cast<const dbl2_t*>(rec + (size_t)jc * LEC_NSTAT);
        const dbl2_t v0 = r[0], v1 = r[1], v2 = r[2];          // the six zonal means [T] [u] [v] [w] [Phi] [Q]
        const double cw = lt[8 * jc + 0];
        if (j0 == 0) nyb = p.box[4 * bi + 3] - p.box[4 * bi + 2] + 1;
        if (jb < nyb) {
            a[0] += cw * v0.x; a[1] += cw * v0.y; a[2] += cw * v1.x; a[3] += cw * v1.y; a[4] += cw * v2.x; a[5] += cw * v2.y;
        }
    }
#pragma unroll
    for (int s = 0; s < 6; ++s) a[s] = wave_sum(a[s]);
    if (lane == 0) {
        double* o = p.am + (size_t)(tl * p.nl + k) * 8;
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
        o[4] = a[4] * p.phi_scale; o[5] = a[5]; o[6] = 0.0; o[7] = 0.0;
    }
}

// BAz's bottom-top term before the area mean, one latitude row:  [2 w'T'] T* + [w] T*^2   (boundary_terms.py:165-168)
__device__ __forceinline__ double baz3_row(const RedParams& p, const double* am_t, int tl, int k, int jb) {
    const double* r = p.rows + ((size_t)(tl * p.nl + k) * p.nyb_max + jb) * LEC_NSTAT;
    const double Ts = r[LEC_S_MT] - am_t[8 * k + 0];
    return (2 * r[LEC_S_WT]) * Ts + r[LEC_S_MW] * (Ts * Ts);
}

// The reference repairs THIS term per latitude, before the area mean and before the division by sigma
// (term3 = self._handle_nans(term3) on a [time, level, lat] array, boundary_terms.py:169): a NaN at (level, lat) is replaced by
// the linear interpolation in p between the nearest valid levels of the same latitude (interior gaps only, no
// extrapolation); what stays NaN makes the level's area mean NaN, and the level is then dropped (lec_vertical_kernel).
__device__ double baz3_repaired(const RedParams& p, const double* am_t, int tl, int k, int jb) {
    int lo = k - 1, hi = k + 1;
    double yl = 0.0, yr = 0.0;
    for (; lo >= 0; --lo) { yl = baz3_row(p, am_t, tl, lo, jb); if (!isnan(yl)) break; }
    for (; hi < p.nl; ++hi) { yr = baz3_row(p, am_t, tl, hi, jb); if (!isnan(yr)) break; }
    if (lo < 0 || hi >= p.nl) return nan("");
    const double xl = p.levtab2[4 * lo], xr = p.levtab2[4 * hi];
    const double slope = (yr - yl) / (xr - xl);
    return slope * (p.levtab2[4 * k] - xl) + yl;
}

// grid (nl * t_count), block 64.  A wave's life here is a chain of memory round trips (a few hundred arithmetic instructions between
// them), so the kernel is arranged to have TWO: every global load -- the level's records, the latitude table, the records of the
// levels above and below -- is addressed from the kernel arguments alone (the staging covers nyb_max rows; the rows below a lower box
// are zero records) and issued before anything waits for the area means or the box height.
__global__ void __launch_bounds__(64) lec_level_terms_kernel(const RedParams p) {
    const int tl = blockIdx.x / p.nl, k = blockIdx.x - tl * p.nl, lane = threadIdx.x;
    const int nl = p.nl, nyb_max = p.nyb_max;
    const int bi = (p.n_box == 1) ? 0 : tl;
    const int km = k > 0 ? k - 1 : k, kp = k < nl - 1 ? k + 1 : k;
    const size_t lstride = (size_t)nyb_max * LEC_NSTAT;
    const double* rec = p.rows + (size_t)(tl * nl + k) * lstride;
    const double* recm = p.rows + (size_t)(tl * nl + km) * lstride;
    const double* recp = p.rows + (size_t)(tl * nl + kp) * lstride;
    const double* lt = p.lattab2 + (size_t)bi * nyb_max * 8;
    const double ps = p.phi_scale;

    // The level's records (256 B each) and the latitude table are staged through LDS 64 rows (+ one row either side) at a time:
    // 16-byte-per-lane loads of consecutive addresses instead of ~45 loads that each touch one 128-byte line per lane.  Row strides of
    // 33 / 9 doubles keep the lane-per-row reads free of bank conflicts.  Same arithmetic and order as a direct read.
    __shared__ double tile[kStageRows * kRecStride];
    __shared__ double ltile[kStageRows * kLatStride];
    constexpr int kIt2 = (kStageRows * (LEC_NSTAT / 2) + 63) / 64, kIt8 = (kStageRows * 4 + 63) / 64;

    double acc[V_COUNT];
#pragma unroll
    for (int i = 0; i < V_COUNT; ++i) acc[i] = 0.0;
    int nyb = 0;
    double aT = 0, aW = 0, aP = 0, aQ = 0, aTm = 0, aTp = 0, pa = 0, pb = 0, pc = 0;
    const double* am = p.am + (size_t)tl * nl * 8;

    for (int j0 = 0; j0 < nyb_max; j0 += 64) {
        const int jlo = j0 > 0 ? j0 - 1 : 0, jhi = min(j0 + 64, nyb_max - 1);
        const int n2 = (jhi - jlo + 1) * (LEC_NSTAT / 2), n8 = (jhi - jlo + 1) * 4;
        const dbl2_t* src = reinterpret_cast<const dbl2_t*>(rec + (size_t)jlo * LEC_NSTAT);
        const dbl2_t* lsrc = reinterpret_cast<const dbl2_t*>(lt + (size_t)jlo * 8);
        dbl2_t v2[kIt2], v8[kIt8];
#pragma unroll
        for (int it = 0; it < kIt2; ++it) v2[it] = src[min(it * 64 + lane, n2 - 1)];
#pragma unroll
        for (int it = 0; it < kIt8; ++it) v8[it] = lsrc[min(it * 64 + lane, n8 - 1)];
        const int jb = j0 + lane, jbc = min(jb, nyb_max - 1);
        static_assert(LEC_S_MT == 0 && LEC_S_MU == 1, "the two means read from the neighbouring levels are one 16-byte load");
        const dbl2_t km2 = *reinterpret_cast<const dbl2_t*>(recm + (size_t)jbc * LEC_NSTAT);
        const dbl2_t kp2 = *reinterpret_cast<const dbl2_t*>(recp + (size_t)jbc * LEC_NSTAT);
        if (j0 == 0) {
            nyb = p.box[4 * bi + 3] - p.box[4 * bi + 2] + 1;
            aT = am[8 * k + 0]; aW = am[8 * k + 3]; aP = am[8 * k + 4]; aQ = am[8 * k + 5];
            aTm = am[8 * km + 0]; aTp = am[8 * kp + 0];
            pa = p.levtab2[4 * k + 1]; pb = p.levtab2[4 * k + 2]; pc = p.levtab2[4 * k + 3];
        }
        __syncthreads();                                     // the previous chunk has been consumed
#pragma unroll
        for (int it = 0; it < kIt2; ++it) {
            const int i = it * 64 + lane;
            if (i < n2) { double* d = tile + (i >> 4) * kRecStride + 2 * (i & 15); d[0] = v2[it].x; d[1] = v2[it].y; }
        }
#pragma unroll
        for (int it = 0; it < kIt8; ++it) {
            const int i = it * 64 + lane;
            if (i < n8) { double* d = ltile + (i >> 2) * kLatStride + 2 * (i & 3); d[0] = v8[it].x; d[1] = v8[it].y; }
        }
        __syncthreads();
        if (jb >= nyb) continue;
        const int jm = jb > 0 ? jb - 1 : jb, jp = jb < nyb - 1 ? jb + 1 : jb;
        const double* r = tile + (jb - jlo) * kRecStride;
        const double* rjm = tile + (jm - jlo) * kRecStride;
        const double* rjp = tile + (jp - jlo) * kRecStride;
        const double* l0 = ltile + (jb - jlo) * kLatStride;
        const double cw = l0[0], wphi = l0[1], c = l0[2], tn = l0[3];
        const double gra = l0[4], grb = l0[5], grc = l0[6];
        const double cm = ltile[(jm - jlo) * kLatStride + 2], cp = ltile[(jp - jlo) * kLatStride + 2];

#define LEC_BAZ3_REPAIR(jb_) baz3_repaired(p, am, tl, k, (jb_))
#define LEC_ROW_COMMON
#define LEC_ROW_PART_A
#define LEC_ROW_PART_B
#include "lec_level_row.inc"
#undef LEC_ROW_COMMON
#undef LEC_ROW_PART_A
#undef LEC_ROW_PART_B
#undef LEC_BAZ3_REPAIR
    }
    // static stability (thermodynamics.py:55-70); the zonal/area mean commutes with the linear d/dp
    const double pk = p.levtab2[4 * k + 0];
    double sig = kG * aT / kCp - (pk * kG / kRd) * (pa * aTm + pb * aT + pc * aTp);
    sig = (sig > 0.03) ? sig : 0.03;

    // 64 partial sums per statistic -> one total, through LDS (the staging tile is free): lane s adds the partials of statistic s
    // in a fixed order (four chains), applies the statistic's divisor and stores it -- no lane-0 epilogue, no butterfly.
    static_assert(V_SIG * kPartStride <= kStageRows * kRecStride, "partial sums must fit the staging tile");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < V_SIG; ++i) tile[i * kPartStride + lane] = acc[i];
    __syncthreads();
    if (lane < LEC_NLEVRAW) {
        const int s = lane;
        double tot = 0.0, div = 1.0;
        if (s < V_SIG) {
            const double* q = tile + s * kPartStride;
            double c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3];
#pragma unroll
            for (int l = 4; l < 64; l += 4) { c0 += q[l]; c1 += q[l + 1]; c2 += q[l + 2]; c3 += q[l + 3]; }
            tot = (c0 + c1) + (c2 + c3);
            const int b = (s >= V_B1) ? (s - V_B1) % 6 : -1;         // boundary pieces: Az Ae | Kz Ke | PhiZ PhiE
            if (s == V_AZ || s == V_AE || b == 0 || b == 1) div = 2 * sig;
            else if (b == 2 || b == 3) div = 2 * kG;
            else if (b == 4 || b == 5) div = kG;
            else if (s == V_CA1) div = 2 * kRe * sig;
            else if (s == V_CA2) div = sig;
            else if (s == V_GZ || s == V_GE) div = kCp * sig;
        } else if (s == V_SIG) {
            tot = sig;
        }
        p.levraw[(size_t)(tl * nl + k) * LEC_NLEVRAW + s] = tot / div;        // x / 1.0 is x
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Boxes of at most 64 latitude rows (the moving framework's 61 x 61 boxes, every regional box on a 2.5-degree grid): one wave per
// (level, time step), one row per lane, the record in registers, the area means formed here (no lec_area_means_kernel launch, no
// `am` round trip), neighbouring rows through lane shuffles, no staging tile.  Same row arithmetic (lec_level_row.inc), same butterfly for the area
// means, same four-chain reduction: the numbers are those of the general kernels, bit for bit.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kSmallRows = 64;
constexpr int kSmallRound = 18;       // sums per reduction round (two rounds: 15 + 18)

// {[T]} of another level exactly as that level's own wave forms it: the products of lec_area_means_kernel summed in the order of
// wave_sum's xor butterfly -- lane 0 ends with ((((l0 + l32) + (l16 + l48)) + ...): pairs 32 apart first, then 16, 8, 4, 2, 1 -- written
// as five nested two-trip loops (rolled: this is the rare NaN-repair path and must not cost the kernel its registers)
__device__ double area_mean_T_small(const RedParams& p, const double* lt, int nyb, int tl, int level) {
    const double* rec = p.rows + (size_t)(tl * p.nl + level) * p.nyb_max * LEC_NSTAT;
    auto leaf = [&](int l) -> double { return l < nyb ? lt[8 * l] * rec[(size_t)l * LEC_NSTAT + LEC_S_MT] : 0.0; };
    double a1 = 0.0, a2 = 0.0, a4 = 0.0, a8 = 0.0, a16 = 0.0;
#pragma unroll 1
    for (int b1 = 0; b1 < 2; b1 += 1) {
#pragma unroll 1
        for (int b2 = 0; b2 < 4; b2 += 2) {
#pragma unroll 1
            for (int b4 = 0; b4 < 8; b4 += 4) {
#pragma unroll 1
                for (int b8 = 0; b8 < 16; b8 += 8) {
#pragma unroll 1
                    for (int b16 = 0; b16 < 32; b16 += 16) {
                        const int i = b1 | b2 | b4 | b8 | b16;
                        const double s32 = leaf(i) + leaf(i ^ 32);
                        a16 = b16 ? a16 + s32 : s32;
                    }
                    a8 = b8 ? a8 + a16 : a16;
                }
                a4 = b4 ? a4 + a8 : a8;
            }
            a2 = b2 ? a2 + a4 : a4;
        }
        a1 = b1 ? a1 + a2 : a2;
    }
    return a1;
}

__device__ double baz3_row_small(const RedParams& p, const double* lt, int nyb, int tl, int level, int jb) {
    const double* r = p.rows + ((size_t)(tl * p.nl + level) * p.nyb_max + jb) * LEC_NSTAT;
    const double Ts = r[LEC_S_MT] - area_mean_T_small(p, lt, nyb, tl, level);
    return (2 * r[LEC_S_WT]) * Ts + r[LEC_S_MW] * (Ts * Ts);
}

// baz3_repaired for the small kernel (no `am` array: the area means of the other levels are re-formed)
__device__ double baz3_repaired_small(const RedParams& p, const double* lt, int nyb, int tl, int k, int jb) {
    int lo = k - 1, hi = k + 1;
    double yl = 0.0, yr = 0.0;
    for (; lo >= 0; --lo) { yl = baz3_row_small(p, lt, nyb, tl, lo, jb); if (!isnan(yl)) break; }
    for (; hi < p.nl; ++hi) { yr = baz3_row_small(p, lt, nyb, tl, hi, jb); if (!isnan(yr)) break; }
    if (lo < 0 || hi >= p.nl) return nan("");
    const double xl = p.levtab2[4 * lo], xr = p.levtab2[4 * hi];
    const double slope = (yr - yl) / (xr - xl);
    return slope * (p.levtab2[4 * k] - xl) + yl;
}

// `acc[i] += x` of lec_level_row.inc, for a kernel in which every sum receives exactly one contribution per lane: the contribution goes
// straight to the lane's slot of the reduction buffer instead of waiting in a register (rows below the box contribute 0)
struct LaneSums {
    double* slot0; int first; bool in;
    struct Ref {
        double* q; bool in;
        __device__ __forceinline__ void operator+=(double x) const { *q = in ? x : 0.0; }
    };
    __device__ __forceinline__ Ref operator[](int i) const { return Ref{slot0 + (i - first) * kPartStride, in}; }
};

// grid (nl * t_count), block 64; requires nyb_max <= 64
#ifndef LEC_SMALL_WAVES
// (rounds 2-5: 193 VGPRs, two waves per SIMD -- the hand-over's fully unrolled loop had sixty LDS values in flight at once.  Round 6:
// four waves per SIMD, and the SAME 85 us per 512 x 37 levels: the kernel is not bound by its occupancy; nor by its record loads --
// staged through LDS as coalesced 16-byte loads: 86 us.  profiles/r06_stage2_pmc.txt)
#define LEC_SMALL_WAVES 4
#endif
__global__ void __launch_bounds__(64, LEC_SMALL_WAVES) lec_level_small_kernel(const RedParams p) {
    __shared__ double part[kSmallRound * kPartStride];
    const int tl = blockIdx.x / p.nl, k = blockIdx.x - tl * p.nl, lane = threadIdx.x;
    const int nl = p.nl, nyb_max = p.nyb_max;
    const int bi = (p.n_box == 1) ? 0 : tl;
    const int km = k > 0 ? k - 1 : k, kp = k < nl - 1 ? k + 1 : k;
    const size_t lstride = (size_t)nyb_max * LEC_NSTAT;
    const double* lt = p.lattab2 + (size_t)bi * nyb_max * 8;
    const double ps = p.phi_scale;
    // every global load is addressed from the kernel arguments alone (rows below a lower box are zero records or are masked) ...
    const int jb = lane, jbc = min(jb, nyb_max - 1);
    const dbl2_t* rr = reinterpret_cast<const dbl2_t*>(p.rows + (size_t)(tl * nl + k) * lstride + (size_t)jbc * LEC_NSTAT);
    constexpr int kUsed = LEC_S_SPARE / 2;                  // the four spare slots of a record are not read
    dbl2_t R[kUsed];
#pragma unroll
    for (int i = 0; i < kUsed; ++i) R[i] = rr[i];
    static_assert(LEC_S_MT == 0 && LEC_S_MU == 1 && LEC_S_MV == 2, "the neighbour rows' [T] [u] [v] are read as r[0..2]");
    const dbl2_t km2 = *reinterpret_cast<const dbl2_t*>(p.rows + (size_t)(tl * nl + km) * lstride + (size_t)jbc * LEC_NSTAT);
    const dbl2_t kp2 = *reinterpret_cast<const dbl2_t*>(p.rows + (size_t)(tl * nl + kp) * lstride + (size_t)jbc * LEC_NSTAT);
    const dbl2_t* ll = reinterpret_cast<const dbl2_t*>(lt + (size_t)jbc * 8);
    const dbl2_t L0 = ll[0], L1 = ll[1], L2 = ll[2], L3 = ll[3];
    // ... and only now the values that arrive through the scalar cache
    const int nyb = p.box[4 * bi + 3] - p.box[4 * bi + 2] + 1;
    const double pk = p.levtab2[4 * k + 0], pa = p.levtab2[4 * k + 1], pb = p.levtab2[4 * k + 2], pc = p.levtab2[4 * k + 3];
    const bool in = jb < nyb;

    double r[LEC_S_SPARE];
#pragma unroll
    for (int i = 0; i < kUsed; ++i) { r[2 * i] = R[i].x; r[2 * i + 1] = R[i].y; }
    const double cw = L0.x, wphi = L0.y, c = L1.x, tn = L1.y, gra = L2.x, grb = L2.y, grc = L3.x;

    // area means: lec_area_means_kernel's products and butterfly ({[u]} and {[v]} are not used by any term)
    const double aT = wave_sum(in ? cw * r[0] : 0.0);
    const double aW = wave_sum(in ? cw * r[3] : 0.0);
    const double aP = wave_sum(in ? cw * r[4] : 0.0) * ps;
    const double aQ = wave_sum(in ? cw * r[5] : 0.0);
    const double aTm = (km == k) ? aT : wave_sum(in ? cw * km2.x : 0.0);
    const double aTp = (kp == k) ? aT : wave_sum(in ? cw * kp2.x : 0.0);

    // rows j-1 / j+1 from the neighbouring lanes (the first and last row of the box see themselves, like the clamped indices of the
    // general kernel)
    const int lm = max(lane - 1, 0), lp = min(lane + 1, max(nyb - 1, 0));
    double rjm[3], rjp[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) { rjm[s] = __shfl(r[s], lm); rjp[s] = __shfl(r[s], lp); }
    const double cm = __shfl(c, lm), cp = __shfl(c, lp);

    // static stability (thermodynamics.py:55-70)
    double sig = kG * aT / kCp - (pk * kG / kRd) * (pa * aTm + pb * aT + pc * aTp);
    sig = (sig > 0.03) ? sig : 0.03;

    // The row's contributions (every lane computes; lanes below the box are masked where the sums are handed over), in two parts,
    // each followed by its reduction through one small LDS buffer: lane s adds the 64 partials of sum s in four chains (the general
    // kernel's order), applies the divisor and stores.
    double* const out = p.levraw + (size_t)(tl * nl + k) * LEC_NLEVRAW;
    auto reduce_round = [&](const int s0, const int ns) {
        __syncthreads();
        if (lane < ns) {
            const int s = s0 + lane;
            const double* q = part + lane * kPartStride;
            double c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3];
#pragma unroll 4        // (sixteen partials in flight, not sixty: fully unrolled the loads alone hold 120 registers)
            for (int l = 4; l < 64; l += 4) { c0 += q[l]; c1 += q[l + 1]; c2 += q[l + 2]; c3 += q[l + 3]; }
            const double tot = (c0 + c1) + (c2 + c3);
            double div = 1.0;
            const int b = (s >= V_B1) ? (s - V_B1) % 6 : -1;         // boundary pieces: Az Ae | Kz Ke | PhiZ PhiE
            if (s == V_AZ || s == V_AE || b == 0 || b == 1) div = 2 * sig;
            else if (b == 2 || b == 3) div = 2 * kG;
            else if (b == 4 || b == 5) div = kG;
            else if (s == V_CA1) div = 2 * kRe * sig;
            else if (s == V_CA2) div = sig;
            else if (s == V_GZ || s == V_GE) div = kCp * sig;
            out[s] = tot / div;
        }
        __syncthreads();
    };
    static_assert(V_B1 <= kSmallRound && V_SIG - V_B1 <= kSmallRound, "a reduction round must fit the buffer");
#define LEC_BAZ3_REPAIR(jb_) (in ? baz3_repaired_small(p, lt, nyb, tl, k, (jb_)) : 0.0)
#define LEC_ROW_COMMON
#include "lec_level_row.inc"
#undef LEC_ROW_COMMON
    {
        const LaneSums acc{part + lane, 0, in};
#define LEC_ROW_PART_A
#include "lec_level_row.inc"
#undef LEC_ROW_PART_A
    }
    reduce_round(0, V_B1);
    {
#pragma unroll
        for (int i = 0; i < V_SIG - V_B1; ++i) part[i * kPartStride + lane] = 0.0;      // the north-south sums take a row at either end only
        const LaneSums acc{part + lane, V_B1, in};
#define LEC_ROW_PART_B
#include "lec_level_row.inc"
#undef LEC_ROW_PART_B
    }
    reduce_round(V_B1, V_SIG - V_B1);
#undef LEC_BAZ3_REPAIR
    if (lane < LEC_NLEVRAW - V_SIG) out[V_SIG + lane] = (lane == 0) ? sig : 0.0;
}

// functions of level handled by lec_vertical_kernel (one lane each)
enum {
    F_AZ = 0, F_AE, F_KZ, F_KE, F_CZ, F_CA, F_CK, F_CE, F_GZ, F_GE,
    F_B1 = 10, F_B2 = 16, F_B3 = 22, F_COUNT = 28
};

// Builds the F_COUNT functions of level of time step `tl` into fn[f][0..nl) and applies the interpolation half of
// _handle_nans (energy_contents.py:190-208): linear in p across interior gaps, no extrapolation.
// BAz's bottom-top term (F_B3) is the exception: the reference interpolates it per latitude before the area mean
// (lec_level_terms_kernel has done that) and only DROPS the levels that are still NaN (boundary_terms.py:169-176).
// Phase 1: one LEVEL per lane (its levraw record read once, every function evaluated without divergence);
// phase 2: one FUNCTION per lane scans its levels in LDS.  Returns, on lane f < F_COUNT, the number of NaN levels
// of function f found before the repair (0 on the other lanes).
__device__ int build_level_functions(const RedParams& p, int tl, double (*fn)[kMaxNl], int lane) {
    const int nl = p.nl;
    const double* raw = p.levraw + (size_t)tl * nl * LEC_NLEVRAW;
    const double* lv = p.levtab2;
    for (int k = lane; k < nl; k += 64) {
        const double* o = raw + (size_t)k * LEC_NLEVRAW;
        const double c1k = kRd / (lv[4 * k] * kG);   // Rd / (p g), conversion_terms.py:146,172
        fn[F_AZ][k] = o[V_AZ];
        fn[F_AE][k] = o[V_AE];
        fn[F_KZ][k] = o[V_KZ];
        fn[F_KE][k] = o[V_KE];
        fn[F_CZ][k] = -(c1k * o[V_CZ2]);
        fn[F_CA][k] = -(o[V_CA1] + o[V_CA2]);
        fn[F_CK][k] = o[V_CK1] + o[V_CK2] + o[V_CK3] + o[V_CK4] + o[V_CK5];
        fn[F_CE][k] = -(c1k * o[V_CE2]);
        fn[F_GZ][k] = o[V_GZ];
        fn[F_GE][k] = o[V_GE];
#pragma unroll
        for (int i = 0; i < 18; ++i) fn[F_B1 + i][k] = o[V_B1 + i];       // V_B1.. V_B3 are contiguous like F_B1..F_B3
    }
    __syncthreads();
    int nnan = 0;
    if (lane < F_COUNT) {
        double* row = fn[lane];
        for (int k = 0; k < nl; ++k) nnan += isnan(row[k]) ? 1 : 0;
        if (nnan && lane != F_B3) {
            int last_ok = -1;
            for (int k = 0; k < nl; ++k) {
                if (!isnan(row[k])) { last_ok = k; continue; }
                int nxt = k + 1;
                while (nxt < nl && isnan(row[nxt])) ++nxt;
                if (last_ok >= 0 && nxt < nl) {
                    const double xl = lv[4 * last_ok], xr = lv[4 * nxt], yl = row[last_ok], yr = row[nxt];
                    const double slope = (yr - yl) / (xr - xl);
                    for (int q = k; q < nxt; ++q) row[q] = slope * (lv[4 * q] - xl) + yl;
                }
                k = nxt - 1;
            }
        }
    }
    return nnan;
}

// grid (t_count), block 64: marks the levels that are still NaN after the interpolation at this time step.
// xarray's dropna(dim=level) on a [time, level] array drops such a level for EVERY time step.
__global__ void __launch_bounds__(64) lec_dropmask_kernel(const RedParams p) {
    __shared__ double fn[F_COUNT][kMaxNl];
    const int tl = blockIdx.x, lane = threadIdx.x;
    if (build_level_functions(p, tl, fn, lane)) {
        for (int k = 0; k < p.nl; ++k)
            if (isnan(fn[lane][k])) atomicOr(&p.dropmask[lane * p.nl + k], 1);
    }
}

// grid (t_count), block 64
__global__ void __launch_bounds__(64) lec_vertical_kernel(const RedParams p) {
    __shared__ double fn[F_COUNT][kMaxNl];
    __shared__ double res[F_COUNT];
    __shared__ int nans[64];
    const int tl = blockIdx.x, lane = threadIdx.x, nl = p.nl;
    const int bi = (p.n_box == 1) ? 0 : tl;
    const double* raw = p.levraw + (size_t)tl * nl * LEC_NLEVRAW;
    const double* lv = p.levtab2;

    int nnan = build_level_functions(p, tl, fn, lane);
    if (lane < F_COUNT) {
        const int f = lane;
        // the dropping half of _handle_nans: levels that are still NaN (here, or at any time step in the fixed framework)
        int k0 = 0, k1 = nl - 1;
        if (p.drop_any_time) {
            const int* dm = p.dropmask + f * nl;
            bool any = false;
            for (int k = 0; k < nl; ++k) if (dm[k]) { fn[f][k] = nan(""); any = true; }
            if (any && !nnan) nnan = 1;
        }
        if (nnan) {
            while (k0 < nl && isnan(fn[f][k0])) ++k0;
            while (k1 >= 0 && isnan(fn[f][k1])) --k1;
        }
        double r;
        if (k0 > k1) {
            // no level left: the reference integrates an EMPTY array after dropna -- xarray's integrate gives 0.0 --, while its
            // bottom-top terms would stop at .isel(level=-1) of nothing (IndexError); here those are NaN
            r = f >= F_B3 ? nan("") : 0.0;
        } else if (f >= F_B3) {
            r = fn[f][k1] - fn[f][k0];                       // .isel(level=-1) - .isel(level=0)
        } else {
            r = 0.0;
            for (int k = k0; k < k1; ++k) r += (lv[4 * (k + 1)] - lv[4 * k]) * 0.5 * (fn[f][k + 1] + fn[f][k]);
        }
        res[f] = r;
    }
    nans[lane] = nnan;
    __syncthreads();

    if (lane == 0) {
        const double c1 = p.boxtab2[4 * bi + 0], c2 = p.boxtab2[4 * bi + 1];
        double* s = p.scalars + (size_t)tl * (size_t)p.sstride;
        s[0] = res[F_AZ];
        s[1] = res[F_AE];
        s[2] = res[F_KZ] / (2 * kG);
        s[3] = res[F_KE] / (2 * kG);
        s[4] = res[F_CZ];
        s[5] = res[F_CA];
        s[6] = res[F_CK] / kG;
        s[7] = res[F_CE];
        for (int i = 0; i < 6; ++i) s[8 + i] = res[F_B1 + i] * c1 + res[F_B2 + i] * c2 - res[F_B3 + i];
        s[14] = res[F_GZ];
        s[15] = res[F_GE];
        int tot = 0;
        for (int i = 0; i < F_COUNT; ++i) tot += nans[i];
        p.nanflag[tl] = tot;
    }

    // per-level tables, order of lec_fixed_framework.py:172-194
    double* L = p.levels + (size_t)tl * (size_t)p.lstride;
    for (int k = lane; k < nl; k += 64) {
        const double* o = raw + (size_t)k * LEC_NLEVRAW;
        const double c1k = kRd / (lv[4 * k] * kG);
        L[0 * nl + k] = fn[F_AZ][k];
        L[1 * nl + k] = fn[F_AE][k];
        L[2 * nl + k] = fn[F_KZ][k];
        L[3 * nl + k] = fn[F_KE][k];
        L[4 * nl + k] = fn[F_GE][k];
        L[5 * nl + k] = fn[F_GZ][k];
        L[6 * nl + k] = fn[F_CZ][k];
        L[7 * nl + k] = c1k;
        L[8 * nl + k] = o[V_CZ2];
        L[9 * nl + k] = fn[F_CA][k];
        L[10 * nl + k] = o[V_CA1];
        L[11 * nl + k] = o[V_CA2];
        L[12 * nl + k] = fn[F_CE][k];
        L[13 * nl + k] = c1k;
        L[14 * nl + k] = o[V_CE2];
        L[15 * nl + k] = fn[F_CK][k];
        L[16 * nl + k] = o[V_CK1];
        L[17 * nl + k] = o[V_CK2];
        L[18 * nl + k] = o[V_CK3];
        L[19 * nl + k] = o[V_CK4];
        L[20 * nl + k] = o[V_CK5];
    }
}

}  // namespace

static int reduce_impl(const lec_reduce_args* a, bool mask_only, const char* who) {
    if (!a) return lec_set_error(LEC_ERR_ARG, "lec_reduce: null args");
    if (a->stage < LEC_STAGE_BOTH || a->stage > LEC_STAGE_VERTICAL) return lec_set_error(LEC_ERR_ARG, "lec_reduce / lec_dropmask: stage must be 0 (both), 1 (levels) or 2 (vertical)");
    const bool levels = a->stage != LEC_STAGE_VERTICAL, vertical = a->stage != LEC_STAGE_LEVELS;
    if (mask_only && a->stage == LEC_STAGE_LEVELS) return lec_set_error(LEC_ERR_ARG, "lec_dropmask: stage LEC_STAGE_LEVELS forms no mask (use lec_reduce)");
    if (!a->boxtab2_d || !a->levtab2_d || !a->levraw_d)
        return lec_set_error(LEC_ERR_ARG, "lec_reduce / lec_dropmask: null pointer argument");
    if (levels && (!a->rows_d || !a->box_d || !a->lattab2_d || !a->am_d)) return lec_set_error(LEC_ERR_ARG, "lec_reduce / lec_dropmask: null pointer argument");
    if (vertical && !mask_only && (!a->scalars_d || !a->levels_d || !a->nanflag_d)) return lec_set_error(LEC_ERR_ARG, "lec_reduce: null output pointer");
    if (a->t_count < 1 || a->nl < 2 || a->nyb_max < 2) return lec_set_error(LEC_ERR_ARG, "lec_reduce: needs t_count>=1, nl>=2, nyb_max>=2");
    if (a->nl > kMaxNl) return lec_set_error(LEC_ERR_UNSUPPORTED, "lec_reduce: more than 160 levels");
    // HIP refuses a launch whose grid x block reaches 2^32 threads: (nl * t_count) workgroups of 64 threads and, for the vertical
    // half, t_count workgroups of 64
    if ((long long)a->t_count * a->nl * 64 > 0xffffffffLL)
        return lec_set_error(LEC_ERR_UNSUPPORTED, "lec_reduce: nl * t_count must stay below 2^26 in one call (cut the series into several calls)");
    if (a->n_box != 1 && a->n_box != a->t_count) return lec_set_error(LEC_ERR_ARG, "lec_reduce: n_box must be 1 or t_count");
    if (a->drop_any_time < 0 || a->drop_any_time > 2) return lec_set_error(LEC_ERR_ARG, "lec_reduce: drop_any_time must be 0, 1 or 2");
    if (levels && ((reinterpret_cast<uintptr_t>(a->rows_d) | reinterpret_cast<uintptr_t>(a->lattab2_d)) & 15))
        return lec_set_error(LEC_ERR_ARG, "lec_reduce / lec_dropmask: rows_d and lattab2_d must be 16-byte aligned");
    if (a->scalars_stride < 0 || a->levels_stride < 0 || (a->scalars_stride && a->scalars_stride < LEC_NSCALAR) ||
        (a->levels_stride && a->levels_stride < (long long)LEC_NLEVTAB * a->nl))
        return lec_set_error(LEC_ERR_ARG, "lec_reduce: scalars_stride / levels_stride must be 0 (dense) or at least one record long");
    if (vertical && (a->drop_any_time || mask_only) && !a->dropmask_d) return lec_set_error(LEC_ERR_ARG, "lec_reduce / lec_dropmask: drop_any_time needs dropmask_d");
    (void)who;
    RedParams p;
    p.rows = a->rows_d; p.t_count = a->t_count; p.nl = a->nl; p.n_box = a->n_box; p.nyb_max = a->nyb_max;
    p.box = a->box_d; p.boxtab2 = a->boxtab2_d; p.lattab2 = a->lattab2_d; p.levtab2 = a->levtab2_d;
    p.phi_scale = a->phi_scale; p.am = a->am_d; p.levraw = a->levraw_d; p.scalars = a->scalars_d;
    p.levels = a->levels_d; p.nanflag = a->nanflag_d;
    p.sstride = a->scalars_stride ? a->scalars_stride : LEC_NSCALAR;
    p.lstride = a->levels_stride ? a->levels_stride : (long long)LEC_NLEVTAB * a->nl;
    p.drop_any_time = (a->drop_any_time || mask_only) ? 1 : 0; p.dropmask = a->dropmask_d;
    hipStream_t st = (hipStream_t)a->stream;
    if (levels) {
        const dim3 grid2((unsigned)a->nl * (unsigned)a->t_count);      // one workgroup per (time step, level), level fastest
        if (a->nyb_max <= kSmallRows) {
            hipLaunchKernelGGL(lec_level_small_kernel, grid2, dim3(64), 0, st, p);       // forms its own area means; am_d is not used
        } else {
            hipLaunchKernelGGL(lec_area_means_kernel, grid2, dim3(64), 0, st, p);
            hipLaunchKernelGGL(lec_level_terms_kernel, grid2, dim3(64), 0, st, p);
        }
    }
    if (vertical) {
        if (mask_only || a->drop_any_time == 1) {
            if (hipMemsetAsync(p.dropmask, 0, sizeof(int) * F_COUNT * a->nl, st) != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, "lec_reduce: hipMemsetAsync failed");
            hipLaunchKernelGGL(lec_dropmask_kernel, dim3(a->t_count), dim3(64), 0, st, p);
        }
        if (!mask_only) hipLaunchKernelGGL(lec_vertical_kernel, dim3(a->t_count), dim3(64), 0, st, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    return LEC_OK;
}

extern "C" int lec_reduce(const lec_reduce_args* a) { return reduce_impl(a, false, "lec_reduce"); }
extern "C" int lec_dropmask(const lec_reduce_args* a) { return reduce_impl(a, true, "lec_dropmask"); }
