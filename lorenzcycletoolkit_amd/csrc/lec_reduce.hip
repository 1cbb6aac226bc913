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
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// grid (nl, t_count), block 64.  Like lec_level_terms_kernel: the record loads are addressed from the kernel arguments alone (nyb_max
// rows) and issued before the box height arrives; rows below a lower box are masked out afterwards.
__global__ void __launch_bounds__(64) lec_area_means_kernel(const RedParams p) {
    const int k = blockIdx.x, tl = blockIdx.y, lane = threadIdx.x;
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

// grid (nl, t_count), block 64.  A wave's life here is a chain of memory round trips (a few hundred arithmetic instructions between
// them), so the kernel is arranged to have TWO: every global load -- the level's records, the latitude table, the records of the
// levels above and below -- is addressed from the kernel arguments alone (the staging covers nyb_max rows; the rows below a lower box
// are zero records) and issued before anything waits for the area means or the box height.
__global__ void __launch_bounds__(64) lec_level_terms_kernel(const RedParams p) {
    const int k = blockIdx.x, tl = blockIdx.y, lane = threadIdx.x;
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

        const double mT = r[LEC_S_MT], mU = r[LEC_S_MU], mV = r[LEC_S_MV], mW = r[LEC_S_MW];
        const double mP = r[LEC_S_MP] * ps, mQ = r[LEC_S_MQ];
        const double Ts = mT - aT, Ws = mW - aW, Ps = mP - aP, Qs = mQ - aQ;      // X* = [X] - {[X]}
        const double sTT = r[LEC_S_TT], sUU = r[LEC_S_UU], sVV = r[LEC_S_VV], sVT = r[LEC_S_VT], sWT = r[LEC_S_WT];
        const double sUV = r[LEC_S_UV], sWU = r[LEC_S_WU], sWV = r[LEC_S_WV], sWP = r[LEC_S_WP] * ps, sQT = r[LEC_S_QT];

        const double dphiTc = gra * (rjm[LEC_S_MT] - aT) * cm + grb * Ts * c + grc * (rjp[LEC_S_MT] - aT) * cp;
        const double dphiUc = gra * (rjm[LEC_S_MU] / cm) + grb * (mU / c) + grc * (rjp[LEC_S_MU] / cp);
        const double dphiV = gra * rjm[LEC_S_MV] + grb * mV + grc * rjp[LEC_S_MV];
        const double dpT = pa * (km2.x - aTm) + pb * Ts + pc * (kp2.x - aTp);
        const double dpU = pa * km2.y + pb * mU + pc * kp2.y;

        acc[V_AZ] += cw * (Ts * Ts);
        acc[V_AE] += cw * sTT;
        acc[V_KZ] += cw * (mU * mU + mV * mV);
        acc[V_KE] += cw * (sUU + sVV);
        acc[V_CZ2] += cw * (Ws * Ts);
        acc[V_CE2] += cw * sWT;
        acc[V_CA1] += cw * (sVT * dphiTc);
        acc[V_CA2] += cw * (sWT * dpT);
        acc[V_CK1] += cw * (c * sUV / kRe * dphiUc);
        acc[V_CK2] += cw * (sVV / kRe * dphiV);
        acc[V_CK3] += cw * (tn * sUU * mV / kRe);
        acc[V_CK4] += cw * (sWU * dpU);
        acc[V_CK5] += cw * (sWV * dpU);   // sic: d[u]/dp, conversion_terms.py:225-229
        acc[V_GZ] += cw * (Qs * Ts);
        acc[V_GE] += cw * sQT;

        // east-west pieces, plain trapezoid over phi (boundary_terms.py:135-147,188-197,237-246,287-296,337-344,377-388)
        const double TW = r[LEC_S_TW], TE = r[LEC_S_TE], uW = r[LEC_S_UW], uE = r[LEC_S_UE], vW = r[LEC_S_VW], vE = r[LEC_S_VE];
        const double TpW = TW - mT, TpE = TE - mT;
        const double upW = uW - mU, upE = uE - mU, vpW = vW - mV, vpE = vE - mV;
        const double EW_ = upW * upW + vpW * vpW, EE_ = upE * upE + vpE * vpE;
        const double KW_ = uW * uW + vW * vW - EW_, KE_ = uE * uE + vE * vE - EE_;
        acc[V_B1 + 0] += wphi * (((2 * Ts * TpE * uE) + (Ts * Ts * uE)) - ((2 * Ts * TpW * uW) + (Ts * Ts * uW)));
        acc[V_B1 + 1] += wphi * (uE * (TpE * TpE) - uW * (TpW * TpW));
        acc[V_B1 + 2] += wphi * (uE * KE_ - uW * KW_);
        acc[V_B1 + 3] += wphi * (uE * EE_ - uW * EW_);
        acc[V_B1 + 4] += wphi * (mV * Ps);
        acc[V_B1 + 5] += wphi * (vpE * Ps - vpW * Ps);

        // north-south pieces (boundary_terms.py:150-163,200-212,249-262,299-310,347-356,390-399)
        const double sgn = (jb == nyb - 1 ? 1.0 : 0.0) - (jb == 0 ? 1.0 : 0.0);
        if (sgn != 0.0) {
            acc[V_B2 + 0] += sgn * (((sVT * 2 * Ts) + (Ts * Ts * mV)) * c);
            acc[V_B2 + 1] += sgn * (r[LEC_S_VTT] * c);
            acc[V_B2 + 2] += sgn * (r[LEC_S_KV] * c);
            acc[V_B2 + 3] += sgn * (r[LEC_S_EV] * c);
            acc[V_B2 + 4] += sgn * (mV * Ps * c);
            acc[V_B2 + 5] += sgn * (mV * Ps * c);   // BPhiE term 2 uses zonal means, boundary_terms.py:390
        }

        // bottom-top pieces, area means (boundary_terms.py:165-176,214-221,264-271,312-318,358-363,401-413)
        double x3 = (2 * sWT) * Ts + mW * (Ts * Ts);
        if (isnan(x3)) x3 = baz3_repaired(p, am, tl, k, jb);        // per latitude, before the area mean (the reference's order)
        acc[V_B3 + 0] += cw * x3;
        acc[V_B3 + 1] += cw * r[LEC_S_WTT];
        acc[V_B3 + 2] += cw * r[LEC_S_KW];
        acc[V_B3 + 3] += cw * r[LEC_S_EW];
        acc[V_B3 + 4] += cw * (Ws * Ps);
        acc[V_B3 + 5] += cw * sWP;
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
            r = nan("");
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
        double* s = p.scalars + (size_t)tl * LEC_NSCALAR;
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
    double* L = p.levels + (size_t)tl * LEC_NLEVTAB * nl;
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
    if (!a->rows_d || !a->box_d || !a->boxtab2_d || !a->lattab2_d || !a->levtab2_d || !a->am_d || !a->levraw_d)
        return lec_set_error(LEC_ERR_ARG, "lec_reduce / lec_dropmask: null pointer argument");
    if (!mask_only && (!a->scalars_d || !a->levels_d || !a->nanflag_d)) return lec_set_error(LEC_ERR_ARG, "lec_reduce: null output pointer");
    if (a->t_count < 1 || a->nl < 2 || a->nyb_max < 2) return lec_set_error(LEC_ERR_ARG, "lec_reduce: needs t_count>=1, nl>=2, nyb_max>=2");
    if (a->nl > kMaxNl) return lec_set_error(LEC_ERR_UNSUPPORTED, "lec_reduce: more than 160 levels");
    if (a->t_count > 65535) return lec_set_error(LEC_ERR_UNSUPPORTED, "lec_reduce: more than 65535 time steps in one call");
    if (a->n_box != 1 && a->n_box != a->t_count) return lec_set_error(LEC_ERR_ARG, "lec_reduce: n_box must be 1 or t_count");
    if (a->drop_any_time < 0 || a->drop_any_time > 2) return lec_set_error(LEC_ERR_ARG, "lec_reduce: drop_any_time must be 0, 1 or 2");
    if ((reinterpret_cast<uintptr_t>(a->rows_d) | reinterpret_cast<uintptr_t>(a->lattab2_d)) & 15)
        return lec_set_error(LEC_ERR_ARG, "lec_reduce / lec_dropmask: rows_d and lattab2_d must be 16-byte aligned");
    if ((a->drop_any_time || mask_only) && !a->dropmask_d) return lec_set_error(LEC_ERR_ARG, "lec_reduce / lec_dropmask: drop_any_time needs dropmask_d");
    (void)who;
    RedParams p;
    p.rows = a->rows_d; p.t_count = a->t_count; p.nl = a->nl; p.n_box = a->n_box; p.nyb_max = a->nyb_max;
    p.box = a->box_d; p.boxtab2 = a->boxtab2_d; p.lattab2 = a->lattab2_d; p.levtab2 = a->levtab2_d;
    p.phi_scale = a->phi_scale; p.am = a->am_d; p.levraw = a->levraw_d; p.scalars = a->scalars_d;
    p.levels = a->levels_d; p.nanflag = a->nanflag_d;
    p.drop_any_time = (a->drop_any_time || mask_only) ? 1 : 0; p.dropmask = a->dropmask_d;
    hipStream_t st = (hipStream_t)a->stream;
    const dim3 grid2(a->nl, a->t_count);
    hipLaunchKernelGGL(lec_area_means_kernel, grid2, dim3(64), 0, st, p);
    hipLaunchKernelGGL(lec_level_terms_kernel, grid2, dim3(64), 0, st, p);
    if (mask_only || a->drop_any_time == 1) {
        if (hipMemsetAsync(p.dropmask, 0, sizeof(int) * F_COUNT * a->nl, st) != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, "lec_reduce: hipMemsetAsync failed");
        hipLaunchKernelGGL(lec_dropmask_kernel, dim3(a->t_count), dim3(64), 0, st, p);
    }
    if (!mask_only) hipLaunchKernelGGL(lec_vertical_kernel, dim3(a->t_count), dim3(64), 0, st, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    return LEC_OK;
}

extern "C" int lec_reduce(const lec_reduce_args* a) { return reduce_impl(a, false, "lec_reduce"); }
extern "C" int lec_dropmask(const lec_reduce_args* a) { return reduce_impl(a, true, "lec_dropmask"); }
