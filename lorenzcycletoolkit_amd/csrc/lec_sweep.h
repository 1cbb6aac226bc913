// lec_sweep.h -- pieces shared by the single-sweep stage-1 kernels (lec_rowsweep.hip, lec_rowblock.hip):
// the 20 shifted sums of one element, the in-wave neighbour shifts and the row epilogue.
#ifndef LEC_SWEEP_H
#define LEC_SWEEP_H

#include <hip/hip_runtime.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"
#include "lec_rowcommon.h"

namespace lec {

constexpr int kNA = 20;     // shifted sums per row
constexpr int kRound = 8;    // statistics per reduction round of a one-wave row: 8 statistics x 8 lanes fill the wave (3 rounds for 20)
constexpr int kRowShift = 3;  // log2 of the lanes that cooperate on one statistic

template <typename TIN, int VEC, int MODE>
constexpr int sweep_min_waves() {
#if LEC_MINW > 0
    return LEC_MINW;
#else
    return (sizeof(TIN) == 4 && MODE != 0 && VEC == 4) ? LEC_MINW_SINGLE - 1 : LEC_MINW_SINGLE;   // float4 all-terms: 149 VGPRs, 3 waves (a 128 cap spills: 13.4 vs 10.7 ms)
#endif
}

// value of the neighbouring lane of the wave (DPP wave shift: VALU only, no LDS, no memory); lanes at
// the end of the wave keep `edge`
__device__ __forceinline__ double from_prev_lane(double v, double edge) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_next_lane(double v, double edge) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// The element arithmetic below is shared by two kernels whose results must be bit-identical (shard
// invariance and kernel choice are tested bit-exact), so nothing is left to the compiler's contraction
// heuristics: contraction is off and every fused multiply-add is written out.

// c0 x0 + c1 x1 + cm xm with the two outer products rounded before their (commutative) add: the result
// does not depend on which neighbour is called x0 -- the row-block kernel swaps sides per wave
__device__ __forceinline__ double stencil3(double c0, double x0, double c1, double x1, double cm, double xm) {
#pragma clang fp contract(off)
    const double p0 = c0 * x0, p1 = c1 * x1;
    return fma(cm, xm, p0 + p1);
}

// the 20 shifted sums of one element (see lec_rowsweep.hip); UNIT: weight 1 (no multiplies by w)
template <bool UNIT>
__device__ __forceinline__ void accum20(double (&acc)[kNA], double w, double a, double b, double c, double d, double ee, double f) {
#pragma clang fp contract(off)
    const double wa = UNIT ? a : w * a, wb = UNIT ? b : w * b, wc = UNIT ? c : w * c, wd = UNIT ? d : w * d;
    const double waa = wa * a, wbb = wb * b, wcc = wc * c;
    acc[0] += wa; acc[1] += wb; acc[2] += wc; acc[3] += wd;
    if (UNIT) { acc[4] += ee; acc[5] += f; } else { acc[4] = fma(w, ee, acc[4]); acc[5] = fma(w, f, acc[5]); }
    acc[6] += waa; acc[7] += wbb; acc[8] += wcc;
    acc[9] = fma(wc, a, acc[9]);       // <ca>
    acc[10] = fma(wd, a, acc[10]);     // <da>
    acc[11] = fma(wb, c, acc[11]);     // <bc>
    acc[12] = fma(wd, b, acc[12]);     // <db>
    acc[13] = fma(wd, c, acc[13]);     // <dc>
    acc[14] = fma(wd, ee, acc[14]);    // <de>
    acc[15] = fma(wa, f, acc[15]);     // <fa>
    acc[16] = fma(waa, c, acc[16]);    // <caa>
    acc[17] = fma(waa, d, acc[17]);    // <daa>
    const double wee = wbb + wcc;      // only the sums <bbc>+<ccc> and <bbd>+<ccd> are ever needed ([Ev], [Ew])
    acc[18] = fma(wee, c, acc[18]);    // <(bb+cc) c>
    acc[19] = fma(wee, d, acc[19]);    // <(bb+cc) d>
}

// wave-uniform constants of one row sweep
struct SweepRow {
    int nxb;                        // box points in the row
    double cT, cU, cV, cW, cP;      // shifts: the row's first box element
    double cTf, cTb;                // the same for the T rows one time step ahead / back (cross-time covariances)
    double cx;                      // uniform longitudes: 0.5 / h_deg / dx_j  (centred d/dlon -> d/dx)
    double inv_dx;                  // table longitudes: 1 / dx_j
    const double* wl;               // table longitudes: trapezoid weights
    const double* gl;               // table longitudes: d/dlon coefficients
};

// T neighbours of the diabatic-heating stencils as loaded (raw storage type, converted at use), and their
// wave-uniform coefficients: per dimension x0 / x1 are the two neighbours in whatever order the kernel fetched
// them, cm multiplies the centre value.  tf / tb: the T row one time step ahead / back.
template <typename OP, int VEC>
struct QRaw { OP tf[VEC], tb[VEC], k0[VEC], k1[VEC], j0[VEC], j1[VEC]; };
struct QCoef { double tb_, tf_, tm, k0, k1, km, j0, j1, jm; };      // tb_ / tf_ multiply T(t-1) / T(t+1) (QMODE 1)

constexpr int kNX = 4;      // cross-time sums: <a a+>, <a+>, <a a->, <a->

// element q of a vector (see sweep_elems).  EDGE = false: the element lies strictly inside its row; true: the general form (selects).
template <int VEC, bool UNIFORM, bool EDGE, int QMODE, bool BOTH, typename OP>
__device__ __forceinline__ void sweep_one(const int q, double (&acc)[kNA], double (&xacc)[kNX], const SweepRow& r, int e0, bool lane_in,
                                          const OP (&fT)[VEC], const OP (&fU)[VEC], const OP (&fV)[VEC],
                                          const OP (&fW)[VEC], const OP (&fP)[VEC], double tl_edge, double tr_edge,
                                          const QRaw<OP, VEC>& qr, const QCoef& qc) {
#pragma clang fp contract(off)
    const int e = e0 + q;
    const bool inside = !EDGE || ((e >= 0) && (e < r.nxb) && lane_in);
    const bool first = EDGE && inside && (e == 0), last = EDGE && inside && (e == r.nxb - 1);
    double w = 1.0;
    if (UNIFORM) { if (EDGE) w = inside ? ((first || last) ? 0.5 : 1.0) : 0.0; }
    else w = EDGE ? (inside ? r.wl[min(max(e, 0), r.nxb - 1)] : 0.0) : r.wl[e];
    const double Tc = (double)fT[q];
    const double Tv = inside ? Tc : r.cT;
    const double Uv = inside ? (double)fU[q] : r.cU;
    const double Vv = inside ? (double)fV[q] : r.cV;
    const double Wv = inside ? (double)fW[q] : r.cW;
    const double Pv = inside ? (double)fP[q] : r.cP;
    const double a = Tv - r.cT;
    double f = 0.0;
    if (QMODE != 0) {
        const double Tl = (q == 0) ? tl_edge : (double)fT[q > 0 ? q - 1 : 0];
        const double Tr = (q == VEC - 1) ? tr_edge : (double)fT[q < VEC - 1 ? q + 1 : q];
        double adv;                                     // u dT/dx
        if (UNIFORM) {
            double d = Tr - Tl;
            if (EDGE) d = first ? 2.0 * (Tr - Tv) : (last ? 2.0 * (Tv - Tl) : d);      // one-sided at the row ends
            adv = (Uv * r.cx) * d;
        } else {
            const int ec = EDGE ? min(max(e, 0), r.nxb - 1) : e;
            adv = Uv * fma(r.gl[3 * ec + 2], Tr, fma(r.gl[3 * ec + 1], Tv, r.gl[3 * ec + 0] * Tl)) * r.inv_dx;
        }
        const double sP = stencil3(qc.j0, (double)qr.j0[q], qc.j1, (double)qr.j1[q], qc.jm, Tc);
        const double sS = stencil3(qc.k0, (double)qr.k0[q], qc.k1, (double)qr.k1[q], qc.km, Tc);
        double rest = adv;
        if (QMODE == 1) rest = stencil3(qc.tb_, (double)qr.tb[q], qc.tf_, (double)qr.tf[q], qc.tm, Tc) + adv;
        if (QMODE == 2) rest = (double)qr.tf[q] + adv;
        f = fma(-Wv, sS, fma(Vv, sP, rest));
        if (EDGE) f = inside ? f : 0.0;
        if (QMODE == 3) {
            // the product a * a+ is rounded before it is weighted, so the row at t and the row at t+1 (as its backward
            // covariance) form bit-identical sums: results do not depend on where a shard or a chunk starts
            // ... and the weighted sum of the neighbour row's deviations is formed exactly like the row's own <a> (accum20: the product
            // w a rounded, then added), because lec_qtime_kernel uses the one in place of the other: <a+> of row t is <a> of row t + 1.
            // (A fused multiply-add here differs from it in the last bit for weights that are not powers of two -- stretched
            // longitudes -- and made shards of such a series differ from the whole by an ulp; tests/soak_gpu.py found it.)
            const double af = inside ? (double)qr.tf[q] - r.cTf : 0.0;
            const double pf = a * af;
            if (UNIFORM && !EDGE) { xacc[0] += pf; xacc[1] += af; }
            else { xacc[0] = fma(w, pf, xacc[0]); const double waf = w * af; xacc[1] += waf; }
            if (BOTH) {
                const double ab = inside ? (double)qr.tb[q] - r.cTb : 0.0;
                const double pb = a * ab;
                if (UNIFORM && !EDGE) { xacc[2] += pb; xacc[3] += ab; }
                else { xacc[2] = fma(w, pb, xacc[2]); const double wab = w * ab; xacc[3] += wab; }
            }
        }
    }
    accum20<UNIFORM && !EDGE>(acc, w, a, Uv - r.cU, Vv - r.cV, Wv - r.cW, Pv - r.cP, f);
}

// One vector (VEC consecutive longitudes starting at box element e0) of every operand -> the 20 sums.
//   EDGE = false: every element of the trip lies strictly inside the row (1 <= e <= nxb - 2): no selects,
//                 and with uniform longitudes the weight is the constant 1 (the row epilogue multiplies by h).
//   EDGE = true : the first / last trips: half weights at the row ends, lanes outside the row contribute 0; lanes whose whole
//                 vector lies past the row (the tail of the last trip) sit the element loop out instead of adding exact zeros
//                 (same bits, and 20 fewer VGPRs in the fp32 all-terms instantiation).  Measured and NOT kept (LEC_EDGE_BALLOT,
//                 profiles/r03_notes.md): per element, a wave-uniform test "is any lane's element a row end or outside" choosing
//                 between the plain and the general form -- the same bits with 190 fewer instructions per 1440-point fp32 row, but
//                 both forms side by side cost 9-30 VGPRs, the capped instantiations spill, and every configuration got slower.
//   QMODE: 0 no Q;
//          1 dT/dt = ta T(t-1) + tb T(t) + tc T(t+1) per point (moving boxes: the neighbours in time sum over other boxes);
//          2 dT/dt read from a cube (in qr.tf): f = Q / cp complete;
//          3 (one fixed box) the same dT/dt as 1, but NOT formed per point: f holds only the
//            advective and static-stability parts; the time-derivative parts of [Q] and [Q'T'] are linear in T(t+-1),
//            so they follow from the zonal means and the cross-time covariances [T'(t) T'(t+1)], [T'(t) T'(t-1)]
//            (lec_qtime_kernel).  A row therefore reads T(t+1) only -- the covariance with T(t-1) is the previous
//            row's forward covariance; BOTH: the row has no processed predecessor and forms the backward one too.
// With uniform longitudes the sums carry RELATIVE trapezoid weights (1, 1/2, 0); Q is accumulated without
// the factor cp (applied in the epilogue).  Operands arrive in their storage type OP and are converted here.
template <int VEC, bool UNIFORM, bool EDGE, int QMODE, bool BOTH, typename OP>
__device__ __forceinline__ void sweep_elems(double (&acc)[kNA], double (&xacc)[kNX], const SweepRow& r, int e0, bool lane_in,
                                            const OP (&fT)[VEC], const OP (&fU)[VEC], const OP (&fV)[VEC],
                                            const OP (&fW)[VEC], const OP (&fP)[VEC], double tl_edge, double tr_edge,
                                            const QRaw<OP, VEC>& qr, const QCoef& qc) {
    if (EDGE && !lane_in) return;          // the lane's vector lies wholly past the row: it would add exact zeros
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        if (EDGE) {
            const int e = e0 + q;
            const bool special = (unsigned)(e - 1) >= (unsigned)(r.nxb - 2);        // e <= 0 or e >= nxb - 1: a row end, or outside the row
#if defined(LEC_EDGE_BALLOT) && LEC_EDGE_BALLOT
            if (__builtin_amdgcn_ballot_w64(special) != 0)
                sweep_one<VEC, UNIFORM, true, QMODE, BOTH>(q, acc, xacc, r, e0, lane_in, fT, fU, fV, fW, fP, tl_edge, tr_edge, qr, qc);
            else
                sweep_one<VEC, UNIFORM, false, QMODE, BOTH>(q, acc, xacc, r, e0, lane_in, fT, fU, fV, fW, fP, tl_edge, tr_edge, qr, qc);
#else
            (void)special;
            sweep_one<VEC, UNIFORM, true, QMODE, BOTH>(q, acc, xacc, r, e0, lane_in, fT, fU, fV, fW, fP, tl_edge, tr_edge, qr, qc);
#endif
        } else {
            sweep_one<VEC, UNIFORM, false, QMODE, BOTH>(q, acc, xacc, r, e0, lane_in, fT, fU, fV, fW, fP, tl_edge, tr_edge, qr, qc);
        }
        // four-element vectors: finish one element before starting the next, or the scheduler interleaves all four and
        // their temporaries push the kernel past 128 VGPRs
        if (VEC > 2) __builtin_amdgcn_sched_barrier(0);
    }
}

// block sums of the accumulators (rounds through the same LDS tile), scaled by `scale` (1 / xlength,
// times the longitude step when the sums carry relative weights), then the centred
// statistics from the shifted sums (lanes 0..21) written to the row record.  XCOV: the cross-time sums go to the
// record's scratch slots 28..31 as [T'(t)T'(t+1)], [T](t+1), [T'(t)T'(t-1)], [T](t-1) for lec_qtime_kernel.
// Contraction is off: the epilogue must give the same bits in every kernel instantiation.
template <int NTHR, int NR = kRound, bool XCOV = false>
__device__ __forceinline__ void finish_row(const double (&acc)[kNA], const double (&xacc)[kNX], double* red, double* tot, int tid,
                                           double scale, const SweepRow& r, double* __restrict__ out, bool store = true) {
#pragma clang fp contract(off)
    constexpr int rshift = (NTHR == 64) ? kRowShift : red_rshift(NTHR);
    static_assert(NR << rshift <= NTHR, "a reduction round must fit the row's threads");
    const double cT = r.cT, cU = r.cU, cV = r.cV, cW = r.cW, cP = r.cP;
    // rounds of NR statistics through the same LDS tile; the cross-time sums ride in the slots the last round has left
    // (20 + 4 = 3 rounds of 8: one round less than a round of their own -- every statistic is summed on its own, so the bits are the same)
    constexpr int NV = kNA + (XCOV ? kNX : 0);
#pragma unroll
    for (int r0 = 0; r0 < NV; r0 += NR) {
        double h[NR];
#pragma unroll
        for (int s = 0; s < NR; ++s) {
            const int v = r0 + s;
            h[s] = (v < kNA) ? acc[(v < kNA) ? v : 0] : ((XCOV && v < NV) ? xacc[(XCOV && v >= kNA && v < NV) ? v - kNA : 0] : 0.0);
        }
        const double t0 = block_sums<NR, NTHR, rshift>(h, red, tid);
        const int v = r0 + (tid >> rshift);
        if ((tid & ((1 << rshift) - 1)) == 0 && (tid >> rshift) < NR && v < NV)
            tot[v] = t0 * ((v == 5 || v == 15) ? scale * kCp : scale);   // <f>, <fa>: Q = cp f
        row_sync<NTHR>();
    }
    if (tid < 22) {
        // lanes 0..15: one branch-free form  o = tot[L] + shift_L - tot[ib] * tot[ic]  (means: product masked out;
        // second moments: no shift); ib / ic are 3-bit codes packed per lane: the d-values tot[0..5]
        //   L:   6 7 8 9 10 11 12 13 14 15
        //   ib:  0 1 2 2  3  1  3  3  3  5       [T'T'] [u'u'] [v'v'] [v'T'] [w'T'] [u'v'] [w'u'] [w'v'] [w'Phi'] [Q'T']
        //   ic:  0 1 2 0  0  2  1  2  4  0
        constexpr unsigned long long kIB = 0x0ULL | (0ULL << 18) | (1ULL << 21) | (2ULL << 24) | (2ULL << 27) | (3ULL << 30) | (1ULL << 33) |
                                           (3ULL << 36) | (3ULL << 39) | (3ULL << 42) | (5ULL << 45);
        constexpr unsigned long long kIC = 0x0ULL | (0ULL << 18) | (1ULL << 21) | (2ULL << 24) | (0ULL << 27) | (0ULL << 30) | (2ULL << 33) |
                                           (1ULL << 36) | (2ULL << 39) | (4ULL << 42) | (0ULL << 45);
        const int L = min(tid, 15);
        const int ib = (int)((kIB >> (3 * L)) & 7), ic = (int)((kIC >> (3 * L)) & 7);
        const double prod = tot[ib] * tot[ic];
        const double shift = (tid == 0) ? cT : (tid == 1) ? cU : (tid == 2) ? cV : (tid == 3) ? cW : (tid == 4) ? cP : 0.0;
        double o = (tot[L] + shift) - ((tid >= 6) ? prod : 0.0);
        if (tid >= 16) {                                  // third-order forms: three pairs of lanes
            const double da = tot[0], db = tot[1], dc = tot[2], dd = tot[3];
            const bool w = (tid & 1) != 0;                // the omega twin of each pair
            const double sTT = tot[6] - da * da, sUU = tot[7] - db * db, sVV = tot[8] - dc * dc;
            if (tid < 18) {                               // [v T'T'], [w T'T']
                const double x0 = w ? tot[17] : tot[16], x1 = w ? tot[10] : tot[9], x2 = w ? dd : dc, sh = w ? cW : cV;
                o = x0 - 2 * da * x1 + da * da * x2 + sh * sTT;
            } else if (tid < 20) {                        // [K v], [K w]
                const double mU = cU + db, mV = cV + dc, m = w ? cW + dd : mV;
                const double s1 = w ? tot[12] - dd * db : tot[11] - db * dc, s2 = w ? tot[13] - dd * dc : sVV;
                o = 2 * mU * s1 + mU * mU * m + 2 * mV * s2 + mV * mV * m;
            } else {                                      // [E v], [E w]
                const double x0 = w ? tot[19] : tot[18], x1 = w ? tot[12] : tot[11], x2 = w ? dd : dc, x3 = w ? tot[13] : tot[8], sh = w ? cW : cV;
                o = x0 - 2 * db * x1 + db * db * x2 - 2 * dc * x3 + dc * dc * x2 + sh * (sUU + sVV);
            }
        }
        if (store) out[tid] = o;
    }
    if (store && tid < 4) {
        double o = 0.0;
        if (XCOV) {
            const double da = tot[0];
            const double x = tot[kNA + (tid & 2)], y = tot[kNA + (tid & 2) + 1];     // forward pair (tid 0, 1), backward pair (tid 2, 3)
            o = (tid & 1) ? ((tid & 2) ? r.cTb : r.cTf) + y : x - da * y;           // [T] of the neighbour row : centred covariance
        }
        out[LEC_S_SPARE + tid] = o;
    }
}

}  // namespace lec
#endif
