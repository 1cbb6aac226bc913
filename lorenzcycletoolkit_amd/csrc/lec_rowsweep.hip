// lec_rowsweep.hip -- stage 1, single-sweep kernel: one wave per (time, level, box-latitude) row, ONE sweep over the row.
//
//  * One sweep per row.  Sums are formed about a shift c (the row's first element) instead of about
//    the row mean: with a = T - cT, b = u - cU, c = v - cV, d = w - cW, e = Phi - cP, f = Q the 20
//    weighted sums  <a> <b> <c> <d> <e> <f>  <aa> <bb> <cc> <ca> <da> <bc> <db> <dc> <de> <fa>
//    <caa> <daa> <(bb+cc)c> <(bb+cc)d>  give every centred statistic exactly, e.g.
//    [T'T'] = <aa> - <a>^2,  [vT'T'] = <caa> - 2<a><ca> + <a>^2<c> + cV [T'T'],
//    [Kv] = 2[u][u'v'] + [u]^2[v] + 2[v][v'v'] + [v]^3   (K = u^2+v^2-u'^2-v'^2 = 2u[u]-[u]^2+2v[v]-[v]^2).
//    The shift keeps the cancellation benign (|row mean - first element| is of the order of the eddy
//    amplitude).  Half the fp64 work of the two-sweep form (lec_rowstats.hip), one reduction instead of
//    two, and the fields need not stay in registers across it.
//  * One wave per row walks it in trips of 64 vectors inside a real (not unrolled) loop: the live state
//    is the 20 accumulators plus one vector of every operand, which fits 4 waves per SIMD.
//  * The helpers shared with the row-block kernel (lec_rowblock.hip) live in lec_sweep.h.
//
// Output: the same LEC_NSTAT row records as lec_rowstats.hip (stage 2 is unchanged).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/lec_hip.h"
#include "lec_internal.h"
#include "lec_rowcommon.h"
#include "lec_sweep.h"

using namespace lec;

namespace {

constexpr int kThreads = 64; // one wave per row: measured best (64 / 128 / 256 threads: 19.0 / 19.1 / 20.1 ms per 64 steps)
// one wave per (time, level, box-latitude) row, ONE sweep over the row (see the header comment)
// MODE: 0 no Q; 1 dT/dt from the cube's time neighbours per point; 2 dT/dt cube; 3 as 1 on one fixed box, through
// cross-time covariances (sweep_elems).  BOTH (MODE 3 only): the row also forms the covariance with T(t-1) -- the first
// processed time step of a launch
template <typename TIN, int VEC, bool UNIFORM, int MODE, bool ONE_TRIP, bool BOTH>
__global__ void __launch_bounds__(kThreads, (sweep_min_waves<TIN, VEC, MODE>())) lec_rowsweep_kernel(const RowParams p) {
    constexpr int NTHR = kThreads;
    constexpr bool WITH_Q = MODE != 0;
    constexpr bool TIME_NB = (MODE == 1 || MODE == 3);      // reads T at t+1 (and t-1)
    constexpr int nthr = NTHR;
    __shared__ double red[kRound * red_stride(NTHR)];
    __shared__ double tot[24];

    const int tid = threadIdx.x;
    int jb, k, tl;
    if (ONE_TRIP) {
        // short rows are instruction-bound (a 61-point row is ~1000 instructions, a third of them index arithmetic): a 3-D
        // grid hands out (latitude, level, time step) without a single integer division
        jb = (int)(blockIdx.x & 7) * p.jchunk + (int)(blockIdx.x >> 3);      // blockIdx.x % 8 labels the XCD: contiguous latitude chunks
        k = blockIdx.y; tl = blockIdx.z;
        if (jb >= p.nyb_max || (int)(blockIdx.x >> 3) >= p.jchunk) return;
    } else if (p.order == 0) {
        int r = blockIdx.x;
        jb = r % p.nyb_max; r /= p.nyb_max;
        k = r % p.nl;
        tl = r / p.nl;
    } else {
        // XCD label (speed only): every XCD owns a contiguous latitude chunk.  Order 2 walks it latitude-fastest
        // per (time, level).  Order 7 (all terms, fixed box) walks tiles of tgroup time steps x jgroup latitudes
        // at one level, levels next: the ~500 one-wave workgroups resident on an XCD then cover a compact (t, k, j)
        // neighbourhood, so T rows at t+-1 as well as j+-1 / k+-1 are rows a sibling is fetching right now
        // (measured: fabric traffic 1.39 -> 1.29 x algorithmic, -8 % time).
        const int xcd = blockIdx.x & 7;
        int q = blockIdx.x >> 3;
        if (p.order == 7) {
            // tiles of tgroup time steps x jgroup latitudes at one level run together on the XCD, levels next
            const int tile = p.tgroup * p.jgroup;
            int tid_ = q / tile;
            const int within = q - tid_ * tile;
            const int t_in = within % p.tgroup, j_in = within / p.tgroup;
            k = tid_ % p.nl; tid_ /= p.nl;
            const int tgc = (p.t_count + p.tgroup - 1) / p.tgroup;
            const int tg = tid_ % tgc, jg = tid_ / tgc;
            tl = tg * p.tgroup + t_in;
            const int jl = jg * p.jgroup + j_in;
            jb = xcd * p.jchunk + jl;
            if (tl >= p.t_count || jl >= p.jchunk) return;
        } else {
            const int per_t = p.jchunk * p.nl;
            tl = q / per_t; q -= tl * per_t;
            k = q / p.jchunk;
            jb = xcd * p.jchunk + (q - k * p.jchunk);
        }
        if (jb >= p.nyb_max) return;
    }
    const int bi = (p.n_box == 1) ? 0 : tl;
    const int iw = p.box[4 * bi + 0], ie = p.box[4 * bi + 1], js = p.box[4 * bi + 2], jn = p.box[4 * bi + 3];
    const int nxb = ie - iw + 1, nyb = jn - js + 1;
    double* __restrict__ out = p.rows + ((size_t)(tl * p.nl + k) * p.nyb_max + jb) * LEC_NSTAT;
    if (jb >= nyb) {  // padding rows of a box smaller than nyb_max
        if (tid < LEC_NSTAT) out[tid] = 0.0;
        return;
    }
    const int j = js + jb, t = p.t_begin + tl;
    const size_t plane = (size_t)p.ny * p.nx;
    const size_t cube = plane * p.nl;
    const size_t rowoff = (size_t)t * cube + (size_t)k * plane + (size_t)j * p.nx + iw;
    const int shift = (VEC > 1) ? (int)(rowoff % VEC) : 0;
    const int e0_last = ((nxb - 1 + shift) / VEC) * VEC - shift;

    const TIN* __restrict__ rT = (const TIN*)p.T + rowoff;
    const TIN* __restrict__ rU = (const TIN*)p.U + rowoff;
    const TIN* __restrict__ rV = (const TIN*)p.V + rowoff;
    const TIN* __restrict__ rW = (const TIN*)p.W + rowoff;
    const TIN* __restrict__ rP = (const TIN*)(p.P ? p.P : p.T) + rowoff;
    const bool has_p = (MODE != 0) || (p.P != nullptr);

    const double inv_xlen = p.boxtab[4 * bi + 0];
    const double h_rad = p.boxtab[4 * bi + 1];
    const double inv_hdeg = p.boxtab[4 * bi + 2];
    const double* __restrict__ wl = UNIFORM ? nullptr : p.wlon + (size_t)bi * p.nxb_max;
    const double* __restrict__ gl = UNIFORM ? nullptr : p.glon + (size_t)bi * p.nxb_max * 3;

    const TIN *rTjm = rT, *rTjp = rT, *rTkm = rT, *rTkp = rT, *rTtm = rT, *rTtp = rT;
    double ga = 0, gb = 0, gc = 0, inv_dx = 0, al = 0, be = 0, gm = 0, ta = 0, tb = 0, tc = 0;
    if (WITH_Q) {
        if (jb > 0) rTjm = rT - p.nx;
        if (jb < nyb - 1) rTjp = rT + p.nx;
        if (k > 0) rTkm = rT - plane;
        if (k < p.nl - 1) rTkp = rT + plane;
        const double* lt = p.lattab + ((size_t)bi * p.nyb_max + jb) * 4;
        ga = lt[0]; gb = lt[1]; gc = lt[2]; inv_dx = lt[3];
        const double* lv = p.levtab + (size_t)k * 3;
        al = lv[0]; be = lv[1]; gm = lv[2];
        if (MODE == 2) {
            rTtm = (const TIN*)p.DT + rowoff;
        } else {
            if (t > 0) rTtm = rT - cube;
            if (t < p.nt - 1) rTtp = rT + cube;
            if (MODE == 1) {
                const double* tcf = p.tcoef + (size_t)t * 3;
                ta = tcf[0]; tb = tcf[1]; tc = tcf[2];
            }
        }
    }

    // shifts: the row's first box element (wave-uniform scalar loads)
    SweepRow r;
    r.nxb = nxb;
    r.cT = (double)rT[0]; r.cU = (double)rU[0]; r.cV = (double)rV[0]; r.cW = (double)rW[0];
    r.cP = (has_p && p.P) ? (double)rP[0] : 0.0;
    r.cx = 0.5 * inv_hdeg * inv_dx; r.inv_dx = inv_dx; r.wl = wl; r.gl = gl;
    r.cTf = (MODE == 3) ? (double)rTtp[0] : 0.0;
    r.cTb = (MODE == 3 && BOTH) ? (double)rTtm[0] : 0.0;
    // T, u, v at the east box column (boundary terms), fetched now so that the row does not end on a load
    const double eT = (double)rT[nxb - 1], eU = (double)rU[nxb - 1], eV = (double)rV[nxb - 1];

    double acc[kNA], xacc[kNX];
#pragma unroll
    for (int s = 0; s < kNA; ++s) acc[s] = 0.0;
#pragma unroll
    for (int s = 0; s < kNX; ++s) xacc[s] = 0.0;

    QCoef qc;
    qc.tb_ = ta; qc.tf_ = tc; qc.tm = tb; qc.k0 = al; qc.k1 = gm; qc.km = be; qc.j0 = ga; qc.j1 = gc; qc.jm = gb;

    // one trip = one vector of every row operand per lane; EDGE trips hold a row end or lanes past it.
    // Operands stay in their storage type (TIN) and are converted where they are used.
    auto trip = [&](auto edge_tag, const int it) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int el = it * nthr * VEC - shift;              // box element of lane 0 (wave-uniform)
        const int e0 = el + tid * VEC;
        const bool lane_in = !EDGE || (e0 <= e0_last);
        const unsigned eo = (unsigned)((EDGE ? min(e0, e0_last) : e0) + shift);
        TIN fT[VEC], fU[VEC], fV[VEC], fW[VEC], fP[VEC];
        QRaw<TIN, VEC> qr;
        double tl_edge = 0.0, tr_edge = 0.0;
        load_vec<TIN, VEC, MODE == 0>(rT - shift, eo, fT);
        load_vec<TIN, VEC, true>(rU - shift, eo, fU);
        load_vec<TIN, VEC, true>(rV - shift, eo, fV);
        load_vec<TIN, VEC, true>(rW - shift, eo, fW);
        if (has_p) load_vec<TIN, VEC, true>(rP - shift, eo, fP);
        if (!has_p || !p.P) {                                // no geopotential cube: its statistics are written as 0
#pragma unroll
            for (int q = 0; q < VEC; ++q) fP[q] = (TIN)0;
        }
        if (WITH_Q) {
            load_vec<TIN, VEC, false>(rTjm - shift, eo, qr.j0);
            load_vec<TIN, VEC, false>(rTjp - shift, eo, qr.j1);
            load_vec<TIN, VEC, false>(rTkm - shift, eo, qr.k0);
            load_vec<TIN, VEC, false>(rTkp - shift, eo, qr.k1);
            // MODE 3: T(t+1) for the cross-time covariance (T(t-1) too when BOTH); MODE 1: both; MODE 2: the dT/dt cube (rTtm points into it).
            // Tiled order: the T(t+1) row is the own row of a sibling workgroup -> keep it cacheable
            if (TIME_NB) {
                if (p.order == 7) load_vec<TIN, VEC, false>(rTtp - shift, eo, qr.tf);
                else load_vec<TIN, VEC, true>(rTtp - shift, eo, qr.tf);
                if (MODE == 1 || BOTH) load_vec<TIN, VEC, true>(rTtm - shift, eo, qr.tb);
            } else {
                load_vec<TIN, VEC, true>(rTtm - shift, eo, qr.tf);
            }
            // in-row neighbours T[i-1], T[i+1]: from the adjacent lanes' registers (DPP); the elements beyond the
            // wave's two end lanes are at wave-uniform addresses: scalar loads
            const int il = EDGE ? min(max(el - 1, 0), nxb - 1) : el - 1;
            const int ir = EDGE ? min(max(el + nthr * VEC, 0), nxb - 1) : el + nthr * VEC;
            tl_edge = from_prev_lane((double)fT[VEC - 1], (double)rT[il]);
            tr_edge = from_next_lane((double)fT[0], (double)rT[ir]);
        }
#if defined(LEC_EXPERIMENT_NOEDGE) && LEC_EXPERIMENT_NOEDGE      // measurement builds only: edge trips with the plain arithmetic (WRONG row ends)
        if (lane_in) sweep_elems<VEC, UNIFORM, false, MODE, BOTH>(acc, xacc, r, e0, lane_in, fT, fU, fV, fW, fP, tl_edge, tr_edge, qr, qc);
#else
        sweep_elems<VEC, UNIFORM, EDGE, MODE, BOTH>(acc, xacc, r, e0, lane_in, fT, fU, fV, fW, fP, tl_edge, tr_edge, qr, qc);
#endif
    };

#if defined(LEC_EXPERIMENT_PREFETCH) && LEC_EXPERIMENT_PREFETCH     // measurement builds only (plain arithmetic everywhere: WRONG row ends)
    struct TripData { TIN fT[VEC], fU[VEC], fV[VEC], fW[VEC], fP[VEC]; QRaw<TIN, VEC> qr; TIN sl, sr; };
    auto issue = [&](TripData& d, const int it) {
        const int el = it * nthr * VEC - shift;
        const int e0 = el + tid * VEC;
        const unsigned eo = (unsigned)(min(e0, e0_last) + shift);
        load_vec<TIN, VEC, MODE == 0>(rT - shift, eo, d.fT);
        load_vec<TIN, VEC, true>(rU - shift, eo, d.fU);
        load_vec<TIN, VEC, true>(rV - shift, eo, d.fV);
        load_vec<TIN, VEC, true>(rW - shift, eo, d.fW);
        load_vec<TIN, VEC, true>(rP - shift, eo, d.fP);
        if (WITH_Q) {
#if LEC_EXPERIMENT_PREFETCH == 1
            load_vec<TIN, VEC, false>(rTjm - shift, eo, d.qr.j0);
            load_vec<TIN, VEC, false>(rTjp - shift, eo, d.qr.j1);
            load_vec<TIN, VEC, false>(rTkm - shift, eo, d.qr.k0);
            load_vec<TIN, VEC, false>(rTkp - shift, eo, d.qr.k1);
#endif
            load_vec<TIN, VEC, false>(rTtp - shift, eo, d.qr.tf);
            d.sl = rT[min(max(el - 1, 0), nxb - 1)];
            d.sr = rT[min(max(el + nthr * VEC, 0), nxb - 1)];
        }
    };
    auto compute = [&](TripData& d, const int it) {
        const int e0 = it * nthr * VEC - shift + tid * VEC;
#if LEC_EXPERIMENT_PREFETCH == 2      // only the streamed operands run ahead; the four stencil rows (L2 hits) are fetched when the trip is computed
        if (WITH_Q) {
            const unsigned eo = (unsigned)(min(e0, e0_last) + shift);
            load_vec<TIN, VEC, false>(rTjm - shift, eo, d.qr.j0);
            load_vec<TIN, VEC, false>(rTjp - shift, eo, d.qr.j1);
            load_vec<TIN, VEC, false>(rTkm - shift, eo, d.qr.k0);
            load_vec<TIN, VEC, false>(rTkp - shift, eo, d.qr.k1);
        }
#endif
        const double tl_edge = from_prev_lane((double)d.fT[VEC - 1], (double)d.sl);
        const double tr_edge = from_next_lane((double)d.fT[0], (double)d.sr);
        if (e0 <= e0_last) sweep_elems<VEC, UNIFORM, false, MODE, BOTH>(acc, xacc, r, e0, true, d.fT, d.fU, d.fV, d.fW, d.fP, tl_edge, tr_edge, d.qr, qc);
    };
    const int ntrips = p.ntrips;
    {
        // the number of loads in flight must be the same on every path into a block, or the compiler's s_waitcnt has to assume the
        // fewest (a conditional issue() made every wait a vmcnt(0) and the pipeline serial): issue() is unconditional inside the
        // loop, the last trips are peeled
        TripData A, B;
        issue(A, 0);
        issue(B, min(1, ntrips - 1));
        int it = 0;
#pragma unroll 1
        for (; it + 3 < ntrips; it += 2) {
            compute(A, it);
            issue(A, it + 2);
            compute(B, it + 1);
            issue(B, it + 3);
        }
        compute(A, it);
        if (it + 2 < ntrips) {
            issue(A, it + 2);
            compute(B, it + 1);
            compute(A, it + 2);
        } else if (it + 1 < ntrips) {
            compute(B, it + 1);
        }
    }
    (void)trip;
#else
    // a real loop (not unrolled): the live state stays at the 20 accumulators plus one vector's worth of
    // operands, which is what lets 4 waves/SIMD fit.  Trips [1, mid_end) lie strictly inside the row.
    const int ntrips = ONE_TRIP ? 1 : p.ntrips;      // short rows (moving boxes): one trip, no loop
    trip(std::true_type{}, 0);
    if (!ONE_TRIP) {
        const int mid_end = min((nxb - 1 + shift) / (nthr * VEC), ntrips);
#pragma unroll 1
        for (int it = 1; it < mid_end; ++it) trip(std::false_type{}, it);
#pragma unroll 1
        for (int it = max(mid_end, 1); it < ntrips; ++it) trip(std::true_type{}, it);
    }
#endif

    finish_row<NTHR, kRound, MODE == 3>(acc, xacc, red, tot, tid, UNIFORM ? h_rad * inv_xlen : inv_xlen, r, out);
    // T, u, v at the west / east box columns (boundary terms): wave-uniform scalar loads
    if (tid == 0) {
        out[LEC_S_TW] = r.cT; out[LEC_S_UW] = r.cU; out[LEC_S_VW] = r.cV;
        out[LEC_S_TE] = eT; out[LEC_S_UE] = eU; out[LEC_S_VE] = eV;
    }
}

// Completes [Q] and [Q'T'] of the time-stencil mode from the row records: the time-derivative part of Q is linear in
// T(t-1), T(t), T(t+1), so  [Q] += cp (ta [T](t-1) + tb [T](t) + tc [T](t+1))  and
// [Q'T'] += cp (ta [T'(t)T'(t-1)] + tb [T'T'] + tc [T'(t)T'(t+1)]).  The backward pieces are the previous row's
// forward ones when that row was processed in this launch, else the row's own (slots 30, 31).  One fixed box only.
__global__ void __launch_bounds__(256) lec_qtime_kernel(const RowParams p) {
#pragma clang fp contract(off)
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_t = (long long)p.nl * p.nyb_max;
    if (row >= per_t * p.t_count) return;
    const int tl = (int)(row / per_t);
    double* __restrict__ rec = p.rows + (size_t)row * LEC_NSTAT;
    const bool from_prev = tl > 0;
    const double* __restrict__ prev = rec - (size_t)per_t * LEC_NSTAT;
    const double mtb = from_prev ? prev[LEC_S_MT] : rec[LEC_S_SPARE + 3];
    const double cb = from_prev ? prev[LEC_S_SPARE + 0] : rec[LEC_S_SPARE + 2];
    const double* tcf = p.tcoef + (size_t)(p.t_begin + tl) * 3;
    const double ta = tcf[0], tb = tcf[1], tc = tcf[2];
    const double dm = (ta * mtb + tc * rec[LEC_S_SPARE + 1]) + tb * rec[LEC_S_MT];
    const double dc = (ta * cb + tc * rec[LEC_S_SPARE + 0]) + tb * rec[LEC_S_TT];
    rec[LEC_S_MQ] = rec[LEC_S_MQ] + kCp * dm;
    rec[LEC_S_QT] = rec[LEC_S_QT] + kCp * dc;
}

// workgroups (= rows, rounded up to whole XCD chunks / tiles) of a launch over p.t_count time steps; 0 = too many
long long grid_blocks(const RowParams& p) {
    long long n;
    if (p.order == 0) n = (long long)p.t_count * p.nl * p.nyb_max;
    else if (p.order == 7) {
        const long long tgc = (p.t_count + p.tgroup - 1) / p.tgroup, jgc = (p.jchunk + p.jgroup - 1) / p.jgroup;
        n = 8LL * jgc * tgc * p.nl * p.tgroup * p.jgroup;
    } else n = (long long)p.t_count * 8 * p.jchunk * p.nl;
    return n > 0x7fffffffLL ? 0 : n;
}

template <typename TIN, int VEC, bool BOTH>
int launch_one(RowParams p, bool uniform, int mode, hipStream_t st) {
    // vectors needed to cover the longest row, plus one for the alignment shift; one wave walks them in trips of 64
    const int nvec = (p.nxb_max + VEC - 1) / VEC + (VEC > 1 ? 1 : 0);
    p.ntrips = (nvec + kThreads - 1) / kThreads;
    if (p.order == 7 && p.t_count < 2) p.order = 2;
    long long nblocks = grid_blocks(p);
    if (nblocks == 0) { p.order = 0; nblocks = grid_blocks(p); }
    if (nblocks == 0) return LEC_ERR_UNSUPPORTED;
    dim3 grid((unsigned)nblocks), block(kThreads);
    if (p.jchunk < 1) p.jchunk = (p.nyb_max + 7) / 8;
    dim3 grid3(8u * (unsigned)p.jchunk, (unsigned)p.nl, (unsigned)p.t_count);    // one-trip rows: (XCD x latitude, level, time step)
    if (p.ntrips == 1 && (p.nl > 65535 || p.t_count > 65535)) return LEC_ERR_UNSUPPORTED;
#define LEC_LAUNCH(U, M) do { if (p.ntrips == 1) hipLaunchKernelGGL((lec_rowsweep_kernel<TIN, VEC, U, M, true, BOTH && M == 3>), grid3, block, 0, st, p); \
                              else hipLaunchKernelGGL((lec_rowsweep_kernel<TIN, VEC, U, M, false, BOTH && M == 3>), grid, block, 0, st, p); } while (0)
#define LEC_MODES(U) do { if (mode == 0) LEC_LAUNCH(U, 0); else if (mode == 1) LEC_LAUNCH(U, 1); else if (mode == 2) LEC_LAUNCH(U, 2); else LEC_LAUNCH(U, 3); } while (0)
    if (uniform) LEC_MODES(true); else LEC_MODES(false);
#undef LEC_MODES
#undef LEC_LAUNCH
    return LEC_OK;
}

// mode 3 (time stencil on one fixed box): the first time step of the launch forms both cross-time covariances, the
// others only the forward one
template <typename TIN, int VEC>
int launch_vec(const RowParams& p, bool uniform, int mode, hipStream_t st) {
    if (mode != 3) return launch_one<TIN, VEC, false>(p, uniform, mode, st);
    RowParams p0 = p;
    p0.t_count = 1;
    int rc = launch_one<TIN, VEC, true>(p0, uniform, mode, st);
    if (rc != LEC_OK || p.t_count < 2) return rc;
    return launch_one<TIN, VEC, false>(later_steps(p), uniform, mode, st);
}

}  // namespace

// `aligned` = every cube base is 16-byte aligned and nx is a multiple of the 16-byte vector;
// `aligned8` (fp32 only) = 8-byte aligned bases and even nx.  fp32 storage uses float4 vectors when it can
// (four elements per lane and trip, operands kept as floats and converted at use, one element finished before the
// next starts: 141 VGPRs, 3 waves/SIMD, 10.5 ms per 64 steps) and float2 otherwise (11.0 ms; tuning.f32_vec = 2 forces it).
int lec_launch_rowsweep(const lec::RowParams& p, int dtype, bool aligned, bool aligned8, bool uniform, int mode, int f32_vec, hipStream_t st) {
    if (dtype == LEC_F64) return aligned ? launch_vec<double, 2>(p, uniform, mode, st) : launch_vec<double, 1>(p, uniform, mode, st);
    if (aligned && f32_vec != 2) return launch_vec<float, 4>(p, uniform, mode, st);
    return aligned8 ? launch_vec<float, 2>(p, uniform, mode, st) : launch_vec<float, 1>(p, uniform, mode, st);
}

int lec_launch_qtime(const lec::RowParams& p, hipStream_t st) {
    const long long rows = (long long)p.t_count * p.nl * p.nyb_max;
    hipLaunchKernelGGL(lec_qtime_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, p);
    return LEC_OK;
}
