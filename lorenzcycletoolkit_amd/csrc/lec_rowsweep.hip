// lec_rowsweep.hip -- stage 1, single-sweep variant: one workgroup per row, ONE sweep over the row.
//
// A 256-thread workgroup owns a chunk of consecutive box latitudes at one (time, level) and walks
// them south to north.  Differences from the per-row kernel (lec_rowstats.hip):
//
//  * T window in registers.  The diabatic-heating residual needs T at j-1, j, j+1; the workgroup
//    keeps those three rows in registers and loads only T(j+2) per step, so each T row is fetched
//    once per chunk instead of three times (measured: the j+-1 re-reads were the most expensive
//    neighbour loads, 17 % of the kernel).
//  * One sweep per row.  Sums are formed about a shift c (the row's first element) instead of about
//    the row mean: with a = T - cT, b = u - cU, c = v - cV, d = w - cW, e = Phi - cP, f = Q the 20
//    weighted sums  <a> <b> <c> <d> <e> <f>  <aa> <bb> <cc> <ca> <da> <bc> <db> <dc> <de> <fa>
//    <caa> <daa> <(bb+cc)c> <(bb+cc)d>  give every centred statistic exactly, e.g.
//    [T'T'] = <aa> - <a>^2,  [vT'T'] = <caa> - 2<a><ca> + <a>^2<c> + cV [T'T'],
//    [Kv] = 2[u][u'v'] + [u]^2[v] + 2[v][v'v'] + [v]^3   (K = u^2+v^2-u'^2-v'^2 = 2u[u]-[u]^2+2v[v]-[v]^2).
//    The shift keeps the cancellation benign (|row mean - first element| is of the order of the eddy
//    amplitude).  Half the fp64 work of the two-sweep form, one block reduction instead of two, and
//    the fields need not stay in registers across it.
//  * Weights and lane geometry are hoisted out of the row loop.
//
// Output: the same LEC_NSTAT row records as lec_rowstats.hip (stage 2 is unchanged).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"
#include "lec_rowcommon.h"

using namespace lec;

namespace {

constexpr int kThreads = 64; // one wave per row: measured best (64 / 128 / 256 threads: 19.0 / 19.1 / 20.1 ms per 64 steps)
constexpr int kNA = 20;     // shifted sums per row
constexpr int kHalf = 10;   // statistics per reduction round (2 rounds x 10)

template <typename TIN, int VEC, int MODE>
constexpr int sweep_min_waves() {
#if LEC_MINW > 0
    return LEC_MINW;
#else
    return (sizeof(TIN) == 4 && MODE != 0 && VEC == 4) ? LEC_MINW_SINGLE - 1 : LEC_MINW_SINGLE;
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

// the 20 shifted sums of one element (see the header comment)
__device__ __forceinline__ void accum20(double (&acc)[kNA], double w, double a, double b, double c, double d, double ee, double f) {
    const double wa = w * a, wb = w * b, wc = w * c, wd = w * d;
    const double waa = wa * a, wbb = wb * b, wcc = wc * c;
    acc[0] += wa; acc[1] += wb; acc[2] += wc; acc[3] += wd;
    acc[4] += w * ee; acc[5] += w * f;
    acc[6] += waa; acc[7] += wbb; acc[8] += wcc;
    acc[9] += wc * a;      // <ca>
    acc[10] += wd * a;     // <da>
    acc[11] += wb * c;     // <bc>
    acc[12] += wd * b;     // <db>
    acc[13] += wd * c;     // <dc>
    acc[14] += wd * ee;    // <de>
    acc[15] += wa * f;     // <fa>
    acc[16] += waa * c;    // <caa>
    acc[17] += waa * d;    // <daa>
    const double wee = wbb + wcc;      // only the sums <bbc>+<ccc> and <bbd>+<ccd> are ever needed ([Ev], [Ew])
    acc[18] += wee * c;    // <(bb+cc) c>
    acc[19] += wee * d;    // <(bb+cc) d>
}

// block sums of the accumulators (two rounds through the same LDS tile), then the centred
// statistics from the shifted sums (lanes 0..21) written to the row record.  Ends with every read of
// `red` / `tot` complete only after the caller's next barrier.
template <int NTHR, int NR = kHalf>
__device__ __forceinline__ void finish_row(const double (&acc)[kNA], double* red, double* tot, int tid, double inv_xlen,
                                           double cT, double cU, double cV, double cW, double cP, double* __restrict__ out) {
    constexpr int rshift = red_rshift(NTHR);
#pragma unroll
    for (int r0 = 0; r0 < kNA; r0 += NR) {      // rounds of NR statistics through the same LDS tile
        double h[NR];
#pragma unroll
        for (int s = 0; s < NR; ++s) h[s] = (r0 + s < kNA) ? acc[(r0 + s < kNA) ? r0 + s : 0] : 0.0;
        const double t0 = block_sums<NR, NTHR>(h, red, tid);
        if ((tid & ((1 << rshift) - 1)) == 0 && (tid >> rshift) < NR && r0 + (tid >> rshift) < kNA)
            tot[r0 + (tid >> rshift)] = t0 * inv_xlen;
        __syncthreads();
    }
    if (tid < 22) {
        const double da = tot[0], db = tot[1], dc = tot[2], dd = tot[3], de = tot[4], df = tot[5];
        const double mT = cT + da, mU = cU + db, mV = cV + dc, mW = cW + dd;
        const double sTT = tot[6] - da * da, sUU = tot[7] - db * db, sVV = tot[8] - dc * dc;
        const double sUV = tot[11] - db * dc, sWU = tot[12] - dd * db, sWV = tot[13] - dd * dc;
        double o;
        switch (tid) {
            case 0: o = mT; break;
            case 1: o = mU; break;
            case 2: o = mV; break;
            case 3: o = mW; break;
            case 4: o = cP + de; break;
            case 5: o = df; break;
            case 6: o = sTT; break;
            case 7: o = sUU; break;
            case 8: o = sVV; break;
            case 9: o = tot[9] - dc * da; break;                       // [v'T']
            case 10: o = tot[10] - dd * da; break;                     // [w'T']
            case 11: o = sUV; break;
            case 12: o = sWU; break;
            case 13: o = sWV; break;
            case 14: o = tot[14] - dd * de; break;                     // [w'Phi']
            case 15: o = tot[15] - df * da; break;                     // [Q'T']
            case 16: o = tot[16] - 2 * da * tot[9] + da * da * dc + cV * sTT; break;     // [v T'T']
            case 17: o = tot[17] - 2 * da * tot[10] + da * da * dd + cW * sTT; break;    // [w T'T']
            case 18: o = 2 * mU * sUV + mU * mU * mV + 2 * mV * sVV + mV * mV * mV; break;   // [K v]
            case 19: o = 2 * mU * sWU + mU * mU * mW + 2 * mV * sWV + mV * mV * mW; break;   // [K w]
            case 20: o = tot[18] - 2 * db * tot[11] + db * db * dc - 2 * dc * tot[8] + dc * dc * dc
                         + cV * (sUU + sVV); break;                                           // [E v]
            default: o = tot[19] - 2 * db * tot[12] + db * db * dd - 2 * dc * tot[13] + dc * dc * dd
                         + cW * (sUU + sVV); break;                                           // [E w]
        }
        out[tid] = o;
    }
    if (tid < 4) out[LEC_S_SPARE + tid] = 0.0;
}

// one workgroup per (time, level, box-latitude) row, ONE sweep over the row (see the header comment)
template <typename TIN, int VEC, bool UNIFORM, int MODE, bool ONE_TRIP>
__global__ void __launch_bounds__(kThreads, (sweep_min_waves<TIN, VEC, MODE>())) lec_rowsweep_kernel(const RowParams p) {
    constexpr int NTHR = kThreads;
    constexpr bool WITH_Q = MODE != 0;
    constexpr int nthr = NTHR;
    __shared__ double red[kHalf * red_stride(NTHR)];
    __shared__ double tot[24];

    const int tid = threadIdx.x;
    int jb, k, tl;
    if (p.order == 0) {
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
    const double phimul = p.P ? 1.0 : 0.0;
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
            const double* tcf = p.tcoef + (size_t)t * 3;
            ta = tcf[0]; tb = tcf[1]; tc = tcf[2];
        }
    }

    // shifts: the row's first box element (wave-uniform scalar loads)
    const double cT = (double)rT[0], cU = (double)rU[0], cV = (double)rV[0], cW = (double)rW[0];
    const double cP = has_p ? (double)rP[0] * phimul : 0.0;

    double acc[kNA];
#pragma unroll
    for (int s = 0; s < kNA; ++s) acc[s] = 0.0;
    double ewT = 0, ewU = 0, ewV = 0, eeT = 0, eeU = 0, eeV = 0;
    bool has_w = false, has_e = false;

    // a real loop (not unrolled): one vector of every row per trip keeps the live state at the 20
    // accumulators plus one vector's worth of operands, which is what lets 4 waves/SIMD fit
    const int ntrips = ONE_TRIP ? 1 : p.ntrips;      // short rows (moving boxes): one trip, no loop
#pragma unroll 1
    for (int it = 0; it < ntrips; ++it) {
        const int e0 = (it * nthr + tid) * VEC - shift;
        const bool lane_in = (e0 <= e0_last);
        const unsigned eo = (unsigned)(min(e0, e0_last) + shift);
        double fT[VEC], fU[VEC], fV[VEC], fW[VEC], fP[VEC];
        double tjm[VEC], tjp[VEC], tkm[VEC], tkp[VEC], tm[VEC], tp[VEC];
        double tl_edge = 0.0, tr_edge = 0.0;
        load_vec<TIN, VEC, MODE == 0>(rT - shift, eo, fT);
        load_vec<TIN, VEC, true>(rU - shift, eo, fU);
        load_vec<TIN, VEC, true>(rV - shift, eo, fV);
        load_vec<TIN, VEC, true>(rW - shift, eo, fW);
        if (has_p) {
            load_vec<TIN, VEC, true>(rP - shift, eo, fP);
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) fP[q] = 0.0;
        }
        if (WITH_Q) {
            load_vec<TIN, VEC, false>(rTjm - shift, eo, tjm);
            load_vec<TIN, VEC, false>(rTjp - shift, eo, tjp);
            load_vec<TIN, VEC, false>(rTkm - shift, eo, tkm);
            load_vec<TIN, VEC, false>(rTkp - shift, eo, tkp);
            if (p.order == 7) {      // tiled order: the T[t+-1] rows are own rows of sibling workgroups -> keep them cacheable
                load_vec<TIN, VEC, false>(rTtm - shift, eo, tm);
                if (MODE == 1) load_vec<TIN, VEC, false>(rTtp - shift, eo, tp);
            } else {
                load_vec<TIN, VEC, true>(rTtm - shift, eo, tm);
                if (MODE == 1) load_vec<TIN, VEC, true>(rTtp - shift, eo, tp);
            }
            // in-row neighbours T[i-1], T[i+1]: from the adjacent lanes' registers; only the first / last lane
            // of a wave reads memory (consecutive lanes hold consecutive vectors)
            // (fp64 storage only: with fp32 storage the extra live values tip the kernel into scratch)
            if (sizeof(TIN) == 8) {
                const int lane = tid & 63;
                if (lane == 0) tl_edge = (double)rT[min(max(e0 - 1, 0), nxb - 1)];
                if (lane == 63) tr_edge = (double)rT[min(max(e0 + VEC, 0), nxb - 1)];
                tl_edge = from_prev_lane(fT[VEC - 1], tl_edge);
                tr_edge = from_next_lane(fT[0], tr_edge);
            } else {
                tl_edge = (double)rT[min(max(e0 - 1, 0), nxb - 1)];
                tr_edge = (double)rT[min(max(e0 + VEC, 0), nxb - 1)];
            }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int e = e0 + q;
            const bool inside = (e >= 0) && (e < nxb) && lane_in;
            const bool first = inside && (e == 0), last = inside && (e == nxb - 1);
            double w;
            if (UNIFORM) w = inside ? ((first || last) ? 0.5 * h_rad : h_rad) : 0.0;
            else w = inside ? wl[min(max(e, 0), nxb - 1)] : 0.0;
            const double Tv = inside ? fT[q] : cT;
            const double Uv = inside ? fU[q] : cU;
            const double Vv = inside ? fV[q] : cV;
            const double Wv = inside ? fW[q] : cW;
            const double Pv = inside ? fP[q] * phimul : cP;
            double f = 0.0;
            if (WITH_Q) {
                const double Tl = (q == 0) ? tl_edge : fT[q > 0 ? q - 1 : 0];
                const double Tr = (q == VEC - 1) ? tr_edge : fT[q < VEC - 1 ? q + 1 : q];
                double dTl;
                if (UNIFORM) {
                    dTl = first ? (Tr - Tv) * inv_hdeg : (last ? (Tv - Tl) * inv_hdeg : (Tr - Tl) * (0.5 * inv_hdeg));
                } else {
                    const int ec = min(max(e, 0), nxb - 1);
                    dTl = gl[3 * ec + 0] * Tl + gl[3 * ec + 1] * Tv + gl[3 * ec + 2] * Tr;
                }
                const double dTdt = (MODE == 1) ? (ta * tm[q] + tb * Tv + tc * tp[q]) : tm[q];
                const double dTphi = ga * tjm[q] + gb * Tv + gc * tjp[q];
                const double S = al * tkm[q] + be * Tv + gm * tkp[q];
                f = kCp * (dTdt + Uv * dTl * inv_dx + Vv * dTphi - Wv * S);
                f = inside ? f : 0.0;
            }
            accum20(acc, w, Tv - cT, Uv - cU, Vv - cV, Wv - cW, Pv - cP, f);
            ewT = first ? Tv : ewT; ewU = first ? Uv : ewU; ewV = first ? Vv : ewV; has_w = has_w || first;
            eeT = last ? Tv : eeT;  eeU = last ? Uv : eeU;  eeV = last ? Vv : eeV;  has_e = has_e || last;
        }
    }

    finish_row<NTHR, kHalf>(acc, red, tot, tid, inv_xlen, cT, cU, cV, cW, cP, out);
    if (has_w) { out[LEC_S_TW] = ewT; out[LEC_S_UW] = ewU; out[LEC_S_VW] = ewV; }
    if (has_e) { out[LEC_S_TE] = eeT; out[LEC_S_UE] = eeU; out[LEC_S_VE] = eeV; }
}

template <typename TIN, int VEC>
int launch_vec(RowParams& p, bool uniform, int mode, int nblocks, hipStream_t st) {
    // vectors needed to cover the longest row, plus one for the alignment shift; one wave walks them in trips of 64
    const int nvec = (p.nxb_max + VEC - 1) / VEC + (VEC > 1 ? 1 : 0);
    p.ntrips = (nvec + kThreads - 1) / kThreads;
    dim3 grid(nblocks), block(kThreads);
#define LEC_LAUNCH(U, M) do { if (p.ntrips == 1) hipLaunchKernelGGL((lec_rowsweep_kernel<TIN, VEC, U, M, true>), grid, block, 0, st, p); \
                              else hipLaunchKernelGGL((lec_rowsweep_kernel<TIN, VEC, U, M, false>), grid, block, 0, st, p); } while (0)
    if (uniform) {
        if (mode == 0) LEC_LAUNCH(true, 0); else if (mode == 1) LEC_LAUNCH(true, 1); else LEC_LAUNCH(true, 2);
    } else {
        if (mode == 0) LEC_LAUNCH(false, 0); else if (mode == 1) LEC_LAUNCH(false, 1); else LEC_LAUNCH(false, 2);
    }
#undef LEC_LAUNCH
    return LEC_OK;
}

}  // namespace

// `aligned` = every cube base is 16-byte aligned and nx is a multiple of the 16-byte vector;
// `aligned8` (fp32 only) = 8-byte aligned bases and even nx.  fp32 storage uses float2 vectors: the
// same two elements per lane and trip as fp64, which is what keeps the kernel at 4 waves/SIMD
// (float4 needs 163 VGPRs and measured 10 % slower).
int lec_launch_rowsweep(lec::RowParams& p, int dtype, bool aligned, bool aligned8, bool uniform, int mode, int nblocks, hipStream_t st) {
    if (dtype == LEC_F64) return aligned ? launch_vec<double, 2>(p, uniform, mode, nblocks, st) : launch_vec<double, 1>(p, uniform, mode, nblocks, st);
    const char* ev = getenv("LEC_F32VEC");
    if (aligned && ev && atoi(ev) == 4) return launch_vec<float, 4>(p, uniform, mode, nblocks, st);
    return aligned8 ? launch_vec<float, 2>(p, uniform, mode, nblocks, st) : launch_vec<float, 1>(p, uniform, mode, nblocks, st);
}
