// lec_rowsweep.hip -- stage 1, second generation: latitude-sweeping workgroups, one sweep per row.
//
// A 256-thread workgroup owns a chunk of consecutive box latitudes at one (time, level) and walks
// them south to north.  Differences from the per-row kernel (lec_rowstats.hip):
//
//  * T window in registers.  The diabatic-heating residual needs T at j-1, j, j+1; the workgroup
//    keeps those three rows in registers and loads only T(j+2) per step, so each T row is fetched
//    once per chunk instead of three times (measured: the j+-1 re-reads were the most expensive
//    neighbour loads, 17 % of the kernel).
//  * One sweep per row.  Sums are formed about a shift c (the row's first element) instead of about
//    the row mean: with a = T - cT, b = u - cU, c = v - cV, d = w - cW, e = Phi - cP, f = Q the 22
//    weighted sums  <a> <b> <c> <d> <e> <f>  <aa> <bb> <cc> <ca> <da> <bc> <db> <dc> <de> <fa>
//    <caa> <daa> <bbc> <ccc> <bbd> <ccd>  give every centred statistic exactly, e.g.
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

constexpr int kHalf = 11;   // statistics per reduction round (2 rounds x 11 = 22)

template <typename TIN, int ITERS, int MODE>
constexpr int sweep_min_waves() {
#if LEC_MINW > 0
    return LEC_MINW;
#else
    return (ITERS > 3) ? 1 : 2;
#endif
}

template <typename TIN, int VEC, int ITERS, bool UNIFORM, int MODE>
__global__ void __launch_bounds__(256, (sweep_min_waves<TIN, ITERS, MODE>())) lec_rowsweep_kernel(const RowParams p) {
    constexpr bool WITH_Q = MODE != 0;
    __shared__ double red[kHalf * kRedStride];
    __shared__ double tot[24];

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int rshift = (nthr == 256) ? 3 : (nthr == 128 ? 2 : 1);

    // ---- workgroup -> (time, level, latitude chunk); blockIdx % 8 labels the XCD (speed only) ----
    const int xcd = blockIdx.x & 7;
    int q = blockIdx.x >> 3;
    const int per_t = p.cpx * p.nl;
    const int tl = q / per_t; q -= tl * per_t;
    const int k = q / p.cpx;
    const int chunk = xcd * p.cpx + (q - k * p.cpx);
    const int jb0 = chunk * p.jrows;
    if (jb0 >= p.nyb_max) return;

    const int bi = (p.n_box == 1) ? 0 : tl;
    const int iw = p.box[4 * bi + 0], ie = p.box[4 * bi + 1], js = p.box[4 * bi + 2], jn = p.box[4 * bi + 3];
    const int nxb = ie - iw + 1, nyb = jn - js + 1;
    const int t = p.t_begin + tl;
    const size_t plane = (size_t)p.ny * p.nx;
    const size_t cube = plane * p.nl;
    const size_t lev0 = (size_t)t * cube + (size_t)k * plane + iw;       // element offset of (t, k, lat 0, iw)
    const int shift = (VEC > 1) ? (int)((lev0 + (size_t)js * p.nx) % VEC) : 0;   // equal for every row (nx % VEC == 0)
    const int e0_last = ((nxb - 1 + shift) / VEC) * VEC - shift;

    const double inv_xlen = p.boxtab[4 * bi + 0];
    const double h_rad = p.boxtab[4 * bi + 1];
    const double inv_hdeg = p.boxtab[4 * bi + 2];
    const double* __restrict__ wl = UNIFORM ? nullptr : p.wlon + (size_t)bi * p.nxb_max;
    const double* __restrict__ gl = UNIFORM ? nullptr : p.glon + (size_t)bi * p.nxb_max * 3;
    const double phimul = p.P ? 1.0 : 0.0;

    double al = 0, be = 0, gm = 0, ta = 0, tb = 0, tc = 0;
    ptrdiff_t okm = 0, okp = 0, otm = 0, otp = 0;     // element offsets of the k-1 / k+1 / t-1 / t+1 rows
    if (WITH_Q) {
        const double* lv = p.levtab + (size_t)k * 3;
        al = lv[0]; be = lv[1]; gm = lv[2];
        if (k > 0) okm = -(ptrdiff_t)plane;
        if (k < p.nl - 1) okp = (ptrdiff_t)plane;
        if (MODE == 1) {
            if (t > 0) otm = -(ptrdiff_t)cube;
            if (t < p.nt - 1) otp = (ptrdiff_t)cube;
            const double* tcf = p.tcoef + (size_t)t * 3;
            ta = tcf[0]; tb = tcf[1]; tc = tcf[2];
        }
    }

    // ---- lane geometry and weights: identical for every row of the chunk ----
    unsigned eo[ITERS];      // lane offset (elements) from the row's 16-byte boundary
    int e0c[ITERS];
    double wg[UNIFORM ? 1 : ITERS][VEC];     // tabulated weights only for non-uniform longitudes
    bool ins[ITERS][VEC];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int e0 = (it * nthr + tid) * VEC - shift;
        e0c[it] = min(e0, e0_last);
        eo[it] = (unsigned)(e0c[it] + shift);
#pragma unroll
        for (int qq = 0; qq < VEC; ++qq) {
            const int e = e0 + qq;
            const bool inside = (e >= 0) && (e < nxb) && (e0 == e0c[it]);
            ins[it][qq] = inside;
            if (!UNIFORM) wg[it][qq] = inside ? wl[min(max(e, 0), nxb - 1)] : 0.0;
        }
    }

    // ---- T window: rows j-1, j, j+1 of the chunk's first latitude ----
    const int jend = min(jb0 + p.jrows, nyb);
    double Tm[ITERS][VEC], T0[ITERS][VEC], Tp[ITERS][VEC];
    if (jb0 < nyb) {
        const TIN* r0 = (const TIN*)p.T + lev0 + (size_t)(js + jb0) * p.nx;
        const TIN* rm = (jb0 > 0) ? r0 - p.nx : r0;
        const TIN* rp = (jb0 < nyb - 1) ? r0 + p.nx : r0;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            load_vec<TIN, VEC, MODE == 0>((r0) - shift, eo[it], T0[it]);
            if (WITH_Q) {
                load_vec<TIN, VEC, false>((rm) - shift, eo[it], Tm[it]);
                load_vec<TIN, VEC, false>((rp) - shift, eo[it], Tp[it]);
            }
        }
    }

    for (int jb = jb0; jb < jend; ++jb) {
        const size_t rowoff = lev0 + (size_t)(js + jb) * p.nx;
        const TIN* __restrict__ rT = (const TIN*)p.T + rowoff;
        const TIN* __restrict__ rU = (const TIN*)p.U + rowoff;
        const TIN* __restrict__ rV = (const TIN*)p.V + rowoff;
        const TIN* __restrict__ rW = (const TIN*)p.W + rowoff;
        const TIN* __restrict__ rP = (const TIN*)(p.P ? p.P : p.T) + rowoff;
        double* __restrict__ out = p.rows + ((size_t)(tl * p.nl + k) * p.nyb_max + jb) * LEC_NSTAT;

        // ---- own rows of this latitude (and T two rows north for the window): all in flight at once ----
        double fU[ITERS][VEC], fV[ITERS][VEC], fW[ITERS][VEC], fP[ITERS][VEC];
        const TIN* rn = (jb + 2 <= nyb - 1) ? rT + 2 * (size_t)p.nx : rT;      // T(j+2), clamped
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            load_vec<TIN, VEC, true>((rU) - shift, eo[it], fU[it]);
            load_vec<TIN, VEC, true>((rV) - shift, eo[it], fV[it]);
            load_vec<TIN, VEC, true>((rW) - shift, eo[it], fW[it]);
            if (MODE != 0 || p.P) {
                load_vec<TIN, VEC, true>((rP) - shift, eo[it], fP[it]);
            } else {
#pragma unroll
                for (int qq = 0; qq < VEC; ++qq) fP[it][qq] = 0.0;
            }
        }
        // shifts: the row's first box element (wave-uniform scalar loads)
        const double cT = (double)rT[0], cU = (double)rU[0], cV = (double)rV[0], cW = (double)rW[0];
        const double cP = (MODE != 0 || p.P) ? (double)rP[0] * phimul : 0.0;

        double ga = 0, gb = 0, gc = 0, inv_dx = 0;
        if (WITH_Q) {
            const double* lt = p.lattab + ((size_t)bi * p.nyb_max + jb) * 4;
            ga = lt[0]; gb = lt[1]; gc = lt[2]; inv_dx = lt[3];
        }

        // ---- the sweep ----
        double acc[22];
#pragma unroll
        for (int s = 0; s < 22; ++s) acc[s] = 0.0;
        double ewT = 0, ewU = 0, ewV = 0, eeT = 0, eeU = 0, eeV = 0;
        bool has_w = false, has_e = false;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int e0 = (it * nthr + tid) * VEC - shift;
            double tkm[VEC], tkp[VEC], ttm[VEC], ttp[VEC];
            double tle = 0.0, tre = 0.0;
            if (WITH_Q) {       // level / time neighbours of this vector (the compiler hoists them above the previous vector's math)
                load_vec<TIN, VEC, false>((rT + okm) - shift, eo[it], tkm);
                load_vec<TIN, VEC, false>((rT + okp) - shift, eo[it], tkp);
                if (MODE == 1) {
                    load_vec<TIN, VEC, true>((rT + otm) - shift, eo[it], ttm);
                    load_vec<TIN, VEC, true>((rT + otp) - shift, eo[it], ttp);
                } else {
                    load_vec<TIN, VEC, true>(((const TIN*)p.DT + rowoff) - shift, eo[it], ttm);
                }
                tle = (double)rT[min(max(e0 - 1, 0), nxb - 1)];
                tre = (double)rT[min(max(e0 + VEC, 0), nxb - 1)];
            }
#pragma unroll
            for (int qq = 0; qq < VEC; ++qq) {
                const bool inside = ins[it][qq];
                const int e = e0 + qq;
                const bool first = inside && (e == 0), last = inside && (e == nxb - 1);
                const double w = UNIFORM ? (inside ? ((first || last) ? 0.5 * h_rad : h_rad) : 0.0) : wg[UNIFORM ? 0 : it][qq];
                const double Tv = inside ? T0[it][qq] : cT;
                const double Uv = inside ? fU[it][qq] : cU;
                const double Vv = inside ? fV[it][qq] : cV;
                const double Wv = inside ? fW[it][qq] : cW;
                const double Pv = inside ? fP[it][qq] * phimul : cP;
                double f = 0.0;
                if (WITH_Q) {
                    const double Tl = (qq == 0) ? tle : T0[it][qq > 0 ? qq - 1 : 0];
                    const double Tr = (qq == VEC - 1) ? tre : T0[it][qq < VEC - 1 ? qq + 1 : qq];
                    double dTl;
                    if (UNIFORM) {
                        dTl = first ? (Tr - Tv) * inv_hdeg : (last ? (Tv - Tl) * inv_hdeg : (Tr - Tl) * (0.5 * inv_hdeg));
                    } else {
                        const int ec = min(max(e, 0), nxb - 1);
                        dTl = gl[3 * ec + 0] * Tl + gl[3 * ec + 1] * Tv + gl[3 * ec + 2] * Tr;
                    }
                    const double dTdt = (MODE == 1) ? (ta * ttm[qq] + tb * Tv + tc * ttp[qq]) : ttm[qq];
                    const double dTphi = ga * Tm[it][qq] + gb * Tv + gc * Tp[it][qq];
                    const double S = al * tkm[qq] + be * Tv + gm * tkp[qq];
                    f = kCp * (dTdt + Uv * dTl * inv_dx + Vv * dTphi - Wv * S);
                    f = inside ? f : 0.0;
                }
                const double a = Tv - cT, b = Uv - cU, c = Vv - cV, d = Wv - cW, ee = Pv - cP;
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
                acc[18] += wbb * c;    // <bbc>
                acc[19] += wcc * c;    // <ccc>
                acc[20] += wbb * d;    // <bbd>
                acc[21] += wcc * d;    // <ccd>
                ewT = first ? Tv : ewT; ewU = first ? Uv : ewU; ewV = first ? Vv : ewV; has_w = has_w || first;
                eeT = last ? Tv : eeT;  eeU = last ? Uv : eeU;  eeV = last ? Vv : eeV;  has_e = has_e || last;
            }
        }

        // T two rows north for the window: in flight behind the reductions
        double Tn[ITERS][VEC];
        if (WITH_Q) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) load_vec<TIN, VEC, false>((rn) - shift, eo[it], Tn[it]);
        }

        // ---- block sums, two rounds of 11 through the same LDS tile ----
        {
            double h[kHalf];
#pragma unroll
            for (int s = 0; s < kHalf; ++s) h[s] = acc[s];
            const double t0 = block_sums<kHalf>(h, red, tid, nthr);
            if ((tid & ((1 << rshift) - 1)) == 0 && (tid >> rshift) < kHalf) tot[tid >> rshift] = t0 * inv_xlen;
            __syncthreads();
#pragma unroll
            for (int s = 0; s < kHalf; ++s) h[s] = acc[kHalf + s];
            const double t1 = block_sums<kHalf>(h, red, tid, nthr);
            if ((tid & ((1 << rshift) - 1)) == 0 && (tid >> rshift) < kHalf) tot[kHalf + (tid >> rshift)] = t1 * inv_xlen;
            __syncthreads();
        }

        // ---- centred statistics from the shifted sums (lanes 0..21), edge columns, spare ----
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
                case 20: o = (tot[18] - 2 * db * tot[11] + db * db * dc) + (tot[19] - 2 * dc * tot[8] + dc * dc * dc)
                             + cV * (sUU + sVV); break;                                           // [E v]
                default: o = (tot[20] - 2 * db * tot[12] + db * db * dd) + (tot[21] - 2 * dc * tot[13] + dc * dc * dd)
                             + cW * (sUU + sVV); break;                                           // [E w]
            }
            out[tid] = o;
        }
        if (has_w) { out[LEC_S_TW] = ewT; out[LEC_S_UW] = ewU; out[LEC_S_VW] = ewV; }
        if (has_e) { out[LEC_S_TE] = eeT; out[LEC_S_UE] = eeU; out[LEC_S_VE] = eeV; }
        if (tid < 4) out[LEC_S_SPARE + tid] = 0.0;

        // ---- slide the T window north ----
        if (WITH_Q) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
#pragma unroll
                for (int qq = 0; qq < VEC; ++qq) { Tm[it][qq] = T0[it][qq]; T0[it][qq] = Tp[it][qq]; Tp[it][qq] = Tn[it][qq]; }
            }
        } else if (jb + 1 < jend) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) load_vec<TIN, VEC, true>((rT + p.nx) - shift, eo[it], T0[it]);
        }
        __syncthreads();   // `tot` / `red` are rewritten by the next row
    }

    // rows of a box smaller than nyb_max inside this chunk: zero records
    for (int jb = max(jend, jb0); jb < min(jb0 + p.jrows, p.nyb_max); ++jb) {
        double* out = p.rows + ((size_t)(tl * p.nl + k) * p.nyb_max + jb) * LEC_NSTAT;
        if (tid < LEC_NSTAT) out[tid] = 0.0;
    }
}

template <typename TIN, int VEC, int ITERS>
void launch_cfg(const RowParams& p, bool uniform, int mode, int nthr, int nblocks, hipStream_t st) {
    dim3 grid(nblocks), block(nthr);
#define LEC_LAUNCH(U, M) hipLaunchKernelGGL((lec_rowsweep_kernel<TIN, VEC, ITERS, U, M>), grid, block, 0, st, p)
    if (uniform) {
        if (mode == 0) LEC_LAUNCH(true, 0); else if (mode == 1) LEC_LAUNCH(true, 1); else LEC_LAUNCH(true, 2);
    } else {
        if (mode == 0) LEC_LAUNCH(false, 0); else if (mode == 1) LEC_LAUNCH(false, 1); else LEC_LAUNCH(false, 2);
    }
#undef LEC_LAUNCH
}

template <typename TIN, int VEC>
int launch_vec(const RowParams& p, bool uniform, int mode, int nblocks, hipStream_t st) {
    const int nvec = (p.nxb_max + VEC - 1) / VEC + (VEC > 1 ? 1 : 0);
    int nthr = 256;
    if (nvec <= 64) nthr = 64;
    else if (nvec <= 128) nthr = 128;
    const int iters = (nvec + nthr - 1) / nthr;
    if (iters <= 1) launch_cfg<TIN, VEC, 1>(p, uniform, mode, nthr, nblocks, st);
    else if (iters <= 2) launch_cfg<TIN, VEC, 2>(p, uniform, mode, nthr, nblocks, st);
    else if (iters <= 3) launch_cfg<TIN, VEC, 3>(p, uniform, mode, nthr, nblocks, st);
    else if (iters <= 4) launch_cfg<TIN, VEC, 4>(p, uniform, mode, nthr, nblocks, st);
    else if (iters <= 6) launch_cfg<TIN, VEC, 6>(p, uniform, mode, nthr, nblocks, st);
    else if (iters <= LEC_MAX_ITERS) launch_cfg<TIN, VEC, LEC_MAX_ITERS>(p, uniform, mode, nthr, nblocks, st);
    else return LEC_ERR_UNSUPPORTED;
    return LEC_OK;
}

}  // namespace

// Called by lec_rowstats() (lec_rowstats.hip) once the arguments are validated.
int lec_launch_rowsweep(lec::RowParams& p, int dtype, bool aligned, bool uniform, int mode, hipStream_t st) {
    const char* ej = getenv("LEC_JROWS");
    p.jrows = ej ? atoi(ej) : 8;
    if (p.jrows < 1) p.jrows = 1;
    const int nchunks = (p.nyb_max + p.jrows - 1) / p.jrows;
    p.cpx = (nchunks + 7) / 8;
    const long long nblocks = (long long)p.t_count * p.nl * p.cpx * 8;
    if (nblocks > 0x7fffffffLL) return LEC_ERR_UNSUPPORTED;
    if (dtype == LEC_F64) return aligned ? launch_vec<double, 2>(p, uniform, mode, (int)nblocks, st) : launch_vec<double, 1>(p, uniform, mode, (int)nblocks, st);
    return aligned ? launch_vec<float, 4>(p, uniform, mode, (int)nblocks, st) : launch_vec<float, 1>(p, uniform, mode, (int)nblocks, st);
}
