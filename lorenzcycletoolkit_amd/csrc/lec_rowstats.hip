// lec_rowstats.hip -- stage 1 of the MI355X Lorenz-Energy-Cycle engine (gfx950, wave64).
//
// One workgroup per (time, level, box-latitude) longitude row.  The row of every field is held in
// registers (16-byte coalesced loads, ITERS vectors per lane); sweep 1 forms the trapezoid-weighted
// zonal means (and the diabatic-heating residual Q point by point), sweep 2 the eddy covariances and
// mixed moments about those means -- the same "deviation from the zonal mean, then average" order as
// the reference (box_data.py:157-231, calc_averages.py:25-43), in fp64.  Block-wide sums go through an
// LDS transpose (conflict-free ds_write_b64 / ds_read_b64) and a fixed-order tree, so results are
// bitwise reproducible run to run.  HBM-bound: no MFMA (nothing here is a contraction).
//
// Reference formulas: SURVEY.md appendix A / F; AdiabaticHEating thermodynamics.py:95-121.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"
#include "lec_rowcommon.h"

namespace {

using namespace lec;

// ---------------------------------------------------------------------------------------------
// the row kernel
//   TIN      storage type of the cubes
//   VEC      elements per 16-byte load (1 = unaligned fallback)
//   ITERS    vectors per lane
//   UNIFORM  uniformly spaced longitudes (weights / d-dlon from two scalars instead of tables)
//   MODE     0: T,u,v,omega only (no Phi, no Q)   1: all terms, dT/dt from the cube (tcoef)
//            2: all terms, dT/dt supplied as a cube
// ---------------------------------------------------------------------------------------------
// waves per SIMD asked of the register allocator, from the round-1 A/B on MI355X (profiles/r01_notes.md):
// the 4-field configuration gains from 4 waves (70 % vs 66 % of HBM peak), fp32 storage from 3,
// the fp64 all-terms kernel is fastest unconstrained (170 VGPRs, 2 waves; forcing 128 spills).
template <typename TIN, int NTHR, int ITERS, int MODE>
constexpr int lec_min_waves() {
#if LEC_MINW > 0
    return LEC_MINW;
#else
    return (ITERS > 3) ? 1 : (sizeof(TIN) == 4 ? 3 : (MODE == 0 ? 4 : 1));
#endif
}

template <typename TIN, int VEC, int NTHR, int ITERS, bool UNIFORM, int MODE>
__global__ void __launch_bounds__(NTHR, (lec_min_waves<TIN, NTHR, ITERS, MODE>())) lec_rowstats_kernel(const RowParams p) {
    constexpr bool WITH_Q = MODE != 0;
    __shared__ double red[16 * red_stride(NTHR)];
    __shared__ double bc[8];

    constexpr int nthr = NTHR;
    const int tid = threadIdx.x;
    int jb, k, tl;
    if (p.order == 0) {
        int r = blockIdx.x;
        jb = r % p.nyb_max; r /= p.nyb_max;
        k = r % p.nl;
        tl = r / p.nl;
    } else {
        // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 = XCD label, speed only):
        // give each XCD a contiguous latitude chunk so that the T rows at j+-1 / k+-1 needed by the
        // diabatic-heating stencil are rows sibling workgroups on the SAME XCD stream at about the
        // same time (L2 hits instead of fabric reads).
        const int xcd = blockIdx.x & 7;
        int q = blockIdx.x >> 3;
        const int per_t = p.jchunk * p.nl;
        tl = q / per_t; q -= tl * per_t;
        k = q / p.jchunk;                    // latitude fastest inside the XCD's chunk
        const int jl = q - k * p.jchunk;
        jb = xcd * p.jchunk + jl;
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
    // first element of the last vector that still touches the row: clamp target for idle lanes
    const int e0_last = ((nxb - 1 + shift) / VEC) * VEC - shift;

    const TIN* __restrict__ rT = (const TIN*)p.T + rowoff;
    const TIN* __restrict__ rU = (const TIN*)p.U + rowoff;
    const TIN* __restrict__ rV = (const TIN*)p.V + rowoff;
    const TIN* __restrict__ rW = (const TIN*)p.W + rowoff;
    // no geopotential cube: read T instead and multiply by 0 (keeps the loads branch-free)
    const TIN* __restrict__ rP = (const TIN*)(p.P ? p.P : p.T) + rowoff;
    const double phimul = p.P ? 1.0 : 0.0;

    const double inv_xlen = p.boxtab[4 * bi + 0];
    const double h_rad = p.boxtab[4 * bi + 1];
    const double inv_hdeg = p.boxtab[4 * bi + 2];
    const double* __restrict__ wl = UNIFORM ? nullptr : p.wlon + (size_t)bi * p.nxb_max;
    const double* __restrict__ gl = UNIFORM ? nullptr : p.glon + (size_t)bi * p.nxb_max * 3;

    // Q neighbours (clamped to an existing row; the matching coefficient is 0 there)
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
            rTtm = (const TIN*)p.DT + rowoff;      // the dT/dt cube rides in the "t-1" slot
            ta = 1.0; tb = 0.0; tc = 0.0;
        } else {
            if (t > 0) rTtm = rT - cube;
            if (t < p.nt - 1) rTtp = rT + cube;
            const double* tcf = p.tcoef + (size_t)t * 3;
            ta = tcf[0]; tb = tcf[1]; tc = tcf[2];
        }
    }

    double fT[ITERS][VEC], fU[ITERS][VEC], fV[ITERS][VEC], fW[ITERS][VEC], fP[ITERS][VEC], fQ[ITERS][VEC];
    double wg[ITERS][VEC];
    double a1[6] = {0, 0, 0, 0, 0, 0};
    double ewT = 0, ewU = 0, ewV = 0, eeT = 0, eeU = 0, eeV = 0;
    bool has_w = false, has_e = false;

    // ---------------- own rows: every load of the row in flight before the first use ----------------
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const unsigned eo = (unsigned)(min((it * nthr + tid) * VEC - shift, e0_last) + shift);
        load_vec<TIN, VEC, MODE == 0>(rT - shift, eo, fT[it]);       // T is re-read by neighbour rows unless MODE 0
        load_vec<TIN, VEC, true>(rU - shift, eo, fU[it]);
        load_vec<TIN, VEC, true>(rV - shift, eo, fV[it]);
        load_vec<TIN, VEC, true>(rW - shift, eo, fW[it]);
        if (MODE != 0 || p.P) {                // wave-uniform
            load_vec<TIN, VEC, true>(rP - shift, eo, fP[it]);
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) fP[it][q] = 0.0;
        }
    }

    // ---------------- sweep 1: Q, weighted sums ----------------
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int e0 = (it * nthr + tid) * VEC - shift;
        const int e0c = min(e0, e0_last);
        const unsigned eo = (unsigned)(e0c + shift);
        double tjm[VEC], tjp[VEC], tkm[VEC], tkp[VEC], tm[VEC], tp[VEC];
        double tl_edge = 0.0, tr_edge = 0.0;
        if (WITH_Q) {
            load_vec<TIN, VEC, false>(rTjm - shift, eo, tjm);
            load_vec<TIN, VEC, false>(rTjp - shift, eo, tjp);
            load_vec<TIN, VEC, false>(rTkm - shift, eo, tkm);
            load_vec<TIN, VEC, false>(rTkp - shift, eo, tkp);
            load_vec<TIN, VEC, true>(rTtm - shift, eo, tm);
            if (MODE == 1) load_vec<TIN, VEC, true>(rTtp - shift, eo, tp);
            const int le = min(max(e0 - 1, 0), nxb - 1);
            const int re = min(max(e0 + VEC, 0), nxb - 1);
            tl_edge = (double)rT[le];
            tr_edge = (double)rT[re];
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int e = e0 + q;
            const bool inside = (e >= 0) && (e < nxb) && (e0 == e0c);
            const bool first = inside && (e == 0), last = inside && (e == nxb - 1);
            double w;
            if (UNIFORM) {
                w = inside ? ((first || last) ? 0.5 * h_rad : h_rad) : 0.0;
            } else {
                w = inside ? wl[min(max(e, 0), nxb - 1)] : 0.0;
            }
            wg[it][q] = w;
            const double T0 = inside ? fT[it][q] : 0.0;
            const double U0 = inside ? fU[it][q] : 0.0;
            const double V0 = inside ? fV[it][q] : 0.0;
            const double W0 = inside ? fW[it][q] : 0.0;
            const double P0 = inside ? fP[it][q] * phimul : 0.0;
            double Q = 0.0;
            if (WITH_Q) {
                const double Tl = (q == 0) ? tl_edge : fT[it][q > 0 ? q - 1 : 0];
                const double Tr = (q == VEC - 1) ? tr_edge : fT[it][q < VEC - 1 ? q + 1 : q];
                double dTl;
                if (UNIFORM) {
                    dTl = first ? (Tr - T0) * inv_hdeg : (last ? (T0 - Tl) * inv_hdeg : (Tr - Tl) * (0.5 * inv_hdeg));
                } else {
                    const int ec = min(max(e, 0), nxb - 1);
                    dTl = gl[3 * ec + 0] * Tl + gl[3 * ec + 1] * T0 + gl[3 * ec + 2] * Tr;
                }
                const double dTdt = (MODE == 1) ? (ta * tm[q] + tb * T0 + tc * tp[q]) : tm[q];
                const double dTphi = ga * tjm[q] + gb * T0 + gc * tjp[q];
                const double S = al * tkm[q] + be * T0 + gm * tkp[q];
                Q = kCp * (dTdt + U0 * dTl * inv_dx + V0 * dTphi - W0 * S);
                Q = inside ? Q : 0.0;
            }
            fT[it][q] = T0; fU[it][q] = U0; fV[it][q] = V0; fW[it][q] = W0; fP[it][q] = P0; fQ[it][q] = Q;
            a1[0] += w * T0;
            a1[1] += w * U0;
            a1[2] += w * V0;
            a1[3] += w * W0;
            a1[4] += w * P0;
            a1[5] += w * Q;
            // west / east box columns (boundary_terms.py:138-140 etc.): captured with selects, stored once
            // after the sweeps so that no store sits between the loads of later iterations
            ewT = first ? T0 : ewT; ewU = first ? U0 : ewU; ewV = first ? V0 : ewV; has_w = has_w || first;
            eeT = last ? T0 : eeT;  eeU = last ? U0 : eeU;  eeV = last ? V0 : eeV;  has_e = has_e || last;
        }
    }

    {
        const double tot = block_sums<6, NTHR>(a1, red, tid);
        constexpr int rshift = red_rshift(NTHR);
        if ((tid & ((1 << rshift) - 1)) == 0 && (tid >> rshift) < 6) {
            const double m = tot * inv_xlen;
            bc[tid >> rshift] = m;
            out[LEC_S_MT + (tid >> rshift)] = m;
        }
        __syncthreads();
    }
    const double mT = bc[0], mU = bc[1], mV = bc[2], mW = bc[3], mP = bc[4], mQ = bc[5];

    // ---------------- sweep 2: eddy covariances and mixed moments about the means ----------------
    double a2[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) a2[s] = 0.0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const double w = wg[it][q];
            const double u = fU[it][q], v = fV[it][q], om = fW[it][q];
            const double Te = fT[it][q] - mT, ue = u - mU, ve = v - mV, we = om - mW;
            const double Pe = fP[it][q] - mP, Qe = fQ[it][q] - mQ;
            const double wTe = w * Te;
            const double TT = Te * Te;
            const double E = ue * ue + ve * ve;            // u'^2 + v'^2           (boundary_terms.py:287)
            const double K = u * u + v * v - E;            // u^2+v^2-u'^2-v'^2     (boundary_terms.py:237)
            a2[0] += wTe * Te;           // [T'T']
            a2[1] += w * ue * ue;        // [u'u']
            a2[2] += w * ve * ve;        // [v'v']
            a2[3] += wTe * ve;           // [v'T']
            a2[4] += wTe * we;           // [w'T']
            a2[5] += w * ue * ve;        // [u'v']
            a2[6] += w * we * ue;        // [w'u']
            a2[7] += w * we * ve;        // [w'v']
            a2[8] += w * we * Pe;        // [w'Phi']
            a2[9] += wTe * Qe;           // [Q'T']
            a2[10] += w * v * TT;        // [v T'T']
            a2[11] += w * om * TT;       // [w T'T']
            a2[12] += w * K * v;         // [K v]
            a2[13] += w * K * om;        // [K w]
            a2[14] += w * E * v;         // [E v]
            a2[15] += w * E * om;        // [E w]
        }
    }
    {
        const double tot = block_sums<16, NTHR>(a2, red, tid);
        constexpr int rshift = red_rshift(NTHR);
        if ((tid & ((1 << rshift) - 1)) == 0 && (tid >> rshift) < 16) out[LEC_S_TT + (tid >> rshift)] = tot * inv_xlen;
    }
    if (has_w) { out[LEC_S_TW] = ewT; out[LEC_S_UW] = ewU; out[LEC_S_VW] = ewV; }
    if (has_e) { out[LEC_S_TE] = eeT; out[LEC_S_UE] = eeU; out[LEC_S_VE] = eeV; }
    if (tid < 4) out[LEC_S_SPARE + tid] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------
template <typename TIN, int VEC, int NTHR, int ITERS>
void launch_cfg(const RowParams& p, bool uniform, int mode, int nblocks, hipStream_t st) {
    dim3 grid(nblocks), block(NTHR);
#define LEC_LAUNCH(U, M) hipLaunchKernelGGL((lec_rowstats_kernel<TIN, VEC, NTHR, ITERS, U, M>), grid, block, 0, st, p)
    if (uniform) {
        if (mode == 0) LEC_LAUNCH(true, 0); else if (mode == 1) LEC_LAUNCH(true, 1); else LEC_LAUNCH(true, 2);
    } else {
        if (mode == 0) LEC_LAUNCH(false, 0); else if (mode == 1) LEC_LAUNCH(false, 1); else LEC_LAUNCH(false, 2);
    }
#undef LEC_LAUNCH
}

template <typename TIN, int VEC>
int launch_vec(const RowParams& p, bool uniform, int mode, int nblocks, hipStream_t st) {
    // vectors needed to cover the longest row, plus one for the alignment shift
    const int nvec = (p.nxb_max + VEC - 1) / VEC + (VEC > 1 ? 1 : 0);
    if (nvec <= 64) launch_cfg<TIN, VEC, 64, 1>(p, uniform, mode, nblocks, st);
    else if (nvec <= 128) launch_cfg<TIN, VEC, 128, 1>(p, uniform, mode, nblocks, st);
    else if (nvec <= 256) launch_cfg<TIN, VEC, 256, 1>(p, uniform, mode, nblocks, st);
    else if (nvec <= 512) launch_cfg<TIN, VEC, 256, 2>(p, uniform, mode, nblocks, st);
    else if (nvec <= 768) launch_cfg<TIN, VEC, 256, 3>(p, uniform, mode, nblocks, st);
    else if (nvec <= 1024) launch_cfg<TIN, VEC, 256, 4>(p, uniform, mode, nblocks, st);
    else if (nvec <= 1536) launch_cfg<TIN, VEC, 256, 6>(p, uniform, mode, nblocks, st);
    else if (nvec <= 256 * LEC_MAX_ITERS) launch_cfg<TIN, VEC, 256, LEC_MAX_ITERS>(p, uniform, mode, nblocks, st);
    else return LEC_ERR_UNSUPPORTED;
    return LEC_OK;
}

}  // namespace

int lec_launch_rowsweep(const lec::RowParams& p, int dtype, bool aligned, bool aligned8, bool uniform, int mode, int f32_vec, hipStream_t st);
int lec_launch_rowblock(lec::RowParams p, int dtype, bool aligned, bool aligned8, bool uniform, int bt, int bk, int bj, hipStream_t st);
int lec_launch_boxtile(const lec::RowParams& p, int dtype, bool uniform, int mode, int tg, hipStream_t st);
int lec_launch_qtime(const lec::RowParams& p, hipStream_t st);
bool lec_boxplane_serves(const lec::RowParams& p, int dtype, bool uniform, int mode);
int lec_launch_boxplane(lec::RowParams p, int dtype, int mode, hipStream_t st);

extern "C" int lec_max_row(int dtype, int aligned, int kernel) {
    (void)dtype;
    // the single-sweep kernels walk a row in trips / strips, so any row the cube can hold is fine; the two-sweep
    // cross-check kernel keeps a whole row in registers: 256 * LEC_MAX_ITERS vectors
    if (kernel == LEC_KERNEL_TWO_SWEEP) return (256 * LEC_MAX_ITERS - 1) * (aligned ? 2 : 1);
    return 1 << 24;
}

extern "C" int lec_rowstats(const lec_rowstats_args* a) {
    if (!a) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: null args");
    if (!a->tair_d || !a->u_d || !a->v_d || !a->omega_d || !a->rows_d || !a->box_d || !a->boxtab_d)
        return lec_set_error(LEC_ERR_ARG, "lec_rowstats: null field / output / box pointer");
    if (a->dtype != LEC_F64 && a->dtype != LEC_F32) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: dtype must be LEC_F64 or LEC_F32");
    if (a->nt < 1 || a->nl < 2 || a->ny < 2 || a->nx < 2) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: cube needs nt>=1, nl>=2, ny>=2, nx>=2");
    if (a->t_begin < 0 || a->t_count < 1 || a->t_begin + a->t_count > a->nt) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: [t_begin, t_begin+t_count) outside the cube");
    if (a->box_per_step != 0 && a->box_per_step != 1) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: box_per_step must be 0 or 1");
    if (a->n_box != (a->box_per_step ? a->t_count : 1)) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: n_box must be 1 (fixed box) or t_count (box_per_step)");
    if (a->reserved0) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: reserved0 must be 0");
    if (a->nxb_max < 2 || a->nyb_max < 2 || a->nxb_max > a->nx || a->nyb_max > a->ny) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: box extents must be 2..nx by 2..ny points");
    if (!a->lon_uniform && (!a->wlon_d || !a->glon_d)) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: non-uniform longitudes need wlon_d and glon_d");
    if (a->with_q) {
        if (!a->lattab_d || !a->levtab_d) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: with_q needs lattab_d and levtab_d");
        if (!a->dTdt_d && !a->tcoef_d) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: with_q needs dTdt_d or tcoef_d");
        if (!a->dTdt_d && !a->tm_d && a->nt < 2) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: dT/dt from the cube needs nt >= 2");
    }
    const bool packed = a->tm_d || a->tp_d;
    if (packed) {      // a box-packed series: per-step boxes at the origin of their slabs, time neighbours in cubes of their own
        if (!a->tm_d || !a->tp_d) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: a box-packed series needs both tm_d and tp_d");
        if (!a->box_per_step || !a->with_q || a->dTdt_d || !a->tcoef_d)
            return lec_set_error(LEC_ERR_ARG, "lec_rowstats: tm_d / tp_d go with box_per_step, with_q, tcoef_d and no dTdt_d");
    }
    const lec_tuning& tu = a->tuning;
    if (tu.kernel < LEC_KERNEL_AUTO || tu.kernel > LEC_KERNEL_BOX_PLANE) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: tuning.kernel is not an enum lec_kernel value");
    if (tu.order != LEC_ORDER_AUTO && tu.order != LEC_ORDER_MEMORY && tu.order != LEC_ORDER_XCD_LAT && tu.order != LEC_ORDER_XCD_TILED)
        return lec_set_error(LEC_ERR_ARG, "lec_rowstats: tuning.order is not an enum lec_order value");
    if (tu.tile_t < 0 || tu.tile_j < 0) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: tuning.tile_t / tile_j must be >= 0 (0 = default)");
    if (tu.f32_vec != 0 && tu.f32_vec != 2) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: tuning.f32_vec must be 0 or 2");
    if (tu.reserved[0] || tu.reserved[1]) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: tuning.reserved must be 0");
    int bt = 2, bk = 1, bj = 2;
    const bool tile_call = tu.kernel == LEC_KERNEL_BOX_TILE || tu.kernel == LEC_KERNEL_BOX_PLANE || (tu.kernel == LEC_KERNEL_AUTO && a->box_per_step);
    if (tile_call) {         // the box-tile kernel reads block_shape as the time steps per workgroup (0 = default)
        if (tu.block_shape != 0 && tu.block_shape != 1 && tu.block_shape != 2 && tu.block_shape != 4)
            return lec_set_error(LEC_ERR_ARG, "lec_rowstats: tuning.block_shape of a box-tile call (time steps per workgroup) must be 0, 1, 2 or 4");
    } else if (tu.block_shape) {
        bt = tu.block_shape / 100; bk = (tu.block_shape / 10) % 10; bj = tu.block_shape % 10;
        if (tu.block_shape < 0 || bt < 1 || bt > 2 || bk < 1 || bk > 2 || bj < 1 || bj > 2 || bt * bk * bj < 2)
            return lec_set_error(LEC_ERR_ARG, "lec_rowstats: tuning.block_shape must be 100 bt + 10 bk + bj with bt, bk, bj in {1, 2} and at least two waves");
    }
    const size_t esz = a->dtype == LEC_F32 ? 4 : 8;
    const int vecw = (int)(16 / esz);
    if (packed && (!tile_call || tu.block_shape > 1))
        return lec_set_error(LEC_ERR_ARG, "lec_rowstats: a box-packed series runs on the box-tile / box-plane kernels, one time step per workgroup");
    const void* cubes[8] = {a->tair_d, a->u_d, a->v_d, a->omega_d, a->geopt_d, a->dTdt_d, a->tm_d, a->tp_d};
    bool aligned = (a->nx % vecw) == 0, aligned8 = (a->nx % 2) == 0;
    for (const void* c : cubes) {
        if (!c) continue;
        if ((uintptr_t)c % esz) return lec_set_error(LEC_ERR_ARG, "lec_rowstats: cube pointer not aligned to its element size");
        if ((uintptr_t)c % 16) aligned = false;
        if ((uintptr_t)c % 8) aligned8 = false;
    }
    if (a->nxb_max > lec_max_row(a->dtype, a->dtype == LEC_F32 ? aligned8 : aligned, tu.kernel)) return lec_set_error(LEC_ERR_UNSUPPORTED, "lec_rowstats: box row longer than lec_max_row()");
    const long long nrows = (long long)a->t_count * a->nl * a->nyb_max;
    if (nrows > 0x7fffffffLL) return lec_set_error(LEC_ERR_UNSUPPORTED, "lec_rowstats: more than 2^31-1 rows in one call");

    RowParams p;
    p.T = a->tair_d; p.U = a->u_d; p.V = a->v_d; p.W = a->omega_d; p.P = a->geopt_d; p.DT = a->dTdt_d;
    p.TM = a->tm_d; p.TP = a->tp_d;
    p.nt = a->nt; p.nl = a->nl; p.ny = a->ny; p.nx = a->nx;
    p.t_begin = a->t_begin; p.t_count = a->t_count;
    p.n_box = a->box_per_step ? 2 : 1;      // kernels index the box tables by time step iff n_box != 1
    p.nxb_max = a->nxb_max; p.nyb_max = a->nyb_max;
    p.box = a->box_d; p.boxtab = a->boxtab_d; p.wlon = a->wlon_d; p.glon = a->glon_d;
    p.lattab = a->lattab_d; p.levtab = a->levtab_d; p.tcoef = a->tcoef_d;
    p.rows = a->rows_d;
    p.ntrips = 0; p.jrows = 0; p.cpx = 0;
    hipStream_t st = (hipStream_t)a->stream;
    const bool uni = a->lon_uniform != 0;
    const int wq = !a->with_q ? 0 : (a->dTdt_d ? 2 : 1);      // Q: none / dT/dt from the cube's time axis / dT/dt cube
    const bool fixed_time_stencil = (wq == 1 && !a->box_per_step);

    // Kernel family.  AUTO (what is measured and shipped): per-time-step boxes (the moving framework) run on the box-tile
    // kernel; all terms with dT/dt from the cube on ONE fixed box in fp64 storage on the row-block kernel (17.3-17.6 vs
    // 17.5-17.9 ms per 64 steps for one wave per row; fp32 storage: no gain measured); everything else on one wave per row.
    // The choice depends only on the kind of call, never on extents, so shards and chunks of one series use one family.
    int kernel = tu.kernel;
    const bool block_ok = fixed_time_stencil && a->geopt_d && a->t_count >= 2;
    if (kernel == LEC_KERNEL_AUTO) {
        if (a->box_per_step) kernel = LEC_KERNEL_BOX_TILE;
        // (stretched longitudes: the row-block instantiation carries per-column weight / gradient tables on top of its block state and
        // spills 28-44 B; one wave per row is 5 % ahead there -- 22.4-23.3 vs 23.7-25.1 ms per 64 steps, profiles/r04_notes.md section 7)
        else if (block_ok && a->dtype == LEC_F64 && tu.order == LEC_ORDER_AUTO && uni) kernel = LEC_KERNEL_ROW_BLOCK;
        else kernel = LEC_KERNEL_ROW_SWEEP;
    }
    if (kernel == LEC_KERNEL_ROW_BLOCK && !block_ok) kernel = LEC_KERNEL_ROW_SWEEP;       // a one-step shard of a row-block series: same bits

    // Workgroup -> row order (speed only; defaults from the A/B runs on MI355X, profiles/r01_notes.md).  Workgroups are dealt
    // to the 8 XCDs round-robin, so blockIdx % 8 labels the XCD: XCD_LAT gives every XCD a contiguous latitude chunk, walked
    // latitude-fastest per (time, level); XCD_TILED (all terms on one fixed box) walks tiles of tile_t time steps x tile_j
    // latitudes at one level, levels next; MEMORY is memory order.
    const bool two_sweep = kernel == LEC_KERNEL_TWO_SWEEP;
    p.order = (fixed_time_stencil && a->t_count >= 2 && !two_sweep) ? 7 : 2;
    if (tu.order == LEC_ORDER_MEMORY) p.order = 0;
    else if (tu.order != LEC_ORDER_AUTO) p.order = tu.order;
    if (p.order == 7 && (!fixed_time_stencil || two_sweep)) p.order = 2;
    p.jchunk = p.order ? (a->nyb_max + 7) / 8 : 0;
    const int ntile = (a->t_count + 3) / 4;                       // time tiles of (almost) equal size, at most 4 steps each
    p.tgroup = tu.tile_t ? tu.tile_t : (a->t_count + ntile - 1) / ntile;
    p.jgroup = tu.tile_j ? tu.tile_j : 8;
    if (p.jgroup > (a->nyb_max + 7) / 8) p.jgroup = (a->nyb_max + 7) / 8;    // never wider than an XCD's latitude chunk

    int rc;
    if (two_sweep) {
        // the two-sweep kernel of this file (deviation from the zonal mean, then products -- the reference's own order; Q per
        // point), kept as an independent formulation for cross-checks
        long long nblocks = p.order ? (long long)a->t_count * 8 * p.jchunk * a->nl : nrows;
        if (nblocks > 0x7fffffffLL) { p.order = 0; nblocks = nrows; }
        if (a->dtype == LEC_F64) rc = aligned ? launch_vec<double, 2>(p, uni, wq, (int)nblocks, st) : launch_vec<double, 1>(p, uni, wq, (int)nblocks, st);
        else                     rc = aligned ? launch_vec<float, 4>(p, uni, wq, (int)nblocks, st) : launch_vec<float, 1>(p, uni, wq, (int)nblocks, st);
    } else if (kernel == LEC_KERNEL_BOX_TILE || kernel == LEC_KERNEL_BOX_PLANE) {
        // Q per point (modes 1 / 2): the time neighbours of a moving box sum over other boxes, so the cross-time covariance
        // form of mode 3 does not apply
        RowParams pt = p;
        pt.tgroup = tu.tile_t;                              // time steps per tile group; 0 = the kernel's default (8)
        pt.jgroup = tu.tile_j;                              // levels per wave; 0 = chosen from the launch size
        // a box-packed fp64 series with its dT/dt cube (what every -t path of the product hands over for fp64 data): the planes come into
        // LDS by DMA (lec_boxplane.hip).  Decided by the kind of call and the slabs' shape, so every shard and chunk of a series agrees;
        // the records are bit-identical to the box-tile kernel's anyway (tested)
        const bool plane_ok = a->box_per_step && tu.block_shape <= 1 && lec_boxplane_serves(pt, a->dtype, uni, wq);
        if (kernel == LEC_KERNEL_BOX_PLANE && !plane_ok)
            return lec_set_error(LEC_ERR_ARG, "lec_rowstats: LEC_KERNEL_BOX_PLANE serves per-step boxes with geopotential on even longitudes, cubes at most 64 columns wide, "
                                              "dT/dt as a cube or (fp32 storage, box-packed) T of the two time neighbours as tm_d / tp_d");
        if (plane_ok && (kernel == LEC_KERNEL_BOX_PLANE || tu.kernel == LEC_KERNEL_AUTO)) rc = lec_launch_boxplane(pt, a->dtype, wq, st);
        else rc = lec_launch_boxtile(pt, a->dtype, uni, wq, tu.block_shape, st);
    } else {
        // Single-sweep row kernels.  All terms with dT/dt from the cube on one fixed box (the headline configuration, mode 3): a
        // row reads T(t+1) only and the time-derivative parts of [Q], [Q'T'] are completed from the records afterwards
        // (lec_qtime_kernel).  The row-block and the one-wave-per-row kernel give bit-identical records.
        const int mode = fixed_time_stencil ? 3 : wq;
        if (kernel == LEC_KERNEL_ROW_BLOCK) {
            RowParams pb = p;
            pb.order = 8; pb.tgroup = tu.tile_t ? tu.tile_t : 2; pb.jgroup = tu.tile_j ? tu.tile_j : 4;      // tile: 2 x 4 blocks at one level, levels next
            rc = lec_launch_rowblock(pb, a->dtype, aligned, aligned8, uni, bt, bk, bj, st);
        } else {
            rc = lec_launch_rowsweep(p, a->dtype, aligned, aligned8, uni, mode, tu.f32_vec, st);
        }
        if (rc == LEC_OK && mode == 3) rc = lec_launch_qtime(p, st);
    }
    if (rc == LEC_ERR_ARG) return lec_set_error(rc, "lec_rowstats: tuning.tile_j: the box-tile kernel walks at most 21 levels per wave, the box-plane kernel 42");
    if (rc != LEC_OK) return lec_set_error(rc, "lec_rowstats: row too long for the compiled kernels");
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    return LEC_OK;
}
