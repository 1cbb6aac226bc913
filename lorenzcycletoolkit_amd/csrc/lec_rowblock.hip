// lec_rowblock.hip -- stage 1, row-block kernel for the all-terms configuration on a fixed box.
//
// The diabatic-heating residual needs T at t+-1, k+-1 and j+-1 besides the row itself: with one
// independent wave per row (lec_rowsweep.hip) that is 7 T loads per row, 6 of them rows that other
// waves fetch as their own.  Here a workgroup of BT x BK x BJ waves owns a block of neighbouring rows
// (BT time steps x BK levels x BJ latitudes, each dimension 1 or 2) and walks them in lock step: every
// trip each wave loads its own vector of T, publishes it in LDS, and reads the in-block neighbours from
// LDS; only the neighbour on the outer side of each dimension is still a global (L2) load.  A 2x2x2
// block issues 8 vector loads per row and trip instead of 11, and the vector-memory pipe, not HBM, is
// what limits this configuration (profiles/r01_notes.md).
//
// The stencils are linear, so which side comes from LDS only swaps coefficients (wave-uniform, hoisted
// out of the loop); no per-lane selects.  In time only T(t+1) is needed (cross-time covariance, see
// sweep_elems): the earlier wave of a time pair reads it from its mate, the later one loads it.  Waves whose row lies outside the processed range still walk
// a real neighbouring row (the halo time step of a shard, the next XCD's first latitude) so that their
// block mates read correct data; they just do not store.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/lec_hip.h"
#include "lec_internal.h"
#include "lec_rowcommon.h"
#include "lec_sweep.h"

using namespace lec;

namespace {

// LDS write -> barrier -> LDS read without draining the global loads still in flight
// (__syncthreads() would wait for vmcnt(0) as well)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename TIN, int VEC, bool UNIFORM, int BT, int BK, int BJ>
__global__ void __launch_bounds__(64 * BT * BK * BJ, LEC_MINW_SINGLE) lec_rowblock_kernel(const RowParams p) {
    constexpr int NW = BT * BK * BJ;
    __shared__ double red[NW][kRound * red_stride(64)];
    __shared__ double tot[NW][24];
    __shared__ __attribute__((aligned(16))) TIN xch[2][NW][64 * VEC];
#if defined(LEC_RB_PAD) && LEC_RB_PAD > 0
    __shared__ double occupancy_pad[LEC_RB_PAD];              // measurement builds: fewer resident workgroups per CU
    if (p.nt < 0) { occupancy_pad[threadIdx.x] = p.rows[threadIdx.x]; __syncthreads(); p.rows[threadIdx.x] = occupancy_pad[(threadIdx.x * 7) % LEC_RB_PAD]; }
#endif

    // wave-uniform: everything derived from the wave index stays in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tid = threadIdx.x & 63;
    const int wt = wave % BT, wk = (wave / BT) % BK, wj = wave / (BT * BK);

    // block -> cell: the XCD owns a latitude chunk; tiles of tgroup x jgroup cells at one level cell, level cells next
    const int xcd = blockIdx.x & 7;
    const int q = blockIdx.x >> 3;
    const int tcells = (p.t_count + BT - 1) / BT, kcells = (p.nl + BK - 1) / BK, jcells = (p.jchunk + BJ - 1) / BJ;
    const int tile = p.tgroup * p.jgroup;
    int tile_id = q / tile;
    const int within = q - tile_id * tile;
    const int t_in = within % p.tgroup, j_in = within / p.tgroup;
    const int kc = tile_id % kcells; tile_id /= kcells;
    const int tgc = (tcells + p.tgroup - 1) / p.tgroup;
    const int tc_ = (tile_id % tgc) * p.tgroup + t_in, jc_ = (tile_id / tgc) * p.jgroup + j_in;
    if (tc_ >= tcells || jc_ >= jcells) return;              // whole workgroup
    if (xcd * p.jchunk + jc_ * BJ >= p.nyb_max) return;       // whole workgroup

    const int tl0 = tc_ * BT + wt, k0 = kc * BK + wk, jl0 = jc_ * BJ + wj;
    const bool store = tl0 < p.t_count && k0 < p.nl && jl0 < p.jchunk && xcd * p.jchunk + jl0 < p.nyb_max;
    const int tl = min(tl0, p.t_count - 1);                   // record index (used only when store)
    const int t = min(p.t_begin + tl0, p.nt - 1);             // data row: the real neighbour when it exists
    const int k = min(k0, p.nl - 1);
    const int jb = min(xcd * p.jchunk + jl0, p.nyb_max - 1);

    const int iw = p.box[0], ie = p.box[1], js = p.box[2];
    const int nxb = ie - iw + 1, nyb = p.nyb_max;
    double* __restrict__ out = p.rows + ((size_t)(tl * p.nl + k) * p.nyb_max + jb) * LEC_NSTAT;
    const int j = js + jb;
    const size_t plane = (size_t)p.ny * p.nx;
    const size_t cube = plane * p.nl;
    const size_t rowoff = (size_t)t * cube + (size_t)k * plane + (size_t)j * p.nx + iw;
    const int shift = (VEC > 1) ? (int)(rowoff % VEC) : 0;
    const int e0_last = ((nxb - 1 + shift) / VEC) * VEC - shift;

    const TIN* __restrict__ rT = (const TIN*)p.T + rowoff;
    const TIN* __restrict__ rU = (const TIN*)p.U + rowoff;
    const TIN* __restrict__ rV = (const TIN*)p.V + rowoff;
    const TIN* __restrict__ rW = (const TIN*)p.W + rowoff;
    const TIN* __restrict__ rP = (const TIN*)p.P + rowoff;

    const double inv_xlen = p.boxtab[0];
    const double h_rad = p.boxtab[1];
    const double inv_hdeg = p.boxtab[2];
    const double* __restrict__ wl = UNIFORM ? nullptr : p.wlon;
    const double* __restrict__ gl = UNIFORM ? nullptr : p.glon;

    // stencil coefficients and neighbour rows (missing neighbours: own row, coefficient 0 in the tables)
    const double* lt = p.lattab + (size_t)jb * 4;
    const double ga = lt[0], gb = lt[1], gc = lt[2], inv_dx = lt[3];
    const double* lv = p.levtab + (size_t)k * 3;
    const double al = lv[0], be = lv[1], gm = lv[2];
    const TIN* rTjm = (jb > 0) ? rT - p.nx : rT;
    const TIN* rTjp = (jb < nyb - 1) ? rT + p.nx : rT;
    const TIN* rTkm = (k > 0) ? rT - plane : rT;
    const TIN* rTkp = (k < p.nl - 1) ? rT + plane : rT;
    const TIN* rTtp = (t < p.nt - 1) ? rT + cube : rT;
    const TIN* rTtm = (t > 0) ? rT - cube : rT;
    // the first time step of the launch has no processed predecessor: it forms the backward cross-time covariance too
    const bool both = (tl0 == 0);

    // per dimension: B == 1 -> both neighbours global (g0 = minus, g1 = plus);
    // B == 2 -> g0 = the outer neighbour (global), the inner one is the block mate's row in LDS
    const bool fwd_global = (BT == 1) || (wt == BT - 1);      // T(t+1): a global load, or the time mate's row in LDS
    const TIN* gK0 = (BK == 1 || wk == 0) ? rTkm : rTkp;  const double cK0 = (BK == 1 || wk == 0) ? al : gm;
    const TIN* gJ0 = (BJ == 1 || wj == 0) ? rTjm : rTjp;  const double cJ0 = (BJ == 1 || wj == 0) ? ga : gc;
    const double cK1 = (BK == 1 || wk == 0) ? gm : al;    // coefficient of the other side (global if B == 1, LDS if B == 2)
    const double cJ1 = (BJ == 1 || wj == 0) ? gc : ga;
    const int mateT = (BT == 2) ? (wt == 0 ? wave + 1 : wave - 1) : wave;
    const int mateK = (BK == 2) ? (wk == 0 ? wave + BT : wave - BT) : wave;
    const int mateJ = (BJ == 2) ? (wj == 0 ? wave + BT * BK : wave - BT * BK) : wave;

    // shifts: the row's first box element (wave-uniform scalar loads)
    SweepRow r;
    r.nxb = nxb;
    r.cT = (double)rT[0]; r.cU = (double)rU[0]; r.cV = (double)rV[0]; r.cW = (double)rW[0]; r.cP = (double)rP[0];
    r.cx = 0.5 * inv_hdeg * inv_dx; r.inv_dx = inv_dx; r.wl = wl; r.gl = gl;
    r.cTf = (double)rTtp[0]; r.cTb = both ? (double)rTtm[0] : 0.0;
    // T, u, v at the east box column (boundary terms), fetched now so that the row does not end on a load
    const double eT = (double)rT[nxb - 1], eU = (double)rU[nxb - 1], eV = (double)rV[nxb - 1];

    double acc[kNA], xacc[kNX];
#pragma unroll
    for (int s = 0; s < kNA; ++s) acc[s] = 0.0;
#pragma unroll
    for (int s = 0; s < kNX; ++s) xacc[s] = 0.0;

    QCoef qc;
    qc.k0 = cK0; qc.k1 = cK1; qc.km = be; qc.j0 = cJ0; qc.j1 = cJ1; qc.jm = gb;

    auto trip = [&](auto edge_tag, auto both_tag, const int it) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr bool BOTH = decltype(both_tag)::value;
        const int el = it * 64 * VEC - shift;                // box element of lane 0 (wave-uniform)
        const int e0 = el + tid * VEC;
        const bool lane_in = !EDGE || (e0 <= e0_last);
        const unsigned eo = (unsigned)((EDGE ? min(e0, e0_last) : e0) + shift);
        TIN fT[VEC], fU[VEC], fV[VEC], fW[VEC], fP[VEC];
        QRaw<TIN, VEC> qr;                                    // x0 = the outer (global) neighbour, x1 = the other side
        load_vec<TIN, VEC, false>(rT - shift, eo, fT);       // first: the block mates wait for it
        if (fwd_global) load_vec<TIN, VEC, false>(rTtp - shift, eo, qr.tf);
        if (BOTH) load_vec<TIN, VEC, true>(rTtm - shift, eo, qr.tb);
        load_vec<TIN, VEC, false>(gK0 - shift, eo, qr.k0);
        load_vec<TIN, VEC, false>(gJ0 - shift, eo, qr.j0);
        if (BK == 1) load_vec<TIN, VEC, false>(rTkp - shift, eo, qr.k1);
        if (BJ == 1) load_vec<TIN, VEC, false>(rTjp - shift, eo, qr.j1);
        load_vec<TIN, VEC, true>(rU - shift, eo, fU);
        load_vec<TIN, VEC, true>(rV - shift, eo, fV);
        load_vec<TIN, VEC, true>(rW - shift, eo, fW);
        load_vec<TIN, VEC, true>(rP - shift, eo, fP);
        // in-row neighbours beyond the wave's end lanes: wave-uniform addresses, scalar loads
        const int il = EDGE ? min(max(el - 1, 0), nxb - 1) : el - 1;
        const int ir = EDGE ? min(max(el + 64 * VEC, 0), nxb - 1) : el + 64 * VEC;
        const double tl0 = (double)rT[il], tr0 = (double)rT[ir];

        // publish the own T vector (storage type), read the block mates'
        if (NW > 1) {
            TIN* mine = &xch[it & 1][wave][tid * VEC];
#pragma unroll
            for (int q2 = 0; q2 < VEC; ++q2) mine[q2] = fT[q2];
            lds_barrier();
            const TIN* mt = &xch[it & 1][mateT][tid * VEC];
            const TIN* mk = &xch[it & 1][mateK][tid * VEC];
            const TIN* mj = &xch[it & 1][mateJ][tid * VEC];
#pragma unroll
            for (int q2 = 0; q2 < VEC; ++q2) {
                if (BT == 2 && !fwd_global) qr.tf[q2] = mt[q2];
                if (BK == 2) qr.k1[q2] = mk[q2];
                if (BJ == 2) qr.j1[q2] = mj[q2];
            }
        }
        const double tl_edge = from_prev_lane((double)fT[VEC - 1], tl0);
        const double tr_edge = from_next_lane((double)fT[0], tr0);
        sweep_elems<VEC, UNIFORM, EDGE, 3, BOTH>(acc, xacc, r, e0, lane_in, fT, fU, fV, fW, fP, tl_edge, tr_edge, qr, qc);
    };

    // every wave of the block runs the same trips (same box row geometry, one barrier per trip in either variant): the
    // LDS barriers stay matched
    const int ntrips = p.ntrips;
    const int mid_end = min((nxb - 1 + shift) / (64 * VEC), ntrips);
    auto sweep = [&](auto both_tag) {
        trip(std::true_type{}, both_tag, 0);
#pragma unroll 1
        for (int it = 1; it < mid_end; ++it) trip(std::false_type{}, both_tag, it);
#pragma unroll 1
        for (int it = max(mid_end, 1); it < ntrips; ++it) trip(std::true_type{}, both_tag, it);
    };
    if (both) sweep(std::true_type{}); else sweep(std::false_type{});

    finish_row<64, kRound, true>(acc, xacc, red[wave], tot[wave], tid, UNIFORM ? h_rad * inv_xlen : inv_xlen, r, out, store);
    if (store && tid == 0) {
        out[LEC_S_TW] = r.cT; out[LEC_S_UW] = r.cU; out[LEC_S_VW] = r.cV;
        out[LEC_S_TE] = eT; out[LEC_S_UE] = eU; out[LEC_S_VE] = eV;
    }
}

template <typename TIN, int VEC, bool UNIFORM>
int launch_block(RowParams& p, int bt, int bk, int bj, hipStream_t st) {
    const int nvec = (p.nxb_max + VEC - 1) / VEC + (VEC > 1 ? 1 : 0);
    p.ntrips = (nvec + 63) / 64;
    p.jchunk = (p.nyb_max + 7) / 8;
    const long long tcells = (p.t_count + bt - 1) / bt, kcells = (p.nl + bk - 1) / bk, jcells = (p.jchunk + bj - 1) / bj;
    if (p.tgroup > tcells) p.tgroup = (int)tcells;
    if (p.jgroup > jcells) p.jgroup = (int)jcells;
    if (p.tgroup < 1) p.tgroup = 1;
    if (p.jgroup < 1) p.jgroup = 1;
    const long long tgc = (tcells + p.tgroup - 1) / p.tgroup, jgc = (jcells + p.jgroup - 1) / p.jgroup;
    const long long nblocks = 8LL * jgc * tgc * kcells * p.tgroup * p.jgroup;
    if (nblocks > 0x7fffffffLL) return LEC_ERR_UNSUPPORTED;
    dim3 grid((unsigned)nblocks), block(64 * bt * bk * bj);
#define LEC_BLK(A, B, C) hipLaunchKernelGGL((lec_rowblock_kernel<TIN, VEC, UNIFORM, A, B, C>), grid, block, 0, st, p)
    const int code = bt * 100 + bk * 10 + bj;
    switch (code) {
        case 222: LEC_BLK(2, 2, 2); break;
        case 221: LEC_BLK(2, 2, 1); break;
        case 212: LEC_BLK(2, 1, 2); break;
        case 122: LEC_BLK(1, 2, 2); break;
        case 211: LEC_BLK(2, 1, 1); break;
        case 121: LEC_BLK(1, 2, 1); break;
        case 112: LEC_BLK(1, 1, 2); break;
        default: return LEC_ERR_ARG;
    }
#undef LEC_BLK
    return LEC_OK;
}

}  // namespace

// all terms, dT/dt from the cube, ONE fixed box, Phi present, fp64 or fp32 storage; p.tgroup / p.jgroup are the
// tile extents in cells (blocks of bt time steps / bj latitudes)
int lec_launch_rowblock(lec::RowParams p, int dtype, bool aligned, bool aligned8, bool uniform, int bt, int bk, int bj, hipStream_t st) {
    if (dtype == LEC_F64) {
        if (aligned) return uniform ? launch_block<double, 2, true>(p, bt, bk, bj, st) : launch_block<double, 2, false>(p, bt, bk, bj, st);
        return uniform ? launch_block<double, 1, true>(p, bt, bk, bj, st) : launch_block<double, 1, false>(p, bt, bk, bj, st);
    }
    if (aligned8) return uniform ? launch_block<float, 2, true>(p, bt, bk, bj, st) : launch_block<float, 2, false>(p, bt, bk, bj, st);
    return uniform ? launch_block<float, 1, true>(p, bt, bk, bj, st) : launch_block<float, 1, false>(p, bt, bk, bj, st);
}
