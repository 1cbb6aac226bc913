// lec_boxplane.hip -- stage 1 for a BOX-PACKED series of the moving framework in fp64 storage (include/lec_hip.h: cubes
// [nt][nl][ny][nx] whose step t holds box t alone, dT/dt as the series' own cube): the data come into LDS by DMA.
//
// lec_boxtile.hip was built for row fragments of a track-extent crop: lanes along longitude, one 488-byte box row per wave
// instruction into registers, the diabatic-heating residual formed there, six values per point transposed through LDS into the
// compute layout.  On a packed series that kernel is bound by its own instruction stream (profiles/r06_notes.md: 0.82 ms per 512
// steps, 0.55 without its global loads, 0.48 without loads AND compute layout -- the load layout, the transpose and the sums' hand-over
// are most of it), while the layout itself streams at 6.2 TB/s (tools/probes/probe_boxdma.hip: 0.54 ms).  In a packed series the rows
// a wave needs of one (step, level) plane are ONE contiguous run of bytes, so here:
//
//   loads          buffer_load_dwordx4 ... lds: 1-KiB pieces of the run straight into LDS (no registers, no ds_write, no address
//                  arithmetic per row), 13 wave instructions per level -- T with one halo row either side (6 rows), u, v, omega, Phi
//                  and dT/dt (4 rows each) -- in THREE buffers: the sets of levels k + 1 and k + 2 are in flight while level k is
//                  computed, waited for with a counted vmcnt.  This access shape needs ~100 KB in flight per CU to stream at the
//                  part's rate (the probes; double buffering -- one set in flight while a wave computes -- ran at 0.79 ms where the
//                  bare DMA stream takes 0.52): 4 waves x 2 sets x 13 KB.  The LDS image of a tile IS the memory image (row pitch =
//                  the slab's nx); a wave's 40 KB of LDS is a quarter of the CU's: one wave per SIMD, registers are plentiful;
//   one layout     lane (r, g) owns row r of the wave's four and columns 4 g + (0..3) -- the compute layout of lec_boxtile.hip -- and
//                  reads everything a point needs from the tiles: its five operands, T at i +- 1 and j +- 1 of the SAME tile.  T at
//                  k +- 1 are the lane's own points one level back and ahead: the T tile runs one level AHEAD of the other five
//                  (set k = {T(k + 1), u v omega Phi dT/dt (k)}), T(k), T(k - 1) and the horizontal stencils of level k (formed while
//                  T(k)'s tile was in LDS) wait in registers -- 16 values per lane;
//   sums           the 20 shifted sums per point, the 16-way hand-over through LDS (in the set that has just been consumed) and the
//                  row epilogue are lec_boxtile.hip's, expression by expression and in its order.
//
// Every expression that decides a bit is the one of lec_boxtile.hip (same products, same rounding points, same summation groups and
// order: groups of four columns, sixteen groups in order, end points in the epilogue), so the records are BIT-IDENTICAL to that
// kernel's and results still do not depend on how a series is sharded, chunked or which of the two kernels ran (tested).
//
// Serves: fp64 storage, even longitudes, dT/dt as a cube, Phi present, slabs at most 64 columns wide.  Everything else of a
// per-step-box call runs on lec_boxtile.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"
#include "lec_rowcommon.h"
#include "lec_sweep.h"

using namespace lec;

#ifndef LEC_BP_ABLATE       // measurement builds only (tools/build_variant.sh), bit mask: 1 = no arithmetic on the tiles, 2 = no DMA
#define LEC_BP_ABLATE 0
#endif
#ifndef LEC_BP_NT           // cache policy of the once-read planes (u, v, omega, Phi, dT/dt): 1 = nontemporal
#define LEC_BP_NT 1
#endif

namespace {

constexpr int kWR = 4;                    // box rows per wave
constexpr int kMaxW = 64;                 // widest slab (columns) the tiles hold
constexpr int kSide = 16;                 // per-row side values (lec_boxtile.hip): 5 shifts, f of the first point, a..f and T u v of the last
constexpr int kLB = 2;                    // levels whose rows are finished together (8 lanes: 2 levels x 4 rows)
#ifndef LEC_BP_AHEAD
#define LEC_BP_AHEAD 2                    // sets in flight behind the one being computed (measurement knob: 1 = double buffering)
#endif
constexpr int kAhead = LEC_BP_AHEAD;
constexpr int kBufs = kAhead + 1;
constexpr int kPS = 65;                   // stride between the statistics of the partial-sum array
constexpr int kMaxLevels = 21;            // a wave keeps its chunk's static-stability coefficients one per lane (3 per level)
constexpr int kMinLevels = 5;
constexpr int kTileT = (kWR + 2) * kMaxW * 8;     // bytes: T with its two halo rows
constexpr int kTileF = kWR * kMaxW * 8;           // bytes: one of the five other planes
constexpr int kPiecesT = kTileT / 1024, kPiecesF = kTileF / 1024;
constexpr int kSet = kTileT + 5 * kTileF;         // one level's set: 13,312 B
constexpr int kPieces = kPiecesT + 5 * kPiecesF;  // DMA instructions per set: 13
constexpr int kStashOff = kNA * kPS * 8;          // row totals of kLB levels, behind the partial sums in the consumed set
constexpr int kLdsBytes = kBufs * kSet + kLB * kWR * kSide * 8;      // three sets + the side values of two levels: 40,960 B = a quarter of a CU's LDS
static_assert(kTileT % 1024 == 0 && kTileF % 1024 == 0, "tiles are whole 1-KiB pieces");
static_assert(kStashOff + kLB * kWR * kNA * 8 <= kSet, "the partial sums and the row totals must fit the (consumed) set they alias");
static_assert(kAhead >= 1 && kAhead <= 2 && kPieces * kAhead + 13 <= 63, "the counted waits must fit vmcnt");
static_assert(3 * kMaxLevels <= 64, "the level coefficients of a wave's chunk must fit one value per lane");

__device__ __forceinline__ double lane_value(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// One 1-KiB piece of a run of box rows: lane l's 16 bytes at (run start + soff + 16 l) -> LDS lds_addr + 16 l, by a BUFFER load whose
// resource is the run itself (base = its first byte, num_records = its length).  The range check is per dword: the piece that straddles
// the run's end (a run of an odd number of doubles ends in the middle of a lane's 16 bytes) delivers its in-range half and zeros, lanes
// past the end deliver zeros and touch no memory (tools/probes/probe_dma_range.hip) -- so every lane issues every piece (the vmcnt
// arithmetic needs a fixed count), nothing is clamped and nothing beyond the run is ever read.  M0 carries the LDS address; the
// compiler does not see a vector-memory instruction here, so the waits are ours (wait_vm) -- its own counted waits can only over-wait
// (the counter is in issue order).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t run_resource(const void* first_byte, unsigned bytes) {
    const unsigned long long b = (unsigned long long)first_byte;
    u32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xffffu;      // 48-bit base, stride 0: a raw buffer
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
template <bool NT>
__device__ __forceinline__ void dma16(u32x4_t run, unsigned voff, unsigned soff, unsigned lds_addr) {
    if (NT) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" :: "s"(lds_addr), "v"(voff), "s"(run), "s"(soff) : "memory");
    else    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(run), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// centred row statistics from the 20 shifted sums: lec_boxtile.hip's finish_lane (the formulas of finish_row, lec_sweep.h)
__device__ __forceinline__ void finish_lane(const double (&tot)[kNA], double cT, double cU, double cV, double cW, double cP, double (&o)[22]) {
#pragma clang fp contract(off)
    const double da = tot[0], db = tot[1], dc = tot[2], dd = tot[3], de = tot[4], df = tot[5];
    o[0] = da + cT; o[1] = db + cU; o[2] = dc + cV; o[3] = dd + cW; o[4] = de + cP; o[5] = df;
    o[6] = tot[6] - da * da;
    o[7] = tot[7] - db * db;
    o[8] = tot[8] - dc * dc;
    o[9] = tot[9] - dc * da;
    o[10] = tot[10] - dd * da;
    o[11] = tot[11] - db * dc;
    o[12] = tot[12] - dd * db;
    o[13] = tot[13] - dd * dc;
    o[14] = tot[14] - dd * de;
    o[15] = tot[15] - df * da;
    const double sTT = o[6], sUU = o[7], sVV = o[8];
    o[16] = tot[16] - 2 * da * tot[9] + da * da * dc + cV * sTT;
    o[17] = tot[17] - 2 * da * tot[10] + da * da * dd + cW * sTT;
    const double mU = cU + db, mV = cV + dc, mW = cW + dd;
    o[18] = 2 * mU * o[11] + mU * mU * mV + 2 * mV * sVV + mV * mV * mV;
    o[19] = 2 * mU * o[12] + mU * mU * mW + 2 * mV * o[13] + mV * mV * mW;
    o[20] = tot[18] - 2 * db * tot[11] + db * db * dc - 2 * dc * tot[8] + dc * dc * dc + cV * (sUU + sVV);
    o[21] = tot[19] - 2 * db * tot[12] + db * db * dd - 2 * dc * tot[13] + dc * dc * dd + cW * (sUU + sVV);
}

__global__ void __launch_bounds__(64, 1) lec_boxplane_kernel(const RowParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char sm_raw[kLdsBytes];
    const unsigned lds0 = (unsigned)(uintptr_t)sm_raw;
    double* const side = reinterpret_cast<double*>(sm_raw + kBufs * kSet);   // [kLB levels][kWR rows][kSide]
    const int lane = threadIdx.x & 63;

    // block -> (time step, level chunk, row block of 4): lec_boxtile.hip's order (every XCD a contiguous chunk of time steps, row
    // block fastest, then time step, then level chunk)
    const int n_rb = p.jrows, kchunk = p.jgroup, n_kc = (p.nl + kchunk - 1) / kchunk;
    const int xcd = blockIdx.x & 7;
    int q0 = blockIdx.x >> 3;
    const int rbi = q0 % n_rb; q0 /= n_rb;
    const int ti = q0 % p.tgroup; q0 /= p.tgroup;
    const int kc = q0 % n_kc;
    const int tin = (q0 / n_kc) * p.tgroup + ti;
    const int tl = xcd * p.jchunk + tin;
    if (tin >= p.jchunk || tl >= p.t_count) return;

    const int bi = (p.n_box == 1) ? 0 : tl;
    const int iw = p.box[4 * bi + 0], ie = p.box[4 * bi + 1], js = p.box[4 * bi + 2], jn = p.box[4 * bi + 3];
    const int nxb = ie - iw + 1, nyb = jn - js + 1;
    const int k0 = kc * kchunk, k1 = min(k0 + kchunk, p.nl);
    const int jb0 = rbi * kWR;
    if (jb0 >= nyb) {                     // a row block that holds only padding rows of a box lower than nyb_max: zero records
        const int nrow = min(jb0 + kWR, p.nyb_max) - jb0;
        for (int k = k0; k < k1; ++k) {
            double* rec = p.rows + ((size_t)(tl * p.nl + k) * p.nyb_max + jb0) * LEC_NSTAT;
            for (int e = lane; e < nrow * LEC_NSTAT; e += 64) rec[e] = 0.0;
        }
        return;
    }

    const int W = p.nx;                   // the slab's row pitch = the tiles' row pitch (<= 64)
    const int t = p.t_begin + tl;
    const size_t plane = (size_t)p.ny * W;
    const size_t cube = plane * p.nl;
    auto lev = [&](int k) -> size_t { return (size_t)min(max(k, 0), p.nl - 1) * plane; };
    // the wave's rows: box rows jb0 .. jb0 + 3 (cut at the box's last row); T also one row either side where the box has one
    const int h0 = max(jb0 - 1, 0), h1 = min(jb0 + kWR + 1, nyb), r1 = min(jb0 + kWR, nyb);
    const unsigned bytesT = (unsigned)((h1 - h0) * W * 8), bytesF = (unsigned)((r1 - jb0) * W * 8);
    const double* const gT = (const double*)p.T + (size_t)t * cube + (size_t)(js + h0) * W;
    const size_t fbase = (size_t)t * cube + (size_t)(js + jb0) * W;
    const double* const gU = (const double*)p.U + fbase;
    const double* const gV = (const double*)p.V + fbase;
    const double* const gW = (const double*)p.W + fbase;
    const double* const gP = (const double*)p.P + fbase;
    const double* const gD = (const double*)p.DT + fbase;
    const unsigned voff = 16u * lane;
    auto issue_T = [&](int k, unsigned set) {
        if (LEC_BP_ABLATE & 2) return;
        const u32x4_t run = run_resource(gT + lev(k), bytesT);
#pragma unroll
        for (int i = 0; i < kPiecesT; ++i) dma16<false>(run, voff, 1024u * i, set + 1024u * i);      // (halo rows are the neighbouring wave's own rows: default policy)
    };
    auto issue_F = [&](int k, unsigned set) {
        if (LEC_BP_ABLATE & 2) return;
        const size_t lk = lev(k);
        const double* g5[5] = {gU + lk, gV + lk, gW + lk, gP + lk, gD + lk};
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const u32x4_t run = run_resource(g5[f], bytesF);
#pragma unroll
            for (int i = 0; i < kPiecesF; ++i) dma16<LEC_BP_NT != 0>(run, voff, 1024u * i, set + kTileT + kTileF * f + 1024u * i);
        }
    };

    // ---- lane roles: (row ci of the wave's four, column group cg of sixteen), columns 4 cg + q.  Lanes 0..31 hold groups 0..7 of all
    // four rows, lanes 32..63 groups 8..15: the 32 lanes that an LDS read serves together then touch 32 different 8-byte banks
    // whenever the row pitch is odd (61 columns) -- with (row, group) = (lane / 16, lane % 16) groups g and g + 8 would collide
    const int ci = (lane >> 3) & 3, cg = (lane & 7) | ((lane >> 5) << 3);
    const int slot16 = ci * 16 + cg;                       // the lane's slot in the hand-over arrays: (row, group) as lec_boxtile.hip numbers them
    const int jb = jb0 + ci, jbc = min(jb, nyb - 1);       // (rows past the box's last one walk along on it and store nothing)
    // element offsets inside the tiles (doubles).  T tile: row 0 = box row h0; the other tiles: row 0 = box row jb0
    const int rT = (jbc - h0) * W + iw, rTm = (max(jbc - 1, 0) - h0) * W + iw, rTp = (min(jbc + 1, nyb - 1) - h0) * W + iw;
    const int rF = (jbc - jb0) * W + iw;
    int col[4], cl[4], cr[4];
    bool zero[4], first[4], last[4];
    double fac[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 4 * cg + q;
        first[q] = c == 0; last[q] = c == nxb - 1;
        zero[q] = c == 0 || c >= nxb - 1;                  // the trapezoid's end points (added in the epilogue with weight 1/2) and columns beside the box
        col[q] = min(c, nxb - 1);
        cl[q] = max(col[q] - 1, 0); cr[q] = min(col[q] + 1, nxb - 1);      // one-sided at the row ends: the point itself stands in for the missing neighbour
        fac[q] = (first[q] || last[q]) ? 2.0 : 1.0;
    }
    // coefficients of the lane's row (d/dlat a, b, c; 1 / dx) and of the wave's levels (one per lane: picked with v_readlane, so that
    // no table load -- a vector load the compiler would wait for with vmcnt(0) -- sits inside the level loop)
    const double* lt = p.lattab + ((size_t)bi * p.nyb_max + jbc) * 4;
    const double ga = lt[0], gb = lt[1], gc = lt[2], idx = lt[3];
    const double levv = p.levtab[(size_t)min(k0 + lane / 3, p.nl - 1) * 3 + lane % 3];
    const double inv_xlen = p.boxtab[4 * bi + 0], h_rad = p.boxtab[4 * bi + 1], inv_hdeg = p.boxtab[4 * bi + 2];
    const double cx = (0.5 * inv_hdeg) * idx;

    // ---- prologue: T(k0)'s tile (as "set k0 - 1", into the last buffer), the lane's points of T(k0 - 1), then the first kAhead sets
    double Tm[4], Tc[4], sPc[4], ddc[4], cT;
    issue_T(k0, lds0 + kAhead * kSet);
    {
        const double* g = (const double*)p.T + (size_t)t * cube + lev(k0 - 1) + (size_t)(js + jbc) * W + iw;
#pragma unroll
        for (int q = 0; q < 4; ++q) Tm[q] = g[col[q]];
    }
    issue_T(k0 + 1, lds0);
    issue_F(k0, lds0);
    if (kAhead > 1 && k0 + 1 < k1) {
        issue_T(k0 + 2, lds0 + kSet);
        issue_F(k0 + 1, lds0 + kSet);
        wait_vm<2 * kPieces>();
    } else {
        wait_vm<kPieces>();
    }
    {
#pragma clang fp contract(off)
        const double* sT = reinterpret_cast<const double*>(sm_raw + kAhead * kSet);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Tc[q] = sT[rT + col[q]];
            ddc[q] = (sT[rT + cr[q]] - sT[rT + cl[q]]) * fac[q];
            sPc[q] = stencil3(ga, sT[rTm + col[q]], gc, sT[rTp + col[q]], gb, Tc[q]);
        }
        cT = sT[rT];
    }

    double acc[kNA];
#pragma unroll
    for (int s = 0; s < kNA; ++s) acc[s] = 0.0;

    double keep0 = 0.0, keep1 = 0.0;      // the row totals of the level in slot 0, until its partner level is done
    bool stored = false;                  // did the previous level's epilogue issue its record stores (at least 13, in front of the next set)?
    int buf = 0;
    for (int k = k0; k < k1; ++k) {
        const int kk = k - k0, slot = kk % kLB;
        wait_lds();                       // the reads of the buffer that is refilled now (the previous level's hand-over and epilogue, the prologue) are done
        // set k + kAhead goes into the buffer level k - 1 has left; then wait for set k -- everything but the sets behind it (and the
        // record stores issued between them: never fewer than 13 where an epilogue ran)
        const int ahead = min(kAhead, k1 - 1 - k);                 // sets behind set k once this level's issue is done
        if (k + kAhead < k1) {
            const int bnew = buf == 0 ? kAhead : buf - 1;          // = (buf + kAhead) % kBufs
            const unsigned set = lds0 + bnew * kSet;
            issue_T(k + kAhead + 1, set);
            issue_F(k + kAhead, set);
        }
        if (ahead == 2) { if (stored) wait_vm<2 * kPieces + 13>(); else wait_vm<2 * kPieces>(); }
        else if (ahead == 1) { if (stored) wait_vm<kPieces + 13>(); else wait_vm<kPieces>(); }
        else wait_vm<0>();
        stored = false;
        const double* sT = reinterpret_cast<const double*>(sm_raw + buf * kSet);
        const double* sU = sT + kTileT / 8;
        const double* sV = sU + kTileF / 8;
        const double* sW = sV + kTileF / 8;
        const double* sP = sW + kTileF / 8;
        const double* sD = sP + kTileF / 8;
        double* const sd = side + (slot * kWR + ci) * kSide;
        if (!(LEC_BP_ABLATE & 1)) {
#pragma clang fp contract(off)
            const double al = lane_value(levv, 3 * kk), be = lane_value(levv, 3 * kk + 1), gm = lane_value(levv, 3 * kk + 2);
            const double cU = sU[rF], cV = sV[rF], cW = sW[rF], cP = sP[rF];      // shifts: the row's first box element
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double Tn = sT[rT + col[q]];                                // T one level down: the lane's own point
                const double U = sU[rF + col[q]], V = sV[rF + col[q]], Wv = sW[rF + col[q]], P = sP[rF + col[q]], D = sD[rF + col[q]];
                const double T = Tc[q];
                const double sS = stencil3(al, Tm[q], gm, Tn, be, T);
                const double adv = (U * cx) * ddc[q];
                const double f = fma(-Wv, sS, fma(V, sPc[q], D + adv));
                const double a = T - cT, b = U - cU, c = V - cV, d = Wv - cW, ee = P - cP;
                if (!zero[q]) accum20<true>(acc, 1.0, a, b, c, d, ee, f);
                if (first[q]) { sd[0] = cT; sd[1] = cU; sd[2] = cV; sd[3] = cW; sd[4] = cP; sd[5] = f; }
                if (last[q]) {
                    sd[6] = a; sd[7] = b; sd[8] = c; sd[9] = d; sd[10] = ee; sd[11] = f;
                    sd[12] = T; sd[13] = U; sd[14] = V;
                }
                // the horizontal stencils of the NEXT level, while its tile is here
                const double ddn = (sT[rT + cr[q]] - sT[rT + cl[q]]) * fac[q];
                const double sPn = stencil3(ga, sT[rTm + col[q]], gc, sT[rTp + col[q]], gb, Tn);
                Tm[q] = T; Tc[q] = Tn; ddc[q] = ddn; sPc[q] = sPn;
            }
            cT = sT[rT];
        }
        double* const part = reinterpret_cast<double*>(sm_raw + buf * kSet);            // (the set is consumed)
        double* const stash = reinterpret_cast<double*>(sm_raw + buf * kSet + kStashOff);   // [kLB levels][kWR rows][kNA]
        const bool finish = slot == kLB - 1 || k == k1 - 1;
        {
            // the level's rows are complete: 16 partial sums per row and statistic -> one total, through LDS: every lane stores its 20
            // partials, then lane (row, s) adds the 16 of its row in a fixed order (lec_boxtile.hip's hand-over).  The totals of the
            // level in slot 0 wait in two registers for their partner level: the consumed set is refilled before that one is done
#pragma clang fp contract(off)
            row_sync<64>();
#pragma unroll
            for (int s = 0; s < kNA; ++s) { part[s * kPS + slot16] = acc[s]; acc[s] = 0.0; }
            row_sync<64>();
            const double* p0 = part + cg * kPS + ci * 16;                 // statistic cg of row ci
            const double* p1 = part + (min(cg, 3) + 16) * kPS + ci * 16;  // statistic 16 + cg (cg < 4)
            double t0 = p0[0], t1 = p1[0];
#pragma unroll
            for (int g = 1; g < 16; ++g) { t0 += p0[g]; t1 += p1[g]; }
            if (finish) {
                if (slot > 0) {
                    double* s0 = stash + ci * kNA;
                    s0[cg] = keep0;
                    if (cg < 4) s0[16 + cg] = keep1;
                }
                double* st = stash + (slot * kWR + ci) * kNA;
                st[cg] = t0;
                if (cg < 4) st[16 + cg] = t1;
            } else {
                keep0 = t0; keep1 = t1;
            }
            row_sync<64>();
        }
        // ---- up to kLB finished levels x 4 rows: one (level, row) per lane finishes its record (lec_boxtile.hip's epilogue)
        if (finish) {
#pragma clang fp contract(off)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int lv = ln >> 2, r = ln & 3;
            const int jr = jb0 + r;
            if (lv <= slot && jr < p.nyb_max) {
                const int kout = k - slot + lv;
                dbl2_t* __restrict__ out = reinterpret_cast<dbl2_t*>(p.rows + ((size_t)(tl * p.nl + kout) * p.nyb_max + jr) * LEC_NSTAT);
                if (jr >= nyb) {                 // padding row of a box lower than nyb_max
#pragma unroll
                    for (int s = 0; s < LEC_NSTAT / 2; ++s) { dbl2_t z; z.x = 0.0; z.y = 0.0; out[s] = z; }
                } else {
                    const double* sr = side + (lv * kWR + r) * kSide;
                    const double* st = stash + (lv * kWR + r) * kNA;
                    double tot[kNA];
#pragma unroll
                    for (int s = 0; s < kNA; ++s) tot[s] = st[s];
                    double scale = inv_xlen;
                    asm volatile("" : "+v"(scale));
                    // the trapezoid's end points, weight 1/2 each: the first point has a = b = c = d = e = 0 (it is the shift), so only
                    // its f counts; the last point brings all 20 monomials
                    accum20<false>(tot, 0.5, sr[6], sr[7], sr[8], sr[9], sr[10], sr[11]);
                    tot[5] = fma(0.5, sr[5], tot[5]);
                    scale = h_rad * inv_xlen;
#pragma unroll
                    for (int s = 0; s < kNA; ++s) tot[s] = tot[s] * ((s == 5 || s == 15) ? scale * kCp : scale);      // <f>, <fa>: Q = cp f
                    double o22[22];
                    finish_lane(tot, sr[0], sr[1], sr[2], sr[3], sr[4], o22);
#pragma unroll
                    for (int s = 0; s < 11; ++s) { dbl2_t v2; v2.x = o22[2 * s]; v2.y = o22[2 * s + 1]; out[s] = v2; }
                    dbl2_t e2;
                    e2.x = sr[0]; e2.y = sr[12]; out[LEC_S_TW / 2] = e2;       // T, u, v at the west / east box column
                    e2.x = sr[1]; e2.y = sr[13]; out[LEC_S_UW / 2] = e2;
                    e2.x = sr[2]; e2.y = sr[14]; out[LEC_S_VW / 2] = e2;
                    e2.x = 0.0; e2.y = 0.0; out[LEC_S_SPARE / 2] = e2; out[LEC_S_SPARE / 2 + 1] = e2;
                }
            }
            row_sync<64>();
            stored = true;               // (lane 0 finishes a record or a padding row whenever this branch runs: at least 13 store instructions)
        }
        buf = buf == kAhead ? 0 : buf + 1;
    }
}

}  // namespace

// true where lec_boxplane_kernel serves the call (the KIND of call and the slabs' shape -- the same for every shard and chunk of a
// series; never the boxes' extents or the step count)
bool lec_boxplane_serves(const lec::RowParams& p, int dtype, bool uniform, int mode) {
    return dtype == LEC_F64 && uniform && mode == 2 && p.DT && p.P && !p.TM && !p.TP && p.n_box != 1 && p.nx <= kMaxW && p.nxb_max <= kMaxW;
}

// p.tgroup: time steps per tile group, p.jgroup: levels per wave (< 1: chosen here; more than 21: LEC_ERR_ARG) -- as lec_launch_boxtile
int lec_launch_boxplane(lec::RowParams p, hipStream_t st) {
    const long long n_rb = (p.nyb_max + kWR - 1) / kWR;
    p.jrows = (int)n_rb;
    p.jchunk = (int)((p.t_count + 7) / 8);                // time steps per XCD
    if (p.jgroup > kMaxLevels) return LEC_ERR_ARG;
    if (p.jgroup < 1) {
        // levels per wave: as many as still leave kTargetWaves waves (the chip holds 1024 of these at four per CU); a chunk's first set
        // costs a T tile and a pipeline fill of its own
        constexpr long long kTargetWaves = 8192;
        const long long per_chunk = 8LL * p.jchunk * n_rb;
        const long long want = (kTargetWaves + per_chunk - 1) / per_chunk;
        const long long most = (p.nl + kMinLevels - 1) / kMinLevels, least = (p.nl + kMaxLevels - 1) / kMaxLevels;
        const long long n_kc0 = want < least ? least : (want > most ? most : want);
        p.jgroup = (int)((p.nl + n_kc0 - 1) / n_kc0);
    }
    if (p.jgroup > p.nl) p.jgroup = p.nl;
    const long long n_kc = (p.nl + p.jgroup - 1) / p.jgroup;
    if (p.tgroup < 1) p.tgroup = 8;
    if (p.tgroup > p.jchunk) p.tgroup = p.jchunk;
    const long long tgroups = (p.jchunk + p.tgroup - 1) / p.tgroup;
    const long long nblocks = 8LL * tgroups * p.tgroup * n_rb * n_kc;
    if (nblocks > 0x7fffffffLL) return LEC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(lec_boxplane_kernel, dim3((unsigned)nblocks), dim3(64), 0, st, p);
    return LEC_OK;
}
