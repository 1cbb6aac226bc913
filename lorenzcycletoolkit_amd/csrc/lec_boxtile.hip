// lec_boxtile.hip -- stage 1 for short rows: per-time-step boxes of the moving (semi-Lagrangian) framework.
//
// A moving box is ~61 x 61 points per (time, level).  With one wave per 61-point row per level (lec_rowsweep.hip) the fixed
// per-row work -- a 20-statistic cross-lane reduction, the row epilogue, ~450 scalar instructions of row set-up -- is 20 x the
// useful arithmetic (~1100 instructions per row; 29 % of the HBM roofline in round 1).  Here ONE WAVE (a 64-thread workgroup: no
// barriers anywhere) owns four box rows of one time step and walks a chunk of levels; one PASS = 4 rows x 64 columns of one level:
//
//   load layout    lanes along longitude, WHOLE BOX ROWS per wave instruction (61 lanes x 8 B = 488 contiguous bytes): the
//                  access shape the memory system serves best for this pattern (tools/probes/probe_boxread.hip: 4.75 TB/s,
//                  against 3.0 TB/s for 128-byte column strips and 5.3 TB/s for a contiguous stream).  Everything a point's
//                  diabatic-heating residual needs is in that lane's registers (T at j+-1 are the wave's neighbouring rows, T at
//                  i+-1 come from the adjacent lanes by DPP, T at k+-1 from a three-level register window that slides down the
//                  levels: every T row is loaded ONCE per level chunk), so f = Q / cp is formed here, with wave-uniform row
//                  coefficients.  The loads of pass p+1 are in flight while pass p is reduced;
//   LDS transpose  only the six shifted values a .. f of a point go to LDS (row stride = 16 mod 32 words, one pad word per
//                  4 columns: conflict-free both ways).  With uniform longitudes the trapezoid's end points and the lanes
//                  outside the box are written as zeros -- they add nothing to any of the 20 monomial sums -- and the two end
//                  points reach the row's finishing lane through a small side array, so the inner loop is unweighted;
//   compute layout lane (r, g) owns row r and columns 4 g + (0..3): six LDS reads and the 20 shifted sums per point;
//   reduction      the 16 partial sums of a row and statistic meet in LDS (the tiles are dead by then): every lane stores its
//                  20 partials, lane (row, s) adds the 16 of its row in a fixed order -- ~80 instructions instead of ~240 for a
//                  DPP butterfly over 16 lanes;
//   epilogue       every four levels 16 lanes finish 4 levels x 4 rows, one record per lane (end-point terms, scaling, centred
//                  statistics from the shifted sums).
//
// Results are deterministic and depend only on the box of the time step (rows, column groups and the summation order are cut in
// box-relative rows / columns), so sharding / chunking a series changes no bit.  Any box size works: rows wider than 64 columns
// take several column chunks per level (no level window then: same arithmetic, T neighbours loaded per pass), four-row blocks
// cover any height.
//
// TIME GROUPS (round 4, template parameter TG > 1; per-point dT/dt from the cube, one column chunk): a workgroup of TG waves owns the
// same four rows of TG CONSECUTIVE time steps, one wave per step, and walks the levels in lock step.  T(t-1) and T(t+1) of a point --
// two of the 7.5 row requests per point, and the kernel is bound by the rate at which a CU's L1 completes line fills -- are then the
// neighbour waves' own T rows: every pass each wave publishes its four centre rows of T in LDS (in the tile that holds f, which is
// dead between the compute layout of one pass and the load layout of the next: no extra LDS) and reads its time neighbours' instead
// of loading them; only the group's first / last wave still loads T(t-1) / T(t+1) from memory: (6 + 8 / TG + 16) / 4 = 6.0 rows per
// point at TG = 4 instead of 7.5.  For that the waves of a group load on COMMON grid rows and columns -- the union of the group's
// boxes, which must fit 64 columns and the launch's row blocks (a track moves a column or a row every few steps; a group that does
// not fit simply loads its time neighbours itself, same arithmetic) -- while everything that decides a BIT stays box-relative: the
// position of a point in the LDS tiles (hence its summation group), the row's shift values, end points, latitude coefficients
// and record slot.  So the records do not depend on TG, on how a series is cut into groups, shards or chunks, or on whether a
// group shared: tested bit for bit against TG = 1 and the one-wave-per-row kernel.
// MEASURED (round 4, profiles/r04_notes.md section 2b): the L1 -> L2 line requests fall by 12 % (TG = 2) and 17 % (TG = 4) as designed,
// and the kernel is 3 % / 8 % SLOWER -- the time neighbours were L2 hits all along, the fabric moves the same bytes, and the lock step
// costs the waves their independence.  TG = 1 is what ships (kDefaultTG); TG = 2 / 4 stay selectable (tuning.block_shape) and tested.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/lec_hip.h"
#include "lec_internal.h"
#include "lec_rowcommon.h"
#include "lec_sweep.h"

using namespace lec;

// measurement builds only (tools/build_variant.sh), bit mask: 1 = no compute-layout phase, 2 = no global loads, 4 = no row epilogue,
// 8 = no quad reduction / hand-over, 16 = no load-layout arithmetic (values straight to LDS)
#ifndef LEC_BT_ABLATE
#define LEC_BT_ABLATE 0
#endif
#ifndef LEC_BT_DEEP2        // 1: fp64 storage with a dT/dt cube (box-packed series) prefetches Phi and dT/dt two passes ahead too (measurement knob)
#define LEC_BT_DEEP2 0
#endif
#ifndef LEC_BT_QUNROLL      // unroll factor of the compute layout's column loop (register pressure against LDS-read latency)
#define LEC_BT_QUNROLL 2
#endif

namespace {

constexpr int kWR = 4;               // box rows per wave = per pass
constexpr int kCW = 64;              // columns per pass (one wave-wide row segment)
constexpr int kS4 = 80;              // tile row stride in doubles: 64 columns + one pad per 4, = 16 (mod 32)
constexpr int kTile = kWR * kS4;
constexpr int kSide = 16;            // per-row side values: 5 shifts, f of the first point, a..f of the last point, T u v at the east column
#ifndef LEC_BT_LEVELS
#define LEC_BT_LEVELS 0
#endif
constexpr int kLevelsFixed = LEC_BT_LEVELS;     // > 0: levels per wave fixed at build time (experiments); 0: p.jgroup, chosen per launch
constexpr int kMinLevels = 5;                   // the T window's prologue (two extra level loads) is paid once per chunk of levels
constexpr int kMaxLevels = 21;                  // a wave keeps its chunk's static-stability coefficients one per lane (3 per level: `levv`)
static_assert(3 * kMaxLevels <= 64, "the level coefficients of a wave's chunk must fit one value per lane");
// measurement builds of the time groups (tools/build_variant.sh): LEC_BT_XABL bit 1 = no workgroup barriers (wrong results), bit 2 =
// the time neighbours still come from memory (publish + barriers kept: the cost of the lock step alone); LEC_BT_XDB = 1: the
// exchange double-buffered in LDS of its own (one barrier per pass; paid for by finishing rows two levels at a time instead of four)
#ifndef LEC_BT_XABL
#define LEC_BT_XABL 0
#endif
#ifndef LEC_BT_XDB
#define LEC_BT_XDB 0
#endif
constexpr int kLB = LEC_BT_XDB ? 2 : 4;      // levels whose rows are finished together (16 lanes: 4 levels x 4 rows)
#ifndef LEC_BT_TG
#define LEC_BT_TG 1                  // time steps per workgroup where the call allows it (tuning.block_shape overrides: 1, 2, 4)
#endif
constexpr int kDefaultTG = LEC_BT_TG;
constexpr int kPS = 65;              // stride between the statistics of the partial-sum array (odd: conflict-free both ways)

template <bool UNIFORM, int MODE> constexpr int n_tiles() { return (MODE == 0 ? 5 : 6) + (UNIFORM ? 0 : 1); }
template <bool UNIFORM, int MODE> constexpr int lds_doubles() { return n_tiles<UNIFORM, MODE>() * kTile + kLB * kWR * (kNA + kSide) + (LEC_BT_XDB ? 2 * kWR * kCW : 0); }
static_assert(kNA * kPS <= 5 * kTile, "the partial sums must fit the (dead) tiles they alias");

__device__ __forceinline__ int pos4(int c) { return c + (c >> 2); }     // LDS column of tile column c (one pad per 4 columns)

// value of lane `src` (wave-uniform) in every lane
__device__ __forceinline__ double lane_value(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// centred row statistics from the 20 shifted sums (already scaled): the formulas of finish_row (lec_sweep.h), one row per lane
__device__ __forceinline__ void finish_lane(const double (&tot)[kNA], double cT, double cU, double cV, double cW, double cP,
                                            double (&o)[22]) {
#pragma clang fp contract(off)
    const double da = tot[0], db = tot[1], dc = tot[2], dd = tot[3], de = tot[4], df = tot[5];
    o[0] = da + cT; o[1] = db + cU; o[2] = dc + cV; o[3] = dd + cW; o[4] = de + cP; o[5] = df;
    o[6] = tot[6] - da * da;        // [T'T']
    o[7] = tot[7] - db * db;        // [u'u']
    o[8] = tot[8] - dc * dc;        // [v'v']
    o[9] = tot[9] - dc * da;        // [v'T']
    o[10] = tot[10] - dd * da;      // [w'T']
    o[11] = tot[11] - db * dc;      // [u'v']
    o[12] = tot[12] - dd * db;      // [w'u']
    o[13] = tot[13] - dd * dc;      // [w'v']
    o[14] = tot[14] - dd * de;      // [w'Phi']
    o[15] = tot[15] - df * da;      // [Q'T']
    const double sTT = o[6], sUU = o[7], sVV = o[8];
    o[16] = tot[16] - 2 * da * tot[9] + da * da * dc + cV * sTT;       // [v T'T']
    o[17] = tot[17] - 2 * da * tot[10] + da * da * dd + cW * sTT;      // [w T'T']
    const double mU = cU + db, mV = cV + dc, mW = cW + dd;
    o[18] = 2 * mU * o[11] + mU * mU * mV + 2 * mV * sVV + mV * mV * mV;              // [K v]
    o[19] = 2 * mU * o[12] + mU * mU * mW + 2 * mV * o[13] + mV * mV * mW;            // [K w]
    o[20] = tot[18] - 2 * db * tot[11] + db * db * dc - 2 * dc * tot[8] + dc * dc * dc + cV * (sUU + sVV);      // [E v]
    o[21] = tot[19] - 2 * db * tot[12] + db * db * dd - 2 * dc * tot[13] + dc * dc * dd + cW * (sUU + sVV);     // [E w]
}

// MODE 0: T, u, v, omega (Phi if present), no Q;  1: dT/dt = ta T(t-1) + tb T(t) + tc T(t+1) per point;  2: dT/dt cube.
// WINDOW: rows fit one column chunk, so a wave walks its level chunk with T(k-1), T(k), T(k+1) sliding through registers;
// otherwise every pass loads its own T neighbours (wide boxes; same arithmetic, same bits).
// TG: waves per workgroup = consecutive time steps that share their T rows through LDS (see the head of the file); 1 = none.
__device__ __forceinline__ void lds_barrier() {           // LDS write -> workgroup barrier -> LDS read, the global loads stay in flight
    if (LEC_BT_XABL & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (the bound says what the kernel is: four-wave groups on table longitudes carry a seventh LDS tile -- 90 KB per workgroup, one per CU)
template <typename TIN, bool UNIFORM, int MODE, bool WINDOW, int TG>
__global__ void __launch_bounds__(64 * TG, (TG == 4 && !UNIFORM) ? 1 : 2) lec_boxtile_kernel(const RowParams p) {
    static_assert(TG == 1 || (MODE == 1 && WINDOW), "time groups: per-point dT/dt from the cube's time neighbours, rows of one column chunk");
    constexpr bool WITH_Q = MODE != 0;
    constexpr int NT = n_tiles<UNIFORM, MODE>();
#ifdef LEC_BT_PAD       // measurement builds: extra LDS per wave to cap the resident waves
    constexpr int kLds = lds_doubles<UNIFORM, MODE>() + LEC_BT_PAD;
#else
    constexpr int kLds = lds_doubles<UNIFORM, MODE>();
#endif
    __shared__ double sm_all[TG * kLds];
    const int wv = TG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;      // the wave = the time step inside the group
    double* const sm = sm_all + wv * kLds;
    double* const part = sm;                              // [kNA][kPS]: the lanes' partial sums of a level (aliases the tiles: they are dead by then)
    double* const stash = sm + NT * kTile;                // [kLB levels][kWR rows][kNA]: row totals waiting for their finishing lane
    double* const side = stash + kLB * kWR * kNA;         // [kLB levels][kWR rows][kSide]

    const int lane = threadIdx.x & 63;

    // block -> (time step [TG > 1: group of TG steps], level chunk, row block of 4).  Every XCD (blockIdx % 8, speed only) owns a contiguous chunk of time steps
    // and walks it in groups of tgroup steps: row block fastest (neighbouring blocks share their halo rows), then time step, then
    // level chunk, so the waves resident on an XCD are neighbours in latitude and time: the T rows at j+-1 (halo) and t+-1 are rows
    // a sibling loads as its own (L2)
    const int n_rb = p.jrows;                        // row blocks of the launch (TG > 1: room for the rows a group's boxes are apart)
    const int kchunk = p.jgroup;                     // levels per wave
    const int n_kc = (p.nl + kchunk - 1) / kchunk;
    const int xcd = blockIdx.x & 7;
    int q0 = blockIdx.x >> 3;
    const int rbi = q0 % n_rb; q0 /= n_rb;
    const int ti = q0 % p.tgroup; q0 /= p.tgroup;
    const int kc = q0 % n_kc;
    const int tin = (q0 / n_kc) * p.tgroup + ti;          // step (TG > 1: group) inside the XCD's chunk (jchunk = steps / groups per XCD here)
    const int grp = xcd * p.jchunk + tin;
    if (tin >= p.jchunk || grp * TG >= p.t_count) return;                 // (the same for every wave of the workgroup)
    // the last group of a series may be partial: its spare waves walk along as copies of the last step (barriers) and store nothing
    const bool live = grp * TG + wv < p.t_count;
    const int tl = live ? grp * TG + wv : p.t_count - 1;

    const int bi = (p.n_box == 1) ? 0 : tl;
    const int iw = p.box[4 * bi + 0], ie = p.box[4 * bi + 1], js = p.box[4 * bi + 2], jn = p.box[4 * bi + 3];
    const int nxb = ie - iw + 1, nyb = jn - js + 1;
    const int k0 = kc * kchunk, k1 = min(k0 + kchunk, p.nl);
    // TG > 1: the group's waves load on the grid rows / columns of the UNION of their boxes, if that fits one wave-wide row and the
    // launch's row blocks (else every wave keeps to its own box and loads its time neighbours itself).  sh / shj: where the wave's
    // own box starts inside the union -- everything that shapes a sum or a record below is in box-relative rows / columns.
    bool shared = false, any_act = true;
    int i0 = iw, j0 = js, uw = nxb, ujn = nyb - 1;
    if constexpr (TG > 1) {
        int imin = iw, imax = ie, jmin = js, jmax = jn;
#pragma unroll
        for (int g = 0; g < TG; ++g) {
            const int b = (p.n_box == 1) ? 0 : min(grp * TG + g, p.t_count - 1);
            imin = min(imin, p.box[4 * b + 0]); imax = max(imax, p.box[4 * b + 1]);
            jmin = min(jmin, p.box[4 * b + 2]); jmax = max(jmax, p.box[4 * b + 3]);
        }
        // the row blocks of a sharing group start at the union's first row: every step's records -- nyb_max rows, the padding rows of a
        // lower box included (they are written as zeros, stage 2 relies on it) -- must still lie inside the launch's blocks
        shared = (imax - imin + 1 <= kCW) && (jmax - jmin + 1 <= n_rb * kWR);
#pragma unroll
        for (int g = 0; g < TG; ++g) {
            const int b = (p.n_box == 1) ? 0 : min(grp * TG + g, p.t_count - 1);
            shared = shared && (p.box[4 * b + 2] - jmin + p.nyb_max <= n_rb * kWR);
        }
        if (shared) { i0 = imin; j0 = jmin; uw = imax - imin + 1; ujn = jmax - jmin; }
        any_act = false;
#pragma unroll
        for (int g = 0; g < TG; ++g) {                   // does the row block hold a box row of ANY step of the group?
            const int b = (p.n_box == 1) ? 0 : min(grp * TG + g, p.t_count - 1);
            const int f0 = rbi * kWR - (shared ? p.box[4 * b + 2] - j0 : 0);
            any_act = any_act || (f0 + kWR > 0 && f0 < p.box[4 * b + 3] - p.box[4 * b + 2] + 1);
        }
    }
    const int sh = TG > 1 ? iw - i0 : 0, shj = TG > 1 ? js - j0 : 0;
    const int jb0 = rbi * kWR - shj;                 // box-relative row of the wave's first row (TG > 1: may be negative)
    if (TG > 1 ? !any_act : jb0 >= nyb) {            // a row block that holds only padding rows of a box lower than nyb_max (or nothing)
        const int jlo = max(jb0, TG > 1 ? nyb : 0), nrow = min(jb0 + kWR, p.nyb_max) - jlo;
        for (int k = k0; k < k1 && live; ++k) {
            double* rec = p.rows + ((size_t)(tl * p.nl + k) * p.nyb_max + jlo) * LEC_NSTAT;
            for (int e = lane; e < nrow * LEC_NSTAT; e += 64) rec[e] = 0.0;
        }
        return;
    }
    // rows of the block at which the box ends inside it (TG > 1): their neighbour across the edge is a row of ANOTHER step's box
    // there -- loaded for that step's sake; the one-sided stencil gives it the coefficient 0, but 0 x NaN is NaN, so it is replaced
    // by the row itself, which is what the clamped row index of the plain kernel reads
    const int r_lo = (TG > 1 && jb0 < 0) ? -jb0 : -1, r_hi = (TG > 1) ? nyb - 1 - jb0 : -1;

    const int t = p.t_begin + tl;
    const size_t plane = (size_t)p.ny * p.nx;
    const size_t cube = plane * p.nl;
    const size_t t0off = (size_t)t * cube + (size_t)i0;
    const TIN* __restrict__ gT = (const TIN*)p.T + t0off;
    const TIN* __restrict__ gU = (const TIN*)p.U + t0off;
    const TIN* __restrict__ gV = (const TIN*)p.V + t0off;
    const TIN* __restrict__ gW = (const TIN*)p.W + t0off;
    const bool has_p = p.P != nullptr;
    const TIN* __restrict__ gP = (const TIN*)(has_p ? p.P : p.T) + t0off;
    // neighbours in time: the own time step where there is none (the coefficient is 0 there); MODE 2: the dT/dt cube
    // (a box-packed series: the cube's neighbouring steps hold other boxes; T(t-1), T(t+1) on THIS step's box come in cubes of their own)
    const bool packed = TG == 1 && MODE == 1 && p.TM != nullptr;
    const TIN* __restrict__ gD0 = (MODE == 2) ? (const TIN*)p.DT + t0off : (packed ? (const TIN*)p.TM + t0off : ((t > 0) ? gT - cube : gT));
    const TIN* __restrict__ gD1 = packed ? (const TIN*)p.TP + t0off : ((t < p.nt - 1) ? gT + cube : gT);
    double ta = 0, tb = 0, tc = 0;
    if (MODE == 1) { const double* tcf = p.tcoef + (size_t)t * 3; ta = tcf[0]; tb = tcf[1]; tc = tcf[2]; }
    // which time neighbours come from the group's LDS (the previous / next wave's own T rows) instead of from memory
    const bool d0_lds = TG > 1 && shared && wv > 0 && !(LEC_BT_XABL & 2);
    const bool d1_lds = TG > 1 && shared && wv < TG - 1 && grp * TG + wv + 1 < p.t_count && !(LEC_BT_XABL & 2);
    constexpr int kXoff = LEC_BT_XDB ? (NT * kTile + kLB * kWR * (kNA + kSide)) : 5 * kTile;
    TIN* const xch = reinterpret_cast<TIN*>(sm + kXoff);                          // own centre rows of T, [4][64] (the f tile, dead between passes)
    const TIN* const xlo = reinterpret_cast<const TIN*>(sm - (d0_lds ? kLds : 0) + kXoff);     // the previous / next wave's tile (the wave's own
    const TIN* const xhi = reinterpret_cast<const TIN*>(sm + (d1_lds ? kLds : 0) + kXoff);     // where it has no such neighbour: read and dropped)
    // TG > 1: the time neighbours' rows are fetched with BUFFER loads through a descriptor per wave whose size is 0 where the rows
    // come from LDS -- every lane is then out of range: the load returns zeros and touches no memory.  So the instruction stream
    // is the same for every wave of the group (no branch around a load: the compiler's s_waitcnt bookkeeping stays exact; with
    // wave-uniform branches around them it drained the prefetched rows with vmcnt(0) at every join and the kernel lost 20 %).
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<TIN*>(TG > 1 ? gD0 : gT), (short)0, d0_lds ? 0 : 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<TIN*>(TG > 1 ? gD1 : gT), (short)0, d1_lds ? 0 : 0x7fffffff, 0x00020000);
    auto ldb = [](__amdgpu_buffer_rsrc_t r, size_t o, unsigned col) -> TIN {
        if constexpr (sizeof(TIN) == 8) {
            const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(col * 8u), (int)(unsigned)(o * 8u), 0);
            return (TIN)__hiloint2double((int)v.y, (int)v.x);
        } else {
            return (TIN)__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(col * 4u), (int)(unsigned)(o * 4u), 0));
        }
    };
    // Row / level coefficients are wave-uniform, but inside the pass loop (which stores row records) the compiler would fetch them
    // with VECTOR loads followed by s_waitcnt vmcnt(0) -- draining the prefetched rows every time.  So they are loaded once, here,
    // spread over the lanes, and picked with v_readlane:
    //   latv: lane 4 i + j = coefficient j (d/dlat a, b, c; 1/dx) of the wave's row i;  levv: lane 3 kk + j = static-stability
    //   coefficient j of level k0 + kk -- so a wave walks at most kMaxLevels = 21 levels (launch_tiles enforces it)
    double latv = 0.0, levv = 0.0;
    if (WITH_Q) {
        const int jrow = min(max(jb0 + ((lane >> 2) & 3), 0), nyb - 1);
        latv = p.lattab[((size_t)bi * p.nyb_max + jrow) * 4 + (lane & 3)];
        levv = p.levtab[(size_t)min(k0 + lane / 3, p.nl - 1) * 3 + lane % 3];
    }
    double lga[kWR], lgb[kWR], lgc[kWR], lidx[kWR];       // the wave's rows (the same for every level): wave-uniform
#pragma unroll
    for (int i = 0; i < kWR; ++i) {
        lga[i] = lane_value(latv, 4 * i); lgb[i] = lane_value(latv, 4 * i + 1); lgc[i] = lane_value(latv, 4 * i + 2);
        lidx[i] = lane_value(latv, 4 * i + 3);
    }
    const double inv_xlen = p.boxtab[4 * bi + 0], h_rad = p.boxtab[4 * bi + 1], inv_hdeg = p.boxtab[4 * bi + 2];
    const double* __restrict__ wl = UNIFORM ? nullptr : p.wlon + (size_t)bi * p.nxb_max;
    const double* __restrict__ gl = UNIFORM ? nullptr : p.glon + (size_t)bi * p.nxb_max * 3;

    const int ncc = WINDOW ? 1 : (nxb + kCW - 1) / kCW;   // column chunks per row
    const int npass = (k1 - k0) * ncc;

    // element offsets (inside a level plane) of the wave's rows: rows -1 .. 4 relative to its first, clamped into the box
    unsigned roff[kWR + 2];
#pragma unroll
    for (int i = 0; i < kWR + 2; ++i) {
        int row = js + min(max(jb0 + i - 1, 0), nyb - 1);
        // a sharing group: the four centre rows are the union's rows whether or not this step's box holds them (a neighbour's may)
        if (TG > 1 && shared && i >= 1 && i <= kWR) row = j0 + min(rbi * kWR + i - 1, ujn);
        roff[i] = (unsigned)__builtin_amdgcn_readfirstlane(row * p.nx);
    }

    // ---- registers of the load layout.  T window: Tn = level k+1 (rows -1 .. 4), Tc = level k, Tm = level k-1 (rows 0 .. 3);
    // En / Ec: T at the columns just outside the chunk (lanes 0..31: c0 - 1, lanes 32..63: c0 + 64) of the centre rows at k+1 / k
    // The operands that do not depend on the level window are prefetched TWO passes ahead (two register sets, picked by pass parity at
    // compile time): the kernel is bound by how many loads a CU keeps in flight.  u, v, omega always; Phi and the dT/dt operands too where
    // the registers allow it (fp32 storage: 0.65 vs 0.69 ms; with fp64 storage their second set spills 108 B and costs 15 %).
    constexpr bool DEEP_ALL = sizeof(TIN) == 4 || (LEC_BT_DEEP2 && MODE == 2);
    TIN Tn[kWR + 2], Tc[kWR + 2], Tm[kWR], En[kWR] = {}, Ec[kWR] = {}, sU[2][kWR], sV[2][kWR], sW[2][kWR], sP[2][kWR], sD0[2][kWR], sD1[2][kWR];
    double rWl = 0.0, rG[3] = {0.0, 0.0, 0.0};            // non-uniform longitudes: the lane's trapezoid weight and d/dlon coefficients
    if (LEC_BT_ABLATE & 2) {
#pragma unroll
        for (int i = 0; i < kWR + 2; ++i) { Tn[i] = (TIN)(281 + lane + i); Tc[i] = (TIN)(280 + lane + i); }
#pragma unroll
        for (int i = 0; i < kWR; ++i) {
            Tm[i] = (TIN)(279 + lane); En[i] = Ec[i] = (TIN)280; sD0[0][i] = sD0[1][i] = sD1[0][i] = sD1[1][i] = (TIN)(281 + lane);
            sU[0][i] = sU[1][i] = (TIN)lane; sV[0][i] = sV[1][i] = (TIN)i; sW[0][i] = sW[1][i] = (TIN)0.1; sP[0][i] = sP[1][i] = (TIN)(lane * i);
        }
    }
    auto lev = [&](int k) -> size_t { return (size_t)min(max(k, 0), p.nl - 1) * plane; };
    // wave-uniform row pointer + the lane's 32-bit element offset
    auto ld = [](const TIN* __restrict__ row, unsigned off) -> TIN {
        return *reinterpret_cast<const TIN*>(reinterpret_cast<const char*>(row) + off * (unsigned)sizeof(TIN));
    };
    auto ldnt = [](const TIN* __restrict__ row, unsigned off) -> TIN {
#if defined(LEC_BT_PLAIN) && LEC_BT_PLAIN
        return *reinterpret_cast<const TIN*>(reinterpret_cast<const char*>(row) + off * (unsigned)sizeof(TIN));
#else
        return __builtin_nontemporal_load(reinterpret_cast<const TIN*>(reinterpret_cast<const char*>(row) + off * (unsigned)sizeof(TIN)));
#endif
    };
    // loads of the pass (level k, column chunk at c0).  `fresh`: the whole T window (no predecessor pass to inherit it from)
    auto issue_loads = [&](const int k, const int c0, const bool fresh) {
        if (LEC_BT_ABLATE & 2) return;
        const unsigned col = (unsigned)min(c0 + lane, uw - 1);
        const unsigned ecol = (unsigned)min(max(lane < 32 ? c0 - 1 : c0 + kCW, 0), nxb - 1);
        const size_t lk = lev(k);
        if (!UNIFORM) {
            const int ec = min(max(c0 + lane - sh, 0), nxb - 1);
            rWl = wl[ec];
            if (WITH_Q) { rG[0] = gl[3 * ec]; rG[1] = gl[3 * ec + 1]; rG[2] = gl[3 * ec + 2]; }
        }
        if (WITH_Q) {
            const size_t lp = lev(k + 1);
#pragma unroll
            for (int i = 0; i < kWR + 2; ++i) Tn[i] = ld(gT + lp + roff[i], col);
            if (!WINDOW) {      // one column chunk: the columns just outside it are outside the box and never used (one-sided ends)
#pragma unroll
                for (int i = 0; i < kWR; ++i) En[i] = ld(gT + lp + roff[i + 1], ecol);
            }
            if (fresh) {
                const size_t lm = lev(k - 1);
#pragma unroll
                for (int i = 0; i < kWR + 2; ++i) Tc[i] = ld(gT + lk + roff[i], col);
#pragma unroll
                for (int i = 0; i < kWR; ++i) { Tm[i] = ld(gT + lm + roff[i + 1], col); if (!WINDOW) Ec[i] = ld(gT + lk + roff[i + 1], ecol); }
            }
        } else {
#pragma unroll
            for (int i = 0; i < kWR; ++i) Tc[i + 1] = ld(gT + lk + roff[i + 1], col);
        }
        if (!DEEP_ALL) {
#pragma unroll
            for (int i = 0; i < kWR; ++i) {
                const size_t o = lk + roff[i + 1];
                if (WITH_Q || has_p) sP[0][i] = ldnt(gP + o, col);
                if (WITH_Q) {
                    if constexpr (TG > 1) {
                        sD0[0][i] = ldb(rs0, o, col);
                        sD1[0][i] = ldb(rs1, o, col);
                    } else {
                        sD0[0][i] = ld(gD0 + o, col);
                        if (MODE == 1) sD1[0][i] = ld(gD1 + o, col);
                    }
                }
            }
        }
    };
    // u, v, omega (and, DEEP_ALL, Phi and the dT/dt operands) of pass `ps` (level, column chunk) into register set SET
    auto issue_stream = [&](auto set_tag, const int ps) {
        constexpr int SET = decltype(set_tag)::value;
        if (LEC_BT_ABLATE & 2) return;
        const int kn = ps / ncc, c0 = (ps - kn * ncc) * kCW;
        const unsigned col = (unsigned)min(c0 + lane, uw - 1);
        const size_t lk = lev(k0 + kn);
#pragma unroll
        for (int i = 0; i < kWR; ++i) {
            const size_t o = lk + roff[i + 1];
            sU[SET][i] = ldnt(gU + o, col);
            sV[SET][i] = ldnt(gV + o, col);
            sW[SET][i] = ldnt(gW + o, col);
            if (DEEP_ALL) {
                if (WITH_Q || has_p) sP[SET][i] = ldnt(gP + o, col);
                if (WITH_Q) {
                    if constexpr (TG > 1) {
                        sD0[SET][i] = ldb(rs0, o, col);
                        sD1[SET][i] = ldb(rs1, o, col);
                    } else {
                        sD0[SET][i] = ld(gD0 + o, col);
                        if (MODE == 1) sD1[SET][i] = ld(gD1 + o, col);
                    }
                }
            }
        }
    };
    // the wave's centre rows of T at the level the NEXT pass works on -> its exchange tile (TG > 1)
    auto publish = [&](const int buf) {
#pragma unroll
        for (int i = 0; i < kWR; ++i) xch[(LEC_BT_XDB ? buf * kWR * kCW * (int)(sizeof(double) / sizeof(TIN)) : 0) + i * kCW + lane] = Tc[i + 1];
    };

    // ---- compute-layout roles: lane -> (row ci of the wave's four, column group cg of sixteen)
    const int ci = lane >> 4, cg = lane & 15;
    double acc[kNA];
#pragma unroll
    for (int s = 0; s < kNA; ++s) acc[s] = 0.0;
    double cT[kWR], cU[kWR], cV[kWR], cW[kWR], cP[kWR];    // shifts of the wave's rows: the row's first box element

    issue_loads(k0, 0, true);
    issue_stream(std::integral_constant<int, 0>{}, 0);
    if (npass > 1) issue_stream(std::integral_constant<int, 1>{}, 1);
    if (TG > 1) { publish(0); lds_barrier(); }                // (every wave of the workgroup makes the same number of passes)
    auto pass = [&](auto set_tag, const int ps) {
        constexpr int SET = decltype(set_tag)::value;
        if (TG > 1) {
            // the time neighbours' rows of this level, published by the previous / next wave at the end of their last pass: selected
            // over what the (empty) buffer load returned -- no branch (groups that do not share, and the group's two end waves, read
            // their own tile and keep the loaded rows); the second barrier lets every wave finish reading before anyone's load layout
            // overwrites the tile with f
            const int xb = LEC_BT_XDB ? (ps & 1) * kWR * kCW * (int)(sizeof(double) / sizeof(TIN)) : 0;
#pragma unroll
            for (int i = 0; i < kWR; ++i) {
                const TIN x0 = xlo[xb + i * kCW + lane], x1 = xhi[xb + i * kCW + lane];
                sD0[DEEP_ALL ? SET : 0][i] = d0_lds ? x0 : sD0[DEEP_ALL ? SET : 0][i];
                sD1[DEEP_ALL ? SET : 0][i] = d1_lds ? x1 : sD1[DEEP_ALL ? SET : 0][i];
            }
            if (!LEC_BT_XDB) lds_barrier();
        }
        const TIN (&rU)[kWR] = sU[SET]; const TIN (&rV)[kWR] = sV[SET]; const TIN (&rW)[kWR] = sW[SET]; const TIN (&rP)[kWR] = sP[DEEP_ALL ? SET : 0];
        const TIN (&rD0)[kWR] = sD0[DEEP_ALL ? SET : 0]; const TIN (&rD1)[kWR] = sD1[DEEP_ALL ? SET : 0];
        const int kk = ps / ncc, cc = ps - kk * ncc, k = k0 + kk, c0 = cc * kCW;
        const int slot = kk % kLB;
        double* const sd = side + slot * kWR * kSide;
        // ================= load layout: one point per lane, the wave's four rows =================
        {
#pragma clang fp contract(off)
            const int e = c0 + lane - sh;                                          // box-relative column
            const bool inside = (unsigned)e < (unsigned)nxb, first = e == 0, last = e == nxb - 1;
            const bool zero = UNIFORM ? (!inside || first || last) : !inside;     // contributes nothing to the sums taken in LDS
            const int llast = nxb - 1 - c0 + sh;                                   // lane of the row's last point (if in this chunk)
            const bool has_last = llast >= 0 && llast < kCW;
            double al = 0, be = 0, gm = 0;
            if (WITH_Q) { al = lane_value(levv, 3 * kk); be = lane_value(levv, 3 * kk + 1); gm = lane_value(levv, 3 * kk + 2); }
            const double wgt = UNIFORM ? 0.0 : (inside ? rWl : 0.0);
            const double g0 = rG[0], g1 = rG[1], g2 = rG[2];
#pragma unroll
            for (int i = 0; i < kWR; ++i) {
                const double T = (double)Tc[i + 1], U = (double)rU[i], V = (double)rV[i], W = (double)rW[i];
                const double P = (WITH_Q || has_p) ? (has_p ? (double)rP[i] : 0.0) : 0.0;
                if (cc == 0) {                                 // the row's first box element is lane 0 (TG > 1: sh) of the first chunk
                    cT[i] = lane_value(T, sh); cU[i] = lane_value(U, sh); cV[i] = lane_value(V, sh); cW[i] = lane_value(W, sh);
                    cP[i] = lane_value(P, sh);
                }
                double f = 0.0;
                if (WITH_Q && !(LEC_BT_ABLATE & 16)) {
                    const double ga_ = lga[i], gb_ = lgb[i], gc_ = lgc[i], idx = lidx[i];
                    const double Tl = from_prev_lane(T, WINDOW ? T : (double)Ec[i]), Tr = from_next_lane(T, WINDOW ? T : (double)Ec[i]);
                    double adv;                               // u dT/dx
                    if (UNIFORM) {
                        // centred; one-sided at the row ends: 2 (T[1] - T[0]), 2 (T[n-1] - T[n-2]) -- selects, no branches
                        const double dd = ((last ? T : Tr) - (first ? T : Tl)) * ((first || last) ? 2.0 : 1.0);
                        adv = (U * ((0.5 * inv_hdeg) * idx)) * dd;
                    } else {
                        // (TG > 1: the lanes beside the box's ends hold real grid points there -- possibly NaN -- where the plain kernel's
                        // clamped loads repeat the end point; the end coefficients are 0, but 0 x NaN is NaN)
                        const double Tl_ = (TG > 1 && first) ? T : Tl, Tr_ = (TG > 1 && last) ? T : Tr;
                        adv = U * fma(g2, Tr_, fma(g1, T, g0 * Tl_)) * idx;
                    }
                    const double Tjm = (TG > 1 && i == r_lo) ? T : (double)Tc[i], Tjp = (TG > 1 && i == r_hi) ? T : (double)Tc[i + 2];
                    const double sP_ = stencil3(ga_, Tjm, gc_, Tjp, gb_, T);
                    const double sS = stencil3(al, (double)Tm[i], gm, (double)Tn[i + 1], be, T);
                    const double dTdt = (MODE == 1) ? stencil3(ta, (double)rD0[i], tc, (double)rD1[i], tb, T) : (double)rD0[i];
                    f = fma(-W, sS, fma(V, sP_, dTdt + adv));
                }
                const double a = T - cT[i], b = U - cU[i], c = V - cV[i], d = W - cW[i], ee = P - cP[i];
                const int dst = i * kS4 + pos4(TG > 1 ? ((lane - sh) & (kCW - 1)) : lane);
                sm[0 * kTile + dst] = zero ? 0.0 : a;
                sm[1 * kTile + dst] = zero ? 0.0 : b;
                sm[2 * kTile + dst] = zero ? 0.0 : c;
                sm[3 * kTile + dst] = zero ? 0.0 : d;
                sm[4 * kTile + dst] = zero ? 0.0 : ee;
                if (WITH_Q) sm[5 * kTile + dst] = zero ? 0.0 : f;
                if (!UNIFORM) sm[(NT - 1) * kTile + dst] = wgt;
                // the row's side values for its finishing lane
                double* sr = sd + i * kSide;
                if (cc == 0 && lane == sh) { sr[0] = cT[i]; sr[1] = cU[i]; sr[2] = cV[i]; sr[3] = cW[i]; sr[4] = cP[i]; sr[5] = f; }
                if (has_last && lane == llast) {
                    sr[6] = a; sr[7] = b; sr[8] = c; sr[9] = d; sr[10] = ee; sr[11] = f;
                    sr[12] = T; sr[13] = U; sr[14] = V;
                }
            }
            if (WITH_Q && WINDOW) {       // slide the window one level down
#pragma unroll
                for (int i = 0; i < kWR; ++i) Tm[i] = Tc[i + 1];
#pragma unroll
                for (int i = 0; i < kWR + 2; ++i) Tc[i] = Tn[i];
            }
        }
        row_sync<64>();                    // one wave: its LDS operations are processed in order; only the compiler must not reorder them
        if (ps + 1 < npass) {              // in flight while this pass is reduced: the window rows and dT/dt operands of the next pass ...
            const int kn = (ps + 1) / ncc, cn = (ps + 1) - kn * ncc;
            issue_loads(k0 + kn, cn * kCW, !WINDOW);
        }
        if (ps + 2 < npass) issue_stream(set_tag, ps + 2);      // ... and the streamed operands of the pass after it, into the set just consumed
        // ================= compute layout: lane (ci, cg), columns 4 cg + q =================
        if (!(LEC_BT_ABLATE & 1)) {
#pragma clang fp contract(off)
#pragma unroll LEC_BT_QUNROLL
            for (int q = 0; q < 4; ++q) {
                const int src = ci * kS4 + 5 * cg + q;      // = row * stride + pos4(4 cg + q)
                const double a = sm[0 * kTile + src], b = sm[1 * kTile + src], c = sm[2 * kTile + src], d = sm[3 * kTile + src];
                const double ee = sm[4 * kTile + src];
                const double f = WITH_Q ? sm[5 * kTile + src] : 0.0;
                if (UNIFORM) accum20<true>(acc, 1.0, a, b, c, d, ee, f);
                else accum20<false>(acc, sm[(NT - 1) * kTile + src], a, b, c, d, ee, f);
            }
        }
        const bool row_done = (cc == ncc - 1);
        if (row_done && !(LEC_BT_ABLATE & 8)) {
            // the level's rows are complete: 16 partial sums per row and statistic -> one total, through LDS (the tiles are dead):
            // every lane stores its 20 partials, then lane (row, s) adds the 16 of its row in a fixed order (stride 65: no bank
            // conflicts either way)
#pragma clang fp contract(off)
            row_sync<64>();
#pragma unroll
            for (int s = 0; s < kNA; ++s) { part[s * kPS + lane] = acc[s]; acc[s] = 0.0; }
            row_sync<64>();
            const double* p0 = part + cg * kPS + ci * 16;               // statistic cg of row ci
            const double* p1 = part + (min(cg, 3) + 16) * kPS + ci * 16; // statistic 16 + cg (cg < 4)
            double t0 = p0[0], t1 = p1[0];
#pragma unroll
            for (int g = 1; g < 16; ++g) { t0 += p0[g]; t1 += p1[g]; }
            double* st = stash + (slot * kWR + ci) * kNA;
            st[cg] = t0;
            if (cg < 4) st[16 + cg] = t1;
            row_sync<64>();
        }
        // ---- up to kLB finished levels x 4 rows: one (level, row) per lane finishes its record
        if (row_done && (slot == kLB - 1 || k == k1 - 1) && !(LEC_BT_ABLATE & 4)) {
#pragma clang fp contract(off)
            // (the epilogue's lane arithmetic and its scale factors are made opaque here: hoisted out of the pass loop as loop invariants
            // they cost the shipped instantiation three spilled registers, reloaded from scratch in every pass -- this branch runs once
            // per four levels)
            // Round 6: the records go to LDS first (the hand-over array is free) and ALL lanes store them, sixteen lanes per 256-byte
            // record -- whole lines; one record per lane straight to memory is 13 store instructions of sixteen 16-byte pieces in sixteen
            // different lines each (lec_boxplane.hip: 0.767 -> 0.753 ms per 512 steps).
            constexpr int kRS = LEC_NSTAT + 2;           // record stride in the buffer (doubles): 16-byte aligned, eight lanes' 16-byte writes on different banks
            static_assert(kLB * kWR * kRS <= kNA * kPS, "the finished records must fit the hand-over array");
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int lv = ln >> 2, r = ln & 3;
            const int jb = jb0 + r;
            if (lv <= slot && jb >= 0 && jb < p.nyb_max && live) {
                dbl2_t* __restrict__ out = reinterpret_cast<dbl2_t*>(part + (lv * kWR + r) * kRS);
                // (few values live at a time: the epilogue runs while two passes' loads are in flight in registers)
                if (jb >= nyb) {                 // padding row of a box lower than nyb_max
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
                    if (UNIFORM) {
                        // the trapezoid's end points, weight 1/2 each: the first point has a = b = c = d = e = 0 (it is the shift),
                        // so only its f counts; the last point brings all 20 monomials
                        accum20<false>(tot, 0.5, sr[6], sr[7], sr[8], sr[9], sr[10], sr[11]);
                        tot[5] = fma(0.5, sr[5], tot[5]);
                        scale = h_rad * inv_xlen;
                    }
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
#pragma unroll
            for (int i = 0; i < kLB * kWR * (LEC_NSTAT / 2) / 64; ++i) {
                const int e = 64 * i + lane, rec = e >> 4, piece = e & 15;      // record (level slot, row) and the lane's 16 bytes of it
                const int rl = rec >> 2, jr = jb0 + (rec & 3);
                if (rl <= slot && jr >= 0 && jr < p.nyb_max && live) {
                    const dbl2_t v2 = *reinterpret_cast<const dbl2_t*>(part + rec * kRS + 2 * piece);
                    reinterpret_cast<dbl2_t*>(p.rows + ((size_t)(tl * p.nl + (k - slot + rl)) * p.nyb_max + jr) * LEC_NSTAT)[piece] = v2;
                }
            }
            row_sync<64>();
        }
        // the f tile is dead until the next load layout: the centre rows of the next level (the window has slid) for the neighbours
        if (TG > 1 && ps + 1 < npass) { publish((ps + 1) & 1); lds_barrier(); }
    };
    for (int ps = 0; ps < npass; ps += 2) {
        pass(std::integral_constant<int, 0>{}, ps);
        if (ps + 1 < npass) pass(std::integral_constant<int, 1>{}, ps + 1);
    }
}

template <typename TIN>
int launch_tiles(RowParams p, bool uniform, int mode, int tg, hipStream_t st) {
    const bool window = mode != 0 && p.nxb_max <= kCW;    // same arithmetic either way: only where the T neighbours come from differs
    // time groups: TG consecutive steps per workgroup share T(t +- 1) through LDS (per-point dT/dt from the cube, one column chunk)
    if (tg == 0) tg = kDefaultTG;
    if (!(mode == 1 && window) || p.t_count < 2 || p.TM) tg = 1;      // (box-packed: the steps' boxes share no grid rows in memory)
    const long long n_rb = tg > 1 ? (p.nyb_max + 3 + kWR - 1) / kWR : (p.nyb_max + kWR - 1) / kWR;     // (room for boxes up to 3 rows apart)
    p.jrows = (int)n_rb;
    const long long n_units = (p.t_count + tg - 1) / tg;  // workgroups along time
    p.jchunk = (int)((n_units + 7) / 8);                  // ... per XCD
    // levels per wave: as many as still leave kTargetWaves waves (four rounds of the 2048 the chip holds at two per SIMD)
    // -- long level walks read best (profiles/r02_notes.md: 4 rows x 37 levels 6 % ahead of 4 x 10), short launches need the waves
    if (kLevelsFixed > 0) p.jgroup = kLevelsFixed;
    if (p.jgroup > kMaxLevels) return LEC_ERR_ARG;       // never clamped: the caller asked for something the kernel cannot do
    if (p.jgroup < 1) {
        constexpr long long kTargetWaves = 8192;
        const long long per_chunk = 8LL * p.jchunk * n_rb * tg;
        const long long want = (kTargetWaves + per_chunk - 1) / per_chunk;
        const long long most = (p.nl + kMinLevels - 1) / kMinLevels, least = (p.nl + kMaxLevels - 1) / kMaxLevels;
        const long long n_kc0 = want < least ? least : (want > most ? most : want);
        p.jgroup = (int)((p.nl + n_kc0 - 1) / n_kc0);     // <= kMaxLevels: n_kc0 >= ceil(nl / kMaxLevels)
    }
    if (p.jgroup > p.nl) p.jgroup = p.nl;
    const long long n_kc = (p.nl + p.jgroup - 1) / p.jgroup;
    if (p.tgroup < 1) p.tgroup = 8;                       // time steps per tile group ...
    p.tgroup = (p.tgroup + tg - 1) / tg;                  // ... in workgroups
    if (p.tgroup > p.jchunk) p.tgroup = p.jchunk;
    const long long tgroups = (p.jchunk + p.tgroup - 1) / p.tgroup;
    const long long nblocks = 8LL * tgroups * p.tgroup * n_rb * n_kc;
    if (nblocks > 0x7fffffffLL) return LEC_ERR_UNSUPPORTED;
    if ((unsigned long long)p.ny * (unsigned long long)p.nx > 0xffffffffULL) return LEC_ERR_UNSUPPORTED;     // 32-bit offsets inside a plane
    dim3 grid((unsigned)nblocks), block(64 * tg);
#define LEC_TILE(U, M, W, G) hipLaunchKernelGGL((lec_boxtile_kernel<TIN, U, M, W, G>), grid, block, 0, st, p)
#define LEC_TILE_W(U, M) do { if (window) LEC_TILE(U, M, true, 1); else LEC_TILE(U, M, false, 1); } while (0)
#define LEC_TILE_G(U) do { if (tg == 4) LEC_TILE(U, 1, true, 4); else if (tg == 2) LEC_TILE(U, 1, true, 2); else LEC_TILE_W(U, 1); } while (0)
    if (uniform) { if (mode == 0) LEC_TILE(true, 0, false, 1); else if (mode == 1) LEC_TILE_G(true); else LEC_TILE_W(true, 2); }
    else { if (mode == 0) LEC_TILE(false, 0, false, 1); else if (mode == 1) LEC_TILE_G(false); else LEC_TILE_W(false, 2); }
#undef LEC_TILE_G
#undef LEC_TILE_W
#undef LEC_TILE
    return LEC_OK;
}

}  // namespace

// ---- dT/dt of a box-packed series as a cube (include/lec_hip.h: lec_dtdt): the kernel above's own per-point expression
namespace {
// a plain stream over the three cubes taken as ONE run of n_steps x step_elems elements: 3 reads + 1 fp64 write per point.  PAIRS: a
// thread takes two neighbouring elements with 16-byte (fp32: 8-byte) accesses; they may belong to two steps (an odd step size), so each
// element looks its own step's coefficients up
template <typename TIN, bool PAIRS>
__global__ void __launch_bounds__(256) lec_dtdt_kernel(const TIN* __restrict__ tm, const TIN* __restrict__ t, const TIN* __restrict__ tp,
                                                       const double* __restrict__ tcoef, double* __restrict__ out, long long step_elems, long long total) {
    const long long stride = (long long)gridDim.x * 256;
    if (PAIRS) {
        typedef TIN in2_t __attribute__((ext_vector_type(2)));
        const in2_t* m2 = reinterpret_cast<const in2_t*>(tm);
        const in2_t* c2 = reinterpret_cast<const in2_t*>(t);
        const in2_t* p2 = reinterpret_cast<const in2_t*>(tp);
        dbl2_t* o2 = reinterpret_cast<dbl2_t*>(out);
        for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total / 2; g += stride) {
            const long long s0 = (2 * g) / step_elems, s1 = (2 * g + 1) / step_elems;
            const in2_t a = __builtin_nontemporal_load(m2 + g), b = __builtin_nontemporal_load(c2 + g), c = __builtin_nontemporal_load(p2 + g);
            dbl2_t r;
            r.x = stencil3(tcoef[3 * s0], (double)a.x, tcoef[3 * s0 + 2], (double)c.x, tcoef[3 * s0 + 1], (double)b.x);
            r.y = stencil3(tcoef[3 * s1], (double)a.y, tcoef[3 * s1 + 2], (double)c.y, tcoef[3 * s1 + 1], (double)b.y);
            o2[g] = r;
        }
        if ((total & 1) && blockIdx.x == 0 && threadIdx.x == 0) {      // the last element of an odd run
            const long long e = total - 1, s = e / step_elems;
            out[e] = stencil3(tcoef[3 * s], (double)tm[e], tcoef[3 * s + 2], (double)tp[e], tcoef[3 * s + 1], (double)t[e]);
        }
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
            const long long s = e / step_elems;
            out[e] = stencil3(tcoef[3 * s], (double)tm[e], tcoef[3 * s + 2], (double)tp[e], tcoef[3 * s + 1], (double)t[e]);
        }
    }
}
}  // namespace

extern "C" int lec_dtdt(const lec_dtdt_args* a) {
    if (!a) return lec_set_error(LEC_ERR_ARG, "lec_dtdt: null args");
    if (!a->tm_d || !a->t_d || !a->tp_d || !a->tcoef_d || !a->out_d) return lec_set_error(LEC_ERR_ARG, "lec_dtdt: null pointer argument");
    if (a->dtype != LEC_F64 && a->dtype != LEC_F32) return lec_set_error(LEC_ERR_ARG, "lec_dtdt: dtype must be LEC_F64 or LEC_F32");
    if (a->n_steps < 1 || a->n_steps > 65535 || a->step_elems < 1) return lec_set_error(LEC_ERR_ARG, "lec_dtdt: needs 1 <= n_steps <= 65535 and step_elems >= 1");
    const size_t esz = a->dtype == LEC_F64 ? 8 : 4;
    const bool pairs = (((uintptr_t)a->tm_d | (uintptr_t)a->t_d | (uintptr_t)a->tp_d) % (2 * esz)) == 0 && ((uintptr_t)a->out_d % 16) == 0;
    const long long total = (long long)a->n_steps * a->step_elems;
    const long long want = ((pairs ? (total + 1) / 2 : total) + 255) / 256;
    dim3 grid((unsigned)(want < 65536 ? want : 65536)), block(256);
    hipStream_t st = (hipStream_t)a->stream;
#define LEC_DTDT(T, P) hipLaunchKernelGGL((lec_dtdt_kernel<T, P>), grid, block, 0, st, (const T*)a->tm_d, (const T*)a->t_d, (const T*)a->tp_d, a->tcoef_d, a->out_d, (long long)a->step_elems, total)
    if (a->dtype == LEC_F64) { if (pairs) LEC_DTDT(double, true); else LEC_DTDT(double, false); }
    else { if (pairs) LEC_DTDT(float, true); else LEC_DTDT(float, false); }
#undef LEC_DTDT
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    return LEC_OK;
}

// mode: 0 no Q, 1 dT/dt from the cube's time neighbours per point, 2 dT/dt cube; p.tgroup: time steps per tile group, p.jgroup: levels
// per wave (< 1: chosen here; more than 21: LEC_ERR_ARG); tg: time steps per workgroup (0 = default, 1, 2 or 4; mode 1 with rows of one
// column chunk only -- other calls run one wave per workgroup)
int lec_launch_boxtile(const lec::RowParams& p, int dtype, bool uniform, int mode, int tg, hipStream_t st) {
    if (tg != 0 && tg != 1 && tg != 2 && tg != 4) return LEC_ERR_ARG;
    return dtype == LEC_F64 ? launch_tiles<double>(p, uniform, mode, tg, st) : launch_tiles<float>(p, uniform, mode, tg, st);
}
