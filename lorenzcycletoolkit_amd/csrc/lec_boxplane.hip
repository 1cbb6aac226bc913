// lec_boxplane.hip -- stage 1 for a BOX-PACKED series of the moving framework in fp64 storage (include/lec_hip.h: cubes
// [nt][nl][ny][nx] whose step t holds box t alone, dT/dt as the series' own cube): the planes' rows go STRAIGHT into the layout the
// sums are taken in.
//
// lec_boxtile.hip was built for row fragments of a track-extent crop: lanes along longitude, one 488-byte box row per wave
// instruction, the diabatic-heating residual formed in that layout, six values per point transposed through LDS into the compute
// layout.  On a packed series that kernel is bound by its own instruction stream (profiles/r06_notes.md: 0.82 ms per 512 steps, 0.55
// without its global loads, 0.48 without loads AND compute layout: the load layout, the transpose and the hand-over of the sums are
// most of it), while the layout itself streams at 6.2 TB/s (tools/probes/probe_boxdma.hip: 0.52-0.55 ms).  Here a lane has ONE role
// -- (row r of the wave's four, columns 4 g .. 4 g + 3), the compute layout of lec_boxtile.hip -- and:
//
//   u v omega Phi dT/dt   need no neighbour: the lane's four points of a plane are 32 contiguous bytes, fetched as two 16-byte buffer
//                  loads straight into its registers (default cache policy: the two loads of a lane touch the same lines, the second
//                  hits in L1; with nt the probe runs 18 % slower).  No LDS, no transpose, no load-layout arithmetic.  The resource of
//                  a load is the run of the wave's rows in that plane, so what lies past it -- the next plane, or nothing at all at the
//                  cube's end -- reads as zero and is never touched (tools/probes/probe_dma_range.hip);
//   T              needs i +- 1, j +- 1, k +- 1.  The wave's rows with one halo row either side are ONE contiguous run of the plane
//                  (6 x 488 bytes): three flat 16-byte-per-lane loads, written to an LDS tile as they lie in memory.  Two tiles
//                  alternate: T(k) -- a lane reads its points and their four horizontal neighbours -- and T(k + 1) -- its points; T(k - 1)
//                  at its points waits in four registers, so every T row is loaded once per level chunk;
//   prefetch       the five planes of level k + 1 are requested before level k's are waited for (two register sets, picked by the level's
//                  parity at compile time), T's run of level k + 2 as soon as that of k + 1 is in its tile; 8 waves per CU as
//                  lec_boxtile.hip (the access shape needs ~100 KB in flight per CU to stream at the part's rate: 8 x 13 KB and more);
//   sums           the 20 shifted sums per point, the 16-way hand-over through LDS and the row epilogue are lec_boxtile.hip's,
//                  expression by expression and in its order.
//
// A first form of this kernel brought all six planes into LDS by LDS-DMA (buffer_load ... lds), three sets deep: bit-identical and no
// faster than lec_boxtile.hip -- 40 KB of LDS per wave leave one wave per SIMD, and a lone wave cannot hide its own hand-over and
// epilogue behind its arithmetic (git history, profiles/r06_notes.md section 3).
//
// Every expression that decides a bit is the one of lec_boxtile.hip (same products, same rounding points, same summation groups and
// order: groups of four columns, sixteen groups in order, end points in the epilogue), so the records are BIT-IDENTICAL to that
// kernel's and results still do not depend on how a series is sharded, chunked or which of the two kernels ran (tested).
//
// Serves per-step boxes on even longitudes with Phi present, cubes at most 64 columns wide: fp64 storage with dT/dt as a cube (what
// pack_series / the streamed -t path hand over for fp64 data), fp32 storage with dT/dt as a cube or with T of the two time neighbours
// as cubes of their own (tm_d / tp_d: a box-packed fp32 series -- the reference's float32 samples, ERA5's int16 + fill value).  An fp32
// lane's four points of a plane are 16 bytes: ONE load per plane.  Everything else of a per-step-box call (crops wider than 64
// columns, table longitudes, fp64 time neighbours: six 32-byte operands per prefetch set do not fit the registers) runs on lec_boxtile.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/lec_hip.h"
#include "lec_internal.h"
#include "lec_rowcommon.h"
#include "lec_sweep.h"

using namespace lec;

#ifndef LEC_BP_ABLATE       // measurement builds only (tools/build_variant.sh), bit mask: 1 = no arithmetic on the points, 2 = no global loads
#define LEC_BP_ABLATE 0
#endif
#ifndef LEC_BP_NT           // cache policy of the once-read planes (u, v, omega, Phi, dT/dt): 1 = nontemporal (measured: slower)
#define LEC_BP_NT 0
#endif

namespace {

constexpr int kWR = 4;                    // box rows per wave
constexpr int kMaxW = 64;                 // widest cube (columns) the T tile holds
constexpr int kSide = 16;                 // per-row side values (lec_boxtile.hip): 5 shifts, f of the first point, a..f and T u v of the last
constexpr int kLB = 4;                    // levels whose rows are finished together (16 lanes: 4 levels x 4 rows)
constexpr int kPS = 65;                   // stride between the statistics of the partial-sum array
constexpr int kMaxLevels = 42;            // a wave keeps its chunk's static-stability coefficients one per lane (3 per level) in two registers
constexpr int kMinLevels = 5;
constexpr int kTileT = (kWR + 2) * kMaxW;         // elements: T with its two halo rows, row pitch = the cube's nx (LDS is laid out for fp64 tiles)
constexpr int kHalf = kNA / 2;                    // the sums are handed over ten statistics at a time
constexpr int kLdsDoubles = 2 * kTileT + kHalf * kPS + kLB * kWR * (kNA + kSide);       // 15,952 B
static_assert(3 * kMaxLevels <= 128, "the level coefficients of a wave's chunk must fit one value per lane of two registers");

__device__ __forceinline__ double lane_value(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// 16 bytes at (run start + byte_off), as two doubles.  The resource of the load is the run itself (base = its first byte, num_records =
// its length) and the range check is per dword: what lies past the run reads as zero and is never touched.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ u32x4_t load16(__amdgpu_buffer_rsrc_t run, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(run, (int)byte_off, 0, NT ? 2 : 0);
}
// the lane's four consecutive elements of a plane: 32 bytes (two loads) in fp64 storage, 16 bytes (one load) in fp32
template <typename TIN> struct Quad;
template <> struct Quad<double> {
    u32x4_t h[2];
    template <bool NT> __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t run, unsigned byte_off) { h[0] = load16<NT>(run, byte_off); h[1] = load16<NT>(run, byte_off + 16u); }
    __device__ __forceinline__ void get(double (&o)[4]) const {
        o[0] = __hiloint2double((int)h[0].y, (int)h[0].x); o[1] = __hiloint2double((int)h[0].w, (int)h[0].z);
        o[2] = __hiloint2double((int)h[1].y, (int)h[1].x); o[3] = __hiloint2double((int)h[1].w, (int)h[1].z);
    }
    __device__ __forceinline__ void fill(double a, double b) {
        h[0].x = h[1].x = (unsigned)__double2loint(a); h[0].y = h[1].y = (unsigned)__double2hiint(a);
        h[0].z = h[1].z = (unsigned)__double2loint(b); h[0].w = h[1].w = (unsigned)__double2hiint(b);
    }
};
template <> struct Quad<float> {
    u32x4_t h;
    template <bool NT> __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t run, unsigned byte_off) { h = load16<NT>(run, byte_off); }
    __device__ __forceinline__ void get(double (&o)[4]) const {
        o[0] = (double)__uint_as_float(h.x); o[1] = (double)__uint_as_float(h.y); o[2] = (double)__uint_as_float(h.z); o[3] = (double)__uint_as_float(h.w);
    }
    __device__ __forceinline__ void fill(double a, double b) { h.x = h.z = __float_as_uint((float)a); h.y = h.w = __float_as_uint((float)b); }
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t run_resource(const void* first, unsigned bytes) {
    // (wave-uniform, and said so: a resource the compiler cannot prove uniform costs a waterfall loop around every load)
    const unsigned long long b = (unsigned long long)first;
    const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), (short)0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

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

// G: waves per workgroup = neighbouring row blocks of one (time step, level chunk).  The waves share nothing and never wait for each other;
// they run on ONE CU at about the same time, so the lines two neighbouring blocks both ask for -- a block's first and last row are its
// neighbours' halo rows, and a run rarely ends on a line boundary -- are asked for once: the second request finds the line in that CU's L1,
// or on its way there (the kernel is bound by the L1's outstanding misses: every merged request is a slot).
// TIN: storage type.  MODE 2: dT/dt is a cube (p.DT);  MODE 1: dT/dt = ta T(t-1) + tb T(t) + tc T(t+1) per point, T of the two time
// neighbours on this step's box in cubes of their own (p.TM, p.TP: a box-packed series).
template <typename TIN, int MODE, int G>
__global__ void __launch_bounds__(64 * G, 2) lec_boxplane_kernel(const RowParams p) {
    static_assert(MODE == 2 || (MODE == 1 && sizeof(TIN) == 4), "time neighbours as operands: fp32 storage (six 32-byte operands per set do not fit)");
    constexpr int kPiecesT = (kTileT * (int)sizeof(TIN) + 1023) / 1024;      // flat 1-KiB pieces of the T run: 3 (fp64) / 2 (fp32)
    constexpr int kTileStride = kPiecesT * 1024 / (int)sizeof(TIN);          // elements between the two tiles: whole pieces (a piece's lanes past the run write zeros)
    static_assert(2 * kTileStride * sizeof(TIN) <= 2 * kTileT * sizeof(double), "the two tiles must fit their LDS");
    constexpr int NX = MODE == 2 ? 1 : 2;                  // dT/dt operands per point
    constexpr unsigned ESZ = (unsigned)sizeof(TIN);
    __shared__ __attribute__((aligned(16))) double sm_all[G * kLdsDoubles];
    double* const sm = sm_all + (G > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0) * kLdsDoubles;
    TIN* const tiles = reinterpret_cast<TIN*>(sm);         // 2 x [6 rows][W]: T of this level and of the next, each as it lies in memory
    double* const part = sm + 2 * kTileT;                  // [kHalf][kPS]: the lanes' partial sums of a level, ten statistics at a time
    double* const stash = part + kHalf * kPS;              // [kLB levels][kWR rows][kNA]: row totals waiting for their finishing lane
    double* const side = stash + kLB * kWR * kNA;          // [kLB levels][kWR rows][kSide]
    const int lane = threadIdx.x & 63;

    // block -> (time step, level chunk, row block of 4): lec_boxtile.hip's order (every XCD a contiguous chunk of time steps, row
    // block fastest, then time step, then level chunk)
    const int n_rb = p.jrows, kchunk = p.jgroup, n_kc = (p.nl + kchunk - 1) / kchunk;
    const int xcd = blockIdx.x & 7;
    int q0 = blockIdx.x >> 3;
    const int rbi = (q0 % n_rb) * G + (G > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0); q0 /= n_rb;     // (n_rb: row-block GROUPS)
    const int ti = q0 % p.tgroup; q0 /= p.tgroup;
    const int kc = q0 % n_kc;
    const int tin = (q0 / n_kc) * p.tgroup + ti;
    const int tl = xcd * p.jchunk + tin;
    if (tin >= p.jchunk || tl >= p.t_count) return;

    const int bi = (p.n_box == 1) ? 0 : tl;
    // (wave-uniform, and said so: everything derived from the box -- the loads' resources above all -- must live in scalar registers)
    const int iw = __builtin_amdgcn_readfirstlane(p.box[4 * bi + 0]), ie = __builtin_amdgcn_readfirstlane(p.box[4 * bi + 1]);
    const int js = __builtin_amdgcn_readfirstlane(p.box[4 * bi + 2]), jn = __builtin_amdgcn_readfirstlane(p.box[4 * bi + 3]);
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

    const int W = p.nx;                   // the cube's row pitch = the T tile's row pitch (<= 64)
    const int t = p.t_begin + tl;
    const size_t plane = (size_t)p.ny * W;
    const size_t cube = plane * p.nl;
    auto lev = [&](int k) -> size_t { return (size_t)min(max(k, 0), p.nl - 1) * plane; };
    // the wave's rows: box rows jb0 .. jb0 + 3 (cut at the box's last row); T also one row either side where the box has one
    const int h0 = max(jb0 - 1, 0), h1 = min(jb0 + kWR + 1, nyb), r1 = min(jb0 + kWR, nyb);
    const unsigned bytesT = (unsigned)((h1 - h0) * W) * ESZ, bytesF = (unsigned)((r1 - jb0) * W) * ESZ;
    const TIN* const gT = (const TIN*)p.T + (size_t)t * cube + (size_t)(js + h0) * W;
    const size_t fbase = (size_t)t * cube + (size_t)(js + jb0) * W;
    const TIN* const gU = (const TIN*)p.U + fbase;
    const TIN* const gV = (const TIN*)p.V + fbase;
    const TIN* const gW = (const TIN*)p.W + fbase;
    const TIN* const gP = (const TIN*)p.P + fbase;
    const TIN* const gX0 = (const TIN*)(MODE == 2 ? p.DT : p.TM) + fbase;     // dT/dt itself, or T(t - 1) ...
    const TIN* const gX1 = (const TIN*)(MODE == 2 ? p.DT : p.TP) + fbase;     // ... and T(t + 1) on this step's box
    double ta = 0.0, tb = 0.0, tc = 0.0;
    if (MODE == 1) { const double* tcf = p.tcoef + (size_t)t * 3; ta = tcf[0]; tb = tcf[1]; tc = tcf[2]; }

    // ---- lane roles: (row ci of the wave's four, column group cg of sixteen), columns 4 cg + q.  Lanes 0..31 hold groups 0..7 of all
    // four rows, lanes 32..63 groups 8..15: the 32 lanes that an LDS read serves together then touch 32 different 8-byte banks
    // whenever the row pitch is odd (61 columns) -- with (row, group) = (lane / 16, lane % 16) groups g and g + 8 would collide
    const int ci = (lane >> 3) & 3, cg = (lane & 7) | ((lane >> 5) << 3);
    // (the hand-over array is indexed by LANE -- contiguous writes --, and the lane that adds row ci's sixteen partials walks that row's
    // lanes in group order: group g of row ci is lane (g & 7) + 8 ci + 32 (g >> 3); every LDS access of the hand-over is conflict-free)
    const int jb = jb0 + ci, jbc = min(jb, nyb - 1);       // (rows past the box's last one walk along on it and store nothing)
    // element offsets inside the T tile (doubles): its row 0 = box row h0
    const int rT = (jbc - h0) * W + iw, rTm = (max(jbc - 1, 0) - h0) * W + iw, rTp = (min(jbc + 1, nyb - 1) - h0) * W + iw;
    // the lane's 32 bytes of a plane's run (its row 0 = box row jb0): columns 4 cg .. 4 cg + 3 of row ci.  Groups past the cube's last
    // column, and rows past the run, lie outside the resource: zeros
    const unsigned offF = (unsigned)((jb - jb0) * W + iw + 4 * cg) * ESZ;
    int col[4];
    bool zero[4], first[4], last[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 4 * cg + q;
        first[q] = c == 0; last[q] = c == nxb - 1;
        zero[q] = c == 0 || c >= nxb - 1;                  // the trapezoid's end points (added in the epilogue with weight 1/2) and columns beside the box
        col[q] = min(c, nxb - 1);
    }
    // d/dlon of T at the lane's point q, centred; one-sided at the row ends (the point itself stands in for the missing neighbour, the
    // difference counts twice: lec_boxtile.hip's ((last ? T : Tr) - (first ? T : Tl)) * ((first || last) ? 2 : 1)).  The neighbours'
    // columns are formed where they are used: eight lane constants fewer in registers that the two prefetch sets need
    auto ddlon = [&](const TIN* tile, int q) -> double {
#pragma clang fp contract(off)
        int c = col[q];
        asm volatile("" : "+v"(c));
        const double d = (double)tile[rT + min(c + 1, nxb - 1)] - (double)tile[rT + max(c - 1, 0)];
        return (first[q] || last[q]) ? d * 2.0 : d;
    };
    // coefficients of the lane's row (d/dlat a, b, c; 1 / dx) and of the wave's levels (one per lane: picked with v_readlane, so that
    // no table load -- a vector load the compiler would wait for with vmcnt(0), draining the prefetch -- sits inside the level loop)
    const double* lt = p.lattab + ((size_t)bi * p.nyb_max + jbc) * 4;
    const double ga = lt[0], gb = lt[1], gc = lt[2], idx = lt[3];
    const double levv = p.levtab[(size_t)min(k0 + lane / 3, p.nl - 1) * 3 + lane % 3];                 // levels k0 .. k0 + 20 (lane 63: level 21's first)
    const double levw = p.levtab[(size_t)min(k0 + (lane + 64) / 3, p.nl - 1) * 3 + (lane + 64) % 3];     // ... and on from there
    const double inv_xlen = p.boxtab[4 * bi + 0], h_rad = p.boxtab[4 * bi + 1], inv_hdeg = p.boxtab[4 * bi + 2];
    const double cx = (0.5 * inv_hdeg) * idx;

    // ---- the loads of one level: the T run as three flat pieces; the lane's 32 bytes of the five other planes into one of TWO register
    // sets, picked by the level's parity at compile time -- the planes of level k + 1 are requested BEFORE level k's are waited for, so a
    // wave has two levels in flight while it waits (as the bare load stream of the probe has) and one while it computes
    u32x4_t tT[kPiecesT];
    Quad<TIN> fU[2], fV[2], fW[2], fP[2], fX[2][NX];
    auto issue_T = [&](int kT) {
        if (LEC_BP_ABLATE & 2) return;
        const __amdgpu_buffer_rsrc_t rt = run_resource(gT + lev(kT), bytesT);
#pragma unroll
        for (int i = 0; i < kPiecesT; ++i) tT[i] = load16<false>(rt, 1024u * i + 16u * lane);      // (halo rows are the neighbouring wave's own rows: default policy)
    };
    auto issue_F = [&](auto set_tag, int kF) {
        constexpr int SET = decltype(set_tag)::value;
        if (LEC_BP_ABLATE & 2) return;
        const size_t lk = lev(kF);
        fU[SET].template load<LEC_BP_NT != 0>(run_resource(gU + lk, bytesF), offF);
        fV[SET].template load<LEC_BP_NT != 0>(run_resource(gV + lk, bytesF), offF);
        fW[SET].template load<LEC_BP_NT != 0>(run_resource(gW + lk, bytesF), offF);
        fP[SET].template load<LEC_BP_NT != 0>(run_resource(gP + lk, bytesF), offF);
        fX[SET][0].template load<LEC_BP_NT != 0>(run_resource(gX0 + lk, bytesF), offF);
        if (NX > 1) fX[SET][NX - 1].template load<LEC_BP_NT != 0>(run_resource(gX1 + lk, bytesF), offF);
    };
    auto tile_in = [&](TIN* tile, const u32x4_t (&xT)[kPiecesT]) {     // a T run into a tile, as it lies in memory (the tile's readers are done: one wave, LDS in order)
        u32x4_t* dst = reinterpret_cast<u32x4_t*>(tile);
#pragma unroll
        for (int i = 0; i < kPiecesT; ++i) dst[64 * i + lane] = xT[i];
    };
    if (LEC_BP_ABLATE & 2) {
        Quad<TIN> z; z.fill(280.0 + lane, 281.0);
#pragma unroll
        for (int i = 0; i < kPiecesT; ++i) tT[i] = sizeof(TIN) == 8 ? reinterpret_cast<const Quad<double>&>(z).h[0] : reinterpret_cast<const Quad<float>&>(z).h;
#pragma unroll
        for (int w = 0; w < 2; ++w) { fU[w].fill(lane, 1.0); fV[w].fill(lane, 2.0); fW[w].fill(0.1, 0.2); fP[w].fill(lane, 3.0); fX[w][0].fill(0.1, 0.2); fX[w][NX - 1].fill(0.1, 0.2); }
    }

    // ---- prologue: T(k0)'s tile, the lane's points of T(k0 - 1); the loads of level k0 and T(k0 + 1)'s run on their way
    double Tm[4];
    {
        const __amdgpu_buffer_rsrc_t rt = run_resource(gT + lev(k0), bytesT);
        u32x4_t x[kPiecesT];
#pragma unroll
        for (int i = 0; i < kPiecesT; ++i) x[i] = load16<false>(rt, 1024u * i + 16u * lane);
        const TIN* g = (const TIN*)p.T + (size_t)t * cube + lev(k0 - 1) + (size_t)(js + jbc) * W + iw;
#pragma unroll
        for (int q = 0; q < 4; ++q) Tm[q] = (double)g[col[q]];
        issue_F(std::integral_constant<int, 0>{}, k0);
        issue_T(k0 + 1);
        tile_in(tiles, x);
    }

    double acc[kNA];
#pragma unroll
    for (int s = 0; s < kNA; ++s) acc[s] = 0.0;

    auto level = [&](auto set_tag, const int k) {
        constexpr int SET = decltype(set_tag)::value;
        const int kk = k - k0, slot = kk % kLB;
        double* const sd = side + (slot * kWR + ci) * kSide;
        // the next level's five planes into the other register set, then T(k + 1)'s run into the other tile and T(k + 2)'s run requested
        // (levels and tiles alternate together: SET is also the tile that holds T(k))
        const TIN* const tileC = tiles + SET * kTileStride;          // T(k): the lane's points and their four horizontal neighbours
        TIN* const tileN = tiles + (SET ^ 1) * kTileStride;          // T(k + 1): the lane's points
        if (k + 1 < k1) issue_F(std::integral_constant<int, SET ^ 1>{}, k + 1);
        tile_in(tileN, tT);
        if (k + 1 < k1) issue_T(k + 2);
        double Uq[4], Vq[4], Wq[4], Pq[4], X0q[4], X1q[4];
        fU[SET].get(Uq); fV[SET].get(Vq); fW[SET].get(Wq); fP[SET].get(Pq); fX[SET][0].get(X0q); fX[SET][NX - 1].get(X1q);
        // the shifts of the other four planes: the row's first box element, which the lane of group 0 holds -- through the side array
        if (first[0]) { sd[0] = (double)tileC[rT]; sd[1] = Uq[0]; sd[2] = Vq[0]; sd[3] = Wq[0]; sd[4] = Pq[0]; }
        row_sync<64>();
        if (!(LEC_BP_ABLATE & 1)) {
#pragma clang fp contract(off)
            auto levc = [&](int e) -> double { return e < 64 ? lane_value(levv, e) : lane_value(levw, e - 64); };      // (wave-uniform choice)
            const double al = levc(3 * kk), be = levc(3 * kk + 1), gm = levc(3 * kk + 2);
            const double cT = (double)tileC[rT], cU = sd[1], cV = sd[2], cW = sd[3], cP = sd[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double T = (double)tileC[rT + col[q]], Tn = (double)tileN[rT + col[q]];     // the lane's point at this level and one level down
                const double U = Uq[q], V = Vq[q], Wv = Wq[q], P = Pq[q];
                const double D = (MODE == 1) ? stencil3(ta, X0q[q], tc, X1q[q], tb, T) : X0q[q];     // dT/dt: lec_boxtile.hip's expression
                const double sS = stencil3(al, Tm[q], gm, Tn, be, T);
                const double adv = (U * cx) * ddlon(tileC, q);
                const double sP_ = stencil3(ga, (double)tileC[rTm + col[q]], gc, (double)tileC[rTp + col[q]], gb, T);
                const double f = fma(-Wv, sS, fma(V, sP_, D + adv));
                const double a = T - cT, b = U - cU, c = V - cV, d = Wv - cW, ee = P - cP;
                if (!zero[q]) accum20<true>(acc, 1.0, a, b, c, d, ee, f);
                if (first[q]) sd[5] = f;
                if (last[q]) {
                    sd[6] = a; sd[7] = b; sd[8] = c; sd[9] = d; sd[10] = ee; sd[11] = f;
                    sd[12] = T; sd[13] = U; sd[14] = V;
                }
                Tm[q] = T;                                                        // (the level above, next time: the one value a lane carries)
                // one point at a time: interleaved, the four points' temporaries push the kernel past its 256 registers (two prefetch sets live)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            // the level's rows are complete: 16 partial sums per row and statistic -> one total, through LDS, ten statistics at a time: every
            // lane stores its partials, then lane (row, s) adds the 16 of its row in a fixed order (lec_boxtile.hip's hand-over and order)
#pragma clang fp contract(off)
            double* st = stash + (slot * kWR + ci) * kNA;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                row_sync<64>();
#pragma unroll
                for (int s = 0; s < kHalf; ++s) { part[s * kPS + lane] = acc[kHalf * h + s]; acc[kHalf * h + s] = 0.0; }
                row_sync<64>();
                const double* p0 = part + min(cg, kHalf - 1) * kPS + 8 * ci;       // statistic 10 h + cg of row ci (cg < 10)
                double t0 = p0[0];
#pragma unroll
                for (int g = 1; g < 16; ++g) t0 += p0[(g & 7) + 32 * (g >> 3)];
                if (cg < kHalf) st[kHalf * h + cg] = t0;
            }
            row_sync<64>();
        }
        // ---- up to kLB finished levels x 4 rows: one (level, row) per lane finishes its record (lec_boxtile.hip's epilogue, expression by
        // expression) -- into LDS (the hand-over array is free); then ALL lanes store the records, sixteen lanes per 256-byte record:
        // four store instructions that write whole lines.  (One record per lane straight to memory is 13 store instructions of sixteen
        // 16-byte pieces in sixteen different lines each: as many partial-line write requests as the kernel has read requests.)
        if (slot == kLB - 1 || k == k1 - 1) {
#pragma clang fp contract(off)
            constexpr int kRS = LEC_NSTAT + 2;           // record stride in the buffer (doubles): 16-byte aligned, and eight lanes' 16-byte writes fall on different banks
            static_assert(kLB * kWR * kRS <= kHalf * kPS, "the finished records must fit the hand-over array");
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int lv = ln >> 2, r = ln & 3;
            if (lv <= slot) {
                dbl2_t* __restrict__ out = reinterpret_cast<dbl2_t*>(part + (lv * kWR + r) * kRS);
                if (jb0 + r >= nyb) {            // padding row of a box lower than nyb_max
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
#pragma unroll
            for (int i = 0; i < kLB * kWR * (LEC_NSTAT / 2) / 64; ++i) {
                const int e = 64 * i + lane, rec = e >> 4, piece = e & 15;      // record (level slot, row) and the lane's 16 bytes of it
                const int rl = rec >> 2, rr = rec & 3;
                if (rl <= slot && jb0 + rr < p.nyb_max) {
                    const dbl2_t v2 = *reinterpret_cast<const dbl2_t*>(part + rec * kRS + 2 * piece);
                    reinterpret_cast<dbl2_t*>(p.rows + ((size_t)(tl * p.nl + (k - slot + rl)) * p.nyb_max + jb0 + rr) * LEC_NSTAT)[piece] = v2;
                }
            }
            row_sync<64>();
        }
    };
    for (int k = k0; k < k1; k += 2) {
        level(std::integral_constant<int, 0>{}, k);
        if (k + 1 < k1) level(std::integral_constant<int, 1>{}, k + 1);
    }
}

}  // namespace

// true where lec_boxplane_kernel serves the call (the KIND of call and the slabs' shape -- the same for every shard and chunk of a
// series; never the boxes' extents or the step count)
bool lec_boxplane_serves(const lec::RowParams& p, int dtype, bool uniform, int mode) {
    if (!uniform || !p.P || p.n_box == 1 || p.nx > kMaxW || p.nxb_max > kMaxW) return false;
    if (mode == 2) return p.DT && !p.TM && !p.TP;                          // dT/dt as a cube: either storage type
    return mode == 1 && dtype == LEC_F32 && p.TM && p.TP;                  // time neighbours as cubes of their own: a box-packed fp32 series
}

// p.tgroup: time steps per tile group, p.jgroup: levels per wave (< 1: chosen here; more than 21: LEC_ERR_ARG) -- as lec_launch_boxtile
int lec_launch_boxplane(lec::RowParams p, int dtype, int mode, hipStream_t st) {
#ifndef LEC_BP_GROUP      // measured (profiles/r06_notes.md): 2 and 4 waves per workgroup 0.76-0.78 ms per 512 steps against 0.757 for 1 -- no gain
#define LEC_BP_GROUP 1
#endif
    constexpr int G = LEC_BP_GROUP;                       // waves (neighbouring row blocks) per workgroup
    const long long n_rb = ((p.nyb_max + kWR - 1) / kWR + G - 1) / G;      // row-block groups
    p.jrows = (int)n_rb;
    p.jchunk = (int)((p.t_count + 7) / 8);                // time steps per XCD
    if (p.jgroup > kMaxLevels) return LEC_ERR_ARG;
    if (p.jgroup < 1) {
        // levels per wave: as many as still leave kTargetWaves waves (the chip holds 2048 of these at eight per CU: sixteen rounds) -- a
        // chunk's first level costs a T tile and a pipeline fill of its own, but short walks run better at every series length -- never
        // more than kWalk levels unless asked (512 steps: 10 levels per wave 0.743 ms, 19: 0.750, 37: 0.754; 2048 steps: 2.77 / 2.78 /
        // 2.87 ms; profiles/r06_notes.md)
        constexpr long long kTargetWaves = 32768, kWalk = 10;
        const long long per_chunk = 8LL * p.jchunk * n_rb * G;
        const long long want = (kTargetWaves + per_chunk - 1) / per_chunk;
        const long long most = (p.nl + kMinLevels - 1) / kMinLevels, least = (p.nl + kWalk - 1) / kWalk;
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
    const dim3 grid((unsigned)nblocks), block(64 * G);
    if (dtype == LEC_F64) hipLaunchKernelGGL((lec_boxplane_kernel<double, 2, G>), grid, block, 0, st, p);
    else if (mode == 2) hipLaunchKernelGGL((lec_boxplane_kernel<float, 2, G>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((lec_boxplane_kernel<float, 1, G>), grid, block, 0, st, p);
    return LEC_OK;
}
