// probe_boxdma.hip -- what the memory system delivers for the BOX-PACKED moving series ([T][37][61][61] fp64, six planes per step:
// T, u, v, omega, Phi, dT/dt), by HOW a wave asks for it.  (measurement tool, not product)
// A wave owns RB box rows of one time step and walks KC levels; per level it needs RB + 2 rows of T (one halo row either side) and
// RB rows of the five other planes -- in the packed layout each of those is ONE contiguous run of bytes.  Three request shapes:
//   MODE 0  today's mapping of lec_boxtile_kernel: one 488-byte row per wave instruction, 8 B per lane, into registers
//   MODE 1  flat: the contiguous run in 1-KiB pieces, 16 B per lane, into registers
//   MODE 2  flat, LDS-DMA: the same pieces by global_load_lds_dwordx4 straight into LDS (no registers, no ds_write), double-buffered
//           per level, waited for with a counted vmcnt
// The loads of level k + 1 are in flight while level k is "consumed" (a few adds / LDS reads: the probe measures the request path,
// not arithmetic).  Resident waves per CU are set by the dynamic LDS size.
// Build: hipcc -O3 --offload-arch=gfx950 probe_boxdma.hip -o probe_boxdma ; run: ./probe_boxdma [T]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NL = 37, NB = 61, NF = 6;
constexpr size_t kPlane = (size_t)NB * NB;           // doubles
struct P { const double* f[NF]; double* out; int T; };

__device__ __forceinline__ void dma16(const void* gbase, unsigned voff, unsigned lds_addr) {
    const unsigned long long gb = (unsigned long long)gbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)gb), hi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
    const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_addr), "v"(voff), "s"(sb) : "memory");
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t run_resource(const void* first_byte, unsigned bytes) {
    const unsigned long long b = (unsigned long long)first_byte;
    u32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xffffu;
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
template <bool NTB>
__device__ __forceinline__ void bdma16(u32x4_t run, unsigned voff, unsigned soff, unsigned lds_addr) {
    if (NTB) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" :: "s"(lds_addr), "v"(voff), "s"(run), "s"(soff) : "memory");
    else     asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(run), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int RB> constexpr int pieces_T() { return ((RB + 2) * NB * 8 + 1023) / 1024; }
template <int RB> constexpr int pieces_F() { return (RB * NB * 8 + 1023) / 1024; }
template <int RB> constexpr int pieces_level() { return pieces_T<RB>() + 5 * pieces_F<RB>(); }

typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));

template <int MODE, int RB, int KC, bool NT>
__global__ void __launch_bounds__(64) probe(const P p) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tchunk = (p.T + 7) / 8, nrb = (NB + RB - 1) / RB, nkc = (NL + KC - 1) / KC;
    const int rbi = q % nrb, tt = (q / nrb) % tchunk, kc = q / (nrb * tchunk);
    const int tl = xcd * tchunk + tt;
    if (tl >= p.T) return;
    const int k0 = kc * KC, nk = min(KC, NL - k0);
    const int r0 = rbi * RB, r1 = min(r0 + RB, NB);              // the wave's rows
    const int h0 = max(r0 - 1, 0), h1 = min(r1 + 1, NB);         // ... with the T halo
    double acc = 0.0;
    if (MODE == 0) {
        constexpr int NR = (RB + 2) + 5 * RB;                    // row loads per level
        double a[NR], b[NR];
        auto issue = [&](double (&d)[NR], int k) {
            const size_t pl = ((size_t)tl * NL + k) * kPlane;
            const int col = min(lane, NB - 1);
#pragma unroll
            for (int i = 0; i < RB + 2; ++i) { const double* g = p.f[0] + pl + (size_t)min(h0 + i, h1 - 1) * NB + col; d[i] = NT ? __builtin_nontemporal_load(g) : *g; }
#pragma unroll
            for (int f = 1; f < NF; ++f)
#pragma unroll
                for (int i = 0; i < RB; ++i) { const double* g = p.f[f] + pl + (size_t)min(r0 + i, r1 - 1) * NB + col; d[RB + 2 + (f - 1) * RB + i] = NT ? __builtin_nontemporal_load(g) : *g; }
        };
        issue(a, k0);
        for (int kk = 0; kk < nk; kk += 2) {
            issue(b, min(k0 + kk + 1, k0 + nk - 1));
#pragma unroll
            for (int i = 0; i < NR; ++i) acc += a[i];
            issue(a, min(k0 + kk + 2, k0 + nk - 1));
#pragma unroll
            for (int i = 0; i < NR; ++i) acc += b[i];
        }
    } else if (MODE == 4) {
        static_assert(MODE != 4 || RB == 4, "the compute layout is 4 rows x 16 column groups");
        constexpr int kT = pieces_T<RB>() * 1024;
        const unsigned lds0 = (unsigned)(uintptr_t)sm;
        const int ci = lane >> 4, cg = lane & 15;
        const int rowc = min(r0 + ci, NB - 1), c0 = min(4 * cg, NB - 4);        // (the last group reads columns 57..60)
        dbl2 a[10], b[10];
        auto issue_T = [&](int buf, int k) {
            const size_t pl = ((size_t)tl * NL + k) * kPlane;
            const u32x4_t rt = run_resource(p.f[0] + pl + (size_t)h0 * NB, (unsigned)((h1 - h0) * NB * 8));
#pragma unroll
            for (int i = 0; i < pieces_T<RB>(); ++i) bdma16<false>(rt, 16u * lane, 1024u * i, lds0 + buf * kT + 1024u * i);
        };
        auto issue_F = [&](dbl2 (&d)[10], int k) {
            const size_t pl = ((size_t)tl * NL + k) * kPlane + (size_t)rowc * NB + c0;
#pragma unroll
            for (int f = 1; f < NF; ++f) {
                const dbl2* g = reinterpret_cast<const dbl2*>(p.f[f] + pl);
                d[2 * (f - 1)] = NT ? __builtin_nontemporal_load(g) : *g;
                d[2 * (f - 1) + 1] = NT ? __builtin_nontemporal_load(g + 1) : g[1];
            }
        };
        // vmcnt bookkeeping is ours for the DMA pieces; the register loads are the compiler's: keep the two apart by draining the
        // DMA of a level before the compiler-visible loads of the next are consumed (in order: DMA(k+1), loads(k+1) issued together)
        issue_T(0, k0); issue_F(a, k0);
        for (int kk = 0; kk < nk; kk += 2) {
            issue_T(1, min(k0 + kk + 1, k0 + nk - 1)); issue_F(b, min(k0 + kk + 1, k0 + nk - 1));
            {
                const double* s = sm;
#pragma unroll
                for (int i = 0; i < 10; ++i) acc += a[i].x + a[i].y;
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(pieces_T<RB>() + 10) : "memory");
#pragma unroll
                for (int i = 0; i < 6; ++i) acc += s[i * 64 + lane];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            issue_T(0, min(k0 + kk + 2, k0 + nk - 1)); issue_F(a, min(k0 + kk + 2, k0 + nk - 1));
            {
                const double* s = sm + kT / 8;
#pragma unroll
                for (int i = 0; i < 10; ++i) acc += b[i].x + b[i].y;
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(pieces_T<RB>() + 10) : "memory");
#pragma unroll
                for (int i = 0; i < 6; ++i) acc += s[i * 64 + lane];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
        wait_vm<0>();
    } else if (MODE == 1) {
        constexpr int NP = pieces_level<RB>();
        dbl2 a[NP], b[NP];
        auto issue = [&](dbl2 (&d)[NP], int k) {
            const size_t pl = ((size_t)tl * NL + k) * kPlane;
            const int nT = (h1 - h0) * NB, nF = (r1 - r0) * NB;     // doubles
#pragma unroll
            for (int i = 0; i < pieces_T<RB>(); ++i) {
                const int e = min(128 * i + 2 * lane, nT - 2);
                const dbl2* g = reinterpret_cast<const dbl2*>(p.f[0] + pl + (size_t)h0 * NB + e);
                d[i] = NT ? __builtin_nontemporal_load(g) : *g;
            }
#pragma unroll
            for (int f = 1; f < NF; ++f)
#pragma unroll
                for (int i = 0; i < pieces_F<RB>(); ++i) {
                    const int e = min(128 * i + 2 * lane, nF - 2);
                    const dbl2* g = reinterpret_cast<const dbl2*>(p.f[f] + pl + (size_t)r0 * NB + e);
                    d[pieces_T<RB>() + (f - 1) * pieces_F<RB>() + i] = NT ? __builtin_nontemporal_load(g) : *g;
                }
        };
        issue(a, k0);
        for (int kk = 0; kk < nk; kk += 2) {
            issue(b, min(k0 + kk + 1, k0 + nk - 1));
#pragma unroll
            for (int i = 0; i < NP; ++i) acc += a[i].x + a[i].y;
            issue(a, min(k0 + kk + 2, k0 + nk - 1));
#pragma unroll
            for (int i = 0; i < NP; ++i) acc += b[i].x + b[i].y;
        }
    } else {
        constexpr int NP = pieces_level<RB>();
        constexpr int kBuf = NP * 1024;                           // bytes per buffer (pieces land back to back)
        const unsigned lds0 = (unsigned)(uintptr_t)sm;
        auto issue = [&](int buf, int k) {
            const size_t pl = ((size_t)tl * NL + k) * kPlane;
            const int bT = (h1 - h0) * NB * 8, bF = (r1 - r0) * NB * 8;      // bytes
            unsigned dst = lds0 + buf * kBuf;
            if (MODE == 3) {        // lec_boxplane.hip's form: buffer loads whose resource is the run (no clamping; T default policy, the rest nt)
                const u32x4_t rt = run_resource(p.f[0] + pl + (size_t)h0 * NB, (unsigned)bT);
#pragma unroll
                for (int i = 0; i < pieces_T<RB>(); ++i) { bdma16<false>(rt, 16u * lane, 1024u * i, dst); dst += 1024; }
#pragma unroll
                for (int f = 1; f < NF; ++f) {
                    const u32x4_t rf = run_resource(p.f[f] + pl + (size_t)r0 * NB, (unsigned)bF);
#pragma unroll
                    for (int i = 0; i < pieces_F<RB>(); ++i) { bdma16<NT>(rf, 16u * lane, 1024u * i, dst); dst += 1024; }
                }
                return;
            }
            const char* gT = reinterpret_cast<const char*>(p.f[0] + pl + (size_t)h0 * NB);
#pragma unroll
            for (int i = 0; i < pieces_T<RB>(); ++i) {
                const unsigned off = min(1024u * i + 16u * lane, (unsigned)bT - 16u);      // every piece is issued by every lane: the vmcnt arithmetic needs a fixed count
                dma16(gT, off, dst);
                dst += 1024;
            }
#pragma unroll
            for (int f = 1; f < NF; ++f) {
                const char* g = reinterpret_cast<const char*>(p.f[f] + pl + (size_t)r0 * NB);
#pragma unroll
                for (int i = 0; i < pieces_F<RB>(); ++i) {
                    const unsigned off = min(1024u * i + 16u * lane, (unsigned)bF - 16u);
                    dma16(g, off, dst);
                    dst += 1024;
                }
            }
        };
        issue(0, k0);
        for (int kk = 0; kk < nk; ++kk) {
            const int buf = kk & 1;
            issue(buf ^ 1, min(k0 + kk + 1, k0 + nk - 1));
            wait_vm<NP>();                                        // level kk has landed; level kk + 1 stays in flight
            const double* s = sm + buf * (kBuf / 8);
#pragma unroll
            for (int i = 0; i < NP; ++i) acc += s[i * 128 + lane] + s[i * 128 + 64 + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        wait_vm<0>();
    }
    p.out[(size_t)blockIdx.x * 64 + lane] = acc;
}

template <int MODE, int RB, int KC, bool NT = true>
void run(const P& p, int waves_per_cu) {
    const int tchunk = (p.T + 7) / 8, nrb = (NB + RB - 1) / RB, nkc = (NL + KC - 1) / KC;
    dim3 grid(8 * tchunk * nrb * nkc), block(64);
    size_t lds = (size_t)(160 * 1024 / waves_per_cu) - 512;
    const size_t need = MODE == 4 ? (size_t)2 * pieces_T<RB>() * 1024 : MODE >= 2 ? (size_t)2 * pieces_level<RB>() * 1024 : 8;
    if (lds < need) { printf("  mode %d rows %2d: %d waves per CU do not fit (%zu B of LDS per wave needed)\n", MODE, RB, waves_per_cu, need); return; }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, RB, KC, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<MODE, RB, KC, NT>), grid, block, lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<MODE, RB, KC, NT>), grid, block, lds, 0, p);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double gb6 = (double)p.T * NL * NB * NB * 8 * 6 / 1e9, gb5 = gb6 * 5 / 6;
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&probe<MODE, RB, KC, NT>)));
    static const char* names[] = {"rows 8 B/lane -> regs", "flat 16 B/lane -> regs", "flat 16 B/lane LDS-DMA", "flat LDS-DMA, buffer loads", "T by DMA, rest -> regs (compute layout)"};
    printf("  %-24s %s rows %2d x levels %2d  waves/CU %2d [%3d VGPRs]: %.3f ms  %5.0f GB/s of the six planes  (algorithmic five: %.3f of 8 TB/s)\n",
           names[MODE], NT ? "nt   " : "plain", RB, KC, waves_per_cu, fa.numRegs, ms, gb6 / ms * 1e3, gb5 / ms * 1e3 / 8000.0);
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 512;
    const size_t n = (size_t)T * NL * kPlane + 64;
    P p; p.T = T;
    for (int f = 0; f < NF; ++f) { double* d; CK(hipMalloc(&d, n * 8)); CK(hipMemset(d, 0, n * 8)); p.f[f] = d; }
    double* out; CK(hipMalloc(&out, (size_t)8 * ((T + 7) / 8) * 61 * 37 * 64 * 8)); p.out = out;
    printf("T=%d box-packed 61 x 61 x 37, six planes: %.3f GB per pass\n", T, (double)T * NL * NB * NB * 48 / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
        printf("-- today's request shape\n");
        run<0, 4, 19>(p, 8); run<0, 4, 19>(p, 16);
        printf("-- flat pieces into registers\n");
        run<1, 4, 19>(p, 8); run<1, 4, 19>(p, 16); run<1, 8, 19>(p, 8); run<1, 16, 19>(p, 8);
        printf("-- flat pieces by LDS-DMA\n");
        run<2, 4, 19>(p, 4); run<2, 4, 19>(p, 5); run<2, 4, 19>(p, 6);
        run<2, 8, 19>(p, 3); run<2, 8, 10>(p, 3); run<2, 16, 19>(p, 1); run<2, 4, 19, false>(p, 5); run<2, 4, 37>(p, 5); run<2, 4, 10>(p, 5);
        printf("-- the same by buffer loads (the run as the resource: range-checked, nothing clamped)\n");
        run<3, 4, 19>(p, 4); run<3, 4, 19>(p, 5); run<3, 4, 19, false>(p, 5); run<3, 4, 10>(p, 5); run<3, 8, 19>(p, 3);
        printf("-- T by DMA, u v omega Phi dT/dt straight into the compute layout's registers (32 bytes per lane and plane)\n");
        run<4, 4, 19>(p, 6); run<4, 4, 19>(p, 8); run<4, 4, 19, false>(p, 8); run<4, 4, 19>(p, 12); run<4, 4, 10>(p, 8);
    }
    return 0;
}
