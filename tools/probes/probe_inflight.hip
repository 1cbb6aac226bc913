// probe_inflight.hip -- how does the delivered bandwidth of the moving-box access pattern depend on the bytes a CU keeps in
// flight?  (measurement tool, not product)
// Same cubes, boxes and block -> (row block, time step, level chunk) mapping as lec_boxtile_kernel: one wave per workgroup, four
// box rows x ten levels per wave, seven 488-byte row loads per (level, row) (five fields + T at t-1 / t+1).  The wave keeps two
// register batches of BATCH row loads (double-buffered: BATCH .. 2 BATCH loads in flight) and only adds what arrives; resident
// waves per CU are set by a dummy LDS allocation.  Output: requested bytes / time for each (waves per CU, BATCH).
// Build: hipcc -O3 --offload-arch=gfx950 probe_inflight.hip -o probe_inflight ; run: ./probe_inflight [T]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NL = 37, NY = 162, NX = 243, NB = 61, NF = 7;
struct P { const double* f[5]; const int* box; double* out; int T; int tshift; };

// load number n of this wave's sequence: (level, row, field) with field fastest
template <int WR, int KC, bool NT>
__device__ __forceinline__ double ld(const P& p, int n, int nmax, int tl, int k0, int j0, int iw, int js, int lane) {
    n = min(n, nmax - 1);
    const int f = n % NF, r = (n / NF) % WR, k = k0 + n / (NF * WR);
    const int row = js + min(j0 + r, NB - 1), col = iw + min(lane, NB - 1);
    const int tt = (f == 5) ? max(tl - p.tshift, 0) : (f == 6) ? min(tl + p.tshift, p.T - 1) : tl;
    const double* base = p.f[f < 5 ? f : 0];
    const double* q = base + (((size_t)tt * NL + k) * NY + row) * NX + col;
    return NT ? __builtin_nontemporal_load(q) : *q;
}

template <int BATCH, int WR, int KC, bool NT, int WT = 1, bool SYNC = false>
__global__ void __launch_bounds__(64 * WT) probe(const P p) {
    extern __shared__ double pad[];
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tchunk = (p.T + 7) / 8, nrb = (NB + WR - 1) / WR, nkc = (NL + KC - 1) / KC;
    const int rbi = q % nrb, kc = (q / nrb) % nkc, tt = q / (nrb * nkc);
    const int tl = xcd * tchunk + tt * WT + wv;
    if (tt * WT >= tchunk) return;                         // whole workgroup
    const bool live = tt * WT + wv < tchunk && tl < p.T;
    if (!live && !SYNC) return;
    const int tls = live ? tl : xcd * tchunk;
    const int iw = p.box[2 * tls], js = p.box[2 * tls + 1];
    const int k0 = kc * KC, nk = min(KC, NL - k0), nmax = nk * WR * NF;
    double a[BATCH], b[BATCH], acc = 0.0;
#pragma unroll
    for (int i = 0; i < BATCH; ++i) a[i] = ld<WR, KC, NT>(p, i, nmax, tls, k0, rbi * WR, iw, js, lane);
    for (int n = BATCH; n < nmax + BATCH; n += 2 * BATCH) {
        if (SYNC) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < BATCH; ++i) b[i] = ld<WR, KC, NT>(p, n + i, nmax, tls, k0, rbi * WR, iw, js, lane);
#pragma unroll
        for (int i = 0; i < BATCH; ++i) acc += a[i];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) a[i] = ld<WR, KC, NT>(p, n + BATCH + i, nmax, tls, k0, rbi * WR, iw, js, lane);
#pragma unroll
        for (int i = 0; i < BATCH; ++i) acc += b[i];
    }
    if (lane == 0) pad[wv] = acc;
    p.out[((size_t)blockIdx.x * WT + wv) * 64 + lane] = acc + pad[wv];
}

template <int BATCH, int WR = 4, int KC = 10, bool NT = true, int WT = 1, bool SYNC = false>
void run(const P& p, int waves_per_cu) {
    const int tchunk = (p.T + 7) / 8, nrb = (NB + WR - 1) / WR, nkc = (NL + KC - 1) / KC;
    dim3 grid(8 * ((tchunk + WT - 1) / WT) * nrb * nkc), block(64 * WT);
    const size_t lds = (size_t)(160 * 1024 / waves_per_cu) * WT - 512;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<BATCH, WR, KC, NT, WT, SYNC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<BATCH, WR, KC, NT, WT, SYNC>), grid, block, lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<BATCH, WR, KC, NT, WT, SYNC>), grid, block, lds, 0, p);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    // requested bytes: the loads beyond a wave's sequence are clamped repeats (L1 hits): count the sequence only, rows clamped to the box
    const double gb = (double)p.T * NL * (nrb * WR) * NB * 8 * NF / 1e9;
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&probe<BATCH, WR, KC, NT, WT, SYNC>)));
    printf("  [%3d VGPRs] %s rows %2d x levels %2d, %d steps per workgroup%s", fa.numRegs, NT ? "nt   " : "plain", WR, KC, WT, SYNC ? " (barrier per batch)" : "");
    printf("  waves/CU %2d  batch %2d (%3d..%3d loads in flight per wave, %3.0f..%3.0f KB per CU): %.3f ms  %.0f GB/s requested\n", waves_per_cu, BATCH, BATCH,
           2 * BATCH, waves_per_cu * BATCH * 0.488, waves_per_cu * 2 * BATCH * 0.488, ms, gb / ms * 1e3);
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 256;
    const size_t n = (size_t)T * NL * NY * NX;
    P p; p.T = T; p.tshift = 1;
    for (int f = 0; f < 5; ++f) { double* d; CK(hipMalloc(&d, n * 8)); CK(hipMemset(d, 0, n * 8)); p.f[f] = d; }
    std::vector<int> box(2 * T);
    for (int t = 0; t < T; ++t) {
        const double clat = -37.5 + 12.0 * sin(2 * M_PI * t / 400.0), clon = -50.0 + 22.0 * cos(2 * M_PI * t / 700.0);
        box[2 * t] = (int)lround((clon - 7.5 + 80.25) / 0.25); box[2 * t + 1] = (int)lround((clat - 7.5 + 57.75) / 0.25);
    }
    int* db; CK(hipMalloc(&db, box.size() * 4)); CK(hipMemcpy(db, box.data(), box.size() * 4, hipMemcpyHostToDevice)); p.box = db;
    const int tchunk = (T + 7) / 8;
    double* out; CK(hipMalloc(&out, (size_t)8 * tchunk * 16 * 37 * 64 * 8)); p.out = out;
    printf("T=%d: box rows of 488 B, %d row loads per (level, row)\n", T, NF);
    if (argc > 2) {
        for (int w : {4, 8, 12, 16, 24, 32}) {
            run<7>(p, w); run<14>(p, w); run<28>(p, w);
            if (w <= 12) run<42>(p, w);
            if (w <= 8) run<56>(p, w);
        }
    }
    printf("several consecutive time steps per workgroup (time-neighbour rows may hit in the CU's L1):\n");
    for (int rep = 0; rep < 2; ++rep) {
        run<28, 4, 10, false, 1>(p, 8); run<28, 4, 10, false, 4, true>(p, 8); run<28, 4, 10, false, 2, true>(p, 8);
        run<14, 4, 10, false, 1>(p, 8); run<14, 4, 10, false, 4, true>(p, 8);
        run<14, 4, 10, false, 1>(p, 16); run<14, 4, 10, false, 4, true>(p, 16); run<14, 4, 10, false, 2, true>(p, 16); run<14, 4, 10, false, 4, false>(p, 16);
        run<14, 4, 10, true, 1>(p, 16); run<14, 4, 10, true, 4, true>(p, 16);
    }
    if (argc > 3) return 0;
    printf("plain loads instead of nontemporal ones:\n");
    run<28, 4, 10, false>(p, 8); run<28, 4, 10, false>(p, 16); run<14, 4, 10, false>(p, 32); run<28, 61, 1, false>(p, 8);
    printf("T(t-1), T(t+1) replaced by two more reads of T(t) (certain cache hits):\n");
    p.tshift = 0; run<28, 4, 10>(p, 8); run<28, 4, 10>(p, 16); run<14, 4, 10>(p, 32); p.tshift = 1;
    // the same bytes, cut differently: more rows of fewer levels per wave (page locality), one level per wave, whole level slabs
    for (int w : {8, 16}) {
        run<28, 4, 10>(p, w); run<28, 8, 5>(p, w); run<28, 16, 3>(p, w); run<28, 16, 1>(p, w); run<28, 4, 1>(p, w); run<28, 4, 37>(p, w); run<28, 61, 1>(p, w); run<28, 1, 37>(p, w);
    }
    return 0;
}
