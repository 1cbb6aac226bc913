// probe_boxread.hip -- what can the memory system deliver for the moving-box access pattern?  (measurement tool, not product)
// NF cubes [T][37][162][243] fp64; per (t, k) a 61 x 61 box whose origin follows bench.py's synthetic track.  Kernels only load and
// add (one store per thread), all loads independent: the ceiling of the pattern, against a contiguous stream of the same bytes.
//   A  strips of 16 columns, 16 lanes per row segment (the box-tile kernel's load mapping), 256 threads per (t, k) tile
//   B  one wave per box row (61 of 64 lanes; the one-wave-per-row kernel's mapping)
//   C  as A, whole 128-byte lines: segments widened to the enclosing 16-element-aligned span (where rows are aligned)
//   S  contiguous stream of the same number of bytes per field
//   E  (round 5) a per-step, box-ALIGNED crop: cubes [T][37][61][64] -- every box row starts on a 128-byte line and has a 512-byte
//      pitch, what a crop written per time step by lec_ingest for the moving framework would give (4 lines per 61-point row instead
//      of 4.8) -- read one wave per row like B; the rate is quoted on the same ALGORITHMIC bytes (61 x 61 x 8 per level and field)
// Build: hipcc -O3 --offload-arch=gfx950 probe_boxread.hip -o probe_boxread ; run: ./probe_boxread [T]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NL = 37, NY = 162, NX = 243, NB = 61;
struct P { const double* f[8]; const double* g[8]; const int* box; double* out; int T; int nx; int ny; };

template <int NF, int VARIANT>
__global__ void __launch_bounds__(256) probe(const P p) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tchunk = (p.T + 7) / 8;
    const int k = q % NL, tl = xcd * tchunk + q / NL;
    if (q / NL >= tchunk || tl >= p.T) return;
    const int iw = p.box[2 * tl], js = p.box[2 * tl + 1];
    const size_t plane = (size_t)p.ny * p.nx;
    const size_t base = ((size_t)tl * NL + k) * plane;
    const int tid = threadIdx.x;
    double acc = 0.0;
    if (VARIANT == 0 || VARIANT == 2) {
        const int lc = tid & 15, lr = tid >> 4;
        const int c_lo = (VARIANT == 2) ? (iw & ~15) - iw : 0;                 // widen to aligned spans
        const int c_hi = (VARIANT == 2) ? ((iw + NB + 15) & ~15) - iw : NB;
        for (int c0 = c_lo; c0 < c_hi; c0 += 16) {
            const int col = min(max(iw + c0 + lc, 0), p.nx - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = js + min(lr + 16 * r, NB - 1);
#pragma unroll
                for (int f = 0; f < NF; ++f) acc += __builtin_nontemporal_load(p.f[f] + base + (size_t)row * p.nx + col);
            }
        }
    } else if (VARIANT == 1) {
        const int lane = tid & 63, w = tid >> 6;
        const int col = iw + min(lane, NB - 1);
        for (int r = w; r < NB; r += 4) {
#pragma unroll
            for (int f = 0; f < NF; ++f) acc += __builtin_nontemporal_load(p.f[f] + base + (size_t)(js + r) * p.nx + col);
        }
    } else if (VARIANT == 5) {
        const int lane = tid & 63, w = tid >> 6;
        const size_t cbase = ((size_t)tl * NL + k) * (NB * 64);
        const int col = min(lane, NB - 1);
        for (int r = w; r < NB; r += 4) {
#pragma unroll
            for (int f = 0; f < NF; ++f) acc += __builtin_nontemporal_load(p.g[f] + cbase + (size_t)r * 64 + col);
        }
    } else if (VARIANT == 4) {
        // D: 16 bytes per lane (two columns), 32 lanes per row, two rows per wave instruction (rows are only 8-byte aligned)
        typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
        const int lane = tid & 63, w = tid >> 6;
        const int half = lane >> 5, l2 = lane & 31;
        const int col = iw + min(2 * l2, NB - 2);
        for (int r = 2 * w + half; r < NB; r += 8) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p.f[f] + base + (size_t)(js + r) * p.nx + col));
                acc += v.x + v.y;
            }
        }
    } else {
        // contiguous: the same bytes per (t, k, field) as a box (61 x 61 doubles), read as one span
        for (int e = tid; e < NB * NB; e += 256) {
#pragma unroll
            for (int f = 0; f < NF; ++f) acc += __builtin_nontemporal_load(p.f[f] + base + e);
        }
    }
    p.out[(size_t)blockIdx.x * 256 + tid] = acc;
}

template <int NF, int V>
float run(const P& p, int reps) {
    const int tchunk = (p.T + 7) / 8;
    dim3 grid(8 * tchunk * NL), block(256);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<NF, V>), grid, block, 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<NF, V>), grid, block, 0, 0, p);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 256;
    const int nx = argc > 2 ? atoi(argv[2]) : NX;          // 243 (bench.py's crop) or e.g. 256 (128-byte-aligned rows)
    const size_t n = (size_t)T * NL * NY * nx;
    P p; p.T = T; p.nx = nx; p.ny = NY;
    for (int f = 0; f < 7; ++f) { double* d; CK(hipMalloc(&d, n * 8)); CK(hipMemset(d, 0, n * 8)); p.f[f] = d; }
    const size_t nc = (size_t)T * NL * NB * 64;
    for (int f = 0; f < 7; ++f) { double* d; CK(hipMalloc(&d, nc * 8)); CK(hipMemset(d, 0, nc * 8)); p.g[f] = d; }
    std::vector<int> box(2 * T);
    for (int t = 0; t < T; ++t) {
        const double clat = -37.5 + 12.0 * sin(2 * M_PI * t / 400.0), clon = -50.0 + 22.0 * cos(2 * M_PI * t / 700.0);
        box[2 * t] = (int)lround((clon - 7.5 + 80.25) / 0.25); box[2 * t + 1] = (int)lround((clat - 7.5 + 57.75) / 0.25);
    }
    int* db; CK(hipMalloc(&db, box.size() * 4)); CK(hipMemcpy(db, box.data(), box.size() * 4, hipMemcpyHostToDevice)); p.box = db;
    double* out; CK(hipMalloc(&out, (size_t)8 * ((T + 7) / 8) * NL * 256 * 8)); p.out = out;
    const double gb = (double)T * NL * NB * NB * 8 / 1e9;      // algorithmic GB per field
    printf("T=%d nx=%d: %.3f GB per field per launch\n", T, nx, gb);
#define ROW(NF) do { \
    const float a = run<NF, 0>(p, 5), b = run<NF, 1>(p, 5), c = run<NF, 2>(p, 5), s = run<NF, 3>(p, 5), d = run<NF, 4>(p, 5), e = run<NF, 5>(p, 5); \
    printf("NF=%d  A strips %.3f ms %.0f GB/s | B rows %.3f ms %.0f GB/s | C aligned spans %.3f ms %.0f GB/s (algorithmic) | D 16B/lane 2 rows %.3f ms %.0f GB/s | S stream %.3f ms %.0f GB/s | E aligned per-step crop, rows %.3f ms %.0f GB/s\n", \
           NF, a, NF * gb / a * 1e3, b, NF * gb / b * 1e3, c, NF * gb / c * 1e3, d, NF * gb / d * 1e3, s, NF * gb / s * 1e3, e, NF * gb / e * 1e3); } while (0)
    ROW(1); ROW(4); ROW(5); ROW(7);
    return 0;
}
