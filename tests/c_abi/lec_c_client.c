/* A plain-C client of liblec_hip.so: no Python, no torch, no C++ -- the C ABI as a reference maintainer's FFI would see it.
 * Reads a bundle (fields + the small host-built tables, written by tests/test_gpu_c_abi.py), copies everything to the GPU
 * with the HIP runtime's C API, calls lec_rowstats + lec_reduce, writes the per-time-step scalars and level tables.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude lec_c_client.c -o lec_c_client \
 *       -L/opt/rocm/lib -lamdhip64 -Llorenzcycletoolkit_amd -llec_hip
 *   ./lec_c_client bundle.bin out.bin
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "lec_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static void* upload(FILE* f, size_t bytes) {
    void* h = malloc(bytes);
    void* d = NULL;
    if (!h || fread(h, 1, bytes, f) != bytes) { fprintf(stderr, "short bundle\n"); exit(3); }
    if (hipMalloc(&d, bytes) != hipSuccess || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); exit(4); }
    free(h);
    return d;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s bundle.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    /* header: nt nl ny nx nxb nyb lon_uniform  (int32 x 7), phi_scale (double) */
    int32_t hd[8];
    double phi_scale;
    if (fread(hd, sizeof(int32_t), 8, f) != 8 || fread(&phi_scale, sizeof(double), 1, f) != 1) { fprintf(stderr, "bad header\n"); return 3; }
    const int nt = hd[0], nl = hd[1], ny = hd[2], nx = hd[3], nxb = hd[4], nyb = hd[5], uni = hd[6];
    const size_t cube = (size_t)nt * nl * ny * nx * sizeof(double);
    if (lec_version() != LEC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 5; }

    lec_rowstats_args ra;
    memset(&ra, 0, sizeof ra);
    ra.tair_d = upload(f, cube); ra.u_d = upload(f, cube); ra.v_d = upload(f, cube); ra.omega_d = upload(f, cube); ra.geopt_d = upload(f, cube);
    ra.dtype = LEC_F64; ra.with_q = 1;
    ra.nt = nt; ra.nl = nl; ra.ny = ny; ra.nx = nx; ra.t_begin = 0; ra.t_count = nt;
    ra.n_box = 1; ra.box_per_step = 0; ra.nxb_max = nxb; ra.nyb_max = nyb; ra.lon_uniform = uni;
    ra.box_d = (const int32_t*)upload(f, 4 * sizeof(int32_t));
    ra.boxtab_d = (const double*)upload(f, 4 * sizeof(double));
    ra.wlon_d = (const double*)upload(f, (size_t)nxb * sizeof(double));
    ra.glon_d = (const double*)upload(f, (size_t)nxb * 3 * sizeof(double));
    ra.lattab_d = (const double*)upload(f, (size_t)nyb * 4 * sizeof(double));
    ra.levtab_d = (const double*)upload(f, (size_t)nl * 3 * sizeof(double));
    ra.tcoef_d = (const double*)upload(f, (size_t)nt * 3 * sizeof(double));
    const double* boxtab2 = (const double*)upload(f, 4 * sizeof(double));
    const double* lattab2 = (const double*)upload(f, (size_t)nyb * 8 * sizeof(double));
    const double* levtab2 = (const double*)upload(f, (size_t)nl * 4 * sizeof(double));
    fclose(f);

    const size_t nrows = (size_t)nt * nl * nyb;
    double *rows, *am, *levraw, *scalars, *levels;
    int32_t *dropmask, *nanflag;
    CHECK_HIP(hipMalloc((void**)&rows, nrows * LEC_NSTAT * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&am, (size_t)nt * nl * 8 * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&levraw, (size_t)nt * nl * LEC_NLEVRAW * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&scalars, (size_t)nt * LEC_NSCALAR * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&levels, (size_t)nt * LEC_NLEVTAB * nl * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&dropmask, (size_t)LEC_NLEVFUN * nl * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void**)&nanflag, (size_t)nt * sizeof(int32_t)));
    ra.rows_d = rows;
    ra.stream = NULL;                                  /* the default stream */
    ra.tuning.kernel = LEC_KERNEL_AUTO;                /* all-zero tuning = the library's defaults; nothing is read from the environment */
    if (lec_rowstats(&ra) != LEC_OK) { fprintf(stderr, "lec_rowstats: %s\n", lec_last_error()); return 6; }

    lec_reduce_args rd;
    memset(&rd, 0, sizeof rd);
    rd.rows_d = rows; rd.t_count = nt; rd.nl = nl; rd.n_box = 1; rd.nyb_max = nyb;
    rd.box_d = ra.box_d; rd.boxtab2_d = boxtab2; rd.lattab2_d = lattab2; rd.levtab2_d = levtab2;
    rd.phi_scale = phi_scale; rd.drop_any_time = 1; rd.dropmask_d = dropmask;
    rd.am_d = am; rd.levraw_d = levraw; rd.scalars_d = scalars; rd.levels_d = levels; rd.nanflag_d = nanflag; rd.stream = NULL;
    if (lec_reduce(&rd) != LEC_OK) { fprintf(stderr, "lec_reduce: %s\n", lec_last_error()); return 7; }
    CHECK_HIP(hipDeviceSynchronize());

    /* error behaviour: a bad argument returns a code and a message, nothing is thrown */
    lec_rowstats_args bad = ra;
    bad.t_count = nt + 1;
    if (lec_rowstats(&bad) != LEC_ERR_ARG || strstr(lec_last_error(), "outside the cube") == NULL) { fprintf(stderr, "error path broken\n"); return 8; }

    bad = ra;
    bad.tuning.tile_t = -1;
    if (lec_rowstats(&bad) != LEC_ERR_ARG || strstr(lec_last_error(), "tile_t") == NULL) { fprintf(stderr, "tuning validation broken\n"); return 9; }

    double* hs = (double*)malloc((size_t)nt * LEC_NSCALAR * sizeof(double));
    double* hl = (double*)malloc((size_t)nt * LEC_NLEVTAB * nl * sizeof(double));
    CHECK_HIP(hipMemcpy(hs, scalars, (size_t)nt * LEC_NSCALAR * sizeof(double), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hl, levels, (size_t)nt * LEC_NLEVTAB * nl * sizeof(double), hipMemcpyDeviceToHost));
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    fwrite(hs, sizeof(double), (size_t)nt * LEC_NSCALAR, o);
    fwrite(hl, sizeof(double), (size_t)nt * LEC_NLEVTAB * nl, o);
    fclose(o);
    printf("ok: %d time steps, Az[0] = %.17g\n", nt, hs[0]);
    return 0;
}
