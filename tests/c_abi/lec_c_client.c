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

    /* box_d lives in device memory: the library cannot see it when it validates the arguments, so a C caller that fills the
     * table itself asks for the scan first (the Python host checks its boxes before uploading them) */
    int32_t* status;
    CHECK_HIP(hipMalloc((void**)&status, 4 * sizeof(int32_t)));
    if (lec_check_boxes(&ra, status) != LEC_OK) { fprintf(stderr, "lec_check_boxes: %s\n", lec_last_error()); return 10; }
    {   /* ... and a table with a box beyond the grid (and one that is too wide for nxb_max) is refused, naming the first */
        lec_rowstats_args chk = ra;
        int32_t three[12] = {0, 1, 0, 1,   0, nx, 0, 1,   0, nxb, 0, 1};        /* ok | ie == nx: outside | nxb + 1 columns: wider than nxb_max */
        int32_t* three_d;
        CHECK_HIP(hipMalloc((void**)&three_d, sizeof three));
        CHECK_HIP(hipMemcpy(three_d, three, sizeof three, hipMemcpyHostToDevice));
        chk.box_d = three_d; chk.n_box = 3;
        if (lec_check_boxes(&chk, status) != LEC_ERR_ARG || strstr(lec_last_error(), "first: box 1") == NULL || strstr(lec_last_error(), "2 of 3") == NULL) {
            fprintf(stderr, "lec_check_boxes missed a bad box: %s\n", lec_last_error()); return 11;
        }
        lec_ingest_args ga;
        memset(&ga, 0, sizeof ga);
        int32_t maps[6] = {0, 1, 2, 0, 5, 1};           /* kmap {0,1,2} of 3 source levels: fine; jmap {0,5} of 4 source rows: jmap[1] is outside */
        int32_t* maps_d;
        CHECK_HIP(hipMalloc((void**)&maps_d, sizeof maps));
        CHECK_HIP(hipMemcpy(maps_d, maps, sizeof maps, hipMemcpyHostToDevice));
        ga.nl_in = 3; ga.ny_in = 4; ga.nx_in = 2; ga.nl = 3; ga.ny = 2; ga.nx = 1;
        ga.kmap_d = maps_d; ga.jmap_d = maps_d + 3; ga.imap_d = maps_d + 5;
        if (lec_check_maps(&ga, status) != LEC_ERR_ARG || strstr(lec_last_error(), "jmap_d[1]") == NULL) {
            fprintf(stderr, "lec_check_maps missed a bad entry: %s\n", lec_last_error()); return 12;
        }
        maps[4] = 3;
        CHECK_HIP(hipMemcpy(maps_d, maps, sizeof maps, hipMemcpyHostToDevice));
        if (lec_check_maps(&ga, status) != LEC_OK) { fprintf(stderr, "lec_check_maps: %s\n", lec_last_error()); return 13; }
    }
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
