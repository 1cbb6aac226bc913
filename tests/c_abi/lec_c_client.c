/* A plain-C client of liblec_hip.so: no Python, no torch, no C++ -- the C ABI as a reference maintainer's FFI would see it.
 * Reads a bundle (fields + the small host-built tables, written by tests/test_gpu_c_abi.py), copies everything to the GPU
 * with the HIP runtime's C API, calls lec_rowstats + lec_reduce, writes the per-time-step scalars and level tables.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude lec_c_client.c -o lec_c_client \
 *       -L/opt/rocm/lib -lamdhip64 -Llorenzcycletoolkit_amd -llec_hip
 *   ./lec_c_client bundle.bin out.bin [out_packed.bin out_table.csv]
 *
 * With the two optional outputs it also walks the ABI-9 additions: the same box handed over as a BOX-PACKED moving series (one box per
 * time step -- here the same one --, each step's box copied to the origin of its slab with hipMemcpy2D, dT/dt as a cube from lec_dtdt),
 * run through lec_rowstats (box_per_step) + lec_reduce, and the Az table of the first run as CSV text from lec_format_csv_rows.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "lec_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static void* upload_keep(FILE* f, size_t bytes, void** host) {
    void* h = malloc(bytes);
    void* d = NULL;
    if (!h || fread(h, 1, bytes, f) != bytes) { fprintf(stderr, "short bundle\n"); exit(3); }
    if (hipMalloc(&d, bytes) != hipSuccess || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); exit(4); }
    if (host) *host = h; else free(h);
    return d;
}

static void* upload(FILE* f, size_t bytes) { return upload_keep(f, bytes, NULL); }

/* `n` copies of a host table, one after the other, on the device (the per-box tables of a series of n identical boxes) */
static void* replicate(const void* h, size_t bytes, int n) {
    char* d = NULL;
    if (hipMalloc((void**)&d, bytes * (size_t)n) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(4); }
    for (int i = 0; i < n; ++i)
        if (hipMemcpy(d + (size_t)i * bytes, h, bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); exit(4); }
    return d;
}

int main(int argc, char** argv) {
    if (argc != 3 && argc != 5) { fprintf(stderr, "usage: %s bundle.bin out.bin [out_packed.bin out_table.csv]\n", argv[0]); return 1; }
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
    void *h_box, *h_boxtab, *h_wlon, *h_glon, *h_lattab, *h_boxtab2, *h_lattab2;      /* host copies: the packed series replicates them per step */
    ra.box_d = (const int32_t*)upload_keep(f, 4 * sizeof(int32_t), &h_box);
    ra.boxtab_d = (const double*)upload_keep(f, 4 * sizeof(double), &h_boxtab);
    ra.wlon_d = (const double*)upload_keep(f, (size_t)nxb * sizeof(double), &h_wlon);
    ra.glon_d = (const double*)upload_keep(f, (size_t)nxb * 3 * sizeof(double), &h_glon);
    ra.lattab_d = (const double*)upload_keep(f, (size_t)nyb * 4 * sizeof(double), &h_lattab);
    ra.levtab_d = (const double*)upload(f, (size_t)nl * 3 * sizeof(double));
    ra.tcoef_d = (const double*)upload(f, (size_t)nt * 3 * sizeof(double));
    const double* boxtab2 = (const double*)upload_keep(f, 4 * sizeof(double), &h_boxtab2);
    const double* lattab2 = (const double*)upload_keep(f, (size_t)nyb * 8 * sizeof(double), &h_lattab2);
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
        /* ABI 10: a per-step gather table (step_d: {source step, latitude offset, longitude offset} per output step) is device memory
         * too; the call carries what bounds it (nt_src, jmap_len, imap_len) and the same scan walks it: a good table, then one whose
         * second step names a source step the source does not hold */
        int32_t steps[6] = {7, 0, 0,   8, 1, 0};        /* two output steps of one row x one column out of a 2-step source that starts with step 7 */
        int32_t* steps_d;
        CHECK_HIP(hipMalloc((void**)&steps_d, sizeof steps));
        CHECK_HIP(hipMemcpy(steps_d, steps, sizeof steps, hipMemcpyHostToDevice));
        ga.nt = 2; ga.ny = 1; ga.step_d = steps_d; ga.step_base = 7; ga.nt_src = 2; ga.jmap_len = 2; ga.imap_len = 1;
        if (lec_check_maps(&ga, status) != LEC_OK) { fprintf(stderr, "lec_check_maps (step table): %s\n", lec_last_error()); return 14; }
        steps[3] = 9;
        CHECK_HIP(hipMemcpy(steps_d, steps, sizeof steps, hipMemcpyHostToDevice));
        if (lec_check_maps(&ga, status) != LEC_ERR_ARG || strstr(lec_last_error(), "step_d[1]") == NULL) {
            fprintf(stderr, "lec_check_maps missed a bad step entry: %s\n", lec_last_error()); return 15;
        }
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
    if (argc != 5) return 0;

    /* ---- ABI 9, part 1: the Az table (first of the 21) as the text the reference's DataFrame.to_csv(header=None) writes ---- */
    {
        char* labels = (char*)malloc((size_t)nt * 19 + 1);
        for (int t = 0; t < nt; ++t) snprintf(labels + (size_t)t * 19, 20, "2005-08-%02d %02d:00:00", 8 + (6 * t) / 24, (6 * t) % 24);
        const long long cap = (long long)nt * (19 + 27 * nl + 1);
        char* text = (char*)malloc((size_t)cap);
        double* az = (double*)malloc((size_t)nt * nl * sizeof(double));
        for (int t = 0; t < nt; ++t) memcpy(az + (size_t)t * nl, hl + (size_t)t * LEC_NLEVTAB * nl, (size_t)nl * sizeof(double));
        const long long n = lec_format_csv_rows(az, nt, nl, nl, labels, 19, text, cap);
        if (n < 0) { fprintf(stderr, "lec_format_csv_rows: %s\n", lec_last_error()); return 20; }
        FILE* c = fopen(argv[4], "wb");
        if (!c) { perror(argv[4]); return 1; }
        fwrite(text, 1, (size_t)n, c);
        fclose(c);
        if (lec_format_csv_rows(az, nt, nl, nl, labels, 19, text, 10) != -1) { fprintf(stderr, "lec_format_csv_rows took a short buffer\n"); return 21; }
    }

    /* ---- ABI 9, part 2: the same box as a BOX-PACKED moving series ---- */
    {
        const int32_t* bx = (const int32_t*)h_box;                       /* iw ie js jn */
        const int iw = bx[0], js = bx[2];
        const size_t slab = (size_t)nl * nyb * nxb, pcube = (size_t)nt * slab * sizeof(double);
        const void* src[5] = {ra.tair_d, ra.u_d, ra.v_d, ra.omega_d, ra.geopt_d};
        double* pk[7];                                                   /* T u v omega Phi, T(t-1), T(t+1): each step's box at its slab's origin */
        for (int c = 0; c < 7; ++c) CHECK_HIP(hipMalloc((void**)&pk[c], pcube));
        for (int c = 0; c < 7; ++c)
            for (int t = 0; t < nt; ++t) {
                const int ts = c < 5 ? t : (c == 5 ? (t > 0 ? t - 1 : t) : (t < nt - 1 ? t + 1 : t));      /* the step itself where a neighbour does not exist */
                const double* s0 = (const double*)(c < 5 ? src[c] : ra.tair_d) + (size_t)ts * nl * ny * nx;
                for (int k = 0; k < nl; ++k)
                    CHECK_HIP(hipMemcpy2D(pk[c] + (size_t)t * slab + (size_t)k * nyb * nxb, (size_t)nxb * sizeof(double),
                                          s0 + ((size_t)k * ny + js) * nx + iw, (size_t)nx * sizeof(double),
                                          (size_t)nxb * sizeof(double), (size_t)nyb, hipMemcpyDeviceToDevice));
            }
        double* dtdt;
        CHECK_HIP(hipMalloc((void**)&dtdt, pcube));
        lec_dtdt_args da;
        memset(&da, 0, sizeof da);
        da.tm_d = pk[5]; da.t_d = pk[0]; da.tp_d = pk[6]; da.dtype = LEC_F64; da.n_steps = nt; da.step_elems = (int64_t)slab;
        da.tcoef_d = ra.tcoef_d; da.out_d = dtdt; da.stream = NULL;
        if (lec_dtdt(&da) != LEC_OK) { fprintf(stderr, "lec_dtdt: %s\n", lec_last_error()); return 22; }

        int32_t* origin = (int32_t*)malloc((size_t)nt * 4 * sizeof(int32_t));
        for (int t = 0; t < nt; ++t) { origin[4 * t] = 0; origin[4 * t + 1] = nxb - 1; origin[4 * t + 2] = 0; origin[4 * t + 3] = nyb - 1; }
        int32_t* origin_d;
        CHECK_HIP(hipMalloc((void**)&origin_d, (size_t)nt * 4 * sizeof(int32_t)));
        CHECK_HIP(hipMemcpy(origin_d, origin, (size_t)nt * 4 * sizeof(int32_t), hipMemcpyHostToDevice));

        lec_rowstats_args rp = ra;
        rp.tair_d = pk[0]; rp.u_d = pk[1]; rp.v_d = pk[2]; rp.omega_d = pk[3]; rp.geopt_d = pk[4]; rp.dTdt_d = dtdt;
        rp.ny = nyb; rp.nx = nxb;                                        /* the slabs */
        rp.n_box = nt; rp.box_per_step = 1;
        rp.box_d = origin_d;                                             /* the boxes as the cubes hold them ... */
        rp.boxtab_d = (const double*)replicate(h_boxtab, 4 * sizeof(double), nt);       /* ... every table from the true grid box */
        rp.wlon_d = (const double*)replicate(h_wlon, (size_t)nxb * sizeof(double), nt);
        rp.glon_d = (const double*)replicate(h_glon, (size_t)nxb * 3 * sizeof(double), nt);
        rp.lattab_d = (const double*)replicate(h_lattab, (size_t)nyb * 4 * sizeof(double), nt);
        rp.tcoef_d = NULL;
        if (lec_check_boxes(&rp, status) != LEC_OK) { fprintf(stderr, "lec_check_boxes (packed): %s\n", lec_last_error()); return 23; }
        if (lec_rowstats(&rp) != LEC_OK) { fprintf(stderr, "lec_rowstats (packed): %s\n", lec_last_error()); return 24; }
        lec_reduce_args r2 = rd;
        r2.n_box = nt; r2.drop_any_time = 0;                             /* the moving framework: one BoxData per step */
        r2.box_d = (const int32_t*)replicate(h_box, 4 * sizeof(int32_t), nt);
        r2.boxtab2_d = (const double*)replicate(h_boxtab2, 4 * sizeof(double), nt);
        r2.lattab2_d = (const double*)replicate(h_lattab2, (size_t)nyb * 8 * sizeof(double), nt);
        if (lec_reduce(&r2) != LEC_OK) { fprintf(stderr, "lec_reduce (packed): %s\n", lec_last_error()); return 25; }
        CHECK_HIP(hipDeviceSynchronize());
        CHECK_HIP(hipMemcpy(hs, scalars, (size_t)nt * LEC_NSCALAR * sizeof(double), hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(hl, levels, (size_t)nt * LEC_NLEVTAB * nl * sizeof(double), hipMemcpyDeviceToHost));
        FILE* o2 = fopen(argv[3], "wb");
        if (!o2) { perror(argv[3]); return 1; }
        fwrite(hs, sizeof(double), (size_t)nt * LEC_NSCALAR, o2);
        fwrite(hl, sizeof(double), (size_t)nt * LEC_NLEVTAB * nl, o2);
        fclose(o2);
        /* one neighbour cube without the other is refused */
        lec_rowstats_args half = rp;
        half.dTdt_d = NULL; half.tm_d = pk[5]; half.tcoef_d = ra.tcoef_d;
        if (lec_rowstats(&half) != LEC_ERR_ARG || strstr(lec_last_error(), "tm_d and tp_d") == NULL) { fprintf(stderr, "packed validation broken\n"); return 26; }
        printf("ok: the same box as a box-packed series, Az[0] = %.17g\n", hs[0]);
    }
    return 0;
}
