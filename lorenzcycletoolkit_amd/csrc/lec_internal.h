// lec_internal.h -- constants and helpers shared by the HIP translation units (not part of the ABI).
#ifndef LEC_INTERNAL_H
#define LEC_INTERNAL_H

// MetPy 1.6.2 constants used by the reference (thermodynamics.py:21-22, conversion_terms.py:31,
// boundary_terms.py:31, energy_contents.py:31); SURVEY.md appendix E.
#define LEC_G 9.80665
#define LEC_RE 6371008.7714
#define LEC_RD (8.314462618 / 28.96546e-3)
#define LEC_CP_D (1.4 * LEC_RD / (1.4 - 1.0))

// vectors per lane of the largest row kernel (rows up to 256 * LEC_MAX_ITERS vectors)
#define LEC_MAX_ITERS 8
// > 0 overrides the per-configuration waves-per-SIMD request of the row kernel (experiments)
#ifndef LEC_MINW
#define LEC_MINW 0
#endif
// waves per SIMD requested for the single-sweep row kernel
#ifndef LEC_MINW_SINGLE
#define LEC_MINW_SINGLE 4
#endif

// records an error message (thread-local) and returns `code`
int lec_set_error(int code, const char* msg);

#endif
