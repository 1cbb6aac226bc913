"""Physical constants: the MetPy 1.6.2 values the reference uses (thermodynamics.py:21-22,
conversion_terms.py:31, boundary_terms.py:31, energy_contents.py:31 of the reference).
They must agree with lorenzcycletoolkit_amd/csrc/lec_internal.h."""
G = 9.80665
RE = 6371008.7714
RD = 8.314462618 / 28.96546e-3
CP_D = 1.4 * RD / (1.4 - 1.0)
KAPPA = RD / CP_D
P0_PA = 100000.0

# column order of the `scalars` output of lec_reduce (include/lec_hip.h: LEC_NSCALAR)
SCALAR_TERMS = ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce",
                "BAz", "BAe", "BKz", "BKe", "BΦZ", "BΦE", "Gz", "Ge"]
# table order of the `levels` output (lec_fixed_framework.py:172-194 of the reference)
LEVEL_TERMS = ["Az", "Ae", "Kz", "Ke", "Ge", "Gz", "Cz", "Cz_1", "Cz_2", "Ca", "Ca_1", "Ca_2",
               "Ce", "Ce_1", "Ce_2", "Ck", "Ck_1", "Ck_2", "Ck_3", "Ck_4", "Ck_5"]
