// lec_format.hip -- host side of the per-level CSV tables (include/lec_hip.h: lec_format_csv_rows).  No device code.
//
// The reference appends its 21 per-level tables through pandas (`_save_vertical_levels`: conversion_terms.py:287-308 and its three
// copies; lec_fixed_framework.py:172-197 writes the headers): DataFrame.to_csv(mode="a", header=None), i.e. every float64 cell as
// Python's repr(float), a NaN as an empty field.  For a month of hourly steps that is 578,000 cells formatted one Python object
// at a time (0.6 s of a 3.7-s run, profiles/r04_notes.md section 6).  This formats a whole table in one call, byte for byte the
// same text:
//   repr(float) = the shortest digit string that reads back to the same double (std::to_chars gives exactly that), laid out by
//   CPython's rule for 'r' (Python/pystrtod.c, format_float_short): with the value = 0.d1d2...dn x 10^decpt, exponent notation
//   when decpt <= -4 or decpt > 16 ("1e-05", "1.5e+16": a sign and at least two exponent digits, no ".0"), otherwise positional
//   with at least one digit after the point ("0.0001", "123.0", "1234567890123456.0").
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/lec_hip.h"
#include "lec_internal.h"

namespace {

// writes repr(v) at p (at most 26 bytes), returns the end
char* put_repr(char* p, double v) {
    if (std::isinf(v)) {
        if (v < 0) *p++ = '-';
        memcpy(p, "inf", 3);
        return p + 3;
    }
    if (std::signbit(v)) {
        *p++ = '-';
        v = -v;
    }
    if (v == 0.0) {
        memcpy(p, "0.0", 3);
        return p + 3;
    }
    char sci[40];
    auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);      // d[.ddd]e[+-]XX, shortest round trip
    char digits[24];
    int nd = 0;
    const char* q = sci;
    for (; q < r.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    int e10 = 0;
    {
        const char* s = q + 1;
        bool neg = (*s == '-');
        if (*s == '-' || *s == '+') ++s;
        for (; s < r.ptr; ++s) e10 = e10 * 10 + (*s - '0');
        if (neg) e10 = -e10;
    }
    const int decpt = e10 + 1;
    if (decpt <= -4 || decpt > 16) {
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            memcpy(p, digits + 1, nd - 1);
            p += nd - 1;
        }
        *p++ = 'e';
        int e = decpt - 1;
        *p++ = (e < 0) ? '-' : '+';
        if (e < 0) e = -e;
        if (e >= 100) {
            *p++ = char('0' + e / 100);
            e %= 100;
            *p++ = char('0' + e / 10);
            *p++ = char('0' + e % 10);
        } else {
            *p++ = char('0' + e / 10);
            *p++ = char('0' + e % 10);
        }
        return p;
    }
    if (decpt <= 0) {
        *p++ = '0';
        *p++ = '.';
        for (int i = 0; i < -decpt; ++i) *p++ = '0';
        memcpy(p, digits, nd);
        return p + nd;
    }
    if (decpt >= nd) {
        memcpy(p, digits, nd);
        p += nd;
        for (int i = nd; i < decpt; ++i) *p++ = '0';
        *p++ = '.';
        *p++ = '0';
        return p;
    }
    memcpy(p, digits, decpt);
    p += decpt;
    *p++ = '.';
    memcpy(p, digits + decpt, nd - decpt);
    return p + (nd - decpt);
}

}  // namespace

extern "C" long long lec_format_csv_rows(const double* values, long long rows, long long cols, long long row_stride,
                                         const char* labels, int label_len, char* out, long long cap) {
    if (!values || !out || rows < 0 || cols < 1 || row_stride < cols || label_len < 0 || (label_len > 0 && !labels)) {
        lec_set_error(LEC_ERR_ARG, "lec_format_csv_rows: null pointer, negative size or a row stride shorter than a row");
        return -1;
    }
    const long long need = rows * ((long long)label_len + cols * 27 + 1);      // a cell: at most 25 characters + its separator
    if (cap < need) {
        lec_set_error(LEC_ERR_ARG, "lec_format_csv_rows: the output buffer must hold rows * (label_len + 27 * cols + 1) bytes");
        return -1;
    }
    char* p = out;
    for (long long r = 0; r < rows; ++r) {
        memcpy(p, labels + r * label_len, label_len);
        p += label_len;
        const double* v = values + r * row_stride;
        for (long long c = 0; c < cols; ++c) {
            *p++ = ',';
            if (!std::isnan(v[c])) p = put_repr(p, v[c]);
        }
        *p++ = '\n';
    }
    return (long long)(p - out);
}
