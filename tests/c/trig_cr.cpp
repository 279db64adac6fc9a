// Host build of csrc/rt_trig.h (the same text the gfx950 kernel compiles) for tests/test_trig_cr.py.
#include "../../ray-tracing-fsharp_amd/csrc/rt_trig.h"

extern "C" void trig_cr(int op, int n, const double *a, const double *b, double *out) {
    for (int i = 0; i < n; ++i) {
        switch (op) {
        case 7: out[i] = rtt::cr_acos(a[i]); break;
        case 8: out[i] = rtt::cr_sin(a[i]); break;
        default: out[i] = rtt::cr_atan2(a[i], b[i]); break;
        }
    }
}
// sin and cos as double-doubles (hi, lo pairs), to measure the accuracy of the intermediate itself
extern "C" void trig_sincos_dd(int n, const double *a, double *s, double *c) {
    for (int i = 0; i < n; ++i) {
        rtt::dd sv, cv;
        rtt::sincos_dd(a[i], sv, cv);
        s[2 * i] = sv.hi; s[2 * i + 1] = sv.lo; c[2 * i] = cv.hi; c[2 * i + 1] = cv.lo;
    }
}
