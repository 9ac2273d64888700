// Host build of csrc/orb_trig.h (the device's cos/sin sequence) + a sweep that compares it with
// this machine's libm cosf/sinf, for tests/test_trig.py.
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "orb_trig.h"

extern "C" void trig_host_sincos(float x, float *s, float *c) { orb_sincosf(x, s, c); }

// Compares on floats with bit patterns lo, lo+step, ... <= hi.  Returns the number of mismatches
// (sin or cos differing from libm in any bit).
extern "C" long trig_host_sweep(uint32_t lo, uint32_t hi, uint32_t step, float *first_bad)
{
    long bad = 0;
    for (uint64_t u = lo; u <= hi; u += step) {
        uint32_t b = (uint32_t)u;
        float f, s, c;
        memcpy(&f, &b, 4);
        orb_sincosf(f, &s, &c);
        float ls = sinf(f), lc = cosf(f);
        if (memcmp(&s, &ls, 4) || memcmp(&c, &lc, 4)) {
            if (!bad && first_bad) *first_bad = f;
            bad++;
        }
    }
    return bad;
}
