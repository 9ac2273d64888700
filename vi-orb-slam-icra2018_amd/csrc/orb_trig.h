// orb_trig.h -- cos/sin of a float angle, bit-identical to glibc (>= 2.28) cosf/sinf.
//
// The reference steers BRIEF with `(float)cos(angle)`, `(float)sin(angle)` on a float argument
// (src/ORBextractor.cc:114-115), i.e. libm's cosf/sinf.  This header restates the published
// algorithm of glibc's s_sincosf (ARM optimized-routines sincosf): x is promoted to double,
// reduced by multiples of pi/2 (n = round(x * 2/pi) via a 2^24-scaled multiply and shift,
// r = x - n*pi/2), and sin/cos are degree-7/8 minimax polynomials in double; the result is rounded
// to float once.  Every operation is an IEEE-754 double multiply or add with NO contraction
// (__dmul_rn/__dadd_rn on the device; -ffp-contract=off on the host), so host and device agree
// bit for bit.  Exhaustively compared with glibc 2.35 on all 1 087 050 388 floats in
// [0, 2*pi*1.01]: 0 mismatches for both functions (DESIGN.md "float reproducibility").
// Valid for |x| < 120 (the fast-reduction range); BRIEF only needs [0, 2*pi].
#ifndef ORB_TRIG_H
#define ORB_TRIG_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ORB_TRIG_FN __host__ __device__ __forceinline__
#else
#define ORB_TRIG_FN static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define ORB_DMUL(a, b) __dmul_rn((a), (b))
#define ORB_DADD(a, b) __dadd_rn((a), (b))
#else
// host: compile with -ffp-contract=off
#define ORB_DMUL(a, b) ((a) * (b))
#define ORB_DADD(a, b) ((a) + (b))
#endif

ORB_TRIG_FN uint32_t orb_f2u(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

// polynomial on the reduced argument; odd n -> cosine series, even n -> sine series.
// cs = +1 or -1 multiplies the cosine coefficients (quadrants 2,3).
ORB_TRIG_FN float orb_sincos_poly(double x, double x2, int n, double cs)
{
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                 C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = ORB_DMUL(x, x2);
        const double s1 = ORB_DADD(S2, ORB_DMUL(x2, S3));
        const double x7 = ORB_DMUL(x3, x2);
        const double s = ORB_DADD(x, ORB_DMUL(x3, S1));
        return (float)ORB_DADD(s, ORB_DMUL(x7, s1));
    } else {
        const double x4 = ORB_DMUL(x2, x2);
        const double c2 = ORB_DADD(cs * C3, ORB_DMUL(x2, cs * C4));
        const double c1 = ORB_DADD(cs * C0, ORB_DMUL(x2, cs * C1));
        const double x6 = ORB_DMUL(x4, x2);
        const double c = ORB_DADD(c1, ORB_DMUL(x4, cs * C2));
        return (float)ORB_DADD(c, ORB_DMUL(x6, c2));
    }
}

// *s = sinf(y), *c = cosf(y) for |y| < 120.
ORB_TRIG_FN void orb_sincosf(float y, float *s, float *c)
{
    const double HPI_INV = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
    const double HPI = 0x1.921FB54442D18p0;        // pi/2
    const uint32_t top = (orb_f2u(y) >> 20) & 0x7ff;
    double x = (double)y;
    if (top < ((orb_f2u(0x1.921FB6p-1f) >> 20) & 0x7ff)) {  // |y| < pi/4
        if (top < ((orb_f2u(0x1p-12f) >> 20) & 0x7ff)) {
            *s = y;
            *c = 1.0f;
            return;
        }
        const double x2 = ORB_DMUL(x, x);
        *s = orb_sincos_poly(x, x2, 0, 1.0);
        *c = orb_sincos_poly(x, x2, 1, 1.0);
        return;
    }
    const double r = ORB_DMUL(x, HPI_INV);
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = ORB_DADD(x, -ORB_DMUL((double)n, HPI));
    const double sgn = (n & 1) ^ ((n >> 1) & 1) ? -1.0 : 1.0;  // {+,-,-,+} by quadrant
    const double cs = (n & 2) ? -1.0 : 1.0;
    const double xs = ORB_DMUL(x, sgn);
    const double x2 = ORB_DMUL(x, x);
    *s = orb_sincos_poly(xs, x2, n, cs);
    *c = orb_sincos_poly(xs, x2, n ^ 1, cs);
}

#endif
