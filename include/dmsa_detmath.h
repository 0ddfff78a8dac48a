/*
 * dmsa_detmath.h — double-precision sin / cos / acos / atan2 as fixed sequences of IEEE-754 operations.
 *
 * Why: the dense pose table (ContinuousTrajectory::updateTrajDenseTforms, ContinuousTrajectory.h:189-226, and the per-keyframe
 * transforms MapManagement.h:140-147) goes through sin, cos, acos and atan2 (Eigen's AngleAxis / Quaternion / slerp and the
 * Rodrigues form of skew(w).exp()).  The reference calls glibc; a GPU calls its own math library, and the two differ in the
 * last bit of a few percent of the results, which is enough to change a float entry of a pose table once in a while and, through
 * the numeric Jacobian, a trajectory in the fifth digit.  Built only from +, -, *, /, sqrt and integer bit tests — each of them
 * correctly rounded on the host and on gfx950, and none of them fused (-ffp-contract=off) — these functions return the SAME
 * bits on both sides, so pose tables can be built on the device and still be compared bit for bit with the CPU oracle.
 *
 * What: the classic fdlibm algorithms (Sun Microsystems' freely distributable libm; restated here, not copied):
 *   sin, cos   Cody–Waite reduction by pi/2 in up to three stages (exact for |x| < 2^20 * pi/2) + degree-13 / degree-14
 *              minimax kernels on [-pi/4, pi/4] with a tail term;
 *   acos       rational approximation R(x^2) on |x| < 0.5, sqrt identities with a split square root above;
 *   atan2      argument reduction to four intervals + odd/even split minimax polynomial.
 * Each is documented < 1 ulp; against this machine's glibc the committed test (tests/test_detmath.py) measures <= 1 ulp on
 * dense samples of the argument ranges the pose tables use.  Deviation from the reference: it evaluates these four functions
 * with glibc, so a result can differ from the reference's by one unit in the last place of a double (stated in
 * oracle/dmsa_oracle.cpp as well, which includes this header: oracle, product host code and device kernels share one definition).
 */
#ifndef DMSA_DETMATH_H
#define DMSA_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define DMSA_DET_HD __host__ __device__ inline
#else
#define DMSA_DET_HD inline
#endif

namespace dmsa_det {

DMSA_DET_HD int32_t hi_word(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return (int32_t)(u >> 32);
}
DMSA_DET_HD uint32_t lo_word(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return (uint32_t)u;
}
DMSA_DET_HD double from_words(int32_t hi, uint32_t lo) {
    const uint64_t u = ((uint64_t)(uint32_t)hi << 32) | lo;
    double x;
    memcpy(&x, &u, 8);
    return x;
}
DMSA_DET_HD double det_fabs(double x) { return from_words(hi_word(x) & 0x7fffffff, lo_word(x)); }
// correctly rounded on both sides: the host compiler emits sqrtsd, hipcc the refined v_sqrt_f64 sequence
DMSA_DET_HD double det_sqrt(double x) { return __builtin_sqrt(x); }

// sin on [-pi/4, pi/4] of x + y (y: tail of the reduced argument)
DMSA_DET_HD double kernel_sin(double x, double y, int have_tail) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    if ((hi_word(x) & 0x7fffffff) < 0x3e400000) return x;  // |x| < 2^-27
    const double z = x * x;
    const double v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (!have_tail) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
// cos on [-pi/4, pi/4] of x + y
DMSA_DET_HD double kernel_cos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const int32_t ix = hi_word(x) & 0x7fffffff;
    if (ix < 0x3e400000) return 1.0;  // |x| < 2^-27
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));  // |x| < 0.3
    const double qx = ix > 0x3fe90000 ? 0.28125 : from_words(ix - 0x00200000, 0);  // ~ x / 4
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}
// x = n * pi/2 + (y0 + y1), |y0 + y1| <= pi/4; returns n.  |x| < 2^20 * pi/2 (the pose tables stay below 10)
DMSA_DET_HD int rem_pio2(double x, double* y0, double* y1) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                 pio2_3t = 8.47842766036889956997e-32;
    const int32_t hx = hi_word(x);
    const int32_t ix = hx & 0x7fffffff;
    const double t0 = det_fabs(x);
    const int n = (int)(t0 * invpio2 + 0.5);
    const double fn = (double)n;
    double r = t0 - fn * pio2_1;  // exact: pio2_1 carries 33 bits
    double w = fn * pio2_1t;
    const int j = ix >> 20;
    double a = r - w;
    int i = j - ((hi_word(a) >> 20) & 0x7ff);
    if (i > 16) {  // cancellation: second piece of pi/2
        double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        a = r - w;
        i = j - ((hi_word(a) >> 20) & 0x7ff);
        if (i > 49) {  // third piece: 151 bits of pi/2 in total
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            a = r - w;
        }
    }
    const double b = (r - a) - w;
    if (hx < 0) {
        *y0 = -a, *y1 = -b;
        return -n;
    }
    *y0 = a, *y1 = b;
    return n;
}
DMSA_DET_HD double det_sin(double x) {
    if ((hi_word(x) & 0x7fffffff) <= 0x3fe921fb) return kernel_sin(x, 0.0, 0);  // |x| <= pi/4
    double y0, y1;
    const int n = rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return kernel_sin(y0, y1, 1);
        case 1: return kernel_cos(y0, y1);
        case 2: return -kernel_sin(y0, y1, 1);
        default: return -kernel_cos(y0, y1);
    }
}
DMSA_DET_HD double det_cos(double x) {
    if ((hi_word(x) & 0x7fffffff) <= 0x3fe921fb) return kernel_cos(x, 0.0);
    double y0, y1;
    const int n = rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return kernel_cos(y0, y1);
        case 1: return -kernel_sin(y0, y1, 1);
        case 2: return -kernel_cos(y0, y1);
        default: return kernel_sin(y0, y1, 1);
    }
}
// acos on [-1, 1] (outside: NaN, like libm)
DMSA_DET_HD double det_acos(double x) {
    const double pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                 pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const int32_t hx = hi_word(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x3ff00000) {  // |x| >= 1
        if (((uint32_t)(ix - 0x3ff00000) | lo_word(x)) == 0) return hx > 0 ? 0.0 : pi + 2.0 * pio2_lo;
        return (x - x) / (x - x);
    }
    if (ix < 0x3fe00000) {  // |x| < 0.5
        if (ix <= 0x3c600000) return pio2_hi + pio2_lo;
        const double z = x * x;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {  // x < -0.5
        const double z = (1.0 + x) * 0.5;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double s = det_sqrt(z);
        const double r = p / q;
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    const double z = (1.0 - x) * 0.5;  // x > 0.5
    const double s = det_sqrt(z);
    const double df = from_words(hi_word(s), 0);
    const double c = (z - df * df) / (s + df);
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double r = p / q;
    const double w = r * s + c;
    return 2.0 * (df + w);
}
DMSA_DET_HD double det_atan(double x) {
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,  -1.11111104054623557880e-01,
                           9.09088713343650656196e-02,  -7.69187620504482999495e-02, 6.66107313738753120669e-02,  -5.83357013379057348645e-02,
                           4.97687799461593236017e-02,  -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    const int32_t hx = hi_word(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x44100000) {  // |x| >= 2^66 (or NaN)
        if (ix > 0x7ff00000 || (ix == 0x7ff00000 && lo_word(x) != 0)) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3fdc0000) {  // |x| < 0.4375
        if (ix < 0x3e200000) return x;
        id = -1;
    } else {
        x = det_fabs(x);
        if (ix < 0x3ff30000) {      // |x| < 1.1875
            if (ix < 0x3fe60000) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0 * x - 1.0) / (2.0 + x);
            } else {  // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - 1.0) / (x + 1.0);
            }
        } else if (ix < 0x40038000) {  // |x| < 2.4375
            id = 2;
            x = (x - 1.5) / (1.0 + 1.5 * x);
        } else {
            id = 3;
            x = -1.0 / x;
        }
    }
    const double z = x * x;
    const double w = z * z;
    const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const double r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}
// atan2 with the full sign / zero / infinity handling of libm
DMSA_DET_HD double det_atan2(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_o_2 = 1.5707963267948965580E+00, pi_o_4 = 7.8539816339744827900E-01, pi_lo = 1.2246467991473531772E-16;
    const int32_t hx = hi_word(x), hy = hi_word(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const uint32_t lx = lo_word(x), ly = lo_word(y);
    if (((uint32_t)ix | ((lx | (0u - lx)) >> 31)) > 0x7ff00000u || ((uint32_t)iy | ((ly | (0u - ly)) >> 31)) > 0x7ff00000u) return x + y;  // NaN
    if (((uint32_t)(hx - 0x3ff00000) | lx) == 0) return det_atan(y);  // x == 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                // 2 * sign(x) + sign(y)
    if (((uint32_t)iy | ly) == 0) {                                   // y == 0
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi;
            default: return -pi;
        }
    }
    if (((uint32_t)ix | lx) == 0) return hy < 0 ? -pi_o_2 : pi_o_2;  // x == 0
    if (ix == 0x7ff00000) {
        if (iy == 0x7ff00000) {
            switch (m) {
                case 0: return pi_o_4;
                case 1: return -pi_o_4;
                case 2: return 3.0 * pi_o_4;
                default: return -3.0 * pi_o_4;
            }
        }
        switch (m) {
            case 0: return 0.0;
            case 1: return -0.0;
            case 2: return pi;
            default: return -pi;
        }
    }
    if (iy == 0x7ff00000) return hy < 0 ? -pi_o_2 : pi_o_2;
    const int k = (iy - ix) >> 20;
    double z;
    if (k > 60)
        z = pi_o_2 + 0.5 * pi_lo;  // |y / x| > 2^60
    else if (hx < 0 && k < -60)
        z = 0.0;  // |y| / x < -2^60
    else
        z = det_atan(det_fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

}  // namespace dmsa_det
#endif /* DMSA_DETMATH_H */
