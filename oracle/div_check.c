/* TEST INFRASTRUCTURE.  Exhaustive check of the reciprocal-with-correction division the HIP ray loop uses
 * (parc_amd/csrc/parc_env.hip: cell_index_rcp) against IEEE division: for every float x with lo <= |x| <= hi,
 *   q0 = x*rd;  r = fma(-d, q0, x);  q = fma(r, rd, q0)   must equal   x / d      (rd = correctly rounded 1/d).
 * Returns the number of x where the two differ (0 expected); first_bad receives one counter-example. */
#include <math.h>
#include <stdint.h>
#include <string.h>

long long parc_check_rcp_division(float d, float lo, float hi, float *first_bad) {
    const float rd = (float)(1.0L / (long double)d);
    uint32_t a, b;
    memcpy(&a, &lo, 4); memcpy(&b, &hi, 4);
    long long bad = 0;
    for (int sign = 0; sign < 2; ++sign) {
        for (uint32_t u = a; u <= b; ++u) {
            uint32_t v = u | (sign ? 0x80000000u : 0u);
            float x; memcpy(&x, &v, 4);
            const float want = x / d;
            const float q0 = x * rd;
            const float r = fmaf(-d, q0, x);
            const float got = fmaf(r, rd, q0);
            if (!(got == want)) { if (!bad && first_bad) *first_bad = x; ++bad; }
        }
    }
    return bad;
}
