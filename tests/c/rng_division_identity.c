/* Exhaustive proof (all 2^32 inputs) of the identity the HIP path's rng_get relies on:
 *     (double) u / 4294967295.0  ==  fma((double) u, 2^-64 + 2^-96, (double) u * 2^-32)        bit for bit,
 * i.e. FloatProducer.toDouble (Float.fs:29) without a division.  Also shows the two-term form is NOT exact.
 * Exit status 0 iff the three-term form never differs.  Build: gcc -O2 -ffp-contract=off rng_division_identity.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
    const double c = 0x1p-64 + 0x1p-96;
    uint64_t bad3 = 0, bad2 = 0;
    for (uint64_t u = 0; u < (1ull << 32); ++u) {
        const double x = (double) (uint32_t) u;
        const double ref = x / 4294967295.0;
        const double q = x * 0x1p-32;
        const double a = fma(x, c, q);
        const double b = fma(q, 0x1p-32, q);
        if (memcmp(&a, &ref, 8) != 0) bad3++;
        if (memcmp(&b, &ref, 8) != 0) bad2++;
    }
    printf("three_term_mismatches=%llu two_term_mismatches=%llu\n", (unsigned long long) bad3, (unsigned long long) bad2);
    return bad3 == 0 ? 0 : 1;
}
