/* Exhaustive check (all 2^32 float bit patterns) that
 *     truncf(x + copysignf(pred(0.5f), x))  ==  roundf(x)          bit for bit,
 * the 3-instruction form of roundf the plane-sweep projection uses
 * (raynet_amd/csrc/raynet_kernels.h: round_half_away).
 *   gcc -O2 -fopenmp -fno-fast-math tools/verify_round_trick.c -lm -o /tmp/verify_round && /tmp/verify_round
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
    long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long i = 0; i < (1LL << 32); i++) {
        const uint32_t u = (uint32_t)i;
        float x;
        memcpy(&x, &u, 4);
        const float a = roundf(x);
        volatile float sum = x + copysignf(0x1.fffffep-2f, x);
        const float b = truncf(sum);
        uint32_t ua, ub;
        memcpy(&ua, &a, 4);
        memcpy(&ub, &b, 4);
        if (isnan(a) ? !isnan(b) : ua != ub) bad++;
    }
    printf("%lld of 4294967296 inputs differ\n", bad);
    return bad != 0;
}
