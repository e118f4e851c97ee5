// Host-side check of the multiply-shift divider the kernels' prologues use (rten_make_div / rten_div, rten_amd/csrc/internal.h): the SAME functions,
// compiled for the host by hipcc (no GPU needed), against the C++ `/` operator over every divisor up to 4096 and the edges of every range.
#include "../../rten_amd/csrc/internal.h"

#include <cstdio>
#include <initializer_list>

int main() {
    long long bad = 0, checked = 0;
    auto check = [&](long long n, long long d, const RtenDiv &m) {
        checked++;
        if (rten_div((int)n, m) != (int)(n / d)) {
            if (bad++ < 8) std::printf("mismatch: %lld / %lld -> %d\n", n, d, rten_div((int)n, m));
        }
    };
    for (long long d = 1; d <= 4096; d++) {
        for (long long n_max : {1LL, 7LL, 100LL, 4097LL, 100352LL, 401408LL, 3211264LL, 33554432LL, 2147483647LL}) {
            const RtenDiv m = rten_make_div(n_max, d);
            long long lim = 2;
            while (lim <= n_max && lim < (1LL << 31)) lim <<= 1; // the range the divider promises: [0, 2^L)
            for (long long q = 0; q < 64; q++) { // multiples of d and their predecessors, from both ends of the range
                const long long lo = q * d, hi = ((lim - 1) / d - q) * d;
                for (long long n : {lo, lo - 1, lo + 1, hi, hi - 1, hi + d - 1}) if (n >= 0 && n < lim) check(n, d, m);
            }
            for (long long k = 0; k < 512; k++) check((k * 2654435761LL) % lim, d, m); // scattered values
            check(lim - 1, d, m);
        }
    }
    for (long long d : {65536LL, 1000003LL, 2147483647LL, 4294967296LL}) { // divisors up to (and past) the range itself
        const RtenDiv m = rten_make_div(2147483647LL, d);
        for (long long n : {0LL, 1LL, 65535LL, 65536LL, 1000002LL, 1000003LL, 2147483646LL, 2147483647LL}) check(n, d, m);
    }
    const RtenDiv z = rten_make_div(100, 0); // a divisor that is not set (e.g. OW of a plain GEMM) behaves like 1
    check(57, 1, z);
    std::printf("%lld quotients checked, %lld mismatches\n", checked, bad);
    return bad ? 1 : 0;
}
