// CPU check of sc_add_ones (rambl_amd/csrc/sc_api.cpp): k additions of 1 in x87 long double, bit for bit equal to
// the literal loop of the reference (NonparametricClustering.cpp:195), on values around binade borders and at random.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
extern "C" long double sc_add_ones(long double a, unsigned long k);
static long double literal(long double a, unsigned long k) { for (unsigned long t = 0; t < k; t++) a += 1; return a; }
int main() {
    std::mt19937_64 g(7);
    long bad = 0, n = 0;
    auto check = [&](long double a, unsigned long k) {
        const long double x = sc_add_ones(a, k), y = literal(a, k);
        n++;
        if (std::memcmp(&x, &y, 10) != 0 && !(x != x && y != y)) { if (bad < 10) std::printf("differs: a=%.21Lg k=%lu -> %.21Lg vs %.21Lg\n", a, k, x, y); bad++; }
    };
    for (int e = -70; e <= 20; e++)
        for (int t = 0; t < 400; t++) {
            long double a = ldexpl(1.0L + (long double)(g() >> 1) / 9223372036854775808.0L, e);
            if (t % 7 == 0) a = ldexpl(1.0L, e) - ldexpl((long double)(g() % 5), e - 64);      // just under a power of two
            if (t % 11 == 0) a = floorl(a);
            check(a, g() % 50000);
            check(a, g() % 5);
        }
    check(0.0L, 40000); check(-3.25L, 10); check(1e30L, 1000); check(INFINITY, 5); check(NAN, 3); check(0.999999999999999999L, 3);
    std::printf("%ld cases, %ld differences\n", n, bad);
    return bad ? 1 : 0;
}
