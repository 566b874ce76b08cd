#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <chrono>
#include <vector>
extern "C" int pm_legacy_choice(uint32_t *key, int *pos, long n, int k, long trials, int32_t *out);
int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 20000, trials = argc > 2 ? atol(argv[2]) : 8000;
    std::vector<uint32_t> key(624);
    uint32_t s = 5489u; key[0] = s;
    for (int i = 1; i < 624; i++) key[i] = 1812433253u * (key[i-1] ^ (key[i-1] >> 30)) + i;
    int pos = 624;
    std::vector<int32_t> out(trials * 4);
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        int rc = pm_legacy_choice(key.data(), &pos, n, 4, trials, out.data());
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        long cs = 0; for (auto v : out) cs += v;
        printf("rc %d n %ld trials %ld: %.3f s  %.3f ns/output  checksum %ld pos %d\n", rc, n, trials, dt, dt / (n * trials) * 1e9, cs, pos);
    }
}
