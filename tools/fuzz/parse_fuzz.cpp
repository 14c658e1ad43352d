// Host-parser fuzz: mutate the reference fixtures and parse them under ASan/UBSan.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include "host/host.h"
namespace wp { void set_error(const std::string&) {} }
int main(int argc, char** argv) {
    std::mt19937 rng(12345);
    long ok = 0, bad = 0;
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb"); if (!f) continue;
        std::vector<uint8_t> d; uint8_t buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        const int iters = d.size() > 200000 ? 60 : 400;
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> m = d;
            const int kind = rng() % 4;
            if (kind == 0) { m.resize(rng() % (m.size() + 1)); }
            else if (kind == 1) { for (int k = 0; k < 1 + int(rng() % 8); ++k) m[rng() % m.size()] = uint8_t(rng()); }
            else if (kind == 2) { for (int k = 0; k < 1 + int(rng() % 4); ++k) m[rng() % std::min<size_t>(m.size(), 64)] = uint8_t(rng()); }
            else { size_t p = rng() % m.size(); size_t l = std::min<size_t>(m.size() - p, 1 + rng() % 64); for (size_t k = 0; k < l; ++k) m[p + k] = uint8_t(rng()); }
            wp_parsed p; std::string err;
            int rc;
            try { rc = wp::parse_webp(m.data(), m.size(), p, err); } catch (const std::bad_alloc&) { rc = -6; }
            if (rc == 0) ++ok; else ++bad;
        }
    }
    printf("parsed ok %ld, rejected %ld\n", ok, bad);
    return 0;
}
