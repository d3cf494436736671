// SimplexNoise constructor tables on the host (reference: js/simplex-noise.js:8-14, js/rng.js:3-6).
#include <cmath>
#include <cstdint>
#include "host_util.h"
#include "noise.h"

namespace wo {

void noise_tables(double seed, uint8_t* perm512, uint8_t* pm12_512) {
    ParkMiller rng(seed);
    uint8_t p[256];
    for (int i = 0; i < 256; ++i) p[i] = (uint8_t)i;
    for (int i = 255; i > 0; --i) {
        const int j = (int)std::floor(rng.next() * (double)(i + 1));
        const uint8_t t = p[i]; p[i] = p[j]; p[j] = t;
    }
    for (int i = 0; i < 512; ++i) { perm512[i] = p[i & 255]; pm12_512[i] = (uint8_t)(perm512[i] % 12); }
}

}  // namespace wo
