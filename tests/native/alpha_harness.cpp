// Test harness: the alphabetic code construction and the key builder of the coded round 0, compiled for the host.
#include "../../deltaq_amd/csrc/dq_alpha_code.h"
#include "../../deltaq_amd/csrc/dq_coded_keys.h"

#include <cstring>

extern "C" int t_alpha_code(const int64_t *hist, uint16_t *tab, double *avg)
{
    dq::AlphaCode c;
    const bool ok = dq::build_alpha_code(hist, &c);
    memcpy(tab, c.tab, sizeof(c.tab));
    *avg = c.avg_len;
    return ok ? c.sigma : -1;
}

// keys of all suffixes of text[0, n) the way the device builds them (text must be followed by >= 24 zero bytes)
extern "C" void t_coded_keys(const uint8_t *text, int64_t n, const uint16_t *tab, uint64_t *keys)
{
    for (int64_t i0 = 0; i0 < n; i0 += 4) {
        uint32_t w[5];
        memcpy(w, text + i0, 20);
        uint64_t k[4];
        dq::coded_keys4(w, tab, k);
        for (int c = 0; c < 4 && i0 + c < n; ++c) keys[i0 + c] = k[c];
    }
}
