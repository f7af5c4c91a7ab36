// host_tsan.cpp -- TEST INFRASTRUCTURE: the host-only pieces of the product (dq_bz2.h, dq_bspatch.h, dq_alpha_code.h)
// called from several threads at once under -fsanitize=thread (tests/test_patch_hardening.py): the library promises
// re-entrant entry points, and these pieces run on the callers' threads (three framing threads per Diff.Create,
// Patch.Apply on whatever thread calls it, the codeword table of a coded round 0 on the sorting thread).
//   g++ -O1 -g -std=c++17 -fsanitize=thread tests/native/host_tsan.cpp -o host_tsan -pthread
#include <atomic>
#include <cstdio>
#include <numeric>
#include <thread>

#include "../../deltaq_amd/csrc/dq_alpha_code.h"
#include "../../deltaq_amd/csrc/dq_bspatch.h"

using namespace dq;

static int doubling_sorter(const uint8_t *t, int64_t n, int32_t *sa)
{
    std::vector<int64_t> rank((size_t)n), tmp((size_t)n);
    std::iota(sa, sa + n, 0);
    for (int64_t i = 0; i < n; ++i) rank[(size_t)i] = t[i];
    for (int64_t h = 1;; h *= 2) {
        auto key2 = [&](int32_t a) { return a + h < n ? rank[(size_t)(a + h)] : (int64_t)-1; };
        auto less = [&](int32_t a, int32_t b) {
            return rank[(size_t)a] != rank[(size_t)b] ? rank[(size_t)a] < rank[(size_t)b] : key2(a) < key2(b);
        };
        std::sort(sa, sa + n, less);
        tmp[(size_t)sa[0]] = 0;
        for (int64_t i = 1; i < n; ++i) tmp[(size_t)sa[i]] = tmp[(size_t)sa[i - 1]] + (less(sa[i - 1], sa[i]) ? 1 : 0);
        rank = tmp;
        if (n == 0 || rank[(size_t)sa[n - 1]] == n - 1) break;
    }
    return 0;
}

static int g_bad = 0;

static void worker(int id)
{
    uint64_t x = 1000 + (uint64_t)id;
    auto rnd = [&]() { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                       z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (int it = 0; it < 6; ++it) {
        // a text-like buffer, its codeword table, a bzip2 round trip, a patch built from it and applied
        std::vector<uint8_t> old(20000 + 1000 * id), nw;
        for (auto &b : old) b = (uint8_t)("etaoin shrdlu"[rnd() % 13]);
        int64_t hist[256] = {0};
        for (uint8_t b : old) hist[b]++;
        AlphaCode code;
        if (!build_alpha_code(hist, &code) || code.sigma < 2) __atomic_fetch_add(&g_bad, 1, __ATOMIC_RELAXED);
        std::vector<uint8_t> z, back;
        if (bz2::bz2_compress(old.data(), old.size(), z, doubling_sorter, 9) != 0 ||
            bz2::bz2_decompress(z.data(), z.size(), back) != 0 || back != old)
            __atomic_fetch_add(&g_bad, 1, __ATOMIC_RELAXED);
        nw = old;
        for (int e = 0; e < 20; ++e) nw[rnd() % nw.size()] ^= 1;
        std::vector<uint8_t> diff(old.size()), ctrl(24), zc, zd, ze, none;
        for (size_t i = 0; i < old.size(); ++i) diff[i] = (uint8_t)(nw[i] - old[i]);
        bsdiff::write_packed_long(&ctrl[0], (int64_t)old.size());
        bsdiff::write_packed_long(&ctrl[8], 0);
        bsdiff::write_packed_long(&ctrl[16], 0);
        bz2::bz2_compress(ctrl.data(), ctrl.size(), zc, doubling_sorter, 9);
        bz2::bz2_compress(diff.data(), diff.size(), zd, doubling_sorter, 9);
        bz2::bz2_compress(none.data(), 0, ze, doubling_sorter, 9);
        std::vector<uint8_t> p(32);
        bsdiff::write_packed_long(&p[0], bsdiff::kSignature);
        bsdiff::write_packed_long(&p[8], (int64_t)zc.size());
        bsdiff::write_packed_long(&p[16], (int64_t)zd.size());
        bsdiff::write_packed_long(&p[24], (int64_t)nw.size());
        p.insert(p.end(), zc.begin(), zc.end());
        p.insert(p.end(), zd.begin(), zd.end());
        p.insert(p.end(), ze.begin(), ze.end());
        std::vector<uint8_t> out(nw.size());
        int64_t len = 0;
        if (bsdiff::apply_patch(old.data(), (int64_t)old.size(), p.data(), (int64_t)p.size(), out.data(), (int64_t)out.size(), &len) != 0 ||
            out != nw)
            __atomic_fetch_add(&g_bad, 1, __ATOMIC_RELAXED);
    }
}

// The hand-over of dq_diff.hip's PatchFramer, on its own: a producer appends to a stream whose capacity is reserved and
// publishes the length that is final (release); a follower reads the state, then the length (acquire), and feeds the
// encoder, whose full blocks go to encoder threads of their own; the result must be the stream framed at once.
static void follow_a_growing_stream(int id)
{
    uint64_t x = 77 + (uint64_t)id;
    auto rnd = [&]() { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                       z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    const size_t total = 260000 + 5000 * (size_t)id;          // level 1: three blocks of 99 981 coded bytes
    std::vector<uint8_t> stream;
    stream.reserve(total);
    const uint8_t *base = stream.data();
    std::atomic<size_t> final_len{0};
    std::atomic<int> state{0};
    bz2::StreamEncoder enc(doubling_sorter, 1);
    std::thread follower([&] {
        for (;;) {
            const int s = state.load(std::memory_order_acquire);
            const size_t upto = final_len.load(std::memory_order_acquire);
            enc.feed(base, upto, s == 1);
            if (s == 1) return;
            std::this_thread::yield();
        }
    });
    while (stream.size() < total) {                              // pieces of text, runs and noise
        const size_t piece = std::min<size_t>(total - stream.size(), 1 + rnd() % 3000);
        const int kind = (int)(rnd() % 3);
        const uint8_t c = (uint8_t)(rnd() % 4);
        for (size_t k = 0; k < piece; ++k)
            stream.push_back(kind == 0 ? c : kind == 1 ? (uint8_t)("etaoin shrdlu"[rnd() % 13]) : (uint8_t)rnd());
        final_len.store(stream.size(), std::memory_order_release);
    }
    state.store(1, std::memory_order_release);
    follower.join();
    std::vector<uint8_t> fed, once, back;
    if (enc.finish(fed) != 0 || bz2::bz2_compress(stream.data(), stream.size(), once, doubling_sorter, 1) != 0 || fed != once ||
        bz2::bz2_decompress(fed.data(), fed.size(), back) != 0 || back != stream)
        __atomic_fetch_add(&g_bad, 1, __ATOMIC_RELAXED);
}

int main()
{
    std::vector<std::thread> ts;
    for (int i = 0; i < 6; ++i) ts.emplace_back(worker, i);
    for (int i = 0; i < 2; ++i) ts.emplace_back(follow_a_growing_stream, i);
    for (auto &t : ts) t.join();
    if (g_bad) { fprintf(stderr, "%d wrong results\n", g_bad); return 1; }
    printf("ok\n");
    return 0;
}
