// Generates tests/golden/std_shuffle.json: what libstdc++'s std::shuffle does with a default-constructed
// std::default_random_engine (= minstd_rand0) -- the selection of contact points in ContactGrasp's loader
// (grasp_analyzer.cpp:346-365 shuffles the contact list, keeps 500, then shuffles the non-contact list with the SAME
// engine and keeps 2000). This is the C++ standard library, not reference code.
//   g++ -O1 -o /tmp/gen_std_shuffle tests/golden/gen_std_shuffle.cpp && /tmp/gen_std_shuffle > tests/golden/std_shuffle.json
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
int main() {
    const size_t sizes[][2] = {{7, 10}, {1, 2}, {1200, 5000}, {46341, 46342}, {70000, 3}};
    printf("[");
    for (size_t c = 0; c < sizeof(sizes) / sizeof(sizes[0]); ++c) {
        auto rng = std::default_random_engine{};
        printf("%s{\"sizes\": [%zu, %zu], \"first\": [", c ? ",\n " : "", sizes[c][0], sizes[c][1]);
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<size_t> idx(sizes[c][pass]);
            std::iota(idx.begin(), idx.end(), 0);
            std::shuffle(idx.begin(), idx.end(), rng);
            const size_t keep = std::min<size_t>(idx.size(), 40);
            for (size_t i = 0; i < keep; ++i) printf("%s%zu", i ? ", " : "", idx[i]);
            unsigned long long h = 1469598103934665603ull;
            for (size_t v : idx) { h ^= v; h *= 1099511628211ull; }
            if (pass == 0) printf("], \"hash0\": \"%llu\", \"second\": [", h);
            else printf("], \"hash1\": \"%llu\"}", h);
        }
    }
    printf("]\n");
    return 0;
}
