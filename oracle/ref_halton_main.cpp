// ORACLE -- test infrastructure only.
// Driver around the reference's vendored external/halton_sampler.h (compiled from where it
// lies under /root/reference via -I; no reference source is copied into this repo).
// Prints HS::Halton_sampler::sample(dim, index) after init_faure() as raw float32:
//   halton_ref <n_dims> <index...>     -> n_dims * n_index floats on stdout, dim-major.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "halton_sampler.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    unsigned n_dims = std::strtoul(argv[1], nullptr, 10);
    std::vector<unsigned> idx;
    for (int i = 2; i < argc; i++) idx.push_back(std::strtoul(argv[i], nullptr, 10));
    HS::Halton_sampler hs;
    hs.init_faure();
    std::vector<float> out;
    for (unsigned d = 0; d < n_dims; d++)
        for (unsigned i : idx) out.push_back(hs.sample(d, i));
    std::fwrite(out.data(), sizeof(float), out.size(), stdout);
    return 0;
}
