// Internal: the object behind the opaque `bj_fri` (shared by fri_prover.hip and prover.hip).
#pragma once
#include "gl.cuh"
#include <vector>
using gl::u64;

struct bj_fri {
    int device = 0;
    size_t cap_size = 0;
    unsigned log_full = 0, log_lde = 0;
    struct Oracle {
        u64 *d_c0 = nullptr, *d_c1 = nullptr;  // leaf sources (oracle 0: the caller's codeword, not owned)
        bool owned = false;
        size_t len = 0;
        unsigned log_e = 0;
        u64 *d_tree = nullptr;
        size_t num_leaves = 0;
        std::vector<u64> cap;
        u64 ch0 = 0, ch1 = 0;
    };
    std::vector<Oracle> oracles;
    u64 *d_last0 = nullptr, *d_last1 = nullptr;  // last folded layer
    size_t last_len = 0;
    std::vector<u64> final_c0, final_c1;
    size_t final_degree = 0;
};

