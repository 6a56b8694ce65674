/* A whole proof from a plain C99 host, one process per GPU, no Python anywhere: the S1 boundary (bj_setup_create_sharded +
 * bj_prove) with the library's own RCCL transport (bj_rccl_unique_id / bj_comm_rccl_create).
 *
 *   host_sharded <rank> <world> <id_file>        rank 0 writes the RCCL unique id to <id_file>, the others read it
 *   host_sharded                                 = rank 0 of a world of 1
 *
 * The circuit is built here from the prover's input types as the reference defines them (SetupBaseStorage / WitnessSet,
 * src/cs/implementations/polynomial_storage.rs:48-75, witness.rs:21-27): 8 variable columns, one evaluator
 * (FmaGateInBaseFieldWithoutConstant, two repetitions per row, constants q and l shared by the row), no lookups, no copy
 * constraints (sigma_c(x) = k_c * x with the reference's non-residues, utils.rs:636-688).  GPU `rank` proves with the cosets
 * [rank * 8 / world, (rank + 1) * 8 / world) of every LDE; every rank ends with the same proof and prints its fingerprint.
 * With world = 1 the proof is also compared with the one of the unsharded entry point, and a broken witness must be refused.
 * Exit code 0 = ok, 2 = no GPU (there is no CPU path), 1 = failure.
 *   gcc -std=c99 -Iinclude examples/host_sharded.c -o host_sharded -Lera_boojum_amd -lboojum_hip -Wl,-rpath,$PWD/era_boojum_amd
 */
#include "boojum_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define P 0xFFFFFFFF00000001ULL
__extension__ typedef unsigned __int128 u128;   /* gcc / clang extension, host-side field helper only */
static uint64_t fmul(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P); }
static uint64_t fadd(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % P); }
static uint64_t fpow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    for (; e; e >>= 1, a = fmul(a, a))
        if (e & 1) r = fmul(r, a);
    return r;
}
static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static uint64_t fnv(const uint64_t *w, size_t n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; i++)
        for (int b = 0; b < 8; b++) h = (h ^ ((w[i] >> (8 * b)) & 0xFF)) * 0x100000001b3ULL;
    return h;
}
#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != 0) {                                                                              \
            fprintf(stderr, "%s -> %s (%s)\n", #call, bj_status_string(rc_), bj_last_error(ctx));   \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char **argv) {
    enum { LOG_N = 10, V = 8, NC = 2, Q = 4, FRI_LDE = 8, CAP = 16, SECURITY = 30 };
    const size_t n = (size_t)1 << LOG_N;
    const unsigned rank = argc > 2 ? (unsigned)atoi(argv[1]) : 0, world = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
    const char *id_file = argc > 3 ? argv[3] : NULL;
    bj_ctx *ctx = NULL;
    if (bj_device_count() <= 0) {
        fprintf(stderr, "no HIP device: this library has no CPU path\n");
        return 2;
    }
    if (bj_ctx_create((int)(rank % (unsigned)bj_device_count()), &ctx) != 0) return 2;

    /* ---- transport: the unique id goes through a file here (MPI / a socket / a Rust channel in a real host) ---- */
    unsigned char id[BJ_RCCL_UNIQUE_ID_BYTES];
    if (rank == 0) {
        CHECK(bj_rccl_unique_id(id));
        if (id_file) {
            FILE *f = fopen(id_file, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 1;
            fclose(f);
        }
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 6000 && !f; tries++) {   /* wait for rank 0 */
            f = id_file ? fopen(id_file, "rb") : NULL;
            if (f && fread(id, 1, sizeof id, f) != sizeof id) {
                fclose(f);
                f = NULL;
            }
            if (!f) {
                const clock_t t0 = clock();
                while ((double)(clock() - t0) / CLOCKS_PER_SEC < 0.01) { }
            }
        }
        if (!f) return 1;
        fclose(f);
    }
    bj_comm comm;
    CHECK(bj_comm_rccl_create(ctx, id, rank, world, &comm));

    /* ---- the circuit: SetupBaseStorage + VerificationKey parameters as plain arrays ---- */
    uint64_t *vars = (uint64_t *)malloc(V * n * 8), *sig = (uint64_t *)malloc(V * n * 8), *con = (uint64_t *)malloc(NC * n * 8);
    uint64_t seed = 7;
    for (size_t r = 0; r < n; r++) {
        const uint64_t qc = splitmix(&seed) % P, lc = splitmix(&seed) % P;
        con[0 * n + r] = qc;
        con[1 * n + r] = lc;
        for (int rep = 0; rep < 2; rep++) {   /* q * a * b + l * c - d = 0 */
            const uint64_t a = splitmix(&seed) % P, b = splitmix(&seed) % P, c = splitmix(&seed) % P;
            uint64_t *w = vars + (size_t)(4 * rep) * n + r;
            w[0] = a; w[n] = b; w[2 * n] = c; w[3 * n] = fadd(fmul(qc, fmul(a, b)), fmul(lc, c));
        }
    }
    uint64_t non_res[V];   /* make_non_residues (utils.rs:636-688): successive quadratic non-residues in distinct cosets */
    {
        uint64_t seen[V];
        unsigned have = 1, n_seen = 0;
        non_res[0] = 1;
        for (uint64_t cur = 2; have < V; cur++) {
            if (fpow(cur, (P - 1) / 2) != P - 1) continue;
            const uint64_t t = fpow(cur, n);
            int dup = t == 1;
            for (unsigned k = 0; k < n_seen; k++) dup |= seen[k] == t;
            if (dup) continue;
            seen[n_seen++] = t;
            non_res[have++] = cur;
        }
    }
    uint64_t omega = 0x185629dcda58878cULL;   /* radix_2_subgroup_generator; omega_n = it^(2^(32 - log n)) (utils.rs:13-28) */
    for (int i = LOG_N; i < 32; i++) omega = fmul(omega, omega);
    for (unsigned c = 0; c < V; c++) {         /* no copy constraints: sigma = identity permutation, k_c * omega^r */
        uint64_t x = non_res[c];
        for (size_t r = 0; r < n; r++, x = fmul(x, omega)) sig[(size_t)c * n + r] = x;
    }
    bj_gate_desc gate;
    memset(&gate, 0, sizeof gate);
    gate.kind = BJ_GATE_FMA_NO_CONSTANT;
    gate.path_len = 0;            /* a single evaluator: no selector */
    gate.num_repetitions = 2;
    gate.var_stride = 4;
    gate.const_stride = 0;
    gate.num_terms = 1;
    bj_circuit circuit;
    memset(&circuit, 0, sizeof circuit);
    circuit.log_n = LOG_N;
    circuit.num_vars = V;
    circuit.num_gp_vars = V;
    circuit.num_constant_cols = NC;
    circuit.quotient_degree = Q;
    circuit.num_gates = 1;
    circuit.gates = &gate;
    circuit.non_residues = non_res;
    bj_proof_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.fri_lde_factor = FRI_LDE;
    cfg.cap_size = CAP;
    cfg.security_level = SECURITY;
    cfg.transcript = BJ_TRANSCRIPT_POSEIDON2;
    cfg.tree_hasher = BJ_HASHER_POSEIDON2;

    /* ---- prove: this GPU's cosets, cap fragments / FRI layer / query openings exchanged by the library over RCCL ---- */
    bj_setup *setup = NULL;
    bj_proof *proof = NULL;
    CHECK(bj_setup_create_sharded(ctx, &circuit, sig, con, NULL, &cfg, &comm, &setup));
    CHECK(bj_prove(ctx, setup, vars, NULL, NULL, &proof));
    const size_t words = bj_proof_size_u64(proof);
    uint64_t *buf = (uint64_t *)malloc(words * 8);
    CHECK(bj_proof_serialize(proof, buf));
    size_t calls = 0, bytes = 0;
    (void)bj_comm_rccl_stats(&comm, &calls, &bytes);
    printf("rank %u of %u: proof of %zu bytes, fingerprint %016llx, %zu collectives, %zu bytes received\n", rank, world, words * 8,
           (unsigned long long)fnv(buf, words), calls, bytes);
    bj_proof_destroy(proof);

    if (world == 1) {   /* the unsharded entry point gives the same proof, and a wrong witness is refused */
        bj_setup *plain = NULL;
        CHECK(bj_setup_create(ctx, &circuit, sig, con, NULL, &cfg, &plain));
        CHECK(bj_prove(ctx, plain, vars, NULL, NULL, &proof));
        uint64_t *buf2 = (uint64_t *)malloc(words * 8);
        if (bj_proof_size_u64(proof) != words) return 1;
        CHECK(bj_proof_serialize(proof, buf2));
        if (memcmp(buf, buf2, words * 8) != 0) {
            fprintf(stderr, "sharded-entry proof differs from the plain one\n");
            return 1;
        }
        bj_proof_destroy(proof);
        {   /* the pipelined drop-in call: five witnesses (here the same one), two proofs in flight from this one thread */
            bj_ticket *prev = NULL, *cur = NULL;
            int k, same = 1;
            CHECK(bj_prove_async(ctx, plain, vars, NULL, NULL, &prev));
            for (k = 1; k <= 5; k++) {
                if (k < 5) CHECK(bj_prove_async(ctx, plain, vars, NULL, NULL, &cur));
                CHECK(bj_proof_wait(prev, &proof));
                if (bj_proof_size_u64(proof) != words) return 1;
                CHECK(bj_proof_serialize(proof, buf2));
                same = same && memcmp(buf, buf2, words * 8) == 0;
                bj_proof_destroy(proof);
                prev = cur;
            }
            if (!same) {
                fprintf(stderr, "a pipelined proof differs from the serial one\n");
                return 1;
            }
            printf("pipelined: 5 proofs through bj_prove_async / bj_proof_wait, all identical to the serial one\n");
        }
        vars[3 * n + 5] = fadd(vars[3 * n + 5], 1);
        proof = NULL;
        const int rc = bj_prove(ctx, plain, vars, NULL, NULL, &proof);
        if (rc == 0 || !strstr(bj_last_error(ctx), "not satisfied")) {
            fprintf(stderr, "a broken witness was not refused (%d: %s)\n", rc, bj_last_error(ctx));
            return 1;
        }
        printf("plain entry point: identical proof; broken witness refused: %s\n", bj_last_error(ctx));
        bj_setup_destroy(plain);
        free(buf2);
    }
    bj_setup_destroy(setup);
    bj_comm_rccl_destroy(&comm);
    bj_ctx_destroy(ctx);
    free(buf); free(vars); free(sig); free(con);
    printf("ok\n");
    return 0;
}
