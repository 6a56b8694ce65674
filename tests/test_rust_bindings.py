"""The Rust side of the boundary is shipped as source (no Rust toolchain in the build image): rust/boojum_hip_sys.rs is GENERATED
from include/boojum_hip.h and must cover every entry point with the right arity; rust/prove_hip.rs (the `prove_hip` sibling of
`prove_cpu_basic` + the BJPF -> `Proof` deserialiser) must be complete source that only uses symbols the header declares."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_rust_bindings", os.path.join(ROOT, "tools", "gen_rust_bindings.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generated_bindings_are_current_and_cover_the_header():
    g = _gen()
    committed = open(os.path.join(ROOT, "rust", "boojum_hip_sys.rs")).read()
    assert g.generate() == committed, "rust/boojum_hip_sys.rs is stale: run python tools/gen_rust_bindings.py"
    h = g.parse_header()
    from era_boojum_amd import binding
    rust_fns = {name: len(ps) for name, _, ps in h["funcs"]}
    assert set(rust_fns) == set(binding._SIGNATURES), set(rust_fns) ^ set(binding._SIGNATURES)
    for name, (_, args) in binding._SIGNATURES.items():        # same arity as the ctypes table the GPU tests call through
        assert rust_fns[name] == len(args), name
    structs = dict(h["structs"])
    assert [f for f, _ in structs["bj_comm"]] == ["rank", "world", "all_gather", "user", "all_gather_stream"]
    assert [f for f, _ in structs["bj_circuit"]][:4] == ["log_n", "num_vars", "num_gp_vars", "num_witness_cols"]
    assert "pub path: [u8; 8]," in committed and "Option<unsafe extern \"C\" fn(" in committed


def test_prove_hip_source_is_complete():
    src = open(os.path.join(ROOT, "rust", "prove_hip.rs")).read()
    assert "unimplemented!" not in src and "todo!" not in src
    for needed in ("pub fn prove_hip<", "pub fn hip_setup<", "pub fn proof_from_bjpf<", "fn from_capture(", "bj_setup_create_sharded(", "bj_prove(",
                   "bj_proof_serialize(", "queries_per_fri_repetition", "fri_intermediate_oracles_caps", "pow_challenge"):
        assert needed in src, needed
    declared = set(re.findall(r"pub fn (bj_[a-z0-9_]+)", open(os.path.join(ROOT, "rust", "boojum_hip_sys.rs")).read()))
    used = set(re.findall(r"\b(bj_[a-z0-9_]+)\(", src))
    assert used <= declared, used - declared
    # the deserialiser reads the header fields the serialiser writes (csrc/prover.hip) in the same order as proof_format.py
    from era_boojum_amd import proof_format
    assert "0x424A_5046" in src and proof_format.MAGIC == 0x424A5046


def test_array_parameters_decay_to_pointers_and_array_fields_stay_arrays():
    """C adjusts a parameter `uint64_t fp[2]` to `uint64_t *fp`: the binding must pass one address, not 16 bytes by value
    (ADVICE round 3); a struct field `unsigned char path[8]` stays an inline array."""
    g = _gen()
    assert g.parse_param("uint64_t fp[2]", {}, set(), decay=True) == ("fp", "*mut u64")
    assert g.parse_param("const uint64_t roots[4]", {}, set(), decay=True) == ("roots", "*const u64")
    assert g.parse_param("uint32_t v[]", {}, set(), decay=True) == ("v", "*mut u32")
    assert g.parse_param("unsigned char path[8]", {}, set()) == ("path", "[u8; 8]")
    funcs = {name: ps for name, _, ps in g.parse_header()["funcs"]}
    assert dict(funcs["bj_gate_program_canonical_info"])["fp"] == "*mut u64"
    for name, ps in funcs.items():
        assert not any(t.startswith("[") for _, t in ps), name        # no by-value array in any extern "C" signature
