"""Worker of tests/test_gpu_sharded_prover.py: one rank of a coset-sharded proof (launched by torch.distributed.run).

Several ranks may share one GPU (backend gloo): the sharding logic is the same, only the transport differs.
Writes this rank's proof to <out>/proof_<rank>.npy."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, log_n, fri, cap, sec = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import torch
    import torch.distributed as dist
    import era_boojum_amd as E
    from era_boojum_amd import synthetic as S
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = rank % ndev
    torch.cuda.set_device(dev)
    backend = "nccl" if ndev >= world else "gloo"
    dist.init_process_group(backend, rank=rank, world_size=world)
    if log_n == 0:      # the real SHA-256 circuit of a 100-byte message (2^14 rows)
        from era_boojum_amd import sha256_circuit as SHA
        circuit = SHA.sha256_circuit(SHA.bench_message(100, seed=7))
    elif log_n < 0:     # the golden proof's circuit class (155 columns, quotient degree 8, Poseidon2 flattened gate, a specialized gate)
        circuit = S.recursion_like_circuit(-log_n, seed=7)
    else:
        circuit = S.sha_shaped_circuit(log_n, seed=7, table_bits=4 if log_n >= 14 else 2,
                                       **({"boolean_columns": 2, "specialized_constant_columns": 3} if log_n == 12 else
                                    {"table_id_as_variable": True, "boolean_columns": 2} if log_n == 13 else {}))
    ctx = E.Context(dev)
    comm = base = E.TorchComm(ctx)
    if os.environ.get("BJ_COMM_BULK") == "peer":   # full-mesh peer copies for everything of at least BJ_COMM_BULK_MIN bytes per rank
        comm = E.PeerComm(ctx, base, bulk_threshold_bytes=int(os.environ.get("BJ_COMM_BULK_MIN", str(1 << 20))))
    setup = E.ProverSetup(ctx, circuit, fri, cap, sec, comm=comm)
    proof, _ = setup.prove()
    np.save(os.path.join(out_dir, "proof_%d.npy" % rank), proof)
    np.save(os.path.join(out_dir, "cap_%d.npy" % rank), setup.cap())
    if rank == 0:
        with open(os.path.join(out_dir, "comm.txt"), "w") as f:
            f.write("%d %d\n" % (comm.calls, comm.bytes))
    if comm is not base:
        import json
        with open(os.path.join(out_dir, "peer_%d.json" % rank), "w") as f:
            json.dump(comm.stats(), f)
        proof2, _ = setup.prove()          # a second proof reuses the mapped allocations
        assert np.array_equal(proof2, proof)
    setup.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
