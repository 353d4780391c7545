"""The pipelined caller (zkir_amd/service.py): proofs of independent runs, produced with host threads overlapping the GPU, are
the same words the sequential path produces, come back in job order, and verify."""
import numpy as np
import pytest

from oracle import stark_api as so
from zkir_amd import runtime as rt, spec

pytestmark = pytest.mark.gpu


def test_prove_many_matches_sequential_and_verifies():
    from zkir_amd import pipeline as pl, service, stark
    k = 10
    cfg = rt.VMConfig(max_cycles=1 << k, enable_execution_trace=True)
    ragged = rt.VMConfig(max_cycles=(1 << k) - 123, enable_execution_trace=True)             # pads to the same 2^k
    jobs = [(spec.fib_endless_program().to_bytes(), [], cfg), (spec.sha256_chain_program().to_bytes(), [], ragged)] * 4
    rep = service.prove_many(jobs, k, producers=3)
    assert rep.runs == 8 and len(rep.proofs) == 8 and rep.rows == 4 * ((2 << k) - 123)
    ctx = stark.StarkContext(k)
    for (blob, inputs, c), proof in zip(jobs[:2], rep.proofs[:2]):
        log = rt.interpret(blob, inputs, c)
        ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
        pub = rt.public_inputs(log, blob, inputs)
        assert np.array_equal(stark.prove(ctx, tr, pub), proof)
        assert rt.verify(proof, pub) == 0
    ctx.close()
    assert all(so.verify(p) == 0 and rt.verify(p) == 0 for p in rep.proofs)
    assert np.array_equal(rep.proofs[0], rep.proofs[2]) and not np.array_equal(rep.proofs[0], rep.proofs[1])     # job order kept


def test_prove_many_reports_bad_jobs():
    from zkir_amd import service
    k = 8
    short = rt.VMConfig(max_cycles=100, enable_execution_trace=True)             # 100 rows pad to 2^7, the context is for 2^8
    with pytest.raises(rt.RuntimeError):
        service.prove_many([(spec.fib_endless_program().to_bytes(), [], short)], k, producers=1)


def test_commit_many_roots_match_sequential_and_keep_job_order():
    """commit_only: the lag-one pipeline (the next run's kernels are queued before the previous root is read back) returns, per job and
    in job order, the root the sequential commit gives and the oracle's."""
    from zkir_amd import pipeline as pl, service, stark
    k = 9
    cfg = rt.VMConfig(max_cycles=1 << k, enable_execution_trace=True)
    ragged = rt.VMConfig(max_cycles=(1 << k) - 77, enable_execution_trace=True)
    jobs = [(spec.fib_endless_program().to_bytes(), [], cfg), (spec.sha256_chain_program().to_bytes(), [], ragged), (spec.fib_endless_program().to_bytes(), [], ragged)] * 3
    rep = service.prove_many(jobs, k, producers=3, commit_only=True)
    assert len(rep.proofs) == 9
    ctx = stark.StarkContext(k)
    for (blob, inputs, c), root in zip(jobs[:3], rep.proofs[:3]):
        log = rt.interpret(blob, inputs, c)
        ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
        want = stark.commit_trace(ctx, tr)[0]
        assert np.array_equal(root, want)
        r = so.commit_root_of_run(blob, inputs, c.max_cycles) if hasattr(so, "commit_root_of_run") else None
        assert r is None or np.array_equal(root, r)
    ctx.close()
    for i in range(3):
        assert np.array_equal(rep.proofs[i], rep.proofs[i + 3]) and np.array_equal(rep.proofs[i], rep.proofs[i + 6])
    assert not np.array_equal(rep.proofs[0], rep.proofs[1]) and not np.array_equal(rep.proofs[0], rep.proofs[2])
