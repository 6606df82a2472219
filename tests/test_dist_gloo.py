"""CPU, world_size 2 over gloo: the data-parallel inference path (batch sharding +
token gather, doc2tex_amd/dist.py).  The per-rank decoder here is the CPU oracle
(the engine needs a GPU); the host logic under test is identical on RCCL."""
import json
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLD, ROOT, oracle_state_dict
from doc2tex_amd import dist as ddist
from doc2tex_amd import synth


def test_shard_bounds_cover_batch():
    for n in [0, 1, 5, 64, 127]:
        for w in [1, 2, 3, 8]:
            spans = [ddist.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _worker(rank, world, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import restatement as R
    torch.set_num_threads(2)
    with open(os.path.join(GOLD, "manifests.json")) as f:
        man = json.load(f)
    cfg, sd = oracle_state_dict("T2", man["T2"], 40, end_bias=1.81)  # rows end at different steps
    img = synth.synth_images(5, 48, 64, seed=1001)  # 5 rows -> shards of 3 and 2

    def decode(x):
        with torch.no_grad():
            text = torch.full((x.shape[0], 1), R.GO, dtype=torch.long)
            return R.forward(cfg, sd, x, text, is_test=True)[0]

    toks = ddist.decode_sharded(decode, img)
    if rank == 0:
        q.put((toks, decode(img)))
    dist.barrier()
    dist.destroy_process_group()


RENDEZVOUS_FAILED = 75  # exit code of a rank that could not join the process group (the only failure that is retried)


def _rank_main(worker, rank, world, store, q, logdir):
    """Entry point of a spawned rank: its stderr (with faulthandler's tracebacks on a fatal signal) goes to a per-rank
    file that the parent attaches to the assertion message; failing to JOIN the group exits with RENDEZVOUS_FAILED."""
    import faulthandler
    import traceback
    err = open(os.path.join(logdir, f"rank{rank}.stderr"), "w", buffering=1)
    os.dup2(err.fileno(), 2)
    sys.stderr = err
    faulthandler.enable(file=err, all_threads=True)
    try:
        dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world,
                                timeout=__import__("datetime").timedelta(seconds=120))
    except Exception:  # noqa: BLE001
        traceback.print_exc(file=err)
        os._exit(RENDEZVOUS_FAILED)
    try:
        worker(rank, world, q)
    except BaseException:  # noqa: BLE001 - reported through the file and the exit code
        traceback.print_exc(file=err)
        os._exit(1)


def _run_world_once(worker, world):
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        store = os.path.join(tmp, "rendezvous")
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_rank_main, args=(worker, r, world, store, q, tmp)) for r in range(world)]
        for p in procs:
            p.start()
        err = None
        try:
            out = q.get(timeout=300)
        except Exception as e:  # noqa: BLE001 - queue.Empty: a rank died or hung before rank 0 produced its result
            out, err = None, e
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
                p.join()
        logs = []
        for r in range(world):
            try:
                with open(os.path.join(tmp, f"rank{r}.stderr")) as f:
                    logs.append(f.read()[-4000:])
            except OSError:
                logs.append("<no stderr file>")
        return out, [p.exitcode for p in procs], err, logs


def _run_world(worker, world=2):
    """Spawn `world` ranks of `worker(rank, world, queue)` (already joined to a gloo group through a file store in a fresh
    temporary directory: no TCP port to collide on); returns what rank 0 put on the queue.  A rank that dies -- by a
    signal or by an exception -- fails the test on the spot with every rank's exit code and captured stderr.  Only a
    failed rendezvous (exit code RENDEZVOUS_FAILED, nothing of the code under test has run yet) is started once more."""
    out, codes, err, logs = _run_world_once(worker, world)

    def report():
        return (f"rank exit codes {codes}: {err!r}\n" +
                "\n".join(f"--- rank {r} stderr ---\n{t}" for r, t in enumerate(logs)))

    if any(c is not None and c < 0 for c in codes):
        pytest.fail("a rank was killed by a signal: " + report())
    if any(c == RENDEZVOUS_FAILED for c in codes) and not any(c not in (0, RENDEZVOUS_FAILED) for c in codes):
        print("[test_dist_gloo] rendezvous failed, starting the world once more: " + report())
        out, codes, err, logs = _run_world_once(worker, world)
    assert out is not None and all(c == 0 for c in codes), report()
    return out


def _until_end(row):
    row = row.tolist()
    return row[: row.index(2) + 1] if 2 in row else row


def test_sharded_decode_matches_single_process():
    toks, full = _run_world(_worker)
    assert toks.shape[0] == full.shape[0] == 5
    # shard step counts differ (per-batch early exit), tokens up to each row's [s] are shard-invariant
    for a, b in zip(toks, full):
        assert _until_end(a) == _until_end(b)


def _grad_worker(rank, world, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import restatement as R
    torch.set_num_threads(2)
    with open(os.path.join(GOLD, "manifests.json")) as f:
        man = json.load(f)
    cfg, sd = oracle_state_dict("T2", man["T2"], 24)
    img = synth.synth_images(4, 48, 64, seed=1040)
    text = synth.synth_labels(4, max_len=24, seed=1040)
    lo, hi = ddist.shard_bounds(4, rank, world)
    _, _, local, _ = R.train_step_grads(cfg, sd, img[lo:hi], text[lo:hi])  # per-rank loss.mean(), per-rank BN statistics
    names = [k for k in sd if k in local]  # state_dict (forward) order, like named_parameters()
    params = [sd[k] for k in names]
    sync = ddist.GradSync(bucket_bytes=8 << 20)
    plan = sync.plan(params[::-1])
    got = sync.collect(lambda n, dst: dst.copy_(local[n].reshape(-1)), names, params)
    # expectation: plain mean over ranks, computed with an independent collective per tensor
    worst = 0.0
    for n, g in zip(names, got):
        ref = local[n].clone()
        dist.all_reduce(ref)
        ref /= world
        assert g.shape == sd[n].shape
        worst = max(worst, float((g - ref).abs().max()))
    if rank == 0:
        q.put((worst, len(plan), len(names)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_is_bucketed_mean_over_ranks():
    """Data-parallel training exchange (config C3): GradSync returns the rank-mean of every gradient, in several
    buckets, with views shaped like the parameters."""
    worst, nbuckets, nparams = _run_world(_grad_worker)
    assert nparams == 164 and nbuckets >= 5  # ~56 M fp32 gradients in 8 MB buckets (some tensors exceed a bucket)
    assert worst <= 1e-7
