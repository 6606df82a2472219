"""Two real ranks over RCCL (backend "nccl"), one process per GPU: the data-parallel paths of SURVEY 8e on hardware.
Skips on a one-GPU box (today's pool); runs the day a multi-GPU box is there.

  * inference (configs[1], [2], [4]): doc2tex_amd.dist.decode_sharded -- every rank decodes its contiguous shard with the
    HIP engine, the int64 token ids are all-gathered -- equals the single-rank decode of the whole batch;
  * training (configs[3]): GradSync -- each rank's backward over its own shard, bucketed all-reduce-mean overlapped with
    the backward -- equals the mean of the per-rank gradients computed one rank at a time.

The ranks are fresh child processes started BEFORE anything in them has touched the GPU (never re-exec or fork a process
that initialised HIP); each writes its stderr to a file that is attached to the failure message; a rank that exits
non-zero takes its sibling with it."""
import os
import subprocess
import sys
import time

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

RANK_PROGRAM = r"""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
from conftest import engine_model
from doc2tex_amd import dist as ddist, synth
mode = sys.argv[1]
if mode == "decode":
    cfg, m = engine_model("T2", 24, end_bias=1.81, device=str(dev))   # rows end at different steps
    img = synth.synth_images(5, 48, 64, seed=1001).to(dev)           # 5 rows -> shards of 3 and 2
    text1 = lambda n: torch.full((n, 1), 1, dtype=torch.long, device=dev)
    with torch.no_grad():
        decode = lambda x: m(x, text1(x.shape[0]), is_train=False, is_test=True)[0]
        toks = ddist.decode_sharded(decode, img)
        full = decode(img)
    def until_end(row):
        row = row.tolist()
        return row[: row.index(2) + 1] if 2 in row else row
    assert toks.shape[0] == 5
    for a, b in zip(toks, full):
        assert until_end(a) == until_end(b), (until_end(a), until_end(b))
    # shard logits are bit-identical to the whole-batch ones (kernels are dispatched by layer shape only)
    lo, hi = ddist.shard_bounds(5, rank, world)
    with torch.no_grad():
        l_sh = m(img[lo:hi], text1(hi - lo), is_train=False)[1]
        l_all = m(img, text1(5), is_train=False)[1]
    assert torch.equal(l_sh, l_all[lo:hi])
else:
    cfg, m = engine_model("T2", 24, device=str(dev))
    m.train()
    img = synth.synth_images(4, 48, 64, seed=1040).to(dev)
    text = synth.synth_labels(4, max_len=24, seed=1040).to(dev)
    from doc2tex_amd.loss import create_criterion
    crit = create_criterion("entropy", {{"ignore_index": 0, "reduction": "none"}})
    def grads(lo, hi, sync):
        m.grad_sync = sync
        m.zero_grad()
        _, preds, _ = m(img[lo:hi], text[lo:hi, :-1])
        crit(preds.view(-1, preds.shape[-1]), text[lo:hi, 1:].contiguous().view(-1)).mean().backward()
        torch.cuda.synchronize(dev)
        return {{n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}}
    lo, hi = ddist.shard_bounds(4, rank, world)
    got = grads(lo, hi, ddist.GradSync(bucket_bytes=8 << 20))
    # expectation: every rank computes EVERY shard's gradient locally (no collective) and averages them
    per = [grads(*ddist.shard_bounds(4, r, world), None) for r in range(world)]
    worst = 0.0
    for n, g in got.items():
        ref = sum(p[n] for p in per) / world
        scale = float(ref.abs().max()) + 1e-12
        worst = max(worst, float((g - ref).abs().max()) / scale)
    assert worst <= 1e-5, worst   # same kernels on the same shard: only the summation order of the all-reduce differs
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _run_ranks(mode, tmp_path, world=2, limit=600):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    prog = tmp_path / "rank_program.py"
    prog.write_text(RANK_PROGRAM.format(root=ROOT))
    procs, logs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        log = open(tmp_path / f"rank{r}.log", "w")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, str(prog), mode], env=env, stdout=log, stderr=subprocess.STDOUT))
    t0 = time.time()
    codes = [None] * world
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = any(c not in (None, 0) for c in codes)
        if failed or time.time() - t0 > limit:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
                    try:
                        codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.2)
    for log in logs:
        log.close()
    text = "\n".join(f"--- rank {r} (exit {codes[r]}) ---\n" + (tmp_path / f"rank{r}.log").read_text()[-4000:] for r in range(world))
    assert all(c == 0 for c in codes), text


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_decode_sharded_over_rccl_two_ranks(tmp_path):
    _run_ranks("decode", tmp_path)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_grad_sync_over_rccl_two_ranks(tmp_path):
    _run_ranks("train", tmp_path)


@pytest.mark.parametrize("mode", ["decode", "train"])
def test_rank_program_with_one_rank(tmp_path, mode):
    """The same rank program as a one-rank RCCL world: runs on today's one-GPU boxes, so the program the two-rank tests
    launch is known to work before a multi-GPU box ever sees it."""
    _run_ranks(mode, tmp_path, world=1)
