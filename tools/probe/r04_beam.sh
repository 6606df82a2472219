#!/bin/bash
# Round 4: device-side beam bookkeeping -- fixtures, rate (batched / per sample), end-bias scan for the bench's second beam leg.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_beam
mkdir -p $out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py tests/test_fp16x2_gpu.py -x -q -k "beam" > $out/tests.log 2>&1; echo "beam tests rc=$?"; tail -6 $out/tests.log
timeout -k 10 300 python3 tools/beam_bench.py 128 5 both > $out/beam_bench.log 2>&1; tail -3 $out/beam_bench.log
timeout -k 10 300 python3 - <<'PY' > $out/end_bias_scan.log 2>&1
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from doc2tex_amd import Model, synth
H, W = synth.crop_shape("C4")
img = synth.synth_images(128, H, W, seed=11).cuda()
for eb in (0.8, 1.0, 1.2, 1.4, 1.6):
    m = Model(synth.make_config("C4", device="cuda", beam_size=5))
    m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}, end_bias=eb), strict=False)
    m = m.cuda().eval()
    with torch.no_grad():
        m.beam_search_batch(img, 5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        seqs = m.beam_search_batch(img, 5)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lens = sorted(int(q.shape[1]) for q, _ in seqs)
    print(f"end_bias {eb}: {128 / dt:.1f} formulas/s, lengths min {lens[0]} q1 {lens[32]} median {lens[64]} q3 {lens[96]} max {lens[-1]}", flush=True)
    del m
PY
cat $out/end_bias_scan.log | grep end_bias
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --precision fp32 --no-secondary > $out/fp32.log 2>&1; grep '^{' $out/fp32.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("fp32", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])'
