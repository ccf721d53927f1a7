#!/bin/bash
# Lab: A/B of the software-pipelined one-wave-per-SIMD backward (round 4, profiles/r04/lab_bwd_pipe.txt) on one box: agreement with
# the shipped 8-wave kernel, then timing.  The kernel lives in the lab patch only: build the lab library first, here (CPU):
#     tools/build_variant.sh lab ""
# and run this through gpurun; RIME_BWD_PIPE=1 selects the kernel inside that library.
cd $GRAFT_REPO_ROOT
export RIME_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/librime_lab.so
python - <<'E'
import os, subprocess, sys, json
code = '''
import torch, numpy as np, sys
sys.path.insert(0, ".")
import bench
from bayeslim_amd import ops
inp = bench.build_inputs("c4", 1)
bench.WORKLOADS["c4"] = dict(bench.WORKLOADS["c4"], Nf=4)
rime, params, attach, _ = bench.build_model(bench.build_inputs("c4", 1), torch.device("cuda", 0), bench.all_baselines(inp))
attach(); v = rime().data; ops.chisq(v).backward(); torch.cuda.synchronize()
torch.save([p.grad.cpu() for p in params], sys.argv[1])
'''
for pz in ('0', '1'):
    subprocess.check_call([sys.executable, '-c', code, '/tmp/g%s.pt' % pz], env=dict(os.environ, RIME_BWD_PIPE=pz), stderr=subprocess.DEVNULL)
import torch
a, b = torch.load('/tmp/g0.pt'), torch.load('/tmp/g1.pt')
print('bitwise equal gradients:', [bool(torch.equal(x, y)) for x, y in zip(a, b)], 'max rel diff', max(float((x - y).abs().max() / x.abs().max()) for x, y in zip(a, b)))
E
for r in 1 2; do for pz in 0 1; do RIME_BWD_PIPE=$pz python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('c4 pipe=$pz', round(d['ms_per_step'],2), {n[11:14]:round(x['total_ms']/5,2) for n,x in k.items()})"; done; done
