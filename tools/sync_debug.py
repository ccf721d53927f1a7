"""List the host<->device synchronisation points of one bench step (torch sync debug mode)."""
import os, sys, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda', 0)
inp = bench.build_inputs(wl, nt)
bls = bench.all_baselines(inp)
rime, params, attach, per_channel = bench.build_model(inp, dev, bls)


def step():
    for p in params:
        p.grad = None
    attach()
    vd = rime()
    v = vd.data
    loss = (v.real ** 2 + v.imag ** 2).sum()
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()


def showwarning(message, category, filename, lineno, file=None, line=None):
    print('SYNC:', message)
    for fs in traceback.extract_stack()[:-1]:
        if 'bayeslim_amd' in fs.filename or 'bench.py' in fs.filename or 'sync_debug' in fs.filename:
            print('    %s:%d %s | %s' % (os.path.basename(fs.filename), fs.lineno, fs.name, fs.line))


warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
step()
torch.cuda.set_sync_debug_mode('default')
