"""Experiment (development tool): per-step time of small block sets of the C4 pair matrix on one GPU -- two one-tile
diagonal blocks vs one 64-antenna diagonal block vs one cross tile -- the numbers behind the remark in DESIGN.md section 6
that a launch costs mostly the generation of its antenna rows (two diagonal tiles 34.6 ms, the 64-antenna diagonal block
WITH its cross tile 34.3 ms, one cross tile 26.9 ms at 8 times x 256 channels).  python tools/exp_tile_blocks.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bayeslim_amd import dist as rdist
dev = torch.device('cuda', 0)
inp = bench.build_inputs('c4', 8)
bls = bench.all_baselines(inp)
idx = {a: i for i, a in enumerate(inp['ants'])}
def timed(my_bls, group):
    rime, params, attach, _ = bench.build_model(inp, dev, my_bls)
    rime.mfma_group, rime.mfma_mode = group, True
    def step():
        for p in params: p.grad = None
        attach(); v = rime().data; (v.real ** 2 + v.imag ** 2).sum().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 4 * 1e3
    blocks = next(iter(rime._geom_cache.values()))['geom'].ant['blocks']
    return ms, [(b['cross'], b['nrows']) for b in blocks]
tile = lambda a: idx[a] // 32
sets = {
 'D0+D1 (g=32: two diagonal tiles)': ([b for b in bls if tile(b[0]) == tile(b[1]) and tile(b[0]) in (0, 1)], 32),
 'D0+D1 (g=64: one 64-antenna diagonal block, cross tile empty)': ([b for b in bls if tile(b[0]) == tile(b[1]) and tile(b[0]) in (0, 1)], 64),
 'O01 (g=32: one cross tile)': ([b for b in bls if (tile(b[0]), tile(b[1])) == (0, 1)], 32),
 'D0+D1+O01 (g=64 full diagonal block)': ([b for b in bls if tile(b[0]) in (0, 1) and tile(b[1]) in (0, 1)], 64),
}
for k, (bl, g) in sets.items():
    ms, blocks = timed(bl, g)
    print('%-62s %5d bl  %.2f ms/step  blocks %s' % (k, len(bl), ms, blocks), flush=True)
