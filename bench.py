#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X-native RIME hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2] [--nt NT]

Metric (BASELINE.json): visibilities/s = Nbl x Ntimes x Nfreqs per wall second of one
forward + backward pass of RIME (loss = sum |V|^2, gradients w.r.t. sky and beam parameters),
inputs resident in HBM, synthetic data of the named configuration, random-init parameters.

Default workload "c4" = BASELINE.json configs[3], the configuration the north_star quotes its
target on and which fits one GPU: HERA-128 (hex-127 + 1 outrigger, 8128 baselines), nside-128
HEALPix diffuse sky + 1e4 point sources, 256 channels, interpolated Airy PixelBeam.
A "step" = one RIME forward + backward over a minibatch of NT time steps.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): visibilities are independent
across baselines and across channels; the work is sharded in contiguous CHANNEL blocks when the
antenna-factored matrix-core kernels apply (their cost does not depend on how many antenna pairs
are requested, and channel blocks also shard the per-channel sky/beam preparation), otherwise in
contiguous BASELINE blocks (--shard).  Visibilities are all-gathered (RCCL), gradients of
replicated parameters are all-reduced (shared) or all-gathered by block (per-channel); total work
is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields), including
  roofline     -- dominant kernel (fused fringe sum): algorithmic flops / measured kernel time
                  (HIP events on the launch stream) against the fp32 peak, and
  cpu_baseline -- the CPU oracle (op-for-op torch restatement of the reference path) timed on
                  the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3          # MI355X fp32 vector == fp32 MFMA dense peak (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense f16/bf16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
LAT, LON = -30.72148, 21.42827


# ---------------------------------------------------------------------------------------------
# synthetic workload
# ---------------------------------------------------------------------------------------------
def hera_array(kind):
    from bayeslim_amd import utils
    if kind == 'hera128':
        ants, vecs = utils._make_hex(7, D=14.6)            # 127 antennas
        ants = ants + [len(ants)]
        vecs = np.vstack([vecs, [[250.0, 0.0, 0.0]]])      # + 1 outrigger -> 128 ants, 8128 bls
    elif kind == 'hera19':
        ants, vecs = utils._make_hex(3, D=14.6)
    elif kind == 'hera37':
        ants, vecs = utils._make_hex(4, D=14.6)
    elif kind == 'ska512':
        # SKA-Low-like core: 512 stations, random within a 500 m radius disc (seeded), nearly coplanar
        rng = np.random.default_rng(512)
        r, ph = 500.0 * np.sqrt(rng.uniform(0, 1, 512)), rng.uniform(0, 2 * np.pi, 512)
        vecs = np.stack([r * np.cos(ph), r * np.sin(ph), rng.normal(0, 0.5, 512)], axis=1)
        ants = list(range(512))
    else:
        raise ValueError(kind)
    return ants, vecs


WORKLOADS = {
    # name: array, nside, Nfreqs, Npoint, default Ntimes per step
    'c4': dict(array='hera128', nside=128, Nf=256, Npt=10000, nt=8,
               desc='HERA-128 (8128 bl), nside=128 diffuse + 1e4 point sources, 256 freqs'),
    'c2': dict(array='hera19', nside=32, Nf=64, Npt=0, nt=30,
               desc='HERA-19 hex (171 bl), nside=32 diffuse sky, 64 freqs, 30 times'),
    'c5': dict(array='ska512', nside=256, Nf=512, Npt=0, nt=1, pol=4,
               desc='SKA-Low-like 512 stations (130816 bl), full-pol 2x2 Jones PixelBeam, nside=256 coherency sky, 512 freqs'),
    'c3': dict(array='hera37', nside=64, Nf=128, Npt=0, nt=60, lmax=128,
               desc='HERA-37 (666 bl), a_lm sky lmax=128 on nside=64 via sph_harm + PixelBeam interp, 128 freqs, 60 times'),
}


def build_inputs(wl, nt, seed=0):
    """seeded host-side description of the workload (numpy, float64); shared by GPU and CPU legs"""
    from bayeslim_amd import healpix, telescope_model
    cfg = WORKLOADS[wl]
    rng = np.random.default_rng(seed)
    ants, vecs = hera_array(cfg['array'])
    freqs = np.linspace(120e6, 180e6, cfg['Nf'])
    times = 2459861.0 + np.arange(nt) * 10.0 / 1440.0
    colat, lon = healpix.pix2ang(cfg['nside'])
    dec, ra = 90.0 - np.rad2deg(colat), np.rad2deg(lon)
    keep = dec < 59.27852                                    # as tests/test_sky.py:18 of the reference
    ra, dec = ra[keep], dec[keep]
    px_area = healpix.nside2pixarea(cfg['nside'])
    inp = dict(cfg=cfg, ants=ants, antvecs=vecs, freqs=freqs, times=times, ra=ra, dec=dec, px_area=px_area)
    inp['zenaz'] = np.stack([np.stack(telescope_model.eq2top((LON, LAT), t, ra, dec)) for t in times])
    if cfg['Npt'] > 0:
        lst = telescope_model.JD2LST(times[0], LON)
        # uniform on the cap within 80 deg of the first zenith
        cz = rng.uniform(np.cos(np.deg2rad(80.0)), 1.0, cfg['Npt'])
        psi = rng.uniform(0, 2 * np.pi, cfg['Npt'])
        zen0, az0 = np.arccos(cz), psi
        # zen/az -> ra/dec at time 0 (inverse of the LST rotation)
        x, y, z = np.sin(zen0) * np.sin(az0), np.sin(zen0) * np.cos(az0), np.cos(zen0)
        p = np.deg2rad(LAT)
        sd = y * np.cos(p) + z * np.sin(p)
        pdec = np.arcsin(np.clip(sd, -1, 1))
        H = np.arctan2(-x, z * np.cos(p) - y * np.sin(p))
        inp['pt_ra'] = np.mod(lst - np.rad2deg(H), 360.0)
        inp['pt_dec'] = np.rad2deg(pdec)
        inp['pt_zenaz'] = np.stack([np.stack(telescope_model.eq2top((LON, LAT), t, inp['pt_ra'], inp['pt_dec']))
                                    for t in times])
    # beam: Airy D = 14 m sampled on a 1-degree (zen, az) grid, linear interpolation
    inp['theta_grid'] = np.arange(0, 90.1, 1.0)
    inp['phi_grid'] = np.arange(0, 360, 1.0)
    return inp


def build_model(inp, dev, bls, seed=0, fblock=None):
    """
    The drop-in modules on the GPU for this rank's shard: the baseline list `bls` and, when
    `fblock = (f0, f1)` is given, the channel block [f0, f1).  Parameters are created FULL-SIZE
    and identical on every rank (same seed); a channel-sharded rank feeds its modules views of
    them, re-attached before every forward by the returned `attach()` (the reference's own
    parameter protocol: params may be non-leaf graph tensors that are re-set each forward).
    Returns (rime, leaf parameters, attach, per-channel parameter descriptors).
    """
    from bayeslim_amd import utils, telescope_model, beam_model, sky_model, rime_model
    cfg = inp['cfg']
    f32 = torch.float32
    f0, f1 = (0, cfg['Nf']) if fblock is None else fblock
    freqs_full = torch.as_tensor(inp['freqs'], dtype=f32, device=dev)
    freqs = freqs_full[f0:f1]
    arr = telescope_model.ArrayModel(utils.AntposDict(inp['ants'], torch.as_tensor(inp['antvecs'])),
                                     freqs=freqs, device=dev, skip_reds=True)
    tel = telescope_model.TelescopeModel((LON, LAT))
    gen = torch.Generator(device='cpu').manual_seed(seed)
    Npix = len(inp['ra'])
    angs = torch.as_tensor(np.stack([inp['ra'], inp['dec']]), device=dev)
    if cfg.get('lmax'):
        # a_lm sky: params (1, 1, Nf, Ncoeff) complex -> map through AlmModel (HIP alm2pix kernels)
        from bayeslim_amd import sph_harm
        l, m = sph_harm.gen_lm(cfg['lmax'], real_field=True)
        A = sph_harm.AlmModel(l, m, real_output=True)
        A.device = dev
        A.setup_Ylm(90.0 - inp['dec'], inp['ra'], generate=True)
        amp = (1.0 / (1.0 + torch.as_tensor(np.asarray(l), dtype=f32)))[None, None, None, :]
        re = torch.randn(1, 1, cfg['Nf'], len(l), generator=gen, dtype=f32) * amp
        im = torch.randn(1, 1, cfg['Nf'], len(l), generator=gen, dtype=f32) * amp
        skyp = torch.nn.Parameter(torch.complex(re, im).to(dev))
        Rsky = sky_model.PixelSkyResponse(freqs, spatial_mode='alm', spat_LM=A, comp_params=False, device=dev)
    else:
        skyp = torch.nn.Parameter(torch.randn(1, 1, cfg['Nf'], Npix, generator=gen, dtype=f32).to(dev))
        Rsky = sky_model.PixelSkyResponse(freqs, device=dev)
    diffuse = sky_model.PixelSky(skyp.detach()[:, :, f0:f1], angs, inp['px_area'], R=Rsky,
                                 parameter=False, name='diffuse')
    leaves = [skyp]
    per_channel = [(skyp, 2)]                      # (parameter, channel axis)
    models = {'diffuse': diffuse}
    for t, za in zip(inp['times'], inp['zenaz']):
        tel.conv_cache[('diffuse', Npix, float(t))] = torch.as_tensor(za)
    if cfg['Npt'] > 0:
        pp = torch.ones(1, 1, 2, cfg['Npt'], dtype=f32)
        pp[..., 1, :] = -2.2
        R = sky_model.PointSkyResponse(freqs, freq_mode='powerlaw', f0=freqs_full[0], device=dev)
        pts = sky_model.PointSky(pp.to(dev), torch.as_tensor(np.stack([inp['pt_ra'], inp['pt_dec']]), device=dev),
                                 R=R, parameter=True, name='points')
        models['points'] = pts
        leaves.append(pts.params)                  # power-law params are shared by all channels
        for t, za in zip(inp['times'], inp['pt_zenaz']):
            tel.conv_cache[('points', cfg['Npt'], float(t))] = torch.as_tensor(za)
    sky = sky_model.CompositeModel(models) if len(models) > 1 else diffuse
    if cfg.get('pol') == 4:
        # Stokes I map + fixed fractional Q, U, V -> (2, 2) coherency sky (complex): sky_model.Stokes2Coherency
        frac = torch.tensor([0.05, -0.03, 0.01], dtype=f32, device=dev).reshape(3, 1, 1, 1)
        s2c = sky_model.Stokes2Coherency(params=frac)

        class CohSky(utils.Module):
            def __init__(self):
                super().__init__(name='cohsky')
                self.sky, self.s2c, self.device = diffuse, s2c, diffuse.device

            def forward(self, prior_cache=None, **kw):
                return self.s2c(self.sky(prior_cache=prior_cache))

        sky = CohSky()
    tg = torch.as_tensor(inp['theta_grid'], device=dev)
    pg = torch.as_tensor(inp['phi_grid'], device=dev)
    b_phi, b_theta = torch.meshgrid(pg, tg, indexing='xy')
    airy = beam_model.airy_disk(b_theta.ravel() * utils.D2R, b_phi.ravel() * utils.D2R, 14.0,
                                freqs_full.double(), square=True).to(f32)
    if cfg.get('pol') == 4:
        # (2, 2) Jones voltage beam: sqrt(Airy) on the diagonal, 3 % leakage terms
        volt = airy.clamp_min(0).sqrt()
        jones = torch.stack([torch.stack([volt, 0.03 * volt]), torch.stack([-0.03 * volt, volt])])   # (2,2,Nf,Npb)
        beamp = torch.nn.Parameter(jones[:, :, None].contiguous())
        R = beam_model.PixelResponse(freqs, 'rect', interp_mode='linear', theta_grid=tg, phi_grid=pg,
                                     freq_mode='channel', powerbeam=False, realbeam=True, device=dev)
        beam = beam_model.PixelBeam(beamp.detach()[..., f0:f1, :], freqs, R=R, pol='e', powerbeam=False,
                                    fov=180, parameter=False)
    else:
        beamp = torch.nn.Parameter(airy[None, None, None].contiguous())
        R = beam_model.PixelResponse(freqs, 'rect', interp_mode='linear', theta_grid=tg, phi_grid=pg,
                                     freq_mode='channel', powerbeam=True, device=dev)
        beam = beam_model.PixelBeam(beamp.detach()[..., f0:f1, :], freqs, R=R, pol='e', powerbeam=True,
                                    fov=180, parameter=False)
    leaves.append(beamp)
    per_channel.append((beamp, 3))
    # geometry frequencies stay float64: an exactly uniform grid lets the fringe kernel use its
    # rotation recurrence (float32-rounded channel centres are not uniform to better than ~8 Hz)
    rime = rime_model.RIME(sky, tel, beam, arr, bls, inp['times'],
                           torch.as_tensor(inp['freqs'][f0:f1], dtype=torch.float64, device=dev))

    def attach():
        """(re-)attach this rank's views of the replicated leaf parameters (new graph each step)"""
        diffuse.params = skyp[:, :, f0:f1]
        beam.params = beamp[..., f0:f1, :]

    return rime, leaves, attach, per_channel


def all_baselines(inp):
    ants = inp['ants']
    return [(ants[i], ants[j]) for i in range(len(ants)) for j in range(i + 1, len(ants))]


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle restatement of the reference path on a bounded sample
# ---------------------------------------------------------------------------------------------
def usable_cpus():
    """host cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (a GPU box exposes 256 logical CPUs in the mask but grants a 16-CPU quota)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n



def cpu_baseline(inp, nbl_sample=48, bl_batch=8):
    """
    forward + backward of the oracle (op for op the reference's per-time loop: FoV cut ->
    interpolated beam -> beam x sky -> (Nbl,Nf,P) fringe -> product -> pixel sum) in float32 on
    the usable host cores, for the first `nbl_sample` baselines x 1 time step of the SAME workload
    (diffuse component), minibatched over baselines as the reference must be to fit host RAM.
    """
    from oracle import rime_oracle as orc
    ncores = usable_cpus()
    torch.set_num_threads(ncores)
    f32 = torch.float32
    freqs = torch.as_tensor(inp['freqs'], dtype=f32)
    ants = inp['ants']
    bls = all_baselines(inp)[:nbl_sample]
    av = torch.as_tensor(inp['antvecs'], dtype=f32)
    blvecs = torch.stack([av[ants.index(j)] - av[ants.index(i)] for i, j in bls])
    Npix = len(inp['ra'])
    sky = torch.randn(1, 1, len(freqs), Npix, dtype=f32, requires_grad=True)
    tg, pg = torch.as_tensor(inp['theta_grid']), torch.as_tensor(inp['phi_grid'])
    b_phi, b_theta = torch.meshgrid(pg, tg, indexing='xy')
    bmap = orc.airy_beam(b_theta.ravel(), b_phi.ravel(), 14.0, freqs.double()).to(f32)[None, None, None]
    bmap.requires_grad_(True)
    zenaz = torch.as_tensor(inp['zenaz'][:1])
    cut = orc.fov_cut(zenaz[0, 0], 180.0)
    inds, wgts = orc.rect_interp_weights(tg, pg, zenaz[0, 0][cut], zenaz[0, 1][cut], 'linear')
    wgts = wgts.to(f32)

    def beam_fn(z, a):
        return orc.interp(orc.pixel_response_forward(bmap), inds, wgts)

    def run():
        for s in range(0, nbl_sample, bl_batch):
            vis = orc.rime_forward(sky * inp['px_area'], zenaz.to(f32), beam_fn, blvecs[s:s + bl_batch],
                                   [(0, 0)] * len(blvecs[s:s + bl_batch]), freqs)
            (vis.real ** 2 + vis.imag ** 2).sum().backward()

    run()                                       # warm-up (allocator, caches)
    sky.grad = None
    bmap.grad = None
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    nvis = nbl_sample * 1 * len(freqs)
    return dict(value=nvis / dt, unit='vis/s', cores=ncores, kind='port',
                sample='%d baselines x 1 time x %d freqs x %d visible pixels (diffuse sky) of the same workload, '
                       'fwd+bwd, float32, baseline minibatch %d, %.1f s' % (nbl_sample, len(freqs), len(cut), bl_batch, dt),
                seconds=dt)


# ---------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c4', choices=sorted(WORKLOADS))
    ap.add_argument('--nt', type=int, default=None, help='time steps per step (minibatch)')
    ap.add_argument('--nf', type=int, default=None, help='override the number of channels (e.g. one rank\'s share of c5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shard', default='auto', choices=['auto', 'freq', 'bl'],
                    help='multi-GPU partition: channel blocks or baseline blocks (auto: channels when the '
                         'antenna-factored matrix-core kernels apply, i.e. workloads c3 / c4; else baselines)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: bayeslim_amd has no CPU path')
    # BENCH_DEVICE / BENCH_BACKEND exist only to rehearse the N > 1 code path on a one-GPU box
    # (all ranks on device 0 over gloo); the driver's runs use one rank per GPU over RCCL
    devidx = int(os.environ.get('BENCH_DEVICE', local_rank))
    torch.cuda.set_device(devidx)
    dev = torch.device('cuda', devidx)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get('BENCH_BACKEND', 'nccl')
        dist.init_process_group(backend)          # 'nccl' == RCCL on ROCm; device chosen by set_device above
    assert world == args.gpus or world == 1, 'launch N ranks with torch.distributed.run for --gpus N'

    from bayeslim_amd import ops, dist as rdist
    cfg = WORKLOADS[args.workload]
    if args.nf:
        cfg = dict(cfg, Nf=args.nf, desc=cfg['desc'] + ' [%d channels]' % args.nf)
        WORKLOADS[args.workload] = cfg
    nt = args.nt or cfg['nt']
    inp = build_inputs(args.workload, nt)
    bls = all_baselines(inp)
    shard = args.shard
    if shard == 'auto':
        shard = 'freq' if args.workload in ('c3', 'c4', 'c5') else 'bl'     # >= 33 antennas: antenna-factored kernels
    if shard == 'freq':
        bounds = rdist.shard_bounds(cfg['Nf'], world)
        my_bls, fblock, gdim = bls, bounds[rank], 4
    else:
        bounds = rdist.shard_bounds(len(bls), world)
        my_bls, fblock, gdim = bls[bounds[rank][0]:bounds[rank][1]], None, 2
    counts = [e - s for s, e in bounds]
    rime, params, attach, per_channel = build_model(inp, dev, my_bls, fblock=fblock)

    prof = []
    ops.PROFILE = prof

    def step():
        for p in params:
            p.grad = None
        attach()
        vd = rime()
        vis = vd.data
        if world > 1:
            vis = rdist.all_gather_vis(vis, counts, dim=gdim)       # RCCL all-gather (differentiable)
        loss = (vis.real ** 2 + vis.imag ** 2).sum()
        loss.backward()
        if world > 1:
            if shard == 'freq':
                # per-channel parameters: every block is produced by exactly one rank -> all-gather
                # of the blocks; parameters shared by all channels -> all-reduce
                pc = {id(p) for p, _ in per_channel}
                for p, ax in per_channel:
                    rdist.all_gather_block_grads(p, ax, bounds)
                rdist.all_reduce_grads([p for p in params if id(p) not in pc])
            else:
                rdist.all_reduce_grads(params)
        return loss

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    prof.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ops.PROFILE = None

    # kernel-level roofline from the HIP events recorded around each C-ABI launch
    kstat = {}
    for name, e0, e1, elems, mflops in prof:
        ms = e0.elapsed_time(e1)
        k = kstat.setdefault(name, [0, 0.0, 0, 0])
        k[0] += 1
        k[1] += ms
        k[2] += elems
        k[3] += mflops
    roof = None
    if kstat:
        dom = max(kstat, key=lambda n: kstat[n][1])
        n, ms, elems, mflops = kstat[dom]
        flop_per_elem = 10.0                 # 6 (phase rotation) + 4 (real psky accumulate), SURVEY 8(d)
        algorithmic = elems * flop_per_elem / (ms * 1e-3) / 1e12
        if mflops > 0:
            # antenna-factored kernels: bounded by the f16 matrix cores; count the MFMA flops they
            # execute (3 hi/lo cross products on the upper-triangular antenna tiles; the forward folds the
            # symmetric products of the diagonal tiles: 7 instead of 12 MFMAs there)
            achieved, peak, pipe = mflops / (ms * 1e-3) / 1e12, F16_MFMA_PEAK_TFLOPS, 'f16 MFMA (v_mfma_f32_32x32x16_f16), executed flops'
        else:
            achieved, peak, pipe = algorithmic, FP32_PEAK_TFLOPS, 'fp32 vector ALU (== fp32 MFMA dense peak), algorithmic flops'
        FP32 = FP32_PEAK_TFLOPS
        # HBM traffic of that kernel per launch: PMC counters cannot be read from inside the
        # process; the committed summary of the separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`
        # passes over this same command is used when it matches the workload (else null)
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'r01', 'traffic.json')
        if args.workload == 'c4' and world == 1 and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        roof = dict(bound='mfma', kernel=dom, achieved=round(achieved, 2), peak=peak, unit='TFLOP/s',
                    frac=round(achieved / peak, 4), traffic=traffic, pipe=pipe,
                    algorithmic_tflops=round(algorithmic, 2), algorithmic_frac_of_fp32_peak=round(algorithmic / FP32, 4),
                    launches=n, avg_launch_ms=round(ms / n, 4), elements_per_launch=elems // n,
                    flop_per_element=flop_per_elem,
                    hbm_equiv_frac=round(elems * 16.0 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                    note='algorithmic = 10 flop per fringe element (SURVEY 8d) of the baseline formulation; '
                         'hbm_equiv_frac = 16 B per fringe element of the unfused formulation / 8 TB/s',
                    kernels={k: dict(launches=v[0], total_ms=round(v[1], 3)) for k, v in kstat.items()})

    if rank == 0:
        nvis = len(bls) * nt * cfg['Nf']
        out = dict(metric='visibilities/sec (Nbl x Ntime x Nfreq) fwd+bwd', value=nvis * args.steps / dt,
                   unit='vis/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='strong',
                   vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload=cfg['desc'], Nbl=len(bls), Ntimes_per_step=nt, Nfreqs=cfg['Nf'],
                               Npix_sky=int(len(inp['ra'])), Npix_visible=int((inp['zenaz'][0, 0] < 90).sum()),
                               Npoint=cfg['Npt'], beam='Airy D=14m on 1deg rect grid, linear PixelBeam interp',
                               parallelism=('channel-sharded x%d' if shard == 'freq' else 'baseline-sharded x%d') % world),
                   roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(inp)
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
