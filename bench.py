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
across baselines and across channels.  Two partitions, BOTH measured (the faster one is `value`, the
other is reported under `alt`), each by a FRESH worker process per rank: the rank process the launcher
started only supervises (supervise_modes) and never touches the GPU, so a crash, hang or failed
self-check of one partition cannot lose the other's finished result:
  --shard bl   the north-star partition.  Arrays served by the antenna-factored matrix-core kernels
               are cut by whole 32 x 32 antenna-pair tile blocks (dist.plan_tile_shards: the MFMA work
               is sharded, every rank regenerates the E operands of the antennas its tiles touch);
               smaller arrays by contiguous baseline blocks.
  --shard freq contiguous channel blocks: shards the MFMA work, the operand generation and the
               per-channel sky / beam preparation.
Visibilities are all-gathered (RCCL) per time chunk, asynchronously, overlapping the next chunk's
kernels; gradients of replicated parameters are summed in place (shared) or all-gathered by block
(per-channel) from autograd hooks inside the last chunk's backward.  Total work is fixed, so scaling
is "strong".  After the timed region every N > 1 worker runs a VALUE SELF-CHECK (`dist.selfcheck`):
one more step of the sharded model with the loss restricted to 64 sampled baselines x the first time
of every chunk, whose gathered visibilities and exchanged gradients rank 0 compares with the
unsharded float64 model (1e-5 / 1e-4; a worker that fails it exits non-zero).
BENCH_FORCE_DIST=1 runs the same code path on ONE rank (RCCL initialised, every collective executed
with world size 1, supervisor + workers + self-check) to rehearse it on a one-GPU box.

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


def build_model(inp, dev, bls, seed=0, fblock=None, nchunks=1, redundant=False, dtype=torch.float32, pblock=None):
    """
    The drop-in modules on the GPU for this rank's shard: the baseline list `bls` and, when
    `fblock = (f0, f1)` is given, the channel block [f0, f1).  Parameters are created FULL-SIZE
    and identical on every rank (same seed); a channel-sharded rank feeds its modules views of
    them, re-attached before every forward by the returned `attach()` (the reference's own
    parameter protocol: params may be non-leaf graph tensors that are re-set each forward).
    `pblock = (r, w)`: the PIXEL partition -- this rank contracts the sky pixels r, r + w, r + 2 w, ... of the
    diffuse component (its a_lm transform: those columns of Ylm) and the point sources r, r + w, ...; a strided deal, so that
    every rank sees the same mix of declinations (contiguous HEALPix blocks are declination bands, some never above the
    horizon); the visibilities of the ranks are partial sums.
    Returns (rime, leaf parameters, attach, per-channel parameter descriptors).
    dtype: float32 (the benchmark) or float64 (parity runs of the SAME model: the random parameters are drawn in
    float32 and widened, so both precisions see identical values; set the default dtype to match).
    """
    from bayeslim_amd import utils, telescope_model, beam_model, sky_model, rime_model
    cfg = inp['cfg']
    f32 = dtype
    draw = torch.float32
    f0, f1 = (0, cfg['Nf']) if fblock is None else fblock
    freqs_full = torch.as_tensor(inp['freqs'], dtype=f32, device=dev)
    freqs = freqs_full[f0:f1]
    arr = telescope_model.ArrayModel(utils.AntposDict(inp['ants'], torch.as_tensor(inp['antvecs'])),
                                     freqs=freqs, device=dev, skip_reds=not redundant)
    tel = telescope_model.TelescopeModel((LON, LAT))
    gen = torch.Generator(device='cpu').manual_seed(seed)
    psel = slice(None) if pblock is None else slice(int(pblock[0]), None, int(pblock[1]))
    ra, dec, zenaz = inp['ra'][psel], inp['dec'][psel], inp['zenaz'][..., psel]
    Npix = len(ra)
    angs = torch.as_tensor(np.stack([ra, dec]), device=dev)
    if cfg.get('lmax'):
        # a_lm sky: params (1, 1, Nf, Ncoeff) complex -> map through AlmModel (HIP alm2pix kernels)
        from bayeslim_amd import sph_harm
        l, m = sph_harm.gen_lm(cfg['lmax'], real_field=True)
        A = sph_harm.AlmModel(l, m, real_output=True)
        A.device = dev
        A.setup_Ylm(90.0 - dec, ra, generate=True)
        amp = (1.0 / (1.0 + torch.as_tensor(np.asarray(l), dtype=draw)))[None, None, None, :]
        re = (torch.randn(1, 1, cfg['Nf'], len(l), generator=gen, dtype=draw) * amp).to(f32)
        im = (torch.randn(1, 1, cfg['Nf'], len(l), generator=gen, dtype=draw) * amp).to(f32)
        skyp = torch.nn.Parameter(torch.complex(re, im).to(dev))
        Rsky = sky_model.PixelSkyResponse(freqs, spatial_mode='alm', spat_LM=A, comp_params=False, device=dev)
    else:
        skyp = torch.nn.Parameter(torch.randn(1, 1, cfg['Nf'], len(inp['ra']), generator=gen, dtype=draw).to(dev, f32))
        Rsky = sky_model.PixelSkyResponse(freqs, device=dev)
    pixel_sky = not cfg.get('lmax')
    sky_view = (lambda: skyp[:, :, f0:f1][..., psel]) if pixel_sky else (lambda: skyp[:, :, f0:f1])
    diffuse = sky_model.PixelSky(sky_view().detach(), angs, inp['px_area'], R=Rsky,
                                 parameter=False, name='diffuse')
    leaves = [skyp]
    per_channel = [(skyp, 2)]                      # (parameter, channel axis)
    models = {'diffuse': diffuse}
    for t, za in zip(inp['times'], zenaz):
        tel.conv_cache[('diffuse', Npix, float(t))] = torch.as_tensor(np.ascontiguousarray(za))
    ptp = None
    if cfg['Npt'] > 0:
        pp = torch.ones(1, 1, 2, cfg['Npt'], dtype=f32)
        pp[..., 1, :] = -2.2
        R = sky_model.PointSkyResponse(freqs, freq_mode='powerlaw', f0=freqs_full[0], device=dev)
        if pblock is None:
            pts = sky_model.PointSky(pp.to(dev), torch.as_tensor(np.stack([inp['pt_ra'], inp['pt_dec']]), device=dev),
                                     R=R, parameter=True, name='points')
            leaves.append(pts.params)              # power-law params are shared by all channels
        else:
            # pixel partition: a full-size leaf, this rank's sources as a view of it (re-attached before every forward)
            ptp = torch.nn.Parameter(pp.to(dev))
            pts = sky_model.PointSky(ptp.detach()[..., psel], torch.as_tensor(np.stack([inp['pt_ra'][psel], inp['pt_dec'][psel]]),
                                                                          device=dev), R=R, parameter=False, name='points')
            leaves.append(ptp)
        models['points'] = pts
        npt = len(inp['pt_ra'][psel])
        for t, za in zip(inp['times'], inp['pt_zenaz'][..., psel]):
            tel.conv_cache[('points', npt, float(t))] = torch.as_tensor(np.ascontiguousarray(za))
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
                                freqs_full.double(), square=True).to(draw).to(f32)
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
    times = inp['times'] if nchunks == 1 else [np.asarray(t) for t in np.array_split(inp['times'], nchunks)]
    data_bls = None
    if redundant:
        # --redundant: simulate ONE baseline per redundant group and inflate to all requested baselines
        # (RIME's data_bls, rime_model.py:148-226, 436-437): the same visibilities, listed by redundant group
        want = set(bls)
        data_bls = [b for b in arr.get_bls(uniq_bls=False, keep_autos=False) if b in want]
        bls = [b for b in arr.get_bls(uniq_bls=True, keep_autos=False) if arr.bl2red[b] in {arr.bl2red[d] for d in data_bls}]
    rime = rime_model.RIME(sky, tel, beam, arr, bls, times,
                           torch.as_tensor(inp['freqs'][f0:f1], dtype=torch.float64, device=dev), data_bls=data_bls)

    def attach():
        """(re-)attach this rank's views of the replicated leaf parameters (new graph each step)"""
        diffuse.params = sky_view()
        beam.params = beamp[..., f0:f1, :]
        if ptp is not None:
            models['points'].params = ptp[..., psel]

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



def vendor_gemm_tflops(dev, n=8192, reps=20):
    """dense f16 rate of torch.matmul (hipBLASLt) on `dev`: the practical ceiling of the f16 MFMA pipe on this box"""
    a = torch.randn(n, n, device=dev, dtype=torch.float16)
    b = torch.randn(n, n, device=dev, dtype=torch.float16)
    for _ in range(3):
        a @ b
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n ** 3 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


def cpu_baseline(inp, nbl_sample=48):
    """
    forward + backward of the oracle (op for op the reference's per-time loop: FoV cut ->
    interpolated beam -> beam x sky -> (Nbl,Nf,P) fringe -> product -> pixel sum) in float32 on
    the usable host cores, for the first `nbl_sample` baselines x 1 time step of the SAME workload
    (diffuse component), minibatched over baselines as the reference must be to fit host RAM (every
    minibatch repeats the beam interpolation and the beam x sky product, as the reference's baseline
    groups do).  Timed at two minibatch sizes; `value` is the faster one, `value_prep_amortised` the
    rate with the per-minibatch preparation removed (linear fit t = a * minibatches + b * baselines):
    an upper bound on what larger host memory could buy the CPU path.
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

    def run(bl_batch):
        for s in range(0, nbl_sample, bl_batch):
            vis = orc.rime_forward(sky * inp['px_area'], zenaz.to(f32), beam_fn, blvecs[s:s + bl_batch],
                                   [(0, 0)] * len(blvecs[s:s + bl_batch]), freqs)
            (vis.real ** 2 + vis.imag ** 2).sum().backward()

    run(8)                                      # warm-up (allocator, caches)
    dts = {}
    for bl_batch in (8, 16):
        sky.grad = None
        bmap.grad = None
        t0 = time.perf_counter()
        run(bl_batch)
        dts[bl_batch] = time.perf_counter() - t0
    nvis = nbl_sample * 1 * len(freqs)
    best = min(dts, key=dts.get)
    # t = a * (nbl / batch) + b * nbl  ->  b from the two batch sizes
    n8, n16 = nbl_sample / 8.0, nbl_sample / 16.0
    a = (dts[8] - dts[16]) / (n8 - n16)
    b = (dts[8] - a * n8) / nbl_sample
    amort = (len(freqs) / b) if b > 0 and a > 0 else None
    return dict(value=nvis / dts[best], unit='vis/s', cores=ncores, kind='port',
                value_prep_amortised=amort,
                sample='%d baselines x 1 time x %d freqs x %d visible pixels (diffuse sky) of the same workload, '
                       'fwd+bwd, float32, baseline minibatches of 8 (%.1f s) and 16 (%.1f s); value = minibatch %d'
                       % (nbl_sample, len(freqs), len(cut), dts[8], dts[16], best),
                seconds=dts[8] + dts[16])



# ---------------------------------------------------------------------------------------------
# `python bench.py --gpus N` invoked plainly: start N fresh rank processes (one per GPU)
# ---------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(nproc, script_argv, env=None, timeout=None, out=None):
    """
    Run `script_argv` (a script path + its arguments) as `nproc` rank processes through
    `python -m torch.distributed.run` (one node, rendezvous on 127.0.0.1, a free port) as CHILD
    processes of this one, which never touches the GPU itself.  Returns (exit code, the ONE JSON
    result line rank 0 printed, or None).  Non-zero exit code when the launcher or any rank failed,
    when the ranks did not finish within `timeout` seconds (the whole process group is killed), or
    when no result line / a result for a different world size came back.  Everything else the
    children write to stdout is passed on to stderr.  The replaced pattern is the reference's
    single-process device loop (optim.py:1539-1566).
    """
    import signal
    import subprocess
    env = dict(os.environ if env is None else env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK'):
        env.pop(k, None)
    env.setdefault('OMP_NUM_THREADS', str(max(1, usable_cpus() // int(nproc))))
    port = _free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(nproc)),
           '--master-addr', '127.0.0.1', '--master-port', str(port)] + list(script_argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=timeout)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        # the ranks are this launcher's own process group (start_new_session): end exactly that group
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            try:
                proc.wait(timeout=15)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        stdout, _ = proc.communicate()
        sys.stderr.write('bench.py launcher: %d ranks did not finish within %s s; killed\n' % (nproc, timeout))
        rc = 124
    line = None
    for ln in stdout.decode(errors='replace').splitlines():
        try:
            obj = json.loads(ln)
        except ValueError:
            obj = None
        if isinstance(obj, dict) and 'metric' in obj and 'value' in obj:
            line = ln
            if rc == 0 and obj.get('n_gpus') != int(nproc):
                sys.stderr.write('bench.py launcher: result reports n_gpus=%r, expected %d\n' % (obj.get('n_gpus'), nproc))
                rc = 3
        elif ln.strip():
            sys.stderr.write(ln + '\n')
    if rc == 0 and line is None:
        sys.stderr.write('bench.py launcher: the ranks exited 0 but printed no result line\n')
        rc = 4
    if rc != 0:
        line = None if rc in (3, 4, 124) else line
    if out is not None and line is not None and rc == 0:
        out.write(line + '\n')
        out.flush()
    return rc, line


# ---------------------------------------------------------------------------------------------
# N > 1: one fresh WORKER process per shard mode under every rank (the rank process itself only supervises)
# ---------------------------------------------------------------------------------------------
SHARD_MODES = ('freq', 'bl')


def merge_mode_results(results):
    """
    results: [(shard, exit code, parsed JSON line or None)] in the order the modes ran.  The fastest mode whose worker
    exited 0 is the line; every other mode is listed under `alt` -- with its numbers when it finished, with
    `failed: <exit code>` when it did not (a mode that failed after printing a line, e.g. on its self-check, keeps the
    line's `selfcheck` beside the code).  Returns (line dict or None, exit code): 0 as soon as ONE mode is good.
    """
    good = [(m, o) for m, rc, o in results if rc == 0 and o is not None]
    if not good:
        # nothing usable: hand back the first line that exists (its numbers are NOT to be trusted) and fail
        first = next((o for _, _, o in results if o is not None), None)
        rc = next((rc for _, rc, _ in results if rc != 0), 4)
        if first is not None:
            first = dict(first, failed=[dict(shard=m, failed=rc) for m, rc, _ in results])
        return first, (rc or 4)
    best_mode, best = min(good, key=lambda mo: mo[1]['ms_per_step'])
    out = dict(best)
    alt = []
    for m, rc, o in results:
        if m == best_mode:
            continue
        if rc == 0 and o is not None:
            alt.append(dict(shard=m, parallelism=o.get('config', {}).get('parallelism'), ms_per_step=o['ms_per_step'],
                            value=o['value'], tile_plan_load=o.get('dist', {}).get('tile_plan_load'),
                            selfcheck=o.get('dist', {}).get('selfcheck')))
        else:
            e = dict(shard=m, failed=rc)
            if o is not None and o.get('dist', {}).get('selfcheck') is not None:
                e['selfcheck'] = o['dist']['selfcheck']
            alt.append(e)
    if alt:
        out['alt'] = alt
    return out, 0


def supervise_modes(modes, worker_argv, rank, world, timeout, out=None, env=None):
    """
    Run by EVERY rank process of an N > 1 job before anything touches the GPU: for each shard mode in turn start a
    fresh worker process (`worker_argv + ['--shard', mode, '--worker']`) that joins the other ranks' workers of the same
    mode in a process group of its own (BENCH_PG_TAG: a PrefixStore on the launcher's store, or a port of its own), wait
    for it with `timeout` seconds (the worker's process group is killed at the limit) and go on to the next mode
    WHATEVER happened to this one: a crash or hang of the second mode cannot lose the first mode's finished result.
    Rank 0 collects its workers' JSON lines, merges them (merge_mode_results) and writes ONE line to `out`.
    Returns the exit code: 0 when at least one mode finished on this rank.
    """
    import signal
    import subprocess
    import tempfile
    base_env = dict(os.environ if env is None else env)
    prev_limit = 0.0
    agent_store = base_env.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'
    base_port = int(base_env.get('MASTER_PORT', '29533'))
    if not agent_store and world == 1:
        base_port = _free_port()
    current = {}

    def forward(signum, frame):
        # the launcher ends its ranks with SIGTERM: take this rank's worker group along
        p = current.get('p')
        if p is not None and p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except ProcessLookupError:
                pass
        raise SystemExit(128 + signum)

    old = {}
    try:
        for sg in (signal.SIGTERM, signal.SIGINT):
            old[sg] = signal.signal(sg, forward)
    except ValueError:                      # not the main thread (tests): no forwarding
        old = {}
    results, took = [], []
    try:
        for k, mode in enumerate(modes):
            e = dict(base_env, BENCH_PG_TAG='%s%d' % (mode, k), RANK=str(rank), WORLD_SIZE=str(world))
            e.setdefault('LOCAL_RANK', str(rank))
            e.setdefault('MASTER_ADDR', '127.0.0.1')
            # without a launcher store the worker of rank 0 hosts the store itself: one port per mode
            e['MASTER_PORT'] = str(base_port if agent_store else (base_port + k if world > 1 else (base_port if k == 0 else _free_port())))
            cmd = [sys.executable] + list(worker_argv) + ['--shard', mode, '--worker']
            # a mode that follows a finished one gets 4 x that one's time (at least 3 min, at most `timeout`): a hang outside any
            # collective (those abort by their own watchdog) must not hold the finished result back for long
            limit = timeout
            done = [t for (_, rc_, _), t in zip(results, took) if rc_ == 0]
            if done and timeout is not None:
                limit = min(timeout, max(180.0, 4.0 * max(done)))
            # The limit runs from the moment the worker reports that its group has FORMED (it touches BENCH_READY_FILE after the
            # first barrier), not from its start: a peer that hung in the previous mode is held there until ITS limit, while
            # this rank's worker of that mode aborted on the collective watchdog long before -- the wait for the late peer at
            # the next rendezvous must not eat the next mode's time.  Until the marker shows up the worker may
            # wait for as long as a peer can still be held by the previous mode (+ the kill grace), then the limit applies anyway.
            lag = 0.0 if k == 0 or timeout is None else max(0.0, prev_limit - took[-1]) + 30.0
            ready = os.path.join(tempfile.gettempdir(), 'bench_ready_%d_%d_%s' % (os.getpid(), k, mode))
            if os.path.exists(ready):
                os.unlink(ready)
            e['BENCH_READY_FILE'] = ready
            prev_limit = limit if limit is not None else 0.0
            t0 = time.perf_counter()
            p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if rank == 0 else sys.stderr, start_new_session=True)
            current['p'] = p
            t_ready, stdout, rc = None, b'', None
            while True:
                try:
                    stdout, _ = p.communicate(timeout=0.5)
                    rc = p.returncode
                    break
                except subprocess.TimeoutExpired:
                    now = time.perf_counter()
                    if t_ready is None and os.path.exists(ready):
                        t_ready = now
                    if limit is None:
                        continue
                    if now > (t_ready if t_ready is not None else t0 + lag) + limit:
                        break
            if rc is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                    try:
                        p.wait(timeout=15)
                    except subprocess.TimeoutExpired:
                        os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                stdout, _ = p.communicate()
                rc = 124
                sys.stderr.write('bench.py rank %d: shard mode %s did not finish within %s s of %s; killed\n'
                                 % (rank, mode, limit, 'its rendezvous' if t_ready is not None else 'its start (+ %.0f s for late peers)' % lag))
            if os.path.exists(ready):
                os.unlink(ready)
            current['p'] = None
            obj = None
            for ln in (stdout or b'').decode(errors='replace').splitlines():
                try:
                    o = json.loads(ln)
                except ValueError:
                    o = None
                if isinstance(o, dict) and 'metric' in o and 'value' in o:
                    obj = o
                elif ln.strip():
                    sys.stderr.write(ln + '\n')
            if rc == 0 and rank == 0 and obj is None:
                rc = 4
            sys.stderr.write('bench.py rank %d: shard mode %s -> exit code %d after %.1f s\n' % (rank, mode, rc, time.perf_counter() - t0))
            sys.stderr.flush()
            results.append((mode, rc, obj))
            took.append(time.perf_counter() - t0)
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    if rank != 0:
        return 0 if any(rc == 0 for _, rc, _ in results) else next(rc for _, rc, _ in results)
    line, rc = merge_mode_results(results)
    if line is not None and out is not None:
        out.write(json.dumps(line) + '\n')
        out.flush()
    return rc


def _init_process_group(backend, rank, world, dev, timeout):
    """
    The workers of one shard mode form their own process group.  Under the launcher (torch.distributed.run keeps a
    TCPStore at MASTER_ADDR:MASTER_PORT and tells its children so) every mode's group lives under its own prefix of
    that store (BENCH_PG_TAG) -- two groups in a row with the default keys would read each other's RCCL unique ids;
    without a launcher store rank 0 hosts one at MASTER_PORT (a port per mode, chosen by the supervisor).
    """
    import torch.distributed as dist
    kw = dict(device_id=dev) if backend == 'nccl' else {}            # bind the communicator to this rank's GPU up front
    tag = os.environ.get('BENCH_PG_TAG')
    if tag is None:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout, **kw)
        return
    agent_store = os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'
    store = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), world,
                          is_master=(rank == 0 and not agent_store), timeout=timeout, wait_for_workers=False)
    dist.init_process_group(backend, store=dist.PrefixStore('bench/' + tag, store), rank=rank, world_size=world,
                            timeout=timeout, **kw)


def _set_collective_timeout(seconds):
    """lower the collective watchdog of the default group (a private torch API: guarded); returns the timeout in effect"""
    import datetime
    try:
        from torch.distributed import distributed_c10d as _c10d
        _c10d._set_pg_timeout(datetime.timedelta(seconds=float(seconds)))
        return float(seconds)
    except Exception as err:                 # API moved or refused: the rendezvous timeout stays
        sys.stderr.write('bench.py: collective timeout left at the default (%s: %s)\n' % (type(err).__name__, err))
        return None


SELFCHECK_TOL = dict(vis=1e-5, grad=1e-4)


def selfcheck_reference(inp, dev, bls, sample, first_times):
    """
    float64 reference of the self-check: the UNSHARDED model (same seed -> same parameters, widened) on the sampled
    baselines x the first time step of every time chunk x all channels, on the float64 vector-ALU kernels, with the
    self-check's loss (sum |V|^2 over exactly those entries).  Returns (vis, [gradients]).  Outside the timed region.
    """
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        sub = dict(inp, times=inp['times'][first_times], zenaz=inp['zenaz'][first_times])
        if 'pt_zenaz' in inp:
            sub['pt_zenaz'] = inp['pt_zenaz'][first_times]
        rime, params, attach, _ = build_model(sub, dev, [bls[i] for i in sample], dtype=torch.float64)
        attach()
        v = rime().data
        (v.real ** 2 + v.imag ** 2).sum().backward()
        torch.cuda.synchronize()
        return v.detach(), [p.grad.detach() for p in params]
    finally:
        torch.set_default_dtype(old)


OTHER_WORKLOADS = (('c3', [], None), ('c2', ['--steps', '20', '--warmup', '5'], None),
                   ('c5', ['--nf', '64', '--steps', '3', '--warmup', '2'], None),
                   # the headline array is a 127-antenna hexagon + outrigger: 63 mirror pairs + the outrigger + the hub, served by the
                   # conjugate-pair kernels (64 image rows).  The same workload with that form switched off (the mirror-pair
                   # kernels: conjugate rows copied, all 128 rows contracted) and with the symmetry search switched off = what an
                   # array WITHOUT point symmetry of this size costs (every phasor evaluated, every row contracted)
                   ('c4', ['--steps', '5', '--warmup', '3'], {'RIME_MIRROR': '0'}),
                   ('c4', ['--steps', '5', '--warmup', '3'], {'RIME_PAIR': '0'}),
                   # one rank's share of the headline workload under the channel partition at N = 8 (32 of the 256 channels), as a
                   # plain one-GPU run: what the scaling at 8 GPUs starts from before any collective
                   ('c4', ['--nf', '32', '--steps', '20', '--warmup', '5'], None))


def other_workloads(budget_s, timeout_each=150.0, script=None):
    """
    VERDICT r04 item 2: the driver's one command times BASELINE configs[3] (C4).  After its timed region and the CPU baseline,
    OUTSIDE both, the N = 1 run also measures configs[2] (C3), configs[1] (C2) and ONE RANK'S SHARE of configs[4] (C5: 64 of
    the 512 channels, the channel partition at N = 8) -- each in a fresh child process of this same script (the C4 model is
    freed first; a crash or hang there cannot touch the line already measured), a few steps each, and attaches
    {workload, ms_per_step, value, steps, kernels: {fwd, bwd: {frac, useful_frac_of_pipe_peak, ms_per_step}}} to the ONE JSON
    line.  Stops starting new ones when `budget_s` is used up, so that the whole command stays within minutes.
    """
    import subprocess
    out, t0 = [], time.perf_counter()
    for wl, extra, env in OTHER_WORKLOADS:
        left = budget_s - (time.perf_counter() - t0)
        tag = wl if env is None else wl + ' [' + ' '.join('%s=%s' % kv for kv in sorted(env.items())) + ']'
        if extra and '--nf' in extra:
            tag += ' [--nf %s]' % extra[extra.index('--nf') + 1]
        if left < 20.0:
            out.append(dict(workload=tag, skipped='time budget of the default run used up'))
            continue
        cmd = [sys.executable, script or os.path.abspath(__file__), '--workload', wl, '--no-cpu-baseline', '--no-other-workloads'] + \
              (extra if extra else ['--steps', '5', '--warmup', '3'])
        t1 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=min(timeout_each, left + 10.0),
                               env=None if env is None else dict(os.environ, **env))
            rc, txt = r.returncode, r.stdout.decode(errors='replace')
        except subprocess.TimeoutExpired:
            rc, txt = 124, ''
        line = None
        for ln in txt.splitlines():
            try:
                o = json.loads(ln)
            except ValueError:
                continue
            if isinstance(o, dict) and 'ms_per_step' in o:
                line = o
        if rc != 0 or line is None:
            out.append(dict(workload=tag, failed=rc))
            continue
        ks = (line.get('roofline') or {}).get('kernels') or {}
        steps = line['steps']

        def kern(prefix):
            hit = [(k, v) for k, v in ks.items() if k.startswith(prefix)]
            if not hit:
                return None
            k, v = max(hit, key=lambda kv: kv[1]['total_ms'])
            return dict(kernel=k, frac=v.get('frac'), useful_frac_of_pipe_peak=v.get('useful_frac_of_pipe_peak'),
                        ms_per_step=round(v['total_ms'] / steps, 4))
        out.append(dict(workload=tag, desc=line['config']['workload'], mirror_groups=line['config'].get('antenna_mirror_groups'),
                        pair_blocks=line['config'].get('antenna_pair_blocks'),
                        ms_per_step=round(line['ms_per_step'], 4),
                        value=line['value'], unit=line['unit'], steps=steps, warmup=line['warmup'],
                        kernels=dict(fwd=kern('fringe_ant_fwd') or kern('fringe_fwd'), bwd=kern('fringe_ant_bwd') or kern('fringe_bwd')),
                        wall_s=round(time.perf_counter() - t1, 1)))
    return out


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
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='N = 1, default workload: skip the short runs of BASELINE configs 3, 2 and the config-5 rank share that '
                         'follow the timed region and the CPU baseline (`other_workloads` in the JSON line); also skipped '
                         'with --no-cpu-baseline (kernel measurements, profiler runs)')
    ap.add_argument('--redundant', action='store_true',
                    help='NOT the headline configuration: simulate one baseline per redundant group and inflate to all '
                         'baselines (the reference\'s data_bls mechanism); N = 1 only')
    ap.add_argument('--shard', default='auto', choices=['auto', 'freq', 'bl', 'pix'],
                    help='multi-GPU partition: channel blocks or baseline (tile) blocks; auto measures both, each in a '
                         'fresh worker process, and reports the faster one as `value`, the other under `alt`')
    ap.add_argument('--chunks', type=int, default=2,
                    help='N > 1: time chunks per step (the all-gather of one chunk overlaps the kernels of the next)')
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)     # set by supervise_modes
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher around it: become the launcher.  Decided BEFORE anything touches the
    # GPU (no torch.cuda call yet); the ranks are fresh child processes, this process only relays the result
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        rc, _ = launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:],
                             timeout=float(os.environ.get('BENCH_LAUNCH_TIMEOUT', '2400')), out=sys.stdout)
        raise SystemExit(rc)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    distributed = world > 1 or os.environ.get('BENCH_FORCE_DIST', '0') == '1'

    # N > 1 with both partitions to measure: this rank process SUPERVISES one fresh worker per partition and never
    # touches the GPU itself (a crash, hang or failed self-check of one partition cannot lose the other's result)
    if distributed and args.shard == 'auto' and not args.worker:
        nants = len(hera_array(WORKLOADS[args.workload]['array'])[0])
        mfma_min = int(os.environ.get('RIME_MFMA_MIN_ANTS', '16'))
        modes = list(SHARD_MODES) if nants >= mfma_min else list(SHARD_MODES[::-1])
        rc = supervise_modes(modes, [os.path.abspath(__file__)] + sys.argv[1:], rank, world,
                             timeout=float(os.environ.get('BENCH_MODE_TIMEOUT', '600')), out=sys.stdout)
        raise SystemExit(rc)

    # stdout carries exactly ONE JSON line: libraries that chat on fd 1 (RCCL's version banner, gloo's
    # connection notes) are sent to stderr for the whole run, the result goes to the saved descriptor
    t_start = time.perf_counter()
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    # BENCH_DEVICE / BENCH_BACKEND exist only to rehearse the N > 1 code path on a one-GPU box
    # (all ranks on device 0 over gloo); the driver's runs use one rank per GPU over RCCL
    devidx = int(os.environ.get('BENCH_DEVICE', local_rank))
    if devidx >= torch.cuda.device_count():              # counting devices does not initialise the GPU
        raise SystemExit('bench.py: rank %d needs GPU %d but %d device(s) are visible (--gpus %d = one rank per GPU)'
                         % (rank, devidx, torch.cuda.device_count(), args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: bayeslim_amd has no CPU path')
    torch.cuda.set_device(devidx)
    dev = torch.device('cuda', devidx)
    import torch.distributed as dist
    collective_timeout = None
    if distributed:
        backend = os.environ.get('BENCH_BACKEND', 'nccl')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        import datetime
        # 'nccl' == RCCL on ROCm; device chosen above.  The rendezvous and everything up to the end of the warm-up get
        # minutes (the first `import torch` on a fresh box pages the image in, ranks arrive staggered, first-use kernel
        # loading and uneven tile plans skew the ranks); after the warm-up the watchdog drops to
        # BENCH_COLLECTIVE_TIMEOUT seconds (default 60; a step of any workload here takes < 6 s), so a collective that some
        # rank never joins aborts the job within a minute instead of hanging it
        _init_process_group(backend, rank, world, dev, datetime.timedelta(minutes=10))
        if dist.get_world_size() != args.gpus:
            raise SystemExit('bench.py: %d ranks joined, --gpus %d asked for' % (dist.get_world_size(), args.gpus))
        dist.barrier()
        collective_timeout = 600.0
        if os.environ.get('BENCH_READY_FILE'):           # the supervisor's per-mode limit starts now (supervise_modes)
            open(os.environ['BENCH_READY_FILE'], 'w').close()

    from bayeslim_amd import ops, dist as rdist

    def dbg(msg):
        if os.environ.get('BENCH_DEBUG'):
            sys.stderr.write('[bench rank %d %.1fs] %s\n' % (rank, time.perf_counter() - t_start, msg))
            sys.stderr.flush()

    cfg = WORKLOADS[args.workload]
    if args.nf:
        cfg = dict(cfg, Nf=args.nf, desc=cfg['desc'] + ' [%d channels]' % args.nf)
        WORKLOADS[args.workload] = cfg
    nt = args.nt or cfg['nt']
    inp = build_inputs(args.workload, nt)
    bls = all_baselines(inp)
    mfma_array = len(inp['ants']) >= ops.MFMA_MIN_ANTS          # antenna-factored matrix-core kernels apply
    if args.shard == 'auto':
        shard = 'freq' if mfma_array else 'bl'                  # N = 1: the partition is the whole problem either way
    else:
        shard = args.shard
    nchunks = 1 if not distributed else max(1, min(args.chunks, nt))
    do_check = os.environ.get('BENCH_SELFCHECK', '1' if distributed else '0') == '1'

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    def run_mode(shard):
        """build this rank's shard, warm up, time `steps` steps; returns the measurements"""
        nonlocal collective_timeout
        plan, pblock = None, None
        if shard == 'freq':
            bounds = rdist.shard_bounds(cfg['Nf'], world)
            my_bls, fblock, gdim, counts, inverse = bls, bounds[rank], 4, [e - s for s, e in bounds], None
            label = 'channel-sharded x%d' % world
        elif shard == 'pix':
            # SURVEY 8(e)'s third axis (round 5; NOT in the default `auto` pair): every rank contracts every w-th sky pixel /
            # point source for ALL baselines, times and channels; forward = all-reduce (sum) of the partial visibilities,
            # backward = all-reduce of the gradients (a rank's sky gradient is non-zero on its own pixels only)
            bounds, fblock, gdim, counts, inverse, my_bls = None, None, None, None, None, bls
            pblock = (rank, world) if distributed else None
            label = 'pixel-sharded x%d (every %d-th sky pixel / point source per rank)' % (world, world)
        else:
            bounds, fblock, gdim = None, None, 2
            if mfma_array and distributed:
                idx = {a: i for i, a in enumerate(inp['ants'])}
                plan = rdist.plan_tile_shards([(idx[a], idx[b]) for a, b in bls], len(inp['ants']), world)
            if plan is not None:
                my_bls = [bls[i] for i in plan['rank_bls'][rank]]
                counts = [len(b) for b in plan['rank_bls']]
                inverse = torch.as_tensor(plan['inverse'], device=dev)
                if os.environ.get('BENCH_BREAK_INVERSE') == '1':
                    # FAULT INJECTION for tests/test_bench_gpu.py only: leave the gathered baselines in rank order, so
                    # that the self-check has something to catch
                    inverse = None
                label = 'baseline-tile-sharded x%d (groups of %d antennas, %s blocks per rank)' % (
                    world, plan['group'], '/'.join(str(n) for n in plan['nblocks']))
            else:
                bb = rdist.shard_bounds(len(bls), world)
                my_bls, counts, inverse = bls[bb[rank][0]:bb[rank][1]], [e - s for s, e in bb], None
                label = 'baseline-sharded x%d' % world
        rime, params, attach, per_channel = build_model(inp, dev, my_bls, fblock=fblock, nchunks=nchunks,
                                                        redundant=args.redundant and not distributed, pblock=pblock)
        if args.redundant and not distributed:
            label += '; %d of %d baselines simulated (one per redundant group), inflated through data_bls' % (
                rime.Nsim_bls, len(my_bls))
        if plan is not None:
            rime.mfma_group, rime.mfma_mode = plan['group'], True
        gsync = None
        if distributed:
            if shard == 'freq':
                # per-channel parameters: every block is produced by exactly one rank -> all-gather of the
                # blocks; parameters shared by all channels -> all-reduce
                pc = {id(p) for p, _ in per_channel}
                # registration order = collective order; the hooks fire point sources, beam, sky (dist.grad_hook_order)
                gsync = rdist.GradSync(shared=[p for p in params if id(p) not in pc], blocks=per_channel[::-1], bounds=bounds)
            else:
                # registration order = collective order: the sky gradient (the large one) is final last (dist.grad_hook_order)
                gsync = rdist.GradSync(shared=params[1:] + params[:1])
        prof = []
        ops.PROFILE = prof

        def chisq_loss(vis, k):
            # sum |V|^2 through the fused chi-square epilogue (one pass forward, one pass backward);
            # the torch composition (vis.real ** 2 + vis.imag ** 2).sum() costs ten small launches and 0.3 ms per C4 step
            return ops.chisq(vis)

        def step(loss_fn=chisq_loss):
            for p in params:
                p.grad = None
            if not distributed:
                attach()
                vis = rime().data
                loss = loss_fn(vis, 0)
                loss.backward()
                return loss

            def forward_chunk(k):
                attach()
                rime.batch_idx = k
                return rime().data

            def gather_start(v):
                if shard == 'pix':
                    return rdist.all_reduce_vis_start(v)                                     # sum of the ranks' partial sums
                return rdist.all_gather_vis_start(v, counts, dim=gdim, inverse=inverse)     # RCCL, async, differentiable

            trace = None
            if os.environ.get('BENCH_DEBUG') == '2':
                def trace(msg):
                    torch.cuda.synchronize()
                    dbg(msg)
            return rdist.pipelined_step(forward_chunk, nchunks, loss_fn, gather_start, gsync, trace=trace)

        dbg('mode %s: model built (%d baselines, %s), %d chunk(s)' % (shard, len(my_bls), label, nchunks))
        for k in range(args.warmup):
            step()
            dbg('mode %s: warm-up step %d enqueued' % (shard, k))
        sync()
        dbg('mode %s: warm-up done' % shard)
        if distributed:
            # every rank has loaded its kernels and run its plan: from here on a collective takes milliseconds
            eff = _set_collective_timeout(os.environ.get('BENCH_COLLECTIVE_TIMEOUT', '60'))
            collective_timeout = eff if eff is not None else collective_timeout
        prof.clear()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        ops.PROFILE = None
        dbg('mode %s: timed steps done, %.1f ms/step' % (shard, dt / args.steps * 1e3))

        # ---- value self-check, OUTSIDE the timed region: one more step of the SAME sharded model with the loss
        # restricted to sampled baselines x the first time step of every chunk x all channels; rank 0 compares the
        # gathered visibilities and the exchanged gradients with the unsharded float64 model below
        check = None
        if do_check:
            sample = sorted(set(np.linspace(0, len(bls) - 1, min(64, len(bls))).round().astype(int).tolist()))
            s_idx = torch.as_tensor(sample, device=dev)
            first = [int(c[0]) for c in np.array_split(np.arange(nt), nchunks)]
            got = {}

            def check_loss(vis, k):
                s = vis.index_select(2, s_idx)[:, :, :, :1]
                got[k] = s.detach().clone()
                return (s.real ** 2 + s.imag ** 2).sum()

            step(check_loss)
            sync()
            check = dict(sample=sample, first=first, vis=torch.cat([got[k] for k in range(nchunks)], dim=3),
                         grads=[p.grad.detach().clone() for p in params])
            dbg('mode %s: self-check step done' % shard)
        if gsync is not None:
            gsync.remove()
        grad_bytes = sum(p.numel() * p.element_size() for p in params)
        vis_bytes = len(bls) * nt * cfg['Nf'] * 8
        mg = sorted({tuple(g) for bg in rime._geom_cache.values() if bg['geom'].ant is not None
                     for g in bg['geom'].ant.get('mirror_groups', [])})
        pb = sorted({tuple(g) for bg in rime._geom_cache.values() if bg['geom'].ant is not None
                     for g in bg['geom'].ant.get('pair_blocks', [])})
        res = dict(shard=shard, label=label, dt=dt, prof=list(prof), vis_bytes=vis_bytes, grad_bytes=grad_bytes, mirror=mg, pair=pb,
                   plan_load=None if plan is None else [round(x, 1) for x in plan['load']],
                   hook_order=None if gsync is None else list(gsync.fired),
                   order_adapted=None if gsync is None else gsync.adapted, check=check)
        del rime, params, attach
        torch.cuda.empty_cache()
        return res

    best = run_mode(shard)
    dt, prof = best['dt'], best['prof']

    selfcheck = None
    if distributed and best['check'] is not None:
        # rank 0 now computes the float64 reference while the others wait at the closing barrier: minutes, not seconds
        _set_collective_timeout(600)
    if best['check'] is not None and rank == 0:
        ck = best['check']
        v64, g64 = selfcheck_reference(inp, dev, bls, ck['sample'], ck['first'])
        vis_rel = float((ck['vis'].to(v64.dtype) - v64).abs().max() / v64.abs().max())
        grad_rel = max(float((a.to(b.dtype) - b).abs().max() / b.abs().max().clamp_min(1e-300)) for a, b in zip(ck['grads'], g64))
        ok = vis_rel < SELFCHECK_TOL['vis'] and grad_rel < SELFCHECK_TOL['grad']
        selfcheck = dict(vis_relmax=vis_rel, grad_relmax=grad_rel, tol=[SELFCHECK_TOL['vis'], SELFCHECK_TOL['grad']], ok=ok,
                         what='%d sampled baselines x first time of each of %d chunk(s) x %d channels: gathered visibilities and '
                              'exchanged gradients (loss = sum |V|^2 over those entries) of one extra step of the sharded model '
                              'after the timed region, against the unsharded float64 model on the vector-ALU kernels (rank 0); '
                              'max |a - b| / max |b|' % (len(ck['sample']), nchunks, cfg['Nf']))
        dbg('self-check: vis %.2e grad %.2e' % (vis_rel, grad_rel))
    best['check'] = None

    # kernel-level roofline from the HIP events recorded around each C-ABI launch
    kstat = {}
    for name, e0, e1, elems, mflops, abytes in prof:
        ms = e0.elapsed_time(e1)
        k = kstat.setdefault(name, [0, 0.0, 0, 0, 0, 0.0, []])
        k[0] += 1
        k[1] += ms
        k[2] += elems
        k[3] += mflops
        k[6].append((abytes, ms))
    for k in kstat.values():
        # the kernel's largest launches (C4: the diffuse component) and their mean duration
        k[4] = max(a for a, _ in k[6])
        big = [m for a, m in k[6] if a == k[4]]
        k[5] = sum(big) / len(big)
    roof = None
    if kstat:
        # dominant kernel = the one with the largest total time over the timed region, nothing else; the hot kernel with
        # the LOWEST fraction of its roof is named beside it (min_frac_kernel / min_frac)
        dom = max(kstat, key=lambda nm: kstat[nm][1])
        n, ms, elems, mflops, abytes_dom, ms_dom = kstat[dom][:6]
        flop_per_elem = 10.0                 # 6 (phase rotation) + 4 (real psky accumulate), SURVEY 8(d)
        algorithmic = elems * flop_per_elem / (ms * 1e-3) / 1e12
        # useful arithmetic of the contraction itself: one complex multiply-accumulate (8 flop) per
        # (antenna pair, pixel, channel, time) -- what an exact-f32 implementation would have to execute
        useful = elems * 8.0 / (ms * 1e-3) / 1e12
        if mflops > 0:
            # antenna-factored kernels: bounded by the f16 matrix cores; count the MFMA flops they
            # execute (3 hi/lo cross products on the upper-triangular antenna tiles, tile padding included;
            # the forward folds the symmetric products of the diagonal tiles: 7 instead of 12 MFMAs there)
            achieved, peak, pipe, bound = (mflops / (ms * 1e-3) / 1e12, F16_MFMA_PEAK_TFLOPS,
                                           'f16 MFMA, EXECUTED flops: 3 hi/lo split products + tile padding', 'mfma')
        else:
            achieved, peak, pipe, bound = algorithmic, FP32_PEAK_TFLOPS, 'fp32 vector ALU (== fp32 MFMA dense peak), algorithmic flops', 'valu'
        # HBM traffic of that kernel's DOMINANT launch (C4: the diffuse component): PMC counters cannot be read from inside
        # the process; a committed summary of separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes over this same
        # command is quoted when one exists for the workload (else null) and its path is given
        traffic, traffic_source, clock, clock_src = None, None, None, None
        for rel in ('profiles/r05/pmc_summary.json', 'profiles/r04/pmc_summary.json', 'profiles/r03/pmc_summary.json'):
            cpath = os.path.join(ROOT, rel)
            if traffic is None and args.workload == 'c4' and not distributed and not args.nf and os.path.exists(cpath):
                try:
                    pm = json.load(open(cpath))
                    # (the family label 'fringe_ant_*' covers the conjugate-pair kernels, profiled as fringe_pair_*)
                    names = (dom, dom.replace('fringe_ant_', 'fringe_pair_')) if best['pair'] else (dom,)
                    inst = [v for k, v in pm.items() if k.split('<')[0] in names and v.get('hbm_bytes_per_launch_larger_half')]
                    big = max(inst, key=lambda v: v['hbm_bytes_per_launch_larger_half'])
                    traffic = big['hbm_bytes_per_launch_larger_half']
                    traffic_source = rel + ' (separate rocprofv3 --pmc passes of this command, not this run; the kernel\'s largest launch)'
                    # the chip holds ~1.75 GHz of its 2.4 GHz under this load (GRBM_GUI_ACTIVE / kernel time)
                    clock, clock_src = big.get('clock_GHz'), rel
                except Exception:
                    traffic = None
        # every kernel's fraction of ITS roof (f16 MFMA peak on executed flops / fp32 peak on algorithmic flops)
        per_kernel = {k: dict(launches=v[0], total_ms=round(v[1], 3),
                              **({'executed_tflops': round(v[3] / (v[1] * 1e-3) / 1e12, 2),
                                  'frac': round(v[3] / (v[1] * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, 4),
                                  'useful_tflops': round(v[2] * 8.0 / (v[1] * 1e-3) / 1e12, 2),
                                  'useful_frac_of_pipe_peak': round(v[2] * 8.0 / (v[1] * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, 4)}
                                 if v[3] > 0 else
                                 {'algorithmic_tflops': round(v[2] * flop_per_elem / (v[1] * 1e-3) / 1e12, 2),
                                  'frac': round(v[2] * flop_per_elem / (v[1] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}))
                      for k, v in kstat.items()}
        # reference point measured in THIS run, after the timed region: the dense f16 rate the vendor GEMM (hipBLASLt through
        # torch.matmul, 8192^3) sustains on this box -- the 2.5 PFLOP/s peak assumes 2.4 GHz, under matrix load the chip
        # holds 1.7-1.8 GHz.  Information only: `frac` stays achieved / peak.
        vendor = None
        if mflops > 0 and world == 1:
            try:
                vendor = vendor_gemm_tflops(dev)
            except Exception:
                vendor = None
        hot = [k for k in kstat if kstat[k][1] >= 0.10 * kstat[dom][1]]      # kernels that matter for the step time
        worst = min(hot, key=lambda k: per_kernel[k]['frac'])
        # SURVEY 8(d)'s own figure for the WHOLE step: 2 E (6 + 4) flop per fringe element (forward + backward of the
        # baseline formulation) / wall time / the fp32 peak.  Above 1 on the matrix-core path: the antenna factorisation
        # moved the contraction to the f16 pipe and cut the exponentials Nant-fold -- not a precision-free comparison
        step_elems = sum(v[2] for v in kstat.values())
        survey_frac = step_elems * flop_per_elem / dt / 1e12 / FP32_PEAK_TFLOPS
        roof = dict(bound=bound, kernel=dom, achieved=round(achieved, 2), peak=peak, unit='TFLOP/s',
                    frac=round(achieved / peak, 4),
                    frac_note='EXECUTED flops of the dominant kernel / dense peak of its pipe' if mflops > 0 else
                              'algorithmic flops of the dominant kernel / fp32 peak',
                    pair_form_note=None if not best['pair'] else
                              'conjugate-pair kernels (point-symmetric array): all pairs from the images of one antenna of every mirror '
                              'pair -- 26 (forward) / 30 (backward) MFMAs per 16-pixel K step instead of 100 / 108, so `frac` (EXECUTED '
                              'flops / f16 peak) is lower than on the generic kernels while the useful rate (useful_frac_of_pipe_peak, '
                              '8 flop per pair x pixel x channel x time) is more than twice theirs; the kernels are now bound by operand '
                              'generation on the vector ALU (phases in f64, sin / cos), see profiles/r05/pair_form.txt',
                    useful_frac_of_pipe_peak=round(useful / peak, 4),
                    useful_tflops=round(useful, 2),
                    useful_note='8 flop (one complex MAC) per antenna pair x pixel x channel x time -- what the kernel is FOR; '
                                'the rest of the executed flops is the price of the 3-product f16 split and of tile padding.  '
                                'The fp32 vector / matrix peak that bounds an exact-f32 contraction is %.1f TFLOP/s' % FP32_PEAK_TFLOPS,
                    frac_vs_fp32_roof_survey8d=round(survey_frac, 4),
                    survey8d_note='whole step: 20 flop per fringe element (fwd + bwd, SURVEY 8d) / wall time / %.1f TFLOP/s'
                                  % FP32_PEAK_TFLOPS,
                    min_frac_kernel=worst, min_frac=per_kernel[worst]['frac'],
                    min_frac_note='lowest fraction of its roof among the kernels with >= 10 % of the dominant kernel\'s time',
                    traffic=traffic, traffic_source=traffic_source,
                    algorithmic_bytes=abytes_dom,
                    traffic_ratio=None if traffic is None or not abytes_dom else round(traffic / abytes_dom, 3),
                    bytes_note='per launch of the kernel\'s largest component (C4: diffuse sky): algorithmic = psky + visibilities '
                               '+ pointing vectors once; traffic = (2 FETCH_SIZE + WRITE_SIZE) KB counters, gfx950 correction',
                    dominant_launch_ms=round(ms_dom, 4),
                    hbm_frac_of_dominant_launch=round(abytes_dom / (ms_dom * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    pipe=pipe,
                    sustained_clock_GHz=None if clock is None else round(clock, 3),
                    frac_of_peak_at_sustained_clock=None if clock is None else round(achieved / (peak * clock / 2.4), 4),
                    sustained_clock_source=None if clock is None else clock_src + ' (separate PMC pass, not this run)',
                    vendor_gemm_f16_tflops=None if vendor is None else round(vendor, 1),
                    frac_of_vendor_gemm=None if vendor is None else round(achieved / vendor, 4),
                    vendor_gemm_note='torch.matmul f16 8192^3 (hipBLASLt) timed on this GPU right after the timed region',
                    algorithmic_tflops=round(algorithmic, 2),
                    launches=n, avg_launch_ms=round(ms / n, 4), elements_per_launch=elems // n,
                    flop_per_element=flop_per_elem,
                    hbm_equiv_frac=round(elems * 16.0 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                    note='algorithmic = 10 flop per fringe element (SURVEY 8d) of the baseline formulation; '
                         'hbm_equiv_frac = 16 B per fringe element of the unfused formulation / 8 TB/s',
                    kernels=per_kernel)


    exit_code = 0
    if rank == 0:
        nvis = len(bls) * nt * cfg['Nf']
        mfma_run = bool(kstat) and max(kstat, key=lambda n: kstat[n][1]).startswith('fringe_ant')
        out = dict(metric='visibilities/sec (Nbl x Ntime x Nfreq) fwd+bwd', value=nvis * args.steps / dt,
                   unit='vis/s', n_gpus=(dist.get_world_size() if distributed else 1), steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='strong',
                   vs_baseline=None,
                   dtype=('f32 (fringe sum: f16x3 split operands on the f16 MFMA -- 22-bit operands, f32 accumulate; '
                          'phases f64)' if mfma_run else 'f32 (phases f64)'),
                   tolerance='visibilities 1e-5, gradients 1e-4 against the fp64 reference, max-norm scaled '
                             '(max |a - b| / max |b|); measured on the MFMA path: <= 3e-6 / 2e-6',
                   data='synthetic',
                   config=dict(workload=cfg['desc'], Nbl=len(bls), Ntimes_per_step=nt, Nfreqs=cfg['Nf'],
                               Npix_sky=int(len(inp['ra'])), Npix_visible=int((inp['zenaz'][0, 0] < 90).sum()),
                               Npoint=cfg['Npt'], beam='Airy D=14m on 1deg rect grid, linear PixelBeam interp',
                               loss='sum |V|^2 (fused chi-square epilogue, rime_chisq_fwd / _bwd)',
                               # (mirror groups, 16-row groups) of the antenna blocks: groups whose second octet of rows holds the
                               # MIRROR antennas of the first (r' - c = -(r - c)): conjugate phasors, not evaluated again
                               antenna_mirror_groups=[list(g) for g in best['mirror']] or None,
                               antenna_pair_blocks=[list(g) for g in best['pair']] or None,
                               parallelism=best['label']),
                   roofline=roof)
        if selfcheck is not None and not distributed:
            out['selfcheck'] = selfcheck
        if distributed:
            out['dist'] = dict(world_size=dist.get_world_size(), backend=dist.get_backend(),
                               nccl_version=('.'.join(str(v) for v in torch.cuda.nccl.version())
                                             if dist.get_backend() == 'nccl' else None),
                               shard=shard, time_chunks=nchunks,
                               all_gather_vis_bytes_per_step=best['vis_bytes'],
                               gradient_bytes_per_step=best['grad_bytes'],
                               tile_plan_load=best['plan_load'],
                               grad_hook_order=best['hook_order'],     # collective-order indices in firing order (rank 0)
                               grad_order_adapted=best['order_adapted'],   # the ranks agreed on the firing order and use it
                               collective_timeout_s=collective_timeout,
                               loss_note='every rank evaluates the loss on the gathered visibilities (%.0f MB per step)'
                                         % (best['vis_bytes'] / 1e6),
                               selfcheck=selfcheck,
                               overlap='vis all-gather of chunk k runs under the kernels of chunk k+1; gradient '
                                       'collectives start from autograd hooks inside the last backward')
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(inp)
            out['cpu_baseline'] = cb
            out['speedup_vs_cpu_baseline'] = out['value'] / cb['value']
            if cb.get('value_prep_amortised'):
                out['speedup_vs_cpu_prep_amortised'] = out['value'] / cb['value_prep_amortised']
            if (args.workload == 'c4' and not args.nf and not args.nt and not args.redundant and not distributed
                    and not args.no_other_workloads):
                # outside the timed region and the CPU baseline; the C4 model is gone (run_mode freed it)
                torch.cuda.empty_cache()
                out['other_workloads'] = other_workloads(float(os.environ.get('BENCH_OTHER_BUDGET', '100')))
                out['other_workloads_note'] = ('BASELINE configs 3, 2, one rank\'s share of config 5 (64 of 512 channels) and the headline '
                                               'workload with the mirror-pair search off (an array of this size without point symmetry), '
                                               'each a short run of this script in a child process after the timed region and the CPU '
                                               'baseline; frac = executed f16 MFMA flops / 2.5 PFLOP/s of the fringe kernel named')
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + '\n').encode())
        if selfcheck is not None and not selfcheck['ok']:
            sys.stderr.write('bench.py: SELF-CHECK FAILED: vis %.3e (tol %.0e), grad %.3e (tol %.0e)\n'
                             % (selfcheck['vis_relmax'], SELFCHECK_TOL['vis'], selfcheck['grad_relmax'], SELFCHECK_TOL['grad']))
            exit_code = 5
    if distributed:
        try:
            dist.barrier()                       # nobody tears its communicator down while rank 0 still checks / reports
        finally:
            dist.destroy_process_group()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == '__main__':
    main()
