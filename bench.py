#!/usr/bin/env python
"""Headline benchmark: env-steps/sec of the PPO2 update (GAE + noptepochs x nminibatches minibatch
steps: gather -> NatureCNN fwd -> loss -> bwd -> [RCCL all-reduce] -> clip -> Adam) on a pre-filled
device rollout, Atari-shaped (84x84x4 uint8, Discrete(6)), num_envs=4096 (whole job), nsteps=128,
reference Atari hyper-parameters (ppo2/defaults.py:15-22: noptepochs=4, nminibatches=4,
ent_coef=0.01, cliprange=0.1, lr=2.5e-4).  BASELINE.json metric; SURVEY.md 8(d) "update-only".

    python bench.py --gpus N --steps K --warmup W

N>1: one rank per GPU over RCCL.  Either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or directly -- without WORLD_SIZE the script re-launches itself through torch.distributed.run with N
ranks on this node.  num_envs is SHARDED over the ranks (strong scaling); the flat 6.75 MB gradient is all-reduced once
per minibatch step by libmrl.so itself (mrl_comm, issued from inside the backward pass on a communication stream).

A "step" = one update over num_envs*nsteps env-steps.  Data: synthetic (device-resident counter-hash env; rollout
filled by real Runner.run() calls).  Prints ONE JSON line on rank 0; with one GPU and default arguments the line also
carries short measurements of BASELINE.json's other single-GPU configurations under "other_configs".
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.6    # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0


def _relaunch(args):
    """`python bench.py --gpus N` with no launcher around it: become the launcher (one rank per GPU of this node)."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.exit(subprocess.call(cmd, env=env))


def host_cpu():
    model = None
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return os.cpu_count() or 1, model


def cpu_baseline(kind):
    """The oracle (CPU port of the reference path: NumPy GAE / shuffle / gather / adv-norm +
    torch-CPU restatement of the TF graph, all host cores like the reference's TF session,
    tf_util.py:58-66) timed on a BOUNDED sample of the same workload: one full update at
    num_envs=128 (Atari) / 256 (MuJoCo-shaped), capped at ~20 s, same nsteps / epochs / minibatches."""
    import numpy as np
    import torch
    from oracle import ppo2_numpy as O
    from oracle.ppo2_torch import OracleModel
    ncpu, cpu_model = host_cpu()
    if kind == 'atari':
        N, T, E, M = 128, 128, 4, 4
        net = dict(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6,
                   value_network=None, ent_coef=0.01)
        lr, clip = 2.5e-4, 0.1
    else:
        N, T, E, M = 256, 128, 10, 32
        net = dict(network='mlp', ob_shape=(376,), ob_dtype=np.float32, pd_kind='gaussian', nact=17,
                   value_network='copy', ent_coef=0.0)
        lr, clip = 3e-4, 0.2
    np.random.seed(0)
    om = OracleModel(vf_coef=0.5, max_grad_norm=0.5, **net)
    ro = O.synthetic_rollout(kind, T, N, 0)
    # thread count: the reference's TF session uses every core (tf_util.py:58-66); on many-core hosts
    # that is slower than a moderate count for these batch sizes, so time one minibatch step at a few
    # counts and keep the fastest (reported as `cores`)
    probe = O.sf01(ro['obs'])[:1024]
    pidx = np.arange(probe.shape[0])
    best = None
    for nt in sorted({ncpu, min(ncpu, 64), min(ncpu, 16)}):
        torch.set_num_threads(nt)
        tb = None
        for _rep in range(3):                      # first call warms the thread pool; keep the fastest
            ta = time.perf_counter()
            om.train(lr, clip, probe, O.sf01(ro['rewards'])[pidx], None, O.sf01(ro['actions'])[pidx],
                     O.sf01(ro['values'])[pidx], O.sf01(ro['neglogpacs'])[pidx])
            dtp = time.perf_counter() - ta
            tb = dtp if tb is None else min(tb, dtp)
        if best is None or tb < best[0] * 0.9:     # more threads must win by >10% to be chosen
            best = (tb, nt)
    cores = best[1]
    torch.set_num_threads(cores)
    np.random.seed(0)
    om = OracleModel(vf_coef=0.5, max_grad_norm=0.5, **net)
    budget_s = 20.0
    t0 = time.perf_counter()
    returns, _ = O.gae(ro['rewards'], ro['values'], ro['dones'], ro['last_values'], ro['last_dones'], 0.99, 0.95)
    f = {k: O.sf01(ro[k]) for k in ('obs', 'actions', 'values', 'neglogpacs')}
    fret = O.sf01(returns)
    done_steps = 0
    for idx in O.minibatch_indices(T * N, T * N // M, E):
        om.train(lr, clip, f['obs'][idx], fret[idx], None, f['actions'][idx], f['values'][idx], f['neglogpacs'][idx])
        done_steps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    # env-steps/s of a whole update, extrapolated from the minibatch steps that fit in the time budget
    frac = done_steps / float(E * M)
    return {'value': N * T * frac / dt, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'host_cpu_count': ncpu, 'host_cpu_model': cpu_model,
            'sample': 'PPO2 update at num_envs=%d nsteps=%d: GAE + sf01 + %d of its %dx%d minibatch steps (%.1f s of CPU '
                      'work, scaled to a whole update); oracle = reference NumPy path + torch-CPU fp32 restatement '
                      'of the TF graph, %d torch threads (fastest of the counts probed on this %d-thread host)'
                      % (N, T, done_steps, E, M, dt, cores, ncpu)}


def arithmetic_mode(_lib):
    """which arithmetic the GEMM sites of the NatureCNN update run in (engine options in effect, include/mrl.h)"""
    x3, x6, dg = _lib.get_option('u8_bf16x3'), _lib.get_option('f32_bf16x6'), _lib.get_option('dgrad_x6')
    x6 = 2 if x6 else 0
    wx = _lib.get_option('wgrad_x8') if x6 == 2 else 0
    nprod = _lib.get_option('f32_products') if x6 else 0        # 6 (product builds) | 8 (-DMRL_PRODUCTS8 builds)
    sites = {}
    for s in ('c1.fwd', 'c1.wgrad'):
        sites[s] = 3 if x3 else 0
    for s in ('c2.fwd', 'c3.fwd', 'fc1.fwd', 'fc1.dgrad'):
        sites[s] = nprod
    for s in ('c2.dgrad', 'c3.dgrad'):
        sites[s] = nprod if dg else 0
    wtr = _lib.get_option('wgrad_tr') if x6 == 2 else 0      # image-resident transpose-read kernel (wgradtr.hip.h)
    for s in ('c2.wgrad', 'c3.wgrad'):
        sites[s] = nprod if (wx >= 2 or wtr) else 0
    sites['fc1.wgrad'] = nprod if wx >= 1 else 0
    text = ('fp32 storage, fp32 accumulation, every product at least as accurate as an IEEE fp32 multiply. '
            'fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise fmaf chain): %s. '
            'bf16 MFMA on EXACT operand splits -- uint8 pixels x 3 exact bf16 planes of the other operand (3 products, '
            'exact): %s; fp32 x fp32 with both operands split into 3 exact bf16 planes (round-to-nearest at each level), '
            '%d of the 9 partial products kept (dropped part <= 2^-%d of a product, %s on average; one fp32 multiply rounds '
            'by up to 2^-24, 2.1e-8 on average): %s.'
            % (', '.join(k for k, v in sites.items() if v == 0) or '-',
               ', '.join(k for k, v in sites.items() if v == 3) or '-', nprod, 33 if nprod == 8 else 24,
               '1e-11' if nprod == 8 else '3.5e-9', ', '.join(k for k, v in sites.items() if v >= 6) or '-'))
    short = 'bf16x%d-rn-split' % nprod if x6 else 'f32-mfma'
    return sites, text, short


def source_sha16():
    """identity of the kernels that are running: sha256 over the HIP sources + the public header (the built .so is not
    tracked and .git does not travel to the GPU box).  scripts/pmc_to_json.py records the same value in the PMC file."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'baselines_amd', 'csrc')
    names = sorted(n for n in os.listdir(src) if n.endswith(('.hip', '.h')))
    for n in names:
        h.update(n.encode())
        h.update(open(os.path.join(src, n), 'rb').read())
    h.update(open(os.path.join(ROOT, 'include', 'mrl.h'), 'rb').read())
    return h.hexdigest()[:16]


def nature_cnn_alg_bytes(B):
    """ALGORITHMIC HBM bytes per launch of each GEMM site of the NatureCNN minibatch step at B samples: every operand
    the launch must read or write, once (fp32 activations / gradients, uint8 observations, 1-bit ReLU masks where the
    kernel consumes or produces them; weights are negligible).  DESIGN.md section 3 lists the same figures."""
    ob, h1, h2, h3, h4 = 84 * 84 * 4, 20 * 20 * 32 * 4, 9 * 9 * 64 * 4, 7 * 7 * 64 * 4, 512 * 4
    m1, m2, m3 = 20 * 20 * 32 // 8, 9 * 9 * 64 // 8, 7 * 7 * 64 // 8
    per = {'c1.fwd': ob + h1 + m1, 'c2.fwd': h1 + h2 + m2, 'c3.fwd': h2 + h3 + m3, 'fc1.fwd': h3 + h4,
           'fc1.dgrad': h4 + h3 + m3, 'fc1.wgrad': h3 + h4, 'c3.dgrad': h3 + h2 + m2, 'c3.wgrad': h2 + h3,
           'c2.dgrad': h2 + h1 + m1, 'c2.wgrad': h1 + h2, 'c1.wgrad': ob + h1}
    return {k: float(v) * B for k, v in per.items()}


def lstm_alg_bytes(B, nin=512, nh=128):
    """the recurrent row's GEMM sites outside the scans (a2c/utils.py:81-102: x @ wx for all steps at once, its data and weight
    gradients, and h @ wh's weight gradient), fp32 operands once each; they run on the generic fp32-MFMA engine"""
    g = 4 * nh
    return {'lstm.fwd': 4.0 * B * (nin + g), 'lstm.dgrad': 4.0 * B * (g + nin), 'lstm.wgrad': 4.0 * B * (nin + g),
            'lstm_h.wgrad': 4.0 * B * (nh + g)}


def mlp_step_alg_bytes(B, P, ob_dim=376, nact=17):
    """the fused MLP minibatch step (one launch = gather + both 2x64 networks forward + loss + backward): per sample the
    observation row, its index, the four per-sample scalars and the action (SURVEY.md 8(d) K4+K5, fused: no activations or
    pdparams reach HBM), plus the parameters read and the gradient written once"""
    return {'mlp_step': float(B) * (4 * ob_dim + 8 + 16 + 4 * nact) + 8.0 * P}


def kernel_rooflines(prof, sites, alg_bytes=None, pmc=None):
    """per launch site: the roofline that BINDS it.  A site with flops is priced on max(algorithmic bytes / HBM peak,
    algorithmic flops / peak of the matrix pipe it runs on); `bound` says which floor is the larger one, `achieved` /
    `peak` / `frac` are in that resource's unit (frac = floor time / measured time), and both fractions are listed.
    `traffic` = HBM-side bytes per launch from the PMC passes of the same sources (else null)."""
    out = {}
    alg_bytes = alg_bytes or {}
    pmc = pmc or {}
    for k, v in prof.items():
        if not v['count'] or v['ms'] <= 0:
            continue
        avg_s = v['ms'] / v['count'] * 1e-3
        nbytes = alg_bytes.get(k, v['bytes'] / v['count'] if v['bytes'] > 0 else 0.0)
        if v['flops'] > 0:
            nprod = sites.get(k, 0)
            peak = PEAK_BF16_MFMA_TFLOPS / nprod if nprod else PEAK_F32_MFMA_TFLOPS
            fl = v['flops'] / v['count']
            t_mfma, t_hbm = fl / (peak * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)
            pipe = ('bf16 MFMA / %d exact products per multiply' % nprod) if nprod else 'fp32 MFMA'
            r = {'pipe': pipe, 'avg_ms': avg_s * 1e3, 'launches': v['count'], 'mfma_tflops': fl / avg_s / 1e12,
                 'mfma_peak_tflops': peak, 'mfma_frac': t_mfma / avg_s,
                 # the contract figure of SURVEY.md 8(d): the same fp32-equivalent flops against the fp32 matrix pipe (157.3 TF)
                 'frac_vs_fp32_mfma': fl / avg_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 'alg_bytes': nbytes,
                 'hbm_gbs': nbytes / avg_s / 1e9, 'hbm_frac': t_hbm / avg_s}
            if t_hbm > t_mfma:
                r.update({'bound': 'hbm', 'achieved': nbytes / avg_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                          'frac': t_hbm / avg_s})
            else:
                r.update({'bound': 'mfma', 'achieved': fl / avg_s / 1e12, 'peak': peak, 'unit': 'TFLOP/s',
                          'frac': t_mfma / avg_s})
            out[k] = r
        elif nbytes > 0:
            ach = nbytes / avg_s / 1e9
            out[k] = {'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach / PEAK_HBM_GBS,
                      'avg_ms': avg_s * 1e3, 'launches': v['count'], 'alg_bytes': nbytes}
        if k in out:
            tr = pmc.get(k, {}).get('hbm_bytes')
            out[k]['traffic'] = tr
            if tr and nbytes > 0:
                out[k]['traffic_over_algorithmic'] = tr / nbytes
    return out


def profile_round_key(path):
    """profiles/rNN<letters>_*.json -> (NN, letters): r10a sorts after r09z and after r9z"""
    import re
    m = re.match(r'r(\d+)([a-z]*)_', os.path.basename(path))
    return (int(m.group(1)), m.group(2)) if m else (-1, os.path.basename(path))


def load_pmc(nbatch_train):
    """newest committed profiles/*_pmc_hbm.json -- used only if it was collected from the sources that are running"""
    import glob
    if nbatch_train != 131072:
        return {}, None
    cand = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_hbm.json')), key=profile_round_key)
    if not cand:
        return {}, 'no profiles/*_pmc_hbm.json'
    with open(cand[-1]) as fh:
        pm = json.load(fh)
    name = 'profiles/' + os.path.basename(cand[-1])
    have, want = pm.get('source_sha16'), source_sha16()
    if have != want:
        return {}, '%s was collected from other kernel sources (%s, running %s): traffic withheld' % (name, have, want)
    return pm.get('per_launch', {}), name


def dp_verify(model, ro, T, N, hp, world, rank, comm):
    """Self-validation of the data-parallel gradient BEFORE the timed region (the >1-rank RCCL path has
    never met a second GPU in the build environment, so the first multi-GPU run has to check itself).  One minibatch of the
    shape the ranks time (same env-major indices on every rank), its gradient formed three ways:
      plain      -- every rank computes its LOCAL gradient with the communicator detached, then one torch.distributed
                    all-reduce(sum) of rank_weight * gradient (mpi_adam_optimizer.py:21,39);
      overlapped -- the in-library path the timed steps use: mrl_model_grad with the communicator attached (two slices
                    all-reduced on the communication stream from inside the backward pass, ordered by events);
      recompute  -- the ranks' minibatches (gathered rows) are all-gathered, rank 0 runs each of them through the same
                    kernels on ITS device with ITS parameters and adds the results up: no collective on the gradient at all.
    A missing event wait, a wrong slice boundary or a rank that reduced a stale buffer shows as a difference between the
    first two; data that is not what the other ranks' kernels saw (or parameters that differ) as a difference to the third.
    -> dict for the JSON line; the caller fails the run above 1e-6 of the gradient's scale."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from baselines_amd import ops
    dev = model.device
    B, P, dm = model.nbatch_train, model.dm.P, model.dm
    w = float(model.mpi_rank_weight)
    idx_h = np.random.RandomState(11).permutation(N * T)[:B]              # identical on all ranks, like the reference's permutations
    idx = torch.from_numpy(idx_h).to(dev)

    def grad(obs, act, ret, val, nlp, index, t, n):
        g = torch.empty(P, dtype=torch.float32, device=dev)
        st = torch.empty(5, dtype=torch.float32, device=dev)
        dm.grad(model.params, obs, act, ret, val, nlp, index, B, t, n, hp['cliprange'], hp['ent_coef'], 0.5, g, st)
        return g

    fields = (ro.obs, ro.actions, ro.returns, ro.values, ro.neglogpacs)
    native = bool(model.native_dp)
    if native:
        dm.attach_comm(None)
    local = grad(*fields, idx, T, N)
    plain = local * w
    dist.all_reduce(plain)
    overl = None
    if native:
        dm.attach_comm(comm.native, w)
        overl = grad(*fields, idx, T, N)
    # the gathered minibatch of every rank on every rank (reference index i = e*T + t lives in storage row t*N + e)
    rows = torch.from_numpy((idx_h % T) * N + idx_h // T).to(dev)
    mine = [f.reshape((T * N,) + tuple(f.shape[2:]))[rows].contiguous() for f in fields]
    everyones = []
    for f in mine:
        if dist.get_backend() == 'nccl':
            buf = torch.empty((world,) + tuple(f.shape), dtype=f.dtype, device=dev)
            dist.all_gather_into_tensor(buf, f)
        else:                                   # gloo rehearsal on a box with fewer GPUs than ranks: gather through the host
            parts = [torch.empty(f.shape, dtype=f.dtype) for _ in range(world)]
            dist.all_gather(parts, f.cpu())
            buf = torch.stack(parts).to(dev)
        everyones.append(buf)
    out = {'minibatch': B, 'ranks': world,
           'ways': 'plain = local gradients (communicator detached) + torch.distributed all-reduce; overlapped = in-library '
                   'two-slice all-reduce inside the backward pass; recompute = rank 0 runs every rank\'s gathered minibatch itself'}
    res = torch.zeros(4, dtype=torch.float64, device=dev)                 # rank 0 fills it, everybody learns the verdict
    rank0_error = None
    if rank == 0:
        # the only asymmetric part of the check: whatever goes wrong here must still reach the broadcast below, or the other
        # ranks would wait in it until the collective timeout
        try:
            if native:
                dm.attach_comm(None)
            single = torch.zeros(P, dtype=torch.float32, device=dev)
            own_identical = None
            for r in range(world):
                g_r = grad(*[e[r] for e in everyones], None, 1, 1)
                if r == 0:
                    own_identical = bool(torch.equal(g_r, local))
                single += w * g_r                                         # bench ranks all carry weight 1 (mpi_rank_weight default)
            scale = float(plain.abs().max())
            res[0] = scale
            res[1] = float((single - plain).abs().max()) / scale
            res[2] = float((overl - plain).abs().max()) / scale if overl is not None else -1.0
            res[3] = (1.0 if own_identical else 0.0) + (2.0 if (overl is not None and torch.equal(overl, plain)) else 0.0)
        except Exception as exc:
            rank0_error = repr(exc)
            res = torch.tensor([0.0, 1e30, 1e30 if native else -1.0, 0.0], dtype=torch.float64, device=dev)    # finite: the line stays JSON
        finally:
            if native:
                dm.attach_comm(comm.native, w)
    dist.broadcast(res, 0)
    r_ = [float(x) for x in res.cpu()]
    out.update({'grad_scale': r_[0], 'recompute_vs_plain_max_abs_diff_over_scale': r_[1],
                'overlapped_vs_plain_max_abs_diff_over_scale': (r_[2] if r_[2] >= 0 else None),
                'overlapped_equals_plain': (r_[2] <= 1e-6 if r_[2] >= 0 else None),
                'overlapped_bit_identical_to_plain': (bool(int(r_[3]) & 2) if r_[2] >= 0 else None),
                'rank0_gathered_rows_reproduce_its_indexed_gradient_bitwise': bool(int(r_[3]) & 1),
                'max_abs_diff_over_scale': max(r_[1], r_[2] if r_[2] >= 0 else 0.0)})
    if rank0_error:
        out['rank0_error'] = rank0_error
    if not native:
        out['note'] = 'in-library communicator not in use (%s): the overlapped path was not exercised' % (
            getattr(comm, 'native_error', None) or dist.get_backend())
    del everyones, mine
    torch.cuda.empty_cache()
    return out


def run_ppo2(workload, total_envs, T, steps, warmup, chunk, world, rank, comm, want_prof):
    """-> dict with value / timing / roofline pieces of one PPO2-update measurement"""
    import numpy as np
    import torch
    from baselines_amd import _lib, ops
    from baselines_amd.common import set_global_seeds
    from baselines_amd.common.policies import build_policy
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv
    from baselines_amd.ppo2 import Model, Runner

    recurrent = workload == 'atari_lstm'
    if workload in ('atari', 'atari_lstm'):
        hp = dict(noptepochs=4, nminibatches=4, ent_coef=0.01, lr=2.5e-4, cliprange=0.1,
                  network='cnn_lstm' if recurrent else 'cnn', value_network=None)
        flops_per_sample_visit = 49.526e6      # SURVEY.md App. B (fwd+bwd); cnn_lstm: + 6 * (512 + 128) * 512 per sample-visit
        if recurrent:
            flops_per_sample_visit += 6.0 * (512 + 128) * 512
    else:
        hp = dict(noptepochs=10, nminibatches=32, ent_coef=0.0, lr=3e-4, cliprange=0.2, network='mlp', value_network='copy')
        flops_per_sample_visit = 248.6e3
    assert total_envs % world == 0
    N = total_envs // world
    nbatch = N * T
    nbatch_train = nbatch // hp['nminibatches']

    set_global_seeds(0)
    env = SyntheticVecEnv('atari' if recurrent else workload, N, seed=1000 + rank)
    policy = build_policy(env, hp['network'], value_network=hp['value_network'])
    model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N,
                  nbatch_train=nbatch_train, nsteps=T, ent_coef=hp['ent_coef'], vf_coef=0.5, max_grad_norm=0.5,
                  comm=comm, chunk=chunk)
    runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # rollouts fill the HBM rollout buffer and give the "full iteration" figure: the first one runs the step loop, the
    # second captures the rollout as a hipGraph (Runner._run_device_env), the third -- the timed one -- replays it
    for _ in range(2):
        runner.run()
    sync()
    t0 = time.perf_counter()
    rollout_out = runner.run()
    sync()
    t_rollout = time.perf_counter() - t0
    ro = runner.rollout
    mb_states = rollout_out[6] if recurrent else None          # LSTM state of every env BEFORE the rollout (runner.py:23)
    last_values = (model.value_dev(runner.obs, runner._states_dev, runner._dones_dev) if recurrent
                   else model.value_dev(runner.obs))

    # self-check: gradient + statistics of one minibatch of the benched shape, computed in the benched
    # chunking (one 131072-sample chunk) and again in 15360-sample chunks must agree: statistics to 1e-5, EVERY gradient entry to 1e-5
    # of the gradient's scale.
    # Why 15360: the class-resident conv forward alternates the sign of staged rows over tiles of 3 (conv2) / 5 (conv3) whole images
    # (DESIGN.md 3.1 / 3.7), so the last bits of its output depend on where an image sits in its tile.  With a chunk size that 3, 5, 128
    # and 256 divide (15360; the 8192-sample remainder starts at sample 122880 = 15 * 8192) every image keeps its tile position and the
    # two chunkings differ only in the order of the per-chunk weight-gradient sums (~1e-7 of the scale).  Round 5 compared against chunk
    # 8192 (8192 % 3 = 2): a ReLU unit whose pre-activation is zero to rounding then took different sides in the two runs, and the
    # gradient bound had to be relaxed to a gross-error guard (1e-3) -- which a subtly wrong tile could have hidden behind.
    self_check = None
    SC_CHUNK = 15360
    if workload == 'atari' and world == 1 and nbatch_train > SC_CHUNK and os.environ.get('MRL_BENCH_SELF_CHECK', '1') != '0':
        ops.gae(ro.rewards, ro.values, ro.dones, last_values, runner._dones_dev, 0.99, 0.95, out=ro.returns)
        idx = torch.from_numpy(np.random.RandomState(7).permutation(nbatch)[:nbatch_train]).to(model.device)
        dm2 = ops.DeviceModel(chunk=SC_CHUNK, device=model.device, **policy.device_model_kwargs())
        outs = []
        for dmx in (model.dm, dm2):
            g = torch.empty(dmx.P, dtype=torch.float32, device=model.device)
            st5 = torch.empty(5, dtype=torch.float32, device=model.device)
            dmx.grad(model.params, ro.obs, ro.actions, ro.returns, ro.values, ro.neglogpacs, idx, nbatch_train, T, N,
                     hp['cliprange'], hp['ent_coef'], 0.5, g, st5)
            outs.append((g, st5))
        gscale = float(outs[1][0].abs().max())
        diff = (outs[0][0] - outs[1][0]).double()
        self_check = {'what': 'mrl_model_grad of one %d-sample minibatch: chunk %d vs chunk %d' % (nbatch_train, model.dm.chunk, SC_CHUNK),
                      'stats_max_abs_diff': float((outs[0][1] - outs[1][1]).abs().max()),
                      'grad_max_abs_diff_over_scale': float(diff.abs().max()) / gscale,
                      'grad_rel_l2_diff': float(diff.norm()) / float(outs[1][0].double().norm()),
                      'entries_above_1e-5_of_scale': int((diff.abs() > 1e-5 * gscale).sum()),
                      'entries': int(diff.numel()),
                      'grad_scale': gscale}
        assert (self_check['stats_max_abs_diff'] <= 1e-5 and self_check['grad_max_abs_diff_over_scale'] <= 1e-5
                and self_check['entries_above_1e-5_of_scale'] == 0), self_check
        del dm2, outs, g, st5
        torch.cuda.empty_cache()

    dp_check = None
    if world > 1 and workload == 'atari' and os.environ.get('MRL_BENCH_DP_VERIFY', '1') != '0':
        ops.gae(ro.rewards, ro.values, ro.dones, last_values, runner._dones_dev, 0.99, 0.95, out=ro.returns)
        # The verdict travels in the JSON line (`dp_verify.ok`); MRL_BENCH_DP_STRICT=1 turns a failed check into a failed run.  The
        # default keeps the measurement: this code path meets its second GPU on the driver's node for the first time, and a scaling
        # curve with `ok: false` next to it says more than no curve.  (The checks are collective and symmetric: every rank takes
        # the same branch, so an exception inside them is raised on all ranks or on none.)
        try:
            dp_check = dp_verify(model, ro, T, N, hp, world, rank, comm)
            dp_check['ok'] = bool(dp_check['max_abs_diff_over_scale'] <= 1e-6)
        except Exception as exc:                      # reported, never silently dropped
            dp_check = {'ok': False, 'error': repr(exc)}
            if model.native_dp:                       # the check detaches the communicator in places: leave it as the timed region expects
                model.dm.attach_comm(comm.native, float(model.mpi_rank_weight))
        if not dp_check['ok']:
            sys.stderr.write('bench.py: data-parallel gradient self-check FAILED: %s\n' % json.dumps(dp_check))
            if os.environ.get('MRL_BENCH_DP_STRICT', '0') == '1':
                raise SystemExit(3)
            # the overlapped in-library all-reduce disagreed while the plain one matches rank 0's recomputation: time the plain path
            # (same verdict on every rank -- it was broadcast -- so every rank takes this branch or none does)
            if (model.native_dp and dp_check.get('overlapped_equals_plain') is False
                    and dp_check.get('recompute_vs_plain_max_abs_diff_over_scale', 1.0) <= 1e-6):
                model.dm.attach_comm(None)
                model.native_dp = False
                dp_check['fallback'] = ('overlapped path rejected: the timed region all-reduces with torch.distributed after the '
                                        'backward pass (Model._apply_gradients)')

    def update():
        """the PPO2 update (ppo2.py:142 GAE part + :154-166)"""
        ops.gae(ro.rewards, ro.values, ro.dones, last_values, runner._dones_dev, 0.99, 0.95, out=ro.returns)
        inds = np.arange(nbatch)
        stats = []
        if recurrent:
            # env-wise minibatches (ppo2.py:167-180): whole trajectories of N / nminibatches shuffled envs, BPTT over nsteps
            per = N // hp['nminibatches']
            envinds, flat = np.arange(N), np.arange(N * T).reshape(N, T)
            for _ in range(hp['noptepochs']):
                np.random.shuffle(envinds)
                for lo in range(0, N, per):
                    mbe = envinds[lo:lo + per]
                    stats.append(model.train_indexed(hp['lr'], hp['cliprange'], ro, torch.from_numpy(flat[mbe].ravel()).to(model.device),
                                                     states=mb_states[torch.from_numpy(mbe).to(mb_states.device)]))
            return torch.stack(stats).mean(dim=0)
        for _ in range(hp['noptepochs']):
            np.random.shuffle(inds)
            inds_dev = model.indices_to_device(inds)          # asynchronous upload, as in ppo2.learn
            stats.extend(model.train_epoch(hp['lr'], hp['cliprange'], ro, inds_dev).unbind(0))
        return torch.stack(stats).mean(dim=0)

    # The launch-bound MLP workload runs each epoch as one replayed hipGraph (Model.train_epoch); HIP events cannot be
    # recorded inside a graph, so there the per-kernel times come from one extra, untimed, eager update after the
    # timed region.  The Atari workload (the metric) launches every kernel individually: events sit inside the timed region.
    graph_mode = (workload == 'mujoco' and world == 1 and os.environ.get('MRL_EPOCH_GRAPH', '1') != '0')
    for _ in range(warmup):
        update()
    if graph_mode:
        update()                                   # the first update after the eager warm-up captures; keep that out too
    if want_prof and not graph_mode:
        _lib.prof_enable(True)
    # host hygiene for the launch-bound configurations: whatever earlier workloads of this process left behind is collected
    # NOW and the survivors are frozen, so that no full collection of the interpreter (~20 ms in this process) lands inside a
    # timed region whose updates take 18 ms
    import gc
    gc.collect()
    gc.freeze()
    try:
        sampler = DeviceStateSampler().start() if (rank == 0 and os.environ.get('MRL_BENCH_SMI', '1') != '0') else None
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            lossvals = update()
        sync()
        t1 = time.perf_counter()
    finally:
        gc.unfreeze()                              # the frozen set is this timing window's only (config.host_gc says so)
    dt = t1 - t0
    device_state = sampler.stop(t0, t1) if sampler else None
    prof = {}
    prof_steps = steps
    if want_prof:
        if graph_mode:
            _lib.prof_enable(True)
            update()
            sync()
            prof_steps = 1
        _lib.prof_enable(False)
        prof = _lib.prof_report()
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    lossvals = lossvals.cpu().numpy()
    assert np.all(np.isfinite(lossvals)), lossvals
    params_synced = None
    if world > 1:              # every rank must end with bit-identical parameters (mpi_adam_optimizer.py:53-68)
        import torch.distributed as dist
        ck = torch.stack([model.params.double().sum(), model.params.double().abs().sum()])
        lo, hi = ck.clone(), ck.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        params_synced = bool(torch.equal(lo, hi))
    res = dict(value=total_envs * T * steps / dt, dt=dt, steps=steps, P=int(model.dm.P), t_rollout=t_rollout, loss=[float(x) for x in lossvals], prof=prof,
               prof_steps=prof_steps, hp=hp, N=N, nbatch_train=nbatch_train, chunk=model.dm.chunk, graph_mode=graph_mode,
               model_tflops=flops_per_sample_visit * total_envs * T * hp['noptepochs'] * steps / dt / 1e12,
               native_dp=bool(getattr(model, 'native_dp', False)), device_state=device_state,
               self_check=self_check, params_synced=params_synced, dp_verify=dp_check)
    del runner, model, env, ro
    torch.cuda.empty_cache()
    return res


TRACE_LABELS = (('mlp_step_kernel', 'mlp_step'), ('mlp_tail_kernel', 'mlp_tail'), ('reduce_slabs_kernel', 'reduce_slabs'),
                ('adam_kernel', 'clip+adam'), ('mlp_advstat_kernel', 'advstat'), ('gae_kernel', 'gae'), ('gae_lane_kernel', 'gae'))


def graph_kernel_trace(workload, num_envs, nsteps):
    """Per-kernel execution times of a workload whose epochs run as REPLAYED hipGraphs (HIP events cannot be recorded inside a
    graph, and an eager update with events around every 40-us kernel reports more than the replayed step takes): a child
    `rocprofv3 --kernel-trace` run of this script on the same workload; the window = whole updates (gae kernel to gae kernel)
    of the child's timed region.  -> ({label: {'ms_per_update', 'launches_per_update', 'avg_us'}}, child ms_per_step) or
    (None, reason)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not found'
    d = tempfile.mkdtemp(prefix='mrl_trace_', dir='/tmp')
    try:
        cmd = ['rocprofv3', '--kernel-trace', '-d', d, '-o', 't', '--', sys.executable, os.path.abspath(__file__), '--workload', workload,
               '--num-envs', str(num_envs), '--nsteps', str(nsteps), '--steps', '4', '--warmup', '2', '--no-prof', '--no-cpu-baseline',
               '--no-other-configs']
        env = dict(os.environ, MRL_BENCH_SMI='0', TMPDIR='/tmp', MRL_BENCH_TRACE_CHILD='1')
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd='/tmp')
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
        if out.returncode != 0 or not lines:
            return None, 'traced child run failed (rc %d): %s' % (out.returncode, out.stderr[-300:])
        child = json.loads(lines[-1])
        dbs = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)
        if not dbs:
            return None, 'no rocpd database written'
        rows = sqlite3.connect(dbs[0]).execute('select name, start, end from kernels order by start').fetchall()
        gae = [i for i, r in enumerate(rows) if 'gae_kernel' in r[0] or 'gae_lane_kernel' in r[0]]      # one per update: the window markers
        if len(gae) < 3:
            return None, 'fewer than 3 updates in the trace'
        lo, hi = gae[-1 - child['steps']], gae[-1]          # the child's timed updates: whole updates, gae kernel to gae kernel
        nupd = child['steps']
        agg = {}
        for name, t0, t1 in rows[lo:hi]:
            label = next((lab for key, lab in TRACE_LABELS if key in name), None)
            if label is None:                              # torch's copy / elementwise kernels between the epochs
                label = 'other:' + name.split('(')[0].split('<')[0].split('::')[-1][-40:]
            a = agg.setdefault(label, [0, 0.0])
            a[0] += 1
            a[1] += (t1 - t0) * 1e-6
        res = {k: {'ms_per_update': v[1] / nupd, 'launches_per_update': v[0] / float(nupd), 'avg_us': v[1] / v[0] * 1e3}
               for k, v in agg.items()}
        return res, child['ms_per_step']
    except Exception as exc:
        return None, repr(exc)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def apply_graph_trace(res, workload, total_envs, T):
    """graph-mode workloads: replace the eager per-kernel times in res['prof'] by the kernel trace of the replayed graphs (the
    eager run keeps supplying each label's flops / bytes per launch); never leaves a breakdown that exceeds its own step"""
    if not res.get('graph_mode') or not res.get('prof') or os.environ.get('MRL_BENCH_TRACE_CHILD'):
        return
    eager = res['prof']
    tr, info = graph_kernel_trace(workload, total_envs, T) if os.environ.get('MRL_BENCH_GRAPH_TRACE', '1') != '0' else (None, 'disabled')
    if tr:
        prof = {}
        for label, v in tr.items():
            e = eager.get(label, {'count': 0, 'flops': 0.0, 'bytes': 0.0})
            per = (e['flops'] / e['count'], e['bytes'] / e['count']) if e['count'] else (0.0, 0.0)
            n = v['launches_per_update']
            prof[label] = {'ms': v['ms_per_update'], 'count': n, 'flops': per[0] * n, 'bytes': per[1] * n}
        res['prof'], res['prof_steps'] = prof, 1
        res['kernel_times_source'] = ('rocprofv3 --kernel-trace of a child run of the same workload (replayed epoch graphs), whole '
                                      'updates of its timed region; that run took %.3f ms per update' % info)
        res['traced_ms_per_step'] = info
        # a trace that exceeds its own run is reported by check_breakdown (`breakdown_check`), it never costs the JSON line
    else:
        # no trace: the eager update's shares, scaled onto the timed graph-replay step (its kernels take longer with an event
        # pair around each of them than inside the replayed graph)
        tot = sum(v['ms'] for v in eager.values()) / max(1, res['prof_steps'])
        step = res['dt'] / res['steps'] * 1e3
        scale = min(1.0, step / tot) if tot > 0 else 1.0
        for v in eager.values():
            v['ms'] *= scale
        res['kernel_times_source'] = ('one eager update with HIP events after the timed region, times scaled by %.3f onto the '
                                      'replayed step (no kernel trace: %s)' % (scale, info))


class DeviceStateSampler:
    """rocm-smi samples (shader clock, socket power) taken on a host thread while the timed region runs.  The update is
    power-limited on MI355X (~1.3 kW, sclk well below the 2.4 GHz the nominal matrix-pipe peaks are quoted at), so the
    roofline fractions against nominal peaks understate what the kernels reach at the clock they are given."""

    def __init__(self, period=0.4):
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        while not self._stop.is_set():
            try:
                out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
                sclk = re.search(r'sclk clock level:\s*\d+:\s*\((\d+)Mhz\)', out)
                pw = re.search(r'Power \(W\):\s*([0-9.]+)', out)
                if sclk and pw:
                    self.samples.append((time.perf_counter(), int(sclk.group(1)), float(pw.group(1))))
            except Exception:
                pass
            self._stop.wait(self.period)

    def start(self):
        self._th.start()
        return self

    def stop(self, t0, t1):
        self._stop.set()
        self._th.join(timeout=6)
        inside = sorted((c, p) for (t, c, p) in self.samples if t0 <= t <= t1)
        if not inside:
            return None
        clk = sorted(c for c, _ in inside)
        pw = sorted(p for _, p in inside)
        return {'sclk_mhz_median': clk[len(clk) // 2], 'socket_power_w_median': pw[len(pw) // 2], 'samples': len(inside),
                'source': 'rocm-smi --showclocks --showpower on a host thread during the timed region'}


def check_breakdown(kernel_ms, ms_per_step, roof, what):
    """a per-kernel breakdown is evidence only if it fits inside the step it was measured in.
    -> None when it does, else the violation as text: the caller records it in the JSON line (`breakdown_check`) and on stderr
    instead of losing the whole measurement to an exception; tests/test_bench_contract.py asserts on the committed line."""
    tot = sum(kernel_ms.values())
    bad = []
    if tot > ms_per_step * 1.002 + 0.01:
        bad.append('%s: kernel times sum to %.3f ms of a %.3f ms step' % (what, tot, ms_per_step))
    if roof and roof.get('avg_ms') and roof.get('launches'):
        if roof['avg_ms'] * roof['launches'] > ms_per_step * max(1, roof.get('prof_steps', 1)) * 1.002 + 0.01:
            bad.append('%s: dominant kernel %.3f ms x %d launches exceeds its steps' % (what, roof['avg_ms'], roof['launches']))
    if bad:
        sys.stderr.write('bench.py: breakdown check FAILED: %s\n' % '; '.join(bad))
        return '; '.join(bad)
    return None


def dominant_roofline(res, sites, workload):
    """roofline object of the kernel with the largest share of the timed region"""
    prof = res['prof']
    if not prof:
        return None, {}
    atari = workload == 'atari'          # counter traffic exists for the feed-forward NatureCNN step at the full-size minibatch
    pmc, pmc_src = load_pmc(res['nbatch_train']) if atari else ({}, None)
    if workload in ('atari', 'atari_lstm'):
        # the conv / fc1 sites of the recurrent row are the same engines on the same pipes; its LSTM GEMMs are fp32-MFMA sites
        alg = nature_cnn_alg_bytes(res['nbatch_train'])
        if workload == 'atari_lstm':
            alg.update(lstm_alg_bytes(res['nbatch_train']))
        per = kernel_rooflines(prof, sites, alg, pmc)
    else:
        per = kernel_rooflines(prof, {}, mlp_step_alg_bytes(res['nbatch_train'], res['P']), pmc)
    tot_ms = sum(v['ms'] for v in prof.values())
    dom = max(per, key=lambda k: prof[k]['ms'])
    roof = dict(per[dom])
    roof['kernel'] = dom
    roof['prof_steps'] = res['prof_steps']           # `launches` are counted over this many timed steps
    roof.setdefault('traffic', None)
    roof['share_of_kernel_time'] = prof[dom]['ms'] / tot_ms
    # HBM-side bytes per launch: PMC counters cannot be collected from inside this process; they come from the committed
    # `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes over one epoch of the same shape AND the same kernel sources
    # (scripts/pmc_epoch.sh -> profiles/*_pmc_hbm.json, gfx950 corrections applied by scripts/pmc_to_json.py)
    if pmc_src:
        roof['traffic_source'] = pmc_src
    if atari and pmc:
        tot_tr = sum(per[k]['traffic'] * prof[k]['count'] for k in per if per[k].get('traffic'))
        tot_alg = sum(per[k]['alg_bytes'] * prof[k]['count'] for k in per if per[k].get('traffic'))
        if tot_alg > 0:
            roof['all_gemm_sites_traffic_over_algorithmic'] = tot_tr / tot_alg
    return roof, per


def replay_config():
    """config 5 (DQN slice): scripts/bench_replay.py in a child process (its 56 GB ring is freed with the process)"""
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'bench_replay.py')], capture_output=True,
                             text=True, timeout=600)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
        r = json.loads(line)
    except Exception as exc:          # reported, never silently dropped
        return {'workload': 'deepq replay 1M transitions', 'error': repr(exc)}
    b = r.get('batch_4096', {})
    g = b.get('replay_gather', {})
    return {'workload': 'deepq Pong-shaped replay, 1M transitions resident in HBM, prioritized sampling + TD kernel',
            'metric': 'sampled transitions/s (learner step: PER sample + gather + TD + priority update, batch 4096)',
            'value': 4096 / (b['wall_us_per_learner_step'] * 1e-6) if b else None, 'unit': 'transitions/s',
            'roofline': ({'bound': 'hbm', 'kernel': 'replay_gather', 'achieved': g.get('GB/s'), 'peak': PEAK_HBM_GBS,
                          'unit': 'GB/s', 'frac': (g.get('GB/s') or 0) / PEAK_HBM_GBS, 'traffic': None} if g else None),
            'detail': r}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='atari', choices=['atari', 'mujoco', 'atari_lstm'])
    ap.add_argument('--num-envs', type=int, default=None, help='whole-job num_envs (default 4096 atari / 1024 mujoco)')
    ap.add_argument('--nsteps', type=int, default=128)
    ap.add_argument('--chunk', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='do not record HIP events per kernel in the timed region')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the short runs of the other BASELINE configs')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        _relaunch(args)

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    comm, collective, observed_world = None, None, 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # one process per GPU over RCCL ("nccl" on ROCm).  MRL_BENCH_BACKEND=gloo lets the N>1 code path be
        # smoke-tested on a box with fewer GPUs than ranks (ranks then share devices; not a measurement).
        backend = os.environ.get('MRL_BENCH_BACKEND', 'nccl')
        ndev = torch.cuda.device_count()
        if backend == 'nccl' and ndev < world:
            raise SystemExit('--gpus %d but only %d HIP devices are visible' % (world, ndev))
        dev = local_rank % max(1, ndev)
        torch.cuda.set_device(dev)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
        else:
            dist.init_process_group(backend)
        observed_world = dist.get_world_size()
        from baselines_amd.common.dist import Comm
        comm = Comm()
        collective = 'torch.distributed all_reduce (%s), after the backward pass' % backend
        if backend == 'nccl' and os.environ.get('MRL_NATIVE_COMM', '1') != '0':
            # collective and fail-safe (common/dist.py): id broadcast, communicator creation and a probe all-reduce are each
            # agreed on by all ranks; any failure leaves every rank on the torch.distributed collectives
            if comm.enable_native():
                collective = ('in-library RCCL (mrl_comm): fc1+heads slice all-reduced on a communication stream while the '
                              'conv layers are back-propagated, rest at the end; / total weight, clip, Adam replicated')
            else:
                collective += ' [in-library RCCL communicator unavailable on some rank: %s]' % comm.native_error
    else:
        torch.cuda.set_device(0)

    from baselines_amd import _lib
    sites, arith_text, arith_short = arithmetic_mode(_lib)
    total_envs = args.num_envs or (4096 if args.workload == 'atari' else 1024)
    T = args.nsteps
    res = run_ppo2(args.workload, total_envs, T, args.steps, args.warmup, args.chunk, world, rank, comm, not args.no_prof)

    if rank == 0:
        hp = res['hp']
        apply_graph_trace(res, args.workload, total_envs, T)
        roof, per = dominant_roofline(res, sites, args.workload)
        ds = res.get('device_state')
        if roof and ds and roof.get('bound') == 'mfma' and ds.get('sclk_mhz_median'):
            # informational: `peak` and `frac` stay the nominal 2.4 GHz figures of MI355X_MICROARCH.md
            roof['frac_at_sampled_sclk'] = roof['frac'] * 2400.0 / ds['sclk_mhz_median']
        out = {
            'metric': 'env-steps/sec (whole node) PPO2 update, num_envs=%d nsteps=%d' % (total_envs, T),
            'value': res['value'], 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': res['dt'] / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'ppo2 update-only (GAE + %dx%d minibatch steps) %s-shaped %s num_envs=%d nsteps=%d'
                                   % (hp['noptepochs'], hp['nminibatches'], args.workload, hp['network'], total_envs, T),
                       'envs_per_gpu': res['N'], 'nbatch_train_per_gpu': res['nbatch_train'], 'chunk': res['chunk'],
                       'arithmetic_mode': arith_short, 'arithmetic': arith_text,
                       'parallelism': 'dp%d (envs sharded, 1 all-reduce of the flat gradient per minibatch step)' % world,
                       'world_size_observed': observed_world, 'collective': collective,
                       # north_star's "Adam fused into the gradient kernel" (SURVEY row n1): measured and declined, DESIGN.md 3.6 / 8
                       'adam': 'separate launch, measured: clip + Adam is ONE 28*P-byte pass right behind the gradient reduction; a '
                               'last-arriving-workgroup tail inside the gradient kernel needs a grid-wide release/acquire across 8 '
                               'non-coherent L2s (~3.5 us) -- more than the 1.5-1.9 us kernel boundary it would replace'},
            'model_tflops': res['model_tflops'],
            # power-limited: the sampled shader clock sits well below the 2.4 GHz the nominal peaks are quoted at
            'device_state': res.get('device_state'),
            'full_iteration_env_steps_per_s': total_envs * T / (res['t_rollout'] + res['dt'] / args.steps),
            'rollout_s': res['t_rollout'],
            'loss': res['loss'],
            'roofline': roof,
        }
        if res.get('self_check'):
            out['self_check'] = res['self_check']
            # a compact copy inside `config`: the driver's record keeps `config` and `roofline` whole and only the NAMES of other keys
            sc = res['self_check']
            out['config']['self_check'] = {k: sc[k] for k in ('stats_max_abs_diff', 'grad_max_abs_diff_over_scale',
                                                              'entries_above_1e-5_of_scale', 'unexplained_entries') if k in sc}
        if res.get('device_state'):
            out['config']['device_state'] = {k: res['device_state'].get(k) for k in ('sclk_mhz_median', 'socket_power_w_median')
                                             if k in res['device_state']}
        if world > 1:
            out['params_synced_across_ranks'] = res.get('params_synced')
            out['dp_verify'] = res.get('dp_verify')
            out['config']['native_dp'] = res.get('native_dp')
            dpv = res.get('dp_verify') or {}
            if dpv.get('fallback'):
                out['config']['collective'] = ('torch.distributed all_reduce (%s), after the backward pass [in-library overlapped path '
                                               'REJECTED by dp_verify]' % backend)
            # impossible to misread: which collective the timed region ran and whether the gradient it
            # produced was verified, at the top level next to `value` and inside `config`
            out['collective'] = ('torch.distributed (overlapped path REJECTED)' if dpv.get('fallback') else
                                 'in-library RCCL, overlapped with the backward pass' if res.get('native_dp') else
                                 'torch.distributed (%s), after the backward pass' % backend)
            out['dp_verify_ok'] = bool(dpv.get('ok')) if dpv else None
            out['config']['dp_verify_ok'] = out['dp_verify_ok']
        if res['graph_mode']:
            out['config']['launch'] = 'one replayed hipGraph per epoch (32 minibatch steps)'
            out['kernel_times_source'] = res.get('kernel_times_source')
        if res['prof']:
            out['kernel_ms_per_step'] = {k: round(v['ms'] / res['prof_steps'], 3) for k, v in
                                         sorted(res['prof'].items(), key=lambda kv: -kv[1]['ms'])}
            out['breakdown_check'] = check_breakdown(out['kernel_ms_per_step'], res.get('traced_ms_per_step') or out['ms_per_step'], roof,
                                                     out['config']['workload']) or 'ok: kernel times fit inside the timed step'
            out['kernel_rooflines'] = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                                       for k, v in per.items()}
        default_run = (world == 1 and args.workload == 'atari' and args.num_envs is None and args.nsteps == 128
                       and args.chunk is None)
        if default_run and not args.no_other_configs:
            # BASELINE.json's other single-GPU configurations, short runs (3 timed updates each), own rooflines
            others = []
            def other_row(wl, n_envs):
                nst, nwu = (12, 3) if wl == 'mujoco' else (3, 1)      # the 28 ms MLP update needs a longer window to time stably
                r = run_ppo2(wl, n_envs, 128, nst, nwu, None, 1, 0, None, not args.no_prof)
                apply_graph_trace(r, wl, n_envs, 128)
                rf, _ = dominant_roofline(r, sites, wl)
                name = 'ppo2 update-only %s-shaped %s num_envs=%d nsteps=128' % (wl.split('_')[0], r['hp']['network'], n_envs)
                if wl == 'atari' and n_envs == 512:
                    # what ONE rank of BASELINE config 4 computes per update (4096 envs sharded over 8 GPUs: 512 envs, 16384-sample
                    # minibatches); the 8-GPU job adds 16 all-reduces of the 6.75 MB flat gradient per update, overlapped with the
                    # conv layers' backward pass
                    name += " (config 4's per-GPU shard: minibatch 16384)"
                row = {'workload': name,
                       'value': r['value'], 'unit': 'env-steps/s', 'ms_per_step': r['dt'] / nst * 1e3, 'steps': nst,
                       'device_state': r.get('device_state'),          # this configuration's own rocm-smi samples
                       'full_iteration_env_steps_per_s': n_envs * 128 / (r['t_rollout'] + r['dt'] / nst),
                       'roofline': rf,
                       'kernel_ms_per_step': {k: round(v['ms'] / r['prof_steps'], 3) for k, v in
                                              sorted(r['prof'].items(), key=lambda kv: -kv[1]['ms'])}}
                if r.get('kernel_times_source'):
                    row['kernel_times_source'] = r['kernel_times_source']
                if wl == 'atari' and n_envs == 512:
                    row['projection_8_gpus'] = {
                        'per_gpu_update_ms': row['ms_per_step'], 'allreduce_bytes_per_minibatch_step': 4 * r['P'],
                        'whole_job_env_steps_per_s_if_the_allreduce_hides': 4096 * 128 / (row['ms_per_step'] * 1e-3),
                        'note': 'upper bound from the measured single-GPU shard; the driver measures the real 8-GPU number'}
                if r['prof']:
                    row['breakdown_check'] = check_breakdown(row['kernel_ms_per_step'], r.get('traced_ms_per_step') or row['ms_per_step'], rf, name) or 'ok'
                return row
            # the headline above is already measured: a failure in one of the secondary rows is recorded in that row, not raised
            for wl, n_envs in (('mujoco', 1024), ('atari', 256), ('atari', 512), ('atari_lstm', 256)):
                try:
                    others.append(other_row(wl, n_envs))
                except Exception as exc:
                    sys.stderr.write('bench.py: other config %s num_envs=%d FAILED: %r\n' % (wl, n_envs, exc))
                    others.append({'workload': 'ppo2 update-only %s num_envs=%d nsteps=128' % (wl, n_envs), 'error': repr(exc)})
            try:
                others.append(replay_config())
            except Exception as exc:
                sys.stderr.write('bench.py: replay config FAILED: %r\n' % (exc,))
                others.append({'workload': 'deepq replay', 'error': repr(exc)})
            out['other_configs'] = others
        if world == 1 and not args.no_cpu_baseline and args.workload in ('atari', 'mujoco'):
            try:
                out['cpu_baseline'] = cpu_baseline(args.workload)
            except Exception as exc:
                sys.stderr.write('bench.py: cpu_baseline FAILED: %r\n' % (exc,))
                out['cpu_baseline'] = {'value': None, 'unit': 'env-steps/s', 'cores': 0, 'kind': 'port', 'sample': 'failed: %r' % (exc,)}
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
