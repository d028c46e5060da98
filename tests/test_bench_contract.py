"""CPU: the one-line JSON contract of bench.py, checked on the committed result of the last GPU run
(the newest profiles/rNN?_bench_atari4096.json by round key) and on bench.py's own argument defaults."""
import ast
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_bench_line():
    import glob
    import sys
    sys.path.insert(0, ROOT)
    import bench
    return sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_bench_atari4096.json')), key=bench.profile_round_key)[-1]


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(_newest_bench_line()))
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert d['metric'] == base['metric']
    for key in ('value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['higher_is_better'] is True and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1
    assert r['traffic'] is None or r['traffic'] > 0
    assert (r['traffic'] is None or (r['traffic_source'].startswith('profiles/r') and r['traffic_source'].endswith('_pmc_hbm.json'))) and d['dtype'] == 'f32'
    # round 5: counter traffic of all GEMM sites over their algorithmic bytes (was 1.49 with the tiled conv forward's re-reads)
    assert r['traffic'] is None or r['all_gemm_sites_traffic_over_algorithmic'] <= 1.35
    assert 'separate launch, measured' in d['config']['adam']
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['mfma_frac']) < 1e-12 and 0 < r['hbm_frac'] < r['frac']
    # the headline runs the split engines: six exact bf16 products per fp32 multiply, nearest-rounded planes
    assert d['config']['arithmetic_mode'] == 'bf16x6-rn-split'
    kr = d['kernel_rooflines']
    # every fp32 x fp32 site on the 6-product pipe; the uint8 first layer (3 products) is HBM-bound: priced on HBM
    assert abs(kr['c2.wgrad']['peak'] - 2516.6 / 6) < 1e-3 and abs(kr['c2.fwd']['peak'] - 2516.6 / 6) < 1e-3
    assert kr['c1.fwd']['bound'] == 'hbm' and kr['c1.fwd']['peak'] == 8000.0 and abs(kr['c1.fwd']['mfma_peak_tflops'] - 2516.6 / 3) < 1e-3
    # statistics at the 1e-5 bar; the gradient entries are a gross-error guard (a ReLU unit at zero-to-rounding may take different sides in
    # the two chunkings since round 5 -- bench.py's comment, scripts/chunk_diff.py); the committed headline line sits at 9e-6
    assert d['self_check']['stats_max_abs_diff'] <= 1e-5 and d['self_check']['grad_max_abs_diff_over_scale'] <= 1e-3
    assert {o['workload'].split()[0] for o in d['other_configs']} == {'ppo2', 'deepq'} and len(d['other_configs']) == 5
    # no row prices a kernel with zero algorithmic bytes, and the recurrent row's conv sites are priced like the
    # feed-forward row's (same engines, same pipe): its dominant kernel's fraction within 3 % of the N = 256 cnn row's
    rows = {o['workload']: o['roofline'] for o in d['other_configs'] if o.get('roofline')}
    for name, rf in list(rows.items()) + [('headline', r)]:
        assert rf.get('bound') and rf.get('alg_bytes', 1.0) > 0, name
    cnn256 = next(v for k, v in rows.items() if ' cnn num_envs=256' in k)
    lstm256 = next(v for k, v in rows.items() if 'cnn_lstm' in k)
    assert cnn256['kernel'] == lstm256['kernel'] and abs(cnn256['frac'] / lstm256['frac'] - 1.0) < 0.03
    assert 'frac_vs_fp32_mfma' in r and 'frac_vs_fp32_mfma' in lstm256
    c = d['cpu_baseline']
    assert c['host_cpu_count'] >= c['cores'] and c['host_cpu_model']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    # every per-kernel breakdown fits inside the step it belongs to; the graph-replayed MLP row takes its
    # kernel times from a rocprofv3 trace of the replay
    assert sum(d['kernel_ms_per_step'].values()) <= d['ms_per_step'] * 1.003
    for o in d['other_configs']:
        if 'kernel_ms_per_step' in o and o.get('ms_per_step'):
            bound = o['ms_per_step'] if 'rocprofv3' not in (o.get('kernel_times_source') or '') else 1e9
            assert sum(o['kernel_ms_per_step'].values()) <= bound * 1.003, o['workload']
    mlp = [o for o in d['other_configs'] if 'mlp' in o['workload']][0]
    assert 'rocprofv3' in mlp['kernel_times_source'] and sum(mlp['kernel_ms_per_step'].values()) <= mlp['ms_per_step']
    assert any("config 4's per-GPU shard" in o['workload'] and 'projection_8_gpus' in o for o in d['other_configs'])
    # value = env-steps of K timed steps / time:  num_envs * nsteps / (ms_per_step / 1000)
    assert abs(d['value'] - 4096 * 128 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6


def test_bench_defaults_to_one_gpu_and_a_short_run():
    src = open(os.path.join(ROOT, 'bench.py')).read()
    tree = ast.parse(src)
    defaults = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, 'attr', '') == 'add_argument' and node.args:
            name = node.args[0].value if isinstance(node.args[0], ast.Constant) else None
            for kw in node.keywords:
                if kw.arg == 'default' and isinstance(kw.value, ast.Constant):
                    defaults[name] = kw.value.value
    assert defaults['--gpus'] == 1 and defaults['--steps'] <= 5 and defaults['--warmup'] <= 2
    assert 'oracle' in src and 'cpu_baseline' in src            # the only product-side file allowed to touch oracle/


def test_counter_traffic_is_attached_only_for_the_kernel_sources_it_was_collected_from():
    """bench.py attaches roofline.traffic only when the newest profiles/*_pmc_hbm.json (by round key, so r10 follows r9) was
    collected from the kernel sources in the tree; after a kernel edit without a new counter pass it says "withheld" instead
    of quoting stale counters -- either outcome is legal on CPU, a mismatch between the two is not"""
    import glob
    import sys
    sys.path.insert(0, ROOT)
    import bench
    names = ['profiles/r9z_pmc_hbm.json', 'profiles/r10a_pmc_hbm.json', 'profiles/r04z_pmc_hbm.json', 'profiles/r04x_pmc_hbm.json']
    assert sorted(names, key=bench.profile_round_key)[-1] == 'profiles/r10a_pmc_hbm.json'
    newest = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_hbm.json')), key=bench.profile_round_key)[-1]
    per_launch, src = bench.load_pmc(131072)
    if json.load(open(newest))['source_sha16'] == bench.source_sha16():
        assert per_launch and src == 'profiles/' + os.path.basename(newest)
    else:
        assert per_launch == {} and 'traffic withheld' in src and os.path.basename(newest) in src
    assert bench.load_pmc(8192) == ({}, None)          # counters exist for the full-size minibatch only


def test_committed_bench_lines_carry_a_clean_breakdown_check():
    """every committed bench line of the newest round: the per-kernel breakdown fits inside its step, for the headline and
    for each other configuration (failures are recorded in `breakdown_check`, not raised, so they have to be looked at here)"""
    import glob
    import sys
    sys.path.insert(0, ROOT)
    import bench
    lines = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_bench_atari4096.json')), key=bench.profile_round_key)
    d = json.load(open(lines[-1]))
    rows = [d] + [o for o in d.get('other_configs', []) if 'error' not in o]
    checked = [r['breakdown_check'] for r in rows if 'breakdown_check' in r]
    if not checked:
        pytest.skip('%s predates the recorded breakdown check' % os.path.basename(lines[-1]))
    assert all(c.startswith('ok') for c in checked), checked


def test_lds_bank_conflict_model_reproduces_the_measured_ratios():
    """scripts/lds_conflicts.py (lane groups / bank moduli of MI355X_MICROARCH.md): the ratios SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    measured before the layouts were changed (profiles/README.md, round 4), and zero for the tiled engines' permuted staging order"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import lds_conflicts as L

    def ratio(t):
        c = sum(r[2] for r in t.rows)
        i = sum(r[3] for r in t.rows)
        return (c - i) / c
    assert abs(ratio(L.gemm_x6(False)) - 0.36) < 0.01 and abs(ratio(L.gemm_x6(False, 128, 128)) - 0.33) < 0.01
    assert ratio(L.gemm_x6(True)) == 0 and ratio(L.gemm_x6(True, 128, 128)) == 0
    assert abs(ratio(L.c1wgrad_half()) - 0.36) < 0.02


def test_every_row_prices_its_sites_with_algorithmic_bytes_and_the_pipe_it_runs_on():
    """the recurrent row's conv / fc1 sites are the SAME engines on the same pipe as the feed-forward row's
    (its c2.dgrad must come out at the same fraction for the same time, not 2.7x higher on the fp32 peak), its LSTM GEMMs are
    fp32-MFMA sites with real bytes, the MLP row's fused step has algorithmic bytes, and every MFMA-priced site also carries
    the SURVEY 8(d) contract figure `frac_vs_fp32_mfma`.  bench.dominant_roofline on synthetic per-kernel times, no GPU."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    sites = {k: 6 for k in ('c2.fwd', 'c3.fwd', 'fc1.fwd', 'fc1.dgrad', 'c2.dgrad', 'c3.dgrad', 'c2.wgrad', 'c3.wgrad', 'fc1.wgrad')}
    sites.update({'c1.fwd': 3, 'c1.wgrad': 3})
    B = 8192
    flop = {'c2.dgrad': 2.0 * B * 81 * 64 * 512, 'c1.fwd': 2.0 * B * 400 * 32 * 256, 'lstm.fwd': 2.0 * B * 512 * 512}

    def prof(extra=()):
        p = {k: {'ms': ms * 16, 'count': 16, 'flops': flop[k] * 16, 'bytes': 0.0}
             for k, ms in (('c2.dgrad', 0.31), ('c1.fwd', 0.18))}
        for k, ms in extra:
            p[k] = {'ms': ms * 16, 'count': 16, 'flops': flop.get(k, 0.0) * 16, 'bytes': 0.0}
        return p

    rows = {}
    for wl, extra in (('atari', ()), ('atari_lstm', (('lstm.fwd', 0.065),))):
        res = dict(prof=prof(extra), nbatch_train=B, prof_steps=1, P=1687719)
        roof, per = bench.dominant_roofline(res, sites, wl)
        rows[wl] = (roof, per)
        for k, r in per.items():
            assert r['alg_bytes'] > 0, (wl, k)
            if r.get('pipe'):
                assert abs(r['frac_vs_fp32_mfma'] - r['mfma_tflops'] / 157.3) < 1e-12
    a, l = rows['atari'][0], rows['atari_lstm'][0]
    assert a['kernel'] == l['kernel'] == 'c2.dgrad' and a['pipe'] == l['pipe'] and '6 exact products' in l['pipe']
    assert abs(a['frac'] - l['frac']) < 1e-12 and a['alg_bytes'] == l['alg_bytes']
    assert rows['atari_lstm'][1]['lstm.fwd']['pipe'] == 'fp32 MFMA'
    assert rows['atari_lstm'][1]['c1.fwd']['bound'] == 'hbm'
    res = dict(prof={'mlp_step': {'ms': 0.041 * 320, 'count': 320, 'flops': 248.6e3 * 4096 * 320, 'bytes': 0.0}},
               nbatch_train=4096, prof_steps=1, P=57763)
    roof, _ = bench.dominant_roofline(res, sites, 'mujoco')
    assert roof['kernel'] == 'mlp_step' and roof['alg_bytes'] == 4096 * (1504 + 8 + 16 + 68) + 8 * 57763 and roof['pipe'] == 'fp32 MFMA'


def test_committed_headline_counters_belong_to_the_committed_kernel_sources():
    """Release-time guard: the newest committed bench line quotes counter traffic from the newest
    profiles/*_pmc_hbm.json, and that file was collected from the kernel sources in the tree.  A kernel edit without a new counter
    pass makes bench.py WITHHOLD the traffic (legal, see above) -- but then the committed headline line is stale evidence: this test
    warns about it (and fails under MRL_RELEASE_CHECK=1, which the round-end evidence script sets)."""
    import glob
    import sys
    import warnings
    sys.path.insert(0, ROOT)
    import bench
    lines = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_bench_atari4096.json')), key=bench.profile_round_key)
    pmcs = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_hbm.json')), key=bench.profile_round_key)
    d = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
    problems = []
    src = (d.get('roofline') or {}).get('traffic_source')
    if src != 'profiles/' + os.path.basename(pmcs[-1]):
        problems.append('%s quotes counters from %r, the newest counter file is %s' % (os.path.basename(lines[-1]), src, os.path.basename(pmcs[-1])))
    if json.load(open(pmcs[-1]))['source_sha16'] != bench.source_sha16():
        problems.append('%s was collected from other kernel sources than the tree holds (%s)' % (os.path.basename(pmcs[-1]), bench.source_sha16()))
    if problems:
        if os.environ.get('MRL_RELEASE_CHECK') == '1':
            pytest.fail('; '.join(problems))
        warnings.warn('stale headline evidence: ' + '; '.join(problems))
