"""CPU: the one-line JSON contract of bench.py, checked on the committed result of the last GPU run
(profiles/r02s_bench_atari4096.json) and on bench.py's own argument defaults."""
import ast
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r02s_bench_atari4096.json')))
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
    assert (r['traffic'] is None or r['traffic_source'].startswith('profiles/r02')) and d['dtype'] == 'f32'
    assert d['config']['arithmetic_mode'] == 'bf16x8-split'          # the headline runs the 8-product (fp32-or-better) mode
    kr = d['kernel_rooflines']
    assert kr['c2.wgrad']['peak'] == 157.3 and abs(kr['c2.fwd']['peak'] - 2516.6 / 8) < 1e-3 and abs(kr['c1.fwd']['peak'] - 2516.6 / 3) < 1e-3
    assert {o['workload'].split()[0] for o in d['other_configs']} == {'ppo2', 'deepq'} and len(d['other_configs']) == 3
    c = d['cpu_baseline']
    assert c['host_cpu_count'] >= c['cores'] and c['host_cpu_model']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
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
