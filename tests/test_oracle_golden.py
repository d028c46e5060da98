"""CPU: the oracle restatement vs. the golden vectors produced by the reference's own code
(oracle/make_golden.py).  This is what pins the oracle."""
import glob
import os

import numpy as np
import pytest

from oracle import ppo2_numpy as O


def _cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, 'runner_*.npz')))


def test_have_golden(golden_dir):
    assert len(_cases(golden_dir)) >= 8


@pytest.mark.parametrize('name', ['t128_n8', 't5_n3', 't37_n70', 't1_n4', 't64_n129_alldone', 't64_n65_nodone',
                                  't300_n33_g1', 't16_n64_lam0'])
def test_gae_sf01_bit_exact(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'runner_%s.npz' % name))
    ret, adv = O.gae(g['in_rewards'], g['in_values'], g['in_dones'], g['in_last_values'], g['in_last_dones'],
                     float(g['gamma']), float(g['lam']))
    assert ret.dtype == np.float32
    np.testing.assert_array_equal(O.sf01(ret), g['out_returns'])          # bit-exact
    np.testing.assert_array_equal(O.sf01(g['in_obs']), g['out_obs'])
    np.testing.assert_array_equal(O.sf01(g['in_dones']), g['out_masks'])
    np.testing.assert_array_equal(O.sf01(g['in_actions']), g['out_actions'])
    np.testing.assert_array_equal(O.sf01(g['in_values']), g['out_values'])
    np.testing.assert_array_equal(O.sf01(g['in_neglogpacs']), g['out_neglogpacs'])
    assert g['out_masks'].dtype == np.bool_


def test_shuffle_stream(golden_dir):
    g = np.load(os.path.join(golden_dir, 'shuffle.npz'))
    for key in g.files:
        seed, nbatch, nmb, nep = [int(x[1:]) for x in key.split('_')]
        np.random.seed(seed)
        rows = np.stack(list(O.minibatch_indices(nbatch, nbatch // nmb, nep)))
        np.testing.assert_array_equal(rows, g[key])


def test_explained_variance(golden_dir):
    g = np.load(os.path.join(golden_dir, 'misc.npz'))
    assert O.explained_variance(g['ev_ypred'], g['ev_y']) == float(g['ev'])
    assert np.isnan(O.explained_variance(np.zeros(4, np.float32), np.ones(4, np.float32)))
