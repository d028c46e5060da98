import random

import numpy as np


def set_global_seeds(i):
    """Seeds NumPy / `random` (and torch's generators, our stand-in for tf.set_random_seed).
    Like the reference (common/misc_util.py:48-62, whose `import MPI` can never succeed) the seed
    is NOT offset by rank: every rank draws the same ortho-init stream and the same minibatch
    permutations."""
    np.random.seed(i)
    random.seed(i)
    if i is not None:
        try:
            import torch
            torch.manual_seed(int(i))
        except ImportError:
            pass
