"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the openai/baselines PPO2 hot path (and the DQN prioritized
replay slice) used as the parity checker for the HIP kernels in
``baselines_amd``.  Nothing in the product package may import this; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.

Pinning status (see DESIGN.md "Oracle"):
  * GAE / sf01 / rollout dtypes   -- pinned: bit-exact against the reference's own
    ``baselines.ppo2.runner.Runner.run`` executed in the build container
    (fixtures under tests/golden/, generator oracle/make_golden.py).
  * segment trees / prioritized replay -- pinned against the reference's code
    run verbatim + its known-answer tests (test_segment_tree.py).
  * TF-graph parts (loss, clip_by_global_norm, Adam, conv/fc forward) --
    "parity unpinned" at the TensorFlow boundary: tensorflow<2 is a third-party
    dependency absent from /root/reference and from this image.  The
    restatement follows model.py:57-114 / distributions.py / a2c/utils.py line
    by line, is cross-checked in fp64 against autograd and finite differences,
    and the Adam formula is cross-pinned by the reference's NumPy mirror
    (common/mpi_adam.py:38-41).
"""
