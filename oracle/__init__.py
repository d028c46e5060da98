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
    by line.  What pins it short of TensorFlow itself (tests/test_oracle_pins.py):
    the Adam formula against the reference's NumPy mirror (common/mpi_adam.py:38-41)
    on the problem of its own test_MpiAdam; validate_probtype
    (distributions.py:320-348) on the neglogp / entropy formulas; the NatureCNN
    forward + heads against the defining sums of tf.nn.conv2d / conv_to_fc / fc
    in float64 NumPy (no library convolution: layout conventions HWIO, NHWC
    flatten, bias broadcast); the loss statistics from the formulas of
    model.py:57-91; the gradient as a directional derivative of that loss;
    clip_by_global_norm; the LSTM cell against a closed-form NumPy backward
    (tests/test_oracle_golden.py); the layer-normalised LSTM cell (a2c/utils.py:104-140),
    mlp(layer_norm=True) (common/models.py:97-98) and the layer-normalised dueling
    Q heads (deepq/models.py:24-44) against float64 NumPy written from those lines.
"""
