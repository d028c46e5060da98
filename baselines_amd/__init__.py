"""baselines_amd -- MI355X-native PPO2 actor-learner behind openai/baselines' ppo2 API.

Layout: csrc/ (hand-written HIP for gfx950 + the C ABI of include/mrl.h), _lib / ops (ctypes
binding), ppo2/ (learn, Runner, Model), common/ (VecEnv contract, policies, registry, spaces),
deepq/ (prioritized-replay slice)."""
__version__ = '0.1.0'
