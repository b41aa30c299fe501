"""Where a controller step spends its time (HIP events around the stages of ControlInterface.step).  usage: [num_envs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet
from rgbmanip_amd.config import ADAPOSE_CFGS, rl_cfg
from rgbmanip_amd.control_interface import ControlInterface
from rgbmanip_amd.estimator import AdaPoseEstimator_v5
from rgbmanip_amd.synthetic_env import SyntheticManipulation, SyntheticMultiVecEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype="bf16", device=0)
est = AdaPoseEstimator_v5(None, dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device"), None, dtype="bf16", net=net)
venv = SyntheticMultiVecEnv(N, dev, seed=0)
ci = ControlInterface(venv, est, SyntheticManipulation(venv), rl_cfg(), device=dev)
acts = [torch.from_numpy(synth.control_actions(N, s, 1) * 0.3).to(dev) for s in range(8)]
times = {}


def timed(name, fn):
    def wrap(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        times.setdefault(name, []).append((e0, e1))
        return r
    return wrap


import rgbmanip_amd.estimator as E
venv.get_image = timed("env.get_image (render)", venv.get_image)
venv.cam_move_to = timed("env.cam_move_to", venv.cam_move_to)
ci.add_view = timed("add_view (queue copy + mask extent)", ci.add_view)
ci.select_views = timed("select_views", ci.select_views)
ci._gather = timed("gather of selected views", ci._gather)
E.prepare_inputs = timed("prepare_inputs (2 per step)", E.prepare_inputs)
class _Net:
    device = net.device
    __call__ = staticmethod(timed("AdaPose forward", net.forward))


est.estimator = _Net()
E.postprocess = timed("postprocess", E.postprocess)
ci.get_reward = timed("get_reward", ci.get_reward)
ci.get_observation = timed("get_observation", ci.get_observation)
ci.action_to_pose = timed("action_to_pose", ci.action_to_pose)
ci.reset = timed("reset (every 4th step)", ci.reset)
for a in acts[:4]:
    ci.step(a)
torch.cuda.synchronize()
times.clear()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for a in acts:
    ci.step(a)
t1.record()
torch.cuda.synchronize()
total = t0.elapsed_time(t1) / len(acts)
print(f"N={N}  step {total:.2f} ms  -> {N / total * 1e3:.0f} env-steps/s")
for k, v in times.items():
    ms = sum(a.elapsed_time(b) for a, b in v) / len(acts)
    print(f"  {k:42s} {ms:8.3f} ms/step  ({len(v) / len(acts):.2f} calls/step)")
