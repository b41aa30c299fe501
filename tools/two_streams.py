"""Does running the batch as two half-batches on two HIP streams (kernel tails of one filled by the other) beat one stream?
usage: two_streams.py [dtype]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet, postprocess
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = 256
inp = synth.adapose_inputs(16, seed=0)
full = {k: torch.from_numpy(np.concatenate([v] * (B // 16), 0)).cuda() for k, v in inp.items()}
halves = [{k: v[i * B // 2:(i + 1) * B // 2].contiguous() for k, v in full.items()} for i in range(2)]
sd = synth.adapose_state_dict(seed=0)
one = AdaPoseNet(sd, dtype=dt)
two = [AdaPoseNet(sd, dtype=dt), AdaPoseNet(sd, dtype=dt)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(net, d, stream=None):
    o = net(d["img1"], d["choose1"], d["img2"], d["choose2"], d["P1"], d["P2"], d["depths"], stream=stream)
    return postprocess(o["view1_nocs"], o["view1_depth"], o["view1_r"], d["choose1"], d["K1"], d["E1"], stream=stream)
def step_one():
    run(one, full)
def step_two():
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            run(two[i], halves[i], streams[i])
def step_seq():
    for i in range(2):
        run(two[i], halves[i])
def step_two_full():            # two whole batches in flight: consecutive steps of a serving loop on alternating streams
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            run(two[i], full, streams[i])
def step_one_twice():
    run(one, full); run(one, full)
for name, f, n in (("one stream, 2 x B=256", step_one_twice, 512), ("two streams, B=256 each", step_two_full, 512), ("one stream, 2 x B=256", step_one_twice, 512), ("two streams, B=256 each", step_two_full, 512)):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(4): f()
    torch.cuda.synchronize()
    print(f"{dt} {name}: {(time.perf_counter() - t) / 4 * 1e3 * 256 / n:.2f} ms per 256 poses")
for name, f in (("one stream B=256", step_one), ("two streams B=128 each", step_two), ("one stream, two B=128 calls", step_seq), ("one stream B=256", step_one), ("two streams B=128 each", step_two)):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    print(f"{dt} {name}: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms per 256 poses")
