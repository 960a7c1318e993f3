"""cfg-2 train_step over the real RCCL backend with ONE rank and every data-parallel branch forced on: does RCCL's internal stream cost the agent one of
the process's four hardware queues (aid_amd/streams.py)?  Times the step with the belief updates as concurrent chains and merged."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
import torch, torch.distributed as dist
import bench
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from aid_amd import parallel
parallel.is_dp = lambda: True
if os.environ.get("AID_OWN_MAIN"):          # run everything on a non-default stream: the legacy default stream synchronises implicitly with blocking streams
    _m = torch.cuda.Stream(); torch.cuda.set_stream(_m)
agent = bench.make_agent(torch.device("cuda"))
agent.dp_autotune_schedule = False      # this tool sets the schedule by hand
agent.enable_data_parallel()
for mode, la in ((True, True), (True, False), (False, False), (True, True), (True, False)):
    agent.concurrent_belief_loops = mode; agent.lookahead_elbo = la
    for _ in range(4): agent.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): agent.train_step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 15
    print(f"RCCL one rank, all DP collectives on: concurrent chains {mode}, look-ahead ELBO {la}: {dt*1e3:7.2f} ms per train_step", flush=True)
import collections
agent.concurrent_belief_loops = True; agent.lookahead_elbo = True
_orig = agent._lookahead_elbo
host = {"la": 0.0, "n": 0}
def timed(*a, **k):
    t = time.perf_counter(); r = _orig(*a, **k); host["la"] += time.perf_counter() - t; host["n"] += 1; return r
agent._lookahead_elbo = timed
t_all = time.perf_counter()
for _ in range(10): agent.train_step()
print(f"RCCL one rank host: look-ahead enqueue {1e3*host['la']/host['n']:.2f} ms per call; whole train_step call {1e3*(time.perf_counter()-t_all)/10:.2f} ms", flush=True)
agent._lookahead_elbo = _orig
n = 8; steps = []
for _ in range(n + 1):
    agent.timeline = []
    agent.train_step()
    steps.append(agent.timeline)
agent.timeline = None
torch.cuda.synchronize()
acc = collections.OrderedDict()
for i in range(n):
    t0, h0 = steps[i][0][1], steps[i][0][2]
    for label, ev, ht in steps[i][1:] + [("start of the next step",) + tuple(steps[i + 1][0][1:])]:
        a = acc.setdefault(label, [0.0, 0.0]); a[0] += t0.elapsed_time(ev) / n; a[1] += 1e3 * (ht - h0) / n
for label, (ms, hms) in acc.items():
    print(f"RCCL one rank {ms:8.2f} ms on the GPU   (enqueued by the host at {hms:7.2f} ms)  {label}")
dist.destroy_process_group()
