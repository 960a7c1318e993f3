"""A/B: the whole main-stream side of a cfg-2 train_step (rollout, estimator, BPTT, value / dynamics) on a stream whose CU mask leaves `k` CUs to the
side stream and the belief chains alone (hipExtStreamCreateWithCUMask; the process's default stream stays idle, so there are still four busy queues).
usage: python3 tools/perf/cu_mask_main.py [k ...]"""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(leave_out):
    n = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n + 31) // 32
    bits = [1] * n
    for i in range(leave_out): bits[n - 1 - i] = 0          # the highest logical CUs (the mask is dealt round-robin over the XCDs)
    arr = (ctypes.c_uint32 * words)(*[sum(bits[32 * w + b] << b for b in range(32) if 32 * w + b < n) for w in range(words)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
agent = bench.make_agent(torch.device("cuda"))
for _ in range(3): agent.train_step()
def run(st, n=10):
    with torch.cuda.stream(st):
        for _ in range(3): agent.train_step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): agent.train_step()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
ks = [int(a) for a in sys.argv[1:]] or [0, 16, 32, 64]
streams = {k: masked_stream(k) for k in ks}
print("default stream:", f"{run(torch.cuda.current_stream()):.2f} ms/step", flush=True)
for rep in range(2):
    for k in ks:
        print(f"main on a stream without {k:3d} CUs: {run(streams[k]):.2f} ms/step", flush=True)
print("default stream:", f"{run(torch.cuda.current_stream()):.2f} ms/step", flush=True)
