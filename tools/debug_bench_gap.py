import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
video, op, graph, _ = bench.build_state(dev)
def sync(): torch.cuda.synchronize()
def run(n, label):
    times = []
    for k in range(n):
        sync(); t = time.perf_counter(); bench.keyframe_step(graph); sync(); times.append((time.perf_counter() - t) * 1e3)
    print(label, [round(x, 1) for x in times])
run(3, "warm")
run(40, "gc on ")
gc.collect(); gc.freeze(); gc.disable()
run(40, "gc off")
gc.enable()
stats = torch.cuda.memory_stats()
print("alloc retries", stats.get("num_alloc_retries"), "segments", stats.get("segment.all.current"), "reserved GB", stats.get("reserved_bytes.all.current") / 2**30,
      "cudaMalloc calls", stats.get("segment.all.allocated"))
