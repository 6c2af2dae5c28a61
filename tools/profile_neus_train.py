import sys, torch, time
sys.path.insert(0,'.')
import bench
import go_slam_amd.neus as neus
from go_slam_amd.neus.mapper import MapTrainer
from torch.profiler import profile, ProfilerActivity
device=torch.device('cuda:0')
g = torch.Generator().manual_seed(43)
model = neus.InstantNeuS({}, [[-5.0, 5.0]] * 3).to(device)
with torch.no_grad():
    model.sdf_network.encoding.encoding.params.copy_((torch.rand(model.sdf_network.encoding.encoding.params.shape, generator=g) - 0.5) * 0.02)
    model.sdf_network.sdf_layer.weight[:, 3:] = torch.randn(32, 32, generator=g).to(device) * 0.1
R = neus.Renderer(N_samples=24, N_surface=48)
n=int(sys.argv[1]) if len(sys.argv) > 1 else 32768
o = (torch.rand(n, 3, generator=g) * 6 - 3).to(device)
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).to(device)
gt = torch.rand(n, generator=g) * 3.5 + 0.5; gt[torch.rand(n, generator=g) < 0.1] = 0; gt=gt.to(device)
col = torch.rand(n, 3, generator=g).to(device); pr = torch.rand(24, generator=g).to(device)
tr = MapTrainer(model, R)
for _ in range(3): tr.step(o,d,col,gt,pr)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): tr.step(o,d,col,gt,pr)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=70))
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print("device kernels per step:", len(ev) / 3.0, " device time per step (us):", sum(e.device_time for e in ev) / 3.0 if hasattr(ev[0], "device_time") else sum(e.cuda_time for e in ev) / 3.0)
t0 = time.perf_counter()
for _ in range(20): tr.step(o,d,col,gt,pr)
torch.cuda.synchronize()
print("wall ms per step:", (time.perf_counter() - t0) / 20 * 1e3)
