import torch, time
dev="cuda"
x = torch.empty(1<<28, dtype=torch.float32, device=dev); y = torch.empty_like(x)   # 1 GiB each
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
print(f"1 GiB copy: {dt*1e3:.3f} ms -> {2*x.numel()*4/dt/1e12:.2f} TB/s (r+w)")
# tiny kernels in a graph: launch floor
a = torch.zeros(256, device=dev)
def chain(n):
    for _ in range(n): a.add_(1.0)
chain(10); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s): chain(10)
torch.cuda.synchronize()
with torch.cuda.graph(g): chain(1000)
g.replay(); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize()
print(f"tiny elementwise kernel in a graph: {(time.perf_counter()-t0)/5/1000*1e6:.2f} us per launch")
# medium: 2 MiB read streaming kernel (sum) repeated on different buffers
bufs = [torch.randn(1<<19, device=dev) for _ in range(64)]
out = torch.zeros(64, device=dev)
def sums():
    for i,b in enumerate(bufs): torch.sum(b, dim=0, out=out[i])
sums(); torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s): sums()
torch.cuda.synchronize()
with torch.cuda.graph(g2):
    for _ in range(4): sums()
g2.replay(); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): g2.replay()
torch.cuda.synchronize()
print(f"torch.sum over 2 MiB in a graph: {(time.perf_counter()-t0)/5/256*1e6:.2f} us per launch")
