"""Does a REPLAYED hipGraph keep a memset node ordered between the kernel nodes around it?

One stream, a linear chain captured K times over:   buf.fill_(garbage)  ->  hipMemsetAsync(buf, 0)  ->  buf += 1  ->  out[k] += buf
(three kernels and one memset node per round).  In stream order every out[k] ends at exactly 1.  The graph is replayed R times and
every replay checked.  python tools/probes/graph_memset_probe.py [floats per buffer] [rounds] [replays] [memset|memcpy]

Mode ``memcpy``: the same chain with a device-to-device hipMemcpyAsync (a memcpy node) in the memset's place:
  src.fill_(k + 1) -> hipMemcpyAsync(buf <- src) -> buf += 1 -> out[k] += buf      (every out[k] ends at k + 2).

Result on ROCm 7.2.0 / MI355X (round 5, gpurun_out/h2/probe_*.txt): eager launches correct; captured, the FIRST replay is correct and
from the second replay on 5-8 of 64 memset rounds end with garbage (values like 4.27e31, or 2 = the fill and the memset both
missing): a hipMemsetAsync captured into a graph must not be relied on."""
import ctypes, sys
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
R = int(sys.argv[3]) if len(sys.argv) > 3 else 12
MODE = sys.argv[4] if len(sys.argv) > 4 else "memset"
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpyAsync.restype = ctypes.c_int
dev = torch.device("cuda:0")
buf = torch.empty(n, device=dev)
big = torch.empty(1 << 22, device=dev)          # something for the neighbouring kernels to chew on (keeps the queue busy)
one = torch.ones(n, device=dev)
src = torch.empty(n, device=dev)
out = torch.zeros(K, n, device=dev)


def chain():
    for k in range(K):
        buf.fill_(1000.0 + k)
        big.mul_(1.0001)
        if MODE == "memcpy":
            src.fill_(float(k + 1))
            rc = hip.hipMemcpyAsync(buf.data_ptr(), src.data_ptr(), n * 4, 3, torch.cuda.current_stream().cuda_stream)   # 3 = device to device
        else:
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, n * 4, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        buf.add_(one)
        out[k].add_(buf)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    big.fill_(1.0); out.zero_(); chain()
torch.cuda.synchronize()
want = (torch.arange(K, device=dev, dtype=torch.float32) + 2.0 if MODE == "memcpy" else torch.ones(K, device=dev))[:, None]
print("eager: wrong rounds", int((out != want).any(dim=1).sum()), "of", K)
g = torch.cuda.CUDAGraph()
out.zero_(); torch.cuda.synchronize()
with torch.cuda.graph(g):
    chain()
bad_total = 0
for r in range(R):
    out.zero_(); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    bad = int((out != want).any(dim=1).sum())
    bad_total += bad
    vals = sorted(set(out[(out != want).any(dim=1)].flatten().tolist()))[:6] if bad else []
    print(f"replay {r}: wrong rounds {bad} of {K} {vals}")
print("RESULT", MODE, "nodes WRONG in replayed graphs" if bad_total else "nodes correct", bad_total)
