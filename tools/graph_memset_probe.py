"""A hipMemsetAsync captured into a hipGraph writes garbage from the second replay on (ROCm 7.2 / torch 2.10).
Poison the pack flag before each replay, let the captured dp_adj_pack + dp_adj_aggregate_packed sequence clear it, read
it back.  With hipMemsetAsync in dp_adj_pack this printed  replay 1 flag [1665138688, 29562]  and a result that differed
from eager (fp32 fallback taken); the library now zeroes with kernels only (dp_rowops.hip zero_fill) and every replay
reads [0, 0] and equals the eager result.  Run on a GPU box:  PYTHONPATH=. python tools/graph_memset_probe.py"""
import torch, time
from graph_pooling_amd import _lib
lib = _lib.load()
B, N, C = 20, 500, 40
torch.manual_seed(0)
A = (torch.rand(B, N, N, device="cuda") < 0.02).float()
V = torch.randn(B, N, C, device="cuda")
U = torch.empty(B, N, C, device="cuda")
nb = lib.dp_adj_pack_bytes(B, N)
pk = torch.empty(nb, device="cuda", dtype=torch.uint8); pkt = torch.empty(nb, device="cuda", dtype=torch.uint8)
flag = torch.full((64,), 5, device="cuda", dtype=torch.int32)
wsb = lib.dp_adj_aggregate_packed_workspace_bytes(B, N, C)
ws = torch.empty(wsb, device="cuda", dtype=torch.uint8)
def seq():
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.dp_adj_pack(A.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(), B, N, st))
    _lib.check(lib.dp_adj_aggregate_packed(A.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(), V.data_ptr(), C,
                                           U.data_ptr(), C, B, N, C, 0, 0.0, 0, ws.data_ptr(), wsb, st))
seq(); torch.cuda.synchronize(); U0 = U.clone(); print("eager flag", flag[:2].tolist())
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    seq()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    seq()
for i in range(3):
    flag.fill_(7)          # poison: the captured memset must clear it
    g.replay(); torch.cuda.synchronize()
    print("replay", i, "flag", flag[:2].tolist(), "equal eager:", bool(torch.equal(U, U0)), float((U-U0).abs().max()))
