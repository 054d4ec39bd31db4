"""What does THIS runtime do with hipMemsetAsync nodes in a captured stream?  (Round 1 had two observations with such
nodes in the step's hipGraph: a host segfault inside torch's capture_end, gpurun_out/det.log, and — when the capture
went through — 256-byte and 4-byte memsets that wrote garbage from the second replay on.)  Run in a CHILD process by
`python tools/capture_memset_probe.py`; the parent only reports how the child ended.  One run; never loop it."""
import ctypes as C
import subprocess
import sys


def child():
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    hip.hipMemsetAsync.restype = C.c_int
    flag = torch.full((64,), 7, device="cuda", dtype=torch.int32)       # 256-byte block, as the pack flag was
    word = torch.full((1,), 9, device="cuda", dtype=torch.int32)        # 4-byte word, as the batch builder's error count
    odd = torch.full((1003,), 5, device="cuda", dtype=torch.uint8)      # unaligned tail, as the softmax ticket block
    acc = torch.zeros(4, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())

    def body(st):
        assert hip.hipMemsetAsync(flag.data_ptr(), 0, 256, st) == 0
        assert hip.hipMemsetAsync(word.data_ptr(), 0, 4, st) == 0
        assert hip.hipMemsetAsync(odd.data_ptr() + 1, 0, 1001, st) == 0
        acc.add_(flag[:4].float() + word.float())                       # a kernel that consumes them

    with torch.cuda.stream(side):
        body(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(torch.cuda.current_stream().cuda_stream)
    print("capture_end returned", flush=True)
    for rep in range(3):
        flag.fill_(7); word.fill_(9); odd.fill_(5); acc.zero_()
        g.replay()
        torch.cuda.synchronize()
        print(f"replay {rep}: flag {flag[:2].tolist()} word {word.tolist()} odd[0:3] {odd[:3].tolist()} "
              f"acc {acc.tolist()}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, timeout=300)
        print(r.stdout)
        print("child exit code", r.returncode, "(negative = killed by that signal; -11 is SIGSEGV)")
        print(r.stderr[-1500:])
