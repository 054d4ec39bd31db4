"""Which fp32-MFMA GEMM launches make up the ER step, by shape.  Two halves:
  (run, on the GPU box)   cd /tmp && DP_GEMM_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d OUT -o gs -- \
                              python3 $R/tools/er_gemm_shapes.py run 2> OUT/shapes.err
  (join, anywhere)        python3 tools/er_gemm_shapes.py join OUT/shapes.err OUT/.../gs_kernel_trace.csv
`run` does three eager ER steps and writes a marker line to stderr before the last; the library's DP_GEMM_TRACE log
has one line per grouped fp32 launch, in launch order, and the kernel trace has their durations in the same order."""
import csv
import sys


def run():
    import torch
    sys.path.insert(0, ".")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    w = bench.WORKLOADS["er"]
    dev = torch.device("cuda:0")
    model, batch, _ = bench.make_model_and_batch(w, False, dev)
    for it in range(3):
        if it == 2:
            torch.cuda.synchronize()
            sys.stderr.write("== traced step\n")
            sys.stderr.flush()
        for p in model.parameters():
            p.grad = None
        y = model(batch["x"], batch["adj"], batch["nn"], assign_x=batch["x"])
        loss = model.loss(y, batch["label"])
        loss.backward()
    torch.cuda.synchronize()


def join(err_path, trace_path):
    lines = open(err_path, errors="replace").read().splitlines()
    at = max(i for i, l in enumerate(lines) if l.startswith("== traced step"))
    shapes = [l for l in lines[at + 1:] if l.startswith("bgemm batch=")]
    rows = [r for r in csv.DictReader(open(trace_path)) if "bgemm_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) - len(shapes):]
    agg = {}
    for s, r in zip(shapes, rows):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tile = r["Kernel_Name"].split("bgemm_kernel")[1].split("(")[0]
        key = (s.split(":", 1)[1].strip(), tile, s.split(":")[0])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += us
    tot = sum(v[1] for v in agg.values())
    print(f"{len(shapes)} grouped fp32 GEMM launches in the step, {tot:.0f} us")
    for (shape, tile, head), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{us:8.1f} us  x{cnt:<2d} {us / cnt:7.1f} each  {tile:16s} {head:24s} {shape}")


def kernels(trace_path):
    """Per-kernel totals of the LAST third of the trace's dispatches (= the third of three identical eager steps)."""
    rows = list(csv.DictReader(open(trace_path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the three steps launch the same sequence: find the period from the end
    names = [r["Kernel_Name"] for r in rows]
    per = next(p for p in range(20, len(names) // 2) if names[-p:] == names[-2 * p:-p])
    last = rows[-per:]
    agg = {}
    for r in last:
        k = r["Kernel_Name"].replace("void dp::", "").replace("(anonymous namespace)::", "")[:64]
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in agg.values())
    span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
    print(f"{per} launches in the step; kernel time {tot:.0f} us, first start to last end {span:.0f} us")
    for k, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{us:8.1f} us  x{cnt:<3d} {us / cnt:7.1f} each  {k}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "kernels":
        kernels(sys.argv[2])
    else:
        join(sys.argv[2], sys.argv[3])
