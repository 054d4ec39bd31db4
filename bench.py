#!/usr/bin/env python3
"""bench.py — graphs/sec, forward + loss + backward, of the MI355X DiffPool path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dd|enzymes|er] [--linkpred] [--no-graph]

Metric (BASELINE.json): graphs/sec fwd+bwd on a DD-shaped padded batch (B=20 per GPU, N_max=500, F=89,
H=E=20, K=50, 3 GCN layers, 1 pooling level, C=2; synthetic seeded data, reference-style init).
A "step" = SoftPoolingGcnEncoder.forward + .loss + loss.backward() on one resident batch (+ the flat
gradient all-reduce over RCCL when N > 1).  No optimizer, no H2D.

For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU; the batch is
sharded by graph (weak scaling: 20 graphs per GPU), and the only collective is one all-reduce of the
flat gradient buffer per step.

Rank 0 prints ONE JSON line with the driver's fields plus
  "roofline":     the step's dominant kernel: for the DD-shaped workload the persistent level-0 backward kernel
                  (k_level0_bwd; its forward twin under "roofline_fwd", the stand-alone aggregation kernel the other
                  plans use under "roofline_aggregate"), for ER the MFMA-bound pooling product.  Algorithmic bytes
                  (flops) per launch over the kernel's average duration — `us_per_launch_events` measured in this run
                  with HIP events on the launch stream, `rocprof_avg_ns` the tracked rocprofv3 figure from profiles/
                  (frac is priced on the latter when present); `traffic` = PMC bytes from profiles/, not this run
  "cpu_baseline": the CPU oracle (oracle/diffpool_oracle.py, a torch-CPU restatement pinned to the
                  reference by tests/golden) timed on this host on the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: B per GPU, N, F, H, C, ratio, p(edge), n_min, onehot
    "dd": dict(B=20, N=500, F=89, H=20, C=2, ratio=0.1, p=0.02, n_min=30, onehot=True,
               name="S-DD: DD-shaped padded batch, B=20/GPU, N_max=500, F=89, H=E=20, K=50, L=3, 1 pool"),
    "enzymes": dict(B=20, N=100, F=3, H=20, C=6, ratio=0.1, p=0.10, n_min=10, onehot=True,
                    name="S-ENZ: ENZYMES-shaped padded batch, B=20/GPU, N=100, F=3, H=E=20, K=10"),
    # BASELINE configs[4] ("ENZYMES with Set2Set global readout + 3 pooling levels") is two models in the reference
    # (SURVEY Appendix E.4): GcnSet2SetEncoder without pooling, and SoftPoolingGcnEncoder with max readout
    "enzymes_s2s": dict(B=20, N=100, F=3, H=20, C=6, ratio=0.1, p=0.10, n_min=10, onehot=True, model="set2set",
                        name="S-S2S: ENZYMES-shaped batch through GcnSet2SetEncoder (Set2Set readout, no pooling)"),
    "enzymes_p3": dict(B=20, N=100, F=3, H=20, C=6, ratio=0.25, p=0.10, n_min=10, onehot=True, num_pooling=3,
                       name="S-ENZ-P3: ENZYMES-shaped batch, SoftPoolingGcnEncoder with 3 pooling levels K=25,6,1 "
                            "(build-defined semantics, SURVEY Appendix B)"),
    "er": dict(B=256, N=1024, F=64, H=20, C=2, ratio=0.25, p=0.01, n_min=1024, onehot=False, num_pooling=2, roofline="mfma",
               name="S-ER: Erdos-Renyi B=256, N=1024, F=64, H=E=20, 2 pooling levels K=256->64 (SURVEY 8d)"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PFLOP/s


def synthetic_batch(w, seed):
    """Seeded synthetic padded batch in the layout graph_sampler.py:97-109 delivers: symmetric 0/1
    zero-diagonal Erdos-Renyi adjacency in the top-left n_b x n_b block, one-hot (or N(0,1)) node features with zero
    rows for n >= n_b, n_b ~ U{n_min..N}, uniform graph labels."""
    g = torch.Generator().manual_seed(seed)
    B, N, F_ = w["B"], w["N"], w["F"]
    sizes = torch.randint(w["n_min"], N + 1, (B,), generator=g)
    adj = torch.zeros(B, N, N)
    x = torch.zeros(B, N, F_)
    for b in range(B):
        n = int(sizes[b])
        upper = torch.triu((torch.rand(n, n, generator=g) < w["p"]).float(), diagonal=1)
        adj[b, :n, :n] = upper + upper.t()
        if w["onehot"]:
            x[b, torch.arange(n), torch.randint(0, F_, (n,), generator=g)] = 1.0
        else:
            x[b, :n] = torch.randn(n, F_, generator=g)
    label = torch.randint(0, w["C"], (B,), generator=g)
    return x, adj, sizes.to(torch.int32), label


def make_model_and_batch(w, linkpred, device, seed_offset=0):
    from graph_pooling_amd.encoders import GcnSet2SetEncoder, SoftPoolingGcnEncoder
    x, adj, nn_, label = synthetic_batch(w, seed=1 + seed_offset)
    torch.manual_seed(0)             # reference init (encoders.py:1225-1229: xavier GraphConv, default nn.Linear)
    if w.get("model") == "set2set":
        model = GcnSet2SetEncoder(w["F"], w["H"], w["H"], w["C"], 3)
    else:
        model = SoftPoolingGcnEncoder(w["N"], w["F"], w["H"], w["H"], w["C"], 3, w["H"], assign_ratio=w["ratio"],
                                      num_pooling=w.get("num_pooling", 1), linkpred=linkpred)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    batch = dict(x=x.to(device), adj=adj.to(device), nn=nn_.to(device), label=label.to(device))
    cpu = dict(x=x, adj=adj, nn=nn_.numpy(), label=label, params=params)
    return model, batch, cpu


def cpu_baseline(cpu, w, linkpred, budget_s=15.0):
    """Time the CPU oracle (fwd + loss + bwd) on the same batch, bounded to ~budget_s seconds of CPU work.
    torch's intra-op pool is tried at a few sizes (all host cpus is pathological on many-core hosts for
    these small matrices); the best is reported with the thread count actually used."""
    from oracle import diffpool_oracle as O
    ncpu = os.cpu_count() or 1
    P = {k: v.clone().requires_grad_(True) for k, v in cpu["params"].items()}

    def step():
        for v in P.values():
            v.grad = None
        if w.get("model") == "set2set":
            y = O.set2set_encoder_forward(P, cpu["x"], cpu["adj"], cpu["nn"], num_layers=3)
            loss = torch.nn.functional.cross_entropy(y, cpu["label"])
        else:
            y, inter = O.softpool_forward(P, cpu["x"], cpu["adj"], cpu["nn"], cpu["x"],
                                          num_pooling=w.get("num_pooling", 1))
            loss, _ = O.softpool_loss(y, cpu["label"], inter["assign_0"], cpu["adj"], cpu["nn"], linkpred)
        loss.backward()

    cands = sorted({t for t in (1, 8, 16, 32) if t <= ncpu})
    best = None
    per = budget_s / len(cands)
    for t in cands:
        torch.set_num_threads(t)
        step()
        t0 = time.perf_counter()
        step()
        one = time.perf_counter() - t0
        n = int(max(2, min(100, per / max(one, 1e-4))))
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        dt = time.perf_counter() - t0
        gps = w["B"] * n / dt
        if best is None or gps > best[0]:
            best = (gps, t, n, dt)
    gps, t, n, dt = best
    return dict(value=round(gps, 2), unit="graphs/s", cores=t, kind="port",
                sample=f"{n} fwd+loss+bwd steps of the same B={w['B']} batch, torch-CPU fp32 oracle, best of "
                       f"{cands} intra-op threads ({dt:.1f} s at {t} threads; host has {ncpu} cpus)")


def step_roofline(w, linkpred, ms_per_step, world=1):
    """Step-level roofline: roofline.achieved = max(t_HBM, t_MFMA) / t_measured, from SURVEY §8(d)'s
    algorithmic per-graph figures — FLOPs forward = sum over GCN stacks and layers of (2 n^2 f_l + 2 n f_l f_{l+1}) +
    assign heads 2 n Da K + pooling 2 K n^2 + 2 K^2 n + 2 K n D (+ link loss 2 N^2 K), fwd+bwd = 3 x fwd; bytes = the
    level-0 adjacency passes (2L+1 forward, 2(L-1)+1 backward, +1 each with the link loss) at the element size THIS
    build reads (bf16 copies written once from the fp32 input when N >= 128, fp32 otherwise) + 8 N (D + Da + K) x 3
    bytes of activations — against 8.0 TB/s HBM and 2.5 PFLOP/s dense bf16 MFMA."""
    N, F_, H, L = w["N"], w["F"], w["H"], 3
    P = w.get("num_pooling", 1) if w.get("model") != "set2set" else 0
    D = H * L
    packed = N >= 128

    def stack(n, fin, hid, fout):
        dims = [fin] + [hid] * (L - 1) + [fout]
        return sum(2.0 * n * n * dims[l] + 2.0 * n * dims[l] * dims[l + 1] for l in range(L))
    flops = stack(N, F_, H, H)
    n, fa = N, F_
    K0 = Da0 = 0
    for j in range(P):
        K = int(n * w["ratio"])
        Da = H * (L - 1) + K
        if j == 0:
            K0, Da0 = K, Da
        flops += stack(n, fa, H, K) + 2.0 * n * Da * K                      # assign stack + head
        flops += 2.0 * K * n * n + 2.0 * K * K * n + 2.0 * K * n * D        # pooling
        flops += stack(K, D, H, H)                                          # embed after pool
        n, fa = K, D
    if linkpred and P:
        flops += 2.0 * N * N * K0
    flops *= 3.0
    pf, pb = 2 * L + 1 + (1 if linkpred else 0), 2 * (L - 1) + 1 + (1 if linkpred else 0)
    if P == 0:
        pf, pb = L, L - 1
    s_a = 2 if packed else 4
    adj_bytes = (pf + pb) * N * N * s_a + (N * N * (4 + 2 * 2) if packed else 0)
    act_bytes = 8.0 * N * (D + Da0 + K0) * 3
    per_graph = adj_bytes + act_bytes
    B = w["B"] * world
    t_hbm = per_graph * B / (HBM_PEAK_GBS * 1e9) * 1e6 / world
    t_mfma = flops * B / (MFMA_BF16_PEAK_TF * 1e12) * 1e6 / world
    t_meas = ms_per_step * 1e3
    return dict(t_hbm_us=round(t_hbm, 2), t_mfma_us=round(t_mfma, 3), achieved=round(max(t_hbm, t_mfma) / t_meas, 4),
                bytes_per_graph=int(per_graph), flops_per_graph=int(flops), adjacency_bytes_per_element=s_a,
                adjacency_passes=pf + pb, note="per GPU; HBM 8.0 TB/s, bf16 MFMA 2.5 PFLOP/s dense")


def _profiled(kernel, shape):
    """What profiles/ holds for `kernel` at `shape`: {"rocprof_avg_ns", "traffic_bytes", "source", ...} written by
    tools/update_pmc_traffic.py from tools/refresh_profiles.sh's rocprofv3 passes (an EARLIER run of this build on a GPU
    box, not this run) — or {} when the kernel / shape was never profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get("kernels", {}).get(kernel, {}).get(shape, {}) or {}
    except Exception:              # noqa: BLE001
        return {}


def _roofline(bound, kernel, work, us_events, prof, peak, unit, **extra):
    """One roofline object.  `work` = algorithmic bytes (or flops) per launch.  Two durations are reported side by side:
    `us_per_launch_events` — HIP events on the launch stream, measured in THIS run — and `rocprof_avg_ns`, the tracked
    rocprofv3 --kernel-trace --stats AverageNs of the same kernel at the same shape from profiles/ (null when profiles/
    has none).  `achieved` / `frac` are priced on the rocprof figure when there is one (so the line can be re-derived
    from the committed profile), else on the events; `achieved_events` is always the events-based rate."""
    scale = 1e9 if unit == "GB/s" else 1e12
    ev = work / (us_events * 1e-6) / scale
    ns = prof.get("rocprof_avg_ns")
    ach = work / (ns * 1e-9) / scale if ns else ev
    return dict(bound=bound, achieved=round(ach, 1), peak=peak, unit=unit, frac=round(ach / peak, 4),
                traffic=prof.get("traffic_bytes"),
                traffic_source=("from " + prof["source"] + ", not this run") if prof.get("traffic_bytes") else None,
                kernel=kernel, us_per_launch_events=round(us_events, 2), rocprof_avg_ns=ns,
                frac_priced_on="rocprof_avg_ns (profiles/)" if ns else "us_per_launch_events (this run)",
                achieved_events=round(ev, 1), **extra)


def level0_probe(w, model, fwd_bwd, steps=30):
    """The DD-shaped step's dominant kernels are the two persistent level-0 kernels (dp_level0.hip): k_level0_fwd runs the
    whole level-0 forward (3 GraphConv layers of both stacks, assign head, X' = S^T Z, A' = S^T A S, readout) and
    k_level0_bwd its mirror, with each workgroup's bf16 adjacency rows resident in LDS.  Their launches are timed live by
    the library's own event pairs (dp_profile_level0) over `steps` eager steps of the SAME model and batch.

    Algorithmic bytes per launch: the adjacency passes the kernel stands for —
    p_f = 2L + 1 forward, p_b = 2(L - 1) + 1 backward — x B x N^2 x 2 bytes (bf16, the element the products multiply;
    the forward kernel reads the fp32 input once and the backward the bf16 A and A^T copies once: see `traffic`),
    plus 8 N (D + D_a + K) bytes of activations per graph and direction.  Returns None when the plan did not take
    the persistent kernels (N < 128, sync-BN, ...)."""
    from graph_pooling_amd import _lib
    import ctypes as C
    lib = _lib.load()
    _lib.check(lib.dp_profile_level0(1))
    for _ in range(steps):
        fwd_bwd()
    torch.cuda.synchronize()
    out = []
    for which in (0, 1):
        us, n = C.c_double(0.0), C.c_int(0)
        _lib.check(lib.dp_profile_level0_read(which, C.byref(us), C.byref(n)))
        out.append((us.value, n.value))
    _lib.check(lib.dp_profile_level0(0))
    if out[0][1] == 0 or out[1][1] == 0:
        return None
    B, N, H, L = w["B"], w["N"], w["H"], 3
    K = int(N * w["ratio"])
    D, Da = H * L, H * (L - 1) + K
    act = 8 * N * (D + Da + K) * B
    shape = f"B{B}_N{N}"
    res = {}
    for which, name, passes in ((1, "k_level0_bwd", 2 * (L - 1) + 1), (0, "k_level0_fwd", 2 * L + 1)):
        us = out[which][0] / out[which][1]
        alg = passes * B * N * N * 2 + act
        res[name] = _roofline(
            "hbm", f"{name}: persistent level-0 {'backward' if which else 'forward'}, one workgroup per (graph, row "
                   f"block), bf16 adjacency rows resident in LDS", alg, us, _profiled(name, shape), HBM_PEAK_GBS, "GB/s",
            algorithmic_bytes=alg, adjacency_passes_stood_for=passes, launches_timed=out[which][1],
            fp32_equiv=(lambda bts, ns: {"bytes": bts, "GB/s": round(bts / ns, 1), "frac": round(bts / ns / HBM_PEAK_GBS, 4),
                                         "note": "the same passes at the reference's 4-byte adjacency element"})(
                passes * B * N * N * 4 + act, (_profiled(name, shape).get("rocprof_avg_ns") or us * 1e3)),
            note="latency-bound by construction at B=20: the kernel reads the adjacency from HBM once instead of "
                 "once per pass, so HBM traffic is a fraction of the algorithmic bytes")
    return res


def roofline_probe(w, device, iters=200):
    """Average duration of the dominant kernel, HIP events on the launch stream: the level-0 adjacency
    aggregation  U[b] = A[b] (N x N) · V[b] (N x C),  C = H_embed + H_assign, exactly as the encoder plan runs it
    for 0/1 adjacency: k_aggregate on the bf16-packed adjacency (dp_adj_pack) with V's three bf16 planes already
    written by V's producer (presplit) — so only the aggregation kernel is inside the timed region.

    Bytes per launch are the ALGORITHMIC bytes at the element sizes the kernel really reads:
      A as bf16: B*N*ld*2;  V as 3 bf16 planes (k padded to 32, C to 16): B*3*ceil(C/16)*ceil(N/32)*4*256;
      U written as fp32: B*N*C*4.
    `fp32_equiv` re-expresses the same launch against the reference's fp32 operands (B*N*N*4 + 2*B*N*C*4)."""
    from graph_pooling_amd import _lib
    lib = _lib.load()
    B, N, Cc = w["B"], w["N"], 2 * w["H"]
    A = (torch.rand(B, N, N, device=device) < w["p"]).float()
    V = torch.randn(B, N, Cc, device=device)
    U = torch.empty(B, N, Cc, device=device)
    st = torch.cuda.current_stream()
    nb = lib.dp_adj_pack_bytes(B, N)
    pk = torch.empty(nb, device=device, dtype=torch.uint8)
    pkt = torch.empty(nb, device=device, dtype=torch.uint8)
    flag = torch.zeros(64, device=device, dtype=torch.int32)
    _lib.check(lib.dp_adj_pack(A.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(), B, N, st.cuda_stream))
    wsb = lib.dp_adj_aggregate_packed_workspace_bytes(B, N, Cc)
    ws = torch.empty(wsb, device=device, dtype=torch.uint8)

    def launch(presplit):
        _lib.check(lib.dp_adj_aggregate_packed(A.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(),
                                               V.data_ptr(), Cc, U.data_ptr(), Cc, B, N, Cc, 0, 0.0, presplit,
                                               ws.data_ptr(), wsb, st.cuda_stream))
    launch(0)                      # writes the split into the workspace
    packed = N >= 128              # below that the plan (and this call) run the fp32 panel kernel
    for _ in range(20):
        launch(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        launch(1)
    e1.record(st)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1000.0 / iters
    ld = lib.dp_adj_pack_ld(N)
    ct, k8 = (Cc + 15) // 16, ((N + 31) // 32) * 4
    fp32_bytes = B * N * N * 4 + 2 * B * N * Cc * 4
    bytes_alg = (B * N * ld * 2 + B * 3 * ct * k8 * 256 + B * N * Cc * 4) if packed else fp32_bytes
    name = "k_aggregate_packed" if packed else "k_aggregate_fp32"
    return _roofline("hbm", (f"k_aggregate_wide<{ct}>" if packed and B * ((N + 127) // 128) >= 512
                             else f"k_aggregate<false,{ct},{32 if B * ((N + 31) // 32) >= 256 else 16},4>")
                     + (" bf16-packed A x 3-plane bf16 V (exact)" if packed else " fp32 panel")
                     + " — level-0 aggregation A·[XW_e|XW_a]", bytes_alg, us, _profiled(name, f"B{B}_N{N}_C{Cc}"),
                     HBM_PEAK_GBS, "GB/s", algorithmic_bytes=bytes_alg,
                     fp32_equiv={"bytes": fp32_bytes, "GB/s": round(fp32_bytes / (us * 1e-6) / 1e9, 1)})


def mfma_probe(w, device, iters=50):
    """The pooling pass T = A^T S (N x N by N x K, K = int(N * ratio) clusters) of S^T A S, as the encoder plan runs it
    for 0/1 adjacency at K > 128: k_aggregate_wide_dma on the bf16-packed A^T against the exact 3-plane bf16 split
    of S (presplit by the softmax producer).  MFMA-bound: `achieved` prices the ALGORITHMIC flops 2*N*N*K per graph
    against the dense bf16 MFMA peak; `mfma_executed` counts the three plane products the exact split really issues
    (K padded to the kernel's 16-column tiles)."""
    from graph_pooling_amd import _lib
    lib = _lib.load()
    B, N = w["B"], w["N"]
    K = int(N * w["ratio"])
    A = (torch.rand(B, N, N, device=device) < w["p"]).float()
    V = torch.softmax(torch.randn(B, N, K, device=device), -1)
    U = torch.empty(B, N, K, device=device)
    st = torch.cuda.current_stream()
    nb = lib.dp_adj_pack_bytes(B, N)
    pk = torch.empty(nb, device=device, dtype=torch.uint8)
    pkt = torch.empty(nb, device=device, dtype=torch.uint8)
    flag = torch.zeros(64, device=device, dtype=torch.int32)
    _lib.check(lib.dp_adj_pack(A.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(), B, N, st.cuda_stream))
    wsb = lib.dp_adj_aggregate_packed_workspace_bytes(B, N, K)
    ws = torch.empty(wsb, device=device, dtype=torch.uint8)

    def launch(presplit):
        _lib.check(lib.dp_adj_aggregate_packed(A.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(),
                                               V.data_ptr(), K, U.data_ptr(), K, B, N, K, 1, 0.0, presplit,
                                               ws.data_ptr(), wsb, st.cuda_stream))
    launch(0)
    for _ in range(5):
        launch(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        launch(1)
    e1.record(st)
    e1.synchronize()
    # two launches per call when K > 128 (the flag-gated fp32 fallback exits at once); its few us are included
    us = e0.elapsed_time(e1) * 1000.0 / iters
    alg = 2.0 * B * N * N * K
    ct = (K + 15) // 16
    ct = ct if ct <= 8 else (ct + 1) // 2 * 2
    executed = 2.0 * B * N * (((N + 31) // 32) * 32) * ct * 16 * 3
    prof = _profiled("k_aggregate_wide_dma", f"B{B}_N{N}_K{K}")
    r = _roofline("mfma", f"k_aggregate_wide_dma<{ct},8>: T = A^T S of the pooling step, bf16-packed A^T x 3-plane bf16 S "
                          f"(exact), B={B} N={N} K={K}", alg, us, prof, MFMA_BF16_PEAK_TF, "TFLOP/s",
                  algorithmic_flops=alg,
                  mfma_executed={"flops": executed, "TFLOP/s": round(executed / (us * 1e-6) / 1e12, 1),
                                 "frac_of_bf16_peak": round(executed / (us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TF, 4)},
                  mfma_busy_pmc=prof.get("mfma_busy"))
    r["traffic"] = None
    r["traffic_source"] = None
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="dd", choices=sorted(WORKLOADS))
    ap.add_argument("--linkpred", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step into a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N > 1: BatchNorm statistics and the link-loss normaliser span all ranks (parity mode: the step "
                         "equals one reference step on the concatenated batch); default is local statistics")
    ap.add_argument("--eval", action="store_true",
                    help="time the forward-only evaluation step (model.predict: DP_MODE_EVAL forward + on-device arg-max, "
                         "train.py:30-58) instead of forward + loss + backward")
    ap.add_argument("--probe-only", action="store_true",
                    help="run only the roofline probe (used to profile the dominant kernel in isolation)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the DiffPool HIP path has no CPU fallback)")
    # DP_BENCH_REHEARSE=1: all ranks share GPU 0 and talk over gloo — lets the N > 1 code path be exercised
    # on a one-GPU box (RCCL refuses two ranks on one device). Never used for reported numbers.
    rehearse = os.environ.get("DP_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # whole-step graph capture next to a collective library: no asynchronous error-handling thread poking at
        # events while a stream is capturing (PyTorch's CUDA-graphs-with-NCCL note)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        os.environ.setdefault("NCCL_ASYNC_ERROR_HANDLING", "0")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; running {world} rank(s)", file=sys.stderr)

    w = WORKLOADS[args.workload]
    if args.probe_only:
        out = {"roofline": roofline_probe(w, device)}
        if w.get("roofline") == "mfma":
            out = {"roofline": mfma_probe(w, device), "roofline_hbm": out["roofline"]}
        print(json.dumps(out))
        return
    model, batch, cpu = make_model_and_batch(w, args.linkpred, device, seed_offset=rank)
    dp = None
    if world > 1:
        from graph_pooling_amd.parallel import DataParallelEncoder
        dp = DataParallelEncoder(model, sync_bn=args.sync_bn)
    # N > 1 on RCCL: the flat-gradient all-reduce is captured INTO the step's hipGraph (torch.distributed's NCCL
    # collectives are capturable), so a step stays one graph launch.  Whether this runtime can capture a collective
    # is tried first on a throw-away communicator: a failed capture can leave a communicator unusable, and the real
    # one must survive to run the collective eagerly instead.  DP_BENCH_GRAPH_ALLREDUCE=0 skips the attempt.
    graph_allreduce = False
    if dp is not None and not rehearse and not args.no_graph and os.environ.get("DP_BENCH_GRAPH_ALLREDUCE", "1") != "0":
        from graph_pooling_amd.parallel import collective_capture_works
        graph_allreduce = collective_capture_works(device, world, rank)
        if rank == 0:
            print(f"[bench] gradient all-reduce {'inside' if graph_allreduce else 'outside'} the hipGraph",
                  file=sys.stderr)

    def fwd_bwd():
        if args.eval:
            return model.predict(batch["x"], batch["adj"], batch["nn"], assign_x=batch["x"])
        model.zero_grad(set_to_none=True)
        ypred = model(batch["x"], batch["adj"], batch["nn"], assign_x=batch["x"])
        if args.linkpred:
            loss = model.loss(ypred, batch["label"], batch["adj"], batch["nn"])
        else:
            loss = model.loss(ypred, batch["label"])
        loss.backward()
        if graph_allreduce:
            dp.reduce_gradients()
        return loss

    # ---- optional hipGraph capture of the whole fwd+loss+bwd launch sequence
    graph = None
    side = torch.cuda.Stream(device)
    # (gloo's collectives on device tensors synchronise with the host: inside a stream capture that invalidates the
    # capture and leaves the stream unusable — a sync-BN rehearsal over gloo therefore runs eagerly)
    if not args.no_graph and not (rehearse and args.sync_bn):
        try:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fwd_bwd()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            model.zero_grad(set_to_none=True)
            # N > 1: other threads of this process (the collective library's watchdog) may touch the runtime while
            # this thread captures; only calls of THIS thread can invalidate the capture
            with torch.cuda.graph(g, **({"capture_error_mode": "thread_local"} if world > 1 else {})):
                static_loss = fwd_bwd()
            graph = g
        except Exception as e:                      # noqa: BLE001
            if rank == 0:
                print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly",
                      file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
        else:
            fwd_bwd()
        if dp is not None and not graph_allreduce:      # (otherwise the collective is part of fwd_bwd / the graph)
            dp.reduce_gradients()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = w["B"] * world * args.steps / dt
        out = {
            "metric": (f"graphs/sec forward-only evaluation (predict), {args.workload}" if args.eval else
                       "graphs/sec fwd+bwd, DD padded batch N_max=500" if args.workload == "dd"
                       else f"graphs/sec fwd+bwd, {args.workload}"),
            "value": round(value, 1), "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["name"] + (" [REHEARSAL: ranks share one GPU over gloo]" if rehearse else ""),
                       "graphs_per_gpu": w["B"], "global_batch": w["B"] * world,
                       "linkpred": bool(args.linkpred), "hip_graph": graph is not None,
                       "parallelism": f"dp{world}",
                       **({"allreduce_in_graph": bool(graph_allreduce and graph is not None),
                           "batchnorm": "sync" if args.sync_bn else "local"} if world > 1 else {})},
        }
        if world == 1 and not args.eval:
            if w.get("roofline") == "mfma":
                # the ER workload is the MFMA-bound one: its dominant kernel is the wide pooling product
                out["roofline"] = mfma_probe(w, device)
                out["roofline_hbm"] = roofline_probe(w, device)
            else:
                l0 = level0_probe(w, model, fwd_bwd) if w.get("model") != "set2set" else None
                if l0 is not None:
                    # the persistent level-0 kernels carry the step: the backward one is the longest kernel of the step
                    out["roofline"] = l0["k_level0_bwd"]
                    out["roofline_fwd"] = l0["k_level0_fwd"]
                    out["roofline_aggregate"] = roofline_probe(w, device)
                else:
                    out["roofline"] = roofline_probe(w, device)
            out["roofline"]["step"] = step_roofline(w, args.linkpred, ms)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(cpu, w, args.linkpred)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
