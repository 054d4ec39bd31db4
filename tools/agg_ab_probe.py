#!/usr/bin/env python3
"""A/B of the dominant DD kernel (k_aggregate on the packed adjacency, presplit V) between library builds, in ONE
process, interleaved rounds (cdna_hip_programming.md §5.4 rule 24): each library is loaded from its own path with
ctypes, the same buffers are used for all, and every round times `iters` back-to-back launches per library with HIP
events; reports median / min per library over the rounds.

    python tools/agg_ab_probe.py graph_pooling_amd/libdiffpool_hip.so graph_pooling_amd/libdiffpool_hip_r01f.so
"""
import ctypes as C
import json
import statistics
import sys

import torch

P, I, F, Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t


def bind(path):
    lib = C.CDLL(path)
    lib.dp_adj_pack_bytes.restype = Z
    lib.dp_adj_pack_bytes.argtypes = [I, I]
    lib.dp_adj_pack.restype = I
    lib.dp_adj_pack.argtypes = [P, P, P, P, I, I, P]
    lib.dp_adj_aggregate_packed_workspace_bytes.restype = Z
    lib.dp_adj_aggregate_packed_workspace_bytes.argtypes = [I, I, I]
    lib.dp_adj_aggregate_packed.restype = I
    lib.dp_adj_aggregate_packed.argtypes = [P, P, P, P, P, I, P, I, I, I, I, I, F, I, P, Z, P]
    return lib


def main():
    paths = sys.argv[1:]
    B, N, Cc, p_edge, iters, rounds = 20, 500, 40, 0.02, 200, 15
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    A = (torch.rand(B, N, N, device=dev) < p_edge).float()
    V = torch.randn(B, N, Cc, device=dev)
    U = torch.empty(B, N, Cc, device=dev)
    st = torch.cuda.current_stream()
    libs, state = [], []
    for path in paths:
        lib = bind(path)
        nb = lib.dp_adj_pack_bytes(B, N)
        pk = torch.empty(nb, device=dev, dtype=torch.uint8)
        pkt = torch.empty(nb, device=dev, dtype=torch.uint8)
        flag = torch.zeros(64, device=dev, dtype=torch.int32)
        assert lib.dp_adj_pack(A.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(), B, N, st.cuda_stream) == 0
        wsb = lib.dp_adj_aggregate_packed_workspace_bytes(B, N, Cc)
        ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        libs.append(lib)
        state.append((pk, pkt, flag, ws, wsb))

    def launch(i, presplit):
        pk, pkt, flag, ws, wsb = state[i]
        rc = libs[i].dp_adj_aggregate_packed(A.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(), V.data_ptr(),
                                             Cc, U.data_ptr(), Cc, B, N, Cc, 0, 0.0, presplit, ws.data_ptr(), wsb,
                                             st.cuda_stream)
        assert rc == 0
    outs = []
    for i in range(len(libs)):
        launch(i, 0)
        torch.cuda.synchronize()
        outs.append(U.clone())
        for _ in range(50):
            launch(i, 1)
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "the builds disagree"
    times = [[] for _ in libs]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(rounds):
        order = list(range(len(libs)))
        if r & 1:
            order.reverse()
        for i in order:
            e0.record(st)
            for _ in range(iters):
                launch(i, 1)
            e1.record(st)
            e1.synchronize()
            times[i].append(e0.elapsed_time(e1) * 1000.0 / iters)
    for path, t in zip(paths, times):
        print(json.dumps({"lib": path, "us_median": round(statistics.median(t), 3), "us_min": round(min(t), 3),
                          "us_max": round(max(t), 3), "rounds": rounds, "iters": iters}))


if __name__ == "__main__":
    main()
