"""MI355X-native DiffPool encoders behind the reference's nn.Module surface.

Same class names, constructor signatures, forward/loss signatures, attributes
(``assign_tensor``, ``link_loss``) and ``state_dict`` keys as the reference's
``encoders.py`` (JiaxuanYou/graph-pooling), so ``train.py`` / ``cross_val.py`` can drive these
modules unchanged.  The arithmetic is NOT torch: each forward / backward is
one call into libdiffpool_hip.so, which enqueues hand-written gfx950 kernels on the current
stream.  There is no CPU path: tensors must live on the GPU.

Reference lines (relative to the reference root):
  GraphConv                 encoders.py:945-974 (the DiffPool variant, commented out there)
  GcnEncoderGraph           encoders.py:976-1134
  GcnSet2SetEncoder         encoders.py:1137-1157
  SoftPoolingGcnEncoder     encoders.py:1160-1334
Reference defects and the semantics chosen here: SURVEY.md Appendix B / DESIGN.md.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init

from . import _lib
from .set2set import Set2Set


# ----------------------------------------------------------------------------- GraphConv
class PackedAdjacency:
    """The level-0 adjacency of a batch in the form the kernels multiply from: bf16 rows `pk` [B, N, ld] of A and `pkt`
    of A^T (ld = dp_adj_pack_ld(N); the same tensor for a symmetric adjacency), int16 storage.  Written by
    `DeviceBatchBuilder.build(..., packed=True)` (dp_build_batch_packed) or `PackedAdjacency.from_dense`; accepted in
    place of the fp32 `adj` by the encoders' forward when the configuration takes the persistent level-0 plan
    (dp_encoder_forward_packed raises DP_ERR_UNSUPPORTED otherwise).  The reference has no counterpart: its adjacency
    is always the dense fp32 batch (train.py:197)."""

    def __init__(self, pk: torch.Tensor, pkt: torch.Tensor, num_nodes_padded: int):
        if pk.dtype != torch.int16 or pkt.dtype != torch.int16 or pk.dim() != 3 or pk.shape != pkt.shape:
            raise ValueError("pk / pkt: int16 [B, N, ld] tensors of one shape")
        self.pk, self.pkt = pk, pkt
        self.B, self.N = int(pk.shape[0]), int(num_nodes_padded)
        if pk.shape[1] != self.N or pk.shape[2] != _lib.load().dp_adj_pack_ld(self.N):
            raise ValueError(f"packed rows must be [B, {self.N}, dp_adj_pack_ld({self.N})]")

    @classmethod
    def from_dense(cls, adj: torch.Tensor) -> "PackedAdjacency":
        """dp_adj_pack of a dense fp32 batch (every entry must be exactly representable in bf16, e.g. 0/1)."""
        _lib.require_gpu_tensor(adj, "adj")
        lib = _lib.load()
        B, N = int(adj.shape[0]), int(adj.shape[1])
        ld = lib.dp_adj_pack_ld(N)
        pk = torch.empty(B, N, ld, device=adj.device, dtype=torch.int16)
        pkt = torch.empty_like(pk)
        flag = torch.zeros(64, device=adj.device, dtype=torch.int32)
        _lib.check(lib.dp_adj_pack(adj.contiguous().float().data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(),
                                   B, N, _lib.current_stream()), "dp_adj_pack")
        if int(flag[0].item()) != 0:
            raise ValueError("the adjacency has entries that bf16 cannot hold exactly; use the fp32 entry")
        return cls(pk, pkt, N)


class _GraphConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, weight, bias, flags):
        lib = _lib.load()
        _lib.require_gpu_tensor(x, "x")
        x = x.contiguous().float()
        adj = adj.contiguous().float()
        B, n, fin = x.shape
        fout = weight.shape[1]
        y = torch.empty(B, n, fout, device=x.device, dtype=torch.float32)
        invn = torch.empty(B, n, device=x.device, dtype=torch.float32)
        wsb = lib.dp_gcn_layer_workspace_bytes(B, n, fin, fout)
        ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
        w = weight.contiguous()
        _lib.check(lib.dp_gcn_layer_fwd(x.data_ptr(), fin, adj.data_ptr(), w.data_ptr(), _lib.ptr(bias),
                                        y.data_ptr(), fout, invn.data_ptr(), B, n, fin, fout, flags,
                                        ws.data_ptr(), wsb, _lib.current_stream()), "dp_gcn_layer_fwd")
        ctx.save_for_backward(x, adj, w, y, invn)
        ctx.flags = flags
        ctx.has_bias = bias is not None
        ctx.ws = ws
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, adj, w, y, invn = ctx.saved_tensors
        B, n, fin = x.shape
        fout = w.shape[1]
        dy = dy.contiguous()
        need_dx, need_dadj = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty_like(x) if need_dx else None
        dadj = torch.empty_like(adj) if need_dadj else None
        dw = torch.empty_like(w)
        db = torch.empty(fout, device=x.device, dtype=torch.float32) if ctx.has_bias else None
        ws = ctx.ws
        _lib.check(lib.dp_gcn_layer_bwd(x.data_ptr(), fin, adj.data_ptr(), w.data_ptr(), y.data_ptr(), fout,
                                        invn.data_ptr(), dy.data_ptr(), fout, _lib.ptr(dx), fin, dw.data_ptr(),
                                        _lib.ptr(db), _lib.ptr(dadj), B, n, fin, fout, ctx.flags,
                                        ws.data_ptr(), ws.numel(), _lib.current_stream()), "dp_gcn_layer_bwd")
        return dx, dadj, dw, db, None


class GraphConv(nn.Module):
    """y = l2norm((adj @ x [+ x]) @ W + b) — the DiffPool GraphConv (encoders.py:945-974)."""

    def __init__(self, input_dim, output_dim, add_self=False, normalize_embedding=False, dropout=0.0, bias=True):
        super().__init__()
        self.add_self = add_self
        self.dropout = dropout
        self.normalize_embedding = normalize_embedding
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.weight = nn.Parameter(torch.zeros(input_dim, output_dim))
        self.bias = nn.Parameter(torch.zeros(output_dim)) if bias else None

    def _flags(self):
        return (_lib.F_ADD_SELF if self.add_self else 0) | (_lib.F_NORMALIZE if self.normalize_embedding else 0)

    def forward(self, x, adj):
        if self.dropout > 0.001:
            x = nn.functional.dropout(x, self.dropout, self.training)
        return _GraphConvFn.apply(x, adj, self.weight, self.bias, self._flags())


# ----------------------------------------------------------------------------- helpers
def _num_nodes_device(batch_num_nodes, device) -> Optional[torch.Tensor]:
    """batch_num_nodes (host numpy array as train.py:200 passes it, a list, or a tensor) -> int32[B] on
    the device.  This upload replaces construct_mask's host loop + H2D of a [B,N,1] mask
    (encoders.py:1035-1046)."""
    if batch_num_nodes is None:
        return None
    if isinstance(batch_num_nodes, torch.Tensor):
        t = batch_num_nodes
        if t.dtype != torch.int32:
            t = t.to(torch.int32)
        return t if t.device == device else t.to(device, non_blocking=True)
    arr = np.ascontiguousarray(np.asarray(batch_num_nodes, dtype=np.int32))
    return torch.from_numpy(arr).to(device, non_blocking=True)


def _fill_stack(cfg_stack, dims, w_offs, b_offs):
    cfg_stack.n_layers = len(dims) - 1
    for i, d in enumerate(dims):
        cfg_stack.dims[i] = int(d)
    for i in range(_lib.DP_MAX_LAYERS):
        cfg_stack.drop_off[i] = -1
    for i in range(len(dims) - 1):
        cfg_stack.w_off[i] = int(w_offs[i])
        cfg_stack.b_off[i] = int(b_offs[i])


class _EncoderFn(torch.autograd.Function):
    """One FFI call per pass: dp_encoder_forward / dp_encoder_backward."""

    @staticmethod
    def forward(ctx, owner, x, adj, assign_x, num_nodes, drop, needs_grad, labels, *params):
        # needs_grad is decided by the caller: grad mode is always off inside Function.forward
        lib = _lib.load()
        plan = owner._plan(x.shape[0], x.shape[1], x.device)
        B = plan.cfg.B
        ypred = torch.empty(B, plan.label_dim, device=x.device, dtype=torch.float32)
        save = torch.empty(plan.save_bytes, device=x.device, dtype=torch.uint8) if needs_grad else plan.eval_save()
        assign = assign_copy = None
        if plan.cfg.num_pooling > 0:
            if needs_grad:
                # a training forward owns a fresh save buffer: assign_tensor is a VIEW of the level-0 S kept there (no
                # second [B, N, K] write: 268 MB at the ER shape); the view keeps the buffer alive
                off, cnt = plan.assign_loc
                assign = save[off:off + 4 * cnt].view(torch.float32).view(B, plan.cfg.N, plan.cfg.n_nodes[1])
            else:   # the evaluation buffer is reused by the next call: hand out a copy
                assign = assign_copy = torch.empty(B, plan.cfg.N, plan.cfg.n_nodes[1], device=x.device,
                                                   dtype=torch.float32)
        stream = _lib.current_stream()
        tail = (_lib.ptr(assign_x), _lib.ptr(num_nodes), _lib.ptr(drop), ypred.data_ptr(), _lib.ptr(assign_copy),
                _lib.ptr(labels), save.data_ptr(), plan.save_bytes, plan.workspace.data_ptr(), plan.ws_bytes,
                _lib.MODE_TRAIN if needs_grad else _lib.MODE_EVAL, stream)
        if isinstance(adj, PackedAdjacency):
            _lib.check(lib.dp_encoder_forward_packed(C.byref(plan.cfg), owner._flat.data_ptr(), x.data_ptr(),
                                                     adj.pk.data_ptr(), adj.pkt.data_ptr(), *tail),
                       "dp_encoder_forward_packed")
        else:
            _lib.check(lib.dp_encoder_forward(C.byref(plan.cfg), owner._flat.data_ptr(), x.data_ptr(), adj.data_ptr(),
                                              *tail), "dp_encoder_forward")
        # a training forward cleared the backward accumulators in the plan's workspace: the FIRST backward of the most
        # recent training forward may skip its zero-fill (any other backward clears them itself)
        plan.prezero_owner = ctx if needs_grad else None
        ctx.owner, ctx.plan, ctx.save = owner, plan, save
        owner._last_save = (plan, save)
        ctx.inputs = (x, adj, assign_x, num_nodes, drop)
        ctx.set_materialize_grads(False)
        if assign is None:
            return ypred
        return ypred, assign

    @staticmethod
    def backward(ctx, d_ypred, d_assign=None):
        lib = _lib.load()
        owner, plan = ctx.owner, ctx.plan
        x, adj, assign_x, num_nodes, drop = ctx.inputs
        if d_ypred is None:
            d_ypred = torch.zeros(plan.cfg.B, plan.label_dim, device=x.device, dtype=torch.float32)
        d_ypred = d_ypred.contiguous()
        if d_assign is not None:
            d_assign = d_assign.contiguous()
        grads = torch.empty(plan.cfg.n_params, device=x.device, dtype=torch.float32)
        prezeroed = 1 if plan.prezero_owner is ctx else 0
        plan.prezero_owner = None
        tail = (_lib.ptr(assign_x), _lib.ptr(num_nodes), _lib.ptr(drop), d_ypred.data_ptr(), _lib.ptr(d_assign),
                grads.data_ptr(), ctx.save.data_ptr(), plan.save_bytes, plan.workspace.data_ptr(), plan.ws_bytes,
                prezeroed, _lib.current_stream())
        if isinstance(adj, PackedAdjacency):
            _lib.check(lib.dp_encoder_backward_packed(C.byref(plan.cfg), owner._flat.data_ptr(), x.data_ptr(),
                                                      adj.pk.data_ptr(), adj.pkt.data_ptr(), *tail),
                       "dp_encoder_backward_packed")
        else:
            _lib.check(lib.dp_encoder_backward(C.byref(plan.cfg), owner._flat.data_ptr(), x.data_ptr(), adj.data_ptr(),
                                               *tail), "dp_encoder_backward")
        owner._last_flat_grad = grads
        out = [None, None, None, None, None, None, None, None]
        for (off, numel, shape) in owner._flat_index:
            out.append(grads[off:off + numel].view(shape))
        return tuple(out)


class _Plan:
    """cfg + workspace for one (batch, padded-nodes) shape of one module."""

    def __init__(self, owner, B, N, device):
        lib = _lib.load()
        self.cfg = owner._build_cfg(B, N)
        self._exchange_cb = None
        sync = getattr(owner, "_sync_bn", None)
        if sync is not None and sync.active:
            # sync-BN: the library calls back between the launch that writes a BatchNorm site's local row partials
            # and the launch that combines them; both blocks live inside this plan's workspace
            def exchange(_user, local, gathered, nbytes, _stream, plan=self, sync=sync):
                try:
                    base = plan.workspace.data_ptr()
                    lo, go = local - base, gathered - base
                    src = plan.workspace[lo:lo + nbytes].view(torch.float32)
                    dst = plan.workspace[go:go + nbytes * sync.world].view(torch.float32)
                    sync.all_gather(dst, src)
                    return 0
                except Exception as e:      # noqa: BLE001 — never let an exception unwind through the C frames
                    sync.fail(e)            # peers are blocked in the same collective: abort the group, then raise
                    return 1
            self._exchange_cb = _lib.EXCHANGE_FN(exchange)
            self.cfg.bn_world = sync.world
            self.cfg.exchange = C.cast(self._exchange_cb, C.c_void_p)
        self.drop_segments, self.drop_total = owner._fill_dropout(self.cfg, B)
        self.label_dim = owner.label_dim
        self.save_bytes = lib.dp_encoder_save_bytes(C.byref(self.cfg))
        self.ws_bytes = lib.dp_encoder_workspace_bytes(C.byref(self.cfg))
        if self.save_bytes == 0 or self.ws_bytes == 0:
            _lib.check(-1, "dp_encoder_save_bytes / dp_encoder_workspace_bytes")
        # zero-filled ONCE: the persistent level-0 kernel's barrier block (start of the workspace) must be zero on first
        # use and cleans up after itself on every launch (diffpool_hip.h, dp_encoder_workspace_bytes)
        self.workspace = torch.zeros(self.ws_bytes, device=device, dtype=torch.uint8)
        self._eval_save = None
        self.prezero_owner = None
        self.device = device
        self.assign_loc = None           # (byte offset, float count) of the level-0 S inside a save buffer
        if self.cfg.num_pooling > 0:
            off, cnt = C.c_size_t(0), C.c_size_t(0)
            _lib.check(lib.dp_encoder_save_locate(C.byref(self.cfg), 0, _lib.SAVE_S, C.byref(off), C.byref(cnt)),
                       "dp_encoder_save_locate")
            self.assign_loc = (off.value, cnt.value)

    def eval_save(self):
        if self._eval_save is None:
            self._eval_save = torch.empty(self.save_bytes, device=self.device, dtype=torch.uint8)
        return self._eval_save


# ----------------------------------------------------------------------------- base encoder
class GcnEncoderGraph(nn.Module):
    """GCN graph encoder with per-layer max readout (encoders.py:976-1134)."""

    _readout = 0          # 0: max readout; 1: Set2Set
    _mask_readout = 0     # the base forward builds the mask but never uses it (encoders.py:1087)

    def __init__(self, input_dim, hidden_dim, embedding_dim, label_dim, num_layers,
                 pred_hidden_dims=[], concat=True, bn=True, dropout=0.0, args=None):
        super().__init__()
        self.concat = concat
        add_self = not concat
        self.bn = bn
        self.num_layers = num_layers
        self.num_aggs = 1
        self.bias = True
        if args is not None:
            self.bias = args.bias
        self.input_dim, self.hidden_dim, self.embedding_dim = input_dim, hidden_dim, embedding_dim
        self.conv_first, self.conv_block, self.conv_last = self.build_conv_layers(
            input_dim, hidden_dim, embedding_dim, num_layers, add_self, normalize=True, dropout=dropout)
        self.act = nn.ReLU()
        self.label_dim = label_dim
        if concat:
            self.pred_input_dim = hidden_dim * (num_layers - 1) + embedding_dim
        else:
            self.pred_input_dim = embedding_dim
        self.pred_model = self.build_pred_layers(self.pred_input_dim, pred_hidden_dims, label_dim,
                                                 num_aggs=self.num_aggs)
        self._init_graph_convs()
        self._flat = None
        self._flat_index = []
        self._flat_params = []
        self._plans = {}
        self._last_flat_grad = None
        self._last_save = None

    # -- construction (same names as the reference so state_dict keys match, Appendix D)
    def build_conv_layers(self, input_dim, hidden_dim, embedding_dim, num_layers, add_self,
                          normalize=False, dropout=0.0):
        conv_first = GraphConv(input_dim=input_dim, output_dim=hidden_dim, add_self=add_self,
                               normalize_embedding=normalize, bias=self.bias)
        conv_block = nn.ModuleList(
            [GraphConv(input_dim=hidden_dim, output_dim=hidden_dim, add_self=add_self,
                       normalize_embedding=normalize, dropout=dropout, bias=self.bias)
             for _ in range(num_layers - 2)])
        conv_last = GraphConv(input_dim=hidden_dim, output_dim=embedding_dim, add_self=add_self,
                              normalize_embedding=normalize, bias=self.bias)
        return conv_first, conv_block, conv_last

    def build_pred_layers(self, pred_input_dim, pred_hidden_dims, label_dim, num_aggs=1):
        pred_input_dim = pred_input_dim * num_aggs
        if len(pred_hidden_dims) == 0:
            return nn.Linear(pred_input_dim, label_dim)
        layers = []
        for pred_dim in pred_hidden_dims:
            layers.append(nn.Linear(pred_input_dim, pred_dim))
            layers.append(self.act)
            pred_input_dim = pred_dim
        layers.append(nn.Linear(pred_dim, label_dim))
        return nn.Sequential(*layers)

    def _init_graph_convs(self):
        # encoders.py:1003-1007 / 1225-1229
        for m in self.modules():
            if isinstance(m, GraphConv):
                init.xavier_uniform_(m.weight.data, gain=nn.init.calculate_gain('relu'))
                if m.bias is not None:
                    init.constant_(m.bias.data, 0.0)

    # -- layout of the flat parameter buffer
    def _stack_modules(self, first, block, last):
        return [first] + list(block) + [last]

    def _graph_param_groups(self):
        """[(kind, level, [GraphConv...] | Linear)] in flat-buffer order."""
        return [("embed", 0, self._stack_modules(self.conv_first, self.conv_block, self.conv_last))]

    def _pred_linears(self) -> List[nn.Linear]:
        if isinstance(self.pred_model, nn.Linear):
            return [self.pred_model]
        return [m for m in self.pred_model if isinstance(m, nn.Linear)]

    def _tail_params(self):
        ps = []
        for lin in self._pred_linears():
            ps += [lin.weight, lin.bias]
        return ps

    def _ordered_params(self):
        ps = []
        for kind, _, mods in self._graph_param_groups():
            if kind == "assign_pred":
                ps += [mods.weight, mods.bias]
            else:
                for m in mods:
                    ps.append(m.weight)
                    if m.bias is not None:
                        ps.append(m.bias)
        n_graph = len(ps)
        ps += self._tail_params()
        return ps, n_graph

    def _ensure_flat(self, device):
        """Keep every parameter a view into ONE flat fp32 buffer (what the kernels index and what the
        data-parallel wrapper all-reduces).  Re-flattens after .cuda()/.to()/load_state_dict re-bound
        the tensors."""
        params, n_graph = self._ordered_params()
        ok = self._flat is not None and self._flat.device == device and len(params) == len(self._flat_index)
        if ok:
            base = self._flat.data_ptr()
            for p, (off, numel, _) in zip(params, self._flat_index):
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
        if ok:
            return
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, device=device, dtype=torch.float32)
        index, off = [], 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            index.append((off, n, tuple(p.shape)))
            off += n
        for p, (o, n, shape) in zip(params, index):
            p.data = flat[o:o + n].view(shape)
        self._flat, self._flat_index, self._flat_params = flat, index, params
        self._n_graph_floats = sum(n for (_, n, _) in index[:n_graph])
        self._plans = {}

    def _offsets(self):
        return {id(p): off for p, (off, _, _) in zip(self._flat_params, self._flat_index)}

    def _flags(self):
        f = 0
        if self.bn:
            f |= _lib.F_BN
        if not self.concat:
            f |= _lib.F_ADD_SELF | _lib.F_LAST_ONLY
        return f

    def _stack_cfg(self, st, mods, offs):
        dims = [mods[0].input_dim] + [m.output_dim for m in mods]
        _fill_stack(st, dims, [offs[id(m.weight)] for m in mods],
                    [offs[id(m.bias)] if m.bias is not None else -1 for m in mods])
        return dims

    def _build_cfg(self, B, N):
        cfg = _lib.EncoderCfg()
        offs = self._offsets()
        cfg.B, cfg.N = B, N
        cfg.num_pooling = 0
        cfg.n_nodes[0] = N
        self._stack_cfg(cfg.embed[0], self._stack_modules(self.conv_first, self.conv_block, self.conv_last), offs)
        self._fill_pred(cfg, offs, self.pred_input_dim)
        cfg.flags = self._flags()
        cfg.readout = self._readout
        cfg.mask_readout = self._mask_readout
        cfg.n_params = self._flat.numel()
        cfg.n_graph_params = self._n_graph_floats
        return cfg

    def _fill_pred(self, cfg, offs, in_dim):
        lins = self._pred_linears()
        if len(lins) > _lib.DP_MAX_PRED + 1:
            raise ValueError(f"pred_model has {len(lins)} Linear layers; the HIP path supports {_lib.DP_MAX_PRED + 1}")
        cfg.n_pred = len(lins)
        cfg.pred_dims[0] = in_dim
        for i, lin in enumerate(lins):
            cfg.pred_dims[i + 1] = lin.out_features
            cfg.pred_w_off[i] = offs[id(lin.weight)]
            cfg.pred_b_off[i] = offs[id(lin.bias)] if lin.bias is not None else -1

    def _plan(self, B, N, device):
        key = (B, N)
        plan = self._plans.get(key)
        if plan is None:
            plan = _Plan(self, B, N, device)
            self._plans[key] = plan
        return plan

    def _fill_dropout(self, cfg, B):
        """GraphConv layers built with dropout > 0 (the conv_block layers of the after-pool stacks of DiffPool, of
        the only stack of the base encoders, when the model is constructed with dropout: encoders.py:1013-1016,
        1180-1183) apply nn.Dropout to their INPUT
        (encoders.py:962-964).  Give each such layer a slot [B, n, d_in] in one mask buffer and record its offset
        in the stack config; returns ([(offset, numel, p)], total floats)."""
        segs, total = [], 0
        n_level = [int(cfg.n_nodes[j]) for j in range(cfg.num_pooling + 1)]
        for kind, lvl, mods in self._graph_param_groups():
            if kind not in ("embed", "assign"):
                continue
            st = cfg.embed[lvl] if kind == "embed" else cfg.assign[lvl]
            for l, m in enumerate(mods):
                if m.dropout > 0.001:
                    if l == 0:
                        raise NotImplementedError("dropout on the first GraphConv of a stack (the reference never "
                                                  "builds one, encoders.py:1010-1012)")
                    numel = B * n_level[lvl] * m.input_dim
                    st.drop_off[l] = total
                    segs.append((total, numel, float(m.dropout)))
                    total += numel
        return segs, total

    def _draw_dropout(self, plan, device):
        """Fresh masks for one training step: 0 or 1/(1-p), torch's generator on the device."""
        if not (self.training and plan.drop_total):
            return None
        drop = torch.empty(plan.drop_total, device=device, dtype=torch.float32)
        for off, numel, p in plan.drop_segments:
            drop[off:off + numel].bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
        return drop

    def _run(self, x, adj, batch_num_nodes, assign_x=None, labels=None):
        _lib.require_gpu_tensor(x, "x")
        if isinstance(adj, PackedAdjacency):
            # the level-0 adjacency in the kernels' own bf16 form (DeviceBatchBuilder(packed=True)): no fp32 batch exists
            if x.dim() != 3 or (adj.B, adj.N) != tuple(x.shape[:2]):
                raise ValueError(f"expected x [B,N,F] for a packed adjacency of {adj.B} graphs x {adj.N} nodes, got "
                                 f"{tuple(x.shape)}")
            if adj.pk.device != x.device:
                raise ValueError("x and the packed adjacency are on different devices")
            x = x.contiguous().float()
        else:
            _lib.require_gpu_tensor(adj, "adj")
            if x.dim() != 3 or adj.dim() != 3 or adj.shape[1] != adj.shape[2] or adj.shape[:2] != x.shape[:2]:
                raise ValueError(f"expected x [B,N,F] and adj [B,N,N], got {tuple(x.shape)} and {tuple(adj.shape)}")
            x = x.contiguous().float()
            adj = adj.contiguous().float()
        if assign_x is not None:
            assign_x = assign_x.contiguous().float()
        self._ensure_flat(x.device)
        nn_dev = _num_nodes_device(batch_num_nodes, x.device)
        if nn_dev is not None and nn_dev.numel() != x.shape[0]:
            raise ValueError("batch_num_nodes must have one entry per graph")
        drop = getattr(self, "_forced_dropout_mask", None)       # tests inject the oracle's masks here
        if drop is None:
            drop = self._draw_dropout(self._plan(x.shape[0], x.shape[1], x.device), x.device)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._flat_params)
        return _EncoderFn.apply(self, x, adj, assign_x, nn_dev, drop, needs_grad, labels, *self._flat_params)

    # -- reference surface
    def construct_mask(self, max_nodes, batch_num_nodes):
        """[B, max_nodes, 1] float mask (encoders.py:1035-1046).  Kept for callers; the kernels take
        num_nodes directly and never materialise it."""
        device = self._flat.device if self._flat is not None else self.conv_first.weight.device
        nn_dev = _num_nodes_device(batch_num_nodes, device)
        idx = torch.arange(max_nodes, device=device).unsqueeze(0)
        return (idx < nn_dev.unsqueeze(1)).float().unsqueeze(2)

    def forward(self, x, adj, batch_num_nodes=None, **kwargs):
        if x.shape[2] != self.input_dim:
            raise ValueError(f"x has {x.shape[2]} features, the encoder was built for {self.input_dim}")
        return self._run(x, adj, batch_num_nodes, labels=getattr(self, "_predict_labels", None))

    def saved_activation(self, level, what):
        """A view of one activation the LAST forward call kept in its save buffer: what in {'assign', 'xpool',
        'adjpool', 'embedding', 'assign_embedding', 'readout_argmax'} of pooling level `level` — every level's S_j, X'_j = S_j^T Z_j,
        A'_j = S_j^T A_j S_j (encoders.py:1273-1279).  The reference keeps only the last S (`assign_tensor`,
        train.py:218-219 logs it); this reads any level back without a second pass.  The view aliases the buffer:
        clone it if it must outlive the next forward under no_grad."""
        field = {"assign": _lib.SAVE_S, "xpool": _lib.SAVE_XPOOL, "adjpool": _lib.SAVE_ADJPOOL,
                 "embedding": _lib.SAVE_Z, "assign_embedding": _lib.SAVE_ZASSIGN,
                 "readout_argmax": _lib.SAVE_ARGMAX}[what]
        if getattr(self, "_last_save", None) is None:
            raise RuntimeError("saved_activation(): no forward pass has run yet")
        plan, save = self._last_save
        off, cnt = C.c_size_t(0), C.c_size_t(0)
        _lib.check(_lib.load().dp_encoder_save_locate(C.byref(plan.cfg), level, field, C.byref(off), C.byref(cnt)),
                   "dp_encoder_save_locate")
        raw = save[off.value:off.value + 4 * cnt.value]
        B = plan.cfg.B
        if what == "readout_argmax":       # int32 [B, readout width]; -1 where a masked (zero) row holds the maximum
            return raw.view(torch.int32).view(B, -1)
        flat = raw.view(torch.float32)
        n = plan.cfg.n_nodes[level]
        if what in ("embedding", "assign_embedding"):
            return flat.view(B, n, -1)
        K = plan.cfg.n_nodes[level + 1]
        return flat.view(B, n, K) if what == "assign" else flat.view(B, K, -1)

    @torch.no_grad()
    def predict(self, x, adj, batch_num_nodes=None, **kwargs):
        """The reference's evaluate() inner loop (train.py:42-44: forward, `torch.max(ypred, 1)`, `.cpu()`), kept on
        the device: a DP_MODE_EVAL forward (activations go to one reusable buffer, nothing is prepared for a backward
        pass) whose prediction-head launch also writes the arg-max class as an int64 [B] device tensor — the caller
        moves B integers, not B x C logits, and no arg-max launch runs."""
        labels = torch.empty(x.shape[0], device=x.device, dtype=torch.int64)
        self._predict_labels = labels
        try:
            self.forward(x, adj, batch_num_nodes, **kwargs)
        finally:
            self._predict_labels = None
        return labels

    def loss(self, pred, label, type='softmax'):
        if type == 'softmax':
            return _loss(self, pred, label, None, None, None, False)
        elif type == 'margin':
            batch_size = pred.size()[0]
            label_onehot = torch.zeros(batch_size, self.label_dim, device=pred.device).long()
            label_onehot.scatter_(1, label.view(-1, 1), 1)
            return torch.nn.MultiLabelMarginLoss()(pred, label_onehot)


# ----------------------------------------------------------------------------- loss
_UNIT_SEED = {}          # device -> the scalar 1.0 that seeds loss.backward()


def _unit_seed(device):
    """A cached device scalar 1.0.  `loss.backward()` (train.py:208) seeds the backward pass with ones_like(loss) — a
    fill launch per step; `_Loss.backward` hands autograd this tensor instead and ARMS the loss node (see _Loss), so
    the gradient (softmax - onehot) / B that the loss kernel already wrote goes to the prediction head as it is: no
    seed fill, no cross-entropy backward launch."""
    t = _UNIT_SEED.get(device)
    if t is None:
        if device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return None            # GPU path only; never allocate the cached scalar from a graph's private pool
        t = torch.ones((), device=device, dtype=torch.float32)
        _UNIT_SEED[device] = t
    return t


class _Loss(torch.Tensor):
    """The loss tensor `model.loss()` returns: an ordinary tensor whose parameterless `.backward()` seeds autograd with
    the cached unit scalar (see _unit_seed) and tells the loss node so EXPLICITLY: `_dp_node` is the _LossFn backward
    node of this very tensor, and `unit_armed` is set on it only for the duration of a root `.backward()` with no
    `gradient=` — the one situation in which the upstream gradient is exactly 1.  Anything else — `(loss * 1)
    .backward()`, `loss.backward(gradient=g)`, `torch.autograd.grad(loss, ...)` — never arms the node and takes the
    general path (tests/test_gpu_model.py::test_loss_backward_fast_path_equals_every_other_way_of_calling_it).  Every
    other operation returns plain tensors."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        with torch._C.DisableTorchFunctionSubclass():
            node = None
            if func is torch.Tensor.backward and len(args) == 1 and kwargs.get("gradient") is None \
                    and isinstance(args[0], _Loss) and args[0].dim() == 0 and args[0].dtype == torch.float32:
                node = getattr(args[0], "_dp_node", None)
                seed = _unit_seed(args[0].device) if node is not None else None
                if seed is not None:
                    kwargs = dict(kwargs)
                    kwargs["gradient"] = seed
                    node.unit_armed = True
                else:
                    node = None
            try:
                return func(*args, **kwargs)
            finally:
                if node is not None:
                    node.unit_armed = False


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label, S, adj, num_nodes, linkpred, link_norm=None):
        lib = _lib.load()
        _lib.require_gpu_tensor(pred, "pred")
        pred = pred.contiguous().float()
        label = label.contiguous().to(device=pred.device, dtype=torch.int64)
        B, Cc = pred.shape
        N = K = 0
        if linkpred:
            S = S.contiguous()
            adj = adj.contiguous().float()
            N, K = S.shape[1], S.shape[2]
        wsb = lib.dp_loss_workspace_bytes(B, max(N, 1), max(K, 1), int(linkpred))
        ws = torch.empty(wsb, device=pred.device, dtype=torch.uint8)
        out = torch.empty(2, device=pred.device, dtype=torch.float32)
        prob = torch.empty(B, Cc, device=pred.device, dtype=torch.float32)
        dunit = torch.empty(B, Cc, device=pred.device, dtype=torch.float32)
        _lib.check(lib.dp_loss_forward(pred.data_ptr(), label.data_ptr(), _lib.ptr(S) if linkpred else None,
                                       _lib.ptr(adj) if linkpred else None, _lib.ptr(num_nodes),
                                       _lib.ptr(link_norm) if linkpred else None,
                                       out.data_ptr(), prob.data_ptr(), dunit.data_ptr(), B, Cc, N, K, int(linkpred),
                                       ws.data_ptr(), wsb, _lib.current_stream()), "dp_loss_forward")
        ctx.saved = (prob, dunit, label, S if linkpred else None, adj if linkpred else None, num_nodes, ws)
        ctx.link_norm = link_norm if linkpred else None
        ctx.dims = (B, Cc, N, K, bool(linkpred))
        total, link = out[0], out[1]
        ctx.mark_non_differentiable(link)
        ctx.set_materialize_grads(False)      # no zero-fill launch for the unused gradient of `link`
        return total, link

    @staticmethod
    def backward(ctx, dtotal, _dlink):
        lib = _lib.load()
        prob, dunit, label, S, adj, num_nodes, ws = ctx.saved
        B, Cc, N, K, linkpred = ctx.dims
        if dtotal is None:
            return None, None, None, None, None, None, None
        unit = bool(getattr(ctx, "unit_armed", False))     # armed by _Loss.backward: the upstream gradient is exactly 1
        if unit and not linkpred:
            return dunit, None, None, None, None, None, None     # written by the loss kernel: nothing to launch
        dtotal = dtotal.contiguous().float()
        dpred = dunit if unit else torch.empty(B, Cc, device=prob.device, dtype=torch.float32)
        dS = torch.empty_like(S) if linkpred else None
        _lib.check(lib.dp_loss_backward(prob.data_ptr(), label.data_ptr(), _lib.ptr(S), _lib.ptr(adj),
                                        _lib.ptr(num_nodes), _lib.ptr(ctx.link_norm),
                                        None if unit else dtotal.data_ptr(),
                                        None if unit else dpred.data_ptr(), _lib.ptr(dS),
                                        B, Cc, N, K, int(linkpred), ws.data_ptr(), ws.numel(),
                                        _lib.current_stream()), "dp_loss_backward")
        return dpred, None, dS, None, None, None, None


def _loss(owner, pred, label, S, adj, batch_num_nodes, linkpred):
    nn_dev = _num_nodes_device(batch_num_nodes, pred.device) if linkpred else None
    link_norm = None
    sync = getattr(owner, "_sync_bn", None)
    if linkpred and sync is not None and sync.active:
        # the link loss divides by sum_b n_b^2 over the WHOLE batch (encoders.py:1326,1331): every rank uses
        # (global sum) / world, so the mean over ranks of the per-rank terms is the single-batch loss
        n_eff = nn_dev.clamp(max=S.shape[1]).float() if nn_dev is not None else \
            torch.full((S.shape[0],), float(S.shape[1]), device=pred.device)
        link_norm = sync.all_reduce_sum((n_eff * n_eff).sum().reshape(1)) / float(sync.world)
    total, link = _LossFn.apply(pred, label, S, adj, nn_dev, linkpred, link_norm)
    if linkpred:
        owner.link_loss = link
    _unit_seed(pred.device)                   # make sure the cached seed exists before anyone captures a graph
    if not total.requires_grad:
        return total
    node = total.grad_fn                      # the _LossFn backward node (== the ctx its backward() receives)
    out = total.as_subclass(_Loss)
    out._dp_node = node
    return out


# ----------------------------------------------------------------------------- Set2Set encoder
class GcnSet2SetEncoder(GcnEncoderGraph):
    """GCN + Set2Set readout (encoders.py:1137-1157)."""

    _readout = 1
    _mask_readout = 1

    def __init__(self, input_dim, hidden_dim, embedding_dim, label_dim, num_layers,
                 pred_hidden_dims=[], concat=True, bn=True, dropout=0.0, args=None):
        super().__init__(input_dim, hidden_dim, embedding_dim, label_dim, num_layers, pred_hidden_dims,
                         concat, bn, dropout, args=args)
        self.s2s = Set2Set(self.pred_input_dim, self.pred_input_dim * 2)

    def _tail_params(self):
        l = self.s2s.lstm
        return [l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0, self.s2s.pred.weight,
                self.s2s.pred.bias] + super()._tail_params()

    def _build_cfg(self, B, N):
        cfg = super()._build_cfg(B, N)
        offs = self._offsets()
        l = self.s2s.lstm
        for i, p in enumerate([l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0, self.s2s.pred.weight,
                               self.s2s.pred.bias]):
            cfg.s2s_off[i] = offs[id(p)]
        return cfg

    def _flags(self):
        # gcn_forward always concatenates every layer (encoders.py:1078); the Set2Set width is pred_input_dim
        f = _lib.F_BN if self.bn else 0
        if not self.concat:
            f |= _lib.F_ADD_SELF
        return f


# ----------------------------------------------------------------------------- DiffPool
class SoftPoolingGcnEncoder(GcnEncoderGraph):
    """DiffPool (encoders.py:1160-1334)."""

    _mask_readout = 1

    def __init__(self, max_num_nodes, input_dim, hidden_dim, embedding_dim, label_dim, num_layers,
                 assign_hidden_dim, assign_ratio=0.25, assign_num_layers=-1, num_pooling=1,
                 pred_hidden_dims=[50], concat=True, bn=True, dropout=0.0, linkpred=True,
                 assign_input_dim=-1, args=None):
        # the reference does not forward bn / dropout to the level-0 encoder (encoders.py:1172-1173)
        super().__init__(input_dim, hidden_dim, embedding_dim, label_dim, num_layers,
                         pred_hidden_dims=pred_hidden_dims, concat=concat, args=args)
        if not concat:
            raise ValueError("SoftPoolingGcnEncoder(concat=False): the reference builds the after-pool GCN for "
                             "input width embedding_dim but feeds it the concatenated width (encoders.py:1078, "
                             "1186) and fails with a shape error; not supported")
        if num_pooling < 1 or num_pooling > _lib.DP_MAX_LEVELS:
            raise ValueError(f"num_pooling must be in [1, {_lib.DP_MAX_LEVELS}]")
        add_self = not concat
        self.num_pooling = num_pooling
        self.linkpred = linkpred
        self.assign_ent = True
        self.max_num_nodes = max_num_nodes
        if assign_num_layers == -1:
            assign_num_layers = num_layers
        if assign_num_layers != num_layers:
            raise ValueError("assign_num_layers must equal num_layers: the reference sizes assign_pred with "
                             "num_layers (encoders.py:1209) and fails otherwise")
        if assign_input_dim == -1:
            assign_input_dim = input_dim
        self.assign_input_dim = assign_input_dim

        # Every level is registered (the reference keeps earlier levels in plain lists, Appendix B D2).
        # The LAST level carries the reference's attribute names so num_pooling == 1 state_dicts match.
        self.conv_first_after_pool, self.conv_block_after_pool, self.conv_last_after_pool = [], [], []
        self.assign_conv_first_modules, self.assign_conv_block_modules = [], []
        self.assign_conv_last_modules, self.assign_pred_modules = [], []
        self.assign_dims = []
        assign_dim = int(max_num_nodes * assign_ratio)
        a_in = assign_input_dim
        for i in range(num_pooling):
            last = i == num_pooling - 1
            cf, cb, cl = self.build_conv_layers(self.pred_input_dim, hidden_dim, embedding_dim, num_layers,
                                                add_self, normalize=True, dropout=dropout)
            af, ab, al = self.build_conv_layers(a_in, assign_hidden_dim, assign_dim, assign_num_layers,
                                                add_self, normalize=True)
            d_a = assign_hidden_dim * (num_layers - 1) + assign_dim if concat else assign_dim
            ap = self.build_pred_layers(d_a, [], assign_dim, num_aggs=1)
            if last:
                self.conv_first2, self.conv_block2, self.conv_last2 = cf, cb, cl
                self.assign_conv_first, self.assign_conv_block, self.assign_conv_last = af, ab, al
                self.assign_pred = ap
            else:
                setattr(self, f"conv_first_after_pool_{i}", cf)
                setattr(self, f"conv_block_after_pool_{i}", cb)
                setattr(self, f"conv_last_after_pool_{i}", cl)
                setattr(self, f"assign_conv_first_{i}", af)
                setattr(self, f"assign_conv_block_{i}", ab)
                setattr(self, f"assign_conv_last_{i}", al)
                setattr(self, f"assign_pred_{i}", ap)
            self.conv_first_after_pool.append(cf)
            self.conv_block_after_pool.append(cb)
            self.conv_last_after_pool.append(cl)
            self.assign_conv_first_modules.append(af)
            self.assign_conv_block_modules.append(ab)
            self.assign_conv_last_modules.append(al)
            self.assign_pred_modules.append(ap)
            self.assign_dims.append(assign_dim)
            a_in = self.pred_input_dim           # Appendix B D4: level >= 1 assign GCN is fed X' (width D)
            assign_dim = int(assign_dim * assign_ratio)
        if min(self.assign_dims) < 1:
            raise ValueError(f"assign_ratio={assign_ratio} gives an empty cluster level: {self.assign_dims}")

        self.pred_model = self.build_pred_layers(self.pred_input_dim * (num_pooling + 1), pred_hidden_dims,
                                                 label_dim, num_aggs=self.num_aggs)
        self._init_graph_convs()
        self.assign_tensor = None
        self.link_loss = None

    def _graph_param_groups(self):
        groups = super()._graph_param_groups()
        for i in range(self.num_pooling):
            groups.append(("embed", i + 1, self._stack_modules(self.conv_first_after_pool[i],
                                                                self.conv_block_after_pool[i],
                                                                self.conv_last_after_pool[i])))
            groups.append(("assign", i, self._stack_modules(self.assign_conv_first_modules[i],
                                                             self.assign_conv_block_modules[i],
                                                             self.assign_conv_last_modules[i])))
            groups.append(("assign_pred", i, self.assign_pred_modules[i]))
        return groups

    def _flags(self):
        return _lib.F_BN          # self.bn is always True here (encoders.py:1172-1173)

    def _build_cfg(self, B, N):
        if N != self.max_num_nodes:
            raise ValueError(f"adj is padded to {N} nodes but the model was built for max_num_nodes="
                             f"{self.max_num_nodes} (the assignment width is fixed at construction, "
                             "encoders.py:1203)")
        cfg = _lib.EncoderCfg()
        offs = self._offsets()
        cfg.B, cfg.N = B, N
        cfg.num_pooling = self.num_pooling
        cfg.n_nodes[0] = N
        for kind, lvl, mods in self._graph_param_groups():
            if kind == "embed":
                self._stack_cfg(cfg.embed[lvl], mods, offs)
            elif kind == "assign":
                self._stack_cfg(cfg.assign[lvl], mods, offs)
                cfg.n_nodes[lvl + 1] = self.assign_dims[lvl]
            else:
                cfg.assign_pred_w_off[lvl] = offs[id(mods.weight)]
                cfg.assign_pred_b_off[lvl] = offs[id(mods.bias)]
        self._fill_pred(cfg, offs, self.pred_input_dim * (self.num_pooling + 1))
        cfg.flags = self._flags()
        cfg.readout = 0
        cfg.mask_readout = 1
        cfg.n_params = self._flat.numel()
        cfg.n_graph_params = self._n_graph_floats
        return cfg

    def forward(self, x, adj, batch_num_nodes, **kwargs):
        x_a = kwargs['assign_x'] if 'assign_x' in kwargs else x
        if x.shape[2] != self.input_dim or x_a.shape[2] != self.assign_input_dim:
            raise ValueError(f"feature widths {x.shape[2]}/{x_a.shape[2]} do not match the model "
                             f"({self.input_dim}/{self.assign_input_dim})")
        ypred, assign = self._run(x, adj, batch_num_nodes, assign_x=x_a, labels=getattr(self, "_predict_labels", None))
        # level-0 assignment [B, N, K_0]; == the reference's attribute when num_pooling == 1
        self.assign_tensor = assign
        return ypred

    def loss(self, pred, label, adj=None, batch_num_nodes=None, adj_hop=1):
        if adj_hop != 1:
            raise NotImplementedError("adj_hop > 1 is never used by the reference's callers (train.py:207)")
        if self.linkpred:
            if adj is None:
                raise ValueError("linkpred=True: loss() needs adj (train.py:207 passes it)")
            if isinstance(adj, PackedAdjacency):
                raise TypeError("the link-prediction loss reads the dense fp32 adjacency; it has no packed form")
            return _loss(self, pred, label, self.assign_tensor, adj, batch_num_nodes, True)
        return _loss(self, pred, label, None, None, None, False)
