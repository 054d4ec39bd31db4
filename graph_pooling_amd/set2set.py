"""Set2Set readout on MI355X behind the reference's module surface (set2set.py:8-57).

``Set2Set(input_dim, hidden_dim)``: n LSTM-attention steps over ``embedding [B, n, d]`` -> ``[B, d]``.
The parameters live in the same ``nn.LSTM`` / ``nn.Linear`` containers as the reference (state_dict
keys ``lstm.weight_ih_l0`` ... ``pred.bias``); the n sequential steps run inside one persistent HIP
kernel per pass (libdiffpool_hip.so: dp_set2set_fwd / dp_set2set_bwd), not as n x 6 torch ops.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib


class _Set2SetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, w_ih, w_hh, b_ih, b_hh, wp, bp):
        lib = _lib.load()
        _lib.require_gpu_tensor(emb, "embedding")
        emb = emb.contiguous().float()
        B, n, d = emb.shape
        ts = [t.contiguous() for t in (w_ih, w_hh, b_ih, b_hh, wp, bp)]
        out = torch.empty(B, d, device=emb.device, dtype=torch.float32)
        sb = lib.dp_set2set_save_bytes(B, n, d)
        save = torch.empty(sb, device=emb.device, dtype=torch.uint8)
        _lib.check(lib.dp_set2set_fwd(emb.data_ptr(), d, *[t.data_ptr() for t in ts], out.data_ptr(), B, n, d,
                                      save.data_ptr(), sb, _lib.current_stream()), "dp_set2set_fwd")
        ctx.save_for_backward(emb, *ts, out)
        ctx.s2s_save = save
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        emb, w_ih, w_hh, b_ih, b_hh, wp, bp, out = ctx.saved_tensors
        B, n, d = emb.shape
        dout = dout.contiguous()
        demb = torch.empty_like(emb)
        grads = [torch.empty_like(t) for t in (w_ih, w_hh, b_ih, b_hh, wp, bp)]
        wsb = lib.dp_set2set_bwd_workspace_bytes(B, n, d)
        ws = torch.empty(wsb, device=emb.device, dtype=torch.uint8)
        save = ctx.s2s_save
        _lib.check(lib.dp_set2set_bwd(emb.data_ptr(), d, w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(),
                                      b_hh.data_ptr(), wp.data_ptr(), bp.data_ptr(), out.data_ptr(),
                                      dout.data_ptr(), demb.data_ptr(), d, *[g.data_ptr() for g in grads],
                                      B, n, d, save.data_ptr(), save.numel(), ws.data_ptr(), wsb,
                                      _lib.current_stream()), "dp_set2set_bwd")
        return (demb, *grads)


class Set2Set(nn.Module):
    def __init__(self, input_dim, hidden_dim, act_fn=nn.ReLU, num_layers=1):
        '''
        Args:
            input_dim: input dim of Set2Set.
            hidden_dim: the dim of set representation (= LSTM input dim) = 2 * input_dim in every
                caller of the reference (encoders.py:1142).
        '''
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        if hidden_dim <= input_dim:
            print('ERROR: Set2Set output_dim should be larger than input_dim')
        self.lstm_output_dim = hidden_dim - input_dim
        self.lstm = nn.LSTM(hidden_dim, input_dim, num_layers=num_layers, batch_first=True)
        self.pred = nn.Linear(hidden_dim, input_dim)
        self.act = act_fn()
        if num_layers != 1 or hidden_dim != 2 * input_dim or not isinstance(self.act, nn.ReLU):
            # set2set.py:53 concatenates q [d_lstm] and r [input_dim] into the LSTM input [hidden_dim]; with
            # lstm hidden = input_dim that only type-checks for hidden_dim == 2 * input_dim.
            raise NotImplementedError("the HIP Set2Set supports num_layers=1, hidden_dim == 2*input_dim, ReLU "
                                      "(the only configuration the reference instantiates)")

    def forward(self, embedding):
        l = self.lstm
        return _Set2SetFn.apply(embedding, l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0,
                                self.pred.weight, self.pred.bias)
