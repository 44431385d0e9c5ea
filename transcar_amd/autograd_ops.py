"""torch.autograd.Function wrappers over the training entry points of the C ABI
(include/transcar_hip.h, "Training").  The reference trains through plain
nn.Modules and ``loss.backward()`` (tools/train.py, mmcv runner), so autograd
IS its boundary for training; here every node's forward and backward is a HIP
kernel and torch only carries the graph, the saved activations and the
gradient accumulation.  fp32, contiguous, on the GPU; no CPU path.
"""
import ctypes as C

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from .ops import _chk, _p, _stream


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _zeros_like(t):
    return torch.zeros(t.shape, dtype=torch.float32, device=t.device)


class _Linear(Function):
    """y = act(x W^T + b), act in {none, ReLU}; x [..., K], W [N, K]."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        _chk(x, 'x'); _chk(weight, 'weight')
        K, N = x.shape[-1], weight.shape[0]
        M = x.numel() // K
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        L.check(L.lib().tc_linear_fwd(_p(x), None, _p(weight), _p(bias), None, _p(y),
                                      M, K, N, int(act), _stream()), 'tc_linear_fwd')
        ctx.act = int(act)
        ctx.save_for_backward(x, weight, y if act else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _c(dy)
        K, N = x.shape[-1], weight.shape[0]
        M = x.numel() // K
        lib = L.lib()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.check(lib.tc_linear_bwd_data(_p(dy), _p(y), None, _p(weight), None, _p(dx),
                                           M, K, N, 1.0, 0, _stream()), 'tc_linear_bwd_data')
        if ctx.needs_input_grad[1]:
            dw = _zeros_like(weight)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = torch.zeros(N, dtype=torch.float32, device=x.device)
            L.check(lib.tc_linear_bwd_weight(_p(x), _p(dy), _p(y), None, _p(dw), _p(db),
                                             M, K, N, 1.0, _stream()), 'tc_linear_bwd_weight')
        return dx, dw, db, None


class _GatedLinearResidual(Function):
    """y = res + (gate > 0 ? x W^T + b : 0)  (HEAD:581, the row-subset update)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, gate):
        _chk(x, 'x'); _chk(weight, 'weight'); _chk(res, 'res')
        K, N = x.shape[-1], weight.shape[0]
        M = x.numel() // K
        y = torch.empty_like(res)
        L.check(L.lib().tc_linear_gated_fwd(_p(x), _p(weight), _p(bias), _p(res), _p(gate),
                                            _p(y), M, K, N, _stream()), 'tc_linear_gated_fwd')
        ctx.save_for_backward(x, weight, gate)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, gate = ctx.saved_tensors
        dy = _c(dy)
        K, N = x.shape[-1], weight.shape[0]
        M = x.numel() // K
        lib = L.lib()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.check(lib.tc_linear_bwd_data(_p(dy), None, _p(gate), _p(weight), None, _p(dx),
                                           M, K, N, 1.0, 0, _stream()), 'tc_linear_bwd_data')
        if ctx.needs_input_grad[1]:
            dw = _zeros_like(weight)
            db = torch.zeros(N, dtype=torch.float32, device=x.device)
            L.check(lib.tc_linear_bwd_weight(_p(x), _p(dy), None, _p(gate), _p(dw), _p(db),
                                             M, K, N, 1.0, _stream()), 'tc_linear_bwd_weight')
        return dx, dw, db, (dy if ctx.needs_input_grad[3] else None), None


class _AddLayerNorm(Function):
    """y = LayerNorm(a (+ b)) * gamma + beta, optional ReLU (eps 1e-5, C = 256)."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, relu):
        _chk(a, 'a')
        Cd = a.shape[-1]
        y = torch.empty_like(a)
        L.check(L.lib().tc_add_layernorm_fwd(_p(a), _p(b), _p(gamma), _p(beta), _p(y),
                                             a.numel() // Cd, Cd, 1 if relu else 0, _stream()),
                'tc_add_layernorm_fwd')
        ctx.save_for_backward(a, b, gamma, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        a, b, gamma, y = ctx.saved_tensors
        dy = _c(dy)
        Cd = a.shape[-1]
        dz = torch.empty_like(a)
        want_w = ctx.needs_input_grad[2]
        dg = torch.zeros(Cd, dtype=torch.float32, device=a.device) if want_w else None
        dbt = torch.zeros(Cd, dtype=torch.float32, device=a.device) if want_w else None
        L.check(L.lib().tc_add_layernorm_bwd(_p(a), _p(b), _p(gamma), _p(dy), _p(y), _p(dz),
                                             _p(dg), _p(dbt), a.numel() // Cd, Cd, _stream()),
                'tc_add_layernorm_bwd')
        return (dz if ctx.needs_input_grad[0] else None,
                dz if (b is not None and ctx.needs_input_grad[1]) else None, dg, dbt, None)


class _RadarAttnCore(Function):
    """Distance-gated attention core (HEAD:549-579): qproj [B,Q,C] (unscaled),
    kv [B,T,2C]; gate geometry (centre, box, token xy) carries no gradient (the
    reference's masks are boolean).  Returns (attn_out [B,Q,C], hits [B,Q])."""

    @staticmethod
    def forward(ctx, qproj, kv, centre, ld_c, box, tokens, pad_mult, rmin, rmax, heads,
                drop=(0.0, 0, 0)):
        for n, t in (('qproj', qproj), ('kv', kv), ('centre', centre), ('box', box),
                     ('tokens', tokens)):
            _chk(t, n)
        B, Q, Cd = qproj.shape
        T = kv.shape[1]
        out = torch.empty_like(qproj)
        hits = torch.empty((B, Q), dtype=torch.int32, device=qproj.device)
        scale = 1.0 / float(Cd // heads) ** 0.5
        meta = (scale, int(ld_c), box.shape[-1], tokens.shape[-1], B, Q, T, Cd, int(heads),
                int(pad_mult), float(rmin), float(rmax))
        L.check(L.lib().tc_radar_attn_core_fwd(
            _p(qproj), scale, _p(kv), _p(centre), meta[1], _p(box), meta[2], _p(tokens), meta[3],
            B, Q, T, Cd, meta[8], meta[9], meta[10], meta[11], _p(out), _p(hits),
            float(drop[0]), int(drop[1]), int(drop[2]), _stream()),
            'tc_radar_attn_core_fwd')
        ctx.meta = meta
        ctx.drop = (float(drop[0]), int(drop[1]), int(drop[2]))
        ctx.save_for_backward(qproj, kv, centre, box, tokens, out)
        ctx.mark_non_differentiable(hits)
        return out, hits

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out, _d_hits):
        qproj, kv, centre, box, tokens, out = ctx.saved_tensors
        scale, ld_c, code, ld_xy, B, Q, T, Cd, heads, pad_mult, rmin, rmax = ctx.meta
        d_out = _c(d_out)
        dq = torch.empty_like(qproj)
        dkv = _zeros_like(kv)
        L.check(L.lib().tc_radar_attn_core_bwd(
            _p(qproj), scale, _p(kv), _p(centre), ld_c, _p(box), code, _p(tokens), ld_xy,
            B, Q, T, Cd, heads, pad_mult, rmin, rmax, _p(out), _p(d_out), _p(dq), _p(dkv),
            ctx.drop[0], ctx.drop[1], ctx.drop[2], _stream()), 'tc_radar_attn_core_bwd')
        return dq, dkv, None, None, None, None, None, None, None, None, None


class _Dropout(Function):
    """nn.Dropout in train mode with the library's counter-based mask of (seed, site):
    y = keep * x / (1 - p); the backward applies the same mask to the gradient."""

    @staticmethod
    def forward(ctx, x, p, seed, site):
        _chk(x, 'x')
        cols = x.shape[-1]
        out = torch.empty_like(x)
        ctx.args = (x.numel() // cols, cols, float(p), int(seed), int(site))
        L.check(L.lib().tc_dropout(_p(x), ctx.args[0], cols, ctx.args[2], ctx.args[3], ctx.args[4],
                                   _p(out), _stream()), 'tc_dropout')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        rows, cols, p, seed, site = ctx.args
        L.check(L.lib().tc_dropout(_p(dy), rows, cols, p, seed, site, _p(dx), _stream()), 'tc_dropout')
        return dx, None, None, None


class _BoxAddRef(Function):
    """box = reg_out; box[..., 0:2] += ref_xy; box[..., 4] += ref_z (HEAD:599-600,
    661-662, 719-720).  With ``prev_box`` the reference is the previous layer's
    box (xy = prev[0:2], z = prev[4]) and receives gradient (HEAD:615-617)."""

    @staticmethod
    def forward(ctx, reg_out, prev_box, add_ref):
        _chk(reg_out, 'reg_out')
        code = reg_out.shape[-1]
        M = reg_out.numel() // code
        box = torch.empty_like(reg_out)
        if prev_box is not None:
            _chk(prev_box, 'prev_box')
            xy, ld_xy, z, ld_z = prev_box.data_ptr(), code, prev_box.data_ptr() + 16, code
        else:
            _chk(add_ref, 'add_ref')
            xy, ld_xy, z, ld_z = add_ref.data_ptr(), 3, add_ref.data_ptr() + 8, 3
        L.check(L.lib().tc_box_add_ref_fwd(_p(reg_out), code, C.c_void_p(xy), ld_xy,
                                           C.c_void_p(z), ld_z, _p(box), M, _stream()),
                'tc_box_add_ref_fwd')
        ctx.with_prev = prev_box is not None
        return box

    @staticmethod
    @once_differentiable
    def backward(ctx, d_box):
        d_box = _c(d_box)
        d_prev = None
        if ctx.with_prev and ctx.needs_input_grad[1]:
            code = d_box.shape[-1]
            d_prev = torch.zeros_like(d_box)
            L.check(L.lib().tc_box_add_ref_bwd(_p(d_box), code, _p(d_prev),
                                               d_box.numel() // code, _stream()),
                    'tc_box_add_ref_bwd')
        return d_box, d_prev, None


def linear(x, weight, bias, act=0):
    return _Linear.apply(x, weight, bias, act)


def gated_linear_residual(x, weight, bias, res, gate):
    return _GatedLinearResidual.apply(x, weight, bias, res, gate)


def add_layernorm(a, b, gamma, beta, relu=False):
    return _AddLayerNorm.apply(a, b, gamma, beta, relu)


def dropout(x, p, seed, site):
    """Identity when p == 0."""
    return x if p <= 0.0 else _Dropout.apply(x, p, seed, site)


def radar_attn_core(qproj, kv, centre, ld_c, box, tokens, pad_mult, rmin, rmax, heads=8,
                    drop=(0.0, 0, 0)):
    return _RadarAttnCore.apply(qproj, kv, centre, ld_c, box, tokens, pad_mult, rmin, rmax, heads, drop)


def box_add_ref(reg_out, prev_box=None, add_ref=None):
    return _BoxAddRef.apply(reg_out, prev_box, add_ref)


def radar_reference_l1(ref, pc_range):
    """HEAD:544-547 / 596-598 (no gradient: the decoder is frozen)."""
    _chk(ref, 'ref')
    M = ref.numel() // 3
    cxy = torch.empty(ref.shape[:-1] + (2,), dtype=torch.float32, device=ref.device)
    addref = torch.empty_like(ref)
    L.check(L.lib().tc_radar_reference_l1(_p(ref), L.f6(pc_range), _p(cxy), _p(addref), M,
                                          _stream()), 'tc_radar_reference_l1')
    return cxy, addref
