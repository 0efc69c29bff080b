"""Differentiable forms of the unfused operators (SPEC.md §16).

``group_points`` / ``gather_points`` / ``max_pool_s`` as ``torch.autograd.Function``s whose forward
AND backward are this package's HIP kernels (float32 only): the classic unfused
``group -> shared MLP (any torch layers) -> max over nsample`` stack becomes trainable without a
PyTorch-side scatter.  The fused inference kernels (``PackedMLP.grouped``) have no backward.
"""
import torch

from . import ops
from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32(t, name, ndim):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != ndim:
        raise TypeError(f"{name}: expected a GPU float32 tensor with {ndim} dims (sad_amd has no CPU path)")
    return t.contiguous()


def group_points_grad(grad_out: torch.Tensor, idx: torch.Tensor, N: int, point_major: bool = False,
                      via_point_major: bool = True) -> torch.Tensor:
    """grad_out [B,C,M,S] (or [B,C,M] with idx [B,M]) -> grad_feat [B,C,N] (scatter-add), or
    [B,N,C] with ``point_major=True``.  By default the sum is formed point-major (contiguous float
    atomics, ~10x faster) and transposed at the end; ``via_point_major=False`` uses the direct
    channel-major scatter."""
    if grad_out.dim() == 3:
        grad_out, idx = grad_out.unsqueeze(-1), idx.unsqueeze(-1)
    grad_out = _f32(grad_out, "grad_out", 4)
    if idx.dtype != torch.int32 or not idx.is_cuda:
        raise TypeError("idx: expected a GPU int32 tensor")
    idx = idx.contiguous()
    B, C, M, S = grad_out.shape
    if tuple(idx.shape) != (B, M, S):
        raise ValueError("idx must be [B,M,S]")
    if point_major or via_point_major:
        g = torch.zeros((B, N, C), dtype=torch.float32, device=grad_out.device)
        check(lib().sad_group_points_grad_pm_f32(grad_out.data_ptr(), idx.data_ptr(), B, C, N, M, S, g.data_ptr(),
                                                 _stream()), "sad_group_points_grad_pm_f32")
        return g if point_major else g.transpose(1, 2).contiguous()
    g = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
    check(lib().sad_group_points_grad_f32(grad_out.data_ptr(), idx.data_ptr(), B, C, N, M, S, g.data_ptr(),
                                          _stream()), "sad_group_points_grad_f32")
    return g


class GroupPoints(torch.autograd.Function):
    """features [B,C,N] f32, idx [B,M,S] int32 -> [B,C,M,S]."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.N = features.shape[2]
        return ops.group_points(_f32(features, "features", 3), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return group_points_grad(grad_out, idx, ctx.N), None


class GatherPoints(torch.autograd.Function):
    """features [B,C,N] f32, idx [B,M] int32 -> [B,C,M]."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.N = features.shape[2]
        return ops.gather_points(_f32(features, "features", 3), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return group_points_grad(grad_out, idx, ctx.N), None


def max_pool_s_with_arg(x: torch.Tensor):
    """x [B,C,M,S] -> (out [B,C,M], arg [B,C,M] int32); ties -> lowest s."""
    x = _f32(x, "x", 4)
    B, C, M, S = x.shape
    out = torch.empty((B, C, M), dtype=torch.float32, device=x.device)
    arg = torch.empty((B, C, M), dtype=torch.int32, device=x.device)
    check(lib().sad_max_pool_s_f32(x.data_ptr(), B, C, M, S, out.data_ptr(), arg.data_ptr(), _stream()),
          "sad_max_pool_s_f32")
    return out, arg


class MaxPoolS(torch.autograd.Function):
    """x [B,C,M,S] -> max over S, gradient routed to the arg-max slot."""

    @staticmethod
    def forward(ctx, x):
        out, arg = max_pool_s_with_arg(x)
        ctx.save_for_backward(arg)
        ctx.S = x.shape[3]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        grad_out = _f32(grad_out, "grad_out", 3)
        B, C, M = grad_out.shape
        gx = torch.empty((B, C, M, ctx.S), dtype=torch.float32, device=grad_out.device)
        check(lib().sad_max_pool_s_grad_f32(grad_out.data_ptr(), arg.data_ptr(), B, C, M, ctx.S, gx.data_ptr(),
                                            _stream()), "sad_max_pool_s_grad_f32")
        return gx


group_points = GroupPoints.apply
gather_points = GatherPoints.apply
max_pool_s = MaxPoolS.apply
