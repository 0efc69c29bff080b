"""``sa_module``: multi-radius set abstraction (SPEC.md §7) on the fused HIP path.

Drop-in surface named by BASELINE.json ``north_star`` (the upstream reference,
``/root/reference/README.md:1-2``, defines none): ``forward(xyz [B,N,3], features [B,C,N]) ->
(new_xyz [B,M,3], new_features [B,C',M])``.  Internally features are kept point-major ([B,N,C]) so a
neighbour's feature row is one contiguous gather; ``forward_pm`` exposes that layout to callers that
chain stages (the detector), skipping the two transposes of the channel-major surface.

Pipeline per call: fps -> gather_xyz -> one multi-radius ball query (d2 evaluated once for all
radii) -> per branch one fused gather+MLP+max kernel writing straight into its slice of the
concatenated [B,M,sum C_b] buffer -> optional aggregation layer.  The grouped tensor
[B,C+3,M,S] is never materialised.
"""
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from .config import SAStage
from .synth import make_mlp_weights


class SAModuleMSG(nn.Module):
    """Multi-scale-grouping set abstraction.  ``weights`` = {"b<i>": [(W,b),...], "agg": [(W,b)]}
    as numpy arrays (BatchNorm folded); seeded Kaiming-uniform weights are drawn when omitted."""

    def __init__(self, in_channels: int, stage: SAStage, device, weights: Optional[dict] = None,
                 seed: int = 0, name: str = "sa", dtype: str = "f32"):
        """``dtype``: "f32" (SPEC.md §6, bit-exact fmaf chains) or "bf16" (SPEC.md §14: bf16 MFMA
        MLPs, float32 accumulation, bf16 stage outputs; index operators unchanged)."""
        super().__init__()
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.dtype = dtype
        mlp_cls = ops.PackedMLP if dtype == "f32" else ops.PackedMLPBf16
        self.stage = stage
        self.in_channels = in_channels
        self.device = torch.device(device)
        if weights is None:
            rng = np.random.default_rng(seed)
            weights = {f"b{i}": make_mlp_weights([in_channels + 3] + list(m), rng)
                       for i, m in enumerate(stage.mlps)}
            if stage.agg:
                weights["agg"] = make_mlp_weights([sum(m[-1] for m in stage.mlps), stage.agg], rng)
        self.branches = [mlp_cls(weights[f"b{i}"], True, self.device, name=f"{name}.b{i}")
                         for i in range(len(stage.mlps))]
        self.cat_channels = sum(m[-1] for m in stage.mlps)
        self.agg = (mlp_cls(weights["agg"], False, self.device, name=f"{name}.agg")
                    if stage.agg else None)
        self.out_channels = stage.agg if stage.agg else self.cat_channels

    def sample(self, xyz: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """fps + gather: (fps_idx [B,M], new_xyz [B,M,3]).  Depends on coordinates only."""
        fidx = ops.fps(xyz, self.stage.npoint)
        return fidx, ops.gather_xyz(xyz, fidx)

    def can_split(self, B: int, N: int, M: int, feat: Optional[torch.Tensor] = None, feat_dtype=None) -> bool:
        """Split pooling (include/sad_amd.h ``sad_mlp_bf16_args.cont``: bf16 pooled rows, plain stores, half the bytes between the
        branches and the aggregation layer) applies: a bf16 stage with an aggregation layer on the row-streaming kernel and every
        branch on the register-resident chain.  ``feat`` / ``feat_dtype`` as for ``query``."""
        if self.dtype != "bf16" or self.agg is None or ops.AUTOTUNE or not ops.SPLIT_POOL:
            return False
        if self.cat_channels % 8 or any(m.out_channels % 16 for m in self.branches) or len(self.branches) > 4:
            return False
        if not self.agg.takes_pooled(B * M, self.agg.out_channels):
            return False
        if feat_dtype is None:
            feat_dtype = torch.bfloat16
        return all(mlp.preferred_geometry == 2 and mlp.wants_prescan(B, N, M, s, self.cat_channels, self.in_channels, feat=feat, feat_dtype=feat_dtype)
                   for mlp, s in zip(self.branches, self.stage.nsamples))

    def split_buffers(self, B: int, M: int, device):
        """(cat [B,M,sum C_b] bfloat16, [continuation rows per branch]) of a split-pooled call, uninitialised."""
        cat = ops._empty((B, M, self.cat_channels), dtype=torch.bfloat16, device=device)
        conts = [ops.cont_buffer(B, M, s, mlp.out_channels, device) for mlp, s in zip(self.branches, self.stage.nsamples)]
        return cat, conts

    def query(self, xyz: torch.Tensor, new_xyz: torch.Tensor, radius_pc: Optional[torch.Tensor] = None,
              radii: Optional[Sequence[float]] = None, prescan: bool = False, cat: Optional[torch.Tensor] = None,
              feat: Optional[torch.Tensor] = None, feat_dtype=None, conts=None):
        """The stage's multi-radius ball query: ([idx_b [B,M,S_b]], [cnt_b [B,M]]) and, with ``prescan``,
        the row-packing tables of the branches as a third element.  ``cat`` (with ``prescan``): the stage's
        UNINITIALISED [B,M,sum C_b] pooling buffer — the scan prepares the slices of the branches that take a table
        (ops.rowscan_multi ``outs``), the slices of the others are zero-filled here; pass the same buffer to
        ``group_and_pool(cat=...)``.  ``feat``: the feature tensor ``group_and_pool`` will get (its stride and alignment
        decide which kernel runs, hence whether a table is wanted); when it does not exist yet (a caller that queries ahead
        on another stream) leave it None and it is taken to be a fresh contiguous [B,N,C] tensor of ``feat_dtype``
        (default: float32, or bfloat16 for a bf16 module).  ``conts`` (with a bfloat16 ``cat``, both from ``split_buffers``): split
        pooling — ask ``can_split`` first."""
        st = self.stage
        idxs, cnts = ops.ball_query_multi(radii if radii is not None else st.radii, st.nsamples, xyz, new_xyz,
                                          radius_pc, return_counts=True)
        if prescan:
            # the row-packing scan needs only the query's output: run it here (on the sampling stream when the
            # detector overlaps), so the MLP stream launches no small latency-bound kernels before its chains;
            # only for the branches whose kernel consumes such a table (the tiled kernel packs for itself)
            B, N, M = xyz.shape[0], xyz.shape[1], new_xyz.shape[1]
            if feat_dtype is None:
                feat_dtype = torch.float32 if self.dtype == "f32" else torch.bfloat16
            pick = [i for i, (mlp, s) in enumerate(zip(self.branches, st.nsamples))
                    if mlp.wants_prescan(B, N, M, s, self.cat_channels, self.in_channels, feat=feat, feat_dtype=feat_dtype)]
            wss = [None] * len(idxs)
            outs = None
            if conts is not None:
                if cat is None or cat.dtype != torch.bfloat16 or len(pick) != len(self.branches):
                    raise RuntimeError("split pooling needs a bfloat16 cat and every branch on the register-resident chain (can_split)")
                offs = [0]
                for mlp in self.branches:
                    offs.append(offs[-1] + mlp.out_channels)
                outs = [(cat, offs[i], self.branches[i].out_channels, conts[i]) for i in pick]
            elif cat is not None:
                offs = [0]
                for mlp in self.branches:
                    offs.append(offs[-1] + mlp.out_channels)
                outs = [(cat, offs[i], self.branches[i].out_channels) for i in pick]
                for i in range(len(self.branches)):
                    if i not in pick:
                        ops._unrecordable("query: zero fill of a branch's pooling slice")
                        cat[:, :, offs[i]:offs[i + 1]].zero_()
            if pick:
                for i, w in zip(pick, ops.rowscan_multi([idxs[i] for i in pick], [cnts[i] for i in pick], N, outs)):
                    wss[i] = w
            return idxs, cnts, wss
        return idxs, cnts

    def group_and_pool(self, xyz: torch.Tensor, feat_pm: Optional[torch.Tensor],
                       new_xyz: torch.Tensor, radius_pc: Optional[torch.Tensor] = None,
                       radii: Optional[Sequence[float]] = None, keep: Optional[dict] = None,
                       query=None, cat: Optional[torch.Tensor] = None, conts=None) -> torch.Tensor:
        """ball query + fused MLP/max for every branch + aggregation -> [B,M,C'] point-major.
        ``query`` = (idxs, cnts) from an earlier ``self.query(...)`` (the ball query needs coordinates
        only, so a caller may run it ahead on another stream)."""
        st = self.stage
        B, M = new_xyz.shape[0], new_xyz.shape[1]
        if query is None and cat is None and self.can_split(B, xyz.shape[1], M, feat=feat_pm):
            cat, conts = self.split_buffers(B, M, xyz.device)
        # (own query: with the row-packing scan behind it, as the detector's sampling stream does — the MLP dispatch is
        # then the same launches in every mode)
        q = query if query is not None else self.query(xyz, new_xyz, radius_pc, radii, prescan=not ops.AUTOTUNE, feat=feat_pm,
                                                       cat=cat if conts is not None else None, conts=conts)
        idxs, cnts = q[0], q[1]
        wss = q[2] if len(q) > 2 else [None] * len(idxs)
        if keep is not None:
            keep["ball_idx"] = idxs
        if cat is None:      # ``cat``: a caller-provided [B,M,sum C_b] float32 buffer, ZERO or prepared by ``query(prescan=True, cat=cat)``
            ops._unrecordable("group_and_pool: zero-filled pooling buffer")
            cat = torch.zeros((B, M, self.cat_channels), dtype=torch.float32, device=xyz.device)
        calls, off = [], 0               # all branches in one dispatch (ops.grouped_multi)
        for bi, (mlp, idx, cnt, ws) in enumerate(zip(self.branches, idxs, cnts, wss)):
            calls.append((mlp, xyz, feat_pm, new_xyz, idx, cat, off, cnt, ws) + ((conts[bi],) if conts is not None else ()))
            off += mlp.out_channels
        ops.grouped_multi(calls)
        if conts is not None:        # split-pooled rows: the aggregation layer takes the maximum with the continuation rows as it reads
            pool = [(ws, cont, s, mlp.out_channels) for ws, cont, s, mlp in zip(wss, conts, st.nsamples, self.branches)]
            return self.agg.rows(cat, out_dtype=torch.bfloat16, pool=pool)
        if self.agg is None:
            return cat
        if self.dtype == "bf16":     # a stage output that feeds another stage is stored as bf16 (SPEC §14)
            return self.agg.rows(cat, out_dtype=torch.bfloat16)
        return self.agg.rows(cat)

    def forward_pm(self, xyz: torch.Tensor, feat_pm: Optional[torch.Tensor]
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Point-major fast path: feat_pm [B,N,C] -> (new_xyz [B,M,3], new_feat_pm [B,M,C'])."""
        _, new_xyz = self.sample(xyz)
        return new_xyz, self.group_and_pool(xyz, feat_pm, new_xyz)

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor]
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Drop-in surface: features [B,C,N] channel-major -> (new_xyz, new_features [B,C',M])."""
        feat_pm = features.transpose(1, 2).contiguous() if features is not None else None
        new_xyz, out_pm = self.forward_pm(xyz, feat_pm)
        return new_xyz, out_pm.transpose(1, 2).contiguous()


class SAModule(SAModuleMSG):
    """Single-radius set abstraction = the one-branch case (BASELINE.json configs[0])."""

    def __init__(self, in_channels: int, npoint: int, radius: float, nsample: int,
                 mlp: Sequence[int], device, weights: Optional[dict] = None, seed: int = 0):
        super().__init__(in_channels, SAStage(npoint, (radius,), (nsample,), (tuple(mlp),), 0),
                         device, weights, seed)


def sa_module(xyz: torch.Tensor, features: Optional[torch.Tensor], npoint: int, radius: float,
              nsample: int, layers) -> Tuple[torch.Tensor, torch.Tensor]:
    """Functional form: one single-radius SA layer with explicit ``layers`` [(W,b),...]."""
    c = 0 if features is None else features.shape[1]
    mod = SAModule(c, npoint, radius, nsample, [w.shape[0] for w, _ in layers], xyz.device,
                   weights={"b0": layers})
    return mod(xyz, features)
