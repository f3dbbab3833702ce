"""PyTorch-CPU restatement of the lifting forward pass - TEST INFRASTRUCTURE ONLY.

A functional (torch.nn.functional) port of the reference module graph, op for op, used for
(a) bench.py's ``cpu_baseline`` leg: it is what "the reference's CPU path" costs on the bench
    host - the same ATen conv1d / batch_norm / addmm kernels the reference would run - without
    shipping any reference source, and
(b) a second, independent checker beside the C oracle (tests/test_oracle.py pins it against the
    reference-generated golden fixtures).
Rules: oracle/ray3d_oracle.h header.  The product package never imports this file.

Cited lines are lib/model/rie.py unless noted.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from ray3d_amd.spec import BRANCHES, GROUPS, OUTPUT_ORDER, LiftConfig


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training=False, eps=1e-5)


def temporal_block(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, nlev: int, tap: int = 1) -> torch.Tensor:
    """:85-105 in the strided (Optimize1f) form; x (B, Cin, RF) -> (B, latent).  tap = 1: res is the centre frame
    (:94); tap = 2: the causal, dilated model's residual (:92 with shift == pad), the last frame of each triple."""
    x = F.leaky_relu(_bn(F.conv1d(x, sd[p + ".expand_conv.weight"], stride=3), sd, p + ".expand_bn"), 0.2)
    for i in range(nlev - 1):
        res = x[:, :, tap::3]                                                            # :94 / :92
        x = F.leaky_relu(_bn(F.conv1d(x, sd["%s.layers_conv.%d.weight" % (p, 2 * i)], stride=3),
                             sd, "%s.layers_bn.%d" % (p, 2 * i)), 0.2)                   # :96
        x = res + F.leaky_relu(_bn(F.conv1d(x, sd["%s.layers_conv.%d.weight" % (p, 2 * i + 1)]),
                                   sd, "%s.layers_bn.%d" % (p, 2 * i + 1)), 0.2)         # :97
    x = F.conv1d(x, sd[p + ".shrink.weight"], sd[p + ".shrink.bias"])                    # :99
    return x[:, :, 0]


def temporal_block_dense(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, nlev: int, causal: bool) -> torch.Tensor:
    """:85-105 in the un-optimised form with dense=True (:49-53): stride-1 convolutions of 3, then 2*3^i + 1 taps;
    res = x[:, :, pad+shift : T-pad+shift] (:91-92), pad = 3^i, shift = pad for causal models."""
    x = F.leaky_relu(_bn(F.conv1d(x, sd[p + ".expand_conv.weight"]), sd, p + ".expand_bn"), 0.2)
    d = 3
    for i in range(nlev - 1):
        shift = d if causal else 0
        res = x[:, :, d + shift: x.shape[2] - d + shift]
        x = F.leaky_relu(_bn(F.conv1d(x, sd["%s.layers_conv.%d.weight" % (p, 2 * i)]), sd, "%s.layers_bn.%d" % (p, 2 * i)), 0.2)
        x = res + F.leaky_relu(_bn(F.conv1d(x, sd["%s.layers_conv.%d.weight" % (p, 2 * i + 1)]),
                                   sd, "%s.layers_bn.%d" % (p, 2 * i + 1)), 0.2)
        d *= 3
    x = F.conv1d(x, sd[p + ".shrink.weight"], sd[p + ".shrink.bias"])
    assert x.shape[2] == 1
    return x[:, :, 0]


def fc_block(sd, p: str, x: torch.Tensor, nblocks: int) -> torch.Tensor:
    """:159-169 with the residual units of :122-135."""
    x = F.leaky_relu(_bn(F.linear(x, sd[p + ".fc_1.weight"], sd[p + ".fc_1.bias"]), sd, p + ".bn_1"), 0.2)
    for n in range(nblocks):
        q = "%s.layers.%d" % (p, n)
        y = F.leaky_relu(_bn(F.linear(x, sd[q + ".w1.weight"], sd[q + ".w1.bias"]), sd, q + ".batch_norm1"), 0.2)
        y = F.leaky_relu(_bn(F.linear(y, sd[q + ".w2.weight"], sd[q + ".w2.bias"]), sd, q + ".batch_norm2"), 0.2)
        x = x + y
    return F.linear(x, sd[p + ".fc_2.weight"], sd[p + ".fc_2.bias"])


def embedding(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """lib/model/embedding.py:15-18 (nn.LeakyReLU() default slope 0.01)."""
    x = F.leaky_relu(_bn(F.linear(x, sd[p + ".w1.weight"], sd[p + ".w1.bias"]), sd, p + ".b1"), 0.01)
    return F.leaky_relu(_bn(F.linear(x, sd[p + ".w2.weight"], sd[p + ".w2.bias"]), sd, p + ".b2"), 0.01)


def _encode(x: torch.Tensor, cfg: LiftConfig):
    """:290-304 -> (in_current (B, J*F), channels-first x/diff/diff_t of shape (B, J*F, RF))."""
    B, RF, J, Fd = x.shape
    tcur = RF // Fd                                                    # quirk Q1
    in_current = x[:, tcur].reshape(B, -1)
    xc = x.reshape(B, RF, J * Fd).permute(0, 2, 1)
    diff = xc - xc[:, 0:Fd, :].repeat(1, J, 1)
    diff_t = xc - xc[:, :, tcur:tcur + 1]
    return in_current, xc, diff, diff_t


def _rows(joints, Fd):
    return [j * Fd + f for j in joints for f in range(Fd)]


def forward(cfg: LiftConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor,
            param: Optional[torch.Tensor]) -> torch.Tensor:
    """RIEModel.forward (:284-434) / RIETrajectoryModel.forward (:518-559), eval mode."""
    B, J, Fd, L = x.shape[0], cfg.num_joints, cfg.in_features, len(cfg.filter_widths)
    in_current, xc, diff, diff_t = _encode(x, cfg)
    x_global = fc_block(sd, "GlobalInfo", in_current, 2)
    emb = embedding(sd, "embedder", param) if cfg.camera_embedding else None
    def tblock(prefix, inp):
        if cfg.dense_convs:
            return temporal_block_dense(sd, prefix, inp, L, cfg.causal)
        return temporal_block(sd, prefix, inp, L, cfg.residual_tap)
    if cfg.kind == "trj":
        local = tblock("LocalLayer", torch.cat((xc, diff, diff_t), dim=1))                                       # :540-546
        feats = [local, x_global] + ([emb] if emb is not None else [])
        return fc_block(sd, "Integration", torch.cat(feats, dim=1), 1).view(B, 1, 1, 3)
    locals_ = []
    for b in BRANCHES:                                                                            # :306-369
        idx = _rows(GROUPS[J][b], Fd)
        locals_.append(tblock("LocalLayer_" + b, torch.cat((xc[:, idx], diff[:, idx], diff_t[:, idx]), dim=1)))
    dec = {}
    for i, b in enumerate(BRANCHES):
        feats = [locals_[i]]
        if cfg.stage != 1:                                                                        # :390-394
            others = torch.cat([locals_[k] for k in range(5) if k != i], dim=1)
            feats.append(fc_block(sd, "FuseBlocks.%d" % i, others, 1))
        feats.append(x_global)
        if emb is not None:
            feats.append(emb)
        dec[b] = fc_block(sd, "Integration_" + b, torch.cat(feats, dim=1), 1).view(B, -1, 3)      # :409-424
    out = torch.stack([dec[b][:, i] for (b, i) in OUTPUT_ORDER[J]], dim=1)                        # :426-431
    return out.view(B, 1, J, 3)
